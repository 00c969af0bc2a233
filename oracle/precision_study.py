"""Study (test infrastructure, not product): how many greedy-decode tokens change when every Linear of the
decoder is evaluated with split low-precision operands instead of f32.  Emulates on the CPU what a k-term
split MFMA product computes:  x = xh + xl (+ residual),  w = wh + wl,  x.w ~ xh.wh + xh.wl + xl.wh.

  python oracle/precision_study.py [--n 512]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import st_oracle as O  # noqa: E402

F32 = np.float32


def to_bf16(x):
    u = np.ascontiguousarray(x, F32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(F32).reshape(x.shape)


def to_f16(x):
    return np.asarray(x, F32).astype(np.float16).astype(F32)


def split(x, rnd):
    h = rnd(x)
    return h, rnd((x - h).astype(F32))


def round_mantissa(x, bits):
    """round f32 to `bits` explicit mantissa bits (f32 exponent range kept): models an 8-bit residual"""
    u = np.ascontiguousarray(x, F32).view(np.uint32).astype(np.uint64)
    sh = 23 - bits
    r = ((u + (1 << (sh - 1)) - 1 + ((u >> sh) & 1)) >> sh) << sh
    return r.astype(np.uint32).view(F32).reshape(np.shape(x))


def to_e4m3(x):
    """round to OCP fp8 e4m3fn (3 mantissa bits, exponent bias 7, subnormals down to 2^-9, saturating at 448): what
    v_cvt_pk_fp8_f32 stores on gfx950"""
    x = np.asarray(x, F32)
    a = np.minimum(np.abs(x), F32(448.0))
    e = np.floor(np.log2(np.maximum(a, F32(2.0 ** -20))))
    e = np.maximum(e, -6.0)                      # below 2^-6 the spacing stays 2^-9 (subnormals)
    q = np.exp2(e - 3.0)
    r = np.round(a / q) * q                      # round half to even (np.round)
    return (np.sign(x) * np.minimum(r, 448.0)).astype(F32)


def make_linear(mode, sd):
    cache = {}
    if mode.startswith("kv:"):
        # exact f32 products everywhere; only what is STORED in the decoder's K/V cache is rounded:
        # kv:f16 = f16 value; kv:bf16; kv:f16+m3 = f16 value + residual with a 3-bit mantissa (3 bytes per element)
        inner = make_linear("f32", sd)
        kind = mode[3:]

        def lin_kv(x, sd_, prefix):
            y = inner(x, sd_, prefix)
            if prefix.startswith("decoder.") and (prefix.endswith("self_attn.linears.1") or prefix.endswith("self_attn.linears.2")):
                if kind == "f16":
                    y = to_f16(y)
                elif kind == "bf16":
                    y = to_bf16(y)
                elif kind == "f16+e4m3":         # the packed cache: f16 value + fp8 residual scaled by 2^12 (3 bytes per element)
                    h = to_f16(y)
                    y = (h + to_e4m3((y - h).astype(F32) * F32(4096.0)) / F32(4096.0)).astype(F32)
                elif kind.startswith("f16+m"):
                    h = to_f16(y)
                    y = (h + round_mantissa((y - h).astype(F32), int(kind[5:]))).astype(F32)
                else:
                    raise ValueError(mode)
            return y
        return lin_kv

    def lin(x, sd_, prefix):
        w = sd_[prefix + ".weight"]
        b = sd_[prefix + ".bias"]
        if mode == "f32":
            return (x @ w.T + b).astype(F32)
        if mode == "f64":
            return (x.astype(np.float64) @ w.T.astype(np.float64) + b).astype(F32)
        rnd = to_f16 if mode.startswith("f16") else to_bf16
        if prefix not in cache:
            cache[prefix] = split(w, rnd)
        wh, wl = cache[prefix]
        xh, xl = split(x.astype(F32), rnd)
        if mode.endswith("x1"):
            y = xh @ wh.T
        elif mode.endswith("x3"):
            y = xh @ wh.T + (xh @ wl.T + xl @ wh.T)
        else:
            raise ValueError(mode)
        return (y + b).astype(F32)

    return lin


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--modes", default="f32,f16x3,bf16x3,f16x1,bf16x1")
    a = ap.parse_args()
    from safetensors.numpy import load_file
    here = os.path.dirname(os.path.abspath(__file__))
    sd = load_file(os.path.join(here, "..", "tests", "golden", "surrogate_ae_yelp.safetensors"))
    sd = {k: v.astype(F32) for k, v in sd.items()}
    g = np.load(os.path.join(here, "..", "tests", "golden", "attack_notnll.npz"))
    lat = g["latent"].reshape(-1, 256)[: a.n]
    orig = O.linear
    O.linear = make_linear("f64", sd)
    ref_ids, ref_gap = O.greedy_decode(sd, lat, early_exit=True, want_gap=True)
    ntok = int((np.cumsum((ref_ids == 3) | (ref_ids == 0), 1) - ((ref_ids == 3) | (ref_ids == 0)) == 0).sum())
    print("decodes %d, tokens until EOS %d, min top-2 gap %.3g" % (lat.shape[0], ntok, ref_gap[ref_gap > 0].min()))
    for mode in a.modes.split(","):
        O.linear = make_linear(mode, sd)
        ids, gap = O.greedy_decode(sd, lat, early_exit=True, want_gap=True)
        bad = (ids != ref_ids).any(1)
        first = [(int(r), int(np.argmax(ids[r] != ref_ids[r]))) for r in np.where(bad)[0]]
        gaps = [float(ref_gap[r, c]) for r, c in first]
        # logit error at step 0
        print("%-7s decodes that differ: %d / %d   ref top-2 gaps at first divergence: %s"
              % (mode, int(bad.sum()), lat.shape[0], ", ".join("%.2g" % x for x in gaps[:12])))
    O.linear = orig


if __name__ == "__main__":
    main()
