"""Deterministic, reference-free weight recipe for shape/stress configurations that ship no checkpoint
(amazon: no AE / Siamese / Predictor weights exist in the reference).  The SAME function runs in the
fixture generator (where the result is loaded into the reference's make_model) and in the tests on
the GPU box (where it feeds the HIP path), so the two sides see bit-identical weights without
shipping a 55 MB file.  numpy Generator streams are stable across platforms.

Layout = the state-dict keys of make_model (model.py:302-324).  Matrices ~ U(-a, a) with the Glorot
bound a = sqrt(6/(fan_in+fan_out)) (model.py:321-323) times `gain`; LayerNorm a_2 = 1, b_2 = 0;
biases ~ U(-0.05, 0.05).  A gain > 1 on the embedding/generator keeps the random decoder from
collapsing onto one token."""
import numpy as np


def ae_state_dict(vocab, n_layers=2, d=256, d_ff=1024, seed=0, gain=1.0, emb_gain=1.0):
    rng = np.random.default_rng(seed)
    sd = {}

    def mat(name, out_f, in_f, g=gain):
        a = np.sqrt(6.0 / (in_f + out_f)) * g
        sd[name] = rng.uniform(-a, a, (out_f, in_f)).astype(np.float32)

    def vec(name, n):
        sd[name] = rng.uniform(-0.05, 0.05, n).astype(np.float32)

    def ln(prefix):
        sd[prefix + ".a_2"] = np.ones(d, np.float32)
        sd[prefix + ".b_2"] = np.zeros(d, np.float32)

    def attn(prefix):
        for i in range(4):
            mat(prefix + ".linears.%d.weight" % i, d, d)
            vec(prefix + ".linears.%d.bias" % i, d)

    def ff(prefix):
        mat(prefix + ".w_1.weight", d_ff, d)
        vec(prefix + ".w_1.bias", d_ff)
        mat(prefix + ".w_2.weight", d, d_ff)
        vec(prefix + ".w_2.bias", d)

    for l in range(n_layers):
        p = "encoder.layers.%d." % l
        attn(p + "self_attn")
        ff(p + "feed_forward")
        ln(p + "sublayer.0.norm")
        ln(p + "sublayer.1.norm")
    ln("encoder.norm")
    for l in range(n_layers):
        p = "decoder.layers.%d." % l
        attn(p + "self_attn")
        attn(p + "src_attn")
        ff(p + "feed_forward")
        for k in range(3):
            ln(p + "sublayer.%d.norm" % k)
    ln("decoder.norm")
    mat("src_embed.0.lut.weight", vocab, d, emb_gain)
    sd["tgt_embed.0.lut.weight"] = sd["src_embed.0.lut.weight"]
    mat("generator.proj.weight", vocab, d, emb_gain)
    vec("generator.proj.bias", vocab)
    return sd
