"""Aggregate a rocprofv3 counter_collection.csv per kernel (sum of each counter over all dispatches), then per PHASE of
the attack step:

  set-up     encoder kernels (enc_*, rp_layer_tail16_kernel<1, ..>, rp_lin16_kernel<0> / rp_gemm of the encoder's FFN) and the
             like / unlike scorer embedding (scorer_kernel<false>)
  edit loop  the scorer edit step (scorer_kernel<true>), every decode kernel (embed, QKV, attention, half-layer kernel <0>,
             generator screening / argmax / finish, cross-attention constants), accept_kernel and the torch glue between them

MFMA-busy of a group = sum SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x sum GRBM_GUI_ACTIVE / 8): GRBM_GUI_ACTIVE is summed
over the 8 XCDs, SQ_VALU_MFMA_BUSY_CYCLES over all SIMDs (profiles/README.md).

    python tools/summarize_pmc.py <counter_collection.csv> [--phases]
"""
import collections
import csv
import re
import sys

SETUP = re.compile(r"^(void )?(enc_|rp_layer_tail16_kernel<1[,>]|scorer(16)?_kernel<false>|rp_lin16_kernel<0>|rp_lin_kernel<0|enc_window)")


def phase(kernel):
    return "set-up" if SETUP.search(kernel) else "edit loop"


def main():
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    for r in csv.DictReader(open(sys.argv[1])):
        agg[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[(r["Kernel_Name"], r["Counter_Name"])] += 1
    names = sorted({c for v in agg.values() for c in v})
    w = csv.writer(sys.stdout)
    busy = lambda v: (100.0 * v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * v["GRBM_GUI_ACTIVE"] / 8.0)) if v.get("GRBM_GUI_ACTIVE") else float("nan")
    w.writerow(["kernel", "phase", "dispatches", "mfma_busy_pct"] + names)
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", max(kv[1].values()))):
        w.writerow([k, phase(k), max(calls[(k, c)] for c in v), "%.1f" % busy(v)] + ["%.6g" % v.get(c, 0.0) for c in names])
    groups = collections.defaultdict(lambda: collections.defaultdict(float))
    for k, v in agg.items():
        for g in (phase(k), "whole run"):
            for c, x in v.items():
                groups[g][c] += x
    for g in ("set-up", "edit loop", "whole run"):
        v = groups[g]
        w.writerow(["== " + g, g, "", "%.1f" % busy(v)] + ["%.6g" % v.get(c, 0.0) for c in names])
    if len(sys.argv) > 2:      # second argument: a JSON file for bench.py (whole_step.mfma_busy_pct)
        import json
        tot = groups["whole run"].get("GRBM_GUI_ACTIVE", 0.0) or 1.0
        json.dump({"mfma_busy_pct": {g: round(busy(groups[g]), 2) for g in ("set-up", "edit loop", "whole run")},
                   "gpu_active_share_pct": {g: round(100.0 * groups[g].get("GRBM_GUI_ACTIVE", 0.0) / tot, 2) for g in ("set-up", "edit loop")},
                   "definition": "sum SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x sum GRBM_GUI_ACTIVE / 8) over the kernels of the phase; "
                                 "one rocprofv3 --pmc pass over one step of the default bench (tools/profile_bench.sh)"},
                  open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
