"""BASELINE config 5 on one GPU: in-batch InfoNCE, B = 8192, 11 views (one rank's share when run with
--rows 1024: its [1024, 8192] row block).  HIP-event timed; flops: fwd 2*Bq*Bk*256, bwd 3x that."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from style_transfer_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=8192)
ap.add_argument("--rows", type=int, default=8192)
ap.add_argument("--views", type=int, default=11)
ap.add_argument("--format", default="f16x3", choices=["f16x3", "f32"])
a = ap.parse_args()
ops.simloss_format(a.format)
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
ori = (torch.rand(a.rows, 256, device=dev) * 10)
like = (torch.rand(a.views, a.B, 256, device=dev) * 10)
g = torch.full((a.rows,), 1.0 / (a.views * a.B), device=dev)


def run():
    for v in range(a.views):
        loss, lse = ops.simloss_fwd(ori, like[v], 0.1, 0)
        ops.simloss_bwd(ori, like[v], lse, g, 0.1, 0)


for _ in range(2):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    run()
e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 3 * 1e-3
fl = a.views * 4 * 2.0 * a.rows * a.B * 256
print("simloss %s fwd+bwd rows=%d B=%d views=%d: %.2f ms  %.1f TFLOP/s algorithmic (f32 MFMA peak 157.3, f16 MFMA peak 2516.6; the f16x3 "
      "backward runs 5 GEMM-sized products where the f32 form runs 4)" % (a.format, a.rows, a.B, a.views, t * 1e3, fl / t / 1e12))
