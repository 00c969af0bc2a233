"""Micro-benchmarks of single kernels through the C-ABI (HIP-event timed, random data)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from style_transfer_amd import ops

dev = torch.device("cuda", 0)


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def argmax_bench(V=9339, sizes=(64, 256, 1024, 2048, 4096, 8192, 16384, 32768, 65536)):
    W = torch.randn(V, 256, device=dev) * 0.1
    b = torch.randn(V, device=dev)
    la, lb = torch.randn(256, device=dev), torch.randn(256, device=dev)
    for n in sizes:
        h = torch.randn(n, 256, device=dev)
        Wp = ops.pack_weight(W)
        t = timeit(lambda: ops.gemm_argmax(h, None, b, la, lb, packed=Wp))
        print("gemm_argmax n=%6d V=%d: %8.1f us  %6.1f TFLOP/s" % (n, V, t * 1e6, 2.0 * n * 256 * V / t / 1e12), flush=True)


if __name__ == "__main__":
    V = int(sys.argv[1]) if len(sys.argv) > 1 else 9339
    if len(sys.argv) > 2:
        argmax_bench(V, tuple(int(x) for x in sys.argv[2:]))
    else:
        argmax_bench(V)
