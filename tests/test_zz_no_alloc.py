"""GPU: the allocation-free entry point st_gemm_argmax_noalloc (static device scratch inside the library) against
st_gemm_argmax (caller-owned scratch) through the C-ABI: identical ids / best / gap, on two calls in a row, the
second with a smaller input (the shared scratch keeps the first call's contents), in both GEMM formats; sizes
beyond the static scratch are rejected."""
import ctypes as C
import gc

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda", 0)


@pytest.mark.parametrize("fmt", ["f32", "f16x3"])
def test_noalloc_argmax_matches_scratch_version_twice(dev, fmt):
    from style_transfer_amd import _lib, ops
    L = _lib.lib()
    rng = np.random.default_rng(7)
    V = 9339
    W = torch.from_numpy((rng.standard_normal((V, 256)) * 0.2).astype(np.float32)).to(dev)
    b = torch.from_numpy((rng.standard_normal(V) * 0.1).astype(np.float32)).to(dev)
    ln_a = torch.from_numpy((1.0 + 0.1 * rng.standard_normal(256)).astype(np.float32)).to(dev)
    ln_b = torch.from_numpy((0.1 * rng.standard_normal(256)).astype(np.float32)).to(dev)
    packed = ops.pack_weight(W, fmt)
    fid = _lib.GEMM_FORMATS[fmt]
    n_max = 5000
    # every buffer of the no-alloc calls exists before the first of them
    ids = torch.empty(n_max, dtype=torch.int32, device=dev)
    best = torch.empty(n_max, dtype=torch.float32, device=dev)
    gap = torch.empty(n_max, dtype=torch.float32, device=dev)
    hs = [torch.from_numpy(rng.standard_normal((n, 256)).astype(np.float32)).to(dev) for n in (n_max, 777)]
    refs = [ops.gemm_argmax(h, None, b, ln_a, ln_b, packed=packed, gemm_format=fmt) for h in hs]
    torch.cuda.synchronize(dev)
    for h, (rid, rbest, rgap) in zip(hs, refs):   # two calls in a row, the second smaller
        n = h.shape[0]
        ids.fill_(-7); best.fill_(float("nan")); gap.fill_(float("nan"))
        gc.collect()   # tensors of earlier tests that die during the call would move the counter too
        mem0 = torch.cuda.memory_allocated(dev)
        ops.gemm_argmax_noalloc(h, packed, b, ids, best, gap, ln_a, ln_b, gemm_format=fmt)
        assert torch.cuda.memory_allocated(dev) == mem0
        torch.cuda.synchronize(dev)
        assert torch.equal(ids[:n], rid), "ids differ (n=%d)" % n
        assert torch.equal(best[:n], rbest) and torch.equal(gap[:n], rgap), "best / gap differ bitwise (n=%d)" % n
        assert bool((ids[n:] == -7).all()), "rows beyond n were written"
        # and against an f64 product (the bar of test_gemm_argmax): argmax exact wherever the top-2 gap is clear
        x = h.double()
        y = (x - x.mean(1, keepdim=True)) / (x.std(1, keepdim=True) + 1e-6) * ln_a.double() + ln_b.double()
        logits = y @ W.double().t() + b.double()
        top = logits.topk(2, dim=1)
        clear = (top.values[:, 0] - top.values[:, 1]) > 1e-4
        assert bool((ids[:n].long()[clear] == top.indices[:, 0][clear]).all())
    # beyond the static scratch: a clear error, nothing launched
    rc = L.st_gemm_argmax_noalloc(hs[0].data_ptr(), None, None, packed.data_ptr(), b.data_ptr(), _lib.ST_NOALLOC_MAX_ROWS + 1,
                                  V, fid, ids.data_ptr(), best.data_ptr(), gap.data_ptr(), None)
    assert rc == -1 and b"ST_NOALLOC_MAX_ROWS" in L.st_last_error()
    # n == 0 is a no-op
    assert L.st_gemm_argmax_noalloc(hs[0].data_ptr(), None, None, packed.data_ptr(), b.data_ptr(), 0, V, fid,
                                    ids.data_ptr(), best.data_ptr(), gap.data_ptr(), None) == 0
