"""Short-K GEMM family (K = 320 / 640 / 1280 linears of the transformer path) under a forced tile (DS_GEMM_TILE, read once per
process): does another tile than choose_tile's pick run a shape faster?   python tools/exp/bench_tile_choice.py [evals]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dynamicscaler_amd import ops, _lib

d = torch.device("cuda:0")
E = int(sys.argv[1]) if len(sys.argv) > 1 else 16


def timeit(fn, iters=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def rnd(*shape):
    return (torch.randn(*shape, device=d) * 0.5).half()


print("tile", os.environ.get("DS_GEMM_TILE", "default"), "evals", E)
for C, H, W in [(320, 40, 64), (640, 20, 32), (1280, 10, 16)]:
    M = E * 16 * H * W
    x = rnd(M, C)
    for name, n, k, res, epi in (("out+res", C, C, True, 0), ("qkv", 3 * C, C, False, 0), ("geglu", 8 * C, C, False, _lib.DS_EPI_GEGLU),
                                 ("ff2+res", C, 4 * C, True, 0)):
        if epi and os.environ.get("DS_GEMM_TILE") in ("0", "3", "5"):
            pass
        A, Wt, b = rnd(M, k), rnd(n, k), torch.randn(n, device=d)
        try:
            t = timeit(lambda: ops.gemm(A, Wt, b, x if res else None, M=M, N=n, K=k, epilogue=epi))
            print(f"  L{C} {name:8s} M={M} N={n} K={k}: {t * 1e6:8.1f} us  {2.0 * M * n * k / t / 1e12:7.1f} TFLOP/s", flush=True)
        except Exception as e:
            print(f"  L{C} {name}: {type(e).__name__}")
        del A, Wt
