"""Tile choice on the short-K (HBM-bound) linears of level 1: python tools/exp_k320.py  (DS_GEMM_TILE=n to force)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicscaler_amd import ops, _lib
d = torch.device("cuda:0")
M = 655360


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def rnd(*s):
    return (torch.randn(*s, device=d) * 0.5).half()


for name, n, k, res, epi in [("out 320", 320, 320, True, 0), ("q 320", 320, 320, False, 0), ("qkv 960", 960, 320, False, 0),
                             ("geglu 2560", 2560, 320, False, _lib.DS_EPI_GEGLU), ("ff2 320x1280", 320, 1280, True, 0),
                             ("L2 out 640", 640, 640, True, 0)]:
    m = M if "L2" not in name else M // 4
    A, W, b = rnd(m, k), rnd(n, k), torch.randn(n, device=d)
    R = rnd(m, n) if res else None
    t = timeit(lambda: ops.gemm(A, W, b, R, M=m, N=n, K=k, epilogue=epi))
    nout = n // 2 if epi else n
    byt = 2.0 * (m * k + m * nout + (m * n if res else 0))
    print(f"tile={os.environ.get('DS_GEMM_TILE','auto'):4s} {name:14s} {t*1e3:7.3f} ms  {2.0*m*n*k/t/1e12:7.1f} TF  {byt/t/1e12:5.2f} TB/s")
