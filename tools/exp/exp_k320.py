"""Tile choice on the short-K (HBM-bound) launches: python tools/exp/exp_k320.py [evals]  (DS_GEMM_TILE=n to force a tile).
Checks every result against a torch fp32 matmul of the same fp16 operands."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dynamicscaler_amd import ops, _lib
d = torch.device("cuda:0")
E = int(sys.argv[1]) if len(sys.argv) > 1 else 16
M1 = E * 16 * 2560


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


tot = 0.0
for name, lvl, n, k, res, mode in [("out 320", 1, 320, 320, True, 0), ("q 320", 1, 320, 320, False, 0), ("qkv 960", 1, 960, 320, False, 0),
                                   ("ff2 320x1280", 1, 320, 1280, True, 0), ("tconv 320", 1, 320, 960, True, 2),
                                   ("conv 320", 1, 320, 2880, False, 1),
                                   ("L2 out 640", 4, 640, 640, True, 0), ("L2 qkv 1920", 4, 1920, 640, False, 0),
                                   ("L2 ff2 640x2560", 4, 640, 2560, True, 0), ("L2 tconv 640", 4, 640, 1920, True, 2),
                                   ("L3 out 1280", 16, 1280, 1280, True, 0), ("L3 qkv 3840", 16, 3840, 1280, False, 0)]:
    m = M1 // lvl
    cin = k if mode == 0 else (k // 9 if mode == 1 else k // 3)
    A, W, b = rnd(m, cin), rnd(n, k) * 0.1, torch.randn(n, device=d)
    R = rnd(m, n) if res else None
    H, Wd = {1: (40, 64), 4: (20, 32), 16: (10, 16)}[lvl]
    kw = {}
    if mode == 1:
        kw = dict(a_mode=_lib.DS_A_CONV3, cin=cin, lda=cin, conv=(E * 16, H, Wd, H, Wd, 1, 0))
    elif mode == 2:
        kw = dict(a_mode=_lib.DS_A_TCONV, cin=cin, lda=cin, tconv=(16, H * Wd))
    out = ops.gemm(A, W, b, R, M=m, N=n, K=k, **kw)
    err = None
    if mode == 0:
        rows = slice(0, 4096)
        ref = A[rows].float() @ W.float().t() + b + (R[rows].float() if res else 0)
        err = float((out[rows].float() - ref).abs().max() / ref.abs().max())
        rows = slice(m - 1000, m)
        ref = A[rows].float() @ W.float().t() + b + (R[rows].float() if res else 0)
        err = max(err, float((out[rows].float() - ref).abs().max() / ref.abs().max()))
    t = timeit(lambda: ops.gemm(A, W, b, R, M=m, N=n, K=k, **kw))
    tot += t
    byt = 2.0 * (m * cin + m * n + (m * n if res else 0))
    print(f"tile={os.environ.get('DS_GEMM_TILE','auto'):4s} {name:16s} M={m:7d} {t*1e3:7.3f} ms  {2.0*m*n*k/t/1e12:7.1f} TF  {byt/t/1e12:5.2f} TB/s  err {err}")
print(f"tile={os.environ.get('DS_GEMM_TILE','auto'):4s} total {tot*1e3:.3f} ms")
