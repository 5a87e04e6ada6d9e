#!/usr/bin/env python3
"""Wide launches (many N tiles, W larger than an XCD's L2) at the UNet's shapes: ms and TFLOP/s.  DS_GEMM_GROUP_M selects the
grouped tile walk (0 = N-fastest walk).   python tools/bench_wide_gemm.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicscaler_amd import ops, _lib
d = torch.device("cuda:0")


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


print("DS_GEMM_GROUP_M =", os.environ.get("DS_GEMM_GROUP_M", "(default)"))
for (M, N, K, ge) in ((655360, 2560, 320, True), (163840, 5120, 640, True), (40960, 10240, 1280, True), (40960, 3840, 1280, False),
                      (163840, 1920, 640, False), (40960, 1280, 5120, False), (327680, 2560, 320, True), (81920, 5120, 640, True),
                      (20480, 10240, 1280, True)):
    A = (torch.randn(M, K, device=d) * 0.5).half()
    W = (torch.randn(N, K, device=d) * K ** -0.5).half()
    b = torch.randn(N, device=d) * 0.1
    out = torch.empty((M, N // 2 if ge else N), dtype=torch.float16, device=d)
    t = timeit(lambda: ops.gemm(A, W, b, None, M=M, N=N, K=K, epilogue=_lib.DS_EPI_GEGLU if ge else 0, out=out))
    print(f"{M:7d} x {N:5d} x {K:5d} {'geglu' if ge else '     '}: {t*1e3:7.3f} ms {2.0*M*N*K/t/1e12:7.1f} TFLOP/s", flush=True)
    del A, W, out
