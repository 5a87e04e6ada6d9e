#!/usr/bin/env python3
"""Spatial self-attention at the UNet's shapes (TFLOP/s).  DS_HIP_LIBRARY selects a build variant for A/B runs."""
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


print(os.path.basename(_lib.LIB_PATH))
for rep in range(2):
    for (B, C, nq) in ((256, 320, 2560), (256, 640, 640), (256, 1280, 160), (128, 320, 2560)):
        heads = C // 64
        qkv = (torch.randn(B * nq, 3 * C, device=d) * 0.5).half()
        o = torch.empty((B * nq, C), dtype=torch.float16, device=d)
        t = timeit(lambda: ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], o, batch=B, heads=heads, nq=nq, nk=nq, ldq=3 * C, ldk=3 * C,
                                         ldv=3 * C, ldo=C, scale=0.125))
        print(f"batch {B} heads {heads} {nq}x{nq}: {t*1e3:8.3f} ms {4.0*B*heads*nq*nq*64/t/1e12:7.1f} TFLOP/s")
