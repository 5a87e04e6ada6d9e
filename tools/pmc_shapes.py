#!/usr/bin/env python3
"""Launch the top shapes of a cfg3 DDIM step a few times each (for `rocprofv3 --pmc ...` passes, program directly after `--`):
the five GEMM shapes that take the most time in profiles/r1_step_shape_breakdown_v2.csv plus the big self-attention.
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY ... -d OUT -- python3 tools/pmc_shapes.py
The position in the dispatch sequence identifies a shape in the counter CSV (tools/pmc_sq_summary.py reads JOB_LABELS / REPS below)."""
import os, sys
REPS = 3
# the GEMM jobs in launch order (REPS launches each), then the self-attention: with persistent workgroups every big launch has the
# same grid, so tools/pmc_sq_summary.py tells the shapes apart by their position in the dispatch sequence
JOB_LABELS = ["geglu 655360x2560x320", "out 655360x320x320 +bias +residual", "qkv 655360x960x320", "geglu 163840x5120x640",
              "conv3 655360x320x2880", "tconv 655360x320x960", "conv3 40960x1280x11520"]
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicscaler_amd import ops, _lib

d = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)


def rnd(*s, scale=0.5, dtype=torch.float16):
    return (torch.randn(*s, generator=g) * scale).to(d, dtype)


M1, M2, M3 = 655360, 163840, 40960   # rows at UNet levels 1-3 for 16 evaluations of a [16,40,64] tile
# (label, kwargs of ops.gemm)
x320, x640, x1280 = rnd(M1, 320), rnd(M2, 640), rnd(M3, 1280)
jobs = [
    ("geglu 655360x2560x320", dict(A=x320, W=rnd(2560, 320), bias=rnd(2560, dtype=torch.float32), M=M1, N=2560, K=320, epilogue=_lib.DS_EPI_GEGLU)),
    ("out 655360x320x320 +bias +residual", dict(A=x320, W=rnd(320, 320), bias=rnd(320, dtype=torch.float32), residual=rnd(M1, 320), M=M1, N=320, K=320)),
    ("qkv 655360x960x320", dict(A=x320, W=rnd(960, 320), M=M1, N=960, K=320)),
    ("geglu 163840x5120x640", dict(A=x640, W=rnd(5120, 640), bias=rnd(5120, dtype=torch.float32), M=M2, N=5120, K=640, epilogue=_lib.DS_EPI_GEGLU)),
    ("conv3 655360x320x2880", dict(A=x320, W=rnd(320, 2880), bias=rnd(320, dtype=torch.float32), M=M1, N=320, K=2880, a_mode=_lib.DS_A_CONV3, cin=320, lda=320, conv=(256, 40, 64, 40, 64, 1, 0))),
    ("tconv 655360x320x960", dict(A=x320, W=rnd(320, 960), bias=rnd(320, dtype=torch.float32), M=M1, N=320, K=960, a_mode=_lib.DS_A_TCONV, cin=320, lda=320, tconv=(16, 2560))),
    ("conv3 40960x1280x11520", dict(A=x1280, W=rnd(1280, 11520), bias=rnd(1280, dtype=torch.float32), M=M3, N=1280, K=11520, a_mode=_lib.DS_A_CONV3, cin=1280, lda=1280, conv=(256, 10, 16, 10, 16, 1, 0))),
]
assert [j[0] for j in jobs] == JOB_LABELS
for label, kw in jobs:
    kw = dict(kw)
    A, W = kw.pop("A"), kw.pop("W")
    bias, res = kw.pop("bias", None), kw.pop("residual", None)
    for _ in range(REPS):
        ops.gemm(A, W, bias, res, **kw)
    torch.cuda.synchronize()
qkv = rnd(M1, 960)
o = torch.empty((M1, 320), dtype=torch.float16, device=d)
for _ in range(REPS):
    ops.attention(qkv, qkv[:, 320:], qkv[:, 640:], o, batch=256, heads=5, nq=2560, nk=2560, ldq=960, ldk=960, ldv=960, ldo=320, scale=0.125)
torch.cuda.synchronize()
print("done")
