#!/usr/bin/env python3
"""Does a GroupNorm launched in row chunks (statistics + apply per chunk) find its second read in the Infinity Cache?
python tools/exp/bench_norm_chunks.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dynamicscaler_amd import ops, _lib
from dynamicscaler_amd.ops import check, _stream
d = torch.device("cuda:0")
lib = _lib.load()


def timeit(fn, n=20):
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


def gn_chunks(x, y, g, b, ninst, rpi, C, nchunk, ws):
    st = _stream()
    per = ninst // nchunk
    for c in range(nchunk):
        off = c * per * rpi * C * 2
        check(lib.ds_groupnorm_f16(x.data_ptr() + off, g.data_ptr(), b.data_ptr(), y.data_ptr() + off, ws.data_ptr(), per, rpi, C, 32,
                                   1e-5, 1, st), "gn")


for E in (16, 8):
    for C, H, W in ((320, 40, 64), (640, 40, 64), (640, 20, 32), (1280, 20, 32), (960, 40, 64)):
        T = 16
        M = E * T * H * W
        x = (torch.randn(M, C, device=d) * 0.5).half()
        y = torch.empty_like(x)
        g, be = torch.ones(C, device=d), torch.zeros(C, device=d)
        ws = torch.empty((lib.ds_groupnorm_stats_workspace_floats(E * T, H * W, 32),), dtype=torch.float32, device=d)
        out = [f"E={E:2d} C={C:5d} {H}x{W} {M*C*2/2**20:6.0f} MiB:"]
        for k in (1, 2, 4, 8, 16, 32):
            t = timeit(lambda: gn_chunks(x, y, g, be, E * T, H * W, C, k, ws))
            out.append(f"k={k}: {t*1e3:6.3f} ms")
        print(" ".join(out), flush=True)
