"""Round-6 experiment: can a memory-bound kernel with few registers run INSIDE a persistent MFMA-bound GEMM (one workgroup per CU, 2 x 232
VGPRs per SIMD: 48 registers free)?  Stream A: the level-3 3x3 convolution (40960 x 1280 x 11520) forced onto the 256 x 256 tile; stream B: a
plain device copy (torch's vectorised elementwise kernel: ~20 VGPRs, no LDS) of `mb` megabytes.  Serial (one stream) against concurrent
(two streams), 20 rounds each.   DS_HIP_LIBRARY=.../libdynscaler_hip_tune.so DS_GEMM_TILE=2 python tools/exp/coresidency.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dynamicscaler_amd import ops, _lib
d = torch.device("cuda:0")
E, T, H, W, C = 16, 16, 10, 16, 1280
M = E * T * H * W
A = (torch.randn(M, C, device=d) * 0.5).half()
Wt = (torch.randn(C, 9 * C, device=d) * 0.05).half()
b = torch.randn(C, device=d)
kw = dict(a_mode=_lib.DS_A_CONV3, cin=C, lda=C, conv=(E * T, H, W, H, W, 1, 0))
gemm = lambda: ops.gemm(A, Wt, b, None, M=M, N=C, K=9 * C, **kw)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for mb in (256, 1024):
    x = torch.empty(mb << 19, dtype=torch.float16, device=d).normal_()
    y = torch.empty_like(x)
    copy = lambda: y.copy_(x)
    ta, tb = timed(gemm), timed(copy)

    def serial():
        gemm(); copy()

    def conc():
        ev = torch.cuda.Event(); ev.record()
        with torch.cuda.stream(sA):
            sA.wait_event(ev); gemm(); ea = torch.cuda.Event(); ea.record()
        with torch.cuda.stream(sB):
            sB.wait_event(ev); copy(); eb = torch.cuda.Event(); eb.record()
        torch.cuda.current_stream().wait_event(ea); torch.cuda.current_stream().wait_event(eb)

    ts, tc = timed(serial), timed(conc)
    print(f"copy {2 * mb} MB of traffic: gemm alone {ta:.3f} ms, copy alone {tb:.3f} ms ({2 * mb / 1e3 / tb:.2f} TB/s), serial {ts:.3f} ms, two streams {tc:.3f} ms "
          f"(max {max(ta, tb):.3f}, sum {ta + tb:.3f})", flush=True)
