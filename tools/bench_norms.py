#!/usr/bin/env python3
"""GroupNorm / LayerNorm streaming rates at the UNet's shapes (GB/s of algorithmic traffic: GroupNorm 3 passes, LayerNorm 2).
python tools/bench_norms.py        (DS_HIP_LIBRARY selects a build variant for A/B runs)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicscaler_amd import ops, _lib
d = torch.device("cuda:0")
T = 16


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


print(os.path.basename(_lib.LIB_PATH), "DS_GN_SPARSE_WGS=" + os.environ.get("DS_GN_SPARSE_WGS", "default"))
for E in [int(a) for a in sys.argv[1:]] or [16]:     # evaluations per launch: 16 = one GPU's batch, 2 / 1 = an 8-GPU rank's
  print(f"-- {E} evaluations per launch")
  for C, H, W in ((320, 40, 64), (640, 40, 64), (640, 20, 32), (1280, 20, 32), (1280, 10, 16), (1920, 20, 32), (960, 40, 64)):
      M = E * T * H * W
      x = (torch.randn(M, C, device=d) * 0.5).half()
      g, be = torch.ones(C, device=d), torch.zeros(C, device=d)
      t1 = timeit(lambda: ops.groupnorm(x, g, be, E * T, H * W, C, 1e-5, True))
      t2 = timeit(lambda: ops.groupnorm(x, g, be, E, T * H * W, C, 1e-5, True))
      t3 = timeit(lambda: ops.layernorm(x, g, be))
      t4 = timeit(lambda: ops.layernorm_stats(x))
      print(f"C={C:5d} {H}x{W} M={M:7d}: groupnorm per frame {t1*1e3:7.3f} ms {3.0*M*C*2/t1/1e9:7.0f} GB/s | joint-T {t2*1e3:7.3f} ms "
            f"{3.0*M*C*2/t2/1e9:7.0f} GB/s | layernorm {t3*1e3:7.3f} ms {2.0*M*C*2/t3/1e9:7.0f} GB/s | stats only {t4*1e3:7.3f} ms {1.0*M*C*2/t4/1e9:7.0f} GB/s")
