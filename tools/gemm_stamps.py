"""Phase breakdown of gemm_f16_kernel from in-kernel s_memtime stamps (diagnostic build, GPU box).

    python tools/gemm_stamps.py M N K [epilogue] [conv cin H W]

Builds tools/exp/libgemm_stamps.so beforehand (see tools/build_stamps.sh).  Prints, over all workgroups, the
median cycles of: prologue (first global load -> LDS), main loop, accumulators -> LDS, epilogue (bias / activation /
store), and the number of workgroup 'rounds' per CU slot.
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicscaler_amd import _lib

lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "exp", os.environ.get("DS_STAMP_LIB", "libgemm_stamps.so")))
lib.ds_gemm_f16.argtypes = [C.c_void_p] * 5 + [C.POINTER(_lib.GemmDesc), C.c_void_p]
lib.ds_dbg_set_stamps.argtypes = [C.c_void_p]

dev = torch.device("cuda:0")
M, N, K = map(int, sys.argv[1:4])
epi = int(sys.argv[4]) if len(sys.argv) > 4 else 0
conv = len(sys.argv) >= 8
cin = int(sys.argv[5]) if conv else K
A = (torch.randn(M, cin, device=dev) * 0.5).half()
W = (torch.randn(N, K, device=dev) * 0.5).half()
b = torch.randn(N, device=dev)
n_out = N // 2 if epi & _lib.DS_EPI_GEGLU else N
out = torch.empty(M, n_out, dtype=torch.float16, device=dev)
d = _lib.GemmDesc()
d.M, d.N, d.K, d.a_mode, d.cin, d.lda = M, N, K, (_lib.DS_A_CONV3 if conv else _lib.DS_A_DENSE), cin, cin
if conv:
    H, Wd = int(sys.argv[6]), int(sys.argv[7])
    d.nimg, d.hin, d.win, d.hout, d.wout, d.stride, d.upsample = M // (H * Wd), H, Wd, H, Wd, 1, 0
d.ldc, d.ldr, d.bias_rows, d.ldbias, d.epilogue = n_out, 0, M, N, epi
nblk = ((M + 127) // 128) * ((N + 63) // 64)  # upper bound over all tile shapes
stamps = torch.zeros(nblk * 8, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream


def run():
    rc = lib.ds_gemm_f16(A.data_ptr(), W.data_ptr(), b.data_ptr(), None, out.data_ptr(), C.byref(d), st)
    assert rc == 0, rc


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    run()
e1.record()
torch.cuda.synchronize()
t_plain = e0.elapsed_time(e1) / 5
assert lib.ds_dbg_set_stamps(stamps.data_ptr()) == 0
e0.record()
run()
e1.record()
torch.cuda.synchronize()
t_st = e0.elapsed_time(e1)
s = stamps.cpu().numpy().reshape(-1, 8)
s = s[s[:, 4] > 0]
seg = np.diff(s[:, :5], axis=1).astype(np.float64)
names = ["prologue", "mainloop", "acc->lds", "epilogue"]
tot = s[:, 4] - s[:, 0]
span = s[:, 4].max() - s[:, 0].min()
# in-kernel shader clock: d(s_memtime) / d(s_memrealtime) x 100 MHz, median over workgroups (MI355X_MICROARCH.md)
rt = (s[:, 6] - s[:, 5]).astype(np.float64)
clk = np.median((s[:, 4] - s[:, 0])[rt > 0] / rt[rt > 0]) * 100.0
print(f"shape M={M} N={N} K={K} epi={epi} conv={conv}: {t_plain:.3f} ms/launch unstamped ({2.0*M*N*K/t_plain/1e9:.0f} TFLOP/s), "
      f"{t_st:.3f} ms stamped; {len(s)} workgroups; in-kernel shader clock {clk:.0f} MHz")
for i, n in enumerate(names):
    print(f"  {n:10s} median {np.median(seg[:, i]):9.0f}  mean {seg[:, i].mean():9.0f}  p90 {np.percentile(seg[:, i], 90):9.0f} cycles")
print(f"  {'total':10s} median {np.median(tot):9.0f}  mean {tot.mean():9.0f}; sum of workgroup cycles / span = {tot.sum() / span:.1f} concurrent workgroups "
      f"({tot.sum() / span / 256:.2f} per CU)")
