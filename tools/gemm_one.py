"""Run one GEMM shape a few times (for rocprofv3 --pmc).  python tools/gemm_one.py M N K [conv cin H W]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicscaler_amd import ops, _lib
d = torch.device("cuda:0")
M, N, K = map(int, sys.argv[1:4])
A = (torch.randn(M, K if len(sys.argv) < 5 else int(sys.argv[4]), device=d) * 0.5).half()
W = (torch.randn(N, K, device=d) * 0.5).half()
b = torch.randn(N, device=d)
for _ in range(4):
    if len(sys.argv) >= 7:
        cin, H, Wd = int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
        ops.gemm(A, W, b, None, M=M, N=N, K=K, a_mode=_lib.DS_A_CONV3, cin=cin, lda=cin, conv=(M // (H * Wd), H, Wd, H, Wd, 1, 0))
    else:
        ops.gemm(A, W, b, None, M=M, N=N, K=K)
torch.cuda.synchronize()
