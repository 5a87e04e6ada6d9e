#!/usr/bin/env python3
"""Kernel-level repeatability of ds_gemm_f16 under concurrency (GPU box): every launch of a mixed list of shapes is run
serially first, then the whole list is replayed many times on two HIP streams at once (captured as two hipGraphs, like
the pipelines do); every concurrent output must equal its serial output bit for bit.
    python tools/gemm_concurrency_check.py [rounds=20]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicscaler_amd import ops, _lib

d = torch.device("cuda:0")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
g = torch.Generator().manual_seed(7)


def rnd(*s, scale=0.5):
    return (torch.randn(*s, generator=g) * scale).to(d)


# toy-UNet-like and a few larger launches: (M, N, K, bias, residual, epilogue)
SHAPES = [(2048, 64, 64, 1, 0, 0), (2048, 192, 64, 0, 0, 0), (2048, 64, 64, 1, 1, 0), (2048, 512, 64, 1, 0, _lib.DS_EPI_GEGLU),
          (2048, 64, 256, 1, 1, 0), (512, 128, 128, 1, 0, 0), (512, 384, 128, 0, 0, 0), (512, 128, 1152, 1, 1, 0),
          (4, 256, 64, 1, 0, _lib.DS_EPI_SILU), (4, 256, 256, 1, 1, 0), (4096, 64, 576, 1, 0, 0), (40960, 320, 320, 1, 1, 0),
          (40960, 960, 320, 0, 0, 0), (8192, 1280, 1280, 1, 0, 0)]


def make_jobs(seed_off):
    jobs = []
    for (M, N, K, hb, hr, epi) in SHAPES:
        A, W = rnd(M, K).half(), rnd(N, K, scale=0.1).half()
        b = rnd(N) if hb else None
        n_out = N // 2 if epi & _lib.DS_EPI_GEGLU else N
        R = rnd(M, n_out).half() if hr else None
        out = torch.empty((M, n_out), dtype=torch.float16, device=d)
        jobs.append((A, W, b, R, out, M, N, K, epi))
    return jobs


def run(jobs):
    for (A, W, b, R, out, M, N, K, epi) in jobs:
        ops.gemm(A, W, b, R, M=M, N=N, K=K, out=out, epilogue=epi)


jobs = [make_jobs(0), make_jobs(1)]
for j in jobs:
    run(j)
torch.cuda.synchronize()
ref = [[o[4].clone() for o in j] for j in jobs]
streams = [torch.cuda.Stream(d), torch.cuda.Stream(d)]
graphs = []
for k in range(2):
    gk = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gk):
        run(jobs[k]); run(jobs[k]); run(jobs[k])
    graphs.append(gk)
bad = 0
for r in range(rounds):
    for j in jobs:
        for o in j:
            o[4].zero_()
    torch.cuda.synchronize()
    for k in range(2):
        with torch.cuda.stream(streams[k]):
            graphs[k].replay()
    torch.cuda.synchronize()
    for k in range(2):
        for i, o in enumerate(jobs[k]):
            if not torch.equal(o[4], ref[k][i]):
                nd = int((o[4] != ref[k][i]).sum())
                print(f"round {r} stream {k} launch {i} {SHAPES[i]}: {nd} elements differ, max |diff| {float((o[4].float() - ref[k][i].float()).abs().max()):.3e}")
                bad += 1
print("CONCURRENT == SERIAL" if bad == 0 else f"{bad} MISMATCHES")
sys.exit(1 if bad else 0)
