#!/usr/bin/env python3
"""Repeatability of the UNet's kernels under concurrency (GPU box): a mixed list of launches (GEMM variants, spatial and
temporal attention, GroupNorm paths, LayerNorm, concat, im2col, SiLU, timestep embedding, rows->NCTHW) is run serially
for reference, then replayed many times on two HIP streams at once as two hipGraphs; every output must equal its
serial output bit for bit.       python tools/kernel_concurrency_check.py [rounds=30]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicscaler_amd import ops, _lib

d = torch.device("cuda:0")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
gen = torch.Generator().manual_seed(11)


def rnd(*s, scale=0.5, dtype=torch.float16):
    return (torch.randn(*s, generator=gen) * scale).to(d, dtype)


def build(seed):
    """Returns a list of (name, fn) where fn() launches one kernel and returns its output tensor."""
    jobs = []
    B, T = 4, 4                       # toy pipeline: 2 tiles x (cond, uncond), 4 frames
    for (C, H, W) in ((64, 8, 16), (128, 4, 8)):
        M = B * T * H * W
        heads = C // 64
        x = rnd(M, C)
        gam, bet = rnd(C, dtype=torch.float32) + 1, rnd(C, dtype=torch.float32)
        jobs.append((f"gn frame C{C}", lambda x=x, gam=gam, bet=bet, C=C, H=H, W=W: ops.groupnorm(x, gam, bet, B * T, H * W, C, 1e-5, True)))
        jobs.append((f"gn joint C{C}", lambda x=x, gam=gam, bet=bet, C=C, H=H, W=W: ops.groupnorm(x, gam, bet, B, T * H * W, C, 1e-5, True)))
        jobs.append((f"ln C{C}", lambda x=x, gam=gam, bet=bet: ops.layernorm(x, gam, bet)))
        qkv = rnd(M, 3 * C)
        o1 = torch.empty((M, C), dtype=torch.float16, device=d)
        jobs.append((f"attn C{C}", lambda qkv=qkv, o1=o1, C=C, H=H, W=W, heads=heads: ops.attention(
            qkv, qkv[:, C:], qkv[:, 2 * C:], o1, batch=B * T, heads=heads, nq=H * W, nk=H * W, ldq=3 * C, ldk=3 * C, ldv=3 * C,
            ldo=C, scale=0.125)))
        o2 = torch.empty((M, C), dtype=torch.float16, device=d)
        jobs.append((f"tattn C{C}", lambda qkv=qkv, o2=o2, C=C, H=H, W=W, heads=heads: ops.temporal_attention(
            qkv, qkv[:, C:], qkv[:, 2 * C:], o2, nseq_batches=B, T=T, hw=H * W, heads=heads, ldq=3 * C, ldk=3 * C, ldv=3 * C,
            ldo=C, scale=0.125)))
        kv = rnd(B * 77, 2 * C)
        q = rnd(M, C)
        o3 = torch.empty((M, C), dtype=torch.float16, device=d)
        jobs.append((f"xattn C{C}", lambda q=q, kv=kv, o3=o3, C=C, H=H, W=W, heads=heads: ops.attention(
            q, kv, kv[:, C:], o3, batch=B * T, heads=heads, nq=H * W, nk=77, ldq=C, ldk=2 * C, ldv=2 * C, ldo=C,
            kv_batch_div=T, scale=0.125)))
        w3 = rnd(C, 9 * C, scale=0.05)
        b3 = rnd(C, dtype=torch.float32)
        emb = rnd(B, 2 * C, dtype=torch.float32)
        jobs.append((f"conv3 pib C{C}", lambda x=x, w3=w3, emb=emb, C=C, H=H, W=W, M=M: ops.gemm(
            x, w3, emb[:, C:], None, M=M, N=C, K=9 * C, a_mode=_lib.DS_A_CONV3, cin=C, lda=C, conv=(B * T, H, W, H, W, 1, 0),
            bias_rows=T * H * W, ldbias=2 * C)))
        r = rnd(M, C)
        jobs.append((f"conv3 res C{C}", lambda x=x, w3=w3, b3=b3, r=r, C=C, H=H, W=W, M=M: ops.gemm(
            x, w3, b3, r, M=M, N=C, K=9 * C, a_mode=_lib.DS_A_CONV3, cin=C, lda=C, conv=(B * T, H, W, H, W, 1, 0))))
        wt = rnd(C, 3 * C, scale=0.05)
        jobs.append((f"tconv C{C}", lambda x=x, wt=wt, b3=b3, C=C, H=H, W=W, M=M: ops.gemm(
            x, wt, b3, None, M=M, N=C, K=3 * C, a_mode=_lib.DS_A_TCONV, cin=C, lda=C, tconv=(T, H * W))))
        wq = rnd(3 * C, C, scale=0.1)
        jobs.append((f"qkv C{C}", lambda x=x, wq=wq, C=C, M=M: ops.gemm(x, wq, None, None, M=M, N=3 * C, K=C)))
        wg = rnd(8 * C, C, scale=0.1)
        bg = rnd(8 * C, dtype=torch.float32)
        jobs.append((f"geglu C{C}", lambda x=x, wg=wg, bg=bg, C=C, M=M: ops.gemm(x, wg, bg, None, M=M, N=8 * C, K=C,
                                                                                  epilogue=_lib.DS_EPI_GEGLU)))
        h4 = rnd(M, 4 * C)
        w2 = rnd(C, 4 * C, scale=0.05)
        jobs.append((f"ff2 C{C}", lambda h4=h4, w2=w2, b3=b3, r=r, C=C, M=M: ops.gemm(h4, w2, b3, r, M=M, N=C, K=4 * C)))
        jobs.append((f"concat C{C}", lambda x=x, r=r: ops.concat_channels(x, r)))
    xin = rnd(B, 4, T, 8, 16)
    jobs.append(("im2col_in", lambda xin=xin: ops.im2col_in(xin, 64)))
    y = rnd(B * T * 8 * 16, 4, dtype=torch.float32)
    jobs.append(("rows_to_ncthw", lambda y=y: ops.rows_to_ncthw(y, (B, 4, T, 8, 16), torch.float32)))
    t = torch.full((B,), 499, dtype=torch.int64, device=d)
    jobs.append(("timestep_embedding", lambda t=t: ops.timestep_embedding(t, 64)))
    e = rnd(B, 256)
    jobs.append(("silu", lambda e=e: ops.silu(e)))
    we, be = rnd(256, 64, scale=0.1), rnd(256, dtype=torch.float32)
    te = rnd(B, 64)
    jobs.append(("time_embed.0", lambda te=te, we=we, be=be: ops.gemm(te, we, be, None, M=B, N=256, K=64, epilogue=_lib.DS_EPI_SILU)))
    wa, ba = rnd(1024, 256, scale=0.05), rnd(1024, dtype=torch.float32)
    jobs.append(("emb_all f32", lambda e=e, wa=wa, ba=ba: ops.gemm(e, wa, ba, None, M=B, N=1024, K=256, epilogue=_lib.DS_EPI_OUT_F32)))
    return jobs


lists = [build(0), build(1)]
ref = []
for L in lists:
    ref.append([fn().clone() for _, fn in L])
torch.cuda.synchronize()
streams = [torch.cuda.Stream(d), torch.cuda.Stream(d)]
graphs, outs = [], []
for k in range(2):
    gk = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gk):
        o = None
        for _rep in range(3):
            o = [fn() for _, fn in lists[k]]
    graphs.append(gk)
    outs.append(o)
bad = {}
for r in range(rounds):
    for k in range(2):
        with torch.cuda.stream(streams[k]):
            graphs[k].replay()
    torch.cuda.synchronize()
    for k in range(2):
        for i, (name, _) in enumerate(lists[k]):
            if not torch.equal(outs[k][i], ref[k][i]):
                bad[name] = bad.get(name, 0) + 1
print("CONCURRENT == SERIAL for all kernels" if not bad else f"MISMATCHES: {bad}")
sys.exit(1 if bad else 0)
