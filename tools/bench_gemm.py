"""GEMM / attention micro-benchmark on the UNet's shapes (GPU box).  python tools/bench_gemm.py [evals]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicscaler_amd import ops, _lib

d = torch.device("cuda:0")
E = int(sys.argv[1]) if len(sys.argv) > 1 else 16   # UNet evaluations batched (tiles x CFG)
T = 16


def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def rnd(*shape):
    return (torch.randn(*shape, device=d) * 0.5).half()


rows = []
levels = [(320, 40, 64), (640, 20, 32), (1280, 10, 16), (1280, 5, 8)]
for C, H, W in levels:
    M = E * T * H * W
    shapes = [("linear qkv", M, 3 * C, C, {}), ("linear out", M, C, C, {}), ("ff1 geglu", M, 8 * C, C, {"epilogue": _lib.DS_EPI_GEGLU}),
              ("ff2", M, C, 4 * C, {})]
    for name, m, n, k, kw in shapes:
        A, Wt = rnd(m, k), rnd(n, k)
        b = torch.randn(n, device=d)
        t = timeit(lambda: ops.gemm(A, Wt, b, None, M=m, N=n, K=k, **kw))
        # measurement only: the vendor library (hipBLASLt through torch) on the same dense shape, plain product, no epilogue -- what
        # the per-shape gap to the hand kernel is (the product path never calls it)
        bh = b.half()
        tv = timeit(lambda: torch.addmm(bh, A, Wt.t()))
        rows.append((f"L{C}x{H}x{W} {name}", m, n, k, t, 2.0 * m * n * k / t / 1e12, 2.0 * m * n * k / tv / 1e12))
    for cin in (C, 2 * C):
        A, Wt = rnd(M, cin), rnd(C, 9 * cin)
        b = torch.randn(C, device=d)
        t = timeit(lambda: ops.gemm(A, Wt, b, None, M=M, N=C, K=9 * cin, a_mode=_lib.DS_A_CONV3, cin=cin, lda=cin,
                                    conv=(E * T, H, W, H, W, 1, 0)))
        rows.append((f"L{C}x{H}x{W} conv3x3 cin={cin}", M, C, 9 * cin, t, 2.0 * M * C * 9 * cin / t / 1e12))
    A, Wt = rnd(M, C), rnd(C, 3 * C)
    t = timeit(lambda: ops.gemm(A, Wt, None, None, M=M, N=C, K=3 * C, a_mode=_lib.DS_A_TCONV, cin=C, lda=C, tconv=(T, H * W)))
    rows.append((f"L{C}x{H}x{W} tconv", M, C, 3 * C, t, 2.0 * M * C * 3 * C / t / 1e12))
    # spatial self-attention
    heads = C // 64
    qkv = rnd(M, 3 * C)
    o = torch.empty((M, C), dtype=torch.float16, device=d)
    nq = H * W
    t = timeit(lambda: ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], o, batch=E * T, heads=heads, nq=nq, nk=nq, ldq=3 * C,
                                     ldk=3 * C, ldv=3 * C, ldo=C, scale=0.125))
    rows.append((f"L{C}x{H}x{W} self-attn", E * T * heads, nq, nq, t, 4.0 * E * T * heads * nq * nq * 64 / t / 1e12))
    t = timeit(lambda: ops.temporal_attention(qkv, qkv[:, C:], qkv[:, 2 * C:], o, nseq_batches=E, T=T, hw=H * W, heads=heads,
                                              ldq=3 * C, ldk=3 * C, ldv=3 * C, ldo=C, scale=0.125))
    rows.append((f"L{C}x{H}x{W} temporal-attn (GB/s)", M, C, 0, t, 4.0 * M * C * 2 / t / 1e9))
    x = rnd(M, C)
    g, be = torch.ones(C, device=d), torch.zeros(C, device=d)
    t = timeit(lambda: ops.groupnorm(x, g, be, E * T, H * W, C, 1e-5, True))
    rows.append((f"L{C}x{H}x{W} groupnorm+silu (GB/s)", M, C, 0, t, 3.0 * M * C * 2 / t / 1e9))
    t = timeit(lambda: ops.layernorm(x, g, be))
    rows.append((f"L{C}x{H}x{W} layernorm (GB/s)", M, C, 0, t, 2.0 * M * C * 2 / t / 1e9))
for r in rows:
    vend = f"   vendor GEMM (plain addmm) {r[6]:7.1f}" if len(r) > 6 else ""
    print(f"{r[0]:42s} M={r[1]:7d} N={r[2]:5d} K={r[3]:6d}  {r[4]*1e3:9.3f} ms  {r[5]:8.1f}{vend}")
