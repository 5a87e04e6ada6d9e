"""-m gpu: parity of every HIP kernel (through the C ABI) against the CPU oracle / a plain torch-CPU fp32
restatement of the same op, on seeded inputs.  Bit-exact for the data-movement and fp32 tile ops; within a stated
fp16 tolerance for the MFMA paths."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import ring as oring
from oracle import ddim as oddim


def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    return torch.device("cuda:0")


def rnd(shape, seed, scale=1.0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype)


def T(a):
    return torch.from_numpy(np.asarray(a))


def relerr(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-12))


# ------------------------------------------------------------------------------------------------ ring tile ops
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("geom", [
    # (pano F,H,W), (tile f,h,w), origins [(f0,y0,x0)]
    ((6, 8, 16), (4, 4, 8), [(0, 0, 0), (4, 6, 8), (5, 7, 16)]),             # vector path (x0 % 8 == 0), wraps F,H,W
    ((6, 8, 16), (3, 5, 7), [(1, 2, 3), (5, 6, 13), (9, 11, 25)]),          # scalar path, max origins (< 2*size)
    ((16, 64, 512), (16, 40, 64), [(0, 3, 8 + 64 * k) for k in range(8)]),   # config-3 step-1 row 0
    ((16, 64, 512), (16, 40, 64), [(0, 27, 456)]),                           # crosses the W seam and the H seam
])
def test_ring_gather_scatter_bit_exact(dtype, geom):
    from dynamicscaler_amd import ops
    d = dev()
    (Fp, Hp, Wp), (tf, th, tw), origins = geom
    pano = rnd((1, 4, Fp, Hp, Wp), 1).to(dtype)
    maskp = (rnd((Fp, Hp, Wp), 2) > 0).to(torch.uint8)
    tiles, mt = ops.ring_gather(pano.to(d), origins, (tf, th, tw), maskp.to(d))
    for i, (f0, y0, x0) in enumerate(origins):
        ref = oring.ring_gather(pano, x0, x0 + tw, y0, y0 + th, f0, f0 + tf)
        assert torch.equal(tiles[i:i + 1].cpu(), ref)
        refm = oring.ring_gather(maskp[None, None], x0, x0 + tw, y0, y0 + th, f0, f0 + tf)[0, 0]
        assert torch.equal(mt[i].cpu(), refm)
    # scatter: disjoint windows only (the pipelines guarantee that per launch)
    from dynamicscaler_amd.parallel import windows_overlap
    wins = [(x0, x0 + tw, y0, y0 + th, f0, f0 + tf) for (f0, y0, x0) in origins]
    keep = []
    for j, w in enumerate(wins):
        if all(not windows_overlap(wins[k], w, (Fp, Hp, Wp)) for k in keep):
            keep.append(j)
    xp = rnd((len(keep), 4, tf, th, tw), 3).to(dtype)
    x0t = rnd((len(keep), 4, tf, th, tw), 4).to(dtype)
    p1, p2, pm = pano.clone().to(d), torch.zeros_like(pano).to(d), torch.zeros((Fp, Hp, Wp), dtype=torch.uint8, device=d)
    ops.ring_scatter3(p1, p2, pm, xp.to(d), x0t.to(d), [origins[j] for j in keep])
    r1, r2, rm = pano.clone(), torch.zeros_like(pano), torch.zeros((1, 1, Fp, Hp, Wp))
    for n, j in enumerate(keep):
        f0, y0, x0 = origins[j]
        oring.ring_scatter(r1, xp[n:n + 1], x0, x0 + tw, y0, y0 + th, f0, f0 + tf)
        oring.ring_scatter(r2, x0t[n:n + 1], x0, x0 + tw, y0, y0 + th, f0, f0 + tf)
        oring.ring_scatter(rm, torch.ones(1, 1, tf, th, tw), x0, x0 + tw, y0, y0 + th, f0, f0 + tf)
    assert torch.equal(p1.cpu(), r1) and torch.equal(p2.cpu(), r2)
    assert torch.equal(pm.cpu().float(), rm[0, 0])


def test_ring_errors_match_reference_asserts():
    from dynamicscaler_amd import ops, _lib
    from dynamicscaler_amd.ring import RingLatent
    d = dev()
    pano = torch.zeros(1, 4, 6, 8, 16, device=d)
    with pytest.raises(_lib.DsError, match="Invalid pos_left"):
        ops.ring_gather(pano, [(0, 0, 10)], (6, 8, 32))          # right edge > 2*W
    with pytest.raises(_lib.DsError, match="warp should not occur"):
        ops.ring_scatter3(pano, None, None, torch.zeros(1, 4, 6, 8, 17, device=d), None, [(0, 0, 0)])
    r = RingLatent(pano)
    with pytest.raises(AssertionError):
        r.set_window_latent(torch.zeros(1, 4, 6, 8, 3, device=d), 0, 4, 0, 8, 0, 6)   # shape mismatch (:190)
    with pytest.raises(AssertionError):
        r.get_window_latent(10, 42, 0, 8, 0, 6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("ratio", [1, 0.3])
def test_renoise_mix_bit_exact(dtype, ratio):
    from dynamicscaler_amd import ops
    d = dev()
    sched = oddim.DDIMSchedule(oddim.DiffusionTables(), 50)
    n, shape = 3, (3, 4, 4, 8, 16)
    x = rnd(shape, 11).to(dtype)
    nz = rnd(shape, 12).to(dtype)
    m = (rnd((n, 4, 8, 16), 13) > 0).to(torch.uint8)
    c, s = sched.renoise_coefficients(20, 21)
    out = ops.renoise_mix_(x.clone().to(d), m.to(d), (1, 4, 4, 8, 16), c, s, ratio, noise=nz.to(d), mask_frame0=True)
    for i in range(n):
        xi = x[i:i + 1].float()
        noised = oddim.re_noise(sched, xi, 20, 21, noise=nz[i:i + 1].float())
        ref = oddim.mix_latents_with_mask(xi, noised, m[i, 0][None].float(), ratio).to(dtype)
        assert torch.equal(out[i:i + 1].cpu(), ref)
    # 5-D (per-frame) mask form
    out5 = ops.renoise_mix_(x.clone().to(d), m.to(d), (1, 4, 4, 8, 16), c, s, ratio, noise=nz.to(d), mask_frame0=False)
    for i in range(n):
        xi = x[i:i + 1].float()
        noised = oddim.re_noise(sched, xi, 20, 21, noise=nz[i:i + 1].float())
        m5 = m[i][None, None].float().expand(1, 4, -1, -1, -1)
        assert torch.equal(out5[i:i + 1].cpu(), oddim.mix_latents_with_mask(xi, noised, m5, ratio).to(dtype))


def test_renoise_philox_statistics():
    from dynamicscaler_amd import ops
    d = dev()
    x = torch.zeros((8, 4, 16, 40, 64), device=d)
    m = torch.ones((8, 16, 40, 64), dtype=torch.uint8, device=d)
    out = ops.renoise_mix_(x, m, (1, 4, 16, 64, 512), 0.0, 1.0, 1.0, noise=None, mask_frame0=False, seed=7, offset=0)
    z = out.float().cpu()
    assert abs(float(z.mean())) < 5e-3 and abs(float(z.std()) - 1.0) < 5e-3
    assert abs(float((z ** 4).mean()) - 3.0) < 0.05      # kurtosis of a normal
    out2 = ops.renoise_mix_(torch.zeros_like(x), m, (1, 4, 16, 64, 512), 0.0, 1.0, 1.0, noise=None, mask_frame0=False,
                            seed=7, offset=0)
    assert torch.equal(out, out2)                          # counter-based: reproducible
    # tile_ids: a tile's noise depends on its number within the step only, not on the batch / rank it is processed in
    args = ((1, 4, 16, 64, 512), 0.0, 1.0, 1.0)
    whole = ops.renoise_mix_(torch.zeros_like(x), m, *args, seed=7, offset=5, tile_ids=list(range(8)), mask_frame0=False)
    ids = [1, 3, 4, 7]
    part = ops.renoise_mix_(torch.zeros_like(x[:4]), m[:4], *args, seed=7, offset=5, tile_ids=ids, mask_frame0=False)
    assert torch.equal(part, whole[ids])
    flat = whole.float().cpu().reshape(8, -1)
    assert abs(float(flat.std()) - 1.0) < 5e-3
    corr = torch.corrcoef(flat)                              # distinct tiles: independent streams
    assert float((corr - torch.eye(8)).abs().max()) < 0.02


def test_renoise_philox_matches_cpu_restatement():
    """rng_mode="device" (bench.py): the normals ds_renoise_mix draws in-kernel are the ones oracle/philox.py computes on the CPU
    (Philox4x32-10 known-answer-tested there, same counter layout, same Box-Muller) -- value by value, to the accuracy of the
    hardware's fast log / sin / cos -- and the re-noise + mix built on them equals the oracle's re_noise + mix fed with that noise."""
    from dynamicscaler_amd import ops
    from oracle import philox
    d = dev()
    shape, seed, offset = (3, 4, 4, 8, 16), 0x1234567890ABCDEF, 977
    m = torch.ones(shape[:1] + shape[2:], dtype=torch.uint8, device=d)
    z_gpu = ops.renoise_mix_(torch.zeros(shape, device=d), m, (1, 4, 4, 12, 64), 0.0, 1.0, 1.0, noise=None, mask_frame0=False, seed=seed,
                             offset=offset).cpu()
    z_cpu = T(philox.tile_noise(shape, seed, offset))
    assert float((z_gpu - z_cpu).abs().max()) < 2e-5, float((z_gpu - z_cpu).abs().max())
    # the whole fused op in device mode against the oracle with the restated noise (mask with holes, ratio 0.3)
    x = rnd(shape, 3)
    mk = (rnd(shape[:1] + shape[2:], 4) > 0).to(torch.uint8)
    c, s_ = 0.83, 0.41
    got = ops.renoise_mix_(x.clone().to(d), mk.to(d), (1, 4, 4, 12, 64), c, s_, 0.3, noise=None, mask_frame0=False, seed=seed, offset=offset).cpu()
    noised = c * x + s_ * z_cpu
    want = torch.stack([oddim.mix_latents_with_mask(x[i:i + 1], noised[i:i + 1], mk[i][None, None].float(), 0.3)[0] for i in range(shape[0])])
    assert float((got - want).abs().max()) < 2e-5
    # tile_ids: tile k of the step draws counters offset + k * numel ... (what keeps rank-sharded runs identical)
    ids = [2, 0]
    part = ops.renoise_mix_(torch.zeros((2,) + shape[1:], device=d), m[:2], (1, 4, 4, 12, 64), 0.0, 1.0, 1.0, noise=None, mask_frame0=False,
                            seed=seed, offset=offset, tile_ids=ids).cpu()
    numel = int(np.prod(shape[1:]))
    for k, j in enumerate(ids):
        zk = T(philox.tile_noise((1,) + shape[1:], seed, offset + j * numel))
        assert float((part[k:k + 1] - zk).abs().max()) < 2e-5


@pytest.mark.parametrize("dtype,edt", [(torch.float32, torch.float32), (torch.float16, torch.float32),
                                       (torch.float16, torch.float16)])
@pytest.mark.parametrize("eta", [0.0, 1.0])
def test_cfg_ddim_bit_exact(dtype, edt, eta):
    from dynamicscaler_amd import ops
    d = dev()
    sched = oddim.DDIMSchedule(oddim.DiffusionTables(), 50, eta=eta)
    shape = (2, 4, 4, 8, 16)
    x, ec, eu = rnd(shape, 21).to(dtype), rnd(shape, 22).to(edt), rnd(shape, 23).to(edt)
    z = rnd(shape, 24).to(dtype)
    for index in (0, 25, 49):
        coef = sched.step_coefficients(index)
        xp, x0 = ops.cfg_ddim(x.to(d), ec.to(d), eu.to(d), (1, 4, 4, 8, 16), 7.5, coef, z.to(d) if eta else None)
        e = oddim.cfg_combine(ec.float(), eu.float(), 7.5)
        rxp, rx0 = oddim.ddim_step(sched, x.float(), e, [index] * 4, noise=z.float())
        assert torch.equal(xp.cpu(), rxp.to(dtype)) and torch.equal(x0.cpu(), rx0.to(dtype))


# ------------------------------------------------------------------------------------------------ GEMM family
def _h(t):
    return t.half().float()


@pytest.mark.parametrize("M,N,K", [(300, 320, 320), (257, 256, 128), (128, 1280, 1024), (640, 64, 64), (5, 192, 64)])
def test_gemm_dense_bias_residual(M, N, K):
    from dynamicscaler_amd import ops
    d = dev()
    A, W = _h(rnd((M, K), 1)), _h(rnd((N, K), 2, 0.1))
    b = rnd((N,), 3)
    R = _h(rnd((M, N), 4))
    out = ops.gemm(A.half().to(d), W.half().to(d), b.to(d), R.half().to(d), M=M, N=N, K=K)
    ref = A @ W.t() + b + R
    assert relerr(out, ref) < 1e-3
    assert float((out.float().cpu() - ref).abs().max()) < 2e-2 * float(ref.abs().max())
    # no bias / no residual, and an exact small-integer A=I style check with an asymmetric W
    Ai = torch.zeros((M, K)); Ai[torch.arange(min(M, K)), torch.arange(min(M, K))] = 1.0
    Wi = (torch.arange(N * K).reshape(N, K) % 17 - 8).float()
    out = ops.gemm(Ai.half().to(d), Wi.half().to(d), None, None, M=M, N=N, K=K)
    assert torch.equal(out.float().cpu(), Ai @ Wi.t())


def test_gemm_epilogues():
    from dynamicscaler_amd import ops, _lib
    from dynamicscaler_amd.unet import _interleave_geglu
    d = dev()
    M, K = 200, 128
    A = _h(rnd((M, K), 1))
    # per-item bias (time-embedding add): 4 items of 50 rows, bias table wider than N (ldbias)
    N = 192
    W = _h(rnd((N, K), 2, 0.1))
    table = rnd((4, 400), 3)
    out = ops.gemm(A.half().to(d), W.half().to(d), table.to(d)[:, 100:], None, M=M, N=N, K=K, bias_rows=50, ldbias=400)
    ref = A @ W.t() + table[:, 100:100 + N].repeat_interleave(50, 0)
    assert relerr(out, ref) < 1e-3
    # SiLU
    b = rnd((N,), 4)
    out = ops.gemm(A.half().to(d), W.half().to(d), b.to(d), None, M=M, N=N, K=K, epilogue=_lib.DS_EPI_SILU)
    assert relerr(out, F.silu(A @ W.t() + b)) < 1e-3
    # fp32 output with N=4 (conv-out shape) and N not a multiple of 8
    W4 = _h(rnd((4, K), 5, 0.1))
    out = ops.gemm(A.half().to(d), W4.half().to(d), b[:4].contiguous().to(d), None, M=M, N=4, K=K, epilogue=_lib.DS_EPI_OUT_F32)
    assert out.dtype == torch.float32 and relerr(out, A @ W4.t() + b[:4]) < 1e-5
    # GEGLU (attention.py:376-383): proj -> chunk(2) -> x * gelu(gate)
    inner = 256
    Wg, bg = _h(rnd((2 * inner, K), 6, 0.1)), rnd((2 * inner,), 7)
    out = ops.gemm(A.half().to(d), _interleave_geglu(Wg).half().to(d), _interleave_geglu(bg).to(d), None, M=M,
                   N=2 * inner, K=K, epilogue=_lib.DS_EPI_GEGLU)
    xg = A @ Wg.t() + bg
    ref = xg[:, :inner] * F.gelu(xg[:, inner:])
    assert out.shape == (M, inner) and relerr(out, ref) < 1e-3


@pytest.mark.parametrize("M,N,K,what", [
    (161 * 256 - 219, 320, 320, "256x320"),        # >= 160 workgroups of 256x320, ragged last row tile
    (160 * 256 + 5, 256, 192, "256x256"),
    (81 * 256 - 1, 640, 128, "256x320 two N tiles"),
    (1024, 1024, 1024, "128x64 deep (4-stage)"),   # <= 128 blocks of 128x128 and K/64 >= 8
    (2048 - 77, 2048, 512, "128x128 deep (4-stage)"),
])
def test_gemm_big_and_deep_tiles(M, N, K, what):
    """The one-workgroup-per-CU tile variants (LDS-DMA, source-side swizzle, hardware zero-fill of tail rows), which the
    small shapes above never select: bias + residual, per-item bias, SiLU, fp32 output, and an exact integer check."""
    from dynamicscaler_amd import ops, _lib
    d = dev()
    A, W = _h(rnd((M, K), 1)), _h(rnd((N, K), 2, 0.1))
    b, R = rnd((N,), 3), _h(rnd((M, N), 4))
    Ad, Wd = A.half().to(d), W.half().to(d)
    base = A @ W.t()
    out = ops.gemm(Ad, Wd, b.to(d), R.half().to(d), M=M, N=N, K=K)
    assert relerr(out, base + b + R) < 1e-3, what
    items = 7
    rows = -(-M // items)
    table = rnd((items, N + 64), 5)
    out = ops.gemm(Ad, Wd, table.to(d)[:, 64:], None, M=M, N=N, K=K, bias_rows=rows, ldbias=N + 64)
    ref = base + table[:, 64:].repeat_interleave(rows, 0)[:M]
    assert relerr(out, ref) < 1e-3, what
    out = ops.gemm(Ad, Wd, b.to(d), None, M=M, N=N, K=K, epilogue=_lib.DS_EPI_SILU)
    assert relerr(out, F.silu(base + b)) < 1e-3, what
    out = ops.gemm(Ad, Wd, b.to(d), None, M=M, N=N, K=K, epilogue=_lib.DS_EPI_OUT_F32)
    assert out.dtype == torch.float32 and relerr(out, base + b) < 1e-5, what
    Ai = torch.zeros((M, K)); Ai[torch.arange(M), torch.arange(M) % K] = 1.0          # row m selects column m % K of W
    Wi = (torch.arange(N * K).reshape(N, K) % 17 - 8).float()
    out = ops.gemm(Ai.half().to(d), Wi.half().to(d), None, None, M=M, N=N, K=K)
    assert torch.equal(out.float().cpu(), Ai @ Wi.t()), what


def test_gemm_big_tiles_geglu_and_conv():
    """GEGLU on the 256x256 tile (32-row [x | gate] interleave across a wave's tile pairs) with a ragged M, and the 3x3
    implicit GEMM (stride 1 and 2, padding taps zero-filled by the buffer loads) at a size that selects 256x320."""
    from dynamicscaler_amd import ops, _lib
    from dynamicscaler_amd.unet import _interleave_geglu
    d = dev()
    M, K, inner = 160 * 256 - 100, 128, 256
    A = _h(rnd((M, K), 1))
    Wg, bg = _h(rnd((2 * inner, K), 6, 0.1)), rnd((2 * inner,), 7)
    out = ops.gemm(A.half().to(d), _interleave_geglu(Wg).half().to(d), _interleave_geglu(bg).to(d), None, M=M,
                   N=2 * inner, K=K, epilogue=_lib.DS_EPI_GEGLU)
    xg = A @ Wg.t() + bg
    assert out.shape == (M, inner) and relerr(out, xg[:, :inner] * F.gelu(xg[:, inner:])) < 1e-3
    for stride, nimg, hin, win in ((1, 17, 40, 64), (2, 66, 40, 64)):
        cin, cout = 64, 320
        x = _h(rnd((nimg, cin, hin, win), 1))
        w = _h(rnd((cout, cin, 3, 3), 2, 0.05))
        b = rnd((cout,), 3)
        ref = F.conv2d(x, w, b, stride=stride, padding=1)
        hout, wout = ref.shape[-2:]
        assert nimg * hout * wout >= 160 * 256
        a = x.permute(0, 2, 3, 1).reshape(-1, cin).half().to(d)
        wp = w.permute(0, 2, 3, 1).reshape(cout, -1).half().to(d)
        out = ops.gemm(a, wp, b.to(d), None, M=nimg * hout * wout, N=cout, K=9 * cin, a_mode=_lib.DS_A_CONV3, cin=cin,
                       lda=cin, conv=(nimg, hin, win, hout, wout, stride, 0))
        got = out.float().cpu().reshape(nimg, hout, wout, cout).permute(0, 3, 1, 2)
        assert relerr(got, ref) < 1e-3, stride


@pytest.mark.parametrize("stride,upsample,hin,win", [(1, 0, 10, 12), (2, 0, 10, 12), (2, 0, 5, 8), (1, 1, 5, 6)])
def test_gemm_conv3x3(stride, upsample, hin, win):
    from dynamicscaler_amd import ops, _lib
    d = dev()
    nimg, cin, cout = 3, 64, 128
    x = _h(rnd((nimg, cin, hin, win), 1))
    w = _h(rnd((cout, cin, 3, 3), 2, 0.05))
    b = rnd((cout,), 3)
    xin = F.interpolate(x, scale_factor=2, mode="nearest") if upsample else x
    ref = F.conv2d(xin, w, b, stride=stride, padding=1)
    hout, wout = ref.shape[-2:]
    a = x.permute(0, 2, 3, 1).reshape(-1, cin).half().to(d)
    wp = w.permute(0, 2, 3, 1).reshape(cout, -1).half().to(d)
    out = ops.gemm(a, wp, b.to(d), None, M=nimg * hout * wout, N=cout, K=9 * cin, a_mode=_lib.DS_A_CONV3, cin=cin, lda=cin,
                   conv=(nimg, hin, win, hout, wout, stride, upsample))
    got = out.float().cpu().reshape(nimg, hout, wout, cout).permute(0, 3, 1, 2)
    assert relerr(got, ref) < 1e-3


def test_gemm_temporal_conv():
    from dynamicscaler_amd import ops, _lib
    d = dev()
    B, T, H, W, Cc = 2, 4, 5, 6, 64
    x = _h(rnd((B, Cc, T, H, W), 1))
    w = _h(rnd((Cc, Cc, 3, 1, 1), 2, 0.05))
    b = rnd((Cc,), 3)
    ref = F.conv3d(x, w, b, padding=(1, 0, 0))
    a = x.permute(0, 2, 3, 4, 1).reshape(-1, Cc).half().to(d)
    wp = w[:, :, :, 0, 0].permute(0, 2, 1).reshape(Cc, -1).half().to(d)
    out = ops.gemm(a, wp, b.to(d), None, M=B * T * H * W, N=Cc, K=3 * Cc, a_mode=_lib.DS_A_TCONV, cin=Cc, lda=Cc,
                   tconv=(T, H * W))
    got = out.float().cpu().reshape(B, T, H, W, Cc).permute(0, 4, 1, 2, 3)
    assert relerr(got, ref) < 1e-3


def test_gemm_rejects_bad_shapes():
    from dynamicscaler_amd import ops, _lib
    d = dev()
    a = torch.zeros((8, 48), dtype=torch.float16, device=d)
    with pytest.raises(_lib.DsError, match="multiple of 64"):
        ops.gemm(a, a, None, None, M=8, N=8, K=48)


# ------------------------------------------------------------------------------------------------ attention
def _attn_ref(q, k, v, scale):
    s = torch.einsum("bhid,bhjd->bhij", q, k) * scale
    return torch.einsum("bhij,bhjd->bhid", s.softmax(-1), v)


@pytest.mark.parametrize("nq,nk,batch,heads,kvdiv", [(160, 160, 3, 2, 1), (700, 700, 2, 1, 1), (40, 40, 4, 3, 1),
                                                      (200, 77, 4, 2, 2), (64, 16, 6, 1, 3), (2560, 2560, 1, 1, 1)])
def test_attention_matches_softmax_reference(nq, nk, batch, heads, kvdiv):
    from dynamicscaler_amd import ops
    d = dev()
    C = heads * 64
    kvb = batch // kvdiv
    q = _h(rnd((batch, nq, C), 1))
    kv = _h(rnd((kvb, nk, 2 * C), 2))
    # spike one key against one query so the running max jumps mid-sequence (online-softmax rescale branch)
    kv[0, nk // 2, :64] = 6.0 * q[0, 0, :64]
    kk, vv = kv[..., :C], kv[..., C:]
    qh = q.reshape(batch, nq, heads, 64).permute(0, 2, 1, 3)
    kh = kk.reshape(kvb, nk, heads, 64).permute(0, 2, 1, 3).repeat_interleave(kvdiv, 0)
    vh = vv.reshape(kvb, nk, heads, 64).permute(0, 2, 1, 3).repeat_interleave(kvdiv, 0)
    ref = _attn_ref(qh, kh, vh, 0.125).permute(0, 2, 1, 3).reshape(batch, nq, C)
    qd, kvd = q.half().to(d).reshape(batch * nq, C), kv.half().to(d).reshape(kvb * nk, 2 * C)
    out = torch.empty((batch * nq, C), dtype=torch.float16, device=d)
    ops.attention(qd, kvd, kvd[:, C:], out, batch=batch, heads=heads, nq=nq, nk=nk, ldq=C, ldk=2 * C, ldv=2 * C, ldo=C,
                  kv_batch_div=kvdiv, scale=0.125)
    got = out.float().cpu().reshape(batch, nq, C)
    assert relerr(got, ref) < 2e-3
    assert float((got - ref).abs().max()) < 1e-2 * max(1.0, float(ref.abs().max()))
    # accumulate form (image-token branch): out += attention
    ops.attention(qd, kvd, kvd[:, C:], out, batch=batch, heads=heads, nq=nq, nk=nk, ldq=C, ldk=2 * C, ldv=2 * C, ldo=C,
                  kv_batch_div=kvdiv, scale=0.125, accumulate=True)
    assert relerr(out.float().cpu().reshape(batch, nq, C), 2 * ref) < 3e-3


@pytest.mark.parametrize("case", ["ramp", "creep", "negative", "late_spike"])
def test_attention_deferred_reference_edge_cases(case):
    """The flash loop moves the exponent's reference only when a tile's maximum exceeds it by more than 8 (log2 units)
    (csrc/attention.hip DEFER_LOG2).  Score profiles that stress that rule: a maximum that grows with the key index (a move in
    every tile), one that creeps by less than the threshold per tile (p stays above 1 for many tiles), scores far below zero
    (the first tile must set the reference, whatever its sign), and a single late key that dominates a query."""
    from dynamicscaler_amd import ops
    d = dev()
    nq, nk, batch, heads = 192, 704, 2, 2
    C = heads * 64
    g = torch.Generator().manual_seed(11)
    q = torch.randn((batch, nq, C), generator=g)
    k = torch.randn((batch, nk, C), generator=g)
    v = torch.randn((batch, nk, C), generator=g)
    u = torch.nn.functional.normalize(torch.randn(64, generator=g), dim=0)
    ramp = torch.arange(nk, dtype=torch.float32) / nk
    if case == "ramp":          # score ~ 8 * 40 * key/nk * 0.125 = up to 40 (58 in log2 units) along the keys
        q[..., :64] = 8.0 * u
        k[..., :64] = 0.3 * k[..., :64] + (40.0 * ramp)[None, :, None] * u
    elif case == "creep":       # +0.4 per 64-key tile in log2 units: below the threshold for ~20 tiles
        q[..., :64] = 4.0 * u
        k[..., :64] = 0.1 * k[..., :64] + (6.0 * ramp)[None, :, None] * u
    elif case == "negative":    # every score of head 0 near -60
        q[..., :64] = 8.0 * u
        k[..., :64] = 0.3 * k[..., :64] - 60.0 * u
    else:                       # one key in the last tile, 12 above the rest
        q[..., :64] = 8.0 * u
        k[..., :64] = 0.3 * k[..., :64]
        k[:, nk - 3, :64] += 12.0 * u
    q, k, v = _h(q), _h(k), _h(v)
    qh = q.reshape(batch, nq, heads, 64).permute(0, 2, 1, 3)
    kh = k.reshape(batch, nk, heads, 64).permute(0, 2, 1, 3)
    vh = v.reshape(batch, nk, heads, 64).permute(0, 2, 1, 3)
    ref = _attn_ref(qh.double(), kh.double(), vh.double(), 0.125).permute(0, 2, 1, 3).reshape(batch, nq, C).float()
    qkv = torch.cat([q, k[:, :nq] * 0, v[:, :nq] * 0], -1)        # q rows in a wider buffer (row stride 3C), k / v separate
    qd = qkv.half().to(d).reshape(batch * nq, 3 * C)
    kd, vd = k.half().to(d).reshape(batch * nk, C), v.half().to(d).reshape(batch * nk, C)
    out = torch.empty((batch * nq, C), dtype=torch.float16, device=d)
    ops.attention(qd, kd, vd, out, batch=batch, heads=heads, nq=nq, nk=nk, ldq=3 * C, ldk=C, ldv=C, ldo=C, scale=0.125)
    got = out.float().cpu().reshape(batch, nq, C)
    assert torch.isfinite(got).all()
    assert relerr(got, ref) < 2e-3
    assert float((got - ref).abs().max()) < 1e-2 * max(1.0, float(ref.abs().max()))
    # a query's result does not depend on which other queries share its wave (the move is decided per wave): 32 queries alone, bit for bit
    one = torch.empty((batch * 32, C), dtype=torch.float16, device=d)
    q32 = qd.reshape(batch, nq, 3 * C)[:, 80:112].contiguous().reshape(batch * 32, 3 * C)   # half of one wave's queries, half of the next one's
    ops.attention(q32, kd, vd, one, batch=batch, heads=heads, nq=32, nk=nk, ldq=3 * C, ldk=C, ldv=C, ldo=C, scale=0.125)
    assert torch.equal(one.reshape(batch, 32, C), out.reshape(batch, nq, C)[:, 80:112])


@pytest.mark.parametrize("T", [16, 4, 24, 17, 32, 1])
def test_temporal_attention(T):
    from dynamicscaler_amd import ops
    d = dev()
    B, hw, heads = 2, 37, 2
    C = heads * 64
    qkv = _h(rnd((B * T * hw, 3 * C), 5))
    x = qkv.reshape(B, T, hw, 3, heads, 64).permute(3, 0, 2, 4, 1, 5)   # [3, B, hw, heads, T, 64]
    ref = _attn_ref(x[0].reshape(-1, heads, T, 64), x[1].reshape(-1, heads, T, 64), x[2].reshape(-1, heads, T, 64), 0.125)
    ref = ref.reshape(B, hw, heads, T, 64).permute(0, 3, 1, 2, 4).reshape(B * T * hw, C)
    qd = qkv.half().to(d)
    out = torch.empty((B * T * hw, C), dtype=torch.float16, device=d)
    ops.temporal_attention(qd, qd[:, C:], qd[:, 2 * C:], out, nseq_batches=B, T=T, hw=hw, heads=heads, ldq=3 * C,
                           ldk=3 * C, ldv=3 * C, ldo=C, scale=0.125)
    assert relerr(out, ref) < 2e-3


# ------------------------------------------------------------------------------------------------ norms / misc
# (from (8, 2560, 320) on: shapes of the opt-in one-launch register-resident form of round 6 -- 52 / 26 / 13 rows per thread, instance
#  counts that are and are not multiples of 8 (the XCD-aware slab placement), the joint-T shape of level 3)
@pytest.mark.parametrize("ninst,rows,C", [(6, 80, 320), (2, 4 * 80, 64), (3, 40, 2560), (2, 700, 1920), (5, 33, 128),
                                          (8, 2560, 320), (3, 2560, 320), (16, 640, 640), (5, 640, 1280), (4, 2560, 1280), (16, 300, 640), (2, 2551, 640)])
@pytest.mark.parametrize("silu", [False, True])
def test_groupnorm(ninst, rows, C, silu):
    from dynamicscaler_amd import ops
    d = dev()
    x = _h(rnd((ninst * rows, C), 1) * 2 + 0.5)
    g, b = 1 + 0.1 * rnd((C,), 2), 0.1 * rnd((C,), 3)
    y = ops.groupnorm(x.half().to(d), g.to(d), b.to(d), ninst, rows, C, 1e-5, silu)
    xr = x.reshape(ninst, rows, C).permute(0, 2, 1)
    ref = F.group_norm(xr, 32, g, b, 1e-5)
    if silu:
        ref = F.silu(ref)
    ref = ref.permute(0, 2, 1).reshape(ninst * rows, C)
    assert relerr(y, ref) < 1e-3
    # the two-step entry points (stats, apply) give the same normalisation
    mean, rstd = ops.groupnorm_stats(x.half().to(d), ninst, rows, C, 1e-5)
    y2 = ops.groupnorm_apply(x.half().to(d), mean, rstd, g.to(d), b.to(d), ninst, rows, C, silu)
    assert relerr(y2, ref) < 1e-3 and relerr(y2, y.float().cpu()) < 2e-3
    mref = x.reshape(ninst, rows, 32, C // 32).permute(0, 2, 1, 3).reshape(ninst * 32, -1).mean(1)
    assert torch.allclose(mean.cpu(), mref, atol=1e-4)
    # input as a column slice of a wider row-major buffer (ds_groupnorm_f16_strided: the UNet's skip tensors): same bits
    wide = torch.full((ninst * rows, C + 64), 7.0, dtype=torch.float16, device=d)
    wide[:, 24:24 + C] = x.half().to(d)
    y3 = ops.groupnorm(wide[:, 24:24 + C], g.to(d), b.to(d), ninst, rows, C, 1e-5, silu)
    assert y3.is_contiguous() and torch.equal(y3, y)
    # the opt-in one-launch read-once form (ds_groupnorm_rows_onepass) where it exists: same normalisation to fp32 rounding of the statistics
    assert ops.groupnorm_onepass_applies(rows, C, torch.float16) == ((rows, C) in {(320, 64), (700, 1920), (2560, 320), (640, 640), (640, 1280), (2560, 1280), (300, 640), (2551, 640)})
    if ops.groupnorm_onepass_applies(rows, C, torch.float16):
        y4 = ops.groupnorm(x.half().to(d), g.to(d), b.to(d), ninst, rows, C, 1e-5, silu, onepass=True)
        assert relerr(y4, ref) < 1e-3 and relerr(y4, y.float().cpu()) < 3e-4
        assert torch.equal(ops.groupnorm(wide[:, 24:24 + C], g.to(d), b.to(d), ninst, rows, C, 1e-5, silu, onepass=True), y4)
        part = ops.groupnorm(x.half().to(d)[:rows].contiguous(), g.to(d), b.to(d), 1, rows, C, 1e-5, silu, onepass=True)
        assert torch.equal(part, y4[:rows])                      # batch-invariant: an instance does not see how many share the launch
    else:
        with pytest.raises(Exception):
            ops.groupnorm(x.half().to(d), g.to(d), b.to(d), ninst, rows, C, 1e-5, silu, onepass=True)


@pytest.mark.parametrize("rows,C", [(40, 1280), (160, 1280), (640, 640), (2560, 320), (16 * 160, 1280)])
@pytest.mark.parametrize("xdt", [torch.float16, torch.float32])
def test_groupnorm_is_batch_invariant(rows, C, xdt):
    """An instance's result must not depend on how many instances share the launch (2 evaluations x 16 frames on an 8-GPU rank,
    16 x 16 on one GPU): the kernel form is chosen from the instance's shape only.  Per-frame norms of every UNet level and the
    joint-T form, fp16 and fp32 input."""
    from dynamicscaler_amd import ops
    d = dev()
    n_big = 256 if rows * C <= 2560 * 320 else 128      # > 1024 workgroups: the dense-grid variant; the small batches below take the sparse one
    x = (rnd((n_big * rows, C), 1) * 2 + 0.5).to(xdt).to(d)
    g, b = (1 + 0.1 * rnd((C,), 2)).to(d), (0.1 * rnd((C,), 3)).to(d)
    whole = ops.groupnorm(x, g, b, n_big, rows, C, 1e-5, True)
    for n_small in (1, 2, 32):
        if n_small > n_big:
            continue
        part = ops.groupnorm(x[:n_small * rows].contiguous(), g, b, n_small, rows, C, 1e-5, True)
        assert torch.equal(part, whole[:n_small * rows]), (rows, C, n_small)


@pytest.mark.parametrize("rows,C", [(1000, 320), (77, 1280), (5, 64), (333, 512)])
def test_layernorm(rows, C):
    from dynamicscaler_amd import ops
    d = dev()
    x = _h(rnd((rows, C), 1) * 3 + 1)
    g, b = 1 + 0.1 * rnd((C,), 2), 0.1 * rnd((C,), 3)
    y = ops.layernorm(x.half().to(d), g.to(d), b.to(d))
    assert relerr(y, F.layer_norm(x, (C,), g, b, 1e-5)) < 1e-3


# ------------------------------------------------------------------------------------------------ strict-precision (fp32 residual stream) forms
@pytest.mark.parametrize("M,N,K", [(300, 320, 320), (257, 256, 128), (161 * 256 - 219, 320, 320), (160 * 256 + 3, 256, 64), (5, 192, 64)])
@pytest.mark.parametrize("out_f32", [True, False])
def test_gemm_fp32_residual(M, N, K, out_f32):
    """DS_EPI_RES_F32: the residual rows are fp32 and are added without an fp16 rounding (ResBlock `skip + h`,
    openaimodel3d.py:237-254; `+ x` of attention.py:216-220); with DS_EPI_OUT_F32 the sum is stored in fp32 too.  The residual
    carries a component below fp16 resolution that has to survive."""
    from dynamicscaler_amd import ops, _lib
    d = dev()
    A, W = _h(rnd((M, K), 1)), _h(rnd((N, K), 2, 0.1))
    b = rnd((N,), 3)
    R = rnd((M, N), 4) * 8 + 1e-4 * rnd((M, N), 5)                  # not fp16-representable
    wide = torch.full((M, N + 12), 3.0, dtype=torch.float32, device=d)    # residual as a column slice of a wider fp32 buffer
    wide[:, 8:8 + N] = R.to(d)
    out = ops.gemm(A.half().to(d), W.half().to(d), b.to(d), wide[:, 8:8 + N], M=M, N=N, K=K,
                   epilogue=_lib.DS_EPI_OUT_F32 if out_f32 else 0)
    ref = (A.double() @ W.double().t() + b.double() + R.double()).float()
    if out_f32:
        assert out.dtype == torch.float32
        assert relerr(out, ref) < 2e-6, relerr(out, ref)
        assert float((out.cpu() - ref).abs().max()) < 2e-5 * float(ref.abs().max())
    else:
        assert out.dtype == torch.float16
        assert torch.equal(out.cpu(), ref.half()) or relerr(out, ref.half().float()) < 2e-5   # one rounding of the fp32 sum
    # into a column slice of a wider fp32 output buffer (the decoder's concat buffers), no bias
    if out_f32:
        obuf = torch.zeros((M, N + 24), dtype=torch.float32, device=d)
        ops.gemm(A.half().to(d), W.half().to(d), None, wide[:, 8:8 + N], M=M, N=N, K=K, out=obuf[:, 16:16 + N])
        ref2 = (A.double() @ W.double().t() + R.double()).float()
        assert relerr(obuf[:, 16:16 + N], ref2) < 2e-6 and float(obuf[:, :16].abs().max()) == 0 and float(obuf[:, 16 + N:].abs().max()) == 0


def test_gemm_fp32_residual_rejects_unsupported():
    from dynamicscaler_amd import ops, _lib
    d = dev()
    A, W = torch.zeros((64, 64), dtype=torch.float16, device=d), torch.zeros((128, 64), dtype=torch.float16, device=d)
    R = torch.zeros((64, 64), dtype=torch.float32, device=d)
    with pytest.raises(_lib.DsError):      # GEGLU + fp32 residual
        ops.gemm(A, W, None, R, M=64, N=128, K=64, epilogue=_lib.DS_EPI_GEGLU)
    with pytest.raises(_lib.DsError):      # per-item bias + fp32 residual
        ops.gemm(A, W[:64].contiguous(), torch.zeros((2, 64), device=d), R, M=64, N=64, K=64, bias_rows=32, ldbias=64)


@pytest.mark.parametrize("ninst,rows,C", [(6, 80, 320), (2, 4 * 80, 64), (3, 40, 2560), (2, 700, 1920), (70, 33, 128), (64, 160, 1280),
                                          (8, 2560, 320), (3, 2560, 320), (16, 640, 640), (5, 640, 1280), (4, 2560, 1280), (16, 300, 640), (2, 2551, 640),
                                          (3, 640, 2560), (2, 640, 1920)])
@pytest.mark.parametrize("silu", [False, True])
def test_groupnorm_fp32_input(ninst, rows, C, silu):
    """ds_groupnorm_rows with DS_F32 rows (the fp32 residual stream): statistics and normalisation from the unrounded values,
    fp16 operand out; optional fp16 copy of the raw rows in the same pass; input as a column slice of a wider buffer."""
    from dynamicscaler_amd import ops
    d = dev()
    x = rnd((ninst * rows, C), 1) * 2 + 0.5 + 1e-4 * rnd((ninst * rows, C), 9)
    g, b = 1 + 0.1 * rnd((C,), 2), 0.1 * rnd((C,), 3)
    y, x16 = ops.groupnorm(x.to(d), g.to(d), b.to(d), ninst, rows, C, 1e-5, silu, raw_f16=True)
    xr = x.double().reshape(ninst, rows, C).permute(0, 2, 1)
    ref = F.group_norm(xr, 32, g.double(), b.double(), 1e-5)
    if silu:
        ref = F.silu(ref)
    ref = ref.permute(0, 2, 1).reshape(ninst * rows, C).float()
    assert y.dtype == torch.float16 and relerr(y, ref) < 4e-4          # one fp16 rounding of the output: 2^-11 / sqrt(3) ~ 2.8e-4
    assert float((y.float().cpu() - ref.half().float()).abs().max()) <= 2 * float(torch.finfo(torch.float16).eps) * float(ref.abs().max())
    assert torch.equal(x16.cpu(), x.half())
    y1 = ops.groupnorm(x.to(d), g.to(d), b.to(d), ninst, rows, C, 1e-5, silu)
    assert torch.equal(y1, y)
    wide = torch.full((ninst * rows, C + 64), 7.0, dtype=torch.float32, device=d)
    wide[:, 24:24 + C] = x.to(d)
    y3 = ops.groupnorm(wide[:, 24:24 + C], g.to(d), b.to(d), ninst, rows, C, 1e-5, silu)
    assert y3.is_contiguous() and torch.equal(y3, y)
    # fp16 input through the same entry point stays what it was
    yh = ops.groupnorm(x.half().to(d), g.to(d), b.to(d), ninst, rows, C, 1e-5, silu)
    assert relerr(yh, ref) < 1.5e-3
    # the opt-in one-launch read-once form on fp32 rows: statistics in fp64 over fp32 partial sums, the raw-copy output, strided input
    if ops.groupnorm_onepass_applies(rows, C, torch.float32):
        y5, x16b = ops.groupnorm(x.to(d), g.to(d), b.to(d), ninst, rows, C, 1e-5, silu, raw_f16=True, onepass=True)
        assert relerr(y5, ref) < 4e-4 and torch.equal(x16b.cpu(), x.half())
        assert float((y5.float().cpu() - ref.half().float()).abs().max()) <= 2 * float(torch.finfo(torch.float16).eps) * float(ref.abs().max())
        assert torch.equal(ops.groupnorm(wide[:, 24:24 + C], g.to(d), b.to(d), ninst, rows, C, 1e-5, silu, onepass=True), y5)
        part = ops.groupnorm(x.to(d)[-rows:].contiguous(), g.to(d), b.to(d), 1, rows, C, 1e-5, silu, onepass=True)
        assert torch.equal(part, y5[-rows:])


@pytest.mark.parametrize("rows,C", [(1000, 320), (77, 1280), (5, 64), (333, 512), (130, 2048)])
def test_layernorm_fp32_input_and_cast(rows, C):
    from dynamicscaler_amd import ops
    d = dev()
    x = rnd((rows, C), 1) * 3 + 1 + 1e-4 * rnd((rows, C), 7)
    g, b = 1 + 0.1 * rnd((C,), 2), 0.1 * rnd((C,), 3)
    y = ops.layernorm(x.to(d), g.to(d), b.to(d))
    ref = F.layer_norm(x.double(), (C,), g.double(), b.double(), 1e-5).float()
    assert y.dtype == torch.float16 and relerr(y, ref) < 4e-4
    # ds_cast_rows_f32_f16: dense and strided input
    assert torch.equal(ops.cast_rows_f16(x.to(d)).cpu(), x.half())
    wide = torch.zeros((rows, C + 16), dtype=torch.float32, device=d)
    wide[:, 8:8 + C] = x.to(d)
    assert torch.equal(ops.cast_rows_f16(wide[:, 8:8 + C]).cpu(), x.half())


@pytest.mark.parametrize("M,N,K,geglu", [(1000, 960, 320, False), (300, 320, 320, False), (77, 3840, 1280, False),
                                        (2100, 2560, 320, True), (130, 1024, 128, True), (5, 192, 64, False)])
def test_gemm_layernorm_folded(M, N, K, geglu):
    """ds_layernorm_stats + ds_gemm_f16_ln == LayerNorm -> Linear (-> GEGLU) in fp32 (attention.py:199-220, 376-403), with a
    row mean that is NOT small against the row's spread (the fold subtracts mean * colsum from the raw product)."""
    from dynamicscaler_amd import ops
    from dynamicscaler_amd._lib import DS_EPI_GEGLU
    from dynamicscaler_amd.unet import _interleave_geglu
    d = dev()
    x = _h(rnd((M, K), 1) * 2 + 1.5)
    g, be = 1 + 0.2 * rnd((K,), 2), 0.3 * rnd((K,), 3)
    w, b = _h(rnd((N, K), 4) * K ** -0.5), 0.1 * rnd((N,), 5)
    wg = (w * g[None, :]).half()
    cs, cb = wg.float().sum(1), w @ be + b
    ref = F.layer_norm(x, (K,), g, be, 1e-5) @ w.t() + b
    if geglu:
        a, gate = ref.chunk(2, dim=-1)
        ref = a * F.gelu(gate)
        wg, cs, cb = _interleave_geglu(wg), _interleave_geglu(cs), _interleave_geglu(cb)
    xd = x.half().to(d)
    st = ops.layernorm_stats(xd)
    mean, var = x.mean(1), x.var(1, unbiased=False)
    assert relerr(st[:, 0], mean) < 1e-5 and relerr(st[:, 1], (var + 1e-5).rsqrt()) < 1e-5
    out = ops.gemm_ln(xd, wg.to(d).contiguous(), st, cs.to(d).contiguous(), cb.to(d).contiguous(), M=M, N=N, K=K,
                      epilogue=DS_EPI_GEGLU if geglu else 0)
    e = relerr(out, ref)
    # the rows' statistics taken inside the GEMM from the operand fragments (ds_gemm_f16_lnk; one-pass variance)
    outk = ops.gemm_ln(xd, wg.to(d).contiguous(), None, cs.to(d).contiguous(), cb.to(d).contiguous(), M=M, N=N, K=K,
                       epilogue=DS_EPI_GEGLU if geglu else 0, eps=1e-5)
    ek = relerr(outk, ref)
    # against the two-kernel path (LayerNorm output rounded to fp16, then the plain GEMM): the fold must not be worse
    n16 = ops.layernorm(xd, g.to(d), be.to(d))
    w_plain = (_interleave_geglu(w) if geglu else w).half().to(d).contiguous()
    b_plain = (_interleave_geglu(b) if geglu else b).to(d).contiguous()
    two = ops.gemm(n16, w_plain, b_plain, None, M=M, N=N, K=K, epilogue=DS_EPI_GEGLU if geglu else 0)
    e2 = relerr(two, ref)
    print(f"gemm_ln {M}x{N}x{K} geglu={geglu}: folded {e:.3e}, in-kernel statistics {ek:.3e}, LayerNorm kernel + GEMM {e2:.3e}")
    assert out.shape == ref.shape and e < 1.2e-3 and e < 1.5 * e2 + 1e-4
    assert outk.shape == ref.shape and ek < 1.2e-3 and ek < 1.5 * e2 + 1e-4


def test_misc_ops():
    from dynamicscaler_amd import ops
    d = dev()
    a, b = _h(rnd((100, 64), 1)), _h(rnd((100, 128), 2))
    assert torch.equal(ops.concat_channels(a.half().to(d), b.half().to(d)).cpu(), torch.cat([a, b], 1).half())
    x = _h(rnd((2, 4, 3, 6, 8), 3))
    for dt in (torch.float32, torch.float16):
        p = ops.im2col_in(x.to(dt).to(d), 64).float().cpu()
        ref = F.unfold(x.permute(0, 2, 1, 3, 4).reshape(6, 4, 6, 8), 3, padding=1)          # [6, C*9, L] (c-major)
        ref = ref.reshape(6, 4, 9, 48).permute(0, 3, 2, 1).reshape(6 * 48, 36)                # -> (tap, c)
        assert torch.equal(p[:, :36], ref) and float(p[:, 36:].abs().max()) == 0.0
    y = rnd((2 * 3 * 6 * 8, 4), 4)
    out = ops.rows_to_ncthw(y.to(d), (2, 4, 3, 6, 8), torch.float32).cpu()
    assert torch.equal(out, y.reshape(2, 3, 6, 8, 4).permute(0, 4, 1, 2, 3))
    t = torch.tensor([0, 20, 499, 999], dtype=torch.int64)
    from oracle.unet import timestep_embedding
    e = ops.timestep_embedding(t.to(d), 320).float().cpu()
    assert float((e - timestep_embedding(t, 320)).abs().max()) < 2e-3
    s = _h(rnd((1000,), 5) * 4)
    assert relerr(ops.silu(s.half().to(d)), F.silu(s)) < 1e-3


def test_gemm_dense_operand_over_2gib_is_chunked():
    """A dense A operand of >= 2 GiB (the buffer-offset range the kernel addresses) is split into row chunks by the C
    entry point; the result equals the un-split math."""
    from dynamicscaler_amd import ops
    d = dev()
    M, K, N = 540000, 2048, 64          # A = 2.2 GB fp16
    torch.manual_seed(0)
    A = (torch.randn(M, K, device=d) * 0.25).half()
    W = (torch.randn(N, K, device=d) * 0.05).half()
    b = torch.randn(N, device=d)
    out = ops.gemm(A, W, b, None, M=M, N=N, K=K)
    for r0 in (0, 262144 - 64, 524288 - 64, M - 128):   # around the chunk seams (rows_max = 524160) and the tail
        ref = A[r0:r0 + 128].float().cpu() @ W.float().cpu().t() + b.cpu()
        assert relerr(out[r0:r0 + 128], ref) < 1e-3, r0


def test_bilinear_splat_and_resize_vs_reference_golden():
    """S5 (set_view_tensor_bilinear: 4-tap splat + normaliser, CSR-per-target kernel, no atomics) bit-exact vs the oracle;
    N1 resize_video_latent: nearest bit-exact, bicubic within fp32 round-off of ATen's kernel."""
    import os
    from oracle import sphere as S
    from dynamicscaler_amd.sphere import PanoramaLatentProxy
    from dynamicscaler_amd.tensor_utils import resize_video_latent
    from dynamicscaler_amd.synth import synth_normal
    d = dev()
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "sphere.npz"))
    pano = torch.from_numpy(z["rt_pano"])
    n = 0
    while f"splat_args_{n}" in z:
        fov, th, ph = z[f"splat_args_{n}"].tolist()
        tile = synth_normal((1, 4, 3, 8, 16), 300 + n)
        proxy = PanoramaLatentProxy(pano.to(d))
        proxy.set_view_tensor_bilinear(tile.to(d), fov, th, ph)
        ref = S.sphere_splat_bilinear(pano.clone(), tile, fov, th, ph)
        assert torch.equal(proxy.get_equirect_tensor().cpu(), ref), n
        assert relerr(proxy.get_equirect_tensor(), torch.from_numpy(z[f"splat_after_{n}"])) < 1e-5
        n += 1
    assert n == 4
    lat = torch.from_numpy(z["resize_in"])
    for key, (h, w, mode) in {"nearest_x2": (32, 64, "nearest"), "nearest_half": (8, 16, "nearest"),
                              "nearest_odd": (20, 48, "nearest")}.items():
        assert torch.equal(resize_video_latent(lat.to(d), h, w, mode=mode).cpu(), torch.from_numpy(z[f"resize_{key}"])), key
        assert torch.equal(resize_video_latent(lat.half().to(d), h, w, mode=mode).cpu(), torch.from_numpy(z[f"resize_{key}"]).half())
    for key, (h, w) in {"bicubic_x2": (32, 64), "bicubic_odd": (24, 40)}.items():
        got = resize_video_latent(lat.to(d), h, w, mode="bicubic").cpu()
        assert float((got - torch.from_numpy(z[f"resize_{key}"])).abs().max()) < 2e-6, key


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_residual_merge_bit_exact(dtype):
    """ds_residual_merge against the reference's tensor expressions (t2v_normal_pipeline.py:456-467), both parities,
    sparse and dense; fp16 = the same fp32 arithmetic on fp16 inputs, rounded once."""
    from dynamicscaler_amd import ops, _lib
    d = dev()
    curr, noised = rnd((1, 4, 3, 8, 12), 1).to(dtype), rnd((1, 4, 3, 8, 12), 2).to(dtype)
    for i, r in ((0, 0.9), (1, 0.35), (4, 0.5), (7, 1.0)):
        c32, n32 = curr.float(), noised.float()
        mixed = c32.clone()
        mixed[..., i % 2::2, ::2] = r * c32[..., (i + 1) % 2::2, ::2] + (1.0 - r) * n32[..., ::2, ::2]
        mixed[..., (i + 1) % 2::2, 1::2] = r * c32[..., i % 2::2, 1::2] + (1.0 - r) * n32[..., ::2, ::2]
        got = ops.residual_merge(curr.to(d), noised.to(d), r, i, sparse=True)
        assert torch.equal(got.cpu(), mixed.to(dtype)), (i, r)
        dense = c32 * r + n32 * (1.0 - r)
        got = ops.residual_merge(curr.to(d), noised.to(d), r, i, sparse=False)
        assert torch.equal(got.cpu(), dense.to(dtype)), (i, r)
    with pytest.raises(_lib.DsError, match="even H and W"):
        ops.residual_merge(curr[..., :7, :].contiguous().to(d), noised[..., :7, :].contiguous().to(d), 0.5, 0)


def test_rccl_level_exchange_single_rank():
    """The per-level tile all-gather (parallel.exchange_level) through RCCL on device tensors, in a one-rank group: the
    call pattern the N > 1 bench uses (world_size-2 logic is covered on CPU with gloo, tests/test_parallel_gloo.py)."""
    import torch.distributed as dist
    from dynamicscaler_amd import parallel
    d = dev()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=d)
    try:
        xp = rnd((8, 4, 4, 8, 16), 1).half().to(d)
        x0 = rnd((8, 4, 4, 8, 16), 2).half().to(d)
        a, b = parallel.exchange_level(xp, x0, 8, force=True)
        torch.cuda.synchronize()
        assert torch.equal(a, xp) and torch.equal(b, x0)
        # the eps all-gather of the cross-rank CFG split (parallel.run_step "units"): fp32 tensors, one per (tile, branch) unit
        e = rnd((6, 4, 4, 8, 16), 3).to(d)
        g = parallel.exchange_units(e, 6)
        parts = parallel.all_gather_tiles(xp, x0, [8])
        torch.cuda.synchronize()
        assert torch.equal(g, e) and g.dtype == torch.float32
        assert torch.equal(parts[0][0], xp) and torch.equal(parts[0][1], x0)
    finally:
        if created:
            dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------ GroupNorm statistics from the producer
def _in_gemmstats_library(request):
    """ds_gemm_f16_stats lives in the `gemmstats` build variant (the product library is built without it, profiles/r4_notes.md
    section 3): the test re-runs itself in a subprocess with DS_HIP_LIBRARY on that library and returns False in the parent."""
    import subprocess
    import sys
    from dynamicscaler_amd import _lib
    if _lib.load().ds_gemm_has_stats():
        return True
    variant = os.path.join(os.path.dirname(_lib.IN_TREE_LIB), "libdynscaler_hip_gemmstats.so")
    if not os.path.exists(variant):
        pytest.skip("libdynscaler_hip_gemmstats.so not built (python -m dynamicscaler_amd.build --variant gemmstats)")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", request.node.nodeid.replace("::", "::", 1)],
                       capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                       env=dict(os.environ, DS_HIP_LIBRARY=variant))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    return False


@pytest.mark.parametrize("M,N,K,what", [
    (300, 320, 320, "res16"),                 # small tiles, row tail inside a 32-row block
    (257, 256, 128, "plain"),                 # fp16 strips, no epilogue operand
    (161 * 256 - 219, 320, 320, "res16"),     # 256x320 tiles
    (160 * 256 + 3, 256, 64, "bias"),         # 256x256 tiles, bias in the accumulators -> fp16 strips
    (160 * 256, 640, 128, "res32"),           # 256x320 tiles, fp32 residual + fp32 output
    (96, 72, 64, "pib"),                      # per-item bias, a partial last column chunk group
    (2560, 1280, 192, "silu"),                # SiLU in the epilogue
])
def test_gemm_column_statistics_of_the_stored_tile(M, N, K, what, request):
    """ds_gemm_f16_stats: colstats[row block of 32][column] = (sum, sumsq) of the values the launch stores, for every epilogue
    form and tile variant; the table may be a column slice of a wider one (a concat buffer's).  Checked against the sums of the
    launch's own output (fp32 outputs: exactly those values; fp16 outputs: up to their rounding)."""
    from dynamicscaler_amd import ops, _lib
    d = dev()
    if not _in_gemmstats_library(request):
        return
    A, W = _h(rnd((M, K), 1)).half().to(d), _h(rnd((N, K), 2, 0.1)).half().to(d)
    kw, out = {}, None
    if what == "res16":
        kw = dict(residual=(rnd((M, N), 4) * 2).half().to(d), bias=rnd((N,), 3).to(d))
    elif what == "bias":
        kw = dict(bias=rnd((N,), 3).to(d))
    elif what == "res32":
        kw = dict(residual=(rnd((M, N), 4) * 2).to(d), bias=rnd((N,), 3).to(d))
        out = torch.empty((M, N), dtype=torch.float32, device=d)
    elif what == "pib":
        kw = dict(bias=rnd((3, N), 3).to(d), bias_rows=32, ldbias=N)
    elif what == "silu":
        kw = dict(bias=rnd((N,), 3).to(d), epilogue=_lib.DS_EPI_SILU)
    table = torch.full(((M + 31) // 32, N + 24, 2), 7.0, dtype=torch.float32, device=d)        # wider table: columns 16 .. 16 + N are ours
    y = ops.gemm(A, W, M=M, N=N, K=K, out=out, colstats=table[:, 16:16 + N], **kw)
    y_plain = ops.gemm(A, W, M=M, N=N, K=K, out=None if out is None else torch.empty_like(out), **kw)
    assert torch.equal(y, y_plain)                                   # the statistics do not touch the output
    assert bool((table[:, :16] == 7.0).all()) and bool((table[:, 16 + N:] == 7.0).all())
    yf = y.float().cpu()
    pad = (-M) % 32
    yp = torch.cat([yf, torch.zeros((pad, N))]) if pad else yf
    blocks = yp.view(-1, 32, N).double()
    ref_s, ref_q = blocks.sum(1), (blocks * blocks).sum(1)
    got = table[:, 16:16 + N].cpu().double()
    tol = 1e-5 if y.dtype == torch.float32 else 1.5e-3               # fp16 outputs of the fp32-strip path: statistics of the unrounded values
    scale = ref_q.sqrt().clamp_min(1e-3) * (32 ** 0.5)
    assert float(((got[..., 0] - ref_s).abs() / scale).max()) < tol, what
    assert float(((got[..., 1] - ref_q).abs() / ref_q.clamp_min(1e-3)).max()) < 2 * tol, what


@pytest.mark.parametrize("ninst,rows,C,xdt", [(8, 2560, 320, torch.float16), (4, 640, 640, torch.float16), (2, 2560, 1280, torch.float32),
                                              (3, 640, 1920, torch.float16), (16, 320, 640, torch.float16)])
def test_groupnorm_from_producer_statistics(ninst, rows, C, xdt, request):
    """ds_groupnorm_rows_colstats: GroupNorm(32) (+ SiLU) of a GEMM output whose statistics come from the producer's colstats table
    -- a 1920-wide concat buffer whose two halves were written by two launches into one table included -- against the GroupNorm
    that reads the tensor itself (same apply kernel; the group sums differ by the order of summation and, for fp16 tensors, by the
    output's own rounding)."""
    from dynamicscaler_amd import ops
    d = dev()
    if not _in_gemmstats_library(request):
        return
    M, K = ninst * rows, 128
    A = _h(rnd((M, K), 1)).half().to(d)
    g, b = (1 + 0.1 * rnd((C,), 2)).to(d), (0.1 * rnd((C,), 3)).to(d)
    x = torch.empty((M, C), dtype=xdt, device=d)
    table = ops.colstats_table(M, C, d)
    split = C if C != 1920 else 1280                                  # decoder concat: [h | skip], two producers, one table
    for c0, c1 in ((0, split), (split, C)):
        if c0 == c1:
            continue
        W = _h(rnd((c1 - c0, K), 5 + c0, 0.2)).half().to(d)
        ops.gemm(A, W, rnd((c1 - c0,), 6 + c0).to(d), M=M, N=c1 - c0, K=K, out=x[:, c0:c1], colstats=table[:, c0:c1])
    y_ref = ops.groupnorm(x, g, b, ninst, rows, C, 1e-5, True)
    y = ops.groupnorm(x, g, b, ninst, rows, C, 1e-5, True, colstats=table)
    assert float((y.float() - y_ref.float()).abs().max()) <= 2e-3 * float(y_ref.float().abs().max())
    assert relerr(y, y_ref) < 3e-4
    # a column slice of the buffer (the encoder-side reader of a skip tensor that lives in the decoder's concat buffer)
    if C == 1920:
        gs, bs = g[:640].contiguous(), b[:640].contiguous()
        ys_ref = ops.groupnorm(x[:, 1280:], gs, bs, ninst, rows, 640, 1e-5, False)
        ys = ops.groupnorm(x[:, 1280:], gs, bs, ninst, rows, 640, 1e-5, False, colstats=table[:, 1280:])
        assert relerr(ys, ys_ref) < 3e-4


# ------------------------------------------------------------------------------------------------ fused tile ops (round 4)
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("x0s", [(56, 8), (5, 61)])          # aligned (wrapping) and misaligned window origins along W
@pytest.mark.parametrize("host_noise", [True, False])
def test_fused_tile_ops_equal_the_separate_kernels(dtype, x0s, host_noise):
    """ds_ring_gather_renoise == ds_ring_gather + ds_renoise_mix and ds_cfg_ddim_scatter == ds_cfg_ddim + ds_ring_scatter3, bit for
    bit: host noise and the in-kernel Philox stream (tile k from its own counter offset, like one ds_renoise_mix call per tile),
    fp32 / fp16, windows wrapping all three axes, origins that do not allow vector loads, mask frame 0 or per frame."""
    from dynamicscaler_amd import ops
    d = dev()
    pano = rnd((1, 4, 6, 12, 64), 1).to(dtype).to(d)
    maskp = (rnd((6, 12, 64), 2) > 0).to(torch.uint8).to(d)
    origins = [(4, 10, x0s[0]), (1, 2, x0s[1])]                # (f0, y0, x0); tile 4 x 4 x 16: the first wraps F, H and W; disjoint in H
    tile = (4, 4, 16)
    n, numel = len(origins), 4 * 4 * 4 * 16
    c, s, ratio = 0.9, 0.43588989, 0.75
    for mask_frame0 in (True, False):
        noise = rnd((n, 4) + tile, 3).to(dtype).to(d) if host_noise else None
        offs = [1000 + 7 * numel, 1000 + 2 * numel]
        t_f, m_f = ops.ring_gather_renoise(pano, maskp, origins, tile, c, s, ratio, noise=noise, mask_frame0=mask_frame0, seed=11,
                                           tile_offsets=offs, want_mask_tiles=True)
        t_s, m_s = ops.ring_gather(pano, origins, tile, maskp)
        if host_noise:
            ops.renoise_mix_(t_s, m_s, tuple(pano.shape), c, s, ratio, noise=noise, mask_frame0=mask_frame0)
        else:
            ops.renoise_mix_(t_s, m_s, tuple(pano.shape), c, s, ratio, noise=None, mask_frame0=mask_frame0, seed=11, offset=1000,
                             tile_ids=[7, 2])
        assert torch.equal(t_f, t_s) and torch.equal(m_f, m_s), (mask_frame0, host_noise)
    # the update, straight into the panoramas (the two windows are disjoint)
    from oracle import ddim as oddim
    sched = oddim.DDIMSchedule(oddim.DiffusionTables(), 10)
    for eta_noise in (False, True):
        coef = dict(sched.step_coefficients(5))
        sn = None
        if eta_noise:
            coef["sigma"] = 0.3
            sn = rnd((n, 4) + tile, 9).to(dtype).to(d)
        e_c, e_u = rnd((n, 4) + tile, 4).to(d), rnd((n, 4) + tile, 5).to(d)
        p1, p1x, k1 = pano.clone(), torch.zeros_like(pano), maskp.clone()
        p2, p2x, k2 = pano.clone(), torch.zeros_like(pano), maskp.clone()
        ops.cfg_ddim_scatter_(p1, p1x, k1, t_s, e_c, e_u, 7.5, coef, origins, sn)
        xp, x0 = ops.cfg_ddim(t_s, e_c, e_u, tuple(pano.shape), 7.5, coef, sn)
        ops.ring_scatter3(p2, p2x, k2, xp, x0, origins)
        assert torch.equal(p1, p2) and torch.equal(p1x, p2x) and torch.equal(k1, k2), eta_noise


def test_gemm_tail_split_and_launch_share_are_bit_identical():
    """The cut of a persistent big-tile launch along M (gemm_entry: the rows of the full rounds on the big tiles, the rest on small
    tiles; conv / temporal modes through m_base) and the launch-share hint (ops.set_launch_share: rounds planned on half the CUs)
    are scheduling only: the same launch with the hint on and off, and against its two half-size launches, bit for bit -- dense with
    bias + fp32 residual rows, the LayerNorm fold, 3x3 conv with a per-item bias table, temporal conv with a residual."""
    from dynamicscaler_amd import ops, _lib
    d = dev()
    M, N, K = 81920, 320, 320                       # 320 row tiles of 256: one full round on 256 CUs + a quarter
    A = rnd((M, K), 1).half().to(d)
    W = rnd((N, K), 2, 0.05).half().to(d)
    b = rnd((N,), 3).to(d)
    R = rnd((M, N), 4).to(d)

    def both(fn):
        outs = []
        for share in (1, 2):
            ops.set_launch_share(share)
            try:
                outs.append(fn())
            finally:
                ops.set_launch_share(1)
        assert torch.equal(outs[0], outs[1])
        return outs[0]

    full = both(lambda: ops.gemm(A, W, b, R, M=M, N=N, K=K, epilogue=_lib.DS_EPI_OUT_F32))
    h = M // 2
    halves = torch.cat([ops.gemm(A[:h], W, b, R[:h], M=h, N=N, K=K, epilogue=_lib.DS_EPI_OUT_F32),
                        ops.gemm(A[h:], W, b, R[h:], M=h, N=N, K=K, epilogue=_lib.DS_EPI_OUT_F32)])
    assert torch.equal(full, halves)
    assert relerr(full, A.float().cpu() @ W.float().cpu().t() + b.cpu() + R.cpu()) < 1e-5
    # LayerNorm fold (row statistics travel with the cut)
    stats = ops.layernorm_stats(A)
    cs, cb = W.float().sum(1).contiguous(), rnd((N,), 5).to(d)
    full = both(lambda: ops.gemm_ln(A, W, stats, cs, cb, M=M, N=N, K=K))
    halves = torch.cat([ops.gemm_ln(A[:h], W, stats[:h], cs, cb, M=h, N=N, K=K), ops.gemm_ln(A[h:], W, stats[h:], cs, cb, M=h, N=N, K=K)])
    assert torch.equal(full, halves)
    # 3x3 conv, per-item bias (the time-embedding add): 32 images of 40x64, items of 16 images
    nimg, hin, win, cin, cout = 32, 40, 64, 64, 320
    x = rnd((nimg * hin * win, cin), 6).half().to(d)
    w = rnd((cout, 9 * cin), 7, 0.05).half().to(d)
    table = rnd((2, cout + 64), 8).to(d)
    conv = (nimg, hin, win, hin, win, 1, 0)
    full = both(lambda: ops.gemm(x, w, table[:, 64:], None, M=nimg * hin * win, N=cout, K=9 * cin, a_mode=_lib.DS_A_CONV3, cin=cin, lda=cin,
                                 conv=conv, bias_rows=16 * hin * win, ldbias=cout + 64))
    hm = 16 * hin * win
    parts = [ops.gemm(x[i * hm:(i + 1) * hm], w, table[i:i + 1, 64:], None, M=hm, N=cout, K=9 * cin, a_mode=_lib.DS_A_CONV3, cin=cin, lda=cin,
                      conv=(16, hin, win, hin, win, 1, 0), bias_rows=hm, ldbias=cout + 64) for i in range(2)]
    assert torch.equal(full, torch.cat(parts))
    # temporal conv with a residual: 2 sequences of 16 frames x 2560 pixels
    T_, hw = 16, 2560
    xt = rnd((2 * T_ * hw, 320), 9).half().to(d)
    wt = rnd((320, 3 * 320), 10, 0.05).half().to(d)
    full = both(lambda: ops.gemm(xt, wt, b, xt, M=2 * T_ * hw, N=320, K=960, a_mode=_lib.DS_A_TCONV, cin=320, lda=320, tconv=(T_, hw)))
    hs = T_ * hw
    parts = [ops.gemm(xt[i * hs:(i + 1) * hs], wt, b, xt[i * hs:(i + 1) * hs], M=hs, N=320, K=960, a_mode=_lib.DS_A_TCONV, cin=320, lda=320,
                      tconv=(T_, hw)) for i in range(2)]
    assert torch.equal(full, torch.cat(parts))
