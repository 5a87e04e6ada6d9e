"""-m gpu: the WIDE operand mode (csrc/wide.hip; ds_unet_config.residual_f32 = 3): split-fp16 products with fp32 storage.
Kernel parity against fp64 torch restatements at fp32-level tolerances (the point of the mode: two orders below the fp16 modes'
1e-3), then the whole UNet against the reference's fp32 CPU goldens."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")

# fp32-level agreement: a K = 2880 sum of 22-bit products accumulated in fp32
WIDE_TOL = 2e-6


def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    return torch.device("cuda:0")


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _planes(w, d):
    from dynamicscaler_amd import ops
    return ops.split_f16(w.to(d).contiguous())


def test_split_planes_reconstruct_the_value():
    from dynamicscaler_amd import ops, _lib
    d = dev()
    w = rnd((1000, 64), 1, 0.05)
    w[0, :8] = torch.tensor([0.0, 1.0, -1.0, 1e-7, 65000.0, 3.1e-5, -7.7e-3, 0.333333])
    hi, lo = ops.split_f16(w.to(d))
    S = _lib.load().ds_wide_lo_scale()
    assert S == 2048.0
    assert torch.equal(hi.cpu(), w.half())
    rec = hi.double().cpu() + lo.double().cpu() / S
    err = (rec - w.double()).abs() / w.double().abs().clamp_min(1e-4)
    assert float(err.max()) < 2.0 ** -20, float(err.max())


@pytest.mark.parametrize("M,N,K", [(300, 320, 320), (2, 1280, 320), (1000, 4, 2880), (129, 192, 64), (517, 960, 1280)])
def test_gemm_wide_dense_epilogues(M, N, K):
    from dynamicscaler_amd import ops, _lib
    d = dev()
    A, W = rnd((M, K), 1), rnd((N, K), 2, 0.05)
    b, R = rnd((N,), 3), rnd((M, N), 4)
    hi, lo = _planes(W, d)
    Ad = A.to(d)
    base = A.double() @ W.double().t()
    out = ops.gemm_wide(Ad, hi, lo, None, None, M=M, N=N, K=K)
    assert out.dtype == torch.float32 and relerr(out, base) < WIDE_TOL
    out = ops.gemm_wide(Ad, hi, lo, b.to(d), R.to(d), M=M, N=N, K=K)
    assert relerr(out, base + b.double() + R.double()) < WIDE_TOL
    out = ops.gemm_wide(Ad, hi, lo, b.to(d), None, M=M, N=N, K=K, epilogue=_lib.DS_EPI_SILU)
    assert relerr(out, F.silu(base + b.double())) < WIDE_TOL
    items = 3
    rows = -(-M // items)
    table = rnd((items, N + 64), 5)
    out = ops.gemm_wide(Ad, hi, lo, table.to(d)[:, 64:], None, M=M, N=N, K=K, bias_rows=rows, ldbias=N + 64)
    ref = base + table[:, 64:].double().repeat_interleave(rows, 0)[:M]
    assert relerr(out, ref) < WIDE_TOL
    # strided A (a column slice of a wider buffer) and strided out / residual
    wide_a = rnd((M, K + 128), 6)
    cat = torch.zeros((M, N + 64), dtype=torch.float32, device=d)
    res = rnd((M, N + 32), 7).to(d)
    ops.gemm_wide(wide_a.to(d)[:, 128:], hi, lo, b.to(d), res[:, 32:], M=M, N=N, K=K, lda=K + 128, out=cat[:, 64:])
    ref = wide_a[:, 128:].double() @ W.double().t() + b.double() + res[:, 32:].double().cpu()
    assert relerr(cat[:, 64:], ref) < WIDE_TOL and float(cat[:, :64].abs().max()) == 0.0


def test_gemm_wide_is_two_orders_closer_than_single_fp16_operands():
    from dynamicscaler_amd import ops
    d = dev()
    M, N, K = 512, 320, 1280
    A, W = rnd((M, K), 1), rnd((N, K), 2, 0.03)
    ref = A.double() @ W.double().t()
    hi, lo = _planes(W, d)
    e_wide = relerr(ops.gemm_wide(A.to(d), hi, lo, None, None, M=M, N=N, K=K), ref)
    e_f16 = relerr(ops.gemm(A.half().to(d), W.half().to(d), None, None, M=M, N=N, K=K, epilogue=4), ref)
    assert e_wide < 1e-6 and e_f16 > 100 * e_wide, (e_wide, e_f16)


def test_gemm_wide_geglu():
    from dynamicscaler_amd import ops, _lib
    from dynamicscaler_amd.unet import _interleave_geglu
    d = dev()
    M, K, inner = 333, 320, 1280
    A = rnd((M, K), 1)
    Wg, bg = rnd((2 * inner, K), 6, 0.05), rnd((2 * inner,), 7)
    hi, lo = _planes(_interleave_geglu(Wg), d)
    out = ops.gemm_wide(A.to(d), hi, lo, _interleave_geglu(bg).to(d), None, M=M, N=2 * inner, K=K, epilogue=_lib.DS_EPI_GEGLU)
    xg = A.double() @ Wg.double().t() + bg.double()
    ref = xg[:, :inner] * F.gelu(xg[:, inner:])
    assert out.shape == (M, inner) and relerr(out, ref) < WIDE_TOL


@pytest.mark.parametrize("stride,upsample,hin,win", [(1, 0, 10, 12), (2, 0, 10, 12), (2, 0, 5, 8), (1, 1, 5, 6)])
def test_gemm_wide_conv3x3(stride, upsample, hin, win):
    from dynamicscaler_amd import ops, _lib
    d = dev()
    nimg, cin, cout = 3, 64, 128
    x = rnd((nimg, cin, hin, win), 1)
    w = rnd((cout, cin, 3, 3), 2, 0.05)
    b = rnd((cout,), 3)
    xin = F.interpolate(x, scale_factor=2, mode="nearest") if upsample else x
    ref = F.conv2d(xin.double(), w.double(), b.double(), stride=stride, padding=1)
    hout, wout = ref.shape[-2:]
    a = x.permute(0, 2, 3, 1).reshape(-1, cin).contiguous().to(d)
    hi, lo = _planes(w.permute(0, 2, 3, 1).reshape(cout, -1), d)
    out = ops.gemm_wide(a, hi, lo, b.to(d), None, M=nimg * hout * wout, N=cout, K=9 * cin, a_mode=_lib.DS_A_CONV3, cin=cin, lda=cin,
                        conv=(nimg, hin, win, hout, wout, stride, upsample))
    got = out.cpu().reshape(nimg, hout, wout, cout).permute(0, 3, 1, 2)
    assert relerr(got, ref) < WIDE_TOL


def test_gemm_wide_temporal_conv_with_residual():
    from dynamicscaler_amd import ops, _lib
    d = dev()
    B, T, H, W, Cc = 2, 4, 5, 6, 64
    x = rnd((B, Cc, T, H, W), 1)
    w = rnd((Cc, Cc, 3, 1, 1), 2, 0.05)
    b = rnd((Cc,), 3)
    ref = F.conv3d(x.double(), w.double(), b.double(), padding=(1, 0, 0))
    a = x.permute(0, 2, 3, 4, 1).reshape(-1, Cc).contiguous().to(d)
    hi, lo = _planes(w[:, :, :, 0, 0].permute(0, 2, 1).reshape(Cc, -1), d)
    out = ops.gemm_wide(a, hi, lo, b.to(d), a, M=B * T * H * W, N=Cc, K=3 * Cc, a_mode=_lib.DS_A_TCONV, cin=Cc, lda=Cc, tconv=(T, H * W))
    got = out.cpu().reshape(B, T, H, W, Cc).permute(0, 4, 1, 2, 3)
    assert relerr(got, ref + x.double()) < WIDE_TOL


def test_gemm_wide_rejects_bad_arguments():
    from dynamicscaler_amd import ops, _lib
    d = dev()
    a = torch.zeros((8, 48), dtype=torch.float32, device=d)
    w = torch.zeros((8, 48), dtype=torch.float16, device=d)
    with pytest.raises(_lib.DsError, match="multiple of 64"):
        ops.gemm_wide(a, w, w, None, None, M=8, N=8, K=48)


@pytest.mark.parametrize("ninst,rows,C,silu,strided", [(6, 160, 320, True, False), (2, 640, 640, False, True), (3, 40, 1280, True, False)])
def test_groupnorm_wide(ninst, rows, C, silu, strided):
    from dynamicscaler_amd import ops
    d = dev()
    x = rnd((ninst * rows, C + (64 if strided else 0)), 1) * 3.0 + 0.5
    g, b = rnd((C,), 2) + 1.0, rnd((C,), 3)
    xv = x[:, 64:] if strided else x
    y = ops.groupnorm_wide(x.to(d)[:, 64:] if strided else x.to(d), g.to(d), b.to(d), ninst, rows, C, 1e-5, silu)
    ref = F.group_norm(xv.double().reshape(ninst, rows, C).permute(0, 2, 1), 32, g.double(), b.double(), 1e-5).permute(0, 2, 1).reshape(-1, C)
    if silu:
        ref = F.silu(ref)
    assert relerr(y, ref) < 1e-6


def test_layernorm_wide():
    from dynamicscaler_amd import ops
    d = dev()
    x = rnd((777, 640), 1) * 2.0 - 0.3
    g, b = rnd((640,), 2) + 1.0, rnd((640,), 3)
    y = ops.layernorm_wide(x.to(d), g.to(d), b.to(d))
    assert relerr(y, F.layer_norm(x.double(), (640,), g.double(), b.double(), 1e-5)) < 1e-6


def _attn_ref(q, k, v, scale):
    s = torch.einsum("bhid,bhjd->bhij", q, k) * scale
    return torch.einsum("bhij,bhjd->bhid", s.softmax(-1), v)


@pytest.mark.parametrize("nq,nk,batch,heads,kvdiv", [(160, 160, 3, 2, 1), (700, 700, 2, 1, 1), (40, 40, 4, 3, 1), (200, 77, 4, 2, 2),
                                                      (64, 16, 6, 1, 3), (130, 93, 2, 5, 2)])
def test_attention_wide(nq, nk, batch, heads, kvdiv):
    from dynamicscaler_amd import ops
    d = dev()
    C = heads * 64
    q = rnd((batch * nq, 3 * C), 1)                                   # fused [q | junk | junk] rows: strides like the UNet's QKV output
    kv = rnd((batch // kvdiv * nk, 2 * C), 2)
    o = torch.zeros((batch * nq, C), dtype=torch.float32, device=d)
    qd, kvd = q.to(d), kv.to(d)
    ops.attention_wide(qd, kvd, kvd[:, C:], o, batch=batch, heads=heads, nq=nq, nk=nk, ldq=3 * C, ldk=2 * C, ldv=2 * C, ldo=C,
                       kv_batch_div=kvdiv, scale=0.125)
    qq = q[:, :C].double().reshape(batch, nq, heads, 64).permute(0, 2, 1, 3)
    kk = kv[:, :C].double().reshape(batch // kvdiv, nk, heads, 64).permute(0, 2, 1, 3).repeat_interleave(kvdiv, 0)
    vv = kv[:, C:].double().reshape(batch // kvdiv, nk, heads, 64).permute(0, 2, 1, 3).repeat_interleave(kvdiv, 0)
    ref = _attn_ref(qq, kk, vv, 0.125).permute(0, 2, 1, 3).reshape(batch * nq, C)
    assert relerr(o, ref) < 2e-6
    ops.attention_wide(qd, kvd, kvd[:, C:], o, batch=batch, heads=heads, nq=nq, nk=nk, ldq=3 * C, ldk=2 * C, ldv=2 * C, ldo=C,
                       kv_batch_div=kvdiv, scale=0.125, accumulate=True)
    assert relerr(o, 2 * ref) < 2e-6


@pytest.mark.parametrize("T", [1, 4, 16, 24])
def test_temporal_attention_wide(T):
    from dynamicscaler_amd import ops
    d = dev()
    nb, hw, heads = 2, 37, 5
    C = heads * 64
    qkv = rnd((nb * T * hw, 3 * C), 1)
    o = torch.zeros((nb * T * hw, C), dtype=torch.float32, device=d)
    qd = qkv.to(d)
    ops.temporal_attention_wide(qd, qd[:, C:], qd[:, 2 * C:], o, nseq_batches=nb, T=T, hw=hw, heads=heads, ldq=3 * C, ldk=3 * C, ldv=3 * C,
                                ldo=C, scale=0.125)

    def seq(t):      # [nb*T*hw, C] -> [nb*hw, heads, T, 64]
        return t.double().reshape(nb, T, hw, heads, 64).permute(0, 2, 3, 1, 4).reshape(nb * hw, heads, T, 64)
    ref = _attn_ref(seq(qkv[:, :C]), seq(qkv[:, C:2 * C]), seq(qkv[:, 2 * C:]), 0.125)
    ref = ref.reshape(nb, hw, heads, T, 64).permute(0, 3, 1, 2, 4).reshape(nb * T * hw, C)
    assert relerr(o, ref) < 2e-6


# ------------------------------------------------------------------------------------------------ the UNet in the wide mode
def T(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("name", ["t2v", "i2v"])
def test_toy_unet_wide_matches_the_reference_at_fp32_level(name):
    """The toy UNet (every block kind, image tokens) through precision='wide' against the reference's own forward
    (tests/golden/make_golden.py g8): fp32-level agreement; the module's default mode still answers next to it (two handles,
    one set of parameters); a [cond | uncond] pair batch with the shared prefix equals the plain batch bit for bit."""
    from tests.test_gpu_unet import build_unet
    d = dev()
    z = np.load(os.path.join(GOLD, f"unet_tiny_{name}.npz"))
    params = json.loads(bytes(z["params_json"]).decode())
    m = build_unet(params, 5, d)
    for case in range(3):
        x, t, ctx = T(z[f"x_{case}"]).to(d), T(z[f"t_{case}"]).to(d), T(z[f"ctx_{case}"]).to(d)
        fps = int(z[f"fps_{case}"])
        ref = T(z[f"eps_{case}"])
        e_def = relerr(m(x, t, context=ctx, fps=fps), ref)
        e_wide = relerr(m(x, t, context=ctx, fps=fps, precision="wide"), ref)
        print(f"toy {name} case {case}: eps rel err default {e_def:.3e}, wide {e_wide:.3e}")
        assert e_wide < 2e-5 and e_def > 10 * e_wide
    assert m.twin("wide")._handle is not None and m._handle is not None and m.twin("wide")._handle.value != m._handle.value
    x, t, ctx = T(z["x_0"]).to(d), T(z["t_0"]).to(d), T(z["ctx_0"]).to(d)
    xx, tt = torch.cat([x[:1], x[:1]]), torch.cat([t[:1], t[:1]])
    cc = torch.cat([ctx[:1], ctx[:1].flip(1)])
    a = m(xx, tt, context=cc, fps=8, precision="wide", cfg_pairs=1)
    b = m(xx, tt, context=cc, fps=8, precision="wide")
    assert torch.equal(a, b)
    # the twin follows the parent's parameters
    m.load_state_dict({k: v * 1.0 for k, v in m.state_dict().items()})
    assert m.twin("wide")._packed is None


def test_full_size_unet_wide_vs_reference_golden():
    """North star, wide mode: the full t2v UNet's eps at the real tile inside 1e-3 of the reference's fp32 CPU forward with two
    orders of margin, and -- what the mode exists for -- CFG 7.5 + config 1's FIRST update (4-step schedule, 999 -> 666:
    x_prev = 2.8 x - 1.9 e_t, pipeline/scheduler.py:83-89) inside 1e-3 on x_prev AND pred_x0."""
    import yaml
    from oracle import ddim as oddim
    from dynamicscaler_amd import ops
    from dynamicscaler_amd.synth import synth_normal
    from tests.test_gpu_unet import build_unet
    d = dev()
    z = np.load(os.path.join(GOLD, "unet_full_t2v.npz"))
    params = yaml.safe_load(open(os.path.join(HERE, "t2v_unet_params.yaml")))
    m = build_unet(params, 0, d)
    x = T(z["x"])
    ctx = torch.cat([synth_normal((1, 77, 1024), 1), synth_normal((1, 77, 1024), 2)])
    eps = m(torch.cat([x, x]).to(d), torch.tensor([int(z["t"])] * 2, device=d), context=ctx.to(d), fps=int(z["fps"]), precision="wide",
            cfg_pairs=1)
    ec, eu = T(z["eps_cond"]), T(z["eps_uncond"])
    e1, e2 = relerr(eps[:1], ec), relerr(eps[1:], eu)
    print(f"full UNet eps rel err, wide mode: cond {e1:.3e} uncond {e2:.3e}")
    assert e1 < 5e-5 and e2 < 5e-5
    for steps, index in ((4, 3), (50, 49), (50, 25)):
        sched = oddim.DDIMSchedule(oddim.DiffusionTables(), steps)
        rxp, rx0 = oddim.ddim_step(sched, x, oddim.cfg_combine(ec, eu, 7.5), [index] * 16, noise=torch.zeros_like(x))
        xp, x0 = ops.cfg_ddim(x.to(d), eps[:1].contiguous(), eps[1:].contiguous(), (1, 4, 16, 40, 64), 7.5, sched.step_coefficients(index))
        l1, l2 = relerr(xp, rxp), relerr(x0, rx0)
        print(f"wide: after CFG 7.5 + DDIM index {index} of {steps}: x_prev rel err {l1:.3e}, pred_x0 rel err {l2:.3e}")
        assert l1 < 1e-3 and l2 < 1e-3


def test_full_size_i2v_unet_wide_vs_reference_golden():
    import yaml
    from tests.test_gpu_unet import build_unet
    d = dev()
    z = np.load(os.path.join(GOLD, "unet_full_i2v.npz"))
    params = yaml.safe_load(open(os.path.join(os.path.dirname(HERE), "dynamicscaler_amd", "configs", "i2v_512_v1_unet.yaml")))
    m = build_unet(params, 3, d)
    eps = m(T(z["x"]).to(d), torch.tensor([int(z["t"])], device=d), context=T(z["ctx"]).to(d), fps=int(z["fps"]), precision="wide")
    e = relerr(eps, T(z["eps"]))
    print(f"full i2v UNet eps rel err, wide mode: {e:.3e}")
    assert eps.shape == (1, 4, 16, 40, 64) and e < 5e-5
