"""-m gpu: ds_unet_* -- the UNet as one C call (include/dynscaler_hip.h; DiffusionWrapper.forward -> UNetModel.forward,
lvdm/models/ddpm3d.py:702-712, lvdm/modules/networks/openaimodel3d.py:657-708).

  * the C launch program == the Python restatement of it, bit for bit (same packed operands, same kernels);
  * the raw C-ABI flow a non-Python host would use (create -> load_weight -> pack -> workspace_bytes -> forward), driven
    through ctypes without dynamicscaler_amd.unet, against the reference's golden eps;
  * the full-size model through the C program against the reference's fp32 CPU forward.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(REPO, "tests", "golden")

from test_gpu_fullsize import dev, T, relerr, t2v_params      # noqa: E402
from test_gpu_unet import EPS_TOL, EPS_TOL_TINY               # noqa: E402


def build_unet(params, seed, device, residual_dtype=torch.float16):
    from dynamicscaler_amd.unet import UNetModel
    from dynamicscaler_amd.unet_spec import param_shapes
    from dynamicscaler_amd.synth import synth_state_dict
    m = UNetModel(**params)
    m.load_state_dict(synth_state_dict(param_shapes(params), seed), strict=True)
    m.residual_dtype, m.residual_scope = residual_dtype, "full"
    return m.to(device).eval()


@pytest.mark.parametrize("name", ["t2v", "i2v"])
@pytest.mark.parametrize("residual_dtype", [torch.float16, torch.float32])
def test_c_program_prefix_sharing_and_batch_invariance(name, residual_dtype):
    """ds_unet_forward on the toy UNets: a [cond | uncond] pair batch with the shared context-free prefix equals the plain batch bit
    for bit, a batch equals its separate forwards, fp16 and fp32 inputs agree where the input is fp16-representable, a wrong
    cfg_pairs is refused.  (Rounds 3-4 also compared the C program bit for bit with a Python restatement of it; round 5 removed the
    restatement -- the program is pinned by the reference goldens, tests below and tests/test_gpu_unet.py.)"""
    d = dev()
    from dynamicscaler_amd.synth import synth_normal
    z = np.load(os.path.join(G, f"unet_tiny_{name}.npz"))
    params = json.loads(bytes(z["params_json"]).decode())
    m = build_unet(params, 5, d, residual_dtype)
    x0, c0 = T(z["x_0"]), T(z["ctx_0"])
    for n in (1, 3):
        x = torch.cat([x0] + [synth_normal(x0.shape, 100 + k) for k in range(1, n)], 0).half().float()
        ctx = torch.cat([synth_normal(c0.shape, 200 + k) for k in range(2 * n)], 0).to(d)
        t = torch.tensor([500] * (2 * n), device=d)
        x2 = torch.cat([x, x], 0).to(d, torch.float16)
        plain, shared = m(x2, t, context=ctx, fps=8), m(x2, t, context=ctx, fps=8, cfg_pairs=n)
        assert plain.dtype == torch.float32 and torch.equal(plain, shared)
        assert torch.equal(m(x2.float(), t, context=ctx, fps=8, cfg_pairs=n), shared)          # the input is rounded to the operand type once
        assert torch.equal(m(x2[:1], t[:1], context=ctx[:1], fps=8), plain[:1])
        assert torch.equal(m(x2, t, context=ctx, fps=torch.tensor([8] * (2 * n))), plain)       # the reference's tensor-valued fps
    with pytest.raises(Exception):
        m(x2, t, context=ctx, fps=8, cfg_pairs=2 * n)
    with pytest.raises(NotImplementedError):
        m(x2, t, context=ctx, fps=torch.tensor([8, 16] * n))


@pytest.mark.parametrize("name", ["t2v", "i2v"])
def test_raw_c_abi_flow_vs_reference_golden(name):
    """What a C / Go / Java host does: no dynamicscaler_amd.unet -- only the shared library and device pointers (torch here is
    the allocator).  eps against the reference's own forward (tests/golden/unet_tiny_*.npz)."""
    from dynamicscaler_amd import _lib
    from dynamicscaler_amd.unet_spec import param_shapes
    from dynamicscaler_amd.synth import synth_state_dict
    d = dev()
    lib = _lib.load()
    z = np.load(os.path.join(G, f"unet_tiny_{name}.npz"))
    params = json.loads(bytes(z["params_json"]).decode())
    cfg = _lib.UNetConfig()
    for k in ("in_channels", "out_channels", "model_channels", "num_res_blocks", "transformer_depth", "context_dim"):
        setattr(cfg, k, int(params[k]))
    cfg.temporal_transformer_depth = int(params.get("temporal_transformer_depth", 1))
    cfg.num_head_channels = int(params["num_head_channels"])
    cfg.n_channel_mult = len(params["channel_mult"])
    cfg.n_attention_resolutions = len(params["attention_resolutions"])
    for i, v in enumerate(params["channel_mult"]):
        cfg.channel_mult[i] = v
    for i, v in enumerate(params["attention_resolutions"]):
        cfg.attention_resolutions[i] = v
    for k in ("use_linear", "temporal_conv", "temporal_attention", "addition_attention", "use_image_attention", "fps_cond"):
        setattr(cfg, k, int(bool(params.get(k, False))))
    cfg.fold_layernorm = 1
    cfg.gn_from_producer = 0          # opt-in, needs the gemmstats build of the library
    cfg.temporal_selfatt_only = 1
    h = C.c_void_p()
    _lib.check(lib.ds_unet_create(C.byref(cfg), C.byref(h)), "ds_unet_create")
    try:
        sd = {k: v.to(d) for k, v in synth_state_dict(param_shapes(params), 5).items()}
        assert lib.ds_unet_num_weights(h) == len(sd)
        for k, v in sd.items():
            sh = (C.c_int64 * 5)(*list(v.shape))
            _lib.check(lib.ds_unet_load_weight(h, k.encode(), v.data_ptr(), _lib.DS_F32, sh, v.dim()), "ds_unet_load_weight")
        packed = torch.empty((lib.ds_unet_packed_bytes(h),), dtype=torch.uint8, device=d)
        _lib.check(lib.ds_unet_pack(h, packed.data_ptr(), packed.numel(), None), "ds_unet_pack")
        torch.cuda.synchronize()
        del sd                                            # the raw tensors are not needed after the packing
        for case in range(3):
            x, t, ctx = T(z[f"x_{case}"]).to(d), T(z[f"t_{case}"]).to(d, torch.int64).reshape(-1), T(z[f"ctx_{case}"]).to(d)
            B, _, Tn, H, W = x.shape
            L = ctx.shape[1]
            nws = lib.ds_unet_workspace_bytes(h, B, Tn, H, W, L, 0)
            assert nws > 0
            ws = torch.empty((nws,), dtype=torch.uint8, device=d)
            eps = torch.empty((B, params["out_channels"], Tn, H, W), dtype=torch.float32, device=d)
            _lib.check(lib.ds_unet_forward(h, x.data_ptr(), _lib.DS_F32, t.data_ptr(), ctx.data_ptr(), _lib.DS_F32, L, int(z[f"fps_{case}"]),
                                           B, Tn, H, W, 0, ws.data_ptr(), nws, eps.data_ptr(), None), "ds_unet_forward")
            torch.cuda.synchronize()
            e = relerr(eps, T(z[f"eps_{case}"]))
            print(f"raw C ABI, toy {name} case {case}: eps rel err {e:.3e}")
            assert e < EPS_TOL_TINY
            # a workspace that is too small is refused BEFORE any launch, not overrun: the buffer really is that small, and the
            # guard bytes behind it (and every byte of it) are untouched afterwards
            for short in (nws // 2, nws - 256):
                guard = 1 << 20
                small = torch.full((short + guard,), 0x5A, dtype=torch.uint8, device=d)
                eps2 = torch.full_like(eps, 7.0)
                rc = lib.ds_unet_forward(h, x.data_ptr(), _lib.DS_F32, t.data_ptr(), ctx.data_ptr(), _lib.DS_F32, L, 8, B, Tn, H, W, 0,
                                         small.data_ptr(), short, eps2.data_ptr(), None)
                torch.cuda.synchronize()
                assert rc != 0 and b"workspace too small" in lib.ds_last_error()
                assert bool((small == 0x5A).all()) and bool((eps2 == 7.0).all())
    finally:
        lib.ds_unet_destroy(h)


def test_c_program_full_size_vs_reference_golden():
    """The real t2v UNet through ds_unet_forward: eps against the reference's fp32 CPU forward, and the host time of an eager forward
    (the launch loop in C++: one ctypes call per evaluation)."""
    import time
    from dynamicscaler_amd.synth import synth_normal
    d = dev()
    z = np.load(os.path.join(G, "unet_full_t2v.npz"))
    m = build_unet(t2v_params(), 0, d)
    x = T(z["x"])
    ctx = torch.cat([synth_normal((1, 77, 1024), 1), synth_normal((1, 77, 1024), 2)]).to(d)
    x2 = torch.cat([x, x]).to(d, torch.float16)
    t = torch.tensor([int(z["t"])] * 2, device=d)
    eps = m(x2, t, context=ctx, fps=int(z["fps"]), cfg_pairs=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        eps = m(x2, t, context=ctx, fps=int(z["fps"]), cfg_pairs=1)
    host = (time.perf_counter() - t0) / 3
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 3
    e1, e2 = relerr(eps[:1], T(z["eps_cond"])), relerr(eps[1:], T(z["eps_uncond"]))
    print(f"C program, full UNet eps rel err: cond {e1:.3e} uncond {e2:.3e}; eager forward of one CFG pair: host {host * 1e3:.1f} ms, wall {wall * 1e3:.1f} ms")
    assert e1 < EPS_TOL and e2 < EPS_TOL and host < 0.5 * wall


@pytest.mark.parametrize("mode", ["default", "wide"])
def test_cpp_host_example_equals_the_python_binding(tmp_path, mode):
    """examples/unet_host (C++: only dynscaler_hip.h + the HIP runtime) on the toy i2v UNet: weights, inputs and geometry handed
    over as raw files, eps.bin compared BIT FOR BIT with UNetModel's result (both run ds_unet_forward on identically packed
    operands) and against the reference's golden eps.  "wide": the same host program with ds_unet_config.residual_f32 = 3 (the
    packed buffer then holds the lo planes too): an fp32-level result from a plain C++ caller."""
    import subprocess
    from dynamicscaler_amd import build
    from dynamicscaler_amd.unet_spec import param_shapes
    from dynamicscaler_amd.synth import synth_state_dict
    d = dev()
    exe = build.build_examples(verbose=False)
    z = np.load(os.path.join(G, "unet_tiny_i2v.npz"))
    params = json.loads(bytes(z["params_json"]).decode())
    m = build_unet(params, 5, d)
    sd = synth_state_dict(param_shapes(params), 5)
    x, t, ctx = T(z["x_0"]), T(z["t_0"]).to(torch.int64).reshape(-1), T(z["ctx_0"])
    B, _, Tn, H, W = x.shape
    cfg = (m.twin("wide") if mode == "wide" else m)._c_config()
    assert cfg.residual_f32 == (3 if mode == "wide" else 0)          # (build_unet of this file puts the module in the fp16 residual mode)
    fields = []
    for name, ctype in cfg._fields_:
        v = getattr(cfg, name)
        fields += list(v) if hasattr(v, "__len__") else [v]
    with open(tmp_path / "config.txt", "w") as f:
        f.write(" ".join(str(int(v)) for v in fields) + "\n")
        f.write(f"{B} {Tn} {H} {W} {ctx.shape[1]} {int(z['fps_0'])} 0\n")
    with open(tmp_path / "weights.bin", "wb") as f:
        for k in param_shapes(params):                       # == ds_unet_weight_info order (tests/test_host_cpu.py)
            f.write(sd[k].contiguous().numpy().astype(np.float32).tobytes())
    (tmp_path / "x.bin").write_bytes(x.contiguous().numpy().astype(np.float32).tobytes())
    (tmp_path / "t.bin").write_bytes(t.numpy().astype(np.int64).tobytes())
    (tmp_path / "ctx.bin").write_bytes(ctx.contiguous().numpy().astype(np.float32).tobytes())
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    print(r.stdout.strip())
    eps_c = torch.from_numpy(np.fromfile(tmp_path / "eps.bin", dtype=np.float32).reshape(B, params["out_channels"], Tn, H, W))
    eps_py = m(x.to(d), t.to(d), context=ctx.to(d), fps=int(z["fps_0"]), precision="wide" if mode == "wide" else None).cpu()
    assert torch.equal(eps_c, eps_py)
    assert relerr(eps_c, T(z["eps_0"])) < (2e-5 if mode == "wide" else EPS_TOL_TINY)
