"""-m gpu: the HIP UNet and the pipelines against golden vectors captured from the reference (tests/golden) and
against the CPU oracle."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")

# NORTH STAR (BASELINE.json): 1e-3 relative on the latents.  Asserted as such wherever a latent is compared with the reference on the
# path the product runs (the default operand policy: wide evaluation on the steps whose update would carry the guided-eps error past
# 1e-3, tests/test_gpu_fullsize.py, test_gpu_schedule50.py), and on eps itself in the wide mode below.
NORTH_STAR = 1e-3
# REGRESSION GUARDS of the single-fp16-operand modes (fp16 operands / fp32 accumulation vs the reference's fp32 CPU path): <= 2x the
# values measured on MI355X (DESIGN.md section 5 has the table and the error budget that explains them).  Not parity claims.
EPS_TOL_TINY = 4.4e-3   # toy UNet eps, measured 2.0e-3 .. 2.2e-3
EPS_TOL = 2.2e-3        # full-size t2v / i2v UNet eps: measured 1.67e-3 / 1.70e-3 / 1.72e-3 in the fast mode, 1.23e-3 / 1.25e-3 in the default
                        # ("outer") mode; in the wide mode 1.8e-6 (asserted at NORTH_STAR / 20 in the same tests)
LATENT_TOL = 7.2e-4     # x_prev after CFG 7.5 + one DDIM update of the 50-step schedule (index 25), measured 3.6e-4 (inside the north star)
PRED_X0_TOL = 8e-3      # pred_x0 of the same update on fp16 operands, measured 3.9e-3: (x - sqrt(1-a) e_t)/sqrt(a) amplifies the CFG-combined
                        # eps error by sqrt((1-a)/a); an intermediate quantity at that index (only the last step's pred_x0 leaves a loop)
# The toy pipelines end to end (4-6 DDIM steps, CFG 7.5, tiny UNet; ring, grid, sphere, i2v, multi-prompt) against the reference's own
# runs: AT THE NORTH STAR since round 5 -- the operand policy runs the steps of these short schedules that would amplify the guided-eps
# error in the wide mode: measured 3e-6 .. 4e-6 with fp32 latents, 4.6e-4 .. 7.9e-4 with fp16 latents (the stored latent's own
# rounding, 2^-11 per step); rounds 2-4, fp16 operands throughout: 3.7e-3 .. 4.4e-3 under a 9e-3 guard
PIPE_TOL = NORTH_STAR
PIPE_TOL_F16 = NORTH_STAR
VAE_TOL = 5.4e-3        # first stage: decode 1.4e-3 (toy) / 2.6e-3 (real config), encode moments 8.8e-4 / 1.0e-3


def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    return torch.device("cuda:0")


def T(a):
    return torch.from_numpy(np.asarray(a))


def relerr(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-12))


def build_unet(params, seed, device):
    from dynamicscaler_amd.unet import UNetModel
    from dynamicscaler_amd.unet_spec import param_shapes
    from dynamicscaler_amd.synth import synth_state_dict
    m = UNetModel(**params)
    m.load_state_dict(synth_state_dict(param_shapes(params), seed), strict=True)
    return m.to(device).eval()


@pytest.mark.parametrize("name", ["t2v", "i2v"])
def test_unet_tiny_vs_reference_golden(name):
    d = dev()
    z = np.load(os.path.join(G, f"unet_tiny_{name}.npz"))
    params = json.loads(bytes(z["params_json"]).decode())
    m = build_unet(params, 5, d)
    for case in range(3):
        x, t, ctx = T(z[f"x_{case}"]), T(z[f"t_{case}"]), T(z[f"ctx_{case}"])
        for xdt in (torch.float32, torch.float16):
            eps = m(x.to(d, xdt), t.to(d), context=ctx.to(d), fps=int(z[f"fps_{case}"]))
            e = relerr(eps, T(z[f"eps_{case}"]))
            print(f"tiny {name} case {case} {xdt}: eps rel err {e:.3e}")
            assert eps.shape == tuple(z[f"eps_{case}"].shape) and eps.dtype == torch.float32
            assert e < EPS_TOL_TINY


def test_unet_layernorm_fold_option_vs_reference_golden():
    """UNetModel.fold_layernorm (LayerNorm folded into the projection it feeds: ds_layernorm_stats + ds_gemm_f16_ln) against the
    LayerNorm-kernel form: same reference goldens, same tolerance; the normalised activation is never rounded to fp16, so the
    distance to the fp32 reference must not grow.  (ds_gemm_f16_lnk, the statistics inside the GEMM, is covered at kernel level.)"""
    d = dev()
    z = np.load(os.path.join(G, "unet_tiny_t2v.npz"))
    params = json.loads(bytes(z["params_json"]).decode())
    errs = {}
    for fold in (False, True):
        m = build_unet(params, 5, d)
        m.fold_layernorm = fold
        m.invalidate()
        e = []
        for case in range(3):
            x, t, ctx = T(z[f"x_{case}"]), T(z[f"t_{case}"]), T(z[f"ctx_{case}"])
            eps = m(x.to(d, torch.float16), t.to(d), context=ctx.to(d), fps=int(z[f"fps_{case}"]))
            e.append(relerr(eps, T(z[f"eps_{case}"])))
        errs[fold] = e
    print(f"toy UNet eps rel err: LayerNorm kernel {errs[False]}, folded {errs[True]}")
    assert max(errs[True]) < EPS_TOL_TINY and max(errs[True]) < 1.1 * max(errs[False])


@pytest.mark.parametrize("name", ["t2v", "i2v"])
def test_unet_cfg_pair_prefix_sharing_is_bit_identical(name):
    """cfg_pairs=n: the context-free prefix of a [cond | uncond] batch (conv_in, init_attn, first ResBlock, GroupNorm /
    proj_in / self-attention of the first SpatialTransformer) evaluated once and duplicated == the plain 2n forward,
    bit for bit; and the pipelines' switch (share_cfg_prefix) does not change a panorama."""
    d = dev()
    z = np.load(os.path.join(G, f"unet_tiny_{name}.npz"))
    params = json.loads(bytes(z["params_json"]).decode())
    m = build_unet(params, 5, d)
    from dynamicscaler_amd.synth import synth_normal
    x0, c0 = T(z["x_0"]), T(z["ctx_0"])
    for n in (1, 3):
        x = torch.cat([x0] + [synth_normal(x0.shape, 100 + k) for k in range(1, n)], 0).to(d, torch.float16)
        t = T(z["t_0"]).to(d).reshape(-1)[:1].expand(2 * n).contiguous()
        ctx_c = torch.cat([c0] + [synth_normal(c0.shape, 200 + k) for k in range(1, n)], 0)
        ctx_u = torch.cat([synth_normal(c0.shape, 300 + k) for k in range(n)], 0)
        ctx = torch.cat([ctx_c, ctx_u], 0).to(d)
        x2 = torch.cat([x, x], 0)
        plain = m(x2, t, context=ctx, fps=int(z["fps_0"]))
        shared = m(x2, t, context=ctx, fps=int(z["fps_0"]), cfg_pairs=n)
        assert torch.equal(plain, shared), (name, n)
        assert not torch.equal(plain[:n], plain[n:])          # the contexts do differ
    with pytest.raises(ValueError):
        m(x2, t, context=ctx, fps=8, cfg_pairs=2)


def test_ring_pipeline_cfg_prefix_sharing_same_panorama():
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    from dynamicscaler_amd.pipelines import VC2_Pipeline_T2V_SpherePano
    d = dev()
    z = np.load(os.path.join(G, "loops_small.npz"))
    meta = json.load(open(os.path.join(G, "loops_small_traces.json")))
    zt = np.load(os.path.join(G, "unet_tiny_t2v.npz"))
    params = json.loads(bytes(zt["params_json"]).decode())
    ld = _host(params, 5, T(z["cond"]), T(z["uncond"]), d)
    outs = []
    for share in (False, True):
        pipe = VC2_Pipeline_T2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld), {"params": {"unet_config": {"params": params}}}).to(d, torch.float16)
        pipe.share_cfg_prefix = share
        torch.manual_seed(2333333)
        _, den = pipe.basic_sample_shift_multi_windows(prompt="a prompt", fps=8, guidance_scale=7.5, output_type="latent",
                                                       **meta["geoms"]["grid4x2"])
        outs.append((den.float().cpu(), pipe.final_latent.float().cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    # split_cfg_over_streams (the option for levels with a single tile batch, DS_SPLIT_CFG): cond and uncond evaluations as two
    # batches on two streams (graph replays), instead of one [cond | uncond] batch -- same panorama, bit for bit
    pipe = VC2_Pipeline_T2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld), {"params": {"unet_config": {"params": params}}}).to(d, torch.float16)
    pipe.num_streams, pipe.use_graph, pipe.split_cfg_over_streams = 2, True, 2
    torch.manual_seed(2333333)
    _, den = pipe.basic_sample_shift_multi_windows(prompt="a prompt", fps=8, guidance_scale=7.5, output_type="latent",
                                                   **meta["geoms"]["grid4x2"])
    assert torch.equal(den.float().cpu(), outs[0][0]) and torch.equal(pipe.final_latent.float().cpu(), outs[0][1])


def test_unet_batch_equals_separate_forwards():
    d = dev()
    z = np.load(os.path.join(G, "unet_tiny_t2v.npz"))
    params = json.loads(bytes(z["params_json"]).decode())
    m = build_unet(params, 5, d)
    x0, x1 = T(z["x_0"]).to(d), (T(z["x_0"]) * 0.5 + 0.1).to(d)
    c0, c1 = T(z["ctx_0"]).to(d), (T(z["ctx_0"]).flip(1)).contiguous().to(d)
    t = torch.tensor([500, 20], device=d)
    both = m(torch.cat([x0, x1]), t, context=torch.cat([c0, c1]), fps=8)
    a = m(x0, t[:1], context=c0, fps=8)
    b = m(x1, t[1:], context=c1, fps=8)
    assert torch.equal(both[:1], a) and torch.equal(both[1:], b)


def test_unet_full_size_vs_reference_golden():
    """Full t2v UNet (1.41 B parameters) at the real tile [1,4,16,40,64]: eps of the cond and uncond contexts against
    the reference's own fp32 CPU forward (tests/golden/unet_full_t2v.npz), then the latent-level tolerance of
    north_star after CFG + one DDIM update."""
    import yaml
    from oracle import ddim as oddim
    from dynamicscaler_amd import ops
    from dynamicscaler_amd.synth import synth_normal
    d = dev()
    z = np.load(os.path.join(G, "unet_full_t2v.npz"))
    params = yaml.safe_load(open(os.path.join(os.path.dirname(G), "t2v_unet_params.yaml")))
    m = build_unet(params, 0, d)
    x = T(z["x"])
    ctx = torch.cat([synth_normal((1, 77, 1024), 1), synth_normal((1, 77, 1024), 2)])
    eps = m(torch.cat([x, x]).to(d, torch.float16), torch.tensor([int(z["t"])] * 2, device=d), context=ctx.to(d),
            fps=int(z["fps"]))
    ec, eu = T(z["eps_cond"]), T(z["eps_uncond"])
    e1, e2 = relerr(eps[:1], ec), relerr(eps[1:], eu)
    print(f"full UNet eps rel err: cond {e1:.3e} uncond {e2:.3e}")
    assert e1 < EPS_TOL and e2 < EPS_TOL
    sched = oddim.DDIMSchedule(oddim.DiffusionTables(), 50)
    index = 25
    rxp, rx0 = oddim.ddim_step(sched, x, oddim.cfg_combine(ec, eu, 7.5), [index] * 16, noise=torch.zeros_like(x))
    xp, x0 = ops.cfg_ddim(x.to(d, torch.float16), eps[:1].contiguous(), eps[1:].contiguous(), (1, 4, 16, 40, 64), 7.5,
                          sched.step_coefficients(index))
    l1, l2 = relerr(xp, rxp), relerr(x0, rx0)
    print(f"after CFG 7.5 + DDIM step {index}/50: x_prev rel err {l1:.3e}, pred_x0 rel err {l2:.3e}")
    assert l1 < LATENT_TOL and l2 < PRED_X0_TOL
    # the same evaluation in the wide operand mode: eps itself, x_prev AND pred_x0 at the north star (with a factor 20 to spare)
    epw = m(torch.cat([x, x]).to(d), torch.tensor([int(z["t"])] * 2, device=d), context=ctx.to(d), fps=int(z["fps"]), precision="wide")
    w1, w2 = relerr(epw[:1], ec), relerr(epw[1:], eu)
    xpw, x0w = ops.cfg_ddim(x.to(d), epw[:1].contiguous(), epw[1:].contiguous(), (1, 4, 16, 40, 64), 7.5, sched.step_coefficients(index))
    print(f"wide mode: eps rel err {w1:.3e} / {w2:.3e}; x_prev {relerr(xpw, rxp):.3e}, pred_x0 {relerr(x0w, rx0):.3e}")
    assert max(w1, w2, relerr(xpw, rxp), relerr(x0w, rx0)) < NORTH_STAR / 20


def test_unet_full_size_i2v_vs_reference_golden():
    """The model gen_pano_360.py runs -- the i2v 512 UNet with image cross-attention (77 text + 16 image tokens) -- at
    the real tile, against the reference's fp32 CPU forward (tests/golden/unet_full_i2v.npz)."""
    import yaml
    d = dev()
    z = np.load(os.path.join(G, "unet_full_i2v.npz"))
    repo = os.path.dirname(os.path.dirname(G))
    params = yaml.safe_load(open(os.path.join(repo, "dynamicscaler_amd", "configs", "i2v_512_v1_unet.yaml")))
    m = build_unet(params, 3, d)
    eps = m(T(z["x"]).to(d, torch.float16), torch.tensor([int(z["t"])], device=d), context=T(z["ctx"]).to(d),
            fps=int(z["fps"]))
    e = relerr(eps, T(z["eps"]))
    print(f"full i2v UNet eps rel err: {e:.3e}")
    assert eps.shape == (1, 4, 16, 40, 64) and e < EPS_TOL
    epw = m(T(z["x"]).to(d), torch.tensor([int(z["t"])], device=d), context=T(z["ctx"]).to(d), fps=int(z["fps"]), precision="wide")
    print(f"wide mode: eps rel err {relerr(epw, T(z['eps'])):.3e}")
    assert relerr(epw, T(z["eps"])) < NORTH_STAR / 20


def _host(params, seed, cond, uncond, device, temporal_length=4):
    from dynamicscaler_amd.host_model import LatentDiffusionHost
    from dynamicscaler_amd.unet_spec import param_shapes
    from dynamicscaler_amd.synth import synth_state_dict
    ld = LatentDiffusionHost({"params": params}, conditioner=lambda p: uncond if p[0] == "" else cond)
    ld.model.diffusion_model.load_state_dict(synth_state_dict(param_shapes(params), seed), strict=True)
    ld.temporal_length = temporal_length
    return ld.to(device).eval()


def test_pipelines_small_vs_reference_golden():
    """P1 (single tile) and P2 (overlapped ring incl. W overlap) on the toy geometry with the tiny UNet, host noise in
    the reference's RNG order; compared with the reference's own final pred_x0 panoramas."""
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    from dynamicscaler_amd.pipelines import VC2_Pipeline_T2V, VC2_Pipeline_T2V_SpherePano
    d = dev()
    z = np.load(os.path.join(G, "loops_small.npz"))
    meta = json.load(open(os.path.join(G, "loops_small_traces.json")))
    zt = np.load(os.path.join(G, "unet_tiny_t2v.npz"))
    params = json.loads(bytes(zt["params_json"]).decode())
    cond, uncond = T(z["cond"]), T(z["uncond"])
    ld = _host(params, 5, cond, uncond, d)
    cfgd = {"params": {"unet_config": {"params": params}}}
    for dt, tol in ((torch.float32, PIPE_TOL), (torch.float16, PIPE_TOL)):
        pipe = VC2_Pipeline_T2V(ld, lvdm_DDIM_Scheduler(ld), cfgd).to(d, dt)
        torch.manual_seed(2333333)
        _, den = pipe.basic_sample(prompt="a prompt", height=64, width=128, frames=4, fps=8, guidance_scale=7.5,
                                   num_inference_steps=4, output_type="latent")
        e = relerr(den, T(z["basic_tiny"]))
        print(f"basic_sample {dt}: rel err {e:.3e}")
        assert e < tol
        for gname in ("grid4x2", "overlapw"):
            pipe = VC2_Pipeline_T2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld), cfgd).to(d, dt)
            torch.manual_seed(2333333)
            trace = []
            _, den = pipe.basic_sample_shift_multi_windows(
                prompt="a prompt", fps=8, guidance_scale=7.5, output_type="latent",
                step_callback=lambda i, t, wins, p, p0: trace.append((i, t, wins)), **meta["geoms"][gname])
            e = relerr(den, T(z[f"ring_{gname}_tiny"]))
            print(f"ring {gname} {dt}: rel err {e:.3e}")
            assert e < tol
            for (i, t, wins), ref in zip(trace, meta["traces"][gname]):
                assert i == ref["i"] and t == ref["t"] and [list(w) for w in wins] == ref["windows"]


def test_ring_pipeline_hipgraph_and_streams_equal_eager():
    """The UNet evaluation as a hipGraph replay (use_graph) and the tile batches of a level on two HIP streams
    (num_streams) must give the panorama of the plain eager loop bit for bit (same kernels, same order per element)."""
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    from dynamicscaler_amd.pipelines import VC2_Pipeline_T2V_SpherePano
    d = dev()
    z = np.load(os.path.join(G, "loops_small.npz"))
    meta = json.load(open(os.path.join(G, "loops_small_traces.json")))
    zt = np.load(os.path.join(G, "unet_tiny_t2v.npz"))
    params = json.loads(bytes(zt["params_json"]).decode())
    ld = _host(params, 5, T(z["cond"]), T(z["uncond"]), d)
    cfgd = {"params": {"unet_config": {"params": params}}}
    outs = {}
    for mode in ("eager", "graph", "streams", "graph+streams"):
        pipe = VC2_Pipeline_T2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld), cfgd).to(d, torch.float16)
        pipe.use_graph = "graph" in mode
        if "streams" in mode:
            pipe.num_streams, pipe.max_tile_batch = 2, 2
        torch.manual_seed(2333333)
        _, den = pipe.basic_sample_shift_multi_windows(prompt="a prompt", fps=8, guidance_scale=7.5, output_type="latent",
                                                       **meta["geoms"]["grid4x2"])
        outs[mode] = (den.float().cpu(), pipe.final_latent.float().cpu())
        if "graph" in mode:
            assert any(isinstance(v, tuple) for v in pipe._graphs.values()), "no graph was captured"
    for mode in ("graph", "streams", "graph+streams"):
        assert torch.equal(outs[mode][0], outs["eager"][0]) and torch.equal(outs[mode][1], outs["eager"][1]), mode


def test_ring_pipeline_hipgraph_under_process_group():
    """hipGraph capture of the UNet evaluation while an RCCL process group (its watchdog thread) is alive -- the state
    the N > 1 bench runs in: capture uses the thread-local error mode then, and the panorama equals the eager one."""
    import torch.distributed as dist
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    from dynamicscaler_amd.pipelines import VC2_Pipeline_T2V_SpherePano
    d = dev()
    z = np.load(os.path.join(G, "loops_small.npz"))
    meta = json.load(open(os.path.join(G, "loops_small_traces.json")))
    zt = np.load(os.path.join(G, "unet_tiny_t2v.npz"))
    params = json.loads(bytes(zt["params_json"]).decode())
    ld = _host(params, 5, T(z["cond"]), T(z["uncond"]), d)
    cfgd = {"params": {"unet_config": {"params": params}}}
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29534")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=d)
    try:
        outs = {}
        for mode in ("eager", "graph"):
            t = torch.ones(4, device=d)
            dist.all_reduce(t)                                   # in-flight collective work for the watchdog to poll
            pipe = VC2_Pipeline_T2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld), cfgd).to(d, torch.float16)
            pipe.use_graph = mode == "graph"
            torch.manual_seed(2333333)
            _, den = pipe.basic_sample_shift_multi_windows(prompt="a prompt", fps=8, guidance_scale=7.5,
                                                           output_type="latent", **meta["geoms"]["grid4x2"])
            dist.barrier()
            outs[mode] = (den.float().cpu(), pipe.final_latent.float().cpu())
            if mode == "graph":
                assert pipe.use_graph and any(isinstance(v, tuple) for v in pipe._graphs.values()), "capture fell back"
        assert torch.equal(outs["graph"][0], outs["eager"][0]) and torch.equal(outs["graph"][1], outs["eager"][1])
    finally:
        if created:
            dist.destroy_process_group()


def test_device_rng_panorama_independent_of_tile_batching():
    """rng_mode="device" (in-kernel Philox, the bench's mode): a tile's re-noise stream is keyed by (step, tile number),
    so the panorama does not depend on how tiles are batched or sharded -- max_tile_batch 1 == 8, bit for bit."""
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    from dynamicscaler_amd.pipelines import VC2_Pipeline_T2V_SpherePano
    d = dev()
    z = np.load(os.path.join(G, "loops_small.npz"))
    meta = json.load(open(os.path.join(G, "loops_small_traces.json")))
    zt = np.load(os.path.join(G, "unet_tiny_t2v.npz"))
    params = json.loads(bytes(zt["params_json"]).decode())
    ld = _host(params, 5, T(z["cond"]), T(z["uncond"]), d)
    cfgd = {"params": {"unet_config": {"params": params}}}
    outs = []
    for tb in (1, 8):
        pipe = VC2_Pipeline_T2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld, rng_mode="device"), cfgd).to(d, torch.float16)
        pipe.max_tile_batch = tb
        torch.manual_seed(2333333)
        _, den = pipe.basic_sample_shift_multi_windows(prompt="a prompt", fps=8, guidance_scale=7.5, output_type="latent",
                                                       **meta["geoms"]["grid4x2"])
        outs.append((den.float().cpu(), pipe.final_latent.float().cpu()))
        assert torch.isfinite(outs[-1][0]).all()
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][0], outs[1][0])


def _oracle_fake(x, ts, ctx):
    return 0.1 * x + 0.01 * ctx.mean()


def test_ring_pipeline_fake_eps_bit_exact_fp32():
    """The whole loop with the survey's fake eps-model (0.1*x + 0.01*mean(ctx)) in fp32 latents: every HIP tile op on
    the path is then bit-exact, so the final panorama must EQUAL the CPU oracle's for all four toy geometries (incl.
    dock_at_h and num_windows_f=2).  The oracle itself is pinned bit-exactly to the reference's panoramas in the build
    container (tests/test_oracle_golden.py); it is re-run HERE because torch's CPU normal stream is not bit-identical
    across CPU vendors (Intel build container vs the GPU box's EPYC), so seeded golden panoramas only match to ~1e-5
    on another host."""
    from oracle import loops as oloops, ddim as oddim
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    from dynamicscaler_amd.pipelines import VC2_Pipeline_T2V_SpherePano
    d = dev()
    z = np.load(os.path.join(G, "loops_small.npz"))
    meta = json.load(open(os.path.join(G, "loops_small_traces.json")))
    cond, uncond = T(z["cond"]), T(z["uncond"])
    ld = _fake_host(cond, uncond, d)
    for gname, geom in meta["geoms"].items():
        pipe = VC2_Pipeline_T2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld), {"params": {"unet_config": {"params": {"in_channels": 4}}}})
        pipe.to(d, torch.float32)
        torch.manual_seed(2333333)
        _, den = pipe.basic_sample_shift_multi_windows(prompt="a prompt", fps=8, guidance_scale=7.5,
                                                       output_type="latent", **geom)
        torch.manual_seed(2333333)
        ref, _, _ = oloops.t2v_ring_sample(_oracle_fake, oddim.DiffusionTables(), cond, uncond, guidance_scale=7.5, **geom)
        assert torch.equal(den.cpu(), ref), (gname, float((den.cpu() - ref).abs().max()))
        assert relerr(den, T(z[f"ring_{gname}_fake"])) < 1e-4          # and the reference's own panorama (other host's RNG)


def test_ring_pipeline_device_rng_equals_oracle_with_the_restated_philox_stream():
    """rng_mode="device" (what bench.py times): the in-kernel Philox draws replace the host's torch.randn.  With oracle/philox.py
    -- the CPU restatement of that stream, counters = (step, tile within the step, element), key = the scheduler's philox_seed --
    injected into the oracle's re_noise, the oracle's panorama equals the HIP pipeline's to the accuracy of the hardware's fast
    log / sin / cos (fake eps, fp32 latents: every other op on the path is bit-exact).  grid4x2 (8 windows per step, wraps both
    seams) and overlapw (10 windows, one dependency chain), every batching of the tiles."""
    from oracle import loops as oloops, ddim as oddim, philox
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    from dynamicscaler_amd.pipelines import VC2_Pipeline_T2V_SpherePano
    d = dev()
    z = np.load(os.path.join(G, "loops_small.npz"))
    meta = json.load(open(os.path.join(G, "loops_small_traces.json")))
    cond, uncond = T(z["cond"]), T(z["uncond"])
    ld = _fake_host(cond, uncond, d)
    seed = 0x5EED5EED1234
    for gname in ("grid4x2", "overlapw"):
        geom = meta["geoms"][gname]
        n_tiles = geom["num_windows_w"] * geom["num_windows_h"] * geom["num_windows_f"]
        init = torch.from_numpy(np.random.RandomState(5).randn(1, 4, geom["frames"] * geom["num_windows_f"], geom["total_h"] // 8,
                                                                geom["total_w"] // 8).astype(np.float32))
        calls = [0]
        orig = oloops.re_noise

        def philox_re_noise(sched, x_a, idx_a, idx_b, noise=None):
            step, tile = divmod(calls[0], n_tiles)
            calls[0] += 1
            numel = x_a.numel()
            off = lvdm_DDIM_Scheduler.tile_philox_offset(step, numel) + tile * numel
            return orig(sched, x_a, idx_a, idx_b, noise=T(philox.tile_noise(tuple(x_a.shape), seed, off)))

        oloops.re_noise = philox_re_noise
        try:
            ref, _, _ = oloops.t2v_ring_sample(_oracle_fake, oddim.DiffusionTables(), cond, uncond, guidance_scale=7.5,
                                               init_panorama_latent=init, **geom)
        finally:
            oloops.re_noise = orig
        assert calls[0] == n_tiles * (geom["num_inference_steps"] - 1)          # every window of every step but the last re-noises
        for tb, streams in ((8, 1), (3, 2), (1, 1)):
            sched = lvdm_DDIM_Scheduler(ld, rng_mode="device")
            sched.philox_seed = seed
            pipe = VC2_Pipeline_T2V_SpherePano(ld, sched, {"params": {"unet_config": {"params": {"in_channels": 4}}}})
            pipe.to(d, torch.float32)
            pipe.max_tile_batch, pipe.num_streams, pipe.use_graph = tb, streams, False
            _, den = pipe.basic_sample_shift_multi_windows(prompt="a prompt", fps=8, guidance_scale=7.5, output_type="latent",
                                                           init_panorama_latent=init, **geom)
            e = float((den.cpu() - ref).abs().max() / ref.abs().max())
            print(f"device-RNG ring pipeline {gname}, tile batch {tb} x {streams} streams: max abs diff / max |ref| = {e:.2e}")
            assert e < 2e-5, (gname, tb, e)


def test_ring_pipeline_device_rng_with_a_real_unet_vs_oracle():
    """The mode bench.py times -- in-kernel Philox noise, hipGraph, two streams -- with a REAL (tiny) UNet as the eps-model: the
    oracle's ring loop on CPU, fed the restated Philox stream (oracle/philox.py) and the fp32 oracle UNet with the same weights,
    against the HIP pipeline.  overlapw: 10 windows per step in one dependency chain, so every re-noised overlap feeds a UNet
    evaluation of a later window.  Tolerance = the toy pipelines' (fp16 matrix-core operands)."""
    from oracle import loops as oloops, ddim as oddim, philox
    from oracle.unet import unet_forward
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    from dynamicscaler_amd.pipelines import VC2_Pipeline_T2V_SpherePano
    from dynamicscaler_amd.unet_spec import param_shapes
    from dynamicscaler_amd.synth import synth_state_dict
    d = dev()
    z = np.load(os.path.join(G, "loops_small.npz"))
    meta = json.load(open(os.path.join(G, "loops_small_traces.json")))
    zt = np.load(os.path.join(G, "unet_tiny_t2v.npz"))
    params = json.loads(bytes(zt["params_json"]).decode())
    cond, uncond = T(z["cond"]), T(z["uncond"])
    sd = synth_state_dict(param_shapes(params), 5)
    ld = _host(params, 5, cond, uncond, d)
    seed = 0x5EED5EED1234
    geom = meta["geoms"]["overlapw"]
    n_tiles = geom["num_windows_w"] * geom["num_windows_h"] * geom["num_windows_f"]
    init = torch.from_numpy(np.random.RandomState(5).randn(1, 4, geom["frames"] * geom["num_windows_f"], geom["total_h"] // 8,
                                                            geom["total_w"] // 8).astype(np.float32))
    calls = [0]
    orig = oloops.re_noise

    def philox_re_noise(sched, x_a, idx_a, idx_b, noise=None):
        step, tile = divmod(calls[0], n_tiles)
        calls[0] += 1
        numel = x_a.numel()
        off = lvdm_DDIM_Scheduler.tile_philox_offset(step, numel) + tile * numel
        return orig(sched, x_a, idx_a, idx_b, noise=T(philox.tile_noise(tuple(x_a.shape), seed, off)))

    oloops.re_noise = philox_re_noise
    try:
        ref, _, _ = oloops.t2v_ring_sample(lambda x, ts, ctx: unet_forward(sd, params, x, ts, ctx, fps=8), oddim.DiffusionTables(), cond,
                                           uncond, guidance_scale=7.5, init_panorama_latent=init, **geom)
    finally:
        oloops.re_noise = orig
    for tb, streams, graph in ((8, 2, True), (1, 1, False)):
        sched = lvdm_DDIM_Scheduler(ld, rng_mode="device")
        sched.philox_seed = seed
        pipe = VC2_Pipeline_T2V_SpherePano(ld, sched, {"params": {"unet_config": {"params": params}}})
        pipe.to(d, torch.float32)
        pipe.max_tile_batch, pipe.num_streams, pipe.use_graph = tb, streams, graph
        _, den = pipe.basic_sample_shift_multi_windows(prompt="a prompt", fps=8, guidance_scale=7.5, output_type="latent",
                                                       init_panorama_latent=init, **geom)
        e = relerr(den, ref)
        print(f"device-RNG ring pipeline with the tiny UNet, tile batch {tb} x {streams} streams, graph {graph}: rel err {e:.3e}")
        assert e < PIPE_TOL, (tb, streams, e)


def test_ring_pipeline_multi_prompt_vs_oracle_and_reference_golden():
    """R13: `window_multi_prompt_dict` (t2v_sphere_panorama_pipeline.py:561-566, utils/multi_prompt_utils.py:1-7) on the toy
    dock geometry.  Fake eps, fp32 latents: bit-equal to the oracle on this host and 1e-4 from the reference's panorama
    (other host's RNG stream); tiny UNet: within tolerance of the reference's panorama; a window whose lower edge wraps
    (grid4x2) trips the same factor assert as the reference."""
    from oracle import loops as oloops, ddim as oddim
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    from dynamicscaler_amd.pipelines import VC2_Pipeline_T2V_SpherePano
    d = dev()
    z = np.load(os.path.join(G, "loops_multiprompt.npz"))
    meta = json.load(open(os.path.join(G, "loops_multiprompt.json")))
    geoms = json.load(open(os.path.join(G, "loops_small_traces.json")))["geoms"]
    emb = {"a prompt": T(z["emb_a_prompt"]), "": T(z["emb_empty"]), "sky": T(z["emb_sky"]), "ground": T(z["emb_ground"])}
    mp = {float(k): v for k, v in meta["multi_prompt_dict"].items()}
    geom = geoms[meta["geom"]]
    ld = _fake_host(emb["a prompt"], emb[""], d)
    ld.get_learned_conditioning = lambda p: emb[p[0]]
    pipe = VC2_Pipeline_T2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld), {"params": {"unet_config": {"params": {"in_channels": 4}}}})
    pipe.to(d, torch.float32)
    trace = []
    torch.manual_seed(2333333)
    _, den = pipe.basic_sample_shift_multi_windows(prompt="a prompt", fps=8, guidance_scale=7.5, output_type="latent",
                                                   window_multi_prompt_dict=mp,
                                                   step_callback=lambda i, t, wins, p, p0: trace.append(wins), **geom)
    torch.manual_seed(2333333)
    ref, _, _ = oloops.t2v_ring_sample(_oracle_fake, oddim.DiffusionTables(), emb["a prompt"], emb[""], guidance_scale=7.5,
                                       window_multi_prompt_dict=mp, get_learned_conditioning=lambda p: emb[p[0]], **geom)
    assert torch.equal(den.cpu(), ref)
    assert relerr(den, T(z["ring_dock_multiprompt_fake"])) < 1e-4
    assert [[list(w) for w in wins] for wins in trace] == [s["windows"] for s in meta["trace"]]
    with pytest.raises(AssertionError, match="not legal"):
        torch.manual_seed(2333333)
        pipe.basic_sample_shift_multi_windows(prompt="a prompt", fps=8, guidance_scale=7.5, output_type="latent",
                                              window_multi_prompt_dict=mp, **geoms["grid4x2"])
    # tiny UNet
    zt = np.load(os.path.join(G, "unet_tiny_t2v.npz"))
    params = json.loads(bytes(zt["params_json"]).decode())
    ld2 = _host(params, 5, emb["a prompt"], emb[""], d)
    ld2.conditioner = lambda p: emb[p[0]]
    pipe = VC2_Pipeline_T2V_SpherePano(ld2, lvdm_DDIM_Scheduler(ld2), {"params": {"unet_config": {"params": params}}}).to(d, torch.float16)
    torch.manual_seed(2333333)
    _, den = pipe.basic_sample_shift_multi_windows(prompt="a prompt", fps=8, guidance_scale=7.5, output_type="latent",
                                                   window_multi_prompt_dict=mp, **geom)
    e = relerr(den, T(z["ring_dock_multiprompt_tiny"]))
    print(f"multi-prompt ring, tiny UNet: rel err {e:.3e}")
    assert e < PIPE_TOL_F16


@pytest.mark.parametrize("name", ["cfg2_2048x512", "cfg3_4096x512", "cfg3_overlap_nw10", "cfg5_8192x1024x24"])
def test_ring_pipeline_baseline_geometries_fake_eps_bit_exact(name):
    """BASELINE.json's full-size geometries (configs 2, 3, 3 with W overlap, 5: up to an 8192x1024x24f panorama, 64 tiles
    per step) through the HIP tile engine with the fake eps-model in fp32: the final pred-x0 panorama EQUALS the CPU
    oracle's run on this host (which is pinned to the reference's SHA-256 in the build container,
    test_oracle_golden.py::test_g9_baseline_geometry_final_panorama_sha); 50-step schedules are cut to 9 steps of the
    same shifted-window sequence -- every shift phase i % loop_step = 0..7 of the window grid, and the wrap back to 0."""
    import hashlib
    from oracle import loops as oloops, ddim as oddim
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    from dynamicscaler_amd.pipelines import VC2_Pipeline_T2V_SpherePano
    d = dev()
    rec = json.load(open(os.path.join(G, "loop_traces.json")))[name]
    geom = dict(rec["geom"])
    full_schedule = geom["num_inference_steps"] <= 10
    if not full_schedule:
        geom["num_inference_steps"] = 9
    z = np.load(os.path.join(G, "loops_small.npz"))
    cond, uncond = T(z["cond"]), T(z["uncond"])
    ld = _fake_host(cond, uncond, d)
    ld.temporal_length = geom["frames"]
    pipe = VC2_Pipeline_T2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld), {"params": {"unet_config": {"params": {"in_channels": 4}}}})
    pipe.to(d, torch.float32)
    torch.manual_seed(2333333)
    _, den = pipe.basic_sample_shift_multi_windows(prompt="a prompt", fps=8, guidance_scale=7.5, output_type="latent", **geom)
    torch.manual_seed(2333333)
    ref, _, _ = oloops.t2v_ring_sample(_oracle_fake, oddim.DiffusionTables(), cond, uncond, guidance_scale=7.5, **geom)
    assert list(den.shape) == rec["shape"]
    assert torch.equal(den.cpu(), ref), (name, float((den.cpu() - ref).abs().max()))
    if full_schedule and hashlib.sha256(ref.numpy().tobytes()).hexdigest() == rec["denoised_sha256"]:
        print(f"{name}: this host's CPU normal stream matches the build container's: panorama SHA-256 == the reference's")


# ------------------------------------------------------------------------------------------------ P4 / P3
class _FakeModel(torch.nn.Module):
    """The survey's fake eps-model 0.1*x + 0.01*mean(ctx); the mean is taken on the host over the same [1,L,D] shape
    as the reference's tensor (torch's CPU reduction order depends on it)."""
    diffusion_model = None

    def forward(self, x, t, c_crossattn=None, fps=None, **kw):
        ctx = torch.cat(c_crossattn, 1)
        m = torch.stack([0.01 * c[None].float().cpu().mean() for c in ctx]).to(x.device)
        return 0.1 * x.float() + m.reshape(-1, 1, 1, 1, 1)


def _fake_host(cond, uncond, d, embed=None):
    from dynamicscaler_amd.scheduler import DiffusionTables

    class Host:
        pass
    tables = DiffusionTables()
    ld = Host()
    ld.model = _FakeModel()
    for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "num_timesteps", "use_scale"):
        setattr(ld, k, getattr(tables, k))
    ld.uncond_type, ld.temporal_length, ld.device = "empty_seq", 4, d
    ld.get_learned_conditioning = lambda p: uncond if p[0] == "" else cond
    if embed is not None:
        ld.get_image_embeds = embed
        ld.embedder = object()
    return ld


def test_grid_pipeline_vs_reference_golden():
    """P4 (t2v_normal_pipeline.py:213-568): plain shift, crossed jump-odd flags, docking in W/H/F -- fake eps bit-exact
    in fp32, tiny UNet within tolerance, window traces identical."""
    from oracle import loops as oloops, ddim as oddim
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    from dynamicscaler_amd.pipelines import VC2_Pipeline_T2V
    d = dev()
    z = np.load(os.path.join(G, "loops_grid_i2v.npz"))
    meta = json.load(open(os.path.join(G, "loops_grid_i2v_traces.json")))
    cond, uncond = T(z["cond"]), T(z["uncond"])
    ld = _fake_host(cond, uncond, d)
    cfgd = {"params": {"unet_config": {"params": {"in_channels": 4}}}}
    for gname, geom in meta["grid_geoms"].items():
        pipe = VC2_Pipeline_T2V(ld, lvdm_DDIM_Scheduler(ld), cfgd).to(d, torch.float32)
        trace = []
        torch.manual_seed(2333333)
        _, den = pipe.basic_sample_shift_multi_windows(prompt="a prompt", height=64, width=128, frames=4, fps=8,
                                                       guidance_scale=7.5, output_type="latent",
                                                       step_callback=lambda i, t, w, p, p0: trace.append((i, t, w)), **geom)
        torch.manual_seed(2333333)
        oref, _ = oloops.t2v_grid_sample(_oracle_fake, oddim.DiffusionTables(), cond, uncond, height=64, width=128, frames=4,
                                         guidance_scale=7.5, **geom)
        assert torch.equal(den.cpu(), oref), (gname, float((den.cpu() - oref).abs().max()))
        assert relerr(den, T(z[f"grid_{gname}_fake"])) < 1e-4
        for (i, t, wins), ref in zip(trace, meta["traces"][f"grid_{gname}"]):
            assert i == ref["i"] and t == ref["t"] and [list(w) for w in wins] == ref["windows"], (gname, i)
    zt = np.load(os.path.join(G, "unet_tiny_t2v.npz"))
    params = json.loads(bytes(zt["params_json"]).decode())
    ldu = _host(params, 5, cond, uncond, d)
    pipe = VC2_Pipeline_T2V(ldu, lvdm_DDIM_Scheduler(ldu), {"params": {"unet_config": {"params": params}}}).to(d, torch.float16)
    torch.manual_seed(2333333)
    _, den = pipe.basic_sample_shift_multi_windows(prompt="a prompt", height=64, width=128, frames=4, fps=8,
                                                   guidance_scale=7.5, output_type="latent", **meta["grid_geoms"]["plain"])
    e = relerr(den, T(z["grid_plain_tiny"]))
    print(f"grid plain tiny fp16: rel err {e:.3e}")
    assert e < PIPE_TOL


def test_grid_pipeline_random_shuffle_init_vs_reference_golden():
    """random_shuffle_init_frame_stride (pipeline/t2v_normal_pipeline.py:328-337; built in round 5): the init latent's slices shuffled
    with Python's global `random`, the reference's statements literally (it indexes dim 3 with frame indices).  Under the same
    random.seed the HIP pipeline reproduces the reference's panorama (make_golden.py g32), fake eps, bit-exact vs the oracle."""
    import random
    from oracle import loops as oloops, ddim as oddim
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    from dynamicscaler_amd.pipelines import VC2_Pipeline_T2V
    d = dev()
    z = np.load(os.path.join(G, "loops_grid_shuffle.npz"))
    geom = json.loads(bytes(z["geom_json"]).decode())
    cond, uncond = T(z["cond"]), T(z["uncond"])
    ld = _fake_host(cond, uncond, d)
    pipe = VC2_Pipeline_T2V(ld, lvdm_DDIM_Scheduler(ld), {"params": {"unet_config": {"params": {"in_channels": 4}}}}).to(d, torch.float32)
    torch.manual_seed(2333333)
    random.seed(int(z["random_seed"]))
    _, den = pipe.basic_sample_shift_multi_windows(prompt="a prompt", height=64, width=128, frames=4, fps=8, guidance_scale=7.5,
                                                   output_type="latent", **geom)
    torch.manual_seed(2333333)
    random.seed(int(z["random_seed"]))
    oref, _ = oloops.t2v_grid_sample(_oracle_fake, oddim.DiffusionTables(), cond, uncond, height=64, width=128, frames=4, guidance_scale=7.5, **geom)
    assert torch.equal(den.cpu(), oref)
    assert relerr(den, T(z["denoised"])) < 1e-4
    # a panorama lower than its frame count fails like the reference does (its frame indices run off the H axis)
    with pytest.raises((RuntimeError, IndexError)):
        pipe.basic_sample_shift_multi_windows(prompt="a prompt", height=16, width=128, frames=4, fps=8, guidance_scale=7.5, output_type="latent",
                                              **dict(geom, num_windows_h=1, num_windows_f=4, random_shuffle_init_frame_stride=4))


def test_grid_pipeline_pre_denoise_and_residual_merge():
    """R11's pre-denoise start / skip-time / progressive skip / given clear latent and the per-step sparse and dense
    residual merge (t2v_normal_pipeline.py:345-412, 445-468) on the HIP path, fake eps in fp32: the oracle on this host
    (bit-exact vs the reference in the build container) within the bicubic kernel's round-off, the reference's
    panoramas within the cross-host RNG bound; toy UNet in fp16 within the usual tolerance."""
    from oracle import loops as oloops, ddim as oddim
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    from dynamicscaler_amd.pipelines import VC2_Pipeline_T2V
    from dynamicscaler_amd.synth import synth_normal
    d = dev()
    z = np.load(os.path.join(G, "loops_grid_i2v.npz"))
    meta = json.load(open(os.path.join(G, "loops_grid_i2v_traces.json")))
    cond, uncond = T(z["cond"]), T(z["uncond"])
    ld = _fake_host(cond, uncond, d)
    cfgd = {"params": {"unet_config": {"params": {"in_channels": 4}}}}
    for gname, geom in meta["grid_pre_geoms"].items():
        gk = dict(geom)
        if "clear_seed" in gk:
            gk["clear_pre_denoised_latent"] = synth_normal((1, 4, 4, 8, 16), gk.pop("clear_seed"))
        pipe = VC2_Pipeline_T2V(ld, lvdm_DDIM_Scheduler(ld), cfgd).to(d, torch.float32)
        torch.manual_seed(2333333)
        _, den = pipe.basic_sample_shift_multi_windows(prompt="a prompt", height=64, width=128, frames=4, fps=8,
                                                       guidance_scale=7.5, output_type="latent", **gk)
        torch.manual_seed(2333333)
        oref, _ = oloops.t2v_grid_sample(_oracle_fake, oddim.DiffusionTables(), cond, uncond, height=64, width=128, frames=4,
                                         guidance_scale=7.5, **gk)
        e_o, e_r = relerr(den, oref), relerr(den, T(z[f"gridpre_{gname}_fake"]))
        print(f"grid {gname}: vs oracle {e_o:.2e}, vs reference golden {e_r:.2e}")
        assert e_o < 2e-6 and e_r < 1e-4, gname
    zt = np.load(os.path.join(G, "unet_tiny_t2v.npz"))
    params = json.loads(bytes(zt["params_json"]).decode())
    ldu = _host(params, 5, cond, uncond, d)
    pipe = VC2_Pipeline_T2V(ldu, lvdm_DDIM_Scheduler(ldu), {"params": {"unet_config": {"params": params}}}).to(d, torch.float16)
    torch.manual_seed(2333333)
    _, den = pipe.basic_sample_shift_multi_windows(prompt="a prompt", height=64, width=128, frames=4, fps=8,
                                                   guidance_scale=7.5, output_type="latent", **meta["grid_pre_geoms"]["pre_sparse"])
    e = relerr(den, T(z["gridpre_pre_sparse_tiny"]))
    print(f"grid pre_sparse tiny fp16: rel err {e:.3e}")
    assert e < PIPE_TOL


def test_grid_pipeline_clear_video_tensor_start():
    """`clear_pre_denoised_video_tensor` (t2v_normal_pipeline.py:363-368): a clear clip in pixel space is resized bicubically
    to the panorama size, encoded by the first stage (posterior noise in the reference's RNG order), noised to the first
    step's level and merged back every step.  HIP path (resize kernel, first-stage encoder, fake eps, fp32 latents) against
    the oracle's composition of the same pinned pieces (resize_video_latent, encode_first_stage_2dae, _add_noise, grid
    loop); the difference is the fp16 first-stage encoder's (1e-3 on its own, test_vae_encode_vs_reference_golden)."""
    from oracle import loops as oloops, ddim as oddim
    from oracle.vae import encode_first_stage_2dae
    from dynamicscaler_amd.host_model import LatentDiffusionHost
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    from dynamicscaler_amd.pipelines import VC2_Pipeline_T2V
    from dynamicscaler_amd.vae_spec import vae_param_shapes
    from dynamicscaler_amd.synth import synth_state_dict, synth_normal
    d = dev()
    dd = json.loads(bytes(np.load(os.path.join(G, "vae_enc_tiny.npz"))["tiny8_dd_json"]).decode())
    cond, uncond = synth_normal((1, 77, 64), 61), synth_normal((1, 77, 64), 62)
    ld = LatentDiffusionHost({"params": {"in_channels": 4, "out_channels": 4, "model_channels": 64, "attention_resolutions": [],
                                         "num_res_blocks": 1, "channel_mult": [1], "num_head_channels": 64, "context_dim": 64,
                                         "use_linear": True, "temporal_attention": False, "use_relative_position": False}},
                             conditioner=lambda p: uncond if p[0] == "" else cond,
                             first_stage_config={"params": {"ddconfig": dd, "embed_dim": 4}}, scale_factor=0.18215)
    vsd = synth_state_dict(vae_param_shapes(dd, 4), seed=23)
    ld.first_stage_model.load_state_dict(vsd)
    ld.temporal_length = 4
    ld = ld.to(d).eval()
    ld.model = _FakeModel()                     # the survey's fake eps-model: only the start differs from the other grid tests
    clip = synth_normal((1, 3, 4, 48, 96), 91).clamp(-1, 1)
    kw = dict(num_windows_w=2, num_windows_h=2, num_windows_f=1, loop_step=4, num_inference_steps=4, use_pre_denoise=True,
              pre_denoise_steps=2, merge_predenoise_ratio_list=[0.5, 0.6, 0.7, 0.8], sparse_add_residual=True)
    pipe = VC2_Pipeline_T2V(ld, lvdm_DDIM_Scheduler(ld), {"params": {"unet_config": {"params": {"in_channels": 4}}}}).to(d, torch.float32)
    torch.manual_seed(2333333)
    _, den = pipe.basic_sample_shift_multi_windows(prompt="a prompt", height=64, width=128, frames=4, fps=8, guidance_scale=7.5,
                                                   output_type="latent", clear_pre_denoised_video_tensor=clip, **kw)
    torch.manual_seed(2333333)
    oref, _ = oloops.t2v_grid_sample(_oracle_fake, oddim.DiffusionTables(), cond, uncond, height=64, width=128, frames=4,
                                     guidance_scale=7.5, clear_pre_denoised_video_tensor=clip,
                                     encode_first_stage=lambda x: encode_first_stage_2dae(vsd, dd, x, scale_factor=0.18215), **kw)
    e = relerr(den, oref)
    print(f"grid loop started from a clear clip (resize + first-stage encode): rel err vs the oracle {e:.3e}")
    assert den.shape == (1, 4, 4, 16, 32) and e < 3e-5          # measured 1.3e-5 (the fp16 first-stage encoder, diluted by the merge ratios)


def test_i2v_ring_pipeline_vs_reference_golden():
    """P3 (i2v_sphere_panorama_pipeline.py:564-996): round() placement, temporal windows + docking, 5-D mask,
    merge-prev, per-window image tokens, begin_index_offset."""
    from helpers import synth_image_embedder, i2v_geom
    from oracle import loops as oloops, ddim as oddim
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    from dynamicscaler_amd.pipelines_i2v import VC2_Pipeline_I2V_SpherePano
    d = dev()
    z = np.load(os.path.join(G, "loops_grid_i2v.npz"))
    meta = json.load(open(os.path.join(G, "loops_grid_i2v_traces.json")))
    cond, uncond = T(z["cond"]), T(z["uncond"])
    embed = synth_image_embedder(64)
    pano_img = T(z["pano_img"])
    ld = _fake_host(cond, uncond, d, embed)
    cfgd = {"params": {"unet_config": {"params": {"in_channels": 4}}}}
    for gname, geom in meta["i2v_geoms"].items():
        geom = i2v_geom(geom)
        pipe = VC2_Pipeline_I2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld), cfgd).to(d, torch.float32)
        trace = []
        torch.manual_seed(2333333)
        _, den = pipe.basic_sample_shift_multi_windows(prompt="a prompt", fps=8, guidance_scale=7.5, output_type="latent",
                                                       pano_image_tensor=pano_img,
                                                       step_callback=lambda i, t, w, p, p0: trace.append((i, t, w)), **geom)
        torch.manual_seed(2333333)
        uc = torch.cat([uncond, embed(torch.zeros(1, 3, 8, 16))], dim=1)
        oref, _, _ = oloops.i2v_ring_sample(_oracle_fake, embed, oddim.DiffusionTables(), cond, uc, pano_img,
                                            guidance_scale=7.5, **geom)
        assert torch.equal(den.cpu(), oref), (gname, float((den.cpu() - oref).abs().max()))
        assert relerr(den, T(z[f"i2v_{gname}_fake"])) < 1e-4
        for (i, t, wins), ref in zip(trace, meta["traces"][f"i2v_{gname}"]):
            assert i == ref["i"] and t == ref["t"] and [list(w) for w in wins] == ref["windows"], (gname, i)
    zt = np.load(os.path.join(G, "unet_tiny_i2v.npz"))
    params = json.loads(bytes(zt["params_json"]).decode())
    ldu = _host(params, 5, cond, uncond, d)
    ldu.get_image_embeds = embed
    ldu.embedder = object()
    pipe = VC2_Pipeline_I2V_SpherePano(ldu, lvdm_DDIM_Scheduler(ldu), {"params": {"unet_config": {"params": params}}}).to(d, torch.float16)
    torch.manual_seed(2333333)
    _, den = pipe.basic_sample_shift_multi_windows(prompt="a prompt", fps=8, guidance_scale=7.5, output_type="latent",
                                                   pano_image_tensor=pano_img, **meta["i2v_geoms"]["ring"])
    e = relerr(den, T(z["i2v_ring_tiny"]))
    print(f"i2v ring tiny fp16: rel err {e:.3e}")
    assert e < PIPE_TOL


def test_i2v_ring_pipeline_cfg4_geometry_fake_eps_bit_exact():
    """BASELINE config 4 at full size (i2v ring 4096x512x16f, 8x2 windows, 77 + 16 = 93 context tokens per window, per-window
    image crops, 5-D mask, merge-prev) through the HIP tile engine with the fake eps-model in fp32: window trace equal to
    the reference's, final pred-x0 panorama EQUAL to the oracle's on this host (the oracle is pinned to the reference's
    SHA-256 in the build container, test_oracle_golden.py::test_g20_...)."""
    import hashlib
    from helpers import synth_image_embedder
    from oracle import loops as oloops, ddim as oddim
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    from dynamicscaler_amd.pipelines_i2v import VC2_Pipeline_I2V_SpherePano
    from dynamicscaler_amd.synth import synth_normal
    d = dev()
    rec = json.load(open(os.path.join(G, "loop_trace_cfg4_i2v.json")))
    geom = rec["geom"]
    cond, uncond = synth_normal((1, 77, 64), 61), synth_normal((1, 77, 64), 62)
    embed = synth_image_embedder(rec["embedder_dim"])
    pano_img = synth_normal((3, 512, 4096), rec["pano_img_seed"]).clamp(-1, 1)
    ld = _fake_host(cond, uncond, d, embed)
    ld.temporal_length = 16
    pipe = VC2_Pipeline_I2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld), {"params": {"unet_config": {"params": {"in_channels": 4}}}})
    pipe.to(d, torch.float32)
    trace = []
    torch.manual_seed(2333333)
    _, den = pipe.basic_sample_shift_multi_windows(prompt="a prompt", fps=8, guidance_scale=7.5, output_type="latent",
                                                   pano_image_tensor=pano_img,
                                                   step_callback=lambda i, t, w, p, p0: trace.append((i, t, w)), **geom)
    torch.manual_seed(2333333)
    uc = torch.cat([uncond, embed(torch.zeros(1, 3, 40, 64))], dim=1)
    oref, _, _ = oloops.i2v_ring_sample(_oracle_fake, embed, oddim.DiffusionTables(), cond, uc, pano_img, guidance_scale=7.5, **geom)
    assert list(den.shape) == rec["shape"]
    assert torch.equal(den.cpu(), oref), float((den.cpu() - oref).abs().max())
    for (i, t, wins), ref in zip(trace, rec["trace"]):
        assert i == ref["i"] and t == ref["t"] and [list(w) for w in wins] == ref["windows"], i
    if hashlib.sha256(oref.numpy().tobytes()).hexdigest() == rec["denoised_sha256"]:
        print("cfg4: this host's CPU normal stream matches the build container's: panorama SHA-256 == the reference's")


# ------------------------------------------------------------------------------------------------ sphere path (S1-S4, P5)
def _sphere_geom(geom):
    g = dict(geom)
    g["phi_theta_dict"] = {int(k): v for k, v in g["phi_theta_dict"].items()}
    if "phi_fov_dict" in g:
        g["phi_fov_dict"] = {int(k): v for k, v in g["phi_fov_dict"].items()}
    return g


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_sphere_gather_scatter_bit_exact(dtype):
    """ds_map_gather / ds_map_scatter3 through the PanoramaLatentProxy drop-in vs the oracle (duplicate-winner rule,
    untouched pixels) and vs the reference's recorded round trips."""
    from oracle import sphere as S
    from dynamicscaler_amd.sphere import PanoramaLatentProxy
    from dynamicscaler_amd.synth import synth_normal
    d = dev()
    z = np.load(os.path.join(G, "sphere.npz"))
    pano = T(z["rt_pano"]).to(dtype)
    n = 0
    while f"rt_args_{n}" in z:
        fov, th, ph = z[f"rt_args_{n}"].tolist()
        proxy = PanoramaLatentProxy(pano.to(d))
        view, _ = proxy.get_view_tensor_no_interpolate(fov, th, ph, 16, 8)
        ref_view, _ = S.sphere_gather(pano.float(), fov, th, ph, 16, 8)
        assert torch.equal(view.cpu().float(), ref_view)
        tile = synth_normal((1, 4, 3, 8, 16), 200 + n).to(dtype)
        proxy.set_view_tensor_no_interpolation(tile.to(d), fov, th, ph)
        ref_after = S.sphere_scatter(pano.float().clone(), tile.float(), fov, th, ph)
        assert torch.equal(proxy.get_equirect_tensor().cpu().float(), ref_after)
        if dtype == torch.float32:
            assert torch.equal(view.cpu(), T(z[f"rt_view_{n}"])) and torch.equal(proxy.get_equirect_tensor().cpu(), T(z[f"rt_after_{n}"]))
        n += 1
    assert n == 6


def test_sphere_pipeline_view_get_scale_factor():
    """view_get_scale_factor 2 / 3 (t2v_sphere_panorama_pipeline.py:45,194-203): the reference gathers the view at g x the tile
    size and resizes it back with 'nearest'; here the same pixels come from a sub-sampled gather map.  fp32 + fake eps:
    bit-equal to the oracle on this host (which is bit-equal to the reference's panoramas in the build container,
    test_g21_...)."""
    from oracle import sphere as S, ddim as oddim
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    from dynamicscaler_amd.sphere import VC2_Pipeline_T2V_SpherePano
    d = dev()
    z = np.load(os.path.join(G, "sphere_scale.npz"))
    meta = json.load(open(os.path.join(G, "sphere_scale.json")))
    cond, uncond = T(z["cond"]), T(z["uncond"])
    ld = _fake_host(cond, uncond, d)
    for gname, geom in meta["geoms"].items():
        g = _sphere_geom(geom)
        pipe = VC2_Pipeline_T2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld), {"params": {"unet_config": {"params": {"in_channels": 4}}}})
        pipe.to(d, torch.float32)
        torch.manual_seed(2333333)
        final, den = pipe.basic_sample_shift_shpere_panorama(prompt="a prompt", fps=8, guidance_scale=7.5, output_type="latent", **g)
        torch.manual_seed(2333333)
        of, od = S.t2v_sphere_sample(_oracle_fake, oddim.DiffusionTables(), cond, uncond, guidance_scale=7.5, **g)
        assert torch.equal(final.cpu(), of) and torch.equal(den.cpu(), od), (gname, float((final.cpu() - of).abs().max()))
        e_host = relerr(den, T(z[f"sphere_{gname}_denoised"]))
        print(f"sphere {gname} (get scale factor) vs the reference's panorama (other host's RNG stream): {e_host:.3e}")
        assert e_host < 5e-2


def test_sphere_pipelines_view_set_scale_factor_and_downsample():
    """view_set_scale_factor 2 / 3 of both sphere loops (t2v_sphere_panorama_pipeline.py:268-275, i2v_sphere_panorama_pipeline.py:
    421-428): x_prev / pred_x0 / the mask's ones are resized up by s with 'nearest' and scattered through the map of the (s h) x (s w)
    view -- here s * s scatters of the tile through sub-sampled maps whose duplicate targets were resolved on the scaled view (last
    source in row-major order: the reference's result on one thread, golden g33) -- and downsample_factor_before_vae_decode
    (:298-305 / :481-488, ds_resize_latent).  Also with a get scale factor, a per-phi fov, frame windows + docking, paste_on_static.
    fp32 + fake eps: bit-exact vs the oracle on this host (the oracle equals the reference bit for bit, test_g33_...); loosely vs
    the reference's panoramas (another host's RNG stream).  merge-prev with a set scale factor raises like the reference."""
    from helpers import synth_image_embedder
    from oracle import sphere as S, ddim as oddim
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    from dynamicscaler_amd.sphere import VC2_Pipeline_T2V_SpherePano, VC2_Pipeline_I2V_SpherePano
    d = dev()
    zs = np.load(os.path.join(G, "sphere_set_scale.npz"))
    meta = json.load(open(os.path.join(G, "sphere_set_scale.json")))
    cond, uncond = T(zs["cond"]), T(zs["uncond"])
    ld = _fake_host(cond, uncond, d)
    for gname, geom in meta["geoms"].items():
        g = _sphere_geom(geom)
        pipe = VC2_Pipeline_T2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld), {"params": {"unet_config": {"params": {"in_channels": 4}}}})
        pipe.to(d, torch.float32)
        torch.manual_seed(2333333)
        final, den = pipe.basic_sample_shift_shpere_panorama(prompt="a prompt", fps=8, guidance_scale=7.5, output_type="latent", **g)
        torch.manual_seed(2333333)
        of, od = S.t2v_sphere_sample(_oracle_fake, oddim.DiffusionTables(), cond, uncond, guidance_scale=7.5, **g)
        assert final.shape == of.shape and den.shape == od.shape
        assert torch.equal(final.cpu(), of) and torch.equal(den.cpu(), od), (gname, float((final.cpu() - of).abs().max()))
        e_host = relerr(den, T(zs[f"sphere_{gname}_denoised"]))
        print(f"sphere {gname} (set scale factor) vs the reference's panorama (other host's RNG stream): {e_host:.3e}")
        assert e_host < 5e-2
    z = np.load(os.path.join(G, "sphere_i2v.npz"))
    pano_img, static = T(z["pano_img"]), T(z["static_latent"])
    embed = synth_image_embedder(64)
    uc = torch.cat([uncond, embed(torch.zeros(1, 3, 8, 16))], dim=1)
    ld = _fake_host(cond, uncond, d, embed)
    for name, geom in meta["i2v_cases"].items():
        g = dict(geom)
        g["phi_theta_dict"] = {int(k): v for k, v in g["phi_theta_dict"].items()}
        pipe = VC2_Pipeline_I2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld), {"params": {"unet_config": {"params": {"in_channels": 4}}}})
        pipe.to(d, torch.float32)
        torch.manual_seed(2333333)
        final, den = pipe.basic_sample_shift_shpere_panorama(prompt="a prompt", fps=8, guidance_scale=7.5, output_type="latent",
                                                             pano_image_tensor=pano_img, static_frame_latent=static, **g)
        torch.manual_seed(2333333)
        of, od = S.i2v_sphere_sample(_oracle_fake, embed, oddim.DiffusionTables(), cond, uc, pano_img, guidance_scale=7.5,
                                     static_frame_latent=static, **g)
        assert final.shape == of.shape and den.shape == od.shape
        assert torch.equal(final.cpu(), of) and torch.equal(den.cpu(), od), (name, float((final.cpu() - of).abs().max()))
        e_hf, e_hd = relerr(final, T(zs[f"i2v_{name}_final"])), relerr(den, T(zs[f"i2v_{name}_denoised"]))
        # (a sanity bound only: the re-noise draws of this host's torch differ from the build container's, and with a ratio < 1 over
        # frame windows that noise is most of the difference -- 6.7e-2 on the first MI355X box; parity is the bit-exact chain above)
        print(f"i2v sphere set scale factor {name} vs the reference's panoramas (other host's RNG stream): {e_hf:.3e} {e_hd:.3e}")
        assert e_hf < 0.2 and e_hd < 0.2
    base = dict(json.load(open(os.path.join(G, "sphere_i2v_traces.json")))["geoms"]["base"], view_set_scale_factor=2)
    base["phi_theta_dict"] = {int(k): v for k, v in base["phi_theta_dict"].items()}
    assert meta["merge_prev_with_set_scale_raises"] == "RuntimeError"
    with pytest.raises(RuntimeError, match="must match the size"):
        pipe.basic_sample_shift_shpere_panorama(prompt="a prompt", fps=8, guidance_scale=7.5, output_type="latent",
                                                pano_image_tensor=pano_img, static_frame_latent=static, **base)


def test_sphere_pipeline_vs_oracle_and_reference_golden():
    """P5 (t2v): the whole sphere loop.  fp32 + fake eps: bit-exact vs the oracle on this host (incl. the reference's
    stride-dependent RNG quirk and a per-phi fov dict); tiny UNet fp16 within tolerance of the reference's panoramas."""
    from oracle import sphere as S, ddim as oddim
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    from dynamicscaler_amd.sphere import VC2_Pipeline_T2V_SpherePano
    d = dev()
    z = np.load(os.path.join(G, "sphere.npz"))
    meta = json.load(open(os.path.join(G, "sphere_traces.json")))
    cond, uncond = T(z["cond"]), T(z["uncond"])
    ld = _fake_host(cond, uncond, d)
    for gname, geom in meta["geoms"].items():
        g = _sphere_geom(geom)
        pipe = VC2_Pipeline_T2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld), {"params": {"unet_config": {"params": {"in_channels": 4}}}})
        pipe.to(d, torch.float32)
        trace = []
        torch.manual_seed(2333333)
        final, den = pipe.basic_sample_shift_shpere_panorama(prompt="a prompt", fps=8, guidance_scale=7.5, output_type="latent",
                                                             step_callback=lambda i, t, v, p, p0: trace.append((i, t, v)), **g)
        torch.manual_seed(2333333)
        of, od = S.t2v_sphere_sample(_oracle_fake, oddim.DiffusionTables(), cond, uncond, guidance_scale=7.5, **g)
        assert torch.equal(final.cpu(), of) and torch.equal(den.cpu(), od), (gname, float((final.cpu() - of).abs().max()))
        # vs the reference's own panorama generated on another CPU: the scalar-path normal stream behind the sphere loop's
        # re_noise (and possibly a floor() flip in an index map) is host dependent, so this is only a loose sanity bound
        e_host = relerr(den, T(z[f"sphere_{gname}_fake_denoised"]))
        print(f"sphere {gname} fake eps vs the reference's panorama (other host's RNG stream): {e_host:.3e}")
        assert e_host < 5e-2
        for (i, t, views), ref in zip(trace, meta["traces"][gname]):
            assert i == ref["i"] and t == ref["t"] and [list(v) for v in views] == ref["views"], (gname, i)
    zt = np.load(os.path.join(G, "unet_tiny_t2v.npz"))
    params = json.loads(bytes(zt["params_json"]).decode())
    ldu = _host(params, 5, cond, uncond, d)
    pipe = VC2_Pipeline_T2V_SpherePano(ldu, lvdm_DDIM_Scheduler(ldu), {"params": {"unet_config": {"params": params}}}).to(d, torch.float16)
    torch.manual_seed(2333333)
    final, den = pipe.basic_sample_shift_shpere_panorama(prompt="a prompt", fps=8, guidance_scale=7.5, output_type="latent",
                                                         **_sphere_geom(meta["geoms"]["base"]))
    e1, e2 = relerr(final, T(z["sphere_base_tiny_final"])), relerr(den, T(z["sphere_base_tiny_denoised"]))
    print(f"sphere base tiny fp16: final rel err {e1:.3e}, denoised rel err {e2:.3e}")
    assert e1 < PIPE_TOL and e2 < PIPE_TOL


def test_i2v_sphere_pipeline_vs_oracle_and_reference_golden():
    """P5 (i2v, i2v_sphere_panorama_pipeline.py:31-495): frame windows over the F ring (+ docking), per-view image tokens,
    5-D mask, merge-prev, paste_on_static.  fp32 + fake eps: bit-exact vs the oracle re-run on this host and (plain randn
    streams only) close to the reference's own panoramas; tiny i2v UNet in fp16 within tolerance of the reference."""
    from helpers import synth_image_embedder
    from oracle import sphere as S, ddim as oddim
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    from dynamicscaler_amd.sphere import VC2_Pipeline_I2V_SpherePano
    d = dev()
    z = np.load(os.path.join(G, "sphere_i2v.npz"))
    meta = json.load(open(os.path.join(G, "sphere_i2v_traces.json")))
    cond, uncond, pano_img, static = T(z["cond"]), T(z["uncond"]), T(z["pano_img"]), T(z["static_latent"])
    embed = synth_image_embedder(64)
    uc = torch.cat([uncond, embed(torch.zeros(1, 3, 8, 16))], dim=1)
    ld = _fake_host(cond, uncond, d, embed)

    def geom_of(name):
        g = dict(meta["geoms"][name])
        g["phi_theta_dict"] = {int(k): v for k, v in g["phi_theta_dict"].items()}
        return g
    for gname in meta["geoms"]:
        g = geom_of(gname)
        pipe = VC2_Pipeline_I2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld), {"params": {"unet_config": {"params": {"in_channels": 4}}}})
        pipe.to(d, torch.float32)
        trace = []
        torch.manual_seed(2333333)
        final, den = pipe.basic_sample_shift_shpere_panorama(
            prompt="a prompt", fps=8, guidance_scale=7.5, output_type="latent", pano_image_tensor=pano_img,
            static_frame_latent=static, step_callback=lambda i, t, v, p, p0: trace.append((i, t, v)), **g)
        torch.manual_seed(2333333)
        of, od = S.i2v_sphere_sample(_oracle_fake, embed, oddim.DiffusionTables(), cond, uc, pano_img, guidance_scale=7.5,
                                     static_frame_latent=static, **g)
        assert torch.equal(final.cpu(), of) and torch.equal(den.cpu(), od), (gname, float((final.cpu() - of).abs().max()))
        # vs the reference's panoramas generated on another CPU: torch's CPU normal stream and a floor() flip in an index
        # map are host dependent (see the t2v sphere test), so this is only a loose sanity bound
        e_hf, e_hd = relerr(final, T(z[f"i2vs_{gname}_fake_final"])), relerr(den, T(z[f"i2vs_{gname}_fake_denoised"]))
        print(f"i2v sphere {gname} fake eps vs the reference's panoramas (other host's RNG stream): {e_hf:.3e} {e_hd:.3e}")
        assert e_hf < 5e-2 and e_hd < 5e-2
        for (i, t, views), ref in zip(trace, meta["traces"][gname]):
            assert i == ref["i"] and t == ref["t"] and [list(v) for v in views] == ref["views"], (gname, i)
    zt = np.load(os.path.join(G, "unet_tiny_i2v.npz"))
    params = json.loads(bytes(zt["params_json"]).decode())
    ldu = _host(params, 5, cond, uncond, d)
    ldu.get_image_embeds = embed
    ldu.embedder = object()
    pipe = VC2_Pipeline_I2V_SpherePano(ldu, lvdm_DDIM_Scheduler(ldu), {"params": {"unet_config": {"params": params}}}).to(d, torch.float16)
    torch.manual_seed(2333333)
    final, den = pipe.basic_sample_shift_shpere_panorama(prompt="a prompt", fps=8, guidance_scale=7.5, output_type="latent",
                                                         pano_image_tensor=pano_img, **geom_of("base"))
    e1, e2 = relerr(final, T(z["i2vs_base_tiny_final"])), relerr(den, T(z["i2vs_base_tiny_denoised"]))
    print(f"i2v sphere base tiny fp16: final rel err {e1:.3e}, denoised rel err {e2:.3e}")
    assert e1 < PIPE_TOL and e2 < PIPE_TOL


def test_i2v_sphere_pipeline_view_get_scale_factor():
    """view_get_scale_factor 2 / 3 of the i2v sphere loop (i2v_sphere_panorama_pipeline.py:58,330-341): a sub-sampled latent
    gather map (the mask view and the scatters stay at the tile size) and the strided-tensor re-noise stream.  fp32 + fake
    eps: bit-exact vs the oracle re-run on this host (the oracle itself equals the reference's goldens bit for bit,
    test_g22_...), loosely vs the reference's panoramas from another host."""
    from helpers import synth_image_embedder
    from oracle import sphere as S, ddim as oddim
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    from dynamicscaler_amd.sphere import VC2_Pipeline_I2V_SpherePano
    d = dev()
    z = np.load(os.path.join(G, "sphere_i2v.npz"))
    zs = np.load(os.path.join(G, "sphere_i2v_scale.npz"))
    cases = json.load(open(os.path.join(G, "sphere_i2v_scale.json")))["cases"]
    cond, uncond, pano_img, static = T(z["cond"]), T(z["uncond"]), T(z["pano_img"]), T(z["static_latent"])
    embed = synth_image_embedder(64)
    uc = torch.cat([uncond, embed(torch.zeros(1, 3, 8, 16))], dim=1)
    ld = _fake_host(cond, uncond, d, embed)
    for name, geom in cases.items():
        g = dict(geom)
        g["phi_theta_dict"] = {int(k): v for k, v in g["phi_theta_dict"].items()}
        pipe = VC2_Pipeline_I2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld), {"params": {"unet_config": {"params": {"in_channels": 4}}}})
        pipe.to(d, torch.float32)
        torch.manual_seed(2333333)
        final, den = pipe.basic_sample_shift_shpere_panorama(prompt="a prompt", fps=8, guidance_scale=7.5, output_type="latent",
                                                             pano_image_tensor=pano_img, static_frame_latent=static, **g)
        torch.manual_seed(2333333)
        of, od = S.i2v_sphere_sample(_oracle_fake, embed, oddim.DiffusionTables(), cond, uc, pano_img, guidance_scale=7.5,
                                     static_frame_latent=static, **g)
        assert torch.equal(final.cpu(), of) and torch.equal(den.cpu(), od), (name, float((final.cpu() - of).abs().max()))
        e_hf, e_hd = relerr(final, T(zs[f"{name}_final"])), relerr(den, T(zs[f"{name}_denoised"]))
        print(f"i2v sphere get scale factor {name} vs the reference's panoramas (other host's RNG stream): {e_hf:.3e} {e_hd:.3e}")
        assert e_hf < 5e-2 and e_hd < 5e-2


@pytest.mark.parametrize("operands", ["f16", "wide"])
def test_vae_decode_vs_reference_golden(operands):
    """N2 decode side: AutoencoderKLDecoder (HIP) against the reference's AutoencoderKL.decode / decode_first_stage_2DAE:
    toy config (fp32 golden) and the real first-stage config on one 40x64 latent frame (320x512 image, fp16 fixture).
    fp16 activations vs the reference's fp32: VAE_TOL (2x measured) rel-L2 on the decoded pixels, a regression guard; the wide
    operand mode (fp32 activations, split-fp16 products): the north star, 1e-3 (the real config's fixture is stored in fp16, whose
    own rounding is ~2.8e-4 of that)."""
    from dynamicscaler_amd.vae import AutoencoderKLDecoder
    from dynamicscaler_amd.vae_spec import decoder_param_shapes
    from dynamicscaler_amd.synth import synth_state_dict
    d = dev()
    tol = NORTH_STAR if operands == "wide" else VAE_TOL
    z = np.load(os.path.join(G, "vae_tiny.npz"))
    dd = json.loads(bytes(z["tiny_dd_json"]).decode())
    m = AutoencoderKLDecoder(dd, 4)
    m.operand_mode = operands
    m.load_state_dict(synth_state_dict(decoder_param_shapes(dd, 4), seed=21))
    zz = T(z["tiny_z"]).to(d)
    frame = m.decode(zz[:, :, 0])
    e1 = relerr(frame, T(z["tiny_frame"]))
    vid = m.decode_frames(zz, in_scale=1.0 / 0.18215)
    e2 = relerr(vid, T(z["tiny_video"]))
    print(f"vae tiny [{operands}]: frame rel err {e1:.3e}, video rel err {e2:.3e}")
    assert frame.shape == (2, 3, 16, 32) and vid.shape == (2, 3, 3, 16, 32) and e1 < tol and e2 < tol
    if operands == "wide":
        assert e1 < 2e-5 and e2 < 2e-5                # fp32 golden: an fp32 evaluation's own rounding
        m.operand_mode = "f16"                        # a mode switch repacks
        assert relerr(m.decode(zz[:, :, 0]), T(z["tiny_frame"])) > 10 * e1
    zf = np.load(os.path.join(G, "vae_full.npz"))
    ddf = json.loads(bytes(zf["full_dd_json"]).decode())
    mf = AutoencoderKLDecoder(ddf, 4)
    mf.operand_mode = operands
    mf.load_state_dict(synth_state_dict(decoder_param_shapes(ddf, 4), seed=22))
    out = mf.decode(T(zf["full_z"]).to(d))
    e3 = relerr(out, T(zf["full_frame"]).float())
    print(f"vae full [{operands}] (1 frame 40x64 -> 320x512): rel err {e3:.3e}")
    assert out.shape == (1, 3, 320, 512) and e3 < tol


@pytest.mark.parametrize("operands", ["f16", "wide"])
def test_vae_decode_in_bands_is_bit_identical(operands):
    """An activation operand of 2 GiB or more cannot go through one launch (32-bit buffer addressing): AutoencoderKL then evaluates a
    3x3 conv image by image and an image in bands of rows with a one-row halo, a 1x1 conv in row chunks, and the mid-block attention
    in blocks of queries (vae.py operand_limit; a 1024 x 8192 frame at 128 fp16 channels is exactly 2 GiB).  Forced here on the real
    config at 40 x 64 and on three toy frames by a tiny limit: the decode must not change by a bit, in either operand mode (bands keep
    an image of >= 2048 pixels on the taps-innermost side of ds_gemm_f16's K-order rule: _conv3_banded)."""
    from dynamicscaler_amd.vae import AutoencoderKLDecoder
    from dynamicscaler_amd.vae_spec import decoder_param_shapes
    from dynamicscaler_amd.synth import synth_state_dict
    d = dev()
    zf = np.load(os.path.join(G, "vae_full.npz"))
    ddf = json.loads(bytes(zf["full_dd_json"]).decode())
    mf = AutoencoderKLDecoder(ddf, 4)
    mf.operand_mode = operands
    mf.load_state_dict(synth_state_dict(decoder_param_shapes(ddf, 4), seed=22))
    zz = T(zf["full_z"]).to(d)
    ref = mf.decode(zz)
    for limit in (3_000_000, 1_000_000, 200_000):
        mf.operand_limit = limit * (2 if operands == "wide" else 1)
        assert torch.equal(mf.decode(zz), ref), limit
    z = np.load(os.path.join(G, "vae_tiny.npz"))
    dd = json.loads(bytes(z["tiny_dd_json"]).decode())
    m = AutoencoderKLDecoder(dd, 4)
    m.operand_mode = operands
    m.load_state_dict(synth_state_dict(decoder_param_shapes(dd, 4), seed=21))
    zt = T(z["tiny_z"]).to(d)
    ref = m.decode_frames(zt, in_scale=1.0 / 0.18215)
    m.operand_limit = 40_000          # several images per chunk, each cut into bands; attention in query blocks of 64
    assert torch.equal(m.decode_frames(zt, in_scale=1.0 / 0.18215), ref)


def test_decode_tail_seam_safe_with_vae():
    """P6: output_type != 'latent' runs the seam-padded per-frame decode (t2v_sphere_panorama_pipeline.py:638-655) through
    the HIP first-stage decoder; checked against the oracle's decoder on the same padded latent."""
    from oracle.vae import decode_first_stage_2dae
    from dynamicscaler_amd.host_model import LatentDiffusionHost
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    from dynamicscaler_amd.pipelines import VC2_Pipeline_T2V_SpherePano
    from dynamicscaler_amd.vae_spec import decoder_param_shapes
    from dynamicscaler_amd.unet_spec import param_shapes
    from dynamicscaler_amd.synth import synth_state_dict
    d = dev()
    z = np.load(os.path.join(G, "loops_small.npz"))
    meta = json.load(open(os.path.join(G, "loops_small_traces.json")))
    zt = np.load(os.path.join(G, "unet_tiny_t2v.npz"))
    params = json.loads(bytes(zt["params_json"]).decode())
    dd = json.loads(bytes(np.load(os.path.join(G, "vae_tiny.npz"))["tiny_dd_json"]).decode())
    cond, uncond = T(z["cond"]), T(z["uncond"])
    ld = LatentDiffusionHost({"params": params}, conditioner=lambda p: uncond if p[0] == "" else cond,
                             first_stage_config={"params": {"ddconfig": dd, "embed_dim": 4}}, scale_factor=0.18215)
    ld.model.diffusion_model.load_state_dict(synth_state_dict(param_shapes(params), 5), strict=True)
    vsd = synth_state_dict(decoder_param_shapes(dd, 4), seed=21)
    ld.first_stage_model.load_state_dict(vsd)
    ld.temporal_length = 4
    ld = ld.to(d).eval()
    pipe = VC2_Pipeline_T2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld), {"params": {"unet_config": {"params": params}}}).to(d, torch.float16)
    torch.manual_seed(2333333)
    videos, den = pipe.basic_sample_shift_multi_windows(prompt="a prompt", fps=8, guidance_scale=7.5, output_type="tensor",
                                                        **meta["geoms"]["grid4x2"])
    # like the reference (t2v_sphere_panorama_pipeline.py:640-642) the second return value is the W-PADDED latent here
    wlat = den.shape[4] * 16 // 18
    assert videos.shape[:3] == (1, 3, den.shape[2]) and videos.shape[3] == den.shape[3] * 2 and videos.shape[4] == wlat * 2
    padded = den.float().cpu()
    inner = padded[..., wlat // 16:-(wlat // 16)]
    assert torch.equal(padded[..., :wlat // 16], inner[..., -(wlat // 16):]) and torch.equal(padded[..., -(wlat // 16):], inner[..., :wlat // 16])
    ref = decode_first_stage_2dae(vsd, dd, padded, scale_factor=0.18215)
    ref = torch.cat(torch.chunk(ref, 18, dim=4)[1:-1], dim=4)
    e = relerr(videos, ref)
    print(f"seam-safe decode tail: rel err {e:.3e}")
    assert e < 1.7e-3                       # measured 8.1e-4


def test_vae_encode_vs_reference_golden():
    """N2 encode side on the HIP kernels: posterior moments (toy 8x config and the real config on a 320x512 image),
    encode_first_stage_2DAE with the reference's seeded posterior noise, and the pipeline's tiled encode."""
    from dynamicscaler_amd.host_model import LatentDiffusionHost
    from dynamicscaler_amd.pipelines_i2v import VC2_Pipeline_I2V_SpherePano
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    from dynamicscaler_amd.vae import AutoencoderKL
    from dynamicscaler_amd.vae_spec import vae_param_shapes
    from dynamicscaler_amd.synth import synth_state_dict
    d = dev()
    z = np.load(os.path.join(G, "vae_enc_tiny.npz"))
    dd = json.loads(bytes(z["tiny8_dd_json"]).decode())
    zt = np.load(os.path.join(G, "unet_tiny_t2v.npz"))
    params = json.loads(bytes(zt["params_json"]).decode())
    ld = LatentDiffusionHost({"params": params}, first_stage_config={"params": {"ddconfig": dd, "embed_dim": 4}},
                             scale_factor=0.18215)
    ld.first_stage_model.load_state_dict(synth_state_dict(vae_param_shapes(dd, 4), seed=23))
    ld = ld.to(d).eval()
    img = T(z["tiny8_img"]).to(d)
    mom, (h, w) = ld.first_stage_model.encode_moments(img[:, :, [0]])
    ref = T(z["tiny8_moments"])                                   # [1,8,h,w]
    got = mom.reshape(1, h, w, 8).permute(0, 3, 1, 2)
    e1 = relerr(got, ref)
    torch.manual_seed(77)
    enc = ld.encode_first_stage_2DAE(img)
    e2 = relerr(enc, T(z["tiny8_encoded"]))
    pipe = VC2_Pipeline_I2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld), {"params": {"unet_config": {"params": params}}})
    torch.manual_seed(78)
    til = pipe.tiled_vae_encode_tensor_simple(T(z["tiny8_big"]).to(d), overlap_h=2, overlap_w=2)
    e3 = relerr(til, T(z["tiny8_tiled"]))
    print(f"vae encode tiny8: moments {e1:.3e}, sampled {e2:.3e}, tiled {e3:.3e}")
    assert e1 < 1.8e-3 and e2 < 6e-4 and e3 < 7e-4 and til.shape == (1, 4, 1, 16, 32)   # measured 8.8e-4 / 2.7e-4 / 3.2e-4
    zf = np.load(os.path.join(G, "vae_enc_full.npz"))
    ddf = json.loads(bytes(zf["full_dd_json"]).decode())
    mf = AutoencoderKL(ddf, 4)
    mf.load_state_dict(synth_state_dict(vae_param_shapes(ddf, 4), seed=24))
    mom, (h, w) = mf.encode_moments(T(zf["full_img"]).float().to(d).unsqueeze(2))
    e4 = relerr(mom.reshape(1, h, w, 8).permute(0, 3, 1, 2), T(zf["full_moments"]))
    print(f"vae encode full (320x512 -> 40x64): moments rel err {e4:.3e}")
    assert (h, w) == (40, 64) and e4 < 2.1e-3   # measured 1.04e-3
    # the wide operand mode: posterior moments inside the north star, toy and real config
    ld.first_stage_model.operand_mode = "wide"
    mom, (h, w) = ld.first_stage_model.encode_moments(img[:, :, [0]])
    e1w = relerr(mom.reshape(1, h, w, 8).permute(0, 3, 1, 2), ref)
    torch.manual_seed(77)
    e2w = relerr(ld.encode_first_stage_2DAE(img), T(z["tiny8_encoded"]))
    mf.operand_mode = "wide"
    mom, (h, w) = mf.encode_moments(T(zf["full_img"]).float().to(d).unsqueeze(2))
    e4w = relerr(mom.reshape(1, h, w, 8).permute(0, 3, 1, 2), T(zf["full_moments"]))
    print(f"vae encode, wide operands: tiny8 moments {e1w:.3e}, sampled {e2w:.3e}; full moments {e4w:.3e}")
    assert e1w < NORTH_STAR and e2w < NORTH_STAR and e4w < NORTH_STAR and e1w < e1 / 10


def test_i2v_sphere_paste_on_static_with_vae_encoder():
    """S6 end to end: paste_on_static with the panorama image VAE-encoded by the tiled first-stage encode at every step (the
    reference redraws the posterior noise each time).  Product (HIP encoder + loop) vs the oracle's composition of its
    pinned parts (tiled_vae_encode + i2v_sphere_sample): the encoder runs in fp16 activations, hence a tolerance."""
    from helpers import synth_image_embedder
    from oracle import sphere as S, ddim as oddim
    from oracle.vae import tiled_vae_encode
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    from dynamicscaler_amd.sphere import VC2_Pipeline_I2V_SpherePano
    from dynamicscaler_amd.vae import AutoencoderKL
    from dynamicscaler_amd.vae_spec import vae_param_shapes
    from dynamicscaler_amd.synth import synth_state_dict
    d = dev()
    z = np.load(os.path.join(G, "sphere_i2v.npz"))
    meta = json.load(open(os.path.join(G, "sphere_i2v_traces.json")))
    dd = json.loads(bytes(np.load(os.path.join(G, "vae_enc_tiny.npz"))["tiny8_dd_json"]).decode())
    cond, uncond, pano_img = T(z["cond"]), T(z["uncond"]), T(z["pano_img"])
    embed = synth_image_embedder(64)
    uc = torch.cat([uncond, embed(torch.zeros(1, 3, 8, 16))], dim=1)
    ld = _fake_host(cond, uncond, d, embed)
    vsd = synth_state_dict(vae_param_shapes(dd, 4), seed=23)
    ld.first_stage_model = AutoencoderKL(dd, 4)
    ld.first_stage_model.load_state_dict(vsd)
    ld.scale_factor = 0.18215

    def enc(x):      # LatentDiffusionHost.encode_first_stage_2DAE on the fake host
        from dynamicscaler_amd.host_model import LatentDiffusionHost
        return LatentDiffusionHost.encode_first_stage_2DAE(ld, x)
    ld.encode_first_stage_2DAE = enc
    g = dict(meta["geoms"]["static"])
    g["phi_theta_dict"] = {int(k): v for k, v in g["phi_theta_dict"].items()}
    pipe = VC2_Pipeline_I2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld), {"params": {"unet_config": {"params": {"in_channels": 4}}}})
    pipe.to(d, torch.float32)
    torch.manual_seed(2333333)
    final, den = pipe.basic_sample_shift_shpere_panorama(prompt="a prompt", fps=8, guidance_scale=7.5, output_type="latent",
                                                         pano_image_tensor=pano_img, **g)
    torch.manual_seed(2333333)
    of, od = S.i2v_sphere_sample(_oracle_fake, embed, oddim.DiffusionTables(), cond, uc, pano_img, guidance_scale=7.5,
                                 static_frame_latent=lambda: tiled_vae_encode(vsd, dd, pano_img[None, :, None], scale_factor=0.18215),
                                 **g)
    e1, e2 = relerr(final, of), relerr(den, od)
    print(f"paste_on_static with the VAE encoder in the loop: final {e1:.3e}, denoised {e2:.3e}")
    assert e1 < 2e-5 and e2 < 2e-5         # measured 9.4e-6 / 5.1e-6 (fake eps: only the first-stage encoder differs)


def test_i2v_grid_pipeline_vs_reference_golden():
    """P4 (i2v): VC2_Pipeline_I2V.basic_sample_shift_multi_windows (i2v_normal_pipeline.py:68-425): fp32 + fake eps bit-exact
    vs the oracle on this host (plain / docking + two frame windows / use_skip_time), window traces equal to the
    reference's, toy i2v UNet in fp16 within tolerance of the reference's panorama."""
    from helpers import synth_image_embedder
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    from dynamicscaler_amd.pipelines_i2v import VC2_Pipeline_I2V
    from dynamicscaler_amd.synth import synth_normal
    from oracle import loops as oloops, ddim as oddim
    d = dev()
    z = np.load(os.path.join(G, "loops_grid_i2v.npz"))
    meta = json.load(open(os.path.join(G, "loops_grid_i2v_traces.json")))
    cond, uncond, img = T(z["cond"]), T(z["uncond"]), T(z["grid_img"])
    embed = synth_image_embedder(64)
    uc = torch.cat([uncond, embed(torch.zeros(1, 3, 8, 16))], dim=1)
    ld = _fake_host(cond, uncond, d, embed)
    cfgd = {"params": {"unet_config": {"params": {"in_channels": 4}}}}

    def geom_of(name):
        g = dict(meta["i2v_grid_geoms"][name])
        if "init_seed" in g:
            g["init_panorama_latent"] = synth_normal((1, 4, g["frames"] * g["num_windows_f"], g["height"] * g["num_windows_h"] // 8,
                                                      g["width"] * g["num_windows_w"] // 8), g.pop("init_seed"))
        return g
    for gname in meta["i2v_grid_geoms"]:
        g = geom_of(gname)
        pipe = VC2_Pipeline_I2V(ld, lvdm_DDIM_Scheduler(ld), cfgd).to(d, torch.float32)
        trace = []
        torch.manual_seed(2333333)
        _, den = pipe.basic_sample_shift_multi_windows(prompt="a prompt", fps=8, guidance_scale=7.5, output_type="latent",
                                                       pano_image_tensor=img,
                                                       step_callback=lambda i, t, w, p, p0: trace.append((i, t, w)), **g)
        torch.manual_seed(2333333)
        oref, _ = oloops.i2v_grid_sample(_oracle_fake, embed, oddim.DiffusionTables(), cond, uc, img, guidance_scale=7.5, **g)
        assert torch.equal(den.cpu(), oref), (gname, float((den.cpu() - oref).abs().max()))
        assert relerr(den, T(z[f"i2vgrid_{gname}_fake"])) < 1e-4
        for (i, t, wins), ref in zip(trace, meta["traces"][f"i2vgrid_{gname}"]):
            assert i == ref["i"] and t == ref["t"] and [list(w) for w in wins] == ref["windows"], (gname, i)
    zt = np.load(os.path.join(G, "unet_tiny_i2v.npz"))
    params = json.loads(bytes(zt["params_json"]).decode())
    ldu = _host(params, 5, cond, uncond, d)
    ldu.get_image_embeds = embed
    ldu.embedder = object()
    pipe = VC2_Pipeline_I2V(ldu, lvdm_DDIM_Scheduler(ldu), {"params": {"unet_config": {"params": params}}}).to(d, torch.float16)
    torch.manual_seed(2333333)
    _, den = pipe.basic_sample_shift_multi_windows(prompt="a prompt", fps=8, guidance_scale=7.5, output_type="latent",
                                                   pano_image_tensor=img, **geom_of("plain"))
    e = relerr(den, T(z["i2vgrid_plain_tiny"]))
    print(f"i2v grid tiny fp16: rel err {e:.3e}")
    assert e < PIPE_TOL


def test_gen_pano_360_stage_chain_runs():
    """The stage chain of gen_pano_360.py:227-370 on toy sizes, every stage on the HIP path: i2v sphere loop with
    paste_on_static (tiled VAE encode each step) and denoise_to_step -> nearest resize -> i2v ring loop resumed with
    use_skip_time -> bicubic x2 + re_noise -> i2v ring loop at 2x -> seam-safe VAE decode.  No golden for the chain (each
    stage is pinned on its own); this checks that the hand-offs compose: shapes, finiteness, and that the last stage's
    decode equals the oracle's decoder on the same latent."""
    from helpers import synth_image_embedder
    from oracle.vae import decode_first_stage_2dae
    from dynamicscaler_amd.host_model import LatentDiffusionHost
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    from dynamicscaler_amd.sphere import VC2_Pipeline_I2V_SpherePano
    from dynamicscaler_amd.tensor_utils import resize_video_latent
    from dynamicscaler_amd.vae_spec import vae_param_shapes
    from dynamicscaler_amd.unet_spec import param_shapes
    from dynamicscaler_amd.synth import synth_state_dict, synth_normal
    d = dev()
    zt = np.load(os.path.join(G, "unet_tiny_i2v.npz"))
    params = json.loads(bytes(zt["params_json"]).decode())
    dd = json.loads(bytes(np.load(os.path.join(G, "vae_enc_tiny.npz"))["tiny8_dd_json"]).decode())
    cond, uncond = synth_normal((1, 77, 64), 61), synth_normal((1, 77, 64), 62)
    embed = synth_image_embedder(64)
    ld = LatentDiffusionHost({"params": params}, conditioner=lambda p: uncond if p[0] == "" else cond,
                             first_stage_config={"params": {"ddconfig": dd, "embed_dim": 4}}, scale_factor=0.18215)
    ld.model.diffusion_model.load_state_dict(synth_state_dict(param_shapes(params), 5), strict=True)
    vsd = synth_state_dict(vae_param_shapes(dd, 4), seed=23)
    ld.first_stage_model.load_state_dict(vsd)
    ld.temporal_length = 4
    ld.get_image_embeds = embed
    ld.embedder = object()
    ld = ld.to(d).eval()
    pipe = VC2_Pipeline_I2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld), {"params": {"unet_config": {"params": params}}}).to(d, torch.float16)
    pano_img = synth_normal((3, 256, 512), 89).clamp(-1, 1)
    N, stop = 6, 3
    torch.manual_seed(1)
    sphere_lat, _ = pipe.basic_sample_shift_shpere_panorama(
        prompt="a prompt", height=64, width=128, frames=4, fps=8, guidance_scale=7.5, pano_image_tensor=pano_img, total_f=4,
        overlap_ratio_list_f=[0.0] * N, loop_step_frame=2, equirect_width=512, equirect_height=256, view_fov=120,
        loop_step_theta=2, phi_theta_dict={0: [0, 120, 240], -60: [0, 180], 60: [0, 180]}, merge_renoised_overlap_latent_ratio=1,
        merge_prev_denoised_ratio_list=[0.3] * N, denoise_to_step=stop, paste_on_static=True, num_inference_steps=N,
        output_type="latent",
        # ... and the rest of what gen_pano_360.py:227-268 passes by name, values as there (several are swallowed by **kwargs in the
        # reference too and travel on to the UNet call, which ignores them)
        img_cond_path=["unused.png"], init_panorama_latent=None, use_skip_time=False, skip_time_step_idx=0, progressive_skip=False,
        loop_step=4, pano_image_path=None, dock_at_f=None, phi_prompt_dict=None, view_get_scale_factor=1, view_set_scale_factor=1,
        downsample_factor_before_vae_decode=1, latents=None, num_videos_per_prompt=1, generator_seed=1)
    assert sphere_lat.shape == (1, 4, 4, 32, 64) and bool(torch.isfinite(sphere_lat.float()).all())
    lat1 = resize_video_latent(sphere_lat.clone(), target_height=32, target_width=64, mode="nearest")
    ring_args = dict(prompt="a prompt", height=64, width=128, frames=4, fps=8, guidance_scale=7.5, num_windows_f=1, loop_step=4,
                     total_f=4, overlap_ratio_list_f=[0.0] * N, loop_step_frame=2, merge_prev_denoised_ratio_list=[0.3] * N,
                     num_inference_steps=N, use_skip_time=True, skip_time_step_idx=stop, progressive_skip=False,
                     # gen_pano_360.py:291-322 / 355-384 also pass:
                     img_cond_path=["unused.png"], pano_image_path=None, dock_at_f=None, latents=None, num_videos_per_prompt=1, generator_seed=1)
    _, lat2 = pipe.basic_sample_shift_multi_windows(init_panorama_latent=lat1, total_h=256, total_w=512, num_windows_h=2 + 3,
                                                    num_windows_w=5, pano_image_tensor=pano_img, output_type="latent", **ring_args)
    assert lat2.shape == lat1.shape and bool(torch.isfinite(lat2.float()).all())
    up = resize_video_latent(lat2.clone(), target_height=64, target_width=128, mode="bicubic")
    pipe.scheduler.make_schedule(N)
    mixed = pipe.scheduler.re_noise(up, 0, N - stop)
    big_img = synth_normal((3, 512, 1024), 90).clamp(-1, 1)
    videos, lat3 = pipe.basic_sample_shift_multi_windows(init_panorama_latent=mixed, total_h=512, total_w=1024, num_windows_h=9,
                                                         num_windows_w=9, pano_image_tensor=big_img, output_type="tensor", **ring_args)
    assert lat3.shape == (1, 4, 4, 64, 144) and videos.shape == (1, 3, 4, 512, 1024) and bool(torch.isfinite(videos).all())
    padded = lat3.float().cpu()                # the decode branch returns the W-padded latent, like the reference
    ref = torch.cat(torch.chunk(decode_first_stage_2dae(vsd, dd, padded, scale_factor=0.18215), 18, dim=4)[1:-1], dim=4)
    e = relerr(videos, ref)
    print(f"stage chain: final decode rel err vs the oracle decoder {e:.3e}")
    assert e < 2.3e-3                       # measured 1.13e-3
