"""-m gpu: full-size parity of the real (1.41 B parameter) UNet over several DDIM steps, and where the distance comes from.

  * BASELINE.json config 1 literally: VC2_Pipeline_T2V.basic_sample (pipeline/t2v_normal_pipeline.py:69-210), one
    512x320x16f tile, 4 DDIM steps, CFG 7.5 -- against per-step vectors captured from the reference itself
    (tests/golden/cfg1_full_t2v.npz, make_golden.py g17).
  * the layer-wise error budget: activations after every block of the HIP UNet against the oracle's (oracle/unet.py).
  * the LDS / register poison run: no kernel of the UNet program reads per-CU state it has not written.

Every measured number is also appended to gpurun_out/measured_parity.jsonl (one JSON object per line) so that the
asserted tolerances can be kept at <= 2x what was measured (DESIGN.md section 5 quotes them).
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(REPO, "tests", "golden")


def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    return torch.device("cuda:0")


def T(a):
    return torch.from_numpy(np.asarray(a))


def relerr(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-12))


def loop_err_stats(got, ref, windows=None):
    """Beyond the global rel-L2: max |diff| / max |ref|, and the WORST rel-L2 over the step's windows (each window's
    box of the panorama latent [1, C, F, H, W], wrapped in H and W like RingLatent) -- an outlier confined to one seam column shows here."""
    a, b = got.float().cpu(), ref.float().cpu()
    out = {"rel_l2": float((a - b).norm() / b.norm().clamp_min(1e-12)), "max_abs_over_ref_inf": float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))}
    if windows:
        H, W = a.shape[-2], a.shape[-1]
        worst = 0.0
        for (l, r, t, dn, f0, f1) in windows:
            ys = torch.arange(t, dn) % H
            xs = torch.arange(l, r) % W
            fa = a[:, :, f0:f1][..., ys, :][..., xs]
            fb = b[:, :, f0:f1][..., ys, :][..., xs]
            worst = max(worst, float((fa - fb).norm() / fb.norm().clamp_min(1e-12)))
        out["worst_window_rel_l2"] = worst
    return out


def record(**kv):
    try:
        os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
        with open(os.path.join(REPO, "gpurun_out", "measured_parity.jsonl"), "a") as f:
            f.write(json.dumps(kv) + "\n")
    except OSError:
        pass


def t2v_params():
    import yaml
    return yaml.safe_load(open(os.path.join(REPO, "dynamicscaler_amd", "configs", "t2v_512_v2_unet.yaml")))


_HOST = {}


def full_host(d):
    """The full t2v model with the synthetic weights of the goldens (seed 0) and conditioner (seeds 1 / 2); built once."""
    if "ld" not in _HOST:
        from dynamicscaler_amd.host_model import LatentDiffusionHost, SyntheticConditioner
        from dynamicscaler_amd.unet_spec import param_shapes
        from dynamicscaler_amd.synth import synth_state_dict
        params = t2v_params()
        ld = LatentDiffusionHost({"params": params}, conditioner=SyntheticConditioner(77, params["context_dim"]))
        sd = synth_state_dict(param_shapes(params), 0)
        ld.model.diffusion_model.load_state_dict(sd, strict=True)
        ld.model.diffusion_model.prepare(d)
        _HOST.update(ld=ld, params=params, sd=sd)
    return _HOST["ld"], _HOST["params"], _HOST["sd"]


# Config 1's 4-step schedule (t = 999, 666, 333, 0).  Its first update (999 -> 666: x_prev = 6.2 x - 3.8 e_t) multiplies the guided-eps
# error; with single fp16 matrix-core operands no residual mode stays inside 1e-3 there (measured, rounds 3-4: x_prev 3.68e-3 fast /
# 2.72e-3 default / 2.05e-3 strict; pred_x0 4.43e-3 / 3.27e-3 / 2.47e-3).  Round 5: the pipelines' operand policy ("auto", the default)
# evaluates those steps in the WIDE operand mode (fp32 storage, split-fp16 products; csrc/wide.hip), and every step of config 1 is
# asserted at the north star on x_prev AND pred_x0, teacher-forced and free-running.
NORTH_STAR = 1e-3
# REGRESSION GUARD of the non-default policy "f16" (single fp16 operands at every step), <= 1.25x measured: not a parity claim
F16_OPERANDS_FIRST_STEP_GUARD = {"float16": dict(e_t=1.25e-2, x_prev=4.6e-3, pred_x0=5.6e-3), "outer": dict(e_t=9.3e-3, x_prev=3.4e-3, pred_x0=4.1e-3),
                                 "float32": dict(e_t=7.0e-3, x_prev=2.6e-3, pred_x0=3.1e-3)}
E_T_F16_STEPS = {"float16": 9.0e-3, "outer": 4.9e-3, "float32": 4.1e-3}      # guided e_t of the steps that run on fp16 operands (reported quantity; 1.25x measured)


@pytest.mark.parametrize("residual,policy", [("float16", "auto"), ("outer", "auto"), ("float32", "auto"), ("outer", "f16")])
@pytest.mark.parametrize("latent_dtype", [torch.float16, torch.float32])
def test_cfg1_full_size_basic_sample_vs_reference_golden(latent_dtype, residual, policy):
    """Config 1: 4 DDIM steps of the real UNet on one 512x320x16f tile, CFG 7.5, against per-step vectors of the reference itself
    (make_golden.py g17).  Teacher-forced (every step starts from the REFERENCE's latent of that step: the error of one step in
    isolation, at schedule indices 3, 2, 1, 0) and free-running (basic_sample end to end: errors compound), in every residual mode.
    Default operand policy: the steps whose update would carry the guided-eps error past 1e-3 (indices 3, 2, 1) run wide, and
    EVERY step is inside 1e-3 on x_prev and pred_x0.  policy "f16" (single fp16 operands throughout): regression guard at the
    measured values."""
    from dynamicscaler_amd import ops
    from dynamicscaler_amd.pipelines import VC2_Pipeline_T2V
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    d = dev()
    z = np.load(os.path.join(G, "cfg1_full_t2v.npz"))
    ld, params, _ = full_host(d)
    _set_mode(ld.model.diffusion_model, residual)
    try:
        cfgd = {"params": {"unet_config": {"params": params}}}
        sched = lvdm_DDIM_Scheduler(ld)
        pipe = VC2_Pipeline_T2V(ld, sched, cfgd).to(d, latent_dtype)
        pipe.operand_policy = policy
        sched.make_schedule(4, verbose=False)
        timesteps = np.flip(sched.ddim_timesteps)
        assert list(timesteps) == list(z["timesteps"])
        cond = ld.get_learned_conditioning(["a prompt"])
        uncond = ld.get_learned_conditioning([""])
        g = float(z["guidance"])
        assert pipe.wide_steps_of(4, g) == ([3, 2, 1] if policy == "auto" else [])
        name = str(latent_dtype).split(".")[1]
        fp16_lat = 1.0 if latent_dtype == torch.float32 else 1.1       # fp16 latents add the stored tile's own rounding (guard only)
        guard = F16_OPERANDS_FIRST_STEP_GUARD[residual]
        # ---- teacher-forced, step by step ----
        for i, t in enumerate(timesteps):
            x_ref = T(z["x_init"]) if i == 0 else T(z[f"x_prev_{i - 1}"])
            x = x_ref.to(d, latent_dtype)
            index = int(z[f"index_{i}"])
            wide = pipe._begin_step(i, index, g) == "wide"
            eps = pipe._eps(torch.cat([x, x], 0), t, [cond, uncond], int(z["fps"]), 16, cfg_pairs=1, clean_cond=True)
            e_t = eps[1:] + g * (eps[:1] - eps[1:])                      # test-side arithmetic, fp32
            xp, x0 = ops.cfg_ddim(x, eps[:1].contiguous(), eps[1:].contiguous(), (1, 4, 16, 40, 64), g,
                                  sched.step_coefficients(index))
            r = dict(test="cfg1_teacher_forced", residual=residual, policy=policy, wide=wide, latents=name, step=i, t=int(t),
                     e_t=relerr(e_t, T(z[f"e_t_{i}"])), x_prev=relerr(xp, T(z[f"x_prev_{i}"])), pred_x0=relerr(x0, T(z[f"pred_x0_{i}"])))
            print(r)
            record(**r)
            if policy == "f16" and i == 0:
                assert r["e_t"] < guard["e_t"] and r["x_prev"] < guard["x_prev"] * fp16_lat and r["pred_x0"] < guard["pred_x0"] * fp16_lat, r
            else:           # the north-star tolerance itself, on x_prev AND pred_x0
                assert r["x_prev"] < NORTH_STAR and r["pred_x0"] < NORTH_STAR, r
                # (fp16 latents: the tile handed to the UNet is itself rounded, 2^-11 relative, which eps inherits)
                assert r["e_t"] < ((1e-4 if latent_dtype == torch.float32 else 5e-4) if wide else E_T_F16_STEPS[residual]), r
        # ---- free-running: the pipeline's own loop from the same init latent ----
        lat = T(z["x_init"]).to(d, latent_dtype)
        for i, t in enumerate(timesteps):
            lat, den = pipe._basic_denoise_one_step(lat, t, i, 4, cond, uncond, g, int(z["fps"]), 16, {})
            r = dict(test="cfg1_free_running", residual=residual, policy=policy, latents=name, step=i, x_prev=relerr(lat, T(z[f"x_prev_{i}"])),
                     pred_x0=relerr(den, T(z[f"pred_x0_{i}"])))
            print(r)
            record(**r)
            if policy == "f16":            # the first step's error is carried along
                assert r["x_prev"] < guard["x_prev"] * fp16_lat and r["pred_x0"] < guard["pred_x0"] * fp16_lat, r
            else:
                assert r["x_prev"] < NORTH_STAR and r["pred_x0"] < NORTH_STAR, r
        if policy == "auto":
            assert pipe.wide_steps_run == [(0, 3), (1, 2), (2, 1)]
        # basic_sample itself (the drop-in entry point) returns the same thing bit for bit
        _, den2 = pipe.basic_sample(prompt="a prompt", height=320, width=512, frames=16, fps=int(z["fps"]), guidance_scale=g,
                                    num_inference_steps=4, output_type="latent", latents=T(z["x_init"]))
    finally:
        _reset_mode(ld.model.diffusion_model)
    assert torch.equal(den2, den)
    assert relerr(den2, T(z["denoised"])) < (guard["pred_x0"] * fp16_lat if policy == "f16" else NORTH_STAR)


def test_error_budget_layerwise_full_size():
    """Activations after every block of the HIP UNet against the oracle (fp32 CPU) on the full-size tile: the
    accumulated rel-L2 distance along the depth of the network, written to gpurun_out/error_budget_full.json.
    Asserts only that the distance grows smoothly (no single block multiplies it by more than 3x from a floor of
    2e-4) and ends below EPS tolerance -- the table itself is the deliverable."""
    from oracle.unet import unet_forward
    from dynamicscaler_amd.synth import synth_normal
    d = dev()
    ld, params, sd = full_host(d)
    m = ld.model.diffusion_model
    z = np.load(os.path.join(G, "unet_full_t2v.npz"))
    x = T(z["x"])
    ctx = synth_normal((1, 77, 1024), 1)
    t = torch.tensor([int(z["t"])])
    got = {}

    def tap_gpu(name, rows, geo):
        got[name] = rows.float().cpu()

    m._tap = tap_gpu
    try:
        eps = m(x.to(d, torch.float16), t.to(d), context=ctx.to(d), fps=int(z["fps"]))
    finally:
        m._tap = None
    table = []

    def tap_ref(name, h):
        ref = h.permute(0, 2, 3, 1).reshape(-1, h.shape[1])
        table.append({"block": name, "rel_l2": relerr(got.pop(name), ref), "rows": ref.shape[0], "channels": ref.shape[1],
                      "ref_rms": float(ref.pow(2).mean().sqrt())})

    ref_eps = unet_forward(sd, params, x, t, ctx, fps=int(z["fps"]), tap=tap_ref)
    assert not got, f"blocks without an oracle counterpart: {sorted(got)}"
    table.append({"block": "out (eps)", "rel_l2": relerr(eps, ref_eps)})
    for row in table:
        print(f"{row['block']:28s} {row['rel_l2']:.3e}")
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    json.dump(table, open(os.path.join(REPO, "gpurun_out", "error_budget_full.json"), "w"), indent=1)
    record(test="error_budget_full", eps=table[-1]["rel_l2"], worst_block=max(table, key=lambda r: r["rel_l2"])["block"])
    prev = 2e-4
    for row in table:
        assert row["rel_l2"] < 3 * max(prev, 2e-4) + 1e-3, f"error jumps at {row['block']}: {prev:.2e} -> {row['rel_l2']:.2e}"
        prev = row["rel_l2"]
    assert max(r["rel_l2"] for r in table) < 5.0e-3      # measured: peaks at 2.5e-3 (output_blocks.2.1)
    assert table[-1]["rel_l2"] < 3.4e-3                   # measured 1.66e-3


@pytest.mark.parametrize("size", ["toy", "full"])
def test_no_kernel_reads_uninitialised_cu_state(size):
    """Every launch of the UNet program preceded by ds_dbg_poison_cu_state (NaN patterns in all LDS and in the vector /
    accumulator register files of every CU): the outputs must be bit-identical to the clean run.  State a kernel reads
    before writing would otherwise depend on what ran on the CU before -- i.e. on timing once two hipGraphs replay
    concurrently (the round-1 hazard, profiles/r1_notes.md)."""
    from dynamicscaler_amd import _lib
    from dynamicscaler_amd.synth import synth_normal, synth_state_dict
    from dynamicscaler_amd.unet import UNetModel
    from dynamicscaler_amd.unet_spec import param_shapes
    d = dev()
    lib = _lib.load()
    if size == "toy":
        zt = np.load(os.path.join(G, "unet_tiny_t2v.npz"))
        params = json.loads(bytes(zt["params_json"]).decode())
        m = UNetModel(**params)
        m.load_state_dict(synth_state_dict(param_shapes(params), 5), strict=True)
        m = m.to(d).eval()
        m.prepare(d)
        cases = [((2, 4, 4, 8, 16), 64, 2), ((1, 4, 6, 8, 8), 64, None), ((1, 4, 24, 8, 8), 64, None)]
    else:
        ld, params, _ = full_host(d)
        m = ld.model.diffusion_model
        cases = [((1, 4, 16, 40, 64), 1024, 1)]

    diag = _lib.load_diag()                  # the poison launch lives outside the product library (csrc/diag.hip)
    calls = [0]

    def poison_in_front(phase, kernel, flops, info):
        # ds_unet_set_hooks: the C launch program calls back on the host right before it enqueues each kernel-family call
        if phase == 0:
            assert diag.ds_dbg_poison_cu_state(torch.cuda.current_stream().cuda_stream) == 0
            calls[0] += 1

    for shape, cdim, pairs in cases:
        tiles = synth_normal(shape, 100).to(d, torch.float16)
        n = shape[0]
        if pairs:
            x = torch.cat([tiles, tiles], 0)
            ctx = torch.cat([synth_normal((1, 77, cdim), 61)] * n + [synth_normal((1, 77, cdim), 62)] * n, 0).to(d)
            kw = dict(cfg_pairs=n)
        else:
            x, ctx, kw = tiles, synth_normal((n, 77, cdim), 61).to(d), {}
        ts = torch.full((x.shape[0],), 500, device=d, dtype=torch.long)
        clean = m(x, ts, context=ctx, fps=8, **kw).clone()
        calls[0] = 0
        m.launch_hook = poison_in_front            # the same ds_unet_forward, the poison launch in front of every kernel
        try:
            dirty = m(x, ts, context=ctx, fps=8, **kw).clone()
            torch.cuda.synchronize()
        finally:
            m.launch_hook = None
        assert calls[0] > 100
        assert bool(torch.isfinite(clean).all())
        assert torch.equal(clean, dirty), f"{size} {shape}: {int((clean != dirty).sum())} elements changed by the poison run " \
                                          f"(nan: {bool(torch.isnan(dirty).any())})"


T24_TOL = {"tiny": 1.0e-2, "full": 1.0e-2}   # tightened to <= 2x measured below once measured


@pytest.mark.parametrize("size", ["tiny", "full"])
def test_unet_t24_vs_reference_golden(size):
    """BASELINE config 5 runs the UNet at T = 24 (temporal attention over 24 tokens: the two-block MFMA kernel; joint-T
    GroupNorm and the (3,1,1) convolutions over 24 frames): one forward against the reference's own
    (tests/golden/unet_t24.npz, make_golden.py g18)."""
    from dynamicscaler_amd.synth import synth_normal, synth_state_dict
    from dynamicscaler_amd.unet import UNetModel
    from dynamicscaler_amd.unet_spec import param_shapes
    d = dev()
    z = np.load(os.path.join(G, "unet_t24.npz"))
    if size == "tiny":
        params = json.loads(bytes(z["tiny_params_json"]).decode())
        m = UNetModel(**params)
        m.load_state_dict(synth_state_dict(param_shapes(params), 5), strict=True)
        m = m.to(d).eval()
        ctx = T(z["tiny_ctx"])
    else:
        ld, params, _ = full_host(d)
        m = ld.model.diffusion_model
        ctx = synth_normal((1, 77, 1024), 1)
    x, t, fps = T(z[f"{size}_x"]), int(z[f"{size}_t"]), int(z[f"{size}_fps"])
    assert x.shape[2] == 24
    eps = m(x.to(d, torch.float16), torch.tensor([t], device=d), context=ctx.to(d), fps=fps)
    e = relerr(eps, T(z[f"{size}_eps"]))
    r = dict(test="unet_t24", size=size, eps=e)
    print(r)
    record(**r)
    assert eps.shape == x.shape and e < T24_TOL[size]


@pytest.mark.parametrize("residual", ["float16", "float32"])
def test_full_size_batch_equals_separate_forwards(residual):
    """What rank sharding rests on, at the metric's tile size: a batch of 16 UNet evaluations (one GPU's [cond | uncond] batch of 8 tiles)
    gives, item for item, the bits of batches of 8, 2 and 1 (an 8-GPU rank's share).  The toy-size form of this test
    (test_gpu_unet.py) cannot see a kernel form chosen from the launch's instance count -- the full-size per-frame GroupNorms of
    levels 3-4 (160 / 40 rows, 16 x batch instances) crossed such a threshold between batch 2 and batch 4 until round 3."""
    from dynamicscaler_amd.synth import synth_normal
    d = dev()
    ld, params, _ = full_host(d)
    m = ld.model.diffusion_model
    old = (m.residual_dtype, m.residual_scope)
    try:
        m.residual_dtype, m.residual_scope = (torch.float32 if residual == "float32" else torch.float16), "full"
        m.prepare(d)
        x = synth_normal((16, 4, 16, 40, 64), 300).to(d, torch.float16)
        ctx = torch.cat([synth_normal((1, 77, 1024), 310 + i) for i in range(16)], 0).to(d)
        ts = torch.tensor([999, 999, 500, 500, 20, 20, 0, 0] * 2, device=d)
        with torch.no_grad():
            whole = m(x, ts, context=ctx, fps=16).float()
            for lo, hi in ((0, 1), (2, 4), (4, 5), (6, 8), (8, 16)):
                part = m(x[lo:hi], ts[lo:hi], context=ctx[lo:hi], fps=16).float()
                assert torch.equal(part, whole[lo:hi]), (residual, lo, hi, relerr(part, whole[lo:hi]))
    finally:
        m.residual_dtype, m.residual_scope = old
        m.prepare(d)


@pytest.mark.parametrize("size", ["toy", "full"])
def test_concurrent_graph_replays_repeatable(size):
    """Two hipGraphs of the batched UNet evaluation replaying CONCURRENTLY on two streams (bench.py's default mode) with
    random relative delays: every replay must reproduce the serial replay bit for bit -- at the toy size (4-stage LDS-DMA
    tiles, the ones that raced in round 1's withdrawn build) with every BLOCK's output compared (block taps of the C program, captured
    into the graph as copy nodes, so every replay refreshes them), and at full size (the
    256-row one-workgroup-per-CU tiles) on the final eps.  The withdrawn build fails this in about 1 round of 20
    (gpurun_out/s3, profiles/r2_notes.md); the deterministic form of the same check is the ISA test in test_host_cpu.py."""
    from dynamicscaler_amd import ops
    from dynamicscaler_amd.synth import synth_normal, synth_state_dict
    from dynamicscaler_amd.unet import UNetModel
    from dynamicscaler_amd.unet_spec import param_shapes
    d = dev()
    if size == "toy":
        zt = np.load(os.path.join(G, "unet_tiny_t2v.npz"))
        params = json.loads(bytes(zt["params_json"]).decode())
        m = UNetModel(**params)
        m.load_state_dict(synth_state_dict(param_shapes(params), 5), strict=True)
        m = m.to(d).eval()
        m.prepare(d)
        shape, cdim, n, rounds = (2, 4, 4, 8, 16), 64, 2, 120
    else:
        ld, params, _ = full_host(d)
        m = ld.model.diffusion_model
        shape, cdim, n, rounds = (1, 4, 16, 40, 64), 1024, 1, 12
    log = []
    keep = size == "toy"

    def keep_block(name, rows, geo):          # UNetModel._tap: a copy of every block's output rows (ds_unet_set_hooks)
        log.append((f"{len(log)}:{name} {tuple(rows.shape)}", rows))

    streams = [torch.cuda.Stream(d), torch.cuda.Stream(d)]
    graphs = []
    try:
        m._tap = keep_block if keep else None
        for slot in range(2):
            tiles = synth_normal(shape, 100 + slot).to(d, torch.float16)
            x = torch.cat([tiles, tiles], 0)
            ctx = torch.cat([synth_normal((1, 77, cdim), 61)] * n + [synth_normal((1, 77, cdim), 62)] * n, 0).to(d)
            ts = torch.full((2 * n,), 500 + slot, device=d, dtype=torch.long)
            with torch.cuda.stream(streams[slot]):
                m(x, ts, context=ctx, fps=8, cfg_pairs=n)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            del log[:]
            with torch.cuda.graph(g):
                out = m(x, ts, context=ctx, fps=8, cfg_pairs=n)
            graphs.append((g, list(log) + [("final:eps", out)], (x, ctx, ts)))
    finally:
        m._tap = None
    assert not keep or all(len(kept) > 10 for _, kept, _ in graphs)
    refs = []
    for slot, (g, kept, _in) in enumerate(graphs):
        with torch.cuda.stream(streams[slot]):
            g.replay()
        torch.cuda.synchronize()
        refs.append([t.clone() for _, t in kept])
    gen = torch.Generator().manual_seed(5)
    for r in range(rounds):
        reps = 1 + int(torch.randint(0, 3, (1,), generator=gen))
        delay = int(torch.randint(0, 200000, (1,), generator=gen))
        for slot in ((0, 1) if r % 2 == 0 else (1, 0)):
            with torch.cuda.stream(streams[slot]):
                if slot == r % 2:
                    torch.cuda._sleep(delay)
                for _ in range(reps):
                    graphs[slot][0].replay()
        torch.cuda.synchronize()
        for slot, (g, kept, _in) in enumerate(graphs):
            bad = [nm for (nm, t), rf in zip(kept, refs[slot]) if not torch.equal(t, rf)]
            assert not bad, f"{size}: round {r}, slot {slot}: first diverging block {bad[0]} ({len(bad)} of {len(kept)} outputs)"


# 4-step schedule: with single fp16 operands its first update (999 -> 666) put the panorama at 3.76e-3 / 2.76e-3 / 2.09e-3 (rounds 3-4).
# The default operand policy runs steps 0-2 wide: asserted at the north star.  policy "f16": regression guard, <= 1.25x measured.
RING_F16_OPERANDS_GUARD = {"outer": 3.45e-3}


@pytest.mark.parametrize("residual,policy", [("float16", "auto"), ("outer", "auto"), ("float32", "auto"), ("outer", "f16")])
def test_ring_pipeline_with_the_real_unet_vs_reference_golden(residual, policy):
    """P2 end to end with the REAL t2v UNet (1.41 B parameters): the reference's own
    VC2_Pipeline_T2V_SpherePano.basic_sample_shift_multi_windows (pipeline/t2v_sphere_panorama_pipeline.py:316-660) on a
    1024x512x16f ring panorama -- 2x2 shifted windows, 40 % H overlap re-noised under the mask, the grid shifting across both
    seams every step, 4 DDIM steps (t = 999, 666, 333, 0), CFG 7.5; 32 CPU forwards, make_golden.py g25 -- against the HIP tile
    engine + ds_unet_forward with the same seed (host RNG in the reference's draw order).  The 4-step schedule's first updates
    multiply the guided-eps error like in config 1: the default operand policy evaluates steps 0-2 in the wide mode (one window
    per evaluation) and the final pred-x0 panorama -- what the loop returns, the product of all four updates -- is inside 1e-3
    (4.5e-6 measured; 2.76e-3 with fp16 operands throughout)."""
    from dynamicscaler_amd.pipelines import VC2_Pipeline_T2V_SpherePano
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    path = os.path.join(G, "ring_real_unet.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/ring_real_unet.npz not generated (make_golden.py --full --only g25)")
    d = dev()
    z = np.load(path)
    rec = json.load(open(os.path.join(G, "ring_real_unet_trace.json")))
    ld, params, _ = full_host(d)
    unet = ld.model.diffusion_model
    _set_mode(unet, residual)
    try:
        pipe = VC2_Pipeline_T2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld, rng_mode="reference"),
                                           {"params": {"unet_config": {"params": params}}}).to(d, torch.float32)
        pipe.operand_policy = policy
        trace = []
        torch.manual_seed(2333333)
        _, den = pipe.basic_sample_shift_multi_windows(prompt="a prompt", fps=int(z["fps"]), guidance_scale=float(z["guidance"]),
                                                       output_type="latent", init_panorama_latent=T(z["init"]),
                                                       step_callback=lambda i, t, w, p, p0: trace.append((i, int(t), [list(x) for x in w])),
                                                       **rec["geom"])
    finally:
        _reset_mode(unet)
    for (i, t, wins), ref in zip(trace, rec["trace"]):
        assert i == ref["i"] and t == ref["t"] and wins == ref["windows"], (i, wins, ref)
    assert pipe.wide_steps_run == ([(0, 3), (1, 2), (2, 1)] if policy == "auto" else [])
    e = relerr(den, T(z["denoised"]))
    r = dict(test="ring_real_unet", residual=residual, policy=policy, denoised=e)
    print(r)
    record(**r)
    tol = NORTH_STAR if policy == "auto" else RING_F16_OPERANDS_GUARD[residual]
    assert tuple(den.shape) == tuple(z["denoised"].shape) and e < tol, r


def test_full_size_panorama_independent_of_the_execution_mode():
    """The same ring panorama (real UNet, the golden's 1024x512x16f geometry, 4 steps, CFG 7.5) under the execution modes a rank
    can find itself in -- one [cond | uncond] batch of all windows of a level on one stream (1 GPU), one window per batch on two
    streams with hipGraph replay, cond / uncond of a single window on two streams (an 8-GPU rank's share) -- must be ONE
    panorama, bit for bit: what lets `bench.py --gpus N` claim the single-GPU result for every N."""
    from dynamicscaler_amd.pipelines import VC2_Pipeline_T2V_SpherePano
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    path = os.path.join(G, "ring_real_unet.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/ring_real_unet.npz not generated (make_golden.py --full --only g25)")
    d = dev()
    z = np.load(path)
    rec = json.load(open(os.path.join(G, "ring_real_unet_trace.json")))
    ld, params, _ = full_host(d)
    outs = {}
    for name, tb, streams, graph, split in (("batched", 8, 1, False, 0), ("tb1_two_streams_graph", 1, 2, True, 0),
                                            ("tb1_cfg_split", 1, 2, True, 2), ("batched_two_streams_graph", 8, 2, True, 1)):
        pipe = VC2_Pipeline_T2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld, rng_mode="reference"),
                                           {"params": {"unet_config": {"params": params}}}).to(d, torch.float32)
        pipe.max_tile_batch, pipe.num_streams, pipe.use_graph, pipe.split_cfg_over_streams = tb, streams, graph, split
        torch.manual_seed(2333333)
        _, den = pipe.basic_sample_shift_multi_windows(prompt="a prompt", fps=int(z["fps"]), guidance_scale=float(z["guidance"]),
                                                       output_type="latent", init_panorama_latent=T(z["init"]), **rec["geom"])
        outs[name] = den.float().cpu()
    ref = outs.pop("batched")
    for name, den in outs.items():
        assert torch.equal(den, ref), (name, relerr(den, ref))


# 4-step schedule with single fp16 operands: 3.46e-3 / 1.93e-3 measured (fast / strict); the default policy runs steps 0-2 wide
I2V_RING_F16_OPERANDS_GUARD = {"float32": 2.4e-3}


@pytest.mark.parametrize("residual,policy", [("float16", "auto"), ("outer", "auto"), ("float32", "f16")])
def test_i2v_ring_pipeline_with_the_real_unet_vs_reference_golden(residual, policy):
    """P3 end to end with the REAL i2v UNet (1.44 B parameters, image cross-attention): the reference's
    VC2_Pipeline_I2V_SpherePano.basic_sample_shift_multi_windows (pipeline/i2v_sphere_panorama_pipeline.py:564-996) on a
    1024x512x16f ring panorama -- 2x2 shifted windows, per-window crops of the panorama image -> 16 image tokens next to the 77
    text tokens, merge-prev, 4 DDIM steps, CFG 7.5; 32 CPU forwards, make_golden.py g27 -- against the HIP pipeline with the same
    seed.  Like config 1, the 4-step schedule's first updates would dominate the distance: the default operand policy runs them
    in the wide mode (the i2v UNet's twin: image-token branch included) and the result is inside 1e-3."""
    import yaml
    from helpers import synth_image_embedder
    from dynamicscaler_amd.host_model import LatentDiffusionHost, SyntheticConditioner
    from dynamicscaler_amd.pipelines_i2v import VC2_Pipeline_I2V_SpherePano
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    from dynamicscaler_amd.unet_spec import param_shapes
    from dynamicscaler_amd.synth import synth_state_dict, synth_normal
    path = os.path.join(G, "i2v_ring_real_unet.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/i2v_ring_real_unet.npz not generated (make_golden.py --full --only g27)")
    d = dev()
    z = np.load(path)
    rec = json.load(open(os.path.join(G, "i2v_ring_real_unet_trace.json")))
    if "i2v" not in _HOST:
        params = yaml.safe_load(open(os.path.join(REPO, "dynamicscaler_amd", "configs", "i2v_512_v1_unet.yaml")))
        ld = LatentDiffusionHost({"params": params}, conditioner=SyntheticConditioner(77, params["context_dim"], cond_seed=11, uncond_seed=12))
        ld.model.diffusion_model.load_state_dict(synth_state_dict(param_shapes(params), 3), strict=True)
        ld.get_image_embeds = synth_image_embedder(params["context_dim"])
        ld.embedder = object()
        ld = ld.to(d)
        _HOST["i2v"] = (ld, params)
    ld, params = _HOST["i2v"]
    unet = ld.model.diffusion_model
    _set_mode(unet, residual)
    pano_img = synth_normal((3, 512, 1024), int(z["pano_img_seed"])).clamp(-1, 1)
    try:
        pipe = VC2_Pipeline_I2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld, rng_mode="reference"),
                                           {"params": {"unet_config": {"params": params}}}).to(d, torch.float32)
        pipe.operand_policy = policy
        trace = []
        torch.manual_seed(2333333)
        _, den = pipe.basic_sample_shift_multi_windows(prompt="a prompt", fps=int(z["fps"]), guidance_scale=float(z["guidance"]),
                                                       output_type="latent", init_panorama_latent=T(z["init"]), pano_image_tensor=pano_img,
                                                       step_callback=lambda i, t, w, p, p0: trace.append((i, int(t), [list(x) for x in w])),
                                                       **rec["geom"])
    finally:
        _reset_mode(unet)
    for (i, t, wins), ref in zip(trace, rec["trace"]):
        assert i == ref["i"] and t == ref["t"] and wins == ref["windows"], (i, wins, ref)
    assert pipe.wide_steps_run == ([(0, 3), (1, 2), (2, 1)] if policy == "auto" else [])
    e = relerr(den, T(z["denoised"]))
    r = dict(test="i2v_ring_real_unet", residual=residual, policy=policy, denoised=e)
    print(r)
    record(**r)
    tol = NORTH_STAR if policy == "auto" else I2V_RING_F16_OPERANDS_GUARD[residual]
    assert tuple(den.shape) == tuple(z["denoised"].shape) and e < tol, r


# ---- the ring loops with the REAL UNets on the schedule the metric runs (50 DDIM steps), first and last six steps ----------------
# Asserted at the north star's 1e-3 on the panorama latent after EVERY recorded step (make_golden.py g28 / g29 ran the reference
# itself: 2 x 96 CPU forwards).  Measured values: gpurun_out/measured_parity.jsonl -> profiles/r4_measured_parity.jsonl.
RING50_TOL = 1e-3


class _Stop(Exception):
    pass


def _i2v_host(d):
    import yaml
    from helpers import synth_image_embedder
    from dynamicscaler_amd.host_model import LatentDiffusionHost, SyntheticConditioner
    from dynamicscaler_amd.unet_spec import param_shapes
    from dynamicscaler_amd.synth import synth_state_dict
    if "i2v" not in _HOST:
        params = yaml.safe_load(open(os.path.join(REPO, "dynamicscaler_amd", "configs", "i2v_512_v1_unet.yaml")))
        ld = LatentDiffusionHost({"params": params}, conditioner=SyntheticConditioner(77, params["context_dim"], cond_seed=11, uncond_seed=12))
        ld.model.diffusion_model.load_state_dict(synth_state_dict(param_shapes(params), 3), strict=True)
        ld.get_image_embeds = synth_image_embedder(params["context_dim"])
        ld.embedder = object()
        ld = ld.to(d)
        _HOST["i2v"] = (ld, params)
    return _HOST["i2v"]


def _set_mode(unet, residual):
    unet.residual_dtype, unet.residual_scope = {"float16": (torch.float16, "full"), "float32": (torch.float32, "full"),
                                                "outer": (torch.float32, "outer")}[residual]


def _reset_mode(unet):
    """Back to the LIBRARY DEFAULT (f32outer): the shared hosts are never left in a non-default mode for the tests that follow."""
    _set_mode(unet, "outer")


@pytest.mark.parametrize("residual", ["float16", "outer", "float32"])
@pytest.mark.parametrize("model", ["t2v", "i2v"])
def test_ring_loops_real_unet_on_the_50_step_schedule_vs_reference(model, residual):
    """What bench.py times, against the reference: the overlapped-ring loop (gather -> mask-gated re-noise at the 50-step sigmas ->
    2 x UNet -> CFG -> DDIM -> scatter; pipeline/t2v_sphere_panorama_pipeline.py:481-634, re-noise :550-559; the i2v loop
    pipeline/i2v_sphere_panorama_pipeline.py:777-970 with per-window image tokens and merge-prev) with the REAL UNet on a
    1024x512x16f panorama, 2x2 shifted windows, CFG 7.5 -- steps 0..5 of the 50-step schedule (t = 999 .. 897, where the
    guided-eps error weighs most) and, through the method's own use_skip_time, its last six steps.  The panorama latent after
    every recorded step and the final pred-x0 panorama: 1e-3, every residual mode."""
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    name = "ring_real_unet_50step" if model == "t2v" else "i2v_ring_real_unet_50step"
    path = os.path.join(G, name + ".npz")
    if not os.path.exists(path):
        pytest.skip(f"tests/golden/{name}.npz not generated (make_golden.py --full --only {'g28' if model == 't2v' else 'g29'})")
    d = dev()
    z = np.load(path)
    rec = json.load(open(os.path.join(G, name + "_trace.json")))
    nrec = int(z["steps"])
    if model == "t2v":
        from dynamicscaler_amd.pipelines import VC2_Pipeline_T2V_SpherePano as Pipe
        ld, params, _ = full_host(d)
        extra = {}
    else:
        from dynamicscaler_amd.pipelines_i2v import VC2_Pipeline_I2V_SpherePano as Pipe
        from dynamicscaler_amd.synth import synth_normal
        ld, params = _i2v_host(d)
        extra = dict(pano_image_tensor=synth_normal((3, 512, 1024), int(z["pano_img_seed"])).clamp(-1, 1))
    unet = ld.model.diffusion_model
    _set_mode(unet, residual)
    out, stats_all = {}, {}
    try:
        for end in ("first", "last"):
            geom = rec["geom"] if model == "t2v" else rec["geoms"][end]
            pipe = Pipe(ld, lvdm_DDIM_Scheduler(ld, rng_mode="reference"), {"params": {"unet_config": {"params": params}}}).to(d, torch.float32)
            snaps, trace = [], []

            def cb(i, t, wins, pano, pano_x0):
                trace.append((i, int(t), [list(x) for x in wins]))
                snaps.append((pano.float().cpu().clone(), pano_x0.float().cpu().clone()))
                if end == "first" and len(snaps) == nrec:
                    raise _Stop()

            kw = dict(prompt="a prompt", fps=int(z["fps"]), guidance_scale=float(z["guidance"]), output_type="latent",
                      init_panorama_latent=T(z[f"{end}_init"]), step_callback=cb, **geom, **extra)
            if end == "last":
                kw.update(use_skip_time=True, skip_time_step_idx=50 - nrec)
            torch.manual_seed(2333333)
            try:
                _, den = pipe.basic_sample_shift_multi_windows(**kw)
            except _Stop:
                den = None
            assert len(snaps) == nrec and (den is None) == (end == "first")
            for (i, t, wins), ref in zip(trace, rec["traces"][end]):
                assert i == ref["i"] and t == ref["t"] and wins == ref["windows"], (end, i, wins, ref)
            errs = {k: relerr(snaps[k][0], T(z[f"{end}_pano_{k}"])) for k in range(nrec) if f"{end}_pano_{k}" in z.files}
            errs["x0"] = relerr(snaps[-1][1], T(z[f"{end}_x0_{nrec - 1}"]))
            if den is not None:
                assert torch.equal(den.float().cpu(), snaps[-1][1])
            out[end] = errs
            stats_all[end] = {str(k): loop_err_stats(snaps[k][0], T(z[f"{end}_pano_{k}"]), trace[k][2]) for k in range(nrec) if f"{end}_pano_{k}" in z.files}
    finally:
        _reset_mode(unet)
    r = dict(test="ring50_real_unet", model=model, residual=residual, first={str(k): v for k, v in out["first"].items()},
             last={str(k): v for k, v in out["last"].items()}, stats=stats_all)
    print(r)
    record(**r)
    for end in ("first", "last"):
        for k, e in out[end].items():
            if end == "first" and k == "x0":
                continue        # pred_x0 at t ~ 900 amplifies the eps error by sqrt((1-a)/a) ~ 10; it never leaves the loop there
            assert e < RING50_TOL, (end, k, e, r)


def test_ring_loop_real_unet_mid_schedule_on_the_headline_window_grid_vs_reference():
    """The middle of the schedule on the headline geometry's grid (make_golden.py g31): the reference's t2v ring loop with the REAL UNet on
    two columns x two rows of BASELINE config 3's window grid (512x320 windows, loop_step = 8 like cfg3: 1/8 window per step, so from
    the second recorded step on the right-hand column's windows straddle the W seam), entered through the method's own use_skip_time
    at step 20 of 50 (schedule indices 29..24, t = 591 .. 489).  Panorama latent after steps 0 / 2 / 5: 1e-3, library default mode (no wide
    step on this schedule); the intermediate pred-x0 panorama after step 5 is reported.  Closes the gap between the first and the last six steps."""
    from dynamicscaler_amd.pipelines import VC2_Pipeline_T2V_SpherePano
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    path = os.path.join(G, "ring_real_unet_50step_mid.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/ring_real_unet_50step_mid.npz not generated (make_golden.py --full --only g31)")
    d = dev()
    z = np.load(path)
    rec = json.load(open(os.path.join(G, "ring_real_unet_50step_mid_trace.json")))
    nrec, skip = int(z["steps"]), int(z["skip"])
    ld, params, _ = full_host(d)
    unet = ld.model.diffusion_model
    _reset_mode(unet)                    # the library default
    pipe = VC2_Pipeline_T2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld, rng_mode="reference"), {"params": {"unet_config": {"params": params}}}).to(d, torch.float32)
    snaps, trace = [], []

    def cb(i, t, wins, pano, pano_x0):
        trace.append((i, int(t), [list(x) for x in wins]))
        snaps.append((pano.float().cpu().clone(), pano_x0.float().cpu().clone()))
        if len(snaps) == nrec:
            raise _Stop()

    torch.manual_seed(2333333)
    try:
        pipe.basic_sample_shift_multi_windows(prompt="a prompt", fps=int(z["fps"]), guidance_scale=float(z["guidance"]), output_type="latent",
                                              init_panorama_latent=T(z["init"]).float(), step_callback=cb, use_skip_time=True,
                                              skip_time_step_idx=skip, **rec["geom"])
    except _Stop:
        pass
    assert len(snaps) == nrec and pipe.wide_steps_run == []
    for (i, t, wins), ref in zip(trace, rec["trace"]):
        assert i == ref["i"] and t == ref["t"] and wins == ref["windows"], (i, t, wins, ref)
    W = rec["geom"]["total_w"] // 8
    assert any(w[1] > W for _, _, wins in trace[1:] for w in wins), "no window of the recorded steps crosses the W seam"
    errs = {k: relerr(snaps[k][0], T(z[f"pano_{k}"])) for k in range(nrec) if f"pano_{k}" in z.files}
    errs["x0"] = relerr(snaps[-1][1], T(z[f"x0_{nrec - 1}"]))
    r = dict(test="ring50_mid_real_unet", residual="outer", errs={str(k): v for k, v in errs.items()},
             stats={str(k): loop_err_stats(snaps[k][0], T(z[f"pano_{k}"]), trace[k][2]) for k in range(nrec) if f"pano_{k}" in z.files})
    print(r)
    record(**r)
    # the LATENT panorama -- what the loop carries forward -- at the north star after every recorded step.  The pred-x0 panorama at
    # t = 489 is (x - 0.82 e_t) / 0.57: it multiplies the guided-eps error by 1.4 and is overwritten by every later step (only the last
    # step's pred-x0 leaves the loop: asserted at 1e-3 by the "last six steps" test above); reported, with a regression guard
    assert len(errs) == 4 and all(e < RING50_TOL for k, e in errs.items() if k != "x0"), r
    assert errs["x0"] < 2.3e-3, r          # 1.25 x measured (1.83e-3)


def test_cfg3_headline_geometry_two_steps_with_the_real_unet_vs_reference():
    """BASELINE config 3 itself against the reference with the real UNet (make_golden.py g35): 4096 x 512 x 16f, 8 x 2 shifted windows of
    512 x 320, loop_step 8, CFG 7.5, the 50-step schedule entered at step 24 (schedule indices 25, 24) -- two whole steps of 16 windows
    each, the second shifted by 1/8 window with its last column wrapping across the W seam; 64 forwards of the reference on the CPU.
    The HIP ring loop in bench.py's execution mode (tile batch 8, two streams, hipGraph replay, shared CFG prefix), library default
    operand / residual modes: the panorama latent after each step at the north star's 1e-3; the intermediate pred-x0 panorama is
    reported with a regression guard (it multiplies the guided-eps error by sqrt(1 - a) / sqrt(a) and is overwritten by every later step)."""
    from dynamicscaler_amd.pipelines import VC2_Pipeline_T2V_SpherePano
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    path = os.path.join(G, "cfg3_real_unet_two_steps.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/cfg3_real_unet_two_steps.npz not generated (make_golden.py --full --only g35)")
    d = dev()
    z = np.load(path)
    rec = json.load(open(os.path.join(G, "cfg3_real_unet_two_steps_trace.json")))
    nrec, skip = int(z["steps"]), int(z["skip"])
    assert rec["geom"]["total_w"] == 4096 and rec["geom"]["num_windows_w"] == 8 and rec["geom"]["num_windows_h"] == 2
    ld, params, _ = full_host(d)
    unet = ld.model.diffusion_model
    _reset_mode(unet)                    # the library default
    out = {}
    for mode in ("bench", "plain"):
        pipe = VC2_Pipeline_T2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld, rng_mode="reference"), {"params": {"unet_config": {"params": params}}}).to(d, torch.float32)
        if mode == "bench":
            pipe.max_tile_batch, pipe.num_streams, pipe.use_graph = 8, 2, True
        snaps, trace = [], []

        def cb(i, t, wins, pano, pano_x0):
            trace.append((i, int(t), [list(x) for x in wins]))
            snaps.append((pano.float().cpu().clone(), pano_x0.float().cpu().clone()))
            if len(snaps) == nrec:
                raise _Stop()

        torch.manual_seed(2333333)
        try:
            pipe.basic_sample_shift_multi_windows(prompt="a prompt", fps=int(z["fps"]), guidance_scale=float(z["guidance"]), output_type="latent",
                                                  init_panorama_latent=T(z["init"]).float(), step_callback=cb, use_skip_time=True,
                                                  skip_time_step_idx=skip, **rec["geom"])
        except _Stop:
            pass
        assert len(snaps) == nrec and pipe.wide_steps_run == []
        for (i, t, wins), ref in zip(trace, rec["trace"]):
            assert i == ref["i"] and t == ref["t"] and wins == ref["windows"] and len(wins) == 16, (i, t, wins, ref)
        out[mode] = snaps
    W = rec["geom"]["total_w"] // 8
    assert any(w[1] > W for w in trace[1][2]), "no window of the second step crosses the W seam"
    for k in range(nrec):                 # the execution mode does not change a bit of the panorama
        assert torch.equal(out["bench"][k][0], out["plain"][k][0]) and torch.equal(out["bench"][k][1], out["plain"][k][1]), k
    snaps = out["bench"]
    errs = {k: relerr(snaps[k][0], T(z[f"pano_{k}"])) for k in range(nrec)}
    errs["x0"] = relerr(snaps[-1][1], T(z[f"x0_{nrec - 1}"]))
    stats = {str(k): loop_err_stats(snaps[k][0], T(z[f"pano_{k}"]), trace[k][2]) for k in range(nrec)}
    r = dict(test="cfg3_real_unet_two_steps", residual="outer", errs={str(k): v for k, v in errs.items()}, stats=stats)
    print(r)
    record(**r)
    assert all(e < NORTH_STAR for k, e in errs.items() if k != "x0"), r
    assert all(v["worst_window_rel_l2"] < NORTH_STAR for v in stats.values()), r       # no window is worse than the budget either
    # pred-x0 at schedule index 24 (a = 0.29): (x - 0.84 e_t) / 0.54 carries the guided-eps error of fp16 operands times 1.56; it never
    # leaves the loop at this t (every later step overwrites it; the last step's is asserted at 1e-3 by the "last six steps" tests).
    # Reported; regression guard 1.25 x measured (2.54e-3)
    assert errs["x0"] < 3.2e-3, r


def test_cfg5_dependency_chain_with_the_real_unet_vs_reference():
    """BASELINE config 5's DEPENDENCY SHAPE against the reference with the real UNet (make_golden.py g41): the t2v ring loop on cfg5's window
    grid cut to two columns -- 1024 x 1024 x 24f, 2 x 4 shifted windows of 512 x 320 x 24 frames: chains of FOUR vertically overlapping,
    dependent tiles per column (plan_levels: 4 levels), the UNet at T = 24 -- loop_step 8, CFG 7.5, the 50-step schedule entered at step 24;
    two whole steps of 8 windows, the second shifted so that its bottom row wraps across the H seam and its right column across the W seam;
    32 forwards of the reference at T = 24 on the CPU.  The HIP ring loop in bench.py's execution mode (tile batch 8, two streams, hipGraph,
    shared CFG prefix), library default modes, bit-identical to the plain one-stream eager run: panorama latent after each step at the north
    star's 1e-3, globally AND per window; max |diff| / max |ref| recorded."""
    from dynamicscaler_amd.pipelines import VC2_Pipeline_T2V_SpherePano
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    from dynamicscaler_amd import parallel
    path = os.path.join(G, "cfg5_chain_real_unet.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/cfg5_chain_real_unet.npz not generated (make_golden.py --full --only g41)")
    d = dev()
    z = np.load(path)
    rec = json.load(open(os.path.join(G, "cfg5_chain_real_unet_trace.json")))
    nrec, skip = int(z["steps"]), int(z["skip"])
    geom = rec["geom"]
    assert geom["frames"] == 24 and geom["num_windows_h"] == 4 and geom["num_windows_w"] == 2
    ld, params, _ = full_host(d)
    unet = ld.model.diffusion_model
    _reset_mode(unet)                    # the library default
    out = {}
    for mode in ("bench", "plain"):
        pipe = VC2_Pipeline_T2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld, rng_mode="reference"), {"params": {"unet_config": {"params": params}}}).to(d, torch.float32)
        if mode == "bench":
            pipe.max_tile_batch, pipe.num_streams, pipe.use_graph = 8, 2, True
        snaps, trace = [], []

        def cb(i, t, wins, pano, pano_x0):
            trace.append((i, int(t), [list(x) for x in wins]))
            snaps.append((pano.float().cpu().clone(), pano_x0.float().cpu().clone()))
            if len(snaps) == nrec:
                raise _Stop()

        torch.manual_seed(2333333)
        try:
            pipe.basic_sample_shift_multi_windows(prompt="a prompt", fps=int(z["fps"]), guidance_scale=float(z["guidance"]), output_type="latent",
                                                  init_panorama_latent=T(z["init"]).float(), step_callback=cb, use_skip_time=True,
                                                  skip_time_step_idx=skip, **geom)
        except _Stop:
            pass
        assert len(snaps) == nrec and pipe.wide_steps_run == []
        for (i, t, wins), ref in zip(trace, rec["trace"]):
            assert i == ref["i"] and t == ref["t"] and wins == ref["windows"] and len(wins) == 8, (i, t, wins, ref)
        out[mode] = snaps
    Hl, Wl = geom["total_h"] // 8, geom["total_w"] // 8
    assert any(w[3] > Hl for w in trace[1][2]) and any(w[1] > Wl for w in trace[1][2]), "the second step must cross the H and the W seam"
    # the dependency shape: every column is a chain of four tiles -> four levels of two windows each
    levels = parallel.plan_levels([tuple(w) for w in trace[1][2]], (geom["frames"], Hl, Wl))
    assert [len(lv) for lv in levels] == [2, 2, 2, 2], levels
    for k in range(nrec):
        assert torch.equal(out["bench"][k][0], out["plain"][k][0]) and torch.equal(out["bench"][k][1], out["plain"][k][1]), k
    snaps = out["bench"]
    stats = {str(k): loop_err_stats(snaps[k][0], T(z[f"pano_{k}"]), trace[k][2]) for k in range(nrec)}
    x0 = loop_err_stats(snaps[-1][1], T(z[f"x0_{nrec - 1}"]), trace[-1][2])
    r = dict(test="cfg5_chain_real_unet", residual="outer", stats=stats, x0=x0)
    print(r)
    record(**r)
    assert all(v["rel_l2"] < NORTH_STAR and v["worst_window_rel_l2"] < NORTH_STAR for v in stats.values()), r
    assert x0["rel_l2"] < 3.5e-3, r         # intermediate pred-x0 (overwritten by every later step): reported, regression guard


def test_cfg4_geometry_one_step_with_the_real_i2v_unet_vs_reference():
    """BASELINE config 4's geometry against the reference with the real i2v UNet (make_golden.py g36): 4096 x 512 x 16f, 8 x 2 shifted
    windows, 77 text + 16 image tokens per window from the crop of a 4096 x 512 panorama image under it, merge-prev, CFG 7.5, one whole
    step (16 windows, 32 CPU forwards of the reference) at step 24 of the 50-step schedule.  bench.py's execution mode (tile batch 8, two
    streams, hipGraph, shared CFG prefix) = the plain run bit for bit; panorama latent after the step: the north star's 1e-3."""
    from dynamicscaler_amd.pipelines_i2v import VC2_Pipeline_I2V_SpherePano
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    from dynamicscaler_amd.synth import synth_normal
    path = os.path.join(G, "cfg4_real_unet_one_step.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/cfg4_real_unet_one_step.npz not generated (make_golden.py --full --only g36)")
    d = dev()
    z = np.load(path)
    rec = json.load(open(os.path.join(G, "cfg4_real_unet_one_step_trace.json")))
    skip = int(z["skip"])
    assert rec["geom"]["total_w"] == 4096 and rec["geom"]["num_windows_w"] == 8 and rec["geom"]["num_windows_h"] == 2
    ld, params = _i2v_host(d)
    unet = ld.model.diffusion_model
    _reset_mode(unet)
    img = synth_normal((3, 512, 4096), int(z["pano_img_seed"])).clamp(-1, 1)
    out = {}
    for mode in ("bench", "plain"):
        pipe = VC2_Pipeline_I2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld, rng_mode="reference"), {"params": {"unet_config": {"params": params}}}).to(d, torch.float32)
        if mode == "bench":
            pipe.max_tile_batch, pipe.num_streams, pipe.use_graph = 8, 2, True
        snaps, trace = [], []

        def cb(i, t, wins, pano, pano_x0):
            trace.append((i, int(t), [list(x) for x in wins]))
            snaps.append((pano.float().cpu().clone(), pano_x0.float().cpu().clone()))
            raise _Stop()

        torch.manual_seed(2333333)
        try:
            pipe.basic_sample_shift_multi_windows(prompt="a prompt", fps=int(z["fps"]), guidance_scale=float(z["guidance"]), output_type="latent",
                                                  init_panorama_latent=T(z["init"]).float(), step_callback=cb, use_skip_time=True,
                                                  skip_time_step_idx=skip, pano_image_tensor=img, **rec["geom"])
        except _Stop:
            pass
        assert len(snaps) == 1 and pipe.wide_steps_run == []
        (i, t, wins), ref = trace[0], rec["trace"][0]
        assert i == ref["i"] and t == ref["t"] and wins == ref["windows"] and len(wins) == 16, (i, t, wins, ref)
        out[mode] = snaps[0]
    assert torch.equal(out["bench"][0], out["plain"][0]) and torch.equal(out["bench"][1], out["plain"][1])
    errs = {"0": relerr(out["bench"][0], T(z["pano_0"])), "x0": relerr(out["bench"][1], T(z["x0_0"]))}
    r = dict(test="cfg4_real_unet_one_step", residual="outer", errs=errs, stats={"0": loop_err_stats(out["bench"][0], T(z["pano_0"]), trace[0][2])})
    print(r)
    record(**r)
    assert errs["0"] < NORTH_STAR, r
    assert errs["x0"] < 3.2e-3, r          # intermediate pred-x0 at schedule index 25: reported (see the config 3 test)


def test_sphere_loop_with_the_real_unet_vs_reference():
    """P5 with the REAL UNet (make_golden.py g37): the reference's t2v sphere loop on a 1024 x 512 equirect, five overlapping perspective
    views of 512 x 320 x 16f per step (fov 120; the theta offset walks), CFG 7.5, the first two steps of the 50-step schedule from a given
    latent, no overlap re-noise (its randn_like draw is host-dependent in the reference): 20 CPU forwards of the reference.  The HIP
    sphere loop -- index maps, nearest gather, dependency levels, UNet, fused CFG + DDIM, last-writer-wins scatter -- in the library
    default mode: the panorama latent at the north star's 1e-3.  The pred-x0 panorama at t = 979 (sqrt((1 - a) / a) = 11) is reported."""
    from dynamicscaler_amd.sphere import VC2_Pipeline_T2V_SpherePano
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    path = os.path.join(G, "sphere_real_unet.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/sphere_real_unet.npz not generated (make_golden.py --full --only g37)")
    d = dev()
    z = np.load(path)
    geom = dict(json.load(open(os.path.join(G, "sphere_real_unet.json")))["geom"])
    geom["phi_theta_dict"] = {int(k): v for k, v in geom["phi_theta_dict"].items()}
    ld, params, _ = full_host(d)
    unet = ld.model.diffusion_model
    _reset_mode(unet)
    pipe = VC2_Pipeline_T2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld, rng_mode="reference"), {"params": {"unet_config": {"params": params}}}).to(d, torch.float32)
    steps = []
    torch.manual_seed(2333333)
    final, den = pipe.basic_sample_shift_shpere_panorama(prompt="a prompt", fps=int(z["fps"]), guidance_scale=float(z["guidance"]), output_type="latent",
                                                         init_sphere_latent=T(z["init"]).float(),
                                                         step_callback=lambda i, t, views, p, p0: steps.append((i, int(t), len(views))), **geom)
    assert steps == [(0, 999, 5), (1, 979, 5)] and pipe.wide_steps_run == []
    ref_f, ref_d = T(z["final"]), T(z["denoised"])
    assert final.shape == ref_f.shape and torch.equal((den.cpu() == 0), (ref_d == 0))        # the same pixels were never written
    e_f, e_d = relerr(final, ref_f), relerr(den, ref_d)
    r = dict(test="sphere_real_unet", residual="outer", final=e_f, denoised=e_d)
    print(r)
    record(**r)
    assert e_f < NORTH_STAR, r
    assert e_d < 3.3e-3, r             # reported (pred-x0 at t = 979 never leaves the loop); regression guard 1.25 x measured (2.61e-3)


def test_grid_loop_with_the_real_unet_vs_reference():
    """P4 with the REAL UNet (make_golden.py g38): the reference's non-overlapping shifted grid loop (pipeline/t2v_normal_pipeline.py:213-568)
    on 2 x 1 tiles of 512 x 320 x 16f, loop_step 4 (quarter-tile shifts wrapping in W, H and F), CFG 7.5, config 1's 4-step schedule, 16
    CPU forwards of the reference.  The HIP grid loop under the pipelines' own operand policy (a 4-step schedule: its first three updates
    run on wide operands, like config 1): the final pred-x0 panorama at the north star's 1e-3."""
    from dynamicscaler_amd.pipelines import VC2_Pipeline_T2V
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    path = os.path.join(G, "grid_real_unet.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/grid_real_unet.npz not generated (make_golden.py --full --only g38)")
    d = dev()
    z = np.load(path)
    geom = json.load(open(os.path.join(G, "grid_real_unet.json")))["geom"]
    ld, params, _ = full_host(d)
    _reset_mode(ld.model.diffusion_model)
    pipe = VC2_Pipeline_T2V(ld, lvdm_DDIM_Scheduler(ld, rng_mode="reference"), {"params": {"unet_config": {"params": params}}}).to(d, torch.float32)
    steps = []
    torch.manual_seed(2333333)
    _, den = pipe.basic_sample_shift_multi_windows(prompt="a prompt", height=320, width=512, frames=16, fps=int(z["fps"]), guidance_scale=float(z["guidance"]),
                                                   output_type="latent", init_panorama_latent=T(z["init"]).float(),
                                                   step_callback=lambda i, t, w, p, p0: steps.append((i, int(t), len(w))), **geom)
    assert [s[1:] for s in steps] == [(999, 2), (666, 2), (333, 2), (0, 2)] and pipe.wide_steps_run == [(0, 3), (1, 2), (2, 1)]
    e = relerr(den, T(z["denoised"]))
    r = dict(test="grid_real_unet", policy="auto", denoised=e)
    print(r)
    record(**r)
    assert tuple(den.shape) == tuple(z["denoised"].shape) and e < NORTH_STAR, r


def test_i2v_grid_loop_with_the_real_unet_vs_reference():
    """P4 (i2v) with the REAL i2v UNet (make_golden.py g40): the reference's non-overlapping shifted grid loop of the i2v base class
    (pipeline/i2v_normal_pipeline.py:68-425) on 2 x 1 tiles of 512 x 320 x 16f, loop_step 4, image tokens from the crop of the panorama image
    under each shifted window, 0/1-mask re-noise, CFG 7.5, 4-step schedule; 16 CPU forwards of the reference.  Under the pipelines' operand
    policy (first three updates on wide operands): the final pred-x0 panorama at 1e-3."""
    from dynamicscaler_amd.pipelines_i2v import VC2_Pipeline_I2V
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    from dynamicscaler_amd.synth import synth_normal
    path = os.path.join(G, "grid_i2v_real_unet.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/grid_i2v_real_unet.npz not generated (make_golden.py --full --only g40)")
    d = dev()
    z = np.load(path)
    geom = json.load(open(os.path.join(G, "grid_i2v_real_unet.json")))["geom"]
    ld, params = _i2v_host(d)
    _reset_mode(ld.model.diffusion_model)
    pipe = VC2_Pipeline_I2V(ld, lvdm_DDIM_Scheduler(ld, rng_mode="reference"), {"params": {"unet_config": {"params": params}}}).to(d, torch.float32)
    steps = []
    torch.manual_seed(2333333)
    _, den = pipe.basic_sample_shift_multi_windows(prompt="a prompt", fps=int(z["fps"]), guidance_scale=float(z["guidance"]), output_type="latent",
                                                   init_panorama_latent=T(z["init"]).float(),
                                                   pano_image_tensor=synth_normal((3, 320, 1024), int(z["grid_img_seed"])).clamp(-1, 1),
                                                   step_callback=lambda i, t, w, p, p0: steps.append((i, int(t), len(w))), **geom)
    assert [s[1:] for s in steps] == [(999, 2), (666, 2), (333, 2), (0, 2)] and pipe.wide_steps_run == [(0, 3), (1, 2), (2, 1)]
    e = relerr(den, T(z["denoised"]))
    r = dict(test="grid_i2v_real_unet", policy="auto", denoised=e)
    print(r)
    record(**r)
    assert tuple(den.shape) == tuple(z["denoised"].shape) and e < NORTH_STAR, r


def test_i2v_sphere_loop_with_the_real_unet_vs_reference():
    """P5 (i2v) with the REAL i2v UNet (make_golden.py g39): the reference's i2v sphere loop on a 1024 x 512 equirect, five overlapping
    512 x 320 x 16f views a step with 16 image tokens each from the view's perspective crop of the panorama image, mask-gated overlap
    re-noise (ratio 1; torch's plain randn stream, reproduced on the host), merge-prev 0.3 / 0.2, CFG 7.5, the first two steps of the
    50-step schedule; 20 CPU forwards of the reference.  Panorama latent: 1e-3; the pred-x0 panorama at t = 979 is reported."""
    from dynamicscaler_amd.sphere import VC2_Pipeline_I2V_SpherePano
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    from dynamicscaler_amd.synth import synth_normal
    path = os.path.join(G, "sphere_i2v_real_unet.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/sphere_i2v_real_unet.npz not generated (make_golden.py --full --only g39)")
    d = dev()
    z = np.load(path)
    geom = dict(json.load(open(os.path.join(G, "sphere_i2v_real_unet.json")))["geom"])
    geom["phi_theta_dict"] = {int(k): v for k, v in geom["phi_theta_dict"].items()}
    ld, params = _i2v_host(d)
    _reset_mode(ld.model.diffusion_model)
    pipe = VC2_Pipeline_I2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld, rng_mode="reference"), {"params": {"unet_config": {"params": params}}}).to(d, torch.float32)
    steps = []
    torch.manual_seed(2333333)
    final, den = pipe.basic_sample_shift_shpere_panorama(prompt="a prompt", fps=int(z["fps"]), guidance_scale=float(z["guidance"]), output_type="latent",
                                                         init_sphere_latent=T(z["init"]).float(),
                                                         pano_image_tensor=synth_normal((3, 512, 1024), int(z["pano_img_seed"])).clamp(-1, 1),
                                                         step_callback=lambda i, t, items, p, p0: steps.append((i, int(t), len(items))), **geom)
    assert steps == [(0, 999, 5), (1, 979, 5)] and pipe.wide_steps_run == []
    ref_f, ref_d = T(z["final"]), T(z["denoised"])
    assert final.shape == ref_f.shape and torch.equal((den.cpu() == 0), (ref_d == 0))
    e_f, e_d = relerr(final, ref_f), relerr(den, ref_d)
    r = dict(test="sphere_i2v_real_unet", residual="outer", final=e_f, denoised=e_d)
    print(r)
    record(**r)
    assert e_f < NORTH_STAR, r
    assert e_d < 3.5e-3, r             # reported (pred-x0 at t = 979 never leaves the loop); regression guard 1.25 x measured (2.78e-3)


def test_i2v_ring_loop_real_unet_mid_schedule_vs_reference():
    """The i2v counterpart of the mid-schedule test (make_golden.py g34): the reference's i2v ring loop with the REAL i2v UNet -- 77 text
    + 16 image tokens per window from the crop of the panorama image under it, merge-prev ratios 0.4 .. 0.2 -- on a 1024 x 512 x 16f
    panorama, 2 x 2 shifted windows, loop_step = 8, entered through use_skip_time at step 20 of 50 (schedule indices 29..24).  Panorama
    latent after steps 0 / 2 / 5: 1e-3 in the library default mode; the intermediate pred-x0 panorama after step 5 is reported."""
    from dynamicscaler_amd.pipelines_i2v import VC2_Pipeline_I2V_SpherePano
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    from dynamicscaler_amd.synth import synth_normal
    path = os.path.join(G, "i2v_ring_real_unet_50step_mid.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/i2v_ring_real_unet_50step_mid.npz not generated (make_golden.py --full --only g34)")
    d = dev()
    z = np.load(path)
    rec = json.load(open(os.path.join(G, "i2v_ring_real_unet_50step_mid_trace.json")))
    nrec, skip = int(z["steps"]), int(z["skip"])
    ld, params = _i2v_host(d)
    unet = ld.model.diffusion_model
    _reset_mode(unet)                    # the library default
    pipe = VC2_Pipeline_I2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld, rng_mode="reference"), {"params": {"unet_config": {"params": params}}}).to(d, torch.float32)
    snaps, trace = [], []

    def cb(i, t, wins, pano, pano_x0):
        trace.append((i, int(t), [list(x) for x in wins]))
        snaps.append((pano.float().cpu().clone(), pano_x0.float().cpu().clone()))
        if len(snaps) == nrec:
            raise _Stop()

    torch.manual_seed(2333333)
    try:
        pipe.basic_sample_shift_multi_windows(prompt="a prompt", fps=int(z["fps"]), guidance_scale=float(z["guidance"]), output_type="latent",
                                              init_panorama_latent=T(z["init"]).float(), step_callback=cb, use_skip_time=True, skip_time_step_idx=skip,
                                              pano_image_tensor=synth_normal((3, 512, 1024), int(z["pano_img_seed"])).clamp(-1, 1), **rec["geom"])
    except _Stop:
        pass
    assert len(snaps) == nrec and pipe.wide_steps_run == []
    for (i, t, wins), ref in zip(trace, rec["trace"]):
        assert i == ref["i"] and t == ref["t"] and wins == ref["windows"], (i, t, wins, ref)
    errs = {k: relerr(snaps[k][0], T(z[f"pano_{k}"])) for k in range(nrec) if f"pano_{k}" in z.files}
    errs["x0"] = relerr(snaps[-1][1], T(z[f"x0_{nrec - 1}"]))
    r = dict(test="i2v_ring50_mid_real_unet", residual="outer", errs={str(k): v for k, v in errs.items()},
             stats={str(k): loop_err_stats(snaps[k][0], T(z[f"pano_{k}"]), trace[k][2]) for k in range(nrec) if f"pano_{k}" in z.files})
    print(r)
    record(**r)
    assert len(errs) == 4 and all(e < RING50_TOL for k, e in errs.items() if k != "x0"), r      # the latent panorama: north star
    assert errs["x0"] < 2.5e-3, r          # intermediate pred-x0 (never leaves the loop at this t): regression guard, like the t2v test's


def test_vae_decodes_a_cfg5_frame():
    """N2 at the largest configuration's size: one 128 x 1024 latent frame -> 1024 x 8192 pixels through the real first-stage config.
    The last level's activations are exactly 2 GiB at 128 fp16 channels (4.3 GB at 256 fp32 ones), the mid-block attention has 131 072
    tokens: bands / row chunks / query blocks (vae.py operand_limit).  No reference output exists at this size (a 69 GB score
    matrix on the reference's side); the two operand modes run on independent GEMM, GroupNorm and softmax kernels with different
    band counts, and must agree to fp16-operand accuracy (measured 1.7e-3; the fp16 decode is 2.6e-3 from the reference at 40 x 64)."""
    from dynamicscaler_amd.vae import AutoencoderKLDecoder
    from dynamicscaler_amd.vae_spec import decoder_param_shapes
    from dynamicscaler_amd.synth import synth_state_dict, synth_normal
    d = torch.device("cuda:0")
    zf = np.load(os.path.join(G, "vae_full.npz"))
    dd = json.loads(bytes(zf["full_dd_json"]).decode())
    m = AutoencoderKLDecoder(dd, 4)
    m.load_state_dict(synth_state_dict(decoder_param_shapes(dd, 4), seed=22))
    z = synth_normal((1, 4, 1, 128, 1024), 5).to(d)
    out = m.decode_frames(z, in_scale=1.0 / 0.18215)
    assert out.shape == (1, 3, 1, 1024, 8192) and bool(torch.isfinite(out).all())
    m.operand_mode = "wide"
    wide = m.decode_frames(z, in_scale=1.0 / 0.18215)
    e = float((out.double() - wide.double()).norm() / wide.double().norm())
    print(f"cfg5 frame decode, fp16 vs wide operands: rel-L2 {e:.3e}")
    assert e < 3.5e-3
    del out, wide
    torch.cuda.empty_cache()


def test_decode_tail_at_cfg3_and_cfg5_sizes():
    """P6 at full size: the seam-safe decode tail (t2v_sphere_panorama_pipeline.py:638-655: W padded with the wrapped outer 1/16
    chunks, one decode per frame, the padding cropped) on a cfg3 panorama latent [1, 4, 16, 64, 512] -> [1, 3, 16, 512, 4096] and on
    one frame of cfg5's [.., 128, 1024] (padded to 1152 columns: 2.4 GB operands, decoded in bands), real first-stage config with
    synthetic weights.  Checked against a direct decode of the padded frame on wide operands (independent kernels, fp16-operand
    accuracy) and for the crop's geometry."""
    import types
    from dynamicscaler_amd.host_model import LatentDiffusionHost
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    from dynamicscaler_amd.pipelines import VC2_Pipeline_T2V_SpherePano
    from dynamicscaler_amd.vae_spec import decoder_param_shapes
    from dynamicscaler_amd.synth import synth_state_dict, synth_normal
    import time
    d = torch.device("cuda:0")
    zt = np.load(os.path.join(G, "unet_tiny_t2v.npz"))
    params = json.loads(bytes(zt["params_json"]).decode())
    dd = json.loads(bytes(np.load(os.path.join(G, "vae_full.npz"))["full_dd_json"]).decode())
    ld = LatentDiffusionHost({"params": params}, conditioner=lambda p: None, first_stage_config={"params": {"ddconfig": dd, "embed_dim": 4}},
                             scale_factor=0.18215)
    ld.first_stage_model.load_state_dict(synth_state_dict(decoder_param_shapes(dd, 4), seed=22))
    ld = ld.to(d).eval()
    pipe = VC2_Pipeline_T2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld), {"params": {"unet_config": {"params": params}}})
    for name, shape, frames in (("cfg3", (1, 4, 16, 64, 512), 16), ("cfg5, one frame", (1, 4, 1, 128, 1024), 1)):
        lat = (synth_normal(shape, 9) * 0.18215).to(d)
        st = types.SimpleNamespace(pano=lat, pano_x0=lat, in_device=torch.device("cpu"))
        torch.cuda.synchronize()
        t0 = time.time()
        videos, padded = pipe._finish(st, "tensor", frames, True)
        torch.cuda.synchronize()
        dt = time.time() - t0
        H, W = shape[3] * 8, shape[4] * 8
        assert videos.shape == (1, 3, frames, H, W) and padded.shape[-1] == shape[4] * 18 // 16 and bool(torch.isfinite(videos).all())
        assert torch.equal(padded[..., shape[4] // 16:-(shape[4] // 16)], lat.cpu())
        ld.first_stage_model.operand_mode = "wide"
        ref = ld.decode_first_stage_2DAE(padded[:, :, [0]].to(d))
        ld.first_stage_model.operand_mode = "f16"
        ref = ref[..., W // 16:-(W // 16)]
        e = float((videos[:, :, [0]].double() - ref.double()).norm() / ref.double().norm())
        print(f"decode tail {name}: {frames} frame(s) {H}x{W} in {dt:.2f} s; frame 0 vs the wide-operand decode {e:.3e}")
        assert e < 3.5e-3
        del videos, ref, st, lat
        torch.cuda.empty_cache()


def test_tiled_vae_encode_at_cfg4_size():
    """N2 encode side at full size: tiled_vae_encode_image (i2v_sphere_panorama_pipeline.py:498-562) of a 512 x 4096 panorama image
    (cfg4's; 4 x 4 tiles with 256-pixel margins) through the real first-stage config -> [1, 4, 1, 64, 512]; fp16 against wide
    operands (same seeded posterior noise; the encoder's moments are 1.04e-3 / 1.3e-6 from the reference at 320 x 512 in the two
    modes)."""
    import time
    from dynamicscaler_amd.host_model import LatentDiffusionHost
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    from dynamicscaler_amd.pipelines_i2v import VC2_Pipeline_I2V_SpherePano
    from dynamicscaler_amd.vae_spec import vae_param_shapes
    from dynamicscaler_amd.synth import synth_state_dict, synth_normal
    d = torch.device("cuda:0")
    zt = np.load(os.path.join(G, "unet_tiny_t2v.npz"))
    params = json.loads(bytes(zt["params_json"]).decode())
    dd = json.loads(bytes(np.load(os.path.join(G, "vae_enc_full.npz"))["full_dd_json"]).decode())
    ld = LatentDiffusionHost({"params": params}, conditioner=lambda p: None, first_stage_config={"params": {"ddconfig": dd, "embed_dim": 4}},
                             scale_factor=0.18215)
    ld.first_stage_model.load_state_dict(synth_state_dict(vae_param_shapes(dd, 4), seed=24))
    ld = ld.to(d).eval()
    pipe = VC2_Pipeline_I2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld), {"params": {"unet_config": {"params": params}}})
    img = synth_normal((3, 512, 4096), 31).clamp(-1, 1)
    outs = {}
    for mode in ("f16", "wide"):
        ld.first_stage_model.operand_mode = mode
        torch.manual_seed(81)
        torch.cuda.synchronize()
        t0 = time.time()
        outs[mode] = pipe.tiled_vae_encode_image(image_tensor=img)
        torch.cuda.synchronize()
        print(f"tiled encode 512x4096 [{mode}]: {time.time() - t0:.2f} s")
    assert outs["f16"].shape == (1, 4, 1, 64, 512) and bool(torch.isfinite(outs["f16"]).all())
    e = float((outs["f16"].double() - outs["wide"].double()).norm() / outs["wide"].double().norm())
    print(f"tiled encode, fp16 vs wide operands: rel-L2 {e:.3e}")
    assert e < 2.5e-3


def test_i2v_end_to_end_at_cfg4_size_image_to_frames():
    """Everything between a panorama image and decoded frames at config 4's size, in one call: the tiled first-stage encode of the
    4096 x 512 image (use_skip_time without an init latent, i2v_sphere_panorama_pipeline.py:704-722), its re-noising to the resumed
    schedule position, two ring-loop steps of the real i2v UNet over the 8 x 2 shifted windows with per-window image tokens
    (:777-970), and the decode of the 16 frames (:972-996) -- real first-stage and UNet configs, synthetic weights.  A run-through at
    size (none of these pieces had met at this size): shapes, finiteness, that the loop moved the latent, and that a second call
    with the same seed repeats bit for bit."""
    import time
    import yaml
    from helpers import synth_image_embedder
    from dynamicscaler_amd.host_model import LatentDiffusionHost, SyntheticConditioner
    from dynamicscaler_amd.pipelines_i2v import VC2_Pipeline_I2V_SpherePano
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    from dynamicscaler_amd.unet_spec import param_shapes
    from dynamicscaler_amd.vae import AutoencoderKL
    from dynamicscaler_amd.vae_spec import vae_param_shapes
    from dynamicscaler_amd.synth import synth_state_dict, synth_normal
    d = dev()
    if "i2v" not in _HOST:
        params = yaml.safe_load(open(os.path.join(REPO, "dynamicscaler_amd", "configs", "i2v_512_v1_unet.yaml")))
        ld = LatentDiffusionHost({"params": params}, conditioner=SyntheticConditioner(77, params["context_dim"], cond_seed=11, uncond_seed=12))
        ld.model.diffusion_model.load_state_dict(synth_state_dict(param_shapes(params), 3), strict=True)
        ld.get_image_embeds = synth_image_embedder(params["context_dim"])
        ld.embedder = object()
        ld = ld.to(d)
        _HOST["i2v"] = (ld, params)
    ld, params = _HOST["i2v"]
    dd = json.loads(bytes(np.load(os.path.join(G, "vae_enc_full.npz"))["full_dd_json"]).decode())
    vae = AutoencoderKL(dd, 4)
    vae.load_state_dict(synth_state_dict(vae_param_shapes(dd, 4), seed=24))
    ld.first_stage_model, ld.scale_factor = vae, 0.18215
    geom = dict(height=320, width=512, frames=16, total_w=4096, total_h=512, num_windows_w=8, num_windows_h=2, num_windows_f=1, loop_step=8,
                num_inference_steps=4, overlap_ratio_list_f=[0.0] * 4)
    img = synth_normal((3, 512, 4096), 77).clamp(-1, 1)
    outs = []
    try:
        for rep in range(2):
            pipe = VC2_Pipeline_I2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld, rng_mode="reference"),
                                               {"params": {"unet_config": {"params": params}}}).to(d, torch.float32)
            steps = []
            torch.manual_seed(2333333)
            torch.cuda.synchronize()
            t0 = time.time()
            videos, den = pipe.basic_sample_shift_multi_windows(prompt="a prompt", fps=8, guidance_scale=7.5, output_type="tensor",
                                                                pano_image_tensor=img, use_skip_time=True, skip_time_step_idx=2,
                                                                step_callback=lambda i, t, w, p, p0: steps.append((i, int(t), len(w))), **geom)
            torch.cuda.synchronize()
            print(f"i2v image -> frames at cfg4's size, run {rep}: {time.time() - t0:.1f} s, steps {steps}")
            assert [s[2] for s in steps] == [16, 16] and [s[1] for s in steps] == [333, 0]
            # (the second return value is the W-PADDED latent in the seam-safe decode branch, like the reference's: 512 + 2 x 32 columns)
            assert videos.shape == (1, 3, 16, 512, 4096) and den.shape == (1, 4, 16, 64, 576)
            assert bool(torch.isfinite(videos).all()) and bool(torch.isfinite(den).all())
            outs.append((videos.cpu(), den.cpu()))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        assert float(outs[0][1].std()) > 0.05
    finally:
        ld.first_stage_model = None
        ld.scale_factor = 1.0


def test_grid_loop_with_pre_denoise_at_full_size():
    """P4 at full size: VC2_Pipeline_T2V.basic_sample_shift_multi_windows (pipeline/t2v_normal_pipeline.py:213-568) on an 8 x 2 grid of
    512 x 320 x 16f tiles (4096 x 640) with the real t2v UNet: a pre-denoised tile (2 steps) resized bicubically to the panorama and
    re-noised (:345-412), then the shifted-grid steps with the per-step sparse residual merge (:445-468), 4-step schedule.  A
    run-through at size (shapes, finiteness, the windows of every step, bit-repeatable); parity of each piece is pinned at the toy
    size against the reference (tests/test_gpu_unet.py)."""
    import time
    from dynamicscaler_amd.pipelines import VC2_Pipeline_T2V
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    d = dev()
    ld, params, _ = full_host(d)
    ld = ld.to(d)
    kw = dict(num_windows_w=8, num_windows_h=2, num_windows_f=1, loop_step=4, num_inference_steps=4, use_pre_denoise=True, pre_denoise_steps=2,
              merge_predenoise_ratio_list=[0.5, 0.6, 0.7, 0.8], sparse_add_residual=True)
    outs = []
    for rep in range(2):
        pipe = VC2_Pipeline_T2V(ld, lvdm_DDIM_Scheduler(ld, rng_mode="reference"), {"params": {"unet_config": {"params": params}}}).to(d, torch.float32)
        trace = []
        torch.manual_seed(2333333)
        torch.cuda.synchronize()
        t0 = time.time()
        _, den = pipe.basic_sample_shift_multi_windows(prompt="a prompt", height=320, width=512, frames=16, fps=8, guidance_scale=7.5,
                                                       output_type="latent", step_callback=lambda i, t, w, p, p0: trace.append((i, int(t), len(w))),
                                                       **kw)
        torch.cuda.synchronize()
        print(f"grid loop 4096x640x16f with pre-denoise, run {rep}: {time.time() - t0:.1f} s, steps {trace}")
        assert den.shape == (1, 4, 16, 80, 512) and bool(torch.isfinite(den).all())
        assert len(trace) >= 2 and all(n == 16 for _, _, n in trace)
        outs.append(den.cpu())
    assert torch.equal(outs[0], outs[1])


def test_t2v_end_to_end_at_cfg5_size_latent_to_frames():
    """Config 5 through to pixels: two ring-loop steps of the real t2v UNet at T = 24 over the 16 x 4 shifted windows of an
    8192 x 1024 x 24f panorama (64 tiles per step) and the seam-safe decode of the 24 frames -- each a 1024 x 9216 padded image whose
    last decoder level is 2.4 GB per operand, i.e. decoded in bands (vae.py operand_limit).  A run-through at size: shapes,
    finiteness, timing."""
    import time
    from dynamicscaler_amd.pipelines import VC2_Pipeline_T2V_SpherePano
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    from dynamicscaler_amd.vae import AutoencoderKL
    from dynamicscaler_amd.vae_spec import decoder_param_shapes
    from dynamicscaler_amd.synth import synth_state_dict
    d = dev()
    ld, params, _ = full_host(d)
    ld = ld.to(d)
    dd = json.loads(bytes(np.load(os.path.join(G, "vae_full.npz"))["full_dd_json"]).decode())
    vae = AutoencoderKL(dd, 4)
    vae.load_state_dict(synth_state_dict(decoder_param_shapes(dd, 4), seed=22))
    old = (ld.first_stage_model, ld.scale_factor, ld.temporal_length)
    ld.first_stage_model, ld.scale_factor = vae.to(d), 0.18215
    geom = dict(height=320, width=512, frames=24, total_w=8192, total_h=1024, num_windows_w=16, num_windows_h=4, num_windows_f=1, loop_step=8,
                num_inference_steps=2)
    try:
        pipe = VC2_Pipeline_T2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld, rng_mode="device"), {"params": {"unet_config": {"params": params}}}).to(d, torch.float32)
        steps = []
        torch.manual_seed(2333333)
        torch.cuda.synchronize()
        t0 = time.time()

        def cb(i, t, w, p, p0):
            torch.cuda.synchronize()
            steps.append((i, int(t), len(w), round(time.time() - t0, 1)))
        videos, den = pipe.basic_sample_shift_multi_windows(prompt="a prompt", fps=8, guidance_scale=7.5, output_type="tensor", step_callback=cb, **geom)
        torch.cuda.synchronize()
        print(f"t2v cfg5 latent -> frames: {time.time() - t0:.1f} s in all, steps (i, t, windows, s since start) {steps}")
        assert [s[2] for s in steps] == [64, 64]
        assert videos.shape == (1, 3, 24, 1024, 8192) and den.shape == (1, 4, 24, 128, 1152)
        assert bool(torch.isfinite(videos).all()) and bool(torch.isfinite(den).all())
    finally:
        ld.first_stage_model, ld.scale_factor, ld.temporal_length = old
        torch.cuda.empty_cache()


def test_gen_pano_360_stage_chain_at_full_size():
    """gen_pano_360.py's three stages (main(): 227-384) at its own sizes, real i2v UNet + real first-stage configs, synthetic weights and
    panorama image, a 4-step schedule cut to two steps per stage: (1) the i2v sphere loop on the 2048 x 1024 equirect -- 44 views a step,
    paste_on_static with the tiled VAE encode of the panorama image at every step, merge-prev, denoise_to_step; (2) nearest resize to
    1024 x 512 and the i2v ring loop resumed with use_skip_time (2 x 2 windows); (3) bicubic x2, re_noise, the ring loop at 2048 x 1024
    (4 x 4 windows) and the seam-safe decode of its 16 frames.  A run-through of the hand-offs at size (shapes, finiteness, timing);
    fp16 operands throughout (the operand policy would run a 4-step schedule's first steps wide: config 1's tests cover that)."""
    import time
    import yaml
    from helpers import synth_image_embedder
    from dynamicscaler_amd.host_model import LatentDiffusionHost, SyntheticConditioner
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    from dynamicscaler_amd.sphere import VC2_Pipeline_I2V_SpherePano
    from dynamicscaler_amd.tensor_utils import resize_video_latent
    from dynamicscaler_amd.unet_spec import param_shapes
    from dynamicscaler_amd.vae import AutoencoderKL
    from dynamicscaler_amd.vae_spec import vae_param_shapes
    from dynamicscaler_amd.synth import synth_state_dict, synth_normal
    d = dev()
    if "i2v" not in _HOST:
        params = yaml.safe_load(open(os.path.join(REPO, "dynamicscaler_amd", "configs", "i2v_512_v1_unet.yaml")))
        ld = LatentDiffusionHost({"params": params}, conditioner=SyntheticConditioner(77, params["context_dim"], cond_seed=11, uncond_seed=12))
        ld.model.diffusion_model.load_state_dict(synth_state_dict(param_shapes(params), 3), strict=True)
        ld.get_image_embeds = synth_image_embedder(params["context_dim"])
        ld.embedder = object()
        ld = ld.to(d)
        _HOST["i2v"] = (ld, params)
    ld, params = _HOST["i2v"]
    dd = json.loads(bytes(np.load(os.path.join(G, "vae_enc_full.npz"))["full_dd_json"]).decode())
    vae = AutoencoderKL(dd, 4)
    vae.load_state_dict(synth_state_dict(vae_param_shapes(dd, 4), seed=24))
    ld.first_stage_model, ld.scale_factor = vae.to(d), 0.18215
    N, stop = 4, 2
    ring6 = [360 * t // 6 for t in range(6)]
    try:
        pipe = VC2_Pipeline_I2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld, rng_mode="device"), {"params": {"unet_config": {"params": params}}}).to(d, torch.float32)
        pipe.operand_policy = "f16"
        pipe.use_graph = True
        img1 = synth_normal((3, 1024, 2048), 89).clamp(-1, 1)
        t0 = time.time()
        sphere_lat, _ = pipe.basic_sample_shift_shpere_panorama(
            prompt="a prompt", img_cond_path=["unused.png"], height=320, width=512, frames=16, fps=8, guidance_scale=7.5, init_panorama_latent=None,
            use_skip_time=False, skip_time_step_idx=0, progressive_skip=False, loop_step=8, pano_image_path=None, pano_image_tensor=img1,
            total_f=16, dock_at_f=False, overlap_ratio_list_f=[0.75, 0.75, 0.5, 0.5], loop_step_frame=8, equirect_width=2048, equirect_height=1024,
            phi_theta_dict={90: [0], -90: [0], 75: ring6, -75: ring6, 60: ring6, -60: ring6, 45: ring6, -45: ring6, 0: ring6},
            phi_prompt_dict=None, view_fov=120, loop_step_theta=10, merge_renoised_overlap_latent_ratio=1, paste_on_static=True,
            view_get_scale_factor=1, view_set_scale_factor=1, denoise_to_step=stop, merge_prev_denoised_ratio_list=[0.5, 0.25, 0, 0],
            downsample_factor_before_vae_decode=1, latents=None, num_inference_steps=N, num_videos_per_prompt=1, generator_seed=1,
            output_type="latent")
        torch.cuda.synchronize()
        t1 = time.time()
        assert sphere_lat.shape == (1, 4, 16, 128, 256) and bool(torch.isfinite(sphere_lat.float()).all())
        lat1 = resize_video_latent(sphere_lat.clone(), target_height=64, target_width=128, mode="nearest")
        ring_args = dict(prompt="a prompt", img_cond_path=["unused.png"], height=320, width=512, frames=16, fps=8, guidance_scale=7.5,
                         use_skip_time=True, skip_time_step_idx=stop, progressive_skip=False, num_windows_f=1, loop_step=8, pano_image_path=None,
                         total_f=16, dock_at_f=False, loop_step_frame=8, merge_prev_denoised_ratio_list=[0.5, 0.25, 0, 0], latents=None,
                         num_inference_steps=N, num_videos_per_prompt=1, generator_seed=1)
        img2 = synth_normal((3, 512, 1024), 90).clamp(-1, 1)
        _, lat2 = pipe.basic_sample_shift_multi_windows(init_panorama_latent=lat1, total_h=512, total_w=1024, num_windows_h=2, num_windows_w=2,
                                                        overlap_ratio_list_f=[0.75, 0.75, 0.5, 0.5], pano_image_tensor=img2, output_type="latent",
                                                        **ring_args)
        torch.cuda.synchronize()
        t2 = time.time()
        assert lat2.shape == (1, 4, 16, 64, 128) and bool(torch.isfinite(lat2.float()).all())
        up = resize_video_latent(lat2.clone(), target_height=128, target_width=256, mode="bicubic")
        pipe.scheduler.make_schedule(N)
        mixed = pipe.scheduler.re_noise(up, 0, N - stop)
        videos, lat3 = pipe.basic_sample_shift_multi_windows(init_panorama_latent=mixed, total_h=1024, total_w=2048, num_windows_h=4, num_windows_w=4,
                                                             overlap_ratio_list_f=[0.75, 0.75, 0.5, 0.5], pano_image_tensor=img1, output_type="tensor",
                                                             **ring_args)
        torch.cuda.synchronize()
        t3 = time.time()
        print(f"gen_pano_360 chain at size: sphere stage (2 steps) {t1 - t0:.1f} s, 1x plane (2 steps) {t2 - t1:.1f} s, 2x plane (2 steps) + decode {t3 - t2:.1f} s")
        assert lat3.shape == (1, 4, 16, 128, 288) and videos.shape == (1, 3, 16, 1024, 2048)
        assert bool(torch.isfinite(videos).all()) and bool(torch.isfinite(lat3.float()).all())
    finally:
        ld.first_stage_model = None
        ld.scale_factor = 1.0
        torch.cuda.empty_cache()
