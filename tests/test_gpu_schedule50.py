"""-m gpu: the north-star tolerance ON THE SCHEDULE THE METRIC RUNS -- 50 DDIM steps, one real-size tile, the real t2v UNet,
CFG 7.5 -- against tests/golden/cfg1_50step_t2v.npz (make_golden.py g23: the reference's own VC2_Pipeline_T2V.basic_sample,
pipeline/t2v_normal_pipeline.py:69-210, and lvdm_DDIM_Scheduler.ddim_step, pipeline/scheduler.py:60-96, 120 forwards of the
reference on CPU):

  * teacher-forced, ONE update per schedule index in TF_INDICES (49 = the first step of every metric run, t = 999, where the
    update multiplies the guided-eps error most ... 0): HIP x_prev / pred_x0 from the reference's own x_t against the
    reference's; **1e-3 asserted on x_prev at every index**, in both residual-stream modes;
  * free-running, all 50 steps from the same init latent: x_prev against the reference's every 5 steps (the drift curve)
    and the final pred_x0 (what gets decoded).

Measured numbers -> gpurun_out/measured_parity.jsonl -> DESIGN.md section 5.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(REPO, "tests", "golden")

from test_gpu_fullsize import dev, T, relerr, record, full_host      # noqa: E402

NORTH_STAR = 1e-3
GOLDEN = os.path.join(G, "cfg1_50step_t2v.npz")


def _golden():
    if not os.path.exists(GOLDEN):
        pytest.skip("tests/golden/cfg1_50step_t2v.npz not generated yet (make_golden.py --full --only g23, ~2.5 h of CPU)")
    return np.load(GOLDEN)


RESIDUAL_MODES = {"float16": (torch.float16, "full"), "float32": (torch.float32, "full"), "outer": (torch.float32, "outer")}


def _set_mode(unet, residual):
    unet.residual_dtype, unet.residual_scope = RESIDUAL_MODES[residual]


def _pipe(d, latent_dtype, residual):
    from dynamicscaler_amd.pipelines import VC2_Pipeline_T2V
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    ld, params, _ = full_host(d)
    _set_mode(ld.model.diffusion_model, residual)
    sched = lvdm_DDIM_Scheduler(ld)
    pipe = VC2_Pipeline_T2V(ld, sched, {"params": {"unet_config": {"params": params}}}).to(d, latent_dtype)
    sched.make_schedule(50, verbose=False)
    return ld, sched, pipe


# Measured on MI355X (round 3, profiles/r3_measured_parity.jsonl).  With fp32 latents (the default) x_prev per index is
#   fp16 stream 5.6e-4 4.8e-4 3.7e-4 2.5e-4 1.0e-4 2.7e-5 1.7e-5 7e-6 1.5e-5 0;  fp32 stream 3.7e-4 3.2e-4 2.5e-4 1.6e-4 6e-5 1.3e-5 ...
# and with fp16 latents (the stored tile's own rounding on top), x_prev / pred_x0 per schedule index:
#   index (t)         49 (999)  48 (979)  47 (958)  45 (917)  40 (816)  30 (612)  25 (510)  10 (204)  1 (20)    0 (0)
#   fp16 stream  x_prev  5.9e-4   5.2e-4   4.3e-4   3.2e-4   2.3e-4   2.1e-4   2.1e-4   2.1e-4   2.1e-4   ~0
#                pred_x0 4.4e-3   3.8e-3   3.2e-3   2.3e-3   1.1e-3   4.2e-4   3.2e-4   2.2e-4   2.1e-4   2.1e-4
#                guided e_t 1.0e-2 ... 6.7e-3 (single forward 1.66e-3 ... 8.0e-4: the UNet is most exact at low t)
#   fp32 stream  x_prev  4.2e-4   3.8e-4   3.2e-4   2.6e-4   2.2e-4   2.1e-4 ...                              (floor 2.1e-4 =
#                pred_x0 3.0e-3   2.5e-3   2.1e-3   1.5e-3   6.7e-4   2.7e-4 ...                               the fp16 rounding
#                guided e_t 6.7e-3 ... 2.7e-3                                                                   of the stored tile)
# x_prev (the latent the loop carries): inside the north star at EVERY index in both modes -- asserted at 1e-3.
# pred_x0 = (x - sqrt(1-a) e)/sqrt(a) multiplies the guided-eps error by sqrt((1-a)/a) (14 at t = 999): it is an intermediate
# estimate at those steps (it only matters at index 0, where it is 2.1e-4); asserted at <= 2x measured.
TF_TOL_X0 = {"float16": 9e-3, "float32": 5e-3, "outer": 6.6e-3}       # measured at index 49: 4.4e-3 / 2.5e-3 / 3.3e-3


@pytest.mark.parametrize("residual", ["float16", "outer", "float32"])
@pytest.mark.parametrize("latents", ["float32", "float16"])
def test_teacher_forced_updates_meet_the_north_star_at_every_index(residual, latents):
    from oracle import ddim as oddim
    from dynamicscaler_amd import ops
    d = dev()
    z = _golden()
    ld, sched, pipe = _pipe(d, getattr(torch, latents), residual)
    osched = oddim.DDIMSchedule(oddim.DiffusionTables(), 50)
    cond, uncond = ld.get_learned_conditioning(["a prompt"]), ld.get_learned_conditioning([""])
    g, fps = float(z["guidance"]), int(z["fps"])
    worst = 0.0
    try:
        for idx in [int(i) for i in z["tf_indices"]]:
            t = int(z[f"tf_t_{idx}"])
            assert t == int(np.flip(sched.ddim_timesteps)[49 - idx])
            x = T(z[f"tf_x_t_{idx}"]).to(d, getattr(torch, latents))            # fp16-representable: both sides start from identical numbers
            eps = pipe._eps(torch.cat([x, x], 0), t, [cond, uncond], fps, 16, cfg_pairs=1, clean_cond=True)
            e_t = eps[1:] + g * (eps[:1] - eps[1:])
            xp, x0 = ops.cfg_ddim(x, eps[:1].contiguous(), eps[1:].contiguous(), (1, 4, 16, 40, 64), g, sched.step_coefficients(idx))
            e_ref = T(z[f"tf_e_t_{idx}"])
            # x_prev: the reference's own.  pred_x0 is stored by SHA-256 only; the oracle's ddim_step reproduces it from (x_t, e_t)
            # bit for bit on the build host (tests/test_oracle_golden.py::test_g23_...) and to an ulp on any other CPU
            rxp = T(z[f"tf_x_prev_{idx}"])
            oxp, rx0 = oddim.ddim_step(osched, x.float().cpu(), e_ref, [idx] * 16, noise=torch.zeros_like(e_ref))
            assert relerr(oxp, rxp) < 1e-6
            r = dict(test="sched50_teacher_forced", residual=residual, latents=latents, index=idx, t=t, e_cond=relerr(eps[:1], T(z[f"tf_e_cond_{idx}"]).float()),
                     e_t=relerr(e_t, e_ref), x_prev=relerr(xp, rxp), pred_x0=relerr(x0, rx0),
                     # the same error against the size of the UPDATE (x_prev - x_t) instead of the latent it is added to: with the
                     # synthetic weights |x_t| grows along the schedule (eps stays ~0.4), which flatters x_prev's relative error
                     update=float((xp.float().cpu() - rxp).norm() / (rxp - x.float().cpu()).norm()))
            # ... and what the same guided-eps error would do to x_prev of a TRAINED model, whose eps and latents have unit
            # scale at high t (x_prev = cx x + ce e with |e| = |x_prev| = 1): |ce| x rel(e_t).  Scale-free; reported, and
            # asserted for the strict mode.
            co = sched.step_coefficients(idx)
            r["x_prev_unit_scale"] = abs(co["dir_coef"] - co["sqrt_a_prev"] * co["sqrt_one_minus_at"] / co["sqrt_at"]) * r["e_t"]
            print(r)
            record(**r)
            worst = max(worst, r["x_prev"])
            assert r["x_prev"] < NORTH_STAR, r
            assert r["pred_x0"] < TF_TOL_X0[residual], r
            assert r["x_prev_unit_scale"] < (1.5e-3 if residual == "float16" else NORTH_STAR), r   # measured 1.27e-3 (fp16 stream) / 8.5e-4 (fp32 stream) at t = 999
    finally:
        _set_mode(ld.model.diffusion_model, "outer")   # the library default
    print(f"worst teacher-forced x_prev over the schedule, {residual} residual stream, {latents} latents: {worst:.3e}")


# free-running drift, measured on MI355X (round 3, gpurun_out/sched50a), x_prev after 5 / 25 / 50 steps and the final pred_x0:
#   fp32 latents (the default, as the reference)   fp16 residual stream 9.2e-4 / 9.8e-4 / 9.8e-4, final 9.8e-4
#                                                  fp32 residual stream 6.0e-4 / 6.4e-4 / 6.4e-4, final 6.4e-4
#   fp16 latents                                   fp16 residual stream 1.02e-3 / 1.40e-3 / 1.74e-3, final 1.75e-3
#                                                  fp32 residual stream 7.6e-4 / 1.19e-3 / 1.58e-3, final 1.59e-3
# fp16 STORAGE of the latent rounds it once per step (2^-11 / sqrt(3) = 2.8e-4 relative) and the roundings random-walk:
# sqrt(50) x 2.8e-4 = 2.0e-3 -- that, not the kernels, is what the fp16-latent rows show.  With fp32 latents the whole 50-step
# run stays inside the north star in BOTH residual-stream modes; asserted.  fp16 latents: <= 2x measured.
FREE_TOL = {("float16", "float32"): NORTH_STAR, ("float32", "float32"): NORTH_STAR, ("outer", "float32"): NORTH_STAR,
            ("float16", "float16"): 3.5e-3, ("float32", "float16"): 3.2e-3, ("outer", "float16"): 3.4e-3}


def test_free_running_50_steps_in_the_wide_operand_mode_track_the_reference():
    """The whole 50-step trajectory of basic_sample (120 forwards of the reference, make_golden.py g23) with EVERY step evaluated in the
    wide operand mode (operand_policy "wide"): x_prev every 5 steps and the final pred_x0 stay within 5e-5 of the reference -- the
    kernels' own contribution to the drift is two orders below the north star; what the fp16-operand rows of the test below show is
    operand rounding, not a defect of the loop."""
    d = dev()
    z = _golden()
    ld, sched, pipe = _pipe(d, torch.float32, "outer")
    pipe.operand_policy = "wide"
    cond, uncond = ld.get_learned_conditioning(["a prompt"]), ld.get_learned_conditioning([""])
    g, fps = float(z["guidance"]), int(z["fps"])
    timesteps = np.flip(sched.ddim_timesteps)
    lat = T(z["x_init"]).to(d, torch.float32)
    curve = {}
    for i, t in enumerate(timesteps):
        lat, den = pipe._basic_denoise_one_step(lat, t, i, 50, cond, uncond, g, fps, 16, {})
        idx = 49 - i
        if idx % 5 == 0:
            curve[idx] = relerr(lat, T(z[f"free_x_prev_{idx}"]))
    final = relerr(den, T(z["free_pred_x0_0"]))
    r = dict(test="sched50_free_running", residual="wide", latents="float32", x_prev_by_index={str(k): v for k, v in curve.items()}, final_pred_x0=final)
    print(r)
    record(**r)
    assert len(pipe.wide_steps_run) == 50 and max(curve.values()) < 5e-5 and final < 5e-5, r


@pytest.mark.parametrize("residual", ["float16", "outer", "float32"])
@pytest.mark.parametrize("latents", ["float16", "float32"])
def test_free_running_50_steps_drift_vs_the_reference(residual, latents):
    """basic_sample's loop, 50 steps end to end: every step's error feeds the next.  The curve (x_prev every 5 steps) and the final
    pred_x0 -- the latent that gets decoded."""
    d = dev()
    z = _golden()
    ld, sched, pipe = _pipe(d, getattr(torch, latents), residual)
    cond, uncond = ld.get_learned_conditioning(["a prompt"]), ld.get_learned_conditioning([""])
    g, fps = float(z["guidance"]), int(z["fps"])
    timesteps = np.flip(sched.ddim_timesteps)
    assert list(timesteps) == list(z["timesteps"])
    try:
        lat = T(z["x_init"]).to(d, getattr(torch, latents))
        curve = {}
        for i, t in enumerate(timesteps):
            lat, den = pipe._basic_denoise_one_step(lat, t, i, 50, cond, uncond, g, fps, 16, {})
            idx = 49 - i
            if idx % 5 == 0:
                curve[idx] = relerr(lat, T(z[f"free_x_prev_{idx}"]))
        final = relerr(den, T(z["free_pred_x0_0"]))
    finally:
        _set_mode(ld.model.diffusion_model, "outer")   # the library default
    r = dict(test="sched50_free_running", residual=residual, latents=latents, x_prev_by_index={str(k): v for k, v in curve.items()},
             final_pred_x0=final)
    print(r)
    record(**r)
    tol = FREE_TOL[(residual, latents)]
    assert max(curve.values()) < tol and final < tol, r


@pytest.mark.parametrize("tag", ["i2v", "t24"])
@pytest.mark.parametrize("residual", ["float16", "float32"])
def test_teacher_forced_updates_of_the_other_configs_models(tag, residual):
    """The same per-update assertion for what configs 4 and 5 run: the i2v UNet with 77 text + 16 image tokens, and the t2v UNet
    on a 24-frame tile -- one update of the 50-step schedule at indices 49 and 25 against the reference's own forward +
    ddim_step (tests/golden/updates_i2v_t24.npz, make_golden.py g26)."""
    import yaml
    from oracle import ddim as oddim
    from dynamicscaler_amd import ops
    from dynamicscaler_amd.unet import UNetModel
    from dynamicscaler_amd.unet_spec import param_shapes
    from dynamicscaler_amd.synth import synth_state_dict
    path = os.path.join(G, "updates_i2v_t24.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/updates_i2v_t24.npz not generated (make_golden.py --full --only g26)")
    d = dev()
    z = np.load(path)
    if tag == "i2v":
        params = yaml.safe_load(open(os.path.join(REPO, "dynamicscaler_amd", "configs", "i2v_512_v1_unet.yaml")))
        m = UNetModel(**params)
        m.load_state_dict(synth_state_dict(param_shapes(params), 3), strict=True)
        m = m.to(d).eval()
    else:
        m = full_host(d)[0].model.diffusion_model
    _set_mode(m, residual)
    osched = oddim.DDIMSchedule(oddim.DiffusionTables(), 50)
    g = float(z["guidance"])
    ctx = torch.cat([T(z[f"{tag}_cond"]), T(z[f"{tag}_uncond"])]).float().to(d)
    try:
        for idx in [int(i) for i in z["indices"]]:
            x = T(z[f"{tag}_x_t_{idx}"]).float().to(d)
            t = int(z[f"{tag}_t_{idx}"])
            eps = m(torch.cat([x, x]), torch.tensor([t, t], device=d), context=ctx, fps=int(z[f"{tag}_fps"]), cfg_pairs=1)
            e_t = eps[1:] + g * (eps[:1] - eps[1:])
            xp, x0 = ops.cfg_ddim(x, eps[:1].contiguous(), eps[1:].contiguous(), (1,) + tuple(x.shape[1:]), g, osched.step_coefficients(idx))
            r = dict(test="update_other_models", model=tag, residual=residual, index=idx, t=t, e_t=relerr(e_t, T(z[f"{tag}_e_t_{idx}"])),
                     x_prev=relerr(xp, T(z[f"{tag}_x_prev_{idx}"])))
            print(r)
            record(**r)
            assert r["x_prev"] < NORTH_STAR, r
    finally:
        _set_mode(m, "outer")   # the library default
