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


def _pipe(d, latent_dtype, residual_dtype):
    from dynamicscaler_amd.pipelines import VC2_Pipeline_T2V
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    ld, params, _ = full_host(d)
    ld.model.diffusion_model.residual_dtype = residual_dtype
    sched = lvdm_DDIM_Scheduler(ld)
    pipe = VC2_Pipeline_T2V(ld, sched, {"params": {"unet_config": {"params": params}}}).to(d, latent_dtype)
    sched.make_schedule(50, verbose=False)
    return ld, sched, pipe


# measured on MI355X (round 3); asserted at <= 2x measured and never above the north star for x_prev
TF_TOL_X0 = {"float16": 9e-3, "float32": 6e-3}          # pred_x0 = (x - sqrt(1-a) e)/sqrt(a): amplified by sqrt((1-a)/a) at high t


@pytest.mark.parametrize("residual", ["float16", "float32"])
def test_teacher_forced_updates_meet_the_north_star_at_every_index(residual):
    from oracle import ddim as oddim
    from dynamicscaler_amd import ops
    d = dev()
    z = _golden()
    rd = getattr(torch, residual)
    ld, sched, pipe = _pipe(d, torch.float16, rd)
    osched = oddim.DDIMSchedule(oddim.DiffusionTables(), 50)
    cond, uncond = ld.get_learned_conditioning(["a prompt"]), ld.get_learned_conditioning([""])
    g, fps = float(z["guidance"]), int(z["fps"])
    worst = 0.0
    try:
        for idx in [int(i) for i in z["tf_indices"]]:
            t = int(z[f"tf_t_{idx}"])
            assert t == int(np.flip(sched.ddim_timesteps)[49 - idx])
            x = T(z[f"tf_x_t_{idx}"]).to(d)                                      # fp16: both sides start from identical numbers
            eps = pipe._eps(torch.cat([x, x], 0), t, [cond, uncond], fps, 16, cfg_pairs=1, clean_cond=True)
            e_t = eps[1:] + g * (eps[:1] - eps[1:])
            xp, x0 = ops.cfg_ddim(x, eps[:1].contiguous(), eps[1:].contiguous(), (1, 4, 16, 40, 64), g, sched.step_coefficients(idx))
            e_ref = T(z[f"tf_e_t_{idx}"])
            rxp, rx0 = oddim.ddim_step(osched, x.float().cpu(), e_ref, [idx] * 16, noise=torch.zeros_like(e_ref))
            assert torch.equal(rxp, T(z[f"tf_x_prev_{idx}"]))                    # the oracle's update IS the reference's (bit-exact)
            r = dict(test="sched50_teacher_forced", residual=residual, index=idx, t=t, e_cond=relerr(eps[:1], T(z[f"tf_e_cond_{idx}"]).float()),
                     e_t=relerr(e_t, e_ref), x_prev=relerr(xp, rxp), pred_x0=relerr(x0, rx0),
                     # the same error against the size of the UPDATE (x_prev - x_t) instead of the latent it is added to: with the
                     # synthetic weights |x_t| grows along the schedule (eps stays ~0.4), which flatters x_prev's relative error
                     update=float((xp.float().cpu() - rxp).norm() / (rxp - x.float().cpu()).norm()))
            print(r)
            record(**r)
            worst = max(worst, r["x_prev"])
            assert r["x_prev"] < NORTH_STAR, r
            assert r["pred_x0"] < TF_TOL_X0[residual], r
    finally:
        ld.model.diffusion_model.residual_dtype = torch.float16
    print(f"worst teacher-forced x_prev over the schedule, {residual} residual stream: {worst:.3e}")


# free-running drift: measured on MI355X (round 3), asserted at <= 2x measured
FREE_TOL = {("float16", "float16"): 4e-3, ("float16", "float32"): 4e-3, ("float32", "float16"): 4e-3, ("float32", "float32"): 4e-3}


@pytest.mark.parametrize("residual", ["float16", "float32"])
@pytest.mark.parametrize("latents", ["float16", "float32"])
def test_free_running_50_steps_drift_vs_the_reference(residual, latents):
    """basic_sample's loop, 50 steps end to end: every step's error feeds the next.  The curve (x_prev every 5 steps) and the final
    pred_x0 -- the latent that gets decoded."""
    d = dev()
    z = _golden()
    ld, sched, pipe = _pipe(d, getattr(torch, latents), getattr(torch, residual))
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
        ld.model.diffusion_model.residual_dtype = torch.float16
    r = dict(test="sched50_free_running", residual=residual, latents=latents, x_prev_by_index={str(k): v for k, v in curve.items()},
             final_pred_x0=final)
    print(r)
    record(**r)
    tol = FREE_TOL[(residual, latents)]
    assert max(curve.values()) < tol and final < tol, r
