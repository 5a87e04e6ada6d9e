"""-m gpu: the strict-precision mode -- UNetModel.residual_dtype = torch.float32 (the residual stream stored, added and
normalised in fp32; matrix-core operands stay fp16) -- against the same reference goldens as the default fp16 mode.

The reference computes in fp32 end to end (openaimodel3d.py:657-708); BASELINE.json's north_star asks for 1e-3 rel on the
latents.  The error budget (profiles/r2_notes.md section 2) attributes 1.31e-3 of the fp16 mode's 1.66e-3 on eps to the
fp16 roundings of the residual stream; this mode removes that term.  Measured numbers go to
gpurun_out/measured_parity.jsonl; DESIGN.md section 5 quotes them; tolerances are <= 2x measured.
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(REPO, "tests", "golden")

from test_gpu_fullsize import dev, T, relerr, record, t2v_params      # noqa: E402

EPS_TOL_TINY_STRICT = 2.6e-3    # toy UNet eps, strict mode: measured 1.18e-3 .. 1.33e-3 (fp16 mode 2.1e-3 .. 2.3e-3)
EPS_TOL_STRICT = 1e-3           # full-size t2v UNet eps, strict mode: measured 9.13e-4 / 9.30e-4 -- the north-star figure on the
                                # UNet output itself (fp16 mode 1.66e-3 / 1.70e-3, "outer" 1.23e-3 / 1.25e-3)


def build_unet(params, seed, device, residual_dtype):
    from dynamicscaler_amd.unet import UNetModel
    from dynamicscaler_amd.unet_spec import param_shapes
    from dynamicscaler_amd.synth import synth_state_dict
    m = UNetModel(**params)
    m.load_state_dict(synth_state_dict(param_shapes(params), seed), strict=True)
    m.residual_dtype, m.residual_scope = residual_dtype, "full"      # strict = fp32 everywhere (the library default is the "outer" scope)
    return m.to(device).eval()


@pytest.mark.parametrize("name", ["t2v", "i2v"])
def test_unet_tiny_strict_vs_reference_golden(name):
    d = dev()
    z = np.load(os.path.join(G, f"unet_tiny_{name}.npz"))
    params = json.loads(bytes(z["params_json"]).decode())
    errs = {}
    for rd in (torch.float16, torch.float32):
        m = build_unet(params, 5, d, rd)
        e = []
        for case in range(3):
            x, t, ctx = T(z[f"x_{case}"]), T(z[f"t_{case}"]), T(z[f"ctx_{case}"])
            eps = m(x.to(d, torch.float16), t.to(d), context=ctx.to(d), fps=int(z[f"fps_{case}"]))
            assert eps.dtype == torch.float32 and eps.shape == tuple(z[f"eps_{case}"].shape)
            e.append(relerr(eps, T(z[f"eps_{case}"])))
        errs[rd] = e
    print(f"toy {name} UNet eps rel err: fp16 stream {errs[torch.float16]}, fp32 stream {errs[torch.float32]}")
    record(test="tiny_strict", model=name, f16=errs[torch.float16], f32=errs[torch.float32])
    assert max(errs[torch.float32]) < EPS_TOL_TINY_STRICT
    assert max(errs[torch.float32]) < 0.8 * max(errs[torch.float16])          # the mode has to buy something


def test_unet_strict_batch_cfg_pairs_and_concat_invariants():
    """The bit-level invariants of the fp16 mode hold in the strict mode too: a batch equals its separate forwards, the
    shared CFG prefix equals the plain 2n forward, the in-place skip tensors equal the concat copy; switching the attribute on
    a live model repacks (LayerNorm fold off: the fold needs the raw activation as an fp16 operand) and switching back
    restores the fp16 mode's bits."""
    d = dev()
    from dynamicscaler_amd.synth import synth_normal
    z = np.load(os.path.join(G, "unet_tiny_i2v.npz"))
    params = json.loads(bytes(z["params_json"]).decode())
    m = build_unet(params, 5, d, torch.float32)
    x0, c0 = T(z["x_0"]), T(z["ctx_0"])
    x1, c1 = synth_normal(x0.shape, 11), synth_normal(c0.shape, 12)
    t = torch.tensor([500, 20], device=d)
    xs = torch.cat([x0, x1]).to(d, torch.float16)
    cs = torch.cat([c0, c1]).to(d)
    both = m(xs, t, context=cs, fps=8)
    a = m(xs[:1], t[:1], context=cs[:1], fps=8)
    b = m(xs[1:], t[1:], context=cs[1:], fps=8)
    assert torch.equal(both[:1], a) and torch.equal(both[1:], b)
    # shared CFG prefix
    t2 = t[:1].expand(2).contiguous()
    x2 = torch.cat([xs[:1], xs[:1]])
    plain = m(x2, t2, context=cs, fps=8)
    shared = m(x2, t2, context=cs, fps=8, cfg_pairs=1)
    assert torch.equal(plain, shared) and not torch.equal(plain[:1], plain[1:])
    # attribute switch on a live model
    m16 = build_unet(params, 5, d, torch.float16)
    ref16 = m16(x2, t2, context=cs, fps=8)
    m.residual_dtype, m.residual_scope = torch.float16, "full"
    assert torch.equal(m(x2, t2, context=cs, fps=8), ref16)
    m.residual_dtype, m.residual_scope = torch.float32, "full"
    assert torch.equal(m(x2, t2, context=cs, fps=8), plain)
    assert not torch.equal(plain, ref16)


def test_unet_full_size_strict_vs_reference_golden():
    """Full t2v UNet at the real tile in both modes against the reference's fp32 CPU forward, and the latent after CFG 7.5 +
    one update of the 50-step schedule at index 25 and at index 49 (the first step of every metric run, where the update
    multiplies the guided-eps error most)."""
    from oracle import ddim as oddim
    from dynamicscaler_amd import ops
    from dynamicscaler_amd.synth import synth_normal
    d = dev()
    z = np.load(os.path.join(G, "unet_full_t2v.npz"))
    params = t2v_params()
    x = T(z["x"])
    ctx = torch.cat([synth_normal((1, 77, 1024), 1), synth_normal((1, 77, 1024), 2)])
    ec, eu = T(z["eps_cond"]), T(z["eps_uncond"])
    sched = oddim.DDIMSchedule(oddim.DiffusionTables(), 50)
    out = {}
    m = build_unet(params, 0, d, torch.float16)
    for rd, scope in ((torch.float16, "full"), (torch.float32, "outer"), (torch.float32, "full")):
        m.residual_dtype, m.residual_scope = rd, scope
        eps = m(torch.cat([x, x]).to(d, torch.float16), torch.tensor([int(z["t"])] * 2, device=d), context=ctx.to(d),
                fps=int(z["fps"]))
        r = dict(test="full_strict", residual=str(rd).split(".")[1] + ("/outer" if scope == "outer" else ""), eps_cond=relerr(eps[:1], ec), eps_uncond=relerr(eps[1:], eu))
        r["e_t"] = relerr(eps[1:] + 7.5 * (eps[:1] - eps[1:]), oddim.cfg_combine(ec, eu, 7.5))
        for index in (25, 49):
            rxp, rx0 = oddim.ddim_step(sched, x, oddim.cfg_combine(ec, eu, 7.5), [index] * 16, noise=torch.zeros_like(x))
            xp, x0 = ops.cfg_ddim(x.to(d, torch.float16), eps[:1].contiguous(), eps[1:].contiguous(), (1, 4, 16, 40, 64), 7.5,
                                  sched.step_coefficients(index))
            r[f"x_prev_{index}"], r[f"pred_x0_{index}"] = relerr(xp, rxp), relerr(x0, rx0)
        print(r)
        record(**r)
        out[(rd, scope)] = r
    o = out[(torch.float32, "outer")]          # fp32 between the blocks only: measured 1.2e-3 (emulated 1.17e-3)
    assert o["eps_cond"] < 2.4e-3 and o["eps_cond"] < 0.85 * out[(torch.float16, "full")]["eps_cond"]
    out = {torch.float16: out[(torch.float16, "full")], torch.float32: out[(torch.float32, "full")]}
    s = out[torch.float32]
    assert s["eps_cond"] < EPS_TOL_STRICT and s["eps_uncond"] < EPS_TOL_STRICT
    assert s["eps_cond"] < 0.8 * out[torch.float16]["eps_cond"]
    assert s["x_prev_25"] < 1e-3


def test_strict_twin_equals_a_model_in_the_strict_mode_and_the_policy_calibrates():
    """Round 6: `UNetModel.twin("strict")` / `forward(..., precision="strict")` -- the operand policy's first rung -- is the strict residual
    mode over the SAME parameters: bit for bit the eps of a model built in that mode, for a plain batch and for a CFG pair batch; the
    model's own forward is untouched by the twin's existence.  And the pipelines' calibration (one wide + one own-mode + one strict
    evaluation pair on the caller's UNet) measures what a direct comparison measures, caches it on the UNet, and orders the modes."""
    from dynamicscaler_amd.host_model import LatentDiffusionHost, SyntheticConditioner
    from dynamicscaler_amd.pipelines import VC2_Pipeline_T2V_SpherePano, OPERAND_SAFETY
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    from dynamicscaler_amd.unet_spec import param_shapes
    from dynamicscaler_amd.synth import synth_state_dict, synth_normal
    d = dev()
    z = np.load(os.path.join(G, "unet_tiny_t2v.npz"))
    params = json.loads(bytes(z["params_json"]).decode())
    own = build_unet(params, 5, d, torch.float32)
    own.residual_scope = "outer"                       # the library default mode
    strict = build_unet(params, 5, d, torch.float32)   # scope "full" = strict
    x = T(z["x_0"]).to(d, torch.float16)
    t, ctx, fps = T(z["t_0"]).to(d), T(z["ctx_0"]).to(d), int(z["fps_0"])
    e_own = own(x, t, context=ctx, fps=fps).clone()
    e_tw = own(x, t, context=ctx, fps=fps, precision="strict")
    assert torch.equal(e_tw, strict(x, t, context=ctx, fps=fps)) and not torch.equal(e_tw, e_own)
    assert torch.equal(own(x, t, context=ctx, fps=fps), e_own)            # the own mode is what it was
    x2, t2, c2 = torch.cat([x, x]), torch.cat([t, t]), torch.cat([ctx, ctx.flip(1)])
    assert torch.equal(own(x2, t2, context=c2, fps=fps, cfg_pairs=1, precision="strict"), strict(x2, t2, context=c2, fps=fps, cfg_pairs=1))
    assert torch.equal(strict(x, t, context=ctx, fps=fps, precision="strict"), strict(x, t, context=ctx, fps=fps))   # already strict: itself
    with pytest.raises(ValueError):
        own(x, t, context=ctx, fps=fps, precision="bf16")
    # the pipeline's calibration on this UNet
    ld = LatentDiffusionHost({"params": params}, conditioner=SyntheticConditioner(77, params["context_dim"]))
    ld.model.diffusion_model.load_state_dict(synth_state_dict(param_shapes(params), 5), strict=True)
    ld = ld.to(d)
    pipe = VC2_Pipeline_T2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld, rng_mode="device"), {"params": {"unet_config": {"params": params}}}).to(d, torch.float32)
    geom = dict(height=64, width=128, frames=4, total_w=256, total_h=96, num_windows_w=2, num_windows_h=2, num_windows_f=1, loop_step=4)
    st = pipe.ring_begin(prompt="a prompt", fps=8, guidance_scale=7.5, init_panorama_latent=synth_normal((1, 4, 4, 12, 32), 3).to(d),
                         num_inference_steps=10, **geom)
    cal = dict(pipe._calibration)
    assert set(cal) == {"f32outer", "f32"} and cal["f32outer"][0] == int(st.timesteps[0])
    assert 0 < cal["f32"][1] < cal["f32outer"][1] < 5e-2, cal             # the strict stream is closer to the wide result than the default one
    assert abs(pipe.guided_eps_error("f32outer", cal["f32outer"][0]) - cal["f32outer"][1] / OPERAND_SAFETY) < 1e-12
    rep = pipe.operand_report(10, 7.5)
    assert rep["guided_eps_err_measured"]["f32outer"]["err"] == cal["f32outer"][1] and rep["guided_eps_err_table"] is None
    assert not set(rep["strict_steps"]) & set(rep["wide_steps"])
    # a second pipeline over the same UNet finds the figures on it: no evaluation is repeated
    u = ld.model.diffusion_model
    n_cached = len(u._operand_calibration)
    pipe2 = VC2_Pipeline_T2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld, rng_mode="device"), {"params": {"unet_config": {"params": params}}}).to(d, torch.float32)
    pipe2.ring_begin(prompt="a prompt", fps=8, guidance_scale=7.5, init_panorama_latent=synth_normal((1, 4, 4, 12, 32), 3).to(d),
                     num_inference_steps=10, **geom)
    assert len(u._operand_calibration) == n_cached and pipe2._calibration == cal
    record(test="operand_calibration_toy", measured={k: v[1] for k, v in cal.items()}, strict_steps=rep["strict_steps"], wide_steps=rep["wide_steps"])
