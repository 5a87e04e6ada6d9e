#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE ITSELF on CPU.

Run in the build container only (needs /root/reference):   python tests/golden/make_golden.py [--full]

The reference (sh-Lin/DynamicScaler) has no tests or golden vectors of its own,
so parity is pinned by what this script captures from the imported reference code:

  G1 ring_segments.json      get_dimension_slices_and_sizes          utils/shift_window_utils.py:14-38
  G2 ring_latent.npz         RingLatent get/set with wrap in F,H,W   utils/shift_window_utils.py:40-206
  G3 mix.npz                 mix_latents_with_mask (3-D and 5-D)     utils/tensor_utils.py:19-39
  G4 scheduler.npz           lvdm_DDIM_Scheduler tables, ddim_step, re_noise   pipeline/scheduler.py:18-110
  G8 unet_tiny_{t2v,i2v}.npz UNetModel forward, toy config           lvdm/modules/networks/openaimodel3d.py:657-708
  G9 loops_small.npz         t2v basic_sample + ring loops, toy geometry, tiny UNet and fake eps
     loop_traces.json        window coordinates + final-panorama SHA-256 for BASELINE configs 2/3/5 (fake eps)
  G11 loops_grid_i2v.npz     non-overlapping grid loop (P4) and i2v ring loop (P3: round() placement, temporal
                             windows + docking, 5-D mask, merge-prev, per-window image tokens), fake eps + tiny UNet
  G12 sphere.npz             _get_uv index maps (gen_pano_360 view set, one view at all 10 theta offsets), sphere
                             gather/scatter round trips (duplicate winners), t2v sphere loop (P5), fake eps + tiny UNet
  G10 unet_full_{t2v,i2v}.npz (--full) full-size t2v / i2v UNet eps at tile [1,4,16,40,64] (3 forwards, ~3 min)
  G18 unet_t24.npz           (--full) the UNet at T = 24 (config 5): toy config and the real t2v UNet, one forward each
  G19 loops_multiprompt.npz  per-window prompt selection (window_multi_prompt_dict) on the toy dock geometry
  G20 loop_trace_cfg4_i2v.json  BASELINE config 4 geometry through the reference's i2v ring loop (fake eps): trace + SHA-256
  G21 sphere_scale.npz       t2v sphere loop with view_get_scale_factor 2 / 3 (fake eps)
  G22 sphere_i2v_scale.npz   i2v sphere loop with view_get_scale_factor 2 / 3 (fake eps; inputs = sphere_i2v.npz)
  G17 cfg1_full_t2v.npz      (--full) BASELINE config 1: basic_sample, real t2v UNet, 512x320x16f, 4 steps, CFG 7.5;
                             per-step x_t / e_t / x_prev / pred_x0 (8 forwards, ~8 min)

  G23 cfg1_50step_t2v.npz    (--full) the metric's 50-step schedule: basic_sample of the reference, real t2v UNet, one tile, free-running
                             (x_prev every 5 steps, final pred_x0) + teacher-forced updates at 10 schedule indices (120 forwards, ~2 h)
  G24 panorama_handlers.npz  PanoramaTensor / RingLatentProxy / RingPanoramaTensor / RingPanoramaLatentProxy: gets / sets / splat
  G25 ring_real_unet.npz     (--full) the t2v ring loop with the REAL UNet: 1024x512x16f, 2x2 shifted windows, 4 steps (32 forwards)
  G26 updates_i2v_t24.npz    (--full) one update of the 50-step schedule for the i2v UNet (93 tokens) and for 24-frame tiles (8 forwards)
  G27 i2v_ring_real_unet.npz (--full) the i2v ring loop with the REAL i2v UNet: per-window image tokens, merge-prev (32 forwards)

  G30 panorama_handlers_uncalled.npz  the handler methods no pipeline calls: get_view_tensor_interpolate, set_view_tensor, ring splat
  G28 ring_real_unet_50step.npz      (--full) the t2v ring loop, REAL UNet, on the metric's 50-step schedule: first and last 6 steps (96 forwards)
  G29 i2v_ring_real_unet_50step.npz  (--full) the same for the i2v ring loop with the REAL i2v UNet (96 forwards)

  G31 ring_real_unet_50step_mid.npz  (--full) the t2v ring loop, REAL UNet, steps 20..25 of 50 on cfg3's window grid cut to 2 x 2 (48 forwards)
  G32 loops_grid_shuffle.npz         random_shuffle_init_frame_stride of the grid loop, bug for bug (fake eps)
  G33 sphere_set_scale.npz           view_set_scale_factor 2 / 3 and downsample_factor_before_vae_decode of both sphere loops (fake eps, one torch thread)
  G34 i2v_ring_real_unet_50step_mid.npz  (--full) the i2v ring loop, REAL i2v UNet, steps 20..25 of 50 (48 forwards)
  G40 grid_i2v_real_unet.npz         (--full) the i2v GRID loop (i2v_normal_pipeline) with the REAL i2v UNet: 2 x 1 tiles, 4 steps (16 forwards)
  G39 sphere_i2v_real_unet.npz       (--full) the i2v SPHERE loop with the REAL i2v UNet: image tokens per view, re-noise, merge-prev, 2 steps (20 forwards)
  G38 grid_real_unet.npz             (--full) the non-overlapping shifted GRID loop with the REAL t2v UNet: 2 x 1 tiles, 4 steps (16 forwards)
  G37 sphere_real_unet.npz           (--full) the t2v SPHERE loop with the REAL UNet: 1024x512 equirect, 5 views a step, first 2 of 50 steps (20 forwards)
  G36 cfg4_real_unet_one_step.npz    (--full) BASELINE config 4's geometry (i2v, 4096x512x16f, 8x2 windows), REAL i2v UNet, step 24 of 50 (32 forwards)
  G35 cfg3_real_unet_two_steps.npz   (--full) BASELINE config 3's own geometry (4096x512x16f, 8x2 windows), REAL t2v UNet, steps 24..25 of 50 (64 forwards)

  G16 encoders_{toy,full}.npz  Resampler (the reference's module, ip_resampler.py) and the CLIP ViT-H/14 text / image
                             towers -- open_clip is absent, so the tower vectors come from transformers' CLIP
                             implementation carrying the same synthetic weights (independent anchor, not the reference)

Only data is written (inputs, expected outputs, seeds); no reference source text.
"""
import argparse
import types
import contextlib
import hashlib
import io
import json
import os
import re
import sys

import numpy as np
import torch
import torch.nn as nn
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

from oracle.ref_import import import_reference, REFERENCE_ROOT  # noqa: E402

import_reference()

from utils.shift_window_utils import RingLatent, get_dimension_slices_and_sizes  # noqa: E402  (reference)
from utils.tensor_utils import mix_latents_with_mask  # noqa: E402  (reference)
from pipeline.scheduler import lvdm_DDIM_Scheduler  # noqa: E402  (reference)
from pipeline.t2v_normal_pipeline import VC2_Pipeline_T2V  # noqa: E402  (reference)
from pipeline.t2v_sphere_panorama_pipeline import VC2_Pipeline_T2V_SpherePano  # noqa: E402  (reference)
from lvdm.models.ddpm3d import DDPM, DiffusionWrapper  # noqa: E402  (reference)
from lvdm.modules.networks.openaimodel3d import UNetModel  # noqa: E402  (reference)

from dynamicscaler_amd.unet_spec import param_shapes  # noqa: E402
from dynamicscaler_amd.synth import synth_state_dict, synth_normal  # noqa: E402

# toy UNet with the real head_dim (64) so the HIP attention kernels run it too: 64 -> 128 channels, 2 levels
TINY = dict(in_channels=4, out_channels=4, model_channels=64, attention_resolutions=[2, 1], num_res_blocks=1,
            channel_mult=[1, 2], num_head_channels=64, transformer_depth=1, context_dim=64, use_linear=True,
            use_checkpoint=True, temporal_conv=True, temporal_attention=True, temporal_selfatt_only=True,
            use_relative_position=False, use_causal_attention=False, temporal_length=4,
            addition_attention=True, fps_cond=True)


def sha(t):
    return hashlib.sha256(np.ascontiguousarray(t.detach().cpu().numpy()).tobytes()).hexdigest()


def save_npz(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrays.items()})
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


class FakeLatentDiffusion(nn.Module):
    """What the pipelines read from `pretrained_t2v`, with the reference's own
    schedule registration (DDPM.register_schedule, ddpm3d.py:113-134) and DiffusionWrapper."""

    def __init__(self, eps_module, cond_ctx, uncond_ctx, temporal_length=16):
        super().__init__()
        self.v_posterior = 0.0           # ctor defaults read by register_schedule (ddpm3d.py:60-75)
        self.parameterization = "eps"
        DDPM.register_schedule(self, given_betas=None, beta_schedule="linear", timesteps=1000,
                               linear_start=0.00085, linear_end=0.012)
        self.model = eps_module
        self.temporal_length = temporal_length
        self.uncond_type = "empty_seq"
        self.use_scale = False
        self.first_stage_model = None
        self.cond_stage_model = None
        self.channels = 4
        self._cond, self._uncond = cond_ctx, uncond_ctx

    @property
    def device(self):
        return torch.device("cpu")

    def get_learned_conditioning(self, prompts):
        return self._uncond if prompts[0] == "" else self._cond


class FakeEps(nn.Module):
    """The survey's fake eps-model: 0.1*x + 0.01*mean(ctx)."""
    diffusion_model = None

    def forward(self, x, t, c_crossattn=None, fps=None, **kw):
        return 0.1 * x + 0.01 * torch.cat(c_crossattn, 1).mean()


def build_reference_unet(params, seed):
    m = UNetModel(**params).eval()
    shapes = param_shapes(params)
    ref_shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert ref_shapes == {k: tuple(v) for k, v in shapes.items()}, "param spec != reference state dict"
    m.load_state_dict(synth_state_dict(shapes, seed), strict=True)
    return m


class WrappedUNet(nn.Module):
    """DiffusionWrapper(conditioning_key='crossattn') around an already-built reference UNetModel."""

    def __init__(self, unet):
        super().__init__()
        self.diffusion_model = unet
        self.conditioning_key = "crossattn"

    def forward(self, *a, **k):
        return DiffusionWrapper.forward(self, *a, **k)


WIN_RE = re.compile(r"window_latent: f\[(\d+) - (\d+)\] h\[(-?\d+) - (-?\d+)\] w\[(\d+) - (\d+)\]")
STEP_RE = re.compile(r"^i = (\d+)(?: => \+offset \d+ )?, t = (\d+)")


def parse_trace(text):
    steps = []
    for line in text.splitlines():
        m = STEP_RE.match(line.strip())
        if m:
            steps.append({"i": int(m.group(1)), "t": int(m.group(2)), "windows": []})
            continue
        m = WIN_RE.search(line)
        if m:
            fb, fe, top, down, left, right = map(int, m.groups())
            steps[-1]["windows"].append([left, right, top, down, fb, fe])
    return steps


def run_ring_pipeline(ld, unet_params, seed, **kw):
    sched = lvdm_DDIM_Scheduler(ld)
    pipe = VC2_Pipeline_T2V_SpherePano(ld, sched, {"params": {"unet_config": {"params": unet_params}}})
    buf = io.StringIO()
    torch.manual_seed(seed)
    with contextlib.redirect_stdout(buf):
        videos, denoised = pipe.basic_sample_shift_multi_windows(prompt="a prompt", output_type="latent", **kw)
    return denoised, parse_trace(buf.getvalue())


# ------------------------------------------------------------------------------------------------
def g1_segments():
    cases = [(456, 520, 512), (27, 67, 64), (500, 1100, 512), (0, 16, 16), (3, 19, 16), (0, 8, 16),
             (15, 17, 16), (16, 32, 16), (5, 6, 7), (6, 20, 7), (0, 40, 64), (24, 64, 64), (45, 85, 64),
             (1016, 1080, 1024), (8, 72, 512)]
    out = []
    for b, e, s in cases:
        slices, sizes = get_dimension_slices_and_sizes(b, e, s)
        out.append({"begin": b, "end": e, "size": s,
                    "slices": [[sl.start, sl.stop] for sl in slices], "sizes": list(sizes)})
    with open(os.path.join(HERE, "ring_segments.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote ring_segments.json")


def g2_ring():
    pano = synth_normal((1, 4, 6, 8, 16), seed=11)
    windows = [  # (left, right, top, down, f_begin, f_end)
        (0, 16, 0, 8, 0, 6), (3, 11, 2, 6, 1, 4), (12, 20, 5, 11, 4, 8), (15, 31, 7, 15, 5, 11),
        (0, 4, 0, 2, 0, 1), (8, 24, 0, 8, 0, 6), (1, 9, 6, 10, 3, 7), (16, 32, 8, 16, 6, 12),
    ]
    arrays = {"pano": pano, "windows": np.array(windows, dtype=np.int64)}
    handler = RingLatent(pano)
    for i, (l, r, t, d, fb, fe) in enumerate(windows):
        arrays[f"get_{i}"] = handler.get_window_latent(l, r, t, d, fb, fe)
    # scatter sequence applied cumulatively (overwrite semantics, later windows win)
    h2 = RingLatent(pano)
    for i, (l, r, t, d, fb, fe) in enumerate(windows):
        tile = synth_normal((1, 4, fe - fb, d - t, r - l), seed=100 + i)
        h2.set_window_latent(tile, l, r, t, d, fb, fe)
        arrays[f"set_after_{i}"] = h2.torch_latent.clone()
    # multi-wrap gather is legal up to 2*size (shift_window_utils.py:73-75)
    arrays["get_multiwrap"] = handler.get_window_latent(10, 32, 0, 16, 0, 12)
    save_npz("ring_latent.npz", **arrays)


def g3_mix():
    l1 = synth_normal((1, 4, 4, 8, 16), 21)
    l2 = synth_normal((1, 4, 4, 8, 16), 22)
    g = torch.Generator().manual_seed(23)
    m3 = (torch.rand((1, 8, 16), generator=g) > 0.5).float()
    m5 = (torch.rand((1, 4, 4, 8, 16), generator=g) > 0.5).float()
    arrays = {"l1": l1, "l2": l2, "m3": m3, "m5": m5}
    for r in (1, 1.0, 0.5, 0.3):
        arrays[f"out3_{r}"] = mix_latents_with_mask(l1, l2, m3, r)
        arrays[f"out5_{r}"] = mix_latents_with_mask(l1, l2, m5, r)
    save_npz("mix.npz", **arrays)


def g4_scheduler():
    ld = FakeLatentDiffusion(FakeEps(), None, None)
    arrays = {"alphas_cumprod": ld.alphas_cumprod, "betas": ld.betas}
    for n in (4, 48, 50):
        s = lvdm_DDIM_Scheduler(ld)
        with contextlib.redirect_stdout(io.StringIO()):
            s.make_schedule(n)
        arrays[f"ts_{n}"] = s.ddim_timesteps
        arrays[f"alphas_{n}"] = s.ddim_alphas
        arrays[f"alphas_prev_{n}"] = np.asarray(s.ddim_alphas_prev, dtype=np.float64)
        arrays[f"sigmas_{n}"] = np.asarray(s.ddim_sigmas, dtype=np.float64)
        arrays[f"sqrt1m_{n}"] = s.ddim_sqrt_one_minus_alphas
    # ddim_step / re_noise with the global RNG seeded (draws are recorded implicitly by the seed)
    for n, eta in ((50, 0.0), (4, 0.0), (50, 1.0)):
        s = lvdm_DDIM_Scheduler(ld)
        with contextlib.redirect_stdout(io.StringIO()):
            s.make_schedule(n, ddim_eta=eta)
        x = synth_normal((1, 4, 4, 8, 16), 31)
        e = synth_normal((1, 4, 4, 8, 16), 32)
        for index in (0, n // 2, n - 1):
            torch.manual_seed(777)
            xp, x0 = s.ddim_step(x, e, [index] * 4)
            arrays[f"step_{n}_{eta}_{index}_xprev"] = xp
            arrays[f"step_{n}_{eta}_{index}_x0"] = x0
        for (a, b) in ((0, 1), (n - 2, n - 1), (n // 2 - 1, n // 2)):
            torch.manual_seed(778)
            arrays[f"renoise_{n}_{eta}_{a}_{b}"] = s.re_noise(x, a, b)
    arrays["x"] = synth_normal((1, 4, 4, 8, 16), 31)
    arrays["e"] = synth_normal((1, 4, 4, 8, 16), 32)
    save_npz("scheduler.npz", **arrays)


def g8_unet_tiny():
    for name, img in (("t2v", False), ("i2v", True)):
        p = dict(TINY)
        p["use_image_attention"] = img
        m = build_reference_unet(p, seed=5)
        L = 77 + (16 if img else 0)
        arrays = {}
        for case, (shape, tval, fps) in enumerate([((1, 4, 4, 8, 16), 500, 8), ((1, 4, 4, 16, 8), 999, 16),
                                                   ((1, 4, 6, 8, 8), 20, 24)]):
            x = synth_normal(shape, 40 + case)
            ctx = synth_normal((1, L, 64), 50 + case)
            t = torch.tensor([tval])
            with torch.no_grad():
                eps = m(x, t, context=ctx, fps=fps)
            arrays.update({f"x_{case}": x, f"ctx_{case}": ctx, f"t_{case}": t, f"fps_{case}": np.int64(fps),
                           f"eps_{case}": eps})
        arrays["weights_sha"] = np.frombuffer(bytes.fromhex(sha(torch.cat([v.flatten() for _, v in sorted(m.state_dict().items())]))), dtype=np.uint8)
        arrays["params_json"] = np.frombuffer(json.dumps(p).encode(), dtype=np.uint8)
        save_npz(f"unet_tiny_{name}.npz", **arrays)


SMALL_GEOMS = {
    "grid4x2": dict(height=64, width=128, frames=4, total_w=512, total_h=96, num_windows_w=4, num_windows_h=2,
                    num_windows_f=1, loop_step=4, num_inference_steps=6),
    "overlapw": dict(height=64, width=128, frames=4, total_w=512, total_h=96, num_windows_w=5, num_windows_h=2,
                     num_windows_f=1, loop_step=4, num_inference_steps=5),
    "frames2": dict(height=64, width=128, frames=4, total_w=512, total_h=96, num_windows_w=4, num_windows_h=2,
                    num_windows_f=2, loop_step=4, num_inference_steps=5),
    "dock": dict(height=64, width=128, frames=4, total_w=512, total_h=96, num_windows_w=4, num_windows_h=2,
                 num_windows_f=1, loop_step=4, num_inference_steps=5, dock_at_h=True),
}


def g9_loops_small():
    cond = synth_normal((1, 77, 64), 61)
    uncond = synth_normal((1, 77, 64), 62)
    unet = build_reference_unet(dict(TINY), seed=5)
    arrays = {"cond": cond, "uncond": uncond}
    traces = {}
    for eps_name, eps_mod in (("fake", FakeEps()), ("tiny", WrappedUNet(unet))):
        ld = FakeLatentDiffusion(eps_mod, cond, uncond, temporal_length=4)
        # P1: single tile
        sched = lvdm_DDIM_Scheduler(ld)
        pipe = VC2_Pipeline_T2V(ld, sched, {"params": {"unet_config": {"params": dict(TINY)}}})
        torch.manual_seed(2333333)
        with contextlib.redirect_stdout(io.StringIO()):
            _, den = pipe.basic_sample(prompt="a prompt", height=64, width=128, frames=4, fps=8,
                                       guidance_scale=7.5, num_inference_steps=4, output_type="latent")
        arrays[f"basic_{eps_name}"] = den
        for gname, geom in SMALL_GEOMS.items():
            if eps_name == "tiny" and gname not in ("grid4x2", "overlapw"):
                continue
            den, trace = run_ring_pipeline(ld, dict(TINY), 2333333, fps=8, guidance_scale=7.5, **geom)
            arrays[f"ring_{gname}_{eps_name}"] = den
            traces[gname] = trace
    save_npz("loops_small.npz", **arrays)
    with open(os.path.join(HERE, "loops_small_traces.json"), "w") as f:
        json.dump({"geoms": SMALL_GEOMS, "traces": traces}, f)
    print("wrote loops_small_traces.json")


BASELINE_GEOMS = {
    "cfg2_2048x512": dict(height=320, width=512, frames=16, total_w=2048, total_h=512, num_windows_w=4,
                          num_windows_h=2, num_windows_f=1, loop_step=8, num_inference_steps=50),
    "cfg3_4096x512": dict(height=320, width=512, frames=16, total_w=4096, total_h=512, num_windows_w=8,
                          num_windows_h=2, num_windows_f=1, loop_step=8, num_inference_steps=50),
    "cfg3_overlap_nw10": dict(height=320, width=512, frames=16, total_w=4096, total_h=512, num_windows_w=10,
                              num_windows_h=2, num_windows_f=1, loop_step=8, num_inference_steps=10),
    "cfg5_8192x1024x24": dict(height=320, width=512, frames=24, total_w=8192, total_h=1024, num_windows_w=16,
                              num_windows_h=4, num_windows_f=1, loop_step=8, num_inference_steps=10),
}


def g9_traces():
    cond = synth_normal((1, 77, 64), 61)
    uncond = synth_normal((1, 77, 64), 62)
    ld = FakeLatentDiffusion(FakeEps(), cond, uncond)
    out = {}
    for name, geom in BASELINE_GEOMS.items():
        den, trace = run_ring_pipeline(ld, {"in_channels": 4}, 2333333, fps=8, guidance_scale=7.5, **geom)
        out[name] = {"geom": geom, "trace": trace, "denoised_sha256": sha(den), "shape": list(den.shape)}
        print(name, "tiles/step:", sorted({len(s["windows"]) for s in trace}), out[name]["denoised_sha256"][:12])
    with open(os.path.join(HERE, "loop_traces.json"), "w") as f:
        json.dump(out, f)
    print("wrote loop_traces.json")


def g10_unet_full():
    params = yaml.safe_load(open(os.path.join(REFERENCE_ROOT, "configs/inference_t2v_512_v2.0.yaml")))
    params = params["model"]["params"]["unet_config"]["params"]
    torch.set_num_threads(os.cpu_count())
    m = build_reference_unet(params, seed=0)
    x = synth_normal((1, 4, 16, 40, 64), 2333333)
    arrays = {"x": x, "t": np.int64(499), "fps": np.int64(8)}
    for name, seed in (("cond", 1), ("uncond", 2)):
        ctx = synth_normal((1, 77, 1024), seed)
        with torch.no_grad():
            eps = m(x, torch.tensor([499]), context=ctx, fps=8)
        arrays[f"eps_{name}"] = eps
        print(name, float(eps.abs().mean()), float(eps.std()))
    save_npz("unet_full_t2v.npz", **arrays)


def g10_unet_full_i2v():
    """The model gen_pano_360.py runs: the i2v 512 UNet (image cross-attention, 77 text + 16 image tokens) at tile
    [1,4,16,40,64], one forward of the reference on CPU."""
    params = yaml.safe_load(open(os.path.join(REFERENCE_ROOT, "configs/inference_i2v_512_v1.0.yaml")))
    params = params["model"]["params"]["unet_config"]["params"]
    torch.set_num_threads(os.cpu_count())
    m = build_reference_unet(params, seed=3)
    x = synth_normal((1, 4, 16, 40, 64), 2333334)
    ctx = synth_normal((1, 77 + 16, 1024), 4)
    with torch.no_grad():
        eps = m(x, torch.tensor([321]), context=ctx, fps=16)
    print("i2v full", float(eps.abs().mean()), float(eps.std()))
    save_npz("unet_full_i2v.npz", x=x, t=np.int64(321), fps=np.int64(16), ctx=ctx, eps=eps)



def g17_cfg1_full():
    """BASELINE config 1 as written: VC2_Pipeline_T2V.basic_sample (pipeline/t2v_normal_pipeline.py:69-210) with the real
    t2v yaml UNet (1.41 B parameters), one 512x320x16f tile, 4 DDIM steps, CFG 7.5 -- 8 forwards of the reference on CPU.
    The init latent is passed in (fp16-representable, `latents=`), so a fp16-latent build starts from identical numbers.
    Per step: the guided prediction e_t, x_prev and pred_x0 (the latent that went in is x_init / the previous x_prev) as the reference's own
    lvdm_DDIM_Scheduler.ddim_step (pipeline/scheduler.py:60-96) returned them."""
    params = yaml.safe_load(open(os.path.join(REFERENCE_ROOT, "configs/inference_t2v_512_v2.0.yaml")))
    params = params["model"]["params"]["unet_config"]["params"]
    torch.set_num_threads(os.cpu_count())
    unet = build_reference_unet(params, seed=0)
    cond, uncond = synth_normal((1, 77, 1024), 1), synth_normal((1, 77, 1024), 2)
    ld = FakeLatentDiffusion(WrappedUNet(unet), cond, uncond, temporal_length=16)
    sched = lvdm_DDIM_Scheduler(ld)
    pipe = VC2_Pipeline_T2V(ld, sched, {"params": {"unet_config": {"params": params}}})
    x_init = synth_normal((1, 4, 16, 40, 64), 2333333)
    rec = []
    orig_step = sched.ddim_step

    def recording_step(sample, noise_pred, indices):
        x_prev, pred_x0 = orig_step(sample=sample, noise_pred=noise_pred, indices=indices)
        rec.append((sample.clone(), noise_pred.clone(), x_prev.clone(), pred_x0.clone(), int(indices[0])))
        print(f"step {len(rec)}: index {indices[0]} |e_t| {float(noise_pred.std()):.4f} |x_prev| {float(x_prev.std()):.4f} "
              f"|x0| {float(pred_x0.std()):.4f}", flush=True)
        return x_prev, pred_x0

    sched.ddim_step = recording_step
    torch.manual_seed(2333333)
    with contextlib.redirect_stdout(io.StringIO()):
        _, den = pipe.basic_sample(prompt="a prompt", height=320, width=512, frames=16, fps=8, guidance_scale=7.5,
                                   num_inference_steps=4, output_type="latent", latents=x_init.clone())
    arrays = {"x_init": x_init, "denoised": den, "fps": np.int64(8), "guidance": np.float32(7.5),
              "timesteps": np.flip(sched.ddim_timesteps).copy()}
    for i, (x_t, e_t, x_prev, x0, idx) in enumerate(rec):
        assert torch.equal(x_t, x_init if i == 0 else rec[i - 1][2])     # x_t of step i is x_prev of step i-1: stored once
        arrays.update({f"e_t_{i}": e_t, f"x_prev_{i}": x_prev, f"pred_x0_{i}": x0, f"index_{i}": np.int64(idx)})
    assert torch.equal(den, rec[-1][3])
    save_npz("cfg1_full_t2v.npz", **arrays)


def g23_cfg1_50step():
    """The schedule the metric runs on (50 DDIM steps), one 512x320x16f tile, real t2v UNet, CFG 7.5, through the reference's
    own VC2_Pipeline_T2V.basic_sample (pipeline/t2v_normal_pipeline.py:69-210) and lvdm_DDIM_Scheduler.ddim_step
    (pipeline/scheduler.py:60-96): ONE free-running 50-step run (100 forwards of the reference on CPU) storing x_prev every
    5 steps + the final pred_x0, and -- at schedule indices TF_INDICES -- one teacher-forced update of the reference from the
    run's own x_t rounded to fp16 (so that an fp16-latent build starts from identical numbers; 2 forwards each): x_t, the
    guided e_t, x_prev; pred_x0 by SHA-256 (the oracle's ddim_step reproduces it from x_t and e_t bit for bit, G4).
    ~2 h on 8 cores.  Partial results are flushed to cfg1_50step_t2v.partial.npz as the run goes."""
    TF_INDICES = (49, 48, 47, 45, 40, 30, 25, 10, 1, 0)
    params = yaml.safe_load(open(os.path.join(REFERENCE_ROOT, "configs/inference_t2v_512_v2.0.yaml")))
    params = params["model"]["params"]["unet_config"]["params"]
    torch.set_num_threads(int(os.environ.get("GOLDEN_THREADS", os.cpu_count())))
    dry = os.environ.get("GOLDEN_DRY") == "1"          # plumbing check of this function on the toy UNet (seconds)
    out_name = "cfg1_50step_dry.npz" if dry else "cfg1_50step_t2v.npz"
    if dry:
        params = TINY
    unet = build_reference_unet(params, seed=0)
    ctx_dim = params["context_dim"]
    cond, uncond = synth_normal((1, 77, ctx_dim), 1), synth_normal((1, 77, ctx_dim), 2)
    guidance, fps, frames = 7.5, 8, (4 if dry else 16)
    ld = FakeLatentDiffusion(WrappedUNet(unet), cond, uncond, temporal_length=frames)
    sched = lvdm_DDIM_Scheduler(ld)
    pipe = VC2_Pipeline_T2V(ld, sched, {"params": {"unet_config": {"params": params}}})
    x_init = synth_normal((1, 4, frames, 8, 8) if dry else (1, 4, 16, 40, 64), 2333333)
    arrays = {"x_init": x_init, "fps": np.int64(fps), "guidance": np.float32(guidance),
              "tf_indices": np.asarray(TF_INDICES, np.int64)}
    stats = []
    orig_step = sched.ddim_step
    n_calls = [0]

    def recording_step(sample, noise_pred, indices):
        import time
        idx = int(indices[0])
        x_prev, pred_x0 = orig_step(sample=sample, noise_pred=noise_pred, indices=indices)
        stats.append((idx, float(sample.std()), float(noise_pred.std()), float(x_prev.std()), float(pred_x0.std())))
        if idx % 5 == 0:
            arrays[f"free_x_prev_{idx}"] = x_prev.clone()
        if idx == 0:
            arrays["free_pred_x0_0"] = pred_x0.clone()
        if idx in TF_INDICES:
            # teacher-forced update of the reference from the fp16-rounded x_t (same calls as basic_sample:171-201)
            x_h = sample.to(torch.float16).to(torch.float32)
            t = int(np.flip(sched.ddim_timesteps)[sched.ddim_timesteps.shape[0] - 1 - idx])
            ts = torch.full((1,), t, dtype=torch.long)
            kw = dict(fps=fps, curr_time_steps=ts, temporal_length=frames, clean_cond=True)
            e_c = ld.model(x_h, ts, c_crossattn=[cond], **kw)
            e_u = ld.model(x_h, ts, c_crossattn=[uncond], **kw)
            e_t = e_u + guidance * (e_c - e_u)
            xp, x0 = orig_step(sample=x_h, noise_pred=e_t, indices=indices)
            arrays.update({f"tf_x_t_{idx}": x_h.to(torch.float16), f"tf_e_t_{idx}": e_t, f"tf_x_prev_{idx}": xp,
                           f"tf_pred_x0_sha_{idx}": np.asarray(sha(x0)), f"tf_t_{idx}": np.int64(t),
                           f"tf_e_cond_{idx}": e_c.to(torch.float16)})
        n_calls[0] += 1
        print(f"[{time.strftime('%H:%M:%S')}] step {n_calls[0]}: index {idx} |x_t| {stats[-1][1]:.4f} |e_t| {stats[-1][2]:.4f} "
              f"|x_prev| {stats[-1][3]:.4f} |x0| {stats[-1][4]:.4f}", flush=True, file=sys.stderr)
        if idx % 5 == 0 or idx in TF_INDICES:
            save_npz(out_name.replace(".npz", ".partial.npz"), stats=np.asarray(stats, np.float64), **arrays)
        return x_prev, pred_x0

    sched.ddim_step = recording_step
    torch.manual_seed(2333333)
    with contextlib.redirect_stdout(io.StringIO()):
        _, den = pipe.basic_sample(prompt="a prompt", height=x_init.shape[3] * 8, width=x_init.shape[4] * 8, frames=frames,
                                   fps=fps, guidance_scale=guidance, num_inference_steps=50, output_type="latent", latents=x_init.clone())
    assert torch.equal(den, arrays["free_pred_x0_0"])
    arrays["timesteps"] = np.flip(sched.ddim_timesteps).copy()
    arrays["stats"] = np.asarray(stats, np.float64)          # (index, |x_t|, |e_t|, |x_prev|, |pred_x0|) per step
    save_npz(out_name, **arrays)
    os.remove(os.path.join(HERE, out_name.replace(".npz", ".partial.npz")))


def g24_panorama_handlers():
    """The four handler classes by name: PanoramaTensor (utils/panorama_tensor_utils.py:5-247), RingLatentProxy,
    RingPanoramaTensor, RingPanoramaLatentProxy (utils/ring_panorama_tensor_utils.py:8-337): seeded tensors through a fixed
    sequence of gets / sets (wrapping frame windows, duplicate scatter targets, the 4-tap splat); inputs and every result."""
    from utils.panorama_tensor_utils import PanoramaTensor
    from utils.ring_panorama_tensor_utils import RingLatentProxy, RingPanoramaTensor, RingPanoramaLatentProxy
    A = {}
    views = [(90.0, 30.0, 20.0), (120.0, -170.0, -60.0), (60.0, 0.0, 90.0)]
    A["views"] = np.asarray(views, np.float64)
    # PanoramaTensor with leading dims, without, and 2-D
    for tag, shape in (("p4", (2, 3, 16, 32)), ("p3", (3, 16, 32)), ("p2", (16, 32))):
        x = synth_normal(shape, 500 + len(shape))
        h = PanoramaTensor(x)
        A[f"{tag}_x"] = x
        for vi, (fov, th, ph) in enumerate(views):
            v, m = h.get_view_tensor_no_interpolate(fov, th, ph, 12, 10)
            A[f"{tag}_get{vi}"], A[f"{tag}_mask{vi}"] = v, m
        lead = shape[:-3] if len(shape) > 3 else (1,)
        C = shape[-3] if len(shape) >= 3 else 1
        for vi, (fov, th, ph) in enumerate(views):
            src = synth_normal((*lead, C, 10, 12), 520 + vi)
            A[f"{tag}_src{vi}"] = src
            h.set_view_tensor_no_interpolation(src, fov, th, ph)
            A[f"{tag}_after_set{vi}"] = h.equirect_tensor.clone()
        src = synth_normal((*lead, C, 10, 12), 530)
        A[f"{tag}_splat_src"] = src
        h.set_view_tensor_bilinear(src, 90.0, 45.0, -30.0)
        A[f"{tag}_after_splat"] = h.equirect_tensor.clone()
    # RingLatentProxy: windows over dim 1
    x = synth_normal((1, 5, 3, 16, 32), 540)
    r = RingLatentProxy(x)
    A["rl_x"] = x
    A["rl_win_3_8"] = r.get_window_latent(3, 8)
    A["rl_win_none"] = r.get_window_latent(None, None)
    A["rl_win_1_10"] = r.get_window_latent(1, 10)
    A["rl_shape_3_8"] = np.asarray(tuple(r.get_operating_shape(3, 8)), np.int64)
    src = synth_normal((1, 3, 3, 16, 32), 541)
    A["rl_src"] = src
    r.set_window_latent(src, 4, 7)
    A["rl_after_set"] = r.get_torch_latent().clone()
    # RingPanoramaTensor [1, N, C, H, W] and RingPanoramaLatentProxy [1, C, N, H, W]
    for tag, cls, shape, vshape in (("rp", RingPanoramaTensor, (1, 5, 3, 16, 32), lambda nf: (1, nf, 3, 10, 12)),
                                    ("rpl", RingPanoramaLatentProxy, (1, 3, 5, 16, 32), lambda nf: (1, 3, nf, 10, 12))):
        x = synth_normal(shape, 550)
        h = cls(x)
        A[f"{tag}_x"] = x
        for vi, ((fov, th, ph), (fb, fe)) in enumerate(zip(views, ((3, 7), (None, None), (4, 9)))):
            v, m = h.get_view_tensor_no_interpolate(fov, th, ph, 12, 10, frame_begin=fb, frame_end=fe)
            A[f"{tag}_get{vi}"], A[f"{tag}_mask{vi}"] = v.clone(), m
        for vi, ((fov, th, ph), (fb, fe)) in enumerate(zip(views, ((3, 7), (None, None), (4, 9)))):
            nf = 5 if fb is None else fe - fb
            src = synth_normal(vshape(nf), 560 + vi)
            A[f"{tag}_src{vi}"] = src
            h.set_view_tensor_no_interpolation(src, fov, th, ph, frame_begin=fb, frame_end=fe)
            full = h.get_equirect_tensor() if tag == "rpl" else h.equirect_tensor_handler.get_torch_latent()
            A[f"{tag}_after_set{vi}"] = full.clone()
    save_npz("panorama_handlers.npz", **A)


def g30_panorama_handlers_uncalled():
    """The handler methods no pipeline of the reference calls (SURVEY 8-a S5's list): get_view_tensor_interpolate (F.grid_sample,
    utils/panorama_tensor_utils.py:28-51, ring :31-57), set_view_tensor (round-to-nearest scatter_, :72-96, ring :80-104) and the
    ring-backed set_view_tensor_bilinear (ring :107-166), on all five classes.  A call the reference itself cannot complete is
    recorded by the name of the exception it raises."""
    from utils.panorama_tensor_utils import PanoramaTensor, PanoramaLatentProxy
    from utils.ring_panorama_tensor_utils import RingPanoramaTensor, RingPanoramaLatentProxy
    A, raised = {}, {}
    views = [(90.0, 30.0, 20.0), (120.0, -170.0, -60.0), (60.0, 0.0, 90.0)]
    modes = [("bilinear", True), ("bilinear", False), ("nearest", True)]
    A["views"] = np.asarray(views, np.float64)

    def attempt(key, fn):
        try:
            return fn()
        except Exception as e:                   # noqa: BLE001 -- the exception TYPE is the recorded behaviour
            raised[key] = type(e).__name__
            return None

    for tag, shape in (("p4", (2, 3, 16, 32)), ("p3", (3, 16, 32)), ("p2", (16, 32)), ("p5", (1, 4, 3, 16, 32))):
        x = synth_normal(shape, 600 + len(shape))
        h = PanoramaTensor(x)
        A[f"{tag}_x"] = x
        for vi, (fov, th, ph) in enumerate(views):
            for mi, (mode, ac) in enumerate(modes):
                A[f"{tag}_interp{vi}_{mi}"] = h.get_view_tensor_interpolate(fov, th, ph, 12, 10, mode, ac)
        lead = shape[:-3] if len(shape) > 3 else (1,)
        C = shape[-3] if len(shape) >= 3 else 1
        for vi, (fov, th, ph) in enumerate(views):
            src = synth_normal((*lead, C, 10, 12), 620 + vi)
            A[f"{tag}_src{vi}"] = src
            if attempt(f"{tag}_set{vi}", lambda: h.set_view_tensor(src, fov, th, ph) or True):
                A[f"{tag}_after_set{vi}"] = h.equirect_tensor.clone()
    # PanoramaLatentProxy [1, C, N, H, W]
    x = synth_normal((1, 3, 4, 16, 32), 640)
    h = PanoramaLatentProxy(x)
    A["pl_x"] = x
    for vi, (fov, th, ph) in enumerate(views):
        A[f"pl_interp{vi}"] = h.get_view_tensor_interpolate(fov, th, ph, 12, 10)
    for vi, (fov, th, ph) in enumerate(views):
        src = synth_normal((1, 3, 4, 10, 12), 645 + vi)
        A[f"pl_src{vi}"] = src
        if attempt(f"pl_set{vi}", lambda: h.set_view_tensor(src, fov, th, ph) or True):
            A[f"pl_after_set{vi}"] = h.get_equirect_tensor().clone()
    # ring-backed classes
    wins = ((3, 7), (None, None), (4, 9), (2, 3))
    for tag, cls, shape, vshape in (("rp", RingPanoramaTensor, (1, 5, 3, 16, 32), lambda nf: (1, nf, 3, 10, 12)),
                                    ("rpl", RingPanoramaLatentProxy, (1, 3, 5, 16, 32), lambda nf: (1, 3, nf, 10, 12))):
        x = synth_normal(shape, 650)
        h = cls(x)
        A[f"{tag}_x"] = x
        full = (lambda: h.get_equirect_tensor()) if tag == "rpl" else (lambda: h.equirect_tensor_handler.get_torch_latent())
        for vi, ((fov, th, ph), (fb, fe)) in enumerate(zip(views + views[:1], wins)):
            for mi, (mode, ac) in enumerate(modes[:2] if vi else modes):
                v = attempt(f"{tag}_interp{vi}_{mi}", lambda: h.get_view_tensor_interpolate(fov, th, ph, 12, 10, frame_begin=fb, frame_end=fe,
                                                                                           interpolate_mode=mode, interpolate_align_corners=ac))
                if v is not None:
                    A[f"{tag}_interp{vi}_{mi}"] = v.clone()
        for vi, ((fov, th, ph), (fb, fe)) in enumerate(zip(views + views[:1], wins)):
            nf = 5 if fb is None else fe - fb
            src = synth_normal(vshape(nf), 660 + vi)
            A[f"{tag}_src{vi}"] = src
            if attempt(f"{tag}_set{vi}", lambda: h.set_view_tensor(src, fov, th, ph, frame_begin=fb, frame_end=fe) or True):
                A[f"{tag}_after_set{vi}"] = full().clone()
        for vi, ((fov, th, ph), (fb, fe)) in enumerate(zip(views + views[:1], wins)):
            nf = 5 if fb is None else fe - fb
            src = synth_normal(vshape(nf), 670 + vi)
            A[f"{tag}_splat_src{vi}"] = src
            if attempt(f"{tag}_splat{vi}", lambda: h.set_view_tensor_bilinear(src, fov, th, ph, frame_begin=fb, frame_end=fe) or True):
                A[f"{tag}_after_splat{vi}"] = full().clone()
    A["raised_json"] = np.frombuffer(json.dumps(raised).encode(), dtype=np.uint8)
    print("raised:", raised)
    save_npz("panorama_handlers_uncalled.npz", **A)


RING_REAL = dict(height=320, width=512, frames=16, total_w=1024, total_h=512, num_windows_w=2, num_windows_h=2,
                 num_windows_f=1, loop_step=4, num_inference_steps=4)


def g25_ring_real_unet():
    """P2 end to end with the REAL t2v UNet: VC2_Pipeline_T2V_SpherePano.basic_sample_shift_multi_windows
    (pipeline/t2v_sphere_panorama_pipeline.py:316-660) on a 1024x512x16f ring panorama, 2x2 shifted windows of 512x320 (the
    H windows overlap by 40 %: re-noise under the mask; every step shifts the grid across both seams), 4 DDIM steps, CFG 7.5 --
    32 forwards of the reference on CPU (~35 min).  The init latent is passed in (fp16-representable); the re-noise draws come
    from the global CPU generator seeded with 2333333 like gen_pano_360.py does."""
    params = yaml.safe_load(open(os.path.join(REFERENCE_ROOT, "configs/inference_t2v_512_v2.0.yaml")))
    params = params["model"]["params"]["unet_config"]["params"]
    torch.set_num_threads(int(os.environ.get("GOLDEN_THREADS", os.cpu_count())))
    unet = build_reference_unet(params, seed=0)
    cond, uncond = synth_normal((1, 77, 1024), 1), synth_normal((1, 77, 1024), 2)
    ld = FakeLatentDiffusion(WrappedUNet(unet), cond, uncond, temporal_length=16)
    init = synth_normal((1, 4, 16, RING_REAL["total_h"] // 8, RING_REAL["total_w"] // 8), 2333334)
    den, trace = run_ring_pipeline(ld, params, 2333333, fps=8, guidance_scale=7.5, init_panorama_latent=init.clone(), **RING_REAL)
    save_npz("ring_real_unet.npz", init=init, denoised=den, fps=np.int64(8), guidance=np.float32(7.5))
    with open(os.path.join(HERE, "ring_real_unet_trace.json"), "w") as f:
        json.dump({"geom": RING_REAL, "trace": trace}, f)
    print("wrote ring_real_unet_trace.json", float(den.std()))


def g26_updates_i2v_and_t24():
    """One teacher-forced update of the 50-step schedule (CFG 7.5; the reference's UNet forward + lvdm_DDIM_Scheduler.ddim_step,
    pipeline/scheduler.py:60-96) for the two model / tile variants the other BASELINE configs run: the i2v UNet with 77 text + 16
    image tokens (config 4) at indices 49 and 25, and the t2v UNet on a 24-frame tile (config 5) at indices 49 and 25.
    x_t = sqrt(a_t) x0 + sqrt(1 - a_t) n with synthetic unit-scale x0, n, rounded to fp16 (both sides start from identical
    numbers).  8 forwards of the reference on CPU."""
    torch.set_num_threads(int(os.environ.get("GOLDEN_THREADS", os.cpu_count())))
    A = {"guidance": np.float32(7.5), "indices": np.asarray([49, 25], np.int64)}
    for tag, yaml_name, seed, frames, L, fps in (("i2v", "configs/inference_i2v_512_v1.0.yaml", 3, 16, 93, 16),
                                                  ("t24", "configs/inference_t2v_512_v2.0.yaml", 0, 24, 77, 8)):
        params = yaml.safe_load(open(os.path.join(REFERENCE_ROOT, yaml_name)))["model"]["params"]["unet_config"]["params"]
        unet = build_reference_unet(params, seed=seed)
        cond, uncond = synth_normal((1, L, 1024), 11), synth_normal((1, L, 1024), 12)
        ld = FakeLatentDiffusion(WrappedUNet(unet), cond, uncond, temporal_length=frames)
        sched = lvdm_DDIM_Scheduler(ld)
        sched.make_schedule(50)
        A[f"{tag}_cond"], A[f"{tag}_uncond"], A[f"{tag}_fps"] = cond.to(torch.float16), uncond.to(torch.float16), np.int64(fps)
        for idx in (49, 25):
            t = int(sched.ddim_timesteps[idx])
            a_t = float(sched.ddim_alphas[idx])
            x0, n = synth_normal((1, 4, frames, 40, 64), 700 + idx), synth_normal((1, 4, frames, 40, 64), 800 + idx)
            x_t = (a_t ** 0.5 * x0 + (1 - a_t) ** 0.5 * n).to(torch.float16).to(torch.float32)
            ts = torch.full((1,), t, dtype=torch.long)
            kw = dict(fps=fps, curr_time_steps=ts, temporal_length=frames, clean_cond=True)
            with torch.no_grad():
                e_c = ld.model(x_t, ts, c_crossattn=[cond], **kw)
                e_u = ld.model(x_t, ts, c_crossattn=[uncond], **kw)
                e_t = e_u + 7.5 * (e_c - e_u)
                xp, x0p = sched.ddim_step(sample=x_t, noise_pred=e_t, indices=[idx] * frames)
            A.update({f"{tag}_x_t_{idx}": x_t.to(torch.float16), f"{tag}_e_t_{idx}": e_t, f"{tag}_x_prev_{idx}": xp,
                      f"{tag}_t_{idx}": np.int64(t), f"{tag}_pred_x0_sha_{idx}": np.asarray(sha(x0p))})
            print(tag, idx, t, float(e_t.std()), float(xp.std()), flush=True)
    save_npz("updates_i2v_t24.npz", **A)


I2V_RING_REAL = dict(height=320, width=512, frames=16, total_w=1024, total_h=512, total_f=16, num_windows_w=2, num_windows_h=2,
                     num_windows_f=1, loop_step=4, num_inference_steps=4, overlap_ratio_list_f=[0.0] * 4,
                     merge_prev_denoised_ratio_list=[0.5, 0.4, 0.3, 0.2])


def g27_i2v_ring_real_unet():
    """P3 end to end with the REAL i2v UNet (image cross-attention, 77 text + 16 image tokens per window): the reference's
    VC2_Pipeline_I2V_SpherePano.basic_sample_shift_multi_windows (pipeline/i2v_sphere_panorama_pipeline.py:564-996) on a
    1024x512x16f ring panorama, 2x2 shifted windows, per-window image crops of a synthetic panorama image through the synthetic
    embedder, merge-prev, 4 DDIM steps, CFG 7.5 -- 32 forwards of the reference on CPU.  Init latent passed in."""
    import utils.shift_window_utils as swu
    from pipeline.i2v_sphere_panorama_pipeline import VC2_Pipeline_I2V_SpherePano
    params = yaml.safe_load(open(os.path.join(REFERENCE_ROOT, "configs/inference_i2v_512_v1.0.yaml")))["model"]["params"]["unet_config"]["params"]
    torch.set_num_threads(int(os.environ.get("GOLDEN_THREADS", os.cpu_count())))
    unet = build_reference_unet(params, seed=3)
    cond, uncond = synth_normal((1, 77, 1024), 11), synth_normal((1, 77, 1024), 12)
    embed = synth_image_embedder(1024)
    pano_img = synth_normal((3, 512, 1024), 189).clamp(-1, 1)
    init = synth_normal((1, 4, 16, 64, 128), 2333335)
    orig_loader = swu.load_image_tensor_from_path
    swu.load_image_tensor_from_path = lambda image_path, height, width, norm_to_1=True: pano_img   # I/O stub (cv2 absent)
    try:
        ld = FakeLatentDiffusion(WrappedUNet(unet), cond, uncond, temporal_length=16)
        ld.get_image_embeds = embed
        ld.embedder = object()
        pipe = VC2_Pipeline_I2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld), {"params": {"unet_config": {"params": params}}})
        pipe._load_imgs_from_paths = lambda img_path_list, height=320, width=512: pano_img[None, :, :height, :width]
        buf = io.StringIO()
        torch.manual_seed(2333333)
        with contextlib.redirect_stdout(buf):
            _, den = pipe.basic_sample_shift_multi_windows(prompt="a prompt", img_cond_path="unused.png", fps=16, guidance_scale=7.5,
                                                           pano_image_path="unused.png", output_type="latent",
                                                           init_panorama_latent=init.clone(), **I2V_RING_REAL)
    finally:
        swu.load_image_tensor_from_path = orig_loader
    save_npz("i2v_ring_real_unet.npz", init=init, denoised=den, pano_img_seed=np.int64(189), fps=np.int64(16), guidance=np.float32(7.5))
    with open(os.path.join(HERE, "i2v_ring_real_unet_trace.json"), "w") as f:
        json.dump({"geom": I2V_RING_REAL, "trace": parse_trace(buf.getvalue())}, f)
    print("wrote i2v_ring_real_unet_trace.json", float(den.std()))


# ---- the metric's 50-step schedule through the ring loops with the REAL UNets (round 4) ----------------------------------
RING50_STEPS = 6                    # steps recorded at each end of the schedule
RING50_LAST_KEPT = (0, 2, 5)        # panoramas kept of the last six steps (the first six are all kept)


class _StopRun(Exception):
    pass


def _trim16(t):
    """fp32 with the low 8 mantissa bits cleared after rounding (relative error <= 2^-17): what the per-step panoramas are
    stored as, so that the deflate stream of the .npz drops a quarter of the bytes."""
    a = np.ascontiguousarray(t.detach().cpu().numpy().astype(np.float32))
    u = a.view(np.uint32).astype(np.uint64)
    u = ((u + 0x80) & 0xFFFFFF00).astype(np.uint32)
    return u.view(np.float32)


def _record_ring_run(run, n_steps):
    """Run `run()` (a call of one of the reference's ring loops) while watching the RingLatent handlers it builds
    (utils/shift_window_utils.py:40-46): the first two are the panorama latent and its pred-x0 twin
    (t2v_sphere_panorama_pipeline.py:438-439, i2v_sphere_panorama_pipeline.py:730-731), every later one is a step's fresh
    mask (:493 / :810) -- i.e. it marks the START of a step, where the panorama after the previous step is snapshotted.
    Stops the loop at the start of step `n_steps` (None: let it finish)."""
    import time
    import utils.shift_window_utils as swu
    handlers, snaps = [], []
    orig_init = swu.RingLatent.__init__

    def init(self, *a, **k):
        orig_init(self, *a, **k)
        handlers.append(self)
        if len(handlers) > 3:
            snaps.append((handlers[0].torch_latent.clone(), handlers[1].torch_latent.clone()))
            print(f"[{time.strftime('%H:%M:%S')}] after step {len(snaps)}: |x| {float(snaps[-1][0].std()):.4f}", file=sys.stderr, flush=True)
            if n_steps is not None and len(snaps) >= n_steps:
                raise _StopRun()

    swu.RingLatent.__init__ = init
    buf = io.StringIO()
    out = None
    try:
        with contextlib.redirect_stdout(buf):
            out = run()
    except _StopRun:
        pass
    finally:
        swu.RingLatent.__init__ = orig_init
    if out is not None:                                   # ran to the end: the last step has no following mask handler
        snaps.append((handlers[0].torch_latent.clone(), handlers[1].torch_latent.clone()))
    return snaps, parse_trace(buf.getvalue()), out


def _late_latent(sched, shape, index, seed):
    """A latent at the noise level of schedule index `index`: sqrt(a) x0 + sqrt(1 - a) n, synthetic unit-scale x0 and n."""
    a = float(sched.ddim_alphas[index])
    return (a ** 0.5 * synth_normal(shape, seed) + (1 - a) ** 0.5 * synth_normal(shape, seed + 1)).half().float()


def g28_ring_real_unet_50step():
    """P2 on the schedule the metric runs: the reference's t2v ring loop (pipeline/t2v_sphere_panorama_pipeline.py:481-634)
    with the REAL t2v UNet, 1024x512x16f, 2x2 shifted windows, CFG 7.5, num_inference_steps = 50 --
    "first": steps 0..5 (schedule indices 49..44, t = 999..897) from the init latent, the loop stopped at the start of step 6;
    "last": the method's own use_skip_time=True, skip_time_step_idx=44 (:393-396): steps of indices 5..0 from a latent at index 5's
    noise level, run to the end.  Both ends re-noise the overlap at the 50-step sigmas (:550-559).  The panorama latent after
    every step (and the final pred-x0 panorama) are stored; 2 x 48 forwards of the reference on CPU."""
    params = yaml.safe_load(open(os.path.join(REFERENCE_ROOT, "configs/inference_t2v_512_v2.0.yaml")))
    params = params["model"]["params"]["unet_config"]["params"]
    torch.set_num_threads(int(os.environ.get("GOLDEN_THREADS", os.cpu_count())))
    dry = os.environ.get("GOLDEN_DRY") == "1"              # plumbing check on the toy UNet (seconds)
    if dry:
        params = dict(TINY)
    unet = build_reference_unet(params, seed=0)
    cd = params["context_dim"]
    cond, uncond = synth_normal((1, 77, cd), 1), synth_normal((1, 77, cd), 2)
    ld = FakeLatentDiffusion(WrappedUNet(unet), cond, uncond, temporal_length=16)
    geom = dict(RING_REAL, num_inference_steps=50)
    shape = (1, 4, 16, geom["total_h"] // 8, geom["total_w"] // 8)
    A = {"fps": np.int64(8), "guidance": np.float32(7.5), "steps": np.int64(RING50_STEPS)}
    traces = {}
    sched = lvdm_DDIM_Scheduler(ld)
    sched.make_schedule(50)
    inits = {"first": synth_normal(shape, 2333336), "last": _late_latent(sched, shape, RING50_STEPS - 1, 2333340)}
    for end in ("first", "last"):
        pipe = VC2_Pipeline_T2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld), {"params": {"unet_config": {"params": params}}})
        kw = dict(prompt="a prompt", output_type="latent", fps=8, guidance_scale=7.5, init_panorama_latent=inits[end].clone(), **geom)
        if end == "last":
            kw.update(use_skip_time=True, skip_time_step_idx=50 - RING50_STEPS)
        torch.manual_seed(2333333)
        snaps, trace, out = _record_ring_run(lambda: pipe.basic_sample_shift_multi_windows(**kw), RING50_STEPS if end == "first" else None)
        assert len(snaps) == RING50_STEPS and (out is None) == (end == "first")
        A[f"{end}_init"] = inits[end]
        for k, (x, x0) in enumerate(snaps):
            if end == "first" or k in RING50_LAST_KEPT:
                A[f"{end}_pano_{k}"] = _trim16(x)
        A[f"{end}_x0_{RING50_STEPS - 1}"] = _trim16(snaps[-1][1])
        if out is not None:
            assert torch.equal(out[1], snaps[-1][1])
        traces[end] = trace[:RING50_STEPS]
        save_npz("ring_real_unet_50step.partial.npz", **A)
    os.remove(os.path.join(HERE, "ring_real_unet_50step.partial.npz"))
    if dry:
        print("dry run ok", [float(np.std(A[f"last_pano_{k}"])) for k in RING50_LAST_KEPT], traces["last"][-1])
        return
    save_npz("ring_real_unet_50step.npz", **A)
    with open(os.path.join(HERE, "ring_real_unet_50step_trace.json"), "w") as f:
        json.dump({"geom": geom, "traces": traces}, f)


RING50_MID_SKIP = 20                # g31: the loop entered at step 20 of 50 (schedule indices 29..24, t = 599..497)


def g31_ring_real_unet_50step_mid():
    """P2 in the MIDDLE of the 50-step schedule, on the headline geometry's window grid: the reference's t2v ring loop
    (pipeline/t2v_sphere_panorama_pipeline.py:481-634) with the REAL t2v UNet on two columns x two rows of BASELINE config 3's
    grid (512x320 windows, loop_step = 8 like cfg3: the grid moves 1/8 window per step, so from the second recorded step on the
    right-hand column's windows straddle the W seam), entered through the method's own use_skip_time (:393-396) at step 20:
    six steps at schedule indices 29..24 from a latent at index 29's noise level; the loop is stopped at the start of the seventh.
    Panorama latent after steps 0 / 2 / 5 and the pred-x0 panorama after step 5 are stored; 48 forwards of the reference on CPU."""
    params = yaml.safe_load(open(os.path.join(REFERENCE_ROOT, "configs/inference_t2v_512_v2.0.yaml")))
    params = params["model"]["params"]["unet_config"]["params"]
    torch.set_num_threads(int(os.environ.get("GOLDEN_THREADS", os.cpu_count())))
    dry = os.environ.get("GOLDEN_DRY") == "1"
    if dry:
        params = dict(TINY)
    unet = build_reference_unet(params, seed=0)
    cd = params["context_dim"]
    cond, uncond = synth_normal((1, 77, cd), 1), synth_normal((1, 77, cd), 2)
    ld = FakeLatentDiffusion(WrappedUNet(unet), cond, uncond, temporal_length=16)
    geom = dict(RING_REAL, num_inference_steps=50, loop_step=8)
    shape = (1, 4, 16, geom["total_h"] // 8, geom["total_w"] // 8)
    sched = lvdm_DDIM_Scheduler(ld)
    sched.make_schedule(50)
    first_index = 49 - RING50_MID_SKIP
    init = _late_latent(sched, shape, first_index, 2333350)
    pipe = VC2_Pipeline_T2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld), {"params": {"unet_config": {"params": params}}})
    kw = dict(prompt="a prompt", output_type="latent", fps=8, guidance_scale=7.5, init_panorama_latent=init.clone(),
              use_skip_time=True, skip_time_step_idx=RING50_MID_SKIP, **geom)
    torch.manual_seed(2333333)
    snaps, trace, out = _record_ring_run(lambda: pipe.basic_sample_shift_multi_windows(**kw), RING50_STEPS)
    assert len(snaps) == RING50_STEPS and out is None
    A = {"fps": np.int64(8), "guidance": np.float32(7.5), "steps": np.int64(RING50_STEPS), "skip": np.int64(RING50_MID_SKIP),
         "first_index": np.int64(first_index), "init": init.numpy().astype(np.float16)}
    assert np.array_equal(A["init"].astype(np.float32), init.numpy())
    for k, (x, x0) in enumerate(snaps):
        if k in RING50_LAST_KEPT:
            A[f"pano_{k}"] = _trim16(x)
    A[f"x0_{RING50_STEPS - 1}"] = _trim16(snaps[-1][1])
    if dry:
        print("dry run ok", [float(np.std(A[f"pano_{k}"])) for k in RING50_LAST_KEPT], trace[:RING50_STEPS][-1])
        return
    save_npz("ring_real_unet_50step_mid.npz", **A)
    with open(os.path.join(HERE, "ring_real_unet_50step_mid_trace.json"), "w") as f:
        json.dump({"geom": geom, "trace": trace[:RING50_STEPS]}, f)


def g29_i2v_ring_real_unet_50step():
    """P3 on the 50-step schedule: the reference's i2v ring loop (pipeline/i2v_sphere_panorama_pipeline.py:777-970) with the REAL
    i2v UNet (77 text + 16 image tokens per window, merge-prev), same geometry and the same two ends as G28
    (use_skip_time :673-675 for the last six)."""
    import utils.shift_window_utils as swu
    from pipeline.i2v_sphere_panorama_pipeline import VC2_Pipeline_I2V_SpherePano
    params = yaml.safe_load(open(os.path.join(REFERENCE_ROOT, "configs/inference_i2v_512_v1.0.yaml")))["model"]["params"]["unet_config"]["params"]
    torch.set_num_threads(int(os.environ.get("GOLDEN_THREADS", os.cpu_count())))
    dry = os.environ.get("GOLDEN_DRY") == "1"
    if dry:
        params = dict(TINY, use_image_attention=True)
    unet = build_reference_unet(params, seed=3)
    cd = params["context_dim"]
    cond, uncond = synth_normal((1, 77, cd), 11), synth_normal((1, 77, cd), 12)
    embed = synth_image_embedder(cd)
    pano_img = synth_normal((3, 512, 1024), 189).clamp(-1, 1)
    ld = FakeLatentDiffusion(WrappedUNet(unet), cond, uncond, temporal_length=16)
    ld.get_image_embeds = embed
    ld.embedder = object()
    shape = (1, 4, 16, 64, 128)
    sched = lvdm_DDIM_Scheduler(ld)
    sched.make_schedule(50)
    inits = {"first": synth_normal(shape, 2333337), "last": _late_latent(sched, shape, RING50_STEPS - 1, 2333342)}
    merge_first = [0.5] * 50
    merge_last = [0.3, 0.3, 0.2, 0.2, 0.1, 0.1]
    A = {"fps": np.int64(16), "guidance": np.float32(7.5), "steps": np.int64(RING50_STEPS), "pano_img_seed": np.int64(189)}
    traces, geoms = {}, {}
    orig_loader = swu.load_image_tensor_from_path
    swu.load_image_tensor_from_path = lambda image_path, height, width, norm_to_1=True: pano_img   # I/O stub (cv2 absent)
    try:
        for end in ("first", "last"):
            n = 50 if end == "first" else RING50_STEPS
            geom = dict(I2V_RING_REAL, num_inference_steps=50, overlap_ratio_list_f=[0.0] * n,
                        merge_prev_denoised_ratio_list=merge_first if end == "first" else merge_last)
            pipe = VC2_Pipeline_I2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld), {"params": {"unet_config": {"params": params}}})
            pipe._load_imgs_from_paths = lambda img_path_list, height=320, width=512: pano_img[None, :, :height, :width]
            kw = dict(prompt="a prompt", img_cond_path="unused.png", fps=16, guidance_scale=7.5, pano_image_path="unused.png",
                      output_type="latent", init_panorama_latent=inits[end].clone(), **geom)
            if end == "last":
                kw.update(use_skip_time=True, skip_time_step_idx=50 - RING50_STEPS)
            torch.manual_seed(2333333)
            snaps, trace, out = _record_ring_run(lambda: pipe.basic_sample_shift_multi_windows(**kw), RING50_STEPS if end == "first" else None)
            assert len(snaps) == RING50_STEPS and (out is None) == (end == "first")
            A[f"{end}_init"] = inits[end]
            for k, (x, x0) in enumerate(snaps):
                if end == "first" or k in RING50_LAST_KEPT:
                    A[f"{end}_pano_{k}"] = _trim16(x)
            A[f"{end}_x0_{RING50_STEPS - 1}"] = _trim16(snaps[-1][1])
            traces[end], geoms[end] = trace[:RING50_STEPS], geom
            save_npz("i2v_ring_real_unet_50step.partial.npz", **A)
    finally:
        swu.load_image_tensor_from_path = orig_loader
    os.remove(os.path.join(HERE, "i2v_ring_real_unet_50step.partial.npz"))
    if dry:
        print("dry run ok", [float(np.std(A[f"last_pano_{k}"])) for k in RING50_LAST_KEPT], traces["last"][-1])
        return
    save_npz("i2v_ring_real_unet_50step.npz", **A)
    with open(os.path.join(HERE, "i2v_ring_real_unet_50step_trace.json"), "w") as f:
        json.dump({"geoms": geoms, "traces": traces}, f)


CFG3_REAL_SKIP, CFG3_REAL_STEPS = 24, 2


def g35_cfg3_real_unet_two_steps():
    """The HEADLINE geometry itself with the real UNet: the reference's t2v ring loop (pipeline/t2v_sphere_panorama_pipeline.py:481-634) on
    BASELINE config 3 -- 4096 x 512 x 16f, 8 x 2 shifted windows of 512 x 320, loop_step 8, CFG 7.5, the 50-step schedule -- entered
    through use_skip_time at step 24 (schedule indices 25, 24): two whole steps, 16 windows each (the second one shifted by 1/8 window,
    its last column wrapping across the W seam), from a latent at index 25's noise level; stopped at the start of the third.  Panorama
    latent and pred-x0 panorama after the second step (and the latent after the first); 64 forwards of the reference on CPU (~55 min)."""
    params = yaml.safe_load(open(os.path.join(REFERENCE_ROOT, "configs/inference_t2v_512_v2.0.yaml")))
    params = params["model"]["params"]["unet_config"]["params"]
    torch.set_num_threads(int(os.environ.get("GOLDEN_THREADS", os.cpu_count())))
    dry = os.environ.get("GOLDEN_DRY") == "1"
    if dry:
        params = dict(TINY)
    unet = build_reference_unet(params, seed=0)
    cd = params["context_dim"]
    cond, uncond = synth_normal((1, 77, cd), 1), synth_normal((1, 77, cd), 2)
    ld = FakeLatentDiffusion(WrappedUNet(unet), cond, uncond, temporal_length=16)
    geom = dict(height=320, width=512, frames=16, total_w=4096, total_h=512, num_windows_w=8, num_windows_h=2, num_windows_f=1, loop_step=8,
                num_inference_steps=50)
    shape = (1, 4, 16, geom["total_h"] // 8, geom["total_w"] // 8)
    sched = lvdm_DDIM_Scheduler(ld)
    sched.make_schedule(50)
    first_index = 49 - CFG3_REAL_SKIP
    init = _late_latent(sched, shape, first_index, 2333360)
    pipe = VC2_Pipeline_T2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld), {"params": {"unet_config": {"params": params}}})
    kw = dict(prompt="a prompt", output_type="latent", fps=8, guidance_scale=7.5, init_panorama_latent=init.clone(),
              use_skip_time=True, skip_time_step_idx=CFG3_REAL_SKIP, **geom)
    torch.manual_seed(2333333)
    snaps, trace, out = _record_ring_run(lambda: pipe.basic_sample_shift_multi_windows(**kw), CFG3_REAL_STEPS)
    assert len(snaps) == CFG3_REAL_STEPS and out is None
    A = {"fps": np.int64(8), "guidance": np.float32(7.5), "steps": np.int64(CFG3_REAL_STEPS), "skip": np.int64(CFG3_REAL_SKIP),
         "first_index": np.int64(first_index), "init": init.numpy().astype(np.float16)}
    assert np.array_equal(A["init"].astype(np.float32), init.numpy())
    for k, (x, x0) in enumerate(snaps):
        A[f"pano_{k}"] = _trim16(x)
    A[f"x0_{CFG3_REAL_STEPS - 1}"] = _trim16(snaps[-1][1])
    if dry:
        print("dry run ok", [float(np.std(A[f"pano_{k}"])) for k in range(CFG3_REAL_STEPS)], len(trace[0]["windows"]), trace[1]["windows"][-1])
        return
    save_npz("cfg3_real_unet_two_steps.npz", **A)
    with open(os.path.join(HERE, "cfg3_real_unet_two_steps_trace.json"), "w") as f:
        json.dump({"geom": geom, "trace": trace[:CFG3_REAL_STEPS]}, f)


CFG5_CHAIN_SKIP, CFG5_CHAIN_STEPS = 24, 2


def g41_cfg5_chain_real_unet():
    """Config 5's DEPENDENCY SHAPE with the real UNet: the reference's t2v ring loop (pipeline/t2v_sphere_panorama_pipeline.py:437-476,
    481-634) on BASELINE config 5's window grid cut to two columns -- 1024 x 1024 x 24f, 2 x 4 shifted windows of 512 x 320 x 24 frames
    (num_windows_h = 4: chains of FOUR vertically overlapping, dependent tiles per column; the UNet runs at T = 24,
    lvdm/modules/networks/openaimodel3d.py:657-708), loop_step 8, CFG 7.5, the 50-step schedule -- entered through use_skip_time at
    step 24 (schedule indices 25, 24): two whole steps of 8 windows; the loop's own counter restarts at 0, so the first step has no
    offset and the second is shifted by 1/8 window step in W and H: its bottom row (top 3 + 3 x 29 = 90, 40 rows of a 128-row latent)
    wraps across the H seam.  Panorama latent after each step and the pred-x0 panorama after the second; 32 forwards of the reference
    at T = 24 on CPU."""
    params = yaml.safe_load(open(os.path.join(REFERENCE_ROOT, "configs/inference_t2v_512_v2.0.yaml")))
    params = params["model"]["params"]["unet_config"]["params"]
    torch.set_num_threads(int(os.environ.get("GOLDEN_THREADS", os.cpu_count())))
    dry = os.environ.get("GOLDEN_DRY") == "1"
    if dry:
        params = dict(TINY)
    unet = build_reference_unet(params, seed=0)
    cd = params["context_dim"]
    cond, uncond = synth_normal((1, 77, cd), 1), synth_normal((1, 77, cd), 2)
    ld = FakeLatentDiffusion(WrappedUNet(unet), cond, uncond, temporal_length=24)
    geom = dict(height=320, width=512, frames=24, total_w=1024, total_h=1024, num_windows_w=2, num_windows_h=4, num_windows_f=1, loop_step=8,
                num_inference_steps=50)
    shape = (1, 4, 24, geom["total_h"] // 8, geom["total_w"] // 8)
    sched = lvdm_DDIM_Scheduler(ld)
    sched.make_schedule(50)
    first_index = 49 - CFG5_CHAIN_SKIP
    init = _late_latent(sched, shape, first_index, 2333370)
    pipe = VC2_Pipeline_T2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld), {"params": {"unet_config": {"params": params}}})
    kw = dict(prompt="a prompt", output_type="latent", fps=8, guidance_scale=7.5, init_panorama_latent=init.clone(),
              use_skip_time=True, skip_time_step_idx=CFG5_CHAIN_SKIP, **geom)
    torch.manual_seed(2333333)
    snaps, trace, out = _record_ring_run(lambda: pipe.basic_sample_shift_multi_windows(**kw), CFG5_CHAIN_STEPS)
    assert len(snaps) == CFG5_CHAIN_STEPS and out is None
    A = {"fps": np.int64(8), "guidance": np.float32(7.5), "steps": np.int64(CFG5_CHAIN_STEPS), "skip": np.int64(CFG5_CHAIN_SKIP),
         "first_index": np.int64(first_index), "init": init.numpy().astype(np.float16)}
    assert np.array_equal(A["init"].astype(np.float32), init.numpy())
    for k, (x, x0) in enumerate(snaps):
        A[f"pano_{k}"] = _trim16(x)
    A[f"x0_{CFG5_CHAIN_STEPS - 1}"] = _trim16(snaps[-1][1])
    if dry:
        print("dry run ok", [float(np.std(A[f"pano_{k}"])) for k in range(CFG5_CHAIN_STEPS)], len(trace[0]["windows"]), trace[1]["windows"])
        return
    save_npz("cfg5_chain_real_unet.npz", **A)
    with open(os.path.join(HERE, "cfg5_chain_real_unet_trace.json"), "w") as f:
        json.dump({"geom": geom, "trace": trace[:CFG5_CHAIN_STEPS]}, f)


def g36_cfg4_real_unet_one_step():
    """BASELINE config 4's geometry with the real i2v UNet: the reference's i2v ring loop (pipeline/i2v_sphere_panorama_pipeline.py:777-970)
    on 4096 x 512 x 16f, 8 x 2 shifted windows, per-window 16 image tokens from the crop of a (synthetic) 4096 x 512 panorama image under
    each window, merge-prev 0.4, CFG 7.5, the 50-step schedule entered through use_skip_time at step 24 (schedule index 25): ONE whole step
    of 16 windows from a latent at that noise level; stopped at the start of the second.  32 forwards of the reference on CPU (~27 min)."""
    import utils.shift_window_utils as swu
    from pipeline.i2v_sphere_panorama_pipeline import VC2_Pipeline_I2V_SpherePano
    params = yaml.safe_load(open(os.path.join(REFERENCE_ROOT, "configs/inference_i2v_512_v1.0.yaml")))["model"]["params"]["unet_config"]["params"]
    torch.set_num_threads(int(os.environ.get("GOLDEN_THREADS", os.cpu_count())))
    dry = os.environ.get("GOLDEN_DRY") == "1"
    if dry:
        params = dict(TINY, use_image_attention=True)
    unet = build_reference_unet(params, seed=3)
    cd = params["context_dim"]
    cond, uncond = synth_normal((1, 77, cd), 11), synth_normal((1, 77, cd), 12)
    embed = synth_image_embedder(cd)
    pano_img = synth_normal((3, 512, 4096), 190).clamp(-1, 1)
    ld = FakeLatentDiffusion(WrappedUNet(unet), cond, uncond, temporal_length=16)
    ld.get_image_embeds = embed
    ld.embedder = object()
    shape = (1, 4, 16, 64, 512)
    sched = lvdm_DDIM_Scheduler(ld)
    sched.make_schedule(50)
    first_index = 49 - CFG3_REAL_SKIP
    init = _late_latent(sched, shape, first_index, 2333370)
    n = 50 - CFG3_REAL_SKIP
    geom = dict(height=320, width=512, frames=16, total_w=4096, total_h=512, total_f=16, num_windows_w=8, num_windows_h=2, num_windows_f=1,
                loop_step=8, num_inference_steps=50, overlap_ratio_list_f=[0.0] * n, merge_prev_denoised_ratio_list=[0.4] + [0.0] * (n - 1))
    orig_loader = swu.load_image_tensor_from_path
    swu.load_image_tensor_from_path = lambda image_path, height, width, norm_to_1=True: pano_img   # I/O stub (cv2 absent)
    try:
        pipe = VC2_Pipeline_I2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld), {"params": {"unet_config": {"params": params}}})
        pipe._load_imgs_from_paths = lambda img_path_list, height=320, width=512: pano_img[None, :, :height, :width]
        kw = dict(prompt="a prompt", img_cond_path="unused.png", fps=16, guidance_scale=7.5, pano_image_path="unused.png",
                  output_type="latent", init_panorama_latent=init.clone(), use_skip_time=True, skip_time_step_idx=CFG3_REAL_SKIP, **geom)
        torch.manual_seed(2333333)
        snaps, trace, out = _record_ring_run(lambda: pipe.basic_sample_shift_multi_windows(**kw), 1)
    finally:
        swu.load_image_tensor_from_path = orig_loader
    assert len(snaps) == 1 and out is None
    A = {"fps": np.int64(16), "guidance": np.float32(7.5), "steps": np.int64(1), "skip": np.int64(CFG3_REAL_SKIP),
         "first_index": np.int64(first_index), "pano_img_seed": np.int64(190), "init": init.numpy().astype(np.float16),
         "pano_0": _trim16(snaps[0][0]), "x0_0": _trim16(snaps[0][1])}
    assert np.array_equal(A["init"].astype(np.float32), init.numpy())
    if dry:
        print("dry run ok", float(np.std(A["pano_0"])), len(trace[0]["windows"]))
        return
    save_npz("cfg4_real_unet_one_step.npz", **A)
    with open(os.path.join(HERE, "cfg4_real_unet_one_step_trace.json"), "w") as f:
        json.dump({"geom": geom, "trace": trace[:1]}, f)


SPHERE_REAL_GEOM = dict(height=320, width=512, frames=16, equirect_width=1024, equirect_height=512, view_fov=120, loop_step_theta=4,
                        phi_theta_dict={"0": [0, 90, 180, 270], "60": [45]}, merge_renoised_overlap_latent_ratio=None,
                        num_inference_steps=50, denoise_to_step=2)


def g37_sphere_real_unet():
    """P5 with the REAL UNet: the reference's t2v sphere loop (pipeline/t2v_sphere_panorama_pipeline.py:23-312) on a 1024 x 512 equirect
    (latent 64 x 128), five overlapping perspective views of 512 x 320 x 16f per step (fov 120; phi 0: theta 0 / 90 / 180 / 270, phi 60:
    theta 45), theta offset walking (loop_step_theta 4), CFG 7.5, the first two steps of the 50-step schedule (denoise_to_step = 2) from a
    given init latent.  merge_renoised_overlap_latent_ratio = None: no re-noise draw in the loop (the reference's randn_like on a strided
    view takes a host-dependent path), so the panoramas can be compared across hosts.  20 forwards of the reference on CPU."""
    params = yaml.safe_load(open(os.path.join(REFERENCE_ROOT, "configs/inference_t2v_512_v2.0.yaml")))
    params = params["model"]["params"]["unet_config"]["params"]
    torch.set_num_threads(int(os.environ.get("GOLDEN_THREADS", os.cpu_count())))
    dry = os.environ.get("GOLDEN_DRY") == "1"
    if dry:
        params = dict(TINY)
    unet = build_reference_unet(params, seed=0)
    cd = params["context_dim"]
    cond, uncond = synth_normal((1, 77, cd), 1), synth_normal((1, 77, cd), 2)
    ld = FakeLatentDiffusion(WrappedUNet(unet), cond, uncond, temporal_length=16)
    g = dict(SPHERE_REAL_GEOM)
    g["phi_theta_dict"] = {int(k): v for k, v in g["phi_theta_dict"].items()}
    init = synth_normal((1, 4, 16, 64, 128), 2333380).half().float()
    pipe = VC2_Pipeline_T2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld), {"params": {"unet_config": {"params": params}}})
    torch.manual_seed(2333333)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        final, den = pipe.basic_sample_shift_shpere_panorama(prompt="a prompt", fps=8, guidance_scale=7.5, output_type="latent",
                                                             init_sphere_latent=init.clone(), **g)
    A = {"fps": np.int64(8), "guidance": np.float32(7.5), "init": init.numpy().astype(np.float16), "final": _trim16(final), "denoised": _trim16(den)}
    assert np.array_equal(A["init"].astype(np.float32), init.numpy())
    if dry:
        print("dry run ok", float(final.std()), float(den.std()), float((den == 0).float().mean()))
        return
    save_npz("sphere_real_unet.npz", **A)
    with open(os.path.join(HERE, "sphere_real_unet.json"), "w") as f:
        json.dump({"geom": SPHERE_REAL_GEOM}, f)


GRID_REAL_GEOM = dict(num_windows_w=2, num_windows_h=1, num_windows_f=1, loop_step=4, num_inference_steps=4,
                      skip_time_step_idx=0)        # (printed unconditionally by the reference, t2v_normal_pipeline.py:306)
I2V_SPHERE_REAL_GEOM = dict(height=320, width=512, frames=16, total_f=16, equirect_width=1024, equirect_height=512, view_fov=120, loop_step_theta=4,
                            phi_theta_dict={"0": [0, 90, 180, 270], "60": [45]}, merge_renoised_overlap_latent_ratio=1,
                            merge_prev_denoised_ratio_list=[0.3, 0.2] + [0.0] * 48, overlap_ratio_list_f=[0.0] * 50, loop_step_frame=8,
                            num_inference_steps=50, denoise_to_step=2)


def g38_grid_real_unet():
    """P4 with the REAL UNet: the reference's non-overlapping shifted grid loop (pipeline/t2v_normal_pipeline.py:213-568) on a 2 x 1 grid of
    512 x 320 x 16f tiles (1024 x 320), loop_step 4 (the grid moves a quarter tile per step, wrapping in W, H and F), CFG 7.5, config 1's
    4-step schedule, from a given init latent; 16 forwards of the reference on CPU."""
    params = yaml.safe_load(open(os.path.join(REFERENCE_ROOT, "configs/inference_t2v_512_v2.0.yaml")))
    params = params["model"]["params"]["unet_config"]["params"]
    torch.set_num_threads(int(os.environ.get("GOLDEN_THREADS", os.cpu_count())))
    dry = os.environ.get("GOLDEN_DRY") == "1"
    if dry:
        params = dict(TINY)
    unet = build_reference_unet(params, seed=0)
    cd = params["context_dim"]
    cond, uncond = synth_normal((1, 77, cd), 1), synth_normal((1, 77, cd), 2)
    ld = FakeLatentDiffusion(WrappedUNet(unet), cond, uncond, temporal_length=16)
    init = synth_normal((1, 4, 16, 40, 128), 2333390).half().float()
    pipe = VC2_Pipeline_T2V(ld, lvdm_DDIM_Scheduler(ld), {"params": {"unet_config": {"params": params}}})
    torch.manual_seed(2333333)
    with contextlib.redirect_stdout(io.StringIO()):
        _, den = pipe.basic_sample_shift_multi_windows(prompt="a prompt", height=320, width=512, frames=16, fps=8, guidance_scale=7.5,
                                                       output_type="latent", init_panorama_latent=init.clone(), **GRID_REAL_GEOM)
    A = {"fps": np.int64(8), "guidance": np.float32(7.5), "init": init.numpy().astype(np.float16), "denoised": _trim16(den)}
    assert np.array_equal(A["init"].astype(np.float32), init.numpy())
    if dry:
        print("dry run ok", tuple(den.shape), float(den.std()))
        return
    save_npz("grid_real_unet.npz", **A)
    with open(os.path.join(HERE, "grid_real_unet.json"), "w") as f:
        json.dump({"geom": GRID_REAL_GEOM}, f)


def g39_i2v_sphere_real_unet():
    """P5 (i2v) with the REAL i2v UNet: the reference's i2v sphere loop (pipeline/i2v_sphere_panorama_pipeline.py:31-495) on a 1024 x 512
    equirect, five overlapping 512 x 320 x 16f views a step, 16 image tokens per view from its perspective crop of a (synthetic) panorama
    image, overlap re-noise at ratio 1 (a contiguous view here: torch's plain randn stream), merge-prev 0.3 / 0.2, CFG 7.5, the first two
    steps of the 50-step schedule; 20 forwards of the reference on CPU."""
    import pipeline.i2v_sphere_panorama_pipeline as mod
    from pipeline.i2v_sphere_panorama_pipeline import VC2_Pipeline_I2V_SpherePano
    params = yaml.safe_load(open(os.path.join(REFERENCE_ROOT, "configs/inference_i2v_512_v1.0.yaml")))["model"]["params"]["unet_config"]["params"]
    torch.set_num_threads(int(os.environ.get("GOLDEN_THREADS", os.cpu_count())))
    dry = os.environ.get("GOLDEN_DRY") == "1"
    if dry:
        params = dict(TINY, use_image_attention=True)
    unet = build_reference_unet(params, seed=3)
    cd = params["context_dim"]
    cond, uncond = synth_normal((1, 77, cd), 11), synth_normal((1, 77, cd), 12)
    embed = synth_image_embedder(cd)
    pano_img = synth_normal((3, 512, 1024), 191).clamp(-1, 1)
    ld = FakeLatentDiffusion(WrappedUNet(unet), cond, uncond, temporal_length=16)
    ld.get_image_embeds = embed
    ld.embedder = object()
    g = dict(I2V_SPHERE_REAL_GEOM)
    g["phi_theta_dict"] = {int(k): v for k, v in g["phi_theta_dict"].items()}
    init = synth_normal((1, 4, 16, 64, 128), 2333395).half().float()
    orig_loader = mod.load_image_tensor_from_path
    mod.load_image_tensor_from_path = lambda image_path, height, width, norm_to_1=True: pano_img   # I/O stub (cv2 absent)
    try:
        pipe = VC2_Pipeline_I2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld), {"params": {"unet_config": {"params": params}}})
        pipe._load_imgs_from_paths = lambda img_path_list, height=320, width=512: pano_img[None, :, :height, :width]
        torch.manual_seed(2333333)
        with contextlib.redirect_stdout(io.StringIO()):
            final, den = pipe.basic_sample_shift_shpere_panorama(prompt="a prompt", img_cond_path="unused.png", fps=8, guidance_scale=7.5,
                                                                 pano_image_path="unused.png", output_type="latent",
                                                                 init_sphere_latent=init.clone(), **g)
    finally:
        mod.load_image_tensor_from_path = orig_loader
    A = {"fps": np.int64(8), "guidance": np.float32(7.5), "pano_img_seed": np.int64(191), "init": init.numpy().astype(np.float16),
         "final": _trim16(final), "denoised": _trim16(den)}
    assert np.array_equal(A["init"].astype(np.float32), init.numpy())
    if dry:
        print("dry run ok", float(final.std()), float(den.std()))
        return
    save_npz("sphere_i2v_real_unet.npz", **A)
    with open(os.path.join(HERE, "sphere_i2v_real_unet.json"), "w") as f:
        json.dump({"geom": I2V_SPHERE_REAL_GEOM}, f)


I2V_GRID_REAL_GEOM = dict(height=320, width=512, frames=16, num_windows_w=2, num_windows_h=1, num_windows_f=1, loop_step=4, num_inference_steps=4)


def g40_i2v_grid_real_unet():
    """P4 (i2v) with the REAL i2v UNet: the reference's non-overlapping shifted grid loop of the i2v base class
    (pipeline/i2v_normal_pipeline.py:68-425) on 2 x 1 tiles of 512 x 320 x 16f (1024 x 320), loop_step 4, per-window image tokens from the
    crop of the (synthetic) panorama image under the shifted window, 0/1-mask re-noise, CFG 7.5, 4-step schedule, given init latent;
    16 forwards of the reference on CPU."""
    import utils.shift_window_utils as swu
    from pipeline.i2v_normal_pipeline import VC2_Pipeline_I2V as RefI2V
    params = yaml.safe_load(open(os.path.join(REFERENCE_ROOT, "configs/inference_i2v_512_v1.0.yaml")))["model"]["params"]["unet_config"]["params"]
    torch.set_num_threads(int(os.environ.get("GOLDEN_THREADS", os.cpu_count())))
    dry = os.environ.get("GOLDEN_DRY") == "1"
    if dry:
        params = dict(TINY, use_image_attention=True)
    unet = build_reference_unet(params, seed=3)
    cd = params["context_dim"]
    cond, uncond = synth_normal((1, 77, cd), 11), synth_normal((1, 77, cd), 12)
    embed = synth_image_embedder(cd)
    grid_img = synth_normal((3, 320, 1024), 192).clamp(-1, 1)
    ld = FakeLatentDiffusion(WrappedUNet(unet), cond, uncond, temporal_length=16)
    ld.get_image_embeds = embed
    ld.embedder = object()
    init = synth_normal((1, 4, 16, 40, 128), 2333398).half().float()
    orig_loader = swu.load_image_tensor_from_path
    swu.load_image_tensor_from_path = lambda image_path, height, width, norm_to_1=True: grid_img
    try:
        pipe = RefI2V(ld, lvdm_DDIM_Scheduler(ld), {"params": {"unet_config": {"params": params}}})
        pipe._load_imgs_from_paths = lambda img_path_list, height=320, width=512: grid_img[None, :, :height, :width]
        torch.manual_seed(2333333)
        with contextlib.redirect_stdout(io.StringIO()), torch.no_grad():
            _, den = pipe.basic_sample_shift_multi_windows(prompt="a prompt", img_cond_path="unused.png", fps=8, guidance_scale=7.5,
                                                           pano_image_path="unused.png", output_type="latent",
                                                           init_panorama_latent=init.clone(), **I2V_GRID_REAL_GEOM)
    finally:
        swu.load_image_tensor_from_path = orig_loader
    A = {"fps": np.int64(8), "guidance": np.float32(7.5), "grid_img_seed": np.int64(192), "init": init.numpy().astype(np.float16),
         "denoised": _trim16(den)}
    assert np.array_equal(A["init"].astype(np.float32), init.numpy())
    if dry:
        print("dry run ok", tuple(den.shape), float(den.std()))
        return
    save_npz("grid_i2v_real_unet.npz", **A)
    with open(os.path.join(HERE, "grid_i2v_real_unet.json"), "w") as f:
        json.dump({"geom": I2V_GRID_REAL_GEOM}, f)


def g34_i2v_ring_real_unet_50step_mid():
    """P3 in the MIDDLE of the 50-step schedule (the i2v counterpart of g31): the reference's i2v ring loop
    (pipeline/i2v_sphere_panorama_pipeline.py:777-970) with the REAL i2v UNet -- 77 text + 16 image tokens per window, merge-prev --
    on g29's 1024 x 512 x 16f panorama with loop_step = 8, entered through use_skip_time (:673-675) at step 20: six steps at schedule
    indices 29..24 from a latent at index 29's noise level, stopped at the start of the seventh.  Panorama latent after steps 0 / 2 / 5
    and the pred-x0 panorama after step 5; 48 forwards of the reference on CPU."""
    import utils.shift_window_utils as swu
    from pipeline.i2v_sphere_panorama_pipeline import VC2_Pipeline_I2V_SpherePano
    params = yaml.safe_load(open(os.path.join(REFERENCE_ROOT, "configs/inference_i2v_512_v1.0.yaml")))["model"]["params"]["unet_config"]["params"]
    torch.set_num_threads(int(os.environ.get("GOLDEN_THREADS", os.cpu_count())))
    dry = os.environ.get("GOLDEN_DRY") == "1"
    if dry:
        params = dict(TINY, use_image_attention=True)
    unet = build_reference_unet(params, seed=3)
    cd = params["context_dim"]
    cond, uncond = synth_normal((1, 77, cd), 11), synth_normal((1, 77, cd), 12)
    embed = synth_image_embedder(cd)
    pano_img = synth_normal((3, 512, 1024), 189).clamp(-1, 1)
    ld = FakeLatentDiffusion(WrappedUNet(unet), cond, uncond, temporal_length=16)
    ld.get_image_embeds = embed
    ld.embedder = object()
    shape = (1, 4, 16, 64, 128)
    sched = lvdm_DDIM_Scheduler(ld)
    sched.make_schedule(50)
    first_index = 49 - RING50_MID_SKIP
    init = _late_latent(sched, shape, first_index, 2333351)
    n = 50 - RING50_MID_SKIP
    geom = dict(I2V_RING_REAL, num_inference_steps=50, loop_step=8, overlap_ratio_list_f=[0.0] * n,
                merge_prev_denoised_ratio_list=[0.4, 0.4, 0.3, 0.3, 0.2, 0.2] + [0.0] * (n - 6))
    orig_loader = swu.load_image_tensor_from_path
    swu.load_image_tensor_from_path = lambda image_path, height, width, norm_to_1=True: pano_img   # I/O stub (cv2 absent)
    try:
        pipe = VC2_Pipeline_I2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld), {"params": {"unet_config": {"params": params}}})
        pipe._load_imgs_from_paths = lambda img_path_list, height=320, width=512: pano_img[None, :, :height, :width]
        kw = dict(prompt="a prompt", img_cond_path="unused.png", fps=16, guidance_scale=7.5, pano_image_path="unused.png",
                  output_type="latent", init_panorama_latent=init.clone(), use_skip_time=True, skip_time_step_idx=RING50_MID_SKIP, **geom)
        torch.manual_seed(2333333)
        snaps, trace, out = _record_ring_run(lambda: pipe.basic_sample_shift_multi_windows(**kw), RING50_STEPS)
    finally:
        swu.load_image_tensor_from_path = orig_loader
    assert len(snaps) == RING50_STEPS and out is None
    A = {"fps": np.int64(16), "guidance": np.float32(7.5), "steps": np.int64(RING50_STEPS), "skip": np.int64(RING50_MID_SKIP),
         "first_index": np.int64(first_index), "pano_img_seed": np.int64(189), "init": init.numpy().astype(np.float16)}
    assert np.array_equal(A["init"].astype(np.float32), init.numpy())
    for k, (x, x0) in enumerate(snaps):
        if k in RING50_LAST_KEPT:
            A[f"pano_{k}"] = _trim16(x)
    A[f"x0_{RING50_STEPS - 1}"] = _trim16(snaps[-1][1])
    if dry:
        print("dry run ok", [float(np.std(A[f"pano_{k}"])) for k in RING50_LAST_KEPT], trace[:RING50_STEPS][-1])
        return
    save_npz("i2v_ring_real_unet_50step_mid.npz", **A)
    with open(os.path.join(HERE, "i2v_ring_real_unet_50step_mid_trace.json"), "w") as f:
        json.dump({"geom": geom, "trace": trace[:RING50_STEPS]}, f)


def g18_unet_t24(full=False):
    """BASELINE config 5 runs the UNet at T = 24 (`frames=24`, t2v_sphere_panorama_pipeline.py:411 -> UNetModel.forward with
    a 24-frame tile, openaimodel3d.py:657-708): one forward of the reference at T = 24, toy config and (--full) the real
    t2v UNet at tile [1,4,24,40,64]."""
    arrays = {}
    p = dict(TINY)
    m = build_reference_unet(p, seed=5)
    x = synth_normal((1, 4, 24, 8, 16), 70)
    ctx = synth_normal((1, 77, 64), 71)
    with torch.no_grad():
        arrays.update(tiny_x=x, tiny_ctx=ctx, tiny_t=np.int64(640), tiny_fps=np.int64(8),
                      tiny_eps=m(x, torch.tensor([640]), context=ctx, fps=8))
    arrays["tiny_params_json"] = np.frombuffer(json.dumps(p).encode(), dtype=np.uint8)
    if full:
        params = yaml.safe_load(open(os.path.join(REFERENCE_ROOT, "configs/inference_t2v_512_v2.0.yaml")))
        params = params["model"]["params"]["unet_config"]["params"]
        torch.set_num_threads(os.cpu_count())
        m = build_reference_unet(params, seed=0)
        x = synth_normal((1, 4, 24, 40, 64), 2333335)
        ctx = synth_normal((1, 77, 1024), 1)
        with torch.no_grad():
            eps = m(x, torch.tensor([777]), context=ctx, fps=8)
        print("t24 full", float(eps.abs().mean()), float(eps.std()))
        arrays.update(full_x=x, full_t=np.int64(777), full_fps=np.int64(8), full_eps=eps)
    save_npz("unet_t24.npz", **arrays)



def g19_multi_prompt():
    """R13: per-window prompt selection (`window_multi_prompt_dict`, t2v_sphere_panorama_pipeline.py:561-566,
    utils/multi_prompt_utils.py:1-7) on the toy `dock` geometry (dock_at_h keeps every window's lower edge inside the
    panorama; without it the reference's own factor assert fires as soon as a window wraps in H -- recorded as
    `grid4x2_raises`).  Prompts map to seeded embeddings; fake eps and the tiny UNet."""
    prompts = {"a prompt": synth_normal((1, 77, 64), 61), "": synth_normal((1, 77, 64), 62),
               "sky": synth_normal((1, 77, 64), 63), "ground": synth_normal((1, 77, 64), 64)}

    class PromptLD(FakeLatentDiffusion):
        def get_learned_conditioning(self, p):
            return prompts[p[0]]

    mp = {0.7: "sky", 1.0: "ground"}
    unet = build_reference_unet(dict(TINY), seed=5)
    arrays = {f"emb_{k or 'empty'}".replace(" ", "_"): v for k, v in prompts.items()}
    out = {"multi_prompt_dict": {str(k): v for k, v in mp.items()}, "geom": "dock"}
    for eps_name, eps_mod in (("fake", FakeEps()), ("tiny", WrappedUNet(unet))):
        ld = PromptLD(eps_mod, prompts["a prompt"], prompts[""], temporal_length=4)
        den, trace = run_ring_pipeline(ld, dict(TINY), 2333333, fps=8, guidance_scale=7.5, window_multi_prompt_dict=mp,
                                       **SMALL_GEOMS["dock"])
        arrays[f"ring_dock_multiprompt_{eps_name}"] = den
        out["trace"] = trace
    ld = PromptLD(FakeEps(), prompts["a prompt"], prompts[""], temporal_length=4)
    try:
        run_ring_pipeline(ld, dict(TINY), 2333333, fps=8, guidance_scale=7.5, window_multi_prompt_dict=mp, **SMALL_GEOMS["grid4x2"])
        out["grid4x2_raises"] = None
    except AssertionError as e:
        out["grid4x2_raises"] = str(e)
    save_npz("loops_multiprompt.npz", **arrays)
    with open(os.path.join(HERE, "loops_multiprompt.json"), "w") as f:
        json.dump(out, f)
    print("multi-prompt:", out["grid4x2_raises"])



CFG4_GEOM = dict(height=320, width=512, frames=16, total_w=4096, total_h=512, total_f=16, num_windows_w=8, num_windows_h=2,
                 num_windows_f=1, loop_step=8, num_inference_steps=8, overlap_ratio_list_f=[0.0] * 8,
                 merge_prev_denoised_ratio_list=[0.5, 0.45, 0.4, 0.35, 0.3, 0.25, 0.2, 0.15])


def g20_cfg4_geometry():
    """BASELINE config 4 (i2v_sphere_panorama 4096x512x16f, 8x2 windows): the reference's i2v overlapped-ring loop
    (i2v_sphere_panorama_pipeline.py:564-996) at the full-size geometry with the fake eps-model, a synthetic panorama image
    (input/pano_surfing_1.png is absent from the reference tree) and the synthetic image embedder (77 + 16 = 93 tokens per
    window): window trace + SHA-256 of the final pred-x0 panorama, 8 of the 50 steps' shifted-window sequence."""
    import utils.shift_window_utils as swu
    from pipeline.i2v_sphere_panorama_pipeline import VC2_Pipeline_I2V_SpherePano
    cond, uncond = synth_normal((1, 77, 64), 61), synth_normal((1, 77, 64), 62)
    embed = synth_image_embedder(64)
    pano_img = synth_normal((3, 512, 4096), 188).clamp(-1, 1)
    orig_loader = swu.load_image_tensor_from_path
    swu.load_image_tensor_from_path = lambda image_path, height, width, norm_to_1=True: pano_img   # I/O stub (cv2 absent)
    try:
        ld = FakeLatentDiffusion(FakeEps(), cond, uncond, temporal_length=16)
        ld.get_image_embeds = embed
        ld.embedder = object()
        pipe = VC2_Pipeline_I2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld), {"params": {"unet_config": {"params": {"in_channels": 4}}}})
        pipe._load_imgs_from_paths = lambda img_path_list, height=320, width=512: pano_img[None, :, :height, :width]
        buf = io.StringIO()
        torch.manual_seed(2333333)
        with contextlib.redirect_stdout(buf):
            _, den = pipe.basic_sample_shift_multi_windows(prompt="a prompt", img_cond_path="unused.png", fps=8, guidance_scale=7.5,
                                                           pano_image_path="unused.png", output_type="latent", **CFG4_GEOM)
    finally:
        swu.load_image_tensor_from_path = orig_loader
    out = {"geom": CFG4_GEOM, "trace": parse_trace(buf.getvalue()), "denoised_sha256": sha(den), "shape": list(den.shape),
           "pano_img_seed": 188, "embedder_dim": 64}
    with open(os.path.join(HERE, "loop_trace_cfg4_i2v.json"), "w") as f:
        json.dump(out, f)
    print("cfg4 i2v ring: tiles/step", sorted({len(s["windows"]) for s in out["trace"]}), out["denoised_sha256"][:12])



SPHERE_SCALE_GEOMS = {
    # view_SET_scale_factor != 1 is deliberately absent: the reference's scatter `pano_flat[:, idx] = view` then has several
    # sources per target NEXT to each other, torch's CPU index_put_ splits the index range over threads, and which duplicate
    # wins at a chunk boundary depends on thread timing -- the reference itself is not repeatable there (a couple of elements
    # per scatter differ from "last source wins" with 8 threads, none with torch.set_num_threads(1)).
    "get2": dict(height=64, width=128, frames=4, equirect_width=512, equirect_height=256, view_fov=120, loop_step_theta=4,
                 phi_theta_dict={"60": [0, 180], "0": [0, 120, 240], "-60": [90, 270]}, merge_renoised_overlap_latent_ratio=1,
                 num_inference_steps=4, view_get_scale_factor=2),
    "get3_fov": dict(height=64, width=128, frames=4, equirect_width=512, equirect_height=256, view_fov=120, loop_step_theta=3,
                     phi_theta_dict={"45": [0, 120, 240], "-45": [60, 180, 300]}, phi_fov_dict={"45": 100},
                     merge_renoised_overlap_latent_ratio=0.6, num_inference_steps=3, view_get_scale_factor=3),
}


def g21_sphere_view_scale():
    """view_get_scale_factor of the t2v sphere loop (t2v_sphere_panorama_pipeline.py:45,194-203): the view gathered at 2x / 3x
    the tile size and resized back with 'nearest'.  Fake eps, fp32."""
    cond, uncond = synth_normal((1, 77, 64), 61), synth_normal((1, 77, 64), 62)
    arrays = {"cond": cond, "uncond": uncond}
    ld = FakeLatentDiffusion(FakeEps(), cond, uncond, temporal_length=4)
    for gname, geom in SPHERE_SCALE_GEOMS.items():
        g = dict(geom)
        g["phi_theta_dict"] = {int(k): v for k, v in g["phi_theta_dict"].items()}
        if "phi_fov_dict" in g:
            g["phi_fov_dict"] = {int(k): v for k, v in g["phi_fov_dict"].items()}
        pipe = VC2_Pipeline_T2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld), {"params": {"unet_config": {"params": dict(TINY)}}})
        torch.manual_seed(2333333)
        with contextlib.redirect_stdout(io.StringIO()):
            final, den = pipe.basic_sample_shift_shpere_panorama(prompt="a prompt", fps=8, guidance_scale=7.5, output_type="latent", **g)
        arrays[f"sphere_{gname}_final"], arrays[f"sphere_{gname}_denoised"] = final, den
    save_npz("sphere_scale.npz", **arrays)
    with open(os.path.join(HERE, "sphere_scale.json"), "w") as f:
        json.dump({"geoms": SPHERE_SCALE_GEOMS}, f)


GRID_GEOMS = {
    "plain": dict(num_windows_w=4, num_windows_h=2, num_windows_f=1, loop_step=4, num_inference_steps=5),
    "jump": dict(num_windows_w=4, num_windows_h=2, num_windows_f=2, loop_step=4, num_inference_steps=5,
                 shift_jump_odd_w=True, shift_jump_odd_h=True, shift_jump_odd_f=True),
    "dock": dict(num_windows_w=3, num_windows_h=2, num_windows_f=2, loop_step=4, num_inference_steps=6,
                 docking_w=True, docking_h=True, docking_f=True, docking_step_range=[1, 2, 4]),
}
# the pre-denoise start and the per-step residual merge of the same method (t2v_normal_pipeline.py:345-412, 445-468)
GRID_PRE_GEOMS = {
    "pre_sparse": dict(num_windows_w=2, num_windows_h=2, num_windows_f=1, loop_step=4, num_inference_steps=5,
                       use_pre_denoise=True, pre_denoise_steps=2, merge_predenoise_ratio_list=[0.9, 0.8, 0.7, 0.6, 0.5]),
    "pre_dense_skip": dict(num_windows_w=2, num_windows_h=2, num_windows_f=1, loop_step=4, num_inference_steps=6,
                           use_pre_denoise=True, pre_denoise_steps=3, skip_steps_after_pre_denoise=1, use_skip_time=True,
                           skip_time_step_idx=2, merge_predenoise_ratio_list=[0.9, 0.8, 0.7, 0.6, 0.5],
                           sparse_add_residual=False),
    "pre_progressive": dict(num_windows_w=2, num_windows_h=1, num_windows_f=1, loop_step=4, num_inference_steps=5,
                            use_pre_denoise=True, pre_denoise_steps=2, use_skip_time=True, skip_time_step_idx=3,
                            progressive_skip=True),
    "clear": dict(num_windows_w=2, num_windows_h=2, num_windows_f=1, loop_step=4, num_inference_steps=5,
                  use_pre_denoise=True, pre_denoise_steps=1, clear_seed=71),
}
I2V_GEOMS = {
    "ring": dict(height=64, width=128, frames=4, total_w=512, total_h=96, total_f=4, num_windows_w=4, num_windows_h=2,
                 num_windows_f=1, loop_step=4, num_inference_steps=5, overlap_ratio_list_f=[0.0] * 5,
                 merge_prev_denoised_ratio_list=[0.5, 0.4, 0.3, 0.2, 0.1]),
    "round_frames": dict(height=64, width=128, frames=4, total_w=512, total_h=96, total_f=8, num_windows_w=5,
                         num_windows_h=2, num_windows_f=2, loop_step=4, num_inference_steps=5, begin_index_offset=1,
                         overlap_ratio_list_f=[0.5, 0.5, 0.25, 0.5, 0.5], loop_step_frame=2, dock_at_f=True,
                         merge_prev_denoised_ratio_list=[0.5, 0.4, 0.3, 0.2, 0.1]),
    # the way gen_pano_360.py:291-312 resumes from the previous stage: given init latent + use_skip_time (schedule cut)
    "skip": dict(height=64, width=128, frames=4, total_w=512, total_h=96, total_f=4, num_windows_w=4, num_windows_h=2,
                 num_windows_f=1, loop_step=4, num_inference_steps=5, overlap_ratio_list_f=[0.0] * 5,
                 merge_prev_denoised_ratio_list=[0.5, 0.4, 0.3, 0.2, 0.1], use_skip_time=True, skip_time_step_idx=2,
                 progressive_skip=False, init_seed=93),
}


I2V_GRID_GEOMS = {    # VC2_Pipeline_I2V.basic_sample_shift_multi_windows (i2v_normal_pipeline.py:68-425)
    "plain": dict(height=64, width=128, frames=4, num_windows_w=2, num_windows_h=2, num_windows_f=1, loop_step=4,
                  num_inference_steps=5),
    "dock": dict(height=64, width=128, frames=4, num_windows_w=2, num_windows_h=2, num_windows_f=2, loop_step=2,
                 num_inference_steps=4, dock_at_h=True, merge_renoised_overlap_latent_ratio=0.6),
    "skip": dict(height=64, width=128, frames=4, num_windows_w=2, num_windows_h=2, num_windows_f=1, loop_step=4,
                 num_inference_steps=5, use_skip_time=True, skip_time_step_idx=2, progressive_skip=False, init_seed=94),
}


GRID_SHUFFLE_GEOM = dict(num_windows_w=4, num_windows_h=2, num_windows_f=2, loop_step=4, num_inference_steps=5,
                         shift_jump_odd_w=True, shift_jump_odd_h=True, shift_jump_odd_f=True, random_shuffle_init_frame_stride=2)


def g32_grid_random_shuffle():
    """random_shuffle_init_frame_stride of the non-overlapping grid loop (pipeline/t2v_normal_pipeline.py:328-337): the init latent's
    slices shuffled with Python's global `random` (seeded here; the reference indexes dim 3 with its frame indices).  Fake eps, fp32."""
    import random
    cond, uncond = synth_normal((1, 77, 64), 61), synth_normal((1, 77, 64), 62)
    ld = FakeLatentDiffusion(FakeEps(), cond, uncond, temporal_length=4)
    pipe = VC2_Pipeline_T2V(ld, lvdm_DDIM_Scheduler(ld), {"params": {"unet_config": {"params": dict(TINY)}}})
    torch.manual_seed(2333333)
    random.seed(4242)
    with contextlib.redirect_stdout(io.StringIO()):
        _, den = pipe.basic_sample_shift_multi_windows(prompt="a prompt", height=64, width=128, frames=4, fps=8, guidance_scale=7.5,
                                                       output_type="latent", skip_time_step_idx=0, **GRID_SHUFFLE_GEOM)
    torch.manual_seed(2333333)
    with contextlib.redirect_stdout(io.StringIO()):
        _, plain = pipe.basic_sample_shift_multi_windows(prompt="a prompt", height=64, width=128, frames=4, fps=8, guidance_scale=7.5,
                                                         output_type="latent", skip_time_step_idx=0,
                                                         **dict(GRID_SHUFFLE_GEOM, random_shuffle_init_frame_stride=0))
    assert not torch.equal(den, plain)
    save_npz("loops_grid_shuffle.npz", cond=cond, uncond=uncond, denoised=den, random_seed=np.int64(4242),
             geom_json=np.frombuffer(json.dumps(GRID_SHUFFLE_GEOM).encode(), dtype=np.uint8))


def synth_image_embedder(dim, tokens=16, seed=77):
    """Deterministic stand-in for get_image_embeds (CLIP image encoder + Resampler are out of scope): a 4x4 average
    pool of the crop projected 3 -> dim by a fixed seeded matrix.  Same function in tests/helpers.py."""
    proj = synth_normal((3, dim), seed)

    def embed(batch_imgs):
        pooled = torch.nn.functional.adaptive_avg_pool2d(batch_imgs.float(), (4, 4))     # [b,3,4,4]
        return pooled.flatten(2).transpose(1, 2) @ proj                            # [b,16,dim]
    return embed


def g11_grid_and_i2v():
    import utils.shift_window_utils as swu
    from pipeline.i2v_sphere_panorama_pipeline import VC2_Pipeline_I2V_SpherePano
    cond = synth_normal((1, 77, 64), 61)
    uncond = synth_normal((1, 77, 64), 62)
    arrays = {"cond": cond, "uncond": uncond}
    traces = {}
    # ---- P4: non-overlapping grid (t2v_normal_pipeline.py:213-568) ----
    unet = build_reference_unet(dict(TINY), seed=5)
    for eps_name, eps_mod in (("fake", FakeEps()), ("tiny", WrappedUNet(unet))):
        ld = FakeLatentDiffusion(eps_mod, cond, uncond, temporal_length=4)
        for gname, geom in GRID_GEOMS.items():
            if eps_name == "tiny" and gname != "plain":
                continue
            pipe = VC2_Pipeline_T2V(ld, lvdm_DDIM_Scheduler(ld), {"params": {"unet_config": {"params": dict(TINY)}}})
            buf = io.StringIO()
            torch.manual_seed(2333333)
            with contextlib.redirect_stdout(buf):
                _, den = pipe.basic_sample_shift_multi_windows(prompt="a prompt", height=64, width=128, frames=4, fps=8,
                                                               guidance_scale=7.5, output_type="latent",
                                                               skip_time_step_idx=0,  # printed unconditionally (:306)
                                                               **geom)
            arrays[f"grid_{gname}_{eps_name}"] = den
            traces[f"grid_{gname}"] = parse_trace(buf.getvalue())
        for gname, geom in GRID_PRE_GEOMS.items():
            if eps_name == "tiny" and gname != "pre_sparse":
                continue
            gk = dict(geom)
            if "clear_seed" in gk:
                gk["clear_pre_denoised_latent"] = synth_normal((1, 4, 4, 8, 16), gk.pop("clear_seed"))
            gk.setdefault("skip_time_step_idx", 0)
            pipe = VC2_Pipeline_T2V(ld, lvdm_DDIM_Scheduler(ld), {"params": {"unet_config": {"params": dict(TINY)}}})
            torch.manual_seed(2333333)
            with contextlib.redirect_stdout(io.StringIO()):
                _, den = pipe.basic_sample_shift_multi_windows(prompt="a prompt", height=64, width=128, frames=4, fps=8,
                                                               guidance_scale=7.5, output_type="latent", **gk)
            arrays[f"gridpre_{gname}_{eps_name}"] = den
    # ---- P3: i2v overlapped ring (i2v_sphere_panorama_pipeline.py:564-996) ----
    p_i2v = dict(TINY)
    p_i2v["use_image_attention"] = True
    unet_i2v = build_reference_unet(p_i2v, seed=5)
    embed = synth_image_embedder(64)
    pano_img = synth_normal((3, 96, 512), 88).clamp(-1, 1)
    orig_loader = swu.load_image_tensor_from_path
    swu.load_image_tensor_from_path = lambda image_path, height, width, norm_to_1=True: pano_img   # I/O stub (cv2 absent)
    try:
        for eps_name, eps_mod in (("fake", FakeEps()), ("tiny", WrappedUNet(unet_i2v))):
            ld = FakeLatentDiffusion(eps_mod, cond, uncond, temporal_length=4)
            ld.get_image_embeds = embed
            ld.embedder = object()          # hasattr(pretrained_t2v, 'embedder') -> uncond gets image tokens (:652-658)
            for gname, geom in I2V_GEOMS.items():
                if eps_name == "tiny" and gname != "ring":
                    continue
                pipe = VC2_Pipeline_I2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld), {"params": {"unet_config": {"params": p_i2v}}})
                pipe._load_imgs_from_paths = lambda img_path_list, height=320, width=512: pano_img[None, :, :height, :width]
                buf = io.StringIO()
                torch.manual_seed(2333333)
                gk = dict(geom)
                if "init_seed" in gk:
                    gk["init_panorama_latent"] = synth_normal((1, 4, gk["total_f"], gk["total_h"] // 8, gk["total_w"] // 8),
                                                              gk.pop("init_seed"))
                with contextlib.redirect_stdout(buf):
                    _, den = pipe.basic_sample_shift_multi_windows(prompt="a prompt", img_cond_path="unused.png", fps=8,
                                                                   guidance_scale=7.5, pano_image_path="unused.png",
                                                                   output_type="latent", **gk)
                arrays[f"i2v_{gname}_{eps_name}"] = den
                traces[f"i2v_{gname}"] = parse_trace(buf.getvalue())
        # ---- P4 (i2v): non-overlapping grid with per-window image crops (i2v_normal_pipeline.py:68-425) ----
        from pipeline.i2v_normal_pipeline import VC2_Pipeline_I2V as RefI2V
        grid_img = synth_normal((3, 128, 256), 87).clamp(-1, 1)
        swu.load_image_tensor_from_path = lambda image_path, height, width, norm_to_1=True: grid_img
        for eps_name, eps_mod in (("fake", FakeEps()), ("tiny", WrappedUNet(unet_i2v))):
            ld = FakeLatentDiffusion(eps_mod, cond, uncond, temporal_length=4)
            ld.get_image_embeds = embed
            ld.embedder = object()
            for gname, geom in I2V_GRID_GEOMS.items():
                if eps_name == "tiny" and gname != "plain":
                    continue
                pipe = RefI2V(ld, lvdm_DDIM_Scheduler(ld), {"params": {"unet_config": {"params": p_i2v}}})
                pipe._load_imgs_from_paths = lambda img_path_list, height=320, width=512: grid_img[None, :, :height, :width]
                gk = dict(geom)
                if "init_seed" in gk:
                    gk["init_panorama_latent"] = synth_normal((1, 4, gk["frames"] * gk["num_windows_f"], gk["height"] * gk["num_windows_h"] // 8,
                                                               gk["width"] * gk["num_windows_w"] // 8), gk.pop("init_seed"))
                buf = io.StringIO()
                torch.manual_seed(2333333)
                with contextlib.redirect_stdout(buf), torch.no_grad():
                    _, den = pipe.basic_sample_shift_multi_windows(prompt="a prompt", img_cond_path="unused.png", fps=8,
                                                                   guidance_scale=7.5, pano_image_path="unused.png",
                                                                   output_type="latent", **gk)
                arrays[f"i2vgrid_{gname}_{eps_name}"] = den
                traces[f"i2vgrid_{gname}"] = parse_trace(buf.getvalue())
        arrays["grid_img"] = grid_img
    finally:
        swu.load_image_tensor_from_path = orig_loader
    arrays["pano_img"] = pano_img
    save_npz("loops_grid_i2v.npz", **arrays)
    with open(os.path.join(HERE, "loops_grid_i2v_traces.json"), "w") as f:
        json.dump({"grid_geoms": GRID_GEOMS, "grid_pre_geoms": GRID_PRE_GEOMS, "i2v_geoms": I2V_GEOMS, "i2v_grid_geoms": I2V_GRID_GEOMS, "traces": traces}, f)
    print("wrote loops_grid_i2v_traces.json")


SPHERE_GEOMS = {
    "base": dict(height=64, width=128, frames=4, equirect_width=512, equirect_height=256, view_fov=120, loop_step_theta=4,
                 phi_theta_dict={"90": [0], "-90": [0], "45": [0, 120, 240], "-45": [0, 120, 240], "0": [0, 90, 180, 270]},
                 merge_renoised_overlap_latent_ratio=1, num_inference_steps=5, denoise_to_step=4),
    "fovdict": dict(height=64, width=128, frames=4, equirect_width=512, equirect_height=256, view_fov=120, loop_step_theta=3,
                    phi_theta_dict={"90": [0], "0": [0, 120, 240], "-60": [0, 180]}, phi_fov_dict={"90": 100, "-60": 110},
                    merge_renoised_overlap_latent_ratio=0.7, num_inference_steps=4),
}
GEN_PANO_VIEWS = {90: [0], -90: [0], 75: [0, 60, 120, 180, 240, 300], -75: [0, 60, 120, 180, 240, 300],
                  60: [0, 60, 120, 180, 240, 300], -60: [0, 60, 120, 180, 240, 300], 45: [0, 60, 120, 180, 240, 300],
                  -45: [0, 60, 120, 180, 240, 300], 0: [0, 60, 120, 180, 240, 300]}   # gen_pano_360.py:444-455 shape (phi_num=6)

VIEW_RE = re.compile(r"window: phi = (-?\d+), theta = (-?\d+), .* fov = (-?\d+)")


def g12_sphere():
    from utils.panorama_tensor_utils import PanoramaTensor, PanoramaLatentProxy
    arrays = {}
    # ---- G5: index maps of the gen_pano_360 view set on a 128x64 latent panorama, view 16x8... and the real 64x40 view on 256x128
    for tag, (W, H, w, h) in {"small": (128, 64, 16, 8), "real": (256, 128, 64, 40)}.items():
        pt = PanoramaTensor(torch.zeros(1, H, W))
        gi, si = [], []
        views = []
        for phi, thetas in GEN_PANO_VIEWS.items():
            for th in thetas:
                for off in ((0, 12, 24, 36, 48, 60, 72, 84, 96, 108) if (phi == 45 and th == 60) else (0,)):
                    u, v = pt._get_uv(fov=120, theta=th + off, phi=phi, width=w, height=h)
                    u0 = torch.floor(u).long() % W
                    v0 = torch.clamp(torch.floor(v).long(), 0, H - 1)
                    gi.append((v0 * W + u0).to(torch.int32))
                    si.append((torch.floor(v).long() * W + torch.floor(u).long()).to(torch.int32))
                    views.append((phi, th + off))
        arrays[f"maps_{tag}_gather"] = torch.stack(gi)
        arrays[f"maps_{tag}_scatter"] = torch.stack(si)
        arrays[f"maps_{tag}_views"] = np.array(views, dtype=np.int32)
    # ---- G6: gather / scatter round trip with duplicate winners and untouched pixels
    pano = synth_normal((1, 4, 3, 32, 64), 91)
    arrays["rt_pano"] = pano
    for n, (fov, th, ph) in enumerate([(120, 0, 0), (120, 60, 45), (120, 300, -75), (120, 0, 90), (90, 12, -90), (120, 348, 60)]):
        p = PanoramaLatentProxy(pano)
        v, m = p.get_view_tensor_no_interpolate(fov, th, ph, 16, 8)
        tile = synth_normal((1, 4, 3, 8, 16), 200 + n)
        p.set_view_tensor_no_interpolation(tile, fov, th, ph)
        arrays[f"rt_view_{n}"] = v
        arrays[f"rt_after_{n}"] = p.get_equirect_tensor()
        arrays[f"rt_args_{n}"] = np.array([fov, th, ph], dtype=np.int32)
    # ---- S5: bilinear splat with normaliser (unused by the pipelines; the only accumulate+normalise code in the tree)
    for n, (fov, th, ph) in enumerate([(120, 0, 0), (120, 60, 45), (120, 0, 90), (100, 200, -60)]):
        p = PanoramaLatentProxy(pano)
        tile = synth_normal((1, 4, 3, 8, 16), 300 + n)
        p.set_view_tensor_bilinear(tile, fov, th, ph)
        arrays[f"splat_after_{n}"] = p.get_equirect_tensor()
        arrays[f"splat_args_{n}"] = np.array([fov, th, ph], dtype=np.int32)
    # ---- N1: resize_video_latent, the two modes gen_pano_360.py uses
    from utils.diffusion_utils import resize_video_latent
    lat = synth_normal((1, 4, 3, 16, 32), 400)
    arrays["resize_in"] = lat
    arrays["resize_nearest_x2"] = resize_video_latent(lat, 32, 64, mode="nearest")
    arrays["resize_nearest_half"] = resize_video_latent(lat, 8, 16, mode="nearest")
    arrays["resize_nearest_odd"] = resize_video_latent(lat, 20, 48, mode="nearest")
    arrays["resize_bicubic_x2"] = resize_video_latent(lat, 32, 64, mode="bicubic")
    arrays["resize_bicubic_odd"] = resize_video_latent(lat, 24, 40, mode="bicubic")
    # ---- P5 (t2v): whole sphere loop, fake eps and tiny UNet
    cond = synth_normal((1, 77, 64), 61)
    uncond = synth_normal((1, 77, 64), 62)
    arrays["cond"], arrays["uncond"] = cond, uncond
    traces = {}
    unet = build_reference_unet(dict(TINY), seed=5)
    for eps_name, eps_mod in (("fake", FakeEps()), ("tiny", WrappedUNet(unet))):
        ld = FakeLatentDiffusion(eps_mod, cond, uncond, temporal_length=4)
        for gname, geom in SPHERE_GEOMS.items():
            if eps_name == "tiny" and gname != "base":
                continue
            g = dict(geom)
            g["phi_theta_dict"] = {int(k): v for k, v in g["phi_theta_dict"].items()}
            if "phi_fov_dict" in g:
                g["phi_fov_dict"] = {int(k): v for k, v in g["phi_fov_dict"].items()}
            pipe = VC2_Pipeline_T2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld), {"params": {"unet_config": {"params": dict(TINY)}}})
            buf = io.StringIO()
            torch.manual_seed(2333333)
            with contextlib.redirect_stdout(buf):
                final, den = pipe.basic_sample_shift_shpere_panorama(prompt="a prompt", fps=8, guidance_scale=7.5,
                                                                     output_type="latent", **g)
            arrays[f"sphere_{gname}_{eps_name}_final"] = final
            arrays[f"sphere_{gname}_{eps_name}_denoised"] = den
            steps = []
            for line in buf.getvalue().splitlines():
                m = STEP_RE.match(line.strip())
                if m:
                    steps.append({"i": int(m.group(1)), "t": int(m.group(2)), "views": []})
                m = VIEW_RE.search(line)
                if m:
                    steps[-1]["views"].append([int(m.group(1)), int(m.group(2)), int(m.group(3))])
            traces[gname] = steps
    save_npz("sphere.npz", **arrays)
    with open(os.path.join(HERE, "sphere_traces.json"), "w") as f:
        json.dump({"geoms": SPHERE_GEOMS, "traces": traces}, f)
    print("wrote sphere_traces.json")


I2V_SPHERE_GEOMS = {
    # total_f == frames: one frame window; merge-prev on
    "base": dict(height=64, width=128, frames=4, total_f=4, equirect_width=512, equirect_height=256, view_fov=120,
                 loop_step_theta=4, phi_theta_dict={"90": [0], "45": [0, 120, 240], "0": [0, 90, 180, 270], "-60": [0, 180]},
                 merge_renoised_overlap_latent_ratio=1, merge_prev_denoised_ratio_list=[0.5, 0.4, 0.3, 0.2, 0.1],
                 overlap_ratio_list_f=[0.0] * 5, loop_step_frame=2, num_inference_steps=5, denoise_to_step=4),
    # temporal windows with docking (total_f = 2 * frames), ratio < 1
    "long": dict(height=64, width=128, frames=4, total_f=8, equirect_width=512, equirect_height=256, view_fov=120,
                 loop_step_theta=3, phi_theta_dict={"90": [0], "0": [0, 120, 240], "-60": [0, 180]},
                 merge_renoised_overlap_latent_ratio=0.7, merge_prev_denoised_ratio_list=[0.5, 0.3, 0.1, 0.0],
                 overlap_ratio_list_f=[0.5, 0.5, 0.25, 0.5], loop_step_frame=2, dock_at_f=True, num_inference_steps=4),
    # paste_on_static: every step the panorama is rebuilt on the re-noised static (image) latent
    "static": dict(height=64, width=128, frames=4, total_f=4, equirect_width=512, equirect_height=256, view_fov=120,
                   loop_step_theta=2, phi_theta_dict={"0": [0, 120, 240], "-60": [0, 180]},
                   merge_renoised_overlap_latent_ratio=1, merge_prev_denoised_ratio_list=[0.5, 0.3, 0.1, 0.0],
                   overlap_ratio_list_f=[0.0] * 4, loop_step_frame=2, paste_on_static=True, num_inference_steps=4),
}
FWIN_RE = re.compile(r"window_latent: f\[(-?\d+) - (-?\d+)\]")
VIEW2_RE = re.compile(r"window: phi = (-?\d+), theta = (-?\d+), prompt")


def g13_i2v_sphere():
    """P5 (i2v): basic_sample_shift_shpere_panorama of pipeline/i2v_sphere_panorama_pipeline.py:31-495 (RingPanoramaLatentProxy
    frame windows, per-view image tokens from the perspective crop of the panorama image, merge-prev, paste_on_static).
    I/O and the VAE are stubbed: the image loader returns a synthetic tensor, tiled_vae_encode_image a synthetic latent."""
    import pipeline.i2v_sphere_panorama_pipeline as mod
    from pipeline.i2v_sphere_panorama_pipeline import VC2_Pipeline_I2V_SpherePano
    cond = synth_normal((1, 77, 64), 61)
    uncond = synth_normal((1, 77, 64), 62)
    pano_img = synth_normal((3, 256, 512), 89).clamp(-1, 1)
    static_latent = synth_normal((1, 4, 1, 32, 64), 90)
    arrays = {"cond": cond, "uncond": uncond, "pano_img": pano_img, "static_latent": static_latent}
    traces = {}
    p_i2v = dict(TINY)
    p_i2v["use_image_attention"] = True
    unet_i2v = build_reference_unet(p_i2v, seed=5)
    embed = synth_image_embedder(64)
    orig_loader = mod.load_image_tensor_from_path
    mod.load_image_tensor_from_path = lambda image_path, height, width, norm_to_1=True: pano_img   # I/O stub (cv2 absent)
    try:
        for eps_name, eps_mod in (("fake", FakeEps()), ("tiny", WrappedUNet(unet_i2v))):
            ld = FakeLatentDiffusion(eps_mod, cond, uncond, temporal_length=4)
            ld.get_image_embeds = embed
            ld.embedder = object()
            for gname, geom in I2V_SPHERE_GEOMS.items():
                if eps_name == "tiny" and gname != "base":
                    continue
                g = dict(geom)
                g["phi_theta_dict"] = {int(k): v for k, v in g["phi_theta_dict"].items()}
                pipe = VC2_Pipeline_I2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld), {"params": {"unet_config": {"params": p_i2v}}})
                pipe._load_imgs_from_paths = lambda img_path_list, height=320, width=512: pano_img[None, :, :height, :width]
                pipe.tiled_vae_encode_image = lambda image_path, image_size: static_latent.clone()      # VAE stub (N2)
                buf = io.StringIO()
                torch.manual_seed(2333333)
                with contextlib.redirect_stdout(buf):
                    final, den = pipe.basic_sample_shift_shpere_panorama(prompt="a prompt", img_cond_path="unused.png", fps=8,
                                                                         guidance_scale=7.5, pano_image_path="unused.png",
                                                                         output_type="latent", **g)
                arrays[f"i2vs_{gname}_{eps_name}_final"] = final
                arrays[f"i2vs_{gname}_{eps_name}_denoised"] = den
                steps, cur_f = [], None
                for line in buf.getvalue().splitlines():
                    m = STEP_RE.match(line.strip())
                    if m:
                        steps.append({"i": int(m.group(1)), "t": int(m.group(2)), "views": []})
                    m = FWIN_RE.search(line)
                    if m:
                        cur_f = (int(m.group(1)), int(m.group(2)))
                    m = VIEW2_RE.search(line)
                    if m:
                        steps[-1]["views"].append([cur_f[0], cur_f[1], int(m.group(1)), int(m.group(2))])
                traces[gname] = steps
    finally:
        mod.load_image_tensor_from_path = orig_loader
    save_npz("sphere_i2v.npz", **arrays)
    with open(os.path.join(HERE, "sphere_i2v_traces.json"), "w") as f:
        json.dump({"geoms": I2V_SPHERE_GEOMS, "traces": traces}, f)
    print("wrote sphere_i2v_traces.json")


def g22_i2v_sphere_view_scale():
    """view_get_scale_factor of the i2v sphere loop (i2v_sphere_panorama_pipeline.py:58,330-341): the latent view gathered at
    2x / 3x the tile size and resized back with 'nearest'.  Same inputs, stubs and seeds as g13 (sphere_i2v.npz holds them);
    only the outputs are stored here.  Fake eps-model."""
    import pipeline.i2v_sphere_panorama_pipeline as mod
    from pipeline.i2v_sphere_panorama_pipeline import VC2_Pipeline_I2V_SpherePano
    cond = synth_normal((1, 77, 64), 61)
    uncond = synth_normal((1, 77, 64), 62)
    pano_img = synth_normal((3, 256, 512), 89).clamp(-1, 1)
    static_latent = synth_normal((1, 4, 1, 32, 64), 90)
    p_i2v = dict(TINY)
    p_i2v["use_image_attention"] = True
    embed = synth_image_embedder(64)
    cases = {"base_g2": dict(I2V_SPHERE_GEOMS["base"], view_get_scale_factor=2),
             "long_g3": dict(I2V_SPHERE_GEOMS["long"], view_get_scale_factor=3),
             "static_g2": dict(I2V_SPHERE_GEOMS["static"], view_get_scale_factor=2)}
    arrays = {}
    orig_loader = mod.load_image_tensor_from_path
    mod.load_image_tensor_from_path = lambda image_path, height, width, norm_to_1=True: pano_img   # I/O stub (cv2 absent)
    try:
        ld = FakeLatentDiffusion(FakeEps(), cond, uncond, temporal_length=4)
        ld.get_image_embeds = embed
        ld.embedder = object()
        for name, geom in cases.items():
            g = dict(geom)
            g["phi_theta_dict"] = {int(k): v for k, v in g["phi_theta_dict"].items()}
            pipe = VC2_Pipeline_I2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld), {"params": {"unet_config": {"params": p_i2v}}})
            pipe._load_imgs_from_paths = lambda img_path_list, height=320, width=512: pano_img[None, :, :height, :width]
            pipe.tiled_vae_encode_image = lambda image_path, image_size: static_latent.clone()      # VAE stub (N2)
            torch.manual_seed(2333333)
            with contextlib.redirect_stdout(io.StringIO()):
                final, den = pipe.basic_sample_shift_shpere_panorama(prompt="a prompt", img_cond_path="unused.png", fps=8,
                                                                     guidance_scale=7.5, pano_image_path="unused.png",
                                                                     output_type="latent", **g)
            arrays[f"{name}_final"] = final
            arrays[f"{name}_denoised"] = den
    finally:
        mod.load_image_tensor_from_path = orig_loader
    save_npz("sphere_i2v_scale.npz", **arrays)
    with open(os.path.join(HERE, "sphere_i2v_scale.json"), "w") as f:
        json.dump({"cases": cases}, f)
    print("wrote sphere_i2v_scale.json")


SPHERE_SET_SCALE_GEOMS = {
    # view_set_scale_factor s: x_prev / pred_x0 resized up by s with 'nearest' before the scatter -- s x s neighbouring sources per
    # target.  Generated with ONE torch thread (see SPHERE_SCALE_GEOMS): the reference's index_put_ is then a sequential loop and
    # the last source in row-major order of the scaled view wins.
    "set2": dict(height=64, width=128, frames=4, equirect_width=512, equirect_height=256, view_fov=120, loop_step_theta=4,
                 phi_theta_dict={"60": [0, 180], "0": [0, 120, 240], "-60": [90, 270]}, merge_renoised_overlap_latent_ratio=1,
                 num_inference_steps=4, view_set_scale_factor=2),
    "set3_get2_fov_down2": dict(height=64, width=128, frames=4, equirect_width=512, equirect_height=256, view_fov=120, loop_step_theta=3,
                                phi_theta_dict={"45": [0, 120, 240], "-45": [60, 180, 300]}, phi_fov_dict={"45": 100},
                                merge_renoised_overlap_latent_ratio=0.6, num_inference_steps=3, view_get_scale_factor=2,
                                view_set_scale_factor=3, downsample_factor_before_vae_decode=2),
}


def g33_sphere_view_set_scale():
    """view_set_scale_factor and downsample_factor_before_vae_decode of both sphere loops (t2v_sphere_panorama_pipeline.py:268-275,
    298-305; i2v_sphere_panorama_pipeline.py:421-428, 481-488), fake eps, fp32, torch.set_num_threads(1) (the reference's scatter
    resolves duplicated targets by thread timing otherwise).  i2v: g13's inputs and stubs, merge-prev off (the reference mixes a
    scaled with an unscaled tensor there and raises)."""
    import pipeline.i2v_sphere_panorama_pipeline as mod
    from pipeline.i2v_sphere_panorama_pipeline import VC2_Pipeline_I2V_SpherePano
    nthreads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        cond, uncond = synth_normal((1, 77, 64), 61), synth_normal((1, 77, 64), 62)
        arrays = {"cond": cond, "uncond": uncond}
        ld = FakeLatentDiffusion(FakeEps(), cond, uncond, temporal_length=4)
        for gname, geom in SPHERE_SET_SCALE_GEOMS.items():
            g = dict(geom)
            g["phi_theta_dict"] = {int(k): v for k, v in g["phi_theta_dict"].items()}
            if "phi_fov_dict" in g:
                g["phi_fov_dict"] = {int(k): v for k, v in g["phi_fov_dict"].items()}
            pipe = VC2_Pipeline_T2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld), {"params": {"unet_config": {"params": dict(TINY)}}})
            torch.manual_seed(2333333)
            with contextlib.redirect_stdout(io.StringIO()):
                final, den = pipe.basic_sample_shift_shpere_panorama(prompt="a prompt", fps=8, guidance_scale=7.5, output_type="latent", **g)
            arrays[f"sphere_{gname}_final"], arrays[f"sphere_{gname}_denoised"] = final, den
        # i2v
        pano_img = synth_normal((3, 256, 512), 89).clamp(-1, 1)
        static_latent = synth_normal((1, 4, 1, 32, 64), 90)
        p_i2v = dict(TINY)
        p_i2v["use_image_attention"] = True
        embed = synth_image_embedder(64)
        cases = {"long_s2": dict(I2V_SPHERE_GEOMS["long"], view_set_scale_factor=2, merge_prev_denoised_ratio_list=None),
                 "static_s2_g2_down2": dict(I2V_SPHERE_GEOMS["static"], view_set_scale_factor=2, view_get_scale_factor=2,
                                            merge_prev_denoised_ratio_list=None, downsample_factor_before_vae_decode=2)}
        orig_loader = mod.load_image_tensor_from_path
        mod.load_image_tensor_from_path = lambda image_path, height, width, norm_to_1=True: pano_img   # I/O stub (cv2 absent)
        try:
            ld = FakeLatentDiffusion(FakeEps(), cond, uncond, temporal_length=4)
            ld.get_image_embeds = embed
            ld.embedder = object()
            for name, geom in cases.items():
                g = dict(geom)
                g["phi_theta_dict"] = {int(k): v for k, v in g["phi_theta_dict"].items()}
                pipe = VC2_Pipeline_I2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld), {"params": {"unet_config": {"params": p_i2v}}})
                pipe._load_imgs_from_paths = lambda img_path_list, height=320, width=512: pano_img[None, :, :height, :width]
                pipe.tiled_vae_encode_image = lambda image_path, image_size: static_latent.clone()      # VAE stub (N2)
                torch.manual_seed(2333333)
                with contextlib.redirect_stdout(io.StringIO()):
                    final, den = pipe.basic_sample_shift_shpere_panorama(prompt="a prompt", img_cond_path="unused.png", fps=8,
                                                                         guidance_scale=7.5, pano_image_path="unused.png",
                                                                         output_type="latent", **g)
                arrays[f"i2v_{name}_final"] = final
                arrays[f"i2v_{name}_denoised"] = den
            # merge-prev with a set scale factor: what the reference raises
            g = dict(I2V_SPHERE_GEOMS["base"], view_set_scale_factor=2)
            g["phi_theta_dict"] = {int(k): v for k, v in g["phi_theta_dict"].items()}
            pipe = VC2_Pipeline_I2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld), {"params": {"unet_config": {"params": p_i2v}}})
            pipe._load_imgs_from_paths = lambda img_path_list, height=320, width=512: pano_img[None, :, :height, :width]
            raised = None
            try:
                with contextlib.redirect_stdout(io.StringIO()):
                    pipe.basic_sample_shift_shpere_panorama(prompt="a prompt", img_cond_path="unused.png", fps=8, guidance_scale=7.5,
                                                            pano_image_path="unused.png", output_type="latent", **g)
            except Exception as e:      # noqa: BLE001 -- recorded, whatever it is
                raised = type(e).__name__
        finally:
            mod.load_image_tensor_from_path = orig_loader
    finally:
        torch.set_num_threads(nthreads)
    save_npz("sphere_set_scale.npz", **arrays)
    with open(os.path.join(HERE, "sphere_set_scale.json"), "w") as f:
        json.dump({"geoms": SPHERE_SET_SCALE_GEOMS, "i2v_cases": cases, "merge_prev_with_set_scale_raises": raised}, f)
    print("wrote sphere_set_scale.json; merge-prev + set scale raised:", raised)


VAE_TINY = dict(double_z=True, z_channels=4, resolution=64, in_channels=3, out_ch=3, ch=64, ch_mult=[1, 2], num_res_blocks=1,
                attn_resolutions=[], dropout=0.0)
VAE_FULL = dict(double_z=True, z_channels=4, resolution=512, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4],
                num_res_blocks=2, attn_resolutions=[], dropout=0.0)      # configs/inference_t2v_512_v2.0.yaml:51-70


def _build_reference_vae(dd, seed, whole=False):
    """whole=False: decoder-side synthetic weights only (the fixtures of g14); whole=True: encoder + decoder (g15)."""
    from lvdm.models.autoencoder import AutoencoderKL
    from dynamicscaler_amd.vae_spec import decoder_param_shapes, vae_param_shapes
    with contextlib.redirect_stdout(io.StringIO()):
        m = AutoencoderKL(ddconfig=dict(dd), lossconfig={"target": "torch.nn.Identity"}, embed_dim=4).eval()
    full = vae_param_shapes(dd, 4)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {k: tuple(v) for k, v in full.items()}, \
        "vae_spec != reference state dict"
    shapes = full if whole else decoder_param_shapes(dd, 4)
    sd = synth_state_dict(shapes, seed)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith(("encoder.", "quant_conv.")) for k in missing)
    return m, sd


def g14_vae_decode(full=False):
    """N2 (decode side): AutoencoderKL.decode of the reference on synthetic weights -- toy config, and with --full the
    real first-stage config on one 40x64 latent frame; also decode_first_stage_2DAE's per-frame loop (ddpm3d.py:556-562)."""
    from lvdm.models.ddpm3d import LatentDiffusion
    arrays = {}
    m, _ = _build_reference_vae(VAE_TINY, seed=21)
    z = synth_normal((2, 4, 3, 8, 16), 31)                  # [B,C,T,h,w]
    with torch.no_grad():
        arrays["tiny_frame"] = m.decode(z[:, :, 0])
        holder = types.SimpleNamespace(first_stage_model=m, scale_factor=0.18215)
        arrays["tiny_video"] = LatentDiffusion.decode_first_stage_2DAE(holder, z)
    arrays["tiny_z"] = z
    arrays["tiny_dd_json"] = np.frombuffer(json.dumps(VAE_TINY).encode(), dtype=np.uint8)
    if full:
        mf, _ = _build_reference_vae(VAE_FULL, seed=22)
        zf = synth_normal((1, 4, 40, 64), 32)
        with torch.no_grad():
            out = mf.decode(zf)
        arrays["full_z"] = zf
        arrays["full_frame"] = out.to(torch.float16)        # 3x320x512: stored as fp16 to keep the fixture small
        arrays["full_dd_json"] = np.frombuffer(json.dumps(VAE_FULL).encode(), dtype=np.uint8)
        save_npz("vae_full.npz", **{k: v for k, v in arrays.items() if k.startswith("full")})
    save_npz("vae_tiny.npz", **{k: v for k, v in arrays.items() if k.startswith("tiny")})


VAE_TINY8 = dict(double_z=True, z_channels=4, resolution=64, in_channels=3, out_ch=3, ch=64, ch_mult=[1, 1, 2, 2],
                 num_res_blocks=1, attn_resolutions=[], dropout=0.0)      # 8x down like the real first stage


def g15_vae_encode(full=False):
    """N2 (encode side): AutoencoderKL.encode moments, encode_first_stage_2DAE (seeded posterior sample) and the pipeline's
    tiled_vae_encode_tensor_simple, run on the reference with synthetic weights."""
    from lvdm.models.ddpm3d import LatentDiffusion
    from pipeline.i2v_sphere_panorama_pipeline import VC2_Pipeline_I2V_SpherePano
    arrays = {}
    m, _ = _build_reference_vae(VAE_TINY8, seed=23, whole=True)
    img = synth_normal((1, 3, 2, 64, 128), 41).clamp(-1, 1)          # [B,3,F,H,W]
    holder = types.SimpleNamespace(first_stage_model=m, scale_factor=0.18215)
    holder.get_first_stage_encoding = lambda post, noise=None: LatentDiffusion.get_first_stage_encoding(holder, post, noise)
    with torch.no_grad():
        arrays["tiny8_moments"] = m.encode(img[:, :, 0]).parameters
        torch.manual_seed(77)
        arrays["tiny8_encoded"] = LatentDiffusion.encode_first_stage_2DAE(holder, img)
        holder.encode_first_stage_2DAE = lambda x: LatentDiffusion.encode_first_stage_2DAE(holder, x)
        pipe = types.SimpleNamespace(pretrained_t2v=holder, vae_scale_factor=8)
        big = synth_normal((1, 3, 1, 128, 256), 42).clamp(-1, 1)
        torch.manual_seed(78)
        arrays["tiny8_tiled"] = VC2_Pipeline_I2V_SpherePano.tiled_vae_encode_tensor_simple.__wrapped__(
            pipe, image_tensor=big, h_tile_num=4, w_tile_num=4, overlap_h=2, overlap_w=2)
    arrays["tiny8_img"], arrays["tiny8_big"] = img, big
    arrays["tiny8_dd_json"] = np.frombuffer(json.dumps(VAE_TINY8).encode(), dtype=np.uint8)
    save_npz("vae_enc_tiny.npz", **arrays)
    if full:
        mf, _ = _build_reference_vae(VAE_FULL, seed=24, whole=True)
        imgf = synth_normal((1, 3, 320, 512), 43).clamp(-1, 1)
        with torch.no_grad():
            mom = mf.encode(imgf).parameters
        save_npz("vae_enc_full.npz", full_img=imgf.to(torch.float16), full_moments=mom,
                 full_dd_json=np.frombuffer(json.dumps(VAE_FULL).encode(), dtype=np.uint8))


# ---------------------------------------------------------------------------------------------- g16: encoders (N3)
RESAMPLER_TOY = dict(dim=128, depth=2, dim_head=64, heads=2, num_queries=4, embedding_dim=192, output_dim=128, ff_mult=4)
CLIP_TOY = dict(vision=dict(image_size=56, layers=2, width=320, head_width=80, patch_size=14, mlp_ratio=4.0),
                text=dict(context_length=77, vocab_size=1000, width=128, heads=2, layers=3, mlp_ratio=4.0))


@contextlib.contextmanager
def _without_reference_stubs():
    """transformers probes optional packages with importlib.util.find_spec, which rejects the spec-less stub modules
    oracle/ref_import.py installs for the reference (torchvision, ...): hide them while transformers is imported."""
    hidden = {k: sys.modules.pop(k) for k in list(sys.modules)
              if k.split(".")[0] in ("torchvision", "cv2", "kornia", "open_clip", "imageio", "diffusers", "omegaconf",
                                     "pytorch_lightning", "xformers", "decord", "av")
              and getattr(sys.modules[k], "__spec__", None) is None}
    try:
        yield
    finally:
        sys.modules.update(hidden)


def _hf_block_map(sd, src, dst, W):
    out = {}
    for nm, sl in (("q_proj", slice(0, W)), ("k_proj", slice(W, 2 * W)), ("v_proj", slice(2 * W, 3 * W))):
        out[f"{dst}.self_attn.{nm}.weight"] = sd[f"{src}.attn.in_proj_weight"][sl]
        out[f"{dst}.self_attn.{nm}.bias"] = sd[f"{src}.attn.in_proj_bias"][sl]
    for a, b in (("attn.out_proj", "self_attn.out_proj"), ("ln_1", "layer_norm1"), ("ln_2", "layer_norm2"),
                 ("mlp.c_fc", "mlp.fc1"), ("mlp.c_proj", "mlp.fc2")):
        out[f"{dst}.{b}.weight"] = sd[f"{src}.{a}.weight"]
        out[f"{dst}.{b}.bias"] = sd[f"{src}.{a}.bias"]
    return out


def _hf_text(text, sd):
    """transformers' CLIPTextModel carrying the open_clip-keyed synthetic weights (independent implementation)."""
    with _without_reference_stubs():
        from transformers import CLIPTextConfig, CLIPTextModel
    W = text["width"]
    cfg = CLIPTextConfig(vocab_size=text["vocab_size"], hidden_size=W, intermediate_size=int(W * text["mlp_ratio"]),
                         num_hidden_layers=text["layers"], num_attention_heads=text["heads"],
                         max_position_embeddings=text["context_length"], hidden_act="gelu", layer_norm_eps=1e-5,
                         bos_token_id=0, eos_token_id=2, pad_token_id=1)
    with contextlib.redirect_stdout(io.StringIO()):
        m = CLIPTextModel(cfg).eval()
    hf = {"text_model.embeddings.token_embedding.weight": sd["model.token_embedding.weight"],
          "text_model.embeddings.position_embedding.weight": sd["model.positional_embedding"],
          "text_model.final_layer_norm.weight": sd["model.ln_final.weight"],
          "text_model.final_layer_norm.bias": sd["model.ln_final.bias"]}
    for i in range(text["layers"]):
        hf.update(_hf_block_map(sd, f"model.transformer.resblocks.{i}", f"text_model.encoder.layers.{i}", W))
    if not any(k.startswith("text_model.") for k in m.state_dict()):      # newer transformers: no wrapper prefix
        hf = {k[len("text_model."):]: v for k, v in hf.items()}
    missing, unexpected = m.load_state_dict(hf, strict=False)
    assert not unexpected and all("position_ids" in k for k in missing), (missing, unexpected)
    return m


def _hf_vision(vision, sd):
    with _without_reference_stubs():
        from transformers import CLIPVisionConfig, CLIPVisionModel
    W = vision["width"]
    cfg = CLIPVisionConfig(hidden_size=W, intermediate_size=int(W * vision["mlp_ratio"]), num_hidden_layers=vision["layers"],
                           num_attention_heads=W // vision["head_width"], image_size=vision["image_size"],
                           patch_size=vision["patch_size"], hidden_act="gelu", layer_norm_eps=1e-5)
    with contextlib.redirect_stdout(io.StringIO()):
        m = CLIPVisionModel(cfg).eval()
    hf = {"vision_model.embeddings.class_embedding": sd["model.visual.class_embedding"],
          "vision_model.embeddings.patch_embedding.weight": sd["model.visual.conv1.weight"],
          "vision_model.embeddings.position_embedding.weight": sd["model.visual.positional_embedding"],
          "vision_model.pre_layrnorm.weight": sd["model.visual.ln_pre.weight"],
          "vision_model.pre_layrnorm.bias": sd["model.visual.ln_pre.bias"]}
    for i in range(vision["layers"]):
        hf.update(_hf_block_map(sd, f"model.visual.transformer.resblocks.{i}", f"vision_model.encoder.layers.{i}", W))
    if not any(k.startswith("vision_model.") for k in m.state_dict()):
        hf = {k[len("vision_model."):]: v for k, v in hf.items()}
    missing, unexpected = m.load_state_dict(hf, strict=False)
    assert not unexpected and all(("post_layernorm" in k or "position_ids" in k) for k in missing), (missing, unexpected)
    return m


def g16_encoders(full=False):
    """N3.  Resampler: the reference's own module (ip_resampler.py) on synthetic weights -- toy and the i2v config of
    ddpm3d.py:683-685.  CLIP towers: open_clip is absent, so the anchor is transformers' CLIP implementation carrying
    the same weights: text = final_layer_norm(hidden_states[-2]) (the 'penultimate' + ln_final of condition.py:216-234),
    vision = the last block's tokens before post_layernorm (condition.py:336-365)."""
    from lvdm.modules.encoders.ip_resampler import Resampler
    from dynamicscaler_amd.encoder_spec import (CLIP_VIT_H_14, RESAMPLER_I2V, clip_text_param_shapes,
                                                clip_vision_param_shapes, resampler_param_shapes)
    from dynamicscaler_amd.synth import synth_encoder_state_dict
    with _without_reference_stubs():
        import transformers
        from transformers import CLIPTextModel, CLIPVisionModel  # noqa: F401  (resolve the lazy modules now)
    arrays = {"transformers_version": np.frombuffer(transformers.__version__.encode(), dtype=np.uint8)}

    def resampler(tag, cfg, seed, x):
        m = Resampler(**cfg).eval()
        shapes = resampler_param_shapes(**cfg)
        assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {k: tuple(v) for k, v in shapes.items()}
        m.load_state_dict(synth_encoder_state_dict(shapes, seed))
        with torch.no_grad():
            arrays[f"{tag}_resampler_out"] = m(x)
        arrays[f"{tag}_resampler_x"] = x

    def towers(tag, spec, seed, batch):
        text, vision = spec["text"], spec["vision"]
        tsd = synth_encoder_state_dict(clip_text_param_shapes(text), seed)
        g = torch.Generator().manual_seed(seed + 1)
        tokens = torch.randint(3, text["vocab_size"], (batch, text["context_length"]), generator=g)
        tokens[:, 0] = 0
        with torch.no_grad():
            tm = _hf_text(text, tsd)
            hs = tm(input_ids=tokens, output_hidden_states=True)
            arrays[f"{tag}_text_out"] = getattr(tm, "text_model", tm).final_layer_norm(hs.hidden_states[-2])
            del tm
        arrays[f"{tag}_text_tokens"] = tokens.to(torch.int32)
        vsd = synth_encoder_state_dict(clip_vision_param_shapes(vision), seed + 2)
        pix = synth_normal((batch, 3, vision["image_size"], vision["image_size"]), seed + 3)
        with torch.no_grad():
            vs = _hf_vision(vision, vsd)(pixel_values=pix, output_hidden_states=True)
        arrays[f"{tag}_vision_out"] = vs.hidden_states[-1]
        arrays[f"{tag}_vision_pixels"] = pix

    resampler("toy", RESAMPLER_TOY, 51, synth_normal((2, 17, 192), 52))
    towers("toy", CLIP_TOY, 53, 2)
    arrays["toy_cfg_json"] = np.frombuffer(json.dumps(dict(resampler=RESAMPLER_TOY, clip=CLIP_TOY)).encode(), dtype=np.uint8)
    save_npz("encoders_toy.npz", **{k: v for k, v in arrays.items() if k.startswith(("toy", "transformers"))})
    if full:
        resampler("full", RESAMPLER_I2V, 61, synth_normal((1, 257, 1280), 62))
        towers("full", CLIP_VIT_H_14, 63, 1)
        keep = {k: (v.to(torch.float16) if torch.is_tensor(v) and v.dtype == torch.float32 and v.numel() > 65536 else v)
                for k, v in arrays.items() if k.startswith(("full", "transformers"))}
        save_npz("encoders_full.npz", **keep)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true", help="also run the full-size UNet fixture (minutes)")
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    steps = {"g1": g1_segments, "g2": g2_ring, "g3": g3_mix, "g4": g4_scheduler, "g8": g8_unet_tiny,
             "g9": g9_loops_small, "g9t": g9_traces, "g11": g11_grid_and_i2v, "g12": g12_sphere, "g13": g13_i2v_sphere, "g14": g14_vae_decode, "g15": g15_vae_encode, "g16": g16_encoders, "g19": g19_multi_prompt, "g20": g20_cfg4_geometry, "g21": g21_sphere_view_scale, "g22": g22_i2v_sphere_view_scale, "g24": g24_panorama_handlers, "g30": g30_panorama_handlers_uncalled, "g32": g32_grid_random_shuffle, "g33": g33_sphere_view_set_scale}
    if args.full:
        steps["g10"] = g10_unet_full
        steps["g10i"] = g10_unet_full_i2v
        steps["g17"] = g17_cfg1_full
        steps["g18"] = lambda: g18_unet_t24(full=True)
        steps["g23"] = g23_cfg1_50step
        steps["g25"] = g25_ring_real_unet
        steps["g26"] = g26_updates_i2v_and_t24
        steps["g27"] = g27_i2v_ring_real_unet
        steps["g28"] = g28_ring_real_unet_50step
        steps["g29"] = g29_i2v_ring_real_unet_50step
        steps["g31"] = g31_ring_real_unet_50step_mid
        steps["g34"] = g34_i2v_ring_real_unet_50step_mid
        steps["g35"] = g35_cfg3_real_unet_two_steps
        steps["g36"] = g36_cfg4_real_unet_one_step
        steps["g41"] = g41_cfg5_chain_real_unet
        steps["g37"] = g37_sphere_real_unet
        steps["g38"] = g38_grid_real_unet
        steps["g39"] = g39_i2v_sphere_real_unet
        steps["g40"] = g40_i2v_grid_real_unet
        steps["g14"] = lambda: g14_vae_decode(full=True)
        steps["g15"] = lambda: g15_vae_encode(full=True)
        steps["g16"] = lambda: g16_encoders(full=True)
    for k, fn in steps.items():
        if args.only and k != args.only:
            continue
        fn()
