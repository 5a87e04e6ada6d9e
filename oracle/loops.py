"""TEST INFRASTRUCTURE (oracle): the per-step tile loops of the panorama pipelines on CPU.

Restates (same tile order, same global-RNG draw order -- SURVEY.md appendix B):
  * VC2_Pipeline_T2V.basic_sample                          pipeline/t2v_normal_pipeline.py:69-210
  * VC2_Pipeline_T2V_SpherePano.basic_sample_shift_multi_windows
                                                           pipeline/t2v_sphere_panorama_pipeline.py:316-660
    - window-grid arithmetic                               :437-476, :483-489, :507-513
    - dock_at_h                                            :500-532
    - re-noise of already denoised overlap under the mask  :550-559
    - CFG, DDIM step, three scatters                       :579-632
`eps_model(x, ts, context)` stands for `pretrained_t2v.model(x, ts, c_crossattn=[context], fps=fps, ...)`.
"""
import numpy as np
import torch

from .ring import ring_gather, ring_scatter
from .ddim import DDIMSchedule, DiffusionTables, ddim_step, re_noise, mix_latents_with_mask, cfg_combine

VAE_SCALE = 8  # t2v_normal_pipeline.py:48


def ring_grid(total, tile, num_windows, loop_step):
    """One axis of t2v_sphere_panorama_pipeline.py:437-476 (pixels in, latent units out).

    Returns (overlap_ratio, latent_window_step, latent_offset_step)."""
    overlap = 1 - (total / tile - 1) / (num_windows - 1)
    image_window_step = int(tile * (1 - overlap))
    latent_window_step = image_window_step // VAE_SCALE
    image_offset_step = int((1 - overlap) * tile / loop_step)
    latent_offset_step = image_offset_step // VAE_SCALE
    if num_windows == 1:  # unreachable (division by zero above) but kept like the reference :443-445
        latent_offset_step = 0
    assert 0 <= overlap < 1, "overlap ratio is not legal"
    assert latent_offset_step > 0, "latent offset step <= 0 ! consider increase windows"
    return overlap, latent_window_step, latent_offset_step


def t2v_ring_windows(i, *, height, width, frames, total_h, total_w, num_windows_h, num_windows_w,
                     num_windows_f, loop_step, dock_at_h=None):
    """Window list of DDIM step i in reference loop order f -> w -> h
    (t2v_sphere_panorama_pipeline.py:483-532).  Each entry: (left, right, top, down, f_begin, f_end)."""
    lh, lw = height // VAE_SCALE, width // VAE_SCALE
    _, step_w, off_w = ring_grid(total_w, width, num_windows_w, loop_step)
    _, step_h, off_h = ring_grid(total_h, height, num_windows_h, loop_step)
    step_f = frames // loop_step
    if num_windows_f == 1:
        step_f = 0
    assert step_f > 0 or num_windows_f == 1
    left0 = (i % loop_step) * off_w
    top0 = (i % loop_step) * off_h
    f0 = (i % loop_step) * step_f
    wins = []
    for fi in range(num_windows_f):
        for wi in range(num_windows_w):
            h_list = list(range(num_windows_h))
            if dock_at_h:
                h_list = [-100] + h_list + [-101]
            for hi in h_list:
                left = left0 + wi * step_w
                right = left + lw
                top = top0 + hi * step_h
                down = top + lh
                fb = f0 + fi * frames
                fe = fb + frames
                if dock_at_h:
                    if hi == -100:
                        if i % loop_step == 0:
                            continue
                        top, down = 0, lh
                    if hi == -101:
                        if i % loop_step == 0:
                            continue
                        top = total_h // VAE_SCALE - lh
                        down = top + lh
                    if down > total_h // VAE_SCALE:
                        continue
                wins.append((left, right, top, down, fb, fe))
    return wins


@torch.no_grad()
def t2v_basic_sample(eps_model, tables: DiffusionTables, cond_ctx, uncond_ctx, *, height=320, width=512,
                     frames=16, guidance_scale=7.5, num_inference_steps=4, latents=None, in_channels=4):
    """VC2_Pipeline_T2V.basic_sample with output_type='latent' (t2v_normal_pipeline.py:69-210).
    Returns (denoised, denoised): the pred_x0 of the last step (:205-210)."""
    sched = DDIMSchedule(tables, num_inference_steps)
    timesteps = np.flip(sched.ddim_timesteps)
    total_steps = num_inference_steps
    if latents is None:
        latents = torch.randn((1, 1, in_channels, frames, height // VAE_SCALE, width // VAE_SCALE))[0]
    denoised = None
    for i, t in enumerate(timesteps):
        ts = torch.full((1,), int(t), dtype=torch.long)
        e_c = eps_model(latents, ts, cond_ctx)
        if guidance_scale != 1.0:
            e_u = eps_model(latents, ts, uncond_ctx)
            e = cfg_combine(e_c, e_u, guidance_scale)
        else:
            e = e_c
        index = total_steps - i - 1
        latents, denoised = ddim_step(sched, latents, e, [index] * latents.shape[2])
    return denoised, denoised


@torch.no_grad()
def t2v_ring_sample(eps_model, tables: DiffusionTables, cond_ctx, uncond_ctx, *, height=320, width=512,
                    frames=16, guidance_scale=7.5, total_w, total_h, num_windows_w, num_windows_h,
                    num_windows_f=1, loop_step=8, dock_at_h=None, num_inference_steps=4,
                    init_panorama_latent=None, merge_renoised_overlap_latent_ratio=1, in_channels=4,
                    trace=None, on_tile=None):
    """VC2_Pipeline_T2V_SpherePano.basic_sample_shift_multi_windows, output_type='latent'
    (t2v_sphere_panorama_pipeline.py:316-660).  Returns (denoised, denoised, final_latent)."""
    sched = DDIMSchedule(tables, num_inference_steps)
    timesteps = np.flip(sched.ddim_timesteps)
    total_steps = len(timesteps)
    total_shape = (1, in_channels, frames * num_windows_f, total_h // VAE_SCALE, total_w // VAE_SCALE)
    if init_panorama_latent is None:
        init_panorama_latent = torch.randn(total_shape)
    else:
        assert tuple(init_panorama_latent.shape) == total_shape
        init_panorama_latent = init_panorama_latent.clone()
    pano = init_panorama_latent.clone()
    pano_x0 = torch.zeros_like(pano)
    for i, t in enumerate(timesteps):
        mask = torch.zeros_like(pano)  # :494 fresh mask every step
        wins = t2v_ring_windows(i, height=height, width=width, frames=frames, total_h=total_h,
                                total_w=total_w, num_windows_h=num_windows_h, num_windows_w=num_windows_w,
                                num_windows_f=num_windows_f, loop_step=loop_step, dock_at_h=dock_at_h)
        if trace is not None:
            trace.append((i, int(t), wins))
        for (l, r, tp, dn, fb, fe) in wins:
            win = ring_gather(pano, l, r, tp, dn, fb, fe)
            wmask = ring_gather(mask, l, r, tp, dn, fb, fe)
            if merge_renoised_overlap_latent_ratio is not None and i < total_steps - 1:
                noised = re_noise(sched, win.clone(), total_steps - i - 2, total_steps - i - 1)
                wmask3 = wmask[0, 0, [0]]  # :555 -> [1,h,w]
                win = mix_latents_with_mask(win, noised, wmask3, merge_renoised_overlap_latent_ratio)
            ts = torch.full((1,), int(t), dtype=torch.long)
            e_c = eps_model(win, ts, cond_ctx)
            if guidance_scale != 1.0:
                e_u = eps_model(win, ts, uncond_ctx)
                e = cfg_combine(e_c, e_u, guidance_scale)
            else:
                e = e_c
            index = total_steps - i - 1
            x_prev, x0 = ddim_step(sched, win, e, [index] * win.shape[2])
            ring_scatter(pano, x_prev, l, r, tp, dn, fb, fe)
            ring_scatter(pano_x0, x0, l, r, tp, dn, fb, fe)
            ring_scatter(mask, torch.ones_like(x_prev), l, r, tp, dn, fb, fe)
            if on_tile is not None:
                on_tile(i, (l, r, tp, dn, fb, fe), pano, pano_x0)
    denoised = pano_x0.clone()
    return denoised, denoised, pano
