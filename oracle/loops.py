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
                    trace=None, on_tile=None, window_multi_prompt_dict=None, get_learned_conditioning=None):
    """VC2_Pipeline_T2V_SpherePano.basic_sample_shift_multi_windows, output_type='latent'
    (t2v_sphere_panorama_pipeline.py:316-660).  Returns (denoised, denoised, final_latent).
    window_multi_prompt_dict / get_learned_conditioning: the per-window prompt (:561-566, utils/multi_prompt_utils.py:1-7):
    the prompt of the first key >= window_down / total_latent_h, re-encoded for every window."""
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
            if window_multi_prompt_dict is not None:
                factor = dn / (total_h // VAE_SCALE)
                assert 0.0 <= factor <= 1.0, f"select_prompt: input factor {factor} not legal"   # multi_prompt_utils.py:2
                keys = sorted(window_multi_prompt_dict.keys())
                chosen = next((window_multi_prompt_dict[k] for k in keys if factor <= k), window_multi_prompt_dict[keys[-1]])
                cond_ctx = get_learned_conditioning([chosen])   # stays bound for the following windows, like `text_emb` (:565)
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


# ================================================================================================
# P4: non-overlapping shifted grid -- VC2_Pipeline_T2V.basic_sample_shift_multi_windows
#     (pipeline/t2v_normal_pipeline.py:213-568), the path without pre-denoise
# ================================================================================================
def t2v_grid_windows(i, *, latent_h, latent_w, frames, num_windows_w, num_windows_h, num_windows_f, loop_step,
                     shift_jump_odd_w=False, shift_jump_odd_h=False, shift_jump_odd_f=False, docking_w=False,
                     docking_h=False, docking_f=False, docking_step_range=None):
    """Window list of step i in reference order f -> w -> h (t2v_normal_pipeline.py:419-432, 441-443, 471-522).
    NB the reference's jump-odd flags are crossed: shift_jump_odd_h moves the LEFT start, shift_jump_odd_w the TOP
    (:471-474) -- kept as is."""
    step_w = 0 if num_windows_w == 1 else (latent_w * VAE_SCALE // loop_step) // VAE_SCALE
    step_h = 0 if num_windows_h == 1 else (latent_h * VAE_SCALE // loop_step) // VAE_SCALE
    step_f = 0 if num_windows_f == 1 else frames // loop_step
    left0 = (i % loop_step) * step_w
    top0 = (i % loop_step) * step_h
    fr0 = (i % loop_step) * step_f
    if i % 2 == 1 and shift_jump_odd_h and num_windows_h > 1:
        left0 = left0 + (latent_w * num_windows_w // 2)
    if i % 2 == 1 and shift_jump_odd_w and num_windows_w > 1:
        top0 = top0 + (latent_h * num_windows_h // 2)
    if i % 2 == 1 and shift_jump_odd_f and num_windows_f > 1:
        fr0 = fr0 + (frames * num_windows_f // 2)
    in_dock = docking_step_range is not None and i in docking_step_range
    wins = []
    for fi in (range(-1, num_windows_f) if docking_f else range(num_windows_f)):
        for wi in (range(-1, num_windows_w) if docking_w else range(num_windows_w)):
            for hi in (range(-1, num_windows_h) if docking_h else range(num_windows_h)):
                left = left0 + wi * latent_w
                right = left + latent_w
                top = top0 + hi * latent_h
                down = top + latent_h
                fb = fr0 + fi * frames
                fe = fb + frames
                if docking_w and in_dock:
                    if wi == -1:
                        left, right = 0, latent_w
                    if wi == num_windows_w - 1:
                        left, right = latent_w * (num_windows_w - 1), latent_w * num_windows_w
                elif wi == -1:
                    continue
                if docking_h and in_dock:
                    if hi == -1:
                        top, down = 0, latent_h
                    if hi == num_windows_h - 1:
                        top, down = latent_h * (num_windows_h - 1), latent_h * num_windows_h
                elif hi == -1:
                    continue
                if docking_f and in_dock:
                    if fi == -1:
                        fb, fe = 0, frames
                    if fi == num_windows_f - 1:
                        fb, fe = frames * (num_windows_f - 1), frames * num_windows_f
                elif fi == -1:
                    continue
                wins.append((left, right, top, down, fb, fe))
    return wins


@torch.no_grad()
def resize_video_latent(x, target_height, target_width, mode="bicubic"):
    """utils/diffusion_utils.py:21-33: F.interpolate per frame."""
    import torch.nn.functional as F
    b, c, f, h, w = x.shape
    y = F.interpolate(x.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w), size=(target_height, target_width), mode=mode,
                      align_corners=None if mode == "nearest" else False)
    return y.view(b, f, c, target_height, target_width).permute(0, 2, 1, 3, 4)


def _add_noise(sched, clear, index):
    """VC2_Pipeline_T2V._add_noise (t2v_normal_pipeline.py:619-625): sqrt(a) x + sqrt(1-a) randn_like(x), a = ddim_alphas[index]."""
    alpha = torch.as_tensor(sched.ddim_alphas[index], dtype=torch.float32)
    beta = 1 - alpha
    return (alpha ** 0.5) * clear.clone() + (beta ** 0.5) * torch.randn_like(clear)


def t2v_grid_sample(eps_model, tables: DiffusionTables, cond_ctx, uncond_ctx, *, height=320, width=512, frames=16,
                    guidance_scale=7.5, num_windows_w, num_windows_h, num_windows_f=1, loop_step=8,
                    num_inference_steps=50, init_panorama_latent=None, in_channels=4, trace=None,
                    use_pre_denoise=False, pre_denoise_steps=None, skip_steps_after_pre_denoise=0,
                    clear_pre_denoised_latent=None, merge_predenoise_ratio_list=None, sparse_add_residual=True,
                    use_skip_time=False, skip_time_step_idx=None, progressive_skip=False,
                    clear_pre_denoised_video_tensor=None, encode_first_stage=None, random_shuffle_init_frame_stride=0, **grid_kw):
    """Returns (denoised, denoised) for output_type='latent' (t2v_normal_pipeline.py:561-568).  Includes the pre-denoise
    start (:345-412: a single tile denoised for a few steps or given, resized bicubically to the panorama, re-noised,
    optionally with a per-frame progressive noise level) and the per-step sparse / dense residual merge (:445-468)."""
    sched = DDIMSchedule(tables, num_inference_steps)
    full_timesteps = np.flip(sched.ddim_timesteps)
    if use_skip_time and not progressive_skip:
        timesteps = full_timesteps[skip_time_step_idx - skip_steps_after_pre_denoise:]          # :299-301
    else:
        timesteps = full_timesteps
    total_steps = len(timesteps)
    lh, lw = height // VAE_SCALE, width // VAE_SCALE
    total_shape = (1, in_channels, frames * num_windows_f, lh * num_windows_h, lw * num_windows_w)
    resized = None

    def basic_step(latent, t, i):                                                           # _basic_denoise_one_step :572-615
        ts = torch.full((1,), int(t), dtype=torch.long)
        e_c = eps_model(latent, ts, cond_ctx)
        e = cfg_combine(e_c, eps_model(latent, ts, uncond_ctx), guidance_scale) if guidance_scale != 1.0 else e_c
        return ddim_step(sched, latent, e, [total_steps - i - 1] * latent.shape[2])

    if init_panorama_latent is None:
        pano = torch.randn(total_shape)
        if random_shuffle_init_frame_stride > 0:                 # t2v_normal_pipeline.py:328-337 (dim 3 = H is what the reference indexes)
            import random
            for frame_index in range(frames, frames * num_windows_f, random_shuffle_init_frame_stride):
                list_index = list(range(frame_index - frames, frame_index + random_shuffle_init_frame_stride - frames))
                random.shuffle(list_index)
                pano[:, :, :, frame_index:frame_index + random_shuffle_init_frame_stride] = pano[:, :, :, list_index]
        if use_skip_time:
            assert use_pre_denoise and pre_denoise_steps > 0 and skip_time_step_idx >= skip_steps_after_pre_denoise
        if use_pre_denoise and pre_denoise_steps > 0:
            if (num_windows_h != 1 or num_windows_w != 1) and num_windows_f != 1:
                raise NotImplementedError()
            latent = torch.randn((1, in_channels, frames, lh, lw))                            # drawn in every branch (:358)
            if clear_pre_denoised_video_tensor is not None:                                   # :363-368, a clear clip in pixel space
                clip = resize_video_latent(clear_pre_denoised_video_tensor.clone(), height * num_windows_h,
                                           width * num_windows_w, "bicubic")
                resized = encode_first_stage(clip).clone()       # pretrained_t2v.encode_first_stage_2DAE (ddpm3d.py:485-490)
                assert tuple(resized.shape) == total_shape
            else:
                if clear_pre_denoised_latent is not None:
                    latent = clear_pre_denoised_latent.clone()
                else:
                    for i, t in enumerate(full_timesteps[:pre_denoise_steps]):
                        latent, _ = basic_step(latent, t, i)
                resized = resize_video_latent(latent.clone(), lh * num_windows_h, lw * num_windows_w, "bicubic")
            pano = _add_noise(sched, resized, total_steps - 1)
            if use_skip_time:
                if progressive_skip:
                    for frame_idx, progs in enumerate(list(reversed(range(skip_time_step_idx)))):
                        pano[:, :, [frame_idx]] = _add_noise(sched, resized[:, :, [frame_idx]], total_steps - progs - 1).clone()
                else:
                    pano = _add_noise(sched, resized, total_steps - 1)
    else:
        pano = init_panorama_latent.clone()
    pano_x0 = torch.zeros_like(pano)
    for i, t in enumerate(timesteps):
        if use_pre_denoise and merge_predenoise_ratio_list is not None and resized is not None:
            assert len(merge_predenoise_ratio_list) == len(timesteps)
            r = merge_predenoise_ratio_list[i]
            curr = pano.clone()
            noised = re_noise(sched, resized.clone(), 0, total_steps - i - 1)
            if sparse_add_residual:
                mixed = curr.clone()
                mixed[..., i % 2::2, ::2] = r * curr[..., (i + 1) % 2::2, ::2] + (1.0 - r) * noised[..., ::2, ::2]
                mixed[..., (i + 1) % 2::2, 1::2] = r * curr[..., i % 2::2, 1::2] + (1.0 - r) * noised[..., ::2, ::2]
            else:
                mixed = curr * r + noised * (1.0 - r)
            pano = mixed.clone()
        wins = t2v_grid_windows(i, latent_h=lh, latent_w=lw, frames=frames, num_windows_w=num_windows_w,
                                num_windows_h=num_windows_h, num_windows_f=num_windows_f, loop_step=loop_step, **grid_kw)
        if trace is not None:
            trace.append((i, int(t), wins))
        for (l, r_, tp, dn, fb, fe) in wins:
            win = ring_gather(pano, l, r_, tp, dn, fb, fe)
            ts = torch.full((1,), int(t), dtype=torch.long)
            e_c = eps_model(win, ts, cond_ctx)
            e = cfg_combine(e_c, eps_model(win, ts, uncond_ctx), guidance_scale) if guidance_scale != 1.0 else e_c
            x_prev, x0 = ddim_step(sched, win, e, [total_steps - i - 1] * win.shape[2])
            ring_scatter(pano, x_prev, l, r_, tp, dn, fb, fe)
            ring_scatter(pano_x0, x0, l, r_, tp, dn, fb, fe)
    return pano_x0.clone(), pano_x0.clone()


# ================================================================================================
# P3: i2v overlapped ring with temporal windows / docking / merge-prev --
#     VC2_Pipeline_I2V_SpherePano.basic_sample_shift_multi_windows (pipeline/i2v_sphere_panorama_pipeline.py:564-996)
# ================================================================================================
_DOCK_START_INDEX = -101
_DOCK_END_INDEX = -111


def i2v_ring_windows(i, *, height, width, frames, total_h, total_w, total_f, num_windows_h, num_windows_w, loop_step,
                     overlap_ratio_f, loop_step_frame=None, dock_at_f=None, begin_index_offset=0):
    """Windows of step i in reference order f -> w -> h (i2v_sphere_panorama_pipeline.py:732-766, 779-854).
    Unlike the t2v ring, window placement is round(idx * float_step) (:818-820) and frame windows wrap modulo
    total_f (:828-830)."""
    import math
    lh, lw = height // VAE_SCALE, width // VAE_SCALE
    ov_w = 1 - (total_w / width - 1) / (num_windows_w - 1)
    step_w = width / VAE_SCALE * (1 - ov_w)
    off_w = int((1 - ov_w) * width / loop_step) // VAE_SCALE
    ov_h = 1 - (total_h / height - 1) / (num_windows_h - 1)
    step_h = height / VAE_SCALE * (1 - ov_h)
    off_h = int((1 - ov_h) * height / loop_step) // VAE_SCALE
    assert 0 <= ov_w < 1 and off_w >= 1 and 0 <= ov_h < 1 and off_h >= 1
    k = (i + begin_index_offset) % loop_step
    left0, top0 = k * off_w, k * off_h
    n_f = math.ceil((total_f // frames - 1) / (1 - overlap_ratio_f)) + 1
    if total_f > frames:
        off_f = max(int(overlap_ratio_f * frames / loop_step_frame), 1)
        fr0 = (i % loop_step_frame) * off_f
        f_ids = list(range(n_f))
        if dock_at_f:
            f_ids = [_DOCK_START_INDEX] + f_ids + [_DOCK_END_INDEX]
    elif total_f == frames:
        fr0, f_ids = 0, [0]
    else:
        raise ValueError(f"total_f {total_f} should >= frames {frames} !")
    wins = []
    for fi in f_ids:
        for wi in range(num_windows_w):
            for hi in range(num_windows_h):
                left = left0 + round(wi * step_w)
                top = top0 + round(hi * step_h)
                fb = (fr0 + fi * int(frames * (1 - overlap_ratio_f))) % total_f
                fe = fb + frames
                if dock_at_f:
                    if fi == _DOCK_START_INDEX:
                        if fr0 == 0:
                            continue
                        fb, fe = 0, frames
                    if fi == _DOCK_END_INDEX:
                        if fr0 == 0:
                            continue
                        fb, fe = total_f - frames, total_f
                    if fe > total_f:
                        continue
                wins.append((left, left + lw, top, top + lh, fb, fe))
    return wins


def i2v_frame_windows(i, *, frames, total_f, overlap_ratio_f, loop_step_frame=None, dock_at_f=None):
    """Frame windows [(f_begin, f_end)] of step i, reference order (i2v_sphere_panorama_pipeline.py:256-315; the same
    arithmetic as :779-854 of the ring loop).  f_begin wraps modulo total_f; f_end = f_begin + frames may exceed total_f."""
    import math
    n_f = math.ceil((total_f // frames - 1) / (1 - overlap_ratio_f)) + 1
    if total_f > frames:
        off_f = max(int(overlap_ratio_f * frames / loop_step_frame), 1)
        fr0 = (i % loop_step_frame) * off_f
        f_ids = list(range(n_f))
        if dock_at_f:
            f_ids = [_DOCK_START_INDEX] + f_ids + [_DOCK_END_INDEX]
    elif total_f == frames:
        fr0, f_ids = 0, [0]
    else:
        raise ValueError(f"total_f {total_f} should >= frames {frames} !")
    out = []
    for fi in f_ids:
        fb = (fr0 + fi * int(frames * (1 - overlap_ratio_f))) % total_f
        fe = fb + frames
        if dock_at_f:
            if fi == _DOCK_START_INDEX:
                if fr0 == 0:
                    continue
                fb, fe = 0, frames
            if fi == _DOCK_END_INDEX:
                if fr0 == 0:
                    continue
                fb, fe = total_f - frames, total_f
            if fe > total_f:
                continue
        out.append((fb, fe))
    return out


@torch.no_grad()
def i2v_ring_sample(eps_model, image_embedder, tables: DiffusionTables, text_ctx, uncond_ctx, pano_image, *, height=320,
                    width=512, frames=16, guidance_scale=7.5, total_w, total_h, total_f=None, num_windows_w,
                    num_windows_h, num_windows_f=1, loop_step=8, begin_index_offset=0, dock_at_f=None,
                    overlap_ratio_list_f=None, loop_step_frame=None, num_inference_steps=4, init_panorama_latent=None,
                    merge_renoised_overlap_latent_ratio=1, merge_prev_denoised_ratio_list=None, in_channels=4,
                    use_skip_time=False, skip_time_step_idx=None, progressive_skip=False, trace=None):
    """output_type='latent'.  `image_embedder(crop [1,3,h,w]) -> [1,L_img,D]` stands for get_image_embeds;
    `pano_image` [3,total_h,total_w] for the RingImageTensor content (utils/shift_window_utils.py:209-276).
    uncond_ctx must already contain the image-token part (i2v_sphere…py:652-658)."""
    sched = DDIMSchedule(tables, num_inference_steps)
    timesteps = np.flip(sched.ddim_timesteps)
    if use_skip_time and not progressive_skip:       # :673-675; the VAE-encoded start of :706-722 is not restated
        assert init_panorama_latent is not None
        timesteps = timesteps[skip_time_step_idx:]
    total_steps = len(timesteps)
    if total_f is None:
        total_f = frames * num_windows_f
    total_shape = (1, in_channels, total_f, total_h // VAE_SCALE, total_w // VAE_SCALE)
    pano = torch.randn(total_shape) if init_panorama_latent is None else init_panorama_latent.clone()
    pano_x0 = torch.zeros_like(pano)
    img5 = pano_image[None, :, None]  # [1,3,1,H,W] so ring_gather can crop with wrap
    for i, t in enumerate(timesteps):
        wins = i2v_ring_windows(i, height=height, width=width, frames=frames, total_h=total_h, total_w=total_w,
                                total_f=total_f, num_windows_h=num_windows_h, num_windows_w=num_windows_w,
                                loop_step=loop_step, overlap_ratio_f=overlap_ratio_list_f[i],
                                loop_step_frame=loop_step_frame, dock_at_f=dock_at_f,
                                begin_index_offset=begin_index_offset)
        if trace is not None:
            trace.append((i, int(t), wins))
        mask = torch.zeros_like(pano)
        for (l, r, tp, dn, fb, fe) in wins:
            win = ring_gather(pano, l, r, tp, dn, fb, fe)
            prev = win.clone()
            wmask = ring_gather(mask, l, r, tp, dn, fb, fe)
            if merge_renoised_overlap_latent_ratio is not None and i < total_steps - 1:
                noised = re_noise(sched, win.clone(), total_steps - i - 2, total_steps - i - 1)
                win = mix_latents_with_mask(win, noised, wmask, merge_renoised_overlap_latent_ratio)  # 5-D mask (:877)
            crop = ring_gather(img5, l * VAE_SCALE, l * VAE_SCALE + width, tp * VAE_SCALE, tp * VAE_SCALE + height, 0, 1)
            img_emb = image_embedder(crop[:, :, 0])
            ctx = torch.cat([text_ctx, img_emb], dim=1)
            ts = torch.full((1,), int(t), dtype=torch.long)
            e_c = eps_model(win, ts, ctx)
            e = cfg_combine(e_c, eps_model(win, ts, uncond_ctx), guidance_scale) if guidance_scale != 1.0 else e_c
            x_prev, x0 = ddim_step(sched, win, e, [total_steps - i - 1] * win.shape[2])
            if merge_prev_denoised_ratio_list is not None and i < total_steps - 1:
                x_prev = mix_latents_with_mask(x_prev, prev, wmask, merge_prev_denoised_ratio_list[i])
            ring_scatter(pano, x_prev, l, r, tp, dn, fb, fe)
            ring_scatter(pano_x0, x0, l, r, tp, dn, fb, fe)
            ring_scatter(mask, torch.ones_like(x_prev), l, r, tp, dn, fb, fe)
    return pano_x0.clone(), pano_x0.clone(), pano


def i2v_grid_windows(i, *, height, width, frames, num_windows_h, num_windows_w, num_windows_f, loop_step, dock_at_h=None):
    """Windows of step i of VC2_Pipeline_I2V.basic_sample_shift_multi_windows (pipeline/i2v_normal_pipeline.py:214-290):
    non-overlapping tiles shifted by (i % loop_step) * tile/loop_step, order f -> w -> h, docking windows (-100 top,
    -101 bottom) FIRST in the h list.  Entries: (left, right, top, down, f_begin, f_end, img_left, img_top): latent
    window + the pixel origin of the image crop (computed separately from the latent one, like the reference)."""
    lh, lw = height // VAE_SCALE, width // VAE_SCALE
    img_sw = width // loop_step
    lat_sw = 0 if num_windows_w == 1 else img_sw // VAE_SCALE
    img_sh = height // loop_step
    lat_sh = 0 if num_windows_h == 1 else img_sh // VAE_SCALE
    lat_sf = 0 if num_windows_f == 1 else frames // loop_step
    k = i % loop_step
    total_lh = height * num_windows_h // VAE_SCALE
    wins = []
    for fi in range(num_windows_f):
        for wi in range(num_windows_w):
            h_ids = list(range(num_windows_h))
            if dock_at_h:
                h_ids = [-100, -101] + h_ids
            for hi in h_ids:
                img_left, img_top = k * img_sw + wi * width, k * img_sh + hi * height
                left, top = k * lat_sw + wi * lw, k * lat_sh + hi * lh
                fb = k * lat_sf + fi * frames
                if dock_at_h:
                    if hi in (-100, -101) and k == 0:
                        continue
                    if hi == -100:
                        top, img_top = 0, 0
                    elif hi == -101:
                        top, img_top = total_lh - lh, height * num_windows_h - height
                    if top + lh > total_lh:
                        continue
                wins.append((left, left + lw, top, top + lh, fb, fb + frames, img_left, img_top))
    return wins


@torch.no_grad()
def i2v_grid_sample(eps_model, image_embedder, tables: DiffusionTables, text_ctx, uncond_ctx, pano_image, *, height=320,
                    width=512, frames=16, guidance_scale=7.5, num_windows_w, num_windows_h, num_windows_f=1, loop_step=8,
                    dock_at_h=None, num_inference_steps=4, init_panorama_latent=None, merge_renoised_overlap_latent_ratio=1,
                    use_skip_time=False, skip_time_step_idx=None, progressive_skip=False, in_channels=4, trace=None):
    """pipeline/i2v_normal_pipeline.py:68-425, output_type='latent': returns (denoised, denoised).  NB total_steps is the
    FULL schedule length here (:147) even when use_skip_time cuts the timesteps (:139-141)."""
    sched = DDIMSchedule(tables, num_inference_steps)
    timesteps = np.flip(sched.ddim_timesteps)
    if use_skip_time and not progressive_skip:
        assert init_panorama_latent is not None
        timesteps = timesteps[skip_time_step_idx:]
    total_steps = sched.ddim_timesteps.shape[0]
    total_shape = (1, in_channels, frames * num_windows_f, height * num_windows_h // VAE_SCALE, width * num_windows_w // VAE_SCALE)
    pano = torch.randn(total_shape) if init_panorama_latent is None else init_panorama_latent.clone()
    pano_x0 = torch.zeros_like(pano)
    img5 = pano_image[None, :, None]
    for i, t in enumerate(timesteps):
        wins = i2v_grid_windows(i, height=height, width=width, frames=frames, num_windows_h=num_windows_h,
                                num_windows_w=num_windows_w, num_windows_f=num_windows_f, loop_step=loop_step, dock_at_h=dock_at_h)
        if trace is not None:
            trace.append((i, int(t), [w[:6] for w in wins]))
        mask = torch.zeros_like(pano)
        for (l, r, tp, dn, fb, fe, il, it) in wins:
            win = ring_gather(pano, l, r, tp, dn, fb, fe)
            crop = ring_gather(img5, il, il + width, it, it + height, 0, 1)
            ctx = torch.cat([text_ctx, image_embedder(crop[:, :, 0])], dim=1)
            wmask = ring_gather(mask, l, r, tp, dn, fb, fe)
            if merge_renoised_overlap_latent_ratio is not None and i < total_steps - 1:
                noised = re_noise(sched, win.clone(), total_steps - i - 2, total_steps - i - 1)
                win = mix_latents_with_mask(win, noised, wmask[0, 0, [0]], merge_renoised_overlap_latent_ratio)
            ts = torch.full((1,), int(t), dtype=torch.long)
            e_c = eps_model(win, ts, ctx)
            e = cfg_combine(e_c, eps_model(win, ts, uncond_ctx), guidance_scale) if guidance_scale != 1.0 else e_c
            x_prev, x0 = ddim_step(sched, win, e, [total_steps - i - 1] * win.shape[2])
            ring_scatter(pano, x_prev, l, r, tp, dn, fb, fe)
            ring_scatter(pano_x0, x0, l, r, tp, dn, fb, fe)
            ring_scatter(mask, torch.ones_like(x_prev), l, r, tp, dn, fb, fe)
    return pano_x0.clone(), pano_x0.clone()
