"""Image-to-video overlapped-ring pipeline (drop-in call surface of
pipeline/i2v_sphere_panorama_pipeline.py:VC2_Pipeline_I2V_SpherePano.basic_sample_shift_multi_windows, :564-996)
and the pixel-space ring crop `RingImageTensor` (utils/shift_window_utils.py:209-276).

Differences from the t2v ring that are kept exactly: window placement is round(idx * float_step) (:818-820),
frame windows wrap modulo total_f and can be docked to both ends (:786-854), the denoised mask is used per frame
(5-D, :877), every window gets its own image tokens from the crop of the panorama image under it (:889-893),
and after the DDIM update x_prev is mixed with the window's pre-re-noise content under the mask (merge-prev,
:938-943).  The tile engine is the one of pipelines.py (batched levels of independent windows).

`pretrained_t2v.get_image_embeds` (CLIP image tower + Resampler, encoders.py) is called once per distinct crop
position -- the reference calls it per tile per step.  `use_skip_time` with a
given `init_panorama_latent` (the way gen_pano_360.py calls it) only shortens the schedule; without one the panorama
image is encoded by the tiled first-stage encode below and re-noised (:704-722).
"""
import math

import numpy as np
import torch

from .pipelines import VC2_Pipeline_T2V, select_prompt_from_multi_prompt_dict_by_factor
from .ring import i2v_ring_windows, i2v_grid_windows

def load_image_tensor_from_path(image_path, height, width, norm_to_1=True):
    """utils/tensor_utils.py:7-16 with PIL's bilinear resize in place of cv2.INTER_LINEAR (cv2 is not in this
    image; the two filters are not bit-identical -- image preprocessing parity is unpinned)."""
    from PIL import Image
    img = Image.open(image_path).convert("RGB").resize((width, height), Image.BILINEAR)
    t = torch.from_numpy(np.array(img, np.float32)).permute(2, 0, 1)
    if norm_to_1:
        t = (t / 255.0 - 0.5) * 2
    return t


class RingImageTensor:
    def __init__(self, image_path=None, image_tensor=None, height=320, width=512, device=None):
        """`device`: keep the panorama image there (the pipelines pass their execution device: a window crop is then a device
        gather of 2 MB instead of a 16 MB host index + upload per window)."""
        self.image_tensor = load_image_tensor_from_path(image_path, height, width) if image_tensor is None else image_tensor
        assert list(self.image_tensor.shape) == [3, height, width], \
            f"[RingImageTensor] image shape {self.image_tensor.shape} does not match {[3, height, width]}"
        if device is not None:
            self.image_tensor = self.image_tensor.to(device)

    def get_shape(self):
        return self.image_tensor.shape

    def get_window_tensor(self, pos_left, pos_right, pos_top=None, pos_down=None):
        height, width = self.get_shape()[-2], self.get_shape()[-1]
        pos_top = 0 if pos_top is None else pos_top
        pos_down = height if pos_down is None else pos_down
        assert 0 <= pos_left < pos_right <= width * 2, f"[RingImageTensor.get_window_tensor] pos_left {pos_left}, pos_right {pos_right} not legal"
        assert 0 <= pos_top < pos_down <= height * 2, f"[RingImageTensor.get_window_tensor] pos_top {pos_top}, pos_down {pos_down} not legal"
        dev = self.image_tensor.device
        yi = torch.arange(pos_top, pos_down, device=dev) % height
        xi = torch.arange(pos_left, pos_right, device=dev) % width
        return self.image_tensor[:, yi][:, :, xi]

    def get_encoded_image_cond(self, pretrained_t2v, pos_left, pos_right, pos_top=None, pos_down=None):
        crop = self.get_window_tensor(pos_left, pos_right, pos_top, pos_down)
        return pretrained_t2v.get_image_embeds(crop.to(pretrained_t2v.device).unsqueeze(0))

    def get_encoded_image_conds(self, pretrained_t2v, boxes):
        """The crops of several windows -- boxes = [(pos_left, pos_right, pos_top, pos_down), ...], all of one size -- through ONE
        get_image_embeds call ([n,3,h,w] -> [n,16,1024]; the reference calls it once per window, ddpm3d.py:689-693 takes a batch):
        a step's new windows cost one pass of the image tower at 257 n rows instead of n passes at 257.  Returns a list of
        [1,16,1024] tensors, item for item what get_encoded_image_cond gives (tests/test_gpu_encoders.py)."""
        crops = torch.stack([self.get_window_tensor(*b) for b in boxes], 0)
        out = pretrained_t2v.get_image_embeds(crops.to(pretrained_t2v.device))
        return [out[k:k + 1] for k in range(len(boxes))]


class VC2_Pipeline_I2V(VC2_Pipeline_T2V):
    """Base of the i2v pipelines (pipeline/i2v_normal_pipeline.py:27): same members as the t2v base."""

    def _load_imgs_from_paths(self, img_path_list, height=320, width=512):
        return torch.stack([load_image_tensor_from_path(p, height, width) for p in img_path_list], dim=0)

    @torch.no_grad()
    def basic_sample_shift_multi_windows(self, prompt=None, img_cond_path=None, height=320, width=512, frames=16, fps=16,
                                         guidance_scale=7.5, num_videos_per_prompt=1, generator=None,
                                         init_panorama_latent=None, num_windows_w=None, num_windows_h=None,
                                         num_windows_f=None, loop_step=None, pano_image_path=None, dock_at_h=None,
                                         latents=None, num_inference_steps=4, prompt_embeds=None, output_type="pil",
                                         merge_renoised_overlap_latent_ratio=1, use_skip_time=False,
                                         skip_time_step_idx=None, progressive_skip=False, pano_image_tensor=None,
                                         step_callback=None, **kwargs):
        """Non-overlapping shifted grid of the i2v base class (pipeline/i2v_normal_pipeline.py:68-425): tiles of the
        panorama shifted by (i % loop_step) * tile/loop_step (wrap-around), docking windows at the top / bottom edge,
        0/1 mask re-noise like the t2v ring, image tokens from the RingImageTensor crop under every window.
        total_steps is the FULL schedule length here (:147) even when use_skip_time cuts the timesteps (:139-141) -- kept.
        `pano_image_tensor` ([3, H*nh, W*nw]) is an extension (tensor instead of a path)."""
        if use_skip_time and init_panorama_latent is None:
            raise NotImplementedError("use_skip_time without init_panorama_latent encodes the panorama image with "
                                      "encode_images_list_to_latent_tensor (utils/precast_latent_utils.py); pass the latent")
        unet_config = self.model_config["params"]["unet_config"]
        frames = self.pretrained_t2v.temporal_length if frames < 0 else frames
        vs = self.vae_scale_factor
        prompt, text_emb, uc_emb = self._encode(prompt, prompt_embeds, guidance_scale)
        if guidance_scale != 1.0 and hasattr(self.pretrained_t2v, "embedder"):      # :143-149
            uc_img = torch.zeros(1, 3, height // vs, width // vs).to(self.pretrained_t2v.device)
            uc_emb = torch.cat([uc_emb.to(uc_img.device), self.pretrained_t2v.get_image_embeds(uc_img)], dim=1)
        self.scheduler.make_schedule(num_inference_steps, verbose=self.verbose)
        timesteps = np.flip(self.scheduler.ddim_timesteps)
        if use_skip_time and not progressive_skip:
            timesteps = timesteps[skip_time_step_idx:]
        total_steps = self.scheduler.ddim_timesteps.shape[0]
        lat_h, lat_w = height // vs, width // vs
        total_h, total_w = height * num_windows_h, width * num_windows_w
        total_shape = (1, unet_config["params"]["in_channels"], frames * num_windows_f, total_h // vs, total_w // vs)
        if init_panorama_latent is None:
            init_panorama_latent = torch.randn(total_shape).to(self._execution_device)  # host draw, reference order; lives on the execution device like the reference's
        else:
            assert tuple(init_panorama_latent.shape) == total_shape, \
                f"[basic_sample_shift_multi_windows] init_panorama_latent shape {tuple(init_panorama_latent.shape)} " \
                f"does not match desired shape {total_shape}"
        assert num_windows_f == 1 or frames // loop_step > 0, \
            f"[basic_sample_shift_multi_windows] loop_step {loop_step} > frames {frames} while num_windows_f {num_windows_f} > 0"
        ring_image = RingImageTensor(image_path=pano_image_path, image_tensor=pano_image_tensor, height=total_h, width=total_w,
                                     device=self._execution_device)
        st = self._new_state(init_panorama_latent, total_shape, timesteps, frames, fps, lat_h, lat_w, guidance_scale,
                             text_emb, uc_emb, merge_renoised_overlap_latent_ratio, kwargs)
        st.total_steps = total_steps
        img_cache = {}
        with self.progress_bar(total=len(timesteps)) as bar:
            for i in range(len(timesteps)):
                st.mask.zero_()                                     # reset mask record (:227)
                wins, crops = i2v_grid_windows(i, height=height, width=width, frames=frames, num_windows_h=num_windows_h,
                                               num_windows_w=num_windows_w, num_windows_f=num_windows_f, loop_step=loop_step,
                                               dock_at_h=dock_at_h)
                ctxs = []
                new = [k for k in dict.fromkeys(crops) if k not in img_cache]      # this step's crops not embedded yet, in order
                if new:
                    embs = ring_image.get_encoded_image_conds(self.pretrained_t2v, [(il, il + width, it, it + height) for (il, it) in new])
                    img_cache.update({k: e.to(st.device) for k, e in zip(new, embs)})
                for (il, it) in crops:
                    ctxs.append(torch.cat([st.text_emb, img_cache[(il, it)].to(st.text_emb.dtype)], dim=1))
                renoise = st.ratio is not None and i < total_steps - 1
                self._denoise_windows(st, i, wins, ctxs, renoise=renoise, mask_frame0=True)
                if step_callback is not None:
                    step_callback(i, int(timesteps[i]), wins, st.pano, st.pano_x0)
                bar.update()
        return self._finish(st, output_type, total_shape[2], seam_safe=False)


class VC2_Pipeline_I2V_SpherePano(VC2_Pipeline_I2V):
    def tiled_vae_encode_image(self, image_path=None, image_size=None, image_tensor=None):
        """i2v_sphere_panorama_pipeline.py:498-503; `image_tensor` [3,H,W] in [-1,1] instead of a path is an extension."""
        if image_tensor is None:
            image_tensor = load_image_tensor_from_path(image_path, image_size[0], image_size[1])
        image_tensor = image_tensor.unsqueeze(1).unsqueeze(0).to(device=self.pretrained_t2v.device, dtype=torch.float32)
        return self.tiled_vae_encode_tensor_simple(image_tensor=image_tensor)

    @torch.no_grad()
    def tiled_vae_encode_tensor_simple(self, image_tensor, h_tile_num=4, w_tile_num=4, overlap_h=32, overlap_w=32):
        """i2v_sphere_panorama_pipeline.py:505-562: image [B,3,F,H,W] -> latent [B,4,F,H/8,W/8] fp32.  Every tile is encoded
        with its overlap margin (first-stage encoder on the HIP kernels, posterior noise drawn on the host in tile order)
        and cropped back to its own cell; the cells do not overlap, so the reference's count normaliser is 1 everywhere."""
        B, _, Fr, H_dec, W_dec = image_tensor.shape
        sf = self.vae_scale_factor
        Hl, Wl = H_dec // sf, W_dec // sf
        th, tw = Hl // h_tile_num, Wl // w_tile_num
        thi, twi = th * sf, tw * sf
        ovh, ovw = overlap_h * sf, overlap_w * sf
        out = torch.zeros((B, 4, Fr, Hl, Wl), dtype=torch.float32, device=image_tensor.device)
        for i in range(h_tile_num):
            for j in range(w_tile_num):
                hs, he, ws, we = i * thi, (i + 1) * thi, j * twi, (j + 1) * twi
                hso, heo, wso, weo = max(hs - ovh, 0), min(he + ovh, H_dec), max(ws - ovw, 0), min(we + ovw, W_dec)
                lt = self.pretrained_t2v.encode_first_stage_2DAE(image_tensor[:, :, :, hso:heo, wso:weo])
                top, left = (hs - hso) // sf, (ws - wso) // sf
                bottom, right = lt.shape[3] - (heo - he) // sf, lt.shape[4] - (weo - we) // sf
                out[:, :, :, i * th:(i + 1) * th, j * tw:(j + 1) * tw] = lt[:, :, :, top:bottom, left:right]
        return out

    @torch.no_grad()
    def basic_sample_shift_multi_windows(self, prompt=None, img_cond_path=None, height=320, width=512, frames=16, fps=16,
                                         guidance_scale=7.5, num_videos_per_prompt=1, generator=None,
                                         init_panorama_latent=None, total_w=None, total_h=None, total_f=None,
                                         num_windows_w=None, num_windows_h=None, num_windows_f=None, loop_step=None,
                                         begin_index_offset=0, dock_at_f=None, overlap_ratio_list_f=None,
                                         loop_step_frame=None, pano_image_path=None, latents=None,
                                         num_inference_steps=4, prompt_embeds=None, output_type="pil",
                                         merge_renoised_overlap_latent_ratio=1, merge_prev_denoised_ratio_list=None,
                                         window_multi_prompt_dict=None, use_skip_time=False, skip_time_step_idx=None,
                                         progressive_skip=False, pano_image_tensor=None, step_callback=None, **kwargs):
        """`pano_image_tensor` ([3,total_h,total_w], optional) is an extension: the panorama image as a tensor
        instead of a path (RingImageTensor accepts both, shift_window_utils.py:211-220)."""
        st = self.ring_begin(prompt=prompt, height=height, width=width, frames=frames, fps=fps, guidance_scale=guidance_scale,
                             init_panorama_latent=init_panorama_latent, total_w=total_w, total_h=total_h, total_f=total_f,
                             num_windows_w=num_windows_w, num_windows_h=num_windows_h, num_windows_f=num_windows_f,
                             loop_step=loop_step, begin_index_offset=begin_index_offset, dock_at_f=dock_at_f,
                             overlap_ratio_list_f=overlap_ratio_list_f, loop_step_frame=loop_step_frame,
                             pano_image_path=pano_image_path, num_inference_steps=num_inference_steps,
                             prompt_embeds=prompt_embeds, merge_renoised_overlap_latent_ratio=merge_renoised_overlap_latent_ratio,
                             merge_prev_denoised_ratio_list=merge_prev_denoised_ratio_list,
                             window_multi_prompt_dict=window_multi_prompt_dict, use_skip_time=use_skip_time,
                             skip_time_step_idx=skip_time_step_idx, progressive_skip=progressive_skip,
                             pano_image_tensor=pano_image_tensor, **kwargs)
        if step_callback is not None:
            st.x0_last_only = False          # the callback sees the pred-x0 panorama of every step: exchange it on every step
        with self.progress_bar(total=st.total_steps) as bar:
            for i in range(st.total_steps):
                wins = self.ring_step(st, i)
                if step_callback is not None:
                    step_callback(i, int(st.timesteps[i]), wins, st.pano, st.pano_x0)
                bar.update()
        return self.ring_finish(st, output_type)

    # ---- the loop in three pieces (like the t2v ring, pipelines.py) so a caller (bench.py) can time single DDIM steps ----
    @torch.no_grad()
    def ring_begin(self, prompt=None, height=320, width=512, frames=16, fps=16, guidance_scale=7.5, init_panorama_latent=None,
                   total_w=None, total_h=None, total_f=None, num_windows_w=None, num_windows_h=None, num_windows_f=None,
                   loop_step=None, begin_index_offset=0, dock_at_f=None, overlap_ratio_list_f=None, loop_step_frame=None,
                   pano_image_path=None, num_inference_steps=4, prompt_embeds=None, merge_renoised_overlap_latent_ratio=1,
                   merge_prev_denoised_ratio_list=None, window_multi_prompt_dict=None, use_skip_time=False,
                   skip_time_step_idx=None, progressive_skip=False, pano_image_tensor=None, **kwargs):
        if use_skip_time and init_panorama_latent is None and getattr(self.pretrained_t2v, "first_stage_model", None) is None:
            raise NotImplementedError("use_skip_time without init_panorama_latent needs the first-stage encoder "
                                      "(first_stage_config); gen_pano_360.py passes the previous stage's latent")
        unet_config = self.model_config["params"]["unet_config"]
        frames = self.pretrained_t2v.temporal_length if frames < 0 else frames
        vs = self.vae_scale_factor
        prompt, text_emb, uc_emb = self._encode(prompt, prompt_embeds, guidance_scale)
        # uncond image tokens (:652-658): a zero image of the LATENT size through get_image_embeds
        if guidance_scale != 1.0 and hasattr(self.pretrained_t2v, "embedder"):
            uc_img = torch.zeros(1, 3, height // vs, width // vs).to(self.pretrained_t2v.device)
            uc_emb = torch.cat([uc_emb.to(uc_img.device), self.pretrained_t2v.get_image_embeds(uc_img)], dim=1)
        self.scheduler.make_schedule(num_inference_steps, verbose=self.verbose)
        timesteps = np.flip(self.scheduler.ddim_timesteps)
        if use_skip_time and not progressive_skip:     # resume from a partly denoised latent (:673-675)
            timesteps = timesteps[skip_time_step_idx:]
        if total_f is None:
            total_f = frames * num_windows_f
        lat_h, lat_w = height // vs, width // vs
        total_shape = (1, unet_config["params"]["in_channels"], total_f, total_h // vs, total_w // vs)
        if init_panorama_latent is None:
            init_panorama_latent = torch.randn(total_shape).to(self._execution_device)  # host draw, reference order; lives on the execution device like the reference's
            if use_skip_time:                                # :704-722: start from the (re-noised) VAE-encoded panorama image
                frame_0 = self.tiled_vae_encode_image(image_path=pano_image_path, image_size=(total_h, total_w),
                                                      image_tensor=pano_image_tensor)
                if progressive_skip:
                    init_panorama_latent = init_panorama_latent.to(frame_0.device)
                    for frame_idx, ps in enumerate(list(reversed(range(skip_time_step_idx)))):
                        init_panorama_latent[:, :, [frame_idx]] = self.scheduler.re_noise(frame_0, 0, num_inference_steps - ps - 1)
                else:
                    init_panorama_latent = self.scheduler.re_noise(frame_0.expand(total_shape).contiguous(), 0,
                                                                   len(timesteps) - 1)
        else:
            assert tuple(init_panorama_latent.shape) == total_shape, \
                f"[basic_sample_shift_multi_windows] init_panorama_latent shape {tuple(init_panorama_latent.shape)} " \
                f"does not match desired shape {total_shape}"
        ov_w = 1 - (total_w / width - 1) / (num_windows_w - 1)
        step_w = width / vs * (1 - ov_w)                     # float, rounded per window (:818)
        off_w = int((1 - ov_w) * width / loop_step) // vs
        assert 0 <= ov_w < 1, "overlap ratio for W is not legal"
        assert off_w >= 1, "latent_offset_step_size_w should > 1"
        ov_h = 1 - (total_h / height - 1) / (num_windows_h - 1)
        step_h = height / vs * (1 - ov_h)
        off_h = int((1 - ov_h) * height / loop_step) // vs
        assert 0 <= ov_h < 1, "overlap ratio for H is not legal"
        assert off_h >= 1, "latent_offset_step_size_h should > 1"
        step_f = 0 if total_f == frames else frames // loop_step
        assert step_f > 0 or total_f == frames, \
            f"[basic_sample_shift_multi_windows] loop_step {loop_step} > frames {frames} while total_f {total_f} > frame"
        st = self._new_state(init_panorama_latent, total_shape, timesteps, frames, fps, lat_h, lat_w, guidance_scale,
                             text_emb, uc_emb, merge_renoised_overlap_latent_ratio, kwargs)
        st.ring_image = RingImageTensor(image_path=pano_image_path, image_tensor=pano_image_tensor, height=total_h, width=total_w,
                                        device=st.device)
        st.img_cache, st.prompt_cache = {}, {}
        st.win_args = dict(latent_h=lat_h, latent_w=lat_w, frames=frames, total_f=total_f, step_w=step_w, step_h=step_h,
                           off_w=off_w, off_h=off_h, num_windows_w=num_windows_w, num_windows_h=num_windows_h,
                           loop_step=loop_step, loop_step_frame=loop_step_frame, dock_at_f=dock_at_f,
                           begin_index_offset=begin_index_offset)
        st.overlap_ratio_list_f = overlap_ratio_list_f
        st.merge_prev_denoised_ratio_list = merge_prev_denoised_ratio_list
        st.window_multi_prompt_dict = window_multi_prompt_dict
        st.height, st.width, st.total_h, st.total_f = height, width, total_h, total_f
        return st

    @torch.no_grad()
    def ring_step(self, st, i):
        """One DDIM step over all windows (i2v_sphere_panorama_pipeline.py:806-960)."""
        vs = self.vae_scale_factor
        st.mask.zero_()  # reset denoised mask record (:810)
        wins = i2v_ring_windows(i, overlap_ratio_f=st.overlap_ratio_list_f[i], **st.win_args)
        if st.world > 1 and not hasattr(st, "x0_last_only"):
            from . import parallel
            last_i = st.total_steps - 1
            st.x0_last_only = self.exchange_x0 == "last" and parallel.windows_cover(
                i2v_ring_windows(last_i, overlap_ratio_f=st.overlap_ratio_list_f[last_i], **st.win_args), st.pano_fhw)
        ctxs = []
        new = [k for k in dict.fromkeys((w[0], w[2]) for w in wins) if k not in st.img_cache]   # crops not embedded yet: one tower pass
        if new:
            embs = st.ring_image.get_encoded_image_conds(
                self.pretrained_t2v, [(l * vs, l * vs + st.width, t * vs, t * vs + st.height) for (l, t) in new])
            st.img_cache.update({k: e.to(st.device) for k, e in zip(new, embs)})
        for (left, _r, top, _d, _fb, _fe) in wins:
            cur_text = st.text_emb
            if st.window_multi_prompt_dict is not None:
                cur = select_prompt_from_multi_prompt_dict_by_factor(st.window_multi_prompt_dict,
                                                                      (top * vs + st.height) / st.total_h)
                if cur not in st.prompt_cache:
                    st.prompt_cache[cur] = self.pretrained_t2v.get_learned_conditioning([cur]).to(st.device)
                cur_text = st.prompt_cache[cur]
            key = (left, top)             # embedded above; the same crop position recurs every loop_step steps
            ctxs.append(torch.cat([cur_text, st.img_cache[key].to(cur_text.dtype)], dim=1))
        renoise = st.ratio is not None and i < st.total_steps - 1
        merge_prev = None
        if st.merge_prev_denoised_ratio_list is not None and i < st.total_steps - 1:
            merge_prev = st.merge_prev_denoised_ratio_list[i]
        self._denoise_windows(st, i, wins, ctxs, renoise=renoise, mask_frame0=False, merge_prev_ratio=merge_prev)
        return wins

    @torch.no_grad()
    def ring_finish(self, st, output_type="latent"):
        return self._finish(st, output_type, st.total_f, seam_safe=True)
