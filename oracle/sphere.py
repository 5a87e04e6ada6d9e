"""TEST INFRASTRUCTURE (oracle): perspective-view <-> equirectangular panorama ops and the t2v sphere loop on CPU.

Restates
  * PanoramaTensor._get_uv                          utils/panorama_tensor_utils.py:204-245  (-> view_uv)
  * _sample_equirect_tensor_nearest / get_view_tensor_no_interpolate   :53-70, :185-202      (-> sphere_gather)
  * set_view_tensor_no_interpolation                :154-183 (duplicate targets: LAST source in row-major view order
                                                    wins, like torch's CPU index_put -- SURVEY.md 8-a S3)          (-> sphere_scatter)
  * PanoramaLatentProxy                             :249-290 ([B,C,N,H,W] <-> [B,N,C,H,W] permutes)
  * VC2_Pipeline_T2V_SpherePano.basic_sample_shift_shpere_panorama   pipeline/t2v_sphere_panorama_pipeline.py:23-312
The uv map is computed with the SAME torch CPU op sequence as the reference (fp32 tan / linspace / norm / matmul /
atan2 / asin), because floor() amplifies any ulp difference into a different pixel.
"""
import numpy as np
import torch

from .ddim import DDIMSchedule, DiffusionTables, ddim_step, re_noise, mix_latents_with_mask, cfg_combine

VAE_SCALE = 8


def view_uv(fov, theta, phi, width, height, W, H, dtype=torch.float32):
    """panorama_tensor_utils.py:204-245: (u, v) float maps [height, width] of a perspective view on a W x H equirect."""
    fov_rad = torch.deg2rad(torch.tensor(fov, dtype=dtype))
    theta_rad = torch.deg2rad(torch.tensor(theta, dtype=dtype))
    phi_rad = torch.deg2rad(torch.tensor(phi, dtype=dtype))
    f = 0.5 * width / torch.tan(fov_rad / 2)
    x = torch.linspace(-width / 2, width / 2 - 1, steps=width, dtype=dtype)
    y = torch.linspace(-height / 2, height / 2 - 1, steps=height, dtype=dtype)
    yv, xv = torch.meshgrid(y, x, indexing="ij")
    zv = torch.full_like(xv, f)
    xyz = torch.stack([xv, yv, zv], dim=-1)
    xyz_norm = xyz / torch.norm(xyz, dim=-1, keepdim=True)
    r_phi = torch.tensor([[1, 0, 0],
                          [0, torch.cos(phi_rad), -torch.sin(phi_rad)],
                          [0, torch.sin(phi_rad), torch.cos(phi_rad)]], dtype=dtype)
    r_theta = torch.tensor([[torch.cos(theta_rad), 0, torch.sin(theta_rad)],
                            [0, 1, 0],
                            [-torch.sin(theta_rad), 0, torch.cos(theta_rad)]], dtype=dtype)
    rot = torch.matmul(r_theta, r_phi)
    xyz_rot = torch.matmul(xyz_norm.view(-1, 3), rot.t()).view(height, width, 3)
    lon = torch.atan2(xyz_rot[..., 0], xyz_rot[..., 2])
    lat = torch.asin(xyz_rot[..., 1])
    lon = (lon + 2 * torch.pi) % (2 * torch.pi)
    u = lon / (2 * torch.pi) * (W - 1)
    v = (lat + torch.pi / 2) / torch.pi * (H - 1)
    return u, v


def gather_index_map(u, v, W, H):
    """:185-202 -> (flat index int64 [h,w] into H*W, valid bool [h,w])."""
    u0 = torch.floor(u).long() % W
    v0 = torch.clamp(torch.floor(v).long(), 0, H - 1)
    valid = (u >= 0) & (u < W) & (v >= 0) & (v < H)
    return v0 * W + u0, valid


def scatter_index_map(u, v, W, H):
    """:163-171 -> (flat target index int64 [h,w], valid bool [h,w])."""
    u_int = torch.floor(u).long()
    v_int = torch.floor(v).long()
    valid = (u_int >= 0) & (u_int < W) & (v_int >= 0) & (v_int < H)
    return v_int * W + u_int, valid


def sphere_gather(pano, fov, theta, phi, width, height):
    """PanoramaLatentProxy.get_view_tensor_no_interpolate: pano [B,C,N,H,W] -> (view [B,C,N,height,width], mask [h,w])."""
    B, C, N, H, W = pano.shape
    u, v = view_uv(fov, theta, phi, width, height, W, H, pano.dtype)
    idx, valid = gather_index_map(u, v, W, H)
    flat = pano.reshape(B, C, N, H * W)
    view = flat[..., idx.view(-1)].reshape(B, C, N, height, width).clone()
    view[..., ~valid] = 0
    mask = torch.ones_like(u)
    mask[~valid] = 0
    return view, mask


def sphere_scatter(pano, view, fov, theta, phi):
    """PanoramaLatentProxy.set_view_tensor_no_interpolation: in place on pano [B,C,N,H,W]; duplicates -> last source wins."""
    B, C, N, H, W = pano.shape
    height, width = view.shape[-2], view.shape[-1]
    u, v = view_uv(fov, theta, phi, width, height, W, H, pano.dtype)
    lin, valid = scatter_index_map(u, v, W, H)
    lin, valid = lin.view(-1), valid.view(-1)
    flat = pano.reshape(B * C * N, H * W)
    src = view.reshape(B * C * N, height * width)
    order = torch.nonzero(valid).view(-1)
    for p in order.tolist():  # explicit row-major order: the last writer of a duplicated target wins
        flat[:, lin[p]] = src[:, p]
    pano.copy_(flat.reshape(pano.shape))
    return pano


def sphere_scatter_fast(pano, view, fov, theta, phi):
    """Same result as sphere_scatter via a precomputed 'winner' selection (what the HIP kernel consumes)."""
    B, C, N, H, W = pano.shape
    height, width = view.shape[-2], view.shape[-1]
    u, v = view_uv(fov, theta, phi, width, height, W, H, pano.dtype)
    lin, valid = scatter_index_map(u, v, W, H)
    lin, valid = lin.view(-1).numpy(), valid.view(-1).numpy()
    winner = winner_mask(lin, valid)
    sel = torch.from_numpy(np.nonzero(winner)[0])
    flat = pano.reshape(B * C * N, H * W)
    flat[:, torch.from_numpy(lin)[sel]] = view.reshape(B * C * N, height * width)[:, sel]
    pano.copy_(flat.reshape(pano.shape))
    return pano


def winner_mask(lin, valid):
    """bool [P]: source p writes iff it is valid and no LATER valid source has the same target."""
    last = {}
    for p in range(len(lin)):
        if valid[p]:
            last[int(lin[p])] = p
    w = np.zeros(len(lin), dtype=bool)
    for p in last.values():
        w[p] = True
    return w


def sphere_renoise_noise(shape, first_view):
    """The torch.randn_like(x_a) of re_noise (scheduler.py:106) as the t2v SPHERE loop sees it -- a layout quirk of the
    reference that decides the RNG stream:
      * PanoramaLatentProxy stores a permuted [B,N,C,H,W] VIEW of the contiguous [B,C,N,H,W] latent; views gathered
        from it come out contiguous in [B,C,N,h,w], and randn_like == torch.randn(shape) (vectorised normal fill);
      * the first set_view_tensor_no_interpolation replaces that storage by a tensor contiguous in [B,N,C,H,W]
        (panorama_tensor_utils.py:183); from then on every gathered view is a NON-contiguous permute, randn_like keeps
        its strides and torch's CPU normal_ takes the scalar path: a different stream, reproduced here by calling
        randn_like on an identically strided tensor.
    So only the very first view of a run uses the plain stream."""
    b, c, n, h, w = shape
    if first_view:
        return torch.randn(shape)
    return torch.randn_like(torch.empty((b, n, c, h, w)).permute(0, 2, 1, 3, 4))


@torch.no_grad()
def t2v_sphere_sample(eps_model, tables: DiffusionTables, cond_ctx, uncond_ctx, *, height=320, width=512, frames=16,
                      guidance_scale=7.5, equirect_width, equirect_height, phi_theta_dict, view_fov, loop_step_theta,
                      merge_renoised_overlap_latent_ratio=None, phi_fov_dict=None, denoise_to_step=None,
                      num_inference_steps=4, init_sphere_latent=None, in_channels=4, trace=None,
                      view_get_scale_factor=1, view_set_scale_factor=1, downsample_factor_before_vae_decode=None):
    """basic_sample_shift_shpere_panorama, output_type='latent' (t2v_sphere_panorama_pipeline.py:23-312).  Returns
    (final_latents, denoised) (:307-312), both resized down with 'nearest' when downsample_factor_before_vae_decode is given
    (:298-305).  view_get_scale_factor g: the view is gathered at g x the tile size and resized
    back with 'nearest' (:194-203); view_set_scale_factor s: x_prev / pred_x0 are resized up by s with 'nearest' before the
    scatter (:268-275), and so is the ones-tensor that marks the mask."""
    from .loops import resize_video_latent
    sched = DDIMSchedule(tables, num_inference_steps)
    timesteps = np.flip(sched.ddim_timesteps)
    if denoise_to_step is not None:
        timesteps = timesteps[:denoise_to_step]
    total_steps = sched.ddim_timesteps.shape[0]
    lh, lw = height // VAE_SCALE, width // VAE_SCALE
    shape = (1, in_channels, frames, equirect_height // VAE_SCALE, equirect_width // VAE_SCALE)
    pano = torch.randn(shape) if init_sphere_latent is None else init_sphere_latent.clone()
    pano_x0 = torch.zeros_like(pano)
    scattered = 0
    for i, t in enumerate(timesteps):
        theta_offset = (i % loop_step_theta) * (view_fov // loop_step_theta)
        mask = torch.zeros((1, 1, 1) + tuple(shape[3:]))
        views = []
        for phi_angle in list(phi_theta_dict.keys()):
            for theta_angle in phi_theta_dict[phi_angle]:
                cphi, cth = phi_angle, theta_angle + theta_offset
                cfov = phi_fov_dict.get(cphi, view_fov) if phi_fov_dict is not None else view_fov
                views.append((cphi, cth, cfov))
                first_view = scattered == 0
                view, _ = sphere_gather(pano, view_fov, cth, cphi, lw * view_get_scale_factor, lh * view_get_scale_factor)
                if view_get_scale_factor != 1:
                    # the resize returns a permuted view of [B,N,C,h,w] storage: randn_like then ALWAYS takes the strided
                    # path, also for the first view of a run (see sphere_renoise_noise)
                    view = resize_video_latent(view, lh, lw, "nearest")
                    first_view = False
                vmask, _ = sphere_gather(mask, cfov, cth, cphi, lw, lh)
                vmask = vmask[0, 0]  # [1,h,w]
                if merge_renoised_overlap_latent_ratio is not None and i < total_steps - 1:
                    noised = re_noise(sched, view.clone(), total_steps - i - 2, total_steps - i - 1,
                                      noise=sphere_renoise_noise(view.shape, first_view))
                    view = mix_latents_with_mask(view, noised, vmask, merge_renoised_overlap_latent_ratio)
                ts = torch.full((1,), int(t), dtype=torch.long)
                e_c = eps_model(view, ts, cond_ctx)
                e = cfg_combine(e_c, eps_model(view, ts, uncond_ctx), guidance_scale) if guidance_scale != 1.0 else e_c
                x_prev, x0 = ddim_step(sched, view, e, [total_steps - i - 1] * view.shape[2])
                ones = torch.ones((1, 1, 1, lh, lw))
                if view_set_scale_factor != 1:
                    sh, sw = lh * view_set_scale_factor, lw * view_set_scale_factor
                    x_prev, x0 = resize_video_latent(x_prev, sh, sw, "nearest"), resize_video_latent(x0, sh, sw, "nearest")
                    ones = torch.ones((1, 1, 1, sh, sw))
                sphere_scatter_fast(pano, x_prev, cfov, cth, cphi)
                scattered += 1
                sphere_scatter_fast(pano_x0, x0, cfov, cth, cphi)
                sphere_scatter_fast(mask, ones, cfov, cth, cphi)
        if trace is not None:
            trace.append((i, int(t), views))
    return _downsampled(pano.clone(), pano_x0.clone(), downsample_factor_before_vae_decode)


def _downsampled(final, denoised, factor):
    """t2v_sphere_panorama_pipeline.py:298-305 / i2v_sphere_panorama_pipeline.py:481-488."""
    from .loops import resize_video_latent
    if factor is None:
        return final, denoised
    H, W = denoised.shape[-2:]
    th, tw = int(H // factor), int(W // factor)
    return resize_video_latent(final.clone(), th, tw, "nearest"), resize_video_latent(denoised.clone(), th, tw, "nearest")


@torch.no_grad()
def i2v_sphere_sample(eps_model, image_embedder, tables: DiffusionTables, text_ctx, uncond_ctx, pano_image, *, height=320,
                      width=512, frames=16, guidance_scale=7.5, total_f=None, dock_at_f=None, overlap_ratio_list_f=None,
                      loop_step_frame=None, equirect_width, equirect_height, phi_theta_dict, view_fov, loop_step_theta,
                      merge_renoised_overlap_latent_ratio=None, merge_prev_denoised_ratio_list=None,
                      denoise_to_step=None, paste_on_static=None, static_frame_latent=None, num_inference_steps=4,
                      init_sphere_latent=None, in_channels=4, trace=None, view_get_scale_factor=1, view_set_scale_factor=1,
                      downsample_factor_before_vae_decode=None):
    """basic_sample_shift_shpere_panorama of the i2v pipeline (i2v_sphere_panorama_pipeline.py:31-495),
    output_type='latent'; returns (final_latents, denoised) (:476-495).  view_set_scale_factor s: x_prev / pred_x0 (and the ones
    that mark the mask) are resized up by s with 'nearest' before the scatters (:421-428); with a merge_prev ratio the reference
    then mixes tensors of two sizes and raises -- so does mix_latents_with_mask here.  view_get_scale_factor g: the latent view is
    gathered at g x the tile size and resized back with 'nearest' (:330-341; the mask view is not, :345-352).
    `image_embedder(crop [1,3,height,width]) -> [1,L,D]` stands for get_image_embeds, `pano_image` [3,H_img,W_img] for
    the loaded panorama image, `static_frame_latent` [1,C,1,H,W] for tiled_vae_encode_image's result (VAE: SURVEY 8-f N2);
    uncond_ctx must already contain the image-token part (:123-129)."""
    from .loops import i2v_frame_windows, resize_video_latent
    sched = DDIMSchedule(tables, num_inference_steps)
    timesteps = np.flip(sched.ddim_timesteps)
    if denoise_to_step is not None:
        timesteps = timesteps[:denoise_to_step]
    total_steps = sched.ddim_timesteps.shape[0]
    lh, lw = height // VAE_SCALE, width // VAE_SCALE
    gsf = view_get_scale_factor
    if total_f is None:
        total_f = frames
    shape = (1, in_channels, total_f, equirect_height // VAE_SCALE, equirect_width // VAE_SCALE)
    pano = torch.randn(shape) if init_sphere_latent is None else init_sphere_latent.clone()
    pano_x0 = torch.zeros_like(pano)
    img5 = pano_image[None, :, None]                       # [1,3,1,H_img,W_img]: PanoramaTensor of the image (:223)
    for i, t in enumerate(timesteps):
        theta_offset = (i % loop_step_theta) * (view_fov // loop_step_theta)
        mask = torch.zeros_like(pano)                      # reset mask record (:242), full 5-D here
        temp = None
        if paste_on_static and i < total_steps - 1:        # :245-254 (static_frame_latent may be a callable: the reference
            sfl = static_frame_latent() if callable(static_frame_latent) else static_frame_latent   # re-encodes every step)
            clear = torch.cat([sfl] * total_f, dim=2)
            temp = re_noise(sched, clear, 0, total_steps - i - 1)
        views = []
        for (fb, fe) in i2v_frame_windows(i, frames=frames, total_f=total_f, overlap_ratio_f=overlap_ratio_list_f[i],
                                          loop_step_frame=loop_step_frame, dock_at_f=dock_at_f):
            fidx = torch.arange(fb, fe) % total_f
            for phi_angle in list(phi_theta_dict.keys()):
                for theta_angle in phi_theta_dict[phi_angle]:
                    cphi, cth = phi_angle, theta_angle + theta_offset
                    views.append((fb, fe, cphi, cth))
                    view, _ = sphere_gather(pano[:, :, fidx], view_fov, cth, cphi, lw * gsf, lh * gsf)
                    if gsf != 1:
                        # the resize hands back a permuted view of [B,N,C,h,w] storage (diffusion_utils.py:27-31): clone() keeps
                        # those strides and re_noise's randn_like then takes torch's strided (scalar) normal path
                        view = resize_video_latent(view, lh, lw, "nearest")
                    prev = view.clone()
                    vmask, _ = sphere_gather(mask[:, :, fidx], view_fov, cth, cphi, lw, lh)
                    if merge_renoised_overlap_latent_ratio is not None and i < total_steps - 1:
                        # RingPanoramaLatentProxy permutes twice (ring_panorama_tensor_utils.py:269,324): its views come
                        # out contiguous in [B,C,N,h,w], so randn_like is the plain stream for EVERY view (unlike the
                        # t2v sphere loop, see sphere_renoise_noise)
                        noised = re_noise(sched, view.clone(), total_steps - i - 2, total_steps - i - 1)
                        view = mix_latents_with_mask(view, noised, vmask, merge_renoised_overlap_latent_ratio)
                    crop, _ = sphere_gather(img5, view_fov, cth, cphi, width, height)
                    ctx = torch.cat([text_ctx, image_embedder(crop[:, :, 0])], dim=1)
                    ts = torch.full((1,), int(t), dtype=torch.long)
                    e_c = eps_model(view, ts, ctx)
                    e = cfg_combine(e_c, eps_model(view, ts, uncond_ctx), guidance_scale) if guidance_scale != 1.0 else e_c
                    x_prev, x0 = ddim_step(sched, view, e, [total_steps - i - 1] * view.shape[2])
                    if view_set_scale_factor != 1:
                        sh, sw = lh * view_set_scale_factor, lw * view_set_scale_factor
                        x_prev, x0 = resize_video_latent(x_prev, sh, sw, "nearest"), resize_video_latent(x0, sh, sw, "nearest")
                    if merge_prev_denoised_ratio_list is not None and i < total_steps - 1:
                        x_prev = mix_latents_with_mask(x_prev, prev, vmask, merge_prev_denoised_ratio_list[i])
                    for dst, src in ((pano, x_prev), (temp, x_prev), (pano_x0, x0), (mask, torch.ones_like(x_prev))):
                        if dst is None:
                            continue
                        sub = dst[:, :, fidx].clone()
                        sphere_scatter_fast(sub, src, view_fov, cth, cphi)
                        dst[:, :, fidx] = sub
        if temp is not None:                               # :473-474
            pano = temp
        if trace is not None:
            trace.append((i, int(t), views))
    return _downsampled(pano.clone(), pano_x0.clone(), downsample_factor_before_vae_decode)


def sphere_splat_bilinear(pano, view, fov, theta, phi):
    """PanoramaLatentProxy.set_view_tensor_bilinear (utils/panorama_tensor_utils.py:98-152, :281-283): 4-tap splat of a
    view [B,C,N,h,w] into pano [B,C,N,H,W] with weight normaliser; in place.  Same op order as the reference
    (index_add_ per tap), hence the same fp32 sums."""
    B, C, N, H, W = pano.shape
    height, width = view.shape[-2], view.shape[-1]
    u, v = view_uv(fov, theta, phi, width, height, W, H, pano.dtype)
    u0 = torch.floor(u).long()
    v0 = torch.floor(v).long()
    u1 = (u0 + 1) % W
    v1 = torch.clamp(v0 + 1, 0, H - 1)
    du = (u - u0.float())
    dv = (v - v0.float())
    ws = [((1 - du) * (1 - dv)).view(-1), ((1 - du) * dv).view(-1), (du * (1 - dv)).view(-1), (du * dv).view(-1)]
    ids = [(v0 * W + u0).view(-1), (v1 * W + u0).view(-1), (v0 * W + u1).view(-1), (v1 * W + u1).view(-1)]
    flat = pano.reshape(B * C * N, H * W)
    src = view.reshape(B * C * N, height * width)
    for r in range(B * C * N):
        acc = torch.zeros(H * W, dtype=pano.dtype)
        wsum = torch.zeros(H * W, dtype=pano.dtype)
        for idx, w in zip(ids, ws):
            acc.index_add_(0, idx, src[r] * w)
            wsum.index_add_(0, idx, w)
        m = wsum > 0
        flat[r][m] = acc[m] / wsum[m]
    pano.copy_(flat.reshape(pano.shape))
    return pano


def sphere_grid_sample(planes, fov, theta, phi, width, height, mode="bilinear", align_corners=True):
    """get_view_tensor_interpolate (utils/panorama_tensor_utils.py:28-51): planes [B, C, H, W] -> [B, C, height, width] through
    F.grid_sample on the normalised (u, v) grid, padding_mode='border' -- the reference's own ops."""
    import torch.nn.functional as F
    B, C, H, W = planes.shape
    u, v = view_uv(fov, theta, phi, width, height, W, H, torch.float32)
    grid = torch.stack(((u / (W - 1)) * 2 - 1, (v / (H - 1)) * 2 - 1), dim=-1).unsqueeze(0).repeat(B, 1, 1, 1)
    return F.grid_sample(planes, grid, mode=mode, padding_mode="border", align_corners=align_corners)


def sphere_round_scatter(planes, view, fov, theta, phi):
    """set_view_tensor (utils/panorama_tensor_utils.py:72-96): planes [B, C, H, W], view [B, C, height, width]; round-to-nearest
    clamped targets, reshaped to [B, -1] like the reference (plane b takes the b-th chunk of the map and as many of its own first
    source pixels), scatter_ along the pixel axis.  Returns the new planes."""
    B, C, H, W = planes.shape
    height, width = view.shape[-2:]
    u, v = view_uv(fov, theta, phi, width, height, W, H, torch.float32)
    u_nn = torch.round(u).long().clamp(0, W - 1)
    v_nn = torch.round(v).long().clamp(0, H - 1)
    flat_view = view.reshape(B, C, -1)
    flat = planes.reshape(B, C, -1).clone()
    lin = (v_nn * W + u_nn).view(B, -1)
    flat.scatter_(2, lin.unsqueeze(1).expand(-1, C, -1), flat_view)
    return flat.view(B, C, H, W)
