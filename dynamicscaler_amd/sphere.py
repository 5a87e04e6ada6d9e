"""Sphere path host side: perspective-view <-> equirect index maps, GPU-backed PanoramaLatentProxy, and the t2v sphere
loop `basic_sample_shift_shpere_panorama` (pipeline/t2v_sphere_panorama_pipeline.py:23-312).

Index maps (S1): `_get_uv` (utils/panorama_tensor_utils.py:204-245) is evaluated ON THE HOST with the same fp32 torch
CPU ops the reference uses -- floor() turns any ulp difference of a device libm into a different pixel -- and cached
per (fov, theta, phi, w, h, W, H) as int32 device tensors (<= 44 views x 10 theta offsets for gen_pano_360.py).
Gather / scatter (S2/S3) are ds_map_gather / ds_map_scatter3; duplicate scatter targets are resolved on the host
(last source in row-major view order wins, like torch's CPU index_put), so the kernel is race-free.
"""
import numpy as np
import torch

from . import ops, parallel
from .pipelines import VC2_Pipeline_T2V_SpherePano as _RingPipe, _RingState
from .pipelines_i2v import VC2_Pipeline_I2V_SpherePano as _I2VRingPipe, load_image_tensor_from_path
from .ring import i2v_frame_windows


def view_uv(fov, theta, phi, width, height, W, H, dtype=torch.float32):
    """(u, v) float maps [height, width] of a pinhole view (fov, yaw theta, pitch phi, degrees) on a W x H equirect.
    Same op sequence and dtype as panorama_tensor_utils.py:204-245 (fp32 on the CPU)."""
    rad = [torch.deg2rad(torch.tensor(a, dtype=dtype)) for a in (fov, theta, phi)]
    fov_r, th, ph = rad
    focal = 0.5 * width / torch.tan(fov_r / 2)
    xs = torch.linspace(-width / 2, width / 2 - 1, steps=width, dtype=dtype)
    ys = torch.linspace(-height / 2, height / 2 - 1, steps=height, dtype=dtype)
    yv, xv = torch.meshgrid(ys, xs, indexing="ij")
    rays = torch.stack([xv, yv, torch.full_like(xv, focal)], dim=-1)
    rays = rays / torch.norm(rays, dim=-1, keepdim=True)
    r_phi = torch.tensor([[1, 0, 0], [0, torch.cos(ph), -torch.sin(ph)], [0, torch.sin(ph), torch.cos(ph)]], dtype=dtype)
    r_theta = torch.tensor([[torch.cos(th), 0, torch.sin(th)], [0, 1, 0], [-torch.sin(th), 0, torch.cos(th)]], dtype=dtype)
    rot = torch.matmul(r_theta, r_phi)
    d = torch.matmul(rays.view(-1, 3), rot.t()).view(height, width, 3)
    lon = (torch.atan2(d[..., 0], d[..., 2]) + 2 * torch.pi) % (2 * torch.pi)
    lat = torch.asin(d[..., 1])
    return lon / (2 * torch.pi) * (W - 1), (lat + torch.pi / 2) / torch.pi * (H - 1)


class ViewMaps:
    """gather / scatter index maps of one view, host (numpy) and device (int32).  gather_only: the gather map alone (the i2v loops'
    crops of the panorama IMAGE: a 512 x 320 view of a 2048 x 1024 image -- the scatter side's winner resolution and footprints
    walk arrays of the panorama's size and cost 80 ms per view there, 3.5 s per step of gen_pano_360's 44 views)."""

    def __init__(self, fov, theta, phi, width, height, W, H, device, gather_only=False):
        u, v = view_uv(fov, theta, phi, width, height, W, H)
        fu, fv = torch.floor(u).long(), torch.floor(v).long()
        g = (torch.clamp(fv, 0, H - 1) * W + fu % W)
        gvalid = (u >= 0) & (u < W) & (v >= 0) & (v < H)                       # :197-200
        g = torch.where(gvalid, g, torch.full_like(g, -1)).view(-1)
        if gather_only:
            self.gather_np = g.numpy().astype(np.int32)
            self.gather = torch.from_numpy(self.gather_np).to(device)
            self.gather_valid_mask = gvalid.to(torch.float32)
            return
        s = fv * W + fu
        svalid = ((fu >= 0) & (fu < W) & (fv >= 0) & (fv < H)).view(-1)        # :166
        s = s.view(-1).numpy()
        sv = svalid.numpy()
        # duplicate targets: the LAST valid source in row-major order wins
        last = np.full(H * W, -1, dtype=np.int64)
        src = np.nonzero(sv)[0]
        last[s[src]] = src                      # numpy fancy assignment keeps the last write for repeated indices
        winner = np.zeros(len(s), dtype=bool)
        winner[last[last >= 0]] = True
        s_final = np.where(winner, s, -1).astype(np.int32)
        self.gather_np = g.numpy().astype(np.int32)
        self.scatter_np = s_final
        self.gather = torch.from_numpy(self.gather_np).to(device)
        self.scatter = torch.from_numpy(s_final).to(device)
        # footprints for the dependency analysis
        self.read_set = np.zeros(H * W, dtype=bool)
        self.read_set[self.gather_np[self.gather_np >= 0]] = True
        self.write_set = np.zeros(H * W, dtype=bool)
        self.write_set[s_final[s_final >= 0]] = True
        self.gather_valid_mask = gvalid.to(torch.float32)


class SplatMaps:
    """Inverse (per-target CSR) form of the 4-tap bilinear splat of one view (panorama_tensor_utils.py:98-152).
    Entry order inside a target = the reference's index_add_ order: tap 00 sources ascending, then 01, 10, 11."""

    def __init__(self, fov, theta, phi, width, height, W, H, device):
        u, v = view_uv(fov, theta, phi, width, height, W, H)
        u0, v0 = torch.floor(u).long(), torch.floor(v).long()
        u1 = (u0 + 1) % W
        v1 = torch.clamp(v0 + 1, 0, H - 1)
        du, dv = u - u0.float(), v - v0.float()
        ws = [((1 - du) * (1 - dv)), ((1 - du) * dv), (du * (1 - dv)), (du * dv)]
        ids = [v0 * W + u0, v1 * W + u0, v0 * W + u1, v1 * W + u1]
        P = width * height
        tgt_all = np.concatenate([i.view(-1).numpy() for i in ids])
        w_all = np.concatenate([w.view(-1).numpy() for w in ws]).astype(np.float32)
        tap_all = np.repeat(np.arange(4), P)
        src_all = np.tile(np.arange(P), 4)
        order = np.lexsort((src_all, tap_all, tgt_all))        # by target, then tap, then source
        tgt_sorted = tgt_all[order]
        uniq, first = np.unique(tgt_sorted, return_index=True)
        row_ptr = np.append(first, len(tgt_sorted)).astype(np.int32)
        dev = device
        self.tgt = torch.from_numpy(uniq.astype(np.int32)).to(dev)
        self.row_ptr = torch.from_numpy(row_ptr).to(dev)
        self.src = torch.from_numpy(src_all[order].astype(np.int32)).to(dev)
        self.wgt = torch.from_numpy(w_all[order]).to(dev)


class RoundScatterMaps:
    """Targets of set_view_tensor (panorama_tensor_utils.py:72-96, ring_panorama_tensor_utils.py:80-104): round-to-nearest,
    clamped.  The reference reshapes the [height, width] target map to [B, -1] -- B = the panorama's leading planes (frames of
    the ring window) -- so plane b scatters only its first P/B source pixels, to the b-th chunk of the map; reproduced as is.
    idx[b][p] = target of source pixel p of plane b, or -1 (p >= P/B, or a later source of the same plane hits the same target:
    torch's CPU scatter_ walks the index dimension in order, the last writer stays)."""

    def __init__(self, fov, theta, phi, width, height, W, H, B, device):
        u, v = view_uv(fov, theta, phi, width, height, W, H)
        u_nn = torch.round(u).long().clamp(0, W - 1)
        v_nn = torch.round(v).long().clamp(0, H - 1)
        P = width * height
        lin = (v_nn * W + u_nn).view(B, -1).numpy()          # raises like the reference when B does not divide height*width
        per = lin.shape[1]
        idx = np.full((B, P), -1, dtype=np.int32)
        for b in range(B):
            last = np.full(H * W, -1, dtype=np.int64)
            last[lin[b]] = np.arange(per)                    # fancy assignment keeps the last write of a repeated index
            winner = np.zeros(per, dtype=bool)
            winner[last[last >= 0]] = True
            idx[b, :per] = np.where(winner, lin[b], -1)
        self.idx_np = idx
        self.idx = torch.from_numpy(idx).to(device)


class TapMaps:
    """F.grid_sample(pano, grid(u, v), mode, padding_mode='border', align_corners) of get_view_tensor_interpolate
    (panorama_tensor_utils.py:28-51) as taps: idx[k][p], wgt[k][p] with torch's own fp32 arithmetic for the normalised grid,
    its un-normalisation, the border clip and the corner weights (ATen GridSampler: nw, ne, sw, se).  A corner outside the panorama
    gets weight 0 (with border padding that only happens where its weight is 0 anyway)."""

    def __init__(self, fov, theta, phi, width, height, W, H, mode, align_corners, device):
        u, v = view_uv(fov, theta, phi, width, height, W, H)
        gx, gy = (u / (W - 1)) * 2 - 1, (v / (H - 1)) * 2 - 1

        def unnorm(g, size):
            x = ((g + 1) / 2) * (size - 1) if align_corners else ((g + 1) * size - 1) / 2
            return torch.clamp(x, 0, size - 1)                 # padding_mode='border'

        x, y = unnorm(gx, W).view(-1), unnorm(gy, H).view(-1)
        if mode == "bilinear":
            x0, y0 = torch.floor(x), torch.floor(y)
            x1, y1 = x0 + 1, y0 + 1
            taps = [(x0, y0, (x1 - x) * (y1 - y)), (x1, y0, (x - x0) * (y1 - y)), (x0, y1, (x1 - x) * (y - y0)), (x1, y1, (x - x0) * (y - y0))]
        elif mode == "nearest":
            taps = [(torch.round(x), torch.round(y), torch.ones_like(x))]      # nearbyint: half to even, like torch.round
        else:
            raise NotImplementedError(f"get_view_tensor_interpolate: interpolate_mode {mode!r} (bilinear and nearest are built)")
        idx, wgt = [], []
        for tx, ty, w in taps:
            ok = (tx >= 0) & (tx <= W - 1) & (ty >= 0) & (ty <= H - 1)
            idx.append(torch.where(ok, ty.long() * W + tx.long(), torch.zeros_like(tx, dtype=torch.long)))
            wgt.append(torch.where(ok, w, torch.zeros_like(w)))
        self.idx = torch.stack(idx).to(torch.int32).contiguous().to(device)
        self.wgt = torch.stack(wgt).to(torch.float32).contiguous().to(device)


class _SubsampledGather:
    """Gather map of a view taken at g x the tile size and resized back with 'nearest' (F.interpolate: source index =
    g * destination index): rows and columns 0, g, 2g, ... of the big view's map."""

    def __init__(self, big, g, height, width, HW):
        sub = big.gather_np.reshape(height * g, width * g)[::g, ::g].reshape(-1).copy()
        self.gather_np = sub
        self.gather = big.gather.view(height * g, width * g)[::g, ::g].reshape(-1).contiguous()
        self.read_set = np.zeros(HW, dtype=bool)
        self.read_set[sub[sub >= 0]] = True


class _UpsampledScatter:
    """Scatter maps of a view that is resized up by s with 'nearest' before set_view_tensor_no_interpolation
    (view_set_scale_factor, t2v_sphere_panorama_pipeline.py:268-275): pixel (y, x) of the tile is pixel (s y + dy, s x + dx) of the
    scaled view for every (dy, dx) -- s * s scatter maps of the tile's size, one per offset.  The winners of duplicated targets are
    resolved on the scaled view (`big`: the last source in ITS row-major order, what the reference's scatter does on one thread),
    so the s * s scatters hit disjoint targets and commute."""

    def __init__(self, big, s, height, width):
        m = big.scatter.view(height, s, width, s)
        self.subs = [m[:, dy, :, dx].reshape(-1).contiguous() for dy in range(s) for dx in range(s)]
        self.write_set = big.write_set


def _set_maps(cache, sub_cache, fov, theta, phi, width, height, W, H, s):
    """what a view's scatters need: a ViewMaps at the tile size (s = 1; .subs = [its scatter map]) or an _UpsampledScatter."""
    if s == 1:
        m = cache.get(fov, theta, phi, width, height, W, H)
        if not hasattr(m, "subs"):
            m.subs = [m.scatter]
        return m
    key = ("set", fov, theta, phi, s)
    r = sub_cache.get(key)
    if r is None:
        r = sub_cache[key] = _UpsampledScatter(cache.get(fov, theta, phi, width * s, height * s, W, H), s, height, width)
    return r


def _downsample_outputs(final_latents, denoised, factor):
    """t2v_sphere_panorama_pipeline.py:298-305 / i2v_sphere_panorama_pipeline.py:481-488: both outputs resized with 'nearest' to
    (H // factor, W // factor) before the decode."""
    if factor is None:
        return final_latents, denoised
    H, W = denoised.shape[-2:]
    th, tw = int(H // factor), int(W // factor)
    return ops.resize_latent(final_latents, th, tw, "nearest"), ops.resize_latent(denoised, th, tw, "nearest")


class ViewMapCache:
    """ViewMaps per (fov, theta, phi, view size, panorama size).  prefetch(): the maps of the NEXT step's views are computed on a
    worker thread while the GPU runs the current step (the theta offsets walk deterministically, so the keys are known): gen_pano_360's
    44 views cost 1.6 s of host time per step otherwise -- as much as the step's UNet evaluations -- for the first loop_step_theta
    steps.  The worker only computes host tensors (its uploads would queue behind the step's kernels on the stream); get() uploads."""

    # Maps depend only on the key (view angles, view size, panorama size): gen_pano_360's 44 views x 10 theta offsets are 440 of them.
    # Uploaded maps are shared by every cache of a device, so a later loop -- the next stage, the next run in the same process -- does
    # not rebuild them (round 6; bounded: the oldest entries leave when more than SHARED_MAX are held).
    _SHARED = {}
    SHARED_MAX = 4096

    def __init__(self, device):
        self.device = device
        self._maps = ViewMapCache._SHARED.setdefault(str(torch.device(device)), {})
        while len(self._maps) > ViewMapCache.SHARED_MAX:
            self._maps.pop(next(iter(self._maps)))
        self._host = {}            # prefetched on the host, not uploaded yet
        self._worker, self._jobs, self._pending, self._error = None, None, None, None

    def get(self, fov, theta, phi, width, height, W, H, gather_only=False):
        key = (fov, theta, phi, width, height, W, H)
        m = self._maps.get(key)
        if m is None or (not gather_only and not hasattr(m, "scatter")):
            m = self._host.pop(key, None)
            if m is not None and (gather_only or hasattr(m, "scatter")):
                m.gather = m.gather.to(self.device)
                if hasattr(m, "scatter"):
                    m.scatter = m.scatter.to(self.device)
                self._maps[key] = m
            else:
                m = self._maps[key] = ViewMaps(fov, theta, phi, width, height, W, H, self.device, gather_only=gather_only)
        return m

    def prefetch(self, requests):
        """requests: [(fov, theta, phi, width, height, W, H, gather_only)] -- computed on the worker thread, picked up by get() after
        wait().  Same op sequence on the same host: the maps are those get() would have built."""
        todo = [r for r in requests if r[:7] not in self._maps and r[:7] not in self._host]
        if not todo:
            return
        import queue
        import threading
        self.wait()
        if self._worker is None:
            # ONE long-lived daemon thread per cache: torch's CPU ops set up a thread team per calling thread, a fresh thread per step
            # pays for that every time (measured: +0.3-0.6 s per step)
            self._jobs = queue.Queue()

            def loop():
                while True:
                    job, done = self._jobs.get()
                    try:
                        for (fov, theta, phi, width, height, W, H, gather_only) in job:
                            self._host[(fov, theta, phi, width, height, W, H)] = ViewMaps(fov, theta, phi, width, height, W, H, "cpu",
                                                                                          gather_only=gather_only)
                    except BaseException as e:      # noqa: BLE001 -- handed to the thread that waits
                        self._error = e
                    done.set()
            self._worker = threading.Thread(target=loop, name="ds-view-maps", daemon=True)
            self._worker.start()
        self._pending = threading.Event()
        self._jobs.put((todo, self._pending))

    def wait(self):
        if self._pending is not None:
            self._pending.wait()
            self._pending = None
        if self._error is not None:
            e, self._error = self._error, None
            raise e


class PanoramaLatentProxy:
    """GPU-backed drop-in for utils/panorama_tensor_utils.py:249-290 (no [B,C,N,H,W] <-> [B,N,C,H,W] permute copies:
    the kernels index the latent layout directly)."""

    def __init__(self, equirect_tensor):
        assert equirect_tensor.dim() == 5, "expects [B, C, N, H, W]"
        if not equirect_tensor.is_cuda:
            raise RuntimeError("PanoramaLatentProxy lives on the GPU in this build (no CPU path)")
        H, W = equirect_tensor.shape[-2:]
        assert W == 2 * H                                            # panorama_tensor_utils.py:9
        self.equirect = equirect_tensor.clone().contiguous()
        self._cache = ViewMapCache(equirect_tensor.device)

    def get_equirect_tensor(self):
        return self.equirect

    def get_view_tensor_no_interpolate(self, fov, theta, phi, width, height):
        B, C, N, H, W = self.equirect.shape
        m = self._cache.get(fov, theta, phi, width, height, W, H)
        view = ops.map_gather(self.equirect, m.gather[None]).reshape(1, C, N, height, width)
        return view, m.gather_valid_mask.to(self.equirect.device)

    def set_view_tensor_bilinear(self, view_tensor, fov, theta, phi):
        """4-tap splat with normaliser (panorama_tensor_utils.py:98-152); unused by the pipelines, kept for parity."""
        B, C, N, H, W = self.equirect.shape
        height, width = view_tensor.shape[-2:]
        key = ("splat", fov, theta, phi, width, height, W, H)
        m = self._cache._maps.get(key)
        if m is None:
            m = self._cache._maps[key] = SplatMaps(fov, theta, phi, width, height, W, H, self.equirect.device)
        ops.map_splat_(self.equirect, view_tensor.to(self.equirect.dtype).contiguous(), m.tgt, m.row_ptr, m.src, m.wgt)

    def set_view_tensor_no_interpolation(self, view_tensor, fov, theta, phi):
        B, C, N, H, W = self.equirect.shape
        height, width = view_tensor.shape[-2:]
        m = self._cache.get(fov, theta, phi, width, height, W, H)
        src = view_tensor.to(self.equirect.dtype).contiguous()
        ops.map_scatter3(self.equirect, None, None, src, None, m.scatter[None])

    def get_view_tensor_interpolate(self, fov, theta, phi, width, height, interpolate_mode='bilinear', interpolate_align_corners=True):
        """:260-266 (F.grid_sample of every frame, panorama_tensor_utils.py:28-51) -> [1, C, N, height, width]."""
        B, C, N, H, W = self.equirect.shape
        key = ("taps", fov, theta, phi, width, height, W, H, interpolate_mode, bool(interpolate_align_corners))
        m = self._cache._maps.get(key)
        if m is None:
            m = self._cache._maps[key] = TapMaps(fov, theta, phi, width, height, W, H, interpolate_mode, bool(interpolate_align_corners),
                                                 self.equirect.device)
        return ops.map_gather_taps(self.equirect, m.idx, m.wgt).reshape(1, C, N, height, width)

    def set_view_tensor(self, view_tensor, fov, theta, phi):
        """:276-278 -> PanoramaTensor.set_view_tensor (:72-96) with the N frames as its leading planes: frame n scatters its first
        height*width/N pixels to the n-th chunk of the rounded target map (the reference's [B, -1] reshape, kept)."""
        B, C, N, H, W = self.equirect.shape
        height, width = view_tensor.shape[-2:]
        assert tuple(view_tensor.shape[:3]) == (1, C, N), f"view {tuple(view_tensor.shape)} does not match the panorama [1, {C}, {N}, ...]"
        key = ("round", fov, theta, phi, width, height, W, H, N)
        m = self._cache._maps.get(key)
        if m is None:
            m = self._cache._maps[key] = RoundScatterMaps(fov, theta, phi, width, height, W, H, N, self.equirect.device)
        src = view_tensor.to(self.equirect.dtype)[0].permute(1, 0, 2, 3).contiguous().view(N, C, 1, height * width)
        f0 = torch.arange(N, dtype=torch.int32, device=self.equirect.device)
        ops.map_scatter3_frames(self.equirect, None, None, src, None, m.idx, f0, 1)


def plan_levels_sets(reads, writes):
    """Dependency levels for items with arbitrary footprints (boolean arrays over the panorama): item j must come after
    an earlier item k when k writes something j reads or writes, or j writes something k reads.  Same guarantees as
    parallel.plan_levels."""
    level = []
    for j in range(len(reads)):
        lv = 0
        touch_j = reads[j] | writes[j]
        for k in range(j):
            if level[k] >= lv and ((writes[k] & touch_j).any() or (reads[k] & writes[j]).any()):
                lv = level[k] + 1
        level.append(lv)
    out = [[] for _ in range(max(level) + 1)] if level else []
    for j, lv in enumerate(level):
        out[lv].append(j)
    return out


class VC2_Pipeline_T2V_SpherePano(_RingPipe):
    """Adds the sphere loop to the ring pipeline class (one class in the reference, t2v_sphere_panorama_pipeline.py:20)."""

    @torch.no_grad()
    def basic_sample_shift_shpere_panorama(self, prompt=None, height=320, width=512, frames=16, fps=16, guidance_scale=7.5,
                                           num_videos_per_prompt=1, generator=None, init_sphere_latent=None,
                                           equirect_width=None, equirect_height=None, phi_theta_dict=None,
                                           phi_prompt_dict=None, view_fov=None, view_get_scale_factor=1,
                                           view_set_scale_factor=1, loop_step_theta=None,
                                           merge_renoised_overlap_latent_ratio=None, phi_fov_dict=None,
                                           denoise_to_step=None, latents=None, num_inference_steps=4, prompt_embeds=None,
                                           output_type="pil", downsample_factor_before_vae_decode=None, use_skip_time=False,
                                           skip_time_step_idx=None, progressive_skip=False, step_callback=None, **kwargs):
        """[sic] name kept from the reference.  Views are perspective crops of the 2:1 equirect latent; they are
        processed with the reference's sequential semantics (levels of views with disjoint footprints are batched).
        Returns (final_latents, denoised) for output_type='latent' (:307-312)."""
        if use_skip_time:
            raise NotImplementedError  # like the reference (:146-148)
        gsf, ssf = int(view_get_scale_factor), int(view_set_scale_factor)
        assert gsf >= 1 and gsf == view_get_scale_factor, "view_get_scale_factor must be a positive integer"
        # view_set_scale_factor s (:268-275): x_prev / pred_x0 / the mask's ones are resized up by s with 'nearest' and scattered
        # through the map of the (s h) x (s w) view.  Several neighbouring sources then share a target; the last one in row-major
        # order stays, which is what the reference's index_put_ does on one thread (with more threads its own result depends on
        # timing at the thread chunks' boundaries: tests/golden/make_golden.py g33 is generated with torch.set_num_threads(1))
        assert ssf >= 1 and ssf == view_set_scale_factor, "view_set_scale_factor must be a positive integer"
        unet_config = self.model_config["params"]["unet_config"]
        frames = self.pretrained_t2v.temporal_length if frames < 0 else frames
        prompt, text_emb, uc_emb = self._encode(prompt, prompt_embeds, guidance_scale)
        self.scheduler.make_schedule(num_inference_steps, verbose=self.verbose)
        timesteps = np.flip(self.scheduler.ddim_timesteps)
        if denoise_to_step is not None:
            timesteps = timesteps[:denoise_to_step]
        total_steps = self.scheduler.ddim_timesteps.shape[0]     # NB: the full schedule length here (:125), unlike the ring loops
        vs = self.vae_scale_factor
        lat_h, lat_w = height // vs, width // vs
        H, W = equirect_height // vs, equirect_width // vs
        shape = (1, unet_config["params"]["in_channels"], frames, H, W)
        if init_sphere_latent is None:
            init_sphere_latent = torch.randn(shape)               # host draw, reference order
        else:
            assert tuple(init_sphere_latent.shape) == shape, \
                f"[basic_sample_shift_multi_windows] init_panorama_latent shape {tuple(init_sphere_latent.shape)} does not match desired shape {shape}"
        assert W == 2 * H                                         # PanoramaTensor (:9)
        st = self._new_state(init_sphere_latent, shape, timesteps, frames, fps, lat_h, lat_w, guidance_scale, text_emb,
                             uc_emb, merge_renoised_overlap_latent_ratio, kwargs)
        st.total_steps = total_steps
        device = st.device
        mask = torch.zeros((H * W,), dtype=torch.uint8, device=device)
        cache = ViewMapCache(device)
        prompt_cache, sc_cache = {}, {}
        sched = self.scheduler
        scattered = 0
        P = lat_h * lat_w
        for i in range(len(timesteps)):
            t = timesteps[i]
            theta_offset = (i % loop_step_theta) * (view_fov // loop_step_theta)
            mask.zero_()                                          # reset mask record (:181)
            views, ctxs = [], []
            for phi_angle in list(phi_theta_dict.keys()):
                for theta_angle in phi_theta_dict[phi_angle]:
                    cphi, cth = phi_angle, theta_angle + theta_offset
                    cfov = phi_fov_dict.get(cphi, view_fov) if phi_fov_dict is not None else view_fov
                    views.append((cphi, cth, cfov))
                    if phi_prompt_dict is not None:
                        cur = phi_prompt_dict[phi_angle]
                        if cur not in prompt_cache:
                            prompt_cache[cur] = self.pretrained_t2v.get_learned_conditioning([cur]).to(device)
                        ctxs.append(prompt_cache[cur])
                    else:
                        ctxs.append(st.text_emb)
            # latent gather: view_fov (:196).  view_get_scale_factor g: the reference gathers a (g h) x (g w) view and resizes
            # it back with 'nearest' (:194-203), i.e. it keeps every g-th pixel of every g-th row: the same gather with a
            # sub-sampled index map, no extra kernel
            lat_maps = [cache.get(view_fov, th, ph, lat_w, lat_h, W, H) if gsf == 1 else
                        _SubsampledGather(cache.get(view_fov, th, ph, lat_w * gsf, lat_h * gsf, W, H), gsf, lat_h, lat_w, H * W)
                        for (ph, th, fv) in views]
            set_maps = [cache.get(fv, th, ph, lat_w, lat_h, W, H) for (ph, th, fv) in views]          # mask gather: curr_fov
            sc_maps = [_set_maps(cache, sc_cache, fv, th, ph, lat_w, lat_h, W, H, ssf) for (ph, th, fv) in views]   # scatters: curr_fov
            renoise = st.ratio is not None and i < total_steps - 1
            coef = sched.step_coefficients(total_steps - i - 1)
            self._begin_step(i, total_steps - i - 1, st.guidance_scale)
            # host noise in reference order; see scheduler.draw_renoise_noise(sphere_view=...) for the layout quirk
            noises = []
            for j in range(len(views)):
                # (with a get scale factor the view handed to re_noise is resize_video_latent's permuted output: always strided)
                nz = sched.draw_renoise_noise(st.tile_shape, "cpu", torch.float32,
                                              sphere_view="first" if (scattered + j == 0 and gsf == 1) else "later") if renoise else None
                sn = sched.draw_step_noise(st.tile_shape, "cpu", torch.float32, coef["sigma"])
                noises.append((nz, sn))
            if renoise:
                c_rn, s_rn = sched.renoise_coefficients(total_steps - i - 2, total_steps - i - 1)
            reads = [lat_maps[j].read_set | set_maps[j].read_set for j in range(len(views))]
            writes = [sc_maps[j].write_set for j in range(len(views))]
            for level in plan_levels_sets(reads, writes):
                mine = parallel.rank_share(level, st.rank, st.world)
                xp_parts, x0_parts = [], []
                tb = max(1, min(self.max_tile_batch, self.wide_tile_batch)) if self._step_precision == "wide" else self.max_tile_batch
                for s0 in range(0, len(mine), tb):
                    ids = mine[s0:s0 + tb]
                    n = len(ids)
                    g_idx = torch.stack([lat_maps[j].gather for j in ids])
                    m_idx = torch.stack([set_maps[j].gather for j in ids])
                    tiles = ops.map_gather(st.pano, g_idx).reshape((n,) + st.tile_shape[1:])
                    if renoise:
                        mt = ops.map_gather(mask, m_idx).reshape(n, 1, lat_h, lat_w).expand(n, frames, lat_h, lat_w).contiguous()
                        nz = None
                        if noises[ids[0]][0] is not None:
                            nz = torch.cat([noises[j][0] for j in ids], 0).to(device=device, dtype=st.pano.dtype)
                        ops.renoise_mix_(tiles, mt, shape, c_rn, s_rn, st.ratio, noise=nz, mask_frame0=True,
                                         seed=sched.philox_seed, offset=sched.tile_philox_offset(i, tiles[0].numel()), tile_ids=ids)
                    if st.guidance_scale != 1.0:
                        eps = self._eps(torch.cat([tiles, tiles], 0), t, [ctxs[j] for j in ids] + [st.uc_emb] * n, fps,
                                        frames, cfg_pairs=n, **st.kwargs)
                        e_c, e_u = eps[:n], eps[n:]
                    else:
                        e_c, e_u = self._eps(tiles, t, [ctxs[j] for j in ids], fps, frames, **st.kwargs), None
                    sn = None
                    if coef["sigma"] != 0.0:
                        sn = torch.cat([noises[j][1] for j in ids], 0).to(device=device, dtype=st.pano.dtype)
                    x_prev, x0 = ops.cfg_ddim(tiles, e_c, e_u, shape, st.guidance_scale, coef, sn)
                    xp_parts.append(x_prev)
                    x0_parts.append(x0)
                if st.world > 1:
                    empty = torch.empty((0,) + st.tile_shape[1:], dtype=st.pano.dtype, device=device)
                    xp_all, x0_all = parallel.exchange_level(torch.cat(xp_parts, 0) if xp_parts else empty,
                                                             torch.cat(x0_parts, 0) if x0_parts else empty, len(level))
                    order = level
                else:
                    xp_all, x0_all, order = torch.cat(xp_parts, 0), torch.cat(x0_parts, 0), mine
                xp_all, x0_all = xp_all.contiguous(), x0_all.contiguous()
                for k in range(ssf * ssf):
                    ops.map_scatter3(st.pano, st.pano_x0, mask, xp_all, x0_all, torch.stack([sc_maps[j].subs[k] for j in order]))
            scattered += len(views)
            if step_callback is not None:
                step_callback(i, int(t), views, st.pano, st.pano_x0)
        final_latents, denoised = _downsample_outputs(st.pano.clone(), st.pano_x0.clone(), downsample_factor_before_vae_decode)
        if output_type == "latent":
            return final_latents, denoised
        return self.pretrained_t2v.decode_first_stage_2DAE(denoised), denoised


def plan_levels_items(frame_sets, view_keys, pix_conflict):
    """Dependency levels for (frame window, view) items: an earlier item k constrains j when their frame sets intersect
    AND their pixel footprints conflict (pix_conflict(view_k, view_j): k writes what j touches, or j writes what k
    reads).  Same guarantees as parallel.plan_levels."""
    level = []
    for j in range(len(view_keys)):
        lv = 0
        for k in range(j):
            if level[k] >= lv and (frame_sets[k] & frame_sets[j]) and pix_conflict(view_keys[k], view_keys[j]):
                lv = level[k] + 1
        level.append(lv)
    out = [[] for _ in range(max(level) + 1)] if level else []
    for j, lv in enumerate(level):
        out[lv].append(j)
    return out


class VC2_Pipeline_I2V_SpherePano(_I2VRingPipe):
    """Adds the i2v sphere loop to the i2v ring pipeline class (one class in the reference,
    pipeline/i2v_sphere_panorama_pipeline.py:26)."""

    @torch.no_grad()
    def basic_sample_shift_shpere_panorama(self, prompt=None, img_cond_path=None, height=320, width=512, frames=16, fps=16,
                                           guidance_scale=7.5, num_videos_per_prompt=1, generator=None,
                                           init_sphere_latent=None, pano_image_path=None, total_f=None, dock_at_f=None,
                                           overlap_ratio_list_f=None, loop_step_frame=None, equirect_width=None,
                                           equirect_height=None, phi_theta_dict=None, phi_prompt_dict=None, view_fov=None,
                                           view_get_scale_factor=1, view_set_scale_factor=1, loop_step_theta=None,
                                           merge_renoised_overlap_latent_ratio=None, merge_prev_denoised_ratio_list=None,
                                           denoise_to_step=None, paste_on_static=None, latents=None,
                                           num_inference_steps=4, prompt_embeds=None, output_type="pil",
                                           downsample_factor_before_vae_decode=None, use_skip_time=False,
                                           skip_time_step_idx=None, progressive_skip=False, pano_image_tensor=None,
                                           static_frame_latent=None, step_callback=None, **kwargs):
        """[sic] name kept from the reference (i2v_sphere_panorama_pipeline.py:31-495).  Frame windows over a ring of
        total_f frames (RingPanoramaLatentProxy), per-view image tokens from the perspective crop of the panorama image,
        5-D denoised mask, merge-prev, paste_on_static.  Extensions: `pano_image_tensor` [3,H_img,W_img] instead of a path,
        `static_frame_latent` [1,C,1,H,W] = a VAE-encoded panorama image to reuse for paste_on_static; without it the tiled
        VAE encode runs every step like the reference's (:247, fresh posterior noise each time).
        Returns (final_latents, denoised) for output_type='latent' (:476-495)."""
        gsf, ssf = int(view_get_scale_factor), int(view_set_scale_factor)
        assert gsf >= 1 and gsf == view_get_scale_factor, "view_get_scale_factor must be a positive integer"
        assert ssf >= 1 and ssf == view_set_scale_factor, "view_set_scale_factor must be a positive integer"   # (see the t2v loop)
        has_vae = getattr(self.pretrained_t2v, "first_stage_model", None) is not None
        if (use_skip_time and init_sphere_latent is None) or (paste_on_static and static_frame_latent is None):
            if not has_vae:
                raise NotImplementedError("use_skip_time without init_sphere_latent / paste_on_static without "
                                          "static_frame_latent need the first-stage encoder (first_stage_config)")
        unet_config = self.model_config["params"]["unet_config"]
        frames = self.pretrained_t2v.temporal_length if frames < 0 else frames
        vs = self.vae_scale_factor
        prompt, text_emb, uc_emb = self._encode(prompt, prompt_embeds, guidance_scale)
        if guidance_scale != 1.0 and hasattr(self.pretrained_t2v, "embedder"):   # uncond image tokens (:123-129)
            uc_img = torch.zeros(1, 3, height // vs, width // vs).to(self.pretrained_t2v.device)
            uc_emb = torch.cat([uc_emb.to(uc_img.device), self.pretrained_t2v.get_image_embeds(uc_img)], dim=1)
        self.scheduler.make_schedule(num_inference_steps, verbose=self.verbose)
        timesteps = np.flip(self.scheduler.ddim_timesteps)
        if use_skip_time and not progressive_skip:                 # :143-145
            timesteps = timesteps[skip_time_step_idx:]
        if denoise_to_step is not None:
            assert not (use_skip_time and not progressive_skip), \
                "should not use denoise_to_step while using Non progressive time step skip"   # :148-149
            timesteps = timesteps[:denoise_to_step]
        total_steps = self.scheduler.ddim_timesteps.shape[0]       # the full schedule length (:163)
        if total_f is None:
            total_f = frames
        lat_h, lat_w = height // vs, width // vs
        H, W = equirect_height // vs, equirect_width // vs
        shape = (1, unet_config["params"]["in_channels"], total_f, H, W)
        image = pano_image_tensor if pano_image_tensor is not None else \
            load_image_tensor_from_path(pano_image_path, equirect_height, equirect_width)

        def encode_static():
            """tiled_vae_encode_image of the panorama image (:186, :247): the reference redoes it at every use, and its
            posterior sample draws fresh noise each time -- reproduced (same host RNG order)."""
            return self.tiled_vae_encode_image(image_tensor=image)

        if init_sphere_latent is None:
            init_sphere_latent = torch.randn(shape)                # host draw, reference order (:183)
            if use_skip_time:                                      # :184-208
                frame_0 = encode_static()
                if progressive_skip:
                    init_sphere_latent = init_sphere_latent.to(frame_0.device)
                    for frame_idx, ps in enumerate(list(reversed(range(skip_time_step_idx)))):
                        init_sphere_latent[:, :, [frame_idx]] = self.scheduler.re_noise(frame_0, 0, total_steps - ps - 1).to(init_sphere_latent.dtype)
                else:
                    init_sphere_latent = self.scheduler.re_noise(frame_0.expand(1, shape[1], total_f, H, W).contiguous(), 0,
                                                                 total_steps - 1)
        else:
            assert tuple(init_sphere_latent.shape) == shape, \
                f"[basic_sample_shift_multi_windows] init_panorama_latent shape {tuple(init_sphere_latent.shape)} does not match desired shape {shape}"
        assert W == 2 * H                                          # RingPanoramaTensor (:12)
        # the view-map caches, and step 0's maps on their way (worker thread, host tensors) BEFORE the set-up below -- weight repack, the
        # operand calibration's evaluations -- instead of in front of the first step (round 6: the first step waited 1.6 s for them)
        Himg, Wimg = image.shape[-2:]
        cache, img_cache = ViewMapCache(self._execution_device), ViewMapCache(self._execution_device)
        emb_cache = {}

        def step_views(i):
            off = (i % loop_step_theta) * (view_fov // loop_step_theta)
            return [(ph, th + off) for ph in list(phi_theta_dict.keys()) for th in phi_theta_dict[ph]]

        def prefetch_maps(i):
            """the index maps step i needs (latent views at the tile / get / set sizes, image crops), on the worker thread"""
            lat, img = [], []
            for (ph, th) in step_views(i):
                for g in sorted({1, gsf, ssf}):
                    lat.append((view_fov, th, ph, lat_w * g, lat_h * g, W, H, False))
                if (ph, th) not in emb_cache:
                    img.append((view_fov, th, ph, width, height, Wimg, Himg, True))
            cache.prefetch(lat)
            img_cache.prefetch(img)

        if len(timesteps):
            prefetch_maps(0)
        st = self._new_state(init_sphere_latent, shape, timesteps, frames, fps, lat_h, lat_w, guidance_scale, text_emb,
                             uc_emb, merge_renoised_overlap_latent_ratio, kwargs)
        st.total_steps = total_steps
        device, sched = st.device, self.scheduler
        mask = torch.zeros((total_f, H, W), dtype=torch.uint8, device=device)   # one byte per (frame, pixel)
        assert Wimg == 2 * Himg                                    # PanoramaTensor of the image (:223)
        image5 = image.to(device=device, dtype=torch.float32).reshape(1, 3, 1, Himg, Wimg).contiguous()
        static = None
        if paste_on_static and static_frame_latent is not None:
            static = static_frame_latent.to(device=device, dtype=st.pano.dtype)
            assert tuple(static.shape) == (1, shape[1], 1, H, W)
        prompt_cache, conflict_cache, sub_cache = {}, {}, {}
        P = lat_h * lat_w

        def lat_map(ph, th):
            """latent gather map of a view.  view_get_scale_factor g (:330-341): the reference gathers a (g h) x (g w) view and
            resizes it back with 'nearest', i.e. keeps every g-th pixel of every g-th row -- the same gather with a
            sub-sampled index map (the mask view and the scatters stay at the tile size, :345-352)."""
            if gsf == 1:
                return cache.get(view_fov, th, ph, lat_w, lat_h, W, H)
            r = sub_cache.get((ph, th))
            if r is None:
                r = sub_cache[(ph, th)] = _SubsampledGather(cache.get(view_fov, th, ph, lat_w * gsf, lat_h * gsf, W, H), gsf,
                                                            lat_h, lat_w, H * W)
            return r

        def set_map(ph, th):
            """scatter maps of a view: at the tile size, or through the (s h) x (s w) view with view_set_scale_factor s (:421-428)"""
            return _set_maps(cache, sub_cache, view_fov, th, ph, lat_w, lat_h, W, H, ssf)

        def pix_conflict(ka, kb):
            r = conflict_cache.get((ka, kb))
            if r is None:
                a, b = cache.get(view_fov, ka[1], ka[0], lat_w, lat_h, W, H), cache.get(view_fov, kb[1], kb[0], lat_w, lat_h, W, H)
                ra, rb = a.read_set | lat_map(ka[0], ka[1]).read_set, b.read_set | lat_map(kb[0], kb[1]).read_set
                wa, wb = set_map(ka[0], ka[1]).write_set, set_map(kb[0], kb[1]).write_set
                r = conflict_cache[(ka, kb)] = bool((wa & (rb | wb)).any() or (ra & wb).any())
            return r

        for i in range(len(timesteps)):
            t = timesteps[i]
            theta_offset = (i % loop_step_theta) * (view_fov // loop_step_theta)
            cache.wait()
            img_cache.wait()
            mask.zero_()                                           # reset mask record (:242)
            live = i < total_steps - 1
            temp = None
            if paste_on_static and live:                           # :245-254 (host randn of the whole panorama first)
                cur_static = static if static is not None else encode_static().to(device=device, dtype=st.pano.dtype)
                temp = sched.re_noise(cur_static.expand(1, shape[1], total_f, H, W).contiguous(), 0, total_steps - i - 1)
            items, ctxs = [], []
            # the crops of this step's views that are not embedded yet (a view recurs every loop_step_theta steps), through the image
            # tower in ONE pass (the reference embeds view by view, :356-366; the towers' kernels give a row the same bits in any batch)
            new = [k for k in dict.fromkeys(step_views(i)) if k not in emb_cache]
            if new:
                idx = torch.stack([img_cache.get(view_fov, th, ph, width, height, Wimg, Himg, gather_only=True).gather for (ph, th) in new])
                crops = ops.map_gather(image5, idx).reshape(len(new), 3, height, width)
                embs = self.pretrained_t2v.get_image_embeds(batch_imgs=crops.to(self.pretrained_t2v.device)).to(device)
                emb_cache.update({k: embs[j:j + 1] for j, k in enumerate(new)})
            for (fb, fe) in i2v_frame_windows(i, frames=frames, total_f=total_f, overlap_ratio_f=overlap_ratio_list_f[i],
                                              loop_step_frame=loop_step_frame, dock_at_f=dock_at_f):
                for phi_angle in list(phi_theta_dict.keys()):
                    for theta_angle in phi_theta_dict[phi_angle]:
                        cphi, cth = phi_angle, theta_angle + theta_offset
                        items.append((fb, fe, cphi, cth))
                        cur_text = st.text_emb
                        if phi_prompt_dict is not None:
                            cur = phi_prompt_dict[phi_angle]
                            if cur not in prompt_cache:
                                prompt_cache[cur] = self.pretrained_t2v.get_learned_conditioning([cur]).to(device)
                            cur_text = prompt_cache[cur]
                        ctxs.append(torch.cat([cur_text, emb_cache[(cphi, cth)].to(cur_text.dtype)], dim=1))
            maps = [cache.get(view_fov, th, ph, lat_w, lat_h, W, H) for (_, _, ph, th) in items]
            lat_maps = [lat_map(ph, th) for (_, _, ph, th) in items]
            if i + 1 < len(timesteps):
                prefetch_maps(i + 1)                               # host work of the next step, under this step's evaluations
            renoise = st.ratio is not None and live
            merge_prev = merge_prev_denoised_ratio_list[i] if (merge_prev_denoised_ratio_list is not None and live) else None
            if merge_prev is not None and ssf != 1:                # the reference mixes the scaled x_prev with the unscaled view (:430-436)
                raise RuntimeError(f"merge_prev_denoised_ratio_list with view_set_scale_factor {ssf}: the size of tensor a "
                                   f"({lat_w * ssf}) must match the size of tensor b ({lat_w}) (the reference raises here, "
                                   "i2v_sphere_panorama_pipeline.py:430-436)")
            coef = sched.step_coefficients(total_steps - i - 1)
            self._begin_step(i, total_steps - i - 1, st.guidance_scale)
            noises = []                                            # host noise in the reference's item order
            for _ in items:
                # (with a get scale factor the view handed to re_noise is resize_video_latent's permuted output: strided, so
                # randn_like takes torch's scalar normal path, scheduler.draw_renoise_noise)
                nz = sched.draw_renoise_noise(st.tile_shape, "cpu", torch.float32,
                                              sphere_view=None if gsf == 1 else "later") if renoise else None
                sn = sched.draw_step_noise(st.tile_shape, "cpu", torch.float32, coef["sigma"])
                noises.append((nz, sn))
            if renoise:
                c_rn, s_rn = sched.renoise_coefficients(total_steps - i - 2, total_steps - i - 1)
            fsets = [frozenset(f % total_f for f in range(fb, fe)) for (fb, fe, _, _) in items]
            tb = max(1, min(self.max_tile_batch, self.wide_tile_batch)) if self._step_precision == "wide" else self.max_tile_batch
            # the step's batches, and the first frames of all of them in ONE upload (a host -> device copy per batch is ordered behind
            # the kernels already queued: it would stop the host from running ahead of the GPU 30 times a step)
            plan, flat = [], []
            for level in plan_levels_items(fsets, [(ph, th) for (_, _, ph, th) in items], pix_conflict):
                mine = parallel.rank_share(level, st.rank, st.world)
                batches = [mine[s0:s0 + tb] for s0 in range(0, len(mine), tb)]
                order = level if st.world > 1 else mine
                plan.append((level, batches, order))
                for grp in batches + [order]:
                    flat += [items[j][0] for j in grp]
            f0_all = torch.tensor(flat, dtype=torch.int32, device=device) if flat else None
            f0_pos = [0]

            def first_frames(n):
                f0_pos[0] += n
                return f0_all[f0_pos[0] - n:f0_pos[0]]

            for level, batches, order in plan:
                xp_parts, x0_parts = [], []
                for ids in batches:
                    n = len(ids)
                    g_idx = torch.stack([maps[j].gather for j in ids])
                    l_idx = g_idx if gsf == 1 else torch.stack([lat_maps[j].gather for j in ids])
                    f0 = first_frames(n)
                    tiles = ops.map_gather_frames(st.pano, l_idx, f0, frames).reshape((n,) + st.tile_shape[1:])
                    prev = tiles.clone() if merge_prev is not None else None
                    mt = ops.map_gather_frames(mask, g_idx, f0, frames).reshape(n, frames, lat_h, lat_w)
                    if renoise:
                        nz = None
                        if noises[ids[0]][0] is not None:
                            nz = torch.cat([noises[j][0] for j in ids], 0).to(device=device, dtype=st.pano.dtype)
                        ops.renoise_mix_(tiles, mt, shape, c_rn, s_rn, st.ratio, noise=nz, mask_frame0=False,
                                         seed=sched.philox_seed, offset=sched.tile_philox_offset(i, tiles[0].numel()), tile_ids=ids)
                    if st.guidance_scale != 1.0:
                        eps = self._eps(torch.cat([tiles, tiles], 0), t, [ctxs[j] for j in ids] + [st.uc_emb] * n, fps,
                                        frames, cfg_pairs=n, **st.kwargs)
                        e_c, e_u = eps[:n], eps[n:]
                    else:
                        e_c, e_u = self._eps(tiles, t, [ctxs[j] for j in ids], fps, frames, **st.kwargs), None
                    sn = None
                    if coef["sigma"] != 0.0:
                        sn = torch.cat([noises[j][1] for j in ids], 0).to(device=device, dtype=st.pano.dtype)
                    x_prev, x0 = ops.cfg_ddim(tiles, e_c, e_u, shape, st.guidance_scale, coef, sn)
                    if merge_prev is not None:                     # :430-436
                        ops.renoise_mix_(x_prev, mt, shape, 0.0, 1.0, merge_prev, noise=prev, mask_frame0=False)
                    xp_parts.append(x_prev)
                    x0_parts.append(x0)
                if st.world > 1:
                    empty = torch.empty((0,) + st.tile_shape[1:], dtype=st.pano.dtype, device=device)
                    xp_all, x0_all = parallel.exchange_level(torch.cat(xp_parts, 0) if xp_parts else empty,
                                                             torch.cat(x0_parts, 0) if x0_parts else empty, len(level))
                else:
                    xp_all, x0_all = torch.cat(xp_parts, 0), torch.cat(x0_parts, 0)
                sc = [set_map(items[j][2], items[j][3]) for j in order]
                f0 = first_frames(len(order))
                xp_all, x0_all = xp_all.contiguous(), x0_all.contiguous()
                for k in range(ssf * ssf):
                    s_idx = torch.stack([m.subs[k] for m in sc])
                    ops.map_scatter3_frames(st.pano, st.pano_x0, mask, xp_all, x0_all, s_idx, f0, frames)
                    if temp is not None:                           # :446-454
                        ops.map_scatter3_frames(temp, None, None, xp_all, None, s_idx, f0, frames)
            if temp is not None:                                   # :473-474
                st.pano = temp
            if step_callback is not None:
                step_callback(i, int(t), items, st.pano, st.pano_x0)
        cache.wait()
        img_cache.wait()
        final_latents, denoised = _downsample_outputs(st.pano.clone(), st.pano_x0.clone(), downsample_factor_before_vae_decode)
        if output_type == "latent":
            return final_latents, denoised
        return self.pretrained_t2v.decode_first_stage_2DAE(denoised), denoised
