"""The reference's panorama tensor handlers by name, GPU-resident (SURVEY.md 8-b import-by-name surface):

    PanoramaTensor            utils/panorama_tensor_utils.py:5-247
    PanoramaLatentProxy       utils/panorama_tensor_utils.py:249-290          (defined in sphere.py, re-exported here)
    RingLatentProxy           utils/ring_panorama_tensor_utils.py:316-337
    RingPanoramaTensor        utils/ring_panorama_tensor_utils.py:8-259
    RingPanoramaLatentProxy   utils/ring_panorama_tensor_utils.py:262-314

Same constructors, methods, argument meaning, shapes and assertion behaviour as the reference classes; the data movement is the
HIP map kernels (ds_map_gather / ds_map_scatter3 / ds_map_splat and their frame-window forms) on index maps computed on the
host with the reference's own fp32 torch-CPU op sequence (sphere.view_uv) -- bit-exact, including the last-writer rule of
torch's CPU index_put for duplicate scatter targets.  The sphere pipelines (sphere.py) call the same kernels directly; these
classes exist so that code written against the reference's `utils.*` modules keeps working (dropin.install()).

No permute copies: the reference reorders [B,C,N,H,W] <-> [B,N,C,H,W] around every call; here one latent-layout buffer
[1,C,N,H,W] is addressed in place and the reordered shapes are returned as views.

The variants no pipeline of the reference calls are built too (round 4): get_view_tensor_interpolate (F.grid_sample resolved into
taps on the host, ds_map_gather_taps), set_view_tensor (round-to-nearest scatter_, including the reference's [B, -1] reshape of
the target map, sphere.RoundScatterMaps) and the ring-backed set_view_tensor_bilinear (ds_map_splat per channel and contiguous
frame run of the window).
"""
import torch

from . import ops
from .ring import RingLatent
from .sphere import PanoramaLatentProxy, RoundScatterMaps, SplatMaps, TapMaps, ViewMapCache

__all__ = ["PanoramaTensor", "PanoramaLatentProxy", "RingLatentProxy", "RingPanoramaTensor", "RingPanoramaLatentProxy"]


def _need_gpu(t, who):
    if not t.is_cuda:
        raise RuntimeError(f"{who} lives on the GPU in this build (no CPU path); pass a HIP tensor")


def _shape_info(equirect_tensor):
    """(tensor with >= 3 dims, C, H, W) exactly as panorama_tensor_utils.py:6-17."""
    assert equirect_tensor.dim() >= 2
    H, W = equirect_tensor.shape[-2], equirect_tensor.shape[-1]
    assert W == 2 * H
    if equirect_tensor.dim() == 2:
        equirect_tensor = equirect_tensor.unsqueeze(0)
    return equirect_tensor, equirect_tensor.shape[-3], H, W


class PanoramaTensor:
    """utils/panorama_tensor_utils.py:5-247 on a [*, C, H, W] (or [C,H,W] / [H,W]) equirect tensor, W == 2H."""

    def __init__(self, equirect_tensor):
        equirect_tensor, C, H, W = _shape_info(equirect_tensor)
        _need_gpu(equirect_tensor, "PanoramaTensor")
        self.equirect_tensor = equirect_tensor.clone().contiguous()
        self.C, self.H, self.W = C, H, W
        self.device, self.dtype = equirect_tensor.device, equirect_tensor.dtype
        self._cache = ViewMapCache(self.device)

    def _planes(self):
        return self.equirect_tensor.view(1, -1, 1, self.H, self.W)          # every (leading, channel) pair is one plane

    def get_view_tensor_no_interpolate(self, fov, theta, phi, width, height):
        """-> (view [*, C, height, width] ([C, height, width] without leading dims), mask [height, width] of the tensor's
        dtype: 1 where the ray hits the panorama, 0 elsewhere; :53-70, :185-202)."""
        m = self._cache.get(fov, theta, phi, width, height, self.W, self.H)
        view = ops.map_gather(self._planes(), m.gather[None])
        lead = tuple(self.equirect_tensor.shape[:-3])
        return view.reshape(*lead, self.C, height, width), m.gather_valid_mask.to(self.device, self.dtype)

    def set_view_tensor_no_interpolation(self, view_tensor, fov, theta, phi):
        """:154-183: floor() targets, sources whose target falls outside the panorama are skipped, the last source in row-major
        view order wins a duplicate target."""
        if view_tensor.dim() == 3:
            view_tensor = view_tensor.unsqueeze(0)
        height, width = view_tensor.shape[-2:]
        assert view_tensor.numel() == self.equirect_tensor.numel() // (self.H * self.W) * height * width, \
            f"view {tuple(view_tensor.shape)} does not match the panorama's leading dims {tuple(self.equirect_tensor.shape[:-2])}"
        m = self._cache.get(fov, theta, phi, width, height, self.W, self.H)
        src = view_tensor.to(self.device, self.dtype).contiguous()
        ops.map_scatter3(self._planes(), None, None, src, None, m.scatter[None])

    def set_view_tensor_bilinear(self, view_tensor, fov, theta, phi):
        """:98-152: 4-tap splat with the weight normaliser."""
        if view_tensor.dim() == 3:
            view_tensor = view_tensor.unsqueeze(0)
        height, width = view_tensor.shape[-2:]
        key = ("splat", fov, theta, phi, width, height, self.W, self.H)
        m = self._cache._maps.get(key)
        if m is None:
            m = self._cache._maps[key] = SplatMaps(fov, theta, phi, width, height, self.W, self.H, self.device)
        src = view_tensor.to(self.device, self.dtype).contiguous().view(1, -1, 1, height, width)
        ops.map_splat_(self._planes(), src, m.tgt, m.row_ptr, m.src, m.wgt)

    def _cached(self, key, make):
        m = self._cache._maps.get(key)
        if m is None:
            m = self._cache._maps[key] = make()
        return m

    def get_view_tensor_interpolate(self, fov, theta, phi, width, height, interpolate_mode='bilinear', interpolate_align_corners=True):
        """:28-51: F.grid_sample(pano, grid, mode, padding_mode='border', align_corners) -> [*, C, height, width]."""
        m = self._cached(("taps", fov, theta, phi, width, height, self.W, self.H, interpolate_mode, bool(interpolate_align_corners)),
                         lambda: TapMaps(fov, theta, phi, width, height, self.W, self.H, interpolate_mode, bool(interpolate_align_corners), self.device))
        view = ops.map_gather_taps(self._planes(), m.idx, m.wgt)
        lead = tuple(self.equirect_tensor.shape[:-3])
        return view.reshape(*lead, self.C, height, width)

    def set_view_tensor(self, view_tensor, fov, theta, phi):
        """:72-96: round-to-nearest targets, scatter_ (last source wins).  With B > 1 leading planes the reference reshapes the target
        map to [B, -1]: plane b scatters its first height*width/B pixels to the b-th chunk of the map -- kept.  Like the reference,
        a panorama without (or with only unit) leading dims ends up as [C, H, W]."""
        if view_tensor.dim() == 3:
            view_tensor = view_tensor.unsqueeze(0)
        lead = tuple(self.equirect_tensor.shape[:-3])
        B = 1
        for n in lead:
            B *= int(n)
        height, width = view_tensor.shape[-2:]
        assert view_tensor.numel() == B * self.C * height * width, \
            f"view {tuple(view_tensor.shape)} does not match the panorama's leading dims {lead} x {self.C} channels"
        m = self._cached(("round", fov, theta, phi, width, height, self.W, self.H, B),
                         lambda: RoundScatterMaps(fov, theta, phi, width, height, self.W, self.H, B, self.device))
        src = view_tensor.to(self.device, self.dtype).contiguous().view(B, 1, self.C, 1, height * width)
        pano = self.equirect_tensor.view(B, 1, self.C, 1, self.H, self.W)
        for b in range(B):
            ops.map_scatter3(pano[b], None, None, src[b], None, m.idx[b:b + 1])
        if B == 1:
            self.equirect_tensor = self.equirect_tensor.view(self.C, self.H, self.W)


class RingLatentProxy:
    """utils/ring_panorama_tensor_utils.py:316-337: a RingLatent over the input with dims 1 and 2 swapped, so the ring's frame
    windows (`frame_begin` / `frame_end`, wrapping, < 2x the size) run over dim 1 of what the caller passes and gets back."""

    def __init__(self, init_latent):
        assert init_latent.dim() >= 4
        self.original_shape = init_latent.shape
        B, C, N, H, W = self.original_shape
        self.managed_ring_latent = RingLatent(init_latent.permute(0, 2, 1, 3, 4))

    def get_torch_latent(self):
        return self.managed_ring_latent.torch_latent.permute(0, 2, 1, 3, 4)

    def get_operating_shape(self, frame_begin, frame_end):
        s = self.managed_ring_latent.get_shape()
        fb = 0 if frame_begin is None else frame_begin
        fe = s[2] if frame_end is None else frame_end
        self.managed_ring_latent._window(None, None, None, None, frame_begin, frame_end)     # the reference's asserts
        return torch.Size((s[0], fe - fb, s[1], s[3], s[4]))

    def get_window_latent(self, frame_begin, frame_end):
        return self.managed_ring_latent.get_window_latent(frame_begin=frame_begin, frame_end=frame_end).permute(0, 2, 1, 3, 4)

    def set_window_latent(self, input_latent, frame_begin, frame_end):
        self.managed_ring_latent.set_window_latent(input_latent=input_latent.permute(0, 2, 1, 3, 4), frame_begin=frame_begin,
                                                   frame_end=frame_end)


class RingPanoramaTensor:
    """utils/ring_panorama_tensor_utils.py:8-259 on [B=1, N, C, H, W]: PanoramaTensor whose gets / sets work on a wrapping
    window of the N axis.  The storage is the RingLatentProxy's ring buffer ([1, C, N, H, W], the latent layout), which the
    frame-window map kernels address in place (frame index modulo N inside the kernel)."""

    def __init__(self, equirect_tensor):
        equirect_tensor, C, H, W = _shape_info(equirect_tensor)
        _need_gpu(equirect_tensor, "RingPanoramaTensor")
        assert equirect_tensor.dim() == 5 and equirect_tensor.shape[0] == 1, \
            "RingPanoramaTensor: [1, N, C, H, W] (RingLatent needs 5 dims; batch 1 like the reference's pipelines)"
        self.equirect_tensor_handler = RingLatentProxy(init_latent=equirect_tensor)
        self.C, self.H, self.W = C, H, W
        self.device, self.dtype = equirect_tensor.device, equirect_tensor.dtype
        self._cache = ViewMapCache(self.device)
        self._f0 = {}

    def _store(self):
        return self.equirect_tensor_handler.managed_ring_latent.torch_latent           # [1, C, N, H, W]

    def _frames(self, frame_begin, frame_end):
        ring = self.equirect_tensor_handler.managed_ring_latent
        _, _, _, _, fb, fe = ring._window(None, None, None, None, frame_begin, frame_end)
        t = self._f0.get(fb)
        if t is None:
            t = self._f0[fb] = torch.tensor([fb], dtype=torch.int32, device=self.device)
        return fb, fe, t

    def get_view_tensor_no_interpolate(self, fov, theta, phi, width, height, frame_begin=None, frame_end=None):
        """-> (view [1, frames, C, height, width], mask [height, width])  (:59-78)."""
        fb, fe, f0 = self._frames(frame_begin, frame_end)
        m = self._cache.get(fov, theta, phi, width, height, self.W, self.H)
        view = ops.map_gather_frames(self._store(), m.gather[None], f0, fe - fb)        # [1, C, tf, P]
        view = view.reshape(1, self.C, fe - fb, height, width).permute(0, 2, 1, 3, 4)
        return view, m.gather_valid_mask.to(self.device, self.dtype)

    def set_view_tensor_no_interpolation(self, view_tensor, fov, theta, phi, frame_begin=None, frame_end=None):
        """view [1, frames, C, height, width]  (:170-199; the window is then written back through RingLatent.set_window_latent,
        whose "warp should not occur" assert limits it to the N axis' size)."""
        if view_tensor.dim() == 3:
            view_tensor = view_tensor.unsqueeze(0)
        fb, fe, f0 = self._frames(frame_begin, frame_end)
        N = self._store().shape[2]
        assert fe - fb <= N, "warp should not occur"
        height, width = view_tensor.shape[-2:]
        assert tuple(view_tensor.shape[-4:-2]) == (fe - fb, self.C), \
            f"view {tuple(view_tensor.shape)} does not match a window of {fe - fb} frames x {self.C} channels"
        m = self._cache.get(fov, theta, phi, width, height, self.W, self.H)
        src = view_tensor.reshape(1, fe - fb, self.C, height, width).permute(0, 2, 1, 3, 4).to(self.device, self.dtype).contiguous()
        ops.map_scatter3_frames(self._store(), None, None, src, None, m.scatter[None], f0, fe - fb)

    def _cached(self, key, make):
        m = self._cache._maps.get(key)
        if m is None:
            m = self._cache._maps[key] = make()
        return m

    def get_view_tensor_interpolate(self, fov, theta, phi, width, height, frame_begin=None, frame_end=None,
                                    interpolate_mode='bilinear', interpolate_align_corners=True):
        """:31-57: grid_sample of the frame window -> [1, frames, C, height, width]."""
        fb, fe, _ = self._frames(frame_begin, frame_end)
        m = self._cached(("taps", fov, theta, phi, width, height, self.W, self.H, interpolate_mode, bool(interpolate_align_corners)),
                         lambda: TapMaps(fov, theta, phi, width, height, self.W, self.H, interpolate_mode, bool(interpolate_align_corners), self.device))
        N = self._store().shape[2]
        view = ops.map_gather_taps(self._store(), m.idx, m.wgt, f0=fb % N, tf=fe - fb)          # [1, C, tf, P]
        return view.reshape(1, self.C, fe - fb, height, width).permute(0, 2, 1, 3, 4)

    def set_view_tensor(self, view_tensor, fov, theta, phi, frame_begin=None, frame_end=None):
        """:80-104: round-to-nearest scatter_ into the frame window, with the reference's [B, -1] reshape of the target map (B = frames
        of the window: frame b scatters its first height*width/B pixels to the b-th chunk of the map).  A one-frame window fails in
        the reference (the squeezed panorama no longer has the five dims set_window_latent permutes) -- refused here as well."""
        if view_tensor.dim() == 3:
            view_tensor = view_tensor.unsqueeze(0)
        fb, fe, _ = self._frames(frame_begin, frame_end)
        tf = fe - fb
        N = self._store().shape[2]
        if tf == 1:
            raise RuntimeError("RingPanoramaTensor.set_view_tensor: a one-frame window is squeezed to [C, H, W] by the reference "
                               "(ring_panorama_tensor_utils.py:103) and its set_window_latent then fails; use a window of >= 2 frames")
        assert tf <= N, "warp should not occur"
        height, width = view_tensor.shape[-2:]
        assert view_tensor.numel() == tf * self.C * height * width, \
            f"view {tuple(view_tensor.shape)} does not match a window of {tf} frames x {self.C} channels"
        m = self._cached(("round", fov, theta, phi, width, height, self.W, self.H, tf),
                         lambda: RoundScatterMaps(fov, theta, phi, width, height, self.W, self.H, tf, self.device))
        f0 = torch.tensor([(fb + b) % N for b in range(tf)], dtype=torch.int32, device=self.device)
        src = view_tensor.to(self.device, self.dtype).contiguous().view(tf, self.C, 1, height * width)
        ops.map_scatter3_frames(self._store(), None, None, src, None, m.idx, f0, 1)

    def set_view_tensor_bilinear(self, view_tensor, fov, theta, phi, frame_begin=None, frame_end=None):
        """:107-166: the 4-tap splat with normaliser on every (frame of the window, channel) plane."""
        if view_tensor.dim() == 3:
            view_tensor = view_tensor.unsqueeze(0)
        fb, fe, _ = self._frames(frame_begin, frame_end)
        tf = fe - fb
        store = self._store()
        N = store.shape[2]
        assert tf <= N, "warp should not occur"
        height, width = view_tensor.shape[-2:]
        assert view_tensor.numel() == tf * self.C * height * width, \
            f"view {tuple(view_tensor.shape)} does not match a window of {tf} frames x {self.C} channels"
        m = self._cached(("splat", fov, theta, phi, width, height, self.W, self.H),
                         lambda: SplatMaps(fov, theta, phi, width, height, self.W, self.H, self.device))
        src = view_tensor.to(self.device, self.dtype).reshape(tf, self.C, height, width).permute(1, 0, 2, 3).contiguous()   # [C, tf, h, w]
        runs = [(fb % N, 0, min(tf, N - fb % N))]                      # (first panorama frame, first window frame, frames)
        if runs[0][2] < tf:
            runs.append((0, runs[0][2], tf - runs[0][2]))
        for c in range(self.C):
            for f_p, f_w, n in runs:
                ops.map_splat_(store[:, c:c + 1, f_p:f_p + n], src[c:c + 1, f_w:f_w + n].unsqueeze(0), m.tgt, m.row_ptr, m.src, m.wgt)


class RingPanoramaLatentProxy:
    """utils/ring_panorama_tensor_utils.py:262-314: the latent-layout ([B=1, C, N, H, W]) face of RingPanoramaTensor, as the i2v
    sphere loop uses it (pipeline/i2v_sphere_panorama_pipeline.py:218-220, 330-336, 438-471)."""

    def __init__(self, equirect_tensor):
        assert equirect_tensor.dim() >= 4, "expects [B, C, N, H, W]"
        self.original_shape = equirect_tensor.shape
        B, C, N, H, W = self.original_shape
        self.panorama_tensor = RingPanoramaTensor(equirect_tensor.permute(0, 2, 1, 3, 4))

    def get_view_tensor_no_interpolate(self, fov, theta, phi, width, height, frame_begin=None, frame_end=None):
        view, mask = self.panorama_tensor.get_view_tensor_no_interpolate(fov, theta, phi, width, height,
                                                                         frame_begin=frame_begin, frame_end=frame_end)
        return view.permute(0, 2, 1, 3, 4), mask

    def set_view_tensor_no_interpolation(self, view_tensor, fov, theta, phi, frame_begin=None, frame_end=None):
        self.panorama_tensor.set_view_tensor_no_interpolation(view_tensor.permute(0, 2, 1, 3, 4), fov, theta, phi,
                                                              frame_begin=frame_begin, frame_end=frame_end)

    def get_equirect_tensor(self):
        return self.panorama_tensor.equirect_tensor_handler.get_torch_latent().permute(0, 2, 1, 3, 4)

    def get_view_tensor_interpolate(self, fov, theta, phi, width, height, interpolate_mode='bilinear', interpolate_align_corners=True,
                                    frame_begin=None, frame_end=None):
        """:271-280 (the proxy's own argument order: the interpolation options come before the frame window)"""
        view = self.panorama_tensor.get_view_tensor_interpolate(fov, theta, phi, width, height, frame_begin=frame_begin, frame_end=frame_end,
                                                                interpolate_mode=interpolate_mode,
                                                                interpolate_align_corners=interpolate_align_corners)
        return view.permute(0, 2, 1, 3, 4)

    def set_view_tensor(self, view_tensor, fov, theta, phi, frame_begin=None, frame_end=None):
        """:292-297"""
        self.panorama_tensor.set_view_tensor(view_tensor.permute(0, 2, 1, 3, 4), fov, theta, phi, frame_begin=frame_begin, frame_end=frame_end)

    def set_view_tensor_bilinear(self, view_tensor, fov, theta, phi, frame_begin=None, frame_end=None):
        """:299-304"""
        self.panorama_tensor.set_view_tensor_bilinear(view_tensor.permute(0, 2, 1, 3, 4), fov, theta, phi,
                                                      frame_begin=frame_begin, frame_end=frame_end)
