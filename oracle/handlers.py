"""TEST INFRASTRUCTURE (oracle): the reference's panorama tensor handlers restated on CPU tensors.

  PanoramaTensor            utils/panorama_tensor_utils.py:5-247   (no-interpolate gather :53-70,:185-202; floor scatter
                                                                    :154-183; 4-tap splat :98-152)
  RingLatentProxy           utils/ring_panorama_tensor_utils.py:316-337
  RingPanoramaTensor        utils/ring_panorama_tensor_utils.py:8-259 (frame windows through RingLatent, :59-78, :170-199)
  RingPanoramaLatentProxy   utils/ring_panorama_tensor_utils.py:262-314

Built on oracle.sphere (index maps, gather, last-writer scatter, splat) and oracle.ring (wrapping windows); pinned against the
reference's own classes by tests/golden/panorama_handlers.npz (tests/test_oracle_golden.py).  The index maps come from fp32
trigonometry on the host, so a map can differ in single pixels between CPU vendors: GPU tests therefore compare the product
with THIS restatement run on the same host (bit-exact) and with the golden recorded on the build host (almost everywhere).
"""
import torch

from . import ring as oring
from . import sphere as osphere


def _planes(t, H, W):
    return t.reshape(1, -1, 1, H, W)


class PanoramaTensor:
    def __init__(self, equirect_tensor):
        assert equirect_tensor.dim() >= 2
        H, W = equirect_tensor.shape[-2:]
        assert W == 2 * H
        if equirect_tensor.dim() == 2:
            equirect_tensor = equirect_tensor.unsqueeze(0)
        self.equirect_tensor = equirect_tensor.clone()
        self.C, self.H, self.W = equirect_tensor.shape[-3], H, W

    def get_view_tensor_no_interpolate(self, fov, theta, phi, width, height):
        view, mask = osphere.sphere_gather(_planes(self.equirect_tensor, self.H, self.W), fov, theta, phi, width, height)
        return view.reshape(*self.equirect_tensor.shape[:-3], self.C, height, width), mask

    def set_view_tensor_no_interpolation(self, view_tensor, fov, theta, phi):
        h, w = view_tensor.shape[-2:]
        p = _planes(self.equirect_tensor, self.H, self.W).clone()
        osphere.sphere_scatter_fast(p, view_tensor.reshape(1, -1, 1, h, w), fov, theta, phi)
        self.equirect_tensor = p.reshape(self.equirect_tensor.shape)

    def set_view_tensor_bilinear(self, view_tensor, fov, theta, phi):
        h, w = view_tensor.shape[-2:]
        p = _planes(self.equirect_tensor, self.H, self.W).clone()
        osphere.sphere_splat_bilinear(p, view_tensor.reshape(1, -1, 1, h, w), fov, theta, phi)
        self.equirect_tensor = p.reshape(self.equirect_tensor.shape)


    def get_view_tensor_interpolate(self, fov, theta, phi, width, height, interpolate_mode="bilinear", interpolate_align_corners=True):
        """:28-51"""
        lead = tuple(self.equirect_tensor.shape[:-3])
        v = osphere.sphere_grid_sample(self.equirect_tensor.reshape(-1, self.C, self.H, self.W), fov, theta, phi, width, height,
                                       interpolate_mode, interpolate_align_corners)
        return v.view(*lead, self.C, height, width) if lead else v.squeeze(0)

    def set_view_tensor(self, view_tensor, fov, theta, phi):
        """:72-96 (a panorama whose leading planes multiply to 1 comes out as [C, H, W], like the reference's)"""
        if view_tensor.dim() == 3:
            view_tensor = view_tensor.unsqueeze(0)
        lead = tuple(self.equirect_tensor.shape[:-3])
        pano = self.equirect_tensor.reshape(-1, self.C, self.H, self.W)
        B = pano.shape[0]
        out = osphere.sphere_round_scatter(pano, view_tensor.reshape(-1, self.C, *view_tensor.shape[-2:]), fov, theta, phi)
        self.equirect_tensor = out.view(*lead, self.C, self.H, self.W) if B > 1 else out.squeeze(0)


class PanoramaLatentProxy:
    """utils/panorama_tensor_utils.py:249-290: the [B, C, N, H, W] face of PanoramaTensor (the uncalled variants only; the others
    are oracle.sphere's functions)."""

    def __init__(self, equirect_tensor):
        self.panorama_tensor = PanoramaTensor(equirect_tensor.permute(0, 2, 1, 3, 4))

    def get_view_tensor_interpolate(self, fov, theta, phi, width, height, interpolate_mode="bilinear", interpolate_align_corners=True):
        return self.panorama_tensor.get_view_tensor_interpolate(fov, theta, phi, width, height, interpolate_mode,
                                                                interpolate_align_corners).permute(0, 2, 1, 3, 4).clone()

    def set_view_tensor(self, view_tensor, fov, theta, phi):
        self.panorama_tensor.set_view_tensor(view_tensor.permute(0, 2, 1, 3, 4), fov, theta, phi)

    def get_equirect_tensor(self):
        return self.panorama_tensor.equirect_tensor.permute(0, 2, 1, 3, 4)


class RingLatentProxy:
    """Frame windows run over dim 1 of the tensor passed in (the ring holds it with dims 1 and 2 swapped)."""

    def __init__(self, init_latent):
        assert init_latent.dim() >= 4
        self.ring = init_latent.permute(0, 2, 1, 3, 4).clone()

    def get_torch_latent(self):
        return self.ring.permute(0, 2, 1, 3, 4)

    def get_window_latent(self, frame_begin, frame_end):
        return oring.ring_gather(self.ring, frame_begin=frame_begin, frame_end=frame_end).permute(0, 2, 1, 3, 4)

    def get_operating_shape(self, frame_begin, frame_end):
        return self.get_window_latent(frame_begin, frame_end).shape

    def set_window_latent(self, input_latent, frame_begin, frame_end):
        oring.ring_scatter(self.ring, input_latent.permute(0, 2, 1, 3, 4), frame_begin=frame_begin, frame_end=frame_end)


class RingPanoramaTensor:
    """[1, N, C, H, W]; every get / set works on the window get_window_latent(frame_begin, frame_end) of the N axis."""

    def __init__(self, equirect_tensor):
        H, W = equirect_tensor.shape[-2:]
        assert W == 2 * H
        self.equirect_tensor_handler = RingLatentProxy(equirect_tensor)
        self.C, self.H, self.W = equirect_tensor.shape[-3], H, W

    def get_view_tensor_no_interpolate(self, fov, theta, phi, width, height, frame_begin=None, frame_end=None):
        win = self.equirect_tensor_handler.get_window_latent(frame_begin, frame_end)          # [1, nf, C, H, W]
        view, mask = osphere.sphere_gather(_planes(win, self.H, self.W), fov, theta, phi, width, height)
        return view.reshape(*win.shape[:-3], self.C, height, width), mask

    def set_view_tensor_no_interpolation(self, view_tensor, fov, theta, phi, frame_begin=None, frame_end=None):
        h, w = view_tensor.shape[-2:]
        win = self.equirect_tensor_handler.get_window_latent(frame_begin, frame_end).contiguous()
        p = _planes(win, self.H, self.W).clone()
        osphere.sphere_scatter_fast(p, view_tensor.reshape(1, -1, 1, h, w), fov, theta, phi)
        self.equirect_tensor_handler.set_window_latent(p.reshape(win.shape), frame_begin, frame_end)


    def get_view_tensor_interpolate(self, fov, theta, phi, width, height, frame_begin=None, frame_end=None,
                                    interpolate_mode="bilinear", interpolate_align_corners=True):
        """:31-57"""
        win = self.equirect_tensor_handler.get_window_latent(frame_begin, frame_end)          # [1, nf, C, H, W]
        v = osphere.sphere_grid_sample(win.reshape(-1, self.C, self.H, self.W), fov, theta, phi, width, height, interpolate_mode,
                                       interpolate_align_corners)
        return v.view(*win.shape[:-3], self.C, height, width)

    def set_view_tensor(self, view_tensor, fov, theta, phi, frame_begin=None, frame_end=None):
        """:80-104 (a one-frame window fails in the reference: the squeezed panorama cannot be permuted by set_window_latent)"""
        win = self.equirect_tensor_handler.get_window_latent(frame_begin, frame_end)
        B = win.shape[0] * win.shape[1]
        if B == 1:
            raise RuntimeError("set_view_tensor on a one-frame window: the reference fails in set_window_latent")
        out = osphere.sphere_round_scatter(win.reshape(-1, self.C, self.H, self.W), view_tensor.reshape(-1, self.C, *view_tensor.shape[-2:]),
                                           fov, theta, phi)
        self.equirect_tensor_handler.set_window_latent(out.view(win.shape), frame_begin, frame_end)

    def set_view_tensor_bilinear(self, view_tensor, fov, theta, phi, frame_begin=None, frame_end=None):
        """:107-166"""
        h, w = view_tensor.shape[-2:]
        win = self.equirect_tensor_handler.get_window_latent(frame_begin, frame_end).contiguous()
        p = _planes(win, self.H, self.W).clone()
        osphere.sphere_splat_bilinear(p, view_tensor.reshape(1, -1, 1, h, w), fov, theta, phi)
        self.equirect_tensor_handler.set_window_latent(p.reshape(win.shape), frame_begin, frame_end)


class RingPanoramaLatentProxy:
    """[1, C, N, H, W] face of RingPanoramaTensor."""

    def __init__(self, equirect_tensor):
        self.panorama_tensor = RingPanoramaTensor(equirect_tensor.permute(0, 2, 1, 3, 4))

    def get_view_tensor_no_interpolate(self, fov, theta, phi, width, height, frame_begin=None, frame_end=None):
        view, mask = self.panorama_tensor.get_view_tensor_no_interpolate(fov, theta, phi, width, height, frame_begin, frame_end)
        return view.permute(0, 2, 1, 3, 4), mask

    def set_view_tensor_no_interpolation(self, view_tensor, fov, theta, phi, frame_begin=None, frame_end=None):
        self.panorama_tensor.set_view_tensor_no_interpolation(view_tensor.permute(0, 2, 1, 3, 4), fov, theta, phi, frame_begin, frame_end)

    def get_equirect_tensor(self):
        return self.panorama_tensor.equirect_tensor_handler.get_torch_latent().permute(0, 2, 1, 3, 4)

    def get_view_tensor_interpolate(self, fov, theta, phi, width, height, interpolate_mode="bilinear", interpolate_align_corners=True,
                                    frame_begin=None, frame_end=None):
        return self.panorama_tensor.get_view_tensor_interpolate(fov, theta, phi, width, height, frame_begin, frame_end, interpolate_mode,
                                                                interpolate_align_corners).permute(0, 2, 1, 3, 4).clone()

    def set_view_tensor(self, view_tensor, fov, theta, phi, frame_begin=None, frame_end=None):
        self.panorama_tensor.set_view_tensor(view_tensor.permute(0, 2, 1, 3, 4), fov, theta, phi, frame_begin, frame_end)

    def set_view_tensor_bilinear(self, view_tensor, fov, theta, phi, frame_begin=None, frame_end=None):
        self.panorama_tensor.set_view_tensor_bilinear(view_tensor.permute(0, 2, 1, 3, 4), fov, theta, phi, frame_begin, frame_end)
