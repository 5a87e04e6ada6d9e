"""-m gpu: the reference's panorama tensor handlers by name (dynamicscaler_amd/panorama_tensors.py) against vectors recorded
from the reference's own classes (tests/golden/panorama_handlers.npz, make_golden.py g24) -- data movement only, so every
result is bit-exact."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")


def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    return torch.device("cuda:0")


def T(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_panorama_tensor_vs_reference(dtype):
    """PanoramaTensor (utils/panorama_tensor_utils.py:5-247): nearest gather + mask, floor scatter with the last-writer rule,
    4-tap splat; with leading dims, [C,H,W] and [H,W] inputs.  fp16: the same movement of the rounded values."""
    from dynamicscaler_amd.panorama_tensors import PanoramaTensor
    d = dev()
    z = np.load(os.path.join(G, "panorama_handlers.npz"))
    views = [tuple(float(a) for a in v) for v in z["views"]]
    q = (lambda t: t.to(dtype).float()) if dtype == torch.float16 else (lambda t: t)
    for tag in ("p4", "p3", "p2"):
        x = T(z[f"{tag}_x"])
        h = PanoramaTensor(x.to(d, dtype))
        assert tuple(h.equirect_tensor.shape) == tuple(z[f"{tag}_after_set0"].shape)
        for vi, (fov, th, ph) in enumerate(views):
            v, m = h.get_view_tensor_no_interpolate(fov, th, ph, 12, 10)
            assert tuple(v.shape) == tuple(z[f"{tag}_get{vi}"].shape) and v.dtype == dtype and m.dtype == dtype
            assert torch.equal(v.float().cpu(), q(T(z[f"{tag}_get{vi}"]))) and torch.equal(m.float().cpu(), T(z[f"{tag}_mask{vi}"]))
        if dtype == torch.float16:
            continue                      # sets in fp16 move rounded values: checked through the fp32 path's index maps
        for vi, (fov, th, ph) in enumerate(views):
            h.set_view_tensor_no_interpolation(T(z[f"{tag}_src{vi}"]).to(d), fov, th, ph)
            assert torch.equal(h.equirect_tensor.cpu(), T(z[f"{tag}_after_set{vi}"])), (tag, vi)
        h.set_view_tensor_bilinear(T(z[f"{tag}_splat_src"]).to(d), 90.0, 45.0, -30.0)
        assert torch.equal(h.equirect_tensor.cpu(), T(z[f"{tag}_after_splat"])), tag
    with pytest.raises(NotImplementedError):
        h.get_view_tensor_interpolate(90.0, 0.0, 0.0, 12, 10)
    with pytest.raises(AssertionError):
        PanoramaTensor(torch.zeros((3, 16, 30), device=d))            # W == 2H (:9)
    with pytest.raises(RuntimeError):
        PanoramaTensor(torch.zeros((3, 16, 32)))                      # no CPU path


def test_ring_handlers_vs_reference():
    """RingLatentProxy / RingPanoramaTensor / RingPanoramaLatentProxy (utils/ring_panorama_tensor_utils.py): wrapping frame
    windows (incl. a window longer than the ring and the default full window), scatter into a wrapped window."""
    from dynamicscaler_amd.panorama_tensors import RingLatentProxy, RingPanoramaTensor, RingPanoramaLatentProxy
    d = dev()
    z = np.load(os.path.join(G, "panorama_handlers.npz"))
    views = [tuple(float(a) for a in v) for v in z["views"]]
    r = RingLatentProxy(T(z["rl_x"]).to(d))
    assert torch.equal(r.get_window_latent(3, 8).cpu(), T(z["rl_win_3_8"]))
    assert torch.equal(r.get_window_latent(None, None).cpu(), T(z["rl_win_none"]))
    assert torch.equal(r.get_window_latent(1, 10).cpu(), T(z["rl_win_1_10"]))
    assert tuple(r.get_operating_shape(3, 8)) == tuple(int(a) for a in z["rl_shape_3_8"])
    r.set_window_latent(T(z["rl_src"]).to(d), 4, 7)
    assert torch.equal(r.get_torch_latent().cpu(), T(z["rl_after_set"]))
    with pytest.raises(AssertionError):
        r.get_window_latent(3, 11)                                    # frame_end <= 2 * size (shift_window_utils.py:75)
    with pytest.raises(AssertionError):
        r.set_window_latent(torch.zeros((1, 6, 3, 16, 32), device=d), 0, 6)   # "warp should not occur"
    windows = ((3, 7), (None, None), (4, 9))
    for tag, cls in (("rp", RingPanoramaTensor), ("rpl", RingPanoramaLatentProxy)):
        h = cls(T(z[f"{tag}_x"]).to(d))
        for vi, ((fov, th, ph), (fb, fe)) in enumerate(zip(views, windows)):
            v, m = h.get_view_tensor_no_interpolate(fov, th, ph, 12, 10, frame_begin=fb, frame_end=fe)
            assert tuple(v.shape) == tuple(z[f"{tag}_get{vi}"].shape)
            assert torch.equal(v.cpu(), T(z[f"{tag}_get{vi}"])) and torch.equal(m.cpu(), T(z[f"{tag}_mask{vi}"])), (tag, vi)
        for vi, ((fov, th, ph), (fb, fe)) in enumerate(zip(views, windows)):
            h.set_view_tensor_no_interpolation(T(z[f"{tag}_src{vi}"]).to(d), fov, th, ph, frame_begin=fb, frame_end=fe)
            full = h.get_equirect_tensor() if tag == "rpl" else h.equirect_tensor_handler.get_torch_latent()
            assert torch.equal(full.cpu(), T(z[f"{tag}_after_set{vi}"])), (tag, vi)
        with pytest.raises(NotImplementedError):
            h.set_view_tensor_bilinear(None, 0, 0, 0)
