"""-m gpu: the reference's panorama tensor handlers by name (dynamicscaler_amd/panorama_tensors.py).

Data movement only, so the product equals the CPU restatement (oracle/handlers.py, pinned on the reference's own classes by
tests/golden/panorama_handlers.npz, tests/test_oracle_golden.py) BIT FOR BIT when both compute their index maps on the same
host.  Against the golden itself (maps computed on the build host) single pixels may differ -- fp32 trigonometry is not
bit-identical across CPU vendors -- so that comparison allows MAP_MISMATCH of the elements to differ."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")
MAP_MISMATCH = 0.03


def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    return torch.device("cuda:0")


def T(a):
    return torch.from_numpy(np.asarray(a))


def near(a, golden, rtol=0.0):
    """At most MAP_MISMATCH of the elements differ (by more than rtol: the splat's weights come from the same host trigonometry,
    so its sums differ in the last bits between hosts even where the index map agrees)."""
    a, golden = a.float().cpu(), T(golden).float()
    assert tuple(a.shape) == tuple(golden.shape)
    return float(((a - golden).abs() > rtol * golden.abs().clamp_min(1.0)).float().mean()) <= MAP_MISMATCH


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_panorama_tensor_vs_oracle_and_reference_golden(dtype):
    """PanoramaTensor (utils/panorama_tensor_utils.py:5-247): nearest gather + mask, floor scatter with the last-writer rule,
    4-tap splat; with leading dims, [C,H,W] and [H,W] inputs.  fp16: the same movement of the rounded values."""
    from dynamicscaler_amd.panorama_tensors import PanoramaTensor
    from oracle import handlers as oh
    d = dev()
    z = np.load(os.path.join(G, "panorama_handlers.npz"))
    views = [tuple(float(a) for a in v) for v in z["views"]]
    for tag in ("p4", "p3", "p2"):
        x = T(z[f"{tag}_x"]).to(dtype)
        h, o = PanoramaTensor(x.to(d)), oh.PanoramaTensor(x.float())
        assert tuple(h.equirect_tensor.shape) == tuple(z[f"{tag}_after_set0"].shape)
        for vi, (fov, th, ph) in enumerate(views):
            v, m = h.get_view_tensor_no_interpolate(fov, th, ph, 12, 10)
            ov, om = o.get_view_tensor_no_interpolate(fov, th, ph, 12, 10)
            assert tuple(v.shape) == tuple(z[f"{tag}_get{vi}"].shape) and v.dtype == dtype and m.dtype == dtype
            assert torch.equal(v.float().cpu(), ov) and torch.equal(m.float().cpu(), om), (tag, vi)
            if dtype == torch.float32:
                assert near(v, z[f"{tag}_get{vi}"]) and near(m, z[f"{tag}_mask{vi}"])
        for vi, (fov, th, ph) in enumerate(views):
            src = T(z[f"{tag}_src{vi}"]).to(dtype)
            h.set_view_tensor_no_interpolation(src.to(d), fov, th, ph)
            o.set_view_tensor_no_interpolation(src.float(), fov, th, ph)
            assert torch.equal(h.equirect_tensor.float().cpu(), o.equirect_tensor), (tag, vi)
            if dtype == torch.float32:
                assert near(h.equirect_tensor, z[f"{tag}_after_set{vi}"])
        if dtype == torch.float32:           # the splat's weighted sums: fp32 only (bit-exact in the reference's index_add_ order)
            src = T(z[f"{tag}_splat_src"])
            h.set_view_tensor_bilinear(src.to(d), 90.0, 45.0, -30.0)
            o.set_view_tensor_bilinear(src, 90.0, 45.0, -30.0)
            assert torch.equal(h.equirect_tensor.cpu(), o.equirect_tensor), tag
            assert near(h.equirect_tensor, z[f"{tag}_after_splat"], rtol=1e-4)
    with pytest.raises(NotImplementedError):
        h.get_view_tensor_interpolate(90.0, 0.0, 0.0, 12, 10, interpolate_mode="bicubic")
    with pytest.raises(AssertionError):
        PanoramaTensor(torch.zeros((3, 16, 30), device=d))            # W == 2H (:9)
    with pytest.raises(RuntimeError):
        PanoramaTensor(torch.zeros((3, 16, 32)))                      # no CPU path


def test_ring_handlers_vs_oracle_and_reference_golden():
    """RingLatentProxy / RingPanoramaTensor / RingPanoramaLatentProxy (utils/ring_panorama_tensor_utils.py): wrapping frame
    windows (incl. the default full window), scatter into a wrapped window."""
    from dynamicscaler_amd.panorama_tensors import RingLatentProxy, RingPanoramaTensor, RingPanoramaLatentProxy
    from oracle import handlers as oh
    d = dev()
    z = np.load(os.path.join(G, "panorama_handlers.npz"))
    views = [tuple(float(a) for a in v) for v in z["views"]]
    r = RingLatentProxy(T(z["rl_x"]).to(d))                            # pure ring arithmetic: no trigonometry, golden bit-exact
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
    for tag, cls, ocls in (("rp", RingPanoramaTensor, oh.RingPanoramaTensor), ("rpl", RingPanoramaLatentProxy, oh.RingPanoramaLatentProxy)):
        h, o = cls(T(z[f"{tag}_x"]).to(d)), ocls(T(z[f"{tag}_x"]))
        for vi, ((fov, th, ph), (fb, fe)) in enumerate(zip(views, windows)):
            v, m = h.get_view_tensor_no_interpolate(fov, th, ph, 12, 10, frame_begin=fb, frame_end=fe)
            ov, om = o.get_view_tensor_no_interpolate(fov, th, ph, 12, 10, frame_begin=fb, frame_end=fe)
            assert tuple(v.shape) == tuple(z[f"{tag}_get{vi}"].shape)
            assert torch.equal(v.cpu(), ov) and torch.equal(m.cpu(), om), (tag, vi)
            assert near(v, z[f"{tag}_get{vi}"]) and near(m, z[f"{tag}_mask{vi}"])
        for vi, ((fov, th, ph), (fb, fe)) in enumerate(zip(views, windows)):
            src = T(z[f"{tag}_src{vi}"])
            h.set_view_tensor_no_interpolation(src.to(d), fov, th, ph, frame_begin=fb, frame_end=fe)
            o.set_view_tensor_no_interpolation(src, fov, th, ph, frame_begin=fb, frame_end=fe)
            full = h.get_equirect_tensor() if tag == "rpl" else h.equirect_tensor_handler.get_torch_latent()
            ofull = o.get_equirect_tensor() if tag == "rpl" else o.equirect_tensor_handler.get_torch_latent()
            assert torch.equal(full.cpu(), ofull), (tag, vi)
            assert near(full, z[f"{tag}_after_set{vi}"])


INTERP_TOL = 2e-6        # fp32 4-tap sums: the product adds the taps in ATen's corner order, F.grid_sample's vector code may fuse


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_uncalled_handler_methods_vs_oracle_and_reference_golden(dtype):
    """The methods no pipeline of the reference calls (SURVEY 8-a S5's list), on all five classes: get_view_tensor_interpolate
    (utils/panorama_tensor_utils.py:28-51: grid_sample taps resolved on the host, ds_map_gather_taps), set_view_tensor (:72-96:
    round-to-nearest scatter_ with the reference's [B, -1] reshape of the target map) and the ring-backed set_view_tensor_bilinear
    (utils/ring_panorama_tensor_utils.py:107-166).  Scatter / splat: bit-equal to the oracle on this host; interpolation: fp32
    rounding of a 4-term sum.  Where the reference raises, so does the product."""
    import json
    from dynamicscaler_amd.panorama_tensors import PanoramaTensor, PanoramaLatentProxy, RingPanoramaTensor, RingPanoramaLatentProxy
    from oracle import handlers as oh
    d = dev()
    z = np.load(os.path.join(G, "panorama_handlers_uncalled.npz"))
    raised = json.loads(bytes(z["raised_json"]).decode())
    views = [tuple(float(a) for a in v) for v in z["views"]]
    modes = [("bilinear", True), ("bilinear", False), ("nearest", True)]
    f32 = dtype == torch.float32
    tol = INTERP_TOL if f32 else 2e-3

    def close(a, b, t):
        a, b = a.float().cpu(), b.float().cpu()
        return tuple(a.shape) == tuple(b.shape) and float((a - b).abs().max()) <= t * max(1.0, float(b.abs().max()))

    for tag in ("p4", "p3", "p2", "p5"):
        x = T(z[f"{tag}_x"]).to(dtype)
        h, o = PanoramaTensor(x.to(d)), oh.PanoramaTensor(x.float())
        for vi, (fov, th, ph) in enumerate(views):
            for mi, (mode, ac) in enumerate(modes):
                v = h.get_view_tensor_interpolate(fov, th, ph, 12, 10, mode, ac)
                assert v.dtype == dtype and close(v, o.get_view_tensor_interpolate(fov, th, ph, 12, 10, mode, ac), tol), (tag, vi, mi)
                if f32 and mode == "bilinear":
                    assert near(v, z[f"{tag}_interp{vi}_{mi}"], rtol=1e-4)     # other host's trigonometry: weights differ in the last bits, single taps flip
        for vi, (fov, th, ph) in enumerate(views):
            src = T(z[f"{tag}_src{vi}"]).to(dtype)
            h.set_view_tensor(src.to(d), fov, th, ph)
            o.set_view_tensor(src.float(), fov, th, ph)
            assert tuple(h.equirect_tensor.shape) == tuple(o.equirect_tensor.shape) == tuple(z[f"{tag}_after_set{vi}"].shape)
            assert torch.equal(h.equirect_tensor.float().cpu(), o.equirect_tensor), (tag, vi)
            if f32:
                assert near(h.equirect_tensor, z[f"{tag}_after_set{vi}"])
    x = T(z["pl_x"]).to(dtype)
    h, o = PanoramaLatentProxy(x.to(d)), oh.PanoramaLatentProxy(x.float())
    for vi, (fov, th, ph) in enumerate(views):
        assert close(h.get_view_tensor_interpolate(fov, th, ph, 12, 10), o.get_view_tensor_interpolate(fov, th, ph, 12, 10), tol)
    for vi, (fov, th, ph) in enumerate(views):
        src = T(z[f"pl_src{vi}"]).to(dtype)
        h.set_view_tensor(src.to(d), fov, th, ph)
        o.set_view_tensor(src.float(), fov, th, ph)
        assert torch.equal(h.get_equirect_tensor().float().cpu(), o.get_equirect_tensor()), vi
        if f32:
            assert near(h.get_equirect_tensor(), z[f"pl_after_set{vi}"])
    wins = ((3, 7), (None, None), (4, 9), (2, 3))
    for tag, cls, ocls in (("rp", RingPanoramaTensor, oh.RingPanoramaTensor), ("rpl", RingPanoramaLatentProxy, oh.RingPanoramaLatentProxy)):
        x = T(z[f"{tag}_x"]).to(dtype)
        h, o = cls(x.to(d)), ocls(x.float())
        full = (lambda: h.get_equirect_tensor()) if tag == "rpl" else (lambda: h.equirect_tensor_handler.get_torch_latent())
        ofull = (lambda: o.get_equirect_tensor()) if tag == "rpl" else (lambda: o.equirect_tensor_handler.get_torch_latent())
        for vi, ((fov, th, ph), (fb, fe)) in enumerate(zip(views + views[:1], wins)):
            for mi, (mode, ac) in enumerate(modes[:2] if vi else modes):
                kw = dict(frame_begin=fb, frame_end=fe, interpolate_mode=mode, interpolate_align_corners=ac)
                v = h.get_view_tensor_interpolate(fov, th, ph, 12, 10, **kw)
                assert close(v, o.get_view_tensor_interpolate(fov, th, ph, 12, 10, **kw), tol), (tag, vi, mi)
                if f32 and mode == "bilinear":
                    assert near(v, z[f"{tag}_interp{vi}_{mi}"], rtol=1e-4)
        for vi, ((fov, th, ph), (fb, fe)) in enumerate(zip(views + views[:1], wins)):
            src = T(z[f"{tag}_src{vi}"]).to(dtype)
            if f"{tag}_set{vi}" in raised:
                with pytest.raises(RuntimeError):
                    h.set_view_tensor(src.to(d), fov, th, ph, frame_begin=fb, frame_end=fe)
                continue
            h.set_view_tensor(src.to(d), fov, th, ph, frame_begin=fb, frame_end=fe)
            o.set_view_tensor(src.float(), fov, th, ph, frame_begin=fb, frame_end=fe)
            assert torch.equal(full().float().cpu(), ofull()), (tag, vi)
            if f32:
                assert near(full(), z[f"{tag}_after_set{vi}"])
        if f32:
            for vi, ((fov, th, ph), (fb, fe)) in enumerate(zip(views + views[:1], wins)):
                src = T(z[f"{tag}_splat_src{vi}"])
                h.set_view_tensor_bilinear(src.to(d), fov, th, ph, frame_begin=fb, frame_end=fe)
                o.set_view_tensor_bilinear(src, fov, th, ph, frame_begin=fb, frame_end=fe)
                assert torch.equal(full().cpu(), ofull()), (tag, vi)
                assert near(full(), z[f"{tag}_after_splat{vi}"], rtol=1e-4)
