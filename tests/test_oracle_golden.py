"""Pin the oracle (CPU restatement, oracle/) against golden vectors captured from the reference
itself by tests/golden/make_golden.py.  CPU only."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from oracle import ring as oring
from oracle import ddim as oddim
from oracle import loops as oloops
from oracle.unet import unet_forward
from dynamicscaler_amd.unet_spec import param_shapes
from dynamicscaler_amd.synth import synth_state_dict, synth_normal

G = os.path.join(os.path.dirname(__file__), "golden")


def npz(name):
    return np.load(os.path.join(G, name))


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_g1_ring_segments():
    for case in json.load(open(os.path.join(G, "ring_segments.json"))):
        segs = oring.ring_segments(case["begin"], case["end"], case["size"])
        assert [list(s) for s in segs] == case["slices"]
        assert [b - a for a, b in segs] == case["sizes"]


def test_g2_ring_gather_scatter_bit_exact():
    z = npz("ring_latent.npz")
    pano = T(z["pano"])
    wins = z["windows"]
    for i, (l, r, t, d, fb, fe) in enumerate(wins.tolist()):
        got = oring.ring_gather(pano, l, r, t, d, fb, fe)
        assert torch.equal(got, T(z[f"get_{i}"]))
    assert torch.equal(oring.ring_gather(pano, 10, 32, 0, 16, 0, 12), T(z["get_multiwrap"]))
    from dynamicscaler_amd.synth import synth_normal
    p2 = pano.clone()
    for i, (l, r, t, d, fb, fe) in enumerate(wins.tolist()):
        tile = synth_normal((1, 4, fe - fb, d - t, r - l), seed=100 + i)
        oring.ring_scatter(p2, tile, l, r, t, d, fb, fe)
        assert torch.equal(p2, T(z[f"set_after_{i}"]))


def test_g2_ring_asserts_like_reference():
    pano = torch.zeros(1, 4, 6, 8, 16)
    with pytest.raises(AssertionError):
        oring.ring_gather(pano, 10, 42, 0, 8, 0, 6)      # > 2*W  (shift_window_utils.py:73)
    with pytest.raises(AssertionError):
        oring.ring_scatter(pano, torch.zeros(1, 4, 6, 8, 17), 0, 17, 0, 8, 0, 6)  # self-overlap (:145)
    with pytest.raises(AssertionError):
        oring.ring_scatter(pano, torch.zeros(1, 4, 6, 8, 3), 0, 4, 0, 8, 0, 6)    # shape mismatch (:190)


def test_g3_mix_bit_exact():
    z = npz("mix.npz")
    l1, l2, m3, m5 = T(z["l1"]), T(z["l2"]), T(z["m3"]), T(z["m5"])
    for r in (1, 1.0, 0.5, 0.3):
        assert torch.equal(oddim.mix_latents_with_mask(l1, l2, m3, r), T(z[f"out3_{r}"]))
        assert torch.equal(oddim.mix_latents_with_mask(l1, l2, m5, r), T(z[f"out5_{r}"]))


def test_g4_scheduler_tables_and_steps_bit_exact():
    z = npz("scheduler.npz")
    tables = oddim.DiffusionTables()
    assert torch.equal(tables.alphas_cumprod, T(z["alphas_cumprod"]))
    assert torch.equal(tables.betas, T(z["betas"]))
    for n in (4, 48, 50):
        s = oddim.DDIMSchedule(tables, n)
        assert np.array_equal(s.ddim_timesteps, z[f"ts_{n}"])
        assert torch.equal(s.ddim_alphas, T(z[f"alphas_{n}"]))
        assert np.array_equal(np.asarray(s.ddim_alphas_prev, dtype=np.float64), z[f"alphas_prev_{n}"])
        assert np.array_equal(np.asarray(s.ddim_sigmas, dtype=np.float64), z[f"sigmas_{n}"])
        assert torch.equal(torch.as_tensor(s.ddim_sqrt_one_minus_alphas), T(z[f"sqrt1m_{n}"]))
    x, e = T(z["x"]), T(z["e"])
    for n, eta in ((50, 0.0), (4, 0.0), (50, 1.0)):
        s = oddim.DDIMSchedule(tables, n, eta=eta)
        for index in (0, n // 2, n - 1):
            torch.manual_seed(777)
            xp, x0 = oddim.ddim_step(s, x, e, [index] * 4)
            assert torch.equal(xp, T(z[f"step_{n}_{eta}_{index}_xprev"]))
            assert torch.equal(x0, T(z[f"step_{n}_{eta}_{index}_x0"]))
            # the scalar-coefficient form the HIP kernel consumes reproduces the same numbers (eta=0)
            if eta == 0.0:
                c = s.step_coefficients(index)
                f = torch.float32
                x0b = (x - torch.tensor(c["sqrt_one_minus_at"], dtype=f) * e) / torch.tensor(c["sqrt_at"], dtype=f)
                xpb = torch.tensor(c["sqrt_a_prev"], dtype=f) * x0b + torch.tensor(c["dir_coef"], dtype=f) * e
                assert torch.equal(x0b, x0)
                assert torch.equal(xpb + 0.0, xp)
        for (a, b) in ((0, 1), (n - 2, n - 1), (n // 2 - 1, n // 2)):
            torch.manual_seed(778)
            assert torch.equal(oddim.re_noise(s, x, a, b), T(z[f"renoise_{n}_{eta}_{a}_{b}"]))


@pytest.mark.parametrize("name", ["t2v", "i2v"])
def test_g8_unet_tiny(name):
    z = npz(f"unet_tiny_{name}.npz")
    params = json.loads(bytes(z["params_json"]).decode())
    sd = synth_state_dict(param_shapes(params), seed=5)
    flat = torch.cat([v.flatten() for _, v in sorted(sd.items())])
    assert hashlib.sha256(flat.numpy().tobytes()).digest() == bytes(z["weights_sha"])
    for case in range(3):
        eps = unet_forward(sd, params, T(z[f"x_{case}"]), T(z[f"t_{case}"]), T(z[f"ctx_{case}"]),
                           fps=int(z[f"fps_{case}"]))
        ref = T(z[f"eps_{case}"])
        assert eps.shape == ref.shape
        err = float((eps - ref).abs().max()) / float(ref.abs().max())
        assert err < 2e-5, err  # fp32; differs only by summation order inside ATen


def _tiny_setup():
    z = npz("unet_tiny_t2v.npz")
    params = json.loads(bytes(z["params_json"]).decode())
    sd = synth_state_dict(param_shapes(params), seed=5)
    return params, sd


def _fake_eps(x, ts, ctx):
    return 0.1 * x + 0.01 * ctx.mean()


def test_g9_small_loops_fake_eps_bit_exact_and_traces():
    z = npz("loops_small.npz")
    meta = json.load(open(os.path.join(G, "loops_small_traces.json")))
    cond, uncond = T(z["cond"]), T(z["uncond"])
    tables = oddim.DiffusionTables()
    torch.manual_seed(2333333)
    den, _ = oloops.t2v_basic_sample(_fake_eps, tables, cond, uncond, height=64, width=128, frames=4,
                                     guidance_scale=7.5, num_inference_steps=4)
    assert torch.equal(den, T(z["basic_fake"]))
    for gname, geom in meta["geoms"].items():
        trace = []
        torch.manual_seed(2333333)
        den, _, _ = oloops.t2v_ring_sample(_fake_eps, tables, cond, uncond, guidance_scale=7.5,
                                           trace=trace, **geom)
        assert torch.equal(den, T(z[f"ring_{gname}_fake"])), gname
        ref_trace = meta["traces"][gname]
        assert len(trace) == len(ref_trace)
        for (i, t, wins), ref in zip(trace, ref_trace):
            assert i == ref["i"] and t == ref["t"]
            assert [list(w) for w in wins] == ref["windows"], (gname, i)


def test_g9_small_loops_tiny_unet():
    z = npz("loops_small.npz")
    meta = json.load(open(os.path.join(G, "loops_small_traces.json")))
    cond, uncond = T(z["cond"]), T(z["uncond"])
    params, sd = _tiny_setup()
    tables = oddim.DiffusionTables()

    def eps(x, ts, ctx):
        return unet_forward(sd, params, x, ts, ctx, fps=8)

    torch.manual_seed(2333333)
    den, _ = oloops.t2v_basic_sample(eps, tables, cond, uncond, height=64, width=128, frames=4,
                                     guidance_scale=7.5, num_inference_steps=4)
    ref = T(z["basic_tiny"])
    assert float((den - ref).abs().max()) / float(ref.abs().max()) < 1e-4
    for gname in ("grid4x2", "overlapw"):
        torch.manual_seed(2333333)
        den, _, _ = oloops.t2v_ring_sample(eps, tables, cond, uncond, guidance_scale=7.5, **meta["geoms"][gname])
        ref = T(z[f"ring_{gname}_tiny"])
        assert float((den - ref).abs().max()) / float(ref.abs().max()) < 1e-4, gname


def test_g19_multi_prompt_ring_loop_vs_reference_golden():
    """R13: per-window prompt selection on the toy dock geometry (fake eps bit-exact, tiny UNet 1e-4), and the reference's
    own factor assert once a window wraps in H (grid4x2)."""
    z = npz("loops_multiprompt.npz")
    meta = json.load(open(os.path.join(G, "loops_multiprompt.json")))
    geoms = json.load(open(os.path.join(G, "loops_small_traces.json")))["geoms"]
    emb = {"a prompt": T(z["emb_a_prompt"]), "": T(z["emb_empty"]), "sky": T(z["emb_sky"]), "ground": T(z["emb_ground"])}
    mp = {float(k): v for k, v in meta["multi_prompt_dict"].items()}
    tables = oddim.DiffusionTables()
    kw = dict(guidance_scale=7.5, window_multi_prompt_dict=mp, get_learned_conditioning=lambda p: emb[p[0]])
    trace = []
    torch.manual_seed(2333333)
    den, _, _ = oloops.t2v_ring_sample(_fake_eps, tables, emb["a prompt"], emb[""], trace=trace, **kw, **geoms[meta["geom"]])
    assert torch.equal(den, T(z["ring_dock_multiprompt_fake"]))
    assert [[list(w) for w in wins] for _, _, wins in trace] == [s["windows"] for s in meta["trace"]]
    params, sd = _tiny_setup()
    torch.manual_seed(2333333)
    den, _, _ = oloops.t2v_ring_sample(lambda x, ts, ctx: unet_forward(sd, params, x, ts, ctx, fps=8), tables, emb["a prompt"],
                                       emb[""], **kw, **geoms[meta["geom"]])
    ref = T(z["ring_dock_multiprompt_tiny"])
    assert float((den - ref).abs().max()) / float(ref.abs().max()) < 1e-4
    assert meta["grid4x2_raises"] is not None
    with pytest.raises(AssertionError, match="not legal"):
        torch.manual_seed(2333333)
        oloops.t2v_ring_sample(_fake_eps, tables, emb["a prompt"], emb[""], **kw, **geoms["grid4x2"])


def test_g9_baseline_geometry_traces():
    """Window coordinates of BASELINE configs 2/3/5 (bit-exact index sequences) -- arithmetic only."""
    data = json.load(open(os.path.join(G, "loop_traces.json")))
    for name, rec in data.items():
        geom = dict(rec["geom"])
        n = geom.pop("num_inference_steps")
        for step in rec["trace"]:
            wins = oloops.t2v_ring_windows(step["i"], height=geom["height"], width=geom["width"],
                                           frames=geom["frames"], total_h=geom["total_h"], total_w=geom["total_w"],
                                           num_windows_h=geom["num_windows_h"], num_windows_w=geom["num_windows_w"],
                                           num_windows_f=geom["num_windows_f"], loop_step=geom["loop_step"])
            assert [list(w) for w in wins] == step["windows"], (name, step["i"])
    # the probe of SURVEY.md 8-a R10: config-3 step 1
    w = oloops.t2v_ring_windows(1, height=320, width=512, frames=16, total_h=512, total_w=4096,
                                num_windows_h=2, num_windows_w=8, num_windows_f=1, loop_step=8)
    assert w[0] == (8, 72, 3, 43, 0, 16) and w[1] == (8, 72, 27, 67, 0, 16) and w[-1] == (456, 520, 27, 67, 0, 16)


@pytest.mark.parametrize("name", ["cfg3_overlap_nw10"])
def test_g9_baseline_geometry_final_panorama_sha(name):
    """Full-size panorama loop with the fake eps-model: final pred_x0 panorama hash equals the reference's."""
    rec = json.load(open(os.path.join(G, "loop_traces.json")))[name]
    z = npz("loops_small.npz")
    cond, uncond = T(z["cond"]), T(z["uncond"])
    torch.manual_seed(2333333)
    den, _, _ = oloops.t2v_ring_sample(_fake_eps, oddim.DiffusionTables(), cond, uncond, guidance_scale=7.5,
                                       **rec["geom"])
    assert list(den.shape) == rec["shape"]
    assert hashlib.sha256(den.numpy().tobytes()).hexdigest() == rec["denoised_sha256"]


# ------------------------------------------------------------------------------------------------ P4 / P3
def test_g32_grid_loop_random_shuffle_init_bit_exact():
    """random_shuffle_init_frame_stride (pipeline/t2v_normal_pipeline.py:328-337): the oracle's restatement against the reference's own run
    under the same `random.seed` (make_golden.py g32) -- the shuffle of the init latent's slices, bug-for-bug on dim 3."""
    import random
    z = npz("loops_grid_shuffle.npz")
    geom = json.loads(bytes(z["geom_json"]).decode())
    cond, uncond = T(z["cond"]), T(z["uncond"])
    torch.manual_seed(2333333)
    random.seed(int(z["random_seed"]))
    den, _ = oloops.t2v_grid_sample(_fake_eps, oddim.DiffusionTables(), cond, uncond, height=64, width=128, frames=4, guidance_scale=7.5, **geom)
    assert torch.equal(den, T(z["denoised"]))


def test_g11_grid_loop_fake_eps_bit_exact_and_traces():
    z = npz("loops_grid_i2v.npz")
    meta = json.load(open(os.path.join(G, "loops_grid_i2v_traces.json")))
    cond, uncond = T(z["cond"]), T(z["uncond"])
    for gname, geom in meta["grid_geoms"].items():
        trace = []
        torch.manual_seed(2333333)
        den, _ = oloops.t2v_grid_sample(_fake_eps, oddim.DiffusionTables(), cond, uncond, height=64, width=128, frames=4,
                                        guidance_scale=7.5, trace=trace, **geom)
        assert torch.equal(den, T(z[f"grid_{gname}_fake"])), gname
        for (i, t, wins), ref in zip(trace, meta["traces"][f"grid_{gname}"]):
            assert i == ref["i"] and t == ref["t"] and [list(w) for w in wins] == ref["windows"], (gname, i)


def test_g11_i2v_ring_loop_fake_eps_bit_exact_and_traces():
    from helpers import synth_image_embedder, i2v_geom
    z = npz("loops_grid_i2v.npz")
    meta = json.load(open(os.path.join(G, "loops_grid_i2v_traces.json")))
    cond, uncond = T(z["cond"]), T(z["uncond"])
    embed = synth_image_embedder(64)
    uc = torch.cat([uncond, embed(torch.zeros(1, 3, 8, 16))], dim=1)
    for gname, geom in meta["i2v_geoms"].items():
        geom = i2v_geom(geom)
        trace = []
        torch.manual_seed(2333333)
        den, _, _ = oloops.i2v_ring_sample(_fake_eps, embed, oddim.DiffusionTables(), cond, uc, T(z["pano_img"]),
                                           guidance_scale=7.5, trace=trace, **geom)
        assert torch.equal(den, T(z[f"i2v_{gname}_fake"])), gname
        for (i, t, wins), ref in zip(trace, meta["traces"][f"i2v_{gname}"]):
            assert i == ref["i"] and t == ref["t"] and [list(w) for w in wins] == ref["windows"], (gname, i)


def test_g20_cfg4_i2v_geometry_trace_and_panorama_sha():
    """BASELINE config 4's geometry (i2v ring 4096x512x16f, 8x2 windows, 93-token contexts) through the oracle's i2v ring
    loop with the fake eps-model: window trace and SHA-256 of the final pred-x0 panorama equal the reference's."""
    import hashlib
    from helpers import synth_image_embedder
    from dynamicscaler_amd.synth import synth_normal
    rec = json.load(open(os.path.join(G, "loop_trace_cfg4_i2v.json")))
    cond, uncond = synth_normal((1, 77, 64), 61), synth_normal((1, 77, 64), 62)
    embed = synth_image_embedder(rec["embedder_dim"])
    pano_img = synth_normal((3, 512, 4096), rec["pano_img_seed"]).clamp(-1, 1)
    uc = torch.cat([uncond, embed(torch.zeros(1, 3, 40, 64))], dim=1)
    trace = []
    torch.manual_seed(2333333)
    den, _, _ = oloops.i2v_ring_sample(_fake_eps, embed, oddim.DiffusionTables(), cond, uc, pano_img, guidance_scale=7.5,
                                       trace=trace, **rec["geom"])
    assert list(den.shape) == rec["shape"]
    for (i, t, wins), ref in zip(trace, rec["trace"]):
        assert i == ref["i"] and t == ref["t"] and [list(w) for w in wins] == ref["windows"], i
    assert hashlib.sha256(den.numpy().tobytes()).hexdigest() == rec["denoised_sha256"]


def test_g11_grid_and_i2v_tiny_unet():
    from helpers import synth_image_embedder
    z = npz("loops_grid_i2v.npz")
    meta = json.load(open(os.path.join(G, "loops_grid_i2v_traces.json")))
    cond, uncond = T(z["cond"]), T(z["uncond"])
    params, sd = _tiny_setup()
    torch.manual_seed(2333333)
    den, _ = oloops.t2v_grid_sample(lambda x, ts, ctx: unet_forward(sd, params, x, ts, ctx, fps=8),
                                    oddim.DiffusionTables(), cond, uncond, height=64, width=128, frames=4,
                                    guidance_scale=7.5, **meta["grid_geoms"]["plain"])
    ref = T(z["grid_plain_tiny"])
    assert float((den - ref).abs().max()) / float(ref.abs().max()) < 1e-4
    p2 = dict(params)
    p2["use_image_attention"] = True
    sd2 = synth_state_dict(param_shapes(p2), seed=5)
    embed = synth_image_embedder(64)
    uc = torch.cat([uncond, embed(torch.zeros(1, 3, 8, 16))], dim=1)
    torch.manual_seed(2333333)
    den, _, _ = oloops.i2v_ring_sample(lambda x, ts, ctx: unet_forward(sd2, p2, x, ts, ctx, fps=8), embed,
                                       oddim.DiffusionTables(), cond, uc, T(z["pano_img"]), guidance_scale=7.5,
                                       **meta["i2v_geoms"]["ring"])
    ref = T(z["i2v_ring_tiny"])
    assert float((den - ref).abs().max()) / float(ref.abs().max()) < 1e-4


# ------------------------------------------------------------------------------------------------ sphere path
def test_g12_sphere_index_maps_bit_exact():
    from oracle import sphere as S
    z = npz("sphere.npz")
    for tag, (W, H, w, h) in {"small": (128, 64, 16, 8), "real": (256, 128, 64, 40)}.items():
        views = z[f"maps_{tag}_views"]
        for n, (phi, th) in enumerate(views.tolist()):
            u, v = S.view_uv(120, th, phi, w, h, W, H)
            gi, gv = S.gather_index_map(u, v, W, H)
            si, sv = S.scatter_index_map(u, v, W, H)
            assert bool(gv.all()) and bool(sv.all())
            assert np.array_equal(gi.numpy().astype(np.int32), z[f"maps_{tag}_gather"][n]), (tag, phi, th)
            assert np.array_equal(si.numpy().astype(np.int32), z[f"maps_{tag}_scatter"][n]), (tag, phi, th)


def test_g12_sphere_gather_scatter_roundtrip_bit_exact():
    from oracle import sphere as S
    from dynamicscaler_amd.synth import synth_normal
    z = npz("sphere.npz")
    pano = T(z["rt_pano"])
    n = 0
    while f"rt_args_{n}" in z:
        fov, th, ph = z[f"rt_args_{n}"].tolist()
        view, _ = S.sphere_gather(pano, fov, th, ph, 16, 8)
        assert torch.equal(view, T(z[f"rt_view_{n}"]))
        tile = synth_normal((1, 4, 3, 8, 16), 200 + n)
        assert torch.equal(S.sphere_scatter(pano.clone(), tile, fov, th, ph), T(z[f"rt_after_{n}"]))
        assert torch.equal(S.sphere_scatter_fast(pano.clone(), tile, fov, th, ph), T(z[f"rt_after_{n}"]))
        n += 1
    assert n == 6


def _sphere_geom(geom):
    g = dict(geom)
    g["phi_theta_dict"] = {int(k): v for k, v in g["phi_theta_dict"].items()}
    if "phi_fov_dict" in g:
        g["phi_fov_dict"] = {int(k): v for k, v in g["phi_fov_dict"].items()}
    return g


def test_g12_sphere_loop_fake_eps_bit_exact():
    from oracle import sphere as S
    z = npz("sphere.npz")
    meta = json.load(open(os.path.join(G, "sphere_traces.json")))
    cond, uncond = T(z["cond"]), T(z["uncond"])
    for gname, geom in meta["geoms"].items():
        trace = []
        torch.manual_seed(2333333)
        final, den = S.t2v_sphere_sample(_fake_eps, oddim.DiffusionTables(), cond, uncond, guidance_scale=7.5, trace=trace,
                                         **_sphere_geom(geom))
        assert torch.equal(final, T(z[f"sphere_{gname}_fake_final"])), gname
        assert torch.equal(den, T(z[f"sphere_{gname}_fake_denoised"])), gname
        for (i, t, views), ref in zip(trace, meta["traces"][gname]):
            assert i == ref["i"] and t == ref["t"] and [list(v) for v in views] == ref["views"], (gname, i)


def test_g21_sphere_loop_view_get_scale_factor_bit_exact():
    """view_get_scale_factor 2 / 3 of the t2v sphere loop (the view gathered at g x the tile size, resized back with 'nearest';
    its re_noise noise always takes the strided randn_like path) -- bit-exact against the reference's panoramas."""
    from oracle import sphere as S
    z = npz("sphere_scale.npz")
    meta = json.load(open(os.path.join(G, "sphere_scale.json")))
    cond, uncond = T(z["cond"]), T(z["uncond"])
    for gname, geom in meta["geoms"].items():
        torch.manual_seed(2333333)
        final, den = S.t2v_sphere_sample(_fake_eps, oddim.DiffusionTables(), cond, uncond, guidance_scale=7.5, **_sphere_geom(geom))
        assert torch.equal(final, T(z[f"sphere_{gname}_final"])) and torch.equal(den, T(z[f"sphere_{gname}_denoised"])), gname


def test_g12_sphere_loop_tiny_unet():
    from oracle import sphere as S
    z = npz("sphere.npz")
    meta = json.load(open(os.path.join(G, "sphere_traces.json")))
    cond, uncond = T(z["cond"]), T(z["uncond"])
    params, sd = _tiny_setup()
    torch.manual_seed(2333333)
    final, den = S.t2v_sphere_sample(lambda x, ts, ctx: unet_forward(sd, params, x, ts, ctx, fps=8), oddim.DiffusionTables(),
                                     cond, uncond, guidance_scale=7.5, **_sphere_geom(meta["geoms"]["base"]))
    for got, key in ((final, "final"), (den, "denoised")):
        ref = T(z[f"sphere_base_tiny_{key}"])
        assert float((got - ref).abs().max()) / float(ref.abs().max()) < 1e-4, key


def test_g12_bilinear_splat_bit_exact():
    from oracle import sphere as S
    from dynamicscaler_amd.synth import synth_normal
    z = npz("sphere.npz")
    pano = T(z["rt_pano"])
    n = 0
    while f"splat_args_{n}" in z:
        fov, th, ph = z[f"splat_args_{n}"].tolist()
        tile = synth_normal((1, 4, 3, 8, 16), 300 + n)
        assert torch.equal(S.sphere_splat_bilinear(pano.clone(), tile, fov, th, ph), T(z[f"splat_after_{n}"])), n
        n += 1
    assert n == 4


def test_g13_i2v_sphere_loop_vs_reference_golden():
    """P5 (i2v): the oracle's sphere loop with frame windows, per-view image tokens, merge-prev and paste_on_static
    against the reference's own final latents (fake eps: bit-exact) and window traces."""
    from oracle.sphere import i2v_sphere_sample
    from oracle.ddim import DiffusionTables
    from helpers import synth_image_embedder
    z = np.load(os.path.join(G, "sphere_i2v.npz"))
    meta = json.load(open(os.path.join(G, "sphere_i2v_traces.json")))
    cond, uncond, pano_img = T(z["cond"]), T(z["uncond"]), T(z["pano_img"])
    embed = synth_image_embedder(64)
    uc = torch.cat([uncond, embed(torch.zeros(1, 3, 8, 16))], dim=1)      # zero image of the LATENT size (:123-129)
    fake = lambda x, ts, ctx: 0.1 * x + 0.01 * ctx.mean()
    for gname, geom in meta["geoms"].items():
        g = dict(geom)
        g["phi_theta_dict"] = {int(k): v for k, v in g["phi_theta_dict"].items()}
        g.pop("dock_at_f", None)
        trace = []
        torch.manual_seed(2333333)
        final, den = i2v_sphere_sample(fake, embed, DiffusionTables(), cond, uc, pano_img, guidance_scale=7.5,
                                       dock_at_f=geom.get("dock_at_f"), static_frame_latent=T(z["static_latent"]),
                                       trace=trace, **g)
        assert torch.equal(final, T(z[f"i2vs_{gname}_fake_final"])), gname
        assert torch.equal(den, T(z[f"i2vs_{gname}_fake_denoised"])), gname
        for (i, t, views), ref in zip(trace, meta["traces"][gname]):
            assert i == ref["i"] and t == ref["t"] and [list(v) for v in views] == ref["views"], (gname, i)


def test_g22_i2v_sphere_loop_view_get_scale_factor_bit_exact():
    """view_get_scale_factor 2 / 3 of the i2v sphere loop (i2v_sphere_panorama_pipeline.py:58,330-341): the latent view is
    gathered at g x the tile size and resized back with 'nearest' (a strided tensor: the re-noise draw takes torch's scalar
    normal path).  Fake eps: bit-exact against the reference's own final latents, incl. frame windows + docking and
    paste_on_static.  Inputs are those of g13 (sphere_i2v.npz)."""
    from oracle.sphere import i2v_sphere_sample
    from oracle.ddim import DiffusionTables
    from helpers import synth_image_embedder
    z = np.load(os.path.join(G, "sphere_i2v.npz"))
    zs = np.load(os.path.join(G, "sphere_i2v_scale.npz"))
    cases = json.load(open(os.path.join(G, "sphere_i2v_scale.json")))["cases"]
    cond, uncond, pano_img = T(z["cond"]), T(z["uncond"]), T(z["pano_img"])
    embed = synth_image_embedder(64)
    uc = torch.cat([uncond, embed(torch.zeros(1, 3, 8, 16))], dim=1)
    fake = lambda x, ts, ctx: 0.1 * x + 0.01 * ctx.mean()
    assert len(cases) == 3
    for name, geom in cases.items():
        g = dict(geom)
        g["phi_theta_dict"] = {int(k): v for k, v in g["phi_theta_dict"].items()}
        dock = g.pop("dock_at_f", None)
        torch.manual_seed(2333333)
        final, den = i2v_sphere_sample(fake, embed, DiffusionTables(), cond, uc, pano_img, guidance_scale=7.5, dock_at_f=dock,
                                       static_frame_latent=T(z["static_latent"]), **g)
        assert torch.equal(final, T(zs[f"{name}_final"])), name
        assert torch.equal(den, T(zs[f"{name}_denoised"])), name


def test_g33_sphere_loops_view_set_scale_factor_bit_exact():
    """view_set_scale_factor 2 / 3 (x_prev, pred_x0 and the mask's ones resized up with 'nearest' before the scatter: several
    neighbouring sources per target, the last one in row-major order wins) and downsample_factor_before_vae_decode of both sphere
    loops, also combined with a get scale factor, a per-phi fov, frame windows + docking and paste_on_static -- bit-exact against
    the reference run with one torch thread (with more, its own scatter is not repeatable).  merge-prev with a set scale factor
    mixes tensors of two sizes: the reference raises, and so does the oracle."""
    from oracle import sphere as S
    from helpers import synth_image_embedder
    zs = npz("sphere_set_scale.npz")
    meta = json.load(open(os.path.join(G, "sphere_set_scale.json")))
    cond, uncond = T(zs["cond"]), T(zs["uncond"])
    assert len(meta["geoms"]) == 2 and len(meta["i2v_cases"]) == 2
    for gname, geom in meta["geoms"].items():
        torch.manual_seed(2333333)
        final, den = S.t2v_sphere_sample(_fake_eps, oddim.DiffusionTables(), cond, uncond, guidance_scale=7.5, **_sphere_geom(geom))
        assert torch.equal(final, T(zs[f"sphere_{gname}_final"])) and torch.equal(den, T(zs[f"sphere_{gname}_denoised"])), gname
    z = np.load(os.path.join(G, "sphere_i2v.npz"))
    pano_img = T(z["pano_img"])
    embed = synth_image_embedder(64)
    uc = torch.cat([uncond, embed(torch.zeros(1, 3, 8, 16))], dim=1)
    fake = lambda x, ts, ctx: 0.1 * x + 0.01 * ctx.mean()
    for name, geom in meta["i2v_cases"].items():
        g = dict(geom)
        g["phi_theta_dict"] = {int(k): v for k, v in g["phi_theta_dict"].items()}
        dock = g.pop("dock_at_f", None)
        torch.manual_seed(2333333)
        final, den = S.i2v_sphere_sample(fake, embed, oddim.DiffusionTables(), cond, uc, pano_img, guidance_scale=7.5, dock_at_f=dock,
                                         static_frame_latent=T(z["static_latent"]), **g)
        assert torch.equal(final, T(zs[f"i2v_{name}_final"])), name
        assert torch.equal(den, T(zs[f"i2v_{name}_denoised"])), name
    assert meta["merge_prev_with_set_scale_raises"] == "RuntimeError"
    base = dict(json.load(open(os.path.join(G, "sphere_i2v_traces.json")))["geoms"]["base"], view_set_scale_factor=2)
    base["phi_theta_dict"] = {int(k): v for k, v in base["phi_theta_dict"].items()}
    with pytest.raises(RuntimeError):
        S.i2v_sphere_sample(fake, embed, oddim.DiffusionTables(), cond, uc, pano_img, guidance_scale=7.5, **base)


def test_g13_i2v_sphere_loop_tiny_unet():
    from oracle.sphere import i2v_sphere_sample
    from helpers import synth_image_embedder
    z = np.load(os.path.join(G, "sphere_i2v.npz"))
    meta = json.load(open(os.path.join(G, "sphere_i2v_traces.json")))
    cond, uncond = T(z["cond"]), T(z["uncond"])
    params, _ = _tiny_setup()
    p2 = dict(params)
    p2["use_image_attention"] = True
    sd2 = synth_state_dict(param_shapes(p2), seed=5)
    embed = synth_image_embedder(64)
    uc = torch.cat([uncond, embed(torch.zeros(1, 3, 8, 16))], dim=1)
    g = dict(meta["geoms"]["base"])
    g["phi_theta_dict"] = {int(k): v for k, v in g["phi_theta_dict"].items()}
    torch.manual_seed(2333333)
    final, den = i2v_sphere_sample(lambda x, ts, ctx: unet_forward(sd2, p2, x, ts, ctx, fps=8), embed, oddim.DiffusionTables(),
                                   cond, uc, T(z["pano_img"]), guidance_scale=7.5, **g)
    for got, key in ((final, "i2vs_base_tiny_final"), (den, "i2vs_base_tiny_denoised")):
        ref = T(z[key])
        assert float((got - ref).abs().max()) / float(ref.abs().max()) < 1e-4, key


def test_g14_vae_decode_vs_reference_golden():
    """N2 decode side: the oracle's AutoencoderKL.decode / decode_first_stage_2DAE against the reference's modules
    (toy config exactly up to ATen summation order; the real first-stage config on one 40x64 latent frame, stored fp16)."""
    from oracle.vae import vae_decode, decode_first_stage_2dae
    from dynamicscaler_amd.vae_spec import decoder_param_shapes
    z = npz("vae_tiny.npz")
    dd = json.loads(bytes(z["tiny_dd_json"]).decode())
    sd = synth_state_dict(decoder_param_shapes(dd, 4), seed=21)
    zz = T(z["tiny_z"])
    out = vae_decode(sd, dd, zz[:, :, 0])
    ref = T(z["tiny_frame"])
    assert out.shape == ref.shape and float((out - ref).abs().max()) / float(ref.abs().max()) < 2e-5
    vid = decode_first_stage_2dae(sd, dd, zz, scale_factor=0.18215)
    ref = T(z["tiny_video"])
    assert vid.shape == ref.shape and float((vid - ref).abs().max()) / float(ref.abs().max()) < 2e-5
    zf = npz("vae_full.npz")
    ddf = json.loads(bytes(zf["full_dd_json"]).decode())
    sdf = synth_state_dict(decoder_param_shapes(ddf, 4), seed=22)
    out = vae_decode(sdf, ddf, T(zf["full_z"]))
    ref = T(zf["full_frame"]).float()
    assert out.shape == ref.shape == (1, 3, 320, 512)
    assert float((out - ref).abs().max()) / float(ref.abs().max()) < 2e-3      # the fixture is rounded to fp16


def test_g15_vae_encode_vs_reference_golden():
    """N2 encode side: moments of AutoencoderKL.encode, the seeded posterior sample of encode_first_stage_2DAE and the
    pipeline's tiled encode against the reference's own outputs (toy 8x config; the real config on a 320x512 image)."""
    from oracle.vae import vae_encode_moments, encode_first_stage_2dae, tiled_vae_encode
    from dynamicscaler_amd.vae_spec import vae_param_shapes
    z = npz("vae_enc_tiny.npz")
    dd = json.loads(bytes(z["tiny8_dd_json"]).decode())
    sd = synth_state_dict(vae_param_shapes(dd, 4), seed=23)
    img = T(z["tiny8_img"])
    mom = vae_encode_moments(sd, dd, img[:, :, 0])
    ref = T(z["tiny8_moments"])
    assert mom.shape == ref.shape and float((mom - ref).abs().max()) / float(ref.abs().max()) < 2e-5
    torch.manual_seed(77)
    enc = encode_first_stage_2dae(sd, dd, img, scale_factor=0.18215)
    ref = T(z["tiny8_encoded"])
    assert enc.shape == ref.shape and float((enc - ref).abs().max()) / float(ref.abs().max()) < 2e-5
    torch.manual_seed(78)
    til = tiled_vae_encode(sd, dd, T(z["tiny8_big"]), scale_factor=0.18215, overlap_h=2, overlap_w=2)
    ref = T(z["tiny8_tiled"])
    assert til.shape == ref.shape == (1, 4, 1, 16, 32) and float((til - ref).abs().max()) / float(ref.abs().max()) < 2e-5
    zf = npz("vae_enc_full.npz")
    ddf = json.loads(bytes(zf["full_dd_json"]).decode())
    sdf = synth_state_dict(vae_param_shapes(ddf, 4), seed=24)
    mom = vae_encode_moments(sdf, ddf, T(zf["full_img"]).float())
    ref = T(zf["full_moments"])
    assert mom.shape == ref.shape == (1, 8, 40, 64) and float((mom - ref).abs().max()) / float(ref.abs().max()) < 2e-5


def test_g11_i2v_grid_loop_vs_reference_golden():
    """P4 (i2v): VC2_Pipeline_I2V.basic_sample_shift_multi_windows (i2v_normal_pipeline.py:68-425) -- non-overlapping
    shifted grid, docking windows first in the h list, per-window image crops, use_skip_time with a given init latent."""
    from helpers import synth_image_embedder
    z = npz("loops_grid_i2v.npz")
    meta = json.load(open(os.path.join(G, "loops_grid_i2v_traces.json")))
    cond, uncond, img = T(z["cond"]), T(z["uncond"]), T(z["grid_img"])
    embed = synth_image_embedder(64)
    uc = torch.cat([uncond, embed(torch.zeros(1, 3, 8, 16))], dim=1)
    for gname, geom in meta["i2v_grid_geoms"].items():
        g = dict(geom)
        if "init_seed" in g:
            g["init_panorama_latent"] = synth_normal((1, 4, g["frames"] * g["num_windows_f"], g["height"] * g["num_windows_h"] // 8,
                                                      g["width"] * g["num_windows_w"] // 8), g.pop("init_seed"))
        trace = []
        torch.manual_seed(2333333)
        den, _ = oloops.i2v_grid_sample(_fake_eps, embed, oddim.DiffusionTables(), cond, uc, img, guidance_scale=7.5,
                                        trace=trace, **g)
        assert torch.equal(den, T(z[f"i2vgrid_{gname}_fake"])), gname
        for (i, t, wins), ref in zip(trace, meta["traces"][f"i2vgrid_{gname}"]):
            assert i == ref["i"] and t == ref["t"] and [list(w) for w in wins] == ref["windows"], (gname, i)
    params, _ = _tiny_setup()
    p2 = dict(params)
    p2["use_image_attention"] = True
    sd2 = synth_state_dict(param_shapes(p2), seed=5)
    torch.manual_seed(2333333)
    den, _ = oloops.i2v_grid_sample(lambda x, ts, ctx: unet_forward(sd2, p2, x, ts, ctx, fps=8), embed, oddim.DiffusionTables(),
                                    cond, uc, img, guidance_scale=7.5, **meta["i2v_grid_geoms"]["plain"])
    ref = T(z["i2vgrid_plain_tiny"])
    assert float((den - ref).abs().max()) / float(ref.abs().max()) < 1e-4


def _enc_cfg(z):
    return json.loads(bytes(z["toy_cfg_json"]).decode())


def test_g16_resampler_vs_reference_golden():
    """N3: oracle Resampler == the reference's ip_resampler.Resampler on the same synthetic weights (toy, and the i2v
    configuration of ddpm3d.py:683-685 on 257 image tokens)."""
    from oracle.encoders import resampler_forward
    from dynamicscaler_amd.encoder_spec import RESAMPLER_I2V, resampler_param_shapes
    from dynamicscaler_amd.synth import synth_encoder_state_dict
    z = npz("encoders_toy.npz")
    r = _enc_cfg(z)["resampler"]
    out = resampler_forward(synth_encoder_state_dict(resampler_param_shapes(**r), 51), T(z["toy_resampler_x"]),
                            depth=r["depth"], heads=r["heads"])
    assert torch.equal(out, T(z["toy_resampler_out"]))
    zf = npz("encoders_full.npz")
    r = RESAMPLER_I2V
    out = resampler_forward(synth_encoder_state_dict(resampler_param_shapes(**r), 61), T(zf["full_resampler_x"]).float(),
                            depth=r["depth"], heads=r["heads"])
    ref = T(zf["full_resampler_out"]).float()
    assert out.shape == (1, 16, 1024) and float((out - ref).abs().max()) < 2e-5


def test_g16_clip_towers_vs_independent_implementation():
    """N3, parity UNPINNED against open_clip (absent): the oracle's towers against transformers' CLIP carrying the same
    weights -- toy sizes here, the ViT-H/14 sizes in the -m gpu suite's fixtures."""
    from oracle.encoders import clip_text_encode, clip_vision_tokens
    from dynamicscaler_amd.encoder_spec import clip_text_param_shapes, clip_vision_param_shapes
    from dynamicscaler_amd.synth import synth_encoder_state_dict
    z = npz("encoders_toy.npz")
    c = _enc_cfg(z)["clip"]
    t, v = c["text"], c["vision"]
    out = clip_text_encode(synth_encoder_state_dict(clip_text_param_shapes(t), 53), T(z["toy_text_tokens"]),
                           heads=t["heads"], layers=t["layers"], layer_idx=1)
    assert float((out - T(z["toy_text_out"])).abs().max()) < 2e-5
    out = clip_vision_tokens(synth_encoder_state_dict(clip_vision_param_shapes(v), 55), T(z["toy_vision_pixels"]),
                             heads=v["width"] // v["head_width"], layers=v["layers"])
    assert out.shape == (2, 17, v["width"]) and float((out - T(z["toy_vision_out"])).abs().max()) < 2e-5


def test_g16_clip_preprocess_properties():
    """kornia's resize restated (unpinned): without shrinking it is F.interpolate(bicubic, align_corners=True); a
    constant image stays constant through blur + resize; output is normalised with the CLIP mean / std."""
    import torch.nn.functional as F
    from oracle.encoders import clip_preprocess
    from dynamicscaler_amd.encoder_spec import CLIP_MEAN, CLIP_STD
    x = synth_normal((1, 3, 20, 24), 9).clamp(-1, 1)
    up = clip_preprocess(x, 32)
    ref = (F.interpolate(x, size=(32, 32), mode="bicubic", align_corners=True) + 1) / 2
    ref = (ref - torch.tensor(CLIP_MEAN).view(1, 3, 1, 1)) / torch.tensor(CLIP_STD).view(1, 3, 1, 1)
    assert torch.allclose(up, ref, atol=1e-6)
    const = clip_preprocess(torch.full((1, 3, 96, 160), 0.25), 32)
    want = ((0.25 + 1) / 2 - torch.tensor(CLIP_MEAN)) / torch.tensor(CLIP_STD)
    assert torch.allclose(const, want.view(1, 3, 1, 1).expand_as(const), atol=1e-5)


def test_g11_grid_pre_denoise_and_residual_merge_vs_reference_golden():
    """R11's remaining branches (t2v_normal_pipeline.py:345-412, 445-468): pre-denoise start (tile denoised for a few
    steps or given, bicubic resize, _add_noise, non-progressive and progressive skip) and the per-step sparse / dense
    residual merge -- oracle == the reference's panoramas, bit for bit with the fake eps-model, and with the toy UNet."""
    z = npz("loops_grid_i2v.npz")
    meta = json.load(open(os.path.join(G, "loops_grid_i2v_traces.json")))
    cond, uncond = T(z["cond"]), T(z["uncond"])
    for gname, geom in meta["grid_pre_geoms"].items():
        gk = dict(geom)
        if "clear_seed" in gk:
            gk["clear_pre_denoised_latent"] = synth_normal((1, 4, 4, 8, 16), gk.pop("clear_seed"))
        torch.manual_seed(2333333)
        den, _ = oloops.t2v_grid_sample(_fake_eps, oddim.DiffusionTables(), cond, uncond, height=64, width=128, frames=4,
                                        guidance_scale=7.5, **gk)
        assert torch.equal(den, T(z[f"gridpre_{gname}_fake"])), (gname, float((den - T(z[f"gridpre_{gname}_fake"])).abs().max()))
    params = json.loads(bytes(npz("unet_tiny_t2v.npz")["params_json"]).decode())
    sd = synth_state_dict(param_shapes(params), 5)
    eps = lambda x, t, c: unet_forward(sd, params, x, t, c, fps=8)
    torch.manual_seed(2333333)
    den, _ = oloops.t2v_grid_sample(eps, oddim.DiffusionTables(), cond, uncond, height=64, width=128, frames=4,
                                    guidance_scale=7.5, **meta["grid_pre_geoms"]["pre_sparse"])
    ref = T(z["gridpre_pre_sparse_tiny"])
    assert float((den - ref).abs().max()) / float(ref.abs().max()) < 1e-4


def test_g23_50_step_schedule_golden_is_the_oracles_update():
    """cfg1_50step_t2v.npz (make_golden.py g23: the reference's basic_sample, 50 steps, real t2v UNet): the schedule is the
    oracle's, every teacher-forced update stored by the reference is reproduced bit for bit by oracle.ddim.ddim_step from the
    stored (x_t, e_t) (x_prev by value, pred_x0 by SHA-256), and the free-running record is self-consistent."""
    import hashlib
    path = os.path.join(G, "cfg1_50step_t2v.npz")
    if not os.path.exists(path):
        pytest.skip("cfg1_50step_t2v.npz not generated yet (make_golden.py --full --only g23)")
    z = np.load(path)
    sched = oddim.DDIMSchedule(oddim.DiffusionTables(), 50)
    assert list(np.flip(sched.ddim_timesteps)) == list(z["timesteps"])
    for idx in [int(i) for i in z["tf_indices"]]:
        x, e = T(z[f"tf_x_t_{idx}"]).float(), T(z[f"tf_e_t_{idx}"])
        assert int(z[f"tf_t_{idx}"]) == int(np.flip(sched.ddim_timesteps)[49 - idx])
        xp, x0 = oddim.ddim_step(sched, x, e, [idx] * 16, noise=torch.zeros_like(e))
        assert torch.equal(xp, T(z[f"tf_x_prev_{idx}"]))
        assert hashlib.sha256(np.ascontiguousarray(x0.numpy()).tobytes()).hexdigest() == str(z[f"tf_pred_x0_sha_{idx}"])
    st = z["stats"]                                  # (index, |x_t|, |e_t|, |x_prev|, |pred_x0|) per step
    assert st.shape == (50, 5) and list(st[:, 0]) == list(range(49, -1, -1))
    assert np.allclose(st[1:, 1], st[:-1, 3])         # x_t of a step is x_prev of the one before
    assert abs(float(T(z["free_x_prev_0"]).std()) - st[-1, 3]) < 1e-4 * st[-1, 3]


def test_g24_panorama_handlers_oracle_vs_reference_golden():
    """oracle/handlers.py (PanoramaTensor, RingLatentProxy, RingPanoramaTensor, RingPanoramaLatentProxy) against vectors recorded
    from the reference's own classes (make_golden.py g24): every get / set / splat bit for bit."""
    from oracle import handlers as oh
    z = npz("panorama_handlers.npz")
    views = [tuple(float(a) for a in v) for v in z["views"]]
    for tag in ("p4", "p3", "p2"):
        o = oh.PanoramaTensor(T(z[f"{tag}_x"]))
        for vi, (fov, th, ph) in enumerate(views):
            v, m = o.get_view_tensor_no_interpolate(fov, th, ph, 12, 10)
            assert torch.equal(v, T(z[f"{tag}_get{vi}"])) and torch.equal(m, T(z[f"{tag}_mask{vi}"]))
        for vi, (fov, th, ph) in enumerate(views):
            o.set_view_tensor_no_interpolation(T(z[f"{tag}_src{vi}"]), fov, th, ph)
            assert torch.equal(o.equirect_tensor, T(z[f"{tag}_after_set{vi}"])), (tag, vi)
        o.set_view_tensor_bilinear(T(z[f"{tag}_splat_src"]), 90.0, 45.0, -30.0)
        assert torch.equal(o.equirect_tensor, T(z[f"{tag}_after_splat"])), tag
    r = oh.RingLatentProxy(T(z["rl_x"]))
    assert torch.equal(r.get_window_latent(3, 8), T(z["rl_win_3_8"])) and torch.equal(r.get_window_latent(1, 10), T(z["rl_win_1_10"]))
    assert torch.equal(r.get_window_latent(None, None), T(z["rl_win_none"]))
    assert tuple(r.get_operating_shape(3, 8)) == tuple(int(a) for a in z["rl_shape_3_8"])
    r.set_window_latent(T(z["rl_src"]), 4, 7)
    assert torch.equal(r.get_torch_latent(), T(z["rl_after_set"]))
    windows = ((3, 7), (None, None), (4, 9))
    for tag, cls in (("rp", oh.RingPanoramaTensor), ("rpl", oh.RingPanoramaLatentProxy)):
        o = cls(T(z[f"{tag}_x"]))
        for vi, ((fov, th, ph), (fb, fe)) in enumerate(zip(views, windows)):
            v, m = o.get_view_tensor_no_interpolate(fov, th, ph, 12, 10, frame_begin=fb, frame_end=fe)
            assert torch.equal(v, T(z[f"{tag}_get{vi}"])) and torch.equal(m, T(z[f"{tag}_mask{vi}"])), (tag, vi)
        for vi, ((fov, th, ph), (fb, fe)) in enumerate(zip(views, windows)):
            o.set_view_tensor_no_interpolation(T(z[f"{tag}_src{vi}"]), fov, th, ph, frame_begin=fb, frame_end=fe)
            full = o.get_equirect_tensor() if tag == "rpl" else o.equirect_tensor_handler.get_torch_latent()
            assert torch.equal(full, T(z[f"{tag}_after_set{vi}"])), (tag, vi)


def test_g30_uncalled_handler_methods_oracle_vs_reference_golden():
    """The handler methods no pipeline of the reference calls -- get_view_tensor_interpolate (F.grid_sample), set_view_tensor
    (round-to-nearest scatter_ with the reference's [B, -1] reshape of the target map) and the ring-backed set_view_tensor_bilinear
    -- restated in oracle/handlers.py, against the reference's own classes (make_golden.py g30): scatters and splats bit for bit,
    the interpolation to fp32 rounding; where the reference raises (a one-frame ring window), the oracle does."""
    import json
    from oracle import handlers as oh
    z = npz("panorama_handlers_uncalled.npz")
    raised = json.loads(bytes(z["raised_json"]).decode())
    assert raised == {"rp_set3": "RuntimeError", "rpl_set3": "RuntimeError"}
    views = [tuple(float(a) for a in v) for v in z["views"]]
    modes = [("bilinear", True), ("bilinear", False), ("nearest", True)]
    for tag in ("p4", "p3", "p2", "p5"):
        o = oh.PanoramaTensor(T(z[f"{tag}_x"]))
        for vi, (fov, th, ph) in enumerate(views):
            for mi, (mode, ac) in enumerate(modes):
                v = o.get_view_tensor_interpolate(fov, th, ph, 12, 10, mode, ac)
                g = T(z[f"{tag}_interp{vi}_{mi}"])
                assert tuple(v.shape) == tuple(g.shape) and torch.allclose(v, g, rtol=1e-6, atol=1e-6), (tag, vi, mi)
        for vi, (fov, th, ph) in enumerate(views):
            o.set_view_tensor(T(z[f"{tag}_src{vi}"]), fov, th, ph)
            g = T(z[f"{tag}_after_set{vi}"])
            assert tuple(o.equirect_tensor.shape) == tuple(g.shape) and torch.equal(o.equirect_tensor, g), (tag, vi)
    o = oh.PanoramaLatentProxy(T(z["pl_x"]))
    for vi, (fov, th, ph) in enumerate(views):
        assert torch.allclose(o.get_view_tensor_interpolate(fov, th, ph, 12, 10), T(z[f"pl_interp{vi}"]), rtol=1e-6, atol=1e-6)
    for vi, (fov, th, ph) in enumerate(views):
        o.set_view_tensor(T(z[f"pl_src{vi}"]), fov, th, ph)
        assert torch.equal(o.get_equirect_tensor(), T(z[f"pl_after_set{vi}"])), vi
    wins = ((3, 7), (None, None), (4, 9), (2, 3))
    for tag, cls in (("rp", oh.RingPanoramaTensor), ("rpl", oh.RingPanoramaLatentProxy)):
        o = cls(T(z[f"{tag}_x"]))
        full = (lambda: o.get_equirect_tensor()) if tag == "rpl" else (lambda: o.equirect_tensor_handler.get_torch_latent())
        for vi, ((fov, th, ph), (fb, fe)) in enumerate(zip(views + views[:1], wins)):
            for mi, (mode, ac) in enumerate(modes[:2] if vi else modes):
                v = o.get_view_tensor_interpolate(fov, th, ph, 12, 10, frame_begin=fb, frame_end=fe, interpolate_mode=mode, interpolate_align_corners=ac)
                g = T(z[f"{tag}_interp{vi}_{mi}"])
                assert tuple(v.shape) == tuple(g.shape) and torch.allclose(v, g, rtol=1e-6, atol=1e-6), (tag, vi, mi)
        for vi, ((fov, th, ph), (fb, fe)) in enumerate(zip(views + views[:1], wins)):
            if f"{tag}_set{vi}" in raised:
                with pytest.raises(RuntimeError):
                    o.set_view_tensor(T(z[f"{tag}_src{vi}"]), fov, th, ph, frame_begin=fb, frame_end=fe)
                continue
            o.set_view_tensor(T(z[f"{tag}_src{vi}"]), fov, th, ph, frame_begin=fb, frame_end=fe)
            assert torch.equal(full(), T(z[f"{tag}_after_set{vi}"])), (tag, vi)
        for vi, ((fov, th, ph), (fb, fe)) in enumerate(zip(views + views[:1], wins)):
            o.set_view_tensor_bilinear(T(z[f"{tag}_splat_src{vi}"]), fov, th, ph, frame_begin=fb, frame_end=fe)
            assert torch.equal(full(), T(z[f"{tag}_after_splat{vi}"])), (tag, vi)


def test_philox_restatement_known_answer_and_moments():
    """oracle/philox.py -- the CPU restatement of ds_renoise_mix's in-kernel noise (rng_mode="device") -- reproduces the Random123
    known-answer vector of Philox4x32-10 (counter 0, key 0) and produces unit normals; distinct counters / seeds decorrelate."""
    from oracle import philox
    r = philox.philox4x32_10(np.array([0], dtype=np.uint64), 0)[0]
    assert [int(v) for v in r] == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    z = philox.tile_noise((8, 4, 16, 40, 64), 7, 0)
    assert z.dtype == np.float32 and abs(float(z.mean())) < 5e-3 and abs(float(z.std()) - 1.0) < 5e-3
    assert abs(float((z.astype(np.float64) ** 4).mean()) - 3.0) < 0.05
    z2 = philox.tile_noise((8, 4, 16, 40, 64), 8, 0)
    assert abs(float(np.corrcoef(z.ravel(), z2.ravel())[0, 1])) < 5e-3
    assert np.array_equal(philox.tile_noise((1, 4, 4, 8, 16), 7, 12)[0].ravel()[:16], philox.normal4(np.uint64(12) + np.arange(4, dtype=np.uint64), 7).ravel())
