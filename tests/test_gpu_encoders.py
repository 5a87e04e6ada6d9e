"""-m gpu: the conditioning producers on the HIP kernels -- Resampler against the reference's own
module (golden), the OpenCLIP towers against the oracle and the independent-implementation fixtures (open_clip is
absent: parity unpinned, see oracle/encoders.py), kernels against plain torch fp32."""
import json
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")
TOL = 3.2e-3        # fp16 activations / fp32 accumulation vs fp32 CPU, rel-L2 over the output tokens: 2x the largest
                    # measured value (image tower 1.6e-3; text tower 1.2e-3, Resampler 8.5e-4, crop chain 7.7e-4)


def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    return torch.device("cuda:0")


def T(a):
    return torch.from_numpy(np.asarray(a))


def relerr(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-12))


@pytest.mark.parametrize("hd,heads,nq,nk,causal,batch", [(64, 2, 77, 77, True, 2), (80, 4, 257, 257, False, 1),
                                                         (64, 12, 16, 273, False, 1), (80, 2, 17, 17, False, 3),
                                                         (64, 3, 130, 130, True, 1)])
def test_attention_enc_vs_torch(hd, heads, nq, nk, causal, batch):
    from dynamicscaler_amd import ops
    from dynamicscaler_amd.synth import synth_normal
    d = dev()
    W = heads * hd
    q = synth_normal((batch * nq, W), 1).to(d, torch.float16)
    kv = synth_normal((batch * nk, 2 * W), 2).to(d, torch.float16)
    out = torch.empty((batch * nq, W), dtype=torch.float16, device=d)
    ops.attention_enc(q, kv, kv[:, W:], out, batch=batch, heads=heads, nq=nq, nk=nk, ldq=W, ldk=2 * W, ldv=2 * W, ldo=W,
                      head_dim=hd, scale=hd ** -0.5, causal=causal)
    qf = q.float().cpu().view(batch, nq, heads, hd).transpose(1, 2)
    kf = kv[:, :W].float().cpu().view(batch, nk, heads, hd).transpose(1, 2)
    vf = kv[:, W:].float().cpu().view(batch, nk, heads, hd).transpose(1, 2)
    w = qf @ kf.transpose(-1, -2) * hd ** -0.5
    if causal:
        w = w + torch.full((nq, nk), float("-inf")).triu_(1)
    ref = (torch.softmax(w, -1) @ vf).transpose(1, 2).reshape(batch * nq, W)
    e = relerr(out, ref)
    print(f"attention_enc hd{hd} h{heads} {nq}x{nk} causal={causal}: rel err {e:.3e}")
    assert e < 2e-3 and torch.isfinite(out).all()


def test_encoder_elementwise_kernels_vs_torch():
    from dynamicscaler_amd import ops
    from dynamicscaler_amd.synth import synth_normal
    d = dev()
    x = synth_normal((77, 512), 3, scale=2.0)
    y = ops.gelu_(x.to(d, torch.float16).clone())
    assert torch.allclose(y.cpu().float(), torch.nn.functional.gelu(x), rtol=2e-3, atol=2e-4)
    table = synth_normal((100, 64), 4)
    pos = synth_normal((7, 64), 5)
    tok = torch.randint(0, 100, (3, 7), generator=torch.Generator().manual_seed(1), dtype=torch.int32)
    e = ops.embed_tokens(tok.to(d), table.to(d, torch.float16), pos.to(d))
    assert torch.equal(e.cpu().view(3, 7, 64), (table[tok.long()] + pos).half())
    patches = synth_normal((2 * 4, 64), 6)
    cls = synth_normal((64,), 7)
    pos2 = synth_normal((5, 64), 8)
    a = ops.vit_assemble(patches.to(d, torch.float16), cls.to(d), pos2.to(d), 2)
    ref = torch.cat([cls.expand(2, 1, 64), patches.view(2, 4, 64)], 1) + pos2
    assert torch.equal(a.cpu().view(2, 5, 64), ref.half())
    img = synth_normal((2, 3, 28, 28), 9)
    rows = ops.patchify(img.to(d), 14, 640)
    ref = img.view(2, 3, 2, 14, 2, 14).permute(0, 2, 4, 1, 3, 5).reshape(8, 588)
    assert torch.equal(rows[:, :588].cpu(), ref.half()) and not rows[:, 588:].any()


@pytest.mark.parametrize("shape,size", [((1, 3, 320, 512), 224), ((2, 3, 40, 64), 56), ((1, 3, 224, 224), 224),
                                        ((1, 3, 512, 1024), 224), ((1, 3, 300, 200), 224)])
def test_clip_preprocess_vs_oracle(shape, size):
    """Blur (reflect border) + bicubic(align_corners) + normalise in one kernel vs the oracle's torch restatement."""
    from oracle.encoders import clip_preprocess
    from dynamicscaler_amd import ops
    from dynamicscaler_amd.encoder_spec import CLIP_MEAN, CLIP_STD
    from dynamicscaler_amd.synth import synth_normal
    d = dev()
    img = synth_normal(shape, 11, scale=0.5).clamp(-1, 1)
    got = ops.clip_preprocess(img.to(d), size, CLIP_MEAN, CLIP_STD)
    ref = clip_preprocess(img, size)
    err = float((got.cpu() - ref).abs().max())
    print(f"clip_preprocess {shape} -> {size}: max abs err {err:.2e}")
    assert got.shape == ref.shape and err < 2e-5
    got16 = ops.clip_preprocess(img.to(d, torch.float16), size, CLIP_MEAN, CLIP_STD)
    assert float((got16.cpu() - clip_preprocess(img.half().float(), size)).abs().max()) < 2e-5


def _toy():
    z = np.load(os.path.join(G, "encoders_toy.npz"))
    return z, json.loads(bytes(z["toy_cfg_json"]).decode())


def test_resampler_vs_reference_golden():
    """Resampler (HIP) against the reference's ip_resampler.Resampler: toy and the i2v configuration on 257 tokens."""
    from dynamicscaler_amd.encoders import Resampler
    from dynamicscaler_amd.encoder_spec import RESAMPLER_I2V, resampler_param_shapes
    from dynamicscaler_amd.synth import synth_encoder_state_dict
    d = dev()
    z, cfg = _toy()
    r = cfg["resampler"]
    m = Resampler(**r)
    m.load_state_dict(synth_encoder_state_dict(resampler_param_shapes(**r), 51))
    out = m(T(z["toy_resampler_x"]).to(d))
    e1 = relerr(out, T(z["toy_resampler_out"]))
    zf = np.load(os.path.join(G, "encoders_full.npz"))
    mf = Resampler(**RESAMPLER_I2V)
    mf.load_state_dict(synth_encoder_state_dict(resampler_param_shapes(**RESAMPLER_I2V), 61))
    outf = mf(T(zf["full_resampler_x"]).to(d))
    e2 = relerr(outf, T(zf["full_resampler_out"]))
    print(f"resampler: toy rel err {e1:.3e}, i2v config rel err {e2:.3e}")
    assert out.shape == (2, 4, 128) and outf.shape == (1, 16, 1024) and e1 < TOL and e2 < TOL


def test_clip_text_tower():
    """Text tower (causal mask, penultimate layer + ln_final): toy and ViT-H/14 sizes, against the fixtures produced by
    the independent implementation and (toy) the oracle on the same host."""
    from oracle.encoders import clip_text_encode
    from dynamicscaler_amd.encoders import FrozenOpenCLIPEmbedder
    from dynamicscaler_amd.encoder_spec import CLIP_VIT_H_14, clip_text_param_shapes
    from dynamicscaler_amd.synth import synth_encoder_state_dict
    d = dev()
    z, cfg = _toy()
    t = cfg["clip"]["text"]
    sd = synth_encoder_state_dict(clip_text_param_shapes(t), 53)
    m = FrozenOpenCLIPEmbedder(layer="penultimate", model_cfg=t)
    m.load_state_dict(sd)
    tok = T(z["toy_text_tokens"])
    out = m.encode_with_transformer(tok.to(d))
    e1 = relerr(out, T(z["toy_text_out"]))
    e1o = relerr(out, clip_text_encode(sd, tok, heads=t["heads"], layers=t["layers"], layer_idx=1))
    last = FrozenOpenCLIPEmbedder(layer="last", model_cfg=t)
    last.load_state_dict(sd)
    e_last = relerr(last(tok.to(d)), clip_text_encode(sd, tok, heads=t["heads"], layers=t["layers"], layer_idx=0))
    with pytest.raises(RuntimeError):
        m(["a prompt without a tokenizer"])
    zf = np.load(os.path.join(G, "encoders_full.npz"))
    tf = CLIP_VIT_H_14["text"]
    mf = FrozenOpenCLIPEmbedder(layer="penultimate")
    mf.load_state_dict(synth_encoder_state_dict(clip_text_param_shapes(tf), 63))
    outf = mf.encode(T(zf["full_text_tokens"]).to(d))
    e2 = relerr(outf, T(zf["full_text_out"]))
    print(f"clip text: toy {e1:.3e} (oracle {e1o:.3e}, layer=last {e_last:.3e}), ViT-H/14 text {e2:.3e}")
    assert out.shape == (2, 77, t["width"]) and outf.shape == (1, 77, 1024)
    assert e1 < TOL and e1o < TOL and e_last < TOL and e2 < TOL


def test_clip_image_tower():
    """Image tower (head width 80, 257 tokens, no ln_post): toy and ViT-H/14 sizes on preprocessed pixels."""
    from dynamicscaler_amd.encoders import FrozenOpenCLIPImageEmbedderV2
    from dynamicscaler_amd.encoder_spec import CLIP_VIT_H_14, clip_vision_param_shapes
    from dynamicscaler_amd.synth import synth_encoder_state_dict
    d = dev()
    z, cfg = _toy()
    v = cfg["clip"]["vision"]
    m = FrozenOpenCLIPImageEmbedderV2(model_cfg=v)
    m.load_state_dict(synth_encoder_state_dict(clip_vision_param_shapes(v), 55))
    pix = T(z["toy_vision_pixels"]).to(d)
    out = m.encode_pixels(pix)                                   # eager (first use)
    e1 = relerr(out, T(z["toy_vision_out"]))
    cap, rep = m.encode_pixels(pix), m.encode_pixels(pix * 0.5)  # hipGraph capture, then a replay on other pixels
    m.use_graph = False
    assert torch.equal(cap, out) and torch.equal(rep, m.encode_pixels(pix * 0.5)) and not torch.equal(rep, out)
    m.use_graph = True
    zf = np.load(os.path.join(G, "encoders_full.npz"))
    vf = CLIP_VIT_H_14["vision"]
    mf = FrozenOpenCLIPImageEmbedderV2()
    mf.load_state_dict(synth_encoder_state_dict(clip_vision_param_shapes(vf), 65))
    outf = mf.encode_pixels(T(zf["full_vision_pixels"]).float().to(d))
    e2 = relerr(outf, T(zf["full_vision_out"]))
    print(f"clip image: toy {e1:.3e}, ViT-H/14 image {e2:.3e}")
    assert out.shape == (2, 17, v["width"]) and outf.shape == (1, 257, 1280) and e1 < TOL and e2 < TOL


def test_get_image_embeds_chain_vs_oracle():
    """LatentVisualDiffusion.get_image_embeds (ddpm3d.py:689-693) end to end on a 320x512 crop: preprocess -> image
    tower -> Resampler, built from the reference's yaml-style configs, against the oracle chain (toy tower sizes keep
    the CPU side in seconds; the Resampler is the i2v one)."""
    from oracle.encoders import get_image_embeds
    from dynamicscaler_amd import dropin
    from dynamicscaler_amd.host_model import LatentDiffusionHost
    from dynamicscaler_amd.encoder_spec import RESAMPLER_I2V, clip_vision_param_shapes, resampler_param_shapes
    from dynamicscaler_amd.synth import synth_encoder_state_dict, synth_normal
    d = dev()
    dropin.install()
    z = np.load(os.path.join(G, "unet_tiny_i2v.npz"))
    params = json.loads(bytes(z["params_json"]).decode())
    vision = dict(image_size=224, layers=2, width=1280, head_width=80, patch_size=14, mlp_ratio=4.0)
    ld = LatentDiffusionHost({"params": params}, finegrained=True, cond_img_config={
        "target": "lvdm.modules.encoders.condition.FrozenOpenCLIPImageEmbedderV2",
        "params": {"freeze": True, "model_cfg": vision}})
    vsd = synth_encoder_state_dict(clip_vision_param_shapes(vision), 71)
    rsd = synth_encoder_state_dict(resampler_param_shapes(**RESAMPLER_I2V), 72)
    ld.embedder.load_state_dict(vsd)
    ld.image_proj_model.load_state_dict(rsd)
    ld = ld.to(d).eval()
    img = synth_normal((1, 3, 320, 512), 73, scale=0.5).clamp(-1, 1)
    emb = ld.get_image_embeds(img)
    ref = get_image_embeds(vsd, rsd, img, vision=vision, resampler=RESAMPLER_I2V)
    e = relerr(emb, ref)
    print(f"get_image_embeds (320x512 crop -> [1,16,1024]): rel err {e:.3e}")
    assert emb.shape == (1, 16, 1024) and e < TOL
    # a batch of crops in one call (what the i2v pipelines do with a step's new windows, RingImageTensor.get_encoded_image_conds)
    # gives, item for item, the bits of the one-crop calls -- at 5 and at 16 crops (the GEMM tile choice changes with the row count)
    from dynamicscaler_amd.pipelines_i2v import RingImageTensor
    pano = synth_normal((3, 320, 2048), 74, scale=0.5).clamp(-1, 1)
    ring = RingImageTensor(image_tensor=pano, height=320, width=2048, device=d)
    for n in (5, 16):
        boxes = [(37 * k * 8 % 2048, 37 * k * 8 % 2048 + 512, 0, 320) for k in range(n)]
        together = ring.get_encoded_image_conds(ld, boxes)
        for b, got in zip(boxes, together):
            one = ring.get_encoded_image_cond(ld, *b)
            assert got.shape == one.shape == (1, 16, 1024) and torch.equal(got, one), (n, b, relerr(got, one))
