"""TEST INFRASTRUCTURE (CPU oracle) -- not part of the product.

Functional restatement (torch CPU fp32) of the reference's conditioning producers:

  Resampler.forward / PerceiverAttention / FeedForward     lvdm/modules/encoders/ip_resampler.py:25-135
      PINNED: tests/golden/encoders.npz holds outputs of the reference's own Resampler (toy and the i2v config of
      ddpm3d.py:683-685) run in the build container (tests/golden/make_golden.py g16).
  FrozenOpenCLIPEmbedder.encode_with_transformer            lvdm/modules/encoders/condition.py:216-234
  FrozenOpenCLIPImageEmbedderV2.encode_with_vision_transformer / preprocess   condition.py:324-365
      PARITY UNPINNED against the reference's dependency: open_clip_torch==2.22.0 (requirements.txt:23) and kornia
      (requirements.txt:24) are absent from this image and the reference has no tests or vectors at this boundary.
      The towers restate open_clip's published modules (CLIP.token_embedding / positional_embedding / attn_mask /
      ln_final; VisionTransformer.conv1 / class_embedding / positional_embedding / ln_pre; Transformer of
      ResidualAttentionBlock: x + attn(ln_1 x), x + c_proj(gelu(c_fc(ln_2 x))), nn.MultiheadAttention with packed
      in_proj) on open_clip's state-dict keys, anchored on the reference's call sites above and cross-checked against
      an INDEPENDENT implementation of the same architecture: transformers' CLIPTextModel / CLIPVisionModel with the
      weights mapped key by key (g16; golden outputs stored).  `clip_preprocess` restates kornia.geometry.resize
      (bicubic, align_corners=True, antialias) from its published source; unpinned.
"""
import math

import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------ Resampler
def _ln(sd, p, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


def _perceiver_attention(sd, p, x, latents, heads, dim_head):
    """ip_resampler.py:62-90."""
    x = _ln(sd, p + ".norm1", x)
    latents = _ln(sd, p + ".norm2", latents)
    b, l, _ = latents.shape
    q = F.linear(latents, sd[p + ".to_q.weight"])
    kv = F.linear(torch.cat((x, latents), dim=-2), sd[p + ".to_kv.weight"])
    k, v = kv.chunk(2, dim=-1)

    def split(t):
        return t.view(b, t.shape[1], heads, -1).transpose(1, 2)

    q, k, v = split(q), split(k), split(v)
    scale = 1 / math.sqrt(math.sqrt(dim_head))
    w = (q * scale) @ (k * scale).transpose(-2, -1)
    w = torch.softmax(w.float(), dim=-1).type(w.dtype)
    out = (w @ v).permute(0, 2, 1, 3).reshape(b, l, -1)
    return F.linear(out, sd[p + ".to_out.weight"])


def resampler_forward(sd, x, *, depth, heads, dim_head=64):
    """Resampler.forward (ip_resampler.py:122-135): x [b, n, embedding_dim] -> [b, num_queries, output_dim]."""
    latents = sd["latents"].repeat(x.size(0), 1, 1)
    x = F.linear(x, sd["proj_in.weight"], sd["proj_in.bias"])
    for i in range(depth):
        latents = _perceiver_attention(sd, f"layers.{i}.0", x, latents, heads, dim_head) + latents
        f = f"layers.{i}.1"
        h = F.linear(_ln(sd, f + ".0", latents), sd[f + ".1.weight"])
        latents = F.linear(F.gelu(h), sd[f + ".3.weight"]) + latents
    latents = F.linear(latents, sd["proj_out.weight"], sd["proj_out.bias"])
    return _ln(sd, "norm_out", latents)


# ------------------------------------------------------------------------------------------------ OpenCLIP towers
def _resblock(sd, p, x, heads, attn_mask=None):
    """open_clip ResidualAttentionBlock (ls_1 / ls_2 are Identity for ViT-H-14); x [b, n, W]."""
    b, n, W = x.shape
    hd = W // heads
    h = _ln(sd, p + ".ln_1", x)
    qkv = F.linear(h, sd[p + ".attn.in_proj_weight"], sd[p + ".attn.in_proj_bias"])
    q, k, v = (t.view(b, n, heads, hd).transpose(1, 2) for t in qkv.chunk(3, dim=-1))
    w = (q * hd ** -0.5) @ k.transpose(-2, -1)
    if attn_mask is not None:
        w = w + attn_mask
    a = (torch.softmax(w, dim=-1) @ v).transpose(1, 2).reshape(b, n, W)
    x = x + F.linear(a, sd[p + ".attn.out_proj.weight"], sd[p + ".attn.out_proj.bias"])
    h = F.linear(_ln(sd, p + ".ln_2", x), sd[p + ".mlp.c_fc.weight"], sd[p + ".mlp.c_fc.bias"])
    return x + F.linear(F.gelu(h), sd[p + ".mlp.c_proj.weight"], sd[p + ".mlp.c_proj.bias"])


def clip_text_encode(sd, tokens, *, heads, layers, layer_idx=1, prefix="model."):
    """encode_with_transformer (condition.py:216-234): tokens int [b, 77] -> [b, 77, W].  layer_idx 1 = 'penultimate'
    (the last resblock is skipped), ln_final applied to what is left (:222)."""
    x = sd[prefix + "token_embedding.weight"][tokens.long()] + sd[prefix + "positional_embedding"]
    n = x.shape[1]
    mask = torch.full((n, n), float("-inf")).triu_(1)                  # open_clip CLIP.build_attention_mask
    for i in range(layers - layer_idx):
        x = _resblock(sd, f"{prefix}transformer.resblocks.{i}", x, heads, mask)
    return _ln(sd, prefix + "ln_final", x)


def clip_vision_tokens(sd, pixels, *, heads, layers, prefix="model.visual."):
    """encode_with_vision_transformer after preprocess (condition.py:341-365): pixels [b,3,S,S] -> [b, 1+g*g, W]
    (all tokens of the last block; no ln_post, no projection)."""
    w = sd[prefix + "conv1.weight"]
    x = F.conv2d(pixels, w, stride=w.shape[-1])
    x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)
    cls = sd[prefix + "class_embedding"] + torch.zeros(x.shape[0], 1, x.shape[-1])
    x = torch.cat([cls, x], dim=1) + sd[prefix + "positional_embedding"]
    x = _ln(sd, prefix + "ln_pre", x)
    for i in range(layers):
        x = _resblock(sd, f"{prefix}transformer.resblocks.{i}", x, heads)
    return x


def _gaussian_blur_reflect(x, ks, sigmas):
    """kornia.filters.gaussian_blur2d(x, ks, sigmas) (border_type='reflect', separable)."""
    def k1d(n, sigma):
        t = torch.arange(n, dtype=torch.float32) - n // 2
        g = torch.exp(-t * t / (2.0 * sigma * sigma))
        return g / g.sum()

    c = x.shape[1]
    kx = k1d(ks[1], sigmas[1]).view(1, 1, 1, -1).repeat(c, 1, 1, 1)
    ky = k1d(ks[0], sigmas[0]).view(1, 1, -1, 1).repeat(c, 1, 1, 1)
    x = F.conv2d(F.pad(x, (ks[1] // 2, ks[1] // 2, 0, 0), mode="reflect"), kx, groups=c)
    return F.conv2d(F.pad(x, (0, 0, ks[0] // 2, ks[0] // 2), mode="reflect"), ky, groups=c)


def clip_preprocess(img, size=224, antialias=True, mean=None, std=None):
    """FrozenOpenCLIPImageEmbedderV2.preprocess (condition.py:324-332) with kornia.geometry.resize restated:
    factors = in/out per axis; if antialias and max(factors) > 1: Gaussian blur with sigma = max((f-1)/2, 0.001),
    kernel = max(int(4 sigma), 3) made odd; then F.interpolate(bicubic, align_corners=True)."""
    from dynamicscaler_amd.encoder_spec import CLIP_MEAN, CLIP_STD
    mean = torch.tensor(CLIP_MEAN if mean is None else mean).view(1, 3, 1, 1)
    std = torch.tensor(CLIP_STD if std is None else std).view(1, 3, 1, 1)
    x = img.float()
    fh, fw = x.shape[-2] / size, x.shape[-1] / size
    if antialias and max(fh, fw) > 1:
        sig = (max((fh - 1.0) / 2.0, 0.001), max((fw - 1.0) / 2.0, 0.001))
        ks = [int(max(2.0 * 2 * sig[0], 3)), int(max(2.0 * 2 * sig[1], 3))]
        ks = [k + 1 if k % 2 == 0 else k for k in ks]
        x = _gaussian_blur_reflect(x, ks, sig)
    x = F.interpolate(x, size=(size, size), mode="bicubic", align_corners=True)
    return ((x + 1.0) / 2.0 - mean) / std


def get_image_embeds(vis_sd, res_sd, batch_imgs, *, vision, resampler, prefix="model.visual."):
    """LatentVisualDiffusion.get_image_embeds (ddpm3d.py:689-693): embedder(batch_imgs) -> image_proj_model."""
    pix = clip_preprocess(batch_imgs, vision["image_size"])
    tok = clip_vision_tokens(vis_sd, pix, heads=vision["width"] // vision["head_width"], layers=vision["layers"], prefix=prefix)
    return resampler_forward(res_sd, tok, depth=resampler["depth"], heads=resampler["heads"], dim_head=resampler["dim_head"])
