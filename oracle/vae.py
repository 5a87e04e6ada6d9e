"""TEST INFRASTRUCTURE (CPU oracle) -- not part of the product.

Functional restatement (torch CPU fp32) of the first-stage decode of the reference:
  LatentDiffusion.decode_first_stage_2DAE      lvdm/models/ddpm3d.py:556-562   (per frame, z / scale_factor)
  AutoencoderKL.decode                          lvdm/models/autoencoder.py:103-107 (post_quant_conv -> Decoder)
  Decoder.forward / ResnetBlock / AttnBlock / Upsample   lvdm/modules/networks/ae_modules.py:466-579, 151-210, 26-78, 111-127
on the reference's own state-dict keys.  Pinned by tests/golden/vae_*.npz (outputs of the reference's modules run in
the build container, tests/golden/make_golden.py g14)."""
import torch
import torch.nn.functional as F

from dynamicscaler_amd.vae_spec import decoder_blocks, encoder_blocks


def _gn(sd, p, x):
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], eps=1e-6)          # Normalize(): ae_modules.py:15-16


def _swish(x):
    return x * torch.sigmoid(x)                                                        # nonlinearity(): :10-12


def _resblock(sd, p, x, cin, cout):
    h = F.conv2d(_swish(_gn(sd, p + ".norm1", x)), sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1)
    h = F.conv2d(_swish(_gn(sd, p + ".norm2", h)), sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1)
    if cin != cout:
        x = F.conv2d(x, sd[p + ".nin_shortcut.weight"], sd[p + ".nin_shortcut.bias"])
    return x + h


def _attn(sd, p, x):
    h = _gn(sd, p + ".norm", x)
    q = F.conv2d(h, sd[p + ".q.weight"], sd[p + ".q.bias"])
    k = F.conv2d(h, sd[p + ".k.weight"], sd[p + ".k.bias"])
    v = F.conv2d(h, sd[p + ".v.weight"], sd[p + ".v.bias"])
    b, c, hh, ww = q.shape
    w_ = torch.bmm(q.reshape(b, c, hh * ww).permute(0, 2, 1), k.reshape(b, c, hh * ww)) * (int(c) ** (-0.5))
    w_ = F.softmax(w_, dim=2)
    h = torch.bmm(v.reshape(b, c, hh * ww), w_.permute(0, 2, 1)).reshape(b, c, hh, ww)
    return x + F.conv2d(h, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])


@torch.no_grad()
def vae_decode(sd, dd, z):
    """AutoencoderKL.decode: z [B, z_channels, h, w] -> image [B, out_ch, 8h.., 8w..] (for ch_mult of length 4)."""
    h = F.conv2d(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
    for kind, p, cin, cout in decoder_blocks(dd):
        if kind == "conv_in":
            h = F.conv2d(h, sd[p + ".weight"], sd[p + ".bias"], padding=1)
        elif kind == "res":
            h = _resblock(sd, p, h, cin, cout)
        elif kind == "attn":
            h = _attn(sd, p, h)
        elif kind == "up":
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = F.conv2d(h, sd[p + ".conv.weight"], sd[p + ".conv.bias"], padding=1)
        elif kind == "norm_out":
            h = _swish(_gn(sd, p, h))
        elif kind == "conv_out":
            h = F.conv2d(h, sd[p + ".weight"], sd[p + ".bias"], padding=1)
    return h


@torch.no_grad()
def decode_first_stage_2dae(sd, dd, z, scale_factor=1.0):
    """ddpm3d.py:556-562: z [B,C,T,h,w] -> [B,3,T,H,W], frame by frame."""
    z = 1.0 / scale_factor * z
    return torch.cat([vae_decode(sd, dd, z[:, :, i]).unsqueeze(2) for i in range(z.shape[2])], dim=2)


@torch.no_grad()
def vae_encode_moments(sd, dd, x):
    """AutoencoderKL.encode up to the posterior parameters (autoencoder.py:97-101): x [B,3,H,W] -> moments [B,2*embed,h,w]
    = quant_conv(Encoder(x)) (Encoder.forward ae_modules.py:442-464; Downsample :102-106 pads bottom/right only)."""
    h = x
    for kind, p, cin, cout in encoder_blocks(dd):
        if kind == "conv_in":
            h = F.conv2d(h, sd[p + ".weight"], sd[p + ".bias"], padding=1)
        elif kind == "res":
            h = _resblock(sd, p, h, cin, cout)
        elif kind == "attn":
            h = _attn(sd, p, h)
        elif kind == "down":
            h = F.conv2d(F.pad(h, (0, 1, 0, 1), mode="constant", value=0), sd[p + ".conv.weight"], sd[p + ".conv.bias"], stride=2)
        elif kind == "norm_out":
            h = _swish(_gn(sd, p, h))
        elif kind == "conv_out":
            h = F.conv2d(h, sd[p + ".weight"], sd[p + ".bias"], padding=1)
    return F.conv2d(h, sd["quant_conv.weight"], sd["quant_conv.bias"])


def posterior_sample(moments, noise=None):
    """DiagonalGaussianDistribution(moments).sample() (lvdm/distributions.py:24-40); noise=None draws torch.randn(mean.shape)."""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    std = torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0))
    if noise is None:
        noise = torch.randn(mean.shape)
    return mean + std * noise


@torch.no_grad()
def encode_first_stage_2dae(sd, dd, x, scale_factor=1.0):
    """ddpm3d.py:485-490 + get_first_stage_encoding :458-465: x [B,3,T,H,W] -> scale_factor * sample, frame by frame
    (one torch.randn draw per frame, in frame order)."""
    return torch.cat([(scale_factor * posterior_sample(vae_encode_moments(sd, dd, x[:, :, i]))).unsqueeze(2)
                      for i in range(x.shape[2])], dim=2)


@torch.no_grad()
def tiled_vae_encode(sd, dd, image, scale_factor=1.0, vae_scale=8, h_tile_num=4, w_tile_num=4, overlap_h=32, overlap_w=32):
    """VC2_Pipeline_I2V_SpherePano.tiled_vae_encode_tensor_simple (i2v_sphere_panorama_pipeline.py:505-562):
    image [B,3,F,H,W] -> latent [B,4,F,H/8,W/8]; every tile is encoded with an overlap margin and cropped back to its own
    cell (so the count normaliser is 1 everywhere)."""
    B, _, Fr, H_dec, W_dec = image.shape
    Hl, Wl = H_dec // vae_scale, W_dec // vae_scale
    th, tw = Hl // h_tile_num, Wl // w_tile_num
    thi, twi = th * vae_scale, tw * vae_scale
    ovh, ovw = overlap_h * vae_scale, overlap_w * vae_scale
    out = torch.zeros((B, 4, Fr, Hl, Wl))
    count = torch.zeros((B, 1, 1, Hl, Wl))
    for i in range(h_tile_num):
        for j in range(w_tile_num):
            hs, he, ws, we = i * thi, (i + 1) * thi, j * twi, (j + 1) * twi
            hso, heo, wso, weo = max(hs - ovh, 0), min(he + ovh, H_dec), max(ws - ovw, 0), min(we + ovw, W_dec)
            lt = encode_first_stage_2dae(sd, dd, image[:, :, :, hso:heo, wso:weo], scale_factor)
            top, left = (hs - hso) // vae_scale, (ws - wso) // vae_scale
            bottom, right = lt.shape[3] - (heo - he) // vae_scale, lt.shape[4] - (weo - we) // vae_scale
            out[:, :, :, i * th:(i + 1) * th, j * tw:(j + 1) * tw] += lt[:, :, :, top:bottom, left:right]
            count[:, :, :, i * th:(i + 1) * th, j * tw:(j + 1) * tw] += 1
    return out / torch.clamp(count, min=1.0)
