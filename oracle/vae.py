"""TEST INFRASTRUCTURE (CPU oracle) -- not part of the product.

Functional restatement (torch CPU fp32) of the first-stage decode of the reference:
  LatentDiffusion.decode_first_stage_2DAE      lvdm/models/ddpm3d.py:556-562   (per frame, z / scale_factor)
  AutoencoderKL.decode                          lvdm/models/autoencoder.py:103-107 (post_quant_conv -> Decoder)
  Decoder.forward / ResnetBlock / AttnBlock / Upsample   lvdm/modules/networks/ae_modules.py:466-579, 151-210, 26-78, 111-127
on the reference's own state-dict keys.  Pinned by tests/golden/vae_*.npz (outputs of the reference's modules run in
the build container, tests/golden/make_golden.py g14)."""
import torch
import torch.nn.functional as F

from dynamicscaler_amd.vae_spec import decoder_blocks


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
