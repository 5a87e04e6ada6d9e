#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: where does the fp16 build's distance to the fp32 reference come from?

The HIP UNet stores every activation in fp16 and accumulates in fp32.  This script replays the ORACLE (oracle/unet.py,
torch-CPU fp32) with fp16 roundings switched on at the places where the product rounds -- one class of tensor at a
time -- and prints the rel-L2 distance of eps to the plain fp32 oracle.  It is the CPU half of the layer-wise error
budget; the GPU half (tests/test_gpu_unet.py::test_error_budget_*) compares block outputs of
the real kernels with the same oracle.

    python tests/error_budget.py [--full] [--variants 'res;norm;mid;p;res,norm,mid,p']

Rounding classes (a tensor is rounded to fp16 and back where the product stores it in fp16):
  res   the residual stream: outputs of every "+ x" (ResBlock out, temporal-conv block out, transformer block adds,
        proj_out + x_in) and the convolutions that write it (conv_in, down / up-sample)
  norm  GroupNorm(+SiLU) / LayerNorm outputs (the A operand of the following GEMM)
  mid   GEMM outputs that are not on the residual stream (ResBlock conv1, temporal convs 1-3, proj_in, q/k/v, GEGLU
        hidden, attention output, time-embedding MLP)
  p     softmax probabilities (the P operand of P.V)
  res@C the residual stream only where it is C channels wide (per-level storage choice)
  resin / resout  the residual stream inside the transformer blocks (their three adds) / everywhere else
  midq  the mid class without the tensors that only a GroupNorm reads (ResBlock conv1 output, temporal convs 1-3)
"""
import argparse
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from oracle import unet as ou  # noqa: E402
from dynamicscaler_amd.unet_spec import param_shapes  # noqa: E402
from dynamicscaler_amd.synth import synth_state_dict, synth_normal  # noqa: E402


def h16(t):
    return t.half().float()


class QNet(ou._Net):
    """oracle _Net with fp16 roundings at the product's storage points."""

    def __init__(self, sd, params, classes):
        super().__init__(sd, params)
        self.q = set(classes)

    def r(self, t, cls):
        """Round `t` if its class is on.  "res@C" rounds the residual stream only where it is C channels wide (320 / 640 / 1280 =
        the UNet's resolution levels 1 / 2 / 3+4): what a per-level choice of the stream's storage type would do."""
        if cls in self.q:
            return h16(t)
        if cls == "res":
            ch = t.shape[-1] if t.dim() == 3 else t.shape[1]
            if f"res@{ch}" in self.q:
                return h16(t)
        if cls == "resin":                                 # the stream INSIDE a transformer (the three adds of a block): part of "res",
            if "res" in self.q or "resin" in self.q:       # selectable alone; "resout" = the res class without it
                return h16(t)
            ch = t.shape[-1]
            return h16(t) if f"res@{ch}" in self.q else t
        if cls == "res" and "resout" in self.q:
            return h16(t)
        if cls == "midgn" and ("mid" in self.q):          # a mid tensor that only a GroupNorm reads (conv1 out, temporal convs 1-3)
            return h16(t)
        if cls == "mid" and "midq" in self.q:             # "midq": every mid tensor EXCEPT the GroupNorm-fed ones
            return h16(t)
        return t

    def gn(self, x, prefix, eps):
        return super().gn(x, prefix, eps)  # rounded by the caller after the SiLU (the product fuses GN+SiLU)

    def ln(self, x, prefix):
        return self.r(super().ln(x, prefix), "norm")

    def attention(self, x, prefix, heads, context=None, img_cross=False):
        q = self.r(self.lin(x, prefix + ".to_q", bias=False), "mid")
        ctx = x if context is None else context
        ctx_img = None
        if context is not None and img_cross:
            ctx, ctx_img = ctx[:, :77, :], ctx[:, 77:, :]
        k = self.r(self.lin(ctx, prefix + ".to_k", bias=False), "mid")
        v = self.r(self.lin(ctx, prefix + ".to_v", bias=False), "mid")
        b, n, _ = q.shape
        dh = q.shape[-1] // heads
        scale = dh ** -0.5

        def split(t):
            return t.reshape(b, t.shape[1], heads, dh).permute(0, 2, 1, 3).reshape(b * heads, t.shape[1], dh)

        qh, kh, vh = split(q), split(k), split(v)

        def sdpa(kh_, vh_):
            sim = torch.einsum("bid,bjd->bij", qh, kh_) * scale
            if "p" in self.q:
                # the flash kernel rounds exp(s - max) (in [0,1]) to fp16 for the P.V MFMA and divides by the fp32 row sum
                m = sim.amax(-1, keepdim=True)
                e = torch.exp(sim - m)
                o = torch.einsum("bij,bjd->bid", h16(e), vh_) / e.sum(-1, keepdim=True)
            else:
                o = torch.einsum("bij,bjd->bid", sim.softmax(-1), vh_)
            return o.reshape(b, heads, n, dh).permute(0, 2, 1, 3).reshape(b, n, heads * dh)

        out = sdpa(kh, vh)
        if ctx_img is not None:
            k_ip = split(self.r(self.lin(ctx_img, prefix + ".to_k_ip", bias=False), "mid"))
            v_ip = split(self.r(self.lin(ctx_img, prefix + ".to_v_ip", bias=False), "mid"))
            out = out + sdpa(k_ip, v_ip)
        out = self.r(out, "mid")
        return self.lin(out, prefix + ".to_out.0")

    def transformer_block(self, x, prefix, heads, context, img_cross):
        x = self.r(self.attention(self.ln(x, prefix + ".norm1"), prefix + ".attn1", heads) + x, "resin")
        x = self.r(self.attention(self.ln(x, prefix + ".norm2"), prefix + ".attn2", heads, context=context,
                                  img_cross=img_cross) + x, "resin")
        h = self.lin(self.ln(x, prefix + ".norm3"), prefix + ".ff.net.0.proj")
        a, gate = h.chunk(2, dim=-1)
        h = self.r(a * F.gelu(gate), "mid")
        return self.r(self.lin(h, prefix + ".ff.net.2") + x, "resin")

    def spatial_transformer(self, x, prefix, heads, context):
        c = self.c
        assert c["use_linear"]
        bt, ch, hh, ww = x.shape
        x_in = x
        x = self.r(super().gn(x, prefix + ".norm", 1e-6), "norm")
        x = x.permute(0, 2, 3, 1).reshape(bt, hh * ww, -1)
        x = self.r(self.lin(x, prefix + ".proj_in"), "mid")
        for d in range(c["transformer_depth"]):
            x = self.transformer_block(x, f"{prefix}.transformer_blocks.{d}", heads, context, c["use_image_attention"])
        x = self.lin(x, prefix + ".proj_out")
        x = x.reshape(bt, hh, ww, -1).permute(0, 3, 1, 2)
        return self.r(x + x_in, "res")

    def temporal_transformer(self, x, prefix, heads, depth):
        b, ch, t, hh, ww = x.shape
        x_in = x
        x = self.r(super().gn(x, prefix + ".norm", 1e-6), "norm")
        x = x.permute(0, 3, 4, 1, 2).reshape(b * hh * ww, ch, t).permute(0, 2, 1)

        def proj(v, name):   # nn.Linear, or a Conv1d with kernel 1 (the same contraction, weight [C, C, 1])
            w = self.p(f"{prefix}.{name}.weight")
            return F.linear(v, w.reshape(w.shape[0], w.shape[1]), self.p(f"{prefix}.{name}.bias"))

        x = self.r(proj(x, "proj_in"), "mid")
        for d in range(depth):
            x = self.transformer_block(x, f"{prefix}.transformer_blocks.{d}", heads, None, False)
        x = proj(x, "proj_out")
        x = x.reshape(b, hh, ww, t, ch).permute(0, 4, 3, 1, 2)
        return self.r(x + x_in, "res")

    def temporal_conv_block(self, x, prefix):
        identity = x
        for i in (1, 2, 3, 4):
            idx = 2 if i == 1 else 3
            x = self.r(F.silu(super().gn(x, f"{prefix}.conv{i}.0", 1e-5)), "norm")
            x = F.conv3d(x, self.p(f"{prefix}.conv{i}.{idx}.weight"), self.p(f"{prefix}.conv{i}.{idx}.bias"), padding=(1, 0, 0))
            if i < 4:
                x = self.r(x, "midgn")
        return self.r(x + identity, "res")

    def resblock(self, x, emb, prefix, cin, cout, b):
        h = self.r(F.silu(super().gn(x, prefix + ".in_layers.0", 1e-5)), "norm")
        h = F.conv2d(h, self.p(prefix + ".in_layers.2.weight"), self.p(prefix + ".in_layers.2.bias"), padding=1)
        emb_out = self.lin(F.silu(emb), prefix + ".emb_layers.1")
        h = self.r(h + emb_out[..., None, None], "midgn")
        h = self.r(F.silu(super().gn(h, prefix + ".out_layers.0", 1e-5)), "norm")
        h = F.conv2d(h, self.p(prefix + ".out_layers.3.weight"), self.p(prefix + ".out_layers.3.bias"), padding=1)
        if cin != cout:
            x = self.r(F.conv2d(x, self.p(prefix + ".skip_connection.weight"), self.p(prefix + ".skip_connection.bias")), "res")
        h = self.r(x + h, "res")
        if self.c["temporal_conv"] and self.has(prefix + ".temopral_conv.conv1.0.weight"):
            bt, ch, hh, ww = h.shape
            h5 = h.reshape(b, bt // b, ch, hh, ww).permute(0, 2, 1, 3, 4)
            h5 = self.temporal_conv_block(h5, prefix + ".temopral_conv")
            h = h5.permute(0, 2, 1, 3, 4).reshape(bt, ch, hh, ww)
        return h

    def run_layers(self, h, layers, prefix, emb, context, b):
        for j, layer in enumerate(layers):
            kind = layer[0]
            h = self._one(h, j, layer, prefix, emb, context, b)
            if kind in ("conv_in", "down", "up"):
                h = self.r(h, "res")
        return h

    def _one(self, h, j, layer, prefix, emb, context, b):
        # same dispatch as the oracle's run_layers for a single layer at index j
        p = f"{prefix}.{j}"
        kind = layer[0]
        if kind == "conv_in":
            return F.conv2d(h, self.p(p + ".weight"), self.p(p + ".bias"), padding=1)
        if kind == "res":
            return self.resblock(h, emb, p, layer[1], layer[2], b)
        if kind == "st":
            return self.spatial_transformer(h, p, layer[2], context)
        if kind == "tt":
            bt, ch, hh, ww = h.shape
            h5 = h.reshape(b, bt // b, ch, hh, ww).permute(0, 2, 1, 3, 4)
            h5 = self.temporal_transformer(h5, p, layer[2], self.c["temporal_transformer_depth"])
            return h5.permute(0, 2, 1, 3, 4).reshape(bt, ch, hh, ww)
        if kind == "down":
            return F.conv2d(h, self.p(p + ".op.weight"), self.p(p + ".op.bias"), stride=2, padding=1)
        if kind == "up":
            h = F.interpolate(h, scale_factor=2, mode="nearest")
            return F.conv2d(h, self.p(p + ".conv.weight"), self.p(p + ".conv.bias"), padding=1)
        raise AssertionError(kind)


@torch.no_grad()
def forward(sd, params, x, timesteps, context, fps, classes):
    """oracle.unet.unet_forward with the roundings of `classes` (empty = the plain fp32 oracle)."""
    net = QNet(sd, params, classes)
    c = net.c
    mc = c["model_channels"]
    t_emb = net.r(ou.timestep_embedding(timesteps, mc), "mid")
    emb = net.lin(net.r(F.silu(net.lin(t_emb, "time_embed.0")), "mid"), "time_embed.2")
    if c["fps_cond"]:
        fps_t = torch.full_like(timesteps, fps)
        fps_emb = net.r(ou.timestep_embedding(fps_t, mc), "mid")
        emb = net.r(emb, "mid") + net.lin(net.r(F.silu(net.lin(fps_emb, "fps_embedding.0")), "mid"), "fps_embedding.2")
    emb = net.r(emb, "mid")
    b, _, t, hh, ww = x.shape
    context = context.repeat_interleave(repeats=t, dim=0)
    emb = emb.repeat_interleave(repeats=t, dim=0)
    h = x.permute(0, 2, 1, 3, 4).reshape(b * t, x.shape[1], hh, ww)
    hs = []
    lay = net.layout
    for i, layers in enumerate(lay["input"]):
        h = net.run_layers(h, layers, f"input_blocks.{i}", emb, context, b)
        if i == 0 and c["addition_attention"]:
            bt, ch, h2, w2 = h.shape
            h5 = h.reshape(b, t, ch, h2, w2).permute(0, 2, 1, 3, 4)
            h5 = net.temporal_transformer(h5, "init_attn.0", 8, c["transformer_depth"])
            h = h5.permute(0, 2, 1, 3, 4).reshape(bt, ch, h2, w2)
        hs.append(h)
    h = net.run_layers(h, lay["middle"], "middle_block", emb, context, b)
    for i, layers in enumerate(lay["output"]):
        h = torch.cat([h, hs.pop()], dim=1)
        h = net.run_layers(h, layers, f"output_blocks.{i}", emb, context, b)
    h = net.r(F.silu(ou._Net.gn(net, h, "out.0", 1e-5)), "norm")
    y = F.conv2d(h, net.p("out.2.weight"), net.p("out.2.bias"), padding=1)
    return y.reshape(b, t, -1, hh, ww).permute(0, 2, 1, 3, 4).contiguous()


def rel(a, b):
    return float((a - b).norm() / b.norm())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true", help="the 1.41 B t2v UNet at [1,4,16,40,64] (about a minute per variant)")
    ap.add_argument("--variants", default="res;norm;mid;p;norm,mid,p;res,norm,mid,p")
    ap.add_argument("--t", type=int, default=None, help="timestep (default: the golden's, 499)")
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count())
    gold = os.path.join(REPO, "tests", "golden")
    if args.full:
        z = np.load(os.path.join(gold, "unet_full_t2v.npz"))
        import yaml
        params = yaml.safe_load(open(os.path.join(REPO, "dynamicscaler_amd", "configs", "t2v_512_v2_unet.yaml")))
        sd = synth_state_dict(param_shapes(params), 0)
        x, t, fps = torch.from_numpy(z["x"]), torch.tensor([int(z["t"])]), int(z["fps"])
        ctx = synth_normal((1, 77, 1024), 1)
        ref_gold = torch.from_numpy(z["eps_cond"])
        if args.t is not None:
            t, ref_gold = torch.tensor([args.t]), None
    else:
        z = np.load(os.path.join(gold, "unet_tiny_t2v.npz"))
        params = json.loads(bytes(z["params_json"]).decode())
        sd = synth_state_dict(param_shapes(params), 5)
        x, ctx, t, fps = (torch.from_numpy(z["x_0"]), torch.from_numpy(z["ctx_0"]), torch.from_numpy(z["t_0"]), int(z["fps_0"]))
        ref_gold = torch.from_numpy(z["eps_0"])
    base = forward(sd, params, x, t, ctx, fps, ())
    if ref_gold is not None:
        print(f"fp32 (no roundings) vs the reference golden: {rel(base, ref_gold):.3e}")
    for v in args.variants.split(";"):
        cls = tuple(c for c in v.split(",") if c)
        e = forward(sd, params, x, t, ctx, fps, cls)
        print(f"roundings {v:24s}: eps rel-L2 vs fp32 {rel(e, base):.3e}", flush=True)


if __name__ == "__main__":
    main()
