"""TEST INFRASTRUCTURE (oracle): VideoCrafter LVDM 3D-UNet forward on torch-CPU fp32.

A functional restatement of the reference network operating directly on a state dict with
the reference's key names (e.g. `input_blocks.1.0.temopral_conv.conv1.0.weight`):

  * UNetModel ctor / forward            lvdm/modules/networks/openaimodel3d.py:340-708
  * ResBlock                            openaimodel3d.py:115-254
  * TemporalConvBlock                   openaimodel3d.py:257-309
  * Downsample / Upsample               openaimodel3d.py:48-112
  * SpatialTransformer                  lvdm/modules/attention.py:223-278
  * TemporalTransformer                 attention.py:281-373 (only_self_att, no rel-pos, no causal mask)
  * BasicTransformerBlock               attention.py:187-220
  * CrossAttention (einsum path)        attention.py:76-127 (+ image-token branch :82-87,117-124)
  * GEGLU / FeedForward                 attention.py:376-403
  * GroupNormSpecific                   lvdm/basics.py:76-86
  * timestep_embedding                  lvdm/models/utils_diffusion.py:8-28
  * DiffusionWrapper 'crossattn'        lvdm/models/ddpm3d.py:710-712

Deliberate extension (documented, inert at b=1): the reference cannot run batch>1 because
`emb` is not repeated over frames (openaimodel3d.py:678-682 vs :237-246).
Here `emb` is repeat_interleave'd over t, which is the identity at b=1 and gives the
"b independent b=1 forwards" semantics for b>1.
"""
import math
import torch
import torch.nn.functional as F

DEFAULTS = dict(dropout=0.0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, context_dim=None,
                use_scale_shift_norm=False, resblock_updown=False, num_heads=-1, num_head_channels=-1,
                transformer_depth=1, use_linear=False, use_checkpoint=False, temporal_conv=False,
                tempspatial_aware=False, temporal_attention=True, temporal_selfatt_only=True,
                use_relative_position=True, use_causal_attention=False, temporal_length=None,
                use_fp16=False, addition_attention=False, use_image_attention=False,
                temporal_transformer_depth=1, fps_cond=False, time_cond_proj_dim=None)


def _cfg(params):
    c = dict(DEFAULTS)
    c.update(params)
    unsupported = []
    if c["use_scale_shift_norm"]:
        unsupported.append("use_scale_shift_norm")
    if c["resblock_updown"]:
        unsupported.append("resblock_updown")
    if c["use_relative_position"]:
        unsupported.append("use_relative_position")
    if c["use_causal_attention"]:
        unsupported.append("use_causal_attention")
    if not c["temporal_selfatt_only"]:
        unsupported.append("temporal_selfatt_only=False")
    if c["tempspatial_aware"]:
        unsupported.append("tempspatial_aware")
    if c["dims"] != 2 or not c["conv_resample"]:
        unsupported.append("dims!=2 / conv_resample=False")
    if unsupported:
        raise NotImplementedError("oracle UNet covers the options of configs/inference_{t2v,i2v}_512*.yaml; "
                                  "unsupported: " + ", ".join(unsupported))
    return c


def unet_layout(params):
    """Enumerate the module tree exactly like UNetModel.__init__ (openaimodel3d.py:421-655).

    Returns dict(input=[[layer,...],...], middle=[...], output=[[...],...]) where a layer is
    ('conv_in',), ('res', cin, cout), ('st', ch, heads, dim_head), ('tt', ch, heads, dim_head),
    ('down', ch), ('up', ch)."""
    c = _cfg(params)
    mc = c["model_channels"]
    nhc = c["num_head_channels"]
    num_heads = c["num_heads"]

    def heads_of(ch):
        if nhc == -1:
            return num_heads, ch // num_heads
        return ch // nhc, nhc

    inp = [[("conv_in",)]]
    chans = [mc]
    ch = mc
    ds = 1
    cm = list(c["channel_mult"])
    for level, mult in enumerate(cm):
        for _ in range(c["num_res_blocks"]):
            layers = [("res", ch, mult * mc)]
            ch = mult * mc
            if ds in c["attention_resolutions"]:
                h, d = heads_of(ch)
                layers.append(("st", ch, h, d))
                if c["temporal_attention"]:
                    layers.append(("tt", ch, h, d))
            inp.append(layers)
            chans.append(ch)
        if level != len(cm) - 1:
            inp.append([("down", ch)])
            chans.append(ch)
            ds *= 2
    h, d = heads_of(ch)
    mid = [("res", ch, ch), ("st", ch, h, d)]
    if c["temporal_attention"]:
        mid.append(("tt", ch, h, d))
    mid.append(("res", ch, ch))
    out = []
    for level, mult in list(enumerate(cm))[::-1]:
        for i in range(c["num_res_blocks"] + 1):
            ich = chans.pop()
            layers = [("res", ch + ich, mult * mc)]
            ch = mult * mc
            if ds in c["attention_resolutions"]:
                h, d = heads_of(ch)
                layers.append(("st", ch, h, d))
                if c["temporal_attention"]:
                    layers.append(("tt", ch, h, d))
            if level and i == c["num_res_blocks"]:
                layers.append(("up", ch))
                ds //= 2
            out.append(layers)
    return dict(input=inp, middle=mid, output=out, cfg=c)


def timestep_embedding(timesteps, dim, max_period=10000):
    """utils_diffusion.py:8-28."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


class _Net:
    def __init__(self, sd, params):
        self.sd = sd
        self.layout = unet_layout(params)
        self.c = self.layout["cfg"]
        self.tap = None   # optional callable(name, h [b*t, C, H, W]) after every block (layer-wise error budget tests)

    def p(self, key):
        return self.sd[key]

    def has(self, key):
        return key in self.sd

    def lin(self, x, prefix, bias=True):
        return F.linear(x, self.p(prefix + ".weight"), self.p(prefix + ".bias") if bias else None)

    def gn(self, x, prefix, eps):
        return F.group_norm(x, 32, self.p(prefix + ".weight"), self.p(prefix + ".bias"), eps)

    def ln(self, x, prefix):
        w = self.p(prefix + ".weight")
        return F.layer_norm(x, (w.shape[0],), w, self.p(prefix + ".bias"), 1e-5)

    # ---- attention.py:76-127 ----
    def attention(self, x, prefix, heads, context=None, img_cross=False):
        q = self.lin(x, prefix + ".to_q", bias=False)
        ctx = x if context is None else context
        ctx_img = None
        if context is not None and img_cross:
            ctx, ctx_img = ctx[:, :77, :], ctx[:, 77:, :]  # text_context_len = 77 (attention.py:60)
        k = self.lin(ctx, prefix + ".to_k", bias=False)
        v = self.lin(ctx, prefix + ".to_v", bias=False)
        b, n, _ = q.shape
        dh = q.shape[-1] // heads
        scale = dh ** -0.5

        def split(t):
            return t.reshape(b, t.shape[1], heads, dh).permute(0, 2, 1, 3).reshape(b * heads, t.shape[1], dh)

        qh, kh, vh = split(q), split(k), split(v)
        sim = torch.einsum("bid,bjd->bij", qh, kh) * scale
        sim = sim.softmax(dim=-1)
        out = torch.einsum("bij,bjd->bid", sim, vh)
        out = out.reshape(b, heads, n, dh).permute(0, 2, 1, 3).reshape(b, n, heads * dh)
        if ctx_img is not None:
            k_ip = split(self.lin(ctx_img, prefix + ".to_k_ip", bias=False))
            v_ip = split(self.lin(ctx_img, prefix + ".to_v_ip", bias=False))
            sim_ip = (torch.einsum("bid,bjd->bij", qh, k_ip) * scale).softmax(dim=-1)
            out_ip = torch.einsum("bij,bjd->bid", sim_ip, v_ip)
            out_ip = out_ip.reshape(b, heads, n, dh).permute(0, 2, 1, 3).reshape(b, n, heads * dh)
            out = out + 1.0 * out_ip  # image_cross_attention_scale (attention.py:59)
        return self.lin(out, prefix + ".to_out.0")

    # ---- attention.py:187-220, 376-403 ----
    def transformer_block(self, x, prefix, heads, context, img_cross):
        x = self.attention(self.ln(x, prefix + ".norm1"), prefix + ".attn1", heads) + x
        x = self.attention(self.ln(x, prefix + ".norm2"), prefix + ".attn2", heads,
                           context=context, img_cross=img_cross) + x
        h = self.lin(self.ln(x, prefix + ".norm3"), prefix + ".ff.net.0.proj")
        a, gate = h.chunk(2, dim=-1)
        h = a * F.gelu(gate)
        return self.lin(h, prefix + ".ff.net.2") + x

    # ---- attention.py:223-278 (use_linear and conv variants) ----
    def spatial_transformer(self, x, prefix, heads, context):
        c = self.c
        bt, ch, hh, ww = x.shape
        x_in = x
        x = self.gn(x, prefix + ".norm", 1e-6)
        if not c["use_linear"]:
            x = F.conv2d(x, self.p(prefix + ".proj_in.weight"), self.p(prefix + ".proj_in.bias"))
        x = x.permute(0, 2, 3, 1).reshape(bt, hh * ww, -1)
        if c["use_linear"]:
            x = self.lin(x, prefix + ".proj_in")
        for d in range(c["transformer_depth"]):
            x = self.transformer_block(x, f"{prefix}.transformer_blocks.{d}", heads, context,
                                       c["use_image_attention"])
        if c["use_linear"]:
            x = self.lin(x, prefix + ".proj_out")
        x = x.reshape(bt, hh, ww, -1).permute(0, 3, 1, 2)
        if not c["use_linear"]:
            x = F.conv2d(x, self.p(prefix + ".proj_out.weight"), self.p(prefix + ".proj_out.bias"))
        return x + x_in

    # ---- attention.py:281-373 ----
    def temporal_transformer(self, x, prefix, heads, depth):
        c = self.c
        b, ch, t, hh, ww = x.shape
        x_in = x
        x = self.gn(x, prefix + ".norm", 1e-6)
        x = x.permute(0, 3, 4, 1, 2).reshape(b * hh * ww, ch, t)  # (b h w) c t
        lin = self.p(prefix + ".proj_in.weight").dim() == 2
        if not lin:
            x = F.conv1d(x, self.p(prefix + ".proj_in.weight"), self.p(prefix + ".proj_in.bias"))
        x = x.permute(0, 2, 1)  # bhw t c
        if lin:
            x = self.lin(x, prefix + ".proj_in")
        for d in range(depth):
            # only_self_att: both attn1 and attn2 are self-attention (context=None)
            x = self.transformer_block(x, f"{prefix}.transformer_blocks.{d}", heads, None, False)
        if lin:
            x = self.lin(x, prefix + ".proj_out")
            x = x.reshape(b, hh, ww, t, ch).permute(0, 4, 3, 1, 2)
        else:
            x = x.permute(0, 2, 1)
            x = F.conv1d(x, self.p(prefix + ".proj_out.weight"), self.p(prefix + ".proj_out.bias"))
            x = x.reshape(b, hh, ww, ch, t).permute(0, 3, 4, 1, 2)
        return x + x_in

    # ---- openaimodel3d.py:257-309 ----
    def temporal_conv_block(self, x, prefix):
        identity = x
        for i in (1, 2, 3, 4):
            idx = 2 if i == 1 else 3
            x = F.silu(self.gn(x, f"{prefix}.conv{i}.0", 1e-5))
            x = F.conv3d(x, self.p(f"{prefix}.conv{i}.{idx}.weight"), self.p(f"{prefix}.conv{i}.{idx}.bias"),
                         padding=(1, 0, 0))
        return x + identity

    # ---- openaimodel3d.py:115-254 ----
    def resblock(self, x, emb, prefix, cin, cout, b):
        h = F.silu(self.gn(x, prefix + ".in_layers.0", 1e-5))
        h = F.conv2d(h, self.p(prefix + ".in_layers.2.weight"), self.p(prefix + ".in_layers.2.bias"), padding=1)
        emb_out = self.lin(F.silu(emb), prefix + ".emb_layers.1")
        h = h + emb_out[..., None, None]
        h = F.silu(self.gn(h, prefix + ".out_layers.0", 1e-5))
        h = F.conv2d(h, self.p(prefix + ".out_layers.3.weight"), self.p(prefix + ".out_layers.3.bias"), padding=1)
        if cin != cout:
            x = F.conv2d(x, self.p(prefix + ".skip_connection.weight"), self.p(prefix + ".skip_connection.bias"))
        h = x + h
        if self.c["temporal_conv"] and self.has(prefix + ".temopral_conv.conv1.0.weight"):
            bt, ch, hh, ww = h.shape
            h5 = h.reshape(b, bt // b, ch, hh, ww).permute(0, 2, 1, 3, 4)
            h5 = self.temporal_conv_block(h5, prefix + ".temopral_conv")
            h = h5.permute(0, 2, 1, 3, 4).reshape(bt, ch, hh, ww)
        return h

    def run_layers(self, h, layers, prefix, emb, context, b):
        for j, layer in enumerate(layers):
            p = f"{prefix}.{j}"
            kind = layer[0]
            if kind == "conv_in":
                h = F.conv2d(h, self.p(p + ".weight"), self.p(p + ".bias"), padding=1)
            elif kind == "res":
                h = self.resblock(h, emb, p, layer[1], layer[2], b)
            elif kind == "st":
                h = self.spatial_transformer(h, p, layer[2], context)
            elif kind == "tt":
                bt, ch, hh, ww = h.shape
                h5 = h.reshape(b, bt // b, ch, hh, ww).permute(0, 2, 1, 3, 4)
                h5 = self.temporal_transformer(h5, p, layer[2], self.c["temporal_transformer_depth"])
                h = h5.permute(0, 2, 1, 3, 4).reshape(bt, ch, hh, ww)
            elif kind == "down":
                h = F.conv2d(h, self.p(p + ".op.weight"), self.p(p + ".op.bias"), stride=2, padding=1)
            elif kind == "up":
                h = F.interpolate(h, scale_factor=2, mode="nearest")
                h = F.conv2d(h, self.p(p + ".conv.weight"), self.p(p + ".conv.bias"), padding=1)
            else:
                raise AssertionError(kind)
            if self.tap is not None:
                self.tap(p, h)
        return h


@torch.no_grad()
def unet_forward(sd, params, x, timesteps, context, fps=16, tap=None):
    """UNetModel.forward (openaimodel3d.py:657-708).  x [b,C,t,h,w] fp32, timesteps int64 [b],
    context [b,L,context_dim], fps python int or int64 tensor [b].  Returns eps [b,C_out,t,h,w].
    tap(name, h): called with the activation [b*t, C, H, W] after every block (names = state-dict prefixes)."""
    net = _Net(sd, params)
    net.tap = tap
    c = net.c
    mc = c["model_channels"]
    t_emb = timestep_embedding(timesteps, mc)
    emb = net.lin(F.silu(net.lin(t_emb, "time_embed.0")), "time_embed.2")
    if c["fps_cond"]:
        if isinstance(fps, int):
            fps = torch.full_like(timesteps, fps)
        fps_emb = timestep_embedding(fps, mc)
        emb = emb + net.lin(F.silu(net.lin(fps_emb, "fps_embedding.0")), "fps_embedding.2")
    b, _, t, hh, ww = x.shape
    context = context.repeat_interleave(repeats=t, dim=0)
    emb = emb.repeat_interleave(repeats=t, dim=0)  # identity at b=1 (see module docstring)
    h = x.permute(0, 2, 1, 3, 4).reshape(b * t, x.shape[1], hh, ww)
    hs = []
    lay = net.layout
    for i, layers in enumerate(lay["input"]):
        h = net.run_layers(h, layers, f"input_blocks.{i}", emb, context, b)
        if i == 0 and c["addition_attention"]:
            bt, ch, h2, w2 = h.shape
            h5 = h.reshape(b, t, ch, h2, w2).permute(0, 2, 1, 3, 4)
            # init_attn: n_heads=8, d_head=num_head_channels, depth=transformer_depth (openaimodel3d.py:425-439)
            h5 = net.temporal_transformer(h5, "init_attn.0", 8, c["transformer_depth"])
            h = h5.permute(0, 2, 1, 3, 4).reshape(bt, ch, h2, w2)
            if tap is not None:
                tap("init_attn.0", h)
        hs.append(h)
    h = net.run_layers(h, lay["middle"], "middle_block", emb, context, b)
    for i, layers in enumerate(lay["output"]):
        h = torch.cat([h, hs.pop()], dim=1)
        h = net.run_layers(h, layers, f"output_blocks.{i}", emb, context, b)
    h = F.silu(net.gn(h, "out.0", 1e-5))
    y = F.conv2d(h, net.p("out.2.weight"), net.p("out.2.bias"), padding=1)
    return y.reshape(b, t, -1, hh, ww).permute(0, 2, 1, 3, 4).contiguous()


@torch.no_grad()
def diffusion_wrapper_forward(sd, params, x, t, c_crossattn, fps=16, **_ignored):
    """DiffusionWrapper.forward, conditioning_key='crossattn' (ddpm3d.py:710-712).
    Extra kwargs (curr_time_steps, temporal_length, clean_cond) are swallowed like
    UNetModel.forward's **kwargs (openaimodel3d.py:665)."""
    cc = torch.cat(c_crossattn, 1)
    return unet_forward(sd, params, x, t, cc, fps=fps)
