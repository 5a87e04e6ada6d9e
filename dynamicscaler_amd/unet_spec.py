"""Structure of the VideoCrafter LVDM 3D-UNet as a flat program + parameter table.

The reference builds the network as an nn.Module tree (lvdm/modules/networks/openaimodel3d.py:340-655);
this build needs only (a) the list of blocks in execution order and (b) the parameter names / shapes,
which must equal the reference's state-dict keys so VideoCrafter checkpoints load unchanged
(including the `temopral_conv` spelling, openaimodel3d.py:196).
"""
from collections import OrderedDict

UNET_DEFAULTS = dict(
    dropout=0.0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, context_dim=None,
    use_scale_shift_norm=False, resblock_updown=False, num_heads=-1, num_head_channels=-1,
    transformer_depth=1, use_linear=False, use_checkpoint=False, temporal_conv=False,
    tempspatial_aware=False, temporal_attention=True, temporal_selfatt_only=True,
    use_relative_position=True, use_causal_attention=False, temporal_length=None, use_fp16=False,
    addition_attention=False, use_image_attention=False, temporal_transformer_depth=1,
    fps_cond=False, time_cond_proj_dim=None,
)

# Options of UNetModel.__init__ that the two shipped yaml configs never enable and the HIP path
# therefore does not implement (they would need kernels that do not exist yet).
_UNSUPPORTED = (
    ("use_scale_shift_norm", True), ("resblock_updown", True), ("use_relative_position", True),
    ("use_causal_attention", True), ("temporal_selfatt_only", False), ("tempspatial_aware", True),
    ("conv_resample", False),
)


class Block:
    """One entry of the forward program. kind in {conv_in, res, st, tt, down, up}."""
    __slots__ = ("kind", "prefix", "cin", "cout", "heads", "dim_head", "depth", "linear_proj", "tconv")

    def __init__(self, kind, prefix, cin=0, cout=0, heads=0, dim_head=0, depth=1, linear_proj=True, tconv=False):
        self.kind, self.prefix = kind, prefix
        self.cin, self.cout, self.heads, self.dim_head = cin, cout, heads, dim_head
        self.depth, self.linear_proj, self.tconv = depth, linear_proj, tconv

    def __repr__(self):
        return f"Block({self.kind}, {self.prefix}, {self.cin}->{self.cout}, h={self.heads})"


def normalize_config(params):
    cfg = dict(UNET_DEFAULTS)
    cfg.update(params)
    if cfg["dims"] != 2:
        raise NotImplementedError("only dims=2 (the VideoCrafter configs) is supported")
    bad = [k for k, v in _UNSUPPORTED if cfg[k] == v]
    if bad:
        raise NotImplementedError(f"UNetModel options not implemented by the HIP path: {bad}")
    if cfg["num_heads"] == -1 and cfg["num_head_channels"] == -1:
        raise AssertionError("Either num_heads or num_head_channels has to be set")
    cfg["channel_mult"] = tuple(cfg["channel_mult"])
    cfg["attention_resolutions"] = tuple(cfg["attention_resolutions"])
    return cfg


def build_program(params):
    """Return (cfg, input_groups, middle_group, output_groups); each group is a list of Blocks.
    Mirrors the constructor's bookkeeping of `ch`, `ds`, `input_block_chans` (openaimodel3d.py:441-649)."""
    cfg = normalize_config(params)
    mc = cfg["model_channels"]

    def heads_for(ch):
        if cfg["num_head_channels"] == -1:
            return cfg["num_heads"], ch // cfg["num_heads"]
        return ch // cfg["num_head_channels"], cfg["num_head_channels"]

    def attn_blocks(prefix, start, ch):
        nh, dh = heads_for(ch)
        blocks = [Block("st", f"{prefix}.{start}", ch, ch, nh, dh, cfg["transformer_depth"], cfg["use_linear"])]
        if cfg["temporal_attention"]:
            blocks.append(Block("tt", f"{prefix}.{start + 1}", ch, ch, nh, dh,
                                cfg["temporal_transformer_depth"], cfg["use_linear"]))
        return blocks

    tconv = bool(cfg["temporal_conv"])
    inputs = [[Block("conv_in", "input_blocks.0.0", cfg["in_channels"], mc)]]
    skip_chans = [mc]
    ch, ds = mc, 1
    mults = cfg["channel_mult"]
    for level, mult in enumerate(mults):
        for _ in range(cfg["num_res_blocks"]):
            idx = len(inputs)
            group = [Block("res", f"input_blocks.{idx}.0", ch, mult * mc, tconv=tconv)]
            ch = mult * mc
            if ds in cfg["attention_resolutions"]:
                group += attn_blocks(f"input_blocks.{idx}", 1, ch)
            inputs.append(group)
            skip_chans.append(ch)
        if level != len(mults) - 1:
            idx = len(inputs)
            inputs.append([Block("down", f"input_blocks.{idx}.0", ch, ch)])
            skip_chans.append(ch)
            ds *= 2
    middle = [Block("res", "middle_block.0", ch, ch, tconv=tconv)]
    middle += attn_blocks("middle_block", 1, ch)
    middle.append(Block("res", f"middle_block.{len(middle)}", ch, ch, tconv=tconv))
    outputs = []
    for level, mult in list(enumerate(mults))[::-1]:
        for i in range(cfg["num_res_blocks"] + 1):
            idx = len(outputs)
            ich = skip_chans.pop()
            group = [Block("res", f"output_blocks.{idx}.0", ch + ich, mult * mc, tconv=tconv)]
            ch = mult * mc
            if ds in cfg["attention_resolutions"]:
                group += attn_blocks(f"output_blocks.{idx}", 1, ch)
            if level and i == cfg["num_res_blocks"]:
                group.append(Block("up", f"output_blocks.{idx}.{len(group)}", ch, ch))
                ds //= 2
            outputs.append(group)
    return cfg, inputs, middle, outputs


def _transformer_params(shapes, prefix, dim_in, heads, dim_head, depth, context_dim, linear_proj, conv1d, img_attn):
    inner = heads * dim_head
    shapes[f"{prefix}.norm.weight"] = (dim_in,)
    shapes[f"{prefix}.norm.bias"] = (dim_in,)
    if linear_proj:
        shapes[f"{prefix}.proj_in.weight"] = (inner, dim_in)
    else:
        shapes[f"{prefix}.proj_in.weight"] = (inner, dim_in, 1) if conv1d else (inner, dim_in, 1, 1)
    shapes[f"{prefix}.proj_in.bias"] = (inner,)
    for d in range(depth):
        p = f"{prefix}.transformer_blocks.{d}"
        for name, ctx in (("attn1", None), ("attn2", context_dim)):
            kv_in = inner if ctx is None else ctx
            shapes[f"{p}.{name}.to_q.weight"] = (inner, inner)
            shapes[f"{p}.{name}.to_k.weight"] = (inner, kv_in)
            shapes[f"{p}.{name}.to_v.weight"] = (inner, kv_in)
            shapes[f"{p}.{name}.to_out.0.weight"] = (inner, inner)
            shapes[f"{p}.{name}.to_out.0.bias"] = (inner,)
            if name == "attn2" and img_attn:
                shapes[f"{p}.{name}.to_k_ip.weight"] = (inner, kv_in)
                shapes[f"{p}.{name}.to_v_ip.weight"] = (inner, kv_in)
        shapes[f"{p}.ff.net.0.proj.weight"] = (inner * 8, inner)
        shapes[f"{p}.ff.net.0.proj.bias"] = (inner * 8,)
        shapes[f"{p}.ff.net.2.weight"] = (inner, inner * 4)
        shapes[f"{p}.ff.net.2.bias"] = (inner,)
        for n in ("norm1", "norm2", "norm3"):
            shapes[f"{p}.{n}.weight"] = (inner,)
            shapes[f"{p}.{n}.bias"] = (inner,)
    if linear_proj:
        shapes[f"{prefix}.proj_out.weight"] = (dim_in, inner)
    else:
        shapes[f"{prefix}.proj_out.weight"] = (dim_in, inner, 1) if conv1d else (dim_in, inner, 1, 1)
    shapes[f"{prefix}.proj_out.bias"] = (dim_in,)


def param_shapes(params):
    """OrderedDict key -> shape, equal (as a set of items) to UNetModel(**params).state_dict()."""
    cfg, inputs, middle, outputs = build_program(params)
    mc = cfg["model_channels"]
    ted = 4 * mc
    shapes = OrderedDict()
    for name in ("time_embed",) + (("fps_embedding",) if cfg["fps_cond"] else ()):
        shapes[f"{name}.0.weight"] = (ted, mc)
        shapes[f"{name}.0.bias"] = (ted,)
        shapes[f"{name}.2.weight"] = (ted, ted)
        shapes[f"{name}.2.bias"] = (ted,)
    if cfg["time_cond_proj_dim"] is not None:
        shapes["time_cond_proj.weight"] = (mc, cfg["time_cond_proj_dim"])

    def add_block(b):
        p = b.prefix
        if b.kind == "conv_in":
            shapes[f"{p}.weight"] = (b.cout, b.cin, 3, 3)
            shapes[f"{p}.bias"] = (b.cout,)
        elif b.kind == "res":
            shapes[f"{p}.in_layers.0.weight"] = (b.cin,)
            shapes[f"{p}.in_layers.0.bias"] = (b.cin,)
            shapes[f"{p}.in_layers.2.weight"] = (b.cout, b.cin, 3, 3)
            shapes[f"{p}.in_layers.2.bias"] = (b.cout,)
            shapes[f"{p}.emb_layers.1.weight"] = (b.cout, ted)
            shapes[f"{p}.emb_layers.1.bias"] = (b.cout,)
            shapes[f"{p}.out_layers.0.weight"] = (b.cout,)
            shapes[f"{p}.out_layers.0.bias"] = (b.cout,)
            shapes[f"{p}.out_layers.3.weight"] = (b.cout, b.cout, 3, 3)
            shapes[f"{p}.out_layers.3.bias"] = (b.cout,)
            if b.cin != b.cout:
                shapes[f"{p}.skip_connection.weight"] = (b.cout, b.cin, 1, 1)
                shapes[f"{p}.skip_connection.bias"] = (b.cout,)
            if b.tconv:
                for i in (1, 2, 3, 4):
                    conv_idx = 2 if i == 1 else 3
                    q = f"{p}.temopral_conv.conv{i}"
                    shapes[f"{q}.0.weight"] = (b.cout,)
                    shapes[f"{q}.0.bias"] = (b.cout,)
                    shapes[f"{q}.{conv_idx}.weight"] = (b.cout, b.cout, 3, 1, 1)
                    shapes[f"{q}.{conv_idx}.bias"] = (b.cout,)
        elif b.kind == "st":
            _transformer_params(shapes, p, b.cin, b.heads, b.dim_head, b.depth, cfg["context_dim"],
                                b.linear_proj, False, cfg["use_image_attention"])
        elif b.kind == "tt":
            _transformer_params(shapes, p, b.cin, b.heads, b.dim_head, b.depth, None,
                                b.linear_proj, True, False)
        elif b.kind == "down":
            shapes[f"{p}.op.weight"] = (b.cout, b.cin, 3, 3)
            shapes[f"{p}.op.bias"] = (b.cout,)
        elif b.kind == "up":
            shapes[f"{p}.conv.weight"] = (b.cout, b.cin, 3, 3)
            shapes[f"{p}.conv.bias"] = (b.cout,)

    for gi, group in enumerate(inputs):
        for b in group:
            add_block(b)
        if gi == 0 and cfg["addition_attention"]:
            # init_attn: TemporalTransformer(model_channels, n_heads=8, d_head=num_head_channels,
            # depth=transformer_depth), use_linear left at its default False (openaimodel3d.py:425-439)
            _transformer_params(shapes, "init_attn.0", mc, 8, cfg["num_head_channels"], cfg["transformer_depth"],
                                None, False, True, False)
    for b in middle:
        add_block(b)
    for group in outputs:
        for b in group:
            add_block(b)
    shapes["out.0.weight"] = (mc,)
    shapes["out.0.bias"] = (mc,)
    shapes["out.2.weight"] = (cfg["out_channels"], mc, 3, 3)
    shapes["out.2.bias"] = (cfg["out_channels"],)
    return shapes
