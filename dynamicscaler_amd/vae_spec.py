"""Structure of the first stage (AutoencoderKL: lvdm/models/autoencoder.py:97-107 -> lvdm/modules/networks/ae_modules.py
Encoder :364-464, Decoder :466-579): the lists of blocks in execution order and the shapes of the parameters under the
reference's state-dict keys.  Shared by the HIP product (vae.py) and checked against the
reference's own state dict when the goldens are made (tests/golden/make_golden.py g14)."""


def decoder_blocks(dd):
    """[(kind, prefix, cin, cout)] in forward order.  kinds: conv_in, res, attn, up, norm_out, conv_out."""
    ch, ch_mult, nrb = dd["ch"], list(dd["ch_mult"]), dd["num_res_blocks"]
    nres = len(ch_mult)
    block_in = ch * ch_mult[nres - 1]
    curr_res = dd["resolution"] // 2 ** (nres - 1)
    out = [("conv_in", "decoder.conv_in", dd["z_channels"], block_in),
           ("res", "decoder.mid.block_1", block_in, block_in),
           ("attn", "decoder.mid.attn_1", block_in, block_in),
           ("res", "decoder.mid.block_2", block_in, block_in)]
    for i_level in reversed(range(nres)):
        block_out = ch * ch_mult[i_level]
        for i_block in range(nrb + 1):
            out.append(("res", f"decoder.up.{i_level}.block.{i_block}", block_in, block_out))
            block_in = block_out
            if curr_res in dd.get("attn_resolutions", []):
                out.append(("attn", f"decoder.up.{i_level}.attn.{i_block}", block_in, block_in))
        if i_level != 0:
            out.append(("up", f"decoder.up.{i_level}.upsample", block_in, block_in))
            curr_res *= 2
    out.append(("norm_out", "decoder.norm_out", block_in, block_in))
    out.append(("conv_out", "decoder.conv_out", block_in, dd["out_ch"]))
    return out


def decoder_param_shapes(dd, embed_dim):
    """key -> shape for post_quant_conv + decoder.* (the keys AutoencoderKL.state_dict() has for the decode path)."""
    s = {"post_quant_conv.weight": (dd["z_channels"], embed_dim, 1, 1), "post_quant_conv.bias": (dd["z_channels"],)}
    for kind, p, cin, cout in decoder_blocks(dd):
        if kind in ("conv_in", "conv_out"):
            s[p + ".weight"], s[p + ".bias"] = (cout, cin, 3, 3), (cout,)
        elif kind == "res":
            s[p + ".norm1.weight"], s[p + ".norm1.bias"] = (cin,), (cin,)
            s[p + ".conv1.weight"], s[p + ".conv1.bias"] = (cout, cin, 3, 3), (cout,)
            s[p + ".norm2.weight"], s[p + ".norm2.bias"] = (cout,), (cout,)
            s[p + ".conv2.weight"], s[p + ".conv2.bias"] = (cout, cout, 3, 3), (cout,)
            if cin != cout:
                s[p + ".nin_shortcut.weight"], s[p + ".nin_shortcut.bias"] = (cout, cin, 1, 1), (cout,)
        elif kind == "attn":
            s[p + ".norm.weight"], s[p + ".norm.bias"] = (cin,), (cin,)
            for n in ("q", "k", "v", "proj_out"):
                s[f"{p}.{n}.weight"], s[f"{p}.{n}.bias"] = (cin, cin, 1, 1), (cin,)
        elif kind == "up":
            s[p + ".conv.weight"], s[p + ".conv.bias"] = (cin, cin, 3, 3), (cin,)
        elif kind == "norm_out":
            s[p + ".weight"], s[p + ".bias"] = (cin,), (cin,)
    return s


def encoder_blocks(dd):
    """[(kind, prefix, cin, cout)] of Encoder.forward (ae_modules.py:442-464).  kinds: conv_in, res, attn, down,
    norm_out, conv_out."""
    ch, ch_mult, nrb = dd["ch"], list(dd["ch_mult"]), dd["num_res_blocks"]
    nres = len(ch_mult)
    in_ch_mult = [1] + ch_mult
    curr_res = dd["resolution"]
    out = [("conv_in", "encoder.conv_in", dd["in_channels"], ch)]
    block_in = ch
    for i_level in range(nres):
        block_in = ch * in_ch_mult[i_level]
        block_out = ch * ch_mult[i_level]
        for i_block in range(nrb):
            out.append(("res", f"encoder.down.{i_level}.block.{i_block}", block_in, block_out))
            block_in = block_out
            if curr_res in dd.get("attn_resolutions", []):
                out.append(("attn", f"encoder.down.{i_level}.attn.{i_block}", block_in, block_in))
        if i_level != nres - 1:
            out.append(("down", f"encoder.down.{i_level}.downsample", block_in, block_in))
            curr_res //= 2
    out += [("res", "encoder.mid.block_1", block_in, block_in), ("attn", "encoder.mid.attn_1", block_in, block_in),
            ("res", "encoder.mid.block_2", block_in, block_in), ("norm_out", "encoder.norm_out", block_in, block_in),
            ("conv_out", "encoder.conv_out", block_in, 2 * dd["z_channels"] if dd.get("double_z", True) else dd["z_channels"])]
    return out


def encoder_param_shapes(dd, embed_dim):
    """key -> shape for encoder.* + quant_conv (the encode path of AutoencoderKL.state_dict())."""
    s = {"quant_conv.weight": (2 * embed_dim, 2 * dd["z_channels"], 1, 1), "quant_conv.bias": (2 * embed_dim,)}
    for kind, p, cin, cout in encoder_blocks(dd):
        if kind in ("conv_in", "conv_out"):
            s[p + ".weight"], s[p + ".bias"] = (cout, cin, 3, 3), (cout,)
        elif kind == "res":
            s[p + ".norm1.weight"], s[p + ".norm1.bias"] = (cin,), (cin,)
            s[p + ".conv1.weight"], s[p + ".conv1.bias"] = (cout, cin, 3, 3), (cout,)
            s[p + ".norm2.weight"], s[p + ".norm2.bias"] = (cout,), (cout,)
            s[p + ".conv2.weight"], s[p + ".conv2.bias"] = (cout, cout, 3, 3), (cout,)
            if cin != cout:
                s[p + ".nin_shortcut.weight"], s[p + ".nin_shortcut.bias"] = (cout, cin, 1, 1), (cout,)
        elif kind == "attn":
            s[p + ".norm.weight"], s[p + ".norm.bias"] = (cin,), (cin,)
            for n in ("q", "k", "v", "proj_out"):
                s[f"{p}.{n}.weight"], s[f"{p}.{n}.bias"] = (cin, cin, 1, 1), (cin,)
        elif kind == "down":
            s[p + ".conv.weight"], s[p + ".conv.bias"] = (cin, cin, 3, 3), (cin,)
        elif kind == "norm_out":
            s[p + ".weight"], s[p + ".bias"] = (cin,), (cin,)
    return s


def vae_param_shapes(dd, embed_dim):
    s = dict(encoder_param_shapes(dd, embed_dim))
    s.update(decoder_param_shapes(dd, embed_dim))
    return s
