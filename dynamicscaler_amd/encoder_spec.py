"""State-dict key -> shape tables of the conditioning producers, on the keys the reference's
checkpoints carry:

  Resampler               lvdm/modules/encoders/ip_resampler.py:93-135 (built by ddpm3d.py:683-685: dim 1024, depth 4,
                          dim_head 64, heads 12, 16 queries, embedding_dim 1280, output_dim 1024)
  OpenCLIP text tower     the `model.*` keys of FrozenOpenCLIPEmbedder (condition.py:186-189: open_clip ViT-H-14 with
                          `visual` deleted)
  OpenCLIP image tower    the `model.visual.*` keys of FrozenOpenCLIPImageEmbedderV2 (condition.py:306-309: `transformer`
                          deleted)

open_clip_torch (requirements.txt:23, 2.22.0) is not in this image; the key names and shapes below are those of its
CLIP / VisionTransformer / Transformer / ResidualAttentionBlock modules (nn.MultiheadAttention packs q,k,v as
`attn.in_proj_weight` [3W, W]).
"""

RESAMPLER_I2V = dict(dim=1024, depth=4, dim_head=64, heads=12, num_queries=16, embedding_dim=1280, output_dim=1024,
                     ff_mult=4)                                                   # ddpm3d.py:666-667, 683-685
CLIP_VIT_H_14 = dict(embed_dim=1024,
                     vision=dict(image_size=224, layers=32, width=1280, head_width=80, patch_size=14, mlp_ratio=4.0),
                     text=dict(context_length=77, vocab_size=49408, width=1024, heads=16, layers=24, mlp_ratio=4.0))
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)                                   # condition.py:319-320
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def resampler_param_shapes(dim=1024, depth=8, dim_head=64, heads=16, num_queries=8, embedding_dim=768, output_dim=1024,
                           ff_mult=4):
    inner = dim_head * heads
    s = {"latents": (1, num_queries, dim),
         "proj_in.weight": (dim, embedding_dim), "proj_in.bias": (dim,),
         "proj_out.weight": (output_dim, dim), "proj_out.bias": (output_dim,),
         "norm_out.weight": (output_dim,), "norm_out.bias": (output_dim,)}
    for i in range(depth):
        a, f = f"layers.{i}.0", f"layers.{i}.1"
        s.update({f"{a}.norm1.weight": (dim,), f"{a}.norm1.bias": (dim,), f"{a}.norm2.weight": (dim,),
                  f"{a}.norm2.bias": (dim,), f"{a}.to_q.weight": (inner, dim), f"{a}.to_kv.weight": (2 * inner, dim),
                  f"{a}.to_out.weight": (dim, inner),
                  f"{f}.0.weight": (dim,), f"{f}.0.bias": (dim,), f"{f}.1.weight": (int(dim * ff_mult), dim),
                  f"{f}.3.weight": (dim, int(dim * ff_mult))})
    return s


def _resblock_shapes(prefix, width, mlp_width):
    p = prefix
    return {f"{p}.ln_1.weight": (width,), f"{p}.ln_1.bias": (width,),
            f"{p}.attn.in_proj_weight": (3 * width, width), f"{p}.attn.in_proj_bias": (3 * width,),
            f"{p}.attn.out_proj.weight": (width, width), f"{p}.attn.out_proj.bias": (width,),
            f"{p}.ln_2.weight": (width,), f"{p}.ln_2.bias": (width,),
            f"{p}.mlp.c_fc.weight": (mlp_width, width), f"{p}.mlp.c_fc.bias": (mlp_width,),
            f"{p}.mlp.c_proj.weight": (width, mlp_width), f"{p}.mlp.c_proj.bias": (width,)}


def clip_text_param_shapes(text, prefix="model."):
    """Keys the text tower reads (text_projection / logit_scale of the checkpoint are not used by the embedder)."""
    W = text["width"]
    s = {f"{prefix}token_embedding.weight": (text["vocab_size"], W),
         f"{prefix}positional_embedding": (text["context_length"], W),
         f"{prefix}ln_final.weight": (W,), f"{prefix}ln_final.bias": (W,)}
    for i in range(text["layers"]):
        s.update(_resblock_shapes(f"{prefix}transformer.resblocks.{i}", W, int(W * text["mlp_ratio"])))
    return s


def clip_vision_param_shapes(vision, prefix="model.visual."):
    """Keys the image tower reads (ln_post / proj are skipped by encode_with_vision_transformer, condition.py:336-365)."""
    W, P = vision["width"], vision["patch_size"]
    g = vision["image_size"] // P
    s = {f"{prefix}conv1.weight": (W, 3, P, P), f"{prefix}class_embedding": (W,),
         f"{prefix}positional_embedding": (g * g + 1, W),
         f"{prefix}ln_pre.weight": (W,), f"{prefix}ln_pre.bias": (W,)}
    for i in range(vision["layers"]):
        s.update(_resblock_shapes(f"{prefix}transformer.resblocks.{i}", W, int(W * vision["mlp_ratio"])))
    return s
