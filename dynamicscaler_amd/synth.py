"""Deterministic synthetic inputs (weights, latents, contexts) for tests and bench.py.

No checkpoints, CLIP or VAE exist in the build / GPU environments, so weights are
generated per state-dict key from (seed, crc32(key)) with torch's CPU generator: the same tensors
on every machine without shipping 5.6 GB.  All values are rounded to fp16-representable numbers
so the fp32 CPU oracle and the fp16 HIP path start from identical parameters (parity).
"""
import math
import zlib

import torch


def _gen(seed, key):
    g = torch.Generator(device="cpu")
    g.manual_seed(((int(seed) & 0x7FFFFFFF) << 32) | (zlib.crc32(key.encode()) & 0xFFFFFFFF))
    return g


def synth_tensor(key, shape, shapes, seed=0):
    g = _gen(seed, key)
    shape = tuple(shape)
    if key.endswith(".weight") and len(shape) >= 2:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        bound = 1.0 / math.sqrt(fan_in)
        t = (torch.rand(shape, generator=g) * 2 - 1) * bound
    elif key.endswith(".weight"):  # norm scale
        t = 1.0 + 0.1 * torch.randn(shape, generator=g)
    else:  # bias
        wkey = key[: -len("bias")] + "weight"
        wshape = shapes.get(wkey, None)
        if wshape is not None and len(wshape) >= 2:
            fan_in = 1
            for s in wshape[1:]:
                fan_in *= s
            t = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(fan_in)
        else:  # norm shift
            t = 0.1 * torch.randn(shape, generator=g)
    return t.half().float()


def synth_state_dict(shapes, seed=0):
    """fp32 CPU tensors (fp16-representable) for every key of `shapes` (dict key -> shape)."""
    return {k: synth_tensor(k, s, shapes, seed) for k, s in shapes.items()}


def synth_normal(shape, seed, scale=1.0):
    """Seeded N(0, scale^2) tensor on CPU, fp16-representable fp32."""
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    return (torch.randn(tuple(shape), generator=g) * scale).half().float()


def synth_encoder_state_dict(shapes, seed=0):
    """Synthetic weights for the CLIP towers / Resampler (encoder_spec.py): like synth_state_dict, with the keys that do
    not follow the `.weight` / `.bias` naming handled by role -- packed attention projections as Linear weights,
    embeddings / queries at CLIP-like scale."""
    out = {}
    for k, shape in shapes.items():
        g = _gen(seed, k)
        shape = tuple(shape)
        if k.endswith("in_proj_weight"):
            t = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(shape[1])
        elif k.endswith("in_proj_bias"):
            t = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(shape[0] // 3)
        elif k.endswith(("positional_embedding", "class_embedding")):
            t = 0.05 * torch.randn(shape, generator=g)
        elif k.endswith("latents"):
            t = torch.randn(shape, generator=g) / math.sqrt(shape[-1])
        else:
            out[k] = synth_tensor(k, shape, shapes, seed)
            continue
        out[k] = t.half().float()
    return out
