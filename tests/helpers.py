"""Shared test helpers (data only / deterministic stand-ins for out-of-scope producers)."""
import torch

from dynamicscaler_amd.synth import synth_normal


def synth_image_embedder(dim, tokens=16, seed=77):
    """Deterministic stand-in for get_image_embeds (CLIP image encoder + Resampler are out of scope):
    4x4 average pool of the crop projected 3 -> dim by a fixed seeded matrix.  Identical to the function
    tests/golden/make_golden.py used when it ran the reference."""
    proj = synth_normal((3, dim), seed)

    def embed(batch_imgs):
        pooled = torch.nn.functional.adaptive_avg_pool2d(batch_imgs.float().cpu(), (4, 4))
        return (pooled.flatten(2).transpose(1, 2) @ proj).to(batch_imgs.device)
    return embed


def i2v_geom(geom):
    """Golden geometry dict -> call kwargs: an `init_seed` entry stands for a synthetic init_panorama_latent (the same
    synth_normal draw make_golden.py handed to the reference)."""
    g = dict(geom)
    if "init_seed" in g:
        g["init_panorama_latent"] = synth_normal((1, 4, g["total_f"], g["total_h"] // 8, g["total_w"] // 8), g.pop("init_seed"))
    return g
