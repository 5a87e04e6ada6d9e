"""Host-side stand-in for the LatentDiffusion object the pipelines read (`pretrained_t2v`).

Only the members on the hot path exist: `.model` (DiffusionWrapper around the HIP UNetModel), the schedule
buffers of register_schedule (lvdm/models/ddpm3d.py:113-134), `get_learned_conditioning`, the plain attributes the
pipelines / scheduler read, and -- when the configs are given -- the first stage (`first_stage_config`:
`decode_first_stage_2DAE` / `encode_first_stage_2DAE`, ddpm3d.py:485-490, 556-562, on vae.AutoencoderKL) and the
conditioning producers of SURVEY.md 8-f N3 (`cond_stage_config`: the OpenCLIP text tower behind
`get_learned_conditioning`, ddpm3d.py:446-456; `cond_img_config` + `finegrained`: image tower + Resampler behind
`get_image_embeds`, ddpm3d.py:659-693).  Without a `cond_stage_config` the conditioner is any callable
prompt-list -> [1, L, context_dim] tensor.
"""
import importlib

import torch
import torch.nn as nn

from .scheduler import DiffusionTables
from .unet import UNetModel, DiffusionWrapper


def get_obj_from_str(string):
    module, cls = string.rsplit(".", 1)
    return getattr(importlib.import_module(module, package=None), cls)


def instantiate_from_config(config):
    """utils/utils.py:56-71."""
    if "target" not in config:
        raise KeyError("Expected key `target` to instantiate.")
    return get_obj_from_str(config["target"])(**config.get("params", dict()))


def load_model_checkpoint(model, ckpt):
    """scripts/evaluation/funcs.py:88-104: the three checkpoint layouts the reference accepts -- a DeepSpeed dump
    ({'module': {'_forward_module.<key>': ...}}: the first 16 characters of every key are stripped), a Lightning checkpoint
    ({'state_dict': {...}}) or a bare state dict.  `ckpt` is a path (torch.load, map_location='cpu') or an already loaded
    mapping.  Loading is strict for everything this build holds parameters for (UNet, first stage, conditioning encoders);
    the reference model's DDPM schedule buffers and EMA copies, which this build recomputes, are the only keys ignored."""
    sd = torch.load(ckpt, map_location="cpu") if isinstance(ckpt, (str, bytes)) or hasattr(ckpt, "__fspath__") else ckpt
    if "module" in sd and isinstance(sd["module"], dict):
        sd = {k[16:]: v for k, v in sd["module"].items()}
    elif "state_dict" in sd:
        sd = sd["state_dict"]
    own = model.state_dict()
    recomputed = ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_", "log_one_minus", "posterior_", "logvar",
                  "lvlb_weights", "scale_arr", "model_ema.")
    use = {k: v for k, v in sd.items() if k in own}
    unexpected = [k for k in sd if k not in own and not k.startswith(recomputed)]
    missing = [k for k in own if k not in sd and not k.startswith(recomputed)]
    if unexpected or missing:
        raise RuntimeError(f"load_model_checkpoint: {len(missing)} missing key(s) {missing[:5]}, "
                           f"{len(unexpected)} unexpected key(s) {unexpected[:5]}")
    model.load_state_dict(use, strict=False)
    return model


class SyntheticConditioner:
    """prompt -> seeded N(0,1) embedding [1, L, dim] (fp16-representable), "" -> a different seed."""

    def __init__(self, length=77, dim=1024, cond_seed=1, uncond_seed=2):
        from .synth import synth_normal
        self.cond = synth_normal((1, length, dim), cond_seed)
        self.uncond = synth_normal((1, length, dim), uncond_seed)

    def __call__(self, prompts):
        return self.uncond if prompts[0] == "" else self.cond


class LatentDiffusionHost(nn.Module):
    def __init__(self, unet_config, timesteps=1000, linear_start=0.00085, linear_end=0.012, uncond_type="empty_seq",
                 use_scale=False, channels=4, conditioner=None, first_stage_config=None, scale_factor=1.0,
                 cond_stage_config=None, cond_img_config=None, finegrained=False, first_stage_operands="f16", **_ignored):
        super().__init__()
        params = unet_config["params"] if "params" in unet_config else unet_config
        self.model = DiffusionWrapper(UNetModel(**params), "crossattn")
        tables = DiffusionTables(timesteps, linear_start, linear_end)
        self.register_buffer("betas", tables.betas)
        self.register_buffer("alphas_cumprod", tables.alphas_cumprod)
        self.register_buffer("alphas_cumprod_prev", tables.alphas_cumprod_prev)
        self.num_timesteps = tables.num_timesteps
        self.use_scale = use_scale
        self.uncond_type = uncond_type
        self.channels = channels
        self.temporal_length = params.get("temporal_length", 16)
        self.scale_factor = scale_factor
        self.first_stage_model = None
        if first_stage_config is not None:
            from .vae import AutoencoderKL
            fp = first_stage_config.get("params", first_stage_config)
            self.first_stage_model = AutoencoderKL(fp["ddconfig"], fp.get("embed_dim", 4))
            # "wide": the first stage on fp32 activations and split-fp16 products (vae.py; ~4x the decode time, inside 1e-3)
            self.first_stage_model.operand_mode = first_stage_operands
        self.cond_stage_model = None
        if cond_stage_config is not None:                                    # ddpm3d.py:427-444 (frozen, eval)
            self.cond_stage_model = self._instantiate_encoder(cond_stage_config)
        if cond_img_config is not None:                                      # LatentVisualDiffusion, ddpm3d.py:659-686
            from .encoders import Resampler
            self.embedder = self._instantiate_encoder(cond_img_config)
            if not finegrained:
                raise NotImplementedError("ImageProjModel (finegrained=False) is not used by the reference's i2v configs")
            self.image_proj_model = Resampler(dim=1024, depth=4, dim_head=64, heads=12, num_queries=16,
                                              embedding_dim=1280, output_dim=1024, ff_mult=4)
        self.conditioner = conditioner

    @staticmethod
    def _instantiate_encoder(config):
        """yaml `target:` strings of the reference (lvdm.modules.encoders.condition.*) resolve to encoders.py."""
        from . import encoders
        name = config["target"].rsplit(".", 1)[1]
        if not hasattr(encoders, name):
            raise NotImplementedError(f"conditioning encoder {config['target']} is not built")
        return getattr(encoders, name)(**config.get("params", dict()))

    @property
    def device(self):
        return self.betas.device

    @torch.no_grad()
    def decode_first_stage_2DAE(self, z, **kwargs):
        """ddpm3d.py:556-562: z [B,C,T,h,w] -> [B,3,T,H,W], every frame through AutoencoderKL.decode of z / scale_factor."""
        if self.first_stage_model is None:
            raise RuntimeError("no first-stage decoder attached (pass first_stage_config, or use output_type='latent')")
        return self.first_stage_model.decode_frames(z.to(self.device), in_scale=1.0 / self.scale_factor)

    @torch.no_grad()
    def encode_first_stage_2DAE(self, x):
        """ddpm3d.py:485-490: x [B,3,T,H,W] -> scale_factor * posterior.sample(), frame by frame; the posterior noise is
        drawn on the host in the reference's order (torch.randn(mean.shape) per frame, lvdm/distributions.py:35-39)."""
        if self.first_stage_model is None:
            raise RuntimeError("no first-stage model attached (pass first_stage_config)")
        x = x.to(self.device)
        mom, (h, w) = self.first_stage_model.encode_moments(x)
        B, _, T = x.shape[:3]
        C = mom.shape[1] // 2
        noise = torch.stack([torch.randn((B, C, h, w)) for _ in range(T)], dim=2).to(self.device)
        from . import ops
        return ops.posterior_sample(mom, (B, C, T, h, w), noise, self.scale_factor)

    def get_learned_conditioning(self, prompts):
        """ddpm3d.py:446-456."""
        if self.cond_stage_model is not None:
            return self.cond_stage_model.encode(prompts)
        if self.conditioner is None:
            raise RuntimeError("no conditioner attached: pass cond_stage_config (the OpenCLIP text tower, token ids in) "
                               "or a callable prompts -> [1, L, context_dim] tensor")
        return self.conditioner(prompts)

    def get_image_embeds(self, batch_imgs):
        """ddpm3d.py:689-693: batch_imgs [b,3,H,W] in [-1,1] -> image tower tokens -> Resampler -> [b,16,1024]."""
        if getattr(self, "embedder", None) is None:
            raise RuntimeError("no image embedder attached (pass cond_img_config and finegrained=True)")
        return self.image_proj_model(self.embedder(batch_imgs.to(self.device)))
