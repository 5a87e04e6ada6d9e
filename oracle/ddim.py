"""TEST INFRASTRUCTURE (oracle): DDIM schedule tables, ddim_step, re_noise, mask mix, CFG.

Restates, op for op and dtype for dtype (so fp32 results are bit-identical on CPU):
  * lvdm/models/utils_diffusion.py:31-35   make_beta_schedule("linear")
  * lvdm/models/ddpm3d.py:113-134          register_schedule (alphas_cumprod fp64 -> fp32 buffers)
  * lvdm/models/utils_diffusion.py:56-78   make_ddim_timesteps("uniform")
  * lvdm/models/utils_diffusion.py:100-112 make_ddim_sampling_parameters
  * pipeline/scheduler.py:18-57            lvdm_DDIM_Scheduler.make_schedule
  * pipeline/scheduler.py:60-96            ddim_step (per-frame loop; one scalar set per tile)
  * pipeline/scheduler.py:98-110           re_noise
  * utils/tensor_utils.py:19-39            mix_latents_with_mask
  * pipeline/t2v_sphere_panorama_pipeline.py:599  classifier-free guidance combine
"""
import numpy as np
import torch


class DiffusionTables:
    """The buffers of LatentDiffusion that the scheduler reads (ddpm3d.py:113-134)."""

    def __init__(self, timesteps=1000, linear_start=0.00085, linear_end=0.012):
        betas = (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, timesteps,
                                dtype=torch.float64) ** 2).numpy()
        alphas = 1.0 - betas
        alphas_cumprod = np.cumprod(alphas, axis=0)
        alphas_cumprod_prev = np.append(1.0, alphas_cumprod[:-1])
        self.num_timesteps = int(timesteps)
        self.betas = torch.tensor(betas, dtype=torch.float32)
        self.alphas_cumprod = torch.tensor(alphas_cumprod, dtype=torch.float32)
        self.alphas_cumprod_prev = torch.tensor(alphas_cumprod_prev, dtype=torch.float32)


class DDIMSchedule:
    """pipeline/scheduler.py:18-57 (only the members the panorama pipelines read)."""

    def __init__(self, tables: DiffusionTables, num_steps: int, eta: float = 0.0):
        n = tables.num_timesteps
        # utils_diffusion.py:56-66
        self.ddim_timesteps = np.linspace(0, n - 1, num_steps).round().copy().astype(np.int64)
        ac = tables.alphas_cumprod  # fp32 tensor
        self.alphas_cumprod = ac.clone()
        # utils_diffusion.py:100-112: alphas -> fp32 tensor, alphas_prev -> numpy fp64, sigmas -> fp64
        self.ddim_alphas = ac[self.ddim_timesteps]
        self.ddim_alphas_prev = np.asarray([ac[0]] + ac[self.ddim_timesteps[:-1]].tolist())
        self.ddim_sigmas = eta * np.sqrt((1 - self.ddim_alphas_prev) / (1 - self.ddim_alphas)
                                         * (1 - self.ddim_alphas / self.ddim_alphas_prev))
        self.ddim_sqrt_one_minus_alphas = np.sqrt(1.0 - self.ddim_alphas)
        self.eta = eta

    # -- scalar coefficients exactly as the reference materialises them (torch.full -> fp32) --
    def step_coefficients(self, index):
        """fp32 scalars used by ddim_step at schedule index `index` (scheduler.py:78-89)."""
        f32 = torch.float32
        a_t = torch.full((1,), float(self.ddim_alphas[index]), dtype=f32)
        a_prev = torch.full((1,), float(self.ddim_alphas_prev[index]), dtype=f32)
        sigma_t = torch.full((1,), float(self.ddim_sigmas[index]), dtype=f32)
        sqrt_one_minus_at = torch.full((1,), float(self.ddim_sqrt_one_minus_alphas[index]), dtype=f32)
        return {
            "sqrt_one_minus_at": float(sqrt_one_minus_at),
            "sqrt_at": float(a_t.sqrt()),
            "sqrt_a_prev": float(a_prev.sqrt()),
            "dir_coef": float((1.0 - a_prev - sigma_t ** 2).sqrt()),
            "sigma": float(sigma_t),
        }

    def renoise_coefficients(self, step_a, step_b):
        """fp32 (c, s) of re_noise (scheduler.py:99-105)."""
        a_a = self.alphas_cumprod[self.ddim_timesteps[step_a]]
        a_b = self.alphas_cumprod[self.ddim_timesteps[step_b]]
        c = torch.sqrt(a_b / a_a)
        s = torch.sqrt(1 - a_b / a_a)
        return float(c), float(s)


def ddim_step(sched: DDIMSchedule, sample, noise_pred, indices, noise=None):
    """scheduler.py:60-96.  sample/noise_pred: [b,c,f,h,w]; indices: one schedule index per frame.

    `noise`: optional [b,c,f,h,w] standing in for the per-frame torch.randn draws (:87);
    when None the global RNG is consumed exactly like the reference (f draws of [b,c,1,h,w])."""
    b = sample.shape[0]
    size = (b, 1, 1, 1, 1)
    x_prevs, pred_x0s = [], []
    for i, index in enumerate(indices):
        x = sample[:, :, [i]]
        e_t = noise_pred[:, :, [i]]
        a_t = torch.full(size, float(sched.ddim_alphas[index]))
        a_prev = torch.full(size, float(sched.ddim_alphas_prev[index]))
        sigma_t = torch.full(size, float(sched.ddim_sigmas[index]))
        sqrt_one_minus_at = torch.full(size, float(sched.ddim_sqrt_one_minus_alphas[index]))
        pred_x0 = (x - sqrt_one_minus_at * e_t) / a_t.sqrt()
        dir_xt = (1.0 - a_prev - sigma_t ** 2).sqrt() * e_t
        z = torch.randn(x.shape) if noise is None else noise[:, :, [i]]
        x_prev = a_prev.sqrt() * pred_x0 + dir_xt + sigma_t * z
        x_prevs.append(x_prev)
        pred_x0s.append(pred_x0)
    return torch.cat(x_prevs, dim=2), torch.cat(pred_x0s, dim=2)


def re_noise(sched: DDIMSchedule, x_a, step_a, step_b, noise=None):
    """scheduler.py:98-110.  `noise` stands in for torch.randn_like(x_a) (:106)."""
    a_a = sched.alphas_cumprod[sched.ddim_timesteps[step_a]]
    a_b = sched.alphas_cumprod[sched.ddim_timesteps[step_b]]
    c = torch.sqrt(a_b / a_a)
    s = torch.sqrt(1 - a_b / a_a)
    eps = torch.randn_like(x_a) if noise is None else noise
    return c * x_a + s * eps


def mix_latents_with_mask(latent_1, latent_to_add, mask, mix_ratio):
    """utils/tensor_utils.py:19-39 (same op order, so fp32 results are bit-identical)."""
    if mask.dim() == 3:
        m = mask.unsqueeze(0).unsqueeze(0).repeat(latent_1.size(0), latent_1.size(1),
                                                  latent_1.size(2), 1, 1)
    elif mask.dim() == 5:
        m = mask
    else:
        raise NotImplementedError("mask must be [1,H,W] or 5-D")
    w1 = latent_1 * (1 - mix_ratio)
    w2 = latent_to_add * mix_ratio
    mixed = w1 + w2
    return latent_1 * (1 - m) + mixed * m


def cfg_combine(eps_cond, eps_uncond, guidance_scale):
    """t2v_sphere_panorama_pipeline.py:599."""
    return eps_uncond + guidance_scale * (eps_cond - eps_uncond)
