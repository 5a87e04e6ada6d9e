"""DDIM scheduler host side (drop-in for pipeline/scheduler.py:lvdm_DDIM_Scheduler) and the diffusion tables.

The tables are a few dozen scalars computed once on the host with the reference's dtypes (fp64 betas ->
fp32 alphas_cumprod buffers, numpy fp64 alphas_prev ...), so the per-tile coefficients handed to the HIP
kernels are the very fp32 numbers the reference materialises with torch.full (scheduler.py:78-85).
The per-element work (ddim_step, re_noise) runs in ds_cfg_ddim / ds_renoise_mix.
"""
import numpy as np
import torch

from . import ops


def make_beta_schedule_linear(n_timestep, linear_start, linear_end):
    """lvdm/models/utils_diffusion.py:31-35."""
    return (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=torch.float64) ** 2).numpy()


class DiffusionTables:
    """Buffers LatentDiffusion registers in register_schedule (lvdm/models/ddpm3d.py:113-134)."""

    def __init__(self, timesteps=1000, linear_start=0.00085, linear_end=0.012, device="cpu"):
        betas = make_beta_schedule_linear(timesteps, linear_start, linear_end)
        alphas_cumprod = np.cumprod(1.0 - betas, axis=0)
        alphas_cumprod_prev = np.append(1.0, alphas_cumprod[:-1])
        self.num_timesteps = int(timesteps)
        self.betas = torch.tensor(betas, dtype=torch.float32)
        self.alphas_cumprod = torch.tensor(alphas_cumprod, dtype=torch.float32)
        self.alphas_cumprod_prev = torch.tensor(alphas_cumprod_prev, dtype=torch.float32)
        self.use_scale = False
        self.device = torch.device(device)


class lvdm_DDIM_Scheduler(object):
    """Same constructor / attributes / methods as the reference class.

    rng_mode:
      "reference" (default) -- every torch.randn / randn_like of the reference is drawn on the HOST from the global
                  torch CPU generator in the reference's order and uploaded: bit-comparable runs.
      "device"    -- re_noise draws Philox normals inside ds_renoise_mix, ddim_step draws nothing when sigma == 0:
                  fastest, statistically equivalent, not bit-comparable with a torch CPU run.
    """

    def __init__(self, model, schedule="linear", rng_mode="reference", **kwargs):
        self.model = model
        self.ddpm_num_timesteps = model.num_timesteps
        self.schedule = schedule
        self.counter = 0
        self.rng_mode = rng_mode
        self._philox_offset = 0
        self.philox_seed = 0                 # settable: Philox key of the in-kernel draws (rng_mode "device")

    # Philox counter spaces (rng_mode "device").  Tile draws of the step loops: key = philox_seed, counter =
    # tile_philox_offset(step, tile, numel) -- a fixed stride of PHILOX_TILES_PER_STEP tiles is reserved per step, so the
    # counters of two steps never overlap whatever their window counts (docking adds windows on some steps only).
    # Panorama-sized draws of add_noise / re_noise: a different KEY (aux_philox_seed), so they can never meet a tile's.
    PHILOX_TILES_PER_STEP = 1 << 16

    @classmethod
    def tile_philox_offset(cls, step, numel):
        return int(step) * cls.PHILOX_TILES_PER_STEP * int(numel)

    @property
    def aux_philox_seed(self):
        return (int(self.philox_seed) ^ 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF

    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0., verbose=True):
        if ddim_discretize != "uniform":
            raise NotImplementedError("only the 'uniform' discretisation is used by DynamicScaler")
        n = self.ddpm_num_timesteps
        # lvdm/models/utils_diffusion.py:56-66
        self.ddim_timesteps = np.linspace(0, n - 1, ddim_num_steps).round().copy().astype(np.int64)
        ac = self.model.alphas_cumprod.detach().to("cpu", torch.float32)
        assert ac.shape[0] == n, "alphas have to be defined for each timestep"
        self.alphas_cumprod = ac.clone()
        self.betas = self.model.betas.detach().to("cpu", torch.float32).clone()
        self.alphas_cumprod_prev = self.model.alphas_cumprod_prev.detach().to("cpu", torch.float32).clone()
        self.use_scale = getattr(self.model, "use_scale", False)
        # lvdm/models/utils_diffusion.py:100-112 (dtypes: fp32 tensor, numpy fp64, fp64)
        self.ddim_alphas = ac[self.ddim_timesteps]
        self.ddim_alphas_prev = np.asarray([ac[0]] + ac[self.ddim_timesteps[:-1]].tolist())
        self.ddim_sigmas = ddim_eta * np.sqrt((1 - self.ddim_alphas_prev) / (1 - self.ddim_alphas)
                                              * (1 - self.ddim_alphas / self.ddim_alphas_prev))
        # numpy's sqrt, like the reference (scheduler.py:52): torch's vectorised CPU sqrt differs by 1 ulp on some entries
        self.ddim_sqrt_one_minus_alphas = torch.from_numpy(np.sqrt((1.0 - self.ddim_alphas).numpy()))
        self.ddim_eta = ddim_eta
        if verbose:
            print(f"Selected timesteps for ddim sampler: {self.ddim_timesteps}")

    # ---- scalar coefficients in the reference's fp32 ----
    def step_coefficients(self, index):
        f32 = torch.float32
        a_t = torch.full((1,), float(self.ddim_alphas[index]), dtype=f32)
        a_prev = torch.full((1,), float(self.ddim_alphas_prev[index]), dtype=f32)
        sigma_t = torch.full((1,), float(self.ddim_sigmas[index]), dtype=f32)
        sq1m = torch.full((1,), float(self.ddim_sqrt_one_minus_alphas[index]), dtype=f32)
        return {"sqrt_one_minus_at": float(sq1m), "sqrt_at": float(a_t.sqrt()), "sqrt_a_prev": float(a_prev.sqrt()),
                "dir_coef": float((1.0 - a_prev - sigma_t ** 2).sqrt()), "sigma": float(sigma_t)}

    def eps_amplification(self, index, relative=True):
        """How much the DDIM update at schedule index `index` multiplies an error of the guided eps (pipeline/scheduler.py:83-89, sigma
        folded into the direction term):  x_prev = (sqrt_a_prev / sqrt_at) x + (dir_coef - sqrt_a_prev sqrt(1 - a_t) / sqrt_at) e_t.
        relative=False: |d x_prev / d e_t| (3.79 for config 1's first update 999 -> 666, 0.127 on the 50-step schedule's first);
        relative=True: that over the growth sqrt_a_prev / sqrt_at of the x term -- the RELATIVE error of x_prev per unit relative
        error of e_t for |e_t| ~ |x| (unit-scale latents): 0.61 and 0.113 for the same two updates.  The pipelines' operand policy
        is written in terms of it."""
        c = self.step_coefficients(index)
        grow = c["sqrt_a_prev"] / c["sqrt_at"]
        amp = abs(c["dir_coef"] - grow * c["sqrt_one_minus_at"])
        return amp / grow if relative else amp

    def renoise_coefficients(self, step_a, step_b):
        a_a = self.alphas_cumprod[self.ddim_timesteps[step_a]]
        a_b = self.alphas_cumprod[self.ddim_timesteps[step_b]]
        return float(torch.sqrt(a_b / a_a)), float(torch.sqrt(1 - a_b / a_a))

    # ---- host-side noise in reference order ----
    def draw_step_noise(self, shape, device, dtype, sigma):
        """The f per-frame torch.randn draws of ddim_step (scheduler.py:87).  Always consumed in 'reference' mode
        (the stream must advance even when sigma == 0); uploaded only when sigma != 0."""
        if self.rng_mode != "reference":
            if sigma != 0.0:
                return torch.randn(shape, device=device, dtype=torch.float32).to(dtype)
            return None
        b, c, f, h, w = shape
        frames = [torch.randn((b, c, 1, h, w)) for _ in range(f)]
        if sigma == 0.0:
            return None
        return torch.cat(frames, dim=2).to(device=device, dtype=dtype)

    def draw_renoise_noise(self, shape, device, dtype, sphere_view=None):
        """torch.randn_like(x_a) of re_noise (scheduler.py:106) drawn on the host, or None in 'device' mode.

        sphere_view: None for the ring loops (their windows are contiguous clones).  In the t2v SPHERE loop the
        tensor handed to re_noise is a view of PanoramaLatentProxy's storage and its STRIDES decide which of torch's CPU
        normal paths runs: the first view of a run is contiguous (plain randn); after the first
        set_view_tensor_no_interpolation the storage becomes [B,N,C,H,W]-contiguous (panorama_tensor_utils.py:183), every
        later view is a non-contiguous permute, randn_like keeps the strides and the scalar normal path produces a
        different stream.  "later" reproduces that by calling randn_like on an identically strided tensor."""
        if self.rng_mode != "reference":
            return None
        if sphere_view == "later":
            b, c, n, h, w = shape
            return torch.randn_like(torch.empty((b, n, c, h, w)).permute(0, 2, 1, 3, 4)).contiguous().to(device=device, dtype=dtype)
        return torch.randn(shape).to(device=device, dtype=dtype)

    def next_philox_offset(self, count):
        off = self._philox_offset
        self._philox_offset += int(count)
        return off

    # ---- drop-in tensor methods ----
    @torch.no_grad()
    def ddim_step(self, sample, noise_pred, indices):
        """scheduler.py:60-96 for the only call pattern of the pipelines: indices == [index] * frames."""
        index = indices[0]
        if any(i != index for i in indices):
            raise NotImplementedError("per-frame schedule indices are never used by the panorama pipelines")
        coef = self.step_coefficients(index)
        noise = self.draw_step_noise(tuple(sample.shape), sample.device, sample.dtype, coef["sigma"])
        b = sample.shape[0]
        pano_shape = (1,) + tuple(sample.shape[1:])
        x_prev, x0 = ops.cfg_ddim(sample.contiguous(), noise_pred.contiguous(), None, pano_shape, 1.0, coef, noise)
        return x_prev, x0

    @torch.no_grad()
    def add_noise(self, clear, index, frames_major_strides=False):
        """VC2_Pipeline_T2V._add_noise (t2v_normal_pipeline.py:619-625): sqrt(a)*x + sqrt(1-a)*randn_like(x) with
        a = ddim_alphas[index] (fp32), through the same fused kernel as re_noise.  frames_major_strides: the reference's
        tensor is the output of resize_video_latent, a permuted view of [B,F,C,H,W] storage (diffusion_utils.py:30-31),
        and randn_like follows its strides (see draw_renoise_noise)."""
        alpha = torch.as_tensor(self.ddim_alphas[index], dtype=torch.float32)
        c, s = float(alpha ** 0.5), float((1 - alpha) ** 0.5)
        x = clear.contiguous().clone()
        noise = self.draw_renoise_noise(tuple(x.shape), x.device, x.dtype, "later" if frames_major_strides else None)
        ones = torch.ones((x.shape[0],) + tuple(x.shape[2:]), dtype=torch.uint8, device=x.device)
        ops.renoise_mix_(x, ones, (1,) + tuple(x.shape[1:]), c, s, 1.0, noise=noise, mask_frame0=False,
                         seed=self.aux_philox_seed, offset=self.next_philox_offset(x.numel()))
        return x

    @torch.no_grad()
    def re_noise(self, x_a, step_a, step_b, frames_major_strides=False):
        """scheduler.py:98-110: x_b = c*x_a + s*randn_like(x_a) (mask of ones, ratio 1 in the fused kernel).
        frames_major_strides: x_a is (a clone of) a resize_video_latent output in the reference, see add_noise."""
        c, s = self.renoise_coefficients(step_a, step_b)
        x = x_a.contiguous().clone()
        n = x.shape[0]
        noise = self.draw_renoise_noise(tuple(x.shape), x.device, x.dtype, "later" if frames_major_strides else None)
        ones = torch.ones((n,) + tuple(x.shape[2:]), dtype=torch.uint8, device=x.device)
        pano_shape = (1,) + tuple(x.shape[1:])
        ops.renoise_mix_(x, ones, pano_shape, c, s, 1.0, noise=noise, mask_frame0=False, seed=self.aux_philox_seed,
                         offset=self.next_philox_offset(x.numel()))
        return x
