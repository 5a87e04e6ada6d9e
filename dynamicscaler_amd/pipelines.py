"""Panorama denoising pipelines (drop-in call surface of pipeline/t2v_normal_pipeline.py and
pipeline/t2v_sphere_panorama_pipeline.py), MI355X host side.

Same constructor `(pretrained_t2v, scheduler, model_config)`, same method names / keywords / return tuples
(SURVEY.md 8-b).  What differs is underneath: the panorama, its pred-x0 twin and a 1-byte mask live in HBM for
the whole loop; a DDIM step is cut into levels of independent windows (parallel.plan_levels) and each level is
  ds_ring_gather (latent + mask, all windows of the level in one launch) -> ds_renoise_mix ->
  ONE batched UNet evaluation for [cond | uncond] x windows -> ds_cfg_ddim -> ds_ring_scatter3
instead of ~300 tiny torch launches per tile.  With torch.distributed initialised, the step's windows are shared
out over the ranks -- whole panorama columns per rank with one all-gather per step, or a strided share of every level
(parallel.run_step).
"""
import os

import numpy as np
import torch

from . import ops, parallel
from .ring import RingLatent, VAE_SCALE_FACTOR, ring_axis_steps, t2v_ring_windows, t2v_grid_windows


def select_prompt_from_multi_prompt_dict_by_factor(prompt_dict, factor):
    """utils/multi_prompt_utils.py:1-7."""
    assert 0.0 <= factor <= 1.0, f"select_prompt: input factor {factor} not legal"
    keys = sorted(prompt_dict.keys())
    for key in keys:
        if factor <= key:
            return prompt_dict[key]
    return prompt_dict[keys[-1]]


# relative error of the CFG-combined eps (guidance 7.5) at t = 999 -- where it is largest -- per residual mode of the fp16-operand
# UNet, measured against the reference on MI355X (profiles/r4_measured_parity.jsonl, tests/test_gpu_fullsize.py)
GUIDED_EPS_ERR = {"f16": 1.0e-2, "f32outer": 7.4e-3, "f32": 5.6e-3}
# The table above was measured on seed-0 synthetic weights (t = 999, CFG 7.5).  Round 6: at set-up the pipelines MEASURE the same quantity
# on the caller's own UNet and contexts (one wide + one own-mode evaluation pair at the schedule's first timestep, the wide result as the
# fp32 reference) and use measured / OPERAND_SAFETY instead; the table is what an uncalibrated pipeline (no device yet) falls back to.
OPERAND_SAFETY = 0.8


def _noise_shape(t):
    """How the guided-eps error falls with the noise level, relative to t = 999 (measured, default mode: 7.4e-3 at 999, 5.0e-3 at 816,
    3.6e-3 at 612, 2.8e-3 below 200; within 5 %)."""
    return 0.375 + 0.625 * (float(t) / 999.0) ** 3.5


class _ProgressBar:
    def __init__(self, total):
        self.total, self.n = total, 0

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def update(self, n=1):
        self.n += n


class VC2_Pipeline_T2V:
    """pipeline/t2v_normal_pipeline.py:25-210 (diffusers.DiffusionPipeline is not needed: only register_modules,
    progress_bar, _execution_device and .to are used by the reference)."""

    def __init__(self, pretrained_t2v, scheduler, model_config=None):
        self.pretrained_t2v = pretrained_t2v
        self.scheduler = scheduler
        self.vae = getattr(pretrained_t2v, "first_stage_model", None)
        self.unet = pretrained_t2v.model.diffusion_model
        self.text_encoder = getattr(pretrained_t2v, "cond_stage_model", None)
        self.model_config = model_config
        self.vae_scale_factor = VAE_SCALE_FACTOR
        self._device = None
        # Storage type of the panorama latent and of the tiles between the UNet evaluations.  fp32 like the reference (default):
        # the panorama is 4-48 MB, so the bytes are irrelevant, while fp16 storage rounds the latent once per DDIM step and
        # those roundings random-walk over a 50-step run (measured, tests/test_gpu_schedule50.py: final pred_x0 1.75e-3 from
        # the reference with fp16 latents, 9.8e-4 with fp32 latents, the same kernels).  `.to(device, torch.float16)` selects
        # fp16 latents (BASELINE.json's wording); the matrix-core operands are fp16 either way.
        self.latent_dtype = torch.float32
        self.max_tile_batch = 8              # windows per batched UNet evaluation (x2 with CFG)
        self.num_streams = 1                 # > 1: tile batches of a level run concurrently on that many HIP streams
        self._pool = None
        self.use_graph = False               # hipGraph replay of the UNet evaluation (see _eps)
        self.share_cfg_prefix = True         # [cond | uncond] pairs: context-free UNet prefix evaluated once (UNetModel.forward)
        self._graphs = {}
        self._graph_generation = None        # UNetModel._generation the cached graphs were captured with
        self._slot = 0                       # stream slot of the tile batch being enqueued (one graph set per slot)
        # a level with a SINGLE tile batch (a rank's share on 8 GPUs: one tile per level; single-chain rings): cond and uncond
        # evaluations on two streams instead of one [cond | uncond] batch -- nothing else could overlap them, and two
        # half-size forwards in flight use the CUs the other's small launches leave idle (80.5 vs 82.5 ms per rank-step measured on
        # the 8-GPU share of cfg3, tools/exp/gpu_gn_sparse.sh, profiles/r3_notes.md sections 4 and 8; bit-identical, a batch equals its separate forwards).  1 (default):
        # only such levels; 0: never; 2: every batch, one after the other (emulates a rank's single-batch levels on one GPU).
        self.split_cfg_over_streams = int(os.environ.get("DS_SPLIT_CFG", "1"))
        # ... and tell the GEMM that two equal launch sequences share the chip then (ops.set_launch_share): each plans its persistent
        # big tiles on half of the CUs and hands the rows of a mostly empty last round to small tiles -- at a rank's size a level-1
        # launch is 160 tiles of 256 rows: two of them in flight used to take two rounds on 256 CUs for 1.25 rounds of work
        # OFF by default: measured on MI355X it LOSES 4 % of the rank-share step (170.5 vs 163.3 ms per two rank-steps, gpurun_out/r5c):
        # the two streams do not stay in lockstep, and a launch planned on 128 CUs leaves the rest idle whenever its partner is
        # between kernels (profiles/r5_notes.md section 4)
        self.share_launches = os.environ.get("DS_SHARE_LAUNCHES", "0") != "0"
        self._launch_share = 1
        # gather + re-noise and CFG + DDIM + scatter as one kernel each (ds_ring_gather_renoise / ds_cfg_ddim_scatter; bit-identical
        # to the separate kernels, 0 = those)
        self.fuse_tile_ops = os.environ.get("DS_FUSE_TILE_OPS", "1") != "0"
        # Which DDIM steps evaluate the UNet in the WIDE operand mode (fp32 storage, split-fp16 products: UNetModel.forward(...,
        # precision="wide"), csrc/wide.hip) instead of the model's own mode.  With single fp16 matrix-core operands eps sits ~1e-3
        # from the reference's fp32 result, classifier-free guidance multiplies that 4-6x (GUIDED_EPS_ERR: measured at t = 999,
        # CFG 7.5, the largest of a schedule), and the update multiplies it again by scheduler.eps_amplification(index): 0.61 for
        # config 1's first update (4-step schedule, 999 -> 666; 2.7e-3 measured on the latent in the default mode: no fp16-operand
        # mode stays inside 1e-3 there), 0.113 on the 50-step schedule's first (4.1e-4 measured).
        #   "auto" (default): wide where  amplification x GUIDED_EPS_ERR[mode] x guidance / 7.5  -- the predicted relative error
        #                     of x_prev for unit-scale latents (GUIDED_EPS_ERR scaled down with the noise level as measured) -- exceeds
        #                     `operand_budget` (1e-3, the north star): steps 1-3 of a 4-step schedule, the first 4 of 10 or of 25
        #                     steps, NONE of the 50-step schedule the headline metric runs (8.4e-4 predicted at its first step in
        #                     the default mode);
        #   "f16": never;  "wide": every step;  or a collection of schedule indices.
        # A wide step costs several times an fp16-mode step (3 MFMAs per product on a plain register-staged kernel, fp32 attention
        # and norms); its windows are evaluated `wide_tile_batch` at a time (fp32 operands: 32-bit buffer addressing bounds a
        # launch's A operand).
        # several ranks: "last" = pred-x0 tiles are exchanged on the loop's last step only (when that step's windows cover the
        # panorama); "every" = on every step (a step callback that reads the pred-x0 panorama needs this; the loops set it then)
        self.exchange_x0 = os.environ.get("DS_EXCHANGE_X0", "last")
        self.operand_policy = os.environ.get("DS_OPERAND_POLICY", "auto")
        self.operand_budget = 1e-3
        # "auto" walks a ladder per step: the model's own mode where its predicted error is inside the budget; else the rungs in this
        # order -- "strict" (the fp32 residual stream, single fp16 operands: +12 % per evaluation) where ITS prediction is inside, "wide"
        # (3.7x) otherwise.  ("wide",) = round 5's behaviour.
        self.operand_rungs = tuple(r for r in os.environ.get("DS_OPERAND_RUNGS", "strict,wide").split(",") if r)
        # "auto": the guided-eps error of a mode is measured on the caller's UNet at set-up (_calibrate_operands; cached on the UNet);
        # "table": GUIDED_EPS_ERR
        self.operand_calibration = os.environ.get("DS_OPERAND_CALIBRATION", "auto")
        self._calibration = {}               # mode -> (t of the measurement, measured guided-eps error)
        self.strict_steps_run = []
        self.wide_tile_batch = 1
        self._step_precision = None
        self.wide_steps_run = []             # (step i, schedule index) of the steps of the last loop that ran wide
        self.verbose = False

    # -- the bits of DiffusionPipeline the reference relies on --
    def to(self, device=None, torch_dtype=None):
        if device is not None:
            self._device = torch.device(device)
        if torch_dtype is not None:
            self.latent_dtype = torch_dtype
        return self

    def progress_bar(self, total=None):
        return _ProgressBar(total)

    @property
    def _execution_device(self):
        if self._device is not None:
            return self._device
        if torch.cuda.is_available():
            return torch.device("cuda", torch.cuda.current_device())
        raise RuntimeError("no HIP device: the DynamicScaler hot path has no CPU fallback")

    def _log(self, *a):
        if self.verbose:
            print(*a)

    # -- operand policy --
    def _unet(self):
        return getattr(getattr(self.pretrained_t2v, "model", None), "diffusion_model", None)

    def precision_for(self, index, guidance_scale):
        """"wide" or None for the DDIM step at schedule index `index` (see operand_policy)."""
        pol = self.operand_policy
        if not hasattr(self._unet(), "twin"):          # not the HIP UNet (fake eps models of the geometry tests)
            return None
        if pol in (None, "f16"):
            return None
        if pol in ("wide", "strict"):
            return pol
        if pol == "auto":
            unet = self._unet()
            mode = self._mode_name(unet)
            if getattr(unet, "operand_mode", "f16") == "wide":
                return None                              # the model itself already evaluates wide
            if self.predicted_error(index, guidance_scale, mode) <= self.operand_budget:
                return None
            for rung in self.operand_rungs:
                if rung == "strict":
                    if mode != "f32" and self.predicted_error(index, guidance_scale, "f32") <= self.operand_budget:
                        return "strict"
                elif rung == "wide":
                    return "wide"
                else:
                    raise ValueError(f"operand_rungs={self.operand_rungs!r}: expected 'strict' and / or 'wide'")
            return None
        if isinstance(pol, str):
            raise ValueError(f"operand_policy={pol!r}: expected 'auto', 'f16', 'strict', 'wide' or a collection of schedule indices")
        return "wide" if index in pol else None

    @staticmethod
    def _mode_name(unet):
        return "f16" if unet.residual_dtype == torch.float16 else ("f32outer" if unet.residual_scope == "outer" else "f32")

    def guided_eps_error(self, mode, t):
        """Relative error of the CFG-combined eps of residual mode `mode` at timestep t: measured on this pipeline's UNet where a
        calibration exists (measured / OPERAND_SAFETY, carried to t along the measured noise-level curve), else the table's."""
        cal = self._calibration.get(mode)
        if cal is not None:
            t0, e0 = cal
            return e0 / OPERAND_SAFETY * _noise_shape(t) / _noise_shape(t0)
        return GUIDED_EPS_ERR[mode] * _noise_shape(t)

    def predicted_error(self, index, guidance_scale, mode=None):
        """Predicted relative error of x_prev (unit-scale latents) of the DDIM step at schedule index `index` evaluated in `mode`."""
        if mode is None:
            mode = self._mode_name(self._unet())
        t = float(self.scheduler.ddim_timesteps[index])
        return self.scheduler.eps_amplification(index) * self.guided_eps_error(mode, t) * max(1.0, abs(float(guidance_scale))) / 7.5

    def _begin_step(self, i, index, guidance_scale):
        """Fix the operand mode of step i (schedule index `index`) for every UNet evaluation until the next call."""
        if i == 0:
            self.wide_steps_run, self.strict_steps_run = [], []
        self._step_precision = self.precision_for(index, guidance_scale)
        if self._step_precision == "wide":
            self.wide_steps_run.append((int(i), int(index)))
        elif self._step_precision == "strict":
            self.strict_steps_run.append((int(i), int(index)))
        return self._step_precision

    def wide_steps_of(self, num_inference_steps, guidance_scale):
        """Schedule indices the current policy evaluates wide on a schedule of `num_inference_steps` (make_schedule must have run)."""
        return [ix for ix in range(num_inference_steps - 1, -1, -1) if self.precision_for(ix, guidance_scale) == "wide"]

    def strict_steps_of(self, num_inference_steps, guidance_scale):
        """... and the ones it evaluates on the strict rung (fp32 residual stream, fp16 operands)."""
        return [ix for ix in range(num_inference_steps - 1, -1, -1) if self.precision_for(ix, guidance_scale) == "strict"]

    @torch.no_grad()
    def _calibrate_operands(self, x, t, ctx_cond, ctx_uncond, guidance_scale, fps, frames, eval_kwargs):
        """Measure, on THIS UNet with THESE contexts, the relative error of the guided eps of the model's own residual mode (and of the
        strict rung) against the wide operand mode's fp32-level result: x [1,C,T,h,w] one tile at the schedule's first noise level,
        t its timestep.  One evaluation pair per mode (~0.2 s with the wide pair); the result is cached on the UNet per (packed
        generation, mode, t, guidance, tile shape, context length), so pipelines that share a model measure once."""
        unet = self._unet()
        if self.operand_policy != "auto" or self.operand_calibration != "auto" or not hasattr(unet, "twin") or ctx_uncond is None:
            return
        if getattr(unet, "operand_mode", "f16") == "wide" or not x.is_cuda:
            return
        model = self.pretrained_t2v.model
        own = self._mode_name(unet)
        cache = unet.__dict__.setdefault("_operand_calibration", {})
        g = float(guidance_scale)
        ctx = torch.cat([ctx_cond.to(x.device), ctx_uncond.to(x.device)], 0)
        xx = torch.cat([x, x], 0).contiguous()
        ts = torch.full((2,), int(t), device=x.device, dtype=torch.long)
        kw = {k: v for k, v in dict(eval_kwargs).items() if k != "precision"}

        def guided(precision):
            e = model(xx, ts, c_crossattn=[ctx], fps=fps, curr_time_steps=ts, temporal_length=frames, **(dict(kw, precision=precision) if precision else kw)).float()
            return e[1:] + g * (e[:1] - e[1:])

        ref = None
        for mode, precision in ((own, None), ("f32", "strict")):
            if mode == "f32" and (own == "f32" or "strict" not in self.operand_rungs):
                continue
            key = (getattr(unet, "_generation", 0), mode, int(t), round(g, 4), tuple(x.shape), int(ctx.shape[1]))
            if key not in cache:
                if ref is None:
                    ref = guided("wide")
                cache[key] = float((guided(precision) - ref).norm() / ref.norm().clamp_min(1e-30))
            self._calibration[mode] = (int(t), cache[key])

    def operand_report(self, num_inference_steps, guidance_scale):
        """What the policy decided and on which figures (bench.py's config.operand_policy)."""
        return {"policy": self.operand_policy if isinstance(self.operand_policy, str) else sorted(self.operand_policy), "budget": self.operand_budget,
                "rungs": list(self.operand_rungs), "safety": OPERAND_SAFETY,
                "guided_eps_err_measured": {m: {"t": t0, "err": e0} for m, (t0, e0) in self._calibration.items()},
                "guided_eps_err_table": dict(GUIDED_EPS_ERR) if not self._calibration else None,
                "predicted_first_step_error": self.predicted_error(num_inference_steps - 1, guidance_scale) if hasattr(self._unet(), "twin") else None,
                "strict_steps": self.strict_steps_of(num_inference_steps, guidance_scale), "wide_steps": self.wide_steps_of(num_inference_steps, guidance_scale)}

    # -- conditioning --
    def _encode(self, prompt, prompt_embeds, guidance_scale):
        if prompt is not None and isinstance(prompt, str):
            batch_size, prompt = 1, [prompt]
        elif prompt is not None and isinstance(prompt, list):
            batch_size = len(prompt)
        else:
            batch_size = prompt_embeds.shape[0]
        assert batch_size == 1, "the reference's latent init is only valid for batch 1 (t2v_sphere_panorama_pipeline.py:418)"
        text_emb = self.pretrained_t2v.get_learned_conditioning(prompt)
        uc_emb = None
        if guidance_scale != 1.0:
            uncond_type = self.pretrained_t2v.uncond_type
            if uncond_type == "empty_seq":
                uc_emb = self.pretrained_t2v.get_learned_conditioning(batch_size * [""])
            elif uncond_type == "zero_embed":
                uc_emb = torch.zeros_like(text_emb)
            else:
                raise NotImplementedError()
        return prompt, text_emb, uc_emb

    def _eps(self, x, t, ctx_list, fps, frames, cfg_pairs=None, **kwargs):
        """One batched evaluation of pretrained_t2v.model: x [n,C,T,h,w], ctx_list n context tensors [1,L,D].
        cfg_pairs=m: x is [tiles | tiles] (m cond + m uncond evaluations of the same latents): the HIP UNet then runs
        its context-free prefix once per pair (bit-identical result, `share_cfg_prefix`).
        With `use_graph` the evaluation is a hipGraph replay (one graph per input signature, captured on its second
        use): the ~1250 launches of a forward cost no host time, which is what bounds small tile batches (a rank's
        share of a level on 4-8 GPUs).  The returned tensor is then the graph's static output buffer: it is
        overwritten by the next evaluation of the same signature on the same stream slot."""
        n = x.shape[0]
        ctx = torch.cat([c.to(x.device) for c in ctx_list], dim=0)
        model = self.pretrained_t2v.model
        if cfg_pairs and self.share_cfg_prefix and hasattr(getattr(model, "diffusion_model", None), "c_program_trace"):      # the HIP UNet
            kwargs = dict(kwargs, cfg_pairs=int(cfg_pairs))
        if self._step_precision in ("wide", "strict") and hasattr(getattr(model, "diffusion_model", None), "twin"):
            kwargs = dict(kwargs, precision=self._step_precision)
        # graph replay only for signatures that a key can identify: python scalars in the kwargs (a tensor-valued kwarg would
        # be baked into the graph by pointer), an int fps
        scalar_kw = all(v is None or isinstance(v, (bool, int, float, str)) for v in kwargs.values())
        if not (self.use_graph and x.is_cuda and isinstance(fps, int) and scalar_kw):
            ts = torch.full((n,), int(t), device=x.device, dtype=torch.long)
            return model(x, ts, c_crossattn=[ctx], fps=fps, curr_time_steps=ts, temporal_length=frames, **kwargs)
        # the captured graphs hold the packed-weight pointers of ONE prepare(): a reload / .to() / invalidate() of the UNet
        # repacks into new buffers (generation + 1) and every older graph is dropped
        unet = getattr(model, "diffusion_model", None)
        if unet is not None and hasattr(unet, "prepare"):
            unet.prepare(x.device)           # no-op when packed; never inside a capture (the eager warm call comes first)
        gen = getattr(unet, "_generation", 0)
        if gen != self._graph_generation:
            self._graphs.clear()
            self._graph_generation = gen
        key = (self._slot, tuple(x.shape), x.dtype, tuple(ctx.shape), ctx.dtype, fps, frames, tuple(sorted(kwargs.items())), self._launch_share)
        ent = self._graphs.get(key)
        if ent is None:                      # first use: eager (loads code objects, sets kernel attributes)
            self._graphs[key] = "warm"
            ts = torch.full((n,), int(t), device=x.device, dtype=torch.long)
            return model(x, ts, c_crossattn=[ctx], fps=fps, curr_time_steps=ts, temporal_length=frames, **kwargs)
        if ent == "warm":                    # second use: capture
            sx, sctx = x.clone(), ctx.clone()
            sts = torch.full((n,), int(t), device=x.device, dtype=torch.long)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            # with a process group up, RCCL's watchdog thread polls events concurrently: only this thread's calls
            # may invalidate the capture then
            multi = torch.distributed.is_available() and torch.distributed.is_initialized()
            try:
                with torch.cuda.graph(g, capture_error_mode="thread_local" if multi else "global"):
                    out = model(sx, sts, c_crossattn=[sctx], fps=fps, curr_time_steps=sts, temporal_length=frames, **kwargs)
            except Exception as e:           # capture is an optimisation: fall back to the eager launches, loudly
                import warnings
                warnings.warn(f"hipGraph capture of the UNet evaluation failed ({type(e).__name__}: {e}); running eagerly")
                torch.cuda.synchronize()
                self.use_graph = False
                return model(sx, sts, c_crossattn=[sctx], fps=fps, curr_time_steps=sts, temporal_length=frames, **kwargs)
            ent = self._graphs[key] = (g, sx, sts, sctx, out)
        else:
            g, sx, sts, sctx, out = ent
            sx.copy_(x)
            sts.fill_(int(t))
            sctx.copy_(ctx)
        ent[0].replay()
        return ent[4]

    @torch.no_grad()
    def basic_sample(self, prompt=None, height=320, width=512, frames=16, fps=16, guidance_scale=7.5,
                     num_videos_per_prompt=1, generator=None, latents=None, num_inference_steps=4, prompt_embeds=None,
                     output_type="pil", skip_time_step_idx=None, **kwargs):
        """Single-tile loop (t2v_normal_pipeline.py:69-210).  Returns (videos | denoised, denoised) where
        `denoised` is the pred_x0 of the LAST step (:205-210)."""
        unet_config = self.model_config["params"]["unet_config"]
        frames = self.pretrained_t2v.temporal_length if frames < 0 else frames
        device = self._execution_device
        prompt, text_emb, uc_emb = self._encode(prompt, prompt_embeds, guidance_scale)
        self.scheduler.make_schedule(num_inference_steps, verbose=self.verbose)
        timesteps = np.flip(self.scheduler.ddim_timesteps)
        if skip_time_step_idx is not None:
            timesteps = timesteps[skip_time_step_idx:]
        total_steps = self.scheduler.ddim_timesteps.shape[0]
        if latents is None:
            assert (skip_time_step_idx is None) or (skip_time_step_idx == 0), \
                "[basic_sample] skip time step should only work with prepared non full noise latents"
            c = unet_config["params"]["in_channels"]
            shape = (1, 1, c, frames, height // self.vae_scale_factor, width // self.vae_scale_factor)
            latents = torch.randn(shape)[0]  # host draw, reference order
        latents = latents.to(device=device, dtype=self.latent_dtype).contiguous()
        kwargs.update({"clean_cond": True})
        denoised = None
        pano_shape = (1,) + tuple(latents.shape[1:])
        if uc_emb is not None and hasattr(self._unet(), "twin") and len(timesteps):
            self._calibrate_operands(latents[:1], timesteps[0], text_emb.to(device), uc_emb.to(device), guidance_scale, fps, frames, kwargs)
        with self.progress_bar(total=len(timesteps)) as bar:
            for i, t in enumerate(timesteps):
                self._begin_step(i, total_steps - i - 1, guidance_scale)
                if guidance_scale != 1.0:
                    eps = self._eps(torch.cat([latents, latents], 0), t, [text_emb, uc_emb], fps, frames,
                                    cfg_pairs=latents.shape[0], **kwargs)
                    e_c, e_u = eps[:1].contiguous(), eps[1:].contiguous()
                else:
                    e_c, e_u = self._eps(latents, t, [text_emb], fps, frames, **kwargs), None
                index = total_steps - i - 1
                coef = self.scheduler.step_coefficients(index)
                noise = self.scheduler.draw_step_noise(tuple(latents.shape), device, latents.dtype, coef["sigma"])
                latents, denoised = ops.cfg_ddim(latents, e_c, e_u, pano_shape, guidance_scale, coef, noise)
                bar.update()
        if not output_type == "latent":
            videos = self.pretrained_t2v.decode_first_stage_2DAE(denoised)
        else:
            videos = denoised
        return videos, denoised


    @torch.no_grad()
    def _basic_denoise_one_step(self, latent, t, i, total_steps, text_emb, uc_emb, guidance_scale, fps, frames, kwargs):
        """t2v_normal_pipeline.py:572-615: one CFG + DDIM step of a single tile at schedule index total_steps - i - 1."""
        kwargs = dict(kwargs)
        kwargs.update({"clean_cond": True})
        self._begin_step(i, total_steps - i - 1, guidance_scale)
        if guidance_scale != 1.0:
            eps = self._eps(torch.cat([latent, latent], 0), t, [text_emb, uc_emb], fps, frames, cfg_pairs=latent.shape[0],
                            **kwargs)
            e_c, e_u = eps[:1].contiguous(), eps[1:].contiguous()
        else:
            e_c, e_u = self._eps(latent, t, [text_emb], fps, frames, **kwargs), None
        coef = self.scheduler.step_coefficients(total_steps - i - 1)
        noise = self.scheduler.draw_step_noise(tuple(latent.shape), latent.device, latent.dtype, coef["sigma"])
        return ops.cfg_ddim(latent, e_c, e_u, (1,) + tuple(latent.shape[1:]), guidance_scale, coef, noise)

    # ------------------------------------------------------------------ the tile engine shared by all ring loops
    @torch.no_grad()
    def _stream_pool(self, device):
        if self._pool is None or self._pool.n != self.num_streams or self._pool.device != device:
            self._pool = parallel.StreamPool(device, self.num_streams)
        return self._pool

    def _denoise_windows(self, st, i, wins, ctxs, renoise, mask_frame0, merge_prev_ratio=None, use_mask=True):
        """Process the windows of DDIM step i (reference order `wins`) with the reference's sequential semantics:
        levels of pairwise-disjoint windows (parallel.plan_levels), each level = one batched gather -> re-noise/mix
        -> UNet [cond | uncond] -> CFG + DDIM (-> merge-prev) -> scatter.  st: _RingState."""
        t = st.timesteps[i]
        sched, device, pano = self.scheduler, st.device, st.pano
        coef = sched.step_coefficients(st.total_steps - i - 1)
        wide = self._begin_step(i, st.total_steps - i - 1, st.guidance_scale) == "wide"
        # host noise for the whole step in the reference's tile order (appendix B): randn_like(tile) of re_noise,
        # then `frames` per-frame draws of ddim_step -- per window.  (No host draws at all in rng_mode "device".)
        noises = []
        for _ in wins:
            nz = sched.draw_renoise_noise(st.tile_shape, "cpu", torch.float32) if renoise else None
            sn = sched.draw_step_noise(st.tile_shape, "cpu", torch.float32, coef["sigma"])
            noises.append((nz, sn))
        if renoise:
            c_rn, s_rn = sched.renoise_coefficients(st.total_steps - i - 2, st.total_steps - i - 1)
        self._log(f"i = {i}, t = {t}: {len(wins)} windows")
        mask = st.mask if use_mask else None

        split_cfg = [False]
        # the tile ops around the UNet fused (round 4; `fuse_tile_ops`, bit-identical to the separate kernels): the window is
        # re-noised while it is gathered (unless merge-prev needs the window as it was), and on one rank the update goes straight
        # into the panoramas -- windows of a level are pairwise disjoint, so a batch may scatter while the level's other batch
        # still gathers; over several ranks the tiles themselves are what is exchanged
        fuse_gather = self.fuse_tile_ops and renoise and mask is not None and merge_prev_ratio is None
        fuse_scatter = self.fuse_tile_ops and st.world == 1 and merge_prev_ratio is None

        def prepare(ids):
            """Gather + re-noise of the windows `ids` -> (tiles, mask tiles, the windows before the re-noise | None)."""
            origins = [(wins[j][4], wins[j][2], wins[j][0]) for j in ids]
            if fuse_gather:
                nz = None
                if noises[ids[0]][0] is not None:
                    nz = torch.cat([noises[j][0] for j in ids], 0).to(device=device, dtype=pano.dtype)
                numel = st.tile_shape[1] * st.tile_fhw[0] * st.tile_fhw[1] * st.tile_fhw[2]
                base = sched.tile_philox_offset(i, numel)
                tiles, mtiles = ops.ring_gather_renoise(pano, mask, origins, st.tile_fhw, c_rn, s_rn, st.ratio, noise=nz,
                                                        mask_frame0=mask_frame0, seed=sched.philox_seed,
                                                        tile_offsets=[base + j * numel for j in ids])
                return ids, tiles, mtiles, None
            tiles, mtiles = ops.ring_gather(pano, origins, st.tile_fhw, mask)
            prev = tiles.clone() if merge_prev_ratio is not None else None
            if renoise:
                nz = None
                if noises[ids[0]][0] is not None:
                    nz = torch.cat([noises[j][0] for j in ids], 0).to(device=device, dtype=pano.dtype)
                ops.renoise_mix_(tiles, mtiles, st.total_shape, c_rn, s_rn, st.ratio, noise=nz,
                                 mask_frame0=mask_frame0, seed=sched.philox_seed,
                                 offset=sched.tile_philox_offset(i, tiles[0].numel()), tile_ids=ids)
            return ids, tiles, mtiles, prev

        def finish(ctx, e_c, e_u, from_units=False):
            """CFG + DDIM update (+ merge-prev) of prepared tiles from their eps tensors -> (x_prev, x0) tiles, or (None, None) when
            the update went straight into the panoramas."""
            ids, tiles, mtiles, prev = ctx
            sn = None
            if coef["sigma"] != 0.0:
                sn = torch.cat([noises[j][1] for j in ids], 0).to(device=device, dtype=pano.dtype)
            if fuse_scatter and not from_units:
                for s0 in range(0, len(ids), ops.DS_MAX_WINDOWS):
                    part = ids[s0:s0 + ops.DS_MAX_WINDOWS]
                    n_ = len(part)
                    ops.cfg_ddim_scatter_(pano, st.pano_x0, mask, tiles[s0:s0 + n_].contiguous(), e_c[s0:s0 + n_].contiguous(),
                                          None if e_u is None else e_u[s0:s0 + n_].contiguous(), st.guidance_scale, coef,
                                          [(wins[j][4], wins[j][2], wins[j][0]) for j in part], None if sn is None else sn[s0:s0 + n_].contiguous())
                return None, None
            x_prev, x0 = ops.cfg_ddim(tiles, e_c, e_u, st.total_shape, st.guidance_scale, coef, sn)
            if merge_prev_ratio is not None:
                # merge-prev (i2v_sphere_panorama_pipeline.py:938-943): mix(x_prev, window_before_renoise, mask, r_i)
                ops.renoise_mix_(x_prev, mtiles, st.total_shape, 0.0, 1.0, merge_prev_ratio, noise=prev,
                                 mask_frame0=mask_frame0)
            return x_prev, x0

        def run_batch(ids):
            ctx = prepare(ids)
            tiles = ctx[1]
            n = len(ids)
            if st.guidance_scale != 1.0 and split_cfg[0]:
                # a level with ONE tile batch (a rank's share on many GPUs, single-tile panoramas): nothing else to overlap it
                # with, so the cond and the uncond evaluations go to two streams instead of one [cond | uncond] batch --
                # the same kernels on the same numbers (a batch equals its separate forwards), without the shared prefix
                def one(cl):
                    return self._eps(tiles, t, cl, st.fps, st.frames, **st.kwargs)
                # two equal launch sequences share the chip: the persistent GEMM tiles of each are planned on half of it
                # (ops.set_launch_share; scheduling only, same bits).  Captured graphs keep the plan they were captured with.
                self._launch_share = 2 if self.share_launches else 1
                ops.set_launch_share(self._launch_share)
                try:
                    e_c, e_u = self._stream_pool(device).map(one, [[ctxs[j] for j in ids], [st.uc_emb] * n], inline=self.use_graph,
                                                             on_slot=lambda k: setattr(self, "_slot", k))
                finally:
                    self._launch_share = 1
                    ops.set_launch_share(1)
            elif st.guidance_scale != 1.0:
                eps = self._eps(torch.cat([tiles, tiles], 0), t, [ctxs[j] for j in ids] + [st.uc_emb] * n,
                                st.fps, st.frames, cfg_pairs=n, **st.kwargs)
                e_c, e_u = eps[:n], eps[n:]
            else:
                e_c, e_u = self._eps(tiles, t, [ctxs[j] for j in ids], st.fps, st.frames, **st.kwargs), None
            return finish(ctx, e_c, e_u)

        # a level with fewer tiles than ranks is shared out by EVALUATION (parallel.run_step "units": cond and uncond of one
        # tile on two ranks).  A unit's eps does not depend on what else is in the batch (a batch equals its separate
        # forwards, tests/test_gpu_unet.py), so the panorama stays bit-identical to the single-process one.
        def unit_eps(ctx, units_mine):
            ids, tiles = ctx[0], ctx[1]
            if not units_mine:
                return torch.empty((0,) + tuple(tiles.shape[1:]), dtype=torch.float32, device=device)
            ks = [k for k, _ in units_mine]
            x = tiles[ks].contiguous() if ks != list(range(len(ids))) else tiles
            cl = [ctxs[ids[k]] if b == 0 else st.uc_emb for k, b in units_mine]
            return self._eps(x, t, cl, st.fps, st.frames, **st.kwargs).float()

        def unit_finish(ctx, e_all):
            return finish(ctx, e_all[:, 0].contiguous(), e_all[:, 1].contiguous(), from_units=True)

        units = parallel.EvalUnits(2, prepare, unit_eps, unit_finish) if (st.guidance_scale != 1.0 and st.world > 1) else None

        def process(mine):
            """(x_prev, x0) tiles of the pairwise-disjoint windows `mine` (this rank's part of a level), in that order.
            The tile batches are independent (the panorama is only read until the scatter): with num_streams > 1 they run
            concurrently on separate HIP streams, so the partial last round of workgroups of one batch's kernels is filled
            by the other batch's kernels."""
            bsz = self.max_tile_batch
            if wide:
                bsz = max(1, min(bsz, self.wide_tile_batch))
            if self.num_streams > 1:         # spread the windows over the streams
                bsz = max(1, min(bsz, -(-len(mine) // self.num_streams)))
            batches = [mine[s:s + bsz] for s in range(0, len(mine), bsz)]
            split_cfg[0] = self.num_streams > 1 and (self.split_cfg_over_streams == 2 or (self.split_cfg_over_streams == 1 and len(batches) == 1))
            if self.num_streams > 1 and len(batches) > 1 and not split_cfg[0]:
                # graph replays are enqueued by this thread (one call per evaluation); eager launches need a host
                # thread per stream to keep both streams fed
                parts = self._stream_pool(device).map(run_batch, batches, inline=self.use_graph,
                                                      on_slot=lambda k: setattr(self, "_slot", k))
            else:
                parts = [run_batch(ids) for ids in batches]
            if parts and parts[0][0] is None:       # already in the panoramas (fuse_scatter)
                return None, None
            return torch.cat([p[0] for p in parts], 0), torch.cat([p[1] for p in parts], 0)

        def scatter(ids, xp, x0):
            if xp is None:
                return
            for s in range(0, len(ids), ops.DS_MAX_WINDOWS):
                part = ids[s:s + ops.DS_MAX_WINDOWS]
                origins = [(wins[j][4], wins[j][2], wins[j][0]) for j in part]
                # x0 None: another rank's tiles on a step that does not exchange pred-x0 (see need_x0 below)
                ops.ring_scatter3(pano, st.pano_x0 if x0 is not None else None, mask, xp[s:s + len(part)].contiguous(),
                                  None if x0 is None else x0[s:s + len(part)].contiguous(), origins)

        def empty_tiles():
            return torch.empty((0,) + st.tile_shape[1:], dtype=pano.dtype, device=device)

        # levels of pairwise-disjoint windows; over several ranks whole components (columns) per rank with one exchange
        # per step, or a strided share of every level (parallel.run_step)
        # The pred-x0 panorama is read when the loop ends (or by a step callback), never by the next step: where the loop's LAST step
        # covers the whole panorama (st.x0_last_only, set by the ring loops from their window grid) only that step exchanges pred-x0
        # tiles -- until then a rank's pred-x0 replica is complete only over the tiles it computed itself.
        need_x0 = not (st.world > 1 and getattr(st, "x0_last_only", False) and i < st.total_steps - 1)
        st.share_mode = self.last_share_mode = parallel.run_step(wins, st.pano_fhw, st.rank, st.world, process, scatter, empty_tiles,
                                                                      units=units, need_x0=need_x0)

    def _new_state(self, init_panorama_latent, total_shape, timesteps, frames, fps, lat_h, lat_w, guidance_scale,
                   text_emb, uc_emb, ratio, kwargs):
        device = self._execution_device
        # weights are repacked here, on the caller's stream and followed by a device synchronisation, never lazily by
        # whichever side stream happens to run the first evaluation
        unet = getattr(getattr(self.pretrained_t2v, "model", None), "diffusion_model", None)
        if unet is not None and hasattr(unet, "prepare"):
            unet.prepare(device)
        st = _RingState()
        st.in_device = init_panorama_latent.device    # overwritten with the execution device when the loop drew the latent itself
        st.pano = init_panorama_latent.to(device=device, dtype=self.latent_dtype).contiguous().clone()
        st.pano_x0 = torch.zeros_like(st.pano)
        st.mask = torch.zeros(total_shape[2:], dtype=torch.uint8, device=device)  # 1 byte per (f,y,x)
        st.rank, st.world = 0, 1
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            st.rank, st.world = torch.distributed.get_rank(), torch.distributed.get_world_size()
        st.timesteps, st.total_steps = timesteps, len(timesteps)
        st.total_shape, st.device, st.frames, st.fps = total_shape, device, frames, fps
        st.lat_h, st.lat_w = lat_h, lat_w
        st.tile_fhw = (frames, lat_h, lat_w)
        st.tile_shape = (1, total_shape[1]) + st.tile_fhw
        st.pano_fhw = tuple(total_shape[2:])
        st.guidance_scale, st.ratio = guidance_scale, ratio
        st.text_emb = text_emb.to(device)
        st.uc_emb = uc_emb.to(device) if uc_emb is not None else None
        kwargs = dict(kwargs)
        kwargs.update({"clean_cond": True})
        st.kwargs = kwargs
        if unet is not None and hasattr(unet, "twin") and st.uc_emb is not None and len(timesteps):
            # operand policy on the weights and contexts it runs on: one tile of the panorama at the loop's first timestep (i2v: the
            # text tokens with the zero-image tokens of the uncond context: the per-window image tokens do not exist yet)
            ctx_c = st.text_emb
            if st.uc_emb.shape[1] > ctx_c.shape[1]:
                ctx_c = torch.cat([ctx_c, st.uc_emb[:, ctx_c.shape[1]:].to(ctx_c.dtype)], 1)
            tile, _ = ops.ring_gather(st.pano, [(0, 0, 0)], st.tile_fhw, None)
            self._calibrate_operands(tile, timesteps[0], ctx_c, st.uc_emb, guidance_scale, fps, frames, kwargs)
            self._prepare_rungs(unet, device, len(timesteps), guidance_scale)
        return st

    def _prepare_rungs(self, unet, device, nsteps, guidance_scale):
        """The packed buffers of the rungs this loop will use, before any side stream runs (strict: a second fp16 image with its own
        LayerNorm handling; wide: hi + lo planes)."""
        need = {self.precision_for(ix, guidance_scale) for ix in range(self.scheduler.ddim_timesteps.shape[0])}     # (a superset for cut schedules)
        for rung in ("strict", "wide"):
            if rung in need:
                unet.twin(rung).prepare(device)

    def _finish(self, st, output_type, total_frames, seam_safe):
        """Return tuple of the ring loops.  Like the reference (t2v_sphere_panorama_pipeline.py:636-660) the second element
        is moved to the device of the `init_panorama_latent` the caller passed (the execution device when the loop drew its
        own), and in the seam-safe decode branch it is the W-PADDED latent (the reference reassigns `denoised` there)."""
        self.final_latent = st.pano  # x_t panorama after the last step (not returned by the reference's ring variants)
        denoised = st.pano_x0.clone()
        out_dev = st.in_device
        if output_type == "latent":
            denoised = denoised.to(out_dev)
            return denoised, denoised
        if not seam_safe:
            return self.pretrained_t2v.decode_first_stage_2DAE(denoised), denoised.to(out_dev)
        # seam-safe decode (t2v_sphere_panorama_pipeline.py:638-655): pad W with wrapped 1/16 chunks, decode per
        # frame, crop (vae.AutoencoderKL behind decode_first_stage_2DAE).
        chunks = list(torch.chunk(denoised, 16, dim=4))
        padded = torch.cat([chunks[-1]] + chunks + [chunks[0]], dim=4)
        frames_out = [self.pretrained_t2v.decode_first_stage_2DAE(padded[:, :, [f]]) for f in range(total_frames)]
        videos = torch.cat(frames_out, dim=2)
        videos = torch.cat(torch.chunk(videos, 18, dim=4)[1:-1], dim=4)
        return videos, padded.to(out_dev)

    @torch.no_grad()
    def basic_sample_shift_multi_windows(self, prompt=None, height=320, width=512, frames=16, fps=16, guidance_scale=7.5,
                                         num_videos_per_prompt=1, generator=None, init_panorama_latent=None,
                                         clear_pre_denoised_latent=None, clear_pre_denoised_video_tensor=None,
                                         num_windows_w=None, num_windows_h=None, num_windows_f=None, loop_step=None,
                                         latents=None, num_inference_steps=50, prompt_embeds=None, output_type="pil",
                                         use_pre_denoise=False, pre_denoise_steps=None, skip_steps_after_pre_denoise=0,
                                         shift_jump_odd_w=False, shift_jump_odd_h=False, shift_jump_odd_f=False,
                                         docking_w=False, docking_h=False, docking_f=False, docking_step_range=None,
                                         merge_predenoise_ratio_list=None, random_shuffle_init_frame_stride=0,
                                         sparse_add_residual=True, use_skip_time=False, skip_time_step_idx=None,
                                         progressive_skip=False, step_callback=None, **kwargs):
        """Non-overlapping shifted grid (pipeline/t2v_normal_pipeline.py:213-568): the panorama is exactly
        num_windows_h x num_windows_w tiles; every step the grid is shifted by (i % loop_step) * tile/loop_step in
        W, H and F (wrap-around), optionally jumped by half the panorama on odd steps and docked to the borders.
        No mask / re-noise in this variant.  Pre-denoise start (:345-412): one tile denoised for `pre_denoise_steps`
        steps (or `clear_pre_denoised_latent`), resized bicubically to the panorama, `_add_noise`d -- with
        `use_skip_time` the schedule is cut (non-progressive) or the first frames get progressively lower noise levels;
        per step the panorama is merged with that resized latent re-noised to the step's level, densely or on the
        reference's sparse checkerboard (:445-468, `ds_residual_merge`)."""
        unet_config = self.model_config["params"]["unet_config"]
        frames = self.pretrained_t2v.temporal_length if frames < 0 else frames
        prompt, text_emb, uc_emb = self._encode(prompt, prompt_embeds, guidance_scale)
        self.scheduler.make_schedule(num_inference_steps, verbose=self.verbose)
        full_timesteps = np.flip(self.scheduler.ddim_timesteps)
        if use_skip_time and not progressive_skip:
            timesteps = full_timesteps[skip_time_step_idx - skip_steps_after_pre_denoise:]       # :299-301
        else:
            timesteps = full_timesteps
        total_steps = len(timesteps)
        vs = self.vae_scale_factor
        lat_h, lat_w = height // vs, width // vs
        c_lat = unet_config["params"]["in_channels"]
        total_shape = (1, c_lat, frames * num_windows_f, lat_h * num_windows_h, lat_w * num_windows_w)
        resized, fm = None, True
        if init_panorama_latent is None:
            init_panorama_latent = torch.randn(total_shape)                              # host draw, reference order
            if random_shuffle_init_frame_stride > 0:
                # t2v_normal_pipeline.py:328-337, literally: the reference shuffles slices of the init latent with Python's global
                # `random` -- and indexes dim 3 (the H axis of [B, C, F, H, W]) with its FRAME indices; the same statements on the same
                # host tensor reproduce it (and its shape error for panoramas lower than their frame count), repeatable under random.seed
                import random
                stride_ = int(random_shuffle_init_frame_stride)
                for frame_index in range(frames, frames * num_windows_f, stride_):
                    list_index = list(range(frame_index - frames, frame_index + stride_ - frames))
                    random.shuffle(list_index)
                    init_panorama_latent[:, :, :, frame_index:frame_index + stride_] = init_panorama_latent[:, :, :, list_index]
            init_panorama_latent = init_panorama_latent.to(self._execution_device)     # lives on the execution device like the reference's
            if use_skip_time:
                assert use_pre_denoise and pre_denoise_steps > 0, \
                    "[basic_sample_shift_multi_windows] skip ts should be used with pre denoise if init_panorama_latent is not provided "
                assert skip_time_step_idx >= skip_steps_after_pre_denoise, \
                    f"[basic_sample_shift_multi_windows] skip_time_step_idx {skip_time_step_idx} should >=skip_steps_after_pre_denoise {skip_steps_after_pre_denoise}"
            if use_pre_denoise and pre_denoise_steps > 0:
                if (num_windows_h != 1 or num_windows_w != 1) and num_windows_f != 1:
                    raise NotImplementedError()
                device = self._execution_device
                basic_shape = (1, c_lat, frames, lat_h, lat_w)
                latent = torch.randn(basic_shape)            # drawn in every branch (:358)
                from .tensor_utils import resize_video_latent
                fm = True     # `resized` is a permuted view in the reference: its randn_like draws follow frames-major strides
                if clear_pre_denoised_video_tensor is not None:
                    # a clear CLIP in pixel space (:363-368): bicubic resize to the panorama size, first-stage encode (posterior
                    # noise drawn on the host in the reference's order); the encoder's output is contiguous [B,C,F,H,W]
                    clip = clear_pre_denoised_video_tensor.to(device=device, dtype=torch.float32).contiguous()
                    resized_clip = resize_video_latent(clip, height * num_windows_h, width * num_windows_w, mode="bicubic")
                    resized = self.pretrained_t2v.encode_first_stage_2DAE(resized_clip).to(self.latent_dtype).contiguous()
                    assert tuple(resized.shape) == total_shape, \
                        f"[basic_sample_shift_multi_windows] encoded clip {tuple(resized.shape)} != panorama latent {total_shape}"
                    fm = False
                else:
                    if clear_pre_denoised_latent is not None:
                        assert tuple(clear_pre_denoised_latent.shape) == basic_shape, \
                            f"[basic_sample_shift_multi_windows] clear_pre_denoised_latent shape :{tuple(clear_pre_denoised_latent.shape)}" \
                            f"not equal to _basic_latent_shape: {basic_shape}"
                        latent = clear_pre_denoised_latent.clone()
                    latent = latent.to(device=device, dtype=self.latent_dtype).contiguous()
                    if clear_pre_denoised_latent is None:
                        self._log(f"[basic_sample_shift_multi_windows] Pre Denosing {pre_denoise_steps} Steps...")
                        for i, t in enumerate(full_timesteps[:pre_denoise_steps]):
                            latent, _ = self._basic_denoise_one_step(latent, t, i, total_steps, text_emb, uc_emb,
                                                                     guidance_scale, fps, frames, kwargs)
                    resized = resize_video_latent(latent, lat_h * num_windows_h, lat_w * num_windows_w, mode="bicubic")
                init_panorama_latent = self.scheduler.add_noise(resized, total_steps - 1, frames_major_strides=fm)
                if use_skip_time:
                    if progressive_skip:
                        for frame_idx, progs_skip_idx in enumerate(list(reversed(range(skip_time_step_idx)))):
                            noised = self.scheduler.add_noise(resized[:, :, [frame_idx]].contiguous(),
                                                              total_steps - progs_skip_idx - 1)
                            init_panorama_latent[:, :, [frame_idx]] = noised
                    else:
                        init_panorama_latent = self.scheduler.add_noise(resized, total_steps - 1, frames_major_strides=fm)
        else:
            assert tuple(init_panorama_latent.shape) == total_shape, \
                f"[basic_sample_shift_multi_windows] init_panorama_latent shape {tuple(init_panorama_latent.shape)} " \
                f"does not match desired shape {total_shape}"
        step_f = 0 if num_windows_f == 1 else frames // loop_step
        assert step_f > 0 or num_windows_f == 1, \
            f"[basic_sample_shift_multi_windows] loop_step {loop_step} > frames {frames} while num_windows_f {num_windows_f} > 0"
        st = self._new_state(init_panorama_latent, total_shape, timesteps, frames, fps, lat_h, lat_w, guidance_scale,
                             text_emb, uc_emb, None, kwargs)
        for i in range(len(timesteps)):
            if use_pre_denoise and merge_predenoise_ratio_list is not None and resized is not None:       # :445-468
                assert len(merge_predenoise_ratio_list) == len(timesteps), \
                    f"merge_predenoise_ratio_list ({len(merge_predenoise_ratio_list)}) should have same length as timesteps({len(timesteps)})"
                noised_resized = self.scheduler.re_noise(resized, 0, total_steps - i - 1, frames_major_strides=fm)
                st.pano = ops.residual_merge(st.pano, noised_resized, merge_predenoise_ratio_list[i], i, sparse_add_residual)
            wins = t2v_grid_windows(i, latent_h=lat_h, latent_w=lat_w, frames=frames, num_windows_w=num_windows_w,
                                    num_windows_h=num_windows_h, num_windows_f=num_windows_f, loop_step=loop_step,
                                    shift_jump_odd_w=shift_jump_odd_w, shift_jump_odd_h=shift_jump_odd_h,
                                    shift_jump_odd_f=shift_jump_odd_f, docking_w=docking_w, docking_h=docking_h,
                                    docking_f=docking_f, docking_step_range=docking_step_range)
            self._denoise_windows(st, i, wins, [st.text_emb] * len(wins), renoise=False, mask_frame0=True, use_mask=False)
            if step_callback is not None:
                step_callback(i, int(timesteps[i]), wins, st.pano, st.pano_x0)
        return self._finish(st, output_type, total_shape[2], seam_safe=False)


class _RingState:
    """Mutable state of one ring sampling run (panoramas resident in HBM)."""


class VC2_Pipeline_T2V_SpherePano(VC2_Pipeline_T2V):
    """pipeline/t2v_sphere_panorama_pipeline.py:20 -- the overlapped-ring plane loop (:316-660), which is the path
    all BASELINE configs take."""

    @torch.no_grad()
    def basic_sample_shift_multi_windows(self, prompt=None, height=320, width=512, frames=16, fps=16,
                                         guidance_scale=7.5, num_videos_per_prompt=1, generator=None,
                                         init_panorama_latent=None, total_w=None, total_h=None, num_windows_w=None,
                                         num_windows_h=None, num_windows_f=None, loop_step=None, dock_at_h=None,
                                         latents=None, num_inference_steps=4, prompt_embeds=None, output_type="pil",
                                         merge_renoised_overlap_latent_ratio=1, window_multi_prompt_dict=None,
                                         use_skip_time=False, skip_time_step_idx=None, progressive_skip=False,
                                         step_callback=None, **kwargs):
        st = self.ring_begin(prompt=prompt, height=height, width=width, frames=frames, fps=fps,
                             guidance_scale=guidance_scale, init_panorama_latent=init_panorama_latent, total_w=total_w,
                             total_h=total_h, num_windows_w=num_windows_w, num_windows_h=num_windows_h,
                             num_windows_f=num_windows_f, loop_step=loop_step, dock_at_h=dock_at_h,
                             num_inference_steps=num_inference_steps, prompt_embeds=prompt_embeds,
                             merge_renoised_overlap_latent_ratio=merge_renoised_overlap_latent_ratio,
                             window_multi_prompt_dict=window_multi_prompt_dict, use_skip_time=use_skip_time,
                             skip_time_step_idx=skip_time_step_idx, progressive_skip=progressive_skip, **kwargs)
        if step_callback is not None:
            st.x0_last_only = False          # the callback sees the pred-x0 panorama of every step: exchange it on every step
        with self.progress_bar(total=len(st.timesteps)) as bar:
            for i in range(len(st.timesteps)):
                wins = self.ring_step(st, i)
                if step_callback is not None:
                    step_callback(i, int(st.timesteps[i]), wins, st.pano, st.pano_x0)
                bar.update()
        return self.ring_finish(st, output_type)

    # ---- the loop in three pieces so a caller (bench.py) can time individual DDIM steps ----
    @torch.no_grad()
    def ring_begin(self, prompt=None, height=320, width=512, frames=16, fps=16, guidance_scale=7.5,
                   init_panorama_latent=None, total_w=None, total_h=None, num_windows_w=None, num_windows_h=None,
                   num_windows_f=None, loop_step=None, dock_at_h=None, num_inference_steps=4, prompt_embeds=None,
                   merge_renoised_overlap_latent_ratio=1, window_multi_prompt_dict=None, use_skip_time=False,
                   skip_time_step_idx=None, progressive_skip=False, **kwargs):
        unet_config = self.model_config["params"]["unet_config"]
        frames = self.pretrained_t2v.temporal_length if frames < 0 else frames
        prompt, text_emb, uc_emb = self._encode(prompt, prompt_embeds, guidance_scale)
        self.scheduler.make_schedule(num_inference_steps, verbose=self.verbose)
        timesteps = np.flip(self.scheduler.ddim_timesteps)
        if use_skip_time and not progressive_skip:
            timesteps = timesteps[skip_time_step_idx:]
        vs = self.vae_scale_factor
        c_lat = unet_config["params"]["in_channels"]
        total_shape = (1, c_lat, frames * num_windows_f, total_h // vs, total_w // vs)
        if init_panorama_latent is None:
            init_panorama_latent = torch.randn(total_shape).to(self._execution_device)  # host draw (global CPU generator), reference order
            if use_skip_time:
                raise NotImplementedError  # same as the reference (:420-422)
        else:
            assert tuple(init_panorama_latent.shape) == total_shape, \
                f"[basic_sample_shift_multi_windows] init_panorama_latent shape {tuple(init_panorama_latent.shape)} " \
                f"does not match desired shape {total_shape}"
        st = self._new_state(init_panorama_latent, total_shape, timesteps, frames, fps, height // vs, width // vs,
                             guidance_scale, text_emb, uc_emb, merge_renoised_overlap_latent_ratio, kwargs)
        ov_w, st.step_w, st.off_w = ring_axis_steps(total_w, width, num_windows_w, loop_step)
        assert 0 <= ov_w < 1, "overlap ratio for W is not legal"
        assert st.off_w, "latent_offset_step_size_w <= 0 ! consider increase W windows"
        ov_h, st.step_h, st.off_h = ring_axis_steps(total_h, height, num_windows_h, loop_step)
        assert 0 <= ov_h < 1, "overlap ratio for H is not legal"
        assert st.off_h > 0, "latent_offset_step_size_h <= 0 ! consider increase H windows"
        st.step_f = frames // loop_step
        if num_windows_f == 1:
            st.step_f = 0
        assert st.step_f > 0 or num_windows_f == 1, \
            f"[basic_sample_shift_multi_windows] loop_step {loop_step} > frames {frames} while num_windows_f {num_windows_f} > 0"
        st.total_lat_h = total_h // vs
        st.nw, st.nh, st.nf, st.loop_step, st.dock_at_h = num_windows_w, num_windows_h, num_windows_f, loop_step, dock_at_h
        st.window_multi_prompt_dict, st.prompt_cache = window_multi_prompt_dict, {}
        return st

    @torch.no_grad()
    def ring_step(self, st, i):
        """One DDIM step over all windows of the panorama (t2v_sphere_panorama_pipeline.py:481-634)."""
        st.mask.zero_()  # fresh mask every step (:494)
        wins = t2v_ring_windows(i, latent_h=st.lat_h, latent_w=st.lat_w, frames=st.frames,
                                total_latent_h=st.total_lat_h, step_w=st.step_w, step_h=st.step_h, off_w=st.off_w,
                                off_h=st.off_h, step_f=st.step_f, num_windows_w=st.nw, num_windows_h=st.nh,
                                num_windows_f=st.nf, loop_step=st.loop_step, dock_at_h=st.dock_at_h)
        if st.world > 1 and not hasattr(st, "x0_last_only"):
            last = t2v_ring_windows(st.total_steps - 1, latent_h=st.lat_h, latent_w=st.lat_w, frames=st.frames,
                                    total_latent_h=st.total_lat_h, step_w=st.step_w, step_h=st.step_h, off_w=st.off_w,
                                    off_h=st.off_h, step_f=st.step_f, num_windows_w=st.nw, num_windows_h=st.nh,
                                    num_windows_f=st.nf, loop_step=st.loop_step, dock_at_h=st.dock_at_h)
            st.x0_last_only = self.exchange_x0 == "last" and parallel.windows_cover(last, st.pano_fhw)
        renoise = st.ratio is not None and i < st.total_steps - 1
        # per-window prompt (R13): embeddings cached per distinct prompt instead of re-running CLIP per tile
        ctxs = []
        for (l, r, tp, dn, fb, fe) in wins:
            if st.window_multi_prompt_dict is not None:
                cur = select_prompt_from_multi_prompt_dict_by_factor(st.window_multi_prompt_dict, dn / st.total_lat_h)
                if cur not in st.prompt_cache:
                    st.prompt_cache[cur] = self.pretrained_t2v.get_learned_conditioning([cur]).to(st.device)
                ctxs.append(st.prompt_cache[cur])
            else:
                ctxs.append(st.text_emb)
        self._denoise_windows(st, i, wins, ctxs, renoise=renoise, mask_frame0=True)
        return wins

    @torch.no_grad()
    def ring_finish(self, st, output_type="latent"):
        return self._finish(st, output_type, st.total_shape[2], seam_safe=True)
