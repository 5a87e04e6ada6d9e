"""UNetModel -- MI355X host side of the VideoCrafter LVDM 3D-UNet.

Drop-in for `lvdm.modules.networks.openaimodel3d.UNetModel` (openaimodel3d.py:312-708): same constructor
keywords (the yaml `unet_config.params`), same `forward(x, timesteps, context, features_adapter, fps,
timestep_cond, **kwargs)`, same state-dict keys.  The arithmetic is NOT torch: the parameters are repacked once
into the layouts the HIP kernels want (fp16 [N][K] GEMM operands, fused QKV / KV / GEGLU-interleaved / all
time-embedding projections in one matrix) and the forward is ONE C call -- ds_unet_forward (csrc/unet_program.hip):
the block program, the scratch arena and the launch loop over the kernels live in C++, on channel-contiguous
"NTHWC" activations.  This module owns the parameters, the packed buffers and the C handle(s).  There is no CPU
fallback.  (Rounds 2-4 kept a Python restatement of the launch program next to the C one; round 5 removed it: its
instrumentation -- per-launch hooks, per-block taps -- is offered by the C program itself, ds_unet_set_hooks.)

Beyond the reference (which only runs batch 1): a batch of b independent evaluations
(cond + uncond CFG branches, several tiles) shares one launch sequence; results per item are those of b
separate b=1 forwards.
"""
import os
import threading

import torch
import torch.nn as nn

from . import ops, _lib
from .unet_spec import build_program, param_shapes

HEAD_DIM = 64


class _Container(nn.Module):
    """Name-only node of the parameter tree (keeps the reference's state-dict keys)."""


def _interleave_geglu(w):
    """[2*inner, ...] rows (x half | gate half) -> 32-row groups [x_g | gate_g] (DS_EPI_GEGLU layout: one 32x32 MFMA
    tile of x next to the tile of its gates, so the product is formed in registers)."""
    inner = w.shape[0] // 2
    assert inner % 32 == 0, "GEGLU inner dim must be a multiple of 32"
    x, g = w[:inner], w[inner:]
    xs = x.reshape(inner // 32, 32, *w.shape[1:])
    gs = g.reshape(inner // 32, 32, *w.shape[1:])
    return torch.cat([xs, gs], dim=1).reshape(w.shape)


class UNetModel(nn.Module):
    def __init__(self, **params):
        super().__init__()
        self.cfg, self._inputs, self._middle, self._outputs = build_program(params)
        # per decoder group: (channels of h, channels of the skip tensor) of its torch.cat([h, skip], dim=1)
        skip_ch, ch = [], 0
        for group in self._inputs:
            for b in group:
                if b.kind in ("conv_in", "res", "down", "up"):
                    ch = b.cout
            skip_ch.append(ch)
        self._cat_ch = [(group[0].cin - c, c) for group, c in zip(self._outputs, reversed(skip_ch))]
        cfg = self.cfg
        self.in_channels = cfg["in_channels"]
        self.model_channels = cfg["model_channels"]
        self.out_channels = cfg["out_channels"]
        self.num_res_blocks = cfg["num_res_blocks"]
        self.attention_resolutions = cfg["attention_resolutions"]
        self.channel_mult = cfg["channel_mult"]
        self.temporal_attention = cfg["temporal_attention"]
        self.addition_attention = cfg["addition_attention"]
        self.use_image_attention = cfg["use_image_attention"]
        self.fps_cond = cfg["fps_cond"]
        self.dtype = torch.float16 if cfg["use_fp16"] else torch.float32  # attribute read by callers
        if cfg["num_head_channels"] != HEAD_DIM and cfg["num_head_channels"] != -1:
            raise NotImplementedError("the HIP attention kernels are built for head_dim 64 (the VideoCrafter configs)")
        if cfg["time_cond_proj_dim"] is not None:
            raise NotImplementedError("time_cond_proj_dim is not used by DynamicScaler")
        self._shapes = param_shapes(params)
        for key, shape in self._shapes.items():
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                if not hasattr(node, p):
                    node.add_module(p, _Container())
                node = getattr(node, p)
            node.register_parameter(parts[-1], nn.Parameter(torch.empty(shape), requires_grad=False))
        self._packed = None
        # LayerNorm folded into the projection it feeds (the normalised activation is never rounded to fp16 nor written to
        # memory): True (default) = ds_layernorm_stats + ds_gemm_f16_ln; False = LayerNorm kernel + plain GEMM.  DS_FOLD_LN = 0 | 1;
        # after changing the attribute call invalidate() (the packed projection weights differ).  profiles/r2_notes.md section 4.
        # (ds_gemm_f16_lnk -- the statistics taken inside the GEMM -- stays a kernel of the library; measured slower at the bench's
        # batch sizes, no launch program uses it.)
        fold = os.environ.get("DS_FOLD_LN", "1")
        if fold not in ("0", "1"):
            raise ValueError(f"DS_FOLD_LN={fold!r}: expected 0 (LayerNorm kernels) or 1 (folded into the consumer GEMM)")
        self.fold_layernorm = fold == "1"
        # Storage type of the RESIDUAL STREAM (every "+ x" / "skip + h" output, conv_in, down / up-sample, proj_in; the skip
        # tensors).  The reference computes in fp32 throughout (openaimodel3d.py:657-708); the matrix-core operands are fp16 in
        # both modes.  torch.float16: everything fp16 (fastest).  torch.float32 ("strict"): the stream is stored, added and
        # normalised in fp32 -- the error budget's largest term (profiles/r2_notes.md section 2: 1.31e-3 of 1.66e-3 on eps) goes
        # away; norms read fp32 and write the fp16 operand, residual epilogues add fp32 rows (DS_EPI_RES_F32 | DS_EPI_OUT_F32).
        # `residual_scope` (fp32 stream only): "full" = also the stream INSIDE the transformers (proj_in output and the three adds
        # of every block; the LayerNorm fold is then off, it needs the raw activation as an fp16 operand); "outer" = only the
        # stream between the blocks (ResBlock / temporal-conv outputs, proj_out + x_in, conv_in, down / up-sample, the skip
        # tensors) -- the error budget puts 1.19e-3 of the stream's 1.31e-3 there (profiles/r3_notes.md section 2) -- while
        # the transformers keep their fp16 inner stream and the fold: most of the strict mode's gain at about a third of its cost.
        # DS_RESIDUAL_DTYPE = f16 | f32 | f32outer; after changing the attributes call invalidate().
        # GroupNorm statistics from the producing GEMM's epilogue (ds_gemm_f16_stats) where producer and norm are adjacent: no
        # statistics pass over the tensor.  OPT-IN (DS_GN_FROM_PRODUCER=1): measured on MI355X it does not pay -- the statistics
        # launches shrink by 7.2 ms per cfg3 step, the producers' epilogues grow by 4.8 ms, the table reduction costs 1.6 ms and the
        # apply pass, which the statistics pass used to leave a warm Infinity Cache for, 1.9 ms (profiles/r4_notes.md section 3).
        # With it the partial sums follow the producer's tile variant, which is chosen from the launch size: results stay run-to-run
        # repeatable, but a batch then equals its separate forwards to fp32 rounding only.  The default keeps every kernel form a
        # function of the instance shape alone: a rank-sharded run reproduces the one-GPU panorama bit for bit (DS_BATCH_INVARIANT=1
        # forces that whatever else is set).
        self.batch_invariant = os.environ.get("DS_BATCH_INVARIANT", "0") == "1"
        self.gn_from_producer = os.environ.get("DS_GN_FROM_PRODUCER", "0") == "1"
        # DEFAULT (round 4): f32outer -- the cheapest mode inside the north star's 1e-3 with margin on every asserted latent bound,
        # on the schedule the metric runs: the ring loop with the real UNet over the first six steps of the 50-step schedule ends
        # at 7.0e-4 (fast mode 9.6e-4, strict 5.3e-4; tests/test_gpu_fullsize.py::test_ring_loops_real_unet_on_the_50_step_schedule...),
        # free-running 50 steps 7.1e-4 (9.8e-4 / 5.3e-4), x_prev of a unit-scale model's first step 9.4e-4 (1.27e-3 / 7.1e-4).
        rd = os.environ.get("DS_RESIDUAL_DTYPE", "f32outer")
        if rd not in ("f16", "f32", "f32outer", "wide"):
            raise ValueError(f"DS_RESIDUAL_DTYPE={rd!r}: expected f16, f32, f32outer or wide")
        self.residual_dtype = torch.float16 if rd == "f16" else torch.float32
        self.residual_scope = "outer" if rd == "f32outer" else "full"
        # Matrix-core OPERANDS.  "f16": single fp16 operands (every mode above).  "wide" (round 5): every activation stored in fp32 and
        # every product formed from two-term fp16 splits of both operands (csrc/wide.hip: ds_gemm_wide, fp32 norms / attention) -- an
        # fp32 evaluation of the UNet, eps within ~1e-5 of the reference instead of ~1e-3, at ~3x the matrix-core work.  Meant for single
        # steps: forward(..., precision="wide") routes one evaluation through a twin of this module that shares its parameters and
        # keeps its own packed buffer (hi + lo planes) and C handle; the pipelines choose the steps (operand_policy).  C program only.
        self.operand_mode = "wide" if rd == "wide" else "f16"
        self._twins = {}                     # operand_mode -> twin module (shared parameters, own handle / packed buffer)
        self._handle = None
        # Instrumentation of the C launch program (ds_unet_set_hooks; diagnostics and measurement, eager launches only):
        #   _tap(name, rows [M, C] tensor, (B, T, H, W))   after every block (a copy of the block's output rows)
        #   launch_hook(phase, kernel, flops, info)        on the host right before (phase 0) / after (phase 1) every kernel-family call
        self._tap = None
        self.launch_hook = None
        self._generation = 0                 # bumped by every prepare(): identifies the packed buffers (hipGraph cache keys)
        self._prepare_lock = threading.Lock()
        # a repack (mode change, .to(), load_state_dict) may run while forwards on other host threads are still inside
        # ds_unet_forward with the previous handle: a handle is destroyed only when no call uses it any more
        self._handle_lock = threading.Lock()
        self._handle_uses = {}          # handle value -> calls in flight
        self._handle_retired = {}       # handle value -> handle, replaced while in use

    # ------------------------------------------------------------------ copies / pickles never share a C handle
    _RUNTIME_STATE = ("_handle", "_packed", "_packed_buf", "_twins", "_prepare_lock", "_handle_lock", "_handle_uses", "_handle_retired",
                      "_ws_bytes", "_tap")

    def __getstate__(self):
        """copy.copy / copy.deepcopy / torch.save of the module: the C handle, the packed buffers, the twins and the locks belong to ONE
        module object (two owners of a handle would destroy it twice); the copy repacks on its first forward."""
        st = self.__dict__.copy()
        for k in self._RUNTIME_STATE:
            st.pop(k, None)
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)
        self._handle, self._packed, self._packed_buf, self._tap = None, None, None, None
        self._twins, self._ws_bytes = {}, {}
        self._prepare_lock, self._handle_lock = threading.Lock(), threading.Lock()
        self._handle_uses, self._handle_retired = {}, {}

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, *a, **k):
        self.invalidate()
        return super().load_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        self.invalidate()
        return super()._apply(fn, *a, **k)

    def invalidate(self):
        """Call after mutating parameters in place."""
        self._packed = None
        for tw in getattr(self, "_twins", {}).values():      # they share the parameters
            tw._packed = None

    def _wide(self):
        if self.operand_mode not in ("f16", "wide"):
            raise ValueError(f"operand_mode={self.operand_mode!r}: expected 'f16' or 'wide'")
        return self.operand_mode == "wide"

    def twin(self, operand_mode="wide"):
        """A module over the SAME parameters (and parameter tree) evaluated in another mode, with its own packed buffer and C handle:
        both stay resident, so single evaluations can be routed to it (forward(..., precision=...)).  "wide": split-fp16 operands, fp32
        storage (csrc/wide.hip); "strict" (round 6): single fp16 operands with the WHOLE residual stream and every GroupNorm input in
        fp32 (residual_dtype float32, scope "full": +12 % per evaluation) -- the rung the pipelines' operand policy tries first."""
        if operand_mode not in ("wide", "strict"):
            raise ValueError(f"twin({operand_mode!r}): expected 'wide' or 'strict'")
        tw = self._twins.get(operand_mode)
        if tw is None:
            import copy
            tw = copy.copy(self)                 # shallow: _parameters / _modules (the parameter tree) are shared objects; no handle,
            tw._generation = 0                   # packed buffer, twin or lock travels with a copy (__getstate__ / __setstate__)
            if operand_mode == "wide":
                tw.operand_mode = "wide"
            else:
                tw.operand_mode, tw.residual_dtype, tw.residual_scope = "f16", torch.float32, "full"
            tw.gn_from_producer = False
            self._twins[operand_mode] = tw
        return tw

    def _is_strict_mode(self):
        return self.operand_mode == "f16" and self.residual_dtype == torch.float32 and self.residual_scope == "full"

    @torch.no_grad()
    def prepare(self, device=None, force=False):
        """Repack parameters for the kernels (fp16 GEMM operands, fp32 biases / norm affine).
        The repack kernels run on the caller's current stream; the call returns after a device synchronisation, so any
        stream (the pipelines' side streams, hipGraph captures) may read the packed buffers afterwards.  Serialised by a
        lock: two threads racing into the first forward repack once.  force=True repacks even when packed buffers for the
        device exist (after in-place parameter edits; invalidate() + the next forward does the same)."""
        with self._prepare_lock:
            dev = torch.device(device) if device is not None else next(self.parameters()).device
            if dev.type == "cuda" and dev.index is None:
                dev = torch.device("cuda", torch.cuda.current_device())
            if not force and self._packed is not None and self._device == dev and self._packed_mode == self._mode():
                return self
            return self._prepare_locked(dev)

    def _strict(self):
        return self.residual_dtype == torch.float32 or self._wide()

    def _inner32(self):
        """The stream inside the transformer blocks is fp32 too (residual_scope "full"; always in the wide mode)."""
        if self.residual_scope not in ("full", "outer"):
            raise ValueError(f"residual_scope={self.residual_scope!r}: expected 'full' or 'outer'")
        return self._wide() or (self._strict() and self.residual_scope == "full")

    def _fold(self):
        """LayerNorm fold in effect: the fold multiplies the RAW activation on the matrix cores, which needs it in fp16 -- with an
        fp32 residual stream the LayerNorm kernel (fp32 in, fp16 operand out) runs instead."""
        return False if self._inner32() else self.fold_layernorm

    def _gn_fused(self):
        return bool(self.gn_from_producer and not self.batch_invariant and not self._wide())

    def _mode(self):
        return (self._fold(), self.residual_dtype, self._inner32(), self._gn_fused(), self._wide())

    def _c_config(self):
        """ds_unet_config of this model in the current mode."""
        cfg = self.cfg
        c = _lib.UNetConfig()
        for k in ("in_channels", "out_channels", "model_channels", "num_res_blocks", "transformer_depth",
                  "temporal_transformer_depth", "context_dim"):
            setattr(c, k, int(cfg[k]))
        c.num_head_channels = HEAD_DIM
        cm, ar = list(cfg["channel_mult"]), list(cfg["attention_resolutions"])
        if len(cm) > 8 or len(ar) > 8:
            raise NotImplementedError("more than 8 channel_mult / attention_resolutions entries")
        c.n_channel_mult, c.n_attention_resolutions = len(cm), len(ar)
        for i, v in enumerate(cm):
            c.channel_mult[i] = int(v)
        for i, v in enumerate(ar):
            c.attention_resolutions[i] = int(v)
        for k in ("use_linear", "temporal_conv", "temporal_attention", "addition_attention", "use_image_attention", "fps_cond"):
            setattr(c, k, int(bool(cfg[k])))
        c.residual_f32 = 3 if self._wide() else ((1 if self._inner32() else 2) if self._strict() else 0)
        c.fold_layernorm = int(bool(self._fold()))
        c.gn_from_producer = int(self._gn_fused())
        c.temporal_selfatt_only = int(bool(cfg.get("temporal_selfatt_only", True)))
        return c

    def _release_handle(self):
        lock = getattr(self, "_handle_lock", None)
        if lock is None:                       # __del__ of a half-constructed module
            return
        with lock:
            h, self._handle = getattr(self, "_handle", None), None
            if not h:
                return
            if self._handle_uses.get(h.value, 0) > 0:
                # destroyed by the last call that still uses it -- TOGETHER with its packed buffer: the kernels such a call enqueues
                # read the old packed weights asynchronously, so the buffer must outlive them (released after a device
                # synchronisation in _done_with_handle), not just the host-side handle
                self._handle_retired[h.value] = (h, getattr(self, "_packed_buf", None))
                return
        _lib.load().ds_unet_destroy(h)

    def _use_handle(self):
        """The current C handle, marked in use (pair with _done_with_handle)."""
        with self._handle_lock:
            h = self._handle
            if h:
                self._handle_uses[h.value] = self._handle_uses.get(h.value, 0) + 1
            return h

    def _done_with_handle(self, h):
        with self._handle_lock:
            n = self._handle_uses.get(h.value, 0) - 1
            if n > 0:
                self._handle_uses[h.value] = n
                return
            self._handle_uses.pop(h.value, None)
            dead = self._handle_retired.pop(h.value, None)
        if dead is not None:
            handle, buf = dead
            if buf is not None and buf.is_cuda:
                torch.cuda.synchronize(buf.device)     # the retired call's kernels (any stream) have read the old buffer
            _lib.load().ds_unet_destroy(handle)
            del buf

    def __del__(self):
        try:
            self._release_handle()
        except Exception:
            pass

    def _prepare_locked(self, device):
        """The packing itself is ds_unet_pack (csrc/unet_program.hip): fp16 [N][K] GEMM operands (K = tap*Cin + c for convs),
        fused QKV / KV / image-KV matrices, the GEGLU projection interleaved in 32-row groups [x_g | gate_g], the
        time-embedding projections of all ResBlocks in one matrix (conv-1 bias folded in), LayerNorm folded into the projection
        it feeds (fp16(gamma*W), column sums, beta.W + b).  self._packed maps operand names to views of the buffer (inspection, tests)."""
        import ctypes as C
        dev = device
        if dev.type != "cuda":
            raise RuntimeError("UNetModel runs on an MI355X only (no CPU path): move the model / inputs to a HIP device")
        lib = _lib.load()
        self._release_handle()
        cc = self._c_config()
        h = C.c_void_p()
        _lib.check(lib.ds_unet_create(C.byref(cc), C.byref(h)), "ds_unet_create")
        self._handle = h
        sd = dict(self.named_parameters())
        n = lib.ds_unet_num_weights(h)
        key, nd, shp = C.c_char_p(), C.c_int(), (C.c_int64 * 5)()
        keep = []                      # device copies of the raw tensors: alive until the packing has run
        with torch.cuda.device(dev):
            for i in range(n):
                _lib.check(lib.ds_unet_weight_info(h, i, C.byref(key), C.byref(nd), shp), "ds_unet_weight_info")
                name = key.value.decode()
                if name not in sd:
                    raise KeyError(f"parameter {name} expected by the UNet program is missing from the module")
                t = sd[name].detach()
                if tuple(t.shape) != tuple(shp[:nd.value]):
                    raise ValueError(f"parameter {name}: shape {tuple(t.shape)} != {tuple(shp[:nd.value])}")
                if t.dtype not in (torch.float32, torch.float16):
                    t = t.float()
                t = t.to(dev).contiguous()
                keep.append(t)
                sh = (C.c_int64 * 5)(*list(t.shape))
                _lib.check(lib.ds_unet_load_weight(h, name.encode(), t.data_ptr(), ops._DT[t.dtype], sh, t.dim()), "ds_unet_load_weight")
            if n != len(sd):
                raise KeyError(f"the module has {len(sd)} parameters, the UNet program expects {n}")
            nbytes = lib.ds_unet_packed_bytes(h)
            buf = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
            _lib.check(lib.ds_unet_pack(h, buf.data_ptr(), nbytes, torch.cuda.current_stream(dev).cuda_stream), "ds_unet_pack")
            torch.cuda.synchronize(dev)      # the packed buffer is complete before any other stream can see it
        del keep
        P = {}
        off, nb, rows, dt = C.c_size_t(), C.c_size_t(), C.c_long(), C.c_int()
        for i in range(lib.ds_unet_num_packed(h)):
            _lib.check(lib.ds_unet_packed_info(h, i, C.byref(key), C.byref(off), C.byref(nb), C.byref(rows), C.byref(dt)), "ds_unet_packed_info")
            raw = buf[off.value:off.value + nb.value]
            if dt.value == _lib.DS_F16:
                P[key.value.decode()] = raw.view(torch.float16).view(rows.value, -1)
            else:
                P[key.value.decode()] = raw.view(torch.float32)
        self._packed_buf = buf
        self._ws_bytes = {}
        self._device = dev
        self._generation += 1
        self._packed_mode = self._mode()
        self._packed = P
        return self

    # ------------------------------------------------------------------ forward
    def forward(self, x, timesteps, context=None, features_adapter=None, fps=16, timestep_cond=None, **kwargs):
        """x [b,C,t,h,w] (fp16|fp32, HIP device), timesteps int64 [b], context [b,L,context_dim].
        Returns eps [b,C_out,t,h,w] fp32 (the reference UNet computes and returns fp32).

        cfg_pairs=n (extension): the batch is [x_1..x_n | x_1..x_n] with the same timesteps / fps in both halves and only
        the CONTEXT differing (classifier-free guidance: cond | uncond of the same tiles).  Everything up to the first
        cross-attention -- conv_in, init_attn, the first ResBlock, and GroupNorm / proj_in / self-attention of the first
        SpatialTransformer -- never sees the context, so it is evaluated once on n items and duplicated.  Same kernels on
        the same numbers: the result is bit-identical to the plain 2n forward (a batch equals its separate forwards)."""
        pairs = kwargs.pop("cfg_pairs", None)
        precision = kwargs.pop("precision", None)          # "wide": this evaluation with split-fp16 operands and fp32 storage (twin)
        if precision not in (None, "f16", "wide", "strict"):
            raise ValueError(f"precision={precision!r}: expected None, 'f16', 'strict' or 'wide'")
        if (precision == "wide" and not self._wide()) or (precision == "strict" and not self._wide() and not self._is_strict_mode()):
            if pairs:
                kwargs["cfg_pairs"] = pairs
            return self.twin(precision).forward(x, timesteps, context=context, features_adapter=features_adapter, fps=fps,
                                                timestep_cond=timestep_cond, **kwargs)
        if features_adapter is not None or timestep_cond is not None:
            raise NotImplementedError("features_adapter / timestep_cond are not used by the DynamicScaler pipelines")
        if not x.is_cuda:
            raise RuntimeError("UNetModel.forward: input is on the CPU; this build has no CPU path (the HIP kernels are the product)")
        if self._gn_fused() and not _lib.load().ds_gemm_has_stats():
            raise RuntimeError("gn_from_producer / DS_GN_FROM_PRODUCER=1 needs a library built with DS_GEMM_STATS: "
                               "`python -m dynamicscaler_amd.build --variant gemmstats` and DS_HIP_LIBRARY=.../libdynscaler_hip_gemmstats.so")
        if self._packed is None or self._device != x.device or self._packed_mode != self._mode():
            self.prepare(x.device)
        B = x.shape[0]
        dev = x.device
        if x.dtype not in ops._DT:
            x = x.float()
        x = x.contiguous()
        timesteps = timesteps.to(dev, torch.int64).reshape(-1)
        if timesteps.numel() == 1 and B > 1:
            timesteps = timesteps.expand(B)
        if pairs and (2 * pairs != B or not any(b.kind == "st" for g in self._inputs for b in g)):
            raise ValueError(f"cfg_pairs={pairs} needs a batch of {2 * pairs} (got {B}) and a SpatialTransformer in the input path")
        if not isinstance(fps, int):             # the reference passes a python int or a [b] tensor of one value (ddpm3d.py:710)
            f = torch.as_tensor(fps).reshape(-1)
            if f.numel() == 0 or bool((f != f[0]).any()):
                raise NotImplementedError("per-item fps values: the DynamicScaler pipelines condition every evaluation of a call on one fps")
            fps = int(f[0])
        return self._forward_c(x, timesteps.contiguous(), context, fps, int(pairs or 0))

    def _forward_c(self, x, timesteps, context, fps, pairs):
        """One ds_unet_forward call: the launch loop runs in C++ (csrc/unet_program.hip) on the caller's current stream, all scratch
        from one workspace tensor sized by ds_unet_workspace_bytes (under hipGraph capture it lives in the graph's pool)."""
        lib = _lib.load()
        dev = x.device
        B, Cin, T, H, W = x.shape
        context = context.to(dev)
        if context.dtype not in ops._DT:
            context = context.float()
        context = context.contiguous()
        L = context.shape[1]
        h = self._use_handle()                # stays valid for this call even if another thread repacks meanwhile
        if not h:
            raise RuntimeError("UNetModel: no prepared C handle (prepare() has not run)")
        try:
            key = (h.value, B, T, H, W, L, pairs)
            nbytes = self._ws_bytes.get(key)
            if nbytes is None:
                nbytes = self._ws_bytes[key] = lib.ds_unet_workspace_bytes(h, B, T, H, W, L, pairs)
                if nbytes == 0:
                    _lib.check(-1, "ds_unet_workspace_bytes")
            ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
            eps = torch.empty((B, self.cfg["out_channels"], T, H, W), dtype=torch.float32, device=dev)
            hooks = self._c_hooks(lib, dev)
            if hooks is not None:
                _lib.check(lib.ds_unet_set_hooks(h, hooks[0], hooks[1], None), "ds_unet_set_hooks")
            try:
                _lib.check(lib.ds_unet_forward(h, x.data_ptr(), ops._DT[x.dtype], timesteps.data_ptr(), context.data_ptr(),
                                               ops._DT[context.dtype], L, int(fps), B, T, H, W, pairs, ws.data_ptr(), nbytes, eps.data_ptr(),
                                               torch.cuda.current_stream(dev).cuda_stream), "ds_unet_forward")
            finally:
                if hooks is not None:
                    lib.ds_unet_set_hooks(h, None, None, None)
                    if hooks[2]:
                        raise hooks[2][0]
        finally:
            self._done_with_handle(h)
        return eps

    def _c_hooks(self, lib, dev):
        """ctypes callbacks for ds_unet_set_hooks from `launch_hook` / `_tap` (None when neither is set).  An exception raised inside a
        callback cannot cross the C frames: it is kept and re-raised after the forward."""
        import ctypes as C
        launch_py = self.launch_hook
        tap_py = self._tap
        if launch_py is None and tap_py is None:
            return None
        errors = []

        def launch_cb(user, phase, kernel, flops, info, n_info, stream):
            try:
                launch_py(int(phase), kernel.decode(), float(flops), tuple(info[i] for i in range(n_info)))
            except BaseException as e:      # noqa: BLE001
                errors.append(e)

        def tap_cb(user, block, rows, nrows, cols, ld, dtype, B, T, H, W, stream):
            try:
                dt = torch.float32 if dtype == _lib.DS_F32 else torch.float16
                t = torch.empty((nrows, cols), dtype=dt, device=dev)
                esz = t.element_size()
                _lib.check(lib.ds_copy_rows(t.data_ptr(), cols * esz, rows, ld * esz, cols * esz, nrows, stream), "ds_copy_rows")
                tap_py(block.decode(), t, (B, T, H, W))
            except BaseException as e:      # noqa: BLE001
                errors.append(e)

        return (_lib.LAUNCH_HOOK(launch_cb) if launch_py is not None else None, _lib.BLOCK_TAP(tap_cb) if tap_py is not None else None, errors)

    def c_program_trace(self, B, T, H, W, ctx_tokens, cfg_pairs=0):
        """The C program's launch sequence for this geometry as a list of text lines (ds_unet_trace; no GPU needed)."""
        import ctypes as C
        lib = _lib.load()
        own = None
        h = self._handle
        if h is None:                       # not prepared (e.g. on a machine without a GPU): a handle just for the trace
            own = C.c_void_p()
            cc = self._c_config()
            _lib.check(lib.ds_unet_create(C.byref(cc), C.byref(own)), "ds_unet_create")
            h = own
        try:
            n = lib.ds_unet_trace(h, B, T, H, W, ctx_tokens, cfg_pairs, None, 0)
            if n < 0:
                _lib.check(int(n), "ds_unet_trace")
            buf = C.create_string_buffer(n + 1)
            lib.ds_unet_trace(h, B, T, H, W, ctx_tokens, cfg_pairs, buf, n + 1)
            return buf.value.decode().splitlines()
        finally:
            if own is not None:
                lib.ds_unet_destroy(own)


class DiffusionWrapper(nn.Module):
    """lvdm/models/ddpm3d.py:696-763, conditioning_key='crossattn' branch (:710-712)."""

    def __init__(self, diffusion_model, conditioning_key="crossattn"):
        super().__init__()
        self.diffusion_model = diffusion_model
        self.conditioning_key = conditioning_key

    def forward(self, x, t, c_concat=None, c_crossattn=None, c_adm=None, s=None, mask=None, **kwargs):
        if self.conditioning_key != "crossattn":
            raise NotImplementedError("only conditioning_key='crossattn' is on the DynamicScaler hot path")
        cc = torch.cat(c_crossattn, 1)
        return self.diffusion_model(x, t, context=cc, **kwargs)
