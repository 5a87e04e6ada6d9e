"""Launch traces of the UNet's Python program without a GPU.

`python_program_trace(model, B, T, H, W, ctx_tokens, cfg_pairs)` runs UNetModel.forward (the Python restatement of the launch
program) on shape-only "meta" tensors with every op of `ops` replaced by a recorder that appends one line per kernel call in
the format of ds_unet_trace (csrc/unet_program.hip) and returns an output of the right shape / dtype / strides.  The packed
operands are meta views shaped from the C handle's layout (ds_unet_packed_info), so no weights are needed either.
tests/test_host_cpu.py compares the result with UNetModel.c_program_trace line by line: the two programs issue the same
kernel calls with the same descriptors.
"""
import ctypes as C

import torch

from . import _lib, ops
from ._lib import DS_A_DENSE, DS_EPI_GEGLU, DS_EPI_OUT_F32, DS_EPI_RES_F32

_DT = {torch.float16: 0, torch.float32: 1}


def _meta(shape, dtype):
    return torch.empty(shape, dtype=dtype, device="meta")


class Recorder:
    def __init__(self):
        self.lines = []

    # ---- the ops UNetModel.forward calls (signatures of ops.py) ----
    def gemm(self, A, W, bias=None, residual=None, *, M, N, K, out=None, a_mode=DS_A_DENSE, lda=None, cin=None, conv=None,
             tconv=None, bias_rows=None, ldbias=None, epilogue=0, stream=None, colstats=None):
        n_out = N // 2 if (epilogue & DS_EPI_GEGLU) else N
        if residual is not None and residual.dtype == torch.float32:
            epilogue |= DS_EPI_RES_F32
        if out is not None and out.dtype == torch.float32:
            epilogue |= DS_EPI_OUT_F32
        if out is None:
            out = _meta((M, n_out), torch.float32 if (epilogue & DS_EPI_OUT_F32) else torch.float16)
        cin_ = K if cin is None else cin
        lda_ = cin_ if lda is None else lda
        cv = list(conv[:7]) if conv is not None else [0] * 7
        tc = list(tconv) if tconv is not None else [0, 0]
        self.lines.append("gemm M=%d N=%d K=%d mode=%d cin=%d lda=%d ldc=%d ldr=%d brows=%d ldb=%d epi=%d conv=%d,%d,%d,%d,%d,%d,%d t=%d,%d bias=%d res=%d stats=%d" % (
            M, N, K, a_mode, cin_, lda_, out.stride(0), residual.stride(0) if residual is not None else 0,
            0x7FFFFFFF if bias_rows is None else bias_rows, N if ldbias is None else ldbias, epilogue, *cv, *tc,
            int(bias is not None), int(residual is not None), colstats.stride(0) // 2 if colstats is not None else 0))
        return out

    def colstats_table(self, rows, cols, device):
        return _meta(((rows + 31) // 32, cols, 2), torch.float32)

    def gemm_ln(self, x, Wg, stats, colsum, colbias=None, *, M, N, K, out=None, epilogue=0, stream=None, eps=1e-5):
        n_out = N // 2 if (epilogue & DS_EPI_GEGLU) else N
        out = _meta((M, n_out), torch.float16)
        self.lines.append("gemm_ln M=%d N=%d K=%d lda=%d ldc=%d epi=%d" % (M, N, K, x.stride(0), out.stride(0), epilogue))
        return out

    def groupnorm(self, x, gamma, beta, ninst, rows_per_inst, Cch, eps, silu, groups=32, stream=None, raw_f16=False, colstats=None):
        self.lines.append("groupnorm xdt=%d ldx=%d ninst=%d rows=%d C=%d silu=%d raw=%d eps=%g stats=%d" % (
            _DT[x.dtype], x.stride(0), ninst, rows_per_inst, Cch, int(bool(silu)), int(bool(raw_f16)), eps,
            colstats.stride(0) // 2 if colstats is not None else 0))
        y = _meta((x.shape[0], Cch), torch.float16)
        return (y, _meta((x.shape[0], Cch), torch.float16)) if raw_f16 else y

    def layernorm(self, x, gamma, beta, eps=1e-5, stream=None, out=None):
        self.lines.append("layernorm xdt=%d rows=%d C=%d" % (_DT[x.dtype], x.shape[0], x.shape[1]))
        return _meta(tuple(x.shape), torch.float16)

    def layernorm_stats(self, x, eps=1e-5, stream=None):
        self.lines.append("layernorm_stats rows=%d C=%d" % (x.shape[0], x.shape[1]))
        return _meta((x.shape[0], 2), torch.float32)

    def cast_rows_f16(self, x, stream=None):
        self.lines.append("cast_rows rows=%d C=%d ldx=%d" % (x.shape[0], x.shape[1], x.stride(0)))
        return _meta(tuple(x.shape), torch.float16)

    def attention(self, q, k, v, out, *, batch, heads, nq, nk, ldq, ldk, ldv, ldo, kv_batch_div=1, scale, accumulate=False,
                  stream=None):
        self.lines.append("attention batch=%d heads=%d nq=%d nk=%d ldq=%d ldk=%d ldv=%d ldo=%d kvdiv=%d acc=%d" % (
            batch, heads, nq, nk, ldq, ldk, ldv, ldo, kv_batch_div, int(bool(accumulate))))
        return out

    def temporal_attention(self, q, k, v, out, *, nseq_batches, T, hw, heads, ldq, ldk, ldv, ldo, scale, stream=None):
        self.lines.append("temporal_attention nb=%d T=%d hw=%d heads=%d ldq=%d ldk=%d ldv=%d ldo=%d" % (
            nseq_batches, T, hw, heads, ldq, ldk, ldv, ldo))
        return out

    def im2col_in(self, x, kpad, stream=None):
        B, Cc, T, H, W = x.shape
        self.lines.append("im2col_in B=%d C=%d T=%d H=%d W=%d kpad=%d" % (B, Cc, T, H, W, kpad))
        return _meta((B * T * H * W, kpad), torch.float16)

    def rows_to_ncthw(self, y, shape, out_dtype, stream=None):
        B, Cc, T, H, W = shape
        self.lines.append("rows_to_ncthw ydt=%d ldy=%d B=%d C=%d T=%d H=%d W=%d" % (_DT[y.dtype], y.stride(0), B, Cc, T, H, W))
        return _meta(tuple(shape), out_dtype)

    def timestep_embedding(self, t, dim, stream=None):
        self.lines.append("timestep_embedding n=%d dim=%d" % (t.shape[0], dim))
        return _meta((t.shape[0], dim), torch.float16)

    def silu(self, x, stream=None):
        self.lines.append("silu n=%d" % x.numel())
        return _meta(tuple(x.shape), x.dtype)

    def concat_channels(self, a, b, stream=None):
        self.lines.append("concat rows=%d c1=%d c2=%d" % (a.shape[0], a.shape[1], b.shape[1]))
        return _meta((a.shape[0], a.shape[1] + b.shape[1]), a.dtype)


_PATCHED = ("gemm", "gemm_ln", "colstats_table", "groupnorm", "layernorm", "layernorm_stats", "cast_rows_f16", "attention", "temporal_attention",
            "im2col_in", "rows_to_ncthw", "timestep_embedding", "silu", "concat_channels")


def _meta_operands(model):
    """name -> meta tensor of every packed operand, shaped from the C handle's layout (nothing is packed or allocated)."""
    lib = _lib.load()
    h = C.c_void_p()
    cc = model._c_config()
    _lib.check(lib.ds_unet_create(C.byref(cc), C.byref(h)), "ds_unet_create")
    try:
        P, emb_off = {}, {}
        key, off, nb, rows, dt = C.c_char_p(), C.c_size_t(), C.c_size_t(), C.c_long(), C.c_int()
        for i in range(lib.ds_unet_num_packed(h)):
            _lib.check(lib.ds_unet_packed_info(h, i, C.byref(key), C.byref(off), C.byref(nb), C.byref(rows), C.byref(dt)), "ds_unet_packed_info")
            if dt.value == _lib.DS_F16:
                P[key.value.decode()] = _meta((rows.value, nb.value // 2 // rows.value), torch.float16)
            else:
                P[key.value.decode()] = _meta((nb.value // 4,), torch.float32)
        for g in list(model._inputs) + [model._middle] + list(model._outputs):
            for b in g:
                if b.kind == "res":
                    emb_off[b.prefix] = lib.ds_unet_emb_offset(h, b.prefix.encode())
        return P, emb_off
    finally:
        lib.ds_unet_destroy(h)


_KERNEL_LINES = ("gemm ", "gemm_ln ", "groupnorm ", "layernorm ", "layernorm_stats ", "cast_rows ", "attention ", "temporal_attention ",
                 "im2col_in ", "rows_to_ncthw ", "timestep_embedding ", "silu ")


def kernel_lines(lines):
    """The kernel launches of a trace (the C program's `copy` / `cast` lines are device copies torch does on the Python side)."""
    return [ln for ln in lines if ln.startswith(_KERNEL_LINES)]


def python_program_trace(model, B, T, H, W, ctx_tokens, cfg_pairs=0):
    rec = Recorder()
    saved = {n: getattr(ops, n) for n in _PATCHED}
    state = {k: getattr(model, k, None) for k in ("_packed", "_emb_off", "_emb_total", "_kpad_in", "_packed_mode", "_device", "program")}
    try:
        for n in _PATCHED:
            setattr(ops, n, getattr(rec, n))
        P, emb_off = _meta_operands(model)
        model._packed, model._emb_off = P, emb_off
        model._emb_total = P["emb_all.w"].shape[0]
        model._kpad_in = P["input_blocks.0.0.w"].shape[1]
        model.program = "python"
        cfg = model.cfg
        x = _meta((B, cfg["in_channels"], T, H, W), torch.float16)
        ts = _meta((B,), torch.int64)
        ctx = _meta((B, ctx_tokens, cfg["context_dim"]), torch.float32)
        model.forward(x, ts, context=ctx, fps=8, **({"cfg_pairs": cfg_pairs} if cfg_pairs else {}))
    finally:
        for n, f in saved.items():
            setattr(ops, n, f)
        for k, v in state.items():
            setattr(model, k, v)
    return rec.lines
