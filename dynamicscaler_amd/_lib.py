"""ctypes binding of libdynscaler_hip.so (the C ABI declared in include/dynscaler_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a symbol is absent this module
raises at import of the first op (HipLibraryMissing), loudly.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# The product loads the in-tree library.  DS_HIP_LIBRARY names a diagnostic build of the same ABI instead (tests/hazard_probe.py,
# A/B runs); load() says so on stderr, so a leftover variable cannot silently swap the kernels under a test or a bench.
IN_TREE_LIB = os.path.join(HERE, "libdynscaler_hip.so")
LIB_PATH = os.environ.get("DS_HIP_LIBRARY") or IN_TREE_LIB

DS_F16, DS_F32 = 0, 1
DS_A_DENSE, DS_A_CONV3, DS_A_TCONV = 0, 1, 2
DS_EPI_GEGLU, DS_EPI_SILU, DS_EPI_OUT_F32, DS_EPI_RES_F32 = 1, 2, 4, 8
DS_MAX_WINDOWS = 64
ABI_VERSION = 3


class HipLibraryMissing(RuntimeError):
    pass


class DsError(RuntimeError):
    pass


class RingGeom(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("C", "F", "H", "W", "tf", "th", "tw", "dtype")]


class GemmDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "M", "N", "K", "a_mode", "lda", "cin", "nimg", "hin", "win", "hout", "wout", "stride", "upsample",
        "t_len", "hw", "ldc", "ldr", "bias_rows", "ldbias", "epilogue", "asym_pad")]


class UNetConfig(C.Structure):
    """ds_unet_config (include/dynscaler_hip.h)."""
    _fields_ = ([(n, C.c_int32) for n in ("in_channels", "out_channels", "model_channels", "num_res_blocks", "n_channel_mult")]
                + [("channel_mult", C.c_int32 * 8), ("n_attention_resolutions", C.c_int32), ("attention_resolutions", C.c_int32 * 8)]
                + [(n, C.c_int32) for n in ("num_head_channels", "transformer_depth", "temporal_transformer_depth", "context_dim",
                                            "use_linear", "temporal_conv", "temporal_attention", "addition_attention",
                                            "use_image_attention", "fps_cond", "residual_f32", "fold_layernorm", "gn_from_producer", "temporal_selfatt_only")])


_vp, _i, _f, _u64, _sz = C.c_void_p, C.c_int, C.c_float, C.c_uint64, C.c_size_t
_pp = C.POINTER
# ds_launch_hook / ds_block_tap (instrumentation of ds_unet_forward)
LAUNCH_HOOK = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_char_p, C.c_double, C.POINTER(C.c_int32), C.c_int, C.c_void_p)
BLOCK_TAP = C.CFUNCTYPE(None, C.c_void_p, C.c_char_p, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p)

# name -> (restype, argtypes); mirrors include/dynscaler_hip.h one to one
SIGNATURES = {
    "ds_last_error": (C.c_char_p, []),
    "ds_abi_version": (_i, []),
    "ds_ring_gather": (_i, [_vp, _vp, _vp, _vp, C.POINTER(RingGeom), C.POINTER(C.c_int32), _i, _vp]),
    "ds_ring_scatter3": (_i, [_vp, _vp, _vp, _vp, _vp, C.POINTER(RingGeom), C.POINTER(C.c_int32), _i, _vp]),
    "ds_ring_gather_renoise": (_i, [_vp, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _i, _u64, _pp(C.c_int64), C.POINTER(RingGeom), C.POINTER(C.c_int32), _i, _vp]),
    "ds_cfg_ddim_scatter": (_i, [_vp, _vp, _vp, _i, _f, _f, _f, _f, _f, _f, _vp, _vp, _vp, _vp, C.POINTER(RingGeom), C.POINTER(C.c_int32), _i, _vp]),
    "ds_renoise_mix": (_i, [_vp, _vp, _vp, _f, _f, _f, _f, _i, _u64, _u64, C.POINTER(RingGeom), _i, _vp]),
    "ds_cfg_ddim": (_i, [_vp, _vp, _vp, _i, _f, _f, _f, _f, _f, _f, _vp, _vp, _vp, C.POINTER(RingGeom), _i, _vp]),
    "ds_map_gather": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ds_map_scatter3": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ds_map_gather_frames": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "ds_map_scatter3_frames": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "ds_map_splat": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ds_map_gather_taps": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "ds_residual_merge": (_i, [_vp, _vp, _vp, _i, C.c_long, _i, _i, _f, _f, _i, _i, _vp]),
    "ds_resize_latent": (_i, [_vp, _vp, _i, C.c_long, _i, _i, _i, _i, _i, _vp]),
    "ds_gemm_f16": (_i, [_vp, _vp, _vp, _vp, _vp, C.POINTER(GemmDesc), _vp]),
    "ds_groupnorm_stats_workspace_floats": (_sz, [_i, _i, _i]),
    "ds_groupnorm_chunk_rows": (_i, [_i, _i]),
    "ds_groupnorm_stats": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "ds_groupnorm_apply": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ds_groupnorm_f16": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp]),
    "ds_groupnorm_f16_strided": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp]),
    "ds_groupnorm_rows": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp]),
    "ds_layernorm_rows": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "ds_cast_rows_f32_f16": (_i, [_vp, _i, _vp, _i, C.c_long, _i, _vp]),
    "ds_layernorm": (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "ds_layernorm_stats": (_i, [_vp, _vp, _i, _i, _f, _vp]),
    "ds_gemm_has_stats": (_i, []),
    "ds_set_launch_share": (_i, [_i]),
    "ds_gemm_f16_stats": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _pp(GemmDesc), _vp]),
    "ds_groupnorm_onepass_applies": (_i, [_i, _i, _i, _i]),
    "ds_groupnorm_rows_onepass": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp]),
    "ds_groupnorm_rows_colstats": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp]),
    "ds_gemm_f16_ln": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(GemmDesc), _vp]),
    "ds_gemm_f16_lnk": (_i, [_vp, _vp, _f, _vp, _vp, _vp, C.POINTER(GemmDesc), _vp]),
    "ds_attention_f16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "ds_temporal_attention_f16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _vp]),
    "ds_concat_channels": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "ds_im2col_in": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ds_im2col_in_affine": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _f, _vp]),
    "ds_softmax_rows": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "ds_posterior_sample": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp]),
    "ds_rows_to_ncthw": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ds_timestep_embedding": (_i, [_vp, _vp, _i, _i, _vp]),
    "ds_silu_f16": (_i, [_vp, _vp, _sz, _vp]),
    "ds_attention_enc_f16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "ds_gelu_f16": (_i, [_vp, _vp, _sz, _vp]),
    "ds_embed_tokens": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ds_vit_assemble": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "ds_clip_preprocess": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "ds_patchify": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ds_cast_to_f16": (_i, [_vp, _i, _vp, _sz, _vp]),
    "ds_wide_lo_scale": (_f, []),
    "ds_split_f16": (_i, [_vp, _i, _vp, _vp, _sz, _vp]),
    "ds_gemm_wide": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(GemmDesc), _vp]),
    "ds_groupnorm_wide_scratch_floats": (_sz, [_i, _i, _i]),
    "ds_groupnorm_wide": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp]),
    "ds_layernorm_wide": (_i, [_vp, _vp, _vp, _vp, C.c_long, _i, _f, _vp]),
    "ds_attention_wide": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "ds_temporal_attention_wide": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _vp]),
    "ds_timestep_embedding_f32": (_i, [_vp, _vp, _i, _i, _vp]),
    "ds_silu_f32": (_i, [_vp, _vp, _sz, _vp]),
    "ds_cast_to_f32": (_i, [_vp, _i, _vp, _sz, _vp]),
    "ds_im2col_in_f32": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ds_im2col_in_affine_f32": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _f, _vp]),
    "ds_softmax_rows_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "ds_unet_create": (_i, [_pp(UNetConfig), _pp(_vp)]),
    "ds_unet_destroy": (_i, [_vp]),
    "ds_unet_num_weights": (_i, [_vp]),
    "ds_unet_weight_info": (_i, [_vp, _i, _pp(C.c_char_p), _pp(_i), _pp(C.c_int64)]),
    "ds_unet_load_weight": (_i, [_vp, C.c_char_p, _vp, _i, _pp(C.c_int64), _i]),
    "ds_unet_packed_bytes": (_sz, [_vp]),
    "ds_unet_pack": (_i, [_vp, _vp, _sz, _vp]),
    "ds_unet_num_packed": (_i, [_vp]),
    "ds_unet_packed_info": (_i, [_vp, _i, _pp(C.c_char_p), _pp(_sz), _pp(_sz), _pp(C.c_long), _pp(_i)]),
    "ds_unet_emb_offset": (_i, [_vp, C.c_char_p]),
    "ds_unet_workspace_bytes": (_sz, [_vp, _i, _i, _i, _i, _i, _i]),
    "ds_unet_forward": (_i, [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp, _vp]),
    "ds_unet_trace": (C.c_long, [_vp, _i, _i, _i, _i, _i, _i, C.c_char_p, _sz]),
    "ds_unet_set_hooks": (_i, [_vp, _vp, _vp, _vp]),
    "ds_copy_rows": (_i, [_vp, _sz, _vp, _sz, _sz, _sz, _vp]),
}

_lib = None


def load():
    """dlopen the library once and bind every symbol of the ABI; raise HipLibraryMissing otherwise."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            f"{LIB_PATH} not found: build it with `python -m dynamicscaler_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback for the DynamicScaler hot path.")
    if os.path.abspath(LIB_PATH) != os.path.abspath(IN_TREE_LIB):
        import sys
        sys.stderr.write(f"[dynamicscaler_amd] DS_HIP_LIBRARY is set: loading {LIB_PATH} instead of the in-tree library\n")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:
        raise HipLibraryMissing(f"cannot load {LIB_PATH}: {e}") from e
    # the version first: a stale prebuilt library may export every name and still disagree on a struct layout
    try:
        lib.ds_abi_version.restype, lib.ds_abi_version.argtypes = _i, []
        have = lib.ds_abi_version()
    except AttributeError as e:
        raise HipLibraryMissing(f"{LIB_PATH} does not export ds_abi_version; rebuild it") from e
    if have != ABI_VERSION:
        raise HipLibraryMissing(f"ABI version mismatch: library {have} != binding {ABI_VERSION} (include/dynscaler_hip.h DS_ABI_VERSION); "
                                "rebuild it with `python -m dynamicscaler_amd.build --force`")
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryMissing(f"{LIB_PATH} does not export {name}; rebuild it") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


DIAG_LIB = os.path.join(HERE, "libdynscaler_diag.so")


def load_diag():
    """The diagnostics library (csrc/diag.hip: ds_dbg_poison_cu_state) -- tests / tools only, never the product path."""
    if not os.path.exists(DIAG_LIB):
        raise HipLibraryMissing(f"{DIAG_LIB} not found: build it with `python -m dynamicscaler_amd.build`")
    lib = C.CDLL(DIAG_LIB)
    lib.ds_dbg_poison_cu_state.restype, lib.ds_dbg_poison_cu_state.argtypes = _i, [_vp]
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().ds_last_error()
        raise DsError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")
