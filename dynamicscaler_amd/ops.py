"""Thin torch-tensor front end over the C ABI (include/dynscaler_hip.h).

torch is used for device memory and streams only; every computation below is a HIP kernel of
libdynscaler_hip.so.  All launches go to torch's current stream (so torch.cuda.Event timing and graph
capture see them).  There is no CPU path: tensors must live on a HIP device.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import (DS_F16, DS_F32, DS_A_DENSE, DS_A_CONV3, DS_A_TCONV, DS_EPI_GEGLU, DS_EPI_SILU,
                   DS_EPI_OUT_F32, DS_EPI_RES_F32, DS_MAX_WINDOWS, RingGeom, GemmDesc, check)

_DT = {torch.float16: DS_F16, torch.float32: DS_F32}

# optional per-launch timing hook used by bench.py: called as hook(name, flops, launch_callable)
_timing_hook = None


def set_timing_hook(hook):
    global _timing_hook
    _timing_hook = hook


def _timed(name, nbytes, launch, info):
    """Tile-op launches under the timing hook (bench.py's roofline.tile_ops): algorithmic bytes instead of FLOPs."""
    if _timing_hook is not None:
        _timing_hook(name, 0.0, launch, ("bytes", int(nbytes)) + tuple(info))
    else:
        launch()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _dev(t, what):
    if not t.is_cuda:
        raise _lib.DsError(f"{what}: tensor is on {t.device}; the DynamicScaler hot path has no CPU fallback")
    if not t.is_contiguous():
        raise _lib.DsError(f"{what}: tensor must be contiguous")
    return t


# ------------------------------------------------------------------------------------------------ ring tile ops
def _geom(pano_shape, tile_shape, dtype):
    _, Cc, F, H, W = pano_shape
    tf, th, tw = tile_shape
    return RingGeom(Cc, F, H, W, tf, th, tw, _DT[dtype])


def _origins(origins):
    n = len(origins)
    if not 1 <= n <= DS_MAX_WINDOWS:
        raise _lib.DsError(f"need 1..{DS_MAX_WINDOWS} windows per launch, got {n}")
    arr = (C.c_int32 * (3 * n))()
    for i, (f0, y0, x0) in enumerate(origins):
        arr[3 * i], arr[3 * i + 1], arr[3 * i + 2] = int(f0), int(y0), int(x0)
    return arr, n


def ring_gather(pano, origins, tile_fhw, mask_pano=None):
    """pano [1,C,F,H,W] (fp16/fp32); origins [(f0,y0,x0)]; -> tiles [n,C,tf,th,tw] (+ mask tiles u8 [n,tf,th,tw])."""
    _dev(pano, "ring_gather")
    lib = _lib.load()
    arr, n = _origins(origins)
    tf, th, tw = tile_fhw
    g = _geom(pano.shape, tile_fhw, pano.dtype)
    tiles = torch.empty((n, pano.shape[1], tf, th, tw), dtype=pano.dtype, device=pano.device)
    mtiles = None
    if mask_pano is not None:
        _dev(mask_pano, "ring_gather(mask)")
        assert mask_pano.dtype == torch.uint8 and tuple(mask_pano.shape) == tuple(pano.shape[2:])
        mtiles = torch.empty((n, tf, th, tw), dtype=torch.uint8, device=pano.device)
    st = _stream()
    _timed("ring_gather", 2 * (tiles.numel() * tiles.element_size() + (mtiles.numel() if mtiles is not None else 0)),
           lambda: check(lib.ds_ring_gather(pano.data_ptr(), _ptr(mask_pano), tiles.data_ptr(), _ptr(mtiles), C.byref(g), arr, n, st),
                         "ds_ring_gather"), (n,))
    return tiles, mtiles


def ring_gather_mask(mask_pano, pano_shape, origins, tile_fhw):
    _dev(mask_pano, "ring_gather_mask")
    lib = _lib.load()
    arr, n = _origins(origins)
    tf, th, tw = tile_fhw
    g = _geom(pano_shape, tile_fhw, torch.float16)
    mtiles = torch.empty((n, tf, th, tw), dtype=torch.uint8, device=mask_pano.device)
    check(lib.ds_ring_gather(0, mask_pano.data_ptr(), 0, mtiles.data_ptr(), C.byref(g), arr, n, _stream()),
          "ds_ring_gather(mask)")
    return mtiles


def ring_scatter3(pano_latent, pano_x0, mask_pano, x_prev_tiles, x0_tiles, origins):
    ref = pano_latent if pano_latent is not None else pano_x0
    src = x_prev_tiles if x_prev_tiles is not None else x0_tiles
    _dev(ref, "ring_scatter3")
    lib = _lib.load()
    arr, n = _origins(origins)
    tile_fhw = tuple(src.shape[2:])
    # RingLatent.set_window_latent shape assert (shift_window_utils.py:190)
    assert src.shape[0] == n and src.shape[1] == ref.shape[1], \
        f"Input latent shape {tuple(src.shape)} does not match {n} windows of the panorama {tuple(ref.shape)}"
    for t in (x_prev_tiles, x0_tiles):
        if t is not None:
            _dev(t, "ring_scatter3(tile)")
            assert t.dtype == ref.dtype and tuple(t.shape) == tuple(src.shape)
    g = _geom(ref.shape, tile_fhw, ref.dtype)
    st = _stream()
    ntens = (pano_latent is not None) + (pano_x0 is not None)
    _timed("ring_scatter3", 2 * ntens * src.numel() * src.element_size() + (src.numel() // src.shape[1] if mask_pano is not None else 0),
           lambda: check(lib.ds_ring_scatter3(_ptr(pano_latent), _ptr(pano_x0), _ptr(mask_pano), _ptr(x_prev_tiles), _ptr(x0_tiles),
                                              C.byref(g), arr, n, st), "ds_ring_scatter3"), (n,))


def ring_gather_renoise(pano, mask_pano, origins, tile_fhw, c, s, mix_ratio, noise=None, mask_frame0=True, seed=0, tile_offsets=None,
                        want_mask_tiles=False):
    """ring_gather + renoise_mix_ in one kernel (ds_ring_gather_renoise): tiles [n,C,tf,th,tw] = the windows, re-noised under the
    mask panorama u8 [F,H,W].  noise [n,C,tf,th,tw] or None = in-kernel Philox with tile k drawing from counter tile_offsets[k] on
    (the stream of renoise_mix_(..., tile_ids=...)).  -> (tiles, mask tiles | None)."""
    _dev(pano, "ring_gather_renoise")
    _dev(mask_pano, "ring_gather_renoise(mask)")
    lib = _lib.load()
    arr, n = _origins(origins)
    tf, th, tw = tile_fhw
    g = _geom(pano.shape, tile_fhw, pano.dtype)
    assert mask_pano.dtype == torch.uint8 and tuple(mask_pano.shape) == tuple(pano.shape[2:])
    tiles = torch.empty((n, pano.shape[1], tf, th, tw), dtype=pano.dtype, device=pano.device)
    mtiles = torch.empty((n, tf, th, tw), dtype=torch.uint8, device=pano.device) if want_mask_tiles else None
    offs = None
    if noise is None:
        assert tile_offsets is not None and len(tile_offsets) == n
        offs = (C.c_int64 * n)(*[int(v) for v in tile_offsets])
    else:
        _dev(noise, "ring_gather_renoise(noise)")
        assert noise.dtype == pano.dtype and tuple(noise.shape) == tuple(tiles.shape) and noise.is_contiguous()
    st = _stream()
    nb = tiles.numel() * tiles.element_size() * (2 + (noise is not None)) + n * tf * th * tw * (1 + bool(want_mask_tiles))
    _timed("ring_gather_renoise", nb,
           lambda: check(lib.ds_ring_gather_renoise(pano.data_ptr(), mask_pano.data_ptr(), tiles.data_ptr(), _ptr(mtiles), _ptr(noise), float(c),
                                                    float(s), float(mix_ratio), float(1 - mix_ratio), int(bool(mask_frame0)), int(seed), offs,
                                                    C.byref(g), arr, n, st), "ds_ring_gather_renoise"), (n,))
    return tiles, mtiles


def cfg_ddim_scatter_(pano_latent, pano_x0, mask_pano, x, eps_c, eps_u, guidance, coef, origins, noise=None):
    """cfg_ddim + ring_scatter3 in one kernel (ds_cfg_ddim_scatter): the update of the pairwise-disjoint windows `origins` goes straight
    into the panoramas (x_prev -> pano_latent, pred_x0 -> pano_x0, mask_pano <- 1)."""
    _dev(x, "cfg_ddim_scatter")
    _dev(eps_c, "cfg_ddim_scatter(eps_c)")
    _dev(pano_latent, "cfg_ddim_scatter(pano)")
    lib = _lib.load()
    arr, n = _origins(origins)
    assert x.shape[0] == n and eps_c.shape == x.shape and pano_latent.dtype == x.dtype and pano_x0.dtype == x.dtype
    assert x.is_contiguous() and eps_c.is_contiguous() and (eps_u is None or (eps_u.is_contiguous() and eps_u.shape == x.shape and eps_u.dtype == eps_c.dtype))
    g = _geom(pano_latent.shape, tuple(x.shape[2:]), x.dtype)
    st = _stream()
    nb = x.numel() * (3 * x.element_size() + eps_c.element_size() * (1 + (eps_u is not None)) + (x.element_size() if noise is not None else 0)) + \
        (x.numel() // x.shape[1] if mask_pano is not None else 0)
    _timed("cfg_ddim_scatter", nb,
           lambda: check(lib.ds_cfg_ddim_scatter(x.data_ptr(), eps_c.data_ptr(), _ptr(eps_u), _DT[eps_c.dtype], float(guidance),
                                                 float(coef["sqrt_one_minus_at"]), float(coef["sqrt_at"]), float(coef["sqrt_a_prev"]),
                                                 float(coef["dir_coef"]), float(coef["sigma"]), _ptr(noise), pano_latent.data_ptr(),
                                                 pano_x0.data_ptr(), _ptr(mask_pano), C.byref(g), arr, n, st), "ds_cfg_ddim_scatter"), (n,))


def renoise_mix_(tiles, mask_tiles, pano_shape, c, s, mix_ratio, noise=None, mask_frame0=True, seed=0, offset=0,
                 tile_ids=None):
    """In place: tiles <- mix(tiles, c*tiles + s*noise, mask, mix_ratio).
    In-kernel Philox mode (noise None): the launch draws the counters offset, offset+1, ... (one counter = 4 normals).
    With `tile_ids` (the tiles' numbers within the step) tile k draws from counter offset + tile_ids[k]*numel on,
    whatever batch / rank it is processed in (one launch per tile; numel counters are reserved per tile), so no two
    tiles of a step share noise and the result does not depend on the batching."""
    _dev(tiles, "renoise_mix")
    _dev(mask_tiles, "renoise_mix(mask)")
    if noise is None and tile_ids is not None:
        numel = tiles[0].numel()
        for k, j in enumerate(tile_ids):
            renoise_mix_(tiles[k:k + 1], mask_tiles[k:k + 1], pano_shape, c, s, mix_ratio, None, mask_frame0, seed,
                         offset + j * numel)
        return tiles
    lib = _lib.load()
    n = tiles.shape[0]
    g = _geom(pano_shape, tuple(tiles.shape[2:]), tiles.dtype)
    if noise is not None:
        _dev(noise, "renoise_mix(noise)")
        assert noise.dtype == tiles.dtype and noise.shape == tiles.shape
    ratio = float(mix_ratio)
    st = _stream()
    _timed("renoise_mix", tiles.numel() * tiles.element_size() * (2 + (noise is not None)) + mask_tiles.numel(),
           lambda: check(lib.ds_renoise_mix(tiles.data_ptr(), mask_tiles.data_ptr(), _ptr(noise), float(c), float(s), ratio,
                                            float(1 - mix_ratio), int(bool(mask_frame0)), int(seed), int(offset), C.byref(g), n, st),
                         "ds_renoise_mix"), (n,))
    return tiles


def cfg_ddim(x, eps_c, eps_u, pano_shape, guidance, coef, noise=None):
    """Returns (x_prev, x0) tiles. coef: dict from the scheduler (sqrt_one_minus_at, sqrt_at, sqrt_a_prev, dir_coef, sigma)."""
    _dev(x, "cfg_ddim")
    _dev(eps_c, "cfg_ddim(eps_c)")
    lib = _lib.load()
    n = x.shape[0]
    g = _geom(pano_shape, tuple(x.shape[2:]), x.dtype)
    assert eps_c.shape == x.shape
    if eps_u is not None:
        _dev(eps_u, "cfg_ddim(eps_u)")
        assert eps_u.shape == x.shape and eps_u.dtype == eps_c.dtype
    x_prev = torch.empty_like(x)
    x0 = torch.empty_like(x)
    st = _stream()
    nb = x.numel() * (3 * x.element_size() + eps_c.element_size() * (1 + (eps_u is not None)) + (x.element_size() if noise is not None else 0))
    _timed("cfg_ddim", nb,
           lambda: check(lib.ds_cfg_ddim(x.data_ptr(), eps_c.data_ptr(), _ptr(eps_u), _DT[eps_c.dtype], float(guidance),
                                         float(coef["sqrt_one_minus_at"]), float(coef["sqrt_at"]), float(coef["sqrt_a_prev"]),
                                         float(coef["dir_coef"]), float(coef["sigma"]), _ptr(noise), x_prev.data_ptr(), x0.data_ptr(),
                                         C.byref(g), n, st), "ds_cfg_ddim"), (n,))
    return x_prev, x0


def map_gather(pano, idx):
    """pano [1,C,F,H,W] (fp16/fp32) or uint8 [H,W]; idx int32 device [n,P] -> tiles [n,C,F,P] (or [n,P] for uint8)."""
    _dev(pano, "map_gather")
    lib = _lib.load()
    n, P = idx.shape
    assert idx.dtype == torch.int32 and idx.is_cuda and idx.is_contiguous()
    if pano.dtype == torch.uint8:
        CF, HW, dt = 1, pano.numel(), 2
        out = torch.empty((n, P), dtype=torch.uint8, device=pano.device)
    else:
        _, Cc, F, H, W = pano.shape
        CF, HW, dt = Cc * F, H * W, _DT[pano.dtype]
        out = torch.empty((n, Cc, F, P), dtype=pano.dtype, device=pano.device)
    check(lib.ds_map_gather(pano.data_ptr(), out.data_ptr(), idx.data_ptr(), CF, HW, P, n, dt, _stream()), "ds_map_gather")
    return out


def map_scatter3(pano_latent, pano_x0, mask_pano, x_prev_tiles, x0_tiles, idx):
    ref = pano_latent if pano_latent is not None else pano_x0
    _dev(ref, "map_scatter3")
    lib = _lib.load()
    n, P = idx.shape
    _, Cc, F, H, W = ref.shape
    for t in (x_prev_tiles, x0_tiles):
        if t is not None:
            _dev(t, "map_scatter3(tile)")
            assert t.dtype == ref.dtype and t.numel() == n * Cc * F * P
    check(lib.ds_map_scatter3(_ptr(pano_latent), _ptr(pano_x0), _ptr(mask_pano), _ptr(x_prev_tiles), _ptr(x0_tiles),
                              idx.data_ptr(), Cc * F, H * W, P, n, _DT[ref.dtype], _stream()), "ds_map_scatter3")


def map_gather_frames(pano, idx, f0, tf):
    """pano [1,C,F,H,W] (fp16/fp32) or uint8 mask [F,H,W]; idx int32 device [n,P]; f0 int32 device [n] (first panorama
    frame of each tile, wraps modulo F) -> tiles [n,C,tf,P] (mask: [n,tf,P])."""
    _dev(pano, "map_gather_frames")
    lib = _lib.load()
    n, P = idx.shape
    assert idx.dtype == torch.int32 and idx.is_cuda and idx.is_contiguous() and f0.dtype == torch.int32 and f0.numel() == n
    if pano.dtype == torch.uint8:
        F, H, W = pano.shape
        Cc, dt = 1, 2
        out = torch.empty((n, tf, P), dtype=torch.uint8, device=pano.device)
    else:
        _, Cc, F, H, W = pano.shape
        dt = _DT[pano.dtype]
        out = torch.empty((n, Cc, tf, P), dtype=pano.dtype, device=pano.device)
    check(lib.ds_map_gather_frames(pano.data_ptr(), out.data_ptr(), idx.data_ptr(), f0.data_ptr(), Cc, F, tf, H * W, P, n, dt,
                                   _stream()), "ds_map_gather_frames")
    return out


def map_scatter3_frames(pano_latent, pano_x0, mask_pano, x_prev_tiles, x0_tiles, idx, f0, tf):
    """Frame-window scatter; mask_pano uint8 [F,H,W] (set per written frame)."""
    ref = pano_latent if pano_latent is not None else pano_x0
    _dev(ref, "map_scatter3_frames")
    lib = _lib.load()
    n, P = idx.shape
    _, Cc, F, H, W = ref.shape
    for t in (x_prev_tiles, x0_tiles):
        if t is not None:
            _dev(t, "map_scatter3_frames(tile)")
            assert t.dtype == ref.dtype and t.numel() == n * Cc * tf * P and t.is_contiguous()
    assert mask_pano is None or (mask_pano.dtype == torch.uint8 and mask_pano.numel() == F * H * W)
    check(lib.ds_map_scatter3_frames(_ptr(pano_latent), _ptr(pano_x0), _ptr(mask_pano), _ptr(x_prev_tiles), _ptr(x0_tiles),
                                     idx.data_ptr(), f0.data_ptr(), Cc, F, tf, H * W, P, n, _DT[ref.dtype], _stream()),
          "ds_map_scatter3_frames")


def map_splat_(pano, view, tgt, row_ptr, src, wgt):
    """In place bilinear splat of one view [1,C,F,h,w] into pano [1,C,F,H,W] (CSR per target on the device)."""
    _dev(pano, "map_splat")
    _dev(view, "map_splat(view)")
    lib = _lib.load()
    _, Cc, F, H, W = pano.shape
    P = view.shape[-2] * view.shape[-1]
    assert view.dtype == pano.dtype
    check(lib.ds_map_splat(pano.data_ptr(), view.data_ptr(), tgt.data_ptr(), row_ptr.data_ptr(), src.data_ptr(),
                           wgt.data_ptr(), Cc * F, H * W, P, tgt.numel(), _DT[pano.dtype], _stream()), "ds_map_splat")
    return pano


def map_gather_taps(pano, idx, wgt, f0=0, tf=None):
    """pano [1,C,F,H,W]; idx int32 / wgt fp32 device [ntaps,P] -> [1,C,tf,P]: sum_k wgt[k,p] * pano[c,(f0+t)%F,idx[k,p]]."""
    _dev(pano, "map_gather_taps")
    lib = _lib.load()
    _, Cc, F, H, W = pano.shape
    tf = F if tf is None else tf
    ntaps, P = idx.shape
    assert idx.dtype == torch.int32 and wgt.dtype == torch.float32 and idx.is_cuda and wgt.is_cuda and tuple(wgt.shape) == (ntaps, P)
    assert idx.is_contiguous() and wgt.is_contiguous() and pano.is_contiguous()
    out = torch.empty((1, Cc, tf, P), dtype=pano.dtype, device=pano.device)
    check(lib.ds_map_gather_taps(pano.data_ptr(), out.data_ptr(), idx.data_ptr(), wgt.data_ptr(), ntaps, Cc, F, int(f0), tf, H * W, P,
                                 _DT[pano.dtype], _stream()), "ds_map_gather_taps")
    return out


def resize_latent(x, target_height, target_width, mode="nearest"):
    """[B,C,F,H,W] -> [B,C,F,target_height,target_width] per frame (resize_video_latent)."""
    _dev(x, "resize_latent")
    lib = _lib.load()
    B, Cc, F, H, W = x.shape
    out = torch.empty((B, Cc, F, target_height, target_width), dtype=x.dtype, device=x.device)
    m = {"nearest": 0, "bicubic": 1}.get(mode)
    if m is None:
        raise NotImplementedError(f"resize mode {mode!r}: only 'nearest' and 'bicubic' are used by gen_pano_360.py")
    check(lib.ds_resize_latent(x.data_ptr(), out.data_ptr(), _DT[x.dtype], B * Cc * F, H, W, target_height, target_width,
                               m, _stream()), "ds_resize_latent")
    return out


def residual_merge(curr, noised, ratio, step, sparse=True):
    """t2v_normal_pipeline.py:445-468 on [B,C,F,H,W] panoramas; returns the merged panorama (new tensor)."""
    _dev(curr, "residual_merge")
    _dev(noised, "residual_merge(noised)")
    assert curr.shape == noised.shape and curr.dtype == noised.dtype
    out = torch.empty_like(curr)
    H, W = curr.shape[-2:]
    check(_lib.load().ds_residual_merge(curr.data_ptr(), noised.data_ptr(), out.data_ptr(), _DT[curr.dtype],
                                        curr.numel() // (H * W), H, W, float(ratio), float(1.0 - ratio), int(step) % 2,
                                        int(bool(sparse)), _stream()), "ds_residual_merge")
    return out


# ------------------------------------------------------------------------------------------------ UNet ops
def set_launch_share(n):
    """ds_set_launch_share: n similar launch sequences run concurrently on n streams (scheduling hint of the persistent GEMM tiles)."""
    check(_lib.load().ds_set_launch_share(int(n)), "ds_set_launch_share")


def colstats_table(rows, cols, device):
    """Table for ds_gemm_f16_stats: [ceil(rows / 32), cols, 2] fp32 -- (sum, sumsq) per 32-row block and column."""
    return torch.empty(((rows + 31) // 32, cols, 2), dtype=torch.float32, device=device)


def gemm(A, W, bias=None, residual=None, *, M, N, K, out=None, a_mode=DS_A_DENSE, lda=None, cin=None,
         conv=None, tconv=None, bias_rows=None, ldbias=None, epilogue=0, stream=None, colstats=None):
    """out[M, N'] = gatherA[M,K] @ W[N,K]^T with fused epilogue.  conv=(nimg,hin,win,hout,wout,stride,upsample),
    tconv=(t_len,hw).  N' = N/2 for GEGLU.  An fp32 `residual` (strict-precision residual stream) sets DS_EPI_RES_F32, an fp32
    `out` tensor DS_EPI_OUT_F32.  colstats: a [row blocks, >= N, 2] fp32 view (column slice of a colstats_table) that receives the
    per-column partial statistics of the stored tile (ds_gemm_f16_stats)."""
    lib = _lib.load()
    n_out = N // 2 if (epilogue & DS_EPI_GEGLU) else N
    if residual is not None and residual.dtype == torch.float32:
        epilogue |= DS_EPI_RES_F32
    if out is not None and out.dtype == torch.float32:
        epilogue |= DS_EPI_OUT_F32
    if out is None:
        out = torch.empty((M, n_out), dtype=torch.float32 if (epilogue & DS_EPI_OUT_F32) else torch.float16,
                          device=A.device)
    d = GemmDesc()
    d.M, d.N, d.K, d.a_mode = M, N, K, a_mode
    d.cin = K if cin is None else cin
    d.lda = d.cin if lda is None else lda
    if conv is not None:
        d.nimg, d.hin, d.win, d.hout, d.wout, d.stride, d.upsample = conv[:7]
        d.asym_pad = conv[7] if len(conv) > 7 else 0
    if tconv is not None:
        d.t_len, d.hw = tconv
    d.ldc = out.stride(0)
    d.ldr = residual.stride(0) if residual is not None else 0
    d.bias_rows = 0x7FFFFFFF if bias_rows is None else bias_rows      # > M: one shared bias vector (summed first)
    d.ldbias = N if ldbias is None else ldbias
    d.epilogue = epilogue
    st = _stream() if stream is None else stream

    if colstats is not None:
        assert colstats.dtype == torch.float32 and colstats.dim() == 3 and colstats.shape[2] == 2 and colstats.stride(2) == 1 and \
            colstats.stride(1) == 2 and colstats.shape[0] >= (M + 31) // 32 and colstats.shape[1] >= N and colstats.stride(0) % 2 == 0

    def launch():
        if colstats is not None:
            check(lib.ds_gemm_f16_stats(A.data_ptr(), W.data_ptr(), _ptr(bias), _ptr(residual), out.data_ptr(), colstats.data_ptr(),
                                        colstats.stride(0) // 2, C.byref(d), st), "ds_gemm_f16_stats")
            return
        check(lib.ds_gemm_f16(A.data_ptr(), W.data_ptr(), _ptr(bias), _ptr(residual), out.data_ptr(), C.byref(d), st),
              "ds_gemm_f16")

    if _timing_hook is not None:
        _timing_hook("gemm", 2.0 * M * N * K, launch, (a_mode, M, N, K, epilogue, int(residual is not None)))
    else:
        launch()
    return out


def groupnorm_onepass_applies(rows_per_inst, Cch, dtype, groups=32):
    return bool(_lib.load().ds_groupnorm_onepass_applies(rows_per_inst, Cch, groups, _DT[dtype]))


def groupnorm(x, gamma, beta, ninst, rows_per_inst, Cch, eps, silu, groups=32, stream=None, raw_f16=False, colstats=None, onepass=False):
    """x [ninst*rows_per_inst, Cch] fp16 or fp32, rows contiguous or a column slice of a wider row-major buffer (row stride
    x.stride(0)); returns a dense fp16 [rows, Cch] tensor.  raw_f16 (fp32 x only): also returns fp16(x) as a dense tensor,
    written in the same pass -> (y, x16).  colstats: the producer's partial-statistics view for x's columns (gemm(..., colstats=)):
    no statistics pass over x (ds_groupnorm_rows_colstats; rows_per_inst % 32 == 0)."""
    lib = _lib.load()
    st = _stream() if stream is None else stream
    assert x.dim() == 2 and x.shape[1] == Cch and x.stride(1) == 1, "groupnorm: x must be [rows, C] with unit column stride"
    ws = torch.empty((lib.ds_groupnorm_stats_workspace_floats(ninst, rows_per_inst, groups),), dtype=torch.float32,
                     device=x.device)
    y = torch.empty((x.shape[0], Cch), dtype=torch.float16, device=x.device)
    x16 = torch.empty((x.shape[0], Cch), dtype=torch.float16, device=x.device) if raw_f16 else None
    if onepass:       # the one-launch read-once form (ds_groupnorm_rows_onepass; opt-in, see include/dynscaler_hip.h)
        check(lib.ds_groupnorm_rows_onepass(x.data_ptr(), _DT[x.dtype], x.stride(0), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), _ptr(x16),
                                            ninst, rows_per_inst, Cch, groups, float(eps), int(bool(silu)), st), "ds_groupnorm_rows_onepass")
        return (y, x16) if raw_f16 else y
    if colstats is not None:
        assert colstats.dtype == torch.float32 and colstats.shape[1] >= Cch and colstats.stride(1) == 2 and colstats.stride(2) == 1
        check(lib.ds_groupnorm_rows_colstats(x.data_ptr(), _DT[x.dtype], x.stride(0), colstats.data_ptr(), colstats.stride(0) // 2,
                                             gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), _ptr(x16), ws.data_ptr(), ninst, rows_per_inst,
                                             Cch, groups, float(eps), int(bool(silu)), st), "ds_groupnorm_rows_colstats")
        return (y, x16) if raw_f16 else y
    check(lib.ds_groupnorm_rows(x.data_ptr(), _DT[x.dtype], x.stride(0), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), _ptr(x16),
                                ws.data_ptr(), ninst, rows_per_inst, Cch, groups, float(eps), int(bool(silu)), st), "ds_groupnorm_rows")
    return (y, x16) if raw_f16 else y


def groupnorm_stats(x, ninst, rows_per_inst, Cch, eps, groups=32, stream=None):
    """mean, rstd fp32 [ninst*groups] (the two-step C-ABI entry points; `groupnorm` uses the fused ds_groupnorm_f16)."""
    lib = _lib.load()
    st = _stream() if stream is None else stream
    stats = torch.empty((2, ninst * groups), dtype=torch.float32, device=x.device)
    ws = torch.empty((lib.ds_groupnorm_stats_workspace_floats(ninst, rows_per_inst, groups),), dtype=torch.float32,
                     device=x.device)
    check(lib.ds_groupnorm_stats(x.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), ws.data_ptr(), ninst,
                                 rows_per_inst, Cch, groups, float(eps), st), "ds_groupnorm_stats")
    return stats[0], stats[1]


def groupnorm_apply(x, mean, rstd, gamma, beta, ninst, rows_per_inst, Cch, silu, groups=32, stream=None):
    lib = _lib.load()
    st = _stream() if stream is None else stream
    y = torch.empty_like(x)
    check(lib.ds_groupnorm_apply(x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                 y.data_ptr(), ninst, rows_per_inst, Cch, groups, int(bool(silu)), st), "ds_groupnorm_apply")
    return y


def layernorm(x, gamma, beta, eps=1e-5, stream=None, out=None):
    lib = _lib.load()
    st = _stream() if stream is None else stream
    rows, Cch = x.shape
    y = torch.empty((rows, Cch), dtype=torch.float16, device=x.device) if out is None else out
    check(lib.ds_layernorm_rows(x.data_ptr(), _DT[x.dtype], gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), rows, Cch, float(eps),
                                st), "ds_layernorm_rows")
    return y


def cast_rows_f16(x, stream=None):
    """fp32 rows [M, C] (unit column stride, any row stride) -> dense fp16 [M, C]."""
    lib = _lib.load()
    st = _stream() if stream is None else stream
    assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
    y = torch.empty(x.shape, dtype=torch.float16, device=x.device)
    check(lib.ds_cast_rows_f32_f16(x.data_ptr(), x.stride(0), y.data_ptr(), y.stride(0), x.shape[0], x.shape[1], st),
          "ds_cast_rows_f32_f16")
    return y


def layernorm_stats(x, eps=1e-5, stream=None):
    """(mean, rstd) per row as fp32 [rows, 2] (the statistics half of nn.LayerNorm; see gemm_ln)."""
    lib = _lib.load()
    st = _stream() if stream is None else stream
    rows, Cch = x.shape
    stats = torch.empty((rows, 2), dtype=torch.float32, device=x.device)
    check(lib.ds_layernorm_stats(x.data_ptr(), stats.data_ptr(), rows, Cch, float(eps), st), "ds_layernorm_stats")
    return stats


def gemm_ln(x, Wg, stats, colsum, colbias=None, *, M, N, K, out=None, epilogue=0, stream=None, eps=1e-5):
    """out = LayerNorm(x) @ W^T (+ b) with the LayerNorm folded into the GEMM: x raw fp16 [M,K], Wg = fp16(gamma*W) [N,K],
    colsum / colbias fp32 [N].  stats from layernorm_stats (ds_gemm_f16_ln), or None: the kernel takes the rows' statistics
    from its own operand fragments (ds_gemm_f16_lnk, `eps`) and no statistics launch exists.  N' = N/2 for GEGLU."""
    lib = _lib.load()
    n_out = N // 2 if (epilogue & DS_EPI_GEGLU) else N
    if out is None:
        out = torch.empty((M, n_out), dtype=torch.float16, device=x.device)
    d = GemmDesc()
    d.M, d.N, d.K, d.a_mode = M, N, K, DS_A_DENSE
    d.cin, d.lda = K, x.stride(0)
    d.ldc, d.ldr = out.stride(0), 0
    d.bias_rows, d.ldbias = 0x7FFFFFFF, N
    d.epilogue = epilogue
    st = _stream() if stream is None else stream

    def launch():
        if stats is None:
            check(lib.ds_gemm_f16_lnk(x.data_ptr(), Wg.data_ptr(), float(eps), colsum.data_ptr(), _ptr(colbias),
                                      out.data_ptr(), C.byref(d), st), "ds_gemm_f16_lnk")
        else:
            check(lib.ds_gemm_f16_ln(x.data_ptr(), Wg.data_ptr(), stats.data_ptr(), colsum.data_ptr(), _ptr(colbias),
                                     out.data_ptr(), C.byref(d), st), "ds_gemm_f16_ln")

    if _timing_hook is not None:
        _timing_hook("gemm", 2.0 * M * N * K, launch, (DS_A_DENSE, M, N, K, epilogue, 0))
    else:
        launch()
    return out


def attention(q, k, v, out, *, batch, heads, nq, nk, ldq, ldk, ldv, ldo, kv_batch_div=1, scale, accumulate=False,
              stream=None):
    lib = _lib.load()
    st = _stream() if stream is None else stream

    def launch():
        check(lib.ds_attention_f16(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), batch, heads, nq, nk, ldq,
                                   ldk, ldv, ldo, kv_batch_div, float(scale), int(bool(accumulate)), st),
              "ds_attention_f16")

    if _timing_hook is not None:
        _timing_hook("attention", 4.0 * batch * heads * nq * nk * 64, launch, (batch, heads, nq, nk))
    else:
        launch()
    return out


def temporal_attention(q, k, v, out, *, nseq_batches, T, hw, heads, ldq, ldk, ldv, ldo, scale, stream=None):
    lib = _lib.load()
    st = _stream() if stream is None else stream
    check(lib.ds_temporal_attention_f16(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), nseq_batches, T, hw,
                                        heads, ldq, ldk, ldv, ldo, float(scale), st), "ds_temporal_attention_f16")
    return out


def concat_channels(a, b, stream=None):
    lib = _lib.load()
    st = _stream() if stream is None else stream
    rows, c1 = a.shape
    c2 = b.shape[1]
    dst = torch.empty((rows, c1 + c2), dtype=a.dtype, device=a.device)
    check(lib.ds_concat_channels(a.data_ptr(), b.data_ptr(), dst.data_ptr(), rows, c1, c2, st), "ds_concat_channels")
    return dst


def im2col_in(x, kpad, stream=None, out_dtype=torch.float16):
    """3x3 patches [B*T*H*W, kpad] of a [B, C, T, H, W] input (conv_in as a dense GEMM); fp32 patches: ds_im2col_in_f32."""
    lib = _lib.load()
    st = _stream() if stream is None else stream
    B, Cc, T, H, W = x.shape
    patches = torch.empty((B * T * H * W, kpad), dtype=out_dtype, device=x.device)
    if out_dtype == torch.float32:
        check(lib.ds_im2col_in_f32(x.data_ptr(), _DT[x.dtype], patches.data_ptr(), B, Cc, T, H, W, kpad, st), "ds_im2col_in_f32")
        return patches
    check(lib.ds_im2col_in(x.data_ptr(), _DT[x.dtype], patches.data_ptr(), B, Cc, T, H, W, kpad, st), "ds_im2col_in")
    return patches


def im2col_in_affine(x, kpad, wmat, bvec, in_scale, stream=None, out_dtype=torch.float16):
    """conv_in patches of post_quant_conv(x * in_scale) (first-stage decoder); wmat fp32 [C,C], bvec fp32 [C] on the device.
    out_dtype fp32: the patches of the wide operand mode (ds_im2col_in_affine_f32)."""
    lib = _lib.load()
    st = _stream() if stream is None else stream
    B, Cc, T, H, W = x.shape
    patches = torch.empty((B * T * H * W, kpad), dtype=out_dtype, device=x.device)
    fn, name = (lib.ds_im2col_in_affine, "ds_im2col_in_affine") if out_dtype == torch.float16 else \
        (lib.ds_im2col_in_affine_f32, "ds_im2col_in_affine_f32")
    check(fn(x.data_ptr(), _DT[x.dtype], patches.data_ptr(), B, Cc, T, H, W, kpad, wmat.data_ptr(), bvec.data_ptr(), float(in_scale), st), name)
    return patches


def softmax_rows(s, scale, out=None, stream=None):
    """fp32 scores [rows, cols] -> softmax(s * scale) over the columns: fp16, or fp32 into an fp32 `out` (ds_softmax_rows_f32)."""
    lib = _lib.load()
    st = _stream() if stream is None else stream
    rows, cols = s.shape
    if out is None:
        out = torch.empty((rows, cols), dtype=torch.float16, device=s.device)
    assert out.shape[0] == rows and out.shape[1] >= cols
    if out.dtype == torch.float32:
        check(lib.ds_softmax_rows_f32(s.data_ptr(), out.data_ptr(), rows, cols, s.stride(0), out.stride(0), float(scale), st),
              "ds_softmax_rows_f32")
        return out
    check(lib.ds_softmax_rows(s.data_ptr(), out.data_ptr(), rows, cols, s.stride(0), out.stride(0), float(scale), st),
          "ds_softmax_rows")
    return out


def posterior_sample(moments, shape, noise, scale, stream=None):
    """moments fp32 rows [M, >=2C] -> scale * (mean + std * noise) as [B,C,T,H,W] fp32 (noise None = the mode)."""
    lib = _lib.load()
    st = _stream() if stream is None else stream
    B, Cc, T, H, W = shape
    out = torch.empty(shape, dtype=torch.float32, device=moments.device)
    check(lib.ds_posterior_sample(moments.data_ptr(), moments.stride(0), _ptr(noise), out.data_ptr(), B, Cc, T, H, W,
                                  float(scale), st), "ds_posterior_sample")
    return out


def rows_to_ncthw(y, shape, out_dtype, stream=None):
    lib = _lib.load()
    st = _stream() if stream is None else stream
    B, Cc, T, H, W = shape
    out = torch.empty(shape, dtype=out_dtype, device=y.device)
    check(lib.ds_rows_to_ncthw(y.data_ptr(), _DT[y.dtype], y.stride(0), out.data_ptr(), _DT[out_dtype], B, Cc, T, H, W,
                               st), "ds_rows_to_ncthw")
    return out


def timestep_embedding(t, dim, stream=None):
    lib = _lib.load()
    st = _stream() if stream is None else stream
    assert t.dtype == torch.int64
    out = torch.empty((t.shape[0], dim), dtype=torch.float16, device=t.device)
    check(lib.ds_timestep_embedding(t.data_ptr(), out.data_ptr(), t.shape[0], dim, st), "ds_timestep_embedding")
    return out


def silu(x, stream=None):
    lib = _lib.load()
    st = _stream() if stream is None else stream
    y = torch.empty_like(x)
    check(lib.ds_silu_f16(x.data_ptr(), y.data_ptr(), x.numel(), st), "ds_silu_f16")
    return y


# ------------------------------------------------------------------------------------------------ encoder ops (N3)
def attention_enc(q, k, v, out, *, batch, heads, nq, nk, ldq, ldk, ldv, ldo, head_dim, scale, causal=False, stream=None):
    lib = _lib.load()
    st = _stream() if stream is None else stream
    check(lib.ds_attention_enc_f16(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), batch, heads, nq, nk, ldq,
                                   ldk, ldv, ldo, head_dim, float(scale), int(bool(causal)), st), "ds_attention_enc_f16")
    return out


def gelu_(x, stream=None):
    lib = _lib.load()
    st = _stream() if stream is None else stream
    check(lib.ds_gelu_f16(x.data_ptr(), x.data_ptr(), x.numel(), st), "ds_gelu_f16")
    return x


def embed_tokens(tokens, table, pos, stream=None):
    """tokens int32 [b, ctx] -> fp16 [b*ctx, width] = table[tokens] + pos."""
    lib = _lib.load()
    st = _stream() if stream is None else stream
    _dev(tokens, "tokens")
    assert tokens.dtype == torch.int32 and tokens.is_contiguous()
    b, ctx = tokens.shape
    vocab, width = table.shape
    out = torch.empty((b * ctx, width), dtype=torch.float16, device=tokens.device)
    check(lib.ds_embed_tokens(tokens.data_ptr(), table.data_ptr(), pos.data_ptr(), out.data_ptr(), b * ctx, ctx, width,
                              vocab, st), "ds_embed_tokens")
    return out


def vit_assemble(patches, cls, pos, nimg, stream=None):
    lib = _lib.load()
    st = _stream() if stream is None else stream
    width = patches.shape[1]
    grid2 = patches.shape[0] // nimg
    out = torch.empty((nimg * (grid2 + 1), width), dtype=torch.float16, device=patches.device)
    check(lib.ds_vit_assemble(patches.data_ptr(), cls.data_ptr(), pos.data_ptr(), out.data_ptr(), nimg, grid2, width, st),
          "ds_vit_assemble")
    return out


def clip_preprocess(img, size, mean, std, antialias=True, stream=None):
    """img [n,3,H,W] fp32/fp16 in [-1,1] on the device -> fp32 [n,3,size,size]."""
    lib = _lib.load()
    st = _stream() if stream is None else stream
    img = _dev(img.contiguous(), "image")
    n, c, H, W = img.shape
    out = torch.empty((n, c, size, size), dtype=torch.float32, device=img.device)
    m3 = (C.c_float * 3)(*[float(x) for x in mean])
    s3 = (C.c_float * 3)(*[float(x) for x in std])
    check(lib.ds_clip_preprocess(img.data_ptr(), _DT[img.dtype], out.data_ptr(), n, c, H, W, size, int(bool(antialias)),
                                 C.cast(m3, C.c_void_p), C.cast(s3, C.c_void_p), st), "ds_clip_preprocess")
    return out


def patchify(img, patch, kpad, stream=None):
    lib = _lib.load()
    st = _stream() if stream is None else stream
    n, c, S, _ = img.shape
    assert img.dtype == torch.float32 and img.is_contiguous()
    g = S // patch
    rows = torch.empty((n * g * g, kpad), dtype=torch.float16, device=img.device)
    check(lib.ds_patchify(img.data_ptr(), rows.data_ptr(), n, c, S, patch, kpad, st), "ds_patchify")
    return rows


# ------------------------------------------------------------------------------------------------ the wide operand mode (csrc/wide.hip)
def split_f16(x, stream=None):
    """(hi, lo) fp16 planes of x (fp32 or fp16): hi = fp16(x), lo = fp16((x - hi) * 2^11) -- the weight operand of gemm_wide."""
    lib = _lib.load()
    _dev(x, "split_f16")
    hi = torch.empty(x.shape, dtype=torch.float16, device=x.device)
    lo = torch.empty(x.shape, dtype=torch.float16, device=x.device)
    check(lib.ds_split_f16(x.data_ptr(), _DT[x.dtype], hi.data_ptr(), lo.data_ptr(), x.numel(), _stream() if stream is None else stream),
          "ds_split_f16")
    return hi, lo


def gemm_wide(A, W_hi, W_lo, bias=None, residual=None, *, M, N, K, out=None, a_mode=DS_A_DENSE, lda=None, cin=None, conv=None,
              tconv=None, bias_rows=None, ldbias=None, epilogue=0, stream=None):
    """gemm() with fp32 A / residual / out and split-fp16 products (ds_gemm_wide); same geometry arguments."""
    lib = _lib.load()
    n_out = N // 2 if (epilogue & DS_EPI_GEGLU) else N
    assert A.dtype == torch.float32 and W_hi.dtype == torch.float16 and W_lo.dtype == torch.float16
    assert residual is None or residual.dtype == torch.float32
    if out is None:
        out = torch.empty((M, n_out), dtype=torch.float32, device=A.device)
    assert out.dtype == torch.float32
    d = GemmDesc()
    d.M, d.N, d.K, d.a_mode = M, N, K, a_mode
    d.cin = K if cin is None else cin
    d.lda = d.cin if lda is None else lda
    if conv is not None:
        d.nimg, d.hin, d.win, d.hout, d.wout, d.stride, d.upsample = conv[:7]
        d.asym_pad = conv[7] if len(conv) > 7 else 0
    if tconv is not None:
        d.t_len, d.hw = tconv
    d.ldc = out.stride(0)
    d.ldr = residual.stride(0) if residual is not None else 0
    d.bias_rows = 0x7FFFFFFF if bias_rows is None else bias_rows
    d.ldbias = N if ldbias is None else ldbias
    d.epilogue = epilogue | DS_EPI_OUT_F32 | (DS_EPI_RES_F32 if residual is not None else 0)
    st = _stream() if stream is None else stream
    check(lib.ds_gemm_wide(A.data_ptr(), W_hi.data_ptr(), W_lo.data_ptr(), _ptr(bias), _ptr(residual), out.data_ptr(), C.byref(d), st),
          "ds_gemm_wide")
    return out


def groupnorm_wide(x, gamma, beta, ninst, rows_per_inst, Cch, eps, silu, groups=32, stream=None):
    """GroupNorm (+ SiLU) of fp32 rows [ninst*rows_per_inst, Cch] (row stride x.stride(0)) -> dense fp32."""
    lib = _lib.load()
    assert x.dtype == torch.float32 and x.dim() == 2 and x.shape[1] == Cch and x.stride(1) == 1
    y = torch.empty((x.shape[0], Cch), dtype=torch.float32, device=x.device)
    stats = torch.empty((lib.ds_groupnorm_wide_scratch_floats(ninst, rows_per_inst, groups),), dtype=torch.float32, device=x.device)
    check(lib.ds_groupnorm_wide(x.data_ptr(), x.stride(0), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), stats.data_ptr(), ninst,
                                rows_per_inst, Cch, groups, float(eps), int(bool(silu)), _stream() if stream is None else stream),
          "ds_groupnorm_wide")
    return y


def layernorm_wide(x, gamma, beta, eps=1e-5, stream=None):
    lib = _lib.load()
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 2
    y = torch.empty_like(x)
    check(lib.ds_layernorm_wide(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), x.shape[0], x.shape[1], float(eps),
                                _stream() if stream is None else stream), "ds_layernorm_wide")
    return y


def attention_wide(q, k, v, out, *, batch, heads, nq, nk, ldq, ldk, ldv, ldo, kv_batch_div=1, scale, accumulate=False, stream=None):
    lib = _lib.load()
    check(lib.ds_attention_wide(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), batch, heads, nq, nk, ldq, ldk, ldv, ldo,
                                kv_batch_div, float(scale), int(bool(accumulate)), _stream() if stream is None else stream),
          "ds_attention_wide")
    return out


def temporal_attention_wide(q, k, v, out, *, nseq_batches, T, hw, heads, ldq, ldk, ldv, ldo, scale, stream=None):
    lib = _lib.load()
    check(lib.ds_temporal_attention_wide(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), nseq_batches, T, hw, heads, ldq, ldk,
                                         ldv, ldo, float(scale), _stream() if stream is None else stream), "ds_temporal_attention_wide")
    return out
