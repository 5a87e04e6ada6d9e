/*
 * dynscaler_hip.h -- C ABI of libdynscaler_hip.so: the MI355X (gfx950) kernels behind the tiled
 * panoramic denoising hot path of DynamicScaler.
 *
 * The reference (pure Python, /root/reference) has no FFI layer; the entry points below are what a
 * binding of its hot-path functions binds to.  Each one cites the reference code it replaces
 * (paths relative to the reference root).  INTEGRATION.md shows the ctypes stub a maintainer adds.
 *
 * Conventions
 *   - return 0 on success, a negative DS_E* code on error; ds_last_error() returns a thread-local message.
 *   - no exceptions, no torch types: plain device pointers + sizes.  The CALLER allocates and owns every
 *     buffer (inputs, outputs, workspaces).  Nothing here allocates, frees or synchronises.
 *   - every call is asynchronous on the `stream` argument (a hipStream_t passed as void*; NULL = default
 *     stream) and is legal inside hipGraph stream capture.
 *   - device = the caller's current HIP device.  Re-entrant across streams.
 *   - activations are "rows x channels", channel-contiguous fp16 ("NTHWC": row = ((b*T + t)*H + y)*W + x).
 *     latents / panoramas keep the reference layout [B=1, C, F, H, W] (fp16 or fp32).
 */
#ifndef DYNSCALER_HIP_H
#define DYNSCALER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DS_OK 0
#define DS_EINVAL (-1)   /* bad argument (message in ds_last_error) */
#define DS_ELAUNCH (-2)  /* HIP launch error */

#define DS_F16 0
#define DS_F32 1

const char* ds_last_error(void);
/* ABI version of this header; bumped on any signature or struct-layout change (round 4: 2 -- ds_unet_config gained
 * temporal_selfatt_only, new entry points; round 5: 3 -- the wide operand mode's entry points, residual_f32 = 3).  Bindings compare
 * it with the version they were written for and refuse a stale library. */
#define DS_ABI_VERSION 3
int ds_abi_version(void);

/* ------------------------------------------------------------------------------------------------
 * Ring (plane) panorama tile ops -- bit-exact restatements of the reference's torch sequences.
 * ---------------------------------------------------------------------------------------------- */

/* Geometry of one batch of equally sized windows on a wrap-around panorama [1,C,F,H,W]. */
typedef struct ds_ring_geom {
    int32_t C, F, H, W;   /* panorama */
    int32_t tf, th, tw;   /* window (tile) size; tf<=F, th<=H, tw<=W for scatter */
    int32_t dtype;        /* DS_F16 / DS_F32: element type of panorama AND tiles */
} ds_ring_geom;

#define DS_MAX_WINDOWS 64

/* RingLatent.get_window_latent (utils/shift_window_utils.py:48-114), n windows in one launch:
 *   tiles[i][c][f][y][x] = pano[c][(f0_i+f)%F][(y0_i+y)%H][(x0_i+x)%W]
 * and, when mask_pano != NULL, mask_tiles[i][f][y][x] = mask_pano[(f0_i+f)%F][...] (uint8, one byte per
 * (f,y,x): the reference's 5-D fp32 mask is only ever written uniformly over C --
 * pipeline/t2v_sphere_panorama_pipeline.py:494,624-632).
 * origins: HOST pointer to n triples (f0,y0,x0); 0 <= origin < 2*size like the reference asserts (:73-75). */
int ds_ring_gather(const void* pano, const uint8_t* mask_pano, void* tiles, uint8_t* mask_tiles,
                   const ds_ring_geom* geom, const int32_t* origins, int n, void* stream);

/* RingLatent.set_window_latent x3 (utils/shift_window_utils.py:116-206; call sites
 * pipeline/t2v_sphere_panorama_pipeline.py:609-632): overwrite the windows of pano_latent with x_prev,
 * of pano_x0 with x0, and set mask_pano to 1.  Any of the three destinations may be NULL.
 * Windows of one launch must be pairwise disjoint (the caller batches only independent columns). */
int ds_ring_scatter3(void* pano_latent, void* pano_x0, uint8_t* mask_pano, const void* x_prev_tiles,
                     const void* x0_tiles, const ds_ring_geom* geom, const int32_t* origins, int n,
                     void* stream);

/* The ring step's tile ops fused around the UNet (round 4): the same fp32 operations in the same order as ds_ring_gather +
 * ds_renoise_mix and ds_cfg_ddim + ds_ring_scatter3, without the tile tensors in between.
 * ds_ring_gather_renoise (utils/shift_window_utils.py:48-114 + pipeline/scheduler.py:98-110 + utils/tensor_utils.py:19-39; call site
 * pipeline/t2v_sphere_panorama_pipeline.py:538-559): tiles[i] = mix(window_i, c*window_i + s*noise, mask window_i, ratio); the mask
 * is read straight from the mask panorama [F][H][W] (frame f0 of the window for every frame when mask_frame0); mask_tiles (may be
 * NULL) receives the window's mask.  noise: [n][C][tf][th][tw] or NULL = in-kernel Philox, tile i drawing from counter
 * tile_offsets[i] (HOST int64 [n]) on -- the stream of one ds_renoise_mix call per tile with that offset.
 * ds_cfg_ddim_scatter (t2v_sphere_panorama_pipeline.py:599-632, pipeline/scheduler.py:60-96, shift_window_utils.py:116-206): the
 * CFG-combined DDIM update of n pairwise-disjoint windows written into pano_latent (x_prev) and pano_x0 (pred_x0), mask_pano set to 1
 * over the windows. */
int ds_ring_gather_renoise(const void* pano, const uint8_t* mask_pano, void* tiles, uint8_t* mask_tiles, const void* noise, float c,
                           float s, float ratio, float one_minus_ratio, int mask_frame0, uint64_t seed, const int64_t* tile_offsets,
                           const ds_ring_geom* geom, const int32_t* origins, int n, void* stream);
int ds_cfg_ddim_scatter(const void* x, const void* eps_c, const void* eps_u, int eps_dtype, float guidance, float sqrt_one_minus_at,
                        float sqrt_at, float sqrt_a_prev, float dir_coef, float sigma, const void* noise, void* pano_latent,
                        void* pano_x0, uint8_t* mask_pano, const ds_ring_geom* geom, const int32_t* origins, int n, void* stream);

/* re_noise + mix_latents_with_mask fused (pipeline/scheduler.py:98-110, utils/tensor_utils.py:19-39;
 * call site pipeline/t2v_sphere_panorama_pipeline.py:550-559):
 *   noised = c*x + s*noise ;  out = x*(1-m) + (x*(1-ratio) + noised*ratio)*m     (fp32, same op order)
 * tiles: [n][C][tf][th][tw] in/out.  mask_tiles uint8 [n][tf][th][tw]; mask_frame0 != 0 uses frame 0 of
 * the window for every frame (the t2v `[0,0,[0]]` slice, :555).  noise: same shape/dtype as tiles, or NULL.
 * `ratio` / `one_minus_ratio` are fp32(mix_ratio) and fp32(1 - mix_ratio) with the subtraction done by the
 * caller in double, exactly as Python evaluates `1 - mix_ratio` before torch narrows it (:30).
 * NULL noise: draw N(0,1) in-kernel from Philox4x32-10 keyed by (seed, element index + offset) -- the
 * in-kernel stream is NOT the torch CPU stream (perf mode, not bit-comparable). */
int ds_renoise_mix(void* tiles, const uint8_t* mask_tiles, const void* noise, float c, float s, float ratio,
                   float one_minus_ratio, int mask_frame0, uint64_t seed, uint64_t offset,
                   const ds_ring_geom* geom, int n, void* stream);

/* CFG combine + DDIM step fused (pipeline/t2v_sphere_panorama_pipeline.py:599, pipeline/scheduler.py:60-96):
 *   e = e_u + g*(e_c - e_u) ; x0 = (x - sqrt_one_minus_at*e)/sqrt_at ; x_prev = sqrt_a_prev*x0 + dir_coef*e + sigma*z
 * All four coefficient scalars are the fp32 values the reference materialises with torch.full (:78-85).
 * eps_u may be NULL (guidance 1.0 path: e = e_c).  eps_dtype: DS_F16/DS_F32 element type of eps_c/eps_u;
 * x / x_prev / x0 use geom->dtype.  noise (sigma*z term) may be NULL when sigma == 0. */
int ds_cfg_ddim(const void* x, const void* eps_c, const void* eps_u, int eps_dtype, float guidance,
                float sqrt_one_minus_at, float sqrt_at, float sqrt_a_prev, float dir_coef, float sigma,
                const void* noise, void* x_prev, void* x0, const ds_ring_geom* geom, int n, void* stream);

/* Sphere path (utils/panorama_tensor_utils.py, utils/ring_panorama_tensor_utils.py): perspective view <-> equirect
 * panorama by a host-computed nearest-neighbour index map (the map comes from the reference's fp32 torch op sequence of
 * _get_uv, :204-245, evaluated on the host so floor() sees identical values; cached per (fov,theta,phi,w,h,W,H)).
 *   ds_map_gather   : get_view_tensor_no_interpolate / _sample_equirect_tensor_nearest (:53-70,:185-202)
 *                     tiles[i][cf][p] = idx[i][p] >= 0 ? pano[cf][idx[i][p]] : 0     (cf = fused channel*frame index)
 *                     dtype DS_F16 / DS_F32 / 2 (uint8, the denoised-mask panorama)
 *   ds_map_scatter3 : set_view_tensor_no_interpolation x3 (:154-183); idx[i][p] < 0 skips the source.  The caller
 *                     resolves duplicate targets on the host (torch's CPU index_put keeps the LAST source in row-major
 *                     view order; losers get -1), so the kernel is race-free; views of one launch must be disjoint.
 * idx is a DEVICE pointer [n][P] (P = view h*w), HW = panorama H*W. */
int ds_map_gather(const void* pano, void* tiles, const int32_t* idx, int CF, int HW, int P, int n, int dtype,
                  void* stream);
int ds_map_scatter3(void* pano_latent, void* pano_x0, uint8_t* mask_pano, const void* x_prev_tiles,
                    const void* x0_tiles, const int32_t* idx, int CF, int HW, int P, int n, int dtype, void* stream);
/* Frame-window forms (RingPanoramaLatentProxy, utils/ring_panorama_tensor_utils.py:262-314, used by the i2v sphere loop
 * pipeline/i2v_sphere_panorama_pipeline.py:330-336,438-471): pano [C][F][HW]; tile i holds tf frames starting at
 * panorama frame f0[i] (DEVICE int32 [n], NULL = 0), wrapping modulo F.  The mask panorama is [F][HW] here (one byte
 * per frame and pixel, gathered with C = 1 / dtype 2, set per written frame by the scatter). */
int ds_map_gather_frames(const void* pano, void* tiles, const int32_t* idx, const int32_t* f0, int C, int F, int tf,
                         int HW, int P, int n, int dtype, void* stream);
int ds_map_scatter3_frames(void* pano_latent, void* pano_x0, uint8_t* mask_pano, const void* x_prev_tiles,
                           const void* x0_tiles, const int32_t* idx, const int32_t* f0, int C, int F, int tf, int HW,
                           int P, int n, int dtype, void* stream);
/* set_view_tensor_bilinear (utils/panorama_tensor_utils.py:98-152): 4-tap splat with weight normaliser,
 * pano[t] = sum_j view[src_j] * w_j / sum_j w_j for every panorama pixel t that receives weight.  The host inverts the
 * view's map into a CSR list per target (tgt[ntgt], row_ptr[ntgt+1], src/wgt[nnz], all DEVICE pointers) whose entry
 * order is the reference's index_add_ order, so the kernel needs no atomics and reproduces the CPU sums bit for bit. */
int ds_map_splat(void* pano, const void* view, const int32_t* tgt, const int32_t* row_ptr, const int32_t* src,
                 const float* wgt, int CF, int HW, int P, int ntgt, int dtype, void* stream);
/* get_view_tensor_interpolate (utils/panorama_tensor_utils.py:28-51, utils/ring_panorama_tensor_utils.py:33-57): the view as a
 * weighted sum of `ntaps` panorama pixels per view pixel -- F.grid_sample's bilinear / border / align_corners taps, resolved on
 * the host into idx[ntaps][P] (DEVICE int32) and wgt[ntaps][P] (DEVICE fp32; a tap outside the panorama has weight 0):
 * out[c][t][p] = sum_k wgt[k][p] * pano[c][(f0 + t) % F][idx[k][p]], summed in tap order in fp32, rounded once.
 * pano [C][F][HW], out [C][tf][P]; F = tf = 1, f0 = 0 for a plain PanoramaTensor with C = all its planes. */
int ds_map_gather_taps(const void* pano, void* out, const int32_t* idx, const float* wgt, int ntaps, int C, int F, int f0, int tf,
                       int HW, int P, int dtype, void* stream);
/* Per-step residual merge of the non-overlapping grid loop (VC2_Pipeline_T2V.basic_sample_shift_multi_windows,
 * pipeline/t2v_normal_pipeline.py:445-468): curr = the panorama latent, noised = the resized pre-denoised latent re-noised
 * to the step's level, both [planes][H][W] (planes = B*C*F).  sparse == 0: out = curr*r + noised*(1-r).  sparse != 0
 * (parity = step & 1, even H and W): out[p::2, ::2] = r*curr[(1-p)::2, ::2] + (1-r)*noised[::2, ::2] and
 * out[(1-p)::2, 1::2] = r*curr[p::2, 1::2] + (1-r)*noised[::2, ::2], the rest copied.  out must not alias the inputs. */
int ds_residual_merge(const void* curr, const void* noised, void* out, int dtype, long planes, int H, int W, float r,
                      float one_minus_r, int parity, int sparse, void* stream);
/* resize_video_latent (utils/diffusion_utils.py:21-33; stage hand-off gen_pano_360.py:287-289,345-347): F.interpolate
 * over H,W of `planes` = B*C*F images.  mode 0 'nearest', mode 1 'bicubic' (align_corners=False, A=-0.75). */
int ds_resize_latent(const void* in, void* out, int dtype, long planes, int hin, int win, int hout, int wout, int mode,
                     void* stream);

/* ------------------------------------------------------------------------------------------------
 * UNet inner blocks (lvdm/modules/networks/openaimodel3d.py, lvdm/modules/attention.py).
 * ---------------------------------------------------------------------------------------------- */

/* A-operand addressing of the implicit GEMM. */
#define DS_A_DENSE 0   /* nn.Linear / 1x1 conv: row m reads A[m*lda + k]                                    */
#define DS_A_CONV3 1   /* Conv2d 3x3 pad 1 (stride 1|2, optional nearest x2 upsample folded in)            */
#define DS_A_TCONV 2   /* Conv3d (3,1,1) pad (1,0,0) over T (TemporalConvBlock, openaimodel3d.py:257-309)   */

/* epilogue flags */
#define DS_EPI_GEGLU 1   /* out[m][j] = (acc[m][x_j]+b) * gelu(acc[m][gate_j]+b)  (attention.py:376-383);
                            weight (and bias) rows must be interleaved in 32-row groups [x0..31 | gate0..31 |
                            x32..63 | gate32..63 | ...] and N counts x+gate columns; out has N/2 columns    */
#define DS_EPI_SILU 2    /* out = silu(acc + bias)                                                          */
#define DS_EPI_OUT_F32 4 /* store fp32 instead of fp16                                                     */
#define DS_EPI_RES_F32 8 /* `residual` holds fp32 rows (ldr in fp32 elements): the strict-precision residual stream
                            (ResBlock `skip + h`, openaimodel3d.py:237-254; transformer `+ x`, attention.py:216-220) added
                            and -- with DS_EPI_OUT_F32 -- stored without an fp16 rounding.  Shared bias only, no GEGLU.  */

typedef struct ds_gemm_desc {
    int32_t M, N, K;        /* C[M,N] = A'[M,K] * W[N,K]^T ; K = taps*Cin for conv modes; K % 64 == 0       */
    int32_t a_mode;         /* DS_A_*                                                                       */
    int32_t lda;            /* A row stride in elements (>= Cin)                                            */
    int32_t cin;            /* channels per tap (DENSE: == K)                                               */
    int32_t nimg, hin, win; /* CONV3: input images (B*T) and PHYSICAL input H,W                             */
    int32_t hout, wout;     /* CONV3: output H,W (M == nimg*hout*wout)                                      */
    int32_t stride;         /* CONV3: 1 or 2                                                                */
    int32_t upsample;       /* CONV3: 1 = input is nearest-x2 upsampled on the fly (F.interpolate + conv,
                               openaimodel3d.py:98-111); logical input = 2*hin x 2*win                     */
    int32_t t_len, hw;      /* TCONV: frames per sequence, rows per frame (M == nseq*t_len*hw)              */
    int32_t ldc;            /* output row stride in elements                                                */
    int32_t ldr;            /* residual row stride (elements)                                               */
    int32_t bias_rows;      /* bias index = (m / bias_rows) * ldbias + n ; bias_rows >= M -> one shared vector.  A vector
                             * DECLARED shared with bias_rows > M (e.g. INT32_MAX) is summed FIRST (the accumulators start
                             * at it); bias_rows <= M is added after the K sum, so a per-item table gives the same bits
                             * whether a launch covers one item or many. */
    int32_t ldbias;         /* row stride of the bias table in floats (>= N)                                */
    int32_t epilogue;       /* DS_EPI_* flags                                                               */
    int32_t asym_pad;       /* CONV3: 0 = zero padding 1 on every side; 1 = no top/left padding, bottom/right only
                               (F.pad(x, (0,1,0,1)) + stride-2 conv of the first-stage encoder's Downsample,
                               ae_modules.py:102-106)                                                          */
} ds_gemm_desc;

/* fp16 x fp16 -> fp32-accumulate MFMA GEMM with fused bias / per-item bias (emb add) / residual / GEGLU /
 * SiLU epilogues.  Replaces nn.Linear, Conv2d 3x3, Conv2d 1x1, Conv1d 1x1 and Conv3d (3,1,1) call sites:
 * attention.py:54-57,63-64,242,258,379,399; openaimodel3d.py:155-159,174-184,186-193,65-72,98-111,275-300.
 * W: [N][K] fp16, K index = tap*Cin + c (conv weights pre-permuted by the host from [Cout,Cin,kh,kw]).
 * bias: fp32 [ceil(M/bias_rows)][ldbias] or NULL.  residual: fp16 (fp32 with DS_EPI_RES_F32) [M][ldr] or NULL.
 * out: fp16/fp32 [M][ldc]. */
int ds_gemm_f16(const void* A, const void* W, const float* bias, const void* residual, void* out,
                const ds_gemm_desc* desc, void* stream);
/* Hint: n similar launch sequences share the device concurrently on n streams (an 8-GPU rank evaluates the cond and the uncond UNet
 * of its one window per level on two streams).  Persistent big-tile launches then plan their rounds on CUs / n and hand the rows of
 * a last, mostly empty round to small tiles, so that the n launches in flight fill the chip with whole rounds.  Scheduling only:
 * every tile variant sums K in the same order, results do not change.  Process-wide; 1 (default) = a launch plans on the whole chip. */
int ds_set_launch_share(int n);
/* ds_gemm_f16 that also writes the per-column partial statistics of what it stores: colstats[(m / 32) * ld_stats + n] = (sum,
 * sum of squares) as float pairs over the valid rows of each 32-row block (after bias / residual / SiLU; fp16 outputs without an
 * epilogue operand: of the rounded values).  One writer per entry, fixed summation order, no atomics.  The GroupNorm that reads the
 * output (basics.py:76-86; openaimodel3d.py:275-292) takes its statistics from there: ds_groupnorm_rows_colstats.  ld_stats >= N may
 * be the width of a wider table (a concat buffer's: the producers of both halves fill one table).  Not with DS_EPI_GEGLU; operands
 * aligned for the vector epilogue (N % 8 == 0, 16-byte aligned rows). */
int ds_gemm_has_stats(void);     /* 1: this build has ds_gemm_f16_stats (the product library is built without it: the statistics cost registers
                                    in every epilogue and measured no gain; `python -m dynamicscaler_amd.build --variant gemmstats`) */
int ds_gemm_f16_stats(const void* A, const void* W, const float* bias, const void* residual, void* out, float* colstats, int ld_stats,
                      const ds_gemm_desc* desc, void* stream);
/* LayerNorm folded into the projection that consumes it (BasicTransformerBlock: norm1 -> to_q/to_k/to_v, norm2 -> to_q,
 * norm3 -> GEGLU proj; attention.py:199-220, 376-403):  out = LayerNorm(x) W^T + b  computed as
 *   rstd[m] * (x[m,:] . Wg[n,:] - mean[m] * colsum[n]) + colbias[n]
 * with Wg = fp16(gamma (.) W) [N][K], colsum[n] = sum_k Wg[n][k] (fp32), colbias[n] = sum_k beta[k] W[n][k] (+ b[n]),
 * stats from ds_layernorm_stats.  x is the RAW activation (dense A operand, K = the LayerNorm width); the normalised
 * activation is never rounded to fp16 nor written to memory.  desc as for ds_gemm_f16 (a_mode DENSE, fp16 output, no
 * residual); DS_EPI_GEGLU / DS_EPI_SILU apply after the fold.  colbias may be NULL. */
int ds_gemm_f16_ln(const void* x, const void* W_gamma, const float* ln_stats, const float* ln_colsum,
                   const float* ln_colbias, void* out, const ds_gemm_desc* desc, void* stream);
/* The same without a statistics launch: every tile walks its rows' whole K = C extent, so the kernel takes each row's sum and
 * sum of squares from the operand fragments it feeds to the matrix cores (one-pass variance in fp32 over the fp16 values)
 * and forms (mean, 1/sqrt(var + ln_eps)) itself.  LayerNorm -> Linear (-> GEGLU) then reads x once and writes only the
 * projection (attention.py:199-220, 376-403). */
int ds_gemm_f16_lnk(const void* x, const void* W_gamma, float ln_eps, const float* ln_colsum, const float* ln_colbias,
                    void* out, const ds_gemm_desc* desc, void* stream);

/* GroupNorm statistics: x fp16 [ninst*rows_per_inst][C]; instance i = rows [i*rows_per_inst, (i+1)*...).
 * Writes mean/rstd fp32 [ninst][groups]; `workspace` = caller scratch of ds_groupnorm_stats_workspace_floats(...)
 * floats (per-chunk partial sums; no global atomics -> reproducible).  rows_per_inst = H*W (per-frame GroupNorm,
 * basics.py:76-86, attention.py:238) or T*H*W (5-D GroupNorm over T jointly, openaimodel3d.py:275-292,
 * attention.py:297). */
size_t ds_groupnorm_stats_workspace_floats(int ninst, int rows_per_inst, int groups);
/* Rows per partial-sum chunk of an instance: the unit of the statistics' summation order.  A function of the instance's SHAPE
 * only -- never of ninst -- so that a batch equals its separate forwards bit for bit; at most ~160 chunks per instance. */
int ds_groupnorm_chunk_rows(int rows_per_inst, int C);
int ds_groupnorm_stats(const void* x, float* mean, float* rstd, float* workspace, int ninst, int rows_per_inst,
                       int C, int groups, float eps, void* stream);
/* y = (x-mean)*rstd*gamma+beta, optional SiLU; fp16 out. */
int ds_groupnorm_apply(const void* x, const float* mean, const float* rstd, const float* gamma,
                       const float* beta, void* y, int ninst, int rows_per_inst, int C, int groups, int silu,
                       void* stream);
/* stats + apply in two launches (the apply reduces the per-chunk partial sums itself; no mean / rstd round trip):
   y = GroupNorm(x) (+ SiLU).  workspace: ds_groupnorm_stats_workspace_floats(...) floats.                            */
int ds_groupnorm_f16(const void* x, const float* gamma, const float* beta, void* y, float* workspace, int ninst,
                     int rows_per_inst, int C, int groups, float eps, int silu, void* stream);
/* the same with a row stride on the input: x row r starts at x + r*ldx elements (ldx >= C, multiple of 8) -- a column slice of
   a wider row-major buffer; y stays dense [rows][C].  (The UNet's skip tensors live in the buffer the decoder side
   concatenates in, torch.cat([h, hs.pop()], dim=1) of openaimodel3d.py:700-703 without the copy.)                   */
int ds_groupnorm_f16_strided(const void* x, int ldx, const float* gamma, const float* beta, void* y, float* workspace,
                             int ninst, int rows_per_inst, int C, int groups, float eps, int silu, void* stream);
/* The general form: x fp16 or fp32 (x_dtype = DS_F16 / DS_F32) with row stride ldx, y fp16 dense [rows][C].  x_f16 (fp32 input
   only, may be NULL): the raw x rounded to fp16, dense [rows][C], written in the same pass -- the matrix-core operand of a
   projection that reads the un-normalised tensor (ResBlock skip_connection, openaimodel3d.py:186-193).  fp32 input is the
   strict-precision mode's residual stream (UNetModel.residual_dtype).                                                  */
int ds_groupnorm_rows(const void* x, int x_dtype, int ldx, const float* gamma, const float* beta, void* y, void* x_f16,
                      float* workspace, int ninst, int rows_per_inst, int C, int groups, float eps, int silu, void* stream);
/* ds_groupnorm_rows as ONE launch that reads the instance ONCE (round 6): a workgroup owns (instance, slab of whole groups) for all rows
 * and keeps its slab in registers (+ LDS) between the statistics and the normalisation -- 6 instead of 10 bytes per fp32 element.  Exists
 * for instances of 257 .. ~5000 rows whose slab fits (ds_groupnorm_onepass_applies: per-frame norms of UNet levels 1-2, joint-T norms of
 * levels 3-4); needs no workspace.  OPT-IN: measured on MI355X it is 0.6-1.0x of the two-launch form per launch but does not pay in the
 * step (profiles/r6_notes.md section 4), and its statistics are summed in another order than ds_groupnorm_rows' (results agree to fp32
 * rounding of mean / rstd).  Replaces GroupNormSpecific / nn.GroupNorm(32, C) like ds_groupnorm_rows (lvdm/basics.py:76-86). */
int ds_groupnorm_onepass_applies(int rows_per_inst, int C, int groups, int x_dtype);
int ds_groupnorm_rows_onepass(const void* x, int x_dtype, int ldx, const float* gamma, const float* beta, void* y, void* x_f16,
                              int ninst, int rows_per_inst, int C, int groups, float eps, int silu, void* stream);
/* ds_groupnorm_rows with the statistics taken from the producer's colstats table (ds_gemm_f16_stats) instead of a pass over x:
   colstats points at the entry of x's first row block and first column; rows_per_inst % 32 == 0.  Two launches: the table is
   folded into per-chunk group sums (1/8 of the bytes of x), the apply reduces those in fp64 as in ds_groupnorm_rows.            */
int ds_groupnorm_rows_colstats(const void* x, int x_dtype, int ldx, const float* colstats, int ld_stats, const float* gamma,
                               const float* beta, void* y, void* x_f16, float* workspace, int ninst, int rows_per_inst, int C,
                               int groups, float eps, int silu, void* stream);
/* nn.LayerNorm(C) eps 1e-5 over each row (attention.py:199-201). x,y fp16 [rows][C]. */
int ds_layernorm(const void* x, const float* gamma, const float* beta, void* y, int rows, int C, float eps,
                 void* stream);
/* the same for x fp16 or fp32 (x_dtype); y fp16. */
int ds_layernorm_rows(const void* x, int x_dtype, const float* gamma, const float* beta, void* y, int rows, int C, float eps,
                      void* stream);
/* The statistics half of nn.LayerNorm (attention.py:199-201): stats[row] = (mean, 1/sqrt(var + eps)) fp32 pairs, two-pass
 * like ds_layernorm.  The normalisation itself is folded into the consumer GEMM by ds_gemm_f16_ln. */
int ds_layernorm_stats(const void* x, float* stats, int rows, int C, float eps, void* stream);

/* softmax(q k^T * scale) v, head_dim 64 (CrossAttention.forward, attention.py:76-127).
 * q: fp16, element (b, i, h, d) at q[(b*nq + i)*ldq + h*64 + d]; k/v likewise with nk rows per kv batch;
 * kv batch of q-batch b is b / kv_batch_div (cross-attention: context shared by the T frames of an eval).
 * out: fp16 [(b*nq+i)*ldo + h*64 + d].  accumulate != 0 adds into out (image-token branch, attention.py:117-124). */
int ds_attention_f16(const void* q, const void* k, const void* v, void* out, int batch, int heads, int nq, int nk,
                     int ldq, int ldk, int ldv, int ldo, int kv_batch_div, float scale, int accumulate,
                     void* stream);

/* Temporal self-attention over T (<=32) tokens per pixel (TemporalTransformer, attention.py:281-373 with
 * only_self_att, no relative position, no causal mask).  qkv rows are NTHWC rows; token t of sequence
 * (b, p) is row (b*T + t)*hw + p.  q/k/v element (row, h, d) at ptr[row*ld + h*64 + d]. */
int ds_temporal_attention_f16(const void* q, const void* k, const void* v, void* out, int nseq_batches, int T,
                              int hw, int heads, int ldq, int ldk, int ldv, int ldo, float scale, void* stream);

/* dst[m][0:c1] = a[m][:], dst[m][c1:c1+c2] = b[m][:]  (torch.cat([h, hs.pop()], dim=1), openaimodel3d.py:701). */
int ds_concat_channels(const void* a, const void* b, void* dst, int rows, int c1, int c2, void* stream);

/* y[r][0:C] = fp16(x[r][0:C]): fp32 rows (stride ldx) -> fp16 rows (stride ldy).  Strict-precision mode: the fp16 matrix-core
 * operand of a convolution that reads the fp32 residual stream directly (Downsample / Upsample, openaimodel3d.py:65-111). */
int ds_cast_rows_f32_f16(const float* x, int ldx, void* y, int ldy, long rows, int C, void* stream);

/* conv-in im2col: x [B][C][T][H][W] (geom dtype) -> patches fp16 [B*T*H*W][kpad], column (ky*3+kx)*C + c,
 * zero padded borders and columns >= 9*C (openaimodel3d.py:421, 682). */
int ds_im2col_in(const void* x, int x_dtype, void* patches, int B, int C, int T, int H, int W, int kpad,
                 void* stream);
/* First-stage decoder (N2): conv_in(post_quant_conv(z * in_scale)) patches -- the CxC channel mix of
 * AutoencoderKL.decode (lvdm/models/autoencoder.py:103-107; in_scale = 1/scale_factor, ddpm3d.py:559) applied inside the
 * image, conv_in's zero padding outside.  wmat fp32 [C][C], bvec fp32 [C], C <= 8. */
int ds_im2col_in_affine(const void* x, int x_dtype, void* patches, int B, int C, int T, int H, int W, int kpad,
                        const float* wmat, const float* bvec, float in_scale, void* stream);
/* p[r][c] = softmax_c(s[r][c] * scale): fp32 scores in, fp16 probabilities out (AttnBlock, ae_modules.py:60-64). */
int ds_softmax_rows(const float* s, void* p, int rows, int cols, int lds, int ldp, float scale, void* stream);
/* First-stage encode tail: scale * DiagonalGaussianDistribution(moments).sample() (lvdm/distributions.py:24-40,
 * ddpm3d.py:458-465).  moments fp32 rows [m][ldm] = (mean[C] | logvar[C]), m = ((b*T+t)*H+y)*W+x; noise / out fp32
 * [B][C][T][H][W]; noise NULL -> the mode. */
int ds_posterior_sample(const float* moments, int ldm, const float* noise, float* out, int B, int C, int T, int H, int W,
                        float scale, void* stream);
/* y rows fp32/fp16 [B*T*H*W][ldy] (first C columns) -> out [B][C][T][H][W] (out_dtype)  (openaimodel3d.py:707). */
int ds_rows_to_ncthw(const void* y, int y_dtype, int ldy, void* out, int out_dtype, int B, int C, int T, int H,
                     int W, void* stream);
/* timestep_embedding (lvdm/models/utils_diffusion.py:8-28): out fp16 [n][dim] = [cos(t*f) | sin(t*f)]. */
int ds_timestep_embedding(const int64_t* t, void* out, int n, int dim, void* stream);
/* y = silu(x), fp16 elementwise (emb_layers SiLU, openaimodel3d.py:172-178). */
int ds_silu_f16(const void* x, void* y, size_t n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * The WIDE operand mode (round 5): the same UNet blocks evaluated as an fp32 computation -- every activation stored in fp32, every
 * matrix product formed from two-term fp16 splits of both operands on the fp16 matrix cores (x = xh + xl / S, S = 2^11; three MFMAs
 * per fragment pair, cross terms in their own fp32 accumulator; csrc/wide.hip).  The reference computes the UNet in fp32
 * (openaimodel3d.py:657-708); with single fp16 operands eps is ~1e-3 away from it, which the highest-noise DDIM updates amplify past
 * the 1e-3 budget on the latent (pipeline/scheduler.py:83-89: config 1's 999 -> 666 update multiplies the guided eps by 1.9).  A
 * precision mode for those steps (ds_unet_config.residual_f32 = 3), ~3x the matrix-core work of the fp16 modes.
 * ---------------------------------------------------------------------------------------------- */
/* S of the split: lo = fp16((x - fp16(x)) * S).  2048. */
float ds_wide_lo_scale(void);
/* hi[i] = fp16(x[i]), lo[i] = fp16((x[i] - hi[i]) * S): the two operand planes of a weight matrix (x fp32 or fp16; ds_unet_pack
 * writes them itself for a handle in the wide mode). */
int ds_split_f16(const void* x, int x_dtype, void* hi, void* lo, size_t n, void* stream);
/* ds_gemm_f16's contract (same ds_gemm_desc, A modes and epilogue flags; nn.Linear / Conv2d / Conv3d call sites listed there) with
 * A fp32 [..][lda] (lda in fp32 elements, multiple of 4), W as two fp16 planes [N][K], fp32 residual [M][ldr], fp32 out [M][ldc]
 * (DS_EPI_OUT_F32 / DS_EPI_RES_F32 are implied).  GELU by erff, SiLU by expf. */
int ds_gemm_wide(const float* A, const void* W_hi, const void* W_lo, const float* bias, const float* residual, float* out,
                 const ds_gemm_desc* desc, void* stream);
/* GroupNorm(groups) (+ SiLU) with fp32 input rows (stride ldx) and dense fp32 output; statistics in fp64, fixed order
 * (basics.py:76-86; openaimodel3d.py:275-292).  scratch: caller buffer of ds_groupnorm_wide_scratch_floats(...) floats, 8-byte aligned
 * (per-chunk fp64 partial sums, then mean / rstd). */
size_t ds_groupnorm_wide_scratch_floats(int ninst, int rows_per_inst, int groups);
int ds_groupnorm_wide(const float* x, int ldx, const float* gamma, const float* beta, float* y, float* scratch, int ninst,
                      int rows_per_inst, int C, int groups, float eps, int silu, void* stream);
/* nn.LayerNorm(C) over dense fp32 rows, fp32 out (attention.py:199-201). */
int ds_layernorm_wide(const float* x, const float* gamma, const float* beta, float* y, long rows, int C, float eps, void* stream);
/* ds_attention_f16 / ds_temporal_attention_f16 on fp32 q / k / v / out, fp32 arithmetic (attention.py:76-127, 281-373). */
int ds_attention_wide(const float* q, const float* k, const float* v, float* out, int batch, int heads, int nq, int nk, int ldq,
                      int ldk, int ldv, int ldo, int kv_batch_div, float scale, int accumulate, void* stream);
int ds_temporal_attention_wide(const float* q, const float* k, const float* v, float* out, int nseq_batches, int T, int hw, int heads,
                               int ldq, int ldk, int ldv, int ldo, float scale, void* stream);
/* fp32 forms of the glue kernels: timestep_embedding (utils_diffusion.py:8-28), SiLU (openaimodel3d.py:172-178), the context cast,
 * conv-in patches (openaimodel3d.py:421, 682). */
int ds_timestep_embedding_f32(const int64_t* t, float* out, int n, int dim, void* stream);
int ds_silu_f32(const float* x, float* y, size_t n, void* stream);
int ds_cast_to_f32(const void* x, int x_dtype, float* y, size_t n, void* stream);
int ds_im2col_in_f32(const void* x, int x_dtype, float* patches, int B, int C, int T, int H, int W, int kpad, void* stream);
/* fp32 forms of the first-stage (AutoencoderKL) glue, for a decode / encode on wide operands: ds_im2col_in_affine with fp32 patches
 * (autoencoder.py:103-107) and ds_softmax_rows with fp32 probabilities (the 512-wide single-head AttnBlock, ae_modules.py:62-64). */
int ds_im2col_in_affine_f32(const void* x, int x_dtype, float* patches, int B, int C, int T, int H, int W, int kpad,
                            const float* wmat, const float* bvec, float in_scale, void* stream);
int ds_softmax_rows_f32(const float* s, float* p, int rows, int cols, int lds, int ldp, float scale, void* stream);

/* ---- conditioning producers: OpenCLIP ViT-H/14 towers and the IP-Adapter Resampler ---- */
/* softmax(q k^T * scale [+ causal mask]) v for head_dim 64 (text tower, open_clip Transformer behind
 * condition.py:216-224; Resampler, ip_resampler.py:62-90) or 80 (image tower, condition.py:358-360); layout as in
 * ds_attention_f16 with head h at column h*head_dim.  causal != 0: key j visible to query i iff j <= i (the text
 * tower's attn_mask, condition.py:220); needs nq == nk. */
int ds_attention_enc_f16(const void* q, const void* k, const void* v, void* out, int batch, int heads, int nq, int nk,
                         int ldq, int ldk, int ldv, int ldo, int head_dim, float scale, int causal, void* stream);
/* y = gelu(x) with the exact erf (nn.GELU: ip_resampler.py:29, open_clip's mlp), fp16 elementwise. */
int ds_gelu_f16(const void* x, void* y, size_t n, void* stream);
/* out fp16 [ntok][width] = table[tokens[i]] + pos[i % ctx]  (condition.py:217-218); table fp16 [vocab][width],
 * pos fp32 [ctx][width]. */
int ds_embed_tokens(const int32_t* tokens, const void* table, const float* pos, void* out, int ntok, int ctx, int width,
                    int vocab, void* stream);
/* out fp16 [nimg*(grid2+1)][width]: token 0 = cls + pos[0], token 1+i = patches[img*grid2 + i] + pos[1+i]
 * (condition.py:346-350); patches fp16, cls / pos fp32. */
int ds_vit_assemble(const void* patches, const float* cls, const float* pos, void* out, int nimg, int grid2, int width,
                    void* stream);
/* FrozenOpenCLIPImageEmbedderV2.preprocess (condition.py:324-332): img [nimg][3][H][W] (fp32 or fp16, values in
 * [-1,1]) -> out fp32 [nimg][3][S][S] = normalize((resize(img) + 1)/2, mean, std); resize = kornia.geometry.resize
 * (bicubic, align_corners=True, antialias: Gaussian pre-blur of sigma (in/out-1)/2 when an axis shrinks).
 * mean / stdv: HOST arrays of 3 floats. */
int ds_clip_preprocess(const void* img, int dtype, float* out, int nimg, int C, int H, int W, int S, int antialias,
                       const float* mean, const float* stdv, void* stream);
/* conv1 (patch embedding, kernel = stride = P, no bias; condition.py:342) as a GEMM: rows fp16 [nimg*(S/P)^2][kpad],
 * column c*P*P + py*P + px (the flattened conv weight's order), columns >= C*P*P zero. */
int ds_patchify(const float* img, void* rows, int nimg, int C, int S, int P, int kpad, void* stream);


/* ------------------------------------------------------------------------------------------------
 * The UNet as ONE call: DiffusionWrapper.forward (lvdm/models/ddpm3d.py:702-712, crossattn branch) ->
 * UNetModel.forward (lvdm/modules/networks/openaimodel3d.py:657-708).  The handle owns the block program (built from the
 * yaml `unet_config.params`, openaimodel3d.py:340-655), the weight packing and the launch sequence over the kernels
 * above; a host in any language with a C FFI can run the UNet.  dynamicscaler_amd.unet.UNetModel binds it (and keeps a
 * Python restatement of the same launch program for per-launch instrumentation; the two are bit-identical).
 * Lifecycle:  create -> load_weight x (every state-dict key) -> packed_bytes / pack -> [workspace_bytes -> forward]* -> destroy.
 * The handle itself lives in host memory (create / destroy); every DEVICE buffer is the caller's.
 * ---------------------------------------------------------------------------------------------- */
typedef struct ds_unet ds_unet;

typedef struct ds_unet_config {          /* the yaml keys of unet_config.params that the VideoCrafter configs set            */
    int32_t in_channels, out_channels, model_channels;
    int32_t num_res_blocks;
    int32_t n_channel_mult, channel_mult[8];
    int32_t n_attention_resolutions, attention_resolutions[8];
    int32_t num_head_channels;           /* 64 (the attention kernels' head width)                                          */
    int32_t transformer_depth, temporal_transformer_depth;
    int32_t context_dim;
    int32_t use_linear;                  /* proj_in / proj_out as nn.Linear (1) or 1x1 conv (0): same arithmetic            */
    int32_t temporal_conv, temporal_attention, addition_attention, use_image_attention, fps_cond;
    int32_t residual_f32;                /* 0: fp16 residual stream; 1: strict mode, the whole stream in fp32 (DS_EPI_RES_F32); 2: fp32
                                            only BETWEEN the blocks (ResBlock / temporal-conv outputs, proj_out + x, conv_in, down /
                                            up-sample, skip tensors), the transformers keep their fp16 inner stream; 3: the WIDE
                                            operand mode -- every activation fp32, split-fp16 products (ds_gemm_wide and the
                                            other *_wide kernels): an fp32 evaluation, for the highest-noise steps; the packed
                                            buffer then holds a second (lo) plane behind every fp16 matrix                        */
    int32_t fold_layernorm;              /* 1: LayerNorm folded into the projections (ds_gemm_f16_ln); forced off by residual_f32 = 1 */
    int32_t gn_from_producer;            /* 1: the GroupNorms whose input is written by the GEMM right before them take their statistics from
                                            that launch's epilogue (ds_gemm_f16_stats) instead of a pass over the tensor.  The partial sums
                                            then follow the producer's tile variant, i.e. the batch size: results stay run-to-run repeatable
                                            but a batch equals its separate forwards only to fp32 rounding.  0 keeps them bit-identical. */
    int32_t temporal_selfatt_only;       /* must be 1 (every VideoCrafter config): the TemporalTransformers attend over the frames only;
                                            0 (cross-attention to the context inside them) is refused with DS_EINVAL            */
} ds_unet_config;

/* Builds the block program; *out receives the handle.  Unsupported option combinations return DS_EINVAL. */
int ds_unet_create(const ds_unet_config* cfg, ds_unet** out);
int ds_unet_destroy(ds_unet* u);
/* Number of state-dict tensors the model expects, and the i-th key / shape (the reference's names, incl. `temopral_conv`). */
int ds_unet_num_weights(const ds_unet* u);
int ds_unet_weight_info(const ds_unet* u, int i, const char** key, int* ndim, int64_t shape[5]);
/* Registers one raw tensor of the reference's state dict: DEVICE pointer, contiguous, DS_F32 or DS_F16, in the reference's
 * layout ([Cout,Cin,3,3] conv weights etc.).  The pointer must stay valid until the work ds_unet_pack enqueues on its stream has
 * COMPLETED (synchronise the stream before freeing the raw tensors: the packing kernels read them asynchronously).  A successful
 * ds_unet_pack forgets the pointers, so packing the same handle again needs every key loaded again. */
int ds_unet_load_weight(ds_unet* u, const char* key, const void* data, int dtype, const int64_t* shape, int ndim);
/* Bytes of the packed-weight buffer, and the packing itself (async on `stream`): fp16 [N][K] GEMM operands (K = tap*Cin + c for
 * convolutions), fused QKV / KV matrices, GEGLU rows interleaved, the time-embedding projections of all ResBlocks in one
 * matrix, LayerNorm folded into the projection it feeds (fp16(gamma*W), column sums, beta.W + b).  `packed` must outlive
 * every forward.  Fails if a key is missing. */
size_t ds_unet_packed_bytes(const ds_unet* u);
int ds_unet_pack(ds_unet* u, void* packed, size_t packed_bytes, void* stream);
/* Layout of the packed buffer, for hosts that want to address single operands (the Python launch program does): item i is
 * `name` at byte `offset`, `bytes` long, dtype DS_F16 (a [rows][bytes/rows/2] matrix) or DS_F32 (a vector of `rows` floats).
 * ds_unet_emb_offset: first column of a ResBlock's slice in the fused time-embedding projection "emb_all" (its total width
 * is the row count of "emb_all.w"). */
int ds_unet_num_packed(const ds_unet* u);
int ds_unet_packed_info(const ds_unet* u, int i, const char** name, size_t* offset, size_t* bytes, long* rows, int* dtype);
int ds_unet_emb_offset(const ds_unet* u, const char* resblock_prefix);
/* Scratch bytes one forward of this geometry needs (peak of the program's own arena: activations, concat buffers, norms'
 * partial sums).  B = batch of independent evaluations, ctx_tokens = 77 (+ image tokens). */
size_t ds_unet_workspace_bytes(ds_unet* u, int B, int T, int H, int W, int ctx_tokens, int cfg_pairs);
/* eps[B][C_out][T][H][W] (fp32) = UNet(x[B][C][T][H][W] (x_dtype), timesteps int64 DEVICE [B], context [B][ctx_tokens][context_dim]
 * (ctx_dtype), fps).  cfg_pairs = n > 0: the batch is [x_1..x_n | x_1..x_n] with only the context differing (classifier-free
 * guidance); the context-free prefix is evaluated once per pair (bit-identical to cfg_pairs = 0).  Async on `stream`,
 * capturable; nothing is allocated: all scratch comes from `workspace`. */
int ds_unet_forward(ds_unet* u, const void* x, int x_dtype, const int64_t* timesteps, const void* context, int ctx_dtype,
                    int ctx_tokens, int fps, int B, int T, int H, int W, int cfg_pairs, void* workspace, size_t workspace_bytes,
                    float* eps, void* stream);
/* Instrumentation of ds_unet_forward (diagnostics and measurement; the model runs without).  launch hook: called on the HOST right
 * before (phase 0) and right after (phase 1) the program enqueues one kernel-family call -- kernel = "gemm" | "attention" |
 * "temporal_attention" | "groupnorm" | "layernorm" | "layernorm_stats" | "cast_rows" | "im2col_in" | ... ; flops = 2 M N K for a gemm,
 * 4 B h nq nk 64 for attention, else 0; info = (a_mode, M, N, K, epilogue, has_residual) for a gemm, (batch, heads, nq, nk) for attention,
 * a few sizes otherwise -- so that a host can bracket every launch with events on `stream` (bench.py's per-launch roofline) or put a
 * launch of its own in front of each (the test suite's LDS / register poison run).  block tap: after every block of the program
 * (reference names: "input_blocks.1.0", "init_attn.0", "middle_block.1", ...) with its output rows (device pointer, row stride ld in
 * elements, dtype DS_F16 / DS_F32) and the geometry at that point; the rows are only valid until the next kernel is enqueued: copy
 * them on `stream` (ds_copy_rows).  NULL clears a hook.  Under stream capture the callbacks run once, at capture time (a copy enqueued
 * by a tap becomes a node of the graph and is refreshed by every replay). */
typedef void (*ds_launch_hook)(void* user, int phase, const char* kernel, double flops, const int32_t* info, int n_info, void* stream);
typedef void (*ds_block_tap)(void* user, const char* block, const void* rows, long nrows, int cols, int ld, int dtype, int B, int T, int H,
                             int W, void* stream);
int ds_unet_set_hooks(ds_unet* u, ds_launch_hook launch, ds_block_tap tap, void* user);
/* dst row r = src row r (row_bytes each; pitches in bytes), device to device, async on `stream`. */
int ds_copy_rows(void* dst, size_t dst_pitch_bytes, const void* src, size_t src_pitch_bytes, size_t row_bytes, size_t rows, void* stream);
/* The launch sequence of such a forward as text, one line per kernel call ("gemm M N K a_mode epilogue lda ldc ldr ..."), without
 * launching anything (no GPU needed): what tests compare with the Python restatement of the program.  Returns the number of
 * bytes written (excluding the terminator), or a negative DS_E* code; buf may be NULL to query the size. */
long ds_unet_trace(ds_unet* u, int B, int T, int H, int W, int ctx_tokens, int cfg_pairs, char* buf, size_t buf_bytes);
/* fp16 <- fp32/fp16 helpers the forward needs on its inputs (exported because a host needs them for its own staging):
 * y[i] = fp16(x[i]). */
int ds_cast_to_f16(const void* x, int x_dtype, void* y, size_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DYNSCALER_HIP_H */
