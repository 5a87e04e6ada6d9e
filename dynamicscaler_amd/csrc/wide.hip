// The WIDE operand mode (ds_unet_config.residual_f32 = 3): the UNet with every activation stored in fp32 and every matrix product
// formed from two-term fp16 splits of BOTH operands on the fp16 matrix cores,
//     x = xh + xl / S,   xh = fp16(x),  xl = fp16((x - xh) * S),  S = 2^11        (the same for the weights, split at pack time)
//     x . w = xh.wh + (xh.wl + xl.wh) / S  [+ xl.wl / S^2: dropped, 2^-22 relative]
// three v_mfma_f32_16x16x32_f16 per fragment pair, the two cross terms in their own fp32 accumulator so that the 1/S is applied
// once, after the K sum.  Products then carry ~22 mantissa bits and the evaluation is an fp32 one (eps within 2e-6 of the
// reference's fp32 CPU result instead of ~1e-3): what the highest-noise DDIM updates need, where sqrt((1 - a) / a) and the
// guidance scale amplify the eps error past the 1e-3 budget on the latent (config 1's 999 -> 666 update: DESIGN.md section 5).
// A precision mode for single steps, not the throughput path: 3.7x the time of a default-mode evaluation (profiles/r5_notes.md section 1).
// ds_gemm_wide is bound by the delivery of its fp32 A rows (96 KB per K-step and 256x128 tile through the L2 -> CU fabric); the other
// kernels of the mode (GroupNorm / LayerNorm with fp32 outputs, attention on v_mfma_f32_16x16x4_f32, fp32 glue) are below the GEMM.
#include <limits.h>
#include "common.h"

namespace {

constexpr int BK = 64;          // halfs per K-step of a plane (128-byte LDS rows)
constexpr int BM = 256, BN = 128;          // 8 waves as 4 x 2, wave tile 64 x 64
constexpr float LO_SCALE = 2048.0f, LO_INV = 1.0f / 2048.0f;

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int swz_chunk(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

struct WideArgs {
    const float* A; const f16* Whi; const f16* Wlo; const float* bias; const float* residual; float* out;
    ds_gemm_desc d;
    int tiles_m, tiles_n;
    unsigned a_bytes, w_bytes;
};

__device__ __forceinline__ float silu_exact(float v) { return v / (1.0f + expf(-v)); }
__device__ __forceinline__ float gelu_exact(float g) { return 0.5f * g * (1.0f + erff(g * 0.70710678118654752440f)); }

// out[M,N] (fp32) = gatherA[M,K] (fp32 rows) * (Whi + Wlo / S)[N,K]^T, epilogue as ds_gemm_f16 (bias / per-item bias / fp32 residual /
// SiLU / GEGLU).  256x128 tile, 8 waves as 4x2 (two per SIMD: one wave's staging arithmetic under the other's MFMAs), wave tile
// 64x64 = 4x4 MFMA tiles of 16x16, two accumulator sets.  The kernel is bound by operand delivery (fp32 A rows: 96 KB per K-step
// and workgroup), which is why the tile is 256 rows tall: a 128x128 tile moved 64 KB per K-step for half the FLOPs.
constexpr int WIDE_NT = 512;
// LDS: A hi, A lo [BM rows] (one stage: split in registers on the way in) | two stages of W hi, W lo [BN rows] (the weight planes need
// no arithmetic: LDS-DMA straight from memory, the next K-step's in flight under the MFMAs of this one): 64 + 2 x 32 = 128 KB
constexpr size_t WIDE_LDS = (size_t)(2 * BM + 4 * BN) * BK * sizeof(f16);
typedef __attribute__((address_space(3))) void wide_lds_void;
// (a __device__ helper: with the builtin written inside the templated kernel clang's host pass drops the kernel's launch stub)
__device__ __forceinline__ void wide_dma16(__amdgpu_buffer_rsrc_t rs, f16* dst, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (wide_lds_void*)dst, 16, voff, soff, 0, 0);
}
template <int AMODE>
__global__ void __launch_bounds__(WIDE_NT)
gemm_wide_kernel(WideArgs ka) {
    extern __shared__ __attribute__((aligned(256))) unsigned char wide_smem[];
    f16* const smA[2] = {reinterpret_cast<f16*>(wide_smem), reinterpret_cast<f16*>(wide_smem) + BM * BK};
    // W stage s: hi plane at smW(s, 0), lo plane at smW(s, 1)
    auto smW = [&](int stage, int plane) { return reinterpret_cast<f16*>(wide_smem) + (2 * BM + (2 * stage + plane) * BN) * BK; };
    const ds_gemm_desc d = ka.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int tile_n = blockIdx.x % ka.tiles_n, tile_m = blockIdx.x / ka.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int ld_row = tid >> 3, ld_chunk = tid & 7;      // 64 rows x 8 chunks (8 elements each) per sweep: 4 sweeps of A, 2 of W

    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ka.A), 0, (int)ka.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsH = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(ka.Whi), 0, (int)ka.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsL = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(ka.Wlo), 0, (int)ka.w_bytes, 0x00020000);

    // per staged A row: source byte offset of its first tap / dense row, and what the tap walk needs
    unsigned base[4];
    int ra_[4], rb_[4];
    bool valid[4];
    unsigned b_off[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ld_row + 64 * i;
        valid[i] = m < d.M;
        const int mm = valid[i] ? m : 0;
        if constexpr (AMODE == DS_A_CONV3) {
            const int hw = d.hout * d.wout;
            const int img = mm / hw, rem = mm - img * hw;
            const int oy = rem / d.wout;
            const int pad = d.asym_pad ? 0 : 1;
            ra_[i] = oy * d.stride - pad; rb_[i] = (rem - oy * d.wout) * d.stride - pad;
            base[i] = (unsigned)(img * d.hin * d.win) * (unsigned)d.lda * 4u;
        } else if constexpr (AMODE == DS_A_TCONV) {
            ra_[i] = (mm / d.hw) % d.t_len; rb_[i] = 0;
            base[i] = (unsigned)mm * (unsigned)d.lda * 4u;
        } else {
            ra_[i] = rb_[i] = 0;
            base[i] = (unsigned)mm * (unsigned)d.lda * 4u;
        }
        if (i < 2) {
            // LDS-DMA: a wave instruction lands 8 rows x 128 B at a wave-uniform address (lane l at + 16 l), so the physical chunk is
            // lane & 7 and the XOR swizzle moves to the SOURCE chunk; wave w stages rows 8 (w + 8 i) .. + 7 of each plane
            const int row = 8 * (wave + 8 * i) + (lane >> 3);
            const int n = n0 + row;
            b_off[i] = n < d.N ? (unsigned)n * (unsigned)d.K * 2u + (unsigned)((lane & 7) ^ ((row >> 1) & 7)) * 16u : OOB;
        }
    }
    const int hl = d.upsample ? 2 * d.hin : d.hin, wl = d.upsample ? 2 * d.win : d.win;
    const int ups = d.upsample ? 1 : 0;
    int tap = 0, cb = 0;
    unsigned kbytes = 0;
    unsigned voff_a[4];
    auto tap_offsets = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bool ok = valid[i];
            unsigned off;
            if constexpr (AMODE == DS_A_CONV3) {
                const int ky = tap / 3, kx = tap - ky * 3;
                const int iy = ra_[i] + ky, ix = rb_[i] + kx;
                ok = ok && iy >= 0 && iy < hl && ix >= 0 && ix < wl;
                off = base[i] + (unsigned)((iy >> ups) * d.win + (ix >> ups)) * (unsigned)d.lda * 4u;
            } else if constexpr (AMODE == DS_A_TCONV) {
                const int tt = ra_[i] + tap - 1;
                ok = ok && tt >= 0 && tt < d.t_len;
                off = base[i] + (unsigned)((tap - 1) * d.hw * d.lda * 4);
            } else {
                off = base[i];
            }
            voff_a[i] = ok ? off + (unsigned)ld_chunk * 32u : OOB;
        }
    };
    tap_offsets();

    f32x4 ga[4][2];
    auto load_global = [&](int wstage) {
        const unsigned soff_a = (unsigned)cb * 4u;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            wide_dma16(rsH, smW(wstage, 0) + 8 * (wave + 8 * i) * BK, b_off[i], kbytes);
            wide_dma16(rsL, smW(wstage, 1) + 8 * (wave + 8 * i) * BK, b_off[i], kbytes);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ga[i][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, voff_a[i], soff_a, 0));
            ga[i][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, voff_a[i] == OOB ? OOB : voff_a[i] + 16u, soff_a, 0));
        }
        kbytes += BK * 2;
        cb += BK;
        if (cb == d.cin) {
            cb = 0;
            ++tap;
            if constexpr (AMODE == DS_A_CONV3 || AMODE == DS_A_TCONV) tap_offsets();
        }
    };
    auto store_lds = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = ld_row + 64 * i;
            const int o = row * BK + swz_chunk(row, ld_chunk) * 8;
            f16x8 hi, lo;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float a = ga[i][j >> 2][j & 3];
                const f16 h = (f16)a;
                hi[j] = h;
                lo[j] = (f16)((a - (float)h) * LO_SCALE);
            }
            *reinterpret_cast<f16x8*>(smA[0] + o) = hi;
            *reinterpret_cast<f16x8*>(smA[1] + o) = lo;
        }
    };

    f32x4 accM[4][4], accX[4][4];     // [n16][m16]: main product, cross terms (x S)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) { accM[a][b] = f32x4{0, 0, 0, 0}; accX[a][b] = f32x4{0, 0, 0, 0}; }

    // every wave waits for ITS loads (A rows in registers, its share of the W planes by LDS-DMA), then the barrier makes all shares visible
    auto landed = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    const int nk = d.K / BK;
    load_global(0);
    landed();
    store_lds();
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        const int ws = kt & 1;
        if (more) load_global(ws ^ 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f16x8 wh[4], wl_[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int row = wn * 64 + t * 16 + l15;
                const int o = row * BK + swz_chunk(row, 4 * ks + l4) * 8;
                wh[t] = *reinterpret_cast<const f16x8*>(smW(ws, 0) + o);
                wl_[t] = *reinterpret_cast<const f16x8*>(smW(ws, 1) + o);
            }
#pragma unroll
            for (int m16 = 0; m16 < 4; ++m16) {
                const int row = wm * 64 + m16 * 16 + l15;
                const int o = row * BK + swz_chunk(row, 4 * ks + l4) * 8;
                const f16x8 ah = *reinterpret_cast<const f16x8*>(smA[0] + o);
                const f16x8 al = *reinterpret_cast<const f16x8*>(smA[1] + o);
#pragma unroll
                for (int n16 = 0; n16 < 4; ++n16) {
                    accM[n16][m16] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[n16], ah, accM[n16][m16], 0, 0, 0);
                    accX[n16][m16] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl_[n16], ah, accX[n16][m16], 0, 0, 0);
                    accX[n16][m16] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[n16], al, accX[n16][m16], 0, 0, 0);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's fragment reads are done before anyone overwrites the A planes
        __syncthreads();
        if (more) {
            landed();
            store_lds();
            __syncthreads();
        }
    }

    // ---- epilogue: a lane owns output row (16-row block, l15) and 4 consecutive columns 4*l4.. of every 16x16 tile ----
    const bool geglu = d.epilogue & DS_EPI_GEGLU, silu = d.epilogue & DS_EPI_SILU;
    const float* __restrict__ bias = ka.bias;
    const float* __restrict__ residual = ka.residual;
    float* __restrict__ out = ka.out;
    const int n_out = geglu ? d.N / 2 : d.N;
    const bool vec_ok = (d.ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0) && (n_out % 4 == 0);
#pragma unroll
    for (int m16 = 0; m16 < 4; ++m16) {
        const int m = m0 + wm * 64 + m16 * 16 + l15;
        if (m >= d.M) continue;
        const float* brow = bias ? bias + (long)(m / d.bias_rows) * d.ldbias : nullptr;
#pragma unroll
        for (int n16 = 0; n16 < (4); ++n16) {
            if (geglu && n16 >= 2) break;
            const int n = n0 + wn * 64 + n16 * 16 + 4 * l4;            // column in the N space (x columns with GEGLU)
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = accM[n16][m16][j] + accX[n16][m16][j] * LO_INV;
            int oc = n;
            if (geglu) {
                if (n >= d.N) continue;                                 // N % 64 == 0: the gate column n + 32 exists too
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float x = v[j], g = accM[n16 + 2][m16][j] + accX[n16 + 2][m16][j] * LO_INV;
                    if (brow) { x += brow[n + j]; g += brow[n + 32 + j]; }
                    v[j] = x * gelu_exact(g);
                }
                oc = (n0 + wn * 64) / 2 + n16 * 16 + 4 * l4;
            } else {
                if (n >= d.N) continue;
                if (brow) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (n + j < d.N) v[j] += brow[n + j];
                }
            }
            if (residual) {
#pragma unroll
                for (int j = 0; j < 4; ++j) if (oc + j < n_out) v[j] += residual[(long)m * d.ldr + oc + j];
            }
            if (silu) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = silu_exact(v[j]);
            }
            float* op = out + (long)m * d.ldc + oc;
            if (vec_ok && oc + 3 < n_out) {
                *reinterpret_cast<f32x4*>(op) = f32x4{v[0], v[1], v[2], v[3]};
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) if (oc + j < n_out) op[j] = v[j];
            }
        }
    }
}

template <int AMODE>
int launch_wide(const WideArgs& ka, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_wide_kernel<AMODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)WIDE_LDS);
        if (e != hipSuccess) {
            ds_set_error("ds_gemm_wide: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return DS_ELAUNCH;
        }
        attr_set = true;
    }
    gemm_wide_kernel<AMODE><<<ka.tiles_m * ka.tiles_n, WIDE_NT, WIDE_LDS, st>>>(ka);
    DS_CHECK_LAUNCH("ds_gemm_wide");
    return DS_OK;
}

// ------------------------------------------------------------------------------------------------ norms with fp32 outputs
// GroupNorm statistics in two stages, fp64 accumulators, fixed summation order.  Stage 1: one workgroup per (32-row chunk, instance)
// -- thousands of workgroups, so that the serial row walk of a thread hides behind the others' -- a thread owns columns t, t + 256, ...
// (coalesced row reads), sums its columns over the chunk's rows, the columns of a group are added through LDS in column order ->
// partial[(inst * nchunk + chunk) * groups + g] = (sum, sum of squares).  Stage 2: one wave per (instance, group): lane k adds chunks
// k, k + 64, ... in order, then a fixed xor tree -> (mean, rstd).  (First form: one workgroup per (group, instance) reading its strided
// columns, 210 us per launch; 256-row chunks: 274 us -- both latency-bound; profiles/r5_notes.md.)
constexpr int GNW_CHUNK = 32, GNW_MAXCOLS = 12;     // rows per chunk; columns per thread (C <= 3072: the widest concat input is 2560)
__global__ void __launch_bounds__(256)
gn_wide_partial_kernel(const float* __restrict__ x, int ldx, double* __restrict__ partial, int rows, int C, int groups, int nchunk) {
    const int chunk = blockIdx.x, inst = blockIdx.y, t = threadIdx.x;
    const int r0 = chunk * GNW_CHUNK, r1 = min(rows, r0 + GNW_CHUNK);
    const float* xp = x + ((long)inst * rows + r0) * ldx;
    double s[GNW_MAXCOLS], q[GNW_MAXCOLS];
#pragma unroll
    for (int j = 0; j < GNW_MAXCOLS; ++j) { s[j] = 0.0; q[j] = 0.0; }
    const int ncol = (C - t + 255) / 256;         // columns this thread owns (<= GNW_MAXCOLS)
#pragma unroll 4
    for (int r = 0; r < r1 - r0; ++r) {
#pragma unroll
        for (int j = 0; j < GNW_MAXCOLS; ++j) {
            if (j < ncol) {
                const double v = (double)xp[(long)r * ldx + t + 256 * j];
                s[j] += v;
                q[j] += v * v;
            }
        }
    }
    __shared__ double cs[256 * GNW_MAXCOLS], cq[256 * GNW_MAXCOLS];
#pragma unroll
    for (int j = 0; j < GNW_MAXCOLS; ++j) {
        const int c = t + 256 * j;
        if (c < C) { cs[c] = s[j]; cq[c] = q[j]; }
    }
    __syncthreads();
    const int cpg = C / groups;
    if (t < groups) {
        double a = 0.0, b = 0.0;
        for (int c = t * cpg; c < (t + 1) * cpg; ++c) { a += cs[c]; b += cq[c]; }
        double* o = partial + 2 * (((long)inst * nchunk + chunk) * groups + t);
        o[0] = a; o[1] = b;
    }
}

__global__ void __launch_bounds__(64)
gn_wide_finish_kernel(const double* __restrict__ partial, float* __restrict__ stats, int rows, int C, int groups, int nchunk, float eps) {
    const int idx = blockIdx.x, lane = threadIdx.x;          // one wave per (instance, group)
    const int inst = idx / groups, g = idx - inst * groups;
    double a = 0.0, b = 0.0;
    for (int ch = lane; ch < nchunk; ch += 64) {
        const double* o = partial + 2 * (((long)inst * nchunk + ch) * groups + g);
        a += o[0]; b += o[1];
    }
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) { a += __shfl_xor(a, sh); b += __shfl_xor(b, sh); }
    if (lane == 0) {
        const double n = (double)rows * (C / groups);
        const double mean = a / n;
        double var = b / n - mean * mean;
        if (var < 0.0) var = 0.0;
        stats[2 * idx] = (float)mean;
        stats[2 * idx + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

__global__ void __launch_bounds__(256)
gn_wide_apply_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ stats, const float* __restrict__ gamma,
                     const float* __restrict__ beta, float* __restrict__ y, long nrows, int rows, int C, int groups, int silu) {
    const int cpg = C / groups;
    const long total = nrows * C;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long r = e / C;
        const int c = (int)(e - r * C);
        const long inst = r / rows;
        const float* st = stats + 2 * (inst * groups + c / cpg);
        float v = (x[r * ldx + c] - st[0]) * st[1] * gamma[c] + beta[c];
        if (silu) v = silu_exact(v);
        y[e] = v;
    }
}

// LayerNorm: one wave per row, two passes over the row in registers
__global__ void __launch_bounds__(256)
ln_wide_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ y,
               long rows, int C, float eps) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xp = x + row * C;
    float s = 0.0f;
    for (int c = lane; c < C; c += 64) s += xp[c];
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) s += __shfl_xor(s, sh);
    const float mean = s / (float)C;
    float q = 0.0f;
    for (int c = lane; c < C; c += 64) { const float dlt = xp[c] - mean; q += dlt * dlt; }
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) q += __shfl_xor(q, sh);
    const float rstd = 1.0f / sqrtf(q / (float)C + eps);
    for (int c = lane; c < C; c += 64) y[row * C + c] = (xp[c] - mean) * rstd * gamma[c] + beta[c];
}

// ------------------------------------------------------------------------------------------------ attention in fp32
// softmax(q k^T scale) v, head_dim 64, fp32 throughout (VALU): one thread per query, 64 queries per workgroup, 64-key tiles of K and V
// staged in LDS and read as broadcasts; online softmax in chunks of 16 keys.
__global__ void __launch_bounds__(64)
attention_wide_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, float* __restrict__ out,
                      int nq, int nk, int ldq, int ldk, int ldv, int ldo, int kvdiv, float scale, int accumulate) {
    __shared__ __attribute__((aligned(16))) float Ks[64 * 64], Vs[64 * 64];
    const int lane = threadIdx.x;
    const int h = blockIdx.y, b = blockIdx.z;
    const int qi = blockIdx.x * 64 + lane;
    const bool q_on = qi < nq;
    const long kvb = b / kvdiv;
    float qr[64], o[64];
    {
        const float* qp = q + ((long)b * nq + (q_on ? qi : 0)) * ldq + h * 64;
#pragma unroll
        for (int d4 = 0; d4 < 16; ++d4) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(qp + 4 * d4);
            qr[4 * d4] = t[0] * scale; qr[4 * d4 + 1] = t[1] * scale; qr[4 * d4 + 2] = t[2] * scale; qr[4 * d4 + 3] = t[3] * scale;
        }
    }
#pragma unroll
    for (int d = 0; d < 64; ++d) o[d] = 0.0f;
    float mrun = -INFINITY, lrun = 0.0f;
    for (int key0 = 0; key0 < nk; key0 += 64) {
        __syncthreads();
        // cooperative tile load: sweep r covers keys 4r..4r+3, lane = (key in sweep, 16-byte chunk)
#pragma unroll 4
        for (int r = 0; r < 16; ++r) {
            const int key = 4 * r + (lane >> 4), ch = lane & 15;
            const bool on = key0 + key < nk;
            const long row = kvb * nk + (on ? key0 + key : 0);
            const f32x4 kk = on ? *reinterpret_cast<const f32x4*>(k + row * ldk + h * 64 + 4 * ch) : f32x4{0, 0, 0, 0};
            const f32x4 vv = on ? *reinterpret_cast<const f32x4*>(v + row * ldv + h * 64 + 4 * ch) : f32x4{0, 0, 0, 0};
            *reinterpret_cast<f32x4*>(&Ks[key * 64 + 4 * ch]) = kk;
            *reinterpret_cast<f32x4*>(&Vs[key * 64 + 4 * ch]) = vv;
        }
        __syncthreads();
        const int nkt = min(64, nk - key0);
        for (int c0 = 0; c0 < nkt; c0 += 16) {
            float s[16];
            float mx = -INFINITY;
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                float acc = 0.0f;
#pragma unroll
                for (int d4 = 0; d4 < 16; ++d4) {
                    const f32x4 kk = *reinterpret_cast<const f32x4*>(&Ks[(c0 + jj) * 64 + 4 * d4]);
                    acc = fmaf(qr[4 * d4], kk[0], acc); acc = fmaf(qr[4 * d4 + 1], kk[1], acc);
                    acc = fmaf(qr[4 * d4 + 2], kk[2], acc); acc = fmaf(qr[4 * d4 + 3], kk[3], acc);
                }
                s[jj] = (c0 + jj < nkt) ? acc : -INFINITY;
                mx = fmaxf(mx, s[jj]);
            }
            const float mnew = fmaxf(mrun, mx);          // finite: a chunk holds at least one key
            const float alpha = expf(mrun - mnew);        // 0 on the first chunk (mrun = -inf)
            lrun *= alpha;
#pragma unroll
            for (int d = 0; d < 64; ++d) o[d] *= alpha;
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                const float p = expf(s[jj] - mnew);       // 0 for masked keys
                lrun += p;
#pragma unroll
                for (int d4 = 0; d4 < 16; ++d4) {
                    const f32x4 vv = *reinterpret_cast<const f32x4*>(&Vs[(c0 + jj) * 64 + 4 * d4]);
                    o[4 * d4] = fmaf(p, vv[0], o[4 * d4]); o[4 * d4 + 1] = fmaf(p, vv[1], o[4 * d4 + 1]);
                    o[4 * d4 + 2] = fmaf(p, vv[2], o[4 * d4 + 2]); o[4 * d4 + 3] = fmaf(p, vv[3], o[4 * d4 + 3]);
                }
            }
            mrun = mnew;
        }
    }
    if (!q_on) return;
    const float inv = 1.0f / lrun;
    float* op = out + ((long)b * nq + qi) * ldo + h * 64;
#pragma unroll
    for (int d4 = 0; d4 < 16; ++d4) {
        f32x4 r = f32x4{o[4 * d4] * inv, o[4 * d4 + 1] * inv, o[4 * d4 + 2] * inv, o[4 * d4 + 3] * inv};
        if (accumulate) r += *reinterpret_cast<const f32x4*>(op + 4 * d4);
        *reinterpret_cast<f32x4*>(op + 4 * d4) = r;
    }
}

// The same on the fp32 matrix path: v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulation; 157 TFLOP/s peak, the VALU form above
// is bound by its broadcast LDS reads at ~7).  A workgroup = 4 waves x 16 queries; K / V tiles of 64 keys in LDS with row strides
// (68 / 80 floats) that make the operand reads -- one float per lane: K[key = l % 16][d = 4 kk + l / 16], V[key = 4 kk + l / 16][d = l % 16]
// -- conflict-free; S = Q K^T in the accumulator layout (lane: 4 queries x 1 key per 16-key tile), row maxima / sums by xor shuffles
// over the 16 lanes of a query group, P back through a per-wave LDS strip (accumulator layout -> A-operand layout), O += P V.
// A workgroup = AW_NW waves x 16 queries.
constexpr int AW_KS = 68, AW_VS = 80, AW_PS = 68;
constexpr int AW_NW = 8;                       // waves per workgroup: 128 queries share one staging of a K / V tile
constexpr int AW_NT = 64 * AW_NW, AW_LD = 1024 / AW_NT;      // threads; float4 loads per thread and operand for a 64 x 64 tile
constexpr size_t AW_LDS = (size_t)(64 * (AW_KS + AW_VS) + AW_NW * 16 * AW_PS) * sizeof(float);
__global__ void __launch_bounds__(AW_NT)
attention_wide_mfma_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, float* __restrict__ out,
                           int nq, int nk, int ldq, int ldk, int ldv, int ldo, int kvdiv, float scale, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) float aw_smem[];      // Ks | Vs | one P strip per wave (AW_LDS bytes, dynamic: > 64 KB)
    float* const Ks = aw_smem;
    float* const Vs = aw_smem + 64 * AW_KS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, lg = lane >> 4;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * (16 * AW_NW) + wave * 16;
    const long kvb = b / kvdiv;
    // Q as the A operand of S = Q K^T: lane holds Q[q0 + l16][4 kk + lg], pre-multiplied by the scale
    float qa[16];
    {
        const int qi = min(q0 + l16, nq - 1);
        const float* qp = q + ((long)b * nq + qi) * ldq + h * 64 + lg;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) qa[kk] = qp[4 * kk] * scale;
    }
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0, 0, 0, 0};
    float mrun[4], lrun[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { mrun[r] = -INFINITY; lrun[r] = 0.0f; }
    float* const P = aw_smem + 64 * (AW_KS + AW_VS) + wave * 16 * AW_PS;
    // K / V tiles travel memory -> registers -> LDS; the NEXT tile's loads are issued right after this tile has been published, so they
    // are in flight under its 128 MFMAs
    f32x4 gk[AW_LD], gv[AW_LD];
    auto fetch = [&](int key0_) {
#pragma unroll
        for (int i = 0; i < AW_LD; ++i) {
            const int idx = tid + AW_NT * i, key = idx >> 4, c4 = idx & 15;
            const bool on = key0_ + key < nk;
            const long row = kvb * nk + (on ? key0_ + key : 0);
            gk[i] = on ? *reinterpret_cast<const f32x4*>(k + row * ldk + h * 64 + 4 * c4) : f32x4{0, 0, 0, 0};
            gv[i] = on ? *reinterpret_cast<const f32x4*>(v + row * ldv + h * 64 + 4 * c4) : f32x4{0, 0, 0, 0};
        }
    };
    fetch(0);
    for (int key0 = 0; key0 < nk; key0 += 64) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < AW_LD; ++i) {
            const int idx = tid + AW_NT * i, key = idx >> 4, c4 = idx & 15;
            *reinterpret_cast<f32x4*>(&Ks[key * AW_KS + 4 * c4]) = gk[i];
            *reinterpret_cast<f32x4*>(&Vs[key * AW_VS + 4 * c4]) = gv[i];
        }
        __syncthreads();
        if (key0 + 64 < nk) fetch(key0 + 64);
        // S[kt]: queries 4 lg + r, key 16 kt + l16
        f32x4 sacc[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) sacc[kt] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const float bk = Ks[(16 * kt + l16) * AW_KS + 4 * kk + lg];
                sacc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[kk], bk, sacc[kt], 0, 0, 0);
            }
        }
        float mt[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) mt[r] = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const bool on = key0 + 16 * kt + l16 < nk;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (!on) sacc[kt][r] = -INFINITY;
                mt[r] = fmaxf(mt[r], sacc[kt][r]);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int sh = 1; sh < 16; sh <<= 1) mt[r] = fmaxf(mt[r], __shfl_xor(mt[r], sh));
        }
        float alpha[4], rs[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float mnew = fmaxf(mrun[r], mt[r]);        // finite: a tile holds at least one key
            alpha[r] = expf(mrun[r] - mnew);                  // 0 on the first tile
            mrun[r] = mnew;
            rs[r] = 0.0f;
        }
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = expf(sacc[kt][r] - mrun[r]);   // 0 for masked keys
                rs[r] += pv;
                P[(4 * lg + r) * AW_PS + 16 * kt + l16] = pv;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int sh = 1; sh < 16; sh <<= 1) rs[r] += __shfl_xor(rs[r], sh);
            lrun[r] = lrun[r] * alpha[r] + rs[r];
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[dt][r] *= alpha[r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the wave's own P strip: LDS operations of a wave complete in order
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const float pa = P[l16 * AW_PS + 4 * kk + lg];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const float bv = Vs[(4 * kk + lg) * AW_VS + 16 * dt + l16];
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa, bv, o[dt], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // P is rewritten by the next tile
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int qi = q0 + 4 * lg + r;
        if (qi >= nq) continue;
        const float inv = 1.0f / lrun[r];
        float* op = out + ((long)b * nq + qi) * ldo + h * 64 + l16;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            float val = o[dt][r] * inv;
            if (accumulate) val += op[16 * dt];
            op[16 * dt] = val;
        }
    }
}

// Temporal self-attention over the T tokens of a pixel: one thread per (sequence, head, query token), keys straight from memory
__global__ void __launch_bounds__(256)
tattention_wide_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, float* __restrict__ out,
                       long nthreads, int T, int hw, int heads, int ldq, int ldk, int ldv, int ldo, float scale) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= nthreads) return;
    // idx = ((b * T + tq) * hw + p) * heads + h: neighbouring threads read neighbouring 256-byte head segments
    const int h = (int)(idx % heads);
    const long row_q = idx / heads;                 // (b*T + tq)*hw + p
    const int p = (int)(row_q % hw);
    const long bt = row_q / hw;
    const long b = bt / T;
    float qr[64], o[64];
    const float* qp = q + row_q * ldq + h * 64;
#pragma unroll
    for (int d4 = 0; d4 < 16; ++d4) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(qp + 4 * d4);
        qr[4 * d4] = t[0] * scale; qr[4 * d4 + 1] = t[1] * scale; qr[4 * d4 + 2] = t[2] * scale; qr[4 * d4 + 3] = t[3] * scale;
    }
#pragma unroll
    for (int d = 0; d < 64; ++d) o[d] = 0.0f;
    float mrun = -INFINITY, lrun = 0.0f;
    for (int tk = 0; tk < T; ++tk) {
        const long row = (b * T + tk) * hw + p;
        const float* kp = k + row * ldk + h * 64;
        const float* vp = v + row * ldv + h * 64;
        float s = 0.0f;
#pragma unroll
        for (int d4 = 0; d4 < 16; ++d4) {
            const f32x4 kk = *reinterpret_cast<const f32x4*>(kp + 4 * d4);
            s = fmaf(qr[4 * d4], kk[0], s); s = fmaf(qr[4 * d4 + 1], kk[1], s); s = fmaf(qr[4 * d4 + 2], kk[2], s); s = fmaf(qr[4 * d4 + 3], kk[3], s);
        }
        const float mnew = fmaxf(mrun, s);
        const float alpha = expf(mrun - mnew), pw = expf(s - mnew);
        lrun = lrun * alpha + pw;
#pragma unroll
        for (int d4 = 0; d4 < 16; ++d4) {
            const f32x4 vv = *reinterpret_cast<const f32x4*>(vp + 4 * d4);
            o[4 * d4] = fmaf(pw, vv[0], o[4 * d4] * alpha); o[4 * d4 + 1] = fmaf(pw, vv[1], o[4 * d4 + 1] * alpha);
            o[4 * d4 + 2] = fmaf(pw, vv[2], o[4 * d4 + 2] * alpha); o[4 * d4 + 3] = fmaf(pw, vv[3], o[4 * d4 + 3] * alpha);
        }
        mrun = mnew;
    }
    const float inv = 1.0f / lrun;
    float* op = out + row_q * ldo + h * 64;
#pragma unroll
    for (int d4 = 0; d4 < 16; ++d4)
        *reinterpret_cast<f32x4*>(op + 4 * d4) = f32x4{o[4 * d4] * inv, o[4 * d4 + 1] * inv, o[4 * d4 + 2] * inv, o[4 * d4 + 3] * inv};
}

// ------------------------------------------------------------------------------------------------ fp32 glue
__global__ void timestep_embedding_f32_kernel(const int64_t* __restrict__ t, float* __restrict__ out, int n, int dim) {
    const int half = dim / 2;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * dim) return;
    const int i = idx / dim, j = idx - i * dim;
    float v = 0.0f;
    if (j < 2 * half) {
        const int f = j < half ? j : j - half;
        const float freq = expf(-9.210340371976184f * (float)f / (float)half);
        const float arg = (float)t[i] * freq;
        v = j < half ? cosf(arg) : sinf(arg);
    }
    out[idx] = v;
}

__global__ void __launch_bounds__(256) silu_f32_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = silu_exact(x[i]);
}

template <typename ST>
__global__ void __launch_bounds__(256) cast_f32_kernel(const ST* __restrict__ x, float* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = (float)x[i];
}

template <typename ST>
__global__ void __launch_bounds__(256) split_f16_kernel(const ST* __restrict__ x, f16* __restrict__ hi, f16* __restrict__ lo, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float a = (float)x[i];
        const f16 h = (f16)a;
        hi[i] = h;
        lo[i] = (f16)((a - (float)h) * LO_SCALE);
    }
}

template <typename T>
__global__ void __launch_bounds__(256)
im2col_in_f32_kernel(const T* __restrict__ x, float* __restrict__ patches, int B, int C, int Tn, int H, int W, int kpad) {
    const long total = (long)B * Tn * H * W * kpad;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long m = idx / kpad;
        const int col = (int)(idx - m * kpad);
        float val = 0.0f;
        if (col < 9 * C) {
            const int tap = col / C, c = col - tap * C;
            const int ky = tap / 3, kx = tap - ky * 3;
            long r = m;
            const int xx = (int)(r % W); r /= W;
            const int yy = (int)(r % H); r /= H;
            const int t = (int)(r % Tn);
            const int b = (int)(r / Tn);
            const int iy = yy + ky - 1, ix = xx + kx - 1;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) val = (float)x[((((long)b * C + c) * Tn + t) * H + iy) * W + ix];
        }
        patches[idx] = val;
    }
}

// first-stage decoder input in the wide mode: im2col_in_affine_kernel (misc.hip) with fp32 patches
template <typename T>
__global__ void __launch_bounds__(256)
im2col_in_affine_f32_kernel(const T* __restrict__ x, float* __restrict__ patches, int B, int C, int Tn, int H, int W, int kpad,
                            const float* __restrict__ wmat, const float* __restrict__ bvec, float in_scale) {
    const long total = (long)B * Tn * H * W * kpad;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long m = idx / kpad;
        const int col = (int)(idx - m * kpad);
        float v = 0.0f;
        if (col < 9 * C) {
            const int tap = col / C, c = col - tap * C;
            const int ky = tap / 3, kx = tap - ky * 3;
            long r = m;
            const int xx = (int)(r % W); r /= W;
            const int yy = (int)(r % H); r /= H;
            const int t = (int)(r % Tn);
            const int b = (int)(r / Tn);
            const int iy = yy + ky - 1, ix = xx + kx - 1;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
                v = bvec[c];
                for (int ci = 0; ci < C; ++ci)
                    v += wmat[c * C + ci] * ((float)x[((((long)b * C + ci) * Tn + t) * H + iy) * W + ix] * in_scale);
            }
        }
        patches[idx] = v;
    }
}

// Row softmax with fp32 probabilities (the first-stage AttnBlock in the wide mode, ae_modules.py:62-64): softmax_rows_kernel (misc.hip)
// with expf instead of the hardware exp2 and an fp32 store; one wave per row.
__global__ void __launch_bounds__(256)
softmax_rows_f32_kernel(const float* __restrict__ s, float* __restrict__ p, int rows, int cols, int lds, int ldp, float scale) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long row = (long)blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const float* sr = s + row * lds;
    float* pr = p + row * ldp;
    float m = -1e30f;
    for (int c = lane * 4; c < cols; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(sr + c);
        m = fmaxf(m, fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
    }
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) m = fmaxf(m, __shfl_xor(m, sh));
    float l = 0.0f;
    for (int c = lane * 4; c < cols; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(sr + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) l += expf((v[j] - m) * scale);
    }
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) l += __shfl_xor(l, sh);
    const float inv = 1.0f / l;
    for (int c = lane * 4; c < cols; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(sr + c);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = expf((v[j] - m) * scale) * inv;
        *reinterpret_cast<f32x4*>(pr + c) = o;
    }
}

inline int grid_for(long work) {
    long g = (work + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 65536 ? 65536 : g));
}

}  // namespace

// ------------------------------------------------------------------------------------------------ C ABI
extern "C" float ds_wide_lo_scale(void) { return LO_SCALE; }

extern "C" int ds_gemm_wide(const float* A, const void* W_hi, const void* W_lo, const float* bias, const float* residual, float* out,
                            const ds_gemm_desc* desc, void* stream) {
    DS_CHECK_ARG(A && W_hi && W_lo && out && desc, "ds_gemm_wide: null argument");
    const ds_gemm_desc& d = *desc;
    DS_CHECK_ARG(d.M > 0 && d.N > 0 && d.K > 0, "ds_gemm_wide: M,N,K must be positive (got %d,%d,%d)", d.M, d.N, d.K);
    DS_CHECK_ARG(d.K % BK == 0, "ds_gemm_wide: K=%d must be a multiple of %d", d.K, BK);
    DS_CHECK_ARG(d.cin > 0 && d.cin % BK == 0 && d.K % d.cin == 0, "ds_gemm_wide: cin=%d must be a multiple of %d dividing K=%d", d.cin, BK, d.K);
    DS_CHECK_ARG(d.lda % 4 == 0 && d.lda >= d.cin && (reinterpret_cast<uintptr_t>(A) & 15) == 0, "ds_gemm_wide: lda=%d must be a multiple of 4 and >= cin, A 16-byte aligned", d.lda);
    DS_CHECK_ARG(d.bias_rows > 0 && d.ldc > 0, "ds_gemm_wide: bias_rows and ldc must be positive");
    DS_CHECK_ARG(!bias || d.ldbias >= d.N, "ds_gemm_wide: ldbias=%d must be >= N=%d", d.ldbias, d.N);
    DS_CHECK_ARG(!residual || d.ldr > 0, "ds_gemm_wide: ldr must be positive with a residual");
    if (d.a_mode == DS_A_DENSE) {
        DS_CHECK_ARG(d.cin == d.K, "ds_gemm_wide: dense mode needs cin == K");
    } else if (d.a_mode == DS_A_CONV3) {
        DS_CHECK_ARG(d.K == 9 * d.cin, "ds_gemm_wide: conv3 mode needs K == 9*cin");
        DS_CHECK_ARG(d.stride == 1 || d.stride == 2, "ds_gemm_wide: conv3 stride must be 1 or 2");
        DS_CHECK_ARG(d.nimg > 0 && d.hin > 0 && d.win > 0 && d.hout > 0 && d.wout > 0, "ds_gemm_wide: conv3 dims");
        DS_CHECK_ARG((long)d.nimg * d.hout * d.wout == d.M, "ds_gemm_wide: conv3 M != nimg*hout*wout");
        DS_CHECK_ARG(!(d.upsample && d.stride != 1), "ds_gemm_wide: upsample needs stride 1");
    } else if (d.a_mode == DS_A_TCONV) {
        DS_CHECK_ARG(d.K == 3 * d.cin, "ds_gemm_wide: tconv mode needs K == 3*cin");
        DS_CHECK_ARG(d.t_len > 0 && d.hw > 0 && d.M % (d.t_len * d.hw) == 0, "ds_gemm_wide: tconv M must be nseq*t_len*hw");
    } else {
        DS_CHECK_ARG(false, "ds_gemm_wide: unknown a_mode %d", d.a_mode);
    }
    if (d.epilogue & DS_EPI_GEGLU) DS_CHECK_ARG(d.N % 64 == 0 && !residual, "ds_gemm_wide: GEGLU needs N %% 64 == 0 and no residual");
    const long a_rows = d.a_mode == DS_A_CONV3 ? (long)d.nimg * d.hin * d.win : (long)d.M;
    const long a_bytes = ((a_rows - 1) * d.lda + d.cin) * 4;
    const long w_bytes = (long)d.N * d.K * 2;
    DS_CHECK_ARG(a_bytes < 0x7FFF0000L && w_bytes < 0x7FFF0000L,
                 "ds_gemm_wide: operand of %ld / %ld bytes exceeds the 2 GiB buffer-addressing range; evaluate fewer tiles per call", a_bytes, w_bytes);
    WideArgs ka;
    ka.A = A; ka.Whi = (const f16*)W_hi; ka.Wlo = (const f16*)W_lo; ka.bias = bias; ka.residual = residual; ka.out = out;
    ka.d = d;
    ka.tiles_m = ds_cdiv(d.M, BM); ka.tiles_n = ds_cdiv(d.N, BN);
    ka.a_bytes = (unsigned)a_bytes; ka.w_bytes = (unsigned)w_bytes;
    hipStream_t st = (hipStream_t)stream;
    if (d.a_mode == DS_A_CONV3) return launch_wide<DS_A_CONV3>(ka, st);
    if (d.a_mode == DS_A_TCONV) return launch_wide<DS_A_TCONV>(ka, st);
    return launch_wide<DS_A_DENSE>(ka, st);
}

extern "C" int ds_split_f16(const void* x, int x_dtype, void* hi, void* lo, size_t n, void* stream) {
    DS_CHECK_ARG(x && hi && lo && n > 0, "ds_split_f16: bad argument");
    DS_CHECK_ARG(x_dtype == DS_F16 || x_dtype == DS_F32, "ds_split_f16: x_dtype must be DS_F16 or DS_F32");
    if (x_dtype == DS_F32) split_f16_kernel<float><<<grid_for((long)n), 256, 0, (hipStream_t)stream>>>((const float*)x, (f16*)hi, (f16*)lo, n);
    else split_f16_kernel<f16><<<grid_for((long)n), 256, 0, (hipStream_t)stream>>>((const f16*)x, (f16*)hi, (f16*)lo, n);
    DS_CHECK_LAUNCH("ds_split_f16");
    return DS_OK;
}

extern "C" size_t ds_groupnorm_wide_scratch_floats(int ninst, int rows_per_inst, int groups) {
    if (ninst <= 0 || rows_per_inst <= 0 || groups <= 0) return 0;
    const size_t nchunk = (size_t)ds_cdiv(rows_per_inst, GNW_CHUNK);
    return (size_t)2 * ninst * groups + 2 + (size_t)4 * ninst * nchunk * groups;      // (mean, rstd) floats | 8-byte aligned fp64 partial sums
}

extern "C" int ds_groupnorm_wide(const float* x, int ldx, const float* gamma, const float* beta, float* y, float* scratch, int ninst,
                                 int rows_per_inst, int C, int groups, float eps, int silu, void* stream) {
    DS_CHECK_ARG(x && gamma && beta && y && scratch, "ds_groupnorm_wide: null argument");
    DS_CHECK_ARG(ninst > 0 && ninst <= 65535 && (long)ninst * groups < (1L << 31) && rows_per_inst > 0 && C > 0 && groups > 0 && groups <= 256 && C % groups == 0 && ldx >= C && C <= 256 * GNW_MAXCOLS,
                 "ds_groupnorm_wide: bad sizes (C <= %d, groups <= 256, ninst <= 65535)", 256 * GNW_MAXCOLS);
    DS_CHECK_ARG((reinterpret_cast<uintptr_t>(scratch) & 7) == 0, "ds_groupnorm_wide: scratch must be 8-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int nchunk = ds_cdiv(rows_per_inst, GNW_CHUNK);
    float* stats = scratch;
    double* partial = reinterpret_cast<double*>(scratch + (((size_t)2 * ninst * groups + 1) & ~(size_t)1));
    gn_wide_partial_kernel<<<dim3(nchunk, ninst), 256, 0, st>>>(x, ldx, partial, rows_per_inst, C, groups, nchunk);
    DS_CHECK_LAUNCH("ds_groupnorm_wide");
    gn_wide_finish_kernel<<<ninst * groups, 64, 0, st>>>(partial, stats, rows_per_inst, C, groups, nchunk, eps);
    DS_CHECK_LAUNCH("ds_groupnorm_wide");
    const long nrows = (long)ninst * rows_per_inst;
    gn_wide_apply_kernel<<<grid_for(nrows * C), 256, 0, st>>>(x, ldx, stats, gamma, beta, y, nrows, rows_per_inst, C, groups, silu);
    DS_CHECK_LAUNCH("ds_groupnorm_wide");
    return DS_OK;
}

extern "C" int ds_layernorm_wide(const float* x, const float* gamma, const float* beta, float* y, long rows, int C, float eps, void* stream) {
    DS_CHECK_ARG(x && gamma && beta && y && rows > 0 && C > 0, "ds_layernorm_wide: bad argument");
    ln_wide_kernel<<<(int)((rows + 3) / 4), 256, 0, (hipStream_t)stream>>>(x, gamma, beta, y, rows, C, eps);
    DS_CHECK_LAUNCH("ds_layernorm_wide");
    return DS_OK;
}

extern "C" int ds_attention_wide(const float* q, const float* k, const float* v, float* out, int batch, int heads, int nq, int nk, int ldq,
                                 int ldk, int ldv, int ldo, int kv_batch_div, float scale, int accumulate, void* stream) {
    DS_CHECK_ARG(q && k && v && out, "ds_attention_wide: null argument");
    DS_CHECK_ARG(batch > 0 && batch <= 65535 && heads > 0 && heads <= 65535 && nq > 0 && nk > 0 && kv_batch_div > 0, "ds_attention_wide: bad sizes");
    DS_CHECK_ARG(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && ldo % 4 == 0, "ds_attention_wide: row strides must be multiples of 4");
    DS_CHECK_ARG(((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(out)) & 15) == 0,
                 "ds_attention_wide: operands must be 16-byte aligned");
    // the fp32 matrix-core form; DS_WIDE_ATTN_VALU=1 ("tune" variant) selects the VALU kernel it replaced
    if (DS_TUNE_INT("DS_WIDE_ATTN_VALU", 0) != 0)
        attention_wide_kernel<<<dim3((nq + 63) / 64, heads, batch), 64, 0, (hipStream_t)stream>>>(q, k, v, out, nq, nk, ldq, ldk, ldv, ldo, kv_batch_div,
                                                                                                   scale, accumulate);
    else {
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_wide_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)AW_LDS);
            if (e != hipSuccess) {
                ds_set_error("ds_attention_wide: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
                return DS_ELAUNCH;
            }
            attr_set = true;
        }
        attention_wide_mfma_kernel<<<dim3((nq + 16 * AW_NW - 1) / (16 * AW_NW), heads, batch), AW_NT, AW_LDS, (hipStream_t)stream>>>(q, k, v, out, nq, nk, ldq, ldk, ldv, ldo,
                                                                                                         kv_batch_div, scale, accumulate);
    }
    DS_CHECK_LAUNCH("ds_attention_wide");
    return DS_OK;
}

extern "C" int ds_temporal_attention_wide(const float* q, const float* k, const float* v, float* out, int nseq_batches, int T, int hw, int heads,
                                          int ldq, int ldk, int ldv, int ldo, float scale, void* stream) {
    DS_CHECK_ARG(q && k && v && out, "ds_temporal_attention_wide: null argument");
    DS_CHECK_ARG(nseq_batches > 0 && T > 0 && hw > 0 && heads > 0, "ds_temporal_attention_wide: bad sizes");
    DS_CHECK_ARG(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && ldo % 4 == 0, "ds_temporal_attention_wide: row strides must be multiples of 4");
    const long nthreads = (long)nseq_batches * T * hw * heads;
    tattention_wide_kernel<<<(int)((nthreads + 255) / 256), 256, 0, (hipStream_t)stream>>>(q, k, v, out, nthreads, T, hw, heads, ldq, ldk, ldv, ldo, scale);
    DS_CHECK_LAUNCH("ds_temporal_attention_wide");
    return DS_OK;
}

extern "C" int ds_timestep_embedding_f32(const int64_t* t, float* out, int n, int dim, void* stream) {
    DS_CHECK_ARG(t && out && n > 0 && dim > 0, "ds_timestep_embedding_f32: bad argument");
    timestep_embedding_f32_kernel<<<(n * dim + 255) / 256, 256, 0, (hipStream_t)stream>>>(t, out, n, dim);
    DS_CHECK_LAUNCH("ds_timestep_embedding_f32");
    return DS_OK;
}

extern "C" int ds_silu_f32(const float* x, float* y, size_t n, void* stream) {
    DS_CHECK_ARG(x && y && n > 0, "ds_silu_f32: bad argument");
    silu_f32_kernel<<<grid_for((long)n), 256, 0, (hipStream_t)stream>>>(x, y, n);
    DS_CHECK_LAUNCH("ds_silu_f32");
    return DS_OK;
}

extern "C" int ds_cast_to_f32(const void* x, int x_dtype, float* y, size_t n, void* stream) {
    DS_CHECK_ARG(x && y && n > 0, "ds_cast_to_f32: bad argument");
    DS_CHECK_ARG(x_dtype == DS_F16 || x_dtype == DS_F32, "ds_cast_to_f32: x_dtype must be DS_F16 or DS_F32");
    if (x_dtype == DS_F32) cast_f32_kernel<float><<<grid_for((long)n), 256, 0, (hipStream_t)stream>>>((const float*)x, y, n);
    else cast_f32_kernel<f16><<<grid_for((long)n), 256, 0, (hipStream_t)stream>>>((const f16*)x, y, n);
    DS_CHECK_LAUNCH("ds_cast_to_f32");
    return DS_OK;
}

extern "C" int ds_im2col_in_f32(const void* x, int x_dtype, float* patches, int B, int C, int T, int H, int W, int kpad, void* stream) {
    DS_CHECK_ARG(x && patches, "ds_im2col_in_f32: null argument");
    DS_CHECK_ARG(B > 0 && C > 0 && T > 0 && H > 0 && W > 0, "ds_im2col_in_f32: sizes must be positive");
    DS_CHECK_ARG(kpad >= 9 * C && kpad % 64 == 0, "ds_im2col_in_f32: kpad=%d must be >= 9*C and a multiple of 64", kpad);
    const long work = (long)B * T * H * W * kpad;
    if (x_dtype == DS_F16) im2col_in_f32_kernel<f16><<<grid_for(work), 256, 0, (hipStream_t)stream>>>((const f16*)x, patches, B, C, T, H, W, kpad);
    else if (x_dtype == DS_F32) im2col_in_f32_kernel<float><<<grid_for(work), 256, 0, (hipStream_t)stream>>>((const float*)x, patches, B, C, T, H, W, kpad);
    else DS_CHECK_ARG(false, "ds_im2col_in_f32: bad dtype %d", x_dtype);
    DS_CHECK_LAUNCH("ds_im2col_in_f32");
    return DS_OK;
}

extern "C" int ds_im2col_in_affine_f32(const void* x, int x_dtype, float* patches, int B, int C, int T, int H, int W, int kpad,
                                       const float* wmat, const float* bvec, float in_scale, void* stream) {
    DS_CHECK_ARG(x && patches && wmat && bvec, "ds_im2col_in_affine_f32: null argument");
    DS_CHECK_ARG(B > 0 && C > 0 && C <= 8 && T > 0 && H > 0 && W > 0, "ds_im2col_in_affine_f32: sizes must be positive, C <= 8");
    DS_CHECK_ARG(kpad >= 9 * C && kpad % 64 == 0, "ds_im2col_in_affine_f32: kpad=%d must be >= 9*C and a multiple of 64", kpad);
    const long work = (long)B * T * H * W * kpad;
    if (x_dtype == DS_F16)
        im2col_in_affine_f32_kernel<f16><<<grid_for(work), 256, 0, (hipStream_t)stream>>>((const f16*)x, patches, B, C, T, H, W, kpad, wmat, bvec, in_scale);
    else if (x_dtype == DS_F32)
        im2col_in_affine_f32_kernel<float><<<grid_for(work), 256, 0, (hipStream_t)stream>>>((const float*)x, patches, B, C, T, H, W, kpad, wmat, bvec, in_scale);
    else
        DS_CHECK_ARG(false, "ds_im2col_in_affine_f32: bad dtype %d", x_dtype);
    DS_CHECK_LAUNCH("ds_im2col_in_affine_f32");
    return DS_OK;
}

extern "C" int ds_softmax_rows_f32(const float* s, float* p, int rows, int cols, int lds, int ldp, float scale, void* stream) {
    DS_CHECK_ARG(s && p, "ds_softmax_rows_f32: null argument");
    DS_CHECK_ARG(rows > 0 && cols > 0 && cols % 4 == 0 && lds % 4 == 0 && ldp % 4 == 0 && lds >= cols && ldp >= cols,
                 "ds_softmax_rows_f32: cols / strides must be positive multiples of 4");
    DS_CHECK_ARG(((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(p)) & 15) == 0, "ds_softmax_rows_f32: 16-byte aligned rows");
    softmax_rows_f32_kernel<<<(rows + 3) / 4, 256, 0, (hipStream_t)stream>>>(s, p, rows, cols, lds, ldp, scale);
    DS_CHECK_LAUNCH("ds_softmax_rows_f32");
    return DS_OK;
}
