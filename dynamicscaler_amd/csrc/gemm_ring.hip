// Implicit-GEMM, LDS-DMA ring variant (same math / same descriptor as gemm.hip).
//
// What changes against the register-staged kernel of gemm.hip: the operand tiles no longer pass through VGPRs.
//   * `buffer_load_dwordx4 ... lds` (LDS-DMA) writes 1 KiB per wave-instruction straight into LDS; a padding tap or a
//     tail row gets an out-of-range offset and the hardware zero-fills the LDS bytes (checked on MI355X,
//     tools/exp/lds_dma_test.hip), so the conv / temporal gathers stay branch-free.  The LDS image is lane-linear, so the
//     XOR swizzle moves to the per-lane SOURCE address (chunk p of row r is loaded from logical chunk p ^ ((r>>1)&7))
//     and the fragment reads apply the same involution.
//   * a ring of 4 x 32 KiB stages per workgroup with the loads running THREE K-steps ahead of the MFMAs
//     (up to 96 KiB in flight per CU -- the register-staged kernel keeps ~32 KiB and is latency-bound at ~2.7k cycles per
//     K-step, profiles/r1_notes.md); counted `s_waitcnt vmcnt(8|4|0)` + one raw `s_barrier` per K-step, never a drain.
//   * persistent workgroups (one per CU, 8 waves = 2 per SIMD): the K-steps of all tiles of a workgroup form ONE stream,
//     so the next tile's operands are landing while the current tile's epilogue runs.
//   * 8 waves as 2 (M) x 4 (N), wave tile 64 x 32; epilogue in two 64-row passes through a dedicated 32 KiB scratch.
// LDS: 4 * 32 KiB ring + 32 KiB scratch = 160 KiB (the whole CU).
#include <type_traits>
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int BK = 64;
constexpr int BM = 128, BN = 128;
constexpr int NS = 4;                                   // ring stages
constexpr int STAGE_HALFS = (BM + BN) * BK;             // 16384 halfs = 32 KiB
constexpr int NTHREADS = 512;

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ int swz_chunk(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

struct ARow {
    unsigned base;   // byte offset of the row's source (DENSE/TCONV: m*lda*2; CONV3: image origin)
    int a, b, c;     // CONV3: -, iy0, ix0 ; TCONV: t
    bool valid;
};

template <int AMODE>
__global__ void __launch_bounds__(NTHREADS, 2)
gemm_ring_kernel(const f16* __restrict__ A, const f16* __restrict__ W, const float* __restrict__ bias,
                 const f16* __restrict__ residual, void* __restrict__ out, ds_gemm_desc d, int tiles_m, int tiles_n,
                 unsigned a_bytes, unsigned w_bytes) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    f16* const ring = reinterpret_cast<f16*>(smem);
    float* const sC = reinterpret_cast<float*>(smem + (size_t)NS * STAGE_HALFS * sizeof(f16));   // [64][128] fp32

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int fr = lane & 31, fh = lane >> 5;
    const int nk = d.K / BK;
    const int l_row = lane >> 3, l_p = lane & 7;      // DMA lane -> (row within the 8-row group, physical chunk)

    const int ntiles = tiles_m * tiles_n;
    const int G = gridDim.x;
    int lb = blockIdx.x;
    {   // XCD-aware bijective remap: blocks sharing an L2 work on neighbouring N tiles of one A panel
        const int xcd = lb & 7, q = G >> 3, r = G & 7;
        lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lb >> 3);
    }
    const int my_tiles = lb < ntiles ? (ntiles - lb + G - 1) / G : 0;
    const int total_steps = my_tiles * nk;

    constexpr unsigned OOB = 0x80000000u;   // >= num_records (< 2 GiB); + soffset cannot wrap
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(A), 0, (int)a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(W), 0, (int)w_bytes, 0x00020000);
    const int hl = d.upsample ? 2 * d.hin : d.hin, wl = d.upsample ? 2 * d.win : d.win;
    const int ups = d.upsample ? 1 : 0;

    // ---- DMA cursor over the stream (tile, kt) ----
    ARow ar[2];
    unsigned voff_a[2], voff_b[2];     // per-lane VGPR byte offsets (OOB = out of range -> hardware zero fill)
    unsigned a_csw[2], b_csw[2];       // byte offset of this lane's logical chunk inside a K-step (source-side swizzle)
    int ld_tile = lb, ld_kt = 0, tap = 0, cb = 0, issued = 0;
    unsigned kbytes = 0;

    auto setup_tile = [&](int tile) {
        const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (wave * 2 + i) * 8 + l_row;              // 0..127 inside the A part of the stage
            a_csw[i] = (unsigned)swz_chunk(row, l_p) * 16u;
            const int m = m0 + row;
            ar[i].valid = m < d.M;
            const int mm = ar[i].valid ? m : 0;
            if constexpr (AMODE == DS_A_CONV3) {
                const int hw = d.hout * d.wout;
                const int img = mm / hw, rem = mm - img * hw;
                const int oy = rem / d.wout;
                ar[i].a = img; ar[i].b = oy * d.stride - 1; ar[i].c = (rem - oy * d.wout) * d.stride - 1;
                ar[i].base = (unsigned)(img * d.hin * d.win) * (unsigned)d.lda * 2u;
            } else if constexpr (AMODE == DS_A_TCONV) {
                ar[i].a = (mm / d.hw) % d.t_len; ar[i].b = 0; ar[i].c = 0;
                ar[i].base = (unsigned)mm * (unsigned)d.lda * 2u;
            } else {
                ar[i].a = ar[i].b = ar[i].c = 0;
                ar[i].base = (unsigned)mm * (unsigned)d.lda * 2u;
            }
            const int nrow = (wave * 2 + i) * 8 + l_row;             // 0..127 inside the W part
            b_csw[i] = (unsigned)swz_chunk(nrow, l_p) * 16u;
            const int n = n0 + nrow;
            voff_b[i] = n < d.N ? (unsigned)n * (unsigned)d.K * 2u + b_csw[i] : OOB;
        }
        ld_kt = 0; tap = 0; cb = 0; kbytes = 0;
    };

    // per-lane VGPR offsets: once per tap (conv / temporal) or per tile (dense); the K position goes in the scalar soffset
    auto tap_offsets = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            bool ok = ar[i].valid;
            unsigned off;
            if constexpr (AMODE == DS_A_CONV3) {
                const int ky = tap / 3, kx = tap - ky * 3;
                const int iy = ar[i].b + ky, ix = ar[i].c + kx;
                ok = ok && iy >= 0 && iy < hl && ix >= 0 && ix < wl;
                off = ar[i].base + (unsigned)((iy >> ups) * d.win + (ix >> ups)) * (unsigned)d.lda * 2u;
            } else if constexpr (AMODE == DS_A_TCONV) {
                const int tt = ar[i].a + tap - 1;
                ok = ok && tt >= 0 && tt < d.t_len;
                off = ar[i].base + (unsigned)((tap - 1) * d.hw * d.lda * 2);
            } else {
                off = ar[i].base;
            }
            voff_a[i] = ok ? off + a_csw[i] : OOB;
        }
    };

    // issue the 4 LDS-DMA instructions of this wave for the next stream step into ring slot `slot`
    auto issue = [&](int slot) {
        if (issued >= total_steps) return;
        f16* st = ring + slot * STAGE_HALFS;
        const unsigned soff_a = (unsigned)cb * 2u;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            lds_void* dst = (lds_void*)(st + ((wave * 2 + i) * 8) * BK);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, dst, 16, voff_a[i], soff_a, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            lds_void* dst = (lds_void*)(st + (BM + (wave * 2 + i) * 8) * BK);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, dst, 16, voff_b[i], kbytes, 0, 0);
        }
        ++issued;
        kbytes += BK * 2;
        cb += BK;
        bool new_tap = false;
        if (cb == d.cin) { cb = 0; ++tap; new_tap = true; }
        if (++ld_kt == nk) {
            ld_tile += G;
            if (ld_tile < ntiles) { setup_tile(ld_tile); new_tap = true; }
        }
        if (new_tap) tap_offsets();
    };

    f32x16 acc[2];
    auto compute = [&](const f16* st) {
        const f16* a_base = st + (wm * 64) * BK;
        const f16* b_base = st + (BM + wn * 32) * BK;
        f16x8 af[4][2], bf[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int row = mi * 32 + fr;
                af[kk][mi] = *reinterpret_cast<const f16x8*>(a_base + row * BK + swz_chunk(wm * 64 + row, 2 * kk + fh) * 8);
            }
            bf[kk] = *reinterpret_cast<const f16x8*>(b_base + fr * BK + swz_chunk(wn * 32 + fr, 2 * kk + fh) * 8);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
                acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[kk], af[kk][mi], acc[mi], 0, 0, 0);
    };

    const bool geglu = d.epilogue & DS_EPI_GEGLU;
    const bool silu = d.epilogue & DS_EPI_SILU;
    const bool out_f32 = d.epilogue & DS_EPI_OUT_F32;
    const bool fast = !out_f32 && (d.N % 8 == 0) && (d.ldc % 8 == 0) && (!residual || d.ldr % 8 == 0) &&
                      (!bias || (d.ldbias % 4 == 0 && (reinterpret_cast<uintptr_t>(bias) & 15) == 0));
    constexpr int CCH = BN / 4;   // 16-byte chunks per fp32 scratch row

    // D[i][j]: j = lane&31 is the output row m, i = (reg&3) + 8*(reg>>2) + 4*(lane>>5) the column n.
    auto epilogue = [&](int tile) {
        const int tile_n = tile % tiles_n;
        const int m0 = (tile / tiles_n) * BM, n0 = tile_n * BN;
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {
            if (wm == pass) {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    const int row = mi * 32 + fr;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int cc = (wn * 32 + 8 * g + 4 * fh) >> 2;
                        f32x4 v = {acc[mi][4 * g], acc[mi][4 * g + 1], acc[mi][4 * g + 2], acc[mi][4 * g + 3]};
                        *reinterpret_cast<f32x4*>(sC + row * BN + ((cc ^ (row & (CCH - 1))) << 2)) = v;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();          // raw barrier: __syncthreads() would drain the next tile's LDS-DMA
            __builtin_amdgcn_sched_barrier(0);
            if (fast) {
                auto run = [&](auto ge_tag) {
                    constexpr bool GE = decltype(ge_tag)::value;
                    constexpr int CPR = GE ? BN / 16 : BN / 8;
                    constexpr int RPI = NTHREADS / CPR;        // 32 (64 for GEGLU) rows per sweep
                    constexpr int NIT = 64 / RPI;              // 2 (1)
                    const int ch = tid % CPR, r0 = tid / CPR;
                    const int nloc = ch * 8;
                    const int n = n0 + nloc;
                    if (n >= d.N) return;
                    const long ocol = GE ? (long)tile_n * (BN / 2) + nloc : (long)n;
                    const bool shared_bias = bias && d.bias_rows >= d.M;
                    f16x8 res[NIT];
                    f32x4 pb0[NIT], pb1[NIT];
                    bool ok[NIT];
#pragma unroll
                    for (int u = 0; u < NIT; ++u) {
                        const int m = m0 + pass * 64 + r0 + u * RPI;
                        ok[u] = m < d.M;
                        const long mm = ok[u] ? m : 0;
                        if (residual) res[u] = *reinterpret_cast<const f16x8*>(residual + mm * d.ldr + ocol);
                        if (bias) {
                            const long brow = shared_bias ? 0 : (mm / d.bias_rows) * d.ldbias;
                            pb0[u] = *reinterpret_cast<const f32x4*>(bias + brow + n);
                            pb1[u] = *reinterpret_cast<const f32x4*>(bias + brow + n + 4);
                        }
                    }
                    f32x4 gb0 = {0, 0, 0, 0}, gb1 = {0, 0, 0, 0};
                    if (GE && bias) {
                        gb0 = *reinterpret_cast<const f32x4*>(bias + n + 64);
                        gb1 = *reinterpret_cast<const f32x4*>(bias + n + 68);
                    }
#pragma unroll
                    for (int u = 0; u < NIT; ++u) {
                        if (!ok[u]) continue;
                        const int row = r0 + u * RPI;
                        const int sw = row & (CCH - 1);
                        const f32x4 p0 = *reinterpret_cast<const f32x4*>(sC + row * BN + (((nloc >> 2) ^ sw) << 2));
                        const f32x4 p1 = *reinterpret_cast<const f32x4*>(sC + row * BN + ((((nloc >> 2) + 1) ^ sw) << 2));
                        float v[8] = {p0[0], p0[1], p0[2], p0[3], p1[0], p1[1], p1[2], p1[3]};
                        if (bias) {
                            v[0] += pb0[u][0]; v[1] += pb0[u][1]; v[2] += pb0[u][2]; v[3] += pb0[u][3];
                            v[4] += pb1[u][0]; v[5] += pb1[u][1]; v[6] += pb1[u][2]; v[7] += pb1[u][3];
                        }
                        if (GE) {
                            const f32x4 g0 = *reinterpret_cast<const f32x4*>(sC + row * BN + ((((nloc + 64) >> 2) ^ sw) << 2));
                            const f32x4 g1 = *reinterpret_cast<const f32x4*>(sC + row * BN + (((((nloc + 64) >> 2) + 1) ^ sw) << 2));
                            const float gte[8] = {g0[0] + gb0[0], g0[1] + gb0[1], g0[2] + gb0[2], g0[3] + gb0[3],
                                                  g1[0] + gb1[0], g1[1] + gb1[1], g1[2] + gb1[2], g1[3] + gb1[3]};
#pragma unroll
                            for (int j = 0; j < 8; ++j)
                                v[j] = v[j] * (0.5f * gte[j] * (1.0f + erff(gte[j] * 0.70710678118654752f)));
                        }
                        if (residual) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) v[j] += (float)res[u][j];
                        }
                        if (silu) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) v[j] = v[j] / (1.0f + __expf(-v[j]));
                        }
                        f16x8 o;
#pragma unroll
                        for (int j = 0; j < 8; ++j) o[j] = (f16)v[j];
                        const long m = m0 + pass * 64 + row;
                        *reinterpret_cast<f16x8*>(reinterpret_cast<f16*>(out) + m * d.ldc + ocol) = o;
                    }
                };
                if (geglu) run(std::true_type{}); else run(std::false_type{});
            } else {
                for (int idx = tid; idx < 64 * BN; idx += NTHREADS) {
                    const int row = idx / BN, col = idx - row * BN;
                    const int m = m0 + pass * 64 + row, n = n0 + col;
                    if (m >= d.M || n >= d.N) continue;
                    float v = sC[row * BN + ((((col >> 2) ^ (row & (CCH - 1))) << 2) | (col & 3))];
                    if (bias) v += bias[(long)(m / d.bias_rows) * d.ldbias + n];
                    if (residual) v += (float)residual[(long)m * d.ldr + n];
                    if (silu) v = v / (1.0f + __expf(-v));
                    if (out_f32) reinterpret_cast<float*>(out)[(long)m * d.ldc + n] = v;
                    else reinterpret_cast<f16*>(out)[(long)m * d.ldc + n] = (f16)v;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    if (my_tiles == 0) return;
    setup_tile(lb);
    tap_offsets();
    issue(0); issue(1); issue(2);        // stream steps 0..2 into slots 0..2
    int g = 0;
    for (int tile = lb; tile < ntiles; tile += G) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[mi][j] = 0.0f;
#pragma unroll 1
        for (int kt = 0; kt < nk; ++kt, ++g) {
            // wait until this wave's DMAs of step g have landed; later steps (at most two) stay in flight
            const int later = total_steps - 1 - g;
            if (later >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (later == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();          // every wave's part of step g is in LDS; step g-1 fully consumed
            __builtin_amdgcn_sched_barrier(0);
            issue((g + 3) & (NS - 1));             // refill the slot of step g-1 with step g+3
            compute(ring + (g & (NS - 1)) * STAGE_HALFS);
        }
        epilogue(tile);
    }
}

template <int AMODE>
int launch_ring(const void* A, const void* W, const float* bias, const void* residual, void* out, const ds_gemm_desc& d,
                hipStream_t st) {
    constexpr size_t lds = (size_t)NS * STAGE_HALFS * sizeof(f16) + (size_t)64 * BN * sizeof(float);   // 160 KiB
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ring_kernel<AMODE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            ds_set_error("ds_gemm_f16(ring): hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return DS_ELAUNCH;
        }
        attr_set = true;
    }
    const int tiles_m = ds_cdiv(d.M, BM), tiles_n = ds_cdiv(d.N, BN);
    const long a_rows = AMODE == DS_A_CONV3 ? (long)d.nimg * d.hin * d.win : (long)d.M;
    const long a_bytes = ((a_rows - 1) * d.lda + d.cin) * 2;
    const long w_bytes = (long)d.N * d.K * 2;
    if (a_bytes >= 0x7FFF0000L || w_bytes >= 0x7FFF0000L) {
        ds_set_error("ds_gemm_f16: operand of %ld / %ld bytes exceeds the 4 GiB buffer-addressing range; lower the tile batch", a_bytes, w_bytes);
        return DS_EINVAL;
    }
    const int ntiles = tiles_m * tiles_n;
    const int grid = ntiles < 256 ? ntiles : 256;     // one persistent workgroup per CU
    gemm_ring_kernel<AMODE><<<grid, NTHREADS, lds, st>>>((const f16*)A, (const f16*)W, bias, (const f16*)residual, out, d,
                                                         tiles_m, tiles_n, (unsigned)a_bytes, (unsigned)w_bytes);
    DS_CHECK_LAUNCH("ds_gemm_f16(ring)");
    return DS_OK;
}

}  // namespace

// internal entry used by ds_gemm_f16 (gemm.hip); arguments already validated there
int dsi_gemm_ring(const void* A, const void* W, const float* bias, const void* residual, void* out,
                  const ds_gemm_desc* d, hipStream_t st) {
    if (d->a_mode == DS_A_CONV3) return launch_ring<DS_A_CONV3>(A, W, bias, residual, out, *d, st);
    if (d->a_mode == DS_A_TCONV) return launch_ring<DS_A_TCONV>(A, W, bias, residual, out, *d, st);
    return launch_ring<DS_A_DENSE>(A, W, bias, residual, out, *d, st);
}
