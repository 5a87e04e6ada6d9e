// Attention kernels for gfx950 (head_dim 64, fp16 in/out, fp32 softmax statistics and accumulation).
//
// 1) ds_attention_f16 -- flash-style softmax(q k^T * scale) v for the spatial self-attention (N = 2560 / 640 /
//    160 / 40 tokens) and the cross-attention (77 text (+16 image) tokens) of SpatialTransformer
//    (lvdm/modules/attention.py:76-127).  q/k/v are read in place from the fused projection outputs (row stride
//    ld*, head h at column h*64), no head split / permute copies.
//      * workgroup = 4 waves; a wave owns QB x 32 query rows; K/V tiles of 64 keys are staged through LDS
//        (global -> VGPR -> LDS, next tile's loads in flight during the MFMAs).
//      * S^T = K Q^T (v_mfma_f32_32x32x16_f16, K fragment as A operand): a lane then owns ONE query column and
//        32 keys in registers -> the softmax max/sum are in-register plus one cross-half shuffle.
//      * O^T += V^T P^T: the S^T accumulator registers, converted to fp16 in place, ARE the B operand of the
//        second MFMA (k order of a 32x32 accumulator: key = 16s + 8(j>>2) + 4*half + (j&3)); V is transposed
//        on the way into LDS so the matching A fragment is two 8-byte LDS reads.
//      * O^T keeps the query on the lane as well, so the online-softmax rescale is a per-lane scalar.
//      * the loop is bound by vector-instruction issue, not by the matrix cores: scalar fp32 softmax ops (no packed ops beside
//        the MFMAs), and the exponent's reference follows the running maximum only when a tile exceeds it by 2^8 -- the rescale
//        of l and O lives in a separate general form of the tile (see DEFER_LOG2 and the tile lambda).
// 2) ds_temporal_attention_f16 -- self-attention over T (<= 32) frames per pixel (TemporalTransformer,
//    attention.py:281-373): 0.1% of the FLOPs, pure HBM traffic; one wave per (pixel, head) on small MFMAs (a VALU form is
//    kept as a diagnostic).
#include <stdlib.h>
#include <type_traits>
#include "common.h"

namespace {

constexpr int HD = 64;
constexpr int KT = 64;          // keys per LDS tile
constexpr int VT_STRIDE = 68;   // halfs per V^T row (64 keys + 4 pad): 34-dword stride -> conflict-free b64 reads
// 1: V tiles stay row-major in LDS ([key][64 d], 128-byte rows like K) and the V^T fragments of the second MFMA come from
// ds_read_b64_tr_b16 (gfx950's transposing LDS read: a 16-lane group reads 4 keys x 16 d and lane i receives column d0 + i of the
// 4 keys) -- no v_perm / 32-bit stores on the way in.  0: round 1's image, V transposed while it is written (build variant "attnvt").
#ifndef DS_ATTN_TRV
#define DS_ATTN_TRV 1
#endif
// 16-byte chunk of row `row` that holds d-chunk `chunk` in the row-major V image: rows key and key + 2 of a 4-key block would
// share 16 banks (a row is 32 banks, a 32-lane half reads 64 B of each of 4 rows), so bit 2 of the chunk flips with (key >> 1) & 1
__device__ __forceinline__ int swz_v(int row, int chunk) { return chunk ^ (((row >> 1) & 1) << 2); }

__device__ __forceinline__ int swz_chunk(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

// The flash loop is VALU-issue bound at head_dim 64 (per 64-key tile and wave: 16 MFMAs = 512 cycles against ~900 cycles of vector
// issue).  Packed fp32 ops (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32) cost several times their two single-issue halves in the
// gaps between MFMAs (MI355X_MICROARCH.md, per-instruction cycle constants), so the softmax is written with scalar ops and the file
// is built with -fno-slp-vectorize (build.py SOURCE_FLAGS: -O3 would pair adjacent adds / multiplies again; inline-asm ops instead
// would sit outside the compiler's hazard handling -- MFMA and v_exp_f32 results need wait states before a consumer).
// Measured (profiles/r4_notes.md section 8): 2560 x 2560 keys 750 -> 802 TFLOP/s.
__device__ __forceinline__ float fma1(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ float add1(float a, float b) { return a + b; }
__device__ __forceinline__ float mul1(float a, float b) { return a * b; }
// the exponent's reference follows the running maximum only when a tile exceeds it by more than this (log2 units): see the loop
constexpr float DEFER_LOG2 = 8.0f;
#ifndef DS_ATTN_WGS
#define DS_ATTN_WGS 3      // workgroups per CU of the QB = 1 kernel (register budget 512 / DS_ATTN_WGS per lane)
#endif

template <int QB>
__global__ void __launch_bounds__(256, QB == 1 ? DS_ATTN_WGS : 1)
attention_kernel(const f16* __restrict__ q, const f16* __restrict__ k, const f16* __restrict__ v, f16* __restrict__ out,
                 int heads, int nq, int nk, int ldq, int ldk, int ldv, int ldo, int kv_batch_div, float scale_log2,
                 int accumulate, int q_tiles) {
    __shared__ __attribute__((aligned(256))) f16 sK2[2][KT * HD];          // double-buffered: one barrier per key tile
#if DS_ATTN_TRV
    __shared__ __attribute__((aligned(256))) f16 sVT2[2][KT * HD];            // (row-major V; the name is kept for the strips below)
#else
    __shared__ __attribute__((aligned(16))) f16 sVT2[2][HD * VT_STRIDE];
#endif

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fh = lane >> 5;
    // XCD-aware block order (bijective for any grid size): workgroup ids go round-robin over the 8 XCDs, each with its own
    // 4 MB L2.  The q-tiles of one (batch, head) all stream the same K / V (655 KB at 2560 keys); in plain order they are
    // spread over all 8 L2s, every L2 sees every head in flight (~25 MB) and K / V come from beyond L2 (7.4 GB per launch
    // measured against 1.26 GB algorithmic, profiles/r2_pmc_mfma_util.json).  Remapped, an XCD works through whole heads.
    int bid = blockIdx.x;
#ifndef DS_ATTN_NO_XCD_REMAP
    {
        const int nwg = gridDim.x, xcd = bid & 7, qn = nwg >> 3, rn = nwg & 7;
        bid = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (bid >> 3);
    }
#endif
    const int qt = bid % q_tiles;
    const int bh = bid / q_tiles;
    const int head = bh % heads, b = bh / heads;
    const int kvb = b / kv_batch_div;
    const int q_base = qt * (128 * QB) + wave * (32 * QB);

    const f16* qp = q + (long)b * nq * ldq + head * HD;
    const f16* kp = k + (long)kvb * nk * ldk + head * HD;
    const f16* vp = v + (long)kvb * nk * ldv + head * HD;

    // ---- Q fragments (B operand: lane = query column, 8 consecutive d per k-step half) ----
    f16x8 qf[QB][4];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int qi = q_base + qb * 32 + fr;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (qi < nq) qf[qb][ks] = *reinterpret_cast<const f16x8*>(qp + (long)qi * ldq + ks * 16 + fh * 8);
            else qf[qb][ks] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }

    f32x16 o[QB][2];
    float mrun[QB], lrun[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        mrun[qb] = -1e30f; lrun[qb] = 0.0f;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int j = 0; j < 16; ++j) o[qb][db][j] = 0.0f;
    }

    // Raw buffer loads: a key row past nk lies beyond num_records and reads as zeros in hardware -- no branch, no
    // select, so the loads of tile t+1 stay in flight behind the MFMAs and the softmax of tile t (a `key < nk ? load : 0`
    // form made the compiler wait for every load right where it was issued: one full memory latency per tile).
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const int ld_row = tid >> 3, ld_chunk = tid & 7;
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(kp), 0, (int)(((long)(nk - 1) * ldk + HD) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(vp), 0, (int)(((long)(nk - 1) * ldv + HD) * 2), 0x00020000);
    unsigned offk[2], offv[2];     // byte offsets of this thread's rows in tile 0; a tile advances them by KT rows
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        offk[i] = (unsigned)((ld_row + 32 * i) * ldk + ld_chunk * 8) * 2u;
#if DS_ATTN_TRV
        offv[i] = (unsigned)((ld_row + 32 * i) * ldv + ld_chunk * 8) * 2u;
#else
        offv[i] = (unsigned)((2 * ld_row + i) * ldv + ld_chunk * 8) * 2u;
#endif
    }
    const unsigned tile_k_bytes = (unsigned)(KT * ldk) * 2u, tile_v_bytes = (unsigned)(KT * ldv) * 2u;
    u32x4 rk[2], rv[2];
    auto load_g = [&](int t) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            rk[i] = __builtin_amdgcn_raw_buffer_load_b128(rsK, offk[i] + (unsigned)t * tile_k_bytes, 0, 0);
            rv[i] = __builtin_amdgcn_raw_buffer_load_b128(rsV, offv[i] + (unsigned)t * tile_v_bytes, 0, 0);
        }
    };
    // K rows go in as 16-byte chunks (thread -> key row tid/8 (+32), chunk tid%8); V is TRANSPOSED on the way in: a
    // thread holds the same 8-column chunk of keys 2p and 2p+1 and writes 8 dwords {V[2p][d], V[2p+1][d]} into
    // V^T[d][2p] (v_perm_b32 pairs the halves; 32-bit LDS stores, no sub-dword writes).
    auto store_l = [&](int buf) {
        f16* sK = sK2[buf];
        f16* sVT = sVT2[buf];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = ld_row + 32 * i;
            *reinterpret_cast<u32x4*>(sK + row * HD + swz_chunk(row, ld_chunk) * 8) = rk[i];
        }
#if DS_ATTN_TRV
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = ld_row + 32 * i;
            *reinterpret_cast<u32x4*>(sVT + row * HD + swz_v(row, ld_chunk) * 8) = rv[i];
        }
#else
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned w = __builtin_amdgcn_perm(rv[1][j >> 1], rv[0][j >> 1], (j & 1) ? 0x07060302u : 0x05040100u);
            *reinterpret_cast<unsigned*>(sVT + (ld_chunk * 8 + j) * VT_STRIDE + 2 * ld_row) = w;
        }
#endif
    };

    const int ntiles = (nk + KT - 1) / KT;
    load_g(0);
    store_l(0);
    __syncthreads();

    // One 64-key tile.  MOVE = true: the general form -- a query whose tile maximum exceeds its reference by more than DEFER_LOG2
    // (always on the first tile) moves the reference there, and l and O of every query are rescaled (by exactly 1 where nothing
    // moved, so a query's result does not depend on its wave neighbours).  MOVE = false: the common form without the 32 rescale
    // multiplies; it returns false BEFORE touching any state when a query of the wave needs the move, and the caller runs the
    // tile again in the general form (the 8 QK^T MFMAs are redone: rare after the first tiles).  Both forms run the same
    // prefetch / LDS store / barrier sequence per tile, so waves of a workgroup may take different forms of the same tile.
    // (A wave-uniform branch around the rescale INSIDE one loop body made the compiler copy the whole O accumulator every tile,
    // whichever side the P V MFMAs were issued from.)
    auto tile = [&](int t, auto move_tag) -> bool {
        constexpr bool MOVE = decltype(move_tag)::value;
        if (t + 1 < ntiles) load_g(t + 1);
        const f16* sK = sK2[t & 1];
        const f16* sVT = sVT2[t & 1];

        // K fragments for this tile are shared by the wave's QB query blocks
        f16x8 kf[2][4];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int row = kb * 32 + fr;
                kf[kb][ks] = *reinterpret_cast<const f16x8*>(sK + row * HD + swz_chunk(row, 2 * ks + fh) * 8);
            }

#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            f32x16 s[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int j = 0; j < 16; ++j) s[kb][j] = 0.0f;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kb][ks], qf[qb][ks], s[kb], 0, 0, 0);
            }
            // tile maximum of the raw scores (scale > 0), keys past nk masked (last tile only)
            if (t == ntiles - 1 && (nk % KT) != 0) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const int key = t * KT + kb * 32 + (j & 3) + 8 * (j >> 2) + 4 * fh;
                        s[kb][j] = key < nk ? s[kb][j] : -1e30f;
                    }
            }
            float mt = -1e30f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int j = 0; j < 16; ++j) mt = fmaxf(mt, s[kb][j]);
            mt = fmaxf(mt, __shfl_xor(mt, 32));
            // Deferred reference: p = 2^((s - mrun) * scale) with mrun <= running maximum <= mrun + DEFER_LOG2 / scale, so
            // p <= 2^DEFER_LOG2 = 256 (fp16 and the fp32 sums hold that with the same relative precision as p <= 1).
            const float mold = mrun[qb];
            const bool move = (mt - mold) * scale_log2 > DEFER_LOG2;
            if constexpr (MOVE) {
                const float mnew = move ? mt : mold;
                const float alpha = __builtin_amdgcn_exp2f((mold - mnew) * scale_log2);
                mrun[qb] = mnew;
                lrun[qb] *= alpha;
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int j = 0; j < 16; ++j) o[qb][db][j] = mul1(o[qb][db][j], alpha);
            } else {
                if (__any(move)) return false;
            }
            // single-issue fp32 ops, two row-sum chains.  The raw v_exp_f32 (no denormal-range fix-up: a result below 2^-126
            // flushes to 0) is what the softmax wants.
            const float mneg = -mrun[qb] * scale_log2;
            float ls0 = 0.0f, ls1 = 0.0f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int j = 0; j < 16; j += 2) {
                    const float p0 = __builtin_amdgcn_exp2f(fma1(s[kb][j], scale_log2, mneg));
                    const float p1 = __builtin_amdgcn_exp2f(fma1(s[kb][j + 1], scale_log2, mneg));
                    s[kb][j] = p0;
                    s[kb][j + 1] = p1;
                    ls0 = add1(ls0, p0);
                    ls1 = add1(ls1, p1);
                }
            lrun[qb] += ls0 + ls1;
            // O^T[d][q] += V^T[d][key] * P^T[key][q]
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int sstep = 0; sstep < 2; ++sstep) {
                    f16x8 pf;
#pragma unroll
                    for (int j = 0; j < 8; ++j) pf[j] = (f16)s[kb][8 * sstep + j];
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        const int drow = db * 32 + fr;
                        const int key0 = kb * 32 + sstep * 16 + 4 * fh;
#if DS_ATTN_TRV
                        // lane 4q + p of a 16-lane group supplies the address of key key0 + q, d = 16-block + 4p .. 4p + 3, and
                        // receives d = db * 32 + fr of the four keys (EXEC is all ones here: no lane-dependent control flow)
                        typedef short s16x4 __attribute__((ext_vector_type(4)));
                        typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
                        const int tq = (lane >> 2) & 3, tp = lane & 3, tg = (lane >> 4) & 1;
                        const int trow = key0 + tq, tchunk = db * 4 + 2 * tg + (tp >> 1);
                        const f16* a0 = sVT + trow * HD + swz_v(trow, tchunk) * 8 + 4 * (tp & 1);
                        const f16* a1 = sVT + (trow + 8) * HD + swz_v(trow + 8, tchunk) * 8 + 4 * (tp & 1);
                        const s16x4 w0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)a0);
                        const s16x4 w1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)a1);
                        const f16x4 v0 = *reinterpret_cast<const f16x4*>(&w0), v1 = *reinterpret_cast<const f16x4*>(&w1);
                        (void)drow;
#else
                        const f16x4 v0 = *reinterpret_cast<const f16x4*>(sVT + drow * VT_STRIDE + key0);
                        const f16x4 v1 = *reinterpret_cast<const f16x4*>(sVT + drow * VT_STRIDE + key0 + 8);
#endif
                        const f16x8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                        o[qb][db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, o[qb][db], 0, 0, 0);
                    }
                }
        }
        if (t + 1 < ntiles) store_l((t + 1) & 1);   // the other buffer: last read in iteration t-1, before its barrier
        __syncthreads();
        return true;
    };
    for (int t = 0; t < ntiles;) {
        tile(t, std::true_type());
        ++t;
        if constexpr (QB == 1)      // (with two query blocks per wave the first could not be taken back: general form throughout)
            while (t < ntiles && tile(t, std::false_type())) ++t;
    }

    // ---- normalise and store: lane owns query fr, d = 32*db + (j&3) + 8*(j>>2) + 4*fh ----
    // Wide form (no accumulate, 16-byte aligned rows): the wave transposes its 32 x 64 block through a private LDS strip (the
    // K / V tiles are dead after the loop's last barrier) so that a lane stores 16 bytes and 8 lanes cover the 128-byte head
    // slice of a row -- whole lines instead of 32 rows x 16 B per store instruction.
#ifndef DS_ATTN_NARROW_STORES
    constexpr int OSW = 72;   // halfs per strip row (64 + 8 pad)
    const bool wide = !accumulate && (ldo % 8 == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    if (wide) {
        f16* so = wave < 2 ? &sK2[0][0] + wave * 32 * OSW : &sVT2[0][0] + (wave - 2) * 32 * OSW;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            const float l = lrun[qb] + __shfl_xor(lrun[qb], 32);
            const float inv = 1.0f / l;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f16x4 w;
#pragma unroll
                    for (int j = 0; j < 4; ++j) w[j] = (f16)(o[qb][db][4 * g + j] * inv);
                    *reinterpret_cast<f16x4*>(so + fr * OSW + db * 32 + 8 * g + 4 * fh) = w;
                }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = (lane >> 3) + 8 * i, ch = lane & 7;
                const int qi = q_base + qb * 32 + row;
                const u32x4 w = *reinterpret_cast<const u32x4*>(so + row * OSW + ch * 8);
                if (qi < nq) *reinterpret_cast<u32x4*>(out + ((long)b * nq + qi) * ldo + head * HD + ch * 8) = w;
            }
            __builtin_amdgcn_wave_barrier();
        }
        return;
    }
#endif
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const float l = lrun[qb] + __shfl_xor(lrun[qb], 32);
        const float inv = 1.0f / l;
        const int qi = q_base + qb * 32 + fr;
        if (qi >= nq) continue;
        f16* op = out + ((long)b * nq + qi) * ldo + head * HD;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int dcol = db * 32 + 8 * g + 4 * fh;
                f16x4 w;
                if (accumulate) {
                    const f16x4 prev = *reinterpret_cast<const f16x4*>(op + dcol);
#pragma unroll
                    for (int j = 0; j < 4; ++j) w[j] = (f16)((float)prev[j] + o[qb][db][4 * g + j] * inv);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) w[j] = (f16)(o[qb][db][4 * g + j] * inv);
                }
                *reinterpret_cast<f16x4*>(op + dcol) = w;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Temporal attention on the matrix cores, T <= 16 * NB (NB = 1: the UNet's 16 frames; NB = 2: up to 32 frames, BASELINE
// config 5 runs the UNet at T = 24): one wave per (batch, pixel, head) item, tokens in NB blocks of 16.
//   S^T = K Q^T   : per (key block, query block) two v_mfma_f32_16x16x32_f16 (K and Q fragments straight from global
//                   memory, 16 B per lane); a lane then owns ONE query per query block (lane&15) and four keys per key
//                   block (4*(lane>>4)..+3): softmax = 4*NB values in registers + two cross-lane steps (xor 16, 32);
//   O^T = V^T P^T : per query block four d-blocks x NB v_mfma_f32_16x16x16f16; the S^T accumulator, converted to fp16 in
//                   place, IS the B operand; V is transposed on its way into a per-wave LDS strip (v_perm pairs, 32-bit
//                   stores);
//   O goes back through a per-wave LDS strip so that rows leave as 16-byte chunks.
// ~120 vector ops per item (NB = 1) instead of ~1250 in the VALU kernel below (which had capped the kernel at 3.3 TB/s).
// ---------------------------------------------------------------------------------------------------------
template <int NB>
__global__ void __launch_bounds__(256)
temporal_attention_mfma_kernel(const f16* __restrict__ q, const f16* __restrict__ k, const f16* __restrict__ v,
                               f16* __restrict__ out, long nseq_total, int T, int hw, int heads, int ldq, int ldk, int ldv,
                               int ldo, float scale_log2) {
    constexpr int TP = 16 * NB;   // token slots
    constexpr int VTS = TP + 4;   // halfs per V^T row (TP keys + 4 pad)
    constexpr int OS = 72;        // halfs per O row (64 d + 8 pad)
    __shared__ __attribute__((aligned(16))) f16 sVT[4][HD * VTS];
    __shared__ __attribute__((aligned(16))) f16 sO[4][TP * OS];
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long item = (long)blockIdx.x * 4 + wave;
    if (item >= nseq_total) return;              // no workgroup barriers below: waves are independent
    const int head = (int)(item % heads);
    const long bp = item / heads;
    const int p = (int)(bp % hw);
    const long b = bp / hw;
    const long row0 = b * T * hw + p;            // row of frame 0; frame t is row0 + t*hw
    const int r16 = lane & 15, c16 = lane >> 4;
    f16* vt = sVT[wave];
    f16* so = sO[wave];

    // fragments: token = 16*block + (lane&15) (clamped: T may be < TP), 8 consecutive d at 32*kh + 8*(lane>>4)
    f16x8 qf[NB][2], kf[NB][2];
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
        const long rowt = row0 + (long)min(16 * blk + r16, T - 1) * hw;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            qf[blk][kh] = *reinterpret_cast<const f16x8*>(q + rowt * ldq + head * HD + 32 * kh + 8 * c16);
            kf[blk][kh] = *reinterpret_cast<const f16x8*>(k + rowt * ldk + head * HD + 32 * kh + 8 * c16);
        }
    }
    // V: lane -> (key pair 8*blk + (lane>>3), 8-column chunk lane&7); V^T[d][2*pair] = {V[2*pair][d], V[2*pair+1][d]}
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
        const int pr = 8 * blk + (lane >> 3), ch = lane & 7;
        const u32x4 v0 = *reinterpret_cast<const u32x4*>(v + (row0 + (long)min(2 * pr, T - 1) * hw) * ldv + head * HD + ch * 8);
        const u32x4 v1 = *reinterpret_cast<const u32x4*>(v + (row0 + (long)min(2 * pr + 1, T - 1) * hw) * ldv + head * HD + ch * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned w = __builtin_amdgcn_perm(v1[j >> 1], v0[j >> 1], (j & 1) ? 0x07060302u : 0x05040100u);
            *reinterpret_cast<unsigned*>(vt + (ch * 8 + j) * VTS + 2 * pr) = w;
        }
    }
    // scores: s[qb][kb], lane: query 16*qb + (lane&15), keys 16*kb + 4*(lane>>4) + r
    f16x4 pf[NB][NB];
    float inv[NB];
#pragma unroll
    for (int qb = 0; qb < NB; ++qb) {
        f32x4 s[NB];
        float m = -1e30f;
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) {
            s[kb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            s[kb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kb][0], qf[qb][0], s[kb], 0, 0, 0);
            s[kb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kb][1], qf[qb][1], s[kb], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (16 * kb + 4 * c16 + r >= T) s[kb][r] = -1e30f;
            m = fmaxf(m, fmaxf(fmaxf(s[kb][0], s[kb][1]), fmaxf(s[kb][2], s[kb][3])));
        }
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        const float mneg = -m * scale_log2;
        float l = 0.0f;
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) {
            float pr4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pr4[r] = __builtin_amdgcn_exp2f(fmaf(s[kb][r], scale_log2, mneg));
                l += pr4[r];
            }
            pf[qb][kb] = f16x4{(f16)pr4[0], (f16)pr4[1], (f16)pr4[2], (f16)pr4[3]};
        }
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        inv[qb] = __builtin_amdgcn_rcpf(l);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's V^T strip is complete (LDS ops of a wave are ordered)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int db = 0; db < 4; ++db) {
        f16x4 vf[NB];
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) vf[kb] = *reinterpret_cast<const f16x4*>(vt + (16 * db + r16) * VTS + 16 * kb + 4 * c16);
#pragma unroll
        for (int qb = 0; qb < NB; ++qb) {
            f32x4 o = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) o = __builtin_amdgcn_mfma_f32_16x16x16f16(vf[kb], pf[qb][kb], o, 0, 0, 0);
            // O^T[d][q]: lane holds q = 16*qb + (lane&15), d = 16*db + 4*(lane>>4) + r
            const f16x4 w = {(f16)(o[0] * inv[qb]), (f16)(o[1] * inv[qb]), (f16)(o[2] * inv[qb]), (f16)(o[3] * inv[qb])};
            *reinterpret_cast<f16x4*>(so + (16 * qb + r16) * OS + 16 * db + 4 * c16) = w;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 2 * NB; ++i) {
        const int idx = lane + 64 * i, tq = idx >> 3, ch = idx & 7;
        if (tq < T)
            *reinterpret_cast<u32x4*>(out + (row0 + (long)tq * hw) * ldo + head * HD + ch * 8) =
                *reinterpret_cast<const u32x4*>(so + tq * OS + ch * 8);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Temporal attention: one wave per (batch, pixel, head); lane = (query frame t = lane & 15 | lane & 31, d-slice).
// TPAD = 16 or 32 query slots; DS = 64 / (64 / TPAD) d values per lane.
// ---------------------------------------------------------------------------------------------------------
template <int TPAD>
__global__ void __launch_bounds__(256)
temporal_attention_kernel(const f16* __restrict__ q, const f16* __restrict__ k, const f16* __restrict__ v,
                          f16* __restrict__ out, long nseq_total, int T, int hw, int heads, int ldq, int ldk, int ldv,
                          int ldo, float scale) {
    constexpr int NSL = 64 / TPAD;   // d-slices per query (4 for T<=16, 2 for T<=32)
    constexpr int DS = HD / NSL;     // d values per lane (16 or 32)
    __shared__ __attribute__((aligned(16))) f16 sKV[4][2][32 * HD];  // per wave: K and V tiles [T][64]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long item = (long)blockIdx.x * 4 + wave;     // (b, p, head)
    const bool active = item < nseq_total;
    const long it = active ? item : 0;
    const int head = (int)(it % heads);
    const long bp = it / heads;
    const int p = (int)(bp % hw);
    const long b = bp / hw;
    const long row0 = b * T * hw + p;                  // row of frame 0; frame t is row0 + t*hw

    f16* sK = sKV[wave][0];
    f16* sV = sKV[wave][1];
    // stage K, V: T rows x 128 bytes = T*8 chunks of 16 B
    for (int c = lane; c < T * 8; c += 64) {
        const int t = c >> 3, ch = c & 7;
        const long row = row0 + (long)t * hw;
        *reinterpret_cast<uint4*>(sK + t * HD + ch * 8) = *reinterpret_cast<const uint4*>(k + row * ldk + head * HD + ch * 8);
        *reinterpret_cast<uint4*>(sV + t * HD + ch * 8) = *reinterpret_cast<const uint4*>(v + row * ldv + head * HD + ch * 8);
    }
    const int tq = lane % TPAD, sl = lane / TPAD;
    const bool qvalid = tq < T;
    float qr[DS];
    {
        const long row = row0 + (long)(qvalid ? tq : 0) * hw;
        const f16* qp = q + row * ldq + head * HD + sl * DS;
#pragma unroll
        for (int c = 0; c < DS / 8; ++c) {
            const f16x8 t8 = *reinterpret_cast<const f16x8*>(qp + c * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) qr[c * 8 + j] = (float)t8[j];
        }
    }
    __syncthreads();

    // scores over all T keys (partial over this lane's d-slice), then reduce across the NSL slices
    float sc[32];
    float mx = -1e30f;
#pragma unroll
    for (int t2 = 0; t2 < 32; ++t2) {
        if (t2 < T) {
            float acc = 0.0f;
#pragma unroll
            for (int c = 0; c < DS / 8; ++c) {
                const f16x8 k8 = *reinterpret_cast<const f16x8*>(sK + t2 * HD + sl * DS + c * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc += qr[c * 8 + j] * (float)k8[j];
            }
#pragma unroll
            for (int sh = TPAD; sh < 64; sh <<= 1) acc += __shfl_xor(acc, sh);
            acc *= scale;
            sc[t2] = acc;
            mx = fmaxf(mx, acc);
        } else {
            sc[t2] = -1e30f;
        }
    }
    float sum = 0.0f;
#pragma unroll
    for (int t2 = 0; t2 < 32; ++t2) {
        const float pz = t2 < T ? __expf(sc[t2] - mx) : 0.0f;
        sc[t2] = pz;
        sum += pz;
    }
    const float inv = 1.0f / sum;
    float oacc[DS];
#pragma unroll
    for (int j = 0; j < DS; ++j) oacc[j] = 0.0f;
#pragma unroll
    for (int t2 = 0; t2 < 32; ++t2) {
        if (t2 < T) {
            const float pz = sc[t2] * inv;
#pragma unroll
            for (int c = 0; c < DS / 8; ++c) {
                const f16x8 v8 = *reinterpret_cast<const f16x8*>(sV + t2 * HD + sl * DS + c * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) oacc[c * 8 + j] += pz * (float)v8[j];
            }
        }
    }
    if (active && qvalid) {
        f16* op = out + (row0 + (long)tq * hw) * ldo + head * HD + sl * DS;
#pragma unroll
        for (int c = 0; c < DS / 8; ++c) {
            f16x8 w;
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] = (f16)oacc[c * 8 + j];
            *reinterpret_cast<f16x8*>(op + c * 8) = w;
        }
    }
}

}  // namespace

extern "C" int ds_attention_f16(const void* q, const void* k, const void* v, void* out, int batch, int heads, int nq,
                                int nk, int ldq, int ldk, int ldv, int ldo, int kv_batch_div, float scale,
                                int accumulate, void* stream) {
    DS_CHECK_ARG(q && k && v && out, "ds_attention_f16: null argument");
    DS_CHECK_ARG(batch > 0 && heads > 0 && nq > 0 && nk > 0, "ds_attention_f16: batch/heads/nq/nk must be positive");
    DS_CHECK_ARG(kv_batch_div > 0 && batch % kv_batch_div == 0, "ds_attention_f16: batch %% kv_batch_div != 0");
    DS_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 4 == 0, "ds_attention_f16: row strides must be multiples of 8 (ldo: 4)");
    DS_CHECK_ARG(ldq >= heads * HD && ldk >= heads * HD && ldv >= heads * HD && ldo >= heads * HD, "ds_attention_f16: row stride < heads*64");
    hipStream_t st = (hipStream_t)stream;
    const float scale_log2 = scale * 1.4426950408889634f;
    const int force_qb = (int)DS_TUNE_INT("DS_ATTN_QB", 0);
    // QB=1 (128 queries / workgroup, 2 waves per SIMD) measured faster than QB=2 at every UNet shape (round 1)
    if (force_qb == 2) {
        const int q_tiles = ds_cdiv(nq, 256);
        attention_kernel<2><<<(long)batch * heads * q_tiles, 256, 0, st>>>((const f16*)q, (const f16*)k, (const f16*)v, (f16*)out,
                                                                         heads, nq, nk, ldq, ldk, ldv, ldo, kv_batch_div, scale_log2, accumulate, q_tiles);
    } else {
        const int q_tiles = ds_cdiv(nq, 128);
        attention_kernel<1><<<(long)batch * heads * q_tiles, 256, 0, st>>>((const f16*)q, (const f16*)k, (const f16*)v, (f16*)out,
                                                                         heads, nq, nk, ldq, ldk, ldv, ldo, kv_batch_div, scale_log2, accumulate, q_tiles);
    }
    DS_CHECK_LAUNCH("ds_attention_f16");
    return DS_OK;
}

extern "C" int ds_temporal_attention_f16(const void* q, const void* k, const void* v, void* out, int nseq_batches,
                                         int T, int hw, int heads, int ldq, int ldk, int ldv, int ldo, float scale,
                                         void* stream) {
    DS_CHECK_ARG(q && k && v && out, "ds_temporal_attention_f16: null argument");
    DS_CHECK_ARG(nseq_batches > 0 && hw > 0 && heads > 0, "ds_temporal_attention_f16: sizes must be positive");
    DS_CHECK_ARG(T >= 1 && T <= 32, "ds_temporal_attention_f16: T=%d must be in [1,32]", T);
    DS_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0, "ds_temporal_attention_f16: strides must be multiples of 8");
    hipStream_t st = (hipStream_t)stream;
    const long items = (long)nseq_batches * hw * heads;
    const int grid = (int)((items + 3) / 4);
    const int valu_kernel = (int)DS_TUNE_INT("DS_TATTN_VALU", 0);   // diagnostic ("tune" build variant)
    if (T <= 16 && !valu_kernel)
        temporal_attention_mfma_kernel<1><<<grid, 256, 0, st>>>((const f16*)q, (const f16*)k, (const f16*)v, (f16*)out, items, T, hw, heads, ldq, ldk, ldv, ldo, scale * 1.4426950408889634f);
    else if (!valu_kernel)
        temporal_attention_mfma_kernel<2><<<grid, 256, 0, st>>>((const f16*)q, (const f16*)k, (const f16*)v, (f16*)out, items, T, hw, heads, ldq, ldk, ldv, ldo, scale * 1.4426950408889634f);
    else if (T <= 16)
        temporal_attention_kernel<16><<<grid, 256, 0, st>>>((const f16*)q, (const f16*)k, (const f16*)v, (f16*)out, items, T, hw, heads, ldq, ldk, ldv, ldo, scale);
    else
        temporal_attention_kernel<32><<<grid, 256, 0, st>>>((const f16*)q, (const f16*)k, (const f16*)v, (f16*)out, items, T, hw, heads, ldq, ldk, ldv, ldo, scale);
    DS_CHECK_LAUNCH("ds_temporal_attention_f16");
    return DS_OK;
}
