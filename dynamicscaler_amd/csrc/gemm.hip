// Implicit-GEMM on the gfx950 matrix cores: out[M,N] = gatherA[M,K] * W[N,K]^T (+bias, +residual, GEGLU, SiLU).
//
// One kernel serves nn.Linear, Conv2d 1x1 / 3x3 (stride 1|2, nearest-x2 upsample folded into the gather)
// and Conv3d (3,1,1): the modes differ only in how a row of the A tile is addressed (DS_A_*).
//
// Structure (CDNA4, wave64; DESIGN.md section 4 has the measurements behind each choice):
//   * tile variants (TileCfg, choose_tile): 256x256 (8 waves as 2x4) and 256x320 (4x2) with ONE workgroup per CU and
//     LDS-DMA staging (buffer_load ... lds, 16 B per lane, the XOR swizzle applied to the SOURCE address, padding taps /
//     tail rows zero-filled by out-of-range offsets), the DMA pieces of K-step k+1 issued between the MFMAs of k;
//     128x128 / 128x64 (4 waves, 2x2) with TWO workgroups per CU and register staging for small launches; 4-stage
//     LDS-DMA forms of those two for grids that leave CUs with a single workgroup.  K-step 64, one barrier per K-step.
//   * LDS tiles are [rows][64 halfs] (128-byte rows) with the 16-byte chunk index XOR-swizzled by
//     ((row >> 1) & 7): every ds_read_b128 lane group of a 32-row fragment read hits 16 distinct 16-byte slots
//     of the 256-byte bank row (conflict-free), and the staging ds_write_b128 (8 lanes per row) is too.
//   * v_mfma_f32_32x32x16_f16 with the WEIGHT fragment as the A operand and the ACTIVATION fragment as B:
//     D[i=n][j=m], so a lane owns one output row m (lane&31) and 4 consecutive columns n per register quad.
//   * epilogue per wave through a private LDS strip, no workgroup barrier: rows are read back as 8-column chunks, bias /
//     per-item bias (time-embedding add) / residual / SiLU applied in fp32, one rounding to fp16, 16-byte stores; GEGLU
//     in registers before the strips; launches without bias / residual transpose fp16 instead of fp32 (the epilogue is
//     LDS-bandwidth bound).
//   * blockIdx is remapped so that the blocks that share an XCD (bid % 8) walk neighbouring N tiles of the same
//     A row panel (L2 reuse; performance only).
#include <type_traits>
#include <stdlib.h>
#include "common.h"

// cache hints of the epilogue's global accesses: DS_EXP_NT bit 0 = fp16-strip output stores, bit 1 = fp32-strip output
// stores, bit 2 = residual loads are non-temporal.  Outputs are never re-read by the launch that writes them; marking
// their stores non-temporal keeps them from displacing A rows / W panels in L2: -1 % of the cfg3 step (mostly in the
// two-stream mode; A/B builds via tools/gpu_ab.sh, profiles/r2_notes.md section 10).  Residual loads: no effect.
#ifndef DS_EXP_NT
#define DS_EXP_NT 3
#endif
#define DS_STORE_NT(ptr, val) __builtin_nontemporal_store((val), (ptr))
#define DS_STORE_PLAIN(ptr, val) (*(ptr) = (val))
#if DS_EXP_NT & 1
#define DS_OUT_STORE_H DS_STORE_NT
#else
#define DS_OUT_STORE_H DS_STORE_PLAIN
#endif
#if DS_EXP_NT & 2
#define DS_OUT_STORE_F DS_STORE_NT
#else
#define DS_OUT_STORE_F DS_STORE_PLAIN
#endif
#if DS_EXP_NT & 4
#define DS_RES_LOAD(ptr) __builtin_nontemporal_load(ptr)
#else
#define DS_RES_LOAD(ptr) (*(ptr))
#endif

// MFMA shape of the K loop.  1: v_mfma_f32_16x16x32_f16 -- same FLOPs per cycle and the same LDS fragment bytes per wave tile as
// the 32x32x16 form, but under these MFMA-dense loops the chip holds a higher clock on it (MI355X_MICROARCH "DVFS give-back"
// item 7; measured here: profiles/r4_notes.md).  The accumulators of a 32x32 output tile are then four 16x16 tiles
// ("quads": q = 2*(row half) + (column half)); the epilogue addresses both layouts through qrow() / qcol() / ACC().
#ifndef DS_MFMA16
#define DS_MFMA16 1
#endif
// 1: the big tiles run persistently with the next tile's first K-step in flight under the epilogue (TileCfg::OVERLAP)
#ifndef DS_PERSIST
#define DS_PERSIST 1
#endif
// 1: the epilogue can also write per-column partial statistics of the stored tile (ds_gemm_f16_stats; profiles/r4_notes.md section 3:
// measured, no gain).  Costs registers in every kernel's epilogue, so the product library is built without it; the "gemmstats" build
// variant (build.py) has it and the kernel tests of the feature run against that library.
#ifndef DS_GEMM_STATS
#define DS_GEMM_STATS 0
#endif

namespace {

constexpr int A_DENSE_LNK = 5;  // internal: as A_DENSE_LN, with the rows' (mean, rstd) computed IN the kernel from the A fragments (ds_gemm_f16_lnk)
constexpr int A_CONV3_TI = 4;   // internal: DS_A_CONV3, stride 1, no upsample, K walked channel-chunk-major with the 9 TAPS INNERMOST
constexpr int A_DENSE_LN = 3;   // internal template value: DS_A_DENSE addressing + the LayerNorm fold after the K loop (ds_gemm_f16_ln)
constexpr int BK = 64;  // halfs per K-step (128-byte LDS rows, 8 chunks of 16 bytes)

__device__ __forceinline__ int swz_chunk(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

// Epilogue activations.  The epilogue is vector-issue bound, so these avoid the IEEE-division sequence (10 VALU ops;
// v_rcp_f32 is accurate to 1 ulp) and the libm erff polynomial (~40 ops).
__device__ __forceinline__ float fast_silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
// exact-erf GELU, two values at a time (packed fp32 math):  gelu(g) = g/2 * (1 + erf(g/sqrt2)) and
// 1 + erf(g/sqrt2) = E(|g|) for g < 0, 2 - E(|g|) otherwise, with E(a) = erfc(a/sqrt2) = 2^-q(a).  q is a degree-6
// polynomial fitted on [0, 6] (max error 6.5e-5 in q = 4.5e-5 RELATIVE in erfc, so the small negative tail keeps its
// relative accuracy; |gelu error| <= 6.5e-6 absolute, two orders below the fp16 rounding of the result).  One v_exp_f32
// and 7 packed ops per pair of values.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 fast_gelu_erf2(f32x2 g) {
    const f32x2 a = {fminf(fabsf(g[0]), 6.0f), fminf(fabsf(g[1]), 6.0f)};
    f32x2 q = {-2.2990398065303452e-05f, -2.2990398065303452e-05f};
    q = __builtin_elementwise_fma(q, a, f32x2{0.0006110938265919685f, 0.0006110938265919685f});
    q = __builtin_elementwise_fma(q, a, f32x2{-0.007195422891527414f, -0.007195422891527414f});
    q = __builtin_elementwise_fma(q, a, f32x2{0.05118447542190552f, 0.05118447542190552f});
    q = __builtin_elementwise_fma(q, a, f32x2{0.46127405762672424f, 0.46127405762672424f});
    q = __builtin_elementwise_fma(q, a, f32x2{1.150172233581543f, 1.150172233581543f});
    q = __builtin_elementwise_fma(q, a, f32x2{6.5313492086716e-05f, 6.5313492086716e-05f});
    const f32x2 e = {__builtin_amdgcn_exp2f(-q[0]), __builtin_amdgcn_exp2f(-q[1])};
    const f32x2 t = {g[0] < 0.0f ? e[0] : 2.0f - e[0], g[1] < 0.0f ? e[1] : 2.0f - e[1]};
    return (g * f32x2{0.5f, 0.5f}) * t;
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

// LDS-DMA: 16 bytes per lane, lane l lands at dst + 16*l (dst wave-uniform).  Kept in a __device__ helper: with the
// builtin written directly inside the templated kernel, clang's host pass silently drops the kernel's launch stub.
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, f16* dst, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)dst, 16, voff, soff, 0, 0);
}

// Diagnostic build only (tools/gemm_stamps.py compiles this file with -DDS_GEMM_STAMPS into its own library):
// s_memtime stamps of the kernel phases, written to a debug buffer nothing else reads.  The product build has no stamps.
#ifdef DS_GEMM_STAMPS
__device__ unsigned long long* ds_dbg_stamps = nullptr;
#define DS_STAMP(i)                                                                                   \
    do {                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        unsigned long long t_;                                                                        \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                    \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        if (ds_dbg_stamps && threadIdx.x == 0) {                                                      \
            ds_dbg_stamps[(size_t)blockIdx.x * 8 + (i)] = t_;                                         \
            if ((i) == 0 || (i) == 4) ds_dbg_stamps[(size_t)blockIdx.x * 8 + 5 + (i) / 4] = __builtin_amdgcn_s_memrealtime(); \
        }                                                                                             \
    } while (0)
#else
#define DS_STAMP(i) do {} while (0)
#endif

struct RowInfo {   // per staged A row, computed once
    unsigned base; // BYTE offset of the row's source (DENSE/TCONV: m*lda*2; CONV3: image origin)
    int a, b, c;   // CONV3: img, oy, ox ; TCONV: t ; validity
    bool valid;
};

// Compile-time shape of one kernel variant: block tile BM x BN, WGM x WGN waves.
template <int BM, int BN, int WGM, int WGN, int NS = 2>
struct TileCfg {
    static constexpr int NSTAGE = NS;                         // LDS stages of BK halfs (2 = double buffering)
    static constexpr int NT = 64 * WGM * WGN;                 // threads
    static constexpr int WM = BM / WGM, WN = BN / WGN;        // wave tile
    static constexpr int TM = WM / 32, TN = WN / 32;          // 32x32 MFMA tiles per wave
    static constexpr int LROWS = NT / 8;                      // rows staged per sweep (8 lanes x 16 B per 128-B row)
    static constexpr int AR = BM / LROWS, BR = BN / LROWS;    // staged rows per thread
    static constexpr size_t STAGE1 = (size_t)(BM + BN) * BK * sizeof(f16);   // one stage: [BM rows of A | BN rows of W], 128 B each
    // NS == 1 selects the SPLIT-RING form (round 6): TWO A stages and ONE W stage -- [A stage 0 | A stage 1 | W] -- so that a 4-wave
    // workgroup with LDS-DMA staging fits in half a CU's LDS and two of them share a CU: they run out of step, and one's prologue /
    // epilogue (memory- and vector-bound) proceeds under the other's K loop.  W (L2-resident) is re-loaded between two barriers at the
    // end of every K-step; the A rows of K-step k+2 go out at the same point (one step ahead of their use).
    static constexpr bool SW = NS == 1;
    static constexpr size_t STAGE = SW ? (size_t)(2 * BM + BN) * BK * sizeof(f16) : (size_t)NS * STAGE1;
    // epilogue: every wave transposes its own accumulators through a private LDS strip of 32 rows x NG MFMA tiles
    // (fp32, +4 floats of padding per row) -- no workgroup barrier after the main loop
    static constexpr int NG0 = TN <= 4 ? TN : (TN + (TN + 3) / 4 - 1) / ((TN + 3) / 4);   // tiles per column group
    static constexpr size_t LDS0 = STAGE > (size_t)(NT / 64) * 32 * (32 * NG0 + 4) * sizeof(float) ? STAGE : (size_t)(NT / 64) * 32 * (32 * NG0 + 4) * sizeof(float);
    static constexpr int WG_PER_CU = LDS0 <= 81920 && NT <= 256 ? 2 : 1;
    // OVERLAP (the two-stage LDS-DMA tiles, one workgroup per CU): the kernel is persistent -- a workgroup walks tiles
    // bid, bid + grid, ... -- and the FIRST K-step of its next tile is issued into stage 0 before the epilogue of the current one,
    // whose strips therefore live BEHIND stage 0 (in stage 1 and the tail of the allocation), two tiles per column group at most
    static constexpr bool OVERLAP = WG_PER_CU == 1 && NS == 2 && DS_PERSIST != 0;
    // HALF (round 6, build variant "epihalf"; the 256 x 320 persistent tile): strips of SIXTEEN rows x ALL the wave's columns instead of
    // 32 rows x two tiles -- the same bytes behind stage 0, but a wave then stores its whole 160-column width at once: 320-byte row pieces
    // instead of 128 + 128 + 64.  A copy kernel with this tile's patterns moves the launch's bytes 7-10 % faster with 320-byte pieces
    // (profiles/r6_notes.md section 1); in the GEMM it measured NO gain (every K = 320 ... 1280 launch within -1 ... +3.6 %, the step
    // 463.8 +- 1 ms with against 463.3 +- 1.5 without: 60 of 64 lanes active, six sweeps of 3 rows instead of four of 8), same bits.  OFF.
#ifndef DS_EPI_HALF
#define DS_EPI_HALF 0
#endif
    static constexpr bool HALF = OVERLAP && NG0 > 2 && TN <= 5 && DS_MFMA16 != 0 && DS_GEMM_STATS == 0 && DS_EPI_HALF != 0;
    static constexpr int NG = HALF ? TN : (OVERLAP && NG0 > 2 ? 2 : NG0);
    static constexpr int SROWS = HALF ? 16 : 32;               // rows of a strip
    static constexpr int STR = 32 * NG + 4;                    // floats per strip row
    static constexpr size_t EPI = (size_t)(NT / 64) * SROWS * STR * sizeof(float);
    static constexpr size_t STRIP_OFF = OVERLAP ? STAGE1 : 0;
    static constexpr size_t LDS = STAGE > STRIP_OFF + EPI ? STAGE : STRIP_OFF + EPI;
    // fragment scheduling: all four k-slices of a K-step up front when that is <= 16 fragments, else one k-slice
    // ahead (double-buffered fragment registers)
    static constexpr bool HOIST_ALL = 4 * (TM + TN) <= 16;
    // one workgroup per CU: operands go global -> LDS by LDS-DMA (no VGPR staging, no ds_write phase in which all
    // eight waves would leave the matrix pipe idle together); two per CU: register staging (the partner workgroup's
    // MFMAs cover the store phase)
    static constexpr bool DMA = WG_PER_CU == 1 || SW;
    static_assert(!SW || WG_PER_CU == 2, "the split-ring form exists for two workgroups per CU");
    static_assert(BM % (32 * WGM) == 0 && BN % (32 * WGN) == 0 && BM % LROWS == 0 && BN % LROWS == 0, "tile shape");
    static_assert(LDS <= 163840, "LDS budget");
};

// Every launch-invariant argument in ONE by-value struct = the kernarg segment from offset 0.  A persistent workgroup re-reads it
// per tile through a laundered pointer: scalar loads instead of ~90 SGPRs kept live around the tile loop (which spilled).
struct GemmArgs {
    const f16* A; const f16* W; const float* bias; const f16* residual; void* out;
    ds_gemm_desc d;
    int tiles_m, tiles_n;
    unsigned a_bytes, w_bytes;
    const float* ln_stats; const float* ln_colsum;
    float ln_eps;
    int group_m;
    float2* colstats;
    int ld_stats;
    int m_base, m_total;      // a launch that covers rows [m_base, m_base + d.M) of a larger one of m_total rows (gemm_entry's cut along M):
                              // conv / temporal row geometry and per-item bias rows are those of the WHOLE launch; 0, d.M otherwise
};
typedef const GemmArgs __attribute__((address_space(4)))* GemmArgsPtr;

// DS_GEMM_VGPR_CAP (A/B builds only): a register budget below what two waves per SIMD allow, so that waves of ANOTHER kernel (the
// memory-bound norms on the second stream) can be resident on a CU next to a persistent GEMM workgroup
#ifdef DS_GEMM_VGPR_CAP
#define DS_GEMM_KERNEL_ATTR __attribute__((amdgpu_num_vgpr(DS_GEMM_VGPR_CAP)))
#else
#define DS_GEMM_KERNEL_ATTR
#endif
template <int BM, int BN, int WGM, int WGN, int AMODE, int NS>
__global__ void __launch_bounds__((TileCfg<BM, BN, WGM, WGN, NS>::NT), (TileCfg<BM, BN, WGM, WGN, NS>::WG_PER_CU)) DS_GEMM_KERNEL_ATTR
gemm_f16_kernel(GemmArgs) {
    using Cfg = TileCfg<BM, BN, WGM, WGN, NS>;
    constexpr int WM = Cfg::WM, WN = Cfg::WN, TM = Cfg::TM, TN = Cfg::TN;
    constexpr int LROWS = Cfg::LROWS, A_ROWS_PER_THREAD = Cfg::AR, B_ROWS_PER_THREAD = Cfg::BR;

    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    int vtile = blockIdx.x;            // virtual block id: blockIdx.x, then + gridDim.x per further tile (TileCfg::OVERLAP)
    bool first_tile = true, ln_parity = false;
    for (;;) {     // tiles of this workgroup (one, unless TileCfg::OVERLAP)
    GemmArgsPtr kp4 = (GemmArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp4));      // per tile: nothing read through it is carried around the loop
    const GemmArgs* kp = (const GemmArgs*)kp4;     // (the address space is inferred back: scalar loads)
    const f16* __restrict__ A = kp->A;
    const f16* __restrict__ W = kp->W;
    const float* __restrict__ bias = kp->bias;
    const f16* __restrict__ residual = kp->residual;
    void* __restrict__ out = kp->out;
    const ds_gemm_desc d = kp->d;
    const int tiles_m = kp->tiles_m, tiles_n = kp->tiles_n;
    const unsigned a_bytes = kp->a_bytes, w_bytes = kp->w_bytes;
    const float* __restrict__ ln_stats = kp->ln_stats;
    const float* __restrict__ ln_colsum = kp->ln_colsum;
    const float ln_eps = kp->ln_eps;
    const int group_m = kp->group_m;
    float2* __restrict__ colstats = kp->colstats;
    const int ld_stats = kp->ld_stats;
    const int m_base = kp->m_base, m_total = kp->m_total;
    // stage-major: stage s = [BM rows of A | BN rows of W] at smem + s * STAGE1
    auto stA = [&](int buf) { return reinterpret_cast<f16*>(smem) + (size_t)buf * (Cfg::SW ? BM : BM + BN) * BK; };
    auto stB = [&](int buf) { return reinterpret_cast<f16*>(smem) + (Cfg::SW ? (size_t)2 * BM : (size_t)buf * (BM + BN) + BM) * BK; };

    // ---- XCD-aware block remap (bijective for any grid size) ----
    const int nwg = tiles_m * tiles_n;
    // virtual block id v (= blockIdx.x, and for a persistent workgroup blockIdx.x + k * gridDim.x: the grid is a multiple of 8 or
    // the whole launch, so v % 8 still names the blocks that share an XCD) -> tile
    auto tile_of = [&](int v, int& tile_m, int& tile_n) {
        int bid = v;
        {
            const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
            bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        }
    // Within an XCD consecutive ids run concurrently (one workgroup per CU, 32 CUs): walking N fastest, a wide launch has
    // ~1 A panel against all tiles_n W panels in flight, and a W that exceeds the 4 MB L2 (GEGLU 5120x640: 6.5 MB, 10240x1280:
    // 26 MB) is streamed from beyond L2 once per A panel (3.3 GB read per launch of the level-2 GEGLU against 0.22 GB
    // algorithmic).  group_m > 1: ids walk group_m A panels x tiles_n W panels with M fastest, so the concurrent set is about
    // group_m x (32 / group_m) panels.  +3-4 % on the two wide GEGLU projections (tools/bench_wide_gemm.py), pure scheduling.
        if (group_m > 1) {
            const int per_group = group_m * tiles_n;
            const int first_m = (bid / per_group) * group_m, in_group = bid % per_group;
            const int gsz = min(tiles_m - first_m, group_m);
            tile_m = first_m + in_group % gsz;
            tile_n = in_group / gsz;
        } else {
            tile_n = bid % tiles_n;
            tile_m = bid / tiles_n;
        }
    };
    int tile_m, tile_n;
    tile_of(vtile, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;      // the tile being computed / stored (the staging cursor moves on to the next one before the epilogue)

    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));     // per tile as well: what derives from the thread id (LDS addresses, row / column offsets) is rebuilt
    const int tid = tid_;              // per tile instead of being hoisted out of the tile loop and held in ~25 registers around it
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    float* sW = reinterpret_cast<float*>(smem + Cfg::STRIP_OFF) + (size_t)wave * Cfg::SROWS * Cfg::STR;   // this wave's epilogue strip
    const int ld_row = tid >> 3;   // 0..LROWS-1
    const int ld_chunk = tid & 7;  // 16-byte chunk within the 64-half K-step
    // LDS-DMA writes lane l of a wave-instruction at LDS offset 16*l (8 rows x 128 B per instruction), so the physical
    // chunk is fixed (= ld_chunk) and the XOR swizzle moves to the SOURCE: the lane fetches logical chunk
    // ld_chunk ^ ((row>>1)&7).  LROWS is a multiple of 16, so the swizzle term is the same for every staged row.
    const unsigned src_chunk_bytes = Cfg::DMA ? (unsigned)(ld_chunk ^ ((ld_row >> 1) & 7)) * 16u : (unsigned)ld_chunk * 16u;

    // ---- per-row source bookkeeping for the A gather.  All global reads are raw BUFFER loads: a padding tap or a
    //      tail row gets byte offset OOB (> num_records), for which the hardware returns zeros -- no branch, no
    //      select, so the loads of the next K-step stay in flight behind the MFMAs of the current one. ----
    constexpr unsigned OOB = 0x80000000u;   // >= num_records (operands are < 2 GiB); + soffset (< 64 KiB) cannot wrap
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(A), 0, (int)a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(W), 0, (int)w_bytes, 0x00020000);
    RowInfo ri[A_ROWS_PER_THREAD];
    // A_CONV3_TI (3x3, stride 1, no upsample): the K loop walks the 64-channel chunks in the OUTER loop and the 9 taps in the
    // inner one.  Tap-major order re-reads a workgroup's rows once per tap with a whole channel sweep (164 KB) in between:
    // with 32 workgroups per XCD that is > 4 MB between reuses, so 8 of 9 tap passes came from beyond L2 (3.6-3.8 GB per
    // level-1 launch against 0.84 GB algorithmic, profiles/r2_pmc_hbm_traffic_v10.json).  Taps innermost, the reuse distance is
    // one K-step.  Per row: the byte offset of the CENTRE pixel and a 9-bit mask of the taps that fall inside the image;
    // the tap's displacement is wave-uniform (scalar), so a K-step costs an add, a bit test and a select per staged row.
    unsigned ctr[A_ROWS_PER_THREAD], vmask[A_ROWS_PER_THREAD];
    unsigned b_off[B_ROWS_PER_THREAD];
    // row bookkeeping of the tile whose operands are staged next: rows [m0_, m0_ + BM) of A, rows [n0_, n0_ + BN) of W
    auto setup_rows = [&](int m0_, int n0_) {
    if constexpr (AMODE == A_CONV3_TI) {
#pragma unroll
        for (int i = 0; i < A_ROWS_PER_THREAD; ++i) {
            const int m = m0_ + ld_row + LROWS * i;
            const bool valid = m < d.M;
            const int mm = (valid ? m : 0) + m_base;
            const int hw = d.hout * d.wout;
            const int img = mm / hw, rem = mm - img * hw;
            const int oy = rem / d.wout, ox = rem - oy * d.wout;
            ctr[i] = (unsigned)(img * d.hin * d.win + oy * d.win + ox) * (unsigned)d.lda * 2u + src_chunk_bytes;
            unsigned mk = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int iy = oy + t / 3 - 1, ix = ox + t % 3 - 1;
                mk |= (valid && iy >= 0 && iy < d.hin && ix >= 0 && ix < d.win) ? (1u << t) : 0u;
            }
            vmask[i] = mk;
        }
    }
#pragma unroll
    for (int i = 0; i < A_ROWS_PER_THREAD; ++i) {
        if constexpr (AMODE == A_CONV3_TI) break;
        const int m = m0_ + ld_row + LROWS * i;
        ri[i].valid = m < d.M;
        // conv / temporal modes address the WHOLE operand (A is not offset by a cut along M); dense rows are relative to A
        const int mm = (ri[i].valid ? m : 0) + ((AMODE == DS_A_CONV3 || AMODE == DS_A_TCONV) ? m_base : 0);
        if constexpr (AMODE == DS_A_CONV3) {
            const int hw = d.hout * d.wout;
            const int img = mm / hw, rem = mm - img * hw;
            const int oy = rem / d.wout;
            const int pad = d.asym_pad ? 0 : 1;
            ri[i].a = img; ri[i].b = oy * d.stride - pad; ri[i].c = (rem - oy * d.wout) * d.stride - pad;
            ri[i].base = (unsigned)(img * d.hin * d.win) * (unsigned)d.lda * 2u;
        } else if constexpr (AMODE == DS_A_TCONV) {
            ri[i].a = (mm / d.hw) % d.t_len; ri[i].b = 0; ri[i].c = 0;
            ri[i].base = (unsigned)mm * (unsigned)d.lda * 2u;
        } else {
            ri[i].a = ri[i].b = ri[i].c = 0;
            ri[i].base = (unsigned)mm * (unsigned)d.lda * 2u;
        }
    }
#pragma unroll
    for (int i = 0; i < B_ROWS_PER_THREAD; ++i) {
        const int n = n0_ + ld_row + LROWS * i;
        b_off[i] = n < d.N ? (unsigned)n * (unsigned)d.K * 2u + src_chunk_bytes : OOB;
    }
    };
    setup_rows(m0, n0);

    u32x4 ra[A_ROWS_PER_THREAD], rb[B_ROWS_PER_THREAD];
    const int hl = d.upsample ? 2 * d.hin : d.hin, wl = d.upsample ? 2 * d.win : d.win;
    const int ups = d.upsample ? 1 : 0;
    int tap = 0, cb = 0;   // position of the next K-step inside (tap, channel)
    unsigned kbytes = 0;   // byte offset of the next K-step inside a W row

    // Per-lane VGPR offsets are computed once per TAP (conv / temporal) or once per tile (dense); inside the K loop
    // the K position travels in the scalar `soffset` operand of the buffer load, so a K-step costs no vector ALU for
    // addressing -- the kernel is vector-issue bound (MI355X: every VALU op takes 4 issue cycles of the SIMD that also
    // has to issue the MFMAs; profiles/r1_notes.md), so instructions removed here are MFMA slots won.
    unsigned voff_a[A_ROWS_PER_THREAD];
    auto tap_offsets = [&]() {
        if constexpr (AMODE == A_CONV3_TI) {
            const int ky = tap / 3, kx = tap - ky * 3;
            const unsigned delta = (unsigned)(((ky - 1) * d.win + (kx - 1)) * d.lda * 2);   // wave-uniform, may be "negative"
#pragma unroll
            for (int i = 0; i < A_ROWS_PER_THREAD; ++i) voff_a[i] = ((vmask[i] >> tap) & 1u) ? ctr[i] + delta : OOB;
            return;
        }
#pragma unroll
        for (int i = 0; i < A_ROWS_PER_THREAD; ++i) {
            bool ok = ri[i].valid;
            unsigned off;
            if constexpr (AMODE == DS_A_CONV3) {
                const int ky = tap / 3, kx = tap - ky * 3;
                const int iy = ri[i].b + ky, ix = ri[i].c + kx;
                ok = ok && iy >= 0 && iy < hl && ix >= 0 && ix < wl;
                off = ri[i].base + (unsigned)((iy >> ups) * d.win + (ix >> ups)) * (unsigned)d.lda * 2u;
            } else if constexpr (AMODE == DS_A_TCONV) {
                const int tt = ri[i].a + tap - 1;
                ok = ok && tt >= 0 && tt < d.t_len;
                off = ri[i].base + (unsigned)((tap - 1) * d.hw * d.lda * 2);
            } else {
                off = ri[i].base;
            }
            voff_a[i] = ok ? off + src_chunk_bytes : OOB;
        }
    };
    tap_offsets();

    // cursor of the next K-step: (tap, channel chunk) -> soffset of the A load (cb), byte offset inside a W row (kbytes)
    auto next_k = [&]() {
        if constexpr (AMODE == A_CONV3_TI) {
            if (++tap == 9) { tap = 0; cb += BK; }
            kbytes = (unsigned)(tap * d.cin + cb) * 2u;
            tap_offsets();
        } else {
            kbytes += BK * 2;
            cb += BK;
            if (cb == d.cin) {
                cb = 0;
                ++tap;
                if constexpr (AMODE == DS_A_CONV3 || AMODE == DS_A_TCONV) tap_offsets();
            }
        }
    };
    // issue the global loads of the NEXT K-step (no waits, no branches, no vector address math); DMA: straight into
    // LDS buffer `buf` (an out-of-range lane zero-fills its 16 bytes)
    auto load_global = [&](int buf) {
        const unsigned soff_a = (unsigned)cb * 2u;
#pragma unroll
        for (int i = 0; i < A_ROWS_PER_THREAD; ++i) {
            if constexpr (Cfg::DMA)
                dma16(rsA, stA(buf) + (LROWS * i + 8 * wave) * BK, voff_a[i], soff_a);
            else
                ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rsA, voff_a[i], soff_a, 0);
        }
#pragma unroll
        for (int i = 0; i < B_ROWS_PER_THREAD; ++i) {
            if constexpr (Cfg::DMA)
                dma16(rsW, stB(buf) + (LROWS * i + 8 * wave) * BK, b_off[i], kbytes);
            else
                rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rsW, b_off[i], kbytes, 0);
        }
        next_k();
    };
    // split-ring form: the A rows of the cursor's K-step into A stage `buf` (cursor moves on); the W rows of K-step kw into the W stage
    // (dense / temporal / tap-major conv: K-step kw of a W row starts at byte kw * 2 BK)
    auto load_a_only = [&](int buf) {
        const unsigned soff_a = (unsigned)cb * 2u;
#pragma unroll
        for (int i = 0; i < A_ROWS_PER_THREAD; ++i) dma16(rsA, stA(buf) + (LROWS * i + 8 * wave) * BK, voff_a[i], soff_a);
        next_k();
    };
    auto load_w_only = [&](int kw) {
#pragma unroll
        for (int i = 0; i < B_ROWS_PER_THREAD; ++i) dma16(rsW, stB(0) + (LROWS * i + 8 * wave) * BK, b_off[i], (unsigned)kw * (BK * 2));
    };
    auto store_lds = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_ROWS_PER_THREAD; ++i) {
            const int row = ld_row + LROWS * i;
            *reinterpret_cast<u32x4*>(stA(buf) + row * BK + swz_chunk(row, ld_chunk) * 8) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_ROWS_PER_THREAD; ++i) {
            const int row = ld_row + LROWS * i;
            *reinterpret_cast<u32x4*>(stB(buf) + row * BK + swz_chunk(row, ld_chunk) * 8) = rb[i];
        }
    };

    const int nk = d.K / BK;
    const int fr = lane & 31, fh = lane >> 5;

    // staging cursor onto a tile: row bookkeeping, K position 0, first tap's offsets
    auto begin_staging = [&](int m0_, int n0_) {
        setup_rows(m0_, n0_);
        tap = 0; cb = 0; kbytes = 0;
        tap_offsets();
    };
    // first K-step(s) of the workgroup's FIRST tile.  A later tile's first K-step went out before the previous tile's epilogue, from
    // row bookkeeping that was dropped again (nothing of it stays in registers across the epilogue): it has been rebuilt above, the
    // cursor moves one K-step on.
    DS_STAMP(0);
    if (first_tile) {
        if constexpr (Cfg::SW) {
            load_global(0);                                  // A(0), W(0)
            if (d.K / BK > 1) load_a_only(1);                // A(1)
        } else if constexpr (Cfg::DMA) {
            for (int s0 = 0; s0 < NS - 1 && s0 < d.K / BK; ++s0) load_global(s0);   // NS-1 K-steps in flight
        } else {
            load_global(0);
        }
    } else {
        next_k();
    }
#if DS_MFMA16
    f32x4 acc[TN][TM][4];
    const int l15 = lane & 15, l4 = lane >> 4;
    // quad q of a 32x32 output tile: one output row (qrow) and 4 consecutive output columns (qcol ..+3) per lane
    auto qrow = [&](int q) { return (q >> 1) * 16 + l15; };
    auto qcol = [&](int q) { return (q & 1) * 16 + 4 * l4; };
    constexpr int NRQ = 2;                                  // distinct rows a lane owns in a tile
    auto qrq = [](int q) { return q >> 1; };
    auto rrow = [&](int rq) { return rq * 16 + l15; };
#define ACC(ni, mi, q, j) acc[ni][mi][q][j]
#else
    f32x16 acc[TN][TM];
    auto qrow = [&](int) { return fr; };
    auto qcol = [&](int q) { return 8 * q + 4 * fh; };
    constexpr int NRQ = 1;
    auto qrq = [](int) { return 0; };
    auto rrow = [&](int) { return fr; };
#define ACC(ni, mi, q, j) acc[ni][mi][4 * (q) + (j)]
#endif
    // A bias vector DECLARED shared (bias_rows > M, e.g. INT32_MAX: one vector for every row) starts in the accumulators:
    // the epilogue then has no bias work, and launches without residual / per-item bias transpose fp16 strips (below).
    // The sum is bias + p1 + p2 + ... instead of (p1 + p2 + ...) + bias -- the same value up to fp32 rounding, chosen per
    // LAYER (never per batch size), so a batch stays bit-identical to its separate forwards; a per-item table
    // (bias_rows <= M, also when it covers the launch with ONE item) is always added after the K sum.
    // Each lane reads the 4 columns of its register quads straight from global memory (two addresses per wave-instruction,
    // L2-resident) while the first K-steps' loads are in flight.
    constexpr bool ln_fold = AMODE == A_DENSE_LN || AMODE == A_DENSE_LNK;   // LayerNorm folded into this GEMM (ds_gemm_f16_ln / _lnk): see the transform after the K loop
    constexpr bool ln_kstats = AMODE == A_DENSE_LNK;   // row statistics from the A fragments of the K loop (no statistics launch)
    const bool res_f32 = residual && (d.epilogue & DS_EPI_RES_F32);   // fp32 residual rows (strict-precision residual stream)
    const bool bias_in_acc = !ln_fold && bias && d.bias_rows > m_total && (d.N % 8 == 0) && (d.ldc % 8 == 0) && (d.ldbias % 4 == 0) &&
                             (reinterpret_cast<uintptr_t>(bias) & 15) == 0 && !(d.epilogue & DS_EPI_OUT_F32) &&
                             (!residual || d.ldr % 8 == 0);
    if (bias_in_acc) {
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = n0 + wn * WN + ni * 32 + qcol(g);
                const f32x4 b = col < d.N ? *reinterpret_cast<const f32x4*>(bias + col) : f32x4{0, 0, 0, 0};
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int j = 0; j < 4; ++j) ACC(ni, mi, g, j) = b[j];
            }
    } else {
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int j = 0; j < 4; ++j) ACC(ni, mi, g, j) = 0.0f;
    }

    // ln_kstats: per lane the partial sum / sum of squares of ITS row's k-chunks (lane = (row fr, k half fh)); every A fragment of
    // the K loop passes through two v_dot2 chains (8 vector ops per fragment, in the shadow of the matrix pipe).  One-pass
    // variance in fp32 -- E[x^2] - mean^2 over the fp16 values the matrix cores multiply.
#if DS_MFMA16
    float ks1[2 * TM], ks2[2 * TM];     // per 16-row block of the wave tile
#pragma unroll
    for (int mi = 0; mi < 2 * TM; ++mi) { ks1[mi] = 0.0f; ks2[mi] = 0.0f; }
#else
    float ks1[TM], ks2[TM];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) { ks1[mi] = 0.0f; ks2[mi] = 0.0f; }
#endif
    auto kstat_op = [&](const f16x8& a, int mi, int h, int kind) {   // one op: pair h of the fragment into the sum / sum-of-squares chain
        if constexpr (ln_kstats) {
            const f16x2 one2 = {(f16)1.0f, (f16)1.0f};
            const f16x2 v = {a[2 * h], a[2 * h + 1]};
            if (kind == 0) ks1[mi] = __builtin_amdgcn_fdot2(v, one2, ks1[mi], false);
            else ks2[mi] = __builtin_amdgcn_fdot2(v, v, ks2[mi], false);
        }
    };
    auto kstats = [&](const f16x8& a, int mi) {
        if constexpr (ln_kstats) {
            const f16x2 one2 = {(f16)1.0f, (f16)1.0f};
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const f16x2 v = {a[2 * h], a[2 * h + 1]};
                ks1[mi] = __builtin_amdgcn_fdot2(v, one2, ks1[mi], false);
                ks2[mi] = __builtin_amdgcn_fdot2(v, v, ks2[mi], false);
            }
        }
    };

    // K-step synchronisation.  Register staging: store the staged operands, one barrier.  DMA: wait for this wave's
    // LDS-DMA of the next K-step, one barrier (then every wave's part has landed and the current buffer is free).
    // NS > 2 (deep variant for grids that leave CUs with a single workgroup): the LDS-DMA of K-step k+NS-1 is issued
    // during step k, so a lone workgroup still has NS-2 K-steps of loads in flight behind the one it waits for
    // (counted vmcnt: only the pieces of the step needed next must have landed).
    // lgkmcnt(0) IN FRONT OF THE BARRIER IS LOAD-BEARING.  A bare s_barrier orders nothing: gfx950 does not drain the memory
    // counters at a barrier, and the machine scheduler is free to hoist it above the MFMAs of the K-step (it does: in the
    // 4-stage tiles the barrier lands right behind the first MFMA).  Without the wait, fragment reads (ds_read) of this
    // K-step can still be in flight when a faster wave leaves the barrier and issues the LDS-DMA of the next K-step INTO
    // THE STAGE THOSE READS ARE FETCHING FROM (nbuf of step kt+1 is the stage read in step kt) -- or, after the last
    // K-step, writes its epilogue strip over it.  The window only opens when the reads are delayed beyond a DMA round trip,
    // i.e. with another kernel contending for the CU's LDS: this was round 1's "not repeatable under two concurrently
    // replaying hipGraphs" (profiles/r2_notes.md: found with tests/hazard_probe.py, ISA evidence, the static check
    // tests/test_host_cpu.py::test_gemm_isa_no_lds_reads_in_flight_at_barriers).
    auto stage_sync = [&](int nbuf, bool more) {
        if constexpr (Cfg::DMA) {
#ifdef DS_EXP_BARE_BARRIER   // round 1's form, kept ONLY so that the static ISA check can show that it detects it (never built into a library)
            if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((A_ROWS_PER_THREAD + B_ROWS_PER_THREAD) * (NS - 2)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
            if (more) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((A_ROWS_PER_THREAD + B_ROWS_PER_THREAD) * (NS - 2)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        } else {
            if (more) store_lds(nbuf);
            __syncthreads();
        }
    };

    // LayerNorm fold: column sums / column bias of this tile's BN columns -> LDS (read after the K loop), while the first K-step's
    // loads are in flight.  Two copies, alternating per tile of a persistent workgroup: a slower wave may still be reading the
    // previous tile's in its fold when this one is written; the copy before that is two barriers back.
    const int ln_off = (AMODE == A_DENSE_LN || AMODE == A_DENSE_LNK) && ln_parity ? 2 * BN : 0;
    if constexpr (AMODE == A_DENSE_LN || AMODE == A_DENSE_LNK) {
        float* sLNw = reinterpret_cast<float*>(smem + Cfg::LDS) + ln_off;
        for (int c = tid; c < BN; c += Cfg::NT) {
            const int col = min(n0 + c, d.N - 1);
            sLNw[c] = ln_colsum[col];
            sLNw[BN + c] = bias ? bias[col] : 0.0f;
        }
    }
    // the tile's first K-step has landed (every wave's share) and every wave has left the previous tile's epilogue
    if constexpr (Cfg::DMA) stage_sync(0, false);
    else stage_sync(0, true);
    DS_STAMP(1);

    // one LDS-DMA piece (8 rows x 128 B of this wave's share) of the next K-step, and the cursor advance after all pieces
    constexpr int NPIECE = A_ROWS_PER_THREAD + B_ROWS_PER_THREAD;
    auto dma_piece = [&](int j, int buf) {
        if (j < A_ROWS_PER_THREAD) dma16(rsA, stA(buf) + (LROWS * j + 8 * wave) * BK, voff_a[j], (unsigned)cb * 2u);
        else dma16(rsW, stB(buf) + (LROWS * (j - A_ROWS_PER_THREAD) + 8 * wave) * BK, b_off[j - A_ROWS_PER_THREAD], kbytes);
    };
    auto advance_k = [&]() { next_k(); };

    auto kstep = [&](int kt, auto more_tag) {
        constexpr bool MORE = decltype(more_tag)::value;   // K-step kt+NS-1 exists: stage it while computing this one
        const int buf = Cfg::SW ? (kt & 1) : kt % NS;
        const int nbuf = Cfg::SW ? buf : (kt + NS - 1) % NS;   // its LDS stage (read last in step kt-1; split-ring: A(kt+2) follows A(kt))
        if constexpr (!Cfg::DMA) {
            if (MORE) load_global(nbuf);
        } else if constexpr (Cfg::HOIST_ALL && !Cfg::SW) {
            if (MORE) load_global(nbuf);                   // small wave tiles: the pieces go out in front of the reads
        }
        const f16* a_base = stA(buf) + wm * WM * BK;
        const f16* b_base_l = stB(buf) + wn * WN * BK;
        auto read_a = [&](int kk, int mi) {
            const int row = mi * 32 + fr;
            return *reinterpret_cast<const f16x8*>(a_base + row * BK + swz_chunk(wm * WM + row, 2 * kk + fh) * 8);
        };
        auto read_b = [&](int kk, int ni) {
            const int row = ni * 32 + fr;
            return *reinterpret_cast<const f16x8*>(b_base_l + row * BK + swz_chunk(wn * WN + row, 2 * kk + fh) * 8);
        };
#if DS_MFMA16
        // 16x16x32: a K-step is two 32-deep k-slices; fragment of 16 rows: lane = (row l15, 16-byte k chunk l4)
        constexpr int TM16 = 2 * TM, TN16 = 2 * TN;
        auto read_a16 = [&](int ks, int t) {
            const int row = t * 16 + l15;
            return *reinterpret_cast<const f16x8*>(a_base + row * BK + swz_chunk(wm * WM + row, 4 * ks + l4) * 8);
        };
        auto read_b16 = [&](int ks, int t) {
            const int row = t * 16 + l15;
            return *reinterpret_cast<const f16x8*>(b_base_l + row * BK + swz_chunk(wn * WN + row, 4 * ks + l4) * 8);
        };
        if constexpr (Cfg::HOIST_ALL) {
            f16x8 af[2][TM16], bf[2][TN16];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int t = 0; t < TM16; ++t) af[ks][t] = read_a16(ks, t);
#pragma unroll
                for (int t = 0; t < TN16; ++t) bf[ks][t] = read_b16(ks, t);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int n16 = 0; n16 < TN16; ++n16)
#pragma unroll
                    for (int m16 = 0; m16 < TM16; ++m16) {
                        f32x4& c = acc[n16 >> 1][m16 >> 1][2 * (m16 & 1) + (n16 & 1)];
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[ks][n16], af[ks][m16], c, 0, 0, 0);
                        if (n16 == 0) kstats(af[ks][m16], m16);
                    }
        } else {
            // Big wave tiles (128x64: 8 x 4 MFMA tiles, 64x160: 4 x 10).  The fragments of the SHORT side (<= 4) are held for
            // a whole k-slice, two sets (the next slice's are read during this one); the LONG side is streamed through a ring
            // of three fragments, each read two fragments ahead of its use and multiplied with every short-side fragment.
            // The LDS-DMA pieces of the next K-step are issued between the MFMAs of the first 3/4 of the step.
            constexpr bool STREAM_M = TM16 >= TN16;
            constexpr int L = STREAM_M ? TM16 : TN16, S = STREAM_M ? TN16 : TM16;
            constexpr int NF = 2 * L, NSLOT = NF * S;                  // streamed fragments / MFMAs per K-step
#ifndef DS_M16_PSPAN4
#define DS_M16_PSPAN4 2          // the pieces go out behind the MFMAs of the first half of the K-step (A/B: profiles/r4_notes.md)
#endif
#ifndef DS_M16_AHEAD
#define DS_M16_AHEAD 2
#endif
            constexpr int PSPAN = NSLOT * DS_M16_PSPAN4 / 4;
            constexpr int AH = DS_M16_AHEAD, RD = AH + 1;              // streamed fragments read ahead of their use / ring depth
            f16x8 sf[2][S], ring[RD];
            auto read_s = [&](int ks, int t) { return STREAM_M ? read_b16(ks, t) : read_a16(ks, t); };
            auto read_l = [&](int f) { return STREAM_M ? read_a16(f / L, f % L) : read_b16(f / L, f % L); };
#pragma unroll
            for (int t = 0; t < S; ++t) sf[0][t] = read_s(0, t);
#pragma unroll
            for (int f = 0; f < AH; ++f) ring[f] = read_l(f);
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int ks = f / L, li = f % L;
                __builtin_amdgcn_sched_barrier(0);
                if (f + AH < NF) ring[(f + AH) % RD] = read_l(f + AH);
                if (ks == 0 && li < S) sf[1][li] = read_s(1, li);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int si = 0; si < S; ++si) {
                    const int m16 = STREAM_M ? li : si, n16 = STREAM_M ? si : li;
                    const f16x8& a_op = STREAM_M ? ring[f % RD] : sf[ks][si];   // activation rows
                    const f16x8& w_op = STREAM_M ? sf[ks][si] : ring[f % RD];   // weight rows
                    f32x4& c = acc[n16 >> 1][m16 >> 1][2 * (m16 & 1) + (n16 & 1)];
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(w_op, a_op, c, 0, 0, 0);
                    if (n16 == 0) kstats(a_op, m16);
                    const int slot = f * S + si;
                    // piece j goes out behind MFMA slot ((j + 1) * PSPAN) / NPIECE - 1
                    if (Cfg::DMA && !Cfg::SW && MORE) {
#pragma unroll
                        for (int j = 0; j < NPIECE; ++j)
                            if (slot == ((j + 1) * PSPAN) / NPIECE - 1) {
                                __builtin_amdgcn_sched_barrier(0);
                                dma_piece(j, nbuf);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (Cfg::DMA && !Cfg::SW) {
                if (MORE) advance_k();
            }
        }
#else
        if constexpr (Cfg::HOIST_ALL) {
            // all fragment reads of the K-step are issued up front (<= 16 ds_read_b128 in flight); the MFMAs of k-slice
            // kk then wait only for their own operands (counted lgkmcnt), so LDS latency hides behind the MFMAs of kk-1
            f16x8 af[4][TM], bf[4][TN];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) af[kk][mi] = read_a(kk, mi);
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) bf[kk][ni] = read_b(kk, ni);
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the reads ahead of the MFMA block (the scheduler would sink them)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                    for (int mi = 0; mi < TM; ++mi) {
                        acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[kk][ni], af[kk][mi], acc[ni][mi], 0, 0, 0);
                        if (ni == 0) kstats(af[kk][mi], mi);
                    }
        } else {
            // big wave tiles: the fragments of k-slice kk+1 are read while the MFMAs of kk run (two register sets), and
            // the LDS-DMA pieces of the next K-step are issued one at a time BETWEEN the MFMAs of the first two k-slices:
            // issuing a piece costs the wave ~60-100 cycles, which fit in the shadow of the matrix pipe instead of
            // delaying the first MFMA after the barrier (in-kernel stamps: 3.07k -> cycles per K-step, profiles/r1_notes.md)
            constexpr int NMF = TM * TN;
            constexpr int P0 = (NPIECE + 1) / 2, P1 = NPIECE - P0;
            f16x8 af[2][TM], bf[2][TN];
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) af[0][mi] = read_a(0, mi);
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) bf[0][ni] = read_b(0, ni);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                __builtin_amdgcn_sched_barrier(0);
                const int np = kk == 0 ? P0 : (kk == 1 ? P1 : 0);     // pieces issued inside this k-slice
                const int stride = np > 0 ? NMF / np : NMF + 1;
#pragma unroll
                for (int idx = 0; idx < NMF; ++idx) {
                    const int ni = idx / TM, mi = idx % TM;
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[kk & 1][ni], af[kk & 1][mi], acc[ni][mi], 0, 0, 0);
                    // the fragments of the next k-slice are read one per MFMA (not as a burst in front of the slice):
                    // right after the barrier only the first slice's reads of the eight waves queue up at the LDS
                    if (kk + 1 < 4 && idx < TM + TN) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (idx < TM) af[(kk + 1) & 1][idx] = read_a(kk + 1, idx);
                        else bf[(kk + 1) & 1][idx - TM] = read_b(kk + 1, idx - TM);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (Cfg::DMA && MORE && np > 0 && idx % stride == stride - 1 && idx / stride < np) {
                        __builtin_amdgcn_sched_barrier(0);
                        dma_piece((kk == 0 ? 0 : P0) + idx / stride, nbuf);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if constexpr (ln_kstats) {
                        // the k-slice's 8*TM statistic ops, a few per MFMA slot BEHIND the slot's reads / DMA piece, consecutive
                        // ops on different chains (a burst of 8 dependent v_dot2 in front of the fragment reads cost 20 %)
                        constexpr int PER = (8 * TM + NMF - 1) / NMF;
#pragma unroll
                        for (int q = 0; q < PER; ++q) {
                            const int o = idx * PER + q;
                            if (o < 8 * TM) kstat_op(af[kk & 1][o % TM], o % TM, (o / TM) >> 1, (o / TM) & 1);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (Cfg::DMA) {
                if (MORE) advance_k();
            }
        }
#endif
        if constexpr (Cfg::SW) {
            // split ring: every wave is done with A(kt) and W(kt) -> W(kt+1) into the W stage, A(kt+2) into A(kt)'s stage; the counted
            // wait leaves only the newest A pieces in flight (W(kt+1) and the older A(kt+1) have landed); second barrier: for every wave
            if (kt + 1 < nk) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                load_w_only(kt + 1);
                if (kt + 2 < nk) {
                    load_a_only(buf);
                    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(A_ROWS_PER_THREAD) : "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            } else {
                stage_sync(nbuf, false);
            }
        } else {
            stage_sync(nbuf, MORE);
        }
    };
#ifdef DS_SETPRIO_HI       // A/B (build variant "setprio"): static priority for the second-dispatched half of an 8-wave workgroup
    if constexpr (WGM * WGN == 8) {
        if (wave >= 4) __builtin_amdgcn_s_setprio(1);
    }
#endif
    {
        int kt = 0;
        if constexpr (!Cfg::SW) {
            for (; kt + NS - 1 < nk; ++kt) kstep(kt, std::true_type{});
        }
        for (; kt < nk; ++kt) kstep(kt, std::false_type{});
    }

    // ---- LayerNorm folded into the GEMM (ds_gemm_f16_ln).  The A operand is the RAW activation x, W holds gamma (.) W
    //      rounded to fp16, and with the row's (mean, rstd) the LayerNorm'ed product is recovered exactly:
    //        sum_k ((x_k - mean) rstd gamma_k + beta_k) W_nk = rstd (acc_n - mean cs_n) + cb_n,
    //      cs_n = sum_k fp16(gamma_k W_nk) (the sum of the operand row actually multiplied), cb_n = sum_k beta_k W_nk (+ bias).
    //      Applied in the accumulator layout (a lane owns one row: two scalars per 32-row block; cs / cb are 16-byte loads
    //      per register quad like the accumulator-init bias), after which the epilogue sees a plain bias-free product.
    //      The normalised activation is never rounded to fp16 and never written to memory. ----
    if constexpr (ln_fold) {
        float2 st[TM][NRQ];
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int rq = 0; rq < NRQ; ++rq) {
                if constexpr (ln_kstats) {
#if DS_MFMA16
                    float s1 = ks1[2 * mi + rq], s2 = ks2[2 * mi + rq];      // the four k quarters of the row
                    s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
                    s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
#else
                    const float s1 = ks1[mi] + __shfl_xor(ks1[mi], 32), s2 = ks2[mi] + __shfl_xor(ks2[mi], 32);   // the two k halves of the row
#endif
                    const float mean = s1 / (float)d.K;
                    st[mi][rq] = make_float2(mean, rsqrtf(fmaxf(s2 / (float)d.K - mean * mean, 0.0f) + ln_eps));
                } else {
                    st[mi][rq] = reinterpret_cast<const float2*>(ln_stats)[min(m0 + wm * WM + mi * 32 + rrow(rq), d.M - 1)];   // tail rows: never stored
                }
            }
        // cs / cb of this tile's BN columns were staged in LDS behind the operand stages at kernel start (sLN, visible after the
        // K loop's first barrier).  One 32-column tile at a time: with all 4*TN quads' vectors loaded in front of the
        // arithmetic (what the compiler does by itself) the 256x320 tile spilled ~300 B per lane.
        const float* sLN = reinterpret_cast<const float*>(smem + Cfg::LDS) + ln_off;
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cl = wn * WN + ni * 32 + qcol(g);
                const f32x4 cs = *reinterpret_cast<const f32x4*>(sLN + cl);
                const f32x4 cb = *reinterpret_cast<const f32x4*>(sLN + BN + cl);
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        ACC(ni, mi, g, j) = fmaf(st[mi][qrq(g)].y, ACC(ni, mi, g, j) - st[mi][qrq(g)].x * cs[j], cb[j]);
            }
            // the tile's values are pinned here: left alone, the arithmetic sinks down to its uses in the epilogue and the
            // 8*TN column vectors stay live across it
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) {
#if DS_MFMA16
#pragma unroll
                for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(acc[ni][mi][g]));
#else
                asm volatile("" : "+v"(acc[ni][mi]));
#endif
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const bool bias_done = bias_in_acc || ln_fold;   // nothing left to add in the epilogue

    DS_STAMP(2);
    // ---- persistent workgroup: its next tile's first K-step goes out NOW, into stage 0 -- every wave is past the K loop's last
    //      barrier, so no fragment read of this tile is outstanding, and the epilogue below works in strips behind stage 0
    //      (TileCfg::STRIP_OFF).  The loads (A rows from HBM, the W panel from L2) land while the epilogue reads its residual rows
    //      and stores; the barrier at the top of the next tile waits for them and for every wave's epilogue. ----
    int vnext = vtile, tile_m_n = 0, tile_n_n = 0;
    bool more_tiles = false;
    if constexpr (Cfg::OVERLAP) {
        vnext = vtile + (int)gridDim.x;
        more_tiles = vnext < nwg;
        if (more_tiles) {
            tile_of(vnext, tile_m_n, tile_n_n);
            begin_staging(tile_m_n * BM, tile_n_n * BN);
            load_global(0);
        }
    }
    const bool geglu = d.epilogue & DS_EPI_GEGLU;
    const bool silu = d.epilogue & DS_EPI_SILU;
    const bool out_f32 = d.epilogue & DS_EPI_OUT_F32;
    // fp32 output on the fast path: plain accumulator dump (+ shared bias), e.g. attention scores that go through memory
    const bool fast32 = out_f32 && (!residual || res_f32) && !geglu && !silu && (d.N % 8 == 0) && (d.ldc % 4 == 0) &&
                        (reinterpret_cast<uintptr_t>(out) & 15) == 0;
    const bool fast = (!out_f32 || fast32) && (d.N % 8 == 0) && (d.ldc % 8 == 0 || fast32) &&
                      (!residual || (res_f32 ? (d.ldr % 4 == 0 && (reinterpret_cast<uintptr_t>(residual) & 15) == 0) : d.ldr % 8 == 0)) &&
                      (!bias || (d.ldbias % 4 == 0 && (reinterpret_cast<uintptr_t>(bias) & 15) == 0));
    const bool shared_bias = bias && d.bias_rows >= m_total;   // (bias_rows == M: a per-item table covering the launch with one item)

    // ---- epilogue.  D[i][j] of an MFMA tile: j = lane&31 is the output row m, i = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    //      the column n.  Each wave moves its own tiles through its private LDS strip (32 rows x up to NG tiles, fp32),
    //      reads them back as 8-column chunks of one row, applies bias / per-item bias / residual / SiLU in fp32,
    //      rounds once to fp16 and stores 16 bytes per lane (row segments of 64..256 bytes).  LDS operations of one wave
    //      execute in order, so the strip needs no barrier -- the eight waves run their epilogues independently. ----
    auto wave_sync = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    };
    // ---- per-column partial statistics of the stored tile (ds_gemm_f16_stats): the GroupNorm that reads this output takes its
    //      (sum, sum of squares) from here instead of from a pass of its own over the tensor.  A lane of the strip sweeps owns 8
    //      columns of every rps-th row of a 32-row block: it accumulates them over the sweeps (cs / cq), the lanes of a column
    //      chunk are then summed through the wave's strip (fixed order: r0 = 0, 1, ...), and the first cpr lanes store
    //      colstats[row block][column] = (sum, sumsq) of the block's valid rows -- no atomics, one writer per entry. ----
    auto stats_flush = [&](const float (&cs)[8], const float (&cq)[8], int cpr, int rps, int ch, int r0, bool lane_on, bool col_on,
                           int mrow0, long ocol) {
        float* const sS = sW;                               // the strip is free: every sweep's reads have been consumed
        const int rs = 16 * cpr;                            // floats per scratch row: cpr chunks x (8 sums + 8 squares)
        wave_sync();
        if (lane_on) {
            *reinterpret_cast<f32x4*>(sS + r0 * rs + ch * 16) = f32x4{cs[0], cs[1], cs[2], cs[3]};
            *reinterpret_cast<f32x4*>(sS + r0 * rs + ch * 16 + 4) = f32x4{cs[4], cs[5], cs[6], cs[7]};
            *reinterpret_cast<f32x4*>(sS + r0 * rs + ch * 16 + 8) = f32x4{cq[0], cq[1], cq[2], cq[3]};
            *reinterpret_cast<f32x4*>(sS + r0 * rs + ch * 16 + 12) = f32x4{cq[4], cq[5], cq[6], cq[7]};
        }
        wave_sync();
        if (lane < cpr) {
            f32x4 t[4] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
            for (int r = 0; r < rps; ++r) {
#pragma unroll
                for (int q = 0; q < 4; ++q) t[q] += *reinterpret_cast<const f32x4*>(sS + r * rs + lane * 16 + 4 * q);
            }
            if (col_on && mrow0 < d.M) {
                float2* dst = colstats + (long)(mrow0 >> 5) * ld_stats + ocol;
                *reinterpret_cast<f32x4*>(dst) = f32x4{t[0][0], t[2][0], t[0][1], t[2][1]};
                *reinterpret_cast<f32x4*>(dst + 2) = f32x4{t[0][2], t[2][2], t[0][3], t[2][3]};
                *reinterpret_cast<f32x4*>(dst + 4) = f32x4{t[1][0], t[3][0], t[1][1], t[3][1]};
                *reinterpret_cast<f32x4*>(dst + 6) = f32x4{t[1][2], t[3][2], t[1][3], t[3][3]};
            }
        }
    };
    auto epilogue = [&](auto ge_tag, auto res_tag, auto pib_tag) {
        constexpr bool GE = decltype(ge_tag)::value;
        constexpr int RMODE = decltype(res_tag)::value;  // residual add: 0 none, 1 fp16 rows, 2 fp32 rows (DS_EPI_RES_F32)
        constexpr bool RES = RMODE != 0, RES32 = RMODE == 2;
        constexpr bool PIB = !GE && decltype(pib_tag)::value;   // per-item bias (time-embedding add), never with GEGLU
        constexpr int TNE = GE ? TN / 2 : TN;          // output tiles per wave row (GEGLU halves the columns)
        constexpr int NG = Cfg::NG, STR = Cfg::STR;
        if constexpr (GE) {
            // GEGLU in registers: weight rows are interleaved in 32-row groups [x | gate], so tile 2p holds x and tile
            // 2p+1 the gate of the same 32 outputs; acc[p] <- (x + b) * gelu(gate + b)
#pragma unroll
            for (int p2 = 0; p2 < TNE; ++p2) {
                const int nx = n0 + wn * WN + 2 * p2 * 32;   // + qcol(g) : x column of quad g
                f32x4 bxq[4], bgq[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const bool okn = nx + qcol(g) + 32 < d.N && bias && !bias_done;
                    bxq[g] = okn ? *reinterpret_cast<const f32x4*>(bias + nx + qcol(g)) : f32x4{0, 0, 0, 0};
                    bgq[g] = okn ? *reinterpret_cast<const f32x4*>(bias + nx + qcol(g) + 32) : f32x4{0, 0, 0, 0};
                }
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int j = 0; j < 4; j += 2) {
                            const f32x2 xv = {ACC(2 * p2, mi, g, j) + bxq[g][j], ACC(2 * p2, mi, g, j + 1) + bxq[g][j + 1]};
                            const f32x2 gv = {ACC(2 * p2 + 1, mi, g, j) + bgq[g][j], ACC(2 * p2 + 1, mi, g, j + 1) + bgq[g][j + 1]};
                            const f32x2 r = xv * fast_gelu_erf2(gv);
                            ACC(p2, mi, g, j) = r[0];
                            ACC(p2, mi, g, j + 1) = r[1];
                        }
            }
        }
        DS_STAMP(3);
        // fp16 strips for launches without bias / residual / per-item bias and fp16 output (QKV and q projections, GEGLU:
        // its bias is added in the stage above): SiLU is applied in the ACCUMULATOR layout and the value rounded to fp16
        // there -- the same fp32 operations and the same single rounding as the fp32-strip path below, so the bits are
        // identical -- and the strip carries halfs: half the LDS bytes (the epilogue of a 256x320 tile moves 655 KB
        // through the LDS in fp32 and is LDS-bandwidth bound), and a sweep is one 16-byte read and one 16-byte store
        // with nothing in between, so all sweeps of a group are in flight together.  A shared bias is already in the
        // accumulators (bias_in_acc above).
        if constexpr (!RES && !PIB) {
            if (fast && !out_f32 && (!bias || GE || bias_done)) {
                constexpr int NGH_ = TNE < 2 * NG ? TNE : 2 * NG;      // tiles per group in the same strip bytes,
                constexpr int NGH = Cfg::HALF ? NGH_ : (NGH_ < 4 ? NGH_ : 4);   // at most 4: 16 chunks per row, 8 sweeps of 4 rows (HALF: 16-row strips, 20 chunks, 6 sweeps of 3)
                constexpr int STRH = 32 * NGH + 8;                     // halfs per strip row (16-byte aligned chunks)
                constexpr int SR = Cfg::SROWS, NH = 32 / SR;           // strip rows; strips per 32-row block
                static_assert(STRH <= 2 * STR, "fp16 strip must fit the fp32 strip");
                f16* const sH = reinterpret_cast<f16*>(sW);
#pragma unroll
                for (int c0 = 0; c0 < TNE; c0 += NGH) {
                    const int gw = (TNE - c0) < NGH ? (TNE - c0) : NGH;
                    const int cpr = gw * 4, rps = 64 / cpr;
                    const int ch = lane % cpr, r0 = lane / cpr;
                    const bool lane_on = lane < rps * cpr;
                    const int ncol = GE ? n0 + wn * WN + 2 * (c0 * 32 + ch * 8 - (ch * 8) % 32) + (ch * 8) % 32
                                        : n0 + wn * WN + c0 * 32 + ch * 8;
                    const long ocol = GE ? (long)tile_n * (BN / 2) + wn * (WN / 2) + c0 * 32 + ch * 8 : (long)ncol;
                    const bool col_on = lane_on && (GE ? ncol + 32 < d.N : ncol < d.N);
                    const int nsw = (SR + rps - 1) / rps;
                    static_assert(NGH <= 4 || SR == 16, "8 sweeps hold a whole strip");
#pragma unroll
                    for (int mi0 = 0; mi0 < TM * NH; ++mi0) {
                        const int mi = mi0 / NH, hb = mi0 % NH;         // 32-row block, 16-row half of it (HALF strips)
                        const int mrow0 = m0 + wm * WM + mi * 32 + hb * SR;
#pragma unroll
                        for (int t = 0; t < NGH; ++t) {
                            if (t < gw) {
#pragma unroll
                                for (int g = 0; g < 4; ++g) {
                                    if (NH == 2 && qrq(g) != hb) continue;
                                    f32x4 v = {ACC(c0 + t, mi, g, 0), ACC(c0 + t, mi, g, 1), ACC(c0 + t, mi, g, 2), ACC(c0 + t, mi, g, 3)};
                                    if constexpr (!GE) {
                                        if (silu) {
#pragma unroll
                                            for (int j = 0; j < 4; ++j) v[j] = fast_silu(v[j]);
                                        }
                                    }
                                    const f16x4 h = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                                    *reinterpret_cast<f16x4*>(sH + (qrow(g) - hb * SR) * STRH + t * 32 + qcol(g)) = h;
                                }
                            }
                        }
                        wave_sync();
                        f16* const out_base = reinterpret_cast<f16*>(out) + (long)(mrow0 + r0) * d.ldc + ocol;
                        const long out_step = (long)rps * d.ldc;
                        u32x4 hv[8];
#pragma unroll
                        for (int sw = 0; sw < 8; ++sw)
                            if (sw < nsw) hv[sw] = *reinterpret_cast<const u32x4*>(sH + min(sw * rps + r0, SR - 1) * STRH + ch * 8);
#pragma unroll
                        for (int sw = 0; sw < 8; ++sw) {
                            if (sw < nsw) {
                                const int row = sw * rps + r0;
#ifdef DS_EXP_NOSTORE
                                asm volatile("" ::"v"(hv[sw]));
#else
                                if (col_on && row < SR && mrow0 + row < d.M) DS_OUT_STORE_H(reinterpret_cast<u32x4*>(out_base + sw * out_step), hv[sw]);
#endif
                            }
                        }
                        if constexpr (!GE && DS_GEMM_STATS) {
                            if (colstats) {        // statistics of the stored (rounded) values
                                float cs[8] = {0, 0, 0, 0, 0, 0, 0, 0}, cq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                                for (int sw = 0; sw < 8; ++sw) {
                                    if (sw < nsw) {
                                        const int row = sw * rps + r0;
                                        const bool on = col_on && row < SR && mrow0 + row < d.M;
                                        const f16x8 hh = __builtin_bit_cast(f16x8, hv[sw]);
#pragma unroll
                                        for (int j = 0; j < 8; ++j) {
                                            const float t = on ? (float)hh[j] : 0.0f;
                                            cs[j] += t;
                                            cq[j] = fmaf(t, t, cq[j]);
                                        }
                                    }
                                }
                                stats_flush(cs, cq, cpr, rps, ch, r0, lane_on, col_on, mrow0, ocol);
                            }
                        }
                        wave_sync();
                    }
                }
                return;
            }
        }
#pragma unroll
        for (int c0 = 0; c0 < TNE; c0 += NG) {
            const int gw = (TNE - c0) < NG ? (TNE - c0) : NG;   // tiles in this column group (compile-time after unroll)
            const int cpr = gw * 4;                             // 8-column chunks per strip row
            const int rps = 64 / cpr;                           // rows per sweep of the wave
            const int ch = lane % cpr, r0 = lane / cpr;
            const bool lane_on = lane < rps * cpr;
            // column of this lane's chunk: in the N space (bias, bounds) and in the output
            // W4 (fp32 residual rows, DS_EPI_RES_F32): a lane's 8 values are TWO groups of 4 columns, cpr * 4 columns apart (group A at
            // 4 ch, group B at 4 (ch + cpr)), so that each of its 16-byte fp32 accesses -- residual loads, fp32 stores -- is contiguous
            // across the wave; 8 consecutive fp32 columns per lane made every access instruction touch twice the lines it used.
#ifdef DS_EXP_NO_W4        // A/B (variant "now4")
            constexpr bool W4 = false;
#else
            constexpr bool W4 = RES32 && !DS_GEMM_STATS;
#endif
            const int dB = W4 ? cpr * 4 : 4;                    // group B = group A + dB columns (strip, bias, residual, output alike)
            const int ncol = GE ? n0 + wn * WN + 2 * (c0 * 32 + ch * 8 - (ch * 8) % 32) + (ch * 8) % 32
                                : n0 + wn * WN + c0 * 32 + ch * (W4 ? 4 : 8);
            const long ocol = GE ? (long)tile_n * (BN / 2) + wn * (WN / 2) + c0 * 32 + ch * 8 : (long)ncol;
            const bool col_on = lane_on && (GE ? ncol + 32 < d.N : ncol < d.N);
            const bool col_onB = W4 ? (lane_on && ncol + dB < d.N) : col_on;
            float bx[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) bx[j] = 0.0f;
            if (!GE && shared_bias && fast && !bias_done) {
                if (col_on) {
                    const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + ncol);
                    bx[0] = b0[0]; bx[1] = b0[1]; bx[2] = b0[2]; bx[3] = b0[3];
                }
                if (col_onB) {
                    const f32x4 b1 = *reinterpret_cast<const f32x4*>(bias + ncol + dB);
                    bx[4] = b1[0]; bx[5] = b1[1]; bx[6] = b1[2]; bx[7] = b1[3];
                }
            }
            constexpr int SR = Cfg::SROWS, NH = 32 / SR;           // strip rows; strips per 32-row block (HALF: two 16-row strips)
#pragma unroll
            for (int mi0 = 0; mi0 < TM * NH; ++mi0) {
                const int mi = mi0 / NH, hb = mi0 % NH;
                const int mrow0 = m0 + wm * WM + mi * 32 + hb * SR;
                // accumulators -> strip
#pragma unroll
                for (int t = 0; t < NG; ++t) {
                    if (t < gw) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            if (NH == 2 && qrq(g) != hb) continue;
                            f32x4 v = {ACC(c0 + t, mi, g, 0), ACC(c0 + t, mi, g, 1), ACC(c0 + t, mi, g, 2), ACC(c0 + t, mi, g, 3)};
                            *reinterpret_cast<f32x4*>(sW + (qrow(g) - hb * SR) * STR + t * 32 + qcol(g)) = v;
                        }
                    }
                }
                wave_sync();
                float cs[8] = {0, 0, 0, 0, 0, 0, 0, 0}, cq[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // column statistics of this 32-row block (colstats)
                if (fast) {
                    // Batches of SB sweeps: first the global loads (residual, per-item bias), then ALL the strip reads of
                    // the batch, then the arithmetic and the stores -- straight-line code (RES / PIB are compile-time),
                    // so the LDS and memory latencies of a batch overlap instead of adding up per sweep.
                    constexpr int SB = TM * TN > 8 ? 2 : 4;   // 160 accumulator registers leave room for two sweeps
                    const int nsw = (SR + rps - 1) / rps;
                    const long out_step = (long)rps * d.ldc, res_step = (long)rps * d.ldr;
                    f16* const out_base = reinterpret_cast<f16*>(out) + (long)(mrow0 + r0) * d.ldc + ocol;
                    const f16* const res_base = (RES && !RES32) ? residual + (long)(mrow0 + r0) * d.ldr + ocol : nullptr;
                    const float* const res32_base = RES32 ? reinterpret_cast<const float*>(residual) + (long)(mrow0 + r0) * d.ldr + ocol : nullptr;
#pragma unroll
                    for (int sb = 0; sb < 8; sb += SB) {
                        if (sb < nsw) {
                            f16x8 res[SB];
                            f32x4 rs0[SB], rs1[SB];
                            f32x4 pb0[SB], pb1[SB], p0[SB], p1[SB];
                            bool ok[SB], okb[SB];
#pragma unroll
                            for (int u = 0; u < SB; ++u) {
                                const int row = (sb + u) * rps + r0;
                                ok[u] = col_on && row < SR && mrow0 + row < d.M;
                                okb[u] = col_onB && row < SR && mrow0 + row < d.M;
                                if constexpr (RES32) {
                                    const float* rp = res32_base + (sb + u) * res_step;
                                    rs0[u] = DS_RES_LOAD(reinterpret_cast<const f32x4*>(ok[u] ? rp : reinterpret_cast<const float*>(residual)));
                                    rs1[u] = DS_RES_LOAD(reinterpret_cast<const f32x4*>(okb[u] ? rp + dB : reinterpret_cast<const float*>(residual)));
                                } else if constexpr (RES) {
                                    res[u] = DS_RES_LOAD(reinterpret_cast<const f16x8*>(ok[u] ? res_base + (sb + u) * res_step : residual));
                                }
                                if constexpr (PIB) {
                                    const int mm = ok[u] ? mrow0 + row + m_base : 0;
                                    const float* bp = bias + (long)(mm / d.bias_rows) * d.ldbias + (ok[u] ? ncol : 0);
                                    pb0[u] = *reinterpret_cast<const f32x4*>(bp);
                                    pb1[u] = *reinterpret_cast<const f32x4*>(bp + 4);
                                }
                            }
#pragma unroll
                            for (int u = 0; u < SB; ++u) {
                                const int row = min((sb + u) * rps + r0, SR - 1);
                                p0[u] = *reinterpret_cast<const f32x4*>(sW + row * STR + ch * (W4 ? 4 : 8));
                                p1[u] = *reinterpret_cast<const f32x4*>(sW + row * STR + ch * (W4 ? 4 : 8) + dB);
                            }
#pragma unroll
                            for (int u = 0; u < SB; ++u) {
                                float v[8] = {p0[u][0] + bx[0], p0[u][1] + bx[1], p0[u][2] + bx[2], p0[u][3] + bx[3],
                                              p1[u][0] + bx[4], p1[u][1] + bx[5], p1[u][2] + bx[6], p1[u][3] + bx[7]};
                                if constexpr (PIB) {
                                    v[0] += pb0[u][0]; v[1] += pb0[u][1]; v[2] += pb0[u][2]; v[3] += pb0[u][3];
                                    v[4] += pb1[u][0]; v[5] += pb1[u][1]; v[6] += pb1[u][2]; v[7] += pb1[u][3];
                                }
                                if constexpr (RES32) {
                                    v[0] += rs0[u][0]; v[1] += rs0[u][1]; v[2] += rs0[u][2]; v[3] += rs0[u][3];
                                    v[4] += rs1[u][0]; v[5] += rs1[u][1]; v[6] += rs1[u][2]; v[7] += rs1[u][3];
                                } else if constexpr (RES) {
#pragma unroll
                                    for (int j = 0; j < 8; ++j) v[j] += (float)res[u][j];
                                }
                                if (silu) {
#pragma unroll
                                    for (int j = 0; j < 8; ++j) v[j] = fast_silu(v[j]);
                                }
                                if constexpr (!GE && DS_GEMM_STATS) {
                                    if (colstats) {
#pragma unroll
                                        for (int j = 0; j < 8; ++j) {
                                            const float t = ok[u] ? v[j] : 0.0f;
                                            cs[j] += t;
                                            cq[j] = fmaf(t, t, cq[j]);
                                        }
                                    }
                                }
                                f16x8 o;
#pragma unroll
                                for (int j = 0; j < 8; ++j) o[j] = (f16)v[j];
#ifdef DS_EXP_NOSTORE   // diagnostic builds only (tools/build_stamps.sh)
                                asm volatile("" ::"v"(o));
#else
                                if (ok[u] || okb[u]) {
                                    if (out_f32) {
                                        float* o32 = reinterpret_cast<float*>(out) + (long)(mrow0 + r0) * d.ldc + ocol + (sb + u) * out_step;
#if DS_EXP_NT & 8       // A/B (variant "nt32"): the fp32 residual stream's stores non-temporal as well
                                        if (ok[u]) DS_STORE_NT(reinterpret_cast<f32x4*>(o32), (f32x4{v[0], v[1], v[2], v[3]}));
                                        if (okb[u]) DS_STORE_NT(reinterpret_cast<f32x4*>(o32 + dB), (f32x4{v[4], v[5], v[6], v[7]}));
#else
                                        if (ok[u]) *reinterpret_cast<f32x4*>(o32) = f32x4{v[0], v[1], v[2], v[3]};
                                        if (okb[u]) *reinterpret_cast<f32x4*>(o32 + dB) = f32x4{v[4], v[5], v[6], v[7]};
#endif
                                    } else if constexpr (W4) {
                                        f16* const o16 = out_base + (sb + u) * out_step;
                                        if (ok[u]) DS_OUT_STORE_F(reinterpret_cast<f16x4*>(o16), (f16x4{o[0], o[1], o[2], o[3]}));
                                        if (okb[u]) DS_OUT_STORE_F(reinterpret_cast<f16x4*>(o16 + dB), (f16x4{o[4], o[5], o[6], o[7]}));
                                    } else {
                                        DS_OUT_STORE_F(reinterpret_cast<f16x8*>(out_base + (sb + u) * out_step), o);
                                    }
                                }
#endif
                            }
                        }
                    }
                    if constexpr (!GE && DS_GEMM_STATS) {
                        if (colstats) stats_flush(cs, cq, cpr, rps, ch, r0, lane_on, col_on, mrow0, ocol);
                    }
                } else {
                    // generic (rare, tiny layers): scalar stores, any N, fp32 or fp16 out; GEGLU not supported here
                    for (int idx = lane; idx < SR * gw * 32; idx += 64) {
                        const int row = idx / (gw * 32), col = idx - row * (gw * 32);
                        const int m = mrow0 + row, n = n0 + wn * WN + c0 * 32 + col;
                        if (m >= d.M || n >= d.N) continue;
                        float v = sW[row * STR + col];
                        if (bias && !bias_done) v += bias[(long)((m + m_base) / d.bias_rows) * d.ldbias + n];
                        if (residual) v += res_f32 ? reinterpret_cast<const float*>(residual)[(long)m * d.ldr + n] : (float)residual[(long)m * d.ldr + n];
                        if (silu) v = fast_silu(v);
                        if (out_f32) reinterpret_cast<float*>(out)[(long)m * d.ldc + n] = v;
                        else reinterpret_cast<f16*>(out)[(long)m * d.ldc + n] = (f16)v;
                    }
                }
                wave_sync();
            }
        }
    };
    const bool pib = bias && !shared_bias;
    auto with_bias_mode = [&](auto ge_tag, auto res_tag) {
        if (pib) epilogue(ge_tag, res_tag, std::true_type{});
        else epilogue(ge_tag, res_tag, std::false_type{});
    };
    using R0 = std::integral_constant<int, 0>;
    using R16 = std::integral_constant<int, 1>;
    using R32 = std::integral_constant<int, 2>;
    if constexpr (ln_fold) {         // ds_gemm_f16_ln: no residual, no per-item bias (checked on the host)
        if (geglu) {
            if constexpr (TN % 2 == 0 && BN % 320 != 0) epilogue(std::true_type{}, R0{}, std::false_type{});
        } else {
            epilogue(std::false_type{}, R0{}, std::false_type{});
        }
    } else if (geglu) {
        if constexpr (TN % 2 == 0 && BN % 320 != 0) {
            if (residual) epilogue(std::true_type{}, R16{}, std::false_type{});     // fp16 residual only (checked on the host)
            else epilogue(std::true_type{}, R0{}, std::false_type{});
        }
    } else if (res_f32) {
        epilogue(std::false_type{}, R32{}, std::false_type{});                      // shared bias only (checked on the host)
    } else if (residual) {
        with_bias_mode(std::false_type{}, R16{});
    } else {
        with_bias_mode(std::false_type{}, R0{});
    }
    DS_STAMP(4);
    if (!more_tiles) break;
    vtile = vnext;
    first_tile = false;
    ln_parity = !ln_parity;
    }   // tiles
}

struct StatOut { float2* p = nullptr; int ld = 0; int m_base = 0, m_total = 0; };   // ds_gemm_f16_stats: where the per-column partial statistics go; a cut along M

// CUs the persistent tiles are spread over (a multiple of 8, so that v % 8 keeps naming the XCD)
static int device_cus() {
    static const int ncu = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
        return n / 8 * 8;
    }();
    return ncu;
}

// Launch share (ds_set_launch_share): n similar launch sequences run concurrently on n streams (an 8-GPU rank's cond / uncond
// evaluations): the persistent tiles of ONE launch are planned on CUs / n, so that the n launches in flight fill the chip with whole
// rounds of big tiles instead of each leaving a partly filled last round to the other's.
static int g_launch_share = 1;
static int launch_cus() { return device_cus() / g_launch_share / 8 * 8; }

template <int BM, int BN, int WGM, int WGN, int AMODE, int NS = 2>
int launch(const void* A, const void* W, const float* bias, const void* residual, void* out,
           const ds_gemm_desc& d, hipStream_t st, const float* ln_stats, const float* ln_colsum, float ln_eps, StatOut so = StatOut()) {
    using Cfg = TileCfg<BM, BN, WGM, WGN, NS>;
    constexpr size_t lds = Cfg::LDS + ((AMODE == A_DENSE_LN || AMODE == A_DENSE_LNK) ? 4 * BN * sizeof(float) : 0);   // + staged column sums / bias, two copies
    static_assert(lds <= 163840, "LDS budget");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16_kernel<BM, BN, WGM, WGN, AMODE, NS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            ds_set_error("ds_gemm_f16: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return DS_ELAUNCH;
        }
        attr_set = true;
    }
    const int tiles_m = ds_cdiv(d.M, BM), tiles_n = ds_cdiv(d.N, BN);
    // grouped walk for wide one-workgroup-per-CU launches (see the kernel); DS_GEMM_GROUP_M: 0 = never (A/B runs)
    const int group_env = (int)DS_TUNE_INT("DS_GEMM_GROUP_M", 6);
    const int group_m = (Cfg::WG_PER_CU == 1 && tiles_n >= 16 && group_env > 1) ? group_env : 1;   // 10 N tiles (2560 x 320): no gain, -2 % at M = 327680
    // buffer-load addressing is 32-bit and offset 2^31 marks 'out of range': the A operand and W must each stay below 2 GiB
    const long a_rows = (AMODE == DS_A_CONV3 || AMODE == A_CONV3_TI) ? (long)d.nimg * d.hin * d.win
                        : (AMODE == DS_A_TCONV && so.m_total) ? (long)so.m_total : (long)d.M;
    const long a_bytes = ((a_rows - 1) * d.lda + d.cin) * 2;
    const long w_bytes = (long)d.N * d.K * 2;
    if (a_bytes >= 0x7FFF0000L || w_bytes >= 0x7FFF0000L) {
        ds_set_error("ds_gemm_f16: operand of %ld / %ld bytes exceeds the 2 GiB buffer-addressing range; lower the tile batch", a_bytes, w_bytes);
        return DS_EINVAL;
    }
    // TileCfg::OVERLAP: one persistent workgroup per CU (a multiple of 8, so that v % 8 keeps naming the XCD)
    const int ncu = launch_cus();
    const int nblk = tiles_m * tiles_n;
    const int grid = (Cfg::OVERLAP && nblk > ncu) ? ncu : nblk;
    GemmArgs ka;
    ka.A = (const f16*)A; ka.W = (const f16*)W; ka.bias = bias; ka.residual = (const f16*)residual; ka.out = out;
    ka.d = d;
    ka.tiles_m = tiles_m; ka.tiles_n = tiles_n; ka.a_bytes = (unsigned)a_bytes; ka.w_bytes = (unsigned)w_bytes;
    ka.ln_stats = ln_stats; ka.ln_colsum = ln_colsum; ka.ln_eps = ln_eps; ka.group_m = group_m;
    ka.colstats = so.p; ka.ld_stats = so.ld;
    ka.m_base = so.m_base; ka.m_total = so.m_total ? so.m_total : d.M;
    gemm_f16_kernel<BM, BN, WGM, WGN, AMODE, NS><<<grid, Cfg::NT, lds, st>>>(ka);
    DS_CHECK_LAUNCH("ds_gemm_f16");
    return DS_OK;
}

// Tile choice.  256-row tiles halve the operand bytes a CU pulls through its vector-memory path and LDS per MFMA
// (profiles/r1_notes.md: on 128x128 tiles each of the three -- loads, LDS, MFMA -- is near its limit), but need one
// workgroup per CU to have work: they are used when the grid still fills the chip.
enum { TILE_128x64 = 0, TILE_128x128 = 1, TILE_256x256 = 2, TILE_256x320 = 3, TILE_128x128_DEEP = 4, TILE_128x64_DEEP = 5,
       TILE_128x320_SW = 6, TILE_128x256_SW = 7 };   // split-ring forms, two workgroups per CU (TileCfg::SW)

int choose_tile(const ds_gemm_desc& d) {
    const int forced = (int)DS_TUNE_INT("DS_GEMM_TILE", -1);
    const bool geglu = d.epilogue & DS_EPI_GEGLU;
    const int waste128 = ds_cdiv(d.N, 128) * 128 - d.N;
    const int small = (geglu || waste128 * 8 <= d.N) ? TILE_128x128 : TILE_128x64;
    if (forced == TILE_128x64 || forced == TILE_128x128) return geglu ? TILE_128x128 : forced;
    const long tiles_m256 = ds_cdiv(d.M, 256);
    int big = -1;
    if (d.N % 256 == 0) big = TILE_256x256;
    else if (d.N % 320 == 0 && !geglu) big = TILE_256x320;
    // N a multiple of both (1280, 3840): the 320-wide tile where it takes FEWER rounds of one tile per CU -- 40960 x 1280: 640 tiles (3
    // rounds) against 800 (4); 10240 x 3840: 480 (2) against 600 (3); measured -3 ... -9 % on the level-3 convolutions, -13 ... -29 % on
    // the level-3/4 QKV projections of small batches, and slower where the round counts tie (profiles/r6_notes.md section 3).  Every tile
    // sums K in the same order: same bits.
    if (big == TILE_256x256 && d.N % 320 == 0 && !geglu && DS_TUNE_INT("DS_GEMM_PREF320", 1) != 0) {
        const long ncu = launch_cus();
        if (ds_cdiv(tiles_m256 * (d.N / 320), ncu) < ds_cdiv(tiles_m256 * (d.N / 256), ncu)) big = TILE_256x320;
    }
    // a grid that leaves every CU with at most one 128x128 workgroup: 4-stage LDS-DMA pipeline instead of relying on a
    // partner workgroup to hide the load latency
    const long nblk128 = (long)ds_cdiv(d.M, 128) * ds_cdiv(d.N, 128);
    int small_or_deep = small;
    if (d.K / BK >= 8 && forced != TILE_128x128 && forced != TILE_128x64) {
        if (!geglu && nblk128 <= 128) small_or_deep = TILE_128x64_DEEP;          // half the CUs or fewer: narrower tiles, twice the workgroups
        else if (small == TILE_128x128 && nblk128 <= 256) small_or_deep = TILE_128x128_DEEP;
    }
    if (forced == TILE_128x128_DEEP || (forced == TILE_128x64_DEEP && !geglu)) return forced;
    if (forced == TILE_128x320_SW && d.N % 320 == 0 && !geglu) return forced;
    if (forced == TILE_128x256_SW && d.N % 256 == 0) return forced;
    if (big < 0) return small_or_deep;
    if (forced == TILE_256x320 && d.N % 320 == 0 && !geglu) return TILE_256x320;      // (N = 1280 k: both big tiles apply)
    if (forced == TILE_256x256 && d.N % 256 == 0) return TILE_256x256;
    if (forced == TILE_256x256 || forced == TILE_256x320) return big;
    const long nblk = tiles_m256 * (d.N / (big == TILE_256x256 ? 256 : 320));
    const long big_min = DS_TUNE_INT("DS_GEMM_BIG_MIN", 160);
    return nblk >= big_min ? big : small_or_deep;
}

template <int AMODE>
int dispatch(int tile, const void* A, const void* W, const float* bias, const void* residual, void* out,
             const ds_gemm_desc& d, hipStream_t st, const float* ln_stats = nullptr, const float* ln_colsum = nullptr, float ln_eps = 0.0f,
             StatOut so = StatOut()) {
    // the split-ring forms stage W by K-step index: not for the taps-innermost K order; deep-K 3x3 convolutions keep the big tiles
    // They are measured-and-not-adopted forms (profiles/r6_notes.md section 2: no gain on any short-K shape at 16 evaluations, the short-K
    // family is bound by the memory system's rate for its access pattern, not by a per-CU serial chain): instantiated in the "tune"
    // build only (choose_tile returns them only when forced there).
#ifdef DS_TUNING_ENV
    if constexpr (AMODE == A_CONV3_TI || AMODE == DS_A_CONV3) {
        if (tile == TILE_128x320_SW) tile = TILE_256x320;
        if (tile == TILE_128x256_SW) tile = TILE_256x256;
    } else {
        if (tile == TILE_128x320_SW) return launch<128, 320, 4, 1, AMODE, 1>(A, W, bias, residual, out, d, st, ln_stats, ln_colsum, ln_eps, so);
        if (tile == TILE_128x256_SW) return launch<128, 256, 2, 2, AMODE, 1>(A, W, bias, residual, out, d, st, ln_stats, ln_colsum, ln_eps, so);
    }
#endif
    switch (tile) {
        case TILE_256x256: return launch<256, 256, 2, 4, AMODE>(A, W, bias, residual, out, d, st, ln_stats, ln_colsum, ln_eps, so);
        case TILE_256x320: return launch<256, 320, 4, 2, AMODE>(A, W, bias, residual, out, d, st, ln_stats, ln_colsum, ln_eps, so);
        case TILE_128x128: return launch<128, 128, 2, 2, AMODE>(A, W, bias, residual, out, d, st, ln_stats, ln_colsum, ln_eps, so);
        case TILE_128x128_DEEP: return launch<128, 128, 2, 2, AMODE, 4>(A, W, bias, residual, out, d, st, ln_stats, ln_colsum, ln_eps, so);
        case TILE_128x64_DEEP: return launch<128, 64, 2, 2, AMODE, 4>(A, W, bias, residual, out, d, st, ln_stats, ln_colsum, ln_eps, so);
        default:           return launch<128, 64, 2, 2, AMODE>(A, W, bias, residual, out, d, st, ln_stats, ln_colsum, ln_eps, so);
    }
}

}  // namespace

#ifdef DS_GEMM_STAMPS
extern "C" int ds_dbg_set_stamps(void* p) {
    unsigned long long* q = (unsigned long long*)p;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(ds_dbg_stamps), &q, sizeof(q));
}
#endif

// ln_colsum != nullptr: LayerNorm folded in; ln_stats == nullptr then selects the in-kernel statistics (ln_eps)
static int gemm_entry(const void* A, const void* W, const float* bias, const void* residual, void* out,
                      const ds_gemm_desc* desc, void* stream, const float* ln_stats, const float* ln_colsum, float ln_eps = 0.0f,
                      StatOut so = StatOut()) {
    DS_CHECK_ARG(A && W && out && desc, "ds_gemm_f16: null argument");
    const ds_gemm_desc& d = *desc;
    DS_CHECK_ARG(d.M > 0 && d.N > 0 && d.K > 0, "ds_gemm_f16: M,N,K must be positive (got %d,%d,%d)", d.M, d.N, d.K);
    DS_CHECK_ARG(d.K % BK == 0, "ds_gemm_f16: K=%d must be a multiple of %d", d.K, BK);
    DS_CHECK_ARG(d.cin > 0 && d.cin % BK == 0 && d.K % d.cin == 0, "ds_gemm_f16: cin=%d must be a multiple of %d dividing K=%d", d.cin, BK, d.K);
    DS_CHECK_ARG(d.lda % 8 == 0 && d.lda >= d.cin, "ds_gemm_f16: lda=%d must be a multiple of 8 and >= cin", d.lda);
    DS_CHECK_ARG(d.bias_rows > 0, "ds_gemm_f16: bias_rows must be positive");
    DS_CHECK_ARG(d.ldc > 0, "ds_gemm_f16: ldc must be positive");
    DS_CHECK_ARG(!bias || d.ldbias >= d.N, "ds_gemm_f16: ldbias=%d must be >= N=%d", d.ldbias, d.N);
    if (d.a_mode == DS_A_DENSE) {
        DS_CHECK_ARG(d.cin == d.K, "ds_gemm_f16: dense mode needs cin == K");
    } else if (d.a_mode == DS_A_CONV3) {
        DS_CHECK_ARG(d.K == 9 * d.cin, "ds_gemm_f16: conv3 mode needs K == 9*cin");
        DS_CHECK_ARG(d.stride == 1 || d.stride == 2, "ds_gemm_f16: conv3 stride must be 1 or 2");
        DS_CHECK_ARG(d.nimg > 0 && d.hin > 0 && d.win > 0 && d.hout > 0 && d.wout > 0, "ds_gemm_f16: conv3 dims");
        DS_CHECK_ARG((long)d.nimg * d.hout * d.wout == d.M, "ds_gemm_f16: conv3 M != nimg*hout*wout");
        DS_CHECK_ARG(!(d.upsample && d.stride != 1), "ds_gemm_f16: upsample needs stride 1");
    } else if (d.a_mode == DS_A_TCONV) {
        DS_CHECK_ARG(d.K == 3 * d.cin, "ds_gemm_f16: tconv mode needs K == 3*cin");
        DS_CHECK_ARG(d.t_len > 0 && d.hw > 0 && d.M % (d.t_len * d.hw) == 0, "ds_gemm_f16: tconv M must be nseq*t_len*hw");
    } else {
        DS_CHECK_ARG(false, "ds_gemm_f16: unknown a_mode %d", d.a_mode);
    }
    if (d.epilogue & DS_EPI_GEGLU) {
        DS_CHECK_ARG(d.N % 64 == 0, "ds_gemm_f16: GEGLU needs N %% 64 == 0");
        DS_CHECK_ARG(!(d.epilogue & DS_EPI_OUT_F32) && d.ldc % 8 == 0, "ds_gemm_f16: GEGLU needs fp16 out, ldc %% 8 == 0");
    }
    if (d.epilogue & DS_EPI_RES_F32) {
        DS_CHECK_ARG(residual, "ds_gemm_f16: DS_EPI_RES_F32 without a residual");
        DS_CHECK_ARG(!(d.epilogue & DS_EPI_GEGLU), "ds_gemm_f16: DS_EPI_RES_F32 is not available with GEGLU");
        DS_CHECK_ARG(!bias || d.bias_rows >= d.M, "ds_gemm_f16: DS_EPI_RES_F32 takes a shared bias vector only");
    }
    hipStream_t st = (hipStream_t)stream;
    const int tile = choose_tile(d);
    // A persistent big-tile launch whose LAST round of tiles fills less than half of the CUs -- the [cond | uncond] pair of one window
    // that an 8-GPU rank evaluates per level: 81920 rows = 320 row tiles on 256 CUs, a second round a quarter full, two rounds of time
    // for 1.25 rounds of work -- is cut along M: the rows of the full rounds go to the big tiles (exactly one tile per CU and round),
    // the remaining rows to a second launch with the tile chosen for THEIR count (small tiles, two workgroups per CU).  Every tile
    // variant sums K in the same order, so the result does not depend on the cut (batch invariance, result_sha256 unchanged).
    if (DS_PERSIST != 0 && (tile == TILE_256x256 || tile == TILE_256x320) && !so.p && so.m_total == 0 &&
        (d.a_mode != DS_A_DENSE || ((long)d.M - 1) * d.lda * 2 + (long)d.cin * 2 < 0x7FFF0000L) && DS_TUNE_INT("DS_GEMM_TAIL_SPLIT", 1) != 0) {
        const long tiles_n = d.N / (tile == TILE_256x256 ? 256 : 320), tiles_m = ds_cdiv(d.M, 256);
        const long nblk = tiles_m * tiles_n, ncu = launch_cus();
        const long rounds = nblk / ncu, rem = nblk % ncu;
        if (rounds >= 1 && rem > 0 && 2 * rem < ncu && (rounds * ncu) % tiles_n == 0) {
            const long r_main = rounds * ncu / tiles_n * 256;          // < M: rem > 0
            const long out_elt = (d.epilogue & DS_EPI_OUT_F32) ? 4 : 2, res_elt = (d.epilogue & DS_EPI_RES_F32) ? 4 : 2;
            const int ti_mode_ = (int)DS_TUNE_INT("DS_CONV_TAPS_INNER", -1);       // the same rule as the uncut launch below (it fixes the K order)
            const bool taps_inner = d.a_mode == DS_A_CONV3 && (ti_mode_ < 0 ? (long)d.hin * d.win >= 2048 : ti_mode_ > 0) && d.stride == 1 &&
                                    !d.upsample && !d.asym_pad;
            for (int part = 0; part < 2; ++part) {
                const long r0 = part ? r_main : 0;
                ds_gemm_desc c = d;
                c.M = (int)(part ? d.M - r_main : r_main);
                const int tl = part ? choose_tile(c) : tile;
                // dense rows are relative to A; the conv / temporal modes address the whole operand through m_base
                const char* a_p = (const char*)A + (d.a_mode == DS_A_DENSE ? r0 * d.lda * 2 : 0);
                const char* r_p = residual ? (const char*)residual + r0 * d.ldr * res_elt : nullptr;
                char* o_p = (char*)out + r0 * d.ldc * out_elt;
                StatOut cut;
                cut.m_base = (int)r0; cut.m_total = d.M;
                int rc;
                if (d.a_mode == DS_A_CONV3)
                    rc = taps_inner ? dispatch<A_CONV3_TI>(tl, a_p, W, bias, r_p, o_p, c, st, nullptr, nullptr, 0.0f, cut)
                                    : dispatch<DS_A_CONV3>(tl, a_p, W, bias, r_p, o_p, c, st, nullptr, nullptr, 0.0f, cut);
                else if (d.a_mode == DS_A_TCONV)
                    rc = dispatch<DS_A_TCONV>(tl, a_p, W, bias, r_p, o_p, c, st, nullptr, nullptr, 0.0f, cut);
                else
                    rc = ln_stats ? dispatch<A_DENSE_LN>(tl, a_p, W, bias, r_p, o_p, c, st, ln_stats + 2 * r0, ln_colsum, 0.0f, cut)
                         : ln_colsum ? dispatch<A_DENSE_LNK>(tl, a_p, W, bias, r_p, o_p, c, st, nullptr, ln_colsum, ln_eps, cut)
                                  : dispatch<DS_A_DENSE>(tl, a_p, W, bias, r_p, o_p, c, st, nullptr, nullptr, 0.0f, cut);
                if (rc) return rc;
            }
            return DS_OK;
        }
    }
    // 32-bit buffer addressing with offset 2^31 as the 'out of range' marker: an A operand of 2 GiB or more (dense
    // only: e.g. the 2048-wide FF hidden of init_attn at 655k rows) is processed in row chunks.
    if (d.a_mode == DS_A_DENSE && ((long)d.M - 1) * d.lda * 2 + (long)d.cin * 2 >= 0x7FFF0000L) {
        DS_CHECK_ARG(!bias || d.bias_rows >= d.M, "ds_gemm_f16: a >= 2 GiB dense operand with a per-item bias is not supported");
        const long rows_max = ((0x7FFF0000L / ((long)d.lda * 2)) / 256) * 256;
        DS_CHECK_ARG(rows_max >= 256, "ds_gemm_f16: lda=%d too large", d.lda);
        const long out_elt = (d.epilogue & DS_EPI_OUT_F32) ? 4 : 2;
        for (long r0 = 0; r0 < d.M; r0 += rows_max) {
            ds_gemm_desc c = d;
            c.M = (int)((d.M - r0) < rows_max ? (d.M - r0) : rows_max);
            c.bias_rows = d.bias_rows;   // shared bias: any value >= c.M
            const char* a_p = (const char*)A + r0 * d.lda * 2;
            const char* r_p = residual ? (const char*)residual + r0 * d.ldr * ((d.epilogue & DS_EPI_RES_F32) ? 4 : 2) : nullptr;
            char* o_p = (char*)out + r0 * d.ldc * out_elt;
            StatOut sc = so;
            if (sc.p) sc.p += (r0 >> 5) * (long)so.ld;       // rows_max is a multiple of 256: whole 32-row blocks
            int rc = ln_stats ? dispatch<A_DENSE_LN>(tile, a_p, W, bias, r_p, o_p, c, st, ln_stats + 2 * r0, ln_colsum)
                     : ln_colsum ? dispatch<A_DENSE_LNK>(tile, a_p, W, bias, r_p, o_p, c, st, nullptr, ln_colsum, ln_eps)
                              : dispatch<DS_A_DENSE>(tile, a_p, W, bias, r_p, o_p, c, st, nullptr, nullptr, 0.0f, sc);
            if (rc) return rc;
        }
        return DS_OK;
    }
    if (d.a_mode == DS_A_CONV3) {
        // taps innermost where the tap-major order thrashes the L2 (large images: a workgroup's rows x all channels no longer
        // fit next to its 31 neighbours'): level-1 tiles 40x64 fetch 0.57 GB instead of 3.8 GB per launch and run 2.5 % faster;
        // on the 20x32 / 10x16 levels L2 already caught the reuse and the per-K-step select costs 1-3 % (gpurun_out/conv)
        const int ti_mode = (int)DS_TUNE_INT("DS_CONV_TAPS_INNER", -1);   // A/B ("tune" build variant): 0 never, 1 always
        const bool taps_inner = ti_mode < 0 ? (long)d.hin * d.win >= 2048 : ti_mode > 0;
        if (taps_inner && d.stride == 1 && !d.upsample && !d.asym_pad) return dispatch<A_CONV3_TI>(tile, A, W, bias, residual, out, d, st, nullptr, nullptr, 0.0f, so);
        return dispatch<DS_A_CONV3>(tile, A, W, bias, residual, out, d, st, nullptr, nullptr, 0.0f, so);
    }
    if (d.a_mode == DS_A_TCONV) return dispatch<DS_A_TCONV>(tile, A, W, bias, residual, out, d, st, nullptr, nullptr, 0.0f, so);
    if (ln_stats) return dispatch<A_DENSE_LN>(tile, A, W, bias, residual, out, d, st, ln_stats, ln_colsum);
    if (ln_colsum) return dispatch<A_DENSE_LNK>(tile, A, W, bias, residual, out, d, st, nullptr, ln_colsum, ln_eps);
    return dispatch<DS_A_DENSE>(tile, A, W, bias, residual, out, d, st, nullptr, nullptr, 0.0f, so);
}

// ds_gemm_f16 that also writes, for every 32-row block of the output, the per-column (sum, sum of squares) of the values it
// stores: colstats[(m / 32) * ld_stats + n], float2.  The GroupNorm that reads this output (ds_groupnorm_rows_colstats) then needs
// no statistics pass over the tensor.  Conditions = the epilogue's vector path: N % 8 == 0, aligned operands, no GEGLU.
// 1 if this library's GEMM epilogues can write the column statistics (built with DS_GEMM_STATS)
extern "C" int ds_gemm_has_stats(void) { return DS_GEMM_STATS != 0; }

extern "C" int ds_gemm_f16_stats(const void* A, const void* W, const float* bias, const void* residual, void* out, float* colstats,
                                 int ld_stats, const ds_gemm_desc* desc, void* stream) {
    DS_CHECK_ARG(DS_GEMM_STATS != 0, "ds_gemm_f16_stats: this library was built without DS_GEMM_STATS (build variant \"gemmstats\" has it)");
    DS_CHECK_ARG(desc && colstats, "ds_gemm_f16_stats: null argument");
    const ds_gemm_desc& d = *desc;
    const bool out_f32 = d.epilogue & DS_EPI_OUT_F32;
    DS_CHECK_ARG(!(d.epilogue & DS_EPI_GEGLU), "ds_gemm_f16_stats: not available with GEGLU");
    DS_CHECK_ARG(d.N % 8 == 0 && ld_stats >= d.N && ld_stats % 2 == 0 && (reinterpret_cast<uintptr_t>(colstats) & 15) == 0,
                 "ds_gemm_f16_stats: N %% 8 == 0, ld_stats >= N and even, colstats 16-byte aligned");
    // the conditions of the epilogue's vector path (`fast` in the kernel): the statistics are formed there
    DS_CHECK_ARG(d.ldc % (out_f32 ? 4 : 8) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0, "ds_gemm_f16_stats: ldc / out alignment");
    DS_CHECK_ARG(!out_f32 || !residual || (d.epilogue & DS_EPI_RES_F32), "ds_gemm_f16_stats: fp32 output with an fp16 residual takes the scalar epilogue");
    DS_CHECK_ARG(!residual || ((d.epilogue & DS_EPI_RES_F32) ? (d.ldr % 4 == 0 && (reinterpret_cast<uintptr_t>(residual) & 15) == 0) : d.ldr % 8 == 0),
                 "ds_gemm_f16_stats: residual alignment");
    DS_CHECK_ARG(!bias || (d.ldbias % 4 == 0 && (reinterpret_cast<uintptr_t>(bias) & 15) == 0), "ds_gemm_f16_stats: bias alignment");
    DS_CHECK_ARG(!(out_f32 && (d.epilogue & DS_EPI_SILU)), "ds_gemm_f16_stats: fp32 output with SiLU takes the scalar epilogue");
    StatOut so;
    so.p = reinterpret_cast<float2*>(colstats);
    so.ld = ld_stats;
    return gemm_entry(A, W, bias, residual, out, desc, stream, nullptr, nullptr, 0.0f, so);
}

extern "C" int ds_set_launch_share(int n) {
    DS_CHECK_ARG(n >= 1 && n <= 8, "ds_set_launch_share: n must be 1..8");
    g_launch_share = n;
    return DS_OK;
}

extern "C" int ds_gemm_f16(const void* A, const void* W, const float* bias, const void* residual, void* out,
                           const ds_gemm_desc* desc, void* stream) {
    return gemm_entry(A, W, bias, residual, out, desc, stream, nullptr, nullptr);
}

extern "C" int ds_gemm_f16_ln(const void* x, const void* W_gamma, const float* ln_stats, const float* ln_colsum,
                              const float* ln_colbias, void* out, const ds_gemm_desc* desc, void* stream) {
    DS_CHECK_ARG(x && W_gamma && ln_stats && ln_colsum && out && desc, "ds_gemm_f16_ln: null argument");
    DS_CHECK_ARG(desc->a_mode == DS_A_DENSE, "ds_gemm_f16_ln: dense A operand only (the LayerNorm row is the K dimension)");
    DS_CHECK_ARG(!(desc->epilogue & DS_EPI_OUT_F32) && desc->N % 8 == 0 && desc->ldc % 8 == 0, "ds_gemm_f16_ln: fp16 output, N and ldc multiples of 8");
    DS_CHECK_ARG((reinterpret_cast<uintptr_t>(ln_stats) & 7) == 0 && (reinterpret_cast<uintptr_t>(ln_colsum) & 15) == 0 &&
                 (!ln_colbias || (reinterpret_cast<uintptr_t>(ln_colbias) & 15) == 0), "ds_gemm_f16_ln: stats 8-byte, column vectors 16-byte aligned");
    return gemm_entry(x, W_gamma, ln_colbias, nullptr, out, desc, stream, ln_stats, ln_colsum);
}

// The same with the rows' statistics computed inside the kernel (every tile walks its rows' whole K = C extent, so each
// A fragment also feeds a sum / sum-of-squares chain): no statistics launch, x is read once in all.
extern "C" int ds_gemm_f16_lnk(const void* x, const void* W_gamma, float ln_eps, const float* ln_colsum,
                               const float* ln_colbias, void* out, const ds_gemm_desc* desc, void* stream) {
    DS_CHECK_ARG(x && W_gamma && ln_colsum && out && desc, "ds_gemm_f16_lnk: null argument");
    DS_CHECK_ARG(desc->a_mode == DS_A_DENSE && desc->cin == desc->K, "ds_gemm_f16_lnk: dense A operand only (the LayerNorm row is the K dimension)");
    DS_CHECK_ARG(!(desc->epilogue & DS_EPI_OUT_F32) && desc->N % 8 == 0 && desc->ldc % 8 == 0, "ds_gemm_f16_lnk: fp16 output, N and ldc multiples of 8");
    DS_CHECK_ARG((reinterpret_cast<uintptr_t>(ln_colsum) & 15) == 0 && (!ln_colbias || (reinterpret_cast<uintptr_t>(ln_colbias) & 15) == 0),
                 "ds_gemm_f16_lnk: colsum / colbias 16-byte aligned");
    DS_CHECK_ARG(ln_eps > 0.0f, "ds_gemm_f16_lnk: eps must be positive");
    return gemm_entry(x, W_gamma, ln_colbias, nullptr, out, desc, stream, nullptr, ln_colsum, ln_eps);
}
