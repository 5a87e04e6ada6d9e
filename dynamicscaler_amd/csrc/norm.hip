// GroupNorm (32 groups) and LayerNorm for channel-contiguous fp16 activations on gfx950.  HBM-bound:
// 16-byte vector loads, fp32 statistics, wavefront / LDS reductions, no atomics in global memory
// (results are bitwise reproducible).
//
// GroupNorm is split into
//   stats   : ds_groupnorm_stats  = partial (sum, sumsq) per (instance, row chunk, group) + finalize in fp64
//   apply   : ds_groupnorm_apply  = y = x*a[c] + b[c] (+ SiLU), a = rstd*gamma, b = beta - mean*rstd*gamma
// so the apply can later be folded into the consumer GEMM's A-operand gather.
// References: GroupNormSpecific lvdm/basics.py:76-86; nn.GroupNorm(32, C, eps=1e-6) attention.py:238,297;
// nn.GroupNorm(32, C) in TemporalConvBlock openaimodel3d.py:275-292; nn.LayerNorm attention.py:199-201.
#include <stdio.h>
#include <stdlib.h>
#include "common.h"

namespace {

typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
constexpr int MAX_C = 4096;
// Rows per partial-sum chunk: the unit of the statistics' summation order, so a function of the instance's SHAPE only (never of
// the launch's instance count: a batch must equal its separate forwards bit for bit).  ~160 KB of fp16 per chunk -- 256 rows at
// C <= 320, 128 at C <= 640, 64 above -- so that the wide, short instances of levels 3-4 (2560 / 640 rows x 1280) give 40 / 10
// workgroups per instance instead of 10 / 3 (an 8-GPU rank launches 2 instances); never more than ~160 chunks per instance
// (every apply workgroup reduces its instance's partial sums itself).
constexpr int GN_CHUNK_ROWS_MAX = 256;
constexpr int GN_CHUNK_ROWS_MIN = 64;
static inline int gn_chunk_rows(int rows_per_inst, int C) {
    // diagnostic: 0 = 256 rows everywhere (rounds 1-2), 2 = 256 / 256 / 128.  It CHANGES the summation order, i.e. the bits: said once on
    // stderr when set, so that a stray value on one rank does not go unnoticed (that rank's replica would differ in the last bits)
    const int rule = (int)DS_TUNE_INT("DS_GN_CHUNK_RULE", 1);       // ("tune" build variant only)
    if (rule == 0) return GN_CHUNK_ROWS_MAX;
    const int by_c = rule == 2 ? (C <= 640 ? 256 : 128) : (C <= 320 ? 256 : (C <= 640 ? 128 : 64));
    int by_rows = GN_CHUNK_ROWS_MIN;
    while (by_rows < GN_CHUNK_ROWS_MAX && (long)by_rows * 160 < rows_per_inst) by_rows *= 2;
    return by_c > by_rows ? by_c : by_rows;
}
constexpr int GN_U = 4;   // rows in flight per thread in the streaming loops (2 and 8 measured 1-8 % slower, tools/bench_norms.py)
constexpr int GN_U_SPARSE = 16;      // ... where a launch has at most GN_SPARSE_WGS workgroups (measured: profiles/r3_notes.md section 8)
constexpr int GN_SPARSE_WGS = 1024;
constexpr int GN_LDS_FLOATS = 4096;   // per array: rl*C (<= 2048 + C) when C <= 2048, C otherwise

__device__ __forceinline__ float fast_silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// 8 consecutive channels of one row as they sit in memory: fp16 (16 bytes) or fp32 (32 bytes: the strict-precision mode
// keeps the UNet's residual stream in fp32, unet.py `residual_dtype`; the norms read it and write the fp16 GEMM operand).
struct f32x8 { f32x4 lo, hi; };
template <typename XT> struct Row8;
// GroupNorm kernels address a thread's 8 channels through chan() / row_load(): for fp16 rows they are 8 consecutive channels (one
// 16-byte access, consecutive threads 16 bytes apart); for fp32 rows two groups of 4, nvec * 4 channels apart (thread `col` owns
// channels 4 col .. 4 col + 3 and 4 (col + nvec) ..): each of its two 16-byte loads is then contiguous across the wave, where 8
// consecutive fp32 channels per thread made every load instruction touch twice the cache lines it used.
template <> struct Row8<f16> {
    typedef f16x8 raw;
    static __device__ __forceinline__ int chan(int col, int j, int) { return col * 8 + j; }
    static __device__ __forceinline__ raw row_load(const f16* row, int col, int) { return *reinterpret_cast<const f16x8*>(row + col * 8); }
    static __device__ __forceinline__ raw row_load_stream(const f16* row, int col, int) { return DS_SLOAD(reinterpret_cast<const f16x8*>(row + col * 8)); }
    static __device__ __forceinline__ void row_store(f16* row, int col, int, const f16x8& o) { DS_SSTORE(reinterpret_cast<f16x8*>(row + col * 8), o); }
    static __device__ __forceinline__ raw load(const f16* p) { return *reinterpret_cast<const f16x8*>(p); }
    static __device__ __forceinline__ raw load_stream(const f16* p) { return DS_SLOAD(reinterpret_cast<const f16x8*>(p)); }
    static __device__ __forceinline__ float get(const raw& v, int j) { return (float)v[j]; }
    static __device__ __forceinline__ raw zero() { return f16x8{0, 0, 0, 0, 0, 0, 0, 0}; }
};
template <> struct Row8<float> {
    typedef f32x8 raw;
#ifdef DS_EXP_NO_W4        // A/B (variant "now4"): 8 consecutive fp32 channels per thread, round 3's form
    static __device__ __forceinline__ int chan(int col, int j, int) { return col * 8 + j; }
    static __device__ __forceinline__ raw row_load(const float* row, int col, int) { return load(row + col * 8); }
    static __device__ __forceinline__ raw row_load_stream(const float* row, int col, int) { return load_stream(row + col * 8); }
    static __device__ __forceinline__ void row_store(f16* row, int col, int, const f16x8& o) { DS_SSTORE(reinterpret_cast<f16x8*>(row + col * 8), o); }
#else
    static __device__ __forceinline__ int chan(int col, int j, int nvec) { return (j < 4 ? col : col + nvec) * 4 + (j & 3); }
    static __device__ __forceinline__ raw row_load(const float* row, int col, int nvec) {
        return f32x8{*reinterpret_cast<const f32x4*>(row + col * 4), *reinterpret_cast<const f32x4*>(row + (col + nvec) * 4)};
    }
    static __device__ __forceinline__ raw row_load_stream(const float* row, int col, int nvec) {
        return f32x8{DS_SLOAD(reinterpret_cast<const f32x4*>(row + col * 4)), DS_SLOAD(reinterpret_cast<const f32x4*>(row + (col + nvec) * 4))};
    }
    static __device__ __forceinline__ void row_store(f16* row, int col, int nvec, const f16x8& o) {      // the fp16 output of an fp32 row
        DS_SSTORE(reinterpret_cast<f16x4*>(row + col * 4), (f16x4{o[0], o[1], o[2], o[3]}));
        DS_SSTORE(reinterpret_cast<f16x4*>(row + (col + nvec) * 4), (f16x4{o[4], o[5], o[6], o[7]}));
    }
#endif
    static __device__ __forceinline__ raw load(const float* p) {
        return f32x8{*reinterpret_cast<const f32x4*>(p), *reinterpret_cast<const f32x4*>(p + 4)};
    }
    static __device__ __forceinline__ raw load_stream(const float* p) {
        return f32x8{DS_SLOAD(reinterpret_cast<const f32x4*>(p)), DS_SLOAD(reinterpret_cast<const f32x4*>(p + 4))};
    }
    static __device__ __forceinline__ float get(const raw& v, int j) { return j < 4 ? v.lo[j & 3] : v.hi[j & 3]; }
    static __device__ __forceinline__ raw zero() { return f32x8{f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}}; }
};

// partial sums: grid (nchunks, ninst), 256 threads. part[(inst*nchunks + chunk)*groups + g] = (sum, sumsq).
// A thread owns one 8-channel vector column and every rl-th row of the chunk; four rows are loaded before they are
// accumulated (one accumulation chain per channel, in row order: the sums do not depend on the unrolling), so a
// workgroup keeps ~15 KB in flight instead of one 16-byte load per thread.
// U = rows in flight per thread: GN_U where the grid fills the chip several times over, GN_U_SPARSE where it does not (an 8-GPU
// rank's batch of 2 evaluations: one workgroup per CU or fewer, so the bytes in flight per CU are what one workgroup keeps in
// flight).  The accumulation chain per channel is in row order for every U: the sums are the same bits.
template <typename XT, int U>
__global__ void __launch_bounds__(256)
gn_partial_kernel(const XT* __restrict__ x, float2* __restrict__ part, int rows_per_inst, int C, int groups, int ldx, int chunk_rows) {
    typedef Row8<XT> R8;
    __shared__ float csum[GN_LDS_FLOATS];
    __shared__ float csq[GN_LDS_FLOATS];
    const int tid = threadIdx.x;
    const int chunk = blockIdx.x, inst = blockIdx.y, nchunks = gridDim.x;
    const int r0 = chunk * chunk_rows;
    const int r1 = min(rows_per_inst, r0 + chunk_rows);
    const int nvec = C / 8;
    const XT* base = x + (long)inst * rows_per_inst * ldx;   // input rows may sit in a wider buffer (row stride ldx >= C)
    const int rl = nvec <= 256 ? 256 / nvec : 1;  // row lanes; rl*C <= 2048 + C when rl > 1
    auto accumulate = [&](int col, int rfirst, int rstep, float* s, float* q) {
        const XT* p = base;
        int r = rfirst;
        for (; r + (U - 1) * rstep < r1; r += U * rstep) {
            typename R8::raw v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = R8::row_load(p + (long)(r + u * rstep) * ldx, col, nvec);
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float f = R8::get(v[u], j); s[j] += f; q[j] += f * f; }
        }
        for (; r < r1; r += rstep) {
            const typename R8::raw v = R8::row_load(p + (long)r * ldx, col, nvec);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float f = R8::get(v, j); s[j] += f; q[j] += f * f; }
        }
    };
    if (nvec <= 256) {
        if (tid < rl * nvec) {
            const int col = tid % nvec, rlane = tid / nvec;
            float s[8], q[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { s[j] = 0.0f; q[j] = 0.0f; }
            accumulate(col, r0 + rlane, rl, s, q);
#pragma unroll
            for (int j = 0; j < 8; ++j) { csum[rlane * C + R8::chan(col, j, nvec)] = s[j]; csq[rlane * C + R8::chan(col, j, nvec)] = q[j]; }
        }
    } else {
        for (int col = tid; col < nvec; col += 256) {
            float s[8], q[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { s[j] = 0.0f; q[j] = 0.0f; }
            accumulate(col, r0, 1, s, q);
#pragma unroll
            for (int j = 0; j < 8; ++j) { csum[R8::chan(col, j, nvec)] = s[j]; csq[R8::chan(col, j, nvec)] = q[j]; }
        }
    }
    __syncthreads();
    const int cpg = C / groups;
    if (tid < groups) {   // fixed summation order -> bitwise reproducible
        float s = 0.0f, q = 0.0f;
        for (int l = 0; l < rl; ++l)
            for (int j = 0; j < cpg; ++j) { s += csum[l * C + tid * cpg + j]; q += csq[l * C + tid * cpg + j]; }
        part[((long)inst * nchunks + chunk) * groups + tid] = make_float2(s, q);
    }
}

__global__ void gn_finalize_kernel(const float2* __restrict__ part, float* __restrict__ mean, float* __restrict__ rstd,
                                   int ninst, int nchunks, int groups, double count, float eps) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= ninst * groups) return;
    const int inst = idx / groups, g = idx - inst * groups;
    double s = 0.0, q = 0.0;
    for (int c = 0; c < nchunks; ++c) {
        const float2 p = part[((long)inst * nchunks + c) * groups + g];
        s += (double)p.x; q += (double)p.y;
    }
    const double m = s / count;
    double var = q / count - m * m;
    if (var < 0.0) var = 0.0;
    mean[idx] = (float)m;
    rstd[idx] = (float)(1.0 / sqrt(var + (double)eps));
}

// grid (nchunks, ninst).  Same thread -> (column, row lane) map as the stats kernel: the per-channel scale / shift of a
// thread's 8 channels live in registers, four rows are in flight per thread.  With `part` != nullptr the workgroup first
// reduces the instance's per-chunk partial sums itself (fp64, fixed order: 8 interleaved slices, then the slices in
// order) instead of reading mean / rstd -- the separate finalize launch (1300 per DDIM step, each latency-bound) is gone.
// `xraw` != nullptr (fp32 input only): the raw x rounded to fp16 is written next to y (dense [rows][C]) -- the fp16 A operand of
// a projection that reads the un-normalised tensor (ResBlock skip_connection, openaimodel3d.py:186-193), for free in this pass.
template <typename XT, int U>
__global__ void __launch_bounds__(256)
gn_apply_kernel(const XT* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                const float2* __restrict__ part, const float* __restrict__ gamma, const float* __restrict__ beta,
                f16* __restrict__ y, int rows_per_inst, int C, int groups, int silu, float eps, int ldx, f16* __restrict__ xraw,
                int nchunks, int apply_rows) {
    typedef Row8<XT> R8;
    __shared__ double sred[2][8][32];
    __shared__ float smean[256], srstd[256];
    const int tid = threadIdx.x;
    const int chunk = blockIdx.x, inst = blockIdx.y;     // nchunks = partial-sum chunks of the instance; this workgroup applies
                                                         // rows [chunk * apply_rows, ...): elementwise, any split gives the same bits
    const int cpg = C / groups;
    if (part) {
        for (int g0 = 0; g0 < groups; g0 += 32) {
            const int g = g0 + (tid & 31), sl = tid >> 5;
            double s = 0.0, q = 0.0;
            if (g < groups)
                for (int c = sl; c < nchunks; c += 8) {
                    const float2 p = part[((long)inst * nchunks + c) * groups + g];
                    s += (double)p.x; q += (double)p.y;
                }
            sred[0][sl][tid & 31] = s;
            sred[1][sl][tid & 31] = q;
            __syncthreads();
            if (tid < 32 && g < groups) {
                double ss = 0.0, qq = 0.0;
                for (int k = 0; k < 8; ++k) { ss += sred[0][k][tid]; qq += sred[1][k][tid]; }
                const double count = (double)rows_per_inst * cpg;
                const double m = ss / count;
                double var = qq / count - m * m;
                if (var < 0.0) var = 0.0;
                smean[g] = (float)m;
                srstd[g] = (float)(1.0 / sqrt(var + (double)eps));
            }
            __syncthreads();
        }
    }
    const int r0 = chunk * apply_rows;
    const int r1 = min(rows_per_inst, r0 + apply_rows);
    const int nvec = C / 8;
    const long base = (long)inst * rows_per_inst * C, xbase = (long)inst * rows_per_inst * ldx;
    const int rl = nvec <= 256 ? 256 / nvec : 1;
    const int ncolpass = nvec <= 256 ? 1 : (nvec + 255) / 256;
    for (int cp = 0; cp < ncolpass; ++cp) {
        const int col = nvec <= 256 ? tid % nvec : tid + cp * 256;
        const int rlane = nvec <= 256 ? tid / nvec : 0;
        if (col >= nvec || rlane >= rl) continue;
        float a[8], b[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = R8::chan(col, j, nvec);
            const int g = c / cpg;
            const float r = part ? srstd[g] : rstd[inst * groups + g], m = part ? smean[g] : mean[inst * groups + g];
            a[j] = r * gamma[c];
            b[j] = beta[c] - m * a[j];
        }
        auto one = [&](const typename R8::raw& v) {
            f16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float f = R8::get(v, j) * a[j] + b[j];
                if (silu) f = fast_silu(f);
                o[j] = (f16)f;
            }
            return o;
        };
        auto rawh = [&](const typename R8::raw& v) {
            f16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (f16)R8::get(v, j);
            return o;
        };
        const XT* px = x + xbase;
        f16* py = y + base;
        f16* pr = xraw ? xraw + base : nullptr;
        int r = r0 + rlane;
        for (; r + (U - 1) * rl < r1; r += U * rl) {
            typename R8::raw v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = R8::row_load_stream(px + (long)(r + u * rl) * ldx, col, nvec);
#pragma unroll
            for (int u = 0; u < U; ++u) R8::row_store(py + (long)(r + u * rl) * C, col, nvec, one(v[u]));
            if (pr) {
#pragma unroll
                for (int u = 0; u < U; ++u) R8::row_store(pr + (long)(r + u * rl) * C, col, nvec, rawh(v[u]));
            }
        }
        for (; r < r1; r += rl) {
            const typename R8::raw v = R8::row_load_stream(px + (long)r * ldx, col, nvec);
            R8::row_store(py + (long)r * C, col, nvec, one(v));
            if (pr) R8::row_store(pr + (long)r * C, col, nvec, rawh(v));
        }
    }
}

// Small instances (rows_per_inst <= GN_SMALL_ROWS: the per-frame GroupNorms of the two coarse UNet levels): statistics
// and normalisation in ONE launch, one 1024-thread workgroup per instance -- the instance (<= 1.3 MB) is read twice,
// the second time from L2; no partial-sum buffer, no second launch.  Same thread -> (column, row lane) map and the same
// fixed summation order idea as above (row lanes first, then the channels of a group in order).
constexpr int GN_SMALL_ROWS = 256;
constexpr int GN_SMALL_NT = 1024;

template <typename XT>
__global__ void __launch_bounds__(GN_SMALL_NT)
gn_small_kernel(const XT* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                f16* __restrict__ y, int rows_per_inst, int C, int groups, int silu, float eps, int ldx, f16* __restrict__ xraw) {
    typedef Row8<XT> R8;
    __shared__ float csum[GN_LDS_FLOATS + MAX_C];
    __shared__ float csq[GN_LDS_FLOATS + MAX_C];
    __shared__ float smean[256], srstd[256];
    const int tid = threadIdx.x, inst = blockIdx.x;
    const int nvec = C / 8, cpg = C / groups;
    const XT* base = x + (long)inst * rows_per_inst * ldx;
    f16* ybase = y + (long)inst * rows_per_inst * C;
    const int rl = nvec <= GN_SMALL_NT ? GN_SMALL_NT / nvec : 1;      // row lanes (nvec <= 512 -> rl >= 2)
    const int cap = (GN_LDS_FLOATS + MAX_C) / C;                       // rlu*C floats must fit the LDS arrays
    const int rlu = min(min(rl, 8), cap);
    const bool on = tid < rlu * nvec;
    const int col = tid % nvec, rlane = tid / nvec;
    if (on) {
        float s[8], q[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { s[j] = 0.0f; q[j] = 0.0f; }
        const XT* p = base;
        int r = rlane;
        for (; r + 3 * rlu < rows_per_inst; r += 4 * rlu) {
            typename R8::raw v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = R8::row_load(p + (long)(r + u * rlu) * ldx, col, nvec);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float f = R8::get(v[u], j); s[j] += f; q[j] += f * f; }
        }
        for (; r < rows_per_inst; r += rlu) {
            const typename R8::raw v = R8::row_load(p + (long)r * ldx, col, nvec);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float f = R8::get(v, j); s[j] += f; q[j] += f * f; }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { csum[rlane * C + R8::chan(col, j, nvec)] = s[j]; csq[rlane * C + R8::chan(col, j, nvec)] = q[j]; }
    }
    __syncthreads();
    if (tid < groups) {
        double s = 0.0, q = 0.0;
        for (int l = 0; l < rlu; ++l)
            for (int j = 0; j < cpg; ++j) { s += (double)csum[l * C + tid * cpg + j]; q += (double)csq[l * C + tid * cpg + j]; }
        const double count = (double)rows_per_inst * cpg;
        const double m = s / count;
        double var = q / count - m * m;
        if (var < 0.0) var = 0.0;
        smean[tid] = (float)m;
        srstd[tid] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    if (!on) return;
    float a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = R8::chan(col, j, nvec), g = c / cpg;
        a[j] = srstd[g] * gamma[c];
        b[j] = beta[c] - smean[g] * a[j];
    }
    auto one = [&](const typename R8::raw& v) {
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float f = R8::get(v, j) * a[j] + b[j];
            if (silu) f = fast_silu(f);
            o[j] = (f16)f;
        }
        return o;
    };
    auto rawh = [&](const typename R8::raw& v) {
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (f16)R8::get(v, j);
        return o;
    };
    const XT* px = base;
    f16* py = ybase;
    f16* pr = xraw ? xraw + (long)inst * rows_per_inst * C : nullptr;
    int r = rlane;
    for (; r + 3 * rlu < rows_per_inst; r += 4 * rlu) {
        typename R8::raw v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = R8::row_load(px + (long)(r + u * rlu) * ldx, col, nvec);
#pragma unroll
        for (int u = 0; u < 4; ++u) R8::row_store(py + (long)(r + u * rlu) * C, col, nvec, one(v[u]));
        if (pr) {
#pragma unroll
            for (int u = 0; u < 4; ++u) R8::row_store(pr + (long)(r + u * rlu) * C, col, nvec, rawh(v[u]));
        }
    }
    for (; r < rows_per_inst; r += rlu) {
        const typename R8::raw v = R8::row_load(px + (long)r * ldx, col, nvec);
        R8::row_store(py + (long)r * C, col, nvec, one(v));
        if (pr) R8::row_store(pr + (long)r * C, col, nvec, rawh(v));
    }
}

// Mid-size instances (round 6): statistics and normalisation in ONE launch with the instance read ONCE.  A 512-thread workgroup owns
// (instance, slab of `gslab` whole groups) for ALL rows of the instance and keeps its slab -- up to NCH rows x 16 bytes per thread,
// ~410 KB per CU -- in REGISTERS between the statistics and the normalisation: 6 instead of 10 bytes of traffic per fp32 element (4
// instead of 6 for fp16 rows), one launch instead of two.  Thread (col, rlane) owns one 16-byte column chunk of the slab and rows
// rlane, rlane + rl, ...; per-thread sums in fp32 in row order, the (row lane x channel) partial sums of a group are added in fp64 by
// one wave per group in a fixed order (strided lanes, xor butterfly): bitwise repeatable, and a function of the instance's shape
// only.  The slabs of one instance sit on one XCD (consecutive dispatch slots there), so the 128-byte lines they share are fetched
// into that L2 once and their narrow row pieces of the output meet in it before they are written back.
template <typename XT> struct Chunk16;
// (plain, not non-temporal accesses: the slabs of an instance share cache lines, read and written)
template <> struct Chunk16<float> {
    static constexpr int EPC = 4;
    static __device__ __forceinline__ float get(const f32x4& v, int j) { return v[j]; }
    // statistics units: 4 per chunk -- a channel each
    static __device__ __forceinline__ void accum(const f32x4& v, float (&s)[4], float (&q)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { s[j] += v[j]; q[j] = fmaf(v[j], v[j], q[j]); }
    }
    static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t rs, unsigned off, const float (&o)[4]) {
        const f16x4 h = {(f16)o[0], (f16)o[1], (f16)o[2], (f16)o[3]};
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, h), rs, off, 0, 0);
    }
};
template <> struct Chunk16<f16> {
    static constexpr int EPC = 8;
    static __device__ __forceinline__ float get(const f32x4& v, int j) { return (float)__builtin_bit_cast(f16x8, v)[j]; }
    // statistics units: 4 per chunk -- a PAIR of adjacent channels each (channels per group are even), summed by v_dot2_f32_f16
    // (exact fp16 products, fp32 accumulate): no fp32 copy of the chunk is ever formed in the statistics pass
    static __device__ __forceinline__ void accum(const f32x4& v, float (&s)[4], float (&q)[4]) {
        const f16x8 h = __builtin_bit_cast(f16x8, v);
        const f16x2 one2 = {(f16)1.0f, (f16)1.0f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f16x2 p = {h[2 * j], h[2 * j + 1]};
            s[j] = __builtin_amdgcn_fdot2(p, one2, s[j], false);
            q[j] = __builtin_amdgcn_fdot2(p, p, q[j], false);
        }
    }
    static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t rs, unsigned off, const float (&o)[8]) {
        const f16x8 h = {(f16)o[0], (f16)o[1], (f16)o[2], (f16)o[3], (f16)o[4], (f16)o[5], (f16)o[6], (f16)o[7]};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, h), rs, off, 0, 0);
    }
};
constexpr int GN_RES_NT = 512;
constexpr int GN_RES_NCH_MAX = 52;
typedef __attribute__((address_space(3))) void gn_lds_void;
// LDS-DMA, 16 bytes per lane: lane l of the wave-instruction lands at dst + 16 l (dst wave-uniform); an out-of-range offset zero-fills
__device__ __forceinline__ void gn_dma16(__amdgpu_buffer_rsrc_t rs, void* dst, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (gn_lds_void*)dst, 16, voff, 0, 0, 0);
}

// NCH chunks per thread, the LAST NL of them parked in LDS (fetched by LDS-DMA: no register ever holds them before the statistics pass
// reads them back), the first NCH - NL in registers.  Every global access is a raw BUFFER access whose offset is out of range for rows
// past the instance's end / idle threads (loads return zeros, stores are dropped): straight-line code, all loads in flight at once.
template <typename XT, int NCH, int NL, bool SILU>
__global__ void __launch_bounds__(GN_RES_NT)
gn_resident_kernel(const XT* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, f16* __restrict__ y,
                   f16* __restrict__ xraw, int rows, int C, int groups, float eps, int ldx, int gslab, int nslab, int ninst) {
    typedef Chunk16<XT> CK;
    constexpr int EPC = CK::EPC, NR = NCH - NL;
    constexpr unsigned OOB = 0x80000000u;
    __shared__ __attribute__((aligned(16))) float red[GN_RES_NT * 4];
    __shared__ float smean[8], srstd[8];
    extern __shared__ __attribute__((aligned(16))) unsigned char gn_park[];     // [NL][GN_RES_NT] chunks of 16 bytes
    int inst, slab;
    if ((ninst & 7) == 0) {            // blockIdx % 8 names the XCD: the slabs of an instance share one
        const int r = blockIdx.x >> 3;
        slab = r % nslab;
        inst = (r / nslab) * 8 + (blockIdx.x & 7);
    } else {
        slab = blockIdx.x % nslab;
        inst = blockIdx.x / nslab;
    }
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cpg = C / groups, slab_ch = gslab * cpg, nch_row = slab_ch / EPC, rl = GN_RES_NT / nch_row;
    const bool on = tid < rl * nch_row;
    const int col = tid % nch_row, rlane = tid / nch_row;
    const int c0 = slab * slab_ch + col * EPC;
    const XT* pinst = x + (long)inst * rows * ldx;
    const unsigned row_bytes = (unsigned)ldx * (unsigned)sizeof(XT), step_bytes = (unsigned)rl * row_bytes;
    const unsigned off0 = (unsigned)rlane * row_bytes + (unsigned)c0 * (unsigned)sizeof(XT);
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<XT*>(pinst), 0, (int)((unsigned)rows * row_bytes), 0x00020000);
    // rows this thread holds: k < nk_mine (rlane + k rl < rows); idle threads none
    const int nk_mine = on ? (rows - rlane + rl - 1) / rl : 0;
    if constexpr (NL > 0) {
#pragma unroll
        for (int k = NR; k < NCH; ++k)
            gn_dma16(rsx, gn_park + ((size_t)(k - NR) * GN_RES_NT + wave * 64) * 16, k < nk_mine ? off0 + (unsigned)k * step_bytes : OOB);
    }
    f32x4 v[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k)
        v[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsx, k < nk_mine ? off0 + (unsigned)k * step_bytes : OOB, 0, 0));
    // statistics: 4 units per chunk (a channel for fp32 rows, a pair of channels for fp16 rows), upg units per group
    constexpr int UPE = EPC / 4;                       // channels per unit
    const int upg = cpg / UPE, slab_units = slab_ch / UPE;
    float s[4] = {0.0f, 0.0f, 0.0f, 0.0f}, q[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < NR; ++k) CK::accum(v[k], s, q);
    const f32x4* mine = reinterpret_cast<const f32x4*>(gn_park) + tid;     // this lane's parked chunks: written by its own DMA lane
    if constexpr (NL > 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < NL; ++k) CK::accum(mine[k * GN_RES_NT], s, q);
    }
    // (row lane x unit) partial sums -> one wave per group adds its rl x upg values in fp64, strided lanes then an xor butterfly: a
    // fixed order.  Sums first, then squares through the same array (8 KB: the parked chunks take the rest of the LDS).
    const int g_red = tid >> 6, lane_red = tid & 63;
    double S = 0.0, Q = 0.0;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        if (on) *reinterpret_cast<f32x4*>(red + rlane * slab_units + col * 4) = pass ? f32x4{q[0], q[1], q[2], q[3]} : f32x4{s[0], s[1], s[2], s[3]};
        __syncthreads();
        if (g_red < gslab) {
            double acc = 0.0;
            const int n = rl * upg;
            for (int i = lane_red; i < n; i += 64) {
                const int l = i / upg, j = i - l * upg;
                acc += (double)red[l * slab_units + g_red * upg + j];
            }
#pragma unroll
            for (int sh = 1; sh < 64; sh <<= 1) acc += __shfl_xor(acc, sh);
            if (pass) Q = acc; else S = acc;
        }
        __syncthreads();
    }
    if (g_red < gslab && lane_red == 0) {
        const double count = (double)rows * cpg;
        const double m = S / count;
        double var = Q / count - m * m;
        if (var < 0.0) var = 0.0;
        smean[g_red] = (float)m;
        srstd[g_red] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    float a[EPC], b[EPC];
#pragma unroll
    for (int j = 0; j < EPC; ++j) {
        const int cc = on ? c0 + j : 0;
        const int g = on ? (col * EPC + j) / cpg : 0;
        a[j] = srstd[g] * gamma[cc];
        b[j] = beta[cc] - smean[g] * a[j];
    }
    // dense fp16 outputs [rows][C]: byte offset of this thread's chunk in row rlane, and the step between its rows
    const unsigned out_row = (unsigned)C * 2u, out_step = (unsigned)rl * out_row;
    const unsigned oo0 = (unsigned)rlane * out_row + (unsigned)c0 * 2u;
    const unsigned out_bytes = (unsigned)rows * out_row;
    const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(y + (long)inst * rows * C, 0, (int)out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsr = __builtin_amdgcn_make_buffer_rsrc(xraw ? xraw + (long)inst * rows * C : y, 0, xraw ? (int)out_bytes : 0, 0x00020000);
    auto emit = [&](const f32x4& t, int k) {
        const unsigned off = k < nk_mine ? oo0 + (unsigned)k * out_step : OOB;
        float o[EPC], raw[EPC];
#pragma unroll
        for (int j = 0; j < EPC; ++j) {
            const float f = CK::get(t, j);
            float u = fmaf(f, a[j], b[j]);
            if (SILU) u = fast_silu(u);
            o[j] = u;
            raw[j] = f;
        }
        CK::store(rsy, off, o);
        if (xraw) CK::store(rsr, off, raw);
    };
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        asm volatile("" : "+v"(v[k]));          // the raw chunk, not a converted copy kept from the statistics pass, is what stays live
        emit(v[k], k);
        if (EPC == 8 || (k & 1)) __builtin_amdgcn_sched_barrier(0);     // bounded interleaving: 8 values in flight hide the SiLU's latency, all chunks' would spill
    }
    if constexpr (NL > 0) {
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            emit(mine[k * GN_RES_NT], NR + k);
            if (EPC == 8 || (k & 1)) __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// groups per slab for the resident form, 0 = not applicable: whole groups, 16-byte column chunks, at most 8 groups (one reducing
// wave each), the widest slab whose rows fit NCH_MAX x row lanes -- a function of (rows, C, groups, element size) only
static inline int gn_resident_gslab(int rows, int C, int groups, int elt) {
    const int cpg = C / groups, epc = 16 / elt;
    int best = 0;
    for (int g = 1; g <= 8 && g <= groups; g *= 2) {
        if (groups % g || (g * cpg) % epc || (elt == 2 && (cpg & 1))) continue;     // (fp16 rows: the statistics add channel PAIRS)
        const int nch_row = g * cpg / epc;
        if (nch_row > GN_RES_NT) break;
        const int rl = GN_RES_NT / nch_row;
        if ((rows + rl - 1) / rl <= GN_RES_NCH_MAX) best = g;
    }
    return best;
}

// LayerNorm: one wave per row, NV 8-channel vectors per lane (C <= 512*NV), RW rows per wave with all their loads issued
// before the first reduction (bytes in flight: the 320-channel rows of the first UNet level are only 640 B each).
template <int NV, int RW, typename XT>
__global__ void __launch_bounds__(256)
layernorm_kernel(const XT* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                 f16* __restrict__ y, int rows, int C, float eps, float2* __restrict__ stats) {
    typedef Row8<XT> R8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long row0 = ((long)blockIdx.x * 4 + wave) * RW;
    if (row0 >= rows) return;
    const int nvec = C / 8;
    typename R8::raw t[RW][NV];
#pragma unroll
    for (int rw = 0; rw < RW; ++rw) {
        const long row = row0 + rw < rows ? row0 + rw : rows - 1;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int col = lane + 64 * i;
            t[rw][i] = col < nvec ? R8::load_stream(x + row * C + col * 8) : R8::zero();
        }
    }
    // stats != nullptr: only (mean, rstd) per row are written (ds_layernorm_stats: the normalisation itself is folded into
    // the consumer GEMM, ds_gemm_f16_ln) -- the same two-pass statistics as the full kernel
    f32x4 g0[NV], g1[NV], b0[NV], b1[NV];
    if (!stats) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int col = lane + 64 * i;
            const int c = (col < nvec ? col : 0) * 8;
            g0[i] = *reinterpret_cast<const f32x4*>(gamma + c); g1[i] = *reinterpret_cast<const f32x4*>(gamma + c + 4);
            b0[i] = *reinterpret_cast<const f32x4*>(beta + c);  b1[i] = *reinterpret_cast<const f32x4*>(beta + c + 4);
        }
    }
#pragma unroll
    for (int rw = 0; rw < RW; ++rw) {
        if (row0 + rw >= rows) break;
        float v[NV][8];
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) { v[i][j] = R8::get(t[rw][i], j); s += v[i][j]; }
#pragma unroll
        for (int sh = 1; sh < 64; sh <<= 1) s += __shfl_xor(s, sh);
        const float mean = s / (float)C;
        float q = 0.0f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (lane + 64 * i < nvec) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float d = v[i][j] - mean; q += d * d; }
            }
        }
#pragma unroll
        for (int sh = 1; sh < 64; sh <<= 1) q += __shfl_xor(q, sh);
        const float rstd = rsqrtf(q / (float)C + eps);
        if (stats) {
            if (lane == 0) stats[row0 + rw] = make_float2(mean, rstd);
            continue;
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int col = lane + 64 * i;
            if (col < nvec) {
                f16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float gm = j < 4 ? g0[i][j & 3] : g1[i][j & 3], bt = j < 4 ? b0[i][j & 3] : b1[i][j & 3];
                    o[j] = (f16)((v[i][j] - mean) * rstd * gm + bt);
                }
                DS_SSTORE(reinterpret_cast<f16x8*>(y + (row0 + rw) * C + col * 8), o);
            }
        }
    }
}

}  // namespace

// The caller supplies the scratch for the per-chunk partial sums (nothing here allocates).
extern "C" size_t ds_groupnorm_stats_workspace_floats(int ninst, int rows_per_inst, int groups) {
    const long nchunks = (rows_per_inst + GN_CHUNK_ROWS_MIN - 1) / GN_CHUNK_ROWS_MIN;   // C is not an argument: the smallest chunk
    return (size_t)ninst * groups * nchunks * 2;
}

extern "C" int ds_groupnorm_chunk_rows(int rows_per_inst, int C) { return gn_chunk_rows(rows_per_inst, C); }

extern "C" int ds_groupnorm_stats(const void* x, float* mean, float* rstd, float* workspace, int ninst,
                                     int rows_per_inst, int C, int groups, float eps, void* stream) {
    DS_CHECK_ARG(x && mean && rstd && workspace, "ds_groupnorm_stats: null argument");
    DS_CHECK_ARG(ninst > 0 && rows_per_inst > 0, "ds_groupnorm_stats: ninst/rows_per_inst must be positive");
    DS_CHECK_ARG(C % 8 == 0 && C <= MAX_C && groups > 0 && groups <= 256 && C % groups == 0, "ds_groupnorm_stats: C=%d groups=%d unsupported", C, groups);
    hipStream_t st = (hipStream_t)stream;
    const int chunk_rows = gn_chunk_rows(rows_per_inst, C);
    const int nchunks = (rows_per_inst + chunk_rows - 1) / chunk_rows;
    gn_partial_kernel<f16, GN_U><<<dim3(nchunks, ninst), 256, 0, st>>>((const f16*)x, (float2*)workspace, rows_per_inst, C, groups, C, chunk_rows);
    DS_CHECK_LAUNCH("ds_groupnorm_stats(partial)");
    const int n = ninst * groups;
    gn_finalize_kernel<<<(n + 255) / 256, 256, 0, st>>>((const float2*)workspace, mean, rstd, ninst, nchunks, groups,
                                                       (double)rows_per_inst * (C / groups), eps);
    DS_CHECK_LAUNCH("ds_groupnorm_stats(finalize)");
    return DS_OK;
}

extern "C" int ds_groupnorm_apply(const void* x, const float* mean, const float* rstd, const float* gamma,
                                  const float* beta, void* y, int ninst, int rows_per_inst, int C, int groups, int silu,
                                  void* stream) {
    DS_CHECK_ARG(x && mean && rstd && gamma && beta && y, "ds_groupnorm_apply: null argument");
    DS_CHECK_ARG(ninst > 0 && rows_per_inst > 0, "ds_groupnorm_apply: ninst/rows_per_inst must be positive");
    DS_CHECK_ARG(C % 8 == 0 && C <= MAX_C && groups > 0 && C % groups == 0, "ds_groupnorm_apply: C=%d groups=%d unsupported", C, groups);
    hipStream_t st = (hipStream_t)stream;
    const int nwg = (rows_per_inst + GN_CHUNK_ROWS_MAX - 1) / GN_CHUNK_ROWS_MAX;
    gn_apply_kernel<f16, GN_U><<<dim3(nwg, ninst), 256, 0, st>>>((const f16*)x, mean, rstd, nullptr, gamma, beta, (f16*)y, rows_per_inst, C, groups, silu, 0.0f, C, nullptr,
                                                                0, GN_CHUNK_ROWS_MAX);
    DS_CHECK_LAUNCH("ds_groupnorm_apply");
    return DS_OK;
}

namespace {
// Per-column partial statistics written by the producing GEMM (ds_gemm_f16_stats: colstats[row block of 32][column] = (sum,
// sumsq)) -> the per-chunk group partial sums gn_apply_kernel reduces: part[(inst * nchunks + chunk) * groups + g].  A chunk is
// `rb_per_chunk` row blocks; grid (nchunks, ninst), 256 threads.  A thread owns columns tid, tid + 256, ...: it adds the chunk's
// row blocks of a column in order (coalesced 8-byte loads, all of a column's loads independent), the column sums go through LDS
// and thread g adds the columns of group g in order -- a fixed order throughout.
__global__ void __launch_bounds__(256)
gn_colstats_part_kernel(const float2* __restrict__ colstats, int ld_stats, float2* __restrict__ part, int rows_per_inst, int C, int groups,
                        int rb_per_chunk) {
    __shared__ float csum[MAX_C];
    __shared__ float csq[MAX_C];
    const int tid = threadIdx.x, chunk = blockIdx.x, inst = blockIdx.y, nchunks = gridDim.x;
    const int cpg = C / groups;
    const int rb_inst = rows_per_inst >> 5;                          // row blocks per instance (rows_per_inst % 32 == 0: checked on the host)
    const int rb0 = chunk * rb_per_chunk, rb1 = min(rb_inst, rb0 + rb_per_chunk);
    const float2* base = colstats + ((long)inst * rb_inst + rb0) * ld_stats;
    const int n = rb1 - rb0;
    for (int c = tid; c < C; c += 256) {
        float s = 0.0f, q = 0.0f;
        int r = 0;
        for (; r + 8 <= n; r += 8) {
            float2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = base[(long)(r + u) * ld_stats + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) { s += v[u].x; q += v[u].y; }
        }
        for (; r < n; ++r) { const float2 v = base[(long)r * ld_stats + c]; s += v.x; q += v.y; }
        csum[c] = s;
        csq[c] = q;
    }
    __syncthreads();
    for (int g = tid; g < groups; g += 256) {
        float s = 0.0f, q = 0.0f;
        for (int j = 0; j < cpg; ++j) { s += csum[g * cpg + j]; q += csq[g * cpg + j]; }
        part[((long)inst * nchunks + chunk) * groups + g] = make_float2(s, q);
    }
}

template <typename XT>
int groupnorm_rows_colstats(const XT* x, int ldx, const float2* colstats, int ld_stats, const float* gamma, const float* beta, f16* y,
                            f16* x_f16, float* workspace, int ninst, int rows_per_inst, int C, int groups, float eps, int silu, hipStream_t st) {
    // chunks of 8 row blocks (256 rows) like the statistics pass; the apply reduces the chunks of its instance in fp64
    int rb_per_chunk = 8;
    const int rb_inst = rows_per_inst >> 5;
    while ((rb_inst + rb_per_chunk - 1) / rb_per_chunk > 160) rb_per_chunk *= 2;
    const int nchunks = (rb_inst + rb_per_chunk - 1) / rb_per_chunk;
    gn_colstats_part_kernel<<<dim3(nchunks, ninst), 256, 0, st>>>(colstats, ld_stats, (float2*)workspace, rows_per_inst, C, groups, rb_per_chunk);
    DS_CHECK_LAUNCH("ds_groupnorm_rows_colstats(partials)");
    constexpr int US = sizeof(XT) == 2 ? GN_U_SPARSE : GN_U_SPARSE / 2;
    int apply_rows = GN_CHUNK_ROWS_MAX;
    while (apply_rows > 32 && (long)((rows_per_inst + apply_rows - 1) / apply_rows) * ninst < 512) apply_rows /= 2;
    const int napply = (rows_per_inst + apply_rows - 1) / apply_rows;
    if ((long)napply * ninst <= GN_SPARSE_WGS)
        gn_apply_kernel<XT, US><<<dim3(napply, ninst), 256, 0, st>>>(x, nullptr, nullptr, (const float2*)workspace, gamma, beta, y,
                                                                     rows_per_inst, C, groups, silu, eps, ldx, x_f16, nchunks, apply_rows);
    else
        gn_apply_kernel<XT, GN_U><<<dim3(napply, ninst), 256, 0, st>>>(x, nullptr, nullptr, (const float2*)workspace, gamma, beta, y,
                                                                       rows_per_inst, C, groups, silu, eps, ldx, x_f16, nchunks, apply_rows);
    DS_CHECK_LAUNCH("ds_groupnorm_rows_colstats(apply)");
    return DS_OK;
}

template <typename XT>
int groupnorm_rows(const XT* x, int ldx, const float* gamma, const float* beta, f16* y, f16* x_f16, float* workspace, int ninst,
                   int rows_per_inst, int C, int groups, float eps, int silu, hipStream_t st, bool onepass = false) {
    // The path depends on the instance's shape only, never on how many instances a launch holds: the two forms sum in different
    // orders, and a batch must equal its separate forwards bit for bit (what keeps rank-sharded runs identical to the
    // single-process panorama).  Until round 3 a `ninst >= 64` condition sat here: a batch of 2 evaluations (an 8-GPU rank's
    // share) took the two-launch form for the 160- and 40-row per-frame norms of levels 3-4, a batch of 4 or more this one.
#ifdef DS_EXP_GN_COUNT_THRESHOLD       // diagnostic variant "gncount" (build.py): round 2's form, fails test_groupnorm_is_batch_invariant
    if (rows_per_inst <= GN_SMALL_ROWS && C / 8 <= 512 && ninst >= 64) {
#else
    if (rows_per_inst <= GN_SMALL_ROWS && C / 8 <= 512) {
#endif
        gn_small_kernel<XT><<<ninst, GN_SMALL_NT, 0, st>>>(x, gamma, beta, y, rows_per_inst, C, groups, silu, eps, ldx, x_f16);
        DS_CHECK_LAUNCH("ds_groupnorm(small)");
        return DS_OK;
    }
    // mid-size instances (per-frame norms of levels 1-2, the joint-T norms of levels 3-4) in ONE launch with the instance read once
    // (gn_resident_kernel): OPT-IN (ds_groupnorm_rows_onepass; DS_GN_RESIDENT=1 in the "tune" build).  Measured on MI355X
    // (profiles/r6_notes.md section 4): per launch 0.6-1.0x of the two-launch form, but in the step it does not pay -- cfg3 -0.2 %, the
    // sphere stage +5 % (a 512-thread, 233-register, 144 KB workgroup needs a whole CU to itself: unlike the small workgroups of the
    // two-launch form it cannot start in the tail of the kernel before it) -- and its sums follow another order (other last bits).
    const int gslab = (onepass || DS_TUNE_INT("DS_GN_RESIDENT", 0) != 0) ? gn_resident_gslab(rows_per_inst, C, groups, (int)sizeof(XT)) : 0;
    if (onepass && gslab == 0) {
        ds_set_error("ds_groupnorm_rows_onepass: no one-pass form for rows_per_inst=%d C=%d groups=%d (ds_groupnorm_onepass_applies)", rows_per_inst, C, groups);
        return DS_EINVAL;
    }
    if (gslab > 0) {
        const int cpg = C / groups, nch_row = gslab * cpg / (16 / (int)sizeof(XT)), rl = GN_RES_NT / nch_row;
        const int rpt = (rows_per_inst + rl - 1) / rl, nslab = groups / gslab;
        const long nwg = (long)ninst * nslab;
        // rows per thread: 13 / 26 in registers; up to 52 as 36 in registers + 16 parked in LDS (128 KB)
        auto go = [&](auto kern, size_t dyn) -> int {
            if (dyn > 0) {
                static bool attr_set = false;
                if (!attr_set) {
                    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) != hipSuccess) {
                        ds_set_error("ds_groupnorm(resident): hipFuncSetAttribute failed");
                        return DS_ELAUNCH;
                    }
                    attr_set = true;
                }
            }
            kern<<<nwg, GN_RES_NT, dyn, st>>>(x, gamma, beta, y, x_f16, rows_per_inst, C, groups, eps, ldx, gslab, nslab, ninst);
            return DS_OK;
        };
        constexpr size_t PARK = (size_t)16 * GN_RES_NT * 16;
        int rc;
        if (rpt <= 13) rc = silu ? go(&gn_resident_kernel<XT, 13, 0, true>, 0) : go(&gn_resident_kernel<XT, 13, 0, false>, 0);
        else if (rpt <= 26) rc = silu ? go(&gn_resident_kernel<XT, 26, 0, true>, 0) : go(&gn_resident_kernel<XT, 26, 0, false>, 0);
        else rc = silu ? go(&gn_resident_kernel<XT, GN_RES_NCH_MAX, 16, true>, PARK) : go(&gn_resident_kernel<XT, GN_RES_NCH_MAX, 16, false>, PARK);
        if (rc) return rc;
        DS_CHECK_LAUNCH("ds_groupnorm(resident)");
        return DS_OK;
    }
    const int chunk_rows = gn_chunk_rows(rows_per_inst, C);
    const int nchunks = (rows_per_inst + chunk_rows - 1) / chunk_rows;
    // What follows is free to depend on the launch's size -- none of it changes a sum: the rows in flight per thread (see
    // gn_partial_kernel) and how the rows of an instance are split over the apply's workgroups (elementwise; 256 rows per
    // workgroup on a dense grid, down to 32 where the launch would not give every CU two workgroups).
    constexpr int US = sizeof(XT) == 2 ? GN_U_SPARSE : GN_U_SPARSE / 2;
    const int sparse_max = (int)DS_TUNE_INT("DS_GN_SPARSE_WGS", GN_SPARSE_WGS);   // 0: dense-grid forms only (diagnostic, "tune" build variant)
    int apply_rows = GN_CHUNK_ROWS_MAX;
    if (sparse_max > 0)
        while (apply_rows > 32 && (long)((rows_per_inst + apply_rows - 1) / apply_rows) * ninst < 512) apply_rows /= 2;
    const int napply = (rows_per_inst + apply_rows - 1) / apply_rows;
    if ((long)nchunks * ninst <= sparse_max)
        gn_partial_kernel<XT, US><<<dim3(nchunks, ninst), 256, 0, st>>>(x, (float2*)workspace, rows_per_inst, C, groups, ldx, chunk_rows);
    else
        gn_partial_kernel<XT, GN_U><<<dim3(nchunks, ninst), 256, 0, st>>>(x, (float2*)workspace, rows_per_inst, C, groups, ldx, chunk_rows);
    DS_CHECK_LAUNCH("ds_groupnorm(stats)");
    if ((long)napply * ninst <= sparse_max)
        gn_apply_kernel<XT, US><<<dim3(napply, ninst), 256, 0, st>>>(x, nullptr, nullptr, (const float2*)workspace, gamma, beta, y,
                                                                     rows_per_inst, C, groups, silu, eps, ldx, x_f16, nchunks, apply_rows);
    else
        gn_apply_kernel<XT, GN_U><<<dim3(napply, ninst), 256, 0, st>>>(x, nullptr, nullptr, (const float2*)workspace, gamma, beta, y,
                                                                       rows_per_inst, C, groups, silu, eps, ldx, x_f16, nchunks, apply_rows);
    DS_CHECK_LAUNCH("ds_groupnorm(apply)");
    return DS_OK;
}
}  // namespace

// x_dtype DS_F16 / DS_F32; ldx: row stride of x in elements (>= C, multiple of 8): the input may be a column slice of a wider
// row-major buffer (the UNet writes skip tensors straight into the buffer the decoder side concatenates in, unet.py).
// y is dense fp16 [rows][C]; x_f16 (fp32 input only, may be NULL): fp16(x), dense [rows][C].
extern "C" int ds_groupnorm_rows(const void* x, int x_dtype, int ldx, const float* gamma, const float* beta, void* y, void* x_f16,
                                 float* workspace, int ninst, int rows_per_inst, int C, int groups, float eps, int silu, void* stream) {
    DS_CHECK_ARG(x && gamma && beta && y && workspace, "ds_groupnorm_rows: null argument");
    DS_CHECK_ARG(x_dtype == DS_F16 || x_dtype == DS_F32, "ds_groupnorm_rows: x_dtype must be DS_F16 or DS_F32");
    DS_CHECK_ARG(ninst > 0 && rows_per_inst > 0, "ds_groupnorm_rows: ninst/rows_per_inst must be positive");
    DS_CHECK_ARG(C % 8 == 0 && C <= MAX_C && groups > 0 && groups <= 256 && C % groups == 0, "ds_groupnorm_rows: C=%d groups=%d unsupported", C, groups);
    DS_CHECK_ARG(ldx >= C && ldx % 8 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0, "ds_groupnorm_rows: ldx=%d (>= C, multiple of 8, x 16-byte aligned)", ldx);
    DS_CHECK_ARG(!x_f16 || x_dtype == DS_F32, "ds_groupnorm_rows: the fp16 copy of x is only produced from fp32 input");
    hipStream_t st = (hipStream_t)stream;
    if (x_dtype == DS_F32)
        return groupnorm_rows<float>((const float*)x, ldx, gamma, beta, (f16*)y, (f16*)x_f16, workspace, ninst, rows_per_inst, C, groups, eps, silu, st);
    return groupnorm_rows<f16>((const f16*)x, ldx, gamma, beta, (f16*)y, nullptr, workspace, ninst, rows_per_inst, C, groups, eps, silu, st);
}

// The one-launch, read-once form (gn_resident_kernel) for the instance shapes it exists for (ds_groupnorm_onepass_applies != 0): same
// arguments and result tolerance as ds_groupnorm_rows, statistics summed in another order.  Opt-in: see groupnorm_rows.
extern "C" int ds_groupnorm_onepass_applies(int rows_per_inst, int C, int groups, int x_dtype) {
    if (rows_per_inst <= GN_SMALL_ROWS || C <= 0 || groups <= 0 || C % groups || C % 8) return 0;
    return gn_resident_gslab(rows_per_inst, C, groups, x_dtype == DS_F32 ? 4 : 2) > 0;
}

extern "C" int ds_groupnorm_rows_onepass(const void* x, int x_dtype, int ldx, const float* gamma, const float* beta, void* y, void* x_f16,
                                         int ninst, int rows_per_inst, int C, int groups, float eps, int silu, void* stream) {
    DS_CHECK_ARG(x && gamma && beta && y, "ds_groupnorm_rows_onepass: null argument");
    DS_CHECK_ARG(x_dtype == DS_F16 || x_dtype == DS_F32, "ds_groupnorm_rows_onepass: x_dtype must be DS_F16 or DS_F32");
    DS_CHECK_ARG(ninst > 0 && rows_per_inst > GN_SMALL_ROWS, "ds_groupnorm_rows_onepass: ninst must be positive, rows_per_inst > %d", GN_SMALL_ROWS);
    DS_CHECK_ARG(C % 8 == 0 && C <= MAX_C && groups > 0 && groups <= 256 && C % groups == 0, "ds_groupnorm_rows_onepass: C=%d groups=%d unsupported", C, groups);
    DS_CHECK_ARG(ldx >= C && ldx % 8 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0, "ds_groupnorm_rows_onepass: ldx=%d (>= C, multiple of 8, x 16-byte aligned)", ldx);
    DS_CHECK_ARG(!x_f16 || x_dtype == DS_F32, "ds_groupnorm_rows_onepass: the fp16 copy of x is only produced from fp32 input");
    hipStream_t st = (hipStream_t)stream;
    if (x_dtype == DS_F32)
        return groupnorm_rows<float>((const float*)x, ldx, gamma, beta, (f16*)y, (f16*)x_f16, nullptr, ninst, rows_per_inst, C, groups, eps, silu, st, true);
    return groupnorm_rows<f16>((const f16*)x, ldx, gamma, beta, (f16*)y, nullptr, nullptr, ninst, rows_per_inst, C, groups, eps, silu, st, true);
}

// GroupNorm whose statistics come from the producing GEMM (ds_gemm_f16_stats): colstats[(row / 32) * ld_stats + column] = (sum,
// sumsq) over the rows of that 32-row block, for the columns of x (the pointer already offset to x's first column when x is a
// column slice).  rows_per_inst % 32 == 0.  workspace: ds_groupnorm_stats_workspace_floats(...) floats, as for ds_groupnorm_rows.
extern "C" int ds_groupnorm_rows_colstats(const void* x, int x_dtype, int ldx, const float* colstats, int ld_stats, const float* gamma,
                                          const float* beta, void* y, void* x_f16, float* workspace, int ninst, int rows_per_inst, int C,
                                          int groups, float eps, int silu, void* stream) {
    DS_CHECK_ARG(x && colstats && gamma && beta && y && workspace, "ds_groupnorm_rows_colstats: null argument");
    DS_CHECK_ARG(x_dtype == DS_F16 || x_dtype == DS_F32, "ds_groupnorm_rows_colstats: x_dtype must be DS_F16 or DS_F32");
    DS_CHECK_ARG(ninst > 0 && rows_per_inst > 0 && rows_per_inst % 32 == 0, "ds_groupnorm_rows_colstats: rows_per_inst=%d must be a positive multiple of 32", rows_per_inst);
    DS_CHECK_ARG(C % 8 == 0 && C <= MAX_C && groups > 0 && groups <= 256 && C % groups == 0, "ds_groupnorm_rows_colstats: C=%d groups=%d unsupported", C, groups);
    DS_CHECK_ARG(ldx >= C && ldx % 8 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0, "ds_groupnorm_rows_colstats: ldx=%d (>= C, multiple of 8, x 16-byte aligned)", ldx);
    DS_CHECK_ARG(ld_stats >= C && (reinterpret_cast<uintptr_t>(colstats) & 7) == 0, "ds_groupnorm_rows_colstats: ld_stats=%d must be >= C", ld_stats);
    DS_CHECK_ARG(!x_f16 || x_dtype == DS_F32, "ds_groupnorm_rows_colstats: the fp16 copy of x is only produced from fp32 input");
    hipStream_t st = (hipStream_t)stream;
    if (x_dtype == DS_F32)
        return groupnorm_rows_colstats<float>((const float*)x, ldx, (const float2*)colstats, ld_stats, gamma, beta, (f16*)y, (f16*)x_f16, workspace,
                                              ninst, rows_per_inst, C, groups, eps, silu, st);
    return groupnorm_rows_colstats<f16>((const f16*)x, ldx, (const float2*)colstats, ld_stats, gamma, beta, (f16*)y, nullptr, workspace, ninst,
                                        rows_per_inst, C, groups, eps, silu, st);
}

extern "C" int ds_groupnorm_f16_strided(const void* x, int ldx, const float* gamma, const float* beta, void* y, float* workspace,
                                        int ninst, int rows_per_inst, int C, int groups, float eps, int silu, void* stream) {
    return ds_groupnorm_rows(x, DS_F16, ldx, gamma, beta, y, nullptr, workspace, ninst, rows_per_inst, C, groups, eps, silu, stream);
}

extern "C" int ds_groupnorm_f16(const void* x, const float* gamma, const float* beta, void* y, float* workspace, int ninst,
                                int rows_per_inst, int C, int groups, float eps, int silu, void* stream) {
    return ds_groupnorm_rows(x, DS_F16, C, gamma, beta, y, nullptr, workspace, ninst, rows_per_inst, C, groups, eps, silu, stream);
}

namespace {
template <typename XT>
void layernorm_launch(const XT* x, const float* gamma, const float* beta, f16* y, int rows, int C, float eps, float2* stats, hipStream_t st) {
    const int nv = (C / 8 + 63) / 64;
    if (nv == 1) layernorm_kernel<1, 4, XT><<<(rows + 15) / 16, 256, 0, st>>>(x, gamma, beta, y, rows, C, eps, stats);
    else if (nv == 2) layernorm_kernel<2, 2, XT><<<(rows + 7) / 8, 256, 0, st>>>(x, gamma, beta, y, rows, C, eps, stats);
    else if (nv == 3) layernorm_kernel<3, 2, XT><<<(rows + 7) / 8, 256, 0, st>>>(x, gamma, beta, y, rows, C, eps, stats);
    else layernorm_kernel<5, 1, XT><<<(rows + 3) / 4, 256, 0, st>>>(x, gamma, beta, y, rows, C, eps, stats);
}
}  // namespace

// x fp16 or fp32 [rows][C] (x_dtype), y fp16.
extern "C" int ds_layernorm_rows(const void* x, int x_dtype, const float* gamma, const float* beta, void* y, int rows, int C,
                                 float eps, void* stream) {
    DS_CHECK_ARG(x && gamma && beta && y, "ds_layernorm: null argument");
    DS_CHECK_ARG(x_dtype == DS_F16 || x_dtype == DS_F32, "ds_layernorm: x_dtype must be DS_F16 or DS_F32");
    DS_CHECK_ARG(rows > 0 && C % 8 == 0 && C <= 2560, "ds_layernorm: rows=%d C=%d unsupported (C %% 8 == 0, C <= 2560)", rows, C);
    hipStream_t st = (hipStream_t)stream;
    DS_CHECK_ARG((reinterpret_cast<uintptr_t>(gamma) & 15) == 0 && (reinterpret_cast<uintptr_t>(beta) & 15) == 0, "ds_layernorm: gamma/beta must be 16-byte aligned");
    if (x_dtype == DS_F32) layernorm_launch<float>((const float*)x, gamma, beta, (f16*)y, rows, C, eps, nullptr, st);
    else layernorm_launch<f16>((const f16*)x, gamma, beta, (f16*)y, rows, C, eps, nullptr, st);
    DS_CHECK_LAUNCH("ds_layernorm");
    return DS_OK;
}

extern "C" int ds_layernorm(const void* x, const float* gamma, const float* beta, void* y, int rows, int C, float eps,
                            void* stream) {
    return ds_layernorm_rows(x, DS_F16, gamma, beta, y, rows, C, eps, stream);
}

extern "C" int ds_layernorm_stats(const void* x, float* stats, int rows, int C, float eps, void* stream) {
    DS_CHECK_ARG(x && stats, "ds_layernorm_stats: null argument");
    DS_CHECK_ARG(rows > 0 && C % 8 == 0 && C <= 2560, "ds_layernorm_stats: rows=%d C=%d unsupported (C %% 8 == 0, C <= 2560)", rows, C);
    DS_CHECK_ARG((reinterpret_cast<uintptr_t>(stats) & 7) == 0, "ds_layernorm_stats: stats must be 8-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    layernorm_launch<f16>((const f16*)x, nullptr, nullptr, nullptr, rows, C, eps, reinterpret_cast<float2*>(stats), st);
    DS_CHECK_LAUNCH("ds_layernorm_stats");
    return DS_OK;
}
