// Ring-panorama tile ops for gfx950: window gather / scatter with wrap-around in F, H, W, the fused
// re-noise + mask-mix, and the fused CFG + DDIM update.  All are HBM-bound data movement / elementwise
// work: 16-byte vector accesses along W when the window origin allows it, a scalar path otherwise.
//
// Bit-exactness contract: the fp32 arithmetic below repeats the reference's torch op sequence
// (pipeline/scheduler.py:60-110, utils/tensor_utils.py:19-39) one rounding per op, so FMA contraction
// is disabled for this translation unit.
#pragma clang fp contract(off)
#include "common.h"

namespace {

struct Origins {
    int n;
    int f0[DS_MAX_WINDOWS], y0[DS_MAX_WINDOWS], x0[DS_MAX_WINDOWS];
};


// ---------------------------------------------------------------------------------------------
// gather: tiles[i][c][f][y][x] = pano[c][(f0+f)%F][(y0+y)%H][(x0+x)%W]
// VEC = elements moved per thread along x (1 = scalar fallback). The vector path requires
// x0 % VEC == 0, W % VEC == 0, tw % VEC == 0 so a group never straddles the seam and stays aligned.
// ---------------------------------------------------------------------------------------------
template <typename T, int VEC>
__global__ void __launch_bounds__(256) ring_gather_kernel(const T* __restrict__ pano, T* __restrict__ tiles,
                                                          ds_ring_geom g, Origins o) {
    const int twv = g.tw / VEC;
    const long per_tile = (long)g.C * g.tf * g.th * twv;
    const long total = per_tile * o.n;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        int i = (int)(idx / per_tile);
        long r = idx - (long)i * per_tile;
        int xv = (int)(r % twv); r /= twv;
        int y = (int)(r % g.th); r /= g.th;
        int f = (int)(r % g.tf);
        int c = (int)(r / g.tf);
        int sf = (o.f0[i] + f) % g.F;
        int sy = (o.y0[i] + y) % g.H;
        int sx = (o.x0[i] + xv * VEC) % g.W;
        const T* src = pano + (((long)c * g.F + sf) * g.H + sy) * g.W + sx;
        T* dst = tiles + idx * VEC;
        if (VEC == 1) {
            *dst = *src;
        } else {
            *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(src);
        }
    }
}

template <int VEC>
__global__ void __launch_bounds__(256) ring_gather_mask_kernel(const uint8_t* __restrict__ pano,
                                                               uint8_t* __restrict__ tiles, ds_ring_geom g, Origins o) {
    const int twv = g.tw / VEC;
    const long per_tile = (long)g.tf * g.th * twv;
    const long total = per_tile * o.n;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        int i = (int)(idx / per_tile);
        long r = idx - (long)i * per_tile;
        int xv = (int)(r % twv); r /= twv;
        int y = (int)(r % g.th);
        int f = (int)(r / g.th);
        int sf = (o.f0[i] + f) % g.F;
        int sy = (o.y0[i] + y) % g.H;
        int sx = (o.x0[i] + xv * VEC) % g.W;
        const uint8_t* src = pano + ((long)sf * g.H + sy) * g.W + sx;
        uint8_t* dst = tiles + idx * VEC;
        if (VEC == 1) {
            *dst = *src;
        } else {
            *reinterpret_cast<uint2*>(dst) = *reinterpret_cast<const uint2*>(src);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// scatter3: pano_lat <- x_prev, pano_x0 <- x0, mask <- 1 (overwrite; windows of a launch are disjoint)
// ---------------------------------------------------------------------------------------------
template <typename T, int VEC>
__global__ void __launch_bounds__(256) ring_scatter3_kernel(T* __restrict__ pano_lat, T* __restrict__ pano_x0,
                                                            uint8_t* __restrict__ mask, const T* __restrict__ xprev,
                                                            const T* __restrict__ x0t, ds_ring_geom g, Origins o) {
    const int twv = g.tw / VEC;
    const long per_tile = (long)g.C * g.tf * g.th * twv;
    const long total = per_tile * o.n;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        int i = (int)(idx / per_tile);
        long r = idx - (long)i * per_tile;
        int xv = (int)(r % twv); r /= twv;
        int y = (int)(r % g.th); r /= g.th;
        int f = (int)(r % g.tf);
        int c = (int)(r / g.tf);
        int sf = (o.f0[i] + f) % g.F;
        int sy = (o.y0[i] + y) % g.H;
        int sx = (o.x0[i] + xv * VEC) % g.W;
        long plane = ((long)sf * g.H + sy) * g.W + sx;
        long dst = (long)c * g.F * g.H * g.W + plane;
        if (VEC == 1) {
            if (pano_lat) pano_lat[dst] = xprev[idx];
            if (pano_x0) pano_x0[dst] = x0t[idx];
            if (mask && c == 0) mask[plane] = 1;
        } else {
            if (pano_lat) *reinterpret_cast<uint4*>(pano_lat + dst) = *reinterpret_cast<const uint4*>(xprev + idx * VEC);
            if (pano_x0) *reinterpret_cast<uint4*>(pano_x0 + dst) = *reinterpret_cast<const uint4*>(x0t + idx * VEC);
            if (mask && c == 0) {
                if (VEC == 8) *reinterpret_cast<uint2*>(mask + plane) = make_uint2(0x01010101u, 0x01010101u);
                else *reinterpret_cast<uint32_t*>(mask + plane) = 0x01010101u;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 + Box-Muller (perf-mode noise; NOT the torch CPU stream)
// ---------------------------------------------------------------------------------------------
__device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                     uint32_t out[4]) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
        uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ inline void normal4(uint64_t seed, uint64_t ctr, float z[4]) {
    uint32_t r[4];
    philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const float two_pi = 6.283185307179586f;
    // (0,1] uniforms
    float u0 = ((float)(r[0] >> 8) + 1.0f) * (1.0f / 16777216.0f);
    float u1 = (float)(r[1] >> 8) * (1.0f / 16777216.0f);
    float u2 = ((float)(r[2] >> 8) + 1.0f) * (1.0f / 16777216.0f);
    float u3 = (float)(r[3] >> 8) * (1.0f / 16777216.0f);
    float ra = sqrtf(-2.0f * __logf(u0)), rb = sqrtf(-2.0f * __logf(u2));
    z[0] = ra * __cosf(two_pi * u1);
    z[1] = ra * __sinf(two_pi * u1);
    z[2] = rb * __cosf(two_pi * u3);
    z[3] = rb * __sinf(two_pi * u3);
}

template <typename T> __device__ inline float ldf(const T* p, long i) { return (float)p[i]; }
template <typename T> __device__ inline void stf(T* p, long i, float v) { p[i] = (T)v; }

// ---------------------------------------------------------------------------------------------
// renoise + mix, 4 elements per thread along x (tw % 4 == 0 required by the host wrapper, else VEC=1)
// ---------------------------------------------------------------------------------------------
template <typename T, int VEC>
__global__ void __launch_bounds__(256) renoise_mix_kernel(T* __restrict__ tiles, const uint8_t* __restrict__ mask,
                                                          const T* __restrict__ noise, float c, float s, float ratio,
                                                          float one_minus_ratio, int mask_frame0, uint64_t seed,
                                                          uint64_t offset, ds_ring_geom g, int n) {
    const int twv = g.tw / VEC;
    const long plane = (long)g.th * g.tw;
    const long per_tile_v = (long)g.C * g.tf * g.th * twv;
    const long total = per_tile_v * n;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        int i = (int)(idx / per_tile_v);
        long r = idx - (long)i * per_tile_v;
        int xv = (int)(r % twv); r /= twv;
        int y = (int)(r % g.th); r /= g.th;
        int f = (int)(r % g.tf);
        long e0 = idx * VEC;
        long m0 = ((long)i * g.tf + (mask_frame0 ? 0 : f)) * plane + (long)y * g.tw + (long)xv * VEC;
        float z[4];
        if (!noise) normal4(seed, offset + (uint64_t)idx, z);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float x = ldf(tiles, e0 + j);
            float zz = noise ? ldf(noise, e0 + j) : z[j & 3];
            float m = (float)mask[m0 + j];
            // scheduler.py:108  x_b = c * x_a + s * eps
            float t1 = c * x;
            float t2 = s * zz;
            float noised = t1 + t2;
            // tensor_utils.py:30-37
            float w1 = x * one_minus_ratio;
            float w2 = noised * ratio;
            float mixed = w1 + w2;
            float non_mask = x * (1.0f - m);
            float mask_area = mixed * m;
            stf(tiles, e0 + j, non_mask + mask_area);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Per-step residual merge of the non-overlapping grid loop (t2v_normal_pipeline.py:445-468):
//   dense : out = curr*r + noised*(1-r)
//   sparse: out = curr, except (p = step parity, rows / columns of each [H][W] plane)
//           out[p::2, ::2]     = r*curr[(1-p)::2, ::2] + (1-r)*noised[::2, ::2]
//           out[(1-p)::2, 1::2] = r*curr[p::2, 1::2]    + (1-r)*noised[::2, ::2]
// Same fp32 operations in the same order as the reference's tensor expressions (two products, one sum).
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) residual_merge_kernel(const T* __restrict__ curr, const T* __restrict__ noised,
                                                              T* __restrict__ out, long planes, int H, int W, float r,
                                                              float one_minus_r, int parity, int sparse) {
    const long total = planes * H * W;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % W);
        const long q = idx / W;
        const int y = (int)(q % H);
        const long base = (q / H) * (long)H * W;
        float v;
        if (!sparse) {
            const float t1 = ldf(curr, idx) * r;
            const float t2 = ldf(noised, idx) * one_minus_r;
            v = t1 + t2;
        } else if ((y & 1) == parity && (x & 1) == 0) {
            const int k = y >> 1;                                     // y = parity + 2k
            const float t1 = r * ldf(curr, base + (long)((1 - parity) + 2 * k) * W + x);
            const float t2 = one_minus_r * ldf(noised, base + (long)(2 * k) * W + x);
            v = t1 + t2;
        } else if ((y & 1) == 1 - parity && (x & 1) == 1) {
            const int k = y >> 1;                                     // y = (1-parity) + 2k
            const float t1 = r * ldf(curr, base + (long)(parity + 2 * k) * W + x);
            const float t2 = one_minus_r * ldf(noised, base + (long)(2 * k) * W + (x - 1));
            v = t1 + t2;
        } else {
            v = ldf(curr, idx);
        }
        stf(out, idx, v);
    }
}

// ---------------------------------------------------------------------------------------------
// CFG + DDIM
// ---------------------------------------------------------------------------------------------
template <typename T, typename E>
__global__ void __launch_bounds__(256) cfg_ddim_kernel(const T* __restrict__ x, const E* __restrict__ ec,
                                                       const E* __restrict__ eu, float guidance, float sq1m,
                                                       float sqrt_at, float sqrt_aprev, float dir_coef, float sigma,
                                                       const T* __restrict__ noise, T* __restrict__ xprev,
                                                       T* __restrict__ x0o, long total) {
    for (long idx = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; idx < total;
         idx += (long)gridDim.x * blockDim.x * 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            long k = idx + j;
            if (k >= total) break;
            float xv = (float)x[k];
            float e = (float)ec[k];
            if (eu) {
                float u = (float)eu[k];
                float d = e - u;          // t2v_sphere_panorama_pipeline.py:599
                float gd = guidance * d;
                e = u + gd;
            }
            float a = sq1m * e;           // scheduler.py:83
            float num = xv - a;
            float p0 = num / sqrt_at;
            float dir = dir_coef * e;     // :85
            float b = sqrt_aprev * p0;    // :89
            float xp = b + dir;
            float nz = noise ? sigma * (float)noise[k] : 0.0f;
            xp = xp + nz;
            xprev[k] = (T)xp;
            x0o[k] = (T)p0;
        }
    }
}


// ---------------------------------------------------------------------------------------------
// Fused forms of the ring step's tile ops (SURVEY R2+R5+R6 and R7+R8+R3): the window is re-noised under the mask while it is
// gathered, and the CFG + DDIM update goes straight into the panoramas -- the tile tensors in between are never written.
// Same fp32 operations in the same order as the separate kernels (bit-identical results), 4 elements per thread along x.
// ---------------------------------------------------------------------------------------------
struct TileIds { long off[DS_MAX_WINDOWS]; };     // Philox counter offset of tile i (in-kernel noise)

// VEC consecutive elements (VEC = 1 or 4; 4: the address is a multiple of 4 elements) as floats and back; one rounding on the way back
template <int VEC> __device__ inline void ldv(const float* p, float o[4]) {
    if (VEC == 4) { const f32x4 v = *reinterpret_cast<const f32x4*>(p); o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3]; }
    else o[0] = p[0];
}
template <int VEC> __device__ inline void ldv(const f16* p, float o[4]) {
    if (VEC == 4) { const f16x4 v = *reinterpret_cast<const f16x4*>(p); o[0] = (float)v[0]; o[1] = (float)v[1]; o[2] = (float)v[2]; o[3] = (float)v[3]; }
    else o[0] = (float)p[0];
}
template <int VEC> __device__ inline void ldv(const uint8_t* p, float o[4]) {
    if (VEC == 4) { const uint32_t v = *reinterpret_cast<const uint32_t*>(p); o[0] = (float)(v & 255u); o[1] = (float)((v >> 8) & 255u); o[2] = (float)((v >> 16) & 255u); o[3] = (float)(v >> 24); }
    else o[0] = (float)p[0];
}
template <int VEC> __device__ inline void stv(float* p, const float v[4]) {
    if (VEC == 4) *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
    else p[0] = v[0];
}
template <int VEC> __device__ inline void stv(f16* p, const float v[4]) {
    if (VEC == 4) *reinterpret_cast<f16x4*>(p) = f16x4{(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
    else p[0] = (f16)v[0];
}

template <typename T, int VEC>
__global__ void __launch_bounds__(256)
ring_gather_renoise_kernel(const T* __restrict__ pano, const uint8_t* __restrict__ mask_pano, T* __restrict__ tiles,
                           uint8_t* __restrict__ mask_tiles, const T* __restrict__ noise, float c, float s, float ratio,
                           float one_minus_ratio, int mask_frame0, uint64_t seed, TileIds ids, ds_ring_geom g, Origins o, int aligned) {
    // VEC = elements per thread = normals per Philox counter (4 whenever tw % 4 == 0, like ds_renoise_mix: the same stream);
    // aligned: the group of VEC neither straddles the W seam nor is misaligned in the panorama -> vector loads, else per element
    const int twv = g.tw / VEC;
    const long per_tile_v = (long)g.C * g.tf * g.th * twv;
    const long total = per_tile_v * o.n;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int i = (int)(idx / per_tile_v);
        const long within = idx - (long)i * per_tile_v;
        long r = within;
        const int xv = (int)(r % twv); r /= twv;
        const int y = (int)(r % g.th); r /= g.th;
        const int f = (int)(r % g.tf);
        const int cc = (int)(r / g.tf);
        const int sy = (o.y0[i] + y) % g.H;
        const int sx = (o.x0[i] + xv * VEC) % g.W;             // VEC == 1, or the group neither straddles the seam nor is misaligned
        const int sf = (o.f0[i] + f) % g.F;
        const int mf = (o.f0[i] + (mask_frame0 ? 0 : f)) % g.F;
        const T* src = pano + (((long)cc * g.F + sf) * g.H + sy) * g.W + sx;
        const uint8_t* msrc = mask_pano + ((long)mf * g.H + sy) * g.W + sx;
        const long e0 = idx * VEC;
        float z[4], xs[4], ms4[4], out4[4];
        if (!noise) normal4(seed, (uint64_t)ids.off[i] + (uint64_t)within, z);
        else ldv<VEC>(noise + e0, z);
        if (aligned) {
            ldv<VEC>(src, xs);
            ldv<VEC>(msrc, ms4);
        } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const int sxj = (o.x0[i] + xv * VEC + j) % g.W;
                xs[j] = (float)pano[(((long)cc * g.F + sf) * g.H + sy) * g.W + sxj];
                ms4[j] = (float)mask_pano[((long)mf * g.H + sy) * g.W + sxj];
            }
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float x = xs[j];
            const float zz = z[j];
            const float m = ms4[j];
            // scheduler.py:108  x_b = c * x_a + s * eps
            const float t1 = c * x;
            const float t2 = s * zz;
            const float noised = t1 + t2;
            // tensor_utils.py:30-37
            const float w1 = x * one_minus_ratio;
            const float w2 = noised * ratio;
            const float mixed = w1 + w2;
            const float non_mask = x * (1.0f - m);
            const float mask_area = mixed * m;
            out4[j] = non_mask + mask_area;
        }
        stv<VEC>(tiles + e0, out4);
        if (mask_tiles && cc == 0) {      // the window's mask itself (merge-prev reads it again after the update)
            const uint8_t* ms = mask_pano + ((long)sf * g.H + sy) * g.W + sx;
            uint8_t* md = mask_tiles + (((long)i * g.tf + f) * g.th + y) * g.tw + (long)xv * VEC;
            if (VEC == 4 && aligned) *reinterpret_cast<uint32_t*>(md) = *reinterpret_cast<const uint32_t*>(ms);
            else {
#pragma unroll
                for (int j = 0; j < VEC; ++j) md[j] = mask_pano[((long)sf * g.H + sy) * g.W + (o.x0[i] + xv * VEC + j) % g.W];
            }
        }
    }
}

template <typename T, typename E, int VEC>
__global__ void __launch_bounds__(256)
cfg_ddim_scatter_kernel(const T* __restrict__ x, const E* __restrict__ ec, const E* __restrict__ eu, float guidance, float sq1m,
                        float sqrt_at, float sqrt_aprev, float dir_coef, float sigma, const T* __restrict__ noise,
                        T* __restrict__ pano_lat, T* __restrict__ pano_x0, uint8_t* __restrict__ mask, ds_ring_geom g, Origins o) {
    const int twv = g.tw / VEC;
    const long per_tile_v = (long)g.C * g.tf * g.th * twv;
    const long total = per_tile_v * o.n;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int i = (int)(idx / per_tile_v);
        long r = idx - (long)i * per_tile_v;
        const int xv = (int)(r % twv); r /= twv;
        const int y = (int)(r % g.th); r /= g.th;
        const int f = (int)(r % g.tf);
        const int cc = (int)(r / g.tf);
        const int sf = (o.f0[i] + f) % g.F;
        const int sy = (o.y0[i] + y) % g.H;
        const int sx = (o.x0[i] + xv * VEC) % g.W;
        const long plane = ((long)sf * g.H + sy) * g.W + sx;
        const long dst = (long)cc * g.F * g.H * g.W + plane;
        const long e0 = idx * VEC;
        float xs[4], es[4], us[4] = {0, 0, 0, 0}, ns[4] = {0, 0, 0, 0}, xp4[4], p04[4];
        ldv<VEC>(x + e0, xs);
        ldv<VEC>(ec + e0, es);
        if (eu) ldv<VEC>(eu + e0, us);
        if (noise) ldv<VEC>(noise + e0, ns);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float xval = xs[j];
            float e = es[j];
            if (eu) {
                const float u = us[j];
                const float dd = e - u;          // t2v_sphere_panorama_pipeline.py:599
                const float gd = guidance * dd;
                e = u + gd;
            }
            const float a = sq1m * e;            // scheduler.py:83
            const float num = xval - a;
            const float p0 = num / sqrt_at;
            const float dir = dir_coef * e;      // :85
            const float b = sqrt_aprev * p0;     // :89
            float xp = b + dir;
            const float nz = noise ? sigma * ns[j] : 0.0f;
            xp = xp + nz;
            xp4[j] = xp;
            p04[j] = p0;
        }
        stv<VEC>(pano_lat + dst, xp4);
        stv<VEC>(pano_x0 + dst, p04);
        if (mask && cc == 0) {
            if (VEC == 4) *reinterpret_cast<uint32_t*>(mask + plane) = 0x01010101u;
            else mask[plane] = 1;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Sphere path: perspective view <-> equirect panorama through a host-computed int32 index map
// (utils/panorama_tensor_utils.py:154-202).  idx[i][p] < 0 = skip (invalid sample / duplicate-target loser).
// ---------------------------------------------------------------------------------------------
// Frame windows (RingPanoramaLatentProxy, utils/ring_panorama_tensor_utils.py:262-314): a tile holds tf frames starting
// at panorama frame f0[i] and wrapping modulo F; tf == F with f0 == nullptr is the plain PanoramaLatentProxy case.
template <typename T>
__global__ void __launch_bounds__(256) map_gather_kernel(const T* __restrict__ pano, T* __restrict__ tiles,
                                                         const int* __restrict__ idx, const int* __restrict__ f0, int C,
                                                         int F, int tf, int HW, int P, int n) {
    const long total = (long)n * C * tf * P;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int p = (int)(t % P);
        long r = t / P;
        const int tt = (int)(r % tf);
        r /= tf;
        const int c = (int)(r % C);
        const int i = (int)(r / C);
        const int fr = ((f0 ? f0[i] : 0) + tt) % F;
        const int src = idx[(long)i * P + p];
        tiles[t] = src >= 0 ? pano[((long)c * F + fr) * HW + src] : (T)0;
    }
}

template <typename T>
__global__ void __launch_bounds__(256) map_scatter3_kernel(T* __restrict__ pano_lat, T* __restrict__ pano_x0,
                                                           uint8_t* __restrict__ mask, const T* __restrict__ xprev,
                                                           const T* __restrict__ x0t, const int* __restrict__ idx,
                                                           const int* __restrict__ f0, int C, int F, int tf, int HW, int P,
                                                           int n, int mask_per_frame) {
    const long total = (long)n * C * tf * P;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int p = (int)(t % P);
        long r = t / P;
        const int tt = (int)(r % tf);
        r /= tf;
        const int c = (int)(r % C);
        const int i = (int)(r / C);
        const int dst = idx[(long)i * P + p];
        if (dst < 0) continue;
        const int fr = ((f0 ? f0[i] : 0) + tt) % F;
        const long o = ((long)c * F + fr) * HW + dst;
        if (pano_lat) pano_lat[o] = xprev[t];
        if (pano_x0) pano_x0[o] = x0t[t];
        if (mask && c == 0) {
            if (mask_per_frame) mask[(long)fr * HW + dst] = 1;
            else if (tt == 0) mask[dst] = 1;
        }
    }
}

// Bilinear splat with normaliser (set_view_tensor_bilinear, utils/panorama_tensor_utils.py:98-152): every view pixel
// contributes to 4 panorama pixels with weights (1-du)(1-dv) ...; a panorama pixel becomes sum(v*w) / sum(w).
// The reference accumulates with index_add_; here the host inverts the map once per view into a CSR list per TARGET
// pixel whose entries keep the reference's summation order (tap 00 sources ascending, then 01, 10, 11), so the sum is
// a plain per-thread loop: no atomics, bit-reproducible, and bit-identical to the CPU index_add_ result.
template <typename T>
__global__ void __launch_bounds__(256)
map_splat_kernel(T* __restrict__ pano, const T* __restrict__ view, const int* __restrict__ tgt, const int* __restrict__ row_ptr,
                 const int* __restrict__ src, const float* __restrict__ wgt, int CF, int HW, int P, int ntgt) {
    const long total = (long)CF * ntgt;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int k = (int)(t % ntgt);
        const int cf = (int)(t / ntgt);
        const int b = row_ptr[k], e = row_ptr[k + 1];
        float acc = 0.0f, ws = 0.0f;
        for (int j = b; j < e; ++j) {
            const float w = wgt[j];
            acc = acc + (float)view[(long)cf * P + src[j]] * w;
            ws = ws + w;
        }
        if (ws > 0.0f) pano[(long)cf * HW + tgt[k]] = (T)(acc / ws);
    }
}

// Weighted multi-tap gather: get_view_tensor_interpolate = F.grid_sample(bilinear, border) with the taps and weights resolved
// on the host (utils/panorama_tensor_utils.py:28-51).  One thread per output element, taps summed in order in fp32.
template <typename T>
__global__ void __launch_bounds__(256)
map_gather_taps_kernel(const T* __restrict__ pano, T* __restrict__ out, const int* __restrict__ idx, const float* __restrict__ wgt,
                       int ntaps, int C, int F, int f0, int tf, int HW, int P) {
    const long total = (long)C * tf * P;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int p = (int)(t % P);
        long r = t / P;
        const int tt = (int)(r % tf);
        const int c = (int)(r / tf);
        const T* plane = pano + ((long)c * F + (f0 + tt) % F) * HW;
        float acc = 0.0f;
        for (int k = 0; k < ntaps; ++k) {
            const float w = wgt[(long)k * P + p];
            if (w != 0.0f) acc = acc + (float)plane[idx[(long)k * P + p]] * w;
        }
        out[t] = (T)acc;
    }
}

int fill_origins(Origins& o, const ds_ring_geom* g, const int32_t* origins, int n, const char* who) {
    DS_CHECK_ARG(g && origins, "%s: null geom/origins", who);
    DS_CHECK_ARG(n >= 1 && n <= DS_MAX_WINDOWS, "%s: n=%d out of [1,%d]", who, n, DS_MAX_WINDOWS);
    DS_CHECK_ARG(g->C > 0 && g->F > 0 && g->H > 0 && g->W > 0 && g->tf > 0 && g->th > 0 && g->tw > 0,
                 "%s: non-positive geometry", who);
    DS_CHECK_ARG(g->dtype == DS_F16 || g->dtype == DS_F32, "%s: dtype must be DS_F16/DS_F32", who);
    o.n = n;
    for (int i = 0; i < n; ++i) {
        int f0 = origins[3 * i], y0 = origins[3 * i + 1], x0 = origins[3 * i + 2];
        // RingLatent asserts 0 <= lo < hi <= 2*size (shift_window_utils.py:73-75)
        DS_CHECK_ARG(f0 >= 0 && f0 + g->tf <= 2 * g->F, "%s: Invalid frame_begin %d and frame_end %d", who, f0, f0 + g->tf);
        DS_CHECK_ARG(y0 >= 0 && y0 + g->th <= 2 * g->H, "%s: Invalid pos_top %d and pos_down %d", who, y0, y0 + g->th);
        DS_CHECK_ARG(x0 >= 0 && x0 + g->tw <= 2 * g->W, "%s: Invalid pos_left %d and pos_right %d", who, x0, x0 + g->tw);
        o.f0[i] = f0; o.y0[i] = y0; o.x0[i] = x0;
    }
    return DS_OK;
}

bool vec_ok(const ds_ring_geom* g, const Origins& o, int vec) {
    if (g->W % vec || g->tw % vec) return false;
    for (int i = 0; i < o.n; ++i)
        if (o.x0[i] % vec) return false;
    return true;
}

inline int grid_for(long work) {
    long b = (work + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace

extern "C" int ds_ring_gather(const void* pano, const uint8_t* mask_pano, void* tiles, uint8_t* mask_tiles,
                              const ds_ring_geom* g, const int32_t* origins, int n, void* stream) {
    Origins o;
    int rc = fill_origins(o, g, origins, n, "ds_ring_gather");
    if (rc) return rc;
    DS_CHECK_ARG((pano && tiles) || (mask_pano && mask_tiles), "ds_ring_gather: nothing to gather");
    DS_CHECK_ARG((mask_pano == nullptr) == (mask_tiles == nullptr), "ds_ring_gather: mask_pano/mask_tiles must both be set or both NULL");
    hipStream_t st = (hipStream_t)stream;
    if (pano) {
        if (g->dtype == DS_F16) {
            if (vec_ok(g, o, 8)) {
                long work = (long)n * g->C * g->tf * g->th * (g->tw / 8);
                ring_gather_kernel<f16, 8><<<grid_for(work), 256, 0, st>>>((const f16*)pano, (f16*)tiles, *g, o);
            } else {
                long work = (long)n * g->C * g->tf * g->th * g->tw;
                ring_gather_kernel<f16, 1><<<grid_for(work), 256, 0, st>>>((const f16*)pano, (f16*)tiles, *g, o);
            }
        } else {
            if (vec_ok(g, o, 4)) {
                long work = (long)n * g->C * g->tf * g->th * (g->tw / 4);
                ring_gather_kernel<float, 4><<<grid_for(work), 256, 0, st>>>((const float*)pano, (float*)tiles, *g, o);
            } else {
                long work = (long)n * g->C * g->tf * g->th * g->tw;
                ring_gather_kernel<float, 1><<<grid_for(work), 256, 0, st>>>((const float*)pano, (float*)tiles, *g, o);
            }
        }
        DS_CHECK_LAUNCH("ds_ring_gather");
    }
    if (mask_pano) {
        if (vec_ok(g, o, 8)) {
            long work = (long)n * g->tf * g->th * (g->tw / 8);
            ring_gather_mask_kernel<8><<<grid_for(work), 256, 0, st>>>(mask_pano, mask_tiles, *g, o);
        } else {
            long work = (long)n * g->tf * g->th * g->tw;
            ring_gather_mask_kernel<1><<<grid_for(work), 256, 0, st>>>(mask_pano, mask_tiles, *g, o);
        }
        DS_CHECK_LAUNCH("ds_ring_gather(mask)");
    }
    return DS_OK;
}

extern "C" int ds_ring_scatter3(void* pano_latent, void* pano_x0, uint8_t* mask_pano, const void* x_prev_tiles,
                                const void* x0_tiles, const ds_ring_geom* g, const int32_t* origins, int n,
                                void* stream) {
    Origins o;
    int rc = fill_origins(o, g, origins, n, "ds_ring_scatter3");
    if (rc) return rc;
    // set_window_latent asserts the window does not overlap itself (shift_window_utils.py:145-147)
    DS_CHECK_ARG(g->tw <= g->W && g->th <= g->H && g->tf <= g->F, "ds_ring_scatter3: warp should not occur");
    DS_CHECK_ARG(!pano_latent || x_prev_tiles, "ds_ring_scatter3: pano_latent without x_prev_tiles");
    DS_CHECK_ARG(!pano_x0 || x0_tiles, "ds_ring_scatter3: pano_x0 without x0_tiles");
    hipStream_t st = (hipStream_t)stream;
    if (g->dtype == DS_F16) {
        if (vec_ok(g, o, 8)) {
            long work = (long)n * g->C * g->tf * g->th * (g->tw / 8);
            ring_scatter3_kernel<f16, 8><<<grid_for(work), 256, 0, st>>>((f16*)pano_latent, (f16*)pano_x0, mask_pano,
                                                                        (const f16*)x_prev_tiles, (const f16*)x0_tiles, *g, o);
        } else {
            long work = (long)n * g->C * g->tf * g->th * g->tw;
            ring_scatter3_kernel<f16, 1><<<grid_for(work), 256, 0, st>>>((f16*)pano_latent, (f16*)pano_x0, mask_pano,
                                                                        (const f16*)x_prev_tiles, (const f16*)x0_tiles, *g, o);
        }
    } else {
        if (vec_ok(g, o, 4)) {
            long work = (long)n * g->C * g->tf * g->th * (g->tw / 4);
            ring_scatter3_kernel<float, 4><<<grid_for(work), 256, 0, st>>>((float*)pano_latent, (float*)pano_x0, mask_pano,
                                                                          (const float*)x_prev_tiles, (const float*)x0_tiles, *g, o);
        } else {
            long work = (long)n * g->C * g->tf * g->th * g->tw;
            ring_scatter3_kernel<float, 1><<<grid_for(work), 256, 0, st>>>((float*)pano_latent, (float*)pano_x0, mask_pano,
                                                                          (const float*)x_prev_tiles, (const float*)x0_tiles, *g, o);
        }
    }
    DS_CHECK_LAUNCH("ds_ring_scatter3");
    return DS_OK;
}

extern "C" int ds_ring_gather_renoise(const void* pano, const uint8_t* mask_pano, void* tiles, uint8_t* mask_tiles, const void* noise,
                                      float c, float s, float ratio, float one_minus_ratio, int mask_frame0, uint64_t seed,
                                      const int64_t* tile_offsets, const ds_ring_geom* g, const int32_t* origins, int n, void* stream) {
    Origins o;
    int rc = fill_origins(o, g, origins, n, "ds_ring_gather_renoise");
    if (rc) return rc;
    DS_CHECK_ARG(pano && mask_pano && tiles, "ds_ring_gather_renoise: null argument");
    DS_CHECK_ARG(noise || tile_offsets, "ds_ring_gather_renoise: in-kernel noise needs the tiles' counter offsets");
    DS_CHECK_ARG(g->dtype == DS_F16 || g->dtype == DS_F32, "ds_ring_gather_renoise: bad dtype");
    TileIds ids;
    for (int i = 0; i < n; ++i) ids.off[i] = tile_offsets ? (long)tile_offsets[i] : 0;
    hipStream_t st = (hipStream_t)stream;
    const bool v4 = (g->tw % 4) == 0;                 // as ds_renoise_mix: 4 normals per counter along x
    const int aligned = vec_ok(g, o, 4) ? 1 : 0;
    const long work = (long)n * g->C * g->tf * g->th * (v4 ? g->tw / 4 : g->tw);
    if (g->dtype == DS_F16) {
        if (v4) ring_gather_renoise_kernel<f16, 4><<<grid_for(work), 256, 0, st>>>((const f16*)pano, mask_pano, (f16*)tiles, mask_tiles, (const f16*)noise, c, s, ratio, one_minus_ratio, mask_frame0, seed, ids, *g, o, aligned);
        else ring_gather_renoise_kernel<f16, 1><<<grid_for(work), 256, 0, st>>>((const f16*)pano, mask_pano, (f16*)tiles, mask_tiles, (const f16*)noise, c, s, ratio, one_minus_ratio, mask_frame0, seed, ids, *g, o, 0);
    } else {
        if (v4) ring_gather_renoise_kernel<float, 4><<<grid_for(work), 256, 0, st>>>((const float*)pano, mask_pano, (float*)tiles, mask_tiles, (const float*)noise, c, s, ratio, one_minus_ratio, mask_frame0, seed, ids, *g, o, aligned);
        else ring_gather_renoise_kernel<float, 1><<<grid_for(work), 256, 0, st>>>((const float*)pano, mask_pano, (float*)tiles, mask_tiles, (const float*)noise, c, s, ratio, one_minus_ratio, mask_frame0, seed, ids, *g, o, 0);
    }
    DS_CHECK_LAUNCH("ds_ring_gather_renoise");
    return DS_OK;
}

extern "C" int ds_cfg_ddim_scatter(const void* x, const void* eps_c, const void* eps_u, int eps_dtype, float guidance,
                                   float sqrt_one_minus_at, float sqrt_at, float sqrt_a_prev, float dir_coef, float sigma, const void* noise,
                                   void* pano_latent, void* pano_x0, uint8_t* mask_pano, const ds_ring_geom* g, const int32_t* origins,
                                   int n, void* stream) {
    Origins o;
    int rc = fill_origins(o, g, origins, n, "ds_cfg_ddim_scatter");
    if (rc) return rc;
    DS_CHECK_ARG(x && eps_c && pano_latent && pano_x0, "ds_cfg_ddim_scatter: null argument");
    DS_CHECK_ARG(g->tw <= g->W && g->th <= g->H && g->tf <= g->F, "ds_cfg_ddim_scatter: warp should not occur");
    DS_CHECK_ARG(sigma == 0.0f || noise, "ds_cfg_ddim_scatter: sigma != 0 needs a noise tensor");
    DS_CHECK_ARG(eps_dtype == DS_F16 || eps_dtype == DS_F32, "ds_cfg_ddim_scatter: bad eps dtype");
    DS_CHECK_ARG(g->dtype == DS_F16 || g->dtype == DS_F32, "ds_cfg_ddim_scatter: bad dtype");
    hipStream_t st = (hipStream_t)stream;
    const bool v4 = vec_ok(g, o, 4);
    const long work = (long)n * g->C * g->tf * g->th * (v4 ? g->tw / 4 : g->tw);
    const int grid = grid_for(work);
#define DS_LAUNCH_CDS(T, E, V)                                                                                                            \
    cfg_ddim_scatter_kernel<T, E, V><<<grid, 256, 0, st>>>((const T*)x, (const E*)eps_c, (const E*)eps_u, guidance, sqrt_one_minus_at,       \
                                                           sqrt_at, sqrt_a_prev, dir_coef, sigma, (const T*)noise, (T*)pano_latent,        \
                                                           (T*)pano_x0, mask_pano, *g, o)
    if (g->dtype == DS_F16) {
        if (eps_dtype == DS_F16) { if (v4) DS_LAUNCH_CDS(f16, f16, 4); else DS_LAUNCH_CDS(f16, f16, 1); }
        else { if (v4) DS_LAUNCH_CDS(f16, float, 4); else DS_LAUNCH_CDS(f16, float, 1); }
    } else {
        if (eps_dtype == DS_F16) { if (v4) DS_LAUNCH_CDS(float, f16, 4); else DS_LAUNCH_CDS(float, f16, 1); }
        else { if (v4) DS_LAUNCH_CDS(float, float, 4); else DS_LAUNCH_CDS(float, float, 1); }
    }
#undef DS_LAUNCH_CDS
    DS_CHECK_LAUNCH("ds_cfg_ddim_scatter");
    return DS_OK;
}

extern "C" int ds_residual_merge(const void* curr, const void* noised, void* out, int dtype, long planes, int H, int W,
                                 float r, float one_minus_r, int parity, int sparse, void* stream) {
    DS_CHECK_ARG(curr && noised && out && out != curr && out != noised, "ds_residual_merge: null or aliased argument");
    DS_CHECK_ARG(planes > 0 && H > 0 && W > 0, "ds_residual_merge: sizes must be positive");
    DS_CHECK_ARG(!sparse || (H % 2 == 0 && W % 2 == 0), "ds_residual_merge: the sparse pattern needs even H and W (got %dx%d)", H, W);
    DS_CHECK_ARG(parity == 0 || parity == 1, "ds_residual_merge: parity must be 0 or 1");
    const long work = planes * H * W;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DS_F16) residual_merge_kernel<f16><<<grid_for(work), 256, 0, st>>>((const f16*)curr, (const f16*)noised, (f16*)out, planes, H, W, r, one_minus_r, parity, sparse);
    else if (dtype == DS_F32) residual_merge_kernel<float><<<grid_for(work), 256, 0, st>>>((const float*)curr, (const float*)noised, (float*)out, planes, H, W, r, one_minus_r, parity, sparse);
    else DS_CHECK_ARG(false, "ds_residual_merge: bad dtype %d", dtype);
    DS_CHECK_LAUNCH("ds_residual_merge");
    return DS_OK;
}

extern "C" int ds_renoise_mix(void* tiles, const uint8_t* mask_tiles, const void* noise, float c, float s,
                              float ratio, float one_minus_ratio, int mask_frame0, uint64_t seed, uint64_t offset,
                              const ds_ring_geom* g, int n, void* stream) {
    DS_CHECK_ARG(tiles && mask_tiles && g, "ds_renoise_mix: null argument");
    DS_CHECK_ARG(n >= 1, "ds_renoise_mix: n=%d", n);
    DS_CHECK_ARG(g->dtype == DS_F16 || g->dtype == DS_F32, "ds_renoise_mix: bad dtype");
    hipStream_t st = (hipStream_t)stream;
    const bool v4 = (g->tw % 4) == 0;
    long work = (long)n * g->C * g->tf * g->th * (v4 ? g->tw / 4 : g->tw);
    if (g->dtype == DS_F16) {
        if (v4) renoise_mix_kernel<f16, 4><<<grid_for(work), 256, 0, st>>>((f16*)tiles, mask_tiles, (const f16*)noise, c, s, ratio, one_minus_ratio, mask_frame0, seed, offset, *g, n);
        else renoise_mix_kernel<f16, 1><<<grid_for(work), 256, 0, st>>>((f16*)tiles, mask_tiles, (const f16*)noise, c, s, ratio, one_minus_ratio, mask_frame0, seed, offset, *g, n);
    } else {
        if (v4) renoise_mix_kernel<float, 4><<<grid_for(work), 256, 0, st>>>((float*)tiles, mask_tiles, (const float*)noise, c, s, ratio, one_minus_ratio, mask_frame0, seed, offset, *g, n);
        else renoise_mix_kernel<float, 1><<<grid_for(work), 256, 0, st>>>((float*)tiles, mask_tiles, (const float*)noise, c, s, ratio, one_minus_ratio, mask_frame0, seed, offset, *g, n);
    }
    DS_CHECK_LAUNCH("ds_renoise_mix");
    return DS_OK;
}

extern "C" int ds_cfg_ddim(const void* x, const void* eps_c, const void* eps_u, int eps_dtype, float guidance,
                           float sqrt_one_minus_at, float sqrt_at, float sqrt_a_prev, float dir_coef, float sigma,
                           const void* noise, void* x_prev, void* x0, const ds_ring_geom* g, int n, void* stream) {
    DS_CHECK_ARG(x && eps_c && x_prev && x0 && g, "ds_cfg_ddim: null argument");
    DS_CHECK_ARG(n >= 1, "ds_cfg_ddim: n=%d", n);
    DS_CHECK_ARG(sigma == 0.0f || noise, "ds_cfg_ddim: sigma != 0 needs a noise tensor");
    DS_CHECK_ARG(eps_dtype == DS_F16 || eps_dtype == DS_F32, "ds_cfg_ddim: bad eps dtype");
    hipStream_t st = (hipStream_t)stream;
    long total = (long)n * g->C * g->tf * g->th * g->tw;
    int grid = grid_for((total + 3) / 4);
#define DS_LAUNCH_CFG(T, E)                                                                                         \
    cfg_ddim_kernel<T, E><<<grid, 256, 0, st>>>((const T*)x, (const E*)eps_c, (const E*)eps_u, guidance,            \
                                                 sqrt_one_minus_at, sqrt_at, sqrt_a_prev, dir_coef, sigma,           \
                                                 (const T*)noise, (T*)x_prev, (T*)x0, total)
    if (g->dtype == DS_F16) {
        if (eps_dtype == DS_F16) DS_LAUNCH_CFG(f16, f16); else DS_LAUNCH_CFG(f16, float);
    } else {
        if (eps_dtype == DS_F16) DS_LAUNCH_CFG(float, f16); else DS_LAUNCH_CFG(float, float);
    }
#undef DS_LAUNCH_CFG
    DS_CHECK_LAUNCH("ds_cfg_ddim");
    return DS_OK;
}

static int map_gather_impl(const void* pano, void* tiles, const int32_t* idx, const int32_t* f0, int C, int F, int tf,
                           int HW, int P, int n, int dtype, void* stream, const char* name) {
    DS_CHECK_ARG(pano && tiles && idx, "%s: null argument", name);
    DS_CHECK_ARG(C > 0 && F > 0 && tf > 0 && tf <= F && HW > 0 && P > 0 && n > 0, "%s: sizes must be positive, tf <= F", name);
    hipStream_t st = (hipStream_t)stream;
    const long work = (long)n * C * tf * P;
    if (dtype == DS_F16) map_gather_kernel<f16><<<grid_for(work), 256, 0, st>>>((const f16*)pano, (f16*)tiles, idx, f0, C, F, tf, HW, P, n);
    else if (dtype == DS_F32) map_gather_kernel<float><<<grid_for(work), 256, 0, st>>>((const float*)pano, (float*)tiles, idx, f0, C, F, tf, HW, P, n);
    else if (dtype == 2) map_gather_kernel<uint8_t><<<grid_for(work), 256, 0, st>>>((const uint8_t*)pano, (uint8_t*)tiles, idx, f0, C, F, tf, HW, P, n);
    else DS_CHECK_ARG(false, "%s: bad dtype %d", name, dtype);
    DS_CHECK_LAUNCH(name);
    return DS_OK;
}

static int map_scatter3_impl(void* pano_latent, void* pano_x0, uint8_t* mask_pano, const void* x_prev_tiles,
                             const void* x0_tiles, const int32_t* idx, const int32_t* f0, int C, int F, int tf, int HW,
                             int P, int n, int mask_per_frame, int dtype, void* stream, const char* name) {
    DS_CHECK_ARG(idx, "%s: null index map", name);
    DS_CHECK_ARG(C > 0 && F > 0 && tf > 0 && tf <= F && HW > 0 && P > 0 && n > 0, "%s: sizes must be positive, tf <= F", name);
    DS_CHECK_ARG(!pano_latent || x_prev_tiles, "%s: pano_latent without x_prev_tiles", name);
    DS_CHECK_ARG(!pano_x0 || x0_tiles, "%s: pano_x0 without x0_tiles", name);
    hipStream_t st = (hipStream_t)stream;
    const long work = (long)n * C * tf * P;
    if (dtype == DS_F16)
        map_scatter3_kernel<f16><<<grid_for(work), 256, 0, st>>>((f16*)pano_latent, (f16*)pano_x0, mask_pano, (const f16*)x_prev_tiles, (const f16*)x0_tiles, idx, f0, C, F, tf, HW, P, n, mask_per_frame);
    else if (dtype == DS_F32)
        map_scatter3_kernel<float><<<grid_for(work), 256, 0, st>>>((float*)pano_latent, (float*)pano_x0, mask_pano, (const float*)x_prev_tiles, (const float*)x0_tiles, idx, f0, C, F, tf, HW, P, n, mask_per_frame);
    else DS_CHECK_ARG(false, "%s: bad dtype %d", name, dtype);
    DS_CHECK_LAUNCH(name);
    return DS_OK;
}

extern "C" int ds_map_gather(const void* pano, void* tiles, const int32_t* idx, int CF, int HW, int P, int n, int dtype,
                             void* stream) {
    return map_gather_impl(pano, tiles, idx, nullptr, CF, 1, 1, HW, P, n, dtype, stream, "ds_map_gather");
}

extern "C" int ds_map_scatter3(void* pano_latent, void* pano_x0, uint8_t* mask_pano, const void* x_prev_tiles,
                               const void* x0_tiles, const int32_t* idx, int CF, int HW, int P, int n, int dtype,
                               void* stream) {
    // fused (channel, frame) index, one mask byte per panorama pixel: C = CF "channels" of one frame each
    return map_scatter3_impl(pano_latent, pano_x0, mask_pano, x_prev_tiles, x0_tiles, idx, nullptr, CF, 1, 1, HW, P, n, 0,
                             dtype, stream, "ds_map_scatter3");
}

extern "C" int ds_map_gather_frames(const void* pano, void* tiles, const int32_t* idx, const int32_t* f0, int C, int F,
                                    int tf, int HW, int P, int n, int dtype, void* stream) {
    return map_gather_impl(pano, tiles, idx, f0, C, F, tf, HW, P, n, dtype, stream, "ds_map_gather_frames");
}

extern "C" int ds_map_scatter3_frames(void* pano_latent, void* pano_x0, uint8_t* mask_pano, const void* x_prev_tiles,
                                      const void* x0_tiles, const int32_t* idx, const int32_t* f0, int C, int F, int tf,
                                      int HW, int P, int n, int dtype, void* stream) {
    return map_scatter3_impl(pano_latent, pano_x0, mask_pano, x_prev_tiles, x0_tiles, idx, f0, C, F, tf, HW, P, n, 1, dtype,
                             stream, "ds_map_scatter3_frames");
}

extern "C" int ds_map_gather_taps(const void* pano, void* out, const int32_t* idx, const float* wgt, int ntaps, int C, int F, int f0,
                                  int tf, int HW, int P, int dtype, void* stream) {
    DS_CHECK_ARG(pano && out && idx && wgt, "ds_map_gather_taps: null argument");
    DS_CHECK_ARG(ntaps > 0 && C > 0 && F > 0 && tf > 0 && f0 >= 0 && HW > 0 && P > 0, "ds_map_gather_taps: sizes must be positive");
    hipStream_t st = (hipStream_t)stream;
    const long work = (long)C * tf * P;
    if (dtype == DS_F16) map_gather_taps_kernel<f16><<<grid_for(work), 256, 0, st>>>((const f16*)pano, (f16*)out, idx, wgt, ntaps, C, F, f0, tf, HW, P);
    else if (dtype == DS_F32) map_gather_taps_kernel<float><<<grid_for(work), 256, 0, st>>>((const float*)pano, (float*)out, idx, wgt, ntaps, C, F, f0, tf, HW, P);
    else DS_CHECK_ARG(false, "ds_map_gather_taps: bad dtype %d", dtype);
    DS_CHECK_LAUNCH("ds_map_gather_taps");
    return DS_OK;
}

extern "C" int ds_map_splat(void* pano, const void* view, const int32_t* tgt, const int32_t* row_ptr, const int32_t* src,
                            const float* wgt, int CF, int HW, int P, int ntgt, int dtype, void* stream) {
    DS_CHECK_ARG(pano && view && tgt && row_ptr && src && wgt, "ds_map_splat: null argument");
    DS_CHECK_ARG(CF > 0 && HW > 0 && P > 0 && ntgt > 0, "ds_map_splat: sizes must be positive");
    hipStream_t st = (hipStream_t)stream;
    const long work = (long)CF * ntgt;
    if (dtype == DS_F16) map_splat_kernel<f16><<<grid_for(work), 256, 0, st>>>((f16*)pano, (const f16*)view, tgt, row_ptr, src, wgt, CF, HW, P, ntgt);
    else if (dtype == DS_F32) map_splat_kernel<float><<<grid_for(work), 256, 0, st>>>((float*)pano, (const float*)view, tgt, row_ptr, src, wgt, CF, HW, P, ntgt);
    else DS_CHECK_ARG(false, "ds_map_splat: bad dtype %d", dtype);
    DS_CHECK_LAUNCH("ds_map_splat");
    return DS_OK;
}
