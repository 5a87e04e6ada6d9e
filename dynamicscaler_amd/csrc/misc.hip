// Small HBM-bound helpers around the UNet body (layout changes, channel concat, time embedding, SiLU).
#include "common.h"

namespace {

__global__ void __launch_bounds__(256)
concat_kernel(const f16* __restrict__ a, const f16* __restrict__ b, f16* __restrict__ dst, long rows, int c1, int c2) {
    const int v1 = c1 / 8, v2 = c2 / 8, vt = v1 + v2;
    const long total = rows * vt;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long r = idx / vt;
        const int c = (int)(idx - r * vt);
        f16x8 val;
        if (c < v1) val = *reinterpret_cast<const f16x8*>(a + r * c1 + c * 8);
        else val = *reinterpret_cast<const f16x8*>(b + r * c2 + (c - v1) * 8);
        *reinterpret_cast<f16x8*>(dst + r * (c1 + c2) + c * 8) = val;
    }
}

// y[r][0:C] = fp16(x[r][0:C]) for fp32 rows with strides ldx / ldy (8 channels per thread: two 16-byte loads, one 16-byte store)
__global__ void __launch_bounds__(256)
cast_rows_kernel(const float* __restrict__ x, f16* __restrict__ y, long rows, int C, int ldx, int ldy) {
    const int nv = C / 8;
    const long total = rows * nv;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long r = idx / nv;
        const int c = (int)(idx - r * nv) * 8;
        const f32x4 a = DS_SLOAD(reinterpret_cast<const f32x4*>(x + r * ldx + c));
        const f32x4 b = DS_SLOAD(reinterpret_cast<const f32x4*>(x + r * ldx + c + 4));
        const f16x8 o = {(f16)a[0], (f16)a[1], (f16)a[2], (f16)a[3], (f16)b[0], (f16)b[1], (f16)b[2], (f16)b[3]};
        *reinterpret_cast<f16x8*>(y + r * ldy + c) = o;
    }
}

// patches[m][(ky*3+kx)*C + c] = x[b][c][t][y+ky-1][x+kx-1] (zero outside), m = ((b*T+t)*H+y)*W+x; columns >= 9C are 0
template <typename T>
__global__ void __launch_bounds__(256)
im2col_in_kernel(const T* __restrict__ x, f16* __restrict__ patches, int B, int C, int Tn, int H, int W, int kpad) {
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
            if (iy >= 0 && iy < H && ix >= 0 && ix < W)
                v = (float)x[((((long)b * C + c) * Tn + t) * H + iy) * W + ix];
        }
        patches[idx] = (f16)v;
    }
}

// The same with a CxC channel mix in front (AutoencoderKL.decode: conv_in(post_quant_conv(z / scale_factor)),
// autoencoder.py:103-107, ddpm3d.py:559): inside the image the patch value is sum_ci wmat[c][ci] * x[ci] * in_scale + bvec[c],
// outside it is conv_in's zero padding.  C <= 8.
template <typename T>
__global__ void __launch_bounds__(256)
im2col_in_affine_kernel(const T* __restrict__ x, f16* __restrict__ patches, int B, int C, int Tn, int H, int W, int kpad,
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
        patches[idx] = (f16)v;
    }
}

// Row softmax: p[r][c] = softmax_c(s[r][c] * scale), fp32 in, fp16 out; one wave per row (AttnBlock of the first-stage
// decoder, ae_modules.py:62-64: the 512-wide single head does not fit the head_dim-64 flash kernel, so its scores go
// through memory as fp32).
__global__ void __launch_bounds__(256)
softmax_rows_kernel(const float* __restrict__ s, f16* __restrict__ p, int rows, int cols, int lds, int ldp, float scale_log2) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long row = (long)blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const float* sr = s + row * lds;
    f16* pr = p + row * ldp;
    float m = -1e30f;
    for (int c = lane * 4; c < cols; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(sr + c);
        m = fmaxf(m, fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
    }
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) m = fmaxf(m, __shfl_xor(m, sh));
    const float mneg = -m * scale_log2;
    float l = 0.0f;
    for (int c = lane * 4; c < cols; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(sr + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) l += __builtin_amdgcn_exp2f(fmaf(v[j], scale_log2, mneg));
    }
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) l += __shfl_xor(l, sh);
    const float inv = 1.0f / l;
    for (int c = lane * 4; c < cols; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(sr + c);
        f16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (f16)(__builtin_amdgcn_exp2f(fmaf(v[j], scale_log2, mneg)) * inv);
        *reinterpret_cast<f16x4*>(pr + c) = o;
    }
}

// DiagonalGaussianDistribution.sample() * scale_factor (lvdm/distributions.py:24-40, ddpm3d.py:458-465): moments rows
// [m][2C] fp32 (mean | logvar), row m = ((b*T + t)*H + y)*W + x -> out [B][C][T][H][W] fp32;
// noise (same layout as out) may be NULL (= the mode).
__global__ void __launch_bounds__(256)
posterior_sample_kernel(const float* __restrict__ mom, int ldm, const float* __restrict__ noise, float* __restrict__ out,
                        int B, int C, int Tn, int H, int W, float scale) {
    const long total = (long)B * C * Tn * H * W;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        long r = idx;
        const int x = (int)(r % W); r /= W;
        const int y = (int)(r % H); r /= H;
        const int t = (int)(r % Tn); r /= Tn;
        const int c = (int)(r % C);
        const int b = (int)(r / C);
        const long m = (((long)b * Tn + t) * H + y) * W + x;
        const float mean = mom[m * ldm + c];
        const float logvar = fminf(fmaxf(mom[m * ldm + C + c], -30.0f), 20.0f);
        const float stdv = expf(0.5f * logvar);
        out[idx] = scale * (noise ? mean + stdv * noise[idx] : mean);
    }
}

template <typename Y, typename O>
__global__ void __launch_bounds__(256)
rows_to_ncthw_kernel(const Y* __restrict__ y, int ldy, O* __restrict__ out, int B, int C, int Tn, int H, int W) {
    const long total = (long)B * C * Tn * H * W;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        long r = idx;
        const int xx = (int)(r % W); r /= W;
        const int yy = (int)(r % H); r /= H;
        const int t = (int)(r % Tn); r /= Tn;
        const int c = (int)(r % C);
        const int b = (int)(r / C);
        const long m = (((long)b * Tn + t) * H + yy) * W + xx;
        out[idx] = (O)(float)y[m * ldy + c];
    }
}

// utils_diffusion.py:8-28: freqs = exp(-ln(10000) * i / half); emb = [cos(t*f) | sin(t*f)]
__global__ void timestep_embedding_kernel(const int64_t* __restrict__ t, f16* __restrict__ out, int n, int dim) {
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
    out[idx] = (f16)v;
}

__global__ void __launch_bounds__(256) silu_kernel(const f16* __restrict__ x, f16* __restrict__ y, size_t n) {
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
        const float f = (float)x[idx];
        y[idx] = (f16)(f / (1.0f + __expf(-f)));
    }
}


// resize_video_latent (utils/diffusion_utils.py:21-33): F.interpolate over the last two dims of [B,C,F,H,W], per frame.
//   mode 0 = 'nearest'  : src = min(floor(dst * (in/out as float)), in-1)                     (exact copy semantics)
//   mode 1 = 'bicubic'  : align_corners=False, A = -0.75, border-clamped taps (ATen upsample_bicubic2d)
__device__ __forceinline__ float cubic1(float x, float A) { return ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f; }
__device__ __forceinline__ float cubic2(float x, float A) { return ((A * x - 5.0f * A) * x + 8.0f * A) * x - 4.0f * A; }

template <typename T>
__global__ void __launch_bounds__(256)
resize_kernel(const T* __restrict__ in, T* __restrict__ out, long planes, int hin, int win, int hout, int wout, int mode,
              float scale_h, float scale_w) {
    const long total = planes * hout * wout;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % wout);
        const long r = idx / wout;
        const int y = (int)(r % hout);
        const long pl = r / hout;
        const T* src = in + pl * hin * win;
        if (mode == 0) {
            const int sy = min((int)floorf((float)y * scale_h), hin - 1);
            const int sx = min((int)floorf((float)x * scale_w), win - 1);
            out[idx] = src[(long)sy * win + sx];
        } else {
            const float A = -0.75f;
            const float fy = scale_h * ((float)y + 0.5f) - 0.5f, fx = scale_w * ((float)x + 0.5f) - 0.5f;
            const int iy = (int)floorf(fy), ix = (int)floorf(fx);
            const float ty = fy - (float)iy, tx = fx - (float)ix;
            const float wy[4] = {cubic2(ty + 1.0f, A), cubic1(ty, A), cubic1(1.0f - ty, A), cubic2(2.0f - ty, A)};
            const float wx[4] = {cubic2(tx + 1.0f, A), cubic1(tx, A), cubic1(1.0f - tx, A), cubic2(2.0f - tx, A)};
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int yy = min(max(iy - 1 + j, 0), hin - 1);
                float row = 0.0f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int xx = min(max(ix - 1 + i, 0), win - 1);
                    row += (float)src[(long)yy * win + xx] * wx[i];
                }
                acc += row * wy[j];
            }
            out[idx] = (T)acc;
        }
    }
}

inline int grid_for(long work) {
    long b = (work + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

}  // namespace

extern "C" int ds_concat_channels(const void* a, const void* b, void* dst, int rows, int c1, int c2, void* stream) {
    DS_CHECK_ARG(a && b && dst, "ds_concat_channels: null argument");
    DS_CHECK_ARG(rows > 0 && c1 > 0 && c2 > 0 && c1 % 8 == 0 && c2 % 8 == 0, "ds_concat_channels: rows=%d c1=%d c2=%d (channels must be multiples of 8)", rows, c1, c2);
    const long work = (long)rows * ((c1 + c2) / 8);
    concat_kernel<<<grid_for(work), 256, 0, (hipStream_t)stream>>>((const f16*)a, (const f16*)b, (f16*)dst, rows, c1, c2);
    DS_CHECK_LAUNCH("ds_concat_channels");
    return DS_OK;
}

extern "C" int ds_cast_rows_f32_f16(const float* x, int ldx, void* y, int ldy, long rows, int C, void* stream) {
    DS_CHECK_ARG(x && y, "ds_cast_rows_f32_f16: null argument");
    DS_CHECK_ARG(rows > 0 && C > 0 && C % 8 == 0 && ldx >= C && ldx % 4 == 0 && ldy >= C && ldy % 8 == 0,
                 "ds_cast_rows_f32_f16: rows=%ld C=%d ldx=%d ldy=%d (C %% 8, ldx %% 4, ldy %% 8)", rows, C, ldx, ldy);
    DS_CHECK_ARG((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0, "ds_cast_rows_f32_f16: 16-byte aligned pointers");
    cast_rows_kernel<<<grid_for(rows * (C / 8)), 256, 0, (hipStream_t)stream>>>(x, (f16*)y, rows, C, ldx, ldy);
    DS_CHECK_LAUNCH("ds_cast_rows_f32_f16");
    return DS_OK;
}

extern "C" int ds_im2col_in(const void* x, int x_dtype, void* patches, int B, int C, int T, int H, int W, int kpad,
                            void* stream) {
    DS_CHECK_ARG(x && patches, "ds_im2col_in: null argument");
    DS_CHECK_ARG(B > 0 && C > 0 && T > 0 && H > 0 && W > 0, "ds_im2col_in: sizes must be positive");
    DS_CHECK_ARG(kpad >= 9 * C && kpad % 64 == 0, "ds_im2col_in: kpad=%d must be >= 9*C and a multiple of 64", kpad);
    const long work = (long)B * T * H * W * kpad;
    if (x_dtype == DS_F16)
        im2col_in_kernel<f16><<<grid_for(work), 256, 0, (hipStream_t)stream>>>((const f16*)x, (f16*)patches, B, C, T, H, W, kpad);
    else if (x_dtype == DS_F32)
        im2col_in_kernel<float><<<grid_for(work), 256, 0, (hipStream_t)stream>>>((const float*)x, (f16*)patches, B, C, T, H, W, kpad);
    else
        DS_CHECK_ARG(false, "ds_im2col_in: bad dtype %d", x_dtype);
    DS_CHECK_LAUNCH("ds_im2col_in");
    return DS_OK;
}

extern "C" int ds_im2col_in_affine(const void* x, int x_dtype, void* patches, int B, int C, int T, int H, int W, int kpad,
                                   const float* wmat, const float* bvec, float in_scale, void* stream) {
    DS_CHECK_ARG(x && patches && wmat && bvec, "ds_im2col_in_affine: null argument");
    DS_CHECK_ARG(B > 0 && C > 0 && C <= 8 && T > 0 && H > 0 && W > 0, "ds_im2col_in_affine: sizes must be positive, C <= 8");
    DS_CHECK_ARG(kpad >= 9 * C && kpad % 64 == 0, "ds_im2col_in_affine: kpad=%d must be >= 9*C and a multiple of 64", kpad);
    const long work = (long)B * T * H * W * kpad;
    if (x_dtype == DS_F16)
        im2col_in_affine_kernel<f16><<<grid_for(work), 256, 0, (hipStream_t)stream>>>((const f16*)x, (f16*)patches, B, C, T, H, W, kpad, wmat, bvec, in_scale);
    else if (x_dtype == DS_F32)
        im2col_in_affine_kernel<float><<<grid_for(work), 256, 0, (hipStream_t)stream>>>((const float*)x, (f16*)patches, B, C, T, H, W, kpad, wmat, bvec, in_scale);
    else
        DS_CHECK_ARG(false, "ds_im2col_in_affine: bad dtype %d", x_dtype);
    DS_CHECK_LAUNCH("ds_im2col_in_affine");
    return DS_OK;
}

extern "C" int ds_softmax_rows(const float* s, void* p, int rows, int cols, int lds, int ldp, float scale, void* stream) {
    DS_CHECK_ARG(s && p, "ds_softmax_rows: null argument");
    DS_CHECK_ARG(rows > 0 && cols > 0 && cols % 4 == 0 && lds % 4 == 0 && ldp % 4 == 0 && lds >= cols && ldp >= cols,
                 "ds_softmax_rows: cols / strides must be positive multiples of 4");
    softmax_rows_kernel<<<(rows + 3) / 4, 256, 0, (hipStream_t)stream>>>(s, (f16*)p, rows, cols, lds, ldp, scale * 1.4426950408889634f);
    DS_CHECK_LAUNCH("ds_softmax_rows");
    return DS_OK;
}

extern "C" int ds_posterior_sample(const float* moments, int ldm, const float* noise, float* out, int B, int C, int T, int H,
                                   int W, float scale, void* stream) {
    DS_CHECK_ARG(moments && out, "ds_posterior_sample: null argument");
    DS_CHECK_ARG(B > 0 && C > 0 && T > 0 && H > 0 && W > 0 && ldm >= 2 * C, "ds_posterior_sample: bad sizes");
    const long work = (long)B * C * T * H * W;
    posterior_sample_kernel<<<grid_for(work), 256, 0, (hipStream_t)stream>>>(moments, ldm, noise, out, B, C, T, H, W, scale);
    DS_CHECK_LAUNCH("ds_posterior_sample");
    return DS_OK;
}

extern "C" int ds_rows_to_ncthw(const void* y, int y_dtype, int ldy, void* out, int out_dtype, int B, int C, int T,
                                int H, int W, void* stream) {
    DS_CHECK_ARG(y && out, "ds_rows_to_ncthw: null argument");
    DS_CHECK_ARG(B > 0 && C > 0 && T > 0 && H > 0 && W > 0 && ldy >= C, "ds_rows_to_ncthw: bad sizes");
    const long work = (long)B * C * T * H * W;
    hipStream_t st = (hipStream_t)stream;
    if (y_dtype == DS_F32 && out_dtype == DS_F32)
        rows_to_ncthw_kernel<float, float><<<grid_for(work), 256, 0, st>>>((const float*)y, ldy, (float*)out, B, C, T, H, W);
    else if (y_dtype == DS_F32 && out_dtype == DS_F16)
        rows_to_ncthw_kernel<float, f16><<<grid_for(work), 256, 0, st>>>((const float*)y, ldy, (f16*)out, B, C, T, H, W);
    else if (y_dtype == DS_F16 && out_dtype == DS_F32)
        rows_to_ncthw_kernel<f16, float><<<grid_for(work), 256, 0, st>>>((const f16*)y, ldy, (float*)out, B, C, T, H, W);
    else if (y_dtype == DS_F16 && out_dtype == DS_F16)
        rows_to_ncthw_kernel<f16, f16><<<grid_for(work), 256, 0, st>>>((const f16*)y, ldy, (f16*)out, B, C, T, H, W);
    else
        DS_CHECK_ARG(false, "ds_rows_to_ncthw: bad dtypes %d %d", y_dtype, out_dtype);
    DS_CHECK_LAUNCH("ds_rows_to_ncthw");
    return DS_OK;
}

extern "C" int ds_timestep_embedding(const int64_t* t, void* out, int n, int dim, void* stream) {
    DS_CHECK_ARG(t && out && n > 0 && dim > 0, "ds_timestep_embedding: bad argument");
    timestep_embedding_kernel<<<(n * dim + 255) / 256, 256, 0, (hipStream_t)stream>>>(t, (f16*)out, n, dim);
    DS_CHECK_LAUNCH("ds_timestep_embedding");
    return DS_OK;
}

extern "C" int ds_silu_f16(const void* x, void* y, size_t n, void* stream) {
    DS_CHECK_ARG(x && y && n > 0, "ds_silu_f16: bad argument");
    silu_kernel<<<grid_for((long)n), 256, 0, (hipStream_t)stream>>>((const f16*)x, (f16*)y, n);
    DS_CHECK_LAUNCH("ds_silu_f16");
    return DS_OK;
}

extern "C" int ds_resize_latent(const void* in, void* out, int dtype, long planes, int hin, int win, int hout, int wout,
                                int mode, void* stream) {
    DS_CHECK_ARG(in && out, "ds_resize_latent: null argument");
    DS_CHECK_ARG(planes > 0 && hin > 0 && win > 0 && hout > 0 && wout > 0, "ds_resize_latent: sizes must be positive");
    DS_CHECK_ARG(mode == 0 || mode == 1, "ds_resize_latent: mode must be 0 (nearest) or 1 (bicubic)");
    const float sh = (float)hin / (float)hout, sw = (float)win / (float)wout;
    const long work = planes * hout * wout;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DS_F16) resize_kernel<f16><<<grid_for(work), 256, 0, st>>>((const f16*)in, (f16*)out, planes, hin, win, hout, wout, mode, sh, sw);
    else if (dtype == DS_F32) resize_kernel<float><<<grid_for(work), 256, 0, st>>>((const float*)in, (float*)out, planes, hin, win, hout, wout, mode, sh, sw);
    else DS_CHECK_ARG(false, "ds_resize_latent: bad dtype %d", dtype);
    DS_CHECK_LAUNCH("ds_resize_latent");
    return DS_OK;
}
