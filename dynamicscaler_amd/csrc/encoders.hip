// Conditioning producers: the kernels the OpenCLIP ViT-H/14 text / image towers and the IP-Adapter
// Resampler need beyond ds_gemm_f16 / ds_layernorm (lvdm/modules/encoders/condition.py:174-235, 298-365,
// lvdm/modules/encoders/ip_resampler.py).  These run once per distinct prompt / image crop (the pipelines cache the
// embeddings), on 77 / 257 / 273 tokens: none of them is on the per-step hot path, so they are written for
// correctness and reasonable efficiency, not tuned.
//
//   ds_attention_enc_f16   softmax(q k^T * scale [+ causal mask]) v for head_dim 64 (text tower, Resampler) and 80
//                          (image tower), any nk; flash-style, one wave per 32 queries, K fragments straight from
//                          global memory, V staged per wave through LDS (the MFMA needs the key index contiguous in a
//                          lane's registers, V has the head dimension contiguous).
//   ds_gelu_f16            exact-erf GELU (nn.GELU) in place or out of place
//   ds_embed_tokens        token_embedding(tokens) + positional_embedding          (condition.py:217-218)
//   ds_vit_assemble        [class_embedding | patch embeddings] + positional_embedding (condition.py:346-350)
//   ds_clip_preprocess     kornia.geometry.resize(224, bicubic, align_corners=True, antialias) + (x+1)/2 + normalise
//                          (condition.py:324-332)
//   ds_patchify            [b,C,S,S] -> rows (b, gy, gx) x (c, py, px) fp16, K padded: conv1 as a GEMM (:342)
#include "common.h"

namespace {

template <int HDIM>
__global__ void __launch_bounds__(256)
enc_attention_kernel(const f16* __restrict__ q, const f16* __restrict__ k, const f16* __restrict__ v, f16* __restrict__ out,
                     int heads, int nq, int nk, int ldq, int ldk, int ldv, int ldo, float scale_log2, int causal,
                     int q_tiles) {
    constexpr int KS = HDIM / 16;            // k-steps of the score MFMA
    constexpr int DB = (HDIM + 31) / 32;     // 32-wide blocks of the head dimension
    constexpr int VS = DB * 32 + 8;          // halfs per staged V row
    constexpr int CH = HDIM / 8;             // 16-byte chunks per V row
    __shared__ __attribute__((aligned(16))) f16 sV[4][32 * VS];
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 31, fh = lane >> 5;
    const int qt = blockIdx.x % q_tiles;
    const int bh = blockIdx.x / q_tiles;
    const int head = bh % heads, b = bh / heads;
    const int q_base = qt * 128 + wave * 32;
    if (q_base >= nq) return;                // waves are independent: no workgroup barrier below
    f16* sv = sV[wave];
    for (int i = lane; i < 32 * VS / 2; i += 64) reinterpret_cast<unsigned*>(sv)[i] = 0u;   // pad columns stay zero

    const int qi = q_base + fr;
    const f16* qp = q + ((long)b * nq + min(qi, nq - 1)) * ldq + head * HDIM;
    const f16* kp = k + (long)b * nk * ldk + head * HDIM;
    const f16* vp = v + (long)b * nk * ldv + head * HDIM;
    f16x8 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const f16x8*>(qp + ks * 16 + fh * 8);

    f32x16 o[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int j = 0; j < 16; ++j) o[db][j] = 0.0f;
    float mrun = -1e30f, lrun = 0.0f;

    int nblk = (nk + 31) / 32;
    if (causal) nblk = min(nblk, min(q_base + 31, nq - 1) / 32 + 1);
    for (int kb = 0; kb < nblk; ++kb) {
        // S^T = K Q^T: lane = key row fr of the block (A operand), 8 consecutive d per k-step half
        const f16* krow = kp + (long)min(kb * 32 + fr, nk - 1) * ldk;
        f32x16 s;
#pragma unroll
        for (int j = 0; j < 16; ++j) s[j] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const f16x8 kf = *reinterpret_cast<const f16x8*>(krow + ks * 16 + fh * 8);
            s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], s, 0, 0, 0);
        }
        // stage the block's V rows (row index clamped: their probabilities are exactly 0)
        __builtin_amdgcn_wave_barrier();     // every lane is done with the previous block's strip
#pragma unroll
        for (int i = 0; i < (32 * CH + 63) / 64; ++i) {
            const int c = lane + 64 * i;
            if (c < 32 * CH) {
                const int row = c / CH, ch = c - row * CH;
                *reinterpret_cast<u32x4*>(sv + row * VS + ch * 8) =
                    *reinterpret_cast<const u32x4*>(vp + (long)min(kb * 32 + row, nk - 1) * ldv + ch * 8);
            }
        }
        // accumulator j of lane (fr, fh): query fr, key (j&3) + 8*(j>>2) + 4*fh of the block
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int key = kb * 32 + (j & 3) + 8 * (j >> 2) + 4 * fh;
            if (key >= nk || (causal && key > qi)) s[j] = -1e30f;
        }
        float mt = -1e30f;
#pragma unroll
        for (int j = 0; j < 16; ++j) mt = fmaxf(mt, s[j]);
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        const float mnew = fmaxf(mrun, mt);
        const float alpha = __builtin_amdgcn_exp2f((mrun - mnew) * scale_log2);
        mrun = mnew;
        const float mneg = -mnew * scale_log2;
        float lsum = 0.0f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            s[j] = __builtin_amdgcn_exp2f(fmaf(s[j], scale_log2, mneg));
            lsum += s[j];
        }
        lrun = lrun * alpha + lsum;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int j = 0; j < 16; ++j) o[db][j] *= alpha;

        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this wave's strip is written (LDS ops of a wave are ordered)
        __builtin_amdgcn_wave_barrier();
        // O^T[d][q] += V^T[d][key] P^T[key][q]: the S^T accumulators, rounded to fp16, are the B operand
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
            f16x8 pf;
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[j] = (f16)s[8 * ss + j];
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                f16x8 vf;
#pragma unroll
                for (int j = 0; j < 8; ++j) vf[j] = sv[(ss * 16 + 4 * fh + (j & 3) + 8 * (j >> 2)) * VS + db * 32 + fr];
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, o[db], 0, 0, 0);
            }
        }
    }

    const float l = lrun + __shfl_xor(lrun, 32);
    const float inv = 1.0f / l;
    if (qi >= nq) return;
    f16* op = out + ((long)b * nq + qi) * ldo + head * HDIM;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int dcol = db * 32 + 8 * g + 4 * fh;
            if (dcol < HDIM) {
                f16x4 w;
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = (f16)(o[db][4 * g + j] * inv);
                *reinterpret_cast<f16x4*>(op + dcol) = w;
            }
        }
}

__global__ void __launch_bounds__(256) gelu_kernel(const f16* __restrict__ x, f16* __restrict__ y, size_t n) {
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
        const float f = (float)x[idx];
        y[idx] = (f16)(0.5f * f * (1.0f + erff(f * 0.70710678118654752f)));
    }
}

__global__ void __launch_bounds__(256)
embed_tokens_kernel(const int* __restrict__ tokens, const f16* __restrict__ table, const float* __restrict__ pos,
                    f16* __restrict__ out, long total, int ctx, int width, int vocab) {
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % width);
        const long row = idx / width;
        const int tok = min(max(tokens[row], 0), vocab - 1);
        out[idx] = (f16)((float)table[(long)tok * width + c] + pos[(row % ctx) * width + c]);
    }
}

__global__ void __launch_bounds__(256)
vit_assemble_kernel(const f16* __restrict__ patches, const float* __restrict__ cls, const float* __restrict__ pos,
                    f16* __restrict__ out, long total, int grid2, int width) {
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % width);
        const long row = idx / width;
        const int tok = (int)(row % (grid2 + 1));
        const long b = row / (grid2 + 1);
        const float e = tok == 0 ? cls[c] : (float)patches[(b * grid2 + tok - 1) * width + c];
        out[idx] = (f16)(e + pos[(long)tok * width + c]);
    }
}

// kornia.geometry.transform.resize(x, (S,S), 'bicubic', align_corners=True, antialias=True): when an axis shrinks
// (factor = in/out > 1) the image is first blurred with a separable Gaussian, sigma = max((factor-1)/2, 0.001), kernel
// size = max(int(4*sigma), 3) made odd, 'reflect' border; then F.interpolate bicubic (A = -0.75, border-clamped taps,
// src = dst * (in-1)/(out-1)).  Then (x+1)/2 and the CLIP mean/std.  The blur is evaluated on the fly at the 16 taps.
__device__ __forceinline__ float cubic1(float x, float A) { return ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f; }
__device__ __forceinline__ float cubic2(float x, float A) { return ((A * x - 5.0f * A) * x + 8.0f * A) * x - 4.0f * A; }
__device__ __forceinline__ int reflect_idx(int i, int n) {
    if (n == 1) return 0;
    const int p = 2 * (n - 1);
    i %= p;
    if (i < 0) i += p;
    return i < n ? i : p - i;
}

constexpr int MAX_BLUR = 65;

struct BlurTaps {
    int kh, kw;
    float wh[MAX_BLUR], ww[MAX_BLUR];
};

template <typename T>
__global__ void __launch_bounds__(256)
clip_preprocess_kernel(const T* __restrict__ img, float* __restrict__ out, int nimg, int C, int H, int W, int S,
                       BlurTaps taps, float mean0, float mean1, float mean2, float std0, float std1, float std2) {
    const long total = (long)nimg * C * S * S;
    const float sh = S > 1 ? (float)(H - 1) / (float)(S - 1) : 0.0f;
    const float sw = S > 1 ? (float)(W - 1) / (float)(S - 1) : 0.0f;
    const float A = -0.75f;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % S);
        const long r = idx / S;
        const int y = (int)(r % S);
        const long plane = r / S;
        const int c = (int)(plane % C);
        const T* src = img + plane * H * W;
        const float fy = sh * y, fx = sw * x;
        const int iy = (int)floorf(fy), ix = (int)floorf(fx);
        const float ty = fy - iy, tx = fx - ix;
        const float cy[4] = {cubic2(ty + 1.0f, A), cubic1(ty, A), cubic1(1.0f - ty, A), cubic2(2.0f - ty, A)};
        const float cx[4] = {cubic2(tx + 1.0f, A), cubic1(tx, A), cubic1(1.0f - tx, A), cubic2(2.0f - tx, A)};
        float acc = 0.0f;
        for (int a = 0; a < 4; ++a) {
            const int yy = min(max(iy - 1 + a, 0), H - 1);
            float rowv = 0.0f;
            for (int bb = 0; bb < 4; ++bb) {
                const int xx = min(max(ix - 1 + bb, 0), W - 1);
                // blurred sample at (yy, xx): horizontal pass inside the vertical pass (separable, reflect border)
                float sv = 0.0f;
                for (int u = 0; u < taps.kh; ++u) {
                    const int ry = reflect_idx(yy + u - taps.kh / 2, H);
                    float hv = 0.0f;
                    for (int w = 0; w < taps.kw; ++w) hv += taps.ww[w] * (float)src[(long)ry * W + reflect_idx(xx + w - taps.kw / 2, W)];
                    sv += taps.wh[u] * hv;
                }
                rowv += cx[bb] * sv;
            }
            acc += cy[a] * rowv;
        }
        const float m = c == 0 ? mean0 : (c == 1 ? mean1 : mean2);
        const float s = c == 0 ? std0 : (c == 1 ? std1 : std2);
        out[idx] = ((acc + 1.0f) * 0.5f - m) / s;
    }
}

__global__ void __launch_bounds__(256)
patchify_kernel(const float* __restrict__ img, f16* __restrict__ rows, int nimg, int C, int S, int P, int kpad) {
    const int g = S / P;
    const long total = (long)nimg * g * g * kpad;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int kcol = (int)(idx % kpad);
        const long row = idx / kpad;
        float val = 0.0f;
        if (kcol < C * P * P) {
            const int c = kcol / (P * P), rem = kcol - c * P * P, py = rem / P, px = rem - py * P;
            const int gx = (int)(row % g), gy = (int)((row / g) % g);
            const long b = row / ((long)g * g);
            val = img[((b * C + c) * S + gy * P + py) * (long)S + gx * P + px];
        }
        rows[idx] = (f16)val;
    }
}

inline int grid_for(long work) {
    long b = (work + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 65535 * 16 ? 65535 * 16 : b));
}

}  // namespace

extern "C" int ds_attention_enc_f16(const void* q, const void* k, const void* v, void* out, int batch, int heads, int nq,
                                    int nk, int ldq, int ldk, int ldv, int ldo, int head_dim, float scale, int causal,
                                    void* stream) {
    DS_CHECK_ARG(q && k && v && out, "ds_attention_enc_f16: null argument");
    DS_CHECK_ARG(batch > 0 && heads > 0 && nq > 0 && nk > 0, "ds_attention_enc_f16: batch/heads/nq/nk must be positive");
    DS_CHECK_ARG(head_dim == 64 || head_dim == 80, "ds_attention_enc_f16: head_dim=%d (supported: 64, 80)", head_dim);
    DS_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 4 == 0, "ds_attention_enc_f16: row strides must be multiples of 8 (ldo: 4)");
    DS_CHECK_ARG(ldq >= heads * head_dim && ldk >= heads * head_dim && ldv >= heads * head_dim && ldo >= heads * head_dim,
                 "ds_attention_enc_f16: row stride < heads*head_dim");
    DS_CHECK_ARG(!causal || nq == nk, "ds_attention_enc_f16: the causal mask needs nq == nk");
    hipStream_t st = (hipStream_t)stream;
    const float scale_log2 = scale * 1.4426950408889634f;
    const int q_tiles = ds_cdiv(nq, 128);
    const long grid = (long)batch * heads * q_tiles;
    if (head_dim == 64)
        enc_attention_kernel<64><<<grid, 256, 0, st>>>((const f16*)q, (const f16*)k, (const f16*)v, (f16*)out, heads, nq, nk,
                                                      ldq, ldk, ldv, ldo, scale_log2, causal, q_tiles);
    else
        enc_attention_kernel<80><<<grid, 256, 0, st>>>((const f16*)q, (const f16*)k, (const f16*)v, (f16*)out, heads, nq, nk,
                                                      ldq, ldk, ldv, ldo, scale_log2, causal, q_tiles);
    DS_CHECK_LAUNCH("ds_attention_enc_f16");
    return DS_OK;
}

extern "C" int ds_gelu_f16(const void* x, void* y, size_t n, void* stream) {
    DS_CHECK_ARG(x && y && n > 0, "ds_gelu_f16: bad argument");
    gelu_kernel<<<grid_for((long)n), 256, 0, (hipStream_t)stream>>>((const f16*)x, (f16*)y, n);
    DS_CHECK_LAUNCH("ds_gelu_f16");
    return DS_OK;
}

extern "C" int ds_embed_tokens(const int32_t* tokens, const void* table, const float* pos, void* out, int ntok, int ctx,
                               int width, int vocab, void* stream) {
    DS_CHECK_ARG(tokens && table && pos && out, "ds_embed_tokens: null argument");
    DS_CHECK_ARG(ntok > 0 && ctx > 0 && width > 0 && vocab > 0 && ntok % ctx == 0, "ds_embed_tokens: ntok must be a positive multiple of ctx");
    const long total = (long)ntok * width;
    embed_tokens_kernel<<<grid_for(total), 256, 0, (hipStream_t)stream>>>((const int*)tokens, (const f16*)table, pos, (f16*)out, total, ctx, width, vocab);
    DS_CHECK_LAUNCH("ds_embed_tokens");
    return DS_OK;
}

extern "C" int ds_vit_assemble(const void* patches, const float* cls, const float* pos, void* out, int nimg, int grid2,
                               int width, void* stream) {
    DS_CHECK_ARG(patches && cls && pos && out, "ds_vit_assemble: null argument");
    DS_CHECK_ARG(nimg > 0 && grid2 > 0 && width > 0, "ds_vit_assemble: sizes must be positive");
    const long total = (long)nimg * (grid2 + 1) * width;
    vit_assemble_kernel<<<grid_for(total), 256, 0, (hipStream_t)stream>>>((const f16*)patches, cls, pos, (f16*)out, total, grid2, width);
    DS_CHECK_LAUNCH("ds_vit_assemble");
    return DS_OK;
}

static int blur_taps(int in, int out, int* ksize, float* w) {
    // kornia resize(): sigma = max((in/out - 1)/2, 0.001), kernel size = max(int(4 sigma), 3) made odd;
    // get_gaussian_kernel1d: x = arange(ks) - ks//2, exp(-x^2 / (2 sigma^2)) normalised to sum 1
    float sigma = ((float)in / (float)out - 1.0f) / 2.0f;
    if (sigma < 0.001f) sigma = 0.001f;
    const float kf = 2.0f * 2.0f * sigma;
    int ks = (int)(kf > 3.0f ? kf : 3.0f);
    if (ks % 2 == 0) ks += 1;
    if (ks > MAX_BLUR) return -1;
    float sum = 0.0f;
    for (int i = 0; i < ks; ++i) {
        const float x = (float)(i - ks / 2);
        w[i] = expf(-(x * x) / (2.0f * sigma * sigma));
        sum += w[i];
    }
    for (int i = 0; i < ks; ++i) w[i] /= sum;
    *ksize = ks;
    return 0;
}

extern "C" int ds_clip_preprocess(const void* img, int dtype, float* out, int nimg, int C, int H, int W, int S,
                                  int antialias, const float* mean, const float* stdv, void* stream) {
    DS_CHECK_ARG(img && out && mean && stdv, "ds_clip_preprocess: null argument");
    DS_CHECK_ARG(nimg > 0 && C == 3 && H > 0 && W > 0 && S > 1, "ds_clip_preprocess: needs 3-channel images, S > 1");
    BlurTaps taps;
    taps.kh = taps.kw = 1;
    taps.wh[0] = taps.ww[0] = 1.0f;
    // kornia blurs both axes as soon as ONE of them shrinks (the other gets sigma 0.001: a 3-tap delta)
    if (antialias && (H > S || W > S))
        DS_CHECK_ARG(blur_taps(H, S, &taps.kh, taps.wh) == 0 && blur_taps(W, S, &taps.kw, taps.ww) == 0,
                     "ds_clip_preprocess: %dx%d -> %d needs a blur wider than %d taps", H, W, S, MAX_BLUR);
    const long total = (long)nimg * C * S * S;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DS_F32)
        clip_preprocess_kernel<float><<<grid_for(total), 256, 0, st>>>((const float*)img, out, nimg, C, H, W, S, taps, mean[0], mean[1], mean[2], stdv[0], stdv[1], stdv[2]);
    else if (dtype == DS_F16)
        clip_preprocess_kernel<f16><<<grid_for(total), 256, 0, st>>>((const f16*)img, out, nimg, C, H, W, S, taps, mean[0], mean[1], mean[2], stdv[0], stdv[1], stdv[2]);
    else DS_CHECK_ARG(false, "ds_clip_preprocess: bad dtype %d", dtype);
    DS_CHECK_LAUNCH("ds_clip_preprocess");
    return DS_OK;
}

extern "C" int ds_patchify(const float* img, void* rows, int nimg, int C, int S, int P, int kpad, void* stream) {
    DS_CHECK_ARG(img && rows, "ds_patchify: null argument");
    DS_CHECK_ARG(nimg > 0 && C > 0 && S > 0 && P > 0 && S % P == 0 && kpad >= C * P * P, "ds_patchify: bad geometry");
    const long total = (long)nimg * (S / P) * (S / P) * kpad;
    patchify_kernel<<<grid_for(total), 256, 0, (hipStream_t)stream>>>(img, (f16*)rows, nimg, C, S, P, kpad);
    DS_CHECK_LAUNCH("ds_patchify");
    return DS_OK;
}
