// ds_unet_*: the VideoCrafter LVDM 3D-UNet as one C call -- block program, weight packing and launch sequence over the
// kernels of this library (include/dynscaler_hip.h has the contract; lvdm/modules/networks/openaimodel3d.py:312-708 is
// what it replaces).  Host C++ + a few packing kernels; no allocation on the device: packed weights and the forward's
// scratch arena are caller buffers.  dynamicscaler_amd/unet.py holds the same launch program in Python (per-launch
// instrumentation, taps); tests/test_host_cpu.py compares the two launch traces, tests/test_gpu_unet_c.py the bits.
#include <limits.h>
#include <string.h>

#include <functional>
#include <type_traits>
#include <array>
#include <map>
#include <mutex>
#include <new>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "common.h"

namespace {

constexpr int HEAD_DIM = 64;

// ------------------------------------------------------------------------------------------------ packing kernels
enum { L_LINEAR = 0, L_CONV3 = 1, L_TCONV = 2 };

template <typename ST>
__global__ void __launch_bounds__(256)
pack_w16_kernel(f16* __restrict__ dst, long n_rows, int kdst, const ST* __restrict__ src, int ksrc, int layout, int cin,
                const float* __restrict__ gamma, int geglu_inner, f16* __restrict__ dst_lo, float lo_scale) {
    const long total = n_rows * kdst;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long n = idx / kdst;
        const int k = (int)(idx - n * kdst);
        long r = n;
        if (geglu_inner > 0) {   // dst rows in 32-row groups [x_g | gate_g]  (DS_EPI_GEGLU layout)
            const long blk = n / 64;
            const int w = (int)(n - blk * 64);
            r = w < 32 ? blk * 32 + w : (long)geglu_inner + blk * 32 + (w - 32);
        }
        float v = 0.0f;
        if (k < ksrc) {
            long si;
            if (layout == L_CONV3) { const int tap = k / cin, c = k - tap * cin; si = r * ksrc + (long)c * 9 + tap; }
            else if (layout == L_TCONV) { const int tap = k / cin, c = k - tap * cin; si = r * ksrc + (long)c * 3 + tap; }
            else si = r * ksrc + k;
            v = (float)src[si];
            if (gamma) v *= gamma[k];
        }
        const f16 h = (f16)v;
        dst[idx] = h;
        if (dst_lo) dst_lo[idx] = (f16)((v - (float)h) * lo_scale);     // the wide mode's second operand plane (csrc/wide.hip)
    }
}

// cs[n] = sum_k float(w16[n][k]): one wave per row, lanes stride over k, xor-shuffle tree (fixed order)
__global__ void __launch_bounds__(256)
colsum_kernel(const f16* __restrict__ w16, float* __restrict__ cs, long n_rows, int K) {
    const long n = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (n >= n_rows) return;
    float s = 0.0f;
    for (int k = lane; k < K; k += 64) s += (float)w16[n * K + k];
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) s += __shfl_xor(s, sh);
    if (lane == 0) cs[n] = s;
}

// cb[n] = sum_k W[r(n)][k] * beta[k] (+ bias[r(n)]) in fp32 on the UNROUNDED weights; r(n) = the GEGLU row map of the pack kernel
template <typename ST>
__global__ void __launch_bounds__(256)
colbias_kernel(float* __restrict__ cb, long n_rows, const ST* __restrict__ w, int K, const float* __restrict__ beta,
               const ST* __restrict__ bias, int geglu_inner) {
    const long n = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (n >= n_rows) return;
    long r = n;
    if (geglu_inner > 0) {
        const long blk = n / 64;
        const int wi = (int)(n - blk * 64);
        r = wi < 32 ? blk * 32 + wi : (long)geglu_inner + blk * 32 + (wi - 32);
    }
    float s = 0.0f;
    for (int k = lane; k < K; k += 64) s += (float)w[r * K + k] * beta[k];
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) s += __shfl_xor(s, sh);
    if (lane == 0) cb[n] = s + (bias ? (float)bias[r] : 0.0f);
}

// dst[i] = a[ra(i)] (+ b[i]) as fp32; geglu_inner > 0 applies the row map to a
template <typename ST>
__global__ void __launch_bounds__(256)
vec_f32_kernel(float* __restrict__ dst, long n, const ST* __restrict__ a, const ST* __restrict__ b, int geglu_inner) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    long r = i;
    if (geglu_inner > 0) {
        const long blk = i / 64;
        const int wi = (int)(i - blk * 64);
        r = wi < 32 ? blk * 32 + wi : (long)geglu_inner + blk * 32 + (wi - 32);
    }
    dst[i] = (float)a[r] + (b ? (float)b[i] : 0.0f);
}

template <typename ST>
__global__ void __launch_bounds__(256)
cast_f16_kernel(const ST* __restrict__ x, f16* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = (f16)(float)x[i];
}

__global__ void fill_i64_kernel(int64_t* p, int n, int64_t v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

inline int grid_for(long work) {
    long g = (work + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 65536 ? 65536 : g));
}

// ------------------------------------------------------------------------------------------------ the block program
struct Block {
    enum Kind { CONV_IN, RES, ST, TT, DOWN, UP } kind;
    std::string prefix;
    int cin = 0, cout = 0, heads = 0, depth = 1;
    bool tconv = false;
};

struct WeightSpec {
    std::string key;
    int ndim = 0;
    int64_t shape[5] = {0, 0, 0, 0, 0};
    const void* data = nullptr;
    int dtype = -1;
    long numel() const { long n = 1; for (int i = 0; i < ndim; ++i) n *= shape[i]; return n; }
};

struct PackItem {                 // one tensor of the packed buffer
    std::string name;
    size_t off = 0, bytes = 0;
    long rows = 0;                // leading dimension of a matrix ([rows][bytes / rows / 2] fp16), element count of an fp32 vector
    int dtype = DS_F32;
    std::function<int(char*, hipStream_t)> fill;    // writes it (async) at its address
};

}  // namespace

struct ds_unet {
    ds_unet_config cfg;
    bool gn_fused = false;        // GroupNorm statistics from the producing GEMM's epilogue (ds_gemm_f16_stats) where producer and norm are adjacent
    bool strict = false, inner32 = false, fold = false;   // strict: the stream between the blocks is fp32; inner32: inside the transformers too
    bool wide = false;            // residual_f32 = 3: every activation fp32, split-fp16 products (csrc/wide.hip); every fp16 matrix of the packed
                                  // buffer is followed by its lo plane of the same size
    std::vector<std::vector<Block>> inputs, outputs;
    std::vector<Block> middle;
    std::vector<std::pair<int, int>> cat_ch;     // per decoder group: (channels of h, channels of the skip tensor)
    std::vector<WeightSpec> weights;
    std::map<std::string, int> windex;
    std::vector<PackItem> items;
    std::map<std::string, int> pindex;
    size_t packed_total = 0;
    char* packed = nullptr;
    std::map<std::string, int> emb_off;
    int emb_total = 0, kpad_in = 0;
    ds_launch_hook launch_hook = nullptr;     // instrumentation (ds_unet_set_hooks): per-launch callbacks, per-block taps
    ds_block_tap block_tap = nullptr;
    void* hook_user = nullptr;
    std::mutex peak_mu;
    std::map<std::array<int, 6>, size_t> peak_cache;   // (B, T, H, W, ctx_tokens, cfg_pairs) -> workspace peak of the dry run

    const WeightSpec* w(const std::string& key) const {
        auto it = windex.find(key);
        return it == windex.end() ? nullptr : &weights[it->second];
    }
    void* P(const std::string& name) const {       // before ds_unet_pack (dry runs): a non-null placeholder that is never dereferenced
        auto it = pindex.find(name);
        if (it == pindex.end()) return nullptr;
        return (packed ? packed : (char*)256) + items[it->second].off;
    }
    bool hasP(const std::string& name) const { return pindex.count(name) != 0; }
};

namespace {

void add_w(ds_unet* u, const std::string& key, std::initializer_list<int64_t> shape) {
    WeightSpec s;
    s.key = key;
    s.ndim = (int)shape.size();
    int i = 0;
    for (int64_t d : shape) s.shape[i++] = d;
    u->windex[key] = (int)u->weights.size();
    u->weights.push_back(s);
}

void transformer_params(ds_unet* u, const std::string& prefix, int dim_in, int heads, int dim_head, int depth, int context_dim,
                        bool linear_proj, bool conv1d, bool img_attn) {
    const int inner = heads * dim_head;
    add_w(u, prefix + ".norm.weight", {dim_in});
    add_w(u, prefix + ".norm.bias", {dim_in});
    if (linear_proj) add_w(u, prefix + ".proj_in.weight", {inner, dim_in});
    else if (conv1d) add_w(u, prefix + ".proj_in.weight", {inner, dim_in, 1});
    else add_w(u, prefix + ".proj_in.weight", {inner, dim_in, 1, 1});
    add_w(u, prefix + ".proj_in.bias", {inner});
    for (int d = 0; d < depth; ++d) {
        const std::string p = prefix + ".transformer_blocks." + std::to_string(d);
        for (int a = 1; a <= 2; ++a) {
            const std::string n = p + ".attn" + std::to_string(a);
            const int kv_in = (a == 1 || context_dim <= 0) ? inner : context_dim;
            add_w(u, n + ".to_q.weight", {inner, inner});
            add_w(u, n + ".to_k.weight", {inner, kv_in});
            add_w(u, n + ".to_v.weight", {inner, kv_in});
            add_w(u, n + ".to_out.0.weight", {inner, inner});
            add_w(u, n + ".to_out.0.bias", {inner});
            if (a == 2 && img_attn) {
                add_w(u, n + ".to_k_ip.weight", {inner, kv_in});
                add_w(u, n + ".to_v_ip.weight", {inner, kv_in});
            }
        }
        add_w(u, p + ".ff.net.0.proj.weight", {inner * 8, inner});
        add_w(u, p + ".ff.net.0.proj.bias", {inner * 8});
        add_w(u, p + ".ff.net.2.weight", {inner, inner * 4});
        add_w(u, p + ".ff.net.2.bias", {inner});
        for (int n = 1; n <= 3; ++n) {
            add_w(u, p + ".norm" + std::to_string(n) + ".weight", {inner});
            add_w(u, p + ".norm" + std::to_string(n) + ".bias", {inner});
        }
    }
    if (linear_proj) add_w(u, prefix + ".proj_out.weight", {dim_in, inner});
    else if (conv1d) add_w(u, prefix + ".proj_out.weight", {dim_in, inner, 1});
    else add_w(u, prefix + ".proj_out.weight", {dim_in, inner, 1, 1});
    add_w(u, prefix + ".proj_out.bias", {dim_in});
}

void block_params(ds_unet* u, const Block& b) {
    const ds_unet_config& c = u->cfg;
    const int ted = 4 * c.model_channels;
    const std::string& p = b.prefix;
    switch (b.kind) {
        case Block::CONV_IN:
            add_w(u, p + ".weight", {b.cout, b.cin, 3, 3});
            add_w(u, p + ".bias", {b.cout});
            break;
        case Block::RES:
            add_w(u, p + ".in_layers.0.weight", {b.cin});
            add_w(u, p + ".in_layers.0.bias", {b.cin});
            add_w(u, p + ".in_layers.2.weight", {b.cout, b.cin, 3, 3});
            add_w(u, p + ".in_layers.2.bias", {b.cout});
            add_w(u, p + ".emb_layers.1.weight", {b.cout, ted});
            add_w(u, p + ".emb_layers.1.bias", {b.cout});
            add_w(u, p + ".out_layers.0.weight", {b.cout});
            add_w(u, p + ".out_layers.0.bias", {b.cout});
            add_w(u, p + ".out_layers.3.weight", {b.cout, b.cout, 3, 3});
            add_w(u, p + ".out_layers.3.bias", {b.cout});
            if (b.cin != b.cout) {
                add_w(u, p + ".skip_connection.weight", {b.cout, b.cin, 1, 1});
                add_w(u, p + ".skip_connection.bias", {b.cout});
            }
            if (b.tconv)
                for (int i = 1; i <= 4; ++i) {
                    const std::string q = p + ".temopral_conv.conv" + std::to_string(i);   // the reference's spelling (openaimodel3d.py:196)
                    const std::string ci = std::to_string(i == 1 ? 2 : 3);
                    add_w(u, q + ".0.weight", {b.cout});
                    add_w(u, q + ".0.bias", {b.cout});
                    add_w(u, q + "." + ci + ".weight", {b.cout, b.cout, 3, 1, 1});
                    add_w(u, q + "." + ci + ".bias", {b.cout});
                }
            break;
        case Block::ST:
            transformer_params(u, p, b.cin, b.heads, HEAD_DIM, b.depth, c.context_dim, c.use_linear != 0, false, c.use_image_attention != 0);
            break;
        case Block::TT:
            transformer_params(u, p, b.cin, b.heads, HEAD_DIM, b.depth, 0, c.use_linear != 0, true, false);
            break;
        case Block::DOWN:
            add_w(u, p + ".op.weight", {b.cout, b.cin, 3, 3});
            add_w(u, p + ".op.bias", {b.cout});
            break;
        case Block::UP:
            add_w(u, p + ".conv.weight", {b.cout, b.cin, 3, 3});
            add_w(u, p + ".conv.bias", {b.cout});
            break;
    }
}

// openaimodel3d.py:441-649: the constructor's bookkeeping of ch / ds / input_block_chans
int build_program(ds_unet* u) {
    const ds_unet_config& c = u->cfg;
    const int mc = c.model_channels;
    auto has_attn = [&](int ds) {
        for (int i = 0; i < c.n_attention_resolutions; ++i)
            if (c.attention_resolutions[i] == ds) return true;
        return false;
    };
    auto attn_blocks = [&](std::vector<Block>& g, const std::string& prefix, int start, int ch) {
        Block st;
        st.kind = Block::ST; st.prefix = prefix + "." + std::to_string(start); st.cin = st.cout = ch;
        st.heads = ch / HEAD_DIM; st.depth = c.transformer_depth;
        g.push_back(st);
        if (c.temporal_attention) {
            Block tt = st;
            tt.kind = Block::TT; tt.prefix = prefix + "." + std::to_string(start + 1); tt.depth = c.temporal_transformer_depth;
            g.push_back(tt);
        }
    };
    const bool tconv = c.temporal_conv != 0;
    Block ci;
    ci.kind = Block::CONV_IN; ci.prefix = "input_blocks.0.0"; ci.cin = c.in_channels; ci.cout = mc;
    u->inputs.push_back({ci});
    std::vector<int> skip_chans = {mc};
    int ch = mc, ds = 1;
    for (int level = 0; level < c.n_channel_mult; ++level) {
        const int mult = c.channel_mult[level];
        for (int r = 0; r < c.num_res_blocks; ++r) {
            const int idx = (int)u->inputs.size();
            Block rb;
            rb.kind = Block::RES; rb.prefix = "input_blocks." + std::to_string(idx) + ".0"; rb.cin = ch; rb.cout = mult * mc; rb.tconv = tconv;
            std::vector<Block> g = {rb};
            ch = mult * mc;
            if (has_attn(ds)) attn_blocks(g, "input_blocks." + std::to_string(idx), 1, ch);
            u->inputs.push_back(g);
            skip_chans.push_back(ch);
        }
        if (level != c.n_channel_mult - 1) {
            const int idx = (int)u->inputs.size();
            Block d;
            d.kind = Block::DOWN; d.prefix = "input_blocks." + std::to_string(idx) + ".0"; d.cin = d.cout = ch;
            u->inputs.push_back({d});
            skip_chans.push_back(ch);
            ds *= 2;
        }
    }
    {
        Block rb;
        rb.kind = Block::RES; rb.prefix = "middle_block.0"; rb.cin = rb.cout = ch; rb.tconv = tconv;
        u->middle.push_back(rb);
        attn_blocks(u->middle, "middle_block", 1, ch);
        rb.prefix = "middle_block." + std::to_string(u->middle.size());
        u->middle.push_back(rb);
    }
    for (int level = c.n_channel_mult - 1; level >= 0; --level) {
        const int mult = c.channel_mult[level];
        for (int i = 0; i <= c.num_res_blocks; ++i) {
            const int idx = (int)u->outputs.size();
            const int ich = skip_chans.back();
            skip_chans.pop_back();
            Block rb;
            rb.kind = Block::RES; rb.prefix = "output_blocks." + std::to_string(idx) + ".0"; rb.cin = ch + ich; rb.cout = mult * mc; rb.tconv = tconv;
            std::vector<Block> g = {rb};
            ch = mult * mc;
            if (has_attn(ds)) attn_blocks(g, "output_blocks." + std::to_string(idx), 1, ch);
            if (level && i == c.num_res_blocks) {
                Block up;
                up.kind = Block::UP; up.prefix = "output_blocks." + std::to_string(idx) + "." + std::to_string(g.size()); up.cin = up.cout = ch;
                g.push_back(up);
                ds /= 2;
            }
            u->outputs.push_back(g);
        }
    }
    // (channels of h, channels of the skip tensor) per decoder group
    std::vector<int> sk;
    int cch = 0;
    for (auto& g : u->inputs) {
        for (auto& b : g)
            if (b.kind != Block::ST && b.kind != Block::TT) cch = b.cout;
        sk.push_back(cch);
    }
    for (size_t gi = 0; gi < u->outputs.size(); ++gi) {
        const int cs = sk[sk.size() - 1 - gi];
        u->cat_ch.push_back({u->outputs[gi][0].cin - cs, cs});
    }
    // parameter table in the reference's state-dict order (unet_spec.param_shapes)
    const int ted = 4 * mc;
    for (int e = 0; e < (c.fps_cond ? 2 : 1); ++e) {
        const std::string n = e == 0 ? "time_embed" : "fps_embedding";
        add_w(u, n + ".0.weight", {ted, mc});
        add_w(u, n + ".0.bias", {ted});
        add_w(u, n + ".2.weight", {ted, ted});
        add_w(u, n + ".2.bias", {ted});
    }
    for (size_t gi = 0; gi < u->inputs.size(); ++gi) {
        for (auto& b : u->inputs[gi]) block_params(u, b);
        if (gi == 0 && c.addition_attention)   // init_attn: TemporalTransformer(mc, 8 heads, depth = transformer_depth), conv1d projections (openaimodel3d.py:425-439)
            transformer_params(u, "init_attn.0", mc, 8, c.num_head_channels, c.transformer_depth, 0, false, true, false);
    }
    for (auto& b : u->middle) block_params(u, b);
    for (auto& g : u->outputs)
        for (auto& b : g) block_params(u, b);
    add_w(u, "out.0.weight", {mc});
    add_w(u, "out.0.bias", {mc});
    add_w(u, "out.2.weight", {c.out_channels, mc, 3, 3});
    add_w(u, "out.2.bias", {c.out_channels});
    return DS_OK;
}

// ------------------------------------------------------------------------------------------------ the packing plan
template <typename F>
int with_src(const WeightSpec* w, F&& f) {     // dispatch on the raw tensor's element type
    if (w->dtype == DS_F32) return f((const float*)w->data);
    return f((const f16*)w->data);
}

struct Planner {
    ds_unet* u;
    void item(const std::string& name, size_t bytes, std::function<int(char*, hipStream_t)> fill, long rows = 0, int dtype = DS_F32) {
        PackItem it;
        it.name = name;
        it.off = u->packed_total;
        it.bytes = bytes;
        it.rows = rows ? rows : (long)(bytes / 4);
        it.dtype = dtype;
        it.fill = std::move(fill);
        u->packed_total += (bytes + 255) / 256 * 256;
        u->pindex[name] = (int)u->items.size();
        u->items.push_back(std::move(it));
    }
    // fp16 [N][kdst] operand from up to three row-concatenated sources of the same K; gamma (packed fp32 vector name) scales the columns
    void w16(const std::string& name, std::vector<std::string> keys, int layout, int cin, int kdst, const std::string& gamma_item, int geglu_inner) {
        ds_unet* uu = u;
        long n_rows = 0;
        for (auto& k : keys) n_rows += uu->w(k)->shape[0];
        const size_t plane = (size_t)n_rows * kdst * 2;
        item(name, uu->wide ? 2 * plane : plane, [=](char* dst, hipStream_t st) {
            long row0 = 0;
            for (auto& k : keys) {
                const WeightSpec* w = uu->w(k);
                const long rows = w->shape[0];
                const int ksrc = (int)(w->numel() / rows);
                const float* g = gamma_item.empty() ? nullptr : (const float*)uu->P(gamma_item);
                f16* d = (f16*)dst + row0 * kdst;
                f16* dl = uu->wide ? (f16*)(dst + plane) + row0 * kdst : nullptr;
                with_src(w, [&](auto* src) {
                    pack_w16_kernel<<<grid_for(rows * kdst), 256, 0, st>>>(d, rows, kdst, src, ksrc, layout, cin, g, geglu_inner, dl, ds_wide_lo_scale());
                    return 0;
                });
                row0 += rows;
            }
            return DS_OK;
        }, n_rows, DS_F16);
    }
    void vec(const std::string& name, const std::string& key, const std::string& add_key = "", int geglu_inner = 0) {
        ds_unet* uu = u;
        const long n = uu->w(key)->numel();
        item(name, (size_t)n * 4, [=](char* dst, hipStream_t st) {
            const WeightSpec* a = uu->w(key);
            const WeightSpec* b = add_key.empty() ? nullptr : uu->w(add_key);
            if (b && b->dtype != a->dtype) { ds_set_error("ds_unet_pack: %s and %s differ in dtype", key.c_str(), add_key.c_str()); return DS_EINVAL; }
            with_src(a, [&](auto* src) {
                using ST = std::remove_cv_t<std::remove_pointer_t<decltype(src)>>;
                vec_f32_kernel<ST><<<grid_for(n), 256, 0, st>>>((float*)dst, n, src, b ? (const ST*)b->data : nullptr, geglu_inner);
                return 0;
            });
            return DS_OK;
        });
    }
    void lin(const std::string& prefix, bool bias = true) {
        const WeightSpec* w = u->w(prefix + ".weight");
        const int k = (int)(w->numel() / w->shape[0]);
        w16(prefix + ".w", {prefix + ".weight"}, L_LINEAR, k, k, "", 0);
        if (bias) vec(prefix + ".b", prefix + ".bias");
    }
    void conv3(const std::string& prefix) {
        const WeightSpec* w = u->w(prefix + ".weight");
        const int cin = (int)w->shape[1];
        w16(prefix + ".w", {prefix + ".weight"}, L_CONV3, cin, 9 * cin, "", 0);
        vec(prefix + ".b", prefix + ".bias");
    }
    void tconv(const std::string& prefix) {
        const WeightSpec* w = u->w(prefix + ".weight");
        const int cin = (int)w->shape[1];
        w16(prefix + ".w", {prefix + ".weight"}, L_TCONV, cin, 3 * cin, "", 0);
        vec(prefix + ".b", prefix + ".bias");
    }
    void norm(const std::string& prefix) {
        vec(prefix + ".g", prefix + ".weight");
        vec(prefix + ".be", prefix + ".bias");
    }
    // a projection fed by LayerNorm `ln`: plain operand (+ bias), or the LayerNorm folded in (ds_gemm_f16_ln)
    void proj(const std::string& name, std::vector<std::string> keys, const std::string& ln, const std::string& bias_key, bool geglu) {
        ds_unet* uu = u;
        const WeightSpec* w0 = u->w(keys[0]);
        const int K = (int)w0->shape[1];
        long n_rows = 0;
        for (auto& k : keys) n_rows += u->w(k)->shape[0];
        const int gi = geglu ? (int)(n_rows / 2) : 0;
        if (!u->fold) {
            w16(name + ".w", keys, L_LINEAR, K, K, "", gi);
            if (!bias_key.empty()) vec(name + ".b", bias_key, "", gi);
            return;
        }
        w16(name + ".wg", keys, L_LINEAR, K, K, ln + ".g", gi);
        const std::string wg = name + ".wg";
        item(name + ".cs", (size_t)n_rows * 4, [=](char* dst, hipStream_t st) {
            colsum_kernel<<<(int)((n_rows + 3) / 4), 256, 0, st>>>((const f16*)uu->P(wg), (float*)dst, n_rows, K);
            return DS_OK;
        });
        item(name + ".cb", (size_t)n_rows * 4, [=](char* dst, hipStream_t st) {
            long row0 = 0;
            if (geglu && keys.size() != 1) { ds_set_error("ds_unet_pack: GEGLU projection from several sources"); return DS_EINVAL; }
            for (auto& k : keys) {
                const WeightSpec* w = uu->w(k);
                const WeightSpec* b = bias_key.empty() ? nullptr : uu->w(bias_key);
                const long rows = w->shape[0];
                with_src(w, [&](auto* src) {
                    using ST = std::remove_cv_t<std::remove_pointer_t<decltype(src)>>;
                    colbias_kernel<ST><<<(int)((rows + 3) / 4), 256, 0, st>>>((float*)dst + row0, rows, src, K, (const float*)uu->P(ln + ".be"),
                                                                                b ? (const ST*)b->data : nullptr, gi);
                    return 0;
                });
                row0 += rows;
            }
            return DS_OK;
        });
    }
    void transformer(const std::string& prefix, int depth, bool cross, bool img) {
        norm(prefix + ".norm");
        lin(prefix + ".proj_in");
        lin(prefix + ".proj_out");
        for (int d = 0; d < depth; ++d) {
            const std::string p = prefix + ".transformer_blocks." + std::to_string(d);
            for (int n = 1; n <= 3; ++n) norm(p + ".norm" + std::to_string(n));
            proj(p + ".attn1.qkv", {p + ".attn1.to_q.weight", p + ".attn1.to_k.weight", p + ".attn1.to_v.weight"}, p + ".norm1", "", false);
            lin(p + ".attn1.to_out.0");
            if (cross) {
                proj(p + ".attn2.to_q", {p + ".attn2.to_q.weight"}, p + ".norm2", "", false);
                const int kc = (int)u->w(p + ".attn2.to_k.weight")->shape[1];
                w16(p + ".attn2.kv.w", {p + ".attn2.to_k.weight", p + ".attn2.to_v.weight"}, L_LINEAR, kc, kc, "", 0);
                if (img) w16(p + ".attn2.kv_ip.w", {p + ".attn2.to_k_ip.weight", p + ".attn2.to_v_ip.weight"}, L_LINEAR, kc, kc, "", 0);
            } else {
                proj(p + ".attn2.qkv", {p + ".attn2.to_q.weight", p + ".attn2.to_k.weight", p + ".attn2.to_v.weight"}, p + ".norm2", "", false);
            }
            lin(p + ".attn2.to_out.0");
            proj(p + ".ff1", {p + ".ff.net.0.proj.weight"}, p + ".norm3", p + ".ff.net.0.proj.bias", true);
            lin(p + ".ff.net.2");
        }
    }
};

void plan_pack(ds_unet* u) {
    Planner pl{u};
    const ds_unet_config& c = u->cfg;
    pl.lin("time_embed.0");
    pl.lin("time_embed.2");
    if (c.fps_cond) { pl.lin("fps_embedding.0"); pl.lin("fps_embedding.2"); }
    std::vector<std::string> emb_keys, emb_prefixes;
    int off = 0;
    auto block = [&](const Block& b) {
        const std::string& p = b.prefix;
        switch (b.kind) {
            case Block::CONV_IN: {
                const int k = 9 * b.cin;
                u->kpad_in = (k + 63) / 64 * 64;
                pl.w16(p + ".w", {p + ".weight"}, L_CONV3, b.cin, u->kpad_in, "", 0);
                pl.vec(p + ".b", p + ".bias");
                break;
            }
            case Block::RES:
                pl.norm(p + ".in_layers.0");
                pl.conv3(p + ".in_layers.2");
                pl.norm(p + ".out_layers.0");
                pl.conv3(p + ".out_layers.3");
                if (b.cin != b.cout) pl.lin(p + ".skip_connection");
                emb_keys.push_back(p + ".emb_layers.1.weight");
                emb_prefixes.push_back(p);
                u->emb_off[p] = off;
                off += b.cout;
                if (b.tconv)
                    for (int i = 1; i <= 4; ++i) {
                        const std::string q = p + ".temopral_conv.conv" + std::to_string(i);
                        pl.norm(q + ".0");
                        pl.tconv(q + "." + std::to_string(i == 1 ? 2 : 3));
                    }
                break;
            case Block::ST: pl.transformer(p, b.depth, true, c.use_image_attention != 0); break;
            case Block::TT: pl.transformer(p, b.depth, false, false); break;
            case Block::DOWN: pl.conv3(p + ".op"); break;
            case Block::UP: pl.conv3(p + ".conv"); break;
        }
    };
    for (size_t gi = 0; gi < u->inputs.size(); ++gi) {
        for (auto& b : u->inputs[gi]) block(b);
        if (gi == 0 && c.addition_attention) pl.transformer("init_attn.0", c.transformer_depth, false, false);
    }
    for (auto& b : u->middle) block(b);
    for (auto& g : u->outputs)
        for (auto& b : g) block(b);
    pl.norm("out.0");
    pl.conv3("out.2");
    u->emb_total = off;
    // the time-embedding projections of all ResBlocks in ONE matrix, the conv-1 bias folded into its bias
    const int ted = 4 * c.model_channels;
    ds_unet* uu = u;
    const size_t emb_plane = (size_t)off * ted * 2;
    pl.item("emb_all.w", uu->wide ? 2 * emb_plane : emb_plane, [=](char* dst, hipStream_t st) {
        long row0 = 0;
        for (auto& k : emb_keys) {
            const WeightSpec* w = uu->w(k);
            const long rows = w->shape[0];
            f16* d = (f16*)dst + row0 * ted;
            f16* dl = uu->wide ? (f16*)(dst + emb_plane) + row0 * ted : nullptr;
            with_src(w, [&](auto* src) {
                pack_w16_kernel<<<grid_for(rows * ted), 256, 0, st>>>(d, rows, ted, src, ted, L_LINEAR, ted, (const float*)nullptr, 0, dl, ds_wide_lo_scale());
                return 0;
            });
            row0 += rows;
        }
        return DS_OK;
    }, off, DS_F16);
    pl.item("emb_all.b", (size_t)off * 4, [=](char* dst, hipStream_t st) {
        long row0 = 0;
        for (auto& p : emb_prefixes) {
            const WeightSpec* a = uu->w(p + ".emb_layers.1.bias");
            const WeightSpec* b = uu->w(p + ".in_layers.2.bias");
            const long n = a->numel();
            if (a->dtype != b->dtype) { ds_set_error("ds_unet_pack: %s biases differ in dtype", p.c_str()); return DS_EINVAL; }
            with_src(a, [&](auto* src) {
                using ST = std::remove_cv_t<std::remove_pointer_t<decltype(src)>>;
                vec_f32_kernel<ST><<<grid_for(n), 256, 0, st>>>((float*)dst + row0, n, src, (const ST*)b->data, 0);
                return 0;
            });
            row0 += n;
        }
        return DS_OK;
    });
}

// ------------------------------------------------------------------------------------------------ scratch arena + tensors
struct Arena {              // first-fit free list over [0, cap): deterministic, so a dry run gives the exact peak and the offsets repeat
    size_t cap = 0, peak = 0;
    std::map<size_t, size_t> free_;     // offset -> bytes
    bool dry = false, failed = false;
    explicit Arena(size_t capacity, bool dry_run) : cap(dry_run ? (size_t)1 << 60 : capacity), dry(dry_run) { free_[0] = cap; }
    size_t alloc(size_t bytes) {
        bytes = (bytes + 255) / 256 * 256;
        if (bytes == 0) bytes = 256;
        for (auto it = free_.begin(); it != free_.end(); ++it) {
            if (it->second >= bytes) {
                const size_t off = it->first, rest = it->second - bytes;
                free_.erase(it);
                if (rest) free_[off + bytes] = rest;
                if (off + bytes > peak) peak = off + bytes;
                return off;
            }
        }
        failed = true;
        return 0;
    }
    void release(size_t off, size_t bytes) {
        bytes = (bytes + 255) / 256 * 256;
        if (bytes == 0) bytes = 256;
        auto it = free_.emplace(off, bytes).first;
        auto nx = std::next(it);
        if (nx != free_.end() && it->first + it->second == nx->first) { it->second += nx->second; free_.erase(nx); }
        if (it != free_.begin()) {
            auto pv = std::prev(it);
            if (pv->first + pv->second == it->first) { pv->second += it->second; free_.erase(it); }
        }
    }
};

struct Buf {
    Arena* a;
    size_t off, bytes;
    bool ok;                        // false: the arena had no room (offset 0 is a placeholder, never handed to a kernel and never released)
    Buf(Arena* a_, size_t b) : a(a_), off(0), bytes(b), ok(false) {
        if (a->failed) return;      // an earlier allocation failed: the program is being wound down, take nothing more
        off = a->alloc(b);
        ok = !a->failed;
    }
    ~Buf() { if (ok) a->release(off, bytes); }
};

struct Ten {                       // rows x cols view (row stride ld elements) of an arena buffer or of caller memory
    std::shared_ptr<Buf> buf;
    char* ext = nullptr;           // caller memory (weights, inputs) when buf is null
    size_t boff = 0;               // byte offset inside buf
    long rows = 0;
    int cols = 0, ld = 0, dt = DS_F16;
    int esz() const { return dt == DS_F32 ? 4 : 2; }
    bool is32() const { return dt == DS_F32; }
    explicit operator bool() const { return buf || ext; }
};

struct Geo { int B, T, H, W; };

// ------------------------------------------------------------------------------------------------ the launch program
struct Prog {
    ds_unet* u;
    Arena arena;
    char* base;                 // workspace (null in a dry run)
    hipStream_t st;
    std::string* log;           // launch trace (dry run) or null
    int rc = DS_OK;
    Prog(ds_unet* u_, char* ws, size_t ws_bytes, hipStream_t s, std::string* lg) : u(u_), arena(ws_bytes, ws == nullptr), base(ws), st(s), log(lg) {}

    char* ptr(const Ten& t) const {
        if (!t) return nullptr;
        if (t.buf) return (char*)((uintptr_t)base + t.buf->off + t.boff);
        return t.ext;
    }
    Ten make(long rows, int cols, int dt) {
        Ten t;
        t.rows = rows; t.cols = cols; t.ld = cols; t.dt = dt;
        t.buf = std::make_shared<Buf>(&arena, (size_t)rows * cols * t.esz());
        return t;
    }
    Ten raw(size_t bytes) { return make((long)((bytes + 3) / 4), 1, DS_F32); }
    static Ten cols(const Ten& t, int c0, int n) {
        Ten v = t;
        v.boff += (size_t)c0 * t.esz();
        if (!t.buf) v.ext = t.ext + (size_t)c0 * t.esz();
        v.cols = n;
        return v;
    }
    static Ten rows(const Ten& t, long r0, long n) {
        Ten v = t;
        v.boff += (size_t)r0 * t.ld * t.esz();
        if (!t.buf) v.ext = t.ext + (size_t)r0 * t.ld * t.esz();
        v.rows = n;
        return v;
    }
    Ten wt(const std::string& name, long rows, int cols, int dt) const {
        Ten t;
        t.ext = (char*)u->P(name);
        t.rows = rows; t.cols = cols; t.ld = cols; t.dt = dt;
        if (!t.ext) t.ext = (char*)16;   // dry run before packing: never dereferenced
        return t;
    }
    bool live() const { return log == nullptr && !arena.failed; }   // nothing is launched once an allocation has failed (ds_unet_forward returns DS_EINVAL)
    void tr(const char* fmt, ...) {
        if (!log) return;
        char line[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(line, sizeof(line), fmt, ap);
        va_end(ap);
        log->append(line);
        log->push_back('\n');
    }
    void chk(int r) { if (r != DS_OK && rc == DS_OK) rc = r; }
    // one kernel-family call of the program: nothing in a dry run; with a launch hook installed the host callback runs right before
    // and right after the call enqueues its kernel(s) (per-launch HIP events, the tests' LDS / register poison launch)
    template <typename F>
    void launch(const char* kernel, double flops, std::initializer_list<int32_t> info, F&& call) {
        if (!live()) return;
        if (u->launch_hook) u->launch_hook(u->hook_user, 0, kernel, flops, info.begin(), (int)info.size(), st);
        chk(call());
        if (u->launch_hook) u->launch_hook(u->hook_user, 1, kernel, flops, info.begin(), (int)info.size(), st);
    }
    void tap(const std::string& block, const Ten& h, const Geo& geo) {
        if (live() && u->block_tap) u->block_tap(u->hook_user, block.c_str(), ptr(h), h.rows, h.cols, h.ld, h.dt, geo.B, geo.T, geo.H, geo.W, st);
    }

    // ---- kernels ----
    struct ConvGeo { int nimg = 0, hin = 0, win = 0, hout = 0, wout = 0, stride = 1, upsample = 0; };
    Ten gemm(const Ten& A, const Ten& W, const float* bias, const Ten& residual, long M, int N, int K, int epilogue, int a_mode = DS_A_DENSE,
             int cin = 0, const ConvGeo* cg = nullptr, int t_len = 0, int hw = 0, int bias_rows = INT_MAX, int ldbias = 0, Ten out = Ten(),
             const Ten& cstats = Ten()) {
        const int n_out = (epilogue & DS_EPI_GEGLU) ? N / 2 : N;
        if (u->wide) epilogue |= DS_EPI_OUT_F32;             // the wide mode stores nothing in fp16
        if (residual && residual.is32()) epilogue |= DS_EPI_RES_F32;
        if (out && out.is32()) epilogue |= DS_EPI_OUT_F32;
        if (!out) out = make(M, n_out, (epilogue & DS_EPI_OUT_F32) ? DS_F32 : DS_F16);
        ds_gemm_desc d;
        memset(&d, 0, sizeof(d));
        d.M = (int)M; d.N = N; d.K = K; d.a_mode = a_mode;
        d.cin = cin ? cin : K;
        d.lda = A.ld;
        if (cg) { d.nimg = cg->nimg; d.hin = cg->hin; d.win = cg->win; d.hout = cg->hout; d.wout = cg->wout; d.stride = cg->stride; d.upsample = cg->upsample; }
        d.t_len = t_len; d.hw = hw;
        d.ldc = out.ld;
        d.ldr = residual ? residual.ld : 0;
        d.bias_rows = bias_rows;
        d.ldbias = ldbias ? ldbias : N;
        d.epilogue = epilogue;
        tr("gemm M=%ld N=%d K=%d mode=%d cin=%d lda=%d ldc=%d ldr=%d brows=%d ldb=%d epi=%d conv=%d,%d,%d,%d,%d,%d,%d t=%d,%d bias=%d res=%d stats=%d",
           M, N, K, a_mode, d.cin, d.lda, d.ldc, d.ldr, bias_rows, d.ldbias, epilogue, d.nimg, d.hin, d.win, d.hout, d.wout, d.stride, d.upsample, t_len,
           hw, bias ? 1 : 0, residual ? 1 : 0, cstats ? cstats.ld / 2 : 0);
        if (u->wide && (!A.is32() || !out.is32() || (residual && !residual.is32()))) {
            ds_set_error("ds_unet_forward: the wide mode met an fp16 tensor (program error)");
            chk(DS_EINVAL);
            return out;
        }
        launch("gemm", 2.0 * (double)M * N * K, {a_mode, (int32_t)M, N, K, epilogue, residual ? 1 : 0}, [&] {
            if (u->wide)       // the lo plane of a packed matrix follows its hi plane (Planner::w16)
                return ds_gemm_wide((const float*)ptr(A), ptr(W), ptr(W) + (size_t)W.rows * W.ld * 2, bias, (const float*)ptr(residual), (float*)ptr(out), &d, st);
            if (cstats) return ds_gemm_f16_stats(ptr(A), ptr(W), bias, ptr(residual), ptr(out), (float*)ptr(cstats), cstats.ld / 2, &d, st);
            return ds_gemm_f16(ptr(A), ptr(W), bias, ptr(residual), ptr(out), &d, st);
        });
        return out;
    }
    Ten gemm_ln(const Ten& x, const std::string& name, const Ten& stats, long M, int N, int K, int epilogue) {
        const int n_out = (epilogue & DS_EPI_GEGLU) ? N / 2 : N;
        Ten out = make(M, n_out, DS_F16);
        ds_gemm_desc d;
        memset(&d, 0, sizeof(d));
        d.M = (int)M; d.N = N; d.K = K; d.a_mode = DS_A_DENSE; d.cin = K; d.lda = x.ld; d.ldc = out.ld; d.ldr = 0;
        d.bias_rows = INT_MAX; d.ldbias = N; d.epilogue = epilogue;
        tr("gemm_ln M=%ld N=%d K=%d lda=%d ldc=%d epi=%d", M, N, K, d.lda, d.ldc, epilogue);
        launch("gemm", 2.0 * (double)M * N * K, {DS_A_DENSE, (int32_t)M, N, K, epilogue, 0}, [&] {
            return ds_gemm_f16_ln(ptr(x), u->P(name + ".wg"), (const float*)ptr(stats), (const float*)u->P(name + ".cs"), (const float*)u->P(name + ".cb"),
                                  ptr(out), &d, st);
        });
        return out;
    }
    // GroupNorm statistics from the producer (ds_gemm_f16_stats -> ds_groupnorm_rows_colstats): where producer and norm are adjacent
    // in the program and the norm would otherwise run its own statistics pass (instances of <= 256 rows keep the one-launch form)
    bool fuse_gn(int rows_per) const { return u->gn_fused && rows_per % 32 == 0 && rows_per > 256; }
    Ten stats_table(long rows, int cols) { return make((rows + 31) / 32, 2 * cols, DS_F32); }   // (sum, sumsq) per 32-row block and column
    Ten groupnorm(const Ten& x, const std::string& prefix, int ninst, int rows_per, int C, float eps, int silu, Ten* raw16 = nullptr,
                  const Ten& cstats = Ten()) {
        if (u->wide) {
            Ten stt = raw(ds_groupnorm_wide_scratch_floats(ninst, rows_per, 32) * 4);
            Ten yw = make(x.rows, C, DS_F32);
            tr("groupnorm_wide ldx=%d ninst=%d rows=%d C=%d silu=%d eps=%g", x.ld, ninst, rows_per, C, silu, (double)eps);
            launch("groupnorm", 0.0, {ninst, rows_per, C}, [&] {
                return ds_groupnorm_wide((const float*)ptr(x), x.ld, (const float*)u->P(prefix + ".g"), (const float*)u->P(prefix + ".be"), (float*)ptr(yw),
                                         (float*)ptr(stt), ninst, rows_per, C, 32, eps, silu, st);
            });
            if (raw16) *raw16 = x;        // the un-normalised operand is the fp32 tensor itself
            return yw;
        }
        Ten ws = raw(ds_groupnorm_stats_workspace_floats(ninst, rows_per, 32) * 4);
        Ten y = make(x.rows, C, DS_F16);
        Ten r;
        if (raw16) r = *raw16 = make(x.rows, C, DS_F16);
        tr("groupnorm xdt=%d ldx=%d ninst=%d rows=%d C=%d silu=%d raw=%d eps=%g stats=%d", x.dt, x.ld, ninst, rows_per, C, silu, raw16 ? 1 : 0, (double)eps,
           cstats ? cstats.ld / 2 : 0);
        launch("groupnorm", 0.0, {ninst, rows_per, C}, [&] {
            if (cstats)
                return ds_groupnorm_rows_colstats(ptr(x), x.dt, x.ld, (const float*)ptr(cstats), cstats.ld / 2, (const float*)u->P(prefix + ".g"),
                                                  (const float*)u->P(prefix + ".be"), ptr(y), ptr(r), (float*)ptr(ws), ninst, rows_per, C, 32, eps, silu, st);
            return ds_groupnorm_rows(ptr(x), x.dt, x.ld, (const float*)u->P(prefix + ".g"), (const float*)u->P(prefix + ".be"), ptr(y), ptr(r),
                                     (float*)ptr(ws), ninst, rows_per, C, 32, eps, silu, st);
        });
        return y;
    }
    Ten layernorm(const Ten& x, const std::string& prefix) {
        if (u->wide) {
            Ten yw = make(x.rows, x.cols, DS_F32);
            tr("layernorm_wide rows=%ld C=%d", x.rows, x.cols);
            if (x.ld != x.cols) { ds_set_error("ds_unet_forward: layernorm_wide needs dense rows"); chk(DS_EINVAL); return yw; }
            launch("layernorm", 0.0, {(int32_t)x.rows, x.cols}, [&] {
                return ds_layernorm_wide((const float*)ptr(x), (const float*)u->P(prefix + ".g"), (const float*)u->P(prefix + ".be"), (float*)ptr(yw), x.rows, x.cols, 1e-5f, st);
            });
            return yw;
        }
        Ten y = make(x.rows, x.cols, DS_F16);
        tr("layernorm xdt=%d rows=%ld C=%d", x.dt, x.rows, x.cols);
        launch("layernorm", 0.0, {(int32_t)x.rows, x.cols}, [&] {
            return ds_layernorm_rows(ptr(x), x.dt, (const float*)u->P(prefix + ".g"), (const float*)u->P(prefix + ".be"), ptr(y), (int)x.rows, x.cols, 1e-5f, st);
        });
        return y;
    }
    Ten layernorm_stats(const Ten& x) {
        Ten s = make(x.rows, 2, DS_F32);
        tr("layernorm_stats rows=%ld C=%d", x.rows, x.cols);
        launch("layernorm_stats", 0.0, {(int32_t)x.rows, x.cols}, [&] { return ds_layernorm_stats(ptr(x), (float*)ptr(s), (int)x.rows, x.cols, 1e-5f, st); });
        return s;
    }
    void attention(const Ten& q, const Ten& k, const Ten& v, const Ten& o, int batch, int heads, int nq, int nk, int kvdiv, float scale, int acc) {
        tr("attention batch=%d heads=%d nq=%d nk=%d ldq=%d ldk=%d ldv=%d ldo=%d kvdiv=%d acc=%d", batch, heads, nq, nk, q.ld, k.ld, v.ld, o.ld, kvdiv, acc);
        launch("attention", 4.0 * batch * heads * (double)nq * nk * HEAD_DIM, {batch, heads, nq, nk}, [&] {
            if (u->wide)
                return ds_attention_wide((const float*)ptr(q), (const float*)ptr(k), (const float*)ptr(v), (float*)ptr(o), batch, heads, nq, nk, q.ld, k.ld, v.ld, o.ld, kvdiv, scale, acc, st);
            return ds_attention_f16(ptr(q), ptr(k), ptr(v), ptr(o), batch, heads, nq, nk, q.ld, k.ld, v.ld, o.ld, kvdiv, scale, acc, st);
        });
    }
    void tattention(const Ten& q, const Ten& k, const Ten& v, const Ten& o, int nb, int T, int hw, int heads, float scale) {
        tr("temporal_attention nb=%d T=%d hw=%d heads=%d ldq=%d ldk=%d ldv=%d ldo=%d", nb, T, hw, heads, q.ld, k.ld, v.ld, o.ld);
        launch("temporal_attention", 4.0 * nb * hw * heads * (double)T * T * HEAD_DIM, {nb, T, hw, heads}, [&] {
            if (u->wide)
                return ds_temporal_attention_wide((const float*)ptr(q), (const float*)ptr(k), (const float*)ptr(v), (float*)ptr(o), nb, T, hw, heads, q.ld, k.ld, v.ld, o.ld, scale, st);
            return ds_temporal_attention_f16(ptr(q), ptr(k), ptr(v), ptr(o), nb, T, hw, heads, q.ld, k.ld, v.ld, o.ld, scale, st);
        });
    }
    Ten cast16(const Ten& x) {
        Ten y = make(x.rows, x.cols, DS_F16);
        tr("cast_rows rows=%ld C=%d ldx=%d", x.rows, x.cols, x.ld);
        launch("cast_rows", 0.0, {(int32_t)x.rows, x.cols}, [&] { return ds_cast_rows_f32_f16((const float*)ptr(x), x.ld, ptr(y), y.ld, x.rows, x.cols, st); });
        return y;
    }
    Ten operand(const Ten& h) { return (h.is32() && !u->wide) ? cast16(h) : h; }
    int adt() const { return u->wide ? DS_F32 : DS_F16; }      // storage type of the tensors that are fp16 in every other mode
    // rows of src -> rows of dst (2-D device copy on the stream; capturable)
    void copy_rows(const Ten& dst, const Ten& src) {
        tr("copy rows=%ld bytes=%ld", src.rows, (long)src.cols * src.esz());
        if (live() && hipMemcpy2DAsync(ptr(dst), (size_t)dst.ld * dst.esz(), ptr(src), (size_t)src.ld * src.esz(), (size_t)src.cols * src.esz(),
                                       (size_t)src.rows, hipMemcpyDeviceToDevice, st) != hipSuccess) {
            ds_set_error("ds_unet_forward: hipMemcpy2DAsync failed");
            chk(DS_ELAUNCH);
        }
    }
    Ten dup(const Ten& t) {       // torch.cat([t, t], 0)
        Ten d = make(2 * t.rows, t.cols, t.dt);
        copy_rows(rows(d, 0, t.rows), t);
        copy_rows(rows(d, t.rows, t.rows), t);
        return d;
    }

    int res_epi(int e = 0) const { return e | (u->strict ? DS_EPI_OUT_F32 : 0); }

    Ten linear(const Ten& a, const std::string& prefix, const Ten& residual = Ten(), int epilogue = 0, Ten out = Ten()) {
        const WeightSpec* w = u->w(prefix + ".weight");
        const int N = (int)w->shape[0], K = (int)(w->numel() / N);
        return gemm(a, wt(prefix + ".w", N, K, DS_F16), (const float*)u->P(prefix + ".b"), residual, a.rows, N, K, epilogue, DS_A_DENSE, 0, nullptr, 0, 0,
                    INT_MAX, 0, out);
    }
    Ten conv3(const Ten& a, const std::string& prefix, int nimg, int hin, int win, int cin, int stride, int upsample, const Ten& residual,
              const float* bias, int bias_rows, int ldbias, int epilogue, Ten out, int* hout_, int* wout_, const Ten& cstats = Ten()) {
        const WeightSpec* w = u->w(prefix + ".weight");
        const int N = (int)w->shape[0], K = 9 * cin;
        ConvGeo cg;
        const int hl = upsample ? 2 * hin : hin, wl = upsample ? 2 * win : win;
        cg.nimg = nimg; cg.hin = hin; cg.win = win; cg.hout = (hl - 1) / stride + 1; cg.wout = (wl - 1) / stride + 1; cg.stride = stride; cg.upsample = upsample;
        if (hout_) { *hout_ = cg.hout; *wout_ = cg.wout; }
        return gemm(a, wt(prefix + ".w", N, K, DS_F16), bias ? bias : (const float*)u->P(prefix + ".b"), residual, (long)nimg * cg.hout * cg.wout, N, K, epilogue,
                    DS_A_CONV3, cin, &cg, 0, 0, bias ? bias_rows : INT_MAX, bias ? ldbias : 0, out, cstats);
    }

    struct Ctx { Ten text, img; int ltxt = 0, limg = 0; };

    Ten transformer_block(Ten x, const std::string& p, int heads, bool spatial, Geo& geo, const Ctx& ctx, bool do_dup, bool last) {
        int B = geo.B;
        const int T = geo.T, H = geo.H, W = geo.W;
        long M = x.rows;
        const int inner = x.cols;
        const float scale = 0.125f;    // HEAD_DIM ** -0.5
        const int rs = u->inner32 ? DS_EPI_OUT_F32 : 0;     // the block's own stream (its three adds)
        auto ln_proj = [&](const Ten& xin, const std::string& ln, const std::string& name, int N, int epilogue) {
            if (u->fold) {
                Ten stt = layernorm_stats(xin);
                return gemm_ln(xin, name, stt, xin.rows, N, inner, epilogue);
            }
            Ten n = layernorm(xin, p + "." + ln);
            return gemm(n, wt(name + ".w", N, inner, DS_F16), u->hasP(name + ".b") ? (const float*)u->P(name + ".b") : nullptr, Ten(), xin.rows, N, inner, epilogue);
        };
        auto self_attn = [&](const std::string& name, const Ten& xin) {
            Ten qkv = ln_proj(xin, name == "attn1" ? "norm1" : "norm2", p + "." + name + ".qkv", 3 * inner, 0);
            Ten o = make(M, inner, adt());
            if (spatial) attention(qkv, cols(qkv, inner, inner), cols(qkv, 2 * inner, inner), o, B * T, heads, H * W, H * W, 1, scale, 0);
            else tattention(qkv, cols(qkv, inner, inner), cols(qkv, 2 * inner, inner), o, B, T, H * W, heads, scale);
            qkv = Ten();
            return linear(o, p + "." + name + ".to_out.0", xin, rs);
        };
        x = self_attn("attn1", x);
        if (do_dup) {
            x = dup(x);
            B *= 2; M *= 2;
        }
        if (spatial) {
            Ten q = ln_proj(x, "norm2", p + ".attn2.to_q", inner, 0);
            const int kc = u->cfg.context_dim;
            Ten kv = gemm(ctx.text, wt(p + ".attn2.kv.w", 2 * inner, kc, DS_F16), nullptr, Ten(), ctx.text.rows, 2 * inner, kc, 0);
            Ten o = make(M, inner, adt());
            attention(q, kv, cols(kv, inner, inner), o, B * T, heads, H * W, ctx.ltxt, T, scale, 0);
            if (ctx.img && u->hasP(p + ".attn2.kv_ip.w")) {
                Ten kvi = gemm(ctx.img, wt(p + ".attn2.kv_ip.w", 2 * inner, kc, DS_F16), nullptr, Ten(), ctx.img.rows, 2 * inner, kc, 0);
                // out = out + 1.0 * out_ip (attention.py:117-124): second softmax over the image tokens, accumulated
                attention(q, kvi, cols(kvi, inner, inner), o, B * T, heads, H * W, ctx.limg, T, scale, 1);
            }
            x = linear(o, p + ".attn2.to_out.0", x, rs);
        } else {
            x = self_attn("attn2", x);
        }
        Ten g = ln_proj(x, "norm3", p + ".ff1", 8 * inner, DS_EPI_GEGLU);
        return linear(g, p + ".ff.net.2", x, last ? 0 : rs);
    }

    Ten transformer(Ten h, const std::string& prefix, int heads, int depth, bool spatial, Geo& geo, const Ctx& ctx, bool do_dup, Ten out) {
        const int C = h.cols;
        Ten a = spatial ? groupnorm(h, prefix + ".norm", geo.B * geo.T, geo.H * geo.W, C, 1e-6f, 0)
                        : groupnorm(h, prefix + ".norm", geo.B, geo.T * geo.H * geo.W, C, 1e-6f, 0);
        Ten x = linear(a, prefix + ".proj_in", Ten(), u->inner32 ? DS_EPI_OUT_F32 : 0);
        a = Ten();
        Geo g2 = geo;
        for (int d = 0; d < depth; ++d) {
            x = transformer_block(x, prefix + ".transformer_blocks." + std::to_string(d), heads, spatial, g2, ctx, do_dup && d == 0, d == depth - 1);
            if (do_dup && d == 0) {
                h = dup(h);
                g2.B *= 2;
            }
        }
        return linear(x, prefix + ".proj_out", h, res_epi(), out);
    }

    Ten resblock(const Ten& h, const Block& b, const Geo& geo, const Ten& emb_all, Ten out) {
        const int B = geo.B, T = geo.T, H = geo.H, W = geo.W;
        const std::string& p = b.prefix;
        const int rs = res_epi();
        const bool need_skip = b.cin != b.cout;
        Ten h16 = h, a;
        if (need_skip && h.is32()) a = groupnorm(h, p + ".in_layers.0", B * T, H * W, b.cin, 1e-5f, 1, &h16);
        else a = groupnorm(h, p + ".in_layers.0", B * T, H * W, b.cin, 1e-5f, 1);
        const int off = u->emb_off.at(p);          // read-only lookup: forwards of one handle may run on several host threads
        // "full" strict mode: the intermediates that only a GroupNorm reads (this conv-1 output, temporal convs 1-3) stay fp32 too
        const int mid = u->inner32 ? DS_EPI_OUT_F32 : 0;
        const long Mrows = (long)B * T * H * W;
        Ten st1 = fuse_gn(H * W) ? stats_table(Mrows, b.cout) : Ten();                  // conv-1 output -> out_layers GroupNorm (per frame)
        Ten h1 = conv3(a, p + ".in_layers.2", B * T, H, W, b.cin, 1, 0, Ten(), (const float*)ptr(emb_all) + off, T * H * W, u->emb_total, mid, Ten(), nullptr, nullptr, st1);
        a = Ten();
        Ten a2 = groupnorm(h1, p + ".out_layers.0", B * T, H * W, b.cout, 1e-5f, 1, nullptr, st1);
        h1 = Ten(); st1 = Ten();
        Ten skip = need_skip ? linear(h16, p + ".skip_connection", Ten(), rs) : h;
        h16 = Ten();
        Ten stx = (b.tconv && fuse_gn(T * H * W)) ? stats_table(Mrows, b.cout) : Ten();   // conv-2 output -> first temporal-conv GroupNorm (over T jointly)
        Ten h2 = conv3(a2, p + ".out_layers.3", B * T, H, W, b.cout, 1, 0, skip, nullptr, 0, 0, rs, b.tconv ? Ten() : out, nullptr, nullptr, stx);
        a2 = Ten(); skip = Ten();
        if (b.tconv) {
            Ten x = h2;
            const long M = x.rows;
            for (int i = 1; i <= 4; ++i) {
                const std::string q = p + ".temopral_conv.conv" + std::to_string(i);
                const std::string ci = std::to_string(i == 1 ? 2 : 3);
                Ten an = groupnorm(x, q + ".0", B, T * H * W, b.cout, 1e-5f, 1, nullptr, stx);
                stx = (i < 4 && fuse_gn(T * H * W)) ? stats_table(Mrows, b.cout) : Ten();   // temporal conv i -> GroupNorm of conv i + 1
                x = gemm(an, wt(q + "." + ci + ".w", b.cout, 3 * b.cout, DS_F16), (const float*)u->P(q + "." + ci + ".b"), i == 4 ? h2 : Ten(), M, b.cout, 3 * b.cout,
                         i == 4 ? rs : mid, DS_A_TCONV, b.cout, nullptr, T, H * W, INT_MAX, 0, i == 4 ? out : Ten(), stx);
            }
            h2 = x;
        }
        return h2;
    }

    int forward(const void* x, int x_dtype, const int64_t* timesteps, const void* context, int ctx_dtype, int L, int fps, int B, int T, int H, int W, int pairs,
                float* eps) {
        const ds_unet_config& c = u->cfg;
        const int mc = c.model_channels;
        const int rdt = u->strict ? DS_F32 : DS_F16;
        // ---- time (+ fps) embedding -> per-ResBlock projections in one GEMM ----
        auto tstep = [&](const int64_t* t) {
            Ten e = make(B, mc, adt());
            tr("timestep_embedding n=%d dim=%d", B, mc);
            launch("timestep_embedding", 0.0, {B, mc}, [&] { return u->wide ? ds_timestep_embedding_f32(t, (float*)ptr(e), B, mc, st) : ds_timestep_embedding(t, ptr(e), B, mc, st); });
            return e;
        };
        Ten t_emb = tstep(timesteps);
        Ten e1 = linear(t_emb, "time_embed.0", Ten(), DS_EPI_SILU);
        Ten emb = linear(e1, "time_embed.2");
        if (c.fps_cond) {
            Ten fps_t = raw((size_t)B * 8);
            launch("fill", 0.0, {B}, [&] {
                fill_i64_kernel<<<(B + 63) / 64, 64, 0, st>>>((int64_t*)ptr(fps_t), B, (int64_t)fps);
                return hipGetLastError() == hipSuccess ? DS_OK : DS_ELAUNCH;
            });
            Ten f_emb = tstep((const int64_t*)ptr(fps_t));
            Ten f1 = linear(f_emb, "fps_embedding.0", Ten(), DS_EPI_SILU);
            emb = linear(f1, "fps_embedding.2", emb);
        }
        t_emb = Ten(); e1 = Ten();
        Ten semb = make(emb.rows, emb.cols, adt());
        tr("silu n=%ld", emb.rows * emb.cols);
        launch("silu", 0.0, {(int32_t)emb.rows, emb.cols}, [&] {
            return u->wide ? ds_silu_f32((const float*)ptr(emb), (float*)ptr(semb), (size_t)emb.rows * emb.cols, st)
                           : ds_silu_f16(ptr(emb), ptr(semb), (size_t)emb.rows * emb.cols, st);
        });
        const int ted = 4 * mc;
        Ten emb_all = gemm(semb, wt("emb_all.w", u->emb_total, ted, DS_F16), (const float*)u->P("emb_all.b"), Ten(), B, u->emb_total, ted, DS_EPI_OUT_F32);
        emb = Ten(); semb = Ten();
        // ---- context: text (+ image) tokens as 2-D fp16 matrices, NOT repeated over frames ----
        Ctx ctx;
        {
            const int D = c.context_dim;
            Ten c16 = make((long)B * L, D, adt());
            tr("cast n=%ld", (long)B * L * D);
            launch("cast", 0.0, {B, L, D}, [&] {
                return u->wide ? ds_cast_to_f32(context, ctx_dtype, (float*)ptr(c16), (size_t)B * L * D, st)
                               : ds_cast_to_f16(context, ctx_dtype, ptr(c16), (size_t)B * L * D, st);
            });
            if (c.use_image_attention && L > 77) {
                ctx.ltxt = 77; ctx.limg = L - 77;
                ctx.text = make((long)B * 77, D, adt());
                ctx.img = make((long)B * (L - 77), D, adt());
                Ten all = c16;                       // B "rows" of L*D elements
                all.rows = B; all.cols = L * D; all.ld = L * D;
                Ten tx = ctx.text; tx.rows = B; tx.cols = 77 * D; tx.ld = 77 * D;
                Ten im = ctx.img; im.rows = B; im.cols = (L - 77) * D; im.ld = (L - 77) * D;
                copy_rows(tx, cols(all, 0, 77 * D));
                copy_rows(im, cols(all, 77 * D, (L - 77) * D));
            } else {
                ctx.text = c16; ctx.ltxt = L;
            }
        }
        bool shared = pairs > 0;
        const void* x_in = x;
        std::function<Ten(const std::vector<Block>&, Ten, Geo&, Ten)> run = [&](const std::vector<Block>& group, Ten h, Geo& geo, Ten out) {
            for (size_t bi = 0; bi < group.size(); ++bi) {
                const Block& b = group[bi];
                Ten o = bi + 1 == group.size() ? out : Ten();
                switch (b.kind) {
                    case Block::CONV_IN: {
                        const int nb = shared ? pairs : B;
                        Ten patches = make((long)nb * T * H * W, u->kpad_in, adt());
                        tr("im2col_in B=%d C=%d T=%d H=%d W=%d kpad=%d", nb, c.in_channels, T, H, W, u->kpad_in);
                        launch("im2col_in", 0.0, {nb, T, H, W}, [&] {
                            return u->wide ? ds_im2col_in_f32(x_in, x_dtype, (float*)ptr(patches), nb, c.in_channels, T, H, W, u->kpad_in, st)
                                           : ds_im2col_in(x_in, x_dtype, ptr(patches), nb, c.in_channels, T, H, W, u->kpad_in, st);
                        });
                        h = gemm(patches, wt(b.prefix + ".w", b.cout, u->kpad_in, DS_F16), (const float*)u->P(b.prefix + ".b"), Ten(), patches.rows, b.cout, u->kpad_in,
                                 res_epi(), DS_A_DENSE, 0, nullptr, 0, 0, INT_MAX, 0, o);
                        break;
                    }
                    case Block::RES: h = resblock(h, b, geo, emb_all, o); break;
                    case Block::ST:
                        h = transformer(h, b.prefix, b.heads, b.depth, true, geo, ctx, shared, o);
                        if (shared) { shared = false; geo.B *= 2; }
                        break;
                    case Block::TT: h = transformer(h, b.prefix, b.heads, b.depth, false, geo, ctx, false, o); break;
                    case Block::DOWN: {
                        int ho, wo;
                        h = conv3(operand(h), b.prefix + ".op", geo.B * geo.T, geo.H, geo.W, b.cin, 2, 0, Ten(), nullptr, 0, 0, res_epi(), o, &ho, &wo);
                        geo.H = ho; geo.W = wo;
                        break;
                    }
                    case Block::UP: {
                        int ho, wo;
                        h = conv3(operand(h), b.prefix + ".conv", geo.B * geo.T, geo.H, geo.W, b.cin, 1, 1, Ten(), nullptr, 0, 0, res_epi(), o, &ho, &wo);
                        geo.H = ho; geo.W = wo;
                        break;
                    }
                }
                tap(b.prefix, h, geo);
            }
            return h;
        };
        auto geo_after = [&](const std::vector<Block>& group, Geo g) {
            for (auto& b : group) {
                if (b.kind == Block::DOWN) { g.H = (g.H - 1) / 2 + 1; g.W = (g.W - 1) / 2 + 1; }
                else if (b.kind == Block::UP) { g.H *= 2; g.W *= 2; }
            }
            return g;
        };
        auto has_st = [](const std::vector<Block>& g) {
            for (auto& b : g)
                if (b.kind == Block::ST) return true;
            return false;
        };
        if (shared) {
            bool any = false;
            for (auto& g : u->inputs) any = any || has_st(g);
            if (2 * pairs != B || !any) { ds_set_error("ds_unet_forward: cfg_pairs=%d needs a batch of %d (got %d) and a SpatialTransformer in the input path", pairs, 2 * pairs, B); return DS_EINVAL; }
        }
        // torch.cat([h, hs.pop()], dim=1) (openaimodel3d.py:700-703) without the copy: every skip tensor is produced straight into the
        // right-hand columns of the buffer its decoder block reads; the decoder-side h into the left-hand columns
        struct Skip { Ten cat; int c_h; Geo geo; };
        std::vector<Skip> hs;
        Geo geo{shared ? pairs : B, T, H, W};
        Ten h;
        const size_t n_in = u->inputs.size();
        for (size_t gi = 0; gi < n_in; ++gi) {
            const auto& group = u->inputs[gi];
            const int c_h = u->cat_ch[n_in - 1 - gi].first, c_skip = u->cat_ch[n_in - 1 - gi].second;
            Geo g_out = geo_after(group, geo);
            const bool init_attn = gi == 0 && c.addition_attention;
            const bool shared_after = shared && !has_st(group);
            Geo full = g_out;
            full.B = B;
            Ten cat = make((long)full.B * full.T * full.H * full.W, c_h + c_skip, rdt);
            Ten dst = shared_after ? Ten() : cols(cat, c_h, c_skip);
            h = run(group, h, geo, init_attn ? Ten() : dst);
            if (init_attn) {
                h = transformer(h, "init_attn.0", 8, c.transformer_depth, false, geo, ctx, false, dst);
                tap("init_attn.0", h, geo);
            }
            if (h.cols != c_skip) { ds_set_error("ds_unet_forward: skip tensor of %d channels where %d were planned", h.cols, c_skip); return DS_EINVAL; }
            if (shared_after) {       // one copy of the pair batch so far: both halves of the skip rows get it
                const long half = h.rows;
                copy_rows(rows(cols(cat, c_h, c_skip), 0, half), h);
                copy_rows(rows(cols(cat, c_h, c_skip), half, half), h);
            }
            hs.push_back({cat, c_h, full});
        }
        {
            Geo ga = geo_after(u->middle, geo);
            const Geo& sg = hs.back().geo;
            if (ga.B != sg.B || ga.H != sg.H || ga.W != sg.W) { ds_set_error("ds_unet_forward: skip connection geometry mismatch (tile h/w must be divisible by 8)"); return DS_EINVAL; }
        }
        h = run(u->middle, h, geo, cols(hs.back().cat, 0, hs.back().c_h));
        for (auto& group : u->outputs) {
            Skip s = hs.back();
            hs.pop_back();
            if (h.cols != s.c_h || s.geo.H != geo.H || s.geo.W != geo.W || s.geo.B != geo.B) { ds_set_error("ds_unet_forward: decoder / skip geometry mismatch"); return DS_EINVAL; }
            if (!hs.empty()) {
                Geo ga = geo_after(group, geo);
                if (ga.H != hs.back().geo.H || ga.W != hs.back().geo.W) { ds_set_error("ds_unet_forward: skip connection geometry mismatch (tile h/w must be divisible by 8)"); return DS_EINVAL; }
            }
            h = Ten();
            Ten cat = std::move(s.cat);
            h = run(group, std::move(cat), geo, hs.empty() ? Ten() : cols(hs.back().cat, 0, hs.back().c_h));
        }
        Ten a = groupnorm(h, "out.0", B * T, H * W, mc, 1e-5f, 1);
        h = Ten();
        Ten y = conv3(a, "out.2", B * T, H, W, mc, 1, 0, Ten(), nullptr, 0, 0, DS_EPI_OUT_F32, Ten(), nullptr, nullptr);
        tr("rows_to_ncthw ydt=%d ldy=%d B=%d C=%d T=%d H=%d W=%d", y.dt, y.ld, B, c.out_channels, T, H, W);
        launch("rows_to_ncthw", 0.0, {B, T, H, W}, [&] { return ds_rows_to_ncthw(ptr(y), y.dt, y.ld, eps, DS_F32, B, c.out_channels, T, H, W, st); });
        if (arena.failed && rc == DS_OK) { ds_set_error("ds_unet_forward: workspace too small (ds_unet_workspace_bytes gives the size)"); rc = DS_EINVAL; }
        return rc;
    }
};

}  // namespace

// ------------------------------------------------------------------------------------------------ C ABI
extern "C" int ds_cast_to_f16(const void* x, int x_dtype, void* y, size_t n, void* stream) {
    DS_CHECK_ARG(x && y && n > 0, "ds_cast_to_f16: bad argument");
    DS_CHECK_ARG(x_dtype == DS_F16 || x_dtype == DS_F32, "ds_cast_to_f16: x_dtype must be DS_F16 or DS_F32");
    if (x_dtype == DS_F32) cast_f16_kernel<float><<<grid_for((long)n), 256, 0, (hipStream_t)stream>>>((const float*)x, (f16*)y, n);
    else cast_f16_kernel<f16><<<grid_for((long)n), 256, 0, (hipStream_t)stream>>>((const f16*)x, (f16*)y, n);
    DS_CHECK_LAUNCH("ds_cast_to_f16");
    return DS_OK;
}

extern "C" int ds_unet_create(const ds_unet_config* cfg, ds_unet** out) {
    DS_CHECK_ARG(cfg && out, "ds_unet_create: null argument");
    const ds_unet_config& c = *cfg;
    DS_CHECK_ARG(c.in_channels > 0 && c.out_channels > 0 && c.model_channels > 0 && c.model_channels % 64 == 0, "ds_unet_create: model_channels must be a positive multiple of 64");
    DS_CHECK_ARG(c.num_res_blocks > 0 && c.n_channel_mult > 0 && c.n_channel_mult <= 8 && c.n_attention_resolutions >= 0 && c.n_attention_resolutions <= 8,
                 "ds_unet_create: num_res_blocks / channel_mult / attention_resolutions out of range");
    DS_CHECK_ARG(c.num_head_channels == HEAD_DIM, "ds_unet_create: the attention kernels are built for head_dim 64 (the VideoCrafter configs)");
    DS_CHECK_ARG(c.transformer_depth > 0 && c.temporal_transformer_depth > 0 && c.context_dim > 0 && c.context_dim % 64 == 0, "ds_unet_create: transformer depth / context_dim");
    DS_CHECK_ARG(c.gn_from_producer == 0 || c.gn_from_producer == 1, "ds_unet_create: gn_from_producer must be 0 or 1");
    DS_CHECK_ARG(c.temporal_selfatt_only == 1, "ds_unet_create: temporal_selfatt_only must be 1 (TemporalTransformers with cross-attention to the context are not built)");
    DS_CHECK_ARG(c.residual_f32 >= 0 && c.residual_f32 <= 3, "ds_unet_create: residual_f32 must be 0 (fp16 stream), 1 (fp32 everywhere), 2 (fp32 between the blocks only) or 3 (wide operands)");
    DS_CHECK_ARG(c.residual_f32 != 3 || c.gn_from_producer == 0, "ds_unet_create: gn_from_producer is not available in the wide mode");
    for (int i = 0; i < c.n_channel_mult; ++i) DS_CHECK_ARG(c.channel_mult[i] > 0, "ds_unet_create: channel_mult[%d] = %d must be positive", i, c.channel_mult[i]);
    for (int i = 0; i < c.n_attention_resolutions; ++i) DS_CHECK_ARG(c.attention_resolutions[i] > 0, "ds_unet_create: attention_resolutions[%d] = %d must be positive", i, c.attention_resolutions[i]);
    // every check above runs before the handle exists; nothing below may leave through the C boundary as an exception
    std::unique_ptr<ds_unet> u;
    try {
        u.reset(new ds_unet());
        u->cfg = c;
        u->wide = c.residual_f32 == 3;
        u->strict = c.residual_f32 != 0;
        u->inner32 = c.residual_f32 == 1 || u->wide;
        u->fold = c.fold_layernorm != 0 && !u->inner32;    // the fold multiplies the RAW activation on the matrix cores: needs it in fp16
        u->gn_fused = c.gn_from_producer != 0;
        build_program(u.get());
        plan_pack(u.get());
    } catch (const std::exception& e) {
        ds_set_error("ds_unet_create: %s", e.what());
        return DS_EINVAL;
    }
    *out = u.release();
    return DS_OK;
}

extern "C" int ds_unet_set_hooks(ds_unet* u, ds_launch_hook launch, ds_block_tap tap, void* user) {
    DS_CHECK_ARG(u, "ds_unet_set_hooks: null handle");
    u->launch_hook = launch;
    u->block_tap = tap;
    u->hook_user = user;
    return DS_OK;
}

extern "C" int ds_copy_rows(void* dst, size_t dst_pitch_bytes, const void* src, size_t src_pitch_bytes, size_t row_bytes, size_t rows, void* stream) {
    DS_CHECK_ARG(dst && src && row_bytes > 0 && rows > 0 && dst_pitch_bytes >= row_bytes && src_pitch_bytes >= row_bytes, "ds_copy_rows: bad argument");
    if (hipMemcpy2DAsync(dst, dst_pitch_bytes, src, src_pitch_bytes, row_bytes, rows, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) {
        ds_set_error("ds_copy_rows: hipMemcpy2DAsync failed");
        return DS_ELAUNCH;
    }
    return DS_OK;
}

extern "C" int ds_unet_destroy(ds_unet* u) {
    delete u;
    return DS_OK;
}

extern "C" int ds_unet_num_weights(const ds_unet* u) { return u ? (int)u->weights.size() : DS_EINVAL; }

extern "C" int ds_unet_weight_info(const ds_unet* u, int i, const char** key, int* ndim, int64_t shape[5]) {
    DS_CHECK_ARG(u && i >= 0 && i < (int)u->weights.size(), "ds_unet_weight_info: index out of range");
    const WeightSpec& w = u->weights[i];
    if (key) *key = w.key.c_str();
    if (ndim) *ndim = w.ndim;
    if (shape) for (int d = 0; d < 5; ++d) shape[d] = d < w.ndim ? w.shape[d] : 0;
    return DS_OK;
}

extern "C" int ds_unet_load_weight(ds_unet* u, const char* key, const void* data, int dtype, const int64_t* shape, int ndim) {
    DS_CHECK_ARG(u && key && data && shape, "ds_unet_load_weight: null argument");
    DS_CHECK_ARG(dtype == DS_F32 || dtype == DS_F16, "ds_unet_load_weight: dtype must be DS_F32 or DS_F16");
    auto it = u->windex.find(key);
    DS_CHECK_ARG(it != u->windex.end(), "ds_unet_load_weight: unexpected key %s", key);
    WeightSpec& w = u->weights[it->second];
    bool same = ndim == w.ndim;
    for (int d = 0; same && d < ndim; ++d) same = shape[d] == w.shape[d];
    DS_CHECK_ARG(same, "ds_unet_load_weight: shape mismatch for %s", key);
    DS_CHECK_ARG((reinterpret_cast<uintptr_t>(data) & (dtype == DS_F32 ? 3 : 1)) == 0, "ds_unet_load_weight: misaligned data for %s", key);
    w.data = data;
    w.dtype = dtype;
    return DS_OK;
}

extern "C" size_t ds_unet_packed_bytes(const ds_unet* u) { return u ? u->packed_total : 0; }

extern "C" int ds_unet_num_packed(const ds_unet* u) { return u ? (int)u->items.size() : DS_EINVAL; }

extern "C" int ds_unet_packed_info(const ds_unet* u, int i, const char** name, size_t* offset, size_t* bytes, long* rows, int* dtype) {
    DS_CHECK_ARG(u && i >= 0 && i < (int)u->items.size(), "ds_unet_packed_info: index out of range");
    const PackItem& it = u->items[i];
    if (name) *name = it.name.c_str();
    if (offset) *offset = it.off;
    if (bytes) *bytes = it.bytes;
    if (rows) *rows = it.rows;
    if (dtype) *dtype = it.dtype;
    return DS_OK;
}

extern "C" int ds_unet_emb_offset(const ds_unet* u, const char* resblock_prefix) {
    DS_CHECK_ARG(u && resblock_prefix, "ds_unet_emb_offset: null argument");
    auto it = u->emb_off.find(resblock_prefix);
    DS_CHECK_ARG(it != u->emb_off.end(), "ds_unet_emb_offset: %s is not a ResBlock", resblock_prefix);
    return it->second;
}

extern "C" int ds_unet_pack(ds_unet* u, void* packed, size_t packed_bytes, void* stream) {
    DS_CHECK_ARG(u && packed, "ds_unet_pack: null argument");
    DS_CHECK_ARG(packed_bytes >= u->packed_total, "ds_unet_pack: buffer of %zu bytes, %zu needed", packed_bytes, u->packed_total);
    DS_CHECK_ARG((reinterpret_cast<uintptr_t>(packed) & 255) == 0, "ds_unet_pack: the packed buffer must be 256-byte aligned");
    for (auto& w : u->weights) DS_CHECK_ARG(w.data, "ds_unet_pack: missing weight %s", w.key.c_str());
    u->packed = (char*)packed;
    for (auto& it : u->items) {          // in plan order: folded projections read the norm vectors packed before them
        int rc = it.fill(u->packed + it.off, (hipStream_t)stream);
        if (rc != DS_OK) { u->packed = nullptr; return rc; }
        DS_CHECK_LAUNCH("ds_unet_pack");
    }
    // the raw pointers are not kept: a second ds_unet_pack needs a fresh ds_unet_load_weight of every key ("missing weight"
    // otherwise) instead of reading tensors the caller may have freed since
    for (auto& w : u->weights) w.data = nullptr;
    return DS_OK;
}

static int unet_check_geometry(const ds_unet* u, int B, int T, int H, int W, int L, int pairs) {
    DS_CHECK_ARG(u, "ds_unet: null handle");
    DS_CHECK_ARG(B > 0 && T > 0 && T <= 32 && H > 0 && W > 0 && L > 0 && pairs >= 0, "ds_unet: bad geometry B=%d T=%d H=%d W=%d ctx=%d cfg_pairs=%d", B, T, H, W, L, pairs);
    return DS_OK;
}

// Peak of the dry run of the launch program for one geometry, cached on the handle (the program is a function of the
// handle's configuration and the geometry only).  0: the dry run failed.
static size_t unet_peak(ds_unet* u, int B, int T, int H, int W, int ctx_tokens, int cfg_pairs) {
    const std::array<int, 6> key = {B, T, H, W, ctx_tokens, cfg_pairs};
    {
        std::lock_guard<std::mutex> g(u->peak_mu);
        auto it = u->peak_cache.find(key);
        if (it != u->peak_cache.end()) return it->second;
    }
    std::string sink;
    Prog p(u, nullptr, 0, nullptr, &sink);
    const size_t peak = p.forward(nullptr, DS_F16, nullptr, nullptr, DS_F16, ctx_tokens, 8, B, T, H, W, cfg_pairs, nullptr) == DS_OK ? p.arena.peak : 0;
    std::lock_guard<std::mutex> g(u->peak_mu);
    u->peak_cache[key] = peak;
    return peak;
}

extern "C" size_t ds_unet_workspace_bytes(ds_unet* u, int B, int T, int H, int W, int ctx_tokens, int cfg_pairs) {
    if (unet_check_geometry(u, B, T, H, W, ctx_tokens, cfg_pairs) != DS_OK) return 0;
    return unet_peak(u, B, T, H, W, ctx_tokens, cfg_pairs);
}

extern "C" long ds_unet_trace(ds_unet* u, int B, int T, int H, int W, int ctx_tokens, int cfg_pairs, char* buf, size_t buf_bytes) {
    if (unet_check_geometry(u, B, T, H, W, ctx_tokens, cfg_pairs) != DS_OK) return DS_EINVAL;
    std::string log;
    Prog p(u, nullptr, 0, nullptr, &log);
    int rc = p.forward(nullptr, DS_F16, nullptr, nullptr, DS_F16, ctx_tokens, 8, B, T, H, W, cfg_pairs, nullptr);
    if (rc != DS_OK) return rc;
    if (buf && buf_bytes > 0) {
        const size_t n = log.size() < buf_bytes - 1 ? log.size() : buf_bytes - 1;
        memcpy(buf, log.data(), n);
        buf[n] = 0;
    }
    return (long)log.size();
}

extern "C" int ds_unet_forward(ds_unet* u, const void* x, int x_dtype, const int64_t* timesteps, const void* context, int ctx_dtype, int ctx_tokens, int fps,
                               int B, int T, int H, int W, int cfg_pairs, void* workspace, size_t workspace_bytes, float* eps, void* stream) {
    int rc = unet_check_geometry(u, B, T, H, W, ctx_tokens, cfg_pairs);
    if (rc != DS_OK) return rc;
    DS_CHECK_ARG(x && timesteps && context && workspace && eps, "ds_unet_forward: null argument");
    DS_CHECK_ARG((x_dtype == DS_F16 || x_dtype == DS_F32) && (ctx_dtype == DS_F16 || ctx_dtype == DS_F32), "ds_unet_forward: dtypes must be DS_F16 or DS_F32");
    DS_CHECK_ARG(u->packed, "ds_unet_forward: ds_unet_pack has not run");
    DS_CHECK_ARG(!u->gn_fused || ds_gemm_has_stats(), "ds_unet_forward: gn_from_producer = 1 needs a library built with DS_GEMM_STATS (build variant \"gemmstats\"; the launch trace works without)");
    DS_CHECK_ARG((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "ds_unet_forward: the workspace must be 256-byte aligned");
    // refused BEFORE the first launch: a short workspace never sees a kernel (the arena's own failure path stops launching too)
    const size_t need = unet_peak(u, B, T, H, W, ctx_tokens, cfg_pairs);
    DS_CHECK_ARG(need > 0, "ds_unet_forward: the launch program could not be planned for this geometry");
    DS_CHECK_ARG(workspace_bytes >= need, "ds_unet_forward: workspace too small: %zu bytes given, %zu needed (ds_unet_workspace_bytes gives the size)", workspace_bytes, need);
    Prog p(u, (char*)workspace, workspace_bytes, (hipStream_t)stream, nullptr);
    return p.forward(x, x_dtype, timesteps, context, ctx_dtype, ctx_tokens, fps, B, T, H, W, cfg_pairs, eps);
}
