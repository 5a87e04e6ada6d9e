// A non-Python host of the UNet: only include/dynscaler_hip.h + the HIP runtime.
//
//   unet_host <dir>
//     reads   <dir>/config.txt   the ds_unet_config fields as integers (field order of the header) then B T H W ctx_tokens fps cfg_pairs
//             <dir>/weights.bin  the raw fp32 tensors of the reference's state dict, concatenated in ds_unet_weight_info order
//             <dir>/x.bin (fp32 [B][C][T][H][W]), <dir>/t.bin (int64 [B]), <dir>/ctx.bin (fp32 [B][ctx_tokens][context_dim])
//     writes  <dir>/eps.bin      fp32 [B][C_out][T][H][W]
//
// What DiffusionWrapper.forward -> UNetModel.forward (lvdm/models/ddpm3d.py:702-712, openaimodel3d.py:657-708) looks like from C++:
// create -> load_weight x N -> pack -> workspace_bytes -> forward.  tests/test_gpu_unet_c.py builds it with hipcc, runs it on the toy
// UNet and compares eps.bin bit for bit with the Python binding's result.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <string>
#include <vector>

#include "dynscaler_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_DS(x) do { int r_ = (x); if (r_ != DS_OK) { fprintf(stderr, "%s failed (%d): %s\n", #x, r_, ds_last_error()); return 3; } } while (0)

static bool read_file(const std::string& path, std::vector<char>& out) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    out.resize((size_t)n);
    size_t got = n ? fread(out.data(), 1, (size_t)n, f) : 0;
    fclose(f);
    return got == (size_t)n;
}

int main(int argc, char** argv) {
    if (argc != 2) { fprintf(stderr, "usage: %s <dir>\n", argv[0]); return 1; }
    const std::string dir = argv[1];
    FILE* cf = fopen((dir + "/config.txt").c_str(), "r");
    if (!cf) { fprintf(stderr, "no config.txt\n"); return 1; }
    ds_unet_config cfg;
    int32_t* fields = reinterpret_cast<int32_t*>(&cfg);
    for (size_t i = 0; i < sizeof(cfg) / sizeof(int32_t); ++i)
        if (fscanf(cf, "%d", &fields[i]) != 1) { fprintf(stderr, "config.txt: too few fields\n"); return 1; }
    int B, T, H, W, L, fps, pairs;
    if (fscanf(cf, "%d %d %d %d %d %d %d", &B, &T, &H, &W, &L, &fps, &pairs) != 7) { fprintf(stderr, "config.txt: geometry missing\n"); return 1; }
    fclose(cf);

    ds_unet* u = nullptr;
    CHECK_DS(ds_unet_create(&cfg, &u));
    std::vector<char> wbin, xbin, tbin, cbin;
    if (!read_file(dir + "/weights.bin", wbin) || !read_file(dir + "/x.bin", xbin) || !read_file(dir + "/t.bin", tbin) || !read_file(dir + "/ctx.bin", cbin)) {
        fprintf(stderr, "missing input file\n");
        return 1;
    }
    void *d_w, *d_x, *d_t, *d_c;
    CHECK_HIP(hipMalloc(&d_w, wbin.size()));
    CHECK_HIP(hipMemcpy(d_w, wbin.data(), wbin.size(), hipMemcpyHostToDevice));
    size_t off = 0;
    const int nw = ds_unet_num_weights(u);
    for (int i = 0; i < nw; ++i) {
        const char* key; int nd; int64_t shape[5];
        CHECK_DS(ds_unet_weight_info(u, i, &key, &nd, shape));
        size_t n = 1;
        for (int d = 0; d < nd; ++d) n *= (size_t)shape[d];
        if (off + n * 4 > wbin.size()) { fprintf(stderr, "weights.bin too short at %s\n", key); return 1; }
        CHECK_DS(ds_unet_load_weight(u, key, (char*)d_w + off, DS_F32, shape, nd));
        off += n * 4;
    }
    if (off != wbin.size()) { fprintf(stderr, "weights.bin has %zu bytes, the model takes %zu\n", wbin.size(), off); return 1; }
    const size_t nb = ds_unet_packed_bytes(u);
    void* packed;
    CHECK_HIP(hipMalloc(&packed, nb));
    hipStream_t st;
    CHECK_HIP(hipStreamCreate(&st));
    CHECK_DS(ds_unet_pack(u, packed, nb, st));
    CHECK_HIP(hipStreamSynchronize(st));
    CHECK_HIP(hipFree(d_w));                         // the raw tensors are not needed after the packing

    CHECK_HIP(hipMalloc(&d_x, xbin.size()));  CHECK_HIP(hipMemcpy(d_x, xbin.data(), xbin.size(), hipMemcpyHostToDevice));
    CHECK_HIP(hipMalloc(&d_t, tbin.size()));  CHECK_HIP(hipMemcpy(d_t, tbin.data(), tbin.size(), hipMemcpyHostToDevice));
    CHECK_HIP(hipMalloc(&d_c, cbin.size()));  CHECK_HIP(hipMemcpy(d_c, cbin.data(), cbin.size(), hipMemcpyHostToDevice));
    const size_t ws = ds_unet_workspace_bytes(u, B, T, H, W, L, pairs);
    if (!ws) { fprintf(stderr, "ds_unet_workspace_bytes: %s\n", ds_last_error()); return 3; }
    void* scratch;
    CHECK_HIP(hipMalloc(&scratch, ws));
    const size_t n_eps = (size_t)B * cfg.out_channels * T * H * W;
    float* d_eps;
    CHECK_HIP(hipMalloc((void**)&d_eps, n_eps * 4));
    CHECK_DS(ds_unet_forward(u, d_x, DS_F32, (const int64_t*)d_t, d_c, DS_F32, L, fps, B, T, H, W, pairs, scratch, ws, d_eps, st));
    CHECK_HIP(hipStreamSynchronize(st));
    std::vector<float> eps(n_eps);
    CHECK_HIP(hipMemcpy(eps.data(), d_eps, n_eps * 4, hipMemcpyDeviceToHost));
    FILE* of = fopen((dir + "/eps.bin").c_str(), "wb");
    if (!of || fwrite(eps.data(), 4, n_eps, of) != n_eps) { fprintf(stderr, "cannot write eps.bin\n"); return 1; }
    fclose(of);
    double s = 0.0;
    for (float v : eps) s += (double)v * v;
    printf("unet_host: %d weights, packed %.1f MB, workspace %.1f MB, eps[%zu] sum of squares %.6e\n", nw, nb / 1e6, ws / 1e6, n_eps, s);
    CHECK_DS(ds_unet_destroy(u));
    return 0;
}
