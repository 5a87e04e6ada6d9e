// Round-6 experiment (not product): what the memory system delivers for the short-K GEMM's read / write ADDRESS PATTERNS, with no
// arithmetic at all.  One persistent workgroup per CU copies a [M][320] fp16 matrix (640-byte rows) tile by tile through registers:
//   read  0 = "strided":     the 256 x 320 GEMM tile's K-step pattern -- 5 passes, each lane 16 B of a 128-byte row slice, a wave-
//                            instruction = 8 rows x 128 B at a 640-byte stride (gemm.hip's LDS-DMA pieces), passes `delay` cycles apart
//         1 = "contiguous":  the tile's 164 KB in address order, a wave-instruction = 1 KB contiguous
//   write 0 = "segments":    the 4 x 2 wave grid's epilogue -- a wave owns 64 rows x 320 B and stores 128 B + 128 B + 64 B row segments
//         1 = "contiguous":  whole rows in address order
// Build + run:  hipcc -O3 --offload-arch=gfx950 tools/exp/access_patterns.hip -o tools/exp/_build/access_patterns && tools/exp/_build/access_patterns
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int RD, int WR, int NT_STORE>
__global__ void __launch_bounds__(512, 1) copy_kernel(const char* __restrict__ x, char* __restrict__ y, int tiles, int delay) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
        const long base = (long)t * 256 * 640;
        u32x4 v[20];
        if (RD == 0) {
#pragma unroll
            for (int k = 0; k < 5; ++k) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = i * 64 + (tid >> 3), ch = tid & 7;
                    v[k * 4 + i] = *reinterpret_cast<const u32x4*>(x + base + (long)row * 640 + k * 128 + ch * 16);
                }
                if (delay > 0) {
                    const long t0 = __builtin_readcyclecounter();
                    while (__builtin_readcyclecounter() - t0 < delay) __builtin_amdgcn_s_sleep(2);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 20; ++j) {
                v[j] = *reinterpret_cast<const u32x4*>(x + base + ((long)j * 512 + tid) * 16);
                if (delay > 0 && (j & 3) == 3) {
                    const long t0 = __builtin_readcyclecounter();
                    while (__builtin_readcyclecounter() - t0 < delay) __builtin_amdgcn_s_sleep(2);
                }
            }
        }
        if (WR == 0) {
            const int wm = wave >> 1, wn = wave & 1;
            char* o = y + base + (long)(wm * 64) * 640 + wn * 320;
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const int row = s * 8 + (lane >> 3), ch = lane & 7;
                    u32x4* p = reinterpret_cast<u32x4*>(o + (long)row * 640 + g * 128 + ch * 16);
                    if (NT_STORE) __builtin_nontemporal_store(v[g * 8 + s], p); else *p = v[g * 8 + s];
                }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int row = s * 16 + (lane >> 2), ch = lane & 3;
                u32x4* p = reinterpret_cast<u32x4*>(o + (long)row * 640 + 256 + ch * 16);
                if (NT_STORE) __builtin_nontemporal_store(v[16 + s], p); else *p = v[16 + s];
            }
        } else if (WR == 2) {
            // the wave's whole width at once: 320-byte pieces, 3.2 rows per wave-instruction (per-wave strips of 16 rows x 160 columns)
            const int wm = wave >> 1, wn = wave & 1;
            char* o = y + base + (long)(wm * 64) * 640 + wn * 320;
#pragma unroll
            for (int i = 0; i < 20; ++i) {
                const int f = i * 64 + lane, row = f / 20, ch = f % 20;
                u32x4* p = reinterpret_cast<u32x4*>(o + (long)row * 640 + ch * 16);
                if (NT_STORE) __builtin_nontemporal_store(v[i], p); else *p = v[i];
            }
        } else if (WR == 3) {
            // pair-cooperative whole rows: per 16-row block a wave stores 8 full 640-byte rows (5 KB contiguous, 1 KB per instruction)
            const int wm = wave >> 1, half = wave & 1;
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    u32x4* p = reinterpret_cast<u32x4*>(y + base + (long)(wm * 64 + b * 16 + half * 8) * 640 + (i * 64 + lane) * 16);
                    if (NT_STORE) __builtin_nontemporal_store(v[b * 5 + i], p); else *p = v[b * 5 + i];
                }
        } else {
#pragma unroll
            for (int j = 0; j < 20; ++j) {
                u32x4* p = reinterpret_cast<u32x4*>(y + base + ((long)j * 512 + tid) * 16);
                if (NT_STORE) __builtin_nontemporal_store(v[j], p); else *p = v[j];
            }
        }
    }
}

template <int RD, int WR, int NTS>
static float run(const char* x, char* y, int tiles, int delay, int grid) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) copy_kernel<RD, WR, NTS><<<grid, 512>>>(x, y, tiles, delay);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int it = 20;
    for (int i = 0; i < it; ++i) copy_kernel<RD, WR, NTS><<<grid, 512>>>(x, y, tiles, delay);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / it;
}

int main(int argc, char** argv) {
    const long M = argc > 1 ? atol(argv[1]) : 655360;
    const int tiles = (int)(M / 256);
    const size_t bytes = (size_t)M * 640;
    char *x, *y;
    CK(hipMalloc(&x, bytes)); CK(hipMalloc(&y, bytes));
    CK(hipMemset(x, 1, bytes)); CK(hipMemset(y, 0, bytes));
    int ncu = 256;
    CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    printf("M %ld, %d tiles of 256 x 640 B, %d CUs; bytes moved per launch %.1f MB (read + write)\n", M, tiles, ncu, 2.0 * bytes / 1e6);
    for (int grid : {ncu, 2 * ncu}) {
        for (int delay : {0, 2500}) {
            const float a = run<0, 0, 1>(x, y, tiles, delay, grid), b = run<0, 1, 1>(x, y, tiles, delay, grid);
            const float c = run<1, 0, 1>(x, y, tiles, delay, grid), d = run<1, 1, 1>(x, y, tiles, delay, grid);
            const float e = run<0, 0, 0>(x, y, tiles, delay, grid), f = run<1, 1, 0>(x, y, tiles, delay, grid);
            auto tb = [&](float ms) { return 2.0 * bytes / (ms * 1e-3) / 1e12; };
            const float g2 = run<0, 2, 1>(x, y, tiles, delay, grid), g3 = run<0, 3, 1>(x, y, tiles, delay, grid);
            printf("grid %4d delay %5d | rd strided + wr 320-byte wave pieces %.3f ms %.2f TB/s | rd strided + wr pair-cooperative whole rows %.3f ms %.2f TB/s\n",
                   grid, delay, g2, tb(g2), g3, tb(g3));
            printf("grid %4d delay %5d | rd strided + wr segments %.3f ms %.2f TB/s | strided + contiguous %.3f ms %.2f | contiguous + segments %.3f ms %.2f | "
                   "contiguous + contiguous %.3f ms %.2f | plain stores: strided + segments %.3f ms %.2f, contiguous %.3f ms %.2f\n",
                   grid, delay, a, tb(a), b, tb(b), c, tb(c), d, tb(d), e, tb(e), f, tb(f));
        }
    }
    return 0;
}
