// Experiment: does `buffer_load_dwordx4 ... lds` (LDS-DMA) zero-fill LDS for out-of-range lanes?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef _Float16 f16;
__global__ void k(const f16* a, f16* out, int nbytes) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2048];
    for (int i = threadIdx.x; i < 512; i += 64) ((unsigned*)lds)[i] = 0x3C003C00u;  // fill with 1.0h
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(a), 0, nbytes, 0x00020000);
    unsigned off = threadIdx.x * 16;
    if (threadIdx.x & 1) off = 0xFFFFFFF0u;          // odd lanes out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, off, 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) out[i] = ((f16*)lds)[i];
}
int main() {
    const int n = 64 * 8;
    std::vector<f16> h(n);
    for (int i = 0; i < n; ++i) h[i] = (f16)(2.0f + i);
    f16 *d, *o;
    hipMalloc(&d, n * 2); hipMalloc(&o, 1024 * 2);
    hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o, n * 2);
    std::vector<f16> r(1024);
    hipMemcpy(r.data(), o, 2048, hipMemcpyDeviceToHost);
    int ok_even = 1, zero_odd = 1, keep_odd = 1;
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 8; ++j) {
            float v = (float)r[l * 8 + j];
            if (!(l & 1)) ok_even &= (v == 2.0f + l * 8 + j);
            else { zero_odd &= (v == 0.0f); keep_odd &= (v == 1.0f); }
        }
    printf("even lanes loaded correctly: %d | odd (OOB) lanes zero-filled: %d | odd lanes left untouched: %d | beyond 1KB untouched: %d\n",
           ok_even, zero_odd, keep_odd, (float)r[600] == 1.0f);
    return 0;
}
