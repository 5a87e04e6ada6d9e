// libdynscaler_diag.so -- diagnostics that are NOT part of the product ABI (include/dynscaler_hip.h does not declare them; the product
// library does not export them).  Built next to the product by dynamicscaler_amd/build.py and loaded only by tests/ and tools/.
#include <hip/hip_runtime.h>
#include <stdio.h>

// ---------------------------------------------------------------------------------------------
// Diagnostic: poison the per-CU state a kernel must never read before writing it -- LDS and the vector / accumulator
// register files -- with NaN patterns.  tests/test_gpu_fullsize.py runs the whole UNet program with this launch in front
// of every kernel and demands bit-identical results: a kernel that read uninitialised LDS or registers (whose content
// would otherwise depend on what ran on that CU before, i.e. on timing when two hipGraphs replay concurrently) shows
// up as NaN / garbage instead of a 1-ulp run-to-run flicker.
// One 256-thread workgroup = one wave per SIMD, each owning the SIMD's whole register file (256 VGPRs + 256 AGPRs),
// with all 160 KB of the CU's LDS; many more workgroups than CUs so that every CU gets several.
// ---------------------------------------------------------------------------------------------
#define DS_R10(p, m) m(p##0) m(p##1) m(p##2) m(p##3) m(p##4) m(p##5) m(p##6) m(p##7) m(p##8) m(p##9)
#define DS_R100(p, m) DS_R10(p##0, m) DS_R10(p##1, m) DS_R10(p##2, m) DS_R10(p##3, m) DS_R10(p##4, m) \
                      DS_R10(p##5, m) DS_R10(p##6, m) DS_R10(p##7, m) DS_R10(p##8, m) DS_R10(p##9, m)
#define DS_ALL256(m) DS_R10(, m) DS_R10(1, m) DS_R10(2, m) DS_R10(3, m) DS_R10(4, m) DS_R10(5, m) DS_R10(6, m) DS_R10(7, m) \
                     DS_R10(8, m) DS_R10(9, m) DS_R100(1, m) DS_R10(20, m) DS_R10(21, m) DS_R10(22, m) DS_R10(23, m)    \
                     DS_R10(24, m) m(250) m(251) m(252) m(253) m(254) m(255)
#define DS_POISON_INSN(n) "v_mov_b32 v" #n ", %0\n\tv_accvgpr_write_b32 a" #n ", v" #n "\n\t"
#define DS_POISON_CLOB(n) "v" #n, "a" #n,

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) poison_cu_state_kernel(unsigned pattern) {
    extern __shared__ unsigned poison_lds[];
    for (int i = threadIdx.x; i < 163840 / 4; i += 256) poison_lds[i] = pattern;
    __syncthreads();
    asm volatile(DS_ALL256(DS_POISON_INSN) "s_nop 0" : : "s"(pattern) : DS_ALL256(DS_POISON_CLOB) "memory");
}

extern "C" int ds_dbg_poison_cu_state(void* stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&poison_cu_state_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        if (e != hipSuccess) {
            fprintf(stderr, "ds_dbg_poison_cu_state: hipFuncSetAttribute failed: %s\n", hipGetErrorString(e));
            return -2;
        }
        attr_set = true;
    }
    poison_cu_state_kernel<<<2048, 256, 163840, (hipStream_t)stream>>>(0x7FC07E00u);   // NaN as fp32 and as two fp16
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
