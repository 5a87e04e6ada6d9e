// Shared helpers for the gfx950 kernels of libdynscaler_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/dynscaler_hip.h"

// thread-local error message (ds_last_error)
void ds_set_error(const char* fmt, ...);

#define DS_CHECK_ARG(cond, ...)                \
    do {                                       \
        if (!(cond)) {                         \
            ds_set_error(__VA_ARGS__);         \
            return DS_EINVAL;                  \
        }                                      \
    } while (0)

#define DS_CHECK_LAUNCH(name)                                                         \
    do {                                                                              \
        hipError_t e_ = hipGetLastError();                                            \
        if (e_ != hipSuccess) {                                                       \
            ds_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));       \
            return DS_ELAUNCH;                                                        \
        }                                                                             \
    } while (0)

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// cache hints of the norm kernels (read-once / write-once streams next to GEMMs that want their operands to stay in L2):
// DS_EXP_STREAM_NT bit 0 = output stores, bit 1 = last-use input loads are non-temporal.  -0.35 % of the cfg3 step
// (profiles/r2_notes.md section 10).  NOT for the attention / concat outputs: non-temporal 8-byte stores there cost +2.6 %.
#ifndef DS_EXP_STREAM_NT
#define DS_EXP_STREAM_NT 3
#endif
#if DS_EXP_STREAM_NT & 1
#define DS_SSTORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#else
#define DS_SSTORE(ptr, val) (*(ptr) = (val))
#endif
#if DS_EXP_STREAM_NT & 2
#define DS_SLOAD(ptr) __builtin_nontemporal_load(ptr)
#else
#define DS_SLOAD(ptr) (*(ptr))
#endif

static inline int ds_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Tuning / diagnostic switches.  The PRODUCT library has none: every choice below is a compile-time constant, so no environment
// variable can change its speed or its bits.  The "tune" build variant (python -m dynamicscaler_amd.build --variant tune, loaded by
// the A/B tools through DS_HIP_LIBRARY) compiles with DS_TUNING_ENV and reads the named variable once per process instead.
#ifdef DS_TUNING_ENV
#include <stdlib.h>
#define DS_TUNE_INT(name, dflt) ([]() -> long { static const long v_ = getenv(name) ? atol(getenv(name)) : (long)(dflt); return v_; }())
#else
#define DS_TUNE_INT(name, dflt) ((long)(dflt))
#endif
