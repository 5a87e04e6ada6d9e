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

static inline int ds_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
