// Shared helpers for the librfx HIP translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/rfx_api.h"

#define RFX_LAUNCH_CHECK()                         \
    do {                                           \
        hipError_t e__ = hipGetLastError();        \
        if (e__ != hipSuccess) return (int)e__;    \
    } while (0)

static inline hipStream_t rfx_stream(void* s) { return (hipStream_t)s; }

static inline int rfx_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
