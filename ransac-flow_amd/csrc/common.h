// Shared helpers for the librfx HIP translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
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

// De-phasing of the co-resident workgroups of a CU.  The conv kernels run 2 workgroups per CU that start together, do the same
// work and therefore stay in the same phase for the whole launch: both in their MFMA-bound K loops (sharing the matrix pipe),
// then both in their memory-bound epilogues (sharing the CU's load/store path) -- the two resources are used one after the
// other instead of side by side.  Speeds are equal in either constellation, so an offset between the two is PRESERVED
// (also across generations of workgroups: a slot is refilled when its workgroup ends); delaying one of the two once, in the first
// generation, is enough.  ctl = ticks of the 100 MHz wall clock (bits 0-23) | which workgroups of the first generation wait
// (bits 24+: 0 the second half, 1 odd blockIdx, 2 odd index inside the XCD, 3 the ones whose wavefront slot on the SIMD is odd).
// Timing only: results are unaffected.
// MEASURED NEGATIVE (round 3, profiles/r03_stagger_experiment.jsonl: fused Bottleneck tail 102.1 -> 100.9-102.5 TFLOP/s, k-major 1x1
// and plain 3x3 within run-to-run noise for every delay and selection mode): the phases of co-resident workgroups are either not
// aligned or, more likely, cannot overlap -- a CU's vector-memory path is in order, so the K loop's operand loads of one
// workgroup queue behind the epilogue burst of the other whatever their relative phase.  Compiled only with
// -DRFX_STAGGER_EXPERIMENT (RFX_C3F_STAGGER / RFX_C3_STAGGER / RFX_C1_STAGGER = ticks, *_MODE = selection).
#ifdef RFX_STAGGER_EXPERIMENT
__device__ __forceinline__ void rfx_stagger(unsigned ctl, unsigned bx, unsigned first_gen) {
    if (!ctl || bx >= first_gen) return;
    const unsigned ticks = ctl & 0xffffffu, mode = ctl >> 24;
    bool late;
    if (mode == 0) late = bx >= first_gen / 2;
    else if (mode == 1) late = bx & 1;
    else if (mode == 2) late = (bx >> 3) & 1;
    else late = __builtin_amdgcn_s_getreg((3 << 11) | 4) & 1;      // HW_REG_HW_ID bits 3:0 = wave slot
    if (late) {
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < (long long)ticks) __builtin_amdgcn_s_sleep(16);
    }
}
static inline unsigned rfx_stagger_env(const char* ticks_var, const char* mode_var) {
    const char* t = getenv(ticks_var);
    const char* m = getenv(mode_var);
    const unsigned ticks = t ? (unsigned)atoi(t) & 0xffffffu : 0u;
    return ticks ? (ticks | ((m ? (unsigned)atoi(m) : 0u) << 24)) : 0u;
}
#else
__device__ __forceinline__ void rfx_stagger(unsigned, unsigned, unsigned) {}
static inline unsigned rfx_stagger_env(const char*, const char*) { return 0u; }
#endif
