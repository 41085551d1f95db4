// Micro-benchmark (round 6): issue rate of every fp32-input MFMA form gfx950 has, and how much fp32 VALU
// work the SAME CU sustains next to a saturated matrix pipe.
//   A  one form per kernel, 4 independent accumulators per wave, operands in registers, 1 / 2 waves per SIMD:
//        v_mfma_f32_32x32x2_f32, 16x16x4_f32, 32x32x1_2b_f32, 16x16x1_4b_f32, 4x4x1_16b_f32  -> cycles per instruction, FLOP/clk/SIMD
//   B  both pipes: per CU, WM waves run the 32x32x2 loop and WV waves run a v_fmac_f32 loop (independent chains)
//        -> matrix TFLOP/s, vector TFLOP/s and their sum, against each alone
//   C  both pipes inside ONE wave: per loop trip 1 MFMA 32x32x2 + NV independent v_fmac_f32 (NV = 0, 8, 16, 24, 32)
//   D  the clock the chip sustains under each loop: s_memtime ticks of one wave / wall time of the launch (the 157.3 TFLOP/s peak
//        assumes 2.4 GHz; a sustained fp32 MFMA stream runs below that, MI355X_MICROARCH.md "DVFS give-back")
// hipcc --offload-arch=gfx950 -O3 mfma_forms.hip -o mfma_forms.bin && ./mfma_forms.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x32 __attribute__((ext_vector_type(32)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr double GHZ = 2.4;

template <int F>
__global__ __launch_bounds__(512) void form_kernel(float* out, int iters) {
    const float a = (float)threadIdx.x * 1e-3f, b = 1.0f + (float)(threadIdx.x & 7);
    float s = 0.f;
    if constexpr (F == 0) {          // 32x32x2: 4096 FLOP
        f32x16 acc[4] = {};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u], 0, 0, 0);
        for (int u = 0; u < 4; ++u) for (int r = 0; r < 16; ++r) s += acc[u][r];
    } else if constexpr (F == 1) {   // 16x16x4: 2048 FLOP
        f32x4 acc[4] = {};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u], 0, 0, 0);
        for (int u = 0; u < 4; ++u) for (int r = 0; r < 4; ++r) s += acc[u][r];
    } else if constexpr (F == 2) {   // 32x32x1, 2 blocks: 4096 FLOP
        f32x32 acc[2] = {};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u & 1] = __builtin_amdgcn_mfma_f32_32x32x1f32(a, b, acc[u & 1], 0, 0, 0);
        for (int u = 0; u < 2; ++u) for (int r = 0; r < 32; ++r) s += acc[u][r];
    } else if constexpr (F == 3) {   // 16x16x1, 4 blocks: 2048 FLOP
        f32x16 acc[4] = {};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, acc[u], 0, 0, 0);
        for (int u = 0; u < 4; ++u) for (int r = 0; r < 16; ++r) s += acc[u][r];
    } else {                         // 4x4x1, 16 blocks: 512 FLOP
        f32x4 acc[4] = {};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[u], 0, 0, 0);
        for (int u = 0; u < 4; ++u) for (int r = 0; r < 4; ++r) s += acc[u][r];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// A1: ONE accumulator per wave (every MFMA depends on the one before: the stem kernels' sub-tile loop), 32x32x2
__global__ __launch_bounds__(512) void chain_kernel(float* out, int iters) {
    const float a = (float)threadIdx.x * 1e-3f, b = 2.0f;
    f32x16 acc = {};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
static void launch_chain(float* o, int it) { extern int g_blocks_, g_threads_; hipLaunchKernelGGL(chain_kernel, dim3(g_blocks_), dim3(g_threads_), 0, 0, o, it); }
int g_blocks_ = 256, g_threads_ = 256;

// B: waves [0, WM) of a workgroup run MFMAs, waves [WM, WM+WV) run fmacs; one workgroup per CU
__global__ __launch_bounds__(1024) void mix_kernel(float* out, int iters, int wm, int nv_per_trip) {
    const int wave = threadIdx.x >> 6;
    float s = 0.f;
    if (wave < wm) {
        const float a = (float)threadIdx.x * 1e-3f, b = 2.0f;
        f32x16 acc[4] = {};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u], 0, 0, 0);
        for (int u = 0; u < 4; ++u) for (int r = 0; r < 16; ++r) s += acc[u][r];
    } else {
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = (float)(threadIdx.x + q);
        const float m = 1.0000001f, c = 1e-7f;
        for (int it = 0; it < iters * nv_per_trip; ++it)
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = __builtin_fmaf(v[q], m, c);
        for (int q = 0; q < 16; ++q) s += v[q];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// C: one wave issues both: 1 MFMA + NV fmacs per trip (independent chains)
template <int NV>
__global__ __launch_bounds__(512) void both_kernel(float* out, int iters) {
    const float a = (float)threadIdx.x * 1e-3f, b = 2.0f;
    f32x16 acc[4] = {};
    float v[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) v[q] = (float)(threadIdx.x + q);
    const float m = 1.0000001f, c = 1e-7f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NV; ++q) v[q] = __builtin_fmaf(v[q], m, c);
        }
    }
    float s = 0.f;
    for (int u = 0; u < 4; ++u) for (int r = 0; r < 16; ++r) s += acc[u][r];
    for (int q = 0; q < 32; ++q) s += v[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static float timed(void (*launch)(float*, int), float* out, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(out, 50);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0); launch(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

static int g_blocks, g_threads, g_wm, g_nv;
template <int F> static void launch_form(float* o, int it) { hipLaunchKernelGGL(form_kernel<F>, dim3(g_blocks), dim3(g_threads), 0, 0, o, it); }
static void launch_mix(float* o, int it) { hipLaunchKernelGGL(mix_kernel, dim3(g_blocks), dim3(g_threads), 0, 0, o, it, g_wm, g_nv); }
template <int NV> static void launch_both(float* o, int it) { hipLaunchKernelGGL(both_kernel<NV>, dim3(g_blocks), dim3(g_threads), 0, 0, o, it); }

// D: the MFMA-only / VALU-only loop with the shader clock read around it by lane 0 of every wave (max over waves taken on the host)
template <int MODE>
__global__ __launch_bounds__(256) void clock_kernel(float* out, unsigned long long* ticks, int iters) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    if constexpr (MODE == 0) {
        const float a = (float)threadIdx.x * 1e-3f, b = 2.0f;
        f32x16 acc[4] = {};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u], 0, 0, 0);
        for (int u = 0; u < 4; ++u) for (int r = 0; r < 16; ++r) s += acc[u][r];
    } else {
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = (float)(threadIdx.x + q);
        for (int it = 0; it < iters * 8; ++it)
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = __builtin_fmaf(v[q], 1.0000001f, 1e-7f);
        for (int q = 0; q < 16; ++q) s += v[q];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
    float* out; hipMalloc(&out, (size_t)4096 * 1024 * 4);
    const int iters = 4000;
    printf("# A: fp32-input MFMA forms (4 independent accumulators per wave; cycles at %.1f GHz nominal)\n", GHZ);
    const char* names[5] = {"v_mfma_f32_32x32x2_f32", "v_mfma_f32_16x16x4_f32", "v_mfma_f32_32x32x1_2b_f32", "v_mfma_f32_16x16x1_4b_f32", "v_mfma_f32_4x4x1_16b_f32"};
    const double flop[5] = {4096, 2048, 4096, 2048, 512};
    for (int wps = 1; wps <= 2; ++wps) {
        g_blocks = 256; g_threads = 256 * wps;
        void (*L[5])(float*, int) = {launch_form<0>, launch_form<1>, launch_form<2>, launch_form<3>, launch_form<4>};
        for (int f = 0; f < 5; ++f) {
            const float ms = timed(L[f], out, iters);
            const double n_per_simd = (double)iters * 4 * wps;
            const double cyc = ms * 1e-3 * GHZ * 1e9 / n_per_simd;
            printf("A %-28s waves/SIMD=%d  %.3f ms  %.1f cyc/instr/SIMD  %.1f FLOP/clk/SIMD  %.1f TFLOP/s\n", names[f], wps, ms, cyc, flop[f] / cyc,
                   flop[f] * n_per_simd * 4 * 256 / ms / 1e9);
        }
    }
    printf("# A1: one DEPENDENT accumulator chain per wave (v_mfma_f32_32x32x2_f32)\n");
    for (int wps = 1; wps <= 4; wps *= 2) {
        g_blocks_ = 256; g_threads_ = 256 * wps;
        if (g_threads_ > 512) { g_blocks_ = 256 * (wps / 2); g_threads_ = 512; }     // 4 waves/SIMD = two 512-thread workgroups per CU
        const float ms = timed(launch_chain, out, iters);
        const double n_per_simd = (double)iters * 4 * wps;
        printf("A1 waves/SIMD=%d  %.3f ms  %.1f cyc/instr/SIMD  %.1f TFLOP/s\n", wps, ms, ms * 1e-3 * GHZ * 1e9 / n_per_simd,
               4096.0 * n_per_simd * 4 * 256 / ms / 1e9);
    }
    printf("# B: separate waves on one CU: WM matrix waves (32x32x2) + WV vector waves (v_fmac_f32), one workgroup per CU\n");
    const int cfgs[][2] = {{4, 0}, {0, 4}, {0, 8}, {4, 4}, {4, 8}, {8, 8}, {4, 12}};
    for (auto& c : cfgs) {
        g_wm = c[0]; g_blocks = 256; g_threads = 64 * (c[0] + c[1]); g_nv = 8;     // 8 x 16 fmacs per 4 MFMAs: ~ equal time per trip
        const float ms = timed(launch_mix, out, iters);
        const double mt = c[0] ? (double)iters * 4 * 4096 * c[0] * 256 / ms / 1e9 : 0.0;
        const double vt = c[1] ? (double)iters * g_nv * 16 * 128 * c[1] * 256 / ms / 1e9 : 0.0;
        printf("B matrix waves=%d vector waves=%d  %.3f ms  matrix %.1f TF  vector %.1f TF  sum %.1f TF\n", c[0], c[1], ms, mt, vt, mt + vt);
    }
    printf("# C: one wave issues both: 1 x 32x32x2 + NV x v_fmac_f32 per trip, 1 and 2 waves per SIMD\n");
    for (int wps = 1; wps <= 2; ++wps) {
        g_blocks = 256; g_threads = 256 * wps;
        void (*L[5])(float*, int) = {launch_both<0>, launch_both<8>, launch_both<16>, launch_both<24>, launch_both<32>};
        const int nv[5] = {0, 8, 16, 24, 32};
        for (int f = 0; f < 5; ++f) {
            const float ms = timed(L[f], out, iters);
            const double trips = (double)iters * 4 * wps * 4 * 256;      // per chip
            const double cyc = ms * 1e-3 * GHZ * 1e9 / ((double)iters * 4 * wps);
            printf("C NV=%2d waves/SIMD=%d  %.3f ms  %.1f cyc/trip/SIMD  matrix %.1f TF + vector %.1f TF = %.1f TF\n", nv[f], wps, ms, cyc,
                   trips * 4096 / ms / 1e9, trips * nv[f] * 128 / ms / 1e9, trips * (4096 + nv[f] * 128) / ms / 1e9);
        }
    }
    printf("# D: sustained clock (s_memtime ticks of the slowest wave / wall time), 256 workgroups x 4 waves, long launches\n");
    {
        unsigned long long* ticks; hipMalloc(&ticks, 1024 * 8);
        unsigned long long host[1024];
        for (int mode = 0; mode < 2; ++mode) {
            const int it = 400000;                       // ~0.5 s: long enough for the power management to settle
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(clock_kernel<0>, dim3(256), dim3(256), 0, 0, out, ticks, it);
                else hipLaunchKernelGGL(clock_kernel<1>, dim3(256), dim3(256), 0, 0, out, ticks, it);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(host, ticks, 1024 * 8, hipMemcpyDeviceToHost);
            unsigned long long mx = 0;
            for (int i = 0; i < 1024; ++i) mx = host[i] > mx ? host[i] : mx;
            const double flop = mode == 0 ? (double)it * 4 * 4096 * 4 * 256 : (double)it * 8 * 16 * 128 * 4 * 256;
            printf("D %-12s %.1f ms  %.1f TFLOP/s  s_memtime ticks %llu -> %.3f GHz if one tick is one shader cycle (%.1f ticks per MFMA / per 16 fmac)\n",
                   mode == 0 ? "MFMA 32x32x2" : "v_fmac_f32", ms, flop / ms / 1e9, mx, mx / (ms * 1e6), (double)mx / ((double)it * (mode == 0 ? 4 : 8)));
        }
        hipFree(ticks);
    }
    hipFree(out);
    return 0;
}
