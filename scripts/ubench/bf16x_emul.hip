// Study (round 6, "what comes next"): float32 convolution sums on the bf16 matrix pipe by operand splitting.
// Every fp32-MFMA-bound kernel of the path sits at 0.70-0.80 of the fp32 matrix peak (157 TFLOP/s); the bf16 forms of gfx950 run
// 16x that rate.  A float32 x has a 24-bit significand = three bf16 pieces of 8 bits: x = hi + mid + lo EXACTLY (hi = bf16(x),
// mid = bf16(x - hi), lo = bf16(x - hi - mid)), and a product of two pieces (8 x 8 bits) is exact in float32.  So
//     a * b = sum over the 9 piece pairs            (bf16x9: every partial product exact; what differs from an fp32 fma chain is
//                                                    only where the float32 accumulator rounds)
//     drop lo*lo, lo*mid, mid*lo                    (bf16x6: relative truncation 2^-24 per product)
//     hi*hi + hi*mid + mid*hi                       (bf16x3: ~2^-16, tf32-like -- NOT a float32 substitute; shown for scale)
// at 9 / 6 / 3 v_mfma_f32_32x32x16_bf16 (32 cycles each) per 16 k against 8 v_mfma_f32_32x32x2_f32 (64 cycles each): 0.56 / 0.375 /
// 0.19 of the fp32 matrix time.
// This program measures, per variant, (1) the error against a float64 sum on convolution-shaped data (weights ~ N(0, 2/K) kaiming,
// activations = ReLU(N(0,1)), K = 576 / 2304 / 4608 = the path's layer1 / layer3 conv2 / NetFlowCoarse conv2 sums), next to the
// two float32 orders the product uses today (one fma chain; chunks of 576 products), and (2) the issue rate of the 9-MFMA block
// with operands in registers.
// hipcc --offload-arch=gfx950 -O3 bf16x_emul.hip -o bf16x_emul.bin && ./bf16x_emul.bin
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float bf16_round(float x) {          // round to nearest even onto the bf16 grid, as a float
    unsigned u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return __uint_as_float(u & 0xffff0000u);
}
__device__ __forceinline__ __bf16 as_bf16(float on_grid) {      // a float already on the bf16 grid -> its bf16 (exact)
    unsigned short h = (unsigned short)(__float_as_uint(on_grid) >> 16);
    __bf16 r;
    __builtin_memcpy(&r, &h, 2);
    return r;
}
__device__ __forceinline__ void split3(float x, __bf16& hi, __bf16& mid, __bf16& lo) {
    const float h = bf16_round(x), r1 = x - h;                   // exact (Sterbenz-like: the difference fits)
    const float m = bf16_round(r1), r2 = r1 - m;
    hi = as_bf16(h); mid = as_bf16(m); lo = as_bf16(bf16_round(r2));
}

// One wavefront per workgroup computes C (32 x 32) = A (32 x K, row-major) . B (K x 32, row-major).
// variant 0: fp32 MFMA, one fma chain over k (pairs in order)         1: fp32 MFMA, chunks of 576 products (the product's KCH form)
//         2: bf16x9, one accumulator, small terms first per 16-k block   3: bf16x6   4: bf16x3
//         5: bf16x9, TWO accumulators (hi*hi | the eight smaller terms), added once at the end
__global__ __launch_bounds__(64) void gemm_variant(const float* A, const float* B, float* C, int K, int variant) {
    const int lane = threadIdx.x, rc = lane & 31, half = lane >> 5;
    const float* a = A + (size_t)blockIdx.x * 32 * K;
    const float* b = B + (size_t)blockIdx.x * K * 32;
    f32x16 acc = {}, acc2 = {};
    if (variant <= 1) {
        f32x16 tot = {};
        for (int k = 0; k < K; k += 2) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(size_t)rc * K + k + half], b[(size_t)(k + half) * 32 + rc], acc, 0, 0, 0);
            if (variant == 1 && (k + 2) % 576 == 0) {
                for (int r = 0; r < 16; ++r) { tot[r] += acc[r]; acc[r] = 0.f; }
            }
        }
        if (variant == 1) for (int r = 0; r < 16; ++r) acc[r] += tot[r];
    } else {
        for (int k = 0; k < K; k += 16) {
            bf16x8 ah, am, al, bh, bm, bl;
            for (int i = 0; i < 8; ++i) {
                __bf16 h, m, l;
                split3(a[(size_t)rc * K + k + 8 * half + i], h, m, l); ah[i] = h; am[i] = m; al[i] = l;
                split3(b[(size_t)(k + 8 * half + i) * 32 + rc], h, m, l); bh[i] = h; bm[i] = m; bl[i] = l;
            }
            f32x16& lowacc = variant == 5 ? acc2 : acc;
            if (variant == 2 || variant == 5) {
                lowacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bl, lowacc, 0, 0, 0);
                lowacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bm, lowacc, 0, 0, 0);
                lowacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bl, lowacc, 0, 0, 0);
            }
            if (variant != 4) {
                lowacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, lowacc, 0, 0, 0);
                lowacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, lowacc, 0, 0, 0);
                lowacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, lowacc, 0, 0, 0);
            }
            lowacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, lowacc, 0, 0, 0);
            lowacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, lowacc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
        }
        if (variant == 5) for (int r = 0; r < 16; ++r) acc[r] += acc2[r];
    }
    float* c = C + (size_t)blockIdx.x * 1024;
    for (int r = 0; r < 16; ++r) c[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + rc] = acc[r];
}

// issue rate: per trip NT bf16 MFMAs of one 16-k block (operands in registers, 4 independent accumulator sets) or 8 fp32 MFMAs
template <int NT>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters) {
    float s = 0.f;
    if constexpr (NT == 0) {
        const float a = (float)threadIdx.x * 1e-3f, b = 1.0f + (float)(threadIdx.x & 7);
        f32x16 acc[4] = {};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u & 3], 0, 0, 0);
        for (int u = 0; u < 4; ++u) for (int r = 0; r < 16; ++r) s += acc[u][r];
    } else {
        bf16x8 p[3], q[3];
        for (int j = 0; j < 3; ++j) for (int i = 0; i < 8; ++i) { p[j][i] = as_bf16(bf16_round((float)(threadIdx.x + i + j))); q[j][i] = as_bf16(bf16_round(1.0f + i * j)); }
        f32x16 acc[4] = {};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < NT; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p[u % 3], q[u / 3], acc[u & 3], 0, 0, 0);
        for (int u = 0; u < 4; ++u) for (int r = 0; r < 16; ++r) s += acc[u][r];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the sustained clock under the six-product loop: s_memtime ticks of a wave around the loop / wall time of the launch (one workgroup of 4
// waves per CU = one wave per SIMD, and four per SIMD), long launches
template <int NT>
__global__ __launch_bounds__(256) void clock_kernel(float* out, unsigned long long* ticks, int iters) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    bf16x8 p[3], q[3];
    for (int j = 0; j < 3; ++j) for (int i = 0; i < 8; ++i) { p[j][i] = as_bf16(bf16_round((float)(threadIdx.x + i + j))); q[j][i] = as_bf16(bf16_round(1.0f + i * j)); }
    f32x16 acc[4] = {};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < NT; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p[u % 3], q[u / 3 % 3], acc[u & 3], 0, 0, 0);
    float s = 0.f;
    for (int u = 0; u < 4; ++u) for (int r = 0; r < 16; ++r) s += acc[u][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

static double gauss() {
    double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0);
    return sqrt(-2.0 * log(u)) * cos(6.283185307179586 * v);
}

int main() {
    const int NWG = 64;
    const char* names[6] = {"fp32_mfma_one_chain", "fp32_mfma_chunks_of_576", "bf16x9_one_accumulator", "bf16x6", "bf16x3", "bf16x9_two_accumulators"};
    printf("{\"numerics\": [\n");
    const int Ks[3] = {576, 2304, 4608};
    for (int ki = 0; ki < 3; ++ki) {
        const int K = Ks[ki];
        srand(1234 + K);
        std::vector<float> A((size_t)NWG * 32 * K), B((size_t)NWG * K * 32), C((size_t)NWG * 1024);
        const double wstd = sqrt(2.0 / K);
        for (auto& x : A) x = (float)(wstd * gauss());
        for (auto& x : B) { const double g = gauss(); x = (float)(g > 0 ? g : 0); }
        std::vector<double> ref((size_t)NWG * 1024);
        double ref_sq = 0;
        for (int w = 0; w < NWG; ++w)
            for (int m = 0; m < 32; ++m)
                for (int n = 0; n < 32; ++n) {
                    double s = 0;
                    for (int k = 0; k < K; ++k) s += (double)A[((size_t)w * 32 + m) * K + k] * (double)B[((size_t)w * K + k) * 32 + n];
                    ref[(size_t)w * 1024 + m * 32 + n] = s;
                    ref_sq += s * s;
                }
        const double ref_rms = sqrt(ref_sq / ref.size());
        float *dA, *dB, *dC;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        for (int v = 0; v < 6; ++v) {
            hipLaunchKernelGGL(gemm_variant, dim3(NWG), dim3(64), 0, 0, dA, dB, dC, K, v);
            hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
            double mx = 0, sq = 0;
            for (size_t i = 0; i < C.size(); ++i) { const double e = fabs((double)C[i] - ref[i]); mx = e > mx ? e : mx; sq += e * e; }
            printf(" {\"K\": %d, \"variant\": \"%s\", \"rms_err_over_rms\": %.3e, \"max_err_over_rms\": %.3e}%s\n", K, names[v],
                   sqrt(sq / C.size()) / ref_rms, mx / ref_rms, (ki == 2 && v == 5) ? "" : ",");
        }
        hipFree(dA); hipFree(dB); hipFree(dC);
    }
    printf("],\n\"rate\": [\n");
    float* dout;
    hipMalloc(&dout, 256 * 1024 * 4 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](auto kern, const char* name, int mfma_per_trip, double flop_per_trip_equiv) {
        const int iters = 20000, grid = 256 * 4;      // 4 workgroups of 4 waves per CU: 4 waves per SIMD
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, dout, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, dout, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double trips = (double)iters * grid * 4;                 // wave trips
        const double ns_per_trip_per_simd = ms * 1e6 / (trips / (256.0 * 4));
        printf(" {\"loop\": \"%s\", \"mfma_per_16k_block\": %d, \"ns_per_block_per_simd\": %.2f, \"fp32_equivalent_TFLOPs\": %.1f}", name, mfma_per_trip,
               ns_per_trip_per_simd, trips * flop_per_trip_equiv / (ms * 1e-3) / 1e12);
    };
    const double F = 2.0 * 32 * 32 * 16;               // float32-equivalent FLOP of one 32 x 32 x 16 block
    run(rate_kernel<0>, "fp32_mfma_32x32x2 x8", 8, F); printf(",\n");
    run(rate_kernel<9>, "bf16x9", 9, F); printf(",\n");
    run(rate_kernel<6>, "bf16x6", 6, F); printf(",\n");
    run(rate_kernel<3>, "bf16x3", 3, F); printf("\n],\n\"clock\": [\n");
    {
        unsigned long long* ticks; hipMalloc(&ticks, 4096 * 8);
        static unsigned long long host[4096];
        for (int cfg = 0; cfg < 4; ++cfg) {
            const int nt = cfg < 2 ? 6 : 8, wgs = (cfg & 1) ? 1024 : 256, it = 600000 / (wgs / 256);   // ~0.2-0.5 s per launch
            hipEvent_t c0, c1; hipEventCreate(&c0); hipEventCreate(&c1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(c0);
                if (nt == 6) hipLaunchKernelGGL(clock_kernel<6>, dim3(wgs), dim3(256), 0, 0, dout, ticks, it);
                else hipLaunchKernelGGL(clock_kernel<8>, dim3(wgs), dim3(256), 0, 0, dout, ticks, it);
                hipEventRecord(c1); hipEventSynchronize(c1);
            }
            float ms; hipEventElapsedTime(&ms, c0, c1);
            hipMemcpy(host, ticks, (size_t)wgs * 4 * 8, hipMemcpyDeviceToHost);
            unsigned long long mx = 0;
            for (int i = 0; i < wgs * 4; ++i) mx = host[i] > mx ? host[i] : mx;
            const double mfma_per_simd = (double)it * nt * (wgs / 256);
            printf(" {\"loop\": \"%d bf16 MFMAs per trip, 4 accumulators\", \"waves_per_simd\": %d, \"ms\": %.1f, \"bf16_TFLOPs\": %.0f, \"ticks_per_mfma\": %.1f, "
                   "\"GHz_if_a_tick_is_a_shader_cycle\": %.3f, \"ns_per_mfma_per_simd\": %.2f}%s\n", nt, wgs / 256, ms,
                   (double)it * nt * wgs * 4 * 32768.0 / ms / 1e9, (double)mx / ((double)it * nt), mx / (ms * 1e6), ms * 1e6 / mfma_per_simd, cfg == 3 ? "" : ",");
        }
        hipFree(ticks);
    }
    printf("]}\n");
    return 0;
}
