// valu_rate.hip -- issue rate of the fp32 VALU forms the 7x7 correlation kernel can be written in (gfx950):
//   0: v_fmac_f32 (scalar FMA)          1: v_pk_fma_f32 (aligned pairs)
//   2: v_pk_fma_f32 with op_sel cross    3: v_pk_mov_b32      4: v_mov_b32
//   5 / 6 / 7: v_fmac_f32_dpp row_shr:1 / row_shl:1 / wave_shr:1 (the y operand taken from the neighbouring lane: round 4, the
//   correlation kernel's window quads exchanged between lanes instead of re-read from LDS)   8: 16 plain : 12 DPP, as in that loop
// Each wave runs REP x 32 independent instructions per loop iteration; grid = 256 CUs x 4 SIMDs x WPS waves.
// Prints instructions/cycle/SIMD at the measured clock-free rate (instr / s / 1024 SIMDs) and the implied TFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate.bin valu_rate.hip && ./valu_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f2 a[16], x = {1.0001f, 0.9999f}, y = {0.5f, 0.25f};
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = f2{(float)threadIdx.x + i, 1.0f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (MODE == 0) {
                    asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i].x) : "v"(x.x), "v"(y.x));
                    asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i].y) : "v"(x.y), "v"(y.y));
                } else if (MODE == 1) {
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
                } else if (MODE == 2) {
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "+v"(a[i]) : "v"(x), "v"(y));
                } else if (MODE == 3) {
                    asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "+v"(a[i]) : "v"(x), "v"(y));
                } else if (MODE == 5) {
                    asm volatile("v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(a[i].x) : "v"(x.x), "v"(y.x));
                    asm volatile("v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(a[i].y) : "v"(x.y), "v"(y.y));
                } else if (MODE == 6) {
                    asm volatile("v_fmac_f32_dpp %0, %1, %2 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(a[i].x) : "v"(x.x), "v"(y.x));
                    asm volatile("v_fmac_f32_dpp %0, %1, %2 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(a[i].y) : "v"(x.y), "v"(y.y));
                } else if (MODE == 7) {
                    asm volatile("v_fmac_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(a[i].x) : "v"(x.x), "v"(y.x));
                    asm volatile("v_fmac_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(a[i].y) : "v"(x.y), "v"(y.y));
                } else if (MODE == 8) {
                    if (i % 7 < 4) {
                        asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i].x) : "v"(x.x), "v"(y.x));
                        asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i].y) : "v"(x.y), "v"(y.y));
                    } else {
                        asm volatile("v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(a[i].x) : "v"(x.x), "v"(y.x));
                        asm volatile("v_fmac_f32_dpp %0, %1, %2 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(a[i].y) : "v"(x.y), "v"(y.y));
                    }
                } else {
                    asm volatile("v_mov_b32 %0, %1" : "+v"(a[i].x) : "v"(x.x));
                    asm volatile("v_mov_b32 %0, %1" : "+v"(a[i].y) : "v"(x.y));
                }
            }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int instr_per_slot, double flop_per_instr, float* d) {
    const int iters = 4000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int blocks = 256 * wps;          // 256-thread blocks = 4 waves = one per SIMD
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double instr = (double)blocks * 4 * iters * 64.0 * instr_per_slot;   // per wave: 64 slots x instr_per_slot
        const double per_simd_per_s = instr / (ms * 1e-3) / 1024.0;
        printf("%-28s waves/SIMD %d: %.3f G wave-instr/s/SIMD = %.2f cyc/instr @2.4GHz, %.1f TFLOP/s\n", name, wps,
               per_simd_per_s * 1e-9, 2.4e9 / per_simd_per_s, instr * 64 * flop_per_instr / (ms * 1e-3) * 1e-12);
    }
}

int main() {
    float* d; hipMalloc(&d, 256 * 4 * 256 * sizeof(float));
    run<0>("v_fmac_f32", 2, 2, d);
    run<1>("v_pk_fma_f32", 1, 4, d);
    run<2>("v_pk_fma_f32 op_sel cross", 1, 4, d);
    run<3>("v_pk_mov_b32", 1, 0, d);
    run<4>("v_mov_b32", 2, 0, d);
    run<5>("v_fmac_f32_dpp row_shr:1", 2, 2, d);
    run<6>("v_fmac_f32_dpp row_shl:1", 2, 2, d);
    run<7>("v_fmac_f32_dpp wave_shr:1", 2, 2, d);
    run<8>("v_fmac_f32 16 plain : 12 dpp", 2, 2, d);
    return 0;
}
