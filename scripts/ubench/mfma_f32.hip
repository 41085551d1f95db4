// Micro-benchmark: how fast can the conv main-loop *compute* structure go on its own?
//   V0  64 x v_mfma_f32_32x32x2_f32 per iteration, 4 accumulators, operands in registers
//   V1  V0 + the 16 ds_read_b128 fragment loads per iteration (LDS image as in conv.hip, no barrier)
//   V2  V1 + one __syncthreads() per iteration (4 wavefronts / workgroup)
//   V3  V2 + 8 ds_write_b128 per iteration (the LDS staging writes)
// hipcc --offload-arch=gfx950 -O3 mfma_f32.hip -o mfma_f32 && ./mfma_f32
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int V>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float As[2][2][128][16];
    __shared__ __attribute__((aligned(16))) float Bs[2][2][128][16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1, lrow = lane >> 5, lcol = lane & 31;
    for (int i = t; i < 2 * 2 * 128 * 16; i += 256) { (&As[0][0][0][0])[i] = (float)(i % 7) * 0.01f; (&Bs[0][0][0][0])[i] = (float)(i % 5) * 0.02f; }
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 af[2][2], bf[2][2];
    for (int i = 0; i < 2; ++i) for (int q = 0; q < 2; ++q) { af[i][q] = (f32x4){1.f, 2.f, 3.f, (float)lane}; bf[i][q] = (f32x4){0.5f, 0.25f, 1.f, (float)t}; }
    for (int it = 0; it < iters; ++it) {
        const int cur = it & 1;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (V >= 1) {
#pragma unroll
                for (int i = 0; i < 2; ++i) { const int m = (wm * 2 + i) * 32 + lcol;
#pragma unroll
                    for (int q = 0; q < 2; ++q) af[i][q] = *reinterpret_cast<const f32x4*>(&As[cur][lrow][m][((half * 2 + q) ^ ((m >> 2) & 3)) * 4]); }
#pragma unroll
                for (int j = 0; j < 2; ++j) { const int pl = (wn * 2 + j) * 32 + lcol;
#pragma unroll
                    for (int q = 0; q < 2; ++q) bf[j][q] = *reinterpret_cast<const f32x4*>(&Bs[cur][lrow][pl][((half * 2 + q) ^ ((pl >> 2) & 3)) * 4]); }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][q][e], bf[j][q][e], acc[i][j], 0, 0, 0);
        }
        if (V >= 3) {
            const int mc = t % 128, h = t / 128;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v = {acc[0][0][q], acc[0][1][q], acc[1][0][q], acc[1][1][q]};
                *reinterpret_cast<f32x4*>(&As[cur ^ 1][h][mc][(q ^ ((mc >> 2) & 3)) * 4]) = v * 1e-9f;
                *reinterpret_cast<f32x4*>(&Bs[cur ^ 1][h][mc][(q ^ ((mc >> 2) & 3)) * 4]) = v * 2e-9f;
            }
        }
        if (V >= 2) __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + t] = s;
}

template <int V>
void run(const char* name, int blocks) {
    float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 * iters * 64 * 4096.0;
    printf("%-44s blocks=%5d  %.3f ms  %.1f TFLOP/s\n", name, blocks, ms, flop / ms / 1e9);
    hipFree(out);
}

int main() {
    for (int blocks : {256, 512, 1024}) {
        run<0>("V0 mfma only", blocks);
        run<1>("V1 + 16 ds_read_b128", blocks);
        run<2>("V2 + barrier", blocks);
        run<3>("V3 + 8 ds_write_b128", blocks);
    }
    return 0;
}
