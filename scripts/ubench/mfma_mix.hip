// Micro-benchmark: what does each ingredient of a conv main loop cost the fp32 matrix pipe?
// 256 threads = 4 wavefronts, 2 workgroups per CU, 64 x v_mfma_f32_32x32x2_f32 per wavefront and iteration (a 128x128x32
// tile step), plus, selectable:
//   RD  1: the 32 ds_read2st64_b32 operand fetches of the k-major layout (conv1x1.hip), one 16-MFMA chunk ahead, feeding the MFMAs
//       2: the same reads into registers the MFMAs do not use
//   GL  1: 8 global_load_dwordx4 per thread (the 32 KB tile of the next step, L2 resident) into registers
//       2: the same 32 KB as 8 global_load_lds_dwordx4 per wavefront (LDS-DMA, no VGPR round trip)
//   WR  1: 8 ds_write_b128 per thread (registers -> LDS)
//   BAR 1: one workgroup barrier per iteration
// hipcc --offload-arch=gfx950 -O3 mfma_mix.hip -o mfma_mix.bin && ./mfma_mix.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// ds_read2st64_b32: two dwords 64*OFF0 and 64*OFF1 dwords from the lane address -- rows 2*kk and 2*kk+2 of one operand
// column (a k-major tile row is 128 floats), i.e. the same operand for two consecutive k-pairs, from ONE base register.
template <int OFF0, int OFF1>
__device__ __forceinline__ void rd2(float& d0, float& d1, unsigned addr) {
    float2 v;
    asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(addr), "n"(OFF0), "n"(OFF1));
    d0 = v.x; d1 = v.y;
}
template <int K>
__device__ __forceinline__ void wait16(float (&a)[16]) {
    asm volatile("s_waitcnt lgkmcnt(%16)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]),
                   "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])
                 : "n"(K));
}
// chunk C of a step: 4 k-pairs x (2 A + 2 B) operands = 8 ds_read2st64_b32; v[e*4 + {0,1,2,3}] = A0, A1, B0, B1 of k-pair e
template <int C>
__device__ __forceinline__ void read_chunk(float (&v)[16], unsigned aaddr, unsigned baddr) {
#define RDP(e)                                                                                     \
    rd2<(C * 4 + e) * 4, (C * 4 + e) * 4 + 4>(v[e * 4 + 0], v[(e + 1) * 4 + 0], aaddr);            \
    rd2<(C * 4 + e) * 4, (C * 4 + e) * 4 + 4>(v[e * 4 + 1], v[(e + 1) * 4 + 1], aaddr + 128);      \
    rd2<(C * 4 + e) * 4, (C * 4 + e) * 4 + 4>(v[e * 4 + 2], v[(e + 1) * 4 + 2], baddr);            \
    rd2<(C * 4 + e) * 4, (C * 4 + e) * 4 + 4>(v[e * 4 + 3], v[(e + 1) * 4 + 3], baddr + 128);
    RDP(0) RDP(2)
#undef RDP
}

template <int RD, int GL, int WR, int BAR, int SCHED = 0>
__global__ __launch_bounds__(256, 2) void k(const float* __restrict__ src, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float As[2][32][128];
    __shared__ __attribute__((aligned(16))) float Bs[2][32][128];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1, lrow = lane >> 5, lcol = lane & 31;
    for (int i = t; i < 2 * 32 * 128; i += 256) { (&As[0][0][0])[i] = (float)(i % 7) * 0.01f; (&Bs[0][0][0])[i] = (float)(i % 5) * 0.02f; }
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float op[2][16];
    for (int s = 0; s < 2; ++s) for (int e = 0; e < 16; ++e) op[s][e] = 0.001f * (float)(e + lane);
    float sink[16];
    for (int e = 0; e < 16; ++e) sink[e] = 0.f;
    f32x4 ra[8], rb[8];
    for (int j = 0; j < 8; ++j) ra[j] = rb[j] = (f32x4){1e-9f, 2e-9f, 3e-9f, (float)t * 1e-9f};
    const unsigned lds_a = (unsigned)(size_t)(lptr_t)&As[0][lrow][wm * 64 + lcol];
    const unsigned lds_b = (unsigned)(size_t)(lptr_t)&Bs[0][lrow][wn * 64 + lcol];
    for (int it = 0; it < iters; ++it) {
        const int cur = it & 1;
        const unsigned aa = lds_a + cur * 32 * 128 * 4, ba = lds_b + cur * 32 * 128 * 4;
        // 32 KB tiles inside an 8 MB region (L2 / MALL resident); GL = 3: inside a 4 GB region (HBM stream), else as GL = 1
        const float* tile = src + (GL == 3 ? (size_t)((blockIdx.x * 1000u + it) & 131071u) : (size_t)((blockIdx.x * 7 + it) & 255)) * 8192;
        float* wr_a = &As[cur ^ 1][0][0];
        float* wr_b = &Bs[cur ^ 1][0][0];
        if (RD == 1) read_chunk<0>(op[0], aa, ba);
        if (RD == 2) read_chunk<0>(sink, aa, ba);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            int newer = 0;   // LDS operations issued after the reads the MFMAs below wait for
            if (RD == 1 && c < 3) {
                if (c == 0) read_chunk<1>(op[1], aa, ba);
                if (c == 1) read_chunk<2>(op[0], aa, ba);
                if (c == 2) read_chunk<3>(op[1], aa, ba);
                newer += 8;
            }
            if (RD == 2 && c < 3) {
                if (c == 0) read_chunk<1>(sink, aa, ba);
                if (c == 1) read_chunk<2>(sink, aa, ba);
                if (c == 2) read_chunk<3>(sink, aa, ba);
            }
            if (SCHED == 0) {
                if (WR == 1 && c < 2) {          // 4 stores in each of the first two chunks
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        *reinterpret_cast<f32x4*>((c == 0 ? wr_a : wr_b) + (t + 256 * j) * 4) = ra[c * 4 + j];
                    newer += 4;
                }
                if ((GL == 1 || GL == 3) && c >= 2) {         // 4 loads in each of the last two chunks
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int q = (c - 2) * 4 + j;
                        ra[q] = *reinterpret_cast<const f32x4*>(tile + (t + 256 * q) * 4);   // (an asm load would hide the pending write from the register allocator)
                    }
                }
            } else if (SCHED == 1) {             // a register is re-loaded right after it was stored: chunks 0 and 1
                if (c < 2) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        *reinterpret_cast<f32x4*>((c == 0 ? wr_a : wr_b) + (t + 256 * j) * 4) = ra[c * 4 + j];
                    newer += 4;
#pragma unroll
                    for (int j = 0; j < 4; ++j) ra[c * 4 + j] = *reinterpret_cast<const f32x4*>(tile + (t + 256 * (c * 4 + j)) * 4);
                }
            } else if (SCHED == 2) {             // two stores + two re-loads in every chunk
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    *reinterpret_cast<f32x4*>((c < 2 ? wr_a : wr_b) + (t + 256 * ((c & 1) * 2 + j)) * 4) = ra[c * 2 + j];
                newer += 2;
#pragma unroll
                for (int j = 0; j < 2; ++j) ra[c * 2 + j] = *reinterpret_cast<const f32x4*>(tile + (t + 256 * (c * 2 + j)) * 4);
            } else if (SCHED == 3) {             // two register sets: the loads of this step are stored in the NEXT step
                if (c < 2) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (cur) rb[c * 4 + j] = *reinterpret_cast<const f32x4*>(tile + (t + 256 * (c * 4 + j)) * 4);
                        else     ra[c * 4 + j] = *reinterpret_cast<const f32x4*>(tile + (t + 256 * (c * 4 + j)) * 4);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        *reinterpret_cast<f32x4*>((c == 0 ? wr_a : wr_b) + (t + 256 * j) * 4) = cur ? ra[c * 4 + j] : rb[c * 4 + j];
                    newer += 4;
                }
            }
            if (GL == 2 && c < 2) {          // one wavefront moves 8 x 1 KB of the next tile
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int q = wave * 4 + j;            // 1 KB piece 0..15 of the A (c = 0) / B (c = 1) tile
                    __builtin_amdgcn_global_load_lds((gptr_t)(tile + c * 4096 + q * 256 + lane * 4), (lptr_t)((c == 0 ? wr_a : wr_b) + q * 256), 16, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (RD == 1) {
                if (newer == 0) wait16<0>(op[c & 1]);
                else if (newer == 4) wait16<4>(op[c & 1]);
                else if (newer == 2) wait16<2>(op[c & 1]);
                else if (newer == 10) wait16<10>(op[c & 1]);
                else if (newer == 8) wait16<8>(op[c & 1]);
                else wait16<12>(op[c & 1]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a0 = op[c & 1][e * 4], a1 = op[c & 1][e * 4 + 1], b0 = op[c & 1][e * 4 + 2], b1 = op[c & 1][e * 4 + 3];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (RD == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (BAR) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    for (int e = 0; e < 16; ++e) s += sink[e];
    for (int j = 0; j < 8; ++j) s += ra[j][0] + rb[j][0];
    out[blockIdx.x * 256 + t] = s;
}

template <int RD, int GL, int WR, int BAR, int SCHED = 0>
void run(const char* name, const float* src, int blocks) {
    float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    const int iters = 1000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("[%s]\n", name); fflush(stdout);
    hipLaunchKernelGGL((k<RD, GL, WR, BAR, SCHED>), dim3(blocks), dim3(256), 0, 0, src, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<RD, GL, WR, BAR, SCHED>), dim3(blocks), dim3(256), 0, 0, src, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 * iters * 64 * 4096.0;
    fflush(stdout); printf("%-64s blocks=%5d  %8.3f ms  %6.1f TFLOP/s  %.3f\n", name, blocks, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3);
    fflush(stdout); hipFree(out);
}

int main() {
    float* src; hipMalloc(&src, (size_t)131072 * 8192 * 4 + 65536); hipMemset(src, 0, (size_t)131072 * 8192 * 4 + 65536);
    for (int rep = 0; rep < 2; ++rep) {
        const int b = 1024;
        run<0, 0, 0, 0>("a  MFMA only", src, b);
        run<1, 0, 0, 0>("b  + 32 ds_read2st64_b32 feeding the MFMAs", src, b);
        run<2, 0, 0, 0>("c  + 32 ds_read2st64_b32 into unrelated registers", src, b);
        run<1, 0, 0, 1>("d  b + barrier", src, b);
        run<1, 0, 1, 1>("e  d + 8 ds_write_b128", src, b);
        run<1, 1, 0, 1>("f  d + 8 global_load_dwordx4", src, b);
        run<1, 1, 1, 1>("g  d + loads + stores (register staging, conv1x1.hip)", src, b);
        run<1, 1, 1, 1, 1>("g/1 loads re-issued right after their stores (chunks 0, 1)", src, b);
        run<1, 1, 1, 1, 2>("g/2 two stores + two loads in every chunk", src, b);
        run<1, 1, 1, 1, 3>("g/3 two register sets: a full step between load and store", src, b);
        run<1, 3, 1, 1, 3>("g2/3 as g/3, HBM stream", src, b);
        run<1, 3, 1, 1>("g2 as g, the tiles streamed from HBM (4 GB region)", src, b);
        run<1, 3, 0, 1>("f2 as f, the tiles streamed from HBM", src, b);
        run<1, 2, 0, 1>("h  d + 8 global_load_lds x4 per wave (LDS-DMA staging)", src, b);
        run<0, 2, 0, 1>("i  MFMA + LDS-DMA + barrier (no operand reads)", src, b);
        run<0, 1, 0, 0>("j  MFMA + 8 global_load_dwordx4", src, b);
        run<0, 0, 1, 0>("k  MFMA + 8 ds_write_b128", src, b);
    }
    return 0;
}
