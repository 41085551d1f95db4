// preproc.hip -- image pre-processing on the device: the reference resizes every pyramid level on
// the host with PIL LANCZOS and converts with torchvision ToTensor + Normalize
// (quick_start/coarseAlignFeatMatch.py:55-57,80-110).
//   lanczos_pass   one separable pass of Pillow's fixed-point resampler (libImaging/Resample.c,
//                  ImagingResampleHorizontal/Vertical_8bpc): acc = 2^21 + sum(pixel * int32 weight), >> 22, clamp
//                  to uint8.  Integer arithmetic only -> bit-identical to Pillow given Pillow's weight tables
//                  (computed on the host by rfx/lanczos.py in Pillow's operation order).
//   u8_to_f32      uint8 HWC -> float32 CHW: x/255 (ToTensor) and optionally (x/255 - mean)/std (Normalize), the
//                  same IEEE operations in the same order as the CPU path -> bit-identical.
// HBM-bound byte work: one thread per output pixel (3 channels), consecutive threads along the output row.
#include "common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ uint8_t clip8(int v) {
    v >>= PRECISION_BITS;  // arithmetic shift, as Pillow's clip8 lookup index
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// vertical == 0: out (N, rows, outW, C) from in (N, inH, inW, C), row y reads source row y + row_off
// vertical == 1: out (N, outH, inW, C) from in (N, inH, inW, C)
// CT = compile-time channel count (3 for RGB: unrolled channel loops, accumulators in registers); 0 = runtime C.
template <int CT>
__global__ __launch_bounds__(256) void lanczos_pass_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                           long long total, int inH, int inW, int oH, int oW, int Crt,
                                                           const int32_t* __restrict__ bounds,
                                                           const int32_t* __restrict__ kk, int ksize, int vertical,
                                                           int row_off) {
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int ox = (int)(idx % oW);
        const long long r = idx / oW;
        const int oy = (int)(r % oH);
        const long long n = r / oH;
        const int C = CT ? CT : Crt;
        const uint8_t* src = in + (size_t)n * inH * inW * C;
        const int o = vertical ? oy : ox;
        const int lo = bounds[2 * o], cnt = bounds[2 * o + 1];
        const int32_t* k = kk + (size_t)o * ksize;
        int acc[4] = {1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1),
                      1 << (PRECISION_BITS - 1)};
        if (vertical) {
            for (int j = 0; j < cnt; ++j) {
                const uint8_t* p = src + ((size_t)(lo + j) * inW + ox) * C;
                const int w = k[j];
#pragma unroll
                for (int c = 0; c < 4; ++c) if (c < C) acc[c] += (int)p[c] * w;
            }
        } else {
            const uint8_t* row = src + (size_t)(oy + row_off) * inW * C;
            for (int j = 0; j < cnt; ++j) {
                const uint8_t* p = row + (size_t)(lo + j) * C;
                const int w = k[j];
#pragma unroll
                for (int c = 0; c < 4; ++c) if (c < C) acc[c] += (int)p[c] * w;
            }
        }
        uint8_t* dst = out + (size_t)idx * C;
#pragma unroll
        for (int c = 0; c < 4; ++c) if (c < C) dst[c] = clip8(acc[c]);
    }
}

__global__ __launch_bounds__(256) void u8_to_f32_kernel(const uint8_t* __restrict__ in, float* __restrict__ raw,
                                                        float* __restrict__ norm, long long total, int HW, float m0,
                                                        float m1, float m2, float s0, float s1, float s2) {
    const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long n = idx / HW;
        const int px = (int)(idx - n * HW);
        const uint8_t* p = in + (size_t)idx * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = __fdiv_rn((float)p[c], 255.0f);
            const size_t o = ((size_t)n * 3 + c) * HW + px;
            if (raw) raw[o] = v;
            if (norm) norm[o] = __fdiv_rn(__fsub_rn(v, mean[c]), sd[c]);
        }
    }
}

inline int grid_for(long long total) {
    long long g = (total + 255) / 256;
    return (int)(g < 8192 ? g : 8192);
}

}  // namespace

extern "C" int rfx_lanczos_pass_u8(const uint8_t* in, uint8_t* out, int N, int inH, int inW, int outH, int outW, int C,
                                   const int32_t* bounds, const int32_t* weights, int ksize, int vertical, int row_offset,
                                   void* stream) {
    if (!in || !out || !bounds || !weights) return RFX_E_ARG;
    if (N <= 0 || inH <= 0 || inW <= 0 || outH <= 0 || outW <= 0 || C <= 0 || C > 4 || ksize <= 0 || row_offset < 0)
        return RFX_E_ARG;
    if (vertical ? (outW != inW) : (outH + row_offset > inH)) return RFX_E_ARG;
    const long long total = (long long)N * outH * outW;
    if (C == 3)
        hipLaunchKernelGGL(lanczos_pass_kernel<3>, dim3(grid_for(total)), dim3(256), 0, rfx_stream(stream), in, out, total, inH,
                           inW, outH, outW, C, bounds, weights, ksize, vertical, row_offset);
    else
        hipLaunchKernelGGL(lanczos_pass_kernel<0>, dim3(grid_for(total)), dim3(256), 0, rfx_stream(stream), in, out, total, inH,
                           inW, outH, outW, C, bounds, weights, ksize, vertical, row_offset);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

extern "C" int rfx_u8_to_f32_chw(const uint8_t* in, float* raw, float* norm, int N, int H, int W, const float* mean3_host,
                                 const float* std3_host, void* stream) {
    if (!in || (!raw && !norm) || N <= 0 || H <= 0 || W <= 0) return RFX_E_ARG;
    if (norm && (!mean3_host || !std3_host)) return RFX_E_ARG;
    const float one[3] = {1.f, 1.f, 1.f}, zero[3] = {0.f, 0.f, 0.f};
    const float* m = mean3_host ? mean3_host : zero;
    const float* s = std3_host ? std3_host : one;
    const long long total = (long long)N * H * W;
    hipLaunchKernelGGL(u8_to_f32_kernel, dim3(grid_for(total)), dim3(256), 0, rfx_stream(stream), in, raw, norm, total,
                       H * W, m[0], m[1], m[2], s[0], s[1], s[2]);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}
