// pool.hip -- HBM-bound element-wise / small-stencil kernels of the feature nets.
//   maxpool      nn.MaxPool2d                       (model/resnet50.py:120, model/model.py:71)
//   blurpool     anti-aliased Downsample            (model/downsample.py:12-46)
//   l2norm       F.normalize(dim=1)                 (quick_start/coarseAlignFeatMatch.py:106,124)
//   flow_head    softmax-49 + tap-offset expectation (model/model.py:228-233)
//   resize       F.interpolate / F.upsample_bilinear (model/model.py:234,309; align2images.py:92)
// One thread per output element, consecutive threads along W (coalesced); grids are capped and
// grid-strided.  All arithmetic fp32 with the operation order of the ATen CPU kernels.
#include "common.h"
#include <math.h>

static inline int grid_for(long long total, int block) {
    long long g = (total + block - 1) / block;
    const long long cap = 256LL * 32;  // 256 CUs x 32 resident blocks of 256 is plenty; grid-stride the rest
    return (int)(g < cap ? g : cap);
}

__global__ __launch_bounds__(256) void maxpool2d_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                        long long total, int Hin, int Win, int Hout, int Wout,
                                                        int k, int stride, int pad) {
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int ow = (int)(idx % Wout);
        const long long r = idx / Wout;
        const int oh = (int)(r % Hout);
        const long long nc = r / Hout;
        const float* src = in + nc * Hin * Win;
        float m = -INFINITY;
        const int ih0 = oh * stride - pad, iw0 = ow * stride - pad;
        for (int i = 0; i < k; ++i) {
            const int ih = ih0 + i;
            if ((unsigned)ih >= (unsigned)Hin) continue;
            for (int j = 0; j < k; ++j) {
                const int iw = iw0 + j;
                if ((unsigned)iw >= (unsigned)Win) continue;
                const float v = src[(size_t)ih * Win + iw];
                m = (v > m || v != v) ? v : m;  // NaN propagates like ATen's max_pool2d
            }
        }
        out[idx] = m;
    }
}

extern "C" int rfx_maxpool2d_f32(const float* in, float* out, int NC, int Hin, int Win, int k, int stride,
                                 int pad, void* stream) {
    if (!in || !out || NC <= 0 || Hin <= 0 || Win <= 0 || k <= 0 || stride <= 0 || pad < 0 || pad > k / 2)
        return RFX_E_ARG;
    const int Hout = (Hin + 2 * pad - k) / stride + 1, Wout = (Win + 2 * pad - k) / stride + 1;
    if (Hout <= 0 || Wout <= 0) return RFX_E_ARG;
    const long long total = (long long)NC * Hout * Wout;
    hipLaunchKernelGGL(maxpool2d_kernel, dim3(grid_for(total, 256)), dim3(256), 0, rfx_stream(stream), in, out,
                       total, Hin, Win, Hout, Wout, k, stride, pad);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

__device__ __forceinline__ int reflect1(int i, int n) {  // ReflectionPad2d(1): -1 -> 1, n -> n-2
    return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i);
}

__global__ __launch_bounds__(256) void blurpool2d_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                         long long total, int Hin, int Win, int Hout, int Wout,
                                                         int stride) {
    // depthwise 3x3 filter [1 2 1]^T [1 2 1] / 16 accumulated in the (kh, kw) order of a direct conv
    const float w[3] = {0.25f, 0.5f, 0.25f};
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int ow = (int)(idx % Wout);
        const long long r = idx / Wout;
        const int oh = (int)(r % Hout);
        const long long nc = r / Hout;
        const float* src = in + nc * Hin * Win;
        float acc = 0.0f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int ih = reflect1(oh * stride - 1 + i, Hin);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int iw = reflect1(ow * stride - 1 + j, Win);
                acc = fmaf(src[(size_t)ih * Win + iw], w[i] * w[j], acc);
            }
        }
        out[idx] = acc;
    }
}

extern "C" int rfx_blurpool2d_f32(const float* in, float* out, int NC, int Hin, int Win, int stride, void* stream) {
    if (!in || !out || NC <= 0 || Hin < 2 || Win < 2 || stride <= 0) return RFX_E_ARG;
    const int Hout = (Hin + 2 - 3) / stride + 1, Wout = (Win + 2 - 3) / stride + 1;
    const long long total = (long long)NC * Hout * Wout;
    hipLaunchKernelGGL(blurpool2d_kernel, dim3(grid_for(total, 256)), dim3(256), 0, rfx_stream(stream), in, out,
                       total, Hin, Win, Hout, Wout, stride);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

// Fused nn.MaxPool2d(kernel 2, stride 1) + BlurPool(stride): the FeatureExtractor stem (model/model.py:71-72).
// The un-fused pair writes and re-reads a full-resolution 64-channel map (the largest tensor of the whole
// pipeline); fused, every input element is fetched once from HBM (the 4x4 input windows of neighbouring outputs
// overlap in L1/L2) and only the /2 map is written.  Same operations in the same order -> bit-identical.
__global__ __launch_bounds__(256) void maxblurpool2d_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                            long long total, int Hin, int Win, int Hout, int Wout,
                                                            int stride) {
    const float w[3] = {0.25f, 0.5f, 0.25f};
    const int Hm = Hin - 1, Wm = Win - 1;  // size of the max-pooled map
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int ow = (int)(idx % Wout);
        const long long r = idx / Wout;
        const int oh = (int)(r % Hout);
        const long long nc = r / Hout;
        const float* src = in + nc * Hin * Win;
        float acc = 0.0f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int my = reflect1(oh * stride - 1 + i, Hm);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int mx = reflect1(ow * stride - 1 + j, Wm);
                const float* q = src + (size_t)my * Win + mx;
                const float a = q[0], b = q[1], c = q[Win], d = q[Win + 1];
                float m = a;
                m = (b > m || b != b) ? b : m;
                m = (c > m || c != c) ? c : m;
                m = (d > m || d != d) ? d : m;
                acc = fmaf(m, w[i] * w[j], acc);
            }
        }
        out[idx] = acc;
    }
}

// Register-blocked form for stride 2 and Win % 4 == 0 (the FeatureExtractor stem: 64 x 480 x 640 planes, the largest
// tensor of the pipeline): one thread produces a 2x2 block of outputs.  Its 25 max-pooled values come from a 6x6
// input window that is fetched as 6 rows x 3 aligned float4 loads (18 loads instead of the 144 scalar ones the
// per-output kernel issues for the same four outputs).  Blocks that touch the reflected border take the per-output
// code.  Same operations in the same order as maxblurpool2d_kernel -> bit-identical.
__device__ __forceinline__ float maxblur_one(const float* __restrict__ src, int oh, int ow, int Win, int Hm, int Wm) {
    const float w[3] = {0.25f, 0.5f, 0.25f};
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int my = reflect1(oh * 2 - 1 + i, Hm);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int mx = reflect1(ow * 2 - 1 + j, Wm);
            const float* q = src + (size_t)my * Win + mx;
            const float a = q[0], b = q[1], c = q[Win], d = q[Win + 1];
            float m = a;
            m = (b > m || b != b) ? b : m;
            m = (c > m || c != c) ? c : m;
            m = (d > m || d != d) ? d : m;
            acc = fmaf(m, w[i] * w[j], acc);
        }
    }
    return acc;
}

__global__ __launch_bounds__(256) void maxblurpool2d_s2_block_kernel(const float* __restrict__ in,
                                                                     float* __restrict__ out, long long nblocks,
                                                                     int Hin, int Win, int Hout, int Wout) {
    const float w[3] = {0.25f, 0.5f, 0.25f};
    const int Hm = Hin - 1, Wm = Win - 1;
    const int BH = (Hout + 1) / 2, BW = (Wout + 1) / 2;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < nblocks;
         idx += (long long)gridDim.x * blockDim.x) {
        const int bx = (int)(idx % BW);
        const long long r = idx / BW;
        const int by = (int)(r % BH);
        const long long nc = r / BH;
        const float* src = in + nc * Hin * Win;
        float* dst = out + nc * Hout * Wout;
        const bool interior = by >= 1 && 4 * by + 4 <= Hin - 1 && bx >= 1 && 4 * bx + 8 <= Win && 2 * by + 1 < Hout &&
                              2 * bx + 1 < Wout;
        if (!interior) {
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int oh = 2 * by + dy, ow = 2 * bx + dx;
                    if (oh < Hout && ow < Wout) dst[(size_t)oh * Wout + ow] = maxblur_one(src, oh, ow, Win, Hm, Wm);
                }
            continue;
        }
        // input rows 4by-1 .. 4by+4, columns 4bx-4 .. 4bx+7 (window columns 4bx-1 .. 4bx+4 = v[3..8])
        float v[6][12];
        const float* p0 = src + (size_t)(4 * by - 1) * Win + (4 * bx - 4);
#pragma unroll
        for (int y = 0; y < 6; ++y)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const f32x4 t4 = *reinterpret_cast<const f32x4*>(p0 + (size_t)y * Win + 4 * q);
                v[y][4 * q] = t4[0]; v[y][4 * q + 1] = t4[1]; v[y][4 * q + 2] = t4[2]; v[y][4 * q + 3] = t4[3];
            }
        // max-pooled 5x5: M[y][x] over input (y..y+1, x..x+1), window column x -> v[.][3 + x]
        float M[5][5];
#pragma unroll
        for (int y = 0; y < 5; ++y)
#pragma unroll
            for (int x = 0; x < 5; ++x) {
                const float a = v[y][3 + x], b = v[y][4 + x], c = v[y + 1][3 + x], d = v[y + 1][4 + x];
                float m = a;
                m = (b > m || b != b) ? b : m;
                m = (c > m || c != c) ? c : m;
                m = (d > m || d != d) ? d : m;
                M[y][x] = m;
            }
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            float o[2];
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                float acc = 0.0f;
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) acc = fmaf(M[2 * dy + i][2 * dx + j], w[i] * w[j], acc);
                o[dx] = acc;
            }
            float* d2 = dst + (size_t)(2 * by + dy) * Wout + 2 * bx;
            d2[0] = o[0];
            d2[1] = o[1];
        }
    }
}

extern "C" int rfx_maxblurpool2d_f32(const float* in, float* out, int NC, int Hin, int Win, int stride, void* stream) {
    if (!in || !out || NC <= 0 || Hin < 3 || Win < 3 || stride <= 0) return RFX_E_ARG;
    const int Hm = Hin - 1, Wm = Win - 1;
    const int Hout = (Hm - 1) / stride + 1, Wout = (Wm - 1) / stride + 1;
    const long long total = (long long)NC * Hout * Wout;
    if (stride == 2 && Win % 4 == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0) {
        const long long nblocks = (long long)NC * ((Hout + 1) / 2) * ((Wout + 1) / 2);
        hipLaunchKernelGGL(maxblurpool2d_s2_block_kernel, dim3(grid_for(nblocks, 256)), dim3(256), 0, rfx_stream(stream),
                           in, out, nblocks, Hin, Win, Hout, Wout);
        RFX_LAUNCH_CHECK();
        return RFX_OK;
    }
    hipLaunchKernelGGL(maxblurpool2d_kernel, dim3(grid_for(total, 256)), dim3(256), 0, rfx_stream(stream), in, out,
                       total, Hin, Win, Hout, Wout, stride);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

// One thread per pixel; the channel loop strides by HW so that a wave reads 64 consecutive floats per
// channel (coalesced).  Two passes over C (the second one hits L2).
// F.normalize(x, dim=1) = x / max(||x||_2, 1e-12) (quick_start/coarseAlignFeatMatch.py:106,124; model/model.py PredFlowMask callers).
// Round 5: the sum of squares runs in ATen's OWN order -- ONE fused-multiply-add chain over the channels, c = 0 .. C-1 (probed
// against torch 2.10's CPU kernel, binary_kernel_reduce with NormTwoOps: `acc + x * x` contracted to an fma; bit-equal on every
// shape and thread count tried, tests/test_oracle.py::test_l2norm_order_is_atens) -- then a correctly rounded sqrt and IEEE division
// (`sqrtf` under -fhip-fp32-correctly-rounded-divide-sqrt: v_sqrt_f32 + the two-fma fix-up; hipcc lowers `__fsqrt_rn` to the BARE
// v_sqrt_f32, 1 ulp off in ~12 % of the cells -- measured: 6 % of the normalised values then differ from torch by one ulp).
// Rounds 1-4 summed four interleaved chains (more accurate: 5e-8 vs 1.8e-7 rms relative error of the norm against float64 -- and
// DIFFERENT from the reference in 81 % of the cells by up to 9.5e-7).  A norm error multiplies EVERY score of its cell coherently, so
// it outweighs the convolution round-off (whose effect on a score averages down to ~2e-8) by an order of magnitude: the reference's
// own two executions share the chain's rounding pattern almost entirely, the device's four-chain norm did not -- this, not the trunk,
// was the bulk of the device's excess arg-max near-tie flips.
__global__ __launch_bounds__(64) void l2norm_nchw_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                         long long NP, int C, int HW, long long obs, long long ocs) {
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < NP;
         p += (long long)gridDim.x * blockDim.x) {
        const long long n = p / HW;
        const int px = (int)(p - n * HW);
        const float* src = in + (size_t)n * C * HW + px;
        float* dst = out + (size_t)n * obs + px;
        float s = 0.f;
        int c = 0;
        for (; c + 15 < C; c += 16) {               // 16 loads in flight, the fma chain stays sequential in c
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = src[(size_t)(c + u) * HW];
#pragma unroll
            for (int u = 0; u < 16; ++u) s = fmaf(v[u], v[u], s);
        }
        for (; c < C; ++c) { const float a0 = src[(size_t)c * HW]; s = fmaf(a0, a0, s); }
        const float nrm = sqrtf(s);                 // correctly rounded (see the Makefile note: __fsqrt_rn is NOT)
        const float d = nrm > 1e-12f ? nrm : 1e-12f;
        for (c = 0; c < C; ++c) dst[(size_t)c * ocs] = __fdiv_rn(src[(size_t)c * HW], d);
    }
}

// Same arithmetic with four wavefronts per 64 pixels doing the LOADS (wavefront q fetches the channels c = q (mod 4), 32 loads in
// flight per thread: on the trunk's 1-5 k-pixel maps the kernel is pure load latency) while the chain itself stays ONE chain: the
// values of 128 consecutive channels go through a 32 KB LDS tile and wavefront 0 walks them in channel order.  Bit-identical to the
// kernel above.  The division pass is shared out over the four wavefronts again.
__global__ __launch_bounds__(256) void l2norm_nchw_q4_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                             long long NP, int C, int HW, long long obs, long long ocs) {
    constexpr int CB = 128;                        // channels per LDS block
    __shared__ float tile[CB][64];
    __shared__ float nrm_s[64];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const long long p = (long long)blockIdx.x * 64 + lane;
    const bool pv = p < NP;
    const long long pc = pv ? p : 0;
    const long long n = pc / HW;
    const int px = (int)(pc - n * HW);
    const float* src = in + (size_t)n * C * HW + px;
    float* dst = out + (size_t)n * obs + px;
    float s = 0.f;                                 // wavefront 0: the chain
    for (int c0 = 0; c0 < C; c0 += CB) {
        const int nb = C - c0 < CB ? C - c0 : CB;  // a multiple of 4 (launch precondition C % 4 == 0)
        float v[CB / 4];
#pragma unroll
        for (int u = 0; u < CB / 4; ++u) v[u] = (q + 4 * u < nb) ? src[(size_t)(c0 + q + 4 * u) * HW] : 0.f;
#pragma unroll
        for (int u = 0; u < CB / 4; ++u) tile[q + 4 * u][lane] = v[u];
        __syncthreads();
        if (q == 0) {
            for (int k = 0; k < nb; ++k) { const float a0 = tile[k][lane]; s = fmaf(a0, a0, s); }
        }
        __syncthreads();
    }
    if (q == 0) nrm_s[lane] = s;
    __syncthreads();
    const float nrm = sqrtf(nrm_s[lane]);
    const float d = nrm > 1e-12f ? nrm : 1e-12f;
    if (!pv) return;
    int c = q;
    for (; c + 124 < C; c += 128) {
        float v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) v[u] = src[(size_t)(c + 4 * u) * HW];
#pragma unroll
        for (int u = 0; u < 32; ++u) dst[(size_t)(c + 4 * u) * ocs] = __fdiv_rn(v[u], d);
    }
    for (; c + 28 < C; c += 32) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(c + 4 * u) * HW];
#pragma unroll
        for (int u = 0; u < 8; ++u) dst[(size_t)(c + 4 * u) * ocs] = __fdiv_rn(v[u], d);
    }
    for (; c < C; c += 4) dst[(size_t)c * ocs] = __fdiv_rn(src[(size_t)c * HW], d);
}

extern "C" int rfx_l2norm_nchw_f32(const float* in, float* out, int N, int C, int HW, long long out_batch_stride,
                                   long long out_chan_stride, void* stream) {
    if (!in || !out || N <= 0 || C <= 0 || HW <= 0 || out_batch_stride < 0 || out_chan_stride < 0) return RFX_E_ARG;
    const long long NP = (long long)N * HW;
    const long long ocs = out_chan_stride ? out_chan_stride : HW;
    const long long obs = out_batch_stride ? out_batch_stride : (long long)C * HW;
    if (C % 4 == 0 && C >= 32 && (NP + 63) / 64 <= 0x7fffffffLL) {
        hipLaunchKernelGGL(l2norm_nchw_q4_kernel, dim3((unsigned)((NP + 63) / 64)), dim3(256), 0, rfx_stream(stream), in, out,
                           NP, C, HW, obs, ocs);
        RFX_LAUNCH_CHECK();
        return RFX_OK;
    }
    hipLaunchKernelGGL(l2norm_nchw_kernel, dim3(grid_for(NP, 64)), dim3(64), 0, rfx_stream(stream), in, out, NP, C, HW,
                       obs, ocs);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

// softmax over K*K taps + expectation of the tap offsets.  One thread per pixel; the K*K logits of a
// pixel are HW apart (coalesced across the wave).  K*K <= 64 logits are kept in registers.
__global__ __launch_bounds__(256) void flow_head_kernel(const float* __restrict__ logits, float* __restrict__ flow,
                                                        long long NP, int K, int rows, int cols) {
    const int HW = rows * cols, KK = K * K, half = K / 2;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < NP;
         p += (long long)gridDim.x * blockDim.x) {
        const long long n = p / HW;
        const int px = (int)(p - n * HW);
        const float* src = logits + (size_t)n * KK * HW + px;
        float mx = -INFINITY;
        for (int q = 0; q < KK; ++q) mx = fmaxf(mx, src[(size_t)q * HW]);
        float sum = 0.f, sx = 0.f, sy = 0.f;
        for (int q = 0; q < KK; ++q) {
            const float e = expf(src[(size_t)q * HW] - mx);
            sum += e;
        }
        // p_q = e_q / sum first (as torch.softmax does), then the two expectations, accumulated in tap order
        for (int q = 0; q < KK; ++q) {
            const float pq = expf(src[(size_t)q * HW] - mx) / sum;
            const int i = q / K, j = q - i * K;
            sx += pq * (float)(j - half);
            sy += pq * (float)(i - half);
        }
        float* dst = flow + (size_t)n * 2 * HW + px;
        dst[0] = sx / (float)cols * 2.0f;
        dst[HW] = sy / (float)rows * 2.0f;
    }
}

extern "C" int rfx_flow_head_f32(const float* logits, float* flow, int N, int K, int rows, int cols, void* stream) {
    if (!logits || !flow || N <= 0 || K <= 0 || (K & 1) == 0 || rows <= 0 || cols <= 0) return RFX_E_ARG;
    const long long NP = (long long)N * rows * cols;
    hipLaunchKernelGGL(flow_head_kernel, dim3(grid_for(NP, 64)), dim3(64), 0, rfx_stream(stream), logits, flow, NP, K,
                       rows, cols);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

// ATen upsample_bilinear2d index rule.
__device__ __forceinline__ float src_index(float scale, int dst, bool align_corners) {
    if (align_corners) return scale * (float)dst;
    const float s = scale * ((float)dst + 0.5f) - 0.5f;
    return s < 0.f ? 0.f : s;
}

__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                              long long total, int Hin, int Win, int Hout, int Wout,
                                                              float sh, float sw, int align) {
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int ox = (int)(idx % Wout);
        const long long r = idx / Wout;
        const int oy = (int)(r % Hout);
        const long long nc = r / Hout;
        const float* src = in + nc * Hin * Win;
        const float fy = src_index(sh, oy, align), fx = src_index(sw, ox, align);
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < Hin - 1 ? 1 : 0), x1 = x0 + (x0 < Win - 1 ? 1 : 0);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const float hy = 1.f - ly, hx = 1.f - lx;
        const float v00 = src[(size_t)y0 * Win + x0], v01 = src[(size_t)y0 * Win + x1];
        const float v10 = src[(size_t)y1 * Win + x0], v11 = src[(size_t)y1 * Win + x1];
        // pinned operation order (ATen's Interpolate<>::eval compiled with FMA contraction: t0*w0, then += t1*w1 as an
        // fma) so that every compiled copy of this loop body rounds identically
        const float r0 = fmaf(v01, lx, __fmul_rn(v00, hx));
        const float r1 = fmaf(v11, lx, __fmul_rn(v10, hx));
        out[idx] = fmaf(r1, ly, __fmul_rn(r0, hy));
    }
}

extern "C" int rfx_resize_bilinear_f32(const float* in, float* out, int NC, int Hin, int Win, int Hout, int Wout,
                                       int align_corners, void* stream) {
    if (!in || !out || NC <= 0 || Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0) return RFX_E_ARG;
    float sh, sw;
    if (align_corners) {
        sh = Hout > 1 ? (float)(Hin - 1) / (float)(Hout - 1) : 0.f;
        sw = Wout > 1 ? (float)(Win - 1) / (float)(Wout - 1) : 0.f;
    } else {
        sh = (float)Hin / (float)Hout;
        sw = (float)Win / (float)Wout;
    }
    const long long total = (long long)NC * Hout * Wout;
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3(grid_for(total, 256)), dim3(256), 0, rfx_stream(stream), in, out,
                       total, Hin, Win, Hout, Wout, sh, sw, align_corners);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

// Row-pitch change: dst[r, 0:wd] = src[r, 0:min(ws, wd)], zero beyond (correlation on maps whose width is not a multiple
// of 4 runs on zero-padded copies: the 7x7 window's own padding is zero, so the result is unchanged).
static __global__ __launch_bounds__(256) void copy_cols_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                               long long total, int ws, int wd) {
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < total;
         p += (long long)gridDim.x * blockDim.x) {
        const long long r = p / wd;
        const int c = (int)(p - r * wd);
        dst[p] = c < ws ? src[r * ws + c] : 0.f;
    }
}

extern "C" int rfx_copy_cols_f32(const float* src, float* dst, long long rows, int w_src, int w_dst, void* stream) {
    if (!src || !dst || rows <= 0 || w_src <= 0 || w_dst <= 0) return RFX_E_ARG;
    const long long total = rows * w_dst;
    hipLaunchKernelGGL(copy_cols_kernel, dim3(grid_for(total, 256)), dim3(256), 0, rfx_stream(stream), src, dst, total, w_src,
                       w_dst);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}
