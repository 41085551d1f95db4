// seg.hip -- the element-wise / small-reduction kernels of the sky-segmentation forward pass (segNet/segEval.py:23-43
// on top of segNet/segModel.py:218-265): everything of SegNet.getSky that is not a convolution (those run on the convolution
// family: conv.hip incl. the dilated entry point, conv3x3.hip, conv1x1.hip), a bilinear resize (pool.hip) or a max-pool.
//   adaptive_avgpool   nn.AdaptiveAvgPool2d(s), s = 1 / 2 / 3 / 6 on the 2048-channel conv5 map   (segModel.py:227)
//   softmax_accum      scores += softmax(logits, dim=1) / 5 at the original image size             (segModel.py:258, segEval.py:34-35)
//   argmax_mask        torch.max(scores, dim=1) -> (pred == segId) as a float32 mask               (segEval.py:37-43)
// HBM-bound, one thread per output element / pixel, fp32, the summation orders of the ATen CPU kernels (row-major window
// sum; max-subtracted softmax with one pass for the sum).
#include "common.h"
#include <math.h>

namespace {

inline int grid_for(long long total, int block) {
    long long g = (total + block - 1) / block;
    const long long cap = 256LL * 32;
    return (int)(g < cap ? g : cap);
}

// ATen adaptive_avg_pool2d (CPU): window [floor(o*In/Out), ceil((o+1)*In/Out)), values summed row-major, divided by the count
__global__ __launch_bounds__(256) void adaptive_avgpool_kernel(const float* __restrict__ in, float* __restrict__ out, long long total,
                                                               int Hin, int Win, int Hout, int Wout) {
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int ow = (int)(idx % Wout);
        const long long r = idx / Wout;
        const int oh = (int)(r % Hout);
        const long long nc = r / Hout;
        const int h0 = (oh * Hin) / Hout, h1 = ((oh + 1) * Hin + Hout - 1) / Hout;
        const int w0 = (ow * Win) / Wout, w1 = ((ow + 1) * Win + Wout - 1) / Wout;
        const float* src = in + nc * Hin * Win;
        float s = 0.0f;
        for (int h = h0; h < h1; ++h)
            for (int w = w0; w < w1; ++w) s += src[(size_t)h * Win + w];
        out[idx] = s / (float)((h1 - h0) * (w1 - w0));
    }
}

// scores[n, c, p] (+)= softmax_c(logits[n, :, p]) / div      (C channel planes of HW pixels each)
__global__ __launch_bounds__(256) void softmax_accum_kernel(const float* __restrict__ logits, float* __restrict__ scores, long long NP,
                                                            int C, long long HW, float div, int accumulate) {
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < NP; idx += (long long)gridDim.x * blockDim.x) {
        const long long n = idx / HW, p = idx - n * HW;
        const float* src = logits + (size_t)n * C * HW + p;
        float* dst = scores + (size_t)n * C * HW + p;
        float mx = -INFINITY;
        for (int c = 0; c < C; ++c) { const float v = src[(size_t)c * HW]; mx = v > mx ? v : mx; }
        float sum = 0.0f;
        for (int c = 0; c < C; ++c) sum += expf(src[(size_t)c * HW] - mx);
        for (int c = 0; c < C; ++c) {
            const float pr = expf(src[(size_t)c * HW] - mx) / sum / div;
            dst[(size_t)c * HW] = accumulate ? dst[(size_t)c * HW] + pr : pr;
        }
    }
}

// pred = first arg-max over the C planes; mask = (pred == id) or its complement, float32; pred_out optional (int32)
__global__ __launch_bounds__(256) void argmax_mask_kernel(const float* __restrict__ scores, long long NP, int C, long long HW, int id,
                                                          int complement, float* __restrict__ mask, int32_t* __restrict__ pred_out) {
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < NP; idx += (long long)gridDim.x * blockDim.x) {
        const long long n = idx / HW, p = idx - n * HW;
        const float* src = scores + (size_t)n * C * HW + p;
        float best = src[0];
        int arg = 0;
        for (int c = 1; c < C; ++c) {
            const float v = src[(size_t)c * HW];
            if (v > best || (v != v && best == best)) { best = v; arg = c; }     // strict >: the first maximum; NaN wins like torch.max
        }
        const float hit = arg == id ? 1.0f : 0.0f;
        mask[idx] = complement ? 1.0f - hit : hit;
        if (pred_out) pred_out[idx] = arg;
    }
}

}  // namespace

extern "C" int rfx_adaptive_avgpool2d_f32(const float* in, float* out, int NC, int Hin, int Win, int Hout, int Wout, void* stream) {
    if (!in || !out || NC <= 0 || Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0) return RFX_E_ARG;
    const long long total = (long long)NC * Hout * Wout;
    hipLaunchKernelGGL(adaptive_avgpool_kernel, dim3(grid_for(total, 256)), dim3(256), 0, rfx_stream(stream), in, out, total, Hin, Win,
                       Hout, Wout);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

extern "C" int rfx_softmax_accum_f32(const float* logits, float* scores, int N, int C, long long HW, float div, int accumulate,
                                     void* stream) {
    if (!logits || !scores || N <= 0 || C <= 0 || HW <= 0 || !(div > 0.0f)) return RFX_E_ARG;
    const long long NP = (long long)N * HW;
    hipLaunchKernelGGL(softmax_accum_kernel, dim3(grid_for(NP, 256)), dim3(256), 0, rfx_stream(stream), logits, scores, NP, C, HW, div,
                       accumulate);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

extern "C" int rfx_argmax_mask_f32(const float* scores, int N, int C, long long HW, int id, int complement, float* mask,
                                   int32_t* pred, void* stream) {
    if (!scores || !mask || N <= 0 || C <= 0 || HW <= 0) return RFX_E_ARG;
    const long long NP = (long long)N * HW;
    hipLaunchKernelGGL(argmax_mask_kernel, dim3(grid_for(NP, 256)), dim3(256), 0, rfx_stream(stream), scores, NP, C, HW, id, complement,
                       mask, pred);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}
