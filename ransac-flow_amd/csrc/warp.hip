// warp.hip -- homography grid, bilinear grid_sample and the fused fine-flow composition.
//   warp_grid     kornia HomographyWarper.warp_grid   (quick_start/align2images.py:61,65)
//   grid_sample   F.grid_sample(bilinear, zeros)      (quick_start/align2images.py:66,95,97;
//                                                       evaluation/evalHpatch/evaluation.py:25,45)
//   compose_flow  F.interpolate + grid add (+clamp) + grid_sample of the coarse grid (+ in-bounds mask)
//                                                      (quick_start/align2images.py:92-95;
//                                                       evaluation/evalHpatch/evaluation.py:40-45,51)
// All HBM-bound gathers: one thread per output pixel, consecutive threads along W; channels looped in
// the thread so the 4 corner offsets / weights are computed once per pixel.
#include "common.h"
#include <math.h>

namespace {

inline int grid_for(long long total, int block) {
    long long g = (total + block - 1) / block;
    const long long cap = 256LL * 32;
    return (int)(g < cap ? g : cap);
}

// torch.linspace(-1, 1, n)[i] as ATen computes it (symmetric about the midpoint).
__device__ __forceinline__ float lin(int i, int n) {
    if (n <= 1) return -1.0f;
    const float step = 2.0f / (float)(n - 1);
    return i < n / 2 ? -1.0f + step * (float)i : 1.0f - step * (float)(n - 1 - i);
}

__device__ __forceinline__ float unnormalize(float g, int size, int align) {
    return align ? (g + 1.0f) * 0.5f * (float)(size - 1) : ((g + 1.0f) * (float)size - 1.0f) * 0.5f;
}

struct Corners {
    int x0, y0;
    float nw, ne, sw, se;
    bool m_nw, m_ne, m_sw, m_se;
};

// ATen grid_sampler_2d (bilinear, zeros): weights from the fractional parts, out-of-image corners contribute 0.
__device__ __forceinline__ Corners corners(float gx, float gy, int Hi, int Wi, int align) {
    Corners c;
    const float ix = unnormalize(gx, Wi, align), iy = unnormalize(gy, Hi, align);
    const float xw = floorf(ix), yn = floorf(iy);
    const float w = ix - xw, e = 1.0f - w, n = iy - yn, s = 1.0f - n;
    c.nw = s * e; c.ne = s * w; c.sw = n * e; c.se = n * w;
    // NaN / huge coordinates: comparisons below are all false -> contributes 0 like ATen's masks
    const bool xin0 = xw >= 0.0f && xw <= (float)(Wi - 1), xin1 = xw + 1.0f >= 0.0f && xw + 1.0f <= (float)(Wi - 1);
    const bool yin0 = yn >= 0.0f && yn <= (float)(Hi - 1), yin1 = yn + 1.0f >= 0.0f && yn + 1.0f <= (float)(Hi - 1);
    c.m_nw = xin0 && yin0; c.m_ne = xin1 && yin0; c.m_sw = xin0 && yin1; c.m_se = xin1 && yin1;
    c.x0 = (xin0 || xin1) ? (int)xw : 0;
    c.y0 = (yin0 || yin1) ? (int)yn : 0;
    return c;
}

__global__ __launch_bounds__(256) void warp_grid_kernel(const float* __restrict__ Hm, float* __restrict__ grid,
                                                        long long total, int h, int w) {
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % w);
        const long long r = idx / w;
        const int y = (int)(r % h);
        const int b = (int)(r / h);
        const float* M = Hm + b * 9;
        const float px = lin(x, w), py = lin(y, h);
        // (x', y', z') = M (px, py, 1): k-ordered fma chain like a K=3 sgemm
        const float xs = fmaf(1.0f, M[2], fmaf(py, M[1], px * M[0]));
        const float ys = fmaf(1.0f, M[5], fmaf(py, M[4], px * M[3]));
        const float zs = fmaf(1.0f, M[8], fmaf(py, M[7], px * M[6]));
        float2 o;
        o.x = xs / zs;
        o.y = ys / zs;
        reinterpret_cast<float2*>(grid)[idx] = o;
    }
}

__global__ __launch_bounds__(256) void grid_sample_kernel(const float* __restrict__ in, const float* __restrict__ grid,
                                                          float* __restrict__ out, long long NP, int C, int Hi, int Wi,
                                                          int Ho, int Wo, int align) {
    const size_t HWi = (size_t)Hi * Wi, HWo = (size_t)Ho * Wo;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < NP;
         p += (long long)gridDim.x * blockDim.x) {
        const long long n = p / (long long)HWo;
        const size_t px = (size_t)(p - n * (long long)HWo);
        const float2 g = reinterpret_cast<const float2*>(grid)[p];
        const Corners c = corners(g.x, g.y, Hi, Wi, align);
        const float* src = in + (size_t)n * C * HWi;
        float* dst = out + (size_t)n * C * HWo + px;
        const long long o00 = (long long)c.y0 * Wi + c.x0;
        for (int ch = 0; ch < C; ++ch) {
            const float* s = src + (size_t)ch * HWi;
            float v = 0.0f;
            if (c.m_nw) v += s[o00] * c.nw;
            if (c.m_ne) v += s[o00 + 1] * c.ne;
            if (c.m_sw) v += s[o00 + Wi] * c.sw;
            if (c.m_se) v += s[o00 + Wi + 1] * c.se;
            dst[(size_t)ch * HWo] = v;
        }
    }
}

__device__ __forceinline__ float src_index(float scale, int dst) {  // align_corners=False
    const float s = scale * ((float)dst + 0.5f) - 0.5f;
    return s < 0.f ? 0.f : s;
}

__global__ __launch_bounds__(256) void compose_flow_kernel(const float* __restrict__ flowDown,
                                                           const float* __restrict__ coarse, float* __restrict__ flow12,
                                                           float* __restrict__ inb, float* __restrict__ flowUp,
                                                           long long NP, int hd, int wd, int Hc, int Wc, int H, int W,
                                                           float sh, float sw, int clampf) {
    const size_t HW = (size_t)H * W, hw = (size_t)hd * wd, HWc = (size_t)Hc * Wc;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < NP;
         p += (long long)gridDim.x * blockDim.x) {
        const long long n = p / (long long)HW;
        const int px = (int)(p - n * (long long)HW);
        const int y = px / W, x = px - y * W;
        // bilinear up-sampling of the residual flow (align_corners=False)
        const float fy = src_index(sh, y), fx = src_index(sw, x);
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < hd - 1 ? 1 : 0), x1 = x0 + (x0 < wd - 1 ? 1 : 0);
        const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
        const float* f = flowDown + (size_t)n * 2 * hw;
        float u[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float* s = f + c * hw;
            u[c] = hy * (hx * s[(size_t)y0 * wd + x0] + lx * s[(size_t)y0 * wd + x1]) +
                   ly * (hx * s[(size_t)y1 * wd + x0] + lx * s[(size_t)y1 * wd + x1]);
        }
        float gx = u[0] + lin(x, W), gy = u[1] + lin(y, H);
        if (clampf) {
            gx = fminf(fmaxf(gx, -1.0f), 1.0f);
            gy = fminf(fmaxf(gy, -1.0f), 1.0f);
        }
        if (flowUp) {
            float2 o; o.x = gx; o.y = gy;
            reinterpret_cast<float2*>(flowUp)[p] = o;
        }
        // sample the coarse grid (N,Hc,Wc,2 == 2 channels interleaved; Hc x Wc = H x W except in the KITTI driver's
        // full-resolution pass, evaluation/evalKITTI/evaluation.py:302) at (gx, gy)
        const Corners c = corners(gx, gy, Hc, Wc, 0);
        const float2* cg = reinterpret_cast<const float2*>(coarse) + (size_t)n * HWc;
        const long long o00 = (long long)c.y0 * Wc + c.x0;
        float ox = 0.f, oy = 0.f;
        if (c.m_nw) { const float2 v = cg[o00]; ox += v.x * c.nw; oy += v.y * c.nw; }
        if (c.m_ne) { const float2 v = cg[o00 + 1]; ox += v.x * c.ne; oy += v.y * c.ne; }
        if (c.m_sw) { const float2 v = cg[o00 + Wc]; ox += v.x * c.sw; oy += v.y * c.sw; }
        if (c.m_se) { const float2 v = cg[o00 + Wc + 1]; ox += v.x * c.se; oy += v.y * c.se; }
        float2 o; o.x = ox; o.y = oy;
        reinterpret_cast<float2*>(flow12)[p] = o;
        if (inb) inb[p] = (ox >= -1.0f && ox <= 1.0f && oy >= -1.0f && oy <= 1.0f) ? 1.0f : 0.0f;
    }
}

// Multi-homography merge of the offline flow assembly (evaluation/evalHpatch/getResults.py:48-61,
// evaluation/evalCorr/getResults.py:121-134, evaluation/evalKITTI/getResults.py:126-138): one thread per pixel walks the
// n homographies in order.  score_i = m12_i (* cyc_i) (* inb_i), multiplied left to right like the reference's
// tensor expression; pixel owner = 0 if score_0 >= th, else the first i >= 1 with score_i >= th (only if multiH), else 0.
__global__ __launch_bounds__(256) void merge_multi_h_kernel(const float* __restrict__ flow, const float* __restrict__ m12,
                                                           long long m12_stride, const float* __restrict__ cyc,
                                                           const float* __restrict__ inb, int n, long long HW, float th,
                                                           int multiH, float* __restrict__ flowG,
                                                           float* __restrict__ matchG, uint8_t* __restrict__ binary) {
    // grid-stride: grid_for() caps the grid, images above 2^21 pixels take several trips
    const int last = multiH ? n : 1;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += (long long)gridDim.x * blockDim.x) {
        int best = 0;
        float mbest = 0.0f;
        bool found = false;
        for (int i = 0; i < last; ++i) {
            float v = m12[(long long)i * m12_stride + p];
            if (cyc) v = v * cyc[(long long)i * HW + p];
            if (inb) v = v * inb[(long long)i * HW + p];
            if (i == 0) mbest = v;
            if (v >= th) {
                best = i;
                mbest = v;
                found = true;
                break;
            }
        }
        const float2 f = reinterpret_cast<const float2*>(flow)[(long long)best * HW + p];
        float2 o;
        o.x = fminf(fmaxf(f.x, -1.0f), 1.0f);
        o.y = fminf(fmaxf(f.y, -1.0f), 1.0f);
        reinterpret_cast<float2*>(flowG)[p] = o;
        if (matchG) matchG[p] = mbest;
        if (binary) binary[p] = found ? 1 : 0;
    }
}

// score = match12 (* cyc) (* inb), left to right: the "match" tensor of evaluation/evalKITTI/getResults.py:120, needed
// materialised only when the host-side small-component filter (:122) sits between it and the merge.
__global__ __launch_bounds__(256) void match_score_kernel(const float* __restrict__ m12, long long m12_stride,
                                                         const float* __restrict__ cyc, const float* __restrict__ inb,
                                                         long long HW, long long total, float* __restrict__ out) {
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long long)gridDim.x * blockDim.x) {
        const long long i = q / HW, p = q - i * HW;
        float v = m12[i * m12_stride + p];
        if (cyc) v = v * cyc[q];
        if (inb) v = v * inb[q];
        out[q] = v;
    }
}

}  // namespace

// predFlowCoarse's tail (model/model.py:333-340): flowGrad = || f[:, :, 1:, 1:] - f[:, :, :-1, :-1] ||_2 over the two flow
// channels, (B,1,H-1,W-1); flow = clamp(f.permute(0,2,3,1) + grid, -1, 1), (B,H,W,2).  One thread per pixel does both.
__global__ __launch_bounds__(256) void flow_grad_clamp_kernel(const float* __restrict__ f, const float* __restrict__ grid,
                                                             float* __restrict__ flowGrad, float* __restrict__ flow, int B,
                                                             int H, int W, int grid_batch) {
    const long long HW = (long long)H * W, total = (long long)B * HW;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long long)gridDim.x * blockDim.x) {
        const long long b = q / HW, p = q - b * HW;
        const int y = (int)(p / W), x = (int)(p - (long long)y * W);
        const float fx = f[(b * 2) * HW + p], fy = f[(b * 2 + 1) * HW + p];
        const float2 g = reinterpret_cast<const float2*>(grid)[(grid_batch > 1 ? b : 0) * HW + p];
        float2 o;
        o.x = fminf(fmaxf(fx + g.x, -1.0f), 1.0f);
        o.y = fminf(fmaxf(fy + g.y, -1.0f), 1.0f);
        reinterpret_cast<float2*>(flow)[q] = o;
        if (flowGrad && y + 1 < H && x + 1 < W) {
            const float dx = f[(b * 2) * HW + p + W + 1] - fx, dy = f[(b * 2 + 1) * HW + p + W + 1] - fy;
            flowGrad[b * (long long)(H - 1) * (W - 1) + (long long)y * (W - 1) + x] = sqrtf(dx * dx + dy * dy);
        }
    }
}

extern "C" int rfx_warp_grid_f32(const float* Hm, float* grid, int B, int h, int w, void* stream) {
    if (!Hm || !grid || B <= 0 || h <= 0 || w <= 0) return RFX_E_ARG;
    const long long total = (long long)B * h * w;
    hipLaunchKernelGGL(warp_grid_kernel, dim3(grid_for(total, 256)), dim3(256), 0, rfx_stream(stream), Hm, grid, total, h, w);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

extern "C" int rfx_grid_sample_f32(const float* in, const float* grid, float* out, int N, int C, int Hi, int Wi,
                                   int Ho, int Wo, int align_corners, void* stream) {
    if (!in || !grid || !out || N <= 0 || C <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return RFX_E_ARG;
    const long long NP = (long long)N * Ho * Wo;
    hipLaunchKernelGGL(grid_sample_kernel, dim3(grid_for(NP, 256)), dim3(256), 0, rfx_stream(stream), in, grid, out, NP,
                       C, Hi, Wi, Ho, Wo, align_corners);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

extern "C" int rfx_compose_flow_f32(const float* flowDown, const float* coarseGrid, float* flow12, float* inb,
                                    float* flowUp, int N, int hd, int wd, int Hc, int Wc, int H, int W, int clamp,
                                    void* stream) {
    if (!flowDown || !coarseGrid || !flow12 || N <= 0 || hd <= 0 || wd <= 0 || Hc <= 0 || Wc <= 0 || H <= 0 || W <= 0)
        return RFX_E_ARG;
    const long long NP = (long long)N * H * W;
    const float sh = (float)hd / (float)H, sw = (float)wd / (float)W;
    hipLaunchKernelGGL(compose_flow_kernel, dim3(grid_for(NP, 256)), dim3(256), 0, rfx_stream(stream), flowDown,
                       coarseGrid, flow12, inb, flowUp, NP, hd, wd, Hc, Wc, H, W, sh, sw, clamp);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

extern "C" int rfx_merge_multi_h_f32(const float* flow, const float* match12, long long match12_stride,
                                     const float* cyc, const float* inb, int n, long long HW, float th, int multiH,
                                     float* flowGlobal, float* matchGlobal, uint8_t* binary, void* stream) {
    if (!flow || !match12 || !flowGlobal || n <= 0 || HW <= 0 || match12_stride < HW) return RFX_E_ARG;
    hipLaunchKernelGGL(merge_multi_h_kernel, dim3(grid_for(HW, 256)), dim3(256), 0, rfx_stream(stream), flow, match12,
                       match12_stride, cyc, inb, n, HW, th, multiH, flowGlobal, matchGlobal, binary);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

extern "C" int rfx_match_score_f32(const float* match12, long long match12_stride, const float* cyc, const float* inb,
                                   int n, long long HW, float* score, void* stream) {
    if (!match12 || !score || n <= 0 || HW <= 0 || match12_stride < HW) return RFX_E_ARG;
    const long long total = (long long)n * HW;
    hipLaunchKernelGGL(match_score_kernel, dim3(grid_for(total, 256)), dim3(256), 0, rfx_stream(stream), match12,
                       match12_stride, cyc, inb, HW, total, score);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

extern "C" int rfx_flow_grad_clamp_f32(const float* flowCoarse, const float* grid, float* flowGrad, float* flow, int B, int H,
                                       int W, int grid_batch, void* stream) {
    if (!flowCoarse || !grid || !flow || B <= 0 || H <= 0 || W <= 0 || (grid_batch != 1 && grid_batch != B)) return RFX_E_ARG;
    if (flowGrad && (H < 2 || W < 2)) return RFX_E_ARG;
    const long long total = (long long)B * H * W;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(flow_grad_clamp_kernel, dim3(blocks), dim3(256), 0, rfx_stream(stream), flowCoarse, grid, flowGrad, flow, B,
                       H, W, grid_batch);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}
