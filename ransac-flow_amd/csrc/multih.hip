// multih.hip -- the host-side glue of the multi-homography rounds as device kernels
// (evaluation/evalHpatch/evaluation.py:211-243, evaluation/evalKITTI/evaluation.py:270-336, utils/outil.py:120).
//
// One round of the reference's loop costs, per pair, a D2H copy of the h x w matchability map, numpy mask algebra, an
// H2D copy of the mask, a bilinear resize + threshold + boolean indexing of the cached matches
// (evaluation/evalHpatch/coarseAlignFeatMatch.py:156-163) and a torch.randint for the RANSAC index draw.  Here a round is
//   filter   (one workgroup per active pair)  explained-region mask -> feature-resolution keep map sampled ONLY at the
//            cells the cached matches point to -> ordered ballot compaction -> match1 / match2 / count, all on the device
//   draw     Philox4x32-10 index draw keyed by (seed, stream, pair, hypothesis), reduced modulo the DEVICE-side match
//            count: no host sync for nbMatch, no 10 000-50 000 x 4 CPU draw + upload per pair and homography
//   accept   gain = mean(newly explained matchability) -> accept rule -> mask update -> result-record store, with ONE
//            small readback (the accept flags) per round for the host's active list.
#include "common.h"
#include <math.h>

namespace {

// ---- Philox4x32-10 (Salmon et al., SC'11; the generator behind torch's device-side randint) -----------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
        const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

__global__ __launch_bounds__(256) void draw_samples_kernel(const int32_t* __restrict__ n, int64_t* __restrict__ samples, int N,
                                                           uint32_t seed_lo, uint32_t seed_hi, uint32_t str_lo, uint32_t str_hi,
                                                           const int32_t* __restrict__ ids) {
    const int b = blockIdx.y;
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= N) return;
    const int nb = n[b];
    uint32_t c[4] = {(uint32_t)h, (uint32_t)(ids ? ids[b] : b), str_lo, str_hi};
    philox4x32_10(c, seed_lo, seed_hi);
    int64_t* o = samples + ((size_t)b * N + h) * 4;
    // torch's random_from_to for a range below 2^32: (32 random bits) % range + base
#pragma unroll
    for (int p = 0; p < 4; ++p) o[p] = nb > 0 ? (int64_t)(c[p] % (uint32_t)nb) : 0;
}

// ---- explained-region mask -> keep map at a feature cell ------------------------------------------------------------------
// fg = ((Mask + (1 - bg)) > 0.5) (evaluation/evalHpatch/evaluation.py:212); MtExtend = 1 - fg, bilinear-resized to the target
// feature map (align_corners=False) and thresholded at 0.5 (evaluation/evalHpatch/coarseAlignFeatMatch.py:158-160).
__device__ __forceinline__ float keep_px(const float* __restrict__ mask, const float* __restrict__ bg, size_t o) {
    const float bgv = bg ? bg[o] : 1.0f;
    const float fg = __fadd_rn(mask[o], __fsub_rn(1.0f, bgv)) > 0.5f ? 1.0f : 0.0f;
    return __fsub_rn(1.0f, fg);
}

__device__ __forceinline__ float src_index_nc(float scale, int dst) {   // ATen upsample_bilinear2d, align_corners=False
    const float s = scale * ((float)dst + 0.5f) - 0.5f;
    return s < 0.f ? 0.f : s;
}

__device__ __forceinline__ bool keep_cell(const float* __restrict__ mask, const float* __restrict__ bg, int h, int w, float sh,
                                          float sw, int r, int c) {
    const float fy = src_index_nc(sh, r), fx = src_index_nc(sw, c);
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    const float v00 = keep_px(mask, bg, (size_t)y0 * w + x0), v01 = keep_px(mask, bg, (size_t)y0 * w + x1);
    const float v10 = keep_px(mask, bg, (size_t)y1 * w + x0), v11 = keep_px(mask, bg, (size_t)y1 * w + x1);
    // same operation order as resize_bilinear_kernel (pool.hip), which is pinned against ATen
    const float r0 = fmaf(v01, lx, __fmul_rn(v00, hx));
    const float r1 = fmaf(v11, lx, __fmul_rn(v10, hx));
    return fmaf(r1, ly, __fmul_rn(r0, hy)) > 0.5f;
}

__global__ __launch_bounds__(256) void fill_ones_kernel(float* __restrict__ p, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 1.0f;
}

// One workgroup per active pair: the cached matches of pair b that fall outside the explained region, in order.
__global__ __launch_bounds__(1024) void filter_matches_kernel(
    const int64_t* __restrict__ idx1, const int64_t* __restrict__ idx2, const int32_t* __restrict__ count, int cap,
    const int32_t* __restrict__ active, const float* __restrict__ mask, const float* __restrict__ bg, int h, int w, int rt,
    int ct, float sh, float sw, const float* __restrict__ xa, const float* __restrict__ ya, const float* __restrict__ xb,
    const float* __restrict__ yb, float* __restrict__ m1, float* __restrict__ m2, int32_t* __restrict__ n_out,
    int32_t* __restrict__ kept) {
    const int k = blockIdx.x;
    const int b = active ? active[k] : k;
    const int n = count[b];
    const size_t HW = (size_t)h * w;
    mask += (size_t)b * HW;
    if (bg) bg += (size_t)b * HW;
    idx1 += (size_t)b * cap; idx2 += (size_t)b * cap;
    m1 += (size_t)k * cap * 3; m2 += (size_t)k * cap * 3;
    if (kept) kept += (size_t)k * cap;
    __shared__ int wsum[16];
    __shared__ int base;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) base = 0;
    __syncthreads();
    for (int s = 0; s < n; s += 1024) {
        const int i = s + t;
        bool keep = false;
        int64_t a = 0, cell = 0;
        if (i < n) {
            a = idx1[i]; cell = idx2[i];
            const int r = (int)(cell / ct), c = (int)(cell - (int64_t)r * ct);
            keep = keep_cell(mask, bg, h, w, sh, sw, r, c);
        }
        const unsigned long long bal = __ballot(keep);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(bal);
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) { const int cq = wsum[q]; if (q < wave) woff += cq; tot += cq; }
        const int b0 = base;
        if (keep) {
            const size_t o = (size_t)(b0 + woff + before) * 3;
            m1[o] = xa[a]; m1[o + 1] = ya[a]; m1[o + 2] = 1.0f;
            m2[o] = xb[cell]; m2[o + 1] = yb[cell]; m2[o + 2] = 1.0f;
            if (kept) kept[b0 + woff + before] = i;
        }
        __syncthreads();
        if (t == 0) base = b0 + tot;
        __syncthreads();
    }
    const int ntot = base;
    for (int i = ntot + t; i < cap; i += 1024) {
        const size_t o = (size_t)i * 3;
        m1[o] = m1[o + 1] = m1[o + 2] = 0.0f;
        m2[o] = m2[o + 1] = m2[o + 2] = 0.0f;
        if (kept) kept[i] = -1;
    }
    if (t == 0) n_out[k] = ntot;
}

// The whole keep map of the active pairs: what CoarseAlign.getCoarse of variants A / C multiplies the target features with before
// EVERY mutual matching (quick_start/coarseAlignFeatMatch.py:136-143, evaluation/evalYFCC/coarseAlignFeatMatch.py:158-166) --
// keep[k][cell] = 1 where the bilinear-resized (1 - fg) exceeds 0.5, fg = ((mask + (1 - bg)) > 0.5).
__global__ __launch_bounds__(256) void keep_mask_kernel(const float* __restrict__ mask, const float* __restrict__ bg,
                                                        const int32_t* __restrict__ active, int h, int w, int rt, int ct, float sh,
                                                        float sw, float* __restrict__ keep) {
    const int k = blockIdx.y, b = active ? active[k] : k;
    const int cell = blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= rt * ct) return;
    const size_t HW = (size_t)h * w;
    const float* m = mask + (size_t)b * HW;
    const float* g = bg ? bg + (size_t)b * HW : nullptr;
    keep[(size_t)k * rt * ct + cell] = keep_cell(m, g, h, w, sh, sw, cell / ct, cell % ct) ? 1.0f : 0.0f;
}

// ---- accept rule ----------------------------------------------------------------------------------------------------
constexpr int NPART = 64;     // partial sums per pair

// mode 0 (evalHpatch/evaluation.py:225,238-239): stat = match * (1 - fg); mask <- (mask + match * (1 - fg)) >= 1
// mode 1 (evalKITTI/evaluation.py:322,332-333):  stat = (match > 0.9999) * (1 - fg); mask <- (mask + match * (1 - fg)) > 0.9999
__device__ __forceinline__ float fg_px(const float* __restrict__ mask, const float* __restrict__ bg, size_t o) {
    const float bgv = bg ? bg[o] : 1.0f;
    return __fadd_rn(mask[o], __fsub_rn(1.0f, bgv)) > 0.5f ? 1.0f : 0.0f;
}

__global__ __launch_bounds__(256) void accept_partial_kernel(const float* __restrict__ match, const float* __restrict__ mask,
                                                             const float* __restrict__ bg, const int32_t* __restrict__ active,
                                                             long long HW, int mode, double* __restrict__ part) {
    const int k = blockIdx.y, b = active ? active[k] : k;
    match += (size_t)k * HW; mask += (size_t)b * HW;
    if (bg) bg += (size_t)b * HW;
    const long long per = (HW + NPART - 1) / NPART;
    const long long p0 = blockIdx.x * per, p1 = p0 + per < HW ? p0 + per : HW;
    double s = 0.0;
    for (long long p = p0 + threadIdx.x; p < p1; p += 256) {
        const float nf = __fsub_rn(1.0f, fg_px(mask, bg, (size_t)p));
        const float m = match[p];
        s += (double)__fmul_rn(mode ? (m > 0.9999f ? 1.0f : 0.0f) : m, nf);
    }
    __shared__ double red[256];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[(size_t)k * NPART + blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void accept_update_kernel(const float* __restrict__ match, float* __restrict__ mask,
                                                            const float* __restrict__ bg, const int32_t* __restrict__ active,
                                                            long long HW, int mode, const double* __restrict__ part,
                                                            const int32_t* __restrict__ res, const int32_t* __restrict__ n_match,
                                                            const int32_t* __restrict__ nbH, double th,
                                                            int32_t* __restrict__ accept, float* __restrict__ gain) {
    const int k = blockIdx.y, b = active ? active[k] : k;
    __shared__ int s_acc;
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int q = 0; q < NPART; ++q) s += part[(size_t)k * NPART + q];     // fixed order: every block gets the same sum
        const float g = (float)(s / (double)HW);
        const bool ok = n_match[k] >= 4 && res[k * 4] == 0 && ((double)g > th || nbH[b] == 0);
        s_acc = ok ? 1 : 0;
        if (blockIdx.x == 0) { accept[k] = ok ? 1 : 0; gain[k] = g; }
    }
    __syncthreads();
    if (!s_acc) return;
    match += (size_t)k * HW; mask += (size_t)b * HW;
    if (bg) bg += (size_t)b * HW;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < HW; p += (long long)gridDim.x * 256) {
        const float mk = mask[p];
        const float bgv = bg ? bg[p] : 1.0f;
        const float fg = __fadd_rn(mk, __fsub_rn(1.0f, bgv)) > 0.5f ? 1.0f : 0.0f;
        const float v = __fadd_rn(mk, __fmul_rn(match[p], __fsub_rn(1.0f, fg)));
        mask[p] = (mode ? v > 0.9999f : v >= 1.0f) ? 1.0f : 0.0f;
    }
}

// The per-pair result record (what evaluation/evalHpatch/evaluation.py:254-260 saves per pair): slot nbH[b] of
// pair b receives H, flowDown8, (match12Down8 | match21Down8) and -- KITTI -- the half-resolution /8 flow; nbH[b] += 1.
__global__ __launch_bounds__(1024) void accept_store_kernel(const int32_t* __restrict__ active, const int32_t* __restrict__ accept,
                                                            int32_t* __restrict__ nbH, const float* __restrict__ bestH,
                                                            const float* __restrict__ flow8, const float* __restrict__ m12,
                                                            const float* __restrict__ m21, int hw8, const float* __restrict__ flowd2,
                                                            int hwd2, float* __restrict__ rec, long long rec_stride, int max_h,
                                                            int off_H, int off_flow, int off_match, int off_d2) {
    const int k = blockIdx.x, b = active ? active[k] : k;
    if (!accept[k]) return;
    const int slot = nbH[b];
    const int t = threadIdx.x;
    if (rec && slot < max_h) {
        float* r = rec + (size_t)b * rec_stride;
        if (t < 9) r[off_H + slot * 9 + t] = bestH[k * 9 + t];
        if (flow8)
            for (int i = t; i < 2 * hw8; i += 1024) r[off_flow + (size_t)slot * 2 * hw8 + i] = flow8[(size_t)k * 2 * hw8 + i];
        if (m12 && m21)
            for (int i = t; i < hw8; i += 1024) {
                r[off_match + (size_t)slot * 2 * hw8 + i] = m12[(size_t)k * hw8 + i];
                r[off_match + (size_t)slot * 2 * hw8 + hw8 + i] = m21[(size_t)k * hw8 + i];
            }
        if (flowd2)
            for (int i = t; i < 2 * hwd2; i += 1024) r[off_d2 + (size_t)slot * 2 * hwd2 + i] = flowd2[(size_t)k * 2 * hwd2 + i];
    }
    __syncthreads();
    if (t == 0) {
        nbH[b] = slot + 1;
        // the record's nbH field never exceeds the slots it holds; a pair that accepted more (only the unbounded KITTI loop
        // can) is flagged with status 3, and the device counter nbH[] keeps the true number
        if (rec) {
            rec[(size_t)b * rec_stride] = (float)(slot + 1 < max_h ? slot + 1 : max_h);
            rec[(size_t)b * rec_stride + 1] = slot + 1 > max_h ? 3.0f : 0.0f;
        }
    }
}

}  // namespace

extern "C" int rfx_draw_samples_i64(const int32_t* n, int64_t* samples, int N, int batch, uint64_t seed, uint64_t stream_id,
                                    const int32_t* pair_ids, void* stream) {
    if (!n || !samples || N <= 0 || batch <= 0) return RFX_E_ARG;
    if (batch > 65535) return RFX_E_LIMIT;
    hipLaunchKernelGGL(draw_samples_kernel, dim3((N + 255) / 256, batch), dim3(256), 0, rfx_stream(stream), n, samples, N,
                       (uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)stream_id, (uint32_t)(stream_id >> 32), pair_ids);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

extern "C" int rfx_filter_matches_f32(const int64_t* idx1, const int64_t* idx2, const int32_t* count, int cap,
                                      const int32_t* active, int n_active, const float* mask, const float* bg, int h, int w,
                                      int rt, int ct, const float* xa, const float* ya, const float* xb, const float* yb,
                                      float* match1, float* match2, int32_t* n_out, int32_t* kept, void* stream) {
    if (!idx1 || !idx2 || !count || !mask || !xa || !ya || !xb || !yb || !match1 || !match2 || !n_out || cap <= 0 ||
        n_active <= 0 || h <= 0 || w <= 0 || rt <= 0 || ct <= 0)
        return RFX_E_ARG;
    const float sh = (float)h / (float)rt, sw = (float)w / (float)ct;      // as rfx_resize_bilinear_f32 (align_corners = 0)
    hipLaunchKernelGGL(filter_matches_kernel, dim3(n_active), dim3(1024), 0, rfx_stream(stream), idx1, idx2, count, cap, active,
                       mask, bg, h, w, rt, ct, sh, sw, xa, ya, xb, yb, match1, match2, n_out, kept);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

extern "C" int rfx_keep_mask_f32(const float* mask, const float* bg, const int32_t* active, int n_active, int h, int w, int rt,
                                 int ct, float* keep, void* stream) {
    if (!keep || n_active <= 0 || h <= 0 || w <= 0 || rt <= 0 || ct <= 0) return RFX_E_ARG;
    if (n_active > 65535) return RFX_E_LIMIT;
    if (!mask && !bg) {                      // nothing explained, no background map: every cell is kept
        hipLaunchKernelGGL(fill_ones_kernel, dim3(((size_t)n_active * rt * ct + 255) / 256), dim3(256), 0, rfx_stream(stream), keep,
                           (size_t)n_active * rt * ct);
        RFX_LAUNCH_CHECK();
        return RFX_OK;
    }
    const float sh = (float)h / (float)rt, sw = (float)w / (float)ct;      // as rfx_resize_bilinear_f32 (align_corners = 0)
    if (!mask) return RFX_E_ARG;             // a background map without a mask: the caller passes a zero mask
    hipLaunchKernelGGL(keep_mask_kernel, dim3((rt * ct + 255) / 256, n_active), dim3(256), 0, rfx_stream(stream), mask, bg, active, h,
                       w, rt, ct, sh, sw, keep);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

extern "C" size_t rfx_multih_accept_ws_bytes(int n_active) { return n_active > 0 ? (size_t)n_active * NPART * sizeof(double) : 0; }

extern "C" int rfx_multih_accept_f32(const float* match, float* mask, const float* bg, const int32_t* active, int n_active, int h,
                                     int w, const int32_t* ransac_result, const int32_t* n_match, int32_t* nbH, double th,
                                     int mode, int32_t* accept, float* gain, void* ws, const float* bestH, const float* flowDown8,
                                     const float* match12Down8, const float* match21Down8, int h8, int w8, const float* flowD2,
                                     int hd2, int wd2, float* rec, long long rec_stride, int max_h, int off_H, int off_flow,
                                     int off_match, int off_d2, void* stream) {
    if (!match || !mask || !ransac_result || !n_match || !nbH || !accept || !gain || !ws || n_active <= 0 || h <= 0 || w <= 0 ||
        (mode != 0 && mode != 1))
        return RFX_E_ARG;
    if (rec && (!bestH || max_h <= 0 || rec_stride <= 0)) return RFX_E_ARG;
    if (n_active > 65535) return RFX_E_LIMIT;
    hipStream_t st = rfx_stream(stream);
    const long long HW = (long long)h * w;
    double* part = static_cast<double*>(ws);
    hipLaunchKernelGGL(accept_partial_kernel, dim3(NPART, n_active), dim3(256), 0, st, match, mask, bg, active, HW, mode, part);
    RFX_LAUNCH_CHECK();
    long long g = (HW + 255) / 256;
    if (g > 256) g = 256;
    hipLaunchKernelGGL(accept_update_kernel, dim3((unsigned)g, n_active), dim3(256), 0, st, match, mask, bg, active, HW, mode, part,
                       ransac_result, n_match, nbH, th, accept, gain);
    RFX_LAUNCH_CHECK();
    hipLaunchKernelGGL(accept_store_kernel, dim3(n_active), dim3(1024), 0, st, active, accept, nbH, bestH, flowDown8, match12Down8,
                       match21Down8, h8 * w8, flowD2, hd2 * wd2, rec, rec_stride, max_h, off_H, off_flow, off_match, off_d2);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}
