// Shared epilogue of the MFMA convolution kernels (conv.hip, conv3x3.hip):
//     y = fma(acc, scale, shift) (+ residual) -> ReLU / sigmoid / none      (folded eval-mode BatchNorm, model/resnet50.py:93-103)
// C/D layout of v_mfma_f32_32x32x2_f32: lane (lrow = lane>>5, lcol = lane&31) holds, for r = 0..15, the element
// (row = (r&3) + 8*(r>>2) + 4*lrow, col = lcol) of a 32x32 sub-tile; rows = output channels, cols = pixels.
//
// Measured with per-workgroup phase stamps (scripts/dbg/conv_trace.py): the first version of this epilogue took
// 13-18 us per workgroup -- as long as the whole K loop of a 128-deep 1x1 convolution -- because every one of the 64
// outputs of a lane went through a runtime activation switch (scalar branches), an exec-mask branch for the store
// predicate and two LDS reads.  Hence two paths:
//   * fast path (wave-uniform test: every output channel of the workgroup tile exists, activation != sigmoid):
//     no per-element control flow -- a lane whose pixel lies outside the image (ragged tile) skips the 16 stores of
//     a sub-tile under ONE exec-mask region; scale/shift are read once per 32-channel group, the residual values of
//     sub-tile s+1 are a batch of 16 independent loads in flight while sub-tile s is finished and stored;
//   * general path (last channel tile of a Cout that is not a multiple of the tile, sigmoid): per-element predicates.
#pragma once
#include "common.h"

// pix_off[j]: element offset of this lane's pixel of sub-tile j inside channel plane 0 of its image
//             (n*Cout*HWo + oh*Wout + ow); pix_ok[j]: that pixel exists.
// Residual values of a whole wavefront tile, fetched ahead of time (PRE): issued before a K loop that is long enough to
// hide their latency, they leave the epilogue with arithmetic and stores only.  Costs TM*TN*16 registers.
template <int TM, int TN>
__device__ __forceinline__ void conv_residual_prefetch(float (&pre)[TM * TN][16], const float* __restrict__ resp, size_t HWo,
                                                       int m0, int wm, int lrow, const size_t (&pix_off)[TN]) {
#pragma unroll
    for (int s = 0; s < TM * TN; ++s) {
        const size_t base = pix_off[s % TN] + (size_t)(m0 + (wm * TM + s / TN) * 32 + 4 * lrow) * HWo;
#pragma unroll
        for (int r = 0; r < 16; ++r) pre[s][r] = resp[base + (size_t)((r & 3) + 8 * (r >> 2)) * HWo];
    }
}

template <int TM, int TN, bool LEAN = false, bool PRE = false>
__device__ __forceinline__ void conv_epilogue(const f32x16 (&acc)[TM][TN], const float* s_scale, const float* s_shift,
                                              const float* __restrict__ resp, float* __restrict__ outp, int act,
                                              int Cout, size_t HWo, int m0, int wm, int lrow,
                                              const size_t (&pix_off)[TN], const bool (&pix_ok)[TN], bool m_full,
                                              const float (*pre)[16] = nullptr) {
    const bool has_res = resp != nullptr;
    if (m_full && act != RFX_ACT_SIGMOID) {
        const bool relu = act == RFX_ACT_RELU;
        constexpr int NS = TM * TN;   // 32x32 sub-tiles of this wavefront, walked i-major
        auto sub_base = [&](int s) { return pix_off[s % TN] + (size_t)(m0 + (wm * TM + s / TN) * 32 + 4 * lrow) * HWo; };
        // residual values: sub-tile s+1 is in flight while s is finished and stored (LEAN: register-tight kernels
        // fetch them just in time instead)
        float rv[2][16];
        if (has_res && !LEAN && !PRE) {
            const size_t b0 = sub_base(0);
#pragma unroll
            for (int r = 0; r < 16; ++r) rv[0][r] = resp[b0 + (size_t)((r & 3) + 8 * (r >> 2)) * HWo];
        }
        float sc[16], sh[16];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int i = s / TN, j = s % TN;
            if (j == 0 || LEAN) {
                const int ml0 = (wm * TM + i) * 32 + 4 * lrow;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    sc[r] = s_scale[ml0 + (r & 3) + 8 * (r >> 2)];
                    sh[r] = s_shift[ml0 + (r & 3) + 8 * (r >> 2)];
                }
            }
            if (has_res && LEAN && !PRE) {
                const size_t bc = sub_base(s);
#pragma unroll
                for (int r = 0; r < 16; ++r) rv[s & 1][r] = resp[bc + (size_t)((r & 3) + 8 * (r >> 2)) * HWo];
            }
            if (has_res && !LEAN && !PRE && s + 1 < NS) {
                const size_t bn = sub_base(s + 1);
#pragma unroll
                for (int r = 0; r < 16; ++r) rv[(s + 1) & 1][r] = resp[bn + (size_t)((r & 3) + 8 * (r >> 2)) * HWo];
            }
            const size_t base = sub_base(s);
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = fmaf(acc[i][j][r], sc[r], sh[r]);
            if (has_res) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] += PRE ? pre[s][r] : rv[s & 1][r];
            }
            if (relu) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = v[r] > 0.0f ? v[r] : 0.0f;
            }
            if (pix_ok[j]) {
#pragma unroll
                for (int r = 0; r < 16; ++r) outp[base + (size_t)((r & 3) + 8 * (r >> 2)) * HWo] = v[r];
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the live ranges of one sub-tile from spreading over all
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float rv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ml = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lrow;
                const int mc = (m0 + ml < Cout) ? (m0 + ml) : (Cout - 1);
                rv[r] = has_res ? resp[pix_off[j] + (size_t)mc * HWo] : 0.0f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ml = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lrow;
                const int m = m0 + ml;
                float v = fmaf(acc[i][j][r], s_scale[ml], s_shift[ml]);
                if (has_res) v += rv[r];
                if (act == RFX_ACT_RELU) v = v > 0.0f ? v : 0.0f;
                else if (act == RFX_ACT_SIGMOID) v = 1.0f / (1.0f + expf(-v));
                if (pix_ok[j] && m < Cout) outp[pix_off[j] + (size_t)m * HWo] = v;
            }
        }
    }
}
