// mutual_nn.hip -- all-pairs feature correlation fused with mutual-nearest-neighbour selection
// (utils/outil.py:32-45, mutualMatching; mask multiply of quick_start/coarseAlignFeatMatch.py:143).
//
// The reference materialises score = A^T B (nA x nB fp32, 41 MB at 480x640, 850 MB at KITTI size) plus
// four more nA x nB temporaries (topk x2, scatter_ x2, product, nonzero).  Here the score tile lives only
// in the MFMA accumulators:
//   1. mnn_tile_kernel    one 128x128 score tile per workgroup on v_mfma_f32_32x32x2_f32 (K = C in
//                         feature order, k-ordered fma chain), then in-register arg-max along both axes
//                         -> per-tile row/column partial maxima (value, index) in the workspace;
//   2. mnn_reduce_kernel  first-maximum reduction of the partials -> row arg-max, column arg-max;
//   3. mnn_compact_kernel mutual test + (v*v > 0) test + ordered compaction (ascending source index,
//                         i.e. the row-major order of the reference's nonzero()).
// Ties (bit-equal scores) resolve to the smallest index on both axes; they only occur for all-zero
// (masked) columns, which the v*v > 0 test drops exactly as the reference's product test does.
#include "common.h"
#include <math.h>
#include <stdlib.h>

namespace {

constexpr int BM = 128, BN = 128, BK = 32;

struct MnnArgs {
    const float* A; const float* B; const float* maskB;
    int ldA, ldB, nA, nB, C;
    int tilesA, tilesB;
    int kch;                                   // chunked accumulation: K steps (of 32 k) per chunk; 0 = one fma chain over C
    long long strideA, strideB, strideMask;   // element strides between the pairs of a batch (blockIdx.y)
    size_t wsStride;                           // byte stride of the per-pair workspace
    size_t oRowPartVal, oRowPartIdx, oColPartVal, oColPartIdx, oRowVal, oRowIdx, oColIdx;  // offsets inside it
    char* ws;
    int64_t* idx1; int64_t* idx2; int32_t* count; long long idxStride;
};

__device__ __forceinline__ void take_min_idx(float& bv, int& bi, float ov, int oi) {
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
}

// Chunked accumulation (round 4).  A score is a sum of C = 1024 non-negative products; as ONE fma chain its round-off is 2.8x that of
// the CPU reference's sgemm (utils/outil.py:34, torch.mm: K-blocked register tiles whose partial sums are added), and it is THIS
// error -- 8e-8 rms against 3e-8 from the trunk features -- that decides which float64 near-tie of the arg-max flips
// (scripts/summation_order_model.py).  Every a.kch K steps (8 x 32 = 256 k) the accumulators are added to a running total
// and restarted; with a.kch == 0 the single close after the last step leaves the chain sum unchanged.
__device__ __forceinline__ void mnn_close_chunk(f32x16 (&tot)[2][2], f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { tot[i][j][r] += acc[i][j][r]; acc[i][j][r] = 0.0f; }
}

// Arg-max epilogue of a (128 WA) x 128 score tile held in the accumulators of the (2 WA) x 2 wavefronts (C/D layout: column = lcol,
// row = (r&3) + 8*(r>>2) + 4*lrow): per-tile column and row maxima (value, first index) -> the workspace.  xval / xidx:
// max(2 WA x 128, 2 x 128 WA) floats / ints of LDS that no wavefront reads any more.  `ta` indexes the column partials: one slot
// per TILE row (a 256-row tile fills every second slot of the 128-row layout; the reduce kernel walks tilesA slots).
template <int WA = 1>
__device__ __forceinline__ void mnn_tile_epilogue(f32x16 (&acc)[2][2], const MnnArgs& a, float* xval, int* xidx, int i0, int j0,
                                                  int ta, int tb, float* rowPartVal, int* rowPartIdx, float* colPartVal,
                                                  int* colPartIdx) {
    constexpr int NWM = 2 * WA, ROWS = 128 * WA;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = lane >> 5, lcol = lane & 31;

    // ---- column arg-max over this tile's rows (C/D: col = lcol, row = (r&3) + 8*(r>>2) + 4*lrow) ----
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gi = i0 + (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lrow;
                const float v = acc[i][j][r];
                if (gi < a.nA && v > bv) { bv = v; bi = gi; }  // rows visited in increasing order: first max kept
            }
        const float ov = __shfl_xor(bv, 32, 64);
        const int oi = __shfl_xor(bi, 32, 64);
        take_min_idx(bv, bi, ov, oi);
        if (lrow == 0) {
            xval[wm * 128 + (wn * 2 + j) * 32 + lcol] = bv;
            xidx[wm * 128 + (wn * 2 + j) * 32 + lcol] = bi;
        }
    }
    __syncthreads();
    if (t < 128 && j0 + t < a.nB) {
        float bv = xval[t];
        int bi = xidx[t];
#pragma unroll
        for (int w = 1; w < NWM; ++w) take_min_idx(bv, bi, xval[w * 128 + t], xidx[w * 128 + t]);   // row blocks in increasing order
        colPartVal[(size_t)ta * a.nB + j0 + t] = bv;
        colPartIdx[(size_t)ta * a.nB + j0 + t] = bi;
    }
    __syncthreads();

    // ---- row arg-max over this tile's columns ----
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float bv = -INFINITY;
            int bj = 0x7fffffff;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int gj = j0 + (wn * 2 + j) * 32 + lcol;
                const float v = acc[i][j][r];
                if (gj < a.nB && v > bv) { bv = v; bj = gj; }
            }
#pragma unroll
            for (int m = 16; m >= 1; m >>= 1) {
                const float ov = __shfl_xor(bv, m, 64);
                const int oj = __shfl_xor(bj, m, 64);
                take_min_idx(bv, bj, ov, oj);
            }
            if (lcol == 0) {
                const int li = (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lrow;
                xval[wn * ROWS + li] = bv;
                xidx[wn * ROWS + li] = bj;
            }
        }
    __syncthreads();
    if (t < ROWS && i0 + t < a.nA) {
        float bv = xval[t];
        int bj = xidx[t];
        take_min_idx(bv, bj, xval[ROWS + t], xidx[ROWS + t]);
        rowPartVal[(size_t)tb * a.nA + i0 + t] = bv;
        rowPartIdx[(size_t)tb * a.nA + i0 + t] = bj;
    }
}

// VEC: both feature matrices have a leading dimension that is a multiple of 4 and 16-byte aligned rows -- the staging then
// uses one float4 per 4 cells and k row (8 loads per thread and K step instead of 32) with the register transpose of
// conv.hip's weight side; the LDS image and everything after it are the same.
template <bool VEC>
__global__ __launch_bounds__(256, 2) void mnn_tile_kernel(MnnArgs a) {
    constexpr int KK = BK / 2;
    // LDS image per operand: [h = k&1][cell][kk = k>>1] (16 k-pairs per row, XOR-swizzled 16-byte chunks): a lane
    // fetches its operands for a whole BK=32 step with four conflict-free ds_read_b128 (same scheme as conv.hip)
    __shared__ __attribute__((aligned(16))) float As[2][2][BM][KK];
    __shared__ __attribute__((aligned(16))) float Bs[2][2][BN][KK];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = lane >> 5, lcol = lane & 31;

    const int pair = blockIdx.y;
    const float* Ab = a.A + (size_t)pair * a.strideA;
    const float* Bb = a.B + (size_t)pair * a.strideB;
    char* ws = a.ws + (size_t)pair * a.wsStride;
    float* rowPartVal = reinterpret_cast<float*>(ws + a.oRowPartVal);
    int* rowPartIdx = reinterpret_cast<int*>(ws + a.oRowPartIdx);
    float* colPartVal = reinterpret_cast<float*>(ws + a.oColPartVal);
    int* colPartIdx = reinterpret_cast<int*>(ws + a.oColPartIdx);

    const int nwg = a.tilesA * a.tilesB;
    int bid = blockIdx.x;
    {   // XCD-aware bijective remap; column tile fastest so an XCD's L2 keeps one A panel hot
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, j = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int tb = bid % a.tilesB, ta = bid / a.tilesB;
    const int i0 = ta * BM, j0 = tb * BN;

    // staging roles.  scalar: cell `col` of both tiles, parity h of k, all 16 k-pairs of the step.
    // VEC: cells 4*cg .. 4*cg+3, parity hv, k-pairs [iv0, iv0+4) of the step.
    const int col = t % 128;
    const int h = __builtin_amdgcn_readfirstlane(t / 128);
    const bool aval = (i0 + col) < a.nA, bval = (j0 + col) < a.nB;
    const float* Ap = Ab + (aval ? i0 + col : 0);
    const float* Bp = Bb + (bval ? j0 + col : 0);
    const float mk = (!VEC && bval && a.maskB) ? a.maskB[(size_t)pair * a.strideMask + j0 + col] : 1.0f;
    const int cg = t % 32, kq = t / 32;
    const int hv = kq & 1, iv0 = (kq >> 1) * 4;
    // a group of 4 cells lies entirely inside or outside the padded row (ld % 4 == 0): outside -> read cell 0, masked
    const float* Av = Ab + ((i0 + 4 * cg) < a.ldA ? i0 + 4 * cg : 0);
    const float* Bv = Bb + ((j0 + 4 * cg) < a.ldB ? j0 + 4 * cg : 0);
    float mkv[4] = {1.0f, 1.0f, 1.0f, 1.0f};
    if (VEC && a.maskB) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (j0 + 4 * cg + e < a.nB) mkv[e] = a.maskB[(size_t)pair * a.strideMask + j0 + 4 * cg + e];
    }

    // The loads carry no select: a value that depends on a just-issued load makes the compiler wait for it on the
    // spot (the whole gather would serialise on memory latency).  Out-of-range rows / cells read a valid dummy
    // address and are zeroed when the registers are written to LDS, one K step later.
    float ra[KK], rb[KK];
    f32x4 va4[4], vb4[4];
    auto load_global = [&](int k0) {
        if (VEC) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = k0 + hv + 2 * (iv0 + j);
                const int kc = k < a.C ? k : 0;
                va4[j] = *reinterpret_cast<const f32x4*>(Av + (size_t)kc * a.ldA);
                vb4[j] = *reinterpret_cast<const f32x4*>(Bv + (size_t)kc * a.ldB);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < KK; ++i) {
            const int k = k0 + h + 2 * i;
            const int kc = k < a.C ? k : 0;
            ra[i] = Ap[(size_t)kc * a.ldA];
            rb[i] = Bp[(size_t)kc * a.ldB];
        }
    };
    auto store_lds = [&](int buf, int k0) {
        if (VEC) {
            // 4x4 register transpose: element (j, e) of the loaded rows is k-pair iv0+j of cell 4*cg+e
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int cell = 4 * cg + e;
                const bool ain = (i0 + cell) < a.nA, bin = (j0 + cell) < a.nB;
                f32x4 wa, wb;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool kin = (k0 + hv + 2 * (iv0 + j)) < a.C;
                    wa[j] = (kin & ain) ? va4[j][e] : 0.0f;
                    wb[j] = (kin & bin) ? vb4[j][e] * mkv[e] : 0.0f;
                }
                const int chunk = ((iv0 >> 2) ^ (cg & 3)) * 4;   // (cell >> 2) & 3 == cg & 3
                *reinterpret_cast<f32x4*>(&As[buf][hv][cell][chunk]) = wa;
                *reinterpret_cast<f32x4*>(&Bs[buf][hv][cell][chunk]) = wb;
            }
            return;
        }
        const int sw = (col >> 2) & 3;
#pragma unroll
        for (int i = 0; i < KK; ++i) {
            const bool kin = (k0 + h + 2 * i) < a.C;
            ra[i] = (kin & aval) ? ra[i] : 0.0f;
            rb[i] = (kin & bval) ? rb[i] * mk : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < KK / 4; ++q) {
            f32x4 va = {ra[4 * q], ra[4 * q + 1], ra[4 * q + 2], ra[4 * q + 3]};
            f32x4 vb = {rb[4 * q], rb[4 * q + 1], rb[4 * q + 2], rb[4 * q + 3]};
            *reinterpret_cast<f32x4*>(&As[buf][h][col][(q ^ sw) * 4]) = va;
            *reinterpret_cast<f32x4*>(&Bs[buf][h][col][(q ^ sw) * 4]) = vb;
        }
    };

    f32x16 acc[2][2], tot[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = tot[i][j][r] = 0.0f;

    const int nk = (a.C + BK - 1) / BK;
    load_global(0);
    store_lds(0, 0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) load_global((kt + 1) * BK);
        __builtin_amdgcn_sched_barrier(0);   // keep the loads ahead of the MFMAs (the scheduler would sink them)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x4 af[2][2], bf[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int m = (wm * 2 + i) * 32 + lcol;
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    af[i][q] = *reinterpret_cast<const f32x4*>(&As[cur][lrow][m][((half * 2 + q) ^ ((m >> 2) & 3)) * 4]);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int pl = (wn * 2 + j) * 32 + lcol;
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    bf[j][q] = *reinterpret_cast<const f32x4*>(&Bs[cur][lrow][pl][((half * 2 + q) ^ ((pl >> 2) & 3)) * 4]);
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
        __builtin_amdgcn_sched_barrier(0);   // ... and their consumers behind them
        if (kt + 1 == nk || (a.kch && (kt + 1) % a.kch == 0)) mnn_close_chunk(tot, acc);
        if (kt + 1 < nk) store_lds(cur ^ 1, (kt + 1) * BK);
        __syncthreads();
        cur ^= 1;
    }
    // all waves are past the last barrier: the staging buffers are free for the exchange
    mnn_tile_epilogue(tot, a, &As[0][0][0][0], reinterpret_cast<int*>(&Bs[0][0][0][0]), i0, j0, ta, tb, rowPartVal, rowPartIdx,
                      colPartVal, colPartIdx);
}

// ---------------------------------------------------------------------------------------------------------------
// Round 4: the k-major form of the tile kernel.  Both feature matrices are (C, cells) -- k-major, exactly like the two
// operands of a 1x1 convolution -- and a lane of v_mfma_f32_32x32x2_f32 supplies ONE float per operand, so with k-major LDS
// images As[k][cell], Bs[k][cell] a lane's operand is a single ds_read_b32 (32 consecutive lanes read 32 consecutive floats:
// conflict-free by construction) and staging is a straight 16-byte copy without the 4x4 register transposes, the XOR swizzle
// and the validity masks of mnn_tile_kernel above (4.3e9 LDS bank-conflict cycles per launch in round 3's counters): the scheme
// of conv1x1.hip, one barrier per K step, every staging register stored and re-loaded in the shadow of the first two MFMA
// chunks.  Cells past the end of a matrix read clamped (valid, finite) addresses: they only feed accumulator rows / columns
// the arg-max skips.  The 0/1 column mask (quick_start/coarseAlignFeatMatch.py:143 multiplies the target features by it) is
// applied to the finished accumulators: for a 0/1 mask the same values up to the sign of a zero, which no comparison sees.
// Same k pairing and order as the kernel above: bit-identical scores.  Requires C % 32 == 0, C >= 64.
// WA = 2 (round 5 experiment; opt-in: RFX_MNN_WA=2): a 256 x 128 tile on 512 threads = 4 x 2 wavefronts, ONE workgroup
// per CU (96 KB of LDS) instead of two 128 x 128 ones: the same 8 wavefronts per CU and the same 64 x 64 wavefront tiles (64 + 64
// accumulator registers), but a K step stages (256 + 128) x 32 floats for 512 MFMAs where two small tiles stage 2 x (128 + 128) x 32
// -- 25 % fewer operand bytes through the CU's load path per MFMA and 25 % fewer panel (re-)fetches from L2 / the Infinity Cache.
// Same k order per score: identical match lists (scripts/ubench/mnn_bench.py).  MEASURED NEGATIVE on one box
// (profiles/r05_mnn_tile_{128x128,256x128}.json): 106.7 -> 104.6 TFLOP/s at config 3's shape (64 x 13 065 x 1 200), 107.0 -> 104.5 at
// quick_start's, 104.2 -> 98.5 at config 5's (8 x 25 747 x 8 250).  The tile kernel is therefore NOT bound by operand bytes per MFMA
// (nor, a fortiori, by the 4.6x panel re-fetch traffic the counters show: it is served by L2 / the Infinity Cache at ~1 TB/s): one
// barrier per K step across EIGHT wavefronts costs more than two independent 4-wave workgroups lose to each other.  128 x 128 stays.
template <bool VEC, int WA = 1>
__global__ __launch_bounds__(256 * WA, WA == 1 ? 2 : 1) void mnn_tile_kmajor_kernel(MnnArgs a) {
    constexpr int BMA = BM * WA, NT = 256 * WA;
    constexpr int A_TPR = VEC ? BMA / 4 : BMA;     // threads per A row
    constexpr int A_RPR = NT / A_TPR;              // A rows per round of all threads: 8 (VEC) / 2 (scalar)
    constexpr int NLA = BK / A_RPR;                // A loads per thread and K step: 4 / 16
    constexpr int B_TPR = VEC ? BN / 4 : BN;
    constexpr int B_RPR = NT / B_TPR;              // 8 * WA / 2 * WA
    constexpr int NLB = BK / B_RPR;                // 4 / WA, 16 / WA
    __shared__ __attribute__((aligned(16))) float As[2][BK][BMA];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = lane >> 5, lcol = lane & 31;
    const int pair = blockIdx.y;
    const float* Ab = a.A + (size_t)pair * a.strideA;
    const float* Bb = a.B + (size_t)pair * a.strideB;
    char* ws = a.ws + (size_t)pair * a.wsStride;
    float* rowPartVal = reinterpret_cast<float*>(ws + a.oRowPartVal);
    int* rowPartIdx = reinterpret_cast<int*>(ws + a.oRowPartIdx);
    float* colPartVal = reinterpret_cast<float*>(ws + a.oColPartVal);
    int* colPartIdx = reinterpret_cast<int*>(ws + a.oColPartIdx);
    const int nwg = a.tilesA * a.tilesB;
    int bid = blockIdx.x;
    {   // XCD-aware bijective remap; column tile fastest so an XCD's L2 keeps one A panel hot
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, j = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int tb = bid % a.tilesB, ta = bid / a.tilesB;
    const int i0 = ta * BMA, j0 = tb * BN;
    // staging roles: row t / TPR (+ RPR per round), cells (t % TPR) * (VEC ? 4 : 1) ..
    const int arow_s = t / A_TPR, acol_s = (t % A_TPR) * (VEC ? 4 : 1);
    const int brow_s = t / B_TPR, bcol_s = (t % B_TPR) * (VEC ? 4 : 1);
    int ca = i0 + acol_s, cb = j0 + bcol_s;
    if (VEC) { if (ca + 4 > a.ldA) ca = a.ldA - 4; if (cb + 4 > a.ldB) cb = a.ldB - 4; }
    else     { if (ca >= a.nA) ca = a.nA - 1;     if (cb >= a.nB) cb = a.nB - 1; }
    const float* asrc = Ab + (size_t)arow_s * a.ldA + ca;
    const float* bsrc = Bb + (size_t)brow_s * a.ldB + cb;
    f32x4 va[VEC ? NLA : 1], vb[VEC ? NLB : 1];
    float ra[VEC ? 1 : NLA], rb[VEC ? 1 : NLB];
    auto load_a = [&](int k0, int j) {
        if (VEC) va[j] = *reinterpret_cast<const f32x4*>(asrc + (size_t)(k0 + A_RPR * j) * a.ldA);
        else     ra[j] = asrc[(size_t)(k0 + A_RPR * j) * a.ldA];
    };
    auto load_b = [&](int k0, int j) {
        if (VEC) vb[j] = *reinterpret_cast<const f32x4*>(bsrc + (size_t)(k0 + B_RPR * j) * a.ldB);
        else     rb[j] = bsrc[(size_t)(k0 + B_RPR * j) * a.ldB];
    };
    auto store_a = [&](int buf, int j) {
        if (VEC) *reinterpret_cast<f32x4*>(&As[buf][arow_s + A_RPR * j][acol_s]) = va[j];
        else     As[buf][arow_s + A_RPR * j][acol_s] = ra[j];
    };
    auto store_b = [&](int buf, int j) {
        if (VEC) *reinterpret_cast<f32x4*>(&Bs[buf][brow_s + B_RPR * j][bcol_s]) = vb[j];
        else     Bs[buf][brow_s + B_RPR * j][bcol_s] = rb[j];
    };
    const int nk = a.C / BK;
#pragma unroll
    for (int j = 0; j < NLA; ++j) load_a(0, j);
#pragma unroll
    for (int j = 0; j < NLB; ++j) load_b(0, j);
#pragma unroll
    for (int j = 0; j < NLA; ++j) store_a(0, j);
#pragma unroll
    for (int j = 0; j < NLB; ++j) store_b(0, j);
#pragma unroll
    for (int j = 0; j < NLA; ++j) load_a(BK, j);
#pragma unroll
    for (int j = 0; j < NLB; ++j) load_b(BK, j);
    __syncthreads();

    f32x16 acc[2][2], tot[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = tot[i][j][r] = 0.0f;
    const float* arow = &As[0][lrow][wm * 64 + lcol];
    const float* brow = &Bs[0][lrow][wn * 64 + lcol];
    for (int s = 0; s < nk; ++s) {
        const int cur = s & 1;
        const float* ap = arow + cur * (BK * BMA);
        const float* bp = brow + cur * (BK * BN);
        const int k2 = (s + 2 < nk ? s + 2 : nk - 1) * BK;      // past the end: re-load the last step (never consumed)
        float af[2][4][2], bf[2][4][2];
        auto read_chunk = [&](int c, int slot) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int kk = c * 4 + e;
#pragma unroll
                for (int i = 0; i < 2; ++i) af[slot][e][i] = ap[2 * kk * BMA + i * 32];
#pragma unroll
                for (int j = 0; j < 2; ++j) bf[slot][e][j] = bp[2 * kk * BN + j * 32];
            }
        };
        read_chunk(0, 0);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c + 1 < 4) read_chunk(c + 1, (c + 1) & 1);
            // the registers hold K step s+1: stored into the other buffer and re-loaded with step s+2 right behind the store
            if (c == 0) {
#pragma unroll
                for (int j = 0; j < NLA; ++j) store_a(cur ^ 1, j);
#pragma unroll
                for (int j = 0; j < NLA; ++j) load_a(k2, j);
            } else if (c == 1) {
#pragma unroll
                for (int j = 0; j < NLB; ++j) store_b(cur ^ 1, j);
#pragma unroll
                for (int j = 0; j < NLB; ++j) load_b(k2, j);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c & 1][e][i], bf[c & 1][e][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (s + 1 == nk || (a.kch && (s + 1) % a.kch == 0)) mnn_close_chunk(tot, acc);
        __syncthreads();
    }
    if (a.maskB) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int gj = j0 + (wn * 2 + j) * 32 + lcol;
            const float mk = gj < a.nB ? a.maskB[(size_t)pair * a.strideMask + gj] : 0.0f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) tot[i][j][r] *= mk;
        }
    }
    mnn_tile_epilogue<WA>(tot, a, &As[0][0][0], reinterpret_cast<int*>(&Bs[0][0][0]), i0, j0, ta, tb, rowPartVal, rowPartIdx, colPartVal,
                          colPartIdx);
}

__global__ __launch_bounds__(256) void mnn_reduce_kernel(MnnArgs a) {
    char* ws = a.ws + (size_t)blockIdx.y * a.wsStride;
    const float* rowPartVal = reinterpret_cast<const float*>(ws + a.oRowPartVal);
    const int* rowPartIdx = reinterpret_cast<const int*>(ws + a.oRowPartIdx);
    const float* colPartVal = reinterpret_cast<const float*>(ws + a.oColPartVal);
    const int* colPartIdx = reinterpret_cast<const int*>(ws + a.oColPartIdx);
    float* rowVal = reinterpret_cast<float*>(ws + a.oRowVal);
    int* rowIdx = reinterpret_cast<int*>(ws + a.oRowIdx);
    int* colIdx = reinterpret_cast<int*>(ws + a.oColIdx);
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < a.nA) {
        float bv = -INFINITY;
        int bj = 0x7fffffff;
        for (int tb = 0; tb < a.tilesB; ++tb)
            take_min_idx(bv, bj, rowPartVal[(size_t)tb * a.nA + g], rowPartIdx[(size_t)tb * a.nA + g]);
        rowVal[g] = bv;
        rowIdx[g] = bj;
    } else if (g - a.nA < a.nB) {
        const int j = g - a.nA;
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int ta = 0; ta < a.tilesA; ++ta)
            take_min_idx(bv, bi, colPartVal[(size_t)ta * a.nB + j], colPartIdx[(size_t)ta * a.nB + j]);
        colIdx[j] = bi;
    }
}

// single workgroup, ordered compaction by ascending source index
__global__ __launch_bounds__(1024) void mnn_compact_kernel(MnnArgs a) {
    char* ws = a.ws + (size_t)blockIdx.x * a.wsStride;
    const float* rowVal = reinterpret_cast<const float*>(ws + a.oRowVal);
    const int* rowIdx = reinterpret_cast<const int*>(ws + a.oRowIdx);
    const int* colIdx = reinterpret_cast<const int*>(ws + a.oColIdx);
    const int nA = a.nA, nB = a.nB;
    int64_t* idx1 = a.idx1 + (size_t)blockIdx.x * a.idxStride;
    int64_t* idx2 = a.idx2 + (size_t)blockIdx.x * a.idxStride;
    int32_t* count = a.count + blockIdx.x;
    __shared__ int wsum[16];
    __shared__ int base;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) base = 0;
    __syncthreads();
    for (int s = 0; s < nA; s += 1024) {
        const int i = s + t;
        bool keep = false;
        int j = 0;
        if (i < nA) {
            j = rowIdx[i];
            const float v = rowVal[i];
            keep = ((unsigned)j < (unsigned)nB) && (colIdx[j] == i) && (v * v > 0.0f);
        }
        const unsigned long long bal = __ballot(keep);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(bal);
        __syncthreads();
        int woff = 0, tot = 0;
        for (int w = 0; w < 16; ++w) {
            const int c = wsum[w];
            if (w < wave) woff += c;
            tot += c;
        }
        const int b = base;
        if (keep) {
            idx1[b + woff + before] = i;
            idx2[b + woff + before] = j;
        }
        __syncthreads();
        if (t == 0) base = b + tot;
        __syncthreads();
    }
    if (t == 0) count[0] = base;
}

struct WsLayout {
    size_t rowPartVal, rowPartIdx, colPartVal, colPartIdx, rowVal, rowIdx, colIdx, total;
};
inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }
inline WsLayout layout(int nA, int nB) {
    const size_t tA = (nA + BM - 1) / BM, tB = (nB + BN - 1) / BN;
    WsLayout L;
    size_t o = 0;
    L.rowPartVal = o; o += al(tB * nA * 4);
    L.rowPartIdx = o; o += al(tB * nA * 4);
    L.colPartVal = o; o += al(tA * nB * 4);
    L.colPartIdx = o; o += al(tA * nB * 4);
    L.rowVal = o; o += al((size_t)nA * 4);
    L.rowIdx = o; o += al((size_t)nA * 4);
    L.colIdx = o; o += al((size_t)nB * 4);
    L.total = o;
    return L;
}

}  // namespace

extern "C" size_t rfx_mutual_nn_ws_bytes(int nA, int nB) {
    if (nA <= 0 || nB <= 0) return 0;
    return layout(nA, nB).total;
}

// score_chunk (ABI 8, an argument of both entry points): products per accumulation chunk of a score -> a.kch, K steps of 32.
// 0 = the default of 256 products; < 0 = one chain over C.  The host mirror resolves it once per pipeline -- an explicit value, or the
// K blocking of the HOST's sgemm (rfx/ops.py::host_sgemm_k_block: 192 products on the GPU box's EPYC, 384 on a Xeon -- MKL 2024.2),
// which makes a score bit-identical to the reference's torch.mm on that host.
static int mnn_chunk_steps(int score_chunk, int* kch) {
    if (score_chunk == 0) { *kch = 256 / BK; return RFX_OK; }
    if (score_chunk < 0) { *kch = 0; return RFX_OK; }
    if (score_chunk % BK != 0) return RFX_E_ARG;
    *kch = score_chunk / BK;
    return RFX_OK;
}

static int mnn_launch(MnnArgs& a, int batch, hipStream_t st) {
    const WsLayout L = layout(a.nA, a.nB);
    a.tilesA = (a.nA + BM - 1) / BM; a.tilesB = (a.nB + BN - 1) / BN;
    a.wsStride = L.total;
    a.oRowPartVal = L.rowPartVal; a.oRowPartIdx = L.rowPartIdx; a.oColPartVal = L.colPartVal; a.oColPartIdx = L.colPartIdx;
    a.oRowVal = L.rowVal; a.oRowIdx = L.rowIdx; a.oColIdx = L.colIdx;
    const long long nwg = (long long)a.tilesA * a.tilesB;
    if (nwg > 0x7fffffffLL || batch > 65535) return RFX_E_LIMIT;
    const bool vec = a.ldA % 4 == 0 && a.ldB % 4 == 0 && a.strideA % 4 == 0 && a.strideB % 4 == 0 &&
                     ((reinterpret_cast<uintptr_t>(a.A) | reinterpret_cast<uintptr_t>(a.B)) & 15) == 0;
    const char* fe = getenv("RFX_MNN_FORM");                  // 1: force the transposed-image kernel (tests, A/B timing)
    const int form = fe ? atoi(fe) : 0;
    const bool kmajor = form != 1 && a.C % BK == 0 && a.C >= 2 * BK && (vec ? (a.ldA >= 4 && a.ldB >= 4) : true);
    // 256 x 128 tiles (WA = 2): measured slower than two 128 x 128 workgroups per CU (see the kernel) -- only with RFX_MNN_WA=2
    static const int wa_env = getenv("RFX_MNN_WA") ? atoi(getenv("RFX_MNN_WA")) : 1;
    const long long nwg2 = (long long)((a.nA + 2 * BM - 1) / (2 * BM)) * a.tilesB;
    if (kmajor && wa_env == 2 && nwg2 * batch >= 512) {
        a.tilesA = (a.nA + 2 * BM - 1) / (2 * BM);
        if (vec) hipLaunchKernelGGL((mnn_tile_kmajor_kernel<true, 2>), dim3((unsigned)nwg2, batch), dim3(512), 0, st, a);
        else hipLaunchKernelGGL((mnn_tile_kmajor_kernel<false, 2>), dim3((unsigned)nwg2, batch), dim3(512), 0, st, a);
    } else if (kmajor) {
        if (vec) hipLaunchKernelGGL((mnn_tile_kmajor_kernel<true, 1>), dim3((unsigned)nwg, batch), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((mnn_tile_kmajor_kernel<false, 1>), dim3((unsigned)nwg, batch), dim3(256), 0, st, a);
    } else if (vec) hipLaunchKernelGGL(mnn_tile_kernel<true>, dim3((unsigned)nwg, batch), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(mnn_tile_kernel<false>, dim3((unsigned)nwg, batch), dim3(256), 0, st, a);
    RFX_LAUNCH_CHECK();
    hipLaunchKernelGGL(mnn_reduce_kernel, dim3((a.nA + a.nB + 255) / 256, batch), dim3(256), 0, st, a);
    RFX_LAUNCH_CHECK();
    hipLaunchKernelGGL(mnn_compact_kernel, dim3(batch), dim3(1024), 0, st, a);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

extern "C" int rfx_mutual_nn_f32(const float* featA, int ldA, int nA, const float* featB, int ldB, int nB, int C,
                                 const float* maskB, int64_t* idx1, int64_t* idx2, int32_t* count, void* ws,
                                 int score_chunk, void* stream) {
    if (!featA || !featB || !idx1 || !idx2 || !count || !ws) return RFX_E_ARG;
    if (nA <= 0 || nB <= 0 || C <= 0 || ldA < nA || ldB < nB) return RFX_E_ARG;
    MnnArgs a;
    if (mnn_chunk_steps(score_chunk, &a.kch) != RFX_OK) return RFX_E_ARG;
    a.A = featA; a.B = featB; a.maskB = maskB; a.ldA = ldA; a.ldB = ldB; a.nA = nA; a.nB = nB; a.C = C;
    a.strideA = a.strideB = a.strideMask = 0; a.ws = static_cast<char*>(ws);
    a.idx1 = idx1; a.idx2 = idx2; a.count = count; a.idxStride = 0;
    return mnn_launch(a, 1, rfx_stream(stream));
}

extern "C" int rfx_mutual_nn_batched_f32(const float* featA, int ldA, int nA, long long strideA, const float* featB, int ldB,
                                         int nB, long long strideB, int C, const float* maskB, int64_t* idx1, int64_t* idx2,
                                         int32_t* count, void* ws, int batch, int score_chunk, void* stream) {
    if (!featA || !featB || !idx1 || !idx2 || !count || !ws || batch <= 0) return RFX_E_ARG;
    if (nA <= 0 || nB <= 0 || C <= 0 || ldA < nA || ldB < nB) return RFX_E_ARG;
    MnnArgs a;
    if (mnn_chunk_steps(score_chunk, &a.kch) != RFX_OK) return RFX_E_ARG;
    a.A = featA; a.B = featB; a.maskB = maskB; a.ldA = ldA; a.ldB = ldB; a.nA = nA; a.nB = nB; a.C = C;
    a.strideA = strideA; a.strideB = strideB; a.strideMask = nB; a.ws = static_cast<char*>(ws);
    a.idx1 = idx1; a.idx2 = idx2; a.count = count; a.idxStride = nA < nB ? nA : nB;
    return mnn_launch(a, batch, rfx_stream(stream));
}
