// stem.hip -- the FeatureExtractor stem as ONE kernel (model/model.py:68-72, forward :106-110):
//     conv3x3(3 -> 64, stride 1, pad 1) -> BatchNorm (folded) -> ReLU -> MaxPool2d(2, stride 1) -> BlurPool(stride 2)
// Un-fused, the 64-channel full-resolution map (480 x 640 x 64 floats per image: the largest tensor of the whole
// pipeline, 10 GB for a batch of 128 images) is written by the convolution and read back by the pooling kernel; both
// are bound by exactly that traffic.  Here a workgroup owns a 4 x 16 tile of POOLED outputs for 32 channels:
//   1. the 3 x 12 x 36 input patch it needs goes to LDS (zero filled outside the image = the conv padding);
//   2. the 10 x 34 conv outputs under the tile are computed on the fp32 MFMA: pixel p = row*34 + col is a column of
//      the 32x32 tile, k = c*9 + kh*3 + kw (27, padded to 28) runs in the same order and the same (2kk, 2kk+1)
//      pairing as conv.hip, so the accumulators are bit-identical to the stand-alone convolution; B operands are
//      single ds_read_b32 from the patch at per-lane addresses;
//   3. fma(acc, scale, shift) -> ReLU -> LDS tile [32 ch][10][34];
//   4. max 2x2 / blur [1 2 1]^2/16 stride 2 with ReflectionPad2d(1) on the max-pooled map, in the operation order of
//      maxblurpool2d_kernel (pool.hip) -> bit-identical to conv2d + maxblurpool2d; a thread produces 2x2 blocks of
//      outputs from a 6x6 window (border blocks take the per-output path that reflects indices).
// Only the /2 map is written: HBM traffic per image drops from (3 + 64 + 64 + 16) to (3 + 16) full-resolution planes.
#include "common.h"
#include "group.h"

namespace {

constexpr int TH = 4, TW = 16;                 // pooled outputs per workgroup
constexpr int CR = 2 * TH + 2, CC = 2 * TW + 2; // conv outputs under the tile: 10 x 34
constexpr int PRW = CR + 2, PCL = CC + 2;       // input patch 12 x 36
constexpr int CST = CC + 1;                     // row stride of the conv tile in LDS (35: odd -> rows on different banks)
// experiments only (scripts/ubench/stem_bench.py; WRONG RESULTS): RFX_STEM3_DBG removes one phase of the 3x3 stem to price it
//   1 no pooling pass / output stores   2 pooling arithmetic but no interior stores   3 no MFMA phase / C-tile stores
#ifndef RFX_STEM3_DBG
#define RFX_STEM3_DBG 0
#endif
#ifndef RFX_STEM3_PLANE_PAD
#define RFX_STEM3_PLANE_PAD 1                   // 0: round 5's 350-word planes (A/B runs)
#endif
constexpr int CPS = CR * CST + RFX_STEM3_PLANE_PAD;   // channel-plane stride 351: ODD, so the four channels a wavefront of the pooling pass covers
                                                // (8 blocks x 2 block rows x 4 channels, block columns 4 words apart) fall into the four bank
                                                // classes mod 4 -- every bank twice, the minimum for 64 lanes; 350 gave two classes: 4-way
                                                // conflicts on each of the 36 window reads (SQ_LDS_BANK_CONFLICT 0.20 per instruction, round 6)
constexpr int NPX = CR * CC;                    // 340 conv pixels
constexpr int NSUB = (NPX + 31) / 32;           // 11 MFMA sub-tiles of 32 pixels
constexpr int KKS = 14;                         // 28 = 27 padded k, as k-pairs
constexpr int MCH = 32;                         // channels per workgroup

struct StemArgs {
    const float* in; const float* wT; const float* scale; const float* shift; float* out;
    int N, H, W, Cout, Mpad, Ho, Wo, tilesH, tilesW, chGroups;
};

__device__ __forceinline__ int reflect1i(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// The pooled values are ReLU outputs: +0, positive, +inf or NaN.  For such floats the IEEE order is the order of the bit
// patterns as unsigned integers and every NaN pattern lies above +inf, so an unsigned integer max IS the NaN-propagating
// float max of MaxPool2d (which NaN payload survives is the only freedom) -- 1 instruction instead of 4 per comparison.
__device__ __forceinline__ float umaxf(float x, float y) {
    const unsigned a = __float_as_uint(x), b = __float_as_uint(y);
    return __uint_as_float(a > b ? a : b);
}

__global__ __launch_bounds__(256) void stem_conv_maxblur_kernel(StemArgs a) {
    __shared__ float P[3][PRW][PCL];          // input patch
    __shared__ float C[MCH * CPS];            // conv + BN + ReLU outputs under the tile: [channel][CR][CST], planes CPS apart

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lrow = lane >> 5, lcol = lane & 31;
    int bid = blockIdx.x;
    const int cg = bid % a.chGroups; bid /= a.chGroups;       // channel group fastest: the groups of a tile share the patch in L2
    const int tw = bid % a.tilesW; bid /= a.tilesW;
    const int th = bid % a.tilesH;
    const int n = bid / a.tilesH;
    const int oh0 = th * TH, ow0 = tw * TW;
    const int cy0 = 2 * oh0 - 1, cx0 = 2 * ow0 - 1;           // first conv row / column under the tile (= first max-map row / column)
    const int m0 = cg * MCH;
    const size_t HW = (size_t)a.H * a.W;

    // ---- 1. input patch (rows cy0-1 .. cy0+10, cols cx0-1 .. cx0+34), zero outside the image
    const float* inn = a.in + (size_t)n * 3 * HW;
    constexpr int NP = (3 * PRW * PCL + 255) / 256;   // 6 patch elements per thread
    float pv_[NP];
    bool pok[NP];
#pragma unroll
    for (int u = 0; u < NP; ++u) {   // all loads first, masks afterwards: a select on a load in flight would wait for it
        const int idx = t + 256 * u;
        const int c = idx / (PRW * PCL), rem = idx - c * (PRW * PCL);
        const int pr = rem / PCL, pc = rem - pr * PCL;
        const int gy = cy0 - 1 + pr, gx = cx0 - 1 + pc;
        pok[u] = idx < 3 * PRW * PCL && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
        pv_[u] = inn[pok[u] ? (size_t)c * HW + (size_t)gy * a.W + gx : 0];
    }
    // A operand of this lane: weights of channel m0 + lcol for k = 2kk + lrow (rows K..Kpad-1 of wT are zero)
    float af[KKS];
#pragma unroll
    for (int kk = 0; kk < KKS; ++kk) af[kk] = a.wT[(size_t)(2 * kk + lrow) * a.Mpad + m0 + lcol];
    // per-lane patch offsets of the 14 k-pairs (tap (c, kh, kw) of k = 2kk + lrow); k = 27 is the zero pad
    int koff[KKS];
#pragma unroll
    for (int kk = 0; kk < KKS; ++kk) {
        const int k = 2 * kk + lrow;
        const int c = k / 9, t9 = k - c * 9, kh = t9 / 3, kw = t9 - kh * 3;
        koff[kk] = k < 27 ? c * (PRW * PCL) + kh * PCL + kw : 0;
    }
    float sc[16], sh[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ch = m0 + 4 * lrow + (r & 3) + 8 * (r >> 2);
        sc[r] = a.scale ? a.scale[ch] : 1.0f;
        sh[r] = a.shift ? a.shift[ch] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < NP; ++u)
        if (t + 256 * u < 3 * PRW * PCL) (&P[0][0][0])[t + 256 * u] = pok[u] ? pv_[u] : 0.0f;
    __syncthreads();

    // ---- 2./3. conv on the MFMA, BN + ReLU, tile -> LDS
    const float* pf = &P[0][0][0];
    for (int s = wave; s < NSUB && RFX_STEM3_DBG != 3; s += 4) {
        const int p = s * 32 + lcol;
        const bool pv = p < NPX;
        const int pc = pv ? p : 0;
        const int py = pc / CC, px = pc - py * CC;
        const int pbase = py * PCL + px;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int kk = 0; kk < KKS; ++kk) {
            float b = pf[pbase + koff[kk]];
            if (kk == KKS - 1) b = lrow ? 0.0f : b;   // k = 27: padded tap
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk], b, acc, 0, 0, 0);
        }
        if (pv) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = fmaf(acc[r], sc[r], sh[r]);
                v = v > 0.0f ? v : 0.0f;
                C[(4 * lrow + (r & 3) + 8 * (r >> 2)) * CPS + py * CST + px] = v;
            }
        }
    }
    __syncthreads();

    // ---- 4. max 2x2 (stride 1) + blur/2 with reflection on the max-pooled map; 2x2 output blocks
    const float w3[3] = {0.25f, 0.5f, 0.25f};
    const int Hm = a.H - 1, Wm = a.W - 1;
    for (int b = t; b < MCH * (TH / 2) * (TW / 2) && RFX_STEM3_DBG != 1; b += 256) {
        const int bx = b % (TW / 2), by = (b / (TW / 2)) % (TH / 2), ch = b / ((TW / 2) * (TH / 2));
        const int oh = oh0 + 2 * by, ow = ow0 + 2 * bx;        // first output of the block
        if (oh >= a.Ho || ow >= a.Wo) continue;
        float* dst = a.out + ((size_t)n * a.Cout + m0 + ch) * a.Ho * a.Wo;
        const float* Cc = C + ch * CPS;
        // interior: the 5x5 max-map window rows 2oh-1 .. 2oh+3, cols 2ow-1 .. 2ow+3 needs no reflection
        const bool interior = oh >= 1 && 2 * oh + 3 <= Hm - 1 && ow >= 1 && 2 * ow + 3 <= Wm - 1 && oh + 1 < a.Ho && ow + 1 < a.Wo;
        if (interior) {
            const int ly = 2 * oh - 1 - cy0, lx = 2 * ow - 1 - cx0;   // = 4*by, 4*bx
            float v[6][6];
#pragma unroll
            for (int y = 0; y < 6; ++y)
#pragma unroll
                for (int x = 0; x < 6; ++x) v[y][x] = Cc[(ly + y) * CST + lx + x];
            float hmx[5][6], M[5][5];   // separable: vertical pair max, then horizontal pair max
#pragma unroll
            for (int y = 0; y < 5; ++y)
#pragma unroll
                for (int x = 0; x < 6; ++x) hmx[y][x] = umaxf(v[y][x], v[y + 1][x]);
#pragma unroll
            for (int y = 0; y < 5; ++y)
#pragma unroll
                for (int x = 0; x < 5; ++x) M[y][x] = umaxf(hmx[y][x], hmx[y][x + 1]);
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    float acc = 0.0f;
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int j = 0; j < 3; ++j) acc = fmaf(M[2 * dy + i][2 * dx + j], w3[i] * w3[j], acc);
                    if (RFX_STEM3_DBG != 2 || acc == 12345.678f) dst[(size_t)(oh + dy) * a.Wo + ow + dx] = acc;
                }
            continue;
        }
        for (int dy = 0; dy < 2; ++dy)
            for (int dx = 0; dx < 2; ++dx) {
                const int o_h = oh + dy, o_w = ow + dx;
                if (o_h >= a.Ho || o_w >= a.Wo) continue;
                float acc = 0.0f;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int my = reflect1i(2 * o_h - 1 + i, Hm) - cy0;
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const int mx = reflect1i(2 * o_w - 1 + j, Wm) - cx0;
                        const float m = umaxf(umaxf(Cc[my * CST + mx], Cc[(my + 1) * CST + mx]), umaxf(Cc[my * CST + mx + 1], Cc[(my + 1) * CST + mx + 1]));
                        acc = fmaf(m, w3[i] * w3[j], acc);
                    }
                }
                dst[(size_t)o_h * a.Wo + o_w] = acc;
            }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// ResNet-50 stem (model/resnet50.py:115-120, forward :157-160 as sliced by quick_start/coarseAlignFeatMatch.py:38-41):
//     conv7x7(3 -> 64, stride 2, pad 3) -> BatchNorm (folded) -> ReLU -> MaxPool2d(3, stride 2, pad 1)
// Same scheme: a workgroup owns TH x 16 = 5 x 16 pooled outputs for 32 channels, computes the 11 x 33 conv outputs under them on
// the MFMA from a 3 x 27 x 71 input patch in LDS (k = c*49 + kh*7 + kw, 147 padded to 148, same order and pairing as
// the implicit-GEMM kernel -> bit-identical accumulators), keeps them in LDS after BN + ReLU and pools from there.
// The stride-2 taps of 32 consecutive conv pixels would hit every second LDS word (2-way bank conflicts): the patch
// rows are stored de-interleaved, [even columns | odd columns], so a lane's tap (kh, kw) sits at
// parity(kw)*36 + px + kw/2 and consecutive pixels read consecutive words.
// Round 6 (scripts/ubench/stem_bench.py, profiles/r06_stem7_variants.json; 64 images per level, TFLOP/s at 960x1280 .. 240x320):
//   TH = 4 pooled rows (9 x 33 conv outputs = 10 MFMA sub-tiles on 4 waves: 3 + 3 + 2 + 2), BN vectors in registers   54-55 (round 5)
//   TH = 4, BN vectors in LDS (32 registers less)                                                                      50-55
//   TH = 5 (11 x 33 conv outputs = 12 sub-tiles: 3 per wave, 83 % instead of 67 % useful MFMA slots)                   56-62  <- default
//   one workgroup walking both 32-channel groups over the staged patch (RFX_STEM7_CGLOOP=1: half the patch loads)      40-50: the
//     two inlined group bodies keep 150 registers spilled around the barriers (rolled sub-tile loops); kept for experiments only
#ifndef RFX_STEM7_CGLOOP
#define RFX_STEM7_CGLOOP 0
#endif
// experiments only (scripts/ubench/stem_bench.py; WRONG RESULTS): RFX_STEM7_DBG removes one phase to price it
//   1 no pooling / output stores   2 no patch loads from global memory   3 no MFMA phase   4 no weight (A fragment) loads
#ifndef RFX_STEM7_DBG
#define RFX_STEM7_DBG 0
#endif
// RFX_STEM7_ROWSTAGE=1 (experiment, round 6): patch staging by wave-uniform rows -- 690 fewer vector-ALU instructions per wave (444
// instead of 1135 before the first barrier), bit-identical, and NO change in the kernel's time (62.1 vs 62.1 TFLOP/s): the staging
// phase's VALU work does not hold the other workgroup's MFMA phase back.  Off: the element-wise form stays the product path.
#ifndef RFX_STEM7_ROWSTAGE
#define RFX_STEM7_ROWSTAGE 0
#endif
#ifndef RFX_STEM7_UNROLL
#define RFX_STEM7_UNROLL 0
#endif
#ifndef RFX_STEM7_TH
#define RFX_STEM7_TH 5      // experiments: make exp NAME=stem4 SRC=stem DEFS=-DRFX_STEM7_TH=4
#endif
namespace r50 {
constexpr int TH = RFX_STEM7_TH, TW = 16;
constexpr int CR = 2 * TH + 1, CC = 2 * TW + 1;    // 9 x 33 conv outputs
constexpr int PR = 2 * CR + 5, PCW = 2 * CC + 5;   // 23 x 71 input patch
constexpr int PHALF = 36, PST = 2 * PHALF + 2;     // de-interleaved row: 36 even + 36 odd columns (+2: the 9 lanes of the column
                                                   // sub-tile below sit 2*PST = 148 words apart = 20 banks: all distinct but one pair)
// conv-output tile in LDS, rows de-interleaved by column parity as well ([17 even | 17 odd at +17], row stride 40): the
// pooling pass reads columns 2*ow + j of 16 consecutive pooled outputs as CONSECUTIVE words, and the four pooled rows of a
// wavefront lie 2*CST = 80 words = 16 banks apart -> every bank is hit exactly twice by the 64 lanes (the minimum).
#ifndef RFX_STEM7_CST
#define RFX_STEM7_CST 40
#endif
constexpr int CHALF = 17, CST = RFX_STEM7_CST;
// MFMA sub-tiles are ROW-ALIGNED (round 4): sub-tile s < 9 = conv row s, columns 0..31 (32 consecutive LDS words per tap: no
// bank conflict); sub-tile 9 = column 32 of the nine rows (9 live lanes).  The old numbering p = s*32 + lcol over the 9 x 33
// map made most sub-tiles straddle two rows that land 2*PST = 16 banks apart on overlapping banks: 2-way conflicts on the B
// reads -- 19 % of the kernel's instructions in round 3's counters (SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_ANY).
constexpr int NSUB = CR + 1;                       // 9 x 33 outputs -> 9 row sub-tiles + 1 column sub-tile
constexpr int KKS = 74;                            // 148 / 2
constexpr int MCH = 32;
__host__ __device__ constexpr int koff(int k) {    // patch offset of tap k = c*49 + kh*7 + kw
    return k >= 147 ? 0 : (k / 49) * (PR * PST) + ((k % 49) / 7) * PST + (((k % 49) % 7) & 1) * PHALF + (((k % 49) % 7) >> 1);
}
}  // namespace r50

struct Stem7Args {
    const float* in; const float* wT; const float* scale; const float* shift; float* out;
    int N, H, W, Cout, Mpad, Hc, Wc, Hp, Wp, tilesH, tilesW, chGroups;
#ifdef RFX_TRACE
    long long* trace;      // experiments only (make trace): 6 shader-clock stamps per workgroup (scripts/dbg/stem_trace.py)
#endif
};
#ifdef RFX_TRACE
extern "C" long long* rfx_debug_trace_ptr();
#define RFX_STAMP7(i) do { if (threadIdx.x == 0 && a.trace) a.trace[(size_t)bx * 8 + (i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define RFX_STAMP7(i)
#endif

__device__ __forceinline__ void stem7_conv_maxpool_body(const Stem7Args& a, const unsigned bx) {
    constexpr int TH = r50::TH, TW = r50::TW, CR = r50::CR, CC = r50::CC, PR = r50::PR, PCW = r50::PCW, PHALF = r50::PHALF,
                  PST = r50::PST, CST = r50::CST, CHALF = r50::CHALF, NSUB = r50::NSUB, KKS = r50::KKS, MCH = r50::MCH;
    using r50::koff;
    __shared__ float P[3][PR][PST];
    __shared__ float C[MCH][CR][CST];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lrow = lane >> 5, lcol = lane & 31;
    int bid = (int)bx;
    // RFX_STEM7_CGLOOP (default since round 6): ONE workgroup per pooled tile walks the Cout / 32 channel groups over the SAME staged
    // input patch (the patch load + its barrier were paid once per channel group before: 2x for the 64-channel stem)
    const int cg_first = RFX_STEM7_CGLOOP ? 0 : bid % a.chGroups;
    const int cg_last = RFX_STEM7_CGLOOP ? a.chGroups : cg_first + 1;
    if (!RFX_STEM7_CGLOOP) bid /= a.chGroups;
    const int tw = bid % a.tilesW; bid /= a.tilesW;
    const int th = bid % a.tilesH;
    const int n = bid / a.tilesH;
    const int oh0 = th * TH, ow0 = tw * TW;
    const int cy0 = 2 * oh0 - 1, cx0 = 2 * ow0 - 1;          // first conv row / column under the tile
    const int iy0 = 2 * cy0 - 3, ix0 = 2 * cx0 - 3;          // first input row / column of the patch
    const size_t HW = (size_t)a.H * a.W;

    RFX_STAMP7(0);
    // ---- input patch -> registers.  RFX_STEM7_ROWSTAGE (round 6): by ROWS -- wave w takes patch rows w, w+4, .. (3 x PR rows of PCW <= 71
    // columns: lane = column, lanes 0..PCW-65 a second one), so that the row's base address, its validity and its LDS row are
    // wave-uniform (scalar ALU) and a load costs ONE vector instruction besides itself (the border select).  The element-wise form
    // (element t + 256u of the patch) spent ~20 vector-ALU instructions per element on index arithmetic, bounds and 64-bit addresses;
    // every fp32 VALU instruction takes ~4 cycles out of the matrix pipe's time (profiles/r06_mfma_forms_and_pipe_concurrency.txt, C).
    const float* inn = a.in + (size_t)n * 3 * HW;
#if RFX_STEM7_ROWSTAGE
    constexpr int NROW = (3 * PR + 3) / 4;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int gx0 = ix0 + lane, gx1 = ix0 + lane + 64;
    const bool cok0 = (unsigned)gx0 < (unsigned)a.W;
    const bool has1 = lane + 64 < PCW;
    const bool cok1 = has1 && (unsigned)gx1 < (unsigned)a.W;
    const int gxc0 = cok0 ? gx0 : 0, gxc1 = cok1 ? gx1 : 0;
    const int lc0 = (lane & 1) * PHALF + (lane >> 1), lc1 = (lane & 1) * PHALF + ((lane + 64) >> 1);
    float pv0[NROW], pv1[NROW];
#pragma unroll
    for (int j = 0; j < NROW; ++j) {
        const int row = wave_u + 4 * j;                       // = c * PR + pr, wave-uniform
        const int c = row / PR, pr = row - c * PR, gy = iy0 + pr;
        const bool rok = row < 3 * PR && (unsigned)gy < (unsigned)a.H;
        const float* rp = inn + (size_t)(row < 3 * PR ? c : 0) * HW + (size_t)(rok ? gy : 0) * a.W;
        pv0[j] = RFX_STEM7_DBG == 2 ? (float)j : rp[gxc0];
        pv1[j] = RFX_STEM7_DBG == 2 ? (float)j : rp[gxc1];
    }
#else
    constexpr int NP = (3 * PR * PCW + 255) / 256;
    float pv_[NP];
    unsigned pok = 0;
    // element t + 256u of the 3 x 23 x 71 patch: (c, pr, pc) advanced incrementally (256 = 3*71 + 43), no divisions
    const int c_0 = t / (PR * PCW), rem_0 = t - c_0 * (PR * PCW), pr_0 = rem_0 / PCW, pc_0 = rem_0 - pr_0 * PCW;
    {
        int c = c_0, pr = pr_0, pc = pc_0;
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int gy = iy0 + pr, gx = ix0 + pc;
            const bool ok = c < 3 && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
            pok |= ok ? (1u << u) : 0u;
            pv_[u] = RFX_STEM7_DBG == 2 ? (float)u : inn[ok ? (size_t)c * HW + (size_t)gy * a.W + gx : 0];
            pc += 256 % PCW; pr += 256 / PCW;
            if (pc >= PCW) { pc -= PCW; ++pr; }
            if (pr >= PR) { pr -= PR; ++c; }
        }
    }
#endif
    float af[KKS];
    // folded-BN vectors of every channel group in LDS (round 6: 32 registers less across the MFMA loop -- with the channel-group
    // loop the kernel sat at the 256-register cap and spilled 112)
    __shared__ float s_bn[2][128];
    for (int c = t; c < a.Cout && c < 128; c += 256) {
        s_bn[0][c] = a.scale ? a.scale[c] : 1.0f;
        s_bn[1][c] = a.shift ? a.shift[c] : 0.0f;
    }
    auto load_weights = [&](int m0) {
        const float* wl = a.wT;
        asm volatile("" : "+s"(wl) :: "memory");     // opaque per call: the loads of the NEXT channel group must not be hoisted into this one
#pragma unroll
        for (int kk = 0; kk < KKS; ++kk) af[kk] = RFX_STEM7_DBG == 4 ? (float)(kk + lcol) : wl[(size_t)(2 * kk + lrow) * a.Mpad + m0 + lcol];
    };
    if (!RFX_STEM7_CGLOOP) load_weights(cg_first * MCH);          // in flight while the patch goes to LDS
#if RFX_STEM7_ROWSTAGE
    {
        float* pflat = &P[0][0][0];
#pragma unroll
        for (int j = 0; j < NROW; ++j) {
            const int row = wave_u + 4 * j;
            if (row < 3 * PR) {                                // wave-uniform
                const bool rok = (unsigned)(iy0 + row - (row / PR) * PR) < (unsigned)a.H;
                pflat[row * PST + lc0] = (rok && cok0) ? pv0[j] : 0.0f;
            }
        }
        if (has1) {                                            // ONE exec-mask region for the few lanes that own a second column
#pragma unroll
            for (int j = 0; j < NROW; ++j) {
                const int row = wave_u + 4 * j;
                if (row < 3 * PR) {
                    const bool rok = (unsigned)(iy0 + row - (row / PR) * PR) < (unsigned)a.H;
                    pflat[row * PST + lc1] = (rok && cok1) ? pv1[j] : 0.0f;
                }
            }
        }
    }
#else
    {
        int c = c_0, pr = pr_0, pc = pc_0;
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            if (c < 3) P[c][pr][(pc & 1) * PHALF + (pc >> 1)] = ((pok >> u) & 1u) ? pv_[u] : 0.0f;
            pc += 256 % PCW; pr += 256 / PCW;
            if (pc >= PCW) { pc -= PCW; ++pr; }
            if (pr >= PR) { pr -= PR; ++c; }
        }
    }
#endif
    RFX_STAMP7(1);
    __syncthreads();
    RFX_STAMP7(2);

    // One channel group (32 channels) from the staged patch: MFMA phase -> C tile -> pooling -> global.  A generic lambda inlined
    // once per group (two groups for the 64-channel stem) rather than a run-time loop: as a loop the three sub-tile passes stay
    // rolled, the scheduler hoists their 74 operand reads on top of the 74 weight registers and the kernel spills ~100 registers.
    auto channel_group = [&](const int cg, const bool first) {
    const int m0 = cg * MCH;
    const float* pf = &P[0][0][0];
    if (RFX_STEM7_CGLOOP) {
        if (!first) __syncthreads();                           // everyone is done pooling from C
        load_weights(m0);
        asm volatile("" : "+v"(pf));                           // the patch is invariant across groups: keep its reads in THIS group
    }
    // ---- conv on the MFMA, BN + ReLU -> LDS
#if RFX_STEM7_UNROLL
#pragma unroll
#endif
    for (int s = wave; s < NSUB && RFX_STEM7_DBG != 3; s += 4) {
        const bool pv = s < CR || lcol < CR;
        const int py = s < CR ? s : (lcol < CR ? lcol : 0), px = s < CR ? lcol : CC - 1;
        const int pbase = 2 * py * PST + px;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int kk = 0; kk < KKS; ++kk) {
            const int off = lrow ? koff(2 * kk + 1) : koff(2 * kk);
            float b = RFX_STEM7_DBG == 5 ? af[(kk + 1) % KKS] : pf[pbase + off];      // 5: no LDS operand reads
            if (kk == KKS - 1) b = lrow ? 0.0f : b;   // k = 147: padded tap
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk], b, acc, 0, 0, 0);
        }
        if (pv) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int chl = m0 + 4 * lrow + (r & 3) + 8 * (r >> 2);
                float v = fmaf(acc[r], s_bn[0][chl], s_bn[1][chl]);
                v = v > 0.0f ? v : 0.0f;
                C[4 * lrow + (r & 3) + 8 * (r >> 2)][py][(px & 1) * CHALF + (px >> 1)] = v;
            }
        }
    }
    RFX_STAMP7(3);
    __syncthreads();
    RFX_STAMP7(4);

    // ---- MaxPool2d(3, stride 2, pad 1): -inf padding = positions outside the conv map are skipped; the values are ReLU
    // outputs, so the unsigned-integer max is the NaN-propagating float max and 0 its identity (see umaxf above)
    for (int o = t; o < MCH * TH * TW && RFX_STEM7_DBG != 1; o += 256) {
        const int owl = o % TW, ohl = (o / TW) % TH, ch = o / (TW * TH);
        const int oh = oh0 + ohl, ow = ow0 + owl;
        if (oh >= a.Hp || ow >= a.Wp) continue;
        float m = 0.0f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int gy = cy0 + 2 * ohl + i;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int gx = cx0 + 2 * owl + j;
                const bool ok = (unsigned)gy < (unsigned)a.Hc && (unsigned)gx < (unsigned)a.Wc;
                const float v = C[ch][2 * ohl + i][(j & 1) * CHALF + owl + (j >> 1)];     // column 2*owl + j
                m = ok ? umaxf(m, v) : m;
            }
        }
        a.out[(((size_t)n * a.Cout + m0 + ch) * a.Hp + oh) * a.Wp + ow] = m;
    }
    RFX_STAMP7(5);
    };   // channel_group
    if (RFX_STEM7_CGLOOP && a.chGroups == 2) {
        channel_group(0, true);
        channel_group(1, false);
    } else {
#pragma unroll 1
        for (int cg = cg_first; cg < cg_last; ++cg) channel_group(cg, cg == cg_first);
    }
}

__global__ __launch_bounds__(256, 2) void stem7_conv_maxpool_kernel(Stem7Args a) {
    stem7_conv_maxpool_body(a, blockIdx.x);
}

// grouped form (group.h): blockIdx.y = problem, the same body on that problem's argument block
__global__ __launch_bounds__(256, 2) void stem7_conv_maxpool_group_kernel(RfxGroupArgs<Stem7Args> g) {
    const unsigned y = blockIdx.y;
    if (blockIdx.x >= g.gx[y]) return;
    stem7_conv_maxpool_body(g.p[y], blockIdx.x);
}

static int stem7_group_launch(const void* blob, const unsigned* gx, int n, hipStream_t st) {
    return rfx_group_launch_impl<Stem7Args>(stem7_conv_maxpool_group_kernel, 256, blob, gx, n, st);
}

}  // namespace

extern "C" int rfx_stem_conv3x3_maxblur_f32(const float* in, const float* wT, const float* scale, const float* shift,
                                            float* out, int N, int H, int W, int Cout, void* stream) {
    if (!in || !wT || !out || N <= 0 || H < 3 || W < 3 || Cout <= 0) return RFX_E_ARG;
    if (Cout % MCH != 0) return RFX_E_ARG;
    StemArgs a;
    a.in = in; a.wT = wT; a.scale = scale; a.shift = shift; a.out = out;
    a.N = N; a.H = H; a.W = W; a.Cout = Cout; a.Mpad = (Cout + 127) / 128 * 128;
    a.Ho = (H - 2) / 2 + 1; a.Wo = (W - 2) / 2 + 1;
    a.tilesH = (a.Ho + TH - 1) / TH; a.tilesW = (a.Wo + TW - 1) / TW; a.chGroups = Cout / MCH;
    const long long nwg = (long long)N * a.tilesH * a.tilesW * a.chGroups;
    if (nwg > 0x7fffffffLL) return RFX_E_LIMIT;
    hipLaunchKernelGGL(stem_conv_maxblur_kernel, dim3((unsigned)nwg), dim3(256), 0, rfx_stream(stream), a);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

extern "C" int rfx_stem_conv7x7_maxpool_f32(const float* in, const float* wT, const float* scale, const float* shift,
                                            float* out, int N, int H, int W, int Cout, void* stream) {
    if (!in || !wT || !out || N <= 0 || H < 1 || W < 1 || Cout <= 0) return RFX_E_ARG;
    if (Cout % r50::MCH != 0) return RFX_E_ARG;
    if (Cout > 128) return RFX_E_LIMIT;                    // folded-BN vectors staged in LDS (s_bn)
    Stem7Args a;
    a.in = in; a.wT = wT; a.scale = scale; a.shift = shift; a.out = out;
    a.N = N; a.H = H; a.W = W; a.Cout = Cout; a.Mpad = (Cout + 127) / 128 * 128;
    a.Hc = (H + 6 - 7) / 2 + 1; a.Wc = (W + 6 - 7) / 2 + 1;
    a.Hp = (a.Hc + 2 - 3) / 2 + 1; a.Wp = (a.Wc + 2 - 3) / 2 + 1;
    a.tilesH = (a.Hp + r50::TH - 1) / r50::TH; a.tilesW = (a.Wp + r50::TW - 1) / r50::TW; a.chGroups = Cout / r50::MCH;
    const long long nwg = (long long)N * a.tilesH * a.tilesW * (RFX_STEM7_CGLOOP ? 1 : a.chGroups);
    if (nwg > 0x7fffffffLL) return RFX_E_LIMIT;
#ifdef RFX_TRACE
    a.trace = rfx_debug_trace_ptr();
#endif
    if (rfx_group_recording()) return rfx_group_record(&stem7_group_launch, &a, sizeof(a), (unsigned)nwg);
    hipLaunchKernelGGL(stem7_conv_maxpool_kernel, dim3((unsigned)nwg), dim3(256), 0, rfx_stream(stream), a);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}
