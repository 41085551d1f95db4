// conv.hip -- implicit-GEMM fp32 convolution on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Replaces the cuDNN convolutions + eval-mode BatchNorm + ReLU (+ residual add) the reference issues
// for the ResNet-50 conv1..layer3 trunk (model/resnet50.py:68-104,112-169 via
// quick_start/coarseAlignFeatMatch.py:106,124), the FeatureExtractor (model/model.py:27-56,106-115)
// and the NetFlowCoarse / NetMatchability conv stacks (model/model.py:210-226,289-305).
//
// GEMM view (NCHW, no layout change anywhere in the pipeline):
//     D[m][p] = sum_k  W^T[k][m] * im2col[k][p]      m = output channel, p = (n,oh,ow), k = (c,kh,kw)
// A operand = packed weights [Kpad][Mpad] (k-major, zero padded, so tile loads need no guards),
// B operand = input patches gathered on the fly (coalesced along ow), both staged through LDS as
// through LDS; a 32x32x2 MFMA lane supplies one float per operand: lane l holds
// A[m = l&31][k = l>>5] and B[k = l>>5][p = l&31].
// The fp32 MFMA is bit-for-bit a k-ordered fmaf chain, so results are deterministic and within
// fp32 round-off of the reference's cuDNN/oneDNN sums.
//
// Block = 256 threads = 2x2 waves; wave tile = (32*TM) x (32*TN); BK = 32; register-prefetch double
// buffering (global loads of step s+1 are in flight while step s runs on the matrix pipe).  LDS holds each
// operand as [k&1][row][k>>1] so that a lane reads its 16 k-pairs of a step with four ds_read_b128
// (XOR-swizzled 16-byte chunks: conflict free) instead of sixteen ds_read_b32.
// Epilogue fused: y = fma(acc, scale[m], shift[m]) (+ residual) -> ReLU / sigmoid; stores coalesced
// along the pixel dimension (the MFMA C/D column index is the pixel).
#include "common.h"
#include "conv_epilogue.h"
#include "group.h"
#include <stdlib.h>

struct ConvArgs {
    const float* in;
    const float* wT;
    const int32_t* ktab;
    const float* scale;
    const float* shift;
    const float* res;
    float* out;
    int N, Cin, Hin, Win, Cout, Hout, Wout, stride, pad, act;
    int Kpad, Mpad;
    long long P;  // N*Hout*Wout
    int tilesM, tilesP;
#ifdef RFX_TRACE
    long long* trace;  // experiments only (make TRACE=1): 4 wall-clock stamps per workgroup
#endif
};
#ifdef RFX_TRACE
#define RFX_STAMP(i) do { if (tid == 0 && a.trace) a.trace[(size_t)blockIdx.x * 4 + (i)] = wall_clock64(); } while (0)
static long long* g_trace = nullptr;
extern "C" void rfx_debug_trace(long long* p) { g_trace = p; }
extern "C" long long* rfx_debug_trace_ptr() { return g_trace; }
#else
#define RFX_STAMP(i)
#endif

// ONE = 1x1 kernel with pad 0 (any stride): k IS the input channel and every tap is in bounds, so the im2col
// gather needs no ktab decode and no bounds logic (address = pixel base + k * Hin*Win).
// WS = wave-specialised form for long K loops: 512 threads = 4 MFMA wavefronts (one per SIMD, 64x64 outputs each,
// nothing but ds_read_b128 + MFMA in their loop) + 4 loader wavefronts (one per SIMD) that run the whole im2col
// gather / weight fetch / LDS staging.  A 32x32x2 fp32 MFMA keeps the matrix pipe busy for 64 cycles per issue,
// so the loader wave's ~600 address/load/store instructions per K step fit in the issue slots the MFMA wave
// leaves free -- the overlap the single-role kernel only gets statistically from a second workgroup.
// VECB (only with ONE, stride 1, Hout*Wout % 4 == 0): a thread fetches 4 consecutive pixels of one input channel
// with one 16-byte load (the im2col row of a 1x1 convolution is the NCHW plane itself), mirroring the weight side:
// 4x fewer gather instructions.
template <int TM, int TN, bool ONE, bool WS, bool VECB>
__device__ __forceinline__ void conv2d_mfma_body(const ConvArgs& a, const unsigned bx) {
    constexpr int BM = 64 * TM, BN = 64 * TN, BK = 32, KK = BK / 2;  // KK k-pairs per step
    constexpr int B_NI = BN / 8;                                      // gathered values per thread per step
    constexpr int A_MG = BM / 4;                                        // groups of 4 consecutive m per tile
    constexpr int A_KQ = 256 / A_MG;                                    // k-roles per m-group (8 or 16)
    constexpr int A_NJ = 32 / A_KQ;                                     // k rows (float4 loads) per thread: 4 or 2
    constexpr int B_PG = BN / 4, B_KQ = 256 / B_PG, B_NJ = 32 / B_KQ;   // the same decomposition for the pixel side (VECB)
    // LDS image per operand: [h = k&1][row (m or pixel)][kk = k>>1], 16 consecutive k-pairs per row, so that a
    // lane fetches its operands for a whole K step with four ds_read_b128.  16-byte chunk q of row r is stored
    // at chunk q ^ ((r>>2)&3): the 16-lane ds_read_b128 service groups then touch 16 distinct bank quads.
    // Round 5, WHICH access conflicts: the A-side (weight) stores below.  A 16-lane group of one store instruction
    // writes rows mg*4 + e, mg = 0..15 -- 256 bytes apart, i.e. the same bank -- at chunk (iA0 >> 2) ^ (mg & 3): FOUR bank quads for
    // 16 lanes, a 4-way conflict on every weight store (the gathered B side writes whole rows: conflict free; all reads are
    // conflict free).  That is the SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_ANY = 0.41-0.48 of round 4's counters.  The cure that
    // needs no second packed weight array -- a K-MAJOR A image As[k][m] (straight 16-byte copies, ds_read_b32 operands, the scheme
    // of conv1x1.hip) -- was built and measured on the layer that still lives here (1x1 / stride 2 projection 256 -> 512 at 120x160,
    // 128 images): 91.7 vs 95.0 TFLOP/s, i.e. 3.5 % SLOWER (profiles/r05_conv_ab.jsonl): 32 ds_read_b32 per K step and wave tile
    // instead of 8 ds_read_b128 cost more issue slots next to the ~10 VALU instructions per gathered B element than the
    // conflicting stores (8 of ~100 LDS instructions per thread and step).  Reverted; the conflicts stay, priced at < 0.2 % of the step.
    __shared__ __attribute__((aligned(16))) float As[2][2][BM][KK];
    __shared__ __attribute__((aligned(16))) float Bs[2][2][BN][KK];
    __shared__ float s_scale[BM], s_shift[BM];

    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3;
    const int t = tid & 255;  // staging index (WS: the loader wavefronts are threads 256..511)
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware bijective remap: each XCD (observed: block b -> XCD b%8) walks a contiguous chunk of
    // the tile space, m-tile fastest, so the blocks that share one im2col pixel tile run on one L2.
    const int nwg = a.tilesM * a.tilesP;
    int bid = (int)bx;
    RFX_STAMP(0);
    {
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, j = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int tm_idx = bid % a.tilesM;
    const int tp_idx = bid / a.tilesM;
    const int m0 = tm_idx * BM;
    const long long n0 = (long long)tp_idx * BN;

    if (tid < BM) {  // folded BatchNorm of this block's output channels -> LDS (visible after the first barrier)
        const int m = m0 + t;
        s_scale[t] = (a.scale && m < a.Cout) ? a.scale[m] : 1.0f;
        s_shift[t] = (a.shift && m < a.Cout) ? a.shift[m] : 0.0f;
    }
    const int HWo = a.Hout * a.Wout;
    // staging roles: this thread owns row mc of the A image and row pc of the B image, parity hA / hB of k,
    // k-pairs [iA0, iA0+A_NI) / [iB0, iB0+B_NI): exactly the values one MFMA lane will read back.
    // A: thread owns 4 consecutive output channels (one float4 per k row) and A_NJ k rows of one parity
    const int mg = t % A_MG;                 // m = mg*4 .. mg*4+3
    const int kqA = t / A_MG;                // 0 .. A_KQ-1
    const int hA = kqA & 1, iA0 = (kqA >> 1) * A_NJ;
    const int pc = t % BN;
    const int gB = __builtin_amdgcn_readfirstlane(t / BN);
    const int hB = gB & 1, iB0 = (gB >> 1) * B_NI;
    const long long p = n0 + pc;
    const bool pvalid = p < a.P;
    int ih0 = 0, iw0 = 0;
    const float* inb = a.in;
    if (pvalid) {
        const int n = (int)(p / HWo);
        const int rem = (int)(p - (long long)n * HWo);
        const int oh = rem / a.Wout, ow = rem - oh * a.Wout;
        ih0 = oh * a.stride - a.pad;
        iw0 = ow * a.stride - a.pad;
        inb = a.in + (size_t)n * a.Cin * a.Hin * a.Win;
        if (ONE) inb += (size_t)ih0 * a.Win + iw0;
    }
    const size_t HWin = (size_t)a.Hin * a.Win;
    const float* wcol = a.wT + m0 + mg * 4;
    // VECB staging role: pixels pg*4 .. pg*4+3 of the tile (same image: HWo % 4 == 0), B_NJ k rows of parity hV
    const int pg = t % B_PG, kqB = t / B_PG;
    const int hV = kqB & 1, iV0 = (kqB >> 1) * B_NJ;
    const float* inv = a.in;
    bool vvalid = false;
    if (VECB) {
        const long long pv4 = n0 + pg * 4;
        vvalid = pv4 < a.P;
        if (vvalid) {
            const int n = (int)(pv4 / HWo);
            inv = a.in + (size_t)n * a.Cin * HWin + (size_t)(pv4 - (long long)n * HWo);  // stride 1, pad 0: same plane offset
        }
    }
    f32x4 rv[B_NJ];

    f32x4 ra[A_NJ];
    float rb[B_NI];
    // Validity of the B-side values of the step in flight, one bit per load.  It is applied when the registers are
    // written to LDS, NOT at the load: a select on a just-loaded value makes the compiler wait for that load on the
    // spot, which serialises the gather on memory latency (and, with the loads pinned ahead of the MFMAs, would stall
    // the whole step).  Invalid elements read a valid dummy address.
    unsigned okmask = 0;

    auto load_global = [&](int k0) {
#pragma unroll
        for (int j = 0; j < A_NJ; ++j)
            ra[j] = *reinterpret_cast<const f32x4*>(wcol + (size_t)(k0 + hA + 2 * (iA0 + j)) * a.Mpad);
        if (VECB) {
            okmask = 0;
#pragma unroll
            for (int j = 0; j < B_NJ; ++j) {
                const int k = k0 + hV + 2 * (iV0 + j);
                const bool ok = vvalid & (k < a.Cin);
                okmask |= ok ? (1u << j) : 0u;
                rv[j] = *reinterpret_cast<const f32x4*>(inv + (size_t)(ok ? k : 0) * HWin);
            }
            return;
        }
        if (ONE) {
            okmask = 0;
#pragma unroll
            for (int i = 0; i < B_NI; ++i) {
                const int k = k0 + hB + 2 * (iB0 + i);
                const bool ok = pvalid & (k < a.Cin);
                okmask |= ok ? (1u << i) : 0u;
                rb[i] = inb[(size_t)(ok ? k : 0) * HWin];
            }
            return;
        }
        // ktab is stored per 32-k block as [16 even k | 16 odd k]: this wave's B_NI entries are contiguous
        // -> one wide scalar load and a single wait instead of a dependent s_load per element
        const int32_t* kt = a.ktab + k0 + hB * KK + iB0;
        int ev[B_NI];
#pragma unroll
        for (int i = 0; i < B_NI; ++i) ev[i] = kt[i];
        okmask = 0;
#pragma unroll
        for (int i = 0; i < B_NI; ++i) {
            const int e = ev[i];
            const int cin = e >> 8, kh = (e >> 4) & 15, kw = e & 15;
            const int ih = ih0 + kh, iw = iw0 + kw;
            // bitwise & (no short-circuit): keeps the gather branch-free (v_cndmask, not exec-mask branches)
            const bool ok = pvalid & (e >= 0) & ((unsigned)ih < (unsigned)a.Hin) & ((unsigned)iw < (unsigned)a.Win);
            const int off = (cin * a.Hin + ih) * a.Win + iw;
            okmask |= ok ? (1u << i) : 0u;
            rb[i] = inb[ok ? off : 0];  // always a valid address; masked in store_lds
        }
    };
    auto store_lds = [&](int buf) {
        const int swb = (pc >> 2) & 3;
        // 4x(A_NJ) register transpose: element (j, e) of the loaded rows is k-pair iA0+j of channel mg*4+e
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int m = mg * 4 + e;
            float* dst = &As[buf][hA][m][0];
            if (A_NJ == 4) {
                f32x4 v = {ra[0][e], ra[1][e], ra[2][e], ra[3][e]};
                *reinterpret_cast<f32x4*>(dst + (((iA0 >> 2) ^ (mg & 3)) << 2)) = v;   // (m>>2)&3 == mg&3
            } else {
                float2 v;
                v.x = ra[0][e]; v.y = ra[1][e];
                *reinterpret_cast<float2*>(dst + (((iA0 >> 2) ^ (mg & 3)) << 2) + (iA0 & 3)) = v;
            }
        }
        if (VECB) {
#pragma unroll
            for (int j = 0; j < B_NJ; ++j) {
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                rv[j] = ((okmask >> j) & 1u) ? rv[j] : z;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float* dst = &Bs[buf][hV][pg * 4 + e][0];
                if (B_NJ == 4) {
                    f32x4 v = {rv[0][e], rv[1][e], rv[2][e], rv[3][e]};
                    *reinterpret_cast<f32x4*>(dst + (((iV0 >> 2) ^ (pg & 3)) << 2)) = v;
                } else {
                    float2 v;
                    v.x = rv[0][e]; v.y = rv[1][e];
                    *reinterpret_cast<float2*>(dst + (((iV0 >> 2) ^ (pg & 3)) << 2) + (iV0 & 3)) = v;
                }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < B_NI; ++i) rb[i] = ((okmask >> i) & 1u) ? rb[i] : 0.0f;
#pragma unroll
        for (int q = 0; q < B_NI / 4; ++q) {
            f32x4 v = {rb[4 * q], rb[4 * q + 1], rb[4 * q + 2], rb[4 * q + 3]};
            *reinterpret_cast<f32x4*>(&Bs[buf][hB][pc][((iB0 / 4 + q) ^ swb) * 4]) = v;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int lrow = lane >> 5, lcol = lane & 31;
    auto compute = [&](int cur) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {  // two halves of 8 k-pairs keep the fragment registers at 32
            f32x4 af[TM][2], bf[TN][2];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = (wm * TM + i) * 32 + lcol;
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    af[i][q] = *reinterpret_cast<const f32x4*>(&As[cur][lrow][m][((half * 2 + q) ^ ((m >> 2) & 3)) * 4]);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int pl = (wn * TN + j) * 32 + lcol;
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    bf[j][q] = *reinterpret_cast<const f32x4*>(&Bs[cur][lrow][pl][((half * 2 + q) ^ ((pl >> 2) & 3)) * 4]);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][q][e], bf[j][q][e], acc[i][j], 0, 0, 0);
        }
    };

    const int nk = a.Kpad / BK;
    if (WS) {
        if (__builtin_amdgcn_readfirstlane(tid >> 8) != 0) {
            // ---- loader wavefronts: tile kt+1 is written to LDS during step kt, tile kt+2 is in flight ----
            load_global(0);
            store_lds(0);
            if (nk > 1) load_global(BK);
            __syncthreads();
            for (int kt = 0; kt < nk; ++kt) {
                if (kt + 1 < nk) store_lds((kt + 1) & 1);
                if (kt + 2 < nk) load_global((kt + 2) * BK);
                __syncthreads();
            }
            return;
        }
        // ---- MFMA wavefronts ----
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            compute(kt & 1);
            __syncthreads();
        }
    } else {
        load_global(0);
        store_lds(0);
        __syncthreads();
        RFX_STAMP(1);
        int cur = 0;
        for (int kt = 0; kt < nk; ++kt) {
            // branch-free body (the last step re-loads its own tile into the unused buffer): one basic block, so the
            // scheduler is free to interleave the gather of step kt+1 with the MFMAs of step kt
            load_global((kt + 1 < nk ? kt + 1 : kt) * BK);
            __builtin_amdgcn_sched_barrier(0);   // the loads stay AHEAD of the MFMAs: left to itself the scheduler sinks them
            compute(cur);                        // to the end of the step and their latency is exposed at every barrier
            __builtin_amdgcn_sched_barrier(0);   // ... and nothing that consumes them is hoisted into the MFMA stream
            store_lds(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }
    }

    RFX_STAMP(2);
    // epilogue (conv_epilogue.h): this lane's pixel of each 32-pixel sub-tile -> plane offset
    size_t pix_off[TN];
    bool pix_ok[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        long long pp = n0 + (wn * TN + j) * 32 + lcol;
        pix_ok[j] = pp < a.P;
        if (!pix_ok[j]) pp = a.P - 1;
        const int n = (int)(pp / HWo);
        const int rem = (int)(pp - (long long)n * HWo);
        pix_off[j] = (size_t)n * a.Cout * HWo + rem;
    }
    conv_epilogue<TM, TN, WS>(acc, s_scale, s_shift, a.res, a.out, a.act, a.Cout, (size_t)HWo, m0, wm, lrow, pix_off, pix_ok,
                              m0 + BM <= a.Cout);
#ifdef RFX_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RFX_STAMP(3);
#endif
}

template <int TM, int TN, bool ONE, bool WS, bool VECB>
__global__ __launch_bounds__(WS ? 512 : 256, WS ? 4 : 2) void conv2d_mfma_kernel(ConvArgs a) {
    conv2d_mfma_body<TM, TN, ONE, WS, VECB>(a, blockIdx.x);
}

// grouped form (group.h): blockIdx.y = problem, the same body on that problem's argument block
template <int TM, int TN, bool ONE, bool WS, bool VECB>
__global__ __launch_bounds__(WS ? 512 : 256, WS ? 4 : 2) void conv2d_mfma_group_kernel(RfxGroupArgs<ConvArgs> g) {
    const unsigned y = blockIdx.y;
    if (blockIdx.x >= g.gx[y]) return;
    conv2d_mfma_body<TM, TN, ONE, WS, VECB>(g.p[y], blockIdx.x);
}

template <int TM, int TN, bool ONE, bool WS, bool VECB>
static int conv_group_launch(const void* blob, const unsigned* gx, int n, hipStream_t st) {
    return rfx_group_launch_impl<ConvArgs>(conv2d_mfma_group_kernel<TM, TN, ONE, WS, VECB>, WS ? 512 : 256, blob, gx, n, st);
}

template <int TM, int TN, bool ONE, bool WS, bool VECB = false>
static int launch_conv(ConvArgs& a, hipStream_t st) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    a.tilesM = (a.Cout + BM - 1) / BM;
    a.tilesP = (int)((a.P + BN - 1) / BN);
    const long long nwg = (long long)a.tilesM * a.tilesP;
    if (nwg > 0x7fffffffLL) return RFX_E_LIMIT;
#ifdef RFX_TRACE
    a.trace = g_trace;
#endif
    if (rfx_group_recording()) return rfx_group_record(&conv_group_launch<TM, TN, ONE, WS, VECB>, &a, sizeof(a), (unsigned)nwg);
    hipLaunchKernelGGL((conv2d_mfma_kernel<TM, TN, ONE, WS, VECB>), dim3((unsigned)nwg), dim3(WS ? 512 : 256), 0, st, a);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

int rfx_conv3x3_direct_launch(const float* in, const float* wP, const float* scale, const float* shift,
                              const float* residual, float* out, int N, int Cin, int H, int W, int Cout, int Mpad,
                              int act, int tm, int patch_cols, hipStream_t st, bool chunked);  // conv3x3.hip
int rfx_conv3x3_patch_cols(int N, int H, int W, bool fused);                                       // conv3x3.hip
int rfx_conv3x3_s2_launch(const float* in, const float* wP, const float* scale, const float* shift, const float* residual,
                          float* out, int N, int Cin, int H, int W, int Cout, int act, int tm, hipStream_t st);   // conv3x3.hip
bool rfx_conv3x3_wide_patch(int N, int H, int W, int Cout, int patch_cols);                          // conv3x3.hip
int rfx_conv1x1_kmajor_launch(const float* in, const float* wT, const float* scale, const float* shift, const float* residual,
                              float* out, int N, int Cin, int HW, int Cout, int Mpad, int act, int tm, bool vec,
                              hipStream_t st, bool chunked);                            // conv1x1.hip
bool rfx_conv3x3_chunked(int Cin);                                                                   // conv3x3.hip
bool rfx_conv3x3_tail_chunked();                                                                     // conv3x3.hip
bool rfx_conv3x3_s2_chunked(int Cin);                                                                // conv3x3.hip

// tile choice: the largest tile that still gives >= ~2 workgroups per CU (256 CUs).
// 0: 128x128 (conv2d_mfma_kernel<2,2>), 1: 64x128 (<1,2>), 2: 64x64 (<1,1>)
extern "C" int rfx_conv2d_tile_variant(int N, int Cout, int Hout, int Wout) {
    static const int forced = getenv("RFX_CONV_FORCE_VARIANT") ? atoi(getenv("RFX_CONV_FORCE_VARIANT")) : -1;  // experiments
    if (forced >= 0) return forced;
    const long long P = (long long)N * Hout * Wout;
    const long long b22 = (long long)((Cout + 127) / 128) * ((P + 127) / 128);
    const long long b12 = (long long)((Cout + 63) / 64) * ((P + 127) / 128);
    if (Cout > 64 && b22 >= 512) return 0;
    if (b12 >= 512 || P >= 8192) return 1;
    return 2;
}

static int conv_ws_env() {
    static const int v = getenv("RFX_CONV_WS") ? atoi(getenv("RFX_CONV_WS")) : -1;
    return v;
}

// Kernel instance rfx_conv2d_f32 launches for this geometry: bits 0-1 tile variant (0: <2,2>, 1: <1,2>, 2: <1,1>),
// bit 2 = 1x1 specialisation (ONE), bit 3 = wave-specialised form (WS), bit 4 = vectorised pixel-side loads
// (VECB), i.e. the template arguments of conv2d_mfma_kernel<TM,TN,ONE,WS,VECB> that rocprofv3 prints.
// bit 10 = the k-major 1x1 / stride 1 kernel of conv1x1.hip (conv1x1_kmajor_kernel<TM, VEC>, TM = 2 - (bits 0-1), VEC = bit 4).
// bit 5 = the direct 3x3 / stride 1 / pad 1 kernel of conv3x3.hip (conv3x3_direct_kernel<TM, PT_C>, TM = 2 - (bits 0-1 != 0),
// output patch 128/PT_C x PT_C with PT_C = 16 / 8 / 4 for bits 6-7 = 0 / 1 / 2).
static int conv_kernel_id(int N, int Cin, int Cout, int KH, int KW, int stride, int pad, int Hout, int Wout, bool allow_direct,
                          int k_chunk = 0) {
    static const int direct_env = getenv("RFX_CONV_DIRECT") ? atoi(getenv("RFX_CONV_DIRECT")) : 1;
    if (allow_direct && direct_env && KH == 3 && KW == 3 && stride == 1 && pad == 1 && Cin >= 8) {
        const int pc = rfx_conv3x3_patch_cols(N, Hout, Wout, false), pr = 128 / pc;
        const long long tiles = (((long long)N * (Hout + 1) + pr - 1) / pr) * ((Wout + pc - 1) / pc);
        const bool big = Cout > 64 && tiles * ((Cout + 127) / 128) >= 512;
        const bool rag = Cin % 8 != 0;
        // bit 14 = chunked accumulation (K >= 2048: conv3x3_direct_kernel<TM, PT_C, false, 2, false, 4>; never on the 256-pixel patches)
        // ... or asked for by the caller (k_chunk > 0: the 3x3 convolution of a Bottleneck tail run on its own, rfx_conv3x3_f32)
        const bool chk = !rag && (rfx_conv3x3_chunked(Cin) || (k_chunk > 0 && rfx_conv3x3_tail_chunked()));
        return 32 | (big ? 0 : 1) | (pc == 16 ? 0 : (pc == 8 ? 64 : 128)) |
               (!big && !rag && rfx_conv3x3_wide_patch(N, Hout, Wout, Cout, pc) ? 2048 : 0) | (rag ? 4096 : 0) | (chk ? 16384 : 0);
    }
    // bit 13 = the direct 3x3 / stride 2 / pad 1 kernel conv3x3_s2_kernel<TM> (Cin % 8 == 0; TM = 2 - bit 0); never inside a
    // grouped launch (it has no grouped form: a recorded group keeps the implicit-GEMM kernel)
    static const int s2_env = getenv("RFX_CONV_S2") ? atoi(getenv("RFX_CONV_S2")) : 1;
    if (allow_direct && direct_env && s2_env && KH == 3 && KW == 3 && stride == 2 && pad == 1 && Cin % 8 == 0 && Cin >= 8) {
        // its 8 x 16 output patches tile every image on their own (no stacked-batch trick at stride 2): on maps that pad badly
        // the implicit-GEMM kernel, which tiles the flattened pixel axis, wins.  Measured break-even (scripts/ubench/
        // conv_s2_bench.py, profiles/r04_conv_s2_ab.json): +8..16 % at 100 % / 94 % useful pixels, +-0 at 88 %, -12 % at 74 %.
        // Round 5: the layers with K = 9 Cin >= 1152 sum in chunks in this kernel (bit 14) -- they take it on EVERY map and inside
        // grouped launches too (it has a grouped form now), so that a layer's sums never depend on the map size or the batch; the
        // shorter-K layers keep the rule above (either kernel: the same chain, bit-identical).
        const long long th = (Hout + 7) / 8, tw = (Wout + 15) / 16;
        const bool chk = rfx_conv3x3_s2_chunked(Cin);
        if (chk || ((long long)Hout * Wout * 100 >= 90 * th * 8 * tw * 16 && !rfx_group_recording())) {
            const bool big = Cout > 64 && (long long)N * th * tw * ((Cout + 127) / 128) >= 512;
            return 8192 | (big ? 0 : 1) | (chk ? 16384 : 0);
        }
    }
    const int variant = rfx_conv2d_tile_variant(N, Cout, Hout, Wout);
    const bool one = (KH == 1 && KW == 1 && pad == 0);
    static const int kmajor_env = getenv("RFX_CONV_1X1") ? atoi(getenv("RFX_CONV_1X1")) : 1;   // experiments: 0 = generic kernel
    // bit 14 = chunked accumulation (K >= 1024: conv1x1_kmajor_kernel<1, VEC, 8>, 64-channel tiles, at EVERY launch size -- a result
    // must not depend on how many pairs share the launch); RFX_C1_CHUNK=0: off
    // round 5: from K = 512 on (layer2 conv1, layer3.0 conv1: two chunks of 256); RFX_C1_CHUNK=<min K> (0: never; 1: the default 512)
    static const int c1chunk = getenv("RFX_C1_CHUNK") ? atoi(getenv("RFX_C1_CHUNK")) : 1;
    const bool chk = c1chunk && Cin >= (c1chunk > 1 ? c1chunk : 512);
    if (kmajor_env && one && stride == 1 && Cin % 32 == 0 && Cin >= 64 && (variant != 2 || chk) && (long long)N * Hout * Wout >= 4) {
        return 1024 | 4 | (chk ? 1 : variant) | (((long long)Hout * Wout) % 4 == 0 ? 16 : 0) | (chk ? 16384 : 0);   // conv1x1.hip
    }
    const int env = conv_ws_env();
    const bool ws = variant == 2 ? false : (env > 0);  // off by default: since the branch-free epilogue the single-role kernel is as fast
    static const int vec_env = getenv("RFX_CONV_VECB") ? atoi(getenv("RFX_CONV_VECB")) : 1;
    const bool vecb = vec_env && one && !ws && stride == 1 && ((long long)Hout * Wout) % 4 == 0;
    return variant | (one ? 4 : 0) | (ws ? 8 : 0) | (vecb ? 16 : 0);
}

extern "C" int rfx_conv2d_kernel_id(int N, int Cin, int Cout, int KH, int KW, int stride, int pad, int Hout,
                                    int Wout) {
    return conv_kernel_id(N, Cin, Cout, KH, KW, stride, pad, Hout, Wout, true);
}

// The direct 3x3 kernel with the weights in its own packed order (conv3x3.hip); bit 5 of rfx_conv2d_kernel_id says when
// it applies.  rfx_conv2d_f32 computes the same convolution, bit for bit, from the generic wT / ktab packing.
extern "C" int rfx_conv3x3_f32(const float* in, const float* wP, const float* scale, const float* shift,
                               const float* residual, float* out, int N, int Cin, int H, int W, int Cout, int act,
                               int k_chunk, void* stream) {
    if (!in || !wP || !out || N <= 0 || Cin < 8 || H <= 0 || W <= 0 || Cout <= 0) return RFX_E_ARG;
    if (reinterpret_cast<uintptr_t>(wP) & 15) return RFX_E_ARG;
    if ((long long)Cin * H * W > 0x7fffffffLL) return RFX_E_LIMIT;
    if (k_chunk != 0 && k_chunk != 4) return RFX_E_ARG;                      // the one chunk length the kernels are built for
    const int kid = conv_kernel_id(N, Cin, Cout, 3, 3, 1, 1, H, W, true, k_chunk);
    const int pc = rfx_conv3x3_patch_cols(N, H, W, false);
    const int tm = (kid & 32) ? ((kid & 3) ? 1 : 2) : (Cout > 64 ? 2 : 1);   // RFX_CONV_DIRECT=0 only changes the host's choice
    return rfx_conv3x3_direct_launch(in, wP, scale, shift, residual, out, N, Cin, H, W, Cout, (Cout + 127) / 128 * 128, act,
                                     tm, pc, rfx_stream(stream), (kid & 16384) != 0);
}

extern "C" int rfx_conv3x3_kernel_id(int N, int Cin, int Cout, int H, int W, int k_chunk) {
    return conv_kernel_id(N, Cin, Cout, 3, 3, 1, 1, H, W, true, k_chunk);
}

// The direct stride-2 kernel with the packed weights of rfx_conv3x3_f32 (bit 13 of rfx_conv2d_kernel_id says when it applies).
extern "C" int rfx_conv3x3_s2_f32(const float* in, const float* wP, const float* scale, const float* shift,
                                  const float* residual, float* out, int N, int Cin, int H, int W, int Cout, int act,
                                  void* stream) {
    if (!in || !wP || !out || N <= 0 || Cin < 8 || Cin % 8 != 0 || H <= 0 || W <= 0 || Cout <= 0) return RFX_E_ARG;
    if (reinterpret_cast<uintptr_t>(wP) & 15) return RFX_E_ARG;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int kid = conv_kernel_id(N, Cin, Cout, 3, 3, 2, 1, Ho, Wo, true);
    int tm = (kid & 8192) ? ((kid & 1) ? 1 : 2) : (Cout > 64 ? 2 : 1);
    if (const char* e = getenv("RFX_S2_TM")) tm = atoi(e) == 1 ? 1 : (atoi(e) == 2 ? 2 : tm);      // experiments
    return rfx_conv3x3_s2_launch(in, wP, scale, shift, residual, out, N, Cin, H, W, Cout, act, tm, rfx_stream(stream));
}

// dil = 1: rfx_conv2d_f32.  dil > 1 (rfx_conv2d_dilated_f32): the gather adds the ktab's (kh, kw) fields to the window origin
// as they are, so a dilated convolution is the SAME kernel with a table that holds kh*dil / kw*dil -- only the output size and the
// 4-bit field limit know about the dilation.
static int conv2d_impl(const float* in, const float* wT, const int32_t* ktab, const float* scale,
                       const float* shift, const float* residual, float* out, int N, int Cin, int Hin,
                       int Win, int Cout, int KH, int KW, int stride, int pad, int dil, int act, void* stream) {
    if (!in || !wT || !ktab || !out) return RFX_E_ARG;
    if (N <= 0 || Cin <= 0 || Hin <= 0 || Win <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0 || dil <= 0)
        return RFX_E_ARG;
    if ((KH - 1) * dil + 1 > 15 || (KW - 1) * dil + 1 > 15 || Cin >= (1 << 22)) return RFX_E_LIMIT;
    ConvArgs a;
    a.in = in; a.wT = wT; a.ktab = ktab; a.scale = scale; a.shift = shift; a.res = residual; a.out = out;
    a.N = N; a.Cin = Cin; a.Hin = Hin; a.Win = Win; a.Cout = Cout; a.stride = stride; a.pad = pad; a.act = act;
    a.Hout = (Hin + 2 * pad - ((KH - 1) * dil + 1)) / stride + 1;
    a.Wout = (Win + 2 * pad - ((KW - 1) * dil + 1)) / stride + 1;
    if (a.Hout <= 0 || a.Wout <= 0) return RFX_E_ARG;
    if ((long long)Cin * Hin * Win > 0x7fffffffLL) return RFX_E_LIMIT;
    const int K = Cin * KH * KW;
    a.Kpad = (K + 31) / 32 * 32;
    a.Mpad = (Cout + 127) / 128 * 128;
    a.P = (long long)N * a.Hout * a.Wout;
    hipStream_t st = rfx_stream(stream);
    const bool one = (KH == 1 && KW == 1 && pad == 0);
    // The wave-specialised form pays off where the gather is the heavy part and the K loop is long: KxK (K > 1)
    // convolutions on the 128x128 tile (measured +5 % there, -5...-15 % on 1x1 and 64-wide tiles, which keep the
    // single-role kernel with two independent workgroups per CU).  RFX_CONV_WS=0/1 forces it off/on for A/B runs.
    int kid = conv_kernel_id(N, Cin, Cout, KH, KW, stride, pad, a.Hout, a.Wout, false);
    if (reinterpret_cast<uintptr_t>(in) & 15) kid &= ~16;  // VECB needs 16-B aligned planes
    if (kid & 1024)
        return rfx_conv1x1_kmajor_launch(in, wT, scale, shift, residual, out, N, Cin, Hin * Win, Cout, a.Mpad, act,
                                         (kid & 3) ? 1 : 2, (kid & 16) != 0, st, (kid & 16384) != 0);
    const int variant = kid & 3;
    const bool ws = (kid & 8) != 0;
    const bool vecb = (kid & 16) != 0;
    if (vecb) {  // 1x1, stride 1, Hout*Wout % 4 == 0: vectorised pixel-side loads (never wave-specialised)
        switch (variant) {
            case 0: return launch_conv<2, 2, true, false, true>(a, st);
            case 1: return launch_conv<1, 2, true, false, true>(a, st);
            default: return launch_conv<1, 1, true, false, true>(a, st);
        }
    }
    switch (variant) {
        case 0:
            if (ws) return one ? launch_conv<2, 2, true, true>(a, st) : launch_conv<2, 2, false, true>(a, st);
            return one ? launch_conv<2, 2, true, false>(a, st) : launch_conv<2, 2, false, false>(a, st);
        case 1:
            if (ws) return one ? launch_conv<1, 2, true, true>(a, st) : launch_conv<1, 2, false, true>(a, st);
            return one ? launch_conv<1, 2, true, false>(a, st) : launch_conv<1, 2, false, false>(a, st);
        default: return one ? launch_conv<1, 1, true, false>(a, st) : launch_conv<1, 1, false, false>(a, st);
    }
}

extern "C" int rfx_conv2d_f32(const float* in, const float* wT, const int32_t* ktab, const float* scale,
                              const float* shift, const float* residual, float* out, int N, int Cin, int Hin,
                              int Win, int Cout, int KH, int KW, int stride, int pad, int act, void* stream) {
    return conv2d_impl(in, wT, ktab, scale, shift, residual, out, N, Cin, Hin, Win, Cout, KH, KW, stride, pad, 1, act, stream);
}

extern "C" int rfx_conv2d_dilated_f32(const float* in, const float* wT, const int32_t* ktab, const float* scale,
                                      const float* shift, const float* residual, float* out, int N, int Cin, int Hin,
                                      int Win, int Cout, int KH, int KW, int stride, int pad, int dilation, int act,
                                      void* stream) {
    if (dilation == 1) return RFX_E_ARG;      // the undilated geometries belong to rfx_conv2d_f32 / rfx_conv3x3_f32
    return conv2d_impl(in, wT, ktab, scale, shift, residual, out, N, Cin, Hin, Win, Cout, KH, KW, stride, pad, dilation, act, stream);
}
