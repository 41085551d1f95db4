// conv3x3.hip -- direct 3x3 / stride 1 / pad 1 convolution on the fp32 MFMA (the heaviest layer class of the path:
// ResNet-50 Bottleneck conv2 at stride 1, model/resnet50.py:75; every BasicBlock conv of the FeatureExtractor,
// model/model.py:32-35; the NetFlowCoarse / NetMatchability stacks, model/model.py:170-181).
//
// The implicit-GEMM kernel of conv.hip gathers its im2col operand element by element: ~10 address / bounds
// instructions per gathered float, which is what holds it at ~60 % of the matrix peak.  Here a workgroup owns an
// 8 x 16 patch of output pixels of ONE image and stages the RAW input patch (10 x 18 per channel, zero filled at the
// image border) in LDS once per 8 input channels; the MFMA B operand for tap (c, kh, kw) of pixel (r, x) is then the
// LDS word [c][r+kh][x+kw] -- fetched with one ds_read_b32 at a per-lane address that is precomputed ONCE (the 36
// k-pair offsets of a K step repeat every step).  No per-element index arithmetic is left in the loop:
//     per K step (72 = 8 channels x 9 taps): 9 float4 weight loads + 6 input loads per thread,
//     then per wavefront 18 ds_read_b128 (A) + 72 ds_read_b32 (B) + 144 MFMAs.
// A operand (weights): LDS image [k&1][m][k>>1] with 36 k-pairs per row (row stride 144 B: the 16-lane ds_read_b128
// groups hit 16 distinct bank quads without a swizzle because 9 is odd).  The host packs the weights ONCE in exactly
// that order (rfx_api.h: "wP"), one contiguous 36 KB image per (128-channel tile, K step), so staging is a straight
// 16-byte-per-lane copy: coalesced global loads, conflict-free ds_write_b128, no register transposes.
// The global loads of step s+1 are not issued as one burst: they are dealt out two at a time between the 16-MFMA
// chunks of step s, where they issue in the shadow of the matrix pipe (a burst in front of the MFMAs cost 13 % of
// the kernel, the bank-conflicting transposed stores another 8 %: scripts/ubench/conv_bench.py, make c3dbgN).
// B patch rows are 48 floats apart so that the two 16-pixel rows a 32-lane half reads fall on disjoint banks.
// Numerics: k runs channel-major, taps row-major -- the same order as conv.hip / the packed weight matrix, so the
// result is bit-identical to the implicit-GEMM kernel.
#include "common.h"
#include "conv_epilogue.h"
#include "group.h"
#include <stdlib.h>
#include <type_traits>

namespace {

// Output patch of a workgroup: 128 pixels as PT_R x PT_C = 8x16, 16x8 or 32x4 (the host picks the shape that wastes
// the fewest pixels on the image at hand: the trunk runs on 25x33 ... 36x48 maps at its small scales).  A 32-lane half
// of a B read covers 32/PT_C patch rows; the LDS row stride BS puts those rows on disjoint banks:
//   PT_C 16 -> 2 rows,  BS 48 (row offsets 0,16);  PT_C 8 -> 4 rows, BS 24 (0,24,16,8);  PT_C 4 -> 8 rows, BS 12 (0,12,24,4,16,28,8,20).
// The images of a batch are tiled as ONE tall map: image n occupies rows n*(H+1) .. n*(H+1)+H-1 of a stack whose row
// n*(H+1)+H is a virtual all-zero row -- at once the bottom halo of image n and the top halo of image n+1.  Patches tile
// the stack, so a patch may straddle two (or, on tiny maps, several) images and only the LAST patch row of the whole batch
// is ragged: a 25x33 map costs 26/25 instead of 32/25 in the row direction (the 7-level pyramid puts a quarter of the
// 3x3 work on such maps).  Outputs that fall on a virtual row are computed and dropped (1/(H+1) of the work).
// TN = 32-pixel MFMA sub-tiles per wavefront: 2 -> 128 pixels per workgroup; 4 -> 256 pixels (16 x 16 patch) for the layers
// whose output channels all fit one 64-channel tile (TM = 1): there the 18 KB weight image of a K step dwarfs the 6 KB input
// patch, and twice the pixels per workgroup halve the weight bytes a CU pulls per MFMA (the 64-channel layers sit on the CU's
// load path).  Same k order per output element: bit-identical to TN = 2.
template <int PT_C_, int TN_ = 2> struct Patch {
    static constexpr int PT_C = PT_C_, PT_R = 64 * TN_ / PT_C_;
    static constexpr int PR = PT_R + 2, PC = PT_C + 2;        // input patch incl. halo
    static constexpr int BS = PT_C_ == 16 ? 48 : (PT_C_ == 8 ? 24 : 12);
    static constexpr int RH = 32 / PT_C_;                     // patch rows per 32-pixel MFMA sub-tile
};
constexpr int CH = 8;                        // input channels per K step
constexpr int KS = CH * 9;                   // 72 k per step
constexpr int KK = KS / 2;                   // 36 k-pairs

struct C3Args {
    const float* in; const float* wT;   // wT: the packed image wP (rfx_api.h)
    const float* scale; const float* shift; const float* res; float* out;
    int N, Cin, H, W, Cout, act, Mpad;
    int tilesM, tilesH, tilesW;
    // FUSE: the 1x1 expansion that follows (Bottleneck conv3 + bn3 + residual + ReLU, model/resnet50.py:77-79,99-103)
    const float* wT3; const float* scale3; const float* shift3;
    int Cexp, Mpad3, act3;
    unsigned stagger;   // common.h: rfx_stagger
#ifdef RFX_TRACE
    long long* trace;
#endif
};
// experiments only (scripts/ubench/conv_bench.py): RFX_C3_DBG removes pieces of the main loop to price them
//   1 no LDS stores in the loop   2 no global loads in the loop   3 neither   4 neither, no barriers   6 no output stores
#ifndef RFX_C3_DBG
#define RFX_C3_DBG 0
#endif
// 1: the fused tail drains the epilogue of expansion pass p inside the K loop of pass p + 1 (round 5, VERDICT r4 #6: "the one structural
// conv idea still on the table"); 0 (default): the burst form.  MEASURED NEGATIVE on one box, same library otherwise
// (profiles/r05_conv_ab.jsonl, 128 / 64 images): 64-channel tail 101.4 -> 98.2 TFLOP/s at 120x160, 101.9 -> 98.4 at 240x320, 86.8 -> 82.4 at
// 100x132; 128-channel tail 110.0 -> 111.2 at 60x80, 86.7 -> 90.3 at 50x66 (four passes: three hidden epilogues).  The residual loads and
// the stores of a piece share the CU's in-order vector-memory path with the expansion's weight stream (2 x 16 bytes per lane and quad
// step from L2): spread over the K loop they delay the weights the next MFMAs wait for, which costs the 2-pass tail more than the one
// hidden epilogue saves.  Kept as a build flag (make exp NAME=inter SRC=conv3x3 DEFS=-DRFX_C3F_INTERLEAVE=1); bit-identical either way.
#ifndef RFX_C3F_INTERLEAVE
#define RFX_C3F_INTERLEAVE 0
#endif
#ifdef RFX_TRACE
#define RFX_STAMP(i) do { if (threadIdx.x == 0 && a.trace) a.trace[(size_t)blockIdx.x * 4 + (i)] = wall_clock64(); } while (0)
extern "C" long long* rfx_debug_trace_ptr();
#else
#define RFX_STAMP(i)
#endif

// FUSE = true: the workgroup's tile holds ALL channels of the 3x3 convolution (Cout == 64*TM); instead of going to HBM
// the tile (after bn + ReLU) becomes, in LDS, the B operand of the 1x1 expansion that follows it in a Bottleneck, and
// the workgroup writes relu(bn3(conv3(.)) + residual) for its 128 pixels and all Cexp channels.  The 64/128-channel
// intermediate never touches HBM, the expansion's launch, prologue and B-operand staging disappear, and its residual /
// output traffic overlaps the MFMA-bound 3x3 main loops of the neighbouring workgroups.  k order and pairing of the
// expansion are those of conv.hip: bit-identical to the two separate kernels.
// KCH > 0 (round 4): CHUNKED ACCUMULATION.  The fp32 MFMA adds a layer's K = 9 Cin products to ONE accumulator in k order; for
// K = 2304 / 4608 (ResNet-50 layer3 conv2, model/resnet50.py:75; NetFlowCoarse / NetMatchability conv2, model/model.py:172) that
// single chain carries 2.3x the round-off of the CPU reference's K-blocked sums (MKL / oneDNN register blocks of a few hundred
// k; profiles/r04_feature_error_*.json, r04_summation_order_model.json) -- the reason the device's arg-max near-ties flip more
// often than the reference's flip against itself.  With KCH the accumulators are added to a second set every KCH K steps
// (KCH * 72 k) and restarted from zero: total = ((c1 + c2) + c3) + ..., the same blocked sum.  The 64 extra registers come from
// the per-lane B-address table: baddr[kk] = pixb + (lrow ? c1[kk] : c0[kk]) with compile-time c0 / c1, i.e. ONE v_mad per
// k-pair (lrow * (c1 - c0) + pixb) and c0 as the immediate offset of the ds_read -- 36 VALU per K step instead of 36 registers.
// Dispatch (rfx_conv3x3_chunk_steps): every layer with K >= 2048, and -- round 5 -- the 3x3 convolution of a Bottleneck tail
// (model/resnet50.py:75 in layer1 / layer2: K = 576 / 1152), fused or not: the caller's k_chunk argument (rfx_conv3x3_f32) asks
// for the same chunks in the stand-alone kernel, so the fused tail stays bit-identical to its two-kernel form.
template <int TM, int PTC, bool FUSE, int TN = 2, bool RAG = false, int KCH = 0>
__device__ __forceinline__ void conv3x3_direct_body(const C3Args& a, const unsigned bx) {
    using G = Patch<PTC, TN>;
    static_assert(KCH == 0 || !RAG, "chunked accumulation: Cin % 8 == 0 instances only");
    // the 36-entry table of B offsets stays in registers where 2 x (TM * TN) accumulator tiles leave room for it (64-channel
    // tiles of 128 pixels: 32 + 32 accumulators); the other chunked instances recompute an offset per use (one v_mad)
    constexpr bool BTAB = KCH == 0 || TM * TN <= 2;
    static_assert(TN == 2 || (TN == 4 && TM == 1 && PTC == 16), "the 256-pixel patch is built for 64-channel tiles, 16 x 16");
    constexpr int NPX = 64 * TN;                     // output pixels of the workgroup
    constexpr int PT_R = G::PT_R, PT_C = G::PT_C, PR = G::PR, PC = G::PC, BS = G::BS, RH = G::RH;
    constexpr int NB = (CH * PR * PC + 255) / 256;   // patch elements per thread (6; 7 for the 32x4 patch)
    static_assert(PC < BS, "the surplus staging slots live in the never-read padding columns");
    constexpr int BM = 64 * TM;
    constexpr int AS_F = 2 * BM * KK, BS_F = CH * PR * BS, T2_F = FUSE ? BM * NPX : 0;
    constexpr int SMEM_F = AS_F + BS_F > T2_F ? AS_F + BS_F : T2_F;
    __shared__ __attribute__((aligned(16))) float smem[SMEM_F];   // FUSE: the main-loop buffers are reused for the mid tile
    float (*As)[BM][KK] = reinterpret_cast<float (*)[BM][KK]>(smem);
    float (*Bs)[PR][BS] = reinterpret_cast<float (*)[PR][BS]>(smem + AS_F);
    __shared__ float s_scale[BM], s_shift[BM];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = lane >> 5, lcol = lane & 31;

    const int tilesP = a.tilesH * a.tilesW;     // tilesH counts patch rows of the whole stack
    const int nwg = a.tilesM * tilesP;
    int bid = (int)bx;
    rfx_stagger(a.stagger, bx, 512u);
    RFX_STAMP(0);
    {   // XCD-aware bijective remap, m-tile fastest: the workgroups sharing one input patch sit on one L2
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, j = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int m0 = (bid % a.tilesM) * BM;
    const int pt = bid / a.tilesM;
    const int R0 = (pt / a.tilesW) * PT_R, ow0 = (pt % a.tilesW) * PT_C;    // first stack row / column of the patch
    const int Hs = a.H + 1, Rtot = a.N * Hs;                                 // stack period, stack height
    const size_t HW = (size_t)a.H * a.W;
    const int nbase = (R0 > 0 ? R0 - 1 : 0) / Hs;                            // image of the first input row of the patch
    const float* inn = a.in + (size_t)nbase * a.Cin * HW;                    // 32-bit patch offsets are relative to it

    if (t < BM) {
        const int m = m0 + t;
        s_scale[t] = (a.scale && m < a.Cout) ? a.scale[m] : 1.0f;
        s_shift[t] = (a.shift && m < a.Cout) ? a.shift[m] : 0.0f;
    }

    // ---- staging roles ----
    // weights: the packed image of this tile and step is A_F4 consecutive float4 (TM = 2: the whole 36 KB image;
    // TM = 1: the rows of this 64-channel half inside both parity planes); thread t copies float4 t, t+256, ...
    // (TM = 1: the 128 surplus slots of the last round re-copy float4 0..127 -- same value to the same place).
    constexpr int A_F4 = 2 * BM * KK / 4, NA = (A_F4 + 255) / 256;     // 2304 / 1152 float4, 9 / 5 per thread
    constexpr int PLANE_B = 128 * KK * 4;                              // bytes of one parity plane of a packed image
    unsigned aoff[NA];    // byte offset inside the packed image of a step
    int alds[NA];         // float4 index inside As
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        int L = t + 256 * j;
        if (L >= A_F4) L -= A_F4;
        alds[j] = L;
        aoff[j] = TM == 2 ? (unsigned)(L * 16) : (unsigned)((L / (A_F4 / 2)) * PLANE_B + (L % (A_F4 / 2)) * 16);
    }
    const size_t step_b = (size_t)2 * PLANE_B;                         // bytes per (tile, step)
    // Cin need not be a multiple of CH (the 49-channel correlation volume in front of the heads, model/model.py:213): the
    // packed weights carry zero rows for the missing channels of the last K step, and the patch slots of those channels are
    // zero filled (their loads point at the first channel of the step: a valid address) -- the sum gains exact zeros only.
    const int nsteps = (a.Cin + CH - 1) / CH;
    const int crem = a.Cin - (nsteps - 1) * CH;                        // channels of the last K step
    const bool ragged = RAG && crem != CH;                             // RAG: its own kernel instance (the host picks it when Cin % 8 != 0)
    const char* wtile = reinterpret_cast<const char*>(a.wT) + (size_t)(m0 / 128) * nsteps * step_b +
                        (TM == 1 ? ((m0 >> 6) & 1) * (PLANE_B / 2) : 0);
    // input patch: NB of the CH*PR*PC (1440 or 1632) patch elements per thread (the surplus slots land in the unused
    // columns of the last patch row, so that every load is consumed unconditionally: no divergent store).
    unsigned boffB[NB];   // byte offset inside one channel group (cl*HW + gy*W + gx)*4; 0 with bok=false -> zero
    bool bok[NB];
    unsigned clok = 0;    // bit u: the channel of slot u exists in the LAST K step
    int blds[NB];        // LDS float index
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const int idx = t + 256 * u;
        const bool real = idx < CH * PR * PC;
        const int cl = idx / (PR * PC), rem = idx - cl * (PR * PC);
        const int pr = rem / PC, px = rem - pr * PC;
        const int Rin = R0 - 1 + pr, gx = ow0 - 1 + px;                      // stack row -> (image, row); row H is virtual
        const int ni = (Rin >= 0 ? Rin : 0) / Hs, gy = Rin - ni * Hs;
        bok[u] = real && Rin >= 0 && Rin < Rtot && gy < a.H && (unsigned)gx < (unsigned)a.W;
        clok |= (cl < crem) ? (1u << u) : 0u;
        boffB[u] = bok[u] ? ((unsigned)((ni - nbase) * a.Cin + cl) * (unsigned)HW + (unsigned)(gy * a.W + gx)) * 4u : 0u;
        blds[u] = real ? (cl * PR + pr) * BS + px : ((CH - 1) * PR + PR - 1) * BS + PC + (idx - CH * PR * PC) % (BS - PC);
    }

    f32x4 ra[NA];
    float rb[NB];
    constexpr int NLD = NB + NA;                 // global loads per thread and step: patch first, then weights
    constexpr int LPC = 2;                       // ... dealt out LPC per 16-MFMA chunk
    static_assert(LPC * 8 >= NLD, "all loads of a step are issued before its last chunk");
    // `last` (wave-uniform): the tile is the ragged last K step -- slots of channels >= Cin load from offset 0 and store zeros
    auto load_one = [&](int id, const char* wstep, const char* base, bool last) {
        if (id < NB) rb[id] = *reinterpret_cast<const float*>(base + ((last && !((clok >> id) & 1u)) ? 0u : boffB[id]));   // masked at store time
        else if (id < NLD) ra[id - NB] = *reinterpret_cast<const f32x4*>(wstep + aoff[id - NB]);
    };
    auto load_global = [&](int s, bool last) {
        const char* wstep = wtile + (size_t)s * step_b;
        const char* base = reinterpret_cast<const char*>(inn + (size_t)s * CH * HW);
#pragma unroll
        for (int id = 0; id < NLD; ++id) load_one(id, wstep, base, last);
    };
    auto store_lds = [&](bool last) {
        f32x4* a4 = reinterpret_cast<f32x4*>(smem);
#pragma unroll
        for (int j = 0; j < NA; ++j) a4[alds[j]] = ra[j];
        float* bflat = &Bs[0][0][0];
#pragma unroll
        for (int u = 0; u < NB; ++u) bflat[blds[u]] = (bok[u] && (!last || ((clok >> u) & 1u))) ? rb[u] : 0.0f;
    };

    // ---- per-lane B addresses of the 36 k-pairs of a step (identical for every step) ----
    const int pixb = (wn * TN * RH + lcol / PT_C) * BS + lcol % PT_C;   // sub-tile j adds RH rows = RH*BS
    // patch offset of tap k = c*9 + kh*3 + kw (compile-time for a constant k)
    auto koff = [](int k) constexpr { return (k / 9) * (PR * BS) + ((k % 9) / 3) * BS + (k % 9) % 3; };
    int baddr[BTAB ? KK : 1];
    if constexpr (BTAB) {
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) baddr[kk] = pixb + (lrow ? koff(2 * kk + 1) : koff(2 * kk));
    }
    int pixb_cur = pixb;               // !BTAB: laundered once per K step so that the 36 step-invariant indices are NOT hoisted
    auto b_index = [&](int kk) {       // !BTAB: recomputed per use (one v_mad); else the table
        if constexpr (BTAB) return baddr[kk];
        else return pixb_cur + koff(2 * kk) + lrow * (koff(2 * kk + 1) - koff(2 * kk));
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    f32x16 tot[KCH ? TM : 1][KCH ? TN : 1];
    if constexpr (KCH > 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) tot[i][j][r] = 0.0f;
    }
    load_global(0, ragged && nsteps == 1);
    store_lds(ragged && nsteps == 1);
    __syncthreads();
    RFX_STAMP(1);
    const float* bflat = &Bs[0][0][0];
    for (int s = 0; s < nsteps; ++s) {
        const int sn = s + 1 < nsteps ? s + 1 : s;     // the last step re-loads its own tile: harmless
        const bool lastn = ragged && sn == nsteps - 1;
        const char* wstep = wtile + (size_t)sn * step_b;
        const char* base = reinterpret_cast<const char*>(inn + (size_t)sn * CH * HW);
        // LDS reads run one 4-k-pair chunk ahead of the MFMAs that consume them (register double buffer)
        f32x4 af[2][TM];
        float bv[2][4 * TN];
        if constexpr (!BTAB) { pixb_cur = pixb; asm volatile("" : "+v"(pixb_cur)); }
        auto read_chunk = [&](int q, int slot) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[slot][i] = *reinterpret_cast<const f32x4*>(&As[lrow][(wm * TM + i) * 32 + lcol][q * 4]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int j = 0; j < TN; ++j) bv[slot][TN * e + j] = bflat[b_index(q * 4 + e) + j * RH * BS];
            }
        };
        read_chunk(0, 0);
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const int cur = q & 1;
            if (q + 1 < 9) read_chunk(q + 1, cur ^ 1);
            if (RFX_C3_DBG != 2 && RFX_C3_DBG != 3 && RFX_C3_DBG != 4) {
#pragma unroll
                for (int i = 0; i < LPC; ++i) load_one(q * LPC + i, wstep, base, lastn);   // tile s+1, in the MFMA shadow
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i][e], bv[cur][TN * e + j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (KCH > 0) {
            if ((s + 1) % KCH == 0 || s + 1 == nsteps) {      // wave-uniform: close the chunk
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) { tot[i][j][r] += acc[i][j][r]; acc[i][j][r] = 0.0f; }
            }
        }
        if (RFX_C3_DBG != 4) __syncthreads();   // everyone is done reading the tile
        if (RFX_C3_DBG != 1 && RFX_C3_DBG != 3 && RFX_C3_DBG != 4)
            store_lds(lastn);   // tile s+1
        else if (RFX_C3_DBG == 1) {   // keep the loads alive (and waited for) without the LDS stores
#pragma unroll
            for (int j = 0; j < NA; ++j) asm volatile("" ::"v"(ra[j]));
#pragma unroll
            for (int u = 0; u < NB; ++u) asm volatile("" ::"v"(rb[u]));
        }
        if (RFX_C3_DBG != 4) __syncthreads();
    }
    if (RFX_C3_DBG == 6) {
        float sacc = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc += acc[i][j][r];
        if (sacc == 123.456f) a.out[t] = sacc;
        return;
    }

    RFX_STAMP(2);
    // ---- epilogue (conv_epilogue.h) ----
    const int Cfinal = FUSE ? a.Cexp : a.Cout;
    // element offset / existence of this lane's pixel in MFMA sub-tile st (32 consecutive pixels = RH patch rows) of the patch
    auto pixel_of = [&](int st, size_t& off, bool& ok) {
        const int R = R0 + st * RH + lcol / PT_C, ow = ow0 + lcol % PT_C;
        const int n = R / Hs, oh = R - n * Hs;
        ok = R < Rtot && oh < a.H && ow < a.W;
        off = ok ? (size_t)n * Cfinal * HW + (size_t)oh * a.W + ow : 0;
    };
    if constexpr (!FUSE) {
        size_t pix_off[TN];
        bool pix_ok[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) pixel_of(wn * TN + j, pix_off[j], pix_ok[j]);
        const bool full = m0 + BM <= a.Cout;
        if constexpr (KCH > 0)
            conv_epilogue<TM, TN, (TM > 1)>(tot, s_scale, s_shift, a.res, a.out, a.act, a.Cout, HW, m0, wm, lrow, pix_off, pix_ok, full);
        else
            conv_epilogue<TM, TN, (TM > 1)>(acc, s_scale, s_shift, a.res, a.out, a.act, a.Cout, HW, m0, wm, lrow, pix_off, pix_ok, full);
    } else {
        // ---- mid tile -> LDS: T2[channel][pixel], pixel = MFMA column numbering (wn*2 + j)*32 + lcol
        float* T2 = smem;   // every wavefront is past the last barrier of the main loop: As / Bs are free
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ch = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lrow;
                    float v;
                    if constexpr (KCH > 0) v = fmaf(tot[i][j][r], s_scale[ch], s_shift[ch]);
                    else v = fmaf(acc[i][j][r], s_scale[ch], s_shift[ch]);
                    if (a.act == RFX_ACT_RELU) v = v > 0.0f ? v : 0.0f;
                    T2[ch * NPX + (wn * TN + j) * 32 + lcol] = v;
                }
        __syncthreads();
        // ---- 1x1 expansion: out[Cexp x 128 px] = W3[Cexp x BM] * T2, 128 output channels per pass, waves 2 x 2.
        // The weights come in "quad" order wQ[q][lrow][m][4] = W3[m][8q + 2j + lrow] (j = 0..3): the four A operands a lane
        // needs for k-pairs 4q..4q+3 are ONE 16-byte load, and consecutive lanes (channels) read consecutive 16 bytes.
        // (a 256-pixel patch runs the expansion over its two 128-pixel halves one after the other: sub-tiles ph*4 + wn*2 + j)
        //
        // Round 5, INTERLEAVED EPILOGUE: a pass ends with 64 outputs per lane that each cost a residual load, a
        // fused multiply-add, an add, a max and a store -- 64 KB in + 64 KB out per pass and workgroup, issued as ONE burst
        // during which this workgroup's share of the matrix pipe idles (~10 us of a 15 us pass, scripts/dbg/fused_trace.py).
        // Here the finished accumulators of pass p stay in registers (`prev`) and are drained during the K loop of pass p + 1, a
        // PV-value piece per 16-MFMA quad step: the piece's residual loads are issued one step ahead (double buffer), its
        // arithmetic and stores run in the shadow of the step's MFMAs.  Only the LAST pass's epilogue is still exposed.  Same
        // operations per element in the same order: bit-identical.  Taken when the tail has a residual and a ReLU (every
        // Bottleneck of the trunk) and its folded bn3 vectors fit the 2 x 512-float LDS image; RFX_C3F_INTERLEAVE=0 (build
        // flag) keeps round 4's burst form for A/B timing.
        constexpr int NQ = BM / 8, AHEAD = 2;
        constexpr int PV = 64 / NQ;                                // outputs of the previous pass finished per quad step
        static_assert(16 % PV == 0, "a piece stays inside one 32x32 sub-tile");
        __shared__ float s_bn3[RFX_C3F_INTERLEAVE ? 1024 : 2];
        const bool inter = RFX_C3F_INTERLEAVE && a.Cexp <= 512 && a.Cexp >= 256 && a.res != nullptr && a.act3 == RFX_ACT_RELU;
        if (inter) {
            for (int i = t; i < a.Cexp; i += 256) { s_bn3[i] = a.scale3[i]; s_bn3[512 + i] = a.shift3[i]; }
            __syncthreads();
        }
#pragma unroll 1
        for (int ph = 0; ph < TN / 2; ++ph) {
        size_t pix_off[2];
        bool pix_ok[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) pixel_of(ph * 4 + wn * 2 + j, pix_off[j], pix_ok[j]);
        const float* t2col = T2 + lrow * NPX + ph * 128 + wn * 64 + lcol;    // + 2kk*NPX (+ 32 for the second sub-tile)
        f32x16 prev[2][2];
        float rres[2][PV];
        // element offset of output v (= sub-tile s = v >> 4 [i = s >> 1, j = s & 1], accumulator register r = v & 15) of the pass at mp
        auto out_off = [&](int v, int mp) {
            const int s_ = v >> 4, r = v & 15;
            return pix_off[s_ & 1] + (size_t)(mp + (wm * 2 + (s_ >> 1)) * 32 + 4 * lrow + (r & 3) + 8 * (r >> 2)) * HW;
        };
        auto piece_loads = [&](int piece, int slot, int pmp) {
#pragma unroll
            for (int u = 0; u < PV; ++u) rres[slot][u] = a.res[out_off(piece * PV + u, pmp)];
        };
        auto piece_finish = [&](int piece, int slot, int pmp) {
            float x[PV];
#pragma unroll
            for (int u = 0; u < PV; ++u) {
                const int v = piece * PV + u, s_ = v >> 4, r = v & 15;
                const int ch = pmp + (wm * 2 + (s_ >> 1)) * 32 + 4 * lrow + (r & 3) + 8 * (r >> 2);
                x[u] = fmaf(prev[s_ >> 1][s_ & 1][r], s_bn3[ch], s_bn3[512 + ch]);
                x[u] += rres[slot][u];
                x[u] = x[u] > 0.0f ? x[u] : 0.0f;
            }
            if (pix_ok[((piece * PV) >> 4) & 1]) {
#pragma unroll
                for (int u = 0; u < PV; ++u) a.out[out_off(piece * PV + u, pmp)] = x[u];
            }
        };
        // one pass: K loop over the mid tile; HAVE_PREV: the previous pass (at pmp) is drained piece by piece inside it
        auto run_pass = [&](auto have_prev, int mp, int pmp, f32x16 (&acc2)[2][2]) {
            constexpr bool HP = decltype(have_prev)::value;
            const f32x4* wq0 = reinterpret_cast<const f32x4*>(a.wT3) + (size_t)lrow * a.Cexp + mp + wm * 64 + lcol;   // + q*2*Cexp (+ 32)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.0f;
            // L2 -> registers, two quads (8 k-pairs = 32 MFMAs) ahead of use
            // (measured, round 3, profiles/r03_fused_tail_expansion_experiment.jsonl + r03_fused_tail_late_prefetch.jsonl: 4 quads
            // ahead +-0; the pass's residual values fetched in front of this loop -6 % (vmcnt retires in order: the loop's weight
            // waits then wait for the residuals too); fetched inside the loop behind its last weight load -3 % on the 64-channel
            // tail, +1 % on the 128-channel one.  The epilogue's cost is not the latency of its residual loads.)
            f32x4 wq[AHEAD + 1][2];
            auto load_w = [&](int q, int slot) {
                wq[slot][0] = wq0[(size_t)q * 2 * a.Cexp];
                wq[slot][1] = wq0[(size_t)q * 2 * a.Cexp + 32];
            };
            // mid-tile operands from LDS run one quad ahead of the MFMAs as well (register double buffer)
            float bq[2][4][2];
            auto read_b = [&](int q, int slot) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bq[slot][e][0] = t2col[2 * (q * 4 + e) * NPX];
                    bq[slot][e][1] = t2col[2 * (q * 4 + e) * NPX + 32];
                }
            };
#pragma unroll
            for (int q = 0; q < AHEAD && q < NQ; ++q) load_w(q, q);
            read_b(0, 0);
            if constexpr (HP) piece_loads(0, 0, pmp);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (q + AHEAD < NQ) load_w(q + AHEAD, (q + AHEAD) % (AHEAD + 1));
                if (q + 1 < NQ) read_b(q + 1, (q + 1) & 1);
                if constexpr (HP) { if (q + 1 < NQ) piece_loads(q + 1, (q + 1) & 1, pmp); }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float b0 = bq[q & 1][e][0], b1 = bq[q & 1][e][1];
                    const float w0 = wq[q % (AHEAD + 1)][0][e], w1 = wq[q % (AHEAD + 1)][1][e];
                    acc2[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0, b0, acc2[0][0], 0, 0, 0);
                    acc2[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0, b1, acc2[0][1], 0, 0, 0);
                    acc2[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1, b0, acc2[1][0], 0, 0, 0);
                    acc2[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1, b1, acc2[1][1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (HP) piece_finish(q, q & 1, pmp);     // VALU + stores of the piece: issued behind the step's MFMAs
            }
        };
        if (inter) {
            f32x16 acc2[2][2];
            run_pass(std::false_type{}, 0, 0, acc2);
            for (int mp = 128; mp < a.Cexp; mp += 128) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) prev[i][j] = acc2[i][j];
                run_pass(std::true_type{}, mp, mp - 128, acc2);
            }
            // the last pass's epilogue is the only exposed one
            const int lp = a.Cexp - 128;
            conv_epilogue<2, 2, false>(acc2, a.scale3 + lp, a.shift3 + lp, a.res, a.out, a.act3, a.Cexp, HW, lp, wm, lrow, pix_off, pix_ok, true);
        } else {
            for (int mp = 0; mp < a.Cexp; mp += 128) {
                f32x16 acc2[2][2];
                run_pass(std::false_type{}, mp, 0, acc2);
                // scale3 / shift3 are read straight from global memory (L1 hits): conv_epilogue only indexes the pointers
                conv_epilogue<2, 2, false>(acc2, a.scale3 + mp, a.shift3 + mp, a.res, a.out, a.act3, a.Cexp, HW, mp, wm, lrow, pix_off,
                                           pix_ok, true);
            }
        }
        }   // pixel halves
    }
#ifdef RFX_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RFX_STAMP(3);
#endif
}

// (three workgroups per CU for the 64-channel fused tail -- 168 VGPRs, 4 spilled -- measured -2.4 %: profiles/r03_fused_tail_3wgs.jsonl)
template <int TM, int PTC, bool FUSE = false, int TN = 2, bool RAG = false, int KCH = 0>
__global__ __launch_bounds__(256, 2) void conv3x3_direct_kernel(C3Args a) {
    conv3x3_direct_body<TM, PTC, FUSE, TN, RAG, KCH>(a, blockIdx.x);
}

// grouped form (group.h): blockIdx.y = problem, the same body on that problem's argument block
template <int TM, int PTC, bool FUSE = false, int KCH = 0>
__global__ __launch_bounds__(256, 2) void conv3x3_direct_group_kernel(RfxGroupArgs<C3Args> g) {
    const unsigned y = blockIdx.y;
    if (blockIdx.x >= g.gx[y]) return;
    conv3x3_direct_body<TM, PTC, FUSE, 2, false, KCH>(g.p[y], blockIdx.x);
}

template <int TM, int PTC, bool FUSE, int KCH = 0>
static int c3_group_launch(const void* blob, const unsigned* gx, int n, hipStream_t st) {
    return rfx_group_launch_impl<C3Args>(conv3x3_direct_group_kernel<TM, PTC, FUSE, KCH>, 256, blob, gx, n, st);
}


// ======================================================================================================================
// Round 4: the direct form for 3x3 / STRIDE 2 / pad 1 (ResNet-50 layer2.0 / layer3.0 conv2, model/resnet50.py:75; the first
// convolution of FeatureExtractor layer2 / layer3, model/model.py:86-95) -- 10 % of the config-3 step on the implicit-GEMM
// kernel at 95 TFLOP/s.  Same scheme as above: a workgroup owns an 8 x 16 patch of OUTPUT pixels of one image and stages the
// raw 17 x 33 input patch of 8 channels per K step; the stride-2 taps of 16 consecutive output pixels would hit every second
// LDS word (2-way bank conflicts), so a patch row is stored de-interleaved, [even columns 0..16 | odd columns at +20], 40 floats
// per row: tap (kh, kw) of output pixel (r, x) is the word (2r + kh) * 40 + (kw & 1) * 20 + x + (kw >> 1) -- consecutive
// pixels read consecutive words, and the two patch rows of a 32-lane half lie 80 words = 16 banks apart.  Weights: the same
// packed images wP as the stride-1 kernel (rfx_api.h), same k order as conv.hip -> bit-identical results.  Images are tiled
// one by one (the stacked-batch trick above needs stride 1).  Cin % 8 == 0.
namespace s2 {
constexpr int PT_R = 8, PT_C = 16, PRI = 2 * PT_R + 1, PCI = 2 * PT_C + 1, PH = 20, BS2 = 40, RH = 2;
constexpr int NEL = CH * PRI * PCI;                  // 4488 input elements per K step
constexpr int NB = (NEL + 255) / 256;                // 18 per thread (the last round is partial)
}  // namespace s2

// KCH > 0 (round 5): chunked accumulation as in conv3x3_direct_body -- ResNet-50 layer2.0 / layer3.0 conv2 (K = 1152 / 2304) and the
// FeatureExtractor's 128 -> 256 strided convolution were the last long-K 3x3 layers summed as ONE fma chain.  The B-offset table is
// recomputed per use there (one v_mad per k-pair) to make room for the second accumulator set.
template <int TM, int KCH = 0>
__device__ __forceinline__ void conv3x3_s2_body(const C3Args& a, const unsigned bx) {
    using namespace s2;
    constexpr int TN = 2, BM = 64 * TM;
    constexpr int AS_F = 2 * BM * KK, BS_F = CH * PRI * BS2;
    __shared__ __attribute__((aligned(16))) float smem[AS_F + BS_F];
    float (*As)[BM][KK] = reinterpret_cast<float (*)[BM][KK]>(smem);
    float* bflat = smem + AS_F;                       // [CH][PRI][BS2]
    __shared__ float s_scale[BM], s_shift[BM];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = lane >> 5, lcol = lane & 31;
    const int Ho = (a.H - 1) / 2 + 1, Wo = (a.W - 1) / 2 + 1;
    const int tilesP = a.tilesH * a.tilesW;           // patches per image
    const int nwg = a.tilesM * tilesP * a.N;
    int bid = (int)bx;
    {   // XCD-aware bijective remap, m-tile fastest
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, j = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int m0 = (bid % a.tilesM) * BM;
    int pt = bid / a.tilesM;
    const int n = pt / tilesP;
    pt -= n * tilesP;
    const int oh0 = (pt / a.tilesW) * PT_R, ow0 = (pt % a.tilesW) * PT_C;
    const size_t HW = (size_t)a.H * a.W, HWo = (size_t)Ho * Wo;
    const float* inn = a.in + (size_t)n * a.Cin * HW;

    if (t < BM) {
        const int m = m0 + t;
        s_scale[t] = (a.scale && m < a.Cout) ? a.scale[m] : 1.0f;
        s_shift[t] = (a.shift && m < a.Cout) ? a.shift[m] : 0.0f;
    }
    // ---- staging roles: weights exactly as in conv3x3_direct_body
    constexpr int A_F4 = 2 * BM * KK / 4, NA = (A_F4 + 255) / 256;
    constexpr int PLANE_B = 128 * KK * 4;
    unsigned aoff[NA];
    int alds[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        int L = t + 256 * j;
        if (L >= A_F4) L -= A_F4;
        alds[j] = L;
        aoff[j] = TM == 2 ? (unsigned)(L * 16) : (unsigned)((L / (A_F4 / 2)) * PLANE_B + (L % (A_F4 / 2)) * 16);
    }
    const size_t step_b = (size_t)2 * PLANE_B;
    const int nsteps = a.Cin / CH;
    const char* wtile = reinterpret_cast<const char*>(a.wT) + (size_t)(m0 / 128) * nsteps * step_b +
                        (TM == 1 ? ((m0 >> 6) & 1) * (PLANE_B / 2) : 0);
    // input patch = 136 rows (R = channel * 17 + patch row) x 33 columns per K step.  Thread t stages column px = t & 31 of
    // the rows R = (t >> 5) + 8 u, u = 0..16 (8 x 17 = 136 exactly) and, for t < 136, column 32 of row t -- 18 loads, and NO
    // per-element address tables: the LDS word of (R, px) is R * BS2 + f(px), affine in u (an immediate of the store), and the
    // global offset is base + u * 8 W + cl(u) * (HW - 17 W) with cl(u) = c0(u) + [rg >= th(u)], c0 / th compile-time constants
    // (the tables cost 36 registers and made the 128-channel instance spill).
    const int rg = t >> 5, px = t & 31;
    const int gx = 2 * ow0 - 1 + px, gy0 = 2 * oh0 - 1;
    const bool xok = (unsigned)gx < (unsigned)a.W;
    const unsigned D4 = (unsigned)(HW - (size_t)PRI * a.W) * 4u;            // channel step minus 17 rows, bytes
    const unsigned W4 = (unsigned)a.W * 4u;
    // base of row R = rg (channel 0, patch row rg): may point outside the image (masked by bokm); keep it inside the tensor
    const unsigned base_b = (unsigned)((gy0 + rg) * a.W + gx) * 4u;
    unsigned bokm = 0, clsel = 0;
    constexpr int NBR = 17;
#pragma unroll
    for (int u = 0; u < NBR; ++u) {
        const int R = rg + 8 * u, cl = R / PRI, pr = R - cl * PRI;
        const int gy = gy0 + pr;
        bokm |= (xok && (unsigned)gy < (unsigned)a.H) ? (1u << u) : 0u;
        clsel |= (cl != (8 * u) / PRI) ? (1u << u) : 0u;                      // cl(u) = c0(u) + 1 for this thread's rg
    }
    const int blds0 = rg * BS2 + (px & 1) * PH + (px >> 1);                  // + u * 8 * BS2
    // column 32 (t < 136): row R = t
    const bool has33 = t < CH * PRI;
    const int cl33 = (has33 ? t : 0) / PRI, pr33 = (has33 ? t : 0) - cl33 * PRI;
    const int gx33 = 2 * ow0 - 1 + 32, gy33 = gy0 + pr33;
    const bool ok33 = has33 && (unsigned)gx33 < (unsigned)a.W && (unsigned)gy33 < (unsigned)a.H;
    const unsigned boff33 = ok33 ? ((unsigned)cl33 * (unsigned)HW + (unsigned)(gy33 * a.W + gx33)) * 4u : 0u;
    const int blds33 = (has33 ? t : 0) * BS2 + 16;                           // even column 32 -> slot 16
    f32x4 ra[NA];
    float rb[NB];
    constexpr int NLD = NB + NA;
    constexpr int LPC = 3;
    static_assert(NB == NBR + 1, "17 regular rows + the 33rd column");
    static_assert(LPC * 9 >= NLD, "all loads of a step are issued inside its 9 chunks");
    auto load_one = [&](int id, const char* wstep, const char* base) {
        if (id < NBR) {
            const bool ok = (bokm >> id) & 1u;
            // byte offset relative to the channel group's plane 0 (mod 2^32: a valid element's offset is non-negative and below
            // 4 GB, checked at launch); invalid elements read offset 0 (masked at store time).  The empty asm keeps the compiler
            // from hoisting the 17 step-invariant offsets out of the K loop into registers -- the point of having no tables.
            unsigned bo = base_b;
            asm volatile("" : "+v"(bo));
            const unsigned off = bo + (unsigned)id * 8u * W4 + (unsigned)((8 * id) / PRI) * D4 + (((clsel >> id) & 1u) ? D4 : 0u);
            rb[id] = *reinterpret_cast<const float*>(base + (ok ? off : 0u));
        } else if (id == NBR) {
            rb[id] = *reinterpret_cast<const float*>(base + boff33);
        } else if (id < NLD) {
            ra[id - NB] = *reinterpret_cast<const f32x4*>(wstep + aoff[id - NB]);
        }
    };
    auto store_lds = [&]() {
        f32x4* a4 = reinterpret_cast<f32x4*>(smem);
#pragma unroll
        for (int j = 0; j < NA; ++j) a4[alds[j]] = ra[j];
#pragma unroll
        for (int u = 0; u < NBR; ++u) bflat[blds0 + u * 8 * BS2] = ((bokm >> u) & 1u) ? rb[u] : 0.0f;
        if (has33) bflat[blds33] = ok33 ? rb[NBR] : 0.0f;
    };
    // ---- per-lane B addresses of the 36 k-pairs of a step
    const int pixb = (2 * (wn * TN * RH + lcol / PT_C)) * BS2 + lcol % PT_C;            // sub-tile j adds 2*RH rows of the patch
    auto koff2 = [](int k) constexpr { return (k / 9) * (PRI * BS2) + ((k % 9) / 3) * BS2 + (((k % 9) % 3) & 1) * PH + (((k % 9) % 3) >> 1); };
    constexpr bool BTAB = KCH == 0 || TM == 1;
    int baddr[BTAB ? KK : 1];
    if constexpr (BTAB) {
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) baddr[kk] = pixb + (lrow ? koff2(2 * kk + 1) : koff2(2 * kk));
    }
    int pixb_cur = pixb;
    auto b_index = [&](int kk) {
        if constexpr (BTAB) return baddr[kk];
        else return pixb_cur + koff2(2 * kk) + lrow * (koff2(2 * kk + 1) - koff2(2 * kk));
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    f32x16 tot[KCH ? TM : 1][KCH ? TN : 1];
    if constexpr (KCH > 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) tot[i][j][r] = 0.0f;
    }
    {
        const char* base = reinterpret_cast<const char*>(inn);
#pragma unroll
        for (int id = 0; id < NLD; ++id) load_one(id, wtile, base);
    }
    store_lds();
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const int sn = s + 1 < nsteps ? s + 1 : s;
        const char* wstep = wtile + (size_t)sn * step_b;
        const char* base = reinterpret_cast<const char*>(inn + (size_t)sn * CH * HW);
        f32x4 af[2][TM];
        float bv[2][4 * TN];
        if constexpr (!BTAB) { pixb_cur = pixb; asm volatile("" : "+v"(pixb_cur)); }
        auto read_chunk = [&](int q, int slot) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[slot][i] = *reinterpret_cast<const f32x4*>(&As[lrow][(wm * TM + i) * 32 + lcol][q * 4]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < TN; ++j) bv[slot][TN * e + j] = bflat[b_index(q * 4 + e) + j * 2 * RH * BS2];
        };
        read_chunk(0, 0);
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const int cur = q & 1;
            if (q + 1 < 9) read_chunk(q + 1, cur ^ 1);
#pragma unroll
            for (int i = 0; i < LPC; ++i) load_one(q * LPC + i, wstep, base);            // tile s+1, in the MFMA shadow
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i][e], bv[cur][TN * e + j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (KCH > 0) {
            if ((s + 1) % KCH == 0 || s + 1 == nsteps) {      // wave-uniform: close the chunk
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) { tot[i][j][r] += acc[i][j][r]; acc[i][j][r] = 0.0f; }
            }
        }
        __syncthreads();
        store_lds();
        __syncthreads();
    }
    size_t pix_off[TN];
    bool pix_ok[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int oh = oh0 + (wn * TN + j) * RH + lcol / PT_C, ow = ow0 + lcol % PT_C;
        pix_ok[j] = oh < Ho && ow < Wo;
        pix_off[j] = pix_ok[j] ? (size_t)n * a.Cout * HWo + (size_t)oh * Wo + ow : 0;
    }
    if constexpr (KCH > 0)
        conv_epilogue<TM, TN, (TM > 1)>(tot, s_scale, s_shift, a.res, a.out, a.act, a.Cout, HWo, m0, wm, lrow, pix_off, pix_ok, m0 + BM <= a.Cout);
    else
        conv_epilogue<TM, TN, (TM > 1)>(acc, s_scale, s_shift, a.res, a.out, a.act, a.Cout, HWo, m0, wm, lrow, pix_off, pix_ok, m0 + BM <= a.Cout);
}

template <int TM, int KCH = 0>
__global__ __launch_bounds__(256, 2) void conv3x3_s2_kernel(C3Args a) {
    conv3x3_s2_body<TM, KCH>(a, blockIdx.x);
}

// grouped form (group.h; round 5): blockIdx.y = problem -- the single-pair trunk pass keeps ONE launch per layer, and a strided
// layer runs the same kernel (hence the same chunked sums) whatever the batch it rides in
template <int TM, int KCH = 0>
__global__ __launch_bounds__(256, 2) void conv3x3_s2_group_kernel(RfxGroupArgs<C3Args> g) {
    const unsigned y = blockIdx.y;
    if (blockIdx.x >= g.gx[y]) return;
    conv3x3_s2_body<TM, KCH>(g.p[y], blockIdx.x);
}

template <int TM, int KCH>
static int c3s2_group_launch(const void* blob, const unsigned* gx, int n, hipStream_t st) {
    return rfx_group_launch_impl<C3Args>(conv3x3_s2_group_kernel<TM, KCH>, 256, blob, gx, n, st);
}

}  // namespace

// Chunked accumulation (KCH above) for the layers whose single fma chain is longest: K = 9 Cin >= 2048.  RFX_C3_CHUNK=0 turns it off
// (A/B runs: the chain form is what rfx_conv2d_f32's implicit-GEMM kernel computes).
bool rfx_conv3x3_chunked(int Cin) {
    static const int en = getenv("RFX_C3_CHUNK") ? atoi(getenv("RFX_C3_CHUNK")) : 1;
    return en && Cin % CH == 0 && Cin * 9 >= 2048;
}
// Round 5: the stride-2 kernel's long-K layers (K >= 1152) -- RFX_C3_S2_CHUNK=0: chains
bool rfx_conv3x3_s2_chunked(int Cin) {
    static const int en = getenv("RFX_C3_S2_CHUNK") ? atoi(getenv("RFX_C3_S2_CHUNK")) : 1;
    return en && Cin * 9 >= 1152;
}
// Round 5: the Bottleneck tails (K = 576 / 1152) close their chunks as well -- RFX_C3_TAIL_CHUNK=0: the round-4 chains (A/B runs)
bool rfx_conv3x3_tail_chunked() {
    static const int en = getenv("RFX_C3_TAIL_CHUNK") ? atoi(getenv("RFX_C3_TAIL_CHUNK")) : 1;
    return en != 0;
}

// 256-pixel (16 x 16) patches for a layer whose output channels fit ONE 64-channel tile: only for launches that still fill the
// chip two generations deep with the larger patch, never inside a grouped launch (latency-bound: more, smaller workgroups win).
bool rfx_conv3x3_wide_patch(int N, int H, int W, int Cout, int patch_cols) {
    static const int en = getenv("RFX_C3_WIDE") ? atoi(getenv("RFX_C3_WIDE")) : 1;
    if (!en || rfx_group_recording() || patch_cols != 16 || Cout > 64) return false;
    const long long tiles = (((long long)N * (H + 1) + 15) / 16) * ((W + 15) / 16);
    return tiles >= 1024;
}

// Patch shape for a batch of N H x W maps stacked as above: the fewest padded pixels (ties: the widest, whose row segments
// coalesce best).  The fused Bottleneck tail writes Cexp channels per pixel from a narrow patch in short row segments:
// measured on equal work its 16x8 patch is ~9 % and its 32x4 patch ~40 % slower than 8x16, while the plain 3x3 kernel is
// indifferent (scripts/ubench/conv_bench.py on the 25x33 ... 112x148 maps of the pyramid) -- hence the weights.
int rfx_conv3x3_patch_cols(int N, int H, int W, bool fused) {
    // grouped launches (group.h): ONE patch shape for every problem of the group, so that a layer is one launch and not
    // one per shape -- the group is latency-bound, a few padded pixels on the small maps cost less than a serial launch
    static const int uniform = getenv("RFX_GROUP_UNIFORM") ? atoi(getenv("RFX_GROUP_UNIFORM")) : 1;
    if (uniform && rfx_group_recording()) return 16;
    int best = 16;
    long long best_cost = -1;
    for (int pc = 16; pc >= 4; pc >>= 1) {
        const int pr = 128 / pc;
        const long long area = (((long long)N * (H + 1) + pr - 1) / pr) * ((W + pc - 1) / pc);
        const long long cost = area * (!fused || pc == 16 ? 100 : (pc == 8 ? 109 : 140));
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = pc; }
    }
    return best;
}

template <int TM, int PTC, bool FUSE = false, int TN = 2, bool RAG = false, int KCH = 0>
static int launch_direct(C3Args& a, hipStream_t st) {
    using G = Patch<PTC, TN>;
    const long long rows = (long long)a.N * (a.H + 1);
    a.tilesH = (int)((rows + G::PT_R - 1) / G::PT_R);
    a.tilesW = (a.W + G::PT_C - 1) / G::PT_C;
    const long long nwg = (long long)a.tilesM * a.tilesH * a.tilesW;
    if (nwg > 0x7fffffffLL || rows > 0x7fffffffLL) return RFX_E_LIMIT;
    // 32-bit byte offsets inside the images one input patch can touch
    const long long span = (G::PR + a.H) / (a.H + 1) + 1;
    if (span * a.Cin * a.H * a.W * 4 > 0xffffffffLL) return RFX_E_LIMIT;
    static const unsigned stagger = FUSE ? rfx_stagger_env("RFX_C3F_STAGGER", "RFX_C3F_STAGGER_MODE")
                                         : rfx_stagger_env("RFX_C3_STAGGER", "RFX_C3_STAGGER_MODE");
    a.stagger = (rfx_group_recording() || nwg < 1024) ? 0u : stagger;
    if constexpr (TN == 2 && !RAG) {
        if (rfx_group_recording()) return rfx_group_record(&c3_group_launch<TM, PTC, FUSE, KCH>, &a, sizeof(a), (unsigned)nwg);
    }
    hipLaunchKernelGGL((conv3x3_direct_kernel<TM, PTC, FUSE, TN, RAG, KCH>), dim3((unsigned)nwg), dim3(256), 0, st, a);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

// Internal entry used by rfx_conv2d_f32 (conv.hip).  Preconditions checked by the caller: 3x3, stride 1, pad 1,
// Cin % 8 == 0.  tm = 2 -> 128 output channels per workgroup, tm = 1 -> 64; patch_cols in {16, 8, 4}.
int rfx_conv3x3_direct_launch(const float* in, const float* wP, const float* scale, const float* shift,
                              const float* residual, float* out, int N, int Cin, int H, int W, int Cout, int Mpad,
                              int act, int tm, int patch_cols, hipStream_t st, bool chunked) {
    C3Args a;
    a.in = in; a.wT = wP; a.scale = scale; a.shift = shift; a.res = residual; a.out = out;
    a.N = N; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.act = act; a.Mpad = Mpad;
    a.wT3 = a.scale3 = a.shift3 = nullptr; a.Cexp = a.Mpad3 = a.act3 = 0;
#ifdef RFX_TRACE
    a.trace = rfx_debug_trace_ptr();
#endif
    const int BM = 64 * tm;
    a.tilesM = (Cout + BM - 1) / BM;
    if (Cin % CH != 0) {          // ragged last K step: separate instances (never recorded into a grouped launch: launched at once)
        if (tm == 2) {
            if (patch_cols == 16) return launch_direct<2, 16, false, 2, true>(a, st);
            if (patch_cols == 8) return launch_direct<2, 8, false, 2, true>(a, st);
            return launch_direct<2, 4, false, 2, true>(a, st);
        }
        if (patch_cols == 16) return launch_direct<1, 16, false, 2, true>(a, st);
        if (patch_cols == 8) return launch_direct<1, 8, false, 2, true>(a, st);
        return launch_direct<1, 4, false, 2, true>(a, st);
    }
    if (tm == 2) {
        if (chunked) {      // K >= 2048 (or asked for by the caller): chunks of 4 K steps (288 k)
            if (patch_cols == 16) return launch_direct<2, 16, false, 2, false, 4>(a, st);
            if (patch_cols == 8) return launch_direct<2, 8, false, 2, false, 4>(a, st);
            return launch_direct<2, 4, false, 2, false, 4>(a, st);
        }
        if (patch_cols == 16) return launch_direct<2, 16>(a, st);
        if (patch_cols == 8) return launch_direct<2, 8>(a, st);
        return launch_direct<2, 4>(a, st);
    }
    if (chunked) {          // a single image / tiny batch runs the long-K layers on 64-channel tiles: same chunks
        if (patch_cols == 16) return rfx_conv3x3_wide_patch(N, H, W, Cout, 16) ? launch_direct<1, 16, false, 4, false, 4>(a, st)
                                                                               : launch_direct<1, 16, false, 2, false, 4>(a, st);
        if (patch_cols == 8) return launch_direct<1, 8, false, 2, false, 4>(a, st);
        return launch_direct<1, 4, false, 2, false, 4>(a, st);
    }
    if (patch_cols == 16) return rfx_conv3x3_wide_patch(N, H, W, Cout, 16) ? launch_direct<1, 16, false, 4>(a, st) : launch_direct<1, 16>(a, st);
    if (patch_cols == 8) return launch_direct<1, 8>(a, st);
    return launch_direct<1, 4>(a, st);
}

// 3x3 / stride 2 / pad 1, Cin % 8 == 0: the direct stride-2 kernel above.  tm = 2 -> 128 output channels per workgroup, 1 -> 64.
int rfx_conv3x3_s2_launch(const float* in, const float* wP, const float* scale, const float* shift, const float* residual,
                          float* out, int N, int Cin, int H, int W, int Cout, int act, int tm, hipStream_t st) {
    C3Args a;
    a.in = in; a.wT = wP; a.scale = scale; a.shift = shift; a.res = residual; a.out = out;
    a.N = N; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.act = act; a.Mpad = (Cout + 127) / 128 * 128;
    a.wT3 = a.scale3 = a.shift3 = nullptr; a.Cexp = a.Mpad3 = a.act3 = 0; a.stagger = 0;
#ifdef RFX_TRACE
    a.trace = nullptr;
#endif
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int BM = 64 * tm;
    a.tilesM = (Cout + BM - 1) / BM;
    a.tilesH = (Ho + s2::PT_R - 1) / s2::PT_R;
    a.tilesW = (Wo + s2::PT_C - 1) / s2::PT_C;
    const long long nwg = (long long)a.tilesM * a.tilesH * a.tilesW * N;
    if (nwg > 0x7fffffffLL) return RFX_E_LIMIT;
    if ((long long)Cin * H * W * 4 > 0xffffffffLL) return RFX_E_LIMIT;                    // 32-bit byte offsets inside one image
    // round 5: K = 9 Cin >= 1152 closes a chunk every 4 K steps (288 products), like the stride-1 kernel (RFX_C3_S2_CHUNK=0: chains)
    const bool chk = rfx_conv3x3_s2_chunked(Cin);
    if (rfx_group_recording()) {
        if (chk) return tm == 2 ? rfx_group_record(&c3s2_group_launch<2, 4>, &a, sizeof(a), (unsigned)nwg)
                                : rfx_group_record(&c3s2_group_launch<1, 4>, &a, sizeof(a), (unsigned)nwg);
        return tm == 2 ? rfx_group_record(&c3s2_group_launch<2, 0>, &a, sizeof(a), (unsigned)nwg)
                       : rfx_group_record(&c3s2_group_launch<1, 0>, &a, sizeof(a), (unsigned)nwg);
    }
    if (chk) {
        if (tm == 2) hipLaunchKernelGGL((conv3x3_s2_kernel<2, 4>), dim3((unsigned)nwg), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((conv3x3_s2_kernel<1, 4>), dim3((unsigned)nwg), dim3(256), 0, st, a);
    } else if (tm == 2) hipLaunchKernelGGL((conv3x3_s2_kernel<2>), dim3((unsigned)nwg), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv3x3_s2_kernel<1>), dim3((unsigned)nwg), dim3(256), 0, st, a);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

// Bottleneck tail in one kernel: out = act3(bn3(conv1x1(act2(bn2(conv3x3(in))))) + residual)  (model/resnet50.py:71-79,93-103).
// The 3x3 convolution must be direct-eligible (stride 1, pad 1, Cin % 8 == 0) with Cmid in {64, 128} so that one workgroup
// tile holds all of its channels; Cexp % 128 == 0.
// Kernel instance rfx_conv3x3_conv1x1_f32 launches (for a profiler-side caller, as rfx_conv2d_kernel_id): bit 9 = fused
// tail, bit 0 = 64-channel mid tile (TM = 1), bits 6-7 = output patch shape (0: 8x16, 1: 16x8, 2: 32x4).
extern "C" int rfx_conv3x3_conv1x1_kernel_id(int N, int H, int W, int Cmid) {
    const int pc = rfx_conv3x3_patch_cols(N, H, W, true);
    return 512 | (Cmid == 64 ? 1 : 0) | (pc == 16 ? 0 : (pc == 8 ? 64 : 128)) | (rfx_conv3x3_tail_chunked() ? 16384 : 0);   // bit 14: chunked
}

extern "C" int rfx_conv3x3_conv1x1_f32(const float* in, const float* wP2, const float* scale2, const float* shift2, int act2,
                                       const float* wQ3, const float* scale3, const float* shift3, const float* residual,
                                       int act3, float* out, int N, int Cin, int H, int W, int Cmid, int Cexp, void* stream) {
    if (!in || !wP2 || !wQ3 || !scale3 || !shift3 || !out || N <= 0 || H <= 0 || W <= 0) return RFX_E_ARG;
    if (Cin <= 0 || Cin % 8 != 0 || (Cmid != 64 && Cmid != 128) || Cexp <= 0 || Cexp % 128 != 0) return RFX_E_ARG;
    if (reinterpret_cast<uintptr_t>(wQ3) & 15) return RFX_E_ARG;
    if (act2 == RFX_ACT_SIGMOID || act3 == RFX_ACT_SIGMOID) return RFX_E_ARG;
    C3Args a;
    a.in = in; a.wT = wP2; a.scale = scale2; a.shift = shift2; a.res = residual; a.out = out;
    a.N = N; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cmid; a.act = act2; a.Mpad = 128;
    a.wT3 = wQ3; a.scale3 = scale3; a.shift3 = shift3; a.Cexp = Cexp; a.Mpad3 = Cexp; a.act3 = act3;
#ifdef RFX_TRACE
    a.trace = rfx_debug_trace_ptr();
#endif
    a.tilesM = 1;
    hipStream_t st = rfx_stream(stream);
    const int pc = rfx_conv3x3_patch_cols(N, H, W, true);
    if (rfx_conv3x3_tail_chunked()) {        // round 5: chunks of 4 K steps (288 k) in the 3x3 phase, like rfx_conv3x3_f32(k_chunk = 4)
        if (Cmid == 128) {
            if (pc == 16) return launch_direct<2, 16, true, 2, false, 4>(a, st);
            if (pc == 8) return launch_direct<2, 8, true, 2, false, 4>(a, st);
            return launch_direct<2, 4, true, 2, false, 4>(a, st);
        }
        if (pc == 16) return launch_direct<1, 16, true, 2, false, 4>(a, st);
        if (pc == 8) return launch_direct<1, 8, true, 2, false, 4>(a, st);
        return launch_direct<1, 4, true, 2, false, 4>(a, st);
    }
    if (Cmid == 128) {
        if (pc == 16) return launch_direct<2, 16, true>(a, st);
        if (pc == 8) return launch_direct<2, 8, true>(a, st);
        return launch_direct<2, 4, true>(a, st);
    }
    if (pc == 16) return launch_direct<1, 16, true>(a, st);   // the 256-pixel patch buys the fused tail nothing (100.9 vs 100.5 TFLOP/s): its loss is the expansion phase
    if (pc == 8) return launch_direct<1, 8, true>(a, st);
    return launch_direct<1, 4, true>(a, st);
}
