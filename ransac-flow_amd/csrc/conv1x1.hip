// conv1x1.hip -- 1x1 / stride 1 convolution (+ folded BatchNorm, residual, ReLU) on the fp32 MFMA: the Bottleneck conv1 /
// conv3 / downsample-free projections of the ResNet-50 trunk (model/resnet50.py:71-79,93-103), a third of the conv time
// of the path.
//
// A 1x1 convolution in NCHW is the plain GEMM  out[m][p] = sum_k W[m][k] * in[k][p]  with BOTH operands already k-major
// in memory: the packed weights wT[k][Mpad] and the input planes in[n][k][hw].  A lane of v_mfma_f32_32x32x2_f32 supplies
// ONE float per operand -- A[m = lane&31][k = lane>>5], B[k = lane>>5][p = lane&31] -- so with k-major LDS images
// As[k][m], Bs[k][p] the operand of a lane is a single ds_read_b32 at an immediate offset, 32 consecutive lanes read 32
// consecutive floats (no bank conflict by construction), and staging is a straight 16-byte copy global -> register ->
// LDS: no transpose, no swizzle, no validity mask (a pixel column of B only feeds the same column of the output, so the
// columns past the last pixel may hold anything; their loads are clamped to a valid address).
// conv.hip's generic kernel transposes both operands into [row][k-pair] images to read them back with ds_read_b128; the
// 4x4 register transposes and the bank-conflicting transposed stores cost it ~20 % of the matrix pipe on these layers.
//
// Pipeline (256 threads = 2x2 wavefronts, tile 64*TM channels x 128 pixels, BK = 32, LDS double buffered, ONE barrier
// per step): while the 16 k-pairs of tile s run on the matrix pipe, tile s+1 goes registers -> LDS (other buffer) in the
// shadow of the first MFMA chunks and each register is re-loaded with tile s+2 right behind its store -- never as a burst
// in front of the MFMAs (that cost the direct 3x3 kernel 13 %, scripts/ubench/conv_bench.py).
// k runs in the same pairs and the same order as in conv.hip: bit-identical results.
#include "common.h"
#include "conv_epilogue.h"
#include "group.h"

// experiments only (scripts/ubench/conv_bench.py, make c1dbgN): RFX_C1_DBG removes pieces of the main loop to price them
//   1 no LDS stores in the loop   2 no global loads in the loop   3 neither   4 neither, no barrier   6 no output stores
#ifndef RFX_C1_DBG
#define RFX_C1_DBG 0
#endif

namespace {

struct C1Args {
    const float* in; const float* wT; const float* scale; const float* shift; const float* res; float* out;
    int Cin, HW, Cout, act, Mpad;
    long long P;   // N*HW
    int tilesM, tilesP;
    unsigned stagger;   // common.h: rfx_stagger
};

// Persistent form: gridDim.x (= 2 workgroups per CU, a multiple of 8) workgroups walk the output tiles v = blockIdx.x,
// blockIdx.x + gridDim.x, ... and the staging pipeline runs ACROSS tiles: during the last two K steps of a tile the registers
// already receive the first two K steps of the workgroup's next tile, so the global-memory round trip that used to open every
// workgroup (load step 0, wait, store, load step 1, barrier: ~2 us of a 25 us lifetime when K = 256) overlaps the MFMAs and
// the epilogue of the tile before.  A tile's folded-BN vectors live in a double-buffered LDS pair (a fast wavefront may be
// one tile ahead of a slow one's epilogue).
// KCH > 0 (round 4): chunked accumulation as in conv3x3.hip -- every KCH K steps (KCH * 32 k) the accumulators are added to a
// second set and restarted: the K-blocked sum of the CPU reference's GEMM instead of one fma chain over K = 1024 products (ResNet-50
// layer3 conv1, model/resnet50.py:71).  64-channel tiles only (TM = 1: 174 + 32 registers; the 128-channel instance has none to spare).
template <int TM, bool VEC, int KCH = 0>
__device__ __forceinline__ void conv1x1_kmajor_body(const C1Args& a, const unsigned bx, const unsigned gsz) {
    static_assert(KCH == 0 || TM == 1, "chunked accumulation: 64-channel tiles");
    constexpr int BM = 64 * TM, BN = 128, BK = 32;
    constexpr int A_C4 = BM / 4;                 // float4 per A row: 32 / 16
    constexpr int A_RS = 256 / A_C4;             // rows covered by one round of the 256 threads: 8 / 16
    constexpr int NA = BK / A_RS;                // float4 per thread and step: 4 / 2
    constexpr int NBV = 4;                       // VEC: float4 per thread and step (32 float4 per row, 8 rows a round)
    constexpr int NBS = 16;                      // scalar: floats per thread and step (128 pixels per row, 2 rows a round)
    constexpr int NB = VEC ? NBV : NBS;
    __shared__ __attribute__((aligned(16))) float As[2][BK][BM];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN];
    __shared__ float s_scale[2][BM], s_shift[2][BM];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = lane >> 5, lcol = lane & 31;
    const size_t HW = (size_t)a.HW;
    const int nwg = a.tilesM * a.tilesP;
    const int nk = a.Cin / BK;                   // >= 2 (launch precondition)

    // tile v -> (m0, n0): XCD-aware bijective remap, m-tile fastest (the workgroups that share one pixel tile sit on one L2;
    // gridDim.x is a multiple of 8, so v % 8 is the XCD of this workgroup for every tile it walks)
    auto tile_origin = [&](int v, int& m0, long long& n0) {
        const int q = nwg / 8, r = nwg % 8, xcd = v % 8, j = v / 8;
        const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
        m0 = (bid % a.tilesM) * BM;
        n0 = (long long)(bid / a.tilesM) * BN;
    };
    // staging roles (fixed) and per-tile source pointers
    const int ar = t / A_C4, ac = t % A_C4;                                  // A: rows ar + A_RS*j, float4 column ac
    const int brow = VEC ? t >> 5 : t >> 7, bcol = VEC ? (t & 31) * 4 : t & 127;   // B: rows brow + 8*j (VEC) / brow + 2*i
    auto tile_sources = [&](int m0, long long n0, const float*& wsrc, const float*& bsrc) {
        wsrc = a.wT + (size_t)ar * a.Mpad + m0 + ac * 4;                     // + (k0 + A_RS*j) * Mpad
        long long p = n0 + bcol;
        if (p >= a.P) p = a.P - (VEC ? 4 : 1);                               // columns past the end: any valid address
        const long long n = p / a.HW;
        bsrc = a.in + (size_t)n * a.Cin * HW + (size_t)(p - n * a.HW) + (size_t)brow * HW;   // + (k0 + row) * HW
    };
    f32x4 ra[NA];
    f32x4 rv[VEC ? NBV : 1];
    float rb[VEC ? 1 : NBS];
    auto load_a = [&](const float* wsrc, int k0, int j) { ra[j] = *reinterpret_cast<const f32x4*>(wsrc + (size_t)(k0 + A_RS * j) * a.Mpad); };
    auto load_b = [&](const float* bsrc, int k0, int j) {
        if (VEC) rv[j] = *reinterpret_cast<const f32x4*>(bsrc + (size_t)(k0 + 8 * j) * HW);
        else     rb[j] = bsrc[(size_t)(k0 + 2 * j) * HW];
    };
    auto store_a = [&](int buf, int j) { *reinterpret_cast<f32x4*>(&As[buf][ar + A_RS * j][ac * 4]) = ra[j]; };
    auto store_b = [&](int buf, int j) {
        if (VEC) *reinterpret_cast<f32x4*>(&Bs[buf][brow + 8 * j][bcol]) = rv[j];
        else     Bs[buf][brow + 2 * j][bcol] = rb[j];
    };
    auto put_scale = [&](int slot, int m0) {
        if (t < BM) {
            const int m = m0 + t;
            s_scale[slot][t] = (a.scale && m < a.Cout) ? a.scale[m] : 1.0f;
            s_shift[slot][t] = (a.shift && m < a.Cout) ? a.shift[m] : 0.0f;
        }
    };

    rfx_stagger(a.stagger, bx, gsz);
    int v = (int)bx;
    int m0; long long n0;
    tile_origin(v, m0, n0);
    const float *wsrc, *bsrc;
    tile_sources(m0, n0, wsrc, bsrc);
    put_scale(0, m0);
    // prologue of the first tile only: K step 0 -> LDS buffer 0, K step 1 -> registers
#pragma unroll
    for (int j = 0; j < NA; ++j) load_a(wsrc, 0, j);
#pragma unroll
    for (int j = 0; j < NB; ++j) load_b(bsrc, 0, j);
#pragma unroll
    for (int j = 0; j < NA; ++j) store_a(0, j);
#pragma unroll
    for (int j = 0; j < NB; ++j) store_b(0, j);
#pragma unroll
    for (int j = 0; j < NA; ++j) load_a(wsrc, BK, j);
#pragma unroll
    for (int j = 0; j < NB; ++j) load_b(bsrc, BK, j);
    __syncthreads();

    // per-lane operand addresses inside a buffer: row 2*kk + lrow, column (32-wide sub-tile) + lcol
    const float* arow = &As[0][lrow][wm * TM * 32 + lcol];
    const float* brow_p = &Bs[0][lrow][wn * 64 + lcol];
    int g = 0;                                    // global step counter of this workgroup: LDS buffer = g & 1
    for (int it = 0; v < nwg; ++it, v += (int)gsz) {
        // the workgroup's next tile (none: this tile again -- its loads are harmless and nobody consumes them)
        const int vn = v + (int)gsz < nwg ? v + (int)gsz : v;
        int m0n; long long n0n;
        tile_origin(vn, m0n, n0n);
        const float *wsrc_n, *bsrc_n;
        tile_sources(m0n, n0n, wsrc_n, bsrc_n);

        f32x16 acc[TM][2];
        f32x16 tot[TM][2];                                    // only live in the KCH > 0 instances
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc[i][j][r] = 0.0f;
                    if constexpr (KCH > 0) tot[i][j][r] = 0.0f;
                }

        for (int s = 0; s < nk; ++s, ++g) {
            const int cur = g & 1;
            const float* ap = arow + cur * (BK * BM);
            const float* bp = brow_p + cur * (BK * BN);
            // the registers hold K step s+1 (stored below into the other buffer) and are re-loaded with step s+2 -- of this
            // tile, or steps 0 / 1 of the next one
            const bool nxt = s + 2 >= nk;
            const float* wl = nxt ? wsrc_n : wsrc;
            const float* bl = nxt ? bsrc_n : bsrc;
            const int k2 = (nxt ? s + 2 - nk : s + 2) * BK;
            // LDS reads run one 4-k-pair chunk ahead of the MFMAs that consume them (register double buffer)
            float af[2][4][TM], bf[2][4][2];
            auto read_chunk = [&](int c, int slot) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int kk = c * 4 + e;
#pragma unroll
                    for (int i = 0; i < TM; ++i) af[slot][e][i] = ap[2 * kk * BM + i * 32];
#pragma unroll
                    for (int j = 0; j < 2; ++j) bf[slot][e][j] = bp[2 * kk * BN + j * 32];
                }
            };
            read_chunk(0, 0);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (c + 1 < 4) read_chunk(c + 1, (c + 1) & 1);
                constexpr bool ST = RFX_C1_DBG != 1 && RFX_C1_DBG != 3 && RFX_C1_DBG != 4;
                constexpr bool LD = RFX_C1_DBG != 2 && RFX_C1_DBG != 3 && RFX_C1_DBG != 4;
                // a register is re-loaded right after it was stored, in chunks 0 / 1: the loads get the rest of the step to
                // land (with the loads in chunks 2 / 3 the next step's first store waits for them: -5 % in
                // scripts/ubench/mfma_mix.hip, variants g and g/1)
                if (c == 0) {
#pragma unroll
                    for (int j = 0; j < NA; ++j) {
                        if (ST) store_a(cur ^ 1, j); else if (LD) asm volatile("" ::"v"(ra[j]));
                    }
#pragma unroll
                    for (int j = 0; j < NA; ++j) if (LD) load_a(wl, k2, j);
                } else if (c == 1) {
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        if (ST) store_b(cur ^ 1, j);
                        else if (LD) { if (VEC) asm volatile("" ::"v"(rv[VEC ? j : 0])); else asm volatile("" ::"v"(rb[VEC ? 0 : j])); }
                    }
#pragma unroll
                    for (int j = 0; j < NB; ++j) if (LD) load_b(bl, k2, j);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c & 1][e][i], bf[c & 1][e][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (KCH > 0) {
                if ((s + 1) % KCH == 0 || s + 1 == nk) {      // wave-uniform: close the chunk
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int r = 0; r < 16; ++r) { tot[i][j][r] += acc[i][j][r]; acc[i][j][r] = 0.0f; }
                }
            }
            if (RFX_C1_DBG != 4) __syncthreads();   // step s+1 is complete in the other buffer; everyone is done reading this one
        }
        put_scale((it + 1) & 1, m0n);             // the next tile's BN vectors (read after >= nk barriers)
        if (RFX_C1_DBG == 6) {
            float sacc = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sacc += acc[i][j][r];
            if (sacc == 123.456f) a.out[t] = sacc;
        } else {
            // ---- epilogue (conv_epilogue.h); the loads of the next tile's K step 1 are in flight meanwhile ----
            size_t pix_off[2];
            bool pix_ok[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                long long pp = n0 + wn * 64 + j * 32 + lcol;
                pix_ok[j] = pp < a.P;
                if (!pix_ok[j]) pp = a.P - 1;
                const long long n = pp / a.HW;
                pix_off[j] = (size_t)n * a.Cout * HW + (size_t)(pp - n * a.HW);
            }
            conv_epilogue<TM, 2, false>(KCH > 0 ? tot : acc, s_scale[it & 1], s_shift[it & 1], a.res, a.out, a.act, a.Cout, HW, m0, wm, lrow,
                                        pix_off, pix_ok, m0 + BM <= a.Cout);
        }
        m0 = m0n; n0 = n0n; wsrc = wsrc_n; bsrc = bsrc_n;
    }
}

template <int TM, bool VEC, int KCH = 0>
__global__ __launch_bounds__(256, 2) void conv1x1_kmajor_kernel(C1Args a) {
    conv1x1_kmajor_body<TM, VEC, KCH>(a, blockIdx.x, gridDim.x);
}

// grouped form (group.h): blockIdx.y = problem; a problem's persistent workgroups stride by that problem's own grid
template <int TM, bool VEC, int KCH = 0>
__global__ __launch_bounds__(256, 2) void conv1x1_kmajor_group_kernel(RfxGroupArgs<C1Args> g) {
    const unsigned y = blockIdx.y;
    if (blockIdx.x >= g.gx[y]) return;
    conv1x1_kmajor_body<TM, VEC, KCH>(g.p[y], blockIdx.x, g.gx[y]);
}

template <int TM, bool VEC, int KCH = 0>
static int c1_group_launch(const void* blob, const unsigned* gx, int n, hipStream_t st) {
    return rfx_group_launch_impl<C1Args>(conv1x1_kmajor_group_kernel<TM, VEC, KCH>, 256, blob, gx, n, st);
}

template <int TM, bool VEC, int KCH = 0>
int launch_1x1(C1Args& a, hipStream_t st) {
    a.tilesM = (a.Cout + 64 * TM - 1) / (64 * TM);
    a.tilesP = (int)((a.P + 127) / 128);
    const long long nwg = (long long)a.tilesM * a.tilesP;
    if (nwg > 0x7fffffffLL) return RFX_E_LIMIT;
    // persistent grid: 2 workgroups per CU (what the kernel's registers and LDS allow), a multiple of 8 (XCDs)
    static int slots = 0;
    if (!slots) {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        slots = (2 * cus + 7) / 8 * 8;
    }
    const unsigned grid = (unsigned)(nwg < slots ? (nwg + 7) / 8 * 8 : slots);
    static const unsigned stagger = rfx_stagger_env("RFX_C1_STAGGER", "RFX_C1_STAGGER_MODE");
    a.stagger = (rfx_group_recording() || (int)grid < slots) ? 0u : stagger;      // only when every CU holds its two workgroups
    if (rfx_group_recording()) return rfx_group_record(&c1_group_launch<TM, VEC, KCH>, &a, sizeof(a), grid);
    hipLaunchKernelGGL((conv1x1_kmajor_kernel<TM, VEC, KCH>), dim3(grid), dim3(256), 0, st, a);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

}  // namespace

// Internal entry used by rfx_conv2d_f32 (conv.hip).  Preconditions checked by the caller: 1x1, stride 1, pad 0,
// Cin % 32 == 0, Cin >= 64, N*HW >= 4; tm = 2 -> 128 output channels per workgroup, tm = 1 -> 64; vec: HW % 4 == 0 and `in` 16-byte
// aligned (16-byte pixel loads).
int rfx_conv1x1_kmajor_launch(const float* in, const float* wT, const float* scale, const float* shift, const float* residual,
                              float* out, int N, int Cin, int HW, int Cout, int Mpad, int act, int tm, bool vec,
                              hipStream_t st, bool chunked) {
    C1Args a;
    a.in = in; a.wT = wT; a.scale = scale; a.shift = shift; a.res = residual; a.out = out;
    a.Cin = Cin; a.HW = HW; a.Cout = Cout; a.act = act; a.Mpad = Mpad;
    a.P = (long long)N * HW;
    if (chunked) return vec ? launch_1x1<1, true, 8>(a, st) : launch_1x1<1, false, 8>(a, st);     // K >= 1024: chunks of 8 steps (256 k)
    if (tm == 2) return vec ? launch_1x1<2, true>(a, st) : launch_1x1<2, false>(a, st);
    return vec ? launch_1x1<1, true>(a, st) : launch_1x1<1, false>(a, st);
}
