// conv3x3s.hip -- 3x3 / pad 1 convolutions, stride 1 and (second half of the file) stride 2 (+ folded BatchNorm, residual, ReLU) with
// float32 results on the bf16 matrix cores by EXACT operand splitting -- the scheme of conv1x1s.hip (read its header first) applied to
// the heaviest layer class of the path: ResNet-50 conv2 of every Bottleneck (model/resnet50.py:75), every BasicBlock convolution of the
// FeatureExtractor (model/model.py:32-35), conv1 / conv2 / conv3 of the NetFlowCoarse / NetMatchability stacks (model/model.py:170-181;
// the 49-channel conv1 through a zero-padded 64-channel copy of its input, rfx/ops.py).
//
//     out[m][p] = sum over taps (kh, kw) and channels c of  W[m][c][kh][kw] * in[c][p + (kh-1, kw-1)]
// = nine shifted 1x1 GEMMs over ONE staged input patch.  A workgroup owns 64*TM output channels x an 8 x 16 pixel patch; per block of
// 16 input channels it stages the 10 x 18 halo patch ONCE -- float32 from HBM, split into the three bf16 pieces, LDS image
//     Bs[piece][h = channel half][patch pixel (180)][8 channels]        (16-byte words; zero outside the image = the padding)
// so that the B fragment of tap (kh, kw) for output pixel (r, x) is the word at patch pixel (r + kh) * 18 + x + kw: one ds_read_b128
// at a per-lane base + a compile-time tap offset.  The weights of a (channel block, tap) stage come split and packed from the host
// (rfx_api.h: "wS3"), 12 KB per stage, global -> LDS by global_load_lds into a double-buffered image (no staging registers; an explicit
// vmcnt in front of the stage's ONE barrier publishes it); a stage = 6 MFMAs per 32 x 32 tile (hi*hi in its
// own accumulator, the five small terms in a second one: conv1x1s.hip).  k order: channel block, tap, 16 channels -- irrelevant for
// the result's quality (every product exact, 16 products per rounding), different from the fp32 kernels' channel-major order.
// The images of a batch are tiled as ONE tall map with a virtual zero row between images (as conv3x3.hip): only the last patch row
// of the whole batch is ragged; outputs on a virtual row are dropped.
#include "common.h"
#include "conv_epilogue.h"
#include "group.h"
#include <stdlib.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Patch geometry.  WM = wavefronts along the channels: 2 -> 128 channels x (8 x 16) pixels per workgroup; 1 (Cout <= 64) -> 64 channels x
// (16 x 16) pixels, the four wavefronts along the rows: every wavefront keeps 2 x 2 MFMA tiles (24 MFMAs per stage) and the weight image of
// a stage is shared by twice the pixels (64 -> 64 at 240 x 320: 172 -> 185 TFLOP/s float32-equivalent).  LDS patch rows are PC pixels apart
// (a stride of 20 / 22 / 24 changes nothing: measured).
constexpr int PT_C = 16, PC = PT_C + 2;
template <int WM> struct Geo {
    static constexpr int BM = 64 * WM, WN = 4 / WM, PT_R = 4 * WN, PR = PT_R + 2, PP = PR * PC;     // PP: 180 / 324 patch pixels
    static constexpr int B_ITEMS = 2 * PP, NBI = (B_ITEMS + 255) / 256;                               // staging items per thread: 2 / 3
};

struct C3SArgs {
    const float* in; const u32x4* wS; const float* scale; const float* shift; const float* res; float* out;
    int N, Cin, H, W, Cout, act, Mpad;
    int tilesM, tilesW, tilesS;     // channel tiles, column tiles, row tiles of the N * (H + 1) - 1 row stack
};

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    const f32x2 v = {a, b};
    const bf16x2 h = __builtin_convertvector(v, bf16x2);
    unsigned u;
    __builtin_memcpy(&u, &h, 4);
    return u;
}
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
    hi = pack_bf16(a, b);
    const float ra = a - __uint_as_float(hi << 16), rb = b - __uint_as_float(hi & 0xffff0000u);            // exact
    mid = pack_bf16(ra, rb);
    lo = pack_bf16(ra - __uint_as_float(mid << 16), rb - __uint_as_float(mid & 0xffff0000u));
}
__device__ __forceinline__ bf16x8 as_frag(const u32x4& w) {
    bf16x8 f;
    __builtin_memcpy(&f, &w, 16);
    return f;
}

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// ADMA: the weight image of a stage goes global -> LDS by global_load_lds (no registers, no ds_write, and no wait for it in front of the
// stage's MFMAs: the wait (vmcnt(0)) sits in front of the barrier that publishes the image).
// Fragment reads of the DMA instances are hand-written ds_read_b128 (as in corr.hip): in front of an LDS read it can see, the compiler
// drains the vector-memory counter whenever an LDS-DMA is in flight (it cannot tell the weight image being written from the one being read) --
// priced by removal, the weight DMA cost 18 % of the kernel that way (224.5 -> 265.5 TFLOP/s float32-equivalent without it).
template <int OFF>
__device__ __forceinline__ void lds_rd(u32x4& d, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
__device__ __forceinline__ void lds_wait8(u32x4& a, u32x4& b, u32x4& c, u32x4& d, u32x4& e, u32x4& f, u32x4& g, u32x4& h) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
}
__device__ __forceinline__ void lds_wait4(u32x4& a, u32x4& b, u32x4& c, u32x4& d) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}

template <int WM, bool ADMA>
__device__ __forceinline__ void conv3x3_split_body(const C3SArgs& a, const unsigned bx) {
    using G = Geo<WM>;
    constexpr int TM = 2;                               // MFMA tiles per wavefront along the channels (and 2 along the pixels)
    constexpr int BM = G::BM, PT_R = G::PT_R, PP = G::PP, B_ITEMS = G::B_ITEMS, NBI = G::NBI;
    constexpr int A_WORDS = 3 * 2 * BM;                 // 16-byte words of a stage's weight image
    constexpr int NA = (A_WORDS + 255) / 256;
    __shared__ u32x4 As[2][3][2][BM];
    __shared__ u32x4 Bs[2][3][2][PP];
    __shared__ float s_scale[BM], s_shift[BM];

    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = WM == 2 ? wave >> 1 : 0, wn = WM == 2 ? wave & 1 : wave;
    const int lrow = lane >> 5, lcol = lane & 31;
    const size_t HW = (size_t)a.H * a.W;
    const int nwg = a.tilesM * a.tilesW * a.tilesS;
    const int nk = a.Cin / 16;
    const int Hs = a.H + 1;                             // rows per image in the stack (the last one virtual)

    int m0, row0, col0;
    {
        const int v = (int)bx, q = nwg / 8, r = nwg % 8, xcd = v % 8, j = v / 8;
        int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
        m0 = (bid % a.tilesM) * BM; bid /= a.tilesM;
        col0 = (bid % a.tilesW) * PT_C;
        row0 = (bid / a.tilesW) * PT_R;
    }
    // A staging: words t + 256 j of the stage image [piece][h][BM] <- wS[(kb * 9 + tap)][piece][h][m0 + m]; 256 words = 2 (TM = 2) or
    // 4 (TM = 1) [piece][h] rows: word t + 256 j sits (256 / BM) * j rows below word t
    constexpr int A_ROWS = 256 / BM;
    const u32x4* wsrc = a.wS + (size_t)(t / BM) * a.Mpad + m0 + t % BM;              // + q * 6 * Mpad + A_ROWS * j * Mpad
    // piece j of a thread exists for every thread (compile time) or for the first wavefronts only (TM = 1: 384 words): no per-lane
    // branches around the loads -- a divergent region makes the compiler drain the vector-memory counter between two loads
    auto a_on = [&](int j) { return (j + 1) * 256 <= A_WORDS || t + 256 * j < A_WORDS; };
    // B staging: item it = t (+ 256): h = it / 180, patch pixel it % 180 -> 8 channels of one input pixel (or zeros)
    const float* bsrc[NBI];
    bool b_ok[NBI], b_on[NBI];
    int b_word[NBI];
#pragma unroll
    for (int u = 0; u < NBI; ++u) {
        const int it = t + 256 * u;
        b_on[u] = it < B_ITEMS;
        const int h = (b_on[u] ? it : 0) / PP, pp = (b_on[u] ? it : 0) % PP;
        const int pr = pp / PC, pc = pp % PC;            // pc >= 18 (padded strides): never read, staged as zeros
        const int sr = row0 - 1 + pr, x = col0 - 1 + pc;
        const int n = sr >= 0 ? sr / Hs : 0, y = sr >= 0 ? sr - n * Hs : 0;
        b_ok[u] = b_on[u] && sr >= 0 && n < a.N && y < a.H && x >= 0 && x < a.W;
        bsrc[u] = b_ok[u] ? a.in + ((size_t)n * a.Cin + 8 * h) * HW + (size_t)y * a.W + x : a.in;   // + kb * 16 * HW + i * HW
        b_word[u] = h * PP + pp;
    }
    u32x4 ra[NA];
    float rb[8];                                        // one staging item at a time: item 0 and item 1 of a block share the registers
    auto load_a = [&](int q) {
#pragma unroll
        for (int j = 0; j < NA; ++j)
            ra[j] = wsrc[((size_t)q * 6 + (a_on(j) ? A_ROWS * j : 0)) * a.Mpad];              // off lanes: any valid word
    };
    auto store_a = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NA; ++j)
            if (a_on(j)) (&As[buf][0][0][0])[t + 256 * j] = ra[j];
    };
    auto dma_a = [&](int q, int buf) {                  // piece j of this wavefront: words 256 j + 64 wave .. + 63 of the stage image
#pragma unroll
        for (int j = 0; j < NA; ++j)
            if ((j + 1) * 256 <= A_WORDS || wave < (A_WORDS - 256 * j) / 64)
                __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + ((size_t)q * 6 + A_ROWS * j) * a.Mpad),
                                                 (lptr_t)(&As[buf][0][0][0] + 256 * j + 64 * wave), 16, 0, 0);
    };
    auto load_b = [&](int kb, int u) {
#pragma unroll
        for (int i = 0; i < 8; ++i) rb[i] = bsrc[u][((size_t)kb * 16 + i) * HW];      // unconditional (pixels outside: image 0's, zeroed below)
    };
    auto store_b = [&](int buf, int u) {
        if (!b_on[u]) return;
        u32x4 hi, mid, lo;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned h_, m_, l_;
            split_pair(b_ok[u] ? rb[2 * i] : 0.0f, b_ok[u] ? rb[2 * i + 1] : 0.0f, h_, m_, l_);
            hi[i] = h_; mid[i] = m_; lo[i] = l_;
        }
        u32x4* dst = &Bs[buf][0][0][0] + b_word[u];
        dst[0] = hi;
        dst[2 * PP] = mid;
        dst[4 * PP] = lo;
    };
    if (t < BM) {
        const int m = m0 + t;
        s_scale[t] = (a.scale && m < a.Cout) ? a.scale[m] : 1.0f;
        s_shift[t] = (a.shift && m < a.Cout) ? a.shift[m] : 0.0f;
    }
    if (ADMA) dma_a(0, 0);
    else { load_a(0); store_a(0); }
#pragma unroll
    for (int u = 0; u < NBI; ++u) { load_b(0, u); store_b(0, u); }
    if (ADMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else load_a(1);
    if (nk > 1) load_b(1, 0);
    __syncthreads();

    f32x16 acc[TM][2], low[TM][2];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.0f; low[i][j][r] = 0.0f; }

    // per-lane fragment bases (16-byte words): A row, B patch pixel of output (4 wn + 2 j + lcol / 16, lcol % 16) at tap (0, 0)
    const int a_base = lrow * BM + wm * TM * 32 + lcol;
    int b_base[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) b_base[j] = lrow * PP + (4 * wn + 2 * j + (lcol >> 4)) * PC + (lcol & 15);
    const int nq = 9 * nk;
    const unsigned as_lds = (unsigned)(uintptr_t)(lptr_t)&As[0][0][0][0], bs_lds = (unsigned)(uintptr_t)(lptr_t)&Bs[0][0][0][0];

    for (int kb = 0; kb < nk; ++kb) {
        const u32x4* bimg = &Bs[kb & 1][0][0][0];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int q = kb * 9 + tap;
            const u32x4* aimg = &As[q & 1][0][0][0];
            const int toff = (tap / 3) * PC + tap % 3;
            // the next block's patch: item 0 was loaded at tap 8 of the block before (prologue: block 1), goes to LDS at tap 3; item 1
            // is loaded behind it and stored at tap 7; the buffer's last readers finished with block kb - 1
            bf16x8 a0[TM], a1[TM], b0[2], b1[2];
            u32x4 ra0[2], ra1[2], rb0[2], rb1[2];           // ADMA: the raw fragment words (TM = 2)
            unsigned a_addr = 0, b_addr[2] = {0, 0};
            if constexpr (ADMA) {
                static_assert(TM == 2, "hand-written fragment reads: two channel tiles per wavefront");
                a_addr = as_lds + (unsigned)(((q & 1) * A_WORDS + a_base) * 16);
                b_addr[0] = bs_lds + (unsigned)(((kb & 1) * 6 * PP + b_base[0] + toff) * 16);
                b_addr[1] = bs_lds + (unsigned)(((kb & 1) * 6 * PP + b_base[1] + toff) * 16);
                lds_rd<0>(ra0[0], a_addr); lds_rd<512>(ra0[1], a_addr);
                lds_rd<2 * BM * 16>(ra1[0], a_addr); lds_rd<2 * BM * 16 + 512>(ra1[1], a_addr);
                lds_rd<0>(rb0[0], b_addr[0]); lds_rd<0>(rb0[1], b_addr[1]);
                lds_rd<2 * PP * 16>(rb1[0], b_addr[0]); lds_rd<2 * PP * 16>(rb1[1], b_addr[1]);
            } else {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    a0[i] = as_frag(aimg[a_base + i * 32]);
                    a1[i] = as_frag(aimg[2 * BM + a_base + i * 32]);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    b0[j] = as_frag(bimg[b_base[j] + toff]);
                    b1[j] = as_frag(bimg[2 * PP + b_base[j] + toff]);
                }
            }
#ifndef RFX_C3S_DBG          // experiments only (WRONG RESULTS): 1 = no weight DMA in the loop, 2 = no activation requests in the loop
#define RFX_C3S_DBG 0
#endif
            if (!ADMA) {
                if (q + 1 < nq) store_a((q + 1) & 1);       // stage q+1's weights: registers -> the other buffer
                if (q + 2 < nq) load_a(q + 2);
            }
            // the items of a thread share the registers: store item u here, request item u + 1 LATER in the stage (b_next / b_kb), behind the
            // lo-piece fragment reads: the compiler drains the vector-memory counter in front of an LDS read that follows an LDS-DMA (it cannot
            // tell the images apart: the ISA shows an s_waitcnt vmcnt(0) there), and a request issued before that point is drained with it
            // (+2 %: 223.5 -> 225.4 on 256 -> 256 at 60 x 80).  Moving the DMA itself behind the stage's last LDS read and making the counted
            // wait a builtin the compiler sees leaves ONE drain per stage, right in front of the barrier -- and measures 1.5 % slower (221.9).
            int b_next = -1, b_kb = 0;
            if (kb + 1 < nk) {
                if (NBI == 2) {
                    if (tap == 3) { store_b((kb + 1) & 1, 0); b_next = 1; b_kb = kb + 1; }
                    if (tap == 7) store_b((kb + 1) & 1, 1);
                } else {
                    if (tap == 2) { store_b((kb + 1) & 1, 0); b_next = 1; b_kb = kb + 1; }
                    if (tap == 5) { store_b((kb + 1) & 1, 1); b_next = NBI - 1; b_kb = kb + 1; }
                    if (tap == 8) store_b((kb + 1) & 1, NBI - 1);
                }
            }
            if (tap == 8 && kb + 2 < nk) { b_next = 0; b_kb = kb + 2; }
            const bool b_req = b_next >= 0;
            // the DMA BEHIND the stage's ds_writes (store_b): the compiler drains the counter in front of an LDS write that follows an LDS-DMA
            if (ADMA && q + 1 < nq && RFX_C3S_DBG != 1) dma_a(q + 1, (q + 1) & 1);      // its readers (stage q - 1) passed the last barrier
            if (!ADMA && b_req) { if (b_next == 0) load_b(b_kb, 0); else if (b_next == 1) load_b(b_kb, 1); else load_b(b_kb, NBI - 1); }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (ADMA) {
                lds_wait8(ra0[0], ra0[1], ra1[0], ra1[1], rb0[0], rb0[1], rb1[0], rb1[1]);
#pragma unroll
                for (int i = 0; i < 2; ++i) { a0[i] = as_frag(ra0[i]); a1[i] = as_frag(ra1[i]); b0[i] = as_frag(rb0[i]); b1[i] = as_frag(rb1[i]); }
            }
            // the twelve products that read the mid pieces first; then the lo pieces are fetched into the mid pieces' registers while the
            // four hi * hi products run; then the lo products
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    low[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[i], b1[j], low[i][j], 0, 0, 0);    // mid * mid
                    low[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[i], b0[j], low[i][j], 0, 0, 0);    // mid * hi
                }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) low[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[i], b1[j], low[i][j], 0, 0, 0);    // hi * mid
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (ADMA) {
                lds_rd<4 * BM * 16>(ra1[0], a_addr); lds_rd<4 * BM * 16 + 512>(ra1[1], a_addr);
                lds_rd<4 * PP * 16>(rb1[0], b_addr[0]); lds_rd<4 * PP * 16>(rb1[1], b_addr[1]);
            } else {
#pragma unroll
                for (int i = 0; i < TM; ++i) a1[i] = as_frag(aimg[4 * BM + a_base + i * 32]);
#pragma unroll
                for (int j = 0; j < 2; ++j) b1[j] = as_frag(bimg[4 * PP + b_base[j] + toff]);
            }
            if (ADMA && b_req && RFX_C3S_DBG != 2) {
                __builtin_amdgcn_sched_barrier(0);
                if (b_next == 0) load_b(b_kb, 0); else if (b_next == 1) load_b(b_kb, 1); else load_b(b_kb, NBI - 1);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[i], b0[j], acc[i][j], 0, 0, 0);    // hi * hi
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (ADMA) {
                lds_wait4(ra1[0], ra1[1], rb1[0], rb1[1]);
#pragma unroll
                for (int i = 0; i < 2; ++i) { a1[i] = as_frag(ra1[i]); b1[i] = as_frag(rb1[i]); }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    low[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[i], b0[j], low[i][j], 0, 0, 0);    // lo  * hi
                    low[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[i], b1[j], low[i][j], 0, 0, 0);    // hi  * lo
                }
            __builtin_amdgcn_sched_barrier(0);
            if (ADMA) {     // the next stage's weight image has landed; the bare barrier + explicit waits instead of __syncthreads()
                // vmcnt(0) in EVERY stage, also the ones that requested activations behind the DMA: a counted vmcnt(8) would rely on the DMA
                // completing before the eight younger register loads, and the compiler's own bookkeeping treats LDS-DMA and register loads as
                // able to complete out of order (it forces 0 wherever both are pending); no result may depend on that (costs about 1 %)
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            } else {
                __syncthreads();
            }
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] += low[i][j][r];

    size_t pix_off[2];
    bool pix_ok[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int sr = row0 + 4 * wn + 2 * j + (lcol >> 4), x = col0 + (lcol & 15);
        const int n = sr / Hs, y = sr - n * Hs;
        pix_ok[j] = n < a.N && y < a.H && x < a.W;
        pix_off[j] = pix_ok[j] ? (size_t)n * a.Cout * HW + (size_t)y * a.W + x : 0;
    }
    conv_epilogue<TM, 2, false>(acc, s_scale, s_shift, a.res, a.out, a.act, a.Cout, HW, m0, wm, lrow, pix_off, pix_ok, m0 + BM <= a.Cout);
}

template <int TM, bool ADMA>
__global__ __launch_bounds__(256, 2) void conv3x3_split_kernel(C3SArgs a) {
    conv3x3_split_body<TM, ADMA>(a, blockIdx.x);
}

template <int TM, bool ADMA>
__global__ __launch_bounds__(256, 2) void conv3x3_split_group_kernel(RfxGroupArgs<C3SArgs> g) {
    const unsigned y = blockIdx.y;
    if (blockIdx.x >= g.gx[y]) return;
    conv3x3_split_body<TM, ADMA>(g.p[y], blockIdx.x);
}

template <int TM, bool ADMA>
static int c3s_group_launch(const void* blob, const unsigned* gx, int n, hipStream_t st) {
    return rfx_group_launch_impl<C3SArgs>(conv3x3_split_group_kernel<TM, ADMA>, 256, blob, gx, n, st);
}

template <int TM, bool ADMA>
int launch_split3_a(C3SArgs& a, hipStream_t st) {
    a.tilesM = (a.Cout + 64 * TM - 1) / (64 * TM);
    a.tilesW = (a.W + PT_C - 1) / PT_C;
    const long long rows = (long long)a.N * (a.H + 1) - 1;
    a.tilesS = (int)((rows + Geo<TM>::PT_R - 1) / Geo<TM>::PT_R);
    const long long nwg = (long long)a.tilesM * a.tilesW * a.tilesS;
    if (nwg > 0x7fffffffLL) return RFX_E_LIMIT;
    if (rfx_group_recording()) return rfx_group_record(&c3s_group_launch<TM, ADMA>, &a, sizeof(a), (unsigned)nwg);
    hipLaunchKernelGGL((conv3x3_split_kernel<TM, ADMA>), dim3((unsigned)nwg), dim3(256), 0, st, a);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

template <int TM>
int launch_split3(C3SArgs& a, hipStream_t st) {
    static const bool adma = !(getenv("RFX_C3S_ADMA") && atoi(getenv("RFX_C3S_ADMA")) == 0);   // 0: weights through registers (A/B runs; +2-4 % with the DMA)
    return adma ? launch_split3_a<TM, true>(a, st) : launch_split3_a<TM, false>(a, st);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Stride 2 (ResNet-50 layer2.0 / layer3.0 conv2, model/resnet50.py:75 with stride; the FeatureExtractor's strided BasicBlock conv1,
// model/model.py:32): out(y, x) = sum W[kh][kw] in(2y + kh - 1, 2x + kw - 1).  128 channels x 8 x 16 outputs per workgroup; the 17 x 33 input
// patch of a 16-channel block is staged ONCE per block, its rows de-interleaved by column parity ([17 even | 16 odd]) so that the 16
// lanes of an output row read consecutive words for every tap: word = row * 33 + (kw & 1) * 17 + x + (kw >> 1).  The patch (54 KB in split
// form) does not double-buffer next to the weights: it is staged between two barriers at the start of a block (the frag registers are
// free then: four staging items in flight per thread) while the CU's other workgroup computes.  Everything else -- weights per
// (block, tap) stage through a double-buffered image, six products per tile and stage, two accumulators -- as the stride-1 kernel.
// Rows: the batch is one tall map in OUTPUT rows (Ho + 1 per image, the last one virtual); input row of output stack row sr, tap kh:
// 2 sr + kh - 1 in an input stack of 2 (Ho + 1) rows per image (rows >= H: zero = the padding / the virtual rows).
constexpr int S2_PR = 2 * 8 + 1, S2_PC = 33, S2_PP = S2_PR * S2_PC;            // 17 x 33 = 561 patch pixels
constexpr int S2_ITEMS = 2 * S2_PP, S2_NBI = (S2_ITEMS + 255) / 256;           // 1122 staging items: 5 per thread (the fifth: 98 threads)

struct C3S2Args {
    const float* in; const u32x4* wS; const float* scale; const float* shift; const float* res; float* out;
    int N, Cin, H, W, Ho, Wo, Cout, act, Mpad;
    int tilesM, tilesW, tilesS;
};

__device__ __forceinline__ void conv3x3_split_s2_body(const C3S2Args& a, const unsigned bx) {
    constexpr int TM = 2, BM = 128;
    constexpr int NA = 3;                               // 768 weight words per stage: 3 per thread
    __shared__ u32x4 As[2][3][2][BM];
    __shared__ u32x4 Bs[3][2][S2_PP];
    __shared__ float s_scale[BM], s_shift[BM];

    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = lane >> 5, lcol = lane & 31;
    const size_t HW = (size_t)a.H * a.W, HWo = (size_t)a.Ho * a.Wo;
    const int nwg = a.tilesM * a.tilesW * a.tilesS;
    const int nk = a.Cin / 16;
    const int Hs = a.Ho + 1, Hin = 2 * Hs;              // output / input rows per image in the stacks

    int m0, row0, col0;
    {
        const int v = (int)bx, q = nwg / 8, r = nwg % 8, xcd = v % 8, j = v / 8;
        int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
        m0 = (bid % a.tilesM) * BM; bid /= a.tilesM;
        col0 = (bid % a.tilesW) * 16;
        row0 = (bid / a.tilesW) * 8;
    }
    const u32x4* wsrc = a.wS + (size_t)(t / BM) * a.Mpad + m0 + t % BM;              // + (q * 6 + 2 j) * Mpad
    const float* bsrc[S2_NBI];
    bool b_ok[S2_NBI];
    int b_word[S2_NBI];
#pragma unroll
    for (int u = 0; u < S2_NBI; ++u) {
        const int it = t + 256 * u;
        const bool on = it < S2_ITEMS;
        const int h = (on ? it : 0) / S2_PP, pp = (on ? it : 0) % S2_PP;
        const int pr = pp / S2_PC, pc = pp % S2_PC;
        const int si = 2 * row0 - 1 + pr, x = 2 * col0 - 1 + pc;
        const int n = si >= 0 ? si / Hin : 0, y = si >= 0 ? si - n * Hin : 0;
        b_ok[u] = on && si >= 0 && n < a.N && y < a.H && x >= 0 && x < a.W;
        bsrc[u] = b_ok[u] ? a.in + ((size_t)n * a.Cin + 8 * h) * HW + (size_t)y * a.W + x : a.in;
        b_word[u] = on ? h * S2_PP + pr * S2_PC + (pc & 1) * 17 + (pc >> 1) : -1;
    }
    auto dma_a = [&](int q, int buf) {                  // the stage's weight image global -> LDS (global_load_lds): words 256 j + 64 wave .. + 63
#pragma unroll
        for (int j = 0; j < NA; ++j)
            __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + ((size_t)q * 6 + 2 * j) * a.Mpad), (lptr_t)(&As[buf][0][0][0] + 256 * j + 64 * wave),
                                             16, 0, 0);
    };
    auto stage_b = [&](int kb) {     // the whole patch of block kb: four items in flight, then the fifth
        float rb[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) rb[u][i] = bsrc[u][((size_t)kb * 16 + i) * HW];
        auto put = [&](const float (&v)[8], int u) {
            if (b_word[u] < 0) return;
            u32x4 hi, mid, lo;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                unsigned h_, m_, l_;
                split_pair(b_ok[u] ? v[2 * i] : 0.0f, b_ok[u] ? v[2 * i + 1] : 0.0f, h_, m_, l_);
                hi[i] = h_; mid[i] = m_; lo[i] = l_;
            }
            u32x4* dst = &Bs[0][0][0] + b_word[u];
            dst[0] = hi;
            dst[2 * S2_PP] = mid;
            dst[4 * S2_PP] = lo;
        };
#pragma unroll
        for (int u = 0; u < 4; ++u) put(rb[u], u);
        float r4[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) r4[i] = bsrc[4][((size_t)kb * 16 + i) * HW];
        put(r4, 4);
    };
    if (t < BM) {
        const int m = m0 + t;
        s_scale[t] = (a.scale && m < a.Cout) ? a.scale[m] : 1.0f;
        s_shift[t] = (a.shift && m < a.Cout) ? a.shift[m] : 0.0f;
    }
    dma_a(0, 0);                                        // lands before the first block's staging barrier (vmcnt(0) below)

    f32x16 acc[TM][2], low[TM][2];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.0f; low[i][j][r] = 0.0f; }

    const int a_base = lrow * BM + wm * TM * 32 + lcol;
    int b_base[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) b_base[j] = lrow * S2_PP + 2 * (4 * wn + 2 * j + (lcol >> 4)) * S2_PC + (lcol & 15);
    const int nq = 9 * nk;
    const u32x4* bimg = &Bs[0][0][0];

    for (int kb = 0; kb < nk; ++kb) {
        stage_b(kb);                                    // the last readers of the patch passed the barrier that closed block kb - 1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int q = kb * 9 + tap;
            const u32x4* aimg = &As[q & 1][0][0][0];
            const int toff = (tap / 3) * S2_PC + (tap % 3 & 1) * 17 + (tap % 3 >> 1);
            bf16x8 a0[TM], a1[TM], b0[2], b1[2];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                a0[i] = as_frag(aimg[a_base + i * 32]);
                a1[i] = as_frag(aimg[2 * BM + a_base + i * 32]);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                b0[j] = as_frag(bimg[b_base[j] + toff]);
                b1[j] = as_frag(bimg[2 * S2_PP + b_base[j] + toff]);
            }
            if (q + 1 < nq) dma_a(q + 1, (q + 1) & 1);      // its readers (stage q - 1) passed the last barrier
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    low[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[i], b1[j], low[i][j], 0, 0, 0);    // mid * mid
                    low[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[i], b0[j], low[i][j], 0, 0, 0);    // mid * hi
                }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) low[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[i], b1[j], low[i][j], 0, 0, 0);    // hi * mid
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; ++i) a1[i] = as_frag(aimg[4 * BM + a_base + i * 32]);
#pragma unroll
            for (int j = 0; j < 2; ++j) b1[j] = as_frag(bimg[4 * S2_PP + b_base[j] + toff]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[i], b0[j], acc[i][j], 0, 0, 0);    // hi * hi
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    low[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[i], b0[j], low[i][j], 0, 0, 0);    // lo  * hi
                    low[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[i], b1[j], low[i][j], 0, 0, 0);    // hi  * lo
                }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the next stage's weight image has landed
            __syncthreads();
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] += low[i][j][r];

    size_t pix_off[2];
    bool pix_ok[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int sr = row0 + 4 * wn + 2 * j + (lcol >> 4), x = col0 + (lcol & 15);
        const int n = sr / Hs, y = sr - n * Hs;
        pix_ok[j] = n < a.N && y < a.Ho && x < a.Wo;
        pix_off[j] = pix_ok[j] ? (size_t)n * a.Cout * HWo + (size_t)y * a.Wo + x : 0;
    }
    conv_epilogue<TM, 2, false>(acc, s_scale, s_shift, a.res, a.out, a.act, a.Cout, HWo, m0, wm, lrow, pix_off, pix_ok, m0 + BM <= a.Cout);
}

__global__ __launch_bounds__(256, 2) void conv3x3_split_s2_kernel(C3S2Args a) {
    conv3x3_split_s2_body(a, blockIdx.x);
}

__global__ __launch_bounds__(256, 2) void conv3x3_split_s2_group_kernel(RfxGroupArgs<C3S2Args> g) {
    const unsigned y = blockIdx.y;
    if (blockIdx.x >= g.gx[y]) return;
    conv3x3_split_s2_body(g.p[y], blockIdx.x);
}

static int c3s2_group_launch(const void* blob, const unsigned* gx, int n, hipStream_t st) {
    return rfx_group_launch_impl<C3S2Args>(conv3x3_split_s2_group_kernel, 256, blob, gx, n, st);
}

}  // namespace

extern "C" int rfx_conv3x3_split_f32(const float* in, const void* wS3, const float* scale, const float* shift, const float* residual,
                                     float* out, int N, int Cin, int H, int W, int Cout, int act, void* stream) {
    if (!in || !wS3 || !out || N <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) return RFX_E_ARG;
    if (Cin % 16 != 0) return RFX_E_ARG;
    if (act != RFX_ACT_NONE && act != RFX_ACT_RELU && act != RFX_ACT_SIGMOID) return RFX_E_ARG;
    if ((long long)N * (H + 1) > 0x7fffffffLL) return RFX_E_LIMIT;
    C3SArgs a;
    a.in = in; a.wS = reinterpret_cast<const u32x4*>(wS3); a.scale = scale; a.shift = shift; a.res = residual; a.out = out;
    a.N = N; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.act = act; a.Mpad = (Cout + 127) / 128 * 128;
    return Cout > 64 ? launch_split3<2>(a, rfx_stream(stream)) : launch_split3<1>(a, rfx_stream(stream));
}

extern "C" int rfx_conv3x3_split_s2_f32(const float* in, const void* wS3, const float* scale, const float* shift, const float* residual,
                                        float* out, int N, int Cin, int H, int W, int Cout, int act, void* stream) {
    if (!in || !wS3 || !out || N <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) return RFX_E_ARG;
    if (Cin % 16 != 0) return RFX_E_ARG;
    if (act != RFX_ACT_NONE && act != RFX_ACT_RELU && act != RFX_ACT_SIGMOID) return RFX_E_ARG;
    C3S2Args a;
    a.in = in; a.wS = reinterpret_cast<const u32x4*>(wS3); a.scale = scale; a.shift = shift; a.res = residual; a.out = out;
    a.N = N; a.Cin = Cin; a.H = H; a.W = W; a.Ho = (H - 1) / 2 + 1; a.Wo = (W - 1) / 2 + 1; a.Cout = Cout; a.act = act;
    a.Mpad = (Cout + 127) / 128 * 128;
    if ((long long)N * (a.Ho + 1) * 2 > 0x7fffffffLL) return RFX_E_LIMIT;
    a.tilesM = (Cout + 127) / 128;
    a.tilesW = (a.Wo + 15) / 16;
    a.tilesS = (int)(((long long)N * (a.Ho + 1) - 1 + 7) / 8);
    const long long nwg = (long long)a.tilesM * a.tilesW * a.tilesS;
    if (nwg > 0x7fffffffLL) return RFX_E_LIMIT;
    hipStream_t st = rfx_stream(stream);
    if (rfx_group_recording()) return rfx_group_record(&c3s2_group_launch, &a, sizeof(a), (unsigned)nwg);
    hipLaunchKernelGGL(conv3x3_split_s2_kernel, dim3((unsigned)nwg), dim3(256), 0, st, a);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}
