// conv1x1s.hip -- 1x1 / stride 1 convolution (+ folded BatchNorm, residual, ReLU) with float32 results on the bf16 matrix pipe by
// EXACT operand splitting (round 6; the Bottleneck conv1 / conv3 layers of the ResNet-50 trunk, model/resnet50.py:71-79,93-103).
//
// A float32 x has a 24-bit significand = three bf16 pieces of 8 bits:  x = hi + mid + lo  EXACTLY, with hi = bf16(x) (round to
// nearest even), mid = bf16(x - hi), lo = bf16(x - hi - mid) (both differences are exact in float32).  A product of two pieces
// (8 x 8 bits) is exact in float32, so
//     w * x = (wh + wm + wl)(xh + xm + xl) = wh xh + [wh xm + wm xh] + [wm xm + wh xl + wl xh] + (three terms below 2^-32 |w x|)
// and six v_mfma_f32_32x32x16_bf16 per 16 k give the float32 sum with EVERY product exact -- what differs from an fp32 fma chain is
// only where the float32 accumulator rounds: the matrix core adds the 16 products of an instruction before it rounds, and the hi*hi
// products run in their own accumulator (the five small terms in a second one, added once at the end).  Measured on convolution-
// shaped data (scripts/ubench/bf16x_emul.hip, profiles/r06_bf16_split_study.json): rms error against the float64 sum 2.3e-7 of the
// output rms at K = 2304, against 6.1e-7 for the fp32 MFMA's single fma chain and 3.0e-7 for the chunked chain the fp32 kernels use
// -- CLOSER to the exact sum than the float32 kernels it replaces, at 2.1x their matrix-pipe rate (6 x 32 cycles against 8 x 64 per 16 k).
// Not bit-identical to the fp32 kernels (conv1x1.hip stays: rfx_conv2d_f32 never routes here; the caller asks for this entry point).
//
// Operands.  Weights: split ONCE on the host into the three pieces and packed in fragment order (rfx_api.h: "wS"):
//     wS[kb = k / 16][piece][h = (k % 16) / 8][m (Mpad)][8 bf16]     -- a lane's A fragment (row m, k = 16 kb + 8 h .. + 7) = one 16-byte word
// Activations: float32 NCHW in HBM as everywhere; a workgroup's staging threads load 8 consecutive k of one pixel (8 dwords, a
// wavefront = 64 consecutive pixels = 256 contiguous bytes per load), split them with v_cvt_pk_bf16_f32 + v_pk_add_f32
// (4.5 vector-ALU instructions per element, once per workgroup) and store three 16-byte words into the LDS image
//     Bs[piece][h][pixel (128)][8 bf16]
// so that both fragments of an MFMA are single conflict-free ds_read_b128 (consecutive lanes, consecutive words).
// Tile 64*TM channels x 128 pixels, 2 x 2 wavefronts, one 16-k block per stage, LDS double buffered, one barrier per stage:
// registers hold block s+1 while block s runs on the matrix pipe and block s+2 is in flight.
// Infinities: x = +-inf gives hi = inf, x - hi = NaN -> NaN where the fp32 kernel returns +-inf (NaN inputs give NaN in both).
#include "common.h"
#include "conv_epilogue.h"
#include "group.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct C1SArgs {
    const float* in; const u32x4* wS; const float* scale; const float* shift; const float* res; float* out;
    int Cin, HW, Cout, act, Mpad;      // HW: OUTPUT pixels per image
    long long P;   // N*HW
    int tilesM, tilesP;
    int stride, Win, Wo, HWin;         // stride 2 (the projection shortcuts, model/resnet50.py:139-143): input pixel (2y, 2x) of a Win-wide map
};

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {      // (bf16(a) | bf16(b) << 16), round to nearest even: v_cvt_pk_bf16_f32
    const f32x2 v = {a, b};
    const bf16x2 h = __builtin_convertvector(v, bf16x2);
    unsigned u;
    __builtin_memcpy(&u, &h, 4);
    return u;
}
// two float32 -> their three bf16 pieces, packed pairwise
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

template <int TM>
__device__ __forceinline__ void conv1x1_split_body(const C1SArgs& a, const unsigned bx) {
    constexpr int BM = 64 * TM, BN = 128, KB = 16;
    constexpr int A_WORDS = 3 * 2 * BM;                 // 16-byte words of a stage's weight image: 768 / 384
    constexpr int NA = (A_WORDS + 255) / 256;           // per thread: 3 / 2 (the second one only for t < 128)
    __shared__ u32x4 As[2][3][2][BM];
    __shared__ u32x4 Bs[2][3][2][BN];
    __shared__ float s_scale[BM], s_shift[BM];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = lane >> 5, lcol = lane & 31;
    const size_t HW = (size_t)a.HW, HWin = (size_t)a.HWin;
    const int nwg = a.tilesM * a.tilesP;
    const int nk = a.Cin / KB;

    // tile -> (m0, n0): XCD-aware remap, m-tile fastest (the workgroups that share one pixel tile sit on one L2)
    int m0; long long n0;
    {
        const int v = (int)bx, q = nwg / 8, r = nwg % 8, xcd = v % 8, j = v / 8;
        const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
        m0 = (bid % a.tilesM) * BM;
        n0 = (long long)(bid / a.tilesM) * BN;
    }
    // staging roles.  A: words t + 256 j of the stage image [piece][h][BM] <- wS[kb][piece][h][m0 + m]
    constexpr int A_ROWS = 256 / BM;                                                  // word t + 256 j sits A_ROWS * j [piece][h] rows below word t
    const u32x4* wsrc = a.wS + (size_t)(t / BM) * a.Mpad + m0 + t % BM;              // + (kb * 6 + A_ROWS * j) * Mpad
    // piece j exists for every thread (compile time) or for the first wavefronts only (TM = 1: 384 words): no per-lane branches around
    // the loads -- a divergent region makes the compiler drain the vector-memory counter between two loads
    auto a_on = [&](int j) { return (j + 1) * 256 <= A_WORDS || t + 256 * j < A_WORDS; };
    // B: thread = (h = t >> 7, pixel t & 127): rows k0 + 8 h .. + 7 of that pixel
    const int bh = t >> 7, bp = t & 127;
    const float* bsrc;
    {
        long long p = n0 + bp;
        if (p >= a.P) p = a.P - 1;                                                  // columns past the end: any valid address
        const long long n = p / a.HW;
        int off = (int)(p - n * a.HW);
        if (a.stride != 1) { const int y = off / a.Wo, x = off - y * a.Wo; off = y * a.stride * a.Win + x * a.stride; }
        bsrc = a.in + ((size_t)n * a.Cin + 8 * bh) * HWin + (size_t)off;                          // + k0 * HWin
    }
    u32x4 ra[NA];
    float rb[8];
    auto load_stage = [&](int kb) {
#pragma unroll
        for (int j = 0; j < NA; ++j)
            ra[j] = wsrc[((size_t)kb * 6 + (a_on(j) ? A_ROWS * j : 0)) * a.Mpad];              // off lanes: any valid word
#pragma unroll
        for (int i = 0; i < 8; ++i) rb[i] = bsrc[(size_t)(kb * KB + i) * HWin];
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NA; ++j)
            if (a_on(j)) (&As[buf][0][0][0])[t + 256 * j] = ra[j];
        u32x4 hi, mid, lo;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned h_, m_, l_;
            split_pair(rb[2 * i], rb[2 * i + 1], h_, m_, l_);
            hi[i] = h_; mid[i] = m_; lo[i] = l_;
        }
        Bs[buf][0][bh][bp] = hi;
        Bs[buf][1][bh][bp] = mid;
        Bs[buf][2][bh][bp] = lo;
    };
    if (t < BM) {
        const int m = m0 + t;
        s_scale[t] = (a.scale && m < a.Cout) ? a.scale[m] : 1.0f;
        s_shift[t] = (a.shift && m < a.Cout) ? a.shift[m] : 0.0f;
    }
    load_stage(0);
    store_stage(0);
    if (nk > 1) load_stage(1);
    __syncthreads();

    f32x16 acc[TM][2], low[TM][2];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.0f; low[i][j][r] = 0.0f; }

    for (int s = 0; s < nk; ++s) {
        const int cur = s & 1;
        // fragments of block s: A rows wm*TM*32 + i*32 + lcol, B pixels wn*64 + j*32 + lcol, k half lrow
        bf16x8 af[3][TM], bf[3][2];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[pc][i] = as_frag(As[cur][pc][lrow][wm * TM * 32 + i * 32 + lcol]);
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[pc][j] = as_frag(Bs[cur][pc][lrow][wn * 64 + j * 32 + lcol]);
        }
        if (s + 1 < nk) store_stage(cur ^ 1);          // block s+1: registers -> the other buffer (its readers passed the last barrier)
        if (s + 2 < nk) load_stage(s + 2);
        __builtin_amdgcn_sched_barrier(0);
        // smallest terms first into the low accumulator; hi*hi alone in the main one
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                low[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][i], bf[0][j], low[i][j], 0, 0, 0);    // lo  * hi
                low[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[2][j], low[i][j], 0, 0, 0);    // hi  * lo
                low[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[1][j], low[i][j], 0, 0, 0);    // mid * mid
            }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                low[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[0][j], low[i][j], 0, 0, 0);    // mid * hi
                low[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[1][j], low[i][j], 0, 0, 0);    // hi  * mid
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[0][j], acc[i][j], 0, 0, 0);    // hi  * hi
            }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
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
        long long pp = n0 + wn * 64 + j * 32 + lcol;
        pix_ok[j] = pp < a.P;
        if (!pix_ok[j]) pp = a.P - 1;
        const long long n = pp / a.HW;
        pix_off[j] = (size_t)n * a.Cout * HW + (size_t)(pp - n * a.HW);
    }
    conv_epilogue<TM, 2, false>(acc, s_scale, s_shift, a.res, a.out, a.act, a.Cout, HW, m0, wm, lrow, pix_off, pix_ok, m0 + BM <= a.Cout);
}

template <int TM>
__global__ __launch_bounds__(256, 2) void conv1x1_split_kernel(C1SArgs a) {
    conv1x1_split_body<TM>(a, blockIdx.x);
}

template <int TM>
__global__ __launch_bounds__(256, 2) void conv1x1_split_group_kernel(RfxGroupArgs<C1SArgs> g) {
    const unsigned y = blockIdx.y;
    if (blockIdx.x >= g.gx[y]) return;
    conv1x1_split_body<TM>(g.p[y], blockIdx.x);
}

template <int TM>
static int c1s_group_launch(const void* blob, const unsigned* gx, int n, hipStream_t st) {
    return rfx_group_launch_impl<C1SArgs>(conv1x1_split_group_kernel<TM>, 256, blob, gx, n, st);
}

template <int TM>
int launch_split(C1SArgs& a, hipStream_t st) {
    a.tilesM = (a.Cout + 64 * TM - 1) / (64 * TM);
    a.tilesP = (int)((a.P + 127) / 128);
    const long long nwg = (long long)a.tilesM * a.tilesP;
    if (nwg > 0x7fffffffLL) return RFX_E_LIMIT;
    if (rfx_group_recording()) return rfx_group_record(&c1s_group_launch<TM>, &a, sizeof(a), (unsigned)nwg);
    hipLaunchKernelGGL((conv1x1_split_kernel<TM>), dim3((unsigned)nwg), dim3(256), 0, st, a);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

}  // namespace

static int conv1x1_split_launch(const float* in, const void* wS, const float* scale, const float* shift, const float* residual,
                                float* out, int N, int Cin, int Hin, int Win, int Cout, int stride, int act, void* stream) {
    if (!in || !wS || !out || N <= 0 || Cin <= 0 || Hin <= 0 || Win <= 0 || Cout <= 0) return RFX_E_ARG;
    if (Cin % 16 != 0 || (stride != 1 && stride != 2)) return RFX_E_ARG;
    if (act != RFX_ACT_NONE && act != RFX_ACT_RELU && act != RFX_ACT_SIGMOID) return RFX_E_ARG;
    const int Ho = (Hin - 1) / stride + 1, Wo = (Win - 1) / stride + 1;
    C1SArgs a;
    a.in = in; a.wS = reinterpret_cast<const u32x4*>(wS); a.scale = scale; a.shift = shift; a.res = residual; a.out = out;
    a.Cin = Cin; a.HW = Ho * Wo; a.Cout = Cout; a.act = act; a.Mpad = (Cout + 127) / 128 * 128;
    a.P = (long long)N * a.HW;
    a.stride = stride; a.Win = Win; a.Wo = Wo; a.HWin = Hin * Win;
    return Cout > 64 ? launch_split<2>(a, rfx_stream(stream)) : launch_split<1>(a, rfx_stream(stream));
}

extern "C" int rfx_conv1x1_split_f32(const float* in, const void* wS, const float* scale, const float* shift, const float* residual,
                                     float* out, int N, int Cin, int HW, int Cout, int act, void* stream) {
    return conv1x1_split_launch(in, wS, scale, shift, residual, out, N, Cin, 1, HW, Cout, 1, act, stream);
}

extern "C" int rfx_conv1x1_split_strided_f32(const float* in, const void* wS, const float* scale, const float* shift, const float* residual,
                                             float* out, int N, int Cin, int Hin, int Win, int Cout, int stride, int act, void* stream) {
    return conv1x1_split_launch(in, wS, scale, shift, residual, out, N, Cin, Hin, Win, Cout, stride, act, stream);
}
