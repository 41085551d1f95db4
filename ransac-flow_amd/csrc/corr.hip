// corr.hip -- 7x7 local correlation volume (CorrNeigh, model/model.py:129-160):
//     out[n, i*7+j, r, c] = sum_ch x[n,ch,r,c] * y[n,ch,r+i-3,c+j-3]        (y zero outside the image)
//
// The reference runs 49 separate multiply+reduce kernels that each re-read x and y (49x the minimal
// traffic).  Here every x/y element is fetched from HBM once per tile (+ halo), staged in LDS by the
// LDS-DMA path (global_load_lds_dwordx4: no VGPR round trip, zero fill by pointing out-of-image
// lanes at a 16-byte zero block) and all 49 taps of 4 horizontally adjacent pixels are accumulated in
// registers (4 px x 28 / 21 taps per lane), so one LDS float feeds ~2.3 FMAs and the kernel sits on the
// fp32 VALU, next to the HBM roofline (algorithmic intensity 11 FLOP/B).
//
// Workgroup = TR*8 threads = tile of TR (64/32/16) rows x 16 cols, two wavefronts (tap-row groups) per 16-row
// strip; lane -> (tc = lane>>4: 4-pixel column group, r = lane&15: row inside the wave's 16-row strip).  LDS per channel: y halo tile [70][24] floats (row
// stride 96 B: the 16-lane ds_read_b128 service groups hit 16 distinct 16-B slots -> conflict free),
// x tile [TR][16].  Channels are streamed CK=2 at a time through a 3-deep LDS ring: one raw s_barrier
// per chunk and a COUNTED s_waitcnt vmcnt(PPW), so the DMA of chunks s+1 and s+2 stays in flight across
// the barrier while chunk s is on the VALU (a __syncthreads() would drain it with vmcnt(0)).
// Channel sums are accumulated in channel order with fmaf (deterministic).
//
// Requires W % 4 == 0 for the 16-byte DMA path; other widths use the plain fallback kernel below.
#include "common.h"

namespace {

constexpr int TC = 16;           // tile cols
constexpr int YQ = 6;            // float4 per halo row (cols c0-4 .. c0+19)
constexpr int CK = 2;            // channels per chunk
constexpr int NS = 3;            // LDS ring depth: the DMA of chunks s+1 and s+2 is in flight while chunk s computes

template <int TR>
struct Geo {
    static constexpr int NW = TR / 16 * 2;                     // waves per workgroup: 16-row strips x 2 tap-row groups
    static constexpr int YR = TR + 6;                          // halo rows
    static constexpr int Y_SLOTS = CK * YR * YQ;               // float4 slots of the y halo tile
    static constexpr int Y_PIECES = (Y_SLOTS + 63) / 64;
    static constexpr int X_SLOTS = CK * TR * (TC / 4);
    static constexpr int X_PIECES = X_SLOTS / 64;
    static constexpr int PPW = (Y_PIECES + X_PIECES + NW - 1) / NW;   // DMA pieces per wave per chunk (padded)
    static constexpr int N_PIECES = PPW * NW;                  // incl. padding pieces (zero source, dump area)
    static constexpr int BUF_SLOTS = N_PIECES * 64;
};

__device__ __attribute__((aligned(16))) float rfx_zero16[4] = {0.f, 0.f, 0.f, 0.f};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// One wavefront = one 16-row strip of the tile x one group of window rows: group 0 accumulates taps i = 0..3
// (112 accumulators / lane), group 1 taps i = 4..6 (84).  Splitting the 49 taps over two wavefronts halves the
// register footprint (3 instead of 2 wavefronts per SIMD) and doubles the number of wavefronts a small batch
// decomposes into; both groups read the same LDS tile, so the DMA traffic is unchanged.
template <int TR, int I0, int I1>
__device__ __forceinline__ void corr7_strip(f32x4 (*smem)[Geo<TR>::BUF_SLOTS], const float* xn, const float* yn,
                                            const int* off, int wave, int strip, int lane, int nchunks, size_t HW,
                                            float* __restrict__ out, int n, int row0, int c0, int H, int W) {
    using G = Geo<TR>;
    constexpr int NI = I1 - I0;
    // every wave issues exactly PPW DMA instructions per chunk, so that the counted vmcnt below is uniform
    auto issue = [&](int chunk, int buf) {
        const size_t cbase = (size_t)chunk * CK * HW;
#pragma unroll
        for (int i = 0; i < G::PPW; ++i) {
            const int pi = wave + G::NW * i;
            const float* base = (pi < G::Y_PIECES ? yn : xn) + cbase;
            const float* src = off[i] >= 0 ? base + off[i] : rfx_zero16;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(&smem[buf][pi * 64]), 16, 0, 0);
        }
    };
    float acc[4][NI * 7];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int q = 0; q < NI * 7; ++q) acc[d][q] = 0.f;
    const int tc = lane >> 4;
    const int tr = strip * 16 + (lane & 15);

    issue(0, 0);
    if (nchunks > 1) issue(1, 1);
    int buf = 0;
    for (int s = 0; s < nchunks; ++s) {
        // chunk s must have landed; chunk s+1 (PPW DMAs of this wave) may stay in flight across the barrier
        if (s + 1 < nchunks) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // all waves: chunk s visible, and everyone is done reading buffer (s-1)%NS
        if (s + 2 < nchunks) issue(s + 2, buf == 0 ? 2 : buf - 1);  // (s+2)%NS == (s-1)%NS
        const f32x4* yb = &smem[buf][0];
        const f32x4* xb = &smem[buf][G::Y_PIECES * 64];
#pragma unroll
        for (int ch = 0; ch < CK; ++ch) {
            const f32x4 xv = xb[ch * (TR * 4) + tr * 4 + tc];
#pragma unroll
            for (int i = I0; i < I1; ++i) {
                const f32x4* yrow = yb + ch * (G::YR * YQ) + (tr + i) * YQ + tc;
                const f32x4 w0 = yrow[0], w1 = yrow[1], w2 = yrow[2];
                const float yw[12] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3], w2[0], w2[1], w2[2], w2[3]};
#pragma unroll
                for (int d = 0; d < 4; ++d)
#pragma unroll
                    for (int j = 0; j < 7; ++j)
                        acc[d][(i - I0) * 7 + j] = fmaf(xv[d], yw[d + j + 1], acc[d][(i - I0) * 7 + j]);
            }
        }
        buf = buf == NS - 1 ? 0 : buf + 1;
    }
    const int gr = row0 + tr, gc = c0 + 4 * tc;
    if (gr < H && gc < W) {
        float* o = out + (size_t)n * 49 * HW + (size_t)gr * W + gc;
#pragma unroll
        for (int q = 0; q < NI * 7; ++q) {
            f32x4 v = {acc[0][q], acc[1][q], acc[2][q], acc[3][q]};
            *reinterpret_cast<f32x4*>(o + (size_t)(I0 * 7 + q) * HW) = v;
        }
    }
}

template <int TR>
__global__ __launch_bounds__(TR * 8, 3) void corr7_dma_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                              float* __restrict__ out, int N, int C, int H, int W,
                                                              int tilesR, int tilesC) {
    using G = Geo<TR>;
    __shared__ __attribute__((aligned(16))) f32x4 smem[NS][G::BUF_SLOTS];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int tpi = tilesR * tilesC;
    const int nwg = N * tpi;
    int bid = blockIdx.x;
    {   // XCD-aware bijective remap: all tiles of one image on one XCD (halo re-reads hit that L2)
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, j = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int n = bid / tpi;
    const int tile = bid - n * tpi;
    const int row0 = (tile / tilesC) * TR, c0 = (tile % tilesC) * TC;
    const size_t HW = (size_t)H * W;
    const float* xn = x + (size_t)n * C * HW;
    const float* yn = y + (size_t)n * C * HW;

    // per-lane source offsets of the PPW DMA pieces this wave issues per chunk; -1 = 16-byte zero block
    int off[G::PPW];
#pragma unroll
    for (int i = 0; i < G::PPW; ++i) {
        const int pi = wave + G::NW * i;
        int o = -1;
        if (pi < G::Y_PIECES) {
            const int s = pi * 64 + lane;
            if (s < G::Y_SLOTS) {
                const int ch = s / (G::YR * YQ), rem = s - ch * (G::YR * YQ);
                const int rr = rem / YQ, q = rem - rr * YQ;
                const int gr = row0 + rr - 3, gc = c0 - 4 + 4 * q;
                if ((unsigned)gr < (unsigned)H && (unsigned)gc < (unsigned)W) o = (int)(ch * HW) + gr * W + gc;
            }
        } else if (pi < G::Y_PIECES + G::X_PIECES) {
            const int s = (pi - G::Y_PIECES) * 64 + lane;
            const int ch = s / (TR * 4), rem = s - ch * (TR * 4);
            const int rr = rem / 4, q = rem - rr * 4;
            const int gr = row0 + rr, gc = c0 + 4 * q;
            if (gr < H && gc < W) o = (int)(ch * HW) + gr * W + gc;
        }
        off[i] = o;
    }
    const int strip = wave >> 1;
    if ((wave & 1) == 0)
        corr7_strip<TR, 0, 4>(smem, xn, yn, off, wave, strip, lane, C / CK, HW, out, n, row0, c0, H, W);
    else
        corr7_strip<TR, 4, 7>(smem, xn, yn, off, wave, strip, lane, C / CK, HW, out, n, row0, c0, H, W);
}

template <int TR>
static void launch_corr(const float* x, const float* y, float* out, int N, int C, int H, int W, hipStream_t st) {
    const int tilesR = (H + TR - 1) / TR, tilesC = (W + TC - 1) / TC;
    hipLaunchKernelGGL((corr7_dma_kernel<TR>), dim3((unsigned)(N * tilesR * tilesC)), dim3(TR * 8), 0, st, x, y, out, N, C,
                       H, W, tilesR, tilesC);
}

// Plain fallback for widths that are not a multiple of 4 (never hit by the reference's /8 feature maps of
// x16-rounded images, kept so the entry point is total): one thread per output pixel.
__global__ __launch_bounds__(256) void corr7_plain_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                          float* __restrict__ out, long long NP, int C, int H, int W) {
    const size_t HW = (size_t)H * W;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < NP;
         p += (long long)gridDim.x * blockDim.x) {
        const long long n = p / HW;
        const int px = (int)(p - n * HW);
        const int r = px / W, c = px - r * W;
        const float* xn = x + (size_t)n * C * HW;
        const float* yn = y + (size_t)n * C * HW;
        for (int i = 0; i < 7; ++i)
            for (int j = 0; j < 7; ++j) {
                const int yr = r + i - 3, yc = c + j - 3;
                float s = 0.f;
                if ((unsigned)yr < (unsigned)H && (unsigned)yc < (unsigned)W)
                    for (int ch = 0; ch < C; ++ch)
                        s = fmaf(xn[ch * HW + px], yn[ch * HW + (size_t)yr * W + yc], s);
                out[(size_t)n * 49 * HW + (size_t)(i * 7 + j) * HW + px] = s;
            }
    }
}

}  // namespace

extern "C" int rfx_corr_neigh_f32(const float* x, const float* y, float* out, int N, int C, int H, int W, int K,
                                  void* stream) {
    if (!x || !y || !out || N <= 0 || C <= 0 || H <= 0 || W <= 0) return RFX_E_ARG;
    if (K != 7) return RFX_E_ARG;
    if ((long long)C * H * W > 0x7fffffffLL) return RFX_E_LIMIT;
    hipStream_t st = rfx_stream(stream);
    if (W % 4 == 0 && C % CK == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(y) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
        // tile height by parallelism: enough workgroups to spread over all 256 CUs several times (the DMA ring
        // hides latency best when other wavefronts can compute meanwhile); taller tiles amortise the 6-row halo
        const long long tc = (W + TC - 1) / TC;
        const long long b64 = (long long)N * ((H + 63) / 64) * tc, b32 = (long long)N * ((H + 31) / 32) * tc;
        if ((long long)N * ((H + 15) / 16) * tc > 0x7fffffffLL) return RFX_E_LIMIT;
        if (b64 >= 1024) launch_corr<64>(x, y, out, N, C, H, W, st);
        else if (b32 >= 1024) launch_corr<32>(x, y, out, N, C, H, W, st);
        else launch_corr<16>(x, y, out, N, C, H, W, st);
    } else {
        const long long NP = (long long)N * H * W;
        long long g = (NP + 255) / 256;
        if (g > 8192) g = 8192;
        hipLaunchKernelGGL(corr7_plain_kernel, dim3((unsigned)g), dim3(256), 0, st, x, y, out, NP, C, H, W);
    }
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}
