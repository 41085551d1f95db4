// ransac.hip -- RANSAC over 4-point DLT homographies (utils/outil.py:68-164).
//
// The reference drives the loop from the host in chunks of 100 hypotheses: 16 device->host copies, a CPU
// LAPACK SVD, one host->device copy and >= 2 synchronising reads per chunk (utils/outil.py:73,84,86,
// 143-150).  Here the whole search is four asynchronous launches with no host round trip:
//   pack     matches -> (x1,y1,x2,y2) float4 + z2, so scoring issues one 16-byte load per match
//   dlt      one hypothesis per lane: duplicate test (:122-133), float64 Householder DLT (dlt.h,
//            LAPACK-sign-exact), float32 cast, det(H) > 1e-6 gate with torch.det's LU order (:108,113)
//   count    one hypothesis per wavefront: lanes stride over the matches, reprojection error in the
//            reference's float32 operation order (k-ordered fma chain for Y.H^T as sgemm does, IEEE
//            divide, separate squares / add / sqrt), __ballot + popcount for the inlier tally (:97-100,110-113)
//   select   chunk-of-100 zero-abort (:140-146), first maximum over the filtered order (:143-160), inlier
//            mask of the winner (:162-163)
// This TU is compiled with -ffp-contract=off so that only the fmaf calls written below fuse.
#include "common.h"
#include "dlt.h"
#include <math.h>

namespace {

constexpr int CHUNK = 100;         // nbMaxIter, utils/outil.py:136
constexpr int MAX_CHUNKS = 8192;   // LDS table in select kernel -> N <= 819200 hypotheses

// Batched forms (rfx_ransac_h4_batched): blockIdx.y = pair b; every per-pair array is `cap` matches / N hypotheses
// apart and the match count comes from the device array narr (nullptr = the single-pair entry points).
__global__ __launch_bounds__(256) void ransac_pack_kernel(const float* __restrict__ m1, const float* __restrict__ m2,
                                                          int n, float4* __restrict__ P, float* __restrict__ Z,
                                                          const int32_t* __restrict__ narr, int cap, size_t wsStride) {
    if (narr) {
        const int b = blockIdx.y;
        n = narr[b];
        m1 += (size_t)b * cap * 3; m2 += (size_t)b * cap * 3;
        P = reinterpret_cast<float4*>(reinterpret_cast<char*>(P) + b * wsStride);
        Z = reinterpret_cast<float*>(reinterpret_cast<char*>(Z) + b * wsStride);
    }
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        float4 p;
        p.x = m1[i * 3 + 0]; p.y = m1[i * 3 + 1];
        p.z = m2[i * 3 + 0]; p.w = m2[i * 3 + 1];
        P[i] = p;
        Z[i] = m2[i * 3 + 2];
    }
}

// X,Y given either as gathered samples (N,4,3) [direct != 0] or through indices into the match arrays.
// WANT_RANK: also extract the bidiagonal and run the Sturm count behind flag bit 2 (15 fp64 registers and divisions per
// hypothesis) -- only the callers that read the bit ask for it (the staged "lapack" search, the *_flags entry point,
// rfx_score_hypotheses with a `degenerate` output); the throughput path (rfx_ransac_h4[_batched]) does not pay for it.
template <bool WANT_RANK>
__global__ __launch_bounds__(64) void ransac_dlt_kernel(const float* __restrict__ m1, const float* __restrict__ m2, int n,
                                                        const int64_t* __restrict__ samples, int N, int filter_dup,
                                                        float* __restrict__ Hout, uint8_t* __restrict__ flags,
                                                        const int32_t* __restrict__ narr, int cap, size_t wsStride) {
    if (narr) {
        const int b = blockIdx.y;
        n = narr[b];
        m1 += (size_t)b * cap * 3; m2 += (size_t)b * cap * 3;
        samples += (size_t)b * N * 4;
        Hout = reinterpret_cast<float*>(reinterpret_cast<char*>(Hout) + b * wsStride);
        flags += b * wsStride;
    }
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= N) return;
    float src[4][2], tgt[4][2];
    bool dup = false, bad = narr && n < 4;   // batched: a pair with fewer than nbPoint matches evaluates nothing
    if (samples) {
        int64_t s[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) s[p] = samples[(size_t)h * 4 + p];
        dup = (s[0] == s[1]) | (s[0] == s[2]) | (s[0] == s[3]) | (s[1] == s[2]) | (s[1] == s[3]) | (s[2] == s[3]);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            int64_t q = s[p];
            if (q < 0) q += n;  // python-style negative index
            if (q < 0 || q >= n) { bad = true; q = 0; }
            src[p][0] = m1[q * 3 + 0]; src[p][1] = m1[q * 3 + 1];
            tgt[p][0] = m2[q * 3 + 0]; tgt[p][1] = m2[q * 3 + 1];
        }
    } else {  // m1 = X (N,4,3), m2 = Y (N,4,3)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            src[p][0] = m1[((size_t)h * 4 + p) * 3 + 0]; src[p][1] = m1[((size_t)h * 4 + p) * 3 + 1];
            tgt[p][0] = m2[((size_t)h * 4 + p) * 3 + 0]; tgt[p][1] = m2[((size_t)h * 4 + p) * 3 + 1];
        }
    }
    uint8_t fl = 0;  // bit0: evaluated (survives the duplicate filter), bit1: det gate passed, bit2: rank-deficient system
    if (!(filter_dup && dup) && !bad) {
        double hv[9], bd[15];
        rfx_dlt4_nullvec(src, tgt, hv, WANT_RANK ? bd : nullptr);
        float hf[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) { hf[j] = (float)hv[j]; Hout[(size_t)h * 9 + j] = hf[j]; }
        fl = 1 | (rfx_det3_lu_f32(hf) > 1e-6f ? 2 : 0);   // torch.det's own LU order (dlt.h)
        if (WANT_RANK) fl |= rfx_bidiag_rank_deficient(bd, RFX_DLT_RANK_REL) ? 4 : 0;
    } else {
#pragma unroll
        for (int j = 0; j < 9; ++j) Hout[(size_t)h * 9 + j] = 0.0f;
    }
    if (flags) flags[h] = fl;
}

__device__ __forceinline__ float reproj_error(const float4 p, const float z, const float* H) {
    // estimX = Y @ H^T : k-ordered fma chain (what the reference's K=3 sgemm evaluates)
    const float px = fmaf(z, H[2], fmaf(p.w, H[1], __fmul_rn(p.z, H[0])));
    const float py = fmaf(z, H[5], fmaf(p.w, H[4], __fmul_rn(p.z, H[3])));
    const float pz = fmaf(z, H[8], fmaf(p.w, H[7], __fmul_rn(p.z, H[6])));
    const float ex = __fdiv_rn(px, pz), ey = __fdiv_rn(py, pz);
    const float dx = __fsub_rn(p.x, ex), dy = __fsub_rn(p.y, ey);
    const float d2 = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));
    return __fsqrt_rn(d2);
}

__device__ __forceinline__ bool is_inlier(const float4 p, const float z, const float* H, const float tol) {
    return reproj_error(p, z, H) < tol;  // NaN (pz == 0) compares false, as in the reference
}

// outil.Prediction (utils/outil.py:97-100): err[h][m] for every (hypothesis, match) pair
__global__ __launch_bounds__(256) void prediction_kernel(const float* __restrict__ m1, const float* __restrict__ m2, int n,
                                                         const float* __restrict__ Hs, int N, float* __restrict__ err) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    const int h = blockIdx.y;
    if (m >= n) return;
    float H[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) H[j] = Hs[(size_t)h * 9 + j];
    float4 p;
    p.x = m1[m * 3 + 0]; p.y = m1[m * 3 + 1]; p.z = m2[m * 3 + 0]; p.w = m2[m * 3 + 1];
    err[(size_t)h * n + m] = reproj_error(p, m2[m * 3 + 2], H);
}

// one hypothesis per wavefront, 4 wavefronts per workgroup
__global__ __launch_bounds__(256) void ransac_count_kernel(const float4* __restrict__ P, const float* __restrict__ Z, int n,
                                                           const float* __restrict__ Hs, const uint8_t* __restrict__ flags,
                                                           int N, float tol, int* __restrict__ counts32,
                                                           int64_t* __restrict__ counts64,
                                                           const int32_t* __restrict__ narr, size_t wsStride) {
    if (narr) {
        const int b = blockIdx.y;
        n = narr[b];
        P = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(P) + b * wsStride);
        Z = reinterpret_cast<const float*>(reinterpret_cast<const char*>(Z) + b * wsStride);
        Hs = reinterpret_cast<const float*>(reinterpret_cast<const char*>(Hs) + b * wsStride);
        flags += b * wsStride;
        counts32 = reinterpret_cast<int*>(reinterpret_cast<char*>(counts32) + b * wsStride);
    }
    const int lane = threadIdx.x & 63;
    const int h = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (h >= N) return;
    const uint8_t fl = flags[h];
    int cnt = 0;
    if (fl & 1) {
        float H[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) H[j] = Hs[(size_t)h * 9 + j];  // wave-uniform -> scalar loads
        for (int m0 = 0; m0 < n; m0 += 64) {  // wave-uniform trip count (the ballot needs every lane)
            const int m = m0 + lane;
            bool in = false;
            if (m < n) in = is_inlier(P[m], Z[m], H, tol);
            cnt += __popcll(__ballot(in));
        }
        if (!(fl & 2)) cnt = 0;  // counts * (det > 1e-6)
    }
    if (lane == 0) {
        if (counts32) counts32[h] = (fl & 1) ? cnt : -1;
        if (counts64) counts64[h] = cnt;
    }
}

__global__ __launch_bounds__(1024) void ransac_select_kernel(const float4* __restrict__ P, const float* __restrict__ Z, int n,
                                                             const float* __restrict__ Hs, const uint8_t* __restrict__ flags,
                                                             const int* __restrict__ counts, int N, float tol,
                                                             float* __restrict__ bestH, uint8_t* __restrict__ inlier,
                                                             int32_t* __restrict__ result,
                                                             const int32_t* __restrict__ narr, int cap, size_t wsStride) {
    bool too_few = false;
    if (narr) {
        const int b = blockIdx.x;
        n = narr[b];
        too_few = n < 4;
        P = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(P) + b * wsStride);
        Z = reinterpret_cast<const float*>(reinterpret_cast<const char*>(Z) + b * wsStride);
        Hs = reinterpret_cast<const float*>(reinterpret_cast<const char*>(Hs) + b * wsStride);
        flags += b * wsStride;
        counts = reinterpret_cast<const int*>(reinterpret_cast<const char*>(counts) + b * wsStride);
        bestH += b * 9; inlier += (size_t)b * cap; result += b * 4;
    }
    __shared__ int chunkmax[MAX_CHUNKS];
    __shared__ int wsum[16];
    __shared__ int base;
    __shared__ unsigned long long bestkey;
    __shared__ int s_status;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int c = t; c < MAX_CHUNKS; c += 1024) chunkmax[c] = 0;
    if (t == 0) { base = 0; bestkey = 0ull; s_status = 0; }
    __syncthreads();
    // rank of every surviving hypothesis in the filtered order, per-chunk maxima, global first maximum
    for (int s = 0; s < N; s += 1024) {
        const int h = s + t;
        const bool keep = h < N && (flags[h] & 1);
        const unsigned long long bal = __ballot(keep);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(bal);
        __syncthreads();
        int woff = 0, tot = 0;
        for (int w = 0; w < 16; ++w) { const int c = wsum[w]; if (w < wave) woff += c; tot += c; }
        const int b = base;
        if (keep) {
            const int rank = b + woff + before;
            const int c = counts[h];
            atomicMax(&chunkmax[rank / CHUNK], c);
            // larger count wins; equal counts: smaller index wins
            const unsigned long long key = ((unsigned long long)(unsigned)c << 32) | (unsigned)(0xffffffffu - (unsigned)h);
            atomicMax(&bestkey, key);
        }
        __syncthreads();
        if (t == 0) base = b + tot;
        __syncthreads();
    }
    const int nUnique = base;
    const int nFull = nUnique / CHUNK;
    bool ab = false;
    for (int c = t; c < nFull; c += 1024) ab |= (chunkmax[c] == 0);
    if (ab) s_status = 1;  // benign race: all writers store 1
    __syncthreads();
    const unsigned long long key = bestkey;
    const int bestCnt = (int)(key >> 32);
    const int bestIdx = (int)(0xffffffffu - (unsigned)(key & 0xffffffffu));
    int status = s_status;
    if (status == 0 && (nUnique == 0 || bestCnt == 0)) status = 2;
    if (too_few) status = 3;
    if (t == 0) {
        result[0] = status;
        result[1] = status == 0 ? bestCnt : 0;
        result[2] = status == 0 ? bestIdx : -1;
        result[3] = nUnique;
    }
    const int nout = narr ? cap : n;   // batched: the padding rows of the mask are defined (0) too
    if (status != 0) {
        for (int m = t; m < nout; m += 1024) inlier[m] = 0;
        if (t < 9) bestH[t] = 0.0f;
        return;
    }
    float H[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) H[j] = Hs[(size_t)bestIdx * 9 + j];
    if (t < 9) bestH[t] = H[t];
    for (int m = t; m < nout; m += 1024) inlier[m] = (m < n && is_inlier(P[m], Z[m], H, tol)) ? 1 : 0;
}

// ---- rank-deficient hypotheses: list them (ascending, per pair), take re-solved homographies back -------------------------
__global__ __launch_bounds__(1024) void ransac_degen_list_kernel(const uint8_t* __restrict__ flags, int N, int32_t* __restrict__ idx,
                                                                 int32_t* __restrict__ cnt, int kcap, size_t wsStride) {
    const int b = blockIdx.x;
    flags += b * wsStride; idx += (size_t)b * kcap;
    __shared__ int wsum[16];
    __shared__ int base;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) base = 0;
    __syncthreads();
    for (int s = 0; s < N; s += 1024) {
        const int h = s + t;
        const bool hit = h < N && (flags[h] & 5) == 5;       // evaluated AND rank deficient
        const unsigned long long bal = __ballot(hit);
        if (lane == 0) wsum[wave] = __popcll(bal);
        __syncthreads();
        int woff = 0, tot = 0;
        for (int w = 0; w < 16; ++w) { const int c = wsum[w]; if (w < wave) woff += c; tot += c; }
        const int pos = base + woff + __popcll(bal & ((1ull << lane) - 1ull));
        if (hit && pos < kcap) idx[pos] = h;
        __syncthreads();
        if (t == 0) base += tot;
        __syncthreads();
    }
    if (t == 0) cnt[b] = base;
}

__global__ __launch_bounds__(256) void ransac_patch_kernel(float* __restrict__ Hs, uint8_t* __restrict__ flags,
                                                           const int32_t* __restrict__ idx, const int32_t* __restrict__ cnt,
                                                           const float* __restrict__ Hp, int kcap, int N, size_t wsStride) {
    const int b = blockIdx.y, k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= kcap || k >= cnt[b]) return;
    const int h = idx[(size_t)b * kcap + k];
    if (h < 0 || h >= N) return;
    Hs = reinterpret_cast<float*>(reinterpret_cast<char*>(Hs) + b * wsStride);
    flags += b * wsStride;
    float hf[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) { hf[j] = Hp[((size_t)b * kcap + k) * 9 + j]; Hs[(size_t)h * 9 + j] = hf[j]; }
    flags[h] = (flags[h] & ~2) | (rfx_det3_lu_f32(hf) > 1e-6f ? 2 : 0);
}

// ---- compact form of the exact mode's exchange: ONE row per flagged hypothesis, all pairs back to back ---------------------
// off[b] = rows of the pairs before b (exclusive prefix of cnt), off[batch] = total; written to the device copy the later
// kernels read and to `off_out` (pinned host memory: what the host reads after the event)
__global__ __launch_bounds__(64) void ransac_degen_scan_kernel(const int32_t* __restrict__ cnt, int batch, int32_t* __restrict__ off,
                                                               int32_t* __restrict__ off_out) {
    if (threadIdx.x != 0) return;
    int acc = 0;
    for (int b = 0; b < batch; ++b) { off[b] = acc; if (off_out) off_out[b] = acc; acc += cnt[b]; }
    off[batch] = acc;
    if (off_out) off_out[batch] = acc;
}

// row off[b]+k = the 4 source points (x, y) then the 4 target points (x, y) of pair b's k-th flagged sample: X[:, :, :2] | Y[:, :, :2]
// of outil.Homography's arguments (utils/outil.py:68-71), 64 bytes, straight into pinned host memory
__global__ __launch_bounds__(256) void ransac_degen_gather_kernel(const float* __restrict__ m1, const float* __restrict__ m2,
                                                                  const int32_t* __restrict__ narr, int cap,
                                                                  const int64_t* __restrict__ samples, int N,
                                                                  const int32_t* __restrict__ idx, const int32_t* __restrict__ cnt,
                                                                  const int32_t* __restrict__ off, float4* __restrict__ xy,
                                                                  int row_cap) {
    const int b = blockIdx.y, k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= cnt[b] || k >= N) return;
    const int row = off[b] + k;
    if (row >= row_cap) return;
    const int h = idx[(size_t)b * N + k];
    const int n = narr[b];
    m1 += (size_t)b * cap * 3; m2 += (size_t)b * cap * 3;
    float v[16];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        int64_t q = samples[((size_t)b * N + h) * 4 + p];
        if (q < 0) q += n;                                   // python-style negative index (as in the DLT kernel)
        if (q < 0 || q >= n) q = 0;
        v[2 * p] = m1[q * 3 + 0]; v[2 * p + 1] = m1[q * 3 + 1];
        v[8 + 2 * p] = m2[q * 3 + 0]; v[8 + 2 * p + 1] = m2[q * 3 + 1];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) xy[(size_t)row * 4 + j] = make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
}

__global__ __launch_bounds__(256) void ransac_patch_compact_kernel(float* __restrict__ Hs, uint8_t* __restrict__ flags,
                                                                   const int32_t* __restrict__ idx, const int32_t* __restrict__ cnt,
                                                                   const int32_t* __restrict__ off, const float* __restrict__ Hp,
                                                                   int row_cap, int N, size_t wsStride) {
    const int b = blockIdx.y, k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= cnt[b] || k >= N) return;
    const int row = off[b] + k;
    if (row >= row_cap) return;
    const int h = idx[(size_t)b * N + k];
    if (h < 0 || h >= N) return;
    Hs = reinterpret_cast<float*>(reinterpret_cast<char*>(Hs) + b * wsStride);
    flags += b * wsStride;
    float hf[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) { hf[j] = Hp[(size_t)row * 9 + j]; Hs[(size_t)h * 9 + j] = hf[j]; }
    flags[h] = (flags[h] & ~2) | (rfx_det3_lu_f32(hf) > 1e-6f ? 2 : 0);
}

struct RansacWs {
    size_t P, Z, H, flags, counts, total;
};
inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }
inline RansacWs ws_layout(int N, int n_cap) {
    RansacWs L;
    size_t o = 0;
    L.P = o; o += al((size_t)n_cap * 16);
    L.Z = o; o += al((size_t)n_cap * 4);
    L.H = o; o += al((size_t)N * 36);
    L.flags = o; o += al((size_t)N);
    L.counts = o; o += al((size_t)N * 4);
    L.total = o;
    return L;
}

}  // namespace

extern "C" int rfx_dlt4_homography(const float* X, const float* Y, int N, float* Hout, void* stream) {
    if (!X || !Y || !Hout || N <= 0) return RFX_E_ARG;
    hipLaunchKernelGGL(ransac_dlt_kernel<false>, dim3((N + 63) / 64), dim3(64), 0, rfx_stream(stream), X, Y, 0,
                       (const int64_t*)nullptr, N, 0, Hout, (uint8_t*)nullptr, (const int32_t*)nullptr, 0, (size_t)0);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

extern "C" int rfx_prediction_f32(const float* match1, const float* match2, int n, const float* Hs, int N, float* err,
                                  void* stream) {
    if (!match1 || !match2 || !Hs || !err || n <= 0 || N <= 0 || N > 65535) return RFX_E_ARG;
    hipLaunchKernelGGL(prediction_kernel, dim3((n + 255) / 256, N), dim3(256), 0, rfx_stream(stream), match1, match2, n,
                       Hs, N, err);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

extern "C" size_t rfx_ransac_ws_bytes(int n, int N) {
    if (N <= 0 || n <= 0) return 0;
    return ws_layout(N, n).total;
}

// stages: 1 = pack + DLT (hypotheses and their flags into the workspace), 2 = count; 3 = both
static int ransac_common(const float* match1, const float* match2, int n, const int64_t* samples, int N, float tol,
                         int filter_dup, const RansacWs& L, char* w, float* Hs, int64_t* counts64, hipStream_t st,
                         const int32_t* narr = nullptr, int cap = 0, int batch = 1, int stages = 3, bool want_rank = false) {
    float4* P = reinterpret_cast<float4*>(w + L.P);
    float* Z = reinterpret_cast<float*>(w + L.Z);
    uint8_t* flags = reinterpret_cast<uint8_t*>(w + L.flags);
    int* counts = reinterpret_cast<int*>(w + L.counts);
    const int nmax = narr ? cap : n;
    if (stages & 1) {
        hipLaunchKernelGGL(ransac_pack_kernel, dim3((nmax + 255) / 256, batch), dim3(256), 0, st, match1, match2, n, P, Z, narr,
                           cap, L.total);
        RFX_LAUNCH_CHECK();
        if (want_rank)
            hipLaunchKernelGGL(ransac_dlt_kernel<true>, dim3((N + 63) / 64, batch), dim3(64), 0, st, match1, match2, n, samples, N,
                               filter_dup, Hs, flags, narr, cap, L.total);
        else
            hipLaunchKernelGGL(ransac_dlt_kernel<false>, dim3((N + 63) / 64, batch), dim3(64), 0, st, match1, match2, n, samples, N,
                               filter_dup, Hs, flags, narr, cap, L.total);
        RFX_LAUNCH_CHECK();
    }
    if (stages & 2) {
        hipLaunchKernelGGL(ransac_count_kernel, dim3((N + 3) / 4, batch), dim3(256), 0, st, P, Z, n, Hs, flags, N, tol, counts,
                           counts64, narr, L.total);
        RFX_LAUNCH_CHECK();
    }
    return RFX_OK;
}

extern "C" int rfx_score_hypotheses(const float* match1, const float* match2, int n, const int64_t* samples, int N,
                                    float tol, float* Hout, int64_t* counts, uint8_t* flags_out, void* ws, void* stream) {
    if (!match1 || !match2 || !samples || !Hout || !counts || !ws || n <= 0 || N <= 0) return RFX_E_ARG;
    const RansacWs L = ws_layout(N, n);
    hipStream_t st = rfx_stream(stream);
    int rc = ransac_common(match1, match2, n, samples, N, tol, 0, L, static_cast<char*>(ws), Hout, counts, st, nullptr, 0, 1, 3,
                           flags_out != nullptr);
    if (rc != RFX_OK || !flags_out) return rc;
    // the per-hypothesis flags of the SAME DLT pass (bit 2 = rank-deficient system): no second solve for the caller that patches
    hipError_t e = hipMemcpyAsync(flags_out, static_cast<char*>(ws) + L.flags, (size_t)N, hipMemcpyDeviceToDevice, st);
    return e == hipSuccess ? RFX_OK : (int)e;
}

extern "C" int rfx_ransac_h4(const float* match1, const float* match2, int n, const int64_t* samples, int N, float tol,
                             float* bestH, uint8_t* inlier, int32_t* result, void* ws, void* stream) {
    if (!match1 || !match2 || !samples || !bestH || !inlier || !result || !ws || n <= 0 || N <= 0) return RFX_E_ARG;
    if ((N + CHUNK - 1) / CHUNK > MAX_CHUNKS) return RFX_E_LIMIT;
    const RansacWs L = ws_layout(N, n);
    char* w = static_cast<char*>(ws);
    float* Hs = reinterpret_cast<float*>(w + L.H);
    hipStream_t st = rfx_stream(stream);
    int rc = ransac_common(match1, match2, n, samples, N, tol, 1, L, w, Hs, nullptr, st);
    if (rc != RFX_OK) return rc;
    hipLaunchKernelGGL(ransac_select_kernel, dim3(1), dim3(1024), 0, st, reinterpret_cast<const float4*>(w + L.P),
                       reinterpret_cast<const float*>(w + L.Z), n, Hs, reinterpret_cast<const uint8_t*>(w + L.flags),
                       reinterpret_cast<const int*>(w + L.counts), N, tol, bestH, inlier, result,
                       (const int32_t*)nullptr, 0, (size_t)0);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

// ---- batched RANSAC: every pair of a batch in ONE launch chain (pack, DLT, count, select), match counts read from a
// device array.  Same kernels, same arithmetic per pair as rfx_ransac_h4.
extern "C" size_t rfx_ransac_batched_ws_bytes(int cap, int N, int batch) {
    if (N <= 0 || cap <= 0 || batch <= 0) return 0;
    return ws_layout(N, cap).total * (size_t)batch;
}

static int ransac_batched_stages(const float* match1, const float* match2, const int32_t* n, int cap, const int64_t* samples,
                                 int N, float tol, float* bestH, uint8_t* inlier, int32_t* result, void* ws, int batch,
                                 int stages, void* stream) {
    if (!match1 || !match2 || !n || !samples || !bestH || !inlier || !result || !ws || cap <= 0 || N <= 0 || batch <= 0 ||
        stages < 1 || stages > 3)
        return RFX_E_ARG;
    if ((N + CHUNK - 1) / CHUNK > MAX_CHUNKS || batch > 65535) return RFX_E_LIMIT;
    const RansacWs L = ws_layout(N, cap);
    char* w = static_cast<char*>(ws);
    float* Hs = reinterpret_cast<float*>(w + L.H);
    hipStream_t st = rfx_stream(stream);
    // the rank flag is read between the stages of the staged ("lapack") search only: stages == 3 is the throughput path
    int rc = ransac_common(match1, match2, 0, samples, N, tol, 1, L, w, Hs, nullptr, st, n, cap, batch, stages, stages == 1);
    if (rc != RFX_OK) return rc;
    if (stages & 2) {
        hipLaunchKernelGGL(ransac_select_kernel, dim3(batch), dim3(1024), 0, st, reinterpret_cast<const float4*>(w + L.P),
                           reinterpret_cast<const float*>(w + L.Z), 0, Hs, reinterpret_cast<const uint8_t*>(w + L.flags),
                           reinterpret_cast<const int*>(w + L.counts), N, tol, bestH, inlier, result, n, cap, L.total);
        RFX_LAUNCH_CHECK();
    }
    return RFX_OK;
}

extern "C" int rfx_ransac_h4_batched(const float* match1, const float* match2, const int32_t* n, int cap,
                                     const int64_t* samples, int N, float tol, float* bestH, uint8_t* inlier,
                                     int32_t* result, void* ws, int batch, void* stream) {
    return ransac_batched_stages(match1, match2, n, cap, samples, N, tol, bestH, inlier, result, ws, batch, 3, stream);
}

extern "C" int rfx_ransac_h4_batched_stage(const float* match1, const float* match2, const int32_t* n, int cap,
                                           const int64_t* samples, int N, float tol, float* bestH, uint8_t* inlier,
                                           int32_t* result, void* ws, int batch, int stages, void* stream) {
    return ransac_batched_stages(match1, match2, n, cap, samples, N, tol, bestH, inlier, result, ws, batch, stages, stream);
}

extern "C" int rfx_ransac_degenerate_list(const void* ws, int cap, int N, int batch, int32_t* idx, int32_t* count, int kcap,
                                          void* stream) {
    if (!ws || !idx || !count || cap <= 0 || N <= 0 || batch <= 0 || kcap <= 0) return RFX_E_ARG;
    if (batch > 65535) return RFX_E_LIMIT;
    const RansacWs L = ws_layout(N, cap);
    hipLaunchKernelGGL(ransac_degen_list_kernel, dim3(batch), dim3(1024), 0, rfx_stream(stream),
                       reinterpret_cast<const uint8_t*>(static_cast<const char*>(ws) + L.flags), N, idx, count, kcap, L.total);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

extern "C" int rfx_ransac_patch_h(void* ws, int cap, int N, int batch, const int32_t* idx, const int32_t* count,
                                  const float* Hpatch, int kcap, void* stream) {
    if (!ws || !idx || !count || !Hpatch || cap <= 0 || N <= 0 || batch <= 0 || kcap <= 0) return RFX_E_ARG;
    if (batch > 65535) return RFX_E_LIMIT;
    const RansacWs L = ws_layout(N, cap);
    char* w = static_cast<char*>(ws);
    hipLaunchKernelGGL(ransac_patch_kernel, dim3((kcap + 255) / 256, batch), dim3(256), 0, rfx_stream(stream),
                       reinterpret_cast<float*>(w + L.H), reinterpret_cast<uint8_t*>(w + L.flags), idx, count, Hpatch, kcap, N,
                       L.total);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

extern "C" int rfx_ransac_degenerate_gather(const void* ws, const float* match1, const float* match2, const int32_t* n, int cap,
                                            const int64_t* samples, int N, int batch, int32_t* idx, int32_t* count, int32_t* off,
                                            int32_t* off_host, float* xy_rows, int row_cap, void* stream) {
    if (!ws || !match1 || !match2 || !n || !samples || !idx || !count || !off || !xy_rows || cap <= 0 || N <= 0 || batch <= 0 ||
        row_cap <= 0)
        return RFX_E_ARG;
    if (batch > 65535) return RFX_E_LIMIT;
    const RansacWs L = ws_layout(N, cap);
    hipStream_t st = rfx_stream(stream);
    hipLaunchKernelGGL(ransac_degen_list_kernel, dim3(batch), dim3(1024), 0, st,
                       reinterpret_cast<const uint8_t*>(static_cast<const char*>(ws) + L.flags), N, idx, count, N, L.total);
    RFX_LAUNCH_CHECK();
    hipLaunchKernelGGL(ransac_degen_scan_kernel, dim3(1), dim3(64), 0, st, count, batch, off, off_host);
    RFX_LAUNCH_CHECK();
    hipLaunchKernelGGL(ransac_degen_gather_kernel, dim3((N + 255) / 256, batch), dim3(256), 0, st, match1, match2, n, cap, samples, N,
                       idx, count, off, reinterpret_cast<float4*>(xy_rows), row_cap);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

extern "C" int rfx_ransac_patch_h_rows(void* ws, int cap, int N, int batch, const int32_t* idx, const int32_t* count,
                                       const int32_t* off, const float* H_rows, int row_cap, void* stream) {
    if (!ws || !idx || !count || !off || !H_rows || cap <= 0 || N <= 0 || batch <= 0 || row_cap <= 0) return RFX_E_ARG;
    if (batch > 65535) return RFX_E_LIMIT;
    const RansacWs L = ws_layout(N, cap);
    char* w = static_cast<char*>(ws);
    hipLaunchKernelGGL(ransac_patch_compact_kernel, dim3((N + 255) / 256, batch), dim3(256), 0, rfx_stream(stream),
                       reinterpret_cast<float*>(w + L.H), reinterpret_cast<uint8_t*>(w + L.flags), idx, count, off, H_rows, row_cap, N,
                       L.total);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

// outil.Homography with the rank flag per system (degenerate (N) uint8: 1 = rank deficient, see dlt.h)
extern "C" int rfx_dlt4_homography_flags(const float* X, const float* Y, int N, float* Hout, uint8_t* degenerate, void* stream) {
    if (!X || !Y || !Hout || !degenerate || N <= 0) return RFX_E_ARG;
    hipLaunchKernelGGL(ransac_dlt_kernel<true>, dim3((N + 63) / 64), dim3(64), 0, rfx_stream(stream), X, Y, 0,
                       (const int64_t*)nullptr, N, 0, Hout, degenerate, (const int32_t*)nullptr, 0, (size_t)0);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

// match1[b,i] = (xa[idx1[b,i]], ya[idx1[b,i]], 1), match2[b,i] = (xb[idx2[b,i]], yb[idx2[b,i]], 1) for i < n[b], zeros after:
// the match lists of quick_start/coarseAlignFeatMatch.py:150-155 for a whole batch in one launch.
namespace {
__global__ __launch_bounds__(256) void gather_matches_kernel(const int64_t* __restrict__ idx1, const int64_t* __restrict__ idx2,
                                                             const int32_t* __restrict__ narr, int cap,
                                                             const float* __restrict__ xa, const float* __restrict__ ya,
                                                             const float* __restrict__ xb, const float* __restrict__ yb,
                                                             float* __restrict__ m1, float* __restrict__ m2) {
    const int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap) return;
    const size_t o = ((size_t)b * cap + i) * 3;
    if (i < narr[b]) {
        const int64_t a = idx1[(size_t)b * cap + i], t = idx2[(size_t)b * cap + i];
        m1[o] = xa[a]; m1[o + 1] = ya[a]; m1[o + 2] = 1.0f;
        m2[o] = xb[t]; m2[o + 1] = yb[t]; m2[o + 2] = 1.0f;
    } else {
        m1[o] = m1[o + 1] = m1[o + 2] = 0.0f;
        m2[o] = m2[o + 1] = m2[o + 2] = 0.0f;
    }
}
}  // namespace

extern "C" int rfx_gather_matches_f32(const int64_t* idx1, const int64_t* idx2, const int32_t* n, int cap, const float* xa,
                                      const float* ya, const float* xb, const float* yb, float* match1, float* match2,
                                      int batch, void* stream) {
    if (!idx1 || !idx2 || !n || !xa || !ya || !xb || !yb || !match1 || !match2 || cap <= 0 || batch <= 0) return RFX_E_ARG;
    if (batch > 65535) return RFX_E_LIMIT;
    hipLaunchKernelGGL(gather_matches_kernel, dim3((cap + 255) / 256, batch), dim3(256), 0, rfx_stream(stream), idx1, idx2, n,
                       cap, xa, ya, xb, yb, match1, match2);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}
