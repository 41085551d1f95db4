// cc.hip -- the small-component filter of the KITTI scripts on the device: zero every 8-connected component of
// (match > match_th) whose area is at most max_area pixels (evaluation/evalKITTI/evaluation.py:85-100 online,
// evaluation/evalKITTI/getResults.py:66-84 offline; both call skimage.measure.label on the host and loop over the
// component ids).  The result does not depend on how components are numbered, so any exact labelling reproduces the
// reference bit for bit: here a lock-free union-find over the pixel grid.
//   1. background pixels get parent -1; foreground pixels are linked along their row (no atomics, see cc_init_kernel);
//   2. unions with the row above, only the non-redundant ones: roots are linked larger -> smaller index with atomicCAS,
//      finds use path halving (see cc_union_kernel);
//   3. label[p] = find(p);  4. area[label] += 1;  5. out = (foreground && area[label] <= max_area && area[label] < H*W) ? 0 : in.
// Steps are separate launches (the grid-wide ordering between them is the stream's).  One int32 triple per pixel of
// workspace.  The area threshold arrives as an integer (the host turns the reference's float64 test
// "area / size <= cc_th" into the largest pixel count that passes it), so no floating point enters the decision.
#include "common.h"

namespace {

// parent[] is read and written by every CU while the forest is being built: device-scope atomic loads / stores keep the
// accesses out of the (non-coherent) per-CU L1, so a find never spins on a stale line.
__device__ __forceinline__ int cc_ld(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void cc_st(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ int cc_find(int* parent, int i) {
    while (true) {
        const int p = cc_ld(parent + i);
        if (p == i) return i;
        const int gp = cc_ld(parent + p);
        if (gp != p) cc_st(parent + i, gp);   // path halving; a late write only keeps a longer (still valid) chain
        i = p;
    }
}

__device__ __forceinline__ void cc_unite(int* parent, int a, int b) {
    while (true) {
        a = cc_find(parent, a);
        b = cc_find(parent, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }          // link the larger root under the smaller
        if (atomicCAS(&parent[a], a, b) == a) return;           // a was still a root: linked; otherwise retry
    }
}

// 1. A wavefront takes 64 consecutive pixels.  Every foreground pixel is linked -- without atomics, nobody else writes
// parent[] yet -- to the first pixel of its horizontal run inside this 64-pixel segment; a run that continues from the
// previous segment hangs its first pixel here on the pixel to its left.  The row direction of the labelling is thereby
// done before the union phase starts: a solid region costs no union at all (see the rules below), and a find walks at most
// one link per 64 pixels of a run.
__global__ __launch_bounds__(256) void cc_init_kernel(const float* __restrict__ in, int* __restrict__ parent, int* __restrict__ area,
                                                     long long total, int W, float th) {
    const int lane = threadIdx.x & 63;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long base = (long long)blockIdx.x * blockDim.x + (threadIdx.x & ~63); base < total; base += stride) {
        const long long i = base + lane;
        const bool ok = i < total;
        const bool fg = ok && in[i] > th;
        const int x = ok ? (int)(i % W) : 0;
        const unsigned long long m = __ballot(fg);
        // left neighbour foreground and in the same row?  (lane 0 looks into the previous segment)
        const bool left = fg && x > 0 && (lane > 0 ? ((m >> (lane - 1)) & 1ull) != 0 : in[i - 1] > th);
        const unsigned long long starts = __ballot(fg && !left) | (m & 1ull);   // run starts inside the segment (+ lane 0)
        if (ok) {
            int par = -1;
            if (fg) {
                const unsigned long long below = starts & (~0ull >> (63 - lane));    // start bits at or below this lane
                const int s = 63 - __clzll((long long)below);                          // below != 0: lane 0 is always a start bit
                par = (lane == 0 && left) ? (int)(i - 1) : (int)(base + s);
            }
            parent[i] = par;
            area[i] = 0;
        }
    }
}

// 2. Unions across rows, only where they can join something new.  With the row links in place, for a foreground pixel p
// (W, NW, N, NE, E = its neighbours):
//   N  is redundant when W and NW are foreground (p - W - NW - N is already a path);
//   NW is redundant when W or N is foreground (W's own N neighbour is NW; N and NW are row neighbours);
//   NE is redundant when N is foreground (row neighbours) or E is foreground (E's N neighbour is NE).
// An interior pixel of a solid region issues nothing; the atomics that remain are the ones at region boundaries.
__global__ __launch_bounds__(256) void cc_union_kernel(int* __restrict__ parent, long long total, int H, int W) {
    const long long HW = (long long)H * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        if (cc_ld(parent + i) < 0) continue;
        const long long p = i % HW;
        const int y = (int)(p / W), x = (int)(p - (long long)y * W);
        if (y == 0) continue;
        const bool w = x > 0 && cc_ld(parent + i - 1) >= 0;
        const bool e = x + 1 < W && cc_ld(parent + i + 1) >= 0;
        const bool n = cc_ld(parent + i - W) >= 0;
        const bool nw = x > 0 && cc_ld(parent + i - W - 1) >= 0;
        const bool ne = x + 1 < W && cc_ld(parent + i - W + 1) >= 0;
        if (n && !(w && nw)) cc_unite(parent, (int)i, (int)(i - W));
        if (nw && !w && !n) cc_unite(parent, (int)i, (int)(i - W - 1));
        if (ne && !n && !e) cc_unite(parent, (int)i, (int)(i - W + 1));
    }
}

// 3./4. label = root; area[root] += 1.  The pixels of one wavefront mostly share their root (a matched region is one large
// component), and a per-pixel atomicAdd would queue every pixel of that component on ONE address: the lanes that hold the
// same root are counted with a ballot and their leader adds the count -- one atomic per (wavefront, root).
__global__ __launch_bounds__(256) void cc_label_kernel(int* __restrict__ parent, int* __restrict__ label, int* __restrict__ area,
                                                      long long total) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long base = (long long)blockIdx.x * blockDim.x + (threadIdx.x & ~63); base < total; base += stride) {
        const long long i = base + (threadIdx.x & 63);
        int r = -1;
        if (i < total) {
            if (cc_ld(parent + i) >= 0) r = cc_find(parent, (int)i);
            label[i] = r;
        }
        unsigned long long todo = __ballot(r >= 0);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int lr = __shfl(r, leader);
            const unsigned long long same = __ballot(r == lr) & todo;
            if ((int)(threadIdx.x & 63) == leader) atomicAdd(&area[lr], __popcll(same));
            todo &= ~same;
        }
    }
}

__global__ __launch_bounds__(256) void cc_apply_kernel(const float* __restrict__ in, const int* __restrict__ label,
                                                      const int* __restrict__ area, float* __restrict__ out, long long total,
                                                      int max_area, int HW) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int l = label[i];
        // a component that IS the whole image is never removed: the reference walks np.unique(label)[1:], taking the first id
        // for the background, and returns early when there is only one id (evaluation/evalKITTI/evaluation.py:90-93)
        const int a = l >= 0 ? area[l] : 0;
        out[i] = (l >= 0 && a <= max_area && a < HW) ? 0.0f : in[i];
    }
}

}  // namespace

extern "C" size_t rfx_remove_small_cc_ws_bytes(int N, int H, int W) { return (size_t)3 * sizeof(int) * (size_t)N * H * W; }

extern "C" int rfx_remove_small_cc_f32(const float* in, float* out, int N, int H, int W, float match_th, int max_area, void* ws,
                                       void* stream) {
    if (!in || !out || !ws || N <= 0 || H <= 0 || W <= 0 || max_area < 0) return RFX_E_ARG;
    const long long total = (long long)N * H * W;
    if (total > 0x7fffffffLL) return RFX_E_LIMIT;
    int* parent = static_cast<int*>(ws);
    int* label = parent + total;
    int* area = label + total;
    hipStream_t st = rfx_stream(stream);
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(cc_init_kernel, dim3(grid), dim3(256), 0, st, in, parent, area, total, W, match_th);
    hipLaunchKernelGGL(cc_union_kernel, dim3(grid), dim3(256), 0, st, parent, total, H, W);
    hipLaunchKernelGGL(cc_label_kernel, dim3(grid), dim3(256), 0, st, parent, label, area, total);
    hipLaunchKernelGGL(cc_apply_kernel, dim3(grid), dim3(256), 0, st, in, label, area, out, total, max_area, H * W);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}
