// group.h -- grouped launches: ONE launch for the same layer of several independent problems of different sizes.
//
// A single 480x640 pair runs the ResNet-50 trunk on 8 images of 8 different sizes (7 pyramid levels + the target,
// quick_start/coarseAlignFeatMatch.py:92-125).  Layer by layer that is 8 launches of 2-150 workgroups each on 256 CUs: the
// pass is bound by one workgroup lifetime per layer AND level.  Between rfx_group_begin() and rfx_group_end() the
// convolution entry points do not launch; they record (kernel instance, argument block, grid), and rfx_group_end() issues,
// per kernel instance, ONE launch whose blockIdx.y selects the problem (argument block p[blockIdx.y], workgroups past that
// problem's own grid exit at once).  The device code of a problem is the single-launch kernel's body, unchanged: results are
// bit-identical by construction.  Everything recorded in one group must be mutually independent.
#pragma once
#include "common.h"

constexpr int RFX_MAX_GROUP = 8;

template <class A> struct RfxGroupArgs {
    A p[RFX_MAX_GROUP];
    unsigned gx[RFX_MAX_GROUP];   // workgroups of problem i (its own gridDim.x)
};

// launcher of one grouped kernel instance: `blob` = n argument blocks of `arg_size` bytes, gx = their grids
typedef int (*rfx_group_launch_fn)(const void* blob, const unsigned* gx, int n, hipStream_t st);

bool rfx_group_recording();
int rfx_group_record(rfx_group_launch_fn fn, const void* args, size_t arg_size, unsigned grid_x);

// Generic launcher body for a grouped kernel K(RfxGroupArgs<A>) with BLOCK threads.
template <class A, class K>
static int rfx_group_launch_impl(K kernel, int block, const void* blob, const unsigned* gx, int n, hipStream_t st) {
    RfxGroupArgs<A> g;
    unsigned gmax = 0;
    const A* src = static_cast<const A*>(blob);
    for (int i = 0; i < RFX_MAX_GROUP; ++i) {
        g.p[i] = src[i < n ? i : 0];
        g.gx[i] = i < n ? gx[i] : 0u;
        if (g.gx[i] > gmax) gmax = g.gx[i];
    }
    hipLaunchKernelGGL(kernel, dim3(gmax, (unsigned)n), dim3(block), 0, st, g);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}
