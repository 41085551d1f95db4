// api.hip -- library identification and the grouped-launch recorder (group.h).
#include "common.h"
#include "group.h"
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

extern "C" const char* rfx_version(void) { return "rfx 0.6.0 gfx950"; }
extern "C" int rfx_abi_version(void) { return RFX_ABI_VERSION; }

namespace {
struct Pending {
    rfx_group_launch_fn fn;
    size_t arg_size;
    std::vector<char> blob;        // n argument blocks back to back
    std::vector<unsigned> gx;
};
struct Recorder {
    bool on = false;
    std::vector<Pending> buckets;  // one per kernel instance, in first-seen order
};
thread_local Recorder g_rec;

// The kernel instances of one group are independent: the second and third run on side streams forked from / joined to the
// caller's stream with events (also inside a HIP-graph capture, where this becomes a fork/join of the captured graph).
constexpr int NSIDE = 2;
struct SidePool {
    int dev = -1;
    hipStream_t owner = nullptr;            // the caller's stream this pool forks from
    hipStream_t s[NSIDE];
    hipEvent_t fork, join[NSIDE];
};
// One pool per (device, caller stream): two chains of grouped launches on two caller streams (rfx/pipeline.py) must not share
// side streams -- the second chain's side work would queue behind the first's -- nor re-record each other's fork / join events.
constexpr int NPOOL = 4;
thread_local SidePool g_pools[NPOOL];
thread_local bool g_side_on = true;      // rfx_group_side_streams

SidePool* side_pool_for(hipStream_t main) {
    static const bool on = !(getenv("RFX_GROUP_STREAMS") && atoi(getenv("RFX_GROUP_STREAMS")) == 0);
    if (!on || !g_side_on) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    SidePool* free_slot = nullptr;
    for (auto& p : g_pools) {
        if (p.dev == dev && p.owner == main) return &p;
        if (p.dev < 0 && !free_slot) free_slot = &p;
    }
    if (!free_slot) return nullptr;                          // more caller streams than pools: those run the serial form
    SidePool& p = *free_slot;
    for (int i = 0; i < NSIDE; ++i) {
        if (hipStreamCreateWithFlags(&p.s[i], hipStreamNonBlocking) != hipSuccess) return nullptr;
        if (hipEventCreateWithFlags(&p.join[i], hipEventDisableTiming) != hipSuccess) return nullptr;
    }
    if (hipEventCreateWithFlags(&p.fork, hipEventDisableTiming) != hipSuccess) return nullptr;
    p.dev = dev;
    p.owner = main;
    return &p;
}
}  // namespace

bool rfx_group_recording() { return g_rec.on; }

int rfx_group_record(rfx_group_launch_fn fn, const void* args, size_t arg_size, unsigned grid_x) {
    for (auto& b : g_rec.buckets)
        if (b.fn == fn && b.arg_size == arg_size) {
            b.blob.insert(b.blob.end(), static_cast<const char*>(args), static_cast<const char*>(args) + arg_size);
            b.gx.push_back(grid_x);
            return RFX_OK;
        }
    Pending p;
    p.fn = fn; p.arg_size = arg_size;
    p.blob.assign(static_cast<const char*>(args), static_cast<const char*>(args) + arg_size);
    p.gx.push_back(grid_x);
    g_rec.buckets.push_back(std::move(p));
    return RFX_OK;
}

extern "C" int rfx_group_begin(void) {
    if (g_rec.on) return RFX_E_ARG;          // groups do not nest
    g_rec.on = true;
    g_rec.buckets.clear();
    return RFX_OK;
}

extern "C" int rfx_group_end(void* stream) {
    if (!g_rec.on) return RFX_E_ARG;
    g_rec.on = false;
    int rc = RFX_OK;
    static const bool dbg = getenv("RFX_GROUP_DEBUG") != nullptr;
    if (dbg) {
        fprintf(stderr, "[rfx group] %zu kernel instance(s):", g_rec.buckets.size());
        for (auto& b : g_rec.buckets) {
            unsigned tot = 0, mx = 0;
            for (unsigned g : b.gx) { tot += g; if (g > mx) mx = g; }
            fprintf(stderr, " %zux(wg %u, max %u)", b.gx.size(), tot, mx);
        }
        fprintf(stderr, "\n");
    }
    hipStream_t main = rfx_stream(stream);
    SidePool* pool = g_rec.buckets.size() > 1 ? side_pool_for(main) : nullptr;
    const bool fork = pool != nullptr;
    if (fork && hipEventRecord(pool->fork, main) != hipSuccess) rc = RFX_E_ARG;
    bool used[NSIDE] = {false, false};
    int bi = 0;
    for (auto& b : g_rec.buckets) {
        hipStream_t st = main;
        if (fork && bi > 0) {
            const int k = (bi - 1) % NSIDE;
            st = pool->s[k];
            if (!used[k]) {
                if (hipStreamWaitEvent(st, pool->fork, 0) != hipSuccess) rc = RFX_E_ARG;
                used[k] = true;
            }
        }
        const int n = (int)b.gx.size();
        for (int i = 0; i < n && rc == RFX_OK; i += RFX_MAX_GROUP) {
            const int m = n - i < RFX_MAX_GROUP ? n - i : RFX_MAX_GROUP;
            rc = b.fn(b.blob.data() + (size_t)i * b.arg_size, b.gx.data() + i, m, st);
        }
        if (rc != RFX_OK) break;
        ++bi;
    }
    for (int k = 0; k < NSIDE; ++k)                         // join: always, also after an error (a capture must not end forked)
        if (used[k]) {
            if (hipEventRecord(pool->join[k], pool->s[k]) != hipSuccess || hipStreamWaitEvent(main, pool->join[k], 0) != hipSuccess)
                if (rc == RFX_OK) rc = RFX_E_ARG;
        }
    g_rec.buckets.clear();
    return rc;
}

extern "C" int rfx_group_abort(void) {       // drop a recording without launching (error paths of the host mirrors)
    g_rec.on = false;
    g_rec.buckets.clear();
    return RFX_OK;
}

extern "C" int rfx_group_side_streams(int enable) {
    const int prev = g_side_on ? 1 : 0;
    g_side_on = enable != 0;
    return prev;
}
