// api.hip -- library identification and the grouped-launch recorder (group.h).
#include "common.h"
#include "group.h"
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

extern "C" const char* rfx_version(void) { return "rfx 0.3.0 gfx950"; }
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
    for (auto& b : g_rec.buckets) {
        const int n = (int)b.gx.size();
        for (int i = 0; i < n && rc == RFX_OK; i += RFX_MAX_GROUP) {
            const int m = n - i < RFX_MAX_GROUP ? n - i : RFX_MAX_GROUP;
            rc = b.fn(b.blob.data() + (size_t)i * b.arg_size, b.gx.data() + i, m, rfx_stream(stream));
        }
        if (rc != RFX_OK) break;
    }
    g_rec.buckets.clear();
    return rc;
}

extern "C" int rfx_group_abort(void) {       // drop a recording without launching (error paths of the host mirrors)
    g_rec.on = false;
    g_rec.buckets.clear();
    return RFX_OK;
}
