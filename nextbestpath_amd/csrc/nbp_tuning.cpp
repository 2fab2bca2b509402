// nbp_tuning.cpp -- the ONE place libnbp_hip.so reads the process environment.
//
// Every A/B switch of the kernels' launch plans (tile choice, split-K bounds, fusions) is a compile-time default that can be
// overridden only when the caller opts in with NBP_TUNING=1: an inherited environment must not change results
// (no global mutable state except the handle).  Each knob is read once per process and recorded with its
// default and effective value; nbp_tuning_report lists the ones that differ, so a benchmark line can state them.
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <mutex>
#include "nbp_hip.h"

namespace {
struct Knob { const char* name; int dflt; int value; };
Knob g_knobs[128];
int g_n = 0;
std::mutex g_mu;
bool tuning_on() {
    static const bool on = [] { const char* e = getenv("NBP_TUNING"); return e && atoi(e) == 1; }();
    return on;
}
}  // namespace

// `name` must be a string literal (the registry keeps the pointer)
int nbp_tune_int(const char* name, int dflt) {
    int v = dflt;
    const char* e = getenv(name);
    const bool set = e && *e;
    if (set && tuning_on()) v = atoi(e);
    std::lock_guard<std::mutex> lk(g_mu);
    for (int i = 0; i < g_n; ++i)
        if (!strcmp(g_knobs[i].name, name)) return g_knobs[i].dflt == dflt ? g_knobs[i].value : v;
    // an A/B script that exports a switch without the opt-in would silently measure the default twice: say so, once
    if (set && !tuning_on()) fprintf(stderr, "[nbp] %s=%s is ignored: set NBP_TUNING=1 to unlock the A/B switches\n", name, e);
    if (g_n < 128) g_knobs[g_n++] = Knob{name, dflt, v};
    else fprintf(stderr, "[nbp] tuning registry full: %s is not listed in nbp_tuning_report\n", name);
    return v;
}

// ---- which kernel symbol a convolution tile id (nbp_layer_timing.tile) was last launched as.  The launchers record the name they
// build from their own template arguments (nbp_split.hip: launch_h2; nbp_conv.hip; nbp_bf16.hip), so a profile reader
// (bench.py: rocprofv3 counter rows by kernel name) asks the library instead of keeping a table of its own that rots when a
// kernel grows a template parameter.
namespace {
struct Sym { char name[96]; long long launches; };
Sym g_sym[64][4];      // a tile id may stand for several instantiations (16- / 32-pixel-wide tiles): the most launched one is reported
}
void nbp_note_kernel_symbol(int tile, const char* symbol) {
    if (tile < 0 || tile >= 64 || !symbol) return;
    std::lock_guard<std::mutex> lk(g_mu);
    for (int k = 0; k < 4; ++k) {
        Sym& e = g_sym[tile][k];
        if (!e.name[0]) { strncpy(e.name, symbol, sizeof(e.name) - 1); e.launches = 1; return; }
        if (!strcmp(e.name, symbol)) { ++e.launches; return; }
    }
}
// returns the length written (0: no launch of that tile id in this process yet)
extern "C" int nbp_tile_kernel_symbol(int tile, char* buf_host, int len) {
    if (!buf_host || len < 1) return 0;
    buf_host[0] = 0;
    if (tile < 0 || tile >= 64) return 0;
    std::lock_guard<std::mutex> lk(g_mu);
    const Sym* best = nullptr;
    for (int k = 0; k < 4; ++k)
        if (g_sym[tile][k].name[0] && (!best || g_sym[tile][k].launches > best->launches)) best = &g_sym[tile][k];
    if (!best) return 0;
    strncpy(buf_host, best->name, (size_t)len - 1);
    buf_host[len - 1] = 0;
    return (int)strlen(buf_host);
}

extern "C" int nbp_tuning_active(void) { return tuning_on() ? 1 : 0; }

// "NAME=value,NAME=value" of the knobs read so far whose value differs from the default; returns how many (the text is truncated
// to len - 1 characters)
extern "C" int nbp_tuning_report(char* buf_host, int len) {
    std::lock_guard<std::mutex> lk(g_mu);
    int n = 0;
    size_t off = 0;
    if (buf_host && len > 0) buf_host[0] = 0;
    for (int i = 0; i < g_n; ++i) {
        if (g_knobs[i].value == g_knobs[i].dflt) continue;
        ++n;
        if (buf_host && len > 0 && off + 1 < (size_t)len) {
            const int w = snprintf(buf_host + off, (size_t)len - off, "%s%s=%d", off ? "," : "", g_knobs[i].name, g_knobs[i].value);
            if (w > 0) off += (size_t)w < (size_t)len - off ? (size_t)w : (size_t)len - off - 1;
        }
    }
    return n;
}
