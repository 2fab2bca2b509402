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
    if (tuning_on()) {
        const char* e = getenv(name);
        if (e && *e) v = atoi(e);
    }
    std::lock_guard<std::mutex> lk(g_mu);
    for (int i = 0; i < g_n; ++i)
        if (!strcmp(g_knobs[i].name, name)) return g_knobs[i].dflt == dflt ? g_knobs[i].value : v;
    if (g_n < 128) g_knobs[g_n++] = Knob{name, dflt, v};
    return v;
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
