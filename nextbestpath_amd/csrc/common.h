// common.h -- shared helpers for the gfx950 kernels of libnbp_hip.so (MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "nbp_hip.h"

#define NBP_ABI_VERSION 1

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int nbp_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// hipGetLastError() is sticky per thread: a benign failure inside the caller's own runtime use (e.g. PyTorch probing a
// pointer) would otherwise be reported by the first launch-status check of this library.  Every entry point clears it.
#define NBP_ENTER() (void)hipGetLastError()

#define NBP_RETURN_IF(cond, code) \
    do {                          \
        if (cond) return (code);  \
    } while (0)

// A/B switch of a launch plan: `dflt` unless the process opted in with NBP_TUNING=1 AND sets the variable (nbp_tuning.cpp: the
// one place the library reads the environment; read once, recorded for nbp_tuning_report).  `name` must be a string literal.
int nbp_tune_int(const char* name, int dflt);
// the kernel symbol a convolution tile id was launched as (nbp_tuning.cpp; read back through nbp_tile_kernel_symbol)
void nbp_note_kernel_symbol(int tile, const char* symbol);

static inline long long nbp_cdiv(long long a, long long b) { return (a + b - 1) / b; }

// grid size for grid-stride elementwise kernels: enough blocks to fill 256 CUs x 8.
static inline int nbp_ew_grid(long long work_items, int block) {
    long long g = nbp_cdiv(work_items, block);
    if (g < 1) g = 1;
    if (g > 256 * 8) g = 256 * 8;
    return (int)g;
}

// ------------------------------------------------------------------ index bijection
// perm_index(j) for j in [0,n) enumerates [0,n) in a pseudo-random order: a bijection on
// b-bit integers (multiply by an odd constant, add, xor-shift; 3 rounds) cycle-walked into
// [0,n).  "The first k of a random permutation" = {perm_index(j) : j < k}: an exact-size
// random subset without a sort (stands in for torch.randperm(n)[:k] of the reference).
// oracle/sampling.py restates it bit for bit.
__host__ __device__ inline unsigned perm_bits(unsigned n) {
    unsigned b = 2;
    while (b < 32 && (1ull << b) < (unsigned long long)n) ++b;
    return b;
}
__host__ __device__ inline unsigned perm_index(unsigned j, unsigned n, unsigned b, unsigned seed) {
    const unsigned mask = b >= 32 ? 0xffffffffu : ((1u << b) - 1u);
    const unsigned sh = (b + 1) >> 1;
    unsigned v = j;
    do {
#pragma unroll
        for (unsigned r = 0; r < 3; ++r) {
            v = (v * 0x9E3779B1u + seed + r * 0x7F4A7C15u) & mask;
            v ^= v >> sh;
        }
    } while (v >= n);
    return v;
}
