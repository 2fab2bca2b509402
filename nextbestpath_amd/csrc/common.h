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

#define NBP_RETURN_IF(cond, code) \
    do {                          \
        if (cond) return (code);  \
    } while (0)

static inline long long nbp_cdiv(long long a, long long b) { return (a + b - 1) / b; }

// grid size for grid-stride elementwise kernels: enough blocks to fill 256 CUs x 8.
static inline int nbp_ew_grid(long long work_items, int block) {
    long long g = nbp_cdiv(work_items, block);
    if (g < 1) g = 1;
    if (g > 256 * 8) g = 256 * 8;
    return (int)g;
}
