// nbp_grid.h -- uniform point grid shared by the coverage metric (nbp_planner.hip) and the scene store
// (nbp_scene.hip): cell lookup and the chip-wide exclusive scan of per-cell counts.  Included inside each
// translation unit (everything is in an anonymous namespace); not part of the C ABI.
#pragma once
#include "common.h"

namespace {

struct Grid { float lo[3]; float inv; int n[3]; };

__device__ __forceinline__ int grid_cell(const Grid& g, float x, float y, float z, int* ijk) {
    int i = (int)floorf((x - g.lo[0]) * g.inv), j = (int)floorf((y - g.lo[1]) * g.inv), k = (int)floorf((z - g.lo[2]) * g.inv);
    i = min(max(i, 0), g.n[0] - 1); j = min(max(j, 0), g.n[1] - 1); k = min(max(k, 0), g.n[2] - 1);
    if (ijk) { ijk[0] = i; ijk[1] = j; ijk[2] = k; }
    return (i * g.n[1] + j) * g.n[2] + k;
}

// K2: exclusive scan of count[0..ncell) -> start[0..ncell] in three launches that use the whole chip (a single
// block walking the array tile by tile pays one global-memory round trip per tile: 34 us for 140 k cells):
// (a) per-block sums of 4096-int tiles, (b) one block scans the <= 4096 tile sums, (c) per-tile scan + offset.
constexpr int SCAN_TILE = 4096;
__device__ __forceinline__ int block_exclusive_scan_256(int mine, int* wtot /*[4]*/, int* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int n = __shfl_up(inc, o);
        if (lane >= o) inc += n;
    }
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += wtot[w];
    if (total) *total = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    return base + inc - mine;
}

__global__ __launch_bounds__(256) void grid_tilesum_kernel(const int* __restrict__ count, long long ncell,
                                                               int* __restrict__ tsum) {
    __shared__ int wtot[4];
    const long long i0 = (long long)blockIdx.x * SCAN_TILE + 16 * (long long)threadIdx.x;
    int mine = 0;
#pragma unroll
    for (int e = 0; e < 16; ++e) mine += i0 + e < ncell ? count[i0 + e] : 0;
    int total;
    (void)block_exclusive_scan_256(mine, wtot, &total);
    if (threadIdx.x == 0) tsum[blockIdx.x] = total;
}

__global__ __launch_bounds__(256) void grid_tilescan_kernel(int* __restrict__ tsum, int ntiles, int* __restrict__ start,
                                                                long long ncell) {
    __shared__ int wtot[4];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int t0 = 0; t0 < ntiles; t0 += 256 * 16) {           // one pass for up to 4096 tiles (16 M cells)
        int v[16], mine = 0;
        const int i0 = t0 + 16 * threadIdx.x;
#pragma unroll
        for (int e = 0; e < 16; ++e) { v[e] = i0 + e < ntiles ? tsum[i0 + e] : 0; mine += v[e]; }
        int total;
        int run = carry + block_exclusive_scan_256(mine, wtot, &total);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            if (i0 + e < ntiles) tsum[i0 + e] = run;
            run += v[e];
        }
        __syncthreads();
        if (threadIdx.x == 0) carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) start[ncell] = carry;
}

__global__ __launch_bounds__(256) void grid_scan_kernel(const int* __restrict__ count, long long ncell,
                                                            const int* __restrict__ toff, int* __restrict__ start) {
    __shared__ int wtot[4];
    const long long i0 = (long long)blockIdx.x * SCAN_TILE + 16 * (long long)threadIdx.x;
    int v[16], mine = 0;
#pragma unroll
    for (int e = 0; e < 16; ++e) { v[e] = i0 + e < ncell ? count[i0 + e] : 0; mine += v[e]; }
    int run = toff[blockIdx.x] + block_exclusive_scan_256(mine, wtot, nullptr);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        if (i0 + e < ncell) start[i0 + e] = run;
        run += v[e];
    }
}


// start[0..ncell] = exclusive scan of count[0..ncell); tsum needs ncell / SCAN_TILE + 1 ints.
inline int grid_exclusive_scan(const int* count, long long ncell, int* tsum, int* start, hipStream_t st) {
    const int ntiles = (int)(ncell / SCAN_TILE + 1);
    grid_tilesum_kernel<<<ntiles, 256, 0, st>>>(count, ncell, tsum);
    grid_tilescan_kernel<<<1, 256, 0, st>>>(tsum, ntiles, start, ncell);
    grid_scan_kernel<<<ntiles, 256, 0, st>>>(count, ncell, tsum, start);
    return nbp_launch_status();
}

}  // namespace
