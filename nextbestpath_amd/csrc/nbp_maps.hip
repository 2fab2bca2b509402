// nbp_maps.hip -- per-step map accumulation: world points -> agent-centred top-down count
// images (HBM-bound scatter; 12 B read per point, fp32 atomics resolve in L2).
//
// Replaces next_best_path/utility/utils.py:160-223 (get_point_position_in_the_img,
// transform_points_to_n_pieces, map_points_to_n_imgs) and the slab split / projections of
// next_best_path/testers/nbp_planning.py:114-127,172-183.  All arithmetic is the
// reference's fp32 sequence: sub, negate, add 40, multiply by fp32(S/80), rint (half to
// even), bounds test -- so the integer cell of every point is bit-identical.
#include "common.h"
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>
#pragma clang fp contract(off)

namespace {

struct Bounds { float b[8]; int n; };

__device__ __forceinline__ bool cell_of(float v0, float v1, float lo, float sc0, float sc1, int S0, int S1,
                                        int& i0, int& i1) {
    const float f0 = rintf((v0 - lo) * sc0);
    const float f1 = rintf((v1 - lo) * sc1);
    const bool ok = f0 >= 0.f && f0 < (float)S0 && f1 >= 0.f && f1 < (float)S1;
    i0 = ok ? (int)f0 : 0;
    i1 = ok ? (int)f1 : 0;
    return ok;
}

__global__ __launch_bounds__(256) void transform_points_kernel(const float* __restrict__ p, long long N, float cx,
                                                               float cz, float* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N;
         i += (long long)gridDim.x * blockDim.x) {
        const float x = p[3 * i], z = p[3 * i + 2];
        out[2 * i] = -(z - cz);
        out[2 * i + 1] = -(x - cx);
    }
}

__global__ __launch_bounds__(256) void map_points_kernel(const float* __restrict__ pts, int n, long long m, int S0,
                                                         int S1, float lo, float sc0, float sc1,
                                                         float* __restrict__ out) {
    const long long total = (long long)n * m;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const float2 v = reinterpret_cast<const float2*>(pts)[i];
        int i0, i1;
        if (cell_of(v.x, v.y, lo, sc0, sc1, S0, S1, i0, i1)) {
            const long long img = i / m;
            atomicAdd(out + (img * S0 + i0) * S1 + i1, 1.0f);
        }
    }
}

__global__ void point_position_kernel(const float* __restrict__ pts, long long K, float lo, float sc0, float sc1,
                                      long long* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < K;
         i += (long long)gridDim.x * blockDim.x) {
        out[i] = (long long)rintf((pts[2 * i] - lo) * sc0);
        out[K + i] = (long long)rintf((pts[2 * i + 1] - lo) * sc1);
    }
}

// ---- fused accumulation with per-workgroup LDS pre-aggregation.
// Walls are vertical, so thousands of points fall into the same top-down cell: sending every
// point as its own device-scope atomic serialises on the hot addresses.  Each workgroup owns
// AGG_POINTS consecutive points, counts them in an LDS open-addressing table (2^SLOT_BITS slots) keyed by
// (channel, cell) with LDS atomics, then flushes one global atomicAdd(count) per distinct key.
constexpr int AGG_EMPTY = -1;
typedef float f32x3 __attribute__((ext_vector_type(3), aligned(4)));

template <int SLOT_BITS>
__device__ __forceinline__ void agg_add(int* keys, int* cnts, int key, float* __restrict__ out) {
    unsigned h = ((unsigned)key * 0x9E3779B1u) >> (32 - SLOT_BITS);
#pragma unroll 1
    for (int probe = 0; probe < 24; ++probe) {
        const int old = atomicCAS(&keys[h], AGG_EMPTY, key);
        if (old == AGG_EMPTY || old == key) { atomicAdd(&cnts[h], 1); return; }
        h = (h + 1) & ((1 << SLOT_BITS) - 1);
    }
    atomicAdd(out + key, 1.0f);                                   // table region saturated: go direct
}

template <int SLOT_BITS>
__device__ __forceinline__ void accumulate_point(float x, float y, float z, float cx, float cz, const Bounds& bd,
                                                 float band_lo, float band_hi, int S, float lo, float sc,
                                                 int* keys, int* cnts, float* __restrict__ out) {
    int i0, i1;
    if (!cell_of(-(z - cz), -(x - cx), lo, sc, sc, S, S, i0, i1)) return;
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) cnt += (k < bd.n && bd.b[k] < y) ? 1 : 0;
    const int bin = cnt - 1;
    const int ch = (bin >= 0 && bin < 4) ? bin : 4;
    const int cell = i0 * S + i1;
    agg_add<SLOT_BITS>(keys, cnts, ch * S * S + cell, out);
    if (band_lo < y && y < band_hi) agg_add<SLOT_BITS>(keys, cnts, 5 * S * S + cell, out);
}

// Trajectory channel of the network input (nbp_planning.py:129-137: the camera positions so far, transformed like the
// cloud and counted per cell): done by ONE extra workgroup of the accumulation launch.  pts = device history (n_old points
// valid), fresh = up to 8 new positions riding in the kernel arguments (appended to pts here), out = [S,S] zeroed.
struct TrajArgs { float* pts; float* out; int n_old, n_fresh; float fresh[24]; };

// One rollout's arguments of the accumulation (the batched launch carries up to MAP_BATCH of them in the kernel arguments)
struct MapItem {
    const float* p; long long N; const long long* n_dev; float cx, cz; Bounds bd; float band_lo, band_hi; float* out; TrajArgs tr;
};

template <int AGG_POINTS, int SLOT_BITS, int THREADS, int ROUNDS>
__device__ __forceinline__ void map_accumulate_body(const MapItem& a, unsigned wg, unsigned n_wg, int S, float lo, float sc, int* keys,
                                                    int* cnts) {
    constexpr int AGG_SLOTS = 1 << SLOT_BITS;
    const float* __restrict__ p = a.p;
    float* __restrict__ out = a.out;
    const TrajArgs& tr = a.tr;
    const float cx = a.cx, cz = a.cz;
    if (tr.out && wg == n_wg - 1) {                    // the trajectory workgroup (appended to the grid)
        for (int i = threadIdx.x; i < tr.n_old + tr.n_fresh; i += THREADS) {
            float x, z;
            if (i < tr.n_old) {
                x = tr.pts[3 * i]; z = tr.pts[3 * i + 2];
            } else {
                float y = 0.f;
                x = z = 0.f;
#pragma unroll
                for (int f = 0; f < 8; ++f)
                    if (i - tr.n_old == f) { x = tr.fresh[3 * f]; y = tr.fresh[3 * f + 1]; z = tr.fresh[3 * f + 2]; }
                tr.pts[3 * i] = x; tr.pts[3 * i + 1] = y; tr.pts[3 * i + 2] = z;
            }
            int i0, i1;
            if (cell_of(-(z - cz), -(x - cx), lo, sc, sc, S, S, i0, i1)) atomicAdd(tr.out + i0 * S + i1, 1.0f);
        }
        return;
    }
    long long N = a.N;
    if (a.n_dev) N = *a.n_dev;                   // cloud size lives on the device (no host sync per step)
    const long long first = (long long)wg * AGG_POINTS;
    if (first >= N) return;
    const long long last = first + AGG_POINTS < N ? first + AGG_POINTS : N;
    for (int i = threadIdx.x; i < AGG_SLOTS; i += THREADS) { keys[i] = AGG_EMPTY; cnts[i] = 0; }
    __syncthreads();
    // 12 B per point: three consecutive dword loads per lane (768 contiguous bytes per wave instruction).  All of a
    // thread's points of a round are fetched BEFORE the first table insert: the probe loop's LDS atomics would otherwise fence
    // every iteration's loads behind the previous iteration (one HBM round trip per point instead of one per thread).
    // ROUNDS > 1 (the batched launch: many workgroups in flight anyway): the same table takes several rounds of points before
    // it is flushed -- consecutive points fall on few distinct cells (walls are vertical), so the flush, one device-scope atomic
    // per distinct key and the kernel's bound (~5 G atomics/s), shrinks per point.
    constexpr int PER = AGG_POINTS / THREADS / ROUNDS;
#pragma unroll 1
    for (int rd = 0; rd < ROUNDS; ++rd) {
        const long long r0 = first + (long long)rd * (AGG_POINTS / ROUNDS);
        if (r0 >= last) break;
        float px[PER], py[PER], pz[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const long long i = r0 + threadIdx.x + (long long)k * THREADS;
            const bool in = i < last;
            // one 12-byte load per point (global_load_dwordx3: a third of the load instructions of three dword loads)
            f32x3 v = {__builtin_nanf(""), 0.f, 0.f};           // NaN fails cell_of: the slot is skipped
            if (in) v = *reinterpret_cast<const f32x3*>(p + 3 * i);
            px[k] = v[0]; py[k] = v[1]; pz[k] = v[2];
        }
#pragma unroll
        for (int k = 0; k < PER; ++k)
            accumulate_point<SLOT_BITS>(px[k], py[k], pz[k], cx, cz, a.bd, a.band_lo, a.band_hi, S, lo, sc, keys, cnts, out);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < AGG_SLOTS; i += THREADS)
        if (keys[i] != AGG_EMPTY) atomicAdd(out + keys[i], (float)cnts[i]);
}

template <int AGG_POINTS, int SLOT_BITS, int THREADS>
__global__ __launch_bounds__(THREADS) void map_accumulate_kernel(MapItem a, int S, float lo, float sc) {
    __shared__ int keys[1 << SLOT_BITS];
    __shared__ int cnts[1 << SLOT_BITS];
    map_accumulate_body<AGG_POINTS, SLOT_BITS, THREADS, 1>(a, blockIdx.x, gridDim.x, S, lo, sc, keys, cnts);
}

// The same for several rollouts in ONE launch (blockIdx.y = rollout): the step's kernels are latency-bound (one workgroup
// per CU, a chain of dependent round trips), so the rollouts of a lock-step group cost one such chain instead of one each.
// The grid's x extent is the largest rollout's; a rollout's trajectory workgroup is its own last one (n_wg[r] - 1).
constexpr int MAP_BATCH = 16;
struct MapBatch { MapItem it[MAP_BATCH]; unsigned n_wg[MAP_BATCH]; };

template <int AGG_POINTS, int SLOT_BITS, int THREADS, int ROUNDS>
__global__ __launch_bounds__(THREADS) void map_accumulate_batch_kernel(MapBatch b, int S, float lo, float sc) {
    __shared__ int keys[1 << SLOT_BITS];
    __shared__ int cnts[1 << SLOT_BITS];
    const unsigned r = blockIdx.y;
    if (blockIdx.x >= b.n_wg[r]) return;
    map_accumulate_body<AGG_POINTS, SLOT_BITS, THREADS, ROUNDS>(b.it[r], blockIdx.x, b.n_wg[r], S, lo, sc, keys, cnts);
}

// ------------------------------------------------------------------ tile-binned shadow copy of the cloud (round 4)
// The six maps are a TRANSLATED window of the world (no rotation: cell = rint((-(v - c) + 40) * S / 80)), so a world-space tile of
// T x T units always lands on a (T S / 80 + 1)^2 block of cells.  The canonical cloud stays in append order (coverage sampling,
// parity); beside it every point is copied ONCE, before the first map build that sees it, into a page of its (x, z) tile:
//   store = [BinDesc | tile_count[nt] | page_hash[H] | page_info[max_pages] | side_list[BIN_OVF] | pages[max_pages][2048][3]]
// A map build is two launches:
//   bin_append_kernel  files the points the store has not seen yet ([n_binned, N): one step's ~29 k) into their tiles' pages;
//   map_binned_kernel  one workgroup per page: a DENSE 16 x 16-cell x 6-channel histogram in LDS (plain LDS adds, no CAS probing),
//                      flushed with one global atomic per non-zero counter; tiles outside the +-40 window are never read; plus a
//                      few workgroups for the side list and the trajectory channel.
// Every point is counted from a page, i.e. pre-aggregated per tile.  A fused single-launch form that counted the new points with one
// global atomic each was measured (profiles/r04/map_bins_in_situ.txt): 2 us faster alone, and 5 % of the lock-step's steps/s lost --
// a frame's points fall on a handful of cells, and thousands of same-address device-scope atomics stall the memory channels the
// concurrent convolutions stream through (the page side alone is free there).
// Counts are integers, so the maps are bit-identical to map_accumulate_kernel's whatever the order (tests/test_gpu_maps.py, every
// rollout parity test).
// Store layout, slot reservation and the wave-cooperative filing: nbp_bins.h (shared with the un-projection launch of nbp_sim.hip,
// which files the points it appends so that a single rollout's build is ONE launch: nbp_step_maps_prefiled_f32).
#include "nbp_bins.h"
constexpr int BIN_R = 16;                                             // LDS histogram: 16 x 16 cells (a 2.5-unit tile is 9 x 9 + slack)
constexpr int BIN_APPEND_WGS = 128, BIN_OVF_WGS = 4;     // 128 x 256 threads: one step's ~29 k new points in one pass

__device__ __forceinline__ int channel_of(float y, const Bounds& bd) {
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) cnt += (k < bd.n && bd.b[k] < y) ? 1 : 0;
    const int bin = cnt - 1;
    return (bin >= 0 && bin < 4) ? bin : 4;
}

__device__ __forceinline__ void count_direct(float x, float y, float z, const MapItem& a, int S, float lo, float sc) {
    int i0, i1;
    if (!cell_of(-(z - a.cz), -(x - a.cx), lo, sc, sc, S, S, i0, i1)) return;
    atomicAdd(a.out + channel_of(y, a.bd) * S * S + i0 * S + i1, 1.0f);
    if (a.band_lo < y && y < a.band_hi) atomicAdd(a.out + 5 * S * S + i0 * S + i1, 1.0f);
}

// workgroup `wg` of `n_wg`: points [n_binned, N) of the cloud are filed into pages; the workgroup that finishes last advances n_binned.
// The launch also clears what the map launch behind it accumulates into (zero6: the six maps, zero1: the trajectory channel, SS floats
// each): two memset launches less per build -- every launch boundary is an L2 write-back / invalidate under the other group's forward.
__device__ __forceinline__ void bin_append_body(char* store, const float* __restrict__ cloud, long long N, unsigned wg, unsigned n_wg,
                                                float* __restrict__ zero6, float* __restrict__ zero1, int SS) {
    bin_clear_maps(zero6, zero1, SS, wg, n_wg);          // (SS % 4 == 0: launcher)
    const BinView v = bin_view(store);
    const long long first = v.d->n_binned;
    if (v.d->error == 0u) {                               // (a broken store files nothing more: its builds scan the cloud)
        const BinGeom g = bin_geom(v);
        for (long long i0 = first + (long long)wg * 256; i0 < N; i0 += (long long)n_wg * 256) {      // uniform per workgroup
            const long long i = i0 + threadIdx.x;
            const bool active = i < N;
            f32x3 p = {__builtin_nanf(""), 0.f, 0.f};
            if (active) p = *reinterpret_cast<const f32x3*>(cloud + 3 * i);
            bin_file_wave(v, g, active, p, i);
        }
    }
    __shared__ int last;
    __syncthreads();
    if (threadIdx.x == 0)       // (no fence: __threadfence() is an L2 write-back + invalidate here, and the last workgroup reads nothing
        last = atomicAdd(&v.d->ticket, 1u) == n_wg - 1;      //  the others wrote: what they filed is read by later launches only)
    __syncthreads();
    if (last && threadIdx.x == 0) {
        v.d->n_binned = N > first ? N : first;
        v.d->ticket = 0;
    }
}

// The launch-constant part of a store's header.  The single-rollout launch gets it in its kernel arguments (the host remembers what
// nbp_cloud_bins_init wrote: one dependent round trip less in a launch that is a chain of them); valid = 0: read from the store.
struct BinStatic { unsigned long long off_count, off_info, off_ovf, off_pages; int nx, max_pages; float x0, z0, tile; int valid; };

// One workgroup of 256 threads.  Roles by index: [0, n_page_wg) pages, then BIN_OVF_WGS side list, 1 trajectory.
// SPEC (the single-rollout launch, latency-bound): a page's 2048 slots are loaded without waiting for its tile's count -- slots no
// point was filed into hold NaN since nbp_cloud_bins_init and fail cell_of -- so the chain is {header word, page info} -> page
// data -> LDS -> flush; without it (the group launch, throughput-bound) only the filled slots are read.
template <bool SPEC>
__device__ __forceinline__ void map_binned_body(const MapItem& a, char* store, unsigned wg, unsigned n_page_wg, int S, float lo,
                                                float sc, int* hist, BinStatic g = BinStatic{0, 0, 0, 0, 0, 0, 0.f, 0.f, 0.f, 0}) {
    float* __restrict__ out = a.out;
    const float cx = a.cx, cz = a.cz;
    const int SS = S * S;
    if (wg == n_page_wg + BIN_OVF_WGS) {                    // trajectory workgroup (as in map_accumulate_body)
        const TrajArgs& tr = a.tr;
        if (!tr.out) return;
        for (int i = threadIdx.x; i < tr.n_old + tr.n_fresh; i += 256) {
            float x, z;
            if (i < tr.n_old) {
                x = tr.pts[3 * i]; z = tr.pts[3 * i + 2];
            } else {
                float y = 0.f;
                x = z = 0.f;
#pragma unroll
                for (int f = 0; f < 8; ++f)
                    if (i - tr.n_old == f) { x = tr.fresh[3 * f]; y = tr.fresh[3 * f + 1]; z = tr.fresh[3 * f + 2]; }
                tr.pts[3 * i] = x; tr.pts[3 * i + 1] = y; tr.pts[3 * i + 2] = z;
            }
            int i0, i1;
            if (cell_of(-(z - cz), -(x - cx), lo, sc, sc, S, S, i0, i1)) atomicAdd(tr.out + i0 * S + i1, 1.0f);
        }
        return;
    }
    BinDesc* const d = reinterpret_cast<BinDesc*>(store);
    if (!g.valid) g = BinStatic{d->off_count, d->off_info, d->off_ovf, d->off_pages, d->nx, d->max_pages, d->x0, d->z0, d->tile, 1};
    const BinView v{d, reinterpret_cast<unsigned*>(store + g.off_count), nullptr, reinterpret_cast<unsigned*>(store + g.off_info),
                    reinterpret_cast<unsigned*>(store + g.off_ovf), reinterpret_cast<float*>(store + g.off_pages)};
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 dyn = *reinterpret_cast<const u32x4*>(store);      // n_pages, n_overflow, error, ticket: one load
    // (this workgroup's first page: its info word is fetched beside the header word, not behind it; page_bound <= max_pages)
    unsigned info_first = wg < n_page_wg && wg < (unsigned)g.max_pages ? v.info[wg] : 0u;
    const bool broken = dyn[2] != 0u;                      // (set by an append launch, never by this one)
    if (wg >= n_page_wg) {                                 // the side list -- or, for a broken store, the whole cloud -- with direct atomics
        const unsigned w = wg - n_page_wg;
        if (broken) {
            long long N = a.N;
            if (a.n_dev) N = *a.n_dev;
            for (long long i = (long long)w * 256 + threadIdx.x; i < N; i += (long long)BIN_OVF_WGS * 256) {
                const f32x3 p = *reinterpret_cast<const f32x3*>(a.p + 3 * i);
                count_direct(p[0], p[1], p[2], a, S, lo, sc);
            }
            return;
        }
        const unsigned n = min(dyn[1], BIN_OVF);
        for (unsigned i = w * 256 + threadIdx.x; i < n; i += BIN_OVF_WGS * 256) {
            const f32x3 p = *reinterpret_cast<const f32x3*>(a.p + 3 * (size_t)v.ovf[i]);
            count_direct(p[0], p[1], p[2], a, S, lo, sc);
        }
        // points no launch has filed yet ([n_binned, N): none after bin_append_kernel, none in the prefiled form when every append
        // went through the filing un-projection) are counted directly as well: a caller's bookkeeping slip costs time, never points
        long long N = a.N;
        if (a.n_dev) N = *a.n_dev;
        for (long long i = v.d->n_binned + (long long)w * 256 + threadIdx.x; i < N; i += (long long)BIN_OVF_WGS * 256) {
            const f32x3 p = *reinterpret_cast<const f32x3*>(a.p + 3 * i);
            count_direct(p[0], p[1], p[2], a, S, lo, sc);
        }
        return;
    }
    if (broken) return;
    const unsigned n_pages = min(dyn[0], (unsigned)g.max_pages);
    // (one page per workgroup when the host's bound n_page_wg covers the pages that exist; a bound that is too low only costs time)
    for (unsigned page = wg; page < n_pages; page += n_page_wg) {
        const unsigned info = page == wg ? info_first : v.info[page];
        const int t = (int)(info & 0xffffu), k = (int)(info >> 16);
        int cnt = BIN_PAGE;
        if (!SPEC) {
            const unsigned tc = v.count[t];
            if (tc <= (unsigned)k * BIN_PAGE) continue;    // (cannot happen: a page exists once its first slot is reserved)
            cnt = (int)min((unsigned)BIN_PAGE, tc - (unsigned)k * BIN_PAGE);
        }
        // the block of cells the tile can reach (one cell of slack each way: a point outside it, or outside the LDS block, goes direct)
        const int tx = t % g.nx, tz = t / g.nx;
        const float T = g.tile;
        const float xa = g.x0 + tx * T, xb = g.x0 + (tx + 1) * T, za = g.z0 + tz * T, zb = g.z0 + (tz + 1) * T;
        const float r_hi = rintf((-(za - cz) - lo) * sc) + 1.f, r_lo = rintf((-(zb - cz) - lo) * sc) - 1.f;
        const float c_hi = rintf((-(xa - cx) - lo) * sc) + 1.f, c_lo = rintf((-(xb - cx) - lo) * sc) - 1.f;
        if (r_hi < 0.f || r_lo >= (float)S || c_hi < 0.f || c_lo >= (float)S) continue;      // the tile is outside the window
        const int r0 = (int)r_lo, c0 = (int)c_lo;
        __syncthreads();                                   // (the previous page's flush has read the histogram)
        for (int i = threadIdx.x; i < 6 * BIN_R * BIN_R; i += 256) hist[i] = 0;
        __syncthreads();
        const float* __restrict__ pg = v.pages + (size_t)page * BIN_PAGE * 3;
        constexpr int PER = BIN_PAGE / 256;
        float px[PER], py[PER], pz[PER];
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int i = threadIdx.x + q * 256;
            f32x3 p = {__builtin_nanf(""), 0.f, 0.f};
            if (i < cnt) p = *reinterpret_cast<const f32x3*>(pg + 3 * i);
            px[q] = p[0]; py[q] = p[1]; pz[q] = p[2];
        }
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            int i0, i1;
            if (!cell_of(-(pz[q] - cz), -(px[q] - cx), lo, sc, sc, S, S, i0, i1)) continue;  // NaN (a slot never written) fails
            const int ch = channel_of(py[q], a.bd);
            const bool band = a.band_lo < py[q] && py[q] < a.band_hi;
            const int li = i0 - r0, lj = i1 - c0;
            if ((unsigned)li < (unsigned)BIN_R && (unsigned)lj < (unsigned)BIN_R) {
                atomicAdd(&hist[(ch * BIN_R + li) * BIN_R + lj], 1);
                if (band) atomicAdd(&hist[(5 * BIN_R + li) * BIN_R + lj], 1);
            } else {
                atomicAdd(out + ch * SS + i0 * S + i1, 1.0f);
                if (band) atomicAdd(out + 5 * SS + i0 * S + i1, 1.0f);
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 6 * BIN_R * BIN_R; i += 256) {
            const int c = hist[i];
            if (c) {
                const int ch = i / (BIN_R * BIN_R), li = (i / BIN_R) % BIN_R, lj = i % BIN_R;
                atomicAdd(out + ch * SS + (r0 + li) * S + (c0 + lj), (float)c);               // (in the window: the point passed cell_of)
            }
        }
    }
}

__global__ __launch_bounds__(256) void bin_append_kernel(char* store, const float* __restrict__ cloud, long long N, const long long* n_dev,
                                                         float* zero6, float* zero1, int SS) {
    bin_append_body(store, cloud, n_dev ? *n_dev : N, blockIdx.x, gridDim.x, zero6, zero1, SS);
}
template <bool SPEC>
__global__ __launch_bounds__(256) void map_binned_kernel(MapItem a, char* store, unsigned n_page_wg, int S, float lo, float sc, BinStatic g) {
    __shared__ int hist[6 * BIN_R * BIN_R];
    map_binned_body<SPEC>(a, store, blockIdx.x, n_page_wg, S, lo, sc, hist, g);
}
struct BinBatch { char* store[MAP_BATCH]; unsigned n_page_wg[MAP_BATCH]; };
__global__ __launch_bounds__(256) void bin_append_batch_kernel(MapBatch b, BinBatch s, int SS) {
    const MapItem& a = b.it[blockIdx.y];
    bin_append_body(s.store[blockIdx.y], a.p, a.n_dev ? *a.n_dev : a.N, blockIdx.x, gridDim.x, a.out, a.tr.out, SS);
}
__global__ __launch_bounds__(256) void map_binned_batch_kernel(MapBatch b, BinBatch s, int S, float lo, float sc) {
    __shared__ int hist[6 * BIN_R * BIN_R];
    const unsigned r = blockIdx.y;
    if (blockIdx.x >= s.n_page_wg[r] + BIN_OVF_WGS + 1) return;
    map_binned_body<false>(b.it[r], s.store[r], blockIdx.x, s.n_page_wg[r], S, lo, sc, hist);
}

__global__ __launch_bounds__(256) void bin_init_kernel(char* store, BinDesc d) {
    BinDesc* dd = reinterpret_cast<BinDesc*>(store);
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    if (gid == 0) *dd = d;
    unsigned* count = reinterpret_cast<unsigned*>(store + d.off_count);
    unsigned long long* table = reinterpret_cast<unsigned long long*>(store + d.off_table);
    float* pages = reinterpret_cast<float*>(store + d.off_pages);
    for (size_t i = gid; i < (size_t)d.nt; i += stride) count[i] = 0u;
    for (size_t i = gid; i <= (size_t)d.hash_mask; i += stride) table[i] = BIN_EMPTY;
    const float nanv = __builtin_nanf("");
    for (size_t i = gid; i < (size_t)d.max_pages * BIN_PAGE * 3; i += stride) pages[i] = nanv;     // empty slots fail cell_of
}

// host: geometry of the store for a scene whose (x, z) extent is [lo, hi] and a cloud of `capacity` points
static BinDesc bin_desc(const float* lo_xz, const float* hi_xz, long long capacity) {
    BinDesc d;
    memset(&d, 0, sizeof d);
    float T = 2.5f;                                        // 8 cells of the 0.3125-unit grid
    const float ex = hi_xz[0] - lo_xz[0], ez = hi_xz[1] - lo_xz[1];
    while (((double)ex / T + 3.0) * ((double)ez / T + 3.0) > 65535.0) T *= 2.f;      // tile ids are 16 bits (2.5-unit tiles up to ~630 x 630 units;
                                                                                        // larger tiles still work: cells beyond the 16 x 16 LDS block go direct)
    d.tile = T; d.inv_t = 1.0f / T;
    d.x0 = lo_xz[0] - T; d.z0 = lo_xz[1] - T;              // one tile of margin
    d.nx = (int)(ex / T) + 3; d.nz = (int)(ez / T) + 3;
    d.nt = d.nx * d.nz;
    // every occupied tile ends in a partial page; scenes are mostly empty space, so the pool holds 4096 of those (a rollout's cloud
    // touches a few hundred tiles) -- beyond, the side list takes over
    d.max_pages = (int)(capacity / BIN_PAGE) + (d.nt < 4096 ? d.nt : 4096) + 64;
    unsigned hsize = 1024;
    while (hsize < 2u * (unsigned)d.max_pages) hsize *= 2;
    d.hash_mask = hsize - 1;
    unsigned long long off = 256;
    auto take = [&](unsigned long long bytes) { const unsigned long long o = off; off += (bytes + 255) / 256 * 256; return o; };
    d.off_count = take((unsigned long long)d.nt * 4);
    d.off_table = take((unsigned long long)hsize * 8);
    d.off_info = take((unsigned long long)d.max_pages * 4);
    d.off_ovf = take((unsigned long long)BIN_OVF * 4);
    d.off_pages = take((unsigned long long)d.max_pages * BIN_PAGE * 12);
    d.total_bytes = off;
    return d;
}
// what nbp_cloud_bins_init wrote into a store's header, by store address (a store re-initialised elsewhere, or one the table does not
// know, is simply read from the device by the launch)
static std::mutex g_bin_mu;
static std::unordered_map<const void*, BinDesc> g_bin_known;
static BinStatic bin_static_of(const void* store) {
    std::lock_guard<std::mutex> lk(g_bin_mu);
    const auto it = g_bin_known.find(store);
    if (it == g_bin_known.end()) return BinStatic{0, 0, 0, 0, 0, 0, 0.f, 0.f, 0.f, 0};
    const BinDesc& d = it->second;
    return BinStatic{d.off_count, d.off_info, d.off_ovf, d.off_pages, d.nx, d.max_pages, d.x0, d.z0, d.tile, 1};
}
static bool bin_geometry_ok(const float* lo_xz, const float* hi_xz, long long capacity) {
    return lo_xz && hi_xz && capacity > 0 && capacity < 65535ll * BIN_PAGE && hi_xz[0] >= lo_xz[0] && hi_xz[1] >= lo_xz[1] &&
           hi_xz[0] - lo_xz[0] < 1e6f && hi_xz[1] - lo_xz[1] < 1e6f;
}

inline float grid_scale(int S, float lo, float hi) { return (float)((double)S / ((double)hi - (double)lo)); }

}  // namespace

extern "C" int nbp_transform_points_f32(const float* points, long long N, float cx, float cy, float cz, float* out_2d,
                                        void* stream) {
    NBP_ENTER();
    (void)cy;
    NBP_RETURN_IF(N < 0, NBP_E_ARG);
    if (N == 0) return 0;
    NBP_RETURN_IF(!points || !out_2d, NBP_E_ARG);
    transform_points_kernel<<<nbp_ew_grid(N, 256), 256, 0, (hipStream_t)stream>>>(points, N, cx, cz, out_2d);
    return nbp_launch_status();
}

extern "C" int nbp_map_points_to_imgs_f32(const float* pts2d, int n, long long m, int S0, int S1, float lo, float hi,
                                          float* out, void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(!out || n < 1 || m < 0 || S0 < 1 || S1 < 1 || !(hi > lo), NBP_E_ARG);
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(out, 0, (size_t)n * S0 * S1 * sizeof(float), st);
    if (e != hipSuccess) return (int)e;
    if (m == 0) return 0;
    NBP_RETURN_IF(!pts2d, NBP_E_ARG);
    map_points_kernel<<<nbp_ew_grid((long long)n * m, 256), 256, 0, st>>>(pts2d, n, m, S0, S1, lo,
                                                                          grid_scale(S0, lo, hi),
                                                                          grid_scale(S1, lo, hi), out);
    return nbp_launch_status();
}

extern "C" int nbp_point_position_i64(const float* pts2d, long long K, int S0, int S1, float lo, float hi,
                                      long long* out_2xK, void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(!pts2d || !out_2xK || K < 1 || S0 < 1 || S1 < 1 || !(hi > lo), NBP_E_ARG);
    point_position_kernel<<<nbp_ew_grid(K, 256), 256, 0, (hipStream_t)stream>>>(pts2d, K, lo, grid_scale(S0, lo, hi),
                                                                                grid_scale(S1, lo, hi), out_2xK);
    return nbp_launch_status();
}

extern "C" int nbp_map_accumulate_f32(const float* points, long long N, const long long* N_dev_or_null, float cx,
                                      float cy, float cz, const float* bounds_host, int n_bounds, float band_lo, float band_hi, int S,
                                      float lo, float hi, float* out6, void* stream) {
    NBP_ENTER();
    (void)cy;
    NBP_RETURN_IF(!out6 || N < 0 || S < 1 || !(hi > lo), NBP_E_ARG);
    NBP_RETURN_IF(n_bounds < 0 || n_bounds > 8 || (n_bounds > 0 && !bounds_host), NBP_E_ARG);
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(out6, 0, (size_t)6 * S * S * sizeof(float), st);
    if (e != hipSuccess) return (int)e;
    if (N == 0) return 0;
    NBP_RETURN_IF(!points, NBP_E_ARG);
    NBP_RETURN_IF(((uintptr_t)points & 3) != 0, NBP_E_ARG);
    NBP_RETURN_IF((long long)6 * S * S >= (1ll << 31), NBP_E_SHAPE);
    Bounds bd;
    for (int k = 0; k < 8; ++k) bd.b[k] = k < n_bounds ? bounds_host[k] : 0.f;
    bd.n = n_bounds;
    // 1024-thread workgroups of 8192 points: the loop is a chain of LDS-atomic round trips per point, so the
    // table is shared by 16 waves in flight (26 us for 1.3 M points; 256-thread workgroups: 37 us; smaller
    // point batches flush more distinct keys to L2 and lose)
    map_accumulate_kernel<8192, 13, 1024><<<(unsigned)nbp_cdiv(N, 8192), 1024, 0, st>>>(
        MapItem{points, N, N_dev_or_null, cx, cz, bd, band_lo, band_hi, out6, TrajArgs{}}, S, lo, grid_scale(S, lo, hi));
    return nbp_launch_status();
}

extern "C" int nbp_step_maps_f32(const float* points, long long N, const long long* N_dev_or_null, float cx, float cy,
                                 float cz, const float* bounds_host, int n_bounds, float band_lo, float band_hi, int S,
                                 float lo, float hi, float* traj_pts, int n_traj_old, const float* traj_fresh_host,
                                 int n_traj_fresh, float* out6, float* net_in5, void* stream) {
    NBP_ENTER();
    (void)cy;
    NBP_RETURN_IF(!out6 || !net_in5 || !traj_pts || N < 0 || S < 1 || !(hi > lo), NBP_E_ARG);
    NBP_RETURN_IF(n_bounds < 0 || n_bounds > 8 || (n_bounds > 0 && !bounds_host), NBP_E_ARG);
    NBP_RETURN_IF(n_traj_old < 0 || n_traj_fresh < 0 || n_traj_fresh > 8 || (n_traj_fresh > 0 && !traj_fresh_host), NBP_E_ARG);
    NBP_RETURN_IF(N > 0 && (!points || ((uintptr_t)points & 3) != 0), NBP_E_ARG);
    NBP_RETURN_IF((long long)6 * S * S >= (1ll << 31), NBP_E_SHAPE);
    hipStream_t st = (hipStream_t)stream;
    const size_t SS = (size_t)S * S;
    hipError_t e = hipMemsetAsync(out6, 0, 6 * SS * sizeof(float), st);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(net_in5 + 4 * SS, 0, SS * sizeof(float), st);
    if (e != hipSuccess) return (int)e;
    Bounds bd;
    for (int k = 0; k < 8; ++k) bd.b[k] = k < n_bounds ? bounds_host[k] : 0.f;
    bd.n = n_bounds;
    TrajArgs tr;
    tr.pts = traj_pts; tr.out = net_in5 + 4 * SS; tr.n_old = n_traj_old; tr.n_fresh = n_traj_fresh;
    for (int i = 0; i < 24; ++i) tr.fresh[i] = i < 3 * n_traj_fresh ? traj_fresh_host[i] : 0.f;
    map_accumulate_kernel<8192, 13, 1024><<<(unsigned)nbp_cdiv(N, 8192) + 1, 1024, 0, st>>>(
        MapItem{points, N, N_dev_or_null, cx, cz, bd, band_lo, band_hi, out6, tr}, S, lo, grid_scale(S, lo, hi));
    int rc = nbp_launch_status();
    if (rc) return rc;
    e = hipMemcpyAsync(net_in5, out6, 4 * SS * sizeof(float), hipMemcpyDeviceToDevice, st);
    return e == hipSuccess ? 0 : (int)e;
}

// nbp_step_maps_f32 for n <= 16 rollouts of a lock-step group in ONE kernel launch, two memsets and one strided copy.
// The rollouts' map stacks are slices of one tensor out6_all [n][6][S][S] and their network inputs of net_in_all [n][5][S][S];
// everything else is per rollout (arrays of n entries, HOST memory; device pointers inside).
extern "C" int nbp_step_maps_batch_f32(int n, const float* const* points, const long long* N_cap, const long long* const* N_dev,
                                       const float* poses_xyz_host, const float* bounds_host, const int* n_bounds,
                                       const float* band_lo_hi_host, int S, float lo, float hi, float* const* traj_pts,
                                       const int* n_traj_old, const float* traj_fresh_host, const int* n_traj_fresh,
                                       float* out6_all, float* net_in_all, void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(n < 1 || n > MAP_BATCH || !points || !N_cap || !N_dev || !poses_xyz_host || !bounds_host || !n_bounds ||
                  !band_lo_hi_host || !traj_pts || !n_traj_old || !traj_fresh_host || !n_traj_fresh || !out6_all || !net_in_all, NBP_E_ARG);
    NBP_RETURN_IF(S < 1 || !(hi > lo) || (long long)6 * S * S >= (1ll << 31), NBP_E_SHAPE);
    hipStream_t st = (hipStream_t)stream;
    const size_t SS = (size_t)S * S;
    // points per workgroup = 8192 x rounds (NBP_MAP_ROUNDS = 1 | 2 | 4): with a group's workgroups side by side the chip is full
    // anyway, and a table that sees more consecutive points flushes fewer keys per point
    constexpr int rounds = 4;
    MapBatch b;
    unsigned max_wg = 1;
    for (int r = 0; r < MAP_BATCH; ++r) {
        const int q = r < n ? r : 0;
        NBP_RETURN_IF(r < n && (N_cap[q] < 0 || (N_cap[q] > 0 && (!points[q] || ((uintptr_t)points[q] & 3) != 0)) || !traj_pts[q] ||
                                n_bounds[q] < 0 || n_bounds[q] > 8 || n_traj_old[q] < 0 || n_traj_fresh[q] < 0 || n_traj_fresh[q] > 8), NBP_E_ARG);
        MapItem& it = b.it[r];
        it.p = points[q]; it.N = N_cap[q]; it.n_dev = N_dev[q];
        it.cx = poses_xyz_host[3 * q]; it.cz = poses_xyz_host[3 * q + 2];
        for (int k = 0; k < 8; ++k) it.bd.b[k] = k < n_bounds[q] ? bounds_host[8 * q + k] : 0.f;
        it.bd.n = n_bounds[q];
        it.band_lo = band_lo_hi_host[2 * q]; it.band_hi = band_lo_hi_host[2 * q + 1];
        it.out = out6_all + (size_t)q * 6 * SS;
        it.tr.pts = traj_pts[q]; it.tr.out = net_in_all + (size_t)q * 5 * SS + 4 * SS;
        it.tr.n_old = n_traj_old[q]; it.tr.n_fresh = n_traj_fresh[q];
        for (int i = 0; i < 24; ++i) it.tr.fresh[i] = i < 3 * n_traj_fresh[q] ? traj_fresh_host[24 * q + i] : 0.f;
        b.n_wg[r] = r < n ? (unsigned)nbp_cdiv(N_cap[q], 8192 * rounds) + 1 : 0;
        if (b.n_wg[r] > max_wg) max_wg = b.n_wg[r];
    }
    hipError_t e = hipMemsetAsync(out6_all, 0, (size_t)n * 6 * SS * sizeof(float), st);
    if (e != hipSuccess) return (int)e;
    e = hipMemset2DAsync(net_in_all + 4 * SS, 5 * SS * sizeof(float), 0, SS * sizeof(float), (size_t)n, st);       // the trajectory channels
    if (e != hipSuccess) return (int)e;
    if (rounds == 4) map_accumulate_batch_kernel<32768, 13, 1024, 4><<<dim3(max_wg, (unsigned)n), 1024, 0, st>>>(b, S, lo, grid_scale(S, lo, hi));
    else if (rounds == 2) map_accumulate_batch_kernel<16384, 13, 1024, 2><<<dim3(max_wg, (unsigned)n), 1024, 0, st>>>(b, S, lo, grid_scale(S, lo, hi));
    else map_accumulate_batch_kernel<8192, 13, 1024, 1><<<dim3(max_wg, (unsigned)n), 1024, 0, st>>>(b, S, lo, grid_scale(S, lo, hi));
    int rc = nbp_launch_status();
    if (rc) return rc;
    e = hipMemcpy2DAsync(net_in_all, 5 * SS * sizeof(float), out6_all, 6 * SS * sizeof(float), 4 * SS * sizeof(float), (size_t)n,
                         hipMemcpyDeviceToDevice, st);
    return e == hipSuccess ? 0 : (int)e;
}

// ---- tile-binned shadow copy of the cloud (kernels above): sizing, initialisation, and the step's map stage on it
extern "C" size_t nbp_cloud_bins_bytes(const float* lo_xz_host, const float* hi_xz_host, long long capacity) {
    if (!bin_geometry_ok(lo_xz_host, hi_xz_host, capacity)) return 0;
    return (size_t)bin_desc(lo_xz_host, hi_xz_host, capacity).total_bytes;
}

extern "C" int nbp_cloud_bins_geometry(const float* lo_xz_host, const float* hi_xz_host, long long capacity, int* nx_nz_maxpages_host,
                                       float* x0_z0_tile_host) {
    NBP_RETURN_IF(!bin_geometry_ok(lo_xz_host, hi_xz_host, capacity) || !nx_nz_maxpages_host || !x0_z0_tile_host, NBP_E_ARG);
    const BinDesc d = bin_desc(lo_xz_host, hi_xz_host, capacity);
    nx_nz_maxpages_host[0] = d.nx; nx_nz_maxpages_host[1] = d.nz; nx_nz_maxpages_host[2] = d.max_pages;
    x0_z0_tile_host[0] = d.x0; x0_z0_tile_host[1] = d.z0; x0_z0_tile_host[2] = d.tile;
    return 0;
}

// (Re-)initialises the store: empty tiles, no pages, every page slot NaN.  Call it whenever the cloud is reset to zero points.
extern "C" int nbp_cloud_bins_init(void* store, size_t store_bytes, const float* lo_xz_host, const float* hi_xz_host, long long capacity,
                                   void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(!store || ((uintptr_t)store & 255) || !bin_geometry_ok(lo_xz_host, hi_xz_host, capacity), NBP_E_ARG);
    const BinDesc d = bin_desc(lo_xz_host, hi_xz_host, capacity);
    NBP_RETURN_IF(store_bytes < d.total_bytes, NBP_E_WS);
    bin_init_kernel<<<1024, 256, 0, (hipStream_t)stream>>>((char*)store, d);
    {
        std::lock_guard<std::mutex> lk(g_bin_mu);
        if (g_bin_known.size() > 4096) g_bin_known.clear();        // (stores come and go with their rollouts: bounded, and only a cache)
        g_bin_known[store] = d;
    }
    return nbp_launch_status();
}

// The step's map stage (nbp_step_maps_f32) on the binned copy: points of the cloud that no build has seen yet are filed into their
// tiles' pages (bin_append_kernel), then the six maps are built from the pages (map_binned_kernel).  page_bound: the host's bound on
// the pages in use (the store's max_pages is always safe; a tighter bound launches fewer idle workgroups; one that is too low only
// costs time: page workgroups stride over every page that exists).
// traj_pts / net_in5 may both be null: only out6 is produced (the reference-API form accumulate_step_maps).
static int step_maps_binned(bool prefiled, void* store, int page_bound, const float* points, long long N, const long long* N_dev_or_null,
                            float cx, float cy, float cz, const float* bounds_host, int n_bounds, float band_lo, float band_hi,
                            int S, float lo, float hi, float* traj_pts, int n_traj_old, const float* traj_fresh_host,
                            int n_traj_fresh, float* out6, float* net_in5, void* stream) {
    NBP_ENTER();
    (void)cy;
    NBP_RETURN_IF(!store || page_bound < 1 || !out6 || N < 0 || S < 1 || !(hi > lo), NBP_E_ARG);
    NBP_RETURN_IF((net_in5 == nullptr) != (traj_pts == nullptr), NBP_E_ARG);
    NBP_RETURN_IF(n_bounds < 0 || n_bounds > 8 || (n_bounds > 0 && !bounds_host), NBP_E_ARG);
    NBP_RETURN_IF(n_traj_old < 0 || n_traj_fresh < 0 || n_traj_fresh > 8 || (n_traj_fresh > 0 && !traj_fresh_host), NBP_E_ARG);
    NBP_RETURN_IF(N > 0 && (!points || ((uintptr_t)points & 3) != 0), NBP_E_ARG);
    NBP_RETURN_IF((long long)6 * S * S >= (1ll << 31), NBP_E_SHAPE);
    hipStream_t st = (hipStream_t)stream;
    const size_t SS = (size_t)S * S;
    NBP_RETURN_IF(SS % 4 != 0 || ((uintptr_t)out6 & 15) || (net_in5 && ((uintptr_t)net_in5 & 15)), NBP_E_SHAPE);
    hipError_t e = hipSuccess;
    Bounds bd;
    for (int k = 0; k < 8; ++k) bd.b[k] = k < n_bounds ? bounds_host[k] : 0.f;
    bd.n = n_bounds;
    TrajArgs tr;
    memset(&tr, 0, sizeof tr);
    if (net_in5) {
        tr.pts = traj_pts; tr.out = net_in5 + 4 * SS; tr.n_old = n_traj_old; tr.n_fresh = n_traj_fresh;
        for (int i = 0; i < 24; ++i) tr.fresh[i] = i < 3 * n_traj_fresh ? traj_fresh_host[i] : 0.f;
    }
    if (!prefiled) {       // files the new points AND clears the maps / the trajectory channel (no memset launches)
        bin_append_kernel<<<BIN_APPEND_WGS, 256, 0, st>>>((char*)store, points, N, N_dev_or_null, out6, tr.out, (int)SS);
        const int rc0 = nbp_launch_status();
        if (rc0) return rc0;
    }
    // NBP_MAP_SPEC=1 (A/B): whole pages are loaded without waiting for their tile's count (one dependent load less, partial pages
    // read in full); measured within the box-to-box noise of the count-bounded form (profiles/r06/scatter_one_launch.txt)
    static const bool spec = nbp_tune_int("NBP_MAP_SPEC", 0) != 0;
    const MapItem item{points, N, N_dev_or_null, cx, cz, bd, band_lo, band_hi, out6, tr};
    const unsigned n_wg = (unsigned)page_bound + BIN_OVF_WGS + 1;
    if (spec) map_binned_kernel<true><<<n_wg, 256, 0, st>>>(item, (char*)store, (unsigned)page_bound, S, lo, grid_scale(S, lo, hi), bin_static_of(store));
    else map_binned_kernel<false><<<n_wg, 256, 0, st>>>(item, (char*)store, (unsigned)page_bound, S, lo, grid_scale(S, lo, hi), bin_static_of(store));
    int rc = nbp_launch_status();
    if (rc || !net_in5) return rc;
    e = hipMemcpyAsync(net_in5, out6, 4 * SS * sizeof(float), hipMemcpyDeviceToDevice, st);
    return e == hipSuccess ? 0 : (int)e;
}

extern "C" int nbp_step_maps_binned_f32(void* store, int page_bound, const float* points, long long N, const long long* N_dev_or_null,
                                        float cx, float cy, float cz, const float* bounds_host, int n_bounds, float band_lo, float band_hi,
                                        int S, float lo, float hi, float* traj_pts, int n_traj_old, const float* traj_fresh_host,
                                        int n_traj_fresh, float* out6, float* net_in5, void* stream) {
    return step_maps_binned(false, store, page_bound, points, N, N_dev_or_null, cx, cy, cz, bounds_host, n_bounds, band_lo, band_hi, S, lo, hi,
                            traj_pts, n_traj_old, traj_fresh_host, n_traj_fresh, out6, net_in5, stream);
}

// The same build as ONE launch (map_binned_kernel alone), for a store whose points were filed by the launch that appended them to
// the cloud and whose outputs that launch cleared (nbp_unproject_append_filed_f32 with store / zero6 = out6 / zero1 = net_in5 + 4 S^2).
// Points the store has not seen (appended by any other route since the last filing launch) are still counted -- directly, slowly.
extern "C" int nbp_step_maps_prefiled_f32(void* store, int page_bound, const float* points, long long N, const long long* N_dev_or_null,
                                          float cx, float cy, float cz, const float* bounds_host, int n_bounds, float band_lo, float band_hi,
                                          int S, float lo, float hi, float* traj_pts, int n_traj_old, const float* traj_fresh_host,
                                          int n_traj_fresh, float* out6, float* net_in5, void* stream) {
    return step_maps_binned(true, store, page_bound, points, N, N_dev_or_null, cx, cy, cz, bounds_host, n_bounds, band_lo, band_hi, S, lo, hi,
                            traj_pts, n_traj_old, traj_fresh_host, n_traj_fresh, out6, net_in5, stream);
}

// nbp_step_maps_batch_f32 on the rollouts' binned copies (stores[n], page_bound[n]: HOST arrays)
extern "C" int nbp_step_maps_binned_batch_f32(int n, void* const* stores, const int* page_bound, const float* const* points,
                                              const long long* N_cap, const long long* const* N_dev, const float* poses_xyz_host,
                                              const float* bounds_host, const int* n_bounds, const float* band_lo_hi_host, int S, float lo,
                                              float hi, float* const* traj_pts, const int* n_traj_old, const float* traj_fresh_host,
                                              const int* n_traj_fresh, float* out6_all, float* net_in_all, void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(n < 1 || n > MAP_BATCH || !stores || !page_bound || !points || !N_cap || !N_dev || !poses_xyz_host || !bounds_host ||
                  !n_bounds || !band_lo_hi_host || !traj_pts || !n_traj_old || !traj_fresh_host || !n_traj_fresh || !out6_all || !net_in_all,
                  NBP_E_ARG);
    NBP_RETURN_IF(S < 1 || !(hi > lo) || (long long)6 * S * S >= (1ll << 31), NBP_E_SHAPE);
    hipStream_t st = (hipStream_t)stream;
    const size_t SS = (size_t)S * S;
    MapBatch b;
    BinBatch s;
    unsigned max_wg = 1;
    for (int r = 0; r < MAP_BATCH; ++r) {
        const int q = r < n ? r : 0;
        NBP_RETURN_IF(r < n && (!stores[q] || page_bound[q] < 1 || N_cap[q] < 0 || (N_cap[q] > 0 && (!points[q] || ((uintptr_t)points[q] & 3) != 0)) ||
                                !traj_pts[q] || n_bounds[q] < 0 || n_bounds[q] > 8 || n_traj_old[q] < 0 || n_traj_fresh[q] < 0 ||
                                n_traj_fresh[q] > 8), NBP_E_ARG);
        MapItem& it = b.it[r];
        it.p = points[q]; it.N = N_cap[q]; it.n_dev = N_dev[q];
        it.cx = poses_xyz_host[3 * q]; it.cz = poses_xyz_host[3 * q + 2];
        for (int k = 0; k < 8; ++k) it.bd.b[k] = k < n_bounds[q] ? bounds_host[8 * q + k] : 0.f;
        it.bd.n = n_bounds[q];
        it.band_lo = band_lo_hi_host[2 * q]; it.band_hi = band_lo_hi_host[2 * q + 1];
        it.out = out6_all + (size_t)q * 6 * SS;
        it.tr.pts = traj_pts[q]; it.tr.out = net_in_all + (size_t)q * 5 * SS + 4 * SS;
        it.tr.n_old = n_traj_old[q]; it.tr.n_fresh = n_traj_fresh[q];
        for (int i = 0; i < 24; ++i) it.tr.fresh[i] = i < 3 * n_traj_fresh[q] ? traj_fresh_host[24 * q + i] : 0.f;
        b.n_wg[r] = 0;
        s.store[r] = (char*)stores[q]; s.n_page_wg[r] = r < n ? (unsigned)page_bound[q] : 0u;
        if (r < n && s.n_page_wg[r] + BIN_OVF_WGS + 1 > max_wg) max_wg = s.n_page_wg[r] + BIN_OVF_WGS + 1;
    }
    NBP_RETURN_IF(SS % 4 != 0 || ((uintptr_t)out6_all & 15) || ((uintptr_t)net_in_all & 15), NBP_E_SHAPE);
    hipError_t e = hipSuccess;
    bin_append_batch_kernel<<<dim3(BIN_APPEND_WGS, (unsigned)n), 256, 0, st>>>(b, s, (int)SS);       // (also clears maps + trajectory channels)
    int rc = nbp_launch_status();
    if (rc) return rc;
    map_binned_batch_kernel<<<dim3(max_wg, (unsigned)n), 256, 0, st>>>(b, s, S, lo, grid_scale(S, lo, hi));
    rc = nbp_launch_status();
    if (rc) return rc;
    e = hipMemcpy2DAsync(net_in_all, 5 * SS * sizeof(float), out6_all, 6 * SS * sizeof(float), 4 * SS * sizeof(float), (size_t)n,
                         hipMemcpyDeviceToDevice, st);
    return e == hipSuccess ? 0 : (int)e;
}
