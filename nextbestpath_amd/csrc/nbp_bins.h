// nbp_bins.h -- layout of the tile-binned shadow copy of a rollout's cloud (utils.CloudBins) and the wave-cooperative filing of
// points into it.  Shared by csrc/nbp_maps.hip (bin_append_kernel / map_binned_kernel: the step's map build) and csrc/nbp_sim.hip
// (unproject_append_kernel with a store: the launch that appends a frame's points to the cloud files them as well, so that the
// map build behind it is ONE launch).  Include inside the including file's anonymous namespace.
//
//   store = [BinDesc | tile_count[nt] | page_hash[H] | page_info[max_pages] | side_list[BIN_OVF] | pages[max_pages][2048][3]]
// Slot reservation: one atomicAdd per (wave, tile), all of a wave's in flight together; the lane whose slot is the first of a page
// allocates it and publishes (tile, page ordinal) -> page id in an open-addressing hash (one 64-bit CAS: key and id appear together;
// entries are never removed), lanes of other waves probe for the key and wait at the first empty slot of its probe sequence
// (bounded) -- within a wave every allocation is issued before any lane waits, so a waiter never blocks its own allocator.  A tile
// may own any number of pages (a wall the agent lingers at collects > 10^5 points).  What cannot be filed (outside the tile grid,
// the page pool exhausted, a wait that timed out) goes to an index list that every build walks with direct atomics; if that list
// fills up too the store marks itself broken (header word 2) and every later build counts the whole cloud directly:
// slow, never wrong.
#pragma once

typedef float bins_f32x3 __attribute__((ext_vector_type(3), aligned(4)));

constexpr int BIN_PAGE_BITS = 11, BIN_PAGE = 1 << BIN_PAGE_BITS;      // 2048 points = 24 KB per page
constexpr unsigned BIN_OVF = 1u << 16;
struct BinDesc {                // head of the store (device memory, 256 B reserved)
    // error is STICKY: set (never cleared) by a filing launch whose side list overflowed, reset only by nbp_cloud_bins_init.  Every
    // workgroup of a filing launch reads it once at entry and another workgroup may set it during the same launch, so late
    // workgroups may file nothing while n_binned still advances to N: correct only because every later build then counts the whole
    // cloud directly (map_binned_kernel's `broken` branch) -- clearing the flag by any other route would lose those points
    // (tests/test_gpu_maps.py::test_binned_maps_points_that_cannot_be_filed_are_still_counted walks through the break).
    unsigned n_pages, n_overflow, error, ticket;
    long long n_binned;                                   // points of the cloud already filed
    int nx, nz, nt, max_pages;
    float x0, z0, inv_t, tile;
    unsigned hash_mask, pad0;                             // page hash: hash_mask + 1 entries (a power of two >= 2 max_pages)
    unsigned long long off_count, off_table, off_info, off_ovf, off_pages, total_bytes;
};
static_assert(sizeof(BinDesc) <= 256, "BinDesc header");

struct BinView { BinDesc* d; unsigned* count; unsigned long long* table; unsigned* info; unsigned* ovf; float* pages; };
constexpr unsigned long long BIN_EMPTY = ~0ull;           // hash entry: (key << 32) | page id; key = tile | page ordinal << 16
constexpr unsigned BIN_POOL_EXHAUSTED = 0xFFFFFFFEu;      // page id published when the pool has no page left
__device__ __forceinline__ BinView bin_view(char* store) {
    BinDesc* d = reinterpret_cast<BinDesc*>(store);
    return BinView{d, reinterpret_cast<unsigned*>(store + d->off_count), reinterpret_cast<unsigned long long*>(store + d->off_table),
                   reinterpret_cast<unsigned*>(store + d->off_info), reinterpret_cast<unsigned*>(store + d->off_ovf),
                   reinterpret_cast<float*>(store + d->off_pages)};
}

// the launch-constant part of a store's geometry, read once per workgroup
struct BinGeom { int nx, nz, max_pages; float x0, z0, inv_t; unsigned mask; };
__device__ __forceinline__ BinGeom bin_geom(const BinView& v) {
    return BinGeom{v.d->nx, v.d->nz, v.d->max_pages, v.d->x0, v.d->z0, v.d->inv_t, v.d->hash_mask};
}

// Files point p (index i of the cloud) of every ACTIVE lane into its tile's page.  All 64 lanes of the wave call this together
// (ballots); inactive lanes pass active = false.
__device__ __forceinline__ void bin_file_wave(const BinView& v, const BinGeom& g, bool active, bins_f32x3 p, long long i) {
    const int lane = threadIdx.x & 63;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const float fx = (p[0] - g.x0) * g.inv_t, fz = (p[2] - g.z0) * g.inv_t;
    const int tx = (int)floorf(fx), tz = (int)floorf(fz);
    const bool inside = active && fx >= 0.f && fz >= 0.f && tx < g.nx && tz < g.nz;       // NaN fails
    const int t = inside ? tz * g.nx + tx : -1;
    // groups of lanes with the same tile (ballots only), then ONE reserving atomic per group, all in flight together
    int leader = lane;
    unsigned rank = 0, gsize = 0;
    unsigned long long todo = __ballot(inside);
    while (todo) {
        const int l0 = __ffsll((long long)todo) - 1;
        const int t0 = __shfl(t, l0);
        const unsigned long long m = __ballot(inside && t == t0);
        if (inside && t == t0) { leader = l0; rank = (unsigned)__popcll(m & lt); gsize = (unsigned)__popcll(m); }
        todo &= ~m;
    }
    unsigned b = 0;
    if (inside && lane == leader) b = atomicAdd(&v.count[t], gsize);
    b = __shfl(b, leader);
    const unsigned slot = b + rank;
    const unsigned k = slot >> BIN_PAGE_BITS;
    const unsigned key = (unsigned)t | (k << 16);
    const unsigned mask = g.mask;
    unsigned pos = (key * 0x9E3779B1u) >> 7 & mask;
    if (inside && (slot & (BIN_PAGE - 1)) == 0) {         // first slot of a page: allocate it and publish (key -> id)
        const unsigned got = atomicAdd(&v.d->n_pages, 1u);
        unsigned pub = BIN_POOL_EXHAUSTED;                // the pool is exhausted: waiters go to the side list
        if (got < (unsigned)g.max_pages) { v.info[got] = key; pub = got; }
        const unsigned long long e = ((unsigned long long)key << 32) | pub;
        unsigned q = pos;
        for (unsigned probe = 0; probe <= mask; ++probe, q = (q + 1) & mask)
            if (atomicCAS(&v.table[q], BIN_EMPTY, e) == BIN_EMPTY) break;         // (2 max_pages entries: a free one exists)
    }
    int pid = -1;
    if (inside) {
        int budget = 1 << 20;
        while (budget > 0) {
            // (relaxed: only the entry itself is awaited -- an acquire here is an L2 invalidate per probe on gfx950)
            const unsigned long long e = __hip_atomic_load(&v.table[pos], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (e == BIN_EMPTY) { --budget; __builtin_amdgcn_s_sleep(2); continue; }    // not published yet: it lands here or later
            if ((unsigned)(e >> 32) == key) { const unsigned id = (unsigned)e; pid = id == BIN_POOL_EXHAUSTED ? -2 : (int)id; break; }
            pos = (pos + 1) & mask;                       // another key's entry (permanent): move on
        }
    }
    if (pid >= 0) {
        float* dst = v.pages + ((size_t)pid * BIN_PAGE + (slot & (BIN_PAGE - 1))) * 3;
        dst[0] = p[0]; dst[1] = p[1]; dst[2] = p[2];
    } else if (active) {
        const unsigned q = atomicAdd(&v.d->n_overflow, 1u);
        if (q < BIN_OVF) v.ovf[q] = (unsigned)i;
        else v.d->error = 1u;                             // broken from the next build on: the whole cloud is counted directly
    }
}

// clears what the map launch behind a filing launch accumulates into (zero6: the six maps, zero1: the trajectory channel, SS floats
// each; SS % 4 == 0): workgroup `wg` of `n_wg`, 256 threads
__device__ __forceinline__ void bin_clear_maps(float* __restrict__ zero6, float* __restrict__ zero1, int SS, unsigned wg, unsigned n_wg) {
    typedef float v4 __attribute__((ext_vector_type(4)));
    const v4 z = {0.f, 0.f, 0.f, 0.f};
    const int n6 = zero6 ? 6 * SS / 4 : 0, n1 = zero1 ? SS / 4 : 0;
    for (int i = (int)(wg * 256 + threadIdx.x); i < n6 + n1; i += (int)(n_wg * 256))
        reinterpret_cast<v4*>(i < n6 ? zero6 : zero1)[i < n6 ? i : i - n6] = z;
}
