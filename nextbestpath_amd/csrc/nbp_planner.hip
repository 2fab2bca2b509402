// nbp_planner.hip -- planner-side kernels: obstacle-map fusion, batched candidate scoring,
// all-edges Bresenham passability, coverage metric (hash grid).
//
// Replaces the per-item Python loops (one .item() device sync per pixel / per lattice pose) of
//   * next_best_path/testers/nbp_planning.py:166-194 (obstacle fusion), :203-231 (scoring) with
//     macarons/utility/macarons_utils.py:86-100 (check_pixel_values);
//   * next_best_path/utility/long_term_utils.py:277-331 (bresenham_line, line_across_image_pixel);
//   * next_best_path/utility/long_term_utils.py:437-468 (calculate_coverage_percentage: torch.cdist
//     G x 2G + row-min) -- here a uniform grid with cell = threshold, 27-cell neighbourhood,
//     direct-difference distances (no G x M matrix).
// Integer results (cells, masks, counts) are bit-exact against oracle/planner.py.
#include "common.h"
#pragma clang fp contract(off)
#include "nbp_grid.h"

namespace {

__device__ __forceinline__ long long cell_index(float v, float lo, float sc) { return (long long)rintf((v - lo) * sc); }
inline float grid_scale(int S, float lo, float hi) { return (float)((double)S / ((double)hi - (double)lo)); }

// obst = (out2 >= thr); where the full projection is non empty take the height-band projection > 0;
// zero along the trajectory.  fullproj = min(sum of the 5 cloud channels, 1).
__device__ __forceinline__ void fuse_obstacle_body(unsigned bx, unsigned gx, const float* __restrict__ out2, const float* __restrict__ maps6,
                                                            const float* __restrict__ traj, float thr, int SS,
                                                            float* __restrict__ obst, float* __restrict__ fullproj) {
    for (int i = bx * blockDim.x + threadIdx.x; i < SS; i += gx * blockDim.x) {
        const float full = (((maps6[i] + maps6[SS + i]) + maps6[2 * SS + i]) + maps6[3 * SS + i]) + maps6[4 * SS + i];
        float o = out2[i] >= thr ? 1.f : 0.f;
        if (full > 0.f) o = maps6[5 * SS + i] > 0.f ? 1.f : 0.f;
        if (traj[i] > 0.f) o = 0.f;
        obst[i] = o;
        fullproj[i] = full > 1.f ? 1.f : full;
    }
}

// One wave per candidate: the 21 x 21 window test (check_pixel_values) is a ballot over 64 pixels at a time
// instead of up to 441 dependent loads in one lane.
__device__ __forceinline__ void score_candidates_body(unsigned bx, unsigned gx, const float* __restrict__ pos, int P, float cx, float cz,
                                                               const float* __restrict__ out1, int V,
                                                               const float* __restrict__ fullproj, int S, float lo,
                                                               float scV, float scS, const unsigned char* __restrict__ skip,
                                                               unsigned char* __restrict__ valid, int* __restrict__ cell,
                                                               double* __restrict__ score) {
    const int i = (bx * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (i >= P) return;
    if (lane == 0) { valid[i] = 0; cell[2 * i] = 0; cell[2 * i + 1] = 0; score[i] = 0.0; }
    if (skip && skip[i]) return;
    const float v0 = -(pos[3 * i + 2] - cz), v1 = -(pos[3 * i] - cx);
    const long long g0 = cell_index(v0, lo, scV), g1 = cell_index(v1, lo, scV);
    if (g0 < 0 || g0 >= V || g1 < 0 || g1 >= V) return;
    const long long s0 = cell_index(v0, lo, scS), s1 = cell_index(v1, lo, scS);
    // torch indexing semantics: a negative index wraps once (the reference does not bounds check)
    const long long w0 = s0 < 0 ? s0 + S : s0, w1 = s1 < 0 ? s1 + S : s1;
    if (w0 < 0 || w0 >= S || w1 < 0 || w1 >= S) return;
    // check_pixel_values: any pixel == 1 in rows [max(s0-10,0), min(s0+11,S)) x cols likewise (unwrapped index)
    const long long r0 = s0 - 10 > 0 ? s0 - 10 : 0, r1 = s0 + 11 < S ? s0 + 11 : S;
    const long long c0 = s1 - 10 > 0 ? s1 - 10 : 0, c1 = s1 + 11 < S ? s1 + 11 : S;
    const int wc = (int)(c1 - c0), total = (r1 > r0 && wc > 0) ? (int)(r1 - r0) * wc : 0;
    bool any1 = false;
    for (int k0 = 0; k0 < total && !any1; k0 += 64) {
        const int k = k0 + lane;
        const bool one = k < total && fullproj[(r0 + k / wc) * S + c0 + k % wc] == 1.f;
        any1 = __ballot(one) != 0ull;
    }
    if (!any1 || lane != 0) return;
    float best = out1[g0 * V + g1];
    for (int c = 1; c < 8; ++c) best = fmaxf(best, out1[(size_t)c * V * V + g0 * V + g1]);
    const float dens = fullproj[w0 * S + w1];
    valid[i] = 1; cell[2 * i] = (int)g0; cell[2 * i + 1] = (int)g1;
    score[i] = (double)best - 10.0 * (double)dens;
}

__device__ __forceinline__ void edges_blocked_body(unsigned bx, unsigned gx, const float* __restrict__ obst, int S, float lo, float sc,
                                                            float cx, float cz, const float* __restrict__ pos,
                                                            const int* __restrict__ edges, int E,
                                                            unsigned char* __restrict__ blocked) {
    const int e = bx * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const float* a = pos + 3 * (size_t)edges[2 * e];
    const float* b = pos + 3 * (size_t)edges[2 * e + 1];
    long long x0 = cell_index(-(a[2] - cz), lo, sc), y0 = cell_index(-(a[0] - cx), lo, sc);
    const long long x1 = cell_index(-(b[2] - cz), lo, sc), y1 = cell_index(-(b[0] - cx), lo, sc);
    if (x0 < 0 || x0 >= S || y0 < 0 || y0 >= S || x1 < 0 || x1 >= S || y1 < 0 || y1 >= S) { blocked[e] = 1; return; }
    const long long dx = x1 > x0 ? x1 - x0 : x0 - x1, dy = y1 > y0 ? y1 - y0 : y0 - y1;
    const long long sx = x0 < x1 ? 1 : -1, sy = y0 < y1 ? 1 : -1;
    long long err = dx - dy;
    int hits = 0;
    for (int guard = 0; guard < 4 * S; ++guard) {
        if (obst[x0 * S + y0] == 1.f) ++hits;
        if (x0 == x1 && y0 == y1) break;
        const long long e2 = 2 * err;
        if (e2 > -dy) { err -= dy; x0 += sx; }
        if (e2 < dx) { err += dx; y0 += sy; }
    }
    blocked[e] = hits >= 2 ? 1 : 0;
}

// ------------------------------------------------------------------ coverage

// Counting sort of the (sub-sampled) cloud by grid cell, then a 16-lanes-per-GT-point query that
// streams the 9 contiguous cell runs of the 3x3x3 neighbourhood (k is the fastest cell index).
// K1: sample (first k of the index bijection when N > k, else everything), cell id, slot in cell.
__global__ __launch_bounds__(256) void coverage_bin_kernel(const float* __restrict__ pc, const long long* __restrict__ n_dev,
                                                           long long n_host, long long k, unsigned seed, Grid g,
                                                           float* __restrict__ sp, int* __restrict__ cell_of,
                                                           int* __restrict__ slot_of, int* __restrict__ count,
                                                           int* __restrict__ m_out) {
    const long long N = n_dev ? *n_dev : n_host;
    const long long M = N > k ? k : N;
    if (blockIdx.x == 0 && threadIdx.x == 0) *m_out = (int)M;
    const unsigned bits = perm_bits((unsigned)N);
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < M; j += (long long)gridDim.x * blockDim.x) {
        const long long src = N > k ? (long long)perm_index((unsigned)j, (unsigned)N, bits, seed) : j;
        const float x = pc[3 * src], y = pc[3 * src + 1], z = pc[3 * src + 2];
        sp[3 * j] = x; sp[3 * j + 1] = y; sp[3 * j + 2] = z;
        const int c = grid_cell(g, x, y, z, nullptr);
        cell_of[j] = c;
        slot_of[j] = atomicAdd(&count[c], 1);
    }
}

// K3: scatter the sampled points into cell order.
__global__ __launch_bounds__(256) void coverage_scatter_kernel(const float* __restrict__ sp, const int* __restrict__ cell_of,
                                                               const int* __restrict__ slot_of, const int* __restrict__ start,
                                                               const int* __restrict__ m_ptr, float* __restrict__ sorted) {
    const int M = *m_ptr;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < M; j += gridDim.x * blockDim.x) {
        const int d = start[cell_of[j]] + slot_of[j];
        sorted[3 * d] = sp[3 * j]; sorted[3 * d + 1] = sp[3 * j + 1]; sorted[3 * d + 2] = sp[3 * j + 2];
    }
}

// K4: 16 lanes per GT point.  The nine (x, y) cell columns around the point (each a contiguous run of up to three
// z cells in the sorted array) are looked up by nine lanes at once; the runs are then walked 16 points at a time,
// centre column first (a covered point usually finds its neighbour there and the group leaves after one round).
__global__ __launch_bounds__(256) void coverage_query_kernel(const float* __restrict__ gt, int G, Grid g, float thr,
                                                             const float* __restrict__ sorted, const int* __restrict__ start,
                                                             int* __restrict__ count) {
    const int sub = threadIdx.x & 15;
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int grp_shift = threadIdx.x & 48;            // first lane of this 16-lane group inside the wave
    bool found = false;
    float x = 0.f, y = 0.f, z = 0.f;
    int lo = 0, hi = 0;
    if (i < G) {
        x = gt[3 * i]; y = gt[3 * i + 1]; z = gt[3 * i + 2];
        int c[3];
        grid_cell(g, x, y, z, c);
        if (sub < 9) {
            // column order: centre, then the 4 edge neighbours, then the 4 corners
            const int ox[9] = {0, -1, 1, 0, 0, -1, -1, 1, 1}, oy[9] = {0, 0, 0, -1, 1, -1, 1, -1, 1};
            const int a = c[0] + ox[sub], b = c[1] + oy[sub];
            if (a >= 0 && a < g.n[0] && b >= 0 && b < g.n[1]) {
                const int d0 = max(c[2] - 1, 0), d1 = min(c[2] + 1, g.n[2] - 1);
                const int base = (a * g.n[1] + b) * g.n[2];
                lo = start[base + d0]; hi = start[base + d1 + 1];
            }
        }
    }
    for (int col = 0; col < 9; ++col) {
        const int clo = __shfl(lo, grp_shift + col), chi = __shfl(hi, grp_shift + col);
        for (int j0 = clo; j0 < chi && !found; j0 += 16) {
            const int j = j0 + sub;
            bool hit = false;
            if (j < chi) {
                const float ex = x - sorted[3 * j], ey = y - sorted[3 * j + 1], ez = z - sorted[3 * j + 2];
                hit = sqrtf((ex * ex + ey * ey) + ez * ez) < thr;
            }
            const unsigned long long bal = __ballot(hit);
            found = ((bal >> grp_shift) & 0xffffull) != 0;
        }
    }
    const unsigned long long fb = __ballot(found && sub == 0 && i < G);
    if ((threadIdx.x & 63) == 0 && fb) atomicAdd(count, __popcll(fb));
}


// ---- planned coverage: the GT cloud is static for a whole rollout, the reconstruction changes every step.  The GT points
// are sorted into the grid ONCE (nbp_coverage_plan_build_f32); a step then runs ONE kernel over the sampled cloud points:
// four lanes per cloud point walk the nine GT cell columns around it and stamp every GT point closer than the threshold
// with the call's epoch (plain stores; a second tiny kernel tallies the stamps).  No per-step sort, no clears.
__global__ __launch_bounds__(256) void points_bin_kernel(const float* __restrict__ pts, int n, Grid g, int* __restrict__ cell_of,
                                                         int* __restrict__ slot_of, int* __restrict__ count) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int c = grid_cell(g, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], nullptr);
        cell_of[i] = c;
        slot_of[i] = atomicAdd(&count[c], 1);
    }
}

__global__ __launch_bounds__(256) void points_scatter4_kernel(const float* __restrict__ pts, int n, const int* __restrict__ cell_of,
                                                              const int* __restrict__ slot_of, const int* __restrict__ start,
                                                              float4* __restrict__ sorted, unsigned* __restrict__ stamp) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        sorted[start[cell_of[i]] + slot_of[i]] = make_float4(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], 0.f);
        stamp[i] = 0u;
    }
}

// The columns around a cloud point are culled by their distance to it: a column whose (x, y) box is farther than the threshold is
// skipped, the others are walked only over the z cells within the remaining budget (a random point meets 20.6 of its 27 cells).
// The test is in cell units against PLAN_R cells = 1.001 x the threshold: the 0.1 % slack covers the fp32 rounding of
// (x - lo) * inv on both sides (~1e-5 cells each), so no GT point within the threshold is ever in a skipped cell.
// PLAN_R = cells to a threshold.  2 (cells of half the threshold, 5 x 5 columns, 2.6 x fewer pair tests) was measured: the same
// 30 us stand-alone and -2 % in the lock-step -- the launch is bound by its dependent loads (the random gather of the sampled
// cloud points, two run bounds per column), not by the pair tests (profiles/r05/rejected_experiments.txt).
constexpr int PLAN_R = 1;
template <int LANES, int UNROLL>
__device__ __forceinline__ void coverage_mark_body(unsigned bx, unsigned by, unsigned gx, unsigned gy, const float* __restrict__ pc, const long long* __restrict__ n_dev,
                                                            long long n_host, long long k, unsigned seed, Grid g, float d2max,
                                                            const float4* __restrict__ gt_sorted, const int* __restrict__ gt_start,
                                                            unsigned* __restrict__ stamp, unsigned epoch,
                                                            int* __restrict__ m_out) {
    // d2max = the largest float whose square root is below the threshold (sq_below, host): d2 <= d2max is the reference's
    // cdist(...) < threshold decision exactly (sqrtf is monotone and correctly rounded on both sides) without the ~12
    // instructions of a correctly rounded square root per point pair -- the kernel is bound by those pair tests
    constexpr int R = PLAN_R, D = 2 * R + 1;
    const long long N = n_dev ? *n_dev : n_host;
    const long long M = N > k ? k : N;
    if (bx == 0 && threadIdx.x == 0) *m_out = (int)M;
    const unsigned bits = perm_bits((unsigned)N);
    const int sub = threadIdx.x % LANES;
    for (long long j = ((long long)bx * blockDim.x + threadIdx.x) / LANES; j < M;
         j += ((long long)gx * blockDim.x) / LANES) {
        const long long src = N > k ? (long long)perm_index((unsigned)j, (unsigned)N, bits, seed) : j;
        const float x = pc[3 * src], y = pc[3 * src + 1], z = pc[3 * src + 2];
        // unclamped cell: a cloud point outside the grid (= the GT box grown by thr) is farther than thr from every GT point
        const float ux = (x - g.lo[0]) * g.inv, uy = (y - g.lo[1]) * g.inv, uz = (z - g.lo[2]) * g.inv;
        const float fi = floorf(ux), fj = floorf(uy), fk = floorf(uz);
        if (!(fi >= (float)-R && fi <= (float)(g.n[0] - 1 + R) && fj >= (float)-R && fj <= (float)(g.n[1] - 1 + R) && fk >= (float)-R &&
              fk <= (float)(g.n[2] - 1 + R)))
            continue;
        const int ci = (int)fi, cj = (int)fj, ck = (int)fk;
        const float ox = ux - fi, oy = uy - fj, oz = uz - fk;          // position inside the cell, [0, 1]
        for (int col = sub; col < D * D; col += LANES) {
            const int da = col / D - R, db = col % D - R;
            const int a = ci + da, b = cj + db;
            if (a < 0 || a >= g.n[0] || b < 0 || b >= g.n[1]) continue;
            const float dx = da > 0 ? (float)da - ox : (da < 0 ? ox - (float)(da + 1) : 0.f);
            const float dy = db > 0 ? (float)db - oy : (db < 0 ? oy - (float)(db + 1) : 0.f);
            const float rem = (float)(R * R) - (dx * dx + dy * dy);
            if (rem < 0.f) continue;
            int k0 = 0, k1 = 0;
#pragma unroll
            for (int dk = 1; dk <= R; ++dk) {
                const float below = oz + (float)(dk - 1), above = (float)dk - oz;
                if (below * below <= rem) k0 = -dk;
                if (above * above <= rem) k1 = dk;
            }
            const int d0 = max(ck + k0, 0), d1 = min(ck + k1, g.n[2] - 1);
            if (d0 > d1) continue;
            const int base = (a * g.n[1] + b) * g.n[2];
            const int hi = gt_start[base + d1 + 1];
            int q = gt_start[base + d0];
            // the run of a column is walked UNROLL points at a time: their stamps and positions are independent loads in
            // flight together (one point per iteration was a chain of ~100 dependent round trips per lane)
            for (; q + UNROLL <= hi; q += UNROLL) {
                unsigned sp[UNROLL];
                float4 t[UNROLL];
                // stamps first, positions only of the points no one has stamped yet in this launch: a covered GT point is near
                // many of the 100 k sampled cloud points and all but the first find it stamped -- a 4-byte load instead of 20 and
                // no distance test (lock-step +1.0 %, profiles/r04/coverage_lazy_positions.txt; a stale "unstamped" read from
                // another XCD's L2 only repeats a test)
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) sp[u] = stamp[q + u];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) t[u] = sp[u] != epoch ? gt_sorted[q + u] : make_float4(3e38f, 3e38f, 3e38f, 0.f);
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    const float ex = t[u].x - x, ey = t[u].y - y, ez = t[u].z - z;
                    if (sp[u] != epoch && (ex * ex + ey * ey) + ez * ez <= d2max) stamp[q + u] = epoch;
                }
            }
            for (; q < hi; ++q) {
                if (stamp[q] == epoch) continue;
                const float4 t = gt_sorted[q];
                const float ex = t.x - x, ey = t.y - y, ez = t.z - z;
                if ((ex * ex + ey * ey) + ez * ez <= d2max) stamp[q] = epoch;          // plain store: every writer writes the
            }                                                                          // same value (device-scope atomics run
        }                                                                              // at ~5 G/s on this part: 30x slower)
    }
}

__device__ __forceinline__ void coverage_tally_body(unsigned bx, unsigned by, unsigned gx, unsigned gy, const unsigned* __restrict__ stamp, int G, unsigned epoch,
                                                             int* __restrict__ count) {
    int mine = 0;
    for (int q = bx * blockDim.x + threadIdx.x; q < G; q += gx * blockDim.x) mine += stamp[q] == epoch ? 1 : 0;
    for (int o = 32; o; o >>= 1) mine += __shfl_xor(mine, o);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(count, mine);
}

// ---- launch forms of the replanning kernels: one rollout, or every replanning rollout of a lock-step group (blockIdx.y)
__global__ __launch_bounds__(256) void fuse_obstacle_kernel(const float* __restrict__ out2, const float* __restrict__ maps6,
                                                            const float* __restrict__ traj, float thr, int SS, float* __restrict__ obst,
                                                            float* __restrict__ fullproj) {
    fuse_obstacle_body(blockIdx.x, gridDim.x, out2, maps6, traj, thr, SS, obst, fullproj);
}
__global__ __launch_bounds__(256) void score_candidates_kernel(const float* __restrict__ pos, int P, float cx, float cz,
                                                               const float* __restrict__ out1, int V, const float* __restrict__ fullproj,
                                                               int S, float lo, float scV, float scS, const unsigned char* __restrict__ skip,
                                                               unsigned char* __restrict__ valid, int* __restrict__ cell,
                                                               double* __restrict__ score) {
    score_candidates_body(blockIdx.x, gridDim.x, pos, P, cx, cz, out1, V, fullproj, S, lo, scV, scS, skip, valid, cell, score);
}
__global__ __launch_bounds__(256) void edges_blocked_kernel(const float* __restrict__ obst, int S, float lo, float sc, float cx, float cz,
                                                            const float* __restrict__ pos, const int* __restrict__ edges, int E,
                                                            unsigned char* __restrict__ blocked) {
    edges_blocked_body(blockIdx.x, gridDim.x, obst, S, lo, sc, cx, cz, pos, edges, E, blocked);
}
constexpr int PLAN_BATCH = 16;
struct PlanItem {
    const float* out2; const float* maps6; const float* traj; float* obst; float* fullproj; const float* pos; const float* out1;
    const unsigned char* skip; unsigned char* valid; int* cell; double* score; const int* edges; unsigned char* blocked;
    int P, E; float cx, cz; unsigned g_score, g_edges;
};
struct PlanBatch { PlanItem it[PLAN_BATCH]; };
__global__ __launch_bounds__(256) void fuse_obstacle_batch_kernel(PlanBatch b, float thr, int SS) {
    const PlanItem& a = b.it[blockIdx.y];
    fuse_obstacle_body(blockIdx.x, gridDim.x, a.out2, a.maps6, a.traj, thr, SS, a.obst, a.fullproj);
}
// candidate scoring (the first g_score workgroups of an item) and the all-edges mask (the next g_edges) in one launch
__global__ __launch_bounds__(256) void score_edges_batch_kernel(PlanBatch b, int V, int S, float lo, float scV, float scS) {
    const PlanItem& a = b.it[blockIdx.y];
    if (blockIdx.x < a.g_score)
        score_candidates_body(blockIdx.x, a.g_score, a.pos, a.P, a.cx, a.cz, a.out1, V, a.fullproj, S, lo, scV, scS, a.skip, a.valid, a.cell,
                              a.score);
    else if (blockIdx.x < a.g_score + a.g_edges)
        edges_blocked_body(blockIdx.x - a.g_score, a.g_edges, a.obst, S, lo, scS, a.cx, a.cz, a.pos, a.edges, a.E, a.blocked);
}

// ---- launch forms of the planned coverage: one rollout, or the rollouts of a lock-step group in one launch (blockIdx.y)
template <int LANES, int UNROLL>
__global__ __launch_bounds__(256) void coverage_mark_kernel(const float* __restrict__ pc, const long long* __restrict__ n_dev,
                                                            long long n_host, long long k, unsigned seed, Grid g, float d2max,
                                                            const float4* __restrict__ gt_sorted, const int* __restrict__ gt_start,
                                                            unsigned* __restrict__ stamp, unsigned epoch, int* __restrict__ m_out) {
    coverage_mark_body<LANES, UNROLL>(blockIdx.x, 0, gridDim.x, 1, pc, n_dev, n_host, k, seed, g, d2max, gt_sorted, gt_start, stamp, epoch, m_out);
}
__global__ __launch_bounds__(256) void coverage_tally_kernel(const unsigned* __restrict__ stamp, int G, unsigned epoch, int* __restrict__ count) {
    coverage_tally_body(blockIdx.x, 0, gridDim.x, 1, stamp, G, epoch, count);
}
constexpr int COV_BATCH = 16;
struct CovItem {
    const float* pc; const long long* n_dev; long long n_host, k; const float4* gt_sorted; const int* gt_start; unsigned* stamp;
    int* count; int* m_out; Grid g; float d2max; unsigned seed, epoch; int G; unsigned gx_mark, gx_tally;
};
struct CovBatch { CovItem it[COV_BATCH]; };
template <int LANES, int UNROLL>
__global__ __launch_bounds__(256) void coverage_mark_batch_kernel(CovBatch b) {
    const CovItem& a = b.it[blockIdx.y];
    if (blockIdx.x >= a.gx_mark) return;
    coverage_mark_body<LANES, UNROLL>(blockIdx.x, 0, a.gx_mark, 1, a.pc, a.n_dev, a.n_host, a.k, a.seed, a.g, a.d2max, a.gt_sorted, a.gt_start,
                                      a.stamp, a.epoch, a.m_out);
}
__global__ __launch_bounds__(256) void coverage_tally_batch_kernel(CovBatch b) {
    const CovItem& a = b.it[blockIdx.y];
    if (blockIdx.x >= a.gx_tally) return;
    coverage_tally_body(blockIdx.x, 0, a.gx_tally, 1, a.stamp, a.G, a.epoch, a.count);
}

}  // namespace

extern "C" int nbp_fuse_obstacle_f32(const float* out2, const float* maps6, const float* traj, float threshold, int S,
                                     float* obst, float* fullproj, void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(!out2 || !maps6 || !traj || !obst || !fullproj || S < 1, NBP_E_ARG);
    fuse_obstacle_kernel<<<nbp_ew_grid((long long)S * S, 256), 256, 0, (hipStream_t)stream>>>(out2, maps6, traj, threshold,
                                                                                              S * S, obst, fullproj);
    return nbp_launch_status();
}

extern "C" int nbp_score_candidates_f32(const float* pos3, int P, float cx, float cz, const float* out1, int V,
                                        const float* fullproj, int S, float lo, float hi, const unsigned char* skip_or_null,
                                        unsigned char* valid, int* cell2, double* score, void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(!pos3 || !out1 || !fullproj || !valid || !cell2 || !score || P < 1 || V < 1 || S < 1 || !(hi > lo),
                  NBP_E_ARG);
    score_candidates_kernel<<<(unsigned)nbp_cdiv((long long)P * 64, 256), 256, 0, (hipStream_t)stream>>>(
        pos3, P, cx, cz, out1, V, fullproj, S, lo, grid_scale(V, lo, hi), grid_scale(S, lo, hi), skip_or_null, valid, cell2,
        score);
    return nbp_launch_status();
}

extern "C" int nbp_edges_blocked_u8(const float* obst, int S, float lo, float hi, float cx, float cz, const float* pos3,
                                    const int* edges2, int E, unsigned char* blocked, void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(!obst || !pos3 || !edges2 || !blocked || S < 1 || E < 1 || !(hi > lo), NBP_E_ARG);
    edges_blocked_kernel<<<(unsigned)nbp_cdiv(E, 256), 256, 0, (hipStream_t)stream>>>(obst, S, lo, grid_scale(S, lo, hi), cx,
                                                                                      cz, pos3, edges2, E, blocked);
    return nbp_launch_status();
}

static int coverage_grid(const float* bbox_lo, const float* bbox_hi, float thr, Grid* g, size_t* ncell) {
    size_t n = 1;
    for (int a = 0; a < 3; ++a) {
        g->lo[a] = bbox_lo[a] - thr;
        const double ext = (double)bbox_hi[a] + thr - g->lo[a];
        if (!(ext > 0)) return NBP_E_ARG;
        g->n[a] = (int)(ext / thr) + 1;
        n *= (size_t)g->n[a];
    }
    g->inv = (float)(1.0 / ((double)thr * 1.001));   // cells a little wider than thr: fp32 rounding of (x - lo) * inv can
                                                     // never put two points within thr more than one cell apart
    if (n > (size_t)1 << 28) return NBP_E_SHAPE;
    *ncell = n;
    return 0;
}

static size_t al256(size_t b) { return (b + 255) / 256 * 256; }

// the grid of a coverage PLAN: cells of thr / PLAN_R (coverage_mark_body), over the GT box grown by thr
static int coverage_plan_grid(const float* bbox_lo, const float* bbox_hi, float thr, Grid* g, size_t* ncell) {
    size_t n = 1;
    const double w = (double)thr / PLAN_R;
    for (int a = 0; a < 3; ++a) {
        g->lo[a] = bbox_lo[a] - thr;
        const double ext = (double)bbox_hi[a] + thr - g->lo[a];
        if (!(ext > 0)) return NBP_E_ARG;
        g->n[a] = (int)(ext / w) + 1;
        n *= (size_t)g->n[a];
    }
    g->inv = (float)(1.0 / (w * 1.001));             // PLAN_R cells = 1.001 thr
    if (n > (size_t)1 << 28) return NBP_E_SHAPE;
    *ncell = n;
    return 0;
}

extern "C" size_t nbp_coverage_workspace_bytes(const float* bbox_lo_host, const float* bbox_hi_host, float threshold,
                                               long long sample_k) {
    Grid g; size_t ncell;
    if (!bbox_lo_host || !bbox_hi_host || !(threshold > 0) || sample_k < 1) return 0;
    if (coverage_grid(bbox_lo_host, bbox_hi_host, threshold, &g, &ncell)) return 0;
    return al256(ncell * 4) + al256((ncell + 1) * 4) + 2 * al256((size_t)sample_k * 4) +
           2 * al256((size_t)sample_k * 12) + al256((ncell / SCAN_TILE + 1) * 4) + 512;
}

extern "C" int nbp_coverage_count_f32(const float* gt3, int G, const float* pc3, long long N, const long long* N_dev_or_null,
                                      long long sample_k, unsigned seed, float threshold, const float* bbox_lo_host,
                                      const float* bbox_hi_host, int* count_out, int* m_out, void* ws, size_t ws_bytes,
                                      void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(!gt3 || !pc3 || !count_out || !m_out || !ws || !bbox_lo_host || !bbox_hi_host, NBP_E_ARG);
    NBP_RETURN_IF(G < 1 || N < 0 || sample_k < 1 || !(threshold > 0) || N > 0xffffffffll, NBP_E_ARG);
    Grid g; size_t ncell;
    int rc = coverage_grid(bbox_lo_host, bbox_hi_host, threshold, &g, &ncell);
    if (rc) return rc;
    NBP_RETURN_IF(ws_bytes < nbp_coverage_workspace_bytes(bbox_lo_host, bbox_hi_host, threshold, sample_k), NBP_E_WS);
    hipStream_t st = (hipStream_t)stream;
    char* p = (char*)(((uintptr_t)ws + 255) / 256 * 256);
    int* count = (int*)p; p += al256(ncell * 4);
    int* start = (int*)p; p += al256((ncell + 1) * 4);
    int* cell_of = (int*)p; p += al256((size_t)sample_k * 4);
    int* slot_of = (int*)p; p += al256((size_t)sample_k * 4);
    float* sp = (float*)p; p += al256((size_t)sample_k * 12);
    float* sorted = (float*)p; p += al256((size_t)sample_k * 12);
    int* tsum = (int*)p;
    hipError_t e = hipMemsetAsync(count, 0, ncell * 4, st);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(count_out, 0, sizeof(int), st);
    if (e != hipSuccess) return (int)e;
    const long long work = N_dev_or_null ? sample_k : (N < sample_k ? N : sample_k);
    const int grid = nbp_ew_grid(work > 0 ? work : 1, 256);
    coverage_bin_kernel<<<grid, 256, 0, st>>>(pc3, N_dev_or_null, N, sample_k, seed, g, sp, cell_of, slot_of, count, m_out);
    if ((rc = nbp_launch_status())) return rc;
    if ((rc = grid_exclusive_scan(count, (long long)ncell, tsum, start, st))) return rc;
    coverage_scatter_kernel<<<grid, 256, 0, st>>>(sp, cell_of, slot_of, start, m_out, sorted);
    if ((rc = nbp_launch_status())) return rc;
    coverage_query_kernel<<<(unsigned)nbp_cdiv((long long)G * 16, 256), 256, 0, st>>>(gt3, G, g, threshold, sorted, start,
                                                                                    count_out);
    return nbp_launch_status();
}

// ---- planned coverage (one GT grid per rollout)
// Largest float x with sqrtf(x) < thr (thr > 0 finite): the squared-distance form of `distance < thr`.
static float sq_below(float thr) {
    float x = thr * thr;
    while (sqrtf(x) >= thr) x = nextafterf(x, 0.f);
    while (sqrtf(nextafterf(x, INFINITY)) < thr) x = nextafterf(x, INFINITY);
    return x;
}

static void plan_carve(void* plan, size_t ncell, int G, int** start, float4** sorted, unsigned** stamp) {
    char* p = (char*)(((uintptr_t)plan + 255) / 256 * 256);
    *start = (int*)p; p += al256((ncell + 1) * 4);
    *sorted = (float4*)p; p += al256((size_t)G * 16);
    *stamp = (unsigned*)p;
}

extern "C" size_t nbp_coverage_plan_bytes(const float* bbox_lo_host, const float* bbox_hi_host, float threshold, int G) {
    Grid g; size_t ncell;
    if (!bbox_lo_host || !bbox_hi_host || !(threshold > 0) || G < 1) return 0;
    if (coverage_plan_grid(bbox_lo_host, bbox_hi_host, threshold, &g, &ncell)) return 0;
    return 256 + al256((ncell + 1) * 4) + al256((size_t)G * 16) + al256((size_t)G * 4);
}

extern "C" size_t nbp_coverage_plan_workspace_bytes(const float* bbox_lo_host, const float* bbox_hi_host, float threshold,
                                                    int G) {
    Grid g; size_t ncell;
    if (!bbox_lo_host || !bbox_hi_host || !(threshold > 0) || G < 1) return 0;
    if (coverage_plan_grid(bbox_lo_host, bbox_hi_host, threshold, &g, &ncell)) return 0;
    return 256 + al256(ncell * 4) + al256((ncell / SCAN_TILE + 1) * 4) + 2 * al256((size_t)G * 4);
}

extern "C" int nbp_coverage_plan_build_f32(const float* gt3, int G, float threshold, const float* bbox_lo_host,
                                           const float* bbox_hi_host, void* plan, size_t plan_bytes, void* ws, size_t ws_bytes,
                                           void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(!gt3 || !plan || !ws || !bbox_lo_host || !bbox_hi_host || G < 1 || !(threshold > 0), NBP_E_ARG);
    Grid g; size_t ncell;
    int rc = coverage_plan_grid(bbox_lo_host, bbox_hi_host, threshold, &g, &ncell);
    if (rc) return rc;
    NBP_RETURN_IF(plan_bytes < nbp_coverage_plan_bytes(bbox_lo_host, bbox_hi_host, threshold, G), NBP_E_WS);
    NBP_RETURN_IF(ws_bytes < nbp_coverage_plan_workspace_bytes(bbox_lo_host, bbox_hi_host, threshold, G), NBP_E_WS);
    hipStream_t st = (hipStream_t)stream;
    int* start; float4* sorted; unsigned* stamp;
    plan_carve(plan, ncell, G, &start, &sorted, &stamp);
    char* p = (char*)(((uintptr_t)ws + 255) / 256 * 256);
    int* count = (int*)p; p += al256(ncell * 4);
    int* tsum = (int*)p; p += al256((ncell / SCAN_TILE + 1) * 4);
    int* cell_of = (int*)p; p += al256((size_t)G * 4);
    int* slot_of = (int*)p;
    hipError_t e = hipMemsetAsync(count, 0, ncell * 4, st);
    if (e != hipSuccess) return (int)e;
    const int grid = nbp_ew_grid(G, 256);
    points_bin_kernel<<<grid, 256, 0, st>>>(gt3, G, g, cell_of, slot_of, count);
    if ((rc = nbp_launch_status())) return rc;
    if ((rc = grid_exclusive_scan(count, (long long)ncell, tsum, start, st))) return rc;
    points_scatter4_kernel<<<grid, 256, 0, st>>>(gt3, G, cell_of, slot_of, start, sorted, stamp);
    return nbp_launch_status();
}

extern "C" int nbp_coverage_count_planned_f32(void* plan, int G, float threshold, const float* bbox_lo_host,
                                              const float* bbox_hi_host, const float* pc3, long long N,
                                              const long long* N_dev_or_null, long long sample_k, unsigned seed,
                                              unsigned epoch, int* count_accum, int* m_out, void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(!plan || !pc3 || !count_accum || !m_out || !bbox_lo_host || !bbox_hi_host, NBP_E_ARG);
    NBP_RETURN_IF(G < 1 || N < 0 || sample_k < 1 || !(threshold > 0) || N > 0xffffffffll || epoch == 0, NBP_E_ARG);
    Grid g; size_t ncell;
    int rc = coverage_plan_grid(bbox_lo_host, bbox_hi_host, threshold, &g, &ncell);
    if (rc) return rc;
    int* start; float4* sorted; unsigned* stamp;
    plan_carve(plan, ncell, G, &start, &sorted, &stamp);
    const long long work = N_dev_or_null ? sample_k : (N < sample_k ? N : sample_k);
    // 4 lanes per point, runs walked 4 GT points at a time: 34 us with the tally against 39 for one point per iteration;
    // 8 or 16 lanes per point or 8 points per iteration measure the same (the kernel is then bound by the ~50 M
    // point-pair tests, not by the chains)
    coverage_mark_kernel<4, 4><<<nbp_ew_grid((work > 0 ? work : 1) * 4, 256), 256, 0, (hipStream_t)stream>>>(
        pc3, N_dev_or_null, N, sample_k, seed, g, sq_below(threshold), sorted, start, stamp, epoch, m_out);
    if ((rc = nbp_launch_status())) return rc;
    coverage_tally_kernel<<<(unsigned)(G < 16384 ? 1 : 16), 256, 0, (hipStream_t)stream>>>(stamp, G, epoch, count_accum);
    return nbp_launch_status();
}

// nbp_coverage_count_planned_f32 for n <= 16 rollouts in TWO launches (mark, tally) instead of 2 n: arrays of n entries, HOST
// memory (plans / pc3 / N_dev / count_accum / m_out: device pointers; bbox_lo / bbox_hi [n][3]).  Identical results.
extern "C" int nbp_coverage_count_planned_batch_f32(int n, void* const* plans, const int* G, float threshold, const float* bbox_lo_host,
                                                    const float* bbox_hi_host, const float* const* pc3, const long long* N,
                                                    const long long* const* N_dev, const long long* sample_k, const unsigned* seed,
                                                    const unsigned* epoch, int* const* count_accum, int* const* m_out, void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(n < 1 || n > COV_BATCH || !plans || !G || !bbox_lo_host || !bbox_hi_host || !pc3 || !N || !N_dev || !sample_k ||
                  !seed || !epoch || !count_accum || !m_out || !(threshold > 0), NBP_E_ARG);
    CovBatch b;
    unsigned gmark = 1, gtally = 1;
    const float d2max = sq_below(threshold);
    for (int r = 0; r < COV_BATCH; ++r) {
        const int q = r < n ? r : 0;
        NBP_RETURN_IF(!plans[q] || !pc3[q] || !count_accum[q] || !m_out[q] || G[q] < 1 || N[q] < 0 || sample_k[q] < 1 ||
                      N[q] > 0xffffffffll || epoch[q] == 0, NBP_E_ARG);
        CovItem& a = b.it[r];
        size_t ncell;
        const int rc = coverage_plan_grid(bbox_lo_host + 3 * q, bbox_hi_host + 3 * q, threshold, &a.g, &ncell);
        if (rc) return rc;
        int* start; float4* sorted; unsigned* stamp;
        plan_carve(plans[q], ncell, G[q], &start, &sorted, &stamp);
        a.pc = pc3[q]; a.n_dev = N_dev[q]; a.n_host = N[q]; a.k = sample_k[q]; a.gt_sorted = sorted; a.gt_start = start; a.stamp = stamp;
        a.count = count_accum[q]; a.m_out = m_out[q]; a.d2max = d2max; a.seed = seed[q]; a.epoch = epoch[q]; a.G = G[q];
        const long long work = N_dev[q] ? sample_k[q] : (N[q] < sample_k[q] ? N[q] : sample_k[q]);
        a.gx_mark = r < n ? (unsigned)nbp_ew_grid((work > 0 ? work : 1) * 4, 256) : 0;
        a.gx_tally = r < n ? (unsigned)(G[q] < 16384 ? 1 : 16) : 0;
        if (a.gx_mark > gmark) gmark = a.gx_mark;
        if (a.gx_tally > gtally) gtally = a.gx_tally;
    }
    coverage_mark_batch_kernel<4, 4><<<dim3(gmark, (unsigned)n), 256, 0, (hipStream_t)stream>>>(b);
    int rc = nbp_launch_status();
    if (rc) return rc;
    coverage_tally_batch_kernel<<<dim3(gtally, (unsigned)n), 256, 0, (hipStream_t)stream>>>(b);
    return nbp_launch_status();
}

// The GPU half of a replan (nbp_fuse_obstacle_f32 + nbp_score_candidates_f32 + nbp_edges_blocked_u8, nbp_planning.py:166-233,
// long_term_utils.py:277-331) for the n <= 16 replanning rollouts of a lock-step group in TWO launches.  HOST arrays of n
// entries; poses_xz_host [n][2] = (cx, cz); skip entries may be NULL.  Identical results to the three single calls.
extern "C" int nbp_replan_batch_f32(int n, const float* const* out2, const float* const* maps6, const float* const* traj, float threshold,
                                    int S, float* const* obst, float* const* fullproj, const float* const* pos3, const int* P,
                                    const float* poses_xz_host, const float* const* out1, int V, float lo, float hi,
                                    const unsigned char* const* skip, unsigned char* const* valid, int* const* cell2,
                                    double* const* score, const int* const* edges2, const int* E, unsigned char* const* blocked,
                                    void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(n < 1 || n > PLAN_BATCH || !out2 || !maps6 || !traj || !obst || !fullproj || !pos3 || !P || !poses_xz_host || !out1 ||
                  !skip || !valid || !cell2 || !score || !edges2 || !E || !blocked || S < 1 || V < 1 || !(hi > lo), NBP_E_ARG);
    PlanBatch b;
    unsigned gmax = 1;
    for (int r = 0; r < PLAN_BATCH; ++r) {
        const int q = r < n ? r : 0;
        NBP_RETURN_IF(!out2[q] || !maps6[q] || !traj[q] || !obst[q] || !fullproj[q] || !pos3[q] || !out1[q] || !valid[q] || !cell2[q] ||
                      !score[q] || !edges2[q] || !blocked[q] || P[q] < 1 || E[q] < 1, NBP_E_ARG);
        PlanItem& a = b.it[r];
        a.out2 = out2[q]; a.maps6 = maps6[q]; a.traj = traj[q]; a.obst = obst[q]; a.fullproj = fullproj[q]; a.pos = pos3[q];
        a.out1 = out1[q]; a.skip = skip[q]; a.valid = valid[q]; a.cell = cell2[q]; a.score = score[q]; a.edges = edges2[q];
        a.blocked = blocked[q]; a.P = P[q]; a.E = E[q]; a.cx = poses_xz_host[2 * q]; a.cz = poses_xz_host[2 * q + 1];
        a.g_score = r < n ? (unsigned)nbp_cdiv((long long)P[q] * 64, 256) : 0;
        a.g_edges = r < n ? (unsigned)nbp_cdiv(E[q], 256) : 0;
        if (a.g_score + a.g_edges > gmax) gmax = a.g_score + a.g_edges;
    }
    hipStream_t st = (hipStream_t)stream;
    fuse_obstacle_batch_kernel<<<dim3((unsigned)nbp_ew_grid((long long)S * S, 256), (unsigned)n), 256, 0, st>>>(b, threshold, S * S);
    int rc = nbp_launch_status();
    if (rc) return rc;
    score_edges_batch_kernel<<<dim3(gmax, (unsigned)n), 256, 0, st>>>(b, V, S, lo, grid_scale(V, lo, hi), grid_scale(S, lo, hi));
    return nbp_launch_status();
}
