// nbp_scene.hip -- device-resident Scene / Cell point store and the per-cell surface coverage metric.
//
// Replaces (macarons/utility/macarons_utils.py): Cell.fill :3000-3028 (strict cell box, fp64 cdist thinning at the cell
// resolution against the points already stored, random capacity cap), Scene.get_cells_for_each_pt :3148-3162 with
// floor_divide (macarons/utility/utils.py:113-117), Scene.fill_cells :3177-3187, Scene.return_entire_pt_cloud
// :3217-3234 and Scene.scene_coverage :3512-3539 (per cell: GT points whose nearest recovered point OF THE SAME CELL is
// closer than epsilon, fp64 distances).
//
// Layout: store_pts [n_cells][capacity][3] fp32 (cell order = cartesian product order i_l, i_w, i_h), store_count
// [n_cells] int32; both live in HBM and are only touched by these kernels.  The reference's dict of per-cell tensors with a
// G x M distance matrix per cell becomes: one uniform point grid over the stored points (edge >= radius, counting sort),
// one thread per query point walking its 27 grid cells with fp64 distances.  The capacity cap keeps the first `capacity`
// entries of the seeded index bijection over [stored | new] (stands in for torch.randperm(len)[:capacity], :3021).
#include "common.h"
#pragma clang fp contract(off)
#include "nbp_grid.h"

namespace {

struct SceneGeom { float x_min[3]; float x_max[3]; int g[3]; float step[3]; };

static SceneGeom make_geom(const float* box6, const int* grid3) {
    SceneGeom s;
    for (int a = 0; a < 3; ++a) {
        s.x_min[a] = box6[a]; s.x_max[a] = box6[3 + a]; s.g[a] = grid3[a];
        s.step[a] = (s.x_max[a] - s.x_min[a]) / (float)grid3[a];          // (x_max - x_min)[a] / grid (fp32)
    }
    return s;
}

// Cell of a point: the reference's floor_divide formula, then strict containment in that cell's box (Cell.fill :3001-3008).
__device__ __forceinline__ int scene_cell_strict(const SceneGeom& s, const float* p) {
    int idx[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (!(p[a] >= s.x_min[a] && p[a] <= s.x_max[a])) return -1;       // get_pts_in_bounding_box (inclusive)
        const float x = p[a] - s.x_min[a];
        float q = (x - fmodf(x, s.step[a])) / s.step[a];
        if (q >= (float)s.g[a]) q = (float)(s.g[a] - 1);
        int i = (int)q;
        idx[a] = i < 0 ? 0 : i;
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float centre = s.x_min[a] + (0.5f + (float)idx[a]) * s.step[a];
        const float half = s.step[a] / 2.f;
        const float cmin = centre - half, cmax = centre + half;
        if (!(p[a] - cmax < 0.f) || !(p[a] - cmin > 0.f)) return -1;
    }
    return (idx[0] * s.g[1] + idx[1]) * s.g[2] + idx[2];
}

// counts[c] += number of lanes of this wave holding cell c (c < 0: none).  One atomic per distinct cell and wave: same-address
// device atomics serialise at ~3.6 ns each on this part (50 k points on 9 cells cost 180 us one by one).
__device__ __forceinline__ void wave_count_cells(int c, int* __restrict__ counts) {
    unsigned long long todo = __ballot(c >= 0);
    const int lane = threadIdx.x & 63;
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int lc = __shfl(c, leader);
        const unsigned long long same = __ballot(c == lc) & todo;
        if (lane == leader) atomicAdd(&counts[lc], __popcll(same));
        todo &= ~same;
    }
}

__global__ __launch_bounds__(256) void scene_assign_kernel(const float* __restrict__ pts, long long n_host,
                                                           const long long* __restrict__ n_dev, SceneGeom s,
                                                           int* __restrict__ cell_of, int* __restrict__ cand_count) {
    const long long n = n_dev ? (*n_dev < n_host ? *n_dev : n_host) : n_host;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long rounds = (n_host + stride - 1) / stride;               // uniform trip count: the ballots need whole waves
    for (long long r = 0; r < rounds; ++r) {
        const long long i = r * stride + (long long)blockIdx.x * blockDim.x + threadIdx.x;
        int c = -1;
        if (i < n) {
            const float p[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
            c = scene_cell_strict(s, p);
        }
        wave_count_cells(c, cand_count);
        if (i < n_host) cell_of[i] = c;
    }
}

// Counting sort of the stored points (all cells) into the point grid.
__global__ __launch_bounds__(256) void store_bin_kernel(const float* __restrict__ store, const int* __restrict__ store_count,
                                                        int n_cells, int cap, Grid g, int* __restrict__ gcell_of,
                                                        int* __restrict__ gslot_of, int* __restrict__ gcount) {
    const long long total = (long long)n_cells * cap;
    for (long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x; s < total; s += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(s / cap), j = (int)(s - (long long)c * cap);
        int gc = -1;
        if (j < store_count[c]) {
            gc = grid_cell(g, store[3 * s], store[3 * s + 1], store[3 * s + 2], nullptr);
            gslot_of[s] = atomicAdd(&gcount[gc], 1);
        }
        gcell_of[s] = gc;
    }
}

__global__ __launch_bounds__(256) void store_scatter_kernel(const float* __restrict__ store, int n_cells, int cap,
                                                            const int* __restrict__ gcell_of, const int* __restrict__ gslot_of,
                                                            const int* __restrict__ gstart, float4* __restrict__ sorted) {
    const long long total = (long long)n_cells * cap;
    for (long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x; s < total; s += (long long)gridDim.x * blockDim.x) {
        const int gc = gcell_of[s];
        if (gc < 0) continue;
        const int c = (int)(s / cap);
        sorted[gstart[gc] + gslot_of[s]] = make_float4(store[3 * s], store[3 * s + 1], store[3 * s + 2], __int_as_float(c));
    }
}

// Is there a stored point of scene cell `c` within radius r of p?  (inclusive: d <= r; else d < r); fp64 distances.
__device__ __forceinline__ bool near_stored(const Grid& g, const float4* __restrict__ sorted, const int* __restrict__ gstart,
                                            const float* p, int c, double r, bool inclusive) {
    int ijk[3];
    grid_cell(g, p[0], p[1], p[2], ijk);
    for (int a = max(ijk[0] - 1, 0); a <= min(ijk[0] + 1, g.n[0] - 1); ++a)
        for (int b = max(ijk[1] - 1, 0); b <= min(ijk[1] + 1, g.n[1] - 1); ++b) {
            const int d0 = max(ijk[2] - 1, 0), d1 = min(ijk[2] + 1, g.n[2] - 1);
            const int base = (a * g.n[1] + b) * g.n[2];
            for (int j = gstart[base + d0]; j < gstart[base + d1 + 1]; ++j) {
                const float4 q = sorted[j];
                if (__float_as_int(q.w) != c) continue;
                const double ex = (double)p[0] - (double)q.x, ey = (double)p[1] - (double)q.y, ez = (double)p[2] - (double)q.z;
                const double d = sqrt((ex * ex + ey * ey) + ez * ez);
                if (inclusive ? d <= r : d < r) return true;
            }
        }
    return false;
}

__global__ __launch_bounds__(256) void scene_thin_kernel(const float* __restrict__ pts, long long n, const int* __restrict__ cell_of,
                                                         const int* __restrict__ cand_count, int n_point_min,
                                                         const int* __restrict__ store_count, Grid g,
                                                         const float4* __restrict__ sorted, const int* __restrict__ gstart,
                                                         double resolution, int* __restrict__ keep, int* __restrict__ kept_count) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c = i < n ? cell_of[i] : -1;
    int k = 0;
    if (c >= 0 && cand_count[c] > n_point_min) {
        k = 1;
        if (store_count[c] > 0) {
            const float p[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
            if (near_stored(g, sorted, gstart, p, c, resolution, true)) k = 0;      // keep iff min distance > resolution
        }
    }
    wave_count_cells(k ? c : -1, kept_count);
    if (i < n) keep[i] = k;
}

__global__ void small_exclusive_scan_kernel(const int* __restrict__ v, int n, int* __restrict__ out) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int run = 0;
        for (int i = 0; i < n; ++i) { out[i] = run; run += v[i]; }
        out[n] = run;
    }
}

// One workgroup per cell: ordered compaction of the kept new points of that cell (point order) into its stage segment,
// 4096 points per round (16 consecutive points per thread, one workgroup scan per round).
__global__ __launch_bounds__(256) void scene_stage_kernel(const float* __restrict__ pts, long long n, const int* __restrict__ cell_of,
                                                          const int* __restrict__ keep, const int* __restrict__ kept_count,
                                                          const int* __restrict__ kept_start, float* __restrict__ stage) {
    __shared__ int wtot[4];
    const int c = blockIdx.x;
    if (kept_count[c] == 0) return;
    float* dst = stage + 3 * (size_t)kept_start[c];
    int base = 0;
    for (long long i0 = 0; i0 < n; i0 += 4096) {
        const long long first = i0 + 16 * (long long)threadIdx.x;
        unsigned flags = 0;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const long long i = first + e;
            if (i < n && cell_of[i] == c && keep[i]) flags |= 1u << e;
        }
        int total;
        int pos = base + block_exclusive_scan_256(__popc(flags), wtot, &total);
#pragma unroll
        for (int e = 0; e < 16; ++e)
            if (flags & (1u << e)) {
                const long long i = first + e;
                dst[3 * pos] = pts[3 * i]; dst[3 * pos + 1] = pts[3 * i + 1]; dst[3 * pos + 2] = pts[3 * i + 2];
                ++pos;
            }
        base += total;
        __syncthreads();
    }
}

// One workgroup per cell: append, or (over capacity) keep an exact-size seeded random subset of [stored | new].
__global__ __launch_bounds__(256) void scene_merge_kernel(float* __restrict__ store, int* __restrict__ store_count, int cap,
                                                          const float* __restrict__ stage, const int* __restrict__ kept_count,
                                                          const int* __restrict__ kept_start, unsigned seed,
                                                          float* __restrict__ temp) {
    const int c = blockIdx.x;
    const int old = store_count[c], kn = kept_count[c];
    if (kn == 0) return;
    const int total = old + kn;
    float* cell = store + 3 * (size_t)c * cap;
    const float* add = stage + 3 * (size_t)kept_start[c];
    if (total <= cap) {
        for (int j = threadIdx.x; j < 3 * kn; j += blockDim.x) cell[3 * old + j] = add[j];
    } else {
        float* tmp = temp + 3 * (size_t)c * cap;
        const unsigned bits = perm_bits((unsigned)total), sd = seed + 0x9E3779B1u * (unsigned)(c + 1);
        for (int j = threadIdx.x; j < cap; j += blockDim.x) {
            const int src = (int)perm_index((unsigned)j, (unsigned)total, bits, sd);
            const float* q = src < old ? cell + 3 * src : add + 3 * (src - old);
            tmp[3 * j] = q[0]; tmp[3 * j + 1] = q[1]; tmp[3 * j + 2] = q[2];
        }
        __syncthreads();                                    // the whole cell belongs to this workgroup
        for (int j = threadIdx.x; j < 3 * cap; j += blockDim.x) cell[j] = tmp[j];
    }
    __syncthreads();
    if (threadIdx.x == 0) store_count[c] = total < cap ? total : cap;
}

__global__ __launch_bounds__(256) void scene_gather_kernel(const float* __restrict__ store, const int* __restrict__ store_count,
                                                           int n_cells, int cap, float* __restrict__ out, long long out_cap,
                                                           long long* __restrict__ n_out) {
    const int c = blockIdx.x;
    long long off = 0, all = 0;
    for (int k = 0; k < n_cells; ++k) { const int v = store_count[k]; if (k < c) off += v; all += v; }
    if (c == 0 && threadIdx.x == 0) *n_out = all < out_cap ? all : out_cap;
    const int cnt = store_count[c];
    const float* src = store + 3 * (size_t)c * cap;
    for (int j = threadIdx.x; j < cnt; j += blockDim.x) {
        if (off + j >= out_cap) break;
        float* d = out + 3 * (off + j);
        d[0] = src[3 * j]; d[1] = src[3 * j + 1]; d[2] = src[3 * j + 2];
    }
}

__global__ __launch_bounds__(256) void scene_coverage_kernel(const float* __restrict__ gt, const int* __restrict__ gt_count,
                                                             int n_cells, int cap_gt, Grid g, const float4* __restrict__ sorted,
                                                             const int* __restrict__ gstart, double eps, int* __restrict__ out2) {
    const long long total = (long long)n_cells * cap_gt;
    int covered = 0, have = 0;
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s < total) {
        const int c = (int)(s / cap_gt), j = (int)(s - (long long)c * cap_gt);
        if (j < gt_count[c]) {
            have = 1;
            const float p[3] = {gt[3 * s], gt[3 * s + 1], gt[3 * s + 2]};
            covered = near_stored(g, sorted, gstart, p, c, eps, false) ? 1 : 0;      // heaviside(eps - d, 0): d < eps
        }
    }
    const unsigned long long bc = __ballot(covered), bh = __ballot(have);
    if ((threadIdx.x & 63) == 0) {
        if (bc) atomicAdd(&out2[0], __popcll(bc));
        if (bh) atomicAdd(&out2[1], __popcll(bh));
    }
}

// out[j] = pc[perm(j)] for j < min(N, k): the first k of a seeded random permutation of the cloud
// (fill_surface_scene's torch.randperm(len(full_pc))[:random_sampling_max_size], mu:715-716).
__global__ __launch_bounds__(256) void sample_points_kernel(const float* __restrict__ pc, long long n_host,
                                                            const long long* __restrict__ n_dev, long long k, unsigned seed,
                                                            float* __restrict__ out, long long* __restrict__ m_out) {
    const long long N = n_dev ? (*n_dev < n_host ? *n_dev : n_host) : n_host;
    const long long M = N > k ? k : N;
    if (blockIdx.x == 0 && threadIdx.x == 0) *m_out = M;
    const unsigned bits = perm_bits((unsigned)N);
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < M; j += (long long)gridDim.x * blockDim.x) {
        const long long src = (long long)perm_index((unsigned)j, (unsigned)N, bits, seed);
        out[3 * j] = pc[3 * src]; out[3 * j + 1] = pc[3 * src + 1]; out[3 * j + 2] = pc[3 * src + 2];
    }
}

size_t al256s(size_t b) { return (b + 255) / 256 * 256; }

// Point grid over the scene box grown by r; edge = max(r, largest extent / 256).
int point_grid(const float* box6, double r, Grid* g, size_t* ncell) {
    double ext_max = 0;
    for (int a = 0; a < 3; ++a) ext_max = fmax(ext_max, (double)box6[3 + a] - box6[a] + 2 * r);
    const double edge = fmax(r, ext_max / 256.0);
    if (!(edge > 0)) return NBP_E_ARG;
    size_t n = 1;
    for (int a = 0; a < 3; ++a) {
        g->lo[a] = (float)(box6[a] - r);
        g->n[a] = (int)(((double)box6[3 + a] + r - g->lo[a]) / edge) + 1;
        n *= (size_t)g->n[a];
    }
    g->inv = (float)(1.0 / (edge * 1.001));     // cells a little WIDER than the radius: fp32 rounding of (x - lo) * inv
                                                // can then never put two points within r more than one cell apart
    *ncell = n;
    return n > ((size_t)1 << 26) ? NBP_E_SHAPE : 0;
}

struct GridWs { int* gcount; int* gstart; int* tsum; int* gcell_of; int* gslot_of; float4* sorted; };
size_t grid_ws_bytes(size_t ncell, size_t slots) {
    return al256s(ncell * 4) + al256s((ncell + 1) * 4) + al256s((ncell / SCAN_TILE + 1) * 4) + 2 * al256s(slots * 4) +
           al256s(slots * 16);
}
char* grid_ws_carve(char* p, size_t ncell, size_t slots, GridWs* w) {
    w->gcount = (int*)p; p += al256s(ncell * 4);
    w->gstart = (int*)p; p += al256s((ncell + 1) * 4);
    w->tsum = (int*)p; p += al256s((ncell / SCAN_TILE + 1) * 4);
    w->gcell_of = (int*)p; p += al256s(slots * 4);
    w->gslot_of = (int*)p; p += al256s(slots * 4);
    w->sorted = (float4*)p; p += al256s(slots * 16);
    return p;
}

int build_store_grid(const float* store, const int* store_count, int n_cells, int cap, const Grid& g, size_t ncell,
                     const GridWs& w, hipStream_t st) {
    hipError_t e = hipMemsetAsync(w.gcount, 0, ncell * 4, st);
    if (e != hipSuccess) return (int)e;
    const int grid = nbp_ew_grid((long long)n_cells * cap, 256);
    store_bin_kernel<<<grid, 256, 0, st>>>(store, store_count, n_cells, cap, g, w.gcell_of, w.gslot_of, w.gcount);
    int rc = nbp_launch_status();
    if (rc) return rc;
    if ((rc = grid_exclusive_scan(w.gcount, (long long)ncell, w.tsum, w.gstart, st))) return rc;
    store_scatter_kernel<<<grid, 256, 0, st>>>(store, n_cells, cap, w.gcell_of, w.gslot_of, w.gstart, w.sorted);
    return nbp_launch_status();
}

bool geom_ok(const float* box6, const int* grid3, int* n_cells) {
    if (!box6 || !grid3) return false;
    long long n = 1;
    for (int a = 0; a < 3; ++a) {
        if (grid3[a] < 1 || !(box6[3 + a] > box6[a])) return false;
        n *= grid3[a];
    }
    if (n > 65535) return false;
    *n_cells = (int)n;
    return true;
}

}  // namespace

extern "C" size_t nbp_scene_fill_workspace_bytes(const float* box6_host, const int* grid3_host, int capacity,
                                                 long long n_pts_max, double resolution) {
    int n_cells; Grid g; size_t ncell;
    if (!geom_ok(box6_host, grid3_host, &n_cells) || capacity < 1 || n_pts_max < 1 || !(resolution > 0)) return 0;
    if (point_grid(box6_host, resolution, &g, &ncell)) return 0;
    const size_t slots = (size_t)n_cells * capacity;
    return 512 + 2 * al256s((size_t)n_pts_max * 4) + 2 * al256s((size_t)n_cells * 4) + al256s((size_t)(n_cells + 1) * 4) +
           al256s((size_t)n_pts_max * 12) + al256s(slots * 12) + grid_ws_bytes(ncell, slots);
}

extern "C" int nbp_scene_fill_cells_f32(const float* pts3, long long n, const long long* n_dev_or_null,
                                        const float* box6_host, const int* grid3_host, int capacity, double resolution,
                                        int n_point_min, unsigned seed, float* store_pts, int* store_count, void* ws,
                                        size_t ws_bytes, void* stream) {
    NBP_ENTER();
    int n_cells; Grid g; size_t ncell;
    NBP_RETURN_IF(!pts3 || !store_pts || !store_count || !ws || n < 1 || capacity < 1 || !(resolution > 0) || n_point_min < 0,
                  NBP_E_ARG);
    NBP_RETURN_IF(!geom_ok(box6_host, grid3_host, &n_cells), NBP_E_ARG);
    int rc = point_grid(box6_host, resolution, &g, &ncell);
    if (rc) return rc;
    NBP_RETURN_IF(ws_bytes < nbp_scene_fill_workspace_bytes(box6_host, grid3_host, capacity, n, resolution), NBP_E_WS);
    hipStream_t st = (hipStream_t)stream;
    const size_t slots = (size_t)n_cells * capacity;
    char* p = (char*)(((uintptr_t)ws + 255) / 256 * 256);
    int* cell_of = (int*)p; p += al256s((size_t)n * 4);
    int* keep = (int*)p; p += al256s((size_t)n * 4);
    int* cand_count = (int*)p; p += al256s((size_t)n_cells * 4);
    int* kept_count = (int*)p; p += al256s((size_t)n_cells * 4);
    int* kept_start = (int*)p; p += al256s((size_t)(n_cells + 1) * 4);
    float* stage = (float*)p; p += al256s((size_t)n * 12);
    float* temp = (float*)p; p += al256s(slots * 12);
    GridWs gw;
    grid_ws_carve(p, ncell, slots, &gw);
    hipError_t e = hipMemsetAsync(cand_count, 0, 2 * al256s((size_t)n_cells * 4), st);      // cand_count + kept_count
    if (e != hipSuccess) return (int)e;
    const SceneGeom geom = make_geom(box6_host, grid3_host);
    scene_assign_kernel<<<nbp_ew_grid(n, 256), 256, 0, st>>>(pts3, n, n_dev_or_null, geom, cell_of, cand_count);
    if ((rc = nbp_launch_status())) return rc;
    if ((rc = build_store_grid(store_pts, store_count, n_cells, capacity, g, ncell, gw, st))) return rc;
    scene_thin_kernel<<<(unsigned)nbp_cdiv(n, 256), 256, 0, st>>>(pts3, n, cell_of, cand_count, n_point_min, store_count, g,
                                                                 gw.sorted, gw.gstart, resolution, keep, kept_count);
    if ((rc = nbp_launch_status())) return rc;
    small_exclusive_scan_kernel<<<1, 64, 0, st>>>(kept_count, n_cells, kept_start);
    scene_stage_kernel<<<n_cells, 256, 0, st>>>(pts3, n, cell_of, keep, kept_count, kept_start, stage);
    if ((rc = nbp_launch_status())) return rc;
    scene_merge_kernel<<<n_cells, 256, 0, st>>>(store_pts, store_count, capacity, stage, kept_count, kept_start, seed, temp);
    return nbp_launch_status();
}

extern "C" int nbp_scene_gather_f32(const float* store_pts, const int* store_count, int n_cells, int capacity, float* out3,
                                    long long out_capacity, long long* n_out, void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(!store_pts || !store_count || !out3 || !n_out || n_cells < 1 || n_cells > 65535 || capacity < 1 ||
                  out_capacity < 1, NBP_E_ARG);
    scene_gather_kernel<<<n_cells, 256, 0, (hipStream_t)stream>>>(store_pts, store_count, n_cells, capacity, out3,
                                                                  out_capacity, n_out);
    return nbp_launch_status();
}

extern "C" size_t nbp_scene_coverage_workspace_bytes(const float* box6_host, const int* grid3_host, int capacity_rec,
                                                     double epsilon) {
    int n_cells; Grid g; size_t ncell;
    if (!geom_ok(box6_host, grid3_host, &n_cells) || capacity_rec < 1 || !(epsilon > 0)) return 0;
    if (point_grid(box6_host, epsilon, &g, &ncell)) return 0;
    return 512 + grid_ws_bytes(ncell, (size_t)n_cells * capacity_rec);
}

extern "C" int nbp_scene_coverage_f32(const float* gt_pts, const int* gt_count, int capacity_gt, const float* rec_pts,
                                      const int* rec_count, int capacity_rec, const float* box6_host, const int* grid3_host,
                                      double epsilon, int* covered_and_total2, void* ws, size_t ws_bytes, void* stream) {
    NBP_ENTER();
    int n_cells; Grid g; size_t ncell;
    NBP_RETURN_IF(!gt_pts || !gt_count || !rec_pts || !rec_count || !covered_and_total2 || !ws || capacity_gt < 1 ||
                  capacity_rec < 1 || !(epsilon > 0), NBP_E_ARG);
    NBP_RETURN_IF(!geom_ok(box6_host, grid3_host, &n_cells), NBP_E_ARG);
    int rc = point_grid(box6_host, epsilon, &g, &ncell);
    if (rc) return rc;
    NBP_RETURN_IF(ws_bytes < nbp_scene_coverage_workspace_bytes(box6_host, grid3_host, capacity_rec, epsilon), NBP_E_WS);
    hipStream_t st = (hipStream_t)stream;
    GridWs gw;
    grid_ws_carve((char*)(((uintptr_t)ws + 255) / 256 * 256), ncell, (size_t)n_cells * capacity_rec, &gw);
    hipError_t e = hipMemsetAsync(covered_and_total2, 0, 2 * sizeof(int), st);
    if (e != hipSuccess) return (int)e;
    if ((rc = build_store_grid(rec_pts, rec_count, n_cells, capacity_rec, g, ncell, gw, st))) return rc;
    scene_coverage_kernel<<<(unsigned)nbp_cdiv((long long)n_cells * capacity_gt, 256), 256, 0, st>>>(
        gt_pts, gt_count, n_cells, capacity_gt, g, gw.sorted, gw.gstart, epsilon, covered_and_total2);
    return nbp_launch_status();
}

extern "C" int nbp_sample_points_f32(const float* pc3, long long N, const long long* N_dev_or_null, long long k, unsigned seed,
                                     float* out3, long long* m_out, void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(!pc3 || !out3 || !m_out || N < 1 || k < 1 || N > 0xffffffffll, NBP_E_ARG);
    sample_points_kernel<<<nbp_ew_grid(k < N ? k : N, 256), 256, 0, (hipStream_t)stream>>>(pc3, N, N_dev_or_null, k, seed, out3,
                                                                                          m_out);
    return nbp_launch_status();
}
