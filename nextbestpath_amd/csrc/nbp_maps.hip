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
    static const int rounds = [] { const int v = nbp_tune_int("NBP_MAP_ROUNDS", 4); return v == 1 || v == 2 ? v : 4; }();
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
