// nbp_sim.hip -- simulator kernels: depth un-projection with exact-size random sub-sampling,
// tile-binned z-buffer rasteriser, segment/mesh and inside-mesh ray tests.
//
// Replaces, on the reference's NBP path:
//   * Camera.compute_partial_point_cloud / project_depth_in_3D
//     (macarons/utility/macarons_utils.py:2788-2847; NDC tables :2270-2279) -- PyTorch3D
//     FoVPerspectiveCameras.unproject_points + torch.randperm sub-sampling;
//   * Camera.capture_image's zbuf (macarons_utils.py:2743-2786, renderer :905-937) --
//     PyTorch3D MeshRasterizer(image_size=(256,456), faces_per_pixel=1, blur_radius=0);
//   * line_segment_mesh_intersection (macarons_utils.py:120-151) and check_camera_in_mesh
//     (next_best_path/utility/long_term_utils.py:158-170) -- trimesh ray queries.
// PyTorch3D / trimesh are third-party and absent from the reference tree: these kernels follow
// the libraries' documented conventions (row-vector X_view = X_world R + T, NDC +X left / +Y
// up, zbuf = view-space z of the nearest face at the pixel centre, -1 background); parity with
// the libraries themselves is UNPINNED (oracle/__init__.py), parity with oracle/ is exact.
#include "common.h"
#pragma clang fp contract(off)

namespace {

// ------------------------------------------------------------------ un-projection
struct Cam { float R[9]; float T[3]; };
constexpr int MAX_CAMS = 8;
struct CamSet { Cam c[MAX_CAMS]; };   // cameras travel in the kernel arguments (no H2D copy, no sync)

__device__ __forceinline__ void to_view(const float* p, const float* R, const float* T, float* o) {
    // X_view = X_world R + T (row vector): o_j = sum_k p_k R[k][j] + T_j
    o[0] = ((p[0] * R[0] + p[1] * R[3]) + p[2] * R[6]) + T[0];
    o[1] = ((p[0] * R[1] + p[1] * R[4]) + p[2] * R[7]) + T[1];
    o[2] = ((p[0] * R[2] + p[1] * R[5]) + p[2] * R[8]) + T[2];
}

// Colour of pixel (row, col) of a frame whose nearest face there is fi: the face record is rebuilt from the mesh with the
// arithmetic of raster_setup_kernel (bit-identical e1, e2, v0, q), the barycentrics come from the same ray cast as the depth,
// colour = ambient x interpolated vertex colours (SoftPhongShader under AmbientLights on a TexturesVertex mesh).
__device__ __forceinline__ void shade_pixel(const float* __restrict__ verts, const int* __restrict__ faces,
                                            const float* __restrict__ vcolors, const Cam& cam, int fi, int row, int col, int H,
                                            int W, float tanh_fov, float ambient, float* rgb3) {
    float v[3][3];
    const int i0 = faces[3 * (size_t)fi], i1 = faces[3 * (size_t)fi + 1], i2 = faces[3 * (size_t)fi + 2];
    to_view(verts + 3 * (size_t)i0, cam.R, cam.T, v[0]);
    to_view(verts + 3 * (size_t)i1, cam.R, cam.T, v[1]);
    to_view(verts + 3 * (size_t)i2, cam.R, cam.T, v[2]);
    float e1[3], e2[3], q[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { e1[c] = v[1][c] - v[0][c]; e2[c] = v[2][c] - v[0][c]; }
    q[0] = e1[1] * v[0][2] - e1[2] * v[0][1];
    q[1] = e1[2] * v[0][0] - e1[0] * v[0][2];
    q[2] = e1[0] * v[0][1] - e1[1] * v[0][0];
    const int s = H < W ? H : W;
    const float dx = (((float)W - (2.f * col + 1.f)) / (float)s) * tanh_fov;
    const float dy = (((float)H - (2.f * row + 1.f)) / (float)s) * tanh_fov;
    // plane forms of the face (raster_setup_body's arithmetic, oracle/raster.py::plane_forms)
    const float a0 = e1[2] * e2[1] - e1[1] * e2[2], a1 = e1[0] * e2[2] - e1[2] * e2[0], a2 = e1[1] * e2[0] - e1[0] * e2[1];
    const float u0 = v[0][1] * e2[2] - v[0][2] * e2[1], u1 = v[0][2] * e2[0] - v[0][0] * e2[2], u2 = v[0][0] * e2[1] - v[0][1] * e2[0];
    const float inv = 1.f / ((a0 * dx + a1 * dy) + a2);
    const float u = ((u0 * dx + u1 * dy) + u2) * inv;
    const float vv = ((dx * q[0] + dy * q[1]) + q[2]) * inv;
    const float w0 = (1.f - u) - vv;
    const float* c0 = vcolors + 3 * (size_t)i0;
    const float* c1 = vcolors + 3 * (size_t)i1;
    const float* c2 = vcolors + 3 * (size_t)i2;
#pragma unroll
    for (int k = 0; k < 3; ++k) rgb3[k] = ambient * ((w0 * c0[k] + u * c1[k]) + vv * c2[k]);
}


// Pixel (row, col) with view-space depth z -> world point.  fp32 op order is part of the
// contract with oracle/camera.py (no FMA contraction in this file).
__device__ __forceinline__ void unproject_pixel(int row, int col, float z, int H, int W, float tanh_fov,
                                                const float* R, const float* T, float* out) {
    const int s = H < W ? H : W;
    const float ndc_x = (float)((double)W / s) - ((float)col / (float)(s - 1)) * 2.f;   // ref mu:2272-2274
    const float ndc_y = (float)((double)H / s) - ((float)row / (float)(s - 1)) * 2.f;   // ref mu:2275-2277
    const float xv = (ndc_x * z) * tanh_fov;
    const float yv = (ndc_y * z) * tanh_fov;
    const float dx = xv - T[0], dy = yv - T[1], dz = z - T[2];
    // X_world = (X_view - T) R^T  (row vectors)
    out[0] = (dx * R[0] + dy * R[1]) + dz * R[2];
    out[1] = (dx * R[3] + dy * R[4]) + dz * R[5];
    out[2] = (dx * R[6] + dy * R[7]) + dz * R[8];
}

// K1a/K1b: ordered compaction of the valid pixel indices (pixel order) with COMPACT_CHUNK pixels
// per 256-thread block: pass a counts per block, pass b recomputes validity and writes at the
// block's exclusive prefix (<= 64 block counts per frame are summed directly).
constexpr int COMPACT_CHUNK = 2048;

__device__ __forceinline__ bool pixel_valid(const float* d, const unsigned char* mk, int p, float fov_range) {
    const float z = d[p];
    return (mk ? mk[p] != 0 : z > -1.f) && z < fov_range;
}

__global__ __launch_bounds__(256) void unproject_count_kernel(const float* __restrict__ depth,
                                                              const unsigned char* __restrict__ mask, int HW, int nblk,
                                                              float fov_range, int* __restrict__ blk_count) {
    __shared__ int tot;
    const int f = blockIdx.y, b = blockIdx.x;
    const float* d = depth + (size_t)f * HW;
    const unsigned char* mk = mask ? mask + (size_t)f * HW : nullptr;
    if (threadIdx.x == 0) tot = 0;
    __syncthreads();
    int c = 0;
    for (int p = b * COMPACT_CHUNK + threadIdx.x; p < min((b + 1) * COMPACT_CHUNK, HW); p += 256)
        c += pixel_valid(d, mk, p, fov_range) ? 1 : 0;
    for (int o = 32; o; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(&tot, c);
    __syncthreads();
    if (threadIdx.x == 0) blk_count[f * nblk + b] = tot;
}

__global__ __launch_bounds__(256) void unproject_compact_kernel(const float* __restrict__ depth,
                                                                const unsigned char* __restrict__ mask, int HW, int nblk,
                                                                float fov_range, double gather,
                                                                const int* __restrict__ blk_count, unsigned* __restrict__ list,
                                                                int* __restrict__ counts) {
    __shared__ int wave_tot[4];
    __shared__ int base_s;
    const int f = blockIdx.y, b = blockIdx.x;
    const float* d = depth + (size_t)f * HW;
    const unsigned char* mk = mask ? mask + (size_t)f * HW : nullptr;
    unsigned* out = list + (size_t)f * HW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) {
        int pre = 0, all = 0;
        for (int k = 0; k < nblk; ++k) { const int v = blk_count[f * nblk + k]; all += v; if (k < b) pre += v; }
        base_s = pre;
        if (b == 0) {
            counts[2 * f] = all;
            counts[2 * f + 1] = (int)((double)all * gather);   // int(len(world_points) * gathering_factor)
        }
    }
    __syncthreads();
    const int pend = min((b + 1) * COMPACT_CHUNK, HW);
    for (int p0 = b * COMPACT_CHUNK; p0 < pend; p0 += 256) {
        const int p = p0 + threadIdx.x;
        const bool ok = p < pend && pixel_valid(d, mk, p, fov_range);
        const unsigned long long bal = __ballot(ok);
        const int within = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wave] = __popcll(bal);
        __syncthreads();
        int off = base_s;
        for (int w = 0; w < wave; ++w) off += wave_tot[w];
        if (ok) out[off + within] = (unsigned)p;
        __syncthreads();
        if (threadIdx.x == 0) base_s += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        __syncthreads();
    }
}

// K1 fast path (H*W a multiple of 4, 16-B aligned frames): 4096-pixel chunks, one 256-thread workgroup per chunk and frame,
// 16 B per lane and load.  Pass a counts the valid pixels of every chunk; pass b recomputes the flags (the frame is L2
// resident by then), ranks them inside the wave with three ballots over the per-lane counts (no barrier per 256 pixels as
// in the generic kernels above), adds the [sub-chunk][wave] totals of the workgroup and the counts of the earlier chunks.
constexpr int FAST_CHUNK = 4096;

__device__ __forceinline__ unsigned quad_flags(const float4* __restrict__ d4, const uchar4* __restrict__ m4, int q, int nq,
                                               float fov_range) {
    if (q >= nq) return 0u;
    const float4 z = d4[q];
    if (m4) {
        const uchar4 m = m4[q];
        return (m.x != 0 && z.x < fov_range ? 1u : 0u) | (m.y != 0 && z.y < fov_range ? 2u : 0u) |
               (m.z != 0 && z.z < fov_range ? 4u : 0u) | (m.w != 0 && z.w < fov_range ? 8u : 0u);
    }
    return (z.x > -1.f && z.x < fov_range ? 1u : 0u) | (z.y > -1.f && z.y < fov_range ? 2u : 0u) |
           (z.z > -1.f && z.z < fov_range ? 4u : 0u) | (z.w > -1.f && z.w < fov_range ? 8u : 0u);
}

__device__ __forceinline__ void unproject_count4_body(unsigned bx, unsigned by, unsigned gx, unsigned gy, const float* __restrict__ depth,
                                                               const unsigned char* __restrict__ mask, int HW, int nblk,
                                                               float fov_range, int* __restrict__ blk_count,
                                                               int* __restrict__ done_ticket) {
    __shared__ int wt[4];
    const int f = by, b = bx, t = threadIdx.x;
    const float4* d4 = reinterpret_cast<const float4*>(depth + (size_t)f * HW);
    const uchar4* m4 = mask ? reinterpret_cast<const uchar4*>(mask + (size_t)f * HW) : nullptr;
    if (f == 0 && b == 0 && t == 0) *done_ticket = 0;      // ticket of the append kernel's "last block updates the size"
    int c = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) c += __popc(quad_flags(d4, m4, b * 1024 + k * 256 + t, HW >> 2, fov_range));
    for (int o = 32; o; o >>= 1) c += __shfl_xor(c, o);
    if ((t & 63) == 0) wt[t >> 6] = c;
    __syncthreads();
    if (t == 0) blk_count[f * nblk + b] = wt[0] + wt[1] + wt[2] + wt[3];
}

__device__ __forceinline__ void unproject_compact4_body(unsigned bx, unsigned by, unsigned gx, unsigned gy, const float* __restrict__ depth,
                                                                 const unsigned char* __restrict__ mask, int HW, int nblk,
                                                                 float fov_range, double gather,
                                                                 const int* __restrict__ blk_count, unsigned* __restrict__ list,
                                                                 int* __restrict__ counts) {
    __shared__ int tot[17];
    __shared__ int base_s;
    const int f = by, b = bx, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float4* d4 = reinterpret_cast<const float4*>(depth + (size_t)f * HW);
    const uchar4* m4 = mask ? reinterpret_cast<const uchar4*>(mask + (size_t)f * HW) : nullptr;
    unsigned* out = list + (size_t)f * HW;
    unsigned fl[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) fl[k] = quad_flags(d4, m4, b * 1024 + k * 256 + t, HW >> 2, fov_range);
    const unsigned long long lt = (1ull << lane) - 1ull;
    int rank[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int n = __popc(fl[k]);
        const unsigned long long b0 = __ballot(n & 1), b1 = __ballot(n & 2), b2 = __ballot(n & 4);
        rank[k] = __popcll(b0 & lt) + 2 * __popcll(b1 & lt) + 4 * __popcll(b2 & lt);
        if (lane == 0) tot[k * 4 + wave] = __popcll(b0) + 2 * __popcll(b1) + 4 * __popcll(b2);
    }
    if (t == 0) {
        int pre = 0, all = 0;
        for (int k = 0; k < nblk; ++k) { const int v = blk_count[f * nblk + k]; all += v; if (k < b) pre += v; }
        base_s = pre;
        if (b == 0) {
            counts[2 * f] = all;
            counts[2 * f + 1] = (int)((double)all * gather);   // int(len(world_points) * gathering_factor)
        }
    }
    __syncthreads();
    if (t == 0) {
        int run = base_s;
        for (int i = 0; i < 16; ++i) { const int v = tot[i]; tot[i] = run; run += v; }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int pos = tot[k * 4 + wave] + rank[k];
        const unsigned p0 = (unsigned)(b * 1024 + k * 256 + t) * 4u;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (fl[k] & (1u << e)) out[pos++] = p0 + e;
    }
}

// ---- filing into the cloud's tile-binned shadow copy (nbp_bins.h; csrc/nbp_maps.hip builds the maps from it).  A launch that gets a
// store files every point it appends (the point is in registers anyway) and clears the buffers the map build behind it accumulates
// into, so that the build of a single rollout is ONE launch (nbp_step_maps_prefiled_f32) instead of bin_append_kernel + map_binned_kernel.
// It files only while the store is in step with the cloud (n_binned == *cloud_count at entry, launch-uniform: both are advanced by
// the last workgroup only); otherwise the points stay unfiled and the next build counts them directly / bin_append_kernel files them.
#include "nbp_bins.h"
struct BinFile { char* store; float* zero6; float* zero1; int SS; };

// K2: gather the sub-sample and append it to the cloud at *cloud_count + sum of earlier frames.
__device__ __forceinline__ void unproject_append_body(unsigned bx, unsigned by, unsigned gx, unsigned gy, const float* __restrict__ depth, const Cam* cams,
                                                               int H, int W, float tanh_fov, unsigned seed,
                                                               const unsigned* __restrict__ list,
                                                               const int* __restrict__ counts, float* __restrict__ cloud,
                                                               long long* __restrict__ cloud_count,
                                                               long long capacity, int n_frames, int* __restrict__ done_ticket,
                                                               const float* __restrict__ rgb, float* __restrict__ cloud_rgb,
                                                               const unsigned long long* __restrict__ zface,
                                                               const float* __restrict__ verts, const int* __restrict__ faces,
                                                               const float* __restrict__ vcolors, float ambient, int fence = 0,
                                                               BinFile bf = BinFile{nullptr, nullptr, nullptr, 0}) {
    const int f = by;
    const int HW = H * W;
    const int nvalid = counts[2 * f], nkeep = counts[2 * f + 1];
    const long long count0 = *cloud_count;
    long long base = count0;
    for (int g = 0; g < f; ++g) base += counts[2 * g + 1];
    const unsigned bits = perm_bits((unsigned)nvalid);
    const unsigned sd = seed + 0x632BE5ABu * (unsigned)(f + 1);
    const Cam cam = cams[f];
    // the store, when given and in step with the cloud: filing needs whole waves in the loop (ballots), hence the rounded trip count
    BinView bv{};
    BinGeom bg{};
    bool in_step = false, filing = false;
    if (bf.store) {
        bin_clear_maps(bf.zero6, bf.zero1, bf.SS, by * gx + bx, gx * gy);
        bv = bin_view(bf.store);
        in_step = bv.d->n_binned == count0;
        filing = in_step && bv.d->error == 0u;            // (a broken store files nothing more: its builds scan the cloud)
        if (filing) bg = bin_geom(bv);
    }
    const int step = (int)(gx * blockDim.x);
    const int n_trip = filing ? (nkeep + 63) / 64 * 64 : nkeep;
    for (int j = bx * blockDim.x + threadIdx.x; j < n_trip; j += step) {
        const bool active = j < nkeep && base + j < capacity;
        if (!filing && !active) break;
        bins_f32x3 pt = {__builtin_nanf(""), 0.f, 0.f};
        unsigned pix = 0;
        int row = 0, col = 0;
        if (active) {
            pix = list[(size_t)f * HW + perm_index((unsigned)j, (unsigned)nvalid, bits, sd)];
            row = (int)(pix / (unsigned)W); col = (int)(pix - (unsigned)row * W);
            float o[3];
            unproject_pixel(row, col, depth[(size_t)f * HW + pix], H, W, tanh_fov, cam.R, cam.T, o);
            float* dst = cloud + (base + j) * 3;
            dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2];
            pt = bins_f32x3{o[0], o[1], o[2]};
        }
        if (filing) bin_file_wave(bv, bg, active, pt, base + j);
        if (!active) continue;
        if (cloud_rgb) {                                   // the colours of the kept pixels (mu:2840-2845)
            float* cd = cloud_rgb + (base + j) * 3;
            if (zface) {                                   // deferred shading: only the ~5 % of pixels that are kept
                const unsigned long long zf = zface[(size_t)f * HW + pix];
                float c[3] = {1.f, 1.f, 1.f};
                if (zf != ~0ull) shade_pixel(verts, faces, vcolors, cam, (int)(unsigned)zf, row, col, H, W, tanh_fov, ambient, c);
                cd[0] = c[0]; cd[1] = c[1]; cd[2] = c[2];
            } else {
                const float* c = rgb + 3 * ((size_t)f * HW + pix);
                cd[0] = c[0]; cd[1] = c[1]; cd[2] = c[2];
            }
        }
    }
    if (!done_ticket) return;
    // The block that finishes last advances the cloud size: every block has read *cloud_count by then.
    __shared__ int last;
    __syncthreads();
    if (threadIdx.x == 0) {
        // No fence: on gfx950 __threadfence() is an L2 write-back + invalidate (buffer_wbl2 sc1 / buffer_inv sc1) -- thousands of
        // them per lock-step group flush the L2 lines of the convolutions running beside this kernel.  None is needed: this block's
        // read of *cloud_count was consumed (the stores above used it) before the barrier, what the last block reads (counts) was
        // written by an earlier launch, and what this launch writes is read by later launches only.  (fence = 1: the round-3 form, A/B.)
        if (fence) __threadfence();
        last = atomicAdd(done_ticket, 1) == (int)(gx * gy) - 1;
    }
    __syncthreads();
    if (last && threadIdx.x == 0) {
        long long n = *cloud_count;
        for (int g = 0; g < n_frames; ++g) n += counts[2 * g + 1];
        *cloud_count = n < capacity ? n : capacity;
        if (in_step) bv.d->n_binned = n < capacity ? n : capacity;      // filed, or a broken store (whose builds scan the cloud)
        *done_ticket = 0;
    }
}

// ---- launch forms of the three un-projection kernels: one call (the arguments as they were), or the rollouts of a lock-step
// group in one launch (blockIdx.z = rollout; up to STEP_BATCH items ride in the kernel arguments).  The step's kernels are
// latency-bound chains of a few workgroups: n of them side by side cost one chain, not n.
constexpr int STEP_BATCH = 12;
struct UnprojItem {   // the frames of an item need not be adjacent in memory (a camera's ring of frames wraps): one pointer each
    const float* depth[4]; const unsigned long long* zface[4]; int* blk_count; int* counts; float* cloud;
    long long* cloud_count; long long capacity; float* cloud_rgb; const float* verts; const int* faces;
    const float* vcolors; unsigned seed; Cam cam[4];
};
// the ticket and the pixel list of an item sit at launch-uniform offsets behind its block counts (one scratch buffer per item)
struct UnprojBatch {
    UnprojItem it[STEP_BATCH]; int ticket_off; unsigned list_off_bytes;
    __device__ int* ticket(int r) const { return it[r].blk_count + ticket_off; }
    __device__ unsigned* list(int r) const { return reinterpret_cast<unsigned*>(reinterpret_cast<char*>(it[r].blk_count) + list_off_bytes); }
};
// kernel arguments stay inside the 4 KB every HIP runtime accepts (12 x 352 B was 4224 B and ran, out of spec)
static_assert(sizeof(UnprojBatch) + 32 <= 4096, "UnprojBatch + the scalar arguments of its kernels must fit 4 KB of kernel arguments");

__global__ __launch_bounds__(256) void unproject_count4_kernel(const float* __restrict__ depth, const unsigned char* __restrict__ mask,
                                                               int HW, int nblk, float fov_range, int* __restrict__ blk_count,
                                                               int* __restrict__ done_ticket) {
    unproject_count4_body(blockIdx.x, blockIdx.y, gridDim.x, gridDim.y, depth, mask, HW, nblk, fov_range, blk_count, done_ticket);
}
__global__ __launch_bounds__(256) void unproject_count4_batch_kernel(UnprojBatch b, int HW, int nblk, float fov_range) {
    const UnprojItem& a = b.it[blockIdx.z];
    // (the bodies address frame f as base + f HW: the base is shifted so that this lands on the frame's own pointer)
    unproject_count4_body(blockIdx.x, blockIdx.y, gridDim.x, gridDim.y, a.depth[blockIdx.y] - (size_t)blockIdx.y * HW, nullptr, HW, nblk,
                          fov_range, a.blk_count, b.ticket(blockIdx.z));
}
__global__ __launch_bounds__(256) void unproject_compact4_kernel(const float* __restrict__ depth, const unsigned char* __restrict__ mask,
                                                                 int HW, int nblk, float fov_range, double gather,
                                                                 const int* __restrict__ blk_count, unsigned* __restrict__ list,
                                                                 int* __restrict__ counts) {
    unproject_compact4_body(blockIdx.x, blockIdx.y, gridDim.x, gridDim.y, depth, mask, HW, nblk, fov_range, gather, blk_count, list, counts);
}
__global__ __launch_bounds__(256) void unproject_compact4_batch_kernel(UnprojBatch b, int HW, int nblk, float fov_range, double gather) {
    const UnprojItem& a = b.it[blockIdx.z];
    unproject_compact4_body(blockIdx.x, blockIdx.y, gridDim.x, gridDim.y, a.depth[blockIdx.y] - (size_t)blockIdx.y * HW, nullptr, HW, nblk,
                            fov_range, gather, a.blk_count, b.list(blockIdx.z), a.counts);
}
__global__ __launch_bounds__(256) void unproject_append_kernel(const float* __restrict__ depth, CamSet cams, int H, int W, float tanh_fov,
                                                               unsigned seed, const unsigned* __restrict__ list,
                                                               const int* __restrict__ counts, float* __restrict__ cloud,
                                                               long long* __restrict__ cloud_count, long long capacity, int n_frames,
                                                               int* __restrict__ done_ticket, const float* __restrict__ rgb,
                                                               float* __restrict__ cloud_rgb, const unsigned long long* __restrict__ zface,
                                                               const float* __restrict__ verts, const int* __restrict__ faces,
                                                               const float* __restrict__ vcolors, float ambient, int fence, BinFile bf) {
    unproject_append_body(blockIdx.x, blockIdx.y, gridDim.x, gridDim.y, depth, cams.c, H, W, tanh_fov, seed, list, counts, cloud, cloud_count,
                          capacity, n_frames, done_ticket, rgb, cloud_rgb, zface, verts, faces, vcolors, ambient, fence, bf);
}
__global__ __launch_bounds__(256) void unproject_append_batch_kernel(UnprojBatch b, int H, int W, float tanh_fov, int n_frames, float ambient,
                                                                     int fence) {
    const UnprojItem& a = b.it[blockIdx.z];
    const size_t shift = (size_t)blockIdx.y * H * W;
    unproject_append_body(blockIdx.x, blockIdx.y, gridDim.x, gridDim.y, a.depth[blockIdx.y] - shift, a.cam, H, W, tanh_fov, a.seed,
                          b.list(blockIdx.z), a.counts, a.cloud, a.cloud_count, a.capacity, n_frames, b.ticket(blockIdx.z), nullptr, a.cloud_rgb,
                          a.zface[blockIdx.y] ? a.zface[blockIdx.y] - shift : nullptr, a.verts, a.faces, a.vcolors, ambient, fence);
}

__global__ void cloud_count_update_kernel(const int* __restrict__ counts, int F, long long* __restrict__ cloud_count,
                                          long long capacity) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        long long n = *cloud_count;
        for (int f = 0; f < F; ++f) n += counts[2 * f + 1];
        *cloud_count = n < capacity ? n : capacity;
    }
}

// ------------------------------------------------------------------ rasteriser
constexpr int TILE = 8;            // 8x8 pixel tiles, one wave per tile
// a face of one frame as the three linear forms of the hit test in the pixel's ray (dx, dy, 1): det = a . d, u det = un . d, v det = q . d;
// z det = tnum (oracle/raster.py::plane_forms)
struct FaceRec { float a[3], un[3], q[3], tnum, pad[2]; };   // 48 bytes

// Two-level binning without capacity limits: fine tiles of 8x8 pixels (one wave), coarse tiles of 8x8 fine tiles.
// raster_setup_kernel (one thread per face and frame) transforms the face, clips its screen box against the near plane and
// appends (face id, fine-tile box) to the list of every coarse tile the box touches -- one wave-aggregated atomic per
// wave and coarse tile; a coarse list has room for every face, so nothing is ever dropped (the reference renders with
// max_faces_per_bin = 500000, macarons/testers/scene.py:440-446).  raster_tile_kernel: four waves per (2 x 2 block of fine tiles,
// list segment) scan the segment once, each wave ray-casts the faces whose box covers its tile and merges its 64 depths
// into the z-buffer with atomicMin on the float bit pattern (positive floats order like unsigned ints), so a tile with
// thousands of faces is shared by many waves instead of being one wave's tail (raster_block_body).
constexpr int COARSE = 8;              // fine tiles per coarse tile side
constexpr int SEG_DEFAULT = 16384;     // list entries per wave (NBP_RASTER_SEG): with the hit list taken through LDS in pieces a wave can walk a
                                       // whole coarse list, and the empty extra segments of shorter ones cost 1-3 us (4096: 50.3 / 65.1 us for 4 frames of
                                       // the 8 k / 29 k-face scenes, 32768: 49.0 / 62.9)
constexpr int HITS = 1024;             // ... taken through LDS in pieces of this many
constexpr unsigned ZBUF_EMPTY = 0x7F7F7F7Fu;   // memset pattern, 3.39e38 as a float

struct BinEntry { int face; unsigned box; };   // box = tx0 | tx1 << 8 | ty0 << 16 | ty1 << 24 (fine-tile coordinates)

__device__ __forceinline__ void raster_setup_body(unsigned bx, unsigned by, unsigned gx, unsigned gy, const float* __restrict__ verts, const int* __restrict__ faces,
                                                           int n_faces, const Cam* cams, int H, int W,
                                                           float tanh_fov, float zclip, FaceRec* __restrict__ recs,
                                                           int ctiles_x, int ctiles_y, int* __restrict__ ccount,
                                                           BinEntry* __restrict__ clist, unsigned* __restrict__ zbuf_bits,
                                                           unsigned long long* __restrict__ zface) {
    const int fr = by;
    const int fi = bx * blockDim.x + threadIdx.x;
    // z-buffer (or the (z, face) buffer of the colour path) of this frame = "empty": the tile kernel merges with atomicMin
    if (zface) for (int p = fi; p < H * W; p += gx * blockDim.x) zface[(size_t)fr * H * W + p] = ~0ull;
    else for (int p = fi; p < H * W; p += gx * blockDim.x) zbuf_bits[(size_t)fr * H * W + p] = ZBUF_EMPTY;
    bool have = false;
    int tx0 = 0, tx1 = 0, ty0 = 0, ty1 = 0;
    if (fi < n_faces) {
        const Cam cam = cams[fr];
        float v[3][3];
#pragma unroll
        for (int k = 0; k < 3; ++k) to_view(verts + 3 * (size_t)faces[3 * (size_t)fi + k], cam.R, cam.T, v[k]);
        if (!(v[0][2] <= zclip && v[1][2] <= zclip && v[2][2] <= zclip)) {          // not entirely behind the clip plane
            FaceRec r;
            float e1[3], e2[3];
            const float* v0 = v[0];
#pragma unroll
            for (int c = 0; c < 3; ++c) { e1[c] = v[1][c] - v[0][c]; e2[c] = v[2][c] - v[0][c]; }
            // q = e1 x v0  (= (-v0) x e1),  tnum = e2 . q;  a = -(e1 x e2),  un = v0 x e2: the hit test's numerators are linear in the ray
            r.q[0] = e1[1] * v0[2] - e1[2] * v0[1];
            r.q[1] = e1[2] * v0[0] - e1[0] * v0[2];
            r.q[2] = e1[0] * v0[1] - e1[1] * v0[0];
            r.tnum = (e2[0] * r.q[0] + e2[1] * r.q[1]) + e2[2] * r.q[2];
            r.a[0] = e1[2] * e2[1] - e1[1] * e2[2]; r.a[1] = e1[0] * e2[2] - e1[2] * e2[0]; r.a[2] = e1[1] * e2[0] - e1[0] * e2[1];
            r.un[0] = v0[1] * e2[2] - v0[2] * e2[1]; r.un[1] = v0[2] * e2[0] - v0[0] * e2[2]; r.un[2] = v0[0] * e2[1] - v0[1] * e2[0];
            r.pad[0] = r.pad[1] = 0.f;
            // screen bbox of the part with z >= zclip (Sutherland-Hodgman against one plane)
            const int s = H < W ? H : W;
            float cmin = 1e30f, cmax = -1e30f, rmin = 1e30f, rmax = -1e30f;
            auto emit = [&](float x, float y, float z) {
                const float nx = x / (z * tanh_fov), ny = y / (z * tanh_fov);
                const float col = ((float)W - (float)s * nx - 1.f) * 0.5f;   // ndc_x(col) = W/s - (2 col + 1)/s
                const float row = ((float)H - (float)s * ny - 1.f) * 0.5f;
                cmin = fminf(cmin, col); cmax = fmaxf(cmax, col); rmin = fminf(rmin, row); rmax = fmaxf(rmax, row);
            };
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float* a = v[k];
                const float* b = v[(k + 1) % 3];
                const bool ain = a[2] >= zclip, bin_ = b[2] >= zclip;
                if (ain) emit(a[0], a[1], a[2]);
                if (ain != bin_) {
                    const float t = (zclip - a[2]) / (b[2] - a[2]);
                    emit(a[0] + t * (b[0] - a[0]), a[1] + t * (b[1] - a[1]), zclip);
                }
            }
            if (cmax >= cmin) {
                int c0 = (int)floorf(fmaxf(cmin, -1e8f)) - 1, c1 = (int)ceilf(fminf(cmax, 1e8f)) + 1;
                int r0 = (int)floorf(fmaxf(rmin, -1e8f)) - 1, r1 = (int)ceilf(fminf(rmax, 1e8f)) + 1;
                c0 = max(c0, 0); r0 = max(r0, 0); c1 = min(c1, W - 1); r1 = min(r1, H - 1);
                if (c0 <= c1 && r0 <= r1) {
                    recs[(size_t)fr * n_faces + fi] = r;
                    have = true;
                    tx0 = c0 / TILE; tx1 = c1 / TILE; ty0 = r0 / TILE; ty1 = r1 / TILE;
                }
            }
        }
    }
    const int lane = threadIdx.x & 63;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const BinEntry e = {fi, (unsigned)tx0 | ((unsigned)tx1 << 8) | ((unsigned)ty0 << 16) | ((unsigned)ty1 << 24)};
    const int nct = ctiles_x * ctiles_y;
    // Up to 64 coarse tiles per pass: lane c first collects how many faces of this wave touch coarse tile c0 + c (ballots
    // only), reserves that many list slots with ONE atomic -- all lanes' atomics are in flight together; a returning
    // device-scope atomic per coarse tile in sequence was 10 us of pure latency -- then the entries are written.
    for (int c0 = 0; c0 < nct; c0 += 64) {
        const int npass = min(64, nct - c0);
        int my_cnt = 0;
        for (int c = 0; c < npass; ++c) {                                  // uniform loop: ballots need the whole wave
            const int cx = (c0 + c) % ctiles_x, cy = (c0 + c) / ctiles_x;
            const bool hit = have && tx0 <= cx * COARSE + COARSE - 1 && tx1 >= cx * COARSE && ty0 <= cy * COARSE + COARSE - 1 &&
                             ty1 >= cy * COARSE;
            const int n = __popcll(__ballot(hit));
            if (lane == c) my_cnt = n;
        }
        int my_base = 0;
        if (my_cnt > 0) my_base = atomicAdd(&ccount[fr * nct + c0 + lane], my_cnt);
        for (int c = 0; c < npass; ++c) {
            const int cx = (c0 + c) % ctiles_x, cy = (c0 + c) / ctiles_x;
            const bool hit = have && tx0 <= cx * COARSE + COARSE - 1 && tx1 >= cx * COARSE && ty0 <= cy * COARSE + COARSE - 1 &&
                             ty1 >= cy * COARSE;
            const unsigned long long bal = __ballot(hit);
            const int base = __shfl(my_base, c);
            if (hit) clist[((size_t)fr * nct + c0 + c) * n_faces + base + __popcll(bal & lt)] = e;
        }
    }
}

// The ray casts, with the coarse list scanned once per 2 x 2 block of fine tiles (round 5; until then every fine tile of a coarse
// tile walked the coarse tile's whole list for the entries whose box covers it -- 64 scans of the same list, 40 % of the launch by
// its own trace).  A workgroup of four waves owns a block of four fine tiles: together they scan the list segment ONCE for the entries
// whose box touches the block (compacted into LDS, HITS at a time), then each wave takes the block's entries 64 at a time, keeps
// those covering its own tile, gathers their face records through LDS and ray-casts.  The set of faces a pixel is tested against is
// the same, and the winner (nearest, lowest face id among equal depths) does not depend on the order: the same image bit for bit.
constexpr int RB = 2;                  // fine tiles per block side
__device__ __forceinline__ void raster_block_body(unsigned bx, unsigned by, const FaceRec* __restrict__ recs, int n_faces, int H, int W,
                                                  float tanh_fov, float zclip, int tiles_x, int tiles_y, int ctiles_x,
                                                  const int* __restrict__ ccount, const BinEntry* __restrict__ clist,
                                                  unsigned* __restrict__ zbuf_bits, unsigned long long* __restrict__ zface, int SEG) {
    __shared__ BinEntry bhits[HITS];
    __shared__ int nb_sh;
    __shared__ __attribute__((aligned(16))) FaceRec sh[4][64];
    __shared__ int shf[4][64];
    const int fr = by;
    const int blocks_x = (tiles_x + RB - 1) / RB, blocks_y = (tiles_y + RB - 1) / RB;
    const int nblocks = blocks_x * blocks_y;
    const int blk = bx % nblocks, seg = bx / nblocks;
    const int bxx = blk % blocks_x, byy = blk / blocks_x;
    const int nct = ctiles_x * ((tiles_y + COARSE - 1) / COARSE);
    const int ct = ((byy * RB) / COARSE) * ctiles_x + (bxx * RB) / COARSE;       // (COARSE is a multiple of RB: a block lies in one coarse tile)
    const int n = ccount[fr * nct + ct];
    const int s0 = seg * SEG;
    if (s0 >= n) return;
    const int s1 = min(n, s0 + SEG);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const BinEntry* lst = clist + ((size_t)fr * nct + ct) * n_faces;
    const int btx0 = bxx * RB, btx1 = btx0 + RB - 1, bty0 = byy * RB, bty1 = bty0 + RB - 1;
    const int tx = btx0 + (wave & 1), ty = bty0 + (wave >> 1);
    const bool live = tx < tiles_x && ty < tiles_y;                               // (an odd tile count leaves a block half empty)
    const int col = tx * TILE + (lane & 7), row = ty * TILE + (lane >> 3);
    const int s = H < W ? H : W;
    const float ndc_x = ((float)W - (2.f * col + 1.f)) / (float)s;
    const float ndc_y = ((float)H - (2.f * row + 1.f)) / (float)s;
    const float dx = ndc_x * tanh_fov, dy = ndc_y * tanh_fov;   // dz = 1
    const FaceRec* rb = recs + (size_t)fr * n_faces;
    float zbest = 3.0e38f;
    int fbest = -1;
    const float eps = 1e-6f;
    for (int c0 = s0; c0 < s1; c0 += HITS) {
        const int c1 = min(s1, c0 + HITS);
        if (tid == 0) nb_sh = 0;
        __syncthreads();
        // pass 1, the four waves together: entries of this piece whose fine-tile box touches the block
        for (int base = c0; base < c1; base += 256) {
            bool in = false;
            BinEntry e = {0, 0u};
            if (base + tid < c1) {
                e = lst[base + tid];
                in = btx1 >= (int)(e.box & 255u) && btx0 <= (int)((e.box >> 8) & 255u) && bty1 >= (int)((e.box >> 16) & 255u) &&
                     bty0 <= (int)(e.box >> 24);
            }
            const unsigned long long bal = __ballot(in);
            int wbase = 0;
            if (lane == 0 && bal) wbase = atomicAdd(&nb_sh, __popcll(bal));
            wbase = __shfl(wbase, 0);
            if (in) bhits[wbase + __popcll(bal & lt)] = e;
        }
        __syncthreads();
        const int nb = nb_sh;
        // pass 2, each wave for its tile: 64 block entries at a time -> those covering the tile -> their records through LDS
        if (live)
            for (int hb = 0; hb < nb; hb += 64) {
                bool in = false;
                int face = 0;
                if (hb + lane < nb) {
                    const BinEntry e = bhits[hb + lane];
                    face = e.face;
                    in = tx >= (int)(e.box & 255u) && tx <= (int)((e.box >> 8) & 255u) && ty >= (int)((e.box >> 16) & 255u) &&
                         ty <= (int)(e.box >> 24);
                }
                const unsigned long long bal = __ballot(in);
                const int m = __popcll(bal);
                if (in) {
                    const int k = __popcll(bal & lt);
                    sh[wave][k] = rb[face];
                    shf[wave][k] = face;
                }
                __builtin_amdgcn_s_waitcnt(0);          // (one wave: its own LDS writes are visible to itself once they have retired)
                __builtin_amdgcn_wave_barrier();
                for (int k = 0; k < m; ++k) {
                    const FaceRec& f = sh[wave][k];
                    const float det = (f.a[0] * dx + f.a[1] * dy) + f.a[2];
                    if (fabsf(det) < 1e-12f) continue;
                    const float inv = 1.f / det;
                    const float u = ((f.un[0] * dx + f.un[1] * dy) + f.un[2]) * inv;
                    const float vv = ((dx * f.q[0] + dy * f.q[1]) + f.q[2]) * inv;
                    const float z = f.tnum * inv;
                    if (u >= -eps && vv >= -eps && u + vv <= 1.f + eps && z > zclip) {
                        const int fid = shf[wave][k];
                        if (z < zbest || (z == zbest && fid < fbest)) { zbest = z; fbest = fid; }
                    }
                }
                __builtin_amdgcn_wave_barrier();        // the next 64 overwrite sh[wave]
            }
        __syncthreads();                                            // the next piece overwrites bhits[]
    }
    if (live && row < H && col < W && zbest < 1.0e38f) {
        const size_t pix = ((size_t)fr * H + row) * W + col;
        if (zface) atomicMin(&zface[pix], ((unsigned long long)__float_as_uint(zbest) << 32) | (unsigned)fbest);
        else atomicMin(&zbuf_bits[pix], __float_as_uint(zbest));
    }
}

__global__ __launch_bounds__(256) void raster_finalize_kernel(float* __restrict__ zbuf, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        if (!(zbuf[i] < 1.0e38f)) zbuf[i] = -1.f;
}

// Shading of Camera.capture_image (mu:2743-2763): SoftPhongShader under AmbientLights = ambient x texel, the texel being
// the barycentric interpolation of the winning face's vertex colours (TexturesVertex; perspective-correct barycentrics =
// those of the view-space ray cast), white background; then the z-buffer value.  One thread per pixel recomputes the
// barycentrics of its winning face from the face record.  gray_sum[frame] accumulates the luminance for adjust_contrast.
__device__ __forceinline__ void raster_shade_body(unsigned bx, unsigned by, unsigned gx, unsigned gy, const unsigned long long* __restrict__ zface,
                                                           const float* __restrict__ verts, const int* __restrict__ faces,
                                                           const float* __restrict__ vcolors, const Cam* cams, int H, int W,
                                                           float tanh_fov, float ambient, float* __restrict__ zbuf_or_null,
                                                           float* __restrict__ rgb_or_null, double* __restrict__ gray_sum) {
    const int fr = by;
    const int p = bx * blockDim.x + threadIdx.x;
    float gray = 0.f;
    if (p < H * W) {
        const size_t pix = (size_t)fr * H * W + p;
        const unsigned long long zf = zface[pix];
        float c[3] = {1.f, 1.f, 1.f};
        if (zbuf_or_null) zbuf_or_null[pix] = zf != ~0ull ? __uint_as_float((unsigned)(zf >> 32)) : -1.f;
        if (rgb_or_null) {
            if (zf != ~0ull) shade_pixel(verts, faces, vcolors, cams[fr], (int)(unsigned)zf, p / W, p % W, H, W, tanh_fov, ambient, c);
            rgb_or_null[3 * pix] = c[0]; rgb_or_null[3 * pix + 1] = c[1]; rgb_or_null[3 * pix + 2] = c[2];
            gray = (0.299f * c[0] + 0.587f * c[1]) + 0.114f * c[2];
        }
    }
    if (!gray_sum) return;
    double gs = (double)gray;
    for (int o = 32; o; o >>= 1) gs += __shfl_xor(gs, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(&gray_sum[fr], gs);
}

// ---- launch forms of the rasteriser's kernels (see the un-projection's above)
struct RasterItem {
    const float* verts; const int* faces; const float* vcolors; FaceRec* recs; int* ccount; BinEntry* clist; float* zbuf;
    unsigned long long* zface; int n_faces; unsigned gx_setup, gx_tile; Cam cam[4];
};
struct RasterBatch { RasterItem it[STEP_BATCH]; };

__global__ __launch_bounds__(256) void raster_setup_kernel(const float* __restrict__ verts, const int* __restrict__ faces, int n_faces,
                                                           CamSet cams, int H, int W, float tanh_fov, float zclip,
                                                           FaceRec* __restrict__ recs, int ctiles_x, int ctiles_y, int* __restrict__ ccount,
                                                           BinEntry* __restrict__ clist, unsigned* __restrict__ zbuf_bits,
                                                           unsigned long long* __restrict__ zface) {
    raster_setup_body(blockIdx.x, blockIdx.y, gridDim.x, gridDim.y, verts, faces, n_faces, cams.c, H, W, tanh_fov, zclip, recs, ctiles_x,
                      ctiles_y, ccount, clist, zbuf_bits, zface);
}
// (the batch's first launch also clears the coarse-tile counters of every item: blockIdx.y == gridDim.y - 1 is that pass, the
// counters are consumed by the NEXT launch's atomics, so no race)
__global__ __launch_bounds__(256) void raster_clear_batch_kernel(RasterBatch b, int n_counters) {
    int* c = b.it[blockIdx.y].ccount;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_counters; i += gridDim.x * blockDim.x) c[i] = 0;
}
__global__ __launch_bounds__(256) void raster_setup_batch_kernel(RasterBatch b, int H, int W, float tanh_fov, float zclip, int ctiles_x,
                                                                 int ctiles_y) {
    const RasterItem& a = b.it[blockIdx.z];
    if (blockIdx.x >= a.gx_setup) return;
    raster_setup_body(blockIdx.x, blockIdx.y, a.gx_setup, gridDim.y, a.verts, a.faces, a.n_faces, a.cam, H, W, tanh_fov, zclip, a.recs,
                      ctiles_x, ctiles_y, a.ccount, a.clist, nullptr, a.zface);
}
__global__ __launch_bounds__(256) void raster_tile_kernel(const FaceRec* __restrict__ recs, int n_faces, int H, int W, float tanh_fov,
                                                          float zclip, int tiles_x, int tiles_y, int ctiles_x, int ctiles_y,
                                                          const int* __restrict__ ccount, const BinEntry* __restrict__ clist,
                                                          unsigned* __restrict__ zbuf_bits, unsigned long long* __restrict__ zface, int SEG) {
    raster_block_body(blockIdx.x, blockIdx.y, recs, n_faces, H, W, tanh_fov, zclip, tiles_x, tiles_y, ctiles_x, ccount, clist, zbuf_bits, zface,
                      SEG);
}
__global__ __launch_bounds__(256) void raster_tile_batch_kernel(RasterBatch b, int H, int W, float tanh_fov, float zclip, int tiles_x,
                                                                int tiles_y, int ctiles_x, int ctiles_y, int SEG) {
    const RasterItem& a = b.it[blockIdx.z];
    if (blockIdx.x >= a.gx_tile) return;
    raster_block_body(blockIdx.x, blockIdx.y, a.recs, a.n_faces, H, W, tanh_fov, zclip, tiles_x, tiles_y, ctiles_x, a.ccount, a.clist, nullptr,
                      a.zface, SEG);
}
__global__ __launch_bounds__(256) void raster_shade_kernel(const unsigned long long* __restrict__ zface, const float* __restrict__ verts,
                                                           const int* __restrict__ faces, const float* __restrict__ vcolors, CamSet cams,
                                                           int H, int W, float tanh_fov, float ambient, float* __restrict__ zbuf_or_null,
                                                           float* __restrict__ rgb_or_null, double* __restrict__ gray_sum) {
    raster_shade_body(blockIdx.x, blockIdx.y, gridDim.x, gridDim.y, zface, verts, faces, vcolors, cams.c, H, W, tanh_fov, ambient, zbuf_or_null,
                      rgb_or_null, gray_sum);
}
// depth image out of the (depth, face) image: the zface path's last pass (no colours are evaluated here)
__global__ __launch_bounds__(256) void raster_depth_batch_kernel(RasterBatch b, int H, int W, float tanh_fov) {
    const RasterItem& a = b.it[blockIdx.z];
    raster_shade_body(blockIdx.x, blockIdx.y, gridDim.x, gridDim.y, a.zface, a.verts, a.faces, a.vcolors, a.cam, H, W, tanh_fov, 0.f, a.zbuf,
                      nullptr, nullptr);
}

// torchvision adjust_contrast (mu:2760): out = clamp(factor * img + (1 - factor) * mean(gray), 0, 1)
__global__ __launch_bounds__(256) void contrast_kernel(float* __restrict__ rgb, int HW3, const double* __restrict__ gray_sum,
                                                       int HW, float factor) {
    const int fr = blockIdx.y;
    const float mean = (float)(gray_sum[fr] / (double)HW);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW3; i += gridDim.x * blockDim.x) {
        const float v = factor * rgb[(size_t)fr * HW3 + i] + (1.f - factor) * mean;
        rgb[(size_t)fr * HW3 + i] = fminf(fmaxf(v, 0.f), 1.f);
    }
}

// ------------------------------------------------------------------ ray / mesh tests (world space)
// Ray o + t d, |d| = 1: returns t of the hit with triangle (a,b,c) or -1.
__device__ __forceinline__ float ray_tri(const float* o, const float* d, const float* a, const float* b, const float* c) {
    const float e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
    const float e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
    const float p[3] = {d[1] * e2[2] - d[2] * e2[1], d[2] * e2[0] - d[0] * e2[2], d[0] * e2[1] - d[1] * e2[0]};
    const float det = (e1[0] * p[0] + e1[1] * p[1]) + e1[2] * p[2];
    if (fabsf(det) < 1e-12f) return -1.f;
    const float inv = 1.f / det;
    const float tv[3] = {o[0] - a[0], o[1] - a[1], o[2] - a[2]};
    const float u = ((tv[0] * p[0] + tv[1] * p[1]) + tv[2] * p[2]) * inv;
    if (u < 0.f || u > 1.f) return -1.f;
    const float q[3] = {tv[1] * e1[2] - tv[2] * e1[1], tv[2] * e1[0] - tv[0] * e1[2], tv[0] * e1[1] - tv[1] * e1[0]};
    const float v = ((d[0] * q[0] + d[1] * q[1]) + d[2] * q[2]) * inv;
    if (v < 0.f || u + v > 1.f) return -1.f;
    const float t = ((e2[0] * q[0] + e2[1] * q[1]) + e2[2] * q[2]) * inv;
    return t > 0.f ? t : -1.f;
}

// grid: x over faces, y over segments.  hit[e] = 1 iff some triangle is hit at distance < |segment|.
__global__ __launch_bounds__(256) void segments_hit_kernel(const float* __restrict__ verts, const int* __restrict__ faces,
                                                           int n_faces, const float* __restrict__ segs, int* __restrict__ hit) {
    const int e = blockIdx.y;
    const int fi = blockIdx.x * blockDim.x + threadIdx.x;
    if (fi >= n_faces) return;
    const float* sg = segs + 6 * (size_t)e;
    float d[3] = {sg[3] - sg[0], sg[4] - sg[1], sg[5] - sg[2]};
    const float len = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
    if (!(len > 0.f)) return;
    d[0] /= len; d[1] /= len; d[2] /= len;
    const int* f = faces + 3 * (size_t)fi;
    const float t = ray_tri(sg, d, verts + 3 * (size_t)f[0], verts + 3 * (size_t)f[1], verts + 3 * (size_t)f[2]);
    if (t >= 0.f && t < len) atomicOr(&hit[e], 1);
}

// counts[k][axis] = number of triangles hit by the ray from pts[k] along +Y (0), +X (1), +Z (2).
__global__ __launch_bounds__(256) void axis_ray_count_kernel(const float* __restrict__ verts, const int* __restrict__ faces,
                                                             int n_faces, const float* __restrict__ pts, int* __restrict__ counts) {
    const int k = blockIdx.y;
    const int fi = blockIdx.x * blockDim.x + threadIdx.x;
    if (fi >= n_faces) return;
    const int* f = faces + 3 * (size_t)fi;
    const float* a = verts + 3 * (size_t)f[0];
    const float* b = verts + 3 * (size_t)f[1];
    const float* c = verts + 3 * (size_t)f[2];
    const float dirs[3][3] = {{0, 1, 0}, {1, 0, 0}, {0, 0, 1}};
#pragma unroll
    for (int ax = 0; ax < 3; ++ax)
        if (ray_tri(pts + 3 * (size_t)k, dirs[ax], a, b, c) >= 0.f) atomicAdd(&counts[3 * k + ax], 1);
}

// Camera.get_points_in_fov (mu:2849-2884): NDC inside the corners of the reference's NDC tables (mu:2270-2279), in front
// of the camera, closer than fov_range to the camera centre.  Also returns the view-space point and its NDC.
__device__ __forceinline__ bool point_in_fov(const float* p, const Cam& cam, int H, int W, float tanh_fov, float fov_range,
                                             float* v, float* nx_out, float* ny_out) {
    to_view(p, cam.R, cam.T, v);
    // camera centre C = -T R^T
    const float cx = -((cam.T[0] * cam.R[0] + cam.T[1] * cam.R[1]) + cam.T[2] * cam.R[2]);
    const float cy = -((cam.T[0] * cam.R[3] + cam.T[1] * cam.R[4]) + cam.T[2] * cam.R[5]);
    const float cz = -((cam.T[0] * cam.R[6] + cam.T[1] * cam.R[7]) + cam.T[2] * cam.R[8]);
    const float dx = p[0] - cx, dy = p[1] - cy, dz = p[2] - cz;
    const float dist = sqrtf((dx * dx + dy * dy) + dz * dz);
    const int s = H < W ? H : W;
    const float nx = v[0] / (v[2] * tanh_fov), ny = v[1] / (v[2] * tanh_fov);
    const float max_x = (float)((double)W / s), min_x = max_x - ((float)(W - 1) / (float)(s - 1)) * 2.f;
    const float max_y = (float)((double)H / s), min_y = max_y - ((float)(H - 1) / (float)(s - 1)) * 2.f;
    *nx_out = nx; *ny_out = ny;
    return nx >= min_x && nx <= max_x && ny >= min_y && ny <= max_y && v[2] > 0.f && dist < fov_range;
}

// mask[cam][i] = point i inside the field of view of camera cam (get_points_in_fov); any[cam] = some point is
// (is_fov_empty, mu:2672-2688, over the mesh vertices).  grid.y = camera.
__global__ __launch_bounds__(256) void points_in_fov_kernel(const float* __restrict__ pts, int P, CamSet cams, int H, int W,
                                                            float tanh_fov, float fov_range, unsigned char* __restrict__ mask,
                                                            int* __restrict__ any) {
    const int c = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool in = false;
    if (i < P) {
        const float p[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
        float v[3], nx, ny;
        in = point_in_fov(p, cams.c[c], H, W, tanh_fov, fov_range, v, &nx, &ny);
        if (mask) mask[(size_t)c * P + i] = in ? 1 : 0;
    }
    if (any && __ballot(in) && (threadIdx.x & 63) == 0) any[c] = 1;      // plain store: every writer writes 1
}

// ------------------------------------------------------------------ view-state vectors of the proxy points
// compute_view_state (macarons/utility/scone_utils.py:799-862): the direction from a proxy point to a camera, in spherical
// coordinates (get_spherical_coords, macarons/utility/CustomGeometry.py:27-45), rounded to the nearest of n_elev x n_azim
// directions; the point's vector [n_elev * n_azim] gets a 1 there.  fp32 with the reference's operation order (floor_divide =
// (x - x % d) / d, macarons/utility/utils.py:113-117, `%` = remainder with the divisor's sign); asin / acos / cos are this
// platform's, so a ray within a few ulp of a bin boundary may fall on the other side than under torch's.
__device__ __forceinline__ float pymodf(float x, float d) {
    float m = fmodf(x, d);
    if (m != 0.f && ((d < 0.f) != (m < 0.f))) m += d;
    return m;
}
__device__ __forceinline__ int view_bin(const float* p, const float* xc, int n_elev, int n_azim, float elev_step, float azim_step,
                                        float half_e, float half_a) {
    const float rx = xc[0] - p[0], ry = xc[1] - p[1], rz = xc[2] - p[2];
    const float r = sqrtf((rx * rx + ry * ry) + rz * rz);
    const float s = ry / r;
    float elev = asinf(s);
    if (s <= -1.f) elev = -1.57079632679489661923f;
    if (s >= 1.f) elev = 1.57079632679489661923f;
    const float c = rz / (r * cosf(elev));
    float azim = acosf(c);
    if (c <= -1.f) azim = 3.14159265358979323846f;
    if (c >= 1.f) azim = 0.f;
    if (rx < 0.f) azim *= -1.f;
    const float me = pymodf(elev, elev_step), ma = pymodf(azim, azim_step);
    float ie = (elev - me) / elev_step, ia = (azim - ma) / azim_step;
    if (me > half_e) ie += 1.f;
    if (ma > half_a) ia += 1.f;
    // floor division of the NEGATED counts, as Python's `-n_elev // 2` parses: (-n) // 2
    const int lo_e = -((n_elev + 1) / 2), lo_a = -((n_azim + 1) / 2);
    if (ie >= (float)n_elev) ie = (float)(n_elev - 1);
    if (ie < (float)lo_e) ie = (float)lo_e;
    if (ia > (float)(n_azim / 2)) ia = (float)lo_a;
    ie += (float)(n_elev / 2);
    if (ia < 0.f) ia += (float)n_azim;
    const long long ind = (long long)ie * n_azim + (long long)ia;
    const int nc = n_elev * n_azim;
    return (int)(((ind % nc) + nc) % nc);
}
struct ViewArgs { float x[3 * MAX_CAMS]; int n_view, n_elev, n_azim; float elev_step, azim_step, half_e, half_a; };

// one thread per (proxy point, camera); mask (or null) / sd < dist (sd or null) select the points, as
// Scene.update_proxy_view_states does (macarons_utils.py:3284-3306)
__global__ __launch_bounds__(256) void view_state_kernel(const float* __restrict__ pts, int P, const unsigned char* __restrict__ mask,
                                                         const float* __restrict__ sd, float dist, ViewArgs va,
                                                         float* __restrict__ view_states) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P || (mask && !mask[i]) || (sd && !(sd[i] < dist))) return;
    const float p[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
    const int nc = va.n_elev * va.n_azim;
    for (int v = 0; v < va.n_view; ++v)
        view_states[(size_t)i * nc + view_bin(p, va.x + 3 * v, va.n_elev, va.n_azim, va.elev_step, va.azim_step, va.half_e, va.half_a)] = 1.f;
}

// Geometric stand-in for the coverage gain of a candidate pose (the reference predicts it with the unreleased SCONE network,
// macarons/testers/scene.py:640-670): the number of proxy points that lie in the candidate's field of view, are still
// believed occupied (supervision occupancy 1) and have NOT been observed from the candidate's direction yet (their view-state
// bit for that direction is 0).  One thread per (proxy point, candidate), wave-reduced, one integer atomic per wave.
__global__ __launch_bounds__(256) void view_gain_kernel(const float* __restrict__ pts, int P, const float* __restrict__ occ,
                                                        const float* __restrict__ view_states, CamSet cams, ViewArgs va, int H,
                                                        int W, float tanh_fov, float fov_range, int* __restrict__ gains) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
    int hit = 0;
    if (i < P) {
        const float p[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
        float v[3], nx, ny;
        if (occ[i] > 0.5f && point_in_fov(p, cams.c[c], H, W, tanh_fov, fov_range, v, &nx, &ny)) {
            const int b = view_bin(p, va.x + 3 * c, va.n_elev, va.n_azim, va.elev_step, va.azim_step, va.half_e, va.half_a);
            hit = view_states[(size_t)i * (va.n_elev * va.n_azim) + b] == 0.f ? 1 : 0;
        }
    }
    const unsigned long long m = __ballot(hit);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(gains + c, __popcll(m));
}

// ------------------------------------------------------------------ depth-map space carving (A20)
// One thread per proxy point: frustum + range test, bilinear depth lookup (torch grid_sample
// semantics: align_corners = False, border padding; invalid pixels read as 1.1 zfar), counters.
__global__ __launch_bounds__(256) void carve_update_kernel(const float* __restrict__ pts, int P, const float* __restrict__ depth,
                                                           const unsigned char* __restrict__ mask, Cam cam, int H, int W,
                                                           float tanh_fov, float zfar, float fov_range, float tol,
                                                           float score_thr, float* __restrict__ n_inside,
                                                           float* __restrict__ n_behind, float* __restrict__ occ,
                                                           float* __restrict__ out_of_field, float* __restrict__ view_states,
                                                           float vs_dist, ViewArgs va, unsigned char* __restrict__ fov_mask,
                                                           float* __restrict__ sd_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const float p[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
    float v[3], nx, ny;
    const bool in_fov = point_in_fov(p, cam, H, W, tanh_fov, fov_range, v, &nx, &ny);
    const int s = H < W ? H : W;
    if (fov_mask) fov_mask[i] = in_fov ? 1 : 0;
    if (!in_fov) return;
    // grid_sample coordinates (mu:2929-2944): gx = -(s/W) ndc_x, gy = -(s/H) ndc_y
    const float gx = (-(float)s / (float)W) * nx, gy = (-(float)s / (float)H) * ny;
    float ix = ((gx + 1.f) * (float)W - 1.f) * 0.5f, iy = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
    ix = fminf(fmaxf(ix, 0.f), (float)(W - 1));
    iy = fminf(fmaxf(iy, 0.f), (float)(H - 1));
    const int x0 = (int)floorf(ix), y0 = (int)floorf(iy);
    const int x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
    const float wx = ix - (float)x0, wy = iy - (float)y0;
    auto at = [&](int y, int x) {
        const float d = depth[y * W + x];
        const bool ok = mask ? mask[y * W + x] != 0 : d > -1.f;
        return ok ? d : 1.1f * zfar;
    };
    const float d00 = at(y0, x0), d01 = at(y0, x1), d10 = at(y1, x0), d11 = at(y1, x1);
    const float dsamp = (d00 * (1.f - wx) * (1.f - wy) + d01 * wx * (1.f - wy)) + (d10 * (1.f - wx) * wy + d11 * wx * wy);
    const float sd = v[2] - dsamp;
    const float ni = n_inside[i] + 1.f;
    const float nb = n_behind[i] + (sd >= -tol ? 1.f : 0.f);
    n_inside[i] = ni; n_behind[i] = nb;
    occ[i] = (nb / ni >= score_thr) ? 1.f : 0.f;
    out_of_field[i] = 0.f;
    if (sd_out) sd_out[i] = sd;
    // Scene.update_proxy_view_states as the NBV driver calls it BEFORE the occupancy update (macarons/testers/scene.py:598-607;
    // the two touch different state, so the order is immaterial): points in the field of view whose signed distance is below
    // vs_dist gain the direction towards the camera
    if (view_states && sd < vs_dist)
        view_states[(size_t)i * (va.n_elev * va.n_azim) +
                    view_bin(p, va.x, va.n_elev, va.n_azim, va.elev_step, va.azim_step, va.half_e, va.half_a)] = 1.f;
}

}  // namespace

// ================================================================== C ABI
extern "C" size_t nbp_unproject_workspace_bytes(int n_frames, int H, int W) {
    if (n_frames < 1 || H < 2 || W < 2) return 0;
    const size_t nblk = (size_t)nbp_cdiv((long long)H * W, COMPACT_CHUNK);
    return (size_t)n_frames * H * W * sizeof(unsigned) + ((size_t)n_frames * nblk * sizeof(int) + 255) / 256 * 256 + 512;
}

static CamSet camset_from_host(const float* cams12_host, int n) {
    CamSet cs;
    for (int f = 0; f < MAX_CAMS; ++f)
        for (int k = 0; k < 12; ++k) {
            const float v = f < n ? cams12_host[12 * f + k] : 0.f;
            if (k < 9) cs.c[f].R[k] = v; else cs.c[f].T[k - 9] = v;
        }
    return cs;
}

static int ticket_fence() { constexpr int v = 0; return v; }
struct ShadeSrc { const unsigned long long* zface; const float* verts; const int* faces; const float* vcolors; float ambient; };
static int unproject_launch(const float* depth, const unsigned char* mask_or_null, const float* cams12_host, int n_frames, int H,
                            int W, float tan_half_fov, float fov_range, double gathering_factor, unsigned seed, int* counts2,
                            float* cloud, long long* cloud_count, long long capacity, void* ws, size_t ws_bytes, void* stream,
                            const float* rgb_or_null, float* cloud_rgb_or_null, ShadeSrc sh = ShadeSrc{nullptr, nullptr, nullptr, nullptr, 0.f},
                            BinFile bf = BinFile{nullptr, nullptr, nullptr, 0});

extern "C" int nbp_unproject_append_f32(const float* depth, const unsigned char* mask_or_null, const float* cams12_host,
                                        int n_frames, int H, int W, float tan_half_fov, float fov_range,
                                        double gathering_factor, unsigned seed, int* counts2, float* cloud,
                                        long long* cloud_count, long long capacity, void* ws, size_t ws_bytes,
                                        void* stream) {
    NBP_ENTER();
    return unproject_launch(depth, mask_or_null, cams12_host, n_frames, H, W, tan_half_fov, fov_range, gathering_factor, seed,
                            counts2, cloud, cloud_count, capacity, ws, ws_bytes, stream, nullptr, nullptr);
}

extern "C" int nbp_unproject_append_rgb_f32(const float* depth, const unsigned char* mask_or_null, const float* rgb,
                                            const float* cams12_host, int n_frames, int H, int W, float tan_half_fov,
                                            float fov_range, double gathering_factor, unsigned seed, int* counts2, float* cloud,
                                            float* cloud_rgb, long long* cloud_count, long long capacity, void* ws,
                                            size_t ws_bytes, void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(!rgb || !cloud_rgb, NBP_E_ARG);
    return unproject_launch(depth, mask_or_null, cams12_host, n_frames, H, W, tan_half_fov, fov_range, gathering_factor, seed,
                            counts2, cloud, cloud_count, capacity, ws, ws_bytes, stream, rgb, cloud_rgb);
}

extern "C" int nbp_unproject_append_shaded_f32(const float* depth, const unsigned char* mask_or_null, const void* zface,
                                               const float* verts, const int* faces, const float* vcolors3,
                                               const float* cams12_host, int n_frames, int H, int W, float tan_half_fov,
                                               float fov_range, double gathering_factor, unsigned seed, float ambient,
                                               int* counts2, float* cloud, float* cloud_rgb, long long* cloud_count,
                                               long long capacity, void* ws, size_t ws_bytes, void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(!zface || !verts || !faces || !vcolors3 || !cloud_rgb, NBP_E_ARG);
    return unproject_launch(depth, mask_or_null, cams12_host, n_frames, H, W, tan_half_fov, fov_range, gathering_factor, seed,
                            counts2, cloud, cloud_count, capacity, ws, ws_bytes, stream, nullptr, cloud_rgb,
                            ShadeSrc{(const unsigned long long*)zface, verts, faces, vcolors3, ambient});
}

// nbp_unproject_append_f32 / _rgb_f32 / _shaded_f32 (by which colour source is given: none, rgb, or zface + verts + faces + vcolors3)
// that also FILES the appended points into the cloud's tile-binned store (nbp_cloud_bins_init) and, when zero6 / zero1 are given,
// clears the 6 S^2 / S^2 floats the map build behind it accumulates into: that build is then nbp_step_maps_prefiled_f32, one launch.
extern "C" int nbp_unproject_append_filed_f32(const float* depth, const unsigned char* mask_or_null, const float* rgb_or_null,
                                              const void* zface_or_null, const float* verts, const int* faces, const float* vcolors3,
                                              const float* cams12_host, int n_frames, int H, int W, float tan_half_fov,
                                              float fov_range, double gathering_factor, unsigned seed, float ambient, int* counts2,
                                              float* cloud, float* cloud_rgb_or_null, long long* cloud_count, long long capacity,
                                              void* bins_store, float* zero6_or_null, float* zero1_or_null, int S, void* ws,
                                              size_t ws_bytes, void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(!bins_store || ((uintptr_t)bins_store & 255), NBP_E_ARG);
    NBP_RETURN_IF(rgb_or_null && zface_or_null, NBP_E_ARG);
    NBP_RETURN_IF((rgb_or_null || zface_or_null) && !cloud_rgb_or_null, NBP_E_ARG);
    NBP_RETURN_IF(zface_or_null && (!verts || !faces || !vcolors3), NBP_E_ARG);
    const long long SS = (long long)S * S;
    if (zero6_or_null || zero1_or_null)
        NBP_RETURN_IF(S < 1 || SS % 4 != 0 || 6 * SS >= (1ll << 31) || ((uintptr_t)zero6_or_null & 15) || ((uintptr_t)zero1_or_null & 15), NBP_E_SHAPE);
    const bool colours = rgb_or_null || zface_or_null;
    return unproject_launch(depth, mask_or_null, cams12_host, n_frames, H, W, tan_half_fov, fov_range, gathering_factor, seed,
                            counts2, cloud, cloud_count, capacity, ws, ws_bytes, stream, rgb_or_null, colours ? cloud_rgb_or_null : nullptr,
                            ShadeSrc{(const unsigned long long*)zface_or_null, verts, faces, vcolors3, ambient},
                            BinFile{(char*)bins_store, zero6_or_null, zero1_or_null, (int)SS});
}

static int unproject_launch(const float* depth, const unsigned char* mask_or_null, const float* cams12_host, int n_frames, int H,
                            int W, float tan_half_fov, float fov_range, double gathering_factor, unsigned seed, int* counts2,
                            float* cloud, long long* cloud_count, long long capacity, void* ws, size_t ws_bytes, void* stream,
                            const float* rgb_or_null, float* cloud_rgb_or_null, ShadeSrc sh, BinFile bf) {
    NBP_ENTER();
    NBP_RETURN_IF(!depth || !cams12_host || !counts2 || !cloud || !cloud_count || !ws, NBP_E_ARG);
    NBP_RETURN_IF(n_frames < 1 || n_frames > MAX_CAMS || H < 2 || W < 2 || capacity < 1, NBP_E_ARG);
    NBP_RETURN_IF(!(gathering_factor >= 0.0 && gathering_factor <= 1.0), NBP_E_ARG);
    NBP_RETURN_IF(ws_bytes < nbp_unproject_workspace_bytes(n_frames, H, W), NBP_E_WS);
    hipStream_t st = (hipStream_t)stream;
    const int HW = H * W, nblk = (int)nbp_cdiv(HW, COMPACT_CHUNK);
    NBP_RETURN_IF(nblk > 4096, NBP_E_SHAPE);
    int* blk_count = (int*)(((uintptr_t)ws + 255) / 256 * 256);
    unsigned* list = (unsigned*)((char*)blk_count + ((size_t)n_frames * nblk * sizeof(int) + 255) / 256 * 256);
    const CamSet cams = camset_from_host(cams12_host, n_frames);   // [F][12]: R row-major (9) then T (3)
    const int max_keep = (int)((double)H * W * gathering_factor) + 1;
    dim3 grid((unsigned)nbp_cdiv(max_keep, 256), (unsigned)n_frames);
    int rc;
    const bool fast = (HW & 3) == 0 && ((uintptr_t)depth & 15) == 0 && (!mask_or_null || ((uintptr_t)mask_or_null & 3) == 0);
    // filing rides on the three-launch form (its last workgroup advances the store with the cloud)
    NBP_RETURN_IF(bf.store && !fast, NBP_E_SHAPE);
    if (fast) {
        // three launches: chunk counts (also resets the ticket), ordered compaction, gather + append + size update
        const int nb4 = (int)nbp_cdiv(HW, FAST_CHUNK);             // <= nblk: the workspace is sized for 2048-pixel chunks
        int* ticket = blk_count + (size_t)n_frames * nb4;
        dim3 g4((unsigned)nb4, (unsigned)n_frames);
        unproject_count4_kernel<<<g4, 256, 0, st>>>(depth, mask_or_null, HW, nb4, fov_range, blk_count, ticket);
        if ((rc = nbp_launch_status())) return rc;
        unproject_compact4_kernel<<<g4, 256, 0, st>>>(depth, mask_or_null, HW, nb4, fov_range, gathering_factor, blk_count, list,
                                                      counts2);
        if ((rc = nbp_launch_status())) return rc;
        if (bf.store && (bf.zero6 || bf.zero1) && grid.x * grid.y < 64u) grid.x = (64u + grid.y - 1) / grid.y;      // enough workgroups for the clear
        unproject_append_kernel<<<grid, 256, 0, st>>>(depth, cams, H, W, tan_half_fov, seed, list, counts2, cloud, cloud_count,
                                                      capacity, n_frames, ticket, rgb_or_null, cloud_rgb_or_null, sh.zface, sh.verts, sh.faces,
                                                      sh.vcolors, sh.ambient, ticket_fence(), bf);
        return nbp_launch_status();
    }
    dim3 gc((unsigned)nblk, (unsigned)n_frames);
    unproject_count_kernel<<<gc, 256, 0, st>>>(depth, mask_or_null, HW, nblk, fov_range, blk_count);
    if ((rc = nbp_launch_status())) return rc;
    unproject_compact_kernel<<<gc, 256, 0, st>>>(depth, mask_or_null, HW, nblk, fov_range, gathering_factor, blk_count,
                                                 list, counts2);
    if ((rc = nbp_launch_status())) return rc;
    unproject_append_kernel<<<grid, 256, 0, st>>>(depth, cams, H, W, tan_half_fov, seed, list, counts2, cloud, cloud_count,
                                                  capacity, n_frames, nullptr, rgb_or_null, cloud_rgb_or_null, sh.zface, sh.verts, sh.faces,
                                                  sh.vcolors, sh.ambient, 0, BinFile{nullptr, nullptr, nullptr, 0});
    if ((rc = nbp_launch_status())) return rc;
    cloud_count_update_kernel<<<1, 64, 0, st>>>(counts2, n_frames, cloud_count, capacity);
    return nbp_launch_status();
}

static size_t raster_ws_bytes(int n_faces, int n_frames, int H, int W, bool rgb) {
    if (n_faces < 1 || n_frames < 1 || H < 1 || W < 1) return 0;
    const size_t nct = (size_t)nbp_cdiv(nbp_cdiv(W, TILE), COARSE) * nbp_cdiv(nbp_cdiv(H, TILE), COARSE) * n_frames;
    size_t b = 256;
    b += ((size_t)n_frames * n_faces * sizeof(FaceRec) + 255) / 256 * 256;
    b += (nct * sizeof(int) + 255) / 256 * 256;
    b += (nct * n_faces * sizeof(BinEntry) + 255) / 256 * 256;
    if (rgb) b += ((size_t)n_frames * H * W * 8 + 255) / 256 * 256 + 256;
    return b + 256;
}

extern "C" size_t nbp_raster_workspace_bytes(int n_faces, int n_frames, int H, int W, int bin_cap) {
    (void)bin_cap;
    return raster_ws_bytes(n_faces, n_frames, H, W, false);
}

extern "C" size_t nbp_raster_rgb_workspace_bytes(int n_faces, int n_frames, int H, int W) {
    return raster_ws_bytes(n_faces, n_frames, H, W, true);
}

static int raster_launch(const float* verts, int n_verts, const int* faces, int n_faces, const float* cams12_host, int n_frames,
                         int H, int W, float tan_half_fov, float z_clip, float* zbuf, const float* vcolors, float ambient,
                         float contrast, float* rgb, void* ws, size_t ws_bytes, void* stream,
                         unsigned long long* zface_out = nullptr) {
    NBP_RETURN_IF(!verts || !faces || !cams12_host || !zbuf || !ws, NBP_E_ARG);
    NBP_RETURN_IF(n_verts < 3 || n_faces < 1 || n_frames < 1 || n_frames > MAX_CAMS || H < 1 || W < 1, NBP_E_ARG);
    const int tiles_x = (int)nbp_cdiv(W, TILE), tiles_y = (int)nbp_cdiv(H, TILE);
    NBP_RETURN_IF(tiles_x > 256 || tiles_y > 256, NBP_E_SHAPE);          // fine-tile coordinates are packed in 8 bits
    NBP_RETURN_IF(ws_bytes < raster_ws_bytes(n_faces, n_frames, H, W, rgb != nullptr && !zface_out), NBP_E_WS);
    hipStream_t st = (hipStream_t)stream;
    const int ctiles_x = (int)nbp_cdiv(tiles_x, COARSE), ctiles_y = (int)nbp_cdiv(tiles_y, COARSE);
    const size_t nct = (size_t)ctiles_x * ctiles_y * n_frames;
    char* p = (char*)(((uintptr_t)ws + 255) / 256 * 256);
    FaceRec* recs = (FaceRec*)p; p += ((size_t)n_frames * n_faces * sizeof(FaceRec) + 255) / 256 * 256;
    int* ccount = (int*)p; p += (nct * sizeof(int) + 255) / 256 * 256;
    BinEntry* clist = (BinEntry*)p; p += (nct * n_faces * sizeof(BinEntry) + 255) / 256 * 256;
    unsigned long long* zface = nullptr;
    double* gray = nullptr;
    const long long npx = (long long)n_frames * H * W;
    if (zface_out) {
        zface = zface_out;
        gray = (double*)p;                                     // 8 doubles fit the workspace's slack
    } else if (rgb) {
        zface = (unsigned long long*)p; p += ((size_t)npx * 8 + 255) / 256 * 256;
        gray = (double*)p;
    }
    hipError_t e = hipMemsetAsync(ccount, 0, nct * sizeof(int), st);
    if (e != hipSuccess) return (int)e;
    dim3 g1((unsigned)nbp_cdiv(n_faces, 256), (unsigned)n_frames);
    raster_setup_kernel<<<g1, 256, 0, st>>>(verts, faces, n_faces, camset_from_host(cams12_host, n_frames), H, W,
                                            tan_half_fov, z_clip, recs, ctiles_x, ctiles_y, ccount, clist, (unsigned*)zbuf, zface);
    int rc = nbp_launch_status();
    if (rc) return rc;
    const bool need_gray = rgb && contrast != 1.f;            // adjust_contrast(1) is the identity: no luminance sum needed
    if (need_gray) {
        e = hipMemsetAsync(gray, 0, (size_t)n_frames * sizeof(double), st);
        if (e != hipSuccess) return (int)e;
    }
    constexpr int SEG = SEG_DEFAULT;
    const int nseg = (int)nbp_cdiv(n_faces, SEG);
    dim3 g2((unsigned)((int)nbp_cdiv(tiles_x, RB) * (int)nbp_cdiv(tiles_y, RB) * nseg), (unsigned)n_frames);
    raster_tile_kernel<<<g2, 256, 0, st>>>(recs, n_faces, H, W, tan_half_fov, z_clip, tiles_x, tiles_y, ctiles_x, ctiles_y,
                                          ccount, clist, (unsigned*)zbuf, zface, SEG);
    if ((rc = nbp_launch_status())) return rc;
    if (!zface) {
        raster_finalize_kernel<<<nbp_ew_grid(npx, 256), 256, 0, st>>>(zbuf, npx);
        return nbp_launch_status();
    }
    dim3 g3((unsigned)nbp_cdiv((long long)H * W, 256), (unsigned)n_frames);
    raster_shade_kernel<<<g3, 256, 0, st>>>(zface, verts, faces, vcolors, camset_from_host(cams12_host, n_frames), H, W,
                                            tan_half_fov, ambient, zbuf, rgb, need_gray ? gray : nullptr);
    if ((rc = nbp_launch_status())) return rc;
    if (need_gray) {
        dim3 g4((unsigned)nbp_ew_grid((long long)H * W * 3, 256), (unsigned)n_frames);
        contrast_kernel<<<g4, 256, 0, st>>>(rgb, H * W * 3, gray, H * W, contrast);
        rc = nbp_launch_status();
    }
    return rc;
}

extern "C" int nbp_raster_zbuf_f32(const float* verts, int n_verts, const int* faces, int n_faces, const float* cams12_host,
                                   int n_frames, int H, int W, float tan_half_fov, float z_clip, int bin_cap, float* zbuf,
                                   int* overflow_flag, void* ws, size_t ws_bytes, void* stream) {
    NBP_ENTER();
    (void)bin_cap; (void)overflow_flag;                      // kept for ABI compatibility: the lists cannot overflow
    return raster_launch(verts, n_verts, faces, n_faces, cams12_host, n_frames, H, W, tan_half_fov, z_clip, zbuf, nullptr, 0.f,
                         1.f, nullptr, ws, ws_bytes, stream);
}

extern "C" int nbp_raster_rgbz_f32(const float* verts, int n_verts, const int* faces, int n_faces, const float* vcolors3,
                                   const float* cams12_host, int n_frames, int H, int W, float tan_half_fov, float z_clip,
                                   float ambient, float contrast_factor, float* zbuf, float* rgb, void* ws, size_t ws_bytes,
                                   void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(!vcolors3 || !rgb, NBP_E_ARG);
    return raster_launch(verts, n_verts, faces, n_faces, cams12_host, n_frames, H, W, tan_half_fov, z_clip, zbuf, vcolors3,
                         ambient, contrast_factor, rgb, ws, ws_bytes, stream);
}

extern "C" int nbp_raster_zface_f32(const float* verts, int n_verts, const int* faces, int n_faces, const float* cams12_host,
                                    int n_frames, int H, int W, float tan_half_fov, float z_clip, float* zbuf, void* zface,
                                    void* ws, size_t ws_bytes, void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(!zface, NBP_E_ARG);
    return raster_launch(verts, n_verts, faces, n_faces, cams12_host, n_frames, H, W, tan_half_fov, z_clip, zbuf, nullptr, 0.f,
                         1.f, nullptr, ws, ws_bytes, stream, (unsigned long long*)zface);
}

// ---- the step's simulator stages for the n <= 12 rollouts of a lock-step group, one launch per kernel instead of n.
// Every array argument is a HOST array of n entries (device pointers inside).  Results are identical to n single calls.
//
// nbp_unproject_append_shaded_batch_f32: item r un-projects its n_frames (<= 4) depth frames depth[r n_frames + f] ([H][W] each: one
// pointer per frame, the frames of a ring need not be adjacent; with the (depth, face) images zface[r n_frames + f] for the
// colours; zface / verts / faces / vcolors / cloud_rgb entries may all be NULL: depth only)
// and appends the sub-sample to cloud[r] at *cloud_count[r]; ws[r] >= nbp_unproject_workspace_bytes(n_frames, H, W) + 256 each
// (counts2[r] = 2 n_frames ints of scratch that receive (valid, kept) per frame).
extern "C" int nbp_unproject_append_shaded_batch_f32(int n, const float* const* depth, const void* const* zface, const float* const* verts,
                                                     const int* const* faces, const float* const* vcolors, const float* cams12_host,
                                                     int n_frames, int H, int W, float tan_half_fov, float fov_range,
                                                     double gathering_factor, const unsigned* seeds, float ambient, int* const* counts2,
                                                     float* const* cloud, float* const* cloud_rgb, long long* const* cloud_count,
                                                     const long long* capacity, void* const* ws, size_t ws_bytes_each, void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(n < 1 || n > STEP_BATCH || !depth || !cams12_host || !seeds || !counts2 || !cloud || !cloud_count || !capacity || !ws, NBP_E_ARG);
    NBP_RETURN_IF(n_frames < 1 || n_frames > 4 || H < 2 || W < 2 || !(gathering_factor >= 0.0 && gathering_factor <= 1.0), NBP_E_ARG);
    NBP_RETURN_IF(ws_bytes_each < nbp_unproject_workspace_bytes(n_frames, H, W), NBP_E_WS);
    const int HW = H * W;
    NBP_RETURN_IF((HW & 3) != 0, NBP_E_SHAPE);                       // the batched form is the 4-pixels-per-lane path
    hipStream_t st = (hipStream_t)stream;
    const int nblk = (int)nbp_cdiv(HW, COMPACT_CHUNK), nb4 = (int)nbp_cdiv(HW, FAST_CHUNK);
    NBP_RETURN_IF(nblk > 4096, NBP_E_SHAPE);
    UnprojBatch b;
    b.ticket_off = n_frames * nb4;
    b.list_off_bytes = (unsigned)(((size_t)n_frames * nblk * sizeof(int) + 255) / 256 * 256);
    for (int r = 0; r < STEP_BATCH; ++r) {
        const int q = r < n ? r : 0;
        NBP_RETURN_IF(!counts2[q] || !cloud[q] || !cloud_count[q] || capacity[q] < 1 || !ws[q], NBP_E_ARG);
        UnprojItem& a = b.it[r];
        int* blk_count = (int*)(((uintptr_t)ws[q] + 255) / 256 * 256);
        a.blk_count = blk_count;
        a.counts = counts2[q]; a.cloud = cloud[q]; a.cloud_count = cloud_count[q]; a.capacity = capacity[q];
        const bool col = zface && zface[(size_t)q * n_frames] && cloud_rgb && cloud_rgb[q] && verts && faces && vcolors;
        a.cloud_rgb = col ? cloud_rgb[q] : nullptr;
        for (int f = 0; f < 4; ++f) {
            const int g = f < n_frames ? f : 0;
            a.depth[f] = depth[(size_t)q * n_frames + g];
            NBP_RETURN_IF(!a.depth[f] || ((uintptr_t)a.depth[f] & 15) != 0, NBP_E_ARG);
            a.zface[f] = col ? (const unsigned long long*)zface[(size_t)q * n_frames + g] : nullptr;
        }
        a.verts = col ? verts[q] : nullptr; a.faces = col ? faces[q] : nullptr; a.vcolors = col ? vcolors[q] : nullptr;
        a.seed = seeds[q];
        for (int f = 0; f < 4; ++f)
            for (int k = 0; k < 12; ++k) {
                const float v = f < n_frames ? cams12_host[((size_t)q * n_frames + f) * 12 + k] : 0.f;
                if (k < 9) a.cam[f].R[k] = v; else a.cam[f].T[k - 9] = v;
            }
    }
    dim3 g4((unsigned)nb4, (unsigned)n_frames, (unsigned)n);
    unproject_count4_batch_kernel<<<g4, 256, 0, st>>>(b, HW, nb4, fov_range);
    int rc = nbp_launch_status();
    if (rc) return rc;
    unproject_compact4_batch_kernel<<<g4, 256, 0, st>>>(b, HW, nb4, fov_range, gathering_factor);
    if ((rc = nbp_launch_status())) return rc;
    const int max_keep = (int)((double)H * W * gathering_factor) + 1;
    dim3 grid((unsigned)nbp_cdiv(max_keep, 256), (unsigned)n_frames, (unsigned)n);
    unproject_append_batch_kernel<<<grid, 256, 0, st>>>(b, H, W, tan_half_fov, n_frames, ambient, ticket_fence());
    return nbp_launch_status();
}

// nbp_raster_zface_batch_f32: item r renders n_frames (<= 4) views of ITS mesh (verts[r], faces[r], n_faces[r]) into zbuf[r]
// [n_frames][H][W] and zface[r] (the (depth, face) images); ws[r] >= nbp_raster_workspace_bytes(n_faces[r], n_frames, H, W, 0).
// Four launches for the group (counter clear, setup, tiles, depth images) instead of 4 n.
extern "C" int nbp_raster_zface_batch_f32(int n, const float* const* verts, const int* n_verts, const int* const* faces, const int* n_faces,
                                          const float* cams12_host, int n_frames, int H, int W, float tan_half_fov, float z_clip,
                                          float* const* zbuf, void* const* zface, void* const* ws, const size_t* ws_bytes, void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(n < 1 || n > STEP_BATCH || !verts || !n_verts || !faces || !n_faces || !cams12_host || !zbuf || !zface || !ws || !ws_bytes,
                  NBP_E_ARG);
    NBP_RETURN_IF(n_frames < 1 || n_frames > 4 || H < 1 || W < 1, NBP_E_ARG);
    const int tiles_x = (int)nbp_cdiv(W, TILE), tiles_y = (int)nbp_cdiv(H, TILE);
    NBP_RETURN_IF(tiles_x > 256 || tiles_y > 256, NBP_E_SHAPE);
    hipStream_t st = (hipStream_t)stream;
    const int ctiles_x = (int)nbp_cdiv(tiles_x, COARSE), ctiles_y = (int)nbp_cdiv(tiles_y, COARSE);
    const size_t nct = (size_t)ctiles_x * ctiles_y * n_frames;
    constexpr int SEG = SEG_DEFAULT;
    RasterBatch b;
    unsigned g_setup = 1, g_tile = 1;
    for (int r = 0; r < STEP_BATCH; ++r) {
        const int q = r < n ? r : 0;
        NBP_RETURN_IF(!verts[q] || !faces[q] || n_verts[q] < 3 || n_faces[q] < 1 || !zbuf[q] || !zface[q] || !ws[q], NBP_E_ARG);
        NBP_RETURN_IF(ws_bytes[q] < raster_ws_bytes(n_faces[q], n_frames, H, W, false), NBP_E_WS);
        RasterItem& a = b.it[r];
        char* p = (char*)(((uintptr_t)ws[q] + 255) / 256 * 256);
        a.recs = (FaceRec*)p; p += ((size_t)n_frames * n_faces[q] * sizeof(FaceRec) + 255) / 256 * 256;
        a.ccount = (int*)p; p += (nct * sizeof(int) + 255) / 256 * 256;
        a.clist = (BinEntry*)p;
        a.verts = verts[q]; a.faces = faces[q]; a.vcolors = nullptr; a.zbuf = zbuf[q]; a.zface = (unsigned long long*)zface[q];
        a.n_faces = n_faces[q];
        a.gx_setup = r < n ? (unsigned)nbp_cdiv(n_faces[q], 256) : 0;
        a.gx_tile = r < n ? (unsigned)((int)nbp_cdiv(tiles_x, RB) * (int)nbp_cdiv(tiles_y, RB) * (int)nbp_cdiv(n_faces[q], SEG)) : 0;
        if (a.gx_setup > g_setup) g_setup = a.gx_setup;
        if (a.gx_tile > g_tile) g_tile = a.gx_tile;
        for (int f = 0; f < 4; ++f)
            for (int k = 0; k < 12; ++k) {
                const float v = f < n_frames ? cams12_host[((size_t)q * n_frames + f) * 12 + k] : 0.f;
                if (k < 9) a.cam[f].R[k] = v; else a.cam[f].T[k - 9] = v;
            }
    }
    raster_clear_batch_kernel<<<dim3(1, (unsigned)n), 256, 0, st>>>(b, (int)nct);
    int rc = nbp_launch_status();
    if (rc) return rc;
    raster_setup_batch_kernel<<<dim3(g_setup, (unsigned)n_frames, (unsigned)n), 256, 0, st>>>(b, H, W, tan_half_fov, z_clip, ctiles_x, ctiles_y);
    if ((rc = nbp_launch_status())) return rc;
    raster_tile_batch_kernel<<<dim3(g_tile, (unsigned)n_frames, (unsigned)n), 256, 0, st>>>(b, H, W, tan_half_fov, z_clip, tiles_x, tiles_y,
                                                                                          ctiles_x, ctiles_y, SEG);
    if ((rc = nbp_launch_status())) return rc;
    raster_depth_batch_kernel<<<dim3((unsigned)nbp_cdiv((long long)H * W, 256), (unsigned)n_frames, (unsigned)n), 256, 0, st>>>(b, H, W,
                                                                                                                          tan_half_fov);
    return nbp_launch_status();
}

extern "C" int nbp_shade_image_f32(const void* zface, const float* verts, const int* faces, const float* vcolors3,
                                   const float* cams12_host, int n_frames, int H, int W, float tan_half_fov, float ambient,
                                   float contrast_factor, float* rgb, void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(!zface || !verts || !faces || !vcolors3 || !cams12_host || !rgb, NBP_E_ARG);
    NBP_RETURN_IF(n_frames < 1 || n_frames > MAX_CAMS || H < 1 || W < 1, NBP_E_ARG);
    NBP_RETURN_IF(contrast_factor != 1.f, NBP_E_SHAPE);        // a contrast change needs the luminance mean: nbp_raster_rgbz_f32
    dim3 g3((unsigned)nbp_cdiv((long long)H * W, 256), (unsigned)n_frames);
    raster_shade_kernel<<<g3, 256, 0, (hipStream_t)stream>>>((const unsigned long long*)zface, verts, faces, vcolors3,
                                                             camset_from_host(cams12_host, n_frames), H, W, tan_half_fov,
                                                             ambient, nullptr, rgb, nullptr);
    return nbp_launch_status();
}

extern "C" int nbp_segments_hit_mesh_f32(const float* verts, const int* faces, int n_faces, const float* segs6, int n_segs,
                                         int* hit, void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(!verts || !faces || !segs6 || !hit || n_faces < 1 || n_segs < 1 || n_segs > 65535, NBP_E_ARG);
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(hit, 0, (size_t)n_segs * sizeof(int), st);
    if (e != hipSuccess) return (int)e;
    dim3 g((unsigned)nbp_cdiv(n_faces, 256), (unsigned)n_segs);
    segments_hit_kernel<<<g, 256, 0, st>>>(verts, faces, n_faces, segs6, hit);
    return nbp_launch_status();
}

extern "C" int nbp_axis_ray_counts_f32(const float* verts, const int* faces, int n_faces, const float* pts3, int n_pts,
                                       int* counts3, void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(!verts || !faces || !pts3 || !counts3 || n_faces < 1 || n_pts < 1 || n_pts > 65535, NBP_E_ARG);
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(counts3, 0, (size_t)n_pts * 3 * sizeof(int), st);
    if (e != hipSuccess) return (int)e;
    dim3 g((unsigned)nbp_cdiv(n_faces, 256), (unsigned)n_pts);
    axis_ray_count_kernel<<<g, 256, 0, st>>>(verts, faces, n_faces, pts3, counts3);
    return nbp_launch_status();
}

extern "C" int nbp_carve_update_f32(const float* proxy_pts3, int P, const float* depth, const unsigned char* mask_or_null,
                                    const float* cam12_host, int H, int W, float tan_half_fov, float zfar, float fov_range,
                                    float tol, float score_threshold, float* n_inside, float* n_behind, float* occ,
                                    float* out_of_field, void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(!proxy_pts3 || !depth || !cam12_host || !n_inside || !n_behind || !occ || !out_of_field, NBP_E_ARG);
    NBP_RETURN_IF(P < 1 || H < 2 || W < 2, NBP_E_ARG);
    Cam cam;
    for (int k = 0; k < 9; ++k) cam.R[k] = cam12_host[k];
    for (int k = 0; k < 3; ++k) cam.T[k] = cam12_host[9 + k];
    carve_update_kernel<<<(unsigned)nbp_cdiv(P, 256), 256, 0, (hipStream_t)stream>>>(
        proxy_pts3, P, depth, mask_or_null, cam, H, W, tan_half_fov, zfar, fov_range, tol, score_threshold, n_inside,
        n_behind, occ, out_of_field, nullptr, 0.f, ViewArgs{}, nullptr, nullptr);
    return nbp_launch_status();
}

static int view_args_from_host(const float* x_view_host, int n_view, int n_elev, int n_azim, ViewArgs* va) {
    NBP_RETURN_IF(!x_view_host || n_view < 1 || n_view > MAX_CAMS || n_elev < 1 || n_azim < 1 || (long long)n_elev * n_azim > 4096, NBP_E_ARG);
    for (int k = 0; k < 3 * MAX_CAMS; ++k) va->x[k] = k < 3 * n_view ? x_view_host[k] : 0.f;
    va->n_view = n_view; va->n_elev = n_elev; va->n_azim = n_azim;
    const double es = 3.14159265358979323846 / (n_elev + 1), as = 2.0 * 3.14159265358979323846 / n_azim;   // np.pi / (n_elev + 1), 2 np.pi / n_azim
    va->elev_step = (float)es; va->azim_step = (float)as; va->half_e = (float)(es / 2.0); va->half_a = (float)(as / 2.0);
    return 0;
}

extern "C" int nbp_view_state_update_f32(const float* pts3, int P, const unsigned char* mask_or_null, const float* sd_or_null,
                                         float distance_to_surface, const float* x_view_host, int n_view, int n_elev, int n_azim,
                                         float* view_states, void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(!pts3 || !view_states || P < 1, NBP_E_ARG);
    ViewArgs va;
    const int rc = view_args_from_host(x_view_host, n_view, n_elev, n_azim, &va);
    if (rc) return rc;
    view_state_kernel<<<(unsigned)nbp_cdiv(P, 256), 256, 0, (hipStream_t)stream>>>(pts3, P, mask_or_null, sd_or_null, distance_to_surface,
                                                                                 va, view_states);
    return nbp_launch_status();
}

extern "C" int nbp_view_gain_i32(const float* pts3, int P, const float* occ, const float* view_states, const float* cams12_host,
                                 const float* x_cams_host, int n_cams, int n_elev, int n_azim, int H, int W, float tan_half_fov,
                                 float fov_range, int* gains, void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(!pts3 || !occ || !view_states || !cams12_host || !x_cams_host || !gains || P < 1 || n_cams < 1 || H < 2 || W < 2, NBP_E_ARG);
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(gains, 0, (size_t)n_cams * sizeof(int), st);
    if (e != hipSuccess) return (int)e;
    for (int c0 = 0; c0 < n_cams; c0 += MAX_CAMS) {                      // candidates travel in the kernel arguments, 8 at a time
        const int nc = n_cams - c0 < MAX_CAMS ? n_cams - c0 : MAX_CAMS;
        ViewArgs va;
        const int rc = view_args_from_host(x_cams_host + 3 * (size_t)c0, nc, n_elev, n_azim, &va);
        if (rc) return rc;
        dim3 g((unsigned)nbp_cdiv(P, 256), (unsigned)nc);
        view_gain_kernel<<<g, 256, 0, st>>>(pts3, P, occ, view_states, camset_from_host(cams12_host + 12 * (size_t)c0, nc), va, H, W,
                                            tan_half_fov, fov_range, gains + c0);
        const int rc2 = nbp_launch_status();
        if (rc2) return rc2;
    }
    return 0;
}

extern "C" int nbp_carve_view_update_f32(const float* proxy_pts3, int P, const float* depth, const unsigned char* mask_or_null,
                                         const float* cam12_host, int H, int W, float tan_half_fov, float zfar, float fov_range,
                                         float tol, float score_threshold, float* n_inside, float* n_behind, float* occ,
                                         float* out_of_field, const float* x_cam_host, int n_elev, int n_azim,
                                         float distance_to_surface, float* view_states, unsigned char* fov_mask_or_null,
                                         float* sd_or_null, void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(!proxy_pts3 || !depth || !cam12_host || !n_inside || !n_behind || !occ || !out_of_field || !view_states, NBP_E_ARG);
    NBP_RETURN_IF(P < 1 || H < 2 || W < 2, NBP_E_ARG);
    Cam cam;
    for (int k = 0; k < 9; ++k) cam.R[k] = cam12_host[k];
    for (int k = 0; k < 3; ++k) cam.T[k] = cam12_host[9 + k];
    ViewArgs va;
    const int rc = view_args_from_host(x_cam_host, 1, n_elev, n_azim, &va);
    if (rc) return rc;
    carve_update_kernel<<<(unsigned)nbp_cdiv(P, 256), 256, 0, (hipStream_t)stream>>>(
        proxy_pts3, P, depth, mask_or_null, cam, H, W, tan_half_fov, zfar, fov_range, tol, score_threshold, n_inside,
        n_behind, occ, out_of_field, view_states, distance_to_surface, va, fov_mask_or_null, sd_or_null);
    return nbp_launch_status();
}

extern "C" int nbp_points_in_fov_u8(const float* pts3, int P, const float* cams12_host, int n_cams, int H, int W,
                                    float tan_half_fov, float fov_range, unsigned char* mask_or_null, int* any_or_null,
                                    void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(!pts3 || !cams12_host || (!mask_or_null && !any_or_null) || P < 1 || n_cams < 1 || H < 2 || W < 2, NBP_E_ARG);
    hipStream_t st = (hipStream_t)stream;
    for (int c0 = 0; c0 < n_cams; c0 += MAX_CAMS) {                      // cameras travel in the kernel arguments, 8 at a time
        const int nc = n_cams - c0 < MAX_CAMS ? n_cams - c0 : MAX_CAMS;
        if (any_or_null) {
            hipError_t e = hipMemsetAsync(any_or_null + c0, 0, (size_t)nc * sizeof(int), st);
            if (e != hipSuccess) return (int)e;
        }
        dim3 g((unsigned)nbp_cdiv(P, 256), (unsigned)nc);
        points_in_fov_kernel<<<g, 256, 0, st>>>(pts3, P, camset_from_host(cams12_host + 12 * (size_t)c0, nc), H, W, tan_half_fov,
                                                fov_range, mask_or_null ? mask_or_null + (size_t)c0 * P : nullptr,
                                                any_or_null ? any_or_null + c0 : nullptr);
        int rc = nbp_launch_status();
        if (rc) return rc;
    }
    return 0;
}

// dst[offset + i] = pts[i], i < n <= 8: the points ride in the kernel arguments (camera trajectory).
namespace { struct Pts8 { float p[24]; };
__global__ void append_points_kernel(float* __restrict__ dst, long long offset, Pts8 pts, int n) {
    const int i = threadIdx.x;
    if (i < 3 * n) dst[3 * offset + i] = pts.p[i];
} }
extern "C" int nbp_append_points_f32(float* dst, long long offset, const float* pts3_host, int n, void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(!dst || !pts3_host || offset < 0 || n < 1 || n > 8, NBP_E_ARG);
    Pts8 p;
    for (int i = 0; i < 24; ++i) p.p[i] = i < 3 * n ? pts3_host[i] : 0.f;
    append_points_kernel<<<1, 64, 0, (hipStream_t)stream>>>(dst, offset, p, n);
    return nbp_launch_status();
}

// Host mirror of the index bijection (the Python driver and the oracle cross-check use it).
extern "C" unsigned nbp_perm_index_host(unsigned j, unsigned n, unsigned seed) {
    return n == 0 ? 0 : perm_index(j, n, perm_bits(n), seed);
}

// ------------------------------------------------------------------ GT obstacle label (training data path)
// get_binary_obstacle_array (next_best_path/utility/utils.py:226-262) draws the mesh / plane(y = camera height)
// intersection with matplotlib, saves a PNG, resizes and thresholds it.  Here: one wave per face; the face's
// intersection segment with the plane is drawn into the S x S label as the set of pixels whose centre lies within
// half_width pixels of it.  Image axes as in the reference after its left-right flip: column grows with
// -(x - cx), row with -(z - cz), the window is [lo, hi] around the camera (pixel AREAS, i.e. centres at +0.5).
namespace {
// Pixel coordinates of a world point: u = ((cx - x) + hu) su, v = ((cz - z) + hv) sv.  cap: how far a segment's stroke extends
// beyond its end points along its direction (0: the round "within half_width of the segment" stroke; > 0: a rectangle with
// matplotlib's default projecting caps).
__global__ __launch_bounds__(256) void slice_obstacle_kernel(const float* __restrict__ verts, const int* __restrict__ faces,
                                                             int F, float y0, float cx, float cz, int S, float hu, float su_,
                                                             float hv, float sv_, float half_width, float cap,
                                                             float* __restrict__ out) {
    const int f = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (f >= F) return;
    float px[3], py[3], pz[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int v = faces[3 * f + k];
        px[k] = verts[3 * v]; py[k] = verts[3 * v + 1] - y0; pz[k] = verts[3 * v + 2];
    }
    float su[2], sv[2];
    int n = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int q = (k + 1) % 3;
        if ((py[k] < 0.f) != (py[q] < 0.f)) {
            const float t = py[k] / (py[k] - py[q]);
            const float x = px[k] + t * (px[q] - px[k]);
            const float z = pz[k] + t * (pz[q] - pz[k]);
            if (n < 2) { su[n] = ((cx - x) + hu) * su_; sv[n] = ((cz - z) + hv) * sv_; }
            ++n;
        }
    }
    if (n != 2) return;
    const float wu = su[1] - su[0], wv = sv[1] - sv[0];
    const float L2 = wu * wu + wv * wv;
    const float pad = fmaxf(half_width, cap) + 1.0f;
    const int c0 = max(0, (int)floorf(fminf(su[0], su[1]) - pad)), c1 = min(S - 1, (int)ceilf(fmaxf(su[0], su[1]) + pad));
    const int r0 = max(0, (int)floorf(fminf(sv[0], sv[1]) - pad)), r1 = min(S - 1, (int)ceilf(fmaxf(sv[0], sv[1]) + pad));
    if (c1 < c0 || r1 < r0) return;
    const int wbox = c1 - c0 + 1, total = wbox * (r1 - r0 + 1);
    const float h2 = half_width * half_width;
    for (int i = lane; i < total; i += 64) {
        const int r = r0 + i / wbox, c = c0 + i % wbox;
        const float qu = ((float)c + 0.5f) - su[0], qv = ((float)r + 0.5f) - sv[0];
        if (cap > 0.f) {                 // rectangle: |normal distance| <= half_width, along-distance in [-cap, L + cap]
            if (!(L2 > 0.f)) continue;
            const float L = sqrtf(L2);
            const float ta = (qu * wu + qv * wv) / L, tn = fabsf(qv * wu - qu * wv) / L;
            if (ta >= -cap && ta <= L + cap && tn <= half_width) out[r * S + c] = 1.0f;
            continue;
        }
        float t = 0.f;
        if (L2 > 0.f) t = fminf(fmaxf((qu * wu + qv * wv) / L2, 0.f), 1.f);
        const float du = qu - t * wu, dv = qv - t * wv;
        if (du * du + dv * dv <= h2) out[r * S + c] = 1.0f;
    }
}
}  // namespace

extern "C" int nbp_slice_obstacle_f32(const float* verts, const int* faces, int n_faces, float y0, float cx, float cz,
                                      int S, float lo, float hi, float half_width_px, float* out, void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(!verts || !faces || !out || n_faces < 1 || S < 1 || !(hi > lo) || !(half_width_px > 0.f), NBP_E_ARG);
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(out, 0, (size_t)S * S * sizeof(float), st);
    if (e != hipSuccess) return (int)e;
    const float scale = (float)((double)S / ((double)hi - (double)lo));
    slice_obstacle_kernel<<<(unsigned)nbp_cdiv((long long)n_faces * 64, 256), 256, 0, st>>>(verts, faces, n_faces, y0, cx, cz,
                                                                                          S, hi, scale, hi, scale, half_width_px, 0.f, out);
    return nbp_launch_status();
}

extern "C" int nbp_slice_obstacle_fig_f32(const float* verts, const int* faces, int n_faces, float y0, float cx, float cz,
                                          int S, float half_u, float scale_u, float half_v, float scale_v, float half_width_px,
                                          float cap_px, float* out, void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(!verts || !faces || !out || n_faces < 1 || S < 1 || !(scale_u > 0.f) || !(scale_v > 0.f) || !(half_width_px > 0.f) ||
                  !(cap_px >= 0.f), NBP_E_ARG);
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(out, 0, (size_t)S * S * sizeof(float), st);
    if (e != hipSuccess) return (int)e;
    slice_obstacle_kernel<<<(unsigned)nbp_cdiv((long long)n_faces * 64, 256), 256, 0, st>>>(verts, faces, n_faces, y0, cx, cz,
                                                                                          S, half_u, scale_u, half_v, scale_v,
                                                                                          half_width_px, cap_px, out);
    return nbp_launch_status();
}
