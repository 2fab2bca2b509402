// nbp_conv.hip -- fp32 implicit-GEMM convolution on the gfx950 matrix cores + the small
// layers around it (first conv, max-pool, attention gate tail, final 1x1, layout helpers).
//
// Replaces the cuDNN/MIOpen calls behind next_best_path/networks/nbp_model.py:8-62
// (conv_block / up_conv / Attention_block).  GEMM view of a convolution on NHWC data:
//     out[m][n] = sum_k A[m][k] * Wt[k][n],   m = (b,y,x) pixel, n = c_out,
//     k = (c_in chunk of 32, tap, channel in chunk)
// A is never materialised: each K chunk (32 input channels of one filter tap) is gathered
// from the activation tensor straight into LDS, with zero padding, the x2 nearest upsample
// (ref :27) and the channel concat (ref :128) folded into the gather address.
//
// MFMA: v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, 64 cyc/SIMD, 157 TFLOP/s chip peak).
// One wave owns a (TM*32) x (TN*32) accumulator tile; per 32-channel chunk it issues
// (TM+TN)*4 ds_read_b128 and TM*TN*16 MFMAs.  The LDS image is [row][32 floats] with the
// 16-byte slot index XOR-swizzled by (row>>1)&7, which makes both the staging
// ds_write_b128 and the fragment ds_read_b128 bank-conflict free.
// K order inside a chunk is permuted (lanes 0-31 take floats 8j..8j+3, lanes 32-63 take
// 8j+4..8j+7); A and B use the same permutation so the sum over k is complete.
#include "common.h"
#include "nbp_internal.h"
#include "nbp_first_conv.h"
#include <cstdlib>



template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(256, 2) void igemm_conv_kernel(IgemmArgs a) {
    int zs = blockIdx.z;
    if (zs >= a.split_k) {     // second group of a grouped launch (decoder 2 next to decoder 1)
        zs -= a.split_k;
        a.src0 = a.g_src0; a.src1 = a.g_src1; a.wpk = a.g_wpk; a.scale = a.g_scale; a.shift = a.g_shift; a.out = a.g_out;
    }
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int RA = BM / 32, RB = BN / 32;
    constexpr int STAGE = (BM + BN) * 32;  // floats per LDS stage
    static_assert(WM * WN == 4, "256-thread workgroup");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // Workgroups are dealt round-robin to the 8 XCDs (each with its own L2) in linear-id order.  A 3x3 tile
    // shares two of its three input rows with the tiles of the neighbouring image rows and its whole A tile
    // with the other n tiles of the same rows, so give each XCD a contiguous run of tiles, n fastest.
    unsigned mt = blockIdx.x, nt = blockIdx.y;
    if (a.xcd_remap) {
        const unsigned L = blockIdx.x + gridDim.x * blockIdx.y, T = gridDim.x * gridDim.y;
        const unsigned xcd = L & 7u, idx = L >> 3, q = T >> 3, r = T & 7u;
        const unsigned v = xcd * q + min(xcd, r) + idx;
        nt = v % gridDim.y;
        mt = v / gridDim.y;
    }
    const long long m0 = (long long)mt * BM;
    const int n0 = nt * BN;
    const int c_begin = zs * a.chunks_per_split;
    const int c_end = min(c_begin + a.chunks_per_split, a.chunks_total);

    // ---- per-thread staging coordinates: thread t moves 16 B: row t/8 (+32 i), slot t%8
    const int lrow = tid >> 3, slot = tid & 7;
    int py[RA], px[RA], pb[RA];
    const int HW = a.H * a.W;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        long long m = m0 + lrow + 32 * i;
        if (m < a.M) {
            int b = (int)(m / HW);
            int rem = (int)(m - (long long)b * HW);
            py[i] = rem / a.W;
            px[i] = rem - py[i] * a.W;
            pb[i] = b * a.Hs * a.Ws;
        } else {
            py[i] = -1000000; px[i] = 0; pb[i] = 0;   // every tap out of bounds -> zeros
        }
    }
    int wsw[RA > RB ? RA : RB];   // swizzled LDS float offset of this thread's 16 B in row i
#pragma unroll
    for (int i = 0; i < (RA > RB ? RA : RB); ++i) {
        int r = lrow + 32 * i;
        wsw[i] = r * 32 + ((slot ^ ((r >> 1) & 7)) << 2);
    }

    // Activation gather through buffer descriptors: out-of-image taps (zero padding) and rows
    // past M get voffset = OOB, which the hardware range check turns into zeros -- the eight
    // loads of a chunk issue back to back with no exec-mask branches.
    const __amdgpu_buffer_rsrc_t rs0 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src0), 0, a.bytes0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src1 ? a.src1 : a.src0), 0, a.bytes1, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    f32x4 ga[RA], gb[RB];
    auto load_global = [&](int c) {
        const int cc = c / a.taps;
        const int tap = c - cc * a.taps;
        int dy = 0, dx = 0;
        if (a.taps == 9) { dy = tap / 3 - 1; dx = tap - (tap / 3) * 3 - 1; }
        const bool first = cc < a.cc0;
        const int Cs = first ? a.C0 : a.C1;
        const int coff = (first ? cc : cc - a.cc0) * 32 + slot * 4;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int yy = py[i] + dy, xx = px[i] + dx;
            const bool ok = (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
            const int pix = pb[i] + (yy >> a.ups) * a.Ws + (xx >> a.ups);
            const unsigned off = ok ? (unsigned)(pix * Cs + coff) * 4u : OOB;
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            u32x4 r = first ? __builtin_amdgcn_raw_buffer_load_b128(rs0, off, 0, 0)
                            : __builtin_amdgcn_raw_buffer_load_b128(rs1, off, 0, 0);
            ga[i] = __builtin_bit_cast(f32x4, r);
        }
        const float* wb = a.wpk + ((long long)c * a.N + n0 + lrow) * 32 + slot * 4;
#pragma unroll
        for (int i = 0; i < RB; ++i) gb[i] = *reinterpret_cast<const f32x4*>(wb + (long long)i * 32 * 32);
    };
    auto store_lds = [&](int buf) {
        float* A = lds + buf * STAGE;
        float* Bt = A + BM * 32;
#pragma unroll
        for (int i = 0; i < RA; ++i) *reinterpret_cast<f32x4*>(A + wsw[i]) = ga[i];
#pragma unroll
        for (int i = 0; i < RB; ++i) *reinterpret_cast<f32x4*>(Bt + wsw[i]) = gb[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment read offsets (floats), without the k-subgroup slot
    int fa_row[TM], fb_row[TN], fa_sw[TM], fb_sw[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        int r = (wm * TM + i) * 32 + (lane & 31);
        fa_row[i] = r * 32; fa_sw[i] = (r >> 1) & 7;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        int r = (wn * TN + j) * 32 + (lane & 31);
        fb_row[j] = r * 32; fb_sw[j] = (r >> 1) & 7;
    }
    const int khalf = lane >> 5;

    if (c_begin < c_end) {
        load_global(c_begin);
        store_lds(0);
    }
    __syncthreads();
    int cur = 0;
    for (int c = c_begin; c < c_end; ++c) {
        const bool more = (c + 1 < c_end);
        if (more) load_global(c + 1);
        const float* A = lds + cur * STAGE;
        const float* Bt = A + BM * 32;
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4) {
            const int s = 2 * j4 + khalf;
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i] = *reinterpret_cast<const f32x4*>(A + fa_row[i] + ((s ^ fa_sw[i]) << 2));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bf[j] = *reinterpret_cast<const f32x4*>(Bt + fb_row[j] + ((s ^ fb_sw[j]) << 2));
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][t], bf[j][t], acc[i][j], 0, 0, 0);
        }
        if (more) store_lds(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue.  C/D map of the 32x32 MFMA: col = lane&31, row = (r&3)+8*(r>>2)+4*(lane>>5)
    const bool final_out = (a.split_k == 1);
    float* outp = final_out ? a.out : a.partial + (long long)blockIdx.z * a.M * a.N;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + (wn * TN + j) * 32 + (lane & 31);
        float sc = 1.f, sh = 0.f;
        if (final_out) { sc = a.scale[n]; sh = a.shift[n]; }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                long long m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < a.M) {
                    float v = acc[i][j][r];
                    if (final_out) {
                        v = v * sc + sh;
                        if (a.relu) v = fmaxf(v, 0.f);
                    }
                    outp[m * a.N + n] = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------ 3x3 convolution from an LDS-resident halo tile
// Same idea as conv3x3_halo_bf16_kernel (nbp_bf16.hip), for the exact-fp32 path: a workgroup owns an 8 x 32 pixel
// tile of one image and BN output channels; per 32-channel chunk the 10 x 34 pixel halo tile is DMA'd into LDS once
// (buffer_load ... lds, out-of-image pixels = out-of-range offsets = zeros) and all nine taps run from it, with the
// next tap's weights streaming into a double buffer behind the current tap's 128 MFMAs per wave.  Compared with the
// implicit GEMM the main loop has no gather arithmetic, no staging registers and no ds_write pass, and a barrier
// every 8192 MFMA cycles instead of every 4096.  K order: (chunk, tap, channel) as in the implicit GEMM.
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// TM = image rows per wave: 2 -> 8 x 32 pixel tiles; 1 -> 4 x 32 pixel tiles (twice the workgroups for small images / B = 1,
// at half the MFMAs per barrier).
template <int TN, int TM>
__global__ __launch_bounds__(256, 2) void conv3x3_halo_f32_kernel(IgemmArgs a) {
    int zs = blockIdx.z;        // split-K slice (of 32-channel chunks), then the group
    if (zs >= a.split_k) {
        zs -= a.split_k;
        a.src0 = a.g_src0; a.src1 = a.g_src1; a.wpk = a.g_wpk; a.scale = a.g_scale; a.shift = a.g_shift; a.out = a.g_out;
    }
    constexpr int BN = TN * 32;
    constexpr int HW_ = 34;
    constexpr int TROWS = 4 * TM;                              // image rows per tile
    constexpr int HPIX = (TROWS + 2) * HW_;                    // halo pixels of 32 floats: 340 (TM = 2) / 204 (TM = 1)
    constexpr int SLOTS = (HPIX + 7) / 8;                      // DMA instructions of 8 halo pixels: 43 / 26
    constexpr int NI = (SLOTS + 3) / 4;                        // per wave
    constexpr int HALO_BYTES = SLOTS * 1024;
    constexpr int WB = BN * 128;
    extern __shared__ __attribute__((aligned(16))) char ldsb[];
    char* const halo = ldsb;
    char* const wbuf = ldsb + HALO_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = a.W >> 5, tiles_y = a.H / TROWS;
    // XCD-aware order: the workgroups one XCD receives (every 8th linear id) take a contiguous run of (pixel tile,
    // channel block) pairs, channel block fastest: the channel blocks of a tile share its halo in that XCD's L2 and
    // neighbouring tiles share their border rows / columns
    unsigned tile = blockIdx.x, nt = blockIdx.y;
    if (a.xcd_remap) {
        const unsigned L = blockIdx.x + gridDim.x * blockIdx.y, T = gridDim.x * gridDim.y;
        const unsigned xcd = L & 7u, idx = L >> 3, q = T >> 3, r = T & 7u;
        const unsigned v = xcd * q + min(xcd, r) + idx;
        nt = v % gridDim.y;
        tile = v / gridDim.y;
    }
    const int tx = tile % tiles_x; tile /= tiles_x;
    const int ty = tile % tiles_y;
    const int b = tile / tiles_y;
    const int y0 = ty * TROWS, x0 = tx * 32;
    const int n0 = nt * BN;

    int hpix[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int q = 4 * i + wave;
        const int hr = 8 * q + (lane >> 3);
        const int hy = hr / HW_, hx = hr - hy * HW_;
        const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
        const bool ok = q < SLOTS && hr < HPIX && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
        hpix[i] = ok ? (b * a.Hs + (yy >> a.ups)) * a.Ws + (xx >> a.ups) : -1;
    }
    const int hslot = lane & 7;
    const int lrow = tid >> 3;
    const int wslot = (tid & 7) ^ ((lrow >> 1) & 7);
    const __amdgpu_buffer_rsrc_t rs0 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src0), 0, a.bytes0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src1 ? a.src1 : a.src0), 0, a.bytes1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wpk), 0, a.bytesw, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;

    auto issue_halo = [&](int cc) {
        const bool first = cc < a.cc0;
        const int Cs = first ? a.C0 : a.C1;
        const int cbase = (first ? cc : cc - a.cc0) * 32;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int q = 4 * i + wave;
            if (q < SLOTS) {
                const int sw = (4 * q + (lane >> 4)) & 7;
                const unsigned off = hpix[i] >= 0 ? (unsigned)(hpix[i] * Cs + cbase + ((hslot ^ sw) << 2)) * 4u : OOB;
                if (first) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (lds_ptr_t)(halo + q * 1024), 16, off, 0, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lds_ptr_t)(halo + q * 1024), 16, off, 0, 0, 0);
            }
        }
    };
    auto issue_w = [&](int t) {      // t = chunk * 9 + tap; packed weights are [chunk][tap][N][32]
        char* dst = wbuf + (t & 1) * WB + wave * 1024;
        const unsigned woff = (unsigned)(((long long)t * a.N + n0 + lrow) * 32 + wslot * 4) * 4u;
#pragma unroll
        for (int j = 0; j < TN; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr_t)(dst + j * 4096), 16, woff + j * 4096, 0, 0, 0);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int fb_row[TN], fb_sw[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int r = j * 32 + (lane & 31);
        fb_row[j] = r * 128; fb_sw[j] = (r >> 1) & 7;
    }
    const int khalf = lane >> 5;
    const int hbase = (TM * wave) * HW_ + (lane & 31);

    // chunks_per_split / chunks_total count (chunk, tap) pairs as in the implicit GEMM; a slice is whole chunks
    const int cc_begin = zs * (a.chunks_per_split / 9);
    const int chunks = min(cc_begin + a.chunks_per_split / 9, a.chunks_total / 9);
    const int t_total = chunks * 9;
    if (cc_begin < chunks) {
        issue_halo(cc_begin);
        issue_w(cc_begin * 9);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int cc = cc_begin; cc < chunks; ++cc) {
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            const int t = cc * 9 + tap;
            if (t + 1 < t_total) issue_w(t + 1);
            const char* Bt = wbuf + (t & 1) * WB;
            int ar[TM], as[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int hr = hbase + (tap / 3 + i) * HW_ + (tap % 3);
                ar[i] = hr * 128; as[i] = (hr >> 1) & 7;
            }
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4) {
                const int s = 2 * j4 + khalf;
                f32x4 xf[TM], wf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) xf[i] = *reinterpret_cast<const f32x4*>(halo + ar[i] + ((s ^ as[i]) << 4));
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    wf[j] = *reinterpret_cast<const f32x4*>(Bt + fb_row[j] + ((s ^ fb_sw[j]) << 4));
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int i = 0; i < TM; ++i)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xf[i][e], wf[j][e], acc[i][j], 0, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        if (cc + 1 < chunks) {
            issue_halo(cc + 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }

    // ---- epilogue (A = pixels, B = weights): col n = lane & 31, pixel x = (r&3) + 8 (r>>2) + 4 (lane>>5)
    const bool final_out = (a.split_k == 1);
    float* outp = final_out ? a.out : a.partial + (long long)blockIdx.z * a.M * a.N;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + j * 32 + (lane & 31);
        float sc = 1.f, sh = 0.f;
        if (final_out) { sc = a.scale[n]; sh = a.shift[n]; }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const long long mrow = ((long long)b * a.H + y0 + TM * wave + i) * a.W + x0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int px = (r & 3) + 8 * (r >> 2) + 4 * khalf;
                float v = acc[i][j][r];
                if (final_out) {
                    v = v * sc + sh;
                    if (a.relu) v = fmaxf(v, 0.f);
                }
                outp[(mrow + px) * a.N + n] = v;
            }
        }
    }
}

// out[m][n] = act(sum_s partial[s][m][n] * scale[n] + shift[n]);  N % 4 == 0; blockIdx.y = group
struct ReduceGroup { const float* scale; const float* shift; float* out; };
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial_all, int split_k,
                                                            long long MN, int N, ReduceGroup g0, ReduceGroup g1,
                                                            int relu) {
    const ReduceGroup g = blockIdx.y ? g1 : g0;
    const float* __restrict__ partial = partial_all + (long long)blockIdx.y * split_k * MN;
    const long long n4 = MN >> 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (long long)gridDim.x * blockDim.x) {
        f32x4 s = *reinterpret_cast<const f32x4*>(partial + i * 4);
        for (int k = 1; k < split_k; ++k) {
            f32x4 p = *reinterpret_cast<const f32x4*>(partial + (long long)k * MN + i * 4);
            s += p;
        }
        const int n = (int)((i * 4) % N);
        f32x4 sc = *reinterpret_cast<const f32x4*>(g.scale + n);
        f32x4 sh = *reinterpret_cast<const f32x4*>(g.shift + n);
        f32x4 v = s * sc + sh;
        if (relu) {
            v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f);
            v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
        }
        *reinterpret_cast<f32x4*>(g.out + i * 4) = v;
    }
}

// ------------------------------------------------------------------ tile table / planning
static TileInfo tile_info(int tile) {
    switch (tile) {
        case NBP_TILE_HALO_128: return {256, 128};
        case NBP_TILE_HALO_64: return {256, 64};
        case NBP_TILE_HALO4_128: return {128, 128};
        case NBP_TILE_HALO4_64: return {128, 64};
        case NBP_TILE_128x128: return {128, 128};
        case NBP_TILE_256x64: return {256, 64};
        case NBP_TILE_256x32: return {256, 32};
        case NBP_TILE_128x64: return {128, 64};
        case NBP_TILE_64x128: return {64, 128};
        default: return {0, 0};
    }
}

// Shared by the forward, the single-layer entry point and the workspace query.
static bool halo_ok_f32(int H, int W, int N, int ksize, int bn, int rows = 8) {
    return ksize == 3 && H >= rows && W >= 32 && H % rows == 0 && (W & 31) == 0 && N % bn == 0;
}
static bool is_halo_tile(int t) {
    return t == NBP_TILE_HALO_128 || t == NBP_TILE_HALO_64 || t == NBP_TILE_HALO4_128 || t == NBP_TILE_HALO4_64;
}

ConvPlan nbp_plan_conv(long long M, int N, int chunks_total, int tile, int split_k, int groups, int H, int W, int ksize) {
    if (tile == NBP_TILE_AUTO && ksize == 3 && split_k <= 0) {
        // halo-tile kernel (no split-K) once tiles alone fill the chip
        constexpr int min_blocks = 256;
        const int bn = N % 128 == 0 ? 128 : 64;
        // split-K over whole 32-channel chunks may fill the chip when the tiles alone do not (each slice keeps
        // >= 2 chunks = 18 (chunk, tap) steps)
        const int cc = chunks_total / 9;
        // 8-row tiles; 4-row tiles (twice the workgroups, half the MFMAs per barrier: tile ids 8 / 9) were measured
        // neutral at B = 1..8 when preferred over a split-K > 2, so the automatic choice uses them only on request
        constexpr int allow4 = 0;
        int sk_of[2] = {0, 0};      // split-K that fills the chip with 8-row / 4-row tiles (0 = not possible)
        for (int v = 0; v < 2; ++v) {
            const int rows = v == 0 ? 8 : 4;
            if ((v == 1 && !allow4) || !halo_ok_f32(H, W, N, ksize, bn, rows)) continue;
            const long long blocks = (M / (rows * 32)) * (N / bn) * groups;
            int sk = 1;
            while (blocks * sk < min_blocks && cc / (sk * 2) >= 2 && sk < 16) sk *= 2;
            if (blocks * sk >= min_blocks) sk_of[v] = sk;
        }
        const int pick = (sk_of[0] && (sk_of[0] <= 2 || !sk_of[1])) ? 0 : (sk_of[1] ? 1 : -1);
        if (pick >= 0) {
            tile = pick == 0 ? (bn == 128 ? NBP_TILE_HALO_128 : NBP_TILE_HALO_64)
                             : (bn == 128 ? NBP_TILE_HALO4_128 : NBP_TILE_HALO4_64);
            split_k = sk_of[pick];
        }
    }
    if (is_halo_tile(tile)) {
        ConvPlan h;
        const int cc = chunks_total / 9;
        int sk = split_k <= 0 ? 1 : split_k;
        if (sk > cc) sk = cc;
        if (sk < 1) sk = 1;              // 1x1 convolution asked for a halo tile: rejected by the caller's shape check
        const int per = cc > 0 ? (int)nbp_cdiv(cc, sk) : 1;
        h.tile = tile; h.split_k = cc > 0 ? (int)nbp_cdiv(cc, per) : 1; h.chunks_per_split = per * 9;
        return h;
    }
    // Policy from tools/bench_conv.py --sweep on MI355X (B = 1, 2, 8; SURVEY.md A.1 shapes): the big
    // 128x128 / 256x64 tiles only pay once they alone give >= 512 workgroups (2 per CU); with fewer
    // tiles the half-size tiles (64x128, 128x64) reach the same workgroup count with half the split-K,
    // i.e. half the partial-sum traffic, and win by 5-30 % at B = 1.
    ConvPlan p;
    if (tile == NBP_TILE_AUTO) {
        if (N % 128 == 0)
            tile = nbp_cdiv(M, 128) * (N / 128) * groups >= 512 ? NBP_TILE_128x128 : NBP_TILE_64x128;
        else if (N % 64 == 0)
            tile = nbp_cdiv(M, 256) * (N / 64) * groups >= 512 ? NBP_TILE_256x64 : NBP_TILE_128x64;
        else
            tile = NBP_TILE_256x32;
    }
    p.tile = tile;
    TileInfo ti = tile_info(tile);
    if (ti.bm == 0) {                 // unknown tile id: the caller's shape check refuses it
        p.split_k = 1; p.chunks_per_split = chunks_total;
        return p;
    }
    if (split_k <= 0) {
        const long long blocks = nbp_cdiv(M, ti.bm) * (N / ti.bn) * groups;
        split_k = 1;
        // aim for >= 2 workgroups per CU (512) while each split keeps >= 4 chunks of K ...
        while (blocks * split_k < 512 && chunks_total / (split_k * 2) >= 4 && split_k < 64) split_k *= 2;
        // ... but a 2-way split of a short K (< 12 chunks each) costs more than it buys
        if (split_k == 2 && chunks_total / 2 < 12) split_k = 1;
    }
    if (split_k > chunks_total) split_k = chunks_total;
    p.chunks_per_split = (int)nbp_cdiv(chunks_total, split_k);
    p.split_k = (int)nbp_cdiv(chunks_total, p.chunks_per_split);
    return p;
}

template <int TN, int TM>
static int launch_halo_f32(const IgemmArgs& a, hipStream_t st, int tile) {
    { char nm[96]; snprintf(nm, sizeof(nm), "conv3x3_halo_f32_kernel<%d, %d>", TN, TM); nbp_note_kernel_symbol(tile, nm); }
    constexpr size_t smem = (size_t)(((4 * TM + 2) * 34 + 7) / 8) * 1024 + 2 * (size_t)TN * 32 * 128;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_f32_kernel<TN, TM>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    dim3 grid((unsigned)(a.M / (128 * TM)), (unsigned)(a.N / (TN * 32)), (unsigned)(a.split_k * a.groups));
    conv3x3_halo_f32_kernel<TN, TM><<<grid, 256, smem, st>>>(a);
    return nbp_launch_status();
}

template <int WM, int WN, int TM, int TN>
static int launch_igemm(const IgemmArgs& a, hipStream_t st, int tile) {
    { char nm[96]; snprintf(nm, sizeof(nm), "igemm_conv_kernel<%d, %d, %d, %d>", WM, WN, TM, TN); nbp_note_kernel_symbol(tile, nm); }
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr size_t smem = 2 * (size_t)(BM + BN) * 32 * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_conv_kernel<WM, WN, TM, TN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    dim3 grid((unsigned)nbp_cdiv(a.M, BM), (unsigned)(a.N / BN), (unsigned)(a.split_k * a.groups));
    igemm_conv_kernel<WM, WN, TM, TN><<<grid, 256, smem, st>>>(a);
    return nbp_launch_status();
}

// Internal entry used by nbp_forward.hip too.  groups == 2 runs two same-shaped convolutions
// (operand set `o` and `o2`) in one launch.

int nbp_conv_igemm_launch_g(const ConvOperands& o, const ConvOperands* o2, int C0, int C1, int ups, int B, int H, int W,
                            int ksize, int N, int relu, int split_k, int tile, void* ws, size_t ws_bytes, hipStream_t st) {
    const int groups = o2 ? 2 : 1;
    NBP_RETURN_IF(!o.src0 || !o.wpk || !o.scale || !o.shift || !o.out, NBP_E_ARG);
    NBP_RETURN_IF(o2 && (!o2->src0 || !o2->wpk || !o2->scale || !o2->shift || !o2->out), NBP_E_ARG);
    NBP_RETURN_IF(B < 1 || H < 1 || W < 1, NBP_E_ARG);
    NBP_RETURN_IF(ksize != 1 && ksize != 3, NBP_E_ARG);
    NBP_RETURN_IF(C0 < 32 || C0 % 32 || C1 < 0 || C1 % 32 || N < 32 || N % 32, NBP_E_SHAPE);
    NBP_RETURN_IF(C1 > 0 && (!o.src1 || (o2 && !o2->src1)), NBP_E_ARG);
    NBP_RETURN_IF(ups && ((H | W) & 1), NBP_E_SHAPE);
    IgemmArgs a;
    a.src0 = o.src0; a.src1 = o.src1; a.C0 = C0; a.C1 = C1; a.cc0 = C0 / 32; a.ups = ups ? 1 : 0;
    a.H = H; a.W = W; a.Hs = ups ? H / 2 : H; a.Ws = ups ? W / 2 : W;
    a.taps = ksize * ksize; a.wpk = o.wpk; a.N = N; a.scale = o.scale; a.shift = o.shift; a.relu = relu;
    a.out = o.out;
    a.groups = groups;
    a.g_src0 = o2 ? o2->src0 : nullptr; a.g_src1 = o2 ? o2->src1 : nullptr; a.g_wpk = o2 ? o2->wpk : nullptr;
    a.g_scale = o2 ? o2->scale : nullptr; a.g_shift = o2 ? o2->shift : nullptr; a.g_out = o2 ? o2->out : nullptr;
    a.M = (long long)B * H * W;
    {
        const long long b0 = (long long)B * a.Hs * a.Ws * C0 * 4, b1 = (long long)B * a.Hs * a.Ws * C1 * 4;
        NBP_RETURN_IF(b0 >= (1ll << 31) || b1 >= (1ll << 31), NBP_E_SHAPE);   // 32-bit buffer offsets
        a.bytes0 = (unsigned)b0; a.bytes1 = C1 ? (unsigned)b1 : (unsigned)b0;
        const long long bw = (long long)(C0 + C1) * a.taps * N * 4;
        a.bytesw = bw < (1ll << 31) ? (unsigned)bw : 0u;
    }
    a.chunks_total = (C0 + C1) / 32 * a.taps;
    ConvPlan p = nbp_plan_conv(a.M, N, a.chunks_total, tile, split_k, groups, H, W, ksize);
    TileInfo ti = tile_info(p.tile);
    NBP_RETURN_IF(ti.bm == 0 || N % ti.bn, NBP_E_SHAPE);
    if (is_halo_tile(p.tile))
        NBP_RETURN_IF(!halo_ok_f32(H, W, N, ksize, ti.bn, ti.bm / 32) || a.bytesw == 0, NBP_E_SHAPE);
    a.split_k = p.split_k; a.chunks_per_split = p.chunks_per_split;
    {   // XCD-contiguous tile runs cut the L2-miss traffic of the 3x3 halo rows; measured on MI355X they are
        // neutral-to-better (+2 %) once the grid is several waves deep and cost up to 10 % on single-wave grids,
        // so only multi-wave grids get them.
        const long long tiles = nbp_cdiv(a.M, ti.bm) * (N / ti.bn);
        const bool halo = is_halo_tile(p.tile);
        a.xcd_remap = (halo ? tiles >= 512 : tiles >= 2048) ? 1 : 0;
    }
    a.partial = nullptr;
    if (p.split_k > 1) {
        NBP_RETURN_IF(!ws || ws_bytes < (size_t)groups * p.split_k * a.M * N * sizeof(float), NBP_E_WS);
        a.partial = (float*)ws;
    }
    int rc;
    switch (p.tile) {
        case NBP_TILE_128x128: rc = launch_igemm<2, 2, 2, 2>(a, st, p.tile); break;
        case NBP_TILE_256x64: rc = launch_igemm<4, 1, 2, 2>(a, st, p.tile); break;
        case NBP_TILE_256x32: rc = launch_igemm<4, 1, 2, 1>(a, st, p.tile); break;
        case NBP_TILE_128x64: rc = launch_igemm<2, 2, 2, 1>(a, st, p.tile); break;
        case NBP_TILE_64x128: rc = launch_igemm<1, 4, 2, 1>(a, st, p.tile); break;
        case NBP_TILE_HALO_128: rc = launch_halo_f32<4, 2>(a, st, p.tile); break;
        case NBP_TILE_HALO_64: rc = launch_halo_f32<2, 2>(a, st, p.tile); break;
        case NBP_TILE_HALO4_128: rc = launch_halo_f32<4, 1>(a, st, p.tile); break;
        case NBP_TILE_HALO4_64: rc = launch_halo_f32<2, 1>(a, st, p.tile); break;
        default: return NBP_E_ARG;
    }
    if (rc) return rc;
    if (p.split_k > 1) {
        long long MN = a.M * N;
        ReduceGroup g0{o.scale, o.shift, o.out}, g1{a.g_scale, a.g_shift, a.g_out};
        dim3 grid((unsigned)nbp_ew_grid(MN / 4, 256), (unsigned)groups);
        splitk_reduce_kernel<<<grid, 256, 0, st>>>((const float*)ws, p.split_k, MN, N, g0, g1, relu);
        rc = nbp_launch_status();
    }
    return rc;
}

int nbp_conv_igemm_launch(const float* src0, int C0, const float* src1, int C1, int ups, int B, int H, int W,
                          int ksize, const float* wpk, int N, const float* scale, const float* shift, int relu,
                          float* out, int split_k, int tile, void* ws, size_t ws_bytes, hipStream_t st) {
    ConvOperands o{src0, src1, wpk, scale, shift, out};
    return nbp_conv_igemm_launch_g(o, nullptr, C0, C1, ups, B, H, W, ksize, N, relu, split_k, tile, ws, ws_bytes, st);
}

extern "C" int nbp_conv_igemm_f32(const float* src0, int C0, const float* src1, int C1, int ups, int B, int H,
                                  int W, int ksize, const float* w_packed, int N, const float* scale,
                                  const float* shift, int relu, float* out, int split_k, int tile, void* ws,
                                  size_t ws_bytes, void* stream) {
    NBP_ENTER();
    return nbp_conv_igemm_launch(src0, C0, src1, C1, ups, B, H, W, ksize, w_packed, N, scale, shift, relu, out,
                                 split_k, tile, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" size_t nbp_conv_igemm_workspace_bytes(int B, int H, int W, int N, int split_k) {
    if (split_k <= 1 && split_k != 0) return 0;
    int sk = split_k <= 0 ? 64 : split_k;
    return (size_t)sk * B * H * W * N * sizeof(float);
}

// ------------------------------------------------------------------ weight packing
__global__ void pack_conv_weight_kernel(const float* __restrict__ w, int N, int C, int taps,
                                        const float* __restrict__ scale, int c_off, float* __restrict__ dst) {
    const long long total = (long long)N * C * taps;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        int tap = (int)(i % taps);
        long long t = i / taps;
        int c = (int)(t % C);
        int n = (int)(t / C);
        float v = w[i];
        if (scale) v *= scale[n];
        int cg = c_off + c;
        dst[(((long long)(cg >> 5) * taps + tap) * N + n) * 32 + (cg & 31)] = v;
    }
}

extern "C" int nbp_pack_conv_weight(const float* w_oihw, int N, int C, int ksize, const float* scale_or_null,
                                    int c_off, int c_total, float* dst, void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(!w_oihw || !dst, NBP_E_ARG);
    NBP_RETURN_IF(ksize != 1 && ksize != 3, NBP_E_ARG);
    NBP_RETURN_IF(N < 1 || C < 1 || c_off < 0 || c_off + C > c_total || c_total % 32, NBP_E_SHAPE);
    long long total = (long long)N * C * ksize * ksize;
    pack_conv_weight_kernel<<<nbp_ew_grid(total, 256), 256, 0, (hipStream_t)stream>>>(w_oihw, N, C, ksize * ksize,
                                                                                      scale_or_null, c_off, dst);
    return nbp_launch_status();
}

// ------------------------------------------------------------------ Conv1.conv.0 (5 -> 64, NCHW in)
// 256 threads = 64 pixels x 4 groups of 16 output channels; a wave is one channel group so
// the weight reads from LDS are wave-uniform broadcasts.
__global__ __launch_bounds__(256) void conv_first_kernel(const float* __restrict__ x, int B, int H, int W,
                                                         const float* __restrict__ w, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float wl[45 * 64];   // [k = ci*9+tap][co]
    for (int i = threadIdx.x; i < 45 * 64; i += 256) {
        int co = i & 63, k = i >> 6;
        wl[i] = w[co * 45 + k];
    }
    __syncthreads();
    const int g = threadIdx.x >> 6;
    const long long HW = (long long)H * W;
    const long long m = (long long)blockIdx.x * 64 + (threadIdx.x & 63);
    if (m >= (long long)B * HW) return;
    const int b = (int)(m / HW);
    const int rem = (int)(m - b * HW);
    const int y = rem / W, xx = rem - y * W;
    float in[45];
#pragma unroll
    for (int ci = 0; ci < 5; ++ci)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            int yy = y + t / 3 - 1, xc = xx + t % 3 - 1;
            bool ok = (unsigned)yy < (unsigned)H && (unsigned)xc < (unsigned)W;
            in[ci * 9 + t] = ok ? x[((long long)(b * 5 + ci) * H + yy) * W + xc] : 0.f;
        }
    float acc[16];
#pragma unroll
    for (int o = 0; o < 16; ++o) acc[o] = 0.f;
#pragma unroll
    for (int k = 0; k < 45; ++k) {
        const float v = in[k];
#pragma unroll
        for (int o4 = 0; o4 < 4; ++o4) {
            f32x4 wv = *reinterpret_cast<const f32x4*>(&wl[k * 64 + g * 16 + o4 * 4]);
            acc[o4 * 4 + 0] = fmaf(v, wv[0], acc[o4 * 4 + 0]);
            acc[o4 * 4 + 1] = fmaf(v, wv[1], acc[o4 * 4 + 1]);
            acc[o4 * 4 + 2] = fmaf(v, wv[2], acc[o4 * 4 + 2]);
            acc[o4 * 4 + 3] = fmaf(v, wv[3], acc[o4 * 4 + 3]);
        }
    }
    float* op = out + m * 64 + g * 16;
#pragma unroll
    for (int o4 = 0; o4 < 4; ++o4) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int co = g * 16 + o4 * 4 + e;
            v[e] = fmaxf(acc[o4 * 4 + e] * scale[co] + shift[co], 0.f);
        }
        *reinterpret_cast<f32x4*>(op + o4 * 4) = v;
    }
}

__global__ __launch_bounds__(256) void conv_first_mfma_kernel(const float* __restrict__ x, int B, int H, int W,
                                                              const float* __restrict__ w, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, float* __restrict__ out,
                                                              unsigned* __restrict__ amax_out) {
    conv_first_mfma_body<float>(x, B, H, W, w, scale, shift, out,
                                [](float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }, amax_out);
}

__global__ __launch_bounds__(256) void conv_first_mfma_linear_kernel(const float* __restrict__ x, int B, int H, int W,
                                                                     const float* __restrict__ w, const float* __restrict__ scale,
                                                                     const float* __restrict__ shift, float* __restrict__ out) {
    auto st4 = [](float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; };
    conv_first_mfma_body<float, decltype(st4), false>(x, B, H, W, w, scale, shift, out, st4, nullptr);
}

// Workgroups of the 8 x 32-tile kernel: it loops over tiles, and a workgroup's fixed cost (launch, weights into LDS, cold
// instruction fetch: ~11 us of the ~19 us a one-tile workgroup takes) is paid once per workgroup, so no more workgroups than
// two per CU (NBP_FIRST_GRID overrides)
static unsigned first_grid(long long M) {
    constexpr int cap = 512;
    const long long tiles = M / 256;
    return (unsigned)(tiles < cap ? tiles : cap);
}

// nbp_conv_first_f32 that also leaves max |out| in the 64 words of amax_out (zeroed by the caller); returns 1 in *did_amax
// when the kernel that ran could do it (the 8 x 32-tile MFMA kernel), 0 otherwise
int nbp_conv_first_amax_launch(const float* x_nchw, int B, int H, int W, const float* w_oihw, const float* scale, const float* shift,
                               float* out_nhwc, unsigned* amax_out, int* did_amax, hipStream_t st) {
    long long M = (long long)B * H * W;
    *did_amax = 0;
    if ((H & 7) == 0 && (W & 31) == 0) {
        conv_first_mfma_kernel<<<first_grid(M), 256, 0, st>>>(x_nchw, B, H, W, w_oihw, scale, shift, out_nhwc, amax_out);
        *did_amax = amax_out ? 1 : 0;
        return nbp_launch_status();
    }
    return nbp_conv_first_f32(x_nchw, B, H, W, w_oihw, scale, shift, out_nhwc, st);
}

extern "C" int nbp_conv_first_f32(const float* x_nchw, int B, int H, int W, const float* w_oihw, const float* scale,
                                  const float* shift, float* out_nhwc, void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(!x_nchw || !w_oihw || !scale || !shift || !out_nhwc, NBP_E_ARG);
    NBP_RETURN_IF(B < 1 || H < 1 || W < 1, NBP_E_ARG);
    long long M = (long long)B * H * W;
    if ((H & 7) == 0 && (W & 31) == 0) {     // 8 x 32 pixel tiles; other sizes take the VALU kernel below
        conv_first_mfma_kernel<<<first_grid(M), 256, 0, (hipStream_t)stream>>>(x_nchw, B, H, W, w_oihw, scale, shift, out_nhwc, nullptr);
        return nbp_launch_status();
    }
    conv_first_kernel<<<(unsigned)nbp_cdiv(M, 64), 256, 0, (hipStream_t)stream>>>(x_nchw, B, H, W, w_oihw, scale,
                                                                                   shift, out_nhwc);
    return nbp_launch_status();
}

// The layer WITHOUT the ReLU (training: out = conv(x) * scale + shift feeds a train-mode BatchNorm); H % 8 == 0, W % 32 == 0.
extern "C" int nbp_conv_first_linear_f32(const float* x_nchw, int B, int H, int W, const float* w_oihw, const float* scale,
                                         const float* shift, float* out_nhwc, void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(!x_nchw || !w_oihw || !scale || !shift || !out_nhwc || B < 1, NBP_E_ARG);
    NBP_RETURN_IF(H < 8 || W < 32 || (H & 7) || (W & 31), NBP_E_SHAPE);
    conv_first_mfma_linear_kernel<<<first_grid((long long)B * H * W), 256, 0, (hipStream_t)stream>>>(x_nchw, B, H, W, w_oihw, scale, shift, out_nhwc);
    return nbp_launch_status();
}

// ------------------------------------------------------------------ MaxPool2d(2,2), NHWC
__global__ __launch_bounds__(256) void maxpool2_kernel(const float* __restrict__ in, int B, int H, int W, int C4,
                                                       float* __restrict__ out) {
    const int Ho = H >> 1, Wo = W >> 1;
    const long long total = (long long)B * Ho * Wo * C4;
    const f32x4* in4 = reinterpret_cast<const f32x4*>(in);
    f32x4* out4 = reinterpret_cast<f32x4*>(out);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        int c = (int)(i % C4);
        long long p = i / C4;
        int xo = (int)(p % Wo);
        long long q = p / Wo;
        int yo = (int)(q % Ho);
        int b = (int)(q / Ho);
        long long base = (((long long)b * H + 2 * yo) * W + 2 * xo) * C4 + c;
        f32x4 v0 = in4[base], v1 = in4[base + C4], v2 = in4[base + (long long)W * C4],
              v3 = in4[base + (long long)W * C4 + C4];
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = fmaxf(fmaxf(v0[e], v1[e]), fmaxf(v2[e], v3[e]));
        out4[i] = r;
    }
}

extern "C" int nbp_maxpool2_nhwc_f32(const float* in, int B, int H, int W, int C, float* out, void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(!in || !out, NBP_E_ARG);
    NBP_RETURN_IF(B < 1 || H < 2 || W < 2 || (H & 1) || (W & 1) || C < 4 || (C & 3), NBP_E_SHAPE);
    long long total = (long long)B * (H / 2) * (W / 2) * (C / 4);
    maxpool2_kernel<<<nbp_ew_grid(total, 256), 256, 0, (hipStream_t)stream>>>(in, B, H, W, C / 4, out);
    return nbp_launch_status();
}

// ------------------------------------------------------------------ attention gate tail
// 16 lanes per pixel: psi = sigmoid((q . w) * s + t); out = x * psi.
__global__ __launch_bounds__(256) void psi_gate_kernel(const float* __restrict__ q, int F4,
                                                       const float* __restrict__ wpsi, const float* __restrict__ st,
                                                       const float* __restrict__ x, int C4, long long M,
                                                       float* __restrict__ out) {
    const int sub = threadIdx.x & 15;
    const f32x4* q4 = reinterpret_cast<const f32x4*>(q);
    const f32x4* w4 = reinterpret_cast<const f32x4*>(wpsi);
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    f32x4* o4 = reinterpret_cast<f32x4*>(out);
    const float s = st[0], t = st[1];
    const long long ppb = blockDim.x >> 4;
    for (long long m = (long long)blockIdx.x * ppb + (threadIdx.x >> 4); m < M; m += (long long)gridDim.x * ppb) {
        float acc = 0.f;
        for (int f = sub; f < F4; f += 16) {
            f32x4 a = q4[m * F4 + f], b = w4[f];
            acc = fmaf(a[0], b[0], acc); acc = fmaf(a[1], b[1], acc);
            acc = fmaf(a[2], b[2], acc); acc = fmaf(a[3], b[3], acc);
        }
        acc += __shfl_xor(acc, 8, 16);
        acc += __shfl_xor(acc, 4, 16);
        acc += __shfl_xor(acc, 2, 16);
        acc += __shfl_xor(acc, 1, 16);
        const float z = acc * s + t;
        const float psi = 1.f / (1.f + expf(-z));
        for (int c = sub; c < C4; c += 16) o4[m * C4 + c] = x4[m * C4 + c] * psi;
    }
}

extern "C" int nbp_psi_gate_f32(const float* q, int F, const float* w_psi, const float* s_t2, const float* x, int C,
                                long long M, float* out, void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(!q || !w_psi || !s_t2 || !x || !out, NBP_E_ARG);
    NBP_RETURN_IF(F < 4 || (F & 3) || C < 4 || (C & 3) || M < 1, NBP_E_SHAPE);
    psi_gate_kernel<<<nbp_ew_grid(M * 16, 256), 256, 0, (hipStream_t)stream>>>(q, F / 4, w_psi, s_t2, x, C / 4, M,
                                                                                out);
    return nbp_launch_status();
}

// ------------------------------------------------------------------ final 1x1 (NHWC -> NCHW, n_out <= 8)
template <int NO>
__global__ __launch_bounds__(256) void final_1x1_kernel(const float* __restrict__ in, int B, int H, int W, int C4,
                                                        const float* __restrict__ w, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, int sigmoid,
                                                        float* __restrict__ out) {
    const int sub = threadIdx.x & 15;
    const f32x4* in4 = reinterpret_cast<const f32x4*>(in);
    const f32x4* w4 = reinterpret_cast<const f32x4*>(w);
    const long long HW = (long long)H * W, M = (long long)B * HW;
    const long long ppb = blockDim.x >> 4;
    for (long long m = (long long)blockIdx.x * ppb + (threadIdx.x >> 4); m < M; m += (long long)gridDim.x * ppb) {
        float acc[NO];
#pragma unroll
        for (int o = 0; o < NO; ++o) acc[o] = 0.f;
        for (int c = sub; c < C4; c += 16) {
            f32x4 v = in4[m * C4 + c];
#pragma unroll
            for (int o = 0; o < NO; ++o) {
                f32x4 ww = w4[o * C4 + c];
                acc[o] = fmaf(v[0], ww[0], acc[o]); acc[o] = fmaf(v[1], ww[1], acc[o]);
                acc[o] = fmaf(v[2], ww[2], acc[o]); acc[o] = fmaf(v[3], ww[3], acc[o]);
            }
        }
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            acc[o] += __shfl_xor(acc[o], 8, 16);
            acc[o] += __shfl_xor(acc[o], 4, 16);
            acc[o] += __shfl_xor(acc[o], 2, 16);
            acc[o] += __shfl_xor(acc[o], 1, 16);
        }
        if (sub == 0) {
            const long long b = m / HW, rem = m - b * HW;
#pragma unroll
            for (int o = 0; o < NO; ++o) {
                float v = acc[o] * scale[o] + shift[o];
                if (sigmoid) v = 1.f / (1.f + expf(-v));
                out[(b * NO + o) * HW + rem] = v;
            }
        }
    }
}

extern "C" int nbp_final_1x1_f32(const float* in, int B, int H, int W, int C, const float* w_oc, int n_out,
                                 const float* scale, const float* shift, int sigmoid, float* out_nchw, void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(!in || !w_oc || !scale || !shift || !out_nchw, NBP_E_ARG);
    NBP_RETURN_IF(B < 1 || H < 1 || W < 1 || C < 4 || (C & 3), NBP_E_SHAPE);
    long long M = (long long)B * H * W;
    int grid = nbp_ew_grid(M * 16, 256);
    hipStream_t st = (hipStream_t)stream;
    if (n_out == 8) final_1x1_kernel<8><<<grid, 256, 0, st>>>(in, B, H, W, C / 4, w_oc, scale, shift, sigmoid, out_nchw);
    else if (n_out == 1) final_1x1_kernel<1><<<grid, 256, 0, st>>>(in, B, H, W, C / 4, w_oc, scale, shift, sigmoid, out_nchw);
    else return NBP_E_SHAPE;
    return nbp_launch_status();
}

// ------------------------------------------------------------------ layout helpers
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ in, int B, int C, long long HW, float* __restrict__ out) {
    const long long total = (long long)B * C * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        int c = (int)(i % C);
        long long p = i / C;
        long long b = p / HW, hw = p - b * HW;
        out[i] = in[(b * C + c) * HW + hw];
    }
}
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ in, int B, int C, long long HW, float* __restrict__ out) {
    const long long total = (long long)B * C * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        long long hw = i % HW;
        long long p = i / HW;
        int c = (int)(p % C);
        long long b = p / C;
        out[i] = in[(b * HW + hw) * C + c];
    }
}
extern "C" int nbp_nchw_to_nhwc_f32(const float* in, int B, int C, int H, int W, float* out, void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(!in || !out || B < 1 || C < 1 || H < 1 || W < 1, NBP_E_ARG);
    long long total = (long long)B * C * H * W;
    nchw_to_nhwc_kernel<<<nbp_ew_grid(total, 256), 256, 0, (hipStream_t)stream>>>(in, B, C, (long long)H * W, out);
    return nbp_launch_status();
}
extern "C" int nbp_nhwc_to_nchw_f32(const float* in, int B, int C, int H, int W, float* out, void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(!in || !out || B < 1 || C < 1 || H < 1 || W < 1, NBP_E_ARG);
    long long total = (long long)B * C * H * W;
    nhwc_to_nchw_kernel<<<nbp_ew_grid(total, 256), 256, 0, (hipStream_t)stream>>>(in, B, C, (long long)H * W, out);
    return nbp_launch_status();
}
