// nbp_bf16.hip -- the bf16 variant of the NBP convolutions (BASELINE.json configs[4]: 512x512 grids, 8 rollouts
// per GPU batched through one bf16 forward).  Activations are NHWC bf16, weights bf16, accumulation fp32 on
// v_mfma_f32_32x32x16_bf16, epilogue (folded BatchNorm + bias, ReLU) in fp32, one rounding (nearest-even) per
// stored activation.  Reference ops: conv_block / up_conv / Attention_block, next_best_path/networks/nbp_model.py:8-62.
//
// Differences from the fp32 kernel (nbp_conv.hip) that the 16x higher MFMA rate forces:
//   * K chunks are 64 channels (128-B rows in LDS, the same 8-slot XOR-swizzled image as the fp32 kernel);
//   * staging is buffer_load ... lds (16 B per lane straight into LDS; no staging VGPRs, no ds_write pass);
//     padding taps and rows past M use an out-of-range buffer offset, which the DMA writes as zeros;
//   * the MFMA runs "transposed" (A operand = weights, B operand = pixels) so that a lane ends up with four
//     consecutive output channels of one pixel: 8-B bf16 stores (16-B fp32 stores for split-K partial sums).
#include "common.h"
#include "nbp_internal.h"
#include "nbp_first_conv.h"
#include <cstdlib>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ unsigned short f2bf(float f) {   // round to nearest even (inputs are finite)
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

struct IgemmArgsH {
    const bf16_t* src0;
    const bf16_t* src1;
    int C0, C1;        // channels of each source (multiples of 64; C1 may be 0)
    int cc0;           // C0 / 64
    int ups;
    int H, W, Hs, Ws;
    int taps;
    const bf16_t* wpk; // [(C0+C1)/64][taps][N][64]
    int N;
    const float* scale;
    const float* shift;
    int relu;
    bf16_t* out;       // [M][N]
    long long M;
    int split_k, chunks_total, chunks_per_split;
    unsigned bytes0, bytes1, bytesw;
    float* partial;    // split-K scratch [group][split][M][N] fp32
    int groups;
    const bf16_t* g_src0;
    const bf16_t* g_src1;
    const bf16_t* g_wpk;
    const float* g_scale;
    const float* g_shift;
    bf16_t* g_out;
    int xcd_remap;     // halo kernel: XCD-contiguous (pixel tile, channel block) runs
};

template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(256, 2) void igemm_bf16_kernel(IgemmArgsH a, GatePsiH ps) {
    int zs = blockIdx.z;
    const int grp = zs >= a.split_k ? 1 : 0;
    if (zs >= a.split_k) {
        zs -= a.split_k;
        a.src0 = a.g_src0; a.src1 = a.g_src1; a.wpk = a.g_wpk; a.scale = a.g_scale; a.shift = a.g_shift; a.out = a.g_out;
    }
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int RA = BM / 32, RB = BN / 32;
    constexpr int STAGE = (BM + BN) * 128;  // bytes per LDS stage
    static_assert(WM * WN == 4, "256-thread workgroup");
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const long long m0 = (long long)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int c_begin = zs * a.chunks_per_split;
    const int c_end = min(c_begin + a.chunks_per_split, a.chunks_total);

    // staging: thread t fills LDS row t/8 (+32 i), physical 16-B slot t%8, with the channels of logical slot
    // (t%8) ^ swz(row): the XOR lives on the SOURCE address, the LDS image of a wave's DMA stays lane-linear.
    const int lrow = tid >> 3;
    const int sslot = (tid & 7) ^ ((lrow >> 1) & 7);
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
            py[i] = -1000000; px[i] = 0; pb[i] = 0;
        }
    }
    const __amdgpu_buffer_rsrc_t rs0 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.src0), 0, a.bytes0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.src1 ? a.src1 : a.src0), 0, a.bytes1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.wpk), 0, a.bytesw, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;

    auto issue = [&](int c, int buf) {
        const int cc = c / a.taps;
        const int tap = c - cc * a.taps;
        int dy = 0, dx = 0;
        if (a.taps == 9) { dy = tap / 3 - 1; dx = tap - (tap / 3) * 3 - 1; }
        const bool first = cc < a.cc0;
        const int Cs = first ? a.C0 : a.C1;
        const int coff = (first ? cc : cc - a.cc0) * 64 + sslot * 8;
        char* A = lds + buf * STAGE + wave * (8 * 128);
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int yy = py[i] + dy, xx = px[i] + dx;
            const bool ok = (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
            const int pix = pb[i] + (yy >> a.ups) * a.Ws + (xx >> a.ups);
            const unsigned off = ok ? (unsigned)(pix * Cs + coff) * 2u : OOB;
            if (first) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (lds_ptr_t)(A + i * (32 * 128)), 16, off, 0, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lds_ptr_t)(A + i * (32 * 128)), 16, off, 0, 0, 0);
        }
        char* Bt = A + BM * 128;
        const unsigned woff = (unsigned)(((long long)c * a.N + n0 + lrow) * 64 + sslot * 8) * 2u;
#pragma unroll
        for (int i = 0; i < RB; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr_t)(Bt + i * (32 * 128)), 16, woff + i * (32 * 128), 0, 0, 0);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int fa_row[TM], fb_row[TN], fa_sw[TM], fb_sw[TN];   // byte offsets of the fragment rows, swizzle keys
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        int r = (wm * TM + i) * 32 + (lane & 31);
        fa_row[i] = r * 128; fa_sw[i] = (r >> 1) & 7;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        int r = (wn * TN + j) * 32 + (lane & 31);
        fb_row[j] = r * 128; fb_sw[j] = (r >> 1) & 7;
    }
    const int khalf = lane >> 5;

    if (c_begin < c_end) issue(c_begin, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int c = c_begin; c < c_end; ++c) {
        if (c + 1 < c_end) issue(c + 1, cur ^ 1);     // DMA into the other stage while this one is multiplied
        const char* A = lds + cur * STAGE;
        const char* Bt = A + BM * 128;
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4) {
            const int s = 2 * j4 + khalf;
            bf16x8 xf[TM], wf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                xf[i] = *reinterpret_cast<const bf16x8*>(A + fa_row[i] + ((s ^ fa_sw[i]) << 4));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                wf[j] = *reinterpret_cast<const bf16x8*>(Bt + fb_row[j] + ((s ^ fb_sw[j]) << 4));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue.  D[n][m] of the 32x32 MFMA: col (pixel) = lane&31, row (channel) = (r&3)+8*(r>>2)+4*(lane>>5)
    const bool final_out = (a.split_k == 1);
    if constexpr (WN == 1) {
        if (ps.wpsi[0]) {
            // attention gate tail (nbp_model.py:55-61): psi = sigmoid(BN(q . w_psi)), gated = x * psi, with q = this wave's BN = N
            // columns of its TM x 32 pixels -- rounded to bf16 as the separate kernel would have read it; q itself is not written
            const float* wp = ps.wpsi[grp];
            float d[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) d[i] = 0.f;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int n = j * 32 + 8 * rq + 4 * khalf;
                    const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + n);
                    const f32x4 sh = *reinterpret_cast<const f32x4*>(a.shift + n);
                    const f32x4 w4 = *reinterpret_cast<const f32x4*>(wp + n);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float t = fmaf(acc[i][j][4 * rq + e], sc[e], sh[e]);
                            if (a.relu) t = fmaxf(t, 0.f);
                            d[i] = fmaf(bf2f(f2bf(t)), w4[e], d[i]);
                        }
                }
            float* psil = reinterpret_cast<float*>(lds) + wave * (TM * 32);     // the stages are free: the loop ended on a barrier
            const float s0 = ps.st[grp][0], t0 = ps.st[grp][1];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const float dot = d[i] + __shfl_xor(d[i], 32);
                if (!khalf) psil[i * 32 + (lane & 31)] = 1.f / (1.f + expf(-(dot * s0 + t0)));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // the wave reads back only what it wrote itself
            // gated = x * psi over the wave's TM x 32 pixels x C channels, 16 bytes per lane, consecutive lanes consecutive pieces;
            // x = source 1, read a moment ago as the second half of K (L2 hits)
            const int C8 = a.C1 >> 3;
            const long long mw = m0 + (long long)wm * TM * 32;
            const u16x8* x8 = reinterpret_cast<const u16x8*>(a.src1) + mw * C8;
            u16x8* g8 = reinterpret_cast<u16x8*>(ps.gated[grp]) + mw * C8;
            const long long lim = (a.M - mw) * C8;
            const int total = TM * 32 * C8;
#pragma unroll 4
            for (int idx = lane; idx < total; idx += 64) {
                if (idx >= lim) break;
                const u16x8 v = x8[idx];
                const float psi = psil[idx / C8];
                u16x8 r;
#pragma unroll
                for (int e = 0; e < 8; ++e) r[e] = f2bf(bf2f(v[e]) * psi);
                g8[idx] = r;
            }
            return;
        }
    }
    float* part = final_out ? nullptr : a.partial + (long long)blockIdx.z * a.M * a.N;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const long long m = m0 + (wm * TM + i) * 32 + (lane & 31);
        if (m >= a.M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int n = n0 + (wn * TN + j) * 32 + 8 * rq + 4 * khalf;
                f32x4 v = {acc[i][j][4 * rq], acc[i][j][4 * rq + 1], acc[i][j][4 * rq + 2], acc[i][j][4 * rq + 3]};
                if (final_out) {
                    const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + n);
                    const f32x4 sh = *reinterpret_cast<const f32x4*>(a.shift + n);
                    u16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = fmaf(v[e], sc[e], sh[e]);
                        if (a.relu) t = fmaxf(t, 0.f);
                        o[e] = f2bf(t);
                    }
                    *reinterpret_cast<u16x4*>(a.out + m * a.N + n) = o;
                } else {
                    *reinterpret_cast<f32x4*>(part + m * a.N + n) = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------ 3x3 convolution from an LDS-resident halo tile
// The implicit-GEMM kernel above re-reads every input pixel nine times (once per tap) through the CU's vector
// memory path, which at 128x128 tiles is as busy as the matrix pipe (64 B/clk/CU).  Here a workgroup owns an
// 8 x 32 pixel tile of ONE image and BN output channels; per 64-channel chunk it DMAs the 10 x 34 pixel halo tile
// into LDS once and runs all nine taps from it (A fragments are the halo rows shifted by the tap), streaming only
// the weights of the next tap (double buffered) behind the current tap's MFMAs.  Vector-memory traffic per chunk
// drops from 9*(256+BN)*128 B to (344 + 9*BN)*128 B.  Two workgroups share a CU (<= 76 KB of LDS each), so one
// covers the other's halo reload.  Same K order as the implicit GEMM (chunk, tap, k): results are bit-identical.
// TPS = filter taps per weight stage (= per barrier): the 64-channel variant pairs taps so that a wave still has 32 MFMAs
// between barriers (its 16 per tap finish in 512 cycles, too close to the barrier + first-fragment latency).
// PH (up_conv = x2 nearest upsample + 3x3, nbp_model.py:25-33): the four output parities are four 2x2 convolutions of the
// LOW-resolution input with pre-summed weights (nbp_split.hip has the fp32 form and the derivation); a workgroup owns one
// parity (blockIdx.z & 3) of an 8 x 32 low-resolution tile and writes its outputs to (2 v + py, 2 u + px).
// epilogue fusions (as the split path has them): the encoder's 2 x 2 max-pool and the one-channel sigmoid head
struct RowsFuse {
    bf16_t* pool_out;           // [B, H/2, W/2, N] or null
    const float* head_w;        // [64] or null: out1[m] = sigmoid((out[m,:] . w) * head_scale[0] + head_shift[0]) INSTEAD of `out`
    const float* head_scale;
    const float* head_shift;
    float* head_out;            // [M]
};

template <int TN, int TPS, bool PH>
__global__ __launch_bounds__(256, 2) void conv3x3_halo_bf16_kernel(IgemmArgsH a, RowsFuse f) {
    int zs = blockIdx.z;        // [parity,] split-K slice (of 64-channel chunks), then the group
    const int py = PH ? (zs >> 1) & 1 : 0, px = PH ? zs & 1 : 0;
    if (PH) zs >>= 2;
    const int zslice = zs;
    constexpr int TAPS = PH ? 4 : 9, TPR = PH ? 2 : 3;
    if (zs >= a.split_k) {
        zs -= a.split_k;
        a.src0 = a.g_src0; a.src1 = a.g_src1; a.wpk = a.g_wpk; a.scale = a.g_scale; a.shift = a.g_shift; a.out = a.g_out;
    }
    constexpr int BN = TN * 32;
    constexpr int HW_ = 34;                    // halo tile width (32 + 2)
    constexpr int HALO_ROWS = 344;             // 10 * 34 = 340 halo pixels, padded to 43 DMA instructions of 8 rows
    constexpr int HALO_BYTES = HALO_ROWS * 128;
    constexpr int WB = TPS * BN * 128;         // one weight stage: TPS taps x BN rows of 64 bf16
    constexpr int NST = (TAPS + TPS - 1) / TPS;   // stages per 64-channel chunk
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* const halo = lds;
    char* const wbuf = lds + HALO_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Ht = PH ? a.Hs : a.H, Wt = PH ? a.Ws : a.W;      // the tile grid: output pixels, or low-resolution pixels for PH
    const int tiles_x = Wt >> 5, tiles_y = Ht >> 3;
    unsigned tile = blockIdx.x, nt = blockIdx.y;
    if (a.xcd_remap) {      // XCD-contiguous runs of (pixel tile, channel block), channel block fastest (see nbp_conv.hip)
        const unsigned L = blockIdx.x + gridDim.x * blockIdx.y, T = gridDim.x * gridDim.y;
        const unsigned xcd = L & 7u, idx = L >> 3, q = T >> 3, r = T & 7u;
        const unsigned v = xcd * q + min(xcd, r) + idx;
        nt = v % gridDim.y;
        tile = v / gridDim.y;
    }
    const int tx = tile % tiles_x; tile /= tiles_x;
    const int ty = tile % tiles_y;
    const int b = tile / tiles_y;
    const int y0 = ty * 8, x0 = tx * 32;
    const int n0 = nt * BN;

    // ---- halo DMA coordinates: instruction q = 4 i + wave covers halo pixels 8 q .. 8 q + 7 (row-major 10 x 34)
    int hpix[11];
#pragma unroll
    for (int i = 0; i < 11; ++i) {
        const int q = 4 * i + wave;
        const int hr = 8 * q + (lane >> 3);
        const int hy = hr / HW_, hx = hr - hy * HW_;
        const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
        const bool ok = q < 43 && hr < 340 && (unsigned)yy < (unsigned)Ht && (unsigned)xx < (unsigned)Wt;
        hpix[i] = ok ? (PH ? (b * a.Hs + yy) * a.Ws + xx : (b * a.Hs + (yy >> a.ups)) * a.Ws + (xx >> a.ups)) : -1;
    }
    const int hslot = lane & 7;                 // physical 16-B slot; logical = hslot ^ swz(halo row)
    const int lrow = tid >> 3;
    const int wslot = (tid & 7) ^ ((lrow >> 1) & 7);
    const __amdgpu_buffer_rsrc_t rs0 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.src0), 0, a.bytes0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.src1 ? a.src1 : a.src0), 0, a.bytes1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.wpk), 0, a.bytesw, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;

    auto issue_halo = [&](int cc) {
        const bool first = cc < a.cc0;
        const int Cs = first ? a.C0 : a.C1;
        const int cbase = (first ? cc : cc - a.cc0) * 64;
#pragma unroll
        for (int i = 0; i < 11; ++i) {
            const int q = 4 * i + wave;
            if (q < 43) {
                const int sw = (4 * q + (lane >> 4)) & 7;                  // ((8 q + lane/8) >> 1) & 7
                const unsigned off = hpix[i] >= 0 ? (unsigned)(hpix[i] * Cs + cbase + ((hslot ^ sw) << 3)) * 2u : OOB;
                if (first) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (lds_ptr_t)(halo + q * 1024), 16, off, 0, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lds_ptr_t)(halo + q * 1024), 16, off, 0, 0, 0);
            }
        }
    };
    // PH: the four parities' weights follow each other, each [chunk][4 taps][N][64]
    const long long pbase = PH ? (long long)(py * 2 + px) * (a.chunks_total / TAPS) * TAPS : 0;
    auto issue_w = [&](int u) {      // u = chunk * NST + stage; the packed weights are [chunk][tap][N][64]
        const int cu = u / NST, su = u - cu * NST;
        char* dst = wbuf + (u & 1) * WB + wave * 1024;
#pragma unroll
        for (int tt = 0; tt < TPS; ++tt) {
            const int tap = su * TPS + tt;
            if (tap < TAPS) {
                const unsigned woff = (unsigned)(((pbase + cu * TAPS + tap) * a.N + n0 + lrow) * 64 + wslot * 8) * 2u;
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr_t)(dst + tt * (BN * 128) + j * 4096), 16,
                                                             woff + j * 4096, 0, 0, 0);
            }
        }
    };

    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
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
    const int hbase = (2 * wave) * HW_ + (lane & 31);     // halo row of this lane's pixel for tap (-1,-1), M tile 0

    const int cc_begin = zs * (a.chunks_per_split / TAPS);
    const int chunks = min(cc_begin + a.chunks_per_split / TAPS, a.chunks_total / TAPS);
    const int u_total = chunks * NST;
    if (cc_begin < chunks) {
        issue_halo(cc_begin);
        issue_w(cc_begin * NST);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int cc = cc_begin; cc < chunks; ++cc) {
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            const int u = cc * NST + st;
            if (u + 1 < u_total) issue_w(u + 1);
#pragma unroll
            for (int tt = 0; tt < TPS; ++tt) {
                const int tap = st * TPS + tt;
                if (tap < TAPS) {
                    const char* Bt = wbuf + (u & 1) * WB + tt * (BN * 128);
                    const int hr0 = hbase + (tap / TPR + py) * HW_ + (tap % TPR) + px;
                    const int hr1 = hr0 + HW_;
                    const int ar0 = hr0 * 128, as0 = (hr0 >> 1) & 7, ar1 = hr1 * 128, as1 = (hr1 >> 1) & 7;
#pragma unroll
                    for (int j4 = 0; j4 < 4; ++j4) {
                        const int s = 2 * j4 + khalf;
                        const bf16x8 x0f = *reinterpret_cast<const bf16x8*>(halo + ar0 + ((s ^ as0) << 4));
                        const bf16x8 x1f = *reinterpret_cast<const bf16x8*>(halo + ar1 + ((s ^ as1) << 4));
                        bf16x8 wf[TN];
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            wf[j] = *reinterpret_cast<const bf16x8*>(Bt + fb_row[j] + ((s ^ fb_sw[j]) << 4));
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j], x0f, acc[0][j], 0, 0, 0);
                            acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j], x1f, acc[1][j], 0, 0, 0);
                        }
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        if (cc + 1 < chunks) {          // every wave is past its last read of this chunk's halo tile
            issue_halo(cc + 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }

    // ---- epilogue: lane = pixel (lane & 31) of image row y0 + 2 wave + i, four consecutive channels per quad
    const bool final_out = (a.split_k == 1);
    if (!final_out) {
        float* part = a.partial + (long long)zslice * a.M * a.N;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long long m = PH ? ((long long)b * a.H + 2 * (y0 + 2 * wave + i) + py) * a.W + 2 * (x0 + (lane & 31)) + px
                                   : ((long long)b * a.H + y0 + 2 * wave + i) * a.W + x0 + (lane & 31);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int n = n0 + j * 32 + 8 * rq + 4 * khalf;
                    const f32x4 v = {acc[i][j][4 * rq], acc[i][j][4 * rq + 1], acc[i][j][4 * rq + 2], acc[i][j][4 * rq + 3]};
                    *reinterpret_cast<f32x4*>(part + m * a.N + n) = v;
                }
        }
        return;
    }
    // A lane holds 8-byte pieces of 32 different pixels: stored directly, a store instruction touches 32 lines with 16 bytes each,
    // and the L2 takes one request per piece (measured on the 64-channel layers at 512 x 512: the stores cost more than the tile's
    // MFMAs).  Each wave transposes its 2 rows x 32 pixels x BN channels through LDS instead (the staging buffers are idle now;
    // pixel rows padded by 16 bytes: 2-way conflicts at most) and stores 16 bytes per lane: whole lines.
    constexpr int TRB = BN * 2 + 16;                  // bytes per pixel row of the transpose image
    constexpr int PIECES = BN / 8;                    // 16-byte pieces per pixel
    static_assert(4 * 64 * TRB <= HALO_BYTES + 2 * WB, "transpose image fits the staging buffers");
    char* const tr = lds + wave * (64 * TRB);
    const bool head = TN == 2 && !PH && f.head_w != nullptr;       // 64 columns = all the head's input channels in one workgroup
    float hdot[2] = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int nl = j * 32 + 8 * rq + 4 * khalf;
                const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + n0 + nl);
                const f32x4 sh = *reinterpret_cast<const f32x4*>(a.shift + n0 + nl);
                f32x4 hw = {0.f, 0.f, 0.f, 0.f};
                if (head) hw = *reinterpret_cast<const f32x4*>(f.head_w + n0 + nl);
                u16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = fmaf(acc[i][j][4 * rq + e], sc[e], sh[e]);
                    if (a.relu) v = fmaxf(v, 0.f);
                    o[e] = f2bf(v);
                    // the head reads what the next layer would have read: the bf16-rounded activations
                    if (head) hdot[i] = fmaf(bf2f(o[e]), hw[e], hdot[i]);
                }
                if (!head) *reinterpret_cast<u16x4*>(tr + (i * 32 + (lane & 31)) * TRB + nl * 2) = o;
            }
    if (head) {
        const float hs = f.head_scale[0], ht = f.head_shift[0];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float d = hdot[i] + __shfl_xor(hdot[i], 32);
            const long long m = ((long long)b * a.H + y0 + 2 * wave + i) * a.W + x0 + (lane & 31);
            if (!khalf) f.head_out[m] = 1.f / (1.f + expf(-(d * hs + ht)));
        }
        return;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the wave reads back only what it wrote itself
#pragma unroll
    for (int it = 0; it < PIECES; ++it) {
        const int idx = it * 64 + lane, p = idx / PIECES, slot = idx % PIECES;      // pixel p = row p / 32, column p % 32
        const u16x8 o = *reinterpret_cast<const u16x8*>(tr + p * TRB + slot * 16);
        const long long m = PH ? ((long long)b * a.H + 2 * (y0 + 2 * wave + (p >> 5)) + py) * a.W + 2 * (x0 + (p & 31)) + px
                               : ((long long)b * a.H + y0 + 2 * wave + (p >> 5)) * a.W + x0 + (p & 31);
        *reinterpret_cast<u16x8*>(a.out + m * a.N + n0 + slot * 8) = o;
    }
    if (!PH && f.pool_out) {       // 2 x 2 max-pool from the same image (post-ReLU bf16: the bit patterns order like the values)
        const int Hp = a.H >> 1, Wp = a.W >> 1;
#pragma unroll
        for (int it = 0; it < PIECES / 4; ++it) {
            const int idx = it * 64 + lane, pc = idx / PIECES, slot = idx % PIECES;      // pooled column pc of the wave's one pooled row
            const char* src = tr + (pc * 2) * TRB + slot * 16;
            const u16x8 v00 = *reinterpret_cast<const u16x8*>(src), v01 = *reinterpret_cast<const u16x8*>(src + TRB);
            const u16x8 v10 = *reinterpret_cast<const u16x8*>(src + 32 * TRB), v11 = *reinterpret_cast<const u16x8*>(src + 33 * TRB);
            u16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = max(max(v00[e], v01[e]), max(v10[e], v11[e]));
            const long long mp = ((long long)b * Hp + ((y0 + 2 * wave) >> 1)) * Wp + (x0 >> 1) + pc;
            *reinterpret_cast<u16x8*>(f.pool_out + mp * a.N + n0 + slot * 8) = o;
        }
    }
}

struct ReduceGroupH { const float* scale; const float* shift; bf16_t* out; };
__global__ __launch_bounds__(256) void splitk_reduce_bf16_kernel(const float* __restrict__ partial_all, int split_k,
                                                                 long long MN, int N, ReduceGroupH g0, ReduceGroupH g1,
                                                                 int relu) {
    const ReduceGroupH g = blockIdx.y ? g1 : g0;
    const float* __restrict__ partial = partial_all + (long long)blockIdx.y * split_k * MN;
    const long long n4 = MN >> 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (long long)gridDim.x * blockDim.x) {
        f32x4 s = *reinterpret_cast<const f32x4*>(partial + i * 4);
        for (int k = 1; k < split_k; ++k) s += *reinterpret_cast<const f32x4*>(partial + (long long)k * MN + i * 4);
        const int n = (int)((i * 4) % N);
        const f32x4 sc = *reinterpret_cast<const f32x4*>(g.scale + n);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(g.shift + n);
        u16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = fmaf(s[e], sc[e], sh[e]);
            if (relu) t = fmaxf(t, 0.f);
            o[e] = f2bf(t);
        }
        *reinterpret_cast<u16x4*>(g.out + i * 4) = o;
    }
}

// ------------------------------------------------------------------ planning / launch
static TileInfo tile_info_h(int tile) {
    switch (tile) {
        case NBP_TILE_HALO_128: return {256, 128};
        case NBP_TILE_HALO_64: return {256, 64};
        case NBP_TILE_HALO_UP_128: return {256, 128};
        case NBP_TILE_HALO_UP_64: return {256, 64};
        case NBP_TILE_128x128: return {128, 128};
        case NBP_TILE_256x64: return {256, 64};
        case NBP_TILE_256x32: return {256, 32};
        case NBP_TILE_128x64: return {128, 64};
        case NBP_TILE_64x128: return {64, 128};
        default: return {0, 0};
    }
}

static bool halo_ok(int H, int W, int N, int ksize, int bn) {
    return ksize == 3 && H >= 8 && W >= 32 && (H & 7) == 0 && (W & 31) == 0 && N % bn == 0;
}

ConvPlan nbp_plan_conv_bf16(long long M, int N, int chunks_total, int tile, int split_k, int groups, int H, int W,
                            int ksize, int ups) {
    ConvPlan p;
    if ((tile == NBP_TILE_AUTO && ups && ksize == 3 && split_k <= 0) || tile == NBP_TILE_HALO_UP_128 ||
        tile == NBP_TILE_HALO_UP_64) {
        // up_conv as four parity convolutions of the low-resolution image (8 x 32 low-resolution tiles)
        const int bn = tile == NBP_TILE_HALO_UP_128 ? 128 : tile == NBP_TILE_HALO_UP_64 ? 64 : (N % 128 == 0 ? 128 : 64);
        if (!((H | W) & 1) && halo_ok(H / 2, W / 2, N, 3, bn)) {
            constexpr int min_blocks_up = 128;
            const long long blocks = (M / 4 / 256) * (N / bn) * groups * 4;
            const int cc = chunks_total / 9;
            int sk = split_k <= 0 ? 1 : split_k;
            if (split_k <= 0) while (blocks * sk < min_blocks_up && cc / (sk * 2) >= 4 && sk < 16) sk *= 2;
            if (sk > cc) sk = cc;
            if (sk < 1) sk = 1;
            const int per = cc > 0 ? (int)nbp_cdiv(cc, sk) : 1;
            p.tile = bn == 128 ? NBP_TILE_HALO_UP_128 : NBP_TILE_HALO_UP_64;
            p.split_k = cc > 0 ? (int)nbp_cdiv(cc, per) : 1; p.chunks_per_split = per * 4;      // kernel units: (chunk, 4 taps)
            return p;
        }
        if (tile != NBP_TILE_AUTO) { p.tile = -1; p.split_k = 1; p.chunks_per_split = chunks_total; return p; }
    }
    if (tile == NBP_TILE_AUTO && ksize == 3 && split_k <= 0) {
        // halo-tile kernel once tiles (x split-K over whole chunks) give >= ~128 workgroups
        // (threshold from tools/bench_forward.py sweeps at B = 1..8, S = 256 / 512)
        const int bn = N % 128 == 0 ? 128 : 64;
        constexpr int min_blocks = 128;
        const long long blocks = (M / 256) * (N / bn) * groups;
        const int cc = chunks_total / 9;
        int sk = 1;
        while (blocks * sk < min_blocks && cc / (sk * 2) >= 4 && sk < 16) sk *= 2;
        if (halo_ok(H, W, N, ksize, bn) && blocks * sk >= min_blocks) {
            tile = bn == 128 ? NBP_TILE_HALO_128 : NBP_TILE_HALO_64;
            split_k = sk;
        }
    }
    if (tile == NBP_TILE_HALO_128 || tile == NBP_TILE_HALO_64) {
        const int cc = chunks_total / 9;
        int sk = split_k <= 0 ? 1 : split_k;
        if (sk > cc) sk = cc;
        if (sk < 1) sk = 1;              // 1x1 convolution asked for a halo tile: rejected by the caller's shape check
        const int per = cc > 0 ? (int)nbp_cdiv(cc, sk) : 1;
        p.tile = tile; p.split_k = cc > 0 ? (int)nbp_cdiv(cc, per) : 1; p.chunks_per_split = per * 9;
        return p;
    }
    if (tile == NBP_TILE_AUTO) {
        if (N % 128 == 0)
            tile = nbp_cdiv(M, 128) * (N / 128) * groups >= 512 ? NBP_TILE_128x128 : NBP_TILE_64x128;
        else if (N % 64 == 0)
            tile = nbp_cdiv(M, 256) * (N / 64) * groups >= 512 ? NBP_TILE_256x64 : NBP_TILE_128x64;
        else
            tile = NBP_TILE_256x32;
    }
    p.tile = tile;
    TileInfo ti = tile_info_h(tile);
    if (ti.bm == 0) {                 // unknown tile id: the caller's shape check refuses it
        p.split_k = 1; p.chunks_per_split = chunks_total;
        return p;
    }
    if (split_k <= 0) {
        const long long blocks = nbp_cdiv(M, ti.bm) * (N / ti.bn) * groups;
        split_k = 1;
        while (blocks * split_k < 512 && chunks_total / (split_k * 2) >= 4 && split_k < 64) split_k *= 2;
        if (split_k == 2 && chunks_total / 2 < 12) split_k = 1;
    }
    if (split_k > chunks_total) split_k = chunks_total;
    p.chunks_per_split = (int)nbp_cdiv(chunks_total, split_k);
    p.split_k = (int)nbp_cdiv(chunks_total, p.chunks_per_split);
    return p;
}

template <int TN, int TPS, bool PH = false>
static int launch_halo(const IgemmArgsH& a, const RowsFuse& f, hipStream_t st, int tile) {
    { char nm[96]; snprintf(nm, sizeof(nm), "conv3x3_halo_bf16_kernel<%d, %d, %s>", TN, TPS, PH ? "true" : "false"); nbp_note_kernel_symbol(tile, nm); }
    constexpr size_t smem = 344 * 128 + 2 * (size_t)TPS * TN * 32 * 128;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_bf16_kernel<TN, TPS, PH>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    dim3 grid((unsigned)(a.M / (PH ? 4 : 1) / 256), (unsigned)(a.N / (TN * 32)), (unsigned)(a.split_k * a.groups * (PH ? 4 : 1)));
    conv3x3_halo_bf16_kernel<TN, TPS, PH><<<grid, 256, smem, st>>>(a, f);
    return nbp_launch_status();
}

template <int WM, int WN, int TM, int TN>
static int launch_igemm_h(const IgemmArgsH& a, hipStream_t st, int tile, const GatePsiH& ps = GatePsiH{{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}}) {
    { char nm[96]; snprintf(nm, sizeof(nm), "igemm_bf16_kernel<%d, %d, %d, %d>", WM, WN, TM, TN); nbp_note_kernel_symbol(tile, nm); }
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr size_t smem = 2 * (size_t)(BM + BN) * 128;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_bf16_kernel<WM, WN, TM, TN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    dim3 grid((unsigned)nbp_cdiv(a.M, BM), (unsigned)(a.N / BN), (unsigned)(a.split_k * a.groups));
    igemm_bf16_kernel<WM, WN, TM, TN><<<grid, 256, smem, st>>>(a, ps);
    return nbp_launch_status();
}

int nbp_conv_igemm_bf16_launch_g(const ConvOperandsH& o, const ConvOperandsH* o2, int C0, int C1, int ups, int B, int H,
                                 int W, int ksize, int N, int relu, int split_k, int tile, void* ws, size_t ws_bytes,
                                 hipStream_t st, bf16_t* const* pool_out, int* pooled, const ConvHead* head, int* headed,
                                 const GatePsiH* psi, int* psi_fused) {
    const int groups = o2 ? 2 : 1;
    if (pooled) *pooled = 0;
    if (headed) *headed = 0;
    if (psi_fused) *psi_fused = 0;
    NBP_RETURN_IF(!o.src0 || !o.wpk || !o.scale || !o.shift || !o.out, NBP_E_ARG);
    NBP_RETURN_IF(o2 && (!o2->src0 || !o2->wpk || !o2->scale || !o2->shift || !o2->out), NBP_E_ARG);
    NBP_RETURN_IF(B < 1 || H < 1 || W < 1, NBP_E_ARG);
    NBP_RETURN_IF(ksize != 1 && ksize != 3, NBP_E_ARG);
    NBP_RETURN_IF(C0 < 64 || C0 % 64 || C1 < 0 || C1 % 64 || N < 32 || N % 32, NBP_E_SHAPE);
    NBP_RETURN_IF(C1 > 0 && (!o.src1 || (o2 && !o2->src1)), NBP_E_ARG);
    NBP_RETURN_IF(ups && ((H | W) & 1), NBP_E_SHAPE);
    IgemmArgsH a;
    a.src0 = o.src0; a.src1 = o.src1; a.C0 = C0; a.C1 = C1; a.cc0 = C0 / 64; a.ups = ups ? 1 : 0;
    a.H = H; a.W = W; a.Hs = ups ? H / 2 : H; a.Ws = ups ? W / 2 : W;
    a.taps = ksize * ksize; a.wpk = o.wpk; a.N = N; a.scale = o.scale; a.shift = o.shift; a.relu = relu;
    a.out = o.out;
    a.groups = groups;
    a.g_src0 = o2 ? o2->src0 : nullptr; a.g_src1 = o2 ? o2->src1 : nullptr; a.g_wpk = o2 ? o2->wpk : nullptr;
    a.g_scale = o2 ? o2->scale : nullptr; a.g_shift = o2 ? o2->shift : nullptr; a.g_out = o2 ? o2->out : nullptr;
    a.M = (long long)B * H * W;
    {
        const long long b0 = (long long)B * a.Hs * a.Ws * C0 * 2, b1 = (long long)B * a.Hs * a.Ws * C1 * 2;
        const long long bw = (long long)(C0 + C1) * a.taps * N * 2;
        NBP_RETURN_IF(b0 >= (1ll << 31) || b1 >= (1ll << 31) || bw >= (1ll << 31), NBP_E_SHAPE);
        a.bytes0 = (unsigned)b0; a.bytes1 = C1 ? (unsigned)b1 : (unsigned)b0; a.bytesw = (unsigned)bw;
    }
    a.chunks_total = (C0 + C1) / 64 * a.taps;
    const bool explicit_up = tile == NBP_TILE_HALO_UP_128 || tile == NBP_TILE_HALO_UP_64;
    const bool have_up = ups && C1 == 0 && (explicit_up || (o.wpk_up && (!o2 || o2->wpk_up)));
    ConvPlan p = nbp_plan_conv_bf16(a.M, N, a.chunks_total, tile, split_k, groups, H, W, ksize, have_up ? 1 : 0);
    TileInfo ti = tile_info_h(p.tile);
    NBP_RETURN_IF(ti.bm == 0 || N % ti.bn, NBP_E_SHAPE);
    const bool ph = p.tile == NBP_TILE_HALO_UP_128 || p.tile == NBP_TILE_HALO_UP_64;
    if (ph) {       // the parity filters: explicit tile id -> they are what wpk points at; automatic -> the operands' wpk_up
        NBP_RETURN_IF(!have_up, NBP_E_SHAPE);
        if (!explicit_up) { a.wpk = o.wpk_up; a.g_wpk = o2 ? o2->wpk_up : nullptr; }
        a.chunks_total = C0 / 64 * 4;
        const long long bwu = (long long)C0 * 16 * N * 2;
        NBP_RETURN_IF(bwu >= (1ll << 31), NBP_E_SHAPE);
        a.bytesw = (unsigned)bwu;
    }
    if (p.tile == NBP_TILE_HALO_128 || p.tile == NBP_TILE_HALO_64)
        NBP_RETURN_IF(!halo_ok(H, W, N, ksize, ti.bn), NBP_E_SHAPE);
    a.split_k = p.split_k; a.chunks_per_split = p.chunks_per_split;
    {
        a.xcd_remap = (a.M / 256) * (N / ti.bn) >= 512 ? 1 : 0;
    }
    a.partial = nullptr;
    if (p.split_k > 1) {
        NBP_RETURN_IF(!ws || ws_bytes < (size_t)groups * p.split_k * a.M * N * sizeof(float), NBP_E_WS);
        a.partial = (float*)ws;
    }
    // epilogue fusions of the halo kernels: only a launch that writes final values of one group can take them
    RowsFuse f{nullptr, nullptr, nullptr, nullptr, nullptr};
    {
        static const int allow_fuse = nbp_tune_int("NBP_BF16_FUSE", 1);
        const bool halo = p.tile == NBP_TILE_HALO_128 || p.tile == NBP_TILE_HALO_64;
        if (allow_fuse && halo && groups == 1 && p.split_k == 1 && !ups && relu && !((H | W) & 1) && pool_out && pool_out[0]) {
            f.pool_out = pool_out[0];
            if (pooled) *pooled = 1;
        }
        if (allow_fuse && !f.pool_out && p.tile == NBP_TILE_HALO_64 && groups == 1 && p.split_k == 1 &&
            N == 64 && relu && head && head->w && head->scale && head->shift && head->out) {
            f.head_w = head->w; f.head_scale = head->scale; f.head_shift = head->shift; f.head_out = head->out;
            if (headed) *headed = 1;
        }
    }
    // the attention gate's tail rides in the 1x1 GEMM over [g | x] when a wave holds all N columns of its pixels
    GatePsiH gp{{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};
    {
        static const int allow_psi = nbp_tune_int("NBP_BF16_PSI", 1);
        const bool whole_n = (p.tile == NBP_TILE_256x64 && N == 64) || (p.tile == NBP_TILE_256x32 && N == 32);
        if (allow_psi && psi && whole_n && ksize == 1 && C1 == C0 && p.split_k == 1 && relu && psi->wpsi[0] && psi->st[0] && psi->gated[0] &&
            (groups == 1 || (psi->wpsi[1] && psi->st[1] && psi->gated[1]))) {
            gp = *psi;
            if (psi_fused) *psi_fused = 1;
        }
    }
    int rc;
    switch (p.tile) {
        case NBP_TILE_128x128: rc = launch_igemm_h<2, 2, 2, 2>(a, st, p.tile); break;
        case NBP_TILE_256x64: rc = launch_igemm_h<4, 1, 2, 2>(a, st, p.tile, gp); break;
        case NBP_TILE_256x32: rc = launch_igemm_h<4, 1, 2, 1>(a, st, p.tile, gp); break;
        case NBP_TILE_128x64: rc = launch_igemm_h<2, 2, 2, 1>(a, st, p.tile); break;
        case NBP_TILE_64x128: rc = launch_igemm_h<1, 4, 2, 1>(a, st, p.tile); break;
        case NBP_TILE_HALO_128: rc = launch_halo<4, 1>(a, f, st, p.tile); break;
        case NBP_TILE_HALO_UP_128: rc = launch_halo<4, 1, true>(a, f, st, p.tile); break;
        case NBP_TILE_HALO_UP_64: rc = launch_halo<2, 2, true>(a, f, st, p.tile); break;
        case NBP_TILE_HALO_64: {
            rc = launch_halo<2, 2>(a, f, st, p.tile);       // (two taps per weight stage; one per stage measured slower, round 2)
            break;
        }
        default: return NBP_E_ARG;
    }
    if (rc) return rc;
    if (p.split_k > 1) {
        long long MN = a.M * N;
        ReduceGroupH g0{o.scale, o.shift, o.out}, g1{a.g_scale, a.g_shift, a.g_out};
        dim3 grid((unsigned)nbp_ew_grid(MN / 4, 256), (unsigned)groups);
        splitk_reduce_bf16_kernel<<<grid, 256, 0, st>>>((const float*)ws, p.split_k, MN, N, g0, g1, relu);
        rc = nbp_launch_status();
    }
    return rc;
}

extern "C" size_t nbp_conv_igemm_bf16_workspace_bytes(int B, int H, int W, int N, int split_k) {
    if (split_k <= 1) return 0;
    return (size_t)split_k * B * H * W * N * sizeof(float);
}

extern "C" int nbp_conv_igemm_bf16(const bf16_t* src0, int C0, const bf16_t* src1, int C1, int ups, int B, int H, int W,
                                   int ksize, const bf16_t* w_packed, int N, const float* scale, const float* shift,
                                   int relu, bf16_t* out, int split_k, int tile, void* ws, size_t ws_bytes,
                                   void* stream) {
    NBP_ENTER();
    ConvOperandsH o{src0, src1, w_packed, scale, shift, out};
    return nbp_conv_igemm_bf16_launch_g(o, nullptr, C0, C1, ups, B, H, W, ksize, N, relu, split_k, tile, ws, ws_bytes,
                                        (hipStream_t)stream);
}

// ------------------------------------------------------------------ weight packing
// dst[((c/64 * taps + tap) * N + n) * 64 + c%64] = bf16(w[n][c][tap] * (scale ? scale[n] : 1))
__global__ void pack_conv_weight_bf16_kernel(const float* __restrict__ w, int N, int C, int taps,
                                             const float* __restrict__ scale, int c_off, bf16_t* __restrict__ dst) {
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
        dst[(((long long)(cg >> 6) * taps + tap) * N + n) * 64 + (cg & 63)] = f2bf(v);
    }
}

// up_conv parity filters in bf16: dst[(((ph * C/64 + c/64) * 4 + tap) * N + n) * 64 + c%64] = bf16(sum in double of the 3x3 taps that
// land on low-resolution tap (r, t) = (tap / 2, tap % 2) of parity ph = py * 2 + px): R(0,0) = {0}, R(0,1) = {1,2}, R(1,0) = {0,1},
// R(1,1) = {2}
__global__ void pack_upconv_weight_bf16_kernel(const float* __restrict__ w, int N, int C, bf16_t* __restrict__ dst) {
    const long long NC = (long long)N * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < NC; i += (long long)gridDim.x * blockDim.x) {
        double v[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) v[k] = (double)w[i * 9 + k];
        const int n = (int)(i / C), c = (int)(i % C);
#pragma unroll
        for (int ph = 0; ph < 4; ++ph)
#pragma unroll
            for (int tap = 0; tap < 4; ++tap) {
                const int py = ph >> 1, px = ph & 1, r = tap >> 1, t = tap & 1;
                const int y_lo = py == 0 ? (r == 0 ? 0 : 1) : (r == 0 ? 0 : 2), y_hi = py == 0 ? (r == 0 ? 0 : 2) : (r == 0 ? 1 : 2);
                const int x_lo = px == 0 ? (t == 0 ? 0 : 1) : (t == 0 ? 0 : 2), x_hi = px == 0 ? (t == 0 ? 0 : 2) : (t == 0 ? 1 : 2);
                double acc = 0.0;
                for (int y = y_lo; y <= y_hi; ++y)
                    for (int x = x_lo; x <= x_hi; ++x) acc += v[y * 3 + x];
                dst[((((long long)ph * (C >> 6) + (c >> 6)) * 4 + tap) * N + n) * 64 + (c & 63)] = f2bf((float)acc);
            }
    }
}

int nbp_pack_upconv_weight_bf16_launch(const float* w_oihw, int N, int C, bf16_t* dst, hipStream_t st) {
    NBP_RETURN_IF(!w_oihw || !dst, NBP_E_ARG);
    NBP_RETURN_IF(N < 1 || C < 64 || C % 64, NBP_E_SHAPE);
    pack_upconv_weight_bf16_kernel<<<nbp_ew_grid((long long)N * C, 256), 256, 0, st>>>(w_oihw, N, C, dst);
    return nbp_launch_status();
}

extern "C" int nbp_pack_upconv_weight_bf16(const float* w_oihw, int N, int C, bf16_t* dst, void* stream) {
    NBP_ENTER();
    return nbp_pack_upconv_weight_bf16_launch(w_oihw, N, C, dst, (hipStream_t)stream);
}

extern "C" int nbp_pack_conv_weight_bf16(const float* w_oihw, int N, int C, int ksize, const float* scale_or_null,
                                         int c_off, int c_total, bf16_t* dst, void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(!w_oihw || !dst, NBP_E_ARG);
    NBP_RETURN_IF(ksize != 1 && ksize != 3, NBP_E_ARG);
    NBP_RETURN_IF(N < 1 || C < 1 || c_off < 0 || c_off + C > c_total || c_total % 64, NBP_E_SHAPE);
    long long total = (long long)N * C * ksize * ksize;
    pack_conv_weight_bf16_kernel<<<nbp_ew_grid(total, 256), 256, 0, (hipStream_t)stream>>>(
        w_oihw, N, C, ksize * ksize, scale_or_null, c_off, dst);
    return nbp_launch_status();
}

// ------------------------------------------------------------------ fp32 <-> bf16
__global__ void f32_to_bf16_kernel(const float* __restrict__ in, long long n, bf16_t* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = f2bf(in[i]);
}
__global__ void bf16_to_f32_kernel(const bf16_t* __restrict__ in, long long n, float* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = bf2f(in[i]);
}
extern "C" int nbp_f32_to_bf16(const float* in, long long n, bf16_t* out, void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(!in || !out || n < 1, NBP_E_ARG);
    f32_to_bf16_kernel<<<nbp_ew_grid(n, 256), 256, 0, (hipStream_t)stream>>>(in, n, out);
    return nbp_launch_status();
}
extern "C" int nbp_bf16_to_f32(const bf16_t* in, long long n, float* out, void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(!in || !out || n < 1, NBP_E_ARG);
    bf16_to_f32_kernel<<<nbp_ew_grid(n, 256), 256, 0, (hipStream_t)stream>>>(in, n, out);
    return nbp_launch_status();
}

// ------------------------------------------------------------------ Conv1.conv.0 (5 -> 64, fp32 NCHW in, bf16 NHWC out)
__global__ __launch_bounds__(256) void conv_first_bf16_kernel(const float* __restrict__ x, int B, int H, int W,
                                                              const float* __restrict__ w, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, bf16_t* __restrict__ out) {
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
    u16x8 o0, o1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        int co = g * 16 + e;
        o0[e] = f2bf(fmaxf(fmaf(acc[e], scale[co], shift[co]), 0.f));
        o1[e] = f2bf(fmaxf(fmaf(acc[8 + e], scale[co + 8], shift[co + 8]), 0.f));
    }
    bf16_t* op = out + m * 64 + g * 16;
    *reinterpret_cast<u16x8*>(op) = o0;
    *reinterpret_cast<u16x8*>(op + 8) = o1;
}

__global__ __launch_bounds__(256) void conv_first_mfma_bf16_kernel(const float* __restrict__ x, int B, int H, int W,
                                                                   const float* __restrict__ w, const float* __restrict__ scale,
                                                                   const float* __restrict__ shift, bf16_t* __restrict__ out) {
    conv_first_mfma_body<bf16_t>(x, B, H, W, w, scale, shift, out, [](bf16_t* p, const f32x4& v) {
        u16x4 o = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
        *reinterpret_cast<u16x4*>(p) = o;
    });
}

int nbp_conv_first_bf16_launch(const float* x_nchw, int B, int H, int W, const float* w_oihw, const float* scale,
                               const float* shift, bf16_t* out_nhwc, hipStream_t st) {
    NBP_RETURN_IF(!x_nchw || !w_oihw || !scale || !shift || !out_nhwc, NBP_E_ARG);
    NBP_RETURN_IF(B < 1 || H < 1 || W < 1, NBP_E_ARG);
    long long M = (long long)B * H * W;
    if ((H & 7) == 0 && (W & 31) == 0) {
        conv_first_mfma_bf16_kernel<<<(unsigned)(M / 256 < 2048 ? M / 256 : 2048), 256, 0, st>>>(x_nchw, B, H, W, w_oihw, scale, shift, out_nhwc);
        return nbp_launch_status();
    }
    conv_first_bf16_kernel<<<(unsigned)nbp_cdiv(M, 64), 256, 0, st>>>(x_nchw, B, H, W, w_oihw, scale, shift, out_nhwc);
    return nbp_launch_status();
}

// ------------------------------------------------------------------ MaxPool2d(2,2), NHWC bf16 (8 channels per lane)
__global__ __launch_bounds__(256) void maxpool2_bf16_kernel(const bf16_t* __restrict__ in, int B, int H, int W, int C8,
                                                            bf16_t* __restrict__ out) {
    const int Ho = H >> 1, Wo = W >> 1;
    const long long total = (long long)B * Ho * Wo * C8;
    const u16x8* in8 = reinterpret_cast<const u16x8*>(in);
    u16x8* out8 = reinterpret_cast<u16x8*>(out);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        int c = (int)(i % C8);
        long long p = i / C8;
        int xo = (int)(p % Wo);
        long long q = p / Wo;
        int yo = (int)(q % Ho);
        int b = (int)(q / Ho);
        long long base = (((long long)b * H + 2 * yo) * W + 2 * xo) * C8 + c;
        u16x8 v0 = in8[base], v1 = in8[base + C8], v2 = in8[base + (long long)W * C8], v3 = in8[base + (long long)W * C8 + C8];
        u16x8 r;
#pragma unroll
        for (int e = 0; e < 8; ++e)
            r[e] = f2bf(fmaxf(fmaxf(bf2f(v0[e]), bf2f(v1[e])), fmaxf(bf2f(v2[e]), bf2f(v3[e]))));
        out8[i] = r;
    }
}

int nbp_maxpool2_bf16_launch(const bf16_t* in, int B, int H, int W, int C, bf16_t* out, hipStream_t st) {
    NBP_RETURN_IF(!in || !out, NBP_E_ARG);
    NBP_RETURN_IF(B < 1 || H < 2 || W < 2 || (H & 1) || (W & 1) || C < 8 || (C & 7), NBP_E_SHAPE);
    long long total = (long long)B * (H / 2) * (W / 2) * (C / 8);
    maxpool2_bf16_kernel<<<nbp_ew_grid(total, 256), 256, 0, st>>>(in, B, H, W, C / 8, out);
    return nbp_launch_status();
}

// ------------------------------------------------------------------ attention gate tail (bf16 q, x, out; fp32 math)
__global__ __launch_bounds__(256) void psi_gate_bf16_kernel(const bf16_t* __restrict__ q, int F8,
                                                            const float* __restrict__ wpsi, const float* __restrict__ st,
                                                            const bf16_t* __restrict__ x, int C8, long long M,
                                                            bf16_t* __restrict__ out) {
    const int sub = threadIdx.x & 15;
    const u16x8* q8 = reinterpret_cast<const u16x8*>(q);
    const u16x8* x8 = reinterpret_cast<const u16x8*>(x);
    u16x8* o8 = reinterpret_cast<u16x8*>(out);
    const float s = st[0], t = st[1];
    const long long ppb = blockDim.x >> 4;
    for (long long m = (long long)blockIdx.x * ppb + (threadIdx.x >> 4); m < M; m += (long long)gridDim.x * ppb) {
        float acc = 0.f;
        for (int f = sub; f < F8; f += 16) {
            const u16x8 a = q8[m * F8 + f];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc = fmaf(bf2f(a[e]), wpsi[f * 8 + e], acc);
        }
        acc += __shfl_xor(acc, 8, 16);
        acc += __shfl_xor(acc, 4, 16);
        acc += __shfl_xor(acc, 2, 16);
        acc += __shfl_xor(acc, 1, 16);
        const float z = acc * s + t;
        const float psi = 1.f / (1.f + expf(-z));
        for (int c = sub; c < C8; c += 16) {
            const u16x8 v = x8[m * C8 + c];
            u16x8 r;
#pragma unroll
            for (int e = 0; e < 8; ++e) r[e] = f2bf(bf2f(v[e]) * psi);
            o8[m * C8 + c] = r;
        }
    }
}

int nbp_psi_gate_bf16_launch(const bf16_t* q, int F, const float* w_psi, const float* s_t2, const bf16_t* x, int C,
                             long long M, bf16_t* out, hipStream_t st) {
    NBP_RETURN_IF(!q || !w_psi || !s_t2 || !x || !out, NBP_E_ARG);
    NBP_RETURN_IF(F < 8 || (F & 7) || C < 8 || (C & 7) || M < 1, NBP_E_SHAPE);
    psi_gate_bf16_kernel<<<nbp_ew_grid(M * 16, 256), 256, 0, st>>>(q, F / 8, w_psi, s_t2, x, C / 8, M, out);
    return nbp_launch_status();
}

// ------------------------------------------------------------------ final 1x1 (bf16 NHWC -> fp32 NCHW, n_out <= 8)
template <int NO>
__global__ __launch_bounds__(256) void final_1x1_bf16_kernel(const bf16_t* __restrict__ in, int B, int H, int W, int C8,
                                                             const float* __restrict__ w, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, int sigmoid,
                                                             float* __restrict__ out) {
    const int sub = threadIdx.x & 15;
    const u16x8* in8 = reinterpret_cast<const u16x8*>(in);
    const long long HW = (long long)H * W, M = (long long)B * HW;
    const long long ppb = blockDim.x >> 4;
    for (long long m = (long long)blockIdx.x * ppb + (threadIdx.x >> 4); m < M; m += (long long)gridDim.x * ppb) {
        float acc[NO];
#pragma unroll
        for (int o = 0; o < NO; ++o) acc[o] = 0.f;
        for (int c = sub; c < C8; c += 16) {
            const u16x8 v = in8[m * C8 + c];
#pragma unroll
            for (int o = 0; o < NO; ++o)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[o] = fmaf(bf2f(v[e]), w[(o * C8 + c) * 8 + e], acc[o]);
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

int nbp_final_1x1_bf16_launch(const bf16_t* in, int B, int H, int W, int C, const float* w_oc, int n_out,
                              const float* scale, const float* shift, int sigmoid, float* out_nchw, hipStream_t st) {
    NBP_RETURN_IF(!in || !w_oc || !scale || !shift || !out_nchw, NBP_E_ARG);
    NBP_RETURN_IF(B < 1 || H < 1 || W < 1 || C < 8 || (C & 7), NBP_E_SHAPE);
    long long M = (long long)B * H * W;
    int grid = nbp_ew_grid(M * 16, 256);
    if (n_out == 8) final_1x1_bf16_kernel<8><<<grid, 256, 0, st>>>(in, B, H, W, C / 8, w_oc, scale, shift, sigmoid, out_nchw);
    else if (n_out == 1) final_1x1_bf16_kernel<1><<<grid, 256, 0, st>>>(in, B, H, W, C / 8, w_oc, scale, shift, sigmoid, out_nchw);
    else return NBP_E_SHAPE;
    return nbp_launch_status();
}
