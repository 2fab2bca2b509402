// nbp_split.hip -- the fp32 3x3 convolutions of NBP on the 16-bit matrix pipe, by operand splitting.
//
// Same operator as conv3x3_halo_f32_kernel (nbp_conv.hip): conv_block / up_conv of next_best_path/networks/nbp_model.py:8-40
// on fp32 NHWC activations with fp32 weights, fp32 accumulation, fp32 epilogue.  The matrix products run on
// v_mfma_f32_32x32x16_f16 (32 cycles for 16 k) instead of v_mfma_f32_32x32x2_f32 (8 x 64 cycles for 16 k):
//   every fp32 operand x is scaled by a power of two s (exact) and cut into two fp16 pieces by round-to-nearest,
//       hi = fp16(s x),   lo = fp16(s x - hi)      (the subtraction is exact; lo keeps 11 of the 13 remaining bits),
//   so s x = hi + lo up to 2^-23 |s x| (unbiased), every fp16 x fp16 product is exact in the fp32 accumulator, and
//       x w = [hi hi + (hi lo + lo hi)] / (s_x s_w)  +  [lo lo <= 2^-22 |x w|, dropped]
//   is THREE MFMAs of 32 cycles (96 cycles per 16 k against 512: 5.3 x the fp32 pipe's rate).
//   s is per TENSOR: 2^(14 - floor(log2 max|x|)), which puts max|x| in [2^14, 2^15) -- no overflow (fp16 max 65504), the full
//   two-piece precision down to 2^-17 max|x|, an absolute representation floor of 2^-40 max|x| below that (SPLIT_EXP below).  max|x| of every
//   activation tensor is produced by the kernel that writes it (one atomicMax per wave in the epilogue) or, for tensors
//   that come from elsewhere, by amax_kernel; max|w| is taken at pack time.  Scales being powers of two, scaling and
//   the final 1 / (s_x s_w) are exact.
//   Measured against fp64 (tools/diag/split_precision.hip, K = 1152 / 9216, normal / half-normal / six-decade inputs): rms
//   error 1.7-1.9e-8 of sum|terms| (4.9e-8 on the six-decade inputs) against 2.8-3.0e-8 (7.6e-8) for the fp32 MFMA chain:
//   the path is at least as accurate as the fp32 pipe and keeps fp32 tensors everywhere outside the MFMA operands.
//   tests/test_gpu_split.py holds both paths against fp64 layer by layer and for the whole network.
// Weights are split once at pack time (two fp16 planes); activations once per workgroup and chunk, on their way from
// global memory into LDS.
#include "common.h"
#include "nbp_internal.h"
#include <cstdlib>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// Operands are scaled so that max |.| lands in [2^SPLIT_EXP, 2^(SPLIT_EXP + 1)): 14 is the largest exponent that keeps the
// tensor's maximum below fp16's 65504.  The headroom is what the SMALL elements of a tensor are worth: an element 2^r below
// the maximum has a normal-fp16 hi piece while r <= SPLIT_EXP + 14 and a normal lo piece (the full two-piece representation,
// 2^-23) while r <= SPLIT_EXP + 2; beyond, lo sits on fp16's subnormal grid (spacing 2^-24 of the scaled value) and the element
// keeps 2^(r - SPLIT_EXP - 25) relative precision (with 12, the first version: rollout tensors span 2^17 between a wall cell and
// the median cell, whose values then carried 2^-20 instead of 2^-23; tests/test_gpu_split.py::test_split_in_tensor_dynamic_range_floor).  The bounds handed in are
// always >= the true maximum (atomicMax of the values written, or an inherited upper bound), so nothing overflows.
constexpr int SPLIT_EXP = 14;
// floor(log2 m) of a positive float given by its bits; SPLIT_EXP for zero (scale 1); clamped so that 2^(SPLIT_EXP - e) is a normal float
__host__ __device__ inline int amax_exponent(unsigned bits) {
    if (!bits) return SPLIT_EXP;
    int e = (int)((bits >> 23) & 255u) - 127;
    return e < -100 ? -100 : (e > 100 ? 100 : e);
}
__host__ __device__ inline float pow2f(int e) {
    union { unsigned u; float f; } v;
    v.u = (unsigned)(127 + e) << 23;
    return v.f;
}

// two scaled fp32 -> packed (low element first) fp16 pairs of their hi / lo pieces
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& h, unsigned& l) {
    const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;
    const _Float16 l0 = (_Float16)(x0 - (float)h0), l1 = (_Float16)(x1 - (float)h1);
    h = __builtin_bit_cast(unsigned, f16x2{h0, h1});
    l = __builtin_bit_cast(unsigned, f16x2{l0, l1});
}

// max |x| of a tensor lives in AMAX_WORDS words (float bits; the tensor's max is the max over them): same-address device
// atomics run at ~0.3 G/s on this part (16 k waves finishing together = 55 us), so writers spread over the words by
// workgroup id and readers take the max of all of them with one 256-B load per wave.
constexpr int AMAX_WORDS = 64;
#define NBP_SPLIT_MAX_K_DEFAULT 2304
#define NBP_SPLIT_MAX_K_SMALL_DEFAULT 1152
__device__ __forceinline__ void wave_amax(float mx, unsigned* out) {
#pragma unroll
    for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    // non-negative floats order like their bits
    if ((threadIdx.x & 63) == 0)
        atomicMax(out + ((blockIdx.x * 4u + (threadIdx.x >> 6) + blockIdx.y * 17u + blockIdx.z * 29u) & (AMAX_WORDS - 1)), __float_as_uint(mx));
}
__device__ __forceinline__ void block_amax(float mx, unsigned* out, unsigned words = AMAX_WORDS) {
    __shared__ float part[4];
#pragma unroll
    for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicMax(out + (blockIdx.x & (words - 1)), __float_as_uint(fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]))));
}
__device__ __forceinline__ unsigned read_amax(const unsigned* slot) {     // every lane returns the max over the words
    unsigned v = slot[threadIdx.x & (AMAX_WORDS - 1)];
#pragma unroll
    for (int o = 32; o; o >>= 1) { const unsigned u = __shfl_xor(v, o); v = u > v ? u : v; }
    return v;
}

struct SplitOps {
    const float* src0; const float* src1; const void* planes; const float* scale; const float* shift; float* out;
    const unsigned* amax0; const unsigned* amax1; const unsigned* wamax; unsigned* amax_out;
    float* pool_out;            // not null: the 2x2 max-pool of the output (nbp_model.py:113-123) is written as well [B,H/2,W/2,N]
    // not null (N == 64 == the tile's columns): the layer feeds only the one-channel sigmoid head (Final2, nbp_model.py:108,
    // :158-159): head_out[m] = sigmoid((out[m, :] . head_w) * head_ss[0] + head_ss[1]) is written INSTEAD of out
    const float* head_w; const float* head_ss[2]; float* head_out;
    // not null (training, kernels instantiated with BS): every workgroup writes the column sums of the values it stores and of
    // their squares, in double, as row (pixel tile [x 4 + parity]) of bn_part[rows][2][N] -- the BatchNorm behind the
    // convolution finalises these instead of reading the tensor once more (nbp_train.hip: colsum_pair)
    double* bn_part;
    // not null (plain 3x3 kernel): source 0 is src0[pixel][c] * psi0[pixel] -- the attention gate's x * psi formed while the halo is
    // staged (the products are the gate kernel's own fp32 products, so the planes are what they were bit for bit)
    const float* psi0;
};
struct SplitArgs {
    SplitOps g[2];              // blockIdx.z >= split_k: the second problem of a grouped launch
    int C0, C1, ups, H, W, Hs, Ws, N, relu;
    long long M;
    int split_k, chunks_total, chunks_per_split;       // in 16-channel chunks
    unsigned bytes0, bytes1, bytesw;
    float* partial;             // split-K scratch [group][split][M][N]
    int groups, xcd_remap;
};

// ------------------------------------------------------------------ 3x3 convolution from LDS-resident fp16 planes
// Workgroup = 16 x TW pixels x 32 TN output channels, 4 waves x (TM x 32 pixels) x (TN x 32 channels) = 8 accumulator tiles
// per wave: <TW = 32, TM = 4, TN = 2> 16 x 32 pixels x 64 channels, <16, 2, 4> 16 x 16 pixels x 128 channels (the
// 16-pixel-wide levels).  The halo tile of a 16-channel chunk goes global -> registers -> (scale, split once) -> two fp16
// planes in LDS, and the tap loop is ds_read_b128 + MFMA only:
//   planes  [hi|lo][k half][halo pixels][8 fp16]   (a lane's fragment = 16 contiguous bytes; any 16-lane group of a
//                                                   ds_read_b128 covers 16 distinct slots when TW = 32: no conflicts)
//   weights [tap of a filter row][hi|lo][k half][32 TN rows][8 fp16], one filter row (3 taps) per stage, two stages
// 64 / 70 KB of LDS: two workgroups per CU.  One MFMA K step (16 channels) per tap; a stage is 3 taps = 72 MFMAs per wave
// between barriers, with the next stage's weights DMA'd (buffer_load ... lds) behind it.  The next chunk's halo loads are
// issued before the chunk's first stage and consumed (split + ds_write) after its last one.
// PH (the x2-nearest-upsample + 3x3 convolution of up_conv, nbp_model.py:25-33): the four output parities (y & 1, x & 1) are four
// 2x2 convolutions of the LOW-resolution input with pre-summed weights -- rows {v - 1 + py, v + py}, filter rows {W[-1], W[0] + W[1]}
// for py = 0 and {W[-1] + W[0], W[1]} for py = 1, columns alike -- 16 tap-products per low-resolution pixel instead of 36.
// A workgroup owns one parity of a 16 x TW low-resolution tile (blockIdx.z & 3), runs the 2 x 2 taps from the same halo planes
// (stage = one row of 2 taps) and writes its outputs to (2 v + py, 2 u + px).
#ifdef NBP_DBG_TS
// debug build only (tools/diag/conv_timeline.py): time stamps of one wave at the phase boundaries of the stage loop
__device__ unsigned long long nbp_dbg_ts[8192];
__device__ unsigned nbp_dbg_n;
#define NBP_WALL(tag) do { if (dbg_on) { const unsigned k_ = dbg_i++; if (k_ < 4000) { nbp_dbg_ts[2 * k_] = (unsigned long long)(tag); nbp_dbg_ts[2 * k_ + 1] = wall_clock64(); } } } while (0)
#define NBP_TS(tag) do { if (dbg_on) { const unsigned k_ = dbg_i++; if (k_ < 4000) { nbp_dbg_ts[2 * k_] = (unsigned long long)(tag); nbp_dbg_ts[2 * k_ + 1] = __builtin_readcyclecounter(); } } } while (0)
#else
#define NBP_TS(tag) do {} while (0)
#endif
// P2 (PH only): a workgroup owns BOTH column parities (px = 0, 1) of its row parity over a tile of half the height -- the same
// accumulator count (2 x TM x TN), the same MFMAs per stage, but ONE staged halo serves two parities' tap products and the three
// halo columns the four (px, tap) pairs touch are read from LDS three times instead of four: the staging work per MFMA of the
// one-parity form was 2.3 x the plain kernel's (4 taps per staged chunk instead of 9), this form's is 1.25 x.
// DG (PH only; training): the DATA GRADIENT of an up_conv layer in the same parity form.  With the output gradient seen as four
// parity planes D[py, px][v, u] = dout[2 v + py, 2 u + px] of the low-resolution grid,
//   din[y, x, c] = sum over (py, px), (r, t), n of Wc[py, px][r][t][n][c] D[py, px][y + 1 - py - r, x + 1 - px - t, n]:
// a convolution over K = 4 parities x N channels with 2 x 2 taps per chunk -- 16 tap-products per low-resolution pixel where the
// full-resolution 3 x 3 convolution of dout followed by the 2 x 2 sum (round 4) does 36.  A 16-channel chunk belongs to ONE parity,
// which sets its halo gather (the parity plane, read straight from dout: + py W + px pixels) and the origin of its 2 x 2 taps
// (1 - py, 1 - px); the filter-row flip r' = 1 - r is baked into the packed planes (pack_upconv_dgrad_h2_kernel).  One workgroup per
// low-resolution tile, plain output (no parity scatter, no 2 x 2 sum pass).  a.H / a.W = the low-resolution output, a.Hs / a.Ws = dout.
template <int TW, int TM, int TN, bool PH, bool BS = false, bool P2 = false, bool DG = false>
__global__ __launch_bounds__(256, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_halo_h2_kernel(SplitArgs a) {
    static_assert(!P2 || PH, "two-parity form: up_conv layers");
    static_assert(!DG || (PH && !BS && !P2), "data gradient of an up_conv layer: the one-parity tap structure");
    constexpr bool PHO = PH && !DG;                            // the launch writes one output parity of the full-resolution image
    int zs = blockIdx.z;
    const int py = PHO ? (P2 ? zs & 1 : (zs >> 1) & 1) : 0, px = (PHO && !P2) ? zs & 1 : 0;
    if (PHO) zs >>= (P2 ? 1 : 2);
    const int zslice = zs;                                     // partial-sum slice: group * split_k + split
    const SplitOps& o = zs >= a.split_k ? a.g[1] : a.g[0];
    if (zs >= a.split_k) zs -= a.split_k;
    constexpr int ROWS = PH ? 2 : 3, TPR = PH ? 2 : 3, TAPS = ROWS * TPR;      // filter rows = stages per chunk; taps per row
    constexpr int BN = TN * 32, NB = BN / 64;
    constexpr int RPB = 32 / TW, TH = 4 * TM * RPB;            // image rows per 32-pixel row block; tile height (16)
    constexpr int HW_ = TW + 2, HPIX = (TH + 2) * HW_;         // 18 x 34 = 612 / 18 x 18 = 324 halo pixels
    constexpr int RS = (HPIX + 7) / 8 * 8 * 16 + 64;           // bytes between (plane, k half) regions (+64: ds_write banks)
    constexpr int HALO_BYTES = 4 * RS;
    constexpr int WI = TPR * 4 * NB;                           // weight DMA instructions (64 rows x 16 B) per stage (and parity)
    constexpr int NPAR = P2 ? 2 : 1;                           // column parities a workgroup computes
    constexpr int WB1 = WI * 1024, WB = NPAR * WB1;
    constexpr int NF = (HPIX * 4 + 255) / 256;                 // float4 pieces of the halo tile per thread
    static_assert(BN % 64 == 0 && (TH == 16 || TH == 8), "tile shape");
    extern __shared__ __attribute__((aligned(16))) char ldsb[];
    char* const halo = ldsb;
    char* const wbuf = ldsb + HALO_BYTES;
    float* const psil = reinterpret_cast<float*>(ldsb + HALO_BYTES + 2 * WB);      // (plain kernel: psi of the tile's halo pixels)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef NBP_DBG_TS
    const bool dbg_on = blockIdx.x == gridDim.x / 2 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 64;
    unsigned dbg_i = 0;
    NBP_WALL(100); NBP_TS(101);
#endif
    const int Ht = PHO ? a.Hs : a.H, Wt = PHO ? a.Ws : a.W;   // the tile grid: output pixels, or low-resolution pixels for PH
    const int tiles_x = Wt / TW, tiles_y = Ht / TH;
    unsigned tile = blockIdx.x, nt = blockIdx.y;
    // Workgroups go round-robin to the 8 XCDs (each with its own L2) in linear-id order.  xcd_remap 1: an XCD takes a contiguous
    // run of (pixel tile, channel block) pairs, channel block fastest -- activations cross the fabric once, every XCD streams
    // all the weights.  xcd_remap 2 (weight-heavy layers): an XCD (or a group of 8 / gridDim.y XCDs) owns channel blocks, so the
    // weights cross once and the activations once per owner.
    if (a.xcd_remap == 1) {
        const unsigned L = blockIdx.x + gridDim.x * blockIdx.y, T = gridDim.x * gridDim.y;
        const unsigned xcd = L & 7u, idx = L >> 3, q = T >> 3, r = T & 7u;
        const unsigned v = xcd * q + min(xcd, r) + idx;
        nt = v % gridDim.y;
        tile = v / gridDim.y;
    } else if (a.xcd_remap == 2) {
        const unsigned L = blockIdx.x + gridDim.x * blockIdx.y;
        const unsigned xcd = L & 7u, idx = L >> 3, nbk = gridDim.y;
        if (nbk >= 8u) {                    // nbk % 8 == 0 (launcher)
            const unsigned per = nbk >> 3;
            nt = xcd * per + idx % per;
            tile = idx / per;
        } else {                            // 8 % nbk == 0 and gridDim.x % (8 / nbk) == 0 (launcher)
            const unsigned g = 8u / nbk;
            nt = xcd / g;
            tile = idx * g + xcd % g;
        }
    }
    const unsigned tile_lin = tile;                           // (BS: the row of the workgroup's BatchNorm partials)
    const int tx = tile % tiles_x; tile /= tiles_x;
    const int ty = tile % tiles_y;
    const int b = tile / tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;
    const int n0 = nt * BN;

    // operand scales (exact powers of two) from the tensors' max |.|
    const unsigned ma0 = read_amax(o.amax0), ma1 = o.amax1 ? read_amax(o.amax1) : 0u;
    const int ea = amax_exponent(ma0 > ma1 ? ma0 : ma1), ew = amax_exponent(*o.wamax);
    const float sa = pow2f(SPLIT_EXP - ea);
    const int einv = ea + ew - 2 * SPLIT_EXP;                  // 1 / (s_x s_w) = 2^einv, applied with v_ldexp_f32 (any exponent)

    // halo staging: piece f = tid + 256 k is channels 4 (f & 3) .. + 3 of halo pixel f >> 2
    // (the LDS destination of piece k is ((f & 3) >> 1) RS + (f >> 2) 16 + (f & 1) 8: recomputed at the store, not kept)
    int hpix[NF];
#pragma unroll
    for (int k = 0; k < NF; ++k) {
        const int f = tid + 256 * k;
        const int hr = f >> 2;
        const int hy = hr / HW_, hx = hr - hy * HW_;
        const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
        const bool ok = hr < HPIX && (unsigned)yy < (unsigned)Ht && (unsigned)xx < (unsigned)Wt;
        hpix[k] = ok ? (DG ? (b * a.Hs + 2 * yy) * a.Ws + 2 * xx          // (the parity's own (py, px) is added per chunk)
                          : PH ? (b * a.Hs + yy) * a.Ws + xx : (b * a.Hs + (yy >> a.ups)) * a.Ws + (xx >> a.ups)) : -1;
    }
    const int hdst0 = ((tid & 3) >> 1) * RS + (tid >> 2) * 16 + (tid & 1) * 8;      // piece k: + k * 64 pixels * 16 B
    const __amdgpu_buffer_rsrc_t rs0 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(o.src0), 0, a.bytes0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(o.src1 ? o.src1 : o.src0), 0, a.bytes1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(o.planes), 0, a.bytesw, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    const int c16_0 = a.C0 >> 4;

    // psi of this tile's halo pixels (0 outside the image, where the loads return zeros anyway), once per workgroup
    bool psi_on = false;
    if constexpr (!PH) {
        psi_on = o.psi0 != nullptr;
        if (psi_on) {
#pragma unroll
            for (int k = 0; k < NF; ++k) {
                const int hr = (tid >> 2) + 64 * k;
                if ((tid & 3) == 0 && hr < HPIX) psil[hr] = hpix[k] >= 0 ? o.psi0[hpix[k]] : 0.f;
            }
            __syncthreads();
        }
    }

    u32x4 hreg[NF];
    auto load_halo = [&](int c) {       // 16-channel chunk c of the concatenated input
        if constexpr (DG) {             // chunk = (parity q, 16 channels of dout): the parity plane of the halo tile
            const int q = c / c16_0, cbase = (c - q * c16_0) * 16 + (tid & 3) * 4;
            const int shift = (q >> 1) * a.Ws + (q & 1);
#pragma unroll
            for (int k = 0; k < NF; ++k) {
                const unsigned off = hpix[k] >= 0 ? (unsigned)((hpix[k] + shift) * a.C0 + cbase) * 4u : OOB;
                hreg[k] = __builtin_amdgcn_raw_buffer_load_b128(rs0, off, 0, 0);
            }
            return;
        }
        const bool first = c < c16_0;
        const int Cs = first ? a.C0 : a.C1;
        const int cbase = (first ? c : c - c16_0) * 16 + (tid & 3) * 4;
#pragma unroll
        for (int k = 0; k < NF; ++k) {
            const unsigned off = hpix[k] >= 0 ? (unsigned)(hpix[k] * Cs + cbase) * 4u : OOB;
            hreg[k] = first ? __builtin_amdgcn_raw_buffer_load_b128(rs0, off, 0, 0) : __builtin_amdgcn_raw_buffer_load_b128(rs1, off, 0, 0);
        }
    };
    auto store_halo = [&](int c) {      // (c: the chunk hreg holds)
        const bool gated = psi_on && c < c16_0;           // a chunk of source 0 behind an attention gate: x * psi
#pragma unroll
        for (int k = 0; k < NF; ++k) {
            if ((tid >> 2) + 64 * k >= HPIX) continue;
            f32x4 v = __builtin_bit_cast(f32x4, hreg[k]);
            if constexpr (!PH) { if (gated) v = v * psil[(tid >> 2) + 64 * k]; }
            unsigned h0, l0, h1, l1;
            split_pair(v[0] * sa, v[1] * sa, h0, l0);
            split_pair(v[2] * sa, v[3] * sa, h1, l1);
            *reinterpret_cast<u32x2*>(halo + hdst0 + k * 1024) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(halo + 2 * RS + hdst0 + k * 1024) = u32x2{l0, l1};
        }
    };
    // weight stage u = chunk * ROWS + filter row: WI DMA instructions of 64 rows x 16 B; wave w issues q = w, w + 4, ...
    // q = (tap in row * 4 + plane * 2 + k half) * NB + 64-row block; packed planes are [chunk][tap][plane][k half][N][8 fp16]
    auto issue_w = [&](int u) {
        const int c = u / ROWS, row = u - ROWS * c;
        char* dst = wbuf + (u & 1) * WB;
        // PH: the four parities' planes follow each other, each [chunk][4 taps][plane][k half][N][8]
#pragma unroll
        for (int pq = 0; pq < NPAR; ++pq) {
            const long long pbase = PHO ? (long long)(py * 2 + (P2 ? pq : px)) * a.chunks_total * TAPS * 4 * a.N : 0;
#pragma unroll
            for (int k = 0; k < WI / 4; ++k) {
                const int q = wave + 4 * k;
                const int tt = q / (4 * NB), r = q - tt * (4 * NB);
                const int r4 = r / NB, nb = r - r4 * NB;
                const unsigned woff = (unsigned)((pbase + (((long long)c * TAPS + row * TPR + tt) * 4 + r4) * a.N + n0 + nb * 64 + lane) * 16);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr_t)(dst + pq * WB1 + q * 1024), 16, woff, 0, 0, 0);
            }
        }
    };

    f32x16 accs[NPAR][TM][TN];
    f32x16 (&acc)[TM][TN] = accs[0];
#pragma unroll
    for (int pq = 0; pq < NPAR; ++pq)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) accs[pq][i][j][r] = 0.f;

    const int khalf = lane >> 5;
    const int hbase = (TM * wave * RPB + ((lane & 31) / TW)) * HW_ + ((lane & 31) % TW);
    const char* const arow = halo + khalf * RS + hbase * 16;               // + plane * 2 RS + halo-row shift * 16
    const int brow = khalf * NB * 1024 + (lane & 31) * 16;                 // + ((tap * 4 + plane * 2) * NB + j / 2) KB + (j & 1) * 512

    const int c_begin = zs * a.chunks_per_split;
    const int c_end = min(c_begin + a.chunks_per_split, a.chunks_total);
    if (c_begin < c_end) {
        load_halo(c_begin);
        issue_w(c_begin * ROWS);
        store_halo(c_begin);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __builtin_amdgcn_s_setprio(2);
    for (int c = c_begin; c < c_end; ++c) {
        const bool more = c + 1 < c_end;
        if (more) load_halo(c + 1);
#pragma unroll 1
        for (int row = 0; row < ROWS; ++row) {
            const int u = c * ROWS + row;
            NBP_TS(1);
            if (row < ROWS - 1 || more) issue_w(u + 1);
            const char* Bt = wbuf + (u & 1) * WB + brow;
            if constexpr (P2) {
                // halo column 0: (px 0, tap 0); column 1: (px 0, tap 1) and (px 1, tap 0); column 2: (px 1, tap 1)
                constexpr int PX[3] = {1, 0, 0};
                constexpr int PW[3] = {0, 1, 0};
#pragma unroll
                for (int col = 0; col < 3; ++col) {
                    u32x4 xp[TM][2];
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int p = 0; p < 2; ++p)
                            xp[i][p] = *reinterpret_cast<const u32x4*>(arow + p * (2 * RS) + ((row + py + i * RPB) * HW_ + col) * 16);
#pragma unroll
                    for (int pq = 0; pq < 2; ++pq) {
                        const int tt = col - pq;
                        if (tt < 0 || tt > 1) continue;
                        u32x4 wp[TN][2];
#pragma unroll
                        for (int j = 0; j < TN; ++j)
#pragma unroll
                            for (int p = 0; p < 2; ++p)
                                wp[j][p] = *reinterpret_cast<const u32x4*>(Bt + pq * WB1 + ((tt * 4 + p * 2) * NB + (j >> 1)) * 1024 + (j & 1) * 512);
#pragma unroll
                        for (int v = 0; v < 3; ++v)
#pragma unroll
                            for (int j = 0; j < TN; ++j)
#pragma unroll
                                for (int i = 0; i < TM; ++i)
                                    accs[pq][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, xp[i][PX[v]]),
                                                                                            __builtin_bit_cast(f16x8, wp[j][PW[v]]), accs[pq][i][j], 0, 0, 0);
                    }
                }
            } else {
            // DG: the chunk's parity sets the origin of its 2 x 2 taps (uniform over the workgroup)
            const int pyc = DG ? 1 - ((c / c16_0) >> 1) : py, pxc = DG ? 1 - ((c / c16_0) & 1) : px;
#pragma unroll
            for (int tt = 0; tt < TPR; ++tt) {
                u32x4 xp[TM][2], wp[TN][2];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int p = 0; p < 2; ++p)
                        xp[i][p] = *reinterpret_cast<const u32x4*>(arow + p * (2 * RS) + ((row + pyc + i * RPB) * HW_ + tt + pxc) * 16);
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int p = 0; p < 2; ++p)
                        wp[j][p] = *reinterpret_cast<const u32x4*>(Bt + ((tt * 4 + p * 2) * NB + (j >> 1)) * 1024 + (j & 1) * 512);
                // small terms first: (lo, hi) (hi, lo), then (hi, hi)
                constexpr int PX[3] = {1, 0, 0};
                constexpr int PW[3] = {0, 1, 0};
#pragma unroll
                for (int v = 0; v < 3; ++v)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int i = 0; i < TM; ++i)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, xp[i][PX[v]]),
                                                                               __builtin_bit_cast(f16x8, wp[j][PW[v]]), acc[i][j], 0, 0, 0);
            }
            }
            NBP_TS(2);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            NBP_TS(3);
            __syncthreads();
            NBP_TS(4);
        }
        if (more) {                     // every wave is past its last read of this chunk's planes
            __builtin_amdgcn_s_setprio(0);      // the staging pass yields issue slots to the co-resident workgroup's MFMAs (1 %)
            store_halo(c + 1);
            NBP_TS(5);
            __syncthreads();
            NBP_TS(6);
            __builtin_amdgcn_s_setprio(2);
        }
    }

    // ---- epilogue (A = pixels, B = weights): col n = lane & 31, pixel of the row block = (r&3) + 8 (r>>2) + 4 (lane>>5)
    const bool final_out = (a.split_k == 1);
    float* outp = final_out ? o.out : a.partial + (long long)zslice * a.M * a.N;
    float mx = 0.f;
    constexpr bool HEADABLE = !PH && TN == 2;              // 64 columns = all channels of the head's input in one wave
    const bool head = HEADABLE && final_out && o.head_out;
    const bool bn_stats = BS && final_out && o.bn_part;
    double* const bsh = reinterpret_cast<double*>(ldsb);       // [wave][BN columns][sum | sum of squares]: the stage buffers are dead
    if constexpr (BS) { if (bn_stats) __syncthreads(); }       // (every wave has left the last stage's LDS reads)
    double bs1[BS ? TN : 1], bs2[BS ? TN : 1];                 // BS: this lane's column sums over its pixels (both parities of P2)
#pragma unroll
    for (int j = 0; j < (BS ? TN : 1); ++j) { bs1[j] = 0.0; bs2[j] = 0.0; }
    float hp[HEADABLE ? TM : 1][16];
    if constexpr (HEADABLE) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) hp[i][r] = 0.f;
    }
#pragma unroll
    for (int pq = 0; pq < NPAR; ++pq)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int pxe = P2 ? pq : px;                          // the column parity these accumulators belong to
        const int n = n0 + j * 32 + (lane & 31);
        float sc = 1.f, sh = 0.f;
        if (final_out) { sc = o.scale[n]; sh = o.shift[n]; }
        const float hw = head ? o.head_w[n] : 0.f;
        float vals[TM][16];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const long long mrow = PHO ? ((long long)b * a.H + 2 * (y0 + (TM * wave + i) * RPB) + py) * a.W + 2 * x0 + pxe
                                       : ((long long)b * a.H + y0 + (TM * wave + i) * RPB) * a.W + x0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pb = (r & 3) + 8 * (r >> 2) + 4 * khalf;
                const int poff = PHO ? 2 * ((pb / TW) * a.W + (pb % TW)) : (pb / TW) * a.W + (pb % TW);
                float v = ldexpf(accs[pq][i][j][r], einv) * sc + sh;
                if (final_out && a.relu) v = fmaxf(v, 0.f);
                mx = fmaxf(mx, fabsf(v));
                if (!head) outp[(mrow + poff) * a.N + n] = v;
                vals[i][r] = v;
                if constexpr (HEADABLE) hp[i][r] = fmaf(v, hw, hp[i][r]);
            }
        }
        if constexpr (BS) {
            if (bn_stats) {             // this lane's column over its TM x 16 pixels, in double
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { const double d = (double)vals[i][r]; bs1[j] += d; bs2[j] = fma(d, d, bs2[j]); }
            }
        }
        // 2x2 max-pool of the same values: the four pixels of a window are registers of ONE lane (a wave's row blocks are
        // consecutive image rows; registers r, r + 1 are neighbouring columns; with 16-pixel rows r + 8 is the row below)
        if (!PH && final_out && o.pool_out) {
            const int Hp = a.H >> 1, Wp = a.W >> 1;
            if constexpr (TW == 32) {
#pragma unroll
                for (int i = 0; i < TM; i += 2) {
                    const long long prow = ((long long)b * Hp + ((y0 + TM * wave + i) >> 1)) * Wp + (x0 >> 1);
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const int xo = ((r & 3) + 8 * (r >> 2) + 4 * khalf) >> 1;
                        o.pool_out[(prow + xo) * a.N + n] = fmaxf(fmaxf(vals[i][r], vals[i][r + 1]), fmaxf(vals[i + 1][r], vals[i + 1][r + 1]));
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const long long prow = ((long long)b * Hp + ((y0 + (TM * wave + i) * RPB) >> 1)) * Wp + (x0 >> 1);
#pragma unroll
                    for (int r = 0; r < 8; r += 2) {
                        const int xo = ((r & 3) + 8 * (r >> 2) + 4 * khalf) >> 1;
                        o.pool_out[(prow + xo) * a.N + n] = fmaxf(fmaxf(vals[i][r], vals[i][r + 1]), fmaxf(vals[i][r + 8], vals[i][r + 9]));
                    }
                }
            }
        }
    }
    if constexpr (HEADABLE) {
        if (head) {
            // per row block: reduce-scatter of the 16 per-lane partial dot products over the 32 lanes of the half wave (as in the
            // gate kernel's psi tail): lane l ends with pixel register r = (l & 31) >> 1
            const float hs = o.head_ss[0][0], ht = o.head_ss[1][0];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                float p8[8], p4[4], p2[2];
                {
                    const bool up = lane & 16;
#pragma unroll
                    for (int k = 0; k < 8; ++k) p8[k] = (up ? hp[i][8 + k] : hp[i][k]) + __shfl_xor(up ? hp[i][k] : hp[i][8 + k], 16);
                }
                {
                    const bool up = lane & 8;
#pragma unroll
                    for (int k = 0; k < 4; ++k) p4[k] = (up ? p8[4 + k] : p8[k]) + __shfl_xor(up ? p8[k] : p8[4 + k], 8);
                }
                {
                    const bool up = lane & 4;
#pragma unroll
                    for (int k = 0; k < 2; ++k) p2[k] = (up ? p4[2 + k] : p4[k]) + __shfl_xor(up ? p4[k] : p4[2 + k], 4);
                }
                const bool up = lane & 2;
                float dot = (up ? p2[1] : p2[0]) + __shfl_xor(up ? p2[0] : p2[1], 2);
                dot += __shfl_xor(dot, 1);
                const int r = (lane & 31) >> 1;
                const int pb = (r & 3) + 8 * (r >> 2) + 4 * khalf;
                const long long mrow = ((long long)b * a.H + y0 + (TM * wave + i) * RPB) * a.W + x0;
                if (!(lane & 1)) o.head_out[mrow + (pb / TW) * a.W + (pb % TW)] = 1.f / (1.f + expf(-(dot * hs + ht)));
            }
        }
    }
    if constexpr (BS) {
        if (bn_stats) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {                     // the other half wave's pixels by one exchange
                const double s1 = bs1[j] + __shfl_xor(bs1[j], 32), s2 = bs2[j] + __shfl_xor(bs2[j], 32);
                if (lane < 32) { bsh[(wave * BN + j * 32 + lane) * 2] = s1; bsh[(wave * BN + j * 32 + lane) * 2 + 1] = s2; }
            }
            __syncthreads();
            if (tid < BN) {                                    // the four waves' pixel rows in a fixed order
                double s1 = 0.0, s2 = 0.0;
#pragma unroll
                for (int w = 0; w < 4; ++w) { s1 += bsh[(w * BN + tid) * 2]; s2 += bsh[(w * BN + tid) * 2 + 1]; }
                const long long row = P2 ? (long long)tile_lin * 2 + py : PH ? (long long)tile_lin * 4 + (py * 2 + px) : (long long)tile_lin;
                o.bn_part[(row * 2) * a.N + n0 + tid] = s1;
                o.bn_part[(row * 2 + 1) * a.N + n0 + tid] = s2;
            }
        }
    }
    if (final_out && o.amax_out) wave_amax(mx, o.amax_out);
#ifdef NBP_DBG_TS
    NBP_TS(7);
    NBP_TS(102); NBP_WALL(103);
    if (dbg_on) nbp_dbg_n = dbg_i;
#endif
}

#ifdef NBP_DBG_TS
extern "C" int nbp_dbg_read(unsigned long long* host, unsigned* n) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(n, HIP_SYMBOL(nbp_dbg_n), sizeof(unsigned));
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(nbp_dbg_ts), sizeof(unsigned long long) * 8192);
}
#endif

// ------------------------------------------------------------------ 1x1 convolution over K = [src0 | src1] (the attention gates)
// q = relu([g | x] W + b), next_best_path/networks/nbp_model.py:44-53 (W_g and W_x as one GEMM, BN scales folded into W).
// These layers are HBM-bound (K = 2 co is small), so the activations are NOT staged in LDS: a workgroup owns 128 pixels x BN
// (<= 128) channels, a wave 32 pixels x BN; each lane loads the 8 + 8 channels of its pixel for the two K steps of a 32-channel
// stage straight from global memory (the two k halves of a pixel are adjacent lanes' 32-byte pieces: whole 128-B lines),
// scales and splits them in registers, and the weights of the stage ([k step][hi|lo][k half][BN rows][8 fp16]) stream through a
// double buffer in LDS by DMA.  Three exact fp16 MFMAs per product as in the 3x3 kernel.
struct GateArgs {
    SplitOps g[2];
    int C, N, relu;             // channels of source 0 (and of source 1 when there is one); output channels (N % 32 == 0)
    int C1;                     // channels of source 1: C (the gates' K = [g | x]) or 0 (a plain 1x1 convolution: training's W_g / W_x
                                // layers and their data gradients, where a BatchNorm with batch statistics sits between the two halves)
    long long M;
    unsigned bytes0, bytesw;
    int groups;
    // PSI form (a workgroup holds all N columns of its pixels): the gate's tail runs in the epilogue -- psi = sigmoid((q . wpsi)
    // * st[0] + st[1]) (nbp_model.py:54-58), gated = src1 * psi (ref :60) -- and q itself is never written
    const float* wpsi[2];
    const float* st[2];
    float* gated[2];
    unsigned* gated_amax[2];    // 64-word slots (or null) that receive max |gated|: psi can be << 1, the bound max |x| is loose
    int psi_only;               // PSI form: psi [M] goes to the head of gated[] and the products x * psi are only measured (gated_amax), not written
};

template <int TN, bool PSI>
__global__ __launch_bounds__(256, 2) void gate1x1_h2_kernel(GateArgs a) {
    const SplitOps& o = a.g[blockIdx.z];
    constexpr int BN = TN * 32;
    constexpr int WST = 8 * BN * 16;                           // bytes of one weight stage: 2 k steps x 2 planes x 2 k halves x BN rows
    constexpr int WI = WST / 1024;                             // DMA instructions per stage (64 rows x 16 B each)
    extern __shared__ __attribute__((aligned(16))) char wbuf[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long m = (long long)blockIdx.x * 128 + wave * 32 + (lane & 31);
    const int n0 = blockIdx.y * BN, khalf = lane >> 5;
    const unsigned ma0 = read_amax(o.amax0), ma1 = o.amax1 ? read_amax(o.amax1) : 0u;
    const int ea = amax_exponent(ma0 > ma1 ? ma0 : ma1), ew = amax_exponent(*o.wamax);
    const float sa = pow2f(SPLIT_EXP - ea);
    const int einv = ea + ew - 2 * SPLIT_EXP;
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(o.src0), 0, a.bytes0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(o.src1 ? o.src1 : o.src0), 0, a.bytes0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(o.planes), 0, a.bytesw, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    const int stages0 = a.C >> 5, stages = stages0 + (a.C1 >> 5);      // 32-channel stages: source 0 then source 1
    const unsigned prow = m < a.M ? (unsigned)(m * a.C) * 4u : OOB;

    // The pixel's channels are prefetched three stages ahead (a ring of four register sets): a stage is only ~0.3 us of MFMAs,
    // a load from HBM 1-2 us.  The weights (L2-resident, shared by every workgroup) are DMA'd one stage ahead.
    constexpr int D = 4;
    u32x4 xr[D][2][2];                                         // [ring slot][k step][16-byte piece]
    auto load_x = [&](int st, int slot) {
        const bool first = st < stages0;
        const unsigned cb = (unsigned)((first ? st : st - stages0) * 32 + khalf * 8) * 4u;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const unsigned off = (prow == OOB || st >= stages) ? OOB : prow + cb + ks * 64 + h * 16;
                xr[slot][ks][h] = first ? __builtin_amdgcn_raw_buffer_load_b128(rs0, off, 0, 0) : __builtin_amdgcn_raw_buffer_load_b128(rs1, off, 0, 0);
            }
    };
    // weight stage st: planes [chunk of 16 = 2 st + ks][1 tap][plane][k half][N][8]; LDS image [ks][plane][k half][BN rows][16 B]
    auto issue_w = [&](int st) {
        char* dst = wbuf + (st & 1) * WST;
#pragma unroll
        for (int k = 0; k < (WI + 3) / 4; ++k) {
            const int q = wave + 4 * k;
            if (q < WI) {
                constexpr int PER = BN / 64 > 0 ? BN / 64 : 1;  // 64-row blocks per (ks, plane, kh)
                const int r8 = BN >= 64 ? q / PER : q * 2 + (lane >> 5), nb = BN >= 64 ? q - (q / PER) * PER : 0;
                const int row = BN >= 64 ? nb * 64 + lane : (lane & 31);
                const int ks = r8 >> 2, r4 = r8 & 3;
                const unsigned woff = (unsigned)((((long long)(2 * st + ks) * 4 + r4) * a.N + n0 + row) * 16);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr_t)(dst + q * 1024), 16, woff, 0, 0, 0);
            }
        }
    };

    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    issue_w(0);
#pragma unroll
    for (int d = 0; d < D - 1; ++d) load_x(d, d);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int base = 0; base < stages; base += D) {            // (the last round may hold stages past the end: loads return zeros, MFMAs are skipped)
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const int st = base + u;
            if (st + 1 < stages) issue_w(st + 1);
            load_x(st + D - 1, (u + D - 1) % D);               // past the end: out-of-range offsets (zeros), never used
            const char* Bt = wbuf + (st & 1) * WST + khalf * (BN * 16) + (lane & 31) * 16;
            if (st < stages)               // (uniform; a stage past the end has no weights in LDS: never-written LDS may hold NaN patterns)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const f32x4 v0 = __builtin_bit_cast(f32x4, xr[u][ks][0]), v1 = __builtin_bit_cast(f32x4, xr[u][ks][1]);
                unsigned h0, l0, h1, l1, h2, l2, h3, l3;
                split_pair(v0[0] * sa, v0[1] * sa, h0, l0); split_pair(v0[2] * sa, v0[3] * sa, h1, l1);
                split_pair(v1[0] * sa, v1[1] * sa, h2, l2); split_pair(v1[2] * sa, v1[3] * sa, h3, l3);
                const u32x4 xh = u32x4{h0, h1, h2, h3}, xl = u32x4{l0, l1, l2, l3};
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const u32x4 wh = *reinterpret_cast<const u32x4*>(Bt + ((ks * 2 + 0) * 2) * (BN * 16) + j * 512);
                    const u32x4 wl = *reinterpret_cast<const u32x4*>(Bt + ((ks * 2 + 1) * 2) * (BN * 16) + j * 512);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, xl), __builtin_bit_cast(f16x8, wh), acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, xh), __builtin_bit_cast(f16x8, wl), acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, xh), __builtin_bit_cast(f16x8, wh), acc[j], 0, 0, 0);
                }
            }
            // the next stage's weights have landed once everything but the four pixel loads issued after them is back
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            __syncthreads();
        }
    }
    // D[pixel][n]: col n = lane & 31, pixel = (r & 3) + 8 (r >> 2) + 4 khalf of the wave's 32
    const long long mw = (long long)blockIdx.x * 128 + wave * 32;
    if constexpr (PSI) {
        // q . wpsi per pixel: per-lane partial sums over this lane's TN columns, then a reduce-scatter over the 32 lanes of the
        // half wave (16 values -> 8 -> 4 -> 2 -> 1 with masks 16, 8, 4, 2, then a plain exchange with mask 1): 16 shuffles
        // instead of 80, and lane l ends with the sum of pixel register r = (l & 31) >> 1
        float p[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) p[r] = 0.f;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = j * 32 + (lane & 31);
            const float sc = o.scale[n], sh = o.shift[n], wp = a.wpsi[blockIdx.z][n];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = ldexpf(acc[j][r], einv) * sc + sh;
                if (a.relu) v = fmaxf(v, 0.f);
                p[r] = fmaf(v, wp, p[r]);
            }
        }
        float p8[8], p4[4], p2[2];
        {
            const bool up = lane & 16;
#pragma unroll
            for (int i = 0; i < 8; ++i) p8[i] = (up ? p[8 + i] : p[i]) + __shfl_xor(up ? p[i] : p[8 + i], 16);
        }
        {
            const bool up = lane & 8;
#pragma unroll
            for (int i = 0; i < 4; ++i) p4[i] = (up ? p8[4 + i] : p8[i]) + __shfl_xor(up ? p8[i] : p8[4 + i], 8);
        }
        {
            const bool up = lane & 4;
#pragma unroll
            for (int i = 0; i < 2; ++i) p2[i] = (up ? p4[2 + i] : p4[i]) + __shfl_xor(up ? p4[i] : p4[2 + i], 4);
        }
        float dot;
        {
            const bool up = lane & 2;
            dot = (up ? p2[1] : p2[0]) + __shfl_xor(up ? p2[0] : p2[1], 2);
        }
        dot += __shfl_xor(dot, 1);
        const float z = dot * a.st[blockIdx.z][0] + a.st[blockIdx.z][1];
        const float psi = 1.f / (1.f + expf(-z));
        float* psil = reinterpret_cast<float*>(wbuf) + wave * 32;      // the weight buffers are free: the loop ended on a barrier
        const int r = (lane & 31) >> 1;
        if (!(lane & 1)) {
            const int pix = (r & 3) + 8 * (r >> 2) + 4 * khalf;
            psil[pix] = psi;
            if (a.psi_only && mw + pix < a.M) a.gated[blockIdx.z][mw + pix] = psi;       // the consumer multiplies while it stages x
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // same wave wrote what it reads
        // gated = x * psi: the wave's 32 pixels x C channels as float4s, 1 KB contiguous per instruction; x = source 1, read a
        // moment ago as the second half of K (L2 / MALL hits)
        const int C4 = a.C >> 2, total = 32 * C4;
        const f32x4* x4 = reinterpret_cast<const f32x4*>(o.src1) + mw * C4;
        f32x4* g4 = reinterpret_cast<f32x4*>(a.gated[blockIdx.z]) + mw * C4;
        const long long lim = (a.M - mw) * C4;                            // float4s of this wave that exist
        float gmx = 0.f;
#pragma unroll 1
        for (int i0 = lane; i0 < total; i0 += 256) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int i = i0 + 64 * u; if (i < total && i < lim) v[u] = x4[i]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + 64 * u;
                if (i < total && i < lim) {
                    const float ps = psil[i / C4];
                    const f32x4 g = v[u] * ps;
                    if (!a.psi_only) g4[i] = g;
                    gmx = fmaxf(fmaxf(gmx, fmaxf(fabsf(g[0]), fabsf(g[1]))), fmaxf(fabsf(g[2]), fabsf(g[3])));
                }
            }
        }
        if (a.gated_amax[blockIdx.z]) wave_amax(gmx, a.gated_amax[blockIdx.z]);
    } else {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + j * 32 + (lane & 31);
        const float sc = o.scale[n], sh = o.shift[n];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long mm = mw + (r & 3) + 8 * (r >> 2) + 4 * khalf;
            if (mm < a.M) {
                float v = ldexpf(acc[j][r], einv) * sc + sh;
                if (a.relu) v = fmaxf(v, 0.f);
                o.out[mm * a.N + n] = v;
            }
        }
    }
    }
}

// ------------------------------------------------------------------ 3x3 weight gradient on the split scheme
// dW[tap][ci][co] = sum over pixels of X[pixel + tap][ci] dY[pixel][co] (training, nbp_utils.py:383-395 through autograd): the
// reduction runs over PIXELS, so both MFMA operands need 8 consecutive pixels of one channel per lane while the tensors are
// channel-contiguous (NHWC).  gfx950's transpose read does that turn for free: the tile sits in LDS as it comes from memory --
// fp16 hi / lo planes [32-channel half][pixel][32 channels], 64 B per pixel -- and ds_read_b64_tr_b16 hands lane i of a 16-lane
// group column i of a [4 pixels][16 channels] block (lane address = pixel row 4 (l >> 4) + ((l & 15) >> 2), channels 4 (l & 3) ..:
// tools/probes/tr16_probe.hip), i.e. exactly a quarter of a 32x32x16 MFMA fragment; a filter tap is a ROW offset of the
// X plane, so no shifted copies.  64-B rows put the four pixel rows of a block on banks 0 / 16 / 32 / 48 and the second
// 16-channel group 8 banks further: conflict-free.
// Workgroup = 64 (ci) x 64 (co) block of dW for all nine taps (wave = 32 x 32, nine accumulator tiles), walking 2 x 32-pixel
// tiles like wgrad_halo_kernel (nbp_train.hip); per tile the 4 x 34 halo of X and the 64 pixels of dY go global -> registers ->
// scale by 2^(14 - e), split -> LDS; a 16-pixel K step is 2 + 2 transpose reads per operand and three exact MFMAs.
struct WgradSplitArgs {
    const float* src0; const float* src1;
    int C0, C1, ups, H, W, Hs, Ws;
    const float* dy; int N;
    unsigned bytes0, bytes1, bytesy;
    int co_tiles, n_tiles, splits;
    const unsigned* amax0; const unsigned* amax1; const unsigned* amaxy;
    float* part;           // [split][tap][Ctot][N]
};
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_t;

__device__ __forceinline__ f16x8 tr_frag(const char* p) {      // 8 consecutive pixel rows (p, p + 4 rows) of this lane's channel
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(p));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(p + 256));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(f16x8, v);
}

// TW = tile width: 2 x 32 pixels, or 4 x 16 for the 16-pixel-wide level
template <int TW>
__global__ __launch_bounds__(256, 2) void wgrad_split_kernel(WgradSplitArgs a) {
    constexpr int TR = 64 / TW, HW_ = TW + 2, HP = (TR + 2) * HW_;      // tile rows; halo pixels of the tile
    constexpr int NX = (HP * 16 + 255) / 256;                          // float4 items of the X halo per thread
    constexpr int XPL = HP * 64, YPL = 64 * 64;                        // bytes of one (plane, channel half) region
    constexpr int XB = 4 * XPL;                                // X: [plane][half] regions, then dY alike
    extern __shared__ __attribute__((aligned(16))) char wl[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1, kh = lane >> 5, ln = lane & 31;
    const int ci0 = (blockIdx.x / a.co_tiles) * 64, co0 = (blockIdx.x % a.co_tiles) * 64;
    const int Ctot = a.C0 + a.C1;
    const bool first = ci0 < a.C0;
    const int Cs = first ? a.C0 : a.C1;
    const int cbase = first ? ci0 : ci0 - a.C0;
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(first ? a.src0 : a.src1), 0, first ? a.bytes0 : a.bytes1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, a.bytesy, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    const int tiles_x = a.W / TW, tiles_y = a.H / TR;
    const int ex = amax_exponent(read_amax(first ? a.amax0 : a.amax1)), ey = amax_exponent(read_amax(a.amaxy));
    const float sx = pow2f(SPLIT_EXP - ex), sy = pow2f(SPLIT_EXP - ey);
    const int einv = ex + ey - 2 * SPLIT_EXP;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // staging roles: item f = tid + 256 k is the float4 of channels 4 (f & 15) .. + 3 of pixel f >> 4
    const int c4 = tid & 15;
    const int sdst = (c4 >> 3) * 1 /* half */;                    // region half; byte offset inside the pixel row: (c4 & 7) * 8
    // fragment lane offsets: pixel row 8 kh + ((lane & 15) >> 2) (+ 4 for the second read), channels 16 ((lane >> 4) & 1) + 4 (lane & 3)
    const int loff = (8 * kh + ((lane & 15) >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    const char* const xa_hi = wl + (0 * 2 + wi) * XPL + loff;
    const char* const xa_lo = wl + (1 * 2 + wi) * XPL + loff;
    const char* const yb_hi = wl + XB + (0 * 2 + wj) * YPL + loff;
    const char* const yb_lo = wl + XB + (1 * 2 + wj) * YPL + loff;

    // The next tile's pixels are fetched while the current tile is multiplied (a tile is 108 MFMAs per wave, ~1.6 us: less than one
    // HBM round trip, so loads issued at the top of an iteration were exposed every time)
    u32x4 xr[NX], yr[4];
    auto fetch = [&](int tile) {
        int t = tile;
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y;
        const int b = t / tiles_y;
        const int y0 = ty * TR, x0 = tx * TW;
#pragma unroll
        for (int k = 0; k < NX; ++k) {         // X halo: HP pixels x 16 float4
            const int f = tid + 256 * k;
            const int hr = f >> 4;
            const int hy = hr / HW_, hx = hr - hy * HW_;
            const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
            const bool ok = hr < HP && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
            const unsigned off = ok ? (unsigned)(((b * a.Hs + (yy >> a.ups)) * a.Ws + (xx >> a.ups)) * Cs + cbase + c4 * 4) * 4u : OOB;
            xr[k] = __builtin_amdgcn_raw_buffer_load_b128(rsx, off, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {          // dY: 64 pixels x 16 float4
            const int pz = (tid + 256 * k) >> 4;
            const long long m = ((long long)b * a.H + y0 + pz / TW) * a.W + x0 + pz % TW;
            yr[k] = __builtin_amdgcn_raw_buffer_load_b128(rsy, (unsigned)((m * a.N + co0 + c4 * 4) * 4), 0, 0);
        }
    };
    if ((int)blockIdx.y < a.n_tiles) fetch(blockIdx.y);
    for (int tile = blockIdx.y; tile < a.n_tiles; tile += a.splits) {
        __syncthreads();                       // every wave is done with the previous tile's planes
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            const int hr = (tid + 256 * k) >> 4;
            if (hr >= HP) continue;
            const f32x4 v = __builtin_bit_cast(f32x4, xr[k]);
            unsigned h0, l0, h1, l1;
            split_pair(v[0] * sx, v[1] * sx, h0, l0);
            split_pair(v[2] * sx, v[3] * sx, h1, l1);
            char* d = wl + sdst * XPL + hr * 64 + (c4 & 7) * 8;
            *reinterpret_cast<u32x2*>(d) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(d + 2 * XPL) = u32x2{l0, l1};
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int pz = (tid + 256 * k) >> 4;
            const f32x4 v = __builtin_bit_cast(f32x4, yr[k]);
            unsigned h0, l0, h1, l1;
            split_pair(v[0] * sy, v[1] * sy, h0, l0);
            split_pair(v[2] * sy, v[3] * sy, h1, l1);
            char* d = wl + XB + sdst * YPL + pz * 64 + (c4 & 7) * 8;
            *reinterpret_cast<u32x2*>(d) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(d + 2 * YPL) = u32x2{l0, l1};
        }
        __syncthreads();
        if (tile + a.splits < a.n_tiles) fetch(tile + a.splits);
#pragma unroll
        for (int r = 0; r < TR; ++r) {
            // the row offset is made opaque to the compiler: left visible, it keeps the fragments of the halo rows that tile rows r
            // and r + 1 share ((r, dy + 1) = (r + 1, dy)) in registers across the r iterations and spills 36 dwords around the
            // staging; re-reading them costs 48 more transpose reads per tile and is 1.5x faster (1.20 -> 0.78 ms at 256^2, 64 x 64).
            // Round 5 measured the explicit form of that sharing -- column-major over the taps, the TR + 2 halo rows of a column
            // serving all (r, dy) pairs: 28 fragment reads per 54 MFMAs instead of 40 -- at 104 B of scratch per lane and 13 % SLOWER
            // (0.727 against 0.642 ms on the same layer, profiles/r05/rejected_experiments.txt): 144 accumulator + 52 prefetch
            // registers leave ~40 for fragments, the shared rows need 48
            int rofs = r * HW_ * 64;
            asm volatile("" : "+v"(rofs));
#pragma unroll
            for (int s = 0; s < TW / 16; ++s) {
                const int p0 = r * TW + 16 * s;                               // first of the step's 16 pixels
                const f16x8 bh = tr_frag(yb_hi + p0 * 64), bl = tr_frag(yb_lo + p0 * 64);
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int hp0 = (tap / 3) * HW_ + 16 * s + tap % 3;       // halo pixel of (row r + dy, column 16 s + dx), less row r
                    const f16x8 ah = tr_frag(xa_hi + rofs + hp0 * 64), al = tr_frag(xa_lo + rofs + hp0 * 64);
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[tap], 0, 0, 0);
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[tap], 0, 0, 0);
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[tap], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        float* out = a.part + (((long long)blockIdx.y * 9 + tap) * Ctot) * a.N;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = ci0 + wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            out[(long long)ci * a.N + co0 + wj * 32 + ln] = ldexpf(acc[tap][r], einv);
        }
    }
}

// ------------------------------------------------------------------ weight gradient of an up_conv layer in parity form
// The layer (x2 nearest upsample + 3 x 3, nbp_model.py:25-33) is four 2 x 2 convolutions of the LOW-resolution input, one per output
// parity (conv3x3_halo_h2_kernel<..., PH>), so its weight gradient is
//   dWc[py, px][r][t][c][n] = sum over low-resolution pixels (v, u) of X[v - 1 + py + r, u - 1 + px + t, c] dY[2 v + py, 2 u + px, n]:
// 16 tap-GEMMs over M / 4 pixels each where the 3 x 3 form over the upsampled image does 9 over M (2.25 x fewer MFMAs), and
//   dW[ky][kx] = sum of dWc[py, px][r][t] over the (py, r) whose pre-summed filter row contains ky (and columns alike)
// is taken by the reduce kernel below.  A workgroup owns a 64 (c) x 64 (n) block of ONE parity (blockIdx.z) and walks 64-pixel tiles of
// the low-resolution image like wgrad_split_kernel: the (TR + 1) x (TW + 1) halo of X it needs (rows v0 - 1 + py .., columns
// u0 - 1 + px ..) and the tile's 64 pixels of dY's parity plane go global -> registers -> scale, split -> LDS fp16 planes
// [32-channel half][pixel][32 channels]; transpose reads hand both MFMA operands 8 consecutive pixels of one channel per lane.
// Four accumulator tiles per wave instead of nine: 64 + 44 registers of accumulators and prefetch, no scratch (two or three workgroups
// per CU measure the same; the bound of two leaves the allocator room).
template <int TW>
__global__ __launch_bounds__(256, 2) void wgrad_up_split_kernel(WgradSplitArgs a) {
    constexpr int TR = 64 / TW, HW_ = TW + 2, HP = (TR + 1) * HW_;     // tile rows; halo pixels ((TR + 1) rows x (TW + 2): TW + 1 are used)
    constexpr int NX = (HP * 16 + 255) / 256;
    constexpr int XPL = HP * 64, YPL = 64 * 64;
    constexpr int XB = 4 * XPL;
    extern __shared__ __attribute__((aligned(16))) char wl[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1, kh = lane >> 5, ln = lane & 31;
    const int ci0 = (blockIdx.x / a.co_tiles) * 64, co0 = (blockIdx.x % a.co_tiles) * 64;
    const int py = blockIdx.z >> 1, px = blockIdx.z & 1;
    const int C = a.C0;
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src0), 0, a.bytes0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, a.bytesy, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    const int tiles_x = a.Ws / TW, tiles_y = a.Hs / TR;              // tiles of the LOW-resolution image (a.H / a.W: dY's size)
    const int ex = amax_exponent(read_amax(a.amax0)), ey = amax_exponent(read_amax(a.amaxy));
    const float sx = pow2f(SPLIT_EXP - ex), sy = pow2f(SPLIT_EXP - ey);
    const int einv = ex + ey - 2 * SPLIT_EXP;

    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int c4 = tid & 15;
    const int sdst = c4 >> 3;
    const int loff = (8 * kh + ((lane & 15) >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    const char* const xa_hi = wl + (0 * 2 + wi) * XPL + loff;
    const char* const xa_lo = wl + (1 * 2 + wi) * XPL + loff;
    const char* const yb_hi = wl + XB + (0 * 2 + wj) * YPL + loff;
    const char* const yb_lo = wl + XB + (1 * 2 + wj) * YPL + loff;

    u32x4 xr[NX], yr[4];
    auto fetch = [&](int tile) {
        int t = tile;
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y;
        const int b = t / tiles_y;
        const int v0 = ty * TR, u0 = tx * TW;
#pragma unroll
        for (int k = 0; k < NX; ++k) {         // X halo: halo pixel (hy, hx) = image (v0 - 1 + py + hy, u0 - 1 + px + hx)
            const int f = tid + 256 * k;
            const int hr = f >> 4;
            const int hy = hr / HW_, hx = hr - hy * HW_;
            const int yy = v0 - 1 + py + hy, xx = u0 - 1 + px + hx;
            const bool ok = hr < HP && (unsigned)yy < (unsigned)a.Hs && (unsigned)xx < (unsigned)a.Ws;
            const unsigned off = ok ? (unsigned)(((b * a.Hs + yy) * a.Ws + xx) * C + ci0 + c4 * 4) * 4u : OOB;
            xr[k] = __builtin_amdgcn_raw_buffer_load_b128(rsx, off, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {          // dY: the tile's 64 pixels of the parity plane
            const int pz = (tid + 256 * k) >> 4;
            const long long m = ((long long)b * a.H + 2 * (v0 + pz / TW) + py) * a.W + 2 * (u0 + pz % TW) + px;
            yr[k] = __builtin_amdgcn_raw_buffer_load_b128(rsy, (unsigned)((m * a.N + co0 + c4 * 4) * 4), 0, 0);
        }
    };
    if ((int)blockIdx.y < a.n_tiles) fetch(blockIdx.y);
    for (int tile = blockIdx.y; tile < a.n_tiles; tile += a.splits) {
        __syncthreads();                       // every wave is done with the previous tile's planes
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            const int hr = (tid + 256 * k) >> 4;
            if (hr >= HP) continue;
            const f32x4 v = __builtin_bit_cast(f32x4, xr[k]);
            unsigned h0, l0, h1, l1;
            split_pair(v[0] * sx, v[1] * sx, h0, l0);
            split_pair(v[2] * sx, v[3] * sx, h1, l1);
            char* d = wl + sdst * XPL + hr * 64 + (c4 & 7) * 8;
            *reinterpret_cast<u32x2*>(d) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(d + 2 * XPL) = u32x2{l0, l1};
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int pz = (tid + 256 * k) >> 4;
            const f32x4 v = __builtin_bit_cast(f32x4, yr[k]);
            unsigned h0, l0, h1, l1;
            split_pair(v[0] * sy, v[1] * sy, h0, l0);
            split_pair(v[2] * sy, v[3] * sy, h1, l1);
            char* d = wl + XB + sdst * YPL + pz * 64 + (c4 & 7) * 8;
            *reinterpret_cast<u32x2*>(d) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(d + 2 * YPL) = u32x2{l0, l1};
        }
        __syncthreads();
        if (tile + a.splits < a.n_tiles) fetch(tile + a.splits);
#pragma unroll
        for (int r = 0; r < TR; ++r) {
            int rofs = r * HW_ * 64;           // opaque to the compiler (as in wgrad_split_kernel: no fragments carried across tile rows)
            asm volatile("" : "+v"(rofs));
#pragma unroll
            for (int s = 0; s < TW / 16; ++s) {
                const int p0 = r * TW + 16 * s;
                const f16x8 bh = tr_frag(yb_hi + p0 * 64), bl = tr_frag(yb_lo + p0 * 64);
#pragma unroll
                for (int tap = 0; tap < 4; ++tap) {
                    const int hp0 = (tap >> 1) * HW_ + 16 * s + (tap & 1);            // halo pixel of (row r + r', column 16 s + t'), less row r
                    const f16x8 ah = tr_frag(xa_hi + rofs + hp0 * 64), al = tr_frag(xa_lo + rofs + hp0 * 64);
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[tap], 0, 0, 0);
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[tap], 0, 0, 0);
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[tap], 0, 0, 0);
                }
            }
        }
    }
    // part [split][parity][tap][C][N]
#pragma unroll
    for (int tap = 0; tap < 4; ++tap) {
        float* out = a.part + ((((long long)blockIdx.y * 4 + blockIdx.z) * 4 + tap) * C) * a.N;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = ci0 + wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            out[(long long)ci * a.N + co0 + wj * 32 + ln] = ldexpf(acc[tap][r], einv);
        }
    }
}

// ------------------------------------------------------------------ weight gradient of a 1x1 layer on the split scheme
// dW[c][n] = sum over pixels of X[m][c] dY[m][n] (training's W_g / W_x layers on the large levels, where the fp32-pipe kernel ran at a
// quarter of the HBM rate and needed dY padded from 32 to 64 channels): wgrad_split_kernel without halo or taps.  A workgroup owns a
// 64 (c) x 64 (n) block -- columns beyond N are zero pixels and are not written -- and walks 64-pixel runs of the [M][.] tensors; one
// accumulator tile per wave, so four workgroups share a CU and cover each other's loads.
__global__ __launch_bounds__(256, 4) void wgrad_1x1_split_kernel(WgradSplitArgs a) {
    constexpr int PL = 64 * 64;                // bytes of one (plane, channel half) region: 64 pixels x 64 B
    extern __shared__ __attribute__((aligned(16))) char wl[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1, kh = lane >> 5, ln = lane & 31;
    const int ci0 = (blockIdx.x / a.co_tiles) * 64, co0 = (blockIdx.x % a.co_tiles) * 64;
    const int C = a.C0;
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src0), 0, a.bytes0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, a.bytesy, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    const int ex = amax_exponent(read_amax(a.amax0)), ey = amax_exponent(read_amax(a.amaxy));
    const float sx = pow2f(SPLIT_EXP - ex), sy = pow2f(SPLIT_EXP - ey);
    const int einv = ex + ey - 2 * SPLIT_EXP;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int c4 = tid & 15, sdst = c4 >> 3;
    const bool ycol = co0 + c4 * 4 < a.N;      // (N % 4 == 0: a float4 of dY's columns exists or not as a whole)
    const int loff = (8 * kh + ((lane & 15) >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    const char* const xa_hi = wl + (0 * 2 + wi) * PL + loff;
    const char* const xa_lo = wl + (1 * 2 + wi) * PL + loff;
    const char* const yb_hi = wl + 4 * PL + (0 * 2 + wj) * PL + loff;
    const char* const yb_lo = wl + 4 * PL + (1 * 2 + wj) * PL + loff;
    u32x4 xr[4], yr[4];
    const long long M = (long long)a.H * a.W;  // (pixels; the launcher passes H = M / W)
    auto fetch = [&](int tile) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long m = (long long)tile * 64 + ((tid + 256 * k) >> 4);
            const bool ok = m < M;
            xr[k] = __builtin_amdgcn_raw_buffer_load_b128(rsx, ok ? (unsigned)((m * C + ci0 + c4 * 4) * 4) : OOB, 0, 0);
            yr[k] = __builtin_amdgcn_raw_buffer_load_b128(rsy, (ok && ycol) ? (unsigned)((m * a.N + co0 + c4 * 4) * 4) : OOB, 0, 0);
        }
    };
    if ((int)blockIdx.y < a.n_tiles) fetch(blockIdx.y);
    for (int tile = blockIdx.y; tile < a.n_tiles; tile += a.splits) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int pz = (tid + 256 * k) >> 4;
            const f32x4 v = __builtin_bit_cast(f32x4, xr[k]), w = __builtin_bit_cast(f32x4, yr[k]);
            unsigned h0, l0, h1, l1;
            split_pair(v[0] * sx, v[1] * sx, h0, l0);
            split_pair(v[2] * sx, v[3] * sx, h1, l1);
            char* d = wl + sdst * PL + pz * 64 + (c4 & 7) * 8;
            *reinterpret_cast<u32x2*>(d) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(d + 2 * PL) = u32x2{l0, l1};
            split_pair(w[0] * sy, w[1] * sy, h0, l0);
            split_pair(w[2] * sy, w[3] * sy, h1, l1);
            d += 4 * PL;
            *reinterpret_cast<u32x2*>(d) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(d + 2 * PL) = u32x2{l0, l1};
        }
        __syncthreads();
        if (tile + a.splits < a.n_tiles) fetch(tile + a.splits);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const f16x8 bh = tr_frag(yb_hi + s * 1024), bl = tr_frag(yb_lo + s * 1024);
            const f16x8 ah = tr_frag(xa_hi + s * 1024), al = tr_frag(xa_lo + s * 1024);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
        }
    }
    float* out = a.part + (long long)blockIdx.y * C * a.N;      // part [split][C][N]
    const int co = co0 + wj * 32 + ln;
    if (co < a.N)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = ci0 + wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            out[(long long)ci * a.N + co] = ldexpf(acc[r], einv);
        }
}

// dW OIHW [N][C][3][3] from the parity partials [split][parity][tap][C][N]: filter row ky collects the (py, r) pairs whose pre-summed
// row contains it -- ky = 0: (0, 0), (1, 0); ky = 1: (0, 1), (1, 0); ky = 2: (0, 1), (1, 1) -- columns alike; slices in slice order,
// the four entries in a fixed order (deterministic)
__global__ __launch_bounds__(256) void wgrad_up_reduce_kernel(const float* __restrict__ part, int splits, int C, int N,
                                                              float* __restrict__ dw) {
    const long long total = (long long)N * C * 9;
    const long long plane = (long long)C * N, slice = 16 * plane;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int co = (int)(i % N);
        long long t = i / N;
        const int ci = (int)(t % C), k9 = (int)(t / C);
        const int ky = k9 / 3, kx = k9 % 3;
        // (parity bit, tap bit) pairs per filter index
        const int pa[3][2] = {{0, 1}, {0, 1}, {0, 1}};
        const int ta[3][2] = {{0, 0}, {1, 0}, {1, 1}};
        float s = 0.f;
        const float* p = part + (long long)ci * N + co;
        for (int k = 0; k < splits; ++k) {
            float v = 0.f;
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    const int q = pa[ky][e] * 2 + pa[kx][f], tap = ta[ky][e] * 2 + ta[kx][f];
                    v += p[(long long)k * slice + (q * 4 + tap) * plane];
                }
            s += v;
        }
        dw[((long long)co * C + ci) * 9 + k9] = s;
    }
}

// Grid-stride over float4s of the output, two positions per thread and iteration, the slices' loads of both issued together
// (four slices at a time) and added in slice order -- a serial load-add chain costs one memory round trip per slice, and
// one float4 per thread is bound by the wave launch rate (measured: 31 us serial / 56 us one-per-thread / see DESIGN.md).
__global__ __launch_bounds__(256) void splitk_reduce_split_kernel(const float* __restrict__ partial_all, int split_k,
                                                                  unsigned MN, unsigned N, SplitOps o0, SplitOps o1, int relu) {
    const float* partial = partial_all + (size_t)blockIdx.y * split_k * MN;
    const float* scale = blockIdx.y ? o1.scale : o0.scale;
    const float* shift = blockIdx.y ? o1.shift : o0.shift;
    float* out = blockIdx.y ? o1.out : o0.out;
    unsigned* amax_out = blockIdx.y ? o1.amax_out : o0.amax_out;
    const unsigned stride = gridDim.x * 256u * 4u;
    float mx = 0.f;
    for (unsigned i0 = (blockIdx.x * 256u + threadIdx.x) * 4u; i0 < MN; i0 += 2 * stride) {
        const unsigned i1 = i0 + stride;
        const bool two = i1 < MN;
        f32x4 v0 = *reinterpret_cast<const f32x4*>(partial + i0);
        f32x4 v1 = two ? *reinterpret_cast<const f32x4*>(partial + i1) : f32x4{0.f, 0.f, 0.f, 0.f};
        for (int s0 = 1; s0 < split_k; s0 += 4) {
            f32x4 u0[4], u1[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool in = s0 + k < split_k;
                const float* ps = partial + (size_t)(s0 + k) * MN;
                u0[k] = in ? *reinterpret_cast<const f32x4*>(ps + i0) : f32x4{0.f, 0.f, 0.f, 0.f};
                u1[k] = in && two ? *reinterpret_cast<const f32x4*>(ps + i1) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (s0 + k < split_k) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v0[e] += u0[k][e]; v1[e] += u1[k][e]; }
                }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h && !two) break;
            const unsigned i = h ? i1 : i0;
            f32x4 v = h ? v1 : v0;
            const unsigned n = i % N;
            const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + n);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = v[e] * sc[e] + sh[e];          // the partial sums already carry 1 / (s_x s_w)
                if (relu) v[e] = fmaxf(v[e], 0.f);
                mx = fmaxf(mx, fabsf(v[e]));
            }
            *reinterpret_cast<f32x4*>(out + i) = v;
        }
    }
    if (amax_out) block_amax(mx, amax_out);
}

// The same reduction for an encoder layer whose 2x2 max-pool follows (nbp_model.py:113-123): a thread owns the four pixels of a pool
// window (four channels of each), adds their slices in slice order -- the sums above, bit for bit -- and writes the window's
// maximum beside the four outputs.  Split-K launches (a single rollout: every encoder level below the first) lost the pool's
// own launch this way; launches that write final values pool in the convolution's epilogue.
__global__ __launch_bounds__(256) void splitk_reduce_pool_kernel(const float* __restrict__ partial, int split_k, unsigned MN, unsigned N,
                                                                 int H, int W, SplitOps o, int relu) {
    const unsigned N4 = N >> 2, Wp = (unsigned)W >> 1, Hp = (unsigned)H >> 1;
    const unsigned total = (MN >> 4);                          // pool windows x float4 columns
    const unsigned rowN = (unsigned)W * N;
    float mx = 0.f;
    for (unsigned p = blockIdx.x * 256u + threadIdx.x; p < total; p += gridDim.x * 256u) {
        const unsigned n4 = p % N4;
        unsigned q = p / N4;
        const unsigned xp = q % Wp; q /= Wp;
        const unsigned yp = q % Hp, b = q / Hp;
        const unsigned base = ((b * (unsigned)H + 2u * yp) * (unsigned)W + 2u * xp) * N + 4u * n4;
        const unsigned idx[4] = {base, base + N, base + rowN, base + rowN + N};
        f32x4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const f32x4*>(partial + idx[k]);
        for (int s0 = 1; s0 < split_k; s0 += 2) {              // two slices x four pixels in flight, added in slice order
            f32x4 u[2][4];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const bool in = s0 + t < split_k;
                const float* ps = partial + (size_t)(s0 + t) * MN;
#pragma unroll
                for (int k = 0; k < 4; ++k) u[t][k] = in ? *reinterpret_cast<const f32x4*>(ps + idx[k]) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
                if (s0 + t < split_k) {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[k][e] += u[t][k][e];
                }
        }
        const f32x4 sc = *reinterpret_cast<const f32x4*>(o.scale + 4u * n4);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(o.shift + 4u * n4);
        f32x4 pm;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float r = v[k][e] * sc[e] + sh[e];
                if (relu) r = fmaxf(r, 0.f);
                mx = fmaxf(mx, fabsf(r));
                v[k][e] = r;
                pm[e] = k ? fmaxf(pm[e], r) : r;
            }
            *reinterpret_cast<f32x4*>(o.out + idx[k]) = v[k];
        }
        *reinterpret_cast<f32x4*>(o.pool_out + 4u * p) = pm;
    }
    if (o.amax_out) block_amax(mx, o.amax_out);
}

// max |x| (float bits, atomicMax: the caller zeroes the slot); optional per-row scale for weights [N][per_row]
// words = 64 (activation convention) or 1 (max |w| of a layer)
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, long long n, const float* __restrict__ row_scale,
                                                   long long per_row, unsigned* __restrict__ out, unsigned words) {
    float mx = 0.f;
    if (!row_scale && (n & 3) == 0 && ((uintptr_t)x & 15) == 0) {
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n / 4; i += (long long)gridDim.x * blockDim.x) {
            const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
            mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        }
    } else {
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
            mx = fmaxf(mx, fabsf(row_scale ? x[i] * row_scale[i / per_row] : x[i]));
    }
    block_amax(mx, out, words);
}

// planes [chunk of 16 channels][tap][hi|lo][k half][N][8 fp16] of w[n][c][tap] * (scale ? scale[n] : 1) * 2^(14 - e_w)
// transposed: the planes of w'[n][c][tap] = w[c][n][taps - 1 - tap] (w is then [C][N][taps]): the data-gradient convolution's
// weights -- input and output channels swapped, taps reversed -- straight from the layer's own tensor
__global__ void pack_conv_weight_h2_kernel(const float* __restrict__ w, int N, int C, int taps, const float* __restrict__ scale,
                                           int c_off, const unsigned* __restrict__ wamax, unsigned short* __restrict__ dst,
                                           int transposed = 0) {
    const float sw = pow2f(SPLIT_EXP - amax_exponent(*wamax));
    const long long total = (long long)N * C * taps;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int tap = (int)(i % taps);
        const long long t = i / taps;
        const int c = (int)(t % C);
        const int n = (int)(t / C);
        float v = transposed ? w[((long long)c * N + n) * taps + (taps - 1 - tap)] : w[i];
        if (scale) v *= scale[n];
        unsigned h, l;
        split_pair(v * sw, 0.f, h, l);
        const int cg = c_off + c;
        const long long base = ((long long)(cg >> 4) * taps + tap) * 4 + ((cg >> 3) & 1);
        dst[((base + 0) * N + n) * 8 + (cg & 7)] = (unsigned short)h;
        dst[((base + 2) * N + n) * 8 + (cg & 7)] = (unsigned short)l;
    }
}

// up_conv weights for the parity kernels: Wc[py * 2 + px][n][c][r * 2 + t] = sum of w[n][c][dy][dx] over the filter rows R(py, r)
// and columns R(px, t), R(0,0) = {0}, R(0,1) = {1,2}, R(1,0) = {0,1}, R(1,1) = {2} (indices 0..2 = offsets -1..1).  The sums
// are formed in double and go straight into the two fp16 pieces (no fp32 rounding in between).
__device__ __forceinline__ double upconv_combined(const double* v, int ph, int r, int t) {
    const int py = ph >> 1, px = ph & 1;
    const int y_lo = py == 0 ? (r == 0 ? 0 : 1) : (r == 0 ? 0 : 2), y_hi = py == 0 ? (r == 0 ? 0 : 2) : (r == 0 ? 1 : 2);
    const int x_lo = px == 0 ? (t == 0 ? 0 : 1) : (t == 0 ? 0 : 2), x_hi = px == 0 ? (t == 0 ? 0 : 2) : (t == 0 ? 1 : 2);
    double acc = 0.0;
    for (int y = y_lo; y <= y_hi; ++y)
        for (int x = x_lo; x <= x_hi; ++x) acc += v[y * 3 + x];
    return acc;
}
// pass 0 (dst == nullptr): max |Wc| into wamax; pass 1: planes [parity][chunk of 16][4 taps][hi|lo][k half][N][8 fp16]
__global__ __launch_bounds__(256) void pack_upconv_h2_kernel(const float* __restrict__ w, int N, int C, unsigned* __restrict__ wamax,
                                                             unsigned short* __restrict__ dst) {
    const long long NC = (long long)N * C;
    const double sw = dst ? (double)pow2f(SPLIT_EXP - amax_exponent(*wamax)) : 1.0;
    float mx = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < NC; i += (long long)gridDim.x * blockDim.x) {
        double v[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) v[k] = (double)w[i * 9 + k];
        const int n = (int)(i / C), c = (int)(i % C);
#pragma unroll
        for (int ph = 0; ph < 4; ++ph)
#pragma unroll
            for (int tap = 0; tap < 4; ++tap) {
                const double wc = upconv_combined(v, ph, tap >> 1, tap & 1);
                if (!dst) { mx = fmaxf(mx, fabsf((float)wc)); continue; }
                const double ws = wc * sw;
                const _Float16 h = (_Float16)(float)ws;
                const _Float16 l = (_Float16)(float)(ws - (double)(float)h);
                const long long base = (((long long)ph * (C >> 4) + (c >> 4)) * 4 + tap) * 4 + ((c >> 3) & 1);
                dst[((base + 0) * N + n) * 8 + (c & 7)] = __builtin_bit_cast(unsigned short, h);
                dst[((base + 2) * N + n) * 8 + (c & 7)] = __builtin_bit_cast(unsigned short, l);
            }
    }
    if (!dst) block_amax(mx, wamax, 1u);
}

// The same parity filters for the DATA GRADIENT of the layer (conv3x3_halo_h2_kernel<..., DG>): K runs over (parity q, output channel
// n of the layer), rows over its input channels c, and the tap (r', t') of chunk q holds Wc[q][1 - r'][1 - t'][n][c] (the kernel walks
// the parity plane with the forward's tap origin mirrored).  pass 0 (dst == nullptr): max |Wc| into wamax; pass 1: planes
// [q (N / 16) + n / 16][4 taps][hi|lo][k half][C][8 fp16].
__global__ __launch_bounds__(256) void pack_upconv_dgrad_h2_kernel(const float* __restrict__ w, int N, int C, unsigned* __restrict__ wamax,
                                                                   unsigned short* __restrict__ dst) {
    const long long NC = (long long)N * C;
    const double sw = dst ? (double)pow2f(SPLIT_EXP - amax_exponent(*wamax)) : 1.0;
    float mx = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < NC; i += (long long)gridDim.x * blockDim.x) {
        double v[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) v[k] = (double)w[i * 9 + k];
        const int n = (int)(i / C), c = (int)(i % C);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int tap = 0; tap < 4; ++tap) {
                const double wc = upconv_combined(v, q, 1 - (tap >> 1), 1 - (tap & 1));
                if (!dst) { mx = fmaxf(mx, fabsf((float)wc)); continue; }
                const double ws = wc * sw;
                const _Float16 h = (_Float16)(float)ws;
                const _Float16 l = (_Float16)(float)(ws - (double)(float)h);
                const long long base = (((long long)q * (N >> 4) + (n >> 4)) * 4 + tap) * 4 + ((n >> 3) & 1);
                dst[((base + 0) * C + c) * 8 + (n & 7)] = __builtin_bit_cast(unsigned short, h);
                dst[((base + 2) * C + c) * 8 + (n & 7)] = __builtin_bit_cast(unsigned short, l);
            }
    }
    if (!dst) block_amax(mx, wamax, 1u);
}

// ---- every weight pack of a training step in two launches (round 5).  A step packed each layer's weights when the layer ran --
// max |w|, forward planes, and again for the data gradient: ~250 launches of 5-20 us, 3 % of the step.  The descriptors (one per
// layer, built once by the host and kept on the device) name the layer's weight tensor, its kind and the persistent plane buffers;
// blockIdx.y = layer.  kind 0: 3x3 (planes as pack_conv_weight_h2_kernel, planes_t its transposed / tap-reversed form for the data
// gradient); kind 1: 1x1 (taps = 1, planes_t = w^T); kind 2: up_conv (parity planes as pack_upconv_h2_kernel, planes_t as
// pack_upconv_dgrad_h2_kernel; the max is over the pre-summed parity filters).  planes_t may be null.
struct PrepackDesc { const float* w; unsigned short* planes; unsigned short* planes_t; unsigned* wamax; int N, C, kind, pad; };
__global__ __launch_bounds__(256) void prepack_amax_kernel(const PrepackDesc* __restrict__ descs) {
    const PrepackDesc d = descs[blockIdx.y];
    float mx = 0.f;
    if (d.kind == 2) {
        const long long NC = (long long)d.N * d.C;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < NC; i += (long long)gridDim.x * blockDim.x) {
            double v[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) v[k] = (double)d.w[i * 9 + k];
#pragma unroll
            for (int ph = 0; ph < 4; ++ph)
#pragma unroll
                for (int tap = 0; tap < 4; ++tap) mx = fmaxf(mx, fabsf((float)upconv_combined(v, ph, tap >> 1, tap & 1)));
        }
    } else {
        const long long total = (long long)d.N * d.C * (d.kind == 0 ? 9 : 1);
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
            mx = fmaxf(mx, fabsf(d.w[i]));
    }
    // (uniform per block: every thread reaches the reduction)
    __shared__ float part[4];
#pragma unroll
    for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(d.wamax, __float_as_uint(fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]))));
}
// One thread packs an 8 x 8 block (rows n, channels c) of one tap: the forward planes take its 8 channels of a row as one 16-byte
// store, the data gradient's planes its 8 rows of a channel, and the 8 stores of either kind are one 128-byte run (the per-layer pack
// kernels write 2 bytes at a time; 0.83 ms for the network's weights that way, measured, against 0.2 ms of traffic).
__device__ __forceinline__ void prepack_store_block(const PrepackDesc& d, const unsigned (&hl)[8][8], int n8, int c8, int slot_f, int slot_t,
                                                    int slots) {
    const int N = d.N, C = d.C, n0 = n8 * 8, c0 = c8 * 8;
    {       // forward planes [.. c / 16][slot][hi|lo][k half][N][8]: rows n0..n0+7, one uint4 (8 channels) each
        const long long base = ((long long)(c0 >> 4) * slots + slot_f) * 4 + ((c0 >> 3) & 1);
        uint4* hi = (uint4*)(d.planes + ((base + 0) * N + n0) * 8);
        uint4* lo = (uint4*)(d.planes + ((base + 2) * N + n0) * 8);
#pragma unroll
        for (int nn = 0; nn < 8; ++nn) {
            uint4 h, l;
            h.x = (hl[nn][0] & 0xffffu) | (hl[nn][1] << 16); h.y = (hl[nn][2] & 0xffffu) | (hl[nn][3] << 16);
            h.z = (hl[nn][4] & 0xffffu) | (hl[nn][5] << 16); h.w = (hl[nn][6] & 0xffffu) | (hl[nn][7] << 16);
            l.x = (hl[nn][0] >> 16) | (hl[nn][1] & 0xffff0000u); l.y = (hl[nn][2] >> 16) | (hl[nn][3] & 0xffff0000u);
            l.z = (hl[nn][4] >> 16) | (hl[nn][5] & 0xffff0000u); l.w = (hl[nn][6] >> 16) | (hl[nn][7] & 0xffff0000u);
            hi[nn] = h; lo[nn] = l;
        }
    }
    if (slot_t >= 0) {       // data-gradient planes [.. n / 16][slot][hi|lo][k half][C][8]: rows c0..c0+7, 8 output channels each
        const long long base = ((long long)(n0 >> 4) * slots + slot_t) * 4 + ((n0 >> 3) & 1);
        uint4* hi = (uint4*)(d.planes_t + ((base + 0) * C + c0) * 8);
        uint4* lo = (uint4*)(d.planes_t + ((base + 2) * C + c0) * 8);
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) {
            uint4 h, l;
            h.x = (hl[0][cc] & 0xffffu) | (hl[1][cc] << 16); h.y = (hl[2][cc] & 0xffffu) | (hl[3][cc] << 16);
            h.z = (hl[4][cc] & 0xffffu) | (hl[5][cc] << 16); h.w = (hl[6][cc] & 0xffffu) | (hl[7][cc] << 16);
            l.x = (hl[0][cc] >> 16) | (hl[1][cc] & 0xffff0000u); l.y = (hl[2][cc] >> 16) | (hl[3][cc] & 0xffff0000u);
            l.z = (hl[4][cc] >> 16) | (hl[5][cc] & 0xffff0000u); l.w = (hl[6][cc] >> 16) | (hl[7][cc] & 0xffff0000u);
            hi[cc] = h; lo[cc] = l;
        }
    }
}
__global__ __launch_bounds__(256) void prepack_pack_kernel(const PrepackDesc* __restrict__ descs) {
    const PrepackDesc d = descs[blockIdx.y];
    const int N = d.N, C = d.C, N8 = N >> 3, C8 = C >> 3;
    const float swf = pow2f(SPLIT_EXP - amax_exponent(*d.wamax));
    if (d.kind == 2) {
        // up_conv: slot = tap (r, t) of parity ph; the forward planes of parity ph are [ph][c / 16][4 taps].., the data gradient's
        // [ph (N / 16) + n / 16][4 taps].. with the tap mirrored -- the same value, stored at tap (1 - r, 1 - t)
        const double sw = (double)swf;
        const long long total = (long long)N8 * C8 * 16;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
            const int pt = (int)(i & 15), ph = pt >> 2, tap = pt & 3;
            const long long t = i >> 4;
            const int c8 = (int)(t % C8), n8 = (int)(t / C8);
            const int py = ph >> 1, px = ph & 1, r = tap >> 1, tt = tap & 1;
            const int y_lo = py == 0 ? (r == 0 ? 0 : 1) : (r == 0 ? 0 : 2), y_hi = py == 0 ? (r == 0 ? 0 : 2) : (r == 0 ? 1 : 2);
            const int x_lo = px == 0 ? (tt == 0 ? 0 : 1) : (tt == 0 ? 0 : 2), x_hi = px == 0 ? (tt == 0 ? 0 : 2) : (tt == 0 ? 1 : 2);
            unsigned hl[8][8];
#pragma unroll
            for (int nn = 0; nn < 8; ++nn)
#pragma unroll
                for (int cc = 0; cc < 8; ++cc) {
                    const float* w9 = d.w + ((long long)(n8 * 8 + nn) * C + c8 * 8 + cc) * 9;
                    double acc = 0.0;                       // (the order of upconv_combined)
                    for (int y = y_lo; y <= y_hi; ++y)
                        for (int x = x_lo; x <= x_hi; ++x) acc += (double)w9[y * 3 + x];
                    const double ws = acc * sw;
                    const _Float16 h = (_Float16)(float)ws;
                    const _Float16 l = (_Float16)(float)(ws - (double)(float)h);
                    hl[nn][cc] = (unsigned)__builtin_bit_cast(unsigned short, h) | ((unsigned)__builtin_bit_cast(unsigned short, l) << 16);
                }
            // forward: chunk index ph (C / 16) + c / 16 -> fold ph into the pointer by offsetting the slot: slots per chunk = 4
            PrepackDesc e = d;
            e.planes = d.planes + (long long)ph * (C >> 4) * 4 * 4 * N * 8;
            if (d.planes_t) e.planes_t = d.planes_t + (long long)ph * (N >> 4) * 4 * 4 * C * 8;
            prepack_store_block(e, hl, n8, c8, tap, d.planes_t ? (1 - r) * 2 + (1 - tt) : -1, 4);
        }
        return;
    }
    const int taps = d.kind == 0 ? 9 : 1;
    const long long total = (long long)N8 * C8 * taps;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int tap = (int)(i % taps);
        const long long t = i / taps;
        const int c8 = (int)(t % C8), n8 = (int)(t / C8);
        unsigned hl[8][8];
#pragma unroll
        for (int nn = 0; nn < 8; ++nn)
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) {
                unsigned h, l;
                split_pair(d.w[((long long)(n8 * 8 + nn) * C + c8 * 8 + cc) * taps + tap] * swf, 0.f, h, l);
                hl[nn][cc] = (h & 0xffffu) | (l << 16);
            }
        prepack_store_block(d, hl, n8, c8, tap, d.planes_t ? taps - 1 - tap : -1, taps);     // data gradient: taps reversed
    }
}

// tile width the layer runs with: 32 (16 x 32 pixel tiles x 64 channels), 16 (16 x 16 x 128 channels) or 0 (not taken)
int split_tile_width(int H, int W, int N, int ksize) {
    if (ksize != 3 || H < 16 || H % 16) return 0;
    if (W >= 32 && W % 32 == 0 && N % 64 == 0) return 32;
    if (W >= 16 && W % 16 == 0 && N % 128 == 0) return 16;
    return 0;
}

template <int TW, int TM, int TN, bool PH, bool BS = false, bool P2 = false, bool DG = false>
int launch_h2(const SplitArgs& a, hipStream_t st, int tile) {
    {
        char nm[96];
        snprintf(nm, sizeof(nm), "conv3x3_halo_h2_kernel<%d, %d, %d, %s, %s, %s, %s>", TW, TM, TN, PH ? "true" : "false", BS ? "true" : "false",
                 P2 ? "true" : "false", DG ? "true" : "false");
        nbp_note_kernel_symbol(tile, nm);
    }
    constexpr int TH = 4 * TM * (32 / TW);
    constexpr int HPIX = (TH + 2) * (TW + 2), RS = (HPIX + 7) / 8 * 8 * 16 + 64, NB = TN / 2;
    constexpr size_t smem = 4 * (size_t)RS + 2 * (size_t)(PH ? 8 : 12) * NB * 1024 * (P2 ? 2 : 1) + (PH ? 0 : (HPIX * 4 + 15) / 16 * 16);      // (+ psi of the halo pixels)
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_h2_kernel<TW, TM, TN, PH, BS, P2, DG>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    // PH: tiles of the low-resolution image, four parities in blockIdx.z (fastest); DG: a.M counts the low-resolution output itself
    constexpr bool PHO = PH && !DG;
    dim3 grid((unsigned)(a.M / (PHO ? 4 : 1) / (TH * TW)), (unsigned)(a.N / (TN * 32)), (unsigned)(a.split_k * a.groups * (PHO ? (P2 ? 2 : 4) : 1)));
    conv3x3_halo_h2_kernel<TW, TM, TN, PH, BS, P2, DG><<<grid, 256, smem, st>>>(a);
    return nbp_launch_status();
}

}  // namespace

// Accuracy-driven split-K: one fp32 accumulator sees 3 roundings per 16 k (three MFMAs), and the distance of a chain's result to
// the exact sum grows like K_chain / sqrt(k per rounding) (measured, tools/diag/chain_error.py, profiles/r03/chain_error.txt: a
// K = 9216 chain sits 4.5x further from fp64 than stock torch CPU fp32 -- whose GEMM accumulates in blocks -- a K = 576 one 1.5x).
// Chains are therefore bounded whatever the occupancy says: slices beyond the occupancy-driven count exist for accuracy.  The
// bound has two tiers because the price is the partial sums' traffic (slices x M x N x 8 bytes): NBP_SPLIT_MAX_K_SMALL products
// (taps x channels) per chain where one slice of all groups is at most NBP_SPLIT_SMALL_MB (the 16 / 32-pixel levels: the extra
// slices are free there -- more workgroups -- or cost a few us), NBP_SPLIT_MAX_K on the larger outputs.  Defaults 1152 / 2304:
// every layer's LOCAL distance to fp64 stays within ~2.5x of torch fp32's at any batch size for +2 % of the B = 12 forward
// (576 / 2304 -- B = 1's natural chains at every batch -- costs 11 %, 1152 / 1152 8 %: profiles/r03/chain_bound_*.txt).  On the
// whole network the effect is modest, because on rollout inputs the forward's distance to fp64 is dominated by the chaotic
// amplification of ANY rounding difference (profiles/r03/layer_substitution_hard.txt: ONE inexact layer at 1e-9 of its range
// moves out1 by 1e-8 .. 9e-8 of its range, whichever arithmetic computes it).
static int chain_bounded_split(int sk, int cc, int taps, long long M, int N, int groups) {
    static const int max_k = nbp_tune_int("NBP_SPLIT_MAX_K", NBP_SPLIT_MAX_K_DEFAULT);
    static const int max_k_small = nbp_tune_int("NBP_SPLIT_MAX_K_SMALL", NBP_SPLIT_MAX_K_SMALL_DEFAULT);
    const bool small = (double)M * N * groups * 4.0 <= 64.0 * 1048576.0;       // one slice of all groups <= 64 MB
    const int mk = small && max_k_small > 0 ? max_k_small : max_k;
    if (mk <= 0) return sk;
    const int max_chunks = mk / (16 * taps) > 1 ? mk / (16 * taps) : 1;
    const int need = (int)nbp_cdiv(cc, max_chunks);
    return need > sk ? need : sk;
}

// tile == 0: the layer does not fit the split kernel (the caller runs the fp32 MFMA kernels on the fp32 pack)
// ups: the layer reads its input through the x2 nearest upsample; when the LOW-resolution image tiles, the parity kernels run
// (tile id NBP_TILE_SPLIT_UP), otherwise the plain kernel with the upsample folded into its gather.
// 16 x 32-pixel tiles leave CUs idle when a launch has fewer of them than CUs (a single rollout: 128 tiles at full resolution), and
// a workgroup's serial chain -- K / 16 chunks x 3 stages of 72 MFMAs per wave -- is what such a launch takes.  Below this many
// workgroups (before split-K) the launch uses 8 x 32-pixel tiles: twice the workgroups, half the MFMAs per stage, the same sums in the
// same order (NBP_SPLIT_R8_BLOCKS, 0 = never)
static long long half_rows_below() {
    static const int v = nbp_tune_int("NBP_SPLIT_R8_BLOCKS", 256);
    return v;
}

// (The split-K slices of an 8-row launch stay those of the 16-row plan: letting the doubled workgroup count buy fewer slices was
// measured no faster, and its longer chains put the rollout-input error statistics of tests/test_gpu_rollout_parity.py at the edge
// of their bound -- round 3.)

// ... and for launches whose last round of workgroups is a thin tail: with S = 512 resident workgroups (two per CU) a launch of 640
// 16-row workgroups is one full round plus a quarter-full one, 1280 half-size ones are 2.5 half rounds.  Measured (B = 5, whose
// full-resolution layers are 640 workgroups): -2.8 % of the forward.  A HALF-full last round (768, 1280 workgroups: B = 12, 20) does not
// gain -- its workgroups run alone on their CUs, 1.7 x faster -- and pays the 8-row tiles' doubled weight traffic (+0.3 .. +0.9 %), so
// the rule takes tails of up to 160 workgroups only.
static bool half_rows_for_tail(long long workgroups) {
    if (workgroups <= 512) return false;
    const long long rem = workgroups % 512;
    return rem > 0 && rem <= 160;
}

ConvPlan nbp_plan_conv_split(long long M, int N, int chunks_total, int split_k, int groups, int H, int W, int ksize, int ups) {
    ConvPlan p{0, 1, chunks_total};
    constexpr int min_blocks = 256;           // split-K workgroup target: one workgroup per CU
    if (ups && !((H | W) & 1)) {
        const int twu = split_tile_width(H / 2, W / 2, N, ksize);
        if (twu) {
            const int cc = chunks_total / 9 * 2;
            const long long blocks = (M / 4 / (16 * twu)) * (N / (twu == 32 ? 64 : 128)) * groups * 4;
            const bool r8 = blocks < half_rows_below();      // 8-row tiles: twice the workgroups, half the chain each
            int sk = split_k;
            if (sk <= 0) {
                sk = 1;
                while (blocks * sk < min_blocks && cc / (sk * 2) >= 4 && sk < 16) sk *= 2;
            }
            if (split_k == 0) sk = chain_bounded_split(sk, cc, 4, M, N, groups);      // (split_k < 0: by occupancy only)
            if (sk > cc) sk = cc;
            const int per = (int)nbp_cdiv(cc, sk);
            p.split_k = (int)nbp_cdiv(cc, per); p.chunks_per_split = per;
            const bool r8t = r8 || half_rows_for_tail((M / 4 / (16 * twu)) * (N / (twu == 32 ? 64 : 128)) * groups * 4 * p.split_k);
            p.tile = r8t ? NBP_TILE_SPLIT_UP_R8 : NBP_TILE_SPLIT_UP;
            return p;
        }
    }
    // one workgroup per CU without split-K beats two with it: the partial sums cost more than the idle barrier slots
    // (B = 4 forward 2.65 ms at 256, 2.85 at 512, 2.89 at 128)
    const int tw = split_tile_width(H, W, N, ksize);
    if (!tw) return p;
    const int cc = chunks_total / 9 * 2;      // the kernel's K chunks are 16 channels (chunks_total counts (32 channels, tap))
    const long long blocks = (M / (16 * tw)) * (N / (tw == 32 ? 64 : 128)) * groups;
    const bool r8 = blocks < half_rows_below();
    int sk = split_k;
    if (sk <= 0) {      // split-K over whole chunks until one workgroup per CU exists (each slice keeps >= 64 channels)
        sk = 1;
        while (blocks * sk < min_blocks && cc / (sk * 2) >= 4 && sk < 16) sk *= 2;
        // one more halving of K towards two workgroups per CU, but only while a slice keeps >= 16 16-channel chunks: the fixed cost
        // of a workgroup (first halo, epilogue, its share of the reduce) is ~1.5 chunks (measured B = 12: 5.07 -> 5.02 ms, B = 8 / 1
        // unchanged, B = 4 +0.6 %; with 8 chunks B = 4 loses 2.5 %)
        if (blocks * sk < 2 * min_blocks && cc / (sk * 2) >= 16 && sk < 16) sk *= 2;
        // split_k < 0 (the training step): slices by occupancy only.  The accuracy-driven slices exist for the eval forward's parity bar
        // on rollout inputs (error vs fp64 within ~2.5x of torch CPU's blocked fp32 GEMM); without them a chain is what the fp32 MFMA
        // pipe's own chain is (rms error 1.9e-8 of sum |terms| at K = 9216 against the pipe's 3.0e-8, tools/diag/split_precision.hip),
        // and training pays for every slice twice (partial sums + their reduce launch, and the BatchNorm statistics the epilogue
        // can only take from a launch that writes final values)
        if (split_k == 0) sk = chain_bounded_split(sk, cc, 9, M, N, groups);
    }
    if (sk > cc) sk = cc;
    const int per = (int)nbp_cdiv(cc, sk);
    p.split_k = (int)nbp_cdiv(cc, per); p.chunks_per_split = per;
    const bool r8t = r8 || half_rows_for_tail((M / (16 * tw)) * (N / (tw == 32 ? 64 : 128)) * groups * p.split_k);
    // (full-height plans of 32-pixel-wide layers with N % 128 == 0 run 8 x 32 pixels x 128 channels: its own tile id, so that a
    // per-layer timing table and a profile's kernel rows name the same instantiation)
    p.tile = r8t ? NBP_TILE_SPLIT_HALO_R8 : (tw == 32 && N % 128 == 0) ? NBP_TILE_SPLIT_HALO_128 : NBP_TILE_SPLIT_HALO_64;
    return p;
}

int nbp_amax_launch(const float* x, long long n, unsigned* amax_inout, hipStream_t st) {
    if (n <= 0) return 0;
    amax_kernel<<<min(nbp_ew_grid(n / 4 + 1, 256), 1024), 256, 0, st>>>(x, n, nullptr, 1, amax_inout, AMAX_WORDS);
    return nbp_launch_status();
}

// Returns NBP_E_SHAPE for layers the kernel does not take.
int nbp_conv_split_launch_g(const ConvOperandsSplit& o, const ConvOperandsSplit* o2, int C0, int C1, int ups, int B, int H, int W,
                            int ksize, int N, int relu, int split_k, void* ws, size_t ws_bytes, hipStream_t st,
                            float* const* pool_out, int* pooled, const ConvHead* head, int* headed, double* bn_part, int* bn_rows) {
    const int groups = o2 ? 2 : 1;
    if (bn_rows) *bn_rows = 0;
    NBP_RETURN_IF(!o.src0 || !o.planes || !o.scale || !o.shift || !o.out || !o.amax0 || !o.wamax, NBP_E_ARG);
    NBP_RETURN_IF(o2 && (!o2->src0 || !o2->planes || !o2->scale || !o2->shift || !o2->out || !o2->amax0 || !o2->wamax), NBP_E_ARG);
    NBP_RETURN_IF(B < 1 || H < 1 || W < 1 || ksize != 3, NBP_E_ARG);
    NBP_RETURN_IF(C0 < 32 || C0 % 32 || C1 < 0 || C1 % 32 || N < 64 || N % 64, NBP_E_SHAPE);
    NBP_RETURN_IF(C1 > 0 && (!o.src1 || !o.amax1 || (o2 && (!o2->src1 || !o2->amax1))), NBP_E_ARG);
    NBP_RETURN_IF(ups && ((H | W) & 1), NBP_E_SHAPE);
    SplitArgs a;
    for (int g = 0; g < 2; ++g) {
        const ConvOperandsSplit& s = (g && o2) ? *o2 : o;
        a.g[g] = SplitOps{s.src0, s.src1, s.planes, s.scale, s.shift, s.out, s.amax0, C1 ? s.amax1 : nullptr, s.wamax, s.amax_out, nullptr, nullptr, {nullptr, nullptr}, nullptr, nullptr, s.psi0};
    }
    NBP_RETURN_IF((o.psi0 || (o2 && o2->psi0)) && ups, NBP_E_SHAPE);       // (psi rides in the plain kernel's staging only)
    a.C0 = C0; a.C1 = C1; a.ups = ups ? 1 : 0;
    a.H = H; a.W = W; a.Hs = ups ? H / 2 : H; a.Ws = ups ? W / 2 : W;
    a.N = N; a.relu = relu; a.groups = groups;
    a.M = (long long)B * H * W;
    const long long b0 = (long long)B * a.Hs * a.Ws * C0 * 4, b1 = (long long)B * a.Hs * a.Ws * C1 * 4;
    const long long bw = (long long)(C0 + C1) * 9 * N * 4;
    NBP_RETURN_IF(b0 >= (1ll << 31) || b1 >= (1ll << 31) || bw >= (1ll << 31), NBP_E_SHAPE);   // 32-bit buffer offsets
    a.bytes0 = (unsigned)b0; a.bytes1 = C1 ? (unsigned)b1 : (unsigned)b0; a.bytesw = (unsigned)bw;
    const bool have_up = ups && o.planes_up && o.wamax_up && (!o2 || (o2->planes_up && o2->wamax_up));
    const ConvPlan p = nbp_plan_conv_split(a.M, N, (C0 + C1) / 32 * 9, split_k, groups, H, W, ksize, have_up ? 1 : 0);
    NBP_RETURN_IF(p.tile != NBP_TILE_SPLIT_HALO_64 && p.tile != NBP_TILE_SPLIT_HALO_128 && p.tile != NBP_TILE_SPLIT_UP && p.tile != NBP_TILE_SPLIT_HALO_R8 &&
                  p.tile != NBP_TILE_SPLIT_UP_R8, NBP_E_SHAPE);
    const bool ph = p.tile == NBP_TILE_SPLIT_UP || p.tile == NBP_TILE_SPLIT_UP_R8;
    const bool r8 = p.tile == NBP_TILE_SPLIT_HALO_R8 || p.tile == NBP_TILE_SPLIT_UP_R8;
    const bool wide = p.tile == NBP_TILE_SPLIT_HALO_128;      // 8 x 32 pixels x 128 channels
    const int th = (r8 || wide) ? 8 : 16;
    a.split_k = p.split_k; a.chunks_per_split = p.chunks_per_split;
    a.chunks_total = (C0 + C1) / 16;
    const int tw = ph ? split_tile_width(H / 2, W / 2, N, ksize) : split_tile_width(H, W, N, ksize);
    if (ph) {
        NBP_RETURN_IF(C1 != 0, NBP_E_SHAPE);
        for (int g = 0; g < groups; ++g) {
            const ConvOperandsSplit& sg = (g && o2) ? *o2 : o;
            a.g[g].planes = sg.planes_up; a.g[g].wamax = sg.wamax_up;
        }
        const long long bwu = (long long)C0 * 16 * N * 4;
        NBP_RETURN_IF(bwu >= (1ll << 31), NBP_E_SHAPE);
        a.bytesw = (unsigned)bwu;
    }
    {
        const long long ptiles = a.M / (ph ? 4 : 1) / (th * tw), nbk = N / ((tw == 32 && !wide) ? 64 : 128);      // = the launch's grid
        const long long tiles = ptiles * nbk;
        // bytes that cross the fabric: mode 1 = 8 x weights + activations, mode 2 = weights + min(nbk, 8) x activations
        const double wb = (double)(C0 + C1) * (ph ? 16 : 9) * N * 4, ab = (double)a.M / (ups ? 4 : 1) * (C0 + C1) * 4;
        const bool fits2 = nbk >= 8 ? nbk % 8 == 0 : (8 % nbk == 0 && ptiles % (8 / nbk) == 0);
        // (measured, B = 8: mode 2 cuts the fabric bytes of the 16-pixel-wide levels 3x -- 114 -> 39 MB and 266 -> 82 MB per launch --
        // and raises them when the two sides are comparable, hence the factor 3; run time is the same either way)
        const bool heavy = 7.0 * wb > ((nbk < 8 ? nbk : 8) - 1) * ab * 3.0;
        int mode = tiles >= 8 ? 1 : 0;      // (no run-time difference between the modes on this kernel: chosen by fabric bytes)
        if (fits2 && heavy && tiles >= 8) mode = 2;
        a.xcd_remap = mode;
    }
    a.partial = nullptr;
    if (p.split_k > 1) {
        NBP_RETURN_IF(!ws || ws_bytes < (size_t)groups * p.split_k * a.M * N * sizeof(float), NBP_E_WS);
        a.partial = (float*)ws;
    }
    // the max-pool that follows an encoder block rides in the epilogue when the launch writes final values (no split-K)
    static const int allow_pool = nbp_tune_int("NBP_CONV_POOL", 1);
    const bool with_pool = allow_pool && pool_out && pool_out[0] && (groups == 1 || pool_out[1]) && p.split_k == 1 && !ph && !ups &&
                           !((H | W) & 1);
    // ... and in the split-K reduce otherwise (one problem per launch: the encoder's layers)
    const bool pool_in_reduce = allow_pool && pool_out && pool_out[0] && groups == 1 && p.split_k > 1 && !ph && !ups && !((H | W) & 1);
    if (pooled) *pooled = with_pool || pool_in_reduce;
    if (with_pool || pool_in_reduce)
        for (int g = 0; g < groups; ++g) a.g[g].pool_out = pool_out[g];
    // the one-channel sigmoid head that consumes this layer alone (Final2) rides in the epilogue instead of the layer's own store
    static const int allow_head = nbp_tune_int("NBP_CONV_HEAD", 1);
    const bool with_head = allow_head && head && head->w && head->scale && head->shift && head->out && groups == 1 && p.split_k == 1 &&
                           !ph && tw == 32 && N == 64 && relu;
    if (headed) *headed = with_head;
    if (with_head) { a.g[0].head_w = head->w; a.g[0].head_ss[0] = head->scale; a.g[0].head_ss[1] = head->shift; a.g[0].head_out = head->out; }
    // training: the BatchNorm partials ride in the epilogue of the full-height tiles when the launch writes final values
    if (bn_part && bn_rows && p.split_k == 1 && !r8 && groups == 1) {
        a.g[0].bn_part = bn_part;
        *bn_rows = (int)(a.M / (16 * tw));               // pixel tiles (x 4 parities for the up_conv form: the same count)
        if (wide) {                                      // 8 x 32 pixels x 128 channels (see below): twice the pixel tiles
            *bn_rows = (int)(a.M / 256);
            return launch_h2<32, 2, 4, false, true>(a, st, p.tile);
        }
        // (up_conv layers: the two-parity form here too -- the one-parity 16-row form carried 136 B of scratch per lane)
        return ph ? (tw == 32 ? launch_h2<32, 2, 2, true, true, true>(a, st, p.tile) : launch_h2<16, 2, 4, true, true>(a, st, p.tile))
                  : (tw == 32 ? launch_h2<32, 4, 2, false, true>(a, st, p.tile) : launch_h2<16, 2, 4, false, true>(a, st, p.tile));
    }
    // up_conv layers on full-height 32-pixel-wide tiles: both column parities in one workgroup of half the height (same workgroup
    // count, one staged halo for two parities).  The 16-pixel-wide levels keep one parity per workgroup (the two-parity form measured
    // 7 % slower per launch there: 619 against 577 us at B = 24); round 3's one-parity 16 x 32 form (120 B of scratch per lane) is gone.
    int rc = (ph && !r8 && tw == 32) ? launch_h2<32, 2, 2, true, false, true>(a, st, p.tile)
           : r8 ? (ph ? (tw == 32 ? launch_h2<32, 2, 2, true>(a, st, p.tile) : launch_h2<16, 1, 4, true>(a, st, p.tile))
                      : (tw == 32 ? launch_h2<32, 2, 2, false>(a, st, p.tile) : launch_h2<16, 1, 4, false>(a, st, p.tile)))
           : ph ? launch_h2<16, 2, 4, true>(a, st, p.tile)
                // 32-pixel-wide levels: 8 x 32 pixels x 128 channels where the layer has them (round 5) -- the same workgroup count, MFMAs
                // per stage and sums in the same order as 16 x 32 x 64, but a 10 x 34 halo staged per 128 output channels instead of an
                // 18 x 34 one per 64: 44 % less halo traffic, splitting and LDS writes per output (every N % 128 == 0 layer 1.4 - 2.7 %
                // faster at B = 24, the forward 9.46 -> 9.38 ms; profiles/r05/tile_8x32x128.txt)
                : (tw == 32 ? (wide ? launch_h2<32, 2, 4, false>(a, st, p.tile) : launch_h2<32, 4, 2, false>(a, st, p.tile))
                            : launch_h2<16, 2, 4, false>(a, st, p.tile));
    if (rc) return rc;
    if (p.split_k > 1) {
        const long long MN = a.M * N;
        NBP_RETURN_IF(MN >= (1ll << 31), NBP_E_SHAPE);
        dim3 grid((unsigned)min(nbp_cdiv(MN / 4, 256 * 4), 1024ll), (unsigned)groups);     // >= 4 float4s per thread
        if (pool_in_reduce)
            splitk_reduce_pool_kernel<<<(unsigned)min(nbp_cdiv(MN / 16, 256), 1024ll), 256, 0, st>>>((const float*)ws, p.split_k, (unsigned)MN,
                                                                                                    (unsigned)N, H, W, a.g[0], relu);
        else
        splitk_reduce_split_kernel<<<grid, 256, 0, st>>>((const float*)ws, p.split_k, (unsigned)MN, (unsigned)N, a.g[0], a.g[1], relu);
        rc = nbp_launch_status();
    }
    return rc;
}

// wamax_out: device word that receives max |w * scale| (float bits); the planes are scaled by 2^(14 - floor(log2 max))
int nbp_pack_conv_weight_split_launch(const float* w_oihw, int N, int C, int ksize, const float* scale_or_null, int c_off,
                                      int c_total, void* dst, unsigned* wamax_out, hipStream_t st) {
    NBP_RETURN_IF(!w_oihw || !dst || !wamax_out, NBP_E_ARG);
    NBP_RETURN_IF(ksize != 1 && ksize != 3, NBP_E_ARG);
    NBP_RETURN_IF(N < 1 || C < 1 || c_off != 0 || C > c_total || c_total % 32, NBP_E_SHAPE);   // one scale per layer: one call
    const long long total = (long long)N * C * ksize * ksize;
    hipError_t e = hipMemsetAsync(wamax_out, 0, sizeof(unsigned), st);
    if (e != hipSuccess) return (int)e;
    amax_kernel<<<min(nbp_ew_grid(total, 256), 256), 256, 0, st>>>(w_oihw, total, scale_or_null, (long long)C * ksize * ksize, wamax_out, 1u);
    pack_conv_weight_h2_kernel<<<nbp_ew_grid(total, 256), 256, 0, st>>>(w_oihw, N, C, ksize * ksize, scale_or_null, c_off, wamax_out,
                                                                       (unsigned short*)dst);
    return nbp_launch_status();
}

// Planes of the four parity filters of an up_conv layer: [parity][chunk][4 taps][plane][k half][N][8 fp16], one max |w| for all.
int nbp_pack_upconv_weight_split_launch(const float* w_oihw, int N, int C, void* dst, unsigned* wamax_out, hipStream_t st) {
    NBP_RETURN_IF(!w_oihw || !dst || !wamax_out, NBP_E_ARG);
    NBP_RETURN_IF(N < 1 || C < 16 || C % 16, NBP_E_SHAPE);
    hipError_t e = hipMemsetAsync(wamax_out, 0, sizeof(unsigned), st);
    if (e != hipSuccess) return (int)e;
    const long long NC = (long long)N * C;
    pack_upconv_h2_kernel<<<min(nbp_ew_grid(NC, 256), 256), 256, 0, st>>>(w_oihw, N, C, wamax_out, nullptr);
    pack_upconv_h2_kernel<<<nbp_ew_grid(NC, 256), 256, 0, st>>>(w_oihw, N, C, wamax_out, (unsigned short*)dst);
    return nbp_launch_status();
}

// ---- attention gates: q = relu([g | x] W + b) as one 1x1 GEMM over K = 2 C on the split scheme
int nbp_pack_gate_weight_split_launch(const float* wg, const float* scale_g, const float* wx, const float* scale_x, int N, int C,
                                      void* dst, unsigned* wamax_out, hipStream_t st) {
    NBP_RETURN_IF(!wg || !wx || !dst || !wamax_out, NBP_E_ARG);
    NBP_RETURN_IF(N < 32 || N % 32 || C < 32 || C % 32, NBP_E_SHAPE);
    hipError_t e = hipMemsetAsync(wamax_out, 0, sizeof(unsigned), st);
    if (e != hipSuccess) return (int)e;
    const long long total = (long long)N * C;
    const int grid = min(nbp_ew_grid(total, 256), 256);
    amax_kernel<<<grid, 256, 0, st>>>(wg, total, scale_g, C, wamax_out, 1u);
    amax_kernel<<<grid, 256, 0, st>>>(wx, total, scale_x, C, wamax_out, 1u);
    pack_conv_weight_h2_kernel<<<nbp_ew_grid(total, 256), 256, 0, st>>>(wg, N, C, 1, scale_g, 0, wamax_out, (unsigned short*)dst);
    pack_conv_weight_h2_kernel<<<nbp_ew_grid(total, 256), 256, 0, st>>>(wx, N, C, 1, scale_x, C, wamax_out, (unsigned short*)dst);
    return nbp_launch_status();
}

// planes: nbp_pack_gate_weight_split_launch; amax0 / amax1: the 64-word slots of the two sources; out [M][N] fp32
int nbp_gate1x1_split_launch_g(const ConvOperandsSplit& o, const ConvOperandsSplit* o2, int C, long long M, int N, int relu,
                               hipStream_t st, const GatePsi* psi, int* fused) {
    const int groups = o2 ? 2 : 1;
    NBP_RETURN_IF(!o.src0 || !o.src1 || !o.planes || !o.scale || !o.shift || !o.out || !o.amax0 || !o.amax1 || !o.wamax, NBP_E_ARG);
    NBP_RETURN_IF(o2 && (!o2->src0 || !o2->src1 || !o2->planes || !o2->out || !o2->amax0 || !o2->amax1 || !o2->wamax), NBP_E_ARG);
    NBP_RETURN_IF(C < 64 || C % 64 || N < 32 || N % 32 || M < 1, NBP_E_SHAPE);
    const long long b0 = M * C * 4, bw = 2ll * C * N * 4;
    NBP_RETURN_IF(b0 >= (1ll << 31) || bw >= (1ll << 31), NBP_E_SHAPE);
    GateArgs a;
    for (int g = 0; g < 2; ++g) {
        const ConvOperandsSplit& s = (g && o2) ? *o2 : o;
        a.g[g] = SplitOps{s.src0, s.src1, s.planes, s.scale, s.shift, s.out, s.amax0, s.amax1, s.wamax, nullptr, nullptr, nullptr, {nullptr, nullptr}, nullptr};
        const int gi = g < groups ? g : 0;
        a.wpsi[g] = psi ? psi->wpsi[gi] : nullptr; a.st[g] = psi ? psi->st[gi] : nullptr; a.gated[g] = psi ? psi->gated[gi] : nullptr;
        a.gated_amax[g] = psi ? psi->gated_amax[gi] : nullptr;
    }
    a.psi_only = psi ? psi->psi_only : 0;
    a.C = C; a.C1 = C; a.N = N; a.relu = relu; a.M = M; a.bytes0 = (unsigned)b0; a.bytesw = (unsigned)bw; a.groups = groups;
    // 128-channel blocks only when they alone fill the chip twice; otherwise more, narrower workgroups
    int bn = N % 128 == 0 ? 128 : (N % 64 == 0 ? 64 : 32);
    // ... unless 128 columns are the whole gate and its psi tail is on offer: the tail then rides in the epilogue (bn == N below) and
    // its own launch goes (level 4 below B = 8: -0.8 % of the B = 1 forward, -1.1 % at B = 4)
    static const int allow_psi = nbp_tune_int("NBP_GATE_PSI", 1);
    const bool psi_wants_128 = allow_psi && psi && N == 128 && psi->wpsi[0] && psi->st[0] && psi->gated[0];
    if (bn == 128 && nbp_cdiv(M, 128) * (N / 128) * groups < 512 && !psi_wants_128) bn = 64;
    // the gate's tail (psi, x * psi) runs in the epilogue when a workgroup holds every column of its pixels
    const bool with_psi = allow_psi && psi && bn == N && psi->wpsi[0] && psi->st[0] && psi->gated[0] &&
                          (groups == 1 || (psi->wpsi[1] && psi->st[1] && psi->gated[1]));
    if (fused) *fused = with_psi;
    dim3 grid((unsigned)nbp_cdiv(M, 128), (unsigned)(N / bn), (unsigned)groups);
    const size_t smem = 2 * (size_t)8 * bn * 16;
    {
        char nm[64];
        snprintf(nm, sizeof(nm), "gate1x1_h2_kernel<%d, %s>", bn / 32, with_psi ? "true" : "false");
        nbp_note_kernel_symbol(NBP_TILE_SPLIT_GATE, nm);
    }
    if (with_psi) {
        if (bn == 128) gate1x1_h2_kernel<4, true><<<grid, 256, smem, st>>>(a);
        else if (bn == 64) gate1x1_h2_kernel<2, true><<<grid, 256, smem, st>>>(a);
        else gate1x1_h2_kernel<1, true><<<grid, 256, smem, st>>>(a);
    } else if (bn == 128) gate1x1_h2_kernel<4, false><<<grid, 256, smem, st>>>(a);
    else if (bn == 64) gate1x1_h2_kernel<2, false><<<grid, 256, smem, st>>>(a);
    else gate1x1_h2_kernel<1, false><<<grid, 256, smem, st>>>(a);
    return nbp_launch_status();
}

// Data gradient of an up_conv layer (x2 nearest upsample + 3 x 3 convolution from C to N channels) in parity form: dx [B,H,W,C] at the
// LOW resolution from dy [B,2H,2W,N]; planes / wamax from nbp_pack_upconv_weight_split_dgrad; amax_in = the 64-word max-|dy| slot.
// split_k <= 0: slices by occupancy.  NBP_E_SHAPE when the low-resolution image does not tile.
int nbp_upconv_dgrad_split_launch(const float* dy, int N, int B, int H, int W, const void* planes, const unsigned* wamax,
                                  const unsigned* amax_in, int C, const float* scale, const float* shift, float* out, unsigned* amax_out,
                                  int split_k, void* ws, size_t ws_bytes, hipStream_t st) {
    NBP_RETURN_IF(!dy || !planes || !wamax || !amax_in || !scale || !shift || !out, NBP_E_ARG);
    NBP_RETURN_IF(B < 1 || N < 16 || N % 16 || C < 64 || C % 64, NBP_E_SHAPE);
    const int tw = split_tile_width(H, W, C, 3);
    NBP_RETURN_IF(!tw, NBP_E_SHAPE);
    SplitArgs a;
    for (int g = 0; g < 2; ++g)
        a.g[g] = SplitOps{dy, nullptr, planes, scale, shift, out, amax_in, nullptr, wamax, amax_out, nullptr, nullptr, {nullptr, nullptr}, nullptr, nullptr};
    a.C0 = N; a.C1 = 0; a.ups = 0; a.H = H; a.W = W; a.Hs = 2 * H; a.Ws = 2 * W; a.N = C; a.relu = 0; a.groups = 1;
    a.M = (long long)B * H * W;
    const long long b0 = (long long)B * 4 * H * W * N * 4, bw = 64ll * N * C;
    NBP_RETURN_IF(b0 >= (1ll << 31) || bw >= (1ll << 31), NBP_E_SHAPE);
    a.bytes0 = (unsigned)b0; a.bytes1 = (unsigned)b0; a.bytesw = (unsigned)bw;
    a.chunks_total = 4 * (N / 16);
    const long long blocks = (a.M / ((tw == 32 ? 8 : 16) * tw)) * (C / (tw == 32 ? 64 : 128));
    int sk = split_k;
    if (sk <= 0) { sk = 1; while (blocks * sk < 256 && a.chunks_total / (sk * 2) >= 4 && sk < 16) sk *= 2; }
    if (sk > a.chunks_total) sk = a.chunks_total;
    a.chunks_per_split = (int)nbp_cdiv(a.chunks_total, sk);
    a.split_k = (int)nbp_cdiv(a.chunks_total, a.chunks_per_split);
    a.xcd_remap = blocks >= 8 ? 1 : 0;
    a.partial = nullptr;
    if (a.split_k > 1) {
        NBP_RETURN_IF(!ws || ws_bytes < (size_t)a.split_k * a.M * C * sizeof(float), NBP_E_WS);
        a.partial = (float*)ws;
    }
    // (8 x 32-pixel tiles on the 32-pixel-wide levels: the 16-row one-parity form carries 124 B of scratch per lane in its staging
    // section and measured the same, 520.9 against 518-521 maps/s for the training step)
    int rc = tw == 32 ? launch_h2<32, 2, 2, true, false, false, true>(a, st, NBP_TILE_SPLIT_UP_DGRAD)
                      : launch_h2<16, 2, 4, true, false, false, true>(a, st, NBP_TILE_SPLIT_UP_DGRAD);
    if (rc) return rc;
    if (a.split_k > 1) {
        const long long MN = a.M * C;
        NBP_RETURN_IF(MN >= (1ll << 31), NBP_E_SHAPE);
        dim3 grid((unsigned)min(nbp_cdiv(MN / 4, 256 * 4), 1024ll), 1u);
        splitk_reduce_split_kernel<<<grid, 256, 0, st>>>((const float*)ws, a.split_k, (unsigned)MN, (unsigned)C, a.g[0], a.g[1], 0);
        rc = nbp_launch_status();
    }
    return rc;
}

// out [M][N] = src [M][C] W + shift (optionally scale / ReLU): a 1x1 convolution on the split scheme through the gates' kernel with
// ONE source (training: the attention gates' W_g / W_x layers and their data gradients ran on the fp32 pipe -- compute-bound there
// on the deep levels, and padded from 32 to 64 output channels on level 2).  planes: nbp_pack_conv_weight_split with ksize = 1
// ([chunk of 16][hi|lo][k half][N][8]); C % 32 == 0, N % 32 == 0.
int nbp_conv1x1_split_launch(const float* src, int C, long long M, const void* planes, const unsigned* wamax, const unsigned* amax_in,
                             int N, const float* scale, const float* shift, int relu, float* out, hipStream_t st) {
    NBP_RETURN_IF(!src || !planes || !wamax || !amax_in || !scale || !shift || !out, NBP_E_ARG);
    NBP_RETURN_IF(C < 32 || C % 32 || N < 32 || N % 32 || M < 1, NBP_E_SHAPE);
    const long long b0 = M * C * 4, bw = (long long)C * N * 4;
    NBP_RETURN_IF(b0 >= (1ll << 31) || bw >= (1ll << 31) || M * N * 4 >= (1ll << 33), NBP_E_SHAPE);
    GateArgs a;
    for (int g = 0; g < 2; ++g) {
        a.g[g] = SplitOps{src, nullptr, planes, scale, shift, out, amax_in, nullptr, wamax, nullptr, nullptr, nullptr, {nullptr, nullptr}, nullptr, nullptr};
        a.wpsi[g] = nullptr; a.st[g] = nullptr; a.gated[g] = nullptr; a.gated_amax[g] = nullptr;
    }
    a.C = C; a.C1 = 0; a.N = N; a.relu = relu; a.M = M; a.bytes0 = (unsigned)b0; a.bytesw = (unsigned)bw; a.groups = 1;
    int bn = N % 128 == 0 ? 128 : (N % 64 == 0 ? 64 : 32);
    if (bn == 128 && nbp_cdiv(M, 128) * (N / 128) < 512) bn = 64;
    dim3 grid((unsigned)nbp_cdiv(M, 128), (unsigned)(N / bn), 1u);
    const size_t smem = 2 * (size_t)8 * bn * 16;
    if (bn == 128) gate1x1_h2_kernel<4, false><<<grid, 256, smem, st>>>(a);
    else if (bn == 64) gate1x1_h2_kernel<2, false><<<grid, 256, smem, st>>>(a);
    else gate1x1_h2_kernel<1, false><<<grid, 256, smem, st>>>(a);
    return nbp_launch_status();
}

// Partial sums [splits][9][C0 + C1][N] of the 3x3 weight gradient (the caller reduces them: wgrad_reduce_kernel, nbp_train.hip).
// amax3 = 3 x 64 zeroed words of scratch: max |src0|, max |src1|, max |dY| are computed here.
int nbp_wgrad_split_launch(const float* src0, int C0, const float* src1, int C1, int ups, int B, int H, int W, const float* dy, int N,
                           int n_tiles, int splits, unsigned* amax3, const unsigned* amax0_in, const unsigned* amax1_in,
                           const unsigned* amaxy_in, float* part, hipStream_t st) {
    const int Hs = ups ? H / 2 : H, Ws = ups ? W / 2 : W;
    const long long b0 = (long long)B * Hs * Ws * C0 * 4, b1 = (long long)B * Hs * Ws * C1 * 4, by = (long long)B * H * W * N * 4;
    NBP_RETURN_IF(b0 >= (1ll << 31) || b1 >= (1ll << 31) || by >= (1ll << 31), NBP_E_SHAPE);
    hipError_t e = hipMemsetAsync(amax3, 0, 3 * AMAX_WORDS * sizeof(unsigned), st);
    if (e != hipSuccess) return (int)e;
    // a max the caller already has (the forward's input slot, the data gradient's) saves a pass over the tensor
    int rc = amax0_in ? 0 : nbp_amax_launch(src0, b0 / 4, amax3, st);
    if (!rc && C1 && !amax1_in) rc = nbp_amax_launch(src1, b1 / 4, amax3 + AMAX_WORDS, st);
    if (!rc && !amaxy_in) rc = nbp_amax_launch(dy, by / 4, amax3 + 2 * AMAX_WORDS, st);
    if (rc) return rc;
    WgradSplitArgs a;
    a.src0 = src0; a.src1 = src1 ? src1 : src0; a.C0 = C0; a.C1 = C1; a.ups = ups ? 1 : 0; a.H = H; a.W = W; a.Hs = Hs; a.Ws = Ws;
    a.dy = dy; a.N = N; a.bytes0 = (unsigned)b0; a.bytes1 = C1 ? (unsigned)b1 : (unsigned)b0; a.bytesy = (unsigned)by;
    a.co_tiles = N / 64; a.n_tiles = n_tiles; a.splits = splits;
    a.amax0 = amax0_in ? amax0_in : amax3; a.amax1 = amax1_in ? amax1_in : amax3 + AMAX_WORDS;
    a.amaxy = amaxy_in ? amaxy_in : amax3 + 2 * AMAX_WORDS; a.part = part;
    const bool wide = W % 32 == 0 && H % 2 == 0;              // 2 x 32 tiles, else 4 x 16 (nbp_wgrad_split_ok)
    const int smem = wide ? 4 * (4 * 34 * 64) + 4 * (64 * 64) : 4 * (6 * 18 * 64) + 4 * (64 * 64);
    static bool attr_set = false;
    if (!attr_set) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_split_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                4 * (4 * 34 * 64) + 4 * (64 * 64));

        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_split_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    4 * (6 * 18 * 64) + 4 * (64 * 64));
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    dim3 grid((unsigned)(((C0 + C1) / 64) * (N / 64)), (unsigned)splits);
    if (wide) wgrad_split_kernel<32><<<grid, 256, smem, st>>>(a);
    else wgrad_split_kernel<16><<<grid, 256, smem, st>>>(a);
    return nbp_launch_status();
}

extern "C" int nbp_pack_conv_weight_split(const float* w_oihw, int N, int C, int ksize, const float* scale_or_null,
                                          int c_off, int c_total, void* dst_planes, void* wamax_out, void* stream) {
    NBP_ENTER();
    return nbp_pack_conv_weight_split_launch(w_oihw, N, C, ksize, scale_or_null, c_off, c_total, dst_planes, (unsigned*)wamax_out,
                                             (hipStream_t)stream);
}

// Planes of the data-gradient convolution of a 3x3 layer with weights w [N][C][3][3]: dx = conv3x3(dy, w') with
// w'[c][n][tap] = w[n][c][8 - tap] -- C output rows, N input channels padded to c_total (flip + permute + pack in one launch).
extern "C" int nbp_pack_conv_weight_split_dgrad(const float* w_oihw, int N, int C, int c_total, void* dst_planes, void* wamax_out,
                                                void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(!w_oihw || !dst_planes || !wamax_out, NBP_E_ARG);
    NBP_RETURN_IF(N < 1 || C < 1 || N > c_total || c_total % 32, NBP_E_SHAPE);
    hipStream_t st = (hipStream_t)stream;
    const long long total = (long long)N * C * 9;
    hipError_t e = hipMemsetAsync(wamax_out, 0, sizeof(unsigned), st);
    if (e != hipSuccess) return (int)e;
    amax_kernel<<<min(nbp_ew_grid(total, 256), 256), 256, 0, st>>>(w_oihw, total, nullptr, 9ll * C, (unsigned*)wamax_out, 1u);
    pack_conv_weight_h2_kernel<<<nbp_ew_grid(total, 256), 256, 0, st>>>(w_oihw, C, N, 9, nullptr, 0, (const unsigned*)wamax_out,
                                                                       (unsigned short*)dst_planes, 1);
    return nbp_launch_status();
}

// All weight packs of a training step: descs_dev = n PrepackDesc records on the device (layout in include/nbp_hip.h), wamax words
// zeroed here.  N % 16 == 0 and C % 16 == 0 for every record (the plane layouts).
extern "C" int nbp_prepack_weights_split(const void* descs_dev, int n, void* wamax_words, void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(!descs_dev || !wamax_words || n < 1 || n > 4096, NBP_E_ARG);
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(wamax_words, 0, (size_t)n * sizeof(unsigned), st);
    if (e != hipSuccess) return (int)e;
    prepack_amax_kernel<<<dim3(64, (unsigned)n), 256, 0, st>>>((const PrepackDesc*)descs_dev);
    prepack_pack_kernel<<<dim3(256, (unsigned)n), 256, 0, st>>>((const PrepackDesc*)descs_dev);
    return nbp_launch_status();
}
extern "C" int nbp_prepack_desc_bytes(void) { return (int)sizeof(PrepackDesc); }

// The training step's forms of the two packs above (round 5: a 3x3 layer cost six launches per step for its weights -- memset, max,
// pack, twice): `_prezeroed` takes a max-|w| word the caller has already zeroed (one fill per forward for all layers) and
// `_dgrad_known` the word the forward's pack of the same weights left, since the data-gradient planes hold the same values.
extern "C" int nbp_pack_conv_weight_split_prezeroed(const float* w_oihw, int N, int C, int ksize, int c_total, void* dst_planes,
                                                    void* wamax_zeroed, void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(!w_oihw || !dst_planes || !wamax_zeroed, NBP_E_ARG);
    NBP_RETURN_IF((ksize != 1 && ksize != 3) || N < 1 || C < 1 || C > c_total || c_total % 32, NBP_E_SHAPE);
    hipStream_t st = (hipStream_t)stream;
    const long long total = (long long)N * C * ksize * ksize;
    amax_kernel<<<min(nbp_ew_grid(total, 256), 256), 256, 0, st>>>(w_oihw, total, nullptr, (long long)C * ksize * ksize, (unsigned*)wamax_zeroed, 1u);
    pack_conv_weight_h2_kernel<<<nbp_ew_grid(total, 256), 256, 0, st>>>(w_oihw, N, C, ksize * ksize, nullptr, 0, (const unsigned*)wamax_zeroed,
                                                                       (unsigned short*)dst_planes);
    return nbp_launch_status();
}
extern "C" int nbp_pack_conv_weight_split_dgrad_known(const float* w_oihw, int N, int C, int c_total, void* dst_planes,
                                                      const void* wamax_known, void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(!w_oihw || !dst_planes || !wamax_known, NBP_E_ARG);
    NBP_RETURN_IF(N < 1 || C < 1 || N > c_total || c_total % 32, NBP_E_SHAPE);
    const long long total = (long long)N * C * 9;
    pack_conv_weight_h2_kernel<<<nbp_ew_grid(total, 256), 256, 0, (hipStream_t)stream>>>(w_oihw, C, N, 9, nullptr, 0, (const unsigned*)wamax_known,
                                                                                        (unsigned short*)dst_planes, 1);
    return nbp_launch_status();
}

extern "C" int nbp_amax_f32(const float* x, long long n, void* amax_inout, void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(!x || !amax_inout || n < 0, NBP_E_ARG);
    return nbp_amax_launch(x, n, (unsigned*)amax_inout, (hipStream_t)stream);
}

extern "C" size_t nbp_conv_split_workspace_bytes(int B, int H, int W, int N, int split_k) {
    const int sk = split_k == 1 ? 0 : (split_k <= 0 ? 16 : split_k);
    return 256 + (size_t)sk * B * H * W * N * sizeof(float);
}

// Workspace of ONE layer as the planner will run it (split_k = 0): 256 B for the max-|x| slot + the split-K slices it plans.
// C = C0 + C1; ups = the layer reads its input through the x2 upsample (parity kernels when the low-resolution image tiles).
extern "C" size_t nbp_conv_split_planned_workspace_bytes_k(int B, int H, int W, int C, int N, int ups, int split_k, int* split_k_out);
extern "C" size_t nbp_conv_split_planned_workspace_bytes(int B, int H, int W, int C, int N, int ups, int* split_k_out) {
    return nbp_conv_split_planned_workspace_bytes_k(B, H, W, C, N, ups, 0, split_k_out);
}
// ... for a given split_k request (0: planner with the accuracy bound, < 0: planner by occupancy only, > 0: as given)
extern "C" size_t nbp_conv_split_planned_workspace_bytes_k(int B, int H, int W, int C, int N, int ups, int split_k, int* split_k_out) {
    if (B < 1 || H < 1 || W < 1 || C < 32 || N < 1) return 0;
    const ConvPlan p = nbp_plan_conv_split((long long)B * H * W, N, C / 32 * 9, split_k, 1, H, W, 3, ups ? 1 : 0);
    if (split_k_out) *split_k_out = p.tile ? p.split_k : 0;
    const int sk = (p.tile && p.split_k > 1) ? p.split_k : 0;
    return 256 + (size_t)sk * B * H * W * N * sizeof(float);
}

static int conv3x3_split_impl(const float* src0, int C0, const float* src1, int C1, int ups, int B, int H, int W,
                              const void* w_planes, const void* wamax, int N, const float* scale, const float* shift,
                              int relu, float* out, const void* amax_in_or_null, void* amax_out_or_null, int split_k,
                              void* ws, size_t ws_bytes, void* stream, double* bn_part, int* bn_rows) {
    NBP_RETURN_IF(!ws || ws_bytes < 256 || !src0, NBP_E_WS);
    hipStream_t st = (hipStream_t)stream;
    const unsigned* amax = (const unsigned*)amax_in_or_null;
    if (!amax) {                // max |x| over both sources, into the head of the workspace
        unsigned* slot = (unsigned*)ws;
        hipError_t e = hipMemsetAsync(slot, 0, 256, st);
        if (e != hipSuccess) return (int)e;
        const long long hw = (long long)B * (ups ? H / 2 : H) * (ups ? W / 2 : W);
        int rc = nbp_amax_launch(src0, hw * C0, slot, st);
        if (!rc && C1 > 0) { NBP_RETURN_IF(!src1, NBP_E_ARG); rc = nbp_amax_launch(src1, hw * C1, slot, st); }
        if (rc) return rc;
        amax = slot;
    }
    ConvOperandsSplit o{src0, src1, w_planes, scale, shift, out, amax, amax, (const unsigned*)wamax, (unsigned*)amax_out_or_null,
                        nullptr, nullptr};
    return nbp_conv_split_launch_g(o, nullptr, C0, C1, ups, B, H, W, 3, N, relu, split_k, (char*)ws + 256, ws_bytes - 256, st, nullptr,
                                   nullptr, nullptr, nullptr, bn_part, bn_rows);
}
extern "C" int nbp_conv3x3_split_f32(const float* src0, int C0, const float* src1, int C1, int ups, int B, int H, int W,
                                     const void* w_planes, const void* wamax, int N, const float* scale, const float* shift,
                                     int relu, float* out, const void* amax_in_or_null, void* amax_out_or_null, int split_k,
                                     void* ws, size_t ws_bytes, void* stream) {
    NBP_ENTER();
    return conv3x3_split_impl(src0, C0, src1, C1, ups, B, H, W, w_planes, wamax, N, scale, shift, relu, out, amax_in_or_null,
                              amax_out_or_null, split_k, ws, ws_bytes, stream, nullptr, nullptr);
}
// The same, and the column sums of the output and of its squares for the BatchNorm that follows (training): bn_part receives
// *bn_rows rows of [2][N] doubles (at most nbp_conv_bn_part_rows(B, H, W) of them); *bn_rows = 0 when this launch did not take them
// (split-K or half-height tiles: the caller's BatchNorm then reads the tensor itself).
extern "C" int nbp_conv_bn_part_rows(int B, int H, int W) { return (int)((long long)B * H * W / 256); }
extern "C" int nbp_conv3x3_split_bn_f32(const float* src0, int C0, const float* src1, int C1, int ups, int B, int H, int W,
                                        const void* w_planes, const void* wamax, int N, const float* scale, const float* shift,
                                        int relu, float* out, const void* amax_in_or_null, void* amax_out_or_null, int split_k,
                                        void* ws, size_t ws_bytes, double* bn_part, int* bn_rows, void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(!bn_part || !bn_rows || ((uintptr_t)bn_part & 7), NBP_E_ARG);
    return conv3x3_split_impl(src0, C0, src1, C1, ups, B, H, W, w_planes, wamax, N, scale, shift, relu, out, amax_in_or_null,
                              amax_out_or_null, split_k, ws, ws_bytes, stream, bn_part, bn_rows);
}

// ---- weight gradient of an up_conv layer in parity form (training; wgrad_up_split_kernel)
static void wgrad_up_plan(int B, int Hs, int Ws, int C, int N, int* n_tiles, int* splits) {
    *n_tiles = (int)((long long)B * Hs * Ws / 64);
    const long long pairs = (long long)(C / 64) * (N / 64) * 4;       // x four parities
    long long sp = nbp_cdiv(768, pairs);                              // three workgroups per CU
    if (sp > *n_tiles) sp = *n_tiles;
    if (sp > 1024) sp = 1024;
    if (sp < 1) sp = 1;
    *splits = (int)sp;
}
static bool wgrad_up_ok(int Hs, int Ws, int C, int N) {
    return C >= 64 && C % 64 == 0 && N >= 64 && N % 64 == 0 && ((Ws % 32 == 0 && Hs % 2 == 0) || (Ws % 16 == 0 && Hs % 4 == 0));
}
extern "C" size_t nbp_upconv_wgrad_split_workspace_bytes(int B, int Hs, int Ws, int C, int N) {
    if (B < 1 || !wgrad_up_ok(Hs, Ws, C, N)) return 0;
    int nt, sp;
    wgrad_up_plan(B, Hs, Ws, C, N, &nt, &sp);
    return 1024 + (size_t)sp * 16 * C * N * sizeof(float);
}
// dW [N][C][3][3] of an up_conv layer from its low-resolution input x [B,Hs,Ws,C] and dy [B,2Hs,2Ws,N]; amax_x / amax_y: the tensors'
// 64-word max-|.| slots (required).  NBP_E_SHAPE when the low-resolution image does not tile (the caller takes nbp_conv_wgrad_split_f32).
extern "C" int nbp_upconv_wgrad_split_f32(const float* x, int C, int B, int Hs, int Ws, const float* dy, int N, float* dw,
                                          const void* amax_x, const void* amax_y, void* ws, size_t ws_bytes, void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(!x || !dy || !dw || !amax_x || !amax_y || !ws || B < 1, NBP_E_ARG);
    NBP_RETURN_IF(!wgrad_up_ok(Hs, Ws, C, N), NBP_E_SHAPE);
    const long long b0 = (long long)B * Hs * Ws * C * 4, by = (long long)B * 4 * Hs * Ws * N * 4;
    NBP_RETURN_IF(b0 >= (1ll << 31) || by >= (1ll << 31), NBP_E_SHAPE);
    int n_tiles, splits;
    wgrad_up_plan(B, Hs, Ws, C, N, &n_tiles, &splits);
    NBP_RETURN_IF(ws_bytes < 1024 + (size_t)splits * 16 * C * N * sizeof(float), NBP_E_WS);
    hipStream_t st = (hipStream_t)stream;
    WgradSplitArgs a;
    a.src0 = x; a.src1 = x; a.C0 = C; a.C1 = 0; a.ups = 1; a.H = 2 * Hs; a.W = 2 * Ws; a.Hs = Hs; a.Ws = Ws;
    a.dy = dy; a.N = N; a.bytes0 = (unsigned)b0; a.bytes1 = (unsigned)b0; a.bytesy = (unsigned)by;
    a.co_tiles = N / 64; a.n_tiles = n_tiles; a.splits = splits;
    a.amax0 = (const unsigned*)amax_x; a.amax1 = a.amax0; a.amaxy = (const unsigned*)amax_y;
    a.part = (float*)((char*)(((uintptr_t)ws + 255) / 256 * 256));
    const bool wide = Ws % 32 == 0 && Hs % 2 == 0;
    constexpr int smem32 = 4 * (3 * 34 * 64) + 4 * (64 * 64), smem16 = 4 * (5 * 18 * 64) + 4 * (64 * 64);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_up_split_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, smem32);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_up_split_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, smem16);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    dim3 grid((unsigned)((C / 64) * (N / 64)), (unsigned)splits, 4u);
    if (wide) wgrad_up_split_kernel<32><<<grid, 256, smem32, st>>>(a);
    else wgrad_up_split_kernel<16><<<grid, 256, smem16, st>>>(a);
    int rc = nbp_launch_status();
    if (rc) return rc;
    wgrad_up_reduce_kernel<<<nbp_ew_grid((long long)N * C * 9, 256), 256, 0, st>>>(a.part, splits, C, N, dw);
    return nbp_launch_status();
}

// ---- weight gradient of a 1x1 layer on the split scheme (wgrad_1x1_split_kernel): partial sums [splits][C][N] into `part`;
// the caller reduces them (nbp_train.hip).  C % 64 == 0, N % 4 == 0.
int nbp_wgrad_1x1_split_launch(const float* x, int C, long long M, const float* dy, int N, int n_tiles, int splits, const unsigned* amax_x,
                               const unsigned* amax_y, float* part, hipStream_t st) {
    NBP_RETURN_IF(M * C * 4 >= (1ll << 31) || M * N * 4 >= (1ll << 31), NBP_E_SHAPE);
    WgradSplitArgs a;
    NBP_RETURN_IF(M % 64 != 0, NBP_E_SHAPE);                       // (every level of the network; the kernel counts pixels as H x 64)
    a.src0 = x; a.src1 = x; a.C0 = C; a.C1 = 0; a.ups = 0; a.H = (int)(M / 64); a.W = 64; a.Hs = a.H; a.Ws = a.W;
    a.dy = dy; a.N = N; a.bytes0 = (unsigned)(M * C * 4); a.bytes1 = a.bytes0; a.bytesy = (unsigned)(M * N * 4);
    a.co_tiles = (N + 63) / 64; a.n_tiles = n_tiles; a.splits = splits;
    a.amax0 = amax_x; a.amax1 = amax_x; a.amaxy = amax_y; a.part = part;
    dim3 grid((unsigned)((C / 64) * a.co_tiles), (unsigned)splits);
    wgrad_1x1_split_kernel<<<grid, 256, 8 * 64 * 64, st>>>(a);
    return nbp_launch_status();
}

// ---- data gradient of an up_conv layer in parity form (training; conv3x3_halo_h2_kernel<..., DG>)
// planes: 32 N C fp16 ([4 parities x N / 16 chunks][4 taps][hi|lo][k half][C][8]) from the layer's own weight [N][C][3][3]
extern "C" int nbp_pack_upconv_weight_split_dgrad(const float* w_oihw, int N, int C, void* dst_planes, void* wamax_out, void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(!w_oihw || !dst_planes || !wamax_out, NBP_E_ARG);
    NBP_RETURN_IF(N < 16 || N % 16 || C < 1, NBP_E_SHAPE);
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(wamax_out, 0, sizeof(unsigned), st);
    if (e != hipSuccess) return (int)e;
    const long long NC = (long long)N * C;
    pack_upconv_dgrad_h2_kernel<<<min(nbp_ew_grid(NC, 256), 256), 256, 0, st>>>(w_oihw, N, C, (unsigned*)wamax_out, nullptr);
    pack_upconv_dgrad_h2_kernel<<<nbp_ew_grid(NC, 256), 256, 0, st>>>(w_oihw, N, C, (unsigned*)wamax_out, (unsigned short*)dst_planes);
    return nbp_launch_status();
}
// workspace of nbp_upconv3x3_split_dgrad_f32 (its split-K slices, by occupancy)
extern "C" size_t nbp_upconv_split_dgrad_workspace_bytes(int B, int H, int W, int N, int C) {
    if (B < 1 || H < 1 || W < 1 || N < 16 || C < 64) return 0;
    const int tw = split_tile_width(H, W, C, 3);
    if (!tw) return 0;
    const long long M = (long long)B * H * W, blocks = (M / ((tw == 32 ? 8 : 16) * tw)) * (C / (tw == 32 ? 64 : 128));
    const int chunks = 4 * (N / 16);
    int sk = 1;
    while (blocks * sk < 256 && chunks / (sk * 2) >= 4 && sk < 16) sk *= 2;
    return 256 + (sk > 1 ? (size_t)sk * M * C * sizeof(float) : 0);
}
// dx [B,H,W,C] (the layer's LOW-resolution input gradient) from dy [B,2H,2W,N]; scale / shift: C ones / zeros (the epilogue's affine)
extern "C" int nbp_upconv3x3_split_dgrad_f32(const float* dy, int N, int B, int H, int W, const void* planes, const void* wamax, int C,
                                             const float* scale, const float* shift, float* dx, const void* amax_in, void* amax_out_or_null,
                                             void* ws, size_t ws_bytes, void* stream) {
    NBP_ENTER();
    return nbp_upconv_dgrad_split_launch(dy, N, B, H, W, planes, (const unsigned*)wamax, (const unsigned*)amax_in, C, scale, shift, dx,
                                         (unsigned*)amax_out_or_null, 0, ws ? (char*)ws + 256 : nullptr, ws_bytes >= 256 ? ws_bytes - 256 : 0,
                                         (hipStream_t)stream);
}

// out [M][N] = src [M][C] (1x1 convolution) on the split scheme; w_planes / wamax from nbp_pack_conv_weight_split(ksize = 1),
// or from nbp_pack_conv1x1_weight_split_dgrad for the data gradient (dx = dy W^T); amax_in: the 64-word max-|src| slot (required).
extern "C" int nbp_conv1x1_split_f32(const float* src, int C, long long M, const void* w_planes, const void* wamax, int N,
                                     const float* scale, const float* shift, int relu, float* out, const void* amax_in, void* stream) {
    NBP_ENTER();
    return nbp_conv1x1_split_launch(src, C, M, w_planes, (const unsigned*)wamax, (const unsigned*)amax_in, N, scale, shift, relu, out,
                                    (hipStream_t)stream);
}
// planes of w^T for the data gradient of a 1x1 layer w [N][C]: a 1x1 convolution from N (dy's channels) to C
extern "C" int nbp_pack_conv1x1_weight_split_dgrad(const float* w_nc, int N, int C, void* dst_planes, void* wamax_out, void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(!w_nc || !dst_planes || !wamax_out, NBP_E_ARG);
    NBP_RETURN_IF(N < 32 || N % 32 || C < 32 || C % 32, NBP_E_SHAPE);
    hipStream_t st = (hipStream_t)stream;
    const long long total = (long long)N * C;
    hipError_t e = hipMemsetAsync(wamax_out, 0, sizeof(unsigned), st);
    if (e != hipSuccess) return (int)e;
    amax_kernel<<<min(nbp_ew_grid(total, 256), 256), 256, 0, st>>>(w_nc, total, nullptr, C, (unsigned*)wamax_out, 1u);
    // (transposed form of the pack: rows = C output channels, K = N input channels, w'[c][n] = w[n][c])
    pack_conv_weight_h2_kernel<<<nbp_ew_grid(total, 256), 256, 0, st>>>(w_nc, C, N, 1, nullptr, 0, (const unsigned*)wamax_out,
                                                                       (unsigned short*)dst_planes, 1);
    return nbp_launch_status();
}

extern "C" int nbp_pack_upconv_weight_split(const float* w_oihw, int N, int C, void* dst_planes, void* wamax_out, void* stream) {
    NBP_ENTER();
    return nbp_pack_upconv_weight_split_launch(w_oihw, N, C, dst_planes, (unsigned*)wamax_out, (hipStream_t)stream);
}

static int upconv3x3_split_impl(const float* src, int C, int B, int H, int W, const void* planes_up, const void* wamax_up,
                                int N, const float* scale, const float* shift, int relu, float* out,
                                const void* amax_in_or_null, void* amax_out_or_null, int split_k, void* ws, size_t ws_bytes,
                                void* stream, double* bn_part, int* bn_rows) {
    NBP_RETURN_IF(!ws || ws_bytes < 256 || !src || !planes_up || !wamax_up, NBP_E_WS);
    NBP_RETURN_IF((H | W) & 1, NBP_E_SHAPE);
    hipStream_t st = (hipStream_t)stream;
    const unsigned* amax = (const unsigned*)amax_in_or_null;
    if (!amax) {
        unsigned* slot = (unsigned*)ws;
        hipError_t e = hipMemsetAsync(slot, 0, 256, st);
        if (e != hipSuccess) return (int)e;
        const int rc = nbp_amax_launch(src, (long long)B * (H / 2) * (W / 2) * C, slot, st);
        if (rc) return rc;
        amax = slot;
    }
    // the plain planes are not needed when the parity kernels take the layer; NBP_E_SHAPE otherwise
    {
        const int t = nbp_plan_conv_split((long long)B * H * W, N, C / 32 * 9, split_k, 1, H, W, 3, 1).tile;
        NBP_RETURN_IF(t != NBP_TILE_SPLIT_UP && t != NBP_TILE_SPLIT_UP_R8, NBP_E_SHAPE);
    }
    ConvOperandsSplit o{src, nullptr, planes_up, scale, shift, out, amax, amax, (const unsigned*)wamax_up, (unsigned*)amax_out_or_null,
                        planes_up, (const unsigned*)wamax_up};
    return nbp_conv_split_launch_g(o, nullptr, C, 0, 1, B, H, W, 3, N, relu, split_k, (char*)ws + 256, ws_bytes - 256, st, nullptr, nullptr,
                                   nullptr, nullptr, bn_part, bn_rows);
}
extern "C" int nbp_upconv3x3_split_f32(const float* src, int C, int B, int H, int W, const void* planes_up, const void* wamax_up,
                                       int N, const float* scale, const float* shift, int relu, float* out,
                                       const void* amax_in_or_null, void* amax_out_or_null, int split_k, void* ws, size_t ws_bytes,
                                       void* stream) {
    NBP_ENTER();
    return upconv3x3_split_impl(src, C, B, H, W, planes_up, wamax_up, N, scale, shift, relu, out, amax_in_or_null, amax_out_or_null, split_k,
                                ws, ws_bytes, stream, nullptr, nullptr);
}
extern "C" int nbp_upconv3x3_split_bn_f32(const float* src, int C, int B, int H, int W, const void* planes_up, const void* wamax_up,
                                          int N, const float* scale, const float* shift, int relu, float* out,
                                          const void* amax_in_or_null, void* amax_out_or_null, int split_k, void* ws, size_t ws_bytes,
                                          double* bn_part, int* bn_rows, void* stream) {
    NBP_ENTER();
    NBP_RETURN_IF(!bn_part || !bn_rows || ((uintptr_t)bn_part & 7), NBP_E_ARG);
    return upconv3x3_split_impl(src, C, B, H, W, planes_up, wamax_up, N, scale, shift, relu, out, amax_in_or_null, amax_out_or_null, split_k,
                                ws, ws_bytes, stream, bn_part, bn_rows);
}
