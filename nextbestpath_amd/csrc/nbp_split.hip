// nbp_split.hip -- the fp32 3x3 convolutions of NBP on the bf16 matrix pipe, by EXACT operand splitting.
//
// Same operator as conv3x3_halo_f32_kernel (nbp_conv.hip): conv_block / up_conv of next_best_path/networks/nbp_model.py:8-40
// on fp32 NHWC activations with fp32 weights, fp32 accumulation, fp32 epilogue.  The matrix products run on
// v_mfma_f32_32x32x16_bf16 (32 cycles for 16 k) instead of v_mfma_f32_32x32x2_f32 (8 x 64 cycles for 16 k):
//   every fp32 operand x is cut into three bf16 pieces of 8 significand bits each by truncation,
//       hi = x & 0xFFFF0000,  mid = (x - hi) & 0xFFFF0000,  lo = (x - hi) - mid      (both subtractions are exact),
//   so x = hi + mid + lo EXACTLY (3 x 8 = 24 bits), every bf16 x bf16 product is exact in the fp32 accumulator, and
//       x w = hi hi + (hi mid + mid hi) + (mid mid + hi lo + lo hi)  +  [mid lo + lo mid + lo lo <= 2^-23 |x w|, dropped]
//   is six MFMAs of 32 cycles (192 cycles per 16 k against 512: 2.67 x the fp32 pipe's rate).  The dropped terms are
//   below one fp32 rounding of the product; measured against fp64 (tools/diag/split_precision.hip, K = 1152 / 9216):
//   rms error 2.6e-8 of sum|terms| for this kernel's order (small terms first), 2.8e-8 for the fp32 MFMA chain --
//   the path is as accurate as the fp32 one and keeps fp32 tensors everywhere outside the MFMA operands.
//   With the five small products in their own accumulator (DUAL, the default) the rms error is 1.0e-8: a third of the
//   fp32 pipe's.  tests/test_gpu_split.py holds both paths against fp64 layer by layer and for the whole network.
// Weights are split once at pack time (three bf16 planes); activations are split once per workgroup and chunk, on their
// way from global memory into LDS.  fp32 tensors everywhere outside the MFMA operands.
//
// Measured (MI355X, B = 4 maps of 256 x 256): 3x3 layers 235-270 TFLOP/s of fp32-equivalent work (1.4-1.6 PFLOP/s of issued
// bf16 MFMA work; the chip clocks near 2.0 GHz under this load) against 126-137 on the fp32 pipe; forward 3.55 ms vs 5.76.
// Variants measured and dropped: activations split per tap between ds_read and MFMA (fp32 halo DMA'd into LDS; five VALU
// instructions per MFMA, 6 % slower); fragment double-buffering across taps (no change: not ds_read-latency bound);
// without any weight DMA the kernel is only 9 % faster (not bound by the weight stream either).
#include "common.h"
#include "nbp_internal.h"
#include <cstdlib>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// two fp32 -> the packed (low element first) bf16 pairs of their hi / mid / lo pieces
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned b0 = __float_as_uint(x0), b1 = __float_as_uint(x1);
    const float r0 = x0 - __uint_as_float(b0 & 0xFFFF0000u);
    const float r1 = x1 - __uint_as_float(b1 & 0xFFFF0000u);
    const unsigned c0 = __float_as_uint(r0), c1 = __float_as_uint(r1);
    const float q0 = r0 - __uint_as_float(c0 & 0xFFFF0000u);
    const float q1 = r1 - __uint_as_float(c1 & 0xFFFF0000u);
    h = __builtin_amdgcn_perm(b1, b0, 0x07060302u);            // (b0 >> 16) | (b1 & 0xFFFF0000)
    m = __builtin_amdgcn_perm(c1, c0, 0x07060302u);
    l = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
}


// ------------------------------------------------------------------ 3x3 convolution from LDS-resident bf16 planes
// Workgroup = 8 x 32 pixels x 64 output channels (4 waves, each 2 image rows x 64 channels = 4 accumulator tiles).  The
// 10 x 34 pixel halo tile of a 16-channel chunk goes global -> registers -> (split once) -> three bf16 planes in LDS, and
// the tap loop is ds_read_b128 + MFMA only:
//   planes  [hi|mid|lo][k half][344 pixels][8 bf16]   33 KB   (a lane's fragment = 16 contiguous bytes; any 16-lane group
//                                                              of a ds_read_b128 covers 16 distinct slots: no conflicts)
//   weights [tap of a filter row][hi|mid|lo][k half][64 rows][8 bf16], one filter row (3 taps) per stage, two stages  36 KB
// 69 KB of LDS: two workgroups per CU.  One MFMA K step (16 channels) per tap; a stage is 3 taps = 72 MFMAs per wave
// between barriers, with the next stage's weights DMA'd (buffer_load ... lds) behind it.  The next chunk's halo loads are
// issued before the chunk's first stage and consumed (split + ds_write) after its last one.
// DUAL: the five small products accumulate apart from hi x hi and join it once at the end.
// TW = tile width: 32 (8 x 32 pixel tiles; a 32-pixel MFMA row block = one image row) or 16 (16 x 16 pixel tiles for the
// 16-pixel-wide levels; a row block = two image rows of 16, whose second half-group of a ds_read_b128 lands 2 slots off
// the conflict-free pattern: 18-pixel halo rows).
template <bool DUAL, int TW>
__global__ __launch_bounds__(256, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_halo_split16_kernel(IgemmArgs a) {
    int zs = blockIdx.z;        // split-K slice (of 16-channel chunks), then the group
    if (zs >= a.split_k) {
        zs -= a.split_k;
        a.src0 = a.g_src0; a.src1 = a.g_src1; a.wpk = a.g_wpk; a.scale = a.g_scale; a.shift = a.g_shift; a.out = a.g_out;
    }
    constexpr int TM = 2, TN = 2, BN = 64;
    constexpr int TH = 256 / TW, RPB = 32 / TW;               // tile height; image rows per 32-pixel row block
    constexpr int HW_ = TW + 2, HPIX = (TH + 2) * HW_;        // 10 x 34 = 340 / 18 x 18 = 324 halo pixels
    constexpr int RS = 344 * 16 + 64;                          // bytes between (plane, k half) regions (+64: ds_write banks)
    constexpr int HALO_BYTES = 6 * RS;
    constexpr int WB = 18 * 1024;                              // one stage: 3 taps x 3 planes x 2 k halves x 64 rows x 16 B
    constexpr int NF = 6;                                      // float4 pieces of the halo tile per thread (1360 / 256)
    extern __shared__ __attribute__((aligned(16))) char ldsb[];
    char* const halo = ldsb;
    char* const wbuf = ldsb + HALO_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = a.W / TW, tiles_y = a.H / TH;
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
    const int y0 = ty * TH, x0 = tx * TW;
    const int n0 = nt * BN;

    // halo staging: piece f = tid + 256 k is channels 4 (f & 3) .. + 3 of halo pixel f >> 2
    int hpix[NF], hdst[NF];
#pragma unroll
    for (int k = 0; k < NF; ++k) {
        const int f = tid + 256 * k;
        const int hr = f >> 2, q = f & 3;
        const int hy = hr / HW_, hx = hr - hy * HW_;
        const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
        const bool ok = hr < HPIX && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
        hpix[k] = ok ? (b * a.Hs + (yy >> a.ups)) * a.Ws + (xx >> a.ups) : -1;
        hdst[k] = hr < HPIX ? (q >> 1) * RS + hr * 16 + (q & 1) * 8 : -1;
    }
    const __amdgpu_buffer_rsrc_t rs0 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src0), 0, a.bytes0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src1 ? a.src1 : a.src0), 0, a.bytes1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wpk), 0, a.bytesw, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    const int c16_0 = a.C0 >> 4;

    u32x4 hreg[NF];
    auto load_halo = [&](int c) {       // 16-channel chunk c of the concatenated input
        const bool first = c < c16_0;
        const int Cs = first ? a.C0 : a.C1;
        const int cbase = (first ? c : c - c16_0) * 16 + (tid & 3) * 4;
#pragma unroll
        for (int k = 0; k < NF; ++k) {
            const unsigned off = hpix[k] >= 0 ? (unsigned)(hpix[k] * Cs + cbase) * 4u : OOB;
            hreg[k] = first ? __builtin_amdgcn_raw_buffer_load_b128(rs0, off, 0, 0) : __builtin_amdgcn_raw_buffer_load_b128(rs1, off, 0, 0);
        }
    };
    auto store_halo = [&]() {
#pragma unroll
        for (int k = 0; k < NF; ++k) {
            if (hdst[k] < 0) continue;
            const f32x4 v = __builtin_bit_cast(f32x4, hreg[k]);
            unsigned h0, m0, l0, h1, m1, l1;
            split_pair(v[0], v[1], h0, m0, l0);
            split_pair(v[2], v[3], h1, m1, l1);
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            *reinterpret_cast<u32x2*>(halo + hdst[k]) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(halo + 2 * RS + hdst[k]) = u32x2{m0, m1};
            *reinterpret_cast<u32x2*>(halo + 4 * RS + hdst[k]) = u32x2{l0, l1};
        }
    };
    // weight stage u = chunk * 3 + filter row: 18 DMA instructions of 64 rows x 16 B; wave w issues q = w, w + 4, ...
    auto issue_w = [&](int u) {
        const int c = u / 3, row = u - 3 * c;
        char* dst = wbuf + (u & 1) * WB;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int q = wave + 4 * k;
            if (q < 18) {
                const int tt = q / 6, r6 = q - 6 * tt;         // r6 = plane * 2 + k half
                const unsigned woff = (unsigned)(((((long long)c * 9 + row * 3 + tt) * 6 + r6) * a.N + n0 + lane) * 16);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr_t)(dst + q * 1024), 16, woff, 0, 0, 0);
            }
        }
    };

    f32x16 acc[TM][TN], accs[DUAL ? TM : 1][DUAL ? TN : 1];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[i][j][r] = 0.f;
                if constexpr (DUAL) accs[i][j][r] = 0.f;
            }

    const int khalf = lane >> 5;
    const int hbase = (TM * wave * RPB + ((lane & 31) / TW)) * HW_ + ((lane & 31) % TW);
    const char* const arow = halo + khalf * RS + hbase * 16;               // + plane * 2 RS + halo-row shift * 16
    const int brow = khalf * 1024 + (lane & 31) * 16;                      // + (tap * 6 + plane * 2) KB + N tile * 512

    // chunks_per_split / chunks_total are in 16-channel chunks here
    const int c_begin = zs * a.chunks_per_split;
    const int c_end = min(c_begin + a.chunks_per_split, a.chunks_total);
    if (c_begin < c_end) {
        load_halo(c_begin);
        issue_w(c_begin * 3);
        store_halo();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int c = c_begin; c < c_end; ++c) {
        const bool more = c + 1 < c_end;
        if (more) load_halo(c + 1);
#pragma unroll 1
        for (int row = 0; row < 3; ++row) {
            const int u = c * 3 + row;
            if (row < 2 || more) issue_w(u + 1);
            const char* Bt = wbuf + (u & 1) * WB + brow;
#pragma unroll
            for (int tt = 0; tt < 3; ++tt) {
                u32x4 xp[TM][3], wp[TN][3];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        xp[i][p] = *reinterpret_cast<const u32x4*>(arow + p * (2 * RS) + ((row + i * RPB) * HW_ + tt) * 16);
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        wp[j][p] = *reinterpret_cast<const u32x4*>(Bt + (tt * 6 + p * 2) * 1024 + j * 512);
                constexpr int PX[6] = {2, 0, 1, 1, 0, 0};
                constexpr int PW[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                for (int v = 0; v < 6; ++v)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int i = 0; i < TM; ++i) {
                            const bf16x8 xa = __builtin_bit_cast(bf16x8, xp[i][PX[v]]);
                            const bf16x8 wa = __builtin_bit_cast(bf16x8, wp[j][PW[v]]);
                            if constexpr (DUAL) {
                                if (v < 5) accs[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, wa, accs[i][j], 0, 0, 0);
                                else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, wa, acc[i][j], 0, 0, 0);
                            } else {
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, wa, acc[i][j], 0, 0, 0);
                            }
                        }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        if (more) {                     // every wave is past its last read of this chunk's planes
            store_halo();
            __syncthreads();
        }
    }

    const bool final_out = (a.split_k == 1);
    float* outp = final_out ? a.out : a.partial + (long long)blockIdx.z * a.M * a.N;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + j * 32 + (lane & 31);
        float sc = 1.f, sh = 0.f;
        if (final_out) { sc = a.scale[n]; sh = a.shift[n]; }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const long long mrow = ((long long)b * a.H + y0 + (TM * wave + i) * RPB) * a.W + x0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pb = (r & 3) + 8 * (r >> 2) + 4 * khalf;              // pixel of the 32-pixel row block
                const int px = (pb / TW) * a.W + (pb % TW);
                float v = acc[i][j][r];
                if constexpr (DUAL) v += accs[i][j][r];
                if (final_out) {
                    v = v * sc + sh;
                    if (a.relu) v = fmaxf(v, 0.f);
                }
                outp[(mrow + px) * a.N + n] = v;
            }
        }
    }
}

// planes for conv3x3_halo_split16_kernel: [chunk of 16 channels][tap][hi|mid|lo][k half][N][8 bf16]
__global__ void pack_conv_weight_split16_kernel(const float* __restrict__ w, int N, int C, int taps,
                                                const float* __restrict__ scale, int c_off, unsigned short* __restrict__ dst) {
    const long long total = (long long)N * C * taps;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int tap = (int)(i % taps);
        const long long t = i / taps;
        const int c = (int)(t % C);
        const int n = (int)(t / C);
        float v = w[i];
        if (scale) v *= scale[n];
        unsigned h, m, l;
        split_pair(v, 0.f, h, m, l);
        const int cg = c_off + c;
        const long long base = ((long long)(cg >> 4) * taps + tap) * 6 + ((cg >> 3) & 1);
        dst[((base + 0) * N + n) * 8 + (cg & 7)] = (unsigned short)h;
        dst[((base + 2) * N + n) * 8 + (cg & 7)] = (unsigned short)m;
        dst[((base + 4) * N + n) * 8 + (cg & 7)] = (unsigned short)l;
    }
}

__global__ __launch_bounds__(256) void splitk_reduce_split_kernel(const float* __restrict__ partial_all, int split_k,
                                                                  long long MN, int N, const float* scale0,
                                                                  const float* shift0, float* out0, const float* scale1,
                                                                  const float* shift1, float* out1, int relu) {
    const float* partial = partial_all + (long long)blockIdx.y * split_k * MN;
    const float* scale = blockIdx.y ? scale1 : scale0;
    const float* shift = blockIdx.y ? shift1 : shift0;
    float* out = blockIdx.y ? out1 : out0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < MN / 4; i += (long long)gridDim.x * blockDim.x) {
        f32x4 v = *reinterpret_cast<const f32x4*>(partial + i * 4);
        for (int s = 1; s < split_k; ++s) {
            const f32x4 u = *reinterpret_cast<const f32x4*>(partial + s * MN + i * 4);
            v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3];
        }
        const int n = (int)((i * 4) % N);
        const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + n);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = v[e] * sc[e] + sh[e];
            if (relu) v[e] = fmaxf(v[e], 0.f);
        }
        *reinterpret_cast<f32x4*>(out + i * 4) = v;
    }
}

template <bool DUAL, int TW>
int launch_split16(const IgemmArgs& a, hipStream_t st) {
    constexpr size_t smem = 6 * (size_t)(344 * 16 + 64) + 2 * (size_t)18 * 1024;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_split16_kernel<DUAL, TW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    dim3 grid((unsigned)(a.M / 256), (unsigned)(a.N / 64), (unsigned)(a.split_k * a.groups));
    conv3x3_halo_split16_kernel<DUAL, TW><<<grid, 256, smem, st>>>(a);
    return nbp_launch_status();
}

// tile width the layer runs with: 32 (8 x 32 pixel tiles), 16 (16 x 16) or 0 (not taken)
int split_tile_width(int H, int W, int N, int ksize) {
    if (ksize != 3 || N % 64) return 0;
    if (H >= 8 && H % 8 == 0 && W >= 32 && W % 32 == 0) return 32;
    if (H >= 16 && H % 16 == 0 && W >= 16 && W % 16 == 0) return 16;
    return 0;
}

}  // namespace

// tile == 0: the layer does not fit the split kernel (the caller runs the fp32 MFMA kernels on the fp32 pack)
ConvPlan nbp_plan_conv_split(long long M, int N, int chunks_total, int split_k, int groups, int H, int W, int ksize) {
    ConvPlan p{0, 1, chunks_total};
    static const int allow = [] { const char* e = getenv("NBP_SPLIT_HALO"); return e ? atoi(e) : 1; }();
    static const int min_blocks = [] { const char* e = getenv("NBP_SPLIT_MIN_BLOCKS"); return e ? atoi(e) : 512; }();
    if (!allow || !split_tile_width(H, W, N, ksize)) return p;
    const int cc = chunks_total / 9 * 2;      // the kernel's K chunks are 16 channels (chunks_total counts (32 channels, tap))
    const long long blocks = (M / 256) * (N / 64) * groups;
    int sk = split_k;
    if (sk <= 0) {      // split-K over whole chunks until two workgroups per CU exist (each slice keeps >= 64 channels)
        sk = 1;
        while (blocks * sk < min_blocks && cc / (sk * 2) >= 4 && sk < 16) sk *= 2;
    }
    if (sk > cc) sk = cc;
    const int per = (int)nbp_cdiv(cc, sk);
    p.tile = NBP_TILE_SPLIT_HALO_64; p.split_k = (int)nbp_cdiv(cc, per); p.chunks_per_split = per;
    return p;
}

// o.wpk / o2->wpk point at the split planes.  Returns NBP_E_SHAPE for layers the kernel does not take.
int nbp_conv_split_launch_g(const ConvOperands& o, const ConvOperands* o2, int C0, int C1, int ups, int B, int H, int W,
                            int ksize, int N, int relu, int split_k, void* ws, size_t ws_bytes, hipStream_t st) {
    const int groups = o2 ? 2 : 1;
    NBP_RETURN_IF(!o.src0 || !o.wpk || !o.scale || !o.shift || !o.out, NBP_E_ARG);
    NBP_RETURN_IF(o2 && (!o2->src0 || !o2->wpk || !o2->scale || !o2->shift || !o2->out), NBP_E_ARG);
    NBP_RETURN_IF(B < 1 || H < 1 || W < 1 || ksize != 3, NBP_E_ARG);
    NBP_RETURN_IF(C0 < 32 || C0 % 32 || C1 < 0 || C1 % 32 || N < 64 || N % 64, NBP_E_SHAPE);
    NBP_RETURN_IF(C1 > 0 && (!o.src1 || (o2 && !o2->src1)), NBP_E_ARG);
    NBP_RETURN_IF(ups && ((H | W) & 1), NBP_E_SHAPE);
    IgemmArgs a;
    a.src0 = o.src0; a.src1 = o.src1; a.C0 = C0; a.C1 = C1; a.cc0 = C0 / 32; a.ups = ups ? 1 : 0;
    a.H = H; a.W = W; a.Hs = ups ? H / 2 : H; a.Ws = ups ? W / 2 : W;
    a.taps = 9; a.wpk = o.wpk; a.N = N; a.scale = o.scale; a.shift = o.shift; a.relu = relu; a.out = o.out;
    a.groups = groups;
    a.g_src0 = o2 ? o2->src0 : nullptr; a.g_src1 = o2 ? o2->src1 : nullptr; a.g_wpk = o2 ? o2->wpk : nullptr;
    a.g_scale = o2 ? o2->scale : nullptr; a.g_shift = o2 ? o2->shift : nullptr; a.g_out = o2 ? o2->out : nullptr;
    a.M = (long long)B * H * W;
    const long long b0 = (long long)B * a.Hs * a.Ws * C0 * 4, b1 = (long long)B * a.Hs * a.Ws * C1 * 4;
    const long long bw = (long long)(C0 + C1) * 9 * N * 6;
    NBP_RETURN_IF(b0 >= (1ll << 31) || b1 >= (1ll << 31) || bw >= (1ll << 31), NBP_E_SHAPE);   // 32-bit buffer offsets
    a.bytes0 = (unsigned)b0; a.bytes1 = C1 ? (unsigned)b1 : (unsigned)b0; a.bytesw = (unsigned)bw;
    a.chunks_total = (C0 + C1) / 32 * 9;
    const ConvPlan p = nbp_plan_conv_split(a.M, N, a.chunks_total, split_k, groups, H, W, ksize);
    NBP_RETURN_IF(p.tile != NBP_TILE_SPLIT_HALO_64, NBP_E_SHAPE);
    a.split_k = p.split_k; a.chunks_per_split = p.chunks_per_split;
    {
        static const int forced = [] { const char* e = getenv("NBP_XCD_REMAP"); return e ? atoi(e) : -1; }();
        const long long tiles = (a.M / 256) * (N / 64);
        a.xcd_remap = forced >= 0 ? forced : (tiles >= 512 ? 1 : 0);
    }
    a.partial = nullptr;
    if (p.split_k > 1) {
        NBP_RETURN_IF(!ws || ws_bytes < (size_t)groups * p.split_k * a.M * N * sizeof(float), NBP_E_WS);
        a.partial = (float*)ws;
    }
    static const int dual = [] { const char* e = getenv("NBP_SPLIT_DUAL"); return e ? atoi(e) : 1; }();
    a.chunks_total = (C0 + C1) / 16;                            // the kernel counts 16-channel chunks
    const int tw = split_tile_width(H, W, N, ksize);
    int rc = tw == 32 ? (dual ? launch_split16<true, 32>(a, st) : launch_split16<false, 32>(a, st))
                      : (dual ? launch_split16<true, 16>(a, st) : launch_split16<false, 16>(a, st));
    if (rc) return rc;
    if (p.split_k > 1) {
        const long long MN = a.M * N;
        dim3 grid((unsigned)nbp_ew_grid(MN / 4, 256), (unsigned)groups);
        splitk_reduce_split_kernel<<<grid, 256, 0, st>>>((const float*)ws, p.split_k, MN, N, o.scale, o.shift, o.out,
                                                         a.g_scale, a.g_shift, a.g_out, relu);
        rc = nbp_launch_status();
    }
    return rc;
}

int nbp_pack_conv_weight_split_launch(const float* w_oihw, int N, int C, int ksize, const float* scale_or_null, int c_off,
                                      int c_total, void* dst, hipStream_t st) {
    NBP_RETURN_IF(!w_oihw || !dst, NBP_E_ARG);
    NBP_RETURN_IF(ksize != 1 && ksize != 3, NBP_E_ARG);
    NBP_RETURN_IF(N < 1 || C < 1 || c_off < 0 || c_off + C > c_total || c_total % 32, NBP_E_SHAPE);
    const long long total = (long long)N * C * ksize * ksize;
    pack_conv_weight_split16_kernel<<<nbp_ew_grid(total, 256), 256, 0, st>>>(w_oihw, N, C, ksize * ksize, scale_or_null, c_off,
                                                                            (unsigned short*)dst);
    return nbp_launch_status();
}

extern "C" int nbp_pack_conv_weight_split(const float* w_oihw, int N, int C, int ksize, const float* scale_or_null,
                                          int c_off, int c_total, void* dst_planes, void* stream) {
    NBP_ENTER();
    return nbp_pack_conv_weight_split_launch(w_oihw, N, C, ksize, scale_or_null, c_off, c_total, dst_planes, (hipStream_t)stream);
}

extern "C" size_t nbp_conv_split_workspace_bytes(int B, int H, int W, int N, int split_k) {
    if (split_k == 1) return 0;
    const int sk = split_k <= 0 ? 16 : split_k;
    return (size_t)sk * B * H * W * N * sizeof(float);
}

extern "C" int nbp_conv3x3_split_f32(const float* src0, int C0, const float* src1, int C1, int ups, int B, int H, int W,
                                     const void* w_planes, int N, const float* scale, const float* shift, int relu,
                                     float* out, int split_k, void* ws, size_t ws_bytes, void* stream) {
    NBP_ENTER();
    ConvOperands o{src0, src1, (const float*)w_planes, scale, shift, out};
    return nbp_conv_split_launch_g(o, nullptr, C0, C1, ups, B, H, W, 3, N, relu, split_k, ws, ws_bytes, (hipStream_t)stream);
}
