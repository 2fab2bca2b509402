// nbp_first_conv.h -- Conv1.conv.0 (5 -> 64 channels, 3x3, NCHW fp32 input) on the matrix cores.
// The VALU version spends 45 x 64 FMAs per pixel in one lane; here a workgroup owns an 8 x 32 pixel tile: the five
// 10 x 34 input planes (halo) and the [46][64] weight matrix (K = 5 * 9 = 45, padded to 46) sit in LDS and each wave
// computes 2 image rows x 64 channels with 23 x 4 v_mfma_f32_32x32x2_f32 (exact fp32 products and sums).
// Operands are swapped (A = weights, B = pixels) so that a lane ends up with four consecutive output channels of one
// pixel: 16-B fp32 stores, or 8-B stores after rounding to bf16 for the bf16 path (ref nbp_model.py:8-21, :66).
#pragma once
#include "common.h"

// RELU = false (training): the layer's own output, conv + bias, for the train-mode BatchNorm that follows
template <typename OutT, typename Store4, bool RELU = true>
__device__ __forceinline__ void conv_first_mfma_body(const float* __restrict__ x, int B, int H, int W,
                                                     const float* __restrict__ w, const float* __restrict__ scale,
                                                     const float* __restrict__ shift, OutT* __restrict__ out, Store4 store4,
                                                     unsigned* __restrict__ amax_out = nullptr) {
    constexpr int HW_ = 34, PLANE = 10 * HW_;            // 340 halo pixels per input channel
    __shared__ __attribute__((aligned(16))) float hal[5 * PLANE];
    __shared__ __attribute__((aligned(16))) float wl[46 * 64];   // [k][co], row 45 = 0
    constexpr int TRB = 144;                             // bytes per pixel row of the epilogue's transpose image
    __shared__ __attribute__((aligned(16))) char tr[4 * 64 * TRB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = W >> 5, tiles_y = H >> 3;
    const int n_tiles = B * tiles_y * tiles_x;
    for (int i = tid; i < 46 * 64; i += 256) {           // the weights stay in LDS for every tile of this workgroup
        const int co = i & 63, k = i >> 6;
        wl[i] = k < 45 ? w[co * 45 + k] : 0.f;
    }
    const int kh = lane >> 5, ln = lane & 31;
    float mx = 0.f;                                      // max |out| of this workgroup's tiles (split path: the next layer's scale)
    // the halo of the NEXT tile is fetched into registers before this tile's MFMAs and epilogue (round 4: a tile was load ->
    // barrier -> MFMA -> store with nothing in flight across tiles; a workgroup walks 12 tiles at B = 24)
    constexpr int NPRE = (5 * PLANE + 255) / 256;
    float pre[NPRE];
    auto fetch = [&](int tile_id) {
        int tile = tile_id;
        const int tx = tile % tiles_x; tile /= tiles_x;
        const int ty = tile % tiles_y;
        const int b = tile / tiles_y;
        const int y0 = ty * 8, x0 = tx * 32;
#pragma unroll
        for (int q = 0; q < NPRE; ++q) {
            const int i = tid + 256 * q;
            const int ci = i / PLANE, r = i - ci * PLANE;
            const int hy = r / HW_, hx = r - hy * HW_;
            const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
            const bool ok = i < 5 * PLANE && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
            pre[q] = ok ? x[((long long)(b * 5 + ci) * H + yy) * W + xx] : 0.f;
        }
    };
    if ((int)blockIdx.x < n_tiles) fetch(blockIdx.x);
  for (int tile_id = blockIdx.x; tile_id < n_tiles; tile_id += gridDim.x) {
    int tile = tile_id;
    const int tx = tile % tiles_x; tile /= tiles_x;
    const int ty = tile % tiles_y;
    const int b = tile / tiles_y;
    const int y0 = ty * 8, x0 = tx * 32;
    __syncthreads();                                     // previous tile's MFMA reads of `hal` are done
#pragma unroll
    for (int q = 0; q < NPRE; ++q) {
        const int i = tid + 256 * q;
        if (i < 5 * PLANE) hal[i] = pre[q];
    }
    __syncthreads();
    if (tile_id + (int)gridDim.x < n_tiles) fetch(tile_id + gridDim.x);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const float* hp = hal + (2 * wave) * HW_ + ln;       // halo pixel of (row 2 wave, column ln) for tap (0,0)
#pragma unroll
    for (int kk = 0; kk < 23; ++kk) {
        // k = 2 kk + kh: channel k / 9, tap k % 9 (k = 45 is the zero pad: weights row 45 = 0, any finite input)
        const int k0 = 2 * kk, k1 = 2 * kk + 1 < 45 ? 2 * kk + 1 : 44;
        const int off0 = (k0 / 9) * PLANE + ((k0 % 9) / 3) * HW_ + (k0 % 9) % 3;
        const int off1 = (k1 / 9) * PLANE + ((k1 % 9) / 3) * HW_ + (k1 % 9) % 3;
        const int off = kh ? off1 : off0;
        const float p0 = hp[off], p1 = hp[off + HW_];
        const float w0 = wl[(2 * kk + kh) * 64 + ln], w1 = wl[(2 * kk + kh) * 64 + 32 + ln];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0, p0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1, p0, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0, p1, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1, p1, acc[1][1], 0, 0, 0);
    }
    // D[n][m]: lane = pixel ln of row 2 wave + i; registers 4 rq .. 4 rq + 3 = channels j*32 + 8 rq + 4 kh + (0..3).
    // A lane holds 16-byte (8-byte for bf16) pieces of one pixel, the 32 lanes of a half wave 32 different pixels: stored directly,
    // every store instruction touches 32 lines with 32 (16) bytes each and the L2 takes one request per piece.  The wave transposes
    // 128 bytes per pixel through LDS instead (fp32: the 32 channels of block j, two passes; bf16: all 64 channels; pixel rows
    // padded to 144 bytes: conflict-free) and stores 16 bytes per lane, 8 lanes per pixel: whole 128-byte lines.
    constexpr int PASSES = sizeof(OutT) == 4 ? 2 : 1;
    char* const trw = tr + wave * (64 * TRB);
#pragma unroll
    for (int pass = 0; pass < PASSES; ++pass) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (PASSES == 2 && j != pass) continue;
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int n = j * 32 + 8 * rq + 4 * kh;
                    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + n);
                    const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + n);
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float u = acc[i][j][4 * rq + e] * sc[e] + sh[e]; v[e] = RELU ? fmaxf(u, 0.f) : u; mx = fmaxf(mx, fabsf(v[e])); }
                    store4(reinterpret_cast<OutT*>(trw + (i * 32 + ln) * TRB) + (PASSES == 2 ? n - 32 * pass : n), v);
                }
            }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = it * 64 + lane, p = idx >> 3, slot = idx & 7;          // pixel p = row p / 32, column p % 32
            const f32x4 v = *reinterpret_cast<const f32x4*>(trw + p * TRB + slot * 16);
            const long long m = ((long long)b * H + y0 + 2 * wave + (p >> 5)) * W + x0 + (p & 31);
            *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(out + m * 64) + pass * 128 + slot * 16) = v;
        }
    }
  }
    if (amax_out) {     // one atomic per wave, spread over the 64 words of the tensor's slot (nbp_split.hip)
#pragma unroll
        for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        if (lane == 0) atomicMax(amax_out + ((blockIdx.x * 4u + wave) & 63u), __float_as_uint(mx));
    }
}
