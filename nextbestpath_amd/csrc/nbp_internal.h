// nbp_internal.h -- declarations shared by the conv / forward translation units (not part of the C ABI).
#pragma once
#include "common.h"

typedef unsigned short bf16_t;   // bf16 storage

enum { NBP_TILE_AUTO = 0, NBP_TILE_128x128 = 1, NBP_TILE_256x64 = 2, NBP_TILE_256x32 = 3, NBP_TILE_128x64 = 4,
       NBP_TILE_64x128 = 5,
       NBP_TILE_HALO_128 = 6, NBP_TILE_HALO_64 = 7,
       NBP_TILE_HALO4_128 = 8, NBP_TILE_HALO4_64 = 9 };   // fp32 only: 4x32-pixel tiles   // 8x32-pixel halo-tile kernels (3x3 only), BN = 128 / 64
struct TileInfo { int bm, bn; };
struct ConvPlan { int tile; int split_k; int chunks_per_split; };

// fp32 path (nbp_conv.hip)
struct ConvOperands { const float* src0; const float* src1; const float* wpk; const float* scale; const float* shift; float* out; };
ConvPlan nbp_plan_conv(long long M, int N, int chunks_total, int tile, int split_k, int groups, int H = 0, int W = 0,
                       int ksize = 0);
int nbp_conv_igemm_launch_g(const ConvOperands& o, const ConvOperands* o2, int C0, int C1, int ups, int B, int H, int W,
                            int ksize, int N, int relu, int split_k, int tile, void* ws, size_t ws_bytes, hipStream_t st);

// bf16 path (nbp_bf16.hip); K chunks are 64 channels
struct ConvOperandsH { const bf16_t* src0; const bf16_t* src1; const bf16_t* wpk; const float* scale; const float* shift; bf16_t* out; };
ConvPlan nbp_plan_conv_bf16(long long M, int N, int chunks_total, int tile, int split_k, int groups, int H = 0, int W = 0,
                            int ksize = 0);
int nbp_conv_igemm_bf16_launch_g(const ConvOperandsH& o, const ConvOperandsH* o2, int C0, int C1, int ups, int B, int H,
                                 int W, int ksize, int N, int relu, int split_k, int tile, void* ws, size_t ws_bytes,
                                 hipStream_t st);
int nbp_conv_first_bf16_launch(const float* x_nchw, int B, int H, int W, const float* w_oihw, const float* scale,
                               const float* shift, bf16_t* out_nhwc, hipStream_t st);
int nbp_maxpool2_bf16_launch(const bf16_t* in, int B, int H, int W, int C, bf16_t* out, hipStream_t st);
int nbp_psi_gate_bf16_launch(const bf16_t* q, int F, const float* w_psi, const float* s_t2, const bf16_t* x, int C,
                             long long M, bf16_t* out, hipStream_t st);
int nbp_final_1x1_bf16_launch(const bf16_t* in, int B, int H, int W, int C, const float* w_oc, int n_out,
                              const float* scale, const float* shift, int sigmoid, float* out_nchw, hipStream_t st);
