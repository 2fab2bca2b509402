// nbp_internal.h -- declarations shared by the conv / forward translation units (not part of the C ABI).
#pragma once
#include "common.h"

typedef unsigned short bf16_t;   // bf16 storage

enum { NBP_TILE_AUTO = 0, NBP_TILE_128x128 = 1, NBP_TILE_256x64 = 2, NBP_TILE_256x32 = 3, NBP_TILE_128x64 = 4,
       NBP_TILE_64x128 = 5,
       NBP_TILE_HALO_128 = 6, NBP_TILE_HALO_64 = 7,
       NBP_TILE_HALO4_128 = 8, NBP_TILE_HALO4_64 = 9,
       NBP_TILE_SPLIT_HALO_64 = 10, NBP_TILE_SPLIT_UP = 11,
       NBP_TILE_HALO_UP_128 = 12, NBP_TILE_HALO_UP_64 = 13,
       NBP_TILE_RESERVED_14 = 14,   // (was the weights-in-registers bf16 kernel of round 3: measured slower, removed)
       // 10 / 11: nbp_split.hip, 16x32 / 16x16-pixel halo tiles on the fp16 matrix pipe (11: up_conv as parity convolutions);
       // 15 / 16: their 8-row forms (launches that would leave CUs idle); 12 / 13: the bf16 up_conv parity kernels
       NBP_TILE_SPLIT_HALO_R8 = 15, NBP_TILE_SPLIT_UP_R8 = 16,
       NBP_TILE_SPLIT_UP_DGRAD = 17,    // nbp_split.hip: data gradient of an up_conv layer in parity form (training)
       NBP_TILE_SPLIT_HALO_128 = 18,
       NBP_TILE_SPLIT_GATE = 19 };      // nbp_split.hip: the attention gates' 1x1 GEMM over [g | x] (gate1x1_h2_kernel)
      //  // nbp_split.hip: 8 x 32 pixels x 128 channels (32-pixel-wide layers with N % 128 == 0)
struct TileInfo { int bm, bn; };
struct ConvPlan { int tile; int split_k; int chunks_per_split; };

// fp32 path (nbp_conv.hip)
// kernel arguments of the fp32-activation convolutions (nbp_conv.hip, nbp_split.hip)
struct IgemmArgs {
    const float* src0;
    const float* src1;
    int C0, C1;        // channels of each source (multiples of 32; C1 may be 0)
    int cc0;           // C0 / 32
    int ups;           // 1: sources are [B,H/2,W/2,C] read through x2 nearest upsample
    int H, W;          // output spatial size
    int Hs, Ws;        // source spatial size
    int taps;          // 1 (1x1) or 9 (3x3)
    const float* wpk;  // [(C0+C1)/32][taps][N][32]
    int N;
    const float* scale;
    const float* shift;
    int relu;
    float* out;        // split_k==1: [M][N] final; else partial [split][M][N]
    long long M;
    int split_k;
    int chunks_total;
    int chunks_per_split;
    unsigned bytes0, bytes1;   // byte sizes of src0 / src1 (buffer-descriptor range, < 2 GiB)
    unsigned bytesw;           // byte size of the packed weights (halo kernel streams them through a descriptor)
    float* partial;    // split-K scratch [group][split][M][N]
    // second problem of a grouped launch (same shapes, other tensors): blockIdx.z >= split_k
    int groups;
    const float* g_src0;
    const float* g_src1;
    const float* g_wpk;
    const float* g_scale;
    const float* g_shift;
    float* g_out;
    int xcd_remap;     // 1: workgroups of one XCD take a contiguous run of (m, n) tiles (n fastest)
};
struct ConvOperands { const float* src0; const float* src1; const float* wpk; const float* scale; const float* shift; float* out; };
ConvPlan nbp_plan_conv(long long M, int N, int chunks_total, int tile, int split_k, int groups, int H = 0, int W = 0,
                       int ksize = 0);
int nbp_conv_igemm_launch_g(const ConvOperands& o, const ConvOperands* o2, int C0, int C1, int ups, int B, int H, int W,
                            int ksize, int N, int relu, int split_k, int tile, void* ws, size_t ws_bytes, hipStream_t st);

// split path (nbp_split.hip): fp32 tensors, 3x3 layers on the fp16 matrix pipe through two-piece operand splitting.
// planes / wamax from nbp_pack_conv_weight_split_launch; amax0 / amax1 = device words holding max |x| (float bits) of the
// sources (amax1 unused without a second source); amax_out (may be null) receives max |out| by atomicMax (caller zeroes it).
// plan.tile == 0: layer not taken.
struct ConvOperandsSplit {
    const float* src0; const float* src1; const void* planes; const float* scale; const float* shift; float* out;
    const unsigned* amax0; const unsigned* amax1; const unsigned* wamax; unsigned* amax_out;
    const void* planes_up; const unsigned* wamax_up;     // up_conv layers: the parity filters (nbp_pack_upconv_weight_split_launch) or null
    // not null (plain 3x3 layers): source 0 is read as src0[m][c] * psi0[m] -- the attention gate's x * psi (nbp_model.py:60) formed in
    // the consumer's halo staging instead of written by the gate and read back (amax0 is then max |x psi|, measured by the gate launch)
    const float* psi0 = nullptr;
};
ConvPlan nbp_plan_conv_split(long long M, int N, int chunks_total, int split_k, int groups, int H, int W, int ksize, int ups = 0);
int nbp_pack_upconv_weight_split_launch(const float* w_oihw, int N, int C, void* dst, unsigned* wamax_out, hipStream_t st);
// head != null offers the one-channel sigmoid head that alone consumes a 64-channel layer: out1[m] = sigmoid((out[m,:] . w) * scale[0]
// + shift[0]); *headed tells whether the launch wrote it INSTEAD of `out` -- if not, the caller runs nbp_final_1x1_f32 as before
struct ConvHead { const float* w; const float* scale; const float* shift; float* out; };
// pool_out != null offers the 2x2 max-pool of the output(s) [B,H/2,W/2,N]; *pooled tells whether the launch wrote it (only
// without split-K) -- if not, the caller runs nbp_maxpool2_nhwc_f32 as before
int nbp_conv_split_launch_g(const ConvOperandsSplit& o, const ConvOperandsSplit* o2, int C0, int C1, int ups, int B, int H, int W,
                            int ksize, int N, int relu, int split_k, void* ws, size_t ws_bytes, hipStream_t st,
                            float* const* pool_out = nullptr, int* pooled = nullptr, const struct ConvHead* head = nullptr,
                            int* headed = nullptr, double* bn_part = nullptr, int* bn_rows = nullptr);
int nbp_pack_conv_weight_split_launch(const float* w_oihw, int N, int C, int ksize, const float* scale_or_null, int c_off,
                                      int c_total, void* dst, unsigned* wamax_out, hipStream_t st);
int nbp_amax_launch(const float* x, long long n, unsigned* amax_inout, hipStream_t st);
// 3x3 weight gradient on the split scheme: partial sums [splits][9][C0 + C1][N]; amax3 = 3 x 64 words of scratch
int nbp_wgrad_split_launch(const float* src0, int C0, const float* src1, int C1, int ups, int B, int H, int W, const float* dy, int N,
                           int n_tiles, int splits, unsigned* amax3, const unsigned* amax0_in, const unsigned* amax1_in,
                           const unsigned* amaxy_in, float* part, hipStream_t st);
// attention gates (1x1 over K = [src0 | src1], both C channels) on the split scheme
int nbp_pack_gate_weight_split_launch(const float* wg, const float* scale_g, const float* wx, const float* scale_x, int N, int C,
                                      void* dst, unsigned* wamax_out, hipStream_t st);
// psi != null offers the gate's tail (per group: psi weights [N], {scale, shift}, gated output [M,C] = src1 * psi); *fused tells
// whether the launch took it (a workgroup must hold all N columns of its pixels) -- if not, q is written and the caller runs
// nbp_psi_gate_f32 as before
struct GatePsi { const float* wpsi[2]; const float* st[2]; float* gated[2]; unsigned* gated_amax[2] = {nullptr, nullptr};     // gated_amax: zeroed 64-word slots for max |gated| (or null)
                 // psi_only: a fused launch writes psi [M] to the head of gated[g] instead of the gated tensor [M, C] (max |x psi| still goes to
                 // gated_amax): the consumer multiplies while it stages its input (ConvOperandsSplit::psi0) -- a third of the gate's bytes less
                 int psi_only = 0; };
int nbp_gate1x1_split_launch_g(const ConvOperandsSplit& o, const ConvOperandsSplit* o2, int C, long long M, int N, int relu,
                               hipStream_t st, const GatePsi* psi = nullptr, int* fused = nullptr);
int nbp_conv_first_amax_launch(const float* x_nchw, int B, int H, int W, const float* w_oihw, const float* scale, const float* shift,
                               float* out_nhwc, unsigned* amax_out, int* did_amax, hipStream_t st);

// bf16 path (nbp_bf16.hip); K chunks are 64 channels
struct ConvOperandsH { const bf16_t* src0; const bf16_t* src1; const bf16_t* wpk; const float* scale; const float* shift; bf16_t* out;
                       const bf16_t* wpk_up; };    // up_conv layers: the parity filters (nbp_pack_upconv_weight_bf16_launch) or null
ConvPlan nbp_plan_conv_bf16(long long M, int N, int chunks_total, int tile, int split_k, int groups, int H = 0, int W = 0,
                            int ksize = 0, int ups = 0);
int nbp_pack_upconv_weight_bf16_launch(const float* w_oihw, int N, int C, bf16_t* dst, hipStream_t st);
// pool_out / head: as for nbp_conv_split_launch_g (taken by the halo kernels; *pooled / *headed tell).  psi: the attention gate's tail
// for a 1x1 launch over [g | x] whose workgroups hold all N columns (per group: psi weights [N], {scale, shift}, gated output
// [M, C] = x * psi); *psi_fused tells whether the launch took it -- if not, q is written and the caller runs nbp_psi_gate_bf16_launch
struct GatePsiH { const float* wpsi[2]; const float* st[2]; bf16_t* gated[2]; };
int nbp_conv_igemm_bf16_launch_g(const ConvOperandsH& o, const ConvOperandsH* o2, int C0, int C1, int ups, int B, int H,
                                 int W, int ksize, int N, int relu, int split_k, int tile, void* ws, size_t ws_bytes,
                                 hipStream_t st, bf16_t* const* pool_out = nullptr, int* pooled = nullptr,
                                 const struct ConvHead* head = nullptr, int* headed = nullptr, const GatePsiH* psi = nullptr,
                                 int* psi_fused = nullptr);
int nbp_conv_first_bf16_launch(const float* x_nchw, int B, int H, int W, const float* w_oihw, const float* scale,
                               const float* shift, bf16_t* out_nhwc, hipStream_t st);
int nbp_maxpool2_bf16_launch(const bf16_t* in, int B, int H, int W, int C, bf16_t* out, hipStream_t st);
int nbp_psi_gate_bf16_launch(const bf16_t* q, int F, const float* w_psi, const float* s_t2, const bf16_t* x, int C,
                             long long M, bf16_t* out, hipStream_t st);
int nbp_final_1x1_bf16_launch(const bf16_t* in, int B, int H, int W, int C, const float* w_oc, int n_out,
                              const float* scale, const float* shift, int sigmoid, float* out_nchw, hipStream_t st);
