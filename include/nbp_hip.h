/* nbp_hip.h -- C ABI of libnbp_hip.so: the MI355X (gfx950) hot path of NextBestPath.
 *
 * The reference (shiyao-li/NextBestPath) is pure Python and has no native boundary; the
 * functions below are what a maintainer's ctypes binding replaces the cited torch /
 * PyTorch3D / trimesh call sites with (INTEGRATION.md shows the stubs).  Citations are
 * relative to the reference repository root.
 *
 * Conventions
 *  - every function returns 0 on success, a negative NBP_E_* code on an argument error
 *    (nothing is launched / written) or a positive hipError_t if a launch failed;
 *  - every pointer is caller-owned DEVICE memory unless the name ends in _host;
 *  - work is enqueued asynchronously on `stream` (a hipStream_t passed as void*; pass
 *    torch.cuda.current_stream().cuda_stream); no function synchronises or allocates device
 *    memory: scratch comes in through (ws, ws_bytes) sized by the matching *_workspace_bytes;
 *  - activations inside the network are NHWC fp32; the public tensors keep the reference's
 *    NCHW layout ([B,5,S,S] in, [B,8,S/4,S/4] and [B,1,S,S] out);
 *  - no exceptions cross the boundary, no global mutable state besides the explicit handle.
 */
#ifndef NBP_HIP_H
#define NBP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NBP_E_ARG      (-1)  /* null pointer / bad shape / bad enum                       */
#define NBP_E_WS       (-2)  /* workspace too small                                       */
#define NBP_E_SHAPE    (-3)  /* shape not supported by the kernels (e.g. S % 16 != 0)     */

/* ---------------------------------------------------------------- library info */
/* ABI version of this header; bumped on any signature change. */
int nbp_abi_version(void);
/* Writes gcnArchName / CU count of the current device (host strings/ints). */
int nbp_device_info(char* arch_host, int arch_len, int* cu_count_host);
/* The library's launch plans (tile choice, split-K bounds, fusions) are compile-time constants.  Their A/B switches (NBP_*
 * environment variables) are honoured ONLY when the process sets NBP_TUNING=1; an inherited environment
 * never changes results.  nbp_tuning_active: 1 when the opt-in is set.  nbp_tuning_report: writes "NAME=value,..." of the
 * switches read so far whose value differs from the default into buf_host (truncated to len - 1) and returns their number. */
int nbp_tuning_active(void);
int nbp_tuning_report(char* buf_host, int len);
/* Kernel symbol (as a profiler demangles it, without "void" / the anonymous namespace / the argument list) that convolution tile
 * id `tile` (nbp_layer_timing.tile) was most recently launched as in this process; returns its length, 0 if that tile id has not
 * been launched yet.  The launchers build the text from their own template arguments. */
int nbp_tile_kernel_symbol(int tile, char* buf_host, int len);

/* ================================================================ A1: NBP network
 * Replaces NBP.forward, next_best_path/networks/nbp_model.py:110-160 (and the layer
 * classes :8-62) for inference (module.eval(): BatchNorm uses running statistics).
 *
 * Canonical order of the 48 convolutions (index -> reference parameter prefix):
 *   0..9   Conv{1..5}.conv.{0,3}
 *   then for decoder d=1 (levels L=5,4) and d=2 (levels L=5,4,3,2), six per level:
 *          Up{L}_d.up.1, Att{L}_d.W_g.0, Att{L}_d.W_x.0, Att{L}_d.psi.0,
 *          Up_conv{L}_d.conv.0, Up_conv{L}_d.conv.3
 *   i.e. 10..21 decoder 1, 22..45 decoder 2, then 46 Final1, 47 Final2.0
 */
#define NBP_N_CONV 48

typedef struct nbp_weights nbp_weights;   /* opaque: host-side table of pointers into `packed` */

/* Bytes of device memory the packed fp32 weights + folded affine terms need. */
size_t nbp_packed_weights_bytes(void);

/* Re-lays the 48 OIHW fp32 weight tensors into the K-chunked layout the implicit-GEMM
 * kernels stream ([c_in/32][tap][c_out][32]) inside caller memory `packed`, and copies the
 * per-output-channel epilogue terms: out = act(acc * scale + shift).
 *   w[i]      OIHW fp32 weight of conv i (device)
 *   scale[i]  [c_out] fp32 (device): gamma/sqrt(var+eps) (1 when the conv has no BatchNorm)
 *   shift[i]  [c_out] fp32 (device): (bias-mean)*scale+beta
 * For the attention pair (W_g, W_x) the two 1x1 convolutions are fused into one GEMM over
 * the concatenated K = [g | x]: their `scale` is multiplied into the packed weights and the
 * caller passes shift[W_g] = shift_g + shift_x (shift[W_x] is ignored).
 * On success *handle_out is a host object to pass to nbp_forward_f32; free it with
 * nbp_free_weights (which never touches `packed`). */
int nbp_pack_weights(const void* const* w_host_array, const void* const* scale_host_array,
                     const void* const* shift_host_array, void* packed, size_t packed_bytes,
                     void* stream, nbp_weights** handle_out);
void nbp_free_weights(nbp_weights* handle);

size_t nbp_forward_workspace_bytes(int B, int S);

/* x [B,5,S,S] -> out1 [B,8,S/4,S/4] (linear), out2 [B,1,S,S] (sigmoid); S % 16 == 0. */
int nbp_forward_f32(const nbp_weights* handle, const float* x, int B, int S, float* out1,
                    float* out2, void* ws, size_t ws_bytes, void* stream);

/* Profiling twin of nbp_forward_f32: brackets every launch group with hipEvents on `stream`,
 * SYNCHRONISES the stream, and fills timings_host[0..*n_entries_host) (one entry per layer, in
 * launch order; `ms` of a split-K layer includes its reduce kernel).  tile: id of the convolution
 * kernel used (see nbp_conv_igemm_f32), -1 for the other kernels.  Not re-entrant. */
typedef struct nbp_layer_timing {
    char name[48];
    double flops;      /* 2*M*N*K */
    float ms;
    int tile, split_k;
    long long M;
    int N, K;
} nbp_layer_timing;
int nbp_forward_timed_f32(const nbp_weights* handle, const float* x, int B, int S, float* out1,
                          float* out2, void* ws, size_t ws_bytes, void* stream,
                          nbp_layer_timing* timings_host, int max_entries, int* n_entries_host);

/* FLOPs (2*MAC over the 48 convolutions) of one forward at batch B, size S. */
double nbp_forward_flops(int B, int S);

/* ---- bf16 variant (BASELINE.json configs[4]: 512x512 grids, 8 rollouts per GPU in one forward).
 * Same network and same fp32 in/out tensors as nbp_forward_f32 (ref nbp_model.py:110-160); inside, activations
 * are NHWC bf16, the 3x3 / attention-gate weights are bf16, accumulation and every epilogue (folded BatchNorm,
 * ReLU, sigmoid, psi gate) are fp32, and each stored activation is rounded once (nearest even).  It cannot
 * meet the 1e-4 fp32 tolerance (bf16 has 8 mantissa bits); tests/test_gpu_bf16.py states its tolerance.
 * nbp_pack_weights_bf16 takes exactly the inputs of nbp_pack_weights (`packed` of nbp_packed_weights_bytes_bf16()). */
int nbp_pack_weights_bf16(const void* const* w_host_array, const void* const* scale_host_array,
                          const void* const* shift_host_array, void* packed, size_t packed_bytes,
                          void* stream, nbp_weights** handle_out);
size_t nbp_packed_weights_bytes_bf16(void);   /* `packed` of nbp_pack_weights_bf16: the common pack + the up_conv parity filters */
size_t nbp_forward_workspace_bytes_bf16(int B, int S);
/* up_conv parity filters in bf16 for nbp_conv_igemm_bf16 with tile 12 / 13 (8 x 32 low-resolution tiles x 128 / 64 channels;
 * ups must be 1, C1 0): dst [parity][C/64][4 taps][N][64] bf16 = 32 N C bytes. */
int nbp_pack_upconv_weight_bf16(const float* w_oihw, int N, int C, unsigned short* dst, void* stream);
int nbp_forward_bf16(const nbp_weights* handle, const float* x, int B, int S, float* out1,
                     float* out2, void* ws, size_t ws_bytes, void* stream);
int nbp_forward_timed_bf16(const nbp_weights* handle, const float* x, int B, int S, float* out1,
                           float* out2, void* ws, size_t ws_bytes, void* stream,
                           nbp_layer_timing* timings_host, int max_entries, int* n_entries_host);

/* ---- fp32 forward with the 3x3 convolutions on the 16-bit matrix pipe ("split" path, nbp_split.hip).
 * Tensors, accumulation and epilogues are fp32 exactly as in nbp_forward_f32; inside the 3x3 kernels every fp32 operand is
 * scaled by a power of two taken from its tensor's max |x| and cut into two fp16 pieces (hi + lo = s x up to 2^-23), and
 * the product is evaluated as three exact fp16 MFMAs (the dropped lo x lo is < 2^-22 of the product): error against fp64
 * not above the fp32 MFMA chain's at 5.3x its matrix rate; up_conv layers run as four 2x2 parity
 * convolutions of the low-resolution input, the attention gates' joint 1x1 GEMM on the same split scheme.  Layers the split
 * kernels do not take (images that are not multiples of 16 x 32 / 16 x 16 pixels, psi, the heads) run nbp_forward_f32's kernels.  The handle
 * of nbp_pack_weights_split (`packed` of nbp_packed_weights_bytes_split()) also serves nbp_forward_f32. */
size_t nbp_packed_weights_bytes_split(void);
int nbp_pack_weights_split(const void* const* w_host_array, const void* const* scale_host_array,
                           const void* const* shift_host_array, void* packed, size_t packed_bytes, void* stream,
                           nbp_weights** handle_out);
size_t nbp_forward_workspace_bytes_split(int B, int S);
int nbp_forward_split_f32(const nbp_weights* handle, const float* x, int B, int S, float* out1, float* out2, void* ws,
                          size_t ws_bytes, void* stream);
int nbp_forward_timed_split_f32(const nbp_weights* handle, const float* x, int B, int S, float* out1, float* out2,
                                void* ws, size_t ws_bytes, void* stream, nbp_layer_timing* timings_host,
                                int max_entries, int* n_entries_host);
/* One 3x3 layer of that path (stride 1, zero padding 1; src1 / ups as nbp_conv_igemm_f32).
 * nbp_pack_conv_weight_split: planes [chunk of 16 channels][tap][hi|lo][k half][N][8] fp16 (4 bytes per weight; zero the
 * buffer first when C < c_total) scaled by 2^(14 - floor(log2 max|w|)); wamax_out = device word that receives max |w * scale|
 * (float bits).  c_off must be 0 (one scale per layer).
 * nbp_conv3x3_split_f32: images of 16 x 32 pixel tiles with N % 64 == 0, or of 16 x 16 pixel tiles with N % 128 == 0; channel
 * counts multiples of 32; NBP_E_SHAPE otherwise.  amax_in = 64 device words (256 B) whose maximum is max |x| over src0 and
 * src1 (float bits; a bound is enough), NULL = computed here (one more pass over the inputs); amax_out (or NULL) = 64 words
 * that receive max |out| by atomicMax spread over the words: zero them before, and hand them to the consumer layer as its
 * amax_in.  nbp_amax_f32 is that pass on its own (same 64-word convention).  split_k 0 = automatic;
 * ws >= nbp_conv_split_workspace_bytes. */
int nbp_pack_conv_weight_split(const float* w_oihw, int N, int C, int ksize, const float* scale_or_null, int c_off,
                               int c_total, void* dst_planes, void* wamax_out, void* stream);
int nbp_amax_f32(const float* x, long long n, void* amax_inout, void* stream);
/* up_conv (x2 nearest upsample + 3x3 convolution, nbp_model.py:25-33) as four 2x2 convolutions of the low-resolution input, one
 * per output parity, with pre-summed weights (16 tap-products per low-resolution pixel instead of 36).
 * nbp_pack_upconv_weight_split: planes [parity][chunk of 16][4 taps][hi|lo][k half][N][8] fp16 (64 N C bytes); the sums are
 * formed in double and split directly.
 * nbp_upconv3x3_split_f32: src [B,H/2,W/2,C] -> out [B,H,W,N]; the low-resolution image must tile (H/2 % 16 == 0 and W/2 % 32 == 0
 * with N % 64 == 0, or W/2 % 16 == 0 with N % 128 == 0), NBP_E_SHAPE otherwise; other arguments as nbp_conv3x3_split_f32. */
int nbp_pack_upconv_weight_split(const float* w_oihw, int N, int C, void* dst_planes, void* wamax_out, void* stream);
int nbp_upconv3x3_split_f32(const float* src, int C, int B, int H, int W, const void* planes_up, const void* wamax_up, int N,
                            const float* scale, const float* shift, int relu, float* out, const void* amax_in_or_null,
                            void* amax_out_or_null, int split_k, void* ws, size_t ws_bytes, void* stream);
size_t nbp_conv_split_workspace_bytes(int B, int H, int W, int N, int split_k);
/* The same for split_k = 0 as the planner will actually run the layer (C = C0 + C1 input channels): its split-K count is
 * returned through split_k_out (may be NULL; 0 = the layer is not taken by the split kernel).  Slices exist for occupancy
 * (small grids) and for accuracy: an accumulation chain is bounded to NBP_SPLIT_MAX_K_SMALL (576) products where a slice of
 * the output is <= 64 MB and to NBP_SPLIT_MAX_K (2304) elsewhere, so that the result's distance to the exact sum does not
 * grow with the batch size. */
size_t nbp_conv_split_planned_workspace_bytes(int B, int H, int W, int C, int N, int ups, int* split_k_out);
/* ... for a split_k request: 0 = the planner with its accuracy-driven chain bound (the eval forward), < 0 = the planner by
 * occupancy only (the training step), > 0 = as given.  The same request goes to nbp_conv3x3_split_f32 / nbp_upconv3x3_split_f32. */
size_t nbp_conv_split_planned_workspace_bytes_k(int B, int H, int W, int C, int N, int ups, int split_k, int* split_k_out);
int nbp_conv3x3_split_f32(const float* src0, int C0, const float* src1, int C1, int ups, int B, int H, int W,
                          const void* w_planes, const void* wamax, int N, const float* scale, const float* shift, int relu,
                          float* out, const void* amax_in_or_null, void* amax_out_or_null, int split_k, void* ws,
                          size_t ws_bytes, void* stream);
/* Weight gradient of Conv1.conv.0 (nbp_model.py:66: 5 -> 64 channels, 3x3) straight from the NCHW network input x [B,5,H,W] and
 * dy [B,H,W,64] (NHWC): dw [64][5][3][3].  H % 8 == 0, W % 32 == 0 (else NBP_E_SHAPE: the caller pads the input to 64 channels and
 * takes nbp_conv_wgrad_split_f32).  The forward of the layer is nbp_conv_first_f32. */
size_t nbp_conv_first_wgrad_workspace_bytes(void);
int nbp_conv_first_wgrad_f32(const float* x_nchw, int B, int H, int W, const float* dy, float* dw, void* ws, size_t ws_bytes, void* stream);
/* Every weight pack of a training step in two launches.  descs_dev: n records on the device, each
 *   { const float* w; uint16_t* planes; uint16_t* planes_t; uint32_t* wamax; int32 N, C, kind, pad; }   (nbp_prepack_desc_bytes() = 48)
 * kind 0: 3x3 layer w [N][C][3][3] -> planes of nbp_pack_conv_weight_split and planes_t of nbp_pack_conv_weight_split_dgrad;
 * kind 1: 1x1 layer w [N][C] -> planes (ksize 1) and planes_t of nbp_pack_conv1x1_weight_split_dgrad;
 * kind 2: up_conv layer -> planes of nbp_pack_upconv_weight_split and planes_t of nbp_pack_upconv_weight_split_dgrad.
 * planes_t may be null.  wamax: the record's own word inside wamax_words [n] (zeroed by the call).  N % 16 == 0, C % 16 == 0. */
int nbp_prepack_weights_split(const void* descs_dev, int n, void* wamax_words, void* stream);
int nbp_prepack_desc_bytes(void);
/* Training-step forms of nbp_pack_conv_weight_split / _dgrad: `_prezeroed` takes a max-|w| word the caller zeroed (no memset launch),
 * `_dgrad_known` reuses the word the forward's pack of the same weights left (the data-gradient planes hold the same values): one
 * launch instead of three. */
int nbp_pack_conv_weight_split_prezeroed(const float* w_oihw, int N, int C, int ksize, int c_total, void* dst_planes,
                                         void* wamax_zeroed, void* stream);
int nbp_pack_conv_weight_split_dgrad_known(const float* w_oihw, int N, int C, int c_total, void* dst_planes, const void* wamax_known,
                                           void* stream);
/* Data gradient of an up_conv layer (nbp_model.py:25-33: x2 nearest upsample + 3x3 convolution, C -> N channels) in the parity form of
 * nbp_upconv3x3_split_f32: dx [B,H,W,C] at the LOW resolution straight from dy [B,2H,2W,N] -- 16 tap-products per low-resolution pixel
 * instead of the 36 of the full-resolution 3x3 data gradient followed by a 2x2 sum.  nbp_pack_upconv_weight_split_dgrad: 32 N C fp16
 * from the layer's own OIHW weight; scale / shift: C ones / zeros; amax_in: 64-word max-|dy| slot; ws: nbp_upconv_split_dgrad_workspace_bytes.
 * NBP_E_SHAPE when the low-resolution image does not tile (H % 16, W % 32 with C % 64, or W % 16 with C % 128). */
int nbp_pack_upconv_weight_split_dgrad(const float* w_oihw, int N, int C, void* dst_planes, void* wamax_out, void* stream);
size_t nbp_upconv_split_dgrad_workspace_bytes(int B, int H, int W, int N, int C);
int nbp_upconv3x3_split_dgrad_f32(const float* dy, int N, int B, int H, int W, const void* planes, const void* wamax, int C,
                                  const float* scale, const float* shift, float* dx, const void* amax_in, void* amax_out_or_null,
                                  void* ws, size_t ws_bytes, void* stream);
/* Weight gradient of the same layer in parity form: dW [N][C][3][3] from the LOW-resolution input x [B,Hs,Ws,C] and dy [B,2Hs,2Ws,N] as
 * 16 tap-GEMMs over M / 4 pixels (four parities x 2 x 2 taps) folded into the 3x3 filter by the reduce kernel; C % 64 == 0, N % 64 == 0,
 * (Ws % 32 == 0 and Hs % 2 == 0) or (Ws % 16 == 0 and Hs % 4 == 0), else NBP_E_SHAPE.  amax_x / amax_y: 64-word max-|.| slots. */
size_t nbp_upconv_wgrad_split_workspace_bytes(int B, int Hs, int Ws, int C, int N);
int nbp_upconv_wgrad_split_f32(const float* x, int C, int B, int Hs, int Ws, const float* dy, int N, float* dw,
                               const void* amax_x, const void* amax_y, void* ws, size_t ws_bytes, void* stream);
/* 1x1 convolution on the same scheme (training: Attention_block.W_g / W_x, nbp_model.py:44-53, and their data gradients):
 * out [M][N] = src [M][C] W * scale + shift, C % 32 == 0, N % 32 == 0.  w_planes / wamax: nbp_pack_conv_weight_split with ksize 1
 * (forward) or nbp_pack_conv1x1_weight_split_dgrad (dx = dy W^T from the layer's own [N][C] weight).  amax_in: 64-word max-|src| slot. */
int nbp_conv1x1_split_f32(const float* src, int C, long long M, const void* w_planes, const void* wamax, int N,
                          const float* scale, const float* shift, int relu, float* out, const void* amax_in, void* stream);
int nbp_pack_conv1x1_weight_split_dgrad(const float* w_nc, int N, int C, void* dst_planes, void* wamax_out, void* stream);
/* Training: planes of the DATA-GRADIENT convolution of a 3x3 layer (dx = conv3x3(dy, w'), w'[c][n][tap] = w[n][c][8 - tap]; autograd
 * of nn.Conv2d, nbp_model.py:14-33) straight from the layer's weights w [N][C][3][3]: C rows, N input channels padded to c_total */
int nbp_pack_conv_weight_split_dgrad(const float* w_oihw, int N, int C, int c_total, void* dst_planes, void* wamax_out, void* stream);

/* Training: the same convolution (and its up_conv form), whose epilogue also leaves the column sums of the output and of its
 * squares for the BatchNorm behind it (nextbestpath_amd/networks/training.py: ConvFn -> BNFn; the reference's nn.Sequential of
 * Conv2d + BatchNorm2d, nbp_model.py:14-33): bn_part receives *bn_rows rows of [2][N] doubles -- at most
 * nbp_conv_bn_part_rows(B, H, W) rows -- finalised by nbp_bn_train_forward_part4_f32 without another pass over the tensor.
 * *bn_rows = 0: this launch did not take them (split-K or half-height tiles); the caller's BatchNorm reads the tensor itself. */
int nbp_conv_bn_part_rows(int B, int H, int W);
int nbp_conv3x3_split_bn_f32(const float* src0, int C0, const float* src1, int C1, int ups, int B, int H, int W,
                             const void* w_planes, const void* wamax, int N, const float* scale, const float* shift, int relu,
                             float* out, const void* amax_in_or_null, void* amax_out_or_null, int split_k, void* ws,
                             size_t ws_bytes, double* bn_part, int* bn_rows, void* stream);
int nbp_upconv3x3_split_bn_f32(const float* src, int C, int B, int H, int W, const void* planes_up, const void* wamax_up, int N,
                               const float* scale, const float* shift, int relu, float* out, const void* amax_in_or_null,
                               void* amax_out_or_null, int split_k, void* ws, size_t ws_bytes, double* bn_part, int* bn_rows,
                               void* stream);
/* BatchNorm2d training forward from those partial sums (zero_row = C zeros): finalize + normalise, x is read once */
int nbp_bn_train_forward_part4_f32(const float* x, long long M, int C, const float* gamma, const float* beta, float eps,
                                   float momentum, float* running_mean, float* running_var, int relu, float* mean, float* invstd,
                                   float* y, void* amax_out, double* stat_out, int stat_doubles, const double* part, int rows,
                                   const float* zero_row, void* stream);
/* Single bf16 layer: as nbp_conv_igemm_f32 with bf16 (uint16 storage) NHWC sources / output, C0, C1 multiples
 * of 64, w_packed from nbp_pack_conv_weight_bf16 ([(c_off+c)/64][tap][N][64] bf16), fp32 scale / shift. */
int nbp_conv_igemm_bf16(const unsigned short* src0, int C0, const unsigned short* src1, int C1, int ups,
                        int B, int H, int W, int ksize, const unsigned short* w_packed, int N,
                        const float* scale, const float* shift, int relu, unsigned short* out,
                        int split_k, int tile, void* ws, size_t ws_bytes, void* stream);
size_t nbp_conv_igemm_bf16_workspace_bytes(int B, int H, int W, int N, int split_k);
int nbp_pack_conv_weight_bf16(const float* w_oihw, int N, int C, int ksize, const float* scale_or_null,
                              int c_off, int c_total, unsigned short* dst, void* stream);
/* Element-wise conversions (round to nearest even), for tests and callers holding fp32 maps. */
int nbp_f32_to_bf16(const float* in, long long n, unsigned short* out, void* stream);
int nbp_bf16_to_f32(const unsigned short* in, long long n, float* out, void* stream);

/* ---- single layers (same kernels the forward uses; exported for layer-level parity tests)
 * Implicit-GEMM convolution on NHWC fp32, k in {1,3}, stride 1, "same" padding:
 *   input channels [0,C0) come from src0, [C0,C0+C1) from src1 (fused torch.cat, ref :128);
 *   ups != 0 reads the sources through a x2 nearest upsample (fused nn.Upsample, ref :27):
 *   sources are then [B,H/2,W/2,C];  C0,C1 multiples of 32 (C1 may be 0), N multiple of 32.
 *   w_packed: layout produced by nbp_pack_conv_weight.  out [B,H,W,N] = act(acc*scale+shift).
 *   split_k >= 1 splits the K loop over blockIdx.z (ws must hold split_k*B*H*W*N floats);
 *   split_k == 0 lets the library choose.  tile: 0 = auto; 1..5 = implicit-GEMM workgroup tiles (pixels x
 *   channels) 128x128, 256x64, 256x32, 128x64, 64x128; 6 / 7 = the halo-tile kernel (3x3 only, H % 8 == 0,
 *   W % 32 == 0) with 128 / 64 output channels per workgroup, whose split-K slices are whole channel chunks.
 *   Ids reported by the timing twins for the other paths: 10 / 11 = split-path 16-row tiles (plain / up_conv parity form), 15 / 16 = the
 *   same on 8 x 32-pixel tiles (launches with fewer 16-row tiles than CUs); nbp_conv_igemm_bf16 also takes 12 / 13 (parity form) and
 *   14 (3x3, H % 16 == 0, W % 32 == 0, N % 64 == 0, no split-K: 16 x 32 pixels x 64 channels, weights in registers). */
int nbp_conv_igemm_f32(const float* src0, int C0, const float* src1, int C1, int ups, int B, int H,
                       int W, int ksize, const float* w_packed, int N, const float* scale,
                       const float* shift, int relu, float* out, int split_k, int tile, void* ws,
                       size_t ws_bytes, void* stream);
size_t nbp_conv_igemm_workspace_bytes(int B, int H, int W, int N, int split_k);

/* OIHW [N][C][k][k] -> packed [(c_off+c)/32][tap][N][32]; optional per-n scale multiply.
 * c_total = total input channels of the fused K (c_off + C <= c_total). */
int nbp_pack_conv_weight(const float* w_oihw, int N, int C, int ksize, const float* scale_or_null,
                         int c_off, int c_total, float* dst, void* stream);

/* Conv1.conv.0: NCHW [B,5,H,W] -> NHWC [B,H,W,64], 3x3, act(acc*scale+shift), ReLU.
 * w is the raw OIHW [64,5,3,3] tensor. */
int nbp_conv_first_f32(const float* x_nchw, int B, int H, int W, const float* w_oihw,
                       const float* scale, const float* shift, float* out_nhwc, void* stream);
/* The same layer without the ReLU (training: the train-mode BatchNorm that follows takes conv + bias); H % 8 == 0, W % 32 == 0. */
int nbp_conv_first_linear_f32(const float* x_nchw, int B, int H, int W, const float* w_oihw, const float* scale,
                              const float* shift, float* out_nhwc, void* stream);
/* nn.MaxPool2d(2,2) on NHWC (ref :68). */
int nbp_maxpool2_nhwc_f32(const float* in, int B, int H, int W, int C, float* out, void* stream);
/* Attention gate tail (ref :59-62): psi = sigmoid((q . w_psi) * s + t) per pixel,
 * out = x * psi.  q [M,F] (already ReLU'd g1+x1), x/out [M,C]. */
int nbp_psi_gate_f32(const float* q, int F, const float* w_psi, const float* s_t2, const float* x,
                     int C, long long M, float* out, void* stream);
/* Final 1x1 convolutions (ref :85,105-106): NHWC [B,H,W,C] -> NCHW [B,n_out,H,W],
 * out = act(acc*scale+shift), act = sigmoid if `sigmoid` else identity.  n_out <= 8. */
int nbp_final_1x1_f32(const float* in, int B, int H, int W, int C, const float* w_oc, int n_out,
                      const float* scale, const float* shift, int sigmoid, float* out_nchw,
                      void* stream);
/* Layout helpers (tests / training path). */
int nbp_nchw_to_nhwc_f32(const float* in, int B, int C, int H, int W, float* out, void* stream);
int nbp_nhwc_to_nchw_f32(const float* in, int B, int C, int H, int W, float* out, void* stream);

/* ================================================================ A4-A7: map accumulation
 * World points -> agent-centred top-down count images.
 * Index rule (next_best_path/utility/utils.py:198-223, :160-164):
 *   v0 = -(p.z - c.z), v1 = -(p.x - c.x)                       (utils.py:166-196, R = I)
 *   i0 = rint((v0 - lo) * (S / (hi - lo))) , i1 likewise        (fp32, half-to-even)
 *   counted iff 0 <= i0,i1 < S;  out[i0*S + i1] += 1
 */
/* utils.py:166-196 transform_points_to_n_pieces (no_rotation=True): [N,3] -> [N,2]. */
int nbp_transform_points_f32(const float* points, long long N, float cx, float cy, float cz,
                             float* out_2d, void* stream);
/* utils.py:198-223 map_points_to_n_imgs: pts2d [n,m,2] -> out [n,S0,S1] fp32 counts.
 * `out` is zeroed by this call. */
int nbp_map_points_to_imgs_f32(const float* pts2d, int n, long long m, int S0, int S1, float lo,
                               float hi, float* out, void* stream);
/* utils.py:160-164 get_point_position_in_the_img for K points: -> [2,K] int64 (no bounds
 * check, like the reference). */
int nbp_point_position_i64(const float* pts2d, long long K, int S0, int S1, float lo, float hi,
                           long long* out_2xK, void* stream);
/* One fused pass over the accumulated cloud (nbp_planning.py:114-127 slab split +
 * :172-183 full / height-band projections): out [6,S,S] fp32, zeroed by this call:
 *   ch 0..3  height slabs: bin = #{k : bounds[k] < p.y} - 1 (torch.bucketize, right=False)
 *            counted iff 0 <= bin < 4 (n_bounds is len(y_bins)-1, normally 4, at most 8)
 *   ch 4     points that fall in no slab (so that ch0+..+ch4 = projection of ALL points)
 *   ch 5     points with  band_lo < p.y < band_hi   (the reference's +-0.1 height band)
 * N_dev_or_null: when non-null the point count is read from device memory (the rollout keeps the
 * cloud size on the device so the step loop never synchronises) and N is only an upper bound. */
int nbp_map_accumulate_f32(const float* points, long long N, const long long* N_dev_or_null, float cx,
                           float cy, float cz, const float* bounds_host, int n_bounds, float band_lo,
                           float band_hi, int S, float lo, float hi, float* out6, void* stream);
/* The "observation -> network input" stage of one exploration step (nbp_planning.py:114-137) in one call:
 * nbp_map_accumulate_f32 into out6, the trajectory channel (camera positions so far, transformed like the cloud:
 * (-(z - cz), -(x - cx)), counted per cell as nbp_map_points_to_imgs_f32 does) into net_in5[4], and
 * net_in5[0..3] = out6[0..3]; net_in5 = one [5,S,S] map of the network's input batch.  traj_pts = device history of
 * camera positions with n_traj_old valid points; traj_fresh_host = up to 8 new positions (host, passed in the kernel
 * arguments), appended to traj_pts[n_traj_old ..] by the same launch.  Two memsets, one kernel, one copy. */
int nbp_step_maps_f32(const float* points, long long N, const long long* N_dev_or_null, float cx, float cy,
                      float cz, const float* bounds_host, int n_bounds, float band_lo, float band_hi, int S,
                      float lo, float hi, float* traj_pts, int n_traj_old, const float* traj_fresh_host,
                      int n_traj_fresh, float* out6, float* net_in5, void* stream);
/* The same for the n <= 16 rollouts of a lock-step group in ONE kernel launch (two memsets, one kernel, one strided copy
 * instead of n times that): the step's kernels are latency-bound -- one workgroup per CU, a chain of dependent round trips --
 * so a group costs one such chain.  Rollout r writes out6_all[r] of [n][6][S][S] and net_in_all[r] of [n][5][S][S]; the other
 * arrays are HOST arrays of n entries (points / N_dev / traj_pts: device pointers; N_cap: an upper bound of the cloud size,
 * it sizes the grid; poses_xyz [n][3]; bounds [n][8]; band_lo_hi [n][2]; traj_fresh [n][24]).  Results are identical to n
 * calls of nbp_step_maps_f32. */
int nbp_step_maps_batch_f32(int n, const float* const* points, const long long* N_cap, const long long* const* N_dev,
                            const float* poses_xyz_host, const float* bounds_host, const int* n_bounds,
                            const float* band_lo_hi_host, int S, float lo, float hi, float* const* traj_pts,
                            const int* n_traj_old, const float* traj_fresh_host, const int* n_traj_fresh,
                            float* out6_all, float* net_in_all, void* stream);
/* ---- Tile-binned shadow copy of a rollout's cloud (round 4).  The six maps are a translated window of the world, so the map build
 * can run on points grouped by (x, z) tile: one workgroup per 2048-point PAGE of a tile on a dense 16 x 16-cell x 6-channel
 * histogram in LDS, tiles outside the +-40 window never read.  The canonical cloud (append order) is untouched; the store holds a
 * second copy of every point, filed when a map build first sees it.  Maps are bit-identical to nbp_step_maps_f32's (integer counts).
 *   nbp_cloud_bins_bytes      bytes of the store for a scene whose (x, z) extent is [lo_xz, hi_xz] and a cloud of `capacity` points
 *                             (2.5-unit tiles with one tile of margin; 0 = bad arguments)
 *   nbp_cloud_bins_geometry   the same plan as numbers: {nx, nz, max_pages}, {x0, z0, tile}
 *   nbp_cloud_bins_init       empties the store (256-byte aligned): call it whenever the cloud is reset to zero points
 *   nbp_step_maps_binned_f32  nbp_step_maps_f32 on the store, two launches: points [n_binned, N) of `points` are filed into their
 *                             tiles' pages, then the maps are built from the pages (every point is counted pre-aggregated per tile:
 *                             a fused form that counted the new points with one global atomic each cost the lock-step 5 %).
 *                             page_bound = page workgroups to launch (max_pages is always enough; fewer only cost time: they
 *                             stride over every page that exists).  traj_pts and net_in5 may both be NULL: only out6 is produced.
 *                             Points that cannot be filed (outside the tile grid, page pool exhausted) are kept in an index list
 *                             and counted with direct atomics; if that list overflows too (65536 entries) header word 2 (error)
 *                             becomes non-zero and every later build counts the whole cloud directly: slower, never wrong.
 *   nbp_step_maps_binned_batch_f32  the same for the n <= 16 rollouts of a lock-step group (stores[n], page_bound[n]: HOST arrays).
 * Store header (device, int32 words): 0 n_pages, 1 n_overflow, 2 error, 3 ticket, 4-5 n_binned (int64). */
size_t nbp_cloud_bins_bytes(const float* lo_xz_host, const float* hi_xz_host, long long capacity);
int nbp_cloud_bins_geometry(const float* lo_xz_host, const float* hi_xz_host, long long capacity, int* nx_nz_maxpages_host,
                            float* x0_z0_tile_host);
int nbp_cloud_bins_init(void* store, size_t store_bytes, const float* lo_xz_host, const float* hi_xz_host, long long capacity,
                        void* stream);
int nbp_step_maps_binned_f32(void* store, int page_bound, const float* points, long long N, const long long* N_dev_or_null,
                             float cx, float cy, float cz, const float* bounds_host, int n_bounds, float band_lo, float band_hi,
                             int S, float lo, float hi, float* traj_pts, int n_traj_old, const float* traj_fresh_host,
                             int n_traj_fresh, float* out6, float* net_in5, void* stream);
/* nbp_step_maps_binned_f32 as ONE launch (round 6): the page build alone, for a store whose points were filed by the launch that
 * appended them to the cloud and whose outputs that launch cleared (nbp_unproject_append_filed_f32 with the same store, zero6 = out6,
 * zero1 = net_in5 + 4 S^2).  Same arguments, same maps.  Points the store has not seen (appended by any other route since) are still
 * counted -- directly, slowly -- so a caller's bookkeeping slip costs time, never points. */
int nbp_step_maps_prefiled_f32(void* store, int page_bound, const float* points, long long N, const long long* N_dev_or_null,
                               float cx, float cy, float cz, const float* bounds_host, int n_bounds, float band_lo, float band_hi,
                               int S, float lo, float hi, float* traj_pts, int n_traj_old, const float* traj_fresh_host,
                               int n_traj_fresh, float* out6, float* net_in5, void* stream);
int nbp_step_maps_binned_batch_f32(int n, void* const* stores, const int* page_bound, const float* const* points,
                                   const long long* N_cap, const long long* const* N_dev, const float* poses_xyz_host,
                                   const float* bounds_host, const int* n_bounds, const float* band_lo_hi_host, int S, float lo,
                                   float hi, float* const* traj_pts, const int* n_traj_old, const float* traj_fresh_host,
                                   const int* n_traj_fresh, float* out6_all, float* net_in_all, void* stream);
/* ---- The other latency-bound stages of an exploration step for the rollouts of a lock-step group (n <= 12; coverage: n <= 16),
 * one launch per kernel instead of n.  Every array argument is a HOST array of n entries holding device pointers / scalars;
 * results are identical to n single calls (tests/test_gpu_rollout.py).
 * nbp_coverage_count_planned_batch_f32: nbp_coverage_count_planned_f32 per item (bbox_lo / bbox_hi: [n][3]).
 * nbp_unproject_append_shaded_batch_f32: nbp_unproject_append_shaded_f32 per item for n_frames <= 4 frames each (H W % 4 == 0);
 *   depth / zface: [n][n_frames] pointers, one per FRAME ([H][W] each; a ring's frames need not be adjacent);
 *   cams12_host [n][n_frames][12]; zface / verts / faces / vcolors / cloud_rgb may be NULL (or hold NULLs): depth only;
 *   counts2[r]: 2 n_frames ints (valid, kept per frame); ws[r] >= nbp_unproject_workspace_bytes(n_frames, H, W) each.
 * nbp_raster_zface_batch_f32: nbp_raster_zface_f32 per item (each its own mesh) for n_frames <= 4 views each;
 *   ws[r] >= nbp_raster_workspace_bytes(n_faces[r], n_frames, H, W, 0). */
int nbp_coverage_count_planned_batch_f32(int n, void* const* plans, const int* G, float threshold, const float* bbox_lo_host,
                                         const float* bbox_hi_host, const float* const* pc3, const long long* N,
                                         const long long* const* N_dev, const long long* sample_k, const unsigned* seed,
                                         const unsigned* epoch, int* const* count_accum, int* const* m_out, void* stream);
int nbp_unproject_append_shaded_batch_f32(int n, const float* const* depth, const void* const* zface, const float* const* verts,
                                          const int* const* faces, const float* const* vcolors, const float* cams12_host,
                                          int n_frames, int H, int W, float tan_half_fov, float fov_range,
                                          double gathering_factor, const unsigned* seeds, float ambient, int* const* counts2,
                                          float* const* cloud, float* const* cloud_rgb, long long* const* cloud_count,
                                          const long long* capacity, void* const* ws, size_t ws_bytes_each, void* stream);
int nbp_raster_zface_batch_f32(int n, const float* const* verts, const int* n_verts, const int* const* faces, const int* n_faces,
                               const float* cams12_host, int n_frames, int H, int W, float tan_half_fov, float z_clip,
                               float* const* zbuf, void* const* zface, void* const* ws, const size_t* ws_bytes, void* stream);

/* ================================================================ A14-A17: simulator
 * PyTorch3D / trimesh conventions restated (third-party; parity with the libraries unpinned):
 * cameras are HOST arrays [n][12] fp32 = R row-major (9) then T (3) with X_view = X_world R + T;
 * n <= 8: they travel in the kernel arguments, so a step never blocks on a host->device copy. */

/* Camera.compute_partial_point_cloud (macarons/utility/macarons_utils.py:2811-2847) for n_frames
 * depth maps at once, appended to a device-resident cloud:
 *   valid  = (mask ? mask != 0 : depth > -1) && depth < fov_range            (mu:2825-2828)
 *   n_keep = int(n_valid * gathering_factor)                                  (mu:2836)
 *   kept   = the first n_keep entries of a seeded pseudo-random permutation of the valid pixels
 *            (stands in for torch.randperm, mu:2837; see perm_index in csrc/common.h)
 *   point  = un-projection of (ndc_x(col), ndc_y(row), depth) through the FoV camera (mu:2788-2809)
 * counts2[f] = {n_valid, n_keep}; points land at cloud[*cloud_count + sum_{g<f} n_keep_g + j];
 * *cloud_count is advanced on the device (clamped to capacity). */
size_t nbp_unproject_workspace_bytes(int n_frames, int H, int W);
int nbp_unproject_append_f32(const float* depth, const unsigned char* mask_or_null, const float* cams12_host,
                             int n_frames, int H, int W, float tan_half_fov, float fov_range,
                             double gathering_factor, unsigned seed, int* counts2, float* cloud,
                             long long* cloud_count, long long capacity, void* ws, size_t ws_bytes,
                             void* stream);

/* Camera.capture_image's depth output (mu:2743-2786): zbuf [n_frames,H,W] = view-space z of the
 * nearest face through each pixel centre (perspective-correct, faces clipped at z_clip), -1 where
 * no face.  Faces are binned in two levels (8x8-pixel tiles inside 64x64-pixel coarse tiles) into lists with room for
 * every face: no face is ever dropped (the reference renders with max_faces_per_bin = 500000,
 * macarons/testers/scene.py:440-446).  bin_cap and overflow_flag are kept for ABI compatibility and ignored. */
size_t nbp_raster_workspace_bytes(int n_faces, int n_frames, int H, int W, int bin_cap);
int nbp_raster_zbuf_f32(const float* verts, int n_verts, const int* faces, int n_faces,
                        const float* cams12_host, int n_frames, int H, int W, float tan_half_fov,
                        float z_clip, int bin_cap, float* zbuf, int* overflow_flag, void* ws,
                        size_t ws_bytes, void* stream);

/* The same render with colours (mu:2743-2763): rgb [n_frames,H,W,3] = ambient x barycentric interpolation of the winning
 * face's vertex colours vcolors3 [V,3] (SoftPhongShader under AmbientLights on a TexturesVertex mesh), white background,
 * then torchvision's adjust_contrast(contrast_factor) (identity for 1); zbuf as above.  ws >= nbp_raster_rgb_workspace_bytes. */
size_t nbp_raster_rgb_workspace_bytes(int n_faces, int n_frames, int H, int W);
int nbp_raster_rgbz_f32(const float* verts, int n_verts, const int* faces, int n_faces, const float* vcolors3,
                        const float* cams12_host, int n_frames, int H, int W, float tan_half_fov, float z_clip,
                        float ambient, float contrast_factor, float* zbuf, float* rgb, void* ws, size_t ws_bytes,
                        void* stream);
/* nbp_unproject_append_f32 that also appends the colours of the kept pixels (rgb [n_frames,H,W,3]) to cloud_rgb
 * [capacity,3] at the same indices (compute_partial_point_cloud with images, mu:2840-2845). */
int nbp_unproject_append_rgb_f32(const float* depth, const unsigned char* mask_or_null, const float* rgb,
                                 const float* cams12_host, int n_frames, int H, int W, float tan_half_fov,
                                 float fov_range, double gathering_factor, unsigned seed, int* counts2,
                                 float* cloud, float* cloud_rgb, long long* cloud_count, long long capacity,
                                 void* ws, size_t ws_bytes, void* stream);

/* Deferred shading (the step loop's form of the colour render): the rasteriser also returns zface [n_frames,H,W] u64 =
 * (depth bits << 32 | nearest face), ~0 for background; colours are then evaluated only where they are consumed --
 * nbp_shade_image_f32 gives the whole rgb image (contrast_factor must be 1: a contrast change needs the eager
 * nbp_raster_rgbz_f32), nbp_unproject_append_shaded_f32 shades just the kept pixels (about 5 %) while appending them.
 * Both rebuild the face record with the rasteriser's arithmetic: colours are bit-identical to nbp_raster_rgbz_f32's. */
int nbp_raster_zface_f32(const float* verts, int n_verts, const int* faces, int n_faces, const float* cams12_host,
                         int n_frames, int H, int W, float tan_half_fov, float z_clip, float* zbuf, void* zface,
                         void* ws, size_t ws_bytes, void* stream);
int nbp_shade_image_f32(const void* zface, const float* verts, const int* faces, const float* vcolors3,
                        const float* cams12_host, int n_frames, int H, int W, float tan_half_fov, float ambient,
                        float contrast_factor, float* rgb, void* stream);
int nbp_unproject_append_shaded_f32(const float* depth, const unsigned char* mask_or_null, const void* zface,
                                    const float* verts, const int* faces, const float* vcolors3,
                                    const float* cams12_host, int n_frames, int H, int W, float tan_half_fov,
                                    float fov_range, double gathering_factor, unsigned seed, float ambient,
                                    int* counts2, float* cloud, float* cloud_rgb, long long* cloud_count,
                                    long long capacity, void* ws, size_t ws_bytes, void* stream);
/* The three calls above in one (by which colour source is given: none, rgb [n_frames,H,W,3], or zface + verts + faces + vcolors3)
 * that also FILES every appended point into the cloud's tile-binned store (nbp_cloud_bins_init; the point is in registers anyway)
 * and, when zero6 / zero1 are given, clears the 6 S^2 / S^2 floats the map build behind it accumulates into: that build is then
 * nbp_step_maps_prefiled_f32 -- one launch instead of two (round 6).  It files only while the store is in step with the cloud
 * (header n_binned == *cloud_count at entry); otherwise the points stay unfiled and the next build counts them directly.
 * Needs the three-launch form (H W % 4 == 0, 16-byte aligned frames): NBP_E_SHAPE otherwise.  Replaces the same reference lines as
 * nbp_unproject_append_f32 (mu:2788-2847) + the scatter's bookkeeping (next_best_path/testers/nbp_planning.py:114-127). */
int nbp_unproject_append_filed_f32(const float* depth, const unsigned char* mask_or_null, const float* rgb_or_null,
                                   const void* zface_or_null, const float* verts, const int* faces, const float* vcolors3,
                                   const float* cams12_host, int n_frames, int H, int W, float tan_half_fov,
                                   float fov_range, double gathering_factor, unsigned seed, float ambient, int* counts2,
                                   float* cloud, float* cloud_rgb_or_null, long long* cloud_count, long long capacity,
                                   void* bins_store, float* zero6_or_null, float* zero1_or_null, int S, void* ws,
                                   size_t ws_bytes, void* stream);

/* line_segment_mesh_intersection (mu:120-151): hit[e] = 1 iff the ray from segs6[e][0:3] towards
 * segs6[e][3:6] meets a triangle at distance < |segment|. */
int nbp_segments_hit_mesh_f32(const float* verts, const int* faces, int n_faces, const float* segs6,
                              int n_segs, int* hit, void* stream);
/* check_camera_in_mesh (next_best_path/utility/long_term_utils.py:158-170): counts3[k] = number
 * of triangles hit from pts3[k] along +Y, +X, +Z (inside iff all three are odd). */
int nbp_axis_ray_counts_f32(const float* verts, const int* faces, int n_faces, const float* pts3,
                            int n_pts, int* counts3, void* stream);
/* GT obstacle label of a training record: get_binary_obstacle_array (next_best_path/utility/utils.py:226-262,
 * trimesh mesh_plane + matplotlib + PIL in the reference).  out [S,S] fp32 in {0,1}: 1 where the pixel centre is
 * within half_width_px of the intersection of the mesh with the plane y = y0; the window is [lo,hi] around
 * (cx, cz), column ~ -(x - cx), row ~ -(z - cz) (the reference's axes after its left-right flip).
 * The reference's line width is 1.5 pt at 100 dpi = 2.08 px: half_width_px = 1.04.  Parity unpinned
 * (third-party rasterisation); oracle/slice_raster.py restates THIS definition bit for bit. */
int nbp_slice_obstacle_f32(const float* verts, const int* faces, int n_faces, float y0, float cx, float cz,
                           int S, float lo, float hi, float half_width_px, float* out, void* stream);
/* The same label on the REFERENCE's pixel grid (round 6; pinned by tests/golden/obstacle_label.npz, which the reference's own
 * draw -> PNG -> resize -> threshold stage produced from this library's mesh / plane segments).  utils.py:232-258 draws into the
 * axes of a 2.56 in x 2.56 in figure at 100 dpi (default subplot box 198.4 x 197.12 px), x limits +-view/2 around the camera and
 * the y limits shrunk to keep the aspect (adjustable='datalim': 79.48 units for 80), saves the axes box ('tight' includes it:
 * 198 x 197 px), resizes to S x S and flips left-right.  So a world point lands on u = ((cx - x) + half_u) scale_u,
 * v = ((cz - z) + half_v) scale_v with scale_u = 2.48 * 256 / 198, scale_v = 2.48 * 256 / 197 px per unit at S = 256, and a
 * 1.5 pt stroke with projecting caps is half_width_px = cap_px = 1.347 px there (nextbestpath_amd/utility/hipops.py::
 * reference_figure_geometry).  cap_px = 0: round stroke ends, as nbp_slice_obstacle_f32.  What stays unpinned is Agg's
 * anti-aliasing and PIL's Lanczos filter themselves: the labels agree with the reference's to within one pixel of line position
 * (tests/test_training_data_cpu.py), not pixel for pixel. */
int nbp_slice_obstacle_fig_f32(const float* verts, const int* faces, int n_faces, float y0, float cx, float cz,
                               int S, float half_u, float scale_u, float half_v, float scale_v, float half_width_px,
                               float cap_px, float* out, void* stream);
/* Depth-map space carving of proxy points (A20): Camera.get_points_in_fov (mu:2849-2884) +
 * get_signed_distance_to_depth_maps (mu:2900-2949) + Scene.update_proxy_supervision_occ /
 * update_proxy_out_of_field (mu:3329-3363) fused per point.  For each proxy point inside the frustum
 * and closer than fov_range: sd = z_view - bilinear(depth) (invalid pixels = 1.1 zfar);
 * n_inside += 1; n_behind += (sd >= -tol); occ = (n_behind/n_inside >= score_threshold);
 * out_of_field = 0.  cam12 is a HOST array (one camera).  Not reachable from the NBP drivers. */
int nbp_carve_update_f32(const float* proxy_pts3, int P, const float* depth,
                         const unsigned char* mask_or_null, const float* cam12_host, int H, int W,
                         float tan_half_fov, float zfar, float fov_range, float tol,
                         float score_threshold, float* n_inside, float* n_behind, float* occ,
                         float* out_of_field, void* stream);
/* View-state vectors of the proxy points -- compute_view_state (macarons/utility/scone_utils.py:799-862) as
 * Scene.update_proxy_view_states applies it (macarons/utility/macarons_utils.py:3268-3327): for every selected point and every
 * camera position x_view_host [n_view <= 8][3] (HOST array), the direction point -> camera in spherical coordinates
 * (macarons/utility/CustomGeometry.py:27-45) is rounded to the nearest of n_elev x n_azim directions and
 * view_states[point][n_elev * n_azim] gets a 1.0 there (the reference's `+= ...; heaviside`, i.e. OR).  A point is selected
 * when mask_or_null[i] != 0 (NULL = all) and sd_or_null[i] < distance_to_surface (NULL = no such test).
 * nbp_carve_view_update_f32 = nbp_carve_update_f32 + that update for ONE camera at x_cam_host[3] in the same launch, over the
 * points inside the field of view with signed distance < distance_to_surface (macarons/testers/scene.py:598-607);
 * fov_mask_or_null [P] / sd_or_null [P] receive the field-of-view mask and the signed distances (sd only where the mask is 1). */
int nbp_view_state_update_f32(const float* pts3, int P, const unsigned char* mask_or_null, const float* sd_or_null,
                              float distance_to_surface, const float* x_view_host, int n_view, int n_elev, int n_azim,
                              float* view_states, void* stream);
/* Geometric coverage-gain model for candidate poses (stand-in for the unreleased SCONE predictor the reference calls at
 * macarons/testers/scene.py:640-670): gains[c] = number of proxy points inside candidate c's field of view (cams12_host [n][12],
 * range fov_range) with occ > 0.5 whose view-state bit for the direction towards x_cams_host[c] is still 0. */
int nbp_view_gain_i32(const float* pts3, int P, const float* occ, const float* view_states, const float* cams12_host,
                      const float* x_cams_host, int n_cams, int n_elev, int n_azim, int H, int W, float tan_half_fov,
                      float fov_range, int* gains, void* stream);
int nbp_carve_view_update_f32(const float* proxy_pts3, int P, const float* depth, const unsigned char* mask_or_null,
                              const float* cam12_host, int H, int W, float tan_half_fov, float zfar, float fov_range,
                              float tol, float score_threshold, float* n_inside, float* n_behind, float* occ,
                              float* out_of_field, const float* x_cam_host, int n_elev, int n_azim,
                              float distance_to_surface, float* view_states, unsigned char* fov_mask_or_null,
                              float* sd_or_null, void* stream);
/* ---- Scene / Cell point store (MACARONS scene objects, macarons/utility/macarons_utils.py:2952-3234), device resident.
 * store_pts [n_cells][capacity][3] fp32 + store_count [n_cells] int32, n_cells = grid3[0]*grid3[1]*grid3[2] in the
 * cartesian product order (i_l, i_w, i_h) of Scene.__init__ (:3063-3066); box6_host = x_min[3], x_max[3] of the scene.
 *
 * nbp_scene_fill_cells_f32 = Scene.fill_cells (:3177-3187) + Cell.fill (:3000-3028): points inside the scene box
 * (inclusive) go to the cell given by floor_divide (macarons/utility/utils.py:113-117) if they lie STRICTLY inside that
 * cell's box; a cell that receives more than n_point_min candidates drops those within `resolution` (fp64 distance,
 * d <= resolution) of a point it already stores -- new points are not compared with each other -- and appends the rest;
 * above `capacity` an exact-size seeded random subset of [stored | new] survives (torch.randperm(len)[:capacity], :3021).
 * n_dev_or_null (device int64) optionally bounds n on the device (cloud counters). */
size_t nbp_scene_fill_workspace_bytes(const float* box6_host, const int* grid3_host, int capacity,
                                      long long n_pts_max, double resolution);
int nbp_scene_fill_cells_f32(const float* pts3, long long n, const long long* n_dev_or_null,
                             const float* box6_host, const int* grid3_host, int capacity, double resolution,
                             int n_point_min, unsigned seed, float* store_pts, int* store_count, void* ws,
                             size_t ws_bytes, void* stream);
/* Scene.return_entire_pt_cloud (:3217-3234): the cells' points back to back, *n_out (device) = their number. */
int nbp_scene_gather_f32(const float* store_pts, const int* store_count, int n_cells, int capacity, float* out3,
                         long long out_capacity, long long* n_out, void* stream);
/* Scene.scene_coverage (:3512-3539): covered_and_total2[0] = number of stored GT points whose nearest recovered point
 * OF THE SAME CELL is closer than epsilon (fp64, strict), [1] = number of stored GT points; coverage = [0] / [1]. */
size_t nbp_scene_coverage_workspace_bytes(const float* box6_host, const int* grid3_host, int capacity_rec,
                                          double epsilon);
int nbp_scene_coverage_f32(const float* gt_pts, const int* gt_count, int capacity_gt, const float* rec_pts,
                           const int* rec_count, int capacity_rec, const float* box6_host, const int* grid3_host,
                           double epsilon, int* covered_and_total2, void* ws, size_t ws_bytes, void* stream);
/* Camera.get_points_in_fov (mu:2849-2884) for n_cams cameras (host [n_cams,12]): mask[cam][i] = 1 iff point i projects
 * inside the image, lies in front of the camera and closer than fov_range to its centre; any[cam] = 1 iff some point
 * does (Camera.is_fov_empty over the mesh vertices, mu:2672-2688).  Either output may be null. */
int nbp_points_in_fov_u8(const float* pts3, int P, const float* cams12_host, int n_cams, int H, int W,
                         float tan_half_fov, float fov_range, unsigned char* mask_or_null, int* any_or_null,
                         void* stream);
/* out3[j] = pc3[perm(j)], j < min(N, k): the first k points of a seeded random permutation of the cloud
 * (fill_surface_scene, mu:715-716); *m_out (device) = their number. */
int nbp_sample_points_f32(const float* pc3, long long N, const long long* N_dev_or_null, long long k, unsigned seed,
                          float* out3, long long* m_out, void* stream);
/* dst[offset + i] = pts3_host[i] for i < n <= 8 (the camera trajectory buffer, without a blocking copy). */
int nbp_append_points_f32(float* dst, long long offset, const float* pts3_host, int n, void* stream);
/* Host mirror of the sampling bijection (driver / tests). */
unsigned nbp_perm_index_host(unsigned j, unsigned n, unsigned seed);

/* ================================================================ A8-A12: planner
 * Obstacle fusion (next_best_path/testers/nbp_planning.py:166-191): obst = out2 >= threshold;
 * where the projection of the whole cloud (maps6 ch0..4 summed) is non-empty take (ch5 > 0);
 * zero where traj > 0.  fullproj = min(sum, 1)  (:172-175). */
int nbp_fuse_obstacle_f32(const float* out2, const float* maps6, const float* traj, float threshold,
                          int S, float* obst, float* fullproj, void* stream);
/* Candidate scoring (nbp_planning.py:203-231 + macarons_utils.py:86-100) for P lattice positions:
 * valid[i] iff the V-grid cell is inside and the 21x21 window of fullproj around the S-grid cell
 * contains a pixel == 1; cell2[i] = V-grid cell; score[i] = max_c out1[c,cell] - 10*fullproj[S cell]
 * in float64 (Python float arithmetic of the reference). */
int nbp_score_candidates_f32(const float* pos3, int P, float cx, float cz, const float* out1, int V,
                             const float* fullproj, int S, float lo, float hi,
                             const unsigned char* skip_or_null, unsigned char* valid, int* cell2,
                             double* score, void* stream);
/* line_across_image_pixel (long_term_utils.py:300-331) for E lattice edges in one launch:
 * blocked iff an endpoint maps outside the S x S image or >= 2 Bresenham pixels equal 1. */
int nbp_edges_blocked_u8(const float* obst, int S, float lo, float hi, float cx, float cz,
                         const float* pos3, const int* edges2, int E, unsigned char* blocked,
                         void* stream);
/* The three of them for the n <= 16 replanning rollouts of a lock-step group in two launches (fusion; scoring + edge mask):
 * HOST arrays of n entries (device pointers / scalars), poses_xz_host [n][2] = (cx, cz), skip entries may be NULL.
 * Identical results to the single calls. */
int nbp_replan_batch_f32(int n, const float* const* out2, const float* const* maps6, const float* const* traj, float threshold,
                         int S, float* const* obst, float* const* fullproj, const float* const* pos3, const int* P,
                         const float* poses_xz_host, const float* const* out1, int V, float lo, float hi,
                         const unsigned char* const* skip, unsigned char* const* valid, int* const* cell2,
                         double* const* score, const int* const* edges2, const int* E, unsigned char* const* blocked,
                         void* stream);
/* HOST function (no kernel, no stream): the candidate loop of nbp_planning.py:233-249 around
 * generate_Dijkstra_path (long_term_utils.py:334-418) on the position lattice, after the three kernels above
 * and their device->host copies.  Nodes 0..P-1 in lexicographic (i,j,k) order (idx3 [P,3], pos3 [P,3] world
 * positions); directed edges edges2 [E,2] grouped by source node in the reference's neighbour order
 * (+x,-x,+z,-z), edge_first [P+1] = first edge of each node.  An edge is usable iff pass_mask[q] or (not
 * blocked[q] and not coll_mask[q]) (ref :350-360); the search tree is the reference's uniform-cost tree (heap
 * ordered by (cost, tuple), came_from fixed at first discovery).  Candidates cand[] are tried in order (already
 * sorted by the caller: stable, score descending): path to the candidate, per node the best-valued heading of
 * out1 [8,V,V] at the node's value cell (cell = rint((-(p - c) - lo) * sc) in fp32) that hist5 [n_hist,5] does
 * not hold yet (ref :390-413), first node dropped (ref :416); when check_first_edge and the first edge crosses
 * the mesh (mesh_hit [E]) the pair is reported in new_coll2 (node ids a,b; the caller appends [a,b] and [b,a] to
 * its collision list) and the next candidate is tried on the rebuilt tree.  Outputs: *path_len = number of nodes
 * written to path_nodes / path_heads (0 = candidate is the start node; -1 = "None": the last candidate tried
 * was unreachable, or there was none); *goal = accepted candidate or -1.  *path_len = -2: a path node lies
 * outside the value map, where the reference draws a random heading -- the caller runs its own (Python) form of
 * this function so that the draw comes from the rollout's random stream; nothing was modified. */
int nbp_plan_search_host(int P, const int* idx3, const float* pos3, int E, const int* edges2,
                         const int* edge_first, const unsigned char* mesh_hit, const unsigned char* blocked,
                         const unsigned char* coll_mask, const unsigned char* pass_mask, const int* cand,
                         int n_cand, int start_id, float cx, float cz, const float* out1, int V, float lo,
                         float sc, const int* hist5, int n_hist, int check_first_edge, int max_path,
                         int* path_nodes, int* path_heads, int* path_len, int* goal, int* new_coll2,
                         int max_new_coll, int* n_new_coll);
/* calculate_coverage_percentage (long_term_utils.py:437-468): *count_out = #{g : min_j |gt_g - s_j|
 * < threshold} where s = the cloud, or a seeded random subset of sample_k points of it when it
 * has more (random_sample_pc, :437-447); *m_out = number of points used.  bbox_* = bounds of gt. */
size_t nbp_coverage_workspace_bytes(const float* bbox_lo_host, const float* bbox_hi_host,
                                    float threshold, long long sample_k);
int nbp_coverage_count_f32(const float* gt3, int G, const float* pc3, long long N,
                           const long long* N_dev_or_null, long long sample_k, unsigned seed,
                           float threshold, const float* bbox_lo_host, const float* bbox_hi_host,
                           int* count_out, int* m_out, void* ws, size_t ws_bytes, void* stream);
/* The same count when the GT cloud is reused over many calls (one rollout): nbp_coverage_plan_build_f32 sorts gt into a
 * grid once (plan = caller-owned device buffer of nbp_coverage_plan_bytes); nbp_coverage_count_planned_f32 then runs one
 * kernel over the sampled cloud points.  It ADDS to *count_accum (caller zeroes it) and needs an `epoch` > 0 that differs
 * between consecutive calls on the same plan (per-GT-point stamps instead of a clear). */
size_t nbp_coverage_plan_bytes(const float* bbox_lo_host, const float* bbox_hi_host, float threshold, int G);
size_t nbp_coverage_plan_workspace_bytes(const float* bbox_lo_host, const float* bbox_hi_host, float threshold, int G);
int nbp_coverage_plan_build_f32(const float* gt3, int G, float threshold, const float* bbox_lo_host,
                                const float* bbox_hi_host, void* plan, size_t plan_bytes, void* ws,
                                size_t ws_bytes, void* stream);
int nbp_coverage_count_planned_f32(void* plan, int G, float threshold, const float* bbox_lo_host,
                                   const float* bbox_hi_host, const float* pc3, long long N,
                                   const long long* N_dev_or_null, long long sample_k, unsigned seed,
                                   unsigned epoch, int* count_accum, int* m_out, void* stream);

/* ================================================================ A2-A3: training step
 * Kernels behind the autograd functions of nextbestpath_amd/networks/training.py, which replace
 * autograd + cuDNN under train_experience_data (next_best_path/utility/nbp_utils.py:340-395) for
 * the layers of next_best_path/networks/nbp_model.py:8-62 in nbp.train() mode.  All [M, C] tensors
 * are NHWC with M = B*H*W.  Reductions are two-stage => deterministic. */
size_t nbp_colreduce_workspace_bytes(long long M, int C);
/* nn.BatchNorm2d training forward (eps, momentum = module values): batch mean / biased variance,
 * running-stat update (unbiased variance; pass NULL to skip), y = [relu]((x-mean)*invstd*gamma+beta);
 * mean / invstd are saved for the backward. */
int nbp_bn_train_forward_f32(const float* x, long long M, int C, const float* gamma, const float* beta,
                             float eps, float momentum, float* running_mean, float* running_var,
                             int relu, float* mean, float* invstd, float* y, void* ws, size_t ws_bytes,
                             void* stream);
/* Backward of the above (+ fused ReLU mask from y when relu != 0): dx, dgamma, dbeta. */
int nbp_bn_train_backward_f32(const float* dy, const float* x, const float* y_or_null, long long M, int C,
                              const float* mean, const float* invstd, const float* gamma, int relu,
                              float* dx, float* dgamma, float* dbeta, void* ws, size_t ws_bytes,
                              void* stream);
/* The same two with their consumers' passes folded in (C % 4 == 0; nbp_bn_backward_fuses(C) tells): the forward also writes
 * max |y| into amax_out (64 zeroed words, float bits by atomicMax: the operand scale of the split convolution that reads y); the
 * backward also writes dx_colsum[c] = sum_m dx[m][c] (the bias gradient of the convolution in front of this BatchNorm) and
 * max |dx| into amax_out (the scale of that convolution's data / weight gradients) -- in the pass that writes dx, instead of
 * two more reads of it.  NULL outputs are skipped. */
int nbp_bn_train_forward_amax_f32(const float* x, long long M, int C, const float* gamma, const float* beta,
                                  float eps, float momentum, float* running_mean, float* running_var,
                                  int relu, float* mean, float* invstd, float* y, void* amax_out, void* ws,
                                  size_t ws_bytes, void* stream);
int nbp_bn_backward_fuses(int C);
int nbp_bn_train_backward_fused_f32(const float* dy, const float* x, const float* y_or_null, long long M, int C,
                                    const float* mean, const float* invstd, const float* gamma, int relu,
                                    float* dx, float* dgamma, float* dbeta, float* dx_colsum, void* amax_out,
                                    void* ws, size_t ws_bytes, void* stream);
/* Round 4: the backward without reading y.  nbp_bn_train_forward_stat4_f32 = nbp_bn_train_forward_amax_f32 that also hands out the
 * UNROUNDED statistics the normalisation used (stat_out: [4 C] doubles, 32-byte aligned: mean | invstd | lo | hi, the last two with
 * relu only: y > 0 <=> lo <= x <= hi exactly, two floats per channel found by bisection with the forward's own arithmetic);
 * nbp_bn_train_backward_stat_f32 = nbp_bn_train_backward_fused_f32 whose ReLU mask (y > 0) is rebuilt from x -- which the pass
 * reads anyway -- through the forward's own arithmetic on those statistics (C % 4 == 0): two tensor reads less per BatchNorm and
 * step, the same mask bit for bit. */
/* (round 6: `stat_doubles` = the doubles behind stat_out, NBP_E_WS below 4 C; without relu the two bounds are written as
 * (-inf, +inf).  The names changed with the signature -- ..._stat_f32 / ..._part_f32 of rounds 4-5 are gone -- so a caller built for the
 * [2 C] contract of round 4 fails to link instead of being overrun.) */
int nbp_bn_train_forward_stat4_f32(const float* x, long long M, int C, const float* gamma, const float* beta,
                                   float eps, float momentum, float* running_mean, float* running_var,
                                   int relu, float* mean, float* invstd, float* y, void* amax_out, double* stat_out, int stat_doubles,
                                   void* ws, size_t ws_bytes, void* stream);
int nbp_bn_train_backward_stat_f32(const float* dy, const float* x, const double* stat_d, const float* beta, long long M, int C,
                                   const float* mean, const float* invstd, const float* gamma, int relu,
                                   float* dx, float* dgamma, float* dbeta, float* dx_colsum, void* amax_out,
                                   void* ws, size_t ws_bytes, void* stream);
/* out[c] = sum_m rows[m]*x[m][c] (rows NULL = 1): conv bias gradients, psi weight gradient. */
int nbp_colsum_f32(const float* x, const float* rows_or_null, long long M, int C, float* out, void* ws,
                   size_t ws_bytes, void* stream);
/* op 0: relu(a+b)  1: a*(b>0)  2: sigmoid(a)  3: a*b*(1-b)  4: a+b  5: a+b[0] */
int nbp_elementwise_f32(int op, const float* a, const float* b, long long n, float* out, void* stream);
/* out[m][c] = sum_k src_k[m ld_k + c] (k = 0 .. n - 1, left to right in fp32; n <= 8; C % 4 == 0; 16-byte aligned pointers; ld_k >= C
 * floats, a multiple of 4: a source may be a channel slice of a wider NHWC tensor).  The gradient of an activation with n consumers
 * (a skip connection: max-pool + two attention gates x two uses) in one pass instead of autograd's n - 1 binary adds -- what
 * loss.backward() does implicitly in next_best_path/utility/nbp_utils.py:383.  srcs_host / ld_host: HOST arrays of n entries. */
int nbp_sum_n_f32(int n, const float* const* srcs_host, const long long* ld_host, long long M, int C, float* out, void* stream);
/* The element-wise middle of Attention_block in TRAINING mode (nbp_model.py:52-60), between the two 1x1 convolutions and x * psi:
 *     g1 = BN_g(g_pre);  x1 = BN_x(x_pre);  q = relu(g1 + x1);  p = q . w_psi + b_psi        (batch statistics, running stats updated)
 * fused: two statistics passes + ONE pass that reads g_pre / x_pre and writes q [M,F] and p [M] (the separate launches were BatchNorm
 * apply twice, add-relu and a row-dot: 8 tensor passes for 3), and its backward from dp = dL/dp: both BatchNorm backwards, the ReLU
 * mask, dq = dp (x) w_psi and the psi weight's gradient in one reduce + one apply pass (15 tensor passes for 8).  Every output is the
 * separate launches' bit for bit (p sums its row in nbp_rowdot_f32's order).  F = F_int: F / 4 a power of two in [4, 64].  stat_g / stat_x: [4 F] doubles, 32-byte aligned; csum_* = column sums of d g_pre / d x_pre (the 1x1 convolutions' bias
 * gradients), amax_* (or NULL) = 64 zeroed words receiving max |.| of them; ws >= nbp_gate_mid_workspace_bytes(M, F). */
size_t nbp_gate_mid_workspace_bytes(long long M, int F);
int nbp_gate_mid_forward_f32(const float* g_pre, const float* x_pre, long long M, int F,
                             const float* gamma_g, const float* beta_g, float* run_mean_g, float* run_var_g, float eps_g, float mom_g,
                             const float* gamma_x, const float* beta_x, float* run_mean_x, float* run_var_x, float eps_x, float mom_x,
                             float* mean_g, float* invstd_g, float* mean_x, float* invstd_x, double* stat_g, double* stat_x,
                             const float* w_psi, const float* b_psi, float* q, float* p, void* ws, size_t ws_bytes, void* stream);
int nbp_gate_mid_backward_f32(const float* dp, const float* w_psi, const float* q, const float* g_pre, const float* x_pre,
                              long long M, int F, const float* mean_g, const float* invstd_g, const float* gamma_g,
                              const float* mean_x, const float* invstd_x, const float* gamma_x, float* dg_pre, float* dx_pre,
                              float* dgamma_g, float* dbeta_g, float* dgamma_x, float* dbeta_x, float* dw_psi, float* csum_g,
                              float* csum_x, void* amax_g, void* amax_x, void* ws, size_t ws_bytes, void* stream);
int nbp_rowscale_f32(const float* x, const float* s, long long M, int C, float* out, void* stream);
/* ... that also leaves max |out| in the 64 zeroed words of amax_out (C % 4 == 0, 16-byte aligned tensors, else NBP_E_SHAPE). */
int nbp_rowscale_amax_f32(const float* x, const float* s, long long M, int C, float* out, void* amax_out, void* stream);   /* x[m][c]*s[m] */
/* backward of out = x * s[m] in one pass: dx[m][c] = dy[m][c] * s[m], ds[m] = sum_c dy[m][c] * x[m][c]; dy's rows are ldy >= C floats
 * apart (a channel slice of a wider gradient is read in place); C, ldy multiples of 4, 16-byte aligned pointers (else NBP_E_SHAPE) */
int nbp_rowscale_backward_f32(const float* dy, long long ldy, const float* x, const float* s, long long M, int C, float* dx,
                              float* ds, void* stream);
int nbp_rowdot_f32(const float* a, const float* b, int b_is_vector, long long M, int C, float* out,
                   void* stream);                                                                     /* sum_c a*b   */
int nbp_outer_f32(const float* s, const float* w, long long M, int C, float* out, void* stream);      /* s[m]*w[c]   */
/* nn.MaxPool2d(2,2) backward (gradient to the first maximum of each window, ATen rule). */
int nbp_maxpool2_backward_f32(const float* x, const float* dy, int B, int H, int W, int C, float* dx,
                              void* stream);
/* nn.Upsample(x2, nearest) backward: 2x2 block sums, dy [B,2Hs,2Ws,C] -> [B,Hs,Ws,C]. */
int nbp_sum2x2_f32(const float* dy, int B, int Hs, int Ws, int C, float* out, void* stream);
int nbp_slice_channels_f32(const float* in, long long M, int Cin, int c0, int Cs, float* out, void* stream);
int nbp_pad_channels_f32(const float* in, long long M, int Cin, int Cout, float* out, void* stream);
/* Forward packing with zero padding of both channel counts: dst[ci/32][tap][Npad][32]. */
int nbp_pack_conv_weight_padded(const float* w_oihw, int N, int C, int ksize, int Cpad, int Npad, float* dst,
                                void* stream);
/* Data gradient = nbp_conv_igemm_f32 on dY with these weights: flipped taps, co <-> ci transposed,
 * zero padded to (Cpad, Npad) multiples of 32: dst[co/32][tap][ci][co%32]. */
int nbp_pack_conv_weight_dgrad(const float* w_oihw, int N, int C, int ksize, int Cpad, int Npad, float* dst,
                               void* stream);
/* Weight gradient on the matrix cores: dW [n_real][c_real][k][k] (OIHW) of
 * out = conv(cat(src0, src1) [x2 upsampled]) given dY [B,H,W,N]; C0, C1, N multiples of 64. */
size_t nbp_conv_wgrad_workspace_bytes(int B, int H, int W, int C0, int C1, int N, int ksize);
int nbp_conv_wgrad_f32(const float* src0, int C0, const float* src1, int C1, int ups, int B, int H, int W,
                       int ksize, const float* dy, int N, int c_real, int n_real, float* dw, void* ws,
                       size_t ws_bytes, void* stream);
/* nbp_conv_wgrad_f32 with the products of the 3x3 layers on the fp16 matrix pipe: every fp32 operand (X, dY) scaled by a
 * power of two from its tensor's max |.| (computed inside), cut into two fp16 pieces, three exact MFMAs per product, fp32
 * accumulation (the scheme of nbp_conv3x3_split_f32; error vs fp64 <= the fp32 MFMA pipe's).  amax*_or_null: 64-word max-|.|
 * slots (nbp_amax_f32) of src0 / src1 / dy when the caller has them (any upper bound of the tensor's max works: e.g. the forward's
 * joint slot for both sources), else taken inside.  Same other arguments and workspace; layers the split kernel does not take (1x1, W % 32 != 0) run nbp_conv_wgrad_f32. */
int nbp_conv_wgrad_split_f32(const float* src0, int C0, const float* src1, int C1, int ups, int B, int H, int W,
                             int ksize, const float* dy, int N, int c_real, int n_real, float* dw,
                             const void* amax0_or_null, const void* amax1_or_null, const void* amaxy_or_null, void* ws,
                             size_t ws_bytes, void* stream);
/* Sparse value targets (nbp_utils.py:373-379): pred[k] = out1[b,c,x,y], coords [K,4] int64; and the
 * scatter-add of its gradient into a zeroed d_out1. */
int nbp_gather_values_f32(const float* out1_nchw, const long long* coords_bcxy, int K, int C, int H, int W,
                          float* pred, void* stream);
int nbp_scatter_values_f32(const float* dpred, const long long* coords_bcxy, int K, int C, int H, int W,
                           float* dout1_nchw_zeroed, void* stream);
/* mode 0: sum (p-t)^2, mode 1: sum BCE(p,t) (log clamped at -100 like torch) into *sum_out (device
 * double); dp (optional) = grad_coef * d(mean loss)/dp. */
int nbp_loss_f32(int mode, const float* p, const float* t, long long n, float grad_coef, double* sum_out,
                 float* dp_or_null, void* ws, size_t ws_bytes, void* stream);

/* ---- The replay store's container in LMDB's on-disk format (csrc/nbp_mdb.cpp; host only).  The reference keeps its experience
 * records in an LMDB environment (next_best_path/trainers/train_nbp_model.py:61-63 lmdb.open(path, map_size);
 * next_best_path/utility/nbp_utils.py:32-141: txn.put per record, ordered cursors, txn.delete of the validation records).  liblmdb is
 * not in this image: these entry points read and write <dir>/data.mdb in LMDB 0.9's file format (4096-byte pages, unnamed main
 * database, memcmp key order, overflow pages for records > 2038 bytes), so that a store written here opens with lmdb and one written
 * by the reference opens here.  One writer; every put / del is its own committed transaction.  Parity unpinned against liblmdb.
 *   nbp_mdb_open     creates <dir> and an empty environment, or loads an existing one (NBP_E_SHAPE: not an LMDB file of this shape --
 *                    other page size, named or duplicate-key databases); sync_each_commit: fdatasync before and after each meta write
 *   nbp_mdb_put      insert or replace; nbp_mdb_del: 0 deleted, 1 not found; nbp_mdb_get: 0 found (*vlen_out = size; the bytes are
 *                    copied when cap >= size), 1 not found
 *   nbp_mdb_keys     every key in order, packed [u16 length][bytes]...; *needed_out = packed size (cap = 0 sizes the buffer)
 *   nbp_mdb_stat     {depth, branch pages, leaf pages, overflow pages, entries, last page, txnid, page size}                          */
int nbp_mdb_open(const char* dir_path, unsigned long long map_size, int sync_each_commit, void** env_out);
int nbp_mdb_close(void* env);
long long nbp_mdb_entries(void* env);
int nbp_mdb_put(void* env, const void* key, size_t klen, const void* val, size_t vlen);
int nbp_mdb_del(void* env, const void* key, size_t klen);
int nbp_mdb_get(void* env, const void* key, size_t klen, void* buf, size_t cap, size_t* vlen_out);
int nbp_mdb_keys(void* env, void* buf, size_t cap, size_t* needed_out);
int nbp_mdb_stat(void* env, unsigned long long* out8);

#ifdef __cplusplus
}
#endif
#endif /* NBP_HIP_H */
