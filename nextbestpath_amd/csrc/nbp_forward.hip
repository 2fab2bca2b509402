// nbp_forward.hip -- weight packing + the whole-network forward of NBP on gfx950.
//
// Replaces NBP.forward, next_best_path/networks/nbp_model.py:110-160: one C call enqueues
// every kernel of the Attention U-Net (shared encoder, two decoders) on one stream.
// Fusions relative to the reference's op list: BatchNorm+bias+ReLU in the conv epilogue,
// nn.Upsample and torch.cat in the conv's operand gather, the attention gate's two 1x1
// convolutions as ONE GEMM over K=[g|x] followed by a psi+multiply tail.
#include "common.h"
#include "nbp_internal.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace {

enum Kind { K_FIRST, K_CONV3, K_ATT_G, K_ATT_X, K_PSI, K_FINAL };
struct LayerSpec { Kind kind; int cin, cout, k; };

// canonical order documented in include/nbp_hip.h
struct Table {
    LayerSpec L[NBP_N_CONV];
    size_t w_off[NBP_N_CONV];   // float offsets into the packed buffer
    size_t s_off[NBP_N_CONV], t_off[NBP_N_CONV];
    size_t total_floats;
    size_t w3_off[NBP_N_CONV];  // byte offsets of the split planes (3x3 layers) in the split region after the fp32 pack
    size_t w3u_off[NBP_N_CONV]; // ... of the parity filters of the up_conv layers (is_up)
    bool is_up[NBP_N_CONV];
    int up_rank[NBP_N_CONV];    // 0..5 among the up_conv layers (its max |w| word is header word NBP_N_CONV + rank)
    size_t total3_bytes;
    size_t wup16_off[NBP_N_CONV], total_up16_bytes;
    size_t w1_off[NBP_N_CONV];  // split region: planes of the attention gates' joint 1x1 GEMM (K_ATT_G layers)
    int gate_rank[NBP_N_CONV];  // its max |w| word: header word NBP_N_CONV + 6 + rank
    Table() {
        int i = 0;
        const int enc[5] = {64, 128, 256, 512, 1024};
        int cin = 5;
        for (int e = 0; e < 5; ++e) {
            L[i++] = {e == 0 ? K_FIRST : K_CONV3, cin, enc[e], 3};
            L[i++] = {K_CONV3, enc[e], enc[e], 3};
            cin = enc[e];
        }
        for (int j = 0; j < NBP_N_CONV; ++j) { is_up[j] = false; up_rank[j] = -1; }
        int n_up = 0;
        auto level = [&](int ci, int co) {
            is_up[i] = true; up_rank[i] = n_up++;
            L[i++] = {K_CONV3, ci, co, 3};         // Up{L}_d.up.1
            L[i++] = {K_ATT_G, co, co / 2, 1};     // Att W_g
            L[i++] = {K_ATT_X, co, co / 2, 1};     // Att W_x
            L[i++] = {K_PSI, co / 2, 1, 1};        // Att psi
            L[i++] = {K_CONV3, ci, co, 3};         // Up_conv.conv.0 (cat(a, d) -> co)
            L[i++] = {K_CONV3, co, co, 3};         // Up_conv.conv.3
        };
        level(1024, 512); level(512, 256);                                   // decoder 1
        level(1024, 512); level(512, 256); level(256, 128); level(128, 64);  // decoder 2
        L[i++] = {K_FINAL, 256, 8, 1};
        L[i++] = {K_FINAL, 64, 1, 1};
        size_t off = 0;
        auto take = [&](size_t n) { size_t o = off; off += (n + 63) / 64 * 64; return o; };  // 256 B aligned
        for (int j = 0; j < NBP_N_CONV; ++j) {
            const LayerSpec& s = L[j];
            size_t wn = 0;
            switch (s.kind) {
                case K_FIRST: case K_PSI: case K_FINAL: wn = (size_t)s.cout * s.cin * s.k * s.k; break;
                case K_CONV3: wn = (size_t)s.cout * s.cin * 9; break;
                case K_ATT_G: wn = (size_t)s.cout * s.cin * 2; break;   // joint [g|x] K
                case K_ATT_X: wn = 0; break;
            }
            w_off[j] = take(wn);
            s_off[j] = take(s.cout);
            t_off[j] = take(s.cout);
        }
        total_floats = off;
        size_t off3 = 256;           // head of the split region: max |w| of each layer (NBP_N_CONV words)
        for (int j = 0; j < NBP_N_CONV; ++j) {
            w3_off[j] = off3;
            if (L[j].kind == K_CONV3) off3 += ((size_t)L[j].cout * L[j].cin * 9 * 4 + 255) / 256 * 256;
        }
        for (int j = 0; j < NBP_N_CONV; ++j) {
            w3u_off[j] = off3;
            if (is_up[j]) off3 += ((size_t)L[j].cout * L[j].cin * 16 * 4 + 255) / 256 * 256;
        }
        int n_gate = 0;
        for (int j = 0; j < NBP_N_CONV; ++j) {
            w1_off[j] = off3; gate_rank[j] = -1;
            if (L[j].kind == K_ATT_G) { gate_rank[j] = n_gate++; off3 += ((size_t)L[j].cout * L[j].cin * 2 * 4 + 255) / 256 * 256; }
        }
        total3_bytes = off3;
        size_t offu = 0;             // bf16 handle: parity filters of the up_conv layers, after the common pack
        for (int j = 0; j < NBP_N_CONV; ++j) {
            wup16_off[j] = offu;
            if (is_up[j]) offu += ((size_t)L[j].cout * L[j].cin * 16 * 2 + 255) / 256 * 256;
        }
        total_up16_bytes = offu;
    }
};
const Table& table() { static Table t; return t; }

__global__ void copy_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        dst[i] = src[i];
}
int copy_f32(const float* src, float* dst, long long n, hipStream_t st) {
    copy_f32_kernel<<<nbp_ew_grid(n, 256), 256, 0, st>>>(src, dst, n);
    return nbp_launch_status();
}

__global__ void fill_f32_kernel(float* __restrict__ dst, float v, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        dst[i] = v;
}
int fill_f32(float* dst, float v, long long n, hipStream_t st) {
    fill_f32_kernel<<<nbp_ew_grid(n, 256), 256, 0, st>>>(dst, v, n);
    return nbp_launch_status();
}

}  // namespace

struct nbp_weights {
    const void* w[NBP_N_CONV];      // fp32 everywhere, except bf16 for the igemm layers of a bf16 handle
    const float* scale[NBP_N_CONV];
    const float* shift[NBP_N_CONV];
    int bf16;
    const void* w3[NBP_N_CONV];     // split handle: hi/lo fp16 planes of the 3x3 layers (nbp_split.hip)
    const unsigned* wamax[NBP_N_CONV];      // ... and max |w| of each (device words, float bits)
    const void* wup16[NBP_N_CONV];          // bf16 handle, up_conv layers: the four parity filters (null elsewhere)
    const void* w1[NBP_N_CONV];             // split handle, attention gates: planes of the joint 1x1 GEMM (null elsewhere)
    const unsigned* wamax1[NBP_N_CONV];
    const void* w3u[NBP_N_CONV];            // up_conv layers: planes of the four parity filters (null elsewhere)
    const unsigned* wamax_u[NBP_N_CONV];
    int split;
};

extern "C" int nbp_abi_version(void) {
    NBP_ENTER(); return NBP_ABI_VERSION; }

extern "C" int nbp_device_info(char* arch_host, int arch_len, int* cu_count_host) {
    NBP_ENTER();
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    hipDeviceProp_t p;
    e = hipGetDeviceProperties(&p, dev);
    if (e != hipSuccess) return (int)e;
    if (arch_host && arch_len > 0) { strncpy(arch_host, p.gcnArchName, arch_len - 1); arch_host[arch_len - 1] = 0; }
    if (cu_count_host) *cu_count_host = p.multiProcessorCount;
    return 0;
}

extern "C" size_t nbp_packed_weights_bytes(void) { return table().total_floats * sizeof(float); }

static int pack_weights_impl(const void* const* w_host_array, const void* const* scale_host_array,
                             const void* const* shift_host_array, void* packed, size_t packed_bytes, void* stream,
                             nbp_weights** handle_out, bool bf16, bool split = false) {
    NBP_RETURN_IF(!w_host_array || !scale_host_array || !shift_host_array || !packed || !handle_out, NBP_E_ARG);
    const Table& T = table();
    NBP_RETURN_IF(packed_bytes < T.total_floats * sizeof(float) + (split ? T.total3_bytes : 0) + (bf16 ? T.total_up16_bytes : 0),
                  NBP_E_WS);
    for (int i = 0; i < NBP_N_CONV; ++i) {
        NBP_RETURN_IF(!w_host_array[i] || !scale_host_array[i], NBP_E_ARG);
        NBP_RETURN_IF(T.L[i].kind != K_ATT_X && !shift_host_array[i], NBP_E_ARG);
    }
    hipStream_t st = (hipStream_t)stream;
    float* base = (float*)packed;
    nbp_weights* h = (nbp_weights*)calloc(1, sizeof(nbp_weights));
    NBP_RETURN_IF(!h, NBP_E_ARG);
    h->bf16 = bf16 ? 1 : 0;
    h->split = split ? 1 : 0;
    char* base3 = (char*)packed + T.total_floats * sizeof(float);
    int rc = 0;
    for (int i = 0; i < NBP_N_CONV && !rc; ++i) {
        const LayerSpec& s = T.L[i];
        const float* w = (const float*)w_host_array[i];
        const float* sc = (const float*)scale_host_array[i];
        const float* sh = (const float*)shift_host_array[i];
        float* wd = base + T.w_off[i];
        float* sd = base + T.s_off[i];
        float* td = base + T.t_off[i];
        h->w[i] = wd; h->scale[i] = sd; h->shift[i] = td;
        switch (s.kind) {
            case K_FIRST: case K_FINAL:
                rc = copy_f32(w, wd, (long long)s.cout * s.cin * s.k * s.k, st);
                if (!rc) rc = copy_f32(sc, sd, s.cout, st);
                if (!rc) rc = copy_f32(sh, td, s.cout, st);
                break;
            case K_PSI:   // psi tail reads {scale, shift} as two consecutive floats
                rc = copy_f32(w, wd, s.cin, st);
                if (!rc) rc = copy_f32(sc, sd, 1, st);
                if (!rc) rc = copy_f32(sh, sd + 1, 1, st);
                if (!rc) rc = copy_f32(sh, td, 1, st);
                break;
            case K_CONV3:
                rc = bf16 ? nbp_pack_conv_weight_bf16(w, s.cout, s.cin, 3, nullptr, 0, s.cin, (bf16_t*)wd, st)
                          : nbp_pack_conv_weight(w, s.cout, s.cin, 3, nullptr, 0, s.cin, wd, st);
                if (!rc) rc = copy_f32(sc, sd, s.cout, st);
                if (!rc) rc = copy_f32(sh, td, s.cout, st);
                if (!rc && bf16 && T.is_up[i]) {
                    h->wup16[i] = base3 + T.wup16_off[i];
                    rc = nbp_pack_upconv_weight_bf16_launch(w, s.cout, s.cin, (bf16_t*)(base3 + T.wup16_off[i]), st);
                }
                if (!rc && split) {
                    h->w3[i] = base3 + T.w3_off[i];
                    h->wamax[i] = (const unsigned*)base3 + i;
                    rc = nbp_pack_conv_weight_split_launch(w, s.cout, s.cin, 3, nullptr, 0, s.cin, base3 + T.w3_off[i],
                                                           (unsigned*)base3 + i, st);
                    if (!rc && T.is_up[i]) {
                        h->w3u[i] = base3 + T.w3u_off[i];
                        h->wamax_u[i] = (const unsigned*)base3 + NBP_N_CONV + T.up_rank[i];
                        rc = nbp_pack_upconv_weight_split_launch(w, s.cout, s.cin, base3 + T.w3u_off[i],
                                                                 (unsigned*)base3 + NBP_N_CONV + T.up_rank[i], st);
                    }
                }
                break;
            case K_ATT_G: {
                // joint GEMM over K=[g|x]: scale folded into the weights, epilogue scale = 1
                const float* wx = (const float*)w_host_array[i + 1];
                const float* scx = (const float*)scale_host_array[i + 1];
                if (bf16) {
                    rc = nbp_pack_conv_weight_bf16(w, s.cout, s.cin, 1, sc, 0, 2 * s.cin, (bf16_t*)wd, st);
                    if (!rc) rc = nbp_pack_conv_weight_bf16(wx, s.cout, s.cin, 1, scx, s.cin, 2 * s.cin, (bf16_t*)wd, st);
                } else {
                    rc = nbp_pack_conv_weight(w, s.cout, s.cin, 1, sc, 0, 2 * s.cin, wd, st);
                    if (!rc) rc = nbp_pack_conv_weight(wx, s.cout, s.cin, 1, scx, s.cin, 2 * s.cin, wd, st);
                }
                if (!rc) rc = fill_f32(sd, 1.0f, s.cout, st);
                if (!rc) rc = copy_f32(sh, td, s.cout, st);
                if (!rc && split) {
                    h->w1[i] = base3 + T.w1_off[i];
                    h->wamax1[i] = (const unsigned*)base3 + NBP_N_CONV + 6 + T.gate_rank[i];
                    rc = nbp_pack_gate_weight_split_launch(w, sc, wx, scx, s.cout, s.cin, base3 + T.w1_off[i],
                                                           (unsigned*)base3 + NBP_N_CONV + 6 + T.gate_rank[i], st);
                }
                break;
            }
            case K_ATT_X:
                h->w[i] = nullptr;
                break;
        }
    }
    if (rc) { free(h); return rc; }
    *handle_out = h;
    return 0;
}

extern "C" int nbp_pack_weights(const void* const* w_host_array, const void* const* scale_host_array,
                                const void* const* shift_host_array, void* packed, size_t packed_bytes, void* stream,
                                nbp_weights** handle_out) {
    NBP_ENTER();
    return pack_weights_impl(w_host_array, scale_host_array, shift_host_array, packed, packed_bytes, stream, handle_out,
                             false);
}

// Same inputs (fp32 OIHW weights, folded fp32 scale / shift); the 3x3 and attention-gate weights are stored as
// bf16 in the 64-channel chunk layout of nbp_bf16.hip, everything else (first conv, psi, heads, epilogues) stays fp32.
extern "C" int nbp_pack_weights_bf16(const void* const* w_host_array, const void* const* scale_host_array,
                                     const void* const* shift_host_array, void* packed, size_t packed_bytes,
                                     void* stream, nbp_weights** handle_out) {
    NBP_ENTER();
    return pack_weights_impl(w_host_array, scale_host_array, shift_host_array, packed, packed_bytes, stream, handle_out,
                             true);
}

// fp32 pack + the hi/mid/lo bf16 planes of every 3x3 layer: the handle serves nbp_forward_f32 and nbp_forward_split_f32
extern "C" size_t nbp_packed_weights_bytes_split(void) { return table().total_floats * sizeof(float) + table().total3_bytes; }
extern "C" int nbp_pack_weights_split(const void* const* w_host_array, const void* const* scale_host_array,
                                      const void* const* shift_host_array, void* packed, size_t packed_bytes,
                                      void* stream, nbp_weights** handle_out) {
    NBP_ENTER();
    return pack_weights_impl(w_host_array, scale_host_array, shift_host_array, packed, packed_bytes, stream, handle_out,
                             false, true);
}

extern "C" size_t nbp_packed_weights_bytes_bf16(void) { return table().total_floats * sizeof(float) + table().total_up16_bytes; }

extern "C" void nbp_free_weights(nbp_weights* handle) { free(handle); }

// ------------------------------------------------------------------ forward
namespace {

struct Bump {
    char* base; size_t size, off; bool dry;
    template <typename T> T* take(size_t count) {
        size_t bytes = (count * sizeof(T) + 255) / 256 * 256;
        size_t o = off; off += bytes;
        if (dry) return (T*)(uintptr_t)256;   // non-null dummy
        return (T*)(base + o);
    }
};

struct Timer {
    hipStream_t st;
    nbp_layer_timing* out; int cap; int n;
    hipEvent_t ev[256];
    int nev;
    int begin() {
        nev = 0; n = 0;
        hipError_t e = hipEventCreate(&ev[0]);
        if (e != hipSuccess) return (int)e;
        nev = 1;
        return (int)hipEventRecord(ev[0], st);
    }
    void mark(const char* name, double flops, int tile, int split_k, long long M, int N, int K) {
        if (n >= cap || nev >= 256) return;
        if (hipEventCreate(&ev[nev]) != hipSuccess) return;
        (void)hipEventRecord(ev[nev], st);
        ++nev;
        nbp_layer_timing& t = out[n++];
        memset(&t, 0, sizeof(t));
        strncpy(t.name, name, sizeof(t.name) - 1);
        t.flops = flops; t.tile = tile; t.split_k = split_k; t.M = M; t.N = N; t.K = K;
    }
    int finish() {
        hipError_t e = hipStreamSynchronize(st);
        for (int i = 0; i + 1 < nev && i < n; ++i) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
            out[i].ms = ms;
        }
        for (int i = 0; i < nev; ++i) (void)hipEventDestroy(ev[i]);
        return (int)e;
    }
};

// The two arithmetic paths behind the same driver: element type of the activations, K-chunk width, launchers.
struct NoCtx {
    int init(Bump&, hipStream_t, bool) { return 0; }
    void offer_gated(void*, void*) {}
    bool took_gated() { return false; }
    void offer_pool(void*) {}
    bool took_pool() { return false; }
    void offer_head(const float*, const float*, const float*, float*) {}
    bool took_head() { return false; }
};
struct PathF32 {
    typedef float T;
    typedef ConvOperands Ops;
    typedef NoCtx Ctx;
    static constexpr int CHUNK = 32;
    static ConvPlan plan(long long M, int N, int chunks, int groups, int H = 0, int ksize = 0, int ups = 0) {
        (void)ups;
        return nbp_plan_conv(M, N, chunks, 0, 0, groups, H, H, ksize);
    }
    static constexpr int MODE = 0;
    template <typename C>
    static int conv(C&, const nbp_weights*, const int*, const Ops& o, const Ops* o2, int C0, int C1, int ups, int B, int H, int ks,
                    int N, void* ws, size_t wsb, hipStream_t st) {
        return nbp_conv_igemm_launch_g(o, o2, C0, C1, ups, B, H, H, ks, N, 1, 0, 0, ws, wsb, st);
    }
    template <typename C>
    static int first(C&, const float* x, int B, int s, const nbp_weights* h, T* out, hipStream_t st) {
        return nbp_conv_first_f32(x, B, s, s, (const float*)h->w[0], h->scale[0], h->shift[0], out, st);
    }
    template <typename C>
    static int pool(C&, const T* in, int B, int H, int Cn, T* out, hipStream_t st) { return nbp_maxpool2_nhwc_f32(in, B, H, H, Cn, out, st); }
    template <typename C>
    static int gate(C&, const T* q, int F, const float* w, const float* s2, const T* x, int Cn, long long M, T* out, hipStream_t st) {
        return nbp_psi_gate_f32(q, F, w, s2, x, Cn, M, out, st);
    }
    static int head(const T* in, int B, int H, int C, const float* w, int no, const float* sc, const float* sh, int sig,
                    float* out, hipStream_t st) {
        return nbp_final_1x1_f32(in, B, H, H, C, w, no, sc, sh, sig, out, st);
    }
};
// fp32 tensors; the 3x3 layers the split kernel takes run on the fp16 matrix pipe (two-piece operand splitting, scales from
// each tensor's max |x|), everything else is PathF32's.  AmaxBook keeps one device word per activation tensor: written by the
// split kernel that produces the tensor, inherited through max-pool and the attention gate (|out| <= |in|), computed by a
// standalone pass for tensors that come from the other kernels (first convolution; 1x1 / odd-size fallbacks).
struct AmaxBook {
    unsigned* slots = nullptr;
    int next = 0, n = 0;
    hipStream_t st = nullptr;
    const void* key[128];
    unsigned* val[128];
    int init(Bump& bp, hipStream_t s, bool dry) {
        st = s; next = n = 0;
        slots = bp.take<unsigned>(128 * 64);       // 64 words per tensor (nbp_split.hip: AMAX_WORDS)
        // zeroed by a KERNEL, not hipMemsetAsync: in a captured forward (packing.ForwardGraph) the memset node of ROCm 7.2's
        // graph replay is not ordered against the kernel nodes around it -- replays on new input kept the previous maxima or
        // lost fresh ones (tools/diag/graph_replay_check.py; DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 hid it).  Same cost when eager.
        return dry ? 0 : fill_f32(reinterpret_cast<float*>(slots), 0.f, 128 * 64, st);
    }
    unsigned* find(const void* p) const {
        for (int i = n - 1; i >= 0; --i) if (key[i] == p) return val[i];
        return nullptr;
    }
    void bind(const void* p, unsigned* v) { if (n < 128) { key[n] = p; val[n] = v; ++n; } }
    unsigned* unbound() { return next < 128 ? slots + 64 * next++ : nullptr; }      // a zeroed slot, not yet attached to a tensor
    unsigned* fresh(const void* p) { unsigned* v = next < 128 ? slots + 64 * next++ : nullptr; if (v) bind(p, v); return v; }
    void alias(const void* p, const void* of) { if (unsigned* v = find(of)) bind(p, v); }
    // the decoder offers the gated tensors before it launches a gate's 1x1 GEMM; a launch that runs the psi tail in its
    // epilogue (nbp_split.hip) takes them, and the separate psi kernel is skipped
    float* gated[2] = {nullptr, nullptr};
    bool gated_taken = false;
    void offer_gated(void* g0, void* g1) { gated[0] = (float*)g0; gated[1] = (float*)g1; gated_taken = false; }
    bool took_gated() { const bool t = gated_taken; gated[0] = gated[1] = nullptr; gated_taken = false; return t; }
    // a fused gate launch may have written psi [M] to the head of a gated buffer instead of x * psi (GatePsi::psi_only): the convolution
    // that then gets that buffer as its source 0 reads x itself and multiplies while it stages (ConvOperandsSplit::psi0)
    const float* psi_key[2] = {nullptr, nullptr};
    const float* psi_x[2] = {nullptr, nullptr};
    // same for the max-pool after an encoder block: offered before the block's second conv, taken by a launch without split-K
    float* pool = nullptr;
    bool pool_taken = false;
    void offer_pool(void* p) { pool = (float*)p; pool_taken = false; }
    bool took_pool() { const bool t = pool_taken; pool = nullptr; pool_taken = false; return t; }
    // and for the sigmoid head after decoder 2's last convolution
    ConvHead head = {nullptr, nullptr, nullptr, nullptr};
    bool head_taken = false;
    void offer_head(const float* w, const float* sc, const float* sh, float* out) { head = ConvHead{w, sc, sh, out}; head_taken = false; }
    bool took_head() { const bool t = head_taken; head = ConvHead{nullptr, nullptr, nullptr, nullptr}; head_taken = false; return t; }
    int ensure(const float* p, long long count, const unsigned** out) {
        unsigned* v = find(p);
        int rc = 0;
        if (!v) {
            v = fresh(p);
            if (!v) return NBP_E_WS;
            rc = nbp_amax_launch(p, count, v, st);
        }
        *out = v;
        return rc;
    }
};
struct PathSplit : PathF32 {
    typedef AmaxBook Ctx;
    static constexpr int MODE = 2;
    static ConvPlan plan(long long M, int N, int chunks, int groups, int H = 0, int ksize = 0, int ups = 0) {
        if (ksize == 1) return ConvPlan{NBP_TILE_SPLIT_GATE, 1, chunks};      // the gates: gate1x1_h2_kernel, no split-K (conv below)
        const ConvPlan p = nbp_plan_conv_split(M, N, chunks, 0, groups, H, H, ksize, ups);
        return p.tile ? p : PathF32::plan(M, N, chunks, groups, H, ksize);
    }
    static int conv(Ctx& ctx, const nbp_weights* h, const int* li, const Ops& o, const Ops* o2, int C0, int C1, int ups, int B,
                    int H, int ks, int N, void* ws, size_t wsb, hipStream_t st) {
        if (ks == 1 && C1 == C0 && C0 % 64 == 0 && !ups && h->w1[li[0]] && (!o2 || h->w1[li[1]])) {
            // attention gate: relu([g | x] W + b) as one 1x1 GEMM on the split scheme
            const long long M = (long long)B * H * H;
            ConvOperandsSplit s[2];
            for (int g = 0; g < (o2 ? 2 : 1); ++g) {
                const Ops& q = g ? *o2 : o;
                s[g] = ConvOperandsSplit{q.src0, q.src1, h->w1[li[g]], q.scale, q.shift, q.out, nullptr, nullptr, h->wamax1[li[g]], nullptr,
                                         nullptr, nullptr};
                int rc = ctx.ensure(q.src0, M * C0, &s[g].amax0);
                if (!rc) rc = ctx.ensure(q.src1, M * C1, &s[g].amax1);
                if (rc) return rc;
            }
            GatePsi psi;
            for (int g = 0; g < 2; ++g) {
                const int l = li[(g && o2) ? 1 : 0] + 2;                      // the gate's psi layer follows W_g, W_x
                psi.wpsi[g] = (const float*)h->w[l]; psi.st[g] = h->scale[l]; psi.gated[g] = ctx.gated[(g && o2) ? 1 : 0];
            }
            // max |x psi| is measured by the fused epilogue (psi << 1 makes the inherited bound max |x| loose, and a loose bound
            // costs the small elements of the consumer's other source their low bits); the slots are bound only if it ran
            unsigned* gslot[2] = {nullptr, nullptr};
            if (ctx.gated[0]) for (int g = 0; g < (o2 ? 2 : 1); ++g) psi.gated_amax[g] = gslot[g] = ctx.unbound();
            // x * psi formed by the consumer (Up_conv{L}.conv.0: cat(x psi, d) through the plain split kernel) instead of written here
            // and read back there: the gate moves g + x instead of g + x + x psi.  Needs the measured max |x psi| (the consumer's
            // operand scale) and a consumer the split kernel takes.  Measured at B = 24 (profiles/r06/psi_on_load_ab.txt): the gates of
            // levels 4 / 3 go 157 -> 133 and 148 -> 122 us and their consumers pay 7-12 us for the multiply in their staging pass; the
            // level-2 gate (64 channels, 12288 workgroups of 4 waves) is not bound by its stores and gains nothing while its consumer pays
            // 16 us -- hence >= 128 channels.  Bit-identical either way.  NBP_SPLIT_PSI_ON_LOAD: 0 = the gated tensor is always
            // written (rounds 3-5), 1 = levels with >= 128 channels (default), 2 = every fused gate.
            static const int psi_on_load = nbp_tune_int("NBP_SPLIT_PSI_ON_LOAD", 1);
            const ConvPlan pc = nbp_plan_conv_split(M, C0, (C0 + C0) / 32 * 9, 0, o2 ? 2 : 1, H, H, 3, 0);
            psi.psi_only = psi_on_load && (C0 >= 128 || psi_on_load >= 2) && ctx.gated[0] && gslot[0] && (!o2 || gslot[1]) && pc.tile != 0 &&
                           M * C0 * 4 < (1ll << 31);
            int fused = 0;
            const int rc = nbp_gate1x1_split_launch_g(s[0], o2 ? &s[1] : nullptr, C0, M, N, 1, st, ctx.gated[0] ? &psi : nullptr, &fused);
            if (fused) {
                ctx.gated_taken = true;
                for (int g = 0; g < (o2 ? 2 : 1); ++g) {
                    if (gslot[g]) ctx.bind(ctx.gated[g], gslot[g]);
                    else ctx.alias(ctx.gated[g], (g ? *o2 : o).src1);      // |x psi| <= |x|
                    if (psi.psi_only) { ctx.psi_key[g] = ctx.gated[g]; ctx.psi_x[g] = (g ? *o2 : o).src1; }
                }
            }
            return rc;
        }
        const ConvPlan p = nbp_plan_conv_split((long long)B * H * H, N, (C0 + C1) / 32 * ks * ks, 0, o2 ? 2 : 1, H, H, ks, ups);
        if (!p.tile) return PathF32::conv(ctx, h, li, o, o2, C0, C1, ups, B, H, ks, N, ws, wsb, st);
        const long long hw = (long long)B * (ups ? H / 2 : H) * (ups ? H / 2 : H);
        ConvOperandsSplit s[2];
        for (int g = 0; g < (o2 ? 2 : 1); ++g) {
            const Ops& q = g ? *o2 : o;
            s[g] = ConvOperandsSplit{q.src0, q.src1, h->w3[li[g]], q.scale, q.shift, q.out, nullptr, nullptr, h->wamax[li[g]], nullptr,
                                     h->w3u[li[g]], h->wamax_u[li[g]]};
            int rc = ctx.ensure(q.src0, hw * C0, &s[g].amax0);
            if (!rc && C1) rc = ctx.ensure(q.src1, hw * C1, &s[g].amax1);
            if (rc) return rc;
            s[g].amax_out = ctx.fresh(q.out);
            for (int k = 0; k < 2; ++k)        // source 0 is a buffer that holds psi, not x * psi: read x and multiply while staging
                if (ctx.psi_key[k] && q.src0 == ctx.psi_key[k]) {
                    s[g].src0 = ctx.psi_x[k]; s[g].psi0 = ctx.psi_key[k];
                    ctx.psi_key[k] = nullptr; ctx.psi_x[k] = nullptr;
                }
        }
        float* pools[2] = {ctx.pool, nullptr};
        int pooled = 0, headed = 0;
        const int rc = nbp_conv_split_launch_g(s[0], o2 ? &s[1] : nullptr, C0, C1, ups, B, H, H, ks, N, 1, 0, ws, wsb, st,
                                               (ctx.pool && !o2) ? pools : nullptr, &pooled, (ctx.head.out && !o2) ? &ctx.head : nullptr,
                                               &headed);
        if (headed) ctx.head_taken = true;
        if (pooled) {
            ctx.pool_taken = true;
            ctx.alias(ctx.pool, o.out);                 // max-pool of a tensor: its max bounds the pooled one
        }
        return rc;
    }
    static int first(Ctx& ctx, const float* x, int B, int s, const nbp_weights* h, T* out, hipStream_t st) {
        unsigned* slot = ctx.fresh(out);
        int did = 0;
        const int rc = nbp_conv_first_amax_launch(x, B, s, s, (const float*)h->w[0], h->scale[0], h->shift[0], out, slot, &did, st);
        if (!did && slot) --ctx.n;          // no max from this kernel: forget the binding, the consumer computes it
        return rc;
    }
    static int pool(Ctx& ctx, const T* in, int B, int H, int Cn, T* out, hipStream_t st) {
        ctx.alias(out, in);
        return nbp_maxpool2_nhwc_f32(in, B, H, H, Cn, out, st);
    }
    static int gate(Ctx& ctx, const T* q, int F, const float* w, const float* s2, const T* x, int Cn, long long M, T* out,
                    hipStream_t st) {
        ctx.alias(out, x);
        return nbp_psi_gate_f32(q, F, w, s2, x, Cn, M, out, st);
    }
};
// bf16 path: the max-pool / sigmoid-head offers of the driver, taken by the 64-channel kernel's epilogue (nbp_bf16.hip)
struct Bf16Ctx : NoCtx {
    bf16_t* gated[2] = {nullptr, nullptr};
    bool gated_taken = false;
    void offer_gated(void* g0, void* g1) { gated[0] = (bf16_t*)g0; gated[1] = (bf16_t*)g1; gated_taken = false; }
    bool took_gated() { const bool t = gated_taken; gated[0] = gated[1] = nullptr; gated_taken = false; return t; }
    bf16_t* pool = nullptr;
    bool pool_taken = false;
    void offer_pool(void* p) { pool = (bf16_t*)p; pool_taken = false; }
    bool took_pool() { const bool t = pool_taken; pool = nullptr; pool_taken = false; return t; }
    ConvHead head = {nullptr, nullptr, nullptr, nullptr};
    bool head_taken = false;
    void offer_head(const float* w, const float* sc, const float* sh, float* out) { head = ConvHead{w, sc, sh, out}; head_taken = false; }
    bool took_head() { const bool t = head_taken; head = ConvHead{nullptr, nullptr, nullptr, nullptr}; head_taken = false; return t; }
};
struct PathBF16 {
    typedef bf16_t T;
    typedef ConvOperandsH Ops;
    typedef Bf16Ctx Ctx;
    static constexpr int CHUNK = 64;
    static ConvPlan plan(long long M, int N, int chunks, int groups, int H = 0, int ksize = 0, int ups = 0) {
        return nbp_plan_conv_bf16(M, N, chunks, 0, 0, groups, H, H, ksize, ups);
    }
    static constexpr int MODE = 1;
    static int conv(Ctx& ctx, const nbp_weights* h, const int* li, const Ops& o, const Ops* o2, int C0, int C1, int ups, int B, int H,
                    int ks, int N, void* ws, size_t wsb, hipStream_t st) {
        Ops a = o, b = o2 ? *o2 : o;
        a.wpk_up = (const bf16_t*)h->wup16[li[0]];
        if (o2) b.wpk_up = (const bf16_t*)h->wup16[li[1]];
        bf16_t* pools[2] = {ctx.pool, nullptr};
        int pooled = 0, headed = 0, psi_fused = 0;
        GatePsiH psi{{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};
        const bool gate = ks == 1 && C1 == C0 && ctx.gated[0] && (!o2 || ctx.gated[1]);
        if (gate)
            for (int g = 0; g < (o2 ? 2 : 1); ++g) {
                const int l = li[g] + 2;                                      // the gate's psi layer follows W_g, W_x
                psi.wpsi[g] = (const float*)h->w[l]; psi.st[g] = h->scale[l]; psi.gated[g] = ctx.gated[g];
            }
        const int rc = nbp_conv_igemm_bf16_launch_g(a, o2 ? &b : nullptr, C0, C1, ups, B, H, H, ks, N, 1, 0, 0, ws, wsb, st,
                                                    (ctx.pool && !o2) ? pools : nullptr, &pooled,
                                                    (ctx.head.out && !o2) ? &ctx.head : nullptr, &headed, gate ? &psi : nullptr, &psi_fused);
        if (pooled) ctx.pool_taken = true;
        if (headed) ctx.head_taken = true;
        if (psi_fused) ctx.gated_taken = true;
        return rc;
    }
    static int first(Ctx&, const float* x, int B, int s, const nbp_weights* h, T* out, hipStream_t st) {
        return nbp_conv_first_bf16_launch(x, B, s, s, (const float*)h->w[0], h->scale[0], h->shift[0], out, st);
    }
    static int pool(Ctx&, const T* in, int B, int H, int C, T* out, hipStream_t st) { return nbp_maxpool2_bf16_launch(in, B, H, H, C, out, st); }
    static int gate(Ctx&, const T* q, int F, const float* w, const float* s2, const T* x, int C, long long M, T* out, hipStream_t st) {
        return nbp_psi_gate_bf16_launch(q, F, w, s2, x, C, M, out, st);
    }
    static int head(const T* in, int B, int H, int C, const float* w, int no, const float* sc, const float* sh, int sig,
                    float* out, hipStream_t st) {
        return nbp_final_1x1_bf16_launch(in, B, H, H, C, w, no, sc, sh, sig, out, st);
    }
};

template <typename P>
size_t splitk_scratch_floats(long long M, int N, int cin_total, int taps, int groups, int H, int ups = 0) {
    ConvPlan p = P::plan(M, N, cin_total / P::CHUNK * taps, groups, H, taps == 9 ? 3 : 1, ups);
    return p.split_k > 1 ? (size_t)groups * p.split_k * M * N : 0;
}

// Runs (or, with h == nullptr, only sizes) the network.
template <typename P>
int run_forward(const nbp_weights* h, const float* x, int B, int S, float* out1, float* out2, Bump& bp,
                hipStream_t st, Timer* tm = nullptr) {
    typedef typename P::T T;
    typedef typename P::Ops Ops;
    char nm[48];
    const bool dry = bp.dry;
    const int enc[5] = {64, 128, 256, 512, 1024};
    int rc = 0;
    // split-K scratch: sized for the worst layer, shared by all (stream-ordered)
    size_t sk = 0;
    {
        int s = S;
        sk = 0;
        for (int e = 0; e < 5; ++e, s /= 2) {
            long long M = (long long)B * s * s;
            if (e > 0) sk = max(sk, splitk_scratch_floats<P>(M, enc[e], enc[e - 1], 9, 1, s));
            sk = max(sk, splitk_scratch_floats<P>(M, enc[e], enc[e], 9, 1, s));
            if (e < 4) {   // decoder level at this resolution: co = enc[e], ci = enc[e+1]
                for (int g = 1; g <= 2; ++g) {      // levels 5 and 4 run both decoders in one grouped launch
                    sk = max(sk, splitk_scratch_floats<P>(M, enc[e], enc[e + 1], 9, g, s));
                    sk = max(sk, splitk_scratch_floats<P>(M, enc[e], enc[e + 1], 9, g, s, 1));       // Up{L}.up.1 reads through the upsample
                    sk = max(sk, splitk_scratch_floats<P>(M, enc[e], enc[e], 9, g, s));
                    sk = max(sk, splitk_scratch_floats<P>(M, enc[e], 2 * enc[e], 9, g, s));
                    sk = max(sk, splitk_scratch_floats<P>(M, enc[e] / 2, 2 * enc[e], 1, g, s));
                }
            }
        }
    }
    float* skws = bp.take<float>(sk ? sk : 64);
    const size_t skbytes = sk * sizeof(float);
    typename P::Ctx ctx;
    rc = ctx.init(bp, st, dry);

    // one (ng = 1) or two (ng = 2: decoder 1 next to decoder 2) same-shaped convolutions per launch
    auto conv2 = [&](const char* name, int ng, const int* li, const T* const* s0, int C0, const T* const* s1,
                     int C1, int ups, int Hh, int ksize, int N, T* const* out) {
        if (dry || rc) return;
        Ops o[2];
        for (int g = 0; g < ng; ++g)
            o[g] = Ops{s0[g], s1 ? s1[g] : nullptr, (const T*)h->w[li[g]], h->scale[li[g]], h->shift[li[g]], out[g]};
        rc = P::conv(ctx, h, li, o[0], ng == 2 ? &o[1] : nullptr, C0, C1, ups, B, Hh, ksize, N, skws, skbytes, st);
        if (tm && !rc) {
            const long long M = (long long)B * Hh * Hh;
            const int K = (C0 + C1) * ksize * ksize;
            ConvPlan p = P::plan(M, N, K / P::CHUNK, ng, Hh, ksize, ups);
            tm->mark(name, 2.0 * ng * M * N * K, p.tile, p.split_k, M, N, K);
        }
    };
    auto conv = [&](const char* name, int li, const T* s0, int C0, const T* s1, int C1, int ups, int Hh,
                    int ksize, int N, T* out) {
        const T* s0a[1] = {s0};
        const T* s1a[1] = {s1};
        T* oa[1] = {out};
        conv2(name, 1, &li, s0a, C0, s1 ? s1a : nullptr, C1, ups, Hh, ksize, N, oa);
    };
    auto stamp = [&](const char* name, double flops, long long M, int N, int K) {
        if (tm && !dry && !rc) tm->mark(name, flops, -1, 1, M, N, K);
    };

    // ---- encoder
    T* skip[5];
    int li = 0;
    int s = S;
    const T* prev = nullptr;
    T* pooled_next = nullptr;
    bool pool_fused = false;
    for (int e = 0; e < 5; ++e) {
        const int co = enc[e];
        const size_t n = (size_t)B * s * s * co;
        T* a = bp.take<T>(n);
        T* b = bp.take<T>(n);
        if (e == 0) {
            if (!dry && !rc) rc = P::first(ctx, x, B, s, h, a, st);
            stamp("Conv1.conv.0(first)", 2.0 * B * s * s * 64 * 45, (long long)B * s * s, 64, 45);
        } else {
            T* pooled = pooled_next;
            if (!dry && !rc && !pool_fused) rc = P::pool(ctx, prev, B, 2 * s, enc[e - 1], pooled, st);
            snprintf(nm, sizeof nm, "Maxpool%d", e);
            stamp(nm, 0, (long long)B * s * s, enc[e - 1], 0);
            snprintf(nm, sizeof nm, "Conv%d.conv.0", e + 1);
            conv(nm, li, pooled, enc[e - 1], nullptr, 0, 0, s, 3, co, a);
        }
        snprintf(nm, sizeof nm, "Conv%d.conv.3", e + 1);
        if (e < 4) {                                  // the block's max-pool is offered to the conv's epilogue (split path)
            pooled_next = bp.take<T>((size_t)B * (s / 2) * (s / 2) * co);
            ctx.offer_pool(pooled_next);
        }
        conv(nm, li + 1, a, co, nullptr, 0, 0, s, 3, co, b);
        pool_fused = ctx.took_pool();
        li += 2;
        skip[e] = b; prev = b;
        if (e < 4) s /= 2;
    }
    // ---- decoders.  Level L (5..2) works at resolution S >> (L-2) with co = enc[L-2], ci = enc[L-1].
    // Levels 5 and 4 exist in both decoders with identical shapes: each of their layers is ONE grouped
    // launch (decoder 1 = group 0, decoder 2 = group 1), which doubles the workgroups per launch at
    // B = 1 and halves the split-K factor.  Decoder 2 then continues alone through levels 3 and 2.
    const int li_d1 = 10, li_d2 = 22;
    bool head2_fused = false;
    const T* cur[2] = {skip[4], skip[4]};
    for (int Lv = 5; Lv >= 2; --Lv) {
        const int ng = Lv >= 4 ? 2 : 1;
        const int g0 = Lv >= 4 ? 0 : 1;                 // first decoder index handled (0-based)
        const int co = enc[Lv - 2], ci = enc[Lv - 1];
        const int sr = S >> (Lv - 2);
        const long long M = (long long)B * sr * sr;
        const T* xs = skip[Lv - 2];
        T *dd[2], *q[2], *ag[2], *u[2], *o[2];
        int lis[2];
        const T *src[2], *xsa[2] = {xs, xs};
        for (int g = 0; g < ng; ++g) {
            dd[g] = bp.take<T>((size_t)M * co); q[g] = bp.take<T>((size_t)M * (co / 2)); ag[g] = bp.take<T>((size_t)M * co);
            u[g] = bp.take<T>((size_t)M * co); o[g] = bp.take<T>((size_t)M * co);
            const int d = g0 + g;                       // 0 = decoder 1, 1 = decoder 2
            lis[g] = (d == 0 ? li_d1 : li_d2) + (5 - Lv) * 6;
            src[g] = cur[d];
        }
        const char* tag = ng == 2 ? "{1,2}" : "2";
        auto shifted = [&](int k, int* dst) { for (int g = 0; g < ng; ++g) dst[g] = lis[g] + k; };
        int l[2];
        snprintf(nm, sizeof nm, "Up%d_%s.up.1", Lv, tag);
        shifted(0, l); conv2(nm, ng, l, src, ci, nullptr, 0, 1, sr, 3, co, dd);                    // upsample + conv3x3
        snprintf(nm, sizeof nm, "Att%d_%s.W_g+W_x", Lv, tag);
        ctx.offer_gated(ag[0], ng == 2 ? ag[1] : nullptr);
        shifted(1, l); conv2(nm, ng, l, (const T* const*)dd, co, xsa, co, 0, sr, 1, co / 2, q);  // relu(W_g g + W_x x)
        const bool psi_fused = ctx.took_gated();            // the split path's gate kernel may have run the tail itself
        for (int g = 0; g < ng && !dry && !rc && !psi_fused; ++g)
            rc = P::gate(ctx, q[g], co / 2, (const float*)h->w[lis[g] + 3], h->scale[lis[g] + 3], xs, co, M, ag[g], st);
        snprintf(nm, sizeof nm, "Att%d_%s.psi*x", Lv, tag);
        stamp(nm, 2.0 * ng * M * (co / 2), M, 1, co / 2);
        snprintf(nm, sizeof nm, "Up_conv%d_%s.conv.0", Lv, tag);
        shifted(4, l); conv2(nm, ng, l, (const T* const*)ag, co, (const T* const*)dd, co, 0, sr, 3, co, u);
        snprintf(nm, sizeof nm, "Up_conv%d_%s.conv.3", Lv, tag);
        if (Lv == 2 && !dry) ctx.offer_head((const float*)h->w[47], h->scale[47], h->shift[47], out2);   // Final2 consumes this layer alone
        shifted(5, l); conv2(nm, ng, l, (const T* const*)u, co, nullptr, 0, 0, sr, 3, co, o);
        if (Lv == 2) head2_fused = ctx.took_head();
        for (int g = 0; g < ng; ++g) cur[g0 + g] = o[g];
        if (Lv == 4) {
            if (!dry && !rc)
                rc = P::head(cur[0], B, S / 4, 256, (const float*)h->w[46], 8, h->scale[46], h->shift[46], 0, out1, st);
            stamp("Final1", 2.0 * B * (S / 4) * (S / 4) * 8 * 256, (long long)B * (S / 4) * (S / 4), 8, 256);
        }
    }
    if (!dry && !rc && !head2_fused)
        rc = P::head(cur[1], B, S, 64, (const float*)h->w[47], 1, h->scale[47], h->shift[47], 1, out2, st);
    stamp("Final2", 2.0 * B * S * S * 64, (long long)B * S * S, 1, 64);
    return rc;
}

}  // namespace

template <typename P>
static size_t workspace_bytes_impl(int B, int S) {
    if (B < 1 || S < 16 || S % 16) return 0;
    Bump bp{nullptr, 0, 0, true};
    run_forward<P>(nullptr, nullptr, B, S, nullptr, nullptr, bp, nullptr);
    return bp.off + 256;
}

template <typename P>
static int forward_impl(const nbp_weights* handle, const float* x, int B, int S, float* out1, float* out2, void* ws,
                        size_t ws_bytes, void* stream, nbp_layer_timing* timings_host, int max_entries,
                        int* n_entries_host) {
    NBP_RETURN_IF(!handle || !x || !out1 || !out2 || !ws, NBP_E_ARG);
    NBP_RETURN_IF(handle->bf16 != (P::MODE == 1 ? 1 : 0) || (P::MODE == 2 && !handle->split), NBP_E_ARG);   // handle packed for another path
    NBP_RETURN_IF(B < 1, NBP_E_ARG);
    NBP_RETURN_IF(S < 16 || S % 16, NBP_E_SHAPE);
    NBP_RETURN_IF(ws_bytes < workspace_bytes_impl<P>(B, S), NBP_E_WS);
    uintptr_t p = ((uintptr_t)ws + 255) / 256 * 256;
    Bump bp{(char*)p, ws_bytes, 0, false};
    if (!timings_host) return run_forward<P>(handle, x, B, S, out1, out2, bp, (hipStream_t)stream);
    NBP_RETURN_IF(!n_entries_host || max_entries < 1, NBP_E_ARG);
    static Timer tm;   // 256 events; the profiling entry points are not re-entrant
    tm.st = (hipStream_t)stream; tm.out = timings_host; tm.cap = max_entries;
    int rc = tm.begin();
    if (rc) return rc;
    rc = run_forward<P>(handle, x, B, S, out1, out2, bp, (hipStream_t)stream, &tm);
    int rc2 = tm.finish();
    *n_entries_host = tm.n;
    return rc ? rc : rc2;
}

extern "C" size_t nbp_forward_workspace_bytes(int B, int S) { return workspace_bytes_impl<PathF32>(B, S); }
extern "C" size_t nbp_forward_workspace_bytes_bf16(int B, int S) { return workspace_bytes_impl<PathBF16>(B, S); }

extern "C" int nbp_forward_f32(const nbp_weights* handle, const float* x, int B, int S, float* out1, float* out2,
                               void* ws, size_t ws_bytes, void* stream) {
    NBP_ENTER();
    return forward_impl<PathF32>(handle, x, B, S, out1, out2, ws, ws_bytes, stream, nullptr, 0, nullptr);
}

extern "C" size_t nbp_forward_workspace_bytes_split(int B, int S) { return workspace_bytes_impl<PathSplit>(B, S); }
extern "C" int nbp_forward_split_f32(const nbp_weights* handle, const float* x, int B, int S, float* out1, float* out2,
                                     void* ws, size_t ws_bytes, void* stream) {
    NBP_ENTER();
    return forward_impl<PathSplit>(handle, x, B, S, out1, out2, ws, ws_bytes, stream, nullptr, 0, nullptr);
}
extern "C" int nbp_forward_timed_split_f32(const nbp_weights* handle, const float* x, int B, int S, float* out1,
                                           float* out2, void* ws, size_t ws_bytes, void* stream,
                                           nbp_layer_timing* timings_host, int max_entries, int* n_entries_host) {
    NBP_ENTER();
    NBP_RETURN_IF(!timings_host || !n_entries_host, NBP_E_ARG);
    return forward_impl<PathSplit>(handle, x, B, S, out1, out2, ws, ws_bytes, stream, timings_host, max_entries,
                                   n_entries_host);
}

extern "C" int nbp_forward_bf16(const nbp_weights* handle, const float* x, int B, int S, float* out1, float* out2,
                                void* ws, size_t ws_bytes, void* stream) {
    NBP_ENTER();
    return forward_impl<PathBF16>(handle, x, B, S, out1, out2, ws, ws_bytes, stream, nullptr, 0, nullptr);
}

extern "C" int nbp_forward_timed_f32(const nbp_weights* handle, const float* x, int B, int S, float* out1,
                                     float* out2, void* ws, size_t ws_bytes, void* stream,
                                     nbp_layer_timing* timings_host, int max_entries, int* n_entries_host) {
    NBP_ENTER();
    NBP_RETURN_IF(!timings_host || !n_entries_host, NBP_E_ARG);
    return forward_impl<PathF32>(handle, x, B, S, out1, out2, ws, ws_bytes, stream, timings_host, max_entries,
                                 n_entries_host);
}

extern "C" int nbp_forward_timed_bf16(const nbp_weights* handle, const float* x, int B, int S, float* out1,
                                      float* out2, void* ws, size_t ws_bytes, void* stream,
                                      nbp_layer_timing* timings_host, int max_entries, int* n_entries_host) {
    NBP_ENTER();
    NBP_RETURN_IF(!timings_host || !n_entries_host, NBP_E_ARG);
    return forward_impl<PathBF16>(handle, x, B, S, out1, out2, ws, ws_bytes, stream, timings_host, max_entries,
                                  n_entries_host);
}

extern "C" double nbp_forward_flops(int B, int S) {
    const Table& T = table();
    // spatial size of each conv's output, canonical order
    double macs = 0;
    auto add = [&](int li, int res) { const LayerSpec& s = T.L[li]; macs += (double)res * res * s.cout * s.cin * s.k * s.k; };
    int li = 0, s = S;
    for (int e = 0; e < 5; ++e, s /= 2) { add(li++, s); add(li++, s); }
    for (int d = 1; d <= 2; ++d)
        for (int Lv = 5; Lv >= (d == 1 ? 4 : 2); --Lv) {
            int sr = S >> (Lv - 2);
            for (int k = 0; k < 6; ++k) add(li++, sr);
        }
    add(46, S / 4); add(47, S);
    return 2.0 * macs * B;
}
