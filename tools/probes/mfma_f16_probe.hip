// Probe: what the 16-bit matrix pipe sustains on this chip, to put the split convolution's rate in context.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_f16_probe tools/probes/mfma_f16_probe.hip && /tmp/mfma_f16_probe
// 512 workgroups x 4 waves (two per CU, as the conv kernel), 8 accumulator tiles per wave like the conv kernel.
// Variants: 0 = v_mfma_f32_32x32x16_f16 only, operands in registers (all-ones data); 1 = the same with random operand bits
// (the power drawn by the multipliers depends on the data); 2 = + 12 ds_read_b128 per 24 MFMAs (the conv kernel's fragment
// traffic); 3 = + one barrier per 72 MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int V>
__global__ __launch_bounds__(256, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void probe(const unsigned* __restrict__ g, float* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 16384; i += 256) ((unsigned*)lds)[i] = g[i];
    __syncthreads();
    f32x16 acc[8];
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    u32x4 a[4], b[2];
    for (int i = 0; i < 4; ++i) a[i] = V == 0 ? u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u} : *(const u32x4*)(g + ((tid * 4 + i) & 4095) * 4);
    for (int j = 0; j < 2; ++j) b[j] = V == 0 ? u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u} : *(const u32x4*)(g + ((tid * 2 + j + 2048) & 4095) * 4);
    const char* base = lds + lane * 16;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
            if (V >= 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = *(const u32x4*)(base + ((it + tap * 4 + i) & 31) * 1024);
#pragma unroll
                for (int j = 0; j < 2; ++j) b[j] = *(const u32x4*)(base + 32768 + ((it + tap * 2 + j) & 31) * 1024);
            }
#pragma unroll
            for (int v = 0; v < 3; ++v)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i]), __builtin_bit_cast(f16x8, b[j]),
                                                                                acc[i * 2 + j], 0, 0, 0);
        }
        if (V >= 3) __syncthreads();
    }
    float s = 0.f;
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int V>
static void run(const unsigned* g, float* out, int blocks, int iters) {
    hipFuncSetAttribute((const void*)&probe<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<V><<<blocks, 256, 65536>>>(g, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) probe<V><<<blocks, 256, 65536>>>(g, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double flops = (double)blocks * 4 * iters * 72 * (2.0 * 32 * 32 * 16);
    printf("variant %d: %.3f ms  %.0f TFLOP/s of fp16 MFMA work (%.1f %% of 2500)\n", V, ms, flops / ms / 1e9, flops / ms / 1e9 / 2500 * 100);
}

int main() {
    std::vector<unsigned> h(16384);
    srand(3);
    for (auto& x : h) {       // two random fp16 values in [0.5, 2) with random signs and mantissas
        unsigned v = 0;
        for (int k = 0; k < 2; ++k) v |= (unsigned)((rand() & 0x8000) | 0x3800 | (rand() & 0x07ff)) << (16 * k);
        x = v;
    }
    unsigned* g; float* out;
    hipMalloc(&g, 65536); hipMalloc(&out, 512 * 256 * 4);
    hipMemcpy(g, h.data(), 65536, hipMemcpyHostToDevice);
    run<0>(g, out, 512, 4000);
    run<1>(g, out, 512, 4000);
    run<2>(g, out, 512, 4000);
    run<3>(g, out, 512, 4000);
    return 0;
}
