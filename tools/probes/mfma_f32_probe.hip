// Probe: how much of the fp32 MFMA peak survives each ingredient of the conv kernel's main loop.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f32_probe tools/probes/mfma_f32_probe.hip && ./mfma_f32_probe
// Variants: 0 = MFMA only, 1 = + LDS fragment reads, 2 = + one barrier per chunk, 3 = + global loads / LDS writes.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int V>
__global__ __launch_bounds__(256, 2) void probe(const float* __restrict__ g, float* __restrict__ out, int chunks) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * 256 * 32; i += 256) lds[i] = g[i & 8191];
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int wm = wave >> 1, wn = wave & 1;
    int fa[2], fb[2], sa[2], sb[2];
    for (int i = 0; i < 2; ++i) { int r = (wm * 2 + i) * 32 + (lane & 31); fa[i] = r * 32; sa[i] = (r >> 1) & 7; }
    for (int j = 0; j < 2; ++j) { int r = 128 + (wn * 2 + j) * 32 + (lane & 31); fb[j] = r * 32; sb[j] = (r >> 1) & 7; }
    const int khalf = lane >> 5;
    f32x4 af[2] = {{1.f, 2.f, 3.f, 4.f}, {1.f, 2.f, 3.f, 4.f}}, bf[2] = {{1.f, 1.f, 1.f, 1.f}, {2.f, 2.f, 2.f, 2.f}};
    f32x4 ga[4], gb[4];
    const float* gp = g + (size_t)blockIdx.x * 8192 + tid * 4;
    int cur = 0;
    for (int c = 0; c < chunks; ++c) {
        if (V >= 3) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { ga[i] = *(const f32x4*)(gp + i * 1024); gb[i] = *(const f32x4*)(gp + 4096 + i * 1024); }
        }
        const float* A = lds + cur * 8192;
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4) {
            if (V >= 1) {
                const int s = 2 * j4 + khalf;
#pragma unroll
                for (int i = 0; i < 2; ++i) af[i] = *(const f32x4*)(A + fa[i] + ((s ^ sa[i]) << 2));
#pragma unroll
                for (int j = 0; j < 2; ++j) bf[j] = *(const f32x4*)(A + fb[j] + ((s ^ sb[j]) << 2));
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][t], bf[j][t], acc[i][j], 0, 0, 0);
        }
        if (V >= 3) {
            float* W = lds + (cur ^ 1) * 8192;
#pragma unroll
            for (int i = 0; i < 4; ++i) { *(f32x4*)(W + (tid + i * 256) * 4) = ga[i]; *(f32x4*)(W + 4096 + (tid + i * 256) * 4) = gb[i]; }
        }
        if (V >= 2) { __syncthreads(); cur ^= 1; }
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int V>
static void run(const float* g, float* out, int blocks, int chunks) {
    hipFuncSetAttribute((const void*)&probe<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<V><<<blocks, 256, 65536>>>(g, out, chunks);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) probe<V><<<blocks, 256, 65536>>>(g, out, chunks);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double flops = (double)blocks * chunks * 4 /*waves*/ * 64 /*mfma*/ * (2.0 * 32 * 32 * 2);
    printf("variant %d blocks %d chunks %d: %.3f ms  %.1f TF (%.1f %% of 157.3)\n", V, blocks, chunks, ms, flops / ms / 1e9,
           flops / ms / 1e9 / 157.3 * 100);
}

int main() {
    float *g, *out;
    hipMalloc(&g, (size_t)1024 * 8192 * 4 + 65536); hipMalloc(&out, 1024 * 256 * 4);
    hipMemset(g, 0, (size_t)1024 * 8192 * 4 + 65536);
    for (int blocks : {512, 1024}) {
        run<0>(g, out, blocks, 288); run<1>(g, out, blocks, 288); run<2>(g, out, blocks, 288); run<3>(g, out, blocks, 288);
    }
    // sustained load (about 0.4 s per variant): does the clock hold?
    for (int rep = 0; rep < 3; ++rep) { run<0>(g, out, 1024, 11520); run<3>(g, out, 1024, 11520); }
    return 0;
}
