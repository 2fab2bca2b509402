// Probe (GPU box): LDS rate of ds_read_b64_tr_b16 with the weight-gradient kernel's plane layout (64-B pixel rows, lane address =
// pixel row 4 (l >> 4) + ((l & 15) >> 2), channels 4 (l & 3)) against a swizzled layout and against ds_read_b128.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/tr16_rate_probe.hip -o /tmp/tr16r && /tmp/tr16r
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_v4;
constexpr int ITER = 2048, UNR = 8;
// mode 0: plain 64-B rows; 1: 32-B half swizzled by bit 2 of the pixel row; 2: 80-B rows; 3: 32-B half swizzled by bits 2 and 3 (xor)
template <int MODE>
__global__ void rate(long long* cycles, short* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) reinterpret_cast<int*>(lds)[i] = i;
    __syncthreads();
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int pix = 4 * (l >> 4) + ((l & 15) >> 2), q = l & 3;
    int off[UNR];
    for (int u = 0; u < UNR; ++u) {
        const int p = pix + 16 * u + 3 * w;                    // different pixel blocks per read and wave (like the tap offsets)
        const int half = w & 1;                                // the wave's 16-channel group
        int a;
        if (MODE == 0) a = p * 64 + half * 32 + q * 8;
        else if (MODE == 1) a = p * 64 + ((half ^ ((p >> 2) & 1)) * 32) + q * 8;
        else if (MODE == 2) a = p * 80 + half * 32 + q * 8;
        else a = p * 64 + ((half ^ ((p >> 2) & 1) ^ ((p >> 3) & 1)) * 32) + q * 8;
        off[u] = a;
    }
    s16x4 acc = {0, 0, 0, 0};
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(lds + off[u] + (it & 3) * 8192));
            acc += v;
        }
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0) cycles[0] = t1 - t0;
    sink[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
__global__ void rate_b128(long long* cycles, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) reinterpret_cast<int*>(lds)[i] = i;
    __syncthreads();
    f32x4 acc = {0, 0, 0, 0};
    const int base = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 1024;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) acc += *reinterpret_cast<const f32x4*>(lds + base + u * 4096 + (it & 3) * 8192 * 0);
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0) cycles[0] = t1 - t0;
    sink[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
int main() {
    long long* c; short* s; float* sf;
    hipMalloc(&c, 8); hipMalloc(&s, 1024 * 2); hipMalloc(&sf, 1024 * 4);
    auto report = [&](const char* name, double bytes_per_wave_instr, int waves) {
        long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
        const double instr = (double)ITER * UNR * waves;
        printf("%-44s %2d waves: %8lld clock64 ticks, %.3f ticks per wave-instruction, %.1f B per tick\n", name, waves, h, h / instr,
               bytes_per_wave_instr * instr / h);
    };
    for (int waves : {1, 4, 8}) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&rate<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&rate<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&rate<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&rate<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&rate_b128), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        for (int rep = 0; rep < 2; ++rep) {
            rate<0><<<1, 64 * waves, 65536>>>(c, s); hipDeviceSynchronize(); if (rep) report("tr16_b64, 64-B rows (the kernel's layout)", 512, waves);
            rate<1><<<1, 64 * waves, 65536>>>(c, s); hipDeviceSynchronize(); if (rep) report("tr16_b64, half swizzled by pixel bit 2", 512, waves);
            rate<3><<<1, 64 * waves, 65536>>>(c, s); hipDeviceSynchronize(); if (rep) report("tr16_b64, half swizzled by pixel bits 2^3", 512, waves);
            rate<2><<<1, 64 * waves, 65536>>>(c, s); hipDeviceSynchronize(); if (rep) report("tr16_b64, 80-B rows", 512, waves);
            rate_b128<<<1, 64 * waves, 65536>>>(c, sf); hipDeviceSynchronize(); if (rep) report("ds_read_b128, lane-contiguous", 1024, waves);
        }
    }
    return 0;
}
