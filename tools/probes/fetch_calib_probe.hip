// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the map accumulation's access pattern (MI355X_MICROARCH.md, HBM section: the
// counter reports 1/2 of the bytes of a 16-B-per-lane streaming read; other widths are uncalibrated).  Three kernels read the SAME
// known number of bytes once, with nothing else going on:
//   calib_points12  three consecutive dword loads per lane (p[3i], p[3i+1], p[3i+2]: 768 contiguous bytes per wave instruction
//                   group) -- map_accumulate_kernel's loads
//   calib_dword     one dword per lane (256 contiguous bytes per wave instruction)
//   calib_dwordx4   16 B per lane (the guide's calibrated case: expect bytes / FETCH_SIZE = 2 with FETCH_SIZE in 64-B units -> KB)
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib tools/probes/fetch_calib_probe.hip &&
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/fc -o fc -- /tmp/fetch_calib
// then bytes_read / (FETCH_SIZE * 1024) per kernel = the factor to apply to FETCH_SIZE for that pattern.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void calib_points12(const float* __restrict__ p, long long n, float* out) {
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        acc += p[3 * i] + p[3 * i + 1] + p[3 * i + 2];
    if (acc == 12345.678f) out[0] = acc;
}
__global__ void calib_dword(const float* __restrict__ p, long long n, float* out) {
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 12345.678f) out[0] = acc;
}
__global__ void calib_dwordx4(const float4* __restrict__ p, long long n, float* out) {
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float4 v = p[i];
        acc += (v.x + v.y) + (v.z + v.w);
    }
    if (acc == 12345.678f) out[0] = acc;
}

// the same for WRITE_SIZE (second pass: --pmc WRITE_SIZE): known bytes written once, 4 B and 16 B per lane
__global__ void calib_store_dword(float* __restrict__ p, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = 1.f;
}
__global__ void calib_store_dwordx4(float4* __restrict__ p, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        p[i] = float4{1.f, 2.f, 3.f, 4.f};
}

int main() {
    const long long n_points = 40ll * 1000 * 1000;                 // 480 MB: past the 256 MB Infinity Cache
    const long long bytes = n_points * 12;
    float *buf, *out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 256) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 0, bytes);
    hipDeviceSynchronize();
    calib_points12<<<4096, 256>>>(buf, n_points, out);
    hipDeviceSynchronize();
    calib_dword<<<4096, 256>>>(buf, bytes / 4, out);
    hipDeviceSynchronize();
    calib_dwordx4<<<4096, 256>>>((const float4*)buf, bytes / 16, out);
    hipDeviceSynchronize();
    calib_store_dword<<<4096, 256>>>(buf, bytes / 4);
    hipDeviceSynchronize();
    calib_store_dwordx4<<<4096, 256>>>((float4*)buf, bytes / 16);
    hipDeviceSynchronize();
    printf("bytes_per_kernel %lld\n", bytes);
    return 0;
}
