// Probe (GPU box): what a launch that only WRITES Conv1.conv.0's output takes -- 24 maps x 256 x 256 pixels x 64 channels of fp32 =
// 403 MB in 16-byte stores, whole 256-byte pixel rows per wave as the convolution's epilogue writes them -- against the 118-120 us the
// convolution takes ("Conv1.conv.0 onto the split scheme, or prove the 403 MB store floor").
//   hipcc --offload-arch=gfx950 -O2 tools/probes/store_floor_probe.hip -o /tmp/store_floor && /tmp/store_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void store_kernel(f32x4* __restrict__ out, long long n4, float v) {
    const f32x4 x = {v, v + 1.f, v + 2.f, v + 3.f};
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) out[i] = x;
}
// the convolution's shape: a workgroup owns tiles of 8 x 32 pixels and writes them pixel row by pixel row (64 channels = 256 bytes)
__global__ __launch_bounds__(256) void store_tiles_kernel(f32x4* __restrict__ out, int B, int H, int W, float v) {
    const int tiles_x = W >> 5, tiles_y = H >> 3, n_tiles = B * tiles_y * tiles_x;
    const f32x4 x = {v, v + 1.f, v + 2.f, v + 3.f};
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
        for (int i = threadIdx.x; i < 256 * 16; i += 256) {          // 256 pixels x 16 float4
            const int p = i >> 4, q = i & 15;
            const long long pix = ((long long)b * H + ty * 8 + (p >> 5)) * W + tx * 32 + (p & 31);
            out[pix * 16 + q] = x;
        }
    }
}

int main() {
    const int B = 24, H = 256, W = 256, C = 64;
    const long long n = (long long)B * H * W * C;
    float* d; hipMalloc(&d, n * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](auto&& f, const char* label) {
        for (int i = 0; i < 3; ++i) f();
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) f();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-58s %7.1f us  %6.0f GB/s\n", label, ms / 20 * 1e3, n * 4.0 / (ms / 20 * 1e-3) / 1e9);
    };
    for (int grid : {1024, 2048, 4096, 8192})
        time([&] { store_kernel<<<grid, 256>>>((f32x4*)d, n / 4, 1.f); }, (std::string("linear 16-byte stores, grid ") + std::to_string(grid)).c_str());
    for (int grid : {1024, 2048, 6144})
        time([&] { store_tiles_kernel<<<grid, 256>>>((f32x4*)d, B, H, W, 1.f); }, (std::string("8 x 32-pixel tiles, grid ") + std::to_string(grid)).c_str());
    time([&] { hipMemsetAsync(d, 0, n * 4, 0); }, "hipMemsetAsync");
    return 0;
}
