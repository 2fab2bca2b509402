// Probe (GPU box): where do float atomics on ordinary device memory execute on gfx950, and at what rate?
//   agent scope (atomicAdd)          -> sc1: resolved at the memory side, coherent across the 8 XCDs
//   workgroup scope (no sc bits)     -> executed in the issuing XCD's L2: fast, but two XCDs adding to one address do not see each other
// If the second form is L2-local, a scatter can add into a PRIVATE copy per XCD (index = HW_REG_XCC_ID) at L2 rates and a
// second pass sums the 8 copies (counts are integers in fp32: any order gives the same bits).
// The probe scatters N adds over a 6 x 256 x 256 plane set (the map accumulation's output) three ways, times them and checks
// every word.  Build + run:  hipcc --offload-arch=gfx950 -O3 -o /tmp/xcd_atomic tools/probes/xcd_atomic_probe.hip && /tmp/xcd_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int CELLS = 6 * 256 * 256;

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
    return v;
}
__device__ __forceinline__ unsigned key_of(unsigned i, unsigned spread) {
    // spread = distinct cells the adds fall on (hot cells when small)
    return (unsigned)(((unsigned long long)(i * 2654435761u) * spread) >> 32);
}

__global__ void k_agent(float* out, unsigned n, unsigned spread) {
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) atomicAdd(out + key_of(i, spread), 1.0f);
}
__global__ void k_wg_shared(float* out, unsigned n, unsigned spread) {
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        __hip_atomic_fetch_add(out + key_of(i, spread), 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__global__ void k_wg_private(float* copies, unsigned n, unsigned spread, unsigned* seen) {
    const unsigned x = xcc_id();
    if (threadIdx.x == 0) atomicOr(seen, 1u << x);
    float* out = copies + (size_t)(x & 7u) * CELLS;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        __hip_atomic_fetch_add(out + key_of(i, spread), 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__global__ void k_sum8(const float4* copies, float4* out) {
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < CELLS / 4; i += gridDim.x * blockDim.x) {
        float4 s = copies[i];
        for (int x = 1; x < 8; ++x) {
            const float4 v = copies[(size_t)x * (CELLS / 4) + i];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        out[i] = s;
    }
}

static double total(const std::vector<float>& h) { double s = 0; for (float v : h) s += v; return s; }

int main() {
    float *out, *copies; unsigned* seen;
    hipMalloc(&out, CELLS * 4); hipMalloc(&copies, 8ull * CELLS * 4); hipMalloc(&seen, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> h(CELLS), ref(CELLS);
    const unsigned n = 4u << 20;
    for (unsigned spread : {(unsigned)CELLS, 65536u, 4096u, 256u}) {
        float ms[4] = {0, 0, 0, 0};
        // reference counts on the host
        std::fill(ref.begin(), ref.end(), 0.f);
        for (unsigned i = 0; i < n; ++i) ref[(unsigned)(((unsigned long long)(i * 2654435761u) * spread) >> 32)] += 1.f;
        auto wrong = [&]() { long w = 0; for (int i = 0; i < CELLS; ++i) w += h[i] != ref[i]; return w; };
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(out, 0, CELLS * 4); hipDeviceSynchronize();
            hipEventRecord(e0); k_agent<<<1024, 256>>>(out, n, spread); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms[0], e0, e1);
        }
        hipMemcpy(h.data(), out, CELLS * 4, hipMemcpyDeviceToHost);
        const long w0 = wrong();
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(out, 0, CELLS * 4); hipDeviceSynchronize();
            hipEventRecord(e0); k_wg_shared<<<1024, 256>>>(out, n, spread); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms[1], e0, e1);
        }
        hipMemcpy(h.data(), out, CELLS * 4, hipMemcpyDeviceToHost);
        const long w1 = wrong(); const double t1 = total(h);
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(copies, 0, 8ull * CELLS * 4); hipMemset(seen, 0, 4); hipDeviceSynchronize();
            hipEventRecord(e0); k_wg_private<<<1024, 256>>>(copies, n, spread, seen); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms[2], e0, e1);
            hipEventRecord(e0); k_sum8<<<384, 256>>>((const float4*)copies, (float4*)out); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms[3], e0, e1);
        }
        hipMemcpy(h.data(), out, CELLS * 4, hipMemcpyDeviceToHost);
        const long w2 = wrong();
        unsigned hs; hipMemcpy(&hs, seen, 4, hipMemcpyDeviceToHost);
        printf("spread %7u: agent %.1f us (%.2f G/s, wrong %ld) | wg-scope shared %.1f us (%.2f G/s, wrong %ld, total %.0f of %u) | "
               "wg-scope per-XCD copy %.1f us (%.2f G/s) + sum8 %.1f us (wrong %ld, xcc ids seen 0x%x)\n",
               spread, ms[0] * 1e3, n / ms[0] / 1e6, w0, ms[1] * 1e3, n / ms[1] / 1e6, w1, t1, n, ms[2] * 1e3, n / ms[2] / 1e6,
               ms[3] * 1e3, w2, hs);
    }
    return 0;
}
