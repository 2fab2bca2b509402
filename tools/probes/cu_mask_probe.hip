// Probe (GPU box): which compute units a stream created with hipExtStreamCreateWithCUMask dispatches to, per mask word / bit --
// the bit layout of the mask against (XCC, SE, CU) as the hardware reports them, and how long a latency-bound kernel takes on
// k of the 256 CUs.   hipcc --offload-arch=gfx950 -O2 tools/probes/cu_mask_probe.hip -o /tmp/cu_mask_probe && /tmp/cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <map>
#include <vector>

__global__ void where_kernel(unsigned* out, int spin) {
    const unsigned hw = __builtin_amdgcn_s_getreg(63492);      // HW_REG_HW_ID, 32 bits
    const unsigned xcc = __builtin_amdgcn_s_getreg(6164);      // HW_REG_XCC_ID, bits 3:0
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}

int main() {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    printf("device: %s, %d CUs\n", pr.name, pr.multiProcessorCount);
    const int NB = 4096;
    unsigned* d; hipMalloc(&d, NB * 8);
    std::vector<unsigned> h(NB * 2);
    auto run = [&](hipStream_t st, const char* label) {
        hipMemsetAsync(d, 0xff, NB * 8, st);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, st);
        where_kernel<<<NB, 64, 0, st>>>(d, 2000);      // 2000 ticks of the 100 MHz wall clock = 20 us per workgroup
        hipEventRecord(e1, st);
        hipStreamSynchronize(st);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), d, NB * 8, hipMemcpyDeviceToHost);
        std::map<unsigned, std::set<unsigned>> per_xcc;
        for (int i = 0; i < NB; ++i) {
            const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 15u;
            const unsigned cu = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;
            per_xcc[xcc].insert(se * 32 + sh * 16 + cu);
        }
        int total = 0;
        printf("%-34s %7.3f ms  ", label, ms);
        for (auto& kv : per_xcc) { printf("xcc%u:%zu ", kv.first, kv.second.size()); total += (int)kv.second.size(); }
        printf(" -> %d distinct (xcc, se, sh, cu)\n", total);
    };
    hipStream_t s0; hipStreamCreate(&s0);
    run(s0, "plain stream");
    const int words = (pr.multiProcessorCount + 31) / 32;
    for (int trial = 0; trial < 6; ++trial) {
        std::vector<unsigned> mask(words, 0u);
        char label[64];
        if (trial == 0) { mask[0] = 0xffffffffu; snprintf(label, 64, "word 0 = all ones"); }
        if (trial == 1) { mask[1] = 0xffffffffu; snprintf(label, 64, "word 1 = all ones"); }
        if (trial == 2) { for (int w = 0; w < words; ++w) mask[w] = 0x0000000fu; snprintf(label, 64, "bits 0-3 of every word"); }
        if (trial == 3) { for (int w = 0; w < words; ++w) mask[w] = 0x01010101u; snprintf(label, 64, "every 8th bit"); }
        if (trial == 4) { for (int w = 0; w < words; ++w) mask[w] = 0x000000ffu; snprintf(label, 64, "bits 0-7 of every word"); }
        if (trial == 5) { for (int w = 0; w < words; ++w) mask[w] = 0xffffffffu; snprintf(label, 64, "all ones"); }
        hipStream_t sm;
        hipError_t e = hipExtStreamCreateWithCUMask(&sm, (unsigned)words, mask.data());
        if (e != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask -> %s\n", label, hipGetErrorString(e)); continue; }
        run(sm, label);
        hipStreamDestroy(sm);
    }
    return 0;
}
