// Probe (GPU box): which 16-bit elements does ds_read_b64_tr_b16 deliver to each lane?
// LDS holds e[i] = i (fp16-sized integers); lane l passes the address of elements 4 l .. 4 l + 3.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/tr16_probe.hip -o /tmp/tr16 && /tmp/tr16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_v4;
__global__ void probe(short* out, int stride_elems) {
    __shared__ __attribute__((aligned(16))) short e[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) e[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x;
    // lane l -> segment (row = l >> 2 within its 16-lane group ... ) generic: address of elements base
    const int base = (l >> 4) * 4 * stride_elems + ((l & 15) >> 2) * stride_elems + (l & 3) * 4;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(e + base));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
    short* d; hipMalloc(&d, 64 * 4 * 2);
    for (int stride : {16, 80}) {
        probe<<<1, 64>>>(d, stride);
        short h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("row stride %d elements; lane l's address = row (l>>4)*4 + ((l&15)>>2), col 4*(l&3)\n", stride);
        for (int l = 0; l < 64; ++l) {
            printf("lane %2d:", l);
            for (int j = 0; j < 4; ++j) printf(" (r%d,c%d)", h[l * 4 + j] / stride, h[l * 4 + j] % stride);
            printf("\n");
        }
    }
    return 0;
}
