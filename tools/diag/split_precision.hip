// Precision of an fp32 GEMM evaluated on the bf16 matrix pipe by exact 3-way splitting (x = hi + mid + lo, 8 bits each),
// against the fp32 MFMA chain, both measured against fp64.  One wave, C[32x32] = A[32xK] B[Kx32].
//   hipcc --offload-arch=gfx950 -O2 tools/diag/split_precision.hip -o /tmp/split_precision && /tmp/split_precision
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

__device__ inline void split3(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
    const unsigned xb = __float_as_uint(x);
    const float hi = __uint_as_float(xb & 0xFFFF0000u);
    const float r1 = x - hi;
    const unsigned r1b = __float_as_uint(r1);
    const float mid = __uint_as_float(r1b & 0xFFFF0000u);
    const float r2 = r1 - mid;
    h = xb >> 16; m = r1b >> 16; l = __float_as_uint(r2) >> 16;
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// x scaled by 2^sa = hi + lo, two fp16 pieces by round-to-nearest (x - hi is exact; lo carries 11 of the remaining 13 bits)
__device__ inline void split2h(float x, _Float16& h, _Float16& l) {
    h = (_Float16)x;
    l = (_Float16)(x - (float)h);
}

// mode 0: fp32 MFMA; 1: 6 terms one accumulator (small first); 2: 6 terms, two accumulators; 3: 9 terms; 4: 3 terms;
// 5: two fp16 pieces, 3 MFMAs (lo hi, hi lo, hi hi), operands pre-scaled by powers of two; 6: the same with two accumulators
__global__ void gemm_kernel(const float* A, const float* B, int K, int mode, float* C, float scaleA, float scaleB) {
    const int lane = threadIdx.x, row = lane & 31, kh = lane >> 5;
    f32x16 acc, acc2;
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
    if (mode == 0) {
        for (int k = 0; k < K; k += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[row * K + k + kh], B[(k + kh) * 32 + row], acc, 0, 0, 0);
    } else if (mode >= 5) {
        // per-tensor power-of-two scales: max |A| -> [2^12, 2^13), max |B| -> [2^12, 2^13)
        const float sa = scaleA, sb = scaleB;
        for (int k = 0; k < K; k += 16) {
            f16x8 ah, al, bh, bl;
            for (int e = 0; e < 8; ++e) {
                _Float16 h, l;
                split2h(A[row * K + k + 8 * kh + e] * sa, h, l); ah[e] = h; al[e] = l;
                split2h(B[(k + 8 * kh + e) * 32 + row] * sb, h, l); bh[e] = h; bl[e] = l;
            }
            if (mode == 5) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
            } else {
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc2, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc2, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
            }
        }
        const float inv = 1.f / (sa * sb);
        for (int r = 0; r < 16; ++r) acc[r] = (acc[r] + acc2[r]) * inv;
    } else {
        for (int k = 0; k < K; k += 16) {
            u16x8 a[3], b[3];
            for (int e = 0; e < 8; ++e) {
                unsigned short h, m, l;
                split3(A[row * K + k + 8 * kh + e], h, m, l); a[0][e] = h; a[1][e] = m; a[2][e] = l;
                split3(B[(k + 8 * kh + e) * 32 + row], h, m, l); b[0][e] = h; b[1][e] = m; b[2][e] = l;
            }
            auto mm = [&](int p, int q, f32x16& c) {
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[p]), __builtin_bit_cast(bf16x8, b[q]), c, 0, 0, 0);
            };
            if (mode == 1) { mm(2, 0, acc); mm(0, 2, acc); mm(1, 1, acc); mm(1, 0, acc); mm(0, 1, acc); mm(0, 0, acc); }
            if (mode == 2) { mm(2, 0, acc2); mm(0, 2, acc2); mm(1, 1, acc2); mm(1, 0, acc2); mm(0, 1, acc2); mm(0, 0, acc); }
            if (mode == 3) { mm(2, 2, acc); mm(2, 1, acc); mm(1, 2, acc); mm(2, 0, acc); mm(0, 2, acc); mm(1, 1, acc); mm(1, 0, acc); mm(0, 1, acc); mm(0, 0, acc); }
            if (mode == 4) { mm(1, 0, acc); mm(0, 1, acc); mm(0, 0, acc); }
        }
        for (int r = 0; r < 16; ++r) acc[r] += acc2[r];
    }
    // C/D layout of 32x32: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * kh) * 32 + row] = acc[r];
}

int main() {
    for (int K : {576, 1152, 9216}) {
        for (int dist = 0; dist < 4; ++dist) {
            std::vector<float> A(32 * K), B(K * 32);
            srand(7 + dist);
            auto rnd = [&]() { double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0);
                               return std::sqrt(-2 * std::log(u)) * std::cos(6.283185307179586 * v); };
            auto uni = [&]() { return 2.0 * rand() / RAND_MAX - 1.0; };
            for (auto& x : A) x = (float)(dist == 3 ? uni() : dist == 1 ? std::fabs(rnd()) : dist == 2 ? rnd() * std::pow(10.0, -6.0 * rand() / RAND_MAX) : rnd());     // 1: non-negative (after ReLU); 2: magnitudes over 6 decades
            for (auto& x : B) x = (float)(dist == 3 ? uni() * 0.1 : rnd() * 0.05);
            std::vector<double> ref(1024, 0.0), mag(1024, 0.0);
            for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
                double s = 0, m = 0;
                for (int k = 0; k < K; ++k) { s += (double)A[i * K + k] * B[k * 32 + j]; m += std::fabs((double)A[i * K + k] * B[k * 32 + j]); }
                ref[i * 32 + j] = s; mag[i * 32 + j] = m;
            }
            float *dA, *dB, *dC;
            hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 4096);
            hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
            float ma = 0, mb = 0;
            for (auto x : A) ma = std::fmax(ma, std::fabs(x));
            for (auto x : B) mb = std::fmax(mb, std::fabs(x));
            const float sA = std::exp2(12 - std::floor(std::log2(ma))), sB = std::exp2(12 - std::floor(std::log2(mb)));
            for (int mode = 0; mode < 7; ++mode) {
                gemm_kernel<<<1, 64>>>(dA, dB, K, mode, dC, sA, sB);
                std::vector<float> C(1024);
                hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
                double mx = 0, rms = 0, bias = 0;
                for (int i = 0; i < 1024; ++i) { const double e = (C[i] - ref[i]) / mag[i]; mx = std::fmax(mx, std::fabs(e)); rms += e * e; bias += e; }
                printf("K=%d dist=%d mode=%d  max|err|/sum|terms| = %.3e  rms = %.3e  mean = %+.3e\n", K, dist, mode, mx, std::sqrt(rms / 1024), bias / 1024);
            }
            hipFree(dA); hipFree(dB); hipFree(dC);
        }
    }
    return 0;
}
