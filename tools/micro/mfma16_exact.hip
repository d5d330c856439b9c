// Is v_mfma_f32_16x16x4_f32 the ascending-k fmaf chain  d = fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0,c))))  bit for bit?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/micro/mfma16_exact.hip -o tools/micro/mfma16_exact
// (v_mfma_f32_32x32x2_f32 is: DESIGN.md "Pinned arithmetic".)  Prints the number of mismatching outputs per candidate order.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* A, const float* B, const float* C, float* D, int chain) {
    const int l = threadIdx.x;
    f32x4 acc;
    for (int r = 0; r < 4; ++r) acc[r] = C[(4 * (l / 16) + r) * 16 + (l % 16)];
    for (int s = 0; s < chain; ++s)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[s * 64 + (l % 16) * 4 + l / 16], B[s * 64 + (l / 16) * 16 + (l % 16)], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * (l / 16) + r) * 16 + (l % 16)] = acc[r];
}
int main() {
    const int chain = 8, trials = 200;
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFFFF) / 16777216.0f - 0.5f; };
    float *dA, *dB, *dC, *dD;
    hipMalloc(&dA, chain * 64 * 4); hipMalloc(&dB, chain * 64 * 4); hipMalloc(&dC, 256 * 4); hipMalloc(&dD, 256 * 4);
    long bad_asc = 0, bad_desc = 0, bad_pair = 0, total = 0;
    for (int t = 0; t < trials; ++t) {
        std::vector<float> A(chain * 64), B(chain * 64), C(256), D(256);
        for (auto& v : A) v = rnd() * std::ldexp(1.0f, (int)(rnd() * 16));
        for (auto& v : B) v = rnd() * std::ldexp(1.0f, (int)(rnd() * 16));
        for (auto& v : C) v = rnd();
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dC, C.data(), 1024, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, chain);
        hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                float asc = C[i * 16 + j], desc = C[i * 16 + j], pr = C[i * 16 + j];
                for (int st = 0; st < chain; ++st) {
                    const float* a = &A[st * 64 + i * 4];
                    auto b = [&](int kk) { return B[st * 64 + kk * 16 + j]; };
                    for (int kk = 0; kk < 4; ++kk) asc = std::fmaf(a[kk], b(kk), asc);
                    for (int kk = 3; kk >= 0; --kk) desc = std::fmaf(a[kk], b(kk), desc);
                    pr = pr + (std::fmaf(a[1], b(1), a[0] * b(0)) + std::fmaf(a[3], b(3), a[2] * b(2)));
                }
                const float d = D[i * 16 + j];
                bad_asc += std::memcmp(&d, &asc, 4) != 0; bad_desc += std::memcmp(&d, &desc, 4) != 0; bad_pair += std::memcmp(&d, &pr, 4) != 0;
                ++total;
            }
    }
    printf("v_mfma_f32_16x16x4_f32 vs ascending fmaf chain: %ld / %ld mismatches (descending: %ld, pairwise: %ld)\n", bad_asc, total, bad_desc, bad_pair);
    return 0;
}
