// Times the PRODUCTION dispu_linear next to the lab's wave-specialised kernel on identical buffers.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Idis-pu_amd/csrc tools/micro/gemm_lab2.hip -o tools/micro/gemm_lab2
#include "../../dis-pu_amd/csrc/linear.hip"
#include "../../dis-pu_amd/csrc/linear_skinny.hip"
#include <cstdio>
#include <vector>

static float time_it(int M, int K, int N, const float* X, const float* W, const float* bias, float* Y, int act) {
    for (int i = 0; i < 2; ++i) dispu_linear(1, M, K, N, X, K, 0, W, N, 0, 0, bias, act, Y, N, 0, nullptr, 0, 0, nullptr, 0, 0, nullptr);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int r = 0; r < 20; ++r) dispu_linear(1, M, K, N, X, K, 0, W, N, 0, 0, bias, act, Y, N, 0, nullptr, 0, 0, nullptr, 0, 0, nullptr);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 20;
}

int main() {
    const int M = 32768, N = 256;
    for (int K : {2048, 256, 128}) {
        float *X, *W, *Y, *B;
        hipMalloc(&X, (size_t)M * K * 4); hipMalloc(&W, (size_t)K * N * 4); hipMalloc(&Y, (size_t)M * N * 4); hipMalloc(&B, N * 4);
        std::vector<float> hx((size_t)M * K), hw((size_t)K * N), hb(N, 0.1f);
        unsigned s = 12345;
        for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; }
        for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; }
        hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(B, hb.data(), N * 4, hipMemcpyHostToDevice);
        float ms = time_it(M, K, N, X, W, nullptr, Y, 0);
        printf("production dispu_linear K %5d: %8.1f us %6.1f TFLOP/s (no bias/act)\n", K, ms * 1e3, 2.0 * M * K * N / ms / 1e9);
        ms = time_it(M, K, N, X, W, B, Y, 1);
        printf("production dispu_linear K %5d: %8.1f us %6.1f TFLOP/s (bias + relu)\n", K, ms * 1e3, 2.0 * M * K * N / ms / 1e9);
#ifdef LIN_CLOCK
        {
            unsigned long long t[4];
            hipMemcpyFromSymbol(t, HIP_SYMBOL(dispu::lin_clock_ticks), sizeof(t));
            printf("   main loop of one workgroup: %llu ticks for %llu slabs = %.0f ticks per slab; if the kernel were all main loop the counter ran at %.2f GHz; MFMA wave 0 spent %.0f ticks per slab in the barrier\n",
                   t[0], t[1], (double)t[0] / t[1], t[0] / (ms * 1e6), (double)t[2] / t[1]);
        }
#endif
        hipFree(X); hipFree(W); hipFree(Y); hipFree(B);
    }
    return 0;
}
