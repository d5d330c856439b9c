// Phase timing of the dense_conv edge kernel (wave 0..3 of workgroup 3).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DEDGE_STAMPS -Idis-pu_amd/csrc tools/micro/edge_lab.hip -o tools/micro/edge_lab
#include "../../dis-pu_amd/csrc/edge.hip"
#include <cstdio>
#include <vector>
int main() {
    const int np = 8192, n = 256, C = 48, ldy = 480;
    std::vector<int> hidx(np * 17);
    unsigned s = 7;
    for (auto& v : hidx) { s = s * 1664525u + 1013904223u; v = (s >> 8) % n; }
    std::vector<float> hf(np * C), hw(20000);
    for (auto& v : hf) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; }
    for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; }
    int* idx; float *F, *W, *Y;
    hipMalloc(&idx, hidx.size() * 4); hipMalloc(&F, hf.size() * 4); hipMalloc(&W, hw.size() * 4); hipMalloc(&Y, (size_t)np * ldy * 4 + 4096);
    hipMemcpy(idx, hidx.data(), hidx.size() * 4, hipMemcpyHostToDevice); hipMemcpy(F, hf.data(), hf.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    const float *W0 = W, *b0 = W + 96 * 24, *W1 = b0 + 24, *b1 = W1 + 72 * 24, *W2 = b1 + 24, *b2 = W2 + 96 * 24;
    for (int rep = 0; rep < 3; ++rep) dispu_edge_dense_conv(np, n, C, F, C, idx, 17, 1, W0, b0, W1, b1, W2, b2, Y, ldy, nullptr);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int rep = 0; rep < 10; ++rep) dispu_edge_dense_conv(np, n, C, F, C, idx, 17, 1, W0, b0, W1, b1, W2, b2, Y, ldy, nullptr);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long st[24];
    hipMemcpy(st, Y + (size_t)np * ldy, sizeof(st), hipMemcpyDeviceToHost);
    printf("edge_dense_conv<48> %.1f us per call\n", ms * 100);
    for (int w = 0; w < 4; ++w) {
        const unsigned long long pro = st[w * 6 + 5] >> 16;
        st[w * 6 + 5] &= 0xFFFF;
        const double n_ = (double)st[w * 6 + 5];
        printf("  wave %d, %llu groups, cycles per group: convert+prefetch %.0f  layer0 (48 MFMA) %.0f  layer1 (36) %.0f  layer2 (48) %.0f  max+store %.0f;  prologue (weight fragments + cloud staging) %llu cycles\n", w,
               st[w * 6 + 5], st[w * 6] / n_, st[w * 6 + 1] / n_, st[w * 6 + 2] / n_, st[w * 6 + 3] / n_, st[w * 6 + 4] / n_, pro);
    }
    return 0;
}
