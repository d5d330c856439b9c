// Phase timing of the fused non-local cell (MFMA waves of one workgroup): S^T product, softmax, P.V product, tile barrier.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DFA_STAMPS -Idis-pu_amd/csrc -Iinclude tools/micro/attention_lab.hip -o tools/micro/_fal
#include "../../dis-pu_amd/csrc/attention.hip"
#include <cstdio>
#include <vector>
int main() {
    const int b = 32, m = 1024;
    std::vector<float> h((size_t)b * m * 320), w(64 * 256 + 256);
    unsigned s = 3;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; }
    for (auto& v : w) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; }
    float *x, *W, *out;
    (void)hipMalloc(&x, h.size() * 4); (void)hipMalloc(&W, w.size() * 4); (void)hipMalloc(&out, (size_t)b * m * 256 * 4);
    (void)hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(W, w.data(), w.size() * 4, hipMemcpyHostToDevice);
    auto run = [&]() { return dispu_attention_project(b, m, m, 64, x + 128, 320, x, 320, x + 64, 320, 0.125f, W, W + 64 * 256, 256, out, 256, nullptr); };
    for (int i = 0; i < 3; ++i) { int rc = run(); if (rc) { printf("rc %d\n", rc); return 1; } }
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) run();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long st[16];
    (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(dispu::fa_stamps), sizeof(st));
    printf("attention_project (32, 1024, 1024): %.1f us per call\n", ms * 100);
    for (int wv = 0; wv < 4; ++wv)
        printf("  mfma wave %d, cycles per 32-key tile: S^T product %.0f  softmax %.0f  P.V product %.0f  barrier %.0f\n", wv, st[wv * 4] / 32.0,
               st[wv * 4 + 1] / 32.0, st[wv * 4 + 2] / 32.0, st[wv * 4 + 3] / 32.0);
    return 0;
}
