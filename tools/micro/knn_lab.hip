// Phase timing of the wave-per-query feature k-NN (workgroup (3, 1)).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DKNN_STAMPS -Idis-pu_amd/csrc tools/micro/knn_lab.hip -o tools/micro/knn_lab
#include "../../dis-pu_amd/csrc/knn_wave.hip"
#include <cstdio>
#include <vector>
namespace dispu { int knn_feat_wave_dispatch(int, int, int, int, int, int, int, const float*, const float*, float*, int*, hipStream_t); }
int main() {
    const int b = 32, n = 256, c = 48, k = 17;
    std::vector<float> hf((size_t)b * n * c);
    unsigned s = 7;
    for (auto& v : hf) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; }
    float* F; int* idx;
    hipMalloc(&F, hf.size() * 4); hipMalloc(&idx, (size_t)b * n * k * 4 + 4096);
    hipMemcpy(F, hf.data(), hf.size() * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep) dispu::knn_feat_wave_dispatch(b, n, n, c, k, c, c, F, F, nullptr, idx, nullptr);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int rep = 0; rep < 10; ++rep) dispu::knn_feat_wave_dispatch(b, n, n, c, k, c, c, F, F, nullptr, idx, nullptr);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long st[20];
    hipMemcpy(st, idx + (size_t)b * n * k, sizeof(st), hipMemcpyDeviceToHost);
    printf("knn_feat (32 x 256, C 48, k 17): %.1f us per call\n", ms * 100);
    for (int w = 0; w < 4; ++w)
        printf("  wave %d: staging+norms %llu cycles; %llu queries, per query: dots %.0f  sort %.0f  select+store %.0f\n", w, st[w * 5], st[w * 5 + 4],
               st[w * 5 + 1] / (double)st[w * 5 + 4], st[w * 5 + 2] / (double)st[w * 5 + 4], st[w * 5 + 3] / (double)st[w * 5 + 4]);
    return 0;
}
