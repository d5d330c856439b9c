// Phase timing of the wave-per-query feature k-NN (workgroup (3, 1)).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DKNN_STAMPS -Idis-pu_amd/csrc tools/micro/knn_lab.hip -o tools/micro/knn_lab
#include "../../dis-pu_amd/csrc/knn_wave.hip"
#include <cstdio>
#include <vector>
namespace dispu { int knn_xyz_wave_dispatch(int, int, int, int, const float*, const float*, int*, float*, int, hipStream_t); int knn_feat_wave_dispatch(int, int, int, int, int, int, int, const float*, const float*, float*, int*, hipStream_t); }
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
    for (int variant = 0; variant < 3; ++variant) {   // xyz k-NN: 32 clouds x 1024 points, k = 16
        // 0: uniform random; 1 / 2: 256 parents with 4 children each within 0.01 (the shape of the generator's coarse
        // clouds), stored parent-major (p * 4 + c) / child-major (c * 256 + p)
        const int b2 = 32, n2 = 1024, k2 = 16;
        std::vector<float> hx((size_t)b2 * n2 * 3);
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; };
        if (variant == 0) for (auto& v : hx) v = rnd();
        else
            for (int c = 0; c < b2; ++c)
                for (int pp = 0; pp < 256; ++pp) {
                    const float px = rnd(), py = rnd(), pz = rnd();
                    for (int ch = 0; ch < 4; ++ch) {
                        const int i = variant == 1 ? pp * 4 + ch : ch * 256 + pp;
                        float* o = &hx[((size_t)c * n2 + i) * 3];
                        o[0] = px + 0.02f * rnd(); o[1] = py + 0.02f * rnd(); o[2] = pz + 0.02f * rnd();
                    }
                }
        float* X; int* idx2;
        hipMalloc(&X, hx.size() * 4); hipMalloc(&idx2, (size_t)b2 * n2 * k2 * 4 + 4096);
        hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
        for (int rep = 0; rep < 3; ++rep) dispu::knn_xyz_wave_dispatch(b2, n2, n2, k2, X, X, idx2, nullptr, 0, nullptr);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int rep = 0; rep < 10; ++rep) dispu::knn_xyz_wave_dispatch(b2, n2, n2, k2, X, X, idx2, nullptr, 0, nullptr);
        hipEventRecord(e1); hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(st, idx2 + (size_t)b2 * n2 * k2, sizeof(st), hipMemcpyDeviceToHost);
        printf("knn_xyz variant %d (32 x 1024, k 16): %.1f us per call\n", variant, ms * 100);
        for (int w = 0; w < 4; ++w) {
            const unsigned long long nq = st[w * 5 + 4] & 0xFFFFFFFFull, nf = st[w * 5 + 4] >> 32;
            printf("  wave %d: %llu queries (%llu fallbacks), per query: distances+keys %.0f  threshold %.0f  compaction %.0f  everything after keys %.0f\n", w, nq, nf,
                   st[w * 5] / (double)nq, st[w * 5 + 1] / (double)nq, st[w * 5 + 2] / (double)nq, st[w * 5 + 3] / (double)nq);
        }
    }
    return 0;
}
