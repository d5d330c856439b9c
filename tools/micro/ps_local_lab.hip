// Phase timing of the wave-specialised PointShuffle2 local cell (MFMA waves of workgroup 5).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DPL_STAMPS -Idis-pu_amd/csrc tools/micro/ps_local_lab.hip -o tools/micro/ps_local_lab
#include "../../dis-pu_amd/csrc/ps_local.hip"
#include <cstdio>
#include <vector>
int main() {
    const long np = 32768; const int n = 1024;
    std::vector<int> hidx(np * 16);
    unsigned s = 7;
    for (auto& v : hidx) { s = s * 1664525u + 1013904223u; v = (s >> 8) % n; }
    std::vector<float> hx(np * 3), hg(np * 128), hw(128 * 128 + 128 + 48 + 16 + 32);
    for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xFFFF) / 65536.0f; }
    for (auto& v : hg) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; }
    for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; }
    int* idx; float *xyz, *G, *A, *W, *out;
    hipMalloc(&idx, hidx.size() * 4); hipMalloc(&xyz, hx.size() * 4); hipMalloc(&G, hg.size() * 4); hipMalloc(&A, hg.size() * 4);
    hipMalloc(&W, hw.size() * 4); hipMalloc(&out, (size_t)np * 2048 * 4 + 4096);
    hipMemcpy(idx, hidx.data(), hidx.size() * 4, hipMemcpyHostToDevice); hipMemcpy(xyz, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(G, hg.data(), hg.size() * 4, hipMemcpyHostToDevice); hipMemcpy(A, hg.data(), hg.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    const float* W1 = W; const float* b1 = W + 16384; const float* Ww = b1 + 128; const float* bw = Ww + 48; const float* sc = bw + 16; const float* sh = sc + 16;
    for (int rep = 0; rep < 3; ++rep) dispu_ps_local(np, n, 16, 128, idx, xyz, G, 128, A, W1, b1, Ww, bw, sc, sh, out, nullptr);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int rep = 0; rep < 10; ++rep) dispu_ps_local(np, n, 16, 128, idx, xyz, G, 128, A, W1, b1, Ww, bw, sc, sh, out, nullptr);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long st[56];
    hipMemcpy(st, out + (size_t)np * 2048, sizeof(st), hipMemcpyDeviceToHost);
    printf("ps_local %.1f us per call\n", ms * 100);
    for (int w = 0; w < 4; ++w)
        printf("  mfma wave %d over %llu groups, cycles per group: mma %.0f  slab-barriers %.0f  park+barrier %.0f  contraction+store %.0f\n", w, st[w * 5 + 4],
               st[w * 5] / (double)st[w * 5 + 4], st[w * 5 + 1] / (double)st[w * 5 + 4], st[w * 5 + 2] / (double)st[w * 5 + 4], st[w * 5 + 3] / (double)st[w * 5 + 4]);
    for (int w = 0; w < 4; ++w)
        printf("  helper wave %d over %llu groups, cycles per group: store_a (incl. waiting for its gathers) %.0f  issue loads / weight net %.0f  barriers %.0f\n", w,
               st[20 + w * 4 + 3], st[20 + w * 4] / (double)st[20 + w * 4 + 3], st[20 + w * 4 + 1] / (double)st[20 + w * 4 + 3],
               st[20 + w * 4 + 2] / (double)st[20 + w * 4 + 3]);
    for (int w = 0; w < 4; ++w)
        printf("  mfma wave %d: wait at the barrier after slab 0 / 1 / 2 / 3: %.0f %.0f %.0f %.0f cycles per group\n", w, st[40 + w * 4] / (double)st[w * 5 + 4],
               st[41 + w * 4] / (double)st[w * 5 + 4], st[42 + w * 4] / (double)st[w * 5 + 4], st[43 + w * 4] / (double)st[w * 5 + 4]);
    return 0;
}
