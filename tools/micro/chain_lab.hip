// Times dispu_mlp_chain (fine head shape) and prints in-kernel cycle stamps (build with -DMC_CLOCK).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -w -DMC_CLOCK -Idis-pu_amd/csrc tools/micro/chain_lab.hip -o tools/micro/chain_lab
#include "../../dis-pu_amd/csrc/mlp_chain.hip"
#include <cstdio>
#include <vector>
int main() {
    const long rows = 32768;
    for (int N1 : {256, 128}) {
        const int K0 = 256, N2 = 256, N3 = 64;
        float *X, *W1, *W2, *W3, *W4, *b, *Y1, *R, *out;
        hipMalloc(&X, rows * K0 * 4); hipMalloc(&W1, K0 * N1 * 4); hipMalloc(&W2, N1 * N2 * 4); hipMalloc(&W3, N2 * N3 * 4); hipMalloc(&W4, 64 * 3 * 4);
        hipMalloc(&b, 1024 * 4); hipMalloc(&Y1, rows * N1 * 4); hipMalloc(&R, rows * 3 * 4); hipMalloc(&out, rows * 3 * 4);
        std::vector<float> h(rows * K0);
        unsigned s = 1;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; }
        hipMemcpy(X, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(W1, h.data(), K0 * N1 * 4, hipMemcpyHostToDevice); hipMemcpy(W2, h.data() + 70000, N1 * N2 * 4, hipMemcpyHostToDevice);
        hipMemcpy(W3, h.data() + 140000, N2 * N3 * 4, hipMemcpyHostToDevice); hipMemcpy(W4, h.data(), 192 * 4, hipMemcpyHostToDevice);
        hipMemcpy(b, h.data(), 4096, hipMemcpyHostToDevice); hipMemcpy(R, h.data(), rows * 3 * 4, hipMemcpyHostToDevice);
        auto call = [&]() { return dispu_mlp_chain(rows, K0, N1, N2, N3, X, K0, W1, b, W2, b + 256, W3, b + 512, W4, b + 768, Y1, N1, 1, R, 3, out, 3, nullptr); };
        for (int i = 0; i < 3; ++i) if (call()) { printf("launch failed\n"); return 1; }
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) call();
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
        const double fl = 2.0 * rows * ((double)K0 * N1 + (double)N1 * N2 + (double)N2 * N3);
        printf("mlp_chain 256 -> %d -> 256 -> 64 -> 3: %.1f us  %.1f TFLOP/s\n", N1, ms * 1e3, fl / ms / 1e9);
#ifdef MC_CLOCK
        unsigned long long t[4];
        hipMemcpyFromSymbol(t, HIP_SYMBOL(dispu::mc_clock_ticks), sizeof(t));
        printf("   MFMA wave 0 of one workgroup: %llu ticks for the three layers = %llu slabs of 32 MFMAs (2048 pipe cycles): %.0f ticks per slab, %.0f of them in the slab barrier\n",
               t[0], t[1], (double)t[0] / t[1], (double)t[2] / t[1]);
#endif
    }
    return 0;
}
