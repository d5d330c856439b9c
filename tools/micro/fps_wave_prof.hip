// Where does a round of csrc/fps_wave.hip go?  Includes the kernel with cycle-counter hooks (s_memtime) and prints, per wave,
// the average cycles of each phase of a round and how many rounds the wave was active.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I include -I dis-pu_amd/csrc tools/micro/fps_wave_prof.hip -o /tmp/fwp && /tmp/fwp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
__device__ unsigned long long fpsw_prof[16][8];
#define FPSW_PROF_BEGIN unsigned long long pt_[7] = {0, 0, 0, 0, 0, 0, 0}, pa_[7] = {0, 0, 0, 0, 0, 0, 0}; unsigned long long pact_ = 0, pgrp_ = 0; bool pwas_ = false;
#define FPSW_TICK(i) { pt_[i] = __builtin_amdgcn_s_memtime(); if ((i) == 0) pwas_ = false; \
    if ((i) == 6) { pa_[0] += pt_[1] - pt_[0]; if (pwas_) { pa_[1] += pt_[2] - pt_[1]; pa_[2] += pt_[3] - pt_[2]; } pa_[3] += pt_[4] - pt_[3]; pa_[4] += pt_[5] - pt_[4]; pa_[5] += pt_[6] - pt_[5]; } }
#define FPSW_PROF_ACTIVE pwas_ = true; ++pact_;
#define FPSW_PROF_GROUPS(mask) pgrp_ += __builtin_popcount(mask);
#define FPSW_PROF_END if (lane == 0) { for (int q = 0; q < 6; ++q) fpsw_prof[wave][q] = pa_[q]; fpsw_prof[wave][6] = pact_; fpsw_prof[wave][7] = pgrp_; }
#include "fps_wave.hip"

int main(int argc, char** argv) {
    const int n = 24576, m = 8192, b = 1;
    const bool sphere = argc > 1 && argv[1][0] == 's';
    std::vector<float> h((size_t)n * 3);
    srand(1);
    for (int i = 0; i < n; ++i) {
        float v[3]; float nn = 0;
        for (int a = 0; a < 3; ++a) { v[a] = sphere ? (float)rand() / RAND_MAX * 2 - 1 : (float)rand() / RAND_MAX; nn += v[a] * v[a]; }
        if (sphere) { if (nn > 1 || nn < 1e-3f) { --i; continue; } nn = sqrtf(nn); for (int a = 0; a < 3; ++a) v[a] /= nn; }
        for (int a = 0; a < 3; ++a) h[(size_t)i * 3 + a] = v[a];
    }
    float* x; int *perm, *out;
    hipMalloc(&x, h.size() * 4); hipMalloc(&perm, n * 4); hipMalloc(&out, m * 4);
    hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(e0, 0);
        int r = dispu::fps_wave_dispatch(b, n, m, x, perm, out, 1, 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("rc %d  %.3f ms  %.0f ns/round\n", r, ms, ms * 1e6 / (m - 1));
    }
    unsigned long long p[16][8];
    hipMemcpyFromSymbol(p, HIP_SYMBOL(fpsw_prof), sizeof(p));
    printf("wave  active  lbtest  dense(act)  wavered(act)  slot->bar  barrier  final   [s_memtime ticks per round]\n");
    for (int w = 0; w < 16; ++w) {
        const double R = m - 1, A = p[w][6] ? (double)p[w][6] : 1;
        printf("%2d   %6llu (%.2f regions)  %7.1f  %7.1f  %7.1f  %7.1f  %7.1f  %7.1f\n", w, p[w][6], p[w][7] / A, p[w][0] / R, p[w][1] / A, p[w][2] / A, p[w][3] / R, p[w][4] / R, p[w][5] / R);
    }
    return 0;
}
