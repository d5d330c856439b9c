// Micro-benchmark: what does the fp32 matrix pipe of this MI355X actually sustain, and at which shader clock?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_clock.hip -o gpurun_out/mfma_clock && gpurun_out/mfma_clock
// Each wave issues ITER x NACC v_mfma_f32_32x32x2_f32 (NACC independent accumulators, dependent within one), no memory
// traffic.  Reports wall time (hipEvents), shader cycles (s_memtime, wave 0 of block 0) -> clock, and TFLOP/s for
// 1 / 2 / 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v16f __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(int iters, float a, float b, float* out, unsigned long long* cycles) {
    v16f acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cycles[0] = t1 - t0;
}

template <int NACC>
static void run(int blocks, int iters) {
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, 4);
    hipMalloc(&cyc, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(256), 0, 0, iters / 10, 1.0f, 1.0f, out, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(256), 0, 0, iters, 1.0f, 1.0f, out, cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double nmfma = (double)blocks * 4 * iters * NACC;
    const double flops = nmfma * 32 * 32 * 2 * 2;
    printf("NACC %d blocks %5d (%.1f waves/SIMD) iters %d: %.3f ms  %.1f TFLOP/s  counter %llu ticks -> %.1f ticks per MFMA per wave, counter rate %.3f GHz\n",
           NACC, blocks, blocks * 4 / 1024.0, iters, ms, flops / ms / 1e9, c, (double)c / ((double)iters * NACC), c / (ms * 1e6));
    hipFree(out);
    hipFree(cyc);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("%s  CUs %d  clockRate %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    const int iters = 200000;
    run<1>(256, iters);
    run<1>(512, iters);
    run<1>(1024, iters);
    run<2>(256, iters);
    run<4>(256, iters);
    run<4>(512, iters / 2);
    run<1>(256, iters * 5);   // ~1 s of sustained load: does the clock drop?
    return 0;
}
