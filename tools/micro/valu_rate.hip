// VALU issue-rate probe for gfx950: lane-operations per second of a few instruction kinds, one dependent chain per
// accumulator, 8 accumulators per lane, 4 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float seed) {
    float a[8];
    f2 p[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i * 0.001f + threadIdx.x * 1e-6f; p[i] = f2{a[i], a[i] * 0.5f}; }
    const float k = 0.999f;
    const f2 k2 = f2{0.999f, 0.998f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) a[i] = __builtin_fmaf(a[i], k, 0.001f);
            if (KIND == 1) p[i] = __builtin_elementwise_fma(p[i], k2, k2);
            if (KIND == 2) a[i] = __builtin_amdgcn_exp2f(a[i] * -0.5f);
            if (KIND == 3) a[i] = __builtin_amdgcn_sqrtf(a[i] + 1.0f);
            if (KIND == 4) a[i] = __builtin_amdgcn_rcpf(a[i] + 1.0f);
            if (KIND == 5) p[i] = p[i] * k2;
            if (KIND == 6) p[i] = p[i] + k2;
            if (KIND == 7) a[i] = __builtin_amdgcn_rsqf(a[i] + 1.0f);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND>
double run(const char* name, int ops_per_iter, int lanes_per_op) {
    float* out;
    const int blocks = 256 * 8, iters = 4096;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<KIND><<<blocks, 256>>>(out, 16, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<KIND><<<blocks, 256>>>(out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)blocks * 256 * iters * 8 * ops_per_iter;
    const double rate = instr * lanes_per_op / (ms * 1e-3);
    printf("%-28s %8.2f T lane-results/s   (%.3f ms)\n", name, rate / 1e12, ms);
    hipFree(out);
    return rate;
}

int main() {
    run<0>("v_fma_f32", 1, 1);
    run<1>("v_pk_fma_f32 (2 results)", 1, 2);
    run<5>("v_pk_mul_f32 (2 results)", 1, 2);
    run<6>("v_pk_add_f32 (2 results)", 1, 2);
    run<2>("v_mul + v_exp_f32", 1, 1);
    run<3>("v_add + v_sqrt_f32", 1, 1);
    run<4>("v_add + v_rcp_f32", 1, 1);
    run<7>("v_add + v_rsq_f32", 1, 1);
    return 0;
}
