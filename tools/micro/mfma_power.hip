// Micro-benchmark: the fp32 matrix pipe under a GEMM-like load -- 8 accumulator tiles per wave, operands read from LDS every k-step
// (2 A + 4 B dwords per lane and 8 MFMAs, the fragment traffic of linear.hip's 128 x 256 tile), RANDOM operand data (mfma_clock.hip
// multiplies the constant 1.0: no toggling).  No global traffic, no barriers, no loaders: what is left is the pipe, the LDS reads and the
// chip's power management.  Reports TFLOP/s for one short launch and for ten launches back to back (~3 ms of sustained load).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_power.hip -o tools/micro/bin/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v16f __attribute__((ext_vector_type(16)));

template <bool READS, bool RANDOM>
__global__ __launch_bounds__(256) void gemm_like(int slabs, const float* __restrict__ src, float* out) {
    __shared__ float lds[2 * 16 * 388];                           // two "stages" of 16 k-rows x (128 + 260) floats
    for (int e = threadIdx.x; e < 2 * 16 * 388; e += 256) lds[e] = RANDOM ? src[e] : 1.0f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fi = lane & 31, fk = lane >> 5, wm = wave >> 1, wn = wave & 1;
    v16f acc[2][4];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float af[2] = {lds[fi], lds[32 + fi]}, bf[4] = {lds[128 + fi], lds[160 + fi], lds[192 + fi], lds[224 + fi]};
    for (int t = 0; t < slabs; ++t) {
        const float* st = lds + (t & 1) * (16 * 388);
#pragma unroll
        for (int kk = 0; kk < 16; kk += 2) {
            if (READS) {
                const float* ar = st + (kk + fk) * 388 + wm * 64 + fi;
                const float* br = st + (kk + fk) * 388 + 128 + wn * 128 + fi;
                af[0] = ar[0]; af[1] = ar[32];
                bf[0] = br[0]; bf[1] = br[32]; bf[2] = br[64]; bf[3] = br[96];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 12345.678f) out[0] = s;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double g_last = 0.0;     // TFLOP/s of the last measurement
template <bool READS, bool RANDOM>
static void run(const char* what, const float* src, float* out) {
    const int slabs = 128, blocks = 256;                          // = one after_conv launch: 128 slabs of 64 MFMAs per wave
    const double flops = (double)blocks * 4 * slabs * 64 * 4096.0;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int reps : {1, 10, 40}) {
        hipLaunchKernelGGL((gemm_like<READS, RANDOM>), dim3(blocks), dim3(256), 0, 0, slabs, src, out);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((gemm_like<READS, RANDOM>), dim3(blocks), dim3(256), 0, 0, slabs, src, out);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-44s %2d launches back to back: %7.1f us per launch  %6.1f TFLOP/s  (%.3f of 157.3)\n", what, reps, ms * 1e3 / reps,
               flops * reps / ms / 1e9, flops * reps / ms / 1e9 / 157.3);
        if (reps == 10) g_last = flops * reps / ms / 1e9;
        CK(hipDeviceSynchronize());
    }
}

int main() {
    std::vector<float> h(2 * 16 * 388);
    unsigned s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; }
    float *src, *out;
    CK(hipMalloc(&src, h.size() * 4)); CK(hipMalloc(&out, 4));
    CK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    run<false, false>("MFMAs only, constant operands", src, out);
    const double only = g_last;
    run<false, true>("MFMAs only, random operands", src, out);
    run<true, false>("MFMAs + LDS fragment reads, constant data", src, out);
    const double reads_const = g_last;
    run<true, true>("MFMAs + LDS fragment reads, random data", src, out);
    // one machine-readable line for bench.py (10 launches back to back each)
    printf("{\"mfma_only_tflops\": %.1f, \"mfma_lds_reads_constant_data_tflops\": %.1f, \"mfma_lds_reads_random_data_tflops\": %.1f}\n", only, reads_const, g_last);
    return 0;
}
