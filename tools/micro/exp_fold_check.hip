// __expf(level * d2) == exp2f(d2 * (level * log2e)) bit for bit for the auction's levels (0 and -(4^j), powers of two)?
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off exp_fold_check.hip -o exp_fold_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k(unsigned* out, const float* x, int n, float level) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = __expf(level * x[i]);
    const float b = __builtin_amdgcn_exp2f(x[i] * (level * 0x1.715476p+0f));
    if (__float_as_uint(a) != __float_as_uint(b)) atomicAdd(out, 1u);
}
int main() {
    const int n = 1 << 22;
    std::vector<float> h(n);
    srand(1);
    for (int i = 0; i < n; ++i) { const float u = rand() / (float)RAND_MAX; h[i] = (i & 1) ? u * u * 4.0f : u * 1e-3f; }
    float* x; unsigned* out;
    hipMalloc(&x, n * 4); hipMalloc(&out, 4);
    hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice);
    unsigned total = 0;
    for (int j = 7; j >= -2; --j) {
        float level = 0.f;
        if (j != -2) { level = -1.f; for (int t = 0; t < (j < 0 ? -j : j); ++t) level = (j < 0) ? level * 0.25f : level * 4.0f; }
        hipMemset(out, 0, 4);
        k<<<n / 256, 256>>>(out, x, n, level);
        unsigned c; hipMemcpy(&c, out, 4, hipMemcpyDeviceToHost);
        printf("level %g: %u of %d differ\n", level, c, n);
        total += c;
    }
    printf("TOTAL %u\n", total);
    return total != 0;
}
