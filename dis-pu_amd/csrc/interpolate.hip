// three_nn / three_interpolate (+grad) for gfx950.  The reference implements these as
// single-threaded CPU ops only (tf_ops/interpolation/tf_interpolate.cpp:60-153, DEVICE_CPU),
// forcing a device->host->device hop inside a GPU graph; here they are device kernels with the
// CPU functions' arithmetic (plain, un-contracted by default).
#include "common.h"

namespace dispu {

constexpr int NN3_BS = 256;
constexpr int NN3_TILE = 1024;

template <bool FMA>
__global__ __launch_bounds__(NN3_BS) void three_nn_kernel(int n, int m, const float* __restrict__ xyz1,
                                                           const float* __restrict__ xyz2, float* __restrict__ dist,
                                                           int* __restrict__ idx) {
    __shared__ float4 tile[NN3_TILE];
    const int cloud = blockIdx.y;
    const float* __restrict__ p1 = xyz1 + (size_t)cloud * n * 3;
    const float* __restrict__ p2 = xyz2 + (size_t)cloud * m * 3;
    const int j = blockIdx.x * NN3_BS + threadIdx.x;
    const bool active = j < n;
    float x1 = 0.f, y1 = 0.f, z1 = 0.f;
    if (active) { x1 = p1[j * 3 + 0]; y1 = p1[j * 3 + 1]; z1 = p1[j * 3 + 2]; }
    // the reference keeps the bests in double initialised to 1e40; every candidate is a float, so
    // float +inf is an equivalent sentinel (and what 1e40 becomes when stored to the float output)
    float b1 = __builtin_inff(), b2 = __builtin_inff(), b3 = __builtin_inff();
    int i1 = 0, i2 = 0, i3 = 0;
    for (int k0 = 0; k0 < m; k0 += NN3_TILE) {
        const int len = min(NN3_TILE, m - k0);
        __syncthreads();
        for (int t = threadIdx.x; t < len; t += NN3_BS)
            tile[t] = make_float4(p2[(k0 + t) * 3 + 0], p2[(k0 + t) * 3 + 1], p2[(k0 + t) * 3 + 2], 0.f);
        __syncthreads();
        for (int t = 0; t < len; ++t) {
            const float4 q = tile[t];
            const float d = sqdist3<FMA>(q.x - x1, q.y - y1, q.z - z1);
            const int k = k0 + t;
            if (d < b1) { b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = k; }
            else if (d < b2) { b3 = b2; i3 = i2; b2 = d; i2 = k; }
            else if (d < b3) { b3 = d; i3 = k; }
        }
    }
    if (active) {
        float* dd = dist + ((size_t)cloud * n + j) * 3;
        int* ii = idx + ((size_t)cloud * n + j) * 3;
        dd[0] = b1; dd[1] = b2; dd[2] = b3;
        ii[0] = i1; ii[1] = i2; ii[2] = i3;
    }
}

// out[b,j,l] = (p[i1,l]*w1 + p[i2,l]*w2) + p[i3,l]*w3   (left to right, every op rounded)
__global__ void three_interpolate_kernel(int m, int c, int n, size_t total, const float* __restrict__ points,
                                         const int* __restrict__ idx, const float* __restrict__ weight,
                                         float* __restrict__ out) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t row = e / c;
        const int l = (int)(e - row * c);
        const size_t cloud = row / n;
        const int* id = idx + row * 3;
        const float* w = weight + row * 3;
        const float* base = points + cloud * m * c + l;
        const float s = base[(size_t)id[0] * c] * w[0] + base[(size_t)id[1] * c] * w[1];
        out[e] = s + base[(size_t)id[2] * c] * w[2];
    }
}

__global__ void three_interpolate_grad_kernel(int m, int c, int n, size_t total, const float* __restrict__ grad_out,
                                              const int* __restrict__ idx, const float* __restrict__ weight,
                                              float* __restrict__ grad_points) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t row = e / c;
        const int l = (int)(e - row * c);
        const size_t cloud = row / n;
        const int* id = idx + row * 3;
        const float* w = weight + row * 3;
        float* base = grad_points + cloud * m * c + l;
        const float g = grad_out[e];
        unsafeAtomicAdd(base + (size_t)id[0] * c, g * w[0]);
        unsafeAtomicAdd(base + (size_t)id[1] * c, g * w[1]);
        unsafeAtomicAdd(base + (size_t)id[2] * c, g * w[2]);
    }
}

}  // namespace dispu

using namespace dispu;

static inline int grid_for(size_t total, int bs) {
    size_t g = (total + bs - 1) / bs;
    if (g > 16384) g = 16384;
    if (g < 1) g = 1;
    return (int)g;
}

DISPU_EXPORT int dispu_three_nn(int b, int n, int m, const float* xyz1, const float* xyz2, float* dist, int* idx,
                                int arith, void* stream) {
    if (b < 0 || n < 0 || m <= 0) return (int)hipErrorInvalidValue;
    if (b == 0 || n == 0) return 0;
    dim3 grid((n + NN3_BS - 1) / NN3_BS, b);
    if ((arith & DISPU_ARITH_CONTRACT))
        hipLaunchKernelGGL((three_nn_kernel<true>), grid, dim3(NN3_BS), 0, (hipStream_t)stream, n, m, xyz1, xyz2, dist, idx);
    else
        hipLaunchKernelGGL((three_nn_kernel<false>), grid, dim3(NN3_BS), 0, (hipStream_t)stream, n, m, xyz1, xyz2, dist, idx);
    return (int)hipGetLastError();
}

DISPU_EXPORT int dispu_three_interpolate(int b, int m, int c, int n, const float* points, const int* idx,
                                         const float* weight, float* out, void* stream) {
    if (b < 0 || n < 0 || m <= 0 || c <= 0) return (int)hipErrorInvalidValue;
    const size_t total = (size_t)b * n * c;
    if (total == 0) return 0;
    hipLaunchKernelGGL(three_interpolate_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, m, c, n,
                       total, points, idx, weight, out);
    return (int)hipGetLastError();
}

DISPU_EXPORT int dispu_three_interpolate_grad(int b, int n, int c, int m, const float* grad_out, const int* idx,
                                              const float* weight, float* grad_points, void* stream) {
    if (b < 0 || n < 0 || m <= 0 || c <= 0) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    if (b) DISPU_TRY(hipMemsetAsync(grad_points, 0, sizeof(float) * (size_t)b * m * c, s));
    const size_t total = (size_t)b * n * c;
    if (total == 0) return 0;
    hipLaunchKernelGGL(three_interpolate_grad_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, m, c, n, total,
                       grad_out, idx, weight, grad_points);
    return (int)hipGetLastError();
}
