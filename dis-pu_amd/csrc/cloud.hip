// Whole-cloud inference glue around the patch generator (DisPU/model.py:306-381, Common/pc_util.py:83-92,147-161):
// patch extraction by k-NN with LARGE k (k = patch size = 256), per-patch normalisation and its inverse.
// In the reference these run on the host (sklearn NearestNeighbors, numpy) once per patch with batch size 1;
// here they are batched device kernels so the whole cloud goes through ONE generator launch.
#include "common.h"

namespace dispu {

// ---------------------------------------------------------------------------------------------
// extract_knn_patch (pc_util.py:83-92): for every query the k nearest cloud points, ascending distance
// (squared L2, plain arithmetic), ties -> lower index (sklearn's tie order is unspecified).
// One workgroup per query: all n keys (ordered distance bits << 32 | index) go to LDS, a bitonic network sorts
// them, the first k indices are written.  n <= 8192 (64 KiB of keys).
__global__ __launch_bounds__(256) void knn_patch_kernel(int n, int npad, int m, int k, const float* __restrict__ cloud,
                                                         const float* __restrict__ queries, int* __restrict__ idx) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
    const int b = blockIdx.y, qi = blockIdx.x;
    const float* __restrict__ pc = cloud + (size_t)b * n * 3;
    const float* __restrict__ q = queries + ((size_t)b * m + qi) * 3;
    const float qx = q[0], qy = q[1], qz = q[2];
    for (int p = threadIdx.x; p < npad; p += 256) {
        unsigned long long key = ~0ull;
        if (p < n) {
            const float d = sqdist3<false>(qx - pc[p * 3 + 0], qy - pc[p * 3 + 1], qz - pc[p * 3 + 2]) + 0.0f;
            key = ((unsigned long long)f32_to_ordered(d) << 32) | (unsigned)p;
        }
        keys[p] = key;
    }
    __syncthreads();
    for (int size = 2; size <= npad; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < (npad >> 1); t += 256) {
                const int lo = 2 * t - (t & (stride - 1));          // index of the lower element of the pair
                const int hi = lo + stride;
                const bool up = ((lo & size) == 0);                 // ascending block?
                const unsigned long long a = keys[lo], c = keys[hi];
                if ((a > c) == up) { keys[lo] = c; keys[hi] = a; }
            }
            __syncthreads();
        }
    }
    for (int t = threadIdx.x; t < k; t += 256) idx[((size_t)b * m + qi) * k + t] = (int)(unsigned)keys[t];
}

// normalize_point_cloud (pc_util.py:147-161) per patch: centroid = mean, p -= centroid, furthest = max |p|, p /= furthest.
// One wave per patch (n <= 64 * 16); sums use a fixed butterfly order (numpy's pairwise order is not reproduced;
// parity is tolerance-based, 1e-6).
__global__ __launch_bounds__(64) void normalize_patches_kernel(int n, const float* __restrict__ in, float* __restrict__ out,
                                                               float* __restrict__ centroid, float* __restrict__ furthest) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const float* __restrict__ p = in + (size_t)b * n * 3;
    // np.mean(input, axis=0 / 1) over a C-ordered [.., n, 3] array reduces the OUTER axis: numpy accumulates row after
    // row in float32 (its pairwise scheme only applies along a contiguous inner axis).  Lanes 0..2 replay exactly that
    // sequential sum for x / y / z so the centroid - and with it every normalised coordinate - is bit-identical.
    float s = 0.f;
    if (lane < 3)
        for (int i = 0; i < n; ++i) s = s + p[i * 3 + lane];
    s = s / (float)n;
    const float cx = __shfl(s, 0, 64), cy = __shfl(s, 1, 64), cz = __shfl(s, 2, 64);
    float mx = 0.f;
    for (int i = lane; i < n; i += 64) {
        const float dx = p[i * 3] - cx, dy = p[i * 3 + 1] - cy, dz = p[i * 3 + 2] - cz;
        mx = fmaxf(mx, sqrtf((dx * dx + dy * dy) + dz * dz));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float* __restrict__ o = out + (size_t)b * n * 3;
    for (int i = lane; i < n; i += 64) {
        o[i * 3] = (p[i * 3] - cx) / mx;
        o[i * 3 + 1] = (p[i * 3 + 1] - cy) / mx;
        o[i * 3 + 2] = (p[i * 3 + 2] - cz) / mx;
    }
    if (lane == 0) { centroid[b * 3] = cx; centroid[b * 3 + 1] = cy; centroid[b * 3 + 2] = cz; furthest[b] = mx; }
}

// pred = centroid + pred * furthest_distance (model.py:310-311), per patch
__global__ void denormalize_patches_kernel(long total, int m, const float* __restrict__ in, const float* __restrict__ centroid,
                                           const float* __restrict__ furthest, float* __restrict__ out) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long b = e / ((long)m * 3);
        const int c = (int)(e % 3);
        out[e] = centroid[b * 3 + c] + in[e] * furthest[b];
    }
}

}  // namespace dispu

using namespace dispu;

namespace dispu {
int knn_general_launch(int mode, int b, int n, int m, int c, int k, long ldp, long ldq, const float* points, const float* queries,
                       float* dist, int* idx, int neg, hipStream_t st);
}

DISPU_EXPORT int dispu_knn_patch(int b, int n, int m, int k, const float* cloud, const float* queries, int* idx, void* stream) {
    if (b < 0 || n <= 0 || m < 0 || k <= 0 || k > n) return (int)hipErrorInvalidValue;
    if (b == 0 || m == 0) return 0;
    // clouds whose 64-bit keys do not fit one workgroup's LDS: radix-select kernel (same distances, same tie rule), k <= 4096
    if (n > 8192) return dispu::knn_general_launch(0, b, n, m, 3, k, 3, 3, cloud, queries, nullptr, idx, 0, (hipStream_t)stream);
    int npad = 1;
    while (npad < n) npad <<= 1;
    if (npad < 2) npad = 2;
    const size_t bytes = (size_t)npad * 8;
    static DevOnce attr;      
    if (attr.needed()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(knn_patch_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        if (e != hipSuccess) return (int)e;
        attr.done();
    }
    hipLaunchKernelGGL(knn_patch_kernel, dim3(m, b), dim3(256), bytes, (hipStream_t)stream, n, npad, m, k, cloud, queries, idx);
    return (int)hipGetLastError();
}

DISPU_EXPORT int dispu_normalize_patches(int b, int n, const float* in, float* out, float* centroid, float* furthest, void* stream) {
    if (b < 0 || n <= 0) return (int)hipErrorInvalidValue;
    if (b == 0) return 0;
    hipLaunchKernelGGL(normalize_patches_kernel, dim3(b), dim3(64), 0, (hipStream_t)stream, n, in, out, centroid, furthest);
    return (int)hipGetLastError();
}

DISPU_EXPORT int dispu_denormalize_patches(int b, int m, const float* in, const float* centroid, const float* furthest, float* out,
                                           void* stream) {
    if (b < 0 || m <= 0) return (int)hipErrorInvalidValue;
    const long total = (long)b * m * 3;
    if (total == 0) return 0;
    long g = (total + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(denormalize_patches_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, total, m, in, centroid, furthest, out);
    return (int)hipGetLastError();
}
