// Bidirectional nearest-neighbour (Chamfer) distance and its gradient for gfx950.
// Replaces NmDistanceKernelLauncher / NmDistanceGradKernelLauncher
// (tf_ops/nn_distance/tf_nndistance_g.cu:128-131,152-157) behind dispu_nn_distance(_grad).
//
// One launch covers BOTH directions (blockIdx.z) instead of the reference's two serial launches
// of a fixed 32x16 grid: a workgroup owns 64 points of the "from" cloud, the "to" cloud streams
// through LDS as float4 so each candidate costs one broadcast ds_read_b128 + 9 VALU ops.
// Tie rule: strict '<' in ascending index order == lowest index wins (tf_nndistance_g.cu:29,119).
#include "common.h"

namespace dispu {

constexpr int NND_BS = 256;
constexpr int NND_TILE = 2048;
constexpr int NND_Q = 64;            // queries per workgroup; its 4 waves scan one quarter of every candidate tile each

// Round 2: a workgroup owns 64 points of the "from" cloud and its FOUR waves split the candidates (wave p scans quarter p
// of every LDS tile), so (32, 1024, 1024) is 4096 waves instead of 1024 (one per SIMD, LDS-latency bound: 40 us).  A wave
// keeps the reference's rule inside its candidates (strict '<' in ascending index order); the four partial results are
// combined by (distance, index) order, which is the lowest index among the global minima - the same answer as one
// sequential scan (tf_nndistance_g.cu:29,119), including its `k == 0` rule: wave 0 starts from candidate 0
// unconditionally, the others from (+inf, INT_MAX).
template <bool FMA>
__global__ __launch_bounds__(NND_BS) void nn_distance_kernel(int n, int m, const float* __restrict__ xyz1,
                                                              const float* __restrict__ xyz2,
                                                              float* __restrict__ dist1, int* __restrict__ idx1,
                                                              float* __restrict__ dist2, int* __restrict__ idx2) {
    __shared__ float4 tile[NND_TILE];
    __shared__ float pd[4][NND_Q];
    __shared__ int pi[4][NND_Q];
    const int cloud = blockIdx.y;
    const bool fwd = blockIdx.z == 0;
    const int nf = fwd ? n : m, nt = fwd ? m : n;
    if (blockIdx.x * NND_Q >= nf) return;  // block-uniform
    const float* __restrict__ from = (fwd ? xyz1 : xyz2) + (size_t)cloud * nf * 3;
    const float* __restrict__ to = (fwd ? xyz2 : xyz1) + (size_t)cloud * nt * 3;
    float* __restrict__ od = (fwd ? dist1 : dist2) + (size_t)cloud * nf;
    int* __restrict__ oi = (fwd ? idx1 : idx2) + (size_t)cloud * nf;
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int j = blockIdx.x * NND_Q + lane;
    const bool active = j < nf;
    float x1 = 0.f, y1 = 0.f, z1 = 0.f;
    if (active) { x1 = from[j * 3 + 0]; y1 = from[j * 3 + 1]; z1 = from[j * 3 + 2]; }
    // `if (k == 0 || d < best)` of the reference (tf_nndistance_g.cu:29): candidate 0 is taken unconditionally
    // (matters only for inf / NaN distances); it is revisited in the loop where d < d is false.
    float best = (part == 0) ? sqdist3<FMA>(to[0] - x1, to[1] - y1, to[2] - z1) : __builtin_inff();
    int besti = (part == 0) ? 0 : 0x7fffffff;
    for (int k0 = 0; k0 < nt; k0 += NND_TILE) {
        const int len = min(NND_TILE, nt - k0);
        __syncthreads();
        for (int t = threadIdx.x; t < len; t += NND_BS)
            tile[t] = make_float4(to[(k0 + t) * 3 + 0], to[(k0 + t) * 3 + 1], to[(k0 + t) * 3 + 2], 0.f);
        __syncthreads();
        const int q4 = (len + 3) >> 2;
        const int t0 = part * q4, t1 = min(len, t0 + q4);
#pragma unroll 8
        for (int t = t0; t < t1; ++t) {
            const float4 q = tile[t];
            const float d = sqdist3<FMA>(q.x - x1, q.y - y1, q.z - z1);
            if (d < best) { best = d; besti = k0 + t; }
        }
    }
    pd[part][lane] = best;
    pi[part][lane] = besti;
    __syncthreads();
    if (part == 0 && active) {
#pragma unroll
        for (int p = 1; p < 4; ++p) {
            const float d = pd[p][lane];
            const int i = pi[p][lane];
            if (d < best || (d == best && i < besti)) { best = d; besti = i; }
        }
        od[j] = best;
        oi[j] = besti;
    }
}

// g = 2*grad_dist[j]; grad_from[j] += g*(p1-p2); grad_to[idx[j]] -= g*(p1-p2)   (tf_nndistance_g.cu:132-151)
__global__ void nn_distance_grad_kernel(int n, int m, const float* __restrict__ xyz1, const float* __restrict__ xyz2,
                                        const float* __restrict__ grad_dist1, const int* __restrict__ idx1,
                                        const float* __restrict__ grad_dist2, const int* __restrict__ idx2,
                                        float* __restrict__ grad_xyz1, float* __restrict__ grad_xyz2) {
    const int cloud = blockIdx.y;
    const bool fwd = blockIdx.z == 0;
    const int nf = fwd ? n : m, nt = fwd ? m : n;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nf) return;
    const float* from = (fwd ? xyz1 : xyz2) + (size_t)cloud * nf * 3;
    const float* to = (fwd ? xyz2 : xyz1) + (size_t)cloud * nt * 3;
    float* gfrom = (fwd ? grad_xyz1 : grad_xyz2) + (size_t)cloud * nf * 3;
    float* gto = (fwd ? grad_xyz2 : grad_xyz1) + (size_t)cloud * nt * 3;
    const float* gd = (fwd ? grad_dist1 : grad_dist2) + (size_t)cloud * nf;
    const int* id = (fwd ? idx1 : idx2) + (size_t)cloud * nf;
    const int j2 = id[j];
    const float g = gd[j] * 2;
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        const float v = g * (from[j * 3 + l] - to[j2 * 3 + l]);
        unsafeAtomicAdd(gfrom + j * 3 + l, v);
        unsafeAtomicAdd(gto + j2 * 3 + l, -v);
    }
}

}  // namespace dispu

using namespace dispu;

DISPU_EXPORT int dispu_nn_distance(int b, int n, const float* xyz1, int m, const float* xyz2, float* dist1, int* idx1,
                                   float* dist2, int* idx2, int arith, void* stream) {
    if (b < 0 || n <= 0 || m <= 0) return (int)hipErrorInvalidValue;
    if (b == 0) return 0;
    const int mx = n > m ? n : m;
    dim3 grid((mx + NND_Q - 1) / NND_Q, b, 2);
    if ((arith & DISPU_ARITH_CONTRACT))
        hipLaunchKernelGGL((nn_distance_kernel<true>), grid, dim3(NND_BS), 0, (hipStream_t)stream, n, m, xyz1, xyz2, dist1,
                           idx1, dist2, idx2);
    else
        hipLaunchKernelGGL((nn_distance_kernel<false>), grid, dim3(NND_BS), 0, (hipStream_t)stream, n, m, xyz1, xyz2, dist1,
                           idx1, dist2, idx2);
    return (int)hipGetLastError();
}

DISPU_EXPORT int dispu_nn_distance_grad(int b, int n, const float* xyz1, int m, const float* xyz2,
                                        const float* grad_dist1, const int* idx1, const float* grad_dist2,
                                        const int* idx2, float* grad_xyz1, float* grad_xyz2, void* stream) {
    if (b < 0 || n <= 0 || m <= 0) return (int)hipErrorInvalidValue;
    if (b == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    DISPU_TRY(hipMemsetAsync(grad_xyz1, 0, sizeof(float) * (size_t)b * n * 3, s));
    DISPU_TRY(hipMemsetAsync(grad_xyz2, 0, sizeof(float) * (size_t)b * m * 3, s));
    const int mx = n > m ? n : m;
    dim3 grid((mx + 255) / 256, b, 2);
    hipLaunchKernelGGL(nn_distance_grad_kernel, grid, dim3(256), 0, s, n, m, xyz1, xyz2, grad_dist1, idx1, grad_dist2,
                       idx2, grad_xyz1, grad_xyz2);
    return (int)hipGetLastError();
}
