// three_nn / three_interpolate (+grad) for gfx950.  The reference implements these as
// single-threaded CPU ops only (tf_ops/interpolation/tf_interpolate.cpp:60-153, DEVICE_CPU),
// forcing a device->host->device hop inside a GPU graph; here they are device kernels with the
// CPU functions' arithmetic (plain, un-contracted by default).
#include "common.h"

namespace dispu {

constexpr int NN3_BS = 256;
constexpr int NN3_TILE = 1024;
constexpr int NN3_Q = 64;             // queries per workgroup; its 4 waves scan one quarter of every candidate tile each

// sorted insert of (d, k) into the ascending triple, branch-free: strict '<' keeps the earlier (lower) index on ties,
// exactly like the if / else-if chain of threenn_cpu (tf_interpolate.cpp:77-93) when candidates arrive in ascending index
__device__ __forceinline__ void nn3_insert(float d, int k, float& b1, float& b2, float& b3, int& i1, int& i2, int& i3) {
    const bool c1 = d < b1, c2 = d < b2, c3 = d < b3;
    i3 = c2 ? i2 : (c3 ? k : i3);
    i2 = c1 ? i1 : (c2 ? k : i2);
    i1 = c1 ? k : i1;
    b3 = c2 ? b2 : (c3 ? d : b3);
    b2 = c1 ? b1 : (c2 ? d : b2);
    b1 = c1 ? d : b1;
}

// Round 2: 64 queries per workgroup, the four waves split the candidates (4x the waves: (32, 1024, 256) had 512 waves on
// 1024 SIMDs) and the divergent three-way insertion became 12 selects.  Every wave keeps a sorted triple of its candidates;
// the four triples are merged by (distance, index) order, which is what one sequential scan with strict '<' produces.
template <bool FMA>
__global__ __launch_bounds__(NN3_BS) void three_nn_kernel(int n, int m, const float* __restrict__ xyz1,
                                                           const float* __restrict__ xyz2, float* __restrict__ dist,
                                                           int* __restrict__ idx) {
    __shared__ float4 tile[NN3_TILE];
    __shared__ float pd[4][3][NN3_Q];
    __shared__ int pi[4][3][NN3_Q];
    const int cloud = blockIdx.y;
    const float* __restrict__ p1 = xyz1 + (size_t)cloud * n * 3;
    const float* __restrict__ p2 = xyz2 + (size_t)cloud * m * 3;
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int j = blockIdx.x * NN3_Q + lane;
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
        const int q4 = (len + 3) >> 2;
        const int t0 = part * q4, t1 = min(len, t0 + q4);
#pragma unroll 4
        for (int t = t0; t < t1; ++t) {
            const float4 q = tile[t];
            nn3_insert(sqdist3<FMA>(q.x - x1, q.y - y1, q.z - z1), k0 + t, b1, b2, b3, i1, i2, i3);
        }
    }
    pd[part][0][lane] = b1; pd[part][1][lane] = b2; pd[part][2][lane] = b3;
    pi[part][0][lane] = i1; pi[part][1][lane] = i2; pi[part][2][lane] = i3;
    __syncthreads();
    if (part == 0 && active) {
        // merge: insert the other waves' entries in (distance, index) order.  An entry that never received a candidate is
        // (+inf, 0): it can only tie with other +inf entries, whose index the reference leaves at 0 as well.
#pragma unroll
        for (int p = 1; p < 4; ++p)
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                const float d = pd[p][e][lane];
                const int k = pi[p][e][lane];
                const bool c1 = d < b1 || (d == b1 && k < i1), c2 = d < b2 || (d == b2 && k < i2), c3 = d < b3 || (d == b3 && k < i3);
                i3 = c2 ? i2 : (c3 ? k : i3);
                i2 = c1 ? i1 : (c2 ? k : i2);
                i1 = c1 ? k : i1;
                b3 = c2 ? b2 : (c3 ? d : b3);
                b2 = c1 ? b1 : (c2 ? d : b2);
                b1 = c1 ? d : b1;
            }
        float* dd = dist + ((size_t)cloud * n + j) * 3;
        int* ii = idx + ((size_t)cloud * n + j) * 3;
        dd[0] = b1; dd[1] = b2; dd[2] = b3;
        ii[0] = i1; ii[1] = i2; ii[2] = i3;
    }
}

// out[b,j,l] = (p[i1,l]*w1 + p[i2,l]*w2) + p[i3,l]*w3   (left to right, every op rounded: threeinterpolate_cpu,
// tf_interpolate.cpp:106-127).  HBM-bound.  Same slot scheme as group_rows_kernel (csrc/grouping.hip): TX lanes
// own an output row and stride over its c/VEC vectors, R rows per slot, so 3 R independent gathers per lane are in flight;
// the row's three indices and weights are loaded once per slot lane instead of once per element; non-temporal float4
// stores; XCD-contiguous row ranges.
template <int VEC, int R>
__global__ __launch_bounds__(256) void three_interpolate_rows_kernel(int m, int c, unsigned n, unsigned rows, int tx_log2,
                                                                     const float* __restrict__ points, const int* __restrict__ idx,
                                                                     const float* __restrict__ weight, float* __restrict__ out) {
    const unsigned TX = 1u << tx_log2, tx = threadIdx.x & (TX - 1), slot = threadIdx.x >> tx_log2, slots = 256u >> tx_log2;
    const unsigned base = xcd_block(blockIdx.x, gridDim.x) * (slots * R) + slot;
    const int cv = c / VEC;
    const float *s0[R], *s1[R], *s2[R];
    float w0[R], w1[R], w2[R];
    bool ok[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const unsigned row = base + r * slots;
        ok[r] = row < rows;
        const unsigned rr = ok[r] ? row : 0u;
        const int* id = idx + (size_t)rr * 3;
        const float* w = weight + (size_t)rr * 3;
        const float* cloud = points + (size_t)(rr / n) * m * c;
        s0[r] = cloud + (size_t)id[0] * c; s1[r] = cloud + (size_t)id[1] * c; s2[r] = cloud + (size_t)id[2] * c;
        w0[r] = w[0]; w1[r] = w[1]; w2[r] = w[2];
    }
    for (int l = tx; l < cv; l += TX) {
        if constexpr (VEC == 4) {
            float4 a[R], b[R], d[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                a[r] = reinterpret_cast<const float4*>(s0[r])[l];
                b[r] = reinterpret_cast<const float4*>(s1[r])[l];
                d[r] = reinterpret_cast<const float4*>(s2[r])[l];
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float4 o;
                o.x = (a[r].x * w0[r] + b[r].x * w1[r]) + d[r].x * w2[r];
                o.y = (a[r].y * w0[r] + b[r].y * w1[r]) + d[r].y * w2[r];
                o.z = (a[r].z * w0[r] + b[r].z * w1[r]) + d[r].z * w2[r];
                o.w = (a[r].w * w0[r] + b[r].w * w1[r]) + d[r].w * w2[r];
                if (ok[r]) store_nt4(out + (size_t)(base + r * slots) * c + 4 * l, o);
            }
        } else {
            float a[R], b[R], d[R];
#pragma unroll
            for (int r = 0; r < R; ++r) { a[r] = s0[r][l]; b[r] = s1[r][l]; d[r] = s2[r][l]; }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float o = (a[r] * w0[r] + b[r] * w1[r]) + d[r] * w2[r];
                if (ok[r]) __builtin_nontemporal_store(o, out + (size_t)(base + r * slots) * c + l);
            }
        }
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
    dim3 grid((n + NN3_Q - 1) / NN3_Q, b);
    if ((arith & DISPU_ARITH_CONTRACT))
        hipLaunchKernelGGL((three_nn_kernel<true>), grid, dim3(NN3_BS), 0, (hipStream_t)stream, n, m, xyz1, xyz2, dist, idx);
    else
        hipLaunchKernelGGL((three_nn_kernel<false>), grid, dim3(NN3_BS), 0, (hipStream_t)stream, n, m, xyz1, xyz2, dist, idx);
    return (int)hipGetLastError();
}

DISPU_EXPORT int dispu_three_interpolate(int b, int m, int c, int n, const float* points, const int* idx,
                                         const float* weight, float* out, void* stream) {
    if (b < 0 || n < 0 || m <= 0 || c <= 0) return (int)hipErrorInvalidValue;
    const size_t rows = (size_t)b * n;
    if (rows == 0) return 0;
    if (rows >= 0x7fffffffull) return (int)hipErrorInvalidValue;            // 32-bit row arithmetic
    const bool vec4 = (c % 4 == 0) && (((uintptr_t)points | (uintptr_t)out) % 16 == 0);
    const int cv = vec4 ? c / 4 : c;
    int tx_log2 = 0;
    while ((1 << tx_log2) < cv && tx_log2 < 6) ++tx_log2;
    const unsigned slots = 256u >> tx_log2;
    constexpr int R = 4;
    const unsigned g = ((unsigned)rows + slots * R - 1) / (slots * R);
    if (vec4)
        hipLaunchKernelGGL((three_interpolate_rows_kernel<4, R>), dim3(g), dim3(256), 0, (hipStream_t)stream, m, c, (unsigned)n,
                           (unsigned)rows, tx_log2, points, idx, weight, out);
    else
        hipLaunchKernelGGL((three_interpolate_rows_kernel<1, R>), dim3(g), dim3(256), 0, (hipStream_t)stream, m, c, (unsigned)n,
                           (unsigned)rows, tx_log2, points, idx, weight, out);
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
