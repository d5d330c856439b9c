// Small device kernels used by the PointNet++ / EdgeConv / loss compositions built on the hot-path ops
// (Common/pointnet_util.py, gcn_lib/tf_vertex.py, Common/loss_utils.py in the reference, where each of these is
// a chain of generic TensorFlow ops).  All are HBM-bound element-wise / small-reduction kernels.
#include "common.h"

namespace dispu {

static inline int mgrid(long total, int bs) {
    long g = (total + bs - 1) / bs;
    if (g > 32768) g = 32768;
    if (g < 1) g = 1;
    return (int)g;
}

// grouped[r, s, :] -= center[r, :]     ("translation normalization", pointnet_util.py:43, loss_utils.py:281)
__global__ void group_center_kernel(long rows, int ns, int c, float* __restrict__ grouped, const float* __restrict__ center) {
    const long total = rows * ns * c;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int l = (int)(e % c);
        const long r = e / ((long)ns * c);
        grouped[e] = grouped[e] - center[r * c + l];
    }
}

// Pooling over the nsample axis of X[rows, ns, c] (pointnet_util.py:121-140):
//   0 max | 1 avg (sum in s order, / ns) | 2 "min" = max(-x) exactly as the reference computes it (it never negates
//   back) | 3 weighted_avg: w_s = exp(-5*|gxyz_s|) / sum_s exp(-5*|gxyz_s|) | 4 max_and_avg -> [max | avg] (2c)
//   5 sum in s order (GIN aggregation, gcn_lib/tf_vertex.py:248)
__global__ void pool_nsample_kernel(long rows, int ns, int c, int mode, const float* __restrict__ X,
                                    const float* __restrict__ gxyz, float* __restrict__ out) {
    const long total = rows * c;
    const int co = (mode == 4) ? 2 * c : c;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int l = (int)(e % c);
        const long r = e / c;
        const float* x = X + r * ns * c + l;
        if (mode == 3) {
            const float* g = gxyz + r * ns * 3;
            float den = 0.f;
            for (int s = 0; s < ns; ++s) {
                const float d = sqrtf((g[s * 3] * g[s * 3] + g[s * 3 + 1] * g[s * 3 + 1]) + g[s * 3 + 2] * g[s * 3 + 2]);
                den += expf(-d * 5.0f);
            }
            float acc = 0.f;
            for (int s = 0; s < ns; ++s) {
                const float d = sqrtf((g[s * 3] * g[s * 3] + g[s * 3 + 1] * g[s * 3 + 1]) + g[s * 3 + 2] * g[s * 3 + 2]);
                acc += x[(long)s * c] * (expf(-d * 5.0f) / den);
            }
            out[r * co + l] = acc;
            continue;
        }
        float mx = -__builtin_inff(), sum = 0.f;
        for (int s = 0; s < ns; ++s) {
            const float v = x[(long)s * c];
            mx = fmaxf(mx, (mode == 2) ? -v : v);
            sum += v;
        }
        if (mode == 0 || mode == 2) out[r * co + l] = mx;
        else if (mode == 5) out[r * co + l] = sum;
        else if (mode == 1) out[r * co + l] = sum / (float)ns;
        else { out[r * co + l] = mx; out[r * co + c + l] = sum / (float)ns; }
    }
}

// tf.nn.l2_normalize(x, axis=-1) (GraphSAGE, gcn_lib/tf_vertex.py:133-134): x * rsqrt(max(sum_c x^2, 1e-12)); the sum runs in
// channel order, the reciprocal square root is 1 / sqrtf (correctly rounded), one lane per row.
__global__ void l2_normalize_rows_kernel(long rows, int c, const float* __restrict__ X, float* __restrict__ out) {
    for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (long)gridDim.x * blockDim.x) {
        const float* x = X + r * c;
        float ss = 0.f;
        for (int l = 0; l < c; ++l) ss += x[l] * x[l];
        const float inv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
        for (int l = 0; l < c; ++l) out[r * c + l] = x[l] * inv;
    }
}

// out = x * alpha + y, elementwise (GIN: inputs * (1 + epsilon) + aggregated, gcn_lib/tf_vertex.py:205)
__global__ void scale_add_kernel(long n, const float* __restrict__ x, float alpha, const float* __restrict__ y, float* __restrict__ out) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) out[e] = x[e] * alpha + y[e];
}

// pointnet_fp_module's inverse-distance weights (pointnet_util.py:204-208): d = max(d, 1e-10); w = (1/d) / sum(1/d)
__global__ void idw_weights_kernel(long rows, const float* __restrict__ dist, float* __restrict__ weight) {
    for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (long)gridDim.x * blockDim.x) {
        const float i0 = 1.0f / fmaxf(dist[r * 3 + 0], 1e-10f);
        const float i1 = 1.0f / fmaxf(dist[r * 3 + 1], 1e-10f);
        const float i2 = 1.0f / fmaxf(dist[r * 3 + 2], 1e-10f);
        const float norm = (i0 + i1) + i2;
        weight[r * 3 + 0] = i0 / norm;
        weight[r * 3 + 1] = i1 / norm;
        weight[r * 3 + 2] = i2 / norm;
    }
}

// tf_util.get_edge_feature (Common/tf_util.py:654-686) / ops.get_edge_feature: out[(i,s)] = [F_i | F_j - F_i]
__global__ void edge_feature_kernel(long rows, int n_per_cloud, int k, int c, const float* __restrict__ F, long ldf,
                                    const int* __restrict__ idx, int ldi, int ioff, float* __restrict__ out, long ldo) {
    const long total = rows * k * c;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int l = (int)(e % c);
        const long pr = e / c;
        const long i = pr / k;
        const int s = (int)(pr - i * k);
        const long j = (i / n_per_cloud) * n_per_cloud + idx[i * ldi + ioff + s];
        const float fi = F[i * ldf + l];
        out[pr * ldo + l] = fi;
        out[pr * ldo + c + l] = F[j * ldf + l] - fi;
    }
}

// per-row mean and max of x[b, n] (Chamfer / Hausdorff reductions, loss_utils.py:59-63,78-83); one workgroup per row,
// fixed reduction order (deterministic).
__global__ __launch_bounds__(256) void row_mean_max_kernel(int n, const float* __restrict__ x, float* __restrict__ mean,
                                                            float* __restrict__ mx) {
    __shared__ float ssum[4], smax[4];
    const float* r = x + (size_t)blockIdx.x * n;
    float s = 0.f, m = -__builtin_inff();
    for (int i = threadIdx.x; i < n; i += 256) { s += r[i]; m = fmaxf(m, r[i]); }
    s = wave_sum_f32(s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) { ssum[threadIdx.x >> 6] = s; smax[threadIdx.x >> 6] = m; }
    __syncthreads();
    if (threadIdx.x == 0) {
        mean[blockIdx.x] = (((ssum[0] + ssum[1]) + ssum[2]) + ssum[3]) / (float)n;
        mx[blockIdx.x] = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
    }
}

// get_repulsion_loss core (loss_utils.py:280-296): for point i with grouped neighbours idx[i, 0:ns] (ball query, padded):
// d_s = |p_j - p_i|^2 (or L1), take the 5 smallest (top_k(-d, 5): ascending, earlier slot first on ties), drop the
// first, out[i] = sum_{t=1..4} max(0, h - d_t).   ns <= 64.
__global__ void repulsion_kernel(long rows, int n_per_cloud, int ns, int use_l1, float h, const float* __restrict__ pred,
                                 const int* __restrict__ idx, float* __restrict__ out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += (long)gridDim.x * blockDim.x) {
        const long base = (i / n_per_cloud) * n_per_cloud;
        const float px = pred[i * 3], py = pred[i * 3 + 1], pz = pred[i * 3 + 2];
        float b[5] = {__builtin_inff(), __builtin_inff(), __builtin_inff(), __builtin_inff(), __builtin_inff()};
        for (int s = 0; s < ns; ++s) {
            const long j = base + idx[i * ns + s];
            const float dx = pred[j * 3] - px, dy = pred[j * 3 + 1] - py, dz = pred[j * 3 + 2] - pz;
            const float d = use_l1 ? ((fabsf(dx) + fabsf(dy)) + fabsf(dz)) : ((dx * dx + dy * dy) + dz * dz);
            if (d < b[4]) {          // stable insertion: strict '<' keeps the earlier slot first on ties
                b[4] = d;
#pragma unroll
                for (int t = 4; t > 0; --t)
                    if (b[t] < b[t - 1]) { const float tmp = b[t]; b[t] = b[t - 1]; b[t - 1] = tmp; }
            }
        }
        float acc = 0.f;
#pragma unroll
        for (int t = 1; t < 5; ++t) acc += fmaxf(0.0f, h - b[t]);
        out[i] = acc;
    }
}

}  // namespace dispu

using namespace dispu;

DISPU_EXPORT int dispu_group_center(long rows, int ns, int c, float* grouped, const float* center, void* stream) {
    if (rows < 0 || ns <= 0 || c <= 0) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    hipLaunchKernelGGL(group_center_kernel, dim3(mgrid(rows * ns * c, 256)), dim3(256), 0, (hipStream_t)stream, rows, ns, c, grouped, center);
    return (int)hipGetLastError();
}

DISPU_EXPORT int dispu_pool_nsample(long rows, int ns, int c, int mode, const float* X, const float* gxyz, float* out,
                                    void* stream) {
    if (rows < 0 || ns <= 0 || c <= 0 || mode < 0 || mode > 5 || (mode == 3 && !gxyz)) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    hipLaunchKernelGGL(pool_nsample_kernel, dim3(mgrid(rows * c, 256)), dim3(256), 0, (hipStream_t)stream, rows, ns, c, mode, X, gxyz, out);
    return (int)hipGetLastError();
}

DISPU_EXPORT int dispu_l2_normalize_rows(long rows, int c, const float* X, float* out, void* stream) {
    if (rows < 0 || c <= 0) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    hipLaunchKernelGGL(l2_normalize_rows_kernel, dim3(mgrid(rows, 256)), dim3(256), 0, (hipStream_t)stream, rows, c, X, out);
    return (int)hipGetLastError();
}

DISPU_EXPORT int dispu_scale_add(long n, const float* x, float alpha, const float* y, float* out, void* stream) {
    if (n < 0) return (int)hipErrorInvalidValue;
    if (n == 0) return 0;
    hipLaunchKernelGGL(scale_add_kernel, dim3(mgrid(n, 256)), dim3(256), 0, (hipStream_t)stream, n, x, alpha, y, out);
    return (int)hipGetLastError();
}

DISPU_EXPORT int dispu_idw_weights(long rows, const float* dist, float* weight, void* stream) {
    if (rows < 0) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    hipLaunchKernelGGL(idw_weights_kernel, dim3(mgrid(rows, 256)), dim3(256), 0, (hipStream_t)stream, rows, dist, weight);
    return (int)hipGetLastError();
}

DISPU_EXPORT int dispu_edge_feature(long rows, int n_per_cloud, int k, int c, const float* F, long ldf, const int* idx,
                                    int ldi, int ioff, float* out, long ldo, void* stream) {
    if (rows < 0 || n_per_cloud <= 0 || k <= 0 || c <= 0) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    hipLaunchKernelGGL(edge_feature_kernel, dim3(mgrid(rows * k * c, 256)), dim3(256), 0, (hipStream_t)stream, rows, n_per_cloud, k, c, F, ldf, idx, ldi, ioff, out, ldo);
    return (int)hipGetLastError();
}

DISPU_EXPORT int dispu_row_mean_max(int b, int n, const float* x, float* mean, float* mx, void* stream) {
    if (b < 0 || n <= 0) return (int)hipErrorInvalidValue;
    if (b == 0) return 0;
    hipLaunchKernelGGL(row_mean_max_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, n, x, mean, mx);
    return (int)hipGetLastError();
}

DISPU_EXPORT int dispu_repulsion(long rows, int n_per_cloud, int ns, int use_l1, float h, const float* pred, const int* idx,
                                 float* out, void* stream) {
    if (rows < 0 || n_per_cloud <= 0 || ns < 5) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    hipLaunchKernelGGL(repulsion_kernel, dim3(mgrid(rows, 256)), dim3(256), 0, (hipStream_t)stream, rows, n_per_cloud, ns, use_l1, h, pred, idx, out);
    return (int)hipGetLastError();
}
