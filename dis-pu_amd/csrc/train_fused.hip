// Training step, round 3: the PointShuffle2 local cell and skip branch WITHOUT their [B*M*16, 134] pair tensors.
//
// The reference graph (Common/ops.py:1012-1087) groups [xyz_j - xyz_i | xyz_j | feat_j] into a [B, M, 16, 134] tensor and runs
// conv0 / the skip max / weight_net over it; TF1 autodiff then walks the same tensors backwards.  Round 2's training step
// materialised all of them (ps_group -> gf, dgf, h0, h1, wl, wv ...: ~0.5 GB of traffic at 8 patches, 113 us of float
// atomics in ps_group_grad alone).  Here the forward is the inference path (conv0 evaluated per SOURCE point:
// relu(G[j] - A[i]), csrc/mlp_misc.hip:ps_prep; local cell fused in csrc/ps_local.hip; skip = gather-max) with BatchNorm on
// batch statistics, and the backward differentiates THAT form:
//   * weight_net (3 -> 16, BN, ReLU): batch statistics and all of its gradients come from the neighbour offsets directly
//     (ps_wnet_* kernels); wl is never stored;
//   * conv0: dz0[(i,s)] scatters to dG[j] and -dA[i]; the k-NN graph is INVERTED once per step (per-cloud CSR built in LDS,
//     lists sorted by pair id) so dG is a deterministic gather-sum of coalesced 512-byte rows instead of 16.8 M atomics;
//     dW0 / dup128 / dcoarse then come from [B*M, 128] matrices (16x fewer rows than the pair tensors);
//   * skip: the max gradient goes straight to the arg-max neighbours (ties share evenly, math_grad._MinOrMaxGrad).
// Every kernel cites the forward op whose gradient it is.  Streaming / gather kernels: lanes along the channel axis.
#include "common.h"

namespace dispu {

static inline int tf_grid(size_t total, int block) {
    size_t g = (total + block - 1) / block;
    return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g));
}

// bf16 activation storage (Trainer(dtype="bf16")): the [B*M*16, 128] pair tensors and dF' are kept as bf16 in HBM
__device__ __forceinline__ float4 tf_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 tf_ld4(const __bf16* p) {
    const uint2 w = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xFFFF0000u), __uint_as_float(w.y << 16), __uint_as_float(w.y & 0xFFFF0000u));
}
__device__ __forceinline__ float tf_ld1(const float* p) { return *p; }
__device__ __forceinline__ float tf_ld1(const __bf16* p) { return (float)*p; }
__device__ __forceinline__ void tf_st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void tf_st1(__bf16* p, float v) { *p = (__bf16)v; }
typedef __bf16 tf_bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void tf_st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void tf_st4(__bf16* p, float4 v) {
    tf_bf16x4 h = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
    *reinterpret_cast<tf_bf16x4*>(p) = h;
}

template <int CTRL>
__device__ __forceinline__ float tf_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float tf_row16_sum(float v) {                 // fixed order: ((v + ror8) + ror4) + ror2) + ror1
    v += tf_dpp<0x128>(v);
    v += tf_dpp<0x124>(v);
    v += tf_dpp<0x122>(v);
    v += tf_dpp<0x121>(v);
    return v;
}

__device__ __forceinline__ double tf_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- sum = relu(after_conv) + relu(skip) + relu(non-local) (ops.py:1072-1075): the three branch gradients in one pass ------
// o_i = d * (Y_i > 0); float4 over [rows, n] matrices (n % 4 == 0, 16-byte aligned rows).
__global__ void mask3_kernel(size_t total4, int n4, const float* __restrict__ d, long ldd, const float* __restrict__ y1, long ld1,
                             const float* __restrict__ y2, long ld2, const float* __restrict__ y3, long ld3, float* __restrict__ o1,
                             float* __restrict__ o2, float* __restrict__ o3, long ldo) {
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total4; e += (size_t)gridDim.x * blockDim.x) {
        const size_t r = e / n4;
        const int q = (int)(e - r * n4) * 4;
        const float4 g = *reinterpret_cast<const float4*>(d + r * ldd + q);
        const float4 a = *reinterpret_cast<const float4*>(y1 + r * ld1 + q);
        const float4 b = *reinterpret_cast<const float4*>(y2 + r * ld2 + q);
        const float4 c = *reinterpret_cast<const float4*>(y3 + r * ld3 + q);
        *reinterpret_cast<float4*>(o1 + r * ldo + q) = make_float4(a.x > 0.f ? g.x : 0.f, a.y > 0.f ? g.y : 0.f, a.z > 0.f ? g.z : 0.f, a.w > 0.f ? g.w : 0.f);
        *reinterpret_cast<float4*>(o2 + r * ldo + q) = make_float4(b.x > 0.f ? g.x : 0.f, b.y > 0.f ? g.y : 0.f, b.z > 0.f ? g.z : 0.f, b.w > 0.f ? g.w : 0.f);
        *reinterpret_cast<float4*>(o3 + r * ldo + q) = make_float4(c.x > 0.f ? g.x : 0.f, c.y > 0.f ? g.y : 0.f, c.z > 0.f ? g.z : 0.f, c.w > 0.f ? g.w : 0.f);
    }
}

// ---- weight_net_hidden in training mode (ops.py:181-191: conv 3 -> 16, contrib batch_norm on batch statistics, ReLU) -------
// One workgroup = 256 threads = the 16 neighbours x 16 channels of ONE point per pass.  wl[(i,s), t] is the fmaf chain of
// ps_weight_net_kernel (csrc/mlp_misc.hip) so forward and backward agree bit for bit on every ReLU decision.
constexpr int WN_K = 16, WN_T = 16;

__device__ __forceinline__ float wn_wl(const float* __restrict__ xyz, long i, long j, const float* __restrict__ Ww, const float* __restrict__ bw,
                                       int t, float& dx, float& dy, float& dz) {
    dx = xyz[j * 3 + 0] - xyz[i * 3 + 0]; dy = xyz[j * 3 + 1] - xyz[i * 3 + 1]; dz = xyz[j * 3 + 2] - xyz[i * 3 + 2];
    float acc = 0.f;
    acc = __builtin_fmaf(dx, Ww[0 * WN_T + t], acc);
    acc = __builtin_fmaf(dy, Ww[1 * WN_T + t], acc);
    acc = __builtin_fmaf(dz, Ww[2 * WN_T + t], acc);
    return acc + bw[t];
}

// part[blk][0][16] = sum wl, part[blk][1][16] = sum wl^2 (double), over the points blk, blk + grid, ...
__global__ __launch_bounds__(256) void ps_wnet_stats_kernel(long rows, int n_per_cloud, const int* __restrict__ idx, const float* __restrict__ xyz,
                                                             const float* __restrict__ Ww, const float* __restrict__ bw,
                                                             double* __restrict__ part) {
    __shared__ double red[2][256];
    const int s = threadIdx.x >> 4, t = threadIdx.x & 15;
    double s1 = 0.0, s2 = 0.0;
    // four points per pass, their index / coordinate loads requested together (one point per pass the workgroup walked its points
    // as a chain of dependent id -> xyz round trips); summed in the same point order
    const float w0 = Ww[0 * WN_T + t], w1 = Ww[1 * WN_T + t], w2 = Ww[2 * WN_T + t], bt = bw[t];
    for (long i0 = blockIdx.x; i0 < rows; i0 += 4 * (long)gridDim.x) {
        long ii[4], jj[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            ii[u] = min(i0 + u * (long)gridDim.x, rows - 1);
            jj[u] = (ii[u] / n_per_cloud) * n_per_cloud + idx[ii[u] * WN_K + s];
        }
        float ox[4], oy[4], oz[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            ox[u] = xyz[jj[u] * 3 + 0] - xyz[ii[u] * 3 + 0];
            oy[u] = xyz[jj[u] * 3 + 1] - xyz[ii[u] * 3 + 1];
            oz[u] = xyz[jj[u] * 3 + 2] - xyz[ii[u] * 3 + 2];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i0 + u * (long)gridDim.x < rows) {
                float acc = 0.f;                                         // wn_wl's chain
                acc = __builtin_fmaf(ox[u], w0, acc);
                acc = __builtin_fmaf(oy[u], w1, acc);
                acc = __builtin_fmaf(oz[u], w2, acc);
                const double v = acc + bt;
                s1 += v;
                s2 += v * v;
            }
        }
    }
    red[0][threadIdx.x] = s1;
    red[1][threadIdx.x] = s2;
    __syncthreads();
    if (threadIdx.x < WN_T) {
        double a = 0.0, b = 0.0;
        for (int g = 0; g < WN_K; ++g) { a += red[0][g * WN_T + threadIdx.x]; b += red[1][g * WN_T + threadIdx.x]; }
        part[((size_t)blockIdx.x * 2 + 0) * WN_T + threadIdx.x] = a;
        part[((size_t)blockIdx.x * 2 + 1) * WN_T + threadIdx.x] = b;
    }
}

// stats[0:16] mean | [16:32] biased variance | [32:48] 1/sqrt(var + eps); scale = gamma * inv_std, shift = beta - mean * scale
// (what ps_local / ps_weight_net apply as wl * scale + shift); moving statistics as bn_finalize_kernel (fused-BN semantics:
// Bessel-corrected variance into the moving average).  One wave per channel.
__global__ __launch_bounds__(64) void ps_wnet_stats_finalize_kernel(long count, int nparts, const double* __restrict__ part, const float* __restrict__ gamma,
                                                                     const float* __restrict__ beta, float eps, float decay, float* __restrict__ stats,
                                                                     float* __restrict__ scale, float* __restrict__ shift,
                                                                     float* __restrict__ moving_mean, float* __restrict__ moving_var) {
    const int ch = blockIdx.x;
    double a = 0.0, b = 0.0;
    for (int p0 = threadIdx.x; p0 < nparts; p0 += 8 * 64) {      // eight partials per lane requested together (was a chain of round trips)
        double va[8], vb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int p = p0 + u * 64;
            va[u] = p < nparts ? part[((size_t)p * 2 + 0) * WN_T + ch] : 0.0;
            vb[u] = p < nparts ? part[((size_t)p * 2 + 1) * WN_T + ch] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { a += va[u]; b += vb[u]; }
    }
    a = tf_wave_sum(a);
    b = tf_wave_sum(b);
    if (threadIdx.x != 0) return;
    const double mean = a / (double)count;
    double var = b / (double)count - mean * mean;
    if (var < 0.0) var = 0.0;
    const double is = 1.0 / sqrt(var + (double)eps);
    stats[ch] = (float)mean;
    stats[WN_T + ch] = (float)var;
    stats[2 * WN_T + ch] = (float)is;
    const float sc = (float)((double)gamma[ch] * is);
    scale[ch] = sc;
    shift[ch] = (float)((double)beta[ch] - mean * (double)sc);
    if (moving_mean) moving_mean[ch] = (float)((double)decay * moving_mean[ch] + (1.0 - (double)decay) * mean);
    if (moving_var) {
        const double unbiased = count > 1 ? var * ((double)count / (double)(count - 1)) : var;
        moving_var[ch] = (float)((double)decay * moving_var[ch] + (1.0 - (double)decay) * unbiased);
    }
}

// backward, pass 1: u = dwv * (wv > 0) with wv = relu(wl * scale + shift) recomputed; part[blk][0][16] = sum u,
// part[blk][1][16] = sum u * xhat, xhat = (wl - mean) * inv_std
__global__ __launch_bounds__(256) void ps_wnet_grad_stats_kernel(long rows, int n_per_cloud, const int* __restrict__ idx, const float* __restrict__ xyz,
                                                                  const float* __restrict__ Ww, const float* __restrict__ bw,
                                                                  const float* __restrict__ stats, const float* __restrict__ scale,
                                                                  const float* __restrict__ shift, const float* __restrict__ dwv,
                                                                  double* __restrict__ part) {
    __shared__ double red[2][256];
    const int s = threadIdx.x >> 4, t = threadIdx.x & 15;
    const float mu = stats[t], is = stats[2 * WN_T + t], sc = scale[t], sh = shift[t];
    double s1 = 0.0, s2 = 0.0;
    const float w0 = Ww[0 * WN_T + t], w1 = Ww[1 * WN_T + t], w2 = Ww[2 * WN_T + t], bt = bw[t];
    for (long i0 = blockIdx.x; i0 < rows; i0 += 4 * (long)gridDim.x) {      // four points per pass (see ps_wnet_stats_kernel)
        long ii[4], jj[4];
        float g[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            ii[u] = min(i0 + u * (long)gridDim.x, rows - 1);
            jj[u] = (ii[u] / n_per_cloud) * n_per_cloud + idx[ii[u] * WN_K + s];
            g[u] = dwv[(ii[u] * WN_K + s) * WN_T + t];
        }
        float ox[4], oy[4], oz[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            ox[u] = xyz[jj[u] * 3 + 0] - xyz[ii[u] * 3 + 0];
            oy[u] = xyz[jj[u] * 3 + 1] - xyz[ii[u] * 3 + 1];
            oz[u] = xyz[jj[u] * 3 + 2] - xyz[ii[u] * 3 + 2];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i0 + u * (long)gridDim.x < rows) {
                float acc = 0.f;
                acc = __builtin_fmaf(ox[u], w0, acc);
                acc = __builtin_fmaf(oy[u], w1, acc);
                acc = __builtin_fmaf(oz[u], w2, acc);
                const float wl = acc + bt;
                const float wv = wl * sc + sh;
                const float uu = (wv > 0.f) ? g[u] : 0.f;
                const float xh = (wl - mu) * is;
                s1 += (double)uu;
                s2 += (double)uu * (double)xh;
            }
        }
    }
    red[0][threadIdx.x] = s1;
    red[1][threadIdx.x] = s2;
    __syncthreads();
    if (threadIdx.x < WN_T) {
        double a = 0.0, b = 0.0;
        for (int g = 0; g < WN_K; ++g) { a += red[0][g * WN_T + threadIdx.x]; b += red[1][g * WN_T + threadIdx.x]; }
        part[((size_t)blockIdx.x * 2 + 0) * WN_T + threadIdx.x] = a;
        part[((size_t)blockIdx.x * 2 + 1) * WN_T + threadIdx.x] = b;
    }
}

// sums[0:16] = sum u, sums[16:32] = sum u * xhat;  dbeta += sum u, dgamma += sum u * xhat
__global__ __launch_bounds__(64) void ps_wnet_grad_finalize_kernel(int nparts, const double* __restrict__ part, float* __restrict__ sums,
                                                                    float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int ch = blockIdx.x;
    double a = 0.0, b = 0.0;
    for (int p0 = threadIdx.x; p0 < nparts; p0 += 8 * 64) {      // eight partials per lane requested together (was a chain of round trips)
        double va[8], vb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int p = p0 + u * 64;
            va[u] = p < nparts ? part[((size_t)p * 2 + 0) * WN_T + ch] : 0.0;
            vb[u] = p < nparts ? part[((size_t)p * 2 + 1) * WN_T + ch] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { a += va[u]; b += vb[u]; }
    }
    a = tf_wave_sum(a);
    b = tf_wave_sum(b);
    if (threadIdx.x != 0) return;
    sums[ch] = (float)a;
    sums[WN_T + ch] = (float)b;
    if (dbeta) dbeta[ch] += (float)a;
    if (dgamma) dgamma[ch] += (float)b;
}

// backward, pass 2: dwl = gamma inv_std (u - sum_u / n - xhat sum_uxhat / n)  (the batch_norm gradient), then the 3 -> 16 conv:
// dWw[c][t] += sum offset_c dwl_t, dbw[t] += sum dwl_t, and d offset_c = sum_t dwl_t Ww[c][t] goes to the two points of the pair:
// dxyz[j] += d offset, dxyz[i] -= d offset.  Workgroup = one point at a time (16 neighbours x 16 channels), looping over points;
// the weight gradients are kept per thread over the loop, reduced once per workgroup through LDS, and leave as 64 atomics.
__global__ __launch_bounds__(256) void ps_wnet_grad_apply_kernel(long rows, int n_per_cloud, const int* __restrict__ idx, const float* __restrict__ xyz,
                                                                  const float* __restrict__ Ww, const float* __restrict__ bw,
                                                                  const float* __restrict__ stats, const float* __restrict__ scale,
                                                                  const float* __restrict__ shift, const float* __restrict__ gamma,
                                                                  const float* __restrict__ sums, const float* __restrict__ dwv,
                                                                  float* __restrict__ dWw, float* __restrict__ dbw, float* __restrict__ dxyz) {
    constexpr int PTS = 8;                         // points per pass: their index / coordinate / dwv loads are all issued before the first use
    __shared__ float red[4][256];
    const int s = threadIdx.x >> 4, t = threadIdx.x & 15;
    const float mu = stats[t], is = stats[2 * WN_T + t], sc = scale[t], sh = shift[t];
    const float inv_n = 1.0f / (float)(rows * WN_K);
    const float m1 = sums[t] * inv_n, m2 = sums[WN_T + t] * inv_n, gi = gamma[t] * is;
    const float w0 = Ww[0 * WN_T + t], w1 = Ww[1 * WN_T + t], w2 = Ww[2 * WN_T + t], bt = bw[t];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, ab = 0.f;
    for (long i0 = (long)blockIdx.x * PTS; i0 < rows; i0 += (long)gridDim.x * PTS) {
        long ii[PTS], jj[PTS];
        float g[PTS], ox[PTS], oy[PTS], oz[PTS];
#pragma unroll
        for (int u = 0; u < PTS; ++u) {
            ii[u] = min(i0 + u, rows - 1);
            jj[u] = (ii[u] / n_per_cloud) * n_per_cloud + idx[ii[u] * WN_K + s];
            g[u] = dwv[(ii[u] * WN_K + s) * WN_T + t];
        }
#pragma unroll
        for (int u = 0; u < PTS; ++u) {
            ox[u] = xyz[jj[u] * 3 + 0] - xyz[ii[u] * 3 + 0];
            oy[u] = xyz[jj[u] * 3 + 1] - xyz[ii[u] * 3 + 1];
            oz[u] = xyz[jj[u] * 3 + 2] - xyz[ii[u] * 3 + 2];
        }
#pragma unroll
        for (int u = 0; u < PTS; ++u) {
            const bool live = i0 + u < rows;
            float acc = 0.f;                                             // wn_wl's chain
            acc = __builtin_fmaf(ox[u], w0, acc);
            acc = __builtin_fmaf(oy[u], w1, acc);
            acc = __builtin_fmaf(oz[u], w2, acc);
            const float wl = acc + bt;
            const float wv = wl * sc + sh;
            const float uu = (wv > 0.f) ? g[u] : 0.f;
            const float xh = (wl - mu) * is;
            const float dwl = live ? gi * ((uu - m1) - xh * m2) : 0.f;
            a0 = __builtin_fmaf(ox[u], dwl, a0); a1 = __builtin_fmaf(oy[u], dwl, a1); a2 = __builtin_fmaf(oz[u], dwl, a2);
            ab += dwl;
            // d offset: sum over the 16 channels (= the 16 lanes of a DPP row)
            float o0 = dwl * w0, o1 = dwl * w1, o2 = dwl * w2;
            // row_ror 8 / 4 / 2 / 1: every lane of the 16-lane DPP row ends with the row sum (four VALU-DPP adds per value; as
            // __shfl_xor butterflies these were twelve ds_bpermute round trips per point)
            o0 = tf_row16_sum(o0); o1 = tf_row16_sum(o1); o2 = tf_row16_sum(o2);
            if (t == 0 && live) { unsafeAtomicAdd(dxyz + jj[u] * 3 + 0, o0); unsafeAtomicAdd(dxyz + jj[u] * 3 + 1, o1); unsafeAtomicAdd(dxyz + jj[u] * 3 + 2, o2); }
            // the point's own share: the 4 pairs of this wave are combined first (one atomic per wave and coordinate, not per pair)
            o0 += __shfl_xor(o0, 16, 64); o1 += __shfl_xor(o1, 16, 64); o2 += __shfl_xor(o2, 16, 64);
            o0 += __shfl_xor(o0, 32, 64); o1 += __shfl_xor(o1, 32, 64); o2 += __shfl_xor(o2, 32, 64);
            if ((threadIdx.x & 63) == 0 && live) {
                unsafeAtomicAdd(dxyz + ii[u] * 3 + 0, -o0); unsafeAtomicAdd(dxyz + ii[u] * 3 + 1, -o1); unsafeAtomicAdd(dxyz + ii[u] * 3 + 2, -o2);
            }
        }
    }
    red[0][threadIdx.x] = a0; red[1][threadIdx.x] = a1; red[2][threadIdx.x] = a2; red[3][threadIdx.x] = ab;
    __syncthreads();
    if (threadIdx.x < 64) {
        const int which = threadIdx.x >> 4, tt = threadIdx.x & 15;
        float v = 0.f;
        for (int gq = 0; gq < WN_K; ++gq) v += red[which][gq * WN_T + tt];
        if (which < 3) unsafeAtomicAdd(dWw + which * WN_T + tt, v);
        else if (dbw) unsafeAtomicAdd(dbw + tt, v);
    }
}

// ---- the k-NN graph inverted (DisPU PointShuffle2 grouping, ops.py:154-179: idx[i, s] = s-th neighbour of point i) -----------
// Per cloud (n <= 4096 points, k neighbours each): off[j] .. off[j+1] delimit, inside inv[], the pair ids i*k + s (cloud-local i)
// with idx[i, s] == j, ascending.  One workgroup per cloud: LDS histogram -> exclusive scan -> fill -> every list sorted
// (insertion sort; lists average k entries), which makes the gather-sum below deterministic.
constexpr int INV_MAXN = 4096;
__global__ __launch_bounds__(1024) void knn_invert_kernel(int n, int k, const int* __restrict__ idx, int* __restrict__ off, int* __restrict__ inv) {
    __shared__ int cnt[INV_MAXN + 1];
    __shared__ int cur[INV_MAXN];
    __shared__ int wsum[16];
    const int cloud = blockIdx.x, tid = threadIdx.x;
    const int* __restrict__ id = idx + (size_t)cloud * n * k;
    int* __restrict__ o = off + (size_t)cloud * (n + 1);
    int* __restrict__ iv = inv + (size_t)cloud * n * k;
    for (int e = tid; e <= n; e += 1024) cnt[e] = 0;
    __syncthreads();
    for (int e = tid; e < n * k; e += 1024) atomicAdd(&cnt[id[e]], 1);
    __syncthreads();
    // exclusive scan of cnt[0..n): thread t owns the 4 entries 4t .. 4t+3 (n <= 4096)
    int v[4], local = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int e = tid * 4 + q; v[q] = e < n ? cnt[e] : 0; local += v[q]; }
    int incl = local;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int up = __shfl_up(incl, d, 64); if ((tid & 63) >= d) incl += up; }
    if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < (tid >> 6); ++w) base += wsum[w];
    int run = base + incl - local;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = tid * 4 + q;
        if (e < n) { cnt[e] = run; cur[e] = run; o[e] = run; }
        run += v[q];
    }
    if (tid == 1023) { o[n] = n * k; }
    __syncthreads();
    for (int e = tid; e < n * k; e += 1024) {
        const int pos = atomicAdd(&cur[id[e]], 1);
        iv[pos] = e;
    }
    __syncthreads();
    __threadfence_block();
    for (int j = tid; j < n; j += 1024) {
        const int lo = cnt[j], hi = (j + 1 < n) ? cnt[j + 1] : n * k;
        for (int a = lo + 1; a < hi; ++a) {
            const int key = iv[a];
            int b = a - 1;
            while (b >= lo && iv[b] > key) { iv[b + 1] = iv[b]; --b; }
            iv[b + 1] = key;
        }
    }
}

// ---- conv0 per source point, backward (h0 = relu(G[j] - A[i]), csrc/mlp_misc.hip:ps_prep) ---------------------------------------
// dh0 [(i,s), c] (c = 128; the gradient w.r.t. h0, NOT yet multiplied by relu') ->  dz0 = dh0 * (G[j] - A[i] > 0), then
// dG[p] = sum over the in-edges of p (pairs (i,s) with idx[i,s] = p) of dz0[(i,s)],  dAneg[p] = -sum_s dz0[(p,s)].
// The ReLU decision is re-derived from G and A (two [B*M, 128] matrices that live in L2; the same fp32 subtraction the forward
// made) instead of from a stored h0, and instead of a masked epilogue on the [B*M*16, 128] GEMM that produces dh0.
// 32 lanes per point, one float4 of the 512-byte rows each; the in-edge rows are fetched 4 at a time.  No atomics: a point's
// sums are formed in a fixed order (the inverted lists are sorted).  Gm == NULL: dh0 is taken as already masked.
__device__ __forceinline__ float4 c0_masked(float4 v, float4 g, float4 a) {
    return make_float4((g.x - a.x > 0.f) ? v.x : 0.f, (g.y - a.y > 0.f) ? v.y : 0.f, (g.z - a.z > 0.f) ? v.z : 0.f, (g.w - a.w > 0.f) ? v.w : 0.f);
}
template <class TZ>
__global__ __launch_bounds__(256) void ps_conv0_gather_grad_kernel(long rows, int n_per_cloud, int k, const int* __restrict__ idx,
                                                                    const int* __restrict__ off, const int* __restrict__ inv,
                                                                    const TZ* __restrict__ dz0, long ldz, const float* __restrict__ Gm,
                                                                    long ldgm, const float* __restrict__ Am, long ldam, float* __restrict__ dG,
                                                                    long ldg, float* __restrict__ dAneg, long lda) {
    const int sub = threadIdx.x & 31;
    const long p = ((long)xcd_block(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x) >> 5;
    if (p >= rows) return;
    const long cloud = p / n_per_cloud, pl = p - cloud * n_per_cloud, base = cloud * n_per_cloud;
    const TZ* __restrict__ zc = dz0 + (size_t)cloud * n_per_cloud * k * ldz;           // this cloud's pair rows
    const bool msk = Gm != nullptr;
    float4 gp = make_float4(0.f, 0.f, 0.f, 0.f), ap = gp;
    if (msk) {
        gp = *reinterpret_cast<const float4*>(Gm + p * ldgm + sub * 4);
        ap = *reinterpret_cast<const float4*>(Am + p * ldam + sub * 4);
    }
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    // the point's own k pair rows, 8 at a time (as a rolled loop with a run-time trip count every row waited for its own loads:
    // 16 dependent round trips per point); added in ascending s as before
    const int mine = msk ? idx[p * k + (sub < k ? sub : 0)] : 0;
    for (int s0 = 0; s0 < k; s0 += 8) {
        float4 v[8], gj[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int s = (s0 + u < k) ? s0 + u : k - 1;
            v[u] = tf_ld4(zc + (size_t)(pl * k + s) * ldz + sub * 4);
            if (msk) gj[u] = *reinterpret_cast<const float4*>(Gm + (base + __shfl(mine, s, 32)) * ldgm + sub * 4);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (s0 + u < k) {
                const float4 w = msk ? c0_masked(v[u], gj[u], ap) : v[u];
                a.x += w.x; a.y += w.y; a.z += w.z; a.w += w.w;
            }
        }
    }
    *reinterpret_cast<float4*>(dAneg + p * lda + sub * 4) = make_float4(-a.x, -a.y, -a.z, -a.w);
    const int* __restrict__ o = off + cloud * (n_per_cloud + 1);
    const int* __restrict__ iv = inv + cloud * (size_t)n_per_cloud * k;
    const int lo = o[pl], hi = o[pl + 1];
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    int e = lo;
    for (; e + 8 <= hi; e += 8) {                               // eight in-edge rows in flight; summed in list order
        int ee[8];
        float4 v[8], am[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) ee[u] = iv[e + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            v[u] = tf_ld4(zc + (size_t)ee[u] * ldz + sub * 4);
            if (msk) am[u] = *reinterpret_cast<const float4*>(Am + (base + ee[u] / k) * ldam + sub * 4);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float4 w = msk ? c0_masked(v[u], gp, am[u]) : v[u];
            g.x += w.x; g.y += w.y; g.z += w.z; g.w += w.w;
        }
    }
    for (; e + 4 <= hi; e += 4) {
        const int e0 = iv[e], e1 = iv[e + 1], e2 = iv[e + 2], e3 = iv[e + 3];
        float4 v0 = tf_ld4(zc + (size_t)e0 * ldz + sub * 4);
        float4 v1 = tf_ld4(zc + (size_t)e1 * ldz + sub * 4);
        float4 v2 = tf_ld4(zc + (size_t)e2 * ldz + sub * 4);
        float4 v3 = tf_ld4(zc + (size_t)e3 * ldz + sub * 4);
        if (msk) {
            v0 = c0_masked(v0, gp, *reinterpret_cast<const float4*>(Am + (base + e0 / k) * ldam + sub * 4));
            v1 = c0_masked(v1, gp, *reinterpret_cast<const float4*>(Am + (base + e1 / k) * ldam + sub * 4));
            v2 = c0_masked(v2, gp, *reinterpret_cast<const float4*>(Am + (base + e2 / k) * ldam + sub * 4));
            v3 = c0_masked(v3, gp, *reinterpret_cast<const float4*>(Am + (base + e3 / k) * ldam + sub * 4));
        }
        g.x += v0.x; g.y += v0.y; g.z += v0.z; g.w += v0.w;
        g.x += v1.x; g.y += v1.y; g.z += v1.z; g.w += v1.w;
        g.x += v2.x; g.y += v2.y; g.z += v2.z; g.w += v2.w;
        g.x += v3.x; g.y += v3.y; g.z += v3.z; g.w += v3.w;
    }
    for (; e < hi; ++e) {
        const int ee = iv[e];
        float4 v = tf_ld4(zc + (size_t)ee * ldz + sub * 4);
        if (msk) v = c0_masked(v, gp, *reinterpret_cast<const float4*>(Am + (base + ee / k) * ldam + sub * 4));
        g.x += v.x; g.y += v.y; g.z += v.z; g.w += v.w;
    }
    *reinterpret_cast<float4*>(dG + p * ldg + sub * 4) = g;
}

// G[p] = feat_p.Wf + xyz_p.(Wc + Wr) + b0,  A[p] = xyz_p.Wc  (W0 rows 0:3 = Wc, 3:6 = Wr; ps_prep).  The xyz side of the backward:
//   dxyz[p] += dG[p].(Wc + Wr)^T + dAneg[p].Wc^T;   dW0[0:3] += xyz^T (dG + dAneg);   dW0[3:6] += xyz^T dG.
// One wave per point (lane = 2 channels); the weight-gradient sums stay in registers over the wave's points, are combined
// across the 4 waves through LDS and leave as 6 x 128 atomics per workgroup.
__global__ __launch_bounds__(256) void ps_prep_grad_kernel(long rows, const float* __restrict__ xyz, const float* __restrict__ W0 /*[134,128]*/,
                                                            const float* __restrict__ dG, long ldg, const float* __restrict__ dAneg, long lda,
                                                            float* __restrict__ dxyz, float* __restrict__ dW0) {
    constexpr int C = 128;
    __shared__ float red[4][6][C];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = lane * 2;
    float wc[3][2], wr[3][2];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        wc[r][0] = W0[r * C + c0]; wc[r][1] = W0[r * C + c0 + 1];
        wr[r][0] = W0[(3 + r) * C + c0]; wr[r][1] = W0[(3 + r) * C + c0 + 1];
    }
    float acc[6][2];
#pragma unroll
    for (int r = 0; r < 6; ++r) acc[r][0] = acc[r][1] = 0.f;
    // four points of the wave per iteration: their loads are requested together (one point at a time the wave walked its 8 points as
    // 8 dependent round trips); the sums are formed in the same point order as before
    const long stride = (long)gridDim.x * 4;
    for (long p0 = (long)blockIdx.x * 4 + wave; p0 < rows; p0 += 4 * stride) {
        float2 g4[4], a4[4];
        float x4[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long p = p0 + u * stride;
            const bool ok = p < rows;
            const long q = ok ? p : p0;
            g4[u] = *reinterpret_cast<const float2*>(dG + q * ldg + c0);
            a4[u] = *reinterpret_cast<const float2*>(dAneg + q * lda + c0);
            x4[u][0] = xyz[q * 3 + 0]; x4[u][1] = xyz[q * 3 + 1]; x4[u][2] = xyz[q * 3 + 2];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long p = p0 + u * stride;
            if (p >= rows) break;
            const float2 g = g4[u], a = a4[u];
            const float* x = x4[u];
            const float sx = g.x + a.x, sy = g.y + a.y;
            float o[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                acc[r][0] = __builtin_fmaf(x[r], sx, acc[r][0]); acc[r][1] = __builtin_fmaf(x[r], sy, acc[r][1]);
                acc[3 + r][0] = __builtin_fmaf(x[r], g.x, acc[3 + r][0]); acc[3 + r][1] = __builtin_fmaf(x[r], g.y, acc[3 + r][1]);
                // dxyz_r = sum_c dG_c (Wc + Wr)[r][c] + dAneg_c Wc[r][c]
                float v = g.x * (wc[r][0] + wr[r][0]) + g.y * (wc[r][1] + wr[r][1]);
                v = __builtin_fmaf(a.x, wc[r][0], v);
                v = __builtin_fmaf(a.y, wc[r][1], v);
                o[r] = wave_sum_f32(v);
            }
            if (lane < 3) unsafeAtomicAdd(dxyz + p * 3 + lane, lane == 0 ? o[0] : lane == 1 ? o[1] : o[2]);
        }
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) { red[wave][r][c0] = acc[r][0]; red[wave][r][c0 + 1] = acc[r][1]; }
    __syncthreads();
    for (int e = threadIdx.x; e < 6 * C; e += 256) {
        const int r = e / C, c = e - r * C;
        const float v = (red[0][r][c] + red[1][r][c]) + (red[2][r][c] + red[3][r][c]);
        unsafeAtomicAdd(dW0 + r * C + c, v);
    }
}

// ---- skip branch: gmax[i] = max_s [xyz_j - xyz_i | xyz_j | feat_j] (ops.py:1049; forward = ps_skip_max16_kernel) -----------------
// backward without the grouped tensor: the 16 candidates of every channel are gathered again (same loads, same subtraction:
// bit-equal to what the forward compared), the gradient is shared evenly by the entries equal to the maximum
// (math_grad._MinOrMaxGrad) and goes straight to its sources: dfeat[j][c] (atomics), dxyz[j] / dxyz[i].
// 32 lanes per point: lane q owns feature channels 4q .. 4q+3, lanes 0..5 the six xyz channels as well.
__global__ __launch_bounds__(256) void ps_skip_max_grad_kernel(long rows, int n_per_cloud, const int* __restrict__ idx, const float* __restrict__ xyz,
                                                                const float* __restrict__ feat, long ldf, const float* __restrict__ gmax, long ldm,
                                                                const float* __restrict__ dgmax, long ldd, float* __restrict__ dxyz,
                                                                float* __restrict__ dfeat, long lddf, int feat_is_relu) {
    const int sub = threadIdx.x & 31;
    const long i = ((long)xcd_block(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x) >> 5;
    if (i >= rows) return;
    const long base = (i / n_per_cloud) * n_per_cloud;
    const int mine = idx[i * 16 + (sub & 15)];
    const float ci = (sub < 3) ? xyz[i * 3 + sub] : 0.f;
    const int xc = sub < 3 ? sub : sub - 3;
    float4 v[16];
    float pj[16];
    long jj[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        jj[s] = base + __shfl(mine, s, 32);
        v[s] = *reinterpret_cast<const float4*>(feat + jj[s] * ldf + sub * 4);
        pj[s] = (sub < 6) ? xyz[jj[s] * 3 + xc] : 0.f;
    }
    const float* mrow = gmax + i * ldm;
    const float* grow = dgmax + i * ldd;
    const float mf[4] = {mrow[6 + sub * 4], mrow[7 + sub * 4], mrow[8 + sub * 4], mrow[9 + sub * 4]};
    const float gf[4] = {grow[6 + sub * 4], grow[7 + sub * 4], grow[8 + sub * 4], grow[9 + sub * 4]};
    int cnt[4] = {0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        cnt[0] += v[s].x == mf[0]; cnt[1] += v[s].y == mf[1]; cnt[2] += v[s].z == mf[2]; cnt[3] += v[s].w == mf[3];
    }
    float sh[4] = {gf[0] / (float)cnt[0], gf[1] / (float)cnt[1], gf[2] / (float)cnt[2], gf[3] / (float)cnt[3]};
    // feat_is_relu: feat is a ReLU output whose relu_grad is applied to dfeat afterwards.  A maximum of 0 means every one of the 16
    // candidates is a ReLU zero (the common case: a 16-way tie) and its share lands on entries that mask kills -- not sent at all.
    if (feat_is_relu) {
#pragma unroll
        for (int c = 0; c < 4; ++c) sh[c] = (mf[c] > 0.f) ? sh[c] : 0.f;
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        float* d = dfeat + jj[s] * lddf + sub * 4;
        if (v[s].x == mf[0] && sh[0] != 0.f) unsafeAtomicAdd(d + 0, sh[0]);
        if (v[s].y == mf[1] && sh[1] != 0.f) unsafeAtomicAdd(d + 1, sh[1]);
        if (v[s].z == mf[2] && sh[2] != 0.f) unsafeAtomicAdd(d + 2, sh[2]);
        if (v[s].w == mf[3] && sh[3] != 0.f) unsafeAtomicAdd(d + 3, sh[3]);
    }
    if (sub < 6) {
        const float mx = mrow[sub], gx = grow[sub];
        int c = 0;
#pragma unroll
        for (int s = 0; s < 16; ++s) c += ((sub < 3 ? pj[s] - ci : pj[s]) == mx);
        const float share = gx / (float)c;
        float self = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s)
            if ((sub < 3 ? pj[s] - ci : pj[s]) == mx) {
                unsafeAtomicAdd(dxyz + jj[s] * 3 + xc, share);
                self += share;
            }
        if (sub < 3) unsafeAtomicAdd(dxyz + i * 3 + sub, -self);
    }
}

// ---- feature x weight product backward with the ReLU of conv1 folded in (csrc/train_ops.hip:ps_point_matmul_grad_kernel) -------
// dz1[(i,s), c] = (h1 > 0) * sum_t dout[i, c*16 + t] wv[(i,s), t];  dwv[(i,s), t] = sum_c h1[(i,s), c] dout[i, c*16 + t]
// Round-3 rewrite: the first version kept dout / h1 / wv of a point in LDS and read two LDS words per multiply-add (512 LDS reads per
// lane and point: 86 us for 8192 points, LDS-issue-bound, 3x its HBM time).  Now
//   dz1: lane (c, half) holds its dout row [c][0..15] in REGISTERS (four ds_read_b128) and its eight h1 values,
//        reads wv[s][0..15] as four broadcast ds_read_b128 per s -> 32 LDS reads for 128 multiply-adds, coalesced stores over c;
//   dwv: lane (s, t-quad, c-quarter) walks 32 channels with one ds_read_b32 (h1) + one ds_read_b128 (dout) per 4 multiply-adds, the four
//        channel quarters are added through DPP in a fixed order.
template <class TS>
__global__ __launch_bounds__(256) void ps_point_matmul_grad_relu_kernel(long rows, const TS* __restrict__ X2, long ldx2,
                                                                         const float* __restrict__ wv, const TS* __restrict__ dout,
                                                                         long ldo, TS* __restrict__ dX2, long lddx2,
                                                                         float* __restrict__ dwv) {
    constexpr int K = 16, T = 16, C = 128, LDO = T + 4, LDX = C + 1;
    __shared__ __attribute__((aligned(16))) float s_do[C * LDO];
    __shared__ float s_x[K * LDX];
    __shared__ __attribute__((aligned(16))) float s_w[K * T];
    const int tid = threadIdx.x;
    const int c = tid & 127, half = tid >> 7;                    // phase 1 role
    const int s2 = tid >> 4, tq = (tid >> 2) & 3, cp = tid & 3;  // phase 2 role
    for (long i = blockIdx.x; i < rows; i += gridDim.x) {
        // coalesced float4 loads (a lane-per-row read of dout would touch every 64-byte line four times), transposed through LDS
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int idx = tid + u * 256;
            *reinterpret_cast<float4*>(&s_do[(idx >> 2) * LDO + (idx & 3) * 4]) = tf_ld4(dout + i * ldo + idx * 4);
            const float4 x = tf_ld4(X2 + (i * K + (idx >> 5)) * ldx2 + (idx & 31) * 4);
            float* xr = &s_x[(idx >> 5) * LDX + (idx & 31) * 4];
            xr[0] = x.x; xr[1] = x.y; xr[2] = x.z; xr[3] = x.w;
        }
        s_w[tid] = wv[(i * K) * T + tid];
        __syncthreads();
        float4 d[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) d[q] = *reinterpret_cast<const float4*>(&s_do[c * LDO + q * 4]);
        float hx[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) hx[j] = s_x[(half * 8 + j) * LDX + c];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int s = half * 8 + j;
            const float4* w4 = reinterpret_cast<const float4*>(&s_w[s * T]);
            float a = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 w = w4[q];
                a = __builtin_fmaf(d[q].x, w.x, a);
                a = __builtin_fmaf(d[q].y, w.y, a);
                a = __builtin_fmaf(d[q].z, w.z, a);
                a = __builtin_fmaf(d[q].w, w.w, a);
            }
            tf_st1(dX2 + (i * K + s) * lddx2 + c, (hx[j] > 0.f) ? a : 0.f);
        }
        {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
            for (int j = 0; j < 32; ++j) {
                const int cc = j * 4 + cp;
                const float x = s_x[s2 * LDX + cc];
                const float4 v = *reinterpret_cast<const float4*>(&s_do[cc * LDO + tq * 4]);
                acc.x = __builtin_fmaf(x, v.x, acc.x);
                acc.y = __builtin_fmaf(x, v.y, acc.y);
                acc.z = __builtin_fmaf(x, v.z, acc.z);
                acc.w = __builtin_fmaf(x, v.w, acc.w);
            }
            // (q0 + q1) + (q2 + q3) over the four channel quarters, the same on every lane of the quad
            acc.x += __shfl_xor(acc.x, 1, 64); acc.y += __shfl_xor(acc.y, 1, 64); acc.z += __shfl_xor(acc.z, 1, 64); acc.w += __shfl_xor(acc.w, 1, 64);
            acc.x += __shfl_xor(acc.x, 2, 64); acc.y += __shfl_xor(acc.y, 2, 64); acc.z += __shfl_xor(acc.z, 2, 64); acc.w += __shfl_xor(acc.w, 2, 64);
            if (cp == 0) *reinterpret_cast<float4*>(dwv + (i * K + s2) * T + tq * 4) = acc;
        }
        __syncthreads();
    }
}


// ---- pu_loss glue (DisPU/model.py:75-87, Common/loss_utils.py:45-64,271-298) ---------------------------------------------------
// Round 2 evaluated CD = mean_b[(mean_k dist_gt + mean_j dist_pred) / radius_b] and its constant per-row gradients with ~15 tiny
// tensor ops per Chamfer term (row means, adds, fills, memsets: 5 - 10 us of launch gap each; the loss took 0.46 ms of a 3 ms step).
// chamfer_value_kernel: ONE workgroup walks the clouds in order -> value[0] = sum_b (mean d_gt[b] + mean d_pred[b]) / radius[b] / B.
__global__ __launch_bounds__(1024) void chamfer_value_kernel(int b, int n_gt, int n_pred, const float* __restrict__ d_gt,
                                                              const float* __restrict__ d_pred, const float* __restrict__ radius,
                                                              float* __restrict__ value) {
    // a WAVE per cloud (clouds w, w + 16, ...), 8 loads in flight per lane; the first version walked the clouds one after the other
    // with all 1024 threads and two barriers each: 15 - 18 us for 8 clouds, on the loss chain.  Fixed order: lane partial sums over
    // ascending i, butterfly, clouds added in ascending order by thread 0.
    __shared__ float term[1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float total = 0.f;
    for (int c0 = 0; c0 < b; c0 += 1024) {                         // batches of up to 1024 clouds (LDS slots)
        for (int c = c0 + wave; c < min(b, c0 + 1024); c += 16) {
            float s1 = 0.f, s2 = 0.f;
            const float* g = d_gt + (size_t)c * n_gt;
            const float* q = d_pred + (size_t)c * n_pred;
            for (int i0 = lane; i0 < n_gt; i0 += 64 * 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = (i0 + 64 * u < n_gt) ? g[i0 + 64 * u] : 0.f;
#pragma unroll
                for (int u = 0; u < 8; ++u) s1 += v[u];
            }
            for (int i0 = lane; i0 < n_pred; i0 += 64 * 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = (i0 + 64 * u < n_pred) ? q[i0 + 64 * u] : 0.f;
#pragma unroll
                for (int u = 0; u < 8; ++u) s2 += v[u];
            }
            s1 = wave_sum_f32(s1);
            s2 = wave_sum_f32(s2);
            if (lane == 0) term[c - c0] = (s1 / (float)n_gt + s2 / (float)n_pred) / radius[c];
        }
        __syncthreads();
        if (threadIdx.x == 0)
            for (int c = c0; c < min(b, c0 + 1024); ++c) total += term[c - c0];
        __syncthreads();
    }
    if (threadIdx.x == 0) value[0] = total / (float)b;
}

// d(coef * CD)/d pred: nn_distance's gradient (tf_nndistance.py:31-37, tf_nndistance_g.cu:132-157) with the constant upstream
// gradients coef / (radius_b n B) of the two row means folded in; pred only (gt is data).  blockIdx.z = 0: every gt point pulls its
// nearest pred point; 1: every pred point is pulled towards its nearest gt point.  dpred must be zero-filled (atomics).
__global__ void chamfer_grad_kernel(int bcount, int n_gt, int n_pred, const float* __restrict__ gt, const float* __restrict__ pred,
                                    const int* __restrict__ i_gt, const int* __restrict__ i_pred, const float* __restrict__ radius,
                                    float coef, float* __restrict__ dpred) {
    const int cloud = blockIdx.y;
    const bool from_gt = blockIdx.z == 0;
    const int nf = from_gt ? n_gt : n_pred;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nf) return;
    const float* g0 = gt + (size_t)cloud * n_gt * 3;
    const float* p0 = pred + (size_t)cloud * n_pred * 3;
    float* dp = dpred + (size_t)cloud * n_pred * 3;
    const float inv_r = 1.0f / radius[cloud];
    const float g = (inv_r * (coef / ((float)nf * (float)bcount))) * 2;
    if (from_gt) {
        const int j2 = i_gt[(size_t)cloud * n_gt + j];
#pragma unroll
        for (int l = 0; l < 3; ++l) unsafeAtomicAdd(dp + j2 * 3 + l, -(g * (g0[j * 3 + l] - p0[j2 * 3 + l])));
    } else {
        const int j2 = i_pred[(size_t)cloud * n_pred + j];
#pragma unroll
        for (int l = 0; l < 3; ++l) unsafeAtomicAdd(dp + j * 3 + l, g * (p0[j * 3 + l] - g0[j2 * 3 + l]));
    }
}

// out[0] = 1000 cd_coarse, out[1] = 1000 cd_fine, out[2] = repulsion_w * mean(rep) / 4 (loss_utils.py:296-297: mean over
// [B, M, 4] of the hinge terms; rep[i] holds the sum over the 4 neighbours), out[3] = pu_loss = out[0] + wf out[1] + out[2]
// (model.py:87), out[4] = wf.  One workgroup, fixed order.
__global__ __launch_bounds__(1024) void pu_loss_finalize_kernel(const float* __restrict__ cd, const float* __restrict__ rep, long nrep, float wf,
                                                                 float rep_w, float* __restrict__ out) {
    __shared__ float red[16];
    float s = 0.f;
    if (rep)
        for (long i = threadIdx.x; i < nrep; i += 1024) s += rep[i];
    s = wave_sum_f32(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f;
        for (int w = 0; w < 16; ++w) a += red[w];
        const float r = rep ? rep_w * (a / ((float)nrep * 4.0f)) : 0.f;
        const float c = 1000.0f * cd[0], f = 1000.0f * cd[1];
        out[0] = c; out[1] = f; out[2] = r; out[3] = (c + wf * f) + r; out[4] = wf;
    }
}

// get_repulsion_loss (loss_utils.py:271-298) value AND gradient in one pass: per point the 20 ball neighbours' squared distances,
// the 2nd..5th smallest (stable, like top_k), out[i] = sum max(0, h - d), dpred += scale * d/d pred (both points of a pair, atomics).
// NS neighbour ids and 3 NS coordinates are requested before the first is used (the one-thread-per-point kernels of round 2 walked
// them as a dependent chain: 42 us for 8192 points); 64-thread workgroups spread the points over the CUs.
template <int NS>
__global__ __launch_bounds__(64) void repulsion_loss_grad_kernel(long rows, int n_per_cloud, float h, float scale, const float* __restrict__ pred,
                                                                  const int* __restrict__ idx, float* __restrict__ out, float* __restrict__ dpred) {
    const long i = (long)blockIdx.x * 64 + threadIdx.x;
    if (i >= rows) return;
    const long base = (i / n_per_cloud) * n_per_cloud;
    const float px = pred[i * 3], py = pred[i * 3 + 1], pz = pred[i * 3 + 2];
    int id[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) id[s] = idx[i * NS + s];
    float dx[NS], dy[NS], dz[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const long j = base + id[s];
        dx[s] = pred[j * 3] - px; dy[s] = pred[j * 3 + 1] - py; dz[s] = pred[j * 3 + 2] - pz;
    }
    float b[5] = {__builtin_inff(), __builtin_inff(), __builtin_inff(), __builtin_inff(), __builtin_inff()};
    int bs[5] = {-1, -1, -1, -1, -1};
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const float d = (dx[s] * dx[s] + dy[s] * dy[s]) + dz[s] * dz[s];
        if (d < b[4]) {                     // stable insertion: strict '<' keeps the earlier slot first on ties
            b[4] = d; bs[4] = s;
#pragma unroll
            for (int t = 4; t > 0; --t)
                if (b[t] < b[t - 1]) {
                    const float tmp = b[t]; b[t] = b[t - 1]; b[t - 1] = tmp;
                    const int ts = bs[t]; bs[t] = bs[t - 1]; bs[t - 1] = ts;
                }
        }
    }
    float acc = 0.f, gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll
    for (int t = 1; t < 5; ++t) {
        acc += fmaxf(0.0f, h - b[t]);
        if (bs[t] < 0 || !(h - b[t] > 0.f)) continue;
        float ex = 0.f, ey = 0.f, ez = 0.f;
        int jl = 0;
#pragma unroll
        for (int s = 0; s < NS; ++s)
            if (s == bs[t]) { ex = dx[s]; ey = dy[s]; ez = dz[s]; jl = id[s]; }
        const long j = base + jl;
        const float cx = 2.f * scale * ex, cy = 2.f * scale * ey, cz = 2.f * scale * ez;
        unsafeAtomicAdd(dpred + j * 3 + 0, -cx);
        unsafeAtomicAdd(dpred + j * 3 + 1, -cy);
        unsafeAtomicAdd(dpred + j * 3 + 2, -cz);
        gx += cx; gy += cy; gz += cz;
    }
    out[i] = acc;
    unsafeAtomicAdd(dpred + i * 3 + 0, gx);
    unsafeAtomicAdd(dpred + i * 3 + 1, gy);
    unsafeAtomicAdd(dpred + i * 3 + 2, gz);
}

// ---- W^T copies of the weight matrices (one launch per step): the dX = dZ . W^T products of the backward pass then read their
// B operand with k contiguous, i.e. take the untransposed (DMA / float4-staged) path of the forward GEMM instead of the transb path.
// desc[m] = {source offset, K, N} (floats / rows / columns of W [K][N] inside `src`); dst gets W^T [N][K] at the same offset.
__global__ __launch_bounds__(256) void transpose_batched_kernel(const int* __restrict__ desc, const float* __restrict__ src, float* __restrict__ dst) {
    __shared__ float tile[32][33];
    const int off = desc[blockIdx.y * 3 + 0], K = desc[blockIdx.y * 3 + 1], N = desc[blockIdx.y * 3 + 2];
    const int tn = (N + 31) / 32, tk = (K + 31) / 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int tI = blockIdx.x; tI < tn * tk; tI += gridDim.x) {
        const int k0 = (tI / tn) * 32, n0 = (tI % tn) * 32;
        __syncthreads();
#pragma unroll
        for (int r = ty; r < 32; r += 8)
            if (k0 + r < K && n0 + tx < N) tile[r][tx] = src[off + (size_t)(k0 + r) * N + n0 + tx];
        __syncthreads();
#pragma unroll
        for (int r = ty; r < 32; r += 8)
            if (n0 + r < N && k0 + tx < K) dst[off + (size_t)(n0 + r) * K + k0 + tx] = tile[tx][r];
    }
}

// h0[(i,s), c] = relu(G[cloud, idx[i,s], c] - A[i, c]) written as bf16 (csrc/mlp_misc.hip:ps_gather_sub_relu_kernel, bf16 storage)
__global__ void ps_gather_sub_relu_bf16_kernel(long rows, int n_per_cloud, int k, int c4n, const int* __restrict__ idx, const float* __restrict__ Gm,
                                               long ldg, const float* __restrict__ A, long lda, __bf16* __restrict__ X1, long ldx1) {
    const long total = rows * k * c4n;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(e % c4n);
        const long pr = e / c4n;
        const long i = pr / k;
        const long j = (i / n_per_cloud) * n_per_cloud + idx[pr];
        const float4 g = *reinterpret_cast<const float4*>(Gm + j * ldg + c4 * 4);
        const float4 a = *reinterpret_cast<const float4*>(A + i * lda + c4 * 4);
        tf_st4(X1 + pr * ldx1 + c4 * 4, make_float4(fmaxf(g.x - a.x, 0.f), fmaxf(g.y - a.y, 0.f), fmaxf(g.z - a.z, 0.f), fmaxf(g.w - a.w, 0.f)));
    }
}

static int wn_blocks(long rows) { return (int)(rows < 1024 ? rows : 1024); }

}  // namespace dispu

using namespace dispu;

DISPU_EXPORT int dispu_mask3(long rows, int n, const float* dY, long lddy, const float* Y1, long ld1, const float* Y2, long ld2,
                             const float* Y3, long ld3, float* o1, float* o2, float* o3, long ldo, void* stream) {
    if (rows < 0 || n <= 0 || (n & 3) || ((lddy | ld1 | ld2 | ld3 | ldo) & 3)) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    const size_t total4 = (size_t)rows * (n / 4);
    hipLaunchKernelGGL(mask3_kernel, dim3(tf_grid(total4, 256)), dim3(256), 0, (hipStream_t)stream, total4, n / 4, dY, lddy, Y1, ld1, Y2, ld2,
                       Y3, ld3, o1, o2, o3, ldo);
    return (int)hipGetLastError();
}

DISPU_EXPORT long dispu_ps_wnet_scratch_bytes(long rows) { return rows <= 0 ? 0 : (long)wn_blocks(rows) * 2 * WN_T * (long)sizeof(double); }

DISPU_EXPORT int dispu_ps_wnet_bn_stats(long rows, int n_per_cloud, int k, int t_n, const int* idx, const float* xyz, const float* Ww,
                                        const float* bw, const float* gamma, const float* beta, float eps, float decay, float* stats,
                                        float* scale, float* shift, float* moving_mean, float* moving_var, void* scratch,
                                        long scratch_bytes, void* stream) {
    if (rows <= 0 || k != WN_K || t_n != WN_T || n_per_cloud <= 0 || !scratch || scratch_bytes < dispu_ps_wnet_scratch_bytes(rows))
        return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    const int nb = wn_blocks(rows);
    hipLaunchKernelGGL(ps_wnet_stats_kernel, dim3(nb), dim3(256), 0, s, rows, n_per_cloud, idx, xyz, Ww, bw, (double*)scratch);
    DISPU_CHECK_LAUNCH();
    hipLaunchKernelGGL(ps_wnet_stats_finalize_kernel, dim3(WN_T), dim3(64), 0, s, rows * WN_K, nb, (const double*)scratch, gamma, beta, eps,
                       decay, stats, scale, shift, moving_mean, moving_var);
    return (int)hipGetLastError();
}

DISPU_EXPORT int dispu_ps_wnet_grad(long rows, int n_per_cloud, int k, int t_n, const int* idx, const float* xyz, const float* Ww,
                                    const float* bw, const float* stats, const float* scale, const float* shift, const float* gamma,
                                    const float* dwv, float* dWw, float* dbw, float* dgamma, float* dbeta, float* dxyz, float* sums,
                                    void* scratch, long scratch_bytes, void* stream) {
    if (rows <= 0 || k != WN_K || t_n != WN_T || n_per_cloud <= 0 || !scratch || !sums || scratch_bytes < dispu_ps_wnet_scratch_bytes(rows))
        return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    const int nb = wn_blocks(rows);
    hipLaunchKernelGGL(ps_wnet_grad_stats_kernel, dim3(nb), dim3(256), 0, s, rows, n_per_cloud, idx, xyz, Ww, bw, stats, scale, shift, dwv,
                       (double*)scratch);
    DISPU_CHECK_LAUNCH();
    hipLaunchKernelGGL(ps_wnet_grad_finalize_kernel, dim3(WN_T), dim3(64), 0, s, nb, (const double*)scratch, sums, dgamma, dbeta);
    DISPU_CHECK_LAUNCH();
    const int nba = (int)((rows + 7) / 8 < 2048 ? (rows + 7) / 8 : 2048);
    hipLaunchKernelGGL(ps_wnet_grad_apply_kernel, dim3(nba), dim3(256), 0, s, rows, n_per_cloud, idx, xyz, Ww, bw, stats, scale, shift, gamma,
                       sums, dwv, dWw, dbw, dxyz);
    return (int)hipGetLastError();
}

DISPU_EXPORT int dispu_knn_invert(int b, int n, int k, const int* idx, int* off, int* inv, void* stream) {
    if (b < 0 || n <= 0 || n > INV_MAXN || k <= 0 || !idx || !off || !inv) return (int)hipErrorInvalidValue;
    if (b == 0) return 0;
    hipLaunchKernelGGL(knn_invert_kernel, dim3(b), dim3(1024), 0, (hipStream_t)stream, n, k, idx, off, inv);
    return (int)hipGetLastError();
}

DISPU_EXPORT int dispu_ps_conv0_gather_grad_s(long rows, int n_per_cloud, int k, int c, const int* idx, const int* off, const int* inv,
                                              const void* dh0, long ldz, int dh0_bf16, const float* Gm, long ldgm, const float* Am, long ldam,
                                              float* dG, long ldg, float* dAneg, long lda, void* stream);

DISPU_EXPORT int dispu_ps_conv0_gather_grad(long rows, int n_per_cloud, int k, int c, const int* idx, const int* off, const int* inv,
                                            const float* dh0, long ldz, const float* Gm, long ldgm, const float* Am, long ldam, float* dG,
                                            long ldg, float* dAneg, long lda, void* stream) {
    return dispu_ps_conv0_gather_grad_s(rows, n_per_cloud, k, c, idx, off, inv, dh0, ldz, 0, Gm, ldgm, Am, ldam, dG, ldg, dAneg, lda, stream);
}

// the same with dh0 optionally STORED as bf16 (dh0_bf16 != 0; ldz in elements)
DISPU_EXPORT int dispu_ps_conv0_gather_grad_s(long rows, int n_per_cloud, int k, int c, const int* idx, const int* off, const int* inv,
                                              const void* dh0, long ldz, int dh0_bf16, const float* Gm, long ldgm, const float* Am, long ldam,
                                              float* dG, long ldg, float* dAneg, long lda, void* stream) {
    if (rows < 0 || c != 128 || n_per_cloud <= 0 || rows % n_per_cloud != 0 || ((ldz | ldg | lda | ldgm | ldam) & 3) || ((Gm == nullptr) != (Am == nullptr)) ||
        (Gm && !idx))
        return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    const dim3 grid((unsigned)((rows * 32 + 255) / 256));
    if (dh0_bf16)
        hipLaunchKernelGGL(ps_conv0_gather_grad_kernel<__bf16>, grid, dim3(256), 0, (hipStream_t)stream, rows, n_per_cloud, k, idx, off, inv,
                           (const __bf16*)dh0, ldz, Gm, ldgm, Am, ldam, dG, ldg, dAneg, lda);
    else
        hipLaunchKernelGGL(ps_conv0_gather_grad_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, rows, n_per_cloud, k, idx, off, inv,
                           (const float*)dh0, ldz, Gm, ldgm, Am, ldam, dG, ldg, dAneg, lda);
    return (int)hipGetLastError();
}

DISPU_EXPORT int dispu_ps_prep_grad(long rows, int co, const float* xyz, const float* W0, const float* dG, long ldg, const float* dAneg,
                                    long lda, float* dxyz, float* dW0, void* stream) {
    if (rows < 0 || co != 128 || ((ldg | lda) & 1)) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    const int nb = (int)((rows + 3) / 4 < 256 ? (rows + 3) / 4 : 256);
    hipLaunchKernelGGL(ps_prep_grad_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, rows, xyz, W0, dG, ldg, dAneg, lda, dxyz, dW0);
    return (int)hipGetLastError();
}

DISPU_EXPORT int dispu_ps_skip_max_grad(long rows, int n_per_cloud, int k, int cf, const int* idx, const float* xyz, const float* feat,
                                        long ldf, const float* gmax, long ldm, const float* dgmax, long ldd, float* dxyz, float* dfeat,
                                        long lddf, int feat_is_relu, void* stream) {
    if (rows < 0 || k != 16 || cf != 128 || n_per_cloud <= 0 || (ldf & 3)) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    hipLaunchKernelGGL(ps_skip_max_grad_kernel, dim3((unsigned)((rows * 32 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rows, n_per_cloud,
                       idx, xyz, feat, ldf, gmax, ldm, dgmax, ldd, dxyz, dfeat, lddf, feat_is_relu);
    return (int)hipGetLastError();
}

DISPU_EXPORT int dispu_ps_point_matmul_grad_relu_s(long rows, int k, int c, int t_n, const void* X2, long ldx2, const float* wv,
                                                   const void* dout, long ldo, void* dX2, long lddx2, float* dwv, int bf16_storage,
                                                   void* stream);

DISPU_EXPORT int dispu_ps_point_matmul_grad_relu(long rows, int k, int c, int t_n, const float* X2, long ldx2, const float* wv,
                                                 const float* dout, long ldo, float* dX2, long lddx2, float* dwv, void* stream) {
    return dispu_ps_point_matmul_grad_relu_s(rows, k, c, t_n, X2, ldx2, wv, dout, ldo, dX2, lddx2, dwv, 0, stream);
}

// the same with X2, dout and dX2 STORED as bf16 when bf16_storage != 0 (wv / dwv stay fp32; strides in elements)
DISPU_EXPORT int dispu_ps_point_matmul_grad_relu_s(long rows, int k, int c, int t_n, const void* X2, long ldx2, const float* wv,
                                                   const void* dout, long ldo, void* dX2, long lddx2, float* dwv, int bf16_storage,
                                                   void* stream) {
    if (rows < 0 || k != 16 || c != 128 || t_n != 16) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    const int nb = (int)(rows < 16384 ? rows : 16384);
    if (bf16_storage)
        hipLaunchKernelGGL(ps_point_matmul_grad_relu_kernel<__bf16>, dim3(nb), dim3(256), 0, (hipStream_t)stream, rows, (const __bf16*)X2, ldx2, wv,
                           (const __bf16*)dout, ldo, (__bf16*)dX2, lddx2, dwv);
    else
        hipLaunchKernelGGL(ps_point_matmul_grad_relu_kernel<float>, dim3(nb), dim3(256), 0, (hipStream_t)stream, rows, (const float*)X2, ldx2, wv,
                           (const float*)dout, ldo, (float*)dX2, lddx2, dwv);
    return (int)hipGetLastError();
}

// dispu_ps_gather_sub_relu writing bf16 (bf16 activation storage of the training step)
DISPU_EXPORT int dispu_ps_gather_sub_relu_bf16(long rows, int n_per_cloud, int k, int c, const int* idx, const float* G, long ldg,
                                               const float* A, long lda, void* X1, long ldx1, void* stream) {
    if (rows < 0 || c <= 0 || (c & 3) || ((ldg | lda | ldx1) & 3)) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    const size_t total = (size_t)rows * k * (c / 4);
    hipLaunchKernelGGL(ps_gather_sub_relu_bf16_kernel, dim3(tf_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, rows, n_per_cloud, k, c / 4,
                       idx, G, ldg, A, lda, (__bf16*)X1, ldx1);
    return (int)hipGetLastError();
}

// Chamfer term of pu_loss: value[0] = CD(gt, pred) (loss_utils.py:45-64, un-scaled) from nn_distance's outputs, and
// dpred = d(coef * CD)/d pred (zero-filled here, then accumulated).  radius [b]: the per-cloud normaliser.
DISPU_EXPORT int dispu_chamfer_loss_grad(int b, int n_gt, const float* gt, int n_pred, const float* pred, const float* d_gt, const int* i_gt,
                                         const float* d_pred, const int* i_pred, const float* radius, float coef, float* value, float* dpred,
                                         void* stream) {
    if (b <= 0 || n_gt <= 0 || n_pred <= 0 || !gt || !pred || !d_gt || !i_gt || !d_pred || !i_pred || !radius || !value || !dpred)
        return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(chamfer_value_kernel, dim3(1), dim3(1024), 0, s, b, n_gt, n_pred, d_gt, d_pred, radius, value);
    DISPU_CHECK_LAUNCH();
    DISPU_TRY(hipMemsetAsync(dpred, 0, sizeof(float) * (size_t)b * n_pred * 3, s));
    const int mx = n_gt > n_pred ? n_gt : n_pred;
    hipLaunchKernelGGL(chamfer_grad_kernel, dim3((mx + 255) / 256, b, 2), dim3(256), 0, s, b, n_gt, n_pred, gt, pred, i_gt, i_pred, radius, coef,
                       dpred);
    return (int)hipGetLastError();
}

// out[5] = 1000 CD_coarse | 1000 CD_fine | repulsion term | pu_loss | weight_fine  (model.py:75-87); cd[2] = the two un-scaled
// Chamfer values, rep [nrep] = per-point hinge sums of dispu_repulsion (NULL: no repulsion term).
DISPU_EXPORT int dispu_pu_loss_finalize(const float* cd, const float* rep, long nrep, float wf, float rep_w, float* out, void* stream) {
    if (!cd || !out || (rep && nrep <= 0)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(pu_loss_finalize_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, cd, rep, nrep, wf, rep_w, out);
    return (int)hipGetLastError();
}

// get_repulsion_loss value + gradient (dispu_repulsion + dispu_repulsion_grad in one launch): out [rows], dpred accumulates.
DISPU_EXPORT int dispu_repulsion_loss_grad(long rows, int n_per_cloud, int ns, float h, float scale, const float* pred, const int* idx,
                                           float* out, float* dpred, void* stream) {
    if (rows < 0 || n_per_cloud <= 0 || ns != 20 || !pred || !idx || !out || !dpred) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    hipLaunchKernelGGL(repulsion_loss_grad_kernel<20>, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, (hipStream_t)stream, rows, n_per_cloud, h,
                       scale, pred, idx, out, dpred);
    return (int)hipGetLastError();
}

// dst[off .. off + K*N) = transpose of src[off ..] viewed as [K][N], for every {off, K, N} triple of desc [count][3] (device ints).
DISPU_EXPORT int dispu_transpose_batched(int count, const int* desc, const float* src, float* dst, void* stream) {
    if (count < 0 || !desc || !src || !dst) return (int)hipErrorInvalidValue;
    if (count == 0) return 0;
    hipLaunchKernelGGL(transpose_batched_kernel, dim3(64, count), dim3(256), 0, (hipStream_t)stream, desc, src, dst);
    return (int)hipGetLastError();
}
