// Backward (and training-mode forward) kernels of the generator's non-GEMM ops for gfx950.
//
// The reference never writes these: TF1 autodiff derives them from the graph of DisPU/generator.py:31-88
// (tf.reduce_max -> _MinOrMaxGrad, gather_nd -> scatter_nd, tf.matmul, tf.nn.softmax, contrib batch_norm, sigmoid,
// top_k) and DisPU/model.py:75-87,178 (loss, Adam).  Each kernel cites the forward op whose gradient it is.
// All are HBM-bound streaming / scatter kernels: lanes run along the channel axis, scatters use hardware float
// atomics (as the reference's own GroupPointGrad / scatteraddpoint do, tf_grouping_g.cu:61-78).
#include "common.h"

namespace dispu {

static inline int tgrid(size_t total, int block) {
    size_t g = (total + block - 1) / block;
    return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g));
}

// ---- tf.reduce_max over the neighbour axis (ops.py:1915 dense_conv, :1049 skip) --------------------------------
__global__ void max_k_kernel(long rows, int ns, int c, const float* __restrict__ X, long ldx, float* __restrict__ out, long ldo) {
    const size_t total = (size_t)rows * c;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t i = e / c;
        const int ch = (int)(e - i * c);
        const float* x = X + i * ns * ldx + ch;
        float m = x[0];
        for (int s = 1; s < ns; ++s) m = fmaxf(m, x[(size_t)s * ldx]);
        out[i * ldo + ch] = m;
    }
}

// math_grad._MinOrMaxGrad: the gradient is shared evenly by the entries equal to the maximum.
__global__ void max_k_grad_kernel(long rows, int ns, int c, const float* __restrict__ X, long ldx, const float* __restrict__ Y,
                                  long ldy, const float* __restrict__ dY, long lddy, float* __restrict__ dX, long lddx,
                                  int accumulate) {
    const size_t total = (size_t)rows * c;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t i = e / c;
        const int ch = (int)(e - i * c);
        const float* x = X + i * ns * ldx + ch;
        float* dx = dX + i * ns * lddx + ch;
        const float y = Y[i * ldy + ch], g = dY[i * lddy + ch];
        int cnt = 0;
        for (int s = 0; s < ns; ++s) cnt += (x[(size_t)s * ldx] == y) ? 1 : 0;
        const float share = g / (float)cnt;
        for (int s = 0; s < ns; ++s) {
            const float v = (x[(size_t)s * ldx] == y) ? share : 0.f;
            if (accumulate) dx[(size_t)s * lddx] += v;
            else dx[(size_t)s * lddx] = v;
        }
    }
}

// max_k_grad into columns [0, c) plus zeros into the `tail` columns behind them: the gradient buffer of a dense block's edge
// tensor is [l2 | l1 | l0 | centre | neighbour - centre]; the pooled part gets the max gradient, the rest only accumulates
// afterwards and used to be cleared by a separate fill of the whole buffer (22 MB, once per block and step)
__global__ void max_k_grad_tail_kernel(long rows, int ns, int c, int tail, const float* __restrict__ X, long ldx, const float* __restrict__ Y,
                                       long ldy, const float* __restrict__ dY, long lddy, float* __restrict__ dX, long lddx) {
    const unsigned w = (unsigned)(c + tail);
    const size_t total = (size_t)rows * w;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t i = e / w;
        const int ch = (int)(e - i * w);
        float* dx = dX + i * ns * lddx + ch;
        if (ch >= c) {
            for (int s = 0; s < ns; ++s) dx[(size_t)s * lddx] = 0.f;
            continue;
        }
        const float* x = X + i * ns * ldx + ch;
        const float y = Y[i * ldy + ch], g = dY[i * lddy + ch];
        int cnt = 0;
        for (int s = 0; s < ns; ++s) cnt += (x[(size_t)s * ldx] == y) ? 1 : 0;
        const float share = g / (float)cnt;
        for (int s = 0; s < ns; ++s) dx[(size_t)s * lddx] = (x[(size_t)s * ldx] == y) ? share : 0.f;
    }
}

// ---- get_edge_feature (ops.py:1856-1877): E[(i,s)] = [F_i | F_j - F_i] --------------------------------------------
// dF_i += sum_s (dE[(i,s), ch] - dE[(i,s), c + ch]);  dF_j += dE[(i,s), c + ch].
__global__ void edge_feature_grad_kernel(long rows, int n_per_cloud, int k, int c, const float* __restrict__ dE, long lde,
                                         const int* __restrict__ idx, int ldi, int ioff, float* __restrict__ dF, long lddf) {
    const size_t total = (size_t)rows * c;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t i = e / c;
        const int ch = (int)(e - i * c);
        const size_t base = (i / n_per_cloud) * n_per_cloud;
        float self = 0.f;
        for (int s = 0; s < k; ++s) {
            const float* g = dE + (i * k + s) * lde;
            const float gc = g[ch], gd = g[c + ch];
            self += gc - gd;
            const size_t j = base + idx[i * ldi + ioff + s];
            unsafeAtomicAdd(dF + j * lddf + ch, gd);
        }
        unsafeAtomicAdd(dF + i * lddf + ch, self);
    }
}

// ---- duplicate_up (ops.py:1152-1199): tile x up, copy-major ------------------------------------------------------
__global__ void dup_sum_grad_kernel(int nclouds, int n, int co, int up, const float* __restrict__ dZ, long lddz,
                                    float* __restrict__ dH, long lddh) {
    const size_t total = (size_t)nclouds * n * co;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t src = e / co;
        const int ch = (int)(e - src * co);
        const size_t cloud = src / n, i = src - cloud * n;
        float s = 0.f;
        for (int r = 0; r < up; ++r) s += dZ[((cloud * up + r) * n + i) * lddz + ch];
        dH[src * lddh + ch] = s;
    }
}

// ---- PointShuffle2 grouping (ops.py:1030-1037, grouping :154-179): gf[(i,s)] = [xyz_j - xyz_i | xyz_j | feat_j] ---
__global__ void ps_group_kernel(long rows, int n_per_cloud, int k, int cf, const int* __restrict__ idx,
                                const float* __restrict__ xyz, const float* __restrict__ feat, long ldf,
                                float* __restrict__ gf, long ldg) {
    const int w = 6 + cf;
    const size_t total = (size_t)rows * k * w;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t pair = e / w;
        const int ch = (int)(e - pair * w);
        const size_t i = pair / k;
        const size_t j = (i / n_per_cloud) * n_per_cloud + idx[pair];
        float v;
        if (ch < 3) v = xyz[j * 3 + ch] - xyz[i * 3 + ch];
        else if (ch < 6) v = xyz[j * 3 + ch - 3];
        else v = feat[j * ldf + ch - 6];
        gf[pair * ldg + ch] = v;
    }
}

__global__ void ps_group_grad_kernel(long rows, int n_per_cloud, int k, int cf, const int* __restrict__ idx,
                                     const float* __restrict__ dgf, long ldg, float* __restrict__ dxyz,
                                     float* __restrict__ dfeat, long lddf) {
    const int w = 6 + cf;
    const size_t total = (size_t)rows * k * w;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t pair = e / w;
        const int ch = (int)(e - pair * w);
        const size_t i = pair / k;
        const size_t j = (i / n_per_cloud) * n_per_cloud + idx[pair];
        const float g = dgf[pair * ldg + ch];
        if (ch < 3) {
            unsafeAtomicAdd(dxyz + j * 3 + ch, g);
            unsafeAtomicAdd(dxyz + i * 3 + ch, -g);
        } else if (ch < 6) {
            unsafeAtomicAdd(dxyz + j * 3 + ch - 3, g);
        } else {
            unsafeAtomicAdd(dfeat + j * lddf + ch - 6, g);
        }
    }
}

// ---- per-point feature x weight product (ops.py:1063-1064): out[i, c*T + t] = sum_s X2[(i,s), c] * wv[(i,s), t] ----
// dX2[(i,s), c] = sum_t dout[i, c*T + t] * wv[(i,s), t];   dwv[(i,s), t] = sum_c X2[(i,s), c] * dout[i, c*T + t].
// one point per workgroup; k == T == 16, c == 128.
__global__ __launch_bounds__(256) void ps_point_matmul_grad_kernel(long rows, const float* __restrict__ X2, long ldx2,
                                                                    const float* __restrict__ wv, const float* __restrict__ dout,
                                                                    long ldo, float* __restrict__ dX2, long lddx2,
                                                                    float* __restrict__ dwv) {
    constexpr int K = 16, T = 16, C = 128;
    __shared__ float s_do[C * T];
    __shared__ float s_x[K][C + 1];
    __shared__ float s_w[K][T + 1];
    for (long i = blockIdx.x; i < rows; i += gridDim.x) {
        for (int e = threadIdx.x; e < C * T; e += 256) s_do[e] = dout[i * ldo + e];
        for (int e = threadIdx.x; e < K * C; e += 256) s_x[e / C][e % C] = X2[(i * K + e / C) * ldx2 + e % C];
        s_w[threadIdx.x >> 4][threadIdx.x & 15] = wv[(i * K) * T + threadIdx.x];
        __syncthreads();
        for (int e = threadIdx.x; e < K * C; e += 256) {
            const int s = e / C, c = e - s * C;
            float a = 0.f;
#pragma unroll
            for (int t = 0; t < T; ++t) a = __builtin_fmaf(s_do[c * T + t], s_w[s][t], a);
            dX2[(i * K + s) * lddx2 + c] = a;
        }
        {
            const int s = threadIdx.x >> 4, t = threadIdx.x & 15;
            float a = 0.f;
            for (int c = 0; c < C; ++c) a = __builtin_fmaf(s_x[s][c], s_do[c * T + t], a);
            dwv[(i * K + s) * T + t] = a;
        }
        __syncthreads();
    }
}

// ---- tf.nn.softmax gradient (ops.py:338): dS = mul * P * (dP - sum_j dP_j P_j), in place over dP; one wave per row --
__global__ __launch_bounds__(256) void softmax_rows_grad_kernel(long rows, int n, float mul, const float* __restrict__ P,
                                                                 long ldp, float* __restrict__ dP, long lddp) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (long r = (long)blockIdx.x * 4 + wave; r < rows; r += (long)gridDim.x * 4) {
        const float* p = P + r * ldp;
        float* d = dP + r * lddp;
        float s = 0.f;
        for (int j = lane; j < n; j += 64) s = __builtin_fmaf(p[j], d[j], s);
        s = wave_sum_f32(s);
        for (int j = lane; j < n; j += 64) d[j] = mul * (p[j] * (d[j] - s));
    }
}

// ---- contrib.layers.batch_norm, training mode (tf_util.py:512-531; weight_net_hidden ops.py:181-191) -------------
// column statistics of x[rows, c] (c <= 64, 256 % c == 0): thread t owns column t % c, rows t / c, t / c + 256 / c, ...
// part[blk][0][c] = sum x, part[blk][1][c] = sum x^2 (double)
__global__ __launch_bounds__(256) void bn_stats_kernel(long rows, int c, int rows_per_block, const float* __restrict__ X, long ldx,
                                                        double* __restrict__ part) {
    __shared__ double red[2][256];
    const int ch = threadIdx.x % c, rr = threadIdx.x / c, step = 256 / c;
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    double s1 = 0.0, s2 = 0.0;
    if (rr < step)
        for (long r = r0 + rr; r < r1; r += step) {
            const double v = X[r * ldx + ch];
            s1 += v;
            s2 += v * v;
        }
    red[0][threadIdx.x] = s1;
    red[1][threadIdx.x] = s2;
    __syncthreads();
    if (threadIdx.x < c) {
        double a = 0.0, b = 0.0;
        for (int g = 0; g < step; ++g) {
            a += red[0][g * c + threadIdx.x];
            b += red[1][g * c + threadIdx.x];
        }
        part[((size_t)blockIdx.x * 2 + 0) * c + threadIdx.x] = a;
        part[((size_t)blockIdx.x * 2 + 1) * c + threadIdx.x] = b;
    }
}

// stats[0:c] = mean, [c:2c] = biased variance, [2c:3c] = 1/sqrt(var + eps); moving statistics updated in place
// (decay * moving + (1 - decay) * batch; the variance fed to the moving average is Bessel-corrected, fused BN).
// one WAVE per channel (blockIdx.x = channel): lanes add the partials s = lane, lane + 64, ... in double, a fixed butterfly
// combines them.  (One THREAD per channel walked all 1024 partials serially: 34 us for 16 channels.)
__device__ __forceinline__ double bn_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(64) void bn_finalize_kernel(long rows, int c, int nparts, const double* __restrict__ part, float eps, float decay,
                                   float* __restrict__ stats, float* __restrict__ moving_mean, float* __restrict__ moving_var) {
    const int ch = blockIdx.x;
    double a = 0.0, b = 0.0;
    for (int s = threadIdx.x; s < nparts; s += 64) {
        a += part[((size_t)s * 2 + 0) * c + ch];
        b += part[((size_t)s * 2 + 1) * c + ch];
    }
    a = bn_wave_sum(a);
    b = bn_wave_sum(b);
    if (threadIdx.x != 0) return;
    const double mean = a / (double)rows;
    double var = b / (double)rows - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[ch] = (float)mean;
    stats[c + ch] = (float)var;
    stats[2 * c + ch] = (float)(1.0 / sqrt(var + (double)eps));
    if (moving_mean) moving_mean[ch] = (float)((double)decay * moving_mean[ch] + (1.0 - (double)decay) * mean);
    if (moving_var) {
        const double unbiased = rows > 1 ? var * ((double)rows / (double)(rows - 1)) : var;
        moving_var[ch] = (float)((double)decay * moving_var[ch] + (1.0 - (double)decay) * unbiased);
    }
}

__global__ void bn_apply_kernel(long rows, int c, const float* __restrict__ X, long ldx, const float* __restrict__ stats,
                                const float* __restrict__ gamma, const float* __restrict__ beta, int act, float* __restrict__ Y,
                                long ldy) {
    const size_t total = (size_t)rows * c;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t r = e / c;
        const int ch = (int)(e - r * c);
        const float xh = (X[r * ldx + ch] - stats[ch]) * stats[2 * c + ch];
        float v = xh * gamma[ch] + beta[ch];
        if (act) v = fmaxf(v, 0.f);
        Y[r * ldy + ch] = v;
    }
}

// part[blk][0][c] = sum dz, part[blk][1][c] = sum dz * xhat, dz = dY * (act ? Y > 0 : 1)
__global__ __launch_bounds__(256) void bn_grad_stats_kernel(long rows, int c, int rows_per_block, const float* __restrict__ X,
                                                             long ldx, const float* __restrict__ Y, long ldy,
                                                             const float* __restrict__ dY, long lddy,
                                                             const float* __restrict__ stats, int act, double* __restrict__ part) {
    __shared__ double red[2][256];
    const int ch = threadIdx.x % c, rr = threadIdx.x / c, step = 256 / c;
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    double s1 = 0.0, s2 = 0.0;
    if (rr < step) {
        const float mu = stats[ch], is = stats[2 * c + ch];
        for (long r = r0 + rr; r < r1; r += step) {
            float g = dY[r * lddy + ch];
            if (act && !(Y[r * ldy + ch] > 0.f)) g = 0.f;
            const float xh = (X[r * ldx + ch] - mu) * is;
            s1 += (double)g;
            s2 += (double)g * (double)xh;
        }
    }
    red[0][threadIdx.x] = s1;
    red[1][threadIdx.x] = s2;
    __syncthreads();
    if (threadIdx.x < c) {
        double a = 0.0, b = 0.0;
        for (int g = 0; g < step; ++g) {
            a += red[0][g * c + threadIdx.x];
            b += red[1][g * c + threadIdx.x];
        }
        part[((size_t)blockIdx.x * 2 + 0) * c + threadIdx.x] = a;
        part[((size_t)blockIdx.x * 2 + 1) * c + threadIdx.x] = b;
    }
}

// sums[0:c] = sum dz, sums[c:2c] = sum dz*xhat;  dbeta += sum dz, dgamma += sum dz*xhat
__global__ __launch_bounds__(64) void bn_grad_finalize_kernel(int c, int nparts, const double* __restrict__ part, float* __restrict__ sums,
                                        float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int ch = blockIdx.x;
    double a = 0.0, b = 0.0;
    for (int s = threadIdx.x; s < nparts; s += 64) {
        a += part[((size_t)s * 2 + 0) * c + ch];
        b += part[((size_t)s * 2 + 1) * c + ch];
    }
    a = bn_wave_sum(a);
    b = bn_wave_sum(b);
    if (threadIdx.x != 0) return;
    sums[ch] = (float)a;
    sums[c + ch] = (float)b;
    if (dbeta) dbeta[ch] += (float)a;
    if (dgamma) dgamma[ch] += (float)b;
}

// dx = gamma * inv_std * (dz - sum_dz / M - xhat * sum_dzxhat / M)
__global__ void bn_grad_apply_kernel(long rows, int c, const float* __restrict__ X, long ldx, const float* __restrict__ Y, long ldy,
                                     const float* __restrict__ dY, long lddy, const float* __restrict__ stats,
                                     const float* __restrict__ gamma, const float* __restrict__ sums, int act,
                                     float* __restrict__ dX, long lddx) {
    const size_t total = (size_t)rows * c;
    const float inv_m = 1.0f / (float)rows;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t r = e / c;
        const int ch = (int)(e - r * c);
        float g = dY[r * lddy + ch];
        if (act && !(Y[r * ldy + ch] > 0.f)) g = 0.f;
        const float is = stats[2 * c + ch];
        const float xh = (X[r * ldx + ch] - stats[ch]) * is;
        dX[r * lddx + ch] = gamma[ch] * is * ((g - sums[ch] * inv_m) - xh * (sums[c + ch] * inv_m));
    }
}

// ---- coordinate_regressor is_off (ops.py:1106-1108; generator.py:80-81): fine = coarse + sigmoid(z) - 0.5 ---------
__global__ void sigmoid_offset_kernel(size_t total, const float* __restrict__ z, const float* __restrict__ base, float* __restrict__ out) {
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const float s = 1.0f / (1.0f + expf(-z[e]));
        out[e] = base[e] + (s - 0.5f);
    }
}

__global__ void sigmoid_offset_grad_kernel(size_t total, const float* __restrict__ z, const float* __restrict__ dout,
                                           float* __restrict__ dz, float* __restrict__ dbase) {
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const float s = 1.0f / (1.0f + expf(-z[e]));
        const float g = dout[e];
        dz[e] = g * (s * (1.0f - s));
        if (dbase) dbase[e] += g;
    }
}

// ---- get_repulsion_loss gradient (loss_utils.py:280-296) ------------------------------------------------------------
// per point i: the 2nd..5th smallest d_s = |p_j - p_i|^2 among its ns ball neighbours (same stable selection as the
// forward kernel); for each with h - d > 0:  dL/dp_j += -2 scale (p_j - p_i),  dL/dp_i += +2 scale (p_j - p_i).
__global__ void repulsion_grad_kernel(long rows, int n_per_cloud, int ns, float h, float scale, const float* __restrict__ pred,
                                      const int* __restrict__ idx, float* __restrict__ dpred) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += (long)gridDim.x * blockDim.x) {
        const long base = (i / n_per_cloud) * n_per_cloud;
        const float px = pred[i * 3], py = pred[i * 3 + 1], pz = pred[i * 3 + 2];
        float b[5] = {__builtin_inff(), __builtin_inff(), __builtin_inff(), __builtin_inff(), __builtin_inff()};
        int bs[5] = {-1, -1, -1, -1, -1};
        for (int s = 0; s < ns; ++s) {
            const long j = base + idx[i * ns + s];
            const float dx = pred[j * 3] - px, dy = pred[j * 3 + 1] - py, dz = pred[j * 3 + 2] - pz;
            const float d = (dx * dx + dy * dy) + dz * dz;
            if (d < b[4]) {
                b[4] = d;
                bs[4] = s;
#pragma unroll
                for (int t = 4; t > 0; --t)
                    if (b[t] < b[t - 1]) {
                        const float tmp = b[t]; b[t] = b[t - 1]; b[t - 1] = tmp;
                        const int ts = bs[t]; bs[t] = bs[t - 1]; bs[t - 1] = ts;
                    }
            }
        }
        float gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll
        for (int t = 1; t < 5; ++t) {
            if (bs[t] < 0 || !(h - b[t] > 0.f)) continue;
            const long j = base + idx[i * ns + bs[t]];
            const float dx = pred[j * 3] - px, dy = pred[j * 3 + 1] - py, dz = pred[j * 3 + 2] - pz;
            const float cx = 2.f * scale * dx, cy = 2.f * scale * dy, cz = 2.f * scale * dz;
            unsafeAtomicAdd(dpred + j * 3 + 0, -cx);
            unsafeAtomicAdd(dpred + j * 3 + 1, -cy);
            unsafeAtomicAdd(dpred + j * 3 + 2, -cz);
            gx += cx; gy += cy; gz += cz;
        }
        unsafeAtomicAdd(dpred + i * 3 + 0, gx);
        unsafeAtomicAdd(dpred + i * 3 + 1, gy);
        unsafeAtomicAdd(dpred + i * 3 + 2, gz);
    }
}

// ---- small helpers -----------------------------------------------------------------------------------------------
__global__ void add3_kernel(size_t total, const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                            float* __restrict__ out) {
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x)
        out[e] = (a[e] + b[e]) + c[e];
}

// out[b, j] = val[b] (per-cloud constant rows: d loss / d dist of the Chamfer mean, loss_utils.py:59-63)
__global__ void fill_rows_kernel(int b, int n, const float* __restrict__ val, float mul, float* __restrict__ out) {
    const size_t total = (size_t)b * n;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x)
        out[e] = val[e / n] * mul;
}

// tf.train.AdamOptimizer (model.py:178): m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= lr_t m / (sqrt(v) + eps)
__global__ void adam_kernel(size_t total, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, float lr_t, float beta1, float beta2, float eps, float gscale) {
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const float gg = g[e] * gscale;
        const float mm = beta1 * m[e] + (1.0f - beta1) * gg;
        const float vv = beta2 * v[e] + (1.0f - beta2) * (gg * gg);
        m[e] = mm;
        v[e] = vv;
        p[e] = p[e] - lr_t * mm / (sqrtf(vv) + eps);
    }
}

static int bn_blocks(long rows, int& rows_per_block) {
    rows_per_block = 1024;
    long nb = (rows + rows_per_block - 1) / rows_per_block;
    if (nb > 1024) {
        rows_per_block = (int)((rows + 1023) / 1024);
        nb = (rows + rows_per_block - 1) / rows_per_block;
    }
    return (int)nb;
}

// Y[r][c] = act((((P_0 + P_1) + ...) + P_{np-1})[r][c] + bias[c]) on float4 quads: the fixed-order sum of the K-chunk partials of a
// split-K product (Trainer.forward: after_conv at 8 - 16 patches, see dispu_linear_splitk_finish in include/dispu_hip.h)
__global__ void splitk_finish_kernel(unsigned quads, unsigned n4, int np, const float* __restrict__ part, size_t stride, const float* __restrict__ bias,
                                     int act, float* __restrict__ Y, unsigned ldy) {
    for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < quads; e += gridDim.x * blockDim.x) {
        const unsigned r = e / n4, c = (e - r * n4) * 4;
        const float* p = part + (size_t)e * 4;
        float4 v[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) v[s] = (s < np) ? *reinterpret_cast<const float4*>(p + (size_t)s * stride) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 t = v[0];
#pragma unroll
        for (int s = 1; s < 8; ++s)
            if (s < np) { t.x += v[s].x; t.y += v[s].y; t.z += v[s].z; t.w += v[s].w; }
        if (bias) { const float4 b = *reinterpret_cast<const float4*>(bias + c); t.x += b.x; t.y += b.y; t.z += b.z; t.w += b.w; }
        if (act) { t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f); }
        *reinterpret_cast<float4*>(Y + (size_t)r * ldy + c) = t;
    }
}

}  // namespace dispu

using namespace dispu;

#define LAUNCH1D(kernel, total, ...)                                                                                  \
    do {                                                                                                              \
        hipLaunchKernelGGL(kernel, dim3(tgrid((size_t)(total), 256)), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); \
        return (int)hipGetLastError();                                                                                \
    } while (0)

DISPU_EXPORT int dispu_max_k(long rows, int ns, int c, const float* X, long ldx, float* out, long ldo, void* stream) {
    if (rows < 0 || ns <= 0 || c <= 0) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    LAUNCH1D(max_k_kernel, (size_t)rows * c, rows, ns, c, X, ldx, out, ldo);
}

DISPU_EXPORT int dispu_max_k_grad(long rows, int ns, int c, const float* X, long ldx, const float* Y, long ldy, const float* dY,
                                  long lddy, float* dX, long lddx, int accumulate, void* stream) {
    if (rows < 0 || ns <= 0 || c <= 0) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    LAUNCH1D(max_k_grad_kernel, (size_t)rows * c, rows, ns, c, X, ldx, Y, ldy, dY, lddy, dX, lddx, accumulate);
}

DISPU_EXPORT int dispu_max_k_grad_tail(long rows, int ns, int c, int tail, const float* X, long ldx, const float* Y, long ldy,
                                       const float* dY, long lddy, float* dX, long lddx, void* stream) {
    if (rows < 0 || ns <= 0 || c <= 0 || tail < 0) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    LAUNCH1D(max_k_grad_tail_kernel, (size_t)rows * (c + tail), rows, ns, c, tail, X, ldx, Y, ldy, dY, lddy, dX, lddx);
}

DISPU_EXPORT int dispu_edge_feature_grad(long rows, int n_per_cloud, int k, int c, const float* dE, long lde, const int* idx,
                                         int ldi, int ioff, float* dF, long lddf, void* stream) {
    if (rows < 0 || n_per_cloud <= 0 || k <= 0 || c <= 0) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    LAUNCH1D(edge_feature_grad_kernel, (size_t)rows * c, rows, n_per_cloud, k, c, dE, lde, idx, ldi, ioff, dF, lddf);
}

DISPU_EXPORT int dispu_dup_sum_grad(int nclouds, int n, int co, int up, const float* dZ, long lddz, float* dH, long lddh,
                                    void* stream) {
    if (nclouds < 0 || n <= 0 || co <= 0 || up <= 0) return (int)hipErrorInvalidValue;
    if (nclouds == 0) return 0;
    LAUNCH1D(dup_sum_grad_kernel, (size_t)nclouds * n * co, nclouds, n, co, up, dZ, lddz, dH, lddh);
}

DISPU_EXPORT int dispu_ps_group(long rows, int n_per_cloud, int k, int cf, const int* idx, const float* xyz, const float* feat,
                                long ldf, float* gf, long ldg, void* stream) {
    if (rows < 0 || n_per_cloud <= 0 || k <= 0 || cf < 0) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    LAUNCH1D(ps_group_kernel, (size_t)rows * k * (6 + cf), rows, n_per_cloud, k, cf, idx, xyz, feat, ldf, gf, ldg);
}

DISPU_EXPORT int dispu_ps_group_grad(long rows, int n_per_cloud, int k, int cf, const int* idx, const float* dgf, long ldg,
                                     float* dxyz, float* dfeat, long lddf, void* stream) {
    if (rows < 0 || n_per_cloud <= 0 || k <= 0 || cf < 0) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    LAUNCH1D(ps_group_grad_kernel, (size_t)rows * k * (6 + cf), rows, n_per_cloud, k, cf, idx, dgf, ldg, dxyz, dfeat, lddf);
}

DISPU_EXPORT int dispu_ps_point_matmul_grad(long rows, int k, int c, int t_n, const float* X2, long ldx2, const float* wv,
                                            const float* dout, long ldo, float* dX2, long lddx2, float* dwv, void* stream) {
    if (rows < 0 || k != 16 || c != 128 || t_n != 16) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    hipLaunchKernelGGL(ps_point_matmul_grad_kernel, dim3((unsigned)(rows > 8192 ? 8192 : rows)), dim3(256), 0, (hipStream_t)stream,
                       rows, X2, ldx2, wv, dout, ldo, dX2, lddx2, dwv);
    return (int)hipGetLastError();
}

DISPU_EXPORT int dispu_softmax_rows_grad(long rows, int n, float mul, const float* P, long ldp, float* dP, long lddp, void* stream) {
    if (rows < 0 || n <= 0) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    const long blocks = (rows + 3) / 4;
    hipLaunchKernelGGL(softmax_rows_grad_kernel, dim3((unsigned)(blocks > 16384 ? 16384 : blocks)), dim3(256), 0,
                       (hipStream_t)stream, rows, n, mul, P, ldp, dP, lddp);
    return (int)hipGetLastError();
}

DISPU_EXPORT long dispu_bn_scratch_bytes(long rows, int c) {
    if (rows <= 0 || c <= 0) return 0;
    int rpb;
    return (long)bn_blocks(rows, rpb) * 2 * c * (long)sizeof(double);
}

DISPU_EXPORT int dispu_bn_train(long rows, int c, const float* X, long ldx, const float* gamma, const float* beta, float eps,
                                float decay, int act, float* Y, long ldy, float* stats, float* moving_mean, float* moving_var,
                                void* scratch, long scratch_bytes, void* stream) {
    if (rows <= 0 || c <= 0 || c > 64 || 256 % c != 0) return (int)hipErrorInvalidValue;
    int rpb;
    const int nb = bn_blocks(rows, rpb);
    if (scratch == nullptr || scratch_bytes < (long)nb * 2 * c * (long)sizeof(double)) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_stats_kernel, dim3(nb), dim3(256), 0, s, rows, c, rpb, X, ldx, (double*)scratch);
    DISPU_CHECK_LAUNCH();
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(c), dim3(64), 0, s, rows, c, nb, (const double*)scratch, eps, decay, stats,
                       moving_mean, moving_var);
    DISPU_CHECK_LAUNCH();
    hipLaunchKernelGGL(bn_apply_kernel, dim3(tgrid((size_t)rows * c, 256)), dim3(256), 0, s, rows, c, X, ldx, stats, gamma, beta,
                       act, Y, ldy);
    return (int)hipGetLastError();
}

DISPU_EXPORT int dispu_bn_train_grad(long rows, int c, const float* X, long ldx, const float* Y, long ldy, const float* dY,
                                     long lddy, const float* stats, const float* gamma, int act, float* dX, long lddx,
                                     float* dgamma, float* dbeta, float* sums, void* scratch, long scratch_bytes, void* stream) {
    if (rows <= 0 || c <= 0 || c > 64 || 256 % c != 0 || sums == nullptr) return (int)hipErrorInvalidValue;
    int rpb;
    const int nb = bn_blocks(rows, rpb);
    if (scratch == nullptr || scratch_bytes < (long)nb * 2 * c * (long)sizeof(double)) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_grad_stats_kernel, dim3(nb), dim3(256), 0, s, rows, c, rpb, X, ldx, Y, ldy, dY, lddy, stats, act,
                       (double*)scratch);
    DISPU_CHECK_LAUNCH();
    hipLaunchKernelGGL(bn_grad_finalize_kernel, dim3(c), dim3(64), 0, s, c, nb, (const double*)scratch, sums, dgamma, dbeta);
    DISPU_CHECK_LAUNCH();
    hipLaunchKernelGGL(bn_grad_apply_kernel, dim3(tgrid((size_t)rows * c, 256)), dim3(256), 0, s, rows, c, X, ldx, Y, ldy, dY,
                       lddy, stats, gamma, sums, act, dX, lddx);
    return (int)hipGetLastError();
}

DISPU_EXPORT int dispu_sigmoid_offset(long total, const float* z, const float* base, float* out, void* stream) {
    if (total < 0) return (int)hipErrorInvalidValue;
    if (total == 0) return 0;
    LAUNCH1D(sigmoid_offset_kernel, total, (size_t)total, z, base, out);
}

DISPU_EXPORT int dispu_sigmoid_offset_grad(long total, const float* z, const float* dout, float* dz, float* dbase, void* stream) {
    if (total < 0) return (int)hipErrorInvalidValue;
    if (total == 0) return 0;
    LAUNCH1D(sigmoid_offset_grad_kernel, total, (size_t)total, z, dout, dz, dbase);
}

DISPU_EXPORT int dispu_repulsion_grad(long rows, int n_per_cloud, int ns, float h, float scale, const float* pred, const int* idx,
                                      float* dpred, void* stream) {
    if (rows < 0 || n_per_cloud <= 0 || ns < 5) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    LAUNCH1D(repulsion_grad_kernel, rows, rows, n_per_cloud, ns, h, scale, pred, idx, dpred);
}

DISPU_EXPORT int dispu_linear_splitk_finish(long rows, int n, int nparts, const float* part, long part_stride, const float* bias, int act,
                                            float* Y, long ldy, void* stream) {
    if (rows < 0 || n <= 0 || (n & 3) || nparts < 1 || nparts > 8 || !part || !Y || (ldy & 3) || (part_stride & 3) ||
        ((((uintptr_t)part) | ((uintptr_t)Y) | ((uintptr_t)bias)) & 15) || (long)rows * n / 4 >= 0x7fffffffl)
        return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    const unsigned quads = (unsigned)((long)rows * n / 4);
    const unsigned g = (quads + 255) / 256;
    hipLaunchKernelGGL(splitk_finish_kernel, dim3(g > 16384 ? 16384 : g), dim3(256), 0, (hipStream_t)stream, quads, (unsigned)(n / 4), nparts, part,
                       (size_t)part_stride, bias, act, Y, (unsigned)ldy);
    return (int)hipGetLastError();
}

DISPU_EXPORT int dispu_add3(long total, const float* a, const float* b, const float* c, float* out, void* stream) {
    if (total < 0) return (int)hipErrorInvalidValue;
    if (total == 0) return 0;
    LAUNCH1D(add3_kernel, total, (size_t)total, a, b, c, out);
}

DISPU_EXPORT int dispu_fill_rows(int b, int n, const float* val, float mul, float* out, void* stream) {
    if (b < 0 || n < 0) return (int)hipErrorInvalidValue;
    if (b == 0 || n == 0) return 0;
    LAUNCH1D(fill_rows_kernel, (size_t)b * n, b, n, val, mul, out);
}

DISPU_EXPORT int dispu_adam(long total, float* p, const float* g, float* m, float* v, float lr_t, float beta1, float beta2,
                            float eps, float gscale, void* stream) {
    if (total < 0) return (int)hipErrorInvalidValue;
    if (total == 0) return 0;
    LAUNCH1D(adam_kernel, total, (size_t)total, p, g, m, v, lr_t, beta1, beta2, eps, gscale);
}
