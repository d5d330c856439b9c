// Gather-fused pieces of the Dis-PU generator that are not plain GEMMs (gfx950).  Each kernel
// cites the reference block it implements; the dense layers themselves are in linear.hip.
// Arithmetic follows oracle/generator.py: per-output fmaf chains in ascending input-channel
// order, one rounded bias add, then the activation.
#include "common.h"

namespace dispu {

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float identity, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(identity), __float_as_int(v), CTRL, 0xF, 0xF, false));
}
// max over the 16 lanes of a DPP row; valid in the row's lane 15
__device__ __forceinline__ float row16_max_to_lane15(float v) {
    const float ninf = -__builtin_inff();
    v = fmaxf(v, dpp_f32<DPP_ROW_SHR1>(ninf, v));
    v = fmaxf(v, dpp_f32<DPP_ROW_SHR2>(ninf, v));
    v = fmaxf(v, dpp_f32<DPP_ROW_SHR4>(ninf, v));
    v = fmaxf(v, dpp_f32<DPP_ROW_SHR8>(ninf, v));
    return v;
}

// ---------------------------------------------------------------------------------------------
// Y[r, 0:N] = act( (chain_k X[r,k] W[k,n] + bias[n]) * scale[n] + shift[n] ),  K <= 4, N <= 32.
// feature_extraction layer0 (3 -> 24, no activation; Common/ops.py:1449-1451).
template <int N>
__global__ void linear_small_k_kernel(long rows, int K, const float* __restrict__ X, long ldx,
                                      const float* __restrict__ W, const float* __restrict__ bias, int act,
                                      float* __restrict__ Y, long ldy) {
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    float x[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K; ++k) x[k] = X[r * ldx + k];
#pragma unroll
    for (int o = 0; o < N; ++o) {
        float acc = 0.f;
        for (int k = 0; k < K; ++k) acc = __builtin_fmaf(x[k], W[k * N + o], acc);
        if (bias) acc = acc + bias[o];
        if (act == 1) acc = fmaxf(acc, 0.f);
        Y[r * ldy + o] = acc;
    }
}

// Same arithmetic, one thread per (row, 4 outputs): the row-per-thread form above runs 8192 rows on 32 workgroups with
// 1920-byte-strided scalar stores (9 us for 0.8 MB); here a wave writes whole 96-byte row segments as float4.
template <int N>
__global__ __launch_bounds__(256) void linear_small_k_v4_kernel(long rows, int K, const float* __restrict__ X, long ldx,
                                                                 const float* __restrict__ W, const float* __restrict__ bias, int act,
                                                                 float* __restrict__ Y, long ldy) {
    constexpr int Q = N / 4;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long r = e / Q;
    const int q = (int)(e - r * Q);
    if (r >= rows) return;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < K; ++k) {
        const float x = X[r * ldx + k];
        const float4 w = *reinterpret_cast<const float4*>(W + k * N + q * 4);
        acc.x = __builtin_fmaf(x, w.x, acc.x); acc.y = __builtin_fmaf(x, w.y, acc.y);
        acc.z = __builtin_fmaf(x, w.z, acc.z); acc.w = __builtin_fmaf(x, w.w, acc.w);
    }
    if (bias) {
        const float4 b = *reinterpret_cast<const float4*>(bias + q * 4);
        acc.x = acc.x + b.x; acc.y = acc.y + b.y; acc.z = acc.z + b.z; acc.w = acc.w + b.w;
    }
    if (act == 1) { acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f); }
    *reinterpret_cast<float4*>(Y + r * ldy + q * 4) = acc;
}

// Y[r, 0:N] = chain_k X[r,k] W[k,n] + bias[n],  N <= 4 (coordinate_regressor fc_layer2, ops.py:1101-1104).
// mode 1: Y = R + (sigmoid(.) - 0.5)  -- the fine branch's offset (ops.py:1106-1108) fused with
// `fine = coarse + offset` (DisPU/generator.py:80-81).
template <int N>
__global__ void linear_small_n_kernel(long rows, int K, const float* __restrict__ X, long ldx,
                                      const float* __restrict__ W, const float* __restrict__ bias, int mode,
                                      const float* __restrict__ R, long ldr, float* __restrict__ Y, long ldy) {
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    float acc[N];
#pragma unroll
    for (int o = 0; o < N; ++o) acc[o] = 0.f;
    const float* xr = X + r * ldx;
    for (int k = 0; k < K; ++k) {
        const float xv = xr[k];
#pragma unroll
        for (int o = 0; o < N; ++o) acc[o] = __builtin_fmaf(xv, W[k * N + o], acc[o]);
    }
#pragma unroll
    for (int o = 0; o < N; ++o) {
        float v = acc[o];
        if (bias) v = v + bias[o];
        if (mode == 1) v = R[r * ldr + o] + (1.0f / (1.0f + expf(-v)) - 0.5f);
        Y[r * ldy + o] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// dense_conv (Common/ops.py:1897-1915) + get_edge_feature (:1856-1877), fused: for every point p and
// each of its 16 feature-space neighbours j
//   y0 = [F_p, F_j - F_p]           -> l0 = relu(y0.W0 + b0)
//   y1 = [l0, F_p]                  -> l1 = relu(y1.W1 + b1)
//   y2 = [l1, l0, F_p]              -> l2 =      y2.W2 + b2
//   out[p] = max over the 16 neighbours of [l2, l1, l0, F_p]      (72 + C channels)
// The reference materialises every concat and the [B,N,16,.] tensors in HBM.  One lane per (p, j)
// pair; a DPP row (16 lanes) is one point, so the max over neighbours is four v_max_f32_dpp.
template <int C>
__global__ __launch_bounds__(256) void edge_dense_conv_kernel(int npoints, int n_per_cloud, const float* __restrict__ F,
                                                               long ldf, const int* __restrict__ idx, int ldi, int ioff,
                                                               const float* __restrict__ W0, const float* __restrict__ b0,
                                                               const float* __restrict__ W1, const float* __restrict__ b1,
                                                               const float* __restrict__ W2, const float* __restrict__ b2,
                                                               float* __restrict__ Y, long ldy) {
    // Weights are indexed with wave-uniform (compile-time) offsets -> scalar loads through the constant
    // cache into SGPRs; the lane's input row stays in VGPRs (static indices, everything fully unrolled).
    constexpr int G = 24, K0 = 2 * C, K1 = G + C, K2 = 2 * G + C;
    const int p = blockIdx.x * 16 + (threadIdx.x >> 4);
    const int s = threadIdx.x & 15;
    const bool ok = p < npoints;
    const int pp = ok ? p : 0;
    const int cloud0 = (pp / n_per_cloud) * n_per_cloud;
    const int j = cloud0 + idx[(size_t)pp * ldi + ioff + s];
    float fi[C], df[C], acc[G], l0[G], l1[G];
#pragma unroll
    for (int c4 = 0; c4 < C / 4; ++c4) {
        const float4 a = *reinterpret_cast<const float4*>(F + (size_t)pp * ldf + c4 * 4);
        const float4 b = *reinterpret_cast<const float4*>(F + (size_t)j * ldf + c4 * 4);
        fi[c4 * 4 + 0] = a.x; fi[c4 * 4 + 1] = a.y; fi[c4 * 4 + 2] = a.z; fi[c4 * 4 + 3] = a.w;
        df[c4 * 4 + 0] = b.x - a.x; df[c4 * 4 + 1] = b.y - a.y; df[c4 * 4 + 2] = b.z - a.z; df[c4 * 4 + 3] = b.w - a.w;
    }
#pragma unroll
    for (int o = 0; o < G; ++o) acc[o] = 0.f;
#pragma unroll
    for (int k = 0; k < K0; ++k) {
        const float xv = (k < C) ? fi[k < C ? k : 0] : df[k < C ? 0 : k - C];
#pragma unroll
        for (int o = 0; o < G; ++o) acc[o] = __builtin_fmaf(xv, W0[k * G + o], acc[o]);
    }
#pragma unroll
    for (int o = 0; o < G; ++o) { l0[o] = fmaxf(acc[o] + b0[o], 0.f); acc[o] = 0.f; }
#pragma unroll
    for (int k = 0; k < K1; ++k) {
        const float xv = (k < G) ? l0[k < G ? k : 0] : fi[k < G ? 0 : k - G];
#pragma unroll
        for (int o = 0; o < G; ++o) acc[o] = __builtin_fmaf(xv, W1[k * G + o], acc[o]);
    }
#pragma unroll
    for (int o = 0; o < G; ++o) { l1[o] = fmaxf(acc[o] + b1[o], 0.f); acc[o] = 0.f; }
#pragma unroll
    for (int k = 0; k < K2; ++k) {
        const float xv = (k < G) ? l1[k < G ? k : 0] : ((k < 2 * G) ? l0[(k >= G && k < 2 * G) ? k - G : 0] : fi[k >= 2 * G ? k - 2 * G : 0]);
#pragma unroll
        for (int o = 0; o < G; ++o) acc[o] = __builtin_fmaf(xv, W2[k * G + o], acc[o]);
    }
    float* __restrict__ yr = Y + (size_t)pp * ldy;
    const bool writer = ok && s == 15;
#pragma unroll
    for (int o = 0; o < G; ++o) {
        const float m2 = row16_max_to_lane15(acc[o] + b2[o]);
        const float m1 = row16_max_to_lane15(l1[o]);
        const float m0 = row16_max_to_lane15(l0[o]);
        if (writer) { yr[o] = m2; yr[G + o] = m1; yr[2 * G + o] = m0; }
    }
    if (writer) {
#pragma unroll
        for (int c = 0; c < C; ++c) yr[3 * G + c] = fi[c];
    }
}

// ---------------------------------------------------------------------------------------------
// duplicate_up conv1 epilogue (Common/ops.py:1161-1191): the first 480 input channels of the 482-wide
// layer are identical for the 4 copies of a point, so H = chain_{k<480} feat.W is computed once per SOURCE
// point by dispu_linear (no bias); this kernel continues the same fmaf chain with the two grid channels of
// copy r and applies bias + ReLU:  out[b, r*n + i, o] = relu( fma(g_r1, W[481,o], fma(g_r0, W[480,o], H[b,i,o])) + bias[o] ).
// Bit-identical to evaluating the full 482-term chain for every copy, with 4x fewer FLOPs.
__global__ void dup_grid_kernel(int nclouds, int n, int co, int up, const float* __restrict__ H, long ldh,
                                const float* __restrict__ Wg /* rows 480,481 of W: [2, co] */,
                                const float* __restrict__ bias, const float* __restrict__ grid /* [up,2] */,
                                float* __restrict__ Y, long ldy) {
    // one source row per workgroup pass: H[src, :] is read once and expanded into its `up` copies (32-bit index
    // arithmetic once per row, float4 lanes over the channels; co % 4 == 0)
    const int co4 = co >> 2;
    const long nsrc = (long)nclouds * n;
    for (long src = blockIdx.x; src < nsrc; src += gridDim.x) {
        const long cloud = src / n;
        const int i = (int)(src - cloud * n);
        for (int o4 = threadIdx.x; o4 < co4; o4 += blockDim.x) {
            const float4 h = reinterpret_cast<const float4*>(H + src * ldh)[o4];
            const float4 w0 = reinterpret_cast<const float4*>(Wg)[o4];
            const float4 w1 = reinterpret_cast<const float4*>(Wg + co)[o4];
            const float4 bb = reinterpret_cast<const float4*>(bias)[o4];
            for (int r = 0; r < up; ++r) {
                const float g0 = grid[r * 2 + 0], g1 = grid[r * 2 + 1];
                float4 v;
                v.x = fmaxf(__builtin_fmaf(g1, w1.x, __builtin_fmaf(g0, w0.x, h.x)) + bb.x, 0.f);
                v.y = fmaxf(__builtin_fmaf(g1, w1.y, __builtin_fmaf(g0, w0.y, h.y)) + bb.y, 0.f);
                v.z = fmaxf(__builtin_fmaf(g1, w1.z, __builtin_fmaf(g0, w0.z, h.z)) + bb.z, 0.f);
                v.w = fmaxf(__builtin_fmaf(g1, w1.w, __builtin_fmaf(g0, w0.w, h.w)) + bb.w, 0.f);
                reinterpret_cast<float4*>(Y + ((cloud * up + r) * n + i) * ldy)[o4] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// PointShuffle2 (Common/ops.py:1012-1087) pieces.  rows = B*N points, k = 16 neighbours (idx from dispu_knn_xyz).

// conv0 (134 -> 128 + ReLU over [B,N,16,134]) is linear in its three input groups
//   [xyz_j - xyz_i (3) | xyz_j (3) | feat_j (128)], so  conv0(i,j) = relu(G[j] - A[i])  with
//   G[j] = feat_j.Wf + xyz_j.(Wc + Wr) + b   (per SOURCE point),  A[i] = xyz_i.Wc.
// 16x fewer FLOPs than the reference's per-pair conv; reassociated, hence tolerance-checked (no index
// decision depends on it).  This kernel adds the xyz terms to G (in place, G holds feat.Wf) and writes A.
__global__ void ps_prep_kernel(long rows, int co, const float* __restrict__ xyz, const float* __restrict__ W0 /*[134,co]*/,
                               const float* __restrict__ bias, float* __restrict__ Gm, long ldg, float* __restrict__ A, long lda) {
    const long total = rows * co;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int o = (int)(e % co);
        const long r = e / co;
        const float x = xyz[r * 3 + 0], y = xyz[r * 3 + 1], z = xyz[r * 3 + 2];
        const float wc0 = W0[0 * co + o], wc1 = W0[1 * co + o], wc2 = W0[2 * co + o];
        const float wr0 = W0[3 * co + o], wr1 = W0[4 * co + o], wr2 = W0[5 * co + o];
        float a = x * wc0; a = __builtin_fmaf(y, wc1, a); a = __builtin_fmaf(z, wc2, a);
        float g = x * (wc0 + wr0); g = __builtin_fmaf(y, wc1 + wr1, g); g = __builtin_fmaf(z, wc2 + wr2, g);
        A[r * lda + o] = a;
        Gm[r * ldg + o] = (Gm[r * ldg + o] + g) + bias[o];
    }
}

// float4 form (co % 4 == 0, 16-byte aligned rows): same per-element expressions.
__global__ __launch_bounds__(256) void ps_prep_v4_kernel(long rows, int co4, const float* __restrict__ xyz, const float* __restrict__ W0,
                                                          const float* __restrict__ bias, float* __restrict__ Gm, long ldg,
                                                          float* __restrict__ A, long lda) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long r = e / co4;
    const int q = (int)(e - r * co4);
    if (r >= rows) return;
    const int co = co4 * 4;
    const float x = xyz[r * 3 + 0], y = xyz[r * 3 + 1], z = xyz[r * 3 + 2];
    const float4 wc0 = *reinterpret_cast<const float4*>(W0 + 0 * co + q * 4), wc1 = *reinterpret_cast<const float4*>(W0 + 1 * co + q * 4),
                 wc2 = *reinterpret_cast<const float4*>(W0 + 2 * co + q * 4), wr0 = *reinterpret_cast<const float4*>(W0 + 3 * co + q * 4),
                 wr1 = *reinterpret_cast<const float4*>(W0 + 4 * co + q * 4), wr2 = *reinterpret_cast<const float4*>(W0 + 5 * co + q * 4);
    const float4 b = *reinterpret_cast<const float4*>(bias + q * 4);
    float4 g0 = *reinterpret_cast<const float4*>(Gm + r * ldg + q * 4);
    float4 a;
#define DISPU_PREP(c)                                                                                              \
    {                                                                                                              \
        float av = x * wc0.c; av = __builtin_fmaf(y, wc1.c, av); av = __builtin_fmaf(z, wc2.c, av);                \
        float gv = x * (wc0.c + wr0.c); gv = __builtin_fmaf(y, wc1.c + wr1.c, gv); gv = __builtin_fmaf(z, wc2.c + wr2.c, gv); \
        a.c = av;                                                                                                  \
        g0.c = (g0.c + gv) + b.c;                                                                                  \
    }
    DISPU_PREP(x) DISPU_PREP(y) DISPU_PREP(z) DISPU_PREP(w)
#undef DISPU_PREP
    *reinterpret_cast<float4*>(A + r * lda + q * 4) = a;
    *reinterpret_cast<float4*>(Gm + r * ldg + q * 4) = g0;
}

// X1[(i,s), c] = relu(G[cloud, idx[i,s], c] - A[i, c])     (float4 over c)
__global__ void ps_gather_sub_relu_kernel(long rows, int n_per_cloud, int k, int c4n, const int* __restrict__ idx,
                                          const float* __restrict__ Gm, long ldg, const float* __restrict__ A, long lda,
                                          float* __restrict__ X1, long ldx1) {
    const long total = rows * k * c4n;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(e % c4n);
        const long pr = e / c4n;         // pair row = i*k + s
        const long i = pr / k;
        const long j = (i / n_per_cloud) * n_per_cloud + idx[pr];
        const float4 g = *reinterpret_cast<const float4*>(Gm + j * ldg + c4 * 4);
        const float4 a = *reinterpret_cast<const float4*>(A + i * lda + c4 * 4);
        float4 o;
        o.x = fmaxf(g.x - a.x, 0.f); o.y = fmaxf(g.y - a.y, 0.f); o.z = fmaxf(g.z - a.z, 0.f); o.w = fmaxf(g.w - a.w, 0.f);
        *reinterpret_cast<float4*>(X1 + pr * ldx1 + c4 * 4) = o;
    }
}

// skip-branch input (ops.py:1049): max over the k neighbours of [xyz_j - xyz_i (3), xyz_j (3), feat_j (cf)].
__global__ void ps_skip_max_kernel(long rows, int n_per_cloud, int k, int cf, const int* __restrict__ idx,
                                   const float* __restrict__ xyz, const float* __restrict__ feat, long ldf,
                                   float* __restrict__ out, long ldo) {
    // 32 lanes per point: lane q owns feature channels 4q..4q+3 (float4 loads of whole 512-byte rows, cf == 128);
    // lanes 0..5 additionally produce the six xyz channels.  Neighbour ids are read once per point.
    const int sub = threadIdx.x & 31;
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    if (i >= rows) return;
    const long base = (i / n_per_cloud) * n_per_cloud;
    const float ninf = -__builtin_inff();
    float4 m = make_float4(ninf, ninf, ninf, ninf);
    float mx = ninf;
    const float ci = (sub < 3) ? xyz[i * 3 + sub] : 0.f;
    for (int s = 0; s < k; ++s) {
        const long j = base + idx[i * k + s];
        for (int c4 = sub; c4 * 4 < cf; c4 += 32) {      // one pass when cf <= 128
            const float4 v = *reinterpret_cast<const float4*>(feat + j * ldf + c4 * 4);
            m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        }
        if (sub < 6) {
            const float pj = xyz[j * 3 + (sub < 3 ? sub : sub - 3)];
            mx = fmaxf(mx, sub < 3 ? pj - ci : pj);
        }
    }
    float* o = out + i * ldo;
    if (sub < 6) o[sub] = mx;
    if (sub * 4 < cf) { o[6 + sub * 4] = m.x; o[7 + sub * 4] = m.y; o[8 + sub * 4] = m.z; o[9 + sub * 4] = m.w; }
}

// k = 16, cf = 128 (the generator's shape): the 16 neighbour ids of a point are fetched by 16 lanes at once and handed
// round with ds_bpermute, so the 16 row gathers (and the xyz reads) are all in flight together instead of forming a
// chain of 16 dependent id -> row round trips.  max is order-independent: same result as the loop above.
__global__ __launch_bounds__(256) void ps_skip_max16_kernel(long rows, int n_per_cloud, const int* __restrict__ idx,
                                                             const float* __restrict__ xyz, const float* __restrict__ feat, long ldf,
                                                             float* __restrict__ out, long ldo) {
    const int sub = threadIdx.x & 31;
    // xcd_block: every XCD works on one contiguous range of points, i.e. on whole clouds -- the neighbour rows of a cloud (512 KB) are
    // then gathered through ONE L2 instead of being pulled into all eight (counter traffic was 2.9x the algorithmic bytes)
    const long i = ((long)xcd_block(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x) >> 5;
    if (i >= rows) return;                                         // whole 32-lane groups leave together
    const long base = (i / n_per_cloud) * n_per_cloud;
    const int mine = idx[i * 16 + (sub & 15)];
    const float ci = (sub < 3) ? xyz[i * 3 + sub] : 0.f;
    const int xc = sub < 3 ? sub : sub - 3;
    float4 v[16];
    float pj[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const long j = base + __shfl(mine, s, 32);
        v[s] = *reinterpret_cast<const float4*>(feat + j * ldf + sub * 4);
        pj[s] = (sub < 6) ? xyz[j * 3 + xc] : 0.f;
    }
    const float ninf = -__builtin_inff();
    float4 m = make_float4(ninf, ninf, ninf, ninf);
    float mx = ninf;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        m.x = fmaxf(m.x, v[s].x); m.y = fmaxf(m.y, v[s].y); m.z = fmaxf(m.z, v[s].z); m.w = fmaxf(m.w, v[s].w);
        mx = fmaxf(mx, sub < 3 ? pj[s] - ci : pj[s]);
    }
    float* o = out + i * ldo;
    if (sub < 6) o[sub] = mx;
    o[6 + sub * 4] = m.x; o[7 + sub * 4] = m.y; o[8 + sub * 4] = m.z; o[9 + sub * 4] = m.w;
}

// weight_net_hidden (ops.py:181-191, 1064): wv[(i,s), t] = relu( (cxyz . Ww[:,t] + bw[t]) * scale[t] + shift[t] ),
// cxyz = xyz_j - xyz_i; scale/shift = inference BatchNorm folded by the host (eps 1e-3).
__global__ void ps_weight_net_kernel(long rows, int n_per_cloud, int k, int t_n, const int* __restrict__ idx,
                                     const float* __restrict__ xyz, const float* __restrict__ Ww, const float* __restrict__ bw,
                                     const float* __restrict__ scale, const float* __restrict__ shift, float* __restrict__ wv) {
    const long total = rows * k * t_n;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int t = (int)(e % t_n);
        const long pr = e / t_n;
        const long i = pr / k;
        const long j = (i / n_per_cloud) * n_per_cloud + idx[pr];
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) acc = __builtin_fmaf(xyz[j * 3 + c] - xyz[i * 3 + c], Ww[c * t_n + t], acc);
        acc = acc + bw[t];
        acc = acc * scale[t] + shift[t];
        wv[e] = fmaxf(acc, 0.f);
    }
}

// feature x weight (ops.py:1066-1067): out[i, c*16 + t] = chain_s X2[(i,s), c] * wv[(i,s), t]   (k = 16, t_n = 16)
// One 256-thread block per point: thread -> (c = tid/2, 8 consecutive t).
__global__ __launch_bounds__(256) void ps_point_matmul_kernel(long rows, const float* __restrict__ X2, long ldx2,
                                                               const float* __restrict__ wv, float* __restrict__ out, long ldo) {
    __shared__ float ws[16 * 16];
    const long i = blockIdx.x;
    ws[threadIdx.x] = wv[i * 256 + threadIdx.x];
    __syncthreads();
    const int c = threadIdx.x >> 1, t0 = (threadIdx.x & 1) * 8;
    float acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const float xv = X2[(i * 16 + s) * ldx2 + c];
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = __builtin_fmaf(xv, ws[s * 16 + t0 + t], acc[t]);
    }
    float* o = out + i * ldo + c * 16 + t0;
    *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    *reinterpret_cast<float4*>(o + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
}

// Row softmax of the attention logits (ops.py:326-338): S <- softmax(S * inv_scale) per row, in place.
// One wave per row; n <= 64*32.
__global__ __launch_bounds__(256) void softmax_rows_kernel(long rows, int n, float mul, float* __restrict__ S, long lds) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    float* p = S + row * lds;
    float v[32];
    float m = -__builtin_inff();
#pragma unroll
    for (int q = 0; q < 32; ++q) {
        const int c = lane + q * 64;
        v[q] = (c < n) ? p[c] * mul : -__builtin_inff();
        m = fmaxf(m, v[q]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < 32; ++q) { v[q] = expf(v[q] - m); sum += v[q]; }
    sum = wave_sum_f32(sum);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int q = 0; q < 32; ++q) {
        const int c = lane + q * 64;
        if (c < n) p[c] = v[q] * inv;
    }
}

static inline int grid_for(long total, int bs) {
    long g = (total + bs - 1) / bs;
    if (g > 32768) g = 32768;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace dispu

using namespace dispu;

DISPU_EXPORT int dispu_linear_small_k(long rows, int K, int N, const float* X, long ldx, const float* W, const float* bias,
                                      int act, float* Y, long ldy, void* stream) {
    if (rows < 0 || K <= 0 || K > 4 || !(N == 16 || N == 24) || !X || !W || !Y) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    const bool v4 = ((ldy & 3) == 0) && (((((uintptr_t)Y) | ((uintptr_t)W) | ((uintptr_t)bias)) & 15) == 0);
    if (v4) {
        const int g4 = (int)((rows * (N / 4) + 255) / 256);
        if (N == 24) hipLaunchKernelGGL((linear_small_k_v4_kernel<24>), dim3(g4), dim3(256), 0, (hipStream_t)stream, rows, K, X, ldx, W, bias, act, Y, ldy);
        else hipLaunchKernelGGL((linear_small_k_v4_kernel<16>), dim3(g4), dim3(256), 0, (hipStream_t)stream, rows, K, X, ldx, W, bias, act, Y, ldy);
        return (int)hipGetLastError();
    }
    const int g = (int)((rows + 255) / 256);
    if (N == 24) hipLaunchKernelGGL((linear_small_k_kernel<24>), dim3(g), dim3(256), 0, (hipStream_t)stream, rows, K, X, ldx, W, bias, act, Y, ldy);
    else hipLaunchKernelGGL((linear_small_k_kernel<16>), dim3(g), dim3(256), 0, (hipStream_t)stream, rows, K, X, ldx, W, bias, act, Y, ldy);
    return (int)hipGetLastError();
}

DISPU_EXPORT int dispu_linear_small_n(long rows, int K, int N, const float* X, long ldx, const float* W, const float* bias,
                                      int mode, const float* R, long ldr, float* Y, long ldy, void* stream) {
    if (rows < 0 || K <= 0 || N != 3 || !X || !W || !Y || (mode == 1 && !R) || mode < 0 || mode > 1) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    const int g = (int)((rows + 255) / 256);
    hipLaunchKernelGGL((linear_small_n_kernel<3>), dim3(g), dim3(256), 0, (hipStream_t)stream, rows, K, X, ldx, W, bias, mode, R, ldr, Y, ldy);
    return (int)hipGetLastError();
}

// VALU formulation (lane = pair, weights through scalar loads); kept as the A/B twin of the MFMA kernel in edge.hip.
DISPU_EXPORT int dispu_edge_dense_conv_valu(int npoints, int n_per_cloud, int C, const float* F, long ldf, const int* idx, int ldi,
                                       int ioff, const float* W0, const float* b0, const float* W1, const float* b1,
                                       const float* W2, const float* b2, float* Y, long ldy, void* stream) {
    if (npoints < 0 || n_per_cloud <= 0 || !(C == 24 || C == 48) || (ldf & 3) || (((uintptr_t)F) & 15)) return (int)hipErrorInvalidValue;
    if (npoints == 0) return 0;
    const int g = (npoints + 15) / 16;
    if (C == 24)
        hipLaunchKernelGGL((edge_dense_conv_kernel<24>), dim3(g), dim3(256), 0, (hipStream_t)stream, npoints, n_per_cloud, F, ldf, idx, ldi, ioff, W0, b0, W1, b1, W2, b2, Y, ldy);
    else
        hipLaunchKernelGGL((edge_dense_conv_kernel<48>), dim3(g), dim3(256), 0, (hipStream_t)stream, npoints, n_per_cloud, F, ldf, idx, ldi, ioff, W0, b0, W1, b1, W2, b2, Y, ldy);
    return (int)hipGetLastError();
}

DISPU_EXPORT int dispu_dup_grid(int nclouds, int n, int co, int up, const float* H, long ldh, const float* Wg, const float* bias,
                                const float* grid, float* Y, long ldy, void* stream) {
    if (nclouds < 0 || n <= 0 || co <= 0 || up <= 0 || (co & 3) || (ldh & 3) || (ldy & 3) ||
        ((((uintptr_t)H) | ((uintptr_t)Wg) | ((uintptr_t)bias) | ((uintptr_t)Y)) & 15))
        return (int)hipErrorInvalidValue;
    const long nsrc = (long)nclouds * n;
    if (nsrc == 0) return 0;
    const int bs = (co / 4 >= 64) ? 64 : 32;
    hipLaunchKernelGGL(dup_grid_kernel, dim3((unsigned)(nsrc > 65536 ? 65536 : nsrc)), dim3(bs), 0, (hipStream_t)stream, nclouds, n, co, up, H, ldh, Wg, bias, grid, Y, ldy);
    return (int)hipGetLastError();
}

DISPU_EXPORT int dispu_ps_prep(long rows, int co, const float* xyz, const float* W0, const float* bias, float* G, long ldg,
                               float* A, long lda, void* stream) {
    if (rows < 0 || co <= 0) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    if ((co & 3) == 0 && (ldg & 3) == 0 && (lda & 3) == 0 && (((((uintptr_t)W0) | ((uintptr_t)bias) | ((uintptr_t)G) | ((uintptr_t)A)) & 15) == 0)) {
        const long nt = rows * (co / 4);
        hipLaunchKernelGGL(ps_prep_v4_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rows, co / 4, xyz, W0, bias, G, ldg, A, lda);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(ps_prep_kernel, dim3(grid_for(rows * co, 256)), dim3(256), 0, (hipStream_t)stream, rows, co, xyz, W0, bias, G, ldg, A, lda);
    return (int)hipGetLastError();
}

DISPU_EXPORT int dispu_ps_gather_sub_relu(long rows, int n_per_cloud, int k, int c, const int* idx, const float* G, long ldg,
                                          const float* A, long lda, float* X1, long ldx1, void* stream) {
    if (rows < 0 || n_per_cloud <= 0 || k <= 0 || c <= 0 || (c & 3) || (ldg & 3) || (lda & 3) || (ldx1 & 3)) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    hipLaunchKernelGGL(ps_gather_sub_relu_kernel, dim3(grid_for(rows * k * (c / 4), 256)), dim3(256), 0, (hipStream_t)stream, rows, n_per_cloud, k, c / 4, idx, G, ldg, A, lda, X1, ldx1);
    return (int)hipGetLastError();
}

DISPU_EXPORT int dispu_ps_skip_max(long rows, int n_per_cloud, int k, int cf, const int* idx, const float* xyz, const float* feat,
                                   long ldf, float* out, long ldo, void* stream) {
    if (rows < 0 || n_per_cloud <= 0 || k <= 0 || cf <= 0 || cf > 128 || (cf & 3) || (ldf & 3) || (((uintptr_t)feat) & 15))
        return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    if (k == 16 && cf == 128)
        hipLaunchKernelGGL(ps_skip_max16_kernel, dim3((unsigned)((rows * 32 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rows, n_per_cloud, idx, xyz, feat, ldf, out, ldo);
    else
        hipLaunchKernelGGL(ps_skip_max_kernel, dim3((unsigned)((rows * 32 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rows, n_per_cloud, k, cf, idx, xyz, feat, ldf, out, ldo);
    return (int)hipGetLastError();
}

DISPU_EXPORT int dispu_ps_weight_net(long rows, int n_per_cloud, int k, int t_n, const int* idx, const float* xyz, const float* Ww,
                                     const float* bw, const float* scale, const float* shift, float* wv, void* stream) {
    if (rows < 0 || n_per_cloud <= 0 || k <= 0 || t_n <= 0) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    hipLaunchKernelGGL(ps_weight_net_kernel, dim3(grid_for(rows * k * t_n, 256)), dim3(256), 0, (hipStream_t)stream, rows, n_per_cloud, k, t_n, idx, xyz, Ww, bw, scale, shift, wv);
    return (int)hipGetLastError();
}

DISPU_EXPORT int dispu_ps_point_matmul(long rows, int k, int c, int t_n, const float* X2, long ldx2, const float* wv, float* out,
                                       long ldo, void* stream) {
    if (rows < 0 || k != 16 || c != 128 || t_n != 16 || (ldo & 3)) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    hipLaunchKernelGGL(ps_point_matmul_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, rows, X2, ldx2, wv, out, ldo);
    return (int)hipGetLastError();
}

DISPU_EXPORT int dispu_softmax_rows(long rows, int n, float mul, float* S, long lds, void* stream) {
    if (rows < 0 || n <= 0 || n > 2048) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, rows, n, mul, S, lds);
    return (int)hipGetLastError();
}
