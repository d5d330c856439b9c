// Approximate earth-mover matching (annealed auction) and its cost / gradients for gfx950.
// Replaces approxmatchLauncher / matchcostLauncher / matchcostgradLauncher
// (tf_ops/approxmatch/tf_approxmatch_g.cu:180-182,226-228,292-295).
//
// The reference runs ONE 512-thread block per cloud for all 10 levels x 3 passes.  The three
// passes of a level depend on each other only through the per-point vectors remainL/R and
// ratioL/R, so here every pass is its own launch over (point tiles x clouds): all CUs work
// even at small batch, and each point's sum is still accumulated sequentially in the reference's
// order (l or k ascending), so results do not depend on the decomposition.
// Scratch `temp` has the reference's size and role: [b][2*(n+m)] floats =
// remainL[n] | remainR[m] | ratioL[n] | ratioR[m]   (tf_approxmatch_g.cu:2).
#include "common.h"

namespace dispu {

constexpr int AM_BS = 256;
constexpr int AM_TILE = 1024;

// PINNED = false: hardware v_exp_f32 like the reference's __expf.  PINNED = true: the same fmaf-chain
// sequence as oracle/dispu_oracle.c:pinned_exp, bit-identical on CPU and GPU (parity mode).
template <bool PINNED>
__device__ __forceinline__ float am_exp(float x) {
    if constexpr (!PINNED) {
        return __expf(x);
    } else {
        if (!(x > -86.0f)) return 0.0f;
        const float t = x * 1.44269504088896341f;
        const float n = __builtin_rintf(t);
        const float f = t - n;
        float p = 1.5403530393381609954e-4f;
        p = __builtin_fmaf(p, f, 1.3333558146428443423e-3f);
        p = __builtin_fmaf(p, f, 9.6181291076284771619e-3f);
        p = __builtin_fmaf(p, f, 5.5504108664821579953e-2f);
        p = __builtin_fmaf(p, f, 2.4022650695910071233e-1f);
        p = __builtin_fmaf(p, f, 6.9314718055994530942e-1f);
        p = __builtin_fmaf(p, f, 1.0f);
        return __int_as_float(__float_as_int(p) + (((int)n) << 23));
    }
}

__global__ void am_init_kernel(int n, int m, float multiL, float multiR, float* __restrict__ temp,
                               float* __restrict__ match) {
    const int cloud = blockIdx.y;
    float* t = temp + (size_t)cloud * (n + m) * 2;
    float* mt = match + (size_t)cloud * n * m;
    const size_t nm = (size_t)n * m;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < nm; e += (size_t)gridDim.x * blockDim.x) mt[e] = 0.f;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) t[e] = multiL;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < m; e += gridDim.x * blockDim.x) t[n + e] = multiR;
}

// PASS 1: ratioL[k] = remainL[k] / (1e-9 + sum_l exp(level*d2)*remainR[l])
// PASS 2: sumr = (sum_k exp(level*d2)*ratioL[k]) * remainR[l]; ratioR, remainR update
// PASS 3: w = exp(level*d2)*ratioL[k]*ratioR[l]; match[l*n+k] += w; remainL update
template <int PASS, bool FMA, bool PINNED>
__global__ __launch_bounds__(AM_BS) void am_pass_kernel(int n, int m, float level, const float* __restrict__ xyz1,
                                                         const float* __restrict__ xyz2, float* __restrict__ temp,
                                                         float* __restrict__ match) {
    __shared__ float4 tile[AM_TILE];
    const int cloud = blockIdx.y;
    float* remainL = temp + (size_t)cloud * (n + m) * 2;
    float* remainR = remainL + n;
    float* ratioL = remainR + m;
    float* ratioR = ratioL + n;
    const float* __restrict__ p1 = xyz1 + (size_t)cloud * n * 3;
    const float* __restrict__ p2 = xyz2 + (size_t)cloud * m * 3;
    float* mt = match + (size_t)cloud * n * m;

    constexpr bool OVER_N = (PASS != 2);           // lanes enumerate cloud 1 (k) in passes 1,3
    const int nown = OVER_N ? n : m, noth = OVER_N ? m : n;
    const float* own = OVER_N ? p1 : p2;
    const float* oth = OVER_N ? p2 : p1;
    const float* othw = (PASS == 1) ? remainR : (PASS == 2 ? ratioL : ratioR);
    const int a = blockIdx.x * AM_BS + threadIdx.x;
    const bool active = a < nown;
    float xa = 0.f, ya = 0.f, za = 0.f;
    if (active) { xa = own[a * 3 + 0]; ya = own[a * 3 + 1]; za = own[a * 3 + 2]; }
    const float rl = (PASS == 3 && active) ? ratioL[a] : 0.f;
    float sum = (PASS == 1) ? 1e-9f : 0.f;
    for (int t0 = 0; t0 < noth; t0 += AM_TILE) {
        const int len = min(AM_TILE, noth - t0);
        __syncthreads();
        for (int t = threadIdx.x; t < len; t += AM_BS)
            tile[t] = make_float4(oth[(t0 + t) * 3 + 0], oth[(t0 + t) * 3 + 1], oth[(t0 + t) * 3 + 2], othw[t0 + t]);
        __syncthreads();
        if (!active) continue;
        for (int t = 0; t < len; ++t) {
            const float4 q = tile[t];
            const float d2 = OVER_N ? sqdist3<FMA>(q.x - xa, q.y - ya, q.z - za) : sqdist3<FMA>(xa - q.x, ya - q.y, za - q.z);
            const float e = am_exp<PINNED>(level * d2);
            if constexpr (PASS == 3) {
                const float w = e * rl * q.w;
                mt[(size_t)(t0 + t) * n + a] += w;
                sum += w;
            } else {
                if constexpr (FMA) sum = __builtin_fmaf(e, q.w, sum);
                else sum = sum + e * q.w;
            }
        }
    }
    if (!active) return;
    if constexpr (PASS == 1) {
        ratioL[a] = remainL[a] / sum;
    } else if constexpr (PASS == 2) {
        const float rr = remainR[a];
        const float sumr = sum * rr;
        const float consumption = fminf(rr / (sumr + 1e-9f), 1.0f);
        ratioR[a] = consumption * rr;
        remainR[a] = fmaxf(0.0f, rr - sumr);
    } else {
        remainL[a] = fmaxf(0.0f, remainL[a] - sum);
    }
}

// cost[b] = sum_{k,l} sqrt(d2(k,l)) * match[l*n+k].  One workgroup per cloud; lane t sums
// k = t, t+BS, ... (outer) x l ascending (inner) exactly like the reference's thread t
// (tf_approxmatch_g.cu:183-225), then partials are combined wave-first (deterministic order).
template <bool FMA>
__global__ __launch_bounds__(1024) void match_cost_kernel(int n, int m, const float* __restrict__ xyz1,
                                                           const float* __restrict__ xyz2,
                                                           const float* __restrict__ match, float* __restrict__ cost) {
    constexpr int BS = 1024;
    __shared__ float4 tile[AM_TILE];
    __shared__ float wsum[BS / kWave];
    const int cloud = blockIdx.x;
    const float* __restrict__ p1 = xyz1 + (size_t)cloud * n * 3;
    const float* __restrict__ p2 = xyz2 + (size_t)cloud * m * 3;
    const float* __restrict__ mt = match + (size_t)cloud * n * m;
    float sub = 0.f;
    for (int k0 = 0; k0 < n; k0 += BS) {
        const int k = k0 + threadIdx.x;
        const bool active = k < n;
        float x1 = 0.f, y1 = 0.f, z1 = 0.f;
        if (active) { x1 = p1[k * 3 + 0]; y1 = p1[k * 3 + 1]; z1 = p1[k * 3 + 2]; }
        for (int l0 = 0; l0 < m; l0 += AM_TILE) {
            const int len = min(AM_TILE, m - l0);
            __syncthreads();
            for (int t = threadIdx.x; t < len; t += BS)
                tile[t] = make_float4(p2[(l0 + t) * 3 + 0], p2[(l0 + t) * 3 + 1], p2[(l0 + t) * 3 + 2], 0.f);
            __syncthreads();
            if (active) {
                for (int t = 0; t < len; ++t) {
                    const float4 q = tile[t];
                    const float d = sqrtf(sqdist3<FMA>(q.x - x1, q.y - y1, q.z - z1));
                    const float w = mt[(size_t)(l0 + t) * n + k];
                    if constexpr (FMA) sub = __builtin_fmaf(d, w, sub);
                    else sub = sub + d * w;
                }
            }
        }
    }
    sub = wave_sum_f32(sub);
    if ((threadIdx.x & (kWave - 1)) == 0) wsum[threadIdx.x / kWave] = sub;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int w = 0; w < BS / kWave; ++w) s += wsum[w];
        cost[cloud] = s;
    }
}

// grad1[b,k,:] = sum_l match[l*n+k] * (p1_k - p2_l) * rsqrt(max(d2,1e-20))   (matchcostgrad1, :270-291)
// grad2[b,l,:] = sum_k match[l*n+k] * (p2_l - p1_k) * rsqrt(max(d2,1e-20))   (matchcostgrad2, :229-269)
// WHICH = 1: lane per k, sequential over l (the reference's order).
// WHICH = 2: one wave per l, lanes stride over k, wave butterfly sum (reference: 256-thread tree).
template <int WHICH, bool FMA>
__global__ __launch_bounds__(AM_BS) void match_cost_grad_kernel(int n, int m, const float* __restrict__ xyz1,
                                                                 const float* __restrict__ xyz2,
                                                                 const float* __restrict__ match,
                                                                 float* __restrict__ grad) {
    const int cloud = blockIdx.y;
    const float* __restrict__ p1 = xyz1 + (size_t)cloud * n * 3;
    const float* __restrict__ p2 = xyz2 + (size_t)cloud * m * 3;
    const float* __restrict__ mt = match + (size_t)cloud * n * m;
    if constexpr (WHICH == 1) {
        __shared__ float4 tile[AM_TILE];
        const int k = blockIdx.x * AM_BS + threadIdx.x;
        const bool active = k < n;
        float x1 = 0.f, y1 = 0.f, z1 = 0.f;
        if (active) { x1 = p1[k * 3 + 0]; y1 = p1[k * 3 + 1]; z1 = p1[k * 3 + 2]; }
        float gx = 0.f, gy = 0.f, gz = 0.f;
        for (int l0 = 0; l0 < m; l0 += AM_TILE) {
            const int len = min(AM_TILE, m - l0);
            __syncthreads();
            for (int t = threadIdx.x; t < len; t += AM_BS)
                tile[t] = make_float4(p2[(l0 + t) * 3 + 0], p2[(l0 + t) * 3 + 1], p2[(l0 + t) * 3 + 2], 0.f);
            __syncthreads();
            if (!active) continue;
            for (int t = 0; t < len; ++t) {
                const float4 q = tile[t];
                const float ex = x1 - q.x, ey = y1 - q.y, ez = z1 - q.z;
                const float d = mt[(size_t)(l0 + t) * n + k] * rsqrtf(fmaxf(sqdist3<FMA>(ex, ey, ez), 1e-20f));
                if constexpr (FMA) { gx = __builtin_fmaf(ex, d, gx); gy = __builtin_fmaf(ey, d, gy); gz = __builtin_fmaf(ez, d, gz); }
                else { gx += ex * d; gy += ey * d; gz += ez * d; }
            }
        }
        if (active) { float* g = grad + ((size_t)cloud * n + k) * 3; g[0] = gx; g[1] = gy; g[2] = gz; }
    } else {
        const int lane = threadIdx.x & (kWave - 1);
        const int l = blockIdx.x * (AM_BS / kWave) + threadIdx.x / kWave;
        if (l >= m) return;  // wave-uniform
        const float x2 = p2[l * 3 + 0], y2 = p2[l * 3 + 1], z2 = p2[l * 3 + 2];
        float gx = 0.f, gy = 0.f, gz = 0.f;
        for (int k = lane; k < n; k += kWave) {
            const float ex = x2 - p1[k * 3 + 0], ey = y2 - p1[k * 3 + 1], ez = z2 - p1[k * 3 + 2];
            const float d = mt[(size_t)l * n + k] * rsqrtf(fmaxf(sqdist3<FMA>(ex, ey, ez), 1e-20f));
            if constexpr (FMA) { gx = __builtin_fmaf(ex, d, gx); gy = __builtin_fmaf(ey, d, gy); gz = __builtin_fmaf(ez, d, gz); }
            else { gx += ex * d; gy += ey * d; gz += ez * d; }
        }
        gx = wave_sum_f32(gx); gy = wave_sum_f32(gy); gz = wave_sum_f32(gz);
        if (lane == 0) { float* g = grad + ((size_t)cloud * m + l) * 3; g[0] = gx; g[1] = gy; g[2] = gz; }
    }
}

template <bool FMA, bool PINNED>
static int run_approx_match(int b, int n, int m, const float* xyz1, const float* xyz2, float* match, float* temp,
                            hipStream_t s) {
    const float multiL = (n >= m) ? 1.0f : (float)(m / n);
    const float multiR = (n >= m) ? (float)(n / m) : 1.0f;
    hipLaunchKernelGGL(am_init_kernel, dim3(64, b), dim3(256), 0, s, n, m, multiL, multiR, temp, match);
    dim3 gn((n + AM_BS - 1) / AM_BS, b), gm((m + AM_BS - 1) / AM_BS, b);
    for (int j = 7; j >= -2; --j) {
        float level = 0.0f;
        if (j != -2) {
            level = -1.0f;
            for (int t = 0; t < (j < 0 ? -j : j); ++t) level = (j < 0) ? level * 0.25f : level * 4.0f;  // -(4^j), exact
        }
        hipLaunchKernelGGL((am_pass_kernel<1, FMA, PINNED>), gn, dim3(AM_BS), 0, s, n, m, level, xyz1, xyz2, temp, match);
        hipLaunchKernelGGL((am_pass_kernel<2, FMA, PINNED>), gm, dim3(AM_BS), 0, s, n, m, level, xyz1, xyz2, temp, match);
        hipLaunchKernelGGL((am_pass_kernel<3, FMA, PINNED>), gn, dim3(AM_BS), 0, s, n, m, level, xyz1, xyz2, temp, match);
    }
    return (int)hipGetLastError();
}

}  // namespace dispu

using namespace dispu;

DISPU_EXPORT size_t dispu_approx_match_scratch_bytes(int b, int n, int m) {
    return sizeof(float) * (size_t)b * ((size_t)n + m) * 2;
}

DISPU_EXPORT int dispu_approx_match(int b, int n, int m, const float* xyz1, const float* xyz2, float* match,
                                    float* temp, int arith, void* stream) {
    if (b < 0 || n <= 0 || m <= 0 || !temp) return (int)hipErrorInvalidValue;
    if (b == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const bool fma = (arith & DISPU_ARITH_CONTRACT) != 0, pin = (arith & DISPU_ARITH_PINNED_EXP) != 0;
    if (fma && pin) return run_approx_match<true, true>(b, n, m, xyz1, xyz2, match, temp, s);
    if (fma) return run_approx_match<true, false>(b, n, m, xyz1, xyz2, match, temp, s);
    if (pin) return run_approx_match<false, true>(b, n, m, xyz1, xyz2, match, temp, s);
    return run_approx_match<false, false>(b, n, m, xyz1, xyz2, match, temp, s);
}

DISPU_EXPORT int dispu_match_cost(int b, int n, int m, const float* xyz1, const float* xyz2, const float* match,
                                  float* cost, int arith, void* stream) {
    if (b < 0 || n <= 0 || m <= 0) return (int)hipErrorInvalidValue;
    if (b == 0) return 0;
    if ((arith & DISPU_ARITH_CONTRACT))
        hipLaunchKernelGGL((match_cost_kernel<true>), dim3(b), dim3(1024), 0, (hipStream_t)stream, n, m, xyz1, xyz2, match, cost);
    else
        hipLaunchKernelGGL((match_cost_kernel<false>), dim3(b), dim3(1024), 0, (hipStream_t)stream, n, m, xyz1, xyz2, match, cost);
    return (int)hipGetLastError();
}

DISPU_EXPORT int dispu_match_cost_grad(int b, int n, int m, const float* xyz1, const float* xyz2, const float* match,
                                       float* grad1, float* grad2, int arith, void* stream) {
    if (b < 0 || n <= 0 || m <= 0) return (int)hipErrorInvalidValue;
    if (b == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    dim3 g1((n + AM_BS - 1) / AM_BS, b), g2((m + AM_BS / kWave - 1) / (AM_BS / kWave), b);
    if ((arith & DISPU_ARITH_CONTRACT)) {
        hipLaunchKernelGGL((match_cost_grad_kernel<1, true>), g1, dim3(AM_BS), 0, s, n, m, xyz1, xyz2, match, grad1);
        hipLaunchKernelGGL((match_cost_grad_kernel<2, true>), g2, dim3(AM_BS), 0, s, n, m, xyz1, xyz2, match, grad2);
    } else {
        hipLaunchKernelGGL((match_cost_grad_kernel<1, false>), g1, dim3(AM_BS), 0, s, n, m, xyz1, xyz2, match, grad1);
        hipLaunchKernelGGL((match_cost_grad_kernel<2, false>), g2, dim3(AM_BS), 0, s, n, m, xyz1, xyz2, match, grad2);
    }
    return (int)hipGetLastError();
}
