// Approximate earth-mover matching (annealed auction) and its cost / gradients for gfx950.
// Replaces approxmatchLauncher / matchcostLauncher / matchcostgradLauncher
// (tf_ops/approxmatch/tf_approxmatch_g.cu:180-182,226-228,292-295).
//
// The reference runs ONE 512-thread block per cloud through 10 levels x 3 passes; a pass is a row sum (passes 1, 3:
// per point k of cloud 1 over all l of cloud 2) or a column sum (pass 2) of exp(level * d2) times a per-point weight,
// and every pass needs ALL results of the previous one.  Here:
//   * a pass is a 2-D tiling: a workgroup owns 256 points (one lane each) x AM_CH = 128 partners and writes one
//     partial sum per point, so a single 4096 x 4096 cloud is 512 workgroups instead of 1 (16 in round 1);
//   * partial sums are combined in ascending chunk order by whoever consumes them (the next pass, while it loads its
//     partner tile): no atomics, no grid barrier, no extra launch -- the association is fixed and restated by
//     oracle/dispu_oracle.c:orc_approx_match_chunked(chunk = 128), bit for bit in DISPU_ARITH_PINNED_EXP mode;
//   * pass 3 of level t and pass 1 of level t+1 are both row sums over the same pairs: one kernel, one d2 per pair;
//   * `match` is not read-modify-written 10 times: the ten (ratioL, ratioR) vector pairs are kept (40 (n + m) floats
//     per cloud) and match[l][k] = sum_t exp(level_t d2) ratioL_t[k] ratioR_t[l] is assembled once at the end in the
//     reference's order (ascending t from 0), written once; pass 3 of the last level only updates remainL, which
//     nobody reads afterwards, and is not evaluated.
// 22 launches per call (init, 10 column passes, 1 + 9 row passes, assembly) instead of 31.
#include "common.h"

#include <cstdlib>

namespace dispu {

constexpr int AM_ROWS = 256;   // points per workgroup, one lane each
constexpr int AM_CH = 128;     // partners per partial sum -- part of the pinned arithmetic (oracle AM_CHUNK)
constexpr int AM_LEVELS = 10;
constexpr int AM_ACH = 64;     // partners per workgroup of the assembly kernel (no sums across partners there)

struct AmLevels { float v[AM_LEVELS]; };

// per-cloud scratch (floats):  remL[10][n] ratL[10][n] remR[10][m] ratR[10][m] p1[nc1][n] p3[nc1][n] p2[nc2][m]
// remX[t] / ratX[t]: remainL/R entering level t, ratioL/R of level t; p1/p3: partial row sums of passes 1 / 3 per
// chunk of cloud-2 points; p2: partial column sums of pass 2 per chunk of cloud-1 points.
struct AmView { float *remL, *ratL, *remR, *ratR, *p1, *p3, *p2; };

__host__ __device__ inline int am_chunks(int x) { return (x + AM_CH - 1) / AM_CH; }
__host__ __device__ inline size_t am_cloud_floats(int n, int m) {
    return (size_t)2 * AM_LEVELS * ((size_t)n + m) + (size_t)2 * am_chunks(m) * n + (size_t)am_chunks(n) * m;
}
__device__ __forceinline__ AmView am_view(float* temp, int cloud, int n, int m) {
    AmView v;
    float* p = temp + (size_t)cloud * am_cloud_floats(n, m);
    v.remL = p; p += (size_t)AM_LEVELS * n;
    v.ratL = p; p += (size_t)AM_LEVELS * n;
    v.remR = p; p += (size_t)AM_LEVELS * m;
    v.ratR = p; p += (size_t)AM_LEVELS * m;
    v.p1 = p; p += (size_t)am_chunks(m) * n;
    v.p3 = p; p += (size_t)am_chunks(m) * n;
    v.p2 = p;
    return v;
}

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

// exp(level * d2) of the auction.  Hardware mode: the reference's __expf(level * d2) is exp2((level * d2) * log2e); every level
// is 0 or -(4^j), a power of two, so level * d2 is exact and (level * d2) * log2e == d2 * (level * log2e) BIT FOR BIT - the
// host folds level * log2e (also exact) and the kernels save one multiply per exponential.
constexpr float AM_LOG2E = 0x1.715476p+0f;       // the float the native exp uses
template <bool PINNED>
__device__ __forceinline__ float am_exp_level(float d2, float level) {
    if constexpr (PINNED) return am_exp<true>(level * d2);
    else return __builtin_amdgcn_exp2f(d2 * (level * AM_LOG2E));   // level * AM_LOG2E is loop-invariant (hoisted)
}

// COH (always false since round 5, kept as the documented way to exchange data between co-resident workgroups): a persistent kernel's
// workgroups exchange the scratch vectors while the kernel runs, across XCDs (separate L2s).  Fencing
// them with agent-scope release / acquire costs an L2 write-back + invalidate per workgroup and stage (measured: ~12 us per stage,
// the persistent form was SLOWER than 22 launches).  Instead every access to the exchanged arrays is itself an agent-scope atomic
// (relaxed): global_load / global_store ... sc1, coherent at the device's memory side without touching the rest of the cache.
template <bool COH>
__device__ __forceinline__ float am_ld(const float* p) {
    if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}
template <bool COH>
__device__ __forceinline__ void am_st(float* p, float v) {
    if constexpr (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// init + p[0] + p[stride] + ... + p[(nc-1) stride], added in ascending order (the pinned association), with the loads issued eight
// at a time: as a loop of dependent "load, add" steps the nc partials were a chain of nc memory round trips (uncached ones in the
// persistent kernel).  first_is_init: the sum starts AS p[0] (no leading addition), like the reference's `tot = p[0]; tot += p[c]`.
template <bool COH>
__device__ __forceinline__ float am_sum_partials(const float* p, size_t stride, int nc, float init, bool first_is_init) {
    float tot = init;
    for (int c0 = 0; c0 < nc; c0 += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (c0 + u < nc) ? am_ld<COH>(p + (size_t)(c0 + u) * stride) : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (c0 + u < nc) tot = (first_is_init && c0 + u == 0) ? v[u] : tot + v[u];
    }
    return tot;
}

__global__ void am_init_kernel(int n, int m, float multiL, float multiR, float* __restrict__ temp) {
    const AmView v = am_view(temp, blockIdx.y, n, m);
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) v.remL[e] = multiL;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < m; e += gridDim.x * blockDim.x) v.remR[e] = multiR;
}

// pass 2 of level t for one partner l, from the column partials:  (tf_approxmatch_g.cu:100-107)
//   sumr = (sum_k e ratioL[k]) * remainR[l];  ratioR = min(remainR/(sumr+1e-9), 1) * remainR;  remainR' = max(0, remainR - sumr)
template <bool COH = false>
__device__ __forceinline__ void am_finish_col(const AmView& v, int m, int nc2, int t, int l, float& ratr, float& remn) {
    const float rr = am_ld<COH>(&v.remR[(size_t)t * m + l]);
    const float tot = am_sum_partials<COH>(v.p2 + l, (size_t)m, nc2, 0.f, true);
    const float sumr = tot * rr;
    const float consumption = fminf(rr / (sumr + 1e-9f), 1.0f);
    ratr = consumption * rr;
    remn = fmaxf(0.0f, rr - sumr);
}

// Row kernel: pass 3 of level t (weights ratioL_t[k] ratioR_t[l]) and pass 1 of level t+1 (weights remainR_{t+1}[l]) over
// one (256 points of cloud 1) x (AM_CH points of cloud 2) tile.  FIRST: only pass 1 of level 0.
// ratioR_t / remainR_{t+1} of the partners are finished here from pass 2's partials (every workgroup of a chunk computes the
// same values; the rb == 0 one stores them for the later kernels).
template <bool FIRST, bool FMA, bool PINNED, bool COH = false>
__device__ __forceinline__ void am_row_body(int cloud, int rb, int c, float4* tile, float* tilew, int n, int m, int t, float level3, float level1,
                                            const float* __restrict__ xyz1, const float* __restrict__ xyz2, float* __restrict__ temp) {
    const int tid = threadIdx.x;
    const AmView v = am_view(temp, cloud, n, m);
    const int nc2 = am_chunks(n);
    const float* __restrict__ p1 = xyz1 + (size_t)cloud * n * 3;
    const float* __restrict__ p2 = xyz2 + (size_t)cloud * m * 3;
    const int l0 = c * AM_CH, len = min(AM_CH, m - l0);
    if (tid < len) {
        const int l = l0 + tid;
        float wr = 0.f, w1;
        if constexpr (FIRST) {
            w1 = am_ld<COH>(&v.remR[l]);
        } else {
            am_finish_col<COH>(v, m, nc2, t, l, wr, w1);
            if (rb == 0) { am_st<COH>(&v.ratR[(size_t)t * m + l], wr); am_st<COH>(&v.remR[(size_t)(t + 1) * m + l], w1); }
        }
        tile[tid] = make_float4(p2[l * 3 + 0], p2[l * 3 + 1], p2[l * 3 + 2], wr);
        tilew[tid] = w1;
    }
    __syncthreads();
    const int k = rb * AM_ROWS + tid;
    const bool active = k < n;
    float x1 = 0.f, y1 = 0.f, z1 = 0.f, rl = 0.f;
    if (active) {
        x1 = p1[k * 3 + 0]; y1 = p1[k * 3 + 1]; z1 = p1[k * 3 + 2];
        if constexpr (!FIRST) rl = am_ld<COH>(&v.ratL[(size_t)t * n + k]);
    }
    // m <= AM_CH: the one tile holds every partner of the point, so pass 1's chain IS the reference's sequential chain when it starts
    // at 1e-9f (tf_approxmatch_g.cu:60 `suml = 1e-9f`) instead of being added to it afterwards; the consumer then takes p1[0] as is
    float s3 = 0.f, s1 = (m <= AM_CH) ? 1e-9f : 0.f;
    if constexpr (!FIRST && FMA && !PINNED) {
        // Production arithmetic (contracted, hardware exp: tolerance-checked, not bit-pinned).  The pair loop is vector-issue bound,
        // so ratioL[k] -- a factor of every term of the row's pass-3 sum -- multiplies the SUM once and the terms become one fma
        // each (14.5 -> 12.5 instructions per pair, same EMD error against the oracle: tools/debug/am_check.py); the last row pass
        // has level1 == 0, its e1 is exactly 1 and is not evaluated.  (exp(level3 d2) as the fourth power of exp(level1 d2) -- the
        // levels descend by factors of four -- saves another exponential but makes pass 3 disagree with passes 1 - 2 of the same
        // level by a few ulp, which the auction's clamps amplify: EMD error 7e-5 on the reference's golden clouds.  Not used.)
        const float c1 = level1 * AM_LOG2E, c3 = level3 * AM_LOG2E;
        if (level1 != 0.f) {
#pragma unroll 4
            for (int i = 0; i < len; ++i) {
                const float4 q = tile[i];
                const float d2 = sqdist3<true>(q.x - x1, q.y - y1, q.z - z1);
                s3 = __builtin_fmaf(__builtin_amdgcn_exp2f(d2 * c3), q.w, s3);
                s1 = __builtin_fmaf(__builtin_amdgcn_exp2f(d2 * c1), tilew[i], s1);
            }
        } else {
#pragma unroll 4
            for (int i = 0; i < len; ++i) {
                const float4 q = tile[i];
                const float d2 = sqdist3<true>(q.x - x1, q.y - y1, q.z - z1);
                s3 = __builtin_fmaf(__builtin_amdgcn_exp2f(d2 * c3), q.w, s3);
                s1 = s1 + tilew[i];
            }
        }
        s3 *= rl;
    } else {
#pragma unroll 4
        for (int i = 0; i < len; ++i) {
            const float4 q = tile[i];
            const float d2 = sqdist3<FMA>(q.x - x1, q.y - y1, q.z - z1);
            if constexpr (!FIRST) {
                const float w = am_exp_level<PINNED>(d2, level3) * rl * q.w;      // the value `match` receives at this level
                s3 += w;
            }
            const float e1 = am_exp_level<PINNED>(d2, level1);
            if constexpr (FMA) s1 = __builtin_fmaf(e1, tilew[i], s1);
            else s1 = s1 + e1 * tilew[i];
        }
    }
    if (active) {
        if constexpr (!FIRST) am_st<COH>(&v.p3[(size_t)c * n + k], s3);
        am_st<COH>(&v.p1[(size_t)c * n + k], s1);
    }
}

template <bool FIRST, bool FMA, bool PINNED>
__global__ __launch_bounds__(AM_ROWS) void am_row_kernel(int n, int m, int t, float level3, float level1,
                                                          const float* __restrict__ xyz1, const float* __restrict__ xyz2,
                                                          float* __restrict__ temp) {
    __shared__ float4 tile[AM_CH];
    __shared__ float tilew[AM_CH];
    am_row_body<FIRST, FMA, PINNED>(blockIdx.z, blockIdx.x, blockIdx.y, tile, tilew, n, m, t, level3, level1, xyz1, xyz2, temp);
}

// Column kernel: pass 2 of level t over one (256 points of cloud 2) x (AM_CH points of cloud 1) tile.
// remainL_t / ratioL_t of the partners are finished here from the row partials (:56-57, :158-159):
//   remainL_t = max(0, remainL_{t-1} - sum_l w);  ratioL_t = remainL_t / (1e-9 + sum_l e remainR[l])
template <bool FMA, bool PINNED, bool COH = false>
__device__ __forceinline__ void am_col_body(int cloud, int cb, int c2, float4* tile, int n, int m, int t, float level, const float* __restrict__ xyz1,
                                            const float* __restrict__ xyz2, float* __restrict__ temp) {
    const int tid = threadIdx.x;
    const AmView v = am_view(temp, cloud, n, m);
    const int nc1 = am_chunks(m);
    const float* __restrict__ p1 = xyz1 + (size_t)cloud * n * 3;
    const float* __restrict__ p2 = xyz2 + (size_t)cloud * m * 3;
    const int k0 = c2 * AM_CH, len = min(AM_CH, n - k0);
    if (tid < len) {
        const int k = k0 + tid;
        float reml;
        if (t == 0) {
            reml = am_ld<COH>(&v.remL[k]);
        } else {
            const float tot = am_sum_partials<COH>(v.p3 + k, (size_t)n, nc1, 0.f, true);
            reml = fmaxf(0.0f, am_ld<COH>(&v.remL[(size_t)(t - 1) * n + k]) - tot);
        }
        const float suml = am_sum_partials<COH>(v.p1 + k, (size_t)n, nc1, 1e-9f, nc1 == 1);     // one chunk: its chain began at 1e-9f
        const float ratl = reml / suml;
        if (cb == 0) {
            if (t > 0) am_st<COH>(&v.remL[(size_t)t * n + k], reml);
            am_st<COH>(&v.ratL[(size_t)t * n + k], ratl);
        }
        tile[tid] = make_float4(p1[k * 3 + 0], p1[k * 3 + 1], p1[k * 3 + 2], ratl);
    }
    __syncthreads();
    const int l = cb * AM_ROWS + tid;
    const bool active = l < m;
    float x2 = 0.f, y2 = 0.f, z2 = 0.f;
    if (active) { x2 = p2[l * 3 + 0]; y2 = p2[l * 3 + 1]; z2 = p2[l * 3 + 2]; }
    float s = 0.f;
#pragma unroll 4
    for (int i = 0; i < len; ++i) {
        const float4 q = tile[i];
        const float d2 = sqdist3<FMA>(x2 - q.x, y2 - q.y, z2 - q.z);
        const float e = am_exp_level<PINNED>(d2, level);
        if constexpr (FMA) s = __builtin_fmaf(e, q.w, s);
        else s = s + e * q.w;
    }
    if (active) am_st<COH>(&v.p2[(size_t)c2 * m + l], s);
}

template <bool FMA, bool PINNED>
__global__ __launch_bounds__(AM_ROWS) void am_col_kernel(int n, int m, int t, float level, const float* __restrict__ xyz1,
                                                          const float* __restrict__ xyz2, float* __restrict__ temp) {
    __shared__ float4 tile[AM_CH];
    am_col_body<FMA, PINNED>(blockIdx.z, blockIdx.x, blockIdx.y, tile, n, m, t, level, xyz1, xyz2, temp);
}

// match[l][k] = sum_t (exp(level_t d2) ratioL_t[k]) ratioR_t[l], t ascending from 0 (the reference's `match += w` per level
// on a zeroed buffer, :16,152).  ratioR of the last level is finished here from pass 2's partials.
struct AmPartner { float x, y, z, pad; float r[AM_LEVELS]; float pad2[2]; };   // 64 bytes

template <bool FMA, bool PINNED, bool COH = false>
__device__ __forceinline__ void am_assemble_body(int cloud, int rb, int c, AmPartner* tile, int n, int m, const AmLevels& lv, const float* __restrict__ xyz1,
                                                 const float* __restrict__ xyz2, float* __restrict__ temp, float* __restrict__ match) {
    const int tid = threadIdx.x;
    const AmView v = am_view(temp, cloud, n, m);
    const float* __restrict__ p1 = xyz1 + (size_t)cloud * n * 3;
    const float* __restrict__ p2 = xyz2 + (size_t)cloud * m * 3;
    const int l0 = c * AM_ACH, len = min(AM_ACH, m - l0);
    if (tid < len) {
        const int l = l0 + tid;
        AmPartner q;
        q.x = p2[l * 3 + 0]; q.y = p2[l * 3 + 1]; q.z = p2[l * 3 + 2]; q.pad = 0.f; q.pad2[0] = q.pad2[1] = 0.f;
#pragma unroll
        for (int t = 0; t < AM_LEVELS - 1; ++t) q.r[t] = am_ld<COH>(&v.ratR[(size_t)t * m + l]);
        float remn;
        am_finish_col<COH>(v, m, am_chunks(n), AM_LEVELS - 1, l, q.r[AM_LEVELS - 1], remn);
        tile[tid] = q;
    }
    __syncthreads();
    const int k = rb * AM_ROWS + tid;
    if (k >= n) return;
    const float x1 = p1[k * 3 + 0], y1 = p1[k * 3 + 1], z1 = p1[k * 3 + 2];
    float rl[AM_LEVELS];
#pragma unroll
    for (int t = 0; t < AM_LEVELS; ++t) rl[t] = am_ld<COH>(&v.ratL[(size_t)t * n + k]);
    float* __restrict__ mt = match + (size_t)cloud * n * m + (size_t)l0 * n + k;
    for (int i = 0; i < len; ++i) {
        const AmPartner& q = tile[i];
        const float d2 = sqdist3<FMA>(q.x - x1, q.y - y1, q.z - z1);
        float acc = am_exp_level<PINNED>(d2, lv.v[0]) * rl[0] * q.r[0];
#pragma unroll
        for (int t = 1; t < AM_LEVELS; ++t) acc += am_exp_level<PINNED>(d2, lv.v[t]) * rl[t] * q.r[t];
        __builtin_nontemporal_store(acc, mt + (size_t)i * n);
    }
}

template <bool FMA, bool PINNED>
__global__ __launch_bounds__(AM_ROWS) void am_assemble_kernel(int n, int m, AmLevels lv, const float* __restrict__ xyz1,
                                                               const float* __restrict__ xyz2, float* __restrict__ temp,
                                                               float* __restrict__ match) {
    __shared__ AmPartner tile[AM_ACH];
    am_assemble_body<FMA, PINNED>(blockIdx.z, blockIdx.x, blockIdx.y, tile, n, m, lv, xyz1, xyz2, temp, match);
}

// ---- the auction inside the reference op's own temp ---------------------------------------------------------------------------
// approxmatchLauncher's caller allocates temp as [b, 2 (n + m)] floats (tf_approxmatch.cpp:164-170) and nothing more.  This form
// needs exactly that: ONE workgroup per cloud (as the reference's kernel, tf_approxmatch_g.cu:180: <<<32,512>>> over clouds), a lane
// per point, the partners streamed through an LDS tile in ascending order, so every sum is the reference's sequential chain
// (oracle/dispu_oracle.c:orc_approx_match, chunk = 0: bit for bit in DISPU_ARITH_PINNED_EXP mode) and `match` is zeroed and then
// read-modify-written once per level like the reference's `match += w` (:16,152).  Latency bound by construction (10 levels x 3 passes
// x (n / 1024) x m dependent steps per workgroup): dispu_approx_match_ws is the fast path.
constexpr int AMC_T = 1024;    // lanes per cloud, also partners per LDS tile

template <bool FMA, bool PINNED>
__global__ __launch_bounds__(AMC_T) void am_cloud_kernel(int b, int n, int m, AmLevels lv, float multiL, float multiR,
                                                          const float* __restrict__ xyz1, const float* __restrict__ xyz2,
                                                          float* __restrict__ match, float* __restrict__ temp) {
    __shared__ float4 tile[AMC_T];
    const int tid = threadIdx.x;
    for (int cloud = blockIdx.x; cloud < b; cloud += gridDim.x) {
        float* __restrict__ remL = temp + (size_t)cloud * 2 * ((size_t)n + m);
        float* __restrict__ remR = remL + n;
        float* __restrict__ ratL = remR + m;
        float* __restrict__ ratR = ratL + n;
        const float* __restrict__ p1 = xyz1 + (size_t)cloud * n * 3;
        const float* __restrict__ p2 = xyz2 + (size_t)cloud * m * 3;
        float* __restrict__ mt = match + (size_t)cloud * n * m;
        for (size_t e = tid; e < (size_t)n * m; e += AMC_T) mt[e] = 0.f;
        for (int k = tid; k < n; k += AMC_T) remL[k] = multiL;
        for (int l = tid; l < m; l += AMC_T) remR[l] = multiR;
        __syncthreads();
        for (int t = 0; t < AM_LEVELS; ++t) {
            const float level = lv.v[t];
            // pass 1 (:56-78): ratioL[k] = remainL[k] / (1e-9 + sum_l e remainR[l]), the chain starting AT 1e-9f
            for (int k0 = 0; k0 < n; k0 += AMC_T) {
                const int k = k0 + tid;
                float x1 = 0.f, y1 = 0.f, z1 = 0.f;
                if (k < n) { x1 = p1[k * 3 + 0]; y1 = p1[k * 3 + 1]; z1 = p1[k * 3 + 2]; }
                float s = 1e-9f;
                for (int l0 = 0; l0 < m; l0 += AMC_T) {
                    const int len = min(AMC_T, m - l0);
                    __syncthreads();
                    if (tid < len) tile[tid] = make_float4(p2[(l0 + tid) * 3 + 0], p2[(l0 + tid) * 3 + 1], p2[(l0 + tid) * 3 + 2], remR[l0 + tid]);
                    __syncthreads();
                    if (k < n)
                        for (int i = 0; i < len; ++i) {
                            const float4 q = tile[i];
                            const float e = am_exp_level<PINNED>(sqdist3<FMA>(q.x - x1, q.y - y1, q.z - z1), level);
                            if constexpr (FMA) s = __builtin_fmaf(e, q.w, s);
                            else s = s + e * q.w;
                        }
                }
                if (k < n) ratL[k] = remL[k] / s;
            }
            __syncthreads();
            // pass 2 (:80-112): sumr = (sum_k e ratioL[k]) remainR[l]; ratioR = min(remainR / (sumr + 1e-9), 1) remainR; remainR -= sumr, >= 0
            for (int l0 = 0; l0 < m; l0 += AMC_T) {
                const int l = l0 + tid;
                float x2 = 0.f, y2 = 0.f, z2 = 0.f;
                if (l < m) { x2 = p2[l * 3 + 0]; y2 = p2[l * 3 + 1]; z2 = p2[l * 3 + 2]; }
                float s = 0.f;
                for (int k0 = 0; k0 < n; k0 += AMC_T) {
                    const int len = min(AMC_T, n - k0);
                    __syncthreads();
                    if (tid < len) tile[tid] = make_float4(p1[(k0 + tid) * 3 + 0], p1[(k0 + tid) * 3 + 1], p1[(k0 + tid) * 3 + 2], ratL[k0 + tid]);
                    __syncthreads();
                    if (l < m)
                        for (int i = 0; i < len; ++i) {
                            const float4 q = tile[i];
                            const float e = am_exp_level<PINNED>(sqdist3<FMA>(x2 - q.x, y2 - q.y, z2 - q.z), level);
                            if constexpr (FMA) s = __builtin_fmaf(e, q.w, s);
                            else s = s + e * q.w;
                        }
                }
                if (l < m) {
                    const float rr = remR[l];
                    const float sumr = s * rr;
                    const float consumption = fminf(rr / (sumr + 1e-9f), 1.0f);
                    ratR[l] = consumption * rr;
                    remR[l] = fmaxf(0.0f, rr - sumr);
                }
            }
            __syncthreads();
            // pass 3 (:114-160): w = e ratioL[k] ratioR[l]; match[l][k] += w; remainL[k] = max(0, remainL[k] - sum_l w)
            for (int k0 = 0; k0 < n; k0 += AMC_T) {
                const int k = k0 + tid;
                float x1 = 0.f, y1 = 0.f, z1 = 0.f, rl = 0.f;
                if (k < n) { x1 = p1[k * 3 + 0]; y1 = p1[k * 3 + 1]; z1 = p1[k * 3 + 2]; rl = ratL[k]; }
                float s = 0.f;
                for (int l0 = 0; l0 < m; l0 += AMC_T) {
                    const int len = min(AMC_T, m - l0);
                    __syncthreads();
                    if (tid < len) tile[tid] = make_float4(p2[(l0 + tid) * 3 + 0], p2[(l0 + tid) * 3 + 1], p2[(l0 + tid) * 3 + 2], ratR[l0 + tid]);
                    __syncthreads();
                    if (k < n) {
                        float* __restrict__ col = mt + (size_t)l0 * n + k;
                        for (int i = 0; i < len; ++i) {
                            const float4 q = tile[i];
                            const float w = am_exp_level<PINNED>(sqdist3<FMA>(q.x - x1, q.y - y1, q.z - z1), level) * rl * q.w;
                            col[(size_t)i * n] += w;
                            s += w;
                        }
                    }
                }
                if (k < n) remL[k] = fmaxf(0.0f, remL[k] - s);
            }
            __syncthreads();
        }
    }
}

// (Rounds 3 - 4 also carried the whole auction as ONE persistent launch with per-cloud software barriers -- bit-identical and slower: an
// in-kernel barrier across XCDs costs more than a kernel boundary; removed in round 5, see profiles/EXPERIMENTS.md.)


// ---- match_cost -----------------------------------------------------------------------------------------------------------
// cost[b] = sum_{k,l} sqrt(d2(k,l)) * match[l*n+k]   (matchcost, tf_approxmatch_g.cu:183-225: one block per cloud).
// Here `match` is streamed once by (256 k) x (MC_CH l) tiles -- lane = k, so every load is a coalesced 1 KB row
// segment -- each lane sums its l's sequentially, the workgroup reduces its 256 lane sums in a fixed order (wave butterfly,
// then waves ascending) to ONE partial, and a second launch adds the partials of a cloud in ascending (row block, chunk)
// order.  Deterministic; the association differs from the reference's 512-lane tree (tolerance-tested, 1e-5).
constexpr int MC_CH = 128;      // partners per tile at most; halved until a launch has >= 2048 workgroups (a lane keeps
                                // 16 loads of 256 B per wave in flight: ~64 KB per CU are needed to cover the HBM latency)

static inline int mc_chunk(int b, int n, int m) {
    int ch = MC_CH;
    const long rb = (n + 255) / 256;
    while (ch > 16 && rb * ((m + ch - 1) / ch) * b < 2048) ch >>= 1;
    return ch;
}

template <bool FMA>
__global__ __launch_bounds__(AM_ROWS) void match_cost_tile_kernel(int n, int m, int ch, const float* __restrict__ xyz1,
                                                                   const float* __restrict__ xyz2,
                                                                   const float* __restrict__ match,
                                                                   float* __restrict__ part) {
    __shared__ float4 tile[MC_CH];
    __shared__ float wsum[AM_ROWS / kWave];
    const int cloud = blockIdx.z, c = blockIdx.y, rb = blockIdx.x, tid = threadIdx.x;
    const float* __restrict__ p1 = xyz1 + (size_t)cloud * n * 3;
    const float* __restrict__ p2 = xyz2 + (size_t)cloud * m * 3;
    const int l0 = c * ch, len = min(ch, m - l0);
    if (tid < len) tile[tid] = make_float4(p2[(l0 + tid) * 3 + 0], p2[(l0 + tid) * 3 + 1], p2[(l0 + tid) * 3 + 2], 0.f);
    __syncthreads();
    const int k = rb * AM_ROWS + tid;
    float s = 0.f;
    if (k < n) {
        const float x1 = p1[k * 3 + 0], y1 = p1[k * 3 + 1], z1 = p1[k * 3 + 2];
        const float* __restrict__ mt = match + (size_t)cloud * n * m + (size_t)l0 * n + k;
#pragma unroll 8
        for (int i = 0; i < len; ++i) {
            const float4 q = tile[i];
            const float w = __builtin_nontemporal_load(mt + (size_t)i * n);
            // v_sqrt_f32 (1 ulp) instead of the correctly rounded sqrtf: the ~20-instruction fix-up sequence made this
            // HBM-bound stream VALU-bound; the cost is tolerance-checked (1e-5, north star), not bit-compared
            const float d = __builtin_amdgcn_sqrtf(sqdist3<FMA>(q.x - x1, q.y - y1, q.z - z1));
            if constexpr (FMA) s = __builtin_fmaf(d, w, s);
            else s = s + d * w;
        }
    }
    s = wave_sum_f32(s);
    if ((tid & (kWave - 1)) == 0) wsum[tid / kWave] = s;
    __syncthreads();
    if (tid == 0) {
        float r = wsum[0];
        for (int w = 1; w < AM_ROWS / kWave; ++w) r += wsum[w];
        part[((size_t)cloud * gridDim.x + rb) * gridDim.y + c] = r;
    }
}

__global__ __launch_bounds__(256) void match_cost_final_kernel(int nparts, const float* __restrict__ part, float* __restrict__ cost) {
    __shared__ float wsum[256 / kWave];
    const float* __restrict__ p = part + (size_t)blockIdx.x * nparts;
    float s = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) s += p[i];
    s = wave_sum_f32(s);
    if ((threadIdx.x & (kWave - 1)) == 0) wsum[threadIdx.x / kWave] = s;
    __syncthreads();
    if (threadIdx.x == 0) cost[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

// Scratch-free form for the reference's launcher signature (dispu_match_cost: no scratch argument): ONE workgroup per cloud
// walks the same (256 k) x (128 l) tiles in (row block, chunk) order and adds each tile's workgroup sum to a running total --
// the association of match_cost_tile_kernel + match_cost_final_kernel for a cloud with <= 256 tiles is different (there the
// partials are summed by lanes), so the two entries agree to rounding, not bit for bit.  HBM-latency bound (one workgroup
// per cloud, like the reference's kernel); dispu_match_cost_ws is the fast path.
template <bool FMA>
__global__ __launch_bounds__(AM_ROWS) void match_cost_cloud_kernel(int n, int m, const float* __restrict__ xyz1,
                                                                    const float* __restrict__ xyz2,
                                                                    const float* __restrict__ match, float* __restrict__ cost) {
    __shared__ float4 tile[MC_CH];
    __shared__ float wsum[AM_ROWS / kWave];
    const int cloud = blockIdx.x, tid = threadIdx.x;
    const float* __restrict__ p1 = xyz1 + (size_t)cloud * n * 3;
    const float* __restrict__ p2 = xyz2 + (size_t)cloud * m * 3;
    float total = 0.f;
    for (int k0 = 0; k0 < n; k0 += AM_ROWS) {
        const int k = k0 + tid;
        float x1 = 0.f, y1 = 0.f, z1 = 0.f;
        if (k < n) { x1 = p1[k * 3 + 0]; y1 = p1[k * 3 + 1]; z1 = p1[k * 3 + 2]; }
        for (int l0 = 0; l0 < m; l0 += MC_CH) {
            const int len = min(MC_CH, m - l0);
            __syncthreads();
            if (tid < len) tile[tid] = make_float4(p2[(l0 + tid) * 3 + 0], p2[(l0 + tid) * 3 + 1], p2[(l0 + tid) * 3 + 2], 0.f);
            __syncthreads();
            float s = 0.f;
            if (k < n) {
                const float* __restrict__ mt = match + (size_t)cloud * n * m + (size_t)l0 * n + k;
#pragma unroll 8
                for (int i = 0; i < len; ++i) {
                    const float4 q = tile[i];
                    const float d = __builtin_amdgcn_sqrtf(sqdist3<FMA>(q.x - x1, q.y - y1, q.z - z1));
                    const float w = __builtin_nontemporal_load(mt + (size_t)i * n);
                    if constexpr (FMA) s = __builtin_fmaf(d, w, s);
                    else s = s + d * w;
                }
            }
            s = wave_sum_f32(s);
            if ((tid & (kWave - 1)) == 0) wsum[tid / kWave] = s;
            __syncthreads();
            float r = wsum[0];
            for (int w = 1; w < AM_ROWS / kWave; ++w) r += wsum[w];
            total += r;
        }
    }
    if (tid == 0) cost[cloud] = total;
}

// grad1[b,k,:] = sum_l match[l*n+k] * (p1_k - p2_l) * rsqrt(max(d2,1e-20))   (matchcostgrad1, :270-291)
// same tiling as the cost: partial sums per (chunk, k), combined in ascending chunk order by match_grad1_combine_kernel.
template <bool FMA>
__global__ __launch_bounds__(AM_ROWS) void match_grad1_tile_kernel(int n, int m, int ch, const float* __restrict__ xyz1,
                                                                    const float* __restrict__ xyz2,
                                                                    const float* __restrict__ match,
                                                                    float* __restrict__ part) {
    __shared__ float4 tile[MC_CH];
    const int cloud = blockIdx.z, c = blockIdx.y, rb = blockIdx.x, tid = threadIdx.x;
    const float* __restrict__ p1 = xyz1 + (size_t)cloud * n * 3;
    const float* __restrict__ p2 = xyz2 + (size_t)cloud * m * 3;
    const int l0 = c * ch, lend = min(m, l0 + ch);          // ch > MC_CH (the scratch-free entry: ch = m): sub-tiles, same lane order
    const int k = rb * AM_ROWS + tid;
    float x1 = 0.f, y1 = 0.f, z1 = 0.f;
    if (k < n) { x1 = p1[k * 3 + 0]; y1 = p1[k * 3 + 1]; z1 = p1[k * 3 + 2]; }
    float gx = 0.f, gy = 0.f, gz = 0.f;
    for (int lt = l0; lt < lend; lt += MC_CH) {
        const int len = min(MC_CH, lend - lt);
        if (lt != l0) __syncthreads();
        if (tid < len) tile[tid] = make_float4(p2[(lt + tid) * 3 + 0], p2[(lt + tid) * 3 + 1], p2[(lt + tid) * 3 + 2], 0.f);
        __syncthreads();
        if (k < n) {
            const float* __restrict__ mt = match + (size_t)cloud * n * m + (size_t)lt * n + k;
#pragma unroll 8
            for (int i = 0; i < len; ++i) {
                const float4 q = tile[i];
                const float ex = x1 - q.x, ey = y1 - q.y, ez = z1 - q.z;
                const float d = __builtin_nontemporal_load(mt + (size_t)i * n) * rsqrtf(fmaxf(sqdist3<FMA>(ex, ey, ez), 1e-20f));
                if constexpr (FMA) { gx = __builtin_fmaf(ex, d, gx); gy = __builtin_fmaf(ey, d, gy); gz = __builtin_fmaf(ez, d, gz); }
                else { gx += ex * d; gy += ey * d; gz += ez * d; }
            }
        }
    }
    if (k < n) {
        float* g = part + (((size_t)cloud * gridDim.y + c) * n + k) * 3;
        g[0] = gx; g[1] = gy; g[2] = gz;
    }
}

__global__ void match_grad1_combine_kernel(int n, int nc, const float* __restrict__ part, float* __restrict__ grad) {
    const int cloud = blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;          // element of [n][3]
    if (e >= n * 3) return;
    const float* __restrict__ p = part + (size_t)cloud * nc * n * 3 + e;
    float s = p[0];
    for (int c = 1; c < nc; ++c) s += p[(size_t)c * n * 3];
    grad[(size_t)cloud * n * 3 + e] = s;
}

// grad2[b,l,:] = sum_k match[l*n+k] * (p2_l - p1_k) * rsqrt(max(d2,1e-20))   (matchcostgrad2, :229-269)
// one wave per l, lanes stride over k (row l of match is contiguous), wave butterfly sum (reference: 256-thread tree).
template <bool FMA>
__global__ __launch_bounds__(256) void match_grad2_kernel(int n, int m, const float* __restrict__ xyz1,
                                                           const float* __restrict__ xyz2, const float* __restrict__ match,
                                                           float* __restrict__ grad) {
    const int cloud = blockIdx.y;
    const float* __restrict__ p1 = xyz1 + (size_t)cloud * n * 3;
    const float* __restrict__ p2 = xyz2 + (size_t)cloud * m * 3;
    const float* __restrict__ mt = match + (size_t)cloud * n * m;
    const int lane = threadIdx.x & (kWave - 1);
    const int l = blockIdx.x * (256 / kWave) + threadIdx.x / kWave;
    if (l >= m) return;  // wave-uniform
    const float x2 = p2[l * 3 + 0], y2 = p2[l * 3 + 1], z2 = p2[l * 3 + 2];
    float gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll 4
    for (int k = lane; k < n; k += kWave) {
        const float ex = x2 - p1[k * 3 + 0], ey = y2 - p1[k * 3 + 1], ez = z2 - p1[k * 3 + 2];
        const float d = __builtin_nontemporal_load(mt + (size_t)l * n + k) * rsqrtf(fmaxf(sqdist3<FMA>(ex, ey, ez), 1e-20f));
        if constexpr (FMA) { gx = __builtin_fmaf(ex, d, gx); gy = __builtin_fmaf(ey, d, gy); gz = __builtin_fmaf(ez, d, gz); }
        else { gx += ex * d; gy += ey * d; gz += ez * d; }
    }
    gx = wave_sum_f32(gx); gy = wave_sum_f32(gy); gz = wave_sum_f32(gz);
    if (lane == 0) { float* g = grad + ((size_t)cloud * m + l) * 3; g[0] = gx; g[1] = gy; g[2] = gz; }
}

// level_t = -(4^(7 - t)), t = 0 .. 8, and 0 for the last (tf_approxmatch_g.cu:50-53: j = 7 .. -2, level = -powf(4, j), 0 at j = -2)
static AmLevels am_levels() {
    AmLevels lv;
    for (int t = 0; t < AM_LEVELS; ++t) {
        const int j = 7 - t;
        float level = 0.0f;
        if (j != -2) {
            level = -1.0f;
            for (int q = 0; q < (j < 0 ? -j : j); ++q) level = (j < 0) ? level * 0.25f : level * 4.0f;  // -(4^j), exact
        }
        lv.v[t] = level;
    }
    return lv;
}

template <bool FMA, bool PINNED>
static int run_approx_match(int b, int n, int m, const float* xyz1, const float* xyz2, float* match, float* temp,
                            hipStream_t s) {
    const float multiL = (n >= m) ? 1.0f : (float)(m / n);
    const float multiR = (n >= m) ? (float)(n / m) : 1.0f;
    const AmLevels lv = am_levels();
    const dim3 blk(AM_ROWS);
    const dim3 grow((n + AM_ROWS - 1) / AM_ROWS, am_chunks(m), b), gcol((m + AM_ROWS - 1) / AM_ROWS, am_chunks(n), b);
    hipLaunchKernelGGL(am_init_kernel, dim3(8, b), dim3(256), 0, s, n, m, multiL, multiR, temp);
    hipLaunchKernelGGL((am_row_kernel<true, FMA, PINNED>), grow, blk, 0, s, n, m, 0, 0.f, lv.v[0], xyz1, xyz2, temp);
    for (int t = 0; t < AM_LEVELS; ++t) {
        hipLaunchKernelGGL((am_col_kernel<FMA, PINNED>), gcol, blk, 0, s, n, m, t, lv.v[t], xyz1, xyz2, temp);
        if (t + 1 < AM_LEVELS)
            hipLaunchKernelGGL((am_row_kernel<false, FMA, PINNED>), grow, blk, 0, s, n, m, t, lv.v[t], lv.v[t + 1], xyz1, xyz2, temp);
    }
    const dim3 gasm((n + AM_ROWS - 1) / AM_ROWS, (m + AM_ACH - 1) / AM_ACH, b);
    hipLaunchKernelGGL((am_assemble_kernel<FMA, PINNED>), gasm, blk, 0, s, n, m, lv, xyz1, xyz2, temp, match);
    return (int)hipGetLastError();
}

}  // namespace dispu

using namespace dispu;

DISPU_EXPORT size_t dispu_approx_match_scratch_bytes(int b, int n, int m) {
    if (b <= 0 || n <= 0 || m <= 0) return 0;
    return sizeof(float) * (size_t)b * am_cloud_floats(n, m) + sizeof(unsigned) * ((size_t)b + 1);   // + per-cloud stage counters, fail flag
}

static int am_check_args(int b, int n, int m) {
    if (b < 0 || n <= 0 || m <= 0) return (int)hipErrorInvalidValue;
    if (b > 65535 || (size_t)n * 3 > 0x7fffffffull || (size_t)m * 3 > 0x7fffffffull) return (int)hipErrorInvalidValue;
    return 0;
}

// The fast path: 2-D tiled passes, `temp` = dispu_approx_match_scratch_bytes(b,n,m) bytes; a smaller scratch is refused, untouched.
DISPU_EXPORT int dispu_approx_match_ws(int b, int n, int m, const float* xyz1, const float* xyz2, float* match,
                                       float* temp, size_t temp_bytes, int arith, void* stream) {
    if (const int e = am_check_args(b, n, m)) return e;
    if (b == 0) return 0;
    if (!temp || temp_bytes < dispu_approx_match_scratch_bytes(b, n, m)) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    const bool fma = (arith & DISPU_ARITH_CONTRACT) != 0, pin = (arith & DISPU_ARITH_PINNED_EXP) != 0;
    if (fma && pin) return run_approx_match<true, true>(b, n, m, xyz1, xyz2, match, temp, s);
    if (fma) return run_approx_match<true, false>(b, n, m, xyz1, xyz2, match, temp, s);
    if (pin) return run_approx_match<false, true>(b, n, m, xyz1, xyz2, match, temp, s);
    return run_approx_match<false, false>(b, n, m, xyz1, xyz2, match, temp, s);
}

// approxmatchLauncher(b,n,m,xyz1,xyz2,match,temp) (tf_approxmatch.cpp:141): `temp` is the op's own [b, 2 (n + m)] float allocation
// (:164-170) and nothing behind it is touched.  One workgroup per cloud, the reference's sequential association.
DISPU_EXPORT int dispu_approx_match(int b, int n, int m, const float* xyz1, const float* xyz2, float* match,
                                    float* temp, int arith, void* stream) {
    if (const int e = am_check_args(b, n, m)) return e;
    if (b == 0) return 0;
    if (!temp) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    const bool fma = (arith & DISPU_ARITH_CONTRACT) != 0, pin = (arith & DISPU_ARITH_PINNED_EXP) != 0;
    const float multiL = (n >= m) ? 1.0f : (float)(m / n);
    const float multiR = (n >= m) ? (float)(n / m) : 1.0f;
    const AmLevels lv = am_levels();
    const dim3 grid(b), blk(AMC_T);
    if (fma && pin) hipLaunchKernelGGL((am_cloud_kernel<true, true>), grid, blk, 0, s, b, n, m, lv, multiL, multiR, xyz1, xyz2, match, temp);
    else if (fma) hipLaunchKernelGGL((am_cloud_kernel<true, false>), grid, blk, 0, s, b, n, m, lv, multiL, multiR, xyz1, xyz2, match, temp);
    else if (pin) hipLaunchKernelGGL((am_cloud_kernel<false, true>), grid, blk, 0, s, b, n, m, lv, multiL, multiR, xyz1, xyz2, match, temp);
    else hipLaunchKernelGGL((am_cloud_kernel<false, false>), grid, blk, 0, s, b, n, m, lv, multiL, multiR, xyz1, xyz2, match, temp);
    return (int)hipGetLastError();
}

DISPU_EXPORT size_t dispu_match_cost_scratch_bytes(int b, int n, int m) {
    if (b <= 0 || n <= 0 || m <= 0) return 0;
    const int ch = mc_chunk(b, n, m);
    const size_t rb = (n + AM_ROWS - 1) / AM_ROWS, nc = (m + ch - 1) / ch;
    return sizeof(float) * (size_t)b * rb * nc;
}

DISPU_EXPORT int dispu_match_cost_ws(int b, int n, int m, const float* xyz1, const float* xyz2, const float* match,
                                     float* cost, float* scratch, int arith, void* stream) {
    if (b < 0 || n <= 0 || m <= 0) return (int)hipErrorInvalidValue;
    if (b == 0) return 0;
    if (!scratch) return (int)hipErrorInvalidValue;
    if (b > 65535) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    const int ch = mc_chunk(b, n, m);
    const dim3 grid((n + AM_ROWS - 1) / AM_ROWS, (m + ch - 1) / ch, b);
    if ((arith & DISPU_ARITH_CONTRACT))
        hipLaunchKernelGGL((match_cost_tile_kernel<true>), grid, dim3(AM_ROWS), 0, s, n, m, ch, xyz1, xyz2, match, scratch);
    else
        hipLaunchKernelGGL((match_cost_tile_kernel<false>), grid, dim3(AM_ROWS), 0, s, n, m, ch, xyz1, xyz2, match, scratch);
    hipLaunchKernelGGL(match_cost_final_kernel, dim3(b), dim3(256), 0, s, (int)(grid.x * grid.y), scratch, cost);
    return (int)hipGetLastError();
}

DISPU_EXPORT size_t dispu_match_cost_grad_scratch_bytes(int b, int n, int m) {
    if (b <= 0 || n <= 0 || m <= 0) return 0;
    const int ch = mc_chunk(b, n, m);
    return sizeof(float) * (size_t)b * ((m + ch - 1) / ch) * (size_t)n * 3;
}

DISPU_EXPORT int dispu_match_cost_grad_ws(int b, int n, int m, const float* xyz1, const float* xyz2, const float* match,
                                          float* grad1, float* grad2, float* scratch, int arith, void* stream) {
    if (b < 0 || n <= 0 || m <= 0) return (int)hipErrorInvalidValue;
    if (b == 0) return 0;
    if (!scratch) return (int)hipErrorInvalidValue;
    if (b > 65535) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    const int ch = mc_chunk(b, n, m);
    const int nc = (m + ch - 1) / ch;
    const dim3 g1((n + AM_ROWS - 1) / AM_ROWS, nc, b), g2((m + 3) / 4, b), gc((n * 3 + 255) / 256, b);
    if ((arith & DISPU_ARITH_CONTRACT)) {
        hipLaunchKernelGGL((match_grad1_tile_kernel<true>), g1, dim3(AM_ROWS), 0, s, n, m, ch, xyz1, xyz2, match, scratch);
        hipLaunchKernelGGL((match_grad2_kernel<true>), g2, dim3(256), 0, s, n, m, xyz1, xyz2, match, grad2);
    } else {
        hipLaunchKernelGGL((match_grad1_tile_kernel<false>), g1, dim3(AM_ROWS), 0, s, n, m, ch, xyz1, xyz2, match, scratch);
        hipLaunchKernelGGL((match_grad2_kernel<false>), g2, dim3(256), 0, s, n, m, xyz1, xyz2, match, grad2);
    }
    hipLaunchKernelGGL(match_grad1_combine_kernel, gc, dim3(256), 0, s, n, nc, scratch, grad1);
    return (int)hipGetLastError();
}

// ---- the reference's launcher signatures (no scratch argument; ABI version 1 of this library had exactly these) ------------
// matchcostLauncher(b,n,m,xyz1,xyz2,match,out) / matchcostgradLauncher(b,n,m,xyz1,xyz2,match,grad1,grad2)
// (tf_approxmatch.cpp:142-143).  Correct without scratch: the cost runs one workgroup per cloud, grad1 with a single partner
// chunk (its "partial" is the result, written straight into grad1).  The *_ws entries above are the fast paths.
DISPU_EXPORT int dispu_match_cost(int b, int n, int m, const float* xyz1, const float* xyz2, const float* match, float* cost,
                                  int arith, void* stream) {
    if (b < 0 || n <= 0 || m <= 0) return (int)hipErrorInvalidValue;
    if (b == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if ((arith & DISPU_ARITH_CONTRACT))
        hipLaunchKernelGGL((match_cost_cloud_kernel<true>), dim3(b), dim3(AM_ROWS), 0, s, n, m, xyz1, xyz2, match, cost);
    else
        hipLaunchKernelGGL((match_cost_cloud_kernel<false>), dim3(b), dim3(AM_ROWS), 0, s, n, m, xyz1, xyz2, match, cost);
    return (int)hipGetLastError();
}

DISPU_EXPORT int dispu_match_cost_grad(int b, int n, int m, const float* xyz1, const float* xyz2, const float* match,
                                       float* grad1, float* grad2, int arith, void* stream) {
    if (b < 0 || n <= 0 || m <= 0) return (int)hipErrorInvalidValue;
    if (b == 0) return 0;
    if (b > 65535) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    const dim3 g1((n + AM_ROWS - 1) / AM_ROWS, 1, b), g2((m + 3) / 4, b);
    if ((arith & DISPU_ARITH_CONTRACT)) {
        hipLaunchKernelGGL((match_grad1_tile_kernel<true>), g1, dim3(AM_ROWS), 0, s, n, m, m, xyz1, xyz2, match, grad1);
        hipLaunchKernelGGL((match_grad2_kernel<true>), g2, dim3(256), 0, s, n, m, xyz1, xyz2, match, grad2);
    } else {
        hipLaunchKernelGGL((match_grad1_tile_kernel<false>), g1, dim3(AM_ROWS), 0, s, n, m, m, xyz1, xyz2, match, grad1);
        hipLaunchKernelGGL((match_grad2_kernel<false>), g2, dim3(256), 0, s, n, m, xyz1, xyz2, match, grad2);
    }
    return (int)hipGetLastError();
}
