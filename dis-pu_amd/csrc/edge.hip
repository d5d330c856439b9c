// dense_conv (Common/ops.py:1897-1915) + get_edge_feature (:1856-1877) on the fp32 matrix cores.
//
// For every point p and each of its 16 feature-space neighbours j (one "pair row"):
//   y0 = [F_p, F_j - F_p] -> l0 = relu(y0.W0 + b0);  y1 = [l0, F_p] -> l1 = relu(y1.W1 + b1);
//   y2 = [l1, l0, F_p]    -> l2 = y2.W2 + b2;        out[p] = max_j [l2 | l1 | l0 | F_p]      (72 + C channels)
//
// MI355X mapping.  A wave owns 32 pair rows (2 points x 16 neighbours) and computes the TRANSPOSED products
// D[channel][row] = sum_k W^T[channel][k] * y[row][k] with v_mfma_f32_32x32x2_f32, so the pair row is the
// MFMA *column*: lane (row = lane & 31, h = lane >> 5) feeds its own row's element k = 2s + h at step s and
// receives, in the C/D layout, accumulator register r <-> A-row (r&3) + 8(r>>2) + 4h.  The weight rows are
// loaded PERMUTED so that this A-row is output channel 2r + h: the 24 outputs of a layer then sit in the lane as
// "channel 2r + h in register r" - exactly the operand layout the next layer's k-loop needs.  The three chained
// layers therefore run register-to-register (no LDS round trip, no concat), 132 MFMAs per 32 rows at C = 48.
// Weight fragments live in LDS, one conflict-free ds_read_b32 per MFMA.  The max over the 16 neighbours is a
// DPP row reduction (a DPP row of 16 lanes is one point).  k ascends through the concatenated input exactly as
// in oracle/generator.py, so the result is bit-identical to the fmaf-chain restatement.
#include "common.h"
#include "knn_select.h"


namespace dispu {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CTRL>
__device__ __forceinline__ float edge_dpp(float identity, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(identity), __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float edge_row16_max(float v) {   // valid in lane 15 of each 16-lane DPP row
    // row_ror (rotate within the 16-lane row) has no invalid source lane: one v_max_f32_dpp per step, and every
    // lane of the row ends up with the row maximum
    v = fmaxf(v, edge_dpp<0x121>(v, v));
    v = fmaxf(v, edge_dpp<0x122>(v, v));
    v = fmaxf(v, edge_dpp<0x124>(v, v));
    v = fmaxf(v, edge_dpp<0x128>(v, v));
    return v;
}

// LDS image of one layer's A operand: frag[s*64 + lane] = W[k = 2s + (lane>>5)][channel(lane & 31)], 0 for padding
// (lanes whose A-row would be channel >= 24).  W [K][24] is read once, linearly (coalesced float4), and scattered
// into place through LDS: a per-element gather from global memory was 33 vector-memory instructions per thread.
__device__ __forceinline__ void edge_fill_frag(float* frag, const float* __restrict__ W, int K, int tid, int nthreads) {
    const int nf4 = K * 24 / 4;                                  // W rows are 96 bytes: every float4 stays inside one row
    for (int e = tid; e < nf4; e += nthreads) {
        const float4 v = *reinterpret_cast<const float4*>(W + e * 4);
        const int k = (e * 4) / 24, ch0 = (e * 4) - k * 24;
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ch = ch0 + u, r = ch >> 1, hp = ch & 1;
            const int i = (r & 3) | (hp << 2) | ((r >> 2) << 3);
            frag[(k >> 1) * 64 + (k & 1) * 32 + i] = vv[u];
        }
    }
    for (int e = tid; e < K * 8; e += nthreads) {                // padding lanes 24..31 of every (s, half)
        const int sh = e >> 3;                                   // s * 2 + half
        frag[sh * 32 + 24 + (e & 7)] = 0.f;
    }
}

// one float4 (index e of W [K][24] read linearly) into its fragment slots; edge_zero_pad: the padding lanes
__device__ __forceinline__ void edge_scatter_f4(float* frag, int e, const float4 v) {
    const int k = (e * 4) / 24, ch0 = (e * 4) - k * 24;
    const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int ch = ch0 + u, r = ch >> 1, hp = ch & 1;
        const int i = (r & 3) | (hp << 2) | ((r >> 2) << 3);
        frag[(k >> 1) * 64 + (k & 1) * 32 + i] = vv[u];
    }
}

// LDSF = true: one workgroup works inside ONE cloud and first copies that cloud's [n_per_cloud, C] feature block into
// LDS; the 2 x C/4 float4 row gathers per lane and group then are ds_read_b128 instead of global loads.  A wave that
// shares its SIMD's issue with its own 64-cycle MFMAs gets a vector-memory instruction out only every ~300 cycles
// (tools/micro/gemm_lab.hip), and the 24 gathers per group cost more than the 132 MFMAs they feed.
// LDSF workgroups have 8 waves (two per SIMD: one wave's DPP max / store epilogue overlaps the other's MFMA chain; the
// rows come from LDS at the top of each group, so no prefetch registers are needed and 256 registers per wave suffice).
// PRE (round 4, LDSF only): layer 0's input is [F_p (C), F_j - F_p (C)], so the first C terms of its fmaf chain depend on the POINT,
// not on the pair.  They are evaluated once per point -- one MFMA pass over 32 points as columns per 32 points of the workgroup,
// instead of C/2 steps in every 32-pair tile, where 16 columns repeat the same point -- parked in LDS, and a pair tile starts its
// layer-0 accumulators from them and continues the chain with the F_j - F_p terms: the same sequence of fused multiply-adds per
// output, bit for bit, with 24 of 132 (C = 48) / 12 of 84 (C = 24) MFMAs per tile gone.
// KNN (round 4, with PRE): the workgroup also FINDS the neighbours.  knn_feat_wave_kernel staged the same cloud into LDS, wrote
// [points][k] indices to HBM, and this kernel read them back after a launch boundary; here the 32 (or 16) points of the workgroup are
// the queries: their 16 x 16 dot tiles against the LDS copy of the cloud run on v_mfma_f32_16x16x4_f32 (the same instruction, operand
// order and distance expression as knn_feat_wave_kernel: bit-identical words), every wave selects the k smallest for the points of
// ITS OWN pair groups (threshold prefilter + rank, knn_select.h) into an LDS table, and the pair loop takes the neighbour ids from
// there -- no workgroup barrier between selection and convolution.  idx_out != nullptr also leaves the [points][ksel] table in HBM
// (the training step's backward pass reads it).
constexpr int EK_NP = 64 * 4 + 4;              // row stride of the distance-word matrix (n <= 256 candidates)
constexpr int EK_SCRATCH = 128 + 4;            // u64 prefilter slots per wave
constexpr int EK_KLD = 20;                     // row stride of the neighbour table (ksel <= 20)
constexpr int EK_MAXQ = 32;                    // points (= queries) per workgroup
template <int C, bool LDSF, bool PRE = false, bool KNN = false>
__global__ __launch_bounds__(LDSF ? 512 : 256) void edge_dense_conv_mfma_kernel(int npoints, int n_per_cloud, const float* __restrict__ F,
                                                                    long ldf, const int* __restrict__ idx, int ldi, int ioff,
                                                                    const float* __restrict__ W0, const float* __restrict__ b0,
                                                                    const float* __restrict__ W1, const float* __restrict__ b1,
                                                                    const float* __restrict__ W2, const float* __restrict__ b2,
                                                                    float* __restrict__ Y, long ldy, int* __restrict__ idx_out, int ksel,
                                                                    const float* __restrict__ Wp, const float* __restrict__ bp, int k_old,
                                                                    float* __restrict__ Pout, long ldp, const float* __restrict__ xyz,
                                                                    const float* __restrict__ Wl, const float* __restrict__ bl,
                                                                    float* __restrict__ Lout, long ldl) {
    constexpr int G = 24, H = C / 2, K0 = 2 * C, K1 = G + C, K2 = 2 * G + C;
    constexpr int S0 = K0 / 2, S1 = K1 / 2, S2 = K2 / 2;
    extern __shared__ __attribute__((aligned(16))) float edge_lds[];
#ifdef EDGE_STAMPS
    const unsigned long long e_k0 = __builtin_readcyclecounter();
#endif
    float* frag = edge_lds;                                      // (S0 + S1 + S2) * 64 weight fragments
    float* stage = frag + (S0 + S1 + S2) * 64;                   // [4 waves][2 points][72 + C] output staging
    // (KNN: one staging row per point of the workgroup -- the rows stay for the bottleneck conv at the end)
    constexpr int OWS = 3 * G + C + (KNN ? 4 : 0);               // staging row stride (KNN: + 4 floats, the tail reads 16 rows at once)
    float* fl = stage + (KNN ? EK_MAXQ : (LDSF ? 8 : 4) * 2) * OWS;   // LDSF: [n_per_cloud][C + 4] features of this cloud
    static_assert(!PRE || LDSF, "the per-point prefix reads the LDS copy of the cloud");
    static_assert(!KNN || PRE, "the fused neighbour search is built on the LDS-resident, prefixed variant");
    const bool vec_store = ((ldy & 3) == 0) && ((((uintptr_t)Y) & 15) == 0);
    constexpr int FLD = C + 4;                                   // row stride 52 / 28 floats: 16-byte aligned, spreads the banks
    float* f0 = frag;
    float* f1 = frag + S0 * 64;
    float* f2 = f1 + S1 * 64;
    constexpr int NWAVE = LDSF ? 8 : 4;
    if constexpr (!LDSF) {
        edge_fill_frag(f0, W0, K0, threadIdx.x, 64 * NWAVE);
        edge_fill_frag(f1, W1, K1, threadIdx.x, 64 * NWAVE);
        edge_fill_frag(f2, W2, K2, threadIdx.x, 64 * NWAVE);
    }
    // LDSF geometry: blockIdx.x = cloud * parts + part; the workgroup handles point groups [g_lo, g_hi) of its cloud
    int cloud0 = 0, g_lo = 0, g_hi = (npoints + 1) / 2, gstep0 = gridDim.x * 4, gfirst = blockIdx.x * 4, per_wg = 0;
    if constexpr (LDSF) {
        const int parts = gridDim.y;                             // workgroups per cloud
        const int cloud = blockIdx.x, part = blockIdx.y;
        cloud0 = cloud * n_per_cloud;
        const int gpc = n_per_cloud / 2;                         // groups per cloud (n_per_cloud even)
        const int per = (gpc + parts - 1) / parts;
        per_wg = per;
        g_lo = cloud * gpc + part * per;
        g_hi = min(cloud * gpc + gpc, g_lo + per);
        gstep0 = NWAVE;
        gfirst = g_lo;
        // Staging, every load of the workgroup in flight at once: the three weight matrices (read linearly) and the first 8 float4 per
        // thread of the cloud are requested before anything is written to LDS.  (One load -> wait -> store round per loop iteration,
        // five to ten L2 round trips in a row, was 5 k cycles = 2 us of every launch.)
        constexpr int NT = 64 * NWAVE, N0 = K0 * 6, N1 = K1 * 6, N2 = K2 * 6, WB = (N0 + N1 + N2 + NT - 1) / NT, CB = 8;
        float4 wv[WB], cv[CB];
#pragma unroll
        for (int u = 0; u < WB; ++u) {
            const int e = min((int)threadIdx.x + u * NT, N0 + N1 + N2 - 1);
            const float* src = (e < N0) ? W0 + e * 4 : (e < N0 + N1) ? W1 + (e - N0) * 4 : W2 + (e - N0 - N1) * 4;
            wv[u] = *reinterpret_cast<const float4*>(src);
        }
        const int ctotal = n_per_cloud * (C / 4);
        // KNN, C = 24, xyz != nullptr: the block's input IS feature_extraction's layer0 (ops.py:1449-1451: a 3 -> 24 conv of the
        // coordinates, no activation) -- every workgroup evaluates it for its whole cloud while it stages (the fmaf chain over k = 0, 1, 2
        // and the separate bias add of linear_small_k_kernel, so the same bits), and writes the rows of its OWN points to Lout.
        const bool layer0 = KNN && C == 24 && xyz != nullptr;        // workgroup-uniform
        if (layer0) {
#pragma unroll
            for (int u = 0; u < CB; ++u) {
                const int e = (int)threadIdx.x + u * NT;
                if (u * NT < ctotal) {                                // (compile-time for the usual n: three rounds of 512 threads)
                    const int ec = min(e, ctotal - 1);
                    const int p = ec / (C / 4), q = ec - p * (C / 4);
                    const float* xr = xyz + (size_t)(cloud0 + p) * 3;
                    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const float x = xr[k];
                        const float4 w = *reinterpret_cast<const float4*>(Wl + k * 24 + q * 4);
                        acc.x = __builtin_fmaf(x, w.x, acc.x); acc.y = __builtin_fmaf(x, w.y, acc.y);
                        acc.z = __builtin_fmaf(x, w.z, acc.z); acc.w = __builtin_fmaf(x, w.w, acc.w);
                    }
                    const float4 b = *reinterpret_cast<const float4*>(bl + q * 4);
                    acc.x = acc.x + b.x; acc.y = acc.y + b.y; acc.z = acc.z + b.z; acc.w = acc.w + b.w;
                    cv[u] = acc;
                    const int pl = p - (g_lo * 2 - cloud0);
                    if (e < ctotal && pl >= 0 && pl < (g_hi - g_lo) * 2)
                        *reinterpret_cast<float4*>(Lout + (size_t)(cloud0 + p) * ldl + q * 4) = acc;
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < CB; ++u) {
                const int e = min((int)threadIdx.x + u * NT, ctotal - 1);
                const int p = e / (C / 4), q = e - p * (C / 4);
                cv[u] = *reinterpret_cast<const float4*>(F + (size_t)(cloud0 + p) * ldf + q * 4);
            }
        }
#pragma unroll
        for (int u = 0; u < WB; ++u) {
            const int e = (int)threadIdx.x + u * NT;
            if (e < N0) edge_scatter_f4(f0, e, wv[u]);
            else if (e < N0 + N1) edge_scatter_f4(f1, e - N0, wv[u]);
            else if (e < N0 + N1 + N2) edge_scatter_f4(f2, e - N0 - N1, wv[u]);
        }
        for (int e = threadIdx.x; e < (K0 + K1 + K2) * 8; e += NT)       // padding lanes 24..31 of every (s, half): the layers are contiguous
            frag[(e >> 3) * 32 + 24 + (e & 7)] = 0.f;
#pragma unroll
        for (int u = 0; u < CB; ++u) {
            const int e = (int)threadIdx.x + u * NT;
            const int p = e / (C / 4), q = e - p * (C / 4);
            if (e < ctotal) *reinterpret_cast<float4*>(fl + p * FLD + q * 4) = cv[u];
        }
        if (!layer0)
            for (int e = threadIdx.x + CB * NT; e < ctotal; e += NT) {       // clouds of more than 8 float4 per thread (C = 48: n > 341)
                const int p = e / (C / 4), q = e - p * (C / 4);
                *reinterpret_cast<float4*>(fl + p * FLD + q * 4) = *reinterpret_cast<const float4*>(F + (size_t)(cloud0 + p) * ldf + q * 4);
            }
    }
    __syncthreads();
#ifdef EDGE_STAMPS
    const unsigned long long e_ka = __builtin_readcyclecounter();
    unsigned long long e_kb = e_ka, e_kc = e_ka;
#endif
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = lane & 31, h = lane >> 5;
    float* pre = fl + n_per_cloud * FLD;                         // PRE: [points of this workgroup][2 halves][12] layer-0 prefixes
    // KNN: candidate norms, distance words [query][candidate], prefilter scratch per wave, neighbour table [query][EK_KLD]
    float* norms = pre + per_wg * 2 * 24;
    uint32_t* dmat = reinterpret_cast<uint32_t*>(norms + ((n_per_cloud + 3) & ~3));
    uint64_t* kscr = reinterpret_cast<uint64_t*>(dmat + EK_MAXQ * EK_NP);
    int* nbr = reinterpret_cast<int*>(kscr + NWAVE * EK_SCRATCH);
    // One phase for everything that only needs the staged cloud: the candidates' norms (one thread per point), the workgroup's distance
    // dot products (all waves: 16 x 16 tiles, accumulators stay in registers over the barrier) and the layer-0 prefixes (PRE).
    const int n = n_per_cloud, p_base = g_lo * 2 - cloud0, ngl = g_hi - g_lo, nq = ngl * 2;
    typedef float ek_f32x4 __attribute__((ext_vector_type(4)));
    ek_f32x4 d00 = {0.f, 0.f, 0.f, 0.f}, d01 = d00, d10 = d00, d11 = d00;       // (query tile, candidate tile) dot products
    const int i16 = lane & 15, q4 = lane >> 4;
    const int ntile = (n + 15) >> 4;                             // <= 16: candidate tiles `wave` and `wave + NWAVE`
    if constexpr (KNN) {
        for (int p = threadIdx.x; p < n; p += 64 * NWAVE) {      // |F_p|^2: the ascending-channel chain of knn_feat_wave_kernel
            const float* fr = fl + p * FLD;
            float r = 0.f;
#pragma unroll
            for (int c4 = 0; c4 < C / 4; ++c4) {
                const float4 v = *reinterpret_cast<const float4*>(fr + c4 * 4);
                r = __builtin_fmaf(v.x, v.x, r); r = __builtin_fmaf(v.y, v.y, r);
                r = __builtin_fmaf(v.z, v.z, r); r = __builtin_fmaf(v.w, v.w, r);
            }
            norms[p] = r;
        }
        if (wave < ntile) {
            // A = 16 query rows (4 channels per step), B = 16 candidates: v_mfma_f32_16x16x4_f32, the instruction, operand order and
            // ascending-channel chain of knn_feat_wave_kernel.  Four independent chains (2 query tiles x 2 candidate tiles) interleave
            // on the pipe; tiles that do not exist (nq <= 16, n <= 128) compute on clamped rows and are dropped below.
            const float* q0row = fl + min(p_base + i16, n - 1) * FLD + q4;
            const float* q1row = fl + min(p_base + 16 + i16, n - 1) * FLD + q4;
            const float* c0p = fl + min(wave * 16 + i16, n - 1) * FLD + q4;
            const float* c1p = fl + min((wave + NWAVE) * 16 + i16, n - 1) * FLD + q4;
#pragma unroll
            for (int c4 = 0; c4 < C / 4; ++c4) {
                const float a0 = q0row[c4 * 4], a1 = q1row[c4 * 4], b0v = c0p[c4 * 4], b1v = c1p[c4 * 4];
                d00 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0v, d00, 0, 0, 0);
                d01 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1v, d01, 0, 0, 0);
                d10 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0v, d10, 0, 0, 0);
                d11 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1v, d11, 0, 0, 0);
            }
        }
    }
    if constexpr (PRE) {
        const int npts_wg = nq;
        const int tw = KNN ? NWAVE - 1 - wave : wave;            // KNN: the last wave (the first four also did the norms)
        for (int t = tw; t * 32 < npts_wg; t += NWAVE) {
            const int pl = min(t * 32 + row, npts_wg - 1);
            const float* fr = fl + (p_base + pl) * FLD;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int sq = 0; sq < C / 4; ++sq) {                 // steps 2 sq, 2 sq + 1: elements k = 4 sq + h and 4 sq + 2 + h of F_p
                const float4 a = *reinterpret_cast<const float4*>(fr + sq * 4);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f0[(2 * sq) * 64 + lane], h ? a.y : a.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f0[(2 * sq + 1) * 64 + lane], h ? a.w : a.z, acc, 0, 0, 0);
            }
            if (t * 32 + row < npts_wg) {
                float* pp = pre + ((t * 32 + row) * 2 + h) * 12;
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    *reinterpret_cast<float4*>(pp + q * 4) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
            }
        }
        __syncthreads();
    }
#ifdef EDGE_STAMPS
    e_kb = __builtin_readcyclecounter();
#endif
    if constexpr (KNN) {
        if (wave < ntile) {   // distance words (rq - 2 dot) + rp, the expression of knn_feat_wave_kernel, as ordered words [query][candidate]
            const int cand0 = wave * 16 + i16, cand1 = cand0 + NWAVE * 16;
            const float rp0 = norms[min(cand0, n - 1)], rp1 = norms[min(cand1, n - 1)];
#pragma unroll
            for (int r = 0; r < 4; ++r) {                        // d..[r] = dot(query 4 q4 + r of the tile, candidate)
                const float rq0 = norms[min(p_base + 4 * q4 + r, n - 1)], rq1 = norms[min(p_base + 16 + 4 * q4 + r, n - 1)];
                uint32_t* m0 = dmat + (4 * q4 + r) * EK_NP;
                uint32_t* m1 = m0 + 16 * EK_NP;
                if (cand0 < n) {
                    m0[cand0] = f32_to_ordered(((rq0 - 2.0f * d00[r]) + rp0) + 0.0f);
                    if (nq > 16) m1[cand0] = f32_to_ordered(((rq1 - 2.0f * d10[r]) + rp0) + 0.0f);
                }
                if (cand1 < n) {
                    m0[cand1] = f32_to_ordered(((rq0 - 2.0f * d01[r]) + rp1) + 0.0f);
                    if (nq > 16) m1[cand1] = f32_to_ordered(((rq1 - 2.0f * d11[r]) + rp1) + 0.0f);
                }
            }
        }
        __syncthreads();
#ifdef EDGE_STAMPS
        e_kc = __builtin_readcyclecounter();
#endif
    }
    // selection: a wave serves the two points of each of its own pair groups (group g_lo + wave + NWAVE i)
    auto knn_select = [&](int gl) {
        if constexpr (KNN) {
            const int qa = 2 * gl, qb = qa + 1;
            uint32_t oda[4], odb[4];
            int cp[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int p = lane + 64 * r;
                cp[r] = p;
                oda[r] = (p < n) ? dmat[qa * EK_NP + p] : 0xFFFFFFFFu;
                odb[r] = (p < n) ? dmat[qb * EK_NP + p] : 0xFFFFFFFFu;
            }
            uint64_t* buf = kscr + (size_t)wave * EK_SCRATCH;
            int* outa = nbr + qa * EK_KLD;
            int* outb = nbr + qb * EK_KLD;
            const auto w2f = [](uint32_t w) { return ordered_to_f32(w); };
            const bool done_a = prefilter_rank<4, true>(oda, cp, buf, lane, ksel, 0xFFFFFFFEu, outa, nullptr, w2f);
            const bool done_b = prefilter_rank<4, true>(odb, cp, buf, lane, ksel, 0xFFFFFFFEu, outb, nullptr, w2f);
            if (!(done_a && done_b)) {                                    // degenerate clouds: sort everything
                uint64_t ka[4], kb[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ka[r] = (cp[r] < n) ? (((uint64_t)oda[r] << 32) | (uint32_t)cp[r]) : KEY_MAX;
                    kb[r] = (cp[r] < n) ? (((uint64_t)odb[r] << 32) | (uint32_t)cp[r]) : KEY_MAX;
                }
                sort_keys<4>(ka);
                sort_keys<4>(kb);
                uint64_t resa, resb;
                select_k2<4>(ka, kb, lane, ksel, resa, resb);
                if (lane < ksel) {
                    if (!done_a) outa[lane] = (int)(uint32_t)resa;
                    if (!done_b) outb[lane] = (int)(uint32_t)resb;
                }
            }
            if (idx_out != nullptr && lane < ksel) {
                int* o = idx_out + (size_t)(cloud0 + p_base + qa) * ksel;
                o[lane] = outa[lane];
                o[ksel + lane] = outb[lane];
            }
        }
    };
    // All of a wave's selections come before its first convolution.  (Tried: half of the waves select group by group, right before
    // each convolution, so that one wave of a SIMD sits in the vector ALU while the other occupies the matrix pipe.  A wave issuing a
    // dependent MFMA chain starves its SIMD-mate's VALU issue: the early selections took 19 k cycles instead of 10 k, 31.7 vs 30.7 us.)
    if constexpr (KNN) {
        for (int gl = wave; gl < ngl; gl += NWAVE) knn_select(gl);
    }
#ifdef EDGE_STAMPS
    const unsigned long long e_k1 = __builtin_readcyclecounter();
#endif

    const int ngroups = LDSF ? g_hi : (npoints + 1) / 2;         // 2 points per wave
    const int s_nb = row & 15;
    // the rows of the NEXT point group are fetched while the MFMAs of the current one run (ra/rb double as prefetch regs)
    float4 ra[LDSF ? 1 : C / 4], rb[LDSF ? 1 : C / 4];
    int jl_next = 0;                                             // LDSF: cloud-local neighbour id of the next group
    auto fetch = [&](int grp) {
        int p = grp * 2 + (row >> 4);
        if (p >= npoints) p = npoints - 1;
        if constexpr (KNN) {
            jl_next = nbr[(p - g_lo * 2) * EK_KLD + ioff + s_nb];
        } else if constexpr (LDSF) {
            jl_next = idx[(size_t)p * ldi + ioff + s_nb];
        } else {
            const int j = (p / n_per_cloud) * n_per_cloud + idx[(size_t)p * ldi + ioff + s_nb];
#pragma unroll
            for (int q = 0; q < C / 4; ++q) {
                ra[q] = *reinterpret_cast<const float4*>(F + (size_t)p * ldf + q * 4);
                rb[q] = *reinterpret_cast<const float4*>(F + (size_t)j * ldf + q * 4);
            }
        }
    };
    const int grp0 = gfirst + wave, gstride = gstep0;
    if (grp0 < ngroups) fetch(grp0);
    constexpr int SKIP = PRE ? H : 0;               // PRE: the first H steps of layer 0 were taken per point
    constexpr int WD = 12, STOT = S0 + S1 + S2 - SKIP;   // ring depth must divide STOT (84, 132; 72, 108 with PRE): the ring wraps into the next group
    static_assert(STOT % WD == 0, "fragment ring depth must divide the step count");
    const float* fragG = frag + SKIP * 64;
    float wq[WD];
#pragma unroll
    for (int i = 0; i < WD; ++i) wq[i] = fragG[i * 64 + lane];
#ifdef EDGE_STAMPS
    unsigned long long e_conv = 0, e_l0 = 0, e_l1 = 0, e_l2 = 0, e_epi = 0, e_n = 0;
#define ED_T(v) __builtin_amdgcn_sched_barrier(0); const unsigned long long v = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0)
#else
#define ED_T(v)
#endif
    // (Tried: requesting the bottleneck conv's operands HERE, before the convolutions, so that they land in registers while the MFMA
    // chains run: 219 VGPRs and every phase slower, 42.0 vs 37.8 us per launch.)
    float tail_bias = 0.f;
    if constexpr (KNN) {
        if (Wp != nullptr) tail_bias = bp[min(wave >> 1, 2) * 16 + i16];
    }
    for (int grp = grp0; grp < ngroups; grp += gstride) {
        ED_T(t0);
        float fp[H], df[H];                                      // elements k = 2t + h of F_p and of F_j - F_p
        const float* fp_ = fl + (grp * 2 + (row >> 4) - cloud0) * FLD;
        const float* fj_ = fl + jl_next * FLD;
#pragma unroll
        for (int q = 0; q < C / 4; ++q) {
            const float4 a = LDSF ? *reinterpret_cast<const float4*>(fp_ + q * 4) : ra[q];
            const float4 b = LDSF ? *reinterpret_cast<const float4*>(fj_ + q * 4) : rb[q];
            const float a0 = h ? a.y : a.x, a1 = h ? a.w : a.z;
            const float b0v = h ? b.y : b.x, b1v = h ? b.w : b.z;
            fp[2 * q] = a0; fp[2 * q + 1] = a1;
            df[2 * q] = b0v - a0; df[2 * q + 1] = b1v - a1;
        }
        // ra / rb are consumed: start the next group's index + row loads now, so both latencies hide behind this
        // group's 132-MFMA chain instead of being exposed at the top of the next iteration
        if (grp + gstride < ngroups) fetch(grp + gstride);
        f32x16 l0, l1, l2;
#pragma unroll
        for (int r = 0; r < 16; ++r) { l0[r] = 0.f; l1[r] = 0.f; l2[r] = 0.f; }
        if constexpr (PRE) {                                     // the chain over F_p, evaluated once for this point
            const float* pp = pre + (((grp - g_lo) * 2 + (row >> 4)) * 2 + h) * 12;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(pp + q * 4);
                l0[4 * q] = v.x; l0[4 * q + 1] = v.y; l0[4 * q + 2] = v.z; l0[4 * q + 3] = v.w;
            }
        }
        // The weight fragments of all three layers are one contiguous LDS array of STOT steps.  They do not depend on
        // the data, so they run through a register ring WD steps ahead of the MFMA that consumes them (wrapping into
        // the next point group): with one wave per SIMD nothing else would hide the ds_read latency.
        auto step = [&](int sg, float bv, f32x16& acc) {
            const float w = wq[sg % WD];
            wq[sg % WD] = fragG[((sg + WD) % STOT) * 64 + lane];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w, bv, acc, 0, 0, 0);
        };
        ED_T(t1);
        // layer 0: k over [F_p (C), F_j - F_p (C)]
#pragma unroll
        for (int s = SKIP; s < S0; ++s) {
            const float bv = (s < H) ? fp[s < H ? s : 0] : df[s < H ? 0 : s - H];
            step(s - SKIP, bv, l0);
        }
#pragma unroll
        for (int r = 0; r < 12; ++r) l0[r] = fmaxf(l0[r] + b0[2 * r + h], 0.f);
        ED_T(t2);
        // layer 1: k over [l0 (24), F_p (C)]
#pragma unroll
        for (int s = 0; s < S1; ++s) {
            const float bv = (s < 12) ? l0[s < 12 ? s : 0] : fp[s < 12 ? 0 : s - 12];
            step(S0 - SKIP + s, bv, l1);
        }
#pragma unroll
        for (int r = 0; r < 12; ++r) l1[r] = fmaxf(l1[r] + b1[2 * r + h], 0.f);
        ED_T(t3);
        // layer 2: k over [l1 (24), l0 (24), F_p (C)], no activation
#pragma unroll
        for (int s = 0; s < S2; ++s) {
            const float bv = (s < 12) ? l1[s < 12 ? s : 0] : ((s < 24) ? l0[(s >= 12 && s < 24) ? s - 12 : 0] : fp[s >= 24 ? s - 24 : 0]);
            step(S0 + S1 - SKIP + s, bv, l2);
        }
        ED_T(t4);
        // max over the 16 neighbours (one DPP row); lane 15 of each row parks [l2 | l1 | l0 | F_p] of its point in an LDS
        // staging row, then the wave writes its two points with ONE float4 store instruction.  (60 separate 4-byte
        // stores from four active lanes cost ~300 cycles of issue each next to the MFMAs: twice the 132-MFMA chain.)
        constexpr int OW = 3 * G + C;                                // 96 / 120 floats per point
        float* stg = stage + (KNN ? (grp - g_lo) : wave) * (2 * OWS);
        float* sp = stg + (row >> 4) * OWS;
        float k2 = 0.f, k1 = 0.f, k0 = 0.f;
#pragma unroll
        for (int r = 0; r < 12; ++r) {
            float m2 = l2[r] + b2[2 * r + h], m1 = l1[r], m0 = l0[r];
            // three interleaved row maxima, v_max_f32 with the rotated partner as its DPP operand (row_ror 1, 2, 4, 8): 12
            // instructions.  The compiler's form of edge_row16_max is v_mov_b32_dpp + v_max_f32 per step (24 + wait states);
            // here the two other chains' instructions are the wait states each VALU-write -> DPP-read needs.
            asm("s_nop 1\n\t"
                "v_max_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
                "v_max_f32_dpp %1, %1, %1 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
                "v_max_f32_dpp %2, %2, %2 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
                "v_max_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
                "v_max_f32_dpp %1, %1, %1 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
                "v_max_f32_dpp %2, %2, %2 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
                "v_max_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                "v_max_f32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                "v_max_f32_dpp %2, %2, %2 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                "v_max_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                "v_max_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                "v_max_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf"
                : "+v"(m2), "+v"(m1), "+v"(m0));
            // every lane of the row now holds the three maxima of register r: lane r of the row keeps them (one select each)
            // and the 36 values leave as THREE LDS stores after the loop instead of 36 stores from one active lane per row
            k2 = (s_nb == r) ? m2 : k2; k1 = (s_nb == r) ? m1 : k1; k0 = (s_nb == r) ? m0 : k0;
        }
        if (s_nb < 12) { sp[2 * s_nb + h] = k2; sp[G + 2 * s_nb + h] = k1; sp[2 * G + 2 * s_nb + h] = k0; }
        {   // F_p (the same in all 16 lanes of a row): lane t of the row stores fp[t], lanes t - 16 the rest
            float fa = 0.f, fb = 0.f;
#pragma unroll
            for (int t = 0; t < H; ++t) {
                if (t < 16) fa = (s_nb == t) ? fp[t] : fa;
                else fb = (s_nb == t - 16) ? fp[t] : fb;
            }
            if (s_nb < (H < 16 ? H : 16)) sp[3 * G + 2 * s_nb + h] = fa;
            if (H > 16 && s_nb < H - 16) sp[3 * G + 2 * (s_nb + 16) + h] = fb;
        }
        // (LDS operations of one wave execute in order: the reads below see the writes above)
        const int p_first = grp * 2;
        if (vec_store) {
            if (lane < 2 * (OW / 4)) {
                const int pt = lane / (OW / 4), q4 = lane - pt * (OW / 4);
                if (p_first + pt < npoints)
                    *reinterpret_cast<float4*>(Y + (size_t)(p_first + pt) * ldy + q4 * 4) = *reinterpret_cast<const float4*>(stg + pt * OWS + q4 * 4);
            }
        } else {
            for (int e = lane; e < 2 * OW; e += 64) {
                const int pt = e / OW, q = e - pt * OW;
                if (p_first + pt < npoints) Y[(size_t)(p_first + pt) * ldy + q] = stg[pt * OWS + q];
            }
        }
#ifdef EDGE_STAMPS
        { ED_T(t5); e_conv += t1 - t0; e_l0 += t2 - t1; e_l1 += t3 - t2; e_l2 += t4 - t3; e_epi += t5 - t4; ++e_n; }
#endif
    }
#ifdef EDGE_STAMPS
    const unsigned long long e_p0 = __builtin_readcyclecounter();
    unsigned long long e_p1 = e_p0, e_p2 = e_p0, e_p3 = e_p0;
#endif
    if constexpr (KNN) {
        // The next block's bottleneck conv (feature_extraction layer<d+1>_prep, ops.py:1455-1462) for the workgroup's own points:
        // relu([this block's 72 + C outputs | the k_old older feature columns to their right in Y] . Wp + bp), 48 channels.  Six waves
        // take one 16-point x 16-channel tile each on v_mfma_f32_16x16x4_f32 -- instruction, operand roles and ascending-k order of
        // linear_skinny_kernel, which ran this conv as a launch of its own (6 - 13 us of exposed latency for 0.1 - 0.3 GFLOP): the new
        // columns come from the LDS staging rows, the older ones and the weights from L2.
        if (Wp != nullptr) {
            // Operands through LDS, once per workgroup: Wp [K][48] over the (now dead) cloud copy .. distance matrix, the older columns
            // of the workgroup's rows [nq][k_old + 4] over the weight fragments; every global load is requested before the first LDS
            // store.  (Per-wave operand loads from L2, round by round, made this tail as slow as the launch it replaces.)
            constexpr int OW = 3 * G + C, NT = 64 * NWAVE;
            const int K = OW + k_old, xld = k_old + 4;
            float* wl = fl;
            float* xl = frag;
            __syncthreads();                                     // every wave is through its convolutions: staging rows complete
#ifdef EDGE_STAMPS
            e_p1 = __builtin_readcyclecounter();
#endif
            const int wtot = K * 12, xq = k_old >> 2, xtot = nq * xq;
            constexpr int TB_W = 9, TB_X = 4;
            float4 wv[TB_W], xv[TB_X];
#pragma unroll
            for (int u = 0; u < TB_W; ++u) wv[u] = *reinterpret_cast<const float4*>(Wp + (size_t)min((int)threadIdx.x + u * NT, wtot - 1) * 4);
#pragma unroll
            for (int u = 0; u < TB_X; ++u) {
                const int e = min((int)threadIdx.x + u * NT, max(xtot - 1, 0));
                const int pl = xq ? e / xq : 0, c4 = e - pl * xq;
                xv[u] = xtot ? *reinterpret_cast<const float4*>(Y + (size_t)(g_lo * 2 + pl) * ldy + OW + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < TB_W; ++u) {
                const int e = (int)threadIdx.x + u * NT;
                if (e < wtot) *reinterpret_cast<float4*>(wl + e * 4) = wv[u];
            }
#pragma unroll
            for (int u = 0; u < TB_X; ++u) {
                const int e = (int)threadIdx.x + u * NT;
                const int pl = xq ? e / xq : 0, c4 = e - pl * xq;
                if (e < xtot) *reinterpret_cast<float4*>(xl + pl * xld + c4 * 4) = xv[u];
            }
            for (int e = threadIdx.x + TB_W * NT; e < wtot; e += NT)   // (K > 384)
                *reinterpret_cast<float4*>(wl + e * 4) = *reinterpret_cast<const float4*>(Wp + (size_t)e * 4);
            for (int e = threadIdx.x + TB_X * NT; e < xtot; e += NT) {
                const int pl = e / xq, c4 = e - pl * xq;
                *reinterpret_cast<float4*>(xl + pl * xld + c4 * 4) = *reinterpret_cast<const float4*>(Y + (size_t)(g_lo * 2 + pl) * ldy + OW + c4 * 4);
            }
            __syncthreads();
#ifdef EDGE_STAMPS
            e_p2 = __builtin_readcyclecounter();
#endif
            const int pt = wave & 1, ct = wave >> 1;
            if (ct < 3 && pt * 16 < nq) {
                const int pl = min(pt * 16 + i16, nq - 1);
                const float* xn = stage + pl * OWS + q4;
                const float* xo = xl + pl * xld + q4;
                const float* wp = wl + q4 * 48 + ct * 16 + i16;
                ek_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                // rounds of six steps (24 input columns); the next round's twelve LDS reads are issued before this round's chain
                const int nround = K / 24;
                float av[6], bv[6], an[6], bn[6];
                auto rd_load = [&](int rd, float (&a)[6], float (&b)[6]) {
                    rd = min(rd, nround - 1);
                    const float* xa = (rd < OW / 24) ? xn + rd * 24 : xo + (rd - OW / 24) * 24;
                    const float* wb = wp + rd * 24 * 48;
#pragma unroll
                    for (int u = 0; u < 6; ++u) { a[u] = xa[u * 4]; b[u] = wb[u * 4 * 48]; }
                };
                rd_load(0, av, bv);
                for (int rd = 0; rd < nround; rd += 2) {             // two rounds per trip: the operand buffers alternate without copies
                    rd_load(rd + 1, an, bn);
#pragma unroll
                    for (int u = 0; u < 6; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
                    if (rd + 1 < nround) {
                        rd_load(rd + 2, av, bv);
#pragma unroll
                        for (int u = 0; u < 6; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(an[u], bn[u], acc, 0, 0, 0);
                    }
                }
#ifdef EDGE_STAMPS
                __builtin_amdgcn_sched_barrier(0); e_p3 = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0);
#endif
                const float bb = tail_bias;
#pragma unroll
                for (int r = 0; r < 4; ++r) {                        // acc[r] = out[point 4 q4 + r of the tile][channel ct * 16 + i16]
                    const int pr = pt * 16 + 4 * q4 + r;
                    if (pr < nq) Pout[(size_t)(g_lo * 2 + pr) * ldp + ct * 16 + i16] = fmaxf(acc[r] + bb, 0.f);
                }
            }
        }
    }
#ifdef EDGE_STAMPS
    if (blockIdx.x == 3 && blockIdx.y == 0 && lane == 0) {
        unsigned long long* st = reinterpret_cast<unsigned long long*>(Y + (size_t)npoints * ldy) + wave * 16;
        st[11] = e_p1 - e_p0; st[12] = e_p2 - e_p1; st[13] = __builtin_readcyclecounter() - e_p2; st[14] = e_p3 - e_p2;
        st[0] = e_conv; st[1] = e_l0; st[2] = e_l1; st[3] = e_l2; st[4] = e_epi; st[5] = e_n; st[6] = e_ka - e_k0; st[7] = e_kb - e_ka;
        st[8] = e_kc - e_kb; st[9] = e_k1 - e_kc; st[10] = __builtin_readcyclecounter() - e_k0;
    }
#endif
}

}  // namespace dispu

using namespace dispu;

DISPU_EXPORT int dispu_edge_dense_conv(int npoints, int n_per_cloud, int C, const float* F, long ldf, const int* idx, int ldi,
                                       int ioff, const float* W0, const float* b0, const float* W1, const float* b1,
                                       const float* W2, const float* b2, float* Y, long ldy, void* stream) {
    if (npoints < 0 || n_per_cloud <= 0 || !(C == 24 || C == 48) || (ldf & 3) || (((uintptr_t)F) & 15)) return (int)hipErrorInvalidValue;
    if (npoints == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const size_t frag_bytes = (size_t)((C == 24 ? 84 : 132) * 64 + 16 * (72 + C)) * sizeof(float);  // weight fragments + output staging (up to 8 waves)
    // LDS-resident cloud features when a cloud fits next to the weight fragments (n <= ~600 points at C = 48) and the
    // points are whole clouds; each cloud is split over `parts` workgroups so that ~256 of them exist
    const size_t feat_bytes = (size_t)n_per_cloud * (C + 4) * sizeof(float);
    if (npoints % n_per_cloud == 0 && n_per_cloud % 2 == 0 && frag_bytes + feat_bytes <= 160 * 1024) {
        const int clouds = npoints / n_per_cloud;
        int parts = (256 + clouds - 1) / clouds;
        const int gpc = n_per_cloud / 2;
        if (parts > (gpc + 7) / 8) parts = (gpc + 7) / 8;                 // at least one group per wave
        if (parts < 1) parts = 1;
        const int per = (gpc + parts - 1) / parts;                        // point groups per workgroup
        const size_t pre_bytes = (size_t)per * 2 * 24 * sizeof(float);    // layer-0 prefixes of the workgroup's points
        const bool pre = frag_bytes + feat_bytes + pre_bytes <= 160 * 1024;   // layer 0's per-point prefix (round 4) whenever it fits
        const size_t bytes = frag_bytes + feat_bytes + (pre ? pre_bytes : 0);
        static DevOnce attr;      
        if (attr.needed()) {
            DISPU_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(edge_dense_conv_mfma_kernel<24, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            DISPU_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(edge_dense_conv_mfma_kernel<48, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            DISPU_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(edge_dense_conv_mfma_kernel<24, true, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            DISPU_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(edge_dense_conv_mfma_kernel<48, true, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr.done();
        }
#define EDGE_LAUNCH(C_, PRE_) hipLaunchKernelGGL((edge_dense_conv_mfma_kernel<C_, true, PRE_>), dim3(clouds, parts), dim3(512), bytes, s, npoints, \
                                                 n_per_cloud, F, ldf, idx, ldi, ioff, W0, b0, W1, b1, W2, b2, Y, ldy, nullptr, 0, nullptr, nullptr, 0, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0)
        if (C == 24) { if (pre) EDGE_LAUNCH(24, true); else EDGE_LAUNCH(24, false); }
        else { if (pre) EDGE_LAUNCH(48, true); else EDGE_LAUNCH(48, false); }
#undef EDGE_LAUNCH
        return (int)hipGetLastError();
    }
    int g = (npoints + 7) / 8;          // 8 points (4 waves x 2) per workgroup pass
    if (g > 256) g = 256;               // workgroups loop over point groups: the LDS weight image is built once per CU
    if (C == 24)
        hipLaunchKernelGGL((edge_dense_conv_mfma_kernel<24, false>), dim3(g), dim3(256), frag_bytes, s, npoints, n_per_cloud, F, ldf, idx, ldi, ioff, W0, b0, W1, b1, W2, b2, Y, ldy, nullptr, 0, nullptr, nullptr, 0, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0);
    else
        hipLaunchKernelGGL((edge_dense_conv_mfma_kernel<48, false>), dim3(g), dim3(256), frag_bytes, s, npoints, n_per_cloud, F, ldf, idx, ldi, ioff, W0, b0, W1, b1, W2, b2, Y, ldy, nullptr, 0, nullptr, nullptr, 0, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0);
    return (int)hipGetLastError();
}

// One dense block of feature_extraction_GCN (Common/ops.py:1437-1486) in one launch: knn_point_2(k + 1, F, F) (tf_util.py:618-651) ->
// get_edge_feature over neighbours ioff .. ioff + 15 (ops.py:1856-1877) -> dense_conv (:1897-1915).  Same results, bit for bit, as
// dispu_knn_feat_strided followed by dispu_edge_dense_conv.  Clouds of up to 256 points (n_per_cloud even, >= ksel); idx_out (nullable):
// the [npoints, ksel] neighbour table.
DISPU_EXPORT int dispu_stem_block(int npoints, int n_per_cloud, int C, const float* F, long ldf, int ksel, int ioff, const float* W0,
                                  const float* b0, const float* W1, const float* b1, const float* W2, const float* b2, float* Y, long ldy,
                                  int* idx_out, const float* Wp, const float* bp, int k_old, float* P, long ldp, const float* xyz,
                                  const float* Wl, const float* bl, float* Lout, long ldl, void* stream) {
    if (xyz != nullptr && (C != 24 || Wl == nullptr || bl == nullptr || Lout == nullptr || (ldl & 3) || (((uintptr_t)Lout) & 15) ||
                           (((uintptr_t)Wl) & 15) || (((uintptr_t)bl) & 15) || n_per_cloud * 6 > 8 * 512))
        return (int)hipErrorInvalidValue;
    if (xyz == nullptr && F == nullptr) return (int)hipErrorInvalidValue;
    if (Wp != nullptr && (bp == nullptr || P == nullptr || k_old < 0 || k_old % 24 != 0 || ldp < 48)) return (int)hipErrorInvalidValue;
    if (npoints < 0 || n_per_cloud <= 0 || !(C == 24 || C == 48) || (ldf & 3) || (((uintptr_t)F) & 15)) return (int)hipErrorInvalidValue;
    if (n_per_cloud > 256 || (n_per_cloud & 1) || npoints % n_per_cloud != 0 || ioff < 0 || ksel != ioff + 16 || ksel > EK_KLD ||
        ksel > n_per_cloud)
        return (int)hipErrorInvalidValue;
    if (npoints == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int clouds = npoints / n_per_cloud, gpc = n_per_cloud / 2;
    int parts = (256 + clouds - 1) / clouds;
    if (parts > (gpc + 7) / 8) parts = (gpc + 7) / 8;
    if (parts < (gpc + EK_MAXQ / 2 - 1) / (EK_MAXQ / 2)) parts = (gpc + EK_MAXQ / 2 - 1) / (EK_MAXQ / 2);   // at most EK_MAXQ points per workgroup
    const int per = (gpc + parts - 1) / parts;
    if (Wp != nullptr) {       // the bottleneck conv's operands reuse LDS regions of the earlier phases: both must fit
        const long wcap = (long)n_per_cloud * (C + 4) + per * 48 + ((n_per_cloud + 3) & ~3) + EK_MAXQ * EK_NP, xcap = (C == 24 ? 84 : 132) * 64;
        if ((long)(72 + C + k_old) * 48 > wcap || (long)per * 2 * (k_old + 4) > xcap || (ldy & 3) || (((uintptr_t)Y) & 15) || (((uintptr_t)Wp) & 15))
            return (int)hipErrorInvalidValue;
    }
    const size_t bytes = (size_t)((C == 24 ? 84 : 132) * 64 + EK_MAXQ * (72 + C + 4)) * sizeof(float) + (size_t)n_per_cloud * (C + 4) * sizeof(float) +
                         (size_t)per * 2 * 24 * sizeof(float) + (size_t)((n_per_cloud + 3) & ~3) * sizeof(float) +
                         (size_t)EK_MAXQ * EK_NP * sizeof(uint32_t) + (size_t)8 * EK_SCRATCH * sizeof(uint64_t) + (size_t)EK_MAXQ * EK_KLD * sizeof(int);
    static DevOnce attr;
    if (attr.needed()) {
        DISPU_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(edge_dense_conv_mfma_kernel<24, true, true, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        DISPU_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(edge_dense_conv_mfma_kernel<48, true, true, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr.done();
    }
    if (C == 24)
        hipLaunchKernelGGL((edge_dense_conv_mfma_kernel<24, true, true, true>), dim3(clouds, parts), dim3(512), bytes, s, npoints, n_per_cloud, F, ldf,
                           nullptr, 0, ioff, W0, b0, W1, b1, W2, b2, Y, ldy, idx_out, ksel, Wp, bp, k_old, P, ldp, xyz, Wl, bl, Lout, ldl);
    else
        hipLaunchKernelGGL((edge_dense_conv_mfma_kernel<48, true, true, true>), dim3(clouds, parts), dim3(512), bytes, s, npoints, n_per_cloud, F, ldf,
                           nullptr, 0, ioff, W0, b0, W1, b1, W2, b2, Y, ldy, idx_out, ksel, Wp, bp, k_old, P, ldp, xyz, Wl, bl, Lout, ldl);
    return (int)hipGetLastError();
}
