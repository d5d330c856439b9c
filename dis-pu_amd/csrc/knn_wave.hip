// Wave-per-query exact k-NN for gfx950 (small clouds: n <= 1024), the fast path behind dispu_knn_xyz and
// dispu_knn_feat(_strided).  Reference semantics and citations are in knn.hip; results are bit-identical to the
// lane-per-query kernels there (same pinned distance arithmetic, ascending distance, ties -> lower index).
//
// Why a second formulation: with one lane per query the sorted-insert (~4 VALU ops per list slot) runs for
// nearly every candidate, because SOME lane of the 64 accepts it; at the generator's sizes (8 K - 32 K queries)
// that also leaves most of the 1024 SIMDs idle.  Here a WAVE owns a query: lane l evaluates candidates
// l, l+64, ... (R per lane), packs (ordered distance bits << 32 | index) into a 64-bit key, sorts its R keys with
// an odd-even merge network in registers, parks them in LDS, and the k results are k rounds of a DPP wave-min
// over the lanes' current heads (the winner advances its head).  All lanes do useful work in every step and
// every CU is busy.  xyz: candidates stay in VGPRs for all queries of the wave, the query sits in SGPRs.
// Feature space: the cloud's features are staged once per workgroup in LDS, channel-quad major
// ([C/4][n] float4 -> conflict-free ds_read_b128), dots are the same ascending-channel fmaf chains.
#include "common.h"
#include "knn_select.h"

#include <cstdlib>

namespace dispu {


template <int R, bool FMA, bool PK = false>
__global__ __launch_bounds__(256, 4) void knn_xyz_wave_kernel(int n, int m, int k, int qpb, const float* __restrict__ support,
                                                            long sstride, const float* __restrict__ query, int* __restrict__ idx,
                                                            float* __restrict__ dist) {
    __shared__ uint64_t sorted[4][R * 64 + 4];
#ifdef KNN_STAMPS
    const unsigned long long x_k0 = __builtin_readcyclecounter();
#endif
    const int cloud = blockIdx.y, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const float* __restrict__ s = support + (size_t)cloud * sstride;      // sstride = 3 n, or 3 n_total when `support` is a chunk of a larger cloud
    const float* __restrict__ q = query + (size_t)cloud * m * 3;
    // lane l holds candidate 64 r + ((l + ROT r) & 63) of every 64-block r.  ROT = 17 for the prefiltered path: clouds
    // whose near neighbours sit a multiple of 64 apart in memory (the generator's coarse clouds: the 4 children of a
    // parent are 256 apart) would otherwise put them all in ONE lane, where they hide behind the lane minimum and
    // the threshold admits ~4x more survivors.  Candidates past n: +inf coordinates (prefilter) / all-ones keys.
    constexpr int ROT = (R >= 8) ? 17 : 0;
    float cx[R], cy[R], cz[R];
    int cp[R];
    // The cloud goes through LDS once per workgroup (coalesced float4, over the selection scratch, which is not in use yet) and every
    // wave takes its candidates from there.  Per-lane global loads (48 strided dwords per lane at R = 16, the same 12 KB in every one
    // of the 4096 waves: 50 MB through L2 for a 0.4 MB input) were 8.8 k of a wave's 40 k cycles.
    {
        float* sl = reinterpret_cast<float*>(&sorted[0][0]);
        stage_cloud_xyz<256, (R >= 16) ? 3 : 1>(sl, s, n);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int p = 64 * r + ((lane + ROT * r) & 63);
            const bool ok = p < n;
            const int pc = ok ? p : 0;
            cp[r] = p;
            cx[r] = ok ? sl[pc * 3 + 0] : __builtin_inff();
            cy[r] = ok ? sl[pc * 3 + 1] : __builtin_inff();
            cz[r] = ok ? sl[pc * 3 + 2] : __builtin_inff();
        }
        __syncthreads();                                             // the scratch is free for the selections from here on
    }
    const int q0 = blockIdx.x * qpb, q1 = min(m, q0 + qpb);
#ifdef KNN_STAMPS
#define KX_T(v) __builtin_amdgcn_sched_barrier(0); const unsigned long long v = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0)
    unsigned long long x_d = 0, x_t = 0, x_c = 0, x_s = 0, x_n = 0, x_f = 0, x_r = 0;
    __builtin_amdgcn_sched_barrier(0); const unsigned long long x_k1 = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0);
#else
#define KX_T(v)
#endif
    for (int qv = q0 + wave; qv < q1; qv += 4) {
        KX_T(u0);
        const int qi = __builtin_amdgcn_readfirstlane(qv);
        const float qx = q[qi * 3 + 0], qy = q[qi * 3 + 1], qz = q[qi * 3 + 2];
        // distance words: the raw bits of the (non-negative) squared distance order like the floats themselves
        uint32_t od[R];
        if constexpr (PK && R >= 2) {                            // candidate pairs on the packed fp32 ops: 64 instead of 128 instructions at R = 16
            const f32x2 qx2 = {qx, qx}, qy2 = {qy, qy}, qz2 = {qz, qz};
#pragma unroll
            for (int r = 0; r < R; r += 2) {
                const f32x2 cxp = {cx[r], cx[r + 1]}, cyp = {cy[r], cy[r + 1]}, czp = {cz[r], cz[r + 1]};
                const f32x2 d = sqdist3_x2<FMA>(qx2 - cxp, qy2 - cyp, qz2 - czp);
                od[r] = (R >= 8 || cp[r] < n) ? __float_as_uint(d.x) : 0xFFFFFFFFu;
                od[r + 1] = (R >= 8 || cp[r + 1] < n) ? __float_as_uint(d.y) : 0xFFFFFFFFu;
            }
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float d = sqdist3<FMA>(qx - cx[r], qy - cy[r], qz - cz[r]);
                od[r] = (R >= 8 || cp[r] < n) ? __float_as_uint(d) : 0xFFFFFFFFu;
            }
        }
        KX_T(u1);
        bool done = false;
        const size_t o = ((size_t)cloud * m + qi) * k;
#ifdef KNN_STAMPS
        unsigned long long pf[4] = {0, 0, 0, 0};
        if constexpr (R >= 8) {
            done = prefilter_rank<R, false>(od, cp, sorted[wave], lane, k, 0x7F7FFFFFu, idx + o, dist ? dist + o : nullptr,
                                            [](uint32_t w) { return __uint_as_float(w); }, pf);
            x_t += pf[1] - pf[0]; x_c += pf[2] - pf[1]; x_r += pf[3] - pf[2];
        }
#else
        if constexpr (R >= 8)      // tmax = largest finite float: never admits the padding (+inf / NaN)
            done = prefilter_rank<R, false>(od, cp, sorted[wave], lane, k, 0x7F7FFFFFu, idx + o, dist ? dist + o : nullptr,
                                            [](uint32_t w) { return __uint_as_float(w); });
#endif
        if (!done) {
#ifdef KNN_STAMPS
            ++x_f;
#endif
            uint64_t key[R];
#pragma unroll
            for (int r = 0; r < R; ++r) key[r] = (cp[r] < n) ? (((uint64_t)od[r] << 32) | (uint32_t)cp[r]) : KEY_MAX;
            sort_keys<R>(key);
            const uint64_t res = select_k<R>(key, sorted[wave], lane, k);
            if (lane < k) {
                idx[o + lane] = (int)(uint32_t)res;
                if (dist) dist[o + lane] = __uint_as_float((uint32_t)(res >> 32));
            }
        }
#ifdef KNN_STAMPS
        { KX_T(u4); x_d += u1 - u0; x_s += u4 - u1; ++x_n; }
#endif
    }
#ifdef KNN_STAMPS
    if (blockIdx.x == 3 && blockIdx.y == 1 && lane == 0) {
        unsigned long long* st = reinterpret_cast<unsigned long long*>(idx + (size_t)gridDim.y * m * k) + wave * 8;
        st[0] = x_d; st[1] = x_t; st[2] = x_c; st[3] = x_s; st[4] = x_n | (x_f << 32); st[5] = x_r; st[6] = x_k1 - x_k0;
        st[7] = __builtin_readcyclecounter() - x_k0;
    }
#endif
}

// Feature-space (GEMM-form) variant: D = (rq - 2 q.p) + rp, fma chains over ascending channels.
// LDS: feats[(c4 * (n+1) + p)] float4 = channels 4c4..4c4+3 of candidate p (zero padded to CP), norms[p].
// NW waves per workgroup share one staged copy of the cloud: the k selection rounds are a dependent chain of ~400
// cycles each (DPP min, readlane, ballot, LDS), so the SIMDs need several resident waves to stay busy; with 4-wave
// workgroups the 49 KB feature image limited a CU to 8 waves.
constexpr int KF_NW = 8;
constexpr int KF_SCRATCH = 128 + 4;          // u64 slots per wave for the prefilter's survivors (R <= 4)
template <int R, int CP>
__global__ __launch_bounds__(64 * KF_NW) void knn_feat_wave_kernel(int n, int m, int c, int k, int qpb, int ldp, int ldq, long pstride,
                                                             const float* __restrict__ points,
                                                             const float* __restrict__ queries, float* __restrict__ dist,
                                                             int* __restrict__ idx) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ns = n + 1;                                                             // row stride of feats (float4): +1 de-conflicts the staging writes
    float4* feats = reinterpret_cast<float4*>(smem);                                  // [CP/4][n+1]
    float* norms = reinterpret_cast<float*>(smem + (size_t)(CP / 4) * ns * 16);       // [n]
    uint64_t* sorted = reinterpret_cast<uint64_t*>(smem + (size_t)(CP / 4) * ns * 16 + (((size_t)n * 4 + 15) & ~15ull));
    // the workgroup's query rows, zero padded to CP: read per channel quad as one broadcast ds_read_b128.  (Scalar loads
    // of the query row inside the channel loop exposed an s_load round trip every two quads.)
    float4* qs = reinterpret_cast<float4*>(reinterpret_cast<char*>(sorted) + (R <= 4 ? (size_t)KF_NW * KF_SCRATCH * 8 : (size_t)KF_NW * R * 64 * 8));   // R <= 4: prefilter scratch only   // [qpb][CP/4]
    constexpr int QLD = CP / 4 + 1;      // query row stride in float4: rows 16 (CP + 4) bytes apart spread the per-query dword reads over the banks
    // R <= 4: the workgroup's 16 x n distance words (computed on the matrix pipe, read back per query) and the query norms
    constexpr int NP = 64 * R + 4;
    uint32_t* dmat = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(qs) + (size_t)16 * QLD * 16);    // [16][NP]
    float* qn = reinterpret_cast<float*>(dmat + 16 * NP);                                                   // [16]
    const int cloud = blockIdx.y, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const float* __restrict__ sp = points + (size_t)cloud * pstride;   // pstride = n ldp, or n_total ldp when `points` is a chunk of a larger cloud
    const float* __restrict__ qp = queries + (size_t)cloud * m * ldq;
#ifdef KNN_STAMPS
#define KN_T(v) __builtin_amdgcn_sched_barrier(0); const unsigned long long v = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0)
    unsigned long long k_dot = 0, k_sort = 0, k_sel = 0, k_nq = 0;
#else
#define KN_T(v)
#endif
    KN_T(ts0);
    const bool vec = (c == CP) && ((ldp & 3) == 0) && ((((uintptr_t)sp) & 15) == 0);
    if (vec) {
        // batches of 8 independent float4 loads per thread, then the 8 LDS stores: one memory round trip per batch
        // instead of one per element (the rolled loop was load -> wait -> store, 6 round trips for C = 48, n = 256)
        constexpr int UB = 8;
        const int total = (CP / 4) * n;
        for (int e0 = threadIdx.x; e0 < total; e0 += UB * 64 * KF_NW) {
            float4 v[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int e = min(e0 + u * 64 * KF_NW, total - 1);
                const int p = e / (CP / 4), c4 = e - p * (CP / 4);
                v[u] = *reinterpret_cast<const float4*>(sp + (size_t)p * ldp + c4 * 4);
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int e = e0 + u * 64 * KF_NW;
                const int p = e / (CP / 4), c4 = e - p * (CP / 4);
                if (e < total) feats[c4 * ns + p] = v[u];
            }
        }
    } else {
        for (int e = threadIdx.x; e < (CP / 4) * n; e += 64 * KF_NW) {
            const int p = e / (CP / 4), c4 = e - p * (CP / 4);       // consecutive lanes read one row's consecutive float4s
            const float* src = sp + (size_t)p * ldp + c4 * 4;
            float4 v;
            v.x = (c4 * 4 + 0 < c) ? src[0] : 0.f;
            v.y = (c4 * 4 + 1 < c) ? src[1] : 0.f;
            v.z = (c4 * 4 + 2 < c) ? src[2] : 0.f;
            v.w = (c4 * 4 + 3 < c) ? src[3] : 0.f;
            feats[c4 * ns + p] = v;
        }
    }
    {
        const int q0s = blockIdx.x * qpb;
        for (int e = threadIdx.x; e < qpb * (CP / 4); e += 64 * KF_NW) {
            const int ql = e / (CP / 4), c4 = e - ql * (CP / 4);
            const int qg = min(q0s + ql, m - 1);
            const float* src = qp + (size_t)qg * ldq + c4 * 4;
            float4 v;
            v.x = (c4 * 4 + 0 < c) ? src[0] : 0.f;
            v.y = (c4 * 4 + 1 < c) ? src[1] : 0.f;
            v.z = (c4 * 4 + 2 < c) ? src[2] : 0.f;
            v.w = (c4 * 4 + 3 < c) ? src[3] : 0.f;
            qs[ql * QLD + c4] = v;
        }
    }
    __syncthreads();
    for (int p = threadIdx.x; p < n; p += 64 * KF_NW) {
        float r = 0.f;
#pragma unroll
        for (int c4 = 0; c4 < CP / 4; ++c4) {
            const float4 v = feats[c4 * ns + p];
            r = __builtin_fmaf(v.x, v.x, r); r = __builtin_fmaf(v.y, v.y, r);
            r = __builtin_fmaf(v.z, v.z, r); r = __builtin_fmaf(v.w, v.w, r);
        }
        norms[p] = r;
    }
    if constexpr (R <= 4) {
        if (threadIdx.x < 16) {                                    // |q|^2, same ascending-channel chain as the candidates' norms
            float r = 0.f;
#pragma unroll
            for (int c4 = 0; c4 < CP / 4; ++c4) {
                const float4 v = qs[threadIdx.x * QLD + c4];
                r = __builtin_fmaf(v.x, v.x, r); r = __builtin_fmaf(v.y, v.y, r);
                r = __builtin_fmaf(v.z, v.z, r); r = __builtin_fmaf(v.w, v.w, r);
            }
            qn[threadIdx.x] = r;
        }
    }
    __syncthreads();
    float rp[R];
#pragma unroll
    for (int r = 0; r < R; ++r) rp[r] = (lane + 64 * r < n) ? norms[lane + 64 * r] : 0.f;

    const int q0 = blockIdx.x * qpb, q1 = min(m, q0 + qpb);
    KN_T(ts1);
    if constexpr (R <= 4) {
        // Phase 1, all waves: the 16 x n dot products of the workgroup on the matrix pipe.  v_mfma_f32_16x16x4_f32 is bit for
        // bit the ascending-k fmaf chain (tools/micro/mfma16_exact.hip), i.e. exactly the chain the VALU form of this kernel
        // ran per (query, candidate): A = the 16 query rows (lane = query, 4 channels per step), B = 16 candidates, one
        // instruction per channel quad.  Wave w takes candidate tiles w, w + 8; distances (rq - 2 dot) + rp go to LDS as
        // ordered words [query][candidate].  (As VALU fmaf chains the dots were half of this kernel's time.)
        {
            typedef float kf_f32x4 __attribute__((ext_vector_type(4)));
            KN_T(t0);
            const int i16 = lane & 15, q4 = lane >> 4;
            const float* qsf = reinterpret_cast<const float*>(qs);
            const float* ff = reinterpret_cast<const float*>(feats);
            float aq[CP / 4];
#pragma unroll
            for (int c4 = 0; c4 < CP / 4; ++c4) aq[c4] = qsf[(i16 * QLD + c4) * 4 + q4];
            float rq4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) rq4[r] = qn[4 * q4 + r];
            const int ntile = (n + 15) >> 4;
            // two candidate tiles per pass: their MFMA chains are independent and interleave on the pipe
            for (int tl = wave; tl < ntile; tl += 2 * KF_NW) {
                const int tl2 = tl + KF_NW;
                const bool two = tl2 < ntile;                       // wave-uniform
                const int cand0 = tl * 16 + i16, cc0 = min(cand0, n - 1);
                const int cand1 = tl2 * 16 + i16, cc1 = min(cand1, n - 1);
                kf_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c4 = 0; c4 < CP / 4; ++c4) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[c4], ff[(c4 * ns + cc0) * 4 + q4], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[c4], ff[(c4 * ns + cc1) * 4 + q4], acc1, 0, 0, 0);
                }
                const float rp0 = norms[cc0], rp1 = norms[cc1];
#pragma unroll
                for (int r = 0; r < 4; ++r) {                       // acc[r] = dot(query 4 q4 + r, candidate cand)
                    const float t0_ = rq4[r] - 2.0f * acc0[r];
                    const float d0 = (t0_ + rp0) + 0.0f;
                    dmat[(4 * q4 + r) * NP + cand0] = (cand0 < n) ? f32_to_ordered(d0) : 0xFFFFFFFFu;
                    if (two) {
                        const float t1_ = rq4[r] - 2.0f * acc1[r];
                        const float d1 = (t1_ + rp1) + 0.0f;
                        dmat[(4 * q4 + r) * NP + cand1] = (cand1 < n) ? f32_to_ordered(d1) : 0xFFFFFFFFu;
                    }
                }
            }
            __syncthreads();
#ifdef KNN_STAMPS
            { KN_T(t1); k_dot += t1 - t0; }
#endif
        }
        // Phase 2: wave w selects for queries w and w + 8
        for (int qv = q0 + wave; qv < q1; qv += 2 * KF_NW) {
            KN_T(t1);
            const int qa = __builtin_amdgcn_readfirstlane(qv);
            const bool has_b = qa + KF_NW < q1;
            const int qb = has_b ? qa + KF_NW : qa;
            uint32_t oda[R], odb[R];
            int cp[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int p = lane + 64 * r;
                cp[r] = p;
                oda[r] = (p < n) ? dmat[(qa - q0) * NP + p] : 0xFFFFFFFFu;
                odb[r] = (p < n) ? dmat[(qb - q0) * NP + p] : 0xFFFFFFFFu;
            }
            const size_t oa = ((size_t)cloud * m + qa) * k, ob = ((size_t)cloud * m + qb) * k;
            uint64_t* buf = sorted + (size_t)wave * KF_SCRATCH;
            const auto w2f = [](uint32_t w) { return ordered_to_f32(w); };
            const bool done_a = prefilter_rank<R, true>(oda, cp, buf, lane, k, 0xFFFFFFFEu, idx + oa, dist ? dist + oa : nullptr, w2f);
            const bool done_b = !has_b || prefilter_rank<R, true>(odb, cp, buf, lane, k, 0xFFFFFFFEu, idx + ob, dist ? dist + ob : nullptr, w2f);
            KN_T(t2);
            if (!(done_a && done_b)) {                                    // degenerate clouds: sort everything
                uint64_t ka[R], kb[R];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    ka[r] = (cp[r] < n) ? (((uint64_t)oda[r] << 32) | (uint32_t)cp[r]) : KEY_MAX;
                    kb[r] = (cp[r] < n) ? (((uint64_t)odb[r] << 32) | (uint32_t)cp[r]) : KEY_MAX;
                }
                sort_keys<R>(ka);
                sort_keys<R>(kb);
                uint64_t resa, resb;
                select_k2<R>(ka, kb, lane, k, resa, resb);
                if (lane < k) {
                    if (!done_a) {
                        idx[oa + lane] = (int)(uint32_t)resa;
                        if (dist) dist[oa + lane] = ordered_to_f32((uint32_t)(resa >> 32));
                    }
                    if (!done_b) {
                        idx[ob + lane] = (int)(uint32_t)resb;
                        if (dist) dist[ob + lane] = ordered_to_f32((uint32_t)(resb >> 32));
                    }
                }
            }
#ifdef KNN_STAMPS
            { KN_T(t3); k_sort += t2 - t1; k_sel += t3 - t2; k_nq += 2; }
#endif
        }
    } else {
    for (int qv = q0 + wave; qv < q1; qv += KF_NW) {
        KN_T(t0);
        const int qi = __builtin_amdgcn_readfirstlane(qv);
        const float4* __restrict__ qrow4 = qs + (qi - q0) * QLD;  // wave-uniform LDS address -> broadcast read
        float dot[R];
#pragma unroll
        for (int r = 0; r < R; ++r) dot[r] = 0.f;
        float rq = 0.f;
#pragma unroll 2   // partial unroll: a full unroll keeps CP/4 x R float4 LDS loads in flight (256 VGPRs, one block per CU)
        for (int c4 = 0; c4 < CP / 4; ++c4) {
            const float4 qv4 = qrow4[c4];
            const float q0v = qv4.x, q1v = qv4.y, q2v = qv4.z, q3v = qv4.w;
            rq = __builtin_fmaf(q0v, q0v, rq); rq = __builtin_fmaf(q1v, q1v, rq);
            rq = __builtin_fmaf(q2v, q2v, rq); rq = __builtin_fmaf(q3v, q3v, rq);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int p = lane + 64 * r;
                const float4 v = feats[c4 * ns + (p < n ? p : 0)];
                dot[r] = __builtin_fmaf(q0v, v.x, dot[r]); dot[r] = __builtin_fmaf(q1v, v.y, dot[r]);
                dot[r] = __builtin_fmaf(q2v, v.z, dot[r]); dot[r] = __builtin_fmaf(q3v, v.w, dot[r]);
            }
        }
        uint64_t key[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int p = lane + 64 * r;
            const float t0 = rq - 2.0f * dot[r];
            const float d = (t0 + rp[r]) + 0.0f;
            key[r] = (p < n) ? (((uint64_t)f32_to_ordered(d) << 32) | (uint32_t)p) : KEY_MAX;
        }
        KN_T(t1);
        sort_keys<R>(key);
        KN_T(t2);
        const uint64_t res = select_k<R>(key, sorted + (size_t)wave * R * 64, lane, k);
        if (lane < k) {
            const size_t o = ((size_t)cloud * m + qi) * k + lane;
            idx[o] = (int)(uint32_t)res;
            if (dist) dist[o] = ordered_to_f32((uint32_t)(res >> 32));
        }
#ifdef KNN_STAMPS
        { KN_T(t3); k_dot += t1 - t0; k_sort += t2 - t1; k_sel += t3 - t2; ++k_nq; }
#endif
    }
    }
#ifdef KNN_STAMPS
    if (blockIdx.x == 3 && blockIdx.y == 1 && lane == 0) {
        unsigned long long* st = reinterpret_cast<unsigned long long*>(idx + (size_t)gridDim.y * m * k) + wave * 5;
        st[0] = ts1 - ts0; st[1] = k_dot; st[2] = k_sort; st[3] = k_sel; st[4] = k_nq;
    }
#endif
}

template <int R>
static int launch_xyz_wave(int b, int n, int m, int k, const float* s, const float* q, int* idx, float* dist, int arith,
                           hipStream_t st, long sstride = 0) {
    if (sstride == 0) sstride = 3l * n;
    // queries per workgroup: a wave's queries are a serial chain, so keep it short while the total stays >= ~4 waves/SIMD
    const int qpb = ((long)b * m >= 32768) ? 32 : 16;
    dim3 grid((m + qpb - 1) / qpb, b);
    if (arith & DISPU_ARITH_CONTRACT)
        hipLaunchKernelGGL((knn_xyz_wave_kernel<R, true>), grid, dim3(256), 0, st, n, m, k, qpb, s, sstride, q, idx, dist);
    else
        hipLaunchKernelGGL((knn_xyz_wave_kernel<R, false>), grid, dim3(256), 0, st, n, m, k, qpb, s, sstride, q, idx, dist);
    return (int)hipGetLastError();
}

template <int R, int CP>
static int launch_feat_wave(int b, int n, int m, int c, int k, int ldp, int ldq, const float* p, const float* q, float* dist,
                            int* idx, hipStream_t st, long pstride = 0) {
    if (pstride == 0) pstride = (long)n * ldp;
    const int qpb = 16;         // 2 queries per wave; the cloud's features are re-staged per workgroup (L2-resident)
    dim3 grid((m + qpb - 1) / qpb, b);
    const size_t lds = (size_t)(CP / 4) * (n + 1) * 16 + (((size_t)n * 4 + 15) & ~15ull) + (R <= 4 ? (size_t)KF_NW * KF_SCRATCH * 8 : (size_t)KF_NW * R * 64 * 8) + (size_t)qpb * (CP + 4) * 4 + (R <= 4 ? (size_t)16 * (64 * R + 4) * 4 + 64 : (size_t)0);
    static DevOnce attr;                 // per instantiation: opt in to more than 64 KB of dynamic LDS
    if (attr.needed()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(knn_feat_wave_kernel<R, CP>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr.done();
    }
    hipLaunchKernelGGL((knn_feat_wave_kernel<R, CP>), grid, dim3(64 * KF_NW), lds, st, n, m, c, k, qpb, ldp, ldq, pstride, p, q, dist, idx);
    return (int)hipGetLastError();
}

// Returns -1 when the shape is outside the fast path (caller falls back to the lane-per-query kernels).
int knn_xyz_wave_dispatch(int b, int n, int m, int k, const float* s, const float* q, int* idx, float* dist, int arith,
                          hipStream_t st) {
    if (n > 1024 || k > 64) return -1;
    if (n <= 64) return launch_xyz_wave<1>(b, n, m, k, s, q, idx, dist, arith, st);
    if (n <= 128) return launch_xyz_wave<2>(b, n, m, k, s, q, idx, dist, arith, st);
    if (n <= 256) return launch_xyz_wave<4>(b, n, m, k, s, q, idx, dist, arith, st);
    if (n <= 512) return launch_xyz_wave<8>(b, n, m, k, s, q, idx, dist, arith, st);
    return launch_xyz_wave<16>(b, n, m, k, s, q, idx, dist, arith, st);
}

// ---- 1024 < n <= 4096 (round 3): ONE pass per query over the whole cloud ---------------------------------------------------------
// The chunked path below runs the n <= 1024 kernel once per 1024-candidate chunk and merges: every chunk pays the selection again
// (lane minima, 64-lane bitonic sort, 16 ballot compactions, rank: ~240 of its ~490 wave instructions per query) and the merge is a
// third launch -- 413 us at (32, 4096, 4096, 16).  Here the cloud's coordinates sit in LDS (SoA, 48 KB; candidates past n are +inf), a
// wave owns a query, every lane evaluates 64 candidates (lane rotation as above), and the threshold / compaction / rank runs ONCE per
// query; a 64-candidate row without a survivor costs a compare and a scalar branch.  Same keys, same order: (distance, index).
constexpr int KXL_N = 4096, KXL_R = KXL_N / 64, KXL_CAP = 128;

__device__ __forceinline__ uint32_t kxl_wave_min_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, o, 64));
    return v;
}

constexpr int KXL_W = 8;                                                   // waves per workgroup: two workgroups (2 x 57 KB of LDS) give a CU 16 waves
template <bool FMA>
__global__ __launch_bounds__(64 * KXL_W, 4) void knn_xyz_lds_kernel(int n, int m, int k, int qpb, const float* __restrict__ support,
                                                          const float* __restrict__ query, int* __restrict__ idx, float* __restrict__ dist) {
    extern __shared__ __attribute__((aligned(16))) float kxl_lds[];
    float* cxs = kxl_lds;
    float* cys = cxs + KXL_N;
    float* czs = cys + KXL_N;
    uint64_t* bufs = reinterpret_cast<uint64_t*>(czs + KXL_N);            // [KXL_W waves][KXL_CAP + 4]
    const int cloud = blockIdx.y, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const float* __restrict__ s = support + (size_t)cloud * n * 3;
    const float* __restrict__ q = query + (size_t)cloud * m * 3;
    // LDS slot (r, l) holds candidate 64 r + ((l + 17 r) & 63): the lane rotation of the n <= 1024 kernel, applied when the cloud is
    // staged, so that the distance loop reads slot 64 r + lane with an immediate offset (rotating in the loop cost 64 index registers)
    for (int e = threadIdx.x; e < KXL_N; e += 64 * KXL_W) {
        const bool ok = e < n;
        const int r = e >> 6, l = ((e & 63) - 17 * r) & 63;
        cxs[64 * r + l] = ok ? s[e * 3 + 0] : __builtin_inff();
        cys[64 * r + l] = ok ? s[e * 3 + 1] : __builtin_inff();
        czs[64 * r + l] = ok ? s[e * 3 + 2] : __builtin_inff();
    }
    __syncthreads();
    uint64_t* buf = bufs + wave * (KXL_CAP + 4);
    const int q0 = blockIdx.x * qpb, q1 = min(m, q0 + qpb);
    for (int qv = q0 + wave; qv < q1; qv += KXL_W) {
        const int qi = __builtin_amdgcn_readfirstlane(qv);
        const float qx = q[qi * 3 + 0], qy = q[qi * 3 + 1], qz = q[qi * 3 + 2];
        uint32_t od[KXL_R];
        uint32_t dmin = 0xFFFFFFFFu;
#pragma unroll
        for (int rb = 0; rb < KXL_R; rb += 8) {
            __builtin_amdgcn_sched_barrier(0);        // 8 candidates' LDS reads in flight at a time: hoisting all 192 spilled the kernel
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = rb + u;
                od[r] = __float_as_uint(sqdist3<FMA>(qx - cxs[64 * r + lane], qy - cys[64 * r + lane], qz - czs[64 * r + lane]));   // +inf: padding
                dmin = min(dmin, od[r]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // k distinct candidates are <= the k-th smallest lane minimum; never admit the padding (+inf)
        uint32_t T = (uint32_t)__builtin_amdgcn_readlane((int)wave_bitonic_sort_u32(dmin, lane), k - 1);
        T = min(T, 0x7F7FFFFFu);
        int cnt = 0;
#pragma unroll
        for (int r = 0; r < KXL_R; ++r) {
            const bool flag = od[r] <= T;
            const unsigned long long mk = __ballot(flag);
            if (mk) {                                                     // wave-uniform: most rows have no survivor
                const int pos = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u));
                int lq = lane;
                asm volatile("" : "+v"(lq));                              // keeps the 64 candidate ids from being hoisted out of the query loop
                if (flag && cnt + pos < KXL_CAP) (buf + cnt)[pos] = ((uint64_t)od[r] << 32) | (uint32_t)(64 * r + ((lq + 17 * r) & 63));
                cnt += __popcll(mk);
            }
        }
        const size_t o = ((size_t)cloud * m + qi) * k;
        if (cnt <= KXL_CAP) {
            // rank of every survivor among the survivors (keys are distinct): the survivor of rank t < k IS result t
            if (lane < 4) buf[cnt + lane] = KEY_MAX;                      // the loop reads in fours
            const uint64_t m0 = (lane < cnt) ? buf[lane] : KEY_MAX;
            const uint64_t m1 = (lane + 64 < cnt) ? buf[lane + 64] : KEY_MAX;
            int r0 = 0, r1 = 0;
            for (int j = 0; j < cnt; j += 4) {
                const uint64_t a0 = buf[j], a1 = buf[j + 1], a2 = buf[j + 2], a3 = buf[j + 3];
                r0 += (int)(a0 < m0) + (int)(a1 < m0) + (int)(a2 < m0) + (int)(a3 < m0);
                r1 += (int)(a0 < m1) + (int)(a1 < m1) + (int)(a2 < m1) + (int)(a3 < m1);
            }
            if (lane < cnt && r0 < k) {
                idx[o + r0] = (int)(uint32_t)m0;
                if (dist) dist[o + r0] = __uint_as_float((uint32_t)(m0 >> 32));
            }
            if (lane + 64 < cnt && r1 < k) {
                idx[o + r1] = (int)(uint32_t)m1;
                if (dist) dist[o + r1] = __uint_as_float((uint32_t)(m1 >> 32));
            }
        } else {
            // degenerate clouds (more than 128 candidates within the bound: many equal distances): k rounds of a wave arg-min over the
            // keys larger than the last one taken -- slow, exact, rare
            uint64_t last = 0;
            bool first = true;
            for (int t = 0; t < k; ++t) {
                uint64_t best = KEY_MAX;
#pragma unroll 1
                for (int r = 0; r < KXL_R; ++r) {                         // distances recomputed (same arithmetic): no register array here
                    const int p = 64 * r + ((lane + 17 * r) & 63);
                    const uint32_t w = __float_as_uint(sqdist3<FMA>(qx - cxs[64 * r + lane], qy - cys[64 * r + lane], qz - czs[64 * r + lane]));
                    const uint64_t key = ((uint64_t)w << 32) | (uint32_t)p;
                    if (w <= 0x7F7FFFFFu && (first || key > last) && key < best) best = key;
                }
                const uint32_t hi = kxl_wave_min_u32((uint32_t)(best >> 32));
                const uint32_t lo = kxl_wave_min_u32(((uint32_t)(best >> 32) == hi) ? (uint32_t)best : 0xFFFFFFFFu);
                last = ((uint64_t)hi << 32) | lo;
                first = false;
                if (lane == 0) {
                    idx[o + t] = (int)lo;
                    if (dist) dist[o + t] = __uint_as_float(hi);
                }
            }
        }
    }
}

// -1: shape outside this path
int knn_xyz_lds_dispatch(int b, int n, int m, int k, const float* s, const float* q, int* idx, float* dist, int arith, hipStream_t st) {
    if (n <= 1024 || n > KXL_N || k > 32) return -1;
    const size_t bytes = (size_t)3 * KXL_N * sizeof(float) + (size_t)KXL_W * (KXL_CAP + 4) * sizeof(uint64_t);
    static DevOnce attr;
    if (attr.needed()) {
        DISPU_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(knn_xyz_lds_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        DISPU_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(knn_xyz_lds_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        attr.done();
    }
    const int qpb = 16 * KXL_W;                                           // 16 queries per wave behind one staging of the cloud
    dim3 grid((m + qpb - 1) / qpb, b);
    if (arith & DISPU_ARITH_CONTRACT) hipLaunchKernelGGL((knn_xyz_lds_kernel<true>), grid, dim3(64 * KXL_W), bytes, st, n, m, k, qpb, s, q, idx, dist);
    else hipLaunchKernelGGL((knn_xyz_lds_kernel<false>), grid, dim3(64 * KXL_W), bytes, st, n, m, k, qpb, s, q, idx, dist);
    return (int)hipGetLastError();
}

// ---- 1024 < n <= 8192: the cloud is cut into nc balanced chunks of <= 1024 candidates, the wave kernel above finds every
// chunk's k nearest (exact, ties -> lower index) into caller scratch, and knn_xyz_merge_kernel merges the nc sorted lists by
// (distance, global index).  Any global k-nearest neighbour is among its chunk's k nearest, so the result equals one scan
// over the whole cloud.  The lane-per-query kernel of knn.hip needs 0.9 ms at (32, 4096, 4096, 16) - the second generator
// pass of 16x upsampling, BASELINE configs[3] - because some lane of a wave accepts nearly every candidate and the 16-slot
// sorted insert then runs for all 64.
constexpr int KX_MAXCHUNKS = 16;

__global__ __launch_bounds__(256) void knn_xyz_merge_kernel(long nq, int k, int nc, int cs, const int* __restrict__ cidx,
                                                            const float* __restrict__ cdist, int* __restrict__ idx,
                                                            float* __restrict__ dist) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    int h[KX_MAXCHUNKS];
    float hd[KX_MAXCHUNKS];
#pragma unroll
    for (int c = 0; c < KX_MAXCHUNKS; ++c) { h[c] = 0; hd[c] = (c < nc) ? cdist[((size_t)c * nq + q) * k] : __builtin_inff(); }
    for (int t = 0; t < k; ++t) {
        int bc = 0;
        float bd = hd[0];
#pragma unroll
        for (int c = 1; c < KX_MAXCHUNKS; ++c)
            if (hd[c] < bd) { bd = hd[c]; bc = c; }          // equal heads: the lower chunk holds the lower global index
        int bi = 0;
#pragma unroll
        for (int c = 0; c < KX_MAXCHUNKS; ++c)
            if (c == bc) {
                bi = cidx[((size_t)c * nq + q) * k + h[c]] + c * cs;
                ++h[c];
                hd[c] = (h[c] < k) ? cdist[((size_t)c * nq + q) * k + h[c]] : __builtin_inff();
            }
        idx[q * k + t] = bi;
        if (dist) dist[q * k + t] = bd;
    }
}

size_t knn_xyz_chunked_scratch(int b, int n, int m, int k) {
    if (n <= 1024 || n > 1024 * 8 || k > 32) return 0;
    const int nc = (n + 1023) / 1024;
    return (size_t)nc * b * m * k * 8;
}

// -1: shape outside this path / scratch too small
int knn_xyz_chunked_dispatch(int b, int n, int m, int k, const float* s, const float* q, int* idx, float* dist, void* scratch,
                             size_t scratch_bytes, int arith, hipStream_t st) {
    {
        const int rl = knn_xyz_lds_dispatch(b, n, m, k, s, q, idx, dist, arith, st);      // 1024 < n <= 4096: one pass, no scratch
        if (rl >= 0) return rl;
    }
    const size_t need = knn_xyz_chunked_scratch(b, n, m, k);
    if (need == 0 || !scratch || scratch_bytes < need) return -1;
    const int nc = (n + 1023) / 1024, cs = (n + nc - 1) / nc;
    if (n - (nc - 1) * cs < k) return -1;
    const size_t nq = (size_t)b * m;
    int* cidx = reinterpret_cast<int*>(scratch);
    float* cdist = reinterpret_cast<float*>(cidx + (size_t)nc * nq * k);
    for (int c = 0; c < nc; ++c) {
        const int len = (c + 1 < nc) ? cs : n - c * cs;
        const float* sc = s + (size_t)c * cs * 3;
        int rc;
        if (len <= 512) rc = launch_xyz_wave<8>(b, len, m, k, sc, q, cidx + (size_t)c * nq * k, cdist + (size_t)c * nq * k, arith, st, 3l * n);
        else rc = launch_xyz_wave<16>(b, len, m, k, sc, q, cidx + (size_t)c * nq * k, cdist + (size_t)c * nq * k, arith, st, 3l * n);
        if (rc != 0) return rc;
    }
    hipLaunchKernelGGL(knn_xyz_merge_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, (long)nq, k, nc, cs, cidx, cdist, idx, dist);
    return (int)hipGetLastError();
}

template <int CP>
static int feat_wave_r(int b, int n, int m, int c, int k, int ldp, int ldq, const float* p, const float* q, float* dist,
                       int* idx, hipStream_t st, long pstride) {
    if (n <= 64) return launch_feat_wave<1, CP>(b, n, m, c, k, ldp, ldq, p, q, dist, idx, st, pstride);
    if (n <= 128) return launch_feat_wave<2, CP>(b, n, m, c, k, ldp, ldq, p, q, dist, idx, st, pstride);
    if (n <= 256) return launch_feat_wave<4, CP>(b, n, m, c, k, ldp, ldq, p, q, dist, idx, st, pstride);
    return launch_feat_wave<8, CP>(b, n, m, c, k, ldp, ldq, p, q, dist, idx, st, pstride);
}

int knn_feat_big_dispatch(int b, int n, int m, int c, int k, int ldp, int ldq, const float* p, const float* q, float* dist, int* idx,
                          hipStream_t st);
int knn_feat_wave_dispatch(int b, int n, int m, int c, int k, int ldp, int ldq, const float* p, const float* q, float* dist,
                           int* idx, hipStream_t st, long pstride = 0) {
    if (n > 512 && pstride == 0) return knn_feat_big_dispatch(b, n, m, c, k, ldp, ldq, p, q, dist, idx, st);
    if (n > 512 || c > 64 || k > 64) return -1;
    const int cp = (c + 3) & ~3;
    const int r = n <= 64 ? 1 : (n <= 128 ? 2 : (n <= 256 ? 4 : 8));
    const size_t lds = (size_t)(cp / 4) * (n + 1) * 16 + (((size_t)n * 4 + 15) & ~15ull) + (r <= 4 ? (size_t)KF_NW * KF_SCRATCH * 8 + (size_t)16 * (64 * r + 4) * 4 + 64 : (size_t)KF_NW * r * 64 * 8) + (size_t)16 * (cp + 4) * 4;
    if (lds > 160 * 1024) return -1;
    if (c <= 4) return feat_wave_r<4>(b, n, m, c, k, ldp, ldq, p, q, dist, idx, st, pstride);
    if (c <= 8) return feat_wave_r<8>(b, n, m, c, k, ldp, ldq, p, q, dist, idx, st, pstride);
    if (c <= 16) return feat_wave_r<16>(b, n, m, c, k, ldp, ldq, p, q, dist, idx, st, pstride);
    if (c <= 24) return feat_wave_r<24>(b, n, m, c, k, ldp, ldq, p, q, dist, idx, st, pstride);
    if (c <= 32) return feat_wave_r<32>(b, n, m, c, k, ldp, ldq, p, q, dist, idx, st, pstride);
    if (c <= 48) return feat_wave_r<48>(b, n, m, c, k, ldp, ldq, p, q, dist, idx, st, pstride);
    return feat_wave_r<64>(b, n, m, c, k, ldp, ldq, p, q, dist, idx, st, pstride);
}

// ---- feature space, 512 < n <= 1024, c <= 48 (round 5): ONE pass per query over the whole cloud -----------------------------------
// The chunked path below cuts such a cloud into four 256-candidate chunks: every chunk stages its features again for every 16
// queries, pays the threshold / compaction / rank again, and a merge kernel follows (637 us for the three C = 48 and the one C = 24
// call of a 1024-point dense-block pass).  Here a workgroup of 8 waves owns 64 queries.  The cloud's features pass through LDS in
// channel halves of 24 (98 KB each, the channel-quad-major image of knn_feat_wave_kernel); the 64 x n dot products accumulate on the
// matrix pipe ACROSS the halves (v_mfma_f32_16x16x4_f32 continues the ascending-channel fmaf chain bit for bit: 4 query tiles x 8
// candidate tiles = 128 accumulator registers per lane), the norms continue their chains the same way.  Then, 16 queries at a time,
// the distance words (rq - 2 dot) + rp go to LDS over the dead feature image and every wave runs the prefilter + rank ONCE over the
// n candidates of its two queries.  Same words, same (distance, index) order as the chunked path and the lane-per-query kernel.
constexpr int KFB_HC = 24, KFB_QB = 64, KFB_W = 8, KFB_NMAX = 1024, KFB_NP = KFB_NMAX + 4, KFB_QLD = KFB_HC / 4 + 1;
constexpr size_t KFB_FEATS = (size_t)(KFB_HC / 4) * (KFB_NMAX + 1) * 16;                 // 98400 B, reused: dmat [16][NP] words + the full-sort scratch
constexpr size_t KFB_SORT = (size_t)KFB_W * (16 * 64 + 4) * 8;                           // degenerate clouds: every key of two... of one query per wave
constexpr size_t KFB_REGION_A = ((size_t)16 * KFB_NP * 4 + KFB_SORT > KFB_FEATS) ? (size_t)16 * KFB_NP * 4 + KFB_SORT : KFB_FEATS;
constexpr size_t KFB_LDS = KFB_REGION_A + (size_t)KFB_NMAX * 4 + (size_t)KFB_QB * KFB_QLD * 16 + (size_t)KFB_QB * 4 + (size_t)KFB_W * (128 + 4) * 8;

__global__ __launch_bounds__(64 * KFB_W) void knn_feat_big_kernel(int n, int m, int c, int k, int ldp, int ldq,
                                                                  const float* __restrict__ points, const float* __restrict__ queries,
                                                                  float* __restrict__ dist, int* __restrict__ idx) {
    typedef float kf_f32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ns = KFB_NMAX + 1;
    float4* feats = reinterpret_cast<float4*>(smem);                                     // [6][ns] (matrix phase)
    uint32_t* dmat = reinterpret_cast<uint32_t*>(smem);                                  // [16][NP] (selection phase)
    uint64_t* fullsort = reinterpret_cast<uint64_t*>(smem + (size_t)16 * KFB_NP * 4);    // [W][16 * 64 + 4]
    float* norms = reinterpret_cast<float*>(smem + KFB_REGION_A);                        // [NMAX]
    float4* qs = reinterpret_cast<float4*>(norms + KFB_NMAX);                            // [QB][QLD]
    float* qn = reinterpret_cast<float*>(qs + KFB_QB * KFB_QLD);                         // [QB]
    uint64_t* pbuf = reinterpret_cast<uint64_t*>(qn + KFB_QB);                           // [W][128 + 4]
    const int cloud = blockIdx.y, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int i16 = lane & 15, q4 = lane >> 4;
    const float* __restrict__ sp = points + (size_t)cloud * n * ldp;
    const float* __restrict__ qp = queries + (size_t)cloud * m * ldq;
    const int q0 = blockIdx.x * KFB_QB;
    const bool vecp = ((c & 3) == 0) && ((ldp & 3) == 0) && ((((uintptr_t)sp) & 15) == 0);
    const bool vecq = ((c & 3) == 0) && ((ldq & 3) == 0) && ((((uintptr_t)qp) & 15) == 0);
    const auto load4 = [&](const float* row, int ch, bool vec) -> float4 {             // channels ch .. ch + 3 of a row, zero past c
        float4 v = {0.f, 0.f, 0.f, 0.f};
        if (vec) { if (ch < c) v = *reinterpret_cast<const float4*>(row + ch); }
        else {
            if (ch + 0 < c) v.x = row[ch + 0];
            if (ch + 1 < c) v.y = row[ch + 1];
            if (ch + 2 < c) v.z = row[ch + 2];
            if (ch + 3 < c) v.w = row[ch + 3];
        }
        return v;
    };
    kf_f32x4 acc[4][8];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[t][j] = kf_f32x4{0.f, 0.f, 0.f, 0.f};
    const int nh = (c + KFB_HC - 1) / KFB_HC;                                            // 1 or 2
    for (int h = 0; h < nh; ++h) {
        if (h) __syncthreads();                                                          // the previous half's operand reads are done
        const int ch0 = h * KFB_HC;
        {   // batches of six independent 16-byte loads per thread, then the six LDS stores
            constexpr int UB = 6, Q4 = KFB_HC / 4;
            const int total = Q4 * n;
            for (int e0 = threadIdx.x; e0 < total; e0 += UB * 64 * KFB_W) {
                float4 v[UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int e = min(e0 + u * 64 * KFB_W, total - 1);
                    const int p = e / Q4, c4 = e - p * Q4;
                    v[u] = load4(sp + (size_t)p * ldp, ch0 + 4 * c4, vecp);
                }
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int e = e0 + u * 64 * KFB_W;
                    const int p = e / Q4, c4 = e - p * Q4;
                    if (e < total) feats[c4 * ns + p] = v[u];
                }
            }
            for (int e = threadIdx.x; e < KFB_QB * Q4; e += 64 * KFB_W) {
                const int ql = e / Q4, c4 = e - ql * Q4;
                const int qg = min(q0 + ql, m - 1);
                qs[ql * KFB_QLD + c4] = load4(qp + (size_t)qg * ldq, ch0 + 4 * c4, vecq);
            }
        }
        __syncthreads();
        // |F_p|^2 and |q|^2 continue their ascending-channel chains over the halves (read after the barrier that ends the matrix phase)
        for (int p = threadIdx.x; p < n; p += 64 * KFB_W) {
            float r = h ? norms[p] : 0.f;
#pragma unroll
            for (int c4 = 0; c4 < KFB_HC / 4; ++c4) {
                const float4 v = feats[c4 * ns + p];
                r = __builtin_fmaf(v.x, v.x, r); r = __builtin_fmaf(v.y, v.y, r);
                r = __builtin_fmaf(v.z, v.z, r); r = __builtin_fmaf(v.w, v.w, r);
            }
            norms[p] = r;
        }
        if (threadIdx.x < KFB_QB) {
            float r = h ? qn[threadIdx.x] : 0.f;
#pragma unroll
            for (int c4 = 0; c4 < KFB_HC / 4; ++c4) {
                const float4 v = qs[threadIdx.x * KFB_QLD + c4];
                r = __builtin_fmaf(v.x, v.x, r); r = __builtin_fmaf(v.y, v.y, r);
                r = __builtin_fmaf(v.z, v.z, r); r = __builtin_fmaf(v.w, v.w, r);
            }
            qn[threadIdx.x] = r;
        }
        // matrix phase: wave w owns candidate tiles w, w + 8, ..; A = 16 query rows per tile, one instruction per channel quad
        const float* qsf = reinterpret_cast<const float*>(qs);
        const float* ff = reinterpret_cast<const float*>(feats);
        int cc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) cc[j] = min((wave + 8 * j) * 16 + i16, n - 1);
#pragma unroll
        for (int c4 = 0; c4 < KFB_HC / 4; ++c4) {
            float a[4], bq[8];
#pragma unroll
            for (int t = 0; t < 4; ++t) a[t] = qsf[((16 * t + i16) * KFB_QLD + c4) * 4 + q4];
#pragma unroll
            for (int j = 0; j < 8; ++j) bq[j] = ff[(c4 * ns + cc[j]) * 4 + q4];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], bq[j], acc[t][j], 0, 0, 0);
        }
    }
    __syncthreads();                                                                     // feature image dead, norms complete
    float rpj[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) rpj[j] = norms[min((wave + 8 * j) * 16 + i16, n - 1)];
    uint64_t* buf = pbuf + (size_t)wave * (128 + 4);
    const auto w2f = [](uint32_t w) { return ordered_to_f32(w); };
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (t) __syncthreads();                                                          // the previous batch's words have been read
        float rq4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) rq4[r] = qn[16 * t + 4 * q4 + r];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int cand = (wave + 8 * j) * 16 + i16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {                                                // acc[r] = dot(query 4 q4 + r, candidate cand)
                const float t0 = rq4[r] - 2.0f * acc[t][j][r];
                const float d = (t0 + rpj[j]) + 0.0f;
                dmat[(4 * q4 + r) * KFB_NP + cand] = (cand < n) ? f32_to_ordered(d) : 0xFFFFFFFFu;
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int s = 0; s < 2; ++s) {
            const int ql = wave + 8 * s;                                                 // query of this batch
            const int qg = __builtin_amdgcn_readfirstlane(q0 + 16 * t + ql);
            if (qg >= m) continue;
            uint32_t od[16];
            int cp[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                cp[r] = lane + 64 * r;
                od[r] = dmat[ql * KFB_NP + lane + 64 * r];                               // words past n are all-ones
            }
            const size_t o = ((size_t)cloud * m + qg) * k;
            if (!prefilter_rank<16, true>(od, cp, buf, lane, k, 0xFFFFFFFEu, idx + o, dist ? dist + o : nullptr, w2f)) {
                uint64_t key[16];                                                        // degenerate clouds: sort everything
#pragma unroll
                for (int r = 0; r < 16; ++r) key[r] = (cp[r] < n) ? (((uint64_t)od[r] << 32) | (uint32_t)cp[r]) : KEY_MAX;
                sort_keys<16>(key);
                const uint64_t res = select_k<16>(key, fullsort + (size_t)wave * (16 * 64 + 4), lane, k);
                if (lane < k) {
                    idx[o + lane] = (int)(uint32_t)res;
                    if (dist) dist[o + lane] = ordered_to_f32((uint32_t)(res >> 32));
                }
            }
        }
    }
}

// -1: shape outside this path
int knn_feat_big_dispatch(int b, int n, int m, int c, int k, int ldp, int ldq, const float* p, const float* q, float* dist, int* idx,
                          hipStream_t st) {
    if (n <= 512 || n > KFB_NMAX || c > 2 * KFB_HC || k > 64 || k > n) return -1;
    static DevOnce attr;
    if (attr.needed()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(knn_feat_big_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr.done();
    }
    static_assert(KFB_LDS <= 160 * 1024, "one workgroup per CU");
    hipLaunchKernelGGL(knn_feat_big_kernel, dim3((m + KFB_QB - 1) / KFB_QB, b), dim3(64 * KFB_W), KFB_LDS, st, n, m, c, k, ldp, ldq, p, q, dist, idx);
    return (int)hipGetLastError();
}

// ---- feature space, 512 < n <= 4096 (the second pass of 16x upsampling runs the dense blocks on 1024-point patches): balanced
// chunks of <= 256 candidates through the wave kernel above (its R <= 4 form: dot products on the matrix pipe, 49 KB of staged
// features per workgroup; 512-candidate chunks re-stage 98 KB for every 16 queries and were slower: 303 vs 466 us unchunked),
// merged like the xyz lists.  The lane-per-query kernel needed
// 470 us per call at (32, 1024, 1024, 48, 17) - a quarter of that pass.
constexpr int KF_CHUNK = 256;
size_t knn_feat_chunked_scratch(int b, int n, int m, int c, int k) {
    if (n <= 512 || n > KF_CHUNK * KX_MAXCHUNKS || c > 64 || k > 32) return 0;
    const int nc = (n + KF_CHUNK - 1) / KF_CHUNK;
    return (size_t)nc * b * m * k * 8;
}

int knn_feat_chunked_dispatch(int b, int n, int m, int c, int k, int ldp, int ldq, const float* p, const float* q, float* dist, int* idx,
                              void* scratch, size_t scratch_bytes, hipStream_t st) {
    {
        const int rb = knn_feat_big_dispatch(b, n, m, c, k, ldp, ldq, p, q, dist, idx, st);      // 512 < n <= 1024, c <= 48: one pass, no scratch
        if (rb >= 0) return rb;
    }
    const size_t need = knn_feat_chunked_scratch(b, n, m, c, k);
    if (need == 0 || !scratch || scratch_bytes < need) return -1;
    const int nc = (n + KF_CHUNK - 1) / KF_CHUNK, cs = (n + nc - 1) / nc;
    if (n - (nc - 1) * cs < k) return -1;
    const size_t nq = (size_t)b * m;
    int* cidx = reinterpret_cast<int*>(scratch);
    float* cdist = reinterpret_cast<float*>(cidx + (size_t)nc * nq * k);
    for (int ch = 0; ch < nc; ++ch) {
        const int len = (ch + 1 < nc) ? cs : n - ch * cs;
        const int rc = knn_feat_wave_dispatch(b, len, m, c, k, ldp, ldq, p + (size_t)ch * cs * ldp, q, cdist + (size_t)ch * nq * k,
                                              cidx + (size_t)ch * nq * k, st, (long)n * ldp);
        if (rc != 0) return rc;                             // incl. -1: a chunk the wave kernel cannot take
    }
    hipLaunchKernelGGL(knn_xyz_merge_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, (long)nq, k, nc, cs, cidx, cdist, idx, dist);
    return (int)hipGetLastError();
}

}  // namespace dispu
