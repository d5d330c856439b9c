// General exact k-NN selection for the shapes outside the register-resident fast paths of knn.hip / knn_wave.hip / cloud.hip:
// k > 32 (up to 4096), c > 128 channels (up to 4096), any n.  The reference has no such limits: nanoflann takes any K
// (libs/nearest_neighbors/knn_.cxx:104-135), knn_point / knn_point_2 are tf.nn.top_k over a full distance matrix
// (tf_ops/grouping/tf_grouping.py:95-141), extract_knn_patch is sklearn over the whole cloud (Common/pc_util.py:83-92).
//
// One workgroup per query, no scratch memory: the k-th smallest distance is found by a 4-pass MSB-first radix select over
// the ORDERED distance bits (the distances are recomputed in every pass - 5 evaluations per pair instead of an n-float
// buffer per query), then one more pass collects every candidate below the threshold plus the first (in index order) of
// the candidates equal to it, and a bitonic network sorts the k survivors by (distance, index).  Result: ascending distance,
// ties -> lower index - tf.nn.top_k's rule and the order the fast paths produce.  Distances are evaluated with exactly the
// arithmetic of the fast paths (xyz: sqdist3 PLAIN / CONTRACT; knn_point: sum((p - q)^2) left to right, every op rounded;
// knn_point_2: (rq - 2 q.p) + rp with ascending-channel fmaf chains), so both paths agree bit for bit where they overlap.
#include "common.h"

namespace dispu {

enum { KG_XYZ_PLAIN = 0, KG_XYZ_FMA = 1, KG_SQ = 2, KG_GEMM = 3 };
constexpr int KG_MAXK = 4096, KG_MAXC = 4096;

template <int MODE>
__device__ __forceinline__ float kg_distance(const float* __restrict__ p, const float* qrow, int c, float rq, bool vec4) {
    if constexpr (MODE == KG_XYZ_PLAIN || MODE == KG_XYZ_FMA) {
        return sqdist3<MODE == KG_XYZ_FMA>(qrow[0] - p[0], qrow[1] - p[1], qrow[2] - p[2]) + 0.0f;
    } else if constexpr (MODE == KG_SQ) {
        float d = 0.f;
        if (vec4) {
            for (int l = 0; l < c; l += 4) {
                const float4 v = *reinterpret_cast<const float4*>(p + l);
                float df;
                df = v.x - qrow[l + 0]; d = d + df * df;
                df = v.y - qrow[l + 1]; d = d + df * df;
                df = v.z - qrow[l + 2]; d = d + df * df;
                df = v.w - qrow[l + 3]; d = d + df * df;
            }
        } else {
            for (int l = 0; l < c; ++l) { const float df = p[l] - qrow[l]; d = d + df * df; }
        }
        return d + 0.0f;
    } else {
        float dot = 0.f, rp = 0.f;
        if (vec4) {
            for (int l = 0; l < c; l += 4) {
                const float4 v = *reinterpret_cast<const float4*>(p + l);
                dot = __builtin_fmaf(qrow[l + 0], v.x, dot); rp = __builtin_fmaf(v.x, v.x, rp);
                dot = __builtin_fmaf(qrow[l + 1], v.y, dot); rp = __builtin_fmaf(v.y, v.y, rp);
                dot = __builtin_fmaf(qrow[l + 2], v.z, dot); rp = __builtin_fmaf(v.z, v.z, rp);
                dot = __builtin_fmaf(qrow[l + 3], v.w, dot); rp = __builtin_fmaf(v.w, v.w, rp);
            }
        } else {
            for (int l = 0; l < c; ++l) { dot = __builtin_fmaf(qrow[l], p[l], dot); rp = __builtin_fmaf(p[l], p[l], rp); }
        }
        const float rq_m2dot = rq - 2.0f * dot;
        return (rq_m2dot + rp) + 0.0f;            // + 0.0f: -0 -> +0, so equal floats have equal ordered bits
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void knn_general_kernel(int n, int m, int c, int k, int kpad, long ldp, long ldq,
                                                           const float* __restrict__ points, const float* __restrict__ queries,
                                                           float* __restrict__ dist, int* __restrict__ idx, int neg, int vec4) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);            // [kpad]
    float* qrow = reinterpret_cast<float*>(keys + kpad);                                 // [c rounded up to 4]
    __shared__ unsigned hist[256];
    __shared__ unsigned sel[3];                   // prefix, remaining rank, survivors written
    __shared__ unsigned wcnt[4];
    __shared__ float srq;
    const int cloud = blockIdx.y, qi = blockIdx.x, tid = threadIdx.x;
    const float* __restrict__ pc = points + (size_t)cloud * n * ldp;
    const float* __restrict__ q = queries + ((size_t)cloud * m + qi) * ldq;
    for (int l = tid; l < c; l += 256) qrow[l] = q[l];
    if (tid == 0) { sel[0] = 0u; sel[1] = (unsigned)k; sel[2] = 0u; }
    __syncthreads();
    if constexpr (MODE == KG_GEMM) {
        if (tid == 0) {
            float r = 0.f;
            for (int l = 0; l < c; ++l) r = __builtin_fmaf(qrow[l], qrow[l], r);
            srq = r;
        }
        __syncthreads();
    }
    const float rq = (MODE == KG_GEMM) ? srq : 0.f;
    auto key_of = [&](int t) -> unsigned { return f32_to_ordered(kg_distance<MODE>(pc + (size_t)t * ldp, qrow, c, rq, vec4 != 0)); };

    // ---- radix select: after pass p the top 8 (p + 1) bits of the k-th smallest key are known
    unsigned mask = 0u;
    for (int shift = 24; shift >= 0; shift -= 8) {
        hist[tid] = 0u;
        __syncthreads();
        const unsigned prefix = sel[0];
        for (int t = tid; t < n; t += 256) {
            const unsigned key = key_of(t);
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            unsigned rank = sel[1], cum = 0u, bin = 0u;
            for (; bin < 256u; ++bin) {
                if (cum + hist[bin] >= rank) break;
                cum += hist[bin];
            }
            sel[0] = prefix | (bin << shift);
            sel[1] = rank - cum;                  // rank of the k-th inside the chosen bin
        }
        mask |= 255u << shift;
        __syncthreads();
    }
    const unsigned T = sel[0], quota = sel[1];    // take every key < T and the first `quota` (index order) keys == T

    // ---- collect
    const int lane = tid & 63, wave = tid >> 6;
    unsigned base = 0u;                           // candidates == T seen in the chunks before this one (same in every thread)
    for (int t0 = 0; t0 < n; t0 += 256) {
        const int t = t0 + tid;
        const unsigned key = (t < n) ? key_of(t) : 0xFFFFFFFFu;
        const bool lt = (t < n) && key < T, eq = (t < n) && key == T;
        const unsigned long long mk = __ballot(eq);
        if (lane == 0) wcnt[wave] = (unsigned)__popcll(mk);
        __syncthreads();
        unsigned before = base;
        for (int w = 0; w < wave; ++w) before += wcnt[w];
        base += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        before += __builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u));
        if (lt || (eq && before < quota)) {
            const unsigned pos = atomicAdd(&sel[2], 1u);
            keys[pos] = ((unsigned long long)key << 32) | (unsigned)t;
        }
        __syncthreads();                          // wcnt is rewritten by the next chunk
    }
    __syncthreads();
    for (int t = k + tid; t < kpad; t += 256) keys[t] = ~0ull;
    __syncthreads();

    // ---- sort the k survivors by (distance bits, index)
    for (int size = 2; size <= kpad; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < (kpad >> 1); t += 256) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool up = ((lo & size) == 0);
                const unsigned long long a = keys[lo], b2 = keys[hi];
                if ((a > b2) == up) { keys[lo] = b2; keys[hi] = a; }
            }
            __syncthreads();
        }
    }
    const size_t o = ((size_t)cloud * m + qi) * k;
    for (int t = tid; t < k; t += 256) {
        const unsigned long long kv = keys[t];
        idx[o + t] = (int)(unsigned)kv;
        if (dist) {
            const float d = ordered_to_f32((unsigned)(kv >> 32));
            dist[o + t] = neg ? -d : d;
        }
    }
}

// mode: KG_*; returns hipErrorInvalidValue outside k <= 4096, c <= 4096
int knn_general_launch(int mode, int b, int n, int m, int c, int k, long ldp, long ldq, const float* points, const float* queries,
                       float* dist, int* idx, int neg, hipStream_t st) {
    if (k > KG_MAXK || c > KG_MAXC || k > n) return (int)hipErrorInvalidValue;
    int kpad = 2;
    while (kpad < k) kpad <<= 1;
    const size_t bytes = (size_t)kpad * 8 + (size_t)((c + 3) & ~3) * 4;
    const int vec4 = (c % 4 == 0) && (ldp % 4 == 0) && (((uintptr_t)points) % 16 == 0);
    const dim3 grid(m, b);
#define KG_LAUNCH(M)                                                                                                          \
    do {                                                                                                                      \
        static DevOnce attr;                                                                                                   \
        if (attr.needed()) {                                                                                                          \
            DISPU_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(knn_general_kernel<M>),                              \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, KG_MAXK * 8 + KG_MAXC * 4));           \
            attr.done();                                                                                                      \
        }                                                                                                                     \
        hipLaunchKernelGGL((knn_general_kernel<M>), grid, dim3(256), bytes, st, n, m, c, k, kpad, ldp, ldq, points, queries, \
                           dist, idx, neg, vec4);                                                                             \
    } while (0)
    if (mode == KG_XYZ_PLAIN) KG_LAUNCH(KG_XYZ_PLAIN);
    else if (mode == KG_XYZ_FMA) KG_LAUNCH(KG_XYZ_FMA);
    else if (mode == KG_SQ) KG_LAUNCH(KG_SQ);
    else KG_LAUNCH(KG_GEMM);
#undef KG_LAUNCH
    return (int)hipGetLastError();
}

}  // namespace dispu
