// Exact k-nearest-neighbour kernels for gfx950.
//   dispu_knn_xyz   - replaces the HOST round trip of the reference forward path:
//                     tf.py_func(knn_query) -> nearest_neighbors.knn_batch -> nanoflann KD-tree
//                     (Common/ops.py:110-118,165; libs/nearest_neighbors/knn_.cxx:104-135).
//   dispu_knn_feat  - tf_grouping.knn_point_2 (tf_ops/grouping/tf_grouping.py:95-114): GEMM-form
//                     distances D = r_q - 2 q.p + r_p followed by top_k(-D).
//   dispu_knn_point - tf_grouping.knn_point (:116-141): sum((p-q)^2) followed by top_k(-dist).
//
// Design: a lane owns one query and keeps its k best (distance, index) pairs SORTED in VGPRs.
// Candidates stream through LDS (one broadcast ds_read_b128 per xyz candidate) in ascending
// index order; a candidate enters the list only when strictly smaller than the current k-th
// best, which makes ties resolve to the lower index - tf.nn.top_k's rule and nanoflann's
// KNNResultSet order for candidates presented in index order.  The insertion is branch-free
// inside the lane (v_med3_f32 per slot) and skipped for the whole wave when no lane accepts.
#include "common.h"

namespace dispu {

constexpr int KNN_BS = 256;
constexpr int KNN_TILE = 1024;

template <int K>
struct TopK {
    float d[K];
    int i[K];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int t = 0; t < K; ++t) { d[t] = __builtin_inff(); i[t] = 0; }
    }
    __device__ __forceinline__ float worst() const { return d[K - 1]; }
    // stable sorted insert; a no-op for lanes with x >= worst()
    __device__ __forceinline__ void insert(float x, int xi) {
        bool c_hi = x < d[K - 1];
#pragma unroll
        for (int t = K - 1; t >= 1; --t) {
            const bool c_lo = x < d[t - 1];
            i[t] = c_lo ? i[t - 1] : (c_hi ? xi : i[t]);
            d[t] = __builtin_amdgcn_fmed3f(d[t - 1], d[t], x);
            c_hi = c_lo;
        }
        i[0] = c_hi ? xi : i[0];
        d[0] = fminf(d[0], x);
    }
};

template <int K, bool FMA>
__global__ __launch_bounds__(KNN_BS) void knn_xyz_kernel(int n, int m, int k, const float* __restrict__ support,
                                                          const float* __restrict__ query, int* __restrict__ idx,
                                                          float* __restrict__ dist) {
    __shared__ float4 tile[KNN_TILE];
    const int cloud = blockIdx.y;
    const float* __restrict__ s = support + (size_t)cloud * n * 3;
    const float* __restrict__ q = query + (size_t)cloud * m * 3;
    const int j = blockIdx.x * KNN_BS + threadIdx.x;
    const bool active = j < m;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (active) { qx = q[j * 3 + 0]; qy = q[j * 3 + 1]; qz = q[j * 3 + 2]; }
    TopK<K> best;
    best.init();
    for (int k0 = 0; k0 < n; k0 += KNN_TILE) {
        const int len = min(KNN_TILE, n - k0);
        __syncthreads();
        for (int t = threadIdx.x; t < len; t += KNN_BS)
            tile[t] = make_float4(s[(k0 + t) * 3 + 0], s[(k0 + t) * 3 + 1], s[(k0 + t) * 3 + 2], 0.f);
        __syncthreads();
        for (int t = 0; t < len; ++t) {
            const float4 p = tile[t];
            const float d = sqdist3<FMA>(qx - p.x, qy - p.y, qz - p.z);
            if (__any(d < best.worst())) best.insert(d, k0 + t);
        }
    }
    if (active) {
        const size_t o = ((size_t)cloud * m + j) * k;
#pragma unroll
        for (int t = 0; t < K; ++t)
            if (t < k) {
                idx[o + t] = best.i[t];
                if (dist) dist[o + t] = best.d[t];
            }
    }
}

// Feature-space kNN.  CP = channel count padded to a multiple of 4 (zero padding is exact:
// fma(0,0,acc) == acc).  GEMM_FORM: D = (rq - 2*dot) + rp with fma chains over channels in
// ascending order (bit-identical to a v_mfma_f32 k-loop); otherwise sum((p-q)^2) left to right
// with every op rounded.  NEG: store -d (knn_point returns top_k values of -dist).
template <int CP, int K, bool GEMM_FORM, bool NEG>
__global__ __launch_bounds__(KNN_BS) void knn_feat_kernel(int n, int m, int c, int k, int ldp, int ldq,
                                                           const float* __restrict__ points,
                                                           const float* __restrict__ queries,
                                                           float* __restrict__ dist, int* __restrict__ idx) {
    constexpr int TILE = (CP <= 16) ? 512 : ((CP <= 48) ? 256 : 64);
    __shared__ float tile[TILE * CP];
    __shared__ float tnorm[TILE];
    const int cloud = blockIdx.y;
    const float* __restrict__ sp = points + (size_t)cloud * n * ldp;   // row strides ldp / ldq (floats):
    const float* __restrict__ qp = queries + (size_t)cloud * m * ldq;  // inputs may be column slices
    const int j = blockIdx.x * KNN_BS + threadIdx.x;
    const bool active = j < m;
    float q[CP];
#pragma unroll
    for (int l = 0; l < CP; ++l) q[l] = (active && l < c) ? qp[(size_t)j * ldq + l] : 0.f;
    float rq = 0.f;
    if constexpr (GEMM_FORM) {
#pragma unroll
        for (int l = 0; l < CP; ++l) rq = __builtin_fmaf(q[l], q[l], rq);
    }
    TopK<K> best;
    best.init();
    for (int k0 = 0; k0 < n; k0 += TILE) {
        const int len = min(TILE, n - k0);
        __syncthreads();
        for (int e = threadIdx.x; e < len * CP; e += KNN_BS) {
            const int t = e / CP, l = e - t * CP;
            tile[e] = (l < c) ? sp[(size_t)(k0 + t) * ldp + l] : 0.f;
        }
        __syncthreads();
        if constexpr (GEMM_FORM) {
            for (int t = threadIdx.x; t < len; t += KNN_BS) {
                float r = 0.f;
                for (int l = 0; l < CP; ++l) r = __builtin_fmaf(tile[t * CP + l], tile[t * CP + l], r);
                tnorm[t] = r;
            }
            __syncthreads();
        }
        // U candidates per iteration: their fmaf chains are independent, which hides the FMA latency of the
        // (bit-pinned, strictly sequential) per-candidate chain; insertion stays in index order.
        constexpr int U = 4;
        for (int t0 = 0; t0 < len; t0 += U) {
          float du[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int t = (t0 + u < len) ? t0 + u : t0;
            const float4* __restrict__ row = reinterpret_cast<const float4*>(tile + t * CP);
            float d;
            if constexpr (GEMM_FORM) {
                float dot = 0.f;
#pragma unroll
                for (int l4 = 0; l4 < CP / 4; ++l4) {
                    const float4 v = row[l4];
                    dot = __builtin_fmaf(q[l4 * 4 + 0], v.x, dot);
                    dot = __builtin_fmaf(q[l4 * 4 + 1], v.y, dot);
                    dot = __builtin_fmaf(q[l4 * 4 + 2], v.z, dot);
                    dot = __builtin_fmaf(q[l4 * 4 + 3], v.w, dot);
                }
                const float rq_m2dot = rq - 2.0f * dot;
                d = rq_m2dot + tnorm[t];
            } else {
                d = 0.f;
#pragma unroll
                for (int l4 = 0; l4 < CP / 4; ++l4) {
                    const float4 v = row[l4];
                    float df;
                    df = v.x - q[l4 * 4 + 0]; d = d + df * df;
                    df = v.y - q[l4 * 4 + 1]; d = d + df * df;
                    df = v.z - q[l4 * 4 + 2]; d = d + df * df;
                    df = v.w - q[l4 * 4 + 3]; d = d + df * df;
                }
            }
            du[u] = (t0 + u < len) ? d : __builtin_inff();
          }
#pragma unroll
          for (int u = 0; u < U; ++u)
            if (__any(du[u] < best.worst())) best.insert(du[u], k0 + t0 + u);
        }
    }
    if (active) {
        const size_t o = ((size_t)cloud * m + j) * k;
#pragma unroll
        for (int t = 0; t < K; ++t)
            if (t < k) {
                idx[o + t] = best.i[t];
                if (dist) dist[o + t] = NEG ? -best.d[t] : best.d[t];
            }
    }
}

template <int K>
static int launch_xyz(int b, int n, int m, int k, const float* s, const float* q, int* idx, float* dist, int arith,
                      hipStream_t st) {
    dim3 grid((m + KNN_BS - 1) / KNN_BS, b);
    if ((arith & DISPU_ARITH_CONTRACT))
        hipLaunchKernelGGL((knn_xyz_kernel<K, true>), grid, dim3(KNN_BS), 0, st, n, m, k, s, q, idx, dist);
    else
        hipLaunchKernelGGL((knn_xyz_kernel<K, false>), grid, dim3(KNN_BS), 0, st, n, m, k, s, q, idx, dist);
    return (int)hipGetLastError();
}

template <int CP, bool GEMM_FORM, bool NEG>
static int launch_feat_k(int b, int n, int m, int c, int k, int ldp, int ldq, const float* p, const float* q, float* dist, int* idx,
                         hipStream_t st) {
    dim3 grid((m + KNN_BS - 1) / KNN_BS, b);
    if (k <= 8)
        hipLaunchKernelGGL((knn_feat_kernel<CP, 8, GEMM_FORM, NEG>), grid, dim3(KNN_BS), 0, st, n, m, c, k, ldp, ldq, p, q, dist, idx);
    else if (k <= 20)
        hipLaunchKernelGGL((knn_feat_kernel<CP, 20, GEMM_FORM, NEG>), grid, dim3(KNN_BS), 0, st, n, m, c, k, ldp, ldq, p, q, dist, idx);
    else
        hipLaunchKernelGGL((knn_feat_kernel<CP, 32, GEMM_FORM, NEG>), grid, dim3(KNN_BS), 0, st, n, m, c, k, ldp, ldq, p, q, dist, idx);
    return (int)hipGetLastError();
}

template <bool GEMM_FORM, bool NEG>
static int launch_feat(int b, int n, int m, int c, int k, int ldp, int ldq, const float* p, const float* q, float* dist, int* idx,
                       hipStream_t st) {
    if (c <= 4) return launch_feat_k<4, GEMM_FORM, NEG>(b, n, m, c, k, ldp, ldq, p, q, dist, idx, st);
    if (c <= 8) return launch_feat_k<8, GEMM_FORM, NEG>(b, n, m, c, k, ldp, ldq, p, q, dist, idx, st);
    if (c <= 16) return launch_feat_k<16, GEMM_FORM, NEG>(b, n, m, c, k, ldp, ldq, p, q, dist, idx, st);
    if (c <= 24) return launch_feat_k<24, GEMM_FORM, NEG>(b, n, m, c, k, ldp, ldq, p, q, dist, idx, st);
    if (c <= 32) return launch_feat_k<32, GEMM_FORM, NEG>(b, n, m, c, k, ldp, ldq, p, q, dist, idx, st);
    if (c <= 48) return launch_feat_k<48, GEMM_FORM, NEG>(b, n, m, c, k, ldp, ldq, p, q, dist, idx, st);
    if (c <= 64) return launch_feat_k<64, GEMM_FORM, NEG>(b, n, m, c, k, ldp, ldq, p, q, dist, idx, st);
    if (c <= 128) return launch_feat_k<128, GEMM_FORM, NEG>(b, n, m, c, k, ldp, ldq, p, q, dist, idx, st);
    return (int)hipErrorInvalidValue;
}

// wave-per-query fast paths (knn_wave.hip); return -1 when the shape is outside their range
int knn_xyz_wave_dispatch(int b, int n, int m, int k, const float* s, const float* q, int* idx, float* dist, int arith,
                          hipStream_t st);
int knn_feat_wave_dispatch(int b, int n, int m, int c, int k, int ldp, int ldq, const float* p, const float* q, float* dist,
                           int* idx, hipStream_t st, long pstride = 0);
size_t knn_feat_chunked_scratch(int b, int n, int m, int c, int k);
int knn_feat_chunked_dispatch(int b, int n, int m, int c, int k, int ldp, int ldq, const float* p, const float* q, float* dist, int* idx,
                              void* scratch, size_t scratch_bytes, hipStream_t st);

// 1024 < n <= 8192, k <= 32 (knn_wave.hip): per-chunk wave kernel + merge; needs caller scratch
size_t knn_xyz_chunked_scratch(int b, int n, int m, int k);
int knn_xyz_lds_dispatch(int b, int n, int m, int k, const float* s, const float* q, int* idx, float* dist, int arith, hipStream_t st);
int knn_xyz_chunked_dispatch(int b, int n, int m, int k, const float* s, const float* q, int* idx, float* dist, void* scratch,
                             size_t scratch_bytes, int arith, hipStream_t st);
// general path (knn_general.hip): any k <= 4096, c <= 4096, any n; mode 0/1 xyz plain/contract, 2 knn_point, 3 knn_point_2
int knn_general_launch(int mode, int b, int n, int m, int c, int k, long ldp, long ldq, const float* points, const float* queries,
                       float* dist, int* idx, int neg, hipStream_t st);

}  // namespace dispu

using namespace dispu;

DISPU_EXPORT int dispu_knn_xyz(int b, int n, int m, int k, const float* support, const float* query, int* idx,
                               float* dist, int arith, void* stream) {
    if (b < 0 || n <= 0 || m < 0 || k <= 0 || k > n) return (int)hipErrorInvalidValue;
    if (b == 0 || m == 0) return 0;
    if (!idx) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    if (k > 32)                                      // nanoflann takes any K (knn_.cxx:104-135): general selection kernel
        return knn_general_launch((arith & DISPU_ARITH_CONTRACT) ? 1 : 0, b, n, m, 3, k, 3, 3, support, query, dist, idx, 0, st);
    if (!(arith & DISPU_KNN_LANE_PER_QUERY)) {       // fast path: wave-per-query (knn_wave.hip), n <= 1024
        const int rc = knn_xyz_wave_dispatch(b, n, m, k, support, query, idx, dist, arith, st);
        if (rc >= 0) return rc;
        const int rl = knn_xyz_lds_dispatch(b, n, m, k, support, query, idx, dist, arith, st);     // 1024 < n <= 4096, no scratch needed
        if (rl >= 0) return rl;
    }
    if (k <= 4) return launch_xyz<4>(b, n, m, k, support, query, idx, dist, arith, st);
    if (k <= 8) return launch_xyz<8>(b, n, m, k, support, query, idx, dist, arith, st);
    if (k <= 16) return launch_xyz<16>(b, n, m, k, support, query, idx, dist, arith, st);
    if (k <= 20) return launch_xyz<20>(b, n, m, k, support, query, idx, dist, arith, st);
    return launch_xyz<32>(b, n, m, k, support, query, idx, dist, arith, st);
}

// dispu_knn_xyz with caller scratch (dispu_knn_xyz_scratch_bytes, 0 for shapes that need none): clouds of 1025 .. 8192 points
// then take the chunked wave path instead of the lane-per-query kernel.  Same results.
DISPU_EXPORT size_t dispu_knn_xyz_scratch_bytes(int b, int n, int m, int k) {
    if (b <= 0 || n <= 0 || m <= 0 || k <= 0) return 0;
    return knn_xyz_chunked_scratch(b, n, m, k);
}

DISPU_EXPORT int dispu_knn_xyz_ws(int b, int n, int m, int k, const float* support, const float* query, int* idx, float* dist,
                                  void* scratch, size_t scratch_bytes, int arith, void* stream) {
    if (b < 0 || n <= 0 || m < 0 || k <= 0 || k > n) return (int)hipErrorInvalidValue;
    if (b == 0 || m == 0) return 0;
    if (!idx) return (int)hipErrorInvalidValue;
    if (!(arith & DISPU_KNN_LANE_PER_QUERY)) {
        const int rc = knn_xyz_chunked_dispatch(b, n, m, k, support, query, idx, dist, scratch, scratch_bytes, arith, (hipStream_t)stream);
        if (rc >= 0) return rc;
    }
    return dispu_knn_xyz(b, n, m, k, support, query, idx, dist, arith, stream);
}

DISPU_EXPORT int dispu_knn_feat(int b, int n, int m, int c, int k, const float* points, const float* queries,
                                float* dist, int* idx, void* stream) {
    if (b < 0 || n <= 0 || m < 0 || c <= 0 || k <= 0 || k > n) return (int)hipErrorInvalidValue;
    if (b == 0 || m == 0) return 0;
    if (!idx) return (int)hipErrorInvalidValue;
    if (k > 32 || c > 128) return knn_general_launch(3, b, n, m, c, k, c, c, points, queries, dist, idx, 0, (hipStream_t)stream);
    const int rc = knn_feat_wave_dispatch(b, n, m, c, k, c, c, points, queries, dist, idx, (hipStream_t)stream);
    if (rc >= 0) return rc;
    return launch_feat<true, false>(b, n, m, c, k, c, c, points, queries, dist, idx, (hipStream_t)stream);
}

// Same as dispu_knn_feat with explicit row strides (floats): points / queries may be column slices of a wider
// activation buffer (the generator keeps its dense-block features inside one [rows, 480] buffer).
DISPU_EXPORT int dispu_knn_feat_strided(int b, int n, int m, int c, int k, const float* points, int ldp,
                                        const float* queries, int ldq, float* dist, int* idx, void* stream) {
    if (b < 0 || n <= 0 || m < 0 || c <= 0 || k <= 0 || k > n || ldp < c || ldq < c) return (int)hipErrorInvalidValue;
    if (b == 0 || m == 0) return 0;
    if (!idx) return (int)hipErrorInvalidValue;
    if (k > 32 || c > 128) return knn_general_launch(3, b, n, m, c, k, ldp, ldq, points, queries, dist, idx, 0, (hipStream_t)stream);
    const int rc = knn_feat_wave_dispatch(b, n, m, c, k, ldp, ldq, points, queries, dist, idx, (hipStream_t)stream);
    if (rc >= 0) return rc;
    return launch_feat<true, false>(b, n, m, c, k, ldp, ldq, points, queries, dist, idx, (hipStream_t)stream);
}

// dispu_knn_feat_strided with caller scratch (dispu_knn_feat_scratch_bytes; 0 = none needed): clouds of 513 .. 4096 points take
// the chunked wave path.  Same results.
DISPU_EXPORT size_t dispu_knn_feat_scratch_bytes(int b, int n, int m, int c, int k) {
    if (b <= 0 || n <= 0 || m <= 0 || c <= 0 || k <= 0) return 0;
    return knn_feat_chunked_scratch(b, n, m, c, k);
}

DISPU_EXPORT int dispu_knn_feat_strided_ws(int b, int n, int m, int c, int k, const float* points, int ldp, const float* queries, int ldq,
                                           float* dist, int* idx, void* scratch, size_t scratch_bytes, void* stream) {
    if (b < 0 || n <= 0 || m < 0 || c <= 0 || k <= 0 || k > n || ldp < c || ldq < c) return (int)hipErrorInvalidValue;
    if (b == 0 || m == 0) return 0;
    if (!idx) return (int)hipErrorInvalidValue;
    const int rc = knn_feat_chunked_dispatch(b, n, m, c, k, ldp, ldq, points, queries, dist, idx, scratch, scratch_bytes, (hipStream_t)stream);
    if (rc >= 0) return rc;
    return dispu_knn_feat_strided(b, n, m, c, k, points, ldp, queries, ldq, dist, idx, stream);
}

DISPU_EXPORT int dispu_knn_point(int b, int n, int m, int c, int k, const float* xyz1, const float* xyz2, float* val,
                                 int* idx, void* stream) {
    if (b < 0 || n <= 0 || m < 0 || c <= 0 || k <= 0 || k > n) return (int)hipErrorInvalidValue;
    if (b == 0 || m == 0) return 0;
    if (!idx) return (int)hipErrorInvalidValue;
    if (k > 32 || c > 128) return knn_general_launch(2, b, n, m, c, k, c, c, xyz1, xyz2, val, idx, 1, (hipStream_t)stream);
    return launch_feat<false, true>(b, n, m, c, k, c, c, xyz1, xyz2, val, idx, (hipStream_t)stream);
}
