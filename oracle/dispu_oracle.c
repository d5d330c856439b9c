/*
 * dispu_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the Dis-PU point-sampling / grouping / distance hot path
 * (SURVEY.md section 8a rows A1..A12).  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may call into this file; the product path
 * (dis-pu_amd/ -> libdispu_hip.so) never does and fails loudly without the HIP library.
 *
 * Every function cites the reference file:line it restates (paths relative to the
 * upstream tree liruihui/Dis-PU).  Nothing here is copied from the reference: loops are
 * re-derived from the documented behaviour, and the thread-layout-dependent tie rules of
 * the reference GPU kernels are reproduced arithmetically instead of by simulating threads.
 *
 * Pinned arithmetic.  The reference has two arithmetic flavours for the 3-term squared
 * distance d2 = dx*dx + dy*dy + dz*dz:
 *   contract = 0 : ((dx*dx + dy*dy) + dz*dz), every op rounded.  This is what g++ -O2 emits
 *                  on x86-64 for the reference's CPU functions (nnsearch, threenn_cpu,
 *                  query_ball_point_cpu, nanoflann L2_Adaptor) and is checked BIT-EXACTLY
 *                  against those functions compiled from /root/reference (oracle/_ref).
 *   contract = 1 : fmaf(dz,dz, fmaf(dx,dx, dy*dy)).  nvcc -O2 (--fmad=true) contracts the
 *                  same source expression; LLVM's (fadd (fmul a a) (fmul b b)) combine fuses the
 *                  FIRST multiply and keeps the second, which gives this form.  The SASS of
 *                  the reference's CUDA objects cannot be inspected here (no nvcc), so the
 *                  contracted form is "parity unpinned" at the 1-ulp level; results differ
 *                  from contract=0 only on near-ties.
 * Build with -ffp-contract=off so the compiler never fuses on its own; with -mfma the
 * explicit fmaf() calls become single vfmadd instructions.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

static inline float sqdist3(float dx, float dy, float dz, int contract) {
    if (contract) return fmaf(dz, dz, fmaf(dx, dx, dy * dy));
    float s = dx * dx + dy * dy;
    return s + dz * dz;
}

/* Pinned exponential: exp(x) for x <= 0 as 2^(x*log2e) with round-to-nearest-even range reduction
 * and a degree-6 Taylor polynomial of 2^f evaluated as an fmaf chain (|f| <= 0.5, remainder 1.2e-7).
 * The HIP library has the SAME sequence (v_rndne_f32 + v_fma_f32), so in DISPU_ARITH_PINNED_EXP mode
 * approx_match is bit-reproducible between CPU and GPU.  The production default uses the hardware
 * v_exp_f32 like the reference's __expf (tf_approxmatch_g.cu:52,97,151). */
static inline float pinned_exp(float x) {
    if (!(x > -86.0f)) return 0.0f;
    const float t = x * 1.44269504088896341f;
    const float n = rintf(t);
    const float f = t - n;
    float p = 1.5403530393381609954e-4f;
    p = fmaf(p, f, 1.3333558146428443423e-3f);
    p = fmaf(p, f, 9.6181291076284771619e-3f);
    p = fmaf(p, f, 5.5504108664821579953e-2f);
    p = fmaf(p, f, 2.4022650695910071233e-1f);
    p = fmaf(p, f, 6.9314718055994530942e-1f);
    p = fmaf(p, f, 1.0f);
    union { float f; int32_t i; } u;
    u.f = p;
    u.i += ((int32_t)n) << 23;
    return u.f;
}
#define ORC_EXP(x) (pinned ? pinned_exp(x) : expf(x))

ORC_API int orc_version(void) { return 1; }

/* Thread policy (cpu_baseline only; results never depend on it): the loops below run ONE CLOUD PER THREAD, the way the
 * reference parallelises its only multi-threaded CPU op (libs/nearest_neighbors/knn_.cxx:108, "#pragma omp parallel for"
 * over the batch).  A region never gets more threads than it has clouds: waking 256 hardware threads for a 32-cloud
 * loop costs more than the loop (round-2 measurement: "all cores" slower than one thread). */
int orc_g_threads = 0;          /* 0 = OpenMP default; shared with mlp_oracle.c */
int orc_nt(long items) {
#ifdef _OPENMP
    int t = orc_g_threads > 0 ? orc_g_threads : omp_get_max_threads();
#else
    int t = 1;
#endif
    if (items < t) t = items > 0 ? (int)items : 1;
    return t;
}

ORC_API void orc_set_threads(int t) {
    orc_g_threads = t > 0 ? t : 0;
#ifdef _OPENMP
    if (t > 0) omp_set_num_threads(t);
#endif
}

/* ------------------------------------------------------------------------------------------
 * A1  farthest point sampling.
 * Restates farthestpointsamplingKernel, tf_ops/sampling/tf_sampling_g.cu:105-170:
 *   idx[0] = 0 (:114-116); temp[k] = 1e38 (:118); round j: d = |p_k - p_old|^2 in the
 *   un-centred (x2-x1) form (:142), d2 = min(d, temp[k]) (:143-145), candidate if d2 > best
 *   with best initialised to -1 and besti to 0 (:126-127,146-149).
 * Tie rule: thread t scans k = t, t+BS, ... keeping the first maximum (strict >), and the
 * shared-memory tree (:153-163) keeps the lower slot on equal values, so the winner is the
 * maximum d2 with the lowest (k mod BS), then lowest k.  BS = 512 (:204).
 * ------------------------------------------------------------------------------------------ */
ORC_API void orc_fps(int b, int n, int m, const float *xyz, int *idx, int contract, int bs) {
    if (m <= 0 || n <= 0) return;
    if (bs <= 0) bs = 512;
#pragma omp parallel for schedule(dynamic) num_threads(orc_nt(b))
    for (int i = 0; i < b; ++i) {
        const float *p = xyz + (size_t)i * n * 3;
        float *temp = (float *)malloc(sizeof(float) * (size_t)n);
        float *sbest = (float *)malloc(sizeof(float) * (size_t)bs);
        int *sbesti = (int *)malloc(sizeof(int) * (size_t)bs);
        for (int k = 0; k < n; ++k) temp[k] = 1e38f;
        int old = 0;
        idx[(size_t)i * m] = 0;
        for (int j = 1; j < m; ++j) {
            for (int t = 0; t < bs; ++t) { sbest[t] = -1.0f; sbesti[t] = 0; }
            const float x1 = p[old * 3 + 0], y1 = p[old * 3 + 1], z1 = p[old * 3 + 2];
            for (int k = 0; k < n; ++k) {
                const float d = sqdist3(p[k * 3 + 0] - x1, p[k * 3 + 1] - y1, p[k * 3 + 2] - z1, contract);
                const float td = temp[k];
                const float d2 = fminf(d, td);
                if (d2 != td) temp[k] = d2;
                const int t = k % bs;
                if (d2 > sbest[t]) { sbest[t] = d2; sbesti[t] = k; }
            }
            float best = sbest[0];
            int besti = sbesti[0];
            for (int t = 1; t < bs; ++t)
                if (best < sbest[t]) { best = sbest[t]; besti = sbesti[t]; }
            /* The pairwise tree keeps the LEFT operand on ties; a left-to-right scan with a
             * strict '<' picks the same element (lowest slot among equal maxima). */
            old = besti;
            idx[(size_t)i * m + j] = old;
        }
        free(temp); free(sbest); free(sbesti);
    }
}

/* A2  gather_point / grad.  gatherpointKernel tf_sampling_g.cu:172-181 (c = 3 only),
 * scatteraddpointKernel :183-192 (atomic float adds into a zero-filled buffer,
 * tf_sampling.cpp:174).  The oracle adds in ascending j (one admissible atomic order). */
ORC_API void orc_gather_point(int b, int n, int m, const float *inp, const int *idx, float *out) {
    for (int i = 0; i < b; ++i)
        for (int j = 0; j < m; ++j) {
            const int a = idx[(size_t)i * m + j];
            for (int l = 0; l < 3; ++l) out[((size_t)i * m + j) * 3 + l] = inp[((size_t)i * n + a) * 3 + l];
        }
}
ORC_API void orc_gather_point_grad(int b, int n, int m, const float *out_g, const int *idx, float *inp_g) {
    memset(inp_g, 0, sizeof(float) * (size_t)b * n * 3);
    for (int i = 0; i < b; ++i)
        for (int j = 0; j < m; ++j) {
            const int a = idx[(size_t)i * m + j];
            for (int l = 0; l < 3; ++l) inp_g[((size_t)i * n + a) * 3 + l] += out_g[((size_t)i * m + j) * 3 + l];
        }
}

/* A3  query_ball_point.  query_ball_point_gpu tf_ops/grouping/tf_grouping_g.cu:3-36 (same body
 * as query_ball_point_cpu, tf_ops/grouping/query_ball_point.cpp:19-47): for query j scan the
 * dataset in index order, d = max(sqrtf(d2), 1e-20f), hit if d < radius[0] (strict; only
 * element 0 of radius is read, :25); first hit fills all nsample slots, later hits overwrite
 * slot cnt; stop at nsample hits.  A query with no hit leaves its idx row UNTOUCHED
 * (caller's buffer content survives) and pts_cnt = 0.  pts_cnt may be NULL (the CPU twin in
 * the reference has no such output). */
ORC_API void orc_query_ball(int b, int n, int m, const float *radius, int nsample, const float *xyz1,
                            const float *xyz2, int *idx, int *pts_cnt, int contract) {
    const float r = radius[0];
#pragma omp parallel for schedule(dynamic) num_threads(orc_nt(b))
    for (int i = 0; i < b; ++i) {
        const float *p1 = xyz1 + (size_t)i * n * 3;
        const float *p2 = xyz2 + (size_t)i * m * 3;
        int *id = idx + (size_t)i * m * nsample;
        for (int j = 0; j < m; ++j) {
            int cnt = 0;
            const float x2 = p2[j * 3 + 0], y2 = p2[j * 3 + 1], z2 = p2[j * 3 + 2];
            for (int k = 0; k < n && cnt < nsample; ++k) {
                const float d2 = sqdist3(x2 - p1[k * 3 + 0], y2 - p1[k * 3 + 1], z2 - p1[k * 3 + 2], contract);
                const float d = fmaxf(sqrtf(d2), 1e-20f);
                if (d < r) {
                    if (cnt == 0)
                        for (int l = 0; l < nsample; ++l) id[j * nsample + l] = k;
                    id[j * nsample + cnt] = k;
                    ++cnt;
                }
            }
            if (pts_cnt) pts_cnt[(size_t)i * m + j] = cnt;
        }
    }
}

/* A4  group_point / grad.  group_point_gpu tf_grouping_g.cu:40-57, group_point_grad_gpu :61-78
 * (atomic adds into a zero-filled buffer, tf_grouping.cpp:208); CPU twins
 * query_ball_point.cpp:52-84. */
ORC_API void orc_group_point(int b, int n, int c, int m, int ns, const float *points, const int *idx, float *out) {
#pragma omp parallel for schedule(dynamic) num_threads(orc_nt(b))
    for (int i = 0; i < b; ++i)
        for (int j = 0; j < m; ++j)
            for (int k = 0; k < ns; ++k) {
                const int ii = idx[((size_t)i * m + j) * ns + k];
                memcpy(out + (((size_t)i * m + j) * ns + k) * c, points + ((size_t)i * n + ii) * c, sizeof(float) * c);
            }
}
ORC_API void orc_group_point_grad(int b, int n, int c, int m, int ns, const float *grad_out, const int *idx,
                                  float *grad_points) {
    memset(grad_points, 0, sizeof(float) * (size_t)b * n * c);
    for (int i = 0; i < b; ++i)
        for (int j = 0; j < m; ++j)
            for (int k = 0; k < ns; ++k) {
                const int ii = idx[((size_t)i * m + j) * ns + k];
                for (int l = 0; l < c; ++l)
                    grad_points[((size_t)i * n + ii) * c + l] += grad_out[(((size_t)i * m + j) * ns + k) * c + l];
            }
}

/* Stable insertion of (d,id) into an ascending list of length k: an element moves in front of
 * stored elements only when STRICTLY smaller, so among equal distances the earlier (lower)
 * index stays first.  This is tf.nn.top_k's documented tie rule (lower index first) and
 * nanoflann's KNNResultSet::addPoint (nanoflann.hpp:115-139) when candidates arrive in index
 * order. */
static inline void topk_insert(float *bd, int *bi, int k, float d, int id) {
    if (!(d < bd[k - 1])) return;
    int p = k - 1;
    while (p > 0 && d < bd[p - 1]) { bd[p] = bd[p - 1]; bi[p] = bi[p - 1]; --p; }
    bd[p] = d; bi[p] = id;
}

/* A7  batched exact kNN on xyz: nearest_neighbors.knn_batch -> cpp_knn_batch_omp
 * (libs/nearest_neighbors/knn_.cxx:104-135) -> nanoflann KD-tree, L2_Adaptor::evalMetric
 * (nanoflann.hpp:323-348: ((0+dx*dx)+dy*dy)+dz*dz, diff = query - point, i.e. contract=0).
 * Result: k nearest in ascending squared distance.  The KD-tree visit order decides exact
 * ties in the reference (no NANOFLANN_FIRST_MATCH); the oracle defines lowest index first.
 * Requires k <= n.  dist may be NULL. */
ORC_API void orc_knn_xyz(int b, int n, int m, int k, const float *support, const float *query, int *idx,
                         float *dist, int contract) {
#pragma omp parallel for schedule(dynamic) num_threads(orc_nt(b))
    for (int i = 0; i < b; ++i) {
        const float *s = support + (size_t)i * n * 3;
        const float *q = query + (size_t)i * m * 3;
        float *bd = (float *)malloc(sizeof(float) * k);
        int *bi = (int *)malloc(sizeof(int) * k);
        for (int j = 0; j < m; ++j) {
            for (int t = 0; t < k; ++t) { bd[t] = INFINITY; bi[t] = 0; }
            const float qx = q[j * 3 + 0], qy = q[j * 3 + 1], qz = q[j * 3 + 2];
            for (int p = 0; p < n; ++p) {
                const float d = sqdist3(qx - s[p * 3 + 0], qy - s[p * 3 + 1], qz - s[p * 3 + 2], contract);
                topk_insert(bd, bi, k, d, p);
            }
            for (int t = 0; t < k; ++t) {
                idx[((size_t)i * m + j) * k + t] = bi[t];
                if (dist) dist[((size_t)i * m + j) * k + t] = bd[t];
            }
        }
        free(bd); free(bi);
    }
}

/* A5  knn_point (tf_ops/grouping/tf_grouping.py:116-141): dist[b,j,p] = sum_c (xyz1[p,c]-xyz2[j,c])^2
 * (sequential over c, every op rounded), val,idx = top_k(-dist,k): val = NEGATIVE squared
 * distance, ascending distance, ties -> lower index.  xyz1 = dataset [b,n,c], xyz2 = queries
 * [b,m,c].  TensorFlow's reduce_sum order is not pinned by anything in the reference tree
 * ("parity unpinned" at that boundary); left-to-right is the restated order. */
ORC_API void orc_knn_point(int b, int n, int m, int c, int k, const float *xyz1, const float *xyz2, float *val,
                           int *idx) {
#pragma omp parallel for schedule(dynamic) num_threads(orc_nt(b))
    for (int i = 0; i < b; ++i) {
        float *bd = (float *)malloc(sizeof(float) * k);
        int *bi = (int *)malloc(sizeof(int) * k);
        for (int j = 0; j < m; ++j) {
            const float *q = xyz2 + ((size_t)i * m + j) * c;
            for (int t = 0; t < k; ++t) { bd[t] = INFINITY; bi[t] = 0; }
            for (int p = 0; p < n; ++p) {
                const float *s = xyz1 + ((size_t)i * n + p) * c;
                float d = 0.0f;
                for (int l = 0; l < c; ++l) { const float df = s[l] - q[l]; d = d + df * df; }
                topk_insert(bd, bi, k, d, p);
            }
            for (int t = 0; t < k; ++t) {
                val[((size_t)i * m + j) * k + t] = -bd[t];
                idx[((size_t)i * m + j) * k + t] = bi[t];
            }
        }
        free(bd); free(bi);
    }
}

/* A6  knn_point_2 (tf_grouping.py:95-114) with batch_distance_matrix_general (:61-66):
 * D = r_q - 2*(q . p) + r_p, top_k(-D, k, sorted) -> returns (+D, point indices).
 * unique=True is a no-op in the reference (:89-91 rebinds a local).  Pinned summation:
 * r = fma chain over c from 0, dot = fma chain over c from 0 (this is exactly what a
 * v_mfma_f32 k-loop computes), D = (r_q - 2*dot) + r_p with two roundings.  TF's matmul /
 * reduce_sum order is not pinned in the reference ("parity unpinned"). */
ORC_API void orc_knn_feat(int b, int n, int m, int c, int k, const float *points, const float *queries,
                          float *dist, int *idx) {
#pragma omp parallel for schedule(dynamic) num_threads(orc_nt(b))
    for (int i = 0; i < b; ++i) {
        float *rp = (float *)malloc(sizeof(float) * n);
        float *bd = (float *)malloc(sizeof(float) * k);
        int *bi = (int *)malloc(sizeof(int) * k);
        for (int p = 0; p < n; ++p) {
            const float *s = points + ((size_t)i * n + p) * c;
            float r = 0.0f;
            for (int l = 0; l < c; ++l) r = fmaf(s[l], s[l], r);
            rp[p] = r;
        }
        for (int j = 0; j < m; ++j) {
            const float *q = queries + ((size_t)i * m + j) * c;
            float rq = 0.0f;
            for (int l = 0; l < c; ++l) rq = fmaf(q[l], q[l], rq);
            for (int t = 0; t < k; ++t) { bd[t] = INFINITY; bi[t] = 0; }
            for (int p = 0; p < n; ++p) {
                const float *s = points + ((size_t)i * n + p) * c;
                float dot = 0.0f;
                for (int l = 0; l < c; ++l) dot = fmaf(q[l], s[l], dot);
                const float t0 = rq - 2.0f * dot;
                const float d = t0 + rp[p];
                topk_insert(bd, bi, k, d, p);
            }
            for (int t = 0; t < k; ++t) {
                if (dist) dist[((size_t)i * m + j) * k + t] = bd[t];
                idx[((size_t)i * m + j) * k + t] = bi[t];
            }
        }
        free(rp); free(bd); free(bi);
    }
}

/* A8  three_nn.  threenn_cpu tf_ops/interpolation/tf_interpolate.cpp:60-103: for each of n
 * unknown points the three smallest squared distances to m known points; the distance is
 * evaluated in float ((x2-x1)^2 + (y2-y1)^2 + (z2-z1)^2, :73) and compared as double against
 * bests initialised to 1e40; strict '<' cascade, so equal distances keep the earlier index.
 * Because every candidate is a float widened to double, comparing as float is identical
 * except for the 1e40 sentinel (float +inf here), visible only when m < 3: the unfilled slots
 * then report idx 0 and dist = (float)1e40 = +inf in the reference as well. */
ORC_API void orc_three_nn(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist, int *idx,
                          int contract) {
#pragma omp parallel for schedule(dynamic) num_threads(orc_nt(b))
    for (int i = 0; i < b; ++i) {
        const float *p1 = xyz1 + (size_t)i * n * 3;
        const float *p2 = xyz2 + (size_t)i * m * 3;
        for (int j = 0; j < n; ++j) {
            const float x1 = p1[j * 3 + 0], y1 = p1[j * 3 + 1], z1 = p1[j * 3 + 2];
            double b1 = 1e40, b2 = 1e40, b3 = 1e40;
            int i1 = 0, i2 = 0, i3 = 0;
            for (int k = 0; k < m; ++k) {
                const double d = (double)sqdist3(p2[k * 3 + 0] - x1, p2[k * 3 + 1] - y1, p2[k * 3 + 2] - z1, contract);
                if (d < b1) { b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = k; }
                else if (d < b2) { b3 = b2; i3 = i2; b2 = d; i2 = k; }
                else if (d < b3) { b3 = d; i3 = k; }
            }
            float *dd = dist + ((size_t)i * n + j) * 3;
            int *ii = idx + ((size_t)i * n + j) * 3;
            dd[0] = (float)b1; dd[1] = (float)b2; dd[2] = (float)b3;
            ii[0] = i1; ii[1] = i2; ii[2] = i3;
        }
    }
}

/* A9  three_interpolate / grad.  threeinterpolate_cpu tf_interpolate.cpp:107-127:
 * out = (p1*w1 + p2*w2) + p3*w3, every op rounded.  threeinterpolate_grad_cpu :131-153:
 * grad_points[i_t] += grad_out*w_t in ascending (j, t) order into a zero-filled buffer. */
ORC_API void orc_three_interpolate(int b, int m, int c, int n, const float *points, const int *idx,
                                   const float *weight, float *out) {
    for (int i = 0; i < b; ++i)
        for (int j = 0; j < n; ++j) {
            const float *w = weight + ((size_t)i * n + j) * 3;
            const int *id = idx + ((size_t)i * n + j) * 3;
            const float *q1 = points + ((size_t)i * m + id[0]) * c;
            const float *q2 = points + ((size_t)i * m + id[1]) * c;
            const float *q3 = points + ((size_t)i * m + id[2]) * c;
            float *o = out + ((size_t)i * n + j) * c;
            for (int l = 0; l < c; ++l) {
                const float s = q1[l] * w[0] + q2[l] * w[1];
                o[l] = s + q3[l] * w[2];
            }
        }
}
ORC_API void orc_three_interpolate_grad(int b, int n, int c, int m, const float *grad_out, const int *idx,
                                        const float *weight, float *grad_points) {
    memset(grad_points, 0, sizeof(float) * (size_t)b * m * c);
    for (int i = 0; i < b; ++i)
        for (int j = 0; j < n; ++j) {
            const float *w = weight + ((size_t)i * n + j) * 3;
            const int *id = idx + ((size_t)i * n + j) * 3;
            const float *g = grad_out + ((size_t)i * n + j) * c;
            for (int l = 0; l < c; ++l)
                for (int t = 0; t < 3; ++t) grad_points[((size_t)i * m + id[t]) * c + l] += g[l] * w[t];
        }
}

/* A10  nn_distance.  NmDistanceKernel tf_ops/nn_distance/tf_nndistance_g.cu:5-127 and the CPU
 * nnsearch tf_nndistance.cpp:21-43: for every point of cloud 1 the minimum squared distance
 * to cloud 2 and its arg-min, diff = p2 - p1, d = x*x + y*y + z*z; first minimum wins
 * (strict '<' inside a tile :29,39; an earlier tile wins a tie across tiles :119), i.e. the
 * lowest index.  Both directions, like NmDistanceKernelLauncher :128-131. */
static void nnsearch_dir(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist, int *idx,
                         int contract) {
#pragma omp parallel for schedule(dynamic) num_threads(orc_nt(b))
    for (int i = 0; i < b; ++i)
        for (int j = 0; j < n; ++j) {
            const float *p = xyz1 + ((size_t)i * n + j) * 3;
            float best = 0.0f;
            int besti = 0;
            for (int k = 0; k < m; ++k) {
                const float *q = xyz2 + ((size_t)i * m + k) * 3;
                const float d = sqdist3(q[0] - p[0], q[1] - p[1], q[2] - p[2], contract);
                if (k == 0 || d < best) { best = d; besti = k; }
            }
            dist[(size_t)i * n + j] = best;
            idx[(size_t)i * n + j] = besti;
        }
}
ORC_API void orc_nn_distance(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist1, int *idx1,
                             float *dist2, int *idx2, int contract) {
    nnsearch_dir(b, n, m, xyz1, xyz2, dist1, idx1, contract);
    nnsearch_dir(b, m, n, xyz2, xyz1, dist2, idx2, contract);
}

/* nn_distance_grad.  NmDistanceGradKernel tf_nndistance_g.cu:132-157 / CPU tf_nndistance.cpp:126-163:
 * g = 2*grad_dist1[j]; grad_xyz1[j] += g*(p1-p2); grad_xyz2[idx1[j]] -= g*(p1-p2); then the same
 * with the roles swapped.  Oracle order = the CPU op's order (direction 1 then 2, ascending j). */
ORC_API void orc_nn_distance_grad(int b, int n, int m, const float *xyz1, const float *xyz2,
                                  const float *grad_dist1, const int *idx1, const float *grad_dist2,
                                  const int *idx2, float *grad_xyz1, float *grad_xyz2) {
    memset(grad_xyz1, 0, sizeof(float) * (size_t)b * n * 3);
    memset(grad_xyz2, 0, sizeof(float) * (size_t)b * m * 3);
    for (int i = 0; i < b; ++i) {
        for (int j = 0; j < n; ++j) {
            const int j2 = idx1[(size_t)i * n + j];
            const float g = grad_dist1[(size_t)i * n + j] * 2;
            for (int l = 0; l < 3; ++l) {
                const float v = g * (xyz1[((size_t)i * n + j) * 3 + l] - xyz2[((size_t)i * m + j2) * 3 + l]);
                grad_xyz1[((size_t)i * n + j) * 3 + l] += v;
                grad_xyz2[((size_t)i * m + j2) * 3 + l] -= v;
            }
        }
        for (int j = 0; j < m; ++j) {
            const int j2 = idx2[(size_t)i * m + j];
            const float g = grad_dist2[(size_t)i * m + j] * 2;
            for (int l = 0; l < 3; ++l) {
                const float v = g * (xyz2[((size_t)i * m + j) * 3 + l] - xyz1[((size_t)i * n + j2) * 3 + l]);
                grad_xyz2[((size_t)i * m + j) * 3 + l] += v;
                grad_xyz1[((size_t)i * n + j2) * 3 + l] -= v;
            }
        }
    }
}

/* A11  approx_match -- the GPU kernel is the spec: approxmatch tf_ops/approxmatch/tf_approxmatch_g.cu:1-179.
 *   multiL/multiR from integer n/m (:4-10); match zeroed, remainL = multiL, remainR = multiR (:15-20);
 *   10 levels j = 7..-2, level = -4^j, the last one 0 (:21-25);
 *   pass 1 (:26-59)  ratioL[k] = remainL[k] / (1e-9 + sum_l exp(level*d2)*remainR[l])   (l ascending)
 *   pass 2 (:75-108) sumr = (sum_k exp(level*d2)*ratioL[k]) * remainR[l];
 *                    ratioR[l] = min(remainR[l]/(sumr+1e-9), 1) * remainR[l];
 *                    remainR[l] = max(0, remainR[l]-sumr)
 *   pass 3 (:127-160) w = exp(level*d2)*ratioL[k]*ratioR[l]; match[l*n+k] += w;
 *                    remainL[k] = max(0, remainL[k] - sum_l w)
 * match layout is [b][m][n] (l-major, :152).  All sums are per-thread sequential in the
 * reference, so they are layout independent.  The reference uses the fast __expf; the oracle
 * uses libm expf (pinned = 0; tolerance-tested, SURVEY 8a A11: <= 1e-5 on match_cost) or the
 * bit-reproducible pinned_exp above (pinned = 1; bit-exact against the HIP library in the same mode).  d2 here is the
 * (x2-x1) form; the products level*d2 and exp*weight are plain multiplies, the running sums
 * are fused (suml += w with w a product -> fma) only in contract mode. */
/* `chunk` selects the association of the three running sums.  chunk <= 0: one sequential chain per point, the
 * reference kernel's order (a thread adds its m or n terms one after the other).  chunk = C > 0: the chain is cut into
 * consecutive pieces of C partners, each piece summed sequentially from 0, and the pieces are added in ascending
 * order (to 1e-9f for pass 1 -- unless ONE piece holds all m partners, then its chain starts at 1e-9f, i.e. m <= C is the reference's
 * order exactly -- to the first piece for passes 2 and 3).  This is the order of the MI355X kernels
 * (csrc/approxmatch.hip: one workgroup per (256 points) x (C partners) tile, AM_CH = 128), which lets one cloud
 * fill the chip; it differs from the sequential chain by reassociation only (tests bound the difference). */
ORC_API void orc_approx_match_chunked(int b, int n, int m, const float *xyz1, const float *xyz2, float *match,
                                      int contract, int pinned, int chunk) {
    const float multiL = (n >= m) ? 1.0f : (float)(m / n);
    const float multiR = (n >= m) ? (float)(n / m) : 1.0f;
    const int seq = chunk <= 0;                 /* sequential: pass 1's chain starts at 1e-9f, as the reference's does */
    if (seq) chunk = (n > m ? n : m);
#pragma omp parallel for schedule(dynamic) num_threads(orc_nt(b))
    for (int i = 0; i < b; ++i) {
        const float *p1 = xyz1 + (size_t)i * n * 3;
        const float *p2 = xyz2 + (size_t)i * m * 3;
        float *mt = match + (size_t)i * n * m;
        float *remainL = (float *)malloc(sizeof(float) * n), *ratioL = (float *)malloc(sizeof(float) * n);
        float *remainR = (float *)malloc(sizeof(float) * m), *ratioR = (float *)malloc(sizeof(float) * m);
        memset(mt, 0, sizeof(float) * (size_t)n * m);
        for (int k = 0; k < n; ++k) remainL[k] = multiL;
        for (int l = 0; l < m; ++l) remainR[l] = multiR;
        for (int j = 7; j >= -2; --j) {
            float level = -powf(4.0f, (float)j);
            if (j == -2) level = 0.0f;
            for (int k = 0; k < n; ++k) {
                const float x1 = p1[k * 3], y1 = p1[k * 3 + 1], z1 = p1[k * 3 + 2];
                float suml = 1e-9f;
                const int seq1 = seq || m <= chunk;      /* one piece holds every partner: its chain starts at 1e-9f like the reference's */
                for (int l0 = 0; l0 < m; l0 += chunk) {
                    const int lend = l0 + chunk < m ? l0 + chunk : m;
                    float s = seq1 ? 1e-9f : 0.0f;
                    for (int l = l0; l < lend; ++l) {
                        const float d2 = sqdist3(p2[l * 3] - x1, p2[l * 3 + 1] - y1, p2[l * 3 + 2] - z1, contract);
                        const float e = ORC_EXP(level * d2);
                        s = contract ? fmaf(e, remainR[l], s) : s + e * remainR[l];
                    }
                    suml = seq1 ? s : suml + s;
                }
                ratioL[k] = remainL[k] / suml;
            }
            for (int l = 0; l < m; ++l) {
                const float x2 = p2[l * 3], y2 = p2[l * 3 + 1], z2 = p2[l * 3 + 2];
                float sumr = 0.0f;
                for (int k0 = 0; k0 < n; k0 += chunk) {
                    const int kend = k0 + chunk < n ? k0 + chunk : n;
                    float s = 0.0f;
                    for (int k = k0; k < kend; ++k) {
                        const float d2 = sqdist3(x2 - p1[k * 3], y2 - p1[k * 3 + 1], z2 - p1[k * 3 + 2], contract);
                        const float e = ORC_EXP(level * d2);
                        s = contract ? fmaf(e, ratioL[k], s) : s + e * ratioL[k];
                    }
                    sumr = (k0 == 0) ? s : sumr + s;
                }
                sumr *= remainR[l];
                const float consumption = fminf(remainR[l] / (sumr + 1e-9f), 1.0f);
                ratioR[l] = consumption * remainR[l];
                remainR[l] = fmaxf(0.0f, remainR[l] - sumr);
            }
            for (int k = 0; k < n; ++k) {
                const float x1 = p1[k * 3], y1 = p1[k * 3 + 1], z1 = p1[k * 3 + 2];
                const float rl = ratioL[k];
                float suml = 0.0f;
                for (int l0 = 0; l0 < m; l0 += chunk) {
                    const int lend = l0 + chunk < m ? l0 + chunk : m;
                    float s = 0.0f;
                    for (int l = l0; l < lend; ++l) {
                        const float d2 = sqdist3(p2[l * 3] - x1, p2[l * 3 + 1] - y1, p2[l * 3 + 2] - z1, contract);
                        const float w = ORC_EXP(level * d2) * rl * ratioR[l];
                        mt[(size_t)l * n + k] += w;
                        s += w;
                    }
                    suml = (l0 == 0) ? s : suml + s;
                }
                remainL[k] = fmaxf(0.0f, remainL[k] - suml);
            }
        }
        free(remainL); free(ratioL); free(remainR); free(ratioR);
    }
}

ORC_API void orc_approx_match(int b, int n, int m, const float *xyz1, const float *xyz2, float *match,
                              int contract, int pinned) {
    orc_approx_match_chunked(b, n, m, xyz1, xyz2, match, contract, pinned, 0);     /* the reference's sequential order */
}

/* A12  match_cost.  matchcost tf_approxmatch_g.cu:183-225: thread t (of BS = 512, :227) sums
 * sqrtf(d2(k,l)) * match[l*n+k] over k = t, t+BS, ... (outer) and l ascending (inner), then a
 * binary shared-memory tree (:213-218) adds the partials.  The oracle reproduces exactly that
 * association so the float result is the reference's, up to sqrt/fma contraction. */
ORC_API void orc_match_cost(int b, int n, int m, const float *xyz1, const float *xyz2, const float *match,
                            float *cost, int contract, int bs) {
    if (bs <= 0) bs = 512;
#pragma omp parallel for schedule(dynamic) num_threads(orc_nt(b))
    for (int i = 0; i < b; ++i) {
        const float *p1 = xyz1 + (size_t)i * n * 3;
        const float *p2 = xyz2 + (size_t)i * m * 3;
        const float *mt = match + (size_t)i * n * m;
        float *part = (float *)calloc((size_t)bs, sizeof(float));
        for (int k = 0; k < n; ++k) {
            const float x1 = p1[k * 3], y1 = p1[k * 3 + 1], z1 = p1[k * 3 + 2];
            float s = part[k % bs];
            for (int l = 0; l < m; ++l) {
                const float d = sqrtf(sqdist3(p2[l * 3] - x1, p2[l * 3 + 1] - y1, p2[l * 3 + 2] - z1, contract));
                s = contract ? fmaf(d, mt[(size_t)l * n + k], s) : s + d * mt[(size_t)l * n + k];
            }
            part[k % bs] = s;
        }
        for (int j = 1; j < bs; j <<= 1)
            for (int t = 0; t + j < bs; t += 2 * j) part[t] += part[t + j];
        cost[i] = part[0];
        free(part);
    }
}

/* match_cost_grad.  matchcostgrad1 tf_approxmatch_g.cu:270-291 (per point of cloud 1, sequential
 * over cloud 2) and matchcostgrad2 :229-269 (per point of cloud 2: 256 threads stride over
 * cloud 1, binary tree).  d = match * rsqrtf(max(d2, 1e-20)); grad += delta * d.
 * The oracle uses 1/sqrtf for rsqrtf (tolerance-tested). */
ORC_API void orc_match_cost_grad(int b, int n, int m, const float *xyz1, const float *xyz2, const float *match,
                                 float *grad1, float *grad2, int contract) {
    const int bs = 256;
#pragma omp parallel for schedule(dynamic) num_threads(orc_nt(b))
    for (int i = 0; i < b; ++i) {
        const float *p1 = xyz1 + (size_t)i * n * 3;
        const float *p2 = xyz2 + (size_t)i * m * 3;
        const float *mt = match + (size_t)i * n * m;
        for (int l = 0; l < n; ++l) {
            const float x1 = p1[l * 3], y1 = p1[l * 3 + 1], z1 = p1[l * 3 + 2];
            float dx = 0, dy = 0, dz = 0;
            for (int k = 0; k < m; ++k) {
                const float ex = x1 - p2[k * 3], ey = y1 - p2[k * 3 + 1], ez = z1 - p2[k * 3 + 2];
                const float d = mt[(size_t)k * n + l] * (1.0f / sqrtf(fmaxf(sqdist3(ex, ey, ez, contract), 1e-20f)));
                if (contract) { dx = fmaf(ex, d, dx); dy = fmaf(ey, d, dy); dz = fmaf(ez, d, dz); }
                else { dx += ex * d; dy += ey * d; dz += ez * d; }
            }
            float *g = grad1 + ((size_t)i * n + l) * 3;
            g[0] = dx; g[1] = dy; g[2] = dz;
        }
        float *part = (float *)malloc(sizeof(float) * bs * 3);
        for (int k = 0; k < m; ++k) {
            const float x2 = p2[k * 3], y2 = p2[k * 3 + 1], z2 = p2[k * 3 + 2];
            memset(part, 0, sizeof(float) * bs * 3);
            for (int j = 0; j < n; ++j) {
                const float ex = x2 - p1[j * 3], ey = y2 - p1[j * 3 + 1], ez = z2 - p1[j * 3 + 2];
                const float d = mt[(size_t)k * n + j] * (1.0f / sqrtf(fmaxf(sqdist3(ex, ey, ez, contract), 1e-20f)));
                float *pp = part + (j % bs) * 3;
                if (contract) { pp[0] = fmaf(ex, d, pp[0]); pp[1] = fmaf(ey, d, pp[1]); pp[2] = fmaf(ez, d, pp[2]); }
                else { pp[0] += ex * d; pp[1] += ey * d; pp[2] += ez * d; }
            }
            for (int j = 1; j < bs; j <<= 1)
                for (int t = 0; t + j < bs; t += 2 * j)
                    for (int c = 0; c < 3; ++c) part[t * 3 + c] += part[(t + j) * 3 + c];
            float *g = grad2 + ((size_t)i * m + k) * 3;
            g[0] = part[0]; g[1] = part[1]; g[2] = part[2];
        }
        free(part);
    }
}

/* select_top_k / SelectionSort (tf_grouping_g.cu:83-123, selection_sort.cpp:20-63): copy dist to
 * out, outi[s] = s, then a partial selection sort of the first k entries of every row (swap of
 * values and indices, strict '<' so the first minimum wins).  Dead in the reference graph but
 * it carries the only deterministic known-answer harness (selection_sort.cpp:65-94). */
ORC_API void orc_selection_sort(int b, int n, int m, int k, const float *dist, int *outi, float *out) {
    for (size_t r = 0; r < (size_t)b * m; ++r) {
        float *o = out + r * n;
        int *oi = outi + r * n;
        for (int s = 0; s < n; ++s) { o[s] = dist[r * n + s]; oi[s] = s; }
        for (int s = 0; s < k; ++s) {
            int mn = s;
            for (int t = s + 1; t < n; ++t)
                if (o[t] < o[mn]) mn = t;
            if (mn != s) {
                const float tf = o[mn]; o[mn] = o[s]; o[s] = tf;
                const int ti = oi[mn]; oi[mn] = oi[s]; oi[s] = ti;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * prob_sample (optional / dead in the reference graph; boundary completeness).
 * Restates probsampleLauncher = cumsumKernel + binarysearchKernel, tf_ops/sampling/tf_sampling_g.cu:7-104.
 *   cumulative sums of inp[b, n] in chunks of 8192 values (:8,14): inside a chunk every aligned quad is summed as
 *   (v0, v0+v1, v2+(v0+v1), (v3+v2)+(v0+v1)) (:19-33; a ragged last quad sequentially from 0, :35-44), the quad totals go
 *   through a work-efficient scan -- up-sweep: total[((2k+2)<<u)-1] += total[((2k+1)<<u)-1] (:47-56), down-sweep:
 *   total[((2k+3)<<u)-1] += total[((2k+2)<<u)-1] for u descending (:57-67); a pair takes part iff its target index exists --
 *   element j of the chunk is then (inquad[j] + total[j/4 - 1]) + runningsum (:69-80), and the running sum is carried from
 *   chunk to chunk with a compensation term (:81-84).  The steps of one level touch disjoint pairs, so executing them one
 *   after the other gives the parallel kernel's values.  temp = the cumulative sums (the op's allocate_temp {b,n}).
 *   search (:90-104): q = r * cum[n-1]; idx = the first position whose cumulative sum is >= q, found by descending
 *   power-of-two steps from n-1.
 * PARITY UNPINNED against the .cu (no CUDA here, no CPU twin): self-checks in tests/ (cumsum within fp32 rounding of a
 * float64 cumsum; the result equals numpy.searchsorted on temp).
 * ------------------------------------------------------------------------------------------ */
ORC_API void orc_prob_sample(int b, int n, int m, const float *inp, const float *inpr, float *temp, int *out) {
    enum { CH = 8192 };
    float *in4 = (float *)malloc(sizeof(float) * CH);
    float *tot = (float *)malloc(sizeof(float) * (CH / 4));
    for (int i = 0; i < b; ++i) {
        const float *x = inp + (size_t)i * n;
        float *cum = temp + (size_t)i * n;
        float running = 0.0f, comp = 0.0f;
        for (int j = 0; j < n; j += CH) {
            const int cnt = n - j < CH ? n - j : CH;
            const int n24 = (cnt + 3) & ~3, n2 = n24 >> 2;
            for (int k = 0; k < cnt; k += 4) {
                if (k + 3 < cnt) {
                    const float v1 = x[j + k];
                    const float v2 = x[j + k + 1] + v1;
                    float v3 = x[j + k + 2];
                    float v4 = x[j + k + 3] + v3;
                    v3 = v3 + v2;
                    v4 = v4 + v2;
                    in4[k] = v1; in4[k + 1] = v2; in4[k + 2] = v3; in4[k + 3] = v4;
                    tot[k >> 2] = v4;
                } else {
                    float v = 0.0f;
                    for (int k2 = k; k2 < cnt; ++k2) { v = v + x[j + k2]; in4[k2] = v; }
                    for (int k2 = cnt; k2 < n24; ++k2) in4[k2] = v;
                    tot[k >> 2] = v;
                }
            }
            int u = 0;
            for (; (2 << u) <= n2; ++u)
                for (int k = 0; k < (n2 >> (u + 1)); ++k) tot[(((k << 1) + 2) << u) - 1] += tot[(((k << 1) + 1) << u) - 1];
            for (--u; u >= 0; --u)
                for (int k = 0; k < ((n2 - (1 << u)) >> (u + 1)); ++k) tot[(((k << 1) + 3) << u) - 1] += tot[(((k << 1) + 2) << u) - 1];
            for (int k = 0; k < cnt; ++k) {
                float v = in4[k];
                if (k >= 4) v = v + tot[(k >> 2) - 1];
                cum[j + k] = v + running;
            }
            const float t = tot[n2 - 1] + comp;
            const float r2 = running + t;
            comp = t - (r2 - running);
            running = r2;
        }
        int base = 1;
        while (base < n) base <<= 1;
        for (int jq = 0; jq < m; ++jq) {
            const float q = inpr[(size_t)i * m + jq] * cum[n - 1];
            int r = n - 1;
            for (int k = base; k >= 1; k >>= 1)
                if (r >= k && cum[r - k] >= q) r -= k;
            out[(size_t)i * m + jq] = r;
        }
    }
    free(in4);
    free(tot);
}
