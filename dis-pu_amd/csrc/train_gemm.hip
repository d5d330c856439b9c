// Weight-gradient GEMM and the column reductions of the training step for gfx950.
//
// The reference gets its gradients from TF1 autodiff (DisPU/model.py:178 `AdamOptimizer.minimize`): for every
// 1x1 conv / matmul  Y = act(X.W + b)  that is  dW = X^T.dZ,  db = colsum(dZ),  dX = dZ.W^T  with dZ = dY * act'(Y)
// (tf_util.py:52-185 layers; conv2d_backprop_filter / bias_add_grad / relu_grad).  dX is an ordinary NT product and
// reuses dispu_linear (transb = 1); this file holds the TN product and the reductions.
//
// linear_tn: out[k][n] (+)= sum_m X[m][k] * Z[m][n].  M (points x neighbours) is 10^4..10^6 while K x N is at most
// 2048 x 256, so the M axis is split over the grid: every workgroup owns one (K-tile, N-tile, M-split), streams its
// M range through LDS in 32-row slabs and feeds v_mfma_f32_32x32x2_f32 straight from the row-major slabs (A operand
// = a row pair of X, B operand = the same row pair of Z: neither needs a transpose).  Splits write partial tiles to
// a scratch buffer and a second kernel sums them in split order, so the result is deterministic (no float atomics).
#include "common.h"

#include <cstdio>
#include <cstdlib>

namespace dispu {

typedef float v16f __attribute__((ext_vector_type(16)));

struct TnArgs {
    int M, K, N;
    const float* X; long ldx; long sx;
    const float* Z; long ldz; long sz;
    float* out; long ldo; long so;       // final destination [K][N] (row stride ldo, batch stride so)
    float* part;                         // scratch [batch][splits][K + 1][N] (compact) when splits > 1 or accumulate;
                                         // row K of every partial = column sums of Z (the bias gradient)
    int splits, rows_per_split, direct;  // direct: single split, no accumulate, no bias -> write `out` from the GEMM kernel
    int want_bias;
};

constexpr int TN_SLAB = 16;              // rows of X / Z per LDS stage

// block = 2 x 2 MFMA waves + 4 loader waves; wave tile (32*TK) x (32*TNN); block tile (64*TK) x (64*TNN).
// Wave-specialised like the forward GEMM (linear.hip): the loader waves move the row slabs of X and Z (global ->
// registers -> LDS, one slab in flight), the MFMA waves only read operand fragments and issue MFMAs; one barrier
// per slab.  EDGE = false: every slab and tile is interior and 16-byte aligned -> no predicates.
template <int TK, int TNN, bool EDGE>
__global__ __launch_bounds__(512) void linear_tn_kernel(TnArgs a) {
    constexpr int BKT = 64 * TK, BNT = 64 * TNN;
    constexpr int LDXS = BKT + 32, LDZS = BNT + 32;      // +32 floats: the two half-waves of an operand read hit disjoint banks
    constexpr int STAGE = TN_SLAB * (LDXS + LDZS);
    constexpr int X_F4 = TN_SLAB * BKT / 4 / 256, Z_F4 = TN_SLAB * BNT / 4 / 256;
    static_assert(X_F4 >= 1 && Z_F4 >= 1, "slab too small for the workgroup");
    extern __shared__ __attribute__((aligned(16))) float tn_lds[];
    const int ntn = (a.N + BNT - 1) / BNT;
    const int tk = blockIdx.x / ntn, tn = blockIdx.x - tk * ntn;
    const int k0 = tk * BKT, n0 = tn * BNT;
    const int split = blockIdx.y, z = blockIdx.z;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int m_begin = split * a.rows_per_split;
    const int m_end = min(a.M, m_begin + a.rows_per_split);
    const int nslab = (m_end - m_begin + TN_SLAB - 1) / TN_SLAB;

    if (wave >= 4) {
        // ------------------------------------------------------------------------------------ loader waves
        const int tid = threadIdx.x - 256;
        const float* __restrict__ X = a.X + (size_t)z * a.sx;
        const float* __restrict__ Z = a.Z + (size_t)z * a.sz;
        const bool x_vec = ((a.ldx & 3) == 0) && ((((uintptr_t)X) & 15) == 0);
        const bool z_vec = ((a.ldz & 3) == 0) && ((((uintptr_t)Z) & 15) == 0);
        // one float4 of a row-major slab: element (m0 + r, c0 + 4*q)
        auto load4 = [&](const float* __restrict__ P, long ld, bool vec, int m0, int c0, int climit, int width, int it) -> float4 {
            const int idx = tid + it * 256;
            const int r = idx / (width / 4), q = idx % (width / 4);
            const int m = m0 + r, c = c0 + q * 4;
            if constexpr (!EDGE) return *reinterpret_cast<const float4*>(P + (size_t)m * ld + c);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < m_end && c < climit) {
                const float* p = P + (size_t)m * ld + c;
                if (vec && c + 3 < climit) {
                    v = *reinterpret_cast<const float4*>(p);
                } else {
                    v.x = p[0];
                    if (c + 1 < climit) v.y = p[1];
                    if (c + 2 < climit) v.z = p[2];
                    if (c + 3 < climit) v.w = p[3];
                }
            }
            return v;
        };
        float4 px[X_F4], pz[Z_F4];
        auto load_slab = [&](int m0) {
#pragma unroll
            for (int it = 0; it < X_F4; ++it) px[it] = load4(X, a.ldx, x_vec, m0, k0, a.K, BKT, it);
#pragma unroll
            for (int it = 0; it < Z_F4; ++it) pz[it] = load4(Z, a.ldz, z_vec, m0, n0, a.N, BNT, it);
        };
        auto store_slab = [&](int stage) {
            float* xs = tn_lds + stage * STAGE;
            float* zs = xs + TN_SLAB * LDXS;
#pragma unroll
            for (int it = 0; it < X_F4; ++it) {
                const int idx = tid + it * 256;
                *reinterpret_cast<float4*>(&xs[(idx / (BKT / 4)) * LDXS + (idx % (BKT / 4)) * 4]) = px[it];
            }
#pragma unroll
            for (int it = 0; it < Z_F4; ++it) {
                const int idx = tid + it * 256;
                *reinterpret_cast<float4*>(&zs[(idx / (BNT / 4)) * LDZS + (idx % (BNT / 4)) * 4]) = pz[it];
            }
        };
        if (nslab > 0) {
            load_slab(m_begin);
            store_slab(0);
            if (nslab > 1) load_slab(m_begin + TN_SLAB);
        }
        __syncthreads();
        for (int t = 0; t < nslab; ++t) {
            if (t + 1 < nslab) {
                store_slab((t + 1) & 1);
                if (t + 2 < nslab) load_slab(m_begin + (t + 2) * TN_SLAB);
            }
            __syncthreads();
        }
        return;
    }

    // ------------------------------------------------------------------------------------------ MFMA waves
    const int tid = threadIdx.x;
    const int wk = wave >> 1, wn = wave & 1;
    v16f acc[TK][TNN];
#pragma unroll
    for (int i = 0; i < TK; ++i)
#pragma unroll
        for (int j = 0; j < TNN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float bsum = 0.f;                    // column sum of Z for column n0 + tid (K-tile 0 only)
    const bool do_bias = a.want_bias && tk == 0 && tid < BNT;
    __syncthreads();
    const int fi = lane & 31, fk = lane >> 5;
    for (int t = 0; t < nslab; ++t) {
        const float* xs = tn_lds + (t & 1) * STAGE;
        const float* zs = xs + TN_SLAB * LDXS;
#pragma unroll
        for (int kk = 0; kk < TN_SLAB; kk += 2) {
            float af[TK], bf[TNN];
#pragma unroll
            for (int i = 0; i < TK; ++i) af[i] = xs[(kk + fk) * LDXS + (wk * TK + i) * 32 + fi];
#pragma unroll
            for (int j = 0; j < TNN; ++j) bf[j] = zs[(kk + fk) * LDZS + (wn * TNN + j) * 32 + fi];
#pragma unroll
            for (int i = 0; i < TK; ++i)
#pragma unroll
                for (int jn = 0; jn < TNN; ++jn)
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[jn], acc[i][jn], 0, 0, 0);
        }
        if (do_bias) {
#pragma unroll
            for (int r = 0; r < TN_SLAB; ++r) bsum += zs[r * LDZS + tid];
        }
        __syncthreads();
    }

    // C/D layout: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    float* __restrict__ dst;
    long ldd;
    if (a.direct) {
        dst = a.out + (size_t)z * a.so;
        ldd = a.ldo;
    } else {
        dst = a.part + ((size_t)z * a.splits + split) * (size_t)(a.K + 1) * a.N;
        ldd = a.N;
        if (do_bias && n0 + tid < a.N) dst[(size_t)a.K * a.N + n0 + tid] = bsum;
    }
#pragma unroll
    for (int i = 0; i < TK; ++i)
#pragma unroll
        for (int jn = 0; jn < TNN; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = k0 + (wk * TK + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                const int n = n0 + (wn * TNN + jn) * 32 + fi;
                if (!EDGE || (k < a.K && n < a.N)) dst[(size_t)k * ldd + n] = acc[i][jn][r];
            }
}

template <int TK, int TNN, bool EDGE>
static int launch_tn(const TnArgs& a, dim3 grid, hipStream_t s) {
    constexpr size_t bytes = (size_t)2 * TN_SLAB * (64 * TK + 32 + 64 * TNN + 32) * sizeof(float);
    static_assert(bytes <= 64 * 1024, "TN stage exceeds the default dynamic LDS limit");
    hipLaunchKernelGGL((linear_tn_kernel<TK, TNN, EDGE>), grid, dim3(512), bytes, s, a);
    return (int)hipGetLastError();
}

template <int TK, int TNN>
static int launch_tn_e(const TnArgs& a, dim3 grid, bool edge, hipStream_t s) {
    return edge ? launch_tn<TK, TNN, true>(a, grid, s) : launch_tn<TK, TNN, false>(a, grid, s);
}

// out[z][k][n] = (accumulate ? out : 0) + sum_s part[z][s][k][n]; row K of the partials goes to dbias[n] (always
// accumulating, batch 0..).  One workgroup = 64 consecutive outputs x 8 split groups: group g sums splits
// g, g+8, ... (coalesced 256-byte rows), the 8 group sums are added in group order -> deterministic.
__global__ __launch_bounds__(512) void tn_reduce_kernel(int batch, int K, int N, int splits, const float* __restrict__ part,
                                                        float* __restrict__ out, long ldo, long so, int accumulate,
                                                        float* __restrict__ dbias) {
    __shared__ float red[8][64];
    const size_t kn1 = (size_t)(K + 1) * N;
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const size_t chunks = (kn1 + 63) / 64;
    for (size_t blk = blockIdx.x; blk < chunks * batch; blk += gridDim.x) {
        const size_t z = blk / chunks, r = (blk - z * chunks) * 64 + lane;
        float v = 0.f;
        if (r < kn1) {
            const float* p = part + z * splits * kn1 + r;
            float v0 = 0.f, v1 = 0.f;
            // splits grp, grp + 8, grp + 16, ... alternate between two running sums (even / odd position); eight loads are
            // requested before the first add (as a rolled load-add loop this was a chain of ~16 L2 round trips: 7 us per launch)
            for (int s = grp; s < splits; s += 64) {
                float b[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) b[j] = (s + 8 * j < splits) ? p[(size_t)(s + 8 * j) * kn1] : 0.f;
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    if (s + 8 * j < splits) v0 += b[j];
                    if (s + 8 * (j + 1) < splits) v1 += b[j + 1];
                }
            }
            v = v0 + v1;
        }
        red[grp][lane] = v;
        __syncthreads();
        if (grp == 0 && r < kn1) {
            float t = red[0][lane];
#pragma unroll
            for (int g = 1; g < 8; ++g) t += red[g][lane];
            const int k = (int)(r / N), n = (int)(r - (size_t)k * N);
            if (k < K) {
                float* o = out + z * so + (size_t)k * ldo + n;
                *o = accumulate ? *o + t : t;
            } else if (dbias) {
                if (batch == 1) dbias[n] += t;
                else unsafeAtomicAdd(dbias + n, t);
            }
        }
        __syncthreads();
    }
}

// The same sum four outputs per lane (N % 4 == 0, 16-byte aligned rows): a workgroup = 32 float4 columns x 8 split groups, every lane
// keeps up to eight 16-byte loads in flight.  Same association as tn_reduce_kernel (group g: splits g, g + 8, ... alternating between
// two running sums; groups added in order), so both kernels give the same bits.  The scalar kernel moved the 67 MB of partials of
// the 2048 x 256 gradient at 2 TB/s (33 us); this one is bandwidth-bound.
__global__ __launch_bounds__(256) void tn_reduce4_kernel(int batch, int K, int N, int splits, const float* __restrict__ part,
                                                         float* __restrict__ out, long ldo, long so, int accumulate,
                                                         float* __restrict__ dbias) {
    __shared__ float4 red[8][32];
    const size_t kn1 = (size_t)(K + 1) * N, q1 = kn1 / 4;
    const int lane = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const size_t chunks = (q1 + 31) / 32;
    for (size_t blk = blockIdx.x; blk < chunks * batch; blk += gridDim.x) {
        const size_t z = blk / chunks, q = (blk - z * chunks) * 32 + lane;
        float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
        if (q < q1) {
            const float4* p = reinterpret_cast<const float4*>(part + z * splits * kn1) + q;
            for (int s = grp; s < splits; s += 64) {
                float4 b[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) b[j] = (s + 8 * j < splits) ? p[(size_t)(s + 8 * j) * q1] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    if (s + 8 * j < splits) { v0.x += b[j].x; v0.y += b[j].y; v0.z += b[j].z; v0.w += b[j].w; }
                    if (s + 8 * (j + 1) < splits) { v1.x += b[j + 1].x; v1.y += b[j + 1].y; v1.z += b[j + 1].z; v1.w += b[j + 1].w; }
                }
            }
        }
        red[grp][lane] = make_float4(v0.x + v1.x, v0.y + v1.y, v0.z + v1.z, v0.w + v1.w);
        __syncthreads();
        if (grp == 0 && q < q1) {
            float4 t = red[0][lane];
#pragma unroll
            for (int g = 1; g < 8; ++g) { t.x += red[g][lane].x; t.y += red[g][lane].y; t.z += red[g][lane].z; t.w += red[g][lane].w; }
            const size_t r = q * 4;
            const int k = (int)(r / N), n = (int)(r - (size_t)k * N);
            if (k < K) {
                float4* o = reinterpret_cast<float4*>(out + z * so + (size_t)k * ldo + n);
                if (accumulate) { const float4 c = *o; t.x = c.x + t.x; t.y = c.y + t.y; t.z = c.z + t.z; t.w = c.w + t.w; }
                *o = t;
            } else if (dbias) {
                if (batch == 1) { dbias[n] += t.x; dbias[n + 1] += t.y; dbias[n + 2] += t.z; dbias[n + 3] += t.w; }
                else { unsafeAtomicAdd(dbias + n, t.x); unsafeAtomicAdd(dbias + n + 1, t.y); unsafeAtomicAdd(dbias + n + 2, t.z); unsafeAtomicAdd(dbias + n + 3, t.w); }
            }
        }
        __syncthreads();
    }
}

// ---- deferred reductions, one grouped launch -------------------------------------------------------------------------------------
// The training step runs ~20 weight-gradient products of very different sizes, each followed by its own split reduction: 20 more
// launches of 4 - 25 us that only Adam waits for.  dispu_tn_defer(&desc) arms a one-shot sink: the next TN product on this thread
// launches its product kernel only and describes the reduction it left undone; dispu_tn_reduce_grouped runs any number of described
// reductions as ONE launch (a workgroup looks its descriptor up by block index), in exactly the association the single launches use.
static thread_local dispu_tn_reduce_desc* tl_tn_defer = nullptr;
dispu_tn_reduce_desc* tn_take_defer() {
    dispu_tn_reduce_desc* d = tl_tn_defer;
    tl_tn_defer = nullptr;
    return d;
}

__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float f4_add(float a, float b) { return a + b; }
template <typename T> __device__ __forceinline__ T f4_zero();
template <> __device__ __forceinline__ float f4_zero<float>() { return 0.f; }
template <> __device__ __forceinline__ float4 f4_zero<float4>() { return make_float4(0.f, 0.f, 0.f, 0.f); }

__host__ __device__ inline bool tn_desc_vec(const dispu_tn_reduce_desc& d) {
    return (d.N % 4 == 0) && (d.ldo % 4 == 0) && (d.stride % 4 == 0) &&
           (((((uintptr_t)d.part) | ((uintptr_t)d.out) | ((uintptr_t)d.dbias)) & 15) == 0);
}
__host__ __device__ inline unsigned tn_desc_chunks(const dispu_tn_reduce_desc& d) {
    const size_t e = (size_t)d.rows_p * d.N;
    return (unsigned)(((tn_desc_vec(d) ? e / 4 : e) + 31) / 32);
}

// one 32-element (32-quad) chunk of one descriptor: 8 split groups x 32 lanes, the associations of tn_reduce(4)_kernel (assoc 0: group g
// adds splits g, g + 8, ... alternating between two running sums) and of gemm_bf16_reduce_kernel / tn_stream_reduce_kernel (assoc 1: one
// running sum per group); the eight group sums are added in group order
template <typename T>
__device__ __forceinline__ void tn_reduce_chunk(const dispu_tn_reduce_desc& d, unsigned chunk, T (*red)[32]) {
    constexpr int W = sizeof(T) / 4;
    const int lane = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const size_t q1 = (size_t)d.rows_p * d.N / W, q = (size_t)chunk * 32 + lane;
    const size_t sq = (size_t)d.stride / W;
    T v = f4_zero<T>();
    if (q < q1) {
        const T* __restrict__ p = reinterpret_cast<const T*>(d.part) + q;
        if (d.assoc == 0) {
            T v0 = f4_zero<T>(), v1 = f4_zero<T>();
            for (int s = grp; s < d.splits; s += 64) {
                T b[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) b[j] = (s + 8 * j < d.splits) ? p[(size_t)(s + 8 * j) * sq] : f4_zero<T>();
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    if (s + 8 * j < d.splits) v0 = f4_add(v0, b[j]);
                    if (s + 8 * (j + 1) < d.splits) v1 = f4_add(v1, b[j + 1]);
                }
            }
            v = f4_add(v0, v1);
        } else {
#pragma unroll 4
            for (int t = grp; t < d.splits; t += 8) v = f4_add(v, p[(size_t)t * sq]);
        }
    }
    red[grp][lane] = v;
    __syncthreads();
    if (grp == 0 && q < q1) {
        T t = red[0][lane];
#pragma unroll
        for (int g = 1; g < 8; ++g) t = f4_add(t, red[g][lane]);
        const size_t r = q * W;
        const int k = (int)(r / d.N), n = (int)(r - (size_t)k * d.N);
        T* o;
        bool acc;
        if (k < d.K) { o = reinterpret_cast<T*>(d.out + (size_t)k * d.ldo + n); acc = d.accumulate != 0; }
        else { o = reinterpret_cast<T*>(d.dbias + n); acc = d.bias_accumulate != 0; }
        if (k < d.K || d.dbias) *o = acc ? f4_add(*o, t) : t;
    }
}

__global__ __launch_bounds__(256) void tn_reduce_grouped_kernel(int count, const dispu_tn_reduce_desc* __restrict__ tab) {
    __shared__ float4 red[8][32];
    unsigned b = blockIdx.x;
    int i = 0;
    dispu_tn_reduce_desc d = tab[0];
    for (;;) {                                                    // uniform: block index -> (descriptor, chunk)
        const unsigned c = tn_desc_chunks(d);
        if (b < c || i + 1 >= count) break;
        b -= c;
        d = tab[++i];
    }
    if (tn_desc_vec(d)) tn_reduce_chunk<float4>(d, b, red);
    else tn_reduce_chunk<float>(d, b, reinterpret_cast<float(*)[32]>(red));
}

static void launch_tn_reduce(int batch, int K, int N, int splits, const float* part, float* out, long ldo, long so, int accumulate,
                             float* dbias, hipStream_t s) {
    const bool vec = (N % 4 == 0) && (ldo % 4 == 0) && (so % 4 == 0) && ((((uintptr_t)part) | ((uintptr_t)out)) & 15) == 0;
    if (vec) {
        const size_t chunks = ((((size_t)(K + 1) * N) / 4 + 31) / 32) * batch;
        hipLaunchKernelGGL(tn_reduce4_kernel, dim3((unsigned)(chunks > 16384 ? 16384 : chunks)), dim3(256), 0, s, batch, K, N, splits, part, out,
                           ldo, so, accumulate, dbias);
    } else {
        const size_t chunks = (((size_t)(K + 1) * N + 63) / 64) * batch;
        hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)(chunks > 8192 ? 8192 : chunks)), dim3(512), 0, s, batch, K, N, splits, part, out, ldo,
                           so, accumulate, dbias);
    }
}

// ---- narrow outputs (N <= 64, K <= 256: the edge-conv layers' [96 x 24]-sized weight gradients over 10^4 .. 10^5 rows) --------
// The tiled kernel above gives such a product ONE 128 x 64 block tile that is < 30 % full and as many workgroups as it has
// M-splits, each paying a full LDS pipeline for a few KB of output (18 us + the reduction for 16 MB of operands).  Here a
// WAVE owns one 16 x 16 output tile (k-tile, n-tile) and one chunk of rows, and runs v_mfma_f32_16x16x4_f32 straight from
// global memory: both operands are row-major with the contraction index as the row, i.e. already in the layout the
// instruction wants (lane (i, q): X[m + q][16 kt + i] and Z[m + q][16 nt + i]) - no LDS, no transposes.  Loads are buffer
// loads with the row offset as the instruction's scalar offset (no VALU per load), 16 steps in flight.  One extra "k-tile"
// whose A operand is 1 in row 0 produces the column sums of Z (the bias gradient) on the same pipe.  Partials go to the
// scratch layout of tn_reduce_kernel ([chunk][K + 1][N]), which sums them in chunk order: deterministic.
typedef float tn_f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(1024) void linear_tn_narrow_kernel(int M, int K, int N, int rows_per_chunk, int KT, int NT,
                                                                 const float* __restrict__ X, long ldx, const float* __restrict__ Z,
                                                                 long ldz, float* __restrict__ part) {
    const int lane = threadIdx.x & 63;
    const int job = blockIdx.y * (blockDim.x >> 6) + (threadIdx.x >> 6);          // (k-tile, n-tile) of this wave
    if (job >= (KT + 1) * NT) return;
    const int kt = job / NT, nt = job - kt * NT;
    const int i = lane & 15, q = lane >> 4;
    const int chunk = blockIdx.x;
    const int m0 = chunk * rows_per_chunk;
    const int steps = min(rows_per_chunk, M - m0) >> 2;                             // M % 4 == 0 (checked by the caller)
    const bool is_bias = kt == KT;
    const int kcol = 16 * kt + i, ncol = 16 * nt + i;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, (int)((long)M * ldx * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Z), 0, (int)((long)M * ldz * 4), 0x00020000);
    const int xoff = (q * (int)ldx + min(kcol, K - 1)) * 4, zoff = (q * (int)ldz + min(ncol, N - 1)) * 4;   // bytes
    const int xstep = 4 * (int)ldx * 4, zstep = 4 * (int)ldz * 4;                   // four rows per MFMA step
    const float amask = (!is_bias && kcol < K) ? 1.f : 0.f, bmask = (ncol < N) ? 1.f : 0.f;
    const float aone = (is_bias && i == 0) ? 1.f : 0.f;
    tn_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    constexpr int U = 16;
    int sx = m0 * (int)ldx * 4, sz = m0 * (int)ldz * 4;
    for (int s0 = 0; s0 < steps; s0 += U) {
        float av[U], bv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool on = s0 + u < steps;                                         // wave-uniform
            av[u] = (on && !is_bias) ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, xoff, sx + u * xstep, 0)) : 0.f;
            bv[u] = on ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rz, zoff, sz + u * zstep, 0)) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (s0 + u < steps) {
                // columns past K / N read a clamped (valid) address: zero them (x * 0 with finite x; the bias tile's A is constant)
                const float a = is_bias ? aone : av[u] * amask;
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[u] * bmask, acc, 0, 0, 0);
            }
        }
        sx += U * xstep; sz += U * zstep;
    }
    // acc[r] = D[k = 16 kt + 4 q + r][n = 16 nt + i]; the bias tile's row 0 = column sums -> row K of the partial
    float* dst = part + (size_t)chunk * (size_t)(K + 1) * N;
    if (ncol < N) {
        if (is_bias) {
            if (q == 0) dst[(size_t)K * N + ncol] = acc[0];
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = 16 * kt + 4 * q + r;
                if (k < K) dst[(size_t)k * N + ncol] = acc[r];
            }
        }
    }
}

// chunks of the narrow path (0 = not applicable): ~2048 waves in total, 128 .. 4096 rows per chunk (multiples of 64)
static int tn_narrow_chunks(int batch, int M, int K, int N, int& rows) {
    rows = 0;
    if (batch != 1 || N > 64 || K > 256 || M < 4096 || (M & 3)) return 0;
    const int jobs = ((K + 15) / 16 + 1) * ((N + 15) / 16);
    int want = (2048 + jobs - 1) / jobs;
    rows = (M + want - 1) / want;
    rows = ((rows + 63) / 64) * 64;
    if (rows < 128) rows = 128;
    if (rows > 4096) rows = 4096;
    return (M + rows - 1) / rows;
}

static void tn_plan(int batch, int M, int K, int N, int& tk, int& tnn, int& splits, int& rows) {
    constexpr int want_wgs = 768, min_slabs = 16, force_tnn = 0, fill = 256;     // (swept in rounds 3 - 4: profiles/r03_tn_bench_variants.txt)
    const int max_splits = (M + min_slabs * TN_SLAB - 1) / (min_slabs * TN_SLAB);  // at least 16 slabs (256 rows) per split
    // the largest block tile whose (tiles x possible M-splits) still fills the chip: with M = 8192 rows (8 training patches) a
    // 256 x 256 output as two 128 x 256 tiles gave 64 workgroups on 256 CUs (44 us); as eight 64 x 128 tiles it is 256
    const int tk_max = (K > 64) ? 2 : 1, tnn_max = (N > 128) ? 4 : (N > 64) ? 2 : 1;
    const int cand[4][2] = {{2, 4}, {2, 2}, {1, 2}, {1, 1}};
    tk = 1, tnn = 1;
    for (int c = 0; c < 4; ++c) {
        const int ck = cand[c][0] < tk_max ? cand[c][0] : tk_max, cn = cand[c][1] < tnn_max ? cand[c][1] : tnn_max;
        const long t = (long)((K + 64 * ck - 1) / (64 * ck)) * ((N + 64 * cn - 1) / (64 * cn)) * batch;
        tk = ck, tnn = cn;
        if (t * max_splits >= fill) break;
    }
    if (force_tnn > 0 && force_tnn < tnn) tnn = force_tnn;
    const int tiles = ((K + 64 * tk - 1) / (64 * tk)) * ((N + 64 * tnn - 1) / (64 * tnn)) * batch;
    int want = (want_wgs + tiles - 1) / tiles;                    // aim at ~768 workgroups (3 per CU); 512 with 512-row splits measured 9 % slower per step
    splits = want < 1 ? 1 : want;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    rows = (M + splits - 1) / splits;
    rows = ((rows + TN_SLAB - 1) / TN_SLAB) * TN_SLAB;
    splits = (M + rows - 1) / rows;
}

// ---- column sums with the activation mask ------------------------------------------------------------------
// dZ[r][j] = dY[r][j] * (act ? Y[r][j] > 0 : 1);  part[blk][j] = sum over the block's rows of dZ[r][j].
// block (64, 4): x runs over columns (coalesced), y over 4 interleaved rows.
__global__ __launch_bounds__(256) void act_bias_grad_kernel(long rows, int n, int rows_per_block, const float* __restrict__ dY,
                                                             long lddy, const float* __restrict__ Y, long ldy, int act,
                                                             float* __restrict__ dZ, long lddz, float* __restrict__ part) {
    __shared__ float red[4][64];
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = min(rows, r0 + rows_per_block);
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int c = c0 + tx;
        float s = 0.f;
        if (c < n) {
            for (long r = r0 + ty; r < r1; r += 4) {
                float g = dY[r * lddy + c];
                if (act && !(Y[r * ldy + c] > 0.f)) g = 0.f;
                if (dZ) dZ[r * lddz + c] = g;
                s += g;
            }
        }
        red[ty][tx] = s;
        __syncthreads();
        if (ty == 0 && c < n && part) part[(size_t)blockIdx.x * n + c] = ((red[0][tx] + red[1][tx]) + red[2][tx]) + red[3][tx];
        __syncthreads();
    }
}

// mask only (no bias): flat, coalesced
__global__ void act_mask_kernel(long rows, int n, const float* __restrict__ dY, long lddy, const float* __restrict__ Y, long ldy,
                                float* __restrict__ dZ, long lddz) {
    const size_t total = (size_t)rows * n;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t r = e / n;
        const int c = (int)(e - r * n);
        const float g = dY[r * lddy + c];
        dZ[r * lddz + c] = (Y[r * ldy + c] > 0.f) ? g : 0.f;
    }
}

// the same mask on float4 quads (n % 4 == 0, 16-byte aligned rows): 32-bit index arithmetic, one division per FOUR elements
// (the scalar kernel divides a 64-bit index per element: 10 us for 2 M elements, launched 23 times per training step)
__global__ void act_mask_v4_kernel(unsigned quads, unsigned n4, const float* __restrict__ dY, unsigned lddy, const float* __restrict__ Y,
                                   unsigned ldy, float* __restrict__ dZ, unsigned lddz) {
    for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < quads; e += gridDim.x * blockDim.x) {
        const unsigned r = e / n4, c = (e - r * n4) * 4;
        const float4 g = *reinterpret_cast<const float4*>(dY + (size_t)r * lddy + c);
        const float4 y = *reinterpret_cast<const float4*>(Y + (size_t)r * ldy + c);
        float4 o;
        o.x = (y.x > 0.f) ? g.x : 0.f; o.y = (y.y > 0.f) ? g.y : 0.f; o.z = (y.z > 0.f) ? g.z : 0.f; o.w = (y.w > 0.f) ? g.w : 0.f;
        *reinterpret_cast<float4*>(dZ + (size_t)r * lddz + c) = o;
    }
}

// out[j] = (accumulate ? out[j] : 0) + sum_s part[s][j]  (s ascending)
__global__ void colsum_reduce_kernel(int n, int nparts, const float* __restrict__ part, float* __restrict__ out, int accumulate) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    float v = accumulate ? out[j] : 0.f;
    for (int s = 0; s < nparts; ++s) v += part[(size_t)s * n + j];
    out[j] = v;
}

static int bias_blocks(long rows, int& rows_per_block) {
    rows_per_block = 256;
    long nb = (rows + rows_per_block - 1) / rows_per_block;
    if (nb > 2048) {
        rows_per_block = (int)((rows + 2047) / 2048);
        rows_per_block = (rows_per_block + 3) & ~3;
        nb = (rows + rows_per_block - 1) / rows_per_block;
    }
    return (int)nb;
}

}  // namespace dispu

using namespace dispu;

DISPU_EXPORT long dispu_linear_tn_scratch_floats(int batch, int M, int K, int N) {
    if (batch <= 0 || M <= 0 || K <= 0 || N <= 0) return 0;
    int tk, tnn, splits, rows;
    tn_plan(batch, M, K, N, tk, tnn, splits, rows);
    int nrows;
    const int nchunks = tn_narrow_chunks(batch, M, K, N, nrows);
    if (nchunks > splits) splits = nchunks;
    return (long)batch * splits * (K + 1) * N;
}

DISPU_EXPORT int dispu_linear_tn(int batch, int M, int K, int N, const float* X, long ldx, long sx, const float* Z, long ldz,
                                 long sz, float* out, long ldo, long so, int accumulate, float* dbias, float* scratch,
                                 long scratch_floats, void* stream) {
    dispu_tn_reduce_desc* sink = tn_take_defer();                 // armed by dispu_tn_defer: describe the reduction instead of launching it
    if (sink) sink->splits = 0;
    if (batch != 1) sink = nullptr;                               // batched products (the unfused attention backward) reduce themselves
    if (batch < 0 || M < 0 || K < 0 || N < 0) return (int)hipErrorInvalidValue;
    if (batch == 0 || K == 0 || N == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (M == 0) {   // nothing to add to dbias
        if (!accumulate)
            for (int z = 0; z < batch; ++z)
                DISPU_TRY(hipMemset2DAsync(out + (size_t)z * so, sizeof(float) * ldo, 0, sizeof(float) * N, K, s));
        return 0;
    }
    {
        int nrows;
        const int nchunks = tn_narrow_chunks(batch, M, K, N, nrows);
        if (nchunks > 0 && (long)M * ldx < (1l << 29) && (long)M * ldz < (1l << 29) && scratch != nullptr &&
            scratch_floats >= (long)nchunks * (K + 1) * N) {
            const int KT = (K + 15) / 16, NT = (N + 15) / 16, jobs = (KT + 1) * NT;
            const int wpb = jobs < 16 ? jobs : 16;
            hipLaunchKernelGGL(linear_tn_narrow_kernel, dim3(nchunks, (jobs + wpb - 1) / wpb), dim3(64 * wpb), 0, s, M, K, N, nrows, KT, NT,
                               X, ldx, Z, ldz, scratch);
            DISPU_CHECK_LAUNCH();
            if (sink) {
                *sink = dispu_tn_reduce_desc{scratch, out, dbias, ldo, (long)(K + 1) * N, K, N, nchunks, K + 1, accumulate, 1, 0, 0};
                return 0;
            }
            launch_tn_reduce(1, K, N, nchunks, scratch, out, ldo, so, accumulate, dbias, s);
            DISPU_CHECK_LAUNCH();
            return 0;
        }
    }
    int tk, tnn, splits, rows;
    tn_plan(batch, M, K, N, tk, tnn, splits, rows);
    // (measured and not kept: every split adding its tile to `out` with float atomics instead of partial tiles + reduction --
    // 124 vs 111 us for the 2048 x 256 gradient, 69 vs 54 us for 131072 rows x 128 x 128, a few us better only on the small ones)
    const int direct = (splits == 1 && !accumulate && !dbias) ? 1 : 0;
    if (!direct && (scratch == nullptr || scratch_floats < (long)batch * splits * (K + 1) * N)) return (int)hipErrorInvalidValue;
    TnArgs a{M, K, N, X, ldx, sx, Z, ldz, sz, out, ldo, so, scratch, splits, rows, direct, dbias ? 1 : 0};
    const int tiles = ((K + 64 * tk - 1) / (64 * tk)) * ((N + 64 * tnn - 1) / (64 * tnn));
    dim3 grid(tiles, splits, batch);
    const bool edge = !(K % (64 * tk) == 0 && N % (64 * tnn) == 0 && M % TN_SLAB == 0 && (ldx & 3) == 0 && (ldz & 3) == 0 &&
                        (sx & 3) == 0 && (sz & 3) == 0 && (((uintptr_t)X) & 15) == 0 && (((uintptr_t)Z) & 15) == 0);
    int rc;
    if (tk == 1 && tnn == 1) rc = launch_tn_e<1, 1>(a, grid, edge, s);
    else if (tk == 1 && tnn == 2) rc = launch_tn_e<1, 2>(a, grid, edge, s);
    else if (tk == 1) rc = launch_tn_e<1, 4>(a, grid, edge, s);
    else if (tnn == 1) rc = launch_tn_e<2, 1>(a, grid, edge, s);
    else if (tnn == 2) rc = launch_tn_e<2, 2>(a, grid, edge, s);
    else rc = launch_tn_e<2, 4>(a, grid, edge, s);
    if (rc != 0) return rc;
    if (!direct) {
        if (sink) {
            *sink = dispu_tn_reduce_desc{scratch, out, dbias, ldo, (long)(K + 1) * N, K, N, splits, K + 1, accumulate, 1, 0, 0};
            return 0;
        }
        launch_tn_reduce(batch, K, N, splits, scratch, out, ldo, so, accumulate, dbias, s);
        DISPU_CHECK_LAUNCH();
    }
    return 0;
}

DISPU_EXPORT int dispu_tn_defer(dispu_tn_reduce_desc* desc) {
    if (!desc) return (int)hipErrorInvalidValue;
    desc->splits = 0;
    tl_tn_defer = desc;
    return 0;
}

DISPU_EXPORT int dispu_tn_reduce_grouped(int count, const dispu_tn_reduce_desc* table_host, const dispu_tn_reduce_desc* table_device,
                                         void* stream) {
    if (count < 0 || (count > 0 && (!table_host || !table_device))) return (int)hipErrorInvalidValue;
    unsigned long chunks = 0;
    for (int i = 0; i < count; ++i) {
        const dispu_tn_reduce_desc& d = table_host[i];
        if (d.splits <= 0 || !d.part || !d.out || d.K <= 0 || d.N <= 0 || d.rows_p < d.K || d.rows_p > d.K + 1 || (unsigned)d.assoc > 1u)
            return (int)hipErrorInvalidValue;
        chunks += tn_desc_chunks(d);
    }
    if (chunks == 0) return 0;
    if (chunks > 0x7fffffffUL) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(tn_reduce_grouped_kernel, dim3((unsigned)chunks), dim3(256), 0, (hipStream_t)stream, count, table_device);
    return (int)hipGetLastError();
}

DISPU_EXPORT long dispu_act_bias_grad_scratch_floats(long rows, int n) {
    if (rows <= 0 || n <= 0) return 0;
    int rpb;
    return (long)bias_blocks(rows, rpb) * n;
}

DISPU_EXPORT int dispu_act_bias_grad(long rows, int n, const float* dY, long lddy, const float* Y, long ldy, int act, float* dZ,
                                     long lddz, float* dbias, int accumulate, float* scratch, long scratch_floats,
                                     void* stream) {
    if (rows < 0 || n < 0 || (act && Y == nullptr)) return (int)hipErrorInvalidValue;
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (rows == 0) {
        if (dbias && !accumulate) DISPU_TRY(hipMemsetAsync(dbias, 0, sizeof(float) * n, s));
        return 0;
    }
    if (!dbias) {
        if (!act || !dZ) return 0;
        const size_t total = (size_t)rows * n;
        if ((n & 3) == 0 && (lddy & 3) == 0 && (ldy & 3) == 0 && (lddz & 3) == 0 && total / 4 < 0x7fffffffull && lddy < 0x7fffffffl &&
            ldy < 0x7fffffffl && lddz < 0x7fffffffl && ((((uintptr_t)dY) | ((uintptr_t)Y) | ((uintptr_t)dZ)) & 15) == 0) {
            const unsigned quads = (unsigned)(total / 4);
            const unsigned gq = (quads + 255) / 256;
            hipLaunchKernelGGL(act_mask_v4_kernel, dim3(gq > 16384 ? 16384 : gq), dim3(256), 0, s, quads, (unsigned)(n / 4), dY, (unsigned)lddy, Y,
                               (unsigned)ldy, dZ, (unsigned)lddz);
            return (int)hipGetLastError();
        }
        const size_t g = (total + 255) / 256;
        hipLaunchKernelGGL(act_mask_kernel, dim3((unsigned)(g > 32768 ? 32768 : g)), dim3(256), 0, s, rows, n, dY, lddy, Y, ldy, dZ, lddz);
        return (int)hipGetLastError();
    }
    int rpb;
    const int nb = bias_blocks(rows, rpb);
    if (dbias && (scratch == nullptr || scratch_floats < (long)nb * n)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(act_bias_grad_kernel, dim3(nb), dim3(256), 0, s, rows, n, rpb, dY, lddy, Y, ldy, act, dZ, lddz,
                       dbias ? scratch : nullptr);
    DISPU_CHECK_LAUNCH();
    if (dbias) {
        hipLaunchKernelGGL(colsum_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, nb, scratch, dbias, accumulate);
        DISPU_CHECK_LAUNCH();
    }
    return 0;
}
