// Mixed-precision GEMMs of the training step (BASELINE configs[4]: "full train step ... bf16"): fp32 tensors in HBM (master
// weights, activations and gradients keep the reference's dtype, Common/tf_util.py:87-105), operands rounded to bf16
// (round-to-nearest-even, v_cvt_pk_bf16_f32) on their way into LDS, products on v_mfma_f32_32x32x16_bf16, fp32 accumulation
// and fp32 epilogue.  The reference is fp32-only: this is an extension with its own tolerance (tests/test_train_bf16_gpu.py:
// exact to fp32 rounding against a float64 product of the bf16-ROUNDED operands; ~3e-3 relative against the fp32 step).
//
// One kernel template covers the three products of a dense layer:
//   NN  Y  = X . W      forward            A = X [m][k] (k contiguous)         B = W [k][n] (n contiguous)
//   NT  dX = dY . W^T   input gradient     A = dY [m][n] (n contiguous)        B = W [k][n] read as [j = k][kk = n] (kk contiguous)
//   TN  dW = X^T . dZ   weight gradient    A = X read as [i = k][kk = m] (i contiguous), B = dZ [kk = m][n] (n contiguous)
// A tile is converted once per workgroup and lands in LDS as [row][k] bf16 (k contiguous, pitch 40 halves = 80 bytes), the
// layout both MFMA operands want: lane (r = lane & 31, h = lane >> 5) reads its 8 consecutive k of row r with one
// ds_read_b128.  Operands whose contiguous direction is not k are transposed in registers: a thread fetches 16 consecutive k of
// ONE row (coalesced across the wave along the rows) and stores them as two 16-byte LDS writes.
// 128 x 128 x 32 tiles, 4 waves (2 x 2, 64 x 64 each = 4 accumulators), register-prefetched double buffering.  With the
// matrix pipe 16x faster than in fp32 these GEMMs are bound by the operand stream (HBM / L2 -> LDS), not by MFMA issue.
// TN splits the long contraction (m = rows of the batch) over workgroups; partial tiles go to caller scratch and are added
// in split order by gemm_bf16_reduce_kernel (deterministic, no float atomics).
#include "common.h"

namespace dispu {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16v __attribute__((ext_vector_type(16)));

// LDS rows hold BK k (+8 pad) bf16: pitch 40 halves = 80 B (BK = 32) or 72 halves = 144 B (BK = 64); either way the 16 lanes of
// a ds_read_b128 phase hit 16 distinct 16-byte bank groups.  BK = 64 halves the number of (global-latency-bound) slab
// iterations of the small-tile configurations; the 128 x 128 tile stays at BK = 32 (40 KB of LDS for two stages).
constexpr int GB_KALIGN = 64;

struct GbArgs {
    int M, N, K;                       // output rows / columns, contraction length
    const float* A; long a_o, a_k, a_z;   // element (i, kk) of A at A + z*a_z + i*a_o + kk*a_k
    const float* B; long b_o, b_k, b_z;   // element (kk, j) of B at B + z*b_z + j*b_o + kk*b_k
    float* C; long ldc, c_z;
    const float* bias; int act;
    const float* R1; long ldr1, r1_z;
    const float* R2; long ldr2, r2_z;
    int splits, k_per_split;           // splits > 1: partial tiles to `part` [z][split][M (+1)][N]
    float* part;
    float* colsum; int colsum_acc;     // TN only: column sums of B (fp32, un-rounded) = the bias gradient; NULL = not wanted
    const float* Mk; long ldm; int mcols;   // optional ReLU-gradient mask, applied last (dispu_linear_bf16_masked)
    int a_bf16, b_bf16, c_bf16;             // operand / output tensors STORED as bf16 (2-byte elements behind the float* pointers;
                                            // strides stay in elements): the training step's bf16 activation storage
};

// four consecutive elements along the operand's contiguous direction, zero beyond `valid`
__device__ __forceinline__ float4 gb_load4(const float* p, int valid) {
    if (valid >= 4 && (((uintptr_t)p) & 15) == 0) return *reinterpret_cast<const float4*>(p);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid > 0) v.x = p[0];
    if (valid > 1) v.y = p[1];
    if (valid > 2) v.z = p[2];
    if (valid > 3) v.w = p[3];
    return v;
}

// Register image of a thread's share of one ROWS x 32 operand slab: ROWS / 8 floats.
//   KFAST  (k contiguous in memory): ROWS / 32 float4 along k; slot s = tid + 256 u -> row s >> 3, k = 4 (s & 7).
//   !KFAST (row index contiguous):   ROWS / 8 scalars along k for ONE row (row = tid % ROWS, k = (ROWS / 8) (tid / ROWS) + u):
//          the loads of a wave are coalesced along the rows, and the thread then owns consecutive k of its row = 16-byte LDS
//          stores, conflict-free at an 80-byte row pitch (transposing float4 loads needed 2-byte stores with 8-way conflicts).
template <int ROWS, int BK> struct GbRegs { float v[ROWS * BK / 256]; };

__device__ __forceinline__ float gb_bf16_to_f32(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }

// four consecutive bf16 elements (8 bytes), zero beyond `valid`
__device__ __forceinline__ float4 gb_load4_bf16(const unsigned short* p, int valid) {
    if (valid >= 4 && (((uintptr_t)p) & 7) == 0) {
        const uint2 w = *reinterpret_cast<const uint2*>(p);
        return make_float4(__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xFFFF0000u), __uint_as_float(w.y << 16), __uint_as_float(w.y & 0xFFFF0000u));
    }
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid > 0) v.x = gb_bf16_to_f32(p[0]);
    if (valid > 1) v.y = gb_bf16_to_f32(p[1]);
    if (valid > 2) v.z = gb_bf16_to_f32(p[2]);
    if (valid > 3) v.w = gb_bf16_to_f32(p[3]);
    return v;
}

template <int ROWS, int BK, bool KFAST>
__device__ __forceinline__ void gb_fetch(GbRegs<ROWS, BK>& reg, const float* base, long s_o, long s_k, int o0, int olim, int k0, int klim, int tid,
                                         int src_bf16) {
    if (src_bf16) {                                            // wave-uniform: the tensor is stored as bf16 (exact in the float registers)
        const unsigned short* hb = reinterpret_cast<const unsigned short*>(base);
        if constexpr (KFAST) {
            constexpr int Q = BK / 4;
#pragma unroll
            for (int u = 0; u < ROWS * Q / 256; ++u) {
                const int s = tid + 256 * u;
                const int o = o0 + s / Q, k = k0 + (s % Q) * 4;
                const float4 t = (o < olim) ? gb_load4_bf16(hb + (long)o * s_o + k, klim - k) : make_float4(0.f, 0.f, 0.f, 0.f);
                reg.v[4 * u + 0] = t.x; reg.v[4 * u + 1] = t.y; reg.v[4 * u + 2] = t.z; reg.v[4 * u + 3] = t.w;
            }
        } else {
            constexpr int KP = ROWS * BK / 256;
            const int o = o0 + (tid % ROWS), kb = k0 + (tid / ROWS) * KP;
            const bool ok = o < olim;
            const unsigned short* p = hb + (long)kb * s_k + o;
#pragma unroll
            for (int u = 0; u < KP; ++u) reg.v[u] = (ok && kb + u < klim) ? gb_bf16_to_f32(p[(long)u * s_k]) : 0.f;
        }
        return;
    }
    if constexpr (KFAST) {
        constexpr int Q = BK / 4;                              // float4 per row
#pragma unroll
        for (int u = 0; u < ROWS * Q / 256; ++u) {
            const int s = tid + 256 * u;
            const int o = o0 + s / Q, k = k0 + (s % Q) * 4;
            const float4 t = (o < olim) ? gb_load4(base + (long)o * s_o + k, klim - k) : make_float4(0.f, 0.f, 0.f, 0.f);
            reg.v[4 * u + 0] = t.x; reg.v[4 * u + 1] = t.y; reg.v[4 * u + 2] = t.z; reg.v[4 * u + 3] = t.w;
        }
    } else {
        constexpr int KP = ROWS * BK / 256;
        const int o = o0 + (tid % ROWS), kb = k0 + (tid / ROWS) * KP;
        const bool ok = o < olim;
        const float* p = base + (long)kb * s_k + o;
#pragma unroll
        for (int u = 0; u < KP; ++u) reg.v[u] = (ok && kb + u < klim) ? p[(long)u * s_k] : 0.f;
    }
}

template <int ROWS, int BK, bool KFAST>
__device__ __forceinline__ void gb_stage(const GbRegs<ROWS, BK>& reg, __bf16* tile, int tid) {
    constexpr int PITCH = BK + 8;
    if constexpr (KFAST) {
        constexpr int Q = BK / 4;
#pragma unroll
        for (int u = 0; u < ROWS * Q / 256; ++u) {
            const int s = tid + 256 * u;
            bf16x4 h = {(__bf16)reg.v[4 * u + 0], (__bf16)reg.v[4 * u + 1], (__bf16)reg.v[4 * u + 2], (__bf16)reg.v[4 * u + 3]};
            *reinterpret_cast<bf16x4*>(tile + (s / Q) * PITCH + (s % Q) * 4) = h;
        }
    } else {
        constexpr int KP = ROWS * BK / 256;
        __bf16* d = tile + (tid % ROWS) * PITCH + (tid / ROWS) * KP;
        if constexpr (KP >= 8) {
#pragma unroll
            for (int h2 = 0; h2 < KP / 8; ++h2) {
                bf16x8 h;
#pragma unroll
                for (int e = 0; e < 8; ++e) h[e] = (__bf16)reg.v[8 * h2 + e];
                *reinterpret_cast<bf16x8*>(d + 8 * h2) = h;
            }
        } else {
            bf16x4 h = {(__bf16)reg.v[0], (__bf16)reg.v[1], (__bf16)reg.v[2], (__bf16)reg.v[3]};
            *reinterpret_cast<bf16x4*>(d) = h;
        }
    }
}

// BM x BN output tile, WM x WN waves (WM * WN == 4), each wave (BM / WM) x (BN / WN) = TI x TJ blocks of 32 x 32.
template <int BM, int BN, int WM, int WN, int BK, bool A_KFAST, bool B_KFAST>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GbArgs a) {
    static_assert(WM * WN == 4 && (BM / WM) % 32 == 0 && (BN / WN) % 32 == 0, "wave grid");
    constexpr int GB_BK = BK, GB_PITCH = BK + 8;
    constexpr int TI = BM / WM / 32, TJ = BN / WN / 32, STAGE = (BM + BN) * GB_PITCH;
    __shared__ __attribute__((aligned(16))) __bf16 lds[2 * STAGE];
    __shared__ float csum_lds[256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, li = lane & 31, kq = lane >> 5;
    const int tiles_n = (a.N + BN - 1) / BN;
    const int m0 = (blockIdx.x / tiles_n) * BM, n0 = (blockIdx.x % tiles_n) * BN;
    const int z = blockIdx.z, split = blockIdx.y;
    const int kbeg = split * a.k_per_split, kend = min(a.K, kbeg + a.k_per_split);
    const float* __restrict__ Ab = a.a_bf16 ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(a.A) + (long)z * a.a_z) : a.A + (long)z * a.a_z;
    const float* __restrict__ Bb = a.b_bf16 ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(a.B) + (long)z * a.b_z) : a.B + (long)z * a.b_z;

    f32x16v acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    GbRegs<BM, BK> ra;
    GbRegs<BN, BK> rb;
    float csum = 0.f;                                          // !B_KFAST: this thread's column n0 + tid % BN of B, its k share
    const bool want_cs = !B_KFAST && a.colsum != nullptr && m0 == 0;
    const int nslab = (kend - kbeg + GB_BK - 1) / GB_BK;
    if (nslab > 0) {
        gb_fetch<BM, BK, A_KFAST>(ra, Ab, a.a_o, a.a_k, m0, a.M, kbeg, kend, tid, a.a_bf16);
        gb_fetch<BN, BK, B_KFAST>(rb, Bb, a.b_o, a.b_k, n0, a.N, kbeg, kend, tid, a.b_bf16);
        gb_stage<BM, BK, A_KFAST>(ra, lds, tid);
        gb_stage<BN, BK, B_KFAST>(rb, lds + BM * GB_PITCH, tid);
        if constexpr (!B_KFAST)
            if (want_cs) {
#pragma unroll
                for (int u = 0; u < BN * BK / 256; ++u) csum += rb.v[u];
            }
    }
    __syncthreads();
    for (int t = 0; t < nslab; ++t) {
        const __bf16* As = lds + (t & 1) * STAGE;
        const __bf16* Bs = As + BM * GB_PITCH;
        if (t + 1 < nslab) {                                   // next slab's global loads fly under this slab's MFMAs
            gb_fetch<BM, BK, A_KFAST>(ra, Ab, a.a_o, a.a_k, m0, a.M, kbeg + (t + 1) * GB_BK, kend, tid, a.a_bf16);
            gb_fetch<BN, BK, B_KFAST>(rb, Bb, a.b_o, a.b_k, n0, a.N, kbeg + (t + 1) * GB_BK, kend, tid, a.b_bf16);
        }
#pragma unroll
        for (int ks = 0; ks < GB_BK / 16; ++ks) {
            bf16x8 fa[TI], fb[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i)
                fa[i] = *reinterpret_cast<const bf16x8*>(As + (wm * (BM / WM) + i * 32 + li) * GB_PITCH + ks * 16 + kq * 8);
#pragma unroll
            for (int j = 0; j < TJ; ++j)
                fb[j] = *reinterpret_cast<const bf16x8*>(Bs + (wn * (BN / WN) + j * 32 + li) * GB_PITCH + ks * 16 + kq * 8);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (t + 1 < nslab) {
            __bf16* nxt = lds + ((t + 1) & 1) * STAGE;
            gb_stage<BM, BK, A_KFAST>(ra, nxt, tid);
            gb_stage<BN, BK, B_KFAST>(rb, nxt + BM * GB_PITCH, tid);
            if constexpr (!B_KFAST)
                if (want_cs) {
#pragma unroll
                    for (int u = 0; u < BN * BK / 256; ++u) csum += rb.v[u];
                }
        }
        __syncthreads();
    }

    // ---- bias gradient: column sums of the B operand's slice [kbeg, kend), combined over the 256 / BN threads of a column
    if constexpr (!B_KFAST) {
        if (a.colsum != nullptr && m0 == 0) {
            csum_lds[tid] = csum;
            __syncthreads();
            if (tid < BN && n0 + tid < a.N) {
                float sacc = csum_lds[tid];
#pragma unroll
                for (int g = 1; g < 256 / BN; ++g) sacc += csum_lds[tid + g * BN];
                if (a.splits > 1) a.part[((long)z * a.splits + split) * ((long)a.M + 1) * a.N + (long)a.M * a.N + n0 + tid] = sacc;
                else a.colsum[n0 + tid] = a.colsum_acc ? a.colsum[n0 + tid] + sacc : sacc;
            }
        }
    }

    // ---- epilogue (fp32): lane holds column .. + li, rows (r & 3) + 8 (r >> 2) + 4 kq of each 32 x 32 block
    if (a.splits > 1) {
        const long rows_p = a.colsum ? (long)a.M + 1 : (long)a.M;
        float* __restrict__ P = a.part + ((long)z * a.splits + split) * rows_p * a.N;
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const int col = n0 + wn * (BN / WN) + j * 32 + li;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kq;
                    if (row < a.M && col < a.N) P[(long)row * a.N + col] = acc[i][j][r];
                }
            }
        return;
    }
    float* __restrict__ C = a.C + (long)z * a.c_z;
    const float* __restrict__ R1 = a.R1 ? a.R1 + (long)z * a.r1_z : nullptr;
    const float* __restrict__ R2 = a.R2 ? a.R2 + (long)z * a.r2_z : nullptr;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int col = n0 + wn * (BN / WN) + j * 32 + li;
            if (col >= a.N) continue;
            const float bv = a.bias ? a.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kq;
                if (row >= a.M) continue;
                float v = acc[i][j][r] + bv;
                if (a.act == 1) v = fmaxf(v, 0.f);
                if (R1) v += R1[(long)row * a.ldr1 + col];
                if (R2) v += R2[(long)row * a.ldr2 + col];
                if (a.Mk && col < a.mcols) v = (a.Mk[(long)row * a.ldm + col] > 0.f) ? v : 0.f;
                if (a.c_bf16) reinterpret_cast<__bf16*>(a.C)[(long)z * a.c_z + (long)row * a.ldc + col] = (__bf16)v;
                else C[(long)row * a.ldc + col] = v;
            }
        }
}

// out[z][row][col] (+)= sum_s part[z][s][row][col]; row M (present when colsum != NULL) -> colsum[col].
// A workgroup = 32 elements x 8 split groups: thread (e, g) adds the partials of splits g, g + 8, ... in ascending order, the
// eight group sums are then added g = 0..7 in order.  Fixed association -> deterministic; 8x shorter serial chains and 8x
// more workgroups than one thread per element (the narrow dW outputs have only a few thousand elements but 100+ splits).
__global__ __launch_bounds__(256) void gemm_bf16_reduce_kernel(int M, int N, int splits, int with_colsum, const float* __restrict__ part,
                                                               float* __restrict__ out, long ldo, long so, int accumulate,
                                                               float* __restrict__ colsum) {
    __shared__ float red[8][32];
    const int z = blockIdx.y, el = threadIdx.x & 31, g = threadIdx.x >> 5;
    const long rows_p = with_colsum ? (long)M + 1 : (long)M;
    const long total = rows_p * N, stride = rows_p * N;
    const long e = (long)blockIdx.x * 32 + el;
    float s = 0.f;
    if (e < total) {
        const float* __restrict__ p = part + (long)z * splits * stride + e;
#pragma unroll 4
        for (int t = g; t < splits; t += 8) s += p[(long)t * stride];
    }
    red[g][el] = s;
    __syncthreads();
    if (g == 0 && e < total) {
        float r = red[0][el];
#pragma unroll
        for (int q = 1; q < 8; ++q) r += red[q][el];
        const long row = e / N, col = e % N;
        float* o = (row < M) ? out + (long)z * so + row * ldo + col : colsum + col;
        *o = accumulate ? *o + r : r;
    }
}

// tile choice: narrow outputs (N <= 32) take 128 x 32 tiles, launches that would not fill the chip 64 x 64, the rest 128 x 128
template <bool A_KFAST, bool B_KFAST>
static int gb_launch(const GbArgs& a, int batch, hipStream_t s) {
    auto tiles = [&](int bm, int bn) { return (long)((a.M + bm - 1) / bm) * ((a.N + bn - 1) / bn); };
    if (a.N <= 32) {
        hipLaunchKernelGGL((gemm_bf16_kernel<128, 32, 4, 1, 64, A_KFAST, B_KFAST>), dim3((unsigned)tiles(128, 32), a.splits, batch), dim3(256), 0, s, a);
    } else if (tiles(128, 128) * a.splits * batch < 512) {
        hipLaunchKernelGGL((gemm_bf16_kernel<64, 64, 2, 2, 64, A_KFAST, B_KFAST>), dim3((unsigned)tiles(64, 64), a.splits, batch), dim3(256), 0, s, a);
    } else {
        hipLaunchKernelGGL((gemm_bf16_kernel<128, 128, 2, 2, 32, A_KFAST, B_KFAST>), dim3((unsigned)tiles(128, 128), a.splits, batch), dim3(256), 0, s, a);
    }
    return (int)hipGetLastError();
}

static void tn_bf16_plan(int batch, int M, int K, int N, int& splits, int& per) {
    const int bm = (N <= 32) ? 128 : 64, bn = (N <= 32) ? 32 : 64;        // the finest tiling gb_launch may pick
    const long tiles = (long)((K + bm - 1) / bm) * ((N + bn - 1) / bn) * batch;
    long want = (1024 + tiles - 1) / tiles;                     // ~4 workgroups per CU: the contraction is a pure operand stream
    const long most = (M + 255) / 256;                          // at least 256 rows (4 - 8 slabs) per split
    if (want > most) want = most;
    if (want < 1) want = 1;
    per = (int)((((M + want - 1) / want) + GB_KALIGN - 1) / GB_KALIGN) * GB_KALIGN;
    splits = (M + per - 1) / per;
}

}  // namespace dispu

using namespace dispu;

// Same contract as dispu_linear (include/dispu_hip.h) with bf16 products: Y = R2 + R1 + act(X . W + bias).
DISPU_EXPORT int dispu_linear_bf16(int batch, int M, int K, int N, const float* X, long ldx, long sx, const float* W, long ldw, long sw,
                                   int transb, const float* bias, int act, float* Y, long ldy, long sy, const float* R1, long ldr1,
                                   long sr1, const float* R2, long ldr2, long sr2, void* stream) {
    if (batch < 0 || M < 0 || K < 0 || N < 0 || !X || !W || !Y || (act != 0 && act != 1)) return (int)hipErrorInvalidValue;
    if (batch == 0 || M == 0 || N == 0) return 0;
    int per = ((K + GB_KALIGN - 1) / GB_KALIGN) * GB_KALIGN;
    if (per == 0) per = GB_KALIGN;
    GbArgs a{M, N, K, X, ldx, 1, sx, W, transb ? ldw : 1, transb ? 1 : ldw, sw, Y, ldy, sy, bias, act, R1, ldr1, sr1, R2, ldr2, sr2,
             1, per, nullptr, nullptr, 0, nullptr, 0, 0};
    return transb ? gb_launch<true, true>(a, batch, (hipStream_t)stream) : gb_launch<true, false>(a, batch, (hipStream_t)stream);
}

// dispu_linear_bf16 with bf16-STORED tensors: storage bit 0: X holds bf16 elements, bit 2: Y is written as bf16 (W, bias, R1 stay
// fp32; R1 must be NULL when Y is bf16).  ldx / ldy / sx / sy count elements.
DISPU_EXPORT int dispu_linear_bf16s(int batch, int M, int K, int N, const void* X, long ldx, long sx, const float* W, long ldw, long sw,
                                    int transb, const float* bias, int act, void* Y, long ldy, long sy, const float* R1, long ldr1, long sr1,
                                    int storage, void* stream) {
    if (batch < 0 || M < 0 || K < 0 || N < 0 || !X || !W || !Y || (act != 0 && act != 1) || ((storage & 4) && R1)) return (int)hipErrorInvalidValue;
    if (batch == 0 || M == 0 || N == 0) return 0;
    int per = ((K + GB_KALIGN - 1) / GB_KALIGN) * GB_KALIGN;
    if (per == 0) per = GB_KALIGN;
    GbArgs a{M, N, K, (const float*)X, ldx, 1, sx, W, transb ? ldw : 1, transb ? 1 : ldw, sw, (float*)Y, ldy, sy, bias, act, R1, ldr1, sr1, nullptr, 0, 0,
             1, per, nullptr, nullptr, 0, nullptr, 0, 0, (storage & 1) ? 1 : 0, 0, (storage & 4) ? 1 : 0};
    return transb ? gb_launch<true, true>(a, batch, (hipStream_t)stream) : gb_launch<true, false>(a, batch, (hipStream_t)stream);
}

// dispu_linear_masked (include/dispu_hip.h) with bf16 products: Y = mask(R1 + act(X . W + bias)), Y = 0 where Mk <= 0 (columns < mcols).
DISPU_EXPORT int dispu_linear_bf16_masked(int batch, int M, int K, int N, const float* X, long ldx, long sx, const float* W, long ldw,
                                          long sw, int transb, const float* bias, int act, float* Y, long ldy, long sy, const float* R1,
                                          long ldr1, long sr1, const float* Mk, long ldm, int mcols, void* stream) {
    if (batch < 0 || M < 0 || K < 0 || N < 0 || !X || !W || !Y || (act != 0 && act != 1) || (Mk && batch != 1)) return (int)hipErrorInvalidValue;
    if (batch == 0 || M == 0 || N == 0) return 0;
    int per = ((K + GB_KALIGN - 1) / GB_KALIGN) * GB_KALIGN;
    if (per == 0) per = GB_KALIGN;
    GbArgs a{M, N, K, X, ldx, 1, sx, W, transb ? ldw : 1, transb ? 1 : ldw, sw, Y, ldy, sy, bias, act, R1, ldr1, sr1, nullptr, 0, 0,
             1, per, nullptr, nullptr, 0, (Mk && mcols > 0) ? Mk : nullptr, ldm, mcols};
    return transb ? gb_launch<true, true>(a, batch, (hipStream_t)stream) : gb_launch<true, false>(a, batch, (hipStream_t)stream);
}

DISPU_EXPORT long dispu_linear_tn_bf16_scratch_floats(int batch, int M, int K, int N) {
    if (batch <= 0 || M <= 0 || K <= 0 || N <= 0) return 0;
    int splits, per;
    tn_bf16_plan(batch, M, K, N, splits, per);
    return splits > 1 ? (long)batch * splits * ((long)K + 1) * N : 0;
}

// out[z] (+)= X[z]^T . Z[z]   (X [M, K], Z [M, N], out [K, N]); bf16 products, fp32 sums, deterministic split reduction.
// dbias (optional, batch == 1): (+)= column sums of Z, accumulated in fp32 from the UN-rounded values inside the same kernel.
DISPU_EXPORT int dispu_linear_tn_bf16s(int batch, int M, int K, int N, const void* X, long ldx, long sx, const void* Z, long ldz, long sz,
                                       float* out, long ldo, long so, int accumulate, float* dbias, float* scratch, long scratch_floats,
                                       int storage, void* stream);

DISPU_EXPORT int dispu_linear_tn_bf16(int batch, int M, int K, int N, const float* X, long ldx, long sx, const float* Z, long ldz,
                                      long sz, float* out, long ldo, long so, int accumulate, float* dbias, float* scratch,
                                      long scratch_floats, void* stream) {
    return dispu_linear_tn_bf16s(batch, M, K, N, X, ldx, sx, Z, ldz, sz, out, ldo, so, accumulate, dbias, scratch, scratch_floats, 0, stream);
}

// dispu_linear_tn_bf16 with bf16-STORED operands: storage bit 0: X holds bf16 elements, bit 1: Z does (the bias gradient is then the
// sum of the stored, i.e. rounded, values).
DISPU_EXPORT int dispu_linear_tn_bf16s(int batch, int M, int K, int N, const void* Xv, long ldx, long sx, const void* Zv, long ldz, long sz,
                                       float* out, long ldo, long so, int accumulate, float* dbias, float* scratch, long scratch_floats,
                                       int storage, void* stream) {
    const float* X = (const float*)Xv;
    const float* Z = (const float*)Zv;
    const int xa = (storage & 1) ? 1 : 0, zb = (storage & 2) ? 1 : 0;
    dispu_tn_reduce_desc* sink = tn_take_defer();                 // dispu_tn_defer (train_gemm.hip): describe the reduction, do not launch it
    if (sink) sink->splits = 0;
    if (batch != 1) sink = nullptr;
    if (batch < 0 || M < 0 || K < 0 || N < 0 || !out || (dbias && batch != 1)) return (int)hipErrorInvalidValue;
    if (batch == 0 || K == 0 || N == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (M == 0) {
        if (!accumulate) {
            for (int z = 0; z < batch; ++z) DISPU_TRY(hipMemset2DAsync(out + (size_t)z * so, sizeof(float) * ldo, 0, sizeof(float) * N, K, s));
            if (dbias) DISPU_TRY(hipMemsetAsync(dbias, 0, sizeof(float) * N, s));
        }
        return 0;
    }
    int splits, per;
    tn_bf16_plan(batch, M, K, N, splits, per);
    if (splits == 1) {       // one workgroup per output tile walks the whole contraction; accumulate = out as its own residual
        GbArgs a{K, N, M, X, 1, ldx, sx, Z, 1, ldz, sz, out, ldo, so, nullptr, 0, accumulate ? out : nullptr, ldo, so, nullptr, 0, 0, 1,
                 per, nullptr, dbias, accumulate, nullptr, 0, 0, xa, zb, 0};
        return gb_launch<false, false>(a, batch, s);
    }
    const long rows_p = dbias ? (long)K + 1 : (long)K;
    if (!scratch || scratch_floats < (long)batch * splits * rows_p * N) return (int)hipErrorInvalidValue;
    GbArgs a{K, N, M, X, 1, ldx, sx, Z, 1, ldz, sz, out, ldo, so, nullptr, 0, nullptr, 0, 0, nullptr, 0, 0, splits, per, scratch, dbias,
             accumulate, nullptr, 0, 0, xa, zb, 0};
    const int rc = gb_launch<false, false>(a, batch, s);
    if (rc != 0) return rc;
    const long total = rows_p * N;
    if (sink) {
        *sink = dispu_tn_reduce_desc{scratch, out, dbias, ldo, total, K, N, splits, (int)rows_p, accumulate, accumulate, 1, 0};
        return 0;
    }
    hipLaunchKernelGGL(gemm_bf16_reduce_kernel, dim3((unsigned)((total + 31) / 32), batch), dim3(256), 0, s, K, N, splits, dbias ? 1 : 0,
                       scratch, out, ldo, so, accumulate, dbias);
    return (int)hipGetLastError();
}
