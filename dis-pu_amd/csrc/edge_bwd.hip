// dense_conv + get_edge_feature (Common/ops.py:1856-1877,1897-1915) BACKWARD in one launch per dense block (training step).
//
// Forward (csrc/edge.hip, per pair row = point p x neighbour j):  y0 = [F_p | F_j - F_p] -> l0 = relu(y0.W0 + b0);
//   y1 = [l0 | F_p] -> l1 = relu(y1.W1 + b1);  y2 = [l1 | l0 | F_p] -> l2 = y2.W2 + b2;  out[p] = max_j [l2 | l1 | l0 | F_p].
// Round 2's training step kept the whole edge tensor [B*N*16, 72 + 2C] and its gradient in HBM and walked them with ~12
// launches per block (max gradient, three relu masks, three dW products with split reductions, three dX products, the
// edge-feature scatter).  Here NOTHING of the forward is stored: a workgroup recomputes l0 / l1 / l2 for 128 pair rows
// (8 points) from F and the neighbour ids -- the same k-ascending v_mfma_f32_32x32x2_f32 chains as the forward kernel, so every
// value, every ReLU decision and every arg-max tie is bit-identical to what the forward saw -- and runs the whole backward on
// chip:
//   max gradient (shared evenly by ties, math_grad._MinOrMaxGrad) -> dz2, da1, da0;
//   dy2 = dz2.W2^T (-> da1, da0, dF_p), dz1 = da1 * (l1 > 0);  dy1 = dz1.W1^T (-> da0, dF_p), dz0 = da0 * (l0 > 0);
//   dy0 = dz0.W0^T (-> dF_p, d(F_j - F_p));   dW_l += y_l^T.dz_l,  db_l += colsum dz_l.
// Activations and gradients of the tile live in LDS as row-major [128][.] matrices with odd pitches (conflict-free for both the
// "lane = row" operand reads of the row products and the "lane = column" reads of the weight-gradient products); the three
// layers' inputs are column slices of ONE matrix [l1 | l0 | F_p | F_j - F_p].  Wave w owns pair rows 32w .. 32w+31 (two points).
// dF leaves through float atomics (the centre's 16 rows are pre-reduced in registers); the weight gradients are accumulated in
// MFMA accumulators over the workgroup's tiles and written as ONE partial per workgroup, summed in a fixed order by
// edge_bwd_reduce_kernel (deterministic, like the other dW products).
#include "common.h"

#include <type_traits>

namespace dispu {

typedef float eb_f32x16 __attribute__((ext_vector_type(16)));

struct EdgeBwdArgs {
    int npoints, n_per_cloud;
    const float* F; long ldf;              // [npoints, C] block input
    const int* idx; int ldi, ioff;         // cloud-local neighbour ids, columns ioff .. ioff + 15
    const float *W0, *b0, *W1, *b1, *W2, *b2;
    const float* dOut; long lddo;          // [npoints, 72 + C] gradient of [max l2 | max l1 | max l0 | F_p]
    float* dF; long lddf;                  // [npoints, C] accumulates (atomics)
    float* part;                           // [gridDim.x][eb_part_floats(C)] per-workgroup weight / bias gradient partials
};

constexpr int EB_ROWS = 128, EB_G = 24, EB_LZ = 25, EB_LW = 25;
__host__ __device__ constexpr int eb_part_floats(int C) { return ((48 + C) + (24 + C) + 2 * C) * EB_G + 3 * EB_G; }

template <int C>
struct EbLds {
    static constexpr int K0 = 2 * C, K1 = EB_G + C, K2 = 2 * EB_G + C;
    static constexpr int WY = 2 * EB_G + 2 * C, LY = WY + 1;                       // [l1 | l0 | F_p | F_j - F_p], odd pitch
    static constexpr int FLOATS = EB_ROWS * LY + 3 * EB_ROWS * EB_LZ + (K0 + K1 + K2) * EB_LW + EB_ROWS + 32 * EB_LW + 64;   // + slack: operand
                                                                                   // reads of lanes whose results are discarded run past the arrays
    static constexpr size_t BYTES = (size_t)FLOATS * sizeof(float);
};

// D[32 x 32] += A[32 x 2 KSTEPS] . B[2 KSTEPS x 32], both operands read from LDS: step s uses the dwords at byte addresses
// a0 + s * SA and b0 + s * SB (the lane's A and B elements of k = 2s + h).  Hand-scheduled: a ring of R = 6 (or 4) operand pairs, the slot an
// MFMA has just consumed is refilled with the operands of R steps later, and the wait in front of an MFMA lets the 2 R - 2 younger reads
// stay in flight (s_waitcnt lgkmcnt(10)).  The compiler's own version of this loop waited for all but the two newest reads, i.e. gave
// every ds_read one MFMA (64 cycles) to land: ~185 cycles per step (tools/micro/edge_bwd_stamps.py).  No lane predicates: lanes whose
// operand lies outside the product (channels >= 24, k beyond the layer) read whatever follows in LDS and only feed accumulator rows /
// columns that are never written out.
__device__ __forceinline__ unsigned eb_lds_addr(const float* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) float*)p;
}
__device__ __forceinline__ void eb_dsread(float& dst, unsigned addr) { asm volatile("ds_read_b32 %0, %1" : "=v"(dst) : "v"(addr) : "memory"); }
template <int N> __device__ __forceinline__ void eb_wait() { asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(N) : "memory"); }

template <int KSTEPS, int SA, int SB, int R = (KSTEPS % 6 == 0 ? 6 : 4)>
__device__ __forceinline__ void eb_prod(eb_f32x16& acc, unsigned a0, unsigned b0) {
    static_assert(KSTEPS % R == 0 && KSTEPS >= 2 * R && 2 * R <= 14, "operand ring");      // lgkmcnt counts to 15
    float ra[R], rb[R];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // nothing of the surrounding code is in flight: the counts below are exact
#pragma unroll
    for (int u = 0; u < R; ++u) { eb_dsread(ra[u], a0 + u * SA); eb_dsread(rb[u], b0 + u * SB); }
    unsigned pa = a0 + R * SA, pb = b0 + R * SB;
#pragma unroll 1
    for (int s0 = 0; s0 < KSTEPS - R; s0 += R) {
#pragma unroll
        for (int u = 0; u < R; ++u) {
            eb_wait<2 * R - 2>();
            mfma_acc(acc, ra[u], rb[u]);
            eb_dsread(ra[u], pa + u * SA);
            eb_dsread(rb[u], pb + u * SB);
        }
        pa += R * SA; pb += R * SB;
    }
#pragma unroll
    for (int u = 0; u < R; ++u) {                                       // drain: 2 (R - 1 - u) younger reads may still be in flight
        if (u == 0) eb_wait<2 * R - 2>();
        else if (u == 1) eb_wait<(2 * R - 4 > 0 ? 2 * R - 4 : 0)>();
        else if (u == 2) eb_wait<(2 * R - 6 > 0 ? 2 * R - 6 : 0)>();
        else if (u == 3) eb_wait<(2 * R - 8 > 0 ? 2 * R - 8 : 0)>();
        else if (u == 4) eb_wait<(2 * R - 10 > 0 ? 2 * R - 10 : 0)>();
        else eb_wait<0>();
        mfma_acc(acc, ra[u], rb[u]);
    }
    mfma_acc_settle();
}

#ifdef EB_STAMPS
#define EB_T(i) do { __builtin_amdgcn_sched_barrier(0); if (threadIdx.x == 0 && blockIdx.x == 1) eb_st[i] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define EB_T(i)
#endif
template <int C>
__global__ __launch_bounds__(256) void edge_bwd_kernel(EdgeBwdArgs a) {
    using Ld = EbLds<C>;
#ifdef EB_STAMPS
    unsigned long long eb_st[16];
    for (int q = 0; q < 16; ++q) eb_st[q] = 0;
#endif
    EB_T(0);
    constexpr int K0 = Ld::K0, K1 = Ld::K1, K2 = Ld::K2, LY = Ld::LY, LZ = EB_LZ, LW = EB_LW, G = EB_G;
    constexpr int L1c = 0, L0c = G, FPc = 2 * G, DFc = 2 * G + C;                  // column offsets inside Y
    extern __shared__ __attribute__((aligned(16))) float eb_lds[];
    float* Y = eb_lds;
    float* Z2 = Y + EB_ROWS * LY;
    float* Z1 = Z2 + EB_ROWS * LZ;
    float* Z0 = Z1 + EB_ROWS * LZ;
    float* Wl0 = Z0 + EB_ROWS * LZ;                                                // [K][25] copies of the weights
    float* Wl1 = Wl0 + K0 * LW;
    float* Wl2 = Wl1 + K1 * LW;
    int* jrow = reinterpret_cast<int*>(Wl2 + K2 * LW);                             // global row id of every pair row's neighbour

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    {   // weights -> LDS [k][25]: float4 loads (a 96-byte row holds whole float4s), every load of a matrix in flight before the first store
        auto fill = [&](float* dst, const float* __restrict__ W, auto kc) {
            constexpr int NF4 = decltype(kc)::value * G / 4, PER = (NF4 + 255) / 256;
            float4 v[PER];
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int e = tid + u * 256;
                v[u] = e < NF4 ? *reinterpret_cast<const float4*>(W + e * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int e = tid + u * 256;
                if (e < NF4) {
                    const int k = (e * 4) / G, c = e * 4 - k * G;
                    float* d = dst + k * LW + c;
                    d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
                }
            }
        };
        fill(Wl0, a.W0, std::integral_constant<int, K0>{});
        fill(Wl1, a.W1, std::integral_constant<int, K1>{});
        fill(Wl2, a.W2, std::integral_constant<int, K2>{});
    }
    const float bias0 = i < G ? a.b0[i] : 0.f, bias1 = i < G ? a.b1[i] : 0.f, bias2 = i < G ? a.b2[i] : 0.f;
    EB_T(1);

    eb_f32x16 gw2, gw1, gw0;                                                       // weight-gradient tiles of waves 0..2 (k rows 32w ..)
#pragma unroll
    for (int r = 0; r < 16; ++r) { gw2[r] = 0.f; gw1[r] = 0.f; gw0[r] = 0.f; }
    float gb2 = 0.f, gb1 = 0.f, gb0 = 0.f;                                         // wave 3: bias gradients (lane = column, two row halves)

    const int ntiles = (a.npoints + 7) / 8;
    const int R0 = wave * 32;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int p0 = tile * 8;
        __syncthreads();                                                           // previous tile fully consumed (and the weights are in place)
        // ---- 1. gather [F_p | F_j - F_p] of the 128 pair rows
        {
            constexpr int PER = EB_ROWS * (C / 4) / 256;                           // 3 (C = 24) / 6 (C = 48) float4 pairs per thread
            int pj[PER], pp[PER];
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int e = tid + u * 256, row = e / (C / 4);
                int p = p0 + (row >> 4);
                if (p >= a.npoints) p = a.npoints - 1;
                pp[u] = p;
                pj[u] = a.idx[(size_t)p * a.ldi + a.ioff + (row & 15)];
            }
            float4 fpv[PER], fjv[PER];
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int e = tid + u * 256, row = e / (C / 4), q = e - row * (C / 4);
                pj[u] += (pp[u] / a.n_per_cloud) * a.n_per_cloud;
                fpv[u] = *reinterpret_cast<const float4*>(a.F + (size_t)pp[u] * a.ldf + q * 4);
                fjv[u] = *reinterpret_cast<const float4*>(a.F + (size_t)pj[u] * a.ldf + q * 4);
            }
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int e = tid + u * 256, row = e / (C / 4), q = e - row * (C / 4);
                const float4 fp = fpv[u], fj = fjv[u];
                float* y = Y + row * LY;
                y[FPc + q * 4 + 0] = fp.x; y[FPc + q * 4 + 1] = fp.y; y[FPc + q * 4 + 2] = fp.z; y[FPc + q * 4 + 3] = fp.w;
                y[DFc + q * 4 + 0] = fj.x - fp.x; y[DFc + q * 4 + 1] = fj.y - fp.y; y[DFc + q * 4 + 2] = fj.z - fp.z; y[DFc + q * 4 + 3] = fj.w - fp.w;
                if (q == 0) jrow[row] = pj[u];
            }
        }
        __syncthreads();
        EB_T(2);
        // ---- 2.-4. forward recompute: the k order of csrc/edge.hip (y0 = [F_p, F_j - F_p]; y1 = [l0, F_p]; y2 = [l1, l0, F_p])
        {
            eb_f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            eb_prod<K0 / 2, 8, 2 * LW * 4>(acc, eb_lds_addr(Y + (R0 + i) * LY + FPc + h), eb_lds_addr(Wl0 + h * LW + i));
            if (i < G) {
#pragma unroll
                for (int r = 0; r < 16; ++r) Y[(R0 + (r & 3) + 8 * (r >> 2) + 4 * h) * LY + L0c + i] = fmaxf(acc[r] + bias0, 0.f);
            }
        }
        __syncthreads();
        {
            eb_f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            eb_prod<K1 / 2, 8, 2 * LW * 4>(acc, eb_lds_addr(Y + (R0 + i) * LY + L0c + h), eb_lds_addr(Wl1 + h * LW + i));
            if (i < G) {
#pragma unroll
                for (int r = 0; r < 16; ++r) Y[(R0 + (r & 3) + 8 * (r >> 2) + 4 * h) * LY + L1c + i] = fmaxf(acc[r] + bias1, 0.f);
            }
        }
        __syncthreads();
        {
            eb_f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            eb_prod<K2 / 2, 8, 2 * LW * 4>(acc, eb_lds_addr(Y + (R0 + i) * LY + L1c + h), eb_lds_addr(Wl2 + h * LW + i));
            if (i < G) {
#pragma unroll
                for (int r = 0; r < 16; ++r) Z2[(R0 + (r & 3) + 8 * (r >> 2) + 4 * h) * LZ + i] = acc[r] + bias2;
            }
        }
        __syncthreads();
        EB_T(3);
        // ---- 5. gradient of the max over the 16 neighbours: (point, channel) items; ties share evenly.  Writes dz2 (over l2),
        //         and the max part of da1 / da0 into Z1 / Z0
        for (int it = tid; it < 8 * 72; it += 256) {
            const int pt = it / 72, c = it - pt * 72;
            const float* V; int ld; float* O;
            if (c < G) { V = Z2 + c; ld = LZ; O = Z2 + c; }
            else if (c < 2 * G) { V = Y + L1c + (c - G); ld = LY; O = Z1 + (c - G); }
            else { V = Y + L0c + (c - 2 * G); ld = LY; O = Z0 + (c - 2 * G); }
            float v[16];
            float m = -__builtin_inff();
#pragma unroll
            for (int s = 0; s < 16; ++s) { v[s] = V[(pt * 16 + s) * ld]; m = fmaxf(m, v[s]); }
            int cnt = 0;
#pragma unroll
            for (int s = 0; s < 16; ++s) cnt += (v[s] == m) ? 1 : 0;
            const int p = p0 + pt;
            const float g = (p < a.npoints) ? a.dOut[(size_t)p * a.lddo + c] : 0.f;
            const float share = g / (float)cnt;
#pragma unroll
            for (int s = 0; s < 16; ++s) O[(pt * 16 + s) * LZ] = (v[s] == m) ? share : 0.f;
        }
        // the centre's pass-through channels: out[p][72 + c] = F_p[c]
        for (int it = tid; it < 8 * C; it += 256) {
            const int pt = it / C, c = it - pt * C, p = p0 + pt;
            if (p < a.npoints) unsafeAtomicAdd(a.dF + (size_t)p * a.lddf + c, a.dOut[(size_t)p * a.lddo + 3 * G + c]);
        }
        __syncthreads();
        EB_T(4);

        // a 32 x 32 tile of d(layer input): column n (this lane) of rows R0 .. R0 + 31.  centre(): the F_p part -- the lane's 8 rows
        // of each of the wave's two points are summed, the two row halves (lanes i, i + 32) combined, one atomic per (point, channel).
        auto centre = [&](const eb_f32x16& acc, int ch, float sign, bool on) {
            float sA = 0.f, sB = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) { sA += acc[r]; sB += acc[r + 8]; }
            sA += __shfl_xor(sA, 32, 64);
            sB += __shfl_xor(sB, 32, 64);
            const int p = p0 + wave * 2 + h;
            if (on && p < a.npoints) unsafeAtomicAdd(a.dF + (size_t)p * a.lddf + ch, sign * (h ? sB : sA));
        };
        // ---- 6. dy2 = dz2 . W2^T over [l1 | l0 | F_p]
        for (int nt = 0; nt * 32 < K2; ++nt) {
            eb_f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const int n = nt * 32 + i;
            eb_prod<G / 2, 8, 8>(acc, eb_lds_addr(Z2 + (R0 + i) * LZ + h), eb_lds_addr(Wl2 + n * LW + h));
            if (n < G) {
#pragma unroll
                for (int r = 0; r < 16; ++r) Z1[(R0 + (r & 3) + 8 * (r >> 2) + 4 * h) * LZ + n] += acc[r];
            } else if (n < 2 * G) {
#pragma unroll
                for (int r = 0; r < 16; ++r) Z0[(R0 + (r & 3) + 8 * (r >> 2) + 4 * h) * LZ + n - G] += acc[r];
            }
            centre(acc, n - 2 * G, 1.f, n >= 2 * G && n < K2);
        }
        __syncthreads();
        for (int e = tid; e < EB_ROWS * G; e += 256) {                             // dz1 = da1 * (l1 > 0)
            const int row = e / G, c = e - row * G;
            if (!(Y[row * LY + L1c + c] > 0.f)) Z1[row * LZ + c] = 0.f;
        }
        __syncthreads();
        // ---- 7. dy1 = dz1 . W1^T over [l0 | F_p]
        for (int nt = 0; nt * 32 < K1; ++nt) {
            eb_f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const int n = nt * 32 + i;
            eb_prod<G / 2, 8, 8>(acc, eb_lds_addr(Z1 + (R0 + i) * LZ + h), eb_lds_addr(Wl1 + n * LW + h));
            if (n < G) {
#pragma unroll
                for (int r = 0; r < 16; ++r) Z0[(R0 + (r & 3) + 8 * (r >> 2) + 4 * h) * LZ + n] += acc[r];
            }
            centre(acc, n - G, 1.f, n >= G && n < K1);
        }
        __syncthreads();
        for (int e = tid; e < EB_ROWS * G; e += 256) {                             // dz0 = da0 * (l0 > 0)
            const int row = e / G, c = e - row * G;
            if (!(Y[row * LY + L0c + c] > 0.f)) Z0[row * LZ + c] = 0.f;
        }
        __syncthreads();
        EB_T(5);
        // ---- 8. dy0 = dz0 . W0^T over [F_p | F_j - F_p]: the second half goes to the neighbour (+) and to the centre (-)
        for (int nt = 0; nt * 32 < K0; ++nt) {
            eb_f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const int n = nt * 32 + i;
            eb_prod<G / 2, 8, 8>(acc, eb_lds_addr(Z0 + (R0 + i) * LZ + h), eb_lds_addr(Wl0 + n * LW + h));
            centre(acc, n < C ? n : n - C, n < C ? 1.f : -1.f, n < K0);
            if (n >= C && n < K0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = R0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (p0 + (row >> 4) < a.npoints) unsafeAtomicAdd(a.dF + (size_t)jrow[row] * a.lddf + (n - C), acc[r]);
                }
            }
        }
        EB_T(6);
        // ---- 9. weight gradients: dW_l[k][n] += sum_rows y_l[row][k] dz_l[row][n]  (contraction over the 128 rows; lane = k for the A
        //         operand, lane = n for the B operand, both read one LDS row per step); wave w < 3 owns k rows 32w .. 32w + 31
        if (wave < 3) {
            const int k = wave * 32 + i;
            if (wave * 32 < K2) eb_prod<EB_ROWS / 2, 2 * LY * 4, 2 * LZ * 4>(gw2, eb_lds_addr(Y + h * LY + L1c + k), eb_lds_addr(Z2 + h * LZ + i));
            if (wave * 32 < K1) eb_prod<EB_ROWS / 2, 2 * LY * 4, 2 * LZ * 4>(gw1, eb_lds_addr(Y + h * LY + L0c + k), eb_lds_addr(Z1 + h * LZ + i));
            if (wave * 32 < K0) eb_prod<EB_ROWS / 2, 2 * LY * 4, 2 * LZ * 4>(gw0, eb_lds_addr(Y + h * LY + FPc + k), eb_lds_addr(Z0 + h * LZ + i));
        } else if (i < G) {
            for (int row = h * 64; row < h * 64 + 64; ++row) { gb2 += Z2[row * LZ + i]; gb1 += Z1[row * LZ + i]; gb0 += Z0[row * LZ + i]; }
        }
    }
    EB_T(7);
#ifdef EB_STAMPS
    if (threadIdx.x == 0 && blockIdx.x == 1) {
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(a.part + (size_t)gridDim.x * eb_part_floats(C));
        for (int q = 0; q < 8; ++q) dbg[q] = eb_st[q];
    }
#endif
    // ---- the workgroup's partial: [W2 rows | W1 rows | W0 rows] x 24, then b2 | b1 | b0
    float* part = a.part + (size_t)blockIdx.x * eb_part_floats(C);
    if (wave < 3) {
        if (i < G) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (k < K2) part[k * G + i] = gw2[r];
                if (k < K1) part[(K2 + k) * G + i] = gw1[r];
                if (k < K0) part[(K2 + K1 + k) * G + i] = gw0[r];
            }
        }
    } else {
        gb2 += __shfl_xor(gb2, 32, 64); gb1 += __shfl_xor(gb1, 32, 64); gb0 += __shfl_xor(gb0, 32, 64);
        if (lane < G) {
            float* pb = part + (K2 + K1 + K0) * G;
            pb[lane] = gb2; pb[G + lane] = gb1; pb[2 * G + lane] = gb0;
        }
    }
}

// dst += sum over the workgroup partials.  One WAVE per element e of [W2 | W1 | W0 | b2 | b1 | b0]: lane l adds partials l, l + 64, ...
// in ascending order, a fixed butterfly combines the lanes (deterministic).  (One THREAD per element walked 256 partials as a
// dependent chain of strided loads: 60 us for 25 KB of output.)
__global__ __launch_bounds__(256) void edge_bwd_reduce_kernel(int C, int nparts, const float* __restrict__ part, float* __restrict__ dW2,
                                                               float* __restrict__ dW1, float* __restrict__ dW0, float* __restrict__ db2,
                                                               float* __restrict__ db1, float* __restrict__ db0) {
    const int K0 = 2 * C, K1 = EB_G + C, K2 = 2 * EB_G + C, total = eb_part_floats(C);
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (e >= total) return;
    float s = 0.f;
    for (int p = lane; p < nparts; p += 64) s += part[(size_t)p * total + e];
    s = wave_sum_f32(s);
    if (lane != 0) return;
    const int w2 = K2 * EB_G, w1 = K1 * EB_G, w0 = K0 * EB_G;
    if (e < w2) dW2[e] += s;
    else if (e < w2 + w1) dW1[e - w2] += s;
    else if (e < w2 + w1 + w0) dW0[e - w2 - w1] += s;
    else {
        const int b = e - (w2 + w1 + w0);
        if (b < EB_G) db2[b] += s;
        else if (b < 2 * EB_G) db1[b - EB_G] += s;
        else db0[b - 2 * EB_G] += s;
    }
}

static int eb_grid(int npoints) {
    const int ntiles = (npoints + 7) / 8;
    return ntiles < 256 ? ntiles : 256;
}

}  // namespace dispu

using namespace dispu;

DISPU_EXPORT long dispu_edge_dense_conv_grad_scratch_floats(int npoints, int C) {
    if (npoints <= 0 || !(C == 24 || C == 48)) return 0;
    return (long)eb_grid(npoints) * eb_part_floats(C) + 32;              // + room for the EB_STAMPS debug record
}

// Backward of dispu_edge_dense_conv (same F / idx / weights): dOut [npoints, 72 + C] -> dF [npoints, C] accumulates (atomics;
// zero-fill it or let it hold the gradient that arrived through other paths), dW* / db* accumulate (+=, deterministic).
// Two halves, so that a caller can keep the second one off its critical path (another stream, ordered after the first by an event):
// _partials runs the block's backward and leaves the weight gradients as per-workgroup partial sums in `scratch`; _reduce adds them,
// in a fixed order, to dW* / db*.  dispu_edge_dense_conv_grad = both on one stream.
DISPU_EXPORT int dispu_edge_dense_conv_grad_partials(int npoints, int n_per_cloud, int C, const float* F, long ldf, const int* idx, int ldi,
                                                     int ioff, const float* W0, const float* b0, const float* W1, const float* b1,
                                                     const float* W2, const float* b2, const float* dOut, long lddo, float* dF, long lddf,
                                                     float* scratch, long scratch_floats, void* stream) {
    if (npoints < 0 || n_per_cloud <= 0 || !(C == 24 || C == 48) || (ldf & 3) || (((uintptr_t)F) & 15) || !scratch ||
        scratch_floats < dispu_edge_dense_conv_grad_scratch_floats(npoints, C))
        return (int)hipErrorInvalidValue;
    if (npoints == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int grid = eb_grid(npoints);
    EdgeBwdArgs a{npoints, n_per_cloud, F, ldf, idx, ldi, ioff, W0, b0, W1, b1, W2, b2, dOut, lddo, dF, lddf, scratch};
    static DevOnce attr;
    if (attr.needed()) {
        DISPU_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(edge_bwd_kernel<24>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)EbLds<24>::BYTES));
        DISPU_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(edge_bwd_kernel<48>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)EbLds<48>::BYTES));
        attr.done();
    }
    if (C == 24) hipLaunchKernelGGL(edge_bwd_kernel<24>, dim3(grid), dim3(256), EbLds<24>::BYTES, s, a);
    else hipLaunchKernelGGL(edge_bwd_kernel<48>, dim3(grid), dim3(256), EbLds<48>::BYTES, s, a);
    return (int)hipGetLastError();
}

DISPU_EXPORT int dispu_edge_dense_conv_grad_reduce(int npoints, int C, const float* scratch, long scratch_floats, float* dW0, float* db0,
                                                   float* dW1, float* db1, float* dW2, float* db2, void* stream) {
    if (npoints < 0 || !(C == 24 || C == 48) || !scratch || scratch_floats < dispu_edge_dense_conv_grad_scratch_floats(npoints, C))
        return (int)hipErrorInvalidValue;
    if (npoints == 0) return 0;
    const int total = eb_part_floats(C);
    hipLaunchKernelGGL(edge_bwd_reduce_kernel, dim3((total + 3) / 4), dim3(256), 0, (hipStream_t)stream, C, eb_grid(npoints), scratch, dW2, dW1,
                       dW0, db2, db1, db0);
    return (int)hipGetLastError();
}

DISPU_EXPORT int dispu_edge_dense_conv_grad(int npoints, int n_per_cloud, int C, const float* F, long ldf, const int* idx, int ldi, int ioff,
                                            const float* W0, const float* b0, const float* W1, const float* b1, const float* W2,
                                            const float* b2, const float* dOut, long lddo, float* dF, long lddf, float* dW0, float* db0,
                                            float* dW1, float* db1, float* dW2, float* db2, float* scratch, long scratch_floats,
                                            void* stream) {
    const int rc = dispu_edge_dense_conv_grad_partials(npoints, n_per_cloud, C, F, ldf, idx, ldi, ioff, W0, b0, W1, b1, W2, b2, dOut, lddo, dF,
                                                       lddf, scratch, scratch_floats, stream);
    if (rc != 0 || npoints == 0) return rc;
    return dispu_edge_dense_conv_grad_reduce(npoints, C, scratch, scratch_floats, dW0, db0, dW1, db1, dW2, db2, stream);
}
