// PointNonLocalCell attention for the TRAINING step (Common/ops.py:326-339 and its TF1 autodiff), fp32, gfx950.
//
// The reference (and round 3's training path here) materialises softmax(Q.K^T / 8) as a [B, M, M] tensor, keeps it for the
// backward pass and runs five more [M, M]-sized products over it (dP, dV, softmax', dQ, dK): 2 x 33 MB per 8 patches written,
// read back three times, 9 launches.  Here nothing of size M x M ever exists:
//
//   forward   O = softmax(scale Q K^T) V  flash-style (csrc/attention.hip's scheme) + the row statistic
//             lse2[q] = log2 sum_k 2^(s2[q][k])  with s2 = scale log2(e) Q.K^T          (one float per query)
//   backward  P is RECOMPUTED tile by tile from Q, K and lse2:  P[q][k] = 2^(s2[q][k] - lse2[q]),  then
//                 D[q]   = sum_d dO[q][d] O[q][d]
//                 dP     = dO . V^T            dS = P o (dP - D)
//                 dQ     = scale dS . K        dK = scale dS^T . Q        dV = P^T . dO
//             as TWO kernels without float atomics (deterministic): bwd_dq (a workgroup owns queries, streams K|V tiles:
//             S^T, dP^T, dQ^T) and bwd_dkv (a workgroup owns keys, streams Q|dO tiles: S, dP, dV^T, dK^T) - 7 products of
//             M x M x 64 instead of 5, none of them through HBM.
//
// All products run on v_mfma_f32_32x32x2_f32 in the orientation that makes the softmax arithmetic per-lane: the owned index
// (query in fwd / bwd_dq, key in bwd_dkv) is the MFMA COLUMN, so a lane keeps its column's 16 tile entries in its accumulator
// registers and those registers are fed back UNMOVED as the B operand of the next product (step r pairs the tile rows the two
// half-waves hold in register r).  Workgroup = 4 MFMA waves + 4 loader waves streaming 32-row tiles through two LDS stages, in
// one of two shapes (FtShape below) chosen so that every SIMD of the chip gets an MFMA wave at 8 as well as at 32 patches.
#include "common.h"

namespace dispu {

typedef float fa_f32x16 __attribute__((ext_vector_type(16)));

constexpr int FT_D = 64, FT_T = 32, FT_P = FT_D + 1;          // tile rows, padded pitch (conflict-free column reads)
constexpr float FT_LOG2E = 1.4426950408889634f;

__device__ __forceinline__ int ft_row(int r, int kh) { return (r & 3) + 8 * (r >> 2) + 4 * kh; }   // accumulator register -> tile row

// XCD-aware (cloud, block): workgroup ids go round-robin over the 8 XCDs; with a cloud count that is a multiple of 8 every XCD
// (= every L2) works on whole clouds, so a cloud's streamed tiles are fetched into one L2 only.
__device__ __forceinline__ void ft_block(int& cloud, int& blk) {
    cloud = blockIdx.y, blk = blockIdx.x;
    if ((gridDim.y & 7u) == 0) {
        const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y, slot = lin >> 3;
        cloud = (int)((lin & 7u) * (gridDim.y >> 3) + slot / gridDim.x);
        blk = (int)(slot % gridDim.x);
    }
}

// Workgroup = 4 MFMA waves + 4 loader waves, two shapes:
//   SPLIT = false  the four MFMA waves own 4 x 32 = 128 columns and all read the ONE streamed tile of a stage (most reuse of a
//                  streamed tile; needs b * rows / 128 >= 256 workgroups to fill the chip: 32 training patches);
//   SPLIT = true   the four MFMA waves own the SAME 32 columns and each takes one of the FOUR tiles of a stage (the streamed
//                  dimension is split four ways inside the workgroup); their partial results are combined through LDS at the end
//                  in a fixed order.  8 training patches: 256 workgroups x 4 MFMA waves = one per SIMD of the chip (with one MFMA
//                  wave per workgroup - 256 waves - three of the four matrix pipes of every CU idle: measured 2.10 vs 1.91 ms/step).
template <bool SPLIT> struct FtShape {
    static constexpr int TS = SPLIT ? 4 : 1;               // tiles per stage
    static constexpr int COLS = SPLIT ? 32 : 128;          // owned columns per workgroup
};

// The loader waves' share of a stage: TS tile pairs [32][64] of A and B (256 threads, 2 TS float4 of each per thread).
// PADB: B is stored with the padded pitch too (scalar stores); otherwise as [32][64] float4 rows.
template <int TS, bool PADB>
struct FtLoader {
    static constexpr int F = 2 * TS;
    static constexpr int TILE = FT_T * FT_P + (PADB ? FT_T * FT_P : FT_T * FT_D);     // floats per tile pair
    float4 pa[F], pb[F];
    __device__ __forceinline__ void load(const float* __restrict__ A, long lda, const float* __restrict__ B, long ldb, int row0, int nrows,
                                         int tid) {
#pragma unroll
        for (int it = 0; it < F; ++it) {
            const int e = tid + it * 256, c4 = e & 15;
            const int row = min(row0 + (e >> 4), nrows - 1);             // rows past the end (last, partial stage) feed unused slots
            pa[it] = *reinterpret_cast<const float4*>(A + (size_t)row * lda + c4 * 4);
            pb[it] = *reinterpret_cast<const float4*>(B + (size_t)row * ldb + c4 * 4);
        }
    }
    // stage layout: slot-major, per slot [A tile | B tile | EXTRA floats]
    template <int SLOT_STRIDE>
    __device__ __forceinline__ void store(float* st, int tid) const {
#pragma unroll
        for (int it = 0; it < F; ++it) {
            const int e = tid + it * 256, row = e >> 4, c4 = e & 15;
            float* base = st + (row >> 5) * SLOT_STRIDE;
            float* a = base + (row & 31) * FT_P + c4 * 4;
            a[0] = pa[it].x; a[1] = pa[it].y; a[2] = pa[it].z; a[3] = pa[it].w;
            if constexpr (PADB) {
                float* b = base + FT_T * FT_P + (row & 31) * FT_P + c4 * 4;
                b[0] = pb[it].x; b[1] = pb[it].y; b[2] = pb[it].z; b[3] = pb[it].w;
            } else {
                *reinterpret_cast<float4*>(base + FT_T * FT_P + (row & 31) * FT_D + c4 * 4) = pb[it];
            }
        }
    }
};

// SPLIT epilogue: the four MFMA waves hold partial sums of the same [64 d][32 columns] tile in acc[2] (register (c, 4 g + u) of
// lane (li, kh) = element d = 32 c + 8 g + 4 kh + u of column li).  All four dump theirs to LDS, then wave w adds the four copies
// of its quarter (c = w >> 1, g in {2 (w & 1), 2 (w & 1) + 1}) in wave order 0, 1, 2, 3 (fixed: deterministic), scales by
// mul * colscale[...] and stores two float4 per lane.  `wsc` (optional, fwd only): per-wave per-column weights in LDS.
__device__ __forceinline__ void ft_split_reduce_store(float* red, const fa_f32x16 (&acc)[2], int wave, int lane, float mul,
                                                      const float* wsc, float* __restrict__ dst_row) {
    const int li = lane & 31, kh = lane >> 5;
    if (wave < 4) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(wave * 32 + c * 16 + r) * 64 + lane] = acc[c][r];
    }
    __syncthreads();
    if (wave < 4) {
        const int c = wave >> 1;
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
            const int g = 2 * (wave & 1) + gg;
            float o[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const float v = red[(w * 32 + c * 16 + 4 * g + u) * 64 + lane];
                    t += wsc ? v * wsc[w * 32 + li] : v;
                }
                o[u] = t * mul;
            }
            *reinterpret_cast<float4*>(dst_row + c * 32 + 8 * g + 4 * kh) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
    __syncthreads();
}

constexpr int FT_RED = 4 * 32 * 64;                          // floats of the SPLIT epilogue buffer (32 KB), aliases the stages

// ------------------------------------------------------------------------------------------------------------ forward
// O[q][:] = softmax_k(scale Q[q].K[k]) . V,  lse2[q] (log2 domain).  Tile pair: K tile [32][65] + V tile [32][64].
template <bool SPLIT>
__global__ __launch_bounds__(512) void fa_train_fwd_kernel(int m, int nk, const float* __restrict__ Q, long ldq,
                                                           const float* __restrict__ K, long ldk, const float* __restrict__ V,
                                                           long ldv, float scale, float* __restrict__ O, long ldo,
                                                           float* __restrict__ lse2) {
    constexpr int TS = FtShape<SPLIT>::TS;
    using Ld = FtLoader<TS, false>;
    constexpr int SLOT = Ld::TILE, STAGE = TS * SLOT;
    extern __shared__ __attribute__((aligned(16))) float lds[];       // 2 stages (>= FT_RED + 2 * 128 floats when SPLIT)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int cloud, qblk;
    ft_block(cloud, qblk);
    const float* __restrict__ kb = K + (size_t)cloud * nk * ldk;
    const float* __restrict__ vb = V + (size_t)cloud * nk * ldv;
    const int ntile = nk / FT_T, nstage = (ntile + TS - 1) / TS;
    const int li = lane & 31, kh = lane >> 5;
    const size_t qrow = (size_t)cloud * m + qblk * FtShape<SPLIT>::COLS + (SPLIT ? 0 : (wave & 3) * 32) + li;
    fa_f32x16 oacc[2];
    float mrun = -__builtin_inff(), lsum = 0.f;

    if (wave >= 4) {
        const int tid = threadIdx.x - 256;
        Ld ld;
        ld.load(kb, ldk, vb, ldv, 0, nk, tid);
        ld.template store<SLOT>(lds, tid);
        if (nstage > 1) ld.load(kb, ldk, vb, ldv, TS * FT_T, nk, tid);
        __syncthreads();
        for (int t = 0; t < nstage; ++t) {
            if (t + 1 < nstage) {
                ld.template store<SLOT>(lds + ((t + 1) & 1) * STAGE, tid);
                if (t + 2 < nstage) ld.load(kb, ldk, vb, ldv, (t + 2) * TS * FT_T, nk, tid);
            }
            __syncthreads();
        }
    } else {
        const float* __restrict__ qp = Q + qrow * ldq;
        const float scale2 = scale * FT_LOG2E;
        float qf[32];
#pragma unroll
        for (int s4 = 0; s4 < 16; ++s4) {
            const float4 v = *reinterpret_cast<const float4*>(qp + s4 * 4);
            qf[2 * s4] = (kh ? v.y : v.x) * scale2;
            qf[2 * s4 + 1] = (kh ? v.w : v.z) * scale2;
        }
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[c][r] = 0.f;

        __syncthreads();
        for (int t = 0; t < nstage; ++t) {
            if (!SPLIT || t * TS + wave < ntile) {
                const float* Kt = lds + (t & 1) * STAGE + (SPLIT ? wave * SLOT : 0);
                const float* Vt = Kt + FT_T * FT_P;
                fa_f32x16 sacc;
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
                for (int s = 0; s < 32; ++s) sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(Kt[li * FT_P + 2 * s + kh], qf[s], sacc, 0, 0, 0);
                float mx = fmaxf(fmaxf(sacc[0], sacc[1]), sacc[2]);
#pragma unroll
                for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, sacc[r]), sacc[r + 1]);
                mx = fmaxf(mx, sacc[15]);
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float mnew = fmaxf(mrun, mx);
                const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);
                float rs = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) { sacc[r] = __builtin_amdgcn_exp2f(sacc[r] - mnew); rs += sacc[r]; }
                lsum = lsum * alpha + rs;
                mrun = mnew;
                if (__any(alpha != 1.0f)) {
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int r = 0; r < 16; ++r) oacc[c][r] = oacc[c][r] * alpha;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = ft_row(r, kh);
#pragma unroll
                    for (int c = 0; c < 2; ++c)
                        oacc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vt[key * FT_D + c * 32 + li], sacc[r], oacc[c], 0, 0, 0);
                }
            }
            __syncthreads();
        }
    }
    if constexpr (!SPLIT) {
        if (wave >= 4) return;
        const float ltot = lsum + __shfl_xor(lsum, 32, 64);
        const float inv = 1.0f / ltot;
        float* __restrict__ op = O + qrow * ldo;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 v;
                v.x = oacc[c][4 * g + 0] * inv; v.y = oacc[c][4 * g + 1] * inv; v.z = oacc[c][4 * g + 2] * inv; v.w = oacc[c][4 * g + 3] * inv;
                *reinterpret_cast<float4*>(op + c * 32 + 8 * g + 4 * kh) = v;
            }
        if (kh == 0) lse2[qrow] = mrun + __builtin_amdgcn_logf(ltot);      // v_log_f32 = log2
    } else {
        // combine the four waves' (running maximum, sum, O^T): weights 2^(m_w - M) / l with M = max_w m_w, l = sum_w l_w 2^(m_w - M)
        float* red = lds;
        float* wm = lds + FT_RED;                // [4][32] maxima, then [4][32] weights
        float* wl = wm + 128;
        if (wave < 4) {
            const float ltot = lsum + __shfl_xor(lsum, 32, 64);
            if (kh == 0) { wm[wave * 32 + li] = mrun; wl[wave * 32 + li] = ltot; }
        }
        __syncthreads();
        float inv = 0.f;
        if (wave < 4) {
            const float m0 = wm[li], m1 = wm[32 + li], m2 = wm[64 + li], m3 = wm[96 + li];
            const float M = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
            const float a0 = __builtin_amdgcn_exp2f(m0 - M), a1 = __builtin_amdgcn_exp2f(m1 - M), a2 = __builtin_amdgcn_exp2f(m2 - M),
                        a3 = __builtin_amdgcn_exp2f(m3 - M);
            const float l = ((wl[li] * a0 + wl[32 + li] * a1) + wl[64 + li] * a2) + wl[96 + li] * a3;
            inv = 1.0f / l;
            if (wave == 0 && kh == 0) lse2[qrow] = M + __builtin_amdgcn_logf(l);
            // every wave computed the same weights; wave w publishes its own (after all four have READ the maxima: the barrier below)
            mrun = (wave == 0) ? a0 : (wave == 1) ? a1 : (wave == 2) ? a2 : a3;
        }
        __syncthreads();
        if (wave < 4 && kh == 0) wm[wave * 32 + li] = mrun;
        // (ft_split_reduce_store's first barrier orders these writes before the weighted sums)
        ft_split_reduce_store(red, oacc, wave, lane, inv, wm, O + qrow * ldo);
    }
}

// ---------------------------------------------------------------------------------------------------------- backward: dQ
// A workgroup owns queries (columns) and streams the cloud's K|V tiles (both with the padded pitch):
//   S^T[k][q] = K.Q^T (log2 domain), dP^T[k][q] = V.dO^T, dS^T = 2^(S^T - lse2) o (dP^T - D), dQ^T[d][q] += K^T[d][k] dS^T[k][q].
// Also writes D[q] = dO[q].O[q] for the dK/dV kernel.
template <bool SPLIT>
__global__ __launch_bounds__(512) void fa_train_bwd_dq_kernel(int m, int nk, const float* __restrict__ Q, long ldq,
                                                              const float* __restrict__ K, long ldk, const float* __restrict__ V,
                                                              long ldv, float scale, const float* __restrict__ O, long ldo,
                                                              const float* __restrict__ lse2, const float* __restrict__ dO, long lddo,
                                                              float* __restrict__ dQ, long lddq, float* __restrict__ Dvec) {
    constexpr int TS = FtShape<SPLIT>::TS;
    using Ld = FtLoader<TS, true>;
    constexpr int SLOT = Ld::TILE, STAGE = TS * SLOT;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int cloud, qblk;
    ft_block(cloud, qblk);
    const float* __restrict__ kb = K + (size_t)cloud * nk * ldk;
    const float* __restrict__ vb = V + (size_t)cloud * nk * ldv;
    const int ntile = nk / FT_T, nstage = (ntile + TS - 1) / TS;
    const int li = lane & 31, kh = lane >> 5;
    const size_t qrow = (size_t)cloud * m + qblk * FtShape<SPLIT>::COLS + (SPLIT ? 0 : (wave & 3) * 32) + li;
    fa_f32x16 qacc[2];

    if (wave >= 4) {
        const int tid = threadIdx.x - 256;
        Ld ld;
        ld.load(kb, ldk, vb, ldv, 0, nk, tid);
        ld.template store<SLOT>(lds, tid);
        if (nstage > 1) ld.load(kb, ldk, vb, ldv, TS * FT_T, nk, tid);
        __syncthreads();
        for (int t = 0; t < nstage; ++t) {
            if (t + 1 < nstage) {
                ld.template store<SLOT>(lds + ((t + 1) & 1) * STAGE, tid);
                if (t + 2 < nstage) ld.load(kb, ldk, vb, ldv, (t + 2) * TS * FT_T, nk, tid);
            }
            __syncthreads();
        }
    } else {
        const float* __restrict__ qp = Q + qrow * ldq;
        const float* __restrict__ dop = dO + qrow * lddo;
        const float* __restrict__ op = O + qrow * ldo;
        const float scale2 = scale * FT_LOG2E;
        float qf[32], dof[32];
        float dsum = 0.f;
#pragma unroll
        for (int s4 = 0; s4 < 16; ++s4) {
            const float4 v = *reinterpret_cast<const float4*>(qp + s4 * 4);
            qf[2 * s4] = (kh ? v.y : v.x) * scale2;
            qf[2 * s4 + 1] = (kh ? v.w : v.z) * scale2;
            const float4 g = *reinterpret_cast<const float4*>(dop + s4 * 4);
            const float4 o = *reinterpret_cast<const float4*>(op + s4 * 4);
            dof[2 * s4] = kh ? g.y : g.x;
            dof[2 * s4 + 1] = kh ? g.w : g.z;
            dsum += dof[2 * s4] * (kh ? o.y : o.x) + dof[2 * s4 + 1] * (kh ? o.w : o.z);
        }
        const float Dq = dsum + __shfl_xor(dsum, 32, 64);
        if (kh == 0 && (!SPLIT || wave == 0)) Dvec[qrow] = Dq;
        const float L = lse2[qrow];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) qacc[c][r] = 0.f;

        __syncthreads();
        for (int t = 0; t < nstage; ++t) {
            if (!SPLIT || t * TS + wave < ntile) {
                const float* Kt = lds + (t & 1) * STAGE + (SPLIT ? wave * SLOT : 0);
                const float* Vt = Kt + FT_T * FT_P;
                fa_f32x16 sacc, pacc;
#pragma unroll
                for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; pacc[r] = 0.f; }
#pragma unroll
                for (int s = 0; s < 32; ++s) {
                    sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(Kt[li * FT_P + 2 * s + kh], qf[s], sacc, 0, 0, 0);
                    pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(Vt[li * FT_P + 2 * s + kh], dof[s], pacc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[r] = __builtin_amdgcn_exp2f(sacc[r] - L) * (pacc[r] - Dq);      // dS^T
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = ft_row(r, kh);
#pragma unroll
                    for (int c = 0; c < 2; ++c)
                        qacc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(Kt[key * FT_P + c * 32 + li], sacc[r], qacc[c], 0, 0, 0);
                }
            }
            __syncthreads();
        }
    }
    float* __restrict__ dqp = dQ + qrow * lddq;
    if constexpr (!SPLIT) {
        if (wave >= 4) return;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 v;
                v.x = qacc[c][4 * g + 0] * scale; v.y = qacc[c][4 * g + 1] * scale; v.z = qacc[c][4 * g + 2] * scale; v.w = qacc[c][4 * g + 3] * scale;
                *reinterpret_cast<float4*>(dqp + c * 32 + 8 * g + 4 * kh) = v;
            }
    } else {
        ft_split_reduce_store(lds, qacc, wave, lane, scale, nullptr, dqp);
    }
}

// ------------------------------------------------------------------------------------------------------- backward: dK, dV
// A workgroup owns keys (columns) and streams the cloud's Q|dO tiles (+ lse2, D of the tile's 32 queries):
//   S[q][k] = Q.K^T (log2 domain), dP[q][k] = dO.V^T, P = 2^(S - lse2[q]), dS = P o (dP - D[q]),
//   dV^T[d][k] += dO^T[d][q] P[q][k],   dK^T[d][k] += Q^T[d][q] dS[q][k].
template <bool SPLIT>
__global__ __launch_bounds__(512) void fa_train_bwd_dkv_kernel(int m, int nk, const float* __restrict__ Q, long ldq,
                                                               const float* __restrict__ K, long ldk, const float* __restrict__ V,
                                                               long ldv, float scale, const float* __restrict__ lse2,
                                                               const float* __restrict__ Dvec, const float* __restrict__ dO, long lddo,
                                                               float* __restrict__ dK, long lddk, float* __restrict__ dV, long lddv) {
    constexpr int TS = FtShape<SPLIT>::TS;
    using Ld = FtLoader<TS, true>;
    constexpr int SLOT = Ld::TILE + 2 * FT_T, STAGE = TS * SLOT;      // per slot: Q tile, dO tile, lse2[32], D[32]
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int cloud, kblk;
    ft_block(cloud, kblk);
    const float* __restrict__ qb = Q + (size_t)cloud * m * ldq;
    const float* __restrict__ gb = dO + (size_t)cloud * m * lddo;
    const float* __restrict__ lb = lse2 + (size_t)cloud * m;
    const float* __restrict__ db = Dvec + (size_t)cloud * m;
    const int ntile = m / FT_T, nstage = (ntile + TS - 1) / TS;
    const int li = lane & 31, kh = lane >> 5;
    const size_t krow = (size_t)cloud * nk + kblk * FtShape<SPLIT>::COLS + (SPLIT ? 0 : (wave & 3) * 32) + li;
    fa_f32x16 kacc[2], vacc[2];

    if (wave >= 4) {
        const int tid = threadIdx.x - 256;
        Ld ld;
        float st_l = 0.f;                                       // thread (slot, j): j < 32 lse2 of the slot's tile, j >= 32 its D
        const int sslot = tid >> 6, sj = tid & 63;
        auto load_stat = [&](int q0) {
            if (sslot < TS) {
                const int q = min(q0 + sslot * FT_T + (sj & 31), m - 1);
                st_l = (sj < 32) ? lb[q] : db[q];
            }
        };
        auto store_stat = [&](float* st) { if (sslot < TS) st[sslot * SLOT + Ld::TILE + sj] = st_l; };
        ld.load(qb, ldq, gb, lddo, 0, m, tid);
        load_stat(0);
        ld.template store<SLOT>(lds, tid);
        store_stat(lds);
        if (nstage > 1) { ld.load(qb, ldq, gb, lddo, TS * FT_T, m, tid); load_stat(TS * FT_T); }
        __syncthreads();
        for (int t = 0; t < nstage; ++t) {
            if (t + 1 < nstage) {
                float* st = lds + ((t + 1) & 1) * STAGE;
                ld.template store<SLOT>(st, tid);
                store_stat(st);
                if (t + 2 < nstage) { ld.load(qb, ldq, gb, lddo, (t + 2) * TS * FT_T, m, tid); load_stat((t + 2) * TS * FT_T); }
            }
            __syncthreads();
        }
    } else {
        const float* __restrict__ kp = K + krow * ldk;
        const float* __restrict__ vp = V + krow * ldv;
        const float scale2 = scale * FT_LOG2E;
        float kf[32], vf[32];
#pragma unroll
        for (int s4 = 0; s4 < 16; ++s4) {
            const float4 a = *reinterpret_cast<const float4*>(kp + s4 * 4);
            const float4 b = *reinterpret_cast<const float4*>(vp + s4 * 4);
            kf[2 * s4] = (kh ? a.y : a.x) * scale2;
            kf[2 * s4 + 1] = (kh ? a.w : a.z) * scale2;
            vf[2 * s4] = kh ? b.y : b.x;
            vf[2 * s4 + 1] = kh ? b.w : b.z;
        }
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) { kacc[c][r] = 0.f; vacc[c][r] = 0.f; }

        __syncthreads();
        for (int t = 0; t < nstage; ++t) {
            if (!SPLIT || t * TS + wave < ntile) {
                const float* Qt = lds + (t & 1) * STAGE + (SPLIT ? wave * SLOT : 0);
                const float* Gt = Qt + FT_T * FT_P;
                const float* Lt = Gt + FT_T * FT_P;
                const float* Dt = Lt + FT_T;
                fa_f32x16 sacc, pacc;
#pragma unroll
                for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; pacc[r] = 0.f; }
#pragma unroll
                for (int s = 0; s < 32; ++s) {
                    sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(Qt[li * FT_P + 2 * s + kh], kf[s], sacc, 0, 0, 0);
                    pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(Gt[li * FT_P + 2 * s + kh], vf[s], pacc, 0, 0, 0);
                }
                // register r of this lane = tile row (query) ft_row(r, kh): rows 8 g + 4 kh .. + 3 are one float4
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 l4 = *reinterpret_cast<const float4*>(Lt + 8 * g + 4 * kh);
                    const float4 d4 = *reinterpret_cast<const float4*>(Dt + 8 * g + 4 * kh);
                    const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, dq[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float p = __builtin_amdgcn_exp2f(sacc[4 * g + u] - lq[u]);
                        sacc[4 * g + u] = p;                                   // P
                        pacc[4 * g + u] = p * (pacc[4 * g + u] - dq[u]);       // dS
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int q = ft_row(r, kh);
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        vacc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(Gt[q * FT_P + c * 32 + li], sacc[r], vacc[c], 0, 0, 0);
                        kacc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(Qt[q * FT_P + c * 32 + li], pacc[r], kacc[c], 0, 0, 0);
                    }
                }
            }
            __syncthreads();
        }
    }
    float* __restrict__ dkp = dK + krow * lddk;
    float* __restrict__ dvp = dV + krow * lddv;
    if constexpr (!SPLIT) {
        if (wave >= 4) return;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 a, b;
                a.x = kacc[c][4 * g + 0] * scale; a.y = kacc[c][4 * g + 1] * scale; a.z = kacc[c][4 * g + 2] * scale; a.w = kacc[c][4 * g + 3] * scale;
                b.x = vacc[c][4 * g + 0]; b.y = vacc[c][4 * g + 1]; b.z = vacc[c][4 * g + 2]; b.w = vacc[c][4 * g + 3];
                *reinterpret_cast<float4*>(dkp + c * 32 + 8 * g + 4 * kh) = a;
                *reinterpret_cast<float4*>(dvp + c * 32 + 8 * g + 4 * kh) = b;
            }
    } else {
        ft_split_reduce_store(lds, kacc, wave, lane, scale, nullptr, dkp);
        ft_split_reduce_store(lds, vacc, wave, lane, 1.0f, nullptr, dvp);
    }
}

// SPLIT when 128-column workgroups would leave CUs idle
static bool ft_split(int b, int rows) { return (rows % 128) != 0 || (long)b * (rows / 128) < 256; }

template <typename KernelT>
static int ft_lds_optin(KernelT kern, size_t bytes, DevOnce& once) {
    if (bytes > 64 * 1024 && once.needed()) {
        DISPU_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        once.done();
    }
    return 0;
}

static bool ft_bad(const void* p, long ld) { return !p || (ld & 3) || (((uintptr_t)p) & 15); }

}  // namespace dispu

using namespace dispu;

// Forward of the non-local cell's attention for training: O[b*m, 64] = softmax(scale Q K^T) V per cloud and
// lse2[b*m] = log2 sum_k 2^(scale log2(e) Q.K) (the row statistic dispu_attention_bwd recomputes P from).
// d == 64, m % 32 == 0, nk % 32 == 0, rows 16-byte aligned.  Replaces tf.matmul / tf.nn.softmax / tf.matmul of ops.py:326-339.
DISPU_EXPORT int dispu_attention_fwd_lse(int b, int m, int nk, int d, const float* Q, long ldq, const float* K, long ldk, const float* V,
                                         long ldv, float scale, float* O, long ldo, float* lse2, void* stream) {
    if (b < 0 || m <= 0 || nk <= 0 || d != FT_D || (m % 32) || (nk % 32) || ft_bad(Q, ldq) || ft_bad(K, ldk) || ft_bad(V, ldv) ||
        ft_bad(O, ldo) || !lse2)
        return (int)hipErrorInvalidValue;
    if (b == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    constexpr size_t tile = (size_t)FtLoader<1, false>::TILE * sizeof(float);
    if (ft_split(b, m)) {
        constexpr size_t bytes = 2 * 4 * tile;
        static_assert(bytes >= (FT_RED + 256) * sizeof(float), "SPLIT epilogue buffer does not fit the stages");
        static DevOnce once;
        if (int e = ft_lds_optin(fa_train_fwd_kernel<true>, bytes, once)) return e;
        hipLaunchKernelGGL((fa_train_fwd_kernel<true>), dim3(m / 32, b), dim3(512), bytes, s, m, nk, Q, ldq, K, ldk, V, ldv, scale, O, ldo, lse2);
    } else {
        hipLaunchKernelGGL((fa_train_fwd_kernel<false>), dim3(m / 128, b), dim3(512), 2 * tile, s, m, nk, Q, ldq, K, ldk, V, ldv, scale, O, ldo, lse2);
    }
    return (int)hipGetLastError();
}

// Backward of the same attention (TF1 autodiff of ops.py:326-339): given dO, writes dQ [b*m, 64], dK, dV [b*nk, 64] (overwritten,
// not accumulated).  P is recomputed from Q, K, lse2; `dvec` [b*m] floats of scratch receives D = rowsum(dO o O).  Two launches
// on `stream` (dQ, then dK|dV), deterministic: no float atomics.  Same shape rules as dispu_attention_fwd_lse.
DISPU_EXPORT int dispu_attention_bwd(int b, int m, int nk, int d, const float* Q, long ldq, const float* K, long ldk, const float* V,
                                     long ldv, float scale, const float* O, long ldo, const float* lse2, const float* dO, long lddo,
                                     float* dQ, long lddq, float* dK, long lddk, float* dV, long lddv, float* dvec, void* stream) {
    if (b < 0 || m <= 0 || nk <= 0 || d != FT_D || (m % 32) || (nk % 32) || ft_bad(Q, ldq) || ft_bad(K, ldk) || ft_bad(V, ldv) ||
        ft_bad(O, ldo) || ft_bad(dO, lddo) || ft_bad(dQ, lddq) || ft_bad(dK, lddk) || ft_bad(dV, lddv) || !lse2 || !dvec)
        return (int)hipErrorInvalidValue;
    if (b == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    {
        constexpr size_t tile = (size_t)FtLoader<1, true>::TILE * sizeof(float);
        if (ft_split(b, m)) {
            constexpr size_t bytes = 2 * 4 * tile;
            static_assert(bytes >= FT_RED * sizeof(float), "SPLIT epilogue buffer does not fit the stages");
            static DevOnce once;
            if (int e = ft_lds_optin(fa_train_bwd_dq_kernel<true>, bytes, once)) return e;
            hipLaunchKernelGGL((fa_train_bwd_dq_kernel<true>), dim3(m / 32, b), dim3(512), bytes, s, m, nk, Q, ldq, K, ldk, V, ldv, scale, O, ldo,
                               lse2, dO, lddo, dQ, lddq, dvec);
        } else {
            hipLaunchKernelGGL((fa_train_bwd_dq_kernel<false>), dim3(m / 128, b), dim3(512), 2 * tile, s, m, nk, Q, ldq, K, ldk, V, ldv, scale, O,
                               ldo, lse2, dO, lddo, dQ, lddq, dvec);
        }
        DISPU_CHECK_LAUNCH();
    }
    {
        constexpr size_t tile = (size_t)(FtLoader<1, true>::TILE + 2 * FT_T) * sizeof(float);
        if (ft_split(b, nk)) {
            constexpr size_t bytes = 2 * 4 * tile;
            static DevOnce once;
            if (int e = ft_lds_optin(fa_train_bwd_dkv_kernel<true>, bytes, once)) return e;
            hipLaunchKernelGGL((fa_train_bwd_dkv_kernel<true>), dim3(nk / 32, b), dim3(512), bytes, s, m, nk, Q, ldq, K, ldk, V, ldv, scale, lse2,
                               dvec, dO, lddo, dK, lddk, dV, lddv);
        } else {
            hipLaunchKernelGGL((fa_train_bwd_dkv_kernel<false>), dim3(nk / 128, b), dim3(512), 2 * tile, s, m, nk, Q, ldq, K, ldk, V, ldv, scale,
                               lse2, dvec, dO, lddo, dK, lddk, dV, lddv);
        }
    }
    return (int)hipGetLastError();
}
