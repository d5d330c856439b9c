// Per-point dense layers (the reference's 1x1 conv1d/conv2d, Common/tf_util.py:52-185) and the
// batched matmuls of PointNonLocalCell (Common/ops.py:326,339) as ONE fp32 MFMA GEMM for gfx950:
//
//   Y[z][m, n] = res2 + res1 + act( (chain_k X[z][m, k] * W[z][k, n] + bias[n]) [* scale[n] + shift[n]] )
//
// * v_mfma_f32_32x32x2_f32: exact fp32, and bit-for-bit the ascending-k fmaf chain the oracle pins
//   (oracle/mlp_oracle.c) -- no split-K, no reassociation, so features feeding the k-NN stages are
//   bit-reproducible.
// * Operands are addressed as (pointer, row stride): inputs can be column slices of a wider
//   activation buffer and outputs land directly inside concatenation buffers (the reference
//   materialises every tf.concat).
// * TRANSB reads W as [n][k] (k contiguous): Q.K^T without a transpose pass.
// * Workgroup = WM x WN waves, each owning TM x TN 32x32 accumulator tiles.  Two LDS stages
//   (A[BK][BM+1] / B[BK][BN+4], conflict-free b32 fragment reads): while the MFMAs of stage t run, the same
//   waves park K-tile t+1 (already in registers) into the other stage and issue the global loads of tile t+2, so
//   there is ONE barrier per K-tile and no load/store-only phase.  Interior tiles take a predicate-free path.
//   Epilogue fuses bias, BatchNorm scale/shift, ReLU and two residual adds.
#include "common.h"

#include <type_traits>

#include <cstdio>
#include <cstdlib>

namespace dispu {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct LinearArgs {
    int M, K, N;
    const float* X; long ldx; long sx;   // activations [M,K], row stride, batch stride
    const float* W; long ldw; long sw;   // weights [K,N] (or [N,K] when TRANSB), row stride, batch stride
    const float* bias;                   // [N] or null
    float* Y; long ldy; long sy;
    const float* R1; long ldr1; long sr1;  // optional residuals added after the activation
    const float* R2; long ldr2; long sr2;
    int act;                             // 0 none, 1 relu
    const float* scale;                  // optional inference BatchNorm folded to v*scale[n] + shift[n], applied
    const float* shift;                  // between the bias add and the activation (tf_util.py:176-185 order)
    const float* Mk; long ldm; int mcols;  // optional ReLU-gradient mask, applied LAST: Y[m][n] = 0 where Mk[m][n] <= 0, for the
                                           // columns n < mcols (training: dX = dZ.W^T masked by the layer input's activation)
};

// DMA mode (interior 128 x 256 x 16 tiles, B not transposed): the loader waves issue global_load_lds_dwordx4 - global
// memory straight into LDS, no data registers, no ds_write - through NST = 4 stages.  A stage then holds
//   A: [BK/4][BM] float4 = X[row][4 kg .. 4 kg + 3]   (one DMA instruction = 64 rows x 16 bytes, lane-linear in LDS)
//   B: [BK][BN + 4] floats                            (one DMA instruction = one k row of 256 floats)
// and an MFMA lane gets its A operands of two k-steps from one ds_read_b128 (component fk / 2 + fk).
template <int BM, int BN, int BK, bool TRANSB, bool EDGE>
struct LinearLds {
    // round 5: every interior, untransposed BK = 16 tile of 64 / 128 rows x 64 / 128 / 256 columns takes the DMA path (it was 128 x 256
    // only; the register-staged loaders of the smaller tiles cost ~30 instructions per slab next to MFMA waves that need 1024 - 2048
    // cycles for theirs: loader-bound by 2 - 3x, the "26 us whatever the tile" of round 3)
    static constexpr bool DMA = !EDGE && !TRANSB && BK == 16 && (BM == 64 || BM == 128) && (BN == 64 || BN == 128 || BN == 256);
    static constexpr int NST = DMA ? 4 : 2;
    static constexpr int LDA = DMA ? BM : BM + 1;   // register path: A tile stored k-major [BK][BM+1]: conflict-free b32 frag reads and writes
    // DMA: one instruction lands 1 KB = 256 / BN consecutive k rows of the B slab back to back; the +4 pad per row only exists for BN = 256
    static constexpr int LDB = TRANSB ? BN + 1 : (DMA && BN < 256) ? BN : BN + 4;
    static constexpr int STAGE = ((BK * (LDA + LDB) + 3) / 4) * 4;          // floats per stage (16-byte multiple)
    static constexpr size_t BYTES = (size_t)NST * STAGE * sizeof(float);
};

// BM x BN block tile, WM x WN MFMA waves + as many LOADER waves, BK k-slab per LDS stage.
//
// Wave specialisation.  v_mfma blocks the wave that issued it, and a wave that also has to wait for its global
// loads (s_waitcnt vmcnt), write them to LDS and issue the next loads cannot keep the matrix pipe fed: measured on
// the [32768 x 2048] x [2048 x 256] product (tools/micro/gemm_lab.hip), the single-role kernel spends ~2600 of every
// ~7300 cycles per slab in that refill phase and reaches 82 TFLOP/s, although the same MFMA loop alone sustains
// 135 TFLOP/s and the pipe itself 156.  So the roles are split: waves 0 .. WM*WN-1 only read operand fragments from
// LDS and issue MFMAs; waves WM*WN .. 2*WM*WN-1 only move tiles (global -> registers -> LDS, one tile in flight).
// One workgroup barrier per slab joins them (127 TFLOP/s on that product).  Each SIMD hosts one wave of each role.
// EDGE = false: every tile is interior and 16-byte aligned (M % BM == N % BN == K % BK == 0): no predicates at all.
#ifdef LIN_CLOCK
__device__ unsigned long long lin_clock_ticks[4];
#endif
template <int BM, int BN, int WM, int WN, int BK, bool TRANSB, bool EDGE, int EPI_T>
__global__ __launch_bounds__(64 * (WM * WN + (LinearLds<BM, BN, BK, TRANSB, EDGE>::DMA ? 4 : WM * WN))) void linear_mfma_kernel(LinearArgs a) {
#ifdef LIN_CLOCK
    const unsigned long long lc0 = __builtin_readcyclecounter();
    unsigned long long lc_wait = 0;
#endif
    using L = LinearLds<BM, BN, BK, TRANSB, EDGE>;
    constexpr bool DMA = L::DMA;
    constexpr int NT = 64 * WM * WN;                  // threads per role
    constexpr int LDA = L::LDA, LDB = L::LDB;
    constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
    constexpr int A_F4 = BM * BK / 4 / NT;            // float4 loads per loader thread for the A tile
    constexpr int B_F4 = BN * BK / 4 / NT;
    static_assert(A_F4 >= 1 && B_F4 >= 1 && TM >= 1 && TN >= 1, "tile too small for the workgroup");
    static_assert((BM * BK / 4) % NT == 0 && (BN * BK / 4) % NT == 0, "tile must split evenly over the threads");
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int z = blockIdx.z;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int M = a.M, K = a.K, N = a.N;
    const int ntile = (K + BK - 1) / BK;

    if (wave >= WM * WN) {
        // ------------------------------------------------------------------------------------ loader waves
        if constexpr (DMA) {
            // Per slab: A = [4 k-groups][BM rows] float4 -> 4 * BM / 64 instructions (64 rows x 16 bytes each), B = 16 k-rows of BN floats
            // -> 16 * BN / 256 instructions (1 KB each: 256 / BN consecutive k rows); dealt round-robin to the four loader waves, PER of them
            // each.  Slab t+3 is requested while slab t is computed; before the barrier that ends slab t, slab t+1 must have landed:
            // at most the 2 * PER newest DMA instructions of this wave may still be in flight.
            const int lw = __builtin_amdgcn_readfirstlane(wave - WM * WN);
            const float* __restrict__ X = a.X + (size_t)z * a.sx;
            const float* __restrict__ W = a.W + (size_t)z * a.sw;
            constexpr int NA = 4 * BM / 64, NB = 16 * BN / 256, PER = (NA + NB) / 4;
            static_assert((NA + NB) % 4 == 0, "DMA instructions must split evenly over the four loader waves");
            constexpr int RPI = 256 / BN;                                 // k rows per B instruction
            const long ldw = a.ldw, ldx = a.ldx;
            // instruction q = lw + 4 u (u < PER): q < NA -> A piece (k-group q / (BM / 64), row block q % (BM / 64)), else B piece q - NA
            const float* gsrc[PER];
            int ldst[PER];                                                // LDS float offset inside a stage
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int q = lw + 4 * u;
                if (q < NA) {
                    const int kg = q / (BM / 64), rb = q % (BM / 64);
                    gsrc[u] = X + (size_t)(m0 + rb * 64 + lane) * ldx + 4 * kg;
                    ldst[u] = (kg * BM + rb * 64) * 4;
                } else {
                    const int pb = q - NA, kr = pb * RPI + lane / (BN / 4);
                    gsrc[u] = W + (size_t)kr * ldw + n0 + 4 * (lane % (BN / 4));
                    ldst[u] = BK * LDA + (pb * RPI) * LDB;
                }
            }
            auto issue = [&](int t) {
                float* st = lds + (t & 3) * L::STAGE;
                const size_t ko = (size_t)t * BK;
#pragma unroll
                for (int u = 0; u < PER; ++u) {
                    const int q = lw + 4 * u;
                    const float* src = (q < NA) ? gsrc[u] + ko : gsrc[u] + ko * ldw;
                    __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(st + ldst[u]), 16, 0, 0);
                }
            };
            // s_waitcnt takes an immediate: wait until at most PER * newer DMA instructions are outstanding.  Raw s_barrier, not
            // __syncthreads(): the workgroup fence in front of it would drain every DMA in flight (vmcnt(0)).
            auto wait_newer = [&](int newer) {
                if (newer >= 2) __builtin_amdgcn_s_waitcnt(0x0F70 | (((2 * PER) & 0xF)) | ((((2 * PER) >> 4) & 0x3) << 14));
                else if (newer == 1) __builtin_amdgcn_s_waitcnt(0x0F70 | ((PER & 0xF)) | (((PER >> 4) & 0x3) << 14));
                else __builtin_amdgcn_s_waitcnt(0x0F70);
            };
            issue(0);
            if (ntile > 1) issue(1);
            if (ntile > 2) issue(2);
            wait_newer(min(2, ntile - 1));                       // slab 0 has landed
            asm volatile("s_barrier" ::: "memory");
            for (int t = 0; t < ntile; ++t) {
                if (t + 3 < ntile) issue(t + 3);                 // into the stage slab t-1 was read from
                wait_newer(max(0, min(2, ntile - 2 - t)));       // slab t+1 has landed
                asm volatile("s_barrier" ::: "memory");
            }
            return;
        }
        const int tid = threadIdx.x - NT;
        const float* __restrict__ X = a.X + (size_t)z * a.sx;
        const float* __restrict__ W = a.W + (size_t)z * a.sw;
        const long ldx = a.ldx, ldw = a.ldw;
        float4 pa[A_F4], pb[B_F4];
        const bool x_vec = ((ldx & 3) == 0) && ((((uintptr_t)X) & 15) == 0);
        const bool w_vec = ((ldw & 3) == 0) && ((((uintptr_t)W) & 15) == 0);

        // rows-of-k loader (A always; B when TRANSB): element (row, k) with k contiguous in memory
        auto load_rowsk = [&](const float* __restrict__ P, long ld, bool vec, int row_base, int rows, int k0, int it) -> float4 {
            const int idx = tid + it * NT;
            const int r = idx / (BK / 4), kq = idx % (BK / 4);
            const int row = row_base + r, k = k0 + kq * 4;
            if constexpr (!EDGE) return *reinterpret_cast<const float4*>(P + (size_t)row * ld + k);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < rows && k < K) {
                const float* p = P + (size_t)row * ld + k;
                if (vec && k + 3 < K) {
                    v = *reinterpret_cast<const float4*>(p);
                } else {
                    v.x = p[0];
                    if (k + 1 < K) v.y = p[1];
                    if (k + 2 < K) v.z = p[2];
                    if (k + 3 < K) v.w = p[3];
                }
            }
            return v;
        };
        auto store_rowsk = [&](float* S, int LD, float4 v, int it) {
            const int idx = tid + it * NT;
            const int r = idx / (BK / 4), kq = idx % (BK / 4);
            S[(kq * 4 + 0) * LD + r] = v.x;
            S[(kq * 4 + 1) * LD + r] = v.y;
            S[(kq * 4 + 2) * LD + r] = v.z;
            S[(kq * 4 + 3) * LD + r] = v.w;
        };
        // k-rows loader for B = W[k][n] (n contiguous)
        auto load_b = [&](int k0, int it) -> float4 {
            const int idx = tid + it * NT;
            const int kr = idx / (BN / 4), nq = idx % (BN / 4);
            const int k = k0 + kr, n = n0 + nq * 4;
            if constexpr (!EDGE) return *reinterpret_cast<const float4*>(W + (size_t)k * ldw + n);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < K && n < N) {
                const float* p = W + (size_t)k * ldw + n;
                if (w_vec && n + 3 < N) {
                    v = *reinterpret_cast<const float4*>(p);
                } else {
                    v.x = p[0];
                    if (n + 1 < N) v.y = p[1];
                    if (n + 2 < N) v.z = p[2];
                    if (n + 3 < N) v.w = p[3];
                }
            }
            return v;
        };
        auto store_b = [&](float* S, float4 v, int it) {
            const int idx = tid + it * NT;
            const int kr = idx / (BN / 4), nq = idx % (BN / 4);
            *reinterpret_cast<float4*>(&S[kr * LDB + nq * 4]) = v;
        };
        // two register sets: the tiles of slabs t+2 and t+3 are both in flight while slab t is computed (a slab lasts
        // ~2 us, about one loaded-HBM round trip: with a single set the MFMA waves waited ~450 cycles per slab at the
        // barrier for the loaders, who were waiting for their data)
        float4 pa1[A_F4], pb1[B_F4];
        auto load_tile = [&](int k0, float4 (&qa)[A_F4], float4 (&qb)[B_F4]) {
#pragma unroll
            for (int it = 0; it < A_F4; ++it) qa[it] = load_rowsk(X, ldx, x_vec, m0, M, k0, it);
#pragma unroll
            for (int it = 0; it < B_F4; ++it) qb[it] = TRANSB ? load_rowsk(W, ldw, w_vec, n0, N, k0, it) : load_b(k0, it);
        };
        auto store_tile = [&](int stage, const float4 (&qa)[A_F4], const float4 (&qb)[B_F4]) {
            float* As = lds + stage * L::STAGE;
            float* Bs = As + BK * LDA;
#pragma unroll
            for (int it = 0; it < A_F4; ++it) store_rowsk(As, LDA, qa[it], it);
#pragma unroll
            for (int it = 0; it < B_F4; ++it) {
                if constexpr (TRANSB) store_rowsk(Bs, LDB, qb[it], it);
                else store_b(Bs, qb[it], it);
            }
        };
        load_tile(0, pa, pb);
        store_tile(0, pa, pb);
        if (ntile > 1) load_tile(BK, pa1, pb1);          // odd tiles travel in set 1, even tiles in set 0
        if (ntile > 2) load_tile(2 * BK, pa, pb);
        __syncthreads();
        // slab t: park tile t+1 in the other stage (its last readers passed the barrier that ended slab t-1) and
        // request tile t+3, while the MFMA waves work through stage t
        for (int t = 0; t < ntile; t += 2) {
            if (t + 1 < ntile) {
                store_tile(1, pa1, pb1);
                if (t + 3 < ntile) load_tile((t + 3) * BK, pa1, pb1);
            }
            __syncthreads();
            if (t + 1 < ntile) {
                if (t + 2 < ntile) {
                    store_tile(0, pa, pb);
                    if (t + 4 < ntile) load_tile((t + 4) * BK, pa, pb);
                }
                __syncthreads();
            }
        }
        return;
    }

    // ------------------------------------------------------------------------------------------ MFMA waves
    const int wm = wave / WN, wn = wave % WN;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int fi = lane & 31, fk = lane >> 5;
    // CI (round 5), interleaved column ownership: the TN accumulator tiles of a lane cover the TN CONSECUTIVE columns TN fi .. TN fi + TN - 1
    // of the wave's column range (tile j <-> column TN fi + j) instead of fi, 32 + fi, ...  Which column an MFMA column stands for is
    // free, the arithmetic per output is untouched; but a lane's B operands of a k-step are now adjacent in the LDS slab -- ONE
    // ds_read_b128 (TN = 4) instead of two ds_read2_b32 -- and in the epilogue a lane owns TN adjacent outputs of a row: one 16-byte
    // store / bias / residual load where there were four 4-byte ones (128 -> 32 global stores per wave and 32 x 32 tile pair).
    constexpr bool CI = !EDGE && !TRANSB && (TN == 4 || TN == 2);
    auto rdB = [&](const float* Bs, int krow, float (&b)[TN]) {
        if constexpr (CI && TN == 4) {
            const float4 t4 = *reinterpret_cast<const float4*>(&Bs[krow * LDB + wn * (TN * 32) + 4 * fi]);
            b[0] = t4.x; b[1] = t4.y; b[2] = t4.z; b[3] = t4.w;
        } else if constexpr (CI && TN == 2) {
            const float2 t2 = *reinterpret_cast<const float2*>(&Bs[krow * LDB + wn * (TN * 32) + 2 * fi]);
            b[0] = t2.x; b[1] = t2.y;
        } else {
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bs[krow * LDB + wn * (TN * 32) + j * 32 + fi];
        }
    };
    __syncthreads();
    // Fragment registers are double-buffered by hand: the ds_reads of k-step s+1 are issued BEFORE the MFMAs of step s.
    // Left to itself the compiler places a step's reads right in front of the MFMAs that need them, behind the previous
    // step's last MFMA: the wave then sits in s_waitcnt for an LDS round trip (~120 cycles) while the matrix pipe drains
    // after 64 - eight times per slab (measured: 4860 cycles per 64-MFMA slab instead of 4096).
    if constexpr (DMA) {
        // unrolled by four = one trip round the stage ring: the stage base of every slab is a compile-time offset, so the per-slab LDS
        // address arithmetic (6 - 8 VALU instructions between the MFMAs, ~25 cycles each on an MFMA wave: tools/micro/chain_lab.hip)
        // becomes immediates
        auto slab = [&](auto stage_c) {
            constexpr int SG = decltype(stage_c)::value;
            const float* st = lds + SG * L::STAGE;
            const float* Bs = st + BK * LDA;
            // operands of k-step s+1 are requested before the MFMAs of step s (see the register path below).  A: lane
            // (row, fk) reads the dwords fk and fk + 2 of its row's float4 (one ds_read2_b32): X[row][4 kg + fk] for the
            // first k-step of the group and X[row][4 kg + 2 + fk] for the second - no select needed.
            const float* Asf = st + fk;
            float a2[2][TM][2];
            float bf[2][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const float* pr = Asf + (wm * (TM * 32) + i * 32 + fi) * 4;
                a2[0][i][0] = pr[0]; a2[0][i][1] = pr[2];
            }
            rdB(Bs, fk, bf[0]);
#pragma unroll
            for (int s2 = 0; s2 < BK / 2; ++s2) {
                const int kg = s2 >> 1, u = s2 & 1, cur = s2 & 1, nxt = cur ^ 1;
                if (s2 + 1 < BK / 2) {
                    const int kk = 2 * (s2 + 1);
                    if (u == 1) {
#pragma unroll
                        for (int i = 0; i < TM; ++i) {
                            const float* pr = Asf + ((kg + 1) * BM + wm * (TM * 32) + i * 32 + fi) * 4;
                            a2[(kg + 1) & 1][i][0] = pr[0]; a2[(kg + 1) & 1][i][1] = pr[2];
                        }
                    }
                    rdB(Bs, kk + fk, bf[nxt]);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[kg & 1][i][u], bf[cur][j], acc[i][j], 0, 0, 0);
                constexpr int NB = CI ? 1 : TN / 2;                                          // B reads per k-step: one b128 / one ds_read2_b32 per pair
                if (u == 1) __builtin_amdgcn_sched_group_barrier(0x100, TM + NB, 0);         // + ds_read2_b32: one per row tile
                else __builtin_amdgcn_sched_group_barrier(0x100, NB, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
            }
#ifdef LIN_CLOCK
            const unsigned long long lb0 = __builtin_readcyclecounter();
            __syncthreads();
            lc_wait += __builtin_readcyclecounter() - lb0;
#else
            __syncthreads();
#endif
        };
        for (int t = 0; t < ntile; t += 4) {                         // (hand-unrolled: `#pragma unroll` is refused for this loop)
            slab(std::integral_constant<int, 0>{});
            if (t + 1 < ntile) slab(std::integral_constant<int, 1>{});
            if (t + 2 < ntile) slab(std::integral_constant<int, 2>{});
            if (t + 3 < ntile) slab(std::integral_constant<int, 3>{});
        }
    } else
    for (int t = 0; t < ntile; ++t) {
        const float* As = lds + (t & 1) * L::STAGE;
        const float* Bs = As + BK * LDA;
        float af[2][TM], bf[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[0][i] = As[fk * LDA + wm * (TM * 32) + i * 32 + fi];
        rdB(Bs, fk, bf[0]);
#pragma unroll
        for (int s2 = 0; s2 < BK / 2; ++s2) {
            const int cur = s2 & 1, nxt = cur ^ 1;
            if (s2 + 1 < BK / 2) {
                const int kk = 2 * (s2 + 1);
#pragma unroll
                for (int i = 0; i < TM; ++i) af[nxt][i] = As[(kk + fk) * LDA + wm * (TM * 32) + i * 32 + fi];
                rdB(Bs, kk + fk, bf[nxt]);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i], bf[cur][j], acc[i][j], 0, 0, 0);
            // pin: this step's DS reads (the next step's operands) first, then its TM*TN MFMAs
            __builtin_amdgcn_sched_group_barrier(0x100, TM + (CI ? 1 : TN), 0);
            __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
        }
#ifdef LIN_CLOCK
        const unsigned long long lb0 = __builtin_readcyclecounter();
        __syncthreads();
        lc_wait += __builtin_readcyclecounter() - lb0;
#else
        __syncthreads();
#endif
    }

#ifdef LIN_CLOCK
    if (blockIdx.x == 0 && blockIdx.y == 5 && threadIdx.x == 0) { lin_clock_ticks[0] = __builtin_readcyclecounter() - lc0; lin_clock_ticks[1] = ntile; lin_clock_ticks[2] = lc_wait; }
#endif
    // epilogue: C/D layout of the 32x32 tile: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
    // What the epilogue does is a TEMPLATE choice (EPI), not a per-element test: an earlier version tested
    // bias / scale / act / R1 / R2 inside the 128-element unrolled store loop, which the compiler unswitched into
    // ~5000 instructions with spills (9 us per launch).  EPI 0: bias + activation; 1: + BatchNorm fold;
    // 4: BatchNorm fold and whatever residuals are given (uniform branches around the residual loads: the compiler
    // keeps those loads in program order, whereas unconditional ones were all hoisted and spilled).
    // EPI_T 6 = EPI 0 under a name of its own: the long contractions (K >= 1024: PointShuffle2's after_conv, the step's dominant
    // kernel) no longer share an instantiation -- hence a row of `rocprofv3 --stats` -- with the K <= 480 products of the same tile
    constexpr int EPI = (EPI_T == 6) ? 0 : EPI_T;
    constexpr bool E_SCALE = (EPI == 1 || EPI == 4), E_R1 = (EPI >= 2), E_R2 = (EPI >= 3);
    float* __restrict__ Y = a.Y + (size_t)z * a.sy;
    const float lo = (a.act == 1) ? 0.f : -__builtin_inff();
    const bool has_bias = a.bias != nullptr;
    const float* __restrict__ R1 = (E_R1 && a.R1) ? a.R1 + (size_t)z * a.sr1 : nullptr;
    const float* __restrict__ R2 = (E_R2 && a.R2) ? a.R2 + (size_t)z * a.sr2 : nullptr;
    // addresses = wave-uniform row base (scalar registers) + one per-lane offset (4*fk rows + column): the 128
    // stores / residual loads of a lane then share three 32-bit offset registers instead of 128 64-bit pointers
    const int wm_u = __builtin_amdgcn_readfirstlane(wm), wn_u = __builtin_amdgcn_readfirstlane(wn);
    const int rbase = m0 + wm_u * (TM * 32), cbase = n0 + wn_u * (TN * 32);
    const long ldy = a.ldy, ldr1 = a.ldr1, ldr2 = a.ldr2, ldm = a.ldm;
    const unsigned offy = (unsigned)(4 * fk * ldy + fi), off1 = (unsigned)(4 * fk * ldr1 + fi), off2 = (unsigned)(4 * fk * ldr2 + fi);
    const unsigned offm = (unsigned)(4 * fk * ldm + fi);
    if constexpr (CI) {
        // a lane owns columns cbase + TN fi + j (j < TN) of rows rowu + 4 fk: TN-wide vector accesses throughout
        typedef float vecT __attribute__((ext_vector_type(TN)));
        const int col0 = cbase + TN * fi;
        const unsigned voy = (unsigned)(4 * fk * ldy + TN * fi), vo1 = (unsigned)(4 * fk * ldr1 + TN * fi), vo2 = (unsigned)(4 * fk * ldr2 + TN * fi);
        const unsigned vom = (unsigned)(4 * fk * ldm + TN * fi);
        vecT bv, sc, sh;
#pragma unroll
        for (int j = 0; j < TN; ++j) { bv[j] = 0.f; sc[j] = 1.f; sh[j] = 0.f; }
        if (has_bias) bv = *reinterpret_cast<const vecT*>(a.bias + col0);
        if constexpr (E_SCALE) {
            if (a.scale) { sc = *reinterpret_cast<const vecT*>(a.scale + col0); sh = *reinterpret_cast<const vecT*>(a.shift + col0); }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int rb = 0; rb < 16; rb += 8) {              // residual rows in batches of 8 (8 + 8 vector loads in flight)
                vecT r1v[8], r2v[8], mkv[8];
                if constexpr (EPI == 5) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const int rowu = rbase + i * 32 + ((rb + r) & 3) + 8 * ((rb + r) >> 2);
#pragma unroll
                        for (int j = 0; j < TN; ++j) mkv[r][j] = 1.f;
                        if (a.Mk != nullptr) {
                            const vecT m = *reinterpret_cast<const vecT*>(a.Mk + (size_t)rowu * ldm + cbase + vom);
#pragma unroll
                            for (int j = 0; j < TN; ++j) mkv[r][j] = (col0 + j < a.mcols) ? m[j] : 1.f;
                        }
                    }
                }
                if constexpr (EPI >= 4) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const int rowu = rbase + i * 32 + ((rb + r) & 3) + 8 * ((rb + r) >> 2);
#pragma unroll
                        for (int j = 0; j < TN; ++j) { r1v[r][j] = 0.f; r2v[r][j] = 0.f; }
                        if (R1) r1v[r] = *reinterpret_cast<const vecT*>(R1 + (size_t)rowu * ldr1 + cbase + vo1);
                    }
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const int rowu = rbase + i * 32 + ((rb + r) & 3) + 8 * ((rb + r) >> 2);
                        if (R2) r2v[r] = *reinterpret_cast<const vecT*>(R2 + (size_t)rowu * ldr2 + cbase + vo2);
                    }
                }
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int rowu = rbase + i * 32 + ((rb + r) & 3) + 8 * ((rb + r) >> 2);   // wave-uniform; the lane's row is rowu + 4*fk
                    vecT v;
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        float e = acc[i][j][rb + r];
                        if (has_bias) e = e + bv[j];
                        if constexpr (E_SCALE) {
                            if (a.scale) e = e * sc[j] + sh[j];
                        }
                        e = fmaxf(e, lo);
                        if constexpr (EPI >= 4) {
                            if (R1) e = e + r1v[r][j];
                            if (R2) e = e + r2v[r][j];
                        }
                        if constexpr (EPI == 5) e = (mkv[r][j] > 0.f) ? e : 0.f;
                        v[j] = e;
                    }
                    *reinterpret_cast<vecT*>(Y + (size_t)rowu * ldy + cbase + voy) = v;
                }
                if constexpr (EPI >= 4) __builtin_amdgcn_sched_barrier(0);
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int colu = cbase + j * 32;                      // wave-uniform first column of this 32-wide tile
        const int col = colu + fi;
        const bool col_ok = !EDGE || col < N;
        const int colc = col_ok ? col : 0;
        const float bv = has_bias ? a.bias[colc] : 0.f;
        float sc = 1.f, sh = 0.f;
        if constexpr (E_SCALE) {
            if (a.scale) { sc = a.scale[colc]; sh = a.shift[colc]; }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            // residuals of one 32x32 tile are fetched as a batch (16 + 16 loads in flight), then consumed: testing
            // R1 / R2 per element serialises load -> wait -> add -> store 128 times (+110 us on the after_conv GEMM)
            float r1v[16], r2v[16], mkv[16];
            if constexpr (EPI == 5) {                                             // EPI 5 = EPI 4 + the ReLU-gradient mask
                // same addressing as the residuals: wave-uniform row base + one 32-bit lane offset, 16 loads in flight
                const bool msk = a.Mk != nullptr && col < a.mcols && col_ok;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rowu = rbase + i * 32 + (r & 3) + 8 * (r >> 2);
                    const bool ok = msk && (!EDGE || rowu + 4 * fk < M);
                    mkv[r] = ok ? (a.Mk + (size_t)rowu * ldm + colu)[offm] : 1.f;
                }
            }
            if constexpr (EPI >= 4) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rowu = rbase + i * 32 + (r & 3) + 8 * (r >> 2);
                    const bool ok = col_ok && (!EDGE || rowu + 4 * fk < M);
                    r1v[r] = (R1 && ok) ? (R1 + (size_t)rowu * ldr1 + colu)[off1] : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rowu = rbase + i * 32 + (r & 3) + 8 * (r >> 2);
                    const bool ok = col_ok && (!EDGE || rowu + 4 * fk < M);
                    r2v[r] = (R2 && ok) ? (R2 + (size_t)rowu * ldr2 + colu)[off2] : 0.f;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rowu = rbase + i * 32 + (r & 3) + 8 * (r >> 2);      // wave-uniform; the lane's row is rowu + 4*fk
                if (col_ok && (!EDGE || rowu + 4 * fk < M)) {
                    float v = acc[i][j][r];
                    if (has_bias) v = v + bv;
                    if constexpr (E_SCALE) {
                        if (a.scale) v = v * sc + sh;
                    }
                    v = fmaxf(v, lo);
                    if constexpr (EPI >= 4) {
                        if (R1) v = v + r1v[r];
                        if (R2) v = v + r2v[r];
                    }
                    if constexpr (EPI == 5) v = (mkv[r] > 0.f) ? v : 0.f;
                    (Y + (size_t)rowu * ldy + colu)[offy] = v;
                }
            }
            if constexpr (EPI >= 4) __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int BM, int BN, int WM, int WN, int BK, bool TRANSB, bool EDGE, int EPI>
static int launch_epi(const LinearArgs& a, dim3 grid, hipStream_t s) {
    constexpr size_t bytes = LinearLds<BM, BN, BK, TRANSB, EDGE>::BYTES;
    auto kern = linear_mfma_kernel<BM, BN, WM, WN, BK, TRANSB, EDGE, EPI>;
    if (bytes > 64 * 1024) {
        static DevOnce done;      // opt in to > 64 KiB of dynamic LDS once per instantiation
        if (done.needed()) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
            if (e != hipSuccess) return (int)e;
            done.done();
        }
    }
    hipLaunchKernelGGL(kern, grid, dim3(64 * (WM * WN + (LinearLds<BM, BN, BK, TRANSB, EDGE>::DMA ? 4 : WM * WN))), bytes, s, a);
    return (int)hipGetLastError();
}

template <int BM, int BN, int WM, int WN, int BK, bool TRANSB, bool EDGE>
static int launch_one(const LinearArgs& a, dim3 grid, hipStream_t s) {
    if (!a.scale && !a.R1 && !a.R2 && !a.Mk) {
        if constexpr (BM == 128 && BN == 256 && BK == 16 && !TRANSB && !EDGE) {
            if (a.K >= 1024) return launch_epi<BM, BN, WM, WN, BK, TRANSB, EDGE, 6>(a, grid, s);
        }
        return launch_epi<BM, BN, WM, WN, BK, TRANSB, EDGE, 0>(a, grid, s);
    }
    if (a.scale && !a.R1 && !a.R2 && !a.Mk) return launch_epi<BM, BN, WM, WN, BK, TRANSB, EDGE, 1>(a, grid, s);
    if (a.Mk) return launch_epi<BM, BN, WM, WN, BK, TRANSB, EDGE, 5>(a, grid, s);
    return launch_epi<BM, BN, WM, WN, BK, TRANSB, EDGE, 4>(a, grid, s);
}

// BK: slab depth of the interior, untransposed path (16 = the DMA pipeline); BKR: of the register-staged paths (transposed B, edge tiles)
template <int BM, int BN, int WM, int WN, int BK, int BKR = BK>
static int launch_linear(const LinearArgs& a, int batch, bool transb, hipStream_t s) {
    dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, batch);
    const bool aligned = ((a.ldx & 3) == 0) && ((a.ldw & 3) == 0) && ((a.sx & 3) == 0) && ((a.sw & 3) == 0) &&
                         ((((uintptr_t)a.X) & 15) == 0) && ((((uintptr_t)a.W) & 15) == 0);
    // (round 5: interior tiles store / fetch TN-wide vectors in the epilogue -- Y, bias and the optional epilogue operands must allow it)
    const auto al16 = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
    const bool out_aligned = ((a.ldy & 3) == 0) && ((a.sy & 3) == 0) && al16(a.Y) && (!a.bias || al16(a.bias)) &&
                             (!a.scale || (al16(a.scale) && al16(a.shift))) &&
                             (!a.R1 || (((a.ldr1 & 3) == 0) && ((a.sr1 & 3) == 0) && al16(a.R1))) &&
                             (!a.R2 || (((a.ldr2 & 3) == 0) && ((a.sr2 & 3) == 0) && al16(a.R2))) &&
                             (!a.Mk || (((a.ldm & 3) == 0) && al16(a.Mk) && a.mcols >= a.N));   // the vector epilogue loads whole TN-wide mask rows:
                                                                                                   // a mask narrower than the product keeps the guarded per-column path
    const bool tiles_ok = aligned && out_aligned && (a.M % BM == 0) && (a.N % BN == 0);
    if (tiles_ok && !transb && (a.K % BK == 0)) return launch_one<BM, BN, WM, WN, BK, false, false>(a, grid, s);
    if (tiles_ok && transb && (a.K % BKR == 0)) return launch_one<BM, BN, WM, WN, BKR, true, false>(a, grid, s);
    if (BKR != BK && tiles_ok && !transb && (a.K % BKR == 0)) return launch_one<BM, BN, WM, WN, BKR, false, false>(a, grid, s);
    if (transb) return launch_one<BM, BN, WM, WN, BKR, true, true>(a, grid, s);
    return launch_one<BM, BN, WM, WN, BKR, false, true>(a, grid, s);
}

int linear_skinny_dispatch(int M, int K, int N, const float* X, long ldx, const float* W, long ldw, int transb, const float* bias,
                           int act, float* Y, long ldy, const float* R1, long ldr1, const float* Mk, long ldm, int mcols,
                           hipStream_t st);   // linear_skinny.hip

}  // namespace dispu

using namespace dispu;

// Block-tile choice of dispu_linear as BM*1000 + BN (128128, 64128, 128064, 64064): the largest tile that still
// yields >= 256 workgroups (one per CU).  Exported so a profiler can name the kernel instantiation.
static int g_tile_override = 0;     // tools/gemm_bench.py only (dispu_debug_linear_tile): force one tile for a sweep; 0 = the rule below
DISPU_EXPORT void dispu_debug_linear_tile(int code) { g_tile_override = code; }

// dma: the product can take the DMA pipeline (untransposed, K % 16 == 0; alignment assumed) -- the 64 x 128 tile only pays there
static int linear_tile_rule(int batch, int M, int N, bool dma) {
    if (g_tile_override > 0) return g_tile_override;
    const long mb128 = (long)((M + 127) / 128) * batch;
    if (N >= 256 && N % 256 == 0 && mb128 * (N / 256) >= 256) return 128257;   // 128x256 tile, BK 16: 64x128 per wave
    if (N > 64 && N % 128 != 0 && N % 64 == 0 && mb128 >= 64) return 64064;    // e.g. N = 320: five full 64-wide tiles beat a half-empty edge tile (64 x 64: 34 us, 128 x 64: 41 us at 32768 x 128 x 320)
    const long mb64 = (long)((M + 63) / 64) * batch;
    if (N > 64) {
        if (mb128 * ((N + 127) / 128) >= 256) return 128128;
        // 64 x 128 tiles once they fill the chip: 512 workgroups on the register-staged paths (round 3), 256 on the DMA pipeline
        // (round 5, tools/gemm_bench.py: 8192 x 480 x 256 25.5 -> 22.9 us, 8192 x 256 x 256 15.8 -> 14.3 us; edge / transposed
        // products -- 8192 x 134 x 256: 13.9 vs 18.3 us -- keep 64 x 64)
        const long wg = mb64 * ((N + 127) / 128);
        return (wg >= 512 || (dma && N % 128 == 0 && wg >= 256)) ? 64128 : 64064;
    }
    return mb128 >= 256 ? 128064 : 64064;
}

DISPU_EXPORT int dispu_linear_tile(int batch, int M, int N) { return linear_tile_rule(batch, M, N, true); }
// The tile the launch of THIS product takes: transposed-B products and K % 16 != 0 never use the DMA pipeline, and with 256 - 511
// workgroups that changes the choice (64 x 64 instead of 64 x 128).  dispu_linear_tile (ABI <= 4) assumes the DMA pipeline.
DISPU_EXPORT int dispu_linear_tile2(int batch, int M, int K, int N, int transb) { return linear_tile_rule(batch, M, N, !transb && (K % 16) == 0); }

static int linear_impl(int batch, int M, int K, int N, const float* X, long ldx, long sx, const float* W, long ldw, long sw, int transb,
                       const float* bias, const float* scale, const float* shift, int act, float* Y, long ldy, long sy, const float* R1,
                       long ldr1, long sr1, const float* R2, long ldr2, long sr2, const float* Mk, long ldm, int mcols, void* stream);

// Y = R2 + R1 + act(X.W + bias); see include/dispu_hip.h for the argument contract.
DISPU_EXPORT int dispu_linear(int batch, int M, int K, int N, const float* X, long ldx, long sx, const float* W,
                              long ldw, long sw, int transb, const float* bias, int act, float* Y, long ldy, long sy,
                              const float* R1, long ldr1, long sr1, const float* R2, long ldr2, long sr2, void* stream) {
    return linear_impl(batch, M, K, N, X, ldx, sx, W, ldw, sw, transb, bias, nullptr, nullptr, act, Y, ldy, sy, R1, ldr1,
                       sr1, R2, ldr2, sr2, nullptr, 0, 0, stream);
}

// dispu_linear followed by a ReLU-gradient mask: Y[m][n] = 0 where Mk[m][n] <= 0, for the columns n < mcols of this product.
// Training step: dX = dZ . W^T (+ R1, the gradient accumulated so far) of a layer whose INPUT was a ReLU output Mk -- the
// relu_grad of the layer below rides in this epilogue instead of a separate pass over dX (tf_util.py:100-115 backward).
DISPU_EXPORT int dispu_linear_masked(int batch, int M, int K, int N, const float* X, long ldx, long sx, const float* W,
                                     long ldw, long sw, int transb, const float* bias, int act, float* Y, long ldy, long sy,
                                     const float* R1, long ldr1, long sr1, const float* Mk, long ldm, int mcols, void* stream) {
    if (Mk && batch != 1) return (int)hipErrorInvalidValue;
    return linear_impl(batch, M, K, N, X, ldx, sx, W, ldw, sw, transb, bias, nullptr, nullptr, act, Y, ldy, sy, R1, ldr1,
                       sr1, nullptr, 0, 0, Mk, ldm, Mk ? mcols : 0, stream);
}

// dispu_linear with an inference-BatchNorm epilogue: Y = R2 + R1 + act( (X.W + bias) * scale + shift ).
DISPU_EXPORT int dispu_linear_bn(int batch, int M, int K, int N, const float* X, long ldx, long sx, const float* W,
                                 long ldw, long sw, int transb, const float* bias, const float* scale, const float* shift,
                                 int act, float* Y, long ldy, long sy, const float* R1, long ldr1, long sr1,
                                 const float* R2, long ldr2, long sr2, void* stream) {
    return linear_impl(batch, M, K, N, X, ldx, sx, W, ldw, sw, transb, bias, scale, shift, act, Y, ldy, sy, R1, ldr1, sr1, R2, ldr2, sr2,
                       nullptr, 0, 0, stream);
}

static int linear_impl(int batch, int M, int K, int N, const float* X, long ldx, long sx, const float* W, long ldw, long sw, int transb,
                       const float* bias, const float* scale, const float* shift, int act, float* Y, long ldy, long sy, const float* R1,
                       long ldr1, long sr1, const float* R2, long ldr2, long sr2, const float* Mk, long ldm, int mcols, void* stream) {
    if (batch < 0 || M < 0 || K <= 0 || N <= 0 || !X || !W || !Y || act < 0 || act > 1 || ((scale == nullptr) != (shift == nullptr)))
        return (int)hipErrorInvalidValue;
    if (batch == 0 || M == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (batch == 1 && !scale && !R2) {
        const int rc = linear_skinny_dispatch(M, K, N, X, ldx, W, ldw, transb, bias, act, Y, ldy, R1, ldr1, Mk, ldm, mcols, s);   // latency-bound shapes
        if (rc >= 0) return rc;
        // a few columns past a multiple of 128 (N = 134: the PointShuffle conv0 gradient): the tiled kernel would spend a second,
        // almost empty column of 128-wide edge tiles on them.  Columns are independent: the multiple of 128 goes to the tiled
        // kernel (interior path when aligned), the tail to the skinny kernel.
        const int tail = N % 128, n0 = N - tail;
        if (n0 > 0 && tail > 0 && tail <= 32) {
            const float* Wt = transb ? W + (size_t)n0 * ldw : W + n0;
            if (linear_skinny_dispatch(M, K, tail, X, ldx, Wt, ldw, transb, bias ? bias + n0 : nullptr, act, Y + n0, ldy,
                                       R1 ? R1 + n0 : nullptr, ldr1, Mk ? Mk + n0 : nullptr, ldm, mcols - n0, s) >= 0)
                return linear_impl(batch, M, K, n0, X, ldx, sx, W, ldw, sw, transb, bias, scale, shift, act, Y, ldy, sy, R1, ldr1, sr1,
                                   R2, ldr2, sr2, Mk, ldm, mcols, stream);
        }
    }
    LinearArgs a{M, K, N, X, ldx, sx, W, ldw, sw, bias, Y, ldy, sy, R1, ldr1, sr1, R2, ldr2, sr2, act, scale, shift,
                 (Mk && mcols > 0) ? Mk : nullptr, ldm, mcols};
    const bool tb = transb != 0;
    switch (linear_tile_rule(batch, M, N, !tb && (K % 16) == 0)) {
        // interior untransposed products: BK = 16 slabs through the DMA pipeline; transposed B / edge tiles: register-staged, BK = 32
        case 128257: return launch_linear<128, 256, 2, 2, 16>(a, batch, tb, s);
        case 128128: return launch_linear<128, 128, 2, 2, 16, 32>(a, batch, tb, s);
        case 64128: return launch_linear<64, 128, 2, 2, 16, 32>(a, batch, tb, s);
        case 128064: return launch_linear<128, 64, 2, 2, 16, 32>(a, batch, tb, s);
        default: return launch_linear<64, 64, 2, 2, 16, 32>(a, batch, tb, s);
    }
}
