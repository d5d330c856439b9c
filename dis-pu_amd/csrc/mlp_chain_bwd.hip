// Backward of the generator's two head chains (csrc/mlp_chain.hip) in ONE launch each, gfx950 fp32 MFMA.
//   forward:  X [rows, K0] -W1-> Y1 [N1] -W2-> Y2 [N2] -W3-> Y3 [64] -W4-> Z [3]      (ReLU after W1, W2, W3)
//   backward: dZ [rows, 3] -> dY3 = (dZ . W4^T) * [Y3 > 0] -> dY2 = (dY3 . W3^T) * [Y2 > 0] -> dY1 = (dY2 . W2^T (+ R + R2)) * [Y1 > 0]
//             -> dX = dY1 . W1^T, written through up to three ReLU masks (the chain input was a ReLU output, or a sum of three)
//   coarse head (ops.py:1186-1192, 1089-1104): 3 -> 64 -> 256 -> 128 (+ the gradient already accumulated in dup128) -> 256 (mask up256)
//   fine head   (ops.py:1079-1083, 1106-1108): 3 -> 64 -> 256 -> 256 -> 256 (three masked copies: after_conv / skip / non-local branch)
// As separate launches a chain's backward is four dX GEMMs of 4 - 32 K-slabs (15 - 40 us each at 8 patches, every one a full
// launch + prologue + epilogue on the critical path of the training step).  Here a workgroup keeps its rows on chip exactly like the
// forward kernel: the gradient tile lives in LDS in the k-major layout of the MFMA A operand, only the TRANSPOSED weights stream in
// (row-major [K, N] copies kept by the trainer: slabs are contiguous 2048-float pieces), and every layer's dZ is also written to
// HBM because the weight-gradient products (dW = Y^T . dZ, on the side streams) read it.  Wave-specialised: waves 0-3 MFMAs,
// waves 4-7 the weight stream.  Each product is the ascending-k fmaf chain of dispu_linear(transb = 1).
#include "common.h"

namespace dispu {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ChainBwdArgs {
    long rows;
    const float* dZ; long lddz;                     // [rows, 3]
    const float* W4;                                // [64, 3] (the forward matrix: the 3 -> 64 product runs on the VALU)
    const float* Wt3; const float* Wt2; const float* Wt1;     // W3^T [64, N2], W2^T [N2, N1], W1^T [N1, K0], row-major
    const float* Y3; long ldy3;                     // ReLU outputs of the forward chain: the masks
    const float* Y2; long ldy2;
    const float* Y1; long ldy1;
    const float* R; long ldr;                       // optional [rows, N1]: added to dY2 . W2^T before the mask
    const float* R2; long ldr2;                     // optional second one
    float* D3; long ldd3;                           // dY3 [rows, 64]
    float* D2; long ldd2;                           // dY2 [rows, N2]
    float* D1; long ldd1;                           // dY1 [rows, N1]   (may alias R)
    const float* Ma; const float* Mb; const float* Mc; long ldm;   // masks of the chain input [rows, K0] (null: unmasked copy)
    float* Da; float* Db; float* Dc; long ldd0;     // dX through mask a / b / c (Db, Dc optional)
};

constexpr int MB_KMAX = 256;
constexpr int MB_ACT = MB_KMAX * (64 + 1);
__host__ __device__ constexpr int mb_bk(int n) { return 2048 / n; }
constexpr int MB_WST = 16 * (128 + 4);                             // floats per weight stage (8 x 260 or 16 x 132)
constexpr size_t MB_LDS_BYTES = (size_t)(MB_ACT + 2 * MB_WST) * sizeof(float);

// one backward layer on the MFMA waves: acc = act[0:K] . Wt (slabs g0 ..), then v = acc (+ R), zeroed where the mask <= 0;
// LAST = false: act[0:N] <- v (k-major) and D <- v;  LAST = true: Da / Db / Dc <- v through their masks.
template <int K, int N, int BM, bool LAST, bool RES = false>
__device__ __forceinline__ void chain_layer_bwd(float* act, const float* wst, int g0, const ChainBwdArgs& a, const float* Mk,
                                                long ldmk, float* D, long ldd, long row0, int wm, int wn, int fi, int fk) {
    // the four MFMA waves as WMW x WNW: 64-row workgroups 2 x 2 (32 rows x N / 2 columns per wave), 32-row workgroups 1 x 4 (32 rows x
    // N / 4 columns) -- the latter doubles the workgroup count where 64-row tiles leave half the chip idle (8 training patches: 8192 rows)
    constexpr int WMW = BM / 32, WNW = 4 / WMW, NW = N / WNW, TNW = NW / 32, LDW = N + 4, BK = mb_bk(N), LDA = BM + 1;
    static_assert(BM == 32 || BM == 64, "the epilogue operands of a layer are prefetched into registers: 32- or 64-row workgroups only");
    static_assert(TNW >= 1, "layer too narrow for the wave grid");
    f32x16 acc[TNW];
    // the epilogue's mask (and residual) values, requested BEFORE the product loop: they arrive while the MFMAs run.  (Loaded inside
    // the epilogue the compiler serialised load -> wait -> store per element: the stores may alias the next loads for all it knows.)
    float mk[TNW][16], rr[RES ? TNW : 1][16], rr2[RES ? TNW : 1][16];
    const size_t rbase = (size_t)(row0 + wm * 32 + 4 * fk);
#pragma unroll
    for (int j = 0; j < TNW; ++j) {
        const int n = wn * NW + j * 32 + fi;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const size_t gr = rbase + (r & 3) + 8 * (r >> 2);
            acc[j][r] = 0.f;
            if constexpr (!LAST) {
                mk[j][r] = Mk[gr * ldmk + n];
                if constexpr (RES) { rr[j][r] = 0.f; rr2[j][r] = 0.f; }
            } else {
                mk[j][r] = a.Ma ? a.Ma[gr * a.ldm + n] : 1.f;
            }
        }
    }
    if constexpr (RES) {
        if (a.R) {
#pragma unroll
            for (int j = 0; j < TNW; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    rr[j][r] = a.R[(rbase + (r & 3) + 8 * (r >> 2)) * a.ldr + wn * NW + j * 32 + fi];
        }
        if (a.R2) {
#pragma unroll
            for (int j = 0; j < TNW; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    rr2[j][r] = a.R2[(rbase + (r & 3) + 8 * (r >> 2)) * a.ldr2 + wn * NW + j * 32 + fi];
        }
    }
    for (int s = 0; s < K / BK; ++s) {
        const float* ws = wst + ((g0 + s) & 1) * MB_WST;
        const float* as = act + (s * BK) * LDA;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float bf[TNW];
            const float af = as[(kk + fk) * LDA + wm * 32 + fi];
#pragma unroll
            for (int j = 0; j < TNW; ++j) bf[j] = ws[(kk + fk) * LDW + wn * NW + j * 32 + fi];
#pragma unroll
            for (int j = 0; j < TNW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf[j], acc[j], 0, 0, 0);
        }
        __syncthreads();
    }
    __syncthreads();                                              // every wave has finished reading this layer's input
    if constexpr (!LAST) {
#pragma unroll
        for (int j = 0; j < TNW; ++j) {
            const int n = wn * NW + j * 32 + fi;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                float v = acc[j][r];
                if constexpr (RES) v = (v + rr[j][r]) + rr2[j][r];
                if (mk[j][r] <= 0.f) v = 0.f;
                act[n * LDA + row] = v;
                D[(size_t)(row0 + row) * ldd + n] = v;
            }
        }
    } else {
        // dX through its masks: Ma was prefetched; Mb / Mc (the fine head: three masked copies) are loaded as a block each
#pragma unroll
        for (int j = 0; j < TNW; ++j) {
            const int n = wn * NW + j * 32 + fi;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                a.Da[(rbase + (r & 3) + 8 * (r >> 2)) * a.ldd0 + n] = (mk[j][r] <= 0.f) ? 0.f : acc[j][r];
        }
        if (a.Db) {
#pragma unroll
            for (int j = 0; j < TNW; ++j) {
                const int n = wn * NW + j * 32 + fi;
#pragma unroll
                for (int r = 0; r < 16; ++r) mk[j][r] = a.Mb[(rbase + (r & 3) + 8 * (r >> 2)) * a.ldm + n];
            }
#pragma unroll
            for (int j = 0; j < TNW; ++j) {
                const int n = wn * NW + j * 32 + fi;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    a.Db[(rbase + (r & 3) + 8 * (r >> 2)) * a.ldd0 + n] = (mk[j][r] <= 0.f) ? 0.f : acc[j][r];
            }
        }
        if (a.Dc) {
#pragma unroll
            for (int j = 0; j < TNW; ++j) {
                const int n = wn * NW + j * 32 + fi;
#pragma unroll
                for (int r = 0; r < 16; ++r) mk[j][r] = a.Mc[(rbase + (r & 3) + 8 * (r >> 2)) * a.ldm + n];
            }
#pragma unroll
            for (int j = 0; j < TNW; ++j) {
                const int n = wn * NW + j * 32 + fi;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    a.Dc[(rbase + (r & 3) + 8 * (r >> 2)) * a.ldd0 + n] = (mk[j][r] <= 0.f) ? 0.f : acc[j][r];
            }
        }
    }
    __syncthreads();                                              // the next layer's input is complete
}

template <int K0, int N1, int N2, int BM, bool RES>
__global__ __launch_bounds__(512) void mlp_chain_bwd_kernel(ChainBwdArgs a) {
    constexpr int LDA = BM + 1;
    static_assert(K0 <= MB_KMAX && N1 <= MB_KMAX && N2 <= MB_KMAX && N1 >= 128 && N2 >= 128 && K0 >= 128, "chain shape outside the LDS plan");
    constexpr int S1 = 64 / mb_bk(N2), S2 = N2 / mb_bk(N1), S3 = N1 / mb_bk(K0), G = S1 + S2 + S3;
    static_assert(S1 % 2 == 0 && S2 % 2 == 0 && S3 % 2 == 0, "layers must start on stage 0");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* act = lds;                                             // [k][LDA]
    float* wst = act + MB_ACT;                                    // [2][bk][N + 4]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long row0 = (long)blockIdx.x * BM;

    // ---- 3 -> 64 on the VALU, all 512 threads: dY3[row][k] = (dz0 W4[k][0] + dz1 W4[k][1] + dz2 W4[k][2]) * [Y3[row][k] > 0];
    //      16 threads cover one row's 64 columns (float4 each): coalesced mask reads / dY3 stores
#pragma unroll
    for (int u = 0; u < BM * 16 / 512; ++u) {
        const int idx = threadIdx.x + u * 512;
        const int row = idx >> 4, k = (idx & 15) * 4;
        const size_t gr = (size_t)(row0 + row);
        const float d0 = a.dZ[gr * a.lddz + 0], d1 = a.dZ[gr * a.lddz + 1], d2 = a.dZ[gr * a.lddz + 2];
        const float4 w0 = *reinterpret_cast<const float4*>(a.W4 + k * 3), w1 = *reinterpret_cast<const float4*>(a.W4 + k * 3 + 4),
                     w2 = *reinterpret_cast<const float4*>(a.W4 + k * 3 + 8);
        const float4 m = *reinterpret_cast<const float4*>(a.Y3 + gr * a.ldy3 + k);
        float4 v;
        v.x = __builtin_fmaf(d2, w0.z, __builtin_fmaf(d1, w0.y, d0 * w0.x));
        v.y = __builtin_fmaf(d2, w1.y, __builtin_fmaf(d1, w1.x, d0 * w0.w));
        v.z = __builtin_fmaf(d2, w2.x, __builtin_fmaf(d1, w1.w, d0 * w1.z));
        v.w = __builtin_fmaf(d2, w2.w, __builtin_fmaf(d1, w2.z, d0 * w2.y));
        if (m.x <= 0.f) v.x = 0.f;
        if (m.y <= 0.f) v.y = 0.f;
        if (m.z <= 0.f) v.z = 0.f;
        if (m.w <= 0.f) v.w = 0.f;
        *reinterpret_cast<float4*>(a.D3 + gr * a.ldd3 + k) = v;
        act[(k + 0) * LDA + row] = v.x;
        act[(k + 1) * LDA + row] = v.y;
        act[(k + 2) * LDA + row] = v.z;
        act[(k + 3) * LDA + row] = v.w;
    }

    if (wave >= 4) {
        // ------------------------------------------------------------------------------------ loader waves: the weight stream
        const int tid = threadIdx.x - 256;
        // Round 5 (as in csrc/mlp_chain.hip): a loader instruction next to an MFMA wave takes 100 - 250 cycles to issue, and with 64-row
        // workgroups a slab is only 16 MFMAs per wave (1024 pipe cycles) -- the ~35 instructions per slab of the rolled loop (runtime layer
        // selects, 64-bit address arithmetic, LDS offsets from divisions) were the bound.  Fully unrolled schedule, buffer loads whose slab
        // offset is a scalar, one precomputed LDS offset per layer width: ~8 instructions per slab.
        float4 wa0, wa1, wb0, wb1;
        typedef unsigned int mb_u32x4 __attribute__((ext_vector_type(4)));
        const __amdgpu_buffer_rsrc_t rw3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Wt3), 0, 64 * N2 * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rw2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Wt2), 0, N2 * N1 * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rw1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Wt1), 0, N1 * K0 * 4, 0x00020000);
        auto bload = [](__amdgpu_buffer_rsrc_t rs, int voff, int soff) -> float4 {
            const mb_u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
            return make_float4(__uint_as_float(r[0]), __uint_as_float(r[1]), __uint_as_float(r[2]), __uint_as_float(r[3]));
        };
        const int wvoff = tid * 16;
        // LDS float index of the thread's first float4 inside a stage, per layer width (the second one is 256 float4 further on)
        const int wd3 = (tid / (N2 / 4)) * (N2 + 4) + (tid % (N2 / 4)) * 4;
        const int wd2 = (tid / (N1 / 4)) * (N1 + 4) + (tid % (N1 / 4)) * 4;
        const int wd1 = (tid / (K0 / 4)) * (K0 + 4) + (tid % (K0 / 4)) * 4;
#define MB_LOAD(g, r0, r1)                                                                    \
        do {                                                                                  \
            if ((g) < S1) { r0 = bload(rw3, wvoff, (g) * 8192); r1 = bload(rw3, wvoff, (g) * 8192 + 4096); }                                   \
            else if ((g) < S1 + S2) { r0 = bload(rw2, wvoff, ((g) - S1) * 8192); r1 = bload(rw2, wvoff, ((g) - S1) * 8192 + 4096); }           \
            else { r0 = bload(rw1, wvoff, ((g) - S1 - S2) * 8192); r1 = bload(rw1, wvoff, ((g) - S1 - S2) * 8192 + 4096); }                    \
        } while (0)
#define MB_STORE(g, r0, r1)                                                                   \
        do {                                                                                  \
            float* st_ = wst + ((g) & 1) * MB_WST;                                            \
            if ((g) < S1) { *reinterpret_cast<float4*>(st_ + wd3) = r0; *reinterpret_cast<float4*>(st_ + wd3 + (1024 / N2) * (N2 + 4)) = r1; }           \
            else if ((g) < S1 + S2) { *reinterpret_cast<float4*>(st_ + wd2) = r0; *reinterpret_cast<float4*>(st_ + wd2 + (1024 / N1) * (N1 + 4)) = r1; }  \
            else { *reinterpret_cast<float4*>(st_ + wd1) = r0; *reinterpret_cast<float4*>(st_ + wd1 + (1024 / K0) * (K0 + 4)) = r1; }                    \
        } while (0)
#define MB_STEP(g, r0, r1) /* r0 r1 hold slab g + 1; refilled with slab g + 3 */              \
        do {                                                                                  \
            if ((g) + 1 < G) {                                                                \
                MB_STORE((g) + 1, r0, r1);                                                    \
                if ((g) + 3 < G) MB_LOAD((g) + 3, r0, r1);                                    \
            }                                                                                 \
            __syncthreads();                                                                  \
            if ((g) + 1 == S1 || (g) + 1 == S1 + S2 || (g) + 1 == G) {                        \
                __syncthreads();                                                              \
                __syncthreads();                                                              \
            }                                                                                 \
        } while (0)
        static_assert(G % 2 == 0, "the loader loop is unrolled by two");
        MB_LOAD(0, wa0, wa1);
        MB_STORE(0, wa0, wa1);
        MB_LOAD(1, wb0, wb1);
        MB_LOAD(2, wa0, wa1);
        __syncthreads();                                          // dY3 tile and slab 0 are in place
#pragma unroll
        for (int g = 0; g < G; g += 2) {
            MB_STEP(g, wb0, wb1);
            MB_STEP(g + 1, wa0, wa1);
        }
#undef MB_LOAD
#undef MB_STORE
#undef MB_STEP
        return;
    }

    // ------------------------------------------------------------------------------------------ MFMA waves
    const int wm = (BM == 64) ? (wave >> 1) : 0, wn = (BM == 64) ? (wave & 1) : wave;
    const int fi = lane & 31, fk = lane >> 5;
    __syncthreads();
    chain_layer_bwd<64, N2, BM, false>(act, wst, 0, a, a.Y2, a.ldy2, a.D2, a.ldd2, row0, wm, wn, fi, fk);
    chain_layer_bwd<N2, N1, BM, false, RES>(act, wst, S1, a, a.Y1, a.ldy1, a.D1, a.ldd1, row0, wm, wn, fi, fk);
    chain_layer_bwd<N1, K0, BM, true>(act, wst, S1 + S2, a, nullptr, 0, nullptr, 0, row0, wm, wn, fi, fk);
}

}  // namespace dispu

using namespace dispu;

// Backward of dispu_mlp_chain(_stash): see the header.  rows % 32 == 0; (K0, N1, N2) in {(256,128,256), (256,256,256)}; every pointer
// 16-byte aligned, every leading dimension a multiple of 4 (except lddz).  D1 may alias R (same element read then written by one lane).
DISPU_EXPORT int dispu_mlp_chain_grad(long rows, int K0, int N1, int N2, const float* dZ, long lddz, const float* W4, const float* Wt3,
                                      const float* Wt2, const float* Wt1, const float* Y3, long ldy3, const float* Y2, long ldy2,
                                      const float* Y1, long ldy1, const float* R, long ldr, const float* R2, long ldr2, float* D3, long ldd3,
                                      float* D2, long ldd2, float* D1, long ldd1, const float* Ma, const float* Mb, const float* Mc, long ldm, float* Da, float* Db,
                                      float* Dc, long ldd0, void* stream) {
    if (rows < 0 || (rows % 32) != 0 || !dZ || !W4 || !Wt3 || !Wt2 || !Wt1 || !Y3 || !Y2 || !Y1 || !D3 || !D2 || !D1 || !Da ||
        (Db && !Mb) || (Dc && !Mc) || ((ldy3 | ldd3) & 3) ||
        ((((uintptr_t)W4) | ((uintptr_t)Wt3) | ((uintptr_t)Wt2) | ((uintptr_t)Wt1) | ((uintptr_t)Y3) | ((uintptr_t)D3)) & 15))
        return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    const bool coarse = (K0 == 256 && N1 == 128 && N2 == 256), fine = (K0 == 256 && N1 == 256 && N2 == 256);
    if (!coarse && !fine) return (int)hipErrorInvalidValue;
    ChainBwdArgs a{rows, dZ, lddz, W4, Wt3, Wt2, Wt1, Y3, ldy3, Y2, ldy2, Y1, ldy1, R, ldr, R2, ldr2, D3, ldd3, D2, ldd2, D1, ldd1,
                   Ma, Mb, Mc, ldm, Da, Db, Dc, ldd0};
    hipStream_t s = (hipStream_t)stream;
    // 64-row workgroups (a layer's epilogue operands -- masks, residual -- wait in registers during its product loop, which leaves no room
    // for a 128-row tile's accumulators); 32-row workgroups while 64-row ones would leave CUs idle (fewer than 192 of them: the 8-patch
    // training step's 8192 rows are 128) or the row count demands it
    const bool half = (rows % 64) != 0 || rows / 64 < 192;
    const dim3 grid((unsigned)(rows / (half ? 32 : 64)));
    auto launch = [&](auto kern) -> int {
        static DevOnce once;
        if (once.needed()) {
            DISPU_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)MB_LDS_BYTES));
            once.done();
        }
        hipLaunchKernelGGL(kern, grid, dim3(512), MB_LDS_BYTES, s, a);
        return (int)hipGetLastError();
    };
    const bool res = R || R2;
    if (res && !coarse) return (int)hipErrorInvalidValue;      // residual inputs: the 128-wide Y1 only (register budget of the prefetch)
    if (half) {
        if (coarse) return res ? launch(mlp_chain_bwd_kernel<256, 128, 256, 32, true>) : launch(mlp_chain_bwd_kernel<256, 128, 256, 32, false>);
        return launch(mlp_chain_bwd_kernel<256, 256, 256, 32, false>);
    }
    if (coarse) return res ? launch(mlp_chain_bwd_kernel<256, 128, 256, 64, true>) : launch(mlp_chain_bwd_kernel<256, 128, 256, 64, false>);
    return launch(mlp_chain_bwd_kernel<256, 256, 256, 64, false>);
}
