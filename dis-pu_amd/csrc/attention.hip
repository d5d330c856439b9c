// PointNonLocalCell attention (Common/ops.py:326-339: softmax(Q.K^T / sqrt(64)) . V) fused on gfx950, fp32.
//
// The reference (and the unfused path here) materialises the [B, 1024, 1024] logits in HBM (134 MB at B = 32),
// runs a softmax pass over them and reads them again for the second matmul.  This kernel keeps everything on
// chip, flash-attention style, with the products on v_mfma_f32_32x32x2_f32 in the TRANSPOSED orientation:
//
//   S^T[key][q] = sum_d K[key][d] Q[q][d]        A operand = K tile (LDS), B operand = Q (registers, per wave)
//   O^T[d][q]  += sum_key V[key][d] P^T[key][q]   A operand = V tile (LDS), B operand = P^T
//
// so a query is an MFMA *column*: lane (q = lane & 31, h = lane >> 5) holds, for its query, 16 of the tile's 32
// logits in its accumulator registers.  The online softmax is therefore per-lane arithmetic plus ONE exchange with
// the partner lane (lane ^ 32), the rescale of O^T is a per-lane multiply, and the exponentiated registers are
// fed back UNMOVED as the B operand of the second product (step r pairs key rows r' and r'+4 - exactly the rows
// the two half-waves hold in register r).  A workgroup = 128 queries of one cloud = 4 MFMA waves (fragment reads, MFMAs and
// the per-lane softmax arithmetic only) + 4 loader waves that stream the K|V tiles of 32 keys through two LDS stages
// (global -> registers -> LDS, one tile in flight, one barrier per tile): a wave that issues its own global loads
// next to its MFMAs keeps the matrix pipe under 50 % busy (see linear.hip).
#include "common.h"

namespace dispu {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int FA_D = 64, FA_TK = 32, FA_LDK = FA_D + 1;
constexpr int FA_STAGE = FA_TK * FA_LDK + FA_TK * FA_D;     // floats: K tile [32][65] + V tile [32][64]

// BP = true: the cell's output projection (conv_back_project, ops.py:341-343: 64 -> 256, bias, ReLU) runs as the epilogue:
//   Y^T[o][q] = relu(b[o] + sum_d W[d][o] O^T[d][q])
// again in the transposed orientation, so the normalised O^T accumulators are the B operand AS THEY ARE (step (c, r) pairs
// the d rows the two half-waves hold in register r) and W [64][256] (64 KB, copied to LDS by the loader waves while the first
// K|V tile is in flight) is the A operand, one conflict-free ds_read_b32 per MFMA.  The [rows, 64] attention output never
// reaches HBM and the separate 64 -> 256 GEMM launch (28 % of the MFMA peak, 21 % of HBM: bound by neither) disappears.
// d is contracted in the order 0,4,1,5,... (not ascending): like the softmax itself this branch is tolerance-checked.
constexpr int FA_BPN = 256;
#ifdef FA_STAMPS
__device__ unsigned long long fa_stamps[16];
#endif

template <bool BP>
__global__ __launch_bounds__(512) void flash_attention_kernel(int m, int nk, const float* __restrict__ Q, long ldq,
                                                               const float* __restrict__ K, long ldk,
                                                               const float* __restrict__ V, long ldv, float scale,
                                                               float* __restrict__ O, long ldo,
                                                               const float* __restrict__ Wbp, const float* __restrict__ bbp) {
    extern __shared__ __attribute__((aligned(16))) float lds[];       // [2 * FA_STAGE] (+ W [64][256] when BP)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // XCD-aware (cloud, query block): workgroup ids go round-robin over the 8 XCDs, so the query blocks of ONE cloud (consecutive ids)
    // landed on all eight and every L2 fetched that cloud's K | V (counter traffic 3x the algorithmic bytes).  With a cloud count that
    // is a multiple of 8 each XCD gets whole clouds.
    int cloud = blockIdx.y, qblk = blockIdx.x;
    if ((gridDim.y & 7u) == 0) {
        const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y, slot = lin >> 3;
        cloud = (int)((lin & 7u) * (gridDim.y >> 3) + slot / gridDim.x);
        qblk = (int)(slot % gridDim.x);
    }
    const float* __restrict__ kb = K + (size_t)cloud * nk * ldk;
    const float* __restrict__ vb = V + (size_t)cloud * nk * ldv;
    const int ntile = nk / FA_TK;

    if (wave >= 4) {
        // ---------------------------------------------------------------------------------- loader waves
        // tile: 32 keys x (64 K + 64 V) floats = 512 float4 of K and 512 of V; loader thread -> 2 + 2 float4
        const int tid = threadIdx.x - 256;
        float4 pk[2], pv[2];
        auto load_tile = [&](int k0) {
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int e = tid + it * 256, key = e >> 4, c4 = e & 15;
                pk[it] = *reinterpret_cast<const float4*>(kb + (size_t)(k0 + key) * ldk + c4 * 4);
                pv[it] = *reinterpret_cast<const float4*>(vb + (size_t)(k0 + key) * ldv + c4 * 4);
            }
        };
        auto store_tile = [&](int stage) {
            float* Kt = lds + stage * FA_STAGE;
            float* Vt = Kt + FA_TK * FA_LDK;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int e = tid + it * 256, key = e >> 4, c4 = e & 15;
                Kt[key * FA_LDK + c4 * 4 + 0] = pk[it].x;
                Kt[key * FA_LDK + c4 * 4 + 1] = pk[it].y;
                Kt[key * FA_LDK + c4 * 4 + 2] = pk[it].z;
                Kt[key * FA_LDK + c4 * 4 + 3] = pk[it].w;
                *reinterpret_cast<float4*>(&Vt[key * FA_D + c4 * 4]) = pv[it];
            }
        };
        load_tile(0);
        if constexpr (BP) {                                  // W [64][256] -> LDS behind the first tile's loads: 16 float4 per thread
            float* Wl = lds + 2 * FA_STAGE;
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                float4 w[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) w[u] = *reinterpret_cast<const float4*>(Wbp + (size_t)(tid + (h * 4 + u) * 256) * 4);
#pragma unroll
                for (int u = 0; u < 4; ++u) *reinterpret_cast<float4*>(Wl + (size_t)(tid + (h * 4 + u) * 256) * 4) = w[u];
            }
        }
        store_tile(0);
        if (ntile > 1) load_tile(FA_TK);
        __syncthreads();
        for (int t = 0; t < ntile; ++t) {
            if (t + 1 < ntile) {
                store_tile((t + 1) & 1);
                if (t + 2 < ntile) load_tile((t + 2) * FA_TK);
            }
            __syncthreads();
        }
        return;
    }

    // ------------------------------------------------------------------------------------------ MFMA waves
    const int qrow = qblk * 128 + wave * 32 + (lane & 31);
    const int kh = lane >> 5, li = lane & 31;
    const bool qok = qrow < m;
    const float* __restrict__ qp = Q + ((size_t)cloud * m + (qok ? qrow : 0)) * ldq;

    const float scale2 = scale * 1.4426950408889634f;  // logits in the log2 domain
    float qf[32];                                    // Q[q][2s + kh] * scale2
#pragma unroll
    for (int s4 = 0; s4 < 16; ++s4) {
        const float4 v = *reinterpret_cast<const float4*>(qp + s4 * 4);
        qf[2 * s4] = (kh ? v.y : v.x) * scale2;
        qf[2 * s4 + 1] = (kh ? v.w : v.z) * scale2;
    }

    f32x16 oacc[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[c][r] = 0.f;
    float mrun = -__builtin_inff(), lsum = 0.f;

    __syncthreads();
#ifdef FA_STAMPS   // tools/micro/attention_lab.hip: cycles per phase of one MFMA wave
    unsigned long long fa_s = 0, fa_sm = 0, fa_pv = 0, fa_bar = 0;
#define FA_T(v) __builtin_amdgcn_sched_barrier(0); const unsigned long long v = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0)
#else
#define FA_T(v)
#endif
    for (int t = 0; t < ntile; ++t) {
        FA_T(c0);
        const float* Kt = lds + (t & 1) * FA_STAGE;
        const float* Vt = Kt + FA_TK * FA_LDK;
        // S^T tile: 32 keys x 32 queries, d ascending
        f32x16 sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 32; ++s) sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(Kt[li * FA_LDK + 2 * s + kh], qf[s], sacc, 0, 0, 0);
        FA_T(c1);
        // online softmax for this lane's query over the tile's 32 keys (16 here, 16 in lane ^ 32), in the log2 domain: Q was
        // multiplied by scale * log2(e) when it was loaded, so the accumulators ARE the log2-logits and p = 2^(s - m) is one
        // v_exp_f32 per logit (tools/micro/attention_lab.hip: the softmax is 730 of a tile's 5200 cycles, 256 of them the 16
        // quarter-rate exponentials; the two products run at 2120 / 2205 cycles for 2048 of pipe time)
        float mx = fmaxf(fmaxf(sacc[0], sacc[1]), sacc[2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, sacc[r]), sacc[r + 1]);      // v_max3_f32
        mx = fmaxf(mx, sacc[15]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mnew = fmaxf(mrun, mx);
        const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sacc[r] = __builtin_amdgcn_exp2f(sacc[r] - mnew); rs += sacc[r]; }
        lsum = lsum * alpha + rs;
        mrun = mnew;
        if (__any(alpha != 1.0f)) {                  // after the first tiles the running maximum rarely moves: x * 1.0f is x
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[c][r] = oacc[c][r] * alpha;
        }
        FA_T(c2);
        // O^T += V^T . P^T : step r pairs key rows kr and kr + 4 (= the rows half 0 / half 1 hold in register r)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = (r & 3) + 8 * (r >> 2) + 4 * kh;
#pragma unroll
            for (int c = 0; c < 2; ++c)
                oacc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vt[key * FA_D + c * 32 + li], sacc[r], oacc[c], 0, 0, 0);
        }
        FA_T(c3);
        __syncthreads();
#ifdef FA_STAMPS
        { FA_T(c4); fa_s += c1 - c0; fa_sm += c2 - c1; fa_pv += c3 - c2; fa_bar += c4 - c3; }
#endif
    }
#ifdef FA_STAMPS
    if (blockIdx.x == 3 && blockIdx.y == 1 && lane == 0) {
        fa_stamps[wave * 4 + 0] = fa_s; fa_stamps[wave * 4 + 1] = fa_sm; fa_stamps[wave * 4 + 2] = fa_pv; fa_stamps[wave * 4 + 3] = fa_bar;
    }
#endif
    const float ltot = lsum + __shfl_xor(lsum, 32, 64);
    const float inv = 1.0f / ltot;
    float* __restrict__ op = O + ((size_t)cloud * m + (qok ? qrow : 0)) * ldo;
    if constexpr (!BP) {
        if (qok) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) op[c * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh] = oacc[c][r] * inv;
        }
    } else {
        const float* Wl = lds + 2 * FA_STAGE;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[c][r] = oacc[c][r] * inv;
#pragma unroll 1
        for (int ot = 0; ot < FA_BPN / 32; ot += 2) {
            f32x16 y0, y1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { y0[r] = 0.f; y1[r] = 0.f; }
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int d = c * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    y0 = __builtin_amdgcn_mfma_f32_32x32x2f32(Wl[d * FA_BPN + ot * 32 + li], oacc[c][r], y0, 0, 0, 0);
                    y1 = __builtin_amdgcn_mfma_f32_32x32x2f32(Wl[d * FA_BPN + ot * 32 + 32 + li], oacc[c][r], y1, 0, 0, 0);
                }
            if (qok) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int o0 = ot * 32 + 8 * g + 4 * kh;
                    const float4 b0 = *reinterpret_cast<const float4*>(bbp + o0), b1 = *reinterpret_cast<const float4*>(bbp + o0 + 32);
                    float4 v0, v1;
                    v0.x = fmaxf(y0[4 * g + 0] + b0.x, 0.f); v0.y = fmaxf(y0[4 * g + 1] + b0.y, 0.f);
                    v0.z = fmaxf(y0[4 * g + 2] + b0.z, 0.f); v0.w = fmaxf(y0[4 * g + 3] + b0.w, 0.f);
                    v1.x = fmaxf(y1[4 * g + 0] + b1.x, 0.f); v1.y = fmaxf(y1[4 * g + 1] + b1.y, 0.f);
                    v1.z = fmaxf(y1[4 * g + 2] + b1.z, 0.f); v1.w = fmaxf(y1[4 * g + 3] + b1.w, 0.f);
                    *reinterpret_cast<float4*>(op + o0) = v0;
                    *reinterpret_cast<float4*>(op + o0 + 32) = v1;
                }
            }
        }
    }
}

}  // namespace dispu

using namespace dispu;

// O[b, m, 64] = softmax(scale * Q.K^T) . V per cloud;  Q [b*m, 64], K/V [b*nk, 64] with row strides ld*.
// Requires d == 64, nk % 32 == 0, 16-byte aligned rows; returns hipErrorInvalidValue otherwise (caller falls back to
// dispu_linear(transb) -> dispu_softmax_rows -> dispu_linear).
DISPU_EXPORT int dispu_attention(int b, int m, int nk, int d, const float* Q, long ldq, const float* K, long ldk, const float* V,
                                 long ldv, float scale, float* O, long ldo, void* stream) {
    if (b < 0 || m <= 0 || nk <= 0 || d != 64 || (nk % 32) != 0 || (ldq & 3) || (ldk & 3) || (ldv & 3) ||
        ((((uintptr_t)Q) | ((uintptr_t)K) | ((uintptr_t)V)) & 15))
        return (int)hipErrorInvalidValue;
    if (b == 0) return 0;
    hipLaunchKernelGGL((flash_attention_kernel<false>), dim3((m + 127) / 128, b), dim3(512), 2 * FA_STAGE * sizeof(float),
                       (hipStream_t)stream, m, nk, Q, ldq, K, ldk, V, ldv, scale, O, ldo, (const float*)nullptr, (const float*)nullptr);
    return (int)hipGetLastError();
}

// Y[b, m, 256] = relu(softmax(scale * Q.K^T) . V . W + bias): the non-local cell with its output projection
// (PointNonLocalCell, Common/ops.py:326-343) in one launch.  W [64, 256] row-major, bias [256]; same shape rules as
// dispu_attention, ldy % 4 == 0, Y / W / bias 16-byte aligned.
DISPU_EXPORT int dispu_attention_project(int b, int m, int nk, int d, const float* Q, long ldq, const float* K, long ldk,
                                         const float* V, long ldv, float scale, const float* W, const float* bias, int n_out,
                                         float* Y, long ldy, void* stream) {
    if (b < 0 || m <= 0 || nk <= 0 || d != 64 || n_out != FA_BPN || (nk % 32) != 0 || (ldq & 3) || (ldk & 3) || (ldv & 3) ||
        (ldy & 3) || !W || !bias ||
        ((((uintptr_t)Q) | ((uintptr_t)K) | ((uintptr_t)V) | ((uintptr_t)W) | ((uintptr_t)bias) | ((uintptr_t)Y)) & 15))
        return (int)hipErrorInvalidValue;
    if (b == 0) return 0;
    constexpr size_t bytes = (size_t)(2 * FA_STAGE + FA_D * FA_BPN) * sizeof(float);
    static DevOnce attr;      
    if (attr.needed()) {
        DISPU_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(flash_attention_kernel<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        attr.done();
    }
    hipLaunchKernelGGL((flash_attention_kernel<true>), dim3((m + 127) / 128, b), dim3(512), bytes, (hipStream_t)stream, m, nk, Q, ldq,
                       K, ldk, V, ldv, scale, Y, ldy, W, bias);
    return (int)hipGetLastError();
}
