// Per-point dense layers (the reference's 1x1 conv1d/conv2d, Common/tf_util.py:52-185) and the
// batched matmuls of PointNonLocalCell (Common/ops.py:326,339) as ONE fp32 MFMA GEMM for gfx950:
//
//   Y[z][m, n] = res2 + res1 + act( chain_k X[z][m, k] * W[z][k, n]  + bias[n] )
//
// * v_mfma_f32_32x32x2_f32: exact fp32, and bit-for-bit the ascending-k fmaf chain the oracle pins
//   (oracle/mlp_oracle.c) -- no split-K, no reassociation, so features feeding the k-NN stages are
//   bit-reproducible.
// * Operands are addressed as (pointer, row stride): inputs can be column slices of a wider
//   activation buffer and outputs land directly inside concatenation buffers (the reference
//   materialises every tf.concat).
// * TRANSB reads W as [n][k] (k contiguous): Q.K^T without a transpose pass.
// * Block = 4 waves (2x2), LDS tiles A[BK][BM+1] / B[BK][BN+4], register prefetch of the next
//   K-tile while the MFMAs of the current one run.  Epilogue fuses bias, ReLU and two residual adds.
#include "common.h"

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
};

constexpr int LIN_BK = 32;

// BM x BN block tile; 4 waves arranged 2 x 2; each wave owns (BM/2) x (BN/2) = TM x TN 32x32 MFMA tiles.
template <int BM, int BN, bool TRANSB>
__global__ __launch_bounds__(256) void linear_mfma_kernel(LinearArgs a) {
    constexpr int BK = LIN_BK;
    constexpr int LDA = BM + 1;   // A tile stored k-major [BK][BM+1]: conflict-free b32 frag reads and writes
    constexpr int LDB = TRANSB ? BN + 1 : BN + 4;
    constexpr int TM = BM / 64, TN = BN / 64;          // 32x32 tiles per wave
    constexpr int A_F4 = BM * BK / 4 / 256;            // float4 loads per thread for the A tile
    constexpr int B_F4 = BN * BK / 4 / 256;
    static_assert(A_F4 >= 1 && B_F4 >= 1, "tile too small for 256 threads");
    __shared__ float As[BK * LDA];
    __shared__ float Bs[BK * LDB];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int z = blockIdx.z;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const float* __restrict__ X = a.X + (size_t)z * a.sx;
    const float* __restrict__ W = a.W + (size_t)z * a.sw;
    const int M = a.M, K = a.K, N = a.N;
    const long ldx = a.ldx, ldw = a.ldw;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 pa[A_F4], pb[B_F4];
    const bool x_vec = ((ldx & 3) == 0) && ((((uintptr_t)X) & 15) == 0);
    const bool w_vec = ((ldw & 3) == 0) && ((((uintptr_t)W) & 15) == 0);

    // rows-of-k loader (A always; B when TRANSB): element (row, k) with k contiguous in memory
    auto load_rowsk = [&](const float* __restrict__ P, long ld, bool vec, int row_base, int rows, int k0, int it) -> float4 {
        const int idx = tid + it * 256;
        const int r = idx / (BK / 4), kq = idx % (BK / 4);
        const int row = row_base + r, k = k0 + kq * 4;
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
        const int idx = tid + it * 256;
        const int r = idx / (BK / 4), kq = idx % (BK / 4);
        S[(kq * 4 + 0) * LD + r] = v.x;
        S[(kq * 4 + 1) * LD + r] = v.y;
        S[(kq * 4 + 2) * LD + r] = v.z;
        S[(kq * 4 + 3) * LD + r] = v.w;
    };
    // k-rows loader for B = W[k][n] (n contiguous)
    auto load_b = [&](int k0, int it) -> float4 {
        const int idx = tid + it * 256;
        const int kr = idx / (BN / 4), nq = idx % (BN / 4);
        const int k = k0 + kr, n = n0 + nq * 4;
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
    auto store_b = [&](float4 v, int it) {
        const int idx = tid + it * 256;
        const int kr = idx / (BN / 4), nq = idx % (BN / 4);
        *reinterpret_cast<float4*>(&Bs[kr * LDB + nq * 4]) = v;
    };

    const int ntile = (K + BK - 1) / BK;
#pragma unroll
    for (int it = 0; it < A_F4; ++it) pa[it] = load_rowsk(X, ldx, x_vec, m0, M, 0, it);
#pragma unroll
    for (int it = 0; it < B_F4; ++it) pb[it] = TRANSB ? load_rowsk(W, ldw, w_vec, n0, N, 0, it) : load_b(0, it);

    const int fi = lane & 31, fk = lane >> 5;
    for (int t = 0; t < ntile; ++t) {
        __syncthreads();   // previous tile fully consumed
#pragma unroll
        for (int it = 0; it < A_F4; ++it) store_rowsk(As, LDA, pa[it], it);
#pragma unroll
        for (int it = 0; it < B_F4; ++it) {
            if constexpr (TRANSB) store_rowsk(Bs, LDB, pb[it], it);
            else store_b(pb[it], it);
        }
        __syncthreads();
        if (t + 1 < ntile) {
            const int k0 = (t + 1) * BK;
#pragma unroll
            for (int it = 0; it < A_F4; ++it) pa[it] = load_rowsk(X, ldx, x_vec, m0, M, k0, it);
#pragma unroll
            for (int it = 0; it < B_F4; ++it) pb[it] = TRANSB ? load_rowsk(W, ldw, w_vec, n0, N, k0, it) : load_b(k0, it);
        }
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = As[(kk + fk) * LDA + wm * (BM / 2) + i * 32 + fi];
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = Bs[(kk + fk) * LDB + wn * (BN / 2) + j * 32 + fi];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    }

    // epilogue: C/D layout of the 32x32 tile: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    float* __restrict__ Y = a.Y + (size_t)z * a.sy;
    const float* __restrict__ R1 = a.R1 ? a.R1 + (size_t)z * a.sr1 : nullptr;
    const float* __restrict__ R2 = a.R2 ? a.R2 + (size_t)z * a.sr2 : nullptr;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * (BN / 2) + j * 32 + fi;
        if (col >= N) continue;
        const float bv = a.bias ? a.bias[col] : 0.f;
        const float sc = a.scale ? a.scale[col] : 1.f, sh = a.scale ? a.shift[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                if (row < M) {
                    float v = acc[i][j][r];
                    if (a.bias) v = v + bv;
                    if (a.scale) v = v * sc + sh;
                    if (a.act == 1) v = fmaxf(v, 0.f);
                    if (R1) v = v + R1[(size_t)row * a.ldr1 + col];
                    if (R2) v = v + R2[(size_t)row * a.ldr2 + col];
                    Y[(size_t)row * a.ldy + col] = v;
                }
            }
        }
    }
}

template <int BM, int BN>
static int launch_linear(const LinearArgs& a, int batch, bool transb, hipStream_t s) {
    dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, batch);
    if (transb) hipLaunchKernelGGL((linear_mfma_kernel<BM, BN, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((linear_mfma_kernel<BM, BN, false>), grid, dim3(256), 0, s, a);
    return (int)hipGetLastError();
}

}  // namespace dispu

using namespace dispu;

// Block-tile choice of dispu_linear as BM*1000 + BN (128128, 64128, 128064, 64064): large tiles when they still
// give >= 256 workgroups (one per CU), smaller ones otherwise.  Exported so a profiler can name the instantiation.
DISPU_EXPORT int dispu_linear_tile(int batch, int M, int N) {
    const long blocks_big = (long)((M + 127) / 128) * ((N + 127) / 128) * batch;
    if (N > 64) return blocks_big >= 256 ? 128128 : 64128;
    const long blocks_mid = (long)((M + 127) / 128) * batch;
    return blocks_mid >= 256 ? 128064 : 64064;
}

DISPU_EXPORT int dispu_linear_bn(int batch, int M, int K, int N, const float* X, long ldx, long sx, const float* W,
                                 long ldw, long sw, int transb, const float* bias, const float* scale, const float* shift,
                                 int act, float* Y, long ldy, long sy, const float* R1, long ldr1, long sr1,
                                 const float* R2, long ldr2, long sr2, void* stream);

// Y = R2 + R1 + act(X.W + bias); see include/dispu_hip.h for the argument contract.
DISPU_EXPORT int dispu_linear(int batch, int M, int K, int N, const float* X, long ldx, long sx, const float* W,
                              long ldw, long sw, int transb, const float* bias, int act, float* Y, long ldy, long sy,
                              const float* R1, long ldr1, long sr1, const float* R2, long ldr2, long sr2, void* stream) {
    if (batch < 0 || M < 0 || K <= 0 || N <= 0 || !X || !W || !Y || act < 0 || act > 1) return (int)hipErrorInvalidValue;
    if (batch == 0 || M == 0) return 0;
    return dispu_linear_bn(batch, M, K, N, X, ldx, sx, W, ldw, sw, transb, bias, nullptr, nullptr, act, Y, ldy, sy, R1, ldr1,
                           sr1, R2, ldr2, sr2, stream);
}

// dispu_linear with an inference-BatchNorm epilogue: Y = R2 + R1 + act( (X.W + bias) * scale + shift ).
DISPU_EXPORT int dispu_linear_bn(int batch, int M, int K, int N, const float* X, long ldx, long sx, const float* W,
                                 long ldw, long sw, int transb, const float* bias, const float* scale, const float* shift,
                                 int act, float* Y, long ldy, long sy, const float* R1, long ldr1, long sr1,
                                 const float* R2, long ldr2, long sr2, void* stream) {
    if (batch < 0 || M < 0 || K <= 0 || N <= 0 || !X || !W || !Y || act < 0 || act > 1 || ((scale == nullptr) != (shift == nullptr)))
        return (int)hipErrorInvalidValue;
    if (batch == 0 || M == 0) return 0;
    LinearArgs a{M, K, N, X, ldx, sx, W, ldw, sw, bias, Y, ldy, sy, R1, ldr1, sr1, R2, ldr2, sr2, act, scale, shift};
    hipStream_t s = (hipStream_t)stream;
    switch (dispu_linear_tile(batch, M, N)) {
        case 128128: return launch_linear<128, 128>(a, batch, transb != 0, s);
        case 64128: return launch_linear<64, 128>(a, batch, transb != 0, s);
        case 128064: return launch_linear<128, 64>(a, batch, transb != 0, s);
        default: return launch_linear<64, 64>(a, batch, transb != 0, s);
    }
}
