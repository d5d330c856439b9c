// EXPLORATORY (verdict round 1, item 9; never the driver's headline): fp32-accurate GEMM on the bf16 matrix pipe by operand
// splitting.  Every fp32 operand is written as the sum of three bf16 numbers, x = x1 + x2 + x3 (x1 = bf16(x), x2 = bf16(x - x1),
// x3 = bf16(x - x1 - x2): 24 mantissa bits in total), and the product a.b is the six partial products of order <= 2
//   a3 b1 + a2 b2 + a1 b3 + a2 b1 + a1 b2 + a1 b1      (smallest first)
// each an exact bf16 x bf16 product accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  Six bf16 MFMAs do the work of sixteen
// fp32 ones (32x32x16 per 32 cycles against 32x32x2 per 64), so the after_conv GEMM (32768 x 2048 x 256, 278 us at 81 % of the
// fp32 MFMA peak) is no longer bound by the fp32 matrix rate.  The result is NOT the ascending-k fmaf chain of dispu_linear
// (different rounding points), so this path is only offered where the generator is tolerance-checked anyway: after the last
// index decision (the refinement branch, `fine` <= 1e-5 against the oracle).  bench.py --split-bf16 reports it BESIDE the
// strict-fp32 line with its own dtype string.
//
// Weights are static at inference: dispu_bf16x3_split_weights writes their three planes once, transposed to [N][K] (k
// contiguous), so the B tiles are plain 16-byte copies into LDS; the activations are split on the fly while they are staged.
// 128 x 128 x 32 tiles, 4 waves (2 x 2), two LDS stages of 6 planes (120 KB), register-prefetched.
//
// MEASURED (MI355X, round 2): as accurate as the fp32 MFMA GEMM (max error 1.7e-7 of sum |a||b| against 1.9e-7), but 317 - 390 us for
// the after_conv shape against 277 us for the wave-specialised fp32 kernel - a NEGATIVE result for this simple pipeline: with
// the matrix work cut to 6/16 the loop is bound by operand bytes and the on-the-fly split, see the comment in the kernel.  Kept
// as an opt-in, tested path; not used by default anywhere.
#include "common.h"

#include <cstdlib>

namespace dispu {

typedef __bf16 x3_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 x3_bf16x4 __attribute__((ext_vector_type(4)));
typedef float x3_f32x16 __attribute__((ext_vector_type(16)));

constexpr int X3_BM = 128, X3_BN = 128, X3_BK = 32, X3_PITCH = 40;
constexpr int X3_PLANE = 128 * X3_PITCH;                 // halves per operand plane
constexpr int X3_STAGE = 6 * X3_PLANE;                   // A planes 0..2, B planes 0..2
constexpr size_t X3_LDS_BYTES = (size_t)2 * X3_STAGE * sizeof(__bf16);

__device__ __forceinline__ void x3_split(float x, __bf16& a, __bf16& b, __bf16& c) {
    a = (__bf16)x;
    const float r = x - (float)a;                         // exact: x and bf16(x) agree in the leading 8 bits
    b = (__bf16)r;
    c = (__bf16)(r - (float)b);
}

// planes[p][n][k] (bf16) of W [K][N] (row stride ldw)
__global__ void bf16x3_split_weights_kernel(int K, int N, const float* __restrict__ W, long ldw, __bf16* __restrict__ planes) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)K * N) return;
    const int n = (int)(e / K), k = (int)(e % K);
    __bf16 a, b, c;
    x3_split(W[(long)k * ldw + n], a, b, c);
    planes[e] = a;
    planes[(long)K * N + e] = b;
    planes[2l * K * N + e] = c;
}

struct X3Args {
    int M, N, K;
    const float* X; long ldx;
    const __bf16* Wp;                                     // [3][N][K]
    const float* bias; int act;
    float* Y; long ldy;
    const float* R1; long ldr1;
    const float* R2; long ldr2;
};

__global__ __launch_bounds__(256) void gemm_bf16x3_kernel(X3Args a) {
    extern __shared__ __attribute__((aligned(16))) __bf16 x3_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, kq = lane >> 5;
    const int tiles_n = a.N / X3_BN;
    const int m0 = (blockIdx.x / tiles_n) * X3_BM, n0 = (blockIdx.x % tiles_n) * X3_BN;
    const long plane_sz = (long)a.N * a.K;

    x3_f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[1][4];
    x3_bf16x8 rb[1][6];
    auto fetch = [&](int k0, float4 (&fa_)[4], x3_bf16x8 (&fb_)[6]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int s = tid + 256 * u;
            fa_[u] = *reinterpret_cast<const float4*>(a.X + (long)(m0 + (s >> 3)) * a.ldx + k0 + (s & 7) * 4);
        }
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int s = tid + 256 * u, p = s >> 9, rem = s & 511;
            fb_[u] = *reinterpret_cast<const x3_bf16x8*>(a.Wp + p * plane_sz + (long)(n0 + (rem >> 2)) * a.K + k0 + (rem & 3) * 8);
        }
    };
    auto stage = [&](__bf16* st, const float4 (&fa_)[4], const x3_bf16x8 (&fb_)[6]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int s = tid + 256 * u;
            x3_bf16x4 h0, h1, h2;
            const float v[4] = {fa_[u].x, fa_[u].y, fa_[u].z, fa_[u].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { __bf16 p0, p1, p2; x3_split(v[e], p0, p1, p2); h0[e] = p0; h1[e] = p1; h2[e] = p2; }
            __bf16* d = st + (s >> 3) * X3_PITCH + (s & 7) * 4;
            *reinterpret_cast<x3_bf16x4*>(d) = h0;
            *reinterpret_cast<x3_bf16x4*>(d + X3_PLANE) = h1;
            *reinterpret_cast<x3_bf16x4*>(d + 2 * X3_PLANE) = h2;
        }
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int s = tid + 256 * u, p = s >> 9, rem = s & 511;
            *reinterpret_cast<x3_bf16x8*>(st + (3 + p) * X3_PLANE + (rem >> 2) * X3_PITCH + (rem & 3) * 8) = fb_[u];
        }
    };
    auto mma = [&](const __bf16* st) {
#pragma unroll
        for (int ks = 0; ks < X3_BK / 16; ++ks) {
            x3_bf16x8 fa[2][3], fb[2][3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    fa[i][p] = *reinterpret_cast<const x3_bf16x8*>(st + p * X3_PLANE + (wm * 64 + i * 32 + li) * X3_PITCH + ks * 16 + kq * 8);
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    fb[j][p] = *reinterpret_cast<const x3_bf16x8*>(st + (3 + p) * X3_PLANE + (wn * 64 + j * 32 + li) * X3_PITCH + ks * 16 + kq * 8);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    x3_f32x16 c = acc[i][j];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fb[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][0], c, 0, 0, 0);
                    acc[i][j] = c;
                }
        }
    };

    // One slab of register prefetch.  Two slabs ahead was measured and is SLOWER (359 vs 317 us at 32768 x 2048 x 256): the loop is
    // not waiting for load latency but for operand bytes - six 2-byte planes per element pair make 1.3 GB of L2 -> LDS traffic
    // per launch against 0.8 GB in the fp32 kernel, whose 128 x 256 tiles also read the activations once.
    const int nslab = a.K / X3_BK;
    fetch(0, ra[0], rb[0]);
    stage(x3_lds, ra[0], rb[0]);
    __syncthreads();
    for (int t = 0; t < nslab; ++t) {
        if (t + 1 < nslab) fetch((t + 1) * X3_BK, ra[0], rb[0]);
        mma(x3_lds + (t & 1) * X3_STAGE);
        if (t + 1 < nslab) stage(x3_lds + ((t + 1) & 1) * X3_STAGE, ra[0], rb[0]);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + li;
            const float bv = a.bias ? a.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kq;
                float v = acc[i][j][r] + bv;
                if (a.act == 1) v = fmaxf(v, 0.f);
                if (a.R1) v += a.R1[(long)row * a.ldr1 + col];
                if (a.R2) v += a.R2[(long)row * a.ldr2 + col];
                a.Y[(long)row * a.ldy + col] = v;
            }
        }
}


// ---- round 4: the same products, wave-specialised -------------------------------------------------------------------------------
// 128 x 256 tile (the activations are read once), K slabs of 32, 4 MFMA waves (64 x 128 each = 2 x 4 blocks of 32 x 32) + 4 loader
// waves; one workgroup per CU; two LDS stages of A [plane][128 rows][32 k] and B [plane][256 cols][32 k] bf16 (144 KB).
// * W planes are stored SLAB-MAJOR by dispu_bf16x3_split_weights for this kernel: [N / 256][K / 32][plane][256 cols][32 k], every
//   slab the exact image of a B stage (swizzle included), so a loader lane copies twelve 16-byte pieces per slab from fully
//   contiguous memory.  (With [plane][n][k] planes a slab touched 768 cache lines for 32 bytes each, 4 KB apart: the loader side alone
//   took 1.0 - 1.2 us per 16-k slab whether it used LDS-DMA or registers, and the launch 199 - 240 us.)
// * X: a loader lane owns (row, k quarter) of rows r and r + 64: two float4 each (four lanes = one 128-byte line), split into the
//   three bf16 terms in registers and written as one 16-byte chunk per plane.
// * Loads travel through REGISTERS, two slabs in flight per loader wave (128 VGPRs); the compiler counts vmcnt.
// A row of a stage is 64 bytes = four 16-byte chunks (8 k each); chunk c of row r sits at position c ^ ((r >> 2) & 3): the sixteen lanes
// of a ds_read_b128 lane group ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...) then touch sixteen different 16-byte slots of the
// 256-byte bank row (rows that agree mod 4 differ in (r >> 2) & 3 within every group).
// Per slab and MFMA wave: 36 ds_read_b128 feed 96 v_mfma_f32_32x32x16_bf16 (6 per block and 16 k: the partial products of order <= 2,
// smallest first) = 3072 matrix-pipe cycles; the fp32 kernel needs 8192 for the same 32 k.
constexpr int W3_BM = 128, W3_BN = 256, W3_BK = 32;
constexpr int W3_B_STAGE = 3 * W3_BN * W3_BK;            // halves (48 KB)
constexpr int W3_A_STAGE = 3 * W3_BM * W3_BK;            // halves (24 KB)
constexpr size_t W3_LDS_BYTES = (size_t)2 * (W3_B_STAGE + W3_A_STAGE) * 2;   // 144 KB

__device__ __host__ inline bool x3_ws_shape(int K, int N) { return N % W3_BN == 0 && K % W3_BK == 0 && K >= 4 * W3_BK; }

// slab-major planes for gemm_bf16x3_ws_kernel: element (k, n) of plane p -> [n / 256][k / 32][p][n % 256][chunk ^ swizzle][k % 8]
__global__ void bf16x3_split_weights_slab_kernel(int K, int N, const float* __restrict__ W, long ldw, __bf16* __restrict__ planes) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)K * N) return;
    const int k = (int)(e / N), n = (int)(e % N);                    // consecutive threads: consecutive n (coalesced reads of W)
    __bf16 q[3];
    x3_split(W[(long)k * ldw + n], q[0], q[1], q[2]);
    const int nt = n / W3_BN, nl = n % W3_BN, t = k / W3_BK, kl = k % W3_BK;
    const int pc = (kl >> 3) ^ ((nl >> 2) & 3);
    const long base = ((long)nt * (K / W3_BK) + t) * W3_B_STAGE + (long)nl * W3_BK + pc * 8 + (kl & 7);
#pragma unroll
    for (int p = 0; p < 3; ++p) planes[base + (long)p * W3_BN * W3_BK] = q[p];
}

__global__ __launch_bounds__(512) void gemm_bf16x3_ws_kernel(X3Args a) {
    extern __shared__ __attribute__((aligned(16))) __bf16 w3_lds[];
    __bf16* Bst = w3_lds;
    __bf16* Ast = w3_lds + 2 * W3_B_STAGE;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m0 = blockIdx.y * W3_BM, n0 = blockIdx.x * W3_BN;
    const int ntile = a.K / W3_BK;

    if (wave >= 4) {
        // ------------------------------------------------------------------------------------------------ loader waves
        const int ltid = threadIdx.x - 256;
        // X: rows (ltid >> 2) and (ltid >> 2) + 64, k quarter ltid & 3 (8 values = two float4)
        const int arow = ltid >> 2, aq = ltid & 3;
        const float* px = a.X + (size_t)(m0 + arow) * a.ldx + aq * 8;
        const size_t px64 = (size_t)64 * a.ldx;
        const int adst = arow * W3_BK + ((aq ^ ((arow >> 2) & 3)) * 8);          // (row + 64: same swizzle bits)
        // W: twelve 16-byte pieces per slab, piece j of this lane = halves [(j * 256 + ltid) * 8, + 8) of the slab image
        const __bf16* pw = a.Wp + (size_t)blockIdx.x * ntile * W3_B_STAGE + (size_t)ltid * 8;
        struct Slab { float4 x[4]; x3_bf16x8 w[12]; };
        auto load = [&](int t, Slab& r) {
            t = min(t, ntile - 1);                                   // past the end: a harmless repeat of the last slab, never stored
#ifdef X3_NOLOAD
            if (a.M > 0) return;
#endif
            const float* xs = px + (size_t)t * W3_BK;
            r.x[0] = *reinterpret_cast<const float4*>(xs);
            r.x[1] = *reinterpret_cast<const float4*>(xs + 4);
            r.x[2] = *reinterpret_cast<const float4*>(xs + px64);
            r.x[3] = *reinterpret_cast<const float4*>(xs + px64 + 4);
            const __bf16* ws = pw + (size_t)t * W3_B_STAGE;
#pragma unroll
            for (int j = 0; j < 12; ++j) r.w[j] = *reinterpret_cast<const x3_bf16x8*>(ws + j * 2048);
        };
        auto store = [&](int t, const Slab& r) {
#ifdef X3_NOLOAD
            if (a.M > 0) return;
#endif
            __bf16* d = Ast + (t & 1) * W3_A_STAGE + adst;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const float4 v0 = r.x[2 * u], v1 = r.x[2 * u + 1];
                const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                x3_bf16x8 h0, h1, h2;
#ifdef X3_NOSPLIT
#pragma unroll
                for (int e = 0; e < 8; ++e) { h0[e] = (__bf16)v[e]; h1[e] = h0[e]; h2[e] = h0[e]; }
#else
#pragma unroll
                for (int e = 0; e < 8; ++e) { __bf16 q0, q1, q2; x3_split(v[e], q0, q1, q2); h0[e] = q0; h1[e] = q1; h2[e] = q2; }
#endif
                *reinterpret_cast<x3_bf16x8*>(d + u * 64 * W3_BK) = h0;
                *reinterpret_cast<x3_bf16x8*>(d + u * 64 * W3_BK + W3_BM * W3_BK) = h1;
                *reinterpret_cast<x3_bf16x8*>(d + u * 64 * W3_BK + 2 * W3_BM * W3_BK) = h2;
            }
            __bf16* b = Bst + (t & 1) * W3_B_STAGE + ltid * 8;
#pragma unroll
            for (int j = 0; j < 12; ++j) *reinterpret_cast<x3_bf16x8*>(b + j * 2048) = r.w[j];
        };
        Slab r0, r1;                                                 // even slabs travel in r0, odd slabs in r1
        load(0, r0); load(1, r1);
        store(0, r0);
        load(2, r0);
        __syncthreads();
        // iteration t: slab t + 1 -> LDS (its stage was last read during slab t - 1), then request slab t + 3 into the freed registers
        for (int t = 0; t < ntile; t += 2) {
            if (t + 1 < ntile) store(t + 1, r1);
            load(t + 3, r1);
            __syncthreads();
            if (t + 1 < ntile) {
                if (t + 2 < ntile) store(t + 2, r0);
                load(t + 4, r0);
                __syncthreads();
            }
        }
        return;
    }

    // ---------------------------------------------------------------------------------------------------- MFMA waves
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, kq = lane >> 5;
    x3_f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // fragment (block, k16 step ks): chunk c = 2 ks + kq of the block's row -> position c ^ ((row >> 2) & 3)
    int aoff[2][2], boff[4][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) { const int row = wm * 64 + i * 32 + li; aoff[i][ks] = row * W3_BK + (((2 * ks + kq) ^ ((row >> 2) & 3)) * 8); }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) { const int col = wn * 128 + j * 32 + li; boff[j][ks] = col * W3_BK + (((2 * ks + kq) ^ ((col >> 2) & 3)) * 8); }
    __syncthreads();
    for (int t = 0; t < ntile; ++t) {
        const __bf16* As = Ast + (t & 1) * W3_A_STAGE;
        const __bf16* Bs = Bst + (t & 1) * W3_B_STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            x3_bf16x8 fa[2][3], fb[4][3];
            // fragments in the order the products below need them: (a3, b1) first
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i][2] = *reinterpret_cast<const x3_bf16x8*>(As + 2 * W3_BM * W3_BK + aoff[i][ks]);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j][0] = *reinterpret_cast<const x3_bf16x8*>(Bs + boff[j][ks]);
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i][1] = *reinterpret_cast<const x3_bf16x8*>(As + W3_BM * W3_BK + aoff[i][ks]);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j][1] = *reinterpret_cast<const x3_bf16x8*>(Bs + W3_BN * W3_BK + boff[j][ks]);
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i][0] = *reinterpret_cast<const x3_bf16x8*>(As + aoff[i][ks]);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j][2] = *reinterpret_cast<const x3_bf16x8*>(Bs + 2 * W3_BN * W3_BK + boff[j][ks]);
            // six rounds over the eight blocks: consecutive MFMAs never share an accumulator
#define W3_ROUND(PA, PB)                                                                                             \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                            \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                            \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][PA], fb[j][PB], acc[i][j], 0, 0, 0);
#ifdef X3_NOMFMA
            W3_ROUND(2, 0)
            if (a.M < 0) { W3_ROUND(1, 1) W3_ROUND(0, 2) W3_ROUND(1, 0) W3_ROUND(0, 1) W3_ROUND(0, 0) }
#else
            W3_ROUND(2, 0) W3_ROUND(1, 1) W3_ROUND(0, 2) W3_ROUND(1, 0) W3_ROUND(0, 1) W3_ROUND(0, 0)
#endif
#undef W3_ROUND
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = n0 + wn * 128 + j * 32 + li;
            const float bv = a.bias ? a.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kq;
                float v = acc[i][j][r] + bv;
                if (a.act == 1) v = fmaxf(v, 0.f);
                if (a.R1) v += a.R1[(long)row * a.ldr1 + col];
                if (a.R2) v += a.R2[(long)row * a.ldr2 + col];
                a.Y[(long)row * a.ldy + col] = v;
            }
        }
}

// ---- round 6: the same products as operand STREAMS (the design of csrc/linear_bf16_stream.hip) ------------------------------------
// gemm_bf16x3_ws_kernel spends its time in the loader waves: 1.2 M fp32 values per launch and CU are split into three bf16 terms by
// four waves that also carry every operand byte through registers (~200 us for the after_conv shape, 6 MFMAs per 16 k notwithstanding).
// Here nothing is carried: a 32-k slab of X (128 rows x 128 B of fp32) and the slab image of the three W planes (48 KB, already in its
// swizzled stage layout: dispu_bf16x3_split_weights) go global -> LDS by DMA (global_load_lds_dwordx4), two stages, one slab in flight;
// ALL eight waves compute (4 x 2 wave grid: 32 rows x 128 columns each) and a wave splits its own A fragment -- 8 consecutive k of one
// row, two ds_read_b128 -- on the VALU, in the shadow of the 24 MFMAs the fragment feeds.  The X bank swizzle is applied on the global
// side (the lane that lands at 16-byte position p of row r fetches chunk p ^ (r & 7)).  Same products in the same order per
// accumulator as the kernels above: bit-identical results.
constexpr int S3_BM = 128, S3_BN = 256, S3_BK = 32, S3_NST = 2;
constexpr int S3_A_BYTES = S3_BM * S3_BK * 4;             // 16 KB
constexpr int S3_B_BYTES = W3_B_STAGE * 2;                // 48 KB
constexpr int S3_STAGE = S3_A_BYTES + S3_B_BYTES;         // 64 KB
constexpr size_t S3_LDS_BYTES = (size_t)S3_NST * S3_STAGE;
static_assert(S3_BN == W3_BN && S3_BK == W3_BK, "the stream kernel reads the wave-specialised kernel's slab-major planes");

__global__ __launch_bounds__(512) void gemm_bf16x3_stream_kernel(X3Args a) {
    extern __shared__ __attribute__((aligned(16))) char s3_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, kq = lane >> 5;
    const int m0 = blockIdx.y * S3_BM, n0 = blockIdx.x * S3_BN;
    const int nt = a.K / S3_BK;

    // this wave's 1 KB pieces of a slab: X pieces wave, wave + 8 (8 rows x 128 B each), W pieces wave + 8 u, u < 6 (contiguous)
    const char* gx[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int q = wave + 8 * u, r = 8 * q + (lane >> 3), p = lane & 7;
        gx[u] = reinterpret_cast<const char*>(a.X + (size_t)(m0 + r) * a.ldx + 4 * (p ^ (r & 7)));
    }
    const char* gw = reinterpret_cast<const char*>(a.Wp + (size_t)blockIdx.x * nt * W3_B_STAGE) + (size_t)wave * 1024 + lane * 16;
    auto issue = [&](int t) {
        char* st = s3_lds + (t & 1) * S3_STAGE;
#ifdef S3_NOX                                                 // lab (tools/debug/x3_lab.py): only the first two slabs' X / W pieces are fetched --
        if (t < 2)                                            // wrong results, the time that is left is what the OTHER operand's stream costs
#endif
#pragma unroll
        for (int u = 0; u < 2; ++u) {   // (float pointers: with char* operands to the DMA builtin the host pass silently drops the kernel's stub)
            const float* src = reinterpret_cast<const float*>(gx[u] + (size_t)t * (S3_BK * 4));
            float* dst = reinterpret_cast<float*>(st + (wave + 8 * u) * 1024);
            __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
#ifdef S3_NOW
        if (t < 2)
#endif
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const float* src = reinterpret_cast<const float*>(gw + (size_t)t * S3_B_BYTES + u * 8192);
            float* dst = reinterpret_cast<float*>(st + S3_A_BYTES + (wave + 8 * u) * 1024);
            __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };

    x3_f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // fragment addresses inside a stage (bytes)
    const int arow = wm * 32 + li, a_row = arow * 128, a_sw = arow & 7;
    int boff[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int col = wn * 128 + j * 32 + li;
            boff[j][ks] = S3_A_BYTES + 2 * (col * W3_BK + (((2 * ks + kq) ^ ((col >> 2) & 3)) * 8));
        }
    constexpr int PLANE = W3_BN * W3_BK * 2;                  // bytes between the planes of a stage

    // The split of a fragment on the VALU, two values per step, as explicit instructions: v_cvt_pk_bf16_f32 packs (first, second) value into
    // (low, high) half -- the element order of the MFMA operand, no permutes -- and the subtractions stay single v_sub_f32 (left to the
    // compiler they are SLP-packed into v_pk_add_f32, which costs ~25 extra cycles apiece next to MFMAs: MI355X_MICROARCH.md).  Same
    // arithmetic as x3_split, element for element.
    typedef unsigned int s3_u32x4 __attribute__((ext_vector_type(4)));
    auto cvt_pk = [](float x0, float x1) -> unsigned int {
        unsigned int r;
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(x0), "v"(x1));
        return r;
    };
    auto sub = [](float x, float y) -> float {
        float r;
        asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
        return r;
    };
    auto split8 = [&](const float4& lo, const float4& hi, x3_bf16x8 (&fa)[3]) {
        const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        s3_u32x4 p0, p1, p2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned int q0 = cvt_pk(v[2 * e], v[2 * e + 1]);
            const float r0 = sub(v[2 * e], __uint_as_float(q0 << 16)), r1 = sub(v[2 * e + 1], __uint_as_float(q0 & 0xffff0000u));
            const unsigned int q1 = cvt_pk(r0, r1);
            const float s0 = sub(r0, __uint_as_float(q1 << 16)), s1 = sub(r1, __uint_as_float(q1 & 0xffff0000u));
            p0[e] = q0; p1[e] = q1; p2[e] = cvt_pk(s0, s1);
        }
        fa[0] = __builtin_bit_cast(x3_bf16x8, p0);
        fa[1] = __builtin_bit_cast(x3_bf16x8, p1);
        fa[2] = __builtin_bit_cast(x3_bf16x8, p2);
    };
    const int a_off[2][2] = {{a_row + (((2 * kq) ^ a_sw) << 4), a_row + (((2 * kq + 1) ^ a_sw) << 4)},
                             {a_row + (((4 + 2 * kq) ^ a_sw) << 4), a_row + (((5 + 2 * kq) ^ a_sw) << 4)}};

    issue(0);
    for (int t = 0; t < nt; ++t) {
        // slab t has landed (this wave's pieces: vmcnt, everybody's: barrier); everybody has finished reading slab t - 1, whose stage
        // slab t + 1 may now overwrite
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        if (t + 1 < nt) issue(t + 1);
        const char* st = s3_lds + (t & 1) * S3_STAGE;
        // k-step 0: its A fragment is read and split in the open (nothing of this slab could be read before the barrier); k-step 1's is
        // read right away and split between the MFMAs of step 0
        x3_bf16x8 fa0[3], fa1[3], fb[4][3];
        {
            const float4 lo = *reinterpret_cast<const float4*>(st + a_off[0][0]), hi = *reinterpret_cast<const float4*>(st + a_off[0][1]);
            split8(lo, hi, fa0);
        }
#ifndef S3_NOSCHED
        __builtin_amdgcn_sched_barrier(0);
#endif
        const float4 lo1 = *reinterpret_cast<const float4*>(st + a_off[1][0]), hi1 = *reinterpret_cast<const float4*>(st + a_off[1][1]);
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j][p] = *reinterpret_cast<const x3_bf16x8*>(st + p * PLANE + boff[j][0]);
        // six rounds over the four blocks, smallest partial products first: consecutive MFMAs never share an accumulator
#define S3_ROUND(FA, PA, PB)                                                                                         \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                \
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[PA], fb[j][PB], acc[j], 0, 0, 0);
        S3_ROUND(fa0, 2, 0) S3_ROUND(fa0, 1, 1) S3_ROUND(fa0, 0, 2) S3_ROUND(fa0, 1, 0) S3_ROUND(fa0, 0, 1) S3_ROUND(fa0, 0, 0)
        split8(lo1, hi1, fa1);
#ifndef S3_NOSCHED
        // pin: the 12 B reads in front, then every MFMA of step 0 followed by two of the 44 split instructions of step 1
        __builtin_amdgcn_sched_group_barrier(0x100, 14, 0);
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j][p] = *reinterpret_cast<const x3_bf16x8*>(st + p * PLANE + boff[j][1]);
        S3_ROUND(fa1, 2, 0) S3_ROUND(fa1, 1, 1) S3_ROUND(fa1, 0, 2) S3_ROUND(fa1, 1, 0) S3_ROUND(fa1, 0, 1) S3_ROUND(fa1, 0, 0)
#undef S3_ROUND
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = n0 + wn * 128 + j * 32 + li;
        const float bv = a.bias ? a.bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kq;
            float v = acc[j][r] + bv;
            if (a.act == 1) v = fmaxf(v, 0.f);
            if (a.R1) v += a.R1[(long)row * a.ldr1 + col];
            if (a.R2) v += a.R2[(long)row * a.ldr2 + col];
            a.Y[(long)row * a.ldy + col] = v;
        }
    }
}

}  // namespace dispu

using namespace dispu;

// the wave-specialised kernel and its slab-major planes wherever the shape allows (round 2's single-role kernel otherwise)
static bool x3_use_ws() { return true; }
// which kernel takes the slab-major shapes: 1 = round 4's wave-specialised one (the default), 0 = the streaming kernel of round 6; same planes,
// bit-identical results.  On uniform random operands the streaming kernel is the faster one by a few per cent (205 - 216 vs 210 - 243 us at the
// after_conv shape, tools/debug/x3_lab.py); inside the generator step, on the local cell's real F' tensor, it is 34 us SLOWER (0.880 vs 0.846 ms
// per step on one stream, tools/debug/x3_step_ab.py) -- both sit at the chip's sustained bf16 matrix rate, see profiles/EXPERIMENTS.md
static int g_x3_kernel = 1;
DISPU_EXPORT void dispu_debug_x3_kernel(int which) { g_x3_kernel = which; }

// planes: 3 * K * N bf16 values (6 K N bytes), [plane][n][k]
DISPU_EXPORT int dispu_bf16x3_split_weights(int K, int N, const float* W, long ldw, void* planes, void* stream) {
    if (K <= 0 || N <= 0 || !W || !planes || ldw < N) return (int)hipErrorInvalidValue;
    const long total = (long)K * N;
    if (x3_use_ws() && x3_ws_shape(K, N)) {      // the wave-specialised kernel's slab-major layout (the same rule picks the kernel below)
        hipLaunchKernelGGL(bf16x3_split_weights_slab_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, K, N, W, ldw,
                           reinterpret_cast<__bf16*>(planes));
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(bf16x3_split_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, K, N, W, ldw,
                       reinterpret_cast<__bf16*>(planes));
    return (int)hipGetLastError();
}

// Y = R2 + R1 + act(X . W + bias) with W given as its split planes.  M % 128 == 0, N % 128 == 0, K % 32 == 0, X rows 16-byte
// aligned (ldx % 4 == 0); anything else returns hipErrorInvalidValue (the caller then uses dispu_linear).
DISPU_EXPORT int dispu_linear_bf16x3(int M, int K, int N, const float* X, long ldx, const void* planes, const float* bias, int act,
                                     float* Y, long ldy, const float* R1, long ldr1, const float* R2, long ldr2, void* stream) {
    if (M < 0 || K <= 0 || N <= 0 || (M % X3_BM) || (N % X3_BN) || (K % X3_BK) || (ldx & 3) || !X || !planes || !Y || (act != 0 && act != 1) ||
        ((((uintptr_t)X) | ((uintptr_t)planes)) & 15))
        return (int)hipErrorInvalidValue;
    if (M == 0) return 0;
    static DevOnce attr;      
    if (attr.needed()) {
        DISPU_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)X3_LDS_BYTES));
        attr.done();
    }
    X3Args a{M, N, K, X, ldx, reinterpret_cast<const __bf16*>(planes), bias, act, Y, ldy, R1, ldr1, R2, ldr2};
    if (x3_use_ws() && x3_ws_shape(K, N)) {
        static DevOnce attr2;
        if (attr2.needed()) {
            DISPU_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16x3_ws_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)W3_LDS_BYTES));
            attr2.done();
        }
        if (g_x3_kernel == 0 && (((uintptr_t)planes) & 15) == 0) {
            static DevOnce attr3;
            if (attr3.needed()) {
                DISPU_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16x3_stream_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)S3_LDS_BYTES));
                attr3.done();
            }
            hipLaunchKernelGGL(gemm_bf16x3_stream_kernel, dim3((unsigned)(N / S3_BN), (unsigned)(M / S3_BM)), dim3(512), S3_LDS_BYTES, (hipStream_t)stream, a);
            return (int)hipGetLastError();
        }
        hipLaunchKernelGGL(gemm_bf16x3_ws_kernel, dim3((unsigned)(N / W3_BN), (unsigned)(M / W3_BM)), dim3(512), W3_LDS_BYTES, (hipStream_t)stream, a);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(gemm_bf16x3_kernel, dim3((unsigned)((M / X3_BM) * (N / X3_BN))), dim3(256), X3_LDS_BYTES, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}
