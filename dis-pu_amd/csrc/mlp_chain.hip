// Fused per-point MLP chains of the generator heads on gfx950 (fp32 MFMA):
//   coarse head: upshuffle conv2 (256 -> 128, ReLU; ops.py:1186-1192) -> coordinate_regressor fc_layer0 (128 -> 256, ReLU)
//                -> fc_layer1 (256 -> 64, ReLU) -> fc_layer2 (64 -> 3)                         (ops.py:1089-1104)
//   fine head:   PointShuffle2 aggregation (256 -> 256, ReLU; ops.py:1079-1083) -> fc_layer0 (256 -> 256, ReLU)
//                -> fc_layer1 (256 -> 64, ReLU) -> fc_layer2 (64 -> 3) -> coarse + sigmoid(.) - 0.5 (ops.py:1106-1108)
// As four separate GEMM launches each chain writes and re-reads three [rows, 64..256] activations through HBM and
// pays four prologues / epilogues on GEMMs that are only 4 - 32 K-slabs long (matrix pipe 46 % busy).  Here a
// workgroup keeps its 128 rows ON CHIP for the whole chain: the activation tile lives in LDS in the k-major layout
// the MFMA A operand wants, a layer's accumulators are written back over it (after a barrier) as the next layer's
// input, and only the weights stream in (8-row slabs through two LDS stages).  Wave-specialised like linear.hip:
// waves 0-3 fragment reads + MFMAs, waves 4-7 loads.  Every layer is the same k-ascending fmaf chain as
// dispu_linear / dispu_linear_small_n, so the results are bit-identical to the unfused launches.
#include "common.h"

namespace dispu {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ChainArgs {
    long rows;
    const float* X; long ldx;                       // [rows, K0]
    const float* W1; const float* b1;               // [K0, N1]
    const float* W2; const float* b2;               // [N1, N2]
    const float* W3; const float* b3;               // [N2, N3]
    const float* W4; const float* b4;               // [N3, 3]
    float* Y1; long ldy1;                           // optional copy of the first layer's output [rows, N1]
    float* Y2; long ldy2;                           // optional copies of the second / third layer's outputs and of the head's
    float* Y3; long ldy3;                           // pre-activation output [rows, 3] (training: the backward pass reads them)
    float* Z; long ldz;
    const float* R; long ldr;                       // mode 1: out = R + sigmoid(.) - 0.5
    float* out; long ldo;                           // [rows, 3]
    int mode;
    // XMODE 1 (round 4): the input tile is (X + X2) + X3, formed by the loader waves -- PointShuffle2's
    //   relu(after_conv) + skip + non-local (ops.py:1069-1075) no longer rides in the after_conv GEMM's epilogue, where 67 MB of
    //   residual reads sat behind the product loop of 256 workgroups that all finish at the same moment
    const float* X2 = nullptr; const float* X3 = nullptr;
    // XMODE 2 (round 4): the input tile is duplicate_up's conv1 output (ops.py:1152-1192) evaluated on the fly from the per-source-
    //   point product X = H [nclouds * n, K0]: row (cloud * up + r) * n + i = relu(fmaf(g[r][1], Wg[1], fmaf(g[r][0], Wg[0], H[cloud * n + i])) + bg)
    //   -- the arithmetic of dup_grid_kernel (csrc/mlp_misc.hip), whose [rows, 256] output and launch disappear
    const float* Wg = nullptr; const float* bg = nullptr; const float* grid = nullptr; int n = 0; int up = 0;
};

constexpr int MC_BM = 128, MC_KMAX = 256, MC_NMAX = 256;          // MC_BM: rows per workgroup of the large-batch variant (BM = 64 below
constexpr int MC_ACT = MC_KMAX * (MC_BM + 1);                     // 8192 rows: twice the workgroups, half the rows each); floats
// a weight slab is bk(N) = 2048 / N rows of N floats (8 / 16 / 32 rows for N = 256 / 128 / 64): every slab carries the
// same 32 MFMAs per wave, so the barrier cadence does not depend on the layer width
__host__ __device__ constexpr int mc_bk(int n) { return 2048 / n; }
constexpr int MC_WST = 32 * (64 + 4);                             // floats per weight stage (largest: 32 x 68)
constexpr int MC_HEAD = 64 * 3 + 4;
constexpr size_t MC_LDS_BYTES = (size_t)(MC_ACT + 2 * MC_WST + MC_HEAD) * sizeof(float);

// one layer on the MFMA waves: acc = act[0:K] . W (slabs g0 .. g0 + K/8 - 1 of the weight stream), then
// act[0:N] <- relu(acc + bias) (k-major), optionally also to global memory.
// MC_CLOCK (tools/micro/chain_lab.hip): cycle stamps of one MFMA wave - total and the part spent in the slab barriers.
// Measured: 2850 cycles per 32-MFMA slab (2048 pipe cycles), 100 - 350 of them in the barrier.  Tried and NOT kept: the weight
// stream by global_load_lds (3 stages), also with the input tile by DMA ([k/4][row] float4 layout, b128 reads): the barrier wait
// drops to ~100 cycles but the MFMA waves' own time per slab grows by 250 - 400, a net loss of 2 - 10 %.
#ifdef MC_CLOCK
__device__ unsigned long long mc_clock_ticks[4];
#define MC_WAIT_PARAM , unsigned long long& mc_wait
#define MC_WAIT_ARG , mc_wait_local
#else
#define MC_WAIT_PARAM
#define MC_WAIT_ARG
#endif
template <int K, int N, int BM>
__device__ __forceinline__ void chain_layer(float* act, const float* wst, int g0, const float* __restrict__ bias, float* __restrict__ Yg,
                                            long ldy, long row0, int wm, int wn, int fi, int fk MC_WAIT_PARAM) {
    constexpr int TNW = N / 64;                                   // 32-wide column blocks per wave (2 x 2 waves over BM x N)
    constexpr int LDW = N + 4, BK = mc_bk(N);
    constexpr int RT = BM / 64, MC_LDA = BM + 1;                  // 32-row tiles per wave (a wave owns BM / 2 rows)
    f32x16 acc[RT][TNW];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < TNW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int s = 0; s < K / BK; ++s) {
        const float* ws = wst + ((g0 + s) & 1) * MC_WST;
        const float* as = act + (s * BK) * MC_LDA;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float af[RT], bf[TNW];
#pragma unroll
            for (int i = 0; i < RT; ++i) af[i] = as[(kk + fk) * MC_LDA + wm * (BM / 2) + i * 32 + fi];
#pragma unroll
            for (int j = 0; j < TNW; ++j) bf[j] = ws[(kk + fk) * LDW + wn * (N / 2) + j * 32 + fi];
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int j = 0; j < TNW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
#ifdef MC_CLOCK
        const unsigned long long b0_ = __builtin_readcyclecounter();
        __syncthreads();
        mc_wait += __builtin_readcyclecounter() - b0_;
#else
        __syncthreads();
#endif
    }
    __syncthreads();                                              // every wave has finished reading this layer's input
#pragma unroll
    for (int j = 0; j < TNW; ++j) {
        const int n = wn * (N / 2) + j * 32 + fi;
        const float bv = bias[n];
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                const float v = fmaxf(acc[i][j][r] + bv, 0.f);
                act[n * MC_LDA + row] = v;
                if (Yg) Yg[(size_t)(row0 + row) * ldy + n] = v;
            }
    }
    __syncthreads();                                              // the next layer's input is complete
}

// STASH: also write the second / third layer's outputs and the head's pre-activation (training forward).  A template flag, not a
// run-time test: with the pointers as run-time arguments the inference launches lost 3 - 4 us each to the extra branches / address
// arithmetic in every layer's store loop.
template <int K0, int N1, int N2, int N3, int BM, bool STASH = false, int XMODE = 0>
__global__ __launch_bounds__(512) void mlp_chain_kernel(ChainArgs a) {
    constexpr int MC_LDA = BM + 1, XU = BM / 32;                  // XU: float4 loads per loader thread and 32-column input chunk
    static_assert(K0 <= MC_KMAX && N1 <= MC_KMAX && N2 <= MC_KMAX && N3 == 64, "chain shape outside the LDS plan");
    static_assert((K0 / mc_bk(N1)) % 2 == 0 && (N1 / mc_bk(N2)) % 2 == 0 && (N2 / mc_bk(N3)) % 2 == 0, "layers must start on stage 0");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* act = lds;                                             // [k][MC_LDA]
    float* wst = act + MC_ACT;                                    // [2][8][N + 4]
    float* whead = wst + 2 * MC_WST;                              // W4 [64][3] + b4
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long row0 = (long)blockIdx.x * BM;
    constexpr int S1 = K0 / mc_bk(N1), S2 = N1 / mc_bk(N2), S3 = N2 / mc_bk(N3), G = S1 + S2 + S3;

    if (wave >= 4) {
        // ------------------------------------------------------------------------------------ loader waves
        const int tid = threadIdx.x - 256;
        // input tile [128][K0] -> k-major act, in chunks of 32 k-columns (128 B per row, 4 float4 per thread); chunk c feeds
        // layer-1 slabs from k = 32 c on, so only chunk 0 is loaded before the MFMA waves start
        float4 xv[XU];
        float4 xr1[XMODE == 1 ? XU : 1], xr2[XMODE == 1 ? XU : 1];
        // XMODE 2: a thread's XU rows of a chunk share one column quad; their source rows / grid codes are fixed for the kernel
        long dsrc[XMODE == 2 ? XU : 1];
        float dg0[XMODE == 2 ? XU : 1], dg1[XMODE == 2 ? XU : 1];
        if constexpr (XMODE == 2) {
#pragma unroll
            for (int u = 0; u < XU; ++u) {
                const long row = row0 + ((tid + u * 256) >> 3);
                const long cr = row / a.n;                        // cloud * up + r
                const long cloud = cr / a.up;
                const int r = (int)(cr - cloud * a.up);
                dsrc[u] = cloud * a.n + (row - cr * a.n);
                dg0[u] = a.grid[r * 2 + 0];
                dg1[u] = a.grid[r * 2 + 1];
            }
        }
        auto load_x = [&](int c) {
            if constexpr (XMODE == 2) {
                const int col = c * 32 + (tid & 7) * 4;
                const float4 w0 = *reinterpret_cast<const float4*>(a.Wg + col), w1 = *reinterpret_cast<const float4*>(a.Wg + K0 + col);
                const float4 bb = *reinterpret_cast<const float4*>(a.bg + col);
                float4 h[XU];
#pragma unroll
                for (int u = 0; u < XU; ++u) h[u] = *reinterpret_cast<const float4*>(a.X + (size_t)dsrc[u] * a.ldx + col);
#pragma unroll
                for (int u = 0; u < XU; ++u) {
                    xv[u].x = fmaxf(__builtin_fmaf(dg1[u], w1.x, __builtin_fmaf(dg0[u], w0.x, h[u].x)) + bb.x, 0.f);
                    xv[u].y = fmaxf(__builtin_fmaf(dg1[u], w1.y, __builtin_fmaf(dg0[u], w0.y, h[u].y)) + bb.y, 0.f);
                    xv[u].z = fmaxf(__builtin_fmaf(dg1[u], w1.z, __builtin_fmaf(dg0[u], w0.z, h[u].z)) + bb.z, 0.f);
                    xv[u].w = fmaxf(__builtin_fmaf(dg1[u], w1.w, __builtin_fmaf(dg0[u], w0.w, h[u].w)) + bb.w, 0.f);
                }
                return;
            }
#pragma unroll
            for (int u = 0; u < XU; ++u) {
                const int idx = tid + u * 256;                    // BM rows x 8 quads
                const size_t o = (size_t)(row0 + (idx >> 3)) * a.ldx + c * 32 + (idx & 7) * 4;
                xv[u] = *reinterpret_cast<const float4*>(a.X + o);
                if constexpr (XMODE == 1) {
                    xr1[u] = *reinterpret_cast<const float4*>(a.X2 + o);
                    xr2[u] = *reinterpret_cast<const float4*>(a.X3 + o);
                }
            }
        };
        auto store_x = [&](int c) {
#pragma unroll
            for (int u = 0; u < XU; ++u) {
                const int idx = tid + u * 256;
                const int r = idx >> 3, k = c * 32 + (idx & 7) * 4;
                if constexpr (XMODE == 1) {                       // (x + x2) + x3, the association of the GEMM epilogue it replaces
                    xv[u].x = (xv[u].x + xr1[u].x) + xr2[u].x; xv[u].y = (xv[u].y + xr1[u].y) + xr2[u].y;
                    xv[u].z = (xv[u].z + xr1[u].z) + xr2[u].z; xv[u].w = (xv[u].w + xr1[u].w) + xr2[u].w;
                }
                act[(k + 0) * MC_LDA + r] = xv[u].x;
                act[(k + 1) * MC_LDA + r] = xv[u].y;
                act[(k + 2) * MC_LDA + r] = xv[u].z;
                act[(k + 3) * MC_LDA + r] = xv[u].w;
            }
        };
        if (tid < 64 * 3) whead[tid] = a.W4[tid];
        if (tid < 3) whead[192 + tid] = a.b4[tid];
        // the weight stream: slab g of the concatenation W1 | W2 | W3, mc_bk(N) rows each = 512 float4 -> 2 per thread.
        // Two slabs in flight (named registers wa0 wa1 / wb0 wb1; slab g travels in set g & 1): with 2048-cycle slabs a
        // single slab of lookahead does not cover an L2 round trip.  (Plain scalars + macros: register arrays passed by
        // reference into lambdas ended up in scratch memory.)
        float4 wa0, wa1, wb0, wb1;
        auto slab_ptr = [&](int g) -> const float* {
            if (g < S1) return a.W1 + (size_t)g * 2048;            // slabs are contiguous 2048-float pieces of the row-major W
            if (g < S1 + S2) return a.W2 + (size_t)(g - S1) * 2048;
            return a.W3 + (size_t)(g - S1 - S2) * 2048;
        };
        auto slab_dst = [&](int g, int idx) -> float* {
            const int n = (g < S1) ? N1 : (g < S1 + S2) ? N2 : N3;
            const int q = n / 4;
            return wst + (g & 1) * MC_WST + (idx / q) * (n + 4) + (idx % q) * 4;
        };
#define MC_LOAD(g, r0, r1)                                                                    \
        do {                                                                                  \
            const float* p_ = slab_ptr(g);                                                    \
            r0 = *reinterpret_cast<const float4*>(p_ + tid * 4);                              \
            r1 = *reinterpret_cast<const float4*>(p_ + (tid + 256) * 4);                      \
        } while (0)
#define MC_STORE(g, r0, r1)                                                                   \
        do {                                                                                  \
            *reinterpret_cast<float4*>(slab_dst(g, tid)) = r0;                                \
            *reinterpret_cast<float4*>(slab_dst(g, tid + 256)) = r1;                          \
        } while (0)
#define MC_STEP(g, r0, r1) /* r0 r1 hold slab g + 1; refilled with slab g + 3 */              \
        do {                                                                                  \
            if ((g) + 1 < G) {                                                                \
                MC_STORE((g) + 1, r0, r1);                                                    \
                if ((g) + 3 < G) MC_LOAD((g) + 3, r0, r1);                                    \
            }                                                                                 \
            if ((g) + 1 < S1 && ((g) + 1) % SPC == 0) {                                       \
                const int c_ = ((g) + 1) / SPC;                                               \
                store_x(c_);                                                                  \
                if (c_ + 1 < NCH) load_x(c_ + 1);                                             \
            }                                                                                 \
            __syncthreads();                                                                  \
            if ((g) + 1 == S1 || (g) + 1 == S1 + S2 || (g) + 1 == G) {                        \
                __syncthreads();                                                              \
                __syncthreads();                                                              \
            }                                                                                 \
        } while (0)
        // layer-1 slab s needs input columns [s * bk1, (s+1) * bk1): chunk c must be stored before slab c * (32 / bk1)
        constexpr int BK1 = mc_bk(N1), SPC = 32 / BK1;            // slabs per input chunk (4 for N1 = 256, 2 for N1 = 128)
        constexpr int NCH = K0 / 32;
        static_assert(G % 2 == 0, "the loader loop is unrolled by two");
        load_x(0);
        store_x(0);
        if (NCH > 1) load_x(1);
        MC_LOAD(0, wa0, wa1);
        MC_STORE(0, wa0, wa1);
        MC_LOAD(1, wb0, wb1);
        MC_LOAD(2, wa0, wa1);
        __syncthreads();                                          // input chunk 0, head weights and slab 0 are in place
        for (int g = 0; g < G; g += 2) {
            MC_STEP(g, wb0, wb1);
            MC_STEP(g + 1, wa0, wa1);
        }
#undef MC_LOAD
#undef MC_STORE
#undef MC_STEP
        return;
    }

    // ------------------------------------------------------------------------------------------ MFMA waves
    const int wm = wave >> 1, wn = wave & 1;
    const int fi = lane & 31, fk = lane >> 5;
#ifdef MC_CLOCK
    unsigned long long mc_wait_local = 0;
    const unsigned long long mc_t0 = __builtin_readcyclecounter();
#endif
    __syncthreads();
    chain_layer<K0, N1, BM>(act, wst, 0, a.b1, a.Y1, a.ldy1, row0, wm, wn, fi, fk MC_WAIT_ARG);
    chain_layer<N1, N2, BM>(act, wst, S1, a.b2, STASH ? a.Y2 : nullptr, a.ldy2, row0, wm, wn, fi, fk MC_WAIT_ARG);
    chain_layer<N2, N3, BM>(act, wst, S1 + S2, a.b3, STASH ? a.Y3 : nullptr, a.ldy3, row0, wm, wn, fi, fk MC_WAIT_ARG);
#ifdef MC_CLOCK
    if (blockIdx.x == 7 && threadIdx.x == 0) { mc_clock_ticks[0] = __builtin_readcyclecounter() - mc_t0; mc_clock_ticks[1] = G; mc_clock_ticks[2] = mc_wait_local; }
#endif
    // head: 64 -> 3 per row, the arithmetic of linear_small_n_kernel (fmaf chain over k, + bias, optional sigmoid offset)
    if (threadIdx.x < BM) {
        const int row = threadIdx.x;
        float o0 = 0.f, o1 = 0.f, o2 = 0.f;
#pragma unroll 8
        for (int k = 0; k < N3; ++k) {
            const float xv = act[k * MC_LDA + row];
            o0 = __builtin_fmaf(xv, whead[k * 3 + 0], o0);
            o1 = __builtin_fmaf(xv, whead[k * 3 + 1], o1);
            o2 = __builtin_fmaf(xv, whead[k * 3 + 2], o2);
        }
        float v[3] = {o0 + whead[192], o1 + whead[193], o2 + whead[194]};
        const long gr = row0 + row;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float w = v[c];
            if constexpr (STASH) { if (a.Z) a.Z[gr * a.ldz + c] = w; }
            if (a.mode == 1) w = a.R[gr * a.ldr + c] + (1.0f / (1.0f + expf(-w)) - 0.5f);
            a.out[gr * a.ldo + c] = w;
        }
    }
}

}  // namespace dispu

using namespace dispu;

// rows % 128 == 0; (K0, N1, N2, N3) in {(256,128,256,64), (256,256,256,64)}; 16-byte aligned X / W*, ldx % 4 == 0.
DISPU_EXPORT int dispu_mlp_chain_stash(long rows, int K0, int N1, int N2, int N3, const float* X, long ldx, const float* W1, const float* b1,
                                       const float* W2, const float* b2, const float* W3, const float* b3, const float* W4, const float* b4,
                                       float* Y1, long ldy1, float* Y2, long ldy2, float* Y3, long ldy3, float* Z, long ldz, int mode,
                                       const float* R, long ldr, float* out, long ldo, void* stream);

// shared launcher of the round-4 input modes (no stash variants: the training forward keeps its materialised inputs)
template <int XMODE>
static int mlp_chain_xmode(const ChainArgs& a, int K0, int N1, int N2, int N3, hipStream_t s) {
    const long rows = a.rows;
    const bool small = (rows % MC_BM) != 0 || rows / MC_BM < 192;
    const bool coarse = (K0 == 256 && N1 == 128 && N2 == 256 && N3 == 64), fine = (K0 == 256 && N1 == 256 && N2 == 256 && N3 == 64);
    if (!coarse && !fine) return (int)hipErrorInvalidValue;
    const dim3 grid((unsigned)(rows / (small ? 64 : MC_BM)));
    auto launch = [&](auto kern) -> int {
        static DevOnce once;
        if (once.needed()) {
            DISPU_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)MC_LDS_BYTES));
            once.done();
        }
        hipLaunchKernelGGL(kern, grid, dim3(512), MC_LDS_BYTES, s, a);
        return (int)hipGetLastError();
    };
    if (coarse) return small ? launch(mlp_chain_kernel<256, 128, 256, 64, 64, false, XMODE>) : launch(mlp_chain_kernel<256, 128, 256, 64, 128, false, XMODE>);
    return small ? launch(mlp_chain_kernel<256, 256, 256, 64, 64, false, XMODE>) : launch(mlp_chain_kernel<256, 256, 256, 64, 128, false, XMODE>);
}

// dispu_mlp_chain whose input is (X + X2) + X3 (three [rows, K0] matrices with one row stride), summed by the loader waves.
DISPU_EXPORT int dispu_mlp_chain_sum3(long rows, int K0, int N1, int N2, int N3, const float* X, const float* X2, const float* X3, long ldx,
                                      const float* W1, const float* b1, const float* W2, const float* b2, const float* W3, const float* b3,
                                      const float* W4, const float* b4, float* Y1, long ldy1, int mode, const float* R, long ldr, float* out,
                                      long ldo, void* stream) {
    if (rows < 0 || (rows % 64) != 0 || (ldx & 3) || !X || !X2 || !X3 || !W1 || !W2 || !W3 || !W4 || !b1 || !b2 || !b3 || !b4 || !out ||
        (mode == 1 && !R) || ((((uintptr_t)X) | ((uintptr_t)X2) | ((uintptr_t)X3) | ((uintptr_t)W1) | ((uintptr_t)W2) | ((uintptr_t)W3)) & 15))
        return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    ChainArgs a{rows, X, ldx, W1, b1, W2, b2, W3, b3, W4, b4, Y1, ldy1, nullptr, 0, nullptr, 0, nullptr, 0, R, ldr, out, ldo, mode};
    a.X2 = X2; a.X3 = X3;
    return mlp_chain_xmode<1>(a, K0, N1, N2, N3, (hipStream_t)stream);
}

// dispu_mlp_chain on duplicate_up's rows (ops.py:1152-1192) WITHOUT the [nclouds * up * n, K0] tensor: row (cloud * up + r) * n + i of
// the chain's input = relu(fmaf(grid[r][1], Wg[1][:], fmaf(grid[r][0], Wg[0][:], H[cloud * n + i][:])) + bg) -- dispu_dup_grid's
// arithmetic, evaluated by the loader waves.  H [nclouds * n, K0] (row stride ldh), Wg [2, K0], bg [K0], grid [up, 2].
DISPU_EXPORT int dispu_mlp_chain_dup(int nclouds, int n, int up, int K0, int N1, int N2, int N3, const float* H, long ldh, const float* Wg,
                                     const float* bg, const float* grid, const float* W1, const float* b1, const float* W2, const float* b2,
                                     const float* W3, const float* b3, const float* W4, const float* b4, float* Y1, long ldy1, int mode,
                                     const float* R, long ldr, float* out, long ldo, void* stream) {
    const long rows = (long)nclouds * up * n;
    if (nclouds < 0 || n <= 0 || up <= 0 || (rows % 64) != 0 || (ldh & 3) || !H || !Wg || !bg || !grid || !W1 || !W2 || !W3 || !W4 || !b1 || !b2 ||
        !b3 || !b4 || !out || (mode == 1 && !R) ||
        ((((uintptr_t)H) | ((uintptr_t)Wg) | ((uintptr_t)bg) | ((uintptr_t)W1) | ((uintptr_t)W2) | ((uintptr_t)W3)) & 15))
        return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    ChainArgs a{rows, H, ldh, W1, b1, W2, b2, W3, b3, W4, b4, Y1, ldy1, nullptr, 0, nullptr, 0, nullptr, 0, R, ldr, out, ldo, mode};
    a.Wg = Wg; a.bg = bg; a.grid = grid; a.n = n; a.up = up;
    return mlp_chain_xmode<2>(a, K0, N1, N2, N3, (hipStream_t)stream);
}

DISPU_EXPORT int dispu_mlp_chain(long rows, int K0, int N1, int N2, int N3, const float* X, long ldx, const float* W1, const float* b1,
                                 const float* W2, const float* b2, const float* W3, const float* b3, const float* W4, const float* b4,
                                 float* Y1, long ldy1, int mode, const float* R, long ldr, float* out, long ldo, void* stream) {
    return dispu_mlp_chain_stash(rows, K0, N1, N2, N3, X, ldx, W1, b1, W2, b2, W3, b3, W4, b4, Y1, ldy1, nullptr, 0, nullptr, 0, nullptr, 0,
                                 mode, R, ldr, out, ldo, stream);
}

// The same chain with every intermediate activation also written to HBM (training forward: the backward pass needs them).
DISPU_EXPORT int dispu_mlp_chain_stash(long rows, int K0, int N1, int N2, int N3, const float* X, long ldx, const float* W1, const float* b1,
                                       const float* W2, const float* b2, const float* W3, const float* b3, const float* W4, const float* b4,
                                       float* Y1, long ldy1, float* Y2, long ldy2, float* Y3, long ldy3, float* Z, long ldz, int mode,
                                       const float* R, long ldr, float* out, long ldo, void* stream) {
    if (rows < 0 || (rows % 64) != 0 || (ldx & 3) || !X || !W1 || !W2 || !W3 || !W4 || !b1 || !b2 || !b3 || !b4 || !out ||
        (mode == 1 && !R) || ((((uintptr_t)X) | ((uintptr_t)W1) | ((uintptr_t)W2) | ((uintptr_t)W3)) & 15))
        return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    ChainArgs a{rows, X, ldx, W1, b1, W2, b2, W3, b3, W4, b4, Y1, ldy1, Y2, ldy2, Y3, ldy3, Z, ldz, R, ldr, out, ldo, mode};
    hipStream_t s = (hipStream_t)stream;
    const bool stash = Y2 || Y3 || Z;
    // 128-row workgroups once they fill the chip (>= 192 of them); below that (the training step's 8 patches = 8192 rows) 64-row
    // workgroups: twice as many, each streaming the same weights for half the rows.  Same arithmetic, bit-identical results.
    const bool small = (rows % MC_BM) != 0 || rows / MC_BM < 192;
    const bool coarse = (K0 == 256 && N1 == 128 && N2 == 256 && N3 == 64), fine = (K0 == 256 && N1 == 256 && N2 == 256 && N3 == 64);
    if (!coarse && !fine) return (int)hipErrorInvalidValue;
    const dim3 grid((unsigned)(rows / (small ? 64 : MC_BM)));
    auto launch = [&](auto kern) -> int {
        static DevOnce once;                                   // per instantiation of this generic lambda, per device
        if (once.needed()) {
            DISPU_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)MC_LDS_BYTES));
            once.done();
        }
        hipLaunchKernelGGL(kern, grid, dim3(512), MC_LDS_BYTES, s, a);
        return (int)hipGetLastError();
    };
#define MC_PICK(K0_, N1_) \
    (small ? (stash ? launch(mlp_chain_kernel<K0_, N1_, 256, 64, 64, true>) : launch(mlp_chain_kernel<K0_, N1_, 256, 64, 64, false>)) \
           : (stash ? launch(mlp_chain_kernel<K0_, N1_, 256, 64, 128, true>) : launch(mlp_chain_kernel<K0_, N1_, 256, 64, 128, false>)))
    return coarse ? MC_PICK(256, 128) : MC_PICK(256, 256);
#undef MC_PICK
}
