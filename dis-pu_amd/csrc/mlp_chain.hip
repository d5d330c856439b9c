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
typedef unsigned int mc_u32x4 __attribute__((ext_vector_type(4)));

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

constexpr int MC_BM = 128, MC_KMAX = 256;                         // MC_BM: rows per workgroup of the large-batch variant (BM = 64 below
constexpr int MC_ACT = MC_KMAX * (MC_BM + 1);                     // 8192 rows: twice the workgroups, half the rows each); floats
// a weight slab is bk(N) = 2048 / N rows of N floats (8 / 16 / 32 rows for N = 256 / 128 / 64): every slab carries the
// same 32 MFMAs per wave, so the barrier cadence does not depend on the layer width
__host__ __device__ constexpr int mc_bk(int n) { return 2048 / n; }
constexpr int MC_WST = 32 * (64 + 4);                             // floats per weight stage (largest: 32 x 68)
constexpr int MC_HEAD = 64 * 3 + 4;
constexpr int MC_NST = 3;                                         // weight stages (round 5): slab g is consumed, g + 1 is complete, g + 2 is being stored
constexpr size_t MC_LDS_BYTES = (size_t)(MC_ACT + MC_NST * MC_WST + MC_HEAD) * sizeof(float);
static_assert(MC_LDS_BYTES <= 160 * 1024, "the chain's LDS plan");

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
    // the four MFMA waves as WMW x WNW over BM x N: 2 x 2 (a wave owns BM / 2 rows, N / 2 columns), or 1 x 4 for the 32-row workgroups
    // of small batches (round 6: 8192 rows as 256 workgroups instead of 128) -- where a layer is narrower than 4 x 32 columns (N3 = 64)
    // only its first N / 32 waves work, the others keep the barrier count
    constexpr int WMW = (BM >= 64) ? 2 : 1, WNW = 4 / WMW;
    constexpr int COLS = (N / WNW >= 32) ? N / WNW : 32, NACT = N / COLS;      // columns per wave, waves along N that have columns
    constexpr int TNW = COLS / 32;                                // 32-wide column blocks per wave
    constexpr int LDW = N + 4, BK = mc_bk(N);
    constexpr int RT = BM / (32 * WMW), MC_LDA = BM + 1, ROWS = BM / WMW;   // 32-row tiles per wave
    constexpr int NS = K / BK, NSTEP = BK / 2;
    if constexpr (NACT < WNW) {
        if (wn >= NACT) {                                         // wave-uniform: no columns in this layer
            for (int s = 0; s < NS; ++s) __syncthreads();
            __syncthreads();
            __syncthreads();
            return;
        }
    }
    f32x16 acc[RT][TNW];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < TNW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // Round 5: the operands of k-step j + 1 are requested BEFORE the MFMAs of step j, across the slab barrier too.  A slab is only 32
    // MFMAs per wave (2048 pipe cycles); with the reads of a slab's first step issued behind its barrier, and every other step's reads
    // placed by the compiler right in front of the MFMAs that need them, a wave sat through an LDS round trip per step with the matrix
    // pipe drained (2600 - 2750 cycles of its own per slab, tools/micro/chain_lab.hip).  Reading ahead over the barrier needs slab
    // s + 1 COMPLETE while slab s is consumed: three weight stages, the loaders store slab g + 2 during slab g (and the input chunk
    // of slab g + 2 likewise).
    static_assert(NSTEP % 2 == 0, "the fragment double buffer starts every slab in set 0");
    float af[2][RT], bf[2][TNW];
#ifdef MC_X_NOA
#define MC_RDA(x) (1.0f + fi)
#else
#define MC_RDA(x) (x)
#endif
#if defined(MC_X_NOB)
    for (int j = 0; j < TNW; ++j) { bf[0][j] = 0.5f + fk; bf[1][j] = 0.25f + fk; }
#endif
#ifdef MC_X_NOB
#define MC_RDB(dst, ptr) do { } while (0)
#elif defined(MC_X_B128)
#define MC_RDB(dst, ptr) do { if constexpr (TNW == 4) { const float4 t_ = *reinterpret_cast<const float4*>((ptr) + 3 * fi); dst[0] = t_.x; dst[1] = t_.y; dst[2] = t_.z; dst[3] = t_.w; } \
                              else if constexpr (TNW == 2) { const float2 t_ = *reinterpret_cast<const float2*>((ptr) + fi); dst[0] = t_.x; dst[1] = t_.y; } \
                              else { dst[0] = *(ptr); } } while (0)
#else
#define MC_RDB(dst, ptr) do { _Pragma("unroll") for (int j = 0; j < TNW; ++j) dst[j] = (ptr)[j * 32]; } while (0)
#endif
    int st = g0 % MC_NST;                                          // stage of the slab being consumed
#ifdef MC_X_NOREAD                                                 // lab: no fragment reads at all (timing experiment, wrong results)
#pragma unroll
    for (int i = 0; i < RT; ++i) { af[0][i] = 1.0f + fi; af[1][i] = 2.0f + fi; }
#pragma unroll
    for (int j = 0; j < TNW; ++j) { bf[0][j] = 0.5f + fk; bf[1][j] = 0.25f + fk; }
#else
    {
        const float* ws = wst + st * MC_WST;
#pragma unroll
        for (int i = 0; i < RT; ++i) af[0][i] = MC_RDA(act[fk * MC_LDA + wm * ROWS + i * 32 + fi]);
        MC_RDB(bf[0], ws + fk * LDW + wn * COLS + fi);
    }
#endif
    // unrolled by 4: the per-slab LDS addresses become immediates (one base per four slabs) instead of 6 - 8 VALU instructions per slab
    // between the MFMAs (-3 % per slab, tools/micro/chain_lab.hip)
    static_assert(NS % 4 == 0, "slab loop unrolled by four");
#pragma unroll 4
    for (int s = 0; s < NS; ++s) {
        const float* ws = wst + st * MC_WST;
        const int stn = (st + 1 == MC_NST) ? 0 : st + 1;
        const float* wsn = wst + stn * MC_WST;
        const float* as = act + (s * BK) * MC_LDA;
        const bool more = s + 1 < NS;
#pragma unroll
        for (int s2 = 0; s2 < NSTEP; ++s2) {
            const int cur = s2 & 1, nxt = cur ^ 1;
#ifndef MC_X_NOREAD
            if (s2 + 1 < NSTEP) {
                const int kk = 2 * (s2 + 1);
#pragma unroll
                for (int i = 0; i < RT; ++i) af[nxt][i] = MC_RDA(as[(kk + fk) * MC_LDA + wm * ROWS + i * 32 + fi]);
                MC_RDB(bf[nxt], ws + (kk + fk) * LDW + wn * COLS + fi);
            } else if (more) {                                     // first step of the next slab (uniform branch; its stage is complete)
#pragma unroll
                for (int i = 0; i < RT; ++i) af[nxt][i] = MC_RDA(as[(BK + fk) * MC_LDA + wm * ROWS + i * 32 + fi]);
                MC_RDB(bf[nxt], wsn + fk * LDW + wn * COLS + fi);
            }
#endif
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int j = 0; j < TNW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i], bf[cur][j], acc[i][j], 0, 0, 0);
#if defined(MC_SCHED) && MC_SCHED == 1
            // lab: the DS reads spread between the MFMAs
#pragma unroll
            for (int q = 0; q < (RT + TNW) / 2; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, (RT * TNW) / ((RT + TNW) / 2), 0);
            }
#elif defined(MC_SCHED) && MC_SCHED == 2
            // lab: MFMAs first, reads behind them
            __builtin_amdgcn_sched_group_barrier(0x008, RT * TNW, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, RT + TNW, 0);
#elif defined(MC_SCHED) && MC_SCHED == 3
            // lab: no pinning
#else
            // pin: this step's DS reads (the next step's operands) first, then its RT * TNW MFMAs
            __builtin_amdgcn_sched_group_barrier(0x100, RT + TNW, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, RT * TNW, 0);
#endif
        }
        st = stn;
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
        const int n = wn * COLS + j * 32 + fi;
        const float bv = bias[n];
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * ROWS + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
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
    float* wst = act + MC_ACT;                                    // [MC_NST][8][N + 4]
    float* whead = wst + MC_NST * MC_WST;                              // W4 [64][3] + b4
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long row0 = (long)blockIdx.x * BM;
    constexpr int S1 = K0 / mc_bk(N1), S2 = N1 / mc_bk(N2), S3 = N2 / mc_bk(N3), G = S1 + S2 + S3;

    if (wave >= 4) {
        // ------------------------------------------------------------------------------------ loader waves
        // Round 5: every instruction a loader wave issues next to an MFMA wave of its SIMD takes ~100 - 250 cycles to get out
        // (tools/micro/chain_lab.hip: with the MFMA waves' LDS reads compiled out the slab still took 2900 cycles -- the loaders, at
        // ~35 instructions per slab, were the bound).  So the loop is fully unrolled (stage, layer, LDS offsets and chunk schedule
        // are compile-time), global memory goes through buffer loads whose per-slab offset is a scalar (no address VALU), and the LDS
        // destinations are one precomputed register per layer plus immediates: ~8 instructions per slab.
        const int tid = threadIdx.x - 256;
        const __amdgpu_buffer_rsrc_t rw1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.W1), 0, K0 * N1 * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rw2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.W2), 0, N1 * N2 * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rw3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.W3), 0, N2 * N3 * 4, 0x00020000);
        auto bload = [](__amdgpu_buffer_rsrc_t rs, int voff, int soff) -> float4 {
            const mc_u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
            return make_float4(__uint_as_float(r[0]), __uint_as_float(r[1]), __uint_as_float(r[2]), __uint_as_float(r[3]));
        };
        // input tile [BM][K0] -> k-major act, in chunks of 32 k-columns (128 B per row, XU float4 per thread); chunk c feeds
        // layer-1 slabs from k = 32 c on, so only chunk 0 is loaded before the MFMA waves start.  Row offsets are per-thread
        // constants (relative to the workgroup's first row: 32-bit offsets whatever the batch), the chunk is the scalar offset.
        const int xbytes = (int)(a.ldx * 4);
        const long xbase = (XMODE == 2) ? 0 : row0 * a.ldx;
        const int xrecs = (XMODE == 2) ? 0x7ffffff0 : BM * xbytes;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.X + xbase), 0, xrecs, 0x00020000);
        const __amdgpu_buffer_rsrc_t rx2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>((XMODE == 1 ? a.X2 : a.X) + xbase), 0, xrecs, 0x00020000);
        const __amdgpu_buffer_rsrc_t rx3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>((XMODE == 1 ? a.X3 : a.X) + xbase), 0, xrecs, 0x00020000);
        float4 xv[XU];
        float4 xr1[XMODE == 1 ? XU : 1], xr2[XMODE == 1 ? XU : 1];
        int xoff[XU];                                             // byte offset of the thread's row u, column quad (tid & 7), chunk 0
        // XMODE 2: a thread's XU rows of a chunk share one column quad; their source rows / grid codes are fixed for the kernel
        float dg0[XMODE == 2 ? XU : 1], dg1[XMODE == 2 ? XU : 1];
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            if constexpr (XMODE == 2) {
                const long row = row0 + ((tid + u * 256) >> 3);
                const long cr = row / a.n;                        // cloud * up + r
                const long cloud = cr / a.up;
                const int r = (int)(cr - cloud * a.up);
                xoff[u] = (int)(cloud * a.n + (row - cr * a.n)) * xbytes + (tid & 7) * 16;
                dg0[u] = a.grid[r * 2 + 0];
                dg1[u] = a.grid[r * 2 + 1];
            } else {
                xoff[u] = ((tid + u * 256) >> 3) * xbytes + (tid & 7) * 16;
            }
        }
        const int xdst = ((tid & 7) * 4) * MC_LDA + (tid >> 3);    // act index of (row of u = 0, first column of the quad), chunk 0
        auto load_x = [&](int c) {
#ifdef MC_X_NOLOADER
            return;
#endif
            if constexpr (XMODE == 2) {
                const int col = c * 32 + (tid & 7) * 4;
                const float4 w0 = *reinterpret_cast<const float4*>(a.Wg + col), w1 = *reinterpret_cast<const float4*>(a.Wg + K0 + col);
                const float4 bb = *reinterpret_cast<const float4*>(a.bg + col);
                float4 h[XU];
#pragma unroll
                for (int u = 0; u < XU; ++u) h[u] = bload(rx, xoff[u], c * 128);
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
                xv[u] = bload(rx, xoff[u], c * 128);
                if constexpr (XMODE == 1) {
                    xr1[u] = bload(rx2, xoff[u], c * 128);
                    xr2[u] = bload(rx3, xoff[u], c * 128);
                }
            }
        };
        auto store_x = [&](int c) {
#ifdef MC_X_NOLOADER
            return;
#endif
#pragma unroll
            for (int u = 0; u < XU; ++u) {
                float* d = act + xdst + (c * 32) * MC_LDA + u * 32;      // rows of u are 32 apart (256 threads / 8 quads)
                if constexpr (XMODE == 1) {                       // (x + x2) + x3, the association of the GEMM epilogue it replaces
                    xv[u].x = (xv[u].x + xr1[u].x) + xr2[u].x; xv[u].y = (xv[u].y + xr1[u].y) + xr2[u].y;
                    xv[u].z = (xv[u].z + xr1[u].z) + xr2[u].z; xv[u].w = (xv[u].w + xr1[u].w) + xr2[u].w;
                }
                d[0 * MC_LDA] = xv[u].x;
                d[1 * MC_LDA] = xv[u].y;
                d[2 * MC_LDA] = xv[u].z;
                d[3 * MC_LDA] = xv[u].w;
            }
        };
        if (tid < 64 * 3) whead[tid] = a.W4[tid];
        if (tid < 3) whead[192 + tid] = a.b4[tid];
        // the weight stream: slab g of the concatenation W1 | W2 | W3, mc_bk(N) rows each = 512 float4 -> 2 per thread (the second
        // one 256 float4 = 4096 bytes further on, in global memory and -- 256 / (N / 4) rows further down -- in the stage).
        // Two slabs in flight (named registers wa0 wa1 / wb0 wb1; slab g travels in set g & 1): with 2048-cycle slabs a
        // single slab of lookahead does not cover an L2 round trip.  (Plain scalars + macros: register arrays passed by
        // reference into lambdas ended up in scratch memory.)
        float4 wa0, wa1, wb0, wb1;
        const int wvoff = tid * 16;
        // LDS float index of the thread's first float4 inside a stage, per layer width
        const int wd1 = (tid / (N1 / 4)) * (N1 + 4) + (tid % (N1 / 4)) * 4;
        const int wd2 = (tid / (N2 / 4)) * (N2 + 4) + (tid % (N2 / 4)) * 4;
        const int wd3 = (tid / (N3 / 4)) * (N3 + 4) + (tid % (N3 / 4)) * 4;
#define MC_LOAD(g, r0, r1)                                                                    \
        do {                                                                                  \
            if ((g) < S1) { r0 = bload(rw1, wvoff, (g) * 8192); r1 = bload(rw1, wvoff, (g) * 8192 + 4096); }                                   \
            else if ((g) < S1 + S2) { r0 = bload(rw2, wvoff, ((g) - S1) * 8192); r1 = bload(rw2, wvoff, ((g) - S1) * 8192 + 4096); }           \
            else { r0 = bload(rw3, wvoff, ((g) - S1 - S2) * 8192); r1 = bload(rw3, wvoff, ((g) - S1 - S2) * 8192 + 4096); }                    \
        } while (0)
#define MC_STORE(g, r0, r1)                                                                   \
        do {                                                                                  \
            float* st_ = wst + ((g) % MC_NST) * MC_WST;                                       \
            if ((g) < S1) { *reinterpret_cast<float4*>(st_ + wd1) = r0; *reinterpret_cast<float4*>(st_ + wd1 + (1024 / N1) * (N1 + 4)) = r1; }           \
            else if ((g) < S1 + S2) { *reinterpret_cast<float4*>(st_ + wd2) = r0; *reinterpret_cast<float4*>(st_ + wd2 + (1024 / N2) * (N2 + 4)) = r1; }  \
            else { *reinterpret_cast<float4*>(st_ + wd3) = r0; *reinterpret_cast<float4*>(st_ + wd3 + (1024 / N3) * (N3 + 4)) = r1; }                    \
        } while (0)
#ifdef MC_X_NOLOADER                                               // lab: the loader waves only keep the barrier count (timing experiment)
#undef MC_LOAD
#undef MC_STORE
#define MC_LOAD(g, r0, r1) do { } while (0)
#define MC_STORE(g, r0, r1) do { } while (0)
#endif
        // During slab g: slab g + 2 goes to its stage (the one slab g - 1 was read from: its readers passed barrier g - 1) and slab g + 4 is
        // requested -- the MFMA waves read the first operands of slab g + 1 before barrier g, so slab g + 1 must be complete by barrier
        // g - 1.  The same holds for the input tile: the chunk that feeds slab g + 2 is stored during slab g.
#define MC_STEP(g, r0, r1) /* r0 r1 hold slab g + 2; refilled with slab g + 4 */              \
        do {                                                                                  \
            if ((g) + 2 < G) {                                                                \
                MC_STORE((g) + 2, r0, r1);                                                    \
                if ((g) + 4 < G) MC_LOAD((g) + 4, r0, r1);                                    \
            }                                                                                 \
            if ((g) + 2 < S1 && ((g) + 2) % SPC == 0) {                                       \
                const int c_ = ((g) + 2) / SPC;                                               \
                store_x(c_);                                                                  \
                if (c_ + 1 < NCH) load_x(c_ + 1);                                             \
            }                                                                                 \
            __syncthreads();                                                                  \
            if ((g) + 1 == S1 || (g) + 1 == S1 + S2 || (g) + 1 == G) {                        \
                __syncthreads();                                                              \
                __syncthreads();                                                              \
            }                                                                                 \
        } while (0)
        // layer-1 slab s needs input columns [s * bk1, (s+1) * bk1): chunk c must be stored before slab c * (32 / bk1) - 1 starts
        constexpr int BK1 = mc_bk(N1), SPC = 32 / BK1;            // slabs per input chunk (4 for N1 = 256, 2 for N1 = 128)
        constexpr int NCH = K0 / 32;
        static_assert(G % 2 == 0 && SPC % 2 == 0 && G >= 4, "the loader loop is unrolled by two; chunk stores fall on even slabs");
        {
            float4 t0, t1, t2, t3;
            load_x(0);
            MC_LOAD(0, t0, t1);
            MC_LOAD(1, t2, t3);
            MC_LOAD(2, wa0, wa1);
            MC_LOAD(3, wb0, wb1);
            store_x(0);
            if (NCH > 1) load_x(1);
            MC_STORE(0, t0, t1);
            MC_STORE(1, t2, t3);
        }
        __syncthreads();                                          // input chunk 0, head weights and slabs 0, 1 are in place
#pragma unroll
        for (int g = 0; g < G; g += 2) {
            MC_STEP(g, wa0, wa1);
            MC_STEP(g + 1, wb0, wb1);
        }
#undef MC_LOAD
#undef MC_STORE
#undef MC_STEP
        return;
    }

    // ------------------------------------------------------------------------------------------ MFMA waves
    const int wm = (BM >= 64) ? (wave >> 1) : 0, wn = (BM >= 64) ? (wave & 1) : wave;
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
    const bool stash = Y2 || Y3 || Z;
    if (rows < 0 || (rows % (stash ? 32 : 64)) != 0 || (ldx & 3) || !X || !W1 || !W2 || !W3 || !W4 || !b1 || !b2 || !b3 || !b4 || !out ||
        (mode == 1 && !R) || ((((uintptr_t)X) | ((uintptr_t)W1) | ((uintptr_t)W2) | ((uintptr_t)W3)) & 15))
        return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    ChainArgs a{rows, X, ldx, W1, b1, W2, b2, W3, b3, W4, b4, Y1, ldy1, Y2, ldy2, Y3, ldy3, Z, ldz, R, ldr, out, ldo, mode};
    hipStream_t s = (hipStream_t)stream;
    // 128-row workgroups once they fill the chip (>= 192 of them); below that (the training step's 8 patches = 8192 rows) 64-row
    // workgroups: twice as many, each streaming the same weights for half the rows.  Same arithmetic, bit-identical results.
    const bool small = (rows % MC_BM) != 0 || rows / MC_BM < 192;
    // round 6, the stashing (training) variant: 32-row workgroups while 64-row ones would leave CUs idle (fewer than 192 of them: the
    // 8-patch training step's 8192 rows are 128), or where the row count demands it
    const bool tiny = stash && small && ((rows % 64) != 0 || rows / 64 < 192);
    const bool coarse = (K0 == 256 && N1 == 128 && N2 == 256 && N3 == 64), fine = (K0 == 256 && N1 == 256 && N2 == 256 && N3 == 64);
    if (!coarse && !fine) return (int)hipErrorInvalidValue;
    const dim3 grid((unsigned)(rows / (tiny ? 32 : small ? 64 : MC_BM)));
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
    tiny ? launch(mlp_chain_kernel<K0_, N1_, 256, 64, 32, true>) : \
    (small ? (stash ? launch(mlp_chain_kernel<K0_, N1_, 256, 64, 64, true>) : launch(mlp_chain_kernel<K0_, N1_, 256, 64, 64, false>)) \
           : (stash ? launch(mlp_chain_kernel<K0_, N1_, 256, 64, 128, true>) : launch(mlp_chain_kernel<K0_, N1_, 256, 64, 128, false>)))
    return coarse ? (MC_PICK(256, 128)) : (MC_PICK(256, 256));
#undef MC_PICK
}
