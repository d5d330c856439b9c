// GEMM inner-loop laboratory for the fp32 MFMA path (interior tiles only): Y[M,N] = X[M,K] . W[K,N].
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/micro/gemm_lab.hip -o tools/micro/gemm_lab
// Variants differ in how far ahead operands are fetched:
//   FDB   fragment registers double-buffered: the ds_reads of k-step s+1 are issued before the MFMAs of step s
//   GD    global prefetch distance in slabs (1: tile t+2 requested during slab t; 2: tile t+3)
// Every variant must print the same checksum.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int BM, int BN, int BK, bool FDB, int GD, int MODE = 0>   // MODE 1: no global/LDS refill in the loop; 2: also no barrier
__global__ __launch_bounds__(256) void gemm_kernel(int M, int K, int N, int ldx, const float* __restrict__ X, const float* __restrict__ W,
                                                    float* __restrict__ Y, unsigned long long* __restrict__ stamps) {
    constexpr int WM = 2, WN = 2, NT = 256;
    constexpr int LDA = BM + 1, LDB = BN + 4;
    constexpr int STAGE = ((BK * (LDA + LDB) + 3) / 4) * 4;
    constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
    constexpr int A_F4 = BM * BK / 4 / NT, B_F4 = BN * BK / 4 / NT;
    constexpr int NSTAGE = 2;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 pa[GD][A_F4], pb[GD][B_F4];
    auto load_tile = [&](int k0, int slot) {
#pragma unroll
        for (int it = 0; it < A_F4; ++it) {
            if ((MODE == 11 || MODE == 12) && k0 > 0) break;    // A tile fetched once only
            const int idx = tid + it * NT;
            const int r = idx / (BK / 4), kq = idx % (BK / 4);
            if (MODE == 9)                    // tile-major X: slab t of row-tile blockIdx.y is one contiguous BM*BK chunk (timing only)
                pa[slot][it] = *reinterpret_cast<const float4*>(X + ((size_t)blockIdx.y * (K / BK) + k0 / BK) * (BM * BK) + (size_t)idx * 4);
            else if (MODE == 5 || MODE == 6)  // cache-resident stand-in addresses (no HBM / TLB pressure), same instruction count
                pa[slot][it] = *reinterpret_cast<const float4*>(X + (size_t)(r & 15) * ldx + (k0 & 63) + kq * 4);
            else
                pa[slot][it] = *reinterpret_cast<const float4*>(X + (size_t)(m0 + r) * ldx + k0 + kq * 4);
        }
        if (MODE == 10 && k0 > 0) return;   // B tile fetched once only
#pragma unroll
        for (int it = 0; it < B_F4; ++it) {
            const int idx = tid + it * NT;
            const int kr = idx / (BN / 4), nq = idx % (BN / 4);
            const int kb = (MODE == 12 || MODE == 13) ? (k0 + (int)(blockIdx.y % 32) * 4 * BK) % K : k0;   // rotated B start (timing only)
            pb[slot][it] = *reinterpret_cast<const float4*>(W + (size_t)(kb + kr) * N + n0 + nq * 4);
        }
    };
    auto store_tile = [&](int stage, int slot) {
        float* As = lds + stage * STAGE;
        float* Bs = As + BK * LDA;
#pragma unroll
        for (int it = 0; it < A_F4; ++it) {
            const int idx = tid + it * NT;
            const int r = idx / (BK / 4), kq = idx % (BK / 4);
            As[(kq * 4 + 0) * LDA + r] = pa[slot][it].x;
            As[(kq * 4 + 1) * LDA + r] = pa[slot][it].y;
            As[(kq * 4 + 2) * LDA + r] = pa[slot][it].z;
            As[(kq * 4 + 3) * LDA + r] = pa[slot][it].w;
        }
#pragma unroll
        for (int it = 0; it < B_F4; ++it) {
            const int idx = tid + it * NT;
            const int kr = idx / (BN / 4), nq = idx % (BN / 4);
            *reinterpret_cast<float4*>(&Bs[kr * LDB + nq * 4]) = pb[slot][it];
        }
    };

    const int ntile = K / BK;
    load_tile(0, 0);
    store_tile(0, 0);
    // tiles 1 .. GD are in flight when the loop starts
#pragma unroll
    for (int g = 0; g < GD; ++g)
        if (1 + g < ntile) load_tile((1 + g) * BK, g % GD);
    __syncthreads();

    const int fi = lane & 31, fk = lane >> 5;
    float dummy_sum = 0.f;
    unsigned long long c_refill = 0, c_compute = 0, c_barrier = 0;
    for (int tb = 0; tb < ntile; tb += GD) {
#pragma unroll
      for (int u = 0; u < GD; ++u) {
        const int t = tb + u;
        if (t >= ntile) break;
        const float* As = lds + (t % NSTAGE) * STAGE;
        const float* Bs = As + BK * LDA;
        unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        if (MODE == 8) s0 = __builtin_readcyclecounter();
        if ((MODE == 0 || MODE >= 3) && t + 1 < ntile) {   // MODE 6 = full refill with cache-resident A addresses
            const int slot = u;                          // == t % GD, compile-time after unrolling (ntile % GD == 0 assumed)
            if (MODE == 3 || MODE == 5) {                // global loads only: fold the loaded values into a dummy
#pragma unroll
                for (int it = 0; it < A_F4; ++it) dummy_sum += pa[slot][it].x + pa[slot][it].w;
#pragma unroll
                for (int it = 0; it < B_F4; ++it) dummy_sum += pb[slot][it].y + pb[slot][it].z;
            } else {
                store_tile((t + 1) % NSTAGE, slot);
            }
            if (MODE != 4 && t + 1 + GD < ntile) load_tile((t + 1 + GD) * BK, slot);
        }
        if (MODE == 8) { __builtin_amdgcn_sched_barrier(0); s1 = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); }
        if constexpr (MODE == 7) {
            // refill only: no fragment reads, no MFMAs
        } else if constexpr (FDB) {
            float af[2][TM], bf[2][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[0][i] = As[fk * LDA + wm * (TM * 32) + i * 32 + fi];
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[0][j] = Bs[fk * LDB + wn * (TN * 32) + j * 32 + fi];
#pragma unroll
            for (int s = 0; s < BK / 2; ++s) {
                const int cur = s & 1, nxt = cur ^ 1;
                if (s + 1 < BK / 2) {
                    const int kk = 2 * (s + 1);
#pragma unroll
                    for (int i = 0; i < TM; ++i) af[nxt][i] = As[(kk + fk) * LDA + wm * (TM * 32) + i * 32 + fi];
#pragma unroll
                    for (int j = 0; j < TN; ++j) bf[nxt][j] = Bs[(kk + fk) * LDB + wn * (TN * 32) + j * 32 + fi];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i], bf[cur][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) {
                float af[TM], bf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = As[(kk + fk) * LDA + wm * (TM * 32) + i * 32 + fi];
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[j] = Bs[(kk + fk) * LDB + wn * (TN * 32) + j * 32 + fi];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        }
        if (MODE == 8) { __builtin_amdgcn_sched_barrier(0); s2 = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); }
        if (MODE < 2 || MODE >= 3) __syncthreads();
        if (MODE == 8) { s3 = __builtin_readcyclecounter(); c_refill += s1 - s0; c_compute += s2 - s1; c_barrier += s3 - s2; }
      }
    }
    if (MODE == 8 && lane == 0 && blockIdx.x == 0 && (blockIdx.y == 7 || blockIdx.y == 100)) {
        unsigned long long* o = stamps + ((blockIdx.y == 7 ? 0 : 4) + wave) * 3;
        o[0] = c_refill; o[1] = c_compute; o[2] = c_barrier;
    }

    if (dummy_sum == 1234.5f) Y[0] = dummy_sum;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * (TN * 32) + j * 32 + fi;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                Y[(size_t)row * N + col] = acc[i][j][r];
            }
    }
}

template <int BM, int BN, int BK, bool FDB, int GD, int MODE = 0>
static void run(const char* name, int M, int K, int N, int ldx, const float* X, const float* W, float* Y) {
    constexpr size_t bytes = (size_t)2 * (((BK * (BM + 1 + BN + 4) + 3) / 4) * 4) * sizeof(float);
    auto kern = gemm_kernel<BM, BN, BK, FDB, GD, MODE>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    dim3 grid(N / BN, M / BM);
    unsigned long long* stamps;
    hipMalloc(&stamps, 24 * 8);
    hipMemset(stamps, 0, 24 * 8);
    hipLaunchKernelGGL(kern, grid, dim3(256), bytes, 0, M, K, N, ldx, X, W, Y, stamps);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int reps = 20;
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, grid, dim3(256), bytes, 0, M, K, N, ldx, X, W, Y, stamps);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    std::vector<float> h(1024);
    hipMemcpy(h.data(), Y + (size_t)(M / 2) * N, 1024 * 4, hipMemcpyDeviceToHost);
    double cs = 0;
    for (float v : h) cs += v;
    printf("%-34s ldx %5d M %6d K %5d N %4d  %8.1f us  %6.1f TFLOP/s  lds %zu B  checksum %.6f  (%s)\n", name, ldx, M, K, N, ms * 1e3,
           2.0 * M * K * N / ms / 1e9, bytes, cs, hipGetErrorString(hipGetLastError()));
    if (MODE == 8) {
        unsigned long long h8[24];
        hipMemcpy(h8, stamps, sizeof(h8), hipMemcpyDeviceToHost);
        for (int w = 0; w < 8; ++w)
            printf("    block %s wave %d: refill %8llu  compute %8llu  barrier %8llu cycles (per slab %.0f / %.0f / %.0f)\n", w < 4 ? "y=7  " : "y=100", w & 3,
                   h8[w * 3], h8[w * 3 + 1], h8[w * 3 + 2], h8[w * 3] / (double)(K / BK), h8[w * 3 + 1] / (double)(K / BK), h8[w * 3 + 2] / (double)(K / BK));
    }
    hipFree(stamps);
}

// ---- wave-specialised variant: waves 0..3 only read fragments + issue MFMAs, waves 4..7 only move tiles
// (global -> registers -> LDS).  One workgroup barrier per slab joins the two roles.
template <int BM, int BN, int BK, int GD = 1, bool AGPR = false>
__global__ __launch_bounds__(512) void gemm_ws_kernel(int M, int K, int N, int ldx, const float* __restrict__ X,
                                                       const float* __restrict__ W, float* __restrict__ Y, unsigned long long* __restrict__ stamps) {
    constexpr int WM = 2, WN = 2, NT = 256;
    constexpr int LDA = BM + 1, LDB = BN + 4;
    constexpr int STAGE = ((BK * (LDA + LDB) + 3) / 4) * 4;
    constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
    constexpr int A_F4 = BM * BK / 4 / NT, B_F4 = BN * BK / 4 / NT;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int ntile = K / BK;
    if (wave >= 4) {
        // ------------------------------------------------------------------ loader waves
        const int tid = threadIdx.x - 256;
        float4 pa0, pa1, pb0, pb1, pb2, pb3;                       // one tile in flight, named registers (no arrays)
        static_assert(A_F4 <= 2 && B_F4 <= 4, "lab loader holds at most 2 + 4 float4");
        const float* xa0 = X + (size_t)(m0 + (tid) / (BK / 4)) * ldx + ((tid) % (BK / 4)) * 4;
        const float* xa1 = X + (size_t)(m0 + (tid + NT) / (BK / 4)) * ldx + ((tid + NT) % (BK / 4)) * 4;
        const float* wb0 = W + (size_t)((tid) / (BN / 4)) * N + n0 + ((tid) % (BN / 4)) * 4;
        const float* wb1 = W + (size_t)((tid + NT) / (BN / 4)) * N + n0 + ((tid + NT) % (BN / 4)) * 4;
        const float* wb2 = W + (size_t)((tid + 2 * NT) / (BN / 4)) * N + n0 + ((tid + 2 * NT) % (BN / 4)) * 4;
        const float* wb3 = W + (size_t)((tid + 3 * NT) / (BN / 4)) * N + n0 + ((tid + 3 * NT) % (BN / 4)) * 4;
#define LAB_LOAD(k0)                                                                                   \
        do {                                                                                           \
            pa0 = *reinterpret_cast<const float4*>(xa0 + (k0));                                        \
            if (A_F4 > 1) pa1 = *reinterpret_cast<const float4*>(xa1 + (k0));                          \
            pb0 = *reinterpret_cast<const float4*>(wb0 + (size_t)(k0) * N);                            \
            if (B_F4 > 1) pb1 = *reinterpret_cast<const float4*>(wb1 + (size_t)(k0) * N);              \
            if (B_F4 > 2) pb2 = *reinterpret_cast<const float4*>(wb2 + (size_t)(k0) * N);              \
            if (B_F4 > 3) pb3 = *reinterpret_cast<const float4*>(wb3 + (size_t)(k0) * N);              \
        } while (0)
        auto st_a = [&](float* As, float4 v, int idx) {
            const int r = idx / (BK / 4), kq = idx % (BK / 4);
            As[(kq * 4 + 0) * LDA + r] = v.x; As[(kq * 4 + 1) * LDA + r] = v.y;
            As[(kq * 4 + 2) * LDA + r] = v.z; As[(kq * 4 + 3) * LDA + r] = v.w;
        };
        auto st_b = [&](float* Bs, float4 v, int idx) {
            *reinterpret_cast<float4*>(&Bs[(idx / (BN / 4)) * LDB + (idx % (BN / 4)) * 4]) = v;
        };
#define LAB_STORE(stage)                                                                               \
        do {                                                                                           \
            float* As_ = lds + (stage) * STAGE;                                                        \
            float* Bs_ = As_ + BK * LDA;                                                               \
            st_a(As_, pa0, tid);                                                                       \
            if (A_F4 > 1) st_a(As_, pa1, tid + NT);                                                    \
            st_b(Bs_, pb0, tid);                                                                       \
            if (B_F4 > 1) st_b(Bs_, pb1, tid + NT);                                                    \
            if (B_F4 > 2) st_b(Bs_, pb2, tid + 2 * NT);                                                \
            if (B_F4 > 3) st_b(Bs_, pb3, tid + 3 * NT);                                                \
        } while (0)
        LAB_LOAD(0);
        LAB_STORE(0);
        if (ntile > 1) LAB_LOAD(BK);
        __syncthreads();
        unsigned long long c_store = 0, c_issue = 0, c_bar = 0;
        for (int t = 0; t < ntile; ++t) {
            const unsigned long long s0 = __builtin_readcyclecounter();
            unsigned long long s1 = s0, s2 = s0;
            if (t + 1 < ntile) {
                LAB_STORE((t + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
                s1 = __builtin_readcyclecounter();
                __builtin_amdgcn_sched_barrier(0);
                if (t + 2 < ntile) LAB_LOAD((t + 2) * BK);
                __builtin_amdgcn_sched_barrier(0);
                s2 = __builtin_readcyclecounter();
            }
            __syncthreads();
            const unsigned long long s3 = __builtin_readcyclecounter();
            c_store += s1 - s0; c_issue += s2 - s1; c_bar += s3 - s2;
        }
        if (stamps && lane == 0 && blockIdx.x == 0 && blockIdx.y == 7) {
            unsigned long long* o = stamps + (wave - 4) * 3;
            o[0] = c_store; o[1] = c_issue; o[2] = c_bar;
        }
        return;
    }
    // ---------------------------------------------------------------------- MFMA waves
    const int wm = wave / WN, wn = wave % WN;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int fi = lane & 31, fk = lane >> 5;
    __syncthreads();
    unsigned long long c_comp = 0, c_mbar = 0;
    for (int t = 0; t < ntile; ++t) {
        const float* As = lds + (t & 1) * STAGE;
        const float* Bs = As + BK * LDA;
        const unsigned long long m0s = __builtin_readcyclecounter();
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = As[(kk + fk) * LDA + wm * (TM * 32) + i * 32 + fi];
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = Bs[(kk + fk) * LDB + wn * (TN * 32) + j * 32 + fi];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if constexpr (AGPR) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(af[i]), "v"(bf[j]));
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
                }
        }
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long m1s = __builtin_readcyclecounter();
        __syncthreads();
        const unsigned long long m2s = __builtin_readcyclecounter();
        c_comp += m1s - m0s; c_mbar += m2s - m1s;
    }
    if (stamps && lane == 0 && blockIdx.x == 0 && blockIdx.y == 7) {
        stamps[12 + wave * 2] = c_comp; stamps[13 + wave * 2] = c_mbar;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * (TN * 32) + j * 32 + fi;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                Y[(size_t)row * N + col] = acc[i][j][r];
            }
    }
}

template <int BM, int BN, int BK, int GD = 1, bool AGPR = false>
static void run_ws(const char* name, int M, int K, int N, int ldx, const float* X, const float* W, float* Y) {
    constexpr size_t bytes = (size_t)2 * (((BK * (BM + 1 + BN + 4) + 3) / 4) * 4) * sizeof(float);
    auto kern = gemm_ws_kernel<BM, BN, BK, GD, AGPR>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    dim3 grid(N / BN, M / BM);
    unsigned long long* stamps;
    hipMalloc(&stamps, 24 * 8);
    hipMemset(stamps, 0, 24 * 8);
    hipLaunchKernelGGL(kern, grid, dim3(512), bytes, 0, M, K, N, ldx, X, W, Y, stamps);
    hipDeviceSynchronize();
    {
        unsigned long long h8[24];
        hipMemcpy(h8, stamps, sizeof(h8), hipMemcpyDeviceToHost);
        const double nt = K / BK;
        for (int w = 0; w < 4; ++w)
            printf("    loader %d per slab: wait+store %.0f  issue %.0f  barrier %.0f   |  mfma wave %d: compute %.0f  barrier %.0f\n", w, h8[w * 3] / nt,
                   h8[w * 3 + 1] / nt, h8[w * 3 + 2] / nt, w, h8[12 + w * 2] / nt, h8[13 + w * 2] / nt);
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int reps = 20;
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, grid, dim3(512), bytes, 0, M, K, N, ldx, X, W, Y, (unsigned long long*)nullptr);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    std::vector<float> h(1024);
    hipMemcpy(h.data(), Y + (size_t)(M / 2) * N, 1024 * 4, hipMemcpyDeviceToHost);
    double cs = 0;
    for (float v : h) cs += v;
    printf("%-34s ldx %5d M %6d K %5d N %4d  %8.1f us  %6.1f TFLOP/s  lds %zu B  checksum %.6f  (%s)\n", name, ldx, M, K, N, ms * 1e3,
           2.0 * M * K * N / ms / 1e9, bytes, cs, hipGetErrorString(hipGetLastError()));
}

int main() {
    const int M = 32768, N = 256;
    for (int K : {2048, 256, 128}) {
      for (int pad : {0}) {
        const int ldx = K + pad;
        float *X, *W, *Y;
        hipMalloc(&X, (size_t)M * ldx * 4);
        hipMalloc(&W, (size_t)K * N * 4);
        hipMalloc(&Y, (size_t)M * N * 4);
        std::vector<float> hx((size_t)M * ldx), hw((size_t)K * N);
        unsigned s = 12345;
        for (size_t i = 0; i < (size_t)M; ++i)
            for (int k = 0; k < K; ++k) { s = s * 1664525u + 1013904223u; hx[i * ldx + k] = ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; }
        for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; }
        hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
        run<128, 256, 16, true, 1, 0>("128x256 bk16 fragDB full", M, K, N, ldx, X, W, Y);
        run_ws<128, 256, 16, 1>("WS 128x256 bk16", M, K, N, ldx, X, W, Y);
        run_ws<128, 256, 16, 1, true>("WS 128x256 bk16 AGPR acc", M, K, N, ldx, X, W, Y);
        run_ws<128, 128, 16, 1>("WS 128x128 bk16", M, K, N, ldx, X, W, Y);
        hipFree(X); hipFree(W); hipFree(Y);
      }
    }
    return 0;
}
