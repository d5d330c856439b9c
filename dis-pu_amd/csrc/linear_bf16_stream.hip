// Streaming bf16-product GEMM for the training step's largest dense products (BASELINE configs[4], Trainer(dtype="bf16")):
//   Y[M][N] = act(X[M][K] . B[K][N] + bias),   X fp32 in HBM (rounded to bf16 on its way into the matrix pipe), B given as
//   Bt[N][K] bf16 (k contiguous: dispu_bf16_pack), fp32 accumulation on v_mfma_f32_32x32x16_bf16, Y fp32 or bf16.
// after_conv forward ([rows x 2048] x [2048 x 256]) and its dX ([rows x 256] x [256 x 2048]) are operand / result STREAMS at bf16
// MFMA speed (the matrix pipe needs a third of the time HBM needs for the fp32 operand): gemm_bf16_kernel (linear_bf16.hip: global
// -> registers -> convert -> LDS, one slab in flight, 16 scalar loads per thread and slab for an n-contiguous B) reaches 3.5 x the
// HBM time on them.  Here:
//   * operands go global -> LDS by DMA (global_load_lds_dwordx4: no data registers, no ds_write), three stages, two slabs in flight;
//     a slab is 32 k: X rows as 128 B of fp32, Bt rows as 64 B of bf16; 8 waves issue 4 (3) one-KB pieces each per slab;
//   * the LDS image of a piece is lane-linear, so the bank swizzle is applied on the GLOBAL side: the lane that lands at 16-byte
//     position p of row r fetches chunk p ^ (r & 7) (X) / p ^ ((n >> 1) & 3) (Bt) -- every 8 lanes of a fragment read then cover all
//     32 banks once;
//   * a lane converts its 8 consecutive k of an X row (two ds_read_b128) with four v_cvt_pk_bf16_f32 (round to nearest even, the
//     rounding of linear_bf16.hip: results are bit-identical to gemm_bf16_kernel's up to the association of the k slabs, which is
//     the same ascending order);
//   * all 8 waves load and compute (2 x 4 or 4 x 2 wave grid, 64 accumulator registers): the pipe is busy a third of the time, there
//     is nothing to specialise for.
// Shapes outside M % 128 == 0, K % 32 == 0, N % 128 == 0 (16-byte aligned rows) return hipErrorInvalidValue: the caller keeps
// dispu_linear_bf16 for them.
#include "common.h"

namespace dispu {

typedef __bf16 sb_bf16x8 __attribute__((ext_vector_type(8)));
typedef float sb_f32x16 __attribute__((ext_vector_type(16)));

constexpr int SB_BM = 128, SB_BK = 32, SB_NST = 3;

struct SbArgs {
    int M, K, N;                             // K: contraction length of ONE split
    const void* X; long ldx;                 // fp32, or bf16 (A_BF16) -- row stride in elements
    const unsigned short* Bt; long ldb;      // [N][K total] bf16, row stride in elements
    const float* bias; int act;
    void* Y; long ldy; int y_bf16;
    long y_split;                            // blockIdx.y = k split: X / Bt advance by K elements, Y by y_split elements (partial products)
};

template <int BN, bool A_BF16>
__global__ __launch_bounds__(512) void gemm_bf16_stream_kernel(SbArgs a) {
    constexpr int WN = BN / 64, WM = 8 / WN;                     // wave grid: 2 x 4 (BN = 256) or 4 x 2 (BN = 128)
    constexpr int TI = SB_BM / WM / 32, TJ = 2;                  // 32 x 32 blocks per wave: 2 x 2 or 1 x 2
    constexpr int A_BYTES = SB_BM * SB_BK * (A_BF16 ? 2 : 4), B_BYTES = BN * SB_BK * 2, STAGE = A_BYTES + B_BYTES;
    constexpr int NA = A_BYTES / 1024, NB = B_BYTES / 1024, PER = (NA + NB) / 8;   // 1 KB pieces per slab, per wave
    static_assert((NA + NB) % 8 == 0 && NA % 8 == 0, "pieces must split evenly over the eight waves");
    extern __shared__ __attribute__((aligned(16))) char sb_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN, li = lane & 31, kq = lane >> 5;
    const int tiles_n = a.N / BN;
    const int m0 = (blockIdx.x / tiles_n) * SB_BM, n0 = (blockIdx.x % tiles_n) * BN;
    const int nt = a.K / SB_BK;
    const size_t koff = (size_t)blockIdx.y * a.K;                // this split's first k

    // this wave's pieces: X pieces wave, wave + 8 (rows 8 q .. 8 q + 7, all 128 B of the slab), Bt pieces wave (, wave + 8)
    const char* gsrc[PER];
    int ldst[PER];
    long gstep[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        if (u < NA / 8) {
            if constexpr (A_BF16) {                              // X stored as bf16: rows of 64 B, the Bt layout
                const int q = wave + 8 * u, r = 16 * q + (lane >> 2), p = lane & 3;
                gsrc[u] = reinterpret_cast<const char*>(reinterpret_cast<const unsigned short*>(a.X) + (size_t)(m0 + r) * a.ldx + koff + 8 * (p ^ ((r >> 1) & 3)));
                ldst[u] = q * 1024;
                gstep[u] = (long)SB_BK * 2;
            } else {
                const int q = wave + 8 * u, r = 8 * q + (lane >> 3), p = lane & 7;
                gsrc[u] = reinterpret_cast<const char*>(reinterpret_cast<const float*>(a.X) + (size_t)(m0 + r) * a.ldx + koff + 4 * (p ^ (r & 7)));
                ldst[u] = q * 1024;
                gstep[u] = (long)SB_BK * 4;
            }
        } else {
            const int q = wave + 8 * (u - NA / 8), n = 16 * q + (lane >> 2), p = lane & 3;
            gsrc[u] = reinterpret_cast<const char*>(a.Bt + (size_t)(n0 + n) * a.ldb + koff + 8 * (p ^ ((n >> 1) & 3)));
            ldst[u] = A_BYTES + q * 1024;
            gstep[u] = (long)SB_BK * 2;
        }
    }
    auto issue = [&](int t) {
        char* st = sb_lds + (t % SB_NST) * STAGE;
#pragma unroll
        for (int u = 0; u < PER; ++u)
        {   // (float pointers: with char* operands to the DMA builtin the host pass silently drops the kernel's stub)
            const float* src = reinterpret_cast<const float*>(gsrc[u] + (size_t)t * gstep[u]);
            float* dst = reinterpret_cast<float*>(st + ldst[u]);
            __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };

    sb_f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addresses inside a stage (bytes): X row r, chunks c0 = 4 s + 2 kq and c0 + 1; Bt row n, chunk 2 s + kq
    int a_row[TI], a_sw[TI], b_row[TJ], b_sw[TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        const int r = wm * (SB_BM / WM) + 32 * i + li;
        a_row[i] = r * (A_BF16 ? 64 : 128);
        a_sw[i] = A_BF16 ? ((r >> 1) & 3) : (r & 7);
    }
#pragma unroll
    for (int j = 0; j < TJ; ++j) { const int n = wn * 64 + 32 * j + li; b_row[j] = A_BYTES + n * 64; b_sw[j] = (n >> 1) & 3; }

    issue(0);
    if (nt > 1) issue(1);
    for (int t = 0; t < nt; ++t) {
        // slab t has landed when at most the pieces of slab t + 1 are outstanding; then everybody's pieces have (barrier), and
        // everybody has finished reading slab t - 1, whose stage slab t + 2 may now overwrite
        if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        if (t + 2 < nt) issue(t + 2);
        const char* st = sb_lds + (t % SB_NST) * STAGE;
#pragma unroll
        for (int s = 0; s < SB_BK / 16; ++s) {
            sb_bf16x8 fa[TI], fb[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                if constexpr (A_BF16) {
                    fa[i] = *reinterpret_cast<const sb_bf16x8*>(st + a_row[i] + (((2 * s + kq) ^ a_sw[i]) << 4));
                } else {
                    const int c0 = 4 * s + 2 * kq;
                    const float4 lo = *reinterpret_cast<const float4*>(st + a_row[i] + ((c0 ^ a_sw[i]) << 4));
                    const float4 hi = *reinterpret_cast<const float4*>(st + a_row[i] + (((c0 + 1) ^ a_sw[i]) << 4));
                    fa[i] = sb_bf16x8{(__bf16)lo.x, (__bf16)lo.y, (__bf16)lo.z, (__bf16)lo.w, (__bf16)hi.x, (__bf16)hi.y, (__bf16)hi.z, (__bf16)hi.w};
                }
            }
#pragma unroll
            for (int j = 0; j < TJ; ++j) fb[j] = *reinterpret_cast<const sb_bf16x8*>(st + b_row[j] + (((2 * s + kq) ^ b_sw[j]) << 4));
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    }

    // epilogue: lane holds column .. + li, rows (r & 3) + 8 (r >> 2) + 4 kq of each 32 x 32 block: 128-byte row segments per store
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int col = n0 + wn * 64 + 32 * j + li;
            const float bv = a.bias ? a.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * (SB_BM / WM) + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * kq;
                float v = acc[i][j][r] + bv;
                if (a.act == 1) v = fmaxf(v, 0.f);
                const size_t o = (size_t)blockIdx.y * a.y_split + (size_t)row * a.ldy + col;
                if (a.y_bf16) reinterpret_cast<__bf16*>(a.Y)[o] = (__bf16)v;
                else reinterpret_cast<float*>(a.Y)[o] = v;
            }
        }
}

// ---- weight gradients: dW[K][N] = X^T . Z over the rows m of the batch (X [M][K], Z [M][N], both fp32 or both bf16 in HBM) ----------
// The contraction index m is the SLOW index of both operands, so a lane's 8 consecutive k of an MFMA operand are 8 rows apart in the
// row-major tiles: slabs of 32 rows land in LDS as they are (DMA, rows of 128 X columns / BN Z columns), a lane gathers its column
// with 8 strided reads and rounds on the way (fp32 storage; bf16 storage is taken as is).  That is 16 LDS reads per MFMA pair of a
// 2 x 2 wave tile -- half the matrix pipe at bf16 speed, still several times what the operand stream allows.  The rows are split over
// blockIdx.y (the output is only K x N: 16 tiles for after_conv), partial tiles go to caller scratch [split][K + 1][N] (row K = the
// column sums of Z = the bias gradient, from the un-rounded values in LDS) and tn_stream_reduce_kernel adds them in split order.
struct StArgs {
    int M, K, N;                     // rows of ONE split, X columns, Z columns
    const void* X; long ldx;
    const void* Z; long ldz;
    float* part; int with_colsum;
};

template <int BN, bool ST_BF16>
__global__ __launch_bounds__(512) void gemm_bf16_tn_stream_kernel(StArgs a) {
    constexpr int ES = ST_BF16 ? 2 : 4, XROW = 128 * ES, ZROW = BN * ES;
    constexpr int X_BYTES = 32 * XROW, Z_BYTES = 32 * ZROW, STAGE = X_BYTES + Z_BYTES;
    constexpr int NX = X_BYTES / 1024, NZ = Z_BYTES / 1024, PER = (NX + NZ) / 8;
    static_assert(NX % 8 == 0 && NZ % 8 == 0, "pieces must split evenly over the eight waves");
    constexpr int WN = BN / 64, WM = 8 / WN, TI = 128 / WM / 32, TJ = 2;
    extern __shared__ __attribute__((aligned(16))) char sb_lds[];
    __shared__ float colred[4][256];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN, li = lane & 31, kq = lane >> 5;
    const int tiles_n = a.N / BN;
    const int k0 = (blockIdx.x / tiles_n) * 128, n0 = (blockIdx.x % tiles_n) * BN;
    const size_t mbeg = (size_t)blockIdx.y * a.M;
    const int nt = a.M / 32;

    const char* gsrc[PER];
    int ldst[PER];
    long gstep[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int p = wave + 8 * u;
        if (u < NX / 8) {
            const int off = p * 1024 + 16 * lane, row = off / XROW, cb = off % XROW;
            gsrc[u] = reinterpret_cast<const char*>(a.X) + ((mbeg + row) * a.ldx + k0) * ES + cb;
            ldst[u] = p * 1024;
            gstep[u] = (long)32 * a.ldx * ES;
        } else {
            const int q = p - NX, off = q * 1024 + 16 * lane, row = off / ZROW, cb = off % ZROW;
            gsrc[u] = reinterpret_cast<const char*>(a.Z) + ((mbeg + row) * a.ldz + n0) * ES + cb;
            ldst[u] = X_BYTES + q * 1024;
            gstep[u] = (long)32 * a.ldz * ES;
        }
    }
    auto issue = [&](int t) {
        char* st = sb_lds + (t % SB_NST) * STAGE;
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const float* src = reinterpret_cast<const float*>(gsrc[u] + (size_t)t * gstep[u]);
            float* dst = reinterpret_cast<float*>(st + ldst[u]);
            __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };
    sb_f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const bool colsum = a.with_colsum && k0 == 0;                  // workgroup-uniform
    constexpr int CP = 512 / BN;                                   // threads per column
    float csum = 0.f;
    int xcol[TI], zcol[TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i) xcol[i] = (wm * (128 / WM) + 32 * i + li) * ES;
#pragma unroll
    for (int j = 0; j < TJ; ++j) zcol[j] = X_BYTES + (wn * 64 + 32 * j + li) * ES;

    issue(0);
    if (nt > 1) issue(1);
    for (int t = 0; t < nt; ++t) {
        if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        if (t + 2 < nt) issue(t + 2);
        const char* st = sb_lds + (t % SB_NST) * STAGE;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            sb_bf16x8 fa[TI], fb[TJ];
            const int mrow = 16 * s2 + 8 * kq;
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if constexpr (ST_BF16) fa[i][e] = *reinterpret_cast<const __bf16*>(st + (mrow + e) * XROW + xcol[i]);
                    else fa[i][e] = (__bf16)*reinterpret_cast<const float*>(st + (mrow + e) * XROW + xcol[i]);
                }
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if constexpr (ST_BF16) fb[j][e] = *reinterpret_cast<const __bf16*>(st + (mrow + e) * ZROW + zcol[j]);
                    else fb[j][e] = (__bf16)*reinterpret_cast<const float*>(st + (mrow + e) * ZROW + zcol[j]);
                }
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (colsum) {
            const int col = tid % BN, part = tid / BN;
#pragma unroll
            for (int r = 0; r < 32 / CP; ++r) {
                const char* q = st + X_BYTES + (part + CP * r) * ZROW + col * ES;
                csum += ST_BF16 ? (float)*reinterpret_cast<const __bf16*>(q) : *reinterpret_cast<const float*>(q);
            }
        }
    }
    const size_t rows_p = (size_t)a.K + (a.with_colsum ? 1 : 0);
    float* __restrict__ P = a.part + (size_t)blockIdx.y * rows_p * a.N;
    if (colsum) {
        colred[tid / BN][tid % BN] = csum;
        __syncthreads();
        if (tid < BN) {
            float v = colred[0][tid];
#pragma unroll
            for (int g = 1; g < CP; ++g) v += colred[g][tid];
            P[(size_t)a.K * a.N + n0 + tid] = v;
        }
    }
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int col = n0 + wn * 64 + 32 * j + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = k0 + wm * (128 / WM) + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * kq;
                P[(size_t)row * a.N + col] = acc[i][j][r];
            }
        }
}

// out[row][col] (+)= sum_s part[s][row][col]; row K (with_colsum) -> dbias[col] (+)=.  32 elements x 8 split groups per workgroup: group g
// adds splits g, g + 8, .. in ascending order, the eight group sums are added g = 0 .. 7: a fixed association (deterministic), with
// chains 8x shorter than one thread per element (up to 256 splits for the one-tile outputs).
__global__ __launch_bounds__(256) void tn_stream_reduce_kernel(int K, int N, int splits, int with_colsum, const float* __restrict__ part,
                                                               float* __restrict__ out, long ldo, int accumulate, float* __restrict__ dbias) {
    __shared__ float red[8][32];
    const int el = threadIdx.x & 31, g = threadIdx.x >> 5;
    const long rows_p = (long)K + (with_colsum ? 1 : 0), total = rows_p * N;
    const long e = (long)blockIdx.x * 32 + el;
    float sum = 0.f;
    if (e < total)
#pragma unroll 4
        for (int t = g; t < splits; t += 8) sum += part[(long)t * total + e];
    red[g][el] = sum;
    __syncthreads();
    if (g == 0 && e < total) {
        float r = red[0][el];
#pragma unroll
        for (int q = 1; q < 8; ++q) r += red[q][el];
        const long row = e / N, col = e % N;
        float* o = (row < K) ? out + row * ldo + col : dbias + col;
        *o = accumulate ? *o + r : r;
    }
}

static int tn_stream_plan(int M, int K, int N, int& splits) {
    if (M <= 0 || K <= 0 || N <= 0 || (K % 128) || (N % 128)) return 0;
    const int bn = (N % 256) == 0 ? 256 : 128;
    const long tiles = (long)(K / 128) * (N / bn);
    int s = 1;
    while (s < 256 && tiles * s < 256) s <<= 1;
    while (s > 1 && (M % (32 * s) != 0 || M / s < 128)) s >>= 1;
    if (M % (32 * s) != 0) return 0;
    splits = s;
    return 1;
}

// out bf16 [rows][cols] = W (transpose = 0) or [cols][rows] = W^T (transpose = 1), round to nearest even; W fp32 [rows][cols], row stride ldw
__global__ void bf16_pack_kernel(int rows, int cols, const float* __restrict__ W, long ldw, int transpose, __bf16* __restrict__ out) {
    __shared__ float t[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 256 threads: 32 x 8
    for (int y = ty; y < 32; y += 8) {
        const int r = r0 + y, c = c0 + tx;
        t[y][tx] = (r < rows && c < cols) ? W[(size_t)r * ldw + c] : 0.f;
    }
    __syncthreads();
    for (int y = ty; y < 32; y += 8) {
        if (!transpose) {
            const int r = r0 + y, c = c0 + tx;
            if (r < rows && c < cols) out[(size_t)r * cols + c] = (__bf16)t[y][tx];
        } else {
            const int c = c0 + y, r = r0 + tx;
            if (r < rows && c < cols) out[(size_t)c * rows + r] = (__bf16)t[tx][y];
        }
    }
}

}  // namespace dispu

using namespace dispu;

// bf16 image of a weight matrix for dispu_linear_bf16_stream: out [rows][cols] (transpose = 0) or [cols][rows] (transpose = 1).
DISPU_EXPORT int dispu_bf16_pack(int rows, int cols, const float* W, long ldw, int transpose, void* out, void* stream) {
    if (rows < 0 || cols < 0 || (rows > 0 && cols > 0 && (!W || !out || ldw < cols))) return (int)hipErrorInvalidValue;
    if (rows == 0 || cols == 0) return 0;
    hipLaunchKernelGGL(bf16_pack_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, (hipStream_t)stream, rows, cols, W, ldw, transpose,
                       reinterpret_cast<__bf16*>(out));
    return (int)hipGetLastError();
}

// Y = act(X . B + bias) with B given as Bt [N][K] bf16 (dispu_bf16_pack of W with transpose = 1; for dX = dZ . W^T pack W itself).
// x_bf16 / y_bf16: X / Y stored as bf16.  splits > 1: the contraction is cut into `splits` equal parts (K / splits a multiple of 32)
// and part s goes, without bias / activation, to Y + s * y_split (fp32 partial products: dispu_linear_splitk_finish adds them in
// order) -- few-row products fill the chip that way.  hipErrorInvalidValue for shapes outside the streaming kernel (file header).
DISPU_EXPORT int dispu_linear_bf16_stream(int M, int K, int N, const void* X, long ldx, int x_bf16, const void* Bt, long ldb, const float* bias,
                                          int act, void* Y, long ldy, int y_bf16, int splits, long y_split, void* stream) {
    if (splits < 1) splits = 1;
    if (M <= 0 || K <= 0 || N <= 0 || !X || !Bt || !Y || (M % SB_BM) || (K % (SB_BK * splits)) || (N % 128) || (ldx & (x_bf16 ? 7 : 3)) || (ldb & 7) ||
        ldx < K || ldb < K || ldy < N || (((uintptr_t)X) & 15) || (((uintptr_t)Bt) & 15) || (splits > 1 && (bias || act || y_bf16)) || splits > 64)
        return (int)hipErrorInvalidValue;
    SbArgs a{M, K / splits, N, X, ldx, reinterpret_cast<const unsigned short*>(Bt), ldb, bias, act, Y, ldy, y_bf16, y_split};
    hipStream_t s = (hipStream_t)stream;
    const bool wide = (N % 256) == 0;
    const int bn = wide ? 256 : 128;
    const size_t lds = (size_t)SB_NST * (SB_BM * SB_BK * (x_bf16 ? 2 : 4) + bn * SB_BK * 2);
    static DevOnce attr;
    if (attr.needed()) {
        DISPU_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_stream_kernel<256, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        DISPU_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_stream_kernel<128, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        DISPU_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_stream_kernel<256, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        DISPU_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_stream_kernel<128, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr.done();
    }
    const dim3 grid((M / SB_BM) * (N / bn), splits), blk(512);
    if (wide && x_bf16) hipLaunchKernelGGL((gemm_bf16_stream_kernel<256, true>), grid, blk, lds, s, a);
    else if (wide) hipLaunchKernelGGL((gemm_bf16_stream_kernel<256, false>), grid, blk, lds, s, a);
    else if (x_bf16) hipLaunchKernelGGL((gemm_bf16_stream_kernel<128, true>), grid, blk, lds, s, a);
    else hipLaunchKernelGGL((gemm_bf16_stream_kernel<128, false>), grid, blk, lds, s, a);
    return (int)hipGetLastError();
}

// dW = X^T . Z on the streaming kernel (see above): X [M][K], Z [M][N] both fp32 (storage = 0) or both bf16 (storage = 3); out [K][N]
// (+)=, dbias [N] (+)= the column sums of Z (optional).  scratch: dispu_linear_tn_bf16_stream_scratch_floats floats (0 = shape outside
// the kernel: K % 128, N % 128, M divisible into 32-row slabs per split).  Same products as dispu_linear_tn_bf16, another split plan.
DISPU_EXPORT long dispu_linear_tn_bf16_stream_scratch_floats(int M, int K, int N) {
    int splits;
    if (!tn_stream_plan(M, K, N, splits)) return 0;
    return (long)splits * ((long)K + 1) * N;
}

DISPU_EXPORT int dispu_linear_tn_bf16_stream(int M, int K, int N, const void* X, long ldx, const void* Z, long ldz, int storage, float* out,
                                             long ldo, int accumulate, float* dbias, float* scratch, long scratch_floats, void* stream) {
    int splits;
    const bool bf = storage == 3;
    dispu_tn_reduce_desc* sink = tn_take_defer();                 // dispu_tn_defer (train_gemm.hip): describe the reduction, do not launch it
    if (sink) sink->splits = 0;
    if (!(storage == 0 || storage == 3) || !tn_stream_plan(M, K, N, splits) || !X || !Z || !out || !scratch ||
        scratch_floats < (long)splits * ((long)K + 1) * N || ldx < K || ldz < N || (ldx & (bf ? 7 : 3)) || (ldz & (bf ? 7 : 3)) ||
        (((uintptr_t)X) & 15) || (((uintptr_t)Z) & 15))
        return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    const bool wide = (N % 256) == 0;
    const int bn = wide ? 256 : 128, es = bf ? 2 : 4;
    StArgs a{M / splits, K, N, X, ldx, Z, ldz, scratch, dbias ? 1 : 0};
    const size_t lds = (size_t)SB_NST * 32 * (128 + bn) * es;
    static DevOnce attr;
    if (attr.needed()) {
        DISPU_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_tn_stream_kernel<256, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));   // + 4 KB static
        DISPU_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_tn_stream_kernel<128, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));   // + 4 KB static
        DISPU_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_tn_stream_kernel<256, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));   // + 4 KB static
        DISPU_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_tn_stream_kernel<128, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));   // + 4 KB static
        attr.done();
    }
    const dim3 grid((K / 128) * (N / bn), splits), blk(512);
    if (wide && bf) hipLaunchKernelGGL((gemm_bf16_tn_stream_kernel<256, true>), grid, blk, lds, s, a);
    else if (wide) hipLaunchKernelGGL((gemm_bf16_tn_stream_kernel<256, false>), grid, blk, lds, s, a);
    else if (bf) hipLaunchKernelGGL((gemm_bf16_tn_stream_kernel<128, true>), grid, blk, lds, s, a);
    else hipLaunchKernelGGL((gemm_bf16_tn_stream_kernel<128, false>), grid, blk, lds, s, a);
    DISPU_CHECK_LAUNCH();
    const long total = ((long)K + (dbias ? 1 : 0)) * N;
    if (sink) {
        *sink = dispu_tn_reduce_desc{scratch, out, dbias, ldo, total, K, N, splits, K + (dbias ? 1 : 0), accumulate, accumulate, 1, 0};
        return 0;
    }
    hipLaunchKernelGGL(tn_stream_reduce_kernel, dim3((unsigned)((total + 31) / 32)), dim3(256), 0, s, K, N, splits, dbias ? 1 : 0, scratch, out, ldo,
                       accumulate, dbias);
    return (int)hipGetLastError();
}
