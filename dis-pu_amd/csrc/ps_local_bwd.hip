// Backward of PointShuffle2's local cell (Common/ops.py:1055-1067; forward: csrc/ps_local.hip) as ONE recomputing kernel, gfx950 fp32 MFMA.
//
//   forward   h0[(i,s),:] = relu(G[j(i,s),:] - A[i,:])        h1 = relu(h0 . W1 + b1)        wv[(i,s),:] = relu(BN(offsets . Ww + bw))
//             F'[i, c*16+t] = sum_s h1[(i,s),c] * wv[(i,s),t]
//   backward  given dF' [rows, 2048]:
//             dwv[(i,s),t] = sum_c dF'[i,c,t] h1[(i,s),c]                                  -> the weight net's backward (dispu_ps_wnet_grad)
//             dz1[(i,s),c] = (sum_t dF'[i,c,t] wv[(i,s),t]) [h1 > 0]                       -> HBM: the side-stream dW1 = h0^T . dz1 reads it
//             dz0          = (dz1 . W1^T) [h0 > 0]
//             dG[j]       += dz0[(i,s)]  (float atomics, like the reference's GroupPointGrad)   dAneg[i] = -sum_s dz0[(i,s)]
//
// TF autodiff gives the reference these gradients (DisPU/model.py:178); the unfused path here is five launches around four [rows*16, 128]
// pair tensors (h1 and dz0 written and read back, dz1 read twice, the weights wv, the inverted neighbour graph): at 8 training patches
// ~300 us on the step's critical chain for ~14 GFLOP.  Here a workgroup of four waves owns 4 points = 64 pair rows and keeps everything on
// chip in 46 KB of LDS, so that three workgroups share a CU and one's gathers / contractions overlap another's MFMAs:
//   T  [128 k][65]  h0 (k-major: the MFMA A operand of conv1), then -- its ReLU decisions saved as one mask register per lane -- h1 (for the
//                   dwv sums), then dz1 (the A operand of the dz1 . W1^T product)                                              33 KB
//   two 8-row weight slabs (W1, then W1^T), the 4 x 16 x 16 weight-net outputs, the 64 neighbour rows                           13 KB
// MEASURED (round 6, tools/debug/ps_local_grad_bench.py): 158 us at 8 patches / 635 us at 32 against 126 / 465 us for the three chain launches it
// replaces (+ 94 / 238 us of side-stream recomputation it makes unnecessary); inside the 8-patch step it is 40 us SLOWER than the five-launch
// path, so Trainer.fused_local_bwd is OFF by default.  Lab switches (LB_NO_*): the two products + gathers 324 us of the 635 (219 us of pure
// matrix-pipe time), the dG atomics 86, the dz1 stores ~70, the contractions ~60; 16-row or 32-row weight slabs with two workgroups per CU
// are slower than 8-row slabs with three.  What it lacks is what ps_local_ws_kernel has: helper waves that overlap the gathers / contractions
// of one group with the MFMAs of another inside one workgroup.
// conv1 is recomputed with the forward's own arithmetic (v_mfma_f32_32x32x2_f32 over ascending k = dispu_linear's chain), so h1, its mask
// and dz1 are bit-identical to the unfused path's; dz0's product likewise; dwv / dAneg are re-associated sums, dG is accumulated with atomics
// (tolerance-checked: tests/test_train_fused_gpu.py).
#include "common.h"

#ifdef LB_NO_DZ1C
#define LB_CONTRACT 0
#else
#define LB_CONTRACT 1
#endif

namespace dispu {

typedef float lb_f32x16 __attribute__((ext_vector_type(16)));

constexpr int LB_NT = 256, LB_K = 128, LB_PTS = 4, LB_ROWS = 16 * LB_PTS, LB_LDA = LB_ROWS + 1, LB_BK = 8, LB_LDB = LB_K + 4;
constexpr int LB_TILE = LB_K * LB_LDA;                         // floats per k-major tile
constexpr int LB_BST = LB_BK * LB_LDB;                         // floats per weight slab stage
constexpr size_t LB_LDS_BYTES = (size_t)(LB_TILE + 2 * LB_BST + LB_PTS * 256 + LB_ROWS) * sizeof(float);
static_assert(3 * LB_LDS_BYTES <= 160 * 1024, "the local cell's backward: three workgroups per CU");

struct LbArgs {
    long npoints; int n_per_cloud;
    const int* idx;                        // [npoints, 16] cloud-local neighbour ids
    const float* xyz;                      // [npoints, 3]
    const float* Gm; long ldg;             // [npoints, 128]
    const float* Am;                       // [npoints, 128]
    const float* W1; const float* b1;      // conv1 [128 k][128 c]
    const float* W1t;                      // its transpose [128 c][128 k], row-major
    const float* Ww; const float* bw;      // weight net 3 -> 16
    const float* scale; const float* shift;   // its BatchNorm folded to v * scale + shift (training: this step's batch statistics)
    const float* dF;                       // [npoints, 2048]
    float* dz1;                            // [npoints * 16, 128]
    float* dwv;                            // [npoints * 16, 16]
    float* dG;                             // [npoints, 128], accumulated (zeroed by the caller)
    float* dAneg;                          // [npoints, 128]
};

// acc[j] += At (k-major tile, LDS) . B (row-major [128 k][128 n] in global memory, streamed through two LDS stages of LB_BK rows).
// 2 x 2 waves over 64 x 128 (32 rows x 64 columns each); ascending k.  Ends with every wave past its last fragment read (barrier).
__device__ __forceinline__ void lb_gemm128(const float* __restrict__ At, const float* __restrict__ Bg, float* __restrict__ bst, lb_f32x16 (&acc)[2],
                                           int tid, int wm, int wn, int fi, int fk) {
    constexpr int NF4 = LB_BK * LB_K / 4 / LB_NT;                  // float4 per thread and slab: rows tid / 32 + 8 u, quad tid % 32
    float4 pb[NF4];
    auto load_slab = [&](int t) {
#pragma unroll
        for (int u = 0; u < NF4; ++u) pb[u] = *reinterpret_cast<const float4*>(Bg + (size_t)(t * LB_BK + (tid >> 5) + 8 * u) * LB_K + (tid & 31) * 4);
    };
    auto store_slab = [&](int stage) {
#pragma unroll
        for (int u = 0; u < NF4; ++u) *reinterpret_cast<float4*>(&bst[stage * LB_BST + ((tid >> 5) + 8 * u) * LB_LDB + (tid & 31) * 4]) = pb[u];
    };
    constexpr int NSLAB = LB_K / LB_BK;
    load_slab(0);
    store_slab(0);
    load_slab(1);
    __syncthreads();
#pragma unroll 1
    for (int t = 0; t < NSLAB; ++t) {
        const float* bs = bst + (t & 1) * LB_BST;
        if (t + 1 < NSLAB) {
            store_slab((t + 1) & 1);
            if (t + 2 < NSLAB) load_slab(t + 2);
        }
        const float* as = At + (t * LB_BK) * LB_LDA;
#pragma unroll
        for (int kk = 0; kk < LB_BK; kk += 2) {
            float bf[2];
            const float af = as[(kk + fk) * LB_LDA + wm * 32 + fi];
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = bs[(kk + fk) * LB_LDB + wn * 64 + j * 32 + fi];
#pragma unroll
#ifndef LB_NO_MFMA
            for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf[j], acc[j], 0, 0, 0);
#else
            for (int j = 0; j < 2; ++j) acc[j][0] += af * bf[j];
#endif
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(LB_NT, 3) void ps_local_bwd_kernel(LbArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lb_lds[];
    float* A0 = lb_lds;                                           // h0, k-major ...
    float* X = A0;                                                // ... then h1, then dz1 (k = c), in the same tile
    float* bst = A0 + LB_TILE;
    float* wv = bst + 2 * LB_BST;                                 // [4][16 s][16 t]
    int* nbr = reinterpret_cast<int*>(wv + LB_PTS * 256);         // global row of the neighbour of pair row r
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, fi = lane & 31, fk = lane >> 5;     // wm 0 .. 1: rows 32 wm .. + 31 = points 2 wm, 2 wm + 1
    const long p0 = (long)blockIdx.x * LB_PTS;

    // ---- neighbour rows and the weight net of the group's 8 points (the forward's arithmetic: csrc/ps_local.hip)
    if (tid < LB_ROWS) {
        const long i = p0 + (tid >> 4);
        nbr[tid] = (int)((i / a.n_per_cloud) * a.n_per_cloud + a.idx[i * 16 + (tid & 15)]);
    }
    __syncthreads();
#pragma unroll
    for (int e = tid; e < LB_PTS * 256; e += LB_NT) {
        const int t = e & 15, s = (e >> 4) & 15, pl = e >> 8;
        const long i = p0 + pl;
        const long j = nbr[pl * 16 + s];
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) acc = __builtin_fmaf(a.xyz[j * 3 + c] - a.xyz[i * 3 + c], a.Ww[c * 16 + t], acc);
        acc = acc + a.bw[t];
        acc = acc * a.scale[t] + a.shift[t];
        wv[e] = fmaxf(acc, 0.f);
    }
    // ---- h0 tile: row r = tid / 8 + 32 it (pair (i, s)), k-quad kq = tid % 8 of every 32-column pass
    {
        const int kq = tid & 7;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int r = (tid >> 3) + 32 * it;
            const long i = p0 + (r >> 4);
            const long j = nbr[r];
#pragma unroll
            for (int k0 = 0; k0 < LB_K; k0 += 32) {
                const float4 g = *reinterpret_cast<const float4*>(a.Gm + j * a.ldg + k0 + kq * 4);
                const float4 q = *reinterpret_cast<const float4*>(a.Am + i * LB_K + k0 + kq * 4);
                A0[(k0 + kq * 4 + 0) * LB_LDA + r] = fmaxf(g.x - q.x, 0.f);
                A0[(k0 + kq * 4 + 1) * LB_LDA + r] = fmaxf(g.y - q.y, 0.f);
                A0[(k0 + kq * 4 + 2) * LB_LDA + r] = fmaxf(g.z - q.z, 0.f);
                A0[(k0 + kq * 4 + 3) * LB_LDA + r] = fmaxf(g.w - q.w, 0.f);
            }
        }
    }
    // (lb_gemm128 begins with a barrier behind its first slab: A0, wv and nbr are complete when the first fragment is read)

    // ---- conv1 again: h1 = relu(h0 . W1 + b1), kept in registers and parked in X for the dwv sums
    lb_f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    lb_gemm128(A0, a.W1, bst, acc, tid, wm, wn, fi, fk);
    const int rb = wm * 32 + 4 * fk;                              // the lane's first row; register r: + (r & 3) + 8 (r >> 2)
    // h0's ReLU decisions for the elements this lane will own in the dz0 product (row rb + .., column k0 = this lane's c): 32 bits; then
    // the tile is free for h1 (lb_gemm128 ended with a barrier: every wave is past its last fragment read)
    unsigned hmask = 0u;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float* __restrict__ hs = A0 + (wn * 64 + j * 32 + fi) * LB_LDA + rb;
#pragma unroll
        for (int r = 0; r < 16; ++r) hmask |= (hs[(r & 3) + 8 * (r >> 2)] > 0.f) ? (1u << (j * 16 + r)) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = wn * 64 + j * 32 + fi;
        const float bias = a.b1[c];
        float* __restrict__ xs = X + c * LB_LDA + rb;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float h = fmaxf(acc[j][r] + bias, 0.f);
            acc[j][r] = h;
            xs[(r & 3) + 8 * (r >> 2)] = h;
        }
    }
    __syncthreads();

    // ---- dwv[(pl, s), t] = sum_c dF'[pl, c, t] h1[(pl, s), c] on the matrix pipe: wave w owns point w, a [16 s x 128 c] x [128 c x 16 t] product as
    //      32 v_mfma_f32_16x16x4_f32 (= the ascending-c fmaf chain): A = h1 from the tile, B = dF' straight from global memory (a lane's 32
    //      operands requested up front: 256 contiguous bytes per instruction).  (As a VALU loop with the dF' loads inside -- 128 dependent
    //      L2 round trips per thread -- this was 37 % of the kernel.)
#ifndef LB_NO_DWV
    {
        typedef float lb_f32x4 __attribute__((ext_vector_type(4)));
        const int pl = wave, i16 = lane & 15, q4 = lane >> 4;
        const float* __restrict__ df = a.dF + (p0 + pl) * 2048 + q4 * 16 + i16;
        const float* __restrict__ xr = X + q4 * LB_LDA + pl * 16 + i16;
        float bv[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) bv[u] = df[u * 64];
        lb_f32x4 dw = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 32; ++u) dw = __builtin_amdgcn_mfma_f32_16x16x4f32(xr[u * 4 * LB_LDA], bv[u], dw, 0, 0, 0);
        float* __restrict__ o = a.dwv + ((p0 + pl) * 16 + 4 * q4) * 16 + i16;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r * 16] = dw[r];
    }
#endif

    // ---- dz1 = (sum_t dF'[pl, c, t] wv[pl, s, t]) [h1 > 0]  (t ascending: the chain of ps_point_matmul_grad_relu), in place of h1
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = wn * 64 + j * 32 + fi;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int pl = wm * 2 + q;
            const float* __restrict__ df = a.dF + (p0 + pl) * 2048 + c * 16;
            const float4 d0 = *reinterpret_cast<const float4*>(df), d1 = *reinterpret_cast<const float4*>(df + 4),
                         d2 = *reinterpret_cast<const float4*>(df + 8), d3 = *reinterpret_cast<const float4*>(df + 12);
#pragma unroll
            for (int uu = 0; uu < (LB_CONTRACT ? 8 : 0); ++uu) {
                const int s = (uu & 3) + 8 * (uu >> 2) + 4 * fk;
                const float4* w4 = reinterpret_cast<const float4*>(wv + pl * 256 + s * 16);
                const float4 w0 = w4[0], w1 = w4[1], w2 = w4[2], w3 = w4[3];
                float v = 0.f;
                v = __builtin_fmaf(d0.x, w0.x, v); v = __builtin_fmaf(d0.y, w0.y, v); v = __builtin_fmaf(d0.z, w0.z, v); v = __builtin_fmaf(d0.w, w0.w, v);
                v = __builtin_fmaf(d1.x, w1.x, v); v = __builtin_fmaf(d1.y, w1.y, v); v = __builtin_fmaf(d1.z, w1.z, v); v = __builtin_fmaf(d1.w, w1.w, v);
                v = __builtin_fmaf(d2.x, w2.x, v); v = __builtin_fmaf(d2.y, w2.y, v); v = __builtin_fmaf(d2.z, w2.z, v); v = __builtin_fmaf(d2.w, w2.w, v);
                v = __builtin_fmaf(d3.x, w3.x, v); v = __builtin_fmaf(d3.y, w3.y, v); v = __builtin_fmaf(d3.z, w3.z, v); v = __builtin_fmaf(d3.w, w3.w, v);
                const int r = q * 8 + uu;
                acc[j][r] = (acc[j][r] > 0.f) ? v : 0.f;
            }
        }
    }
    __syncthreads();                                              // every dwv sum has read h1 from X
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = wn * 64 + j * 32 + fi;
        float* __restrict__ xs = X + c * LB_LDA + rb;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ro = (r & 3) + 8 * (r >> 2);
            const float v = acc[j][r];
            xs[ro] = v;

            acc[j][r] = 0.f;
        }
    }
    __syncthreads();
    // dz1 leaves as 16-byte stores from the tile (a wave = two rows x 512 contiguous bytes): as 4-byte stores from the accumulator layout it
    // was 18 % of the kernel
#ifndef LB_NO_DZ1ST
    {
        const int quad = tid & 31;
        float* __restrict__ zs = a.dz1 + (p0 * 16) * LB_K + quad * 4;
#pragma unroll
        for (int it = 0; it < LB_ROWS / 8; ++it) {
            const int row = (tid >> 5) + 8 * it;
            const float* __restrict__ xs = X + (quad * 4) * LB_LDA + row;
            *reinterpret_cast<float4*>(zs + row * LB_K) = make_float4(xs[0], xs[LB_LDA], xs[2 * LB_LDA], xs[3 * LB_LDA]);
        }
    }
#endif

    // ---- dz0 = (dz1 . W1^T) [h0 > 0]; dAneg = - row sums per point; dG by atomics
    lb_gemm128(X, a.W1t, bst, acc, tid, wm, wn, fi, fk);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int k0 = wn * 64 + j * 32 + fi;
        const int* __restrict__ nb = nbr + rb;
        float psum[2] = {0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ro = (r & 3) + 8 * (r >> 2);
            float v = acc[j][r];
            if (!((hmask >> (j * 16 + r)) & 1u)) v = 0.f;
            psum[r >> 3] += v;
#ifndef LB_NO_ATOM
            if (v != 0.f) unsafeAtomicAdd(a.dG + (size_t)nb[ro] * LB_K + k0, v);
#endif
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float tot = psum[q] + __shfl_xor(psum[q], 32, 64);
            if (fk == 0) a.dAneg[(p0 + wm * 2 + q) * LB_K + k0] = -tot;
        }
    }
}

}  // namespace dispu

using namespace dispu;

// Backward of dispu_ps_local (see the header of this file).  k = 16 neighbours, c = 128 channels, t = 16 weights: the shipped graph's
// shape.  npoints % 4 == 0 (a workgroup owns 4 points); dG must be zero on entry (it is accumulated with atomics); every pointer 16-byte
// aligned, ldg % 4 == 0.
DISPU_EXPORT int dispu_ps_local_grad(long npoints, int n_per_cloud, const int* idx, const float* xyz, const float* Gm, long ldg, const float* Am,
                                     const float* W1, const float* b1, const float* W1t, const float* Ww, const float* bw, const float* scale,
                                     const float* shift, const float* dF, float* dz1, float* dwv, float* dG, float* dAneg, void* stream) {
    if (npoints < 0 || (npoints % LB_PTS) || n_per_cloud <= 0 || !idx || !xyz || !Gm || !Am || !W1 || !b1 || !W1t || !Ww || !bw || !scale || !shift || !dF || !dz1 ||
        !dwv || !dG || !dAneg || (ldg & 3) ||
        ((((uintptr_t)Gm) | ((uintptr_t)Am) | ((uintptr_t)W1) | ((uintptr_t)W1t) | ((uintptr_t)dF) | ((uintptr_t)dwv)) & 15))
        return (int)hipErrorInvalidValue;
    if (npoints == 0) return 0;
    if (npoints / LB_PTS > 0x7fffffffl) return (int)hipErrorInvalidValue;
    static DevOnce attr;
    if (attr.needed()) {
        DISPU_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(ps_local_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LB_LDS_BYTES));
        attr.done();
    }
    LbArgs a{npoints, n_per_cloud, idx, xyz, Gm, ldg, Am, W1, b1, W1t, Ww, bw, scale, shift, dF, dz1, dwv, dG, dAneg};
    hipLaunchKernelGGL(ps_local_bwd_kernel, dim3((unsigned)(npoints / LB_PTS)), dim3(LB_NT), LB_LDS_BYTES, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}
