// PointShuffle2 "local cell" (Common/ops.py:1055-1067) as ONE kernel on gfx950:
//
//   X1[(i,s), :] = relu(G[j(i,s), :] - A[i, :])          conv0 evaluated per source point (see ps_prep_kernel)
//   X2           = relu(X1 . W1 + b1)                     conv1, 128 -> 128, one row per (point, neighbour) pair
//   wv[(i,s), t] = relu(BN(cxyz . Ww + bw))               weight_net_hidden (ops.py:181-191)
//   F'[i, c*16+t] = sum_s X2[(i,s), c] * wv[(i,s), t]     tf.matmul(grouped_feat^T, weight) (ops.py:1066-1067)
//
// The unfused pipeline (gather_sub_relu -> dispu_linear -> weight_net -> point_matmul) moves the two
// [B*16384, 128] pair tensors through HBM twice each (1.07 GB per step at B = 32).  Here the GEMM's A operand is
// produced on the fly by the tile loader (gathered G rows are L2-resident), the 128x128 accumulator tile never
// leaves registers, and the epilogue contracts it with the per-point 16x16 weight matrix: only F' is written.
// Arithmetic (k-ascending MFMA chain, s-ascending fmaf chain) is identical to the unfused kernels, so the two
// paths are bit-identical (tests/test_generator_gpu.py).
//
// Geometry: workgroup = 4 waves (2x2) = 128 pair rows (8 points x 16 neighbours) x 128 output channels, K = 128
// in 4 slabs of 32 through two LDS stages (same pipeline as linear.hip).
#include "common.h"

namespace dispu {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int PL_BM = 128, PL_BN = 128, PL_BK = 32, PL_K = 128, PL_NT = 256;
constexpr int PL_LDA = PL_BM + 1, PL_LDB = PL_BN + 4;
constexpr int PL_STAGE = PL_BK * (PL_LDA + PL_LDB);                 // floats per stage
constexpr size_t PL_LDS_BYTES = (size_t)(2 * PL_STAGE + 8 * 256) * sizeof(float);   // + wv[8 points][16][16]

__global__ __launch_bounds__(PL_NT) void ps_local_kernel(long npoints, int n_per_cloud, const int* __restrict__ idx,
                                                          const float* __restrict__ xyz, const float* __restrict__ Gm,
                                                          const float* __restrict__ Am, const float* __restrict__ W1,
                                                          const float* __restrict__ b1, const float* __restrict__ Ww,
                                                          const float* __restrict__ bw, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* wv = lds + 2 * PL_STAGE;                                  // [8][16 s][16 t]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const long p0 = (long)blockIdx.x * 8;                             // first point of this workgroup

    // ---- per-point 16x16 weight matrices (weight_net): 2048 values, 8 per thread
#pragma unroll
    for (int e = tid; e < 8 * 256; e += PL_NT) {
        const int t = e & 15, s = (e >> 4) & 15, pl = e >> 8;
        const long i = p0 + pl;
        float v = 0.f;
        if (i < npoints) {
            const long j = (i / n_per_cloud) * n_per_cloud + idx[i * 16 + s];
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) acc = __builtin_fmaf(xyz[j * 3 + c] - xyz[i * 3 + c], Ww[c * 16 + t], acc);
            acc = acc + bw[t];
            acc = acc * scale[t] + shift[t];
            v = fmaxf(acc, 0.f);
        }
        wv[e] = v;
    }

    // ---- A-tile rows owned by this thread: row r = tid/8 + 32*it  (pair (i, s)), k-quad kq = tid % 8
    const int kq = tid & 7;
    long gi[4], gj[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int r = (tid >> 3) + 32 * it;
        long i = p0 + (r >> 4);
        if (i >= npoints) i = npoints - 1;
        gi[it] = i;
        gj[it] = (i / n_per_cloud) * n_per_cloud + idx[i * 16 + (r & 15)];
    }
    float4 pg[4], pa[4], pb[4];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            pg[it] = *reinterpret_cast<const float4*>(Gm + gj[it] * PL_K + k0 + kq * 4);
            pa[it] = *reinterpret_cast<const float4*>(Am + gi[it] * PL_K + k0 + kq * 4);
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int e = tid + it * PL_NT;
            const int kr = e / (PL_BN / 4), nq = e % (PL_BN / 4);
            pb[it] = *reinterpret_cast<const float4*>(W1 + (size_t)(k0 + kr) * PL_BN + nq * 4);
        }
    };
    auto store_tile = [&](int stage) {
        float* As = lds + stage * PL_STAGE;
        float* Bs = As + PL_BK * PL_LDA;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int r = (tid >> 3) + 32 * it;
            As[(kq * 4 + 0) * PL_LDA + r] = fmaxf(pg[it].x - pa[it].x, 0.f);
            As[(kq * 4 + 1) * PL_LDA + r] = fmaxf(pg[it].y - pa[it].y, 0.f);
            As[(kq * 4 + 2) * PL_LDA + r] = fmaxf(pg[it].z - pa[it].z, 0.f);
            As[(kq * 4 + 3) * PL_LDA + r] = fmaxf(pg[it].w - pa[it].w, 0.f);
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int e = tid + it * PL_NT;
            const int kr = e / (PL_BN / 4), nq = e % (PL_BN / 4);
            *reinterpret_cast<float4*>(&Bs[kr * PL_LDB + nq * 4]) = pb[it];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    constexpr int NTILE = PL_K / PL_BK;
    load_tile(0);
    store_tile(0);
    load_tile(PL_BK);
    __syncthreads();
    const int fi = lane & 31, fk = lane >> 5;
#pragma unroll
    for (int t = 0; t < NTILE; ++t) {
        const float* As = lds + (t & 1) * PL_STAGE;
        const float* Bs = As + PL_BK * PL_LDA;
        if (t + 1 < NTILE) {
            store_tile((t + 1) & 1);
            if (t + 2 < NTILE) load_tile((t + 2) * PL_BK);
        }
#pragma unroll
        for (int kk = 0; kk < PL_BK; kk += 2) {
            float af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = As[(kk + fk) * PL_LDA + wm * 64 + i * 32 + fi];
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = Bs[(kk + fk) * PL_LDB + wn * 64 + j * 32 + fi];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue.  32x32 tile (i, j): lane (col = fi, fk), reg r -> row (r&3) + 8(r>>2) + 4fk.  A tile holds two
    // points (rows 0-15 / 16-31 -> regs 0-7 / 8-15); within a point this lane has s in {0-3, 8-11} + 4fk, its
    // partner lane (fk ^ 1) the other eight.  After exchanging them the lane contracts all 16 s in ascending order
    // with wv[p][s][8fk .. 8fk+7] and stores 8 contiguous outputs.
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = wn * 64 + j * 32 + fi;
        const float bias = b1[c];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int pl = wm * 4 + i * 2 + q;                    // point within the workgroup
                float own[8], oth[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    own[u] = fmaxf(acc[i][j][q * 8 + u] + bias, 0.f);
                    oth[u] = __shfl_xor(own[u], 32, 64);
                }
                // own[u] <-> s = (u&3) + 8(u>>2) + 4fk ; oth[u] <-> s = (u&3) + 8(u>>2) + 4(fk^1)
                float o[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) o[t] = 0.f;
                const float* wp = wv + pl * 256 + fk * 8;
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const int u = (s & 3) + 4 * (s >> 3);             // register slot holding this s
                    const bool hi = (s >> 2) & 1;                     // s in the "+4" half
                    const float xv = (hi == (fk != 0)) ? own[u] : oth[u];
                    const float4 w0 = *reinterpret_cast<const float4*>(wp + s * 16);
                    const float4 w1 = *reinterpret_cast<const float4*>(wp + s * 16 + 4);
                    o[0] = __builtin_fmaf(xv, w0.x, o[0]); o[1] = __builtin_fmaf(xv, w0.y, o[1]);
                    o[2] = __builtin_fmaf(xv, w0.z, o[2]); o[3] = __builtin_fmaf(xv, w0.w, o[3]);
                    o[4] = __builtin_fmaf(xv, w1.x, o[4]); o[5] = __builtin_fmaf(xv, w1.y, o[5]);
                    o[6] = __builtin_fmaf(xv, w1.z, o[6]); o[7] = __builtin_fmaf(xv, w1.w, o[7]);
                }
                const long pi = p0 + pl;
                if (pi < npoints) {
                    float* dst = out + pi * 2048 + c * 16 + fk * 8;
                    *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
                    *reinterpret_cast<float4*>(dst + 4) = make_float4(o[4], o[5], o[6], o[7]);
                }
            }
        }
    }
}

}  // namespace dispu

using namespace dispu;

DISPU_EXPORT int dispu_ps_local(long npoints, int n_per_cloud, int k, int c, const int* idx, const float* xyz, const float* G,
                                const float* A, const float* W1, const float* b1, const float* Ww, const float* bw,
                                const float* scale, const float* shift, float* out, void* stream) {
    if (npoints < 0 || n_per_cloud <= 0 || k != 16 || c != 128) return (int)hipErrorInvalidValue;
    if ((((uintptr_t)G) | ((uintptr_t)A) | ((uintptr_t)W1) | ((uintptr_t)out)) & 15) return (int)hipErrorInvalidValue;
    if (npoints == 0) return 0;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ps_local_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)PL_LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    hipLaunchKernelGGL(ps_local_kernel, dim3((unsigned)((npoints + 7) / 8)), dim3(PL_NT), PL_LDS_BYTES, (hipStream_t)stream, npoints,
                       n_per_cloud, idx, xyz, G, A, W1, b1, Ww, bw, scale, shift, out);
    return (int)hipGetLastError();
}
