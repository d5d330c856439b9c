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

#include <cstdlib>

namespace dispu {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int PL_BM = 128, PL_BN = 128, PL_BK = 32, PL_K = 128, PL_NT = 256;
constexpr int PL_LDA = PL_BM + 1, PL_LDB = PL_BN + 4;
constexpr int PL_STAGE = PL_BK * (PL_LDA + PL_LDB);                 // floats per stage
constexpr size_t PL_LDS_BYTES = (size_t)(2 * PL_STAGE + 8 * 256) * sizeof(float);   // + wv[8 points][16][16]

__global__ __launch_bounds__(PL_NT) void ps_local_kernel(long npoints, int n_per_cloud, const int* __restrict__ idx,
                                                          const float* __restrict__ xyz, const float* __restrict__ Gm, long ldg,
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
            pg[it] = *reinterpret_cast<const float4*>(Gm + gj[it] * ldg + k0 + kq * 4);
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


// ---------------------------------------------------------------------------------------------------------------
// Wave-specialised, persistent formulation (default).  The single-role kernel above keeps the matrix pipe ~39 % busy:
// every workgroup pays the index / row gather latencies, the 16 x 16 weight-net evaluation and the VALU contraction
// epilogue in the same waves that issue the MFMAs, and streams 192 KB per 8 points through the CU's load path
// (16 x redundant A rows, W1 again for every group).  Here a workgroup of 8 waves loops over point groups:
//   * waves 0-3 (MFMA waves) issue fragment reads + MFMAs only: conv1 from the staged A slabs and the LDS-RESIDENT W1,
//     then the per-point contraction F'[c][t] = sum_s X2[s][c] * wv[s][t] on the matrix pipe as well, two points at a
//     time as a [32 ch x 32] x [32 x (2 x 16)] product with a block-diagonal B (point q's weights in the rows of
//     point q, zeros elsewhere);
//   * waves 4-7 (helpers) gather the neighbour rows, form relu(G_j - A_i) into the next A slab, and evaluate the next
//     group's weight net.
// The pair rows of every 32-row block are laid out in the order the MFMA result layout dictates (tile row
// (k>>1 & 3) + 8 (k>>3) + 4 (k & 1) holds contraction index k = 16 * point + s), so relu(acc + b1) is already the A
// operand of the contraction in the lane that holds it: the [128 x 128] pair tensor never leaves the registers.
// k ascends through (point, s) and the foreign point contributes exact zeros, so every output is the same
// s-ascending fmaf chain as above -> bit-identical output.
template <int CTRL>
__device__ __forceinline__ float pl_quad(float v) {      // v from the lane given by the quad permutation CTRL
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

// Barrier of the slab pipeline: the waves exchange data through LDS only, so only the LDS counter has to drain.  __syncthreads()
// also waits for vmcnt(0): the helpers would sit out the latency of the gathers they have just issued for the NEXT slab (and the
// MFMA waves that of their F' stores) at every barrier -- measured 900 cycles per barrier on the MFMA waves (PL_STAMPS).
__device__ __forceinline__ void pl_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// One A slab, ROW-major [128 pair rows][32 k (+4)]: a helper thread owns 4 consecutive k of a row = ONE 16-byte LDS store (the
// k-major slab of round 1 needed four 4-byte stores, 8-way bank conflicted); the helpers share their SIMD's issue port with a
// wave that issues MFMAs back to back, so their instruction count is what the MFMA waves end up waiting for (PL_STAMPS: 900
// cycles per barrier before).  A 16x16x4 fragment read (16 rows x 4 consecutive k) is 2-way conflicted at the 36-float pitch,
// the resident W1 (pitch 144: k-groups 16 banks apart) likewise: 64 lanes x 4 B over 32 banks cannot do better.
constexpr int PW_LDAR = PL_BK + 4;
constexpr int PW_ASTG = PL_BM * PW_LDAR;
constexpr int PW_LDB = PL_BN + 16;
constexpr int PW_WRES = PL_K * PW_LDB;                               // resident W1 [128 k][144]
constexpr int PW_FLOATS = PW_WRES + 2 * PW_ASTG + 2 * 2048 + 2 * 1024 + 512;
constexpr size_t PW_LDS_BYTES = (size_t)PW_FLOATS * sizeof(float);

// UNI (round 5): n_per_cloud % 8 == 0 and npoints % 8 == 0 -- every 8-point group lies inside one cloud and none is ragged.  The helpers'
// instruction count is what the MFMA waves end up waiting for (PL_STAMPS: 1450 - 2600 cycles per group at the slab barriers): with UNI
// a group's cloud base is ONE scalar division (it was an unsigned vector division, ~25 instructions, per gathered row), every index /
// coordinate / A-row / G-row load is a buffer load whose group offset is a scalar, and nothing is clamped.
template <bool UNI>
__global__ __launch_bounds__(512) void ps_local_ws_kernel(long npoints, int n_per_cloud, const int* __restrict__ idx,
                                                           const float* __restrict__ xyz, const float* __restrict__ Gm, int ldg,
                                                           const float* __restrict__ Am, const float* __restrict__ W1,
                                                           const float* __restrict__ b1, const float* __restrict__ Ww,
                                                           const float* __restrict__ bw, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* wres = lds;                                               // [128][PW_LDB]
    float* astg = wres + PW_WRES;                                    // [2][128][PW_LDAR]
    float* wvbuf = astg + 2 * PW_ASTG;                               // [2][8 points][16 s][16 t]
    float* abuf = wvbuf + 2 * 2048;                                  // [2][8 points][128]
    float* cxbuf = abuf + 2 * 1024;                                  // [128 pairs][4]: x_j - x_i
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int np = (int)npoints;
    // group -> workgroup: workgroup ids go round-robin over the 8 XCDs; with group = id + n * grid every XCD touched every cloud and each
    // L2 fetched every cloud's G / A rows.  When the counts divide, XCD x (ids = x mod 8) walks ONE contiguous eighth of the groups --
    // whole clouds -- with its own workgroups striding through it.
    const int ng_all = (np + 7) / 8;
    const bool xcd_map = ((gridDim.x & 7u) == 0) && (ng_all % 8 == 0) && ((int)gridDim.x <= ng_all);
    const int gstep = xcd_map ? (int)(gridDim.x >> 3) : (int)gridDim.x;
    const int g_first = xcd_map ? (int)(blockIdx.x & 7u) * (ng_all / 8) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int ng = xcd_map ? ((int)(blockIdx.x & 7u) + 1) * (ng_all / 8) : ng_all;      // end of this workgroup's range
    if (g_first >= ng) return;
    constexpr int NTILE = PL_K / PL_BK;                              // 4 slabs of 32

    if (wave >= 4) {
        // ------------------------------------------------------------------------------------- helper waves
        const int ht = threadIdx.x - 256;
        const int kq = ht & 7;
        // A-tile rows of this thread: r = rho + 32 * it, rho = ht >> 3.  The 16 tile rows of a point hold its neighbours in the
        // order the 16x16 MFMA result layout dictates: row 4 a + b of the block <-> s = 4 b + a (see the MFMA waves).
        const int rho = ht >> 3;
        const int rq = rho >> 4, rs = 4 * (rho & 3) + ((rho >> 2) & 3);
        auto clampi = [&](int i) { return (UNI || i < np) ? i : np - 1; };
        auto cloud_base = [&](int i) { return (int)((unsigned)i / (unsigned)n_per_cloud) * n_per_cloud; };
        // UNI: the cloud base of group g as a scalar
        auto cloud_base_g = [&](int g) { return __builtin_amdgcn_readfirstlane((int)((unsigned)(g * 8) / (unsigned)n_per_cloud) * n_per_cloud); };
        const __amdgpu_buffer_rsrc_t r_idx = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(idx), 0, (int)((unsigned)np * 64u), 0x00020000);
        const __amdgpu_buffer_rsrc_t r_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Gm), 0, (int)((unsigned)np * (unsigned)ldg * 4u), 0x00020000);
        const __amdgpu_buffer_rsrc_t r_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Am), 0, (int)((unsigned)np * 512u), 0x00020000);
        const __amdgpu_buffer_rsrc_t r_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xyz), 0, (int)((unsigned)np * 12u), 0x00020000);
        auto bload4 = [](__amdgpu_buffer_rsrc_t rs, int voff, int soff) -> float4 {
            const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
            return make_float4(__uint_as_float(r[0]), __uint_as_float(r[1]), __uint_as_float(r[2]), __uint_as_float(r[3]));
        };
        int goff[4], goff_n[4];                                      // offsets of the gathered G rows (+ kq * 4): elements; UNI: bytes
        int ioff[4];                                                 // UNI: byte offset of this thread's four neighbour ids inside a group's [8][16] block
#pragma unroll
        for (int it = 0; it < 4; ++it) ioff[it] = ((2 * it + rq) * 16 + rs) * 4;
        auto rows_of = [&](int g, int (&go)[4]) {
            if constexpr (UNI) {
                const int cb = cloud_base_g(g);
#pragma unroll
                for (int it = 0; it < 4; ++it)
                    go[it] = ((cb + (int)__builtin_amdgcn_raw_buffer_load_b32(r_idx, ioff[it], g * 512, 0)) * ldg + kq * 4) * 4;
            } else {
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int i = clampi(g * 8 + 2 * it + rq);
                    go[it] = (cloud_base(i) + idx[i * 16 + rs]) * ldg + kq * 4;
                }
            }
        };
        int ri[4];                                                   // neighbour ids of the next group: loaded one interval before use
        auto rows_load = [&](int g) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                if constexpr (UNI) ri[it] = (int)__builtin_amdgcn_raw_buffer_load_b32(r_idx, ioff[it], g * 512, 0);
                else ri[it] = idx[clampi(g * 8 + 2 * it + rq) * 16 + rs];
            }
        };
        auto rows_finish = [&](int g, int (&go)[4]) {
            if constexpr (UNI) {
                const int cb = cloud_base_g(g);
#pragma unroll
                for (int it = 0; it < 4; ++it) go[it] = ((cb + ri[it]) * ldg + kq * 4) * 4;
            } else {
#pragma unroll
                for (int it = 0; it < 4; ++it) go[it] = (cloud_base(clampi(g * 8 + 2 * it + rq)) + ri[it]) * ldg + kq * 4;
            }
        };
        // two register sets: the gathers of a slab are issued TWO slab intervals before they are consumed (one interval =
        // 4300 cycles of MFMAs; a gather from the 17 MB working set of G takes about that long: with one set the helpers spent the
        // whole interval waiting in store_a and the MFMA waves 900 cycles at every barrier, PL_STAMPS)
        float4 pg0[4], pg1[4];
        auto load_g = [&](float4 (&pg)[4], int k0, const int (&go)[4]) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                if constexpr (UNI) pg[it] = bload4(r_g, go[it], k0 * 4);
                else pg[it] = *reinterpret_cast<const float4*>(Gm + go[it] + k0);
            }
        };
        auto store_a = [&](const float4 (&pg)[4], int stage, int k0, const float* ab) {     // relu(G_j - A_i) -> slab `stage`
            float* As = astg + stage * PW_ASTG;
            // the four A-row reads go out together, ahead of the arithmetic: one LDS round trip per slab instead of four in a row (the
            // compiler's order was read, wait, subtract, store per row)
            float4 avs[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) avs[it] = *reinterpret_cast<const float4*>(ab + (2 * it + rq) * PL_K + k0 + kq * 4);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const float4 av = avs[it];
                const int r = rho + 32 * it;
                const f32x2 dlo = f32x2{pg[it].x, pg[it].y} - f32x2{av.x, av.y};      // v_pk_add_f32 with neg
                const f32x2 dhi = f32x2{pg[it].z, pg[it].w} - f32x2{av.z, av.w};
                *reinterpret_cast<float4*>(&As[r * PW_LDAR + kq * 4]) =
                    make_float4(fmaxf(dlo.x, 0.f), fmaxf(dlo.y, 0.f), fmaxf(dhi.x, 0.f), fmaxf(dhi.y, 0.f));
            }
        };
        // A rows of the 8 points of a group: 8 x 128 floats = 256 float4, one per helper thread
        float4 arow;
        auto load_arow = [&](int g) {
            if constexpr (UNI) {
                arow = bload4(r_a, ht * 16, g * (8 * PL_K * 4));          // the group's 8 rows are 256 consecutive float4
            } else {
                const int i = clampi(g * 8 + (ht >> 5));
                arow = *reinterpret_cast<const float4*>(Am + (size_t)i * PL_K + (ht & 31) * 4);
            }
        };
        auto store_arow = [&](float* ab) { *reinterpret_cast<float4*>(ab + (ht >> 5) * PL_K + (ht & 31) * 4) = arow; };
        // weight net, staged: (a) neighbour id of pair ht (threads < 128), (b) its coordinates, (c) x_j - x_i -> cxbuf,
        // (d) every thread: wv[u][s][t] for its (s, t) and the 8 points
        const int wt = ht & 15, wsn = (ht >> 4) & 15;
        const float ww0 = Ww[wt], ww1 = Ww[16 + wt], ww2 = Ww[32 + wt], wbw = bw[wt], wsc = scale[wt], wsh = shift[wt];
        int pj = 0, pi_ = 0;
        float px[6];
        auto wn_a = [&](int g) {
            if (ht < 128) {
                if constexpr (UNI) {
                    pi_ = g * 8 + (ht >> 4);
                    pj = cloud_base_g(g) + (int)__builtin_amdgcn_raw_buffer_load_b32(r_idx, ht * 4, g * 512, 0);
                } else {
                    pi_ = clampi(g * 8 + (ht >> 4));
                    pj = cloud_base(pi_) + idx[pi_ * 16 + (ht & 15)];
                }
            }
        };
        auto wn_b = [&]() {
            if (ht < 128) {
                if constexpr (UNI) {
                    const int oj = pj * 12, oi = pi_ * 12;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        px[c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_x, oj, c * 4, 0));
                        px[3 + c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_x, oi, c * 4, 0));
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < 3; ++c) { px[c] = xyz[pj * 3 + c]; px[3 + c] = xyz[pi_ * 3 + c]; }
                }
            }
        };
        auto wn_c = [&]() {
            if (ht < 128) {
#pragma unroll
                for (int c = 0; c < 3; ++c) cxbuf[ht * 4 + c] = px[c] - px[3 + c];
                cxbuf[ht * 4 + 3] = 0.f;                              // read as part of a float4
            }
        };
        // wn_d(g, wv, U0, U1): points U0 .. U1 - 1 of the group.  Round 5: the eight points are spread over three slab intervals (the
        // last one of the previous group and the first two of the group itself: cxbuf is rewritten in the third, the contraction
        // reads wv after the fourth) -- all eight in one interval made that interval's helper work twice the others' and the MFMA
        // waves waited 1.5 - 2 k cycles per group at its barrier (PL_STAMPS)
        auto wn_d = [&](int g, float* wv, int U0, int U1) {
#pragma unroll
            for (int u = U0; u < U1; ++u) {
                float v = 0.f;
                if (UNI || g * 8 + u < np) {
                    const float4 cx = *reinterpret_cast<const float4*>(cxbuf + (u * 16 + wsn) * 4);
                    float acc = 0.f;
                    acc = __builtin_fmaf(cx.x, ww0, acc);
                    acc = __builtin_fmaf(cx.y, ww1, acc);
                    acc = __builtin_fmaf(cx.z, ww2, acc);
                    acc = acc + wbw;
                    acc = acc * wsc + wsh;
                    v = fmaxf(acc, 0.f);
                }
                wv[ht + u * PL_NT] = v;
            }
        };

        // ---- prologue: W1 into LDS for the lifetime of the workgroup; first group's operands
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int e = ht + it * PL_NT;                            // 4096 float4
            const int kr = e >> 5, nq = e & 31;
            *reinterpret_cast<float4*>(&wres[kr * PW_LDB + nq * 4]) = *reinterpret_cast<const float4*>(W1 + (size_t)kr * PL_BN + nq * 4);
        }
        int g = g_first;
        rows_of(g, goff);
        load_arow(g);
        store_arow(abuf);
        wn_a(g);
        wn_b();
        wn_c();
        load_g(pg0, 0, goff);
        load_g(pg1, PL_BK, goff);
        __syncthreads();                                             // #1: abuf, cxbuf, wres visible
        wn_d(g, wvbuf, 0, 8);
        store_a(pg0, 0, 0, abuf);
        load_g(pg0, 2 * PL_BK, goff);
        __syncthreads();                                             // #2: slab 0 and wv of the first group ready
        int n = 0;
#ifdef PL_STAMPS
        unsigned long long h_sa = 0, h_rest = 0, h_bar = 0;
#define PL_H(v) const unsigned long long v = __builtin_readcyclecounter()
#else
#define PL_H(v)
#endif
        for (; g < ng; g += gstep, ++n) {
            const int gn = g + gstep;
            const bool has_next = gn < ng;
            const float* ab = abuf + (n & 1) * 1024;
            float* ab_n = abuf + ((n + 1) & 1) * 1024;
            // slab 0 interval
            PL_H(a0);
            store_a(pg1, 1, PL_BK, ab);
            PL_H(a1);
            load_g(pg1, 3 * PL_BK, goff);
            if (n > 0) wn_d(g, wvbuf + (n & 1) * 2048, 3, 6);         // this group's weight net, second part (cxbuf still holds its offsets)
            if (has_next) { rows_load(gn); load_arow(gn); wn_a(gn); }
            PL_H(a2);
            pl_lds_barrier();
            // slab 1 interval
            PL_H(b0);
            store_a(pg0, 0, 2 * PL_BK, ab);
            PL_H(b1);
            if (n > 0) wn_d(g, wvbuf + (n & 1) * 2048, 6, 8);         // ... third part; cxbuf is free for the next group from here on
            if (has_next) { rows_finish(gn, goff_n); load_g(pg0, 0, goff_n); store_arow(ab_n); wn_b(); }
            PL_H(b2);
            pl_lds_barrier();
            // slab 2 interval
            PL_H(c0);
            store_a(pg1, 1, 3 * PL_BK, ab);
            PL_H(c1);
            if (has_next) { load_g(pg1, PL_BK, goff_n); wn_c(); }
            PL_H(c2);
            pl_lds_barrier();
            // slab 3 interval: slab buffer 0 is free again (read during slab 2) -> next group's first slab
            PL_H(d0);
            if (has_next) {
                store_a(pg0, 0, 0, ab_n);
            }
            PL_H(d1);
            if (has_next) {
                load_g(pg0, 2 * PL_BK, goff_n);
                wn_d(gn, wvbuf + ((n + 1) & 1) * 2048, 0, 3);          // next group's weight net, first part (that buffer's last reader was the contraction of group n-1)
            }
            PL_H(d2);
            pl_lds_barrier();
#ifdef PL_STAMPS
            { PL_H(e0); h_sa += (a1 - a0) + (b1 - b0) + (c1 - c0) + (d1 - d0); h_rest += (a2 - a1) + (b2 - b1) + (c2 - c1) + (d2 - d1);
              h_bar += (b0 - a2) + (c0 - b2) + (d0 - c2) + (e0 - d2); }
#endif
#pragma unroll
            for (int it = 0; it < 4; ++it) goff[it] = goff_n[it];
        }
#ifdef PL_STAMPS
        if (blockIdx.x == 5 && (threadIdx.x & 63) == 0) {
            unsigned long long* st = reinterpret_cast<unsigned long long*>(out + (size_t)npoints * 2048) + 20 + (wave - 4) * 4;
            st[0] = h_sa; st[1] = h_rest; st[2] = h_bar; st[3] = n;
        }
#endif
        return;
    }

    // ----------------------------------------------------------------------------------------------- MFMA waves
    // v_mfma_f32_16x16x4_f32 (same 256 flop/cycle/CU as 32x32x2, bit-identical to the ascending-k fmaf chain as well): a wave owns
    // 4 points (64 pair rows) x 64 channels = 4 x 4 accumulator tiles of 16 x 16.  Lane (c = lane % 16, a = lane / 16) of a tile holds
    // rows 4 a + b (b = register 0..3) of column c; the helpers put neighbour s = 4 b + a of the point in that row, so register b of
    // relu(acc + b1) IS the A operand (channel c, k = a) of contraction step b over s = 4 b .. 4 b + 3:
    //     F'[ch][t] = sum_s X2[s][ch] * wv[s][t]   =  4 MFMAs of [16 ch x 4 s] x [4 s x 16 t] per (point, channel block)
    // -- no padding zeros (the 32x32x2 form spent half its contraction MFMAs on a block-diagonal B), s ascending.  The result
    // tiles are written out DURING the next group's product loop (one tile every other k-step, in the shadow of its MFMAs).
    const int wm = wave >> 1, wn = wave & 1;
    const int lc = lane & 15, la = lane >> 4;
    float bias[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) bias[cb] = b1[wn * 64 + cb * 16 + lc];
    __syncthreads();                                                 // #1
    __syncthreads();                                                 // #2
    int n = 0;
#ifdef PL_STAMPS
    unsigned long long c_mma = 0, c_bar = 0, c_con = 0, c_bt[4] = {0, 0, 0, 0};
#define PL_T(v) const unsigned long long v = __builtin_readcyclecounter()
#else
#define PL_T(v)
#endif
    f32x4 o[4][4];                                                   // F' tiles of the previous group, [point][channel block]
#pragma unroll
    for (int pt = 0; pt < 4; ++pt)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) o[pt][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)((unsigned)np * 8192u), 0x00020000);
    int g_prev = 0;
    // tile (pt, cb) of group gp -> F'.  Round 5: the contraction runs TRANSPOSED, O^T[t][ch] = sum_s wv[s][t] * X2[s][ch] (wv^T is the A
    // operand, relu(acc + b1) -- already in the lane that holds it, see above -- the B operand; the same products in the same s order,
    // multiplication commutes: identical bits), so lane (column = channel 16 cb + lc, a) holds t = 4 a .. 4 a + 3 of ITS channel in
    // registers 0..3 = one 16-byte store as it is.  (Rounds 2 - 4 computed O[ch][t] and transposed every tile inside lane quads: 8 DPP /
    // select pairs = ~20 VALU instructions per tile, 3200 cycles per group that the MFMAs did not hide, PL_STAMPS / PL_NOFLUSH.)
    auto flush_tile = [&](int gp, int pt, int cb, bool live) {
        const int pi = gp * 8 + 4 * wm + pt;
        const int ch = wn * 64 + cb * 16 + lc;
        // a BUFFER store: an offset beyond the resource's range is dropped by the hardware, so "nothing to flush yet" (first group)
        // and "point beyond np" (ragged last group) need no branch -- a branch here ends the scheduling region
        const unsigned off = (live && pi < np) ? ((unsigned)pi * 2048u + (unsigned)(ch * 16 + 4 * la)) * 4u : 0xFFFFFFF0u;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o[pt][cb]), out_rsrc, (int)off, 0, 0);
    };
    for (int g = g_first; g < ng; g += gstep, ++n) {
        const bool have_prev = n > 0;
        f32x4 acc[4][4];
#pragma unroll
        for (int pt = 0; pt < 4; ++pt)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) acc[pt][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NTILE; ++t) {
            const float* As = astg + (t & 1) * PW_ASTG;
            const float* Bs = wres + (t * PL_BK) * PW_LDB;
            PL_T(s0);
            // operand fragments ping-pong: the reads of k-step ks + 1 are issued before the MFMAs of k-step ks (left to itself the
            // scheduler read each B pair right before its 8 MFMAs and waited out the LDS latency 4 times per k-step)
            float af[2][4], bf[2][4];
            auto load_frags = [&](int k0, float (&a)[4], float (&b)[4]) {
#ifdef PL_X_NOA                                                      // lab (tools/micro/ps_local_lab.hip): timing experiments, wrong results
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) a[pt] = 1.0f + lc;
#else
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) a[pt] = As[(wm * 64 + pt * 16 + lc) * PW_LDAR + k0 + la];
#endif
#ifdef PL_X_NOB
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) b[cb] = 0.5f + la;
#else
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) b[cb] = Bs[(k0 + la) * PW_LDB + wn * 64 + cb * 16 + lc];
#endif
            };
            load_frags(0, af[0], bf[0]);
#pragma unroll
            for (int ks = 0; ks < PL_BK / 4; ++ks) {
                const int cur = ks & 1;
                if (ks + 1 < PL_BK / 4) load_frags(4 * (ks + 1), af[cur ^ 1], bf[cur ^ 1]);
#pragma unroll
                for (int pt = 0; pt < 4; ++pt)
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb)
                        acc[pt][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cur][pt], bf[cur][cb], acc[pt][cb], 0, 0, 0);
#ifndef PL_NOFLUSH
                if ((ks & 1) == 0) {                                 // 16 tiles over the 32 k-steps of a group
                    const int u = t * 4 + (ks >> 1);
                    flush_tile(g_prev, u >> 2, u & 3, have_prev);
                }
#endif
            }
            PL_T(s1);
            pl_lds_barrier();
            PL_T(s2);
#ifdef PL_STAMPS
            c_mma += s1 - s0; c_bar += s2 - s1; c_bt[t] += s2 - s1;
#endif
        }
        PL_T(p0s);
        const float* wv = wvbuf + (n & 1) * 2048;
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            float bw[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) bw[b] = wv[(4 * wm + pt) * 256 + (4 * b + la) * 16 + lc];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) o[pt][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int b = 0; b < 4; ++b)                               // four independent chains (cb) between dependent MFMAs
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
                    o[pt][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[b], fmaxf(acc[pt][cb][b] + bias[cb], 0.f), o[pt][cb], 0, 0, 0);
        }
        g_prev = g;
#ifdef PL_STAMPS
        { PL_T(p2s); c_con += p2s - p0s; }
#endif
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) flush_tile(g_prev, u >> 2, u & 3, n > 0);
#ifdef PL_STAMPS
    if (blockIdx.x == 5 && lane == 0) {
        unsigned long long* st = reinterpret_cast<unsigned long long*>(out + (size_t)npoints * 2048) + wave * 5;
        st[0] = c_mma; st[1] = c_bar; st[2] = 0; st[3] = c_con; st[4] = n;
        unsigned long long* sb = reinterpret_cast<unsigned long long*>(out + (size_t)npoints * 2048) + 40 + wave * 4;
        sb[0] = c_bt[0]; sb[1] = c_bt[1]; sb[2] = c_bt[2]; sb[3] = c_bt[3];
    }
#endif
}

}  // namespace dispu

using namespace dispu;

DISPU_EXPORT int dispu_ps_local(long npoints, int n_per_cloud, int k, int c, const int* idx, const float* xyz, const float* G,
                                long ldg, const float* A, const float* W1, const float* b1, const float* Ww, const float* bw,
                                const float* scale, const float* shift, float* out, void* stream) {
    if (npoints < 0 || n_per_cloud <= 0 || k != 16 || c != 128 || ldg < 128 || (ldg & 3) || npoints * ldg > 0x7fffffffL) return (int)hipErrorInvalidValue;
    if ((((uintptr_t)G) | ((uintptr_t)A) | ((uintptr_t)W1) | ((uintptr_t)out)) & 15) return (int)hipErrorInvalidValue;
    if (npoints == 0) return 0;
    const int mode = 1;                 // the wave-specialised persistent kernel (mode 0 = round 1's single-role kernel, kept as the documented baseline)
    static DevOnce attr;      
    if (attr.needed()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ps_local_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)PL_LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(ps_local_ws_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)PW_LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(ps_local_ws_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)PW_LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr.done();
    }
    if (mode == 0) {
        hipLaunchKernelGGL(ps_local_kernel, dim3((unsigned)((npoints + 7) / 8)), dim3(PL_NT), PL_LDS_BYTES, (hipStream_t)stream, npoints,
                           n_per_cloud, idx, xyz, G, ldg, A, W1, b1, Ww, bw, scale, shift, out);
    } else {
        // the F' tiles leave through a buffer resource whose byte range is a 32-bit field: launches of at most 2^18 points
        // (2 GB of F'), cut at cloud boundaries (neighbour ids are cloud-local, the kernel derives a point's cloud from its index)
        const long max_pts = 262144;
        if (n_per_cloud > max_pts) return (int)hipErrorInvalidValue;
        const long per = npoints <= max_pts ? npoints : (max_pts / n_per_cloud) * (long)n_per_cloud;
        for (long p0 = 0; p0 < npoints; p0 += per) {
            const long np = (npoints - p0 < per) ? npoints - p0 : per;
            const long ngroups = (np + 7) / 8;
            const unsigned grid = (unsigned)(ngroups < 256 ? ngroups : 256);      // one persistent workgroup per CU
            // whole 8-point groups inside one cloud, 32-bit byte offsets for the helpers' buffer loads
            const bool uni = (n_per_cloud % 8) == 0 && (np % 8) == 0 && np * ldg * 4 < 0x7fffffffL;
            if (uni)
                hipLaunchKernelGGL(ps_local_ws_kernel<true>, dim3(grid), dim3(512), PW_LDS_BYTES, (hipStream_t)stream, np, n_per_cloud, idx + p0 * 16,
                                   xyz + p0 * 3, G + p0 * ldg, (int)ldg, A + p0 * PL_K, W1, b1, Ww, bw, scale, shift, out + p0 * 2048);
            else
                hipLaunchKernelGGL(ps_local_ws_kernel<false>, dim3(grid), dim3(512), PW_LDS_BYTES, (hipStream_t)stream, np, n_per_cloud, idx + p0 * 16,
                                   xyz + p0 * 3, G + p0 * ldg, (int)ldg, A + p0 * PL_K, W1, b1, Ww, bw, scale, shift, out + p0 * 2048);
        }
    }
    return (int)hipGetLastError();
}
