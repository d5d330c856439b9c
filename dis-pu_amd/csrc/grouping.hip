// Ball query, group_point and its gradient for gfx950.
// Replaces queryBallPointLauncher / groupPointLauncher / groupPointGradLauncher
// (tf_ops/grouping/tf_grouping_g.cu:125-141) behind dispu_query_ball / dispu_group_point /
// dispu_group_point_grad.
//
// Ball query: the reference runs ONE 256-thread block per cloud with a thread per query that
// re-reads the AoS dataset from global memory.  Here a workgroup owns 256 queries of one
// cloud; the dataset streams through LDS in 1024-point tiles (float4-padded so a candidate is
// one broadcast ds_read_b128), every lane scans in index order and the wave leaves the tile
// loop as soon as all of its lanes are full (cnt == nsample).
#include "common.h"

namespace dispu {

constexpr int QB_BS = 256;
constexpr int QB_TILE = 1024;

template <bool FMA>
__global__ __launch_bounds__(QB_BS) void query_ball_kernel(int n, int m, const float* __restrict__ radius, int nsample,
                                                            const float* __restrict__ xyz1,
                                                            const float* __restrict__ xyz2, int* __restrict__ idx,
                                                            int* __restrict__ pts_cnt) {
    __shared__ float4 tile[QB_TILE];
    const int cloud = blockIdx.y;
    const float* __restrict__ p1 = xyz1 + (size_t)cloud * n * 3;
    const float* __restrict__ p2 = xyz2 + (size_t)cloud * m * 3;
    const int j = blockIdx.x * QB_BS + threadIdx.x;
    const bool active = j < m;
    const float r = radius[0];  // the reference reads element 0 only (tf_grouping_g.cu:25)
    float x2 = 0.f, y2 = 0.f, z2 = 0.f;
    if (active) { x2 = p2[j * 3 + 0]; y2 = p2[j * 3 + 1]; z2 = p2[j * 3 + 2]; }
    int* __restrict__ row = idx + ((size_t)cloud * m + (active ? j : 0)) * nsample;
    int cnt = active ? 0 : nsample;
    int first = 0;
    for (int k0 = 0; k0 < n; k0 += QB_TILE) {
        const int len = min(QB_TILE, n - k0);
        __syncthreads();
        for (int t = threadIdx.x; t < len; t += QB_BS)
            tile[t] = make_float4(p1[(k0 + t) * 3 + 0], p1[(k0 + t) * 3 + 1], p1[(k0 + t) * 3 + 2], 0.f);
        __syncthreads();
        if (__all(cnt >= nsample)) continue;  // wave-uniform: this wave is done, keep feeding the barriers
        for (int t = 0; t < len; ++t) {
            const float4 q = tile[t];
            const float d2 = sqdist3<FMA>(x2 - q.x, y2 - q.y, z2 - q.z);
            const float d = fmaxf(sqrtf(d2), 1e-20f);
            if (cnt < nsample && d < r) {
                if (cnt == 0) first = k0 + t;
                row[cnt] = k0 + t;
                ++cnt;
            }
        }
    }
    if (active) {
        // first hit replicated into the unused tail; a row without any hit stays untouched
        if (cnt > 0)
            for (int l = cnt; l < nsample; ++l) row[l] = first;
        pts_cnt[(size_t)cloud * m + j] = cnt;
    }
}

// n <= 1024: a WAVE owns a query.  Lane l keeps candidates 64 r + l (r < R) in registers for all queries of the wave;
// per 64-candidate block the hits are a ballot mask, a hit's output slot is cnt + (hits in lower lanes), so the first
// nsample hits are written in index order exactly as the serial scan does, and the wave stops at the block that fills
// the row.  The lane-per-query kernel above runs b * m / 64 waves that each scan n candidates serially (141 us for the
// repulsion loss's (8, 1024, 1024, 20)); here all 64 lanes of b * m waves work.
template <int R, bool FMA>
__global__ __launch_bounds__(256) void query_ball_wave_kernel(int n, int m, int qpb, const float* __restrict__ radius, int nsample,
                                                               const float* __restrict__ xyz1, const float* __restrict__ xyz2,
                                                               int* __restrict__ idx, int* __restrict__ pts_cnt) {
    const int cloud = blockIdx.y, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const float* __restrict__ p1 = xyz1 + (size_t)cloud * n * 3;
    const float* __restrict__ p2 = xyz2 + (size_t)cloud * m * 3;
    const float rad = radius[0];
    float cx[R], cy[R], cz[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int p = min(64 * r + lane, n - 1);
        cx[r] = p1[p * 3 + 0]; cy[r] = p1[p * 3 + 1]; cz[r] = p1[p * 3 + 2];
    }
    const int q0 = blockIdx.x * qpb, q1 = min(m, q0 + qpb);
    for (int qv = q0 + wave; qv < q1; qv += 4) {
        const int j = __builtin_amdgcn_readfirstlane(qv);
        const float x2 = p2[j * 3 + 0], y2 = p2[j * 3 + 1], z2 = p2[j * 3 + 2];
        int* __restrict__ row = idx + ((size_t)cloud * m + j) * nsample;
        int cnt = 0, first = 0;                                    // wave-uniform
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (cnt < nsample && 64 * r < n) {
                const float d2 = sqdist3<FMA>(x2 - cx[r], y2 - cy[r], z2 - cz[r]);
                const float d = fmaxf(sqrtf(d2), 1e-20f);
                const bool hit = (d < rad) && (64 * r + lane < n);
                const unsigned long long mk = __ballot(hit);
                if (mk) {
                    if (cnt == 0) first = 64 * r + (int)__builtin_ctzll(mk);
                    const int pos = cnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u));
                    if (hit && pos < nsample) row[pos] = 64 * r + lane;
                    cnt = min(nsample, cnt + (int)__popcll(mk));
                }
            }
        }
        // first hit replicated into the unused tail; a row without any hit stays untouched
        if (cnt > 0)
            for (int l = cnt + lane; l < nsample; l += 64) row[l] = first;
        if (lane == 0) pts_cnt[(size_t)cloud * m + j] = cnt;
    }
}

template <int R>
static void launch_query_ball_wave(int b, int n, int m, const float* radius, int nsample, const float* xyz1, const float* xyz2,
                                   int* idx, int* pts_cnt, int arith, hipStream_t st) {
    const int qpb = ((long)b * m >= 32768) ? 32 : 16;
    dim3 grid((m + qpb - 1) / qpb, b);
    if (arith & DISPU_ARITH_CONTRACT)
        hipLaunchKernelGGL((query_ball_wave_kernel<R, true>), grid, dim3(256), 0, st, n, m, qpb, radius, nsample, xyz1, xyz2, idx, pts_cnt);
    else
        hipLaunchKernelGGL((query_ball_wave_kernel<R, false>), grid, dim3(256), 0, st, n, m, qpb, radius, nsample, xyz1, xyz2, idx, pts_cnt);
}

// Flat gather: element e of out[b,m,ns,c] <- points[cloud, idx[row], e % c].  VEC = floats per lane.
// Gather rows: out[row, :] = points[cloud(row), idx[row], :].  TX lanes (a power of two) share one output row and
// stride over its c/VEC vectors; a workgroup covers 256/TX consecutive rows, so every wave writes whole contiguous
// rows (coalesced) and the index / cloud arithmetic is done once per row in 32-bit (no per-element 64-bit division).
template <int VEC>
__global__ __launch_bounds__(256) void group_point_kernel(int n, int c, int rows_per_cloud, long rows, int tx_log2,
                                                          const float* __restrict__ points, const int* __restrict__ idx,
                                                          float* __restrict__ out) {
    const int TX = 1 << tx_log2;
    const int tx = threadIdx.x & (TX - 1);
    const int rpb = 256 >> tx_log2;                         // rows per workgroup pass
    const int cv = c / VEC;
    for (long row = (long)blockIdx.x * rpb + (threadIdx.x >> tx_log2); row < rows; row += (long)gridDim.x * rpb) {
        const long cloud = row / rows_per_cloud;
        const float* __restrict__ src = points + (cloud * n + idx[row]) * c;
        float* __restrict__ dst = out + row * c;
        for (int l = tx; l < cv; l += TX) {
            if constexpr (VEC == 4) reinterpret_cast<float4*>(dst)[l] = reinterpret_cast<const float4*>(src)[l];
            else dst[l] = src[l];
        }
    }
}

__global__ void group_point_grad_kernel(int n, int c, int rows_per_cloud, size_t total,
                                        const float* __restrict__ grad_out, const int* __restrict__ idx,
                                        float* __restrict__ grad_points) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t row = e / c;
        const int l = (int)(e - row * c);
        const size_t cloud = row / rows_per_cloud;
        unsafeAtomicAdd(grad_points + (cloud * n + idx[row]) * c + l, grad_out[e]);
    }
}

}  // namespace dispu

using namespace dispu;

static inline int grid_for(size_t total, int bs) {
    size_t g = (total + bs - 1) / bs;
    if (g > 16384) g = 16384;
    if (g < 1) g = 1;
    return (int)g;
}

DISPU_EXPORT int dispu_query_ball(int b, int n, int m, const float* radius, int nsample, const float* xyz1,
                                  const float* xyz2, int* idx, int* pts_cnt, int arith, void* stream) {
    if (b < 0 || n <= 0 || m < 0 || nsample <= 0 || !radius) return (int)hipErrorInvalidValue;
    if (b == 0 || m == 0) return 0;
    if (n <= 1024) {
        hipStream_t st = (hipStream_t)stream;
        if (n <= 64) launch_query_ball_wave<1>(b, n, m, radius, nsample, xyz1, xyz2, idx, pts_cnt, arith, st);
        else if (n <= 128) launch_query_ball_wave<2>(b, n, m, radius, nsample, xyz1, xyz2, idx, pts_cnt, arith, st);
        else if (n <= 256) launch_query_ball_wave<4>(b, n, m, radius, nsample, xyz1, xyz2, idx, pts_cnt, arith, st);
        else if (n <= 512) launch_query_ball_wave<8>(b, n, m, radius, nsample, xyz1, xyz2, idx, pts_cnt, arith, st);
        else launch_query_ball_wave<16>(b, n, m, radius, nsample, xyz1, xyz2, idx, pts_cnt, arith, st);
        return (int)hipGetLastError();
    }
    dim3 grid((m + QB_BS - 1) / QB_BS, b);
    if ((arith & DISPU_ARITH_CONTRACT))
        hipLaunchKernelGGL((query_ball_kernel<true>), grid, dim3(QB_BS), 0, (hipStream_t)stream, n, m, radius, nsample,
                           xyz1, xyz2, idx, pts_cnt);
    else
        hipLaunchKernelGGL((query_ball_kernel<false>), grid, dim3(QB_BS), 0, (hipStream_t)stream, n, m, radius, nsample,
                           xyz1, xyz2, idx, pts_cnt);
    return (int)hipGetLastError();
}

DISPU_EXPORT int dispu_group_point(int b, int n, int c, int m, int nsample, const float* points, const int* idx,
                                   float* out, void* stream) {
    if (b < 0 || n <= 0 || c <= 0 || m < 0 || nsample < 0) return (int)hipErrorInvalidValue;
    const size_t rows = (size_t)b * m * nsample;
    if (rows == 0) return 0;
    const bool vec4 = (c % 4 == 0) && (((uintptr_t)points | (uintptr_t)out) % 16 == 0);
    const int cv = vec4 ? c / 4 : c;
    int tx_log2 = 0;
    while ((1 << tx_log2) < cv && tx_log2 < 6) ++tx_log2;           // lanes per row: next power of two >= cv, at most 64
    const size_t rpb = (size_t)256 >> tx_log2;
    size_t g = (rows + rpb - 1) / rpb;
    if (g > 65536) g = 65536;
    if (vec4)
        hipLaunchKernelGGL((group_point_kernel<4>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, n, c, m * nsample,
                           (long)rows, tx_log2, points, idx, out);
    else
        hipLaunchKernelGGL((group_point_kernel<1>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, n, c, m * nsample,
                           (long)rows, tx_log2, points, idx, out);
    return (int)hipGetLastError();
}

DISPU_EXPORT int dispu_group_point_grad(int b, int n, int c, int m, int nsample, const float* grad_out, const int* idx,
                                        float* grad_points, void* stream) {
    if (b < 0 || n <= 0 || c <= 0 || m < 0 || nsample < 0) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    if (b) DISPU_TRY(hipMemsetAsync(grad_points, 0, sizeof(float) * (size_t)b * n * c, s));
    const size_t total = (size_t)b * m * nsample * c;
    if (total == 0) return 0;
    hipLaunchKernelGGL(group_point_grad_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, n, c, m * nsample, total,
                       grad_out, idx, grad_points);
    return (int)hipGetLastError();
}
