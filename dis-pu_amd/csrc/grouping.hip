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

#include <cstdlib>

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
// hit <=> max(sqrtf(d2), 1e-20f) < radius (tf_grouping_g.cu:20-27), decided WITHOUT the correctly rounded square root for all but the
// candidates within 4e-6 (relative) of radius^2: sqrt is monotone and correctly rounded, radius^2 is rounded once, so d2 below
// r2 (1 - 2^-18) is a hit and d2 above r2 (1 + 2^-18) is a miss whatever the roundings; a wave computes the exact form only when one
// of its lanes falls into that band (or is unordered: NaN coordinates take the reference's path).  The square root with its fix-up
// was ~2/3 of the instructions of a candidate.
struct QbBand {
    float rad, lo, hi;
    bool always_exact;
};
__device__ __forceinline__ QbBand qb_band(float rad) {
    const float r2 = rad * rad;
    return QbBand{rad, r2 * (1.0f - 3.8146973e-6f), r2 * (1.0f + 3.8146973e-6f), !(rad > 1e-19f) || !(r2 > 1e-30f) || !(r2 < 1e30f)};
}
// The hit masks of R 64-candidate blocks at once: R independent compares and ballots, ONE band test for all of them (block by block
// a query was a chain of scalar branches, each waiting for a vector compare).
template <int R>
__device__ __forceinline__ void qb_masks(const QbBand& b, const float (&d2)[R], int base, int lane, int n, unsigned long long (&mk)[R]) {
    // every predicate is a vector compare written straight to a scalar mask; the rest is scalar logic (as per-lane bools the masks cost
    // ~6 vector instructions per block on top of the distance: 450 vector instructions per query, which is what bounded the kernel)
    unsigned long long band = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        mk[r] = __ballot(d2[r] < b.lo);
        band |= ~mk[r] & __ballot(!(d2[r] > b.hi));                      // includes unordered (NaN) distances
    }
    if (b.always_exact || band) {                                        // wave-uniform, rare
#pragma unroll
        for (int r = 0; r < R; ++r) mk[r] = __ballot(fmaxf(sqrtf(d2[r]), 1e-20f) < b.rad);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {                                        // candidates past n (clamped loads): masked out, scalar
        const int left = n - (base + 64 * r);
        mk[r] &= left >= 64 ? ~0ull : left <= 0 ? 0ull : ((1ull << left) - 1ull);
    }
}

template <int R, bool FMA>
__global__ __launch_bounds__(256) void query_ball_wave_kernel(int n, int m, int qpb, const float* __restrict__ radius, int nsample,
                                                               const float* __restrict__ xyz1, const float* __restrict__ xyz2,
                                                               int* __restrict__ idx, int* __restrict__ pts_cnt) {
    const int cloud = blockIdx.y, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const float* __restrict__ p1 = xyz1 + (size_t)cloud * n * 3;
    const float* __restrict__ p2 = xyz2 + (size_t)cloud * m * 3;
    const float rad = radius[0];
    const QbBand band = qb_band(rad);
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
        float d2[R];
#pragma unroll
        for (int r = 0; r < R; ++r) d2[r] = sqdist3<FMA>(x2 - cx[r], y2 - cy[r], z2 - cz[r]);
        unsigned long long mk[R];
        qb_masks<R>(band, d2, 0, lane, n, mk);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (mk[r] && cnt < nsample) {                          // scalar values: no vector compare to wait for
                if (cnt == 0) first = 64 * r + (int)__builtin_ctzll(mk[r]);
                const int pos = cnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mk[r] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk[r], 0u));
                if (((mk[r] >> lane) & 1ull) && pos < nsample) row[pos] = 64 * r + lane;
                cnt = min(nsample, cnt + (int)__popcll(mk[r]));
            }
        }
        // first hit replicated into the unused tail; a row without any hit stays untouched
        if (cnt > 0)
            for (int l = cnt + lane; l < nsample; l += 64) row[l] = first;
        if (lane == 0) pts_cnt[(size_t)cloud * m + j] = cnt;
    }
}

// n > 1024: the same wave-per-query scan over CHUNKS of 1024 candidates.  The chunk's candidates sit in registers for all
// queries of the workgroup; a query's progress (hits so far, first hit) waits in LDS between chunks, and a wave skips a
// query whose row is already full - the scan is in index order, so the rows fill exactly as in the serial loop.  The
// lane-per-query kernel ran (8, 4096, 4096, 20) in 564 us (5 % of the VALU rate: b * m / 64 = 512 waves, serial 4096-long
// scans with divergent stores).
template <bool FMA>
__global__ __launch_bounds__(256) void query_ball_wave_chunked_kernel(int n, int m, int qpb, const float* __restrict__ radius, int nsample,
                                                                       const float* __restrict__ xyz1, const float* __restrict__ xyz2,
                                                                       int* __restrict__ idx, int* __restrict__ pts_cnt) {
    constexpr int R = 16;
    __shared__ int s_cnt[64], s_first[64];                       // qpb <= 64 queries per workgroup
    const int cloud = blockIdx.y, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const float* __restrict__ p1 = xyz1 + (size_t)cloud * n * 3;
    const float* __restrict__ p2 = xyz2 + (size_t)cloud * m * 3;
    const float rad = radius[0];
    const QbBand band = qb_band(rad);
    const int q0 = blockIdx.x * qpb, q1 = min(m, q0 + qpb);
    if (threadIdx.x < 64) { s_cnt[threadIdx.x] = 0; s_first[threadIdx.x] = 0; }
    __syncthreads();
    for (int c0 = 0; c0 < n; c0 += 64 * R) {
        float cx[R], cy[R], cz[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int p = min(c0 + 64 * r + lane, n - 1);
            cx[r] = p1[p * 3 + 0]; cy[r] = p1[p * 3 + 1]; cz[r] = p1[p * 3 + 2];
        }
        for (int qv = q0 + wave; qv < q1; qv += 4) {            // a query always belongs to the same wave: no barrier needed
            const int j = __builtin_amdgcn_readfirstlane(qv);
            int cnt = s_cnt[j - q0], first = s_first[j - q0];   // wave-uniform
            if (cnt >= nsample) continue;
            const float x2 = p2[j * 3 + 0], y2 = p2[j * 3 + 1], z2 = p2[j * 3 + 2];
            int* __restrict__ row = idx + ((size_t)cloud * m + j) * nsample;
            float d2[R];
#pragma unroll
            for (int r = 0; r < R; ++r) d2[r] = sqdist3<FMA>(x2 - cx[r], y2 - cy[r], z2 - cz[r]);
            unsigned long long mk[R];
            qb_masks<R>(band, d2, c0, lane, n, mk);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (mk[r] && cnt < nsample) {
                    if (cnt == 0) first = c0 + 64 * r + (int)__builtin_ctzll(mk[r]);
                    const int pos = cnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mk[r] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk[r], 0u));
                    if (((mk[r] >> lane) & 1ull) && pos < nsample) row[pos] = c0 + 64 * r + lane;
                    cnt = min(nsample, cnt + (int)__popcll(mk[r]));
                }
            }
            if (lane == 0) { s_cnt[j - q0] = cnt; s_first[j - q0] = first; }
        }
    }
    for (int qv = q0 + wave; qv < q1; qv += 4) {
        const int j = __builtin_amdgcn_readfirstlane(qv);
        const int cnt = s_cnt[j - q0], first = s_first[j - q0];
        int* __restrict__ row = idx + ((size_t)cloud * m + j) * nsample;
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

// Gather rows: out[row, :] = points[cloud(row), idx[row], :]  -- a pure copy, HBM-bound (the one family where
// >= 60 % of the HBM peak is the right target).  TX lanes (a power of two) form a row SLOT and stride over the row's
// c/VEC vectors; a workgroup has 256/TX slots and every slot copies R rows per launch: the R index loads are issued
// together, then R x (c / VEC / TX) independent 16-byte loads per lane are in flight before the first store (round 1 had
// ONE load in flight per lane behind a dependent index load: 46 % of HBM at c = 256).  At a fixed r the slots cover
// consecutive rows, so every store instruction of a wave writes one contiguous segment; stores are non-temporal (the
// output is written once and never re-read here) so they do not evict the gathered rows from L2, and workgroup ids are
// re-mapped so each XCD copies one contiguous range of rows (= the same few clouds share one L2).
template <int VEC, int R>
__global__ __launch_bounds__(256) void group_rows_kernel(int n, int c, unsigned rows_per_cloud, unsigned rows, int tx_log2,
                                                         const float* __restrict__ points, const int* __restrict__ idx,
                                                         float* __restrict__ out) {
    const unsigned TX = 1u << tx_log2, tx = threadIdx.x & (TX - 1), slot = threadIdx.x >> tx_log2, slots = 256u >> tx_log2;
    const unsigned base = xcd_block(blockIdx.x, gridDim.x) * (slots * R) + slot;
    const int cv = c / VEC;
    const float* src[R];
    bool ok[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const unsigned row = base + r * slots;
        ok[r] = row < rows;
        const unsigned rr = ok[r] ? row : 0u;
        src[r] = points + ((size_t)(rr / rows_per_cloud) * n + idx[rr]) * c;
    }
    for (int l = tx; l < cv; l += TX) {
        if constexpr (VEC == 4) {
            float4 v[R];
#pragma unroll
            for (int r = 0; r < R; ++r) v[r] = reinterpret_cast<const float4*>(src[r])[l];
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (ok[r]) store_nt4(out + (size_t)(base + r * slots) * c + 4 * l, v[r]);
        } else {
            float v[R];
#pragma unroll
            for (int r = 0; r < R; ++r) v[r] = src[r][l];
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (ok[r]) __builtin_nontemporal_store(v[r], out + (size_t)(base + r * slots) * c + l);
        }
    }
}

// c == 3 (xyz rows: gather_point, and group_point on coordinates): a lane owns R rows -- index load, one 12-byte load
// each -- and the 256 x 3 floats of a pass leave through LDS as 192 coalesced float4 stores instead of 768 scattered
// 4-byte ones.
template <int R>
__global__ __launch_bounds__(256) void gather_xyz_kernel(int n, unsigned rows_per_cloud, unsigned rows, const float* __restrict__ inp,
                                                         const int* __restrict__ idx, float* __restrict__ out) {
    __shared__ float stage[R][768];
    const unsigned base = xcd_block(blockIdx.x, gridDim.x) * (256u * R);
    float vx[R], vy[R], vz[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const unsigned row = base + r * 256 + threadIdx.x;
        const unsigned rr = row < rows ? row : 0u;
        const float* s = inp + ((size_t)(rr / rows_per_cloud) * n + idx[rr]) * 3;
        vx[r] = s[0]; vy[r] = s[1]; vz[r] = s[2];
    }
    const bool full = base + 256u * R <= rows && ((uintptr_t)out % 16 == 0);
    if (full) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            stage[r][threadIdx.x * 3 + 0] = vx[r]; stage[r][threadIdx.x * 3 + 1] = vy[r]; stage[r][threadIdx.x * 3 + 2] = vz[r];
        }
        __syncthreads();
        if (threadIdx.x < 192) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float4 v = reinterpret_cast<const float4*>(stage[r])[threadIdx.x];
                store_nt4(out + (size_t)(base + r * 256) * 3 + 4 * threadIdx.x, v);
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const unsigned row = base + r * 256 + threadIdx.x;
            if (row < rows) { float* d = out + (size_t)row * 3; d[0] = vx[r]; d[1] = vy[r]; d[2] = vz[r]; }
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
    {                                               // wave-per-query over 1024-candidate chunks
        const int qpb = ((long)b * m >= 32768) ? 32 : 16;
        dim3 g((m + qpb - 1) / qpb, b);
        if ((arith & DISPU_ARITH_CONTRACT))
            hipLaunchKernelGGL((query_ball_wave_chunked_kernel<true>), g, dim3(256), 0, (hipStream_t)stream, n, m, qpb, radius, nsample, xyz1,
                               xyz2, idx, pts_cnt);
        else
            hipLaunchKernelGGL((query_ball_wave_chunked_kernel<false>), g, dim3(256), 0, (hipStream_t)stream, n, m, qpb, radius, nsample, xyz1,
                               xyz2, idx, pts_cnt);
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

namespace dispu {
// rows x c gather shared by dispu_group_point and dispu_gather_point (csrc/sampling.hip)
int launch_gather_rows(size_t rows, int n, int c, size_t rows_per_cloud, const float* points, const int* idx, float* out,
                       hipStream_t st) {
    if (rows == 0) return 0;
    if (rows >= 0x7fffffffull || rows_per_cloud >= 0x7fffffffull) return (int)hipErrorInvalidValue;   // 32-bit row arithmetic
    const unsigned rws = (unsigned)rows, rpc = (unsigned)rows_per_cloud;
    if (c == 3) {
        constexpr int R = 4;
        const unsigned g = (rws + 256 * R - 1) / (256 * R);
        hipLaunchKernelGGL((gather_xyz_kernel<R>), dim3(g), dim3(256), 0, st, n, rpc, rws, points, idx, out);
        return (int)hipGetLastError();
    }
    const bool vec4 = (c % 4 == 0) && (((uintptr_t)points | (uintptr_t)out) % 16 == 0);
    const int cv = vec4 ? c / 4 : c;
    int tx_log2 = 0;
    while ((1 << tx_log2) < cv && tx_log2 < 6) ++tx_log2;           // lanes per row: next power of two >= cv, at most 64
    const unsigned slots = 256u >> tx_log2;
    constexpr int R = 8;
    const unsigned g = (rws + slots * R - 1) / (slots * R);
    if (vec4)
        hipLaunchKernelGGL((group_rows_kernel<4, R>), dim3(g), dim3(256), 0, st, n, c, rpc, rws, tx_log2, points, idx, out);
    else
        hipLaunchKernelGGL((group_rows_kernel<1, R>), dim3(g), dim3(256), 0, st, n, c, rpc, rws, tx_log2, points, idx, out);
    return (int)hipGetLastError();
}
}  // namespace dispu

DISPU_EXPORT int dispu_group_point(int b, int n, int c, int m, int nsample, const float* points, const int* idx,
                                   float* out, void* stream) {
    if (b < 0 || n <= 0 || c <= 0 || m < 0 || nsample < 0) return (int)hipErrorInvalidValue;
    return launch_gather_rows((size_t)b * m * nsample, n, c, (size_t)m * nsample, points, idx, out, (hipStream_t)stream);
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

// ---- selection_sort (tf_grouping_g.cu:83-123; optional in the reference: select_top_k is never called by the shipped graph) ----
// out / outi [b, m, n]: a copy of dist with its first k entries put in place by k rounds of "swap position s with the FIRST
// minimum of positions s..n-1" (strict <, so the lowest position wins ties and position s itself wins against equal values).
// The reference gives a row to one thread; here a row belongs to one wave: the lanes scan s..n-1 with a stride of 64, keep the
// first minimum each, and a DPP arg-min over (ordered value, position) keys -- the position is the tie-break, which is the
// serial scan's first-minimum rule -- names the winner; lane 0 swaps.  -0.0f is canonicalised to +0.0f in the KEY only (the
// float compare treats them as equal), the values themselves are moved unchanged.
namespace dispu {
__global__ __launch_bounds__(kWave) void selection_sort_kernel(int n, int k, const float* __restrict__ dist, int* __restrict__ outi,
                                                                float* __restrict__ out) {
    const size_t row = blockIdx.x;
    const int lane = threadIdx.x;
    const float* __restrict__ d = dist + row * n;
    float* o = out + row * n;
    int* oi = outi + row * n;
    for (int s = lane; s < n; s += kWave) { o[s] = d[s]; oi[s] = s; }
    __syncthreads();                                              // single-wave workgroup: orders the wave's own stores and loads
    const int rounds = k < n ? k : n;
    for (int s = 0; s < rounds; ++s) {
        uint64_t best = ~0ull;
        for (int t = s + lane; t < n; t += kWave) {
            const uint64_t key = ((uint64_t)f32_to_ordered(o[t] + 0.0f) << 32) | (uint32_t)t;
            best = u64_min(best, key);
        }
        const uint64_t win = wave_min_u64(best);
        const int mn = (int)(uint32_t)win;
        if (lane == 0 && mn != s) {
            const float tf = o[mn]; o[mn] = o[s]; o[s] = tf;
            const int ti = oi[mn]; oi[mn] = oi[s]; oi[s] = ti;
        }
        __syncthreads();
    }
}
}  // namespace dispu

// selectionSortLauncher(b,n,m,k,dist,outi,out)   tf_grouping.cpp:112,141.
DISPU_EXPORT int dispu_selection_sort(int b, int n, int m, int k, const float* dist, int* outi, float* out, void* stream) {
    if (b < 0 || n <= 0 || m < 0 || k <= 0) return (int)hipErrorInvalidValue;
    const size_t rows = (size_t)b * m;
    if (rows == 0) return 0;
    if (!dist || !outi || !out || rows > 0x7fffffffull) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(dispu::selection_sort_kernel, dim3((unsigned)rows), dim3(dispu::kWave), 0, (hipStream_t)stream, n, k, dist, outi, out);
    return (int)hipGetLastError();
}
