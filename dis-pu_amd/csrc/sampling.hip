// Farthest point sampling, gather_point and its gradient for gfx950.
// Replaces the launchers of tf_ops/sampling/tf_sampling_g.cu:194-211 (reference) behind
// dispu_fps / dispu_gather_point / dispu_gather_point_grad (include/dispu_hip.h).
//
// FPS design (MI355X): one workgroup per cloud, the cloud's coordinates AND the running
// min-distance array live in VGPRs for the whole kernel (the reference round-trips `temp`
// through global memory every round and keeps 3072 points in shared memory).  A round is
//   P x (3 sub, mul, 2 fma, min, 64-bit key compare)  ->  DPP wave arg-max  ->  one LDS slot per
//   wave, ONE barrier (double-buffered slots)  ->  scalar load of the winner's coordinates.
// The reference's winner rule (max d, ties -> lowest k mod 512, then lowest k; see
// tf_sampling_g.cu:146,158) is reproduced arithmetically with a 64-bit key
//   key = float_bits(d) << 32 | (0xFFFFFFFF - ((k & 511) << 22 | k >> 9))
// whose unsigned maximum is exactly that winner, independent of this kernel's thread layout.
#include "common.h"

namespace dispu {

__device__ __forceinline__ uint32_t fps_tiekey(int k) {
    return 0xFFFFFFFFu - ((((uint32_t)k & 511u) << 22) | ((uint32_t)k >> 9));
}
__device__ __forceinline__ int fps_key_to_index(uint64_t key) {
    const uint32_t t = 0xFFFFFFFFu - (uint32_t)key;
    return (int)(((t & 0x3FFFFFu) << 9) | (t >> 22));
}

// Wave arg-max of (distance, tie key) on 32-bit DPP steps (round 4).  The 64-bit key needs two DPP moves, a 64-bit compare and two
// selects per step (wave_max_u64: ~45 dependent-ish instructions per round); running distances are >= 0 (or -1 = "no point"), so
// their bit patterns + 1 order like unsigned integers and fold with ONE v_max_u32_dpp per step.  Two such reductions - the
// largest distance, then the largest tie key among the lanes that hold it (tie keys are unique, 0 elsewhere) - give the same
// (distance, key) pair, wave-uniform, in 12 single-instruction steps and two v_readlane.
__device__ __forceinline__ uint32_t fps_dkey(float d) { return d >= 0.f ? __float_as_uint(d) + 1u : 0u; }
template <int CTRL, int RM = 0xF>
__device__ __forceinline__ uint32_t fps_umax_step(uint32_t v) {
    const uint32_t o = dpp_u32<CTRL, RM>(0u, v);
    return v > o ? v : o;
}
__device__ __forceinline__ uint32_t fps_wave_max_u32(uint32_t v) {
    v = fps_umax_step<DPP_ROW_SHR1>(v);
    v = fps_umax_step<DPP_ROW_SHR2>(v);
    v = fps_umax_step<DPP_ROW_SHR4>(v);
    v = fps_umax_step<DPP_ROW_SHR8>(v);
    v = fps_umax_step<DPP_ROW_BCAST15, 0xA>(v);
    v = fps_umax_step<DPP_ROW_BCAST31, 0xC>(v);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// -> the wave's best as the 64-bit key of fps_tiekey's scheme (distance bits << 32 | tie key), 0 when the wave holds no point
__device__ __forceinline__ uint64_t fps_wave_best(float bd, int k) {
    const uint32_t dk = fps_dkey(bd);
    const uint32_t dmax = fps_wave_max_u32(dk);
    const uint32_t tk = (dk == dmax) ? fps_tiekey(k) : 0u;
    const uint32_t tmax = fps_wave_max_u32(tk);
    return dmax ? (((uint64_t)(dmax - 1u) << 32) | tmax) : 0ull;
}

// The winner among the W <= 16 per-wave slots: lane l < W reads slot l (ONE ds_read_b64 per lane instead of W broadcast reads and
// W 64-bit compare/select triples in every lane), then the same two-stage 32-bit reduction over DPP row 0.
template <int W>
__device__ __forceinline__ uint64_t fps_slots_best(const uint64_t* slots, int lane) {
    static_assert(W <= 16, "one DPP row");
    const uint64_t v = (lane < W) ? slots[lane] : 0ull;
    auto row0_max = [](uint32_t x) -> uint32_t {
        x = fps_umax_step<DPP_ROW_SHR1>(x);
        if (W > 2) x = fps_umax_step<DPP_ROW_SHR2>(x);
        if (W > 4) x = fps_umax_step<DPP_ROW_SHR4>(x);
        if (W > 8) x = fps_umax_step<DPP_ROW_SHR8>(x);
        return (uint32_t)__builtin_amdgcn_readlane((int)x, W - 1 < 15 ? (W <= 2 ? 1 : W <= 4 ? 3 : W <= 8 ? 7 : 15) : 15);
    };
    // distance bits are compared as (bits + 1) with 0 = empty slot, exactly as inside the waves
    const uint32_t hi = (uint32_t)(v >> 32), lo = (uint32_t)v;
    const uint32_t dk = v ? hi + 1u : 0u;
    const uint32_t dmax = row0_max(dk);
    const uint32_t tmax = row0_max(dk == dmax ? lo : 0u);
    return dmax ? (((uint64_t)(dmax - 1u) << 32) | tmax) : 0ull;
}

// Visit order of a thread's P points such that the reference tie priority (k mod 512, then k) is
// non-decreasing along the visit: then a strict '>' update keeps exactly the reference's winner
// among equal distances inside the thread, and the 64-bit key is only built once per round.
//   BS = 64  (P <= 8):  k = tid + 64 i < 512          -> natural order
//   BS = 256 (P <= 8):  k mod 512 alternates tid / tid+256 -> even i first, then odd i
//   BS = 1024:          k mod 512 is the same for all i    -> natural order
template <int BS, int P>
__device__ __forceinline__ constexpr int fps_visit(int s) {
    if (BS == 256 && P > 2) return (s < P / 2) ? 2 * s : 2 * (s - P / 2) + 1;
    return s;
}

// BS threads, P points per thread (thread t owns k = t + i*BS), FMA = arithmetic flavour.
template <int BS, int P, bool FMA>
__global__ __launch_bounds__(BS) void fps_reg_kernel(int n, int m, const float* __restrict__ xyz,
                                                      int* __restrict__ out) {
    static_assert(BS == 64 || BS == 256 || BS == 1024, "visit order is derived for these block sizes");
    static_assert(BS != 64 || P <= 8, "BS=64 needs k < 512");
    constexpr int W = BS / kWave;
    __shared__ uint64_t slot[2][W > 1 ? W : 1];
    const int cloud = blockIdx.x;
    const float* __restrict__ p = xyz + (size_t)cloud * n * 3;
    int* __restrict__ o = out + (size_t)cloud * m;
    const int tid = threadIdx.x;

    float x[P], y[P], z[P], td[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int k = tid + i * BS;
        const bool ok = k < n;
        x[i] = ok ? p[k * 3 + 0] : 0.f;
        y[i] = ok ? p[k * 3 + 1] : 0.f;
        z[i] = ok ? p[k * 3 + 2] : 0.f;
        td[i] = ok ? 1e38f : -1.0f;  // -1 never beats the initial best of -1: "no point here"
    }
    // Small clouds also keep a copy of the coordinates in LDS: the winner's xyz is then one broadcast ds_read
    // (~60 ns) instead of a scalar load that misses the constant cache every round (~250 ns of a ~450 ns round).
    constexpr bool LDSXYZ = (BS * P <= 4096);
    __shared__ float sxyz[LDSXYZ ? BS * P * 3 : 1];
    if constexpr (LDSXYZ) {
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const int k = tid + i * BS;
            sxyz[k * 3 + 0] = x[i]; sxyz[k * 3 + 1] = y[i]; sxyz[k * 3 + 2] = z[i];
        }
        __syncthreads();
    }
    if (tid == 0) o[0] = 0;
    int old = 0;
    for (int j = 1; j < m; ++j) {
        float x1, y1, z1;
        if constexpr (LDSXYZ) {
            x1 = sxyz[old * 3 + 0]; y1 = sxyz[old * 3 + 1]; z1 = sxyz[old * 3 + 2];
        } else {
            const int so = __builtin_amdgcn_readfirstlane(old);
            x1 = p[so * 3 + 0]; y1 = p[so * 3 + 1]; z1 = p[so * 3 + 2];
        }
        float bd = -1.0f;
        int bi = 0;
#pragma unroll
        for (int s = 0; s < P; ++s) {
            const int i = fps_visit<BS, P>(s);
            const float d = sqdist3<FMA>(x[i] - x1, y[i] - y1, z[i] - z1);
            const float d2 = fminf(d, td[i]);
            td[i] = d2;
            if (d2 > bd) { bd = d2; bi = i; }
        }
        uint64_t best = fps_wave_best(bd, tid + bi * BS);
        if constexpr (W > 1) {
            const int par = j & 1;
            if ((tid & (kWave - 1)) == 0) slot[par][tid / kWave] = best;
            __syncthreads();
            best = fps_slots_best<W>(slot[par], tid & (kWave - 1));
        }
        old = fps_key_to_index(best);
        if (tid == 0) o[j] = old;
    }
}

// Fallback for clouds that do not fit the register-resident variants: running distances in
// caller scratch `temp` [b,n] (same role as the reference's temp, tf_sampling.cpp:115).
template <bool FMA>
__global__ __launch_bounds__(1024) void fps_mem_kernel(int n, int m, const float* __restrict__ xyz,
                                                        float* __restrict__ temp, int* __restrict__ out) {
    constexpr int BS = 1024, W = BS / kWave;
    __shared__ uint64_t slot[2][W];
    const int cloud = blockIdx.x;
    const float* __restrict__ p = xyz + (size_t)cloud * n * 3;
    float* __restrict__ t = temp + (size_t)cloud * n;
    int* __restrict__ o = out + (size_t)cloud * m;
    const int tid = threadIdx.x;
    for (int k = tid; k < n; k += BS) t[k] = 1e38f;
    if (tid == 0) o[0] = 0;
    int old = 0;
    for (int j = 1; j < m; ++j) {
        const int so = __builtin_amdgcn_readfirstlane(old);
        const float x1 = p[so * 3 + 0], y1 = p[so * 3 + 1], z1 = p[so * 3 + 2];
        float bd = -1.0f;
        int bk = 0;
        for (int k = tid; k < n; k += BS) {  // k mod 512 is constant per thread: ascending k == tie priority
            const float d = sqdist3<FMA>(p[k * 3 + 0] - x1, p[k * 3 + 1] - y1, p[k * 3 + 2] - z1);
            const float d2 = fminf(d, t[k]);
            t[k] = d2;
            if (d2 > bd) { bd = d2; bk = k; }
        }
        uint64_t best = fps_wave_best(bd, bk);
        const int par = j & 1;
        if ((tid & (kWave - 1)) == 0) slot[par][tid / kWave] = best;
        __syncthreads();
        best = fps_slots_best<W>(slot[par], tid & (kWave - 1));
        old = fps_key_to_index(best);
        if (tid == 0) o[j] = old;
    }
}

template <int BS, int P>
static int launch_fps_reg(int b, int n, int m, const float* xyz, int* out, int arith, hipStream_t s) {
    if ((arith & DISPU_ARITH_CONTRACT))
        hipLaunchKernelGGL((fps_reg_kernel<BS, P, true>), dim3(b), dim3(BS), 0, s, n, m, xyz, out);
    else
        hipLaunchKernelGGL((fps_reg_kernel<BS, P, false>), dim3(b), dim3(BS), 0, s, n, m, xyz, out);
    return (int)hipGetLastError();
}

// out[i,j,:] = inp[i, idx[i,j], :]   (c == 3, like the reference kernel tf_sampling_g.cu:172-181): gather_xyz_kernel of
// csrc/grouping.hip (12-byte loads per lane, LDS-staged float4 stores).
int launch_gather_rows(size_t rows, int n, int c, size_t rows_per_cloud, const float* points, const int* idx, float* out,
                       hipStream_t st);

// inp_g[i, idx[i,j], :] += out_g[i,j,:]  on a zero-filled buffer (tf_sampling_g.cu:183-192, tf_sampling.cpp:174)
__global__ void gather_point_grad_kernel(int n, int m, size_t total, const float* __restrict__ out_g,
                                         const int* __restrict__ idx, float* __restrict__ inp_g) {
    for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < total; r += (size_t)gridDim.x * blockDim.x) {
        const size_t cloud = r / m;
        float* dst = inp_g + (cloud * n + idx[r]) * 3;
        const float* src = out_g + r * 3;
        unsafeAtomicAdd(dst + 0, src[0]);
        unsafeAtomicAdd(dst + 1, src[1]);
        unsafeAtomicAdd(dst + 2, src[2]);
    }
}

// csrc/fps_wave.hip: exact sampling with wave-level skipping for 8192 < n <= 24576 (scratch = the sorted permutation)
bool fps_wave_wants_scratch(int n, int m);
int fps_wave_dispatch(int b, int n, int m, const float* xyz, void* temp, int* out, int arith, hipStream_t s);

static bool fps_dense_only() { return false; }     // (the region-skipping kernels of csrc/fps_wave.hip whenever scratch is given)

// the dense (scratch-free for n <= 24576) kernels
static int fps_dense(int b, int n, int m, const float* inp, float* temp, int* out, int arith, hipStream_t s) {
    if (n <= 64) return launch_fps_reg<64, 1>(b, n, m, inp, out, arith, s);
    if (n <= 128) return launch_fps_reg<64, 2>(b, n, m, inp, out, arith, s);
    if (n <= 256) return launch_fps_reg<64, 4>(b, n, m, inp, out, arith, s);
    if (n <= 512) return launch_fps_reg<256, 2>(b, n, m, inp, out, arith, s);
    if (n <= 1024) return launch_fps_reg<256, 4>(b, n, m, inp, out, arith, s);
    if (n <= 2048) return launch_fps_reg<256, 8>(b, n, m, inp, out, arith, s);
    if (n <= 4096) return launch_fps_reg<1024, 4>(b, n, m, inp, out, arith, s);
    if (n <= 8192) return launch_fps_reg<1024, 8>(b, n, m, inp, out, arith, s);
    if (n <= 16384) return launch_fps_reg<1024, 16>(b, n, m, inp, out, arith, s);
    if (n <= 24576) return launch_fps_reg<1024, 24>(b, n, m, inp, out, arith, s);
    if (!temp) return (int)hipErrorInvalidValue;
    if ((arith & DISPU_ARITH_CONTRACT))
        hipLaunchKernelGGL((fps_mem_kernel<true>), dim3(b), dim3(1024), 0, s, n, m, inp, temp, out);
    else
        hipLaunchKernelGGL((fps_mem_kernel<false>), dim3(b), dim3(1024), 0, s, n, m, inp, temp, out);
    return (int)hipGetLastError();
}

// ---- prob_sample: cumulative sums + binary search (tf_sampling_g.cu:7-104) ----------------------------------------------
// One workgroup per row.  The association of the sums is part of the result (an index flips when a cumulative sum moves by
// one ulp across r * total), so the kernel keeps the reference's: chunks of 8192 values; per aligned quad
// (v0, v0+v1, v2+(v0+v1), (v3+v2)+(v0+v1)); the 2048 quad totals through a work-efficient scan in LDS (up-sweep
// t[((2k+2)<<u)-1] += t[((2k+1)<<u)-1], then down-sweep t[((2k+3)<<u)-1] += t[((2k+2)<<u)-1], a pair taking part iff its
// target exists); element = (inquad + t[quad-1]) + running; the running total is carried between chunks with its
// compensation term.  Restated in oracle/dispu_oracle.c:orc_prob_sample.
constexpr int PSM_CH = 8192, PSM_Q = PSM_CH / 4, PSM_BS = 1024;

__global__ __launch_bounds__(PSM_BS) void prob_cumsum_kernel(int n, const float* __restrict__ inp, float* __restrict__ out) {
    __shared__ float in4[PSM_CH];
    __shared__ float tot[PSM_Q];
    const int tid = threadIdx.x;
    const float* __restrict__ x = inp + (size_t)blockIdx.x * n;
    float* __restrict__ cum = out + (size_t)blockIdx.x * n;
    float running = 0.0f, comp = 0.0f;                  // wave-uniform: every thread tracks the same two values
    for (int j = 0; j < n; j += PSM_CH) {
        const int cnt = min(n - j, PSM_CH);
        const int n24 = (cnt + 3) & ~3, n2 = n24 >> 2;
        for (int q = tid; q < n2; q += PSM_BS) {
            const int k = q * 4;
            if (k + 3 < cnt) {
                const float v1 = x[j + k];
                const float v2 = x[j + k + 1] + v1;
                float v3 = x[j + k + 2];
                float v4 = x[j + k + 3] + v3;
                v3 = v3 + v2;
                v4 = v4 + v2;
                in4[k] = v1; in4[k + 1] = v2; in4[k + 2] = v3; in4[k + 3] = v4;
                tot[q] = v4;
            } else {
                float v = 0.0f;
                for (int k2 = k; k2 < n24; ++k2) {
                    if (k2 < cnt) v = v + x[j + k2];
                    in4[k2] = v;
                }
                tot[q] = v;
            }
        }
        int levels = 0;
        while ((2 << levels) <= n2) ++levels;
        for (int u = 0; u < levels; ++u) {
            __syncthreads();
            for (int k = tid; k < (n2 >> (u + 1)); k += PSM_BS) tot[((2 * k + 2) << u) - 1] += tot[((2 * k + 1) << u) - 1];
        }
        for (int u = levels - 1; u >= 0; --u) {
            __syncthreads();
            for (int k = tid; k < ((n2 - (1 << u)) >> (u + 1)); k += PSM_BS) tot[((2 * k + 3) << u) - 1] += tot[((2 * k + 2) << u) - 1];
        }
        __syncthreads();
        for (int k = tid; k < cnt; k += PSM_BS) {
            float v = in4[k];
            if (k >= 4) v = v + tot[(k >> 2) - 1];
            cum[j + k] = v + running;
        }
        const float t = tot[n2 - 1] + comp;
        const float r2 = running + t;
        comp = t - (r2 - running);
        running = r2;
        __syncthreads();
    }
}

__global__ void prob_search_kernel(int n, int m, int base, const float* __restrict__ cum, const float* __restrict__ r,
                                   int* __restrict__ out) {
    const float* __restrict__ c = cum + (size_t)blockIdx.y * n;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const float q = r[(size_t)blockIdx.y * m + j] * c[n - 1];
    int pos = n - 1;
    for (int k = base; k >= 1; k >>= 1)
        if (pos >= k && c[pos - k] >= q) pos -= k;
    out[(size_t)blockIdx.y * m + j] = pos;
}

}  // namespace dispu

using namespace dispu;

DISPU_EXPORT size_t dispu_fps_scratch_bytes(int b, int n, int m) {
    if (n > 24576 || (fps_wave_wants_scratch(n, m) && !fps_dense_only())) return (size_t)b * n * sizeof(float);
    return 0;
}

DISPU_EXPORT int dispu_fps_ws(int b, int n, int m, const float* inp, float* temp, size_t temp_bytes, int* out, int arith,
                              void* stream) {
    if (b < 0 || n <= 0 || m <= 0) return (int)hipErrorInvalidValue;
    if (b == 0) return 0;                              // empty tensors carry null pointers
    if (!inp || !out) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    const size_t need = (size_t)b * n * sizeof(float);
    if (!temp) temp_bytes = 0;
    if (n > 4096 && n <= 24576 && !fps_dense_only() && temp_bytes >= need) {   // region skipping needs the b*n-int permutation
        const int r = fps_wave_dispatch(b, n, m, inp, temp, out, arith, s);
        if (r >= 0) return r;
    }
    if (n > 24576 && temp_bytes < need) return (int)hipErrorInvalidValue;       // running distances of the streaming kernel
    return fps_dense(b, n, m, inp, temp, out, arith, s);
}

// The reference's launcher signature: its op allocates temp as {32, n} floats whatever b is (tf_sampling.cpp:115), so this entry
// never touches more than min(b, 32) * n floats of it: batches above 32 clouds run as consecutive groups of 32 on the stream.
DISPU_EXPORT int dispu_fps(int b, int n, int m, const float* inp, float* temp, int* out, int arith, void* stream) {
    if (b < 0 || n <= 0 || m <= 0) return (int)hipErrorInvalidValue;
    for (int lo = 0; lo < b; lo += 32) {
        const int nb = b - lo < 32 ? b - lo : 32;
        const int r = dispu_fps_ws(nb, n, m, inp ? inp + (size_t)lo * n * 3 : nullptr, temp,
                                   temp ? (size_t)nb * n * sizeof(float) : 0, out ? out + (size_t)lo * m : nullptr, arith, stream);
        if (r != 0) return r;
    }
    return 0;
}

// probsampleLauncher(b,n,m,inp_p,inp_r,temp,out)  tf_sampling.cpp:65,83-89: temp [b, n] receives the cumulative sums.
DISPU_EXPORT int dispu_prob_sample(int b, int n, int m, const float* inp_p, const float* inp_r, float* temp, int* out,
                                   void* stream) {
    if (b < 0 || n <= 0 || m < 0) return (int)hipErrorInvalidValue;
    if (b == 0) return 0;
    if (!inp_p || !temp || (m > 0 && (!inp_r || !out))) return (int)hipErrorInvalidValue;
    if (b > 65535) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(prob_cumsum_kernel, dim3(b), dim3(PSM_BS), 0, s, n, inp_p, temp);
    DISPU_CHECK_LAUNCH();
    if (m == 0) return 0;
    int base = 1;
    while (base < n) base <<= 1;
    hipLaunchKernelGGL(prob_search_kernel, dim3((m + 255) / 256, b), dim3(256), 0, s, n, m, base, temp, inp_r, out);
    return (int)hipGetLastError();
}

static inline int grid_for(size_t total, int bs) {
    size_t g = (total + bs - 1) / bs;
    if (g > 8192) g = 8192;
    if (g < 1) g = 1;
    return (int)g;
}

DISPU_EXPORT int dispu_gather_point(int b, int n, int m, const float* inp, const int* idx, float* out, void* stream) {
    if (b < 0 || n <= 0 || m < 0) return (int)hipErrorInvalidValue;
    return launch_gather_rows((size_t)b * m, n, 3, (size_t)m, inp, idx, out, (hipStream_t)stream);
}

DISPU_EXPORT int dispu_gather_point_grad(int b, int n, int m, const float* out_g, const int* idx, float* inp_g,
                                         void* stream) {
    if (b < 0 || n <= 0 || m < 0) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    if ((size_t)b * n) DISPU_TRY(hipMemsetAsync(inp_g, 0, sizeof(float) * (size_t)b * n * 3, s));
    const size_t total = (size_t)b * m;
    if (total == 0) return 0;
    hipLaunchKernelGGL(gather_point_grad_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, n, m, total, out_g, idx,
                       inp_g);
    return (int)hipGetLastError();
}
