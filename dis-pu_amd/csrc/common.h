// Shared device helpers for libdispu_hip.so (gfx950 / CDNA4 only, wave64).
// Compiled with -ffp-contract=off: every fused multiply-add in this library is an explicit
// __builtin_fmaf, so the arithmetic is pinned (see DESIGN.md "Pinned arithmetic").
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

#define DISPU_EXPORT extern "C" __attribute__((visibility("default")))

// Arithmetic flavour flags shared with include/dispu_hip.h
#define DISPU_ARITH_PLAIN 0     // ((dx*dx + dy*dy) + dz*dz): the reference's CPU functions
#define DISPU_ARITH_CONTRACT 1  // fmaf(dz,dz, fmaf(dx,dx, dy*dy)): nvcc-contracted GPU kernels
#define DISPU_ARITH_PINNED_EXP 2  // OR-able: approx_match uses the bit-reproducible exp (parity mode)
#define DISPU_KNN_LANE_PER_QUERY 4  // OR-able, dispu_knn_xyz: force the lane-per-query kernel (A/B tests)

#define DISPU_CHECK_LAUNCH()                         \
    do {                                             \
        hipError_t e__ = hipGetLastError();          \
        if (e__ != hipSuccess) return (int)e__;      \
    } while (0)

#define DISPU_TRY(expr)                              \
    do {                                             \
        hipError_t e__ = (expr);                     \
        if (e__ != hipSuccess) return (int)e__;      \
    } while (0)

// A split reduction a TN (weight-gradient) product leaves undone when the caller asked for that (dispu_tn_defer, include/dispu_hip.h):
// out[k][n] (+)= sum_s part[s][k][n], row K of the partials (rows_p == K + 1) -> dbias[n].  Mirrors dispu_tn_reduce_desc.
struct dispu_tn_reduce_desc {
    const float* part; float* out; float* dbias;
    long ldo, stride;                       // row stride of out; floats between consecutive partials
    int K, N, splits, rows_p;
    int accumulate, bias_accumulate, assoc, reserved;   // assoc 0: the fp32 kernels' order (tn_reduce_kernel), 1: the bf16 kernels'
};
static_assert(sizeof(dispu_tn_reduce_desc) == 72, "descriptor layout is part of the C-ABI");

namespace dispu {

constexpr int kWave = 64;

// the one-shot sink dispu_tn_defer arms for the NEXT TN product on this thread (train_gemm.hip); taking it disarms it
dispu_tn_reduce_desc* tn_take_defer();

// "done once per DEVICE" flag for hipFuncSetAttribute (function attributes are per device: a process that drives several
// GPUs must opt every one of them in to > 64 KB of dynamic LDS).  Two threads racing through the first call both set the
// attribute, which is idempotent.
struct DevOnce {
    std::atomic<unsigned long long> mask{0};
    static int cur() { int d = 0; (void)hipGetDevice(&d); return d & 63; }
    bool needed() const { return !((mask.load(std::memory_order_acquire) >> cur()) & 1ull); }
    void done() { mask.fetch_or(1ull << cur(), std::memory_order_release); }
};

template <bool FMA>
__device__ __forceinline__ float sqdist3(float dx, float dy, float dz) {
    if constexpr (FMA) {
        return __builtin_fmaf(dz, dz, __builtin_fmaf(dx, dx, dy * dy));
    } else {
        const float s = dx * dx + dy * dy;
        return s + dz * dz;
    }
}

// Two distances per instruction on the packed fp32 ops (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32); element-wise the arithmetic is
// sqdist3's, operation for operation.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <bool FMA>
__device__ __forceinline__ f32x2 sqdist3_x2(f32x2 dx, f32x2 dy, f32x2 dz) {
    if constexpr (FMA) {
        return __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dx, dx, dy * dy));
    } else {
        const f32x2 s = dx * dx + dy * dy;
        return s + dz * dz;
    }
}

// A cloud [n][3] into LDS once per workgroup of NT threads (coalesced float4, UB per thread in flight), for the wave-per-query kernels
// whose lanes keep R candidates each in registers: per-lane global loads (3 R strided dwords per lane, the same 12 KB in every wave of
// the launch) cost a quarter of such a kernel (knn_xyz_wave_kernel: 8.8 k of 40 k cycles per wave).  Caller: __syncthreads() after.
template <int NT, int UB>
__device__ __forceinline__ void stage_cloud_xyz(float* sl, const float* __restrict__ s, int n) {
    const int nf = 3 * n;
    if ((((uintptr_t)s) & 15) == 0) {
        const int nf4 = nf >> 2;
        for (int e0 = threadIdx.x; e0 < nf4; e0 += UB * NT) {
            float4 v[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) v[u] = *reinterpret_cast<const float4*>(s + (size_t)min(e0 + u * NT, nf4 - 1) * 4);
#pragma unroll
            for (int u = 0; u < UB; ++u)
                if (e0 + u * NT < nf4) *reinterpret_cast<float4*>(sl + (e0 + u * NT) * 4) = v[u];
        }
        for (int e = (nf4 << 2) + threadIdx.x; e < nf; e += NT) sl[e] = s[e];
    } else {
        for (int e = threadIdx.x; e < nf; e += NT) sl[e] = s[e];
    }
}

// ---- DPP wave64 reductions (result valid in lane 63; use readlane to broadcast) ---------------
// row_shr:1,2,4,8 leave each 16-lane row's reduction in its lane 15; row_bcast:15 / :31 fold the
// four rows into lane 63.  `old` (identity) fills lanes that have no DPP source.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t identity, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, CTRL, ROW_MASK, 0xF, false);
}

constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118,
              DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;

__device__ __forceinline__ uint64_t u64_max(uint64_t a, uint64_t b) { return a > b ? a : b; }

template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ uint64_t dpp_step_max_u64(uint64_t v) {
    const uint32_t hi = dpp_u32<CTRL, ROW_MASK>(0u, (uint32_t)(v >> 32));
    const uint32_t lo = dpp_u32<CTRL, ROW_MASK>(0u, (uint32_t)v);
    return u64_max(v, ((uint64_t)hi << 32) | lo);
}

// Wave-wide maximum of an unsigned 64-bit key, returned wave-uniform (in SGPRs).
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) {
    v = dpp_step_max_u64<DPP_ROW_SHR1>(v);
    v = dpp_step_max_u64<DPP_ROW_SHR2>(v);
    v = dpp_step_max_u64<DPP_ROW_SHR4>(v);
    v = dpp_step_max_u64<DPP_ROW_SHR8>(v);
    v = dpp_step_max_u64<DPP_ROW_BCAST15, 0xA>(v);
    v = dpp_step_max_u64<DPP_ROW_BCAST31, 0xC>(v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), 63);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 63);
    return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ uint64_t u64_min(uint64_t a, uint64_t b) { return a < b ? a : b; }

template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ uint64_t dpp_step_min_u64(uint64_t v) {
    const uint32_t hi = dpp_u32<CTRL, ROW_MASK>(0xFFFFFFFFu, (uint32_t)(v >> 32));
    const uint32_t lo = dpp_u32<CTRL, ROW_MASK>(0xFFFFFFFFu, (uint32_t)v);
    return u64_min(v, ((uint64_t)hi << 32) | lo);
}

// Wave-wide minimum of an unsigned 64-bit key, returned wave-uniform.
__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v) {
    v = dpp_step_min_u64<DPP_ROW_SHR1>(v);
    v = dpp_step_min_u64<DPP_ROW_SHR2>(v);
    v = dpp_step_min_u64<DPP_ROW_SHR4>(v);
    v = dpp_step_min_u64<DPP_ROW_SHR8>(v);
    v = dpp_step_min_u64<DPP_ROW_BCAST15, 0xA>(v);
    v = dpp_step_min_u64<DPP_ROW_BCAST31, 0xC>(v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), 63);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 63);
    return ((uint64_t)hi << 32) | lo;
}

// float <-> unsigned key whose unsigned order is the float order (handles negative values)
__device__ __forceinline__ uint32_t f32_to_ordered(float f) {
    const uint32_t b = __float_as_uint(f);
    return b ^ ((uint32_t)((int32_t)b >> 31) | 0x80000000u);
}
__device__ __forceinline__ float ordered_to_f32(uint32_t u) {
    const uint32_t b = (u & 0x80000000u) ? (u ^ 0x80000000u) : ~u;
    return __uint_as_float(b);
}

// v_mfma_f32_32x32x2_f32 with the accumulator pinned to the ACC register file.  hipcc prefers VGPR accumulators when
// the budget allows; a wave issuing back-to-back MFMAs then reads and writes 16 VGPRs x 64 lanes per instruction and
// keeps the VGPR ports of its SIMD busy, so that the OTHER wave on the SIMD (a loader / helper wave) needs ~340
// cycles to issue one global load instead of ~40 (tools/micro/gemm_lab.hip, "AGPR acc").  The compiler does not see
// an MFMA in the asm statement: call mfma_acc_settle() once before the accumulators are read.
typedef float dispu_f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void mfma_acc(dispu_f32x16& acc, float a, float b) {
    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_acc_settle() {   // >= 18 wait states between the last 16-pass MFMA and a read of its result
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
}

// 16-byte non-temporal store (global_store_dwordx4 ... nt): streamed outputs that are not re-read by the kernel
typedef float dispu_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_nt4(float* dst, float4 v) {
    __builtin_nontemporal_store(dispu_f32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<dispu_f32x4*>(dst));
}

// Workgroup ids are dealt round-robin to the 8 XCDs (each with its own L2).  xcd_block() turns the hardware id into a
// logical block index such that every XCD works on ONE contiguous range of logical blocks: blocks that gather from
// the same cloud then share an L2 instead of pulling the cloud's rows into all eight.
__device__ __forceinline__ unsigned xcd_block(unsigned bid, unsigned grid) {
    const unsigned per = grid >> 3;
    return (grid & 7u) ? bid : (bid & 7u) * per + (bid >> 3);
}

__device__ __forceinline__ float wave_sum_f32(float v) {
    // butterfly; order fixed (xor 32,16,8,4,2,1) so the result is deterministic
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace dispu
