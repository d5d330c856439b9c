// Latency-bound GEMMs of the dense blocks' bottleneck convs (feature_extraction layer<d>_prep, ops.py:1455-1462):
//   Y[M, N] = act(X[M, K] . W[K, N] + bias),  M = B * 256 rows, N = 48, K = 120 / 240 / 360.
// The tiled kernel in linear.hip gives them 128 workgroups that walk K slab by slab (load -> LDS -> barrier -> 16 MFMAs):
// 7 - 15 us for 0.1 - 0.3 GFLOP, nearly all of it exposed memory latency.  Here NOTHING is staged: a wave owns one
// 16 x 16 output tile, requests its whole A strip (16 rows x K, float4 per lane) and B strip (K x 16) up front - every
// load of the kernel is in flight at once - and then runs K / 4 v_mfma_f32_16x16x4_f32.  That instruction is bit for bit
// the ascending-k fmaf chain (tools/micro/mfma16_exact.hip: 0 of 51200 outputs differ), i.e. the same arithmetic as
// v_mfma_f32_32x32x2_f32 in linear.hip and as the oracle's loop, with a 4x shorter dependent chain per k.
// A operand: lane (i = lane & 15, q = lane >> 4) must supply X[i][4 s + q] for step s, but loads float4 X[i][16 u + 4 q ..]:
// a 4 x 4 transpose between the four 16-lane rows and the float4 components, done with gfx950's
// v_permlane32_swap / v_permlane16_swap (4 instructions per float4, no LDS).
#include "common.h"

namespace dispu {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct SkinnyArgs {
    int M, K, N;
    const float* X; long ldx;
    const float* W; long ldw;
    const float* bias;
    int act;
    float* Y; long ldy;
};

__device__ __forceinline__ void swap32(float& a, float& b) {      // a's lanes 32..63 <-> b's lanes 0..31
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void swap16(float& a, float& b) {      // a's odd 16-lane rows <-> b's even rows
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}

template <int NG>                                                  // K <= 16 NG, everything in registers
__global__ __launch_bounds__(256) void linear_skinny_kernel(SkinnyArgs a) {
    const int lane = threadIdx.x & 63, t = threadIdx.x >> 6;        // t: 16-column tile of this wave
    const int i = lane & 15, q = lane >> 4;
    const int row0 = blockIdx.x * 16;
    const int K = a.K;
    const int col = 16 * t + i;
    const bool cok = col < a.N;
    // every load of the kernel is issued here, unconditionally (predicated loads compiled to 120 exec branches with a
    // vmcnt(0) in the second one).  Buffer loads: descriptor in SGPRs, ONE per-lane byte offset per operand, and the k
    // part of the address as the instruction's scalar offset - no VALU work per load (flat 64-bit per-lane addresses
    // cost ~3 VALU instructions per load, longer in total than the memory round trip).  k past K is clamped into valid
    // memory; its B value is zeroed below.
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.X), 0, (int)((long)a.M * a.ldx * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.W), 0, (int)((long)K * a.ldw * 4), 0x00020000);
    const int xoff = (min(row0 + i, a.M - 1) * (int)a.ldx + 4 * q) * 4;                    // bytes
    const int boff = (q * (int)a.ldw + min(col, a.N - 1)) * 4;
    const int ldw4 = (int)a.ldw * 4;
    float4 av[NG];
    float bv[4 * NG];
#pragma unroll
    for (int u = 0; u < NG; ++u) {
        // full groups: scalar k offset; the partial group needs the per-lane clamp (lanes whose 4 q part is past K)
        const u32x4 r = (16 * u + 16 <= K) ? __builtin_amdgcn_raw_buffer_load_b128(rx, xoff, 64 * u, 0)
                                           : __builtin_amdgcn_raw_buffer_load_b128(rx, xoff + 4 * min(16 * u, K - 4 - 4 * q) , 0, 0);
        av[u] = make_float4(__uint_as_float(r[0]), __uint_as_float(r[1]), __uint_as_float(r[2]), __uint_as_float(r[3]));
    }
#pragma unroll
    for (int s = 0; s < 4 * NG; ++s) bv[s] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rw, boff, min(4 * s, K - 4) * ldw4, 0));
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NG; ++u) {
        if (16 * u < K) {                                           // wave-uniform (a `break` here kept the loop rolled: arrays in scratch)
            float4 v = av[u];
            float b0 = bv[4 * u + 0], b1 = bv[4 * u + 1], b2 = bv[4 * u + 2], b3 = bv[4 * u + 3];
            if (16 * u + 16 > K) {                                  // the one partial group: A holds (finite) clamped data, B must be 0
                b0 = (16 * u + 0 + q < K) ? b0 : 0.f; b1 = (16 * u + 4 + q < K) ? b1 : 0.f;
                b2 = (16 * u + 8 + q < K) ? b2 : 0.f; b3 = (16 * u + 12 + q < K) ? b3 : 0.f;
            }
            swap32(v.x, v.z); swap32(v.y, v.w);
            swap16(v.x, v.y); swap16(v.z, v.w);                     // v.c (row q) = X[i][16 u + 4 c + q]
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v.x, b0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v.y, b1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v.z, b2, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v.w, b3, acc, 0, 0, 0);
        }
    }
    // C/D layout: acc[r] = Y[row0 + 4 q + r][16 t + i]
    const float lo = (a.act == 1) ? 0.f : -__builtin_inff();
    const float bb = (a.bias && cok) ? a.bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = row0 + 4 * q + r;
        float v = acc[r];
        if (a.bias) v = v + bb;
        v = fmaxf(v, lo);
        if (row < a.M && cok) a.Y[(size_t)row * a.ldy + col] = v;
    }
}

// Returns -1 when the shape is outside this path (the caller then uses the tiled kernel).
int linear_skinny_dispatch(int M, int K, int N, const float* X, long ldx, const float* W, long ldw, const float* bias, int act,
                           float* Y, long ldy, hipStream_t st) {
    if (N > 64 || K > 384 || K < 4 || (K & 3) || (ldx & 3) || (((uintptr_t)X) & 15)) return -1;
    if ((long)M * ldx >= (1l << 29) || (long)K * ldw >= (1l << 29)) return -1;     // 32-bit byte offsets (buffer loads)
    const long tiles64 = (long)((M + 63) / 64) * ((N + 63) / 64);
    if (tiles64 >= 256) return -1;                                  // enough workgroups for the tiled kernel to be MFMA-bound
    SkinnyArgs a{M, K, N, X, ldx, W, ldw, bias, act, Y, ldy};
    const dim3 grid((M + 15) / 16), block(64 * ((N + 15) / 16));
    if (K <= 128) hipLaunchKernelGGL(linear_skinny_kernel<8>, grid, block, 0, st, a);
    else if (K <= 256) hipLaunchKernelGGL(linear_skinny_kernel<16>, grid, block, 0, st, a);
    else hipLaunchKernelGGL(linear_skinny_kernel<24>, grid, block, 0, st, a);
    return (int)hipGetLastError();
}

}  // namespace dispu
