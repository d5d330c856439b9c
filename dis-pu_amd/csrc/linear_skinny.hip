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
    const float* W; long ldw;         // [K, N], or [N, K] when TRANSB
    const float* bias;
    int act;
    float* Y; long ldy;
    const float* R1; long ldr1;       // optional residual added after the activation
    const float* Mk; long ldm; int mcols;   // optional ReLU-gradient mask (dispu_linear_masked): Y = 0 where Mk <= 0, columns < mcols
};

__device__ __forceinline__ void swap32(float& a, float& b) {      // a's lanes 32..63 <-> b's lanes 0..31
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void swap16(float& a, float& b) {      // a's odd 16-lane rows <-> b's even rows
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
// v.c of 16-lane row q := (component q of row c): afterwards v.c holds element 4 c + q of the lane's 16-element group
__device__ __forceinline__ void transpose4(float4& v) {
    swap32(v.x, v.z); swap32(v.y, v.w);
    swap16(v.x, v.y); swap16(v.z, v.w);
}

typedef unsigned int sk_u32x4 __attribute__((ext_vector_type(4)));

// float4 strip of one operand: rows = the lane's row (A: X row, TRANSB B: W row = output column), 16 k per group
template <int NG>
__device__ __forceinline__ void load_strip(float4 (&v)[NG], __amdgpu_buffer_rsrc_t rs, int off, int K, int q) {
#pragma unroll
    for (int u = 0; u < NG; ++u) {
        // full groups: scalar k offset; the partial group needs the per-lane clamp (lanes whose 4 q part is past K)
        const sk_u32x4 r = (16 * u + 16 <= K) ? __builtin_amdgcn_raw_buffer_load_b128(rs, off, 64 * u, 0)
                                              : __builtin_amdgcn_raw_buffer_load_b128(rs, off + 4 * min(16 * u, K - 4 - 4 * q), 0, 0);
        v[u] = make_float4(__uint_as_float(r[0]), __uint_as_float(r[1]), __uint_as_float(r[2]), __uint_as_float(r[3]));
    }
}

template <int NG, bool TRANSB>                                     // K <= 16 NG, everything in registers
__global__ __launch_bounds__(256) void linear_skinny_kernel(SkinnyArgs a) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.y * 4 + (threadIdx.x >> 6);               // 16-column tile of this wave
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
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.X), 0, (int)((long)a.M * a.ldx * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.W), 0, (int)((long)(TRANSB ? a.N : K) * a.ldw * 4), 0x00020000);
    const int xoff = (min(row0 + i, a.M - 1) * (int)a.ldx + 4 * q) * 4;                    // bytes
    float4 av[NG];
    load_strip<NG>(av, rx, xoff, K, q);
    float bv[4 * NG];
    if constexpr (TRANSB) {
        // B[k][n] = W[n][k]: the same float4-along-k strip as A, of W's row `col`, and the same lane transpose
        float4 bw[NG];
        load_strip<NG>(bw, rw, (min(col, a.N - 1) * (int)a.ldw + 4 * q) * 4, K, q);
#pragma unroll
        for (int u = 0; u < NG; ++u) {
            if (16 * u < K) {
                transpose4(bw[u]);
                bv[4 * u + 0] = bw[u].x; bv[4 * u + 1] = bw[u].y; bv[4 * u + 2] = bw[u].z; bv[4 * u + 3] = bw[u].w;
            }
        }
    } else {
        const int boff = (q * (int)a.ldw + min(col, a.N - 1)) * 4;
        const int ldw4 = (int)a.ldw * 4;
#pragma unroll
        for (int s = 0; s < 4 * NG; ++s) bv[s] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rw, boff, min(4 * s, K - 4) * ldw4, 0));
    }
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
            transpose4(v);                                          // v.c (row q) = X[i][16 u + 4 c + q]
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
        if (row < a.M && cok) {
            float v = acc[r];
            if (a.bias) v = v + bb;
            v = fmaxf(v, lo);
            if (a.R1) v = v + a.R1[(size_t)row * a.ldr1 + col];
            if (a.Mk && col < a.mcols) v = (a.Mk[(size_t)row * a.ldm + col] > 0.f) ? v : 0.f;
            a.Y[(size_t)row * a.ldy + col] = v;
        }
    }
}

template <int NG>
static void launch_skinny(const SkinnyArgs& a, bool transb, hipStream_t st) {
    const int ntile = (a.N + 15) / 16;
    const dim3 grid((a.M + 15) / 16, (ntile + 3) / 4), block(64 * (ntile < 4 ? ntile : 4));
    if (transb) hipLaunchKernelGGL((linear_skinny_kernel<NG, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((linear_skinny_kernel<NG, false>), grid, block, 0, st, a);
}

// Returns -1 when the shape is outside this path (the caller then uses the tiled kernel).
int linear_skinny_dispatch(int M, int K, int N, const float* X, long ldx, const float* W, long ldw, int transb, const float* bias,
                           int act, float* Y, long ldy, const float* R1, long ldr1, const float* Mk, long ldm, int mcols,
                           hipStream_t st) {
    if (K > 384 || K < 4 || (K & 3) || (ldx & 3) || (((uintptr_t)X) & 15)) return -1;
    if (transb && ((ldw & 3) || (((uintptr_t)W) & 15))) return -1;
    if ((long)M * ldx >= (1l << 29) || (long)(transb ? N : K) * ldw >= (1l << 29)) return -1;     // 32-bit byte offsets (buffer loads)
    // where the tiled kernel does badly: too few 64 x 64 tiles to fill the chip (latency-bound), outputs narrower than half
    // a tile, or a contraction shorter than one K-slab (then it only moves data, mostly through predicated edge paths)
    const long tiles64 = (long)((M + 63) / 64) * ((N + 63) / 64);
    if (!(N <= 64 && tiles64 < 256) && !(N <= 32) && !(K <= 32 && N <= 128)) return -1;
    SkinnyArgs a{M, K, N, X, ldx, W, ldw, bias, act, Y, ldy, R1, ldr1, (Mk && mcols > 0) ? Mk : nullptr, ldm, mcols};
    if (K <= 32) launch_skinny<2>(a, transb != 0, st);
    else if (K <= 128) launch_skinny<8>(a, transb != 0, st);
    else if (K <= 256) launch_skinny<16>(a, transb != 0, st);
    else launch_skinny<24>(a, transb != 0, st);
    return (int)hipGetLastError();
}

}  // namespace dispu
