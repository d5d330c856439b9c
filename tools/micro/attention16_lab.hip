// EXPERIMENT: the fused non-local cell on v_mfma_f32_16x16x4_f32 tiles with EIGHT MFMA waves per workgroup (16 queries each,
// two per SIMD) instead of four waves of 32 queries: at B = 32 (32768 queries, 256 workgroups of 128) the 32x32x2 kernel has one
// MFMA wave per SIMD, and nothing covers its per-tile softmax (590 of 5000 cycles) and barrier.  Compared here against
// dispu_attention_project (result within the softmax branch's tolerance) and timed.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Idis-pu_amd/csrc -Iinclude tools/micro/attention16_lab.hip -o tools/micro/_fa16
#include "../../dis-pu_amd/csrc/attention.hip"
#include <cstdio>
#include <cmath>
#include <vector>

namespace lab {
using namespace dispu;
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int A_TK = 32, A_P = 68;                       // keys per tile, row pitch of the K / V tiles (2-way conflicts at most)
constexpr int A_STAGE = 2 * A_TK * A_P;                  // K tile + V tile
constexpr int A_WP = 260;                                // pitch of W [64][256] in LDS
constexpr size_t A_BYTES = (size_t)(2 * A_STAGE + 64 * A_WP) * sizeof(float);

__global__ __launch_bounds__(768) void attention16_kernel(int m, int nk, const float* __restrict__ Q, long ldq, const float* __restrict__ K,
                                                          long ldk, const float* __restrict__ V, long ldv, float scale,
                                                          float* __restrict__ Y, long ldy, const float* __restrict__ W,
                                                          const float* __restrict__ bias) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Wl = lds + 2 * A_STAGE;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cloud = blockIdx.y;
    const float* __restrict__ kb = K + (size_t)cloud * nk * ldk;
    const float* __restrict__ vb = V + (size_t)cloud * nk * ldv;
    const int ntile = nk / A_TK;
    if (wave >= 8) {
        // ------------------------------------------------------------ loader waves: 32 keys x 64 floats of K and of V per tile
        const int tid = threadIdx.x - 512;
        float4 pk[2], pv[2];
        auto load_tile = [&](int k0) {
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int e = tid + it * 256, key = e >> 4, c4 = e & 15;
                pk[it] = *reinterpret_cast<const float4*>(kb + (size_t)(k0 + key) * ldk + c4 * 4);
                pv[it] = *reinterpret_cast<const float4*>(vb + (size_t)(k0 + key) * ldv + c4 * 4);
            }
        };
        auto store_tile = [&](int stage) {
            float* Kt = lds + stage * A_STAGE;
            float* Vt = Kt + A_TK * A_P;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int e = tid + it * 256, key = e >> 4, c4 = e & 15;
                *reinterpret_cast<float4*>(&Kt[key * A_P + c4 * 4]) = pk[it];
                *reinterpret_cast<float4*>(&Vt[key * A_P + c4 * 4]) = pv[it];
            }
        };
        load_tile(0);
#pragma unroll
        for (int h = 0; h < 4; ++h) {                    // W [64][256] -> LDS, pitch 260
            float4 w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) w[u] = *reinterpret_cast<const float4*>(W + (size_t)(tid + (h * 4 + u) * 256) * 4);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = tid + (h * 4 + u) * 256, d = e >> 6, o4 = e & 63;
                *reinterpret_cast<float4*>(Wl + d * A_WP + o4 * 4) = w[u];
            }
        }
        store_tile(0);
        if (ntile > 1) load_tile(A_TK);
        __syncthreads();
        for (int t = 0; t < ntile; ++t) {
            if (t + 1 < ntile) {
                store_tile((t + 1) & 1);
                if (t + 2 < ntile) load_tile((t + 2) * A_TK);
            }
            __syncthreads();
        }
        return;
    }
    // ---------------------------------------------------------------- MFMA waves: 16 queries each; lane (q = lane % 16, g = lane / 16)
#ifdef A_PRIO
    if (wave < 4) __builtin_amdgcn_s_setprio(2);         // the two MFMA waves of a SIMD out of step: one's softmax under the other's MFMAs
#endif
    const int lc = lane & 15, lg = lane >> 4;
    const int qrow = blockIdx.x * 128 + wave * 16 + lc;
    const bool qok = qrow < m;
    const float* __restrict__ qp = Q + ((size_t)cloud * m + (qok ? qrow : 0)) * ldq;
    const float scale2 = scale * 1.4426950408889634f;
    float qf[16];                                        // Q[q][4 s + g] * scale2: the B operand of S^T = K . Q^T
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const float4 v = *reinterpret_cast<const float4*>(qp + s * 4);
        qf[s] = (lg == 0 ? v.x : lg == 1 ? v.y : lg == 2 ? v.z : v.w) * scale2;
    }
    f32x4 oacc[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) oacc[db] = f32x4{0.f, 0.f, 0.f, 0.f};
    float mrun = -__builtin_inff(), lsum = 0.f;
    __syncthreads();
    for (int t = 0; t < ntile; ++t) {
        const float* Kt = lds + (t & 1) * A_STAGE;
        const float* Vt = Kt + A_TK * A_P;
        // S^T tiles: [16 keys x 16 queries] x 2, contraction over d in steps of 4 (k = g)
        f32x4 sacc[2];
        sacc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
        sacc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            sacc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(Kt[lc * A_P + 4 * s + lg], qf[s], sacc[0], 0, 0, 0);
            sacc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(Kt[(16 + lc) * A_P + 4 * s + lg], qf[s], sacc[1], 0, 0, 0);
        }
        // lane holds the logits of keys 16 kb + 4 g + r (r = 0..3): 8 of the tile's 32 for its query; the other 24 sit in the
        // lanes q + 16 g'
        float mx = fmaxf(fmaxf(sacc[0][0], sacc[0][1]), fmaxf(sacc[0][2], sacc[0][3]));
        mx = fmaxf(mx, fmaxf(fmaxf(sacc[1][0], sacc[1][1]), fmaxf(sacc[1][2], sacc[1][3])));
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mnew = fmaxf(mrun, mx);
        const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);
        float rs = 0.f;
#pragma unroll
        for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
            for (int r = 0; r < 4; ++r) { sacc[kb2][r] = __builtin_amdgcn_exp2f(sacc[kb2][r] - mnew); rs += sacc[kb2][r]; }
        lsum = lsum * alpha + rs;                        // per-lane partial sum (its 8 keys per tile); combined over g at the end
        mrun = mnew;
        if (__any(alpha != 1.0f)) {
#pragma unroll
            for (int db = 0; db < 4; ++db) oacc[db] = oacc[db] * alpha;
        }
        // O^T[d][q] += sum_key V[key][d] P[key][q]: step (kb, r) contracts the keys 16 kb + 4 g + r (k = g)
#pragma unroll
        for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = 16 * kb2 + 4 * lg + r;
#pragma unroll
                for (int db = 0; db < 4; ++db)
                    oacc[db] = __builtin_amdgcn_mfma_f32_16x16x4f32(Vt[key * A_P + 16 * db + lc], sacc[kb2][r], oacc[db], 0, 0, 0);
            }
        __syncthreads();
    }
    float ltot = lsum + __shfl_xor(lsum, 16, 64);
    ltot = ltot + __shfl_xor(ltot, 32, 64);
    const float inv = 1.0f / ltot;
#pragma unroll
    for (int db = 0; db < 4; ++db) oacc[db] = oacc[db] * inv;
    // Y^T[o][q] = relu(b[o] + sum_d W[d][o] O^T[d][q]); lane holds d = 16 db + 4 g + r in oacc[db][r] -> step (db, r), k = g
    float* __restrict__ op = Y + ((size_t)cloud * m + (qok ? qrow : 0)) * ldy;
#pragma unroll 1
    for (int ob = 0; ob < 16; ob += 4) {
        f32x4 y[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) y[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int d = 16 * db + 4 * lg + r;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    y[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(Wl[d * A_WP + 16 * (ob + u) + lc], oacc[db][r], y[u], 0, 0, 0);
            }
        if (qok) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int o0 = 16 * (ob + u) + 4 * lg;   // lane holds o = o0 .. o0 + 3 of its query
                const float4 bb = *reinterpret_cast<const float4*>(bias + o0);
                float4 v;
                v.x = fmaxf(y[u][0] + bb.x, 0.f); v.y = fmaxf(y[u][1] + bb.y, 0.f);
                v.z = fmaxf(y[u][2] + bb.z, 0.f); v.w = fmaxf(y[u][3] + bb.w, 0.f);
                *reinterpret_cast<float4*>(op + o0) = v;
            }
        }
    }
}
}  // namespace lab

int main() {
    const int b = 32, m = 1024;
    std::vector<float> h((size_t)b * m * 320), w(64 * 256 + 256);
    unsigned s = 3;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; }
    for (auto& v : w) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; }
    float *x, *W, *out, *out2;
    (void)hipMalloc(&x, h.size() * 4); (void)hipMalloc(&W, w.size() * 4);
    (void)hipMalloc(&out, (size_t)b * m * 256 * 4); (void)hipMalloc(&out2, (size_t)b * m * 256 * 4);
    (void)hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(W, w.data(), w.size() * 4, hipMemcpyHostToDevice);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lab::attention16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lab::A_BYTES);
    auto run_ref = [&]() { return dispu_attention_project(b, m, m, 64, x + 128, 320, x, 320, x + 64, 320, 2.0f, W, W + 64 * 256, 256, out, 256, nullptr); };
    auto run_new = [&]() {
        hipLaunchKernelGGL(lab::attention16_kernel, dim3(m / 128, b), dim3(768), lab::A_BYTES, 0, m, m, x + 128, 320L, x, 320L, x + 64, 320L, 2.0f, out2,
                           256L, W, W + 64 * 256);
    };
    run_ref(); run_new();
    (void)hipDeviceSynchronize();
    std::vector<float> a((size_t)b * m * 256), c(a.size());
    (void)hipMemcpy(a.data(), out, a.size() * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(c.data(), out2, c.size() * 4, hipMemcpyDeviceToHost);
    double md = 0, mv = 0;
    for (size_t i = 0; i < a.size(); ++i) { md = std::fmax(md, std::fabs((double)a[i] - c[i])); mv = std::fmax(mv, std::fabs((double)a[i])); }
    printf("max |16x16 - 32x32| = %.3g   (max |value| %.3g)\n", md, mv);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int which = 0; which < 2; ++which) {
        for (int i = 0; i < 3; ++i) { if (which) run_new(); else run_ref(); }
        (void)hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) { if (which) run_new(); else run_ref(); }
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.1f us per call\n", which ? "16x16x4, 8 MFMA waves" : "32x32x2, 4 MFMA waves (production)", ms * 100);
    }
    return 0;
}
