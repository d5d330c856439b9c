// PointNet++ set-abstraction hot loop in ONE kernel (Common/pointnet_util.py:91-149 with pooling='max', mlp2=None):
//
//     group_point(xyz | points, idx) -> grouped_xyz -= new_xyz -> concat -> conv2d x nl (bias, BatchNorm fold, ReLU) -> max over nsample
//
// The reference (and dis-pu_amd/pointnet_util.py's composition of the single ops) writes the grouped tensor [b, m, ns, 3 + c]
// and every layer's [b, m, ns, C] output to HBM and reads it back: at the shapes Common/ops.py:505-550 uses
// (hierachy_feature_extractor: ns = 64, m = 1024 / 384 / 128, MLPs up to 256 wide) that is 0.4 GB per 4 clouds for 7 MB of
// results.  Here one workgroup owns ONE centre: its 64 x (3 + c) grouped rows are gathered straight into LDS, the layers
// run LDS -> v_mfma_f32_32x32x2_f32 -> LDS (weights stream from L2 as the MFMA B operand, every workgroup reads the same
// few hundred KB), and the last layer's 64 x C_out tile is max-reduced in registers: only [b, m, C_out] reaches HBM.
//
// Arithmetic = the unfused path's, bit for bit: a 32x32x2 MFMA chain over ascending k IS the ascending-k fmaf chain of
// dispu_linear (tests/test_linear_matches_chain_exactly), then one rounded bias add, v * scale + shift as two rounded
// operations (-ffp-contract=off), ReLU; max is order-free.  Zero padding of an odd K adds fmaf(0, 0, acc) = acc.
#include "common.h"

namespace dispu {

typedef float sa_f32x16 __attribute__((ext_vector_type(16)));

constexpr int SA_MAXL = 3;

struct SaArgs {
    int mode;                            // 0: set abstraction rows [xyz_j - centre | points_j]; 1: EdgeConv rows [F_i | F_j - F_i]
    int gk;                              // rows pooled together: NS (mode 0), k in {16, 32, 64} (mode 1: NS / k points per workgroup)
    long total;                          // mode 1: number of points b * n (the last workgroup may be partial)
    long ldf; int ldi;                   // mode 1: row stride of the feature matrix / of the neighbour table
    int act_last;                        // ReLU on the last layer (hidden layers always have it)
    int n, m, c, nl;                     // dataset points / centres per cloud, feature channels of `points` (0: xyz only), layers
    const float* xyz;                    // [b, n, 3]
    const float* new_xyz;                // [b, m, 3]
    const float* points;                 // [b, n, c] or null
    const int* idx;                      // [b, m, NS]
    const float* W[SA_MAXL];             // [cin_l, cout_l] row-major
    const float* bias[SA_MAXL];
    const float* scale[SA_MAXL];         // BatchNorm fold (null: none)
    const float* shift[SA_MAXL];
    int cout[SA_MAXL];
    float* out;                          // [b, m, cout[nl - 1]]
    int pitch;                           // floats per LDS row (odd)
};

// One layer on the workgroup's NS rows: in [NS][pitch] (K columns, zero-padded to even) -> act(bn(in . W + b)).
// Tasks = (row tile of 32, column tile of 32) dealt round-robin to the 4 waves.  LAST: no store, per-column maxima of every 16-row
// half tile go to red[NS / 16][cout] (the caller combines gk / 16 of them per pooled group).
template <int NS, bool LAST>
__device__ __forceinline__ void sa_layer(const float* __restrict__ in, float* __restrict__ outb, int pitch, int K, int cout,
                                         const float* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ scale,
                                         const float* __restrict__ shift, float* __restrict__ red, int wave, int lane, bool relu_last) {
    constexpr int RT = NS / 32;
    const int li = lane & 31, kh = lane >> 5;
    const int ct_n = (cout + 31) >> 5, ntask = RT * ct_n;
    const int ksteps = (K + 1) >> 1;
    for (int task = wave; task < ntask; task += 4) {
        const int rt = task % RT, ct = task / RT;
        const int col = ct * 32 + li;
        const bool cok = col < cout;
        const float* __restrict__ arow = in + (rt * 32 + li) * pitch + kh;
        const float* __restrict__ wcol = W + (size_t)kh * cout + (cok ? col : 0);
        sa_f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        constexpr int U = 8;
        for (int s0 = 0; s0 < ksteps; s0 += U) {
            float bv[U], av[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = 2 * (s0 + u) + kh;
                bv[u] = (cok && k < K) ? wcol[(size_t)2 * (s0 + u) * cout] : 0.f;
                av[u] = (s0 + u < ksteps) ? arow[2 * (s0 + u)] : 0.f;          // columns K .. 2 ksteps - 1 of `in` hold zeros
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (s0 + u < ksteps) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
        }
        const float bb = cok ? bias[col] : 0.f;
        const float sc = (cok && scale) ? scale[col] : 1.f, sh = (cok && scale) ? shift[col] : 0.f;
        float mlo = -__builtin_inff(), mhi = -__builtin_inff();      // maxima over the tile's rows 0..15 / 16..31 (registers r < 8 / r >= 8)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = acc[r] + bb;
            if (scale) { v = v * sc; v = v + sh; }
            if (!LAST || relu_last) v = fmaxf(v, 0.f);
            if constexpr (LAST) { if (r < 8) mlo = fmaxf(mlo, v); else mhi = fmaxf(mhi, v); }
            else if (cok) outb[(rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh) * pitch + col] = v;
        }
        if constexpr (LAST) {
            mlo = fmaxf(mlo, __shfl_xor(mlo, 32, 64));
            mhi = fmaxf(mhi, __shfl_xor(mhi, 32, 64));
            if (kh == 0 && cok) { red[(2 * rt) * cout + col] = mlo; red[(2 * rt + 1) * cout + col] = mhi; }
        }
    }
}

template <int NS>
__global__ __launch_bounds__(256) void sa_fused_kernel(SaArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sa_lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long blk = xcd_block(blockIdx.x, gridDim.x);               // XCD-aware: one L2 serves a contiguous range of centres (= few clouds)
    float* bufA = sa_lds;
    float* bufB = sa_lds + NS * a.pitch;
    float* red = bufB + NS * a.pitch;                                  // [NS / 16][cout_last]
    const int K0 = a.mode == 0 ? 3 + a.c : 2 * a.c, K0p = (K0 + 1) & ~1;

    if (a.mode == 0) {
        // ---- set abstraction: row s = [xyz[idx[s]] - centre | points[idx[s]]], zero-padded to an even width
        const long cloud = blk / a.m;
        const int* __restrict__ ip = a.idx + blk * NS;
        const float* __restrict__ xb = a.xyz + cloud * a.n * 3;
        const float* __restrict__ pb = a.points ? a.points + cloud * (long)a.n * a.c : nullptr;
        const float cx = a.new_xyz[blk * 3], cy = a.new_xyz[blk * 3 + 1], cz = a.new_xyz[blk * 3 + 2];
        for (int s = wave; s < NS; s += 4) {
            const int j = ip[s];
            float* row = bufA + s * a.pitch;
            for (int ch = lane; ch < K0p; ch += 64) {
                float v = 0.f;
                if (ch < 3) v = xb[(long)j * 3 + ch] - (ch == 0 ? cx : ch == 1 ? cy : cz);
                else if (ch < K0) v = pb[(long)j * a.c + (ch - 3)];
                row[ch] = v;
            }
        }
    } else {
        // ---- EdgeConv (tf_util.get_edge_feature): row (p, s) = [F[i] | F[j] - F[i]], i = first point + p, j = idx[i][s] in i's cloud
        const int ppw = NS / a.gk;
        for (int s = wave; s < NS; s += 4) {
            long i = blk * ppw + s / a.gk;
            if (i >= a.total) i = a.total - 1;                       // partial last workgroup: computed, not stored
            const long j = (i / a.n) * a.n + a.idx[i * a.ldi + (s % a.gk)];
            const float* __restrict__ fi = a.points + i * a.ldf;
            const float* __restrict__ fj = a.points + j * a.ldf;
            float* row = bufA + s * a.pitch;
            for (int ch = lane; ch < K0p; ch += 64) {
                float v = 0.f;
                if (ch < a.c) v = fi[ch];
                else if (ch < K0) v = fj[ch - a.c] - fi[ch - a.c];
                row[ch] = v;
            }
        }
    }
    __syncthreads();

    const float* in = bufA;
    float* outb = bufB;
    int K = K0;
    for (int l = 0; l < a.nl; ++l) {
        const int cout = a.cout[l];
        if (l + 1 < a.nl) {
            sa_layer<NS, false>(in, outb, a.pitch, K, cout, a.W[l], a.bias[l], a.scale[l], a.shift[l], red, wave, lane, true);
            if (cout & 1) {                                            // the next layer reads an even number of columns
                for (int s = threadIdx.x; s < NS; s += 256) outb[s * a.pitch + cout] = 0.f;
            }
        } else {
            sa_layer<NS, true>(in, outb, a.pitch, K, cout, a.W[l], a.bias[l], a.scale[l], a.shift[l], red, wave, lane, a.act_last != 0);
        }
        __syncthreads();
        const float* t = in; in = outb; outb = const_cast<float*>(t);
        K = cout;
    }
    // pooled groups: gk / 16 consecutive half tiles each
    const int co = a.cout[a.nl - 1], halves = a.gk / 16, groups = NS / a.gk;
    for (int e = threadIdx.x; e < groups * co; e += 256) {
        const int g = e / co, ch = e - g * co;
        const long orow = blk * groups + g;
        if (a.mode == 1 && orow >= a.total) continue;
        float v = red[(g * halves) * co + ch];
        for (int h = 1; h < halves; ++h) v = fmaxf(v, red[(g * halves + h) * co + ch]);
        a.out[orow * co + ch] = v;
    }
}

}  // namespace dispu

using namespace dispu;

// Fused set abstraction (pointnet_sa_module with pooling 'max', no mlp2, use_xyz, inference BatchNorm), Common/pointnet_util.py:91-149:
//   out[b, m, cout[nl-1]] = max_s mlp([xyz[idx[b,m,s]] - new_xyz[b,m] | points[idx[b,m,s]]])
// idx [b, m, ns] from dispu_query_ball / dispu_knn_point; ns in {32, 64}; nl <= 3 layers, W[l] [cin_l, cout_l] row-major with
// cin_0 = 3 + c, cin_l = cout[l-1]; scale / shift (BatchNorm fold) may be NULL per layer.  Bit-identical to the chain
// dispu_group_point -> dispu_group_center -> dispu_linear_bn x nl -> dispu_pool_nsample(max).
DISPU_EXPORT int dispu_sa_fused(int b, int n, int m, int ns, int c, const float* xyz, const float* new_xyz, const float* points,
                                const int* idx, int nl, const float* const* W, const float* const* bias, const float* const* scale,
                                const float* const* shift, const int* cout, float* out, void* stream) {
    if (b < 0 || n <= 0 || m <= 0 || (ns != 32 && ns != 64) || c < 0 || nl < 1 || nl > SA_MAXL || !xyz || !new_xyz || !idx || !W ||
        !bias || !cout || !out || (c > 0 && !points))
        return (int)hipErrorInvalidValue;
    if (b == 0) return 0;
    SaArgs a{};
    a.mode = 0; a.gk = ns; a.act_last = 1;
    a.n = n; a.m = m; a.c = c; a.nl = nl;
    a.xyz = xyz; a.new_xyz = new_xyz; a.points = c > 0 ? points : nullptr; a.idx = idx; a.out = out;
    int width = (3 + c + 1) & ~1;
    for (int l = 0; l < nl; ++l) {
        if (cout[l] <= 0 || !W[l] || !bias[l]) return (int)hipErrorInvalidValue;
        a.W[l] = W[l]; a.bias[l] = bias[l];
        a.scale[l] = (scale && shift && scale[l] && shift[l]) ? scale[l] : nullptr;
        a.shift[l] = a.scale[l] ? shift[l] : nullptr;
        a.cout[l] = cout[l];
        if (l + 1 < nl) width = width > ((cout[l] + 1) & ~1) ? width : ((cout[l] + 1) & ~1);
    }
    a.pitch = width | 1;
    const size_t bytes = ((size_t)2 * ns * a.pitch + (size_t)(ns / 16) * cout[nl - 1]) * sizeof(float);
    if (bytes > 160 * 1024) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)((long)b * m));
    // up to 160 KB of dynamic LDS (the widest layer of ops.py:505-550: 64 x 133 floats x 2 buffers = 68 KB): opt in once per device
    static DevOnce once64, once32;
    if (ns == 64) {
        if (once64.needed()) {
            DISPU_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(sa_fused_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            once64.done();
        }
        hipLaunchKernelGGL((sa_fused_kernel<64>), grid, dim3(256), bytes, s, a);
    } else {
        if (once32.needed()) {
            DISPU_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(sa_fused_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            once32.done();
        }
        hipLaunchKernelGGL((sa_fused_kernel<32>), grid, dim3(256), bytes, s, a);
    }
    return (int)hipGetLastError();
}

// Fused EdgeConv (gcn_lib/tf_vertex.py:81-101 with tf_util.get_edge_feature, Common/tf_util.py:654-686): per point i
//   out[i, :] = max_{s < k} mlp([F[i] | F[idx[i, s]] - F[i]])        nl <= 3 conv2d layers (bias, optional BatchNorm fold, ReLU;
// the LAST layer's ReLU only when act_last), k in {16, 32, 64}; feat [b*n, c] with row stride ldf, idx [b*n, >= k] (row stride ldi,
// cloud-relative neighbour ids).  The [b, n, k, 2c] edge tensor and the [b, n, k, C] layer outputs never reach HBM.  Bit-identical to
// dispu_edge_feature -> dispu_linear_bn x nl -> dispu_pool_nsample(max).
DISPU_EXPORT int dispu_edge_conv_fused(int b, int n, int k, int c, const float* feat, long ldf, const int* idx, int ldi, int nl,
                                       const float* const* W, const float* const* bias, const float* const* scale,
                                       const float* const* shift, const int* cout, int act_last, float* out, void* stream) {
    if (b < 0 || n <= 0 || (k != 16 && k != 32 && k != 64) || c <= 0 || nl < 1 || nl > SA_MAXL || !feat || !idx || !W || !bias || !cout ||
        !out || ldf < c || ldi < k)
        return (int)hipErrorInvalidValue;
    if (b == 0) return 0;
    SaArgs a{};
    a.mode = 1; a.gk = k; a.total = (long)b * n; a.ldf = ldf; a.ldi = ldi; a.act_last = act_last;
    a.n = n; a.m = n; a.c = c; a.nl = nl;
    a.points = feat; a.idx = idx; a.out = out;
    int width = (2 * c + 1) & ~1;
    for (int l = 0; l < nl; ++l) {
        if (cout[l] <= 0 || !W[l] || !bias[l]) return (int)hipErrorInvalidValue;
        a.W[l] = W[l]; a.bias[l] = bias[l];
        a.scale[l] = (scale && shift && scale[l] && shift[l]) ? scale[l] : nullptr;
        a.shift[l] = a.scale[l] ? shift[l] : nullptr;
        a.cout[l] = cout[l];
        if (l + 1 < nl) width = width > ((cout[l] + 1) & ~1) ? width : ((cout[l] + 1) & ~1);
    }
    a.pitch = width | 1;
    constexpr int ns = 64;
    const size_t bytes = ((size_t)2 * ns * a.pitch + (size_t)(ns / 16) * cout[nl - 1]) * sizeof(float);
    if (bytes > 160 * 1024) return (int)hipErrorInvalidValue;
    static DevOnce once;
    if (once.needed()) {
        DISPU_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(sa_fused_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        once.done();
    }
    const long ppw = ns / k;
    hipLaunchKernelGGL((sa_fused_kernel<64>), dim3((unsigned)((a.total + ppw - 1) / ppw)), dim3(256), bytes, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}
