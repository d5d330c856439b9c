// EXPERIMENT, NOT BUILT INTO libdispu_hip.so (round 2).  Exact FPS with Morton-sorted buckets whose bounding-box test skips
// provably unchanged points.  Index-exact against the oracle in its unrolled form (all tie / duplicate / grid cases), but the
// per-round dependent chain (bound test -> per-bucket refresh with a cross-lane max -> wave candidate -> LDS -> barrier -> winner)
// costs 1.4 - 2.4 us per round on one CU, no better than the dense register kernel's 2.1 us (measured, MI355X):
//   (1, 24576, 8192)  dense 17.4 ms | buckets, if-chain per slot: 15.0 ms (sphere) / 19.3 ms (cube) | jump per set bit: 28.9 / 38.6 ms
// The per-round floor of the reduction alone is ~0.44 us (4 waves, (32,1024,384)); see DESIGN.md section 11.
// To compile it again: copy next to csrc/common.h and add the dispatch hook in sampling.hip (git history of round 2).
// Exact farthest point sampling for LARGE clouds (4096 < n <= 24576: the whole-cloud test path samples 8192 of 24576
// merged points, DisPU/model.py:375; tf_sampling_g.cu:105-170 is the spec) with bucket skipping.
//
// fps_reg_kernel (sampling.hip) evaluates all n distances every round on ONE CU - at n = 24576 that is 1.7 us of VALU work
// per dependent round, 17.4 ms for 8192 samples.  Spreading a cloud over several CUs needs a chip-wide arg-max per round
// (agent-scope atomics across XCDs: slower than the round itself).  Instead the work per round is cut, exactly:
//
//   * a pre-pass sorts the cloud's points by the Morton code of a 16^3 cell grid (counting sort, one workgroup per cloud), so
//     that 64 consecutive points - the 64 lanes of one register slot of one wave = a BUCKET - are spatially compact;
//   * every bucket keeps its bounding box and a record of its current farthest candidate (max running distance, tie key and
//     coordinates of that point) in the registers of lane b of its wave;
//   * in a round, lane b tests its bucket against the new sample s: if the squared distance from s to the box (shrunk by
//     1e-5 relative, which covers fp32 rounding of both sides) exceeds the bucket's max running distance, then
//     min(td, d(p, s)) == td for every point p of the bucket - nothing in the bucket changes and it is skipped.  Only flagged
//     buckets evaluate their 64 distances and refresh their record;
//   * the round's winner is the max over the bucket records: per wave over its P lanes, then over the 8 waves through LDS
//     (one barrier per round, double-buffered slots).  The winner's coordinates travel with the record - no global load.
//
// After the first few dozen samples a new sample only reaches a handful of buckets, so a round costs the bound tests and
// two small reductions instead of n distance updates.  Results are IDENTICAL to the dense kernel and to the oracle: skipped
// updates are provably no-ops, and ties are decided by the reference's rule on the ORIGINAL index (64-bit key order:
// distance bits, then lowest k mod 512, then lowest k), independent of how the sort arranged the points.
#include "../../dis-pu_amd/csrc/common.h"

namespace dispu {

__device__ __forceinline__ uint32_t fpsb_tiekey(int k) {      // same key as sampling.hip:fps_tiekey
    return 0xFFFFFFFFu - ((((uint32_t)k & 511u) << 22) | ((uint32_t)k >> 9));
}
__device__ __forceinline__ int fpsb_key_to_index(uint32_t tk) {
    const uint32_t t = 0xFFFFFFFFu - tk;
    return (int)(((t & 0x3FFFFFu) << 9) | (t >> 22));
}

template <int CTRL, int RM = 0xF>
__device__ __forceinline__ float dpp_max_f32_step(float v) {
    return fmaxf(v, __uint_as_float(dpp_u32<CTRL, RM>(0xFF800000u, __float_as_uint(v))));
}
__device__ __forceinline__ float wave_max_f32(float v) {       // wave-uniform result
    v = dpp_max_f32_step<DPP_ROW_SHR1>(v);
    v = dpp_max_f32_step<DPP_ROW_SHR2>(v);
    v = dpp_max_f32_step<DPP_ROW_SHR4>(v);
    v = dpp_max_f32_step<DPP_ROW_SHR8>(v);
    v = dpp_max_f32_step<DPP_ROW_BCAST15, 0xA>(v);
    v = dpp_max_f32_step<DPP_ROW_BCAST31, 0xC>(v);
    return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), 63));
}
template <int CTRL, int RM = 0xF>
__device__ __forceinline__ uint32_t dpp_max_u32_step(uint32_t v) {
    const uint32_t o = dpp_u32<CTRL, RM>(0u, v);
    return v > o ? v : o;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    v = dpp_max_u32_step<DPP_ROW_SHR1>(v);
    v = dpp_max_u32_step<DPP_ROW_SHR2>(v);
    v = dpp_max_u32_step<DPP_ROW_SHR4>(v);
    v = dpp_max_u32_step<DPP_ROW_SHR8>(v);
    v = dpp_max_u32_step<DPP_ROW_BCAST15, 0xA>(v);
    v = dpp_max_u32_step<DPP_ROW_BCAST31, 0xC>(v);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ float wave_all_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_all_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ unsigned spread4(unsigned v) {      // abcd -> a00b00c00d
    return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4) | ((v & 8u) << 6);
}

// ---- pre-pass: perm[cloud][0..n) = point indices ordered by the Morton code of their cell in a 16^3 grid over the cloud's
// bounding box.  The order inside a cell comes from LDS atomics and is not reproducible; it only decides which points share
// a bucket (speed), never a result.
__global__ __launch_bounds__(1024) void fps_cellsort_kernel(int n, const float* __restrict__ xyz, int* __restrict__ perm) {
    __shared__ float bb[6][16];
    __shared__ unsigned cnt[4096];
    __shared__ unsigned wsum[16];
    const int cloud = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* __restrict__ p = xyz + (size_t)cloud * n * 3;
    int* __restrict__ pm = perm + (size_t)cloud * n;
    float mn[3] = {3e38f, 3e38f, 3e38f}, mx[3] = {-3e38f, -3e38f, -3e38f};
    for (int k = tid; k < n; k += 1024)
#pragma unroll
        for (int a = 0; a < 3; ++a) { const float v = p[k * 3 + a]; mn[a] = fminf(mn[a], v); mx[a] = fmaxf(mx[a], v); }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float lo = wave_all_min(mn[a]), hi = wave_all_max(mx[a]);
        if (lane == 0) { bb[a][wave] = lo; bb[3 + a][wave] = hi; }
    }
    for (int e = tid; e < 4096; e += 1024) cnt[e] = 0u;
    __syncthreads();
    float lo[3], sc[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float l = bb[a][0], h = bb[3 + a][0];
        for (int w = 1; w < 16; ++w) { l = fminf(l, bb[a][w]); h = fmaxf(h, bb[3 + a][w]); }
        lo[a] = l;
        sc[a] = (h > l) ? 16.0f / (h - l) : 0.f;
    }
    auto cell = [&](int k) -> unsigned {
        unsigned c[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int q = (int)((p[k * 3 + a] - lo[a]) * sc[a]);
            c[a] = (unsigned)min(15, max(0, q));
        }
        return spread4(c[0]) | (spread4(c[1]) << 1) | (spread4(c[2]) << 2);
    };
    for (int k = tid; k < n; k += 1024) atomicAdd(&cnt[cell(k)], 1u);
    __syncthreads();
    // exclusive prefix over the 4096 cells: thread t owns cells 4t .. 4t + 3
    unsigned c0 = cnt[4 * tid], c1 = cnt[4 * tid + 1], c2 = cnt[4 * tid + 2], c3 = cnt[4 * tid + 3];
    unsigned tot = c0 + c1 + c2 + c3, inc = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned u = __shfl_up(inc, o, 64);
        if (lane >= o) inc += u;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    unsigned base = 0;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    base += inc - tot;
    __syncthreads();
    cnt[4 * tid] = base; cnt[4 * tid + 1] = base + c0; cnt[4 * tid + 2] = base + c0 + c1; cnt[4 * tid + 3] = base + c0 + c1 + c2;
    __syncthreads();
    for (int k = tid; k < n; k += 1024) pm[atomicAdd(&cnt[cell(k)], 1u)] = k;
}

constexpr int FB_BS = 512, FB_W = FB_BS / kWave;

// P register slots per lane; bucket (wave, i) = the 64 points at sorted positions wave * 64 P + 64 i + lane.
template <int P, bool FMA>
__global__ __launch_bounds__(FB_BS) void fps_bucket_kernel(int n, int m, const float* __restrict__ xyz, const int* __restrict__ perm,
                                                           int* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* perm_l = reinterpret_cast<int*>(smem);                               // [FB_BS * P] original index of a sorted position
    float* box = reinterpret_cast<float*>(perm_l + FB_BS * P);                // [FB_W * P][8]: min xyz, max xyz (+2 pad)
    float* slot = box + FB_W * P * 8;                                         // [2][FB_W][8]: d, tie key, x, y, z
    const int cloud = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* __restrict__ p = xyz + (size_t)cloud * n * 3;
    const int* __restrict__ pm = perm + (size_t)cloud * n;
    int* __restrict__ o = out + (size_t)cloud * m;
    const int wbase = wave * 64 * P;

    float x[P], y[P], z[P], td[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int pos = wbase + i * 64 + lane;
        const bool ok = pos < n;
        const int k = ok ? pm[pos] : 0;
        perm_l[pos] = k;
        x[i] = ok ? p[k * 3 + 0] : 0.f;
        y[i] = ok ? p[k * 3 + 1] : 0.f;
        z[i] = ok ? p[k * 3 + 2] : 0.f;
        td[i] = ok ? 1e38f : -1.0f;                      // -1: "no point here", never the farthest
        const float lx = wave_all_min(ok ? x[i] : 3e38f), hx = wave_all_max(ok ? x[i] : -3e38f);
        const float ly = wave_all_min(ok ? y[i] : 3e38f), hy = wave_all_max(ok ? y[i] : -3e38f);
        const float lz = wave_all_min(ok ? z[i] : 3e38f), hz = wave_all_max(ok ? z[i] : -3e38f);
        if (lane == 0) {
            float* bx = box + (size_t)(wave * P + i) * 8;
            bx[0] = lx; bx[1] = ly; bx[2] = lz; bx[3] = 0.f; bx[4] = hx; bx[5] = hy; bx[6] = hz; bx[7] = 0.f;
        }
    }
    // record of bucket `lane` of this wave (lanes < P): max running distance, sorted position and coordinates of that point
    float bmax = -1.0f, bwx = 0.f, bwy = 0.f, bwz = 0.f;
    int bpos = 0;
    if (lane < P && wbase + lane * 64 < n) bmax = 1e38f;   // non-empty bucket: flagged in the first round
    __syncthreads();
    float4 blo = make_float4(0.f, 0.f, 0.f, 0.f), bhi = blo;
    if (lane < P) {
        blo = *reinterpret_cast<const float4*>(box + (size_t)(wave * P + lane) * 8);
        bhi = *reinterpret_cast<const float4*>(box + (size_t)(wave * P + lane) * 8 + 4);
    }
    if (tid == 0) o[0] = 0;
    float x1 = p[0], y1 = p[1], z1 = p[2];               // sample 0 is point 0 (tf_sampling_g.cu:122-124)

    // refresh of bucket I (a compile-time register slot): 64 distances, running minima, new record
#define FPSB_REFRESH(I)                                                                                                   \
    case I: if constexpr ((I) < P) {                                                                                      \
        const float d = sqdist3<FMA>(x[I] - x1, y[I] - y1, z[I] - z1);                                                    \
        const float t = fminf(d, td[I]);                                                                                  \
        td[I] = t;                                                                                                        \
        const float bm = wave_max_f32(t);                                                                                 \
        unsigned long long tm = __ballot(t == bm);                                                                        \
        int wl = (int)__builtin_ctzll(tm);                                                                                \
        if (tm & (tm - 1)) {          /* several lanes share the maximum: the reference's tie rule decides */           \
            const bool in = t == bm;                                                                                      \
            const uint32_t key = in ? fpsb_tiekey(perm_l[wbase + I * 64 + lane]) : 0u;                                    \
            const uint32_t mk = wave_max_u32(key);                                                                        \
            wl = (int)__builtin_ctzll(__ballot(in && key == mk));                                                         \
        }                                                                                                                 \
        const float cx = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(x[I]), wl));           \
        const float cy = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(y[I]), wl));           \
        const float cz = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(z[I]), wl));           \
        if (lane == I) { bmax = bm; bpos = wbase + I * 64 + wl; bwx = cx; bwy = cy; bwz = cz; }                           \
    } break;
#define FPSB_R4(I) FPSB_REFRESH(I) FPSB_REFRESH(I + 1) FPSB_REFRESH(I + 2) FPSB_REFRESH(I + 3)
#define FPSB_R16(I) FPSB_R4(I) FPSB_R4(I + 4) FPSB_R4(I + 8) FPSB_R4(I + 12)

    for (int j = 1; j < m; ++j) {
        // ---- which buckets can change?
        const float ex = fmaxf(fmaxf(blo.x - x1, x1 - bhi.x), 0.f);
        const float ey = fmaxf(fmaxf(blo.y - y1, y1 - bhi.y), 0.f);
        const float ez = fmaxf(fmaxf(blo.z - z1, z1 - bhi.z), 0.f);
        const float lb = __builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex));
        unsigned long long mask = __ballot(lane < P && lb * 0.99999f <= bmax);
        // ---- refresh the flagged buckets: a jump per SET bit (an if per slot costs more than the refreshes once few are flagged)
        while (mask) {
            const int i = (int)__builtin_ctzll(mask);
            mask &= mask - 1;
            switch (i) {
                FPSB_R16(0) FPSB_R16(16) FPSB_R16(32)       // slots >= P compile to nothing
                default: break;
            }
        }
        // ---- this wave's candidate = best record among its lanes < P
        const int par = j & 1;
        {
            const float v = (lane < P) ? bmax : -2.0f;
            const float wm = wave_max_f32(v);
            unsigned long long tm = __ballot(v == wm);
            int bl = (int)__builtin_ctzll(tm);
            if (tm & (tm - 1)) {
                const bool in = v == wm;
                const uint32_t key = in ? fpsb_tiekey(perm_l[bpos]) : 0u;
                const uint32_t mk = wave_max_u32(key);
                bl = (int)__builtin_ctzll(__ballot(in && key == mk));
            }
            if (lane == bl) {
                float* s = slot + (size_t)(par * FB_W + wave) * 8;
                *reinterpret_cast<float4*>(s) = make_float4(bmax, __int_as_float(bpos), bwx, bwy);
                s[4] = bwz;
            }
        }
        __syncthreads();
        // ---- winner among the waves (every wave computes it: lanes < FB_W read one slot each)
        {
            float sd = -2.0f, sx = 0.f, sy = 0.f, sz = 0.f;
            int spos = 0;
            if (lane < FB_W) {
                const float* s = slot + (size_t)(par * FB_W + lane) * 8;
                const float4 a = *reinterpret_cast<const float4*>(s);
                sd = a.x; spos = __float_as_int(a.y); sx = a.z; sy = a.w; sz = s[4];
            }
            const float gm = wave_max_f32(sd);
            unsigned long long tm = __ballot(sd == gm);
            int gl = (int)__builtin_ctzll(tm);
            if (tm & (tm - 1)) {
                const bool in = sd == gm;
                const uint32_t key = in ? fpsb_tiekey(perm_l[spos]) : 0u;
                const uint32_t mk = wave_max_u32(key);
                gl = (int)__builtin_ctzll(__ballot(in && key == mk));
            }
            x1 = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(sx), gl));
            y1 = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(sy), gl));
            z1 = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(sz), gl));
            if (tid == 0) o[j] = perm_l[__builtin_amdgcn_readlane(spos, gl)];
        }
    }
#undef FPSB_R16
#undef FPSB_R4
#undef FPSB_REFRESH
}

template <int P>
static int launch_fps_bucket(int b, int n, int m, const float* xyz, int* perm, int* out, int arith, hipStream_t s) {
    const size_t bytes = (size_t)FB_BS * P * 4 + (size_t)FB_W * P * 8 * 4 + 2 * FB_W * 8 * 4;
    hipLaunchKernelGGL(fps_cellsort_kernel, dim3(b), dim3(1024), 0, s, n, xyz, perm);
    DISPU_CHECK_LAUNCH();
    if ((arith & DISPU_ARITH_CONTRACT)) {
        static bool attr = false;
        if (!attr) {
            DISPU_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fps_bucket_kernel<P, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
            attr = true;
        }
        hipLaunchKernelGGL((fps_bucket_kernel<P, true>), dim3(b), dim3(FB_BS), bytes, s, n, m, xyz, perm, out);
    } else {
        static bool attr = false;
        if (!attr) {
            DISPU_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fps_bucket_kernel<P, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
            attr = true;
        }
        hipLaunchKernelGGL((fps_bucket_kernel<P, false>), dim3(b), dim3(FB_BS), bytes, s, n, m, xyz, perm, out);
    }
    return (int)hipGetLastError();
}

// -1: shape outside this path
int fps_bucket_dispatch(int b, int n, int m, const float* xyz, void* temp, int* out, int arith, hipStream_t s) {
    if (!temp || n <= 4096 || n > FB_BS * 48 || m < 64) return -1;
    int* perm = reinterpret_cast<int*>(temp);
    if (n <= FB_BS * 16) return launch_fps_bucket<16>(b, n, m, xyz, perm, out, arith, s);
    if (n <= FB_BS * 24) return launch_fps_bucket<24>(b, n, m, xyz, perm, out, arith, s);
    return launch_fps_bucket<48>(b, n, m, xyz, perm, out, arith, s);
}

bool fps_bucket_wants_scratch(int n, int m) { return n > 4096 && n <= FB_BS * 48 && m >= 64; }

}  // namespace dispu
