// Exact farthest point sampling for LARGE clouds (4096 < n <= 24576, m >= 64: the whole-cloud test path samples 8192 of the
// 24576 merged points, DisPU/model.py:375; tf_sampling_g.cu:105-170 is the spec) with WAVE-level skipping.
//
// fps_reg_kernel (sampling.hip) updates all n running distances every round on the one CU that owns the cloud: at n = 24576
// that is 16 waves x 24 points per lane = 1.75 us of VALU work per dependent round on 4 SIMDs (17.4 ms for 8192 samples).
// Cross-workgroup rounds cost >= 0.8 us per exchange (tools/micro/xwg_sync.hip), bucket-level skipping drowns in its own
// dependent chains (tools/micro/fps_bucket_lab.hip).  Here the dense kernel is kept and whole WAVES are skipped:
//   * a pre-pass orders the cloud by the Morton code of a 16^3 cell grid, so the 64 P points of one wave are one compact
//     region, and - inside every wave's range - by the reference's tie priority (k mod 512, then k), so a lane that visits its
//     P slots in order with a strict '>' keeps the reference's winner among equal distances;
//   * a wave keeps its region's bounding box, the largest running distance of its points and its current candidate
//     (distance, position, coordinates).  If the squared distance from the new sample to the box (shrunk by 1e-5 relative:
//     covers the fp32 rounding of both sides) exceeds that largest running distance, min(td, d) == td for every point of
//     the wave: nothing changes, the cached candidate is still the wave's answer, and the wave goes straight to the barrier;
//   * the other waves do the dense update exactly as before.  After the first few dozen samples a new sample reaches a few
//     of the 16 regions, and consecutive Morton ranges sit on different SIMDs (wave w -> SIMD w mod 4).
// Results are IDENTICAL to the dense kernel: skipped updates are provably no-ops and ties are decided by the original index.
// Measured (MI355X, tools/fps_bench.py): (8, 24576, 8192) 17.4 -> 9.6 ms (sphere) / 10.3 ms (cube), with four regions per wave
// (fps_wave4_kernel below) 8.5 / 9.5 ms; (8, 8192, 2048) 2.43 -> 1.5 ms;
// (32, 4097, 1024) 1.22 -> 0.75 ms.  tools/micro/fps_wave_prof.hip splits a round (2900 cycles at n = 24576): dense update of
// the busiest wave 1330, its arg-max 575, slot + barrier 120, cross-wave winner 460 - 1100 (16 waves, oldest first).
#include "common.h"

#include <cstdlib>

#ifndef FPSW_TICK            // tools/micro/fps_wave_prof.hip includes this file with cycle-counter hooks; none in the library
#define FPSW_TICK(i)
#define FPSW_PROF_BEGIN
#define FPSW_PROF_ACTIVE
#define FPSW_PROF_GROUPS(mask)
#define FPSW_PROF_END
#endif

namespace dispu {

__device__ __forceinline__ uint32_t fpsw_tiekey(int k) {      // same key as sampling.hip:fps_tiekey
    return 0xFFFFFFFFu - ((((uint32_t)k & 511u) << 22) | ((uint32_t)k >> 9));
}
__device__ __forceinline__ uint32_t fpsw_order(int k) {       // ascending = the reference's tie priority
    return (((uint32_t)k & 511u) << 22) | (uint32_t)k;
}

// Running distances are >= 0 (or -1 = "no point"): as unsigned keys (bits + 1, 0 for "no point") they order like the floats, and
// an unsigned max folds into ONE v_max_u32_dpp per step (a float max costs a DPP move plus two canonicalising v_max per step).
__device__ __forceinline__ uint32_t fpsw_dkey(float d) { return d >= 0.f ? __float_as_uint(d) + 1u : 0u; }
__device__ __forceinline__ float fpsw_dkey_value(uint32_t k) { return k ? __uint_as_float(k - 1u) : -1.0f; }
template <int CTRL, int RM = 0xF>
__device__ __forceinline__ uint32_t fpsw_umax_step(uint32_t v) {
    const uint32_t o = dpp_u32<CTRL, RM>(0u, v);
    return v > o ? v : o;
}
__device__ __forceinline__ uint32_t fpsw_wave_max_u32(uint32_t v) {
    v = fpsw_umax_step<DPP_ROW_SHR1>(v);
    v = fpsw_umax_step<DPP_ROW_SHR2>(v);
    v = fpsw_umax_step<DPP_ROW_SHR4>(v);
    v = fpsw_umax_step<DPP_ROW_SHR8>(v);
    v = fpsw_umax_step<DPP_ROW_BCAST15, 0xA>(v);
    v = fpsw_umax_step<DPP_ROW_BCAST31, 0xC>(v);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t fpsw_row0_max_u32(uint32_t v) {   // max over lanes 0..15 (row 0), wave-uniform result
    v = fpsw_umax_step<DPP_ROW_SHR1>(v);
    v = fpsw_umax_step<DPP_ROW_SHR2>(v);
    v = fpsw_umax_step<DPP_ROW_SHR4>(v);
    v = fpsw_umax_step<DPP_ROW_SHR8>(v);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 15);
}
__device__ __forceinline__ float fpsw_all_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float fpsw_all_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float fpsw_uniform(float v) {
    return __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(v)));
}
__device__ __forceinline__ unsigned fpsw_spread4(unsigned v) { return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4) | ((v & 8u) << 6); }

// ---- pre-pass, one workgroup per cloud: (1) counting sort by Morton cell into LDS, (2) every range of `chunk` = 64 P sorted
// positions (one wave of the sampling kernel) re-sorted by the reference's tie priority (bitonic, 2048 keys), written to perm.
// The order inside a cell comes from LDS atomics; it only decides which points share a wave (speed), never a result - the
// second sort makes the final permutation deterministic within a wave's range anyway.
__global__ __launch_bounds__(1024) void fps_wavesort_kernel(int n, int chunk, const float* __restrict__ xyz, int* __restrict__ perm) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* sorted = reinterpret_cast<int*>(smem);                    // [n]
    unsigned* cnt = reinterpret_cast<unsigned*>(sorted + n);       // [4096]
    unsigned* keys = cnt + 4096;                                   // [2048]
    __shared__ float bb[6][16];
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
        const float lo = fpsw_all_min(mn[a]), hi = fpsw_all_max(mx[a]);
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
        return fpsw_spread4(c[0]) | (fpsw_spread4(c[1]) << 1) | (fpsw_spread4(c[2]) << 2);
    };
    for (int k = tid; k < n; k += 1024) atomicAdd(&cnt[cell(k)], 1u);
    __syncthreads();
    const unsigned c0 = cnt[4 * tid], c1 = cnt[4 * tid + 1], c2 = cnt[4 * tid + 2], c3 = cnt[4 * tid + 3];
    const unsigned tot = c0 + c1 + c2 + c3;
    unsigned inc = tot;
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
    cnt[4 * tid] = base; cnt[4 * tid + 1] = base + c0; cnt[4 * tid + 2] = base + c0 + c1; cnt[4 * tid + 3] = base + c0 + c1 + c2;
    __syncthreads();
    for (int k = tid; k < n; k += 1024) sorted[atomicAdd(&cnt[cell(k)], 1u)] = k;
    __syncthreads();
    // (2) every range of `chunk` positions (one wave of fps_wave_kernel, or one of a wave's four groups in fps_wave4_kernel): ascending
    // tie priority.  Chunks of <= 512 positions are sorted four at a time (the network stops at width 512, the final merge of
    // every 512-block ascending); larger ones one at a time in the 2048-wide network.
    const int W = (chunk <= 512) ? 512 : 2048, per = 2048 / W;
    for (int c0p = 0; c0p < n; c0p += chunk * per) {
        for (int e = tid; e < 2048; e += 1024) {
            const int blk = e / W, r = e - blk * W, pos = c0p + blk * chunk + r;
            keys[e] = (r < chunk && pos < n) ? fpsw_order(sorted[pos]) : 0xFFFFFFFFu;
        }
        __syncthreads();
        for (int size = 2; size <= W; size <<= 1)
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                const int t = tid;                                   // 1024 pairs
                const int l = 2 * t - (t & (stride - 1)), h = l + stride;
                const bool up = (size == W) || ((l & size) == 0);
                const unsigned a = keys[l], b = keys[h];
                if ((a > b) == up) { keys[l] = b; keys[h] = a; }
                __syncthreads();
            }
        for (int e = tid; e < 2048; e += 1024) {
            const int blk = e / W, r = e - blk * W, pos = c0p + blk * chunk + r;
            if (r < chunk && pos < n) pm[pos] = (int)(keys[e] & 0x3FFFFFu);
        }
        __syncthreads();
    }
}

constexpr int FW_BS = 1024, FW_W = FW_BS / kWave;
static_assert(FW_W == 16, "the cross-wave reduction reads one slot per lane of DPP row 0");

// P points per lane; wave w owns the sorted positions [w 64 P, (w + 1) 64 P): lane l, slot i <-> position w 64 P + 64 i + l.
// The coordinates of the last PL slots live in LDS (read-only, conflict-free columns) instead of VGPRs: at P = 24 the four
// arrays would need 96 of the 128 registers a 16-wave workgroup gets and the compiler spills coordinates to scratch.
template <int P, int PL, bool FMA>
__global__ __launch_bounds__(FW_BS) void fps_wave_kernel(int n, int m, const float* __restrict__ xyz, const int* __restrict__ perm,
                                                         int* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* perm_l = reinterpret_cast<int*>(smem);                               // [FW_BS * P] original index of a sorted position
    float* slot = reinterpret_cast<float*>(perm_l + FW_BS * P);               // [2][FW_W][8]: d, position, x, y, z
    float* xl = slot + 2 * FW_W * 8;                                          // [PL][3][FW_BS]
    constexpr int PR = P - PL;
    const int cloud = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);               // scalar: everything per-wave below lives in SGPRs
    const float* __restrict__ p = xyz + (size_t)cloud * n * 3;
    const int* __restrict__ pm = perm + (size_t)cloud * n;
    int* __restrict__ o = out + (size_t)cloud * m;
    const int wbase = wave * 64 * P;

    float x[PR], y[PR], z[PR], td[P];
    float lx = 3e38f, ly = 3e38f, lz = 3e38f, hx = -3e38f, hy = -3e38f, hz = -3e38f;
    int kk[P];
#pragma unroll
    for (int i = 0; i < P; ++i) kk[i] = pm[min(wbase + i * 64 + lane, n - 1)];     // unconditional: all loads in flight at once
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int pos = wbase + i * 64 + lane;
        const bool ok = pos < n;
        const int k = kk[i];
        perm_l[pos] = k;
        const float px = p[k * 3 + 0], py = p[k * 3 + 1], pz = p[k * 3 + 2];
        if (i < PR) { x[i < PR ? i : 0] = px; y[i < PR ? i : 0] = py; z[i < PR ? i : 0] = pz; }
        else { float* c = xl + (size_t)(i - PR) * 3 * FW_BS + tid; c[0] = px; c[FW_BS] = py; c[2 * FW_BS] = pz; }
        td[i] = ok ? 1e38f : -1.0f;                      // -1: "no point here", never the farthest
        if (ok) { lx = fminf(lx, px); hx = fmaxf(hx, px); ly = fminf(ly, py); hy = fmaxf(hy, py); lz = fminf(lz, pz); hz = fmaxf(hz, pz); }
    }
    // the wave's region: bounding box (wave-uniform), largest running distance, cached candidate
    lx = fpsw_uniform(fpsw_all_min(lx)); ly = fpsw_uniform(fpsw_all_min(ly)); lz = fpsw_uniform(fpsw_all_min(lz));
    hx = fpsw_uniform(fpsw_all_max(hx)); hy = fpsw_uniform(fpsw_all_max(hy)); hz = fpsw_uniform(fpsw_all_max(hz));
    float wtd = (wbase < n) ? 1e38f : -1.0f;             // non-empty wave: active in the first round
    uint32_t rk = 0u;                                    // candidate record: distance key, sorted position, coordinates
    float rx = 0.f, ry = 0.f, rz = 0.f;
    int rpos = 0;
    if (tid == 0) o[0] = 0;
    float x1 = p[0], y1 = p[1], z1 = p[2];               // sample 0 is point 0 (tf_sampling_g.cu:122-124)
    __syncthreads();                                      // perm_l complete
    FPSW_PROF_BEGIN
    for (int j = 1; j < m; ++j) {
        FPSW_TICK(0)
        const float ex = fmaxf(fmaxf(lx - x1, x1 - hx), 0.f);
        const float ey = fmaxf(fmaxf(ly - y1, y1 - hy), 0.f);
        const float ez = fmaxf(fmaxf(lz - z1, z1 - hz), 0.f);
        const float lb = __builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex));
        FPSW_TICK(1)
        if (lb * 0.99999f <= wtd) {                       // wave-uniform: can any running distance of this wave change?
            FPSW_PROF_ACTIVE
            float bd = -1.0f;
            int bi = 0;
#pragma unroll
            for (int i = 0; i < P; ++i) {                 // slots in ascending tie priority: strict '>' keeps the reference's winner
                float px, py, pz;
                if (i < PR) { px = x[i < PR ? i : 0]; py = y[i < PR ? i : 0]; pz = z[i < PR ? i : 0]; }
                else { const float* c = xl + (size_t)(i - PR) * 3 * FW_BS + tid; px = c[0]; py = c[FW_BS]; pz = c[2 * FW_BS]; }
                const float d = sqdist3<FMA>(px - x1, py - y1, pz - z1);
                const float t = fminf(d, td[i]);
                td[i] = t;
                const bool gt = t > bd;
                bd = gt ? t : bd; bi = gt ? i : bi;
            }
            FPSW_TICK(2)
            const uint32_t ub = fpsw_dkey(bd);
            const uint32_t um = fpsw_wave_max_u32(ub);
            unsigned long long tm = __ballot(ub == um);
            int wl = (int)__builtin_ctzll(tm);
            if (tm & (tm - 1)) {                          // several lanes share the maximum: the reference's tie rule decides
                const bool in = ub == um;
                const uint32_t key = in ? fpsw_tiekey(perm_l[wbase + bi * 64 + lane]) : 0u;
                const uint32_t mk = fpsw_wave_max_u32(key);
                wl = (int)__builtin_ctzll(__ballot(in && key == mk));
            }
            const int sbi = __builtin_amdgcn_readlane(bi, wl);   // scalar slot of the winner: a scalar branch picks its registers
            float cx = 0.f, cy = 0.f, cz = 0.f;
            switch (sbi) {
#define FPSW_CASE(I) case I: if constexpr ((I) < PR) { cx = x[(I) < PR ? (I) : 0]; cy = y[(I) < PR ? (I) : 0]; cz = z[(I) < PR ? (I) : 0]; asm volatile("" : "+v"(cx), "+v"(cy), "+v"(cz)); } break;
                FPSW_CASE(0) FPSW_CASE(1) FPSW_CASE(2) FPSW_CASE(3) FPSW_CASE(4) FPSW_CASE(5) FPSW_CASE(6) FPSW_CASE(7)
                FPSW_CASE(8) FPSW_CASE(9) FPSW_CASE(10) FPSW_CASE(11) FPSW_CASE(12) FPSW_CASE(13) FPSW_CASE(14) FPSW_CASE(15)
                FPSW_CASE(16) FPSW_CASE(17) FPSW_CASE(18) FPSW_CASE(19) FPSW_CASE(20) FPSW_CASE(21) FPSW_CASE(22) FPSW_CASE(23)
#undef FPSW_CASE
                default: break;
            }
            if (PL > 0 && sbi >= PR) { const float* c = xl + (size_t)(sbi - PR) * 3 * FW_BS + tid; cx = c[0]; cy = c[FW_BS]; cz = c[2 * FW_BS]; }
            wtd = fpsw_dkey_value(um);
            rk = um;
            rpos = wbase + sbi * 64 + wl;
            rx = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(cx), wl));
            ry = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(cy), wl));
            rz = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(cz), wl));
        }
        FPSW_TICK(3)
        const int par = j & 1;
        if (lane == 0) {
            float* s = slot + (size_t)(par * FW_W + wave) * 8;
            *reinterpret_cast<float4*>(s) = make_float4(__uint_as_float(rk), __int_as_float(rpos), rx, ry);
            s[4] = rz;
        }
        FPSW_TICK(4)
        __syncthreads();
        FPSW_TICK(5)
        {   // winner among the waves (every wave computes it: lanes < FW_W read one slot each)
            uint32_t sk = 0u;
            float sx = 0.f, sy = 0.f, sz = 0.f;
            int spos = 0;
            if (lane < FW_W) {
                const float* s = slot + (size_t)(par * FW_W + lane) * 8;
                const float4 a = *reinterpret_cast<const float4*>(s);
                sk = __float_as_uint(a.x); spos = __float_as_int(a.y); sx = a.z; sy = a.w; sz = s[4];
            }
            const uint32_t gm = fpsw_row0_max_u32(sk);  // > 0: some wave holds points
            unsigned long long tm = __ballot(sk == gm);
            int gl = (int)__builtin_ctzll(tm);
            if (tm & (tm - 1)) {
                const bool in = sk == gm;
                const uint32_t key = in ? fpsw_tiekey(perm_l[spos]) : 0u;
                const uint32_t mk = fpsw_wave_max_u32(key);
                gl = (int)__builtin_ctzll(__ballot(in && key == mk));
            }
            x1 = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(sx), gl));
            y1 = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(sy), gl));
            z1 = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(sz), gl));
            if (tid == 0) o[j] = perm_l[__builtin_amdgcn_readlane(spos, gl)];
        }
        FPSW_TICK(6)
    }
    FPSW_PROF_END
}

// ---- four skip regions per wave ------------------------------------------------------------------------------------------
// fps_wave_kernel skips whole waves (64 P points).  Its round at n = 24576 (tools/micro/fps_wave_prof.hip) is led by the dense
// update of the busiest wave (1330 of 2900 cycles): a new sample usually touches a corner of that wave's region only.  Here a
// wave's P slots form G = 4 groups of PG = P / 4 slots = 64 PG consecutive sorted positions each (the pre-pass orders every
// GROUP's range by tie priority), with their own bounding boxes (kept in lanes 0..3 of six VGPRs: ONE 14-instruction bound test
// serves the four groups, its ballot bits steer scalar branches).  A group whose box is farther from the new sample than the
// wave's largest running distance is skipped; every lane caches its best (distance, slot) per group, so the candidates of skipped
// groups are still there.  Equal distances in different groups of one lane, or in different lanes, go to a (rare) path that
// compares the reference's tie keys of every (lane, group) candidate.  Sorted positions are written to `out` during the loop
// and replaced by the original indices at the end (the position -> index table is read from global memory: it is needed for the
// rare ties and the final pass only, which frees 96 KB of LDS for the coordinates of 8 of the 24 slots).
template <int P, int PL, bool FMA, int G = 4>
__global__ __launch_bounds__(FW_BS) void fps_wave4_kernel(int n, int m, const float* __restrict__ xyz, const int* __restrict__ perm,
                                                          int* __restrict__ out) {
    constexpr int PG = P / G, PR = P - PL;                // G = 4 (default) or 8 skip groups per wave (round 3: measured)
    static_assert(P % G == 0, "groups must tile the slots");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* slot = reinterpret_cast<float*>(smem);                             // [2][FW_W][8]: distance key, position, x, y, z
    float* xl = slot + 2 * FW_W * 8;                                          // [PL][3][FW_BS]
    const int cloud = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* __restrict__ p = xyz + (size_t)cloud * n * 3;
    const int* __restrict__ pm = perm + (size_t)cloud * n;
    int* __restrict__ o = out + (size_t)cloud * m;
    const int wbase = wave * 64 * P;

    float x[PR], y[PR], z[PR], td[P];
    float blx = 3e38f, bly = 3e38f, blz = 3e38f, bhx = -3e38f, bhy = -3e38f, bhz = -3e38f;   // lane g < 4: box of group g
    {
        int kk[P];
#pragma unroll
        for (int i = 0; i < P; ++i) kk[i] = pm[min(wbase + i * 64 + lane, n - 1)];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float lx = 3e38f, ly = 3e38f, lz = 3e38f, hx = -3e38f, hy = -3e38f, hz = -3e38f;
#pragma unroll
            for (int s = 0; s < PG; ++s) {
                const int i = g * PG + s;
                const bool ok = wbase + i * 64 + lane < n;
                const int k = kk[i];
                const float px = p[k * 3 + 0], py = p[k * 3 + 1], pz = p[k * 3 + 2];
                if (i < PR) { x[i < PR ? i : 0] = px; y[i < PR ? i : 0] = py; z[i < PR ? i : 0] = pz; }
                else { float* c = xl + (size_t)(i - PR) * 3 * FW_BS + tid; c[0] = px; c[FW_BS] = py; c[2 * FW_BS] = pz; }
                td[i] = ok ? 1e38f : -1.0f;
                if (ok) { lx = fminf(lx, px); hx = fmaxf(hx, px); ly = fminf(ly, py); hy = fmaxf(hy, py); lz = fminf(lz, pz); hz = fmaxf(hz, pz); }
            }
            lx = fpsw_all_min(lx); ly = fpsw_all_min(ly); lz = fpsw_all_min(lz);
            hx = fpsw_all_max(hx); hy = fpsw_all_max(hy); hz = fpsw_all_max(hz);
            if (lane == g) { blx = lx; bly = ly; blz = lz; bhx = hx; bhy = hy; bhz = hz; }
        }
    }
    float gbd[G];                                         // per lane: best running distance of its PG slots in group g ...
    int gbi[G];                                           // ... and the slot (0 .. PG - 1) that holds it
#pragma unroll
    for (int g = 0; g < G; ++g) { gbd[g] = -1.0f; gbi[g] = 0; }
    float wtd = (wbase < n) ? 1e38f : -1.0f;
    uint32_t rk = 0u;
    float rx = 0.f, ry = 0.f, rz = 0.f;
    int rpos = 0;
    float x1 = p[0], y1 = p[1], z1 = p[2];
    __syncthreads();                                      // xl complete
    FPSW_PROF_BEGIN
    for (int j = 1; j < m; ++j) {
        FPSW_TICK(0)
        const float ex = fmaxf(fmaxf(blx - x1, x1 - bhx), 0.f);
        const float ey = fmaxf(fmaxf(bly - y1, y1 - bhy), 0.f);
        const float ez = fmaxf(fmaxf(blz - z1, z1 - bhz), 0.f);
        const float lb = __builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex));
        const unsigned act = (unsigned)__ballot(lane < G && lb * 0.99999f <= wtd) & ((1u << G) - 1u);      // wave-uniform group mask
        FPSW_TICK(1)
        if (act) {
            FPSW_PROF_ACTIVE
            FPSW_PROF_GROUPS(act)
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (act & (1u << g)) {
                    float bd = -1.0f;
                    int bi = 0;
#pragma unroll
                    for (int s = 0; s < PG; ++s) {        // slots of a group in ascending tie priority: strict '>' keeps the reference's winner
                        const int i = g * PG + s;
                        float px, py, pz;
                        if (i < PR) { px = x[i < PR ? i : 0]; py = y[i < PR ? i : 0]; pz = z[i < PR ? i : 0]; }
                        else { const float* c = xl + (size_t)(i - PR) * 3 * FW_BS + tid; px = c[0]; py = c[FW_BS]; pz = c[2 * FW_BS]; }
                        const float d = sqdist3<FMA>(px - x1, py - y1, pz - z1);
                        const float t = fminf(d, td[i]);
                        td[i] = t;
                        const bool gt = t > bd;
                        bd = gt ? t : bd; bi = gt ? s : bi;
                    }
                    gbd[g] = bd; gbi[g] = bi;
                }
            }
            FPSW_TICK(2)
            // the lane's best over its four groups; `amb`: two groups tie for it (their tie keys have to decide)
            float bd = gbd[0];
            int bs = gbi[0];
            bool amb = false;
#pragma unroll
            for (int g = 1; g < G; ++g) {
                const bool gt = gbd[g] > bd, eq = gbd[g] == bd;
                amb = gt ? false : (amb || eq);
                bs = gt ? g * PG + gbi[g] : bs;
                bd = gt ? gbd[g] : bd;
            }
            const uint32_t ub = fpsw_dkey(bd);
            const uint32_t um = fpsw_wave_max_u32(ub);
            const unsigned long long tm = __ballot(ub == um);
            int wl = (int)__builtin_ctzll(tm);
            int sbi = __builtin_amdgcn_readlane(bs, wl);
            if ((tm & (tm - 1)) || __ballot(amb && ub == um)) {
                // several (lane, group) candidates share the maximum: the reference's tie rule on their original indices
                uint32_t bk = 0u;
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const bool in = fpsw_dkey(gbd[g]) == um;
                    const int pos = wbase + (g * PG + gbi[g]) * 64 + lane;
                    const uint32_t key = in ? fpsw_tiekey(pm[min(pos, n - 1)]) : 0u;
                    const uint32_t mk = fpsw_wave_max_u32(key);
                    if (mk > bk) {                        // wave-uniform
                        bk = mk;
                        wl = (int)__builtin_ctzll(__ballot(in && key == mk));
                        sbi = g * PG + __builtin_amdgcn_readlane(gbi[g], wl);
                    }
                }
            }
            float cx = 0.f, cy = 0.f, cz = 0.f;
            switch (sbi) {
#define FPSW_CASE(I) case I: if constexpr ((I) < PR) { cx = x[(I) < PR ? (I) : 0]; cy = y[(I) < PR ? (I) : 0]; cz = z[(I) < PR ? (I) : 0]; asm volatile("" : "+v"(cx), "+v"(cy), "+v"(cz)); } break;
                FPSW_CASE(0) FPSW_CASE(1) FPSW_CASE(2) FPSW_CASE(3) FPSW_CASE(4) FPSW_CASE(5) FPSW_CASE(6) FPSW_CASE(7)
                FPSW_CASE(8) FPSW_CASE(9) FPSW_CASE(10) FPSW_CASE(11) FPSW_CASE(12) FPSW_CASE(13) FPSW_CASE(14) FPSW_CASE(15)
                FPSW_CASE(16) FPSW_CASE(17) FPSW_CASE(18) FPSW_CASE(19) FPSW_CASE(20) FPSW_CASE(21) FPSW_CASE(22) FPSW_CASE(23)
#undef FPSW_CASE
                default: break;
            }
            if (PL > 0 && sbi >= PR) { const float* c = xl + (size_t)(sbi - PR) * 3 * FW_BS + tid; cx = c[0]; cy = c[FW_BS]; cz = c[2 * FW_BS]; }
            wtd = fpsw_dkey_value(um);
            rk = um;
            rpos = wbase + sbi * 64 + wl;
            rx = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(cx), wl));
            ry = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(cy), wl));
            rz = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(cz), wl));
        }
        FPSW_TICK(3)
        const int par = j & 1;
        if (lane == 0) {
            float* s = slot + (size_t)(par * FW_W + wave) * 8;
            *reinterpret_cast<float4*>(s) = make_float4(__uint_as_float(rk), __int_as_float(rpos), rx, ry);
            s[4] = rz;
        }
        FPSW_TICK(4)
        __syncthreads();
        FPSW_TICK(5)
        {   // winner among the waves
            uint32_t sk = 0u;
            float sx = 0.f, sy = 0.f, sz = 0.f;
            int spos = 0;
            if (lane < FW_W) {
                const float* s = slot + (size_t)(par * FW_W + lane) * 8;
                const float4 a = *reinterpret_cast<const float4*>(s);
                sk = __float_as_uint(a.x); spos = __float_as_int(a.y); sx = a.z; sy = a.w; sz = s[4];
            }
            const uint32_t gm = fpsw_row0_max_u32(sk);
            const unsigned long long tm = __ballot(sk == gm);
            int gl = (int)__builtin_ctzll(tm);
            if (tm & (tm - 1)) {
                const bool in = sk == gm;
                const uint32_t key = in ? fpsw_tiekey(pm[min(spos, n - 1)]) : 0u;
                const uint32_t mk = fpsw_wave_max_u32(key);
                gl = (int)__builtin_ctzll(__ballot(in && key == mk));
            }
            x1 = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(sx), gl));
            y1 = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(sy), gl));
            z1 = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(sz), gl));
            if (tid == 0) o[j] = __builtin_amdgcn_readlane(spos, gl);        // a sorted POSITION; translated below
        }
        FPSW_TICK(6)
    }
    FPSW_PROF_END
    __syncthreads();                                      // thread 0's stores are visible to the workgroup
    for (int j = 1 + tid; j < m; j += FW_BS) o[j] = pm[o[j]];
    if (tid == 0) o[0] = 0;                               // sample 0 is point 0 (tf_sampling_g.cu:122-124)
}

template <int P, int PL, int G = 4>
static int launch_fps_wave4(int b, int n, int m, const float* xyz, int* perm, int* out, int arith, hipStream_t s) {
    const size_t sort_bytes = (size_t)n * 4 + 4096 * 4 + 2048 * 4;
    const size_t bytes = 2 * FW_W * 8 * 4 + (size_t)PL * 3 * FW_BS * 4;
    static DevOnce attr;
    if (attr.needed()) {
        DISPU_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fps_wavesort_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 24576 * 4 + 4096 * 4 + 2048 * 4));
        DISPU_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fps_wave4_kernel<P, PL, true, G>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        DISPU_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fps_wave4_kernel<P, PL, false, G>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        attr.done();
    }
    hipLaunchKernelGGL(fps_wavesort_kernel, dim3(b), dim3(1024), sort_bytes, s, n, 64 * (P / G), xyz, perm);
    DISPU_CHECK_LAUNCH();
    if ((arith & DISPU_ARITH_CONTRACT))
        hipLaunchKernelGGL((fps_wave4_kernel<P, PL, true, G>), dim3(b), dim3(FW_BS), bytes, s, n, m, xyz, perm, out);
    else
        hipLaunchKernelGGL((fps_wave4_kernel<P, PL, false, G>), dim3(b), dim3(FW_BS), bytes, s, n, m, xyz, perm, out);
    return (int)hipGetLastError();
}

template <int P, int PL>
static int launch_fps_wave(int b, int n, int m, const float* xyz, int* perm, int* out, int arith, hipStream_t s) {
    const size_t sort_bytes = (size_t)n * 4 + 4096 * 4 + 2048 * 4;
    const size_t bytes = (size_t)FW_BS * P * 4 + 2 * FW_W * 8 * 4 + (size_t)PL * 3 * FW_BS * 4;
    static DevOnce attr;      
    if (attr.needed()) {
        DISPU_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fps_wavesort_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 24576 * 4 + 4096 * 4 + 2048 * 4));
        DISPU_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fps_wave_kernel<P, PL, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        DISPU_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fps_wave_kernel<P, PL, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        attr.done();
    }
    hipLaunchKernelGGL(fps_wavesort_kernel, dim3(b), dim3(1024), sort_bytes, s, n, 64 * P, xyz, perm);
    DISPU_CHECK_LAUNCH();
    if ((arith & DISPU_ARITH_CONTRACT))
        hipLaunchKernelGGL((fps_wave_kernel<P, PL, true>), dim3(b), dim3(FW_BS), bytes, s, n, m, xyz, perm, out);
    else
        hipLaunchKernelGGL((fps_wave_kernel<P, PL, false>), dim3(b), dim3(FW_BS), bytes, s, n, m, xyz, perm, out);
    return (int)hipGetLastError();
}

bool fps_wave_wants_scratch(int n, int m) { return n > 4096 && n <= FW_BS * 24 && m >= 64; }

// -1: shape outside this path (or no scratch for the permutation)
int fps_wave_dispatch(int b, int n, int m, const float* xyz, void* temp, int* out, int arith, hipStream_t s) {
    if (!temp || !fps_wave_wants_scratch(n, m)) return -1;
    int* perm = reinterpret_cast<int*>(temp);
    // n <= 8192: a skip region is a wave; above: four regions per wave.  (Measured and not kept as switches: whole-wave regions above
    // 8192 points, 9.6 - 10.4 vs 8.5 - 9.5 ms at (8, 24576, 8192); eight regions per wave, -2 %: profiles/EXPERIMENTS.md.)
    if (n <= FW_BS * 8) return launch_fps_wave<8, 0>(b, n, m, xyz, perm, out, arith, s);
    if (n <= FW_BS * 16) return launch_fps_wave4<16, 0>(b, n, m, xyz, perm, out, arith, s);
    return launch_fps_wave4<24, 8>(b, n, m, xyz, perm, out, arith, s);
}

}  // namespace dispu
