// Selection helpers shared by the wave-per-query k-NN kernels (knn_wave.hip) and the fused dense-block kernel (edge.hip):
// register sorting networks, DPP wave minima, the threshold prefilter + rank and the two-query head selection.
#pragma once
#include "common.h"

namespace dispu {

constexpr uint64_t KEY_MAX = ~0ull;

__device__ __forceinline__ void cswap(uint64_t& a, uint64_t& b) {
    const bool c = a > b;
    const uint64_t lo = c ? b : a, hi = c ? a : b;
    a = lo; b = hi;
}

// Batcher odd-even merge sort, fully unrolled (R a power of two): ascending.
template <int R>
__device__ __forceinline__ void sort_keys(uint64_t (&a)[R]) {
#pragma unroll
    for (int p = 1; p < R; p <<= 1)
#pragma unroll
        for (int k = p; k >= 1; k >>= 1)
#pragma unroll
            for (int j = k % p; j <= R - 1 - k; j += 2 * k)
#pragma unroll
                for (int i = 0; i < k; ++i)
                    if (i + j + k < R && (i + j) / (2 * p) == (i + j + k) / (2 * p)) cswap(a[i + j], a[i + j + k]);
}

// k rounds of wave-min over the lanes' heads; lane t keeps result t.  sorted: this wave's [R][64] key columns.
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    v = min(v, dpp_u32<DPP_ROW_SHR1>(0xFFFFFFFFu, v));
    v = min(v, dpp_u32<DPP_ROW_SHR2>(0xFFFFFFFFu, v));
    v = min(v, dpp_u32<DPP_ROW_SHR4>(0xFFFFFFFFu, v));
    v = min(v, dpp_u32<DPP_ROW_SHR8>(0xFFFFFFFFu, v));
    v = min(v, dpp_u32<DPP_ROW_BCAST15, 0xA>(0xFFFFFFFFu, v));
    v = min(v, dpp_u32<DPP_ROW_BCAST31, 0xC>(0xFFFFFFFFu, v));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// The round is a serial dependency chain, so it is kept short: a 32-bit DPP min over the heads' distance words,
// a ballot to find the owner (the index word only matters when two heads tie on the distance - rare), and the
// winner's next key is already in a register (prefetched from LDS one advance ahead).
template <int R>
__device__ __forceinline__ uint64_t select_k(const uint64_t (&key)[R], uint64_t* sorted, int lane, int k) {
#pragma unroll
    for (int i = 1; i < R; ++i) sorted[i * 64 + lane] = key[i];
    uint32_t hd = (uint32_t)(key[0] >> 32), hi = (uint32_t)key[0];
    uint64_t nk = (R > 1) ? key[R > 1 ? 1 : 0] : KEY_MAX;         // head + 1, kept in registers
    int hp = 0;
    uint64_t res = 0;
    for (int t = 0; t < k; ++t) {
        const uint32_t md = wave_min_u32(hd);
        unsigned long long mask = __ballot(hd == md);
        if (__popcll(mask) != 1) {                                  // wave-uniform; equal distances: lowest index wins
            const uint32_t mi = wave_min_u32(hd == md ? hi : 0xFFFFFFFFu);
            mask = __ballot(hd == md && hi == mi);
        }
        const int win = __builtin_ctzll(mask);
        const uint32_t wi = (uint32_t)__builtin_amdgcn_readlane((int)hi, win);
        if (lane == t) res = ((uint64_t)md << 32) | wi;
        if (lane == win) {
            hd = (uint32_t)(nk >> 32); hi = (uint32_t)nk;
            ++hp;
            nk = (hp + 1 < R) ? sorted[(hp + 1) * 64 + lane] : KEY_MAX;
        }
    }
    return res;
}

// ---- threshold prefilter + one cross-lane sort (n > 256: R >= 8 keys per lane) --------------------------------
// Sorting 16 keys in every lane and then popping k heads costs ~1200 instructions per query although only k of the
// 64 R candidates matter.  Instead: T = an upper bound of the k-th smallest distance, obtained from the lanes' minima
// (the k-th smallest of the 64 lane minima: k distinct candidates are <= T); the survivors (distance <= T, typically
// 1.2 k - 2 k of the 1024) are compacted into LDS with
// ballot / mbcnt, and every survivor counts the survivors with a smaller (distance, index) key: that rank is its
// position in the result.  Same result as the full sort: ascending distance, ties -> lower index.  More than 128
// survivors (degenerate clouds: many candidates at exactly the same distance) take the full path.
// Ascending sort of one 32-bit word per lane over the 64 lanes: the bitonic network in its "flip" form (phase K2 first
// pairs lane i with i ^ (K2 - 1), then with i ^ J for J = K2/4 .. 1; the lower lane of a pair always keeps the
// minimum).  21 compare-exchange steps in 56 VALU instructions: the min / max take the partner through the DPP operand
// directly, and wherever "lower lane" is a whole DPP bank (4 lanes) the bank_mask of the two instructions does the
// select (min written to the lower banks, max to the upper ones).  Inline asm because the compiler keeps v_mov_dpp +
// v_min + v_max + v_cndmask per step; the leading s_nop covers the VALU-write -> DPP-read hazard it cannot see.
#define DISPU_SORT_BANK(v, ctl_lo, ctl_hi, bm_lo, bm_hi)                                                          \
    {                                                                                                              \
        uint32_t t_;                                                                                               \
        asm("s_nop 1\n\tv_min_u32_dpp %0, %1, %1 " ctl_lo " row_mask:0xf bank_mask:" bm_lo                         \
            "\n\tv_max_u32_dpp %0, %1, %1 " ctl_hi " row_mask:0xf bank_mask:" bm_hi                                 \
            : "=&v"(t_) : "v"(v));                                                                                 \
        v = t_;                                                                                                    \
    }
#define DISPU_SORT_QUAD(v, perm, lower)                                                                           \
    {                                                                                                              \
        uint32_t lo_, hi_;                                                                                         \
        asm("s_nop 1\n\tv_min_u32_dpp %0, %2, %2 quad_perm:" perm " row_mask:0xf bank_mask:0xf"                    \
            "\n\tv_max_u32_dpp %1, %2, %2 quad_perm:" perm " row_mask:0xf bank_mask:0xf"                            \
            : "=&v"(lo_), "=&v"(hi_) : "v"(v));                                                                    \
        v = (lower) ? lo_ : hi_;                                                                                   \
    }
#define DISPU_SORT_X8(v) DISPU_SORT_BANK(v, "row_ror:8", "row_ror:8", "0x3", "0xc")
#define DISPU_SORT_X4(v) DISPU_SORT_BANK(v, "row_shl:4", "row_shr:4", "0x5", "0xa")
#define DISPU_SORT_X2(v) DISPU_SORT_QUAD(v, "[2,3,0,1]", e2)
#define DISPU_SORT_X1(v) DISPU_SORT_QUAD(v, "[1,0,3,2]", e1)
__device__ __forceinline__ uint32_t wave_bitonic_sort_u32(uint32_t v, int lane) {
    const bool e1 = (lane & 1) == 0, e2 = (lane & 2) == 0, e16 = (lane & 16) == 0, e32 = lane < 32;
    DISPU_SORT_X1(v);                                                                             // K2 = 2
    DISPU_SORT_QUAD(v, "[3,2,1,0]", e2); DISPU_SORT_X1(v);                                        // K2 = 4
    DISPU_SORT_BANK(v, "row_half_mirror", "row_half_mirror", "0x5", "0xa"); DISPU_SORT_X2(v); DISPU_SORT_X1(v);   // 8
    DISPU_SORT_BANK(v, "row_mirror", "row_mirror", "0x3", "0xc"); DISPU_SORT_X4(v); DISPU_SORT_X2(v); DISPU_SORT_X1(v);   // 16
    {   // K2 = 32: partner lane ^ 31 (ds_swizzle bit mode: and 0x1F, xor 0x1F)
        const uint32_t o = (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x7C1F);
        v = e16 ? min(v, o) : max(v, o);
    }
    DISPU_SORT_X8(v); DISPU_SORT_X4(v); DISPU_SORT_X2(v); DISPU_SORT_X1(v);
    {   // K2 = 64: partner 63 - lane, then lane ^ 16
        uint32_t o = (uint32_t)__builtin_amdgcn_ds_bpermute((63 - lane) * 4, (int)v);
        v = e32 ? min(v, o) : max(v, o);
        o = (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x401F);
        v = e16 ? min(v, o) : max(v, o);
    }
    DISPU_SORT_X8(v); DISPU_SORT_X4(v); DISPU_SORT_X2(v); DISPU_SORT_X1(v);
    return v;
}

// Threshold prefilter + rank (see above).  od: distance words (any encoding whose unsigned order is the distance order;
// candidates that do not exist carry a word > tmax), cp: candidate indices, buf: this wave's LDS scratch (wave-uniform
// pointer; CAP + 4 slots when GUARD, else room for every candidate + 4).  Writes the k results (ascending distance,
// ties -> lower index) as idx_out[t] / word_out[t] and returns true; returns false - nothing written - when fewer than
// k or more than 128 candidates pass the threshold (the caller then sorts everything).
#ifdef KNN_STAMPS
#define PF_STAMP(i) { __builtin_amdgcn_sched_barrier(0); if (pf_stamps) pf_stamps[i] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); }
#else
#define PF_STAMP(i)
#endif
template <int R, bool GUARD, typename WORD2F>
__device__ __forceinline__ bool prefilter_rank(const uint32_t (&od)[R], const int (&cp)[R], uint64_t* buf, int lane, int k, uint32_t tmax,
                                               int* __restrict__ idx_out, float* __restrict__ dist_out, WORD2F word_to_float,
                                               unsigned long long* pf_stamps = nullptr) {
    constexpr int CAP = 128;
    PF_STAMP(0);
    uint32_t dmin = od[0];
#pragma unroll
    for (int r = 1; r < R; ++r) dmin = min(dmin, od[r]);
    // k-th smallest of the 64 lane minima: k distinct candidates are <= T, and on average only ~1.2 k candidates are
    uint32_t T = (uint32_t)__builtin_amdgcn_readlane((int)wave_bitonic_sort_u32(dmin, lane), k - 1);
    T = min(T, tmax);
    PF_STAMP(1);
    int cnt = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const bool flag = od[r] <= T;
        const unsigned long long mk = __ballot(flag);
        const int pos = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u));
        if (flag && (!GUARD || cnt + pos < CAP)) (buf + cnt)[pos] = ((uint64_t)od[r] << 32) | (uint32_t)cp[r];
        cnt += __popcll(mk);
    }
    PF_STAMP(2);
    if (cnt < k || cnt > CAP) return false;                             // wave-uniform
    // rank of every survivor among the survivors (keys are distinct): broadcast reads of the compacted list, one
    // compare + one add-with-carry per pair; the survivor of rank t < k IS result t
    if (lane < 4) buf[cnt + lane] = KEY_MAX;                            // the loop reads in fours
    const uint64_t m0 = (lane < cnt) ? buf[lane] : KEY_MAX;
    if (cnt <= 64) {
        int r0 = 0;
        for (int j = 0; j < cnt; j += 4) {
            const uint64_t a0 = buf[j], a1 = buf[j + 1], a2 = buf[j + 2], a3 = buf[j + 3];
            r0 += (int)(a0 < m0) + (int)(a1 < m0) + (int)(a2 < m0) + (int)(a3 < m0);
        }
        if (r0 < k) {
            idx_out[r0] = (int)(uint32_t)m0;
            if (dist_out) dist_out[r0] = word_to_float((uint32_t)(m0 >> 32));
        }
    } else {
        const uint64_t m1 = (lane + 64 < cnt) ? buf[lane + 64] : KEY_MAX;
        int r0 = 0, r1 = 0;
        for (int j = 0; j < cnt; j += 4) {
            const uint64_t a0 = buf[j], a1 = buf[j + 1], a2 = buf[j + 2], a3 = buf[j + 3];
            r0 += (int)(a0 < m0) + (int)(a1 < m0) + (int)(a2 < m0) + (int)(a3 < m0);
            r1 += (int)(a0 < m1) + (int)(a1 < m1) + (int)(a2 < m1) + (int)(a3 < m1);
        }
        if (r0 < k) {
            idx_out[r0] = (int)(uint32_t)m0;
            if (dist_out) dist_out[r0] = word_to_float((uint32_t)(m0 >> 32));
        }
        if (r1 < k) {
            idx_out[r1] = (int)(uint32_t)m1;
            if (dist_out) dist_out[r1] = word_to_float((uint32_t)(m1 >> 32));
        }
    }
    PF_STAMP(3);
    return true;
}

// Two queries at once, heads and the whole per-lane sorted list in registers (R <= 4: the next head after a win is a
// 3-deep select on the lane's position instead of an LDS read).  The two selection chains are independent, so their
// DPP / readlane / ballot latencies overlap; lane t keeps result t of both.
template <int R>
__device__ __forceinline__ void select_k2(const uint64_t (&ka)[R], const uint64_t (&kb)[R], int lane, int k, uint64_t& ra, uint64_t& rb) {
    static_assert(R <= 4, "register-resident selection is written for up to four keys per lane");
    uint32_t hda = (uint32_t)(ka[0] >> 32), hia = (uint32_t)ka[0], hdb = (uint32_t)(kb[0] >> 32), hib = (uint32_t)kb[0];
    int hpa = 0, hpb = 0;
    ra = 0; rb = 0;
    auto nth = [&](const uint64_t (&key)[R], int hp) -> uint64_t {
        uint64_t v = KEY_MAX;
#pragma unroll
        for (int i = R - 1; i >= 1; --i) v = (hp == i) ? key[i] : v;
        return v;
    };
    for (int t = 0; t < k; ++t) {
        const uint32_t mda = wave_min_u32(hda), mdb = wave_min_u32(hdb);
        unsigned long long ma = __ballot(hda == mda), mb = __ballot(hdb == mdb);
        if (__popcll(ma) != 1) {                                    // wave-uniform; equal distances: lowest index wins
            const uint32_t mi = wave_min_u32(hda == mda ? hia : 0xFFFFFFFFu);
            ma = __ballot(hda == mda && hia == mi);
        }
        if (__popcll(mb) != 1) {
            const uint32_t mi = wave_min_u32(hdb == mdb ? hib : 0xFFFFFFFFu);
            mb = __ballot(hdb == mdb && hib == mi);
        }
        const int wa = __builtin_ctzll(ma), wb = __builtin_ctzll(mb);
        const uint32_t wia = (uint32_t)__builtin_amdgcn_readlane((int)hia, wa), wib = (uint32_t)__builtin_amdgcn_readlane((int)hib, wb);
        if (lane == t) { ra = ((uint64_t)mda << 32) | wia; rb = ((uint64_t)mdb << 32) | wib; }
        if (lane == wa) { ++hpa; const uint64_t nk = nth(ka, hpa); hda = (uint32_t)(nk >> 32); hia = (uint32_t)nk; }
        if (lane == wb) { ++hpb; const uint64_t nk = nth(kb, hpb); hdb = (uint32_t)(nk >> 32); hib = (uint32_t)nk; }
    }
}

}  // namespace dispu
