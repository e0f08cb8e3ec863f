// Wave-wide selection primitives shared by the grid self-kNN (grid.hip) and the brute-force candidate selection
// (knn_bruteforce.hip): a sorted top-k held one entry per lane, a 64-lane bitonic network on DPP moves, and insertion of
// one chunk of candidates by ballot + one-lane shift.
// A candidate is ordered by (squared distance, index): for non-negative floats the bit pattern orders like the value, so
// the pair packs into one 64-bit key and "nearer, ties to the lower index" is a single unsigned compare.
#pragma once
#include "sp_common.h"

namespace sp {

struct Cand {
    unsigned long long key;  // (float bits of d2) << 32 | index
    int pos;                 // position in grid order
};
__device__ __forceinline__ unsigned long long cand_key(float d, int idx) {
    return ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)idx;
}
__device__ __forceinline__ float key_d2(unsigned long long k) { return __uint_as_float((unsigned)(k >> 32)); }
__device__ __forceinline__ int key_idx(unsigned long long k) { return (int)(unsigned)k; }
constexpr unsigned long long kNoCand = ((unsigned long long)0x7f7fffffu << 32) | 0x7fffffffu;  // (FLT_MAX, INT_MAX)
// Broadcast from a wave-uniform lane: v_readlane_b32 (VALU -> SGPR), not the LDS crossbar a generic __shfl uses.
__device__ __forceinline__ int bcast_i(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ float bcast_f(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
__device__ __forceinline__ unsigned long long bcast_k(unsigned long long v, int lane) {
    return ((unsigned long long)(unsigned)bcast_i((int)(v >> 32), lane) << 32) | (unsigned)bcast_i((int)(unsigned)v, lane);
}
// Lane i receives lane i-1's value (lane 0 keeps its own): one DPP move, wave_shr:1 (gfx9 DPP control 0x138).
__device__ __forceinline__ int shift_up1_i(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ float shift_up1_f(float v) { return __int_as_float(shift_up1_i(__float_as_int(v))); }
__device__ __forceinline__ unsigned long long shift_up1_k(unsigned long long v) {
    return ((unsigned long long)(unsigned)shift_up1_i((int)(v >> 32)) << 32) | (unsigned)shift_up1_i((int)(unsigned)v);
}

// The value of lane (i ^ STRIDE), for one 32-bit register. Strides below 16 stay inside a row of 16 lanes: DPP moves on the
// VALU (quad_perm for 1 and 2, row_half_mirror + reversed quads for 4, row_ror:8 for 8) instead of ds_bpermute through the
// LDS crossbar — 18 of the 21 stages of the 64-lane network, 54 of its 63 permutes.
template <int STRIDE>
__device__ __forceinline__ int xor_lane(int v) {
    // (mov_dpp, not update_dpp(v, v, ...): every lane is written, and without an `old` operand the compiler neither copies the
    // register first nor is kept from folding the move into the instruction that consumes it)
    if (STRIDE == 1) return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true);   // quad_perm:[1,0,3,2]
    if (STRIDE == 2) return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, true);   // quad_perm:[2,3,0,1]
    if (STRIDE == 4) {
        const int m = __builtin_amdgcn_mov_dpp(v, 0x141, 0xf, 0xf, true);        // row_half_mirror: i -> i ^ 7
        return __builtin_amdgcn_mov_dpp(m, 0x1B, 0xf, 0xf, true);                // quad_perm:[3,2,1,0]: ^ 3  => i ^ 4
    }
    if (STRIDE == 8) return __builtin_amdgcn_mov_dpp(v, 0x128, 0xf, 0xf, true);  // row_ror:8: i -> i ^ 8
    return __shfl_xor(v, STRIDE, 64);
}

// Lanes that keep the SMALLER key of their pair in the stage (SIZE, STRIDE) of the ascending 64-lane network: a compile-time
// constant, so the per-lane direction logic of a stage is two scalar instructions on the compare's lane mask instead of
// half a dozen vector ones (the sort is VALU-issue-bound: 818 -> ~630 VALU wave instructions per query).
template <int SIZE, int STRIDE>
constexpr unsigned long long keeps_smaller_mask() {
    unsigned long long m = 0;
    for (int lane = 0; lane < 64; ++lane)
        if (((lane & STRIDE) == 0) == ((lane & SIZE) == 0)) m |= 1ull << lane;
    return m;
}
template <int SIZE, int STRIDE>
__device__ __forceinline__ void bitonic_step(Cand& v, unsigned lane) {
    (void)lane;
    Cand o;
    o.key = ((unsigned long long)(unsigned)xor_lane<STRIDE>((int)(v.key >> 32)) << 32) | (unsigned)xor_lane<STRIDE>((int)(unsigned)v.key);
    o.pos = xor_lane<STRIDE>(v.pos);
    // keys of different lanes differ (distance, index) except between two empty slots, where either choice is the same:
    // "take the partner" = partner smaller on the lanes that keep the smaller key, partner not smaller on the others
    constexpr unsigned long long kSmaller = keeps_smaller_mask<SIZE, STRIDE>();
    const unsigned long long lt = __ballot(o.key < v.key);
    const bool take = __builtin_amdgcn_inverse_ballot_w64(~(lt ^ kSmaller));
    v.key = take ? o.key : v.key;
    v.pos = take ? o.pos : v.pos;
}
template <int SIZE, int STRIDE>
__device__ __forceinline__ void bitonic_merge(Cand& v, unsigned lane) {
    bitonic_step<SIZE, STRIDE>(v, lane);
    if constexpr (STRIDE > 1) bitonic_merge<SIZE, STRIDE / 2>(v, lane);
}
__device__ __forceinline__ Cand bitonic_sort64(Cand v, unsigned lane) {
    bitonic_merge<2, 1>(v, lane);
    bitonic_merge<4, 2>(v, lane);
    bitonic_merge<8, 4>(v, lane);
    bitonic_merge<16, 8>(v, lane);
    bitonic_merge<32, 16>(v, lane);
    bitonic_merge<64, 32>(v, lane);
    return v;
}

// One chunk of candidates (one per lane) against the sorted top-k in `best` (lane i = i-th best): every candidate nearer
// than the current k-th is inserted at its rank, the entries behind it move up by one lane.
__device__ __forceinline__ void insert_candidates(const Cand& c, Cand& best, unsigned long long& kth, int k,
                                                  unsigned long long kmask, unsigned lane) {
    unsigned long long m = __ballot(c.key < kth);
    while (m) {
        const int L = __builtin_ctzll(m);
        m &= m - 1;
        const unsigned long long vk = bcast_k(c.key, L);
        if (!(vk < kth)) continue;  // the k-th entry moved since the ballot
        const int vp = bcast_i(c.pos, L);
        const int rank = __builtin_popcountll(__ballot(best.key < vk) & kmask);  // entries that stay in front of the newcomer
        const unsigned long long uk = shift_up1_k(best.key);
        const int up = shift_up1_i(best.pos);
        if ((int)lane == rank) { best.key = vk; best.pos = vp; }
        else if ((int)lane > rank) { best.key = uk; best.pos = up; }
        kth = bcast_k(best.key, k - 1);
    }
}

}  // namespace sp
