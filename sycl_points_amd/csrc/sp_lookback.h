// Decoupled look-back (Merrill & Garland) with a wave-wide window, shared by the scan (radix_sort.hip) and the voxel
// aggregation (voxel.hip).
//
// Tiles are handed out by an atomic ticket, so tile t only ever waits for tiles < t, which hold earlier tickets and are
// therefore already running: the wait ends. A tile publishes its own sum (AGGREGATE) and, once it has looked back, its
// inclusive prefix (PREFIX); value and state share ONE 32-bit word (value < 2^30), stored / polled with relaxed agent-scope
// atomics, so no fence is needed. The calling wave looks back 64 predecessors per step — lane l polls tile t - 1 - l — and
// stops at the nearest PREFIX: even with every tile resident at once the walk is a handful of steps. The poll is bounded: a
// word that never arrives raises *error instead of hanging the queue.
#pragma once
#include "sp_common.h"

namespace sp {

constexpr unsigned kLbAggregate = 1u << 30, kLbPrefix = 2u << 30, kLbValue = (1u << 30) - 1u;
constexpr unsigned kLbSpinLimit = 1u << 24;

// Called by ONE whole wave (all 64 lanes) of the workgroup that owns `tile`, after the state words were zeroed: publishes
// tile_total and returns (to every lane) the sum of the totals of all earlier tiles.
__device__ __forceinline__ unsigned lookback_exclusive(unsigned* __restrict__ state, unsigned tile, unsigned tile_total,
                                                       unsigned* __restrict__ error) {
    const unsigned lane = threadIdx.x & 63u;
    if (tile == 0u) {
        if (lane == 0) __hip_atomic_store(state, kLbPrefix | tile_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return 0u;
    }
    if (lane == 0) __hip_atomic_store(state + tile, kLbAggregate | tile_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned excl = 0;
    for (int t = (int)tile - 1; t >= 0; t -= 64) {
        const int idx = t - (int)lane;
        unsigned word = kLbPrefix;  // before tile 0: an empty prefix
        if (idx >= 0) {
            unsigned spins = 0;
            do {
                word = __hip_atomic_load(state + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (word >> 30) break;
                __builtin_amdgcn_s_sleep(1);
            } while (++spins < kLbSpinLimit);
            if ((word >> 30) == 0u) {  // (never: the tile holds an earlier ticket) — reported through launch_status()
                if (error) __hip_atomic_store(error, kDevErrLookback, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                word = kLbPrefix;
            }
        }
        const unsigned long long pm = __ballot((word >> 30) == 2u);  // nearest PREFIX in the window (lane 0 = nearest tile)
        const unsigned first = pm ? (unsigned)__builtin_ctzll(pm) : 63u;
        excl += wave_sum_u32(lane <= first ? (word & kLbValue) : 0u);
        if (pm) break;
    }
    if (lane == 0) __hip_atomic_store(state + tile, kLbPrefix | (excl + tile_total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return excl;
}

}  // namespace sp
