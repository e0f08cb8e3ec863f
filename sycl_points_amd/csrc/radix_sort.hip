// Stable LSD radix sort of (u32 key, u32 value) pairs for gfx950 — the sort behind the voxel keys (voxel.hip), the cell
// ids of the grid build (grid.hip) and the per-alignment cell order of an unsorted source (registration.hip).
//
// Why not the library sort: at the sizes of this path (1 M pairs, 22-24 key bits) rocPRIM's Onesweep spends 27 us per
// 8-bit pass plus a histogram kernel and five buffer fills — 110-125 us per sort (profiles/r01_l_kernel_stats_bench_stages.csv)
// for 16 MB of traffic per pass. Here a pass is three small launches and no fill:
//   rs_count_kernel    per tile of 2048 keys: digit histogram in LDS -> tile_hist[digit][tile]
//   rs_scan_kernel     one workgroup per digit: the exclusive scan over the tiles of its row, and the row's total
//   rs_scatter_kernel  per tile: every key gets its stable rank among the tile's keys of the same digit (wave match-any by
//                      eight ballots, per-wave counters in LDS, wave order = memory order), the tile is reordered by digit in
//                      LDS, and runs of equal digit leave as contiguous, coalesced stores
// No spinning, no cross-workgroup ordering inside a launch (nothing can hang), deterministic, stable.
//
// Round 3, measured and not kept (profiles/r03_stage_timings_lookback_sort_not_kept.json): one launch per pass with the tile
// offsets by decoupled look-back (Onesweep's scheme: ticketed tiles, one AGGREGATE / PREFIX word per tile and digit, each
// digit's lane walking back over its predecessors). At these sizes every tile of a pass is resident at once, all of them
// publish their counts at the same moment and the walk of tile t is ~t/2 dependent L2 round trips long: a pass took ~35 us
// against 21 us for the three launches (voxel downsampling 0.202 ms against 0.161, grid build 0.38 against 0.31).
// The same kernels are templated on the key type: 64-bit keys (the uncompressed voxel keys) sort in ceil(bits / 8) passes of
// the same three launches, which replaced the library's merge sort on that path (21 launches, 200 us per 1M keys).
#include "radix_sort.h"

#include "sp_common.h"
#include "sp_internal.h"
#include "sp_lookback.h"
#include "sp_math.h"

namespace sp {
namespace {

constexpr int kRsThreads = 512;             // 8 waves
constexpr int kRsWaves = kRsThreads / 64;
constexpr int kRsItems = 4;                 // keys per lane (tile = 2048 keys: two workgroups per CU at 1 M keys)
constexpr int kRsTile = kRsThreads * kRsItems;
constexpr int kRsMaxBins = 512;             // digits of 8 bits, or of 9 where that saves a pass (18 key bits: 2 passes, not 3)
constexpr int kRsScanThreads = 256;

// tile_hist is stored [digit][tile]: the scan over the tiles of a digit reads one contiguous row.
template <typename KEY, int kRsBins>
__global__ __launch_bounds__(kRsThreads) void rs_count_kernel(const KEY* __restrict__ keys, unsigned n, unsigned shift,
                                                              unsigned mask, unsigned tiles, unsigned* __restrict__ tile_hist) {
    __shared__ unsigned h[kRsWaves][kRsBins];  // one histogram per wave: eight times less contention on a popular digit
    for (unsigned i = threadIdx.x; i < kRsWaves * kRsBins; i += kRsThreads) (&h[0][0])[i] = 0u;
    __syncthreads();
    const unsigned base = blockIdx.x * kRsTile, w = threadIdx.x >> 6;
    KEY k[kRsItems];
#pragma unroll
    for (int c = 0; c < kRsItems; ++c) {
        const unsigned e = base + c * kRsThreads + threadIdx.x;
        k[c] = e < n ? keys[e] : (KEY)0;
    }
#pragma unroll
    for (int c = 0; c < kRsItems; ++c) {
        const unsigned e = base + c * kRsThreads + threadIdx.x;
        const bool valid = e < n;
        const unsigned d = (unsigned)(k[c] >> shift) & mask;
        const unsigned d0 = (unsigned)__builtin_amdgcn_readfirstlane((int)d);
        const unsigned long long vm = __ballot(valid);
        if (__ballot(valid && d != d0) == 0ull) {  // the high digits of a compact key range: the whole wave agrees
            if ((threadIdx.x & 63u) == 0u && vm) h[w][d0] += (unsigned)__builtin_popcountll(vm);
        } else if (valid) {
            atomicAdd(&h[w][d], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < kRsBins) {
        unsigned v = 0;
#pragma unroll
        for (int i = 0; i < kRsWaves; ++i) v += h[i][threadIdx.x];
        tile_hist[(size_t)threadIdx.x * tiles + blockIdx.x] = v;
    }
}

// Exclusive scan of one value per lane over a workgroup of NW waves (wave scan by shuffles + the wave totals in LDS);
// *total_out = the sum over the workgroup.
template <int NW>
__device__ __forceinline__ unsigned block_excl_scan(unsigned v, unsigned* wave_tot /* [NW] */, unsigned* total_out) {
    const unsigned lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    unsigned inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned o = __shfl_up(inc, off, 64);
        if ((int)lane >= off) inc += o;
    }
    if (lane == 63u) wave_tot[w] = inc;
    __syncthreads();
    unsigned before = 0, all = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const unsigned t = wave_tot[i];
        before += (unsigned)i < w ? t : 0u;
        all += t;
    }
    __syncthreads();
    if (total_out) *total_out = all;
    return before + inc - v;
}

// One workgroup per digit: the exclusive scan over the tiles of the digit's row, and the row's total (the scatter kernel
// turns the 256 totals into the digits' starting positions itself: no global atomics, nothing to clear between passes).
__global__ __launch_bounds__(kRsScanThreads) void rs_scan_kernel(unsigned* __restrict__ tile_hist, unsigned tiles,
                                                                 unsigned* __restrict__ digit_total) {
    __shared__ unsigned wave_tot[kRsScanThreads / 64];
    const unsigned d = blockIdx.x;
    unsigned carry = 0;
    unsigned* const row = tile_hist + (size_t)d * tiles;
    for (unsigned t0 = 0; t0 < tiles; t0 += kRsScanThreads) {
        const unsigned t = t0 + threadIdx.x;
        const unsigned c = t < tiles ? row[t] : 0u;
        unsigned chunk;
        const unsigned ex = block_excl_scan<kRsScanThreads / 64>(c, wave_tot, &chunk);
        if (t < tiles) row[t] = carry + ex;
        carry += chunk;
    }
    if (threadIdx.x == 0) digit_total[d] = carry;
}

// FOLD (passes of up to kRsFoldTiles tiles: 131 k keys): there is no scan launch — tile_off holds the RAW tile histograms and
// every workgroup sums, for each digit, the counts of the tiles before its own and of all tiles itself (a row of at most 64
// words per lane, L2-resident): at these sizes a launch costs more than reading 64 KB again in every workgroup.
constexpr unsigned kRsFoldTiles = 64;
template <typename KEY, int kRsBins, bool FOLD = false>
__global__ __launch_bounds__(kRsThreads) void rs_scatter_kernel(const KEY* __restrict__ kin,
                                                                const uint32_t* __restrict__ vin,
                                                                KEY* __restrict__ kout, uint32_t* __restrict__ vout,
                                                                unsigned n, unsigned shift, unsigned mask, unsigned tiles,
                                                                const unsigned* __restrict__ tile_off,
                                                                const unsigned* __restrict__ digit_total) {
    __shared__ unsigned cnt[kRsWaves][kRsBins];  // per wave: keys of each digit seen so far -> then the wave's offset in the digit
    __shared__ unsigned tstart[kRsBins];         // tile-local position of the digit's first key
    __shared__ unsigned dbase[kRsBins];          // global position of the digit's first key of this tile
    __shared__ unsigned wave_tot[kRsWaves];
    __shared__ KEY lk[kRsTile];
    __shared__ uint32_t lv[kRsTile];
    const unsigned tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    const unsigned base = blockIdx.x * kRsTile;
    const unsigned m = min((unsigned)kRsTile, n - base);
    for (unsigned i = tid; i < kRsWaves * kRsBins; i += kRsThreads) (&cnt[0][0])[i] = 0u;
    if constexpr (FOLD) {
        unsigned tot = 0, bef = 0;
        if (tid < kRsBins) {
            const unsigned* const row = tile_off + (size_t)tid * tiles;
#pragma unroll 8
            for (unsigned t = 0; t < tiles; ++t) {
                const unsigned c = row[t];
                tot += c;
                bef += t < blockIdx.x ? c : 0u;
            }
        }
        const unsigned before = block_excl_scan<kRsWaves>(tot, wave_tot, nullptr);
        if (tid < kRsBins) dbase[tid] = before + bef;
    } else {   // where this tile's first key of digit `tid` goes: the keys of smaller digits + the same digit in earlier tiles
        const unsigned before = block_excl_scan<kRsWaves>(tid < kRsBins ? digit_total[tid] : 0u, wave_tot, nullptr);
        if (tid < kRsBins) dbase[tid] = before + tile_off[(size_t)tid * tiles + blockIdx.x];
    }
    __syncthreads();
    // wave w owns the keys [w * 512, (w + 1) * 512) of the tile, 64 at a time: memory order = (wave, chunk, lane)
    KEY k[kRsItems];
    uint32_t v[kRsItems];
    unsigned rank[kRsItems];
    const unsigned long long lt = (1ull << lane) - 1ull;
    volatile unsigned* const my_cnt = cnt[w];
#pragma unroll
    for (int c = 0; c < kRsItems; ++c) {
        const unsigned e = w * (kRsTile / kRsWaves) + c * 64 + lane;
        k[c] = e < m ? kin[base + e] : (KEY)0;
        v[c] = e < m ? vin[base + e] : 0u;
    }
#pragma unroll
    for (int c = 0; c < kRsItems; ++c) {
        const unsigned e = w * (kRsTile / kRsWaves) + c * 64 + lane;
        const bool valid = e < m;
        const unsigned d = (unsigned)(k[c] >> shift) & mask;
        unsigned long long peers = __ballot(valid);  // lanes of this chunk holding the same digit
        constexpr int kDigitBits = kRsBins == 512 ? 9 : 8;
        static_assert(kRsBins == 256 || kRsBins == 512, "digits of 8 or 9 bits");
        static_assert(kRsBins <= kRsThreads, "one lane per digit in the offset steps");
#pragma unroll
        for (int b = 0; b < kDigitBits; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long mm = __ballot(bit);
            peers &= bit ? mm : ~mm;
        }
        const unsigned old = my_cnt[d];  // read by every lane BEFORE the digit's first lane bumps it (LDS ops of a wave are in order)
        const unsigned lower = (unsigned)__builtin_popcountll(peers & lt);
        rank[c] = old + lower;
        if (valid && lower == 0u) my_cnt[d] = old + (unsigned)__builtin_popcountll(peers);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    {   // digit tid: totals of the waves -> each wave's offset inside the digit, and the digit's place in the tile
        unsigned tot = 0;
        if (tid < kRsBins) {
#pragma unroll
            for (int i = 0; i < kRsWaves; ++i) { const unsigned c = cnt[i][tid]; cnt[i][tid] = tot; tot += c; }
        }
        const unsigned ex = block_excl_scan<kRsWaves>(tot, wave_tot, nullptr);  // (lanes >= 256 contribute 0; two barriers inside)
        if (tid < kRsBins) tstart[tid] = ex;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < kRsItems; ++c) {
        const unsigned e = w * (kRsTile / kRsWaves) + c * 64 + lane;
        if (e < m) {
            const unsigned d = (unsigned)(k[c] >> shift) & mask;
            const unsigned pos = tstart[d] + cnt[w][d] + rank[c];
            lk[pos] = k[c];
            lv[pos] = v[c];
        }
    }
    __syncthreads();
    for (unsigned j = tid; j < m; j += kRsThreads) {
        const KEY kk = lk[j];
        const unsigned d = (unsigned)(kk >> shift) & mask;
        const unsigned dst = dbase[d] + (j - tstart[d]);
        kout[dst] = kk;
        vout[dst] = lv[j];
    }
}


}  // namespace

// tile histograms [digits][tiles] + the digit totals, for the wider (9-bit) digit
size_t radix_sort_u32_workspace_bytes(size_t n) {
    return ((size_t)div_up(n ? n : 1, (size_t)kRsTile) * kRsMaxBins + kRsMaxBins) * sizeof(unsigned);
}
size_t radix_sort_u64_workspace_bytes(size_t n) { return radix_sort_u32_workspace_bytes(n); }

namespace {
template <typename KEY>
int radix_sort_pairs(KEY* keys_a, KEY* keys_b, uint32_t* vals_a, uint32_t* vals_b, size_t n, unsigned bits, void* workspace,
                     bool* result_in_b, hipStream_t st, unsigned first_bit, bool first_hist_ready = false) {
    const unsigned tiles = div_up(n, (size_t)kRsTile);
    // 9-bit digits where they save a pass (18 key bits — the cell ids of a 6-points-per-cell grid on 1 M points, a dense voxel
    // box — sort in 2 passes instead of 3; 24 bits stay at 3 passes of 8)
    const unsigned span = bits - first_bit;
    const unsigned digit = (span + 8u) / 9u < (span + 7u) / 8u ? 9u : 8u;
    const unsigned bins = 1u << digit;
    unsigned* const tile_hist = static_cast<unsigned*>(workspace);
    unsigned* const digit_total = tile_hist + (size_t)tiles * bins;
    KEY *kin = keys_a, *kout = keys_b;
    uint32_t *vin = vals_a, *vout = vals_b;
    bool in_b = false;
    for (unsigned shift = first_bit; shift < bits; shift += digit) {
        const unsigned width = bits - shift < digit ? bits - shift : digit;
        const unsigned mask = (1u << width) - 1u;
        const bool counted = first_hist_ready && shift == first_bit;  // (the keys' producer left this pass's tile histograms)
        if (tiles <= kRsFoldTiles && digit == 9u) {
            if (!counted) rs_count_kernel<KEY, 512><<<tiles, kRsThreads, 0, st>>>(kin, (unsigned)n, shift, mask, tiles, tile_hist);
            rs_scatter_kernel<KEY, 512, true><<<tiles, kRsThreads, 0, st>>>(kin, vin, kout, vout, (unsigned)n, shift, mask, tiles, tile_hist,
                                                                            digit_total);
        } else if (tiles <= kRsFoldTiles) {
            if (!counted) rs_count_kernel<KEY, 256><<<tiles, kRsThreads, 0, st>>>(kin, (unsigned)n, shift, mask, tiles, tile_hist);
            rs_scatter_kernel<KEY, 256, true><<<tiles, kRsThreads, 0, st>>>(kin, vin, kout, vout, (unsigned)n, shift, mask, tiles, tile_hist,
                                                                            digit_total);
        } else if (digit == 9u) {
            if (!counted) rs_count_kernel<KEY, 512><<<tiles, kRsThreads, 0, st>>>(kin, (unsigned)n, shift, mask, tiles, tile_hist);
            rs_scan_kernel<<<512, kRsScanThreads, 0, st>>>(tile_hist, tiles, digit_total);
            rs_scatter_kernel<KEY, 512><<<tiles, kRsThreads, 0, st>>>(kin, vin, kout, vout, (unsigned)n, shift, mask, tiles, tile_hist,
                                                                      digit_total);
        } else {
            if (!counted) rs_count_kernel<KEY, 256><<<tiles, kRsThreads, 0, st>>>(kin, (unsigned)n, shift, mask, tiles, tile_hist);
            rs_scan_kernel<<<256, kRsScanThreads, 0, st>>>(tile_hist, tiles, digit_total);
            rs_scatter_kernel<KEY, 256><<<tiles, kRsThreads, 0, st>>>(kin, vin, kout, vout, (unsigned)n, shift, mask, tiles, tile_hist,
                                                                      digit_total);
        }
        KEY* t = kin; kin = kout; kout = t;
        uint32_t* tv = vin; vin = vout; vout = tv;
        in_b = !in_b;
    }
    *result_in_b = in_b;
    return launch_status();
}
}  // namespace

int radix_sort_pairs_u32(uint32_t* keys_a, uint32_t* keys_b, uint32_t* vals_a, uint32_t* vals_b, size_t n, unsigned bits,
                         void* workspace, size_t workspace_bytes, bool* result_in_b, hipStream_t st, unsigned first_bit,
                         bool first_hist_ready) {
    *result_in_b = false;
    if (n == 0 || bits <= first_bit) return SP_OK;
    if (n >= (1ull << 32) - kRsTile || bits > 32 || !workspace || workspace_bytes < radix_sort_u32_workspace_bytes(n))
        return SP_ERR_INVALID_ARGUMENT;
    return radix_sort_pairs<uint32_t>(keys_a, keys_b, vals_a, vals_b, n, bits, workspace, result_in_b, st, first_bit, first_hist_ready);
}
RadixFirstPass radix_first_pass(size_t n, unsigned bits) {  // (mirrors radix_sort_pairs: first_bit = 0)
    const unsigned digit = (bits + 8u) / 9u < (bits + 7u) / 8u ? 9u : 8u;
    const unsigned width = bits < digit ? bits : digit;
    return RadixFirstPass{(unsigned)div_up(n, (size_t)kRsTile), (unsigned)kRsTile, digit, (1u << width) - 1u};
}

// 64-bit keys (the uncompressed voxel keys, voxel_constants.hpp:36-62): the same passes over the low `bits` <= 64 key bits.
int radix_sort_pairs_u64(uint64_t* keys_a, uint64_t* keys_b, uint32_t* vals_a, uint32_t* vals_b, size_t n, unsigned bits,
                         void* workspace, size_t workspace_bytes, bool* result_in_b, hipStream_t st) {
    *result_in_b = false;
    if (n == 0 || bits == 0) return SP_OK;
    if (n >= (1ull << 32) - kRsTile || bits > 64 || !workspace || workspace_bytes < radix_sort_u64_workspace_bytes(n))
        return SP_ERR_INVALID_ARGUMENT;
    return radix_sort_pairs<uint64_t>(keys_a, keys_b, vals_a, vals_b, n, bits, workspace, result_in_b, st, 0);
}

// ------------------------------------------------------------------ exclusive scan (u32), one launch
// Tiles of 2048 values, tile prefixes by decoupled look-back with a wave-wide window (sp_lookback.h): even with every tile
// resident at once the walk is a handful of steps (it is the per-digit, one-lane walk of a radix pass that does not pay,
// see above).
namespace {
__global__ __launch_bounds__(kRsThreads) void scan_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, unsigned n,
                                                          unsigned* __restrict__ state, unsigned* __restrict__ ticket,
                                                          unsigned* __restrict__ error, uint32_t* __restrict__ total_out) {
    __shared__ unsigned wave_tot[kRsWaves];
    __shared__ unsigned s_tile, s_excl;
    const unsigned tid = threadIdx.x, w = tid >> 6;
    if (tid == 0) s_tile = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const unsigned tile = s_tile;
    const unsigned base = tile * kRsTile + tid * kRsItems;  // blocked: a lane owns kRsItems consecutive values
    uint32_t v[kRsItems];
    unsigned sum = 0;
#pragma unroll
    for (int c = 0; c < kRsItems; ++c) {
        v[c] = base + c < n ? in[base + c] : 0u;
        sum += v[c];
    }
    unsigned tile_total;
    const unsigned before = block_excl_scan<kRsWaves>(sum, wave_tot, &tile_total);
    if (w == 0) {
        const unsigned excl = lookback_exclusive(state, tile, tile_total, error);
        if (tid == 0) {
            s_excl = excl;
            if (total_out && (size_t)(tile + 1) * kRsTile >= n) *total_out = excl + tile_total;  // the last tile
        }
    }
    __syncthreads();
    unsigned run = s_excl + before;
#pragma unroll
    for (int c = 0; c < kRsItems; ++c) {
        if (base + c < n) out[base + c] = run;
        run += v[c];
    }
}
}  // namespace

size_t exclusive_scan_u32_workspace_bytes(size_t n) { return ((size_t)div_up(n ? n : 1, (size_t)kRsTile) + 64) * sizeof(unsigned); }

int exclusive_scan_u32(const uint32_t* in, uint32_t* out, size_t n, uint32_t* total_out, void* workspace, size_t workspace_bytes,
                       hipStream_t st) {
    if (n == 0) return total_out ? zero_async(total_out, sizeof(uint32_t), st) : SP_OK;
    // (the state word of a tile packs a 2-bit flag with a 30-bit running sum: the total must stay below 2^30, which n < 2^30
    // guarantees for the flag arrays this is used on)
    if (n >= (1ull << 30) || !workspace || workspace_bytes < exclusive_scan_u32_workspace_bytes(n)) return SP_ERR_INVALID_ARGUMENT;
    const unsigned tiles = div_up(n, (size_t)kRsTile);
    unsigned* const ws = static_cast<unsigned*>(workspace);  // [0] ticket, [64 ...] one state word per tile
    if (zero_async(ws, (tiles + 64) * sizeof(unsigned), st) != SP_OK) return SP_ERR_HIP;
    scan_kernel<<<tiles, kRsThreads, 0, st>>>(in, out, (unsigned)n, ws + 64, ws, device_error_word(), total_out);
    return launch_status();
}

// ------------------------------------------------------------------ stable compaction by flags, one launch
// Flags (or the box filter's test that makes them), their exclusive scan and the move of the kept rows of every attribute in ONE
// kernel: tiles of 2048 elements handed out by ticket, a lane owns four consecutive elements, the tile's offset comes by the
// decoupled look-back of the scan above. The chain it replaces was flags | widen to u32 | zero | scan | one compaction launch per
// attribute: at the sizes of a scan (70 k points) every one of them is launch latency.
namespace {
template <bool BOX>
__global__ __launch_bounds__(kRsThreads) void compact_fused_kernel(CompactArrays A, const uint8_t* __restrict__ flags,
                                                                   const float4* __restrict__ box_pts, float mn, float mx,
                                                                   uint8_t* __restrict__ flags_out, unsigned n,
                                                                   int32_t* __restrict__ new_idx, uint32_t* __restrict__ n_out,
                                                                   unsigned* __restrict__ state, unsigned* __restrict__ ticket,
                                                                   unsigned* __restrict__ error) {
    __shared__ unsigned wave_tot[kRsWaves];
    __shared__ unsigned s_tile, s_excl;
    const unsigned tid = threadIdx.x, w = tid >> 6;
    if (tid == 0) s_tile = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const unsigned tile = s_tile;
    const unsigned base = tile * kRsTile + tid * kRsItems;
    unsigned f[kRsItems];
    // BOX: the four points stay in registers for the move below — one read of the cloud, which may be the caller's pinned HOST
    // copy (read over PCIe in place: no upload, no device copy of the unfiltered cloud at all)
    float4 pt[BOX ? kRsItems : 1];
    if constexpr (BOX) {
        // read as whole 1 KB rows per wave instruction (every 64-byte line of the — possibly host — memory is asked for once;
        // a lane reading its own four consecutive points asks for each line four times: 34 GB/s over PCIe), regrouped through LDS
        __shared__ float4 lp[kRsTile];
#pragma unroll
        for (int c = 0; c < kRsItems; ++c) {
            const unsigned l = c * kRsThreads + tid, e = tile * kRsTile + l;
            if (e < n) lp[l] = box_pts[e];
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < kRsItems; ++c) pt[c] = lp[tid * kRsItems + c];  // (past n: never looked at)
    }
    unsigned sum = 0;
#pragma unroll
    for (int c = 0; c < kRsItems; ++c) {
        const unsigned e = base + c;
        f[c] = 0u;
        if (e < n) {
            if constexpr (BOX) {  // BoxFilter (preprocess_operator/box_filter_operator.hpp:31-45), as box_filter_kernel (voxel.hip)
                const float4 p = pt[c];
                unsigned keep = 1u;
                if (!(isfinite(p.x) && isfinite(p.y) && isfinite(p.z) && isfinite(p.w))) {
                    keep = 0u;
                } else {
                    const float linf = sycl_max(fabsf(p.x), sycl_max(fabsf(p.y), fabsf(p.z)));
                    if (linf < mn || linf > mx) keep = 0u;
                }
                f[c] = keep;
                if (flags_out) flags_out[e] = (uint8_t)keep;
            } else {
                f[c] = flags[e] == 1 ? 1u : 0u;  // INCLUDE_FLAG == 1 (filter_by_flags.hpp:12)
            }
        }
        sum += f[c];
    }
    unsigned tile_total;
    const unsigned before = block_excl_scan<kRsWaves>(sum, wave_tot, &tile_total);
    if (w == 0) {
        const unsigned excl = lookback_exclusive(state, tile, tile_total, error);
        if (tid == 0) {
            s_excl = excl;
            // (the last tile; a system-scope store: the caller may have handed a host-mapped word it is spinning on)
            if ((size_t)(tile + 1) * kRsTile >= n)
                __hip_atomic_store(n_out, excl + tile_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    __syncthreads();
    unsigned run = s_excl + before;
#pragma unroll
    for (int c = 0; c < kRsItems; ++c) {
        const unsigned e = base + c;
        if (e < n) {
            if (new_idx) new_idx[e] = f[c] ? (int32_t)run : -1;
            if (f[c]) {
                for (int a = 0; a < A.n_arrays; ++a) {
                    const unsigned words = A.words[a];
                    if constexpr (BOX) {
                        if (reinterpret_cast<const void*>(A.src[a]) == reinterpret_cast<const void*>(box_pts) && words == 4u &&
                            (reinterpret_cast<uintptr_t>(A.dst[a]) & 15u) == 0u) {
                            reinterpret_cast<float4*>(A.dst[a])[run] = pt[c];
                            continue;
                        }
                    }
                    if ((words & 3u) == 0u && ((reinterpret_cast<uintptr_t>(A.src[a]) | reinterpret_cast<uintptr_t>(A.dst[a])) & 15u) == 0u) {  // rows of whole, aligned 16-byte quads (points, covariances, normals)
                        const uint4* const src = reinterpret_cast<const uint4*>(A.src[a]) + (size_t)e * (words / 4);
                        uint4* const dst = reinterpret_cast<uint4*>(A.dst[a]) + (size_t)run * (words / 4);
                        for (unsigned q = 0; q < words / 4; ++q) dst[q] = src[q];
                    } else {
                        for (unsigned d = 0; d < words; ++d) A.dst[a][(size_t)run * words + d] = A.src[a][(size_t)e * words + d];
                    }
                }
            }
            run += f[c];
        }
    }
}
}  // namespace

size_t compact_fused_workspace_bytes(size_t n) { return exclusive_scan_u32_workspace_bytes(n); }

int compact_rows_fused(const CompactArrays& arrays, size_t n, const uint8_t* flags, const float4* box_pts, float box_min,
                       float box_max, uint8_t* flags_out, int32_t* new_indices_out, uint32_t* n_out_dev, void* workspace,
                       size_t workspace_bytes, hipStream_t st) {
    if (n == 0) return zero_async(n_out_dev, sizeof(uint32_t), st);
    if (n >= (1ull << 30) || !workspace || workspace_bytes < compact_fused_workspace_bytes(n)) return SP_ERR_INVALID_ARGUMENT;
    const unsigned tiles = div_up(n, (size_t)kRsTile);
    unsigned* const ws = static_cast<unsigned*>(workspace);  // [0] ticket, [64 ...] one state word per tile
    if (zero_async(ws, (tiles + 64) * sizeof(unsigned), st) != SP_OK) return SP_ERR_HIP;
    if (box_pts)
        compact_fused_kernel<true><<<tiles, kRsThreads, 0, st>>>(arrays, nullptr, box_pts, box_min, box_max, flags_out, (unsigned)n,
                                                                  new_indices_out, n_out_dev, ws + 64, ws, device_error_word());
    else
        compact_fused_kernel<false><<<tiles, kRsThreads, 0, st>>>(arrays, flags, nullptr, 0.0f, 0.0f, nullptr, (unsigned)n,
                                                                   new_indices_out, n_out_dev, ws + 64, ws, device_error_word());
    return launch_status();
}

}  // namespace sp

extern "C" size_t sp_internal_radix_sort_workspace_bytes(size_t n) { return sp::radix_sort_u32_workspace_bytes(n); }
extern "C" int sp_internal_radix_sort_u32(uint32_t* keys_a, uint32_t* keys_b, uint32_t* vals_a, uint32_t* vals_b, size_t n,
                                          unsigned bits, void* workspace, size_t workspace_bytes, int* result_in_b_out,
                                          void* stream) {
    bool in_b = false;
    const int rc = sp::radix_sort_pairs_u32(keys_a, keys_b, vals_a, vals_b, n, bits, workspace, workspace_bytes, &in_b,
                                            sp::as_stream(stream));
    if (result_in_b_out) *result_in_b_out = in_b ? 1 : 0;
    return rc;
}
