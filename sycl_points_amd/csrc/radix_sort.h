// Stable LSD radix sort of (u32 key, u32 value) pairs on the low `bits` bits of the key (radix_sort.hip). Internal.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace sp {
size_t radix_sort_u32_workspace_bytes(size_t n);
// Sorts by the key bits [first_bit, bits) in ceil((bits - first_bit) / 8) passes that ping-pong between the (a) and (b)
// buffers; both are overwritten. *result_in_b tells where the sorted pairs are. Only enqueues on `st`. Key bits at and above
// `bits` are ignored, as are those below `first_bit` (a stable sort by the remaining ones).
int radix_sort_pairs_u32(uint32_t* keys_a, uint32_t* keys_b, uint32_t* vals_a, uint32_t* vals_b, size_t n, unsigned bits,
                         void* workspace, size_t workspace_bytes, bool* result_in_b, hipStream_t st, unsigned first_bit = 0,
                         bool first_hist_ready = false);
// The first pass's per-tile digit histograms may come from the kernel that MADE the keys (it has them in registers: no launch
// and no read of the keys for the count). radix_first_pass(n, bits) says what that kernel has to leave at the START of the
// workspace: hist[digit * tiles + tile] = keys of tile `tile` (keys [tile * tile_keys, (tile + 1) * tile_keys)) whose low
// `digit_bits` bits (masked to the sort's `bits`) equal `digit`; then call radix_sort_pairs_u32(..., first_hist_ready = true).
struct RadixFirstPass {
    unsigned tiles, tile_keys, digit_bits, mask;
};
RadixFirstPass radix_first_pass(size_t n, unsigned bits);
// The same for 64-bit keys, on the low `bits` <= 64 key bits.
size_t radix_sort_u64_workspace_bytes(size_t n);
int radix_sort_pairs_u64(uint64_t* keys_a, uint64_t* keys_b, uint32_t* vals_a, uint32_t* vals_b, size_t n, unsigned bits,
                         void* workspace, size_t workspace_bytes, bool* result_in_b, hipStream_t st);
// Exclusive prefix sum of n u32 values whose total is below 2^30 (flags, counts), in != out or in == out; *total_out (device,
// optional) receives the total. One zeroing launch + one scan launch; only enqueues.
size_t exclusive_scan_u32_workspace_bytes(size_t n);
int exclusive_scan_u32(const uint32_t* in, uint32_t* out, size_t n, uint32_t* total_out, void* workspace, size_t workspace_bytes,
                       hipStream_t st);
// Stable compaction by flags in one launch (+ one zeroing launch): rows_out[a][j] = rows[a][i] for the j-th element i whose flag is
// 1, for up to 16 attribute arrays with rows of `words` 32-bit words. flags: u8 (1 = keep) — or, with box_pts != nullptr, the box
// filter's test on those points (box_filter_operator.hpp:31-45), the flags it makes written to flags_out (optional).
// new_indices_out (optional): the output position of every kept element, -1 for the others. *n_out_dev: the number kept.
struct CompactArrays {
    const uint32_t* src[16];
    uint32_t* dst[16];
    unsigned words[16];
    int n_arrays;
};
size_t compact_fused_workspace_bytes(size_t n);
int compact_rows_fused(const CompactArrays& arrays, size_t n, const uint8_t* flags, const float4* box_pts, float box_min,
                       float box_max, uint8_t* flags_out, int32_t* new_indices_out, uint32_t* n_out_dev, void* workspace,
                       size_t workspace_bytes, hipStream_t st);
}  // namespace sp
