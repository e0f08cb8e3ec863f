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
                         void* workspace, size_t workspace_bytes, bool* result_in_b, hipStream_t st, unsigned first_bit = 0);
// The same for 64-bit keys, on the low `bits` <= 64 key bits.
size_t radix_sort_u64_workspace_bytes(size_t n);
int radix_sort_pairs_u64(uint64_t* keys_a, uint64_t* keys_b, uint32_t* vals_a, uint32_t* vals_b, size_t n, unsigned bits,
                         void* workspace, size_t workspace_bytes, bool* result_in_b, hipStream_t st);
// Exclusive prefix sum of n u32 values whose total is below 2^30 (flags, counts), in != out or in == out; *total_out (device,
// optional) receives the total. One zeroing launch + one scan launch; only enqueues.
size_t exclusive_scan_u32_workspace_bytes(size_t n);
int exclusive_scan_u32(const uint32_t* in, uint32_t* out, size_t n, uint32_t* total_out, void* workspace, size_t workspace_bytes,
                       hipStream_t st);
}  // namespace sp
