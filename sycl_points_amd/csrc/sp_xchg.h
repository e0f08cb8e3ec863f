// Direct exchange of the per-iteration row between the ranks of a sharded alignment, without a collective launch
// (comm.hip creates / connects it, registration.hip's kernels use it).
//
// Every rank owns a slot buffer [2 (parity of the iteration)][world][32] of 8-byte granules {float value, uint32 tag} in
// uncached device memory and has its peers' buffers mapped (hipIpc handles exchanged once, through the caller's own channel).
// The last-arriving workgroup of a streaming launch stores its 32-float row, granule by granule, into slot [k & 1][rank] of
// EVERY rank's buffer (its own too) with system-scope 8-byte stores; the one-workgroup solve launch that follows polls
// [k & 1][r][*] for r = 0 .. world - 1 until every tag equals this iteration's sequence number, sums the rows in rank order
// (the same order on every rank: identical bits, identical pose) and solves. A granule carries its own tag, so no ordering
// between stores is needed; the tag is unique per (alignment, iteration), so a stale slot never matches. Two parities are
// enough: a rank can only write iteration k + 2 after every rank has consumed iteration k (it needs all rows of k + 1, and
// a rank writes row k + 1 only after its solve of iteration k). The poll is bounded by a wall-clock budget: a row that does
// not arrive sets an error flag in the alignment's state block (sp_gicp_align_status) instead of hanging the queue.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

constexpr int kXchgRow = 32;        // floats per row (kFanRow)
constexpr int kXchgMaxWorld = 8;    // ranks of one node

struct sp_xchg {
    int rank = 0, world = 1;
    unsigned long long* local = nullptr;          // this rank's slot buffer: 2 * world * kXchgRow granules
    unsigned long long** peers_dev = nullptr;     // device array [world]: every rank's slot buffer as mapped here
    void* peers_host[kXchgMaxWorld] = {};         // the same pointers (for hipIpcCloseMemHandle)
    bool connected = false;
    unsigned* epoch_dev = nullptr;                // device word: alignments run so far + 1 (the tag's high part). On the device
                                                  // so that a captured hipGraph of an alignment stays valid when replayed: the
                                                  // last launch of every alignment increments it
    unsigned timeout_ms = 2000;
    hipIpcMemHandle_t handle;
};

// what the kernels need of it
struct XchgArgs {
    unsigned long long* const* peers;  // nullptr: no direct exchange
    unsigned long long* local;
    int rank, world;
    const unsigned* epoch;             // tag of iteration k's rows = *epoch * 256 + k + 1
    unsigned long long budget;         // wall_clock64 ticks (100 MHz) a poll may take
};
