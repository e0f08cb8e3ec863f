// Direct exchange of the per-iteration row between the ranks of a sharded alignment, without a collective launch
// (comm.hip creates / connects it, registration.hip's kernels use it).
//
// Every rank owns a slot buffer [2 (parity of the alignment)][2 (parity of the iteration)][world][32] of 8-byte granules
// {float value, uint32 tag} in uncached device memory and has its peers' buffers mapped (hipIpc handles exchanged once, through
// the caller's own channel). The last-arriving workgroup of a streaming launch stores its 32-float row, granule by granule,
// into slot [epoch & 1][k & 1][rank] of EVERY rank's buffer (its own too) with system-scope 8-byte stores; the next launch's
// prologue (the finish launch after the last iteration) polls [epoch & 1][k & 1][r][*] for r = 0 .. world - 1 until every tag
// equals this iteration's sequence number, sums the rows in rank order (the same order on every rank: identical bits, identical
// pose) and solves. A granule carries its own tag, so no ordering between stores is needed; the tag is unique per (alignment,
// iteration), so a stale slot never matches.
// Why four slots are enough. Inside one alignment two parities of k suffice: a rank can only write iteration k + 2 after every
// rank has consumed iteration k (it needs all rows of k + 1, and a rank writes row k + 1 only after its solve of iteration k).
// Across alignments the parity of k alone is NOT enough (round 3 had only that): launch 0 of alignment N + 1 stores row 0
// without waiting for anything, and when alignment N ended on an even iteration a slower peer may not have polled that very
// slot for this rank's last row of N yet — its tag would never match and the peer would run into its time limit. With the
// parity of the alignment in the slot index, rows of N + 1 never land on rows of N; and a rank can only start alignment N + 2
// after it has received row 0 of N + 1 from every peer, which a peer sends only after it has consumed all rows of N.
// The poll is bounded by a wall-clock budget: a row that does not arrive sets an error flag in the alignment's state block
// (sp_gicp_align_status) instead of hanging the queue.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

constexpr int kXchgRow = 32;        // floats per row (kFanRow)
constexpr int kXchgMaxWorld = 8;    // ranks of one node
constexpr int kXchgSlots = 4;       // [alignment parity][iteration parity]
__host__ __device__ inline unsigned xchg_slot(unsigned epoch, int k) { return (epoch & 1u) * 2u + ((unsigned)k & 1u); }

struct sp_xchg {
    int rank = 0, world = 1;
    unsigned long long* local = nullptr;          // this rank's slot buffer: kXchgSlots * world * kXchgRow granules
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
