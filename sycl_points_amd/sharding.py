"""Multi-GPU sharding rules of the GICP path: contiguous source tiles, replicated target, one
all-reduce of the 192-byte linear system per iteration. Pure host logic, importable without a GPU."""
import numpy as np

LIN_FLOATS = 48          # sizeof(sp_linearized) / 4
IDX_ERROR, IDX_INLIER_U32, IDX_LO, IDX_HI = 42, 43, 44, 45


def shard_range(n_total, rank, world):
    """Contiguous index range [lo, hi) of `rank`: tiles differ by at most one point."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_indices(n_total, rank, world, chunk=1024):
    """Source indices of `rank`: chunks of `chunk` consecutive points of the (spatially ordered) source, dealt round-robin.

    Contiguous tiles (shard_range) of a spatially ordered source are slabs of space: the slabs far from the rotation
    centre keep searching for more iterations, and every all-reduce waits for the slowest rank. Dealing out chunks gives
    every rank a uniform sample of the whole cloud — the same mix of converged and still-moving regions in every
    iteration — while the points inside a chunk (a whole workgroup's worth) stay neighbours, which the search needs.
    chunk <= 0 or >= n_total / world: the contiguous tile."""
    if chunk <= 0 or chunk * world >= n_total:
        lo, hi = shard_range(n_total, rank, world)
        return np.arange(lo, hi, dtype=np.int64)
    starts = np.arange(rank * chunk, n_total, world * chunk, dtype=np.int64)
    idx = (starts[:, None] + np.arange(chunk, dtype=np.int64)[None, :]).reshape(-1)
    return idx[idx < n_total]


def default_chunk(n_total, world):
    """Chunk size of shard_indices for a cell-ordered source of n_total points on `world` ranks. From 4 ranks on: half a rank's
    tile, so that rank r holds slabs r and r + world of 2 * world — every pair lies as far from the middle of the cloud, taken
    together, as any other (the slabs that move furthest under a rotation keep searching longest), and a rank's searches touch two
    slabs of the replicated target instead of all of it (per-rank step of the emulated 8-rank run 48.2 against 50.4 us, slowest
    rank; 4 ranks 40.2 against 42.6; profiles/r05_zzz_shard_chunk_by_rank.txt). Two ranks: 1024-point chunks (35.7 against 36.7)."""
    if world >= 4 and n_total >= 4 * world:
        return max(1, (n_total // world) // 2)
    return 1024


def split_count(count):
    """Inlier count as two floats that stay exact under a float sum over ranks (count = hi * 4096 + lo)."""
    return float(count & 4095), float(count >> 12)


def fold_count(lo, hi):
    return int(np.float32(hi)) * 4096 + int(np.float32(lo))  # integer fold, as sp_gn_update does


def pack_linearized(H, b, error, inlier):
    """Host image of sp_linearized as 48 floats (what final_reduce_kernel writes on the device)."""
    buf = np.zeros(LIN_FLOATS, np.float32)
    buf[:36] = np.asarray(H, np.float32).reshape(-1)
    buf[36:42] = np.asarray(b, np.float32)
    buf[IDX_ERROR] = np.float32(error)
    buf[IDX_INLIER_U32:IDX_INLIER_U32 + 1].view(np.uint32)[0] = np.uint32(inlier)
    buf[IDX_LO], buf[IDX_HI] = split_count(int(inlier))
    return buf


def unpack_linearized(buf):
    buf = np.asarray(buf, np.float32)
    return {"H": buf[:36].reshape(6, 6).copy(), "b": buf[36:42].copy(), "error": float(buf[IDX_ERROR]),
            "inlier": fold_count(buf[IDX_LO], buf[IDX_HI])}
