"""How a sharded run (N > 1 ranks) picks the carrier of the per-iteration row, and how it falls back — the logic of
bench.py's --exchange auto, kept free of any device call so that it can be rehearsed with world_size-2 `gloo` ranks on a CPU
(tests/test_distributed_cpu.py): the first run on a real multi-GPU node must not die in Python that never ran.

Ladder:  direct stores into the peers' IPC-mapped slot buffers (sp_gicp_align_direct)
                                     -> 128-byte all-reduce by the library's own RCCL communicator (sp_gicp_align_sharded)
                                     -> the same row through torch.distributed.
Every decision is taken by ALL ranks together (an all-reduce MIN of each rank's local outcome): a carrier is used only when it
works on every rank, and every rank leaves the ladder on the same rung.
"""
import time


def _all_agree(dist, torch, dev, local_ok, world):
    """True iff `local_ok` holds on every rank."""
    if world <= 1:
        return bool(local_ok)
    flag = torch.tensor([1.0 if local_ok else 0.0], device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return flag.item() > 0.5


def open_carriers(dist, torch, dev, rank, world, want, make_direct, make_comm, log=print):
    """Creates the carriers `want` ('auto' | 'direct' | 'rccl-row' | 'torch-row' | 'torch-rows') asks for.
    make_direct() / make_comm() return a handle or raise. Returns (xchg, comm)."""
    xchg = comm = None
    if want in ("auto", "direct"):
        try:
            xchg = make_direct()
        except Exception as e:  # the peers' buffers could not be mapped here
            if rank == 0:
                log(f"bench: direct exchange unavailable ({e!r})")
        if not _all_agree(dist, torch, dev, xchg is not None, world):  # all ranks or none
            xchg = None
        if xchg is None and want == "direct":
            raise RuntimeError("--exchange direct: the peers' slot buffers could not be mapped")
    if want in ("auto", "rccl-row"):
        try:
            comm = make_comm()
        except Exception as e:  # no RCCL behind the C ABI on this machine: torch.distributed moves the row instead
            if rank == 0:
                log(f"bench: sp_comm unavailable ({e!r}); exchanging the row through torch.distributed")
        if not _all_agree(dist, torch, dev, comm is not None, world):
            comm = None
    return xchg, comm


def verify_and_fall_back(dist, torch, dev, rank, world, want, xchg, comm, try_alignment, log=print):
    """One eager alignment per rung must land on the ground truth on EVERY rank before anything is timed.
    try_alignment(xchg, comm) -> this rank's max abs pose error (inf / nan: failed; it may raise).
    Returns (xchg, comm, fallback_note or None, legs) where legs = [{'carrier', 'ok', 'pose_err', 'seconds'}, ...]."""
    legs = []

    def leg(x, c):
        t0 = time.perf_counter()
        try:
            err = float(try_alignment(x, c))
        except Exception as e:  # a carrier that throws on one rank must not leave the others in a collective
            if rank == 0:
                log(f"bench: alignment over {carrier_name(x, c)} raised {e!r}")
            err = float("inf")
        local = err == err and err < 1e-3
        ok = _all_agree(dist, torch, dev, local, world)
        legs.append({"carrier": carrier_name(x, c), "ok": ok, "pose_err": err, "seconds": time.perf_counter() - t0})
        return ok, err

    note = None
    good, err = leg(xchg, comm)
    if not good and xchg is not None:
        if want == "direct":
            raise RuntimeError(f"--exchange direct: pose error {err:.3g} on an eager alignment")
        note = f"direct exchange gave pose error {err:.3g} on an eager alignment; using the RCCL row"
        xchg = None
        good, err = leg(xchg, comm)
    if not good and comm is not None:
        note = f"sp_comm exchange gave pose error {err:.3g} on an eager alignment; using torch.distributed"
        comm = None
        good, err = leg(xchg, comm)
    if not good:
        raise RuntimeError(f"sharded alignment does not reach the ground truth (max abs pose error {err:.3g})")
    if rank == 0 and note:
        log("bench: " + note)
    return xchg, comm, note, legs


def verified_carriers(dist, torch, dev, rank, world, want, xchg, comm, try_alignment, log=print):
    """Every carrier that is open AND lands one eager alignment on the ground truth on every rank, in ladder order — what
    bench.py times for N > 1 (`exchange_legs`: the direct stores and the RCCL row side by side; torch.distributed only
    stands in when neither works, or when it was asked for). try_alignment as in verify_and_fall_back.
    Returns (carriers, legs): carriers = [(name, xchg or None, comm or None), ...], never empty (raises otherwise)."""
    candidates = []
    if xchg is not None:
        candidates.append((xchg, None))
    if comm is not None:
        candidates.append((None, comm))
    legs, good = [], []

    def leg(x, c):
        t0 = time.perf_counter()
        try:
            err = float(try_alignment(x, c))
        except Exception as e:
            if rank == 0:
                log(f"bench: alignment over {carrier_name(x, c)} raised {e!r}")
            err = float("inf")
        ok = _all_agree(dist, torch, dev, err == err and err < 1e-3, world)
        legs.append({"carrier": carrier_name(x, c), "ok": ok, "pose_err": err, "seconds": time.perf_counter() - t0})
        if ok:
            good.append((carrier_name(x, c), x, c))
        return err

    last = float("nan")
    for x, c in candidates:
        last = leg(x, c)
    if want == "direct" and not any(g[0] == "direct" for g in good):
        raise RuntimeError(f"--exchange direct: pose error {last:.3g} on an eager alignment")
    if not good:  # neither of the library's own carriers: the row travels through torch.distributed
        last = leg(None, None)
    if not good:
        raise RuntimeError(f"sharded alignment does not reach the ground truth (max abs pose error {last:.3g})")
    return good, legs


def carrier_name(xchg, comm):
    return "direct" if xchg is not None else ("rccl-row" if comm is not None else "torch.distributed")
