#!/usr/bin/env python3
"""GICP hot-path benchmark (BASELINE.json: "GICP iterations/sec & correspondences/sec at 1M pts").

A step = one GICP iteration over the rank's source shard: nearest neighbour (k=1, query transformed by the current
pose) -> linearise + reduce -> [all-reduce of the linear system over ranks] -> 6x6 solve + pose update, all on the
device with inputs resident in HBM. Steps run in alignments of 20 iterations from the identity initial guess
(BASELINE config 4: GICP, GN lambda=1, max_corr 2.0, robust NONE, convergence criteria 0).
INSIDE the timed region, once per alignment: the source-side preparation (plane-regularised source covariances; with
--source-order random also the sort of the source by target cell). Untimed set-up ("preprocessing timed
separately", as KDTree::build and covariance estimation are separate stages in the reference's own flow): the clouds,
their k=20 covariances, the target's NN structure (grid) and everything that depends on the target alone — its
plane-regularised covariance rows and reuse certificates (sp_gicp_target_create) — and, with the default
--source-order grid, storing the source in cell order (what voxel downsampling hands to align()).
N = 1: 1M-vs-1M clouds (config 4). N > 1: config 5 generalised — N x 1M source points sharded 1M per GPU,
target (N x 1M points, same density) replicated on every GPU: weak scaling.

The block of --steps steps is timed --repeats times (each block bracketed by a barrier + synchronize on both sides);
`ms_per_step` / `value` come from the MEDIAN block, so a 20-step run is not one 0.7 ms sample.

Prints ONE JSON line on rank 0 (contract in the task statement), with `roofline` for the dominant kernel,
`launch_classes` (the launches of one alignment split into searching / steady, each with its algorithmic and
measured-traffic fraction of the HBM peak), `until_converged` (an alignment with the reference's default criteria
1e-3: what a caller of align() gets) and `cpu_baseline` (the CPU oracle timed on the same workload, rank 0, N = 1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ITERS_PER_ALIGN = 20
PER_GPU_POINTS = 1_000_000
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md chip table)
BYTES_NN = 24          # in-loop NN k=1 lower bound (16 B query + 8 B result)
BYTES_K11 = 168        # K11 per source point at the API layouts
BYTES_ITER = 192       # one GICP iteration (NN 24 B + K11 168 B) per correspondence


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=31,
                    help="the block of --steps steps is timed this many times; the median block is reported")
    ap.add_argument("--points", type=int, default=PER_GPU_POINTS, help="source points per GPU (default: the config)")
    ap.add_argument("--path", choices=["fused", "generic"], default="fused",
                    help="fused: GridKNN + prepared covariances, NN+K11 in one kernel; generic: KNNBase search + K11")
    ap.add_argument("--nn", choices=["grid", "kdtree"], default="grid", help="KNNBase used by --path generic")
    ap.add_argument("--reg", choices=["gicp", "p2d"], default="gicp",
                    help="factor: gicp = RegType::GICP (the reference's default and what config 1 uses); p2d = "
                         "RegType::POINT_TO_DISTRIBUTION (factor.hpp:311-373), the reading of BASELINE config 4's "
                         "'GICP (point-to-distribution)' — both run on the prepared / fused path")
    ap.add_argument("--ppc", type=float, default=0.5, help="GridKNN points per cell for the in-loop k=1 search")
    ap.add_argument("--source-order", choices=["grid", "random"], default="grid",
                    help="grid: the source is stored in the cell order of a grid on itself (what voxel downsampling "
                         "yields; done in set-up). random: as generated; every alignment then sorts it by target cell")
    ap.add_argument("--force-sharded", action="store_true",
                    help="rehearsal: run the multi-GPU code path (RCCL all-reduce of the 192-byte system every iteration) "
                         "with a world of one rank")
    ap.add_argument("--exchange", choices=["auto", "direct", "rccl-row", "torch-row", "torch-rows"], default="auto",
                    help="sharded runs: what moves the linear system between the ranks every iteration. direct: every rank "
                         "stores its 128-byte row straight into the peers' IPC-mapped slot buffers, no collective launch "
                         "(sp_gicp_align_direct); auto: direct when it maps and reproduces the ground truth on an eager "
                         "alignment, else rccl-row, else torch-row. rccl-row: the 128-byte "
                         "fan-in row through the library's own RCCL communicator (sp_gicp_align_sharded, the C ABI path); "
                         "torch-row: the same row through torch.distributed; torch-rows: round 1's 32 KB of partial rows")
    ap.add_argument("--no-graph", action="store_true",
                    help="sharded runs: issue every launch / collective from the host instead of replaying one "
                         "captured hipGraph per alignment")
    ap.add_argument("--emulate-rank", type=int, default=0, help="with --emulate-world: which rank's tile to run")
    ap.add_argument("--shard-chunk", type=int, default=-1,
                    help="N > 1: the (spatially ordered) source is dealt to the ranks in chunks of this many consecutive "
                         "points, round-robin (every rank sees the same mix of converged and still-moving regions); 0: one "
                         "contiguous tile per rank (a spatial slab); -1 (default): sharding.default_chunk — half a rank's tile "
                         "from 4 ranks on (rank r holds slabs r and r + N of 2 N: as far from the rotation centre together as "
                         "any other pair, and only two slabs of the target to keep in cache), 1024 points below")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="rehearsal on one GPU: build the clouds of an N-rank run (N x --points, config 5 density) and time "
                         "rank 0's tile against the full target, without the collective")
    ap.add_argument("--internal", action="append", default=[], metavar="NAME=VALUE",
                    help="measurement only: a per-handle switch of csrc/sp_internal.h on the prepared source, e.g. reuse=0 "
                         "(every launch searches every point); such a line is not a benchmark result")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--timed-only", action="store_true",
                    help="only what the timed region launches (alignments of 20 iterations, criteria 0): no per-launch profile, "
                         "no until-converged / reuse-off / stage legs — the run profiles/collect_traffic.py wraps, so that every "
                         "dispatch of the dominant kernel it sees belongs to the timed mix of launches")
    ap.add_argument("--no-stages", action="store_true",
                    help="skip the `stages` block (BASELINE configs 2 and 3 and the pre-loop of config 4, timed after the GICP "
                         "region) and the reuse-off comparison")
    ap.add_argument("--cpu-sample", type=int, default=1_000_000, help="points in the CPU-baseline workload")
    ap.add_argument("--no-example", action="store_true",
                    help="skip stages.example_registration_config1 (a child process: a profiler wrapped around this command "
                         "would trace it too)")
    return ap.parse_args()


def _free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv, child_cmd=None, env=None, stdout=None):
    """`python bench.py --gpus N` without a launcher around it: THIS process — which has made no GPU call and never will —
    starts N children of the same command line with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set (what
    torch.distributed.run would have set; under torchrun WORLD_SIZE is there already and this function is not reached), relays
    rank 0's stdout (the ONE JSON line) to its own, lets the children's stderr through, and returns 0 only if every child did.
    A child that dies takes the others with it (they would wait in a collective for ever). Never an exec of a process that has
    touched the GPU: children are started, this process only waits."""
    import signal
    import subprocess

    cmd = list(child_cmd) if child_cmd is not None else [sys.executable, os.path.abspath(__file__)]
    base = dict(os.environ if env is None else env)
    base.setdefault("MASTER_ADDR", "127.0.0.1")
    base.setdefault("MASTER_PORT", str(_free_port()))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    base["WORLD_SIZE"] = base["LOCAL_WORLD_SIZE"] = str(n)
    out = sys.stdout if stdout is None else stdout
    procs = []
    for r in range(n):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(cmd + list(argv), env=e, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                      start_new_session=True))
    rc = 0
    chunks = []
    import threading

    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()  # rank 0's pipe is drained while waiting (its line is small, but a full pipe must never block it)
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    print(f"bench: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr, flush=True)
                    for q in pending:
                        try:
                            os.killpg(procs[q].pid, signal.SIGTERM)  # the exact process groups started above
                        except ProcessLookupError:
                            pass
            if pending:
                time.sleep(0.05)
        reader.join(timeout=10.0)
    finally:
        for q in procs:
            if q.poll() is None:
                try:
                    os.killpg(q.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
    line0 = b"".join(c for c in chunks if c)
    out.write(line0.decode(errors="replace"))
    out.flush()
    return rc


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    # The contract is ONE JSON line on stdout. Libraries below print there too (RCCL writes a version banner to stdout when a
    # communicator is created): everything written to fd 1 from here on goes to stderr, and the line is written to the real
    # stdout at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # SP_BENCH_SHARE_GPU=1 (rehearsal on the one-GPU box, never the driver's run): every rank uses cuda:0 and gloo carries
    # torch.distributed, so that this file's N > 1 branches run on the real kernels; the line then says "rehearsal" in `data`
    share_gpu = os.environ.get("SP_BENCH_SHARE_GPU") == "1" and world > 1
    if share_gpu:
        # the direct stores need every rank's launch resident at once (one GPU runs them one after the other: rows time out)
        # and gloo cannot be captured into a hipGraph: the rehearsal carries the row through torch.distributed, eagerly
        local_rank, args.exchange, args.no_graph = 0, "torch-row", True
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or args.force_sharded:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    import sycl_points_amd.api as sp
    from sycl_points_amd import _lib
    from sycl_points_amd.sharding import default_chunk, shard_indices
    from sycl_points_amd.synthetic import gicp_pair

    n_gpu = args.points
    shards = args.emulate_world if (args.emulate_world > 1 and world == 1) else world
    n_total = n_gpu * shards
    if args.shard_chunk < 0:
        args.shard_chunk = default_chunk(n_total, shards)
    rng_range = 10.0 * (n_total / 1e6) ** (1.0 / 3.0)  # config 4 density at every size (R=20 at 8M)

    # ---- untimed set-up: clouds, k=20 covariances (fused self-kNN on a grid), NN structure on the target
    t_setup = time.time()
    src, tgt, T_gt = gicp_pair(n_total, rng_range)
    to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    Tg = sp.PointCloudShared(to_dev(tgt), device=dev)
    g6 = sp.GridKNN.build(Tg.points, points_per_cell=6.0)
    preloop = None
    if world > 1:
        # pre-loop sharded by query: each rank computes the k = 20 covariances of 1/N of the target's grid
        # positions, one all-gather shares the rows (torch.distributed here: the library's communicator is created below)
        Tg.covs = g6.covariances_sharded(20, rank, world, lambda send, recv: dist.all_gather_into_tensor(recv, send))
    else:
        Tg.covs = g6.self_knn(20, want_knn=False, want_covs=True)[1]
        if shards > 1:  # rehearsal: what ONE rank of the N-rank run would spend on its share of the pre-loop
            preloop = preloop_times(sp, _lib, torch, g6, Tg.covs, args.emulate_rank, shards)
    del g6
    tile = torch.from_numpy(shard_indices(n_total, rank if shards == world else args.emulate_rank, shards,
                                          args.shard_chunk)).to(dev)
    S_all = to_dev(src)
    if args.source_order == "grid":
        # The reference's pipeline hands align() a voxel-downsampled scan, i.e. a cloud sorted by voxel key
        # (voxel_downsampling.hpp:146-288). The synthetic cloud is in random order, so it is put into the cell order of
        # a grid built on itself here, in the untimed pre-processing next to its covariances; alignments then need no sort.
        S_all = S_all[sp.GridKNN.build(S_all, points_per_cell=1.0).order()].contiguous()
    covs_all = sp.GridKNN.build(S_all, points_per_cell=6.0).self_knn(20, want_knn=False, want_covs=True)[1]
    S = sp.PointCloudShared(S_all[tile].contiguous(), covs=covs_all[tile].contiguous(), device=dev)
    del S_all, covs_all
    n_local = S.size()
    grid = sp.GridKNN.build(Tg.points, points_per_cell=args.ppc) if (args.path == "fused" or args.nn == "grid") else None
    knn = grid if (args.path == "fused" or args.nn == "grid") else sp.KDTree.build(tgt)
    REG_TYPE = {"gicp": "GICP", "p2d": "POINT_TO_DISTRIBUTION"}[args.reg]
    prep = sp.PreparedTarget(grid, Tg.covs, reg_type=REG_TYPE) if args.path == "fused" else None
    torch.cuda.synchronize()
    t_setup = time.time() - t_setup

    SORT_MODE = "presorted" if args.source_order == "grid" else True
    params = sp.RegistrationParams(reg_type=REG_TYPE, optimization_method="GN", max_iterations=ITERS_PER_ALIGN,
                                   criteria_translation=0.0, criteria_rotation=0.0)
    reg = sp.Registration(params)
    for kv in args.internal:
        name, value = kv.split("=")
        reg._set_source_option(name, int(value))
    T_dev = torch.zeros(16, dtype=torch.float32, device=dev)
    T_ident = torch.eye(4, dtype=torch.float32, device=dev).reshape(-1).contiguous()
    delta = torch.zeros(8, dtype=torch.float32, device=dev)
    group = dist.group.WORLD if (world > 1 or args.force_sharded) else None
    use_graph = group is not None and not args.no_graph and os.environ.get("SP_BENCH_GRAPH", "1") == "1"
    comm = None
    xchg = None
    elog = lambda m: print(m, file=sys.stderr, flush=True)  # noqa: E731
    if group is not None and args.path == "fused":
        from sycl_points_amd.exchange_select import open_carriers
        xchg, comm = open_carriers(dist, torch, dev, rank, world, args.exchange,
                                   lambda: sp.Exchange.from_process_group(group),
                                   lambda: sp.Communicator.from_process_group(group), elog)
    exchange = "rows" if args.exchange == "torch-rows" else "row"

    cur = {"xchg": xchg, "comm": comm}  # the carrier the loop below runs on (N > 1: every verified one is timed in turn)

    def align_chunk(iters, first):
        if args.path == "fused":
            reg.align_fused_loop(S, prep, iterations=iters, group=group, T_dev=T_dev, delta_dev=delta, prepare=first,
                                 sort_by_cell=SORT_MODE, graph=use_graph and cur["xchg"] is None, comm=cur["comm"],
                                 exchange=exchange, xchg=cur["xchg"])
        else:
            reg.align_device_loop(S, Tg, knn, iterations=iters, group=group, T_dev=T_dev, delta_dev=delta)

    def run_steps(k):
        done = 0
        while done < k:
            first = done % ITERS_PER_ALIGN == 0
            if first:
                T_dev.copy_(T_ident)  # a new alignment starts from the identity initial guess
            chunk = min(ITERS_PER_ALIGN - done % ITERS_PER_ALIGN, k - done)
            align_chunk(chunk, first)
            done += chunk

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- N > 1: the exchange is checked BEFORE anything is timed. One eager alignment must land on the ground truth on every
    # rank; if the library's own RCCL communicator does not deliver that (it cannot be rehearsed with more than one rank on
    # the one-GPU test box), the row travels through torch.distributed instead and the line says so.
    exchange_fallback = None
    exchange_legs = None
    carriers = [(None, xchg, comm)]
    if group is not None and world > 1 and args.path == "fused":
        from sycl_points_amd.exchange_select import verified_carriers

        def try_alignment(x, c):
            T_dev.copy_(T_ident)
            reg.align_fused_loop(S, prep, iterations=ITERS_PER_ALIGN, group=group, T_dev=T_dev, delta_dev=delta, prepare=True,
                                 sort_by_cell=SORT_MODE, graph=False, comm=c, exchange=exchange, xchg=x)
            torch.cuda.synchronize()
            err = np.abs(reg.T_from_device(T_dev) - T_gt).max()
            if x is not None:
                try:
                    reg.direct_status()
                except sp.SpError:
                    err = float("inf")  # a peer's row did not arrive within the bound
            return err

        carriers, exchange_legs = verified_carriers(dist, torch, dev, rank, world, args.exchange, xchg, comm, try_alignment, elog)
        bad = [l["carrier"] for l in exchange_legs if not l["ok"]]
        if bad:
            exchange_fallback = f"{', '.join(bad)}: no eager alignment on the ground truth on every rank; not timed"

    def timed_blocks():
        if use_graph and cur["xchg"] is None:
            # set-up, like building the grid: capture the hipGraph of every chunk length the warm-up and the timed region
            # will use (first call of a shape runs eagerly, the second is captured), so no capture falls into the timed region
            for k in (args.warmup, args.steps):
                for chunk in {ITERS_PER_ALIGN if k >= ITERS_PER_ALIGN else 0, k % ITERS_PER_ALIGN} - {0}:
                    poses = []
                    for _ in range(3):  # eager, capture + first replay, replay
                        T_dev.copy_(T_ident)
                        align_chunk(chunk, True)
                        torch.cuda.synchronize()
                        poses.append(T_dev.clone())
                    # a replayed alignment must reproduce the eagerly launched one (to rounding: a collective may pick another
                    # summation order inside a graph); otherwise stay on eager launches
                    same = all(bool(torch.isfinite(p).all()) and float((p - poses[0]).abs().max()) < 1e-6 for p in poses[1:])
                    ok = torch.ones(1, device=dev) if same else torch.zeros(1, device=dev)
                    if world > 1:
                        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                    if ok.item() < 0.5:
                        for key in list(getattr(reg, "_loop_graphs", {})):
                            reg._loop_graphs[key] = False
                        if rank == 0:
                            print("bench: hipGraph replay did not reproduce the eager alignment; using per-call launches",
                                  file=sys.stderr, flush=True)
            fence()
        run_steps(args.warmup)
        fence()
        blocks = []
        for _ in range(max(1, args.repeats)):  # every block: EXACTLY --steps steps between two barrier + synchronize fences
            t0 = time.perf_counter()
            run_steps(args.steps)
            fence()
            blocks.append(time.perf_counter() - t0)
        if world > 1:  # a block takes as long as its slowest rank
            tmax = torch.tensor(blocks, dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            blocks = [float(x) for x in tmax.tolist()]
        return blocks

    # N > 1: the direct stores AND the RCCL all-reduce north_star names are both timed (every carrier that verified), the
    # line's `value` is the faster one and `exchange_legs` holds both. N = 1: one pass.
    timed = []
    for name, x, c in carriers:
        cur["xchg"], cur["comm"] = x, c
        timed.append((name, x, c, timed_blocks()))
    best = min(range(len(timed)), key=lambda i: float(np.median(timed[i][3])))
    _, xchg, comm, blocks = timed[best]
    cur["xchg"], cur["comm"] = xchg, comm
    elapsed = float(np.median(blocks))
    timed_legs = None
    if world > 1:
        timed_legs = [{"carrier": nm, "ms_per_step": 1e3 * float(np.median(b)) / args.steps,
                       "min_ms_per_step": 1e3 * min(b) / args.steps, "max_ms_per_step": 1e3 * max(b) / args.steps,
                       "reported": i == best} for i, (nm, _, _, b) in enumerate(timed)]
    rccl_ranks = None
    for _, _, c, _ in timed:
        if c is not None:
            rccl_ranks = int(_lib.lib().sp_comm_world(c._h))  # what the communicator itself says

    # ---- correctness of what was timed: pose after a full alignment vs the ground truth used to make the data
    T_dev.copy_(T_ident)
    align_chunk(ITERS_PER_ALIGN, True)
    torch.cuda.synchronize()
    T_final = reg.T_from_device(T_dev)
    lin = reg._read_lin(reg._lin)
    pose_err = float(np.abs(T_final - T_gt).max())

    # ---- per-kernel durations over one alignment, by HIP events on the launch stream (rank 0's numbers are reported)
    kern = kernel_times(sp, _lib, torch, args, reg, S, Tg, knn, prep, T_dev, T_ident, delta, n_local, SORT_MODE)
    launches = classes = converged = None
    if args.path == "fused" and group is None and not args.timed_only:
        launches, classes = launch_profile(sp, _lib, torch, reg, prep, T_dev, T_ident, delta, n_local)
        converged = until_converged(sp, torch, S, prep, T_dev, T_ident, delta, n_local, SORT_MODE, REG_TYPE, internal=args.internal)

    stages = reuse0 = p2d = hard = lm = None
    if world == 1 and shards == 1 and not args.no_stages and not args.timed_only:
        if args.path == "fused":
            reuse0 = reuse_off_comparison(sp, torch, args, S, prep, T_dev, T_ident, delta, SORT_MODE, REG_TYPE, n_local)
            if args.reg == "gicp":
                p2d = p2d_block(sp, torch, S, Tg, grid, T_dev, T_ident, delta, n_local, SORT_MODE, T_gt)
                hard = hard_init_block(sp, _lib, torch, S, prep, n_local, SORT_MODE, T_gt)
                lm = lm_block(sp, torch, S, prep, n_local, SORT_MODE, T_gt)
        stages = stage_block(sp, _lib, torch, cpu=not args.no_cpu_baseline)
        raw = raw_scan_block(sp, torch)
        if raw is not None:
            stages["raw_scans_full_resolution_gn"] = raw
        if not args.no_example:
            stages["example_registration_config1"] = example_block(cpu=not args.no_cpu_baseline)
    graphs_live = use_graph and any(not isinstance(v, (str, bool)) for v in getattr(reg, "_loop_graphs", {}).values())
    if rank == 0:
        dom = max((k for k in kern if kern[k].get("per_iteration", True)), key=lambda k: kern[k]["ms"])
        out = {
            "metric": "gicp_correspondences_per_sec",
            "value": (n_gpu if shards != world else n_total) * args.steps / elapsed,  # (rehearsal: one tile only)
            "unit": "correspondences/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "timed_blocks": {"n": len(blocks), "statistic": "median", "min_ms_per_step": 1e3 * min(blocks) / args.steps,
                             "max_ms_per_step": 1e3 * max(blocks) / args.steps},
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if not share_gpu else "synthetic; REHEARSAL: all ranks on one GPU over gloo, not a measurement",
            "config": {"workload": f"{'GICP' if args.reg == 'gicp' else 'point-to-distribution ICP'} {n_total}-vs-{n_total} "
                                   f"uniform-random clouds (BASELINE config "
                                   f"{'4' if world == 1 else '5 generalised'}), k=20 covariances, GN lambda=1, "
                                   f"max_corr 2.0, robust NONE, {ITERS_PER_ALIGN} iterations per alignment",
                       "source_points_per_gpu": n_gpu, "target_points": n_total, "path": args.path, "reg_type": REG_TYPE,
                       "nn": "grid(k=1)" if (args.path == "fused" or args.nn == "grid") else "kdtree(k=1)",
                       "target_preparation": "once, in set-up with its NN structure (grid build, plane-regularised covariances, "
                                             "safe radii); per alignment only the source is prepared",
                       "source_order": ("cell order of a grid on the source (set-up), no per-alignment sort"
                                        if args.source_order == "grid" else "random; sorted by target cell in every alignment"),
                       "sharding": ((f"source dealt to the ranks in chunks of {args.shard_chunk} consecutive points"
                                     if args.shard_chunk > 0 else "source in contiguous tiles") +
                                    ", target replicated" if shards > 1 else "none"),
                       "exchange": (None if group is None else
                                    "128-byte row per iteration stored directly into the peers' IPC-mapped slot buffers "
                                    "(sp_gicp_align_direct): no collective launch" if xchg is not None else
                                    ("128-byte fan-in row per iteration, " if exchange == "row" else "32 KB of partial rows per iteration, ") +
                                    ("sp_gicp_align_sharded over the library's RCCL communicator" if comm is not None
                                     else "torch.distributed all-reduce")),
                       "exchange_fallback": exchange_fallback,
                       "exchange_verification_legs": exchange_legs,
                       "rccl_ranks": rccl_ranks,
                       "launch": ("one C call per alignment: one launch per iteration (its prologue waits for the peers' rows of the "
                                  "previous iteration and solves), nothing from the host in between" if xchg is not None else
                                  "one hipGraph replay per alignment (kernels + all-reduces captured)" if graphs_live
                                  else ("per-iteration launches + all-reduce from the host" if group is not None
                                        else "one C call per alignment"))},
            "iterations_per_sec": args.steps / elapsed,
            "exchange_legs": timed_legs,
            "pose_max_abs_err_vs_ground_truth": pose_err,
            "inliers_last_iteration": int(lin.inlier),
            "setup_s": t_setup,
            "preloop_per_rank": preloop,
            "kernels": kern,
            "launches_of_one_alignment": launches,
            "launch_classes": classes,
            "until_converged": converged,
            "reuse_off_comparison": reuse0,
            "point_to_distribution": p2d,
            "hard_init": hard,
            "lm_geman_mcclure": lm,
            "stages": stages,
            "roofline": roofline_block(dom, kern[dom]),
        }
        # beside `value` (which counts every source point of every iteration, as BASELINE.md 4 defines the metric): the same
        # alignment when every iteration searches every point, and what a caller of align() gets with the default criteria
        if reuse0:
            out["correspondences_per_s_every_iteration_searches"] = reuse0["correspondences_per_s"]
        if converged:
            out["correspondences_per_s_until_converged"] = converged["correspondences_per_s"]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample, REG_TYPE)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if world > 1 or args.force_sharded:
        dist.barrier()
        dist.destroy_process_group()


HBM_COPY_GBS = 6290.0  # measured float4 streaming copy of the part (MI355X_MICROARCH.md chip table)


def measured_traffic(kernel):
    """HBM bytes per launch of the TIMED MIX of launches (one alignment = 20 launches from the identity guess, criteria 0:
    the first ones search, the rest stream), from the rocprofv3 PMC passes over `bench.py --timed-only` (FETCH_SIZE /
    WRITE_SIZE in separate runs, corrected as MI355X_MICROARCH.md §HBM prescribes: profiles/collect_traffic.py), if a
    summary for this kernel has been committed under profiles/. Returns (bytes or None, how it was obtained)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(path):
        return None, None
    try:
        k = json.load(open(path)).get(kernel, {})
        return k.get("timed_mix_hbm_bytes_per_launch", k.get("hbm_bytes_per_launch")), k.get("correction")
    except Exception:
        return None, None


def roofline_block(dom, k):
    traffic, how = measured_traffic(dom)
    t = k["ms"] * 1e-3
    return {"bound": "hbm", "kernel": dom, "achieved": k["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": k["GBps"] / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes_per_launch": k["bytes"],
            "frac_of_measured_copy_peak": k["GBps"] / HBM_COPY_GBS,
            "hbm_frac_measured_traffic": (traffic / t / 1e9 / HBM_PEAK_GBS) if traffic else None,
            "copy_peak_frac_measured_traffic": (traffic / t / 1e9 / HBM_COPY_GBS) if traffic else None,
            "traffic_note": ("mean over the launches of the timed mix (profiles/traffic.json, collected by "
                             "profiles/collect_traffic.py over `bench.py --timed-only`); " + how) if traffic else None}


def preloop_times(sp, _lib, torch, g6, covs_full, rank, world):
    """Pre-loop (k = 20 neighbours + covariances of the replicated target) sharded by query: time of this rank's range
    (sp_grid_self_knn_range) and of the row re-ordering either side of the all-gather, against the whole-cloud call every
    rank made in round 1. The all-gather itself (64 B x n rows over xGMI) cannot be measured on one GPU."""
    L = _lib.lib()
    n = g6.n
    c = (n + world - 1) // world
    first, count = min(rank * c, n), min(c, n - min(rank * c, n))
    nbytes = L.sp_grid_self_workspace_bytes(g6._h)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=covs_full.device)
    covs = torch.empty_like(covs_full)
    send = torch.zeros((c, 16), dtype=torch.float32, device=covs_full.device)
    recv = torch.zeros((world * c, 16), dtype=torch.float32, device=covs_full.device)

    def timed(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    whole = timed(lambda: _lib.check(L.sp_grid_self_knn(g6._h, 20, None, None, sp._ptr(covs), None, sp._ptr(ws), nbytes, sp._stream())))
    part = timed(lambda: _lib.check(L.sp_grid_self_knn_range(g6._h, 20, first, count, None, None, sp._ptr(covs), None, sp._ptr(ws),
                                                             nbytes, sp._stream())))
    gat = timed(lambda: _lib.check(L.sp_grid_gather_rows(g6._h, sp._ptr(covs), 64, first, count, sp._ptr(send), sp._stream())))
    sca = timed(lambda: _lib.check(L.sp_grid_scatter_rows(g6._h, sp._ptr(recv), 64, 0, n, sp._ptr(covs), sp._stream())))
    return {"target_points": n, "ranks": world, "whole_cloud_ms": whole, "this_rank_range_ms": part, "gather_rows_ms": gat,
            "scatter_rows_ms": sca, "all_gather_bytes_received_per_rank": 64 * c * (world - 1),
            "all_gather_ms_at_7x153GBps_links": 64 * c * (world - 1) / (7 * 153e9) * 1e3}


def class_traffic(cls):
    """Measured HBM bytes per launch of one class of gicp_align_kernel launches (profiles/traffic.json, per-dispatch PMC)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        return json.load(open(path)).get("gicp_align_kernel", {}).get("classes", {}).get(cls, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def launch_profile(sp, _lib, torch, reg, prep, T_dev, T_ident, delta, n, reps=7):
    """Duration of EACH of the 20 launches of one alignment (identity guess, criteria 0), by HIP events recorded between
    the launches on the launch stream, and how many source points each launch had to search for (device-side log,
    csrc/sp_internal.h). A launch is 'searching' when more than 1 % of its points were searched — the first poses of an
    alignment — and 'steady' otherwise (correspondences carried over by certificate: a pure stream). The device queue is
    pre-filled behind a spin kernel so that the host's launch rate does not show up as gaps between the events."""
    L = _lib.lib()
    ws, lin = reg._buffers(T_dev.device)
    fp = reg._factor_params(reg.params.robust_default_scale)
    gn = _lib.GnParams(reg.params.gn_lambda, 0.0, 0.0)
    nlog = C.c_size_t(0)
    log_ptr = L.sp_internal_align_searched_log(sp._ptr(ws), C.byref(nlog))
    log_off = log_ptr - ws.data_ptr()
    us = np.zeros((reps, ITERS_PER_ALIGN))
    for r in range(reps):
        T_dev.copy_(T_ident)
        reg._psrc.prepare(prep, reg._bench_source, T_dev, reg._bench_sort_mode)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(ITERS_PER_ALIGN + 1)]
        torch.cuda._sleep(3_000_000)  # ~1.5 ms: the 20 launches + 21 events are all enqueued before the first one starts
        for k in range(ITERS_PER_ALIGN):
            if k == 1:
                ev[0].record()  # (launch 0 is preceded by the reset of the log / ticket words: its event starts behind it)
            _lib.check(L.sp_gicp_align_step(prep._h, reg._psrc._h, sp._ptr(T_dev), C.byref(fp), C.byref(gn), k, 0, None,
                                            None, sp._ptr(lin), sp._ptr(ws), ws.numel(), sp._stream()))
            ev[k + 1].record()
        _lib.check(L.sp_gicp_align_finish(reg._psrc._h, sp._ptr(T_dev), C.byref(gn), ITERS_PER_ALIGN - 1, 0, sp._ptr(lin),
                                          sp._ptr(delta), None, sp._ptr(ws), ws.numel(), sp._stream()))
        torch.cuda.synchronize()
        us[r, 1:] = [1e3 * ev[k].elapsed_time(ev[k + 1]) for k in range(1, ITERS_PER_ALIGN)]
    # launch 0 on its own (an event cannot be recorded between the reset kernel and the launch inside one C call): the whole
    # call between two events, minus the same pair around the reset alone
    t0 = []
    for r in range(reps):
        T_dev.copy_(T_ident)
        reg._psrc.prepare(prep, reg._bench_source, T_dev, reg._bench_sort_mode)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(500_000)
        e0.record()
        _lib.check(L.sp_gicp_align_step(prep._h, reg._psrc._h, sp._ptr(T_dev), C.byref(fp), C.byref(gn), 0, 0, None, None,
                                        sp._ptr(lin), sp._ptr(ws), ws.numel(), sp._stream()))
        e1.record()
        torch.cuda.synchronize()
        t0.append(1e3 * e0.elapsed_time(e1))
    # restore the log of a full alignment for the searched counts
    T_dev.copy_(T_ident)
    reg._psrc.prepare(prep, reg._bench_source, T_dev, reg._bench_sort_mode)
    for k in range(ITERS_PER_ALIGN):
        _lib.check(L.sp_gicp_align_step(prep._h, reg._psrc._h, sp._ptr(T_dev), C.byref(fp), C.byref(gn), k, 0, None, None,
                                        sp._ptr(lin), sp._ptr(ws), ws.numel(), sp._stream()))
    _lib.check(L.sp_gicp_align_finish(reg._psrc._h, sp._ptr(T_dev), C.byref(gn), ITERS_PER_ALIGN - 1, 0, sp._ptr(lin),
                                      sp._ptr(delta), None, sp._ptr(ws), ws.numel(), sp._stream()))
    torch.cuda.synchronize()
    searched = ws[log_off:log_off + 4 * ITERS_PER_ALIGN].view(torch.int32).cpu().numpy().astype(np.int64)
    med = np.median(us, axis=0)
    med[0] = float(np.median(t0))
    launches = [{"k": k, "us": round(float(med[k]), 2), "searched_points": int(searched[k])} for k in range(ITERS_PER_ALIGN)]
    launches[0]["note"] = "includes the one-off reset kernel of an alignment (~2 us)"
    classes = {}
    for name, sel in (("searching", searched > 0.01 * n), ("steady", searched <= 0.01 * n)):
        if not sel.any():
            continue
        t = float(med[sel].mean()) * 1e-6
        traffic = class_traffic(name)
        classes[name] = {"launches": int(sel.sum()), "mean_us": round(t * 1e6, 2),
                         "algorithmic_bytes_per_launch": BYTES_ITER * n,
                         "frac_of_hbm_peak_algorithmic": BYTES_ITER * n / t / 1e9 / HBM_PEAK_GBS,
                         "measured_hbm_bytes_per_launch": traffic,
                         "frac_of_hbm_peak_measured_traffic": (traffic / t / 1e9 / HBM_PEAK_GBS) if traffic else None}
    return launches, classes


def until_converged(sp, torch, S, prep, T_dev, T_ident, delta, n, sort_mode, reg_type="GICP", reps=31, internal=()):
    """What a caller of align() gets: ONE alignment with the reference's default convergence criteria (1e-3 / 1e-3,
    registration_params.hpp:94-96) from the identity guess, timed whole (source preparation + max_iterations launches, the
    ones after convergence returning at once, + finish) with HIP events; correspondences/s = points x executed iterations
    / that time. Median over `reps` alignments."""
    p = sp.RegistrationParams(reg_type=reg_type, optimization_method="GN", max_iterations=ITERS_PER_ALIGN)
    reg = sp.Registration(p)
    for kv in internal:
        reg._set_source_option(kv.split("=")[0], int(kv.split("=")[1]))
    ms = []
    for _ in range(reps + 2):
        T_dev.copy_(T_ident)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        reg.align_fused_loop(S, prep, T_dev=T_dev, delta_dev=delta, prepare=True, sort_by_cell=sort_mode)
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    ms = float(np.median(ms[2:]))
    iters = int(reg._iters_dev[0])
    return {"criteria_rotation": p.criteria_rotation, "criteria_translation": p.criteria_translation,
            "iterations_executed": iters, "converged": bool(float(delta[6]) > 0.5), "ms_per_alignment": ms,
            "correspondences_per_s": n * iters / (ms * 1e-3), "alignments_timed": reps,
            "note": "includes the per-alignment source preparation and the launches after convergence (they return at once)"}


def p2d_block(sp, torch, S, Tg, grid, T_dev, T_ident, delta, n, sort_mode, T_gt):
    """BASELINE config 4 read literally ("GICP (point-to-distribution)"): the same clouds with RegType::POINT_TO_DISTRIBUTION
    (factor.hpp:311-373: M = inverse(Ct), no source covariance) through the same one-call loop. Its Gauss-Newton iteration
    converges more slowly than GICP's, so more of its 20 launches search. Median of 7 alignments each."""
    prep = sp.PreparedTarget(grid, Tg.covs, reg_type="POINT_TO_DISTRIBUTION")
    p = sp.RegistrationParams(reg_type="POINT_TO_DISTRIBUTION", optimization_method="GN", max_iterations=ITERS_PER_ALIGN,
                              criteria_translation=0.0, criteria_rotation=0.0)
    reg = sp.Registration(p)

    def one():
        T_dev.copy_(T_ident)
        reg.align_fused_loop(S, prep, T_dev=T_dev, delta_dev=delta, prepare=True, sort_by_cell=sort_mode)

    ms, runs = median_ms(torch, one, 7)
    err = float(np.abs(reg.T_from_device(T_dev) - T_gt).max())
    conv = until_converged(sp, torch, S, prep, T_dev, T_ident, delta, n, sort_mode, "POINT_TO_DISTRIBUTION", reps=7)
    return {"reg_type": "POINT_TO_DISTRIBUTION", "ms_per_step": ms / ITERS_PER_ALIGN, "ms_per_alignment_20_iterations": ms,
            "correspondences_per_s": n * ITERS_PER_ALIGN / (ms * 1e-3), "alignments_timed": runs,
            "pose_max_abs_err_vs_ground_truth": err, "until_converged": conv,
            "note": "includes the per-alignment source preparation; the target's rows (inverse covariances) are set-up"}


def median_ms(torch, fn, runs=11):
    """Median over `runs` single launches of fn (HIP events on the launch stream, one warm-up)."""
    fn()
    torch.cuda.synchronize()
    ms = []
    for _ in range(runs):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    return float(np.median(ms)), len(ms)


def reuse_off_comparison(sp, torch, args, S, prep, T_dev, T_ident, delta, sort_mode, reg_type, n):
    """A stated comparison, not the benchmark: the same 20-iteration alignment with the correspondence reuse switched off
    (csrc/sp_internal.h, reuse=0: every launch searches every point) — what the certificates buy. Same outputs, bit for bit
    (tests/test_gpu_benchmarked_path.py)."""
    p = sp.RegistrationParams(reg_type=reg_type, optimization_method="GN", max_iterations=ITERS_PER_ALIGN,
                              criteria_translation=0.0, criteria_rotation=0.0)
    reg = sp.Registration(p)
    reg._set_source_option("reuse", 0)

    def one():
        T_dev.copy_(T_ident)
        reg.align_fused_loop(S, prep, T_dev=T_dev, delta_dev=delta, prepare=True, sort_by_cell=sort_mode)

    ms, runs = median_ms(torch, one, 7)
    return {"ms_per_alignment": ms, "ms_per_step": ms / ITERS_PER_ALIGN, "correspondences_per_s": n * ITERS_PER_ALIGN / (ms * 1e-3),
            "alignments_timed": runs, "note": "reuse=0: every iteration searches all points; includes the source preparation"}


def hard_init_block(sp, _lib, torch, S, prep, n, sort_mode, T_gt, reps=7):
    """Config 4 away from its friendliest input (tests/test_gpu_hard_init.py holds each case to a full oracle alignment at this
    size): (i) an initial guess ten cells from the truth at the rim, (ii) a third of the source outside the target with
    max_correspondence_distance 0.3, (iii) Geman-McClure. One alignment with the reference's default criteria each: time,
    iterations executed, source points searched per launch."""
    from sycl_points_amd.synthetic import se3_exp_f64

    out = {}
    cases = {"far_initial_guess": dict(twist=[0.05, -0.03, 0.04, 0.4, -0.3, 0.2], max_corr=2.0, loss="NONE", scale=10.0, shift=False),
             "partial_overlap": dict(twist=None, max_corr=0.3, loss="NONE", scale=10.0, shift=True),
             "geman_mcclure": dict(twist=None, max_corr=2.0, loss="GEMAN_MCCLURE", scale=0.5, shift=False)}
    L = _lib.lib()
    for name, c in cases.items():
        src = S
        if c["shift"]:
            pts = S.points.clone()
            run = (torch.arange(n, device=pts.device) // 1024) % 3 == 0
            pts[run, 0] += 100.0
            src = sp.PointCloudShared(pts, covs=S.covs, device=pts.device)
        T0 = np.eye(4, dtype=np.float32) if c["twist"] is None else se3_exp_f64(c["twist"]).astype(np.float32)
        p = sp.RegistrationParams(max_correspondence_distance=c["max_corr"], robust_type=c["loss"],
                                  robust_default_scale=c["scale"], max_iterations=ITERS_PER_ALIGN)
        reg = sp.Registration(p)
        T0_dev = torch.from_numpy(np.ascontiguousarray(T0.T).reshape(-1).copy()).to(S.points.device)
        T_dev = T0_dev.clone()
        dl = torch.zeros(8, dtype=torch.float32, device=S.points.device)
        ms = []
        for _ in range(reps + 2):
            T_dev.copy_(T0_dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            reg.align_fused_loop(src, prep, T_dev=T_dev, delta_dev=dl, prepare=True, sort_by_cell=sort_mode)
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        ws, lin = reg._buffers(S.points.device)
        nlog = C.c_size_t(0)
        off = L.sp_internal_align_searched_log(sp._ptr(ws), C.byref(nlog)) - ws.data_ptr()
        iters = int(reg._iters_dev[0])
        searched = ws[off:off + 4 * ITERS_PER_ALIGN].view(torch.int32).cpu().numpy().astype(np.int64)[:iters]
        out[name] = {"ms_per_alignment": float(np.median(ms[2:])), "iterations": iters, "converged": bool(float(dl[6]) > 0.5),
                     "inliers": int(reg._read_lin(lin).inlier), "searched_points_per_launch": [int(x) for x in searched],
                     "max_correspondence_distance": c["max_corr"], "robust": c["loss"],
                     "pose_max_abs_err_vs_ground_truth": float(np.abs(reg.T_from_device(T_dev) - T_gt).max())}
    return out


def lm_block(sp, torch, S, prep, n, sort_mode, T_gt, reps=7):
    """The optimiser the reference's callers select (example_registration.cpp:35-36, lidar_odometry.yaml:221) at config-4 size:
    Levenberg-Marquardt with Geman-McClure, the whole loop as ONE launch and ONE read-back (sp_gicp_align_optimize), one level
    and the example's three annealing levels (robust scale 10 -> 5 -> 2.5)."""
    out = {}
    for label, scales in (("one_level_scale_10", [10.0]), ("three_annealing_levels_10_5_2.5", [10.0, 5.0, 2.5])):
        p = sp.RegistrationParams(robust_type="GEMAN_MCCLURE", optimization_method="LM", max_iterations=10)
        reg = sp.Registration(p)
        ms, res = [], None
        for _ in range(reps + 2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            res = reg.align_optimize(S, prep, None, scales, sort_mode)
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        if res is None:
            out[label] = {"available": False}
            continue
        t = float(np.median(ms[2:]))
        steps = res.linearizations + res.trials
        out[label] = {"ms_per_alignment": t, "linearizations": res.linearizations, "trial_evaluations": res.trials,
                      "us_per_step": 1e3 * t / max(steps, 1), "searched_points": res.searched, "converged": res.converged,
                      "correspondences_per_s": n * res.linearizations / (t * 1e-3),
                      "pose_max_abs_err_vs_ground_truth": float(np.abs(res.T - T_gt).max()),
                      "note": "includes the source preparation and the read-back of the 1.4 KB result block"}
    return out


def raw_scan_block(sp, torch, iters=20, reps=5):
    """The reference's bundled scans as they are (69 792 vs 69 088 points, no filter: 5032 returns sit in the sensor's own cell of
    the in-loop grid) through the device-resident Gauss-Newton loop, `iters` iterations: one WAVE per source point (what
    Registration::align asks for on a target with crowded cells) against one lane per point. None when the scans are not there."""
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden")
    if not (os.path.exists(os.path.join(gold, "source.ply")) and os.path.exists(os.path.join(gold, "target.ply"))):
        return None
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests"))
    try:
        from test_gpu_facade import read_ply_xyz
    except Exception:
        return None
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    S = sp.PointCloudShared(dev(read_ply_xyz(os.path.join(gold, "source.ply"))))
    Tg = sp.PointCloudShared(dev(read_ply_xyz(os.path.join(gold, "target.ply"))))
    for c in (S, Tg):
        sp.covariance.estimate(sp.BVH.build(c.points).self_knn(20), c)
    grid = sp.GridKNN.build(Tg.points, points_per_cell=0.5)
    prep = sp.PreparedTarget(grid, Tg.covs)
    out = {"source_points": S.size(), "target_points": Tg.size(), "fullest_cell_of_the_target_grid": grid.max_cell_points(),
           "iterations": iters}
    poses = {}
    for label, mode in (("wave_per_point", 2), ("lane_per_point", 0)):
        reg = sp.Registration(sp.RegistrationParams(max_iterations=iters, criteria_rotation=0.0, criteria_translation=0.0,
                                                    optimization_method="GN"))
        reg._set_source_option("opt_wave_query", mode)
        ms, res = [], None
        for _ in range(reps + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            res = reg.align_optimize(S, prep, None, None, True)
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        if res is None:
            out[label] = {"available": False}
            continue
        poses[label] = res.T
        out[label] = {"us_per_iteration": 1e3 * float(np.median(ms[1:])) / iters, "inliers": res.inlier, "searched_points": res.searched}
    if len(poses) == 2:
        out["pose_max_abs_difference_between_the_two"] = float(np.abs(poses["wave_per_point"] - poses["lane_per_point"]).max())
    return out


def count_launches(torch, fn):
    """Kernel / copy / memset nodes of ONE call of fn, counted by capturing it into a hipGraph on a side stream (the C ABI's
    enqueue-only entry points are capturable). None when the call cannot be captured (it synchronises or allocates)."""
    hip = C.CDLL("libamdhip64.so")
    st = torch.cuda.Stream()
    graph = C.c_void_p()
    n = C.c_size_t(0)
    try:
        fn()
        torch.cuda.synchronize()
        with torch.cuda.stream(st):
            if hip.hipStreamBeginCapture(C.c_void_p(st.cuda_stream), 2) != 0:  # hipStreamCaptureModeRelaxed
                return None
            try:
                fn()
            finally:
                rc = hip.hipStreamEndCapture(C.c_void_p(st.cuda_stream), C.byref(graph))
        if rc != 0 or not graph:
            return None
        if hip.hipGraphGetNodes(graph, None, C.byref(n)) != 0:
            return None
        return int(n.value)
    except Exception:
        return None
    finally:
        if graph:
            hip.hipGraphDestroy(graph)
        torch.cuda.synchronize()


def cpu_ms(fn, runs=3):
    """Median wall time of fn over `runs` runs, ms (the CPU oracle beside a stage)."""
    t = []
    for _ in range(runs):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(t))


def example_block(cpu=True, loops=100, warmup=10):
    """BASELINE config 1: the reference's cpp/examples/example_registration.cpp loop on the bundled scans through the C++ facade
    (tests/cpp/example_registration, built by __graft_entry__.build(): a child process, this one keeps its device) — the
    reference's own per-stage timers — with the CPU oracle's time for the same stage on the same clouds beside each."""
    import re
    import subprocess

    exe = os.path.join(ROOT, "tests", "cpp", "example_registration")
    gold = os.path.join(ROOT, "tests", "golden")
    if not os.path.exists(exe):
        return {"available": False, "note": "tests/cpp/example_registration is not built"}
    # the loop is host-latency-bound (≈ 50 launches and a dozen waits per pass): three child runs, the one with the median TOTAL is
    # reported and the three totals stand beside it
    runs = []
    for _ in range(3):
        r = subprocess.run([exe, os.path.join(gold, "source.ply"), os.path.join(gold, "target.ply"), str(loops), str(warmup)],
                           capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            return {"available": False, "note": r.stderr[-300:]}
        st = {}
        for m in re.finditer(r"^\s*(\d[a-z]?\. [^:]+|TOTAL):\s+([0-9.]+) us", r.stdout, re.M):
            st[m.group(1).strip()] = float(m.group(2)) / 1e3
        runs.append(st)
    runs.sort(key=lambda x: x.get("TOTAL", 0.0))
    st = runs[1]
    out = {"loops": loops, "warmup": warmup, "gpu_ms": st, "total_ms_of_the_three_runs": [x.get("TOTAL") for x in runs],
           "note": "LM + Geman-McClure + 3 annealing levels on a 1000-point sample; box filter [0.5, 50] m, voxel 0.25 m, k = 10"}
    if cpu:
        out["cpu_oracle_ms"], out["cpu_cores"] = example_cpu_oracle()
    return out


def example_cpu_oracle():
    """The same pipeline on the CPU oracle (all host cores), stage by stage like the reference's timers."""
    from oracle.pyoracle import LOSS, OPT, REG, Oracle, RegParams

    orc = Oracle()
    gold = os.path.join(ROOT, "tests", "golden")

    def read(path):
        raw = open(path, "rb").read()
        head, body = raw.split(b"end_header\n", 1)
        n = int([l for l in head.split(b"\n") if l.startswith(b"element vertex")][0].split()[-1])
        a = np.frombuffer(body, dtype="<f4", count=n * 4).reshape(n, 4)
        pts = np.ones((n, 4), np.float32)
        pts[:, :3] = a[:, :3]
        return pts

    src, tgt = read(os.path.join(gold, "source.ply")), read(os.path.join(gold, "target.ply"))
    t = {}

    def timed(name, fn):
        out = []
        t[name] = t.get(name, 0.0) + cpu_ms(lambda: out.append(fn()), 3)
        return out[-1]

    def down(p):
        return orc.voxel_downsample(p[orc.box_filter(p, 0.5, 50.0) == 1], 0.25, 1, stable=True)["points"]

    s = timed("2. Downsampling", lambda: down(src))
    g = timed("2. Downsampling", lambda: down(tgt))
    ns = timed("3. KNN structure build", lambda: orc.kdtree_build(s))
    ng = timed("3. KNN structure build", lambda: orc.kdtree_build(g))
    si = timed("4. kNN Search", lambda: orc.kdtree_knn(ns, s, 10)[0])
    gi = timed("4. kNN Search", lambda: orc.kdtree_knn(ng, g, 10)[0])
    sc = timed("5. compute Covariances", lambda: orc.cov_estimate(s, si))
    gc = timed("5. compute Covariances", lambda: orc.cov_estimate(g, gi))
    keep = orc.random_sampling_flags(1234, len(s), 1000) == 1
    p = RegParams.defaults(reg_type=REG["GICP"], robust_type=LOSS["GEMAN_MCCLURE"], optimization_method=OPT["LM"], max_iterations=10,
                           max_correspondence_distance=2.0, robust_default_scale=10.0, auto_scale=1, init_scale=10.0, min_scale=2.5,
                           auto_scaling_iter=3)
    timed("7. Registration", lambda: orc.registration_align(p, s[keep], sc[keep], g, gc, nodes=ng))
    t["TOTAL (stages 2-5, 7)"] = sum(t.values())
    return t, orc.num_threads()


def stage_block(sp, _lib, torch, cpu=True):
    """BASELINE configs 2 and 3 and the pre-loop of config 4 in the driver's line (outside the timed GICP region): each
    stage with inputs resident in HBM, the median of 11 single runs by HIP events, its rate and the fraction of the roofline
    that bounds it (brute force against the bf16 MFMA peak of its bounding pass, with the fp32 VALU roofline of
    the reference's 9-operation expression beside it; HBM at the API-layout bytes otherwise).
    Outputs are preallocated; the voxel stage is the C-ABI call alone (key box of the previous cloud known, as from the
    second frame of a sensor on) without the host's read-back of the voxel count."""
    from sycl_points_amd.synthetic import Mt19937Cloud

    FP32 = 157.3e12
    L = _lib.lib()
    out = {}
    orc = None
    if cpu:
        from oracle.pyoracle import Oracle
        orc = Oracle()
    g = Mt19937Cloud(1234)
    tgt = torch.from_numpy(g.uniform_points(100000, 10.0)).cuda()
    qry = torch.from_numpy(g.uniform_points(100000, 10.0)).cuda()
    tgt_np, qry_np = tgt.cpu().numpy(), qry.cpu().numpy()
    BF16 = 2.5e15  # dense bf16 MFMA peak (MI355X_MICROARCH.md)
    for k in (1, 20):
        ms, runs = median_ms(torch, lambda: sp.knn_search_bruteforce(qry, tgt, k))
        _lib.check(L.sp_knn_bruteforce_set_pass_a(1))
        try:
            ms_valu, _ = median_ms(torch, lambda: sp.knn_search_bruteforce(qry, tgt, k), 5)
        finally:
            _lib.check(L.sp_knn_bruteforce_set_pass_a(0))
        floor_ms = 1e10 * 2 * 16 / BF16 * 1e3  # pass A alone at the bf16 peak: one 32x32x16 MFMA (32 flop per pair) per 1024 pairs
        out[f"bruteforce_100k_x_100k_k{k}"] = {
            "ms": ms, "runs": runs, "pairs_per_s": 1e10 / (ms * 1e-3),
            "bound": "bf16 MFMA: every pair goes through pass A (|p|^2 - 2 q.p on bf16-split operands, 32 flop per pair); the "
                     "reference's fp32 expression is evaluated only for the (query, chunk) pairs that can hold a neighbour",
            "frac_of_bound": floor_ms / ms,
            "rate_vs_fp32_valu_roofline_of_the_reference_expression": 9 * 1e10 / (ms * 1e-3) / FP32,
            "ms_with_pass_a_on_packed_fp32_valu": ms_valu,
            "hbm_GBps_algorithmic": (16 * 2e5 + 8 * 1e5 * k) / (ms * 1e-3) / 1e9,
            "launches": bf_launches(sp, _lib, torch, qry, tgt, k)}
        if orc is not None:  # the oracle's brute force on a 10 k-query subsample (all host cores), scaled to the 100 k queries
            sub = 10000
            t_cpu = cpu_ms(lambda: orc.knn_bruteforce(qry_np[:sub], tgt_np, k))
            out[f"bruteforce_100k_x_100k_k{k}"].update(cpu_oracle_ms=t_cpu * (100000 / sub), cpu_cores=orc.num_threads(),
                                                         cpu_sample=f"{sub} of the 100000 queries, time scaled by {100000 // sub}")
    del tgt, qry
    # a cloud of the size the reference's example searches (its 6 k-point downsampled scans, k = 10): ONE launch with the cloud in
    # LDS (knn_bf_small_kernel); issue-bound, reported against the fp32 VALU roofline of the reference's 9-operation expression
    small = torch.from_numpy(Mt19937Cloud(77).uniform_points(6096, 10.0)).cuda()
    small_np = small.cpu().numpy()
    ms, runs = median_ms(torch, lambda: sp.knn_search_bruteforce(small, small, 10))
    out["bruteforce_6k_x_6k_k10_one_launch"] = {
        "ms": ms, "runs": runs, "pairs_per_s": 6096.0 ** 2 / (ms * 1e-3), "launches": bf_launches(sp, _lib, torch, small, small, 10),
        "bound": "fp32 VALU issue / LDS latency: two scans of the cloud (12 VALU operations per pair) + two 64-lane sorts per query",
        "rate_vs_fp32_valu_roofline_of_the_reference_expression": 9 * 6096.0 ** 2 / (ms * 1e-3) / FP32}
    if orc is not None:
        out["bruteforce_6k_x_6k_k10_one_launch"].update(cpu_oracle_ms=cpu_ms(lambda: orc.knn_bruteforce(small_np, small_np, 10)),
                                                         cpu_cores=orc.num_threads())
    del small
    for name, R in (("sparse_R10", 10.0), ("dense_R2.5", 2.5)):
        n = 1_000_000
        P = torch.from_numpy(Mt19937Cloud(1234).uniform_points(n, R)).cuda()
        vg = sp.VoxelGrid(0.1)
        nvox = vg.downsampling(P).size()  # (remembers the key box)
        box = vg._key_box
        nbytes = L.sp_voxel_downsample_workspace_bytes(n)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=P.device)
        o_p = torch.empty((n, 4), dtype=torch.float32, device=P.device)
        info = torch.zeros(8, dtype=torch.int32, device=P.device)  # the call's record: voxels, points outside the box, key box

        def run():
            _lib.check(L.sp_voxel_downsample_report(sp._ptr(P), n, vg.voxel_size_inv, 1, None, None, None, sp._ptr(o_p), None,
                                                    None, None, None, None, box.ctypes.data_as(C.c_void_p), sp._ptr(info),
                                                    sp._ptr(ws), nbytes, sp._stream()))

        ms, runs = median_ms(torch, run)
        assert int(info[0]) == nvox and int(info[1]) == 0
        ms_api, _ = median_ms(torch, lambda: vg.downsampling(P), 5)
        out[f"voxel_downsample_1M_{name}"] = {
            "ms": ms, "runs": runs, "voxels": nvox, "points_per_s": n / (ms * 1e-3), "bound": "HBM at 40 B per point",
            "GBps": 40 * n / (ms * 1e-3) / 1e9, "frac_of_bound": 40 * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "ms_whole_api_call_with_count_read_back": ms_api, "launches": count_launches(torch, run)}
        if orc is not None:
            P_np = P.cpu().numpy()
            out[f"voxel_downsample_1M_{name}"].update(cpu_oracle_ms=cpu_ms(lambda: orc.voxel_downsample(P_np, 0.1, 1, stable=True)),
                                                      cpu_cores=1, cpu_note="the reference aggregates on ONE host thread "
                                                      "(voxel_downsampling.hpp:146-288: std::sort + a sequential run-length mean)")
        del P, ws, o_p
    n = 1_000_000
    P = torch.from_numpy(Mt19937Cloud(1234).uniform_points(n, 10.0)).cuda()
    grid = sp.GridKNN.build(P, points_per_cell=6.0)
    for name, (knn, cov), bytes_pt in (("self_knn_k20_1M", (True, False), 176), ("self_knn_k20_plus_covariance_1M", (False, True), 176 + 464)):
        ms, runs = median_ms(torch, lambda: grid.self_knn(20, knn, cov, False))
        out[name] = {"ms": ms, "runs": runs, "points_per_s": n / (ms * 1e-3),
                     "bound": f"HBM at {bytes_pt} B per point (API layouts" + (": kNN 176 + K5 464)" if cov else ")"),
                     "GBps": bytes_pt * n / (ms * 1e-3) / 1e9, "frac_of_bound": bytes_pt * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "launches": count_launches(torch, lambda: grid.self_knn(20, knn, cov, False))}
    if orc is not None:  # KD-tree k = 20 of the cloud against itself (+ K5), all host cores; the tree build is its own figure
        P_np = P.cpu().numpy()
        nodes = orc.kdtree_build(P_np)
        t_knn = cpu_ms(lambda: orc.kdtree_knn(nodes, P_np, 20))
        idx20 = orc.kdtree_knn(nodes, P_np, 20)[0]
        t_cov = cpu_ms(lambda: orc.cov_estimate(P_np, idx20))
        t_build = cpu_ms(lambda: orc.kdtree_build(P_np), 1)
        out["self_knn_k20_1M"].update(cpu_oracle_ms=t_knn, cpu_cores=orc.num_threads(), cpu_kdtree_build_ms=t_build)
        out["self_knn_k20_plus_covariance_1M"].update(cpu_oracle_ms=t_knn + t_cov, cpu_cores=orc.num_threads(),
                                                      cpu_kdtree_build_ms=t_build)
        del idx20
    for label, cloud in (("", P), ("_cell_ordered", P[sp.GridKNN.build(P, points_per_cell=1.0).order()].contiguous())):
        g = grid if cloud is P else sp.GridKNN.build(cloud, points_per_cell=6.0)
        res = g.self_knn(20, True, False, False)[0]
        ms, runs = median_ms(torch, lambda: sp.covariance.estimate(res, cloud))
        out["covariance_K5_alone_1M_k20" + label] = {
            "ms": ms, "runs": runs, "points_per_s": n / (ms * 1e-3), "bound": "HBM at 464 B per point",
            "cloud_order": "cell order (voxel-downsampled / GridKNN.order())" if label else "as generated (random): every gather misses L2",
            "GBps": 464 * n / (ms * 1e-3) / 1e9, "frac_of_bound": 464 * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        del res
    ms, runs = median_ms(torch, lambda: sp.GridKNN.build(P, points_per_cell=0.5), 5)
    out["grid_build_1M"] = {"ms": ms, "runs": runs, "note": "device build of the in-loop NN structure (synchronises once)"}
    del grid
    # KDTree::build + knn_search on a cloud the uniform grid is bad at (what the facade's KDTree::build answers from the
    # device-built hierarchy, csrc/bvh.hip): three noisy planes, a slab and a fifth of the points in a 20 cm ball
    NU = torch.from_numpy(nonuniform_cloud(n)).cuda()
    ms_b, _ = median_ms(torch, lambda: sp.BVH.build(NU), 5)
    bvh = sp.BVH.build(NU)
    res = sp.KNNResult()
    ms_self, runs = median_ms(torch, lambda: bvh.self_knn(20), 7)
    ms_q, _ = median_ms(torch, lambda: bvh.knn_search_async(NU, 20, res), 5)
    ms_u, _ = median_ms(torch, lambda: sp.BVH.build(P).self_knn(20), 3)
    out["hierarchy_knn_k20_nonuniform_1M"] = {
        "ms": ms_self, "runs": runs, "points_per_s": n / (ms_self * 1e-3), "bound": "HBM at 176 B per point (API layouts)",
        "GBps": 176 * n / (ms_self * 1e-3) / 1e9, "frac_of_bound": 176 * n / (ms_self * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "ms_queries_in_cloud_order": ms_q, "ms_build": ms_b, "ms_build_plus_self_knn_uniform_1M": ms_u,
        "cloud": "three noisy planes + slab + 200 k points in a 20 cm ball (tests/test_gpu_bvh.py::nonuniform_cloud)"}
    return out


def bf_launches(sp, _lib, torch, qry, tgt, k):
    L = _lib.lib()
    nq, nt = qry.shape[0], tgt.shape[0]
    nbytes = L.sp_knn_bruteforce_workspace_bytes(nq, nt, k)
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=qry.device)
    idx = torch.empty((nq, k), dtype=torch.int32, device=qry.device)
    d2 = torch.empty((nq, k), dtype=torch.float32, device=qry.device)
    return count_launches(torch, lambda: _lib.check(L.sp_knn_bruteforce(sp._ptr(qry), nq, sp._ptr(tgt), nt, k, sp._ptr(idx), sp._ptr(d2),
                                                                       sp._ptr(ws), nbytes, sp._stream())))


def nonuniform_cloud(n, seed=7):
    """The non-uniform cloud of tests/test_gpu_bvh.py (density varies by more than four orders of magnitude)."""
    rs = np.random.RandomState(seed)
    m = n // 5
    parts = []
    for axis in range(3):
        p = rs.uniform(-40, 40, (m, 3))
        p[:, axis] = rs.normal(0.0, 0.01, m)
        parts.append(p)
    parts.append(rs.uniform(-40, 40, (n - 4 * m, 3)) * np.array([1.0, 1.0, 0.1]))
    c = rs.normal(0.0, 1.0, (m, 3))
    parts.append(np.array([3.0, -2.0, 1.0]) + 0.2 * c / np.maximum(np.linalg.norm(c, axis=1, keepdims=True), 1e-9) * rs.uniform(0, 1, (m, 1)) ** (1 / 3))
    pts = np.ones((n, 4), np.float32)
    pts[:, :3] = np.concatenate(parts)[:n].astype(np.float32)
    return pts


def kernel_times(sp, _lib, torch, args, reg, S, Tg, knn, prep, T_dev, T_ident, delta, n, sort_mode, reps=3):
    """Average launch duration of the hot kernel(s) over ONE ALIGNMENT (20 poses, from the identity guess to
    convergence), measured with HIP events recorded on the stream the kernels are launched on (the C ABI is handed
    torch's current stream). For the fused path the second launch (partial sums + solve) is masked off while the first
    is being timed, so the figure is the duration of the per-iteration kernel alone, the number rocprofv3 reports."""
    L = _lib.lib()
    res = {}
    scale = reg.params.robust_default_scale

    def timed(fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    if args.path == "fused":
        ws, lin = reg._buffers(S.points.device)
        fp = reg._factor_params(scale)
        gn = _lib.GnParams(reg.params.gn_lambda, 0.0, 0.0)

        iters_dev = torch.zeros(1, dtype=torch.int32, device=S.points.device)

        def align_launches():
            T_dev.copy_(T_ident)
            _lib.check(L.sp_gicp_align_fused(prep._h, reg._psrc._h, sp._ptr(T_dev), C.byref(fp), C.byref(gn),
                                             ITERS_PER_ALIGN, None, None, sp._ptr(lin), sp._ptr(delta), sp._ptr(iters_dev),
                                             sp._ptr(ws), ws.numel(), sp._stream()))

        reg.align_fused_loop(S, prep, iterations=0, T_dev=T_dev, delta_dev=delta, prepare=True, sort_by_cell=sort_mode)
        reg._bench_source, reg._bench_sort_mode = S, sort_mode
        align_launches()
        torch.cuda.synchronize()
        # (a) the 20 per-iteration launches of one alignment, back to back, without the finish kernel: every launch
        #     after the first starts from the pose the previous one's partial sums give, exactly as in the timed loop
        reg._set_source_option("stage_mask", 1)
        T_dev.copy_(T_ident)
        torch.cuda.synchronize()
        reps_a = 5
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tot = 0.0
        for _ in range(reps_a):
            T_dev.copy_(T_ident)
            e0.record()
            _lib.check(L.sp_gicp_align_fused(prep._h, reg._psrc._h, sp._ptr(T_dev), C.byref(fp), C.byref(gn),
                                             ITERS_PER_ALIGN, None, None, sp._ptr(lin), sp._ptr(delta), sp._ptr(iters_dev),
                                             sp._ptr(ws), ws.numel(), sp._stream()))
            e1.record()
            torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
        ms = tot / reps_a / ITERS_PER_ALIGN
        res["gicp_align_kernel"] = {"ms": ms, "bytes": BYTES_ITER * n, "GBps": BYTES_ITER * n / (ms * 1e-3) / 1e9,
                                    "note": "per-iteration launch: prologue (every workgroup sums the previous launch's 256 "
                                            "partial rows, solves the 6x6 system and updates the pose), then per point certified "
                                            "reuse of the previous correspondence or NN(k=1) search, linearise, workgroup "
                                            "reduction; mean over the 20 launches of an alignment started at the identity"}
        # (b) the finish kernel alone
        reg._set_source_option("stage_mask", 2)
        align_launches()
        torch.cuda.synchronize()
        ms2 = timed(align_launches)
        reg._set_source_option("stage_mask", 3)
        res["align_finish_kernel"] = {"ms": ms2, "bytes": 256 * 128, "GBps": 256 * 128 / (ms2 * 1e-3) / 1e9,
                                      "per_iteration": False,
                                      "note": "once per alignment (incl. a 64-byte pose copy): last iteration's partial "
                                              "sums + solve + outputs"}
        ms3 = timed(lambda: reg._psrc.prepare(prep, S, T_ident, sort_mode))
        res["source_prepare"] = {"ms": ms3, "bytes": 120 * n, "GBps": 120 * n / (ms3 * 1e-3) / 1e9, "per_iteration": False,
                                 "note": "once per alignment: " + ("" if sort_mode == "presorted" else "cell-order sort + ") +
                                         "gather + plane-regularised source covariances"}
    else:
        T_dev.copy_(T_ident)
        tot_nn = tot_k11 = 0.0
        for _ in range(ITERS_PER_ALIGN):
            knn.nearest_neighbor_search_async(S, reg.neighbors, T_dev)
            torch.cuda.synchronize()
            tot_nn += timed(lambda: knn.nearest_neighbor_search_async(S, reg.neighbors, T_dev))
            tot_k11 += timed(lambda: reg._linearize("linearize", S, Tg, T_dev, scale, reg._lin))
            reg.align_device_loop(S, Tg, knn, iterations=1, T_dev=T_dev, delta_dev=delta)
        for name, ms, bpp in (("nn_search_k1", tot_nn / ITERS_PER_ALIGN, BYTES_NN),
                              ("gicp_linearize_reduce", tot_k11 / ITERS_PER_ALIGN, BYTES_K11)):
            res[name] = {"ms": ms, "bytes": bpp * n, "GBps": bpp * n / (ms * 1e-3) / 1e9}
    return res


def cpu_baseline(n_cpu, reg_type="GICP"):
    """The CPU oracle (kind "port": our restatement of the reference's algorithms, OpenMP over points, all host cores)
    on the same workload: n_cpu-vs-n_cpu points at config-4 density, alignments of 20 GN iterations with KD-tree NN,
    repeated for >= 12 s. KD-tree build and covariances are outside the timed loop, as on the GPU side."""
    from oracle.pyoracle import REG, Oracle, RegParams
    from sycl_points_amd.synthetic import gicp_pair

    orc = Oracle()
    r = 10.0 * (n_cpu / 1e6) ** (1.0 / 3.0)
    src, tgt, _ = gicp_pair(n_cpu, r)
    nodes_t = orc.kdtree_build(tgt)
    ti, _ = orc.kdtree_knn(nodes_t, tgt, 20)
    si, _ = orc.kdtree_knn(orc.kdtree_build(src), src, 20)
    scov, tcov = orc.cov_estimate(src, si), orc.cov_estimate(tgt, ti)
    p = RegParams.defaults(reg_type=REG[reg_type], crit_translation=0.0, crit_rotation=0.0, max_iterations=ITERS_PER_ALIGN)
    orc.registration_align(p, src[:2000], scov[:2000], tgt, tcov, nodes=nodes_t)  # warm-up
    t0 = time.perf_counter()
    runs = 0
    while True:
        orc.registration_align(p, src, scov, tgt, tcov, nodes=nodes_t)
        runs += 1
        if time.perf_counter() - t0 > 12.0:
            break
    dt = time.perf_counter() - t0
    return {"value": n_cpu * ITERS_PER_ALIGN * runs / dt, "unit": "correspondences/s", "cores": orc.num_threads(),
            "kind": "port",
            "sample": f"{runs} alignments x {ITERS_PER_ALIGN} iterations of {reg_type} {n_cpu}-vs-{n_cpu} (config-4 density, "
                      f"KD-tree NN k=1 + linearise per iteration), {dt:.1f} s"}


if __name__ == "__main__":
    main()
