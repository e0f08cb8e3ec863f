"""N > 1 path with the real kernels: two ranks share the one GPU of the test box (gloo moves the partial rows / the
192-byte system, which is what RCCL does on a multi-GPU node), each linearises its tile of the source, the sums are
all-reduced and every rank solves identically on the device. The pose must equal the single-rank run to rounding, for the
one-launch-per-iteration loop (partial rows all-reduced, solve in the next launch's prologue), the two-launch fused
loop and the generic loop."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(n):
    import sycl_points_amd.api as sp
    from sycl_points_amd.synthetic import gicp_pair

    src, tgt, T_gt = gicp_pair(n, 10.0 * (n / 1e6) ** (1.0 / 3.0))
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    Tg = sp.PointCloudShared(dev(tgt))
    Tg.covs = sp.GridKNN.build(Tg.points, points_per_cell=6.0).self_knn(20, want_knn=False, want_covs=True)[1]
    S = sp.PointCloudShared(dev(src))
    S.covs = sp.GridKNN.build(S.points, points_per_cell=6.0).self_knn(20, want_knn=False, want_covs=True)[1]
    return sp, S, Tg, T_gt


def _worker(rank, world, port, n, iters, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sycl_points_amd.sharding import shard_range

    sp, S, Tg, _ = _make(n)
    lo, hi = shard_range(n, rank, world)
    Sh = sp.PointCloudShared(S.points[lo:hi].contiguous(), covs=S.covs[lo:hi].contiguous())
    grid = sp.GridKNN.build(Tg.points)
    prep = sp.PreparedTarget(grid, Tg.covs)
    p = sp.RegistrationParams(criteria_translation=0.0, criteria_rotation=0.0, max_iterations=iters)
    reg = sp.Registration(p)
    T1, lin1, _ = reg.align_fused_loop(Sh, prep, iterations=iters, group=dist.group.WORLD)  # fan-in row exchange (default)
    regr = sp.Registration(p)
    T1r, _, _ = regr.align_fused_loop(Sh, prep, iterations=iters, group=dist.group.WORLD, exchange="rows")  # 32 KB rows
    torch.cuda.synchronize()
    assert float((T1 - T1r).abs().max()) < 2e-6
    torch.cuda.synchronize()
    inl = reg._read_lin(lin1).inlier
    iters_done = int(reg._iters_dev[0])
    reg2 = sp.Registration(p)
    T2, _, _ = reg2.align_device_loop(Sh, Tg, grid, iterations=iters, group=dist.group.WORLD)
    reg3 = sp.Registration(p)
    T3, lin3, _ = reg3.align_fused_loop(Sh, prep, iterations=iters, group=dist.group.WORLD, per_iteration_launches=True)
    torch.cuda.synchronize()
    inl3 = reg3._read_lin(lin3).inlier
    # pre-loop sharded by query: each rank computes the k = 20 covariances of half of the target's grid positions, one
    # all-gather shares them; every row must equal the covariance the whole-cloud kernel computes, bit for bit
    g6 = sp.GridKNN.build(Tg.points, points_per_cell=6.0)
    full = g6.self_knn(20, want_knn=False, want_covs=True)[1]
    shared = g6.covariances_sharded(20, rank, world, lambda send, recv: dist.all_gather_into_tensor(recv, send))
    torch.cuda.synchronize()
    assert torch.equal(shared, full)
    g2 = sp.GridKNN.build(Tg.points[:5001].contiguous(), points_per_cell=2.0)  # short lists (lane kernel), odd size
    full2 = g2.self_knn(7, want_knn=False, want_covs=True)[1]
    shared2 = g2.covariances_sharded(7, rank, world, lambda send, recv: dist.all_gather_into_tensor(recv, send))
    torch.cuda.synchronize()
    assert torch.equal(shared2, full2)
    np.save(out_path % rank, np.concatenate([T1.cpu().numpy(), T2.cpu().numpy(), [np.float32(inl)], T3.cpu().numpy(),
                                             [np.float32(inl3), np.float32(iters_done)]]))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_one_gpu_match_single_rank(tmp_path):
    n, iters, world = 60000, 8, 2
    out = str(tmp_path / "rank%d.npy")
    mp.spawn(_worker, args=(world, _free_port(), n, iters, out), nprocs=world, join=True)
    r0, r1 = np.load(out % 0), np.load(out % 1)
    assert np.array_equal(r0, r1)  # identical pose and count on every rank
    sp, S, Tg, T_gt = _make(n)
    grid = sp.GridKNN.build(Tg.points)
    prep = sp.PreparedTarget(grid, Tg.covs)
    p = sp.RegistrationParams(criteria_translation=0.0, criteria_rotation=0.0, max_iterations=iters)
    reg = sp.Registration(p)
    T_single, _, _ = reg.align_fused_loop(S, prep, iterations=iters)
    single = T_single.cpu().numpy()
    assert np.abs(r0[:16] - single).max() < 2e-6    # fused loop, sharded vs not
    assert np.abs(r0[16:32] - single).max() < 2e-6  # generic loop, sharded
    assert int(r0[32]) == n                          # inlier count summed exactly over ranks (float rows)
    assert np.abs(r0[33:49] - single).max() < 2e-6  # two-launch fused loop, sharded
    assert int(r0[49]) == n and int(r0[50]) == iters
    assert np.abs(single.reshape(4, 4).T - T_gt).max() < 5e-4


def _graph_worker(rank, port, n, iters, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    sp, S, Tg, _ = _make(n)
    prep = sp.PreparedTarget(sp.GridKNN.build(Tg.points), Tg.covs)
    p = sp.RegistrationParams(criteria_translation=0.0, criteria_rotation=0.0, max_iterations=iters)
    ident = torch.eye(4, dtype=torch.float32, device="cuda").reshape(-1).contiguous()
    res = []
    for use_graph in (False, True):
        reg = sp.Registration(p)
        T_dev = torch.zeros(16, dtype=torch.float32, device="cuda")
        delta = torch.zeros(8, dtype=torch.float32, device="cuda")
        for _ in range(4):  # eager, capture + replay, replay, replay
            T_dev.copy_(ident)
            reg.align_fused_loop(S, prep, iterations=iters, group=dist.group.WORLD, T_dev=T_dev, delta_dev=delta,
                                 graph=use_graph)
            torch.cuda.synchronize()
            res.append(np.concatenate([T_dev.cpu().numpy(), [np.float32(reg._read_lin(reg._lin).inlier)]]))
        live = [v for v in getattr(reg, "_loop_graphs", {}).values() if not isinstance(v, (str, bool))]
        res.append(np.full(17, float(len(live)), np.float32))
    np.save(out_path, np.stack(res))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_sharded_loop_as_one_hipgraph_equals_per_call_launches(tmp_path):
    """The sharded per-iteration loop (kernel, RCCL all-reduce of the partial rows, kernel, ...) captured once into a
    hipGraph and replayed per alignment gives the same bits as issuing every launch and collective from the host."""
    out = str(tmp_path / "graph.npy")
    mp.spawn(_graph_worker, args=(_free_port(), 60000, 8, out), nprocs=1, join=True)
    r = np.load(out)
    eager, n_eager, graph, n_graph = r[0:4], r[4, 0], r[5:9], r[9, 0]
    assert n_eager == 0 and n_graph == 1  # the graph path really replays a captured graph
    for row in list(eager) + list(graph):
        assert np.array_equal(row, eager[0])
    assert int(eager[0][16]) == 60000


def _fanin_worker(rank, port, sizes, iters, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=0, world_size=1)
    import sycl_points_amd.api as sp_api

    comm = sp_api.Communicator.from_process_group(dist.group.WORLD)  # the library's own RCCL communicator (one rank)
    assert comm.world == 1 and comm.rank == 0
    rows = []
    for n in sizes:
        sp, S, Tg, _ = _make(n)
        order = sp.GridKNN.build(S.points, points_per_cell=1.0).order()
        S = S.reordered(order)
        prep = sp.PreparedTarget(sp.GridKNN.build(Tg.points, points_per_cell=0.5), Tg.covs)
        p = sp.RegistrationParams(criteria_translation=0.0, criteria_rotation=0.0, max_iterations=iters)
        ident = torch.eye(4, dtype=torch.float32, device="cuda").reshape(-1).contiguous()

        def run(**kw):
            reg = sp.Registration(p)
            T_dev = ident.clone()
            delta = torch.zeros(8, dtype=torch.float32, device="cuda")
            out = []
            for _ in range(3):  # repeated alignments: the ticket counter must come back to zero every launch
                T_dev.copy_(ident)
                _, lin, _ = reg.align_fused_loop(S, prep, iterations=iters, T_dev=T_dev, delta_dev=delta,
                                                 sort_by_cell="presorted", **kw)
                torch.cuda.synchronize()
                out.append(np.concatenate([T_dev.cpu().numpy(), lin.cpu().numpy(), delta.cpu().numpy(),
                                           [np.float32(int(reg._iters_dev[0]))]]))
            return out

        single = run()                                              # one C call, every workgroup sums the 256 rows
        row_gloo = run(group=dist.group.WORLD, exchange="row")      # fan-in row, all-reduced by torch.distributed
        row_rccl = run(comm=comm)                                   # sp_gicp_align_sharded: fan-in row + sp_allreduce_rows
        row_graph = run(comm=comm, graph=True)                      # the same, captured into one hipGraph and replayed
        rows.append(np.stack(single + row_gloo + row_rccl + row_graph))
    # sp_allreduce_f32 / sp_allgather with one rank: identity
    t = torch.arange(48, dtype=torch.float32, device="cuda")
    comm.all_reduce(t)
    g = torch.zeros(48, dtype=torch.float32, device="cuda")
    comm.all_gather(t, g)
    torch.cuda.synchronize()
    assert torch.equal(t, torch.arange(48, dtype=torch.float32, device="cuda")) and torch.equal(g, t)
    np.save(out_path, np.stack(rows))
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_fanin_row_equals_the_256_row_sum_bit_for_bit(tmp_path):
    """rows_all_reduced = 2: the last-arriving workgroup of a launch sums that launch's partial rows (sc1 stores, agent-scope
    ticket, sc1 loads — no fence) in the fixed order of the single-GPU prologue. With one rank the all-reduce is the
    identity, so pose, linear system, delta and iteration count of the sharded loop must equal the one-call single-GPU loop
    BIT FOR BIT — any stale or torn row read by the fan-in would change a sum. 1M points (256 workgroups, every CU), a size
    that leaves the last workgroups nearly empty (uneven arrival), and a small cloud (fewer workgroups than CUs); through
    torch.distributed, through the library's own RCCL communicator (sp_gicp_align_sharded), and replayed from a hipGraph."""
    out = str(tmp_path / "fanin.npy")
    sizes = (1_000_000, 263_173, 5_000)
    mp.spawn(_fanin_worker, args=(_free_port(), sizes, 12, out), nprocs=1, join=True)
    r = np.load(out)  # [size][12 runs][row]
    for si, n in enumerate(sizes):
        ref = r[si][0]
        assert int(ref[-1]) == 12 and int(np.frombuffer(ref[16 + 43:16 + 44].tobytes(), np.uint32)[0]) == n
        for run in r[si][1:]:
            assert np.array_equal(run, ref), f"n = {n}: sharded fan-in differs from the single-GPU loop"


def _direct_worker(rank, world, port, n, iters, out_path, absent):
    """Two processes on the one GPU exchange their rows DIRECTLY (sp_xchg: hipIpc-mapped slot buffers, tagged granules, bounded
    wait) instead of through a collective; gloo only carries the 64-byte handles once. `absent`: rank 1 connects and then never
    runs the loop — rank 0's wait must end in an error code, not in a hang."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sycl_points_amd.sharding import shard_range

    sp, S, Tg, _ = _make(n)
    lo, hi = shard_range(n, rank, world)
    Sh = sp.PointCloudShared(S.points[lo:hi].contiguous(), covs=S.covs[lo:hi].contiguous())
    grid = sp.GridKNN.build(Tg.points)
    prep = sp.PreparedTarget(grid, Tg.covs)
    p = sp.RegistrationParams(criteria_translation=0.0, criteria_rotation=0.0, max_iterations=iters)
    x = sp.Exchange.from_process_group(dist.group.WORLD, timeout_ms=300 if absent else 20000)
    res = {}
    if not absent:
        for rep in range(2):  # twice: the sequence numbers move on, the slots are reused
            reg = sp.Registration(p)
            T, lin, _ = reg.align_fused_loop(Sh, prep, iterations=iters, xchg=x)
            torch.cuda.synchronize()
            reg.direct_status()
            res[f"T{rep}"] = T.cpu().numpy()
            res[f"inl{rep}"] = np.float32(reg._read_lin(lin).inlier)
            res[f"it{rep}"] = np.float32(int(reg._iters_dev[0]))
        regc = sp.Registration(sp.RegistrationParams(max_iterations=iters))  # default criteria: every rank stops at the same launch
        Tc, _, dc = regc.align_fused_loop(Sh, prep, xchg=x)
        torch.cuda.synchronize()
        regc.direct_status()
        res["Tc"] = Tc.cpu().numpy()
        res["itc"] = np.float32(int(regc._iters_dev[0]))
        # Back to back, nothing synchronised in between, odd iteration counts and early convergence, one rank enqueueing late:
        # the slot an alignment's LAST row sits in must not be the one the next alignment's FIRST row goes to (round 3 indexed
        # the slots by the parity of the iteration alone: a late peer then missed a row and ran into its time limit).
        regs = []
        for rep, it in enumerate((7, 5, 7, 3, 0, 0)):
            if rank == rep % 2:
                torch.cuda._sleep(4_000_000)  # this rank's launches of the alignment start ~2 ms after the peer's
            pk = (sp.RegistrationParams(max_iterations=iters) if it == 0 else  # 0: default criteria, stops early
                  sp.RegistrationParams(criteria_translation=0.0, criteria_rotation=0.0, max_iterations=it))
            regk = sp.Registration(pk)
            Tk, _, _ = regk.align_fused_loop(Sh, prep, iterations=None if it == 0 else it, xchg=x)
            regs.append((regk, Tk.clone()))
        torch.cuda.synchronize()
        for rep, (regk, Tk) in enumerate(regs):
            res[f"Ts{rep}"] = Tk.cpu().numpy()
        for regk, _ in regs:
            regk.direct_status()  # raises when a row did not arrive in time
        res["stress_ok"] = np.float32(1)
        regg = sp.Registration(p)
        Tg_, _, _ = regg.align_fused_loop(Sh, prep, iterations=iters, group=dist.group.WORLD)  # the gloo row exchange
        torch.cuda.synchronize()
        res["Tgloo"] = Tg_.cpu().numpy()
    elif rank == 0:
        reg = sp.Registration(p)
        reg.align_fused_loop(Sh, prep, iterations=iters, xchg=x)
        torch.cuda.synchronize()  # returns: the wait is bounded
        try:
            reg.direct_status()
            res["err"] = np.float32(0)
        except sp.SpError as e:
            res["err"] = np.float32(1)
            assert "did not arrive" in str(e)
    dist.barrier()
    np.savez(out_path % rank, **res)
    del x
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_direct_exchange_two_processes_one_gpu(tmp_path):
    n, iters, world = 60000, 8, 2
    out = str(tmp_path / "direct%d.npz")
    mp.spawn(_direct_worker, args=(world, _free_port(), n, iters, out, False), nprocs=world, join=True)
    r0, r1 = np.load(out % 0), np.load(out % 1)
    for key in r0.files:
        assert np.array_equal(r0[key], r1[key]), key  # every rank holds the identical pose, count and iteration number
    assert np.array_equal(r0["T0"], r0["T1"])  # a second alignment over the same slots reproduces the first, bit for bit
    assert np.abs(r0["T0"] - r0["Tgloo"]).max() < 2e-6  # = the collective exchange's pose (the summation order differs)
    assert int(r0["inl0"]) == n and int(r0["it0"]) == iters
    assert 1 <= int(r0["itc"]) < iters and np.abs(r0["Tc"] - r0["T0"]).max() < 2e-3
    # the back-to-back series: same poses on both ranks (checked above for every key), repeats identical, all near the answer
    assert int(r0["stress_ok"]) == 1
    assert np.array_equal(r0["Ts0"], r0["Ts2"]) and np.array_equal(r0["Ts4"], r0["Ts5"])
    for rep in range(6):
        assert np.isfinite(r0[f"Ts{rep}"]).all() and np.abs(r0[f"Ts{rep}"] - r0["T0"]).max() < 5e-3, rep
    sp, S, Tg, T_gt = _make(n)
    prep = sp.PreparedTarget(sp.GridKNN.build(Tg.points), Tg.covs)
    reg = sp.Registration(sp.RegistrationParams(criteria_translation=0.0, criteria_rotation=0.0, max_iterations=iters))
    T_single, _, _ = reg.align_fused_loop(S, prep, iterations=iters)
    assert np.abs(r0["T0"] - T_single.cpu().numpy()).max() < 2e-6


@pytest.mark.timeout(600)
def test_direct_exchange_missing_peer_is_an_error_not_a_hang(tmp_path):
    n, iters, world = 60000, 4, 2
    out = str(tmp_path / "absent%d.npz")
    mp.spawn(_direct_worker, args=(world, _free_port(), n, iters, out, True), nprocs=world, join=True)
    assert int(np.load(out % 0)["err"]) == 1
