"""Runs the C++ facade tests and the example_registration equivalent (BASELINE config 1) on the GPU box.

The binaries are built by __graft_entry__.build() (tests/cpp/Makefile, host-only g++ against libsycl_points_amd.so)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")
GOLD = os.path.join(ROOT, "tests", "golden")


def _build():
    subprocess.check_call(["make", "-C", CPP, "-s"])


def test_cpp_facade_suite():
    _build()
    env = dict(os.environ, SP_GOLDEN_DIR=os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    r = subprocess.run([os.path.join(CPP, "test_facade")], capture_output=True, text=True, timeout=600, env=env)
    print(r.stdout[-3000:], r.stderr[-2000:])
    assert r.returncode == 0, r.stdout[-3000:]
    assert " 0 failed" in r.stdout


def read_ply_xyz(path):
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    n = int([l for l in head.split(b"\n") if l.startswith(b"element vertex")][0].split()[-1])
    a = np.frombuffer(body, dtype="<f4", count=n * 4).reshape(n, 4)
    pts = np.ones((n, 4), np.float32)
    pts[:, :3] = a[:, :3]
    return pts


def oracle_example(orc, src, tgt, stable):
    """The reference's example_registration flow (cpp/examples/example_registration.cpp:57-121) restated on the oracle."""
    from oracle.pyoracle import LOSS, OPT, REG, RegParams

    def prep(p):
        p = p[orc.box_filter(p, 0.5, 50.0) == 1]
        p = orc.voxel_downsample(p, 0.25, 1, stable=stable)["points"]
        nodes = orc.kdtree_build(p)
        idx, _ = orc.kdtree_knn(nodes, p, 10)
        return p, orc.cov_estimate(p, idx)

    s, sc = prep(src)
    t, tc = prep(tgt)
    keep = orc.random_sampling_flags(1234, len(s), 1000) == 1
    p = RegParams.defaults(reg_type=REG["GICP"], robust_type=LOSS["GEMAN_MCCLURE"], optimization_method=OPT["LM"],
                           max_iterations=10, max_correspondence_distance=2.0, robust_default_scale=10.0, auto_scale=1,
                           init_scale=10.0, min_scale=2.5, auto_scaling_iter=3)
    return orc.registration_align(p, s[keep], sc[keep], t, tc), len(s), len(t)


@pytest.mark.parametrize("knn", ["kdtree", "grid"])
def test_example_registration_config1(orc, knn):
    _build()
    src_ply, tgt_ply = os.path.join(GOLD, "source.ply"), os.path.join(GOLD, "target.ply")
    args = [os.path.join(CPP, "example_registration"), src_ply, tgt_ply, "1", "0"] + (["--grid"] if knn == "grid" else [])
    r = subprocess.run(args, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][0]
    T = np.array([float(x) for x in line.split()[1:]], np.float32).reshape(4, 4).T
    # Oracle with the device path's voxel summation order (ascending index): the whole pipeline agrees tightly.
    ref, ns, nt = oracle_example(orc, read_ply_xyz(src_ply), read_ply_xyz(tgt_ply), stable=True)
    assert f"source {ns}, target {nt}" in r.stdout          # same voxel counts as the oracle's downsampling
    assert np.abs(T - ref["T"]).max() < 2e-5, np.abs(T - ref["T"]).max()
    # Oracle with the reference's own (unstable std::sort) order: the 1000-point, early-stopping LM pipeline amplifies
    # the 1e-7 differences of the voxel means to ~2e-4 — in the oracle itself, not only on the GPU.
    ref2, _, _ = oracle_example(orc, read_ply_xyz(src_ply), read_ply_xyz(tgt_ply), stable=False)
    assert np.abs(T - ref2["T"]).max() < 1e-3
    # sanity against the bundled ground truth (cpp/data/T_target_source.txt; not a 1e-5 pin)
    T_gt = np.loadtxt(os.path.join(GOLD, "T_target_source.txt")).astype(np.float32)
    assert np.abs(T[:3, 3] - T_gt[:3, 3]).max() < 0.05 and np.abs(T[:3, :3] - T_gt[:3, :3]).max() < 0.01


def test_python_kdtree_takes_the_facades_decisions(orc):
    """sycl_points_amd.api.KDTree.build(points, accelerate=True) mirrors the facade's KDTree (knn.hpp): the same structure
    answers the same query on the same cloud as in tests/cpp/test_facade.cpp (kdtree_backend_on_the_bundled_scan,
    kdtree_self_knn_large_clouds) — the raw scan of surfaces goes to the device-built hierarchy, a large cloud of uniform
    density to the grid for 8 <= k <= 20, small clouds / k > 32 / a tree with removed nodes to the reference's tree — and the
    lists are the exact ones (oracle brute force) whichever structure answers."""
    import torch

    import sycl_points_amd.api as sp

    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    scan = dev(read_ply_xyz(os.path.join(GOLD, "target.ply")))
    tree = sp.KDTree.build(scan, accelerate=True)
    for k in (1, 10, 20, 32):
        assert tree.backend_for(scan, k) == "bvh"
    assert tree.backend_for(scan, 33) == "kdtree"
    sub = scan[:3000].contiguous()
    oi, od = orc.knn_bruteforce(sub.cpu().numpy(), scan.cpu().numpy(), 10)
    r = tree.knn_search(sub, 10)
    assert np.array_equal(r.distances.cpu().numpy(), od)  # (indices may differ inside groups of exactly equal distances)
    uni = dev(orc.rng(99).uniform_points(40000, 10.0))
    t2 = sp.KDTree.build(uni, accelerate=True)
    assert t2.backend_for(uni, 20) == "grid" and t2.backend_for(uni, 8) == "grid"
    assert t2.backend_for(uni, 5) == "bvh" and t2.backend_for(uni, 40) == "kdtree"
    assert t2.backend_for(uni[:100].contiguous(), 20) == "bvh"  # other queries than the tree's own cloud
    oi, od = orc.knn_bruteforce(uni.cpu().numpy()[:2000], uni.cpu().numpy(), 20)
    for k, q in ((20, uni), (5, uni)):
        r = t2.knn_search(q, k)
        assert np.array_equal(r.indices.cpu().numpy()[:2000], oi[:, :k]) and np.array_equal(r.distances.cpu().numpy()[:2000], od[:, :k])
    small = dev(orc.rng(5).uniform_points(800, 10.0))
    assert sp.KDTree.build(small, accelerate=True).backend_for(small, 10) == "kdtree"
    # a few thousand points (the example's downsampled scans): exact brute force, no hierarchy built
    mid = dev(orc.rng(6).uniform_points(6000, 10.0))
    tm = sp.KDTree.build(mid, accelerate=True)
    assert tm.backend_for(mid, 10) == "bruteforce" and tm.backend_for(mid, 20) == "bruteforce" and tm.backend_for(mid, 24) == "bvh"
    assert tm.backend_for(mid, 10, np.eye(4, dtype=np.float32)) == "bvh"
    oi, od = orc.knn_bruteforce(mid.cpu().numpy(), mid.cpu().numpy(), 10)
    r = tm.knn_search(mid, 10)
    assert tm._hier is None  # (never built)
    assert np.array_equal(r.indices.cpu().numpy(), oi) and np.array_equal(r.distances.cpu().numpy(), od)
    # after a lazy delete the hierarchy keeps answering (sp_bvh_remove_by_flags); the grid shortcut for the own cloud is gone
    flags = np.ones(40000, np.uint8)
    flags[::10] = 0  # (1 = keep, with the kept points' new indices: test_kdtree.cpp:459-512)
    new_idx = np.where(flags == 1, np.cumsum(flags) - 1, -1).astype(np.int32)
    t2.remove_nodes_by_flags(dev(flags), dev(new_idx))
    assert t2.backend_for(uni, 20) == "bvh"
    kept = uni.cpu().numpy()[flags == 1]
    oi, od = orc.knn_bruteforce(kept[:1500], kept, 20)
    r = t2.knn_search(dev(kept[:1500]), 20)
    assert np.array_equal(r.indices.cpu().numpy(), oi) and np.array_equal(r.distances.cpu().numpy(), od)


def test_cpp_harness_config4_matches_the_python_path(tmp_path):
    """examples/bench_registration.cpp: BASELINE config 4 (1 M vs 1 M, GICP, Gauss-Newton, 20 iterations, criteria 0) driven
    entirely through the header facade — VoxelGrid -> KDTree::build -> knn_search -> covariance::estimate_async ->
    Registration::align — with the reference harness's per-stage microseconds (cpp/examples/example_registration.cpp:126-161).
    The same clouds through the Python mirror give the same pose to 1e-5 (the two host layers drive the same kernels), and
    the facade's loop costs at most 10 % (+ the read-back of the result) more per iteration than the C ABI loop timed with
    HIP events: what a drop-in C++ caller pays at 1 M points."""
    import torch

    import sycl_points_amd.api as sp
    from sycl_points_amd.synthetic import gicp_pair

    _build()
    n = 1_000_000
    src, tgt, T_gt = gicp_pair(n, 10.0)
    sp_path, tp_path = str(tmp_path / "src.bin"), str(tmp_path / "tgt.bin")
    np.ascontiguousarray(src, np.float32).tofile(sp_path)
    np.ascontiguousarray(tgt, np.float32).tofile(tp_path)
    r = subprocess.run([os.path.join(CPP, "bench_registration"), "--points", sp_path, tp_path, "--loops", "5", "--warmup", "2"],
                       capture_output=True, text=True, timeout=900)
    print(r.stdout[-3000:], r.stderr[-1500:])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "first and second alignment give the same pose: yes" in r.stdout
    T_cpp = np.array([float(x) for x in [l for l in r.stdout.splitlines() if l.startswith("RESULT")][0].split()[1:]],
                     np.float32).reshape(4, 4).T
    us_cpp = float([l for l in r.stdout.splitlines() if l.startswith("US_PER_ITERATION ")][0].split()[1])

    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    vg = sp.VoxelGrid(0.02)
    S = vg.downsampling(sp.PointCloudShared(dev(src)))
    Tg = vg.downsampling(sp.PointCloudShared(dev(tgt)))
    assert f"downsampled: source {S.size()}, target {Tg.size()}" in r.stdout
    for c in (S, Tg):
        c.covs = sp.GridKNN.build(c.points, points_per_cell=6.0).self_knn(20, want_knn=False, want_covs=True)[1]
    prep = sp.PreparedTarget(sp.GridKNN.build(Tg.points, points_per_cell=0.5), Tg.covs)
    p = sp.RegistrationParams(criteria_translation=0.0, criteria_rotation=0.0, max_iterations=20)
    reg = sp.Registration(p)
    T_dev = torch.zeros(16, dtype=torch.float32, device="cuda")
    T_ident = torch.eye(4, dtype=torch.float32, device="cuda").reshape(-1).contiguous()
    delta = torch.zeros(8, dtype=torch.float32, device="cuda")
    ms = []
    for _ in range(8):
        T_dev.copy_(T_ident)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        reg.align_fused_loop(S, prep, T_dev=T_dev, delta_dev=delta, prepare=True, sort_by_cell="presorted")
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    us_py = 1e3 * float(np.median(ms[2:])) / 20
    T_py = reg.T_from_device(T_dev)
    assert np.abs(T_cpp - T_py).max() < 1e-5, np.abs(T_cpp - T_py).max()
    assert np.abs(T_cpp - T_gt).max() < 2e-4
    print(f"us per iteration: facade {us_cpp:.2f} (host clock, incl. uploads of the pose and the read-back), C ABI {us_py:.2f} (HIP events)")
    # one alignment = 20 iterations; the facade adds a 64-byte upload, a 112 + 192-byte read-back and a stream synchronisation
    # per ALIGNMENT (measured 20-40 us on the host clock): allow 10 % + 2 us per iteration
    assert us_cpp <= 1.10 * us_py + 2.0, (us_cpp, us_py)
