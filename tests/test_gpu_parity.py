"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): bit-exact for KNN indices / distances, voxel keys, covariances (pure +,*,/ arithmetic
with the reference's fma chains); tolerance for anything that goes through acosf/cosf (eigen-decomposition) or a
re-ordered fp32 reduction; final SE(3) within 1e-5.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
FLT_MAX = np.finfo(np.float32).max


@pytest.fixture(scope="module")
def sp():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device (no CPU fallback exists)")
    import sycl_points_amd.api as api

    return api


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def cloud(orc, seed, n, r=10.0):
    return orc.rng(seed).uniform_points(n, r)


# ------------------------------------------------------------------ K1 brute force
@pytest.mark.parametrize("k", [1, 3, 5, 10, 20])
def test_bruteforce_matches_oracle_bitwise(sp, orc, k):
    g = orc.rng(1234)
    tgt = g.uniform_points(5000, 10.0)
    qry = g.uniform_points(777, 10.0)  # ragged: not a multiple of the workgroup
    r = sp.knn_search_bruteforce(dev(qry), dev(tgt), k)
    oi, od = orc.knn_bruteforce(qry, tgt, k)
    assert np.array_equal(r.indices.cpu().numpy(), oi)
    assert np.array_equal(r.distances.cpu().numpy(), od)


def test_bruteforce_ties_lowest_index_wins(sp, orc):
    # duplicated targets: strict '<' keeps the lowest index (bruteforce.hpp:71-83), across LDS tiles and chunks
    base = cloud(orc, 7, 1500)
    tgt = np.concatenate([base, base, base])  # every point three times, 1500 apart (crosses the 1024 tile)
    qry = base[:300].copy()
    for k in (1, 4):
        r = sp.knn_search_bruteforce(dev(qry), dev(tgt), k)
        oi, od = orc.knn_bruteforce(qry, tgt, k)
        assert np.array_equal(r.indices.cpu().numpy(), oi)
        assert np.array_equal(r.distances.cpu().numpy(), od)
    assert (oi[:, 0] == np.arange(300)).all()


def test_bruteforce_fewer_targets_than_k_and_empty(sp, orc):
    tgt = cloud(orc, 3, 3)
    qry = cloud(orc, 4, 10)
    r = sp.knn_search_bruteforce(dev(qry), dev(tgt), 5)
    oi, od = orc.knn_bruteforce(qry, tgt, 5)
    assert np.array_equal(r.indices.cpu().numpy(), oi) and np.array_equal(r.distances.cpu().numpy(), od)
    assert (oi[:, 3:] == -1).all() and (od[:, 3:] == FLT_MAX).all()
    r = sp.knn_search_bruteforce(dev(qry), dev(np.zeros((0, 4), np.float32)), 2)
    assert (r.indices.cpu().numpy() == -1).all() and (r.distances.cpu().numpy() == FLT_MAX).all()
    r = sp.knn_search_bruteforce(dev(np.zeros((0, 4), np.float32)), dev(tgt), 2)
    assert r.indices.shape == (0, 2)
    with pytest.raises(sp.SpError):
        sp.knn_search_bruteforce(dev(qry), dev(tgt), 21)  # MAX_K = 20


def test_bruteforce_config2_100k_k1(sp, orc):
    # BASELINE config 2 at full size: 100k x 100k, k = 1. Oracle on a 2000-query sample; size-independent
    # property on all: the reported distance equals the distance to the reported index, recomputed on the host.
    g = orc.rng(1234)
    tgt = g.uniform_points(100000, 10.0)
    qry = g.uniform_points(100000, 10.0)
    r = sp.knn_search_bruteforce(dev(qry), dev(tgt), 1)
    idx = r.indices.cpu().numpy()[:, 0]
    d2 = r.distances.cpu().numpy()[:, 0]
    sel = np.arange(0, 100000, 50)
    oi, od = orc.knn_bruteforce(qry[sel], tgt, 1)
    assert np.array_equal(idx[sel], oi[:, 0]) and np.array_equal(d2[sel], od[:, 0])
    diff = qry[:, :3].astype(np.float64) - tgt[idx, :3].astype(np.float64)
    assert np.allclose((diff**2).sum(1), d2, rtol=1e-5)
    assert (idx >= 0).all() and (idx < 100000).all()


@pytest.mark.parametrize("k", [2, 5, 10, 20])
def test_bruteforce_two_pass_path_matches_oracle_bitwise(sp, orc, k):
    # >= 16 LDS tiles of targets and at least k chunks: the bound-then-collect path (knn_bruteforce.hip). Every fourth target
    # is a duplicate of an earlier one (ties across tiles and chunks, lowest index first), the target count is ragged, a
    # query far outside the cloud and one with a NaN coordinate ride along.
    g = orc.rng(99)
    base = g.uniform_points(30001, 10.0)
    tgt = np.concatenate([base, base[::3][:10000], g.uniform_points(7, 10.0)])
    qry = np.concatenate([g.uniform_points(1500, 10.0), base[:200], np.float32([[500.0, -300.0, 40.0, 1.0]])])
    r = sp.knn_search_bruteforce(dev(qry), dev(tgt), k)
    oi, od = orc.knn_bruteforce(qry, tgt, k)
    assert np.array_equal(r.indices.cpu().numpy(), oi)
    assert np.array_equal(r.distances.cpu().numpy(), od)
    bad = qry[:64].copy()
    bad[5, 1] = np.nan
    r = sp.knn_search_bruteforce(dev(bad), dev(tgt), k)
    oi, od = orc.knn_bruteforce(bad, tgt, k)
    assert np.array_equal(r.indices.cpu().numpy(), oi) and np.array_equal(r.distances.cpu().numpy(), od)


@pytest.mark.parametrize("k", [2, 20])
def test_bruteforce_two_pass_overflow_paths(sp, orc, k):
    # the collect path keeps at most 64 candidates per query and lists a query for at most 32 chunks; beyond either, the
    # query's wave rescans all targets. One point repeated 600 times all over the target array (ties in > 32 chunks), another
    # 150 times inside two chunks (> 64 candidates from few chunks), queries on and near both, and targets with inf / NaN.
    g = orc.rng(77)
    tgt = g.uniform_points(40000, 10.0)
    a, b = np.float32([1.25, -2.5, 3.0, 1.0]), np.float32([-4.0, 0.5, 0.25, 1.0])
    tgt[np.arange(600) * 61 + 7] = a
    tgt[2048 + np.arange(150) * 9] = b
    tgt[5] = [np.inf, 0.0, 0.0, 1.0]
    tgt[3000] = [np.nan, 1.0, 1.0, 1.0]
    qry = np.concatenate([g.uniform_points(300, 10.0), a[None], b[None], (a + np.float32([0.01, 0, 0, 0]))[None],
                          (b + np.float32([0, 0.02, 0, 0]))[None]])
    r = sp.knn_search_bruteforce(dev(qry), dev(tgt), k)
    oi, od = orc.knn_bruteforce(qry, tgt, k)
    assert np.array_equal(r.indices.cpu().numpy(), oi)
    assert np.array_equal(r.distances.cpu().numpy(), od)
    # every target the same point: all distances tie, the first k indices win
    same = np.tile(a, (20000, 1))
    r = sp.knn_search_bruteforce(dev(qry[:70]), dev(same), k)
    oi, od = orc.knn_bruteforce(qry[:70], same, k)
    assert np.array_equal(r.indices.cpu().numpy(), oi) and np.array_equal(r.distances.cpu().numpy(), od)


@pytest.mark.parametrize("k", [1, 3, 20])
def test_bruteforce_approximate_bound_is_safe(sp, orc, k):
    # pass A of the large-problem path bounds the k-th distance with an approximate expression and a proven error term
    # (knn_bf_chunkmin_kernel): clouds far from the origin (the term is relative to the CENTRED extent), a thin slab, targets
    # on a lattice (many exactly equal distances), one target 10^30 away (the bound is useless: every query is rescanned).
    g = orc.rng(2024)
    off = np.float32([1000.0, -2000.0, 50.0, 0.0])
    tgt = g.uniform_points(20000, 10.0) + off
    qry = np.concatenate([g.uniform_points(700, 10.0) + off, np.float32([[0.0, 0.0, 0.0, 1.0]]), tgt[:50]])
    for T, Q in ((tgt, qry),
                 (tgt * np.float32([1, 1, 1e-3, 1]), qry * np.float32([1, 1, 1e-3, 1])),
                 (np.round(tgt), np.round(qry[:300]) + np.float32([0.5, 0, 0, 0]))):
        r = sp.knn_search_bruteforce(dev(Q), dev(T), k)
        oi, od = orc.knn_bruteforce(Q, T, k)
        assert np.array_equal(r.indices.cpu().numpy(), oi)
        assert np.array_equal(r.distances.cpu().numpy(), od)
    far = tgt.copy()
    far[77] = [1e30, 0.0, 0.0, 1.0]
    r = sp.knn_search_bruteforce(dev(qry[:40]), dev(far), k)
    oi, od = orc.knn_bruteforce(qry[:40], far, k)
    assert np.array_equal(r.indices.cpu().numpy(), oi) and np.array_equal(r.distances.cpu().numpy(), od)


@pytest.mark.parametrize("valu", [0, 1])
def test_bruteforce_pass_a_error_stays_inside_its_bound(sp, orc, valu):
    # The chunk minima of pass A (matrix cores: bf16-split operands; or packed VALU) are approximate by construction; the
    # bound kernel adds E = c (|q - centre| + pmax)^2 with c = 2.0e-5 / 20 * 2^-24 (knn_bruteforce.hip). Read the minima back
    # from the workspace (first G x nq floats, G chunks of `chunk` targets) and compare with float64 chunk minima: the error
    # must stay inside E — with margin, or the constant is not conservative — and the lists must not depend on the form.
    import ctypes as C
    from sycl_points_amd import _lib
    L = _lib.lib()
    g = orc.rng(5)
    nt, nq, k = 40000, 3000, 5
    tgt = g.uniform_points(nt, 10.0) + np.float32([300.0, -100.0, 20.0, 0.0])
    qry = g.uniform_points(nq, 12.0) + np.float32([300.0, -100.0, 20.0, 0.0])
    T, Q = dev(tgt), dev(qry)
    nbytes = L.sp_knn_bruteforce_workspace_bytes(nq, nt, k)
    ws = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    idx = torch.empty((nq, k), dtype=torch.int32, device="cuda")
    d2 = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    _lib.check(L.sp_knn_bruteforce_set_pass_a(valu))
    try:
        _lib.check(L.sp_knn_bruteforce(sp._ptr(Q), nq, sp._ptr(T), nt, k, sp._ptr(idx), sp._ptr(d2), sp._ptr(ws), nbytes, sp._stream()))
        torch.cuda.synchronize()
    finally:
        _lib.check(L.sp_knn_bruteforce_set_pass_a(0))
    oi, od = orc.knn_bruteforce(qry, tgt, k)
    assert np.array_equal(idx.cpu().numpy(), oi) and np.array_equal(d2.cpu().numpy(), od)
    chunk = -(-(-(-nt // 256)) // 256) * 256  # (plan_bounded: at most 256 chunks, whole multiples of 256 targets)
    G = -(-nt // chunk)
    amin = ws[: G * nq * 4].view(torch.float32).reshape(G, nq).cpu().numpy().astype(np.float64)
    t64, q64 = tgt[:, :3].astype(np.float64), qry[:, :3].astype(np.float64)
    lo, hi = tgt[:, :3].min(0).astype(np.float64), tgt[:, :3].max(0).astype(np.float64)
    centre = 0.5 * (lo + hi)
    pmax = np.linalg.norm(np.maximum(hi - centre, centre - lo))
    coeff = 20.0 * 2.0**-24 if valu else 2.0e-5
    E = coeff * (np.linalg.norm(q64 - centre, axis=1) + pmax) ** 2
    worst = 0.0
    for c in range(G):
        # (up to 65536 targets the chunks are interleaved — chunk c = targets c, c + G, c + 2G, ... — so that each is a uniform
        # sample of the cloud whatever its storage order: run_bounded, knn_bruteforce.hip)
        blk = t64[c::G] if nt <= 65536 else t64[c * chunk:(c + 1) * chunk]
        ref = ((q64[:, None, :] - blk[None, :, :]) ** 2).sum(2).min(1)
        worst = max(worst, float((np.abs(amin[c] - ref) / E).max()))
    assert worst < 0.6, worst


@pytest.mark.parametrize("n, k", [(2048, 8), (2560, 10), (6000, 10), (6000, 20), (16384, 20), (30000, 5)])
def test_bruteforce_small_spatially_ordered_clouds(sp, orc, n, k):
    # Clouds of a few thousand points take the two-pass search too (from 2048 targets that split into k chunks), with
    # interleaved chunks; a cloud in spatial order (sorted by voxel key, as downsampling leaves it) with duplicates is the
    # case contiguous chunks handled worst. Lists bit-identical to the oracle's.
    g = orc.rng(n + k)
    pts = g.uniform_points(n, 10.0)
    pts[:, 2] = np.round(pts[:, 2] * 0.05) / 0.05 * 0.01  # a few sheets: surfaces, not a filled box
    pts[::97] = pts[5]                                      # duplicates: distance ties, lowest index first
    order = np.lexsort((pts[:, 0], pts[:, 1], pts[:, 2]))
    pts = np.ascontiguousarray(pts[order])
    q = np.ascontiguousarray(pts[:: max(1, n // 1500)])
    r = sp.knn_search_bruteforce(dev(q), dev(pts), k)
    oi, od = orc.knn_bruteforce(q, pts, k)
    assert np.array_equal(r.indices.cpu().numpy(), oi) and np.array_equal(r.distances.cpu().numpy(), od)


@pytest.mark.parametrize("nt, nq, k", [(256, 1, 1), (700, 333, 5), (6096, 6096, 10), (6096, 1000, 20), (12032, 2500, 10),
                                       (3001, 4099, 3)])
def test_bruteforce_small_cloud_in_one_launch(sp, orc, nt, nq, k):
    # Target clouds of 256 .. 12032 points: the whole cloud in LDS, one launch, a wave per four queries (lane minima -> bound
    # of the k-th distance -> the targets within it -> 64-lane sort). Surfaces with duplicates and a block of identical points
    # (more than 64 targets at the k-th distance: the kernel's overflow path). Bit-identical to the oracle and to the general
    # paths of the same library (switch 2 of sp_knn_bruteforce_set_pass_a).
    from sycl_points_amd import _lib

    g = orc.rng(nt + nq + k)
    pts = g.uniform_points(nt, 10.0)
    pts[:, 2] = np.round(pts[:, 2] * 0.05) / 0.05 * 0.01
    pts[::53] = pts[7]
    if nt >= 700:
        pts[100:260] = pts[100]  # 160 identical points
    q = g.uniform_points(nq, 10.0)
    q[: min(nq, 64)] = pts[: min(nq, 64)]  # queries ON targets (distance 0, and the block of identical points)
    if nq > 150:
        q[100:150] = pts[100]
    oi, od = orc.knn_bruteforce(q, pts, k)
    L = _lib.lib()
    r = sp.knn_search_bruteforce(dev(q), dev(pts), k)
    assert np.array_equal(r.indices.cpu().numpy(), oi) and np.array_equal(r.distances.cpu().numpy(), od)
    _lib.check(L.sp_knn_bruteforce_set_pass_a(2))
    try:
        general = sp.knn_search_bruteforce(dev(q), dev(pts), k)
    finally:
        _lib.check(L.sp_knn_bruteforce_set_pass_a(3))
    assert torch.equal(general.indices, r.indices) and torch.equal(general.distances, r.distances)


def test_bruteforce_config2_size_k20(sp, orc):
    # 100k x 100k at k = 20 (the reference's MAX_K): oracle on a 500-query sample, size-independent properties on all
    # (ascending distances, distance == distance to the reported index, no index twice).
    g = orc.rng(4321)
    tgt = g.uniform_points(100000, 10.0)
    qry = g.uniform_points(100000, 10.0)
    r = sp.knn_search_bruteforce(dev(qry), dev(tgt), 20)
    idx = r.indices.cpu().numpy()
    d2 = r.distances.cpu().numpy()
    sel = np.arange(0, 100000, 200)
    oi, od = orc.knn_bruteforce(qry[sel], tgt, 20)
    assert np.array_equal(idx[sel], oi) and np.array_equal(d2[sel], od)
    assert (np.diff(d2, axis=1) >= 0).all() and (idx >= 0).all() and (idx < 100000).all()
    diff = qry[:, None, :3].astype(np.float64)[::10] - tgt[idx[::10], :3].astype(np.float64)
    assert np.allclose((diff**2).sum(2), d2[::10], rtol=1e-5)
    srt = np.sort(idx, axis=1)
    assert (np.diff(srt, axis=1) > 0).all()


# ------------------------------------------------------------------ K2/K3/K4 KD-tree
@pytest.mark.parametrize("k", [1, 3, 5, 10, 20, 30])
def test_kdtree_matches_oracle_bitwise(sp, orc, k):
    g = orc.rng(1234)
    tgt = g.uniform_points(20000, 10.0)
    qry = g.uniform_points(3001, 10.0)
    tree = sp.KDTree.build(dev(tgt))
    r = tree.knn_search(dev(qry), k)
    nodes = orc.kdtree_build(tgt)
    oi, od = orc.kdtree_knn(nodes, qry, k)
    assert np.array_equal(r.indices.cpu().numpy(), oi)
    assert np.array_equal(r.distances.cpu().numpy(), od)


def test_kdtree_with_transform_and_device_pose(sp, orc):
    g = orc.rng(99)
    tgt = g.uniform_points(30000, 10.0)
    qry = g.uniform_points(5000, 10.0)
    T = orc.se3_exp([0.05, -0.02, 0.03, 0.3, -0.2, 0.1])
    tree = sp.KDTree.build(dev(tgt))
    nodes = orc.kdtree_build(tgt)
    oi, od = orc.kdtree_knn(nodes, qry, 1, T)
    r = tree.knn_search(dev(qry), 1, T)
    assert np.array_equal(r.indices.cpu().numpy(), oi) and np.array_equal(r.distances.cpu().numpy(), od)
    T_dev = dev(np.ascontiguousarray(T.T).reshape(-1))  # column-major on the device
    r2 = tree.knn_search(dev(qry), 1, T_dev)
    assert np.array_equal(r2.indices.cpu().numpy(), oi) and np.array_equal(r2.distances.cpu().numpy(), od)


def test_kdtree_small_and_edge_cases(sp, orc):
    # SinglePoint known answer (test_kdtree.cpp:358-389)
    tree = sp.KDTree.build(dev(np.array([[0, 0, 0, 1]], np.float32)))
    r = tree.knn_search(dev(np.array([[1, 1, 1, 1]], np.float32)), 1)
    assert int(r.indices[0, 0]) == 0 and abs(float(r.distances[0, 0]) - 3.0) < 1e-6
    # sizes {10,100,500} x {5,20}, k = 3 (test_kdtree.cpp:320-355), incl. k > number of points
    g = orc.rng(5)
    for nt in (2, 10, 100, 500):
        for nq in (5, 20):
            tgt = g.uniform_points(nt, 10.0)
            qry = g.uniform_points(nq, 10.0)
            r = sp.KDTree.build(dev(tgt)).knn_search(dev(qry), 3)
            oi, od = orc.kdtree_knn(orc.kdtree_build(tgt), qry, 3)
            assert np.array_equal(r.indices.cpu().numpy(), oi) and np.array_equal(r.distances.cpu().numpy(), od)
    # empty tree, empty queries, k too large
    empty = sp.KDTree.build(dev(np.zeros((0, 4), np.float32)))
    r = empty.knn_search(dev(cloud(orc, 1, 4)), 2)
    assert (r.indices.cpu().numpy() == -1).all() and (r.distances.cpu().numpy() == FLT_MAX).all()
    r = tree.knn_search(dev(np.zeros((0, 4), np.float32)), 2)
    assert r.indices.shape[0] == 0
    with pytest.raises(sp.SpError):
        tree.knn_search(dev(cloud(orc, 1, 4)), 101)


def test_kdtree_duplicates_same_tie_rule(sp, orc):
    base = cloud(orc, 21, 4000)
    tgt = np.concatenate([base, base])  # exact ties: first visited wins, identical traversal -> identical answer
    tree = sp.KDTree.build(dev(tgt))
    nodes = orc.kdtree_build(tgt)
    for k in (1, 5):
        r = tree.knn_search(dev(base[:1000]), k)
        oi, od = orc.kdtree_knn(nodes, base[:1000], k)
        assert np.array_equal(r.indices.cpu().numpy(), oi) and np.array_equal(r.distances.cpu().numpy(), od)


def test_kdtree_radius_and_remove_by_flags(sp, orc):
    g = orc.rng(1234)
    tgt = g.uniform_points(1000, 10.0)
    qry = g.uniform_points(100, 10.0)
    tree = sp.KDTree.build(dev(tgt))
    nodes = orc.kdtree_build(tgt)
    res = sp.KNNResult()
    tree.radius_search_async(dev(qry), 10, 5.0, res)
    oi, od = orc.kdtree_radius(nodes, qry, 10, 5.0)
    assert np.array_equal(res.indices.cpu().numpy(), oi) and np.array_equal(res.distances.cpu().numpy(), od)
    tree.radius_search_async(dev(qry), 5, 0.05, res)
    oi, od = orc.kdtree_radius(nodes, qry, 5, 0.05)
    assert np.array_equal(res.indices.cpu().numpy(), oi) and np.array_equal(res.distances.cpu().numpy(), od)
    # lazy delete (test_kdtree.cpp:459-512)
    flags = np.ones(1000, np.uint8)
    flags[::10] = 0
    new_idx = np.where(flags == 1, np.cumsum(flags) - 1, -1).astype(np.int32)
    tree.remove_nodes_by_flags(dev(flags), dev(new_idx))
    orc.kdtree_remove_by_flags(nodes, flags, new_idx)
    removed = tgt[flags == 1]
    r = tree.knn_search(dev(removed), 10)
    oi, od = orc.kdtree_knn(nodes, removed, 10)
    assert np.array_equal(r.indices.cpu().numpy(), oi) and np.array_equal(r.distances.cpu().numpy(), od)
    bi, bd = orc.knn_bruteforce(removed, removed, 10)
    assert np.array_equal(od, bd)


def test_grid_radius_search_matches_kdtree_oracle(sp, orc):
    """GridKNN::radius_search_async: the max_k nearest within the radius, bit-identical to the reference's KD-tree radius
    search (kdtree.hpp:574-719) as the oracle restates it — with a query transform, radii that cut rows short, none or all."""
    g = orc.rng(4321)
    tgt = g.uniform_points(5000, 5.0)
    qry = g.uniform_points(400, 5.5)
    T = orc.se3_exp([0.03, -0.02, 0.01, 0.1, -0.05, 0.02])
    grid = sp.GridKNN.build(dev(tgt))
    nodes = orc.kdtree_build(tgt)
    for max_k, radius in ((10, 0.8), (20, 0.5), (5, 0.05), (3, 100.0), (1, 0.3)):
        res = grid.radius_search(dev(qry), max_k, radius, transT=T)
        oi, od = orc.kdtree_radius(nodes, orc.transform_points(qry, T), max_k, radius)
        assert np.array_equal(res.indices.cpu().numpy(), oi), (max_k, radius)
        assert np.array_equal(res.distances.cpu().numpy(), od), (max_k, radius)
    assert (grid.radius_search(dev(qry), 5, 0.05).indices.cpu().numpy() == -1).mean() > 0.5  # most rows are cut short
    with pytest.raises(sp.SpError):
        grid.radius_search(dev(qry), 21, 1.0)
    assert grid.radius_search(dev(qry[:0]), 5, 1.0).indices.numel() == 0


def test_grid_remove_nodes_by_flags(sp, orc):
    """GridKNN::remove_nodes_by_flags (the grid's lazy delete, test_kdtree.cpp:459-512 for the KD-tree): afterwards the
    grid answers exactly like a search over the kept points under their new indices — k-NN, k = 1 staged search, radius
    search and the self-kNN tiling — and like the KD-tree after the same removal."""
    g = orc.rng(99)
    tgt = g.uniform_points(30000, 4.0)
    flags = np.ones(len(tgt), np.uint8)
    flags[::3] = 0
    flags[1000:3000] = 0  # a removed region too: empty cells
    new_idx = np.where(flags == 1, np.cumsum(flags) - 1, -1).astype(np.int32)
    kept = tgt[flags == 1]
    grid = sp.GridKNN.build(dev(tgt))
    grid.remove_nodes_by_flags(dev(flags), dev(new_idx))
    torch.cuda.synchronize()
    qry = g.uniform_points(2000, 4.5)
    for k in (1, 5, 20):
        r = grid.knn_search(dev(qry), k)
        bi, bd = orc.knn_bruteforce(qry, kept, k)
        assert np.array_equal(r.indices.cpu().numpy(), bi) and np.array_equal(r.distances.cpu().numpy(), bd), k
    nodes = orc.kdtree_build(tgt)
    orc.kdtree_remove_by_flags(nodes, flags, new_idx)
    oi, od = orc.kdtree_knn(nodes, qry, 10)
    r = grid.knn_search(dev(qry), 10)
    assert np.array_equal(r.indices.cpu().numpy(), oi) and np.array_equal(r.distances.cpu().numpy(), od)
    rr = grid.radius_search(dev(qry), 8, 0.3)
    oi, od = orc.kdtree_radius(nodes, qry, 8, 0.3)
    assert np.array_equal(rr.indices.cpu().numpy(), oi) and np.array_equal(rr.distances.cpu().numpy(), od)
    # the self-kNN tiling was rebuilt: neighbour lists of the kept points among themselves
    knn, _, _ = grid.self_knn(6, want_knn=True, want_covs=False)
    bi, bd = orc.knn_bruteforce(kept, kept, 6)
    assert np.array_equal(knn.indices.cpu().numpy(), bi) and np.array_equal(knn.distances.cpu().numpy(), bd)
    # removing everything leaves an empty, usable grid
    grid.remove_nodes_by_flags(dev(np.zeros(len(kept), np.uint8)), dev(np.full(len(kept), -1, np.int32)))
    r = grid.knn_search(dev(qry[:10]), 3)
    assert (r.indices.cpu().numpy() == -1).all()
    with pytest.raises(sp.SpError):
        grid.remove_nodes_by_flags(dev(np.zeros(4, np.uint8)), dev(np.zeros(5, np.int32)))


def test_buffer_pool_reuse_across_sizes(sp, orc):
    """Structure builds take their buffers from a pool inside the library (csrc/sp_common.h): many builds and destroys of
    different sizes, interleaved and with objects alive in between, must keep answering exactly."""
    g = orc.rng(2024)
    alive = []
    for it, n in enumerate([5000, 300, 20000, 1, 7000, 64, 12000, 5000, 2, 9000, 150, 20000]):
        tgt = g.uniform_points(n, 3.0)
        qry = g.uniform_points(200, 3.2)
        grid = sp.GridKNN.build(dev(tgt), points_per_cell=[0.5, 2.0, 6.0][it % 3])
        k = min(5, n)
        r = grid.knn_search(dev(qry), k)
        bi, bd = orc.knn_bruteforce(qry, tgt, k)
        assert np.array_equal(r.indices.cpu().numpy(), bi) and np.array_equal(r.distances.cpu().numpy(), bd), (it, n)
        if n >= 64:
            knn, covs, _ = grid.self_knn(min(10, n), want_knn=True, want_covs=True)
            si, sd = orc.knn_bruteforce(tgt, tgt, min(10, n))
            assert np.array_equal(knn.indices.cpu().numpy(), si), (it, n)
            prep = sp.PreparedTarget(grid, covs)
            alive.append((grid, prep, tgt, qry))
        if it % 4 == 3:
            alive = alive[-1:]  # destroy most of what is alive: their arrays go back to the pool
    for grid, prep, tgt, qry in alive:  # survivors still answer from their own, untouched arrays
        r = grid.knn_search(dev(qry), 3)
        bi, bd = orc.knn_bruteforce(qry, tgt, 3)
        assert np.array_equal(r.indices.cpu().numpy(), bi) and np.array_equal(r.distances.cpu().numpy(), bd)


def test_kdtree_1m_k1_sampled(sp, orc):
    # full-size NN (1M targets): oracle KD-tree on a sample of queries + exactness property vs brute force on a sample
    g = orc.rng(1234)
    tgt = g.uniform_points(1000000, 10.0)
    qry = tgt[::100].copy()
    qry[:, :3] += orc.rng(5).normal(3 * len(qry), 0.02).reshape(-1, 3)
    tree = sp.KDTree.build(dev(tgt))
    r = tree.knn_search(dev(qry), 1)
    oi, od = orc.kdtree_knn(orc.kdtree_build(tgt), qry, 1)
    assert np.array_equal(r.indices.cpu().numpy(), oi) and np.array_equal(r.distances.cpu().numpy(), od)
    bi, bd = orc.knn_bruteforce(qry[:200], tgt, 1)
    assert np.array_equal(oi[:200], bi) and np.array_equal(od[:200], bd)


# ------------------------------------------------------------------ K5/K6/K7 covariance & normals
def test_covariance_bit_exact_and_normals(sp, orc):
    pts = cloud(orc, 1234, 20000, 3.0)
    nodes = orc.kdtree_build(pts)
    idx, _ = orc.kdtree_knn(nodes, pts, 20)
    idx[5, 3:] = -1            # fewer than 4 valid neighbours -> identity
    idx[6, 10:] = -1           # partially filled row
    covs = sp.covariance.estimate(dev(idx), dev(pts)).cpu().numpy()
    ocov = orc.cov_estimate(pts, idx)
    assert np.array_equal(covs, ocov)
    assert np.array_equal(covs[5].reshape(4, 4), np.diag([1, 1, 1, 0]).astype(np.float32))
    nrm = sp.covariance.estimate_normals(dev(idx), dev(pts)).cpu().numpy()
    onrm = orc.normals_from_knn(pts, idx)
    # smallest-eigenvalue eigenvector through acosf/cosf: a few ulp of angle; compare direction
    dots = np.abs((nrm[:, :3] * onrm[:, :3]).sum(1))
    assert np.percentile(dots, 1) > 1 - 1e-4 and (nrm[:, 3] == 0).all()
    assert np.mean(np.sign((nrm[:, :3] * onrm[:, :3]).sum(1)) > 0) > 0.999  # same flip rule
    nrm2 = sp.covariance.extract_normals(dev(pts), dev(ocov)).cpu().numpy()
    assert np.allclose(nrm2, nrm, atol=1e-6)
    plane = sp.covariance.update_covariance_plane(dev(ocov)).cpu().numpy()
    oplane = orc.update_covariance_plane(ocov)
    assert np.allclose(plane, oplane, atol=2e-5)


# ------------------------------------------------------------------ K9 + voxel grid
def test_voxel_keys_bit_exact(sp, orc):
    pts = cloud(orc, 1234, 200000, 10.0)
    pts[5, 0] = np.nan
    pts[6, 1] = np.inf
    pts[7, 2] = 3e6  # out of the 21-bit range
    for vs in (0.1, 0.25, 1.0):
        keys = sp.VoxelGrid(vs).compute_voxel_bit(dev(pts)).cpu().numpy().view(np.uint64)
        assert np.array_equal(keys, orc.voxel_keys(pts, vs))
    assert keys[5] == keys[6] == keys[7] == np.uint64(0xFFFFFFFFFFFFFFFF)


@pytest.mark.parametrize("r,vs,minc", [(10.0, 0.5, 1), (2.5, 0.25, 1), (2.5, 0.5, 3)])
def test_voxel_downsample_matches_oracle(sp, orc, r, vs, minc):
    n = 100000
    pts = cloud(orc, 1234, n, r)
    pts[11, 0] = np.nan
    rs = np.random.RandomState(0)
    rgb = rs.uniform(0, 1, (n, 4)).astype(np.float32)
    inten = rs.uniform(0, 255, n).astype(np.float32)
    ts = rs.uniform(0, 100, n).astype(np.float32)
    vg = sp.VoxelGrid(vs)
    vg.set_min_voxel_count(minc)
    pc = sp.PointCloudShared(dev(pts), rgb=dev(rgb), intensities=dev(inten), timestamp_offsets=dev(ts))
    out, keys = vg.downsampling(pc, return_keys=True)
    o = orc.voxel_downsample(pts, vs, minc, rgb, inten, ts, stable=True)
    assert np.array_equal(keys.cpu().numpy().view(np.uint64), o["keys"])      # voxel hash indices: bit-exact
    assert np.array_equal(out.points.cpu().numpy(), o["points"])             # same summation order -> bit-exact
    assert np.array_equal(out.rgb.cpu().numpy(), o["rgb"])
    assert np.array_equal(out.timestamp_offsets.cpu().numpy(), o["timestamps"])
    assert np.array_equal(out.intensities.cpu().numpy(), o["intensities"])
    # against the reference's own (unstable std::sort) order: same voxels, means equal to rounding
    o2 = orc.voxel_downsample(pts, vs, minc, rgb, inten, ts, stable=False)
    assert np.array_equal(o2["keys"], o["keys"])
    assert np.allclose(out.points.cpu().numpy(), o2["points"], atol=1e-5)


@pytest.mark.parametrize("attrs", ["none", "rgb", "ts", "rgb+ts+intensity"])
def test_voxel_runs_of_thousands_of_points(sp, orc, attrs):
    """Voxels of a few to several thousand points (a scan's near range): the sequential sums of runs that span waves, the
    workgroup's 256 positions and several 256-member rounds of the fetch past them — every attribute combination the kernel is
    instantiated for, bit for bit against the oracle's loop; with members that are -0.0 (0 + -0.0 = +0.0 in the oracle's sum)."""
    rs = np.random.RandomState(17)
    n = 60000
    pts = cloud(orc, 99, n, 1.0)                      # 8 voxels of 1.0 m hold ~4000 points each ...
    pts[:20000, :3] *= np.float32(6.0)                # ... 20000 points spread over ~1700 voxels (runs of ~12)
    pts[30000:30400, :3] = np.float32([0.25, 0.5, 0.75])  # 400 identical points
    pts[40000:40003, 0] = np.float32(-0.0)
    pts[50000:50600, :3] = np.float32([-0.0, 0.5, 0.5])   # a run whose x sum is made of -0.0 terms only... inside a bigger voxel
    rgb = rs.uniform(0, 1, (n, 4)).astype(np.float32) if "rgb" in attrs else None
    ts = rs.uniform(0, 100, n).astype(np.float32) if "ts" in attrs else None
    inten = rs.uniform(0, 255, n).astype(np.float32) if "intensity" in attrs else None
    if inten is not None:
        inten[3000:] = np.round(inten[3000:])  # (the O(L^2) median of 4000-point runs: ties by position)
    pc = sp.PointCloudShared(dev(pts), rgb=None if rgb is None else dev(rgb), intensities=None if inten is None else dev(inten),
                             timestamp_offsets=None if ts is None else dev(ts))
    for minc in (1, 50):
        vg = sp.VoxelGrid(1.0)
        vg.set_min_voxel_count(minc)
        out, keys = vg.downsampling(pc, return_keys=True)
        o = orc.voxel_downsample(pts, 1.0, minc, rgb, inten, ts, stable=True)
        assert np.array_equal(keys.cpu().numpy().view(np.uint64), o["keys"])
        assert np.array_equal(out.points.cpu().numpy().view(np.uint32), o["points"].view(np.uint32))  # (bits: the sign of a zero too)
        if rgb is not None:
            assert np.array_equal(out.rgb.cpu().numpy(), o["rgb"])
        if ts is not None:
            assert np.array_equal(out.timestamp_offsets.cpu().numpy(), o["timestamps"])
        if inten is not None:
            assert np.array_equal(out.intensities.cpu().numpy(), o["intensities"])
    only_zero = np.zeros((700, 4), np.float32)
    only_zero[:, :3] = np.float32(-0.0)
    only_zero[:, 3] = 1.0
    out = sp.VoxelGrid(1.0).downsampling(dev(only_zero))
    o = orc.voxel_downsample(only_zero, 1.0, 1, stable=True)
    assert np.array_equal(out.points.cpu().numpy().view(np.uint32), o["points"].view(np.uint32))


def test_voxel_known_answer_and_edges(sp, orc):
    # test_downsampling_filters.cpp:27-88
    pts = np.array([[0.10, 0, 0, 1], [0.40, 0, 0, 1], [1.10, 0, 0, 1], [1.40, 0, 0, 1], [0.20, 0, 0, 1]], np.float32)
    rgb = np.array([[10, 20, 30, 1], [20, 40, 60, 1], [30, 60, 90, 1], [50, 70, 90, 1], [70, 80, 90, 1]], np.float32)
    inten = np.array([1, 3, 5, 7, 100], np.float32)
    ts = np.array([0, 2, 4, 6, 8], np.float32)
    vg = sp.VoxelGrid(1.0)
    vg.set_min_voxel_count(2)
    out = vg.downsampling(sp.PointCloudShared(dev(pts), rgb=dev(rgb), intensities=dev(inten), timestamp_offsets=dev(ts)))
    p = out.points.cpu().numpy()
    assert p.shape[0] == 2 and abs(p[0, 0] - 0.233333) < 1e-5 and abs(p[1, 0] - 1.25) < 1e-5
    assert abs(float(out.intensities[0]) - 3.0) < 1e-5 and abs(float(out.timestamp_offsets[0]) - 3.333333) < 1e-5
    assert np.allclose(out.rgb.cpu().numpy()[0, :3], [33.333333, 46.666667, 60.0], atol=1e-5)
    assert vg.downsampling(dev(np.zeros((0, 4), np.float32))).size() == 0
    allbad = np.full((10, 4), np.nan, np.float32)
    assert vg.downsampling(dev(allbad)).size() == 0
    with pytest.raises(sp.SpError):
        sp.VoxelGrid(0.0)


def test_voxel_boxed_path_equals_64bit_path(sp, orc):
    """sp_voxel_downsample_boxed sorts keys compressed to the cloud's bounding box; every output (and the 63-bit keys) must
    equal the 64-bit path, including clouds with invalid points, a box too large for 32 bits (falls back) and a stale
    box (status != 0)."""
    import ctypes as C

    rs = np.random.RandomState(3)
    n = 60000
    for pts, vs in ((cloud(orc, 5, n, 8.0), 0.3), (cloud(orc, 6, n, 2000.0), 0.002)):  # second: > 2^32 cells -> fallback
        pts = pts.copy()
        pts[7, 1] = np.inf
        pts[9, 0] = 3e9  # coordinate outside the 21-bit range: invalid key
        inten = rs.uniform(0, 255, n).astype(np.float32)
        pc = sp.PointCloudShared(dev(pts), intensities=dev(inten))
        vg = sp.VoxelGrid(vs)
        b, kb = vg.downsampling(pc, return_keys=True, boxed=False)
        for _ in range(2):  # first call: the box is computed first; second call: the remembered (widened) box
            a, ka = vg.downsampling(pc, return_keys=True, boxed=True)
            assert torch.equal(ka, kb) and torch.equal(a.points, b.points) and torch.equal(a.intensities, b.intensities)
        # a cloud that leaves the remembered box is redone with its own box
        far = sp.PointCloudShared(dev(pts + np.float32([500.0, 0, 0, 0])), intensities=dev(inten))
        c, kc = vg.downsampling(far, return_keys=True, boxed=True)
        d, kd = sp.VoxelGrid(vs).downsampling(far, return_keys=True, boxed=False)
        assert torch.equal(kc, kd) and torch.equal(c.points, d.points)
    # a box that does not cover the cloud is reported, not silently accepted
    L = sp._lib.lib()
    pts = cloud(orc, 7, 5000, 4.0)
    P = dev(pts)
    nb = L.sp_voxel_downsample_workspace_bytes(5000)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    o = torch.empty((5000, 4), dtype=torch.float32, device="cuda")
    cnt = torch.zeros(2, dtype=torch.int32, device="cuda")
    box = np.array([1 << 20, 1 << 20, 1 << 20, (1 << 20) + 3, (1 << 20) + 3, (1 << 20) + 3], np.int32)  # positive octant only
    shards = torch.zeros((sp._lib.VOXEL_BOX_SHARDS, sp._lib.VOXEL_BOX_SHARD_STRIDE), dtype=torch.int32, device="cuda")
    sp._lib.check(L.sp_voxel_downsample_boxed(sp._ptr(P), 5000, 1.0, 1, None, None, None, sp._ptr(o), None, None, None, None,
                                              sp._ptr(cnt), box.ctypes.data_as(C.c_void_p), C.c_void_p(cnt.data_ptr() + 4),
                                              sp._ptr(shards), sp._ptr(ws), nb, sp._stream()))
    assert int(cnt[1]) > 0
    boxd = torch.empty(6, dtype=torch.int32, device="cuda")
    sp._lib.check(L.sp_voxel_key_box(sp._ptr(P), 5000, 1.0, sp._ptr(boxd), sp._stream()))
    keys = sp.VoxelGrid(1.0).compute_voxel_bit(P).cpu().numpy().view(np.uint64)
    f = [(keys >> s) & ((1 << 21) - 1) for s in (0, 21, 42)]
    true_box = [int(x.min()) for x in f] + [int(x.max()) for x in f]
    assert boxd.cpu().numpy().tolist() == true_box
    # ... and the key kernel of the boxed call found the same box on the way (sharded), although the box it was given was wrong
    sh = shards.cpu().numpy().astype(np.int64)
    assert sh[:, :3].min(0).tolist() + sh[:, 3:6].max(0).tolist() == true_box


def test_voxel_report_record(sp, orc):
    """sp_voxel_downsample_report: the boxed call with ONE 8-word record {voxels, points outside the box, this cloud's key box}
    stored by the call's last kernel (no status word, no sharded box, no initialisation launch): outputs identical to
    sp_voxel_downsample_boxed, the record equal to that call's count / status / folded shards — for a covering box, a box the cloud
    leaves, no box at all (64-bit keys), an empty cloud, and above the offsets fold (a report launch of its own)."""
    import ctypes as C

    L = sp._lib.lib()

    def run(pts, vs, box, report_api):
        n = len(pts)
        P = dev(pts) if n else torch.empty((0, 4), dtype=torch.float32, device="cuda")
        nb = L.sp_voxel_downsample_workspace_bytes(max(n, 1))
        ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
        o = torch.zeros((max(n, 1), 4), dtype=torch.float32, device="cuda")
        keys = torch.zeros(max(n, 1), dtype=torch.int64, device="cuda")
        bx = None if box is None else np.asarray(box, np.int32).ctypes.data_as(C.c_void_p)
        if report_api:
            rep = torch.full((8,), 0x7fffffff, dtype=torch.int32, device="cuda")
            sp._lib.check(L.sp_voxel_downsample_report(sp._ptr(P), n, 1.0 / vs, 1, None, None, None, sp._ptr(o), None, None, None,
                                                       sp._ptr(keys), None, bx, sp._ptr(rep), sp._ptr(ws), nb, sp._stream()))
            r = rep.cpu().numpy().astype(np.int64)
            cnt, status, kb = int(r[0]), int(r[1]), r[2:8].tolist()
        else:
            cs = torch.zeros(2, dtype=torch.int32, device="cuda")
            shards = torch.zeros((sp._lib.VOXEL_BOX_SHARDS, sp._lib.VOXEL_BOX_SHARD_STRIDE), dtype=torch.int32, device="cuda")
            sp._lib.check(L.sp_voxel_downsample_boxed(sp._ptr(P), n, 1.0 / vs, 1, None, None, None, sp._ptr(o), None, None, None,
                                                      sp._ptr(keys), sp._ptr(cs), bx, C.c_void_p(cs.data_ptr() + 4), sp._ptr(shards),
                                                      sp._ptr(ws), nb, sp._stream()))
            sh = shards.cpu().numpy().astype(np.int64)
            cnt, status = int(cs[0]), int(cs[1])
            kb = sh[:, :3].min(0).tolist() + sh[:, 3:6].max(0).tolist()
        return cnt, status, kb, o[:cnt].cpu().numpy(), keys[:cnt].cpu().numpy()

    pts = cloud(orc, 41, 50000, 6.0)
    pts[11, 2] = np.nan
    keys = sp.VoxelGrid(0.25).compute_voxel_bit(dev(pts)).cpu().numpy().view(np.uint64)
    ok = keys != np.uint64(0xFFFFFFFFFFFFFFFF)
    f = [(keys[ok] >> np.uint64(s)) & np.uint64((1 << 21) - 1) for s in (0, 21, 42)]
    true_box = [int(x.min()) for x in f] + [int(x.max()) for x in f]
    wide = [b - 3 for b in true_box[:3]] + [b + 5 for b in true_box[3:]]
    tight = true_box[:3] + [true_box[3] - 4] + true_box[4:]  # the cloud leaves it in x
    for box in (wide, true_box, tight, None):
        a, b = run(pts, 0.25, box, True), run(pts, 0.25, box, False)
        assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2] == true_box, (box, a[:3], b[:3])
        assert (a[1] > 0) == (box is tight)
        if a[1] == 0:
            assert np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])
            o = orc.voxel_downsample(pts, 0.25, 1, stable=True)
            assert np.array_equal(a[4].view(np.uint64), o["keys"]) and np.array_equal(a[3], o["points"])
    e = run(np.zeros((0, 4), np.float32), 0.25, None, True)
    assert e[0] == 0 and e[1] == 0 and e[2][0] > e[2][3]  # empty cloud: an empty box
    big = cloud(orc, 42, 1_200_000, 6.0)  # 4688 aggregation workgroups: offsets launch + a report launch
    a, b = run(big, 0.2, None, True), run(big, 0.2, None, False)
    kb = a[2]
    c, d = run(big, 0.2, kb, True), run(big, 0.2, kb, False)
    for x, y in ((a, b), (c, d), (a, c)):
        assert x[0] == y[0] and x[1] == y[1] == 0 and x[2] == y[2] and np.array_equal(x[3], y[3]) and np.array_equal(x[4], y[4])


def test_voxel_config3_1m(sp, orc):
    # BASELINE config 3 at full size: 1M points, voxel 0.1; oracle keys everywhere, oracle means on the whole cloud
    pts = cloud(orc, 1234, 1000000, 10.0)
    out, keys = sp.VoxelGrid(0.1).downsampling(dev(pts), return_keys=True)
    k = keys.cpu().numpy().view(np.uint64)
    assert (np.diff(k.astype(np.int64)) > 0).all()  # sortedness / uniqueness
    o = orc.voxel_downsample(pts, 0.1, 1, stable=True)
    assert np.array_equal(k, o["keys"]) and np.array_equal(out.points.cpu().numpy(), o["points"])
    # idempotence-like property: every output point lies in its own voxel
    k2 = sp.VoxelGrid(0.1).compute_voxel_bit(out.points.contiguous()).cpu().numpy().view(np.uint64)
    assert np.mean(k2 == k) > 0.999  # means of one-point voxels are the points themselves


def test_voxel_above_the_offsets_fold(sp, orc):
    """More than 4096 aggregation workgroups (1 048 576 sorted positions): the scatter takes its offsets from the
    block_offsets launch again instead of summing the kept counts itself (voxel.hip kFoldBlocks) — both key widths against the
    oracle, and a size just below the limit on the folded path."""
    for n in (1_300_000, 1_048_576):
        pts = cloud(orc, 77, n, 6.0)
        o = orc.voxel_downsample(pts, 0.2, 1, stable=True)
        for boxed in (True, False):
            out, keys = sp.VoxelGrid(0.2).downsampling(dev(pts), return_keys=True, boxed=boxed)
            assert np.array_equal(keys.cpu().numpy().view(np.uint64), o["keys"]), (n, boxed)
            assert np.array_equal(out.points.cpu().numpy(), o["points"]), (n, boxed)


def test_voxel_config3_dense_1m_boxed(sp, orc):
    """BASELINE config 3's dense variant at full size (R = 2.5, voxel 0.1, 1 M points, about 8 points per voxel)
    through the path bench.py times for it: keys compressed to the cloud's key box (18 key bits: the sort takes two passes of
    9-bit digits, radix_sort.hip) — keys, means and counts against the oracle, and against the 64-bit path."""
    pts = cloud(orc, 1234, 1000000, 2.5)
    rs = np.random.RandomState(11)
    inten = rs.uniform(0, 255, len(pts)).astype(np.float32)
    pc = sp.PointCloudShared(dev(pts), intensities=dev(inten))
    vg = sp.VoxelGrid(0.1)
    o = orc.voxel_downsample(pts, 0.1, 1, intensity=inten, stable=True)
    assert 100_000 < len(o["keys"]) < 160_000  # ~8 points per voxel: the segmented reduce does real work
    for _ in range(2):  # the box is computed by the first call and remembered (widened) for the second
        out, keys = vg.downsampling(pc, return_keys=True, boxed=True)
        k = keys.cpu().numpy().view(np.uint64)
        assert np.array_equal(k, o["keys"])
        assert np.array_equal(out.points.cpu().numpy(), o["points"])
        assert np.array_equal(out.intensities.cpu().numpy(), o["intensities"])  # medians
    b, kb = sp.VoxelGrid(0.1).downsampling(pc, return_keys=True, boxed=False)
    assert torch.equal(kb, keys) and torch.equal(b.points, out.points)
    # min_voxel_count drops sparse voxels identically
    vg6 = sp.VoxelGrid(0.1)
    vg6.set_min_voxel_count(6)
    out3, k3 = vg6.downsampling(pc, return_keys=True, boxed=True)
    o3 = orc.voxel_downsample(pts, 0.1, 6, intensity=inten, stable=True)
    assert np.array_equal(k3.cpu().numpy().view(np.uint64), o3["keys"]) and np.array_equal(out3.points.cpu().numpy(), o3["points"])


# ------------------------------------------------------------------ K10 / K14
def test_box_filter_transform_compaction(sp, orc):
    pts = cloud(orc, 8, 50000, 60.0)
    pts[3, 0] = np.nan
    flags = sp.box_filter_flags(dev(pts), 0.5, 50.0)
    of = orc.box_filter(pts, 0.5, 50.0)
    assert np.array_equal(flags.cpu().numpy(), of)
    kept, new_idx = sp.compact_by_flags(dev(pts), flags, want_indices=True)
    assert np.array_equal(kept.cpu().numpy(), pts[of == 1])
    assert np.array_equal(new_idx.cpu().numpy(), np.where(of == 1, np.cumsum(of.astype(np.int64)) - 1, -1))
    # known answer (test_preprocess_filter.cpp:29-53)
    small = np.array([[0.5, 0, 0, 1], [2, 0, 0, 1], [0, 0, 4, 1], [np.nan, 1, 0, 1]], np.float32)
    assert sp.box_filter_flags(dev(small), 1.0, 3.0).cpu().tolist() == [0, 1, 0, 0]
    T = orc.se3_exp([0.3, -0.2, 0.1, 1.0, 2.0, -3.0])
    good = pts[of == 1][:10000]
    idx, _ = orc.knn_bruteforce(good[:2000], good[:2000], 8)
    covs = orc.cov_estimate(good[:2000], idx)
    nrm = orc.normals_from_cov(good[:2000], covs)
    pc = sp.PointCloudShared(dev(good[:2000]), covs=dev(covs), normals=dev(nrm))
    out = sp.transform_copy(pc, T)
    assert np.array_equal(out.points.cpu().numpy(), orc.transform_points(good[:2000], T))
    assert np.array_equal(out.covs.cpu().numpy(), orc.transform_covs(covs, T))
    assert np.array_equal(out.normals.cpu().numpy(), orc.transform_normals(nrm, T))


def test_grid_with_a_box_vouched_for_by_the_caller(sp, orc):
    # sp_grid_create_bounded: the caller's box instead of a bounding-box kernel and its read-back. Any box that holds the cloud
    # gives the same neighbours (indices and distances, bit for bit) as the grid that measured its own; a finite point outside the
    # box is reported — by a later call: the check runs on the device — instead of silently sitting in a cell that does not hold it.
    from sycl_points_amd import _lib

    g = orc.rng(31)
    pts = g.uniform_points(40000, 10.0)
    pts[:, 2] *= 0.2
    pts[17, 0] = np.nan
    q = g.uniform_points(5000, 11.0)
    own = sp.GridKNN.build(dev(pts), points_per_cell=2.0)
    for box in ((-10.0, -10.0, -2.0, 10.0, 10.0, 2.0), (-25.0, -11.0, -2.5, 12.0, 30.0, 9.0)):
        bounded = sp.GridKNN.build(dev(pts), points_per_cell=2.0, bounds=box)
        for k in (1, 7):
            a, b = own.knn_search(dev(q), k), bounded.knn_search(dev(q), k)
            assert torch.equal(a.indices, b.indices) and torch.equal(a.distances, b.distances)
    oi, od = orc.knn_bruteforce(q[:500], pts[np.isfinite(pts[:, 0])], 1)
    torch.cuda.synchronize()
    with pytest.raises(sp.SpError, match="outside the bounds"):  # (by the build itself when its kernel has finished by then, else by the next call)
        bad = sp.GridKNN.build(dev(pts), points_per_cell=2.0, bounds=(-5.0, -10.0, -2.0, 10.0, 10.0, 2.0))  # a fifth of the cloud lies outside
        torch.cuda.synchronize()
        _lib.check(_lib.lib().sp_grid_order(bad._h, torch.empty(40000, dtype=torch.int32, device="cuda").data_ptr(), sp._stream()))
    torch.cuda.synchronize()
    own.knn_search(dev(q), 1)  # (the error word is cleared by the report: later calls are clean)


def test_small_grids_are_built_by_one_workgroup_to_the_same_structure(sp, orc):
    """Grids of up to 8192 points and fewer than 32768 cells: cell ids, sort, gather and cell table in ONE launch of one workgroup
    (grid_build_small_kernel) — the same structure as the general chain of launches: the cell-order permutation, the neighbours
    (k = 1, 10, bit for bit, and against the brute-force oracle), the fullest cell. Uniform clouds of 1 .. 8192 points (and 8193: the
    general build either way), a surface, two far clusters (one long run of empty cells), points that are not finite, duplicates,
    one cell for everything, a given cell size, a caller's box — and a point outside that box is still reported."""
    from sycl_points_amd import _lib

    L = _lib.lib()
    g = orc.rng(2024)
    rs = np.random.RandomState(5)
    clouds = {}
    for n in (1, 2, 63, 64, 65, 1000, 4097, 8191, 8192, 8193):
        clouds[f"uniform{n}"] = (g.uniform_points(n, 6.0), {})
    surf = g.uniform_points(6000, 20.0)
    surf[:, 2] = np.float32(0.05) * np.sin(surf[:, 0]) + np.float32(1.0)
    clouds["surface"] = (surf, {})
    two = g.uniform_points(5000, 1.0)
    two[2500:, :3] += np.float32([300.0, -200.0, 40.0])
    clouds["two clusters"] = (two, {})
    bad = g.uniform_points(3000, 4.0)
    bad[::7, 1] = np.nan
    bad[5, 0] = np.inf
    clouds["not finite"] = (bad, {})
    dup = g.uniform_points(2000, 3.0)
    dup[1000:] = dup[:1000]
    clouds["duplicates"] = (dup, {})
    clouds["one cell"] = (g.uniform_points(700, 1.0), dict(cell_size=50.0))
    clouds["cell size"] = (g.uniform_points(8000, 5.0), dict(cell_size=0.4))
    clouds["box"] = (g.uniform_points(6100, 9.0), dict(bounds=(-9.5, -9.0, -9.25, 9.0, 10.0, 9.5)))
    assert L.sp_internal_grid_small_build(-1) == 1
    try:
        for name, (pts, kw) in clouds.items():
            q = np.concatenate([pts[rs.choice(len(pts), min(len(pts), 400))], g.uniform_points(100, 8.0)])
            q[np.isnan(q)] = 0.0
            q[np.isinf(q)] = 0.0
            built = {}
            for small in (1, 0):
                L.sp_internal_grid_small_build(small)
                gr = sp.GridKNN.build(dev(pts), **kw)
                res = [gr.knn_search(dev(q), k) for k in (1, 10)]
                built[small] = (gr.order().cpu().numpy(), [(r.indices.cpu().numpy(), r.distances.cpu().numpy()) for r in res],
                                gr.max_cell_points(), gr.cell_size())
            a, b = built[1], built[0]
            assert np.array_equal(a[0], b[0]), name
            assert a[2] == b[2] and a[3] == b[3], name
            for (ai, ad), (bi, bd) in zip(a[1], b[1]):
                assert np.array_equal(ai, bi) and np.array_equal(ad, bd), name
            fin = np.isfinite(pts[:, :3]).all(1)
            oi, od = orc.knn_bruteforce(q, pts[fin], 1)
            assert np.array_equal(a[1][0][1], od), name
            assert np.array_equal(np.flatnonzero(fin)[oi[:, 0]], a[1][0][0][:, 0]) or name == "duplicates", name
        L.sp_internal_grid_small_build(1)
        pts = g.uniform_points(5000, 4.0)
        torch.cuda.synchronize()
        with pytest.raises(sp.SpError, match="outside the bounds"):
            badg = sp.GridKNN.build(dev(pts), bounds=(-2.0, -4.0, -4.0, 4.0, 4.0, 4.0))
            torch.cuda.synchronize()
            _lib.check(L.sp_grid_order(badg._h, torch.empty(5000, dtype=torch.int32, device="cuda").data_ptr(), sp._stream()))
        torch.cuda.synchronize()
    finally:
        L.sp_internal_grid_small_build(1)


@pytest.mark.parametrize("n", [1, 777, 2048, 69088, 300001])
def test_box_filter_and_compaction_in_one_launch(sp, orc, n):
    # sp_box_filter_compact_multi: the box test, the scan of its flags (decoupled look-back over tiles of 2048) and the stable move
    # of every attribute's kept rows in one kernel. Against the oracle's flags and a numpy compaction: flags, the kept rows of three
    # arrays of different row sizes (16 B points, 64 B covariances, 4 B intensities), the new indices, the count; non-finite points,
    # points on both bounds, several tiles, a last tile that is not full.
    import ctypes as C

    from sycl_points_amd import _lib

    pts = cloud(orc, n + 3, n, 60.0)
    if n > 10:
        pts[3, 0] = np.nan
        pts[5, 2] = np.inf
        pts[7, :3] = (0.5, 0.0, 0.0)    # on the lower bound: kept
        pts[8, :3] = (0.0, -50.0, 1.0)  # on the upper bound: kept
    covs = np.arange(n * 16, dtype=np.float32).reshape(n, 16)
    inten = np.arange(n, dtype=np.float32)
    of = orc.box_filter(pts, 0.5, 50.0)
    keep = of == 1
    L = _lib.lib()
    d_pts, d_cov, d_int = dev(pts), dev(covs), dev(inten)
    o_pts, o_cov, o_int = torch.zeros_like(d_pts), torch.zeros_like(d_cov), torch.zeros_like(d_int)
    flags = torch.zeros(n, dtype=torch.uint8, device="cuda")
    new_idx = torch.zeros(n, dtype=torch.int32, device="cuda")
    count = torch.zeros(1, dtype=torch.int32, device="cuda")
    ws = torch.empty(max(L.sp_compact_workspace_bytes(n), 16), dtype=torch.uint8, device="cuda")
    rows = (C.c_void_p * 3)(d_pts.data_ptr(), d_cov.data_ptr(), d_int.data_ptr())
    outs = (C.c_void_p * 3)(o_pts.data_ptr(), o_cov.data_ptr(), o_int.data_ptr())
    sizes = (C.c_size_t * 3)(16, 64, 4)
    _lib.check(L.sp_box_filter_compact_multi(d_pts.data_ptr(), n, 0.5, 50.0, rows, sizes, outs, 3, flags.data_ptr(), new_idx.data_ptr(),
                                             count.data_ptr(), ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream))
    m = int(count.cpu()[0])
    assert m == int(keep.sum())
    assert np.array_equal(flags.cpu().numpy(), of)
    assert np.array_equal(o_pts.cpu().numpy()[:m].view(np.uint32), pts[keep].view(np.uint32))
    assert np.array_equal(o_cov.cpu().numpy()[:m], covs[keep]) and np.array_equal(o_int.cpu().numpy()[:m], inten[keep])
    assert np.array_equal(new_idx.cpu().numpy(), np.where(keep, np.cumsum(of.astype(np.int64)) - 1, -1))


# ------------------------------------------------------------------ K11/K12/K13 + align
def gicp_inputs(orc, n, seed=1234):
    from sycl_points_amd.synthetic import gicp_pair

    r = 10.0 * (n / 1e6) ** (1.0 / 3.0)  # the density of BASELINE config 4 at every n
    src, tgt, T_gt = gicp_pair(n, r, seed)
    ti, _ = orc.kdtree_knn(orc.kdtree_build(tgt), tgt, 20)
    si, _ = orc.kdtree_knn(orc.kdtree_build(src), src, 20)
    return src, orc.cov_estimate(src, si), tgt, orc.cov_estimate(tgt, ti), T_gt


@pytest.fixture(scope="module")
def gicp20k(orc):
    return gicp_inputs(orc, 20000)


@pytest.mark.parametrize("reg", ["GICP", "POINT_TO_DISTRIBUTION", "POINT_TO_POINT", "POINT_TO_PLANE", "GENZ"])
@pytest.mark.parametrize("loss", ["NONE", "HUBER", "GEMAN_MCCLURE"])
def test_linearize_error_weights_match_oracle(sp, orc, gicp20k, reg, loss):
    src, scov, tgt, tcov, T_gt = gicp20k
    T = orc.se3_exp([0.004, -0.01, 0.008, 0.02, -0.01, 0.005])
    nrm = orc.normals_from_cov(tgt, tcov)
    nodes = orc.kdtree_build(tgt)
    idx, d2 = orc.kdtree_knn(nodes, src, 1, T)
    d2 = d2.copy()
    d2[::97] = 100.0  # rejected correspondences (> max_corr^2)
    scale = 0.5
    alpha = 0.7
    ref = orc.gicp_linearize(src, scov, tgt, tcov, nrm, idx, d2, T, 2.0, reg, loss, scale, alpha)
    reg_ = sp.Registration(sp.RegistrationParams(reg_type=reg, robust_type=loss))
    reg_.genz_alpha = alpha
    S = sp.PointCloudShared(dev(src), covs=dev(scov))
    Tg = sp.PointCloudShared(dev(tgt), covs=dev(tcov), normals=dev(nrm))
    reg_.neighbors.indices, reg_.neighbors.distances = dev(idx), dev(d2)
    _, lin = reg_._buffers(S.points.device)
    reg_._linearize("linearize", S, Tg, T, scale, lin)
    got = reg_._read_lin(lin)
    H = np.array(got.H, np.float32).reshape(6, 6)
    b = np.array(got.b, np.float32)
    assert got.inlier == ref["inlier"]
    hs = np.abs(ref["H"]).max()
    assert np.allclose(H, ref["H"], atol=2e-5 * hs), np.abs(H - ref["H"]).max() / hs
    assert np.allclose(b, ref["b"], atol=2e-5 * max(np.abs(ref["b"]).max(), 1e-3 * hs))
    assert abs(got.error - ref["error"]) <= 2e-5 * abs(ref["error"])
    assert np.array_equal(H, H.T)
    # deterministic reduction: a second launch gives the same bits
    reg_._linearize("linearize", S, Tg, T, scale, lin)
    again = reg_._read_lin(lin)
    assert bytes(again)[:176] == bytes(got)[:176]
    e, inl = reg_.compute_error_frozen(S, Tg, T, scale)
    oe, oinl = orc.gicp_error(src, scov, tgt, tcov, nrm, idx, d2, T, 2.0, reg, loss, scale, alpha)
    assert inl == oinl and abs(e - oe) <= 2e-5 * abs(oe)
    if reg != "GENZ":
        class Frozen(sp.KNNBase):
            def knn_search_async(self, queries, k, result, transT=None):
                result.indices, result.distances = dev(idx), dev(d2)
        w = reg_.compute_icp_robust_weights(S, Tg, Frozen(), T, scale).cpu().numpy()
        ow = orc.icp_robust_weights(src, scov, tgt, tcov, nrm, idx, d2, T, 2.0, reg, loss, scale)
        # GICP / P2D weights go through the eigen-decomposition (acosf/cosf): a few 1e-5 relative on the norm
        assert np.allclose(w, ow, atol=2e-4), float(np.abs(w - ow).max())


def test_validate_params_errors(sp, orc, gicp20k):
    src, scov, tgt, tcov, _ = gicp20k
    reg = sp.Registration(sp.RegistrationParams(reg_type="GICP"))
    S = sp.PointCloudShared(dev(src))
    Tg = sp.PointCloudShared(dev(tgt), covs=dev(tcov))
    tree = sp.KDTree.build(Tg.points)
    with pytest.raises(sp.SpError):  # registration.hpp:144-150
        reg.align(S, Tg, tree)
    assert np.array_equal(reg.align(sp.PointCloudShared(dev(np.zeros((0, 4), np.float32))), Tg, tree).T, np.eye(4))


@pytest.mark.parametrize("method,loss", [("GN", "NONE"), ("LM", "GEMAN_MCCLURE"), ("GN", "HUBER")])
def test_align_matches_oracle_transform(sp, orc, gicp20k, method, loss):
    from oracle.pyoracle import LOSS, OPT, REG, RegParams

    src, scov, tgt, tcov, T_gt = gicp20k
    S = sp.PointCloudShared(dev(src), covs=dev(scov))
    Tg = sp.PointCloudShared(dev(tgt), covs=dev(tcov))
    tree = sp.KDTree.build(Tg.points)
    p = sp.RegistrationParams(reg_type="GICP", optimization_method=method, robust_type=loss, robust_default_scale=1.0,
                              criteria_translation=0.0, criteria_rotation=0.0, max_iterations=12)
    res = sp.Registration(p).align(S, Tg, tree)
    op = RegParams.defaults(reg_type=REG["GICP"], robust_type=LOSS[loss], optimization_method=OPT[method],
                            robust_default_scale=1.0, crit_translation=0.0, crit_rotation=0.0, max_iterations=12)
    ref = orc.registration_align(op, src, scov, tgt, tcov)
    assert np.abs(res.T - ref["T"]).max() < 1e-5          # BASELINE: final SE(3) within 1e-5 of the oracle
    assert np.abs(res.T - T_gt).max() < 5e-4              # and it is the right pose
    assert res.inlier == ref["inlier"]


def test_align_converges_with_criteria(sp, orc, gicp20k):
    src, scov, tgt, tcov, T_gt = gicp20k
    S = sp.PointCloudShared(dev(src), covs=dev(scov))
    Tg = sp.PointCloudShared(dev(tgt), covs=dev(tcov))
    tree = sp.KDTree.build(Tg.points)
    res = sp.Registration(sp.RegistrationParams()).align(S, Tg, tree)
    assert res.converged and res.iterations < 10 and np.abs(res.T - T_gt).max() < 5e-4


def test_device_loop_equals_host_loop_and_bruteforce_knn(sp, orc, gicp20k):
    src, scov, tgt, tcov, T_gt = gicp20k
    S = sp.PointCloudShared(dev(src), covs=dev(scov))
    Tg = sp.PointCloudShared(dev(tgt), covs=dev(tcov))
    tree = sp.KDTree.build(Tg.points)
    p = sp.RegistrationParams(criteria_translation=0.0, criteria_rotation=0.0, max_iterations=8)
    host = sp.Registration(p).align(S, Tg, tree)
    reg = sp.Registration(p)
    T_dev, lin, delta = reg.align_device_loop(S, Tg, tree, iterations=8)
    Td = reg.T_from_device(T_dev)
    assert np.abs(Td - host.T).max() < 2e-6
    # the KNNBase seam: a different exact NN structure gives the same pose
    bf = sp.Registration(p).align(S, Tg, sp.BruteForceKNN(Tg.points))
    assert np.abs(bf.T - host.T).max() < 2e-6


def test_gicp_config4_1m(sp, orc):
    # BASELINE config 4 at full size on the GPU (1M vs 1M, k=20 covariances computed by the HIP path, 20 GN
    # iterations); oracle comparison on the linear system of a 50k-point sample of the same correspondence set.
    from sycl_points_amd.synthetic import gicp_pair

    n = 1000000
    src, tgt, T_gt = gicp_pair(n, 10.0)
    S = sp.PointCloudShared(dev(src))
    Tg = sp.PointCloudShared(dev(tgt))
    ttree = sp.KDTree.build(Tg.points)
    stree = sp.KDTree.build(S.points)
    sp.covariance.estimate(ttree.knn_search(Tg, 20), Tg)
    sp.covariance.estimate(stree.knn_search(S, 20), S)
    p = sp.RegistrationParams(criteria_translation=0.0, criteria_rotation=0.0, max_iterations=20)
    reg = sp.Registration(p)
    T_dev, lin, delta = reg.align_device_loop(S, Tg, ttree, iterations=20)
    T = reg.T_from_device(T_dev)
    assert np.abs(T - T_gt).max() < 1e-4
    got = reg._read_lin(lin)
    assert got.inlier == n
    # size-independent property: the converged update is a fixed point (|delta| tiny)
    assert np.abs(delta.cpu().numpy()[:6]).max() < 1e-5
    # sampled oracle check of K11 at the converged pose with the GPU's own correspondences
    sel = np.arange(0, n, 20)
    idx = reg.neighbors.indices.cpu().numpy()[sel]
    d2 = reg.neighbors.distances.cpu().numpy()[sel]
    scov = S.covs.cpu().numpy()[sel]
    tcov = Tg.covs.cpu().numpy()
    ref = orc.gicp_linearize(src[sel], scov, tgt, tcov, None, idx, d2, T, 2.0, "GICP", "NONE", 10.0)
    reg2 = sp.Registration(p)
    S2 = sp.PointCloudShared(dev(src[sel]), covs=dev(scov))
    reg2.neighbors.indices, reg2.neighbors.distances = dev(idx), dev(d2)
    _, lin2 = reg2._buffers(S2.points.device)
    reg2._linearize("linearize", S2, Tg, T, 10.0, lin2)
    g2 = reg2._read_lin(lin2)
    H = np.array(g2.H, np.float32).reshape(6, 6)
    assert np.allclose(H, ref["H"], atol=2e-5 * np.abs(ref["H"]).max()) and g2.inlier == ref["inlier"]


def test_gicp_config5_8m_tiles(sp):
    """BASELINE config 5 at full size on one GPU (8M vs 8M, R = 20): the one-call loop over the whole source converges to
    the ground truth, and the linear system of the whole source equals the sum of its 8 tiles' systems (what the per-
    iteration all-reduce adds up on an 8-GPU node) — size-independent properties, no oracle at this size."""
    import ctypes as C

    from sycl_points_amd.sharding import shard_range
    from sycl_points_amd.synthetic import gicp_pair

    n, world = 8000000, 8
    src, tgt, T_gt = gicp_pair(n, 20.0)
    Tg = sp.PointCloudShared(dev(tgt))
    Tg.covs = sp.GridKNN.build(Tg.points, points_per_cell=6.0).self_knn(20, want_knn=False, want_covs=True)[1]
    S_all = dev(src)
    S_all = S_all[sp.GridKNN.build(S_all, points_per_cell=1.0).order()].contiguous()  # cell order, as bench.py stores it
    covs = sp.GridKNN.build(S_all, points_per_cell=6.0).self_knn(20, want_knn=False, want_covs=True)[1]
    S = sp.PointCloudShared(S_all, covs=covs)
    grid = sp.GridKNN.build(Tg.points, points_per_cell=0.5)
    prep = sp.PreparedTarget(grid, Tg.covs)
    p = sp.RegistrationParams(criteria_translation=0.0, criteria_rotation=0.0, max_iterations=20)
    reg = sp.Registration(p)
    T_dev, lin, delta = reg.align_fused_loop(S, prep, iterations=20, sort_by_cell="presorted")
    T = reg.T_from_device(T_dev)
    whole = reg._read_lin(lin)
    assert whole.inlier == n
    assert np.abs(T - T_gt).max() < 1e-4
    assert np.abs(delta.cpu().numpy()[:6]).max() < 1e-5  # fixed point of the update

    L = sp._lib.lib()

    def system_of(cloud):
        r = sp.Registration(p)
        ws, ln = r._buffers(cloud.points.device)
        ps = sp.PreparedSource(cloud.size())
        ps.prepare(prep, cloud, T, sort_by_cell="presorted")
        fp = r._factor_params(10.0)
        Tc = np.ascontiguousarray(T.T).reshape(-1)
        sp.check(L.sp_gicp_iteration_fused(prep._h, ps._h, Tc.ctypes.data_as(C.c_void_p), 0, C.byref(fp), None, None, None,
                                           sp._ptr(ln), None, sp._ptr(ws), ws.numel(), sp._stream()))
        g = r._read_lin(ln)
        return np.array(g.H, np.float64).reshape(6, 6), np.array(g.b, np.float64), float(g.error), int(g.inlier)

    Hw, bw, ew, iw = system_of(S)
    Hs, bs, es, is_ = np.zeros((6, 6)), np.zeros(6), 0.0, 0
    for r in range(world):
        lo, hi = shard_range(n, r, world)
        tile = sp.PointCloudShared(S.points[lo:hi].contiguous(), covs=S.covs[lo:hi].contiguous())
        H, b, e, i = system_of(tile)
        Hs += H; bs += b; es += e; is_ += i
    assert is_ == iw == n
    assert np.abs(Hs - Hw).max() <= 2e-5 * np.abs(Hw).max()
    assert abs(es - ew) <= 2e-5 * abs(ew)
    # b is a sum of 8M signed terms that cancel at the optimum: compare against the scale of its terms, H's rows
    assert np.abs(bs - bw).max() <= 2e-5 * np.abs(Hw).max() * 1e-2


@pytest.mark.parametrize("radius", [1.0, 0.05, 0.004])
def test_align_powell_dogleg_matches_oracle(sp, orc, gicp20k, radius):
    """optimize_powell_dogleg (registration.hpp:897-965): Gauss-Newton branch (large radius), dogleg / Cauchy branches and
    radius growth (small radii), against the oracle's restatement."""
    from oracle.pyoracle import OPT, RegParams

    src, scov, tgt, tcov, T_gt = gicp20k
    S = sp.PointCloudShared(dev(src), covs=dev(scov))
    Tg = sp.PointCloudShared(dev(tgt), covs=dev(tcov))
    tree = sp.KDTree.build(tgt)
    p = sp.RegistrationParams(optimization_method="DOGLEG", max_iterations=15, criteria_translation=1e-5,
                              criteria_rotation=1e-5, dogleg_initial_trust_region_radius=radius)
    got = sp.Registration(p).align(S, Tg, tree)
    ref = orc.registration_align(RegParams.defaults(optimization_method=OPT["DOGLEG"], max_iterations=15,
                                                    crit_translation=1e-5, crit_rotation=1e-5, dl_initial_radius=radius),
                                 src, scov, tgt, tcov)
    assert np.abs(got.T - ref["T"]).max() < 1e-5
    assert got.iterations == ref["iterations"] and got.converged == ref["converged"] and got.inlier == ref["inlier"]
    if radius >= 0.05:
        assert np.abs(got.T - T_gt).max() < 5e-4


# ------------------------------------------------------------------ default-off registration terms
@pytest.mark.parametrize("reg,loss", [("GICP", "NONE"), ("GICP", "HUBER"), ("POINT_TO_POINT", "CAUCHY"),
                                      ("POINT_TO_DISTRIBUTION", "TUKEY")])
def test_rotation_constraint_term_matches_oracle(sp, orc, gicp20k, reg, loss):
    """K11 / K12 with the Jensen-Bregman LogDet rotation constraint (rotation_constraint.hpp:15-128,
    registration.hpp:630-650, 758-766): per inlier correspondence, weighted, with its own robust scale."""
    src, scov, tgt, tcov, _ = gicp20k
    T = orc.se3_exp([0.08, -0.05, 0.06, 0.02, -0.01, 0.005])  # a visible rotation: the divergence is well above 0
    idx, d2 = orc.kdtree_knn(orc.kdtree_build(tgt), src, 1, T)
    d2 = d2.copy()
    d2[::53] = 100.0
    weight, rot_scale, scale = 60.0, 0.05, 0.5
    ref, (oe, oinl) = orc.gicp_linearize_rot(src, scov, tgt, tcov, None, idx, d2, T, 2.0, reg, loss, scale, 1.0, weight,
                                             rot_scale)
    plain = orc.gicp_linearize(src, scov, tgt, tcov, None, idx, d2, T, 2.0, reg, loss, scale, 1.0)
    assert np.abs(ref["H"][:3, :3] - plain["H"][:3, :3]).max() > 1e-3 * np.abs(plain["H"]).max()  # the term matters
    assert np.array_equal(ref["H"][3:, 3:], plain["H"][3:, 3:])
    reg_ = sp.Registration(sp.RegistrationParams(reg_type=reg, robust_type=loss, rotation_constraint_enable=True,
                                                 rotation_constraint_weight=weight,
                                                 rotation_constraint_robust_default_scale=rot_scale))
    S = sp.PointCloudShared(dev(src), covs=dev(scov))
    Tg = sp.PointCloudShared(dev(tgt), covs=dev(tcov))
    reg_.neighbors.indices, reg_.neighbors.distances = dev(idx), dev(d2)
    _, lin = reg_._buffers(S.points.device)
    reg_._linearize("linearize", S, Tg, T, scale, lin)
    got = reg_._read_lin(lin)
    H, b = np.array(got.H, np.float32).reshape(6, 6), np.array(got.b, np.float32)
    hs = np.abs(ref["H"]).max()
    assert got.inlier == ref["inlier"]
    assert np.allclose(H, ref["H"], atol=3e-5 * hs), np.abs(H - ref["H"]).max() / hs
    assert np.allclose(b, ref["b"], atol=3e-5 * max(np.abs(ref["b"]).max(), 1e-3 * hs))
    assert abs(got.error - ref["error"]) <= 3e-5 * abs(ref["error"])
    e, inl = reg_.compute_error_frozen(S, Tg, T, scale)
    assert inl == oinl and abs(e - oe) <= 3e-5 * abs(oe)
    # an explicit per-call scale overrides the default (ExecutionOptions::rotation_robust_scale)
    e2, _ = reg_.compute_error_frozen(S, Tg, T, scale, rotation_robust_scale=10.0)
    _, (oe2, _) = orc.gicp_linearize_rot(src, scov, tgt, tcov, None, idx, d2, T, 2.0, reg, loss, scale, 1.0, weight, 10.0)
    assert abs(e2 - oe2) <= 3e-5 * abs(oe2)
    # source covariances are required
    with pytest.raises(sp.SpError):
        reg_.align(sp.PointCloudShared(dev(src)), Tg, sp.KDTree.build(tgt))


@pytest.mark.parametrize("method", ["GN", "LM", "DOGLEG"])
def test_align_with_default_off_terms_matches_oracle(sp, orc, gicp20k, method):
    """Registration::align with the rotation constraint, NL-Reg degenerate regularisation and a MAP prior all active
    (registration.hpp:236-253, 854, 933) against the oracle's restatement of the same loop."""
    from oracle.pyoracle import OPT, RegParams

    src, scov, tgt, tcov, T_gt = gicp20k
    S = sp.PointCloudShared(dev(src), covs=dev(scov))
    Tg = sp.PointCloudShared(dev(tgt), covs=dev(tcov))
    tree = sp.KDTree.build(tgt)
    # previous frame: a plain alignment, whose raw system feeds the prior of this frame
    prev = sp.Registration(sp.RegistrationParams(max_iterations=3)).align(S, Tg, tree)
    oprev = orc.registration_align(RegParams.defaults(max_iterations=3), src, scov, tgt, tcov)
    assert np.abs(prev.H_raw - oprev["H_raw"]).max() <= 2e-5 * np.abs(oprev["H_raw"]).max()
    assert abs(prev.error_raw - oprev["error_raw"]) <= 2e-5 * abs(oprev["error_raw"])
    T_pred = orc.isometry_mul(prev.T, orc.se3_exp([0.002, -0.001, 0.001, 0.01, 0.0, -0.005]))
    has, Om, Tinv = orc.map_prior_update(oprev["H_raw"], oprev["error_raw"], oprev["inlier"], oprev["T"], T_pred)
    thr_rot = float(np.linalg.eigvalsh(oprev["H_raw"][:3, :3].astype(np.float64))[1] / oprev["inlier"]) * 1.001
    thr_tr = float(np.linalg.eigvalsh(oprev["H_raw"][3:, 3:].astype(np.float64))[0] / oprev["inlier"]) * 1.5
    p = sp.RegistrationParams(optimization_method=method, max_iterations=12, criteria_translation=1e-5,
                              criteria_rotation=1e-5, rotation_constraint_enable=True, rotation_constraint_weight=3.0,
                              degenerate_reg_type="NL_REG", degenerate_reg_rot_eigenvalue_threshold=thr_rot,
                              degenerate_reg_trans_eigenvalue_threshold=thr_tr, degenerate_reg_base_factor=0.5,
                              map_prior_enabled=True)
    reg = sp.Registration(p)
    assert reg.set_map_prior_state(prev, T_pred) and has
    got = reg.align(S, Tg, tree, initial_guess=T_pred)
    ref = orc.registration_align(
        RegParams.defaults(optimization_method=OPT[method], max_iterations=12, crit_translation=1e-5, crit_rotation=1e-5,
                           rot_enable=1, rot_weight=3.0, dr_type=1, dr_rot_threshold=thr_rot, dr_trans_threshold=thr_tr,
                           dr_base_factor=0.5, map_prior=(Om, Tinv)),
        src, scov, tgt, tcov, init_T=T_pred)
    assert np.abs(got.T - ref["T"]).max() < 1e-5
    assert got.iterations == ref["iterations"] and got.converged == ref["converged"] and got.inlier == ref["inlier"]
    hs = np.abs(ref["H"]).max()
    assert np.abs(got.H - ref["H"]).max() <= 5e-5 * hs            # regularised + prior system of the last iteration
    assert np.abs(got.H_raw - ref["H_raw"]).max() <= 5e-5 * hs    # and the raw one
    # the terms are really in: the penalty alone adds base_factor * inlier per penalised direction to the trace
    assert np.trace(got.H - got.H_raw) > 0.5 * 0.5 * got.inlier
    # each term alone is visible in the first iteration's system: the rotation constraint in the raw error (K11), the
    # two host terms in H (after the raw system was recorded)
    T1 = orc.isometry_mul(T_pred, orc.se3_exp([0.01, 0, 0, 0.02, 0, 0]))
    two = sp.Registration(sp.RegistrationParams(max_iterations=1)).align(S, Tg, tree, initial_guess=T1)
    for kw in (dict(rotation_constraint_enable=True, rotation_constraint_weight=3.0),
               dict(degenerate_reg_type="NL_REG", degenerate_reg_rot_eigenvalue_threshold=thr_rot,
                    degenerate_reg_trans_eigenvalue_threshold=thr_tr),
               dict(map_prior_enabled=True)):
        r1 = sp.Registration(sp.RegistrationParams(max_iterations=1, **kw))
        r1.set_map_prior_state(prev, T_pred)
        one = r1.align(S, Tg, tree, initial_guess=T1)
        if "rotation_constraint_enable" in kw:
            assert one.error_raw > two.error_raw * (1 + 1e-3), kw
        else:
            assert np.array_equal(one.H_raw, two.H_raw) and np.trace(one.H - two.H) > 1.0, kw


# ------------------------------------------------------------------ K8: M-estimated covariance, normalize_covariance
@pytest.mark.parametrize("loss", ["HUBER", "TUKEY", "CAUCHY", "GEMAN_MCCLURE", "NONE"])
@pytest.mark.parametrize("k", [10, 20, 33])
def test_robust_covariance_bitexact(sp, orc, loss, k):
    pts = cloud(orc, 21, 6000, 3.0)
    pts[:3000, 2] *= 0.02  # a thin slab: strongly anisotropic neighbourhoods next to isotropic ones
    oi, _ = orc.kdtree_knn(orc.kdtree_build(pts), pts, k)  # (the brute-force oracle stops at k = 20 like the reference)
    oi = np.ascontiguousarray(oi)
    oi[5, 7:] = -1         # fewer neighbours
    oi[6, 3:] = -1         # < 4 neighbours: identity
    idx = dev(oi)
    for iters, mad, mn in ((1, 1.0, 1.0), (3, 2.5, 0.05)):
        got = sp.covariance.estimate_robust(idx, dev(pts), loss, mad, mn, iters).cpu().numpy()
        ref = orc.cov_estimate_robust(pts, oi, loss, mad, mn, iters)
        assert np.array_equal(got, ref), (loss, k, iters, np.abs(got - ref).max())
    plain = sp.covariance.estimate(idx, dev(pts)).cpu().numpy()
    assert np.array_equal(sp.covariance.estimate_robust(idx, dev(pts), loss, 1.0, 1.0, 0).cpu().numpy(), plain)


def test_robust_covariance_k_limit_and_normalize(sp, orc):
    pts = cloud(orc, 22, 2000, 2.0)
    oi, _ = orc.knn_bruteforce(pts, pts, 20)
    with pytest.raises(sp.SpError):
        sp.covariance.estimate_robust(dev(np.zeros((10, 65), np.int32)), dev(pts[:10]))
    covs = orc.cov_estimate(pts, oi)
    covs[3] = 0.0  # largest eigenvalue below FLT_MIN: identity block
    got = sp.covariance.normalize_covariance(dev(covs)).cpu().numpy()
    ref = orc.cov_normalize(covs)
    assert np.abs(got - ref).max() < 2e-5  # through acosf / cosf (eigen-decomposition), like the normals
    assert np.array_equal(got[3].reshape(4, 4)[:3, :3], np.eye(3, dtype=np.float32))


# ------------------------------------------------------------------ GridKNN (MI355X-native KNNBase)
@pytest.mark.parametrize("k", [1, 2, 5, 10, 20])
@pytest.mark.parametrize("ppc", [0.3, 2.0, 8.0])
def test_grid_knn_equals_bruteforce_bitwise(sp, orc, k, ppc):
    g = orc.rng(77)
    tgt = g.uniform_points(20000, 5.0)
    qry = g.uniform_points(2500, 6.0)  # some queries outside the target bounding box
    grid = sp.GridKNN.build(dev(tgt), points_per_cell=ppc)
    T = orc.se3_exp([0.02, -0.01, 0.03, 0.2, -0.1, 0.05])
    r = grid.knn_search(dev(qry), k, T)
    oi, od = orc.knn_bruteforce(orc.transform_points(qry, T), tgt, k)
    assert np.array_equal(r.distances.cpu().numpy(), od)
    assert np.array_equal(r.indices.cpu().numpy(), oi)


def test_grid_knn_edge_cases(sp, orc):
    # duplicates (ties -> lowest index), a flat cloud (zero extent on one axis), fewer points than k, non-finite points
    base = cloud(orc, 3, 3000, 4.0)
    tgt = np.concatenate([base, base])
    grid = sp.GridKNN.build(dev(tgt))
    for k in (1, 3):
        r = grid.knn_search(dev(base[:500]), k)
        oi, od = orc.knn_bruteforce(base[:500], tgt, k)
        assert np.array_equal(r.indices.cpu().numpy(), oi) and np.array_equal(r.distances.cpu().numpy(), od)
    flat = cloud(orc, 4, 5000, 3.0)
    flat[:, 2] = 1.5
    r = sp.GridKNN.build(dev(flat)).knn_search(dev(flat[:300] + np.float32([0.01, 0, 0.4, 0])), 4)
    oi, od = orc.knn_bruteforce(flat[:300] + np.float32([0.01, 0, 0.4, 0]), flat, 4)
    assert np.array_equal(r.indices.cpu().numpy(), oi) and np.array_equal(r.distances.cpu().numpy(), od)
    few = cloud(orc, 5, 3)
    r = sp.GridKNN.build(dev(few)).knn_search(dev(cloud(orc, 6, 7)), 5)
    oi, od = orc.knn_bruteforce(cloud(orc, 6, 7), few, 5)
    assert np.array_equal(r.indices.cpu().numpy(), oi) and np.array_equal(r.distances.cpu().numpy(), od)
    bad = cloud(orc, 7, 1000)
    bad[10, 0] = np.nan
    bad[20, 1] = np.inf
    r = sp.GridKNN.build(dev(bad)).knn_search(dev(cloud(orc, 8, 200)), 2)
    oi, od = orc.knn_bruteforce(cloud(orc, 8, 200), bad, 2)
    assert np.array_equal(r.indices.cpu().numpy(), oi) and np.array_equal(r.distances.cpu().numpy(), od)
    r = sp.GridKNN.build(dev(np.zeros((0, 4), np.float32))).knn_search(dev(cloud(orc, 8, 5)), 2)
    assert (r.indices.cpu().numpy() == -1).all() and (r.distances.cpu().numpy() == FLT_MAX).all()
    one = sp.GridKNN.build(dev(np.array([[0, 0, 0, 1]], np.float32)))
    r = one.knn_search(dev(np.array([[1, 1, 1, 1]], np.float32)), 1)
    assert int(r.indices[0, 0]) == 0 and abs(float(r.distances[0, 0]) - 3.0) < 1e-6


@pytest.mark.parametrize("ppc", [0.02, 0.5, 4.0])
def test_grid_nn1_staged_walk_far_queries(sp, orc, ppc):
    # k = 1 goes 2x2x2 block -> batched 4x4x4 block -> seeded ring walk; queries far from every target point (between two
    # clusters, outside the bounding box, very sparse grids) exercise the later stages. Bit-exact vs brute force.
    g = orc.rng(91)
    a = g.uniform_points(6000, 2.0)
    b = g.uniform_points(6000, 2.0) + np.float32([9.0, 7.0, 5.0, 0.0])
    tgt = np.concatenate([a, b])
    qry = np.concatenate([g.uniform_points(3000, 14.0), g.uniform_points(1000, 2.0), a[:500] + np.float32([1e-3, 0, 0, 0])])
    grid = sp.GridKNN.build(dev(tgt), points_per_cell=ppc)
    r = grid.knn_search(dev(qry), 1)
    oi, od = orc.knn_bruteforce(qry, tgt, 1)
    assert np.array_equal(r.distances.cpu().numpy(), od)
    assert np.array_equal(r.indices.cpu().numpy(), oi)


def test_grid_knn_1m_matches_kdtree(sp, orc):
    from sycl_points_amd.synthetic import gicp_pair

    src, tgt, T_gt = gicp_pair(1000000, 10.0)
    Tg = sp.PointCloudShared(dev(tgt))
    S = sp.PointCloudShared(dev(src))
    grid = sp.GridKNN.build(Tg.points)
    tree = sp.KDTree.build(tgt)
    a = grid.knn_search(S, 1)
    b = tree.knn_search(S, 1)
    assert torch.equal(a.distances, b.distances) and torch.equal(a.indices, b.indices)  # tie-free data
    g20 = sp.GridKNN.build(Tg.points, points_per_cell=8.0)
    a = g20.knn_search(Tg.points[:200000].contiguous(), 20)
    b = tree.knn_search(Tg.points[:200000].contiguous(), 20)
    assert torch.equal(a.distances, b.distances) and torch.equal(a.indices, b.indices)


# ------------------------------------------------------------------ prepared / fused GICP iteration
@pytest.mark.parametrize("loss", ["NONE", "HUBER", "CAUCHY"])
@pytest.mark.parametrize("sort,fast", [(True, False), (False, False), (True, True)])
def test_fused_iteration_matches_oracle_linear_system(sp, orc, gicp20k, loss, sort, fast):
    import ctypes as C

    src, scov, tgt, tcov, T_gt = gicp20k
    T = orc.se3_exp([0.004, -0.01, 0.008, 0.02, -0.01, 0.005])
    idx, d2 = orc.knn_bruteforce(orc.transform_points(src, T), tgt, 1)
    ref = orc.gicp_linearize(src, scov, tgt, tcov, None, idx, d2, T, 0.05, "GICP", loss, 0.5)  # max_corr rejects some
    assert 0 < ref["inlier"] < len(src)
    S = sp.PointCloudShared(dev(src), covs=dev(scov))
    grid = sp.GridKNN.build(dev(tgt))
    prep = sp.PreparedTarget(grid, dev(tcov))
    reg = sp.Registration(sp.RegistrationParams(max_correspondence_distance=0.05, robust_type=loss,
                                                criteria_translation=0.0, criteria_rotation=0.0))
    L = sp._lib.lib()
    ws, lin = reg._buffers(S.points.device)
    psrc = sp.PreparedSource(len(src))
    psrc._set_option("fast_nn", 1 if fast else 0)  # csrc/sp_internal.h: per-handle, nothing to restore
    psrc.prepare(prep, S, T, sort_by_cell=sort)
    fp = reg._factor_params(0.5)
    reg.neighbors.resize(len(src), 1, S.points.device)
    Tc = np.ascontiguousarray(T.T).reshape(-1)
    sp.check(L.sp_gicp_iteration_fused(prep._h, psrc._h, Tc.ctypes.data_as(C.c_void_p), 0, C.byref(fp), None,
                                       sp._ptr(reg.neighbors.indices), sp._ptr(reg.neighbors.distances), sp._ptr(lin),
                                       None, sp._ptr(ws), ws.numel(), sp._stream()))
    got = reg._read_lin(lin)
    # correspondences come back in ORIGINAL source order whatever the internal order
    assert np.array_equal(reg.neighbors.indices.cpu().numpy(), idx) and np.array_equal(reg.neighbors.distances.cpu().numpy(), d2)
    H = np.array(got.H, np.float32).reshape(6, 6)
    hs = np.abs(ref["H"]).max()
    assert got.inlier == ref["inlier"]
    assert np.allclose(H, ref["H"], atol=2e-5 * hs), np.abs(H - ref["H"]).max() / hs
    assert np.allclose(np.array(got.b), ref["b"], atol=2e-5 * max(np.abs(ref["b"]).max(), 1e-3 * hs))
    assert abs(got.error - ref["error"]) <= 2e-5 * abs(ref["error"])
    assert np.array_equal(H, H.T)


def test_fused_loop_equals_generic_loop_and_oracle(sp, orc, gicp20k):
    from oracle.pyoracle import RegParams

    src, scov, tgt, tcov, T_gt = gicp20k
    S = sp.PointCloudShared(dev(src), covs=dev(scov))
    Tg = sp.PointCloudShared(dev(tgt), covs=dev(tcov))
    grid = sp.GridKNN.build(Tg.points)
    prep = sp.PreparedTarget(grid, Tg.covs)
    p = sp.RegistrationParams(criteria_translation=0.0, criteria_rotation=0.0, max_iterations=10)
    reg = sp.Registration(p)
    T_dev, lin, delta = reg.align_fused_loop(S, prep, iterations=10)
    Tf = reg.T_from_device(T_dev)
    generic = sp.Registration(p).align(S, Tg, grid)
    assert np.abs(Tf - generic.T).max() < 2e-6
    ref = orc.registration_align(RegParams.defaults(crit_translation=0.0, crit_rotation=0.0, max_iterations=10), src, scov,
                                 tgt, tcov)
    assert np.abs(Tf - ref["T"]).max() < 1e-5
    assert reg._read_lin(lin).inlier == ref["inlier"]
    assert float(delta[7]) == 1.0


@pytest.mark.parametrize("loss", ["NONE", "HUBER"])
def test_align_fused_one_call_loop(sp, orc, gicp20k, loss):
    """sp_gicp_align_fused (reduction + solve as the prologue of the next launch, convergence on the device) against
    the two-launch fixed-length loop and the oracle, incl. the reference's early exit (registration.hpp:266-268)."""
    from oracle.pyoracle import LOSS, RegParams

    src, scov, tgt, tcov, T_gt = gicp20k
    S = sp.PointCloudShared(dev(src), covs=dev(scov))
    Tg = sp.PointCloudShared(dev(tgt), covs=dev(tcov))
    grid = sp.GridKNN.build(Tg.points)
    prep = sp.PreparedTarget(grid, Tg.covs)
    # fixed length (criteria 0): equals the per-iteration launches to rounding (different partial-row shape)
    p = sp.RegistrationParams(criteria_translation=0.0, criteria_rotation=0.0, max_iterations=8, robust_type=loss,
                              robust_default_scale=0.5)
    reg = sp.Registration(p)
    T_dev, lin, delta = reg.align_fused_loop(S, prep, iterations=8, write_neighbors=True)
    T_one = reg.T_from_device(T_dev)
    assert int(reg._iters_dev[0]) == 8
    nn_one = reg.neighbors.indices.clone()
    reg2 = sp.Registration(p)
    T_dev2, lin2, _ = reg2.align_fused_loop(S, prep, iterations=8, per_iteration_launches=True, write_neighbors=True)
    assert np.abs(T_one - reg2.T_from_device(T_dev2)).max() < 2e-6
    assert torch.equal(nn_one, reg2.neighbors.indices)
    a, b = reg._read_lin(lin), reg2._read_lin(lin2)
    assert a.inlier == b.inlier
    Ha, Hb = np.array(a.H, np.float32), np.array(b.H, np.float32)
    assert np.allclose(Ha, Hb, atol=2e-5 * np.abs(Hb).max())
    ref = orc.registration_align(RegParams.defaults(crit_translation=0.0, crit_rotation=0.0, max_iterations=8,
                                                    robust_type=LOSS[loss], robust_default_scale=0.5), src, scov, tgt, tcov)
    assert np.abs(T_one - ref["T"]).max() < 1e-5
    # with criteria: the device loop stops where the reference's does, later launches are no-ops
    pc = sp.RegistrationParams(criteria_translation=1e-4, criteria_rotation=1e-4, max_iterations=30, robust_type=loss,
                               robust_default_scale=0.5)
    regc = sp.Registration(pc)
    T_devc, linc, deltac = regc.align_fused_loop(S, prep)
    refc = orc.registration_align(RegParams.defaults(crit_translation=1e-4, crit_rotation=1e-4, max_iterations=30,
                                                     robust_type=LOSS[loss], robust_default_scale=0.5), src, scov, tgt, tcov)
    assert refc["converged"] and float(deltac[6]) == 1.0
    assert int(regc._iters_dev[0]) == refc["iterations"] + 1  # result.iterations is the index of the last iteration
    assert np.abs(regc.T_from_device(T_devc) - refc["T"]).max() < 1e-5
    assert regc._read_lin(linc).inlier == refc["inlier"]
    # max_iterations = 0: nothing changes
    T0 = dev(np.eye(4, dtype=np.float32).reshape(-1))
    regc.align_fused_loop(S, prep, iterations=0, T_dev=T0)
    assert torch.equal(T0.cpu(), torch.eye(4).reshape(-1))


@pytest.mark.parametrize("case", [
    dict(ns=1, nt=5000), dict(ns=63, nt=5000), dict(ns=1025, nt=3000), dict(ns=5000, nt=1),
    dict(ns=4097, nt=7000, loss="CAUCHY", scale=0.3), dict(ns=3000, nt=9000, max_corr=0.12),
    dict(ns=6000, nt=6000, init=[0.05, -0.03, 0.04, 0.4, -0.3, 0.2]), dict(ns=2500, nt=2500, max_corr=0.0005, iters=3),
])
def test_align_fused_stress_shapes_and_parameters(sp, orc, case):
    """The one-call loop on awkward shapes (one point, not a multiple of the wave / workgroup size, one target point,
    sources much smaller than the workgroup count), with a robust kernel, a tight correspondence gate (most points
    rejected, even all), a far initial guess: same pose as the oracle, same inlier count, and bit-identical to the
    always-search path."""
    from oracle.pyoracle import LOSS, RegParams

    ns, nt = case["ns"], case["nt"]
    n = max(ns, nt)
    src, scov, tgt, tcov, T_gt = gicp_inputs(orc, n, seed=77)
    src, scov, tgt, tcov = src[:ns], scov[:ns], tgt[:nt], tcov[:nt]
    loss, scale = case.get("loss", "NONE"), case.get("scale", 10.0)
    max_corr, iters = case.get("max_corr", 2.0), case.get("iters", 6)
    T0 = orc.se3_exp(case["init"]) if "init" in case else np.eye(4, dtype=np.float32)
    ref = orc.registration_align(RegParams.defaults(crit_translation=0.0, crit_rotation=0.0, max_iterations=iters,
                                                    robust_type=LOSS[loss], robust_default_scale=scale,
                                                    max_correspondence_distance=max_corr), src, scov, tgt, tcov, init_T=T0)
    outs = []
    for reuse in (2, 0):
        S = sp.PointCloudShared(dev(src), covs=dev(scov))
        Tg = sp.PointCloudShared(dev(tgt), covs=dev(tcov))
        prep = sp.PreparedTarget(sp.GridKNN.build(Tg.points), Tg.covs)
        p = sp.RegistrationParams(criteria_translation=0.0, criteria_rotation=0.0, max_iterations=iters, robust_type=loss,
                                  robust_default_scale=scale, max_correspondence_distance=max_corr)
        reg = sp.Registration(p)
        reg._set_source_option("reuse", reuse)
        T_dev, lin, _ = reg.align_fused_loop(S, prep, initial_guess=T0, write_neighbors=True)
        outs.append((reg.T_from_device(T_dev), reg._read_lin(lin).inlier, lin.cpu().numpy(),
                     reg.neighbors.indices.cpu().numpy().copy()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][2], outs[1][2])
    assert np.array_equal(outs[0][3], outs[1][3])
    assert outs[0][1] == ref["inlier"]
    tol = 1e-5 if ref["inlier"] >= 6 else 1e-3  # (fewer inliers than unknowns: the damped system is all that is solved)
    assert np.abs(outs[0][0] - ref["T"]).max() < tol


def test_correspondence_reuse_is_exact(sp, orc, gicp20k):
    """The iteration kernel keeps a correspondence without searching when the query is provably still nearest to its
    previous winner (|q - t| < half of t's distance to its nearest other target point). Every output must be bit-identical
    to the always-search path — also with duplicated target points (radius 0: never reused) and a target with one point."""
    src, scov, tgt, tcov, T_gt = gicp20k
    tgt2, tcov2 = np.concatenate([tgt, tgt[:500]]), np.concatenate([tcov, tcov[:500]])  # 500 exact duplicates
    outs = []
    for reuse in (2, 1, 0):
        S = sp.PointCloudShared(dev(src), covs=dev(scov))
        Tg = sp.PointCloudShared(dev(tgt2), covs=dev(tcov2))
        grid = sp.GridKNN.build(Tg.points)
        prep = sp.PreparedTarget(grid, Tg.covs)
        p = sp.RegistrationParams(criteria_translation=0.0, criteria_rotation=0.0, max_iterations=10)
        reg = sp.Registration(p)
        reg._set_source_option("reuse", reuse)
        T_dev, lin, delta = reg.align_fused_loop(S, prep, write_neighbors=True)
        outs.append((T_dev.cpu().numpy(), lin.cpu().numpy(), reg.neighbors.indices.cpu().numpy().copy(),
                     reg.neighbors.distances.cpu().numpy().copy()))
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert np.array_equal(a, b)
    oi, od = orc.knn_bruteforce(orc.transform_points(src, orc.registration_align(
        __import__("oracle.pyoracle", fromlist=["RegParams"]).RegParams.defaults(crit_translation=0.0, crit_rotation=0.0,
                                                                                max_iterations=9), src, scov, tgt, tcov)["T"]),
        tgt2, 1)
    assert (outs[0][2].ravel() == oi.ravel()).mean() > 0.999  # lowest index among the duplicates, as brute force
    one = sp.PointCloudShared(dev(tgt[:1]), covs=dev(tcov[:1]))
    g1 = sp.GridKNN.build(one.points)
    reg = sp.Registration(sp.RegistrationParams(max_iterations=3, max_correspondence_distance=100.0))
    S = sp.PointCloudShared(dev(src[:256]), covs=dev(scov[:256]))
    T_dev, lin, _ = reg.align_fused_loop(S, sp.PreparedTarget(g1, one.covs), write_neighbors=True)
    assert (reg.neighbors.indices.cpu().numpy() == 0).all() and reg._read_lin(lin).inlier == 256


def test_search_bounded_by_max_correspondence_distance_is_exact(sp, orc, gicp20k):
    """Without a neighbour output the fused kernels look for a neighbour only below max_correspondence_distance (a farther
    one is rejected anyway; with partial overlap that search was most of an iteration). Partial overlap on purpose: a third
    of the source shifted far outside the target, a tight distance. Pose and linear system must be bit-identical to the run
    that writes the neighbours (unbounded search), for the sorted and the unsorted source path, and the inliers must be the
    oracle's."""
    src, scov, tgt, tcov, T_gt = gicp20k
    src2 = src.copy()
    src2[::3, :3] += np.float32([37.0, -21.0, 9.0])  # no target point within any useful distance of these
    outs = []
    for write, sort in ((True, True), (False, True), (True, False), (False, False)):
        S = sp.PointCloudShared(dev(src2), covs=dev(scov))
        Tg = sp.PointCloudShared(dev(tgt), covs=dev(tcov))
        prep = sp.PreparedTarget(sp.GridKNN.build(Tg.points), Tg.covs)
        reg = sp.Registration(sp.RegistrationParams(criteria_translation=0.0, criteria_rotation=0.0, max_iterations=6,
                                                    max_correspondence_distance=0.3))
        T_dev, lin, _ = reg.align_fused_loop(S, prep, write_neighbors=write, sort_by_cell=sort)
        outs.append((T_dev.cpu().numpy().copy(), lin.cpu().numpy().copy(), reg._read_lin(lin).inlier))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    assert np.array_equal(outs[2][0], outs[3][0]) and np.array_equal(outs[2][1], outs[3][1])
    assert 0 < outs[0][2] <= len(src) - len(src[::3])
    T = np.array(outs[0][0], np.float32).reshape(4, 4).T
    ref = orc.registration_align(__import__("oracle.pyoracle", fromlist=["RegParams"]).RegParams.defaults(
        crit_translation=0.0, crit_rotation=0.0, max_iterations=6, max_correspondence_distance=0.3), src2, scov, tgt, tcov)
    assert np.abs(T - ref["T"]).max() < 1e-5


def test_grid_order_and_presorted_source(sp, orc, gicp20k):
    """sp_grid_order is the grid's cell-order permutation; a source stored in that order aligns without the per-alignment
    sort (SP_SOURCE_PRESORTED) to the same pose, and reports its neighbours in the caller's (reordered) indexing."""
    from oracle.pyoracle import RegParams

    src, scov, tgt, tcov, T_gt = gicp20k
    S = sp.PointCloudShared(dev(src), covs=dev(scov))
    Tg = sp.PointCloudShared(dev(tgt), covs=dev(tcov))
    order = sp.GridKNN.build(S.points, points_per_cell=1.0).order()
    assert sorted(order.cpu().tolist()) == list(range(len(src)))
    S2 = S.reordered(order)
    assert torch.equal(S2.points, S.points[order]) and torch.equal(S2.covs, S.covs[order])
    grid = sp.GridKNN.build(Tg.points)
    prep = sp.PreparedTarget(grid, Tg.covs)
    p = sp.RegistrationParams(criteria_translation=0.0, criteria_rotation=0.0, max_iterations=8)
    ref = orc.registration_align(RegParams.defaults(crit_translation=0.0, crit_rotation=0.0, max_iterations=8), src, scov,
                                 tgt, tcov)
    poses = {}
    for mode, cloud in ((True, S), ("presorted", S2), (False, S2)):
        reg = sp.Registration(p)
        T_dev, lin, _ = reg.align_fused_loop(cloud, prep, sort_by_cell=mode, write_neighbors=True)
        poses[mode] = reg.T_from_device(T_dev)
        assert np.abs(poses[mode] - ref["T"]).max() < 1e-5
        assert reg._read_lin(lin).inlier == ref["inlier"]
        # neighbours of the last linearisation, against brute force at the pose they were searched at
        nn = reg.neighbors.indices.cpu().numpy().ravel()
        T7 = orc.registration_align(RegParams.defaults(crit_translation=0.0, crit_rotation=0.0, max_iterations=7), src, scov,
                                    tgt, tcov)["T"]
        pts = cloud.points.cpu().numpy()
        oi, _ = orc.knn_bruteforce(orc.transform_points(pts, T7), tgt, 1)
        assert (nn == oi.ravel()).mean() > 0.999  # the pose differs from the oracle's by rounding only


# ------------------------------------------------------------------ tile self-kNN (+ fused covariance / normals)
@pytest.mark.parametrize("k", [4, 10, 20])
@pytest.mark.parametrize("shape", ["uniform", "clustered"])
def test_grid_self_knn_and_fused_covariance(sp, orc, k, shape):
    n = 12000
    pts = cloud(orc, 11, n, 4.0)
    if shape == "clustered":  # dense blobs + empty space + far outliers: exercises the to-do (ring walk) path
        rs = np.random.RandomState(1)
        pts[: n // 2, :3] = (rs.normal(0, 0.05, (n // 2, 3)) + rs.randint(-3, 4, (n // 2, 1)) * 1.0).astype(np.float32)
        pts[-20:, :3] *= 50.0
    for ppc in (1.0, 8.0):
        grid = sp.GridKNN.build(dev(pts), points_per_cell=ppc)
        res, covs, nrm = grid.self_knn(k, want_knn=True, want_covs=True, want_normals=True)
        oi, od = orc.knn_bruteforce(pts, pts, k)
        assert np.array_equal(res.distances.cpu().numpy(), od)
        assert np.array_equal(res.indices.cpu().numpy(), oi)
        ocov = orc.cov_estimate(pts, oi)
        assert np.array_equal(covs.cpu().numpy(), ocov)
        onrm = orc.normals_from_cov(pts, ocov)
        dots = np.abs((nrm.cpu().numpy()[:, :3] * onrm[:, :3]).sum(1))
        assert np.percentile(dots, 2) > 1 - 1e-3


def test_grid_self_knn_1m_matches_kdtree_covariances(sp, orc):
    pts = cloud(orc, 1234, 1000000, 10.0)
    P = dev(pts)
    grid = sp.GridKNN.build(P, points_per_cell=8.0)
    res, covs, _ = grid.self_knn(20, want_knn=True, want_covs=True)
    tree = sp.KDTree.build(pts)
    ref = tree.knn_search(P, 20)
    assert torch.equal(res.distances, ref.distances)
    # indices may differ only inside exact-distance tie groups (KD-tree: first visited; grid: lowest index)
    diff = (res.indices != ref.indices).any(1)
    d = res.distances[diff]
    assert diff.float().mean().item() < 1e-3 and bool(((d[:, 1:] == d[:, :-1]).any(1)).all())
    same = ~diff
    assert torch.equal(covs[same], sp.covariance.estimate(ref, P)[same])
