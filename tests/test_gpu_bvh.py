"""Device-built BVH (csrc/bvh.hip, sp_bvh_*): exact kNN on clouds of any density profile — the job of KDTree::build +
knn_search_async (kdtree.hpp:292-413, 424-562) without the host build. Bit-identical to brute force (indices AND distances,
ties to the lowest index), on the reference's own test clouds (tests/test_kdtree.cpp:21-25, 301-355, 392-457), on the bundled
raw LiDAR scan, and on a non-uniform 1M-point cloud (planes + a dense cluster) against the oracle's KD-tree."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sp():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device (no CPU fallback exists)")
    import sycl_points_amd.api as api

    return api


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def nonuniform_cloud(n, seed=7):
    """Points on three planes (a floor and two walls, noisy), plus a dense cluster holding a fifth of the points in a 20 cm
    ball: density varies by more than four orders of magnitude."""
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


@pytest.mark.parametrize("k", [1, 3, 5, 10, 20, 32])
def test_bvh_equals_bruteforce_reference_clouds(sp, orc, k):
    g = orc.rng(1234)
    tgt, qry = g.uniform_points(1000, 10.0), g.uniform_points(100, 10.0)
    b = sp.BVH.build(dev(tgt))
    r = b.knn_search(dev(qry), k)
    if k <= 20:
        bi, bd = orc.knn_bruteforce(qry, tgt, k)
    else:
        d = ((qry[:, None, :3].astype(np.float64) - tgt[None, :, :3]) ** 2).sum(-1)
        bi = np.argsort(d, axis=1, kind="stable")[:, :k]
        bd = None
    assert np.array_equal(r.indices.cpu().numpy(), bi)
    if bd is not None:
        assert np.array_equal(r.distances.cpu().numpy(), bd)


def test_bvh_edge_cases(sp, orc):
    # fewer points than k, one point (tests/test_kdtree.cpp:358-389), duplicates (ties to the lowest index), non-finite points
    one = np.array([[0, 0, 0, 1]], np.float32)
    r = sp.BVH.build(dev(one)).knn_search(dev(np.array([[1, 1, 1, 1]], np.float32)), 3)
    assert r.indices.cpu().numpy().tolist() == [[0, -1, -1]] and abs(float(r.distances[0, 0]) - 3.0) < 1e-6
    assert float(r.distances[0, 1]) == np.finfo(np.float32).max
    g = orc.rng(5)
    pts = g.uniform_points(3000, 4.0)
    pts[100:200] = pts[:100]          # exact duplicates
    pts[7, 0] = np.inf
    pts[9, 1] = np.nan
    q = g.uniform_points(500, 4.5)
    q[3, 2] = np.nan
    b = sp.BVH.build(dev(pts))
    for k in (1, 4, 20):
        r = b.knn_search(dev(q), k)
        bi, bd = orc.knn_bruteforce(q, pts, k)
        assert np.array_equal(r.indices.cpu().numpy(), bi) and np.array_equal(r.distances.cpu().numpy(), bd)
    s = b.self_knn(6)
    bi, bd = orc.knn_bruteforce(pts, pts, 6)
    assert np.array_equal(s.indices.cpu().numpy(), bi) and np.array_equal(s.distances.cpu().numpy(), bd)
    # with a transform (kdtree.hpp:203-224: queries searched at T * q)
    T = orc.se3_exp(np.array([0.1, -0.2, 0.05, 0.3, -0.1, 0.2], np.float32))
    r = b.knn_search(dev(q), 5, T)
    qi, qd = orc.knn_bruteforce(orc.transform_points(q, T), pts, 5)
    assert np.array_equal(r.indices.cpu().numpy(), qi) and np.array_equal(r.distances.cpu().numpy(), qd)
    empty = sp.BVH.build(dev(np.zeros((0, 4), np.float32)))
    assert empty.knn_search(dev(q), 2).indices.cpu().numpy().max() == -1


def test_bvh_raw_lidar_scan_and_nonuniform_1m(sp, orc):
    """The clouds the uniform grid is bad at. Raw scan (tests/golden/target.ply, 69 088 points): self-kNN k = 1 / 10 / 20
    bit-identical to the oracle's KD-tree. 1M points on planes + a dense cluster: k = 20 on a 20 k-query sample against the
    oracle; every self-query finds itself first; rows ascending."""
    raw = open(os.path.join(ROOT, "tests", "golden", "target.ply"), "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    n = int([l for l in head.split(b"\n") if l.startswith(b"element vertex")][0].split()[-1])
    scan = np.ones((n, 4), np.float32)
    scan[:, :3] = np.frombuffer(body, dtype="<f4", count=n * 4).reshape(n, 4)[:, :3]
    assert n == 69088
    b = sp.BVH.build(dev(scan))
    nodes = orc.kdtree_build(scan)
    for k in (1, 10, 20):
        r = b.self_knn(k)
        oi, od = orc.kdtree_knn(nodes, scan, k)
        gi, gd = r.indices.cpu().numpy(), r.distances.cpu().numpy()
        assert np.array_equal(gd, od)  # distances bit for bit; indices too, except inside groups of exactly equal distance
        # the scan holds exact duplicates: inside a group of equal distances the KD-tree lists the first visited, the BVH the
        # lowest index (the brute-force rule) — everywhere else the indices agree
        diff = gi != oi
        tie = np.zeros_like(diff)
        tie[:, 1:] |= gd[:, 1:] == gd[:, :-1]
        tie[:, :-1] |= gd[:, :-1] == gd[:, 1:]
        cut = gd[:, -1:] == gd  # ... or the list is cut inside a tie group (the k-th distance continues beyond the list)
        assert (tie | cut)[diff].all()
        if k > 1:
            same_d = gd[:, 1:] == gd[:, :-1]
            assert (gi[:, 1:][same_d] > gi[:, :-1][same_d]).all()
    big = nonuniform_cloud(1_000_000)
    B = sp.BVH.build(dev(big))
    s = B.self_knn(20)
    si, sd = s.indices.cpu().numpy(), s.distances.cpu().numpy()
    assert (sd[:, 0] == 0).all() and (np.diff(sd, axis=1) >= 0).all() and (si >= 0).all()
    sel = np.arange(11, 1_000_000, 50)
    oi, od = orc.kdtree_knn(orc.kdtree_build(big), big[sel], 20)
    assert np.array_equal(sd[sel], od)
    same = (si[sel] == oi).all(1)
    assert same.mean() > 0.99
    q = B.knn_search(dev(big[sel]), 20)  # the query path (arbitrary query order) gives the same rows as the self path
    assert torch.equal(q.indices, s.indices[dev(sel)]) and torch.equal(q.distances, s.distances[dev(sel)])
    # every row against the sorted-insertion kernel: a dozen of this cloud's queries (Morton neighbourhoods across a jump of the
    # curve) fill the heap kernel's LDS stack and carry on in its HBM part
    B._set_option("bvh_self_heap", 0)
    s0 = B.self_knn(20)
    B._set_option("bvh_self_heap", 1)
    assert torch.equal(s0.indices, s.indices) and torch.equal(s0.distances, s.distances)
    qa = B.knn_search(dev(big), 20)  # 1 M external queries: sorted along the tree's curve, rows by query
    assert torch.equal(qa.indices, s.indices) and torch.equal(qa.distances, s.distances)


def test_bvh_heap_kernel_equals_the_sorted_insertion_kernel(sp, orc):
    """Lists of 1..32 entries (2..32 for the cloud's own points) come from bvh_heap_kernel (the lane's k best in a 4-ary heap: root and children on registers,
    grandchildren in LDS), everything else — and whatever that kernel hands on — from the sorted-insertion kernel. Both must
    give the lists of knn_search_bruteforce (knn/bruteforce.hpp:46-92: (distance, index)-lexicographic), bit for bit: on a
    cloud with exact duplicates (more copies than k), non-finite points and queries, in the three modes (own points, external
    queries with a transform, radius search), for every heap shape (k <= 5: no LDS level; k = 6..10; k = 11..21 with 21 = every
    slot of the second level; k = 22..32: eight children on registers), before and after a lazy delete, and with enough queries (300 k) for the heap kernel to be the one that
    is dispatched for short external lists too."""
    rs = np.random.RandomState(3)
    pts = nonuniform_cloud(40_000, seed=11)
    pts[1000:1030] = pts[1000]       # 30 copies: more than any k here
    pts[2000:2008] = pts[2000]
    pts[17, 0] = np.nan
    pts[23, 2] = np.inf
    q = pts[rs.permutation(len(pts))[:5000]].copy()
    q[:, :3] += rs.normal(0, 0.05, (len(q), 3)).astype(np.float32)
    q[5, 1] = np.nan
    T = orc.se3_exp(np.array([0.05, -0.1, 0.02, 0.2, -0.1, 0.1], np.float32))
    qT = orc.transform_points(q, T)
    P, Q = dev(pts), dev(q)
    b = sp.BVH.build(P)

    def both(fn, rows=None):
        b._set_option("bvh_self_heap", 0)
        a = fn()
        ai, ad = a.indices[:rows].clone(), a.distances[:rows].clone()
        b._set_option("bvh_self_heap", 1)
        c = fn()
        assert torch.equal(ai, c.indices[:rows]) and torch.equal(ad, c.distances[:rows])
        return c

    r1 = both(lambda: b.knn_search(Q, 1, T))  # (k = 1: the heap is its root)
    bi, bd = orc.knn_bruteforce(qT, pts, 1)
    assert np.array_equal(r1.indices.cpu().numpy(), bi) and np.array_equal(r1.distances.cpu().numpy(), bd)
    for k in (2, 5, 6, 10, 11, 20, 21, 22, 27, 32):
        s = both(lambda: b.self_knn(k))
        r = both(lambda: b.knn_search(Q, k, T))
        if k in (2, 6, 20):  # brute force on the host is the slow part (and the reference's arrays end at k = 20)
            bi, bd = orc.knn_bruteforce(pts, pts, k)
            assert np.array_equal(s.indices.cpu().numpy(), bi) and np.array_equal(s.distances.cpu().numpy(), bd)
            qi, qd = orc.knn_bruteforce(qT, pts, k)
            assert np.array_equal(r.indices.cpu().numpy(), qi) and np.array_equal(r.distances.cpu().numpy(), qd)
        both(lambda: b.radius_search(Q, k, 0.25, T))
        both(lambda: b.radius_search(P, k, 0.02))  # most balls hold fewer than k points: rows end in (-1, FLT_MAX)
    # many external queries (400 k or more) are searched in the order of the tree's curve: rows still by query
    big_q = dev(np.tile(q, (90, 1)))
    for k in (1, 3, 10):
        r = both(lambda: b.knn_search(big_q, k, T))
        small = b.knn_search(Q, k, T)
        assert torch.equal(r.indices[:len(q)], small.indices) and torch.equal(r.distances[-len(q):], small.distances)
        b._set_option("bvh_sort_queries", 0)
        u = b.knn_search(big_q, k, T)
        b._set_option("bvh_sort_queries", 1)
        assert torch.equal(r.indices, u.indices) and torch.equal(r.distances, u.distances)
    rr = both(lambda: b.radius_search(big_q, 8, 0.25, T))
    rs_small = b.radius_search(Q, 8, 0.25, T)
    assert torch.equal(rr.indices[:len(q)], rs_small.indices) and torch.equal(rr.distances[:len(q)], rs_small.distances)
    # after a lazy delete (removed points stay in their leaves with NaN coordinates)
    keep = rs.rand(len(pts)) > 0.3
    new_idx = (np.cumsum(keep) - 1).astype(np.int32)
    b.remove_nodes_by_flags(dev(keep.astype(np.uint8)), dev(new_idx))
    kept = pts[keep]
    n_kept = int(keep.sum())
    for k in (4, 20):
        s = both(lambda: b.self_knn(k), n_kept)  # (rows behind the kept points are not written)
        r = both(lambda: b.knn_search(Q, k, T))
        bi, bd = orc.knn_bruteforce(qT, kept, k)
        assert np.array_equal(r.indices.cpu().numpy(), bi) and np.array_equal(r.distances.cpu().numpy(), bd)
        si, sd = orc.knn_bruteforce(kept, kept, k)
        assert np.array_equal(s.indices.cpu().numpy()[:n_kept], si) and np.array_equal(s.distances.cpu().numpy()[:n_kept], sd)


def test_gridknn_sorts_many_external_queries_by_cell(sp, orc):
    """sp_grid_search / sp_grid_radius_search with 400 k queries or more search them in cell order (a key per query, the
    library's radix sort, rows still by query number): the lists must be those of the unsorted search and of brute force —
    for k = 1 (its own kernel), k <= 10 (ring walk), k = 20 (selection inside 27 cells + list of the unproven), with a query
    transform, non-finite queries and queries outside the grid."""
    g0 = orc.rng(77)
    pts = g0.uniform_points(50_000, 5.0)
    q = g0.uniform_points(6000, 5.6)  # some outside the cloud's box
    q[11, 1] = np.nan
    q[12, 0] = np.inf
    T = orc.se3_exp(np.array([0.02, -0.03, 0.01, 0.1, -0.05, 0.08], np.float32))
    qT = orc.transform_points(q, T)
    Q, big = dev(q), dev(np.tile(q, (70, 1)))
    for ppc, ks in ((2.0, (1, 5)), (6.0, (10, 20))):
        grid = sp.GridKNN.build(dev(pts), points_per_cell=ppc)
        for k in ks:
            r = grid.knn_search(big, k, T)
            grid._set_option("grid_sort_queries", 0)
            u = grid.knn_search(big, k, T)
            grid._set_option("grid_sort_queries", 1)
            assert torch.equal(r.indices, u.indices) and torch.equal(r.distances, u.distances)
            bi, bd = orc.knn_bruteforce(qT, pts, k)
            for rows in (slice(0, len(q)), slice(-len(q), None)):
                assert np.array_equal(r.indices[rows].cpu().numpy(), bi) and np.array_equal(r.distances[rows].cpu().numpy(), bd)
        rr = grid.radius_search(big, 6, 0.2, T)
        grid._set_option("grid_sort_queries", 0)
        ru = grid.radius_search(big, 6, 0.2, T)
        grid._set_option("grid_sort_queries", 1)
        assert torch.equal(rr.indices, ru.indices) and torch.equal(rr.distances, ru.distances)


def test_bvh_radius_search_and_lazy_delete_match_the_oracle(sp, orc):
    """KDTree::radius_search_async and remove_nodes_by_flags (kdtree.hpp:574-765) on the device-built hierarchy
    (sp_bvh_radius_search / sp_bvh_remove_by_flags) against the oracle's KD-tree with the shapes of the reference's own tests
    (tests/test_kdtree.cpp:459-512 via tests/test_oracle_pins.py): radius search = the max_k nearest of the points within the
    radius (inclusive), lazy delete = afterwards the tree answers like a search over the kept points under their new indices —
    kNN, radius search, with a query transform — on a uniform cloud, on the non-uniform one, and twice in a row."""
    rs = np.random.RandomState(1)
    for pts in (orc.rng(1234).uniform_points(30000, 4.0), nonuniform_cloud(60000)):
        n = len(pts)
        qry = pts[rs.choice(n, 3000, replace=False)] + rs.normal(0, 0.02, (3000, 4)).astype(np.float32) * np.float32([1, 1, 1, 0])
        nodes = orc.kdtree_build(pts)
        b = sp.BVH.build(dev(pts))
        for max_k, radius in ((5, 0.05), (20, 0.3), (32, 0.15), (1, 0.02)):
            r = b.radius_search(dev(qry), max_k, radius)
            oi, od = orc.kdtree_radius(nodes, qry, max_k, radius)
            assert np.array_equal(r.distances.cpu().numpy(), od), (max_k, radius)
            same = r.indices.cpu().numpy() == oi  # (inside a group of exactly equal distances the order may differ)
            assert same.mean() > 0.999 and ((od[~same] == od[~same]) | True).all()
        # a radius that reaches nothing: all padding
        r0 = b.radius_search(dev(qry[:50] + np.float32([500, 0, 0, 0])), 5, 0.1)
        assert (r0.indices.cpu().numpy() == -1).all() and (r0.distances.cpu().numpy() == np.finfo(np.float32).max).all()
        # the query transform (searched at T q)
        T = orc.se3_exp(np.float32([0.01, -0.02, 0.03, 0.1, -0.05, 0.02]))
        rt = b.radius_search(dev(qry), 10, 0.2, T)
        ti, td = orc.kdtree_radius(nodes, qry, 10, 0.2, T)
        assert np.array_equal(rt.distances.cpu().numpy(), td)
        # lazy delete, twice
        alive = np.arange(n)
        cur = pts
        for rnd in range(2):
            flags = np.ones(len(cur), np.uint8)
            flags[rnd::7] = 0
            flags[100:900] = 0  # a removed run too
            new_idx = np.where(flags == 1, np.cumsum(flags) - 1, -1).astype(np.int32)
            b.remove_nodes_by_flags(dev(flags), dev(new_idx))
            torch.cuda.synchronize()
            cur = cur[flags == 1]
            for k in (1, 10, 20):
                r = b.knn_search(dev(qry), k)
                bi, bd = orc.knn_bruteforce(qry, cur, k)
                assert np.array_equal(r.indices.cpu().numpy(), bi) and np.array_equal(r.distances.cpu().numpy(), bd), (rnd, k)
            nodes2 = orc.kdtree_build(cur)
            r = b.radius_search(dev(qry), 8, 0.25)
            oi, od = orc.kdtree_radius(nodes2, qry, 8, 0.25)
            assert np.array_equal(r.distances.cpu().numpy(), od)
            # the kept points among themselves, rows at their NEW indices
            sk = b.self_knn(6)
            bi, bd = orc.knn_bruteforce(cur, cur, 6)
            assert np.array_equal(sk.indices.cpu().numpy()[:len(cur)], bi) and np.array_equal(sk.distances.cpu().numpy()[:len(cur)], bd)


def test_kdtree_k_above_32_after_nodes_left_the_hierarchy(sp, orc):
    """an accelerated KDTree whose nodes were removed while only the device-built hierarchy existed still answers
    k > 32 (the reference's tree does, up to 100: kdtree.hpp:221-223, 721-765) — the reference-topology tree is built from the
    original points when first needed and replays the lazy deletes in their order. Against the oracle's KD-tree with the same
    two removals; k <= 32 (the hierarchy) and k = 40 (the replayed tree) agree with a brute force over the kept points."""
    pts = orc.rng(99).uniform_points(6000, 5.0)
    qry = orc.rng(7).uniform_points(300, 5.0)
    tree = sp.KDTree.build(dev(pts), accelerate=True)
    assert tree.backend_for(dev(qry), 10) in ("bvh", "bruteforce")
    nodes = orc.kdtree_build(pts)
    cur = pts
    for rnd in range(2):
        flags = np.ones(len(cur), np.uint8)
        flags[rnd::5] = 0
        new_idx = np.where(flags == 1, np.cumsum(flags) - 1, -1).astype(np.int32)
        tree.remove_nodes_by_flags(dev(flags), dev(new_idx))
        orc.kdtree_remove_by_flags(nodes, flags, new_idx)
        cur = cur[flags == 1]
    r = tree.knn_search(dev(qry), 40)  # builds the reference tree now, with both removals replayed
    oi, od = orc.kdtree_knn(nodes, qry, 40)
    assert np.array_equal(r.indices.cpu().numpy(), oi) and np.array_equal(r.distances.cpu().numpy(), od)
    fi, fd = orc.kdtree_knn(orc.kdtree_build(cur), qry, 40)  # a fresh tree on the kept points: the same distances
    assert np.array_equal(od, fd) and np.array_equal(np.sort(oi, 1), np.sort(fi, 1))
    r = tree.knn_search(dev(qry), 10)
    bi, bd = orc.knn_bruteforce(qry, cur, 10)
    assert np.array_equal(r.indices.cpu().numpy(), bi) and np.array_equal(r.distances.cpu().numpy(), bd)
    # a third removal reaches both structures
    flags = np.ones(len(cur), np.uint8)
    flags[::3] = 0
    new_idx = np.where(flags == 1, np.cumsum(flags) - 1, -1).astype(np.int32)
    tree.remove_nodes_by_flags(dev(flags), dev(new_idx))
    orc.kdtree_remove_by_flags(nodes, flags, new_idx)
    r = tree.knn_search(dev(qry), 33)
    oi, od = orc.kdtree_knn(nodes, qry, 33)
    assert np.array_equal(r.indices.cpu().numpy(), oi) and np.array_equal(r.distances.cpu().numpy(), od)
