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
