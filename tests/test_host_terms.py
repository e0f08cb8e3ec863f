"""Host-side pose terms of Registration::align (degenerate regularisation, MAP prior, se3_log): C-ABI host functions
against the oracle's restatement, and the oracle against independent linear algebra (numpy / scipy in float64).
No GPU needed: these entry points run on the host (the reference runs them on the host with Eigen)."""
import ctypes as C

import numpy as np
import pytest

from oracle.pyoracle import Oracle


@pytest.fixture(scope="module")
def orc():
    return Oracle()


@pytest.fixture(scope="module")
def L():
    from sycl_points_amd import _lib
    _lib.build()  # hipcc cross-compiles without a GPU; nothing happens when the library is current
    return _lib.lib()


def _spd(rng, scale=1.0):
    A = rng.normal(size=(6, 40))
    return (A @ A.T * scale).astype(np.float32)


def test_oracle_degenerate_regularisation_against_numpy(orc):
    """degenerate_regularization.hpp:60-110 restated in float64 numpy: P = lambda * sum of v v^T over weak directions."""
    rng = np.random.default_rng(0)
    for _ in range(8):
        H, b = _spd(rng), rng.normal(size=6).astype(np.float32)
        Tc, Ti = orc.se3_exp(rng.normal(size=6) * 0.2), orc.se3_exp(rng.normal(size=6) * 0.2)
        inl = 10
        wr, Vr = np.linalg.eigh(H[:3, :3].astype(np.float64))
        wt, Vt = np.linalg.eigh(H[3:, 3:].astype(np.float64))
        rt, tt = float(wr[1] / inl) * 1.01, float(wt[0] / inl) * 1.01  # two weak rotation axes, one weak translation axis
        P = np.zeros((6, 6))
        for i in range(3):
            if wr[i] / inl < rt:
                P[:3, :3] += 2.0 * inl * np.outer(Vr[:, i], Vr[:, i])
            if wt[i] / inl < tt:
                P[3:, 3:] += 2.0 * inl * np.outer(Vt[:, i], Vt[:, i])
        tw = orc.se3_log((np.linalg.inv(Ti.astype(np.float64)) @ Tc).astype(np.float32)).astype(np.float64)
        H2, b2 = orc.degenerate_regularize(H, b, inl, Tc, Ti, rot_thr=rt, trans_thr=tt, base_factor=2.0)
        assert np.abs(H2 - (H + P)).max() <= 1e-5 * np.abs(H).max()
        assert np.abs(b2 - (b + P @ tw)).max() <= 1e-5 * max(1.0, np.abs(P @ tw).max())
        assert np.linalg.matrix_rank(P) == 3
    # type none / no inliers: untouched
    H, b = _spd(rng), rng.normal(size=6).astype(np.float32)
    H3, b3 = orc.degenerate_regularize(H, b, 0, np.eye(4), np.eye(4))
    assert np.array_equal(H3, H) and np.array_equal(b3, b)


def test_oracle_map_prior_against_numpy(orc):
    """map_prior.hpp:97-201: Omega = (H_curr^-1 + Q)^-1 and the cost e^T Omega e / 2, in float64 numpy / scipy."""
    from scipy.spatial.transform import Rotation

    rng = np.random.default_rng(1)
    for _ in range(8):
        H = _spd(rng, 50.0)
        Tprev = orc.se3_exp(rng.normal(size=6) * 0.3)
        Tpred = orc.isometry_mul(Tprev, orc.se3_exp(rng.normal(size=6) * 0.1))
        err, inl, sig = 700.0, 400, (0.8, 1.2, 0.05, 0.02)
        has, Om, Tinv = orc.map_prior_update(H, err, inl, Tprev, Tpred, sig)
        assert has
        s2 = max(1.0, 2 * err / (3 * inl - 6))
        Rrel = Tprev[:3, :3].T.astype(np.float64) @ Tpred[:3, :3]
        rv = Rotation.from_matrix(Rrel).as_rotvec()
        dt = Tpred[:3, :3].T.astype(np.float64) @ (Tpred[:3, 3] - Tprev[:3, 3])
        q = np.concatenate([np.abs(rv) * sig[0] ** 2 + sig[2] ** 2, np.abs(dt) * sig[1] ** 2 + sig[3] ** 2])
        Ad = np.zeros((6, 6))
        Ad[:3, :3] = Ad[3:, 3:] = Rrel
        Om_ref = np.linalg.inv(np.linalg.inv(Ad.T @ (H.astype(np.float64) / s2) @ Ad) + np.diag(q))
        assert np.abs(Om - Om_ref).max() <= 2e-5 * np.abs(Om_ref).max()
        assert np.abs(Tinv - np.linalg.inv(Tpred.astype(np.float64))).max() < 1e-6
        Test = orc.isometry_mul(Tpred, orc.se3_exp(rng.normal(size=6) * 0.05))
        e = orc.se3_log((np.linalg.inv(Tpred.astype(np.float64)) @ Test).astype(np.float32)).astype(np.float64)
        b = rng.normal(size=6).astype(np.float32)
        H2, b2, e2, pe = orc.map_prior_apply(Om, Tinv, H, b, 7.0, Test)
        cost = 0.5 * e @ Om_ref @ e
        assert abs(pe - cost) <= 1e-3 * max(cost, 1e-6) and abs(e2 - 7.0 - cost) <= 1e-3 * max(cost, 1e-3)
        assert np.abs(H2 - (H + Om_ref)).max() <= 2e-5 * np.abs(H2).max()
        assert np.abs(b2 - (b + Om_ref @ e)).max() <= 1e-3 * max(1.0, np.abs(Om_ref @ e).max())
    # no prior when it cannot be formed: too few inliers, negative / non-finite error
    assert not orc.map_prior_update(H, 1.0, 2, Tprev, Tpred)[0]
    assert not orc.map_prior_update(H, -1.0, 100, Tprev, Tpred)[0]
    assert not orc.map_prior_update(H, float("nan"), 100, Tprev, Tpred)[0]


def test_oracle_rotation_divergence_properties(orc):
    """rotation_constraint.hpp:15-128: D(C, C) = 0 at the identity; D >= 0; the gradient matches a finite difference of
    D along body-frame rotations."""
    rng = np.random.default_rng(2)
    n = 64
    pts = np.zeros((n, 4), np.float32)
    pts[:, 3] = 1
    covs = np.zeros((n, 4, 4), np.float32)
    for i in range(n):
        A = rng.normal(size=(3, 3)) * np.array([1.0, 0.3, 0.05])
        covs[i, :3, :3] = (A @ A.T + 1e-3 * np.eye(3)).astype(np.float32)
    covs = covs.reshape(n, 16)
    idx = np.arange(n, dtype=np.int32)
    d2 = np.zeros(n, np.float32)
    same, (e0, _) = orc.gicp_linearize_rot(pts, covs, pts, covs, None, idx, d2, np.eye(4), 2.0, "POINT_TO_POINT", "NONE",
                                           10.0, 1.0, 1.0, 10.0)
    assert abs(e0) < 1e-6 and np.abs(same["b"]).max() < 1e-6  # D == 0: no error, no gradient
    covs_t = covs.reshape(n, 4, 4)[rng.permutation(n)].reshape(n, 16)

    def err(tw):
        T = orc.se3_exp(tw)
        return orc.gicp_linearize_rot(pts, covs, pts, covs_t, None, idx, d2, T, 2.0, "POINT_TO_POINT", "NONE", 10.0, 1.0,
                                      1.0, 10.0)

    w0 = np.array([0.2, -0.1, 0.15, 0, 0, 0], np.float32)
    lin, (e_here, _) = err(w0)
    assert e_here > 1e-2
    T0 = orc.se3_exp(w0)
    h = 2e-3
    # The points coincide, so the ICP term is 0 and the error is the constraint's alone. The reference feeds
    # sqrt(D^2 / 2) to compute_error (rho(r) = r^2 / 2 for NONE), i.e. error = sum D^2 / 4, while b = sum D J
    # (registration.hpp:634-650): d error / d w_a = b[a] / 2.
    for a in range(3):
        dw = np.zeros(6, np.float32)
        dw[a] = h
        Tp, Tm = orc.isometry_mul(T0, orc.se3_exp(dw)), orc.isometry_mul(T0, orc.se3_exp(-dw))
        ep = orc.gicp_linearize_rot(pts, covs, pts, covs_t, None, idx, d2, Tp, 2.0, "POINT_TO_POINT", "NONE", 10.0, 1.0,
                                    1.0, 10.0)[1][0]
        em = orc.gicp_linearize_rot(pts, covs, pts, covs_t, None, idx, d2, Tm, 2.0, "POINT_TO_POINT", "NONE", 10.0, 1.0,
                                    1.0, 10.0)[1][0]
        fd = (ep - em) / (2 * h)
        assert abs(fd - 0.5 * lin["b"][a]) <= 0.03 * max(abs(fd), 0.5 * np.abs(lin["b"][:3]).max()), (a, fd, lin["b"][:3])


def test_cabi_se3_log_matches_oracle(orc, L):
    rng = np.random.default_rng(3)
    for mag in (1e-8, 1e-3, 0.3, 1.5, 3.0):
        for _ in range(6):
            T = orc.se3_exp(rng.normal(size=6).astype(np.float32) * np.float32(mag))
            Tc = np.ascontiguousarray(T.T)
            out = np.zeros(6, np.float32)
            L.sp_se3_log_host(Tc.ctypes.data, out.ctypes.data)
            ref = orc.se3_log(T)
            assert np.abs(out - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max())
            assert np.abs(orc.se3_exp(out) - T).max() < 1e-5  # exp(log(T)) == T (tests/test_eigen_utils.cpp:702-720)


def test_cabi_degenerate_regularisation_matches_oracle(orc, L):
    from sycl_points_amd._lib import DegenerateRegParams

    rng = np.random.default_rng(4)
    for _ in range(10):
        H, b = _spd(rng), rng.normal(size=6).astype(np.float32)
        Tc, Ti = orc.se3_exp(rng.normal(size=6) * 0.2), orc.se3_exp(rng.normal(size=6) * 0.2)
        rt = float(np.median(np.linalg.eigvalsh(H[:3, :3]))) / 10 * 1.01
        tt = float(np.median(np.linalg.eigvalsh(H[3:, 3:]))) / 10 * 1.01
        Ho, bo = orc.degenerate_regularize(H, b, 10, Tc, Ti, rot_thr=rt, trans_thr=tt, base_factor=2.0)
        Hp, bp = H.copy(), b.copy()
        dp = DegenerateRegParams(1, rt, tt, 2.0)
        Tcc, Tic = np.ascontiguousarray(Tc.T), np.ascontiguousarray(Ti.T)
        assert L.sp_degenerate_regularize_host(C.byref(dp), Hp.ctypes.data, bp.ctypes.data, 10, Tcc.ctypes.data,
                                               Tic.ctypes.data) == 0
        assert np.abs(Hp - H).max() > 1.0
        assert np.abs(Hp - Ho).max() <= 1e-5 * np.abs(Ho).max()
        assert np.abs(bp - bo).max() <= 1e-5 * max(1.0, np.abs(bo).max())
        # thresholds <= 0 switch a block off; type NONE and inlier == 0 leave everything untouched
        Hq, bq = H.copy(), b.copy()
        dq = DegenerateRegParams(1, 0.0, tt, 2.0)
        L.sp_degenerate_regularize_host(C.byref(dq), Hq.ctypes.data, bq.ctypes.data, 10, Tcc.ctypes.data, Tic.ctypes.data)
        assert np.array_equal(Hq[:3, :3], H[:3, :3]) and not np.array_equal(Hq[3:, 3:], H[3:, 3:])
        for d0, inl in ((DegenerateRegParams(0, rt, tt, 2.0), 10), (dp, 0)):
            Hn, bn = H.copy(), b.copy()
            assert L.sp_degenerate_regularize_host(C.byref(d0), Hn.ctypes.data, bn.ctypes.data, inl, Tcc.ctypes.data,
                                                   Tic.ctypes.data) == 0
            assert np.array_equal(Hn, H) and np.array_equal(bn, b)


def test_cabi_map_prior_matches_oracle(orc, L):
    from sycl_points_amd._lib import MapPriorParams, MapPriorState

    rng = np.random.default_rng(5)
    sig = (0.8, 1.2, 0.05, 0.02)
    for _ in range(10):
        H = _spd(rng, 50.0)
        Tprev = orc.se3_exp(rng.normal(size=6) * 0.3)
        Tpred = orc.isometry_mul(Tprev, orc.se3_exp(rng.normal(size=6) * 0.1))
        has, Om, Tinv = orc.map_prior_update(H, 123.0, 500, Tprev, Tpred, sig)
        st, mp = MapPriorState(), MapPriorParams(1, *sig)
        Hc, Tp, Tq = np.ascontiguousarray(H), np.ascontiguousarray(Tprev.T), np.ascontiguousarray(Tpred.T)
        assert L.sp_map_prior_update_host(C.byref(mp), Hc.ctypes.data, 123.0, 500, Tp.ctypes.data, Tq.ctypes.data,
                                          C.byref(st)) == 0
        assert st.has_prior == 1 and has
        Omp = np.array(st.omega, np.float32).reshape(6, 6)
        assert np.abs(Omp - Om).max() <= 2e-5 * np.abs(Om).max()
        assert np.abs(np.array(st.T_pred_inv, np.float32).reshape(4, 4).T - Tinv).max() < 1e-6
        Test = orc.isometry_mul(Tpred, orc.se3_exp(rng.normal(size=6) * 0.05))
        Te = np.ascontiguousarray(Test.T)
        b = rng.normal(size=6).astype(np.float32)
        Ho, bo, eo, pe = orc.map_prior_apply(Om, Tinv, H, b, 7.0, Test)
        Hp, bp, err = H.copy(), b.copy(), C.c_float(7.0)
        pep = L.sp_map_prior_apply_host(C.byref(st), Te.ctypes.data, Hp.ctypes.data, bp.ctypes.data, C.byref(err))
        assert abs(pep - pe) <= 1e-4 * max(1.0, abs(pe)) and abs(err.value - eo) <= 1e-4 * max(1.0, abs(eo))
        assert np.abs(Hp - Ho).max() <= 2e-5 * np.abs(Ho).max()
        assert np.abs(bp - bo).max() <= 1e-4 * max(1.0, np.abs(bo).max())
        # cost only (LM / dogleg trial poses): H and b untouched
        assert abs(L.sp_map_prior_apply_host(C.byref(st), Te.ctypes.data, None, None, None) - pep) == 0.0
    # disabled / degenerate inputs: no prior, apply is a no-op returning 0
    for mp2, e, inl in ((MapPriorParams(0, *sig), 123.0, 500), (mp, 123.0, 2), (mp, -1.0, 500), (mp, float("inf"), 500)):
        st2 = MapPriorState()
        st2.has_prior = 1
        assert L.sp_map_prior_update_host(C.byref(mp2), Hc.ctypes.data, e, inl, Tp.ctypes.data, Tq.ctypes.data,
                                          C.byref(st2)) == 0
        assert st2.has_prior == 0
        assert L.sp_map_prior_apply_host(C.byref(st2), Te.ctypes.data, None, None, None) == 0.0
