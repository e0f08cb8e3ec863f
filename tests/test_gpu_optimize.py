"""The device-resident optimiser loop (sp_gicp_align_optimize, csrc/registration_opt.hip): Registration::align for GN / LM /
Powell dog-leg (registration.hpp:201-276, 803-964) and pipeline::RobustAligner's annealing levels (pipeline/robust.hpp:78-111)
as ONE launch and ONE read-back. Checker: the oracle's align() / align_robust_annealing() on the same clouds — final pose to
1e-5, the same iteration count / convergence flag / inliers, and the same SEQUENCE of optimiser decisions (trial evaluations
and accepted / rejected per outer iteration). One workgroup (the reference pipeline's 1000-point sample), several workgroups
(arrival counter between steps), and the host-driven loop of the same library beside it."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def sp():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device (no CPU fallback exists)")
    import sycl_points_amd.api as api

    return api


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def inputs(orc, n, density_scale=1.0, seed=1234):
    from sycl_points_amd.synthetic import gicp_pair

    r = 10.0 * (n * density_scale / 1e6) ** (1.0 / 3.0)
    src, tgt, T_gt = gicp_pair(n, r, seed)
    ti, _ = orc.kdtree_knn(orc.kdtree_build(tgt), tgt, 20)
    si, _ = orc.kdtree_knn(orc.kdtree_build(src), src, 20)
    return src, orc.cov_estimate(src, si), tgt, orc.cov_estimate(tgt, ti), T_gt


@pytest.fixture(scope="module", params=[(900, 8.0), (20000, 1.0), (20000, 8.0)],
                ids=["one-workgroup-900", "20k-config4-density", "20k-sparse"])
def clouds(orc, request):
    n, d = request.param
    return inputs(orc, n, d)


CASES = [
    dict(opt="LM", reg_type="GICP", loss="GEMAN_MCCLURE", scale=0.5),   # BASELINE config 1's optimiser + kernel
    dict(opt="LM", reg_type="POINT_TO_DISTRIBUTION", loss="NONE", scale=10.0),
    dict(opt="LM", reg_type="GICP", loss="HUBER", scale=0.5),
    dict(opt="LM", reg_type="GICP", loss="NONE", scale=10.0),
    dict(opt="DOGLEG", reg_type="GICP", loss="NONE", scale=10.0),
    dict(opt="DOGLEG", reg_type="POINT_TO_DISTRIBUTION", loss="NONE", scale=10.0),
    dict(opt="DOGLEG", reg_type="GICP", loss="CAUCHY", scale=0.3),
    dict(opt="GN", reg_type="POINT_TO_DISTRIBUTION", loss="CAUCHY", scale=0.3),
    dict(opt="GN", reg_type="GICP", loss="NONE", scale=10.0),
]


def oracle_params(case, **kw):
    from oracle.pyoracle import LOSS, OPT, REG, RegParams

    return RegParams.defaults(reg_type=REG[case["reg_type"]], robust_type=LOSS[case["loss"]], robust_default_scale=case["scale"],
                              optimization_method=OPT[case["opt"]], **kw)


def check_against_oracle(res, ref, case, levels=1):
    assert np.abs(res.T - ref["T"]).max() < 1e-5, (case, np.abs(res.T - ref["T"]).max())
    # The optimisers branch on float comparisons (LM: new_error <= current_error; dog-leg: rho against eta1 / eta2). The oracle's
    # trace says how far each iteration's closest decision was from its threshold: wherever that margin is outside rounding
    # (the K11 / K12 sums of the two implementations agree to ~1e-5 relative) the decisions have to be the oracle's; at the
    # first iteration decided inside rounding (two errors of a converged alignment that differ in the last bits) the branch may
    # differ and everything after it with it — the pose bound above holds regardless.
    got = [(e["trials"], e["accepted"]) for e in res.log]
    want = [(s["trials"], s["accepted"]) for s in ref["steps"]]
    tol = 2e-4  # relative change of an error sum that would flip the closest decision
    decided = next((i for i, s in enumerate(ref["steps"]) if s["margin"] < tol), len(want))
    assert got[:decided] == want[:decided], (case, got, want, decided)
    for e, s in zip(res.log[:decided], ref["steps"][:decided]):
        assert abs(e["damping"] - s["damping"]) <= 1e-6 * abs(s["damping"]), (case, e, s)
        assert abs(e["error"] - s["error"]) <= 2e-4 * abs(s["error"]), (case, e, s)
    if decided == len(want):
        assert got == want
        assert res.converged == ref["converged"] and res.iterations == ref["iterations"], case
        assert res.inlier == ref["inlier"]
        assert abs(res.error - ref["error"]) <= 2e-4 * abs(ref["error"])
        assert res.linearizations == len(want) and res.trials == sum(t for t, _ in want)
    assert decided >= min(2, len(want)), "a test case whose FIRST decisions are inside rounding checks nothing: pick another"
    assert max(e["level"] for e in res.log) == levels - 1


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c['opt']}-{c['reg_type']}-{c['loss']}")
def test_device_resident_optimiser_matches_oracle(sp, orc, clouds, case):
    src, scov, tgt, tcov, T_gt = clouds
    S = sp.PointCloudShared(dev(src), covs=dev(scov))
    Tg = sp.PointCloudShared(dev(tgt), covs=dev(tcov))
    prep = sp.PreparedTarget(sp.GridKNN.build(Tg.points), Tg.covs, reg_type=case["reg_type"])
    T0 = orc.se3_exp([0.01, -0.005, 0.02, 0.05, -0.04, 0.03])
    p = sp.RegistrationParams(reg_type=case["reg_type"], robust_type=case["loss"], robust_default_scale=case["scale"],
                              optimization_method=case["opt"], max_iterations=25)
    ref = orc.registration_align(oracle_params(case, max_iterations=25), src, scov, tgt, tcov, init_T=T0, steps=True)
    reg = sp.Registration(p)
    res = reg.align_optimize(S, prep, T0, [case["scale"]])
    assert res is not None, "the persistent launch must be available on an idle MI355X"
    check_against_oracle(res, ref, case)
    # the same alignment through the host-driven loop of the library (one read-back per step): same decisions, same pose
    host = sp.Registration(p).align_prepared(S, prep, initial_guess=T0, device_resident=False)
    assert np.abs(host.T - res.T).max() < 2e-6 and host.iterations == res.iterations and host.converged == res.converged
    assert host.inlier == res.inlier
    # and align_prepared takes the device-resident loop by default
    again = sp.Registration(p).align_prepared(S, prep, initial_guess=T0)
    assert hasattr(again, "log") and np.array_equal(again.T, res.T)


@pytest.mark.parametrize("opt", ["LM", "DOGLEG", "GN"])
@pytest.mark.parametrize("n", [1000, 30000])
def test_annealing_levels_in_one_launch(sp, orc, opt, n):
    """The reference example's registration stage (example_registration.cpp:29-55): Geman-McClure, robust scale annealed
    10 -> 5 -> 2.5 over three levels, each level one align() of at most 10 iterations starting from the previous level's pose."""
    src, scov, tgt, tcov, T_gt = inputs(orc, n, 8.0, seed=99)
    case = dict(opt=opt, reg_type="GICP", loss="GEMAN_MCCLURE", scale=10.0)
    ref = orc.registration_align(oracle_params(case, max_iterations=10, auto_scale=1, auto_scaling_iter=3, init_scale=10.0,
                                               min_scale=2.5), src, scov, tgt, tcov, steps=True)
    S = sp.PointCloudShared(dev(src), covs=dev(scov))
    Tg = sp.PointCloudShared(dev(tgt), covs=dev(tcov))
    prep = sp.PreparedTarget(sp.GridKNN.build(Tg.points), Tg.covs)
    p = sp.RegistrationParams(robust_type="GEMAN_MCCLURE", optimization_method=opt, max_iterations=10)
    scales = [float(s) for s in orc.robust_annealing_scales("GEMAN_MCCLURE", True, 10.0, 10.0, 2.5, 3)]
    assert len(scales) == 3 and abs(scales[1] - 5.0) < 1e-5
    res = sp.Registration(p).align_optimize(S, prep, None, scales)
    assert res is not None
    check_against_oracle(res, ref, case, levels=3)
    assert np.abs(res.T - T_gt).max() < 2e-3
    # level by level through separate calls (what RobustAligner does around a generic aligner): the same pose
    T = None
    for s in scales:
        step = sp.Registration(p).align_optimize(S, prep, T, [s])
        T = step.T
    assert np.abs(T - res.T).max() < 2e-6


def test_optimiser_result_block_and_frozen_error(sp, orc):
    """sp_align_result carries what Registration::compute_error_frozen needs after align(): the pose of the last
    linearisation; the correspondence cache is frozen at it."""
    src, scov, tgt, tcov, _ = inputs(orc, 5000, 8.0)
    S = sp.PointCloudShared(dev(src), covs=dev(scov))
    prep = sp.PreparedTarget(sp.GridKNN.build(dev(tgt)), dev(tcov))
    p = sp.RegistrationParams(optimization_method="LM", max_iterations=6, criteria_rotation=0.0, criteria_translation=0.0)
    reg = sp.Registration(p)
    res = reg.align_optimize(S, prep, None)
    assert res.iterations == 5 and not res.converged and res.linearizations == 6
    L = sp._lib.lib()
    ws, lin = reg._buffers(S.points.device)
    fp = reg._factor_params(p.robust_default_scale)
    Tl = np.ascontiguousarray(res.T_lin.T).reshape(-1)
    Tt = np.ascontiguousarray(res.T.T).reshape(-1)
    sp.check(L.sp_gicp_error_prepared(prep._h, reg._psrc._h, Tl.ctypes.data_as(C.c_void_p), Tt.ctypes.data_as(C.c_void_p), 0,
                                      C.byref(fp), sp._ptr(lin), sp._ptr(ws), ws.numel(), sp._stream()))
    got = reg._read_lin(lin)
    # the last accepted trial evaluated exactly this: error at the final pose over the correspondences of the last linearisation
    last = res.log[-1]
    assert last["accepted"] in (1, 2) and got.inlier == res.inlier and abs(got.error - res.error) <= 1e-6 * abs(res.error)


def test_optimiser_argument_errors_and_switch(sp, orc):
    src, scov, tgt, tcov, _ = inputs(orc, 3000, 8.0)
    S = sp.PointCloudShared(dev(src), covs=dev(scov))
    prep = sp.PreparedTarget(sp.GridKNN.build(dev(tgt)), dev(tcov))
    reg = sp.Registration(sp.RegistrationParams(optimization_method="LM", max_iterations=5))
    with pytest.raises(sp.SpError):
        reg.align_optimize(S, prep, None, [1.0] * 9)  # more levels than SP_OPT_MAX_LEVELS
    ok = reg.align_optimize(S, prep, None)
    assert ok is not None
    # persistent launches switched off for this source: "not available", the caller's per-step loop takes over
    L = sp._lib.lib()
    sp.check(L.sp_gicp_source_set_persistent(reg._psrc._h, 0))
    assert reg.align_optimize(S, prep, None) is None
    res = reg.align_prepared(S, prep)  # falls back to the host-driven loop, same answer
    assert np.abs(res.T - ok.T).max() < 2e-6 and res.iterations == ok.iterations
    sp.check(L.sp_gicp_source_set_persistent(reg._psrc._h, 1))
    assert reg.align_optimize(S, prep, None) is not None


@pytest.mark.parametrize("n", [700, 1900])
@pytest.mark.parametrize("opt,reg_type", [("LM", "GICP"), ("DOGLEG", "POINT_TO_DISTRIBUTION")])
def test_wave_per_point_and_lane_per_point_agree(sp, orc, n, opt, reg_type):
    """Sources of up to 2048 points take one WAVE per point in the linearisation steps (fused_query_wave: certificate, else one
    ball scan by 64 lanes seeded with the previous winner). Every correspondence is the exact nearest neighbour either way, so
    the lane-per-point form of the same launch (internal switch) must search the same number of points, take the same decisions
    and end on the same pose up to the grouping of the sums. A third of the source lies 100 m away from the target: no
    correspondence in any iteration (unseeded scans of the whole bound's ball, negative certificates)."""
    src, scov, tgt, tcov, T_gt = inputs(orc, n, 8.0, seed=7)
    src = src.copy()
    src[::3, 0] += 100.0
    S = sp.PointCloudShared(dev(src), covs=dev(scov))
    prep = sp.PreparedTarget(sp.GridKNN.build(dev(tgt)), dev(tcov), reg_type=reg_type)
    T0 = orc.se3_exp([0.01, -0.005, 0.02, 0.05, -0.04, 0.03])
    case = dict(opt=opt, reg_type=reg_type, loss="GEMAN_MCCLURE", scale=1.0)
    p = sp.RegistrationParams(reg_type=reg_type, robust_type="GEMAN_MCCLURE", robust_default_scale=1.0, optimization_method=opt,
                              max_iterations=15, max_correspondence_distance=0.6)
    scales = [float(v) for v in orc.robust_annealing_scales("GEMAN_MCCLURE", True, 1.0, 2.0, 1.0, 2)]
    assert len(scales) == 2
    out = {}
    for mode in (1, 0):
        reg = sp.Registration(p)
        reg._set_source_option("opt_wave_query", mode)
        out[mode] = reg.align_optimize(S, prep, T0, scales)
        assert out[mode] is not None
    w, l = out[1], out[0]
    ref = orc.registration_align(oracle_params(case, max_iterations=15, max_correspondence_distance=0.6, auto_scale=1,
                                               auto_scaling_iter=2, init_scale=2.0, min_scale=1.0), src, scov, tgt, tcov, init_T=T0,
                                 steps=True)
    check_against_oracle(w, ref, case, levels=2)
    check_against_oracle(l, ref, case, levels=2)
    # (the two forms group the sums differently: a decision the oracle's trace marks as inside rounding may fall either way, and
    # everything after it with it — check_against_oracle holds each form to the oracle up to there; the pose bound holds always)
    assert np.abs(w.T - l.T).max() < 1e-5
    assert n // 2 < w.inlier <= n - (n + 2) // 3 and n // 2 < l.inlier <= n - (n + 2) // 3
    same_path = [(e["trials"], e["accepted"]) for e in w.log] == [(e["trials"], e["accepted"]) for e in l.log]
    if same_path:
        assert w.searched == l.searched and w.inlier == l.inlier and w.iterations == l.iterations and w.converged == l.converged


@pytest.mark.parametrize("n", [900, 1900])
@pytest.mark.parametrize("opt,reg_type,loss", [("LM", "GICP", "GEMAN_MCCLURE"), ("DOGLEG", "GICP", "CAUCHY"),
                                               ("LM", "POINT_TO_DISTRIBUTION", "NONE")])
def test_fused_trial_and_linearisation_steps_change_nothing(sp, orc, n, opt, reg_type, loss):
    """Wave-per-point launches run an LM / dog-leg trial and the linearisation at the trial pose as ONE step (the linearisation
    goes to a second set of cache rows and is adopted when the trial is accepted). Same per-point arithmetic, same grouping of
    the sums, same sequence of state-machine calls as with the switch off: every output bit-identical — pose, H, b, error,
    counters, the per-iteration log — over three annealing levels, with a third of the source outside the target, and the
    source's cache rows afterwards (the frozen error of compute_error_frozen reads them)."""
    src, scov, tgt, tcov, T_gt = inputs(orc, n, 8.0, seed=13)
    src = src.copy()
    src[::3, 0] += 100.0
    S = sp.PointCloudShared(dev(src), covs=dev(scov))
    prep = sp.PreparedTarget(sp.GridKNN.build(dev(tgt)), dev(tcov), reg_type=reg_type)
    T0 = orc.se3_exp([0.02, -0.01, 0.015, 0.08, -0.05, 0.04])
    p = sp.RegistrationParams(reg_type=reg_type, robust_type=loss, robust_default_scale=1.0, optimization_method=opt,
                              max_iterations=8, max_correspondence_distance=0.8)
    scales = [4.0, 2.0, 1.0] if loss != "NONE" else [1.0]
    out, frozen = {}, {}
    L = sp._lib.lib()
    for fuse in (1, 0):
        reg = sp.Registration(p)
        reg._set_source_option("opt_fuse_trials", fuse)
        r = reg.align_optimize(S, prep, T0, scales)
        assert r is not None
        out[fuse] = r
        ws, lin = reg._buffers(S.points.device)
        fp = reg._factor_params(scales[-1])
        Tl = np.ascontiguousarray(r.T_lin.T).reshape(-1)
        Tt = np.ascontiguousarray(r.T.T).reshape(-1)
        sp.check(L.sp_gicp_error_prepared(prep._h, reg._psrc._h, Tl.ctypes.data_as(C.c_void_p), Tt.ctypes.data_as(C.c_void_p), 0,
                                          C.byref(fp), sp._ptr(lin), sp._ptr(ws), ws.numel(), sp._stream()))
        got = reg._read_lin(lin)
        frozen[fuse] = (got.error, got.inlier)
    a, b = out[1], out[0]
    assert np.array_equal(a.T, b.T) and np.array_equal(a.T_lin, b.T_lin) and np.array_equal(a.H, b.H) and np.array_equal(a.b, b.b)
    assert (a.error, a.inlier, a.iterations, a.converged) == (b.error, b.inlier, b.iterations, b.converged)
    assert (a.linearizations, a.trials, a.searched) == (b.linearizations, b.trials, b.searched)
    assert a.log == b.log and len(a.log) >= 3
    assert frozen[1] == frozen[0]
    assert a.trials >= 2


@pytest.mark.parametrize("opt,n_src", [("LM", 1500), ("GN", 1500), ("GN", 6000)])
def test_margin_certificates_prove_their_correspondences(sp, orc, opt, n_src):
    """The wave-per-point launch keeps, beside a point's cache row, where the point was when it was searched and how far it may move
    before its winner can change ((d2 - d1) / 2 from the scan's runner-up): a certificate that holds on real scans, where the
    spacing-based one never does. Every correspondence it lets through must be the one a fresh search returns: with the reuse of
    correspondences switched off (every point searched in every linearisation) all outputs are bit-identical — on the reference's
    bundled scans (box filter + 0.25 m voxels: surfaces, 50 points in the fullest cell), three annealing levels, 12 iterations
    a level — while the searched count drops to a fraction."""
    import os

    from test_gpu_facade import GOLD, read_ply_xyz

    def prep(p):
        p = p[orc.box_filter(p, 0.5, 50.0) == 1]
        p = orc.voxel_downsample(p, 0.25, 1, stable=True)["points"]
        idx, _ = orc.kdtree_knn(orc.kdtree_build(p), p, 10)
        return p, orc.cov_estimate(p, idx)

    s, sc = prep(read_ply_xyz(os.path.join(GOLD, "source.ply")))
    t, tc = prep(read_ply_xyz(os.path.join(GOLD, "target.ply")))
    keep = orc.random_sampling_flags(99, len(s), min(n_src, len(s) - 1)) == 1
    S = sp.PointCloudShared(dev(s[keep]), covs=dev(sc[keep]))
    prep_t = sp.PreparedTarget(sp.GridKNN.build(dev(t)), dev(tc))
    p = sp.RegistrationParams(robust_type="GEMAN_MCCLURE", optimization_method=opt, max_iterations=12, criteria_rotation=0.0,
                              criteria_translation=0.0)
    out = {}
    for reuse in (2, 0):
        reg = sp.Registration(p)
        reg._set_source_option("reuse", reuse)
        reg._set_source_option("opt_wave_query", 2)
        r = reg.align_optimize(S, prep_t, None, [10.0, 5.0, 2.5])
        assert r is not None
        out[reuse] = r
    a, b = out[2], out[0]
    assert np.array_equal(a.T, b.T) and np.array_equal(a.H, b.H) and np.array_equal(a.b, b.b)
    assert (a.error, a.inlier, a.iterations, a.converged, a.linearizations, a.trials) == \
           (b.error, b.inlier, b.iterations, b.converged, b.linearizations, b.trials)
    assert a.log == b.log
    assert b.searched == b.linearizations * S.size()
    assert a.searched < 0.5 * b.searched, (a.searched, b.searched)


def test_wave_per_point_forced_on_a_crowded_target(sp, orc):
    """sp_gicp_source_set_wave_per_point(source, 2): a wave per source point for sources of up to 131072 points, what the facade
    asks for when the target grid's fullest cell holds hundreds of points (a raw LiDAR scan: thousands of returns at the sensor).
    6000 points of which 1500 sit in a 2 cm ball (one crowded cell), sixteen waves a workgroup, several points per wave: the
    same searched counts, decisions and pose as a lane per point, and the oracle's alignment."""
    src, scov, tgt, tcov, T_gt = inputs(orc, 6000, 8.0, seed=21)
    g = orc.rng(5)
    ball = g.uniform_points(1500, 0.02)
    ball[:, :3] += tgt[17, :3]
    tgt = tgt.copy(); src = src.copy()
    tgt[:1500] = ball
    src[:1500] = ball
    src[:1500, :3] += 0.003
    ti, _ = orc.kdtree_knn(orc.kdtree_build(tgt), tgt, 20)
    si, _ = orc.kdtree_knn(orc.kdtree_build(src), src, 20)
    scov, tcov = orc.cov_estimate(src, si), orc.cov_estimate(tgt, ti)
    S = sp.PointCloudShared(dev(src), covs=dev(scov))
    grid = sp.GridKNN.build(dev(tgt))
    assert grid.max_cell_points() >= 1000
    prep = sp.PreparedTarget(grid, dev(tcov))
    T0 = orc.se3_exp([0.004, -0.002, 0.003, 0.01, -0.02, 0.01])
    case = dict(opt="LM", reg_type="GICP", loss="HUBER", scale=0.5)
    p = sp.RegistrationParams(robust_type="HUBER", robust_default_scale=0.5, optimization_method="LM", max_iterations=12)
    out = {}
    for mode in (2, 0):
        reg = sp.Registration(p)
        reg._prepared_source(S.size())
        sp.check(sp._lib.lib().sp_gicp_source_set_wave_per_point(reg._psrc._h, mode))
        out[mode] = reg.align_optimize(S, prep, T0, [0.5])
        assert out[mode] is not None
    w, l = out[2], out[0]
    ref = orc.registration_align(oracle_params(case, max_iterations=12), src, scov, tgt, tcov, init_T=T0, steps=True)
    check_against_oracle(w, ref, case)
    check_against_oracle(l, ref, case)
    assert np.abs(w.T - l.T).max() < 1e-5
    if [(e["trials"], e["accepted"]) for e in w.log] == [(e["trials"], e["accepted"]) for e in l.log]:
        assert w.searched == l.searched and w.inlier == l.inlier and w.iterations == l.iterations and w.converged == l.converged


def test_adaptive_grid_on_a_cloud_of_surfaces(sp, orc):
    """sp_grid_create_adaptive on the reference's bundled scan, voxel-downsampled as its pipelines do before any search
    (points on surfaces: the volume rule packs ~40 of them into an occupied cell): the cell shrinks until the occupied cells
    are as full as a uniform cloud's, the lists stay the exact ones (brute-force oracle), and a uniform cloud is left alone."""
    import os

    from test_gpu_facade import GOLD, read_ply_xyz

    scan = read_ply_xyz(os.path.join(GOLD, "target.ply"))
    scan = scan[orc.box_filter(scan, 0.5, 50.0) == 1]
    pts = orc.voxel_downsample(scan, 0.25, 1, stable=True)["points"]
    P = dev(pts)
    plain = sp.GridKNN.build(P, points_per_cell=0.5)
    adapt = sp.GridKNN.build(P, points_per_cell=0.5, adaptive=True)
    assert adapt.cell_size() < 0.5 * plain.cell_size()
    assert adapt.max_cell_points() <= 16 < plain.max_cell_points()
    q = dev(pts[::3] + np.float32([0.07, -0.05, 0.03, 0.0]))
    oi, od = orc.knn_bruteforce(q.cpu().numpy(), pts, 1)
    for g in (plain, adapt):
        r = sp.KNNResult()
        g.knn_search_async(q, 1, r)
        assert np.array_equal(r.indices.cpu().numpy().reshape(-1), oi.reshape(-1))
        assert np.array_equal(r.distances.cpu().numpy().reshape(-1), od.reshape(-1))
    uni = dev(orc.rng(3).uniform_points(50000, 10.0))
    a, b = sp.GridKNN.build(uni, points_per_cell=0.5), sp.GridKNN.build(uni, points_per_cell=0.5, adaptive=True)
    assert a.cell_size() == b.cell_size()
