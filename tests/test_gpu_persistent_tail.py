"""sp_gicp_align_fused with convergence criteria runs the tail of an alignment as ONE launch that loops on the device
(gicp_align_persistent_kernel, csrc/registration.hip): every iteration behind the first few waits for the other workgroups'
partial rows through an arrival counter instead of a kernel boundary. The reference's loop is Registration::align
(algorithms/registration/registration.hpp:229-276: iterate, is_converged -> break).

Whatever the split between per-iteration launches and the device-side tail, the alignment must give the same bits: pose,
linear system, delta, iteration count, neighbours — and the state block that compute_error_frozen / the linearisation-pose
query read afterwards."""
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


@pytest.fixture(scope="module")
def clouds(sp):
    from sycl_points_amd.synthetic import gicp_pair

    out = {}
    for n in (60_000, 300_000):  # 59 workgroups (fewer than CUs) and a full grid of 256
        src, tgt, T_gt = gicp_pair(n, 10.0 * (n / 1e6) ** (1.0 / 3.0))
        Tg = sp.PointCloudShared(dev(tgt))
        Tg.covs = sp.GridKNN.build(Tg.points, points_per_cell=6.0).self_knn(20, want_knn=False, want_covs=True)[1]
        S_all = dev(src)
        S_all = S_all[sp.GridKNN.build(S_all, points_per_cell=1.0).order()].contiguous()
        covs = sp.GridKNN.build(S_all, points_per_cell=6.0).self_knn(20, want_knn=False, want_covs=True)[1]
        S = sp.PointCloudShared(S_all, covs=covs)
        prep = sp.PreparedTarget(sp.GridKNN.build(Tg.points, points_per_cell=0.5), Tg.covs)
        out[n] = (S, prep, T_gt)
    return out


def run(sp, S, prep, crit, max_iterations, opts, loss="NONE"):
    p = sp.RegistrationParams(criteria_translation=crit, criteria_rotation=crit, max_iterations=max_iterations,
                              robust_type=loss)
    reg = sp.Registration(p)
    for k, v in opts.items():
        reg._set_source_option(k, v)
    T_dev, lin, delta = reg.align_fused_loop(S, prep, sort_by_cell="presorted", write_neighbors=True)
    torch.cuda.synchronize()
    # the pose of the last linearisation, as compute_error_frozen reads it from the state block (sp_gicp_align_linearization_pose)
    ws, _ = reg._buffers(T_dev.device)
    TL = torch.zeros(16, dtype=torch.float32, device=T_dev.device)
    sp.check(sp._lib.lib().sp_gicp_align_linearization_pose(sp._ptr(ws), max_iterations - 1, sp._ptr(TL), sp._stream()))
    torch.cuda.synchronize()
    T_lin = TL.cpu().numpy().copy()
    return dict(T=T_dev.cpu().numpy().copy(), lin=lin.cpu().numpy().copy(), delta=delta.cpu().numpy().copy(),
                iters=int(reg._iters_dev[0]), idx=reg.neighbors.indices.cpu().numpy().ravel().copy(),
                d2=reg.neighbors.distances.cpu().numpy().ravel().copy(), T_lin=T_lin)


def same(a, b):
    for key in ("T", "lin", "delta", "idx", "d2"):
        assert np.array_equal(a[key], b[key]), key
    assert a["iters"] == b["iters"]
    if a["T_lin"] is not None:
        assert np.array_equal(a["T_lin"], b["T_lin"])


@pytest.mark.parametrize("n", [60_000, 300_000])
def test_tail_on_the_device_equals_a_launch_per_iteration(sp, clouds, n):
    S, prep, T_gt = clouds[n]
    for crit, max_it in ((1e-3, 20), (1e-7, 20), (1e-3, 3), (1e-3, 2), (1e-3, 1), (1e-7, 5), (1e-7, 4)):
        ref = run(sp, S, prep, crit, max_it, {"persistent": 0})
        assert 1 <= ref["iters"] <= max_it
        if max_it == 20:
            assert np.abs(ref["T"].reshape(4, 4).T - T_gt).max() < 2e-4
        for frm in (0, 1, 2, 3, 4, 6):  # the whole alignment on the device ... the tail behind six launches
            same(ref, run(sp, S, prep, crit, max_it, {"persistent_from": frm}))
    # the default split, twice in a row on the same objects (the arrival counters start from zero every time)
    a = run(sp, S, prep, 1e-3, 20, {})
    b = run(sp, S, prep, 1e-3, 20, {})
    same(a, b)
    same(a, run(sp, S, prep, 1e-3, 20, {"persistent": 0}))


def test_tail_with_a_robust_kernel(sp, clouds):
    """Another instantiation of the kernel (robust weights) through both forms."""
    S, prep, _ = clouds[60_000]
    ref = run(sp, S, prep, 1e-5, 12, {"persistent": 0}, loss="HUBER")
    same(ref, run(sp, S, prep, 1e-5, 12, {"persistent_from": 0}, loss="HUBER"))
    same(ref, run(sp, S, prep, 1e-5, 12, {"persistent_from": 2}, loss="HUBER"))


def test_fixed_iteration_count_never_takes_the_tail(sp, clouds):
    """Criteria 0 cannot be met: every iteration is a launch of its own (the benchmarked configuration), and switching the
    device-side tail off changes nothing."""
    S, prep, _ = clouds[60_000]
    same(run(sp, S, prep, 0.0, 8, {}), run(sp, S, prep, 0.0, 8, {"persistent": 0}))
