"""VoxelHashMap on the GPU (sp_vhm_*, algorithms/mapping/voxel_hash_map.hpp) against (a) the reference's own known
answers (cpp/tests/test_voxel_hash_map.cpp, the same cases tests/test_oracle_voxel_hash_map.py pins the oracle with) and
(b) the oracle on large seeded clouds: the SET of voxels (keys) and the counts are exact; point / colour / intensity sums
are accumulated with relaxed float atomics (as in the reference: order unspecified) and compared at 2e-6 relative to the
voxel's coordinate scale; covariances go through device logf / expf and the analytic fp32 eigen-decomposition and are
held to that decomposition's own error profile (1e-4 at the 99th percentile)."""
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


def P(rows):
    a = np.ones((len(rows), 4), np.float32)
    a[:, :3] = np.asarray(rows, np.float32).reshape(-1, 3)
    return a


def cloud(sp, pts, covs=None, rgb=None, intensities=None):
    d = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()  # noqa: E731
    return sp.PointCloudShared(d(pts), covs=d(covs), rgb=d(rgb), intensities=d(intensities))


def cov16(xx, xy, xz, yy, yz, zz):
    m = np.zeros((4, 4), np.float32)
    m[:3, :3] = [[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]]
    return m.T.reshape(-1)


def test_reference_known_answers(sp, orc):
    from scipy.linalg import expm, logm

    with pytest.raises(sp.SpError) as e:  # :90-96
        sp.VoxelHashMap(0.0)
    assert e.value.code == 1
    # :98-147
    m = sp.VoxelHashMap(0.1)
    m.add_point_cloud(cloud(sp, P([[0.02, 0.02, 0.0], [0.03, 0.04, 0.0], [0.11, 0.02, 0.0], [0.12, 0.03, 0.0]])))
    out = m.downsampling().points.cpu().numpy()
    out = out[np.lexsort((out[:, 2], out[:, 1], out[:, 0]))]
    assert out.shape == (2, 4) and np.allclose(out[:, :3], [[0.025, 0.03, 0], [0.115, 0.025, 0]], atol=1e-5)
    # :195-250 covariance (log-Euclidean mean), rgb, intensity
    c = [(1.0, 0.2, 0.3, 2.0, 0.4, 3.0), (3.0, 0.6, 0.9, 4.0, 0.8, 5.0)]
    m = sp.VoxelHashMap(0.5)
    m.add_point_cloud(cloud(sp, P([[0, 0, 0], [0.1, 0, 0]]), covs=np.stack([cov16(*x) for x in c]),
                            rgb=[[0.2, 0.4, 0.6, 1.0], [0.6, 0.2, 0.0, 1.0]], intensities=[10.0, 20.0]))
    r = m.downsampling()
    assert r.size() == 1 and r.has_cov() and r.has_rgb() and r.has_intensity()
    mats = [np.array([[a, b, cc], [b, d, e], [cc, e, f]], np.float64) for a, b, cc, d, e, f in c]
    expect = np.real(expm(sum(np.real(logm(x)) for x in mats) / 2))
    got = r.covs.cpu().numpy()[0].reshape(4, 4).T
    assert np.allclose(got[:3, :3], expect, atol=1e-5) and np.abs(got[3]).max() == 0 and np.abs(got[:, 3]).max() == 0
    assert np.allclose(r.rgb.cpu().numpy()[0], [0.4, 0.3, 0.3, 1.0], atol=1e-5)
    assert abs(float(r.intensities[0]) - 15.0) < 1e-5 and np.allclose(r.points.cpu().numpy()[0, :3], [0.05, 0, 0], atol=1e-5)
    # :252-294 rotation into the map frame
    pose = np.eye(4, dtype=np.float32)
    th = np.float32(np.pi / 2)
    pose[:3, :3] = [[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]]
    pose[:3, 3] = [1.0, 0.0, 0.0]
    c = [(1.0, 0.0, 0.0, 4.0, 0.0, 9.0), (9.0, 0.0, 0.0, 16.0, 0.0, 25.0)]
    m = sp.VoxelHashMap(0.5)
    m.add_point_cloud(cloud(sp, P([[0, 0, 0], [0.1, 0, 0]]), covs=np.stack([cov16(*x) for x in c])), pose)
    r = m.downsampling()
    R = pose[:3, :3].astype(np.float64)
    expect = R @ np.real(expm(sum(np.real(logm(np.diag([a, d, f]).astype(np.float64))) for a, _, _, d, _, f in c) / 2)) @ R.T
    assert r.size() == 1 and np.allclose(r.covs.cpu().numpy()[0].reshape(4, 4).T[:3, :3], expect, atol=1e-4)
    # :296-313, :315-344, :346-378
    m = sp.VoxelHashMap(0.5)
    m.add_point_cloud(cloud(sp, P([[0, 0, 0], [0.1, 0, 0]])))
    assert m.downsampling().size() == 1 and not m.downsampling().has_cov()
    m = sp.VoxelHashMap(0.2)
    m.set_min_num_point(2)
    m.add_point_cloud(cloud(sp, P([[0.01, 0.01, 0], [0.02, 0.01, 0], [0.30, 0.30, 0]])))
    r = m.downsampling()
    assert r.size() == 1 and np.allclose(r.points.cpu().numpy()[0, :3], [0.015, 0.01, 0], atol=1e-5)
    m = sp.VoxelHashMap(0.2)
    m.add_point_cloud(cloud(sp, P([[1.05, 0, 0], [1.12, 0, 0], [1.35, 0, 0], [1.00, 0.25, 0]])))
    r = m.downsampling(center=(1.0, 0.0, 0.0), distance=0.2)
    assert r.size() == 1 and np.allclose(r.points.cpu().numpy()[0, :3], [1.085, 0, 0], atol=1e-5)
    # :380-419 overlap ratio
    m = sp.VoxelHashMap(0.5)
    mp = cloud(sp, P([[0.1, 0.1, 0.0], [1.1, 0.0, 0.0]]))
    m.add_point_cloud(mp)
    q = cloud(sp, P([[-0.9, 0.1, 0.0], [0.1, 0.0, 0.0], [1.0, 0.0, 0.0]]))
    sensor = np.eye(4, dtype=np.float32)
    sensor[0, 3] = 1.0
    assert abs(m.compute_overlap_ratio(q, sensor) - 2.0 / 3.0) < 1e-5
    m.set_min_num_point(2)
    assert abs(m.compute_overlap_ratio(q, sensor)) < 1e-5
    m.add_point_cloud(mp)
    assert abs(m.compute_overlap_ratio(q, sensor) - 2.0 / 3.0) < 1e-5
    # :421-453, :455-502
    m = sp.VoxelHashMap(1.0)
    m.add_point_cloud(cloud(sp, P([[i * 2.0 + 0.5, 0.5, 0.5] for i in range(100)])))
    r = m.downsampling(distance=1000.0)
    assert r.size() == 100 and np.allclose(np.sort(r.points.cpu().numpy()[:, 0]), np.arange(100) * 2.0 + 0.5, atol=1e-5)
    m = sp.VoxelHashMap(1.0)
    m.set_rehash_threshold(0.0)
    m.add_point_cloud(cloud(sp, P([[0.5, 0.5, 0.5], [10.5, 0.5, 0.5], [20.5, 0.5, 0.5]])))
    assert m.info("capacity") == 30029
    m.add_point_cloud(cloud(sp, P([[30.5, 0.5, 0.5], [40.5, 0.5, 0.5]])))
    assert m.info("capacity") == 60013
    r = m.downsampling()
    assert r.size() == 5 and np.allclose(np.sort(r.points.cpu().numpy()[:, 0]), [0.5, 10.5, 20.5, 30.5, 40.5], atol=1e-5)
    # :504-540 staleness
    m = sp.VoxelHashMap(0.1)
    m.set_max_staleness(1)
    m.set_remove_old_data_cycle(1)
    m.add_point_cloud(cloud(sp, P([[0, 0, 0]])))
    assert m.downsampling().size() == 1
    m.add_point_cloud(cloud(sp, P([[1.0, 0, 0]])))
    assert m.downsampling().size() == 2
    m.add_point_cloud(sp.PointCloudShared())
    r = m.downsampling()
    assert r.size() == 1 and np.allclose(r.points.cpu().numpy()[0, :3], [1.0, 0, 0], atol=1e-5)


def by_key(keys, *arrays):
    o = np.argsort(keys, kind="stable")
    return [keys[o]] + [None if a is None else a[o] for a in arrays]


@pytest.mark.parametrize("attrs", ["plain", "all"])
def test_frames_match_oracle(sp, orc, attrs):
    """Frames of 3 k ... 150 k points from moving sensor poses (the table is rehashed from 30 029 to 480 013 slots on the way,
    stale voxels removed, NaN points skipped), voxel 0.25: same voxel set, counts, means, attributes as the oracle. The
    frames are sized so that the table never overflows its 100 probes: the reference grows one capacity step per call, and
    WHICH entries an overflowing table drops depends on the insertion order (unspecified there, sequential in the oracle)."""
    rs = np.random.RandomState(7)
    gm = sp.VoxelHashMap(0.25)
    om = orc.voxel_hash_map(0.25)
    removal = attrs == "plain"  # stale voxels are removed in one variant only, see the overlap check below
    for m in (gm, om):
        if removal:
            (m.set_max_staleness if m is gm else lambda v: m.set("max_staleness", v))(2)
            (m.set_remove_old_data_cycle if m is gm else lambda v: m.set("remove_old_data_cycle", v))(2)
        (m.set_rehash_threshold if m is gm else lambda v: m.set("rehash_threshold", v))(0.05)
    for frame, n in enumerate((3_000, 6_000, 20_000, 60_000, 150_000)):
        pts = np.ones((n, 4), np.float32)
        pts[:, :3] = rs.uniform(-12, 12, (n, 3)).astype(np.float32)
        pts[::500, 1] = np.nan  # invalid key: skipped (voxel_constants.hpp:41-43)
        covs = rgb = inten = None
        if attrs == "all":
            A = rs.normal(0, 0.3, (n, 3, 3)).astype(np.float32)
            C3 = A @ A.transpose(0, 2, 1) + 0.02 * np.eye(3, dtype=np.float32)
            covs = np.zeros((n, 4, 4), np.float32)
            covs[:, :3, :3] = C3
            covs = covs.reshape(n, 16)
            rgb = rs.uniform(0, 1, (n, 4)).astype(np.float32)
            inten = rs.uniform(0, 100, n).astype(np.float32)
        pose = orc.se3_exp([0.02 * frame, -0.01 * frame, 0.03 * frame, 1.5 * frame, -0.5 * frame, 0.1 * frame])
        gm.add_point_cloud(cloud(sp, pts, covs, rgb, inten), pose)
        om.add_point_cloud(pts, pose, covs=covs, rgb=rgb, intensities=inten)
        assert gm.info("voxel_num") == om.info("voxel_num") and gm.info("capacity") == om.info("capacity")
    assert gm.info("capacity") == 480013 and gm.info("staleness_counter") == 5
    for center, dist in (((0.0, 0.0, 0.0), 100.0), ((3.0, -2.0, 1.0), 6.0)):
        g = gm.downsampling(center, dist)
        o = om.downsampling(center, dist)
        gk, gp, gc, gr, gi = by_key(g.voxel_keys.cpu().numpy().view(np.uint64), g.points.cpu().numpy(),
                                    None if g.covs is None else g.covs.cpu().numpy(),
                                    None if g.rgb is None else g.rgb.cpu().numpy(),
                                    None if g.intensities is None else g.intensities.cpu().numpy())
        ok, op, oc, orgb, oi = by_key(o["keys"], o["points"], o["covs"], o["rgb"], o["intensities"])
        # voxels whose centroid sits on the box face to rounding may differ (sum order): allow a handful, compare the rest
        common, ig, io = np.intersect1d(gk, ok, return_indices=True)
        assert len(common) >= max(len(gk), len(ok)) - 3 and len(common) > 1000
        assert np.abs(gp[ig] - op[io]).max() <= 2e-6 * 20.0
        if attrs == "all":
            assert np.abs(gr[ig] - orgb[io]).max() <= 2e-6 and np.abs(gi[ig] - oi[io]).max() <= 2e-4
            # exp(mean log C) through the reference's analytic fp32 eigen-decomposition: its own round trip exp(log(C)) is off
            # by 1e-6 (median) ... 1.3e-4 (99.9th percentile) ... 5e-3 (worst of 20 000 random covariances, near-equal
            # eigenvalues) relative to |C| on the CPU already, so device-vs-oracle is held to the same profile
            rel = np.abs(gc[ig] - oc[io]).max(axis=1) / np.abs(oc[io]).max(axis=1)
            assert np.percentile(rel, 99) <= 1e-4 and np.percentile(rel, 99.9) <= 1e-3 and rel.max() <= 3e-2, \
                (np.percentile(rel, [50, 99, 99.9]), rel.max())
        else:
            assert gc is None and gr is None and gi is None
    # counts: exact (through the overlap ratio with a threshold that only multi-point voxels pass). Not after a removal:
    # the reference clears a stale slot to "empty" (no tombstone, :806-840), which cuts the probe chains running through it,
    # so which later keys a lookup still reaches depends on the slot layout, i.e. on the (unspecified) insertion order.
    q = np.ones((50_000, 4), np.float32)
    q[:, :3] = rs.uniform(-12, 12, (50_000, 3)).astype(np.float32)
    for thr in (1, 2, 3):
        gm.set_min_num_point(thr)
        om.set("min_num_point", thr)
        if not removal:
            assert gm.compute_overlap_ratio(cloud(sp, q)) == pytest.approx(om.overlap_ratio(q), abs=1e-7)
        else:
            assert gm.compute_overlap_ratio(cloud(sp, q)) == pytest.approx(om.overlap_ratio(q), abs=2e-3)
    gm.clear()
    assert gm.info("voxel_num") == 0 and gm.info("capacity") == 30029 and gm.downsampling().size() == 0
