"""Pins the oracle's VoxelHashMap restatement (oracle/oracle_voxel_hash_map.hpp) against the reference's own known
answers: every test below restates one TEST of /root/reference/cpp/tests/test_voxel_hash_map.cpp (cited by line) with
the same inputs, expected values and tolerances. The log-Euclidean covariance mean is additionally checked against
scipy's logm / expm in float64 (not against the oracle's own log_spd / exp_spd, which the reference's helper uses)."""
import numpy as np
import pytest
from scipy.linalg import expm, logm


def P(rows):
    a = np.ones((len(rows), 4), np.float32)
    a[:, :3] = np.asarray(rows, np.float32)
    return a


def cov16(xx, xy, xz, yy, yz, zz):
    m = np.zeros((4, 4), np.float32)
    m[:3, :3] = [[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]]
    return m.T.reshape(-1)  # column-major (symmetric anyway)


def sort_xyz(p):
    return p[np.lexsort((p[:, 2], p[:, 1], p[:, 0]))]


def test_constructor_rejects_non_positive_voxel_size(orc):
    # test_voxel_hash_map.cpp:90-96
    for v in (0.0, -0.1):
        with pytest.raises(ValueError):
            orc.voxel_hash_map(v)


def test_aggregates_points_within_same_voxel(orc):
    # :98-147
    m = orc.voxel_hash_map(0.1)
    m.add_point_cloud(P([[0.02, 0.02, 0.0], [0.03, 0.04, 0.0], [0.11, 0.02, 0.0], [0.12, 0.03, 0.0]]))
    out = m.downsampling()
    assert len(out["points"]) == 2
    got = sort_xyz(out["points"])
    assert np.allclose(got[:, :3], [[0.025, 0.03, 0.0], [0.115, 0.025, 0.0]], atol=1e-5) and (got[:, 3] == 1).all()


def test_aggregates_rgb_and_intensity_within_voxel(orc):
    # :149-193
    m = orc.voxel_hash_map(0.5)
    m.add_point_cloud(P([[0, 0, 0], [0.1, 0, 0]]), rgb=[[0.2, 0.4, 0.6, 1.0], [0.6, 0.2, 0.0, 1.0]], intensities=[10.0, 20.0])
    out = m.downsampling()
    assert len(out["points"]) == 1 and out["rgb"] is not None and out["intensities"] is not None and out["covs"] is None
    assert np.allclose(out["points"][0, :3], [0.05, 0, 0], atol=1e-5)
    assert np.allclose(out["rgb"][0], [0.4, 0.3, 0.3, 1.0], atol=1e-5)
    assert abs(out["intensities"][0] - 15.0) < 1e-5


def log_euclidean_mean_f64(covs, R=np.eye(3)):
    acc = np.zeros((3, 3))
    for c in covs:
        acc += np.real(logm(np.asarray(c, np.float64)))
    return R @ np.real(expm(acc / len(covs))) @ R.T


def test_aggregates_covariances_within_voxel(orc):
    # :195-250 — expected value: mean of log(C) mapped back by exp (ComputeExpectCovariance, :74-86)
    c = [(1.0, 0.2, 0.3, 2.0, 0.4, 3.0), (3.0, 0.6, 0.9, 4.0, 0.8, 5.0)]
    m = orc.voxel_hash_map(0.5)
    m.add_point_cloud(P([[0, 0, 0], [0.1, 0, 0]]), covs=[cov16(*x) for x in c],
                      rgb=[[0.2, 0.4, 0.6, 1.0], [0.6, 0.2, 0.0, 1.0]], intensities=[10.0, 20.0])
    out = m.downsampling()
    assert len(out["points"]) == 1 and out["covs"] is not None
    got = out["covs"][0].reshape(4, 4).T
    mats = [np.array([[a, b, cc], [b, d, e], [cc, e, f]]) for a, b, cc, d, e, f in c]
    assert np.allclose(got[:3, :3], log_euclidean_mean_f64(mats), atol=1e-5)
    assert np.abs(got[3]).max() == 0 and np.abs(got[:, 3]).max() == 0
    assert np.allclose(out["rgb"][0], [0.4, 0.3, 0.3, 1.0], atol=1e-5) and abs(out["intensities"][0] - 15.0) < 1e-5


def test_rotates_covariances_into_map_frame(orc):
    # :252-294: 90 degrees about z, translation (1, 0, 0); tolerance 1e-4 as the reference
    c = [(1.0, 0.0, 0.0, 4.0, 0.0, 9.0), (9.0, 0.0, 0.0, 16.0, 0.0, 25.0)]
    pose = np.eye(4, dtype=np.float32)
    th = np.float32(np.pi / 2)
    pose[:3, :3] = [[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]]
    pose[:3, 3] = [1.0, 0.0, 0.0]
    m = orc.voxel_hash_map(0.5)
    m.add_point_cloud(P([[0, 0, 0], [0.1, 0, 0]]), pose=pose, covs=[cov16(*x) for x in c])
    out = m.downsampling()
    assert len(out["points"]) == 1
    mats = [np.diag([a, d, f]) for a, _, _, d, _, f in c]
    expect = log_euclidean_mean_f64(mats, pose[:3, :3].astype(np.float64))
    assert np.allclose(out["covs"][0].reshape(4, 4).T[:3, :3], expect, atol=1e-4)


def test_covariance_output_disabled_without_input_covariances(orc):
    # :296-313
    m = orc.voxel_hash_map(0.5)
    m.add_point_cloud(P([[0, 0, 0], [0.1, 0, 0]]))
    out = m.downsampling()
    assert len(out["points"]) == 1 and out["covs"] is None


def test_minimum_point_threshold_per_voxel(orc):
    # :315-344
    m = orc.voxel_hash_map(0.2)
    m.set("min_num_point", 2)
    m.add_point_cloud(P([[0.01, 0.01, 0], [0.02, 0.01, 0], [0.30, 0.30, 0]]))
    out = m.downsampling()
    assert len(out["points"]) == 1 and np.allclose(out["points"][0, :3], [0.015, 0.01, 0.0], atol=1e-5)


def test_downsampling_respects_bounding_box(orc):
    # :346-378
    m = orc.voxel_hash_map(0.2)
    m.add_point_cloud(P([[1.05, 0, 0], [1.12, 0, 0], [1.35, 0, 0], [1.00, 0.25, 0]]))
    out = m.downsampling(center=(1.0, 0.0, 0.0), distance=0.2)
    assert len(out["points"]) == 1 and np.allclose(out["points"][0, :3], [1.085, 0, 0], atol=1e-5)


def test_overlap_ratio(orc):
    # :380-419
    m = orc.voxel_hash_map(0.5)
    map_pts = P([[0.1, 0.1, 0.0], [1.1, 0.0, 0.0]])
    m.add_point_cloud(map_pts)
    q = P([[-0.9, 0.1, 0.0], [0.1, 0.0, 0.0], [1.0, 0.0, 0.0]])
    pose = np.eye(4, dtype=np.float32)
    pose[0, 3] = 1.0
    assert abs(m.overlap_ratio(q, pose) - 2.0 / 3.0) < 1e-5
    m.set("min_num_point", 2)
    assert abs(m.overlap_ratio(q, pose)) < 1e-5
    m.add_point_cloud(map_pts)
    assert abs(m.overlap_ratio(q, pose) - 2.0 / 3.0) < 1e-5


def test_counts_voxels_for_large_batch(orc):
    # :421-453
    m = orc.voxel_hash_map(1.0)
    pts = P([[i * 2.0 + 0.5, 0.5, 0.5] for i in range(100)])
    m.add_point_cloud(pts)
    out = m.downsampling(distance=1000.0)
    assert len(out["points"]) == 100 and m.info("voxel_num") == 100
    assert np.allclose(np.sort(out["points"][:, 0]), np.arange(100) * 2.0 + 0.5, atol=1e-5)


def test_preserves_data_after_rehash(orc):
    # :455-502: threshold 0 forces a rehash on the second batch
    m = orc.voxel_hash_map(1.0)
    m.set("rehash_threshold", 0.0)
    m.add_point_cloud(P([[0.5, 0.5, 0.5], [10.5, 0.5, 0.5], [20.5, 0.5, 0.5]]))
    assert m.info("capacity") == 30029
    m.add_point_cloud(P([[30.5, 0.5, 0.5], [40.5, 0.5, 0.5]]))
    assert m.info("capacity") == 60013
    out = m.downsampling()
    assert len(out["points"]) == 5
    assert np.allclose(np.sort(out["points"][:, 0]), [0.5, 10.5, 20.5, 30.5, 40.5], atol=1e-5)


def test_removes_stale_voxels_after_configured_cycles(orc):
    # :504-540
    m = orc.voxel_hash_map(0.1)
    m.set("max_staleness", 1)
    m.set("remove_old_data_cycle", 1)
    m.add_point_cloud(P([[0, 0, 0]]))
    assert len(m.downsampling()["points"]) == 1
    m.add_point_cloud(P([[1.0, 0, 0]]))
    assert len(m.downsampling()["points"]) == 2
    m.add_point_cloud(P(np.zeros((0, 3))))
    out = m.downsampling()
    assert len(out["points"]) == 1 and np.allclose(out["points"][0, :3], [1.0, 0.0, 0.0], atol=1e-5)


def test_log_exp_spd_round_trip_and_float64(orc):
    rs = np.random.RandomState(5)
    for _ in range(50):
        A = rs.normal(size=(3, 3))
        C = (A @ A.T + 0.05 * np.eye(3)).astype(np.float32)
        L = orc.log_spd3(C)
        assert np.allclose(L, np.real(logm(C.astype(np.float64))), atol=2e-4 * max(1.0, np.abs(L).max()))
        assert np.allclose(orc.exp_spd3(L), C, rtol=2e-4, atol=2e-5 * np.abs(C).max())
