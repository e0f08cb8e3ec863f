"""Edge cases of the structure builds rewritten in round 2 (hand-written radix sort, gap-fill cell table, block-offset voxel
compaction): tiny clouds, exactly one tile, clustered clouds with huge runs of empty cells, non-finite points, 8 M points.
GridKNN must stay bit-identical to brute force; voxel downsampling to the oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sp():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device (no CPU fallback exists)")
    import sycl_points_amd.api as api

    return api


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 2047, 2048, 2049, 4097])
def test_grid_on_tiny_and_tile_sized_clouds(sp, orc, n):
    g = orc.rng(1000 + n)
    tgt = g.uniform_points(n, 5.0)
    qry = g.uniform_points(257, 6.0)
    for ppc in (0.5, 6.0):
        grid = sp.GridKNN.build(dev(tgt), points_per_cell=ppc)
        k = min(3, n)
        r = grid.knn_search(dev(qry), k)
        oi, od = orc.knn_bruteforce(qry, tgt, k)
        assert np.array_equal(r.indices.cpu().numpy(), oi) and np.array_equal(r.distances.cpu().numpy(), od)


def test_grid_on_two_distant_clusters_and_a_flat_cloud(sp, orc):
    # almost every cell of the bounding box is empty: the cell table is one long gap between the clusters
    g = orc.rng(77)
    a, b = g.uniform_points(3000, 1.0), g.uniform_points(3000, 1.0)
    b[:, :3] += np.float32([400.0, -250.0, 90.0])
    tgt = np.concatenate([a, b])
    qry = np.concatenate([g.uniform_points(200, 1.5), g.uniform_points(200, 1.5) + np.float32([400.0, -250.0, 90.0, 0.0]),
                          np.float32([[200.0, -125.0, 45.0, 1.0]])])  # one query in the void between them
    grid = sp.GridKNN.build(dev(tgt), points_per_cell=0.5)
    r = grid.knn_search(dev(qry), 4)
    oi, od = orc.knn_bruteforce(qry, tgt, 4)
    assert np.array_equal(r.indices.cpu().numpy(), oi) and np.array_equal(r.distances.cpu().numpy(), od)
    flat = g.uniform_points(5000, 10.0)
    flat[:, 2] = 0.25  # a plane: the z extent is zero
    grid = sp.GridKNN.build(dev(flat), points_per_cell=2.0)
    r = grid.knn_search(dev(flat[:300]), 5)
    oi, od = orc.knn_bruteforce(flat[:300], flat, 5)
    assert np.array_equal(r.indices.cpu().numpy(), oi) and np.array_equal(r.distances.cpu().numpy(), od)


def test_voxel_downsampling_edge_sizes(sp, orc):
    g = orc.rng(5)
    for n, size in ((1, 0.1), (255, 0.1), (256, 0.1), (257, 5.0), (2049, 100.0), (50_000, 0.05)):
        pts = g.uniform_points(n, 3.0)
        vg = sp.VoxelGrid(size)
        for _ in range(2):  # the second call sorts keys compressed to the first call's key box
            out = vg.downsampling(dev(pts))
            ref = orc.voxel_downsample(pts, size)["points"]
            assert out.size() == len(ref) and np.array_equal(out.points.cpu().numpy(), ref)
    # every point in one voxel (one run spanning many workgroups of the aggregation)
    pts = g.uniform_points(10_000, 0.01)
    vg = sp.VoxelGrid(10.0)
    for _ in range(2):
        out = vg.downsampling(dev(pts))
        ref = orc.voxel_downsample(pts, 10.0)["points"]
        assert out.size() == len(ref) and np.array_equal(out.points.cpu().numpy(), ref)


def test_structures_at_8m_points(sp, orc):
    # config-5 size on one GPU: 3907 sort tiles, 16 M cells; sampled against brute force on the host
    g = orc.rng(8)
    tgt = g.uniform_points(8_000_000, 20.0)
    grid = sp.GridKNN.build(dev(tgt), points_per_cell=0.5)
    qry = g.uniform_points(64, 20.0)
    r = grid.knn_search(dev(qry), 2)
    oi, od = orc.knn_bruteforce(qry, tgt, 2)
    assert np.array_equal(r.indices.cpu().numpy(), oi) and np.array_equal(r.distances.cpu().numpy(), od)
    vg = sp.VoxelGrid(0.1)
    vg.downsampling(dev(tgt))  # (remembers the key box: the next call takes the compressed-key path)
    out, keys = vg.downsampling(dev(tgt), return_keys=True)
    k = keys.cpu().numpy()
    assert (np.diff(k) > 0).all()  # ascending, no voxel twice
    assert np.array_equal(np.unique(orc.voxel_keys(tgt, 0.1).astype(np.int64)), k)


@pytest.mark.parametrize("k", [11, 16, 20])
def test_self_knn_lists_with_duplicates_and_sparse_clouds(sp, orc, k):
    # the wave-cooperative self-kNN (lane-minimum-first sort on DPP, stage directions as scalar lane masks): exact ties
    # (duplicated points), a dense-ish cloud, and a cloud so sparse that most queries need the ring expansion / to-do pass
    g = orc.rng(31 + k)
    for n, r in ((20000, 5.0), (3000, 40.0)):
        pts = g.uniform_points(n, r)
        for cloud in (pts, np.concatenate([pts, pts[:500]])):
            grid = sp.GridKNN.build(dev(cloud), points_per_cell=6.0)
            res, covs, _ = grid.self_knn(k, want_knn=True, want_covs=True)
            oi, od = orc.knn_bruteforce(cloud, cloud, k)
            assert np.array_equal(res.indices.cpu().numpy(), oi) and np.array_equal(res.distances.cpu().numpy(), od)
            assert np.array_equal(covs.cpu().numpy().reshape(-1, 16), np.asarray(orc.cov_estimate(cloud, oi), np.float32).reshape(-1, 16))


@pytest.mark.parametrize("k", [8, 9, 13, 20])
def test_self_knn_select_kernel_equals_the_wave_kernel_bit_for_bit(sp, orc, k):
    # grid_self_knn_select_kernel (lane per query: 7-bit keys, threshold, rank; its unproven queries go to the wave-per-query
    # list kernel) against the wave-cooperative kernel (internal switch self_knn_mode = 2): lists, distances, covariances and
    # normals, on a uniform cloud, on a cloud whose density varies by four orders of magnitude (most cells overflow the
    # candidate cap: everything goes through the list kernel), on a tiny cloud, on one point repeated (all distances tie: no
    # threshold separates k .. 32 of them), with non-finite points, and over a range of positions.
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from test_gpu_bvh import nonuniform_cloud

    g = orc.rng(77 + k)
    same = np.tile(np.float32([[1.0, 2.0, 3.0, 1.0]]), (300, 1))
    holes = g.uniform_points(5000, 6.0)
    holes[::13, 1] = np.nan
    holes[7::31, 0] = np.inf
    for cloud, ppc in ((g.uniform_points(30000, 6.0), 6.0), (nonuniform_cloud(20000), 6.0), (g.uniform_points(37, 2.0), 6.0),
                       (np.concatenate([same, g.uniform_points(2000, 4.0)]), 6.0), (holes, 6.0), (g.uniform_points(30000, 6.0), 2.0)):
        outs = []
        for mode in (0, 2):
            grid = sp.GridKNN.build(dev(cloud), points_per_cell=ppc)
            grid._set_option("self_knn_mode", mode)
            res, covs, nrm = grid.self_knn(k, want_knn=True, want_covs=True, want_normals=True)
            outs.append((res.indices.cpu().numpy(), res.distances.cpu().numpy(), covs.cpu().numpy(), nrm.cpu().numpy()))
        finite = np.isfinite(cloud[:, :3]).all(1)  # (rows of non-finite points are not defined)
        for a, b in zip(outs[0], outs[1]):
            assert np.array_equal(a[finite], b[finite], equal_nan=True)
    cloud = g.uniform_points(20000, 6.0)
    grid = sp.GridKNN.build(dev(cloud), points_per_cell=6.0)
    whole = grid.self_knn(k, want_knn=False, want_covs=True)[1].cpu().numpy()
    oi, _ = orc.knn_bruteforce(cloud, cloud, k)
    assert np.array_equal(whole.reshape(-1, 16), np.asarray(orc.cov_estimate(cloud, oi), np.float32).reshape(-1, 16))


@pytest.mark.parametrize("k", [11, 20])
def test_grid_external_queries_select_path(sp, orc, k):
    # KNNBase::knn_search_async on a GridKNN with k > 10 runs grid_search_select_kernel (lane per query inside the 27 cells)
    # and finishes the unproven queries a wave each: queries far outside the cloud and in empty neighbourhoods (no candidate
    # at all in their 27 cells), non-finite queries, a sparse grid (nearly everything unproven), dense cells (candidate cap),
    # duplicated targets (ties), a transform, fewer targets than k.
    g = orc.rng(900 + k)
    tgt = g.uniform_points(15000, 5.0)
    tgt_dup = np.concatenate([tgt, tgt[:300]])
    qry = np.concatenate([g.uniform_points(3000, 5.0), g.uniform_points(500, 30.0), np.float32([[1e4, -1e4, 0.0, 1.0]])])
    qry[7, 0] = np.nan
    qry[11, 2] = np.inf
    T = orc.se3_exp([0.01, 0.02, -0.03, 0.3, -0.2, 0.1])
    for cloud, ppc in ((tgt, 6.0), (tgt, 0.3), (tgt, 40.0), (tgt_dup, 6.0), (tgt[:k - 3], 6.0)):
        grid = sp.GridKNN.build(dev(cloud), points_per_cell=ppc)
        r = grid.knn_search(dev(qry), k, T)
        oi, od = orc.knn_bruteforce(orc.transform_points(qry, T), cloud, k)
        assert np.array_equal(r.indices.cpu().numpy(), oi)
        assert np.array_equal(r.distances.cpu().numpy(), od)


def test_grid_with_non_finite_points(sp, orc):
    # non-finite points go to a trash cell past the grid (never searched): the gap-fill cell table must close exactly there
    g = orc.rng(404)
    tgt = g.uniform_points(6000, 8.0)
    tgt[::17, 0] = np.nan
    tgt[5::29, 2] = np.inf
    qry = g.uniform_points(500, 8.0)
    grid = sp.GridKNN.build(dev(tgt), points_per_cell=0.5)
    r = grid.knn_search(dev(qry), 3)
    oi, od = orc.knn_bruteforce(qry, tgt, 3)
    assert np.array_equal(r.indices.cpu().numpy(), oi) and np.array_equal(r.distances.cpu().numpy(), od)
