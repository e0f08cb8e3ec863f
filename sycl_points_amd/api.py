"""Host-side mirror of the reference's operator interface for the hot path, over the C ABI.

Names, argument meaning and error behaviour follow the reference (paths relative to
/root/reference/cpp/include/sycl_points/): PointCloudShared (points/point_cloud.hpp:73-476), KNNResult
(algorithms/knn/result.hpp), KNNBase / KDTree / knn_search_bruteforce (algorithms/knn/), covariance::estimate*
(algorithms/feature/covariance.hpp), VoxelGrid (algorithms/filter/voxel_downsampling.hpp), Registration
(algorithms/registration/registration.hpp). Every array is a torch CUDA tensor resident in HBM; torch is used for
device memory, streams and torch.distributed only — all compute goes through libsycl_points_amd.so.
"""
import ctypes as C
from dataclasses import dataclass, field

import numpy as np
import torch

from . import _lib
from ._lib import (DegenerateRegParams, FactorParams, GnParams, Linearized, MapPriorParams, MapPriorState, SpError,
                   check)

# sp_gicp_source_prepare's sort_by_cell: False keep order / True sort per alignment / "presorted" (GridKNN.order() or
# voxel-downsampling output order): keep order, block-walk search
_SOURCE_ORDER = {False: 0, True: 1, 0: 0, 1: 1, 2: 2, "presorted": 2}
REG = {"POINT_TO_POINT": 0, "POINT_TO_PLANE": 1, "POINT_TO_DISTRIBUTION": 2, "GICP": 3, "GENZ": 4}
LOSS = {"NONE": 0, "HUBER": 1, "TUKEY": 2, "CAUCHY": 3, "GEMAN_MCCLURE": 4}
FLT_MAX = float(np.finfo(np.float32).max)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _dev_f32(t, cols=None):
    if t is None:
        return None
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise SpError(1, "expected a contiguous float32 CUDA tensor")
    if cols is not None and (t.dim() != 2 or t.shape[1] != cols):
        raise SpError(1, f"expected shape (N, {cols}), got {tuple(t.shape)}")
    return t


def _T16(T):
    """4x4 (row-major numpy / torch CPU) -> 16 floats column-major (Eigen::Matrix4f::data())."""
    a = np.ascontiguousarray(np.asarray(T, dtype=np.float32).T)
    return a


def identity():
    return np.eye(4, dtype=np.float32)


class PointCloudShared:
    """points/point_cloud.hpp:73-476 — attribute containers resident in HBM. covs are (N,16) column-major 4x4."""

    def __init__(self, points=None, covs=None, normals=None, rgb=None, intensities=None, timestamp_offsets=None,
                 device="cuda"):
        self.device = torch.device(device)
        self.points = points if points is not None else torch.empty((0, 4), dtype=torch.float32, device=self.device)
        self.covs = covs
        self.normals = normals
        self.rgb = rgb
        self.intensities = intensities
        self.timestamp_offsets = timestamp_offsets

    @staticmethod
    def from_numpy(points, device="cuda", **attrs):
        pc = PointCloudShared(torch.from_numpy(np.ascontiguousarray(points, np.float32)).to(device), device=device)
        for k, v in attrs.items():
            if v is not None:
                setattr(pc, k, torch.from_numpy(np.ascontiguousarray(v, np.float32)).to(device))
        return pc

    def size(self):
        return int(self.points.shape[0])

    def reordered(self, perm):
        """A copy with every attribute gathered through `perm` (e.g. GridKNN.order())."""
        g = lambda t: None if t is None else t[perm].contiguous()
        return PointCloudShared(g(self.points), covs=g(self.covs), normals=g(self.normals), rgb=g(self.rgb),
                                intensities=g(self.intensities), timestamp_offsets=g(self.timestamp_offsets),
                                device=self.device)

    def has_cov(self):
        return self.covs is not None and self.covs.shape[0] == self.points.shape[0]

    def has_normal(self):
        return self.normals is not None and self.normals.shape[0] == self.points.shape[0]

    def has_rgb(self):
        return self.rgb is not None and self.rgb.shape[0] == self.points.shape[0]

    def has_intensity(self):
        return self.intensities is not None and self.intensities.shape[0] == self.points.shape[0]

    def has_timestamps(self):
        return (self.timestamp_offsets is not None and self.timestamp_offsets.shape[0] == self.points.shape[0]
                and self.points.shape[0] > 0)


@dataclass
class KNNResult:
    """algorithms/knn/result.hpp:12-34 — row-major (query_size, k), squared distances, -1 / FLT_MAX padding."""
    indices: torch.Tensor = None
    distances: torch.Tensor = None
    query_size: int = 0
    k: int = 0

    def allocate(self, query_size, k, device="cuda"):
        self.query_size, self.k = query_size, k
        self.indices = torch.full((query_size, k), -1, dtype=torch.int32, device=device)
        self.distances = torch.full((query_size, k), FLT_MAX, dtype=torch.float32, device=device)

    def resize(self, query_size, k, device="cuda"):
        """result.resize(query_size, k) (result.hpp:28-33); reallocates only when the shape changes."""
        if self.indices is None or tuple(self.indices.shape) != (query_size, k) or self.indices.device != torch.device(device):
            self.allocate(query_size, k, device)
        self.query_size, self.k = query_size, k


def _points_of(q):
    return q.points if isinstance(q, PointCloudShared) else q


def knn_search_bruteforce(queries, targets, k):
    """algorithms/knn/bruteforce.hpp:24-96 (synchronous in the reference; here enqueued on the current stream)."""
    q = _dev_f32(_points_of(queries), 4)
    t = _dev_f32(_points_of(targets), 4)
    L = _lib.lib()
    res = KNNResult()
    res.allocate(q.shape[0], k, q.device)
    nbytes = L.sp_knn_bruteforce_workspace_bytes(q.shape[0], t.shape[0], k)
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=q.device)
    check(L.sp_knn_bruteforce(_ptr(q), q.shape[0], _ptr(t), t.shape[0], k, _ptr(res.indices), _ptr(res.distances),
                              _ptr(ws), nbytes, _stream()))
    return res


class KNNBase:
    """algorithms/knn/knn.hpp:14-61 — the operator boundary Registration::align sits on."""

    def knn_search_async(self, queries, k, result, transT=None):
        raise NotImplementedError

    def knn_search(self, queries, k, transT=None):
        r = KNNResult()
        self.knn_search_async(queries, k, r, transT)
        torch.cuda.current_stream().synchronize()
        return r

    def nearest_neighbor_search_async(self, queries, result, transT=None):
        return self.knn_search_async(queries, 1, result, transT)

    def nearest_neighbor_search(self, queries, result, transT=None):
        self.nearest_neighbor_search_async(queries, result, transT)
        torch.cuda.current_stream().synchronize()


def _trans_arg(transT):
    """-> (pointer, on_device flag, keepalive). Accepts None, a 4x4 host matrix, or a 16-float CUDA tensor
    (column-major) for device-resident loops."""
    if transT is None:
        return None, 0, None
    if isinstance(transT, torch.Tensor) and transT.is_cuda:
        if transT.numel() != 16 or transT.dtype != torch.float32:
            raise SpError(1, "device transT must be 16 float32 (column-major)")
        return _ptr(transT), 1, transT
    a = _T16(transT.cpu().numpy() if isinstance(transT, torch.Tensor) else transT)
    return a.ctypes.data_as(C.c_void_p), 0, a


class KDTree(KNNBase):
    """algorithms/knn/kdtree.hpp:142-766.

    build(points) is the reference's tree (host build with its rule, device search). build(points, accelerate=True) mirrors the
    C++ facade's KDTree (include/sycl_points/amd/knn.hpp): from 1024 points on the hierarchy is built on the device (BVH) and
    knn_search answers from it for k <= 32; a search of the tree's OWN cloud with 8 <= k <= 20 on at least 32768 points of
    near-uniform density (fullest cell of a 6-points-per-cell grid <= 48 points) is answered by that grid; radius search, lazy
    delete, k > 32 use the reference's tree, built the first time it is needed. `backend_for` is the decision itself
    (tests/test_gpu_facade.py holds it to the facade's on the same clouds)."""

    DEVICE_BUILD_MIN_POINTS = 1024   # knn.hpp: KDTree::kDeviceBuildMinPoints
    GRID_SELF_MIN_POINTS = 32768     # knn.hpp: KDTree::kGridSelfMinPoints
    GRID_SELF_MAX_CELL = 48          # knn.hpp: KDTree::kGridSelfMaxCell
    GRID_SELF_POINTS_PER_CELL = 6.0
    BRUTE_FORCE_MAX_TARGETS = 16384  # knn.hpp: KDTree::kBruteForceMaxTargets
    BRUTE_FORCE_MAX_QUERIES = 65536  # knn.hpp: KDTree::kBruteForceMaxQueries

    def __init__(self, handle, n, device):
        self._h = handle
        self.n = n
        self.device = device
        self._hier = None         # the device-built hierarchy, built when a search first needs it (accelerate=True)
        self._points_version = 0
        self._accelerated = False
        self._points = None       # ... the points it was built on (device tensor), for the lazily built reference tree / the grid
        self._self_grid = None
        self._self_grid_tried = False
        self._removals = []       # lazy deletes the (not yet built) reference tree has to replay
        self._pristine = True
        self._leaf_threshold = 16

    @staticmethod
    def build(points, leaf_threshold=16, accelerate=False):
        p = _points_of(points)
        if accelerate and p.shape[0] >= KDTree.DEVICE_BUILD_MIN_POINTS:
            pd = _dev_f32(p, 4)
            t = KDTree(None, pd.shape[0], pd.device)
            t._accelerated = True  # (the hierarchy itself is built lazily, like the facade's: many trees never need it)
            t._points = pd
            t._points_version = pd._version  # (an in-place edit of the caller's tensor after build() ends the own-cloud shortcuts)
            t._leaf_threshold = leaf_threshold
            return t
        host = np.ascontiguousarray(p.detach().cpu().numpy() if isinstance(p, torch.Tensor) else p, np.float32)
        dev = p.device if isinstance(p, torch.Tensor) and p.is_cuda else torch.device("cuda")
        h = C.c_void_p()
        check(_lib.lib().sp_kdtree_create(host.ctypes.data_as(C.c_void_p), host.shape[0], leaf_threshold, _stream(),
                                          C.byref(h)))
        return KDTree(h, host.shape[0], dev)

    @property
    def _bvh(self):
        if self._accelerated and self._hier is None:
            if self._points._version != self._points_version:
                # (the facade's tree owns a copy of the points taken at build(); this mirror borrows the caller's tensor)
                raise SpError(2, "[KDTree] the points tensor was modified in place after build() and before the first search: "
                                 "build the tree again (or hand build() a clone)")
            self._hier = BVH.build(self._points)
        return self._hier

    def _host_tree(self):
        if not self._h:
            if self._points._version != self._points_version:
                raise SpError(2, "[KDTree] the points tensor was modified in place after build() and before the reference tree was "
                                 "needed: build the tree again (or hand build() a clone)")
            host = np.ascontiguousarray(self._points.detach().cpu().numpy(), np.float32)
            h = C.c_void_p()
            check(_lib.lib().sp_kdtree_create(host.ctypes.data_as(C.c_void_p), host.shape[0], self._leaf_threshold, _stream(),
                                              C.byref(h)))
            self._h = h
            # nodes removed while only the hierarchy existed: the same lazy deletes, in their order (KDTree::host_tree)
            for flags, indices in self._removals:
                check(_lib.lib().sp_kdtree_remove_by_flags(self._h, _ptr(flags), _ptr(indices), flags.shape[0], _stream()))
            torch.cuda.current_stream().synchronize()
            self._removals = []
        return self._h

    def _uniform_grid(self):
        if not self._self_grid_tried:
            self._self_grid_tried = True
            if self.n >= KDTree.GRID_SELF_MIN_POINTS:
                g = GridKNN.build(self._points, points_per_cell=KDTree.GRID_SELF_POINTS_PER_CELL)
                if g.max_cell_points() <= KDTree.GRID_SELF_MAX_CELL:
                    self._self_grid = g
        return self._self_grid

    def backend_for(self, queries, k, transT=None):
        """'kdtree' | 'bvh' | 'grid' | 'bruteforce': what knn_search_async(queries, k, ..., transT) answers from
        (KDTree::backend_for)."""
        if not self._accelerated or k > 32:
            return "kdtree"
        q = _points_of(queries)
        unchanged = self._points is not None and self._points._version == self._points_version
        own = (self._pristine and transT is None and isinstance(q, torch.Tensor) and q.is_cuda and q.shape[0] == self.n and
               q.data_ptr() == self._points.data_ptr() and unchanged)
        if own and 8 <= k <= 20 and self._uniform_grid() is not None:
            return "grid"
        # a small cloud (the reference example's 6 k-point downsampled scans): exact brute force beats building a hierarchy
        # (up to 12 k targets and 8 * 10^7 pairs: one launch with the cloud in LDS; beyond, from 2048 targets: the bounded passes)
        small = self.n <= 12032 and q.shape[0] * self.n <= 80_000_000
        if (self._pristine and unchanged and transT is None and k <= 20 and q.shape[0] <= KDTree.BRUTE_FORCE_MAX_QUERIES and
                (small or (2048 <= self.n <= KDTree.BRUTE_FORCE_MAX_TARGETS and self.n >= 256 * k))):
            return "bruteforce"
        return "bvh"

    def __del__(self):
        try:
            if self._h:
                _lib.lib().sp_kdtree_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def knn_search_async(self, queries, k, result, transT=None):
        q = _dev_f32(_points_of(queries), 4)
        if k > 100:
            raise SpError(2, "[KDTree::knn_search_async] `k` is too large. not support.")
        backend = self.backend_for(queries, k, transT) if self._accelerated else "kdtree"
        if backend == "bruteforce":
            res = knn_search_bruteforce(q, self._points, k)
            result.indices, result.distances, result.query_size, result.k = res.indices, res.distances, res.query_size, res.k
            return
        if backend == "grid":
            res = self._uniform_grid().self_knn(k)[0]
            result.indices, result.distances, result.query_size, result.k = res.indices, res.distances, res.query_size, res.k
            return
        if backend == "bvh":
            own = (self._pristine and transT is None and q.shape[0] == self.n and q.data_ptr() == self._points.data_ptr() and
                   self._points._version == self._points_version)
            if own:
                res = self._bvh.self_knn(k)
                result.indices, result.distances, result.query_size, result.k = res.indices, res.distances, res.query_size, res.k
            else:
                self._bvh.knn_search_async(queries, k, result, transT)
            return
        result.resize(q.shape[0], k, q.device)
        if q.shape[0] == 0:
            return
        tp, on_dev, keep = _trans_arg(transT)
        check(_lib.lib().sp_kdtree_search(self._host_tree(), _ptr(q), q.shape[0], k, tp, on_dev, _ptr(result.indices),
                                          _ptr(result.distances), _stream()))

    def radius_search_async(self, queries, max_k, radius, result, transT=None):
        q = _dev_f32(_points_of(queries), 4)
        if max_k > 100:
            raise SpError(2, "[KDTree::radius_search_async] `max_k` is too large. not support.")
        if q.shape[0] == 0 or max_k == 0:
            result.resize(0, 0, q.device)
            return
        if self._accelerated and max_k <= 32:
            self._bvh.radius_search_async(queries, max_k, radius, result, transT)
            return
        result.resize(q.shape[0], max_k, q.device)
        tp, on_dev, keep = _trans_arg(transT)
        check(_lib.lib().sp_kdtree_radius_search(self._host_tree(), _ptr(q), q.shape[0], max_k, radius, tp, on_dev,
                                                 _ptr(result.indices), _ptr(result.distances), _stream()))

    def remove_nodes_by_flags(self, flags, indices):
        if flags.shape[0] != indices.shape[0]:
            raise SpError(2, "[KDTree::remove_nodes_by_flags_impl] flags and indices must have the same size.")
        if self._accelerated:
            self._bvh.remove_nodes_by_flags(flags, indices)
        if self._h or not self._accelerated:
            check(_lib.lib().sp_kdtree_remove_by_flags(self._host_tree(), _ptr(flags), _ptr(indices), flags.shape[0], _stream()))
        else:  # (a reference tree built from here on starts from the original points: it replays these)
            self._removals.append((flags.clone(), indices.clone()))
        torch.cuda.current_stream().synchronize()
        self._pristine = False  # no own-cloud shortcut / grid any more: the points carry other indices now
        self._self_grid = None
        self._self_grid_tried = True


class BVH(KNNBase):
    """MI355X-native KNNBase for clouds of any density profile (csrc/bvh.hip): a bounding-volume hierarchy over the
    Morton-sorted points, built entirely on the device (the job of KDTree::build, kdtree.hpp:292-413, without its host
    build); exact kNN, k <= 32, bit-identical to knn_search_bruteforce."""

    def __init__(self, handle, n, device):
        self._h = handle
        self.n = n
        self.device = device

    @staticmethod
    def build(points):
        p = _dev_f32(_points_of(points), 4)
        h = C.c_void_p()
        check(_lib.lib().sp_bvh_create(_ptr(p), p.shape[0], _stream(), C.byref(h)))
        return BVH(h, p.shape[0], p.device)

    def __del__(self):
        try:
            if self._h:
                _lib.lib().sp_bvh_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def knn_search_async(self, queries, k, result, transT=None):
        q = _dev_f32(_points_of(queries), 4)
        if k > 32:
            raise SpError(2, "[BVH::knn_search_async] `k` is too large. not support.")
        result.resize(q.shape[0], k, q.device)
        if q.shape[0] == 0:
            return
        tp, on_dev, keep = _trans_arg(transT)
        check(_lib.lib().sp_bvh_search(self._h, _ptr(q), q.shape[0], k, tp, on_dev, _ptr(result.indices),
                                       _ptr(result.distances), _stream()))

    def _set_option(self, name, value):
        """csrc/sp_internal.h switches of this handle (tests, comparisons)."""
        check(_lib.lib().sp_internal_bvh_option(self._h, _lib.INTERNAL_OPTION[name], int(value)))

    def self_knn(self, k):
        """The cloud's own points as queries, in tree order (row i = neighbours of point i)."""
        res = KNNResult()
        res.resize(self.n, k, self.device)
        if self.n:
            check(_lib.lib().sp_bvh_self_knn(self._h, k, _ptr(res.indices), _ptr(res.distances), _stream()))
        return res

    def radius_search_async(self, queries, max_k, radius, result, transT=None):
        """KDTree::radius_search_async (kdtree.hpp:574-719) on the device-built hierarchy (sp_bvh_radius_search), max_k <= 32."""
        q = _dev_f32(_points_of(queries), 4)
        if max_k > 32:
            raise SpError(2, "[BVH::radius_search_async] `max_k` is too large (max 32).")
        if q.shape[0] == 0 or max_k == 0:
            result.resize(0, 0, q.device)
            return
        result.resize(q.shape[0], max_k, q.device)
        tp, on_dev, keep = _trans_arg(transT)
        check(_lib.lib().sp_bvh_radius_search(self._h, _ptr(q), q.shape[0], max_k, radius, tp, on_dev, _ptr(result.indices),
                                              _ptr(result.distances), _stream()))

    def radius_search(self, queries, max_k, radius, transT=None):
        r = KNNResult()
        self.radius_search_async(queries, max_k, radius, r, transT)
        torch.cuda.current_stream().synchronize()
        return r

    def remove_nodes_by_flags(self, flags, indices):
        """KDTree::remove_nodes_by_flags (kdtree.hpp:721-765), lazily: sp_bvh_remove_by_flags."""
        if flags.shape[0] != indices.shape[0]:
            raise SpError(2, "[BVH::remove_nodes_by_flags] flags and indices must have the same size.")
        check(_lib.lib().sp_bvh_remove_by_flags(self._h, _ptr(flags), _ptr(indices), flags.shape[0], _stream()))


class GridKNN(KNNBase):
    """MI355X-native KNNBase: exact kNN on a device-built uniform grid (csrc/grid.hip); bit-identical to
    knn_search_bruteforce. `points_per_cell` tunes the cell size (about 2 for k = 1, about 6-8 for k = 20)."""

    def __init__(self, handle, n, device):
        self._h = handle
        self.n = n
        self.device = device

    @staticmethod
    def build(points, cell_size=0.0, points_per_cell=2.0, adaptive=False, bounds=None):
        """adaptive: the cell size follows the measured occupancy (sp_grid_create_adaptive) — clouds of surfaces.
        bounds: (min x, y, z, max x, y, z) of a box the caller knows to hold every finite point (sp_grid_create_bounded: no
        bounding-box kernel, no read-back)."""
        p = _dev_f32(_points_of(points), 4)
        h = C.c_void_p()
        if bounds is not None:
            b = (C.c_float * 6)(*[float(v) for v in bounds])
            check(_lib.lib().sp_grid_create_bounded(_ptr(p), p.shape[0], b, cell_size, points_per_cell, _stream(), C.byref(h)))
        elif adaptive:
            check(_lib.lib().sp_grid_create_adaptive(_ptr(p), p.shape[0], points_per_cell, _stream(), C.byref(h)))
        else:
            check(_lib.lib().sp_grid_create(_ptr(p), p.shape[0], cell_size, points_per_cell, _stream(), C.byref(h)))
        return GridKNN(h, p.shape[0], p.device)

    def _set_option(self, name, value):
        """csrc/sp_internal.h tuning switch of this grid (tests / scratch only)."""
        check(_lib.lib().sp_internal_grid_option(self._h, _lib.INTERNAL_OPTION[name], int(value)))

    def cell_size(self):
        return float(_lib.lib().sp_grid_cell_size(self._h))

    def max_cell_points(self):
        """Points in the fullest cell at build time (sp_grid_max_cell_points): far above the points-per-cell target means the
        cloud is not of near-uniform density and a BVH serves it better."""
        return int(_lib.lib().sp_grid_max_cell_points(self._h))

    def order(self):
        """Original indices of the points in this grid's cell order (int64 tensor, usable for tensor indexing). A source
        cloud stored in this order needs no per-alignment sort (align_fused_loop(sort_by_cell="presorted"))."""
        out = torch.empty(self.n, dtype=torch.int32, device=self.device)
        check(_lib.lib().sp_grid_order(self._h, _ptr(out), _stream()))
        return out.long()

    def __del__(self):
        try:
            if self._h:
                _lib.lib().sp_grid_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def self_knn(self, k, want_knn=True, want_covs=False, want_normals=False):
        """kNN of the grid's own cloud (tile kernel, csrc/grid.hip) with covariance / normal estimation optionally
        fused in — covariance::estimate_async(knn, points, k) (covariance.hpp:305-311) when the KNNBase is a GridKNN
        built on `points`. Returns (KNNResult | None, covs | None, normals | None), rows in original point order."""
        if k > 20:
            raise SpError(2, "[GridKNN::knn_search_async] `k` is too large (max 20).")
        L = _lib.lib()
        dev = self.device
        res = None
        if want_knn:
            res = KNNResult()
            res.allocate(self.n, k, dev)
        covs = torch.empty((self.n, 16), dtype=torch.float32, device=dev) if want_covs else None
        nrm = torch.empty((self.n, 4), dtype=torch.float32, device=dev) if want_normals else None
        nbytes = L.sp_grid_self_workspace_bytes(self._h)
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
        check(L.sp_grid_self_knn(self._h, k, _ptr(res.indices) if res else None, _ptr(res.distances) if res else None,
                                 _ptr(covs), _ptr(nrm), _ptr(ws), nbytes, _stream()))
        self._keep = ws
        return res, covs, nrm

    def covariances_sharded(self, k, rank, world, all_gather):
        """The pre-loop of a multi-GPU run sharded by query: this rank computes the k-neighbour covariances
        of the grid positions [rank * c, (rank + 1) * c), c = ceil(n / world) (sp_grid_self_knn_range), the chunks — made
        contiguous by sp_grid_gather_rows — are exchanged by `all_gather(send, recv)` (Communicator.all_gather, or a
        torch.distributed all_gather_into_tensor), and sp_grid_scatter_rows puts all n rows back into the cloud's order.
        Bit-identical to self_knn(k, want_covs=True) of the whole cloud."""
        L = _lib.lib()
        dev = self.device
        n = self.n
        c = (n + world - 1) // world
        first = min(rank * c, n)
        count = min(c, n - first)
        covs = torch.empty((n, 16), dtype=torch.float32, device=dev)
        nbytes = L.sp_grid_self_workspace_bytes(self._h)
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
        check(L.sp_grid_self_knn_range(self._h, k, first, count, None, None, _ptr(covs), None, _ptr(ws), nbytes, _stream()))
        send = torch.zeros((c, 16), dtype=torch.float32, device=dev)  # equal chunks; the last rank's tail stays zero
        check(L.sp_grid_gather_rows(self._h, _ptr(covs), 64, first, count, _ptr(send), _stream()))
        recv = torch.empty((world * c, 16), dtype=torch.float32, device=dev)
        all_gather(send, recv)
        check(L.sp_grid_scatter_rows(self._h, _ptr(recv), 64, 0, n, _ptr(covs), _stream()))
        self._keep = (ws, send, recv)
        return covs

    def knn_search_async(self, queries, k, result, transT=None):
        q = _dev_f32(_points_of(queries), 4)
        if k > 20:
            raise SpError(2, "[GridKNN::knn_search_async] `k` is too large (max 20).")
        result.resize(q.shape[0], k, q.device)
        if q.shape[0] == 0:
            return
        tp, on_dev, keep = _trans_arg(transT)
        check(_lib.lib().sp_grid_search(self._h, _ptr(q), q.shape[0], k, tp, on_dev, _ptr(result.indices),
                                        _ptr(result.distances), _stream()))

    def radius_search_async(self, queries, max_k, radius, result, transT=None):
        """The grid's counterpart of KDTree::radius_search_async (kdtree.hpp:251-280): the max_k nearest within radius."""
        q = _dev_f32(_points_of(queries), 4)
        if max_k > 20:
            raise SpError(2, "[GridKNN::radius_search_async] `max_k` is too large (max 20).")
        if q.shape[0] == 0 or max_k == 0:
            result.resize(0, 0, q.device)
            return
        result.resize(q.shape[0], max_k, q.device)
        tp, on_dev, keep = _trans_arg(transT)
        check(_lib.lib().sp_grid_radius_search(self._h, _ptr(q), q.shape[0], max_k, radius, tp, on_dev,
                                               _ptr(result.indices), _ptr(result.distances), _stream()))

    def radius_search(self, queries, max_k, radius, transT=None):
        result = KNNResult()
        self.radius_search_async(queries, max_k, radius, result, transT)
        return result

    def remove_nodes_by_flags(self, flags, indices):
        """The grid's counterpart of KDTree::remove_nodes_by_flags (kdtree.hpp:282-284): flags 1 = keep; kept point p is
        relabelled indices[p]. Prepared targets built on this grid must be re-created."""
        if flags.shape[0] != indices.shape[0]:
            raise SpError(2, "[GridKNN::remove_nodes_by_flags] flags and indices must have the same size.")
        check(_lib.lib().sp_grid_remove_by_flags(self._h, _ptr(flags), _ptr(indices), flags.shape[0], _stream()))
        self.n = int(_lib.lib().sp_grid_size(self._h))


class BruteForceKNN(KNNBase):
    """A KNNBase over knn_search_bruteforce (the reference tests inject such host fakes through the same seam,
    tests/test_registration_pipeline.cpp:16-61). The query transform is applied with sp_transform first."""

    def __init__(self, targets):
        self.targets = _dev_f32(_points_of(targets), 4)

    def knn_search_async(self, queries, k, result, transT=None):
        q = _dev_f32(_points_of(queries), 4)
        if transT is not None:
            if isinstance(transT, torch.Tensor) and transT.is_cuda:
                transT = transT.cpu().numpy().reshape(4, 4).T
            q = transform_points(q, transT)
        r = knn_search_bruteforce(q, self.targets, k)
        result.indices, result.distances, result.query_size, result.k = r.indices, r.distances, r.query_size, r.k


# ------------------------------------------------------------------ covariance / normals
class covariance:
    """algorithms/feature/covariance.hpp"""

    @staticmethod
    def estimate(neighbors, points):
        """estimate_async (covariance.hpp:260-311): returns (N,16) column-major covariances."""
        p = _dev_f32(_points_of(points), 4)
        idx = neighbors.indices if isinstance(neighbors, KNNResult) else neighbors
        covs = torch.empty((p.shape[0], 16), dtype=torch.float32, device=p.device)
        check(_lib.lib().sp_cov_estimate(_ptr(p), p.shape[0], _ptr(idx), idx.shape[1] if idx.dim() == 2 else 0,
                                         _ptr(covs), _stream()))
        if isinstance(points, PointCloudShared):
            points.covs = covs
        return covs

    @staticmethod
    def estimate_robust(neighbors, points, robust_type="CAUCHY", mad_scale=1.0, min_robust_scale=1.0,
                        robust_max_iterations=1):
        """estimate_robust_async (covariance.hpp:323-381): M-estimated covariances, same layout as estimate()."""
        p = _dev_f32(_points_of(points), 4)
        idx = neighbors.indices if isinstance(neighbors, KNNResult) else neighbors
        covs = torch.empty((p.shape[0], 16), dtype=torch.float32, device=p.device)
        check(_lib.lib().sp_cov_estimate_robust(_ptr(p), p.shape[0], _ptr(idx), idx.shape[1] if idx.dim() == 2 else 0,
                                                LOSS[robust_type], mad_scale, min_robust_scale, robust_max_iterations,
                                                _ptr(covs), _stream()))
        if isinstance(points, PointCloudShared):
            points.covs = covs
        return covs

    @staticmethod
    def normalize_covariance(covs):
        """kernel::normalize_covariance over an array (covariance.hpp:76-95)"""
        out = torch.empty_like(covs)
        check(_lib.lib().sp_cov_normalize(_ptr(covs), covs.shape[0], _ptr(out), _stream()))
        return out

    @staticmethod
    def estimate_normals(neighbors, points):
        """estimate_normals_async (covariance.hpp:417-459)"""
        p = _dev_f32(_points_of(points), 4)
        idx = neighbors.indices if isinstance(neighbors, KNNResult) else neighbors
        nrm = torch.empty((p.shape[0], 4), dtype=torch.float32, device=p.device)
        check(_lib.lib().sp_normals_from_knn(_ptr(p), p.shape[0], _ptr(idx), idx.shape[1], _ptr(nrm), _stream()))
        if isinstance(points, PointCloudShared):
            points.normals = nrm
        return nrm

    @staticmethod
    def extract_normals(points, covs=None):
        """extract_normals_async (covariance.hpp:465-503)"""
        p = _dev_f32(_points_of(points), 4)
        c = covs if covs is not None else (points.covs if isinstance(points, PointCloudShared) else None)
        if c is None:
            raise SpError(2, "[covariance::extract_normals_async] covariances not computed")
        nrm = torch.empty((p.shape[0], 4), dtype=torch.float32, device=p.device)
        check(_lib.lib().sp_normals_from_cov(_ptr(p), _ptr(c), p.shape[0], _ptr(nrm), _stream()))
        if isinstance(points, PointCloudShared):
            points.normals = nrm
        return nrm

    @staticmethod
    def update_covariance_plane(covs):
        """kernel::update_covariance_plane over an array (covariance.hpp:67-74)"""
        out = torch.empty_like(covs)
        check(_lib.lib().sp_cov_update_plane(_ptr(covs), covs.shape[0], _ptr(out), _stream()))
        return out


# ------------------------------------------------------------------ voxel grid
class VoxelGrid:
    """algorithms/filter/voxel_downsampling.hpp:14-289"""

    def __init__(self, voxel_size):
        if voxel_size <= 0.0:
            raise SpError(1, "voxel_size must be positive")
        self.voxel_size = float(voxel_size)
        self.voxel_size_inv = float(np.float32(1.0) / np.float32(voxel_size))
        self._key_box = None  # the remembered key box is in units of the old voxel size  # voxel_downsampling.hpp:27
        self.min_voxel_count = 1

    def set_voxel_size(self, voxel_size):
        if voxel_size <= 0.0:
            raise SpError(1, "voxel_size must be positive")
        self.voxel_size = float(voxel_size)
        self.voxel_size_inv = float(np.float32(1.0) / np.float32(voxel_size))

    def get_voxel_size(self):
        return self.voxel_size

    def set_min_voxel_count(self, n):
        self.min_voxel_count = int(n)

    def compute_voxel_bit(self, points):
        p = _dev_f32(_points_of(points), 4)
        keys = torch.empty(p.shape[0], dtype=torch.int64, device=p.device)  # bit pattern of the uint64 keys
        check(_lib.lib().sp_voxel_keys(_ptr(p), p.shape[0], self.voxel_size_inv, _ptr(keys), _stream()))
        return keys

    def downsampling(self, cloud, return_keys=False, boxed=True):
        """downsampling(cloud, result) (voxel_downsampling.hpp:64-79). Synchronises to read the voxel count, as the
        reference's host aggregation does. boxed=True sorts keys compressed to the (widened) bounding box of the PREVIOUS
        cloud's voxel coordinates — scans of one sensor have similar extents — verifies on the device that the cloud fits,
        and falls back to the 64-bit sort when it does not: identical results either way."""
        pc = cloud if isinstance(cloud, PointCloudShared) else PointCloudShared(cloud)
        p = _dev_f32(pc.points, 4)
        n = p.shape[0]
        out = PointCloudShared(device=p.device)
        if n == 0:
            return (out, torch.empty(0, dtype=torch.int64, device=p.device)) if return_keys else out
        L = _lib.lib()
        nbytes = L.sp_voxel_downsample_workspace_bytes(n)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=p.device)
        o_p = torch.empty((n, 4), dtype=torch.float32, device=p.device)
        rgb = pc.rgb if pc.has_rgb() else None
        inten = pc.intensities if pc.has_intensity() else None
        ts = pc.timestamp_offsets if pc.has_timestamps() else None
        o_c = torch.empty((n, 4), dtype=torch.float32, device=p.device) if rgb is not None else None
        o_i = torch.empty(n, dtype=torch.float32, device=p.device) if inten is not None else None
        o_t = torch.empty(n, dtype=torch.float32, device=p.device) if ts is not None else None
        o_k = torch.empty(n, dtype=torch.int64, device=p.device) if return_keys else None
        # One read-back at the end: voxel count, boxed-path status, and this cloud's key box (sharded, found by the key
        # kernel on the way), which the NEXT call uses, widened by a margin, to sort keys compressed to that box. A cloud that
        # leaves the remembered box is detected (status != 0) and redone with its own exact box, so results never depend on
        # the guess. The first call of a VoxelGrid has no guess: it computes the box first (sp_voxel_key_box, one small
        # read-back) — every call sorts compressed keys; the 64-bit sort is left for boxes of >= 2^32 cells.
        # (sp_voxel_downsample_report: all of it in ONE 8-word record the call's last kernel stores — no status word and sharded
        # box to initialise and fold)
        info = torch.zeros(16, dtype=torch.int32, device=p.device)
        base = info.data_ptr()
        args = (_ptr(p), n, self.voxel_size_inv, self.min_voxel_count, _ptr(rgb), _ptr(inten), _ptr(ts), _ptr(o_p), _ptr(o_c),
                _ptr(o_i), _ptr(o_t), _ptr(o_k), None)

        def run(box):
            check(L.sp_voxel_downsample_report(*args, None if box is None else box.ctypes.data_as(C.c_void_p),
                                               C.c_void_p(base), _ptr(ws), nbytes, _stream()))
            c = info.cpu().numpy()
            return c, c[2:8].astype(np.int64)

        guess = getattr(self, "_key_box", None) if boxed else None
        if guess is None and boxed:
            check(L.sp_voxel_key_box(_ptr(p), n, self.voxel_size_inv, C.c_void_p(base + 32), _stream()))
            b0 = info[8:14].cpu().numpy().astype(np.int64)
            guess = np.ascontiguousarray(b0.astype(np.int32)) if (b0[:3] <= b0[3:]).all() else None
        counts, box = run(guess)
        if counts[1] != 0:  # the cloud left the remembered box: again, with its own
            counts, box = run(np.ascontiguousarray(box.astype(np.int32)))
        if boxed and (box[:3] <= box[3:]).all():
            margin = np.maximum(2, (box[3:] - box[:3] + 1) // 8)
            lo = np.maximum(box[:3] - margin, 0)
            hi = np.minimum(box[3:] + margin, (1 << 21) - 1)
            # Keep what earlier clouds needed as well (one VoxelGrid usually serves several scans in turn — source and target
            # of a registration —, and a guess that forgets the other scan is redone every call), unless that has grown to
            # more than 8x the cells this cloud needs.
            prev = getattr(self, "_key_box", None)
            if prev is not None:
                ulo, uhi = np.minimum(lo, prev[:3]), np.maximum(hi, prev[3:])
                if np.prod((uhi - ulo + 1).astype(np.float64)) <= 8.0 * np.prod((hi - lo + 1).astype(np.float64)):
                    lo, hi = ulo, uhi
            self._key_box = np.ascontiguousarray(np.concatenate([lo, hi]).astype(np.int32))
        v = int(counts[0])
        out.points = o_p[:v]
        out.rgb = None if o_c is None else o_c[:v]
        out.intensities = None if o_i is None else o_i[:v]
        out.timestamp_offsets = None if o_t is None else o_t[:v]
        return (out, o_k[:v]) if return_keys else out


# ------------------------------------------------------------------ transform / filters
def transform_points(points, T):
    p = _dev_f32(points, 4)
    out = torch.empty_like(p)
    a = _T16(T)
    check(_lib.lib().sp_transform(_ptr(p), None, None, p.shape[0], a.ctypes.data_as(C.c_void_p), _ptr(out), None, None,
                                  _stream()))
    return out


def transform(cloud, T):
    """transform::transform (common/transform.hpp:45-101), in place on the cloud's attributes."""
    a = _T16(T)
    n = cloud.size()
    covs = cloud.covs if cloud.has_cov() else None
    nrm = cloud.normals if cloud.has_normal() else None
    check(_lib.lib().sp_transform(_ptr(cloud.points), _ptr(covs), _ptr(nrm), n, a.ctypes.data_as(C.c_void_p),
                                  _ptr(cloud.points), _ptr(covs), _ptr(nrm), _stream()))
    return cloud


def transform_copy(cloud, T):
    out = PointCloudShared(cloud.points.clone(), None if cloud.covs is None else cloud.covs.clone(),
                           None if cloud.normals is None else cloud.normals.clone(), cloud.rgb, cloud.intensities,
                           cloud.timestamp_offsets, device=cloud.points.device)
    return transform(out, T)


def box_filter_flags(points, min_distance, max_distance):
    p = _dev_f32(_points_of(points), 4)
    flags = torch.empty(p.shape[0], dtype=torch.uint8, device=p.device)
    check(_lib.lib().sp_box_filter_flags(_ptr(p), p.shape[0], min_distance, max_distance, _ptr(flags), _stream()))
    return flags


def compact_by_flags(rows, flags, want_indices=False):
    """FilterByFlags::filter_by_flags / calculate_indices (common/filter_by_flags.hpp:30-99) on the device."""
    L = _lib.lib()
    n = rows.shape[0]
    row_bytes = rows.element_size() * (rows.numel() // max(n, 1)) if n else 4
    out = torch.empty_like(rows)
    idx = torch.empty(n, dtype=torch.int32, device=rows.device) if want_indices else None
    n_out = torch.zeros(1, dtype=torch.int32, device=rows.device)
    nbytes = L.sp_compact_workspace_bytes(n)
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=rows.device)
    check(L.sp_compact_by_flags(_ptr(rows), n, row_bytes, _ptr(flags), _ptr(out), _ptr(idx), _ptr(n_out), _ptr(ws),
                                nbytes, _stream()))
    v = int(n_out.item())
    return (out[:v], idx) if want_indices else out[:v]


class Communicator:
    """sp_comm: the library's own RCCL communicator (one process per GPU). The 128-byte unique id is made by rank 0
    (sp_comm_unique_id) and handed to the other ranks through whatever channel the application has — here a
    torch.distributed broadcast over an existing process group (any backend)."""

    def __init__(self, handle, rank, world):
        self._h, self.rank, self.world = handle, rank, world

    @staticmethod
    def from_process_group(group=None, device=None):
        import torch.distributed as dist

        L = _lib.lib()
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        backend = dist.get_backend(group)
        dev = torch.device("cuda", torch.cuda.current_device()) if (device is None and backend == "nccl") else (device or "cpu")
        ident = torch.zeros(_lib.COMM_ID_BYTES, dtype=torch.uint8)
        if rank == 0:
            buf = (C.c_ubyte * _lib.COMM_ID_BYTES)()
            check(L.sp_comm_unique_id(buf))
            ident = torch.tensor(list(buf), dtype=torch.uint8)
        ident = ident.to(dev)
        dist.broadcast(ident, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        raw = bytes(ident.cpu().tolist())
        h = C.c_void_p()
        check(L.sp_comm_create(raw, rank, world, C.byref(h)))  # collective: every rank of the group gets here
        return Communicator(h, rank, world)

    def all_reduce(self, t):
        """In-place float32 sum over the ranks on the current stream (sp_allreduce_f32)."""
        check(_lib.lib().sp_allreduce_f32(self._h, _ptr(t), t.numel(), _stream()))
        return t

    def all_gather(self, send, recv):
        check(_lib.lib().sp_allgather(self._h, _ptr(send), _ptr(recv), send.numel() * send.element_size(), _stream()))
        return recv

    def __del__(self):
        try:
            if self._h:
                _lib.lib().sp_comm_destroy(self._h)
                self._h = None
        except Exception:
            pass


class Exchange:
    """Direct exchange of the per-iteration row between the ranks of a sharded alignment (sp_xchg_*, csrc/sp_xchg.h): no
    collective per iteration. `all_gather(bytes) -> list of bytes in rank order` is the caller's own channel, used once."""

    def __init__(self, rank, world, all_gather, timeout_ms=None):
        L = _lib.lib()
        h = C.c_void_p()
        check(L.sp_xchg_create(int(rank), int(world), C.byref(h)))
        self._h = h
        self.rank, self.world = int(rank), int(world)
        mine = (C.c_char * 64)()
        check(L.sp_xchg_handle(self._h, mine))
        handles = all_gather(bytes(mine))
        blob = b"".join(handles)
        assert len(blob) == 64 * self.world
        check(L.sp_xchg_connect(self._h, C.c_char_p(blob)))
        if timeout_ms is not None:
            check(L.sp_xchg_set_timeout_ms(self._h, int(timeout_ms)))

    @staticmethod
    def from_process_group(group=None, timeout_ms=None):
        import torch.distributed as dist

        rank, world = dist.get_rank(group), dist.get_world_size(group)

        def gather(b):
            out = [None] * world
            dist.all_gather_object(out, b, group=group)
            return out

        return Exchange(rank, world, gather, timeout_ms)

    def __del__(self):
        try:
            if self._h:
                _lib.lib().sp_xchg_destroy(self._h)
                self._h = None
        except Exception:
            pass


class VoxelHashMap:
    """algorithms/mapping/voxel_hash_map.hpp:22-250 over the sp_vhm_* entry points: submap accumulation in HBM.
    add_point_cloud takes a PointCloudShared in the sensor frame and the sensor pose (4x4, map frame); downsampling
    returns a PointCloudShared of the voxel means inside the query box (plus `.voxel_keys`, the keys in output order)."""
    _PARAM = {"voxel_size": 0, "max_staleness": 1, "remove_old_data_cycle": 2, "rehash_threshold": 3, "min_num_point": 4}
    _INFO = {"voxel_num": 0, "capacity": 1, "staleness_counter": 2, "has_cov": 3, "has_rgb": 4, "has_intensity": 5}

    def __init__(self, voxel_size, device="cuda"):
        self.device = torch.device(device)
        h = C.c_void_p()
        check(_lib.lib().sp_vhm_create(float(voxel_size), _stream(), C.byref(h)))
        self._h = h

    def __del__(self):
        try:
            if self._h:
                _lib.lib().sp_vhm_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def _set(self, name, value):
        check(_lib.lib().sp_vhm_set(self._h, self._PARAM[name], float(value)))

    def _get(self, name):
        return float(_lib.lib().sp_vhm_get(self._h, self._PARAM[name]))

    def info(self, name):
        return int(_lib.lib().sp_vhm_info(self._h, self._INFO[name]))

    def set_voxel_size(self, v): self._set("voxel_size", v)  # noqa: E704
    def get_voxel_size(self): return self._get("voxel_size")  # noqa: E704
    def set_max_staleness(self, v): self._set("max_staleness", v)  # noqa: E704
    def get_max_staleness(self): return int(self._get("max_staleness"))  # noqa: E704
    def set_remove_old_data_cycle(self, v): self._set("remove_old_data_cycle", v)  # noqa: E704
    def get_remove_old_data_cycle(self): return int(self._get("remove_old_data_cycle"))  # noqa: E704
    def set_rehash_threshold(self, v): self._set("rehash_threshold", v)  # noqa: E704
    def get_rehash_threshold(self): return self._get("rehash_threshold")  # noqa: E704
    def set_min_num_point(self, v): self._set("min_num_point", v)  # noqa: E704
    def get_min_num_point(self): return int(self._get("min_num_point"))  # noqa: E704

    def clear(self):
        check(_lib.lib().sp_vhm_clear(self._h, _stream()))

    def add_point_cloud(self, cloud, sensor_pose=None):
        T = _T16(identity() if sensor_pose is None else sensor_pose).copy()
        n = cloud.size()
        check(_lib.lib().sp_vhm_add_point_cloud(
            self._h, _ptr(cloud.points) if n else None, _ptr(cloud.covs) if (n and cloud.has_cov()) else None,
            _ptr(cloud.rgb) if (n and cloud.has_rgb()) else None,
            _ptr(cloud.intensities) if (n and cloud.has_intensity()) else None, n, T.ctypes.data_as(C.c_void_p), _stream()))

    def downsampling(self, center=(0.0, 0.0, 0.0), distance=100.0):
        cap = self.info("voxel_num")
        dev = self.device
        pts = torch.empty((max(cap, 1), 4), dtype=torch.float32, device=dev)
        covs = torch.empty((max(cap, 1), 16), dtype=torch.float32, device=dev) if self.info("has_cov") else None
        rgb = torch.empty((max(cap, 1), 4), dtype=torch.float32, device=dev) if self.info("has_rgb") else None
        inten = torch.empty(max(cap, 1), dtype=torch.float32, device=dev) if self.info("has_intensity") else None
        keys = torch.empty(max(cap, 1), dtype=torch.int64, device=dev)
        c = np.asarray(center, np.float32).copy()
        n_out = C.c_size_t(0)
        check(_lib.lib().sp_vhm_downsampling(self._h, c.ctypes.data_as(C.c_void_p), float(distance), _ptr(pts), _ptr(covs),
                                             _ptr(rgb), _ptr(inten), _ptr(keys), cap, C.byref(n_out), _stream()))
        n = n_out.value
        out = PointCloudShared(pts[:n], covs=None if covs is None else covs[:n], rgb=None if rgb is None else rgb[:n],
                               intensities=None if inten is None else inten[:n], device=dev)
        out.voxel_keys = keys[:n]
        return out

    def compute_overlap_ratio(self, cloud, sensor_pose=None):
        T = _T16(identity() if sensor_pose is None else sensor_pose).copy()
        r = C.c_float(0.0)
        n = cloud.size()
        check(_lib.lib().sp_vhm_overlap_ratio(self._h, _ptr(cloud.points) if n else None, n, T.ctypes.data_as(C.c_void_p),
                                              C.byref(r), _stream()))
        return float(r.value)

    def remove_old_data(self):
        check(_lib.lib().sp_vhm_remove_old_data(self._h, _stream()))


class PreparedTarget:
    """Plane-regularised target covariances stored in the cell order of a GridKNN (sp_gicp_target_*): the target half of
    the prepared / fused GICP iteration. Holds a reference to the grid, which it borrows."""

    def __init__(self, grid, covs, reg_type="GICP"):
        self.grid = grid
        self.covs = _dev_f32(covs, 16)
        self.reg_type = reg_type
        h = C.c_void_p()
        check(_lib.lib().sp_gicp_target_create(grid._h, _ptr(self.covs), self.covs.shape[0], _stream(), C.byref(h)))
        self._h = h
        if reg_type != "GICP":  # POINT_TO_DISTRIBUTION: the rows hold inverse(Ct) (sp_gicp_target_prepare)
            check(_lib.lib().sp_gicp_target_prepare(self._h, _ptr(self.covs), REG[reg_type], _stream()))

    def update(self, covs=None, reg_type=None):
        if covs is not None:
            self.covs = _dev_f32(covs, 16)
        if reg_type is not None:
            self.reg_type = reg_type
        check(_lib.lib().sp_gicp_target_prepare(self._h, _ptr(self.covs), REG[self.reg_type], _stream()))

    def __del__(self):
        try:
            if self._h:
                _lib.lib().sp_gicp_target_destroy(self._h)
                self._h = None
        except Exception:
            pass


class PreparedSource:
    """Prepared source of the fused GICP iteration (sp_gicp_source_*): packed plane-regularised covariances and,
    optionally, the cloud reordered by target-grid cell. Buffers are allocated once for n_max points."""

    def __init__(self, n_max):
        h = C.c_void_p()
        check(_lib.lib().sp_gicp_source_create(n_max, C.byref(h)))
        self._h = h
        self.n_max = n_max

    def _set_option(self, name, value):
        """csrc/sp_internal.h measurement / tuning switch of this prepared source (tests / bench / scratch only)."""
        check(_lib.lib().sp_internal_source_option(self._h, _lib.INTERNAL_OPTION[name], int(value)))

    def prepare(self, prepared_target, source, transT=None, sort_by_cell=True):
        p2d = getattr(prepared_target, "reg_type", "GICP") == "POINT_TO_DISTRIBUTION"  # no source covariance in that factor
        if not source.has_cov() and not p2d:
            raise SpError(2, "[Registration::validate_params] Covariance matrices of source and target must be "
                             "pre-computed before performing GICP matching.")
        tp, on_dev, keep = _trans_arg(transT)
        check(_lib.lib().sp_gicp_source_prepare(self._h, prepared_target._h, _ptr(source.points),
                                                _ptr(source.covs if source.has_cov() else None),
                                                source.size(), tp, on_dev, _SOURCE_ORDER[sort_by_cell], _stream()))
        self.n = source.size()

    def __del__(self):
        try:
            if self._h:
                _lib.lib().sp_gicp_source_destroy(self._h)
                self._h = None
        except Exception:
            pass


# ------------------------------------------------------------------ registration
@dataclass
class RegistrationParams:
    """registration_params.hpp:46-114 (defaults copied from there)."""
    reg_type: str = "GICP"
    max_correspondence_distance: float = 2.0
    robust_type: str = "NONE"
    robust_default_scale: float = 10.0
    genz_planarity_threshold: float = 0.2
    optimization_method: str = "GN"  # GN | LM
    gn_lambda: float = 1.0
    lm_max_inner_iterations: int = 10
    lm_lambda_factor: float = 2.0
    lm_init_lambda: float = 1.0
    lm_max_lambda: float = 1e3
    lm_min_lambda: float = 1e-6
    dogleg_initial_trust_region_radius: float = 1.0  # registration_params.hpp:84-92
    dogleg_min_trust_region_radius: float = 1e-4
    dogleg_max_trust_region_radius: float = 10.0
    dogleg_eta1: float = 0.25
    dogleg_eta2: float = 0.75
    dogleg_gamma_decrease: float = 0.25
    dogleg_gamma_increase: float = 2.0
    max_iterations: int = 20
    criteria_translation: float = 1e-3
    criteria_rotation: float = 1e-3
    verbose: bool = False
    # default-off terms
    rotation_constraint_enable: bool = False  # registration_params.hpp:56-64
    rotation_constraint_weight: float = 1.0
    rotation_constraint_robust_default_scale: float = 10.0
    degenerate_reg_type: str = "NONE"  # NONE | NL_REG (degenerate_regularization.hpp:41-46)
    degenerate_reg_rot_eigenvalue_threshold: float = 10.0
    degenerate_reg_trans_eigenvalue_threshold: float = 1.0
    degenerate_reg_base_factor: float = 1.0
    map_prior_enabled: bool = False  # map_prior.hpp:14-35
    map_prior_rot_vel_sigma: float = 1.0
    map_prior_trans_vel_sigma: float = 1.0
    map_prior_rot_base_sigma: float = 3.16e-2
    map_prior_trans_base_sigma: float = 1e-2


@dataclass
class RegistrationResult:
    """result.hpp:12-28"""
    T: np.ndarray = field(default_factory=identity)
    converged: bool = False
    iterations: int = 0
    H: np.ndarray = field(default_factory=lambda: np.zeros((6, 6), np.float32))
    b: np.ndarray = field(default_factory=lambda: np.zeros(6, np.float32))
    error: float = FLT_MAX
    inlier: int = 0
    H_raw: np.ndarray = field(default_factory=lambda: np.zeros((6, 6), np.float32))  # before regularisation / prior
    b_raw: np.ndarray = field(default_factory=lambda: np.zeros(6, np.float32))
    error_raw: float = FLT_MAX


class Registration:
    """algorithms/registration/registration.hpp:88-965 for the GN / LM optimisers.

    align() is the reference's host-driven loop (one 192-byte read-back per iteration).
    align_device_loop() keeps pose, linear system and solve on the device for a fixed number of iterations — the
    form the benchmark times — and takes an optional torch.distributed process group: each rank linearises its shard
    of the source and the 176-byte system is summed over ranks (RCCL over xGMI) before the solve.
    """

    def __init__(self, params=None):
        self.params = params or RegistrationParams()
        self.neighbors = KNNResult()
        self._ws = None
        self._lin = None
        self.genz_alpha = 1.0
        self._rot_scale = self.params.rotation_constraint_robust_default_scale
        self._map_prior = MapPriorState()
        self._psrc = None
        self._source_options = {}  # csrc/sp_internal.h switches applied to the prepared source (tests / bench only)

    def _set_source_option(self, name, value):
        self._source_options[name] = int(value)
        if self._psrc is not None:
            self._psrc._set_option(name, value)

    def _prepared_source(self, n):
        if self._psrc is None or self._psrc.n_max < n:
            self._psrc = PreparedSource(n)
            for k, v in self._source_options.items():
                self._psrc._set_option(k, v)
            return self._psrc, True
        return self._psrc, False

    def set_map_prior_state(self, prev_result, T_pred):
        """registration.hpp:124-126 / MapPrior::update (map_prior.hpp:97-174): call once per frame, after motion
        prediction and before align(); `prev_result` is the previous frame's RegistrationResult."""
        p = self.params
        mp = MapPriorParams(int(p.map_prior_enabled), p.map_prior_rot_vel_sigma, p.map_prior_trans_vel_sigma,
                            p.map_prior_rot_base_sigma, p.map_prior_trans_base_sigma)
        Hraw = np.ascontiguousarray(prev_result.H_raw, np.float32)
        Tprev, Tpred = _T16(prev_result.T).copy(), _T16(T_pred).copy()
        check(_lib.lib().sp_map_prior_update_host(C.byref(mp), Hraw.ctypes.data_as(C.c_void_p),
                                                  C.c_float(prev_result.error_raw), int(prev_result.inlier),
                                                  Tprev.ctypes.data_as(C.c_void_p), Tpred.ctypes.data_as(C.c_void_p),
                                                  C.byref(self._map_prior)))
        return bool(self._map_prior.has_prior)

    def _prior_error(self, T16):
        if not (self.params.map_prior_enabled and self._map_prior.has_prior):
            return np.float32(0.0)
        return np.float32(_lib.lib().sp_map_prior_apply_host(C.byref(self._map_prior), T16.ctypes.data_as(C.c_void_p),
                                                             None, None, None))

    # -- helpers
    def _buffers(self, device):
        L = _lib.lib()
        if self._ws is None or self._ws.device != device:
            self._ws = torch.empty(L.sp_gicp_workspace_bytes(0), dtype=torch.uint8, device=device)
            self._lin = torch.zeros(48, dtype=torch.float32, device=device)
        return self._ws, self._lin

    def _factor_params(self, robust_scale):
        p = self.params
        return FactorParams(REG[p.reg_type], LOSS[p.robust_type], p.max_correspondence_distance, robust_scale,
                            self.genz_alpha, p.genz_planarity_threshold, int(p.rotation_constraint_enable),
                            p.rotation_constraint_weight, self._rot_scale)

    def _validate(self, source, target):
        p = self.params
        if p.reg_type == "POINT_TO_PLANE" and not target.has_normal():
            if not target.has_cov():
                raise SpError(2, "[Registration::validate_params] Normal vector or covariance matrices of target must "
                                 "be pre-computed before performing Point-to-Plane ICP matching.")
            covariance.extract_normals(target)
        if p.reg_type == "GICP" and (not source.has_cov() or not target.has_cov()):
            raise SpError(2, "[Registration::validate_params] Covariance matrices of source and target must be "
                             "pre-computed before performing GICP matching.")
        if p.reg_type == "GENZ":
            if not target.has_cov():
                raise SpError(2, "[Registration::validate_params] Covariance matrices of target must be pre-computed "
                                 "before performing GenZ-ICP matching.")
            if not target.has_normal():
                covariance.extract_normals(target)
        if p.reg_type == "POINT_TO_DISTRIBUTION" and not target.has_cov():
            raise SpError(2, "[Registration::validate_params] Covariance matrices of target must be pre-computed "
                             "before performing Point-to-Distribution ICP matching.")
        if p.rotation_constraint_enable and not source.has_cov():
            raise SpError(2, "[Registration::validate_params] Covariance matrices of source are required for "
                             "performing rotation constraint matching.")
        if p.rotation_constraint_enable and not target.has_cov():
            raise SpError(2, "[Registration::validate_params] Covariance matrices of target are required for "
                             "performing rotation constraint matching.")
        if p.robust_type != "NONE" and p.robust_default_scale <= 0.0:
            p.robust_type = "NONE"

    def _linearize(self, which, source, target, transT, robust_scale, lin):
        L = _lib.lib()
        ws, _ = self._buffers(source.points.device)
        fp = self._factor_params(robust_scale)
        tp, on_dev, keep = _trans_arg(transT)
        fn = L.sp_gicp_linearize if which == "linearize" else L.sp_gicp_error
        check(fn(_ptr(source.points), _ptr(source.covs if source.has_cov() else None), source.size(),
                 _ptr(target.points), _ptr(target.covs if target.has_cov() else None),
                 _ptr(target.normals if target.has_normal() else None), _ptr(self.neighbors.indices),
                 _ptr(self.neighbors.distances), tp, on_dev, C.byref(fp), _ptr(lin), _ptr(ws), ws.numel(), _stream()))

    def _genz_alpha(self, source, target):
        cnt = torch.zeros(2, dtype=torch.int32, device=source.points.device)
        check(_lib.lib().sp_genz_counts(_ptr(target.covs), _ptr(self.neighbors.indices), _ptr(self.neighbors.distances),
                                        source.size(), self.params.max_correspondence_distance,
                                        self.params.genz_planarity_threshold, _ptr(cnt), _stream()))
        inl, pl = [int(x) for x in cnt.cpu().tolist()]
        return 1.0 if inl == 0 else float(np.float32(pl) / np.float32(inl))

    @staticmethod
    def _read_lin(lin):
        h = lin.cpu().numpy()
        out = Linearized()
        C.memmove(C.byref(out), h.ctypes.data, 192)
        return out

    def compute_linearized_result(self, source, target, target_knn, pose, robust_scale=-1.0, rotation_robust_scale=-1.0,
                                  initial_pose=None):
        """registration.hpp:312-331; with `initial_pose` the degenerate regularisation is applied (:323)."""
        scale = robust_scale if robust_scale > 0 else self.params.robust_default_scale
        self._rot_scale = (rotation_robust_scale if rotation_robust_scale > 0
                           else self.params.rotation_constraint_robust_default_scale)
        _, lin = self._buffers(source.points.device)
        target_knn.nearest_neighbor_search_async(source, self.neighbors, pose)
        if self.params.reg_type == "GENZ":
            self.genz_alpha = self._genz_alpha(source, target)
        self._linearize("linearize", source, target, pose, scale, lin)
        r = self._read_lin(lin)
        p = self.params
        if initial_pose is not None and p.degenerate_reg_type.upper() != "NONE":
            dreg = DegenerateRegParams(1, p.degenerate_reg_rot_eigenvalue_threshold,
                                       p.degenerate_reg_trans_eigenvalue_threshold, p.degenerate_reg_base_factor)
            Tc, Ti = _T16(pose).copy(), _T16(initial_pose).copy()
            check(_lib.lib().sp_degenerate_regularize_host(C.byref(dreg), r.H, r.b, r.inlier,
                                                           Tc.ctypes.data_as(C.c_void_p), Ti.ctypes.data_as(C.c_void_p)))
        return {"H": np.array(r.H, np.float32).reshape(6, 6), "b": np.array(r.b, np.float32), "error": float(r.error),
                "inlier": int(r.inlier)}

    def compute_error_frozen(self, source, target, pose, robust_scale=-1.0, rotation_robust_scale=-1.0):
        """registration.hpp:350-359 — error at `pose` with the cached correspondences."""
        scale = robust_scale if robust_scale > 0 else self.params.robust_default_scale
        self._rot_scale = (rotation_robust_scale if rotation_robust_scale > 0
                           else self.params.rotation_constraint_robust_default_scale)
        _, lin = self._buffers(source.points.device)
        self._linearize("error", source, target, pose, scale, lin)
        r = self._read_lin(lin)
        return float(r.error), int(r.inlier)

    def compute_icp_robust_weights(self, source, target, target_knn, pose, robust_scale):
        """registration.hpp:279-294"""
        out = torch.zeros(source.size(), dtype=torch.float32, device=source.points.device)
        if source.size() == 0:
            return out
        target_knn.nearest_neighbor_search_async(source, self.neighbors, pose)
        fp = self._factor_params(robust_scale)
        tp, on_dev, keep = _trans_arg(pose)
        check(_lib.lib().sp_icp_robust_weights(
            _ptr(source.points), _ptr(source.covs if source.has_cov() else None), source.size(), _ptr(target.points),
            _ptr(target.covs if target.has_cov() else None), _ptr(target.normals if target.has_normal() else None),
            _ptr(self.neighbors.indices), _ptr(self.neighbors.distances), tp, on_dev, C.byref(fp), _ptr(out), _stream()))
        return out

    # -- the reference's host-driven loop
    def align(self, source, target, target_knn, initial_guess=None, robust_scale=-1.0, rotation_robust_scale=-1.0):
        """registration.hpp:201-276 with optimize_gauss_newton (:803-828) / optimize_levenberg_marquardt (:830-895) /
        optimize_powell_dogleg (:897-965); degenerate regularisation and the MAP prior act on the reduced system
        between the device reduction and the solve (:236-253)."""
        L = _lib.lib()
        p = self.params
        result = RegistrationResult()
        result.T = identity() if initial_guess is None else np.array(initial_guess, np.float32)
        if source.size() == 0:
            return result
        self._validate(source, target)
        scale = robust_scale if robust_scale > 0 else p.robust_default_scale
        self._rot_scale = (rotation_robust_scale if rotation_robust_scale > 0
                           else p.rotation_constraint_robust_default_scale)
        _, lin = self._buffers(source.points.device)
        T = _T16(result.T).copy().reshape(-1)  # column-major working copy

        def linearize_at(Tcol):
            Tmat = Tcol.reshape(4, 4).T
            target_knn.nearest_neighbor_search_async(source, self.neighbors, Tmat)
            if p.reg_type == "GENZ":
                self.genz_alpha = self._genz_alpha(source, target)
            self._linearize("linearize", source, target, Tmat, scale, lin)
            return self._read_lin(lin)  # the reference's wait_and_throw + toCPU (registration.hpp:674-675)

        def error_at(Tcol, Tlin_col):
            return self.compute_error_frozen(source, target, Tcol.reshape(4, 4).T, scale, self._rot_scale)

        return self._host_loop(result, T, linearize_at, error_at, scale)

    def opt_params(self):
        """sp_opt_params of these RegistrationParams."""
        p = self.params
        return _lib.OptParams({"GN": 0, "LM": 1, "DOGLEG": 2}[p.optimization_method], p.max_iterations, p.criteria_rotation,
                              p.criteria_translation, p.gn_lambda, p.lm_max_inner_iterations, p.lm_lambda_factor,
                              p.lm_init_lambda, p.lm_max_lambda, p.lm_min_lambda, p.dogleg_initial_trust_region_radius,
                              p.dogleg_min_trust_region_radius, p.dogleg_max_trust_region_radius, p.dogleg_eta1, p.dogleg_eta2,
                              p.dogleg_gamma_decrease, p.dogleg_gamma_increase)

    def align_optimize(self, source, prepared_target, initial_guess=None, robust_scales=None, sort_by_cell=True, prepare=True,
                       enqueue_only=False):
        """Registration::align for every optimiser — and pipeline::RobustAligner's annealing levels around it (robust_scales:
        one align() per entry, each from the pose the previous one ended on) — as ONE launch and ONE read-back
        (sp_gicp_align_optimize). Returns a RegistrationResult that also carries `log` (one entry per outer iteration:
        level, iteration, trials, accepted, damping, error), `linearizations`, `trials`, `searched`; None when the launch is
        not available on this device right now or its bounded wait ran out (the caller takes its per-step loop).
        enqueue_only: returns the device tensor of the sp_align_result instead (nothing synchronised)."""
        L = _lib.lib()
        p = self.params
        n = source.size()
        dev = source.points.device
        ws, _ = self._buffers(dev)
        T0 = identity() if initial_guess is None else np.asarray(initial_guess, np.float32)
        if getattr(self, "_opt_T_dev", None) is None or self._opt_T_dev.device != dev:
            self._opt_T_dev = torch.zeros(16, dtype=torch.float32, device=dev)
            self._opt_res_dev = torch.zeros(C.sizeof(_lib.AlignResult), dtype=torch.uint8, device=dev)
            self._opt_res_host = torch.zeros(C.sizeof(_lib.AlignResult), dtype=torch.uint8).pin_memory()
        self._opt_T_dev.copy_(torch.from_numpy(_T16(T0).reshape(-1).copy()), non_blocking=True)
        psrc, _ = self._prepared_source(n)
        if prepare:
            psrc.prepare(prepared_target, source, self._opt_T_dev, sort_by_cell)
        scales = list(robust_scales) if robust_scales else [p.robust_default_scale]
        fp = self._factor_params(scales[0])
        op = self.opt_params()
        sc = (C.c_float * len(scales))(*scales)
        rc = L.sp_gicp_align_optimize(prepared_target._h, psrc._h, _ptr(self._opt_T_dev), C.byref(fp), C.byref(op), sc, len(scales),
                                      _ptr(self._opt_res_dev), _ptr(ws), ws.numel(), _stream())
        if rc == _lib.SP_ERR_RUNTIME and b"not available" in L.sp_last_error():
            return None
        check(rc)
        if enqueue_only:
            return self._opt_res_dev
        self._opt_res_host.copy_(self._opt_res_dev, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        r = _lib.AlignResult.from_buffer_copy(self._opt_res_host.numpy().tobytes())
        if r.status != 0:
            return None
        return self._result_from_align_result(r)

    @staticmethod
    def _result_from_align_result(r):
        res = RegistrationResult()
        res.T = np.array(r.T, np.float32).reshape(4, 4).T.copy()
        res.converged, res.iterations = bool(r.converged), int(r.iterations)
        res.H = np.array(r.H, np.float32).reshape(6, 6)
        res.b = np.array(r.b, np.float32)
        res.error, res.inlier = float(r.error), int(r.inlier)
        res.H_raw, res.b_raw, res.error_raw = res.H.copy(), res.b.copy(), float(r.error_raw)
        res.T_lin = np.array(r.T_lin, np.float32).reshape(4, 4).T.copy()
        res.linearizations, res.trials, res.searched, res.damping = int(r.linearizations), int(r.trials), int(r.searched), float(r.damping)
        res.log = [dict(level=e.level, iteration=e.iteration, trials=e.trials, accepted=e.accepted, damping=e.damping, error=e.error)
                   for e in r.log[:r.log_entries]]
        return res

    def align_prepared(self, source, prepared_target, initial_guess=None, robust_scale=-1.0, sort_by_cell=True,
                       device_resident=True):
        """Registration::align (registration.hpp:201-276) with every optimiser (GN / LM / DOGLEG) on the PREPARED path:
        each outer iteration linearises with the fused kernel (sp_gicp_iteration_fused: search or certified reuse +
        linearise + reduce), and the trial steps of LM (:830-895) and dog-leg (:897-965) evaluate the frozen-correspondence
        error with sp_gicp_error_prepared — a coalesced stream over the correspondence cache instead of the generic K12's
        gathers and two eigen-decompositions per point. GICP and POINT_TO_DISTRIBUTION (the prepared target's reg_type)."""
        L = _lib.lib()
        p = self.params
        result = RegistrationResult()
        result.T = identity() if initial_guess is None else np.array(initial_guess, np.float32)
        n = source.size()
        if n == 0:
            return result
        if p.reg_type not in ("GICP", "POINT_TO_DISTRIBUTION") or p.rotation_constraint_enable:
            raise SpError(1, "align_prepared: GICP / POINT_TO_DISTRIBUTION without the rotation constraint")
        if prepared_target.reg_type != p.reg_type:
            raise SpError(1, "align_prepared: the prepared target holds the rows of another reg_type")
        if p.robust_type != "NONE" and p.robust_default_scale <= 0.0:
            p.robust_type = "NONE"
        scale = robust_scale if robust_scale > 0 else p.robust_default_scale
        ws, lin = self._buffers(source.points.device)
        T = _T16(result.T).copy().reshape(-1)
        pose_terms = (p.degenerate_reg_type.upper() != "NONE") or (p.map_prior_enabled and bool(self._map_prior.has_prior))
        if device_resident and not pose_terms and p.max_iterations > 0:
            # the whole optimiser loop as one launch, one read-back (sp_gicp_align_optimize); the host loop below is what is
            # left for the host-side pose terms and for a device that cannot hold the launch resident
            res = self.align_optimize(source, prepared_target, result.T, [scale], sort_by_cell)
            if res is not None:
                return res
        psrc, _ = self._prepared_source(n)
        psrc.prepare(prepared_target, source, result.T, sort_by_cell)
        fp = self._factor_params(scale)

        def linearize_at(Tcol):
            check(L.sp_gicp_iteration_fused(prepared_target._h, psrc._h, Tcol.ctypes.data_as(C.c_void_p), 0, C.byref(fp), None,
                                            None, None, _ptr(lin), None, _ptr(ws), ws.numel(), _stream()))
            return self._read_lin(lin)

        def error_at(Tcol, Tlin_col):
            check(L.sp_gicp_error_prepared(prepared_target._h, psrc._h, Tlin_col.ctypes.data_as(C.c_void_p),
                                           Tcol.ctypes.data_as(C.c_void_p), 0, C.byref(fp), _ptr(lin), _ptr(ws), ws.numel(),
                                           _stream()))
            r = self._read_lin(lin)
            return float(r.error), int(r.inlier)

        return self._host_loop(result, T, linearize_at, error_at, scale)

    def _host_loop(self, result, T, linearize_at, error_at, scale):
        """The optimiser loop of Registration::align shared by the generic and the prepared path: `linearize_at(T)` returns
        the reduced system at pose T (column-major 16 floats), `error_at(T_trial, T_lin)` the frozen-correspondence error."""
        L = _lib.lib()
        p = self.params
        lm_lambda = p.lm_init_lambda
        radius = np.float32(p.dogleg_initial_trust_region_radius)
        T_initial = T.copy()
        delta8 = np.zeros(8, np.float32)
        dreg = DegenerateRegParams({"NONE": 0, "NL_REG": 1, "NL-REG": 1}[p.degenerate_reg_type.upper()],
                                   p.degenerate_reg_rot_eigenvalue_threshold,
                                   p.degenerate_reg_trans_eigenvalue_threshold, p.degenerate_reg_base_factor)
        prior_on = p.map_prior_enabled and bool(self._map_prior.has_prior)
        for it in range(p.max_iterations):
            lr = linearize_at(T)
            result.H_raw = np.array(lr.H, np.float32).reshape(6, 6)  # registration.hpp:244-246
            result.b_raw = np.array(lr.b, np.float32)
            result.error_raw = float(lr.error)
            if dreg.type:  # registration.hpp:249-250
                check(L.sp_degenerate_regularize_host(C.byref(dreg), lr.H, lr.b, lr.inlier,
                                                      T.ctypes.data_as(C.c_void_p), T_initial.ctypes.data_as(C.c_void_p)))
            if prior_on:  # registration.hpp:253
                err = C.c_float(lr.error)
                L.sp_map_prior_apply_host(C.byref(self._map_prior), T.ctypes.data_as(C.c_void_p), lr.H, lr.b, C.byref(err))
                lr.error = err.value
            H = np.array(lr.H, np.float32).reshape(6, 6)
            b = np.array(lr.b, np.float32)
            if p.optimization_method == "GN":
                L.sp_gn_update_host(C.byref(lr), T.ctypes.data_as(C.c_void_p), p.gn_lambda, p.criteria_rotation,
                                    p.criteria_translation, delta8.ctypes.data_as(C.c_void_p))
                result.converged = bool(delta8[6] > 0.5)
                result.iterations, result.H, result.b = it, H, b
                result.error, result.inlier = float(lr.error), int(lr.inlier)
            elif p.optimization_method == "DOGLEG":  # optimize_powell_dogleg (registration.hpp:897-965)
                f32 = np.float32
                result.iterations, result.H, result.b = it, H, b
                result.error, result.inlier = float(lr.error), int(lr.inlier)
                clamp = lambda r: f32(min(max(r, f32(p.dogleg_min_trust_region_radius)),  # noqa: E731
                                          f32(p.dogleg_max_trust_region_radius)))
                radius = clamp(radius)
                Hrow = np.ascontiguousarray(H)
                p6, sn, pred = np.zeros(6, np.float32), C.c_float(0.0), C.c_float(0.0)
                L.sp_dogleg_step_host(Hrow.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), C.c_float(radius),
                                      p6.ctypes.data_as(C.c_void_p), C.byref(sn), C.byref(pred))
                if pred.value <= 0.0:
                    radius = clamp(f32(radius * f32(p.dogleg_gamma_decrease)))
                else:
                    E, Ttry = np.zeros(16, np.float32), np.zeros(16, np.float32)
                    L.sp_se3_exp_host(p6.ctypes.data_as(C.c_void_p), E.ctypes.data_as(C.c_void_p))
                    L.sp_rigid_mul_host(T.ctypes.data_as(C.c_void_p), E.ctypes.data_as(C.c_void_p),
                                        Ttry.ctypes.data_as(C.c_void_p))
                    new_error, inl = error_at(Ttry, T)
                    new_error = f32(f32(new_error) + self._prior_error(Ttry))  # registration.hpp:933
                    rho = f32(f32(lr.error) - f32(new_error)) / f32(pred.value)
                    if rho < p.dogleg_eta1:
                        radius = clamp(f32(radius * f32(p.dogleg_gamma_decrease)))
                    else:
                        nr = np.sqrt(f32(p6[0] * p6[0] + p6[1] * p6[1] + p6[2] * p6[2]))
                        nt = np.sqrt(f32(p6[3] * p6[3] + p6[4] * p6[4] + p6[5] * p6[5]))
                        result.converged = bool(nr < p.criteria_rotation and nt < p.criteria_translation)
                        T = Ttry
                        result.error, result.inlier = float(new_error), inl
                        if rho > p.dogleg_eta2 and sn.value >= radius * f32(0.99):
                            radius = clamp(f32(radius * f32(p.dogleg_gamma_increase)))
            else:  # LM
                current_error = np.float32(lr.error)
                last_error = np.float32(FLT_MAX)
                for _inner in range(p.lm_max_inner_iterations):
                    Ttry = T.copy()
                    L.sp_gn_update_host(C.byref(lr), Ttry.ctypes.data_as(C.c_void_p), lm_lambda, p.criteria_rotation,
                                        p.criteria_translation, delta8.ctypes.data_as(C.c_void_p))
                    conv = bool(delta8[6] > 0.5)
                    result.converged = conv
                    new_error, inl = error_at(Ttry, T)
                    new_error = np.float32(np.float32(new_error) + self._prior_error(Ttry))  # registration.hpp:854
                    if new_error <= current_error:
                        result.converged, T = conv, Ttry
                        result.error, result.inlier = float(new_error), inl
                        lm_lambda = float(np.clip(np.float32(lm_lambda) / np.float32(p.lm_lambda_factor),
                                                  p.lm_min_lambda, p.lm_max_lambda))
                        break
                    elif abs(np.float32(new_error - last_error)) <= 1e-6:
                        result.converged, T = conv, Ttry
                        result.error, result.inlier = float(new_error), inl
                        break
                    else:
                        lm_lambda = float(np.clip(np.float32(lm_lambda) * np.float32(p.lm_lambda_factor),
                                                  p.lm_min_lambda, p.lm_max_lambda))
                    last_error = new_error
                result.iterations, result.H, result.b = it, H, b
            if result.converged:
                break
        result.T = T.reshape(4, 4).T.copy()
        return result

    # -- device-resident fixed-length loop (GN), optionally sharded over ranks
    def align_device_loop(self, source, target, target_knn, initial_guess=None, iterations=None, robust_scale=-1.0,
                          group=None, T_dev=None, delta_dev=None):
        """`iterations` Gauss-Newton steps with no host round trip: NN(k=1) -> K11 -> [all-reduce] -> device solve.
        Returns the device tensors (T_dev 16 floats column-major, lin 48 floats, delta 8 floats); nothing is
        synchronised here."""
        import torch.distributed as dist

        L = _lib.lib()
        p = self.params
        if p.optimization_method != "GN":
            raise SpError(1, "align_device_loop implements the Gauss-Newton optimiser only")
        self._validate(source, target)
        iters = p.max_iterations if iterations is None else iterations
        scale = robust_scale if robust_scale > 0 else p.robust_default_scale
        dev = source.points.device
        _, lin = self._buffers(dev)
        if T_dev is None:
            T0 = identity() if initial_guess is None else np.asarray(initial_guess, np.float32)
            T_dev = torch.from_numpy(_T16(T0).reshape(-1).copy()).to(dev)
        if delta_dev is None:
            delta_dev = torch.zeros(8, dtype=torch.float32, device=dev)
        sharded = group is not None or (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)
        for _ in range(iters):
            target_knn.nearest_neighbor_search_async(source, self.neighbors, T_dev)
            self._linearize("linearize", source, target, T_dev, scale, lin)
            if sharded:
                dist.all_reduce(lin, op=dist.ReduceOp.SUM, group=group)  # 192 B over xGMI; latency-bound
            check(L.sp_gn_update(_ptr(lin), _ptr(T_dev), p.gn_lambda, p.criteria_rotation, p.criteria_translation,
                                 _ptr(delta_dev), _stream()))
        return T_dev, lin, delta_dev

    def align_fused_loop(self, source, prepared_target, initial_guess=None, iterations=None, robust_scale=-1.0,
                         group=None, T_dev=None, delta_dev=None, prepare=True, sort_by_cell=True,
                         write_neighbors=False, per_iteration_launches=False, update_target=False, graph=False,
                         comm=None, exchange="row", xchg=None):
        """The same fixed-length Gauss-Newton loop as align_device_loop on the prepared / fused path
        (sp_gicp_iteration_fused): one launch per iteration does NN + linearise + reduce, and, on a single GPU, the
        second (one-workgroup) launch also solves and updates the pose. On one GPU the default is
        sp_gicp_align_fused: the reduction + solve of iteration k-1 runs as the prologue of launch k, and once the
        convergence criteria hold the remaining launches return immediately (self._iters_dev holds the number of
        steps applied); per_iteration_launches=True keeps the two-launch fixed-length form. The prepared target is
        target-side pre-processing like its grid (built once by PreparedTarget(...)): pass update_target=True, or call
        prepared_target.update(covs), when the target's covariances have changed since. With prepare=True (a new alignment) the
        per-alignment preparation — plane regularisation of both clouds' covariances and the cell-order sort of the
        source at the initial pose — is enqueued first."""
        import torch.distributed as dist

        L = _lib.lib()
        p = self.params
        if p.reg_type not in ("GICP", "POINT_TO_DISTRIBUTION") or p.optimization_method != "GN":
            raise SpError(1, "align_fused_loop implements GICP / POINT_TO_DISTRIBUTION with the Gauss-Newton optimiser")
        iters = p.max_iterations if iterations is None else iterations
        scale = robust_scale if robust_scale > 0 else p.robust_default_scale
        dev = source.points.device
        ws, lin = self._buffers(dev)
        n = source.size()
        if T_dev is None:
            T0 = identity() if initial_guess is None else np.asarray(initial_guess, np.float32)
            T_dev = torch.from_numpy(_T16(T0).reshape(-1).copy()).to(dev)
        if delta_dev is None:
            delta_dev = torch.zeros(8, dtype=torch.float32, device=dev)
        if self._prepared_source(n)[1]:
            prepare = True
        if prepare:
            if update_target:
                prepared_target.update()
            self._psrc.prepare(prepared_target, source, T_dev, sort_by_cell)
        sharded = (comm is not None or group is not None or
                   (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1))
        fp = self._factor_params(scale)
        gn = GnParams(p.gn_lambda, p.criteria_rotation, p.criteria_translation)
        if xchg is not None:
            # rows exchanged directly between the ranks' buffers (sp_gicp_align_direct): one C call, no collective
            if getattr(self, "_iters_dev", None) is None or self._iters_dev.device != dev:
                self._iters_dev = torch.zeros(1, dtype=torch.int32, device=dev)
            check(L.sp_gicp_align_direct(prepared_target._h, self._psrc._h, _ptr(T_dev), C.byref(fp),
                                         C.byref(gn), iters,
                                         xchg._h, None, None, _ptr(lin), _ptr(delta_dev), _ptr(self._iters_dev), _ptr(ws),
                                         ws.numel(), _stream()))
            self._direct_last_k = iters - 1
            return T_dev, lin, delta_dev
        if write_neighbors:
            self.neighbors.resize(n, 1, dev)
        ni = _ptr(self.neighbors.indices) if write_neighbors else None
        nd = _ptr(self.neighbors.distances) if write_neighbors else None
        if not sharded and not per_iteration_launches:
            # one C call enqueues the whole loop: one launch per iteration (+ one to finish), convergence on the device
            if getattr(self, "_iters_dev", None) is None or self._iters_dev.device != dev:
                self._iters_dev = torch.zeros(1, dtype=torch.int32, device=dev)
            check(L.sp_gicp_align_fused(prepared_target._h, self._psrc._h, _ptr(T_dev), C.byref(fp), C.byref(gn), iters,
                                        ni, nd, _ptr(lin), _ptr(delta_dev), _ptr(self._iters_dev), _ptr(ws), ws.numel(),
                                        _stream()))
            return T_dev, lin, delta_dev
        if sharded and not per_iteration_launches:
            # one launch + one collective per iteration. exchange="row" (default): the launch reduces its partial rows to
            # ONE 128-byte row inside the kernel (fan-in, rows_all_reduced = 2) and only that row is all-reduced;
            # exchange="rows": the 32 KB of partial rows are all-reduced (round 1's form, kept for comparison). Launch k+1's
            # prologue then finishes iteration k identically on every rank. With `comm` (the library's own RCCL
            # communicator) the whole loop is ONE C call, sp_gicp_align_sharded.
            if getattr(self, "_iters_dev", None) is None or self._iters_dev.device != dev:
                self._iters_dev = torch.zeros(1, dtype=torch.int32, device=dev)
            mode = 2 if (exchange == "row" or comm is not None) else 1
            nf = C.c_size_t(0)
            wsf = ws.view(torch.float32)
            base = ws.data_ptr()
            rows = []
            for k in (0, 1):
                fn = L.sp_gicp_align_row if mode == 2 else L.sp_gicp_align_rows
                off = (fn(_ptr(ws), k, C.byref(nf)) - base) // 4
                rows.append(wsf[off:off + nf.value])
            wsp, linp, Tp = _ptr(ws), _ptr(lin), _ptr(T_dev)

            def enqueue():
                st = _stream()
                if comm is not None:
                    check(L.sp_gicp_align_sharded(prepared_target._h, self._psrc._h, Tp, C.byref(fp), C.byref(gn), iters,
                                                  comm._h, ni, nd, linp, _ptr(delta_dev), _ptr(self._iters_dev), wsp,
                                                  ws.numel(), st))
                    return
                for k in range(iters):
                    check(L.sp_gicp_align_step(prepared_target._h, self._psrc._h, Tp, C.byref(fp), C.byref(gn), k, mode, ni, nd,
                                               linp, wsp, ws.numel(), st))
                    dist.all_reduce(rows[k & 1], op=dist.ReduceOp.SUM, group=group)
                if iters > 0:
                    check(L.sp_gicp_align_finish(self._psrc._h, _ptr(T_dev), C.byref(gn), iters - 1, mode, _ptr(lin),
                                                 _ptr(delta_dev), _ptr(self._iters_dev), _ptr(ws), ws.numel(), st))

            if not graph:
                enqueue()
                return T_dev, lin, delta_dev
            # The loop touches fixed buffers only (pose, partial rows, state), so launches + collectives of a whole
            # alignment are captured once into a hipGraph and replayed: the host then issues one launch per alignment
            # instead of 2 x iterations calls, and the GPU-side chain (kernel -> all-reduce -> kernel) is what is left.
            key = (iters, prepared_target._h.value if hasattr(prepared_target._h, "value") else id(prepared_target),
                   id(self._psrc), T_dev.data_ptr(), delta_dev.data_ptr(), ws.data_ptr(), lin.data_ptr(), ni, nd, n,
                   scale, id(group), id(comm), mode)
            graphs = self.__dict__.setdefault("_loop_graphs", {})
            g = graphs.get(key)
            if g is None:
                enqueue()  # first alignment of this shape runs eagerly (RCCL warm-up), the second one is captured
                graphs[key] = "warm"
            elif g == "warm":
                try:
                    torch.cuda.synchronize()
                    cg = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(cg):
                        enqueue()
                    graphs[key] = cg
                    cg.replay()
                except Exception as e:  # capture not supported for this collective / runtime: stay on eager launches
                    graphs[key] = False
                    import warnings
                    warnings.warn(f"align_fused_loop: hipGraph capture failed ({e!r}); using per-call launches")
                    torch.cuda.synchronize()
                    enqueue()
            elif g is False:
                enqueue()
            else:
                g.replay()
            return T_dev, lin, delta_dev
        for _ in range(iters):
            check(L.sp_gicp_iteration_fused(prepared_target._h, self._psrc._h, _ptr(T_dev), 1, C.byref(fp),
                                            None if sharded else C.byref(gn), ni, nd, _ptr(lin), _ptr(delta_dev),
                                            _ptr(ws), ws.numel(), _stream()))
            if sharded:
                dist.all_reduce(lin, op=dist.ReduceOp.SUM, group=group)
                check(L.sp_gn_update(_ptr(lin), _ptr(T_dev), p.gn_lambda, p.criteria_rotation, p.criteria_translation,
                                     _ptr(delta_dev), _stream()))
        return T_dev, lin, delta_dev

    def direct_status(self):
        """sp_gicp_align_status after align_fused_loop(..., xchg=...): raises SpError when a peer's row did not arrive."""
        ws, _ = self._buffers(torch.device("cuda", torch.cuda.current_device()))
        check(_lib.lib().sp_gicp_align_status(_ptr(ws), int(self._direct_last_k), _stream()))

    @staticmethod
    def T_from_device(T_dev):
        return T_dev.cpu().numpy().reshape(4, 4).T.copy()
