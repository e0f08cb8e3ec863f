"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes binding of oracle/liboracle.so (the CPU restatement of the reference's hot path).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product package (sycl_points_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")

REG = {"POINT_TO_POINT": 0, "POINT_TO_PLANE": 1, "POINT_TO_DISTRIBUTION": 2, "GICP": 3, "GENZ": 4}
LOSS = {"NONE": 0, "HUBER": 1, "TUKEY": 2, "CAUCHY": 3, "GEMAN_MCCLURE": 4}
OPT = {"GN": 0, "LM": 1, "DOGLEG": 2}


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".cpp", ".hpp"))]
    stale = (not os.path.exists(_LIB)) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


class RegParams(C.Structure):
    _fields_ = [
        ("reg_type", C.c_int), ("robust_type", C.c_int), ("optimization_method", C.c_int), ("max_iterations", C.c_int),
        ("max_correspondence_distance", C.c_float), ("robust_default_scale", C.c_float), ("gn_lambda", C.c_float),
        ("lm_init_lambda", C.c_float), ("lm_lambda_factor", C.c_float), ("lm_min_lambda", C.c_float),
        ("lm_max_lambda", C.c_float), ("lm_max_inner_iterations", C.c_int),
        ("crit_translation", C.c_float), ("crit_rotation", C.c_float),
        ("auto_scale", C.c_int), ("auto_scaling_iter", C.c_int), ("init_scale", C.c_float), ("min_scale", C.c_float),
        ("dl_initial_radius", C.c_float), ("dl_min_radius", C.c_float), ("dl_max_radius", C.c_float),
        ("dl_eta1", C.c_float), ("dl_eta2", C.c_float), ("dl_gamma_decrease", C.c_float), ("dl_gamma_increase", C.c_float),
        # default-off terms: rotation constraint, degenerate regularisation (NL-Reg), MAP prior (see map_prior_update)
        ("rot_enable", C.c_int), ("rot_weight", C.c_float), ("rot_robust_default_scale", C.c_float),
        ("dr_type", C.c_int), ("dr_rot_threshold", C.c_float), ("dr_trans_threshold", C.c_float),
        ("dr_base_factor", C.c_float),
        ("mp_active", C.c_int), ("mp_omega", C.c_float * 36), ("mp_T_pred_inv", C.c_float * 16),
    ]

    @staticmethod
    def defaults(**kw):
        # registration_params.hpp:46-114 defaults
        p = RegParams(reg_type=REG["GICP"], robust_type=LOSS["NONE"], optimization_method=OPT["GN"], max_iterations=20,
                      max_correspondence_distance=2.0, robust_default_scale=10.0, gn_lambda=1.0,
                      lm_init_lambda=1.0, lm_lambda_factor=2.0, lm_min_lambda=1e-6, lm_max_lambda=1e3,
                      lm_max_inner_iterations=10, crit_translation=1e-3, crit_rotation=1e-3,
                      auto_scale=0, auto_scaling_iter=4, init_scale=10.0, min_scale=0.5,
                      dl_initial_radius=1.0, dl_min_radius=1e-4, dl_max_radius=10.0, dl_eta1=0.25, dl_eta2=0.75,
                      dl_gamma_decrease=0.25, dl_gamma_increase=2.0,
                      rot_enable=0, rot_weight=1.0, rot_robust_default_scale=10.0,
                      dr_type=0, dr_rot_threshold=10.0, dr_trans_threshold=1.0, dr_base_factor=1.0, mp_active=0)
        for k, v in kw.items():
            if k == "map_prior":  # (omega row-major 6x6, T_pred_inv row-major 4x4) from Oracle.map_prior_update
                om, tinv = v
                p.mp_active = 1
                p.mp_omega = (C.c_float * 36)(*np.asarray(om, np.float32).T.ravel())
                p.mp_T_pred_inv = (C.c_float * 16)(*np.asarray(tinv, np.float32).T.ravel())
                continue
            setattr(p, k, v)
        return p


class RegResult(C.Structure):
    _fields_ = [("T", C.c_float * 16), ("H", C.c_float * 36), ("b", C.c_float * 6), ("error", C.c_float),
                ("inlier", C.c_uint32), ("iterations", C.c_int), ("converged", C.c_int),
                ("H_raw", C.c_float * 36), ("b_raw", C.c_float * 6), ("error_raw", C.c_float)]


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t=C.c_float):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


class Oracle:
    def __init__(self):
        self.lib = C.CDLL(build())
        L = self.lib
        L.orc_rng_new.restype = C.c_void_p
        L.orc_rng_new.argtypes = [C.c_uint32]
        L.orc_rng_free.argtypes = [C.c_void_p]
        L.orc_rng_uniform_points.argtypes = [C.c_void_p, C.c_float, C.c_size_t, C.c_void_p]
        L.orc_rng_normal.argtypes = [C.c_void_p, C.c_float, C.c_size_t, C.c_void_p]
        L.orc_det3.restype = C.c_float
        L.orc_robust_weight.restype = C.c_float
        L.orc_robust_weight.argtypes = [C.c_int, C.c_float, C.c_float]
        L.orc_robust_error.restype = C.c_float
        L.orc_robust_error.argtypes = [C.c_int, C.c_float, C.c_float]
        L.orc_kdtree_build.restype = C.c_size_t
        L.orc_voxel_downsample.restype = C.c_size_t
        L.orc_num_threads.restype = C.c_int

    # ---- rng / synthetic clouds
    def rng(self, seed):
        return _Rng(self, seed)

    def random_sampling_flags(self, seed, n, num):
        flags = np.zeros(n, np.uint8)
        self.lib.orc_random_sampling_flags(C.c_uint32(seed), C.c_size_t(n), C.c_size_t(num), _p(flags, C.c_uint8))
        return flags

    def num_threads(self):
        return self.lib.orc_num_threads()

    def set_num_threads(self, n):
        self.lib.orc_set_num_threads(C.c_int(n))

    # ---- math probes (matrices in/out as numpy row-major [i,j]; converted to column-major storage)
    def eigen3(self, A):
        A = _f(np.asarray(A).T)  # row-major of A^T == column-major of A
        vals = np.zeros(3, np.float32)
        vecs = np.zeros((3, 3), np.float32)
        self.lib.orc_eigen3(_p(A), _p(vals), _p(vecs))
        return vals, vecs.T.copy()

    def inverse3(self, A):
        A = _f(np.asarray(A).T)
        out = np.zeros((3, 3), np.float32)
        self.lib.orc_inverse3(_p(A), _p(out))
        return out.T.copy()

    def det3(self, A):
        A = _f(np.asarray(A).T)
        return float(self.lib.orc_det3(_p(A)))

    _EIGEN_UTIL = {"dot3": (0, ()), "dot4": (1, ()), "cross": (2, (3,)), "outer4": (3, (4, 4)), "transpose33": (4, (3, 3)),
                   "transpose46": (5, (6, 4)), "ensure_symmetric3": (6, (3, 3)), "frobenius_norm33": (7, ()),
                   "frobenius_norm3": (8, ()), "frobenius_norm_squared3": (9, ()), "cwise33": (10, (3, 3)),
                   "cwise44": (11, (4, 4))}

    def eigen_util(self, name, a, b=None):
        """eigen_utils helpers (orc_eigen_util): operands / results as row-major numpy arrays."""
        op, shape = self._EIGEN_UTIL[name]
        col = lambda x: _f(np.asarray(x).T if np.ndim(x) == 2 else np.asarray(x))  # noqa: E731
        a = col(a)
        b = a if b is None else col(b)
        out = np.zeros(max(int(np.prod(shape)), 1), np.float32)
        self.lib.orc_eigen_util(C.c_int(op), _p(a), _p(b), _p(out))
        if shape == ():
            return float(out[0])
        return out.reshape(shape[::-1]).T.copy() if len(shape) == 2 else out.copy()

    def matmul4(self, A, B):
        A = _f(np.asarray(A).T)
        B = _f(np.asarray(B).T)
        out = np.zeros((4, 4), np.float32)
        self.lib.orc_matmul4(_p(A), _p(B), _p(out))
        return out.T.copy()

    def robust_annealing_scales(self, loss, auto_scale, default_scale, init_scale, min_scale, auto_scaling_iter):
        """Scales the annealing wrapper (pipeline/robust.hpp:42-114) hands to the aligner, level by level."""
        out = np.zeros(64, np.float32)
        f = self.lib.orc_robust_annealing_scales
        f.restype = C.c_int
        f.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_int]
        n = f(1 if loss == "NONE" else 0, 1 if auto_scale else 0, default_scale, init_scale, min_scale, int(auto_scaling_iter),
              _p(out), 64)
        return out[:n].copy()

    def se3_exp(self, twist):
        t = _f(twist)
        out = np.zeros((4, 4), np.float32)
        self.lib.orc_se3_exp(_p(t), _p(out))
        return out.T.copy()

    def se3_log(self, T):
        Tc = _f(np.asarray(T).T)
        out = np.zeros(6, np.float32)
        self.lib.orc_se3_log(_p(Tc), _p(out))
        return out

    def so3_exp(self, w):
        w = _f(w)
        out = np.zeros(4, np.float32)
        self.lib.orc_so3_exp(_p(w), _p(out))
        return out

    def so3_log(self, q):
        q = _f(q)
        out = np.zeros(3, np.float32)
        self.lib.orc_so3_log(_p(q), _p(out))
        return out

    def isometry_mul(self, A, B):
        A = _f(np.asarray(A).T)
        B = _f(np.asarray(B).T)
        out = np.zeros((4, 4), np.float32)
        self.lib.orc_isometry_mul(_p(A), _p(B), _p(out))
        return out.T.copy()

    def ldlt6_solve(self, H, b):
        Hc = _f(np.asarray(H).T)
        b = _f(b)
        x = np.zeros(6, np.float32)
        ok = self.lib.orc_ldlt6_solve(_p(Hc), _p(b), _p(x))
        return bool(ok), x

    def robust_weight(self, loss, r, s):
        return float(self.lib.orc_robust_weight(LOSS[loss], r, s))

    def robust_error(self, loss, r, s):
        return float(self.lib.orc_robust_error(LOSS[loss], r, s))

    # ---- knn
    def knn_bruteforce(self, q, t, k):
        # knn/bruteforce.hpp:46-48 keeps `float kDistances[MAX_K = 20]` with no check on k; the restatement keeps the same
        # fixed arrays, so a larger k would overrun the stack here exactly as it does in the reference. Refuse it.
        if not 1 <= k <= 20:
            raise ValueError(f"oracle knn_bruteforce: k = {k} outside the reference's MAX_K = 20 arrays (use kdtree_knn)")
        q, t = _f(q), _f(t)
        idx = np.empty((len(q), k), np.int32)
        d2 = np.empty((len(q), k), np.float32)
        self.lib.orc_knn_bruteforce(_p(q), C.c_size_t(len(q)), _p(t), C.c_size_t(len(t)), C.c_size_t(k),
                                    _p(idx, C.c_int32), _p(d2))
        return idx, d2

    def kdtree_build(self, pts, leaf=16):
        pts = _f(pts)
        nodes = np.zeros(max(2 * len(pts), 1) * 32, np.uint8)
        n = self.lib.orc_kdtree_build(_p(pts), C.c_size_t(len(pts)), C.c_size_t(leaf), nodes.ctypes.data_as(C.c_void_p))
        return nodes[: n * 32].copy()

    def kdtree_knn(self, nodes, q, k, T=None):
        q = _f(q)
        Tc = _f(np.eye(4) if T is None else np.asarray(T).T)
        idx = np.empty((len(q), k), np.int32)
        d2 = np.empty((len(q), k), np.float32)
        rc = self.lib.orc_kdtree_knn(nodes.ctypes.data_as(C.c_void_p), C.c_size_t(len(nodes) // 32), _p(q),
                                     C.c_size_t(len(q)), C.c_size_t(k), _p(Tc), _p(idx, C.c_int32), _p(d2))
        if rc != 0:
            raise RuntimeError("[KDTree::knn_search_async] `k` is too large. not support.")
        return idx, d2

    def kdtree_radius(self, nodes, q, max_k, radius, T=None):
        q = _f(q)
        Tc = _f(np.eye(4) if T is None else np.asarray(T).T)
        idx = np.empty((len(q), max_k), np.int32)
        d2 = np.empty((len(q), max_k), np.float32)
        rc = self.lib.orc_kdtree_radius(nodes.ctypes.data_as(C.c_void_p), C.c_size_t(len(nodes) // 32), _p(q),
                                        C.c_size_t(len(q)), C.c_size_t(max_k), C.c_float(radius), _p(Tc),
                                        _p(idx, C.c_int32), _p(d2))
        if rc != 0:
            raise RuntimeError("[KDTree::radius_search_async] `max_k` is too large. not support.")
        return idx, d2

    def kdtree_remove_by_flags(self, nodes, flags, new_idx):
        flags = np.ascontiguousarray(flags, np.uint8)
        new_idx = np.ascontiguousarray(new_idx, np.int32)
        self.lib.orc_kdtree_remove_by_flags(nodes.ctypes.data_as(C.c_void_p), C.c_size_t(len(nodes) // 32),
                                            _p(flags, C.c_uint8), _p(new_idx, C.c_int32), C.c_size_t(len(flags)))

    # ---- features
    def cov_estimate(self, pts, idx):
        pts = _f(pts)
        idx = np.ascontiguousarray(idx, np.int32)
        covs = np.empty((len(pts), 16), np.float32)
        self.lib.orc_cov_estimate(_p(pts), C.c_size_t(len(pts)), _p(idx, C.c_int32), C.c_size_t(idx.shape[1]), _p(covs))
        return covs

    def cov_estimate_robust(self, pts, idx, robust_type="CAUCHY", mad_scale=1.0, min_robust_scale=1.0, max_iterations=1):
        pts = _f(pts)
        idx = np.ascontiguousarray(idx, np.int32)
        covs = np.empty((len(pts), 16), np.float32)
        self.lib.orc_cov_estimate_robust(_p(pts), C.c_size_t(len(pts)), _p(idx, C.c_int32), C.c_size_t(idx.shape[1]),
                                         C.c_int(LOSS[robust_type]), C.c_float(mad_scale), C.c_float(min_robust_scale),
                                         C.c_size_t(max_iterations), _p(covs))
        return covs

    def cov_normalize(self, covs):
        out = _f(covs).copy()
        self.lib.orc_cov_normalize(_p(out), C.c_size_t(len(out)))
        return out

    def normals_from_knn(self, pts, idx):
        pts = _f(pts)
        idx = np.ascontiguousarray(idx, np.int32)
        out = np.empty((len(pts), 4), np.float32)
        self.lib.orc_normals_from_knn(_p(pts), C.c_size_t(len(pts)), _p(idx, C.c_int32), C.c_size_t(idx.shape[1]), _p(out))
        return out

    def normals_from_cov(self, pts, covs):
        pts, covs = _f(pts), _f(covs)
        out = np.empty((len(pts), 4), np.float32)
        self.lib.orc_normals_from_cov(_p(pts), _p(covs), C.c_size_t(len(pts)), _p(out))
        return out

    def update_covariance_plane(self, covs):
        covs = _f(covs).copy()
        self.lib.orc_update_covariance_plane(_p(covs), C.c_size_t(len(covs)))
        return covs

    def normalize_covariance(self, covs):
        covs = _f(covs).copy()
        self.lib.orc_normalize_covariance(_p(covs), C.c_size_t(len(covs)))
        return covs

    def transform_points(self, pts, T):
        pts = _f(pts)
        Tc = _f(np.asarray(T).T)
        out = np.empty_like(pts)
        self.lib.orc_transform_points(_p(pts), _p(out), C.c_size_t(len(pts)), _p(Tc))
        return out

    def transform_covs(self, covs, T):
        covs = _f(covs)
        Tc = _f(np.asarray(T).T)
        out = np.empty_like(covs)
        self.lib.orc_transform_covs(_p(covs), _p(out), C.c_size_t(len(covs)), _p(Tc))
        return out

    def transform_normals(self, nrm, T):
        nrm = _f(nrm)
        Tc = _f(np.asarray(T).T)
        out = np.empty_like(nrm)
        self.lib.orc_transform_normals(_p(nrm), _p(out), C.c_size_t(len(nrm)), _p(Tc))
        return out

    def voxel_keys(self, pts, voxel_size):
        pts = _f(pts)
        keys = np.empty(len(pts), np.uint64)
        self.lib.orc_voxel_keys(_p(pts), C.c_size_t(len(pts)), C.c_float(voxel_size), _p(keys, C.c_uint64))
        return keys

    def voxel_downsample(self, pts, voxel_size, min_count=1, rgb=None, intensity=None, ts=None, stable=True):
        pts = _f(pts)
        n = len(pts)
        rgb = None if rgb is None else _f(rgb)
        intensity = None if intensity is None else _f(intensity)
        ts = None if ts is None else _f(ts)
        o_p = np.empty((max(n, 1), 4), np.float32)
        o_c = np.empty((max(n, 1), 4), np.float32)
        o_i = np.empty(max(n, 1), np.float32)
        o_t = np.empty(max(n, 1), np.float32)
        o_k = np.empty(max(n, 1), np.uint64)
        v = self.lib.orc_voxel_downsample(_p(pts), C.c_size_t(n), C.c_float(voxel_size), C.c_size_t(min_count), _p(rgb),
                                          _p(intensity), _p(ts), C.c_int(1 if stable else 0), _p(o_p), _p(o_c), _p(o_i),
                                          _p(o_t), _p(o_k, C.c_uint64))
        return {"points": o_p[:v].copy(), "rgb": None if rgb is None else o_c[:v].copy(),
                "intensities": None if intensity is None else o_i[:v].copy(),
                "timestamps": None if ts is None else o_t[:v].copy(), "keys": o_k[:v].copy()}

    def box_filter(self, pts, min_d, max_d):
        pts = _f(pts)
        flags = np.empty(len(pts), np.uint8)
        self.lib.orc_box_filter(_p(pts), C.c_size_t(len(pts)), C.c_float(min_d), C.c_float(max_d), _p(flags, C.c_uint8))
        return flags

    # ---- registration
    def gicp_linearize(self, src, src_cov, tgt, tgt_cov, tgt_nrm, nn_idx, nn_d2, T, max_corr=2.0, reg="GICP",
                       loss="NONE", robust_scale=10.0, genz_alpha=1.0, per_point=False):
        src, tgt = _f(src), _f(tgt)
        src_cov = None if src_cov is None else _f(src_cov)
        tgt_cov = None if tgt_cov is None else _f(tgt_cov)
        tgt_nrm = None if tgt_nrm is None else _f(tgt_nrm)
        nn_idx = np.ascontiguousarray(nn_idx, np.int32).reshape(-1)
        nn_d2 = _f(nn_d2).reshape(-1)
        Tc = _f(np.asarray(T).T)
        out = np.zeros(44, np.float32)
        pp = np.zeros((len(src), 44), np.float32) if per_point else None
        self.lib.orc_gicp_linearize(_p(src), _p(src_cov), C.c_size_t(len(src)), _p(tgt), _p(tgt_cov), _p(tgt_nrm),
                                    _p(nn_idx, C.c_int32), _p(nn_d2), _p(Tc), C.c_float(max_corr), REG[reg], LOSS[loss],
                                    C.c_float(robust_scale), C.c_float(genz_alpha), _p(out), _p(pp))
        res = {"H": out[:36].reshape(6, 6).copy(), "b": out[36:42].copy(), "error": float(out[42]),
               "inlier": int(out[43:44].view(np.uint32)[0])}
        if per_point:
            res["per_point"] = pp
        return res

    def gicp_error(self, src, src_cov, tgt, tgt_cov, tgt_nrm, nn_idx, nn_d2, T, max_corr=2.0, reg="GICP", loss="NONE",
                   robust_scale=10.0, genz_alpha=1.0):
        src, tgt = _f(src), _f(tgt)
        src_cov = None if src_cov is None else _f(src_cov)
        tgt_cov = None if tgt_cov is None else _f(tgt_cov)
        tgt_nrm = None if tgt_nrm is None else _f(tgt_nrm)
        nn_idx = np.ascontiguousarray(nn_idx, np.int32).reshape(-1)
        nn_d2 = _f(nn_d2).reshape(-1)
        Tc = _f(np.asarray(T).T)
        out = np.zeros(2, np.float32)
        self.lib.orc_gicp_error(_p(src), _p(src_cov), C.c_size_t(len(src)), _p(tgt), _p(tgt_cov), _p(tgt_nrm),
                                _p(nn_idx, C.c_int32), _p(nn_d2), _p(Tc), C.c_float(max_corr), REG[reg], LOSS[loss],
                                C.c_float(robust_scale), C.c_float(genz_alpha), _p(out))
        return float(out[0]), int(out[1:2].view(np.uint32)[0])

    def gicp_linearize_rot(self, src, src_cov, tgt, tgt_cov, tgt_nrm, nn_idx, nn_d2, T, max_corr=2.0, reg="GICP",
                           loss="NONE", robust_scale=10.0, genz_alpha=1.0, rot_weight=1.0, rot_scale=10.0):
        """K11 / K12 with the rotation-constraint term: returns (linearised dict, (error, inlier) of K12)."""
        src, tgt = _f(src), _f(tgt)
        src_cov, tgt_cov = _f(src_cov), _f(tgt_cov)
        tgt_nrm = None if tgt_nrm is None else _f(tgt_nrm)
        nn_idx = np.ascontiguousarray(nn_idx, np.int32).reshape(-1)
        nn_d2 = _f(nn_d2).reshape(-1)
        Tc = _f(np.asarray(T).T)
        out = np.zeros(44, np.float32)
        out2 = np.zeros(2, np.float32)
        args = (_p(src), _p(src_cov), C.c_size_t(len(src)), _p(tgt), _p(tgt_cov), _p(tgt_nrm), _p(nn_idx, C.c_int32),
                _p(nn_d2), _p(Tc), C.c_float(max_corr), REG[reg], LOSS[loss], C.c_float(robust_scale),
                C.c_float(genz_alpha), C.c_float(rot_weight), C.c_float(rot_scale))
        self.lib.orc_gicp_linearize_rot(*args, _p(out))
        self.lib.orc_gicp_error_rot(*args, _p(out2))
        res = {"H": out[:36].reshape(6, 6).copy(), "b": out[36:42].copy(), "error": float(out[42]),
               "inlier": int(out[43:44].view(np.uint32)[0])}
        return res, (float(out2[0]), int(out2[1:2].view(np.uint32)[0]))

    def degenerate_regularize(self, H, b, inlier, T_cur, T_init, rot_thr=10.0, trans_thr=1.0, base_factor=1.0):
        Hc = _f(np.asarray(H).T).copy()
        bc = _f(b).copy()
        self.lib.orc_degenerate_regularize(C.c_int(1), C.c_float(rot_thr), C.c_float(trans_thr), C.c_float(base_factor),
                                           _p(Hc), _p(bc), C.c_uint32(inlier), _p(_f(np.asarray(T_cur).T)),
                                           _p(_f(np.asarray(T_init).T)))
        return Hc.T.copy(), bc

    def map_prior_update(self, H_raw, error_raw, inlier, T_prev, T_pred, sigmas=(1.0, 1.0, 3.16e-2, 1e-2)):
        """-> (has_prior, Omega 6x6, T_pred_inv 4x4); sigmas = rot_vel, trans_vel, rot_base, trans_base."""
        om = np.zeros(36, np.float32)
        tinv = np.zeros(16, np.float32)
        sg = _f(sigmas)
        has = self.lib.orc_map_prior_update(_p(sg), _p(_f(np.asarray(H_raw).T)), C.c_float(error_raw), C.c_uint32(inlier),
                                            _p(_f(np.asarray(T_prev).T)), _p(_f(np.asarray(T_pred).T)), _p(om), _p(tinv))
        return bool(has), om.reshape(6, 6).T.copy(), tinv.reshape(4, 4).T.copy()

    def map_prior_apply(self, omega, T_pred_inv, H, b, error, T_est):
        """-> (H + Omega, b + Omega e, error + e'Omega e / 2, prior_error(T_est))"""
        Hc = _f(np.asarray(H).T).copy()
        bc = _f(b).copy()
        err = C.c_float(error)
        self.lib.orc_map_prior_apply.restype = C.c_float
        pe = self.lib.orc_map_prior_apply(_p(_f(np.asarray(omega).T)), _p(_f(np.asarray(T_pred_inv).T)), _p(Hc), _p(bc),
                                          C.byref(err), _p(_f(np.asarray(T_est).T)))
        return Hc.T.copy(), bc, float(err.value), float(pe)

    def icp_robust_weights(self, src, src_cov, tgt, tgt_cov, tgt_nrm, nn_idx, nn_d2, T, max_corr=2.0, reg="GICP",
                           loss="NONE", robust_scale=10.0):
        src, tgt = _f(src), _f(tgt)
        src_cov = None if src_cov is None else _f(src_cov)
        tgt_cov = None if tgt_cov is None else _f(tgt_cov)
        tgt_nrm = None if tgt_nrm is None else _f(tgt_nrm)
        nn_idx = np.ascontiguousarray(nn_idx, np.int32).reshape(-1)
        nn_d2 = _f(nn_d2).reshape(-1)
        Tc = _f(np.asarray(T).T)
        out = np.zeros(len(src), np.float32)
        self.lib.orc_icp_robust_weights(_p(src), _p(src_cov), C.c_size_t(len(src)), _p(tgt), _p(tgt_cov), _p(tgt_nrm),
                                        _p(nn_idx, C.c_int32), _p(nn_d2), _p(Tc), C.c_float(max_corr), REG[reg],
                                        LOSS[loss], C.c_float(robust_scale), _p(out))
        return out

    def voxel_hash_map(self, voxel_size):
        return OracleVoxelHashMap(self, voxel_size)

    def log_spd3(self, A):
        A = _f(np.asarray(A).T)
        out = np.zeros((3, 3), np.float32)
        self.lib.orc_log_spd3(_p(A), _p(out))
        return out.T.copy()

    def exp_spd3(self, A):
        A = _f(np.asarray(A).T)
        out = np.zeros((3, 3), np.float32)
        self.lib.orc_exp_spd3(_p(A), _p(out))
        return out.T.copy()

    def registration_align(self, params, src, src_cov, tgt, tgt_cov, tgt_nrm=None, init_T=None, nn_mode="kdtree",
                           trace=False, nodes=None, steps=False):
        src, tgt = _f(src), _f(tgt)
        src_cov = None if src_cov is None else _f(src_cov)
        tgt_cov = None if tgt_cov is None else _f(tgt_cov)
        tgt_nrm = None if tgt_nrm is None else _f(tgt_nrm)
        Tc = _f(np.eye(4) if init_T is None else np.asarray(init_T).T)
        res = RegResult()
        tr = np.zeros((max(params.max_iterations, 1), 16), np.float32) if trace else None
        trn = C.c_int(0)
        st = np.zeros((4096, 5), np.float32) if steps else None
        stn = C.c_int(0)
        self.lib.orc_registration_align(C.byref(params), _p(src), _p(src_cov), C.c_size_t(len(src)), _p(tgt), _p(tgt_cov),
                                        _p(tgt_nrm), C.c_size_t(len(tgt)), _p(Tc), C.c_int(0 if nn_mode == "kdtree" else 1),
                                        C.byref(res), _p(tr), C.byref(trn),
                                        None if nodes is None else nodes.ctypes.data_as(C.c_void_p),
                                        C.c_size_t(0 if nodes is None else len(nodes) // 32),
                                        _p(st), C.c_int(0 if st is None else len(st)), C.byref(stn))
        out = {"T": np.array(res.T, np.float32).reshape(4, 4).T.copy(),
               "H": np.array(res.H, np.float32).reshape(6, 6).T.copy(), "b": np.array(res.b, np.float32),
               "error": float(res.error), "inlier": int(res.inlier), "iterations": int(res.iterations),
               "converged": bool(res.converged),
               "H_raw": np.array(res.H_raw, np.float32).reshape(6, 6).T.copy(), "b_raw": np.array(res.b_raw, np.float32),
               "error_raw": float(res.error_raw)}
        if steps:  # per outer iteration (all annealing levels in order): trials, accepted, damping after, result.error after, margin of
            # the closest decision to its threshold
            out["steps"] = [dict(trials=int(r[0]), accepted=int(r[1]), damping=float(r[2]), error=float(r[3]), margin=float(r[4]))
                            for r in st[:stn.value]]
        if trace:
            out["trace"] = np.stack([tr[i].reshape(4, 4).T for i in range(trn.value)]) if trn.value else np.zeros((0, 4, 4))
        return out


class OracleVoxelHashMap:
    """algorithms/mapping/voxel_hash_map.hpp restated (oracle/oracle_voxel_hash_map.hpp). Poses are row-major 4x4 numpy."""
    PARAM = {"voxel_size": 0, "max_staleness": 1, "remove_old_data_cycle": 2, "rehash_threshold": 3, "min_num_point": 4}
    INFO = {"voxel_num": 0, "capacity": 1, "staleness_counter": 2, "has_cov": 3, "has_rgb": 4, "has_intensity": 5}

    def __init__(self, orc, voxel_size):
        L = orc.lib
        L.orc_vhm_new.restype = C.c_void_p
        L.orc_vhm_new.argtypes = [C.c_float]
        L.orc_vhm_info.restype = C.c_size_t
        L.orc_vhm_info.argtypes = [C.c_void_p, C.c_int]
        L.orc_vhm_set.argtypes = [C.c_void_p, C.c_int, C.c_float]
        L.orc_vhm_downsampling.restype = C.c_size_t
        L.orc_vhm_overlap_ratio.restype = C.c_float
        for f in (L.orc_vhm_free, L.orc_vhm_clear, L.orc_vhm_remove_old_data):
            f.argtypes = [C.c_void_p]
        if not voxel_size > 0.0:
            raise ValueError("voxel_size must be positive.")  # voxel_hash_map.hpp:41-43: std::invalid_argument
        self.orc = orc
        self.h = C.c_void_p(L.orc_vhm_new(voxel_size))

    def set(self, name, value):
        self.orc.lib.orc_vhm_set(self.h, self.PARAM[name], float(value))

    def info(self, name):
        return int(self.orc.lib.orc_vhm_info(self.h, self.INFO[name]))

    def clear(self):
        self.orc.lib.orc_vhm_clear(self.h)

    def add_point_cloud(self, pts, pose=None, covs=None, rgb=None, intensities=None):
        pts = _f(pts).reshape(-1, 4)
        covs = None if covs is None else _f(covs)
        rgb = None if rgb is None else _f(rgb)
        inten = None if intensities is None else _f(intensities)
        Tc = _f(np.eye(4) if pose is None else np.asarray(pose).T)
        self.orc.lib.orc_vhm_add(self.h, _p(pts), _p(covs), _p(rgb), _p(inten), C.c_size_t(len(pts)), _p(Tc))

    def downsampling(self, center=(0.0, 0.0, 0.0), distance=100.0):
        cap = max(self.info("voxel_num"), 1)
        pts = np.zeros((cap, 4), np.float32)
        cov = np.zeros((cap, 16), np.float32)
        rgb = np.zeros((cap, 4), np.float32)
        inten = np.zeros(cap, np.float32)
        keys = np.zeros(cap, np.uint64)
        c = _f(center)
        n = self.orc.lib.orc_vhm_downsampling(self.h, _p(c), C.c_float(distance), _p(pts), _p(cov), _p(rgb), _p(inten),
                                              _p(keys, C.c_uint64))
        return {"points": pts[:n], "covs": cov[:n] if self.info("has_cov") else None,
                "rgb": rgb[:n] if self.info("has_rgb") else None,
                "intensities": inten[:n] if self.info("has_intensity") else None, "keys": keys[:n]}

    def overlap_ratio(self, pts, pose=None):
        pts = _f(pts).reshape(-1, 4)
        Tc = _f(np.eye(4) if pose is None else np.asarray(pose).T)
        return float(self.orc.lib.orc_vhm_overlap_ratio(self.h, _p(pts), C.c_size_t(len(pts)), _p(Tc)))

    def remove_old_data(self):
        self.orc.lib.orc_vhm_remove_old_data(self.h)

    def __del__(self):
        try:
            self.orc.lib.orc_vhm_free(self.h)
        except Exception:
            pass


class _Rng:
    def __init__(self, orc, seed):
        self.orc = orc
        self.h = C.c_void_p(orc.lib.orc_rng_new(C.c_uint32(seed)))

    def uniform_points(self, n, rng_range):
        out = np.empty((n, 4), np.float32)
        self.orc.lib.orc_rng_uniform_points(self.h, C.c_float(rng_range), C.c_size_t(n), out.ctypes.data_as(C.c_void_p))
        return out

    def normal(self, n, stddev):
        out = np.empty(n, np.float32)
        self.orc.lib.orc_rng_normal(self.h, C.c_float(stddev), C.c_size_t(n), out.ctypes.data_as(C.c_void_p))
        return out

    def __del__(self):
        try:
            self.orc.lib.orc_rng_free(self.h)
        except Exception:
            pass
