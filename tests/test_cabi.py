"""CPU-side checks of the drop-in boundary: the library loads, exports every symbol include/sycl_points_amd.h declares,
the ctypes table matches the header, and the host-only entry points agree with the oracle. No GPU compute calls."""
import ctypes as C
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "sycl_points_amd.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sp_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_path():
    syms = declared_symbols()
    for must in ("sp_knn_bruteforce", "sp_kdtree_search", "sp_cov_estimate", "sp_normals_from_knn", "sp_voxel_keys",
                 "sp_voxel_downsample", "sp_gicp_linearize", "sp_gicp_error", "sp_gn_update", "sp_transform"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from sycl_points_amd import _lib

    path = _lib.build()
    lib = C.CDLL(path)
    syms = declared_symbols()
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, f"declared in the header but not exported: {missing}"
    assert set(_lib.SIGNATURES) == set(syms), set(_lib.SIGNATURES) ^ set(syms)
    assert _lib.lib().sp_abi_version() == 6
    # measurement / tuning switches are per handle and live in csrc/sp_internal.h, not in the public header
    assert not [s for s in syms if s.startswith(("sp_debug", "sp_internal"))]
    assert all(hasattr(lib, s) for s in _lib.INTERNAL_SIGNATURES)


def test_per_iteration_kernels_do_not_spill():
    """The per-iteration kernel of the device-resident loop runs 1024-thread workgroups, i.e. at most 128 VGPRs; a spill in
    it costs a measured +1 us and ~1 MB of scratch writes per launch (profiles/README.md). The build keeps the compiler's
    resource report. Held to ZERO VGPR spills: every GICP instantiation on the benchmarked search form (all five robust
    losses — NONE is the benchmarked one). The point-to-distribution instantiations are held to the two registers the
    compiler spills today for NONE / TUKEY / CAUCHY (a regression shows; they are listed in DESIGN.md)."""
    import re

    from sycl_points_amd import _lib

    _lib.build()
    report = os.path.join(ROOT, "sycl_points_amd", "lib", "registration.resources.txt")
    if not os.path.exists(report):  # a library built before the report existed: rebuild that one object
        import subprocess
        csrc = os.path.join(ROOT, "sycl_points_amd", "csrc")
        os.utime(os.path.join(csrc, "registration.hip"))
        subprocess.run(["make", "-C", csrc, "-s", "-j8"], check=True)
    rows = [l for l in open(report) if "gicp_align_kernelILi" in l]
    assert len(rows) == 5 * 2 * 2 * 2, rows  # loss x search form x factor x (single GPU | sharded)
    for row in rows:
        spills = int(re.search(r"VGPRs Spill: (\d+)", row).group(1))
        assert int(re.search(r"VGPRs: (\d+)", row).group(1)) <= 128, row
        p2d = re.search(r"gicp_align_kernelILi\dELb[01]ELb1E", row) is not None
        tukey = "gicp_align_kernelILi2E" in row
        single_gpu = re.search(r"gicp_align_kernelILi\dELb[01]ELb[01]ELb0E", row) is not None
        if "gicp_align_kernelILi0ELb1ELb0ELb0E" in row:  # the benchmarked instantiation
            assert spills == 0, row
        # GICP: none (Tukey's weight keeps one more value alive: one register); POINT_TO_DISTRIBUTION: up to five registers
        # (its own point loop is still open)
        assert spills <= (5 if p2d else (1 if tukey else 0)), (single_gpu, row)


def test_sp_linearized_is_192_bytes():
    from sycl_points_amd import _lib

    assert C.sizeof(_lib.Linearized) == 192
    assert _lib.Linearized.error.offset == 42 * 4 and _lib.Linearized.inlier.offset == 43 * 4
    assert _lib.Linearized.inlier_lo.offset == 44 * 4 and _lib.Linearized.inlier_hi.offset == 45 * 4


def test_host_solver_twins_match_oracle(orc):
    """sp_se3_exp_host / sp_rigid_mul_host / sp_ldlt6_solve_host / sp_gn_update_host vs the oracle, bit for bit
    (same IEEE operations in the same order; sinf/cosf are the host libm on both sides)."""
    from sycl_points_amd import _lib

    L = _lib.lib()
    rs = np.random.RandomState(3)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    for _ in range(200):
        tw = rs.uniform(-0.5, 0.5, 6).astype(np.float32)
        T = np.zeros(16, np.float32)
        L.sp_se3_exp_host(vp(tw), vp(T))
        Tm = T.reshape(4, 4).T
        assert np.array_equal(Tm, orc.se3_exp(tw))
        tw2 = rs.uniform(-0.5, 0.5, 6).astype(np.float32)
        T2 = np.zeros(16, np.float32)
        L.sp_se3_exp_host(vp(tw2), vp(T2))
        out = np.zeros(16, np.float32)
        L.sp_rigid_mul_host(vp(T), vp(T2), vp(out))
        assert np.array_equal(out.reshape(4, 4).T, orc.isometry_mul(Tm, T2.reshape(4, 4).T))
        J = rs.uniform(-1, 1, (30, 6))
        H = np.ascontiguousarray((J.T @ J + np.eye(6)).astype(np.float32))
        H = ((H + H.T) * np.float32(0.5)).astype(np.float32)
        b = rs.uniform(-1, 1, 6).astype(np.float32)
        x = np.zeros(6, np.float32)
        assert L.sp_ldlt6_solve_host(vp(H), vp(b), vp(x)) == 0
        ok, xo = orc.ldlt6_solve(H, b)
        assert ok and np.array_equal(x, xo)
        # one Gauss-Newton update: delta = solve(H + lambda I, -b); T <- T * exp(delta)
        lin = _lib.Linearized()
        for i in range(36):
            lin.H[i] = float(H.reshape(-1)[i])
        for i in range(6):
            lin.b[i] = float(b[i])
        Tcur = T.copy()
        d8 = np.zeros(8, np.float32)
        assert L.sp_gn_update_host(C.byref(lin), vp(Tcur), 1.0, 1e-3, 1e-3, vp(d8)) == 0
        ok, delta = orc.ldlt6_solve(H + np.eye(6, dtype=np.float32), -b)
        assert np.array_equal(d8[:6], delta)
        assert np.array_equal(Tcur.reshape(4, 4).T, orc.isometry_mul(Tm, orc.se3_exp(delta)))
        conv = np.linalg.norm(delta[:3]) < 1e-3 and np.linalg.norm(delta[3:]) < 1e-3
        assert bool(d8[6]) == bool(conv) and d8[7] == 1.0


def test_error_codes_without_gpu():
    from sycl_points_amd import _lib

    L = _lib.lib()
    # argument validation happens before any device work
    assert L.sp_knn_bruteforce(None, 4, None, 4, 21, None, None, None, 0, None) == _lib.SP_ERR_INVALID_ARGUMENT
    assert b"MAX_K" in L.sp_last_error()
    assert L.sp_kdtree_search(None, None, 4, 1, None, 0, None, None, None) == _lib.SP_ERR_INVALID_ARGUMENT


def test_synthetic_generator_is_the_reference_idiom(orc):
    # std::mt19937 + uniform_real_distribution<float> (cpp/tests/test_kdtree.cpp:69-75) reproduced in numpy
    from sycl_points_amd.synthetic import Mt19937Cloud, gicp_pair

    g, h = Mt19937Cloud(1234), orc.rng(1234)
    assert np.array_equal(g.uniform_points(1000, 10.0), h.uniform_points(1000, 10.0))
    assert np.array_equal(g.uniform_points(100, 10.0), h.uniform_points(100, 10.0))  # the stream continues
    src, tgt, T = gicp_pair(2000, 1.26)
    back = src[:, :3].astype(np.float64) @ T[:3, :3].T.astype(np.float64) + T[:3, 3]
    assert np.abs(back - tgt[:, :3]).max() < 0.05 and np.abs(back - tgt[:, :3]).std() < 0.01


def test_build_entry_point_runs():
    """__graft_entry__.build() (the driver's does-it-build check): compiles what is stale, loads the library, checks the ABI."""
    import __graft_entry__ as g

    g.build()
