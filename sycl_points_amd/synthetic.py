"""Deterministic synthetic clouds for the benchmark configs.

The reference's tests draw clouds with `std::mt19937 gen(seed); std::uniform_real_distribution<float> U(-R, R);
p = (U(gen), U(gen), U(gen), 1)` (cpp/tests/test_kdtree.cpp:69-75). This module reproduces that stream with numpy:
RandomState(seed) is MT19937 with the same init_genrand seeding, and libstdc++'s generate_canonical<float, 24> with a
32-bit engine is one draw: float(u) / 2^32 (clamped below 1), then (b - a) * r + a in float.
"""
import numpy as np


class Mt19937Cloud:
    def __init__(self, seed):
        self.rs = np.random.RandomState(seed)

    def _canonical(self, n):
        u = self.rs.randint(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32)  # raw 32-bit outputs
        r = u.astype(np.float32) / np.float32(4294967296.0)
        one_below = np.nextafter(np.float32(1.0), np.float32(0.0))
        return np.where(r >= np.float32(1.0), one_below, r).astype(np.float32)

    def uniform_points(self, n, rng_range):
        a = np.float32(-rng_range)
        b = np.float32(rng_range)
        r = self._canonical(3 * n)
        xyz = ((b - a) * r + a).astype(np.float32).reshape(n, 3)
        out = np.ones((n, 4), np.float32)
        out[:, :3] = xyz
        return out


def se3_exp_f64(twist):
    """Rotation-first SE(3) exponential in float64 (ground-truth pose for the synthetic GICP configs)."""
    w = np.asarray(twist[:3], np.float64)
    v = np.asarray(twist[3:], np.float64)
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        R, V = np.eye(3) + K, np.eye(3)
    else:
        R = np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th**2 * K @ K
        V = np.eye(3) + (1 - np.cos(th)) / th**2 * K + (th - np.sin(th)) / th**3 * K @ K
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = V @ v
    return T


GICP_TWIST = (0.01, -0.02, 0.015, 0.03, -0.02, 0.01)


def gicp_pair(n, rng_range, seed=1234, noise_seed=4321, noise_std=0.005):
    """target ~ U(-R,R)^3 ; source = T_gt^-1 * target + N(0, noise_std^2) ; returns (source, target, T_gt)."""
    tgt = Mt19937Cloud(seed).uniform_points(n, rng_range)
    T = se3_exp_f64(GICP_TWIST)
    Tinv = np.linalg.inv(T)
    src = np.ones((n, 4), np.float32)
    src[:, :3] = (tgt[:, :3].astype(np.float64) @ Tinv[:3, :3].T + Tinv[:3, 3]).astype(np.float32)
    noise = np.random.RandomState(noise_seed).normal(0.0, noise_std, size=(n, 3)).astype(np.float32)
    src[:, :3] += noise
    return src, tgt, T.astype(np.float32)
