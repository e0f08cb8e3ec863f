"""sycl_points_amd — MI355X-native (gfx950) implementation of the sycl_points registration hot path.

Compute lives in csrc/ (hand-written HIP kernels behind the C ABI of include/sycl_points_amd.h);
api.py mirrors the reference's operator interface over that ABI for tests and the benchmark.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
