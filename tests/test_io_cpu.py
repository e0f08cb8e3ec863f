"""File I/O of the C++ facade (PLY / PCD readers and writers): host-only C++ tests modelled on the
reference's cpp/tests/test_file_io.cpp, built with g++ and run here — no GPU work."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")


def test_point_cloud_reader_writer_round_trips():
    from sycl_points_amd import _lib

    _lib.build()  # the facade headers link libsycl_points_amd.so (host helpers only are used here)
    subprocess.check_call(["make", "-C", CPP, "-s", "test_io"])
    env = dict(os.environ, SP_GOLDEN_DIR=os.path.join(ROOT, "tests", "golden"))
    r = subprocess.run([os.path.join(CPP, "test_io")], capture_output=True, text=True, timeout=300, env=env)
    print(r.stdout[-3000:], r.stderr[-2000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " 0 failed" in r.stdout
