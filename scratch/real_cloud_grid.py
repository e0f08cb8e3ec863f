"""GridKNN on the reference's bundled (real LiDAR) cloud: search time against the points-per-cell target."""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sycl_points_amd.api as sp

def read_ply_xyz(path):
    with open(path, "rb") as f:
        header = []
        while True:
            l = f.readline().decode().strip(); header.append(l)
            if l == "end_header": break
        n = [int(l.split()[-1]) for l in header if l.startswith("element vertex")][0]
        props = [l.split()[1:] for l in header if l.startswith("property")]
        m = {"float": "<f4", "float32": "<f4", "double": "<f8", "uchar": "u1", "uint8": "u1", "int": "<i4", "uint": "<u4"}
        dt = np.dtype([(name, m[t]) for t, name in props])
        arr = np.frombuffer(f.read(n * dt.itemsize), dtype=dt, count=n)
    P = np.ones((n, 4), np.float32)
    P[:, 0], P[:, 1], P[:, 2] = arr["x"], arr["y"], arr["z"]
    return P

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T = torch.from_numpy(read_ply_xyz(os.path.join(root, "tests/golden/target.ply"))).cuda()
S = torch.from_numpy(read_ply_xyz(os.path.join(root, "tests/golden/source.ply"))).cuda()
def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
for ppc in (6.0, 2.0, 0.5, 0.1, 0.02, 0.005):
    g = sp.GridKNN.build(T, points_per_cell=ppc)
    t_build = timed(lambda: sp.GridKNN.build(T, points_per_cell=ppc))
    t1 = timed(lambda: g.knn_search(S, 1))
    t10 = timed(lambda: g.knn_search(T, 10))
    t20 = timed(lambda: g.knn_search(T, 20))
    print(f"ppc {ppc:6}: cell {g.cell_size():.3f} m  build {t_build:.3f} ms  NN(k=1, source->target) {t1:.3f}  k=10 self {t10:.3f}  k=20 self {t20:.3f} ms")
kd = sp.KDTree.build(T)
print("KD-tree: k=1 %.3f  k=10 %.3f  k=20 %.3f ms" % (timed(lambda: kd.knn_search(S, 1)), timed(lambda: kd.knn_search(T, 10)), timed(lambda: kd.knn_search(T, 20))))
