"""VERDICT r04 item 4(a), go / no-go before any kernel work: how many target x-rows do the 2x2x2 blocks of one wave's 64
consecutive (cell-ordered) source points touch at config 4, and how long are the row segments? (numpy, no GPU.)
The premise of wave-private row staging is "<= 3x3 (tilted <= 4x4) contiguous row segments per wave"."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sycl_points_amd.synthetic import gicp_pair

n = int(os.environ.get("N", 1_000_000))
R = 10.0 * (n / 1e6) ** (1 / 3)
src, tgt, T_gt = gicp_pair(n, R)
rho = n / (2 * R) ** 3
h_t = (0.5 / rho) ** (1 / 3)      # target grid, 0.5 points per cell (bench.py --ppc)
h_s = (1.0 / rho) ** (1 / 3)      # the source is stored in the cell order of a grid on itself at 1 point per cell
lo = tgt[:, :3].min(0)
# cell order of the source: z-major, then y, then x (GridKNN.order())
c = np.floor((src[:, :3] - src[:, :3].min(0)) / h_s).astype(np.int64)
order = np.lexsort((c[:, 0], c[:, 1], c[:, 2]))
s = src[order, :3]
for name, T in (("launch 0 (identity guess)", np.eye(4)), ("converged pose", T_gt.astype(np.float64))):
    q = s @ T[:3, :3].T + T[:3, 3]
    f = (q - lo) / h_t
    cell = np.floor(f).astype(np.int64)
    frac = f - cell
    lo_side = frac < 0.5                     # block = {cell-1, cell} when t < 0.5 else {cell, cell+1}
    b0 = cell - lo_side
    nw = n // 64
    b0 = b0[: nw * 64].reshape(nw, 64, 3)
    rows, xs = [], []
    for w in range(0, nw, max(1, nw // 2000)):  # a sample of the waves
        y = np.concatenate([b0[w, :, 1], b0[w, :, 1] + 1])
        z = np.concatenate([b0[w, :, 2], b0[w, :, 2] + 1])
        yz = set()
        for yy in (0, 1):
            for zz in (0, 1):
                yz.update(zip((b0[w, :, 1] + yy).tolist(), (b0[w, :, 2] + zz).tolist()))
        rows.append(len(yz))
        xs.append(int(b0[w, :, 0].max() + 1 - b0[w, :, 0].min() + 1))
        bbox = (y.max() - y.min() + 1) * (z.max() - z.min() + 1)
        rows[-1] = (len(yz), int(bbox))
    r = np.array([a for a, _ in rows]); bb = np.array([b for _, b in rows]); xs = np.array(xs)
    print(f"{name}: target cell {h_t:.3f} m, source cell {h_s:.3f} m; per wave of 64 consecutive source points:")
    print(f"   distinct (y,z) rows touched by the 2x2x2 blocks: median {np.median(r):.0f}, 90 % {np.percentile(r, 90):.0f}, max {r.max()}")
    print(f"   rows of their bounding box:                      median {np.median(bb):.0f}, 90 % {np.percentile(bb, 90):.0f}")
    print(f"   x-extent of the wave in target cells:            median {np.median(xs):.0f}, 90 % {np.percentile(xs, 90):.0f}")
    print(f"   cells in the bounding box (rows x extent):       median {np.median(bb * xs):.0f} -> {np.median(bb * xs) * 0.5:.0f} points to stage; "
          f"cells actually needed 64 x 8 = 512")
