"""Development check of the tile-search form (csrc/gicp_tile.h): bit-identity against the streaming form and per-launch
times. python scratch/tile_check.py [n] [ppc]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sycl_points_amd.api as sp  # noqa: E402
from sycl_points_amd import _lib  # noqa: E402
from sycl_points_amd.synthetic import gicp_pair  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
ppc = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ITERS = 20
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
src, tgt, T_gt = gicp_pair(n, 10.0 * (n / 1e6) ** (1.0 / 3.0))
Tg = sp.PointCloudShared(dev(tgt))
Tg.covs = sp.GridKNN.build(Tg.points, points_per_cell=6.0).self_knn(20, want_knn=False, want_covs=True)[1]
grid = sp.GridKNN.build(Tg.points, points_per_cell=ppc)
prep = sp.PreparedTarget(grid, Tg.covs)
p = sp.RegistrationParams(criteria_translation=0.0, criteria_rotation=0.0, max_iterations=ITERS)

for order in ("grid", "random"):
    S_all = dev(src)
    if order == "grid":
        S_all = S_all[sp.GridKNN.build(S_all, points_per_cell=1.0).order()].contiguous()
    covs = sp.GridKNN.build(S_all, points_per_cell=6.0).self_knn(20, want_knn=False, want_covs=True)[1]
    S = sp.PointCloudShared(S_all, covs=covs)
    mode = "presorted" if order == "grid" else True

    def run(search_mode, reuse=2, iters=ITERS):
        reg = sp.Registration(p)
        reg._set_source_option("reuse", reuse)
        reg._set_source_option("search_mode", search_mode)
        T_dev, lin, delta = reg.align_fused_loop(S, prep, iterations=iters, sort_by_cell=mode, write_neighbors=True)
        torch.cuda.synchronize()
        return (T_dev.cpu().numpy().copy(), lin.cpu().numpy().copy(), reg.neighbors.indices.cpu().numpy().ravel().copy(),
                reg.neighbors.distances.cpu().numpy().ravel().copy(), delta.cpu().numpy().copy(), reg)

    ref = run(0)
    names = ("pose", "lin", "nn_idx", "nn_d2", "delta")
    for sm, reuse in ((1, 2), (-1, 2), (1, 0)):
        got = run(sm, reuse)
        same = [bool(np.array_equal(x, y)) for x, y in zip(ref[:5], got[:5])]
        print(f"[{order}] search_mode={sm} reuse={reuse}: identical to the streaming form: {dict(zip(names, same))}", flush=True)
        if not all(same):
            bad = np.flatnonzero(ref[2] != got[2])
            print("   differing neighbours:", len(bad), bad[:10], ref[2][bad[:10]], got[2][bad[:10]])
            print("   pose diff", np.abs(ref[0] - got[0]).max())
    T = ref[0].reshape(4, 4).T
    print(f"[{order}] pose error vs ground truth {np.abs(T - T_gt).max():.2e}", flush=True)
    # one iteration after a single launch (every point searched at the identity pose): neighbours against grid.knn_search
    one = run(1, 2, 1)
    exact = grid.knn_search(S, 1, np.eye(4, dtype=np.float32))
    print(f"[{order}] launch 0 neighbours == grid.knn_search: "
          f"{np.array_equal(exact.indices.cpu().numpy().ravel(), one[2])} {np.array_equal(exact.distances.cpu().numpy().ravel(), one[3])}")

    # per-launch times (events between the launches of one alignment), both forms
    L = _lib.lib()
    for sm in (0, 1, -1):
        reg = sp.Registration(p)
        reg._set_source_option("search_mode", sm)
        T_dev = torch.zeros(16, dtype=torch.float32, device="cuda")
        T_ident = torch.eye(4, dtype=torch.float32, device="cuda").reshape(-1).contiguous()
        delta = torch.zeros(8, dtype=torch.float32, device="cuda")
        T_dev.copy_(T_ident)
        reg.align_fused_loop(S, prep, iterations=ITERS, T_dev=T_dev, delta_dev=delta, prepare=True, sort_by_cell=mode)
        torch.cuda.synchronize()
        ws, lin = reg._buffers(T_dev.device)
        fp = reg._factor_params(reg.params.robust_default_scale)
        gn = _lib.GnParams(reg.params.gn_lambda, 0.0, 0.0)
        nlog = C.c_size_t(0)
        log_off = L.sp_internal_align_searched_log(sp._ptr(ws), C.byref(nlog)) - ws.data_ptr()
        reps = 5
        us = np.zeros((reps, ITERS))
        prep_us = []
        for r in range(reps):
            T_dev.copy_(T_ident)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            reg._psrc.prepare(prep, S, T_dev, mode)
            e1.record()
            torch.cuda.synchronize()
            prep_us.append(1e3 * e0.elapsed_time(e1))
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(ITERS + 1)]
            torch.cuda._sleep(3_000_000)
            ev[0].record()
            for k in range(ITERS):
                _lib.check(L.sp_gicp_align_step(prep._h, reg._psrc._h, sp._ptr(T_dev), C.byref(fp), C.byref(gn), k, 0, None,
                                                None, sp._ptr(lin), sp._ptr(ws), ws.numel(), sp._stream()))
                ev[k + 1].record()
            _lib.check(L.sp_gicp_align_finish(reg._psrc._h, sp._ptr(T_dev), C.byref(gn), ITERS - 1, 0, sp._ptr(lin),
                                              sp._ptr(delta), None, sp._ptr(ws), ws.numel(), sp._stream()))
            torch.cuda.synchronize()
            us[r] = [1e3 * ev[k].elapsed_time(ev[k + 1]) for k in range(ITERS)]
        searched = ws[log_off:log_off + 4 * ITERS].view(torch.int32).cpu().numpy()
        dbg = ws[log_off + 160:log_off + 256].view(torch.int32).cpu().numpy()
        print("   dbg [segments, unfit, q1, q2, nonfinal, short parts, rows, nstarts]:", dbg[:8], " WG3 x10ns [bbox, starts, scan, points+queue, stage1, stage2+end, pts copy, long rows, starts store + q writes]:", dbg[8:18], " all WGs x10ns [max p1, max p2, sum p1/16, sum p2/16, max total]:", dbg[16:21])
        med = np.median(us, axis=0)
        print(f"[{order}] search_mode={sm:2d}: prepare {np.median(prep_us):.1f} us; launches "
              + " ".join(f"{m:.1f}" for m in med[:6]) + f" ... steady {np.median(med[6:]):.1f}; searched {searched[:5]}"
              + f"; sum20 {med.sum():.0f} us", flush=True)
