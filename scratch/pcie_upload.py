import time, numpy as np, torch
n=1000000
host=[np.random.rand(n,4).astype(np.float32), np.random.rand(n,16).astype(np.float32)]
for pinned in (False, True):
    ts=[torch.from_numpy(a) for a in host]
    if pinned: ts=[t.pin_memory() for t in ts]
    for _ in range(2): d=[t.cuda(non_blocking=True) for t in ts]; torch.cuda.synchronize()
    t0=time.perf_counter()
    for _ in range(5): d=[t.cuda(non_blocking=True) for t in ts]; torch.cuda.synchronize()
    dt=(time.perf_counter()-t0)/5
    print("pinned" if pinned else "pageable", "upload of one cloud (points + covariances, 80 MB): %.2f ms = %.1f GB/s"%(dt*1e3, 80e6/dt/1e9))
