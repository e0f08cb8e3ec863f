import ctypes as C, torch
hip = C.CDLL("libamdhip64.so")
side = torch.cuda.Stream()
a = torch.zeros(1024, device="cuda")
def attempt(name, fn):
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            a.add_(1.0)
            rc = fn()
            a.mul_(2.0)
        g.replay(); torch.cuda.synchronize()
        print(name, "capture OK, rc", rc, flush=True)
    except Exception as e:
        print(name, "CAPTURE BROKEN:", str(e).split("\n")[0], flush=True)
        torch.cuda.synchronize()
ev = C.c_void_p()
attempt("nothing", lambda: 0)
attempt("hipEventCreateWithFlags", lambda: hip.hipEventCreateWithFlags(C.byref(ev), 2))
attempt("hipEventRecord(side)", lambda: hip.hipEventRecord(ev, C.c_void_p(side.cuda_stream)))
attempt("hipEventQuery", lambda: hip.hipEventQuery(ev))
attempt("hipEventDestroy", lambda: hip.hipEventDestroy(ev))
attempt("hipStreamQuery(side)", lambda: hip.hipStreamQuery(C.c_void_p(side.cuda_stream)))
