"""where the time of the host-buffer LK call goes (PCIe-inclusive path)"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dr3 = importlib.import_module("3dr_amd"); synth = importlib.import_module("3dr_amd.synth")
ctx = dr3.Context(0)
A, B, T = synth.frame_pair()
pts = synth.grid_points(1241, 376, 15, n=2000)
p = dr3.lk_params(win=(15, 15), max_level=3, max_iters=30, eps=0.01)
def t(f, n=20):
    f(); ctx.sync(); t0 = time.perf_counter()
    for _ in range(n): f()
    ctx.sync(); return (time.perf_counter() - t0) / n * 1e3
print("lk_track host API      %.3f ms" % t(lambda: ctx.calcOpticalFlowPyrLK(A, B, pts, None, p)))
b = dr3.FrameBatch(ctx, 1241, 376, win=(15, 15), max_level=3, capacity=2)
print("batch.upload 1 frame   %.3f ms" % t(lambda: b.upload(0, A)))
d = ctx.alloc(A.nbytes)
print("1-D h2d of the frame   %.3f ms" % t(lambda: d.upload(A)))
print("batch.build 2 slots    %.3f ms" % t(lambda: b.build(0, 2, True)))
P = ctx.to_device(pts.astype(np.float32)); N = ctx.to_device(pts.astype(np.float32)); S = ctx.alloc(2000); E = ctx.alloc(8000)
print("batch.track 1 pair     %.3f ms" % t(lambda: b.track(1, 0, P, N, 2000, S, E, p)))
print("d2h 16 KB              %.3f ms" % t(lambda: N.download((2000, 2), np.float32)))
ctx.profile_enable(True); ctx.profile_reset()
for _ in range(20): b.track(1, 0, P, N, 2000, S, E, p)
ms, n = ctx.profile_read(dr3.PROF_LK)
print("  of which kernel      %.3f ms (dispatch timestamps)" % (ms / n))
for _ in range(20): b.build(0, 2, True)
ms, n = ctx.profile_read(dr3.PROF_PYR)
print("build kernels          %.3f ms (first kernel start to last kernel end)" % (ms / n))
