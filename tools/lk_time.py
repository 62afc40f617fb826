"""experiment helper: LK launch time at the bench's shape for the library named by DR3HIP_LIB (default: the built one).
usage: python tools/lk_time.py [win] [levels] ; prints ms per launch for (iters 30, eps 0.01) and for iters 0"""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dr3 = importlib.import_module("3dr_amd"); synth = importlib.import_module("3dr_amd.synth")
WIN = int(sys.argv[1]) if len(sys.argv) > 1 else 15
ML = int(sys.argv[2]) - 1 if len(sys.argv) > 2 else 3
B, N, W, H = 256, 2000, 1241, 376
ctx = dr3.Context(0)
frames, base = bench.stream_frames(synth, B + 1, W, H)
batch = dr3.FrameBatch(ctx, W, H, win=(WIN, WIN), max_level=ML, capacity=B + 1)
for s0 in range(0, B + 1, 64):
    batch.upload(s0, np.stack(frames[s0:s0 + 64]))
batch.build(0, B + 1, True)
pts = synth.grid_points(W, H, 15, n=N)
P = np.ascontiguousarray(np.broadcast_to(pts[None], (B, N, 2)), np.float32)
d_prev, d_next = ctx.to_device(P), ctx.to_device(P)
d_st, d_err = ctx.alloc(B * N), ctx.alloc(B * N * 4)
ctx.profile_enable(True)
out = []
for it, eps in ((30, 0.01), (0, 0.0), (30, 0.01)):
    p = dr3.lk_params(win=(WIN, WIN), max_level=ML, max_iters=it, eps=eps, flags=0)
    for rep in range(6):
        batch.track(B, 0, d_prev, d_next, N, d_st, d_err, p)
    ctx.sync(); ctx.profile_reset()
    for rep in range(10):
        batch.track(B, 0, d_prev, d_next, N, d_st, d_err, p)
    ms, n = ctx.profile_read(dr3.PROF_LK)
    out.append("iters %2d: %.3f ms" % (it, ms / n))
for deriv in (True, False):          # the pyramid pass with and without the derivative planes (HIP events around the launches of one pass)
    for rep in range(4):
        batch.build(1, B, deriv)
    ctx.sync(); ctx.profile_reset()
    for rep in range(10):
        batch.build(1, B, deriv)
    ms, n = ctx.profile_read(dr3.PROF_PYR)
    out.append("pyramid pass %s: %.3f ms" % ("+ Scharr" if deriv else "only", ms / max(n, 1)))
print("%-28s win %d levels %d  %s" % (os.path.basename(os.environ.get("DR3HIP_LIB", "libdr3hip.so")), WIN, ML + 1, "   ".join(out)), flush=True)
