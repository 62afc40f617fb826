"""experiment: the bench step (pyramid pass + tracker over 256 pairs) on ONE stream versus two independent frame batches on two
streams of the same GPU (the pyramid pass is HBM-bound, the tracker VALU-bound: do they overlap?).  prints ms per step for both."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dr3 = importlib.import_module("3dr_amd"); synth = importlib.import_module("3dr_amd.synth")
B, N, W, H, WIN, ML = 256, 2000, 1241, 376, 15, 3
frames, base = bench.stream_frames(synth, B + 1, W, H)
pts = synth.grid_points(W, H, 15, n=N)
P = np.ascontiguousarray(np.broadcast_to(pts[None], (B, N, 2)), np.float32)
p = dr3.lk_params(win=(WIN, WIN), max_level=ML, max_iters=30, eps=0.01, flags=0)


def make(ctx):
    b = dr3.FrameBatch(ctx, W, H, win=(WIN, WIN), max_level=ML, capacity=B + 1)
    for s0 in range(0, B + 1, 64):
        b.upload(s0, np.stack(frames[s0:s0 + 64]))
    return b, ctx.to_device(P), ctx.to_device(P), ctx.alloc(B * N), ctx.alloc(B * N * 4)


def step(b, d_prev, d_next, d_st, d_err):
    b.build(0, B + 1, True)
    b.track(B, 0, d_prev, d_next, N, d_st, d_err, p)


ctxs = [dr3.Context(0), dr3.Context(0)]
sets = [make(c) for c in ctxs]
K = 20
for c, s in zip(ctxs, sets):
    for _ in range(3):
        step(*s)
    c.sync()
t0 = time.perf_counter()
for _ in range(2 * K):
    step(*sets[0])
ctxs[0].sync()
serial = (time.perf_counter() - t0) / (2 * K) * 1e3
t0 = time.perf_counter()
for _ in range(K):
    step(*sets[0]); step(*sets[1])
ctxs[0].sync(); ctxs[1].sync()
two = (time.perf_counter() - t0) / (2 * K) * 1e3
print("one stream %.4f ms per step   two streams %.4f ms per step   (%.1f %%)" % (serial, two, 100 * (serial / two - 1)), flush=True)
