"""experiment: how long ONE pair (2000 points = 500 waves, the reference's own call shape) takes as a kernel, alone on an idle GPU
and right behind a full-size launch (clocks up)"""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dr3 = importlib.import_module("3dr_amd"); synth = importlib.import_module("3dr_amd.synth")
B, N, W, H, WIN, ML = 256, 2000, 1241, 376, 15, 3
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
p = dr3.lk_params(win=(WIN, WIN), max_level=ML, max_iters=30, eps=0.01, flags=0)
ctx.profile_enable(True)
for n_pairs in (1, 2, 4, 8, 16, 64, 256):
    for rep in range(5):
        batch.track(n_pairs, 0, d_prev, d_next, N, d_st, d_err, p)
    ctx.sync(); ctx.profile_reset()
    for rep in range(20):
        batch.track(n_pairs, 0, d_prev, d_next, N, d_st, d_err, p)
    ms, n = ctx.profile_read(dr3.PROF_LK)
    print("%3d pairs alone: %.3f ms per launch" % (n_pairs, ms / n), flush=True)
ctx.sync(); ctx.profile_reset()
tot = 0.0
for rep in range(10):
    batch.track(256, 0, d_prev, d_next, N, d_st, d_err, p)
    ctx.sync(); a, na = ctx.profile_read(dr3.PROF_LK)
    batch.track(1, 0, d_prev, d_next, N, d_st, d_err, p)
    ctx.sync(); b, nb = ctx.profile_read(dr3.PROF_LK)
    tot += b - a
print("  1 pair right behind a 256-pair launch: %.3f ms" % (tot / 10), flush=True)
