"""experiment: LK kernel time vs (max_level, max_iters) on the bench workload"""
import importlib, sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dr3 = importlib.import_module("3dr_amd"); synth = importlib.import_module("3dr_amd.synth")
B, N = 256, 2000
W, H = 1241, 376
ctx = dr3.Context(0)
frames, base = bench.stream_frames(synth, B + 1, W, H)
batch = dr3.FrameBatch(ctx, W, H, win=(15, 15), max_level=3, capacity=B + 1)
for s0 in range(0, B + 1, 64):
    batch.upload(s0, np.stack(frames[s0:s0 + 64]))
batch.build(0, B + 1, True)
pts = synth.grid_points(W, H, 15, n=N)
P = np.ascontiguousarray(np.broadcast_to(pts[None], (B, N, 2)), np.float32)
d_prev, d_next = ctx.to_device(P), ctx.to_device(P)
d_st, d_err = ctx.alloc(B * N), ctx.alloc(B * N * 4)
ctx.profile_enable(True)
for ml in (3, 2, 1, 0):
    for it in (0, 1, 2, 3, 30):
        p = dr3.lk_params(win=(15, 15), max_level=ml, max_iters=it, eps=0.01 if it == 30 else 0.0, flags=0)
        for rep in range(2):
            batch.track(B, 0, d_prev, d_next, N, d_st, d_err, p)
        ctx.sync(); ctx.profile_reset()
        for rep in range(5):
            batch.track(B, 0, d_prev, d_next, N, d_st, d_err, p)
        ms, n = ctx.profile_read(dr3.PROF_LK)
        print("levels %d iters %2d : %.3f ms" % (ml + 1, it, ms / n), flush=True)
