"""experiment: the tracker at the bench's shape with every pair reading the SAME two frame slots (all loads cache hits) against the
real stream of 257 frames: how much of the launch time is memory"""
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
same_prev, same_next = ctx.to_device(np.zeros(B, np.int32)), ctx.to_device(np.ones(B, np.int32))
ctx.profile_enable(True)
for name, kw in (("stream of 257 frames", {}), ("every pair = slots (0, 1)", dict(d_prev_slot=same_prev, d_next_slot=same_next)), ("stream of 257 frames", {})):
    for rep in range(6):
        batch.track(B, 0, d_prev, d_next, N, d_st, d_err, p, **kw)
    ctx.sync(); ctx.profile_reset()
    for rep in range(10):
        batch.track(B, 0, d_prev, d_next, N, d_st, d_err, p, **kw)
    ms, n = ctx.profile_read(dr3.PROF_LK)
    print("%-28s %.3f ms per launch" % (name, ms / n), flush=True)
