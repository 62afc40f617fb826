"""experiment: LK kernel time with the bench's points vs the same number of points kept away from every level's border"""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dr3 = importlib.import_module("3dr_amd"); synth = importlib.import_module("3dr_amd.synth")
B, N, W, H = 256, 2000, 1241, 376
ctx = dr3.Context(0)
frames, base = bench.stream_frames(synth, B + 1, W, H)
batch = dr3.FrameBatch(ctx, W, H, win=(15, 15), max_level=3, capacity=B + 1)
for s0 in range(0, B + 1, 64):
    batch.upload(s0, np.stack(frames[s0:s0 + 64]))
batch.build(0, B + 1, True)
pts = synth.grid_points(W, H, 15, n=N)
inner = pts.copy()
inner[:, 0] = 72 + (pts[:, 0] - pts[:, 0].min()) / np.ptp(pts[:, 0]) * (W - 144)
inner[:, 1] = 72 + (pts[:, 1] - pts[:, 1].min()) / np.ptp(pts[:, 1]) * (H - 144)
p = dr3.lk_params(win=(15, 15), max_level=3, max_iters=30, eps=0.01, flags=0)
for name, q in (("bench points", pts), ("interior points", inner.astype(np.float32))):
    P = np.ascontiguousarray(np.broadcast_to(q[None], (B, N, 2)), np.float32)
    d_prev, d_next = ctx.to_device(P), ctx.to_device(P)
    d_st, d_err = ctx.alloc(B * N), ctx.alloc(B * N * 4)
    ctx.profile_enable(True)
    for rep in range(2):
        batch.track(B, 0, d_prev, d_next, N, d_st, d_err, p)
    ctx.sync(); ctx.profile_reset()
    for rep in range(5):
        batch.track(B, 0, d_prev, d_next, N, d_st, d_err, p)
    ms, n = ctx.profile_read(dr3.PROF_LK)
    print("%-16s %.3f ms  tracked %.3f" % (name, ms / n, d_st.download((B, N), np.uint8).mean()), flush=True)
