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
P = np.ascontiguousarray(np.broadcast_to(pts[None], (B, N, 2)), np.float32)
d_prev, d_next = ctx.to_device(P), ctx.to_device(P)
d_st, d_err, d_it = ctx.alloc(B * N), ctx.alloc(B * N * 4), ctx.alloc(B * N * 16)
p = dr3.lk_params(win=(15, 15), max_level=3, max_iters=30, eps=0.01)
for rep in range(3):
    batch.track(B, 0, d_prev, d_next, N, d_st, d_err, p, d_iters=d_it)
ctx.sync()
it = d_it.download((B, N, 4), np.int32).astype(np.float64)
m = it.reshape(-1, 4).mean(0)
print("cycles per wave (sum over 4 levels): prologue+stage %.0f | extract %.0f | extract+reduce+solve %.0f | iterations %.0f" % tuple(m))
