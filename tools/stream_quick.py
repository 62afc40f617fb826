"""bench.stream_leg alone: the headline step fed from the host (PCIe inclusive), with a resident run of the same frames to compare"""
import importlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
import bench
dr3 = importlib.import_module("3dr_amd"); synth = importlib.import_module("3dr_amd.synth")
dist = bench.Dist(1, backend=None)
ctx = dr3.Context(0)
B, N, W, H = 256, 2000, 1241, 376
frames, base = bench.stream_frames(synth, B + 1, W, H)
batch = dr3.FrameBatch(ctx, W, H, win=(bench.WIN, bench.WIN), max_level=bench.MAX_LEVEL, capacity=B + 1)
for s0 in range(0, B + 1, 64):
    batch.upload(s0, np.stack(frames[s0:s0 + 64]))
batch.build(0, B + 1, True)
pts = synth.grid_points(W, H, 15, n=N)
P = np.ascontiguousarray(np.broadcast_to(pts[None], (B, N, 2)), np.float32)
d_prev, d_next = ctx.to_device(P), ctx.to_device(P)
d_st, d_err = ctx.alloc(B * N), ctx.alloc(B * N * 4)
params = dr3.lk_params(win=(bench.WIN, bench.WIN), max_level=bench.MAX_LEVEL, max_iters=bench.MAX_ITERS, eps=bench.EPS, flags=0)
batch.track(B, 0, d_prev, d_next, N, d_st, d_err, params); ctx.sync()
rx, rs = d_next.download((B, N, 2), np.float32), d_st.download((B, N), np.uint8)
batch.close()
def sync():
    ctx.sync(); torch.cuda.synchronize()
r = bench.stream_leg(dist, ctx, dr3, frames, P, B, N, (W, H), params, sync, 20, rx, rs, True)
print(json.dumps(r))
ctx.close()
