"""the detector of the front-end leg alone (256 frames 1241x376, 3 half-sample levels, FAST-10 + grid-best Shi-Tomasi), for rocprofv3"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dr3 = importlib.import_module("3dr_amd"); synth = importlib.import_module("3dr_amd.synth")
import bench
B, W, H = 256, 1241, 376
ctx = dr3.Context(0)
frames, _ = bench.stream_frames(synth, B, W, H, period=9)
batch = dr3.FrameBatch(ctx, W, H, win=(15, 15), max_level=3, capacity=B)
for s0 in range(0, B, 64):
    batch.upload(s0, np.stack(frames[s0:s0 + 64]))
n_cells = 83 * 26
d_pts = ctx.alloc(B * n_cells * 8)
for _ in range(3):
    batch.detect(0, B, 3, 15, 20, 20.0, d_xy=d_pts)
ctx.sync()
ctx.profile_enable(True); ctx.profile_reset()
for _ in range(int(os.environ.get("N", "10"))):
    batch.detect(0, B, 3, 15, 20, 20.0, d_xy=d_pts)
ctx.sync()
ms, n = ctx.profile_read(dr3.PROF_FAST)
print("detect %.4f ms per batch of %d frames (%d launches)" % (ms / n, B, n))
