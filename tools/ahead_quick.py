"""the configs[1] step with the next group's pyramid levels 1.. built under the tracker (dr3hip_batch_build_ahead) against the serial
step, same process, interleaved.  usage: python tools/ahead_quick.py"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dr3 = importlib.import_module("3dr_amd"); synth = importlib.import_module("3dr_amd.synth")
B, N, W, H = 256, 2000, 1241, 376
ctx = dr3.Context(0)
frames, _ = bench.stream_frames(synth, B + 1, W, H)
G = B + 1
batch = dr3.FrameBatch(ctx, W, H, win=(15, 15), max_level=3, capacity=2 * G)
for g in range(2):
    for s0 in range(0, G, 64):
        batch.upload(g * G + s0, np.stack(frames[s0:s0 + 64]))
batch.build(0, 2 * G, True)
pts = synth.grid_points(W, H, 15, n=N)
P = np.ascontiguousarray(np.broadcast_to(pts[None], (B, N, 2)), np.float32)
d_prev, d_next = ctx.to_device(P), [ctx.to_device(P), ctx.to_device(P)]
d_st, d_err = [ctx.alloc(B * N), ctx.alloc(B * N)], ctx.alloc(B * N * 4)
p = dr3.lk_params(win=(15, 15), max_level=3, max_iters=30, eps=0.01)
it = [0]
def serial():
    g = it[0] % 2; it[0] += 1
    batch.build(g * G + 1, B, True)
    batch.track(B, g * G, d_prev, d_next[g], N, d_st[g], d_err, p)
def ahead():
    g = it[0] % 2; it[0] += 1
    batch.build_ahead((1 - g) * G + 1, B, True)
    batch.track(B, g * G, d_prev, d_next[g], N, d_st[g], d_err, p)
ref = None
for name, fn in (("serial", serial), ("ahead", ahead), ("serial", serial), ("ahead", ahead)):
    for _ in range(200): fn()
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(100): fn()
    ctx.sync(); el = time.perf_counter() - t0
    xy = d_next[0].download((B, N, 2), np.float32)
    if ref is None: ref = xy.copy()
    print("%-7s %.4f ms per step  %.1f M keypoints/s  same results %s" % (name, el / 100 * 1e3, B * N / (el / 100) / 1e6, np.array_equal(xy.view(np.uint32), ref.view(np.uint32))), flush=True)
ctx.close()
