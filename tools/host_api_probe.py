"""which earlier step of bench.py changes the time of the host-buffer LK call (dr3hip_lk_track, PCIe-inclusive)"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dr3 = importlib.import_module("3dr_amd"); synth = importlib.import_module("3dr_amd.synth")
import bench
import torch
TORCH_FIRST = os.environ.get("TORCH_FIRST", "1") == "1"
if TORCH_FIRST:
    torch.cuda.init(); torch.cuda.synchronize()
ctx = dr3.Context(0)
W, H, B = 1241, 376, 256
frames, base = bench.stream_frames(synth, B + 1, W, H, period=9)
pts = synth.grid_points(W, H, 15, n=2000)
p = dr3.lk_params(win=(15, 15), max_level=3, max_iters=30, eps=0.01)


def t(tag, n=40):
    ctx.calcOpticalFlowPyrLK(frames[0], frames[1], pts, None, p)
    ts = []
    for i in range(n):
        t0 = time.perf_counter()
        ctx.calcOpticalFlowPyrLK(frames[i % B], frames[i % B + 1], pts, None, p)
        ts.append((time.perf_counter() - t0) * 1e3)
    ts = np.array(ts)
    print("%-46s mean %.3f  median %.3f  max %.3f ms (call %d)  calls over 1 ms: %s" % (tag, ts.mean(), np.median(ts), ts.max(), ts.argmax(),
          np.flatnonzero(ts > 1.0).tolist()), flush=True)


t("fresh context (torch initialised first: %s)" % TORCH_FIRST)
t("again")
batch = dr3.FrameBatch(ctx, W, H, win=(15, 15), max_level=3, capacity=B + 1)
for s0 in range(0, B + 1, 64):
    batch.upload(s0, np.stack(frames[s0:s0 + 64]))
batch.build(0, B + 1, True); ctx.sync()
t("after a 257-frame resident batch")
a = torch.empty(1 << 30, dtype=torch.uint8, device="cuda:0"); b = torch.empty_like(a); b.copy_(a); torch.cuda.synchronize()
t("after torch allocated and copied 1 GiB")
del a, b
t("after torch freed it (cached)")
from oracle import oracle as O
po = O.lk_params(win=(15, 15), max_level=3, max_iters=30, eps=0.01, flags=0)
O.lk_track(frames[0], frames[1], pts, None, po, order=O.ORDER_WAVE64)
t("after an oracle run (OpenMP threads alive)")
ctx.profile_enable(True); t("profiling on"); ctx.profile_enable(False)
t("profiling off")
