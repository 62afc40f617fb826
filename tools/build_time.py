"""experiment helper: time of one pyramid + gradient pass (dr3hip_batch_build, 256 KITTI frames) per profile slot, for DR3HIP_LIB"""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dr3 = importlib.import_module("3dr_amd"); synth = importlib.import_module("3dr_amd.synth")
B, W, H = 256, 1241, 376
ctx = dr3.Context(0)
frames, base = bench.stream_frames(synth, 9, W, H)
batch = dr3.FrameBatch(ctx, W, H, win=(15, 15), max_level=3, capacity=B + 1)
for s0 in range(0, B + 1, 64):
    batch.upload(s0, np.stack([frames[i % 9] for i in range(s0, min(s0 + 64, B + 1))]))
for rep in range(5):
    batch.build(1, B, True)
ctx.sync(); ctx.profile_enable(True); ctx.profile_reset()
for rep in range(20):
    batch.build(1, B, True)
ms, n = ctx.profile_read(dr3.PROF_PYR); ms0, n0 = ctx.profile_read(dr3.PROF_PYR_L0)
print("%-28s pass %.4f ms  level-0 kernel %.4f ms" % (os.path.basename(os.environ.get("DR3HIP_LIB", "libdr3hip.so")), ms / n, ms0 / n0), flush=True)
