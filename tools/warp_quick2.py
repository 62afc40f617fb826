"""8UC3 legs of bench.warp_leg alone (configs[2] tiles and the KITTI-sized frames): kernel ms per library variant"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
import bench
dr3 = importlib.import_module("3dr_amd")
synth = importlib.import_module("3dr_amd.synth")
dist = bench.Dist(1, backend=None)
ctx = dr3.Context(0)
TW_, TH_, NT_ = 4000, 3000, 8
tile = bench.cheap_tile(synth, TW_, TH_)
srcs = [ctx.to_device(tile) for _ in range(NT_)]
dsts = [ctx.alloc(TH_ * TW_ * 3) for _ in range(NT_)]
jobs = []
for i in range(NT_):
    M = synth.tile_homography(i, TW_, TH_)
    M[0, 2] -= 60.0; M[1, 2] -= 60.0
    jobs.append(dict(src=srcs[i].ptr, sw=TW_, sh=TH_, sstride=TW_ * 3, dst=dsts[i].ptr, dw=TW_, dh=TH_, dstride=TW_ * 3, M=M))
ctx.profile_enable(True)
dr3.warp_batch(ctx, jobs, dr3.T_8UC3); ctx.sync(); ctx.profile_reset()
for _ in range(10):
    dr3.warp_batch(ctx, jobs, dr3.T_8UC3)
ctx.sync()
ms, n = ctx.profile_read(dr3.PROF_WARP)
k = bench.kitti_warp_leg(dist, ctx, dr3, synth, ctx.sync, tile, 10) if os.environ.get("WQ_KITTI", "1") == "1" else {"kernel_ms": 0.0, "roofline": {"frac": 0.0}, "verified_vs_oracle": None}
print("plan", dr3.warp_plan(jobs[0], dr3.T_8UC3), end=" ")
print("8uc3 %.4f ms frac %.3f | kitti %.4f ms frac %.3f ok %s | %s" % (ms / n, 576e6 / (ms / n * 1e-3) / 8e12, k["kernel_ms"], k["roofline"]["frac"], k["verified_vs_oracle"],
      os.path.basename(os.environ.get("DR3HIP_LIB", ""))))
ctx.close()
