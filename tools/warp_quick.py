"""warp leg of bench.py alone (configs[2]); prints kernel ms and roofline fraction per type"""
import importlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (one HIP runtime per process, as in bench.py)
import bench
dr3 = importlib.import_module("3dr_amd")
synth = importlib.import_module("3dr_amd.synth")
dist = bench.Dist(1, backend=None)
ctx = dr3.Context(0)
r = bench.warp_leg(dist, ctx, dr3, synth, ctx.sync, steps=10)
print(" ".join("%s %.4f ms frac %.3f" % (k, v["kernel_ms"], v["roofline"]["frac"]) for k, v in r.items() if isinstance(v, dict) and "kernel_ms" in v), "| paste add %.4f ms frac %.3f norm %.4f ms frac %.3f step %.3f ms" % (r["paste"]["add_kernel_ms"], r["paste"]["roofline_add"]["frac"], r["paste"]["roofline_normalize"]["kernel_ms"], r["paste"]["roofline_normalize"]["frac"], r["paste"]["step_ms"]), "stitch %.4f ms" % r["stitch"]["kernel_ms"], "half %.4f ms frac %.3f" % (r["halfsample"]["kernel_ms"], r["halfsample"]["roofline"]["frac"]), "remap %.4f ms frac %.3f / fused maps %.4f ms frac %.3f ok %s" % (r["remap"]["maps_from_memory"]["kernel_ms"], r["remap"]["maps_from_memory"]["roofline"]["frac"], r["remap"]["spherical_maps_in_kernel"]["kernel_ms"], r["remap"]["spherical_maps_in_kernel"]["roofline"]["frac"], r["remap"]["verified_vs_oracle"]), "verified", [r[k].get("verified_vs_oracle") for k in ("8uc3", "32fc4", "kitti_frames_8uc3", "paste", "stitch", "halfsample")],
      os.path.basename(os.environ.get("DR3HIP_LIB", "")))
ctx.close()
