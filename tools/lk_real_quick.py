"""tracker on the reference's own KITTI pair at the headline launch shape (bench.lk_real_pair_leg alone) per library variant"""
import importlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import bench
dr3 = importlib.import_module("3dr_amd")
dist = bench.Dist(1, backend=None)
ctx = dr3.Context(0)
r = bench.lk_real_pair_leg(dist, ctx, dr3, ctx.sync, True)
print(" | ".join("%s %.3f ms it/pt %.1f slowest-of-%d %.1f clamp %.3f ok %s" % (k, v["kernel_ms"], v["mean_iteration_levels_per_point"], v["points_per_wave"],
      v["mean_iteration_levels_of_slowest_in_group"], v["points_at_the_iteration_clamp_frac"], v["verified_vs_oracle"]) for k, v in r.items() if isinstance(v, dict)),
      os.path.basename(os.environ.get("DR3HIP_LIB", "")))
ctx.close()
