"""one rank of the world_size-2 gloo rehearsal of bench.py's multi-GPU protocol (CPU only)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    out_path = sys.argv[1]
    d = bench.Dist(2, backend="gloo")
    intr = d.bcast_intrinsics(bench.INTRINSICS if d.rank == 0 else (0,) * 6)
    calls = {"n": 0}

    def step():                     # rank 1 is the slow one: the job time must be ITS time
        calls["n"] += 1
        time.sleep(0.01 * (1 + 2 * d.rank))

    el = bench.timed_region(d, step, lambda: None, steps=5, warmup=2)
    units = 1000 * (d.rank + 1)     # uneven shards: whole-job rate must use the SUM of units
    rate = bench.aggregate_rate(d, units, el)
    json.dump({"rank": d.rank, "world": d.world, "intr": intr, "elapsed": el, "rate": rate,
               "calls": calls["n"]}, open(out_path, "w"))
    d.close()


if __name__ == "__main__":
    main()
