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
    # the reference-pyramid broadcast leg of bench.py with the GPU library replaced by stand-ins: what is rehearsed is the
    # protocol -- the id hand-over and the rule that every torch.distributed call runs on every rank whatever happened
    # locally.  DR3_TEST_FAIL_RANK makes that rank's communicator init raise.
    import numpy as np
    import torch

    fail_rank = int(os.environ.get("DR3_TEST_FAIL_RANK", "-1"))

    class Ctx:
        def comm_unique_id(self): return bytes(range(128))
        def comm_init(self, uid, world, rank):
            assert uid == bytes(range(128)) and world == 2
            if rank == fail_rank:
                raise RuntimeError("injected communicator failure")
        def comm_destroy(self): pass
        def timer_start(self): self.t = time.perf_counter()
        def timer_stop(self): return (time.perf_counter() - self.t) * 1e3

    class Batch:
        def __init__(self): self.plane = None
        def upload(self, slot, frame): self.plane = np.array(frame, np.int64)
        def build(self, *a): pass
        def bcast_slot(self, slot, root=0, with_deriv=True):
            if fail_rank >= 0:                 # the real RCCL call would never be reached by the failed rank; the healthy one
                return 0                       # must not wait for it either (a real hang is what the bench's watchdog is for)
            t = torch.from_numpy(self.plane)
            d.dist.broadcast(t, src=root)
            return int(t.numel() * 8)
        def download_image(self, slot, level): return self.plane
        def download_deriv(self, slot, level): return self.plane

    class Dr3:
        COMM_ID_BYTES = 128

    frames = [np.full((4, 6), 7 + 0 * d.rank, np.uint8)]
    leg = bench.ref_bcast_leg(d, Ctx(), Dr3, Batch(), frames, 1, lambda: None)
    json.dump({"rank": d.rank, "world": d.world, "intr": intr, "elapsed": el, "rate": rate,
               "calls": calls["n"], "leg": leg}, open(out_path, "w"))
    d.close()


if __name__ == "__main__":
    main()
