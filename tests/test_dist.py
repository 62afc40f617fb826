"""world_size-2 rehearsal (gloo, CPU) of the only multi-GPU logic the path has: broadcast of the
intrinsics from rank 0, barrier-bracketed timing with MAX over ranks, whole-job rate from the SUM
of every rank's units (independent sequences per GPU, no data-path collective)."""
import json
import os
import socket
import subprocess
import sys

import bench


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _run_two_ranks(tmp_path, extra_env=None):
    port = _free_port()
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_worker.py")
    procs, outs = [], []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1", **(extra_env or {}))
        outs.append(str(tmp_path / ("r%d.json" % r)))
        procs.append(subprocess.Popen([sys.executable, worker, outs[-1]], env=env))
    for p in procs:
        assert p.wait(timeout=180) == 0
    return [json.load(open(o)) for o in outs]


def test_ref_pyramid_leg_survives_a_failing_rank(tmp_path):
    """bench.ref_bcast_leg: a rank whose communicator init fails must neither hang the other rank nor go unreported"""
    res = _run_two_ranks(tmp_path, {"DR3_TEST_FAIL_RANK": "1"})
    for r in res:
        assert r["leg"]["ranks_with_error"] == 1 and r["leg"]["identical_on_all_ranks"] is False
    assert "injected" in res[1]["leg"]["error"] and res[0]["leg"]["error"] is None


def test_two_rank_protocol(tmp_path):
    res = _run_two_ranks(tmp_path)
    for r in res:                                                     # the reference-pyramid leg: same planes everywhere
        assert r["leg"]["ranks_with_error"] == 0 and r["leg"]["identical_on_all_ranks"] is True and r["leg"]["bytes_per_rank"] == 4 * 6 * 8
    for r in res:
        assert r["world"] == 2 and r["calls"] == 7                    # 2 warm-up + exactly 5 timed steps
        assert tuple(r["intr"]) == tuple(bench.INTRINSICS)            # rank 1 received rank 0's values
    # both ranks agree on the job time and it is the slow rank's: 5 steps x 30 ms
    assert abs(res[0]["elapsed"] - res[1]["elapsed"]) < 1e-9
    assert 0.14 < res[0]["elapsed"] < 0.6
    assert abs(res[0]["rate"] - 3000 / res[0]["elapsed"]) < 1e-6 * res[0]["rate"]


def test_single_process_paths_and_byte_formulas():
    d = bench.Dist(1)
    assert not d.pg and d.bcast_intrinsics((1, 2, 3)) == (1, 2, 3)
    assert d.max_over_ranks(2.5) == 2.5 and d.sum_over_ranks(4) == 4.0
    # 1241x376, L=4 -> 3 865 562 B/frame; L=3 -> 3 792 336; 1920x1080 L=5 -> 17 253 360
    assert bench.pyr_grad_bytes_per_frame(1241, 376, 4) == 3865562
    assert bench.pyr_grad_bytes_per_frame(1241, 376, 3) == 3792336
    assert bench.pyr_grad_bytes_per_frame(1920, 1080, 5) == 17253360
    assert bench.pyr_grad_level0_bytes_per_frame(1241, 376) == 2 * 466616 + 116748 + 4 * 466616
    assert bench.lk_unique_bytes_per_pair(2000, 4, 15) == 2000 * 4 * 256 * 6 + 21 * 2000   # 12.33 MB
    assert bench.host_cores() >= 1
