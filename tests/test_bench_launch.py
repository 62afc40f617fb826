"""The launch path the driver uses: `python bench.py --gpus N ...` with no launcher around it must start N ranks, time the
region as MAX over ranks, sum the units of all ranks and print ONE JSON line from rank 0 with n_gpus == N.  Runs here on the
CPU over gloo with the product package replaced by tests/bench_standin.py (sleeping steps, labelled as such in the line)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
COMMON = ["--steps", "5", "--warmup", "1", "--batch", "2", "--no-warp", "--no-frontend", "--no-host-api", "--no-cpu"]


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.update(DR3_BENCH_STANDIN="tests.bench_standin", DR3_BENCH_BACKEND="gloo", OMP_NUM_THREADS="1", **kw)
    return env


def _run(args, env, timeout=300):
    r = subprocess.run([sys.executable, BENCH] + args, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, lines


def test_gpus_2_launches_two_ranks_and_aggregates():
    r, lines = _run(["--gpus", "2"] + COMMON, _env())
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    assert r.stdout.strip() == lines[0], "nothing but that line on stdout (library banners go to stderr)"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["sequences"] == 2 and d["steps"] == 5 and d["warmup"] == 1
    assert d["data"].startswith("standin") and d["verified_vs_oracle"] is None
    # rank 1 sleeps 30 ms per step, rank 0 10 ms: the job time is rank 1's (MAX), not the mean and not rank 0's
    assert 28.0 < d["ms_per_step"] < 80.0
    units = 2 * d["config"]["frame_pairs_per_step_per_gpu"] * d["config"]["points_per_pair"]        # SUM over both ranks
    assert d["value"] == pytest.approx(units / (d["ms_per_step"] * 1e-3), rel=1e-6)
    assert d["config"]["intrinsics_bcast"][:2] == [718.856, 718.856]


def test_gpus_1_line_is_a_single_rank_without_process_group():
    r, lines = _run(["--gpus", "1"] + COMMON, _env())
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["sequences"] == 1 and 9.0 < d["ms_per_step"] < 40.0


def test_launcher_env_must_agree_with_gpus():
    r, lines = _run(["--gpus", "8"] + COMMON, _env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999"))
    assert r.returncode == 2 and not lines and "disagrees with WORLD_SIZE" in r.stderr


def test_more_ranks_than_gpus_is_refused_not_shared():
    """over RCCL (the default backend) ranks never share a device: with fewer GPUs than ranks every rank exits non-zero
    before any rendezvous, and the parent propagates the code"""
    env = _env()
    del env["DR3_BENCH_BACKEND"]
    r, lines = _run(["--gpus", "2"] + COMMON, env)
    assert r.returncode == 4 and not lines and "dr3hip_device_count() = 0" in r.stderr


def test_canvas_sharded_paste_runs_on_two_ranks_and_equals_the_single_rank_canvas():
    """SURVEY 8(e), warp side: two gloo ranks paste ONE panorama -- each all 8 tiles into its own slab of canvas rows -- and rank 0
    checks the gathered slab hashes against the canvas it pastes alone (the stand-in pastes through the CPU checker)"""
    r, lines = _run(["--gpus", "2", "--sharded-paste", "96x64"] + COMMON, _env())
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(lines[0])
    p = d["paste_canvas_sharded"]
    assert p["slabs"] == 2 and sum(p["rows_per_slab"]) == p["canvas"][1] and p["slabs_equal_the_single_gpu_canvas"] is True
    assert p["scaling"].startswith("strong") and p["tile_mpix_per_s_job"] > 0


def test_single_rank_has_no_sharded_paste_leg():
    r, lines = _run(["--gpus", "1", "--sharded-paste", "96x64"] + COMMON, _env())
    assert r.returncode == 0 and json.loads(lines[0])["paste_canvas_sharded"] is None
