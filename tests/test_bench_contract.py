"""The newest committed bench line (profiles/*/bench_default_run.json, written by `python bench.py` on an MI355X) carries
every field of the driver's contract, with consistent arithmetic.  CPU-only: it checks the recorded line, not the GPU."""
import glob
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _newest():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_v*", "bench_default_run.json")))
    if not files:
        pytest.skip("no committed bench line")
    return json.load(open(files[-1])), files[-1]


def test_contract_fields_and_arithmetic():
    d, path = _newest()
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                 ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict)):
        assert isinstance(d[k], t), (path, k)
    assert "vs_baseline" in d and d["vs_baseline"] is None                    # BASELINE.md publishes no number for this metric
    assert d["scaling"] == "weak" and d["higher_is_better"] is True and d["data"] == "synthetic" and d["dtype"] == "u8"
    cfg = d["config"]
    assert "workload" in cfg and "model" not in cfg and "1241x376" in cfg["workload"]
    units = cfg["frame_pairs_per_step_per_gpu"] * cfg["points_per_pair"] * d["n_gpus"]
    assert d["value"] == pytest.approx(units / (d["ms_per_step"] * 1e-3), rel=1e-6)       # whole-job units / max-over-ranks time
    assert d["verified_vs_oracle"] is True
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] == 8000.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-9)
    assert r["achieved"] == pytest.approx(r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9, rel=1e-6)
    assert r["traffic"] is None or 0.5 * r["algorithmic_bytes_per_launch"] < r["traffic"] < 2 * r["algorithmic_bytes_per_launch"]
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and isinstance(c["sample"], str) and c["unit"] == d["unit"]
    assert isinstance(d["warped_mpix_per_sec_1241x376"], float)               # the second half of BASELINE's metric


def test_live_timing_agrees_with_rocprof_summary_of_the_same_run():
    """profiles/<round>/bench_line_under_rocprof.json is the line bench.py printed INSIDE the rocprofv3 --kernel-trace --stats
    run whose summary is kernel_stats.csv: the live HIP-event averages and the profiler's averages of that one run agree"""
    _, path = _newest()
    here = os.path.dirname(path)
    line, stats = os.path.join(here, "bench_line_under_rocprof.json"), os.path.join(here, "kernel_stats.csv")
    if not (os.path.exists(line) and os.path.exists(stats)):
        pytest.skip("no rocprof run recorded beside the bench line")
    import csv
    d = json.loads(open(line).read())
    rows = {r["Name"]: float(r["AverageNs"]) * 1e-6 for r in csv.DictReader(open(stats))}
    l0 = [v for k, v in rows.items() if "pyr_level_kernel<true, true, 0>" in k]
    lk = [v for k, v in rows.items() if "lk_track_kernel" in k]
    assert l0 and lk
    assert d["roofline"]["avg_launch_ms"] == pytest.approx(l0[0], rel=0.05)
    assert d["lk_kernel"]["avg_launch_ms"] == pytest.approx(lk[0], rel=0.05)
