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
    # every roofline fraction in the line is a fraction: a value above 1 means a wrong denominator (round 1's LK figure was)
    def fracs(o, path=""):
        if isinstance(o, dict):
            for k, v in o.items():
                if k == "frac" and v is not None:
                    assert 0.0 < v <= 1.0, (path, v)
                fracs(v, path + "/" + k)
    fracs(d)
    vi = d["lk_kernel"]["valu_issue"]
    if vi is not None:                                                        # None when the profile was made for other kernel sources
        assert vi["frac"] == pytest.approx(vi["achieved"] / vi["peak"], rel=1e-9) and 2.0 <= vi["avg_issue_clocks_per_valu"] <= 8.0
        assert vi["peak"] == pytest.approx(1024 * 2.4 / vi["avg_issue_clocks_per_valu"], rel=1e-9)
    # every leg that ran carries its check against the oracle
    assert d["raster_order_agreement"]["same_path_max_dev_px"] <= 1.25e-3 and 0.99 <= d["raster_within_1e-3_frac"] <= 1.0
    if os.path.basename(os.path.dirname(path)) >= "r04":
        # north_star's 1e-3 px bar against the order the reference's x86-64 OpenCV executes, on the metric's configuration AND on
        # configs[4]'s pair (the one round 3 found over the bar against the scalar order): every converged point, no exception list
        assert d["pre_roll_steps"] == 256
        recs = [d["summation_order_agreement"], d["hd_share"]["summation_order_agreement"]]
        recs += [leg["summation_order_agreement"] for leg in d["lk_other_windows"].values() if leg.get("summation_order_agreement")]    # configs[0], literals
        for rec in recs:
            ks = rec["kernel_vs_sse2"]
            assert ks["status_equal"] and ks["same_path_within_tol"] and ks["converged_within_tol"], ks
            assert ks["tracked_both"] - ks["clamped"] >= 0.99 * ks["tracked_both"]
        assert d["hd_share"]["verified_vs_oracle"] is True and 0 < d["hd_share"]["roofline"]["frac"] <= 1.0
        assert d["cpu_baseline"]["port_opt_equals_port_bit_for_bit"] is True
        assert d["cpu_baseline"]["value"] >= d["cpu_baseline"]["readable_port"]["value"]       # the >= 50x target is quoted against the faster port
    for leg in ("8uc3", "32fc4", "kitti_frames_8uc3", "paste", "stitch"):
        assert d["warp"][leg]["verified_vs_oracle"] is True, leg
    for leg in d["lk_other_windows"].values():
        assert leg["verified_vs_oracle"] is True
    assert d["sparse_align"]["verified_vs_cpu_statement"] is True


def test_live_timing_agrees_with_rocprof_summary_of_the_same_run():
    """profiles/<round>/bench_line_under_rocprof.json is the line bench.py printed INSIDE the rocprofv3 --kernel-trace --stats
    run whose summary is kernel_stats.csv: the live HIP-event averages and the profiler's durations of that one run agree.  The
    live figures cover the timed launches only, so they are compared with the profiler's mean over exactly those launches
    (kernel_trace_timed.json, from the same trace); kernel_stats.csv's average also contains the slower pre-roll launches and
    must not be below it"""
    _, path = _newest()
    here = os.path.dirname(path)
    line, stats, timed = (os.path.join(here, f) for f in ("bench_line_under_rocprof.json", "kernel_stats.csv", "kernel_trace_timed.json"))
    if not all(os.path.exists(f) for f in (line, stats, timed)):
        pytest.skip("no rocprof run recorded beside the bench line")
    import csv
    d = json.loads(open(line).read())
    t = json.load(open(timed))["kernels"]
    l0 = [v for k, v in t.items() if "pyr_level_kernel<true, true, 0>" in k]
    lk = [v for k, v in t.items() if d["lk_kernel"]["kernel"] in k]
    assert l0 and lk and lk[0]["timed_launches"] == d["steps"]
    # a HIP event pair around a launch also holds the few microseconds between the events and the kernel: 5 % of a 0.12 ms kernel
    assert l0[0]["mean_ms_timed"] <= d["roofline"]["avg_launch_ms"] <= 1.07 * l0[0]["mean_ms_timed"]
    assert d["lk_kernel"]["avg_launch_ms"] == pytest.approx(lk[0]["mean_ms_timed"], rel=0.03)
    rows = {r["Name"]: float(r["AverageNs"]) * 1e-6 for r in csv.DictReader(open(stats))}
    all_lk = [v for k, v in rows.items() if d["lk_kernel"]["kernel"] in k]
    # the stats file averages every launch of the run: the 256 pre-roll launches at steady state and the few timed ones right behind a
    # device sync, which run a few per cent slower (tools/step_decay.sh) -- the two means agree within that
    assert all_lk and all_lk[0] == pytest.approx(lk[0]["mean_ms_all"], rel=1e-3) and all_lk[0] == pytest.approx(lk[0]["mean_ms_timed"], rel=0.08)


def test_committed_profiles_were_made_from_the_kernel_sources_in_the_tree():
    """the counters the bench line quotes (PMC traffic, the tracker's VALU counts and opcode mix) carry the hash of the kernel
    sources they were collected from; the newest committed set must be that of the sources now in the tree, or the next bench
    run would report `null` for them (bench.newest_profile)"""
    import importlib.util
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench
    spec = importlib.util.spec_from_file_location("summarize_profiles", os.path.join(ROOT, "tools", "summarize_profiles.py"))
    S = importlib.util.module_from_spec(spec); spec.loader.exec_module(S)
    d, path = _newest()
    prof = os.path.dirname(path)
    lk = bench.kernel_source_hash(S.LK_HASH_FILES)
    tr = bench.kernel_source_hash(S.TRAFFIC_HASH_FILES)
    for f, want in (("pmc_lk.json", lk), ("valu_mix.json", lk), ("pmc_traffic.json", tr), ("pmc_traffic_warp.json", tr)):
        assert json.load(open(os.path.join(prof, f)))["source_hash"] == want, f
    assert d["lk_kernel"]["valu_issue"] is not None and d["roofline"]["traffic"] is not None
    assert os.path.dirname(os.path.join(ROOT, d["roofline"]["traffic_source"])) == prof


def test_strip_kernel_launch_shape_matches_the_recorded_traffic_shapes():
    """bench.strip_grid_threads restates launch_warp's grid of warp_tile_8uc3_kernel (64 x 64 tiles, padded to 8 x run blocks, 256 threads):
    the per-shape PMC traffic is looked up by it, so it must name the shapes the committed summary recorded"""
    import bench
    assert bench.strip_grid_threads(4000, 3000, 8) == 6094848          # 63 x 47 tiles -> 2976 blocks (runs of 4 per XCD) x 8 jobs
    assert bench.strip_grid_threads(1241, 376, 256) == 7864320         # 20 x 6 tiles, raster order, x 256 jobs
    prof, _ = bench.newest_profile("pmc_traffic_warp.json")
    if prof:
        strip = [v for k, v in prof.get("kernels", prof).items() if isinstance(v, dict) and "warp_tile_8uc3_kernel" in k]
        if strip:
            assert {"6094848", "7864320"} <= set(strip[0].get("shapes", {}).keys())


def test_hd_workload_line_meets_the_bar_too():
    """`python bench.py --workload hd` (configs[4]'s one-GPU share, the long form of `hd_share`): pair 0 of ITS stream -- the pair round 3
    found over the bar against OpenCV's scalar order -- bit-exact in the kernel's order and within 1e-3 px of the x86-64 order on every
    converged point"""
    _, path = _newest()
    f = os.path.join(os.path.dirname(path), "bench_hd_run.json")
    if not os.path.exists(f) or os.path.basename(os.path.dirname(path)) < "r04":
        pytest.skip("no hd line recorded beside the default one")
    d = json.loads(open(f).read().strip().splitlines()[-1])
    assert "1920x1080" in d["config"]["workload"] and d["verified_vs_oracle"] is True
    ks = d["summation_order_agreement"]["kernel_vs_sse2"]
    assert ks["status_equal"] and ks["same_path_within_tol"] and ks["converged_within_tol"], ks
    assert 0 < d["roofline"]["frac"] <= 1.0
