#!/usr/bin/env python3
"""Condense the rocprofv3 output that tools/profile_bench.sh leaves under gpurun_out/ into the small,
tracked files of profiles/<name>/:
  kernel_stats.csv              -- rocprofv3 --kernel-trace --stats summary (copied as is)
  pmc_fetch_write_summary.json  -- per kernel: dispatches, median / max FETCH_SIZE and WRITE_SIZE (KB)
  pmc_traffic.json              -- per kernel HBM traffic per launch in bytes, corrected as
                                   /opt/skills/guides/MI355X_MICROARCH.md "HBM" prescribes: FETCH_SIZE x2
                                   on gfx950 (128-B requests tallied at 64 B), WRITE_SIZE as is, counters
                                   collected in separate --pmc passes.  bench.py reads this file for
                                   roofline.traffic.
usage: tools/summarize_profiles.py gpurun_out profiles/r01_v4
"""
import csv
import re
import glob
import hashlib
import json
import os
import shutil
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_hash(files):
    """the stamp bench.py compares (bench.kernel_source_hash): a summary is only reported for the kernel sources it was made from"""
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(ROOT, "3dr_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


TRAFFIC_HASH_FILES = ("pyr_kernels.hip", "warp_kernels.hip", "dr3hip_internal.h")
LK_HASH_FILES = ("lk_kernels.hip", "dr3hip_internal.h")


def per_kernel(path, counter, largest_grid_only=False):
    """counter values per kernel; largest_grid_only keeps the dispatches of the largest launch shape of each kernel (a kernel that
    runs at two shapes in one script -- the configs[2] tiles and the KITTI-sized frames -- must not be averaged over both)"""
    acc, grid = {}, {}
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            k, g = row["Kernel_Name"], int(row["Grid_Size"])
            if largest_grid_only:
                if g > grid.get(k, 0):
                    grid[k] = g; acc[k] = []
                if g < grid[k]:
                    continue
            acc.setdefault(k, []).append(float(row["Counter_Value"]))
    return acc


def pyr_pass_bytes(fetch_csv, write_csv):
    """HBM bytes of one pyramid + gradient pass: every (pyramid kernel, launch shape) the timed steps repeat (>= 3 dispatches), median
    FETCH_SIZE x 2 + WRITE_SIZE each, summed; None if a pass is missing"""
    tot = 0.0
    for path, counter, mul in ((fetch_csv, "FETCH_SIZE", 2.0), (write_csv, "WRITE_SIZE", 1.0)):
        if not os.path.exists(path):
            return None
        acc = {}
        for row in csv.DictReader(open(path, newline="")):
            if row["Counter_Name"] == counter and ("pyr_level_kernel" in row["Kernel_Name"] or "pad_reflect_kernel" in row["Kernel_Name"]):
                acc.setdefault((row["Kernel_Name"], int(row["Grid_Size"])), []).append(float(row["Counter_Value"]))
        rep = [statistics.median(v) for v in acc.values() if len(v) >= 3]
        if not rep:
            return None
        tot += sum(rep) * 1024.0 * mul
    return tot


def main(src, dst):
    os.makedirs(dst, exist_ok=True)
    for name in ("hostinfo.txt",):
        p = os.path.join(src, name)
        if os.path.exists(p):
            shutil.copy(p, os.path.join(dst, name))
    ks = os.path.join(src, "prof_r1", "bench_kernel_stats.csv")
    if os.path.exists(ks):
        shutil.copy(ks, os.path.join(dst, "kernel_stats.csv"))
    log = os.path.join(src, "prof_bench.log")
    if os.path.exists(log):
        lines = [l for l in open(log) if l.startswith("{")]
        if lines:
            open(os.path.join(dst, "bench_line_under_rocprof.json"), "w").write(lines[-1])

    # the rocprofv3 average of a kernel covers every launch of the run, the pre-roll and warm-up ones included, which run a few per
    # cent slower (the part is still settling); the bench times the LAST `steps` launches.  Record the mean of exactly those from the
    # same kernel trace, so that the live HIP-event figures can be compared like for like (tests/test_bench_contract.py).
    kt, bl = os.path.join(src, "prof_r1", "bench_kernel_trace.csv"), os.path.join(dst, "bench_line_under_rocprof.json")
    if os.path.exists(kt) and os.path.exists(bl):
        line = json.loads(open(bl).read())
        steps = line["steps"]
        after = (line.get("step_structure") or {}).get("launches_after_timed_region") or {}      # launches of the same kernels behind the timed region
        per = {}
        for row in csv.DictReader(open(kt, newline="")):
            per.setdefault(row["Kernel_Name"], []).append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
        timed = {}
        for k, v in per.items():
            if "lk_track" in k or "pyr_level_kernel<true, true, 0>" in k:
                v.sort()
                skip = int(after.get("lk" if "lk_track" in k else "pyr_l0", 0))
                last = [d for _, d in (v[-(steps + skip):-skip] if skip else v[-steps:])]
                timed[k] = {"launches_in_run": len(v), "timed_launches": len(last), "mean_ms_timed": sum(last) / len(last) * 1e-6,
                            "mean_ms_all": sum(d for _, d in v) / len(v) * 1e-6}
        json.dump({"note": "rocprofv3 --kernel-trace of the bench run whose line is bench_line_under_rocprof.json: mean duration of the last "
                           "`steps` launches (the timed region) and of all launches (what kernel_stats.csv averages)", "kernels": timed},
                  open(os.path.join(dst, "kernel_trace_timed.json"), "w"), indent=1)

    summary, traffic = {}, {}
    for counter, sub, stem in (("FETCH_SIZE", "pmc_fetch", "fetch"), ("WRITE_SIZE", "pmc_write", "write")):
        p = os.path.join(src, sub, stem + "_counter_collection.csv")
        if not os.path.exists(p):
            continue
        summary[counter] = {}
        for k, v in per_kernel(p, counter).items():
            summary[counter][k] = {"dispatches": len(v), "median_KB": statistics.median(v), "max_KB": max(v)}
            t = traffic.setdefault(k, {})
            kb = statistics.median(v)
            if counter == "FETCH_SIZE":
                t["fetch_bytes_raw"] = kb * 1024.0
                t["fetch_bytes"] = kb * 1024.0 * 2.0       # gfx950 correction
            else:
                t["write_bytes"] = kb * 1024.0
    for k, t in traffic.items():
        if "fetch_bytes" in t and "write_bytes" in t:
            t["hbm_bytes"] = t["fetch_bytes"] + t["write_bytes"]
    json.dump(summary, open(os.path.join(dst, "pmc_fetch_write_summary.json"), "w"), indent=1)
    json.dump({"note": "bytes per launch, median over dispatches; FETCH_SIZE doubled (gfx950), WRITE_SIZE as reported; "
                       "bench.py --steps 2 --warmup 1 --no-cpu --no-warp (256 frames / 256 pairs x 2000 points per launch); "
                       "pyr_pass_hbm_bytes = the launches of ONE pyramid + gradient pass of the timed step summed (every launch shape of the "
                       "pyramid kernels that the steps repeat; the one-off full-batch build is left out)",
               "batch": 256, "source_hash": source_hash(TRAFFIC_HASH_FILES), "kernels": traffic,
               "pyr_pass_hbm_bytes": pyr_pass_bytes(os.path.join(src, "pmc_fetch", "fetch_counter_collection.csv"),
                                                    os.path.join(src, "pmc_write", "write_counter_collection.csv"))},
              open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
    # configs[4]'s per-GPU share (bench.py --workload hd: 64 pairs of 1920 x 1080 per launch)
    for name, dstname in (("prof_hd", "kernel_stats_hd.csv"), ("prof_all", "kernel_stats_all_legs.csv")):
        ks2 = os.path.join(src, name, "bench_kernel_stats.csv")
        if os.path.exists(ks2):
            shutil.copy(ks2, os.path.join(dst, dstname))
    hd = {}
    for counter, sub, stem in (("FETCH_SIZE", "pmc_hd_fetch", "fetch"), ("WRITE_SIZE", "pmc_hd_write", "write")):
        p = os.path.join(src, sub, stem + "_counter_collection.csv")
        if not os.path.exists(p):
            continue
        for k, v in per_kernel(p, counter, largest_grid_only=True).items():
            t = hd.setdefault(k, {})
            kb = statistics.median(v)
            if counter == "FETCH_SIZE":
                t["fetch_bytes_raw"] = kb * 1024.0; t["fetch_bytes"] = kb * 1024.0 * 2.0
            else:
                t["write_bytes"] = kb * 1024.0
    for k, t in hd.items():
        if "fetch_bytes" in t and "write_bytes" in t:
            t["hbm_bytes"] = t["fetch_bytes"] + t["write_bytes"]
    if hd:
        json.dump({"note": "bytes per launch of bench.py --workload hd (64 pairs of 1920 x 1080, 5000 points, 5 levels, 21 x 21), the largest "
                           "dispatch of each kernel; FETCH_SIZE doubled (gfx950)", "batch": 64, "source_hash": source_hash(TRAFFIC_HASH_FILES),
                   "kernels": hd,
                   "pyr_pass_hbm_bytes": pyr_pass_bytes(os.path.join(src, "pmc_hd_fetch", "fetch_counter_collection.csv"),
                                                        os.path.join(src, "pmc_hd_write", "write_counter_collection.csv"))},
                  open(os.path.join(dst, "pmc_traffic_hd.json"), "w"), indent=1)
    # warp / paste legs (tools/profile_warp.sh): same treatment, separate files
    ks = os.path.join(src, "prof_warp", "warp_kernel_stats.csv")
    if os.path.exists(ks):
        shutil.copy(ks, os.path.join(dst, "kernel_stats_warp.csv"))
    wt = {}
    for counter, sub, stem in (("FETCH_SIZE", "pmc_warp_fetch", "fetch"), ("WRITE_SIZE", "pmc_warp_write", "write")):
        p = os.path.join(src, sub, stem + "_counter_collection.csv")
        if not os.path.exists(p):
            continue
        shapes = {}                                          # (kernel, grid threads) -> values: a kernel runs at several launch shapes
        for row in csv.DictReader(open(p, newline="")):
            if row["Counter_Name"] == counter:
                shapes.setdefault((row["Kernel_Name"], int(row["Grid_Size"])), []).append(float(row["Counter_Value"]))
        for (k, g), v in shapes.items():
            t = wt.setdefault(k, {"shapes": {}})["shapes"].setdefault(str(g), {"dispatches": len(v)})
            kb = statistics.median(v)
            if counter == "FETCH_SIZE":
                t["fetch_bytes_raw"] = kb * 1024.0
                t["fetch_bytes"] = kb * 1024.0 * 2.0
            else:
                t["write_bytes"] = kb * 1024.0
    for k, e in wt.items():
        for g, t in e["shapes"].items():
            if "fetch_bytes" in t and "write_bytes" in t:
                t["hbm_bytes"] = t["fetch_bytes"] + t["write_bytes"]
        big = max(e["shapes"], key=int)                      # top-level fields: the largest launch shape of the kernel
        e.update({kk: vv for kk, vv in e["shapes"][big].items() if kk != "dispatches"}); e["grid_threads"] = int(big)
    if wt:
        json.dump({"note": "bytes per launch of the kernels of tools/warp_quick.py, per launch shape (`shapes`, keyed by grid threads) -- a kernel "
                           "runs at several: the configs[2] tiles, the KITTI-sized frames -- and, at top level, of the largest one; "
                           "FETCH_SIZE doubled (gfx950)", "source_hash": source_hash(TRAFFIC_HASH_FILES),
                   "kernels": wt}, open(os.path.join(dst, "pmc_traffic_warp.json"), "w"), indent=1)

    # pipe utilisation of the tracker kernels (tools/pmc_lk.sh: passes a, b, c of the bench incl. the other-window legs)
    acc = {}
    for f in glob.glob(os.path.join(src, "pmc_lk_[abc]", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f, newline="")):
            if "lk_track" not in row["Kernel_Name"]:
                continue
            c = acc.setdefault((row["Kernel_Name"], row["Grid_Size"]), {})
            c.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
            if row["Counter_Name"] == "GRBM_GUI_ACTIVE" and row.get("End_Timestamp"):       # the dispatch's own duration, for its clock
                c.setdefault("_clock_ghz", []).append(float(row["Counter_Value"]) / 8.0 / max(1.0, float(row["End_Timestamp"]) - float(row["Start_Timestamp"])))
    kernels = {}
    for (name, grid), c in acc.items():
        m = {k: statistics.median(v) for k, v in c.items()}
        if not m.get("SQ_WAVES"):
            continue
        mm = re.search(r"(lk_track\w*<[^>]*>)", name)
        short = mm.group(1) if mm else name
        w = m["SQ_WAVES"]
        cyc = m["GRBM_GUI_ACTIVE"] / 8.0                      # the counter is summed over the 8 XCDs
        e = {"grid_size": int(grid), "waves": w, "points_per_wave": int(re.search(r"<\d+, (\d+)", short).group(1)) if "lk_trackg" in short else 1,
             "valu_per_wave": m["SQ_INSTS_VALU"] / w, "salu_per_wave": m["SQ_INSTS_SALU"] / w,
             "elapsed_clocks": cyc,
             "active_inst_valu_per_cu_clock": m["SQ_ACTIVE_INST_VALU"] / m["SQ_BUSY_CU_CYCLES"] if m.get("SQ_BUSY_CU_CYCLES") else None,
             "salu_per_cu_clock": m["SQ_INSTS_SALU"] / 256.0 / cyc,
             "wait_inst_any_frac": m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"] if m.get("SQ_WAVE_CYCLES") else None,
             "wait_any_frac": m.get("SQ_WAIT_ANY", 0) / m["SQ_WAVE_CYCLES"] if m.get("SQ_WAVE_CYCLES") else None,
             "lds_per_wave": m.get("SQ_INSTS_LDS", 0) / w, "vmem_rd_per_wave": m.get("SQ_INSTS_VMEM_RD", 0) / w,
             "branch_per_wave": m.get("SQ_INSTS_BRANCH", 0) / w, "smem_per_wave": m.get("SQ_INSTS_SMEM", 0) / w,
             "clock_ghz": m.get("_clock_ghz")}
        # keep the bench's own launch shape when a kernel ran at several: the shape with the most dispatches (the timed region and
        # its pre-roll), the larger one on a tie
        e["dispatches"] = len(c.get("SQ_WAVES", []))
        if short not in kernels or (e["dispatches"], e["waves"]) > (kernels[short]["dispatches"], kernels[short]["waves"]):
            kernels[short] = e
    if kernels:
        for mixf in glob.glob(os.path.join(dst, "valu_mix*.json")):
            mix = json.load(open(mixf))
            for short, e in kernels.items():
                if short.replace(" ", "") in mix.get("kernel_demangled", "").replace(" ", ""):
                    clk = mix["avg_issue_clocks_per_valu"]
                    e["avg_issue_clocks_per_valu"] = clk
                    e["valu_issue_busy"] = e["valu_per_wave"] * e["waves"] * clk / (1024.0 * e["elapsed_clocks"])
                    e["salu_issue_busy"] = e["salu_per_cu_clock"]           # one scalar issue per CU clock
        json.dump({"note": "tracker kernels at the bench's launch shapes, rocprofv3 --pmc (tools/pmc_lk.sh), medians over dispatches; "
                           "active_inst_valu_per_cu_clock calibrates to 0.97 for a saturated 4-clock stream and 1.94 for a saturated "
                           "2-clock stream (tools/pmc_ubench.sh, profiles/r02_v1/pmc_ubench.txt); valu_issue_busy = "
                           "VALU instructions x average issue clocks of the kernel's mix (valu_mix.json) / (1024 SIMDs x elapsed clocks)",
                   "source_hash": source_hash(LK_HASH_FILES), "kernels": kernels}, open(os.path.join(dst, "pmc_lk.json"), "w"), indent=1)
    for name in ("bench_default_run.json", "bench_hd_run.json", "pmc_lk_summary.txt", "pmc_warpq_summary.txt", "ubench_valu_rate.json",
                 "pmc_ubench.txt", "bench_gpus2_gloo_rehearsal.json"):
        p = os.path.join(src, name)
        if os.path.exists(p) and os.path.getsize(p):
            shutil.copy(p, os.path.join(dst, name))
    print("wrote", dst, sorted(os.listdir(dst)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
