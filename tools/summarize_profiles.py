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
import json
import os
import shutil
import statistics
import sys


def per_kernel(path, counter):
    acc = {}
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            acc.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
    return acc


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
                       "bench.py --steps 2 --warmup 1 --no-cpu --no-warp (256 frames / 256 pairs x 2000 points per launch)",
               "kernels": traffic}, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
    # warp / paste legs (tools/profile_warp.sh): same treatment, separate files
    ks = os.path.join(src, "prof_warp", "warp_kernel_stats.csv")
    if os.path.exists(ks):
        shutil.copy(ks, os.path.join(dst, "kernel_stats_warp.csv"))
    wt = {}
    for counter, sub, stem in (("FETCH_SIZE", "pmc_warp_fetch", "fetch"), ("WRITE_SIZE", "pmc_warp_write", "write")):
        p = os.path.join(src, sub, stem + "_counter_collection.csv")
        if not os.path.exists(p):
            continue
        for k, v in per_kernel(p, counter).items():
            t = wt.setdefault(k, {})
            kb = statistics.median(v)
            if counter == "FETCH_SIZE":
                t["fetch_bytes_raw"] = kb * 1024.0
                t["fetch_bytes"] = kb * 1024.0 * 2.0
            else:
                t["write_bytes"] = kb * 1024.0
    for k, t in wt.items():
        if "fetch_bytes" in t and "write_bytes" in t:
            t["hbm_bytes"] = t["fetch_bytes"] + t["write_bytes"]
    if wt:
        json.dump({"note": "bytes per launch, median over dispatches of tools/warp_quick.py; FETCH_SIZE doubled (gfx950)",
                   "kernels": wt}, open(os.path.join(dst, "pmc_traffic_warp.json"), "w"), indent=1)

    # per-wave instruction counters of the tracker at the bench's launch shape (tools/profile_all.sh, pmc_lkshape pass)
    p = os.path.join(src, "pmc_lkshape", "p_counter_collection.csv")
    if os.path.exists(p):
        med = {}
        for c in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS", "SQ_INSTS_BRANCH"):
            for k, v in per_kernel(p, c).items():
                if "lk_track_kernel<1" in k:
                    big = [x for x in v] if c != "SQ_WAVES" else v
                    med[c] = statistics.median(big)
        if med.get("SQ_WAVES"):
            w = med["SQ_WAVES"]
            json.dump({"note": "lk_track_kernel<1, 15> at the bench launch shape (256 pairs x 2000 points), rocprofv3 --pmc, "
                               "median over the dispatches of bench.py --steps 2 --warmup 1",
                       "waves": w, "valu_per_wave": med["SQ_INSTS_VALU"] / w, "salu_per_wave": med["SQ_INSTS_SALU"] / w,
                       "vmem_rd_per_wave": med["SQ_INSTS_VMEM_RD"] / w, "lds_per_wave": med["SQ_INSTS_LDS"] / w,
                       "branch_per_wave": med["SQ_INSTS_BRANCH"] / w}, open(os.path.join(dst, "pmc_lk.json"), "w"), indent=1)
    for name in ("bench_default_run.json", "bench_hd_run.json", "pmc_lk_summary.txt"):
        p = os.path.join(src, name)
        if os.path.exists(p) and os.path.getsize(p):
            shutil.copy(p, os.path.join(dst, name))
    print("wrote", dst, sorted(os.listdir(dst)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
