#!/usr/bin/env python3
"""Regenerates profiles/<round>/valu_mix*.json (tools/lk_valu_mix.py) for the three tracker kernels the bench runs, with the weights the
committed summaries hold: measured VALU per wave from pmc_lk.json, iteration-levels a wave executes (sum over levels of its lock-step group's slowest point) from the
bench line (headline leg, configs[0] leg, the real-frame leg for the 30x30 kernel, whose counters come from that launch).
usage: python tools/refresh_mix.py profiles/r03_v1     (after tools/summarize_profiles.py; summarize once more afterwards)"""
import json, os, subprocess, sys

prof = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
line = json.loads(open(os.path.join(prof, "bench_default_run.json")).read().strip().splitlines()[-1])
pmc = json.load(open(os.path.join(prof, "pmc_lk.json")))["kernels"]


def measured(tag):
    return [v["valu_per_wave"] for k, v in pmc.items() if tag in k][0]


jobs = [("lk_trackg_kernelILi15ELi8ELi2E", "valu_mix.json", line["lk_kernel"]["kernel"].split("lk_trackg_kernel")[1], 4,
         line["lk_kernel"]["iterations"].get("executed_iteration_levels_per_wave") or line["lk_kernel"]["iterations"]["mean_iteration_levels_of_slowest_in_group"]),
        ("lk_trackg_kernelILi21ELi4ELi4E", "valu_mix_configs0_21x21.json", "<21, 4, 4>", line["lk_other_windows"]["configs0_21x21_3_levels"]["levels"],
         line["lk_other_windows"]["configs0_21x21_3_levels"].get("executed_iteration_levels_per_wave") or
         line["lk_other_windows"]["configs0_21x21_3_levels"]["mean_iteration_levels_of_slowest_in_group"]),
        ("lk_trackg_kernelILi30ELi1ELi2E", "valu_mix_ref_literals_30x30.json", "<30, 1, 2>", line["lk_real_frames"]["reference_literals_30x30"]["levels"],
         line["lk_real_frames"]["reference_literals_30x30"].get("executed_iteration_levels_per_wave") or
         line["lk_real_frames"]["reference_literals_30x30"]["mean_iteration_levels_of_slowest_in_group"])]
for mangled, fname, tag, levels, iters in jobs:
    out = subprocess.check_output([sys.executable, os.path.join(root, "tools", "lk_valu_mix.py"), mangled, str(levels), str(iters), str(measured(tag))])
    open(os.path.join(prof, fname), "wb").write(out)
    j = json.loads(out)
    print("%-36s levels %s, iteration-levels %.3f, predicted / measured VALU per wave %.4f" % (fname, levels, iters, j["predicted_over_measured"]))
