"""Rewrites the generated rows of DESIGN.md section 6 (the targets table and the cpu_baseline / hd_share figures) from the committed bench
line profiles/<round>/bench_default_run.json, so that the document quotes the run that is in the tree.
usage: python tools/design_numbers.py [profiles/r04_v1]"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r04_v1")
d = json.loads(open(os.path.join(prof, "bench_default_run.json")).read().strip().splitlines()[-1])
c, h, w, p = d["cpu_baseline"], d["hd_share"], d["warp"], d["roofline_pyr_grad_pass"]
M = lambda v: "%.0f" % (v / 1e6)
f = os.path.join(ROOT, "DESIGN.md")
s = open(f).read()


def between(s, a, b, new):
    i, j = s.index(a), s.index(b)
    return s[:i] + new + s[j:]


s = re.sub(r"frames resident: \d+ M keypoints/s in the committed run \([0-9.]+ ms per step", "frames resident: %s M keypoints/s in the committed run (%.2f ms per step" % (M(d["value"]), d["ms_per_step"]), s)
s = between(s, "| >= 50x the CPU `calcOpticalFlowPyrLK` throughput at 1 GPU |", "| >= 50x the CPU warp throughput |",
            "| >= 50x the CPU `calcOpticalFlowPyrLK` throughput at 1 GPU | **%s M keypoints/s** in the committed run (462 - 513 M box to box; %s M with two sequences per GPU on two "
            "streams) against **%.2f M** for the CPU port organised like an optimised library (\"port-opt\", §6 `cpu_baseline`: separable pyrDown, padded levels, the SSE2 window "
            "loops of OpenCV's x86-64 build) on the box's %d host cores = **%.0fx** (1.6 - 2.2 M and 240 - 315x over the round's boxes); %.2f M on one thread; the readable port "
            "of round 3 reaches %.2f M on the same cores (%.0fx).  OpenCV itself is not in the image |\n"
            % (M(d["value"]), M(d["two_streams"]["keypoints_per_sec_this_gpu"]), c["value"] / 1e6, c["cores"], d["value"] / c["value"], c["single_thread_value"] / 1e6,
               c["readable_port"]["value"] / 1e6, d["value"] / c["readable_port"]["value"]))
s = between(s, "| >= 50x the CPU warp throughput |", "| >= 40 % of the HBM roofline for the pyramid + gradient pass |",
            "| >= 50x the CPU warp throughput | 8UC3 backward warp of 1241x376 frames %.3f ms per 256 frames (%.0f Mpix/s kernel, strip kernel §4.3.1) and %.3f ms per eight "
            "4000x3000 tiles against %.0f Mpix/s for the port on %d cores |\n"
            % (w["kitti_frames_8uc3"]["kernel_ms"], 256 * 1241 * 376 / 1e6 / (w["kitti_frames_8uc3"]["kernel_ms"] * 1e-3), w["8uc3"]["kernel_ms"], c.get("warp_8uc3_mpix_per_s", 0), c["cores"]))
s = between(s, "| >= 40 % of the HBM roofline for the pyramid + gradient pass |", "| tracked (u, v) within 1e-3 px, warped pixels within 1 ULP |",
            "| >= 40 %% of the HBM roofline for the pyramid + gradient pass | level-0 fused kernel %.0f %% of 8 TB/s by §8(d)'s byte count (%.0f %% by the bytes the counters saw move, "
            "`frac_moved`; a plain 1 GiB device-to-device copy reaches %.1f TB/s on the same box), whole four-level pass incl. the borders %.0f %% (%.0f %% by the %.0f MB the "
            "counters saw for %.0f MB algorithmic); configs[4]'s share (`hd_share`): level-0 kernel %.0f %% |\n"
            % (100 * d["roofline"]["frac"], 100 * (d["roofline"]["frac_moved"] or 0), d["hbm_copy_measured_GBps"] / 1e3, 100 * p["frac"], 100 * (p.get("frac_moved") or 0),
               (p.get("traffic") or 0) / 1e6, p["algorithmic_bytes_per_pass"] / 1e6, 100 * h["roofline"]["frac"]))
s = re.sub(r"in its summation order\) -- [0-9.]+ M keypoints/s on \d+ cores, [0-9.]+ M on one thread --, and\n  `readable_port` = the checker's readable form \(per-tap index reflection, 25-tap pyrDown, scalar order\): [0-9.]+ M / [0-9.]+ M\.",
           "in its summation order) -- %.2f M keypoints/s on %d cores, %.2f M on one thread --, and\n  `readable_port` = the checker's readable form (per-tap index reflection, 25-tap pyrDown, scalar order): %.2f M / %.3f M."
           % (c["value"] / 1e6, c["cores"], c["single_thread_value"] / 1e6, c["readable_port"]["value"] / 1e6, c["readable_port"]["single_thread_value"] / 1e6), s)
s = re.sub(r"timed steps\): [0-9.]+ ms per step = \d+ M keypoints/s \(1.80 ms / 178 M before",
           "timed steps): %.2f ms per step = %s M keypoints/s (1.80 ms / 178 M before" % (h["ms_per_step"], M(h["keypoints_per_sec"])), s)
open(f, "w").write(s)
print("DESIGN.md: %s M keypoints/s, %.3f ms, roofline %.2f, cpu %.2f M (%dx), hd %.2f ms" % (M(d["value"]), d["ms_per_step"], d["roofline"]["frac"], c["value"] / 1e6, d["value"] / c["value"], h["ms_per_step"]))
