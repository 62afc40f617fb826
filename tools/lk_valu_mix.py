#!/usr/bin/env python3
"""VALU issue cost of the tracker's instruction mix.

MI355X issues a wave64 VALU instruction in 2, 4 or 8 shader clocks depending on the opcode (measured: tools/ubench/valu_rate.hip ->
profiles/r02_lk/ubench_valu_rate*.json): plain VOP1/VOP2 integer / f32 add, sub, mul, logic, right shifts and moves whose operands
are VGPRs or inline constants take 2; everything VOP3, every dot / perm / convert / compare / DPP / 64-bit / packed instruction and
any instruction with an SGPR or literal operand takes 4; transcendentals take 8.  A kernel's VALU-issue roofline therefore depends
on its own mix.  This tool compiles lk_kernels.hip for gfx950, walks the ISA of one kernel, weights every basic block by its loop
nest (level loop x iteration loop; deeper loops are the rare re-staging paths) and prints the weighted instruction count per cost
class and the average issue cost.  bench.py combines the average with the PMC instruction count of the same kernel
(profiles/*/pmc_lk.json) into lk_kernel.valu_issue.

usage: tools/lk_valu_mix.py [kernel-substring] [levels] [iteration-levels per wave] [measured VALU per wave] > profiles/<round>/valu_mix.json
"""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "3dr_amd", "csrc", "lk_kernels.hip")

FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_ashrrev_i32",
        "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_mov_b32"}
TRANS = {"v_sqrt_f32", "v_rcp_f32", "v_rsq_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32"}
NOT_VALU_ISSUE = {"v_readlane_b32", "v_readfirstlane_b32", "v_writelane_b32"}      # counted by SQ_INSTS_VALU, cost 4 like the rest


def source_hash():
    h = hashlib.sha256()
    for f in ("lk_kernels.hip", "dr3hip_internal.h"):
        h.update(open(os.path.join(ROOT, "3dr_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def compile_isa():
    """the ISA of both translation units lk_kernels.hip is built into (3dr_amd/csrc/Makefile): LKG_TU = 0 with the default flags, and
    LKG_TU = 1 -- the eight-points-per-wave forms -- under the max-ILP machine scheduler"""
    out_all = ""
    with tempfile.TemporaryDirectory() as d:
        for tu, extra in ((0, []), (1, ["-DLKG_TU=1", "-mllvm", "-amdgpu-sched-strategy=max-ilp"])):
            out = os.path.join(d, "lk%d.s" % tu)
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                                   "-fno-fast-math", "-Wno-unused-function", "-I" + os.path.join(ROOT, "include")] + extra +
                                  ["-S", "--cuda-device-only", "-o", out, SRC], stderr=subprocess.DEVNULL)
            out_all += open(out).read()
    return out_all


def cost_of(line):
    m = re.match(r"\s*(v_[a-z0-9_]+)\s*(.*)", line)
    if not m:
        return None
    op, rest = m.group(1), m.group(2)
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if base in TRANS:
        return 8, "trans"
    enc64 = op.endswith("_e64") or op.endswith("_dpp") or op.endswith("_sdwa")
    if base in FAST and not enc64:
        ops = [o.strip() for o in rest.split(",")]
        srcs = ops[1:]
        scalar = any(re.match(r"^(s\d+|s\[|vcc|exec|m0|0x[0-9a-f]+$|-?\d{3,}$)", o) for o in srcs)      # SGPR or literal operand
        big_int = any(re.match(r"^-?\d+$", o) and not (-16 <= int(o) <= 64) for o in srcs)
        if not scalar and not big_int:
            return 2, "fast"
    return 4, "std"


def main():
    want = sys.argv[1] if len(sys.argv) > 1 else "lk_trackg_kernelILi15ELi8ELi2E"
    levels = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
    iter_levels = float(sys.argv[3]) if len(sys.argv) > 3 else 11.3
    measured = float(sys.argv[4]) if len(sys.argv) > 4 else None          # SQ_INSTS_VALU / SQ_WAVES of the same launch shape (pmc_lk.json)
    isa = compile_isa()
    m = re.search(r"^(_Z\S*%s\S*):[^\n]*\n(.*?)s_endpgm" % re.escape(want), isa, re.S | re.M)
    if not m:
        sys.exit("kernel %s not found" % want)
    body = m.group(2).split("\n")
    # loop depth of every block from the compiler's annotations.  Blocks of the level loop (depth 1) BEHIND the last block of the
    # iteration loop (depth 2) are the level-0 residual and the per-level bookkeeping after the iterations: the residual runs once
    # per wave, not once per level (it is the bulk of that tail), so the whole tail is weighted 1.
    depth, header = 0, {}
    w_of_depth = {0: 1.0, 1: levels, 2: iter_levels}
    last_d2 = max((i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l) and
                   re.search(r"Depth=2", " ".join(x for x in body[i:i + 8] if x.startswith(".LBB") or x.strip().startswith(";")))), default=-1)
    # the end of that last depth-2 block: the next label
    tail_from = next((i for i in range(last_d2 + 1, len(body)) if re.match(r"^\.LBB\d+_\d+:", body[i])), len(body)) if last_d2 >= 0 else len(body)
    # The compiler peels the first Gauss-Newton iteration of a level out of the iteration loop: the copy sits in the level loop (depth 1)
    # in front of the loop proper and runs once per level.  It is the only code in front of the first depth-2 block that reads LDS (the
    # template comes from global memory), which is how it is recognised; the loop proper then runs (iteration-levels - levels) times.
    first_d2 = next((i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l) and
                     re.search(r"Depth=2", " ".join(x for x in body[i:i + 8] if x.startswith(".LBB") or x.strip().startswith(";")))), len(body))
    peeled = any(re.match(r"\s*ds_read", l) for l in body[:first_d2])
    if peeled:
        w_of_depth[2] = max(0.0, iter_levels - levels)
    tot = {"fast": 0.0, "std": 0.0, "trans": 0.0}
    static = {"fast": 0, "std": 0, "trans": 0}
    cur_depth = 0
    for i, line in enumerate(body):
        lm = re.match(r"^\.LBB\d+_\d+:", line)
        if lm:
            # the annotation is on this line or the following comment lines
            ann = " ".join(body[i:i + 8])
            ann = ann.split("\n")[0]
            d = 0
            look = " ".join(l for l in body[i:i + 8] if l.startswith(".LBB") or l.strip().startswith(";"))
            md = re.search(r"Depth=(\d+)", look)
            if md:
                d = int(md.group(1))
            cur_depth = d
            continue
        c = cost_of(line)
        if c is None:
            continue
        clk, cls = c
        static[cls] += 1
        tot[cls] += 1.0 if (cur_depth == 1 and i >= tail_from) else w_of_depth.get(cur_depth, 0.0)
    n = sum(tot.values())
    cycles = 2 * tot["fast"] + 4 * tot["std"] + 8 * tot["trans"]
    dem = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
    out = {"kernel": m.group(1), "kernel_demangled": dem, "source_hash": source_hash(),
           "weights": {"levels": levels, "iteration_levels_per_wave": iter_levels, "deeper_loops": 0.0, "level_loop_tail_behind_the_iterations": 1.0,
                       "first_iteration_peeled_into_the_level_loop": bool(peeled), "iteration_loop_proper": w_of_depth[2]},
           "measured_valu_per_wave": measured, "predicted_over_measured": None if not measured else round(n / measured, 4),
           "static_valu": static, "weighted_valu_per_wave": {k: round(v, 1) for k, v in tot.items()},
           "weighted_valu_per_wave_total": round(n, 1), "issue_clocks_per_wave": round(cycles, 1),
           "avg_issue_clocks_per_valu": round(cycles / n, 4),
           "cost_table": "2 clk: VOP1/VOP2 add/sub/mul f32, add/sub/and/or/xor u32, right shifts, v_mov with VGPR / inline operands; 8 clk: transcendentals; "
                         "4 clk: everything else (tools/ubench/valu_rate.hip at 8 waves per SIMD, profiles/r02_lk/ubench_valu_rate*.txt)"}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
