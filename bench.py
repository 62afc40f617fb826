#!/usr/bin/env python3
"""bench.py -- throughput of the dr3hip hot path on N MI355X GPUs of one node.

Workload (BASELINE.json configs[1], "Single MI355X: KITTI-00 stream, 2000 corners/frame, 4-level
pyr, 15x15 window, 30 LK iters"): a stream of synthetic 1241x376 gray frames.  One STEP = one batch
of B new frames: the fused pyramid + Scharr-gradient pass for the B frames, then pyramidal LK
tracking of the B consecutive frame pairs, 2000 points each.  Inputs (level-0 frames, points) are
resident in HBM before the timed region starts.  value = tracked keypoints / s over the whole job.

N > 1 (launched by torch.distributed.run, one rank per GPU): every rank streams its own sequence
(configs[3], weak scaling); the only collective before the timed region is one RCCL broadcast of the
camera intrinsics (the reference-pyramid broadcast of SURVEY 8e is exercised once, untimed, afterwards).
The JSON line also carries: the HBM roofline of the dominant HBM-bound kernel (fused pyrDown + Scharr,
level 0) measured with HIP events on the library's stream and its PMC traffic; the tracker against the
VALU issue roofline; the warp legs of configs[2] (8 tiles 4000x3000 as 8UC3 and 32FC4, 256 KITTI-sized
frames -- the "warped Mpix/s on 1241x376 frames" half of the metric --, the fused panorama paste and the
fused stitch compose); the device-resident front end (pyramids + FAST detect + track); the host-buffer
call with PCIe included; a measured device-to-device copy bandwidth; and the CPU baseline (the oracle
port on all host cores and on one thread, bounded sample).  --workload hd runs one GPU's share of
configs[4] (1920x1080, 5000 corners, 5 levels, 21x21) instead.
"""
import argparse
from ctypes import c_float as C_float
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PRE_ROLL_STEPS = 256
                             # untimed steady-state pre-roll before the W warm-up steps (about 0.3 s of device time).  The part keeps
                             # settling for longer than round 2's 32 steps allowed on some boxes: 20 timed steps right after them took
                             # 1.19 ms each where 400 took 1.12 (round 3, same process); the metric is a stream's steady state
HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (a measured copy rate is reported as hbm_copy_measured_GBps)

KITTI_W, KITTI_H = 1241, 376
N_PTS, WIN, MAX_LEVEL, MAX_ITERS, EPS = 2000, 15, 3, 30, 0.01
# tests/test_pipeline.cpp:62-64 -- the only state shared between GPUs (fx, fy, cx, cy, W, H)
INTRINSICS = (718.856, 718.856, 607.1928, 185.2157, float(KITTI_W), float(KITTI_H))


# ----------------------------------------------------------------------------------------------
# distributed plumbing (torch.distributed: nccl == RCCL on ROCm; gloo for the CPU tests)
# ----------------------------------------------------------------------------------------------
class Dist:
    def __init__(self, n_gpus, backend=None):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != int(n_gpus):
            raise ValueError("--gpus %d but WORLD_SIZE is %d: one rank per GPU" % (n_gpus, self.world))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.torch = None
        self.device = None
        self.pg = False
        if self.world > 1 or backend:
            import torch
            import torch.distributed as dist
            self.torch, self.dist = torch, dist
            backend = backend or "nccl"
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            if backend == "nccl":
                torch.cuda.set_device(self.local_rank)
                self.device = torch.device("cuda", self.local_rank)
            else:
                self.device = torch.device("cpu")
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world)
            self.pg = True
            # barriers and the scalar reductions of the timing protocol go over a gloo group on the CPU: a barrier of the RCCL group is an
            # all-reduce kernel plus a host wait (0.5 ms measured at world size 1), which would sit inside every timed region of N > 1
            # runs and nowhere in the N = 1 run they are compared with
            self.cpu_group = None
            if backend == "nccl" and os.environ.get("DR3_BENCH_NCCL_BARRIER") != "1":
                try:
                    self.cpu_group = dist.new_group(backend="gloo")
                except Exception:                                # noqa: BLE001 -- the RCCL group does the same, slower
                    self.cpu_group = None

    def barrier(self):
        if self.pg:
            if getattr(self, "cpu_group", None) is not None:
                self.dist.barrier(group=self.cpu_group)
            else:
                self.dist.barrier()

    def bcast_intrinsics(self, values):
        """rank 0's (fx, fy, cx, cy, W, H) to every rank -- the path's only collective"""
        if not self.pg:
            return tuple(values)
        t = self.torch.tensor(list(values) if self.rank == 0 else [0.0] * len(values),
                              dtype=self.torch.float64, device=self.device)
        self.dist.broadcast(t, src=0)
        return tuple(float(v) for v in t.cpu().tolist())

    def bcast_bytes(self, payload, n):
        """rank 0's `n` bytes to every rank over the torch.distributed group (bootstrap of the RCCL id)"""
        if not self.pg:
            return bytes(payload)
        t = self.torch.tensor(list(payload) if self.rank == 0 else [0] * n, dtype=self.torch.uint8, device=self.device)
        self.dist.broadcast(t, src=0)
        return bytes(t.cpu().tolist())

    def all_gather_bytes(self, payload):
        """every rank's `payload` (equal lengths) as a list indexed by rank"""
        if not self.pg:
            return [bytes(payload)]
        t = self.torch.tensor(list(payload), dtype=self.torch.uint8, device=self.device)
        out = [self.torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [bytes(o.cpu().tolist()) for o in out]

    def _reduce(self, x, op):
        if not self.pg:
            return float(x)
        if getattr(self, "cpu_group", None) is not None:
            t = self.torch.tensor([float(x)], dtype=self.torch.float64)
            self.dist.all_reduce(t, op=op, group=self.cpu_group)
        else:
            t = self.torch.tensor([float(x)], dtype=self.torch.float64, device=self.device)
            self.dist.all_reduce(t, op=op)
        return float(t.item())

    def max_over_ranks(self, x):
        return self._reduce(x, self.dist.ReduceOp.MAX) if self.pg else float(x)

    def sum_over_ranks(self, x):
        return self._reduce(x, self.dist.ReduceOp.SUM) if self.pg else float(x)

    def close(self):
        if self.pg:
            self.dist.destroy_process_group()
            self.pg = False


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher around it: start N fresh child processes, one rank per GPU
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, exactly what torch.distributed.run would set), relay
    rank 0's stdout (the JSON line), return the first non-zero exit code.  This process has not imported torch and has
    made no HIP call, and it never replaces itself: the children are ordinary subprocesses."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = str(s.getsockname()[1])
        s.close()
    cmd = [sys.executable, os.path.abspath(__file__)] + list(argv)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"), MASTER_PORT=port)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL across processes needs it on this host driver
        procs.append(subprocess.Popen(cmd, env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    rc, live = 0, list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
                print("bench.py: rank %d exited with code %d; stopping the other ranks" % (procs.index(p), code), file=sys.stderr)
                for q in live:                                     # exactly the PIDs started above
                    q.terminate()
        time.sleep(0.05)
    return rc


def timed_region(dist, step_fn, sync_fn, steps, warmup):
    """W untimed steps, then exactly K steps bracketed by barrier + device sync on both sides.
    Returns the MAX elapsed seconds over ranks."""
    import gc
    for _ in range(warmup):
        step_fn()
    sync_fn()
    # the interpreter's cyclic collector stays out of the timed loop (a full collection of this process takes 35 - 45 ms, twice the
    # 20 default steps; it was caught in the host-buffer leg, one call in 200).  It is switched off, not run, here: 40 ms of idle
    # device right before the loop cost the first 20 launches 11 % (tools/step_decay.sh) -- the caller collects before its warm-up
    gc.disable()
    try:
        dist.barrier()
        sync_fn()
        t0 = time.perf_counter()
        for _ in range(steps):
            step_fn()
        sync_fn()
        dist.barrier()
        elapsed = time.perf_counter() - t0
    finally:
        gc.enable()
    return dist.max_over_ranks(elapsed)


def aggregate_rate(dist, units_this_rank, elapsed_max):
    """whole-job units / s: sum of every rank's units over the slowest rank's time"""
    return dist.sum_over_ranks(units_this_rank) / elapsed_max


# ----------------------------------------------------------------------------------------------
# algorithmic bytes
# ----------------------------------------------------------------------------------------------
def kernel_source_hash(files):
    """hash of kernel sources, the stamp tools/summarize_profiles.py and tools/lk_valu_mix.py put on what they record: a profile
    of an older kernel must not be reported for the one being timed"""
    import hashlib
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(ROOT, "3dr_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def newest_profile(fname, want_hash=None):
    """newest committed profiles/<round>/<fname> whose source_hash (when the file carries one) matches the sources in the tree"""
    root = os.path.join(ROOT, "profiles")
    try:
        for d in sorted(os.listdir(root), reverse=True):
            f = os.path.join(root, d, fname)
            if os.path.exists(f):
                j = json.load(open(f))
                if want_hash is None or j.get("source_hash") in (None, want_hash):
                    return j, "profiles/%s/%s" % (d, fname)
    except (OSError, ValueError):
        pass
    return None, None


def pmc_traffic(kernel_substr, batch, fname="pmc_traffic.json", grid_threads=None):
    """HBM bytes per launch of one kernel from the newest committed PMC summary (profiles/*/pmc_traffic.json,
    made by tools/profile_bench.sh + tools/summarize_profiles.py from separate rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE passes of this same script; FETCH_SIZE doubled as the microarch guide prescribes for gfx950).
    PMC counters cannot be collected from inside the process being timed, so the figure is the recorded one
    for the same launch shape (default --batch); any other shape, or a summary recorded for other kernel sources, reports None."""
    j, src = newest_profile(fname, kernel_source_hash(("pyr_kernels.hip", "warp_kernels.hip", "dr3hip_internal.h")))
    if j and j.get("batch", 256) != batch:
        return None, None
    if j:
        for name, t in j.get("kernels", {}).items():
            if kernel_substr not in name:
                continue
            if grid_threads is not None:                     # a kernel that runs at several launch shapes: this one, or nothing
                t = t.get("shapes", {}).get(str(int(grid_threads)), {})
            if "hbm_bytes" in t:
                return t["hbm_bytes"], src
    return None, None


def pyr_pass_traffic(batch, fname="pmc_traffic.json"):
    """HBM bytes of one whole pyramid + gradient pass (all its launches) from the same committed PMC summary, or None"""
    j, src = newest_profile(fname, kernel_source_hash(("pyr_kernels.hip", "warp_kernels.hip", "dr3hip_internal.h")))
    if not j or j.get("batch", 256) != batch or not j.get("pyr_pass_hbm_bytes"):
        return None
    return j["pyr_pass_hbm_bytes"]


def lk_valu_issue(lk_ms, waves, kernel):
    """The tracker against the VALU issue roofline of ITS OWN instruction mix.  gfx950 issues a wave64 VALU instruction in 2, 4
    or 8 clocks depending on the opcode (tools/ubench/valu_rate.hip, profiles/r02_lk/ubench_valu_rate*.txt), so the ceiling is
    1024 SIMDs x 2.4 GHz / (average issue clocks of the mix).  VALU instructions per wave: rocprofv3 --pmc SQ_INSTS_VALU /
    SQ_WAVES of this launch shape (profiles/*/pmc_lk.json); average issue clocks: tools/lk_valu_mix.py on the kernel's ISA
    (profiles/*/valu_mix.json).  Both files carry the hash of the kernel source they were made from; a mismatch reports None
    rather than numbers of another kernel."""
    want = kernel_source_hash(("lk_kernels.hip", "dr3hip_internal.h"))
    pmc, psrc = newest_profile("pmc_lk.json", want)
    mix, msrc = newest_profile("valu_mix.json", want)
    if not pmc or not mix or pmc.get("source_hash") != want or mix.get("source_hash") != want:
        return None
    k = next((v for name, v in pmc.get("kernels", {}).items() if kernel in name), None)
    if not k:
        return None
    clk = mix["avg_issue_clocks_per_valu"]
    n_wave_instr = k["valu_per_wave"] * (waves / k["points_per_wave"])
    ach = n_wave_instr / (lk_ms * 1e-3) / 1e9
    peak = 1024 * 2.4 / clk
    ghz = k.get("clock_ghz")
    return {"bound": "valu-issue", "achieved": ach, "peak": peak, "unit": "G wave-instructions/s", "frac": ach / peak,
            "frac_at_profiled_clock": None if not ghz else ach / (1024 * ghz / clk),
            "valu_mix_predicted_over_measured": mix.get("predicted_over_measured"),
            "avg_issue_clocks_per_valu": clk, "valu_per_wave": k["valu_per_wave"], "salu_per_wave": k.get("salu_per_wave"),
            "points_per_wave": k["points_per_wave"],
            "valu_pipe_busy_profiled": k.get("valu_issue_busy"), "scalar_pipe_busy_profiled": k.get("salu_issue_busy"),
            "clock_ghz_profiled": k.get("clock_ghz"), "source": [psrc, msrc],
            "note": "peak = 1024 SIMDs x 2.4 GHz / avg issue clocks of this kernel's VALU mix (2-clock and 4-clock opcodes measured by "
                    "tools/ubench/valu_rate.hip); *_busy_profiled = issue clocks / elapsed clocks inside the rocprofv3 run at its own clock"}


def lk_kernel_of(win_w, win_h):
    """Name of the tracker instantiation launch_lk_track (3dr_amd/csrc/lk_kernels.hip) picks for a window: <WIN, points per wave,
    lane-items per lane>; WIN = 0 is the run-time-window form."""
    n8 = win_w * ((win_h + 7) // 8)
    if win_w <= 16 and win_h <= 16:
        ppw, ni, fixed = 8, 2, 15
    elif win_w == 21 and win_h == 21:
        ppw, ni, fixed = 4, 4, 21
    elif n8 <= 32:
        ppw, ni, fixed = 2, 1, -1
    elif n8 <= 64:
        ppw, ni, fixed = 2, 2, 21
    else:
        ppw, ni, fixed = 1, 2, 30
    return "lk_trackg_kernel<%d, %d, %d>" % (win_w if win_w == win_h == fixed else 0, ppw, ni)


def level_sizes(w, h, levels):
    out = []
    for _ in range(levels):
        out.append((w, h))
        w, h = (w + 1) // 2, (h + 1) // 2
    return out


def pyr_grad_bytes_per_frame(w, h, levels):
    """pyramid: read levels 0..L-1, write levels 1..L (1 B/px); gradient: read S, write 4 S"""
    px = [a * b for a, b in level_sizes(w, h, levels)]
    S = sum(px)
    return sum(px[:-1]) + sum(px[1:]) + S + 4 * S


def pyr_grad_level0_bytes_per_frame(w, h):
    """the level-0 launch alone: pyrDown reads W0H0 and writes W1H1, Scharr reads W0H0, writes 4 W0H0"""
    p0 = w * h
    p1 = ((w + 1) // 2) * ((h + 1) // 2)
    return p0 + p1 + p0 + 4 * p0


def lk_unique_bytes_per_pair(n, levels, win):
    return n * levels * (win + 1) ** 2 * 6 + 21 * n


# ----------------------------------------------------------------------------------------------
def stream_frames(synth, n_slots, w, h, period=9):
    """ping-pong over `period` distinct analytic frames: consecutive slots always differ by one
    motion step (forward or backward), like consecutive frames of a sequence"""
    base, _ = synth.frame_stream(period, w, h)
    seq = list(range(period)) + list(range(period - 2, 0, -1))
    return [base[seq[i % len(seq)]] for i in range(n_slots)], base


def strip_grid_threads(w, h, n_jobs):
    """threads of one warp_tile_8uc3_kernel launch (launch_warp in warp_kernels.hip): 64 x 64 tiles per job, the tile count
    padded to a multiple of 8 x run blocks (run = 4 tiles of a row per XCD turn when a row has >= 32 tiles), 256 threads each"""
    tx, ty = -(-w // 64), -(-h // 64)
    per = 8 * (4 if tx >= 32 else 1)
    return -(-tx * ty // per) * per * n_jobs * 256


def direct_grid_threads(w, h, n_jobs, four_px_lanes):
    """threads of one warp_perspective_kernel launch (launch_warp in warp_kernels.hip): a wave tile is WaveTile<PXT>::LX lanes x LY rows
    (DR3_WT4_LOG = 4: 16 lanes of 4 pixels x 4 rows for 8UC1 / 8UC3; DR3_WT1_LOG = 6: 64 one-pixel lanes x 1 row otherwise), a block
    four of them stacked in y"""
    lx, ly, pxt = (16, 4, 4) if four_px_lanes else (64, 1, 1)
    return -(-w // (lx * pxt)) * -(-h // (4 * ly)) * n_jobs * 256


def cheap_tile(synth, w, h):
    """4000x3000x3 u8 tile: low-res analytic texture upsampled + deterministic fine pattern
    (content does not influence warp timing; the full analytic generator would take minutes)"""
    f = 8
    low = synth.Texture(-(-w // f), -(-h // f), channels=3).render()
    big = np.repeat(np.repeat(low, f, axis=0), f, axis=1)[:h, :w].astype(np.int16)
    yy, xx = np.mgrid[0:h, 0:w]
    big += (((xx * 7 + yy * 13) & 15) - 8)[..., None].astype(np.int16)
    return np.clip(big, 0, 255).astype(np.uint8)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="frame pairs per step and per GPU (default 256; 64 for --workload hd)")
    ap.add_argument("--workload", choices=("kitti", "hd"), default="kitti",
                    help="kitti = BASELINE configs[1] (the metric's configuration, default); hd = the per-GPU share of configs[4]: "
                         "1920x1080 pairs, 5000 corners, 5-level pyramid, 21x21 window (default batch 64)")
    ap.add_argument("--no-warp", action="store_true", help="skip the configs[2] warp leg")
    ap.add_argument("--sharded-paste", default=None, metavar="WxH", help="N > 1: run the canvas-sharded paste leg (one panorama per step across "
                    "the job) on tiles of this size even with --no-warp (default with the warp leg: 4000x3000)")
    ap.add_argument("--no-frontend", action="store_true", help="skip the detect -> track front-end leg")
    ap.add_argument("--no-hd-share", action="store_true", help="skip the bounded configs[4] leg (it launches the 21x21 tracker at a second shape, which a "
                    "profiler pass that attributes counters per kernel symbol must not mix with the configs[0] leg's)")
    ap.add_argument("--no-host-api", action="store_true", help="skip the PCIe-inclusive host-buffer leg (its single-pair launches "
                    "share kernel symbols with the timed region and would skew a profiler's per-kernel averages)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args(argv)
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if env_world is None and args.gpus > 1:
        return launch_ranks(args.gpus, sys.argv[1:] if argv is None else list(argv))
    if env_world is not None and int(env_world) != args.gpus:
        print("bench.py: --gpus %d disagrees with WORLD_SIZE=%s (launch N ranks with --gpus N, or drop the launcher)"
              % (args.gpus, env_world), file=sys.stderr)
        return 2
    global N_PTS, WIN, MAX_LEVEL, INTRINSICS
    grid_cell, period, workload_name = 15, 9, ("configs[1]: KITTI-00-shaped stream 1241x376 u8, 2000 corners/frame, 4-level pyr (maxLevel 3), "
                                               "15x15 window, 30 LK iters (COUNT+EPS 0.01)")
    if args.workload == "hd":
        N_PTS, WIN, MAX_LEVEL = 5000, 21, 4
        INTRINSICS = (1050.0, 1050.0, 959.5, 539.5, 1920.0, 1080.0)
        grid_cell, period = 20, 3
        workload_name = ("configs[4], one GPU's share: batched 1920x1080 u8 pairs, 5000 corners, 5-level pyr (maxLevel 4), 21x21 window, "
                         "30 LK iters (COUNT+EPS 0.01)")
        args.no_frontend = True; args.no_warp = True; args.no_host_api = True
    if args.batch is None:
        args.batch = 64 if args.workload == "hd" else 256

    # The contract is ONE JSON line on stdout.  RCCL and Gloo print banners on file descriptor 1 ("RCCL version : ...", "[Gloo] Rank 0
    # is connected ..."): keep a private handle on the real stdout for the line and point fd 1 at stderr for everything else.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    # torch first: its bundled libamdhip64 and ours share one SONAME, so whichever is loaded first
    # serves both; loading torch's first keeps a single HIP runtime in the process.
    import torch
    # DR3_BENCH_BACKEND=gloo rehearses the N>1 protocol on a box with fewer GPUs than ranks (the ranks then
    # share devices; RCCL refuses two ranks on one GPU).  The driver's runs use the default, nccl == RCCL.
    # DR3_BENCH_FORCE_PG=1 runs the process-group code (RCCL id hand-over, broadcast leg) even at world size 1.
    world = int(env_world or "1")
    multi = world > 1 or os.environ.get("DR3_BENCH_FORCE_PG") == "1"
    backend = os.environ.get("DR3_BENCH_BACKEND", "nccl") if multi else None
    # DR3_BENCH_STANDIN=<module under tests/>: TEST-ONLY rehearsal of the launch path on a box without GPUs.  The module
    # replaces the product package with timing stubs (no compute, no oracle); the JSON line then says "data": "standin"
    # and is not a measurement.  Without it the product library and a HIP device are required (no CPU fallback).
    standin = os.environ.get("DR3_BENCH_STANDIN")
    dr3 = importlib.import_module(standin if standin else "3dr_amd")
    synth = importlib.import_module("3dr_amd.synth")
    n_dev = dr3.lib().dr3hip_device_count()
    if world > 1 and backend != "gloo" and n_dev < world:
        # every rank sees the same count and leaves here, before any rendezvous: nobody is left waiting
        print("bench.py: %d ranks need %d GPUs, dr3hip_device_count() = %d (ranks never share a device over RCCL)"
              % (world, world, n_dev), file=sys.stderr)
        return 4
    dist = Dist(args.gpus, backend=backend)

    fx, fy, cx, cy, Wf, Hf = dist.bcast_intrinsics(INTRINSICS)
    W, H = int(Wf), int(Hf)
    B = args.batch
    ctx = dr3.Context(dist.local_rank % max(n_dev, 1))
    have_gpu = not standin

    def sync():
        ctx.sync()
        if have_gpu:
            torch.cuda.synchronize()

    # ---- resident inputs ------------------------------------------------------------------------
    # A ring of two slot groups (G = B + 1 slots each: the frame carried over from the previous batch + B new frames): while the tracker
    # works on group k, the pyramid levels 1.. of group k + 1 are built under it on the library's auxiliary stream
    # (dr3hip_batch_build_ahead; the level-0 launch -- the graded HBM kernel -- runs alone, before the tracker).  The groups hold the
    # same frames, so every step does the same work: B new frames through the pyramid + gradient pass, B pairs through the tracker.
    frames, base = stream_frames(synth, B + 1, W, H, period=period)
    G = B + 1
    batch = dr3.FrameBatch(ctx, W, H, win=(WIN, WIN), max_level=MAX_LEVEL, capacity=2 * G)
    levels = batch.levels
    for g_ in range(2):
        for s0 in range(0, G, 64):
            batch.upload(g_ * G + s0, np.stack(frames[s0:s0 + 64]))
    pts = synth.grid_points(W, H, grid_cell, n=N_PTS)
    P = np.ascontiguousarray(np.broadcast_to(pts[None], (B, N_PTS, 2)), np.float32)
    d_prev, d_next = ctx.to_device(P), ctx.to_device(P)
    d_st, d_err = ctx.alloc(B * N_PTS), ctx.alloc(B * N_PTS * 4)
    params = dr3.lk_params(win=(WIN, WIN), max_level=MAX_LEVEL, max_iters=MAX_ITERS, eps=EPS, flags=0)
    # slot 0 stands for the frame carried over from the previous batch of the stream.  Build it with a
    # full-size launch so that every launch rocprofv3 averages for this kernel has the same shape.
    batch.build(0, G, True)
    batch.build(G, G, True)
    ring = [0]

    def step():
        g = ring[0] & 1; ring[0] += 1
        batch.build_ahead((1 - g) * G + 1, B, True)              # pyramid + gradient pass, B new frames of the OTHER group: level 0 now ...
        batch.track(B, g * G, d_prev, d_next, N_PTS, d_st, d_err, params)   # ... B pairs x 2000 points of this group, levels 1.. of the other under it

    def serial_step():
        batch.build(1, B, True)
        batch.track(B, 0, d_prev, d_next, N_PTS, d_st, d_err, params)

    # ---- timed region --------------------------------------------------------------------------
    # Pre-roll, independent of --warmup: code objects load at first launch and the part needs some tens of milliseconds
    # of load to settle its clocks (5 timed steps right after 2 warm-up steps measured 2.18 ms per step, 20 steps 2.06).
    import gc
    gc.collect()                                                 # before the pre-roll, so that no collection falls into the timed steps
    for _ in range(PRE_ROLL_STEPS):
        step()
    sync()
    ctx.profile_enable(True)
    for _ in range(args.warmup):
        step()
    sync()
    ctx.profile_reset()
    elapsed = timed_region(dist, step, sync, args.steps, 0)
    kps = aggregate_rate(dist, B * N_PTS * args.steps, elapsed)
    ms_l0, n_l0 = ctx.profile_read(dr3.PROF_PYR_L0)
    ms_pass_ov, n_pass_ov = ctx.profile_read(dr3.PROF_PYR)       # level 0 alone + levels 1.. under the tracker: first launch's start to last launch's end
    ms_lk, n_lk = ctx.profile_read(dr3.PROF_LK)
    # the same step without the overlap, right behind the timed region (untimed extra): the whole pass with every launch running alone
    # (roofline_pyr_grad_pass), and what the overlap is worth
    batch.join(); sync(); ctx.profile_reset()
    t0_ = time.perf_counter()
    for _ in range(max(args.steps, 10)):
        serial_step()
    sync()
    serial_ms_per_step = (time.perf_counter() - t0_) / max(args.steps, 10) * 1e3
    # launches of the headline kernels behind the timed region (for a profiler's trace of this run: tools/summarize_profiles.py picks the
    # timed launches by counting back from the end).  Only known when no other leg launches the same kernel symbols
    after_region = {"lk": max(args.steps, 10) + (1 if have_gpu else 0), "pyr_l0": max(args.steps, 10)} if (args.no_frontend and args.no_host_api) else None
    ms_pass, n_pass = ctx.profile_read(dr3.PROF_PYR)
    ctx.profile_enable(False)

    # ---- result check against the oracle (rank 0, pair 0) -------------------------------------------
    verified, raster, orders = None, None, None
    tracked_frac = float(d_st.download((B, N_PTS), np.uint8).mean())
    if dist.rank == 0 and have_gpu:
        from oracle import oracle as O
        po = O.lk_params(win=(WIN, WIN), max_level=MAX_LEVEL, max_iters=MAX_ITERS, eps=EPS, flags=0)
        o_xy, o_st, o_err, o_L, o_it = O.lk_track(frames[0], frames[1], pts, None, po, order=O.ORDER_WAVE64, want_iters=True)
        g_xy = d_next.download((B, N_PTS, 2), np.float32)[0]
        g_st = d_st.download((B, N_PTS), np.uint8)[0]
        verified = bool(np.array_equal(g_xy, o_xy) and np.array_equal(g_st, o_st))
        # the same pair in OpenCV's scalar summation order: north_star's 1e-3 px bar (oracle.order_agreement says how it is read)
        r_res = O.lk_track(frames[0], frames[1], pts, None, po, order=O.ORDER_RASTER, want_iters=True)
        raster = O.order_agreement((g_xy, g_st, None, o_L, o_it), r_res, po, 1e-3)
        # ... and in the order an x86-64 OpenCV 3.x build executes (LKTrackerInvoker's CV_SSE2 blocks): what the reference's CPU path
        # computes behind src/initialization.cpp:608-613.  The third pair (OpenCV's two builds against each other) says how much of a
        # deviation is the summation order's own
        s_res = O.lk_track(frames[0], frames[1], pts, None, po, order=O.ORDER_SSE2, want_iters=True)
        orders = {"kernel_vs_sse2": O.order_agreement((g_xy, g_st, None, o_L, o_it), s_res, po, 1e-3),
                  "kernel_vs_raster": raster,
                  "raster_vs_sse2": O.order_agreement(r_res, s_res, po, 1e-3),
                  "note": "pair 0 of the timed launches; kernel = the GPU's output (equal to the kernel-order oracle bit for bit), sse2 = the "
                          "summation order of OpenCV 3.x's x86-64 build (oracle ORDER_SSE2), raster = its scalar build; all four BASELINE "
                          "configurations: profiles/r04_v1/order_agreement.json (tools/order_agreement.py)"}

    # ---- how many Gauss-Newton iteration-levels a point takes, and the slowest point of a lock-step group (what a wave executes):
    # one more launch, untimed, that also writes the iteration counts
    lk_iters = None
    if have_gpu:
        d_it = ctx.alloc(B * N_PTS * levels * 4)
        batch.track(B, 0, d_prev, d_next, N_PTS, d_st, d_err, params, d_iters=d_it); sync()
        it = d_it.download((B, N_PTS, levels), np.int32).sum(2)
        ppw = int(lk_kernel_of(WIN, WIN).split(",")[1])
        itl = d_it.download((B, N_PTS, levels), np.int32)[:, :N_PTS // ppw * ppw].reshape(B, -1, ppw, levels)
        lk_iters = {"mean_iteration_levels_per_point": float(it.mean()), "points_per_wave": ppw,
                    "mean_iteration_levels_of_slowest_in_group": float(it[:, :N_PTS // ppw * ppw].reshape(B, -1, ppw).max(2).mean()),
                    "executed_iteration_levels_per_wave": float(itl.max(2).sum(2).mean()),
                    "executed_note": "what a wave executes: sum over levels of the slowest point of its lock-step group on that level"}
        d_it.free()

    # ---- what this box's HBM delivers on a plain device-to-device copy (report beside the nominal 8 TB/s)
    copy_gbps = None
    if dist.rank == 0 and have_gpu:
        try:
            a_t = torch.empty(1 << 30, dtype=torch.uint8, device="cuda:%d" % ctx.device); b_t = torch.empty_like(a_t)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            b_t.copy_(a_t); torch.cuda.synchronize()
            e0.record()
            for _ in range(5):
                b_t.copy_(a_t)
            e1.record(); torch.cuda.synchronize()
            copy_gbps = 5 * 2 * (1 << 30) / (e0.elapsed_time(e1) * 1e-3) / 1e9          # bytes read + bytes written
            del a_t, b_t
        except Exception:                                        # noqa: BLE001 -- informational only
            copy_gbps = None

    # ---- the reference's own call shape, PCIe included: one dr3hip_lk_track per pair with host buffers (2 frames up,
    # pyramids + gradients + track, results down).  Reported beside `value`, never as `value` (inputs are not resident).
    host_api = None
    if dist.rank == 0 and not args.no_host_api:
        n_host = 200
        ctx.calcOpticalFlowPyrLK(frames[0], frames[1], pts, None, params)                    # warm-up (staging buffers)
        t_call = np.empty(n_host)
        import gc
        gc.collect(); gc.disable()                               # the interpreter's collector is not part of the C ABI's time
        try:
            for i in range(n_host):
                t0 = time.perf_counter()
                ctx.calcOpticalFlowPyrLK(frames[i % B], frames[i % B + 1], pts, None, params)
                t_call[i] = time.perf_counter() - t0
        finally:
            gc.enable()
        # the rate is the mean over all calls; the median and the slowest call stand beside it because a single host-side stall of
        # some tens of milliseconds (seen on some boxes, in one call of a run) is otherwise indistinguishable from a slow path
        host_api = {"keypoints_per_sec": n_host * N_PTS / t_call.sum(), "ms_per_pair": t_call.mean() * 1e3,
                    "ms_per_pair_median": float(np.median(t_call)) * 1e3, "ms_slowest_call": float(t_call.max()) * 1e3,
                    "sample": "%d calls of dr3hip_lk_track (host frames 1241x376 + %d points in, positions/status/err out)" % (n_host, N_PTS)}

    # ---- the whole front end on the device: pyramid + gradients, FAST detector, LK from the detected points -----
    frontend = None
    if not args.no_frontend:
        frontend = frontend_leg(dist, ctx, dr3, batch, B, sync, max(3, min(args.steps, 10)))

    # ---- the tracker's other instantiations: the reference's own literals and configs[0] ---------------------
    lk_other = None
    if args.workload == "kitti" and not args.no_frontend:
        lk_other = lk_other_windows_leg(dist, ctx, dr3, synth, frames, sync, have_gpu)

    # ---- the tracker where flows differ from point to point: the reference's own KITTI pair at the headline launch shape ----------
    lk_real = None
    if args.workload == "kitti" and not args.no_frontend:
        lk_real = lk_real_pair_leg(dist, ctx, dr3, sync, have_gpu)

    # ---- SURVEY 8(f) row 5 (extension): sparse image alignment of every consecutive pair of the stream ---------
    align = None
    if args.workload == "kitti" and not args.no_frontend:
        align = align_leg(dist, ctx, dr3, synth, batch, frames, B, (fx, fy, cx, cy), sync, have_gpu, max(3, min(args.steps, 10)))

    # ---- Normalize + the CheckFundamental vote on the tracker's device-resident output -----------------------
    post_lk = None
    if args.workload == "kitti" and not args.no_frontend and have_gpu:
        post_lk = post_lk_leg(dist, ctx, dr3, d_prev, d_next, d_st, pts, N_PTS, sync)

    # ---- two sequences per GPU on two streams: the HBM-bound pyramid pass of one overlaps the VALU-bound tracker of the other
    two_streams = None
    if args.workload == "kitti" and not args.no_frontend and have_gpu:
        two_streams = two_streams_leg(dist, ctx, dr3, frames, P, B, N_PTS, (W, H), params, serial_step, (d_next, d_st), sync, max(4, min(args.steps, 20)))

    # ---- the same step fed from the host every step (PCIe inclusive), pipelined ------------------------------------------------
    stream = None
    if args.workload == "kitti" and not args.no_host_api and have_gpu:
        rx = d_next.download((B, N_PTS, 2), np.float32); rs = d_st.download((B, N_PTS), np.uint8)
        stream = stream_leg(dist, ctx, dr3, frames, P, B, N_PTS, (W, H), params, sync, max(6, min(args.steps, 20)), rx, rs, have_gpu)

    # ---- configs[4], one GPU's share, bounded (64 pairs, 5 timed steps): the driver's default line carries it ---------------------
    hd_share = None
    if args.workload == "kitti" and not args.no_frontend and not args.no_hd_share and have_gpu:
        hd_share = hd_share_leg(dist, ctx, dr3, synth, sync)

    # ---- configs[2]: panorama backward warp, 8 tiles 4000x3000 -----------------------------------------
    warp = None
    if not args.no_warp:
        warp = warp_leg(dist, ctx, dr3, synth, sync, steps=max(3, min(args.steps, 10)))

    # ---- SURVEY 8(e), warp side: the canvas of ONE panorama sharded over the ranks by rows (N > 1) -----------------------------
    paste_sharded = None
    if dist.world > 1 and (args.sharded_paste or not args.no_warp):
        tw_, th_ = (int(v) for v in (args.sharded_paste or "4000x3000").lower().split("x"))
        paste_sharded = paste_sharded_leg(dist, ctx, dr3, synth, sync, cheap_tile(synth, tw_, th_), max(3, min(args.steps, 10)),
                                          feather=max(8, min(300, tw_ // 12)))

    # ---- RCCL broadcast of a reference frame's pyramid through the library's own communicator.
    # Untimed, after every measurement, N > 1 over real GPUs only (RCCL refuses ranks that share a device), and under
    # a watchdog: a communicator that cannot come up must cost the run this one sub-object, not its JSON line.
    ref_bcast, leg_hung = None, False
    if dist.pg and backend == "nccl" and os.environ.get("DR3_BENCH_NO_REF_BCAST") != "1":
        import threading
        box = {}
        th = threading.Thread(target=lambda: box.update(r=ref_bcast_leg(dist, ctx, dr3, batch, frames, levels, sync)), daemon=True)
        th.start()
        th.join(float(os.environ.get("DR3_BENCH_REF_BCAST_TIMEOUT", "120")))
        leg_hung = th.is_alive()
        ref_bcast = {"error": "timed out (communicator bootstrap or broadcast did not complete)"} if leg_hung else box.get("r")

    # ---- CPU baseline (rank 0 only, bounded sample) -----------------------------------------------------
    cpu = None
    if dist.rank == 0 and not args.no_cpu and dist.world == 1:
        cpu = cpu_baseline(frames, pts, args.cpu_seconds, warp is not None, synth)

    if dist.rank == 0:
        b_l0 = pyr_grad_level0_bytes_per_frame(W, H) * B
        b_pass = pyr_grad_bytes_per_frame(W, H, levels) * B
        l0_ms = ms_l0 / max(n_l0, 1)
        pass_ms = ms_pass / max(n_pass, 1)
        lk_ms = ms_lk / max(n_lk, 1)
        ach_l0 = b_l0 / (l0_ms * 1e-3) / 1e9
        ach_pass = b_pass / (pass_ms * 1e-3) / 1e9
        traffic_l0, traffic_src = pmc_traffic("pyr_level_kernel<true, true, 0>", B, "pmc_traffic.json" if args.workload == "kitti" else "pmc_traffic_hd.json")
        traffic_pass = pyr_pass_traffic(B, "pmc_traffic.json" if args.workload == "kitti" else "pmc_traffic_hd.json")
        lk_kernel_name = lk_kernel_of(WIN, WIN)
        out = {
            "metric": "tracked_keypoints_per_sec",
            "value": kps,
            "unit": "keypoints/s",
            "n_gpus": dist.world,
            "steps": args.steps,
            "warmup": args.warmup,
            "pre_roll_steps": PRE_ROLL_STEPS,
            "steady_state_note": "value is a stream's steady state: %d untimed pre-roll steps precede the %d warm-up steps" % (PRE_ROLL_STEPS, args.warmup),
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic" if have_gpu else "standin (launch-path rehearsal without a GPU: NOT a measurement)",
            "config": {"workload": workload_name,
                       "frame_pairs_per_step_per_gpu": B, "points_per_pair": N_PTS, "new_frames_per_step_per_gpu": B,
                       "sequences": dist.world, "parallelism": "one sequence per GPU, no data-path collective",
                       "intrinsics_bcast": [fx, fy, cx, cy, W, H]},
            "frames_per_sec": kps / N_PTS,
            "warped_mpix_per_sec_1241x376": (warp or {}).get("kitti_frames_8uc3", {}).get("mpix_per_s"),
            "tracked_fraction": tracked_frac,
            "verified_vs_oracle": verified,
            "raster_within_1e-3_frac": None if raster is None else raster["within_tol_frac"],
            "raster_max_dev_px": None if raster is None else raster["max_dev_px"],
            "raster_order_agreement": raster,
            "sse2_within_1e-3_frac": None if orders is None else orders["kernel_vs_sse2"]["within_tol_frac"],
            "sse2_converged_max_dev_px": None if orders is None else orders["kernel_vs_sse2"]["converged_max_dev_px"],
            "summation_order_agreement": orders,
            "roofline": {"kernel": "pyr_level_kernel<DOWN,DERIV,0> (fused pyrDown+Scharr, level 0)",
                         "bound": "hbm", "achieved": ach_l0, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": ach_l0 / HBM_PEAK_GBPS, "traffic": traffic_l0,
                         "frac_moved": None if not traffic_l0 else traffic_l0 / (l0_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                         "frac_moved_note": "bytes the counters saw per launch / launch time / peak: SURVEY 8(d) counts the level-0 read twice "
                                            "(once for pyrDown, once for Scharr); the fused kernel fetches it once",
                         "traffic_unit": "bytes per launch (rocprofv3 PMC, FETCH_SIZE x2 + WRITE_SIZE)",
                         "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": b_l0, "avg_launch_ms": l0_ms, "launches": n_l0},
            "step_structure": {"ring": "two groups of %d slots; step = dr3hip_batch_build_ahead(other group: level 0 on the stream, levels 1.. on the auxiliary stream) "
                                       "+ dr3hip_batch_track(this group)" % G,
                               "ms_per_step_without_overlap": serial_ms_per_step,
                               "launches_after_timed_region": after_region,
                               "pass_span_ms_in_timed_region": ms_pass_ov / max(n_pass_ov, 1),
                               "note": "the level-0 launch (roofline) and the tracker run alone; the three launches of levels 1.. (a quarter of the pass's bytes) run "
                                       "under the tracker.  roofline_pyr_grad_pass is taken on serial steps right behind the timed region, every launch alone"},
            "roofline_pyr_grad_pass": {"kernel": "all %d level launches of one pyramid+gradient pass" % levels,
                                       "bound": "hbm", "achieved": ach_pass, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                       "frac": ach_pass / HBM_PEAK_GBPS, "traffic": traffic_pass,
                                       "frac_moved": None if not traffic_pass else traffic_pass / (pass_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                       "algorithmic_bytes_per_pass": b_pass, "avg_pass_ms": pass_ms},
            "lk_kernel": {"kernel": lk_kernel_name, "avg_launch_ms": lk_ms,
                          "keypoints_per_sec_kernel_only": B * N_PTS / (lk_ms * 1e-3),
                          "unique_bytes_per_launch": lk_unique_bytes_per_pair(N_PTS, levels, WIN) * B,
                          "iterations": lk_iters,
                          "note": "VALU-issue bound, not HBM-bound (SURVEY 8d): no HBM fraction claimed",
                          "valu_issue": lk_valu_issue(lk_ms, B * N_PTS, lk_kernel_name) if args.workload == "kitti" else None},
            "hbm_copy_measured_GBps": copy_gbps,
            "host_api_pcie_inclusive": host_api,
            "stream_h2d_inclusive": stream,
            "frontend": frontend,
            "lk_other_windows": lk_other,
            "lk_real_frames": lk_real,
            "hd_share": hd_share,
            "post_lk_geometry": post_lk,
            "two_streams": two_streams,
            "sparse_align": align,
            "warp": warp,
            "paste_canvas_sharded": paste_sharded,
            "ref_pyramid_bcast": ref_bcast,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out), file=json_out, flush=True)

    if leg_hung:                 # a thread is still inside RCCL: no orderly teardown is possible, and the run FAILED
        print("bench.py: the reference-pyramid broadcast leg did not complete (see ref_pyramid_bcast.error)", file=sys.stderr)
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(3)
    for d in (d_prev, d_next, d_st, d_err):
        d.free()
    batch.close()
    ctx.close()
    dist.close()
    return 0


def hd_share_leg(dist, ctx, dr3, synth, sync, steps=5, pairs=64):
    """BASELINE configs[4], one GPU's share, bounded so that the driver's default run carries it: `pairs` 1920x1080 frame pairs per step
    (5000 corners, 5-level pyramid, 21x21 window, 30 iterations), `steps` timed steps of pyramid + gradient pass and tracker, inputs
    resident; the level-0 kernel against the HBM roofline with the committed PMC traffic of this launch shape; pair 0 against the oracle
    in the kernel's order (bit for bit) and in OpenCV's two (x86-64 SIMD and scalar).  `python bench.py --workload hd` is the long form."""
    W_, H_, NPTS, WIN_, ML = 1920, 1080, 5000, 21, 4
    frames, _ = stream_frames(synth, pairs + 1, W_, H_, period=3)
    G = pairs + 1                                               # the ring of the headline step: two slot groups (main())
    b = dr3.FrameBatch(ctx, W_, H_, win=(WIN_, WIN_), max_level=ML, capacity=2 * G)
    for g_ in range(2):
        for s0 in range(0, G, 16):
            b.upload(g_ * G + s0, np.stack(frames[s0:s0 + 16]))
    pts = synth.grid_points(W_, H_, 20, n=NPTS)
    P = np.ascontiguousarray(np.broadcast_to(pts[None], (pairs, NPTS, 2)), np.float32)
    d_prev, d_next = ctx.to_device(P), ctx.to_device(P)
    d_st, d_err = ctx.alloc(pairs * NPTS), ctx.alloc(pairs * NPTS * 4)
    kw = dict(win=(WIN_, WIN_), max_level=ML, max_iters=MAX_ITERS, eps=EPS, flags=0)
    params = dr3.lk_params(**kw)
    b.build(0, 2 * G, True)
    ring = [0]

    def step():
        g = ring[0] & 1; ring[0] += 1
        b.build_ahead((1 - g) * G + 1, pairs, True)
        b.track(pairs, g * G, d_prev, d_next, NPTS, d_st, d_err, params)

    for _ in range(8):
        step()
    sync()
    ctx.profile_enable(True); ctx.profile_reset()
    el = timed_region(dist, step, sync, steps, 0)
    ms_l0, n_l0 = ctx.profile_read(dr3.PROF_PYR_L0)
    ms_pass, n_pass = ctx.profile_read(dr3.PROF_PYR)
    ms_lk, n_lk = ctx.profile_read(dr3.PROF_LK)
    ctx.profile_enable(False)
    l0_ms, pass_ms, lk_ms = ms_l0 / max(n_l0, 1), ms_pass / max(n_pass, 1), ms_lk / max(n_lk, 1)
    b_l0 = pyr_grad_level0_bytes_per_frame(W_, H_) * pairs
    b_pass = pyr_grad_bytes_per_frame(W_, H_, b.levels) * pairs
    traffic, tsrc = pmc_traffic("pyr_level_kernel<true, true, 0>", pairs, "pmc_traffic_hd.json")
    verified, orders = None, None
    if dist.rank == 0:
        from oracle import oracle as O
        po = O.lk_params(**kw)
        g_xy = d_next.download((pairs, NPTS, 2), np.float32)[0]; g_st = d_st.download((pairs, NPTS), np.uint8)[0]
        res = {o: O.lk_track(frames[0], frames[1], pts, None, po, order=o, want_iters=True, opt=True) for o in (O.ORDER_WAVE64, O.ORDER_SSE2, O.ORDER_RASTER)}
        k = res[O.ORDER_WAVE64]
        verified = bool(np.array_equal(g_xy.view(np.uint32), k[0].view(np.uint32)) and np.array_equal(g_st, k[1]))
        g = (g_xy, g_st, None, k[3], k[4])
        orders = {"kernel_vs_sse2": O.order_agreement(g, res[O.ORDER_SSE2], po, 1e-3), "kernel_vs_raster": O.order_agreement(g, res[O.ORDER_RASTER], po, 1e-3),
                  "raster_vs_sse2": O.order_agreement(res[O.ORDER_RASTER], res[O.ORDER_SSE2], po, 1e-3)}
    out = {"workload": "configs[4], one GPU's share, bounded: %d pairs 1920x1080 per step, %d corners, %d-level pyramid, %dx%d window, %d timed steps"
                       % (pairs, NPTS, b.levels, WIN_, WIN_, steps),
           "keypoints_per_sec": aggregate_rate(dist, pairs * NPTS * steps, el), "ms_per_step": el / steps * 1e3,
           "roofline": {"kernel": "pyr_level_kernel<DOWN,DERIV,0> (fused pyrDown+Scharr, level 0)", "bound": "hbm",
                        "achieved": b_l0 / (l0_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": b_l0 / (l0_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                        "traffic": traffic, "traffic_source": tsrc,
                        "frac_moved": None if not traffic else traffic / (l0_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                        "algorithmic_bytes_per_launch": b_l0, "avg_launch_ms": l0_ms, "launches": n_l0},
           "pyr_grad_pass": {"span_ms_level0_start_to_last_level_end": pass_ms, "algorithmic_bytes_per_pass": b_pass,
                             "note": "the ring of the headline step: level 0 alone, levels 1.. under the tracker (`--workload hd` reports the pass with every launch alone)"},
           "lk_kernel": {"kernel": lk_kernel_of(WIN_, WIN_), "avg_launch_ms": lk_ms, "keypoints_per_sec_kernel_only": pairs * NPTS / (lk_ms * 1e-3)},
           "tracked_fraction": float(d_st.download((pairs, NPTS), np.uint8).mean()),
           "verified_vs_oracle": verified, "summation_order_agreement": orders}
    for d in (d_prev, d_next, d_st, d_err):
        d.free()
    b.close()
    return out


def frontend_leg(dist, ctx, dr3, batch, B, sync, steps):
    """Frame::Frame -> FastDetector::detect -> calcOpticalFlowPyrLK as one device-resident step over B frames (reference
    src/frame.cpp:13-20, :57, src/initialization.cpp:552-613): LK pyramids + gradients, the reference's 3-level
    half-sample pyramid with FAST-10 + grid-best Shi-Tomasi (cell 15 -> 83 x 26 = 2158 cells ~ configs[1]'s 2000
    corners), then every frame's corners tracked into the next frame.  Untimed extra, not the headline metric."""
    gc, gr = -(-batch.w // 15), -(-batch.h_px // 15)
    n_cells = gc * gr
    d_pts, d_next = ctx.alloc(B * n_cells * 8), ctx.alloc(B * n_cells * 8)
    d_st, d_err = ctx.alloc(B * n_cells), ctx.alloc(B * n_cells * 4)
    params = dr3.lk_params(win=(WIN, WIN), max_level=MAX_LEVEL, max_iters=MAX_ITERS, eps=EPS, flags=0)

    def step():
        batch.build(1, B, True)
        batch.detect(0, B, 3, 15, 20, 20.0, d_xy=d_pts)
        batch.track(B, 0, d_pts, d_next, n_cells, d_st, d_err, params)

    ctx.profile_enable(True)
    step(); sync()
    ctx.profile_reset()
    el = timed_region(dist, step, sync, steps, 0)
    ms_fast, n_fast = ctx.profile_read(dr3.PROF_FAST)
    ms_lk, n_lk = ctx.profile_read(dr3.PROF_LK)
    ctx.profile_enable(False)
    pts = d_pts.download((B, n_cells, 2), np.float32)
    st = d_st.download((B, n_cells), np.uint8)
    have = pts[..., 0] != dr3.NO_POINT
    out = {"frames_per_sec": aggregate_rate(dist, B * steps, el), "step_ms": el / steps * 1e3,
           "cells_per_frame": int(n_cells), "features_per_frame": float(have.sum() / B),
           "tracked_fraction_of_features": float(st[have].mean()) if have.any() else 0.0,
           "detect_ms_per_batch": ms_fast / max(n_fast, 1), "lk_ms_per_batch": ms_lk / max(n_lk, 1),
           "workload": "B = %d frames per step: LK pyramid + Scharr, half-sample pyramid (3 levels) + FAST-10 + grid-best "
                       "Shi-Tomasi (cell 15), LK %dx%d / %d levels from the detected corners" % (B, WIN, WIN, MAX_LEVEL + 1)}
    for d in (d_pts, d_next, d_st, d_err):
        d.free()
    return out


def align_leg(dist, ctx, dr3, synth, batch, frames, B, intr, sync, have_gpu, steps):
    """6-DoF inverse-compositional sparse image alignment (an extension specified by this build, the reference
    only names the step at src/handler.cpp:23): every consecutive pair of the resident stream, 546 features on a fronto-parallel
    plane 10 m away (the synthetic stream IS such a scene: a similarity of the image is a camera translation + roll in front of a
    plane), levels 3 -> 1 of the pyramid the batch already holds, 10 Gauss-Newton iterations per level, from the identity.
    One launch for all pairs; pair 0 against the CPU statement of the algorithm, bit for bit.  Untimed extra."""
    fx, fy, cx, cy = intr
    W_, H_ = batch.w, batch.h_px
    Kc = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    uv = synth.grid_points(W_, H_, 30)
    xyz1 = synth.plane_points(uv, Kc, n=(0.0, 0.0, 1.0), d=10.0)
    n = len(xyz1)
    xyz = np.ascontiguousarray(np.broadcast_to(xyz1[None], (B, n, 3)), np.float32)
    T0 = np.ascontiguousarray(np.broadcast_to(np.eye(3, 4)[None], (B, 3, 4)), np.float64)
    pa = dr3.align_params(fx, fy, cx, cy, max_level=3, min_level=1, n_iter=10, eps=1e-8)
    d_xyz, d_T = ctx.to_device(xyz), ctx.to_device(T0)
    d_n, d_c, d_i = ctx.alloc(4 * B), ctx.alloc(4 * B), ctx.alloc(4 * 8 * B)

    def step():
        d_T.upload(T0)
        batch.align(B, 0, d_xyz, n, pa, d_T, d_n, d_c, d_i)

    ctx.profile_enable(True)
    step(); sync()
    ctx.profile_reset()
    el = timed_region(dist, step, sync, steps, 0)
    ms, cnt = ctx.profile_read(dr3.PROF_ALIGN) if have_gpu else (1.0, 1)
    ctx.profile_enable(False)
    iters = d_i.download((B, 8), np.int32)
    ok = None
    if dist.rank == 0 and have_gpu:
        from oracle import oracle as O
        po = O.align_params(fx, fy, cx, cy, max_level=3, min_level=1, n_iter=10, eps=1e-8)
        oT, on, oc, oi = O.sparse_align(O.lk_pyramid(frames[0], batch.levels), O.lk_pyramid(frames[1], batch.levels), xyz1, po)
        ok = bool(np.array_equal(d_T.download((B, 3, 4), np.float64)[0].view(np.uint64), oT.view(np.uint64)) and
                  int(d_n.download((B,), np.int32)[0]) == on and np.array_equal(iters[0][:batch.levels], oi))
    k_ms = ms / max(cnt, 1)
    out = {"pairs_per_launch": B, "features_per_pair": n, "levels": [3, 1], "max_iterations_per_level": 10,
           "kernel_ms": k_ms, "pairs_per_sec_kernel_only": B / (k_ms * 1e-3),
           "pairs_per_sec_with_prior_upload": aggregate_rate(dist, B * steps, el),
           "patch_pixel_evaluations_per_sec_kernel_only": float(iters.sum()) * n * 16 / (k_ms * 1e-3),
           "mean_iterations_per_pair": float(iters.sum() / B), "mean_measurements": float(d_n.download((B,), np.int32).mean()),
           "verified_vs_cpu_statement": ok, "note": "extension specified by this build: no reference parity exists"}
    for d in (d_xyz, d_T, d_n, d_c, d_i):
        d.free()
    return out


def post_lk_leg(dist, ctx, dr3, d_prev, d_next, d_st, pts, n_pts, sync):
    """What the reference does with the tracked points (src/initialization.cpp:80-127, :365-410): InitHelper::Normalize of both
    point sets and the CheckFundamental vote of 200 RANSAC hypotheses, on pair 0's device-resident tracker output with its status
    bytes as the mask (nothing leaves the GPU).  Both kernels are sequential f32 sums by definition (a lane per hypothesis / per
    coordinate walks the matches in order), i.e. latency-bound: reported as time per call, host clock over 50 calls."""
    rng = np.random.default_rng(5)
    n_hyp = 200
    F = (rng.standard_normal((n_hyp, 9)) * np.array([1e-6, 1e-5, 1e-3, 1e-5, 1e-6, 1e-3, 1e-3, 1e-3, 1.0])).astype(np.float32)
    d_F, d_sc, d_inl = ctx.to_device(F), ctx.alloc(4 * n_hyp), ctx.alloc(n_hyp * n_pts)
    d_n1, d_n2, d_T = ctx.alloc(8 * n_pts), ctx.alloc(8 * n_pts), ctx.alloc(4 * 9 * 2)
    out = {"matches": n_pts, "hypotheses": n_hyp}

    def vote():
        ctx.check_fundamental_dev(d_F, n_hyp, d_prev, d_next, d_st, n_pts, 1.0, d_sc, d_inl)

    def norm():
        ctx.normalize_points_dev(d_prev, d_st, n_pts, 1, d_n1, d_T)
        ctx.normalize_points_dev(d_next, d_st, n_pts, 1, d_n2, d_T.ptr + 36)

    for name, fn in (("check_fundamental_ms", vote), ("normalize_two_sets_ms", norm)):
        fn(); sync()
        # host clock around batches of 10 calls, the median batch: one host-side stall of some tens of milliseconds (seen on some boxes,
        # see host_api_pcie_inclusive) inside a single window of 50 calls would otherwise read as a ten times slower kernel
        per_batch = []
        for _ in range(7):
            t0 = time.perf_counter()
            for _ in range(10):
                fn()
            sync()
            per_batch.append((time.perf_counter() - t0) / 10 * 1e3)
        out[name] = float(np.median(per_batch))
        out[name.replace("_ms", "_slowest_batch_ms")] = float(max(per_batch))
    if dist.rank == 0:
        from oracle import oracle as O
        p1 = d_prev.download((n_pts, 2), np.float32); p2 = d_next.download((n_pts, 2), np.float32); st = d_st.download((n_pts,), np.uint8)
        keep = st != 0
        sc = d_sc.download((n_hyp,), np.float32); inl = d_inl.download((n_hyp, n_pts), np.uint8)
        ok = True
        for k in (0, 57, 199):
            o_sc, o_inl = O.check_fundamental(F[k].reshape(3, 3), p1[keep], p2[keep], 1.0)
            ok = ok and np.float32(sc[k]).view(np.uint32) == np.float32(o_sc).view(np.uint32) and np.array_equal(inl[k][keep], o_inl)
        o_n1, o_T1 = O.normalize_points(p1, st)
        g_n1 = d_n1.download((n_pts, 2), np.float32); g_T = d_T.download((2, 3, 3), np.float32)
        ok = ok and np.array_equal(g_n1[keep].view(np.uint32), o_n1[keep].view(np.uint32)) and np.array_equal(g_T[0].view(np.uint32), o_T1.view(np.uint32))
        out["verified_vs_oracle"] = bool(ok)
    else:
        out["verified_vs_oracle"] = None
    return out


def two_streams_leg(dist, ctx, dr3, frames, P, B, n_pts, size, params, step_a, out_a, sync, steps):
    """The same step on TWO frame batches (two sequences of configs[3]'s kind on one GPU), each on its own context = its own HIP
    stream, enqueued alternately: nothing orders the two streams, so the pyramid pass of one batch (HBM-bound) runs beside the
    tracker of the other (VALU-issue bound).  Host clock around 2 x steps steps, against the same count of SERIAL steps (build, then
    track) on one stream measured the same way right before.  An extra beside `value` (in the timed region only the three short launches
    of levels 1.. run under the tracker, so its level-0 and tracker event times are those of kernels running alone)."""
    W_, H_ = size
    ctx2 = dr3.Context(ctx.device)
    b2 = dr3.FrameBatch(ctx2, W_, H_, win=(params.win_w, params.win_h), max_level=params.max_level, capacity=B + 1)
    for s0 in range(0, B + 1, 64):
        b2.upload(s0, np.stack(frames[s0:s0 + 64]))
    d_prev, d_next = ctx2.to_device(P), ctx2.to_device(P)
    d_st, d_err = ctx2.alloc(B * n_pts), ctx2.alloc(B * n_pts * 4)
    b2.build(0, B + 1, True)

    def step_b():
        b2.build(1, B, True)
        b2.track(B, 0, d_prev, d_next, n_pts, d_st, d_err, params)

    for _ in range(3):
        step_a(); step_b()
    sync(); ctx2.sync()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(2 * steps):
        step_a()
    sync()
    one = time.perf_counter() - t0
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_a(); step_b()
    sync(); ctx2.sync()
    two = time.perf_counter() - t0
    same = bool(np.array_equal(d_next.download((B, n_pts, 2), np.float32).view(np.uint32), out_a[0].download((B, n_pts, 2), np.float32).view(np.uint32))
                and np.array_equal(d_st.download((B, n_pts), np.uint8), out_a[1].download((B, n_pts), np.uint8)))
    out = {"steps": 2 * steps, "one_stream_ms_per_step": one / (2 * steps) * 1e3, "two_streams_ms_per_step": two / (2 * steps) * 1e3,
           "keypoints_per_sec_this_gpu": 2 * steps * B * n_pts / two, "gain": one / two - 1.0, "identical_to_one_stream": same,
           "note": "two frame batches on two HIP streams of this GPU, steps enqueued alternately; host clock, this rank only"}
    b2.close()
    ctx2.close()
    return out


def lk_other_windows_leg(dist, ctx, dr3, synth, frames, sync, have_gpu):
    """The tracker at the parameters the reference's call site really passes (src/initialization.cpp:593-613: 30x30 window,
    maxLevel 4 -> 3 at this size, 1000 -> 100 iterations, eps 1e-3, USE_INITIAL_FLOW with the identity guess of :578), one
    point per wave, and at BASELINE configs[0] (21x21, 3 levels, ~500 corners), two points per wave; 128 pairs x 546 grid corners per launch, resident inputs, kernel time from HIP events, pair 0 against the oracle.
    Untimed extras beside the headline metric."""
    W_, H_, NP = len(frames[0][0]), len(frames[0]), 128
    pts = synth.grid_points(W_, H_, 30)
    n = len(pts)
    P = np.ascontiguousarray(np.broadcast_to(pts[None], (NP, n, 2)), np.float32)
    out = {}
    cases = (("reference_literals_30x30", dict(win=(30, 30), max_level=4, max_iters=1000, eps=1e-3, flags=dr3.LK_USE_INITIAL_FLOW if have_gpu else 4),
              lk_kernel_of(30, 30)),
             ("configs0_21x21_3_levels", dict(win=(21, 21), max_level=2, max_iters=30, eps=0.01, flags=0), lk_kernel_of(21, 21)))
    for name, kw, kernel in cases:
        b = dr3.FrameBatch(ctx, W_, H_, win=kw["win"], max_level=kw["max_level"], capacity=NP + 1)
        for s0 in range(0, NP + 1, 64):
            b.upload(s0, np.stack(frames[s0:min(s0 + 64, NP + 1)]))
        b.build(0, NP + 1, True)
        d_prev, d_next = ctx.to_device(P), ctx.to_device(P)
        d_st, d_err = ctx.alloc(NP * n), ctx.alloc(NP * n * 4)
        params = dr3.lk_params(**kw)
        ctx.profile_enable(True)
        for _ in range(2):
            d_next.upload(P); b.track(NP, 0, d_prev, d_next, n, d_st, d_err, params)
        sync(); ctx.profile_reset()
        reps = 5
        for _ in range(reps):
            d_next.upload(P)                              # USE_INITIAL_FLOW reads next_xy: every launch starts from the identity guess
            b.track(NP, 0, d_prev, d_next, n, d_st, d_err, params)
        sync()
        ms, cnt = ctx.profile_read(dr3.PROF_LK)
        ctx.profile_enable(False)
        ok, orders = None, None
        if dist.rank == 0 and have_gpu:
            from oracle import oracle as O
            po = O.lk_params(**kw)
            o = O.lk_track(frames[0], frames[1], pts, pts.copy() if kw["flags"] & 4 else None, po, order=O.ORDER_WAVE64)
            g_xy = d_next.download((NP, n, 2), np.float32)[0]; g_st = d_st.download((NP, n), np.uint8)[0]
            ok = bool(np.array_equal(g_xy.view(np.uint32), o[0].view(np.uint32)) and np.array_equal(g_st, o[1]) and
                      np.array_equal(d_err.download((NP, n), np.float32)[0].view(np.uint32), o[2].view(np.uint32)))
            # the three summation orders on this configuration's pair 0 (north_star's 1e-3 px bar)
            guess = pts.copy() if kw["flags"] & 4 else None
            res = {od: O.lk_track(frames[0], frames[1], pts, guess, po, order=od, want_iters=True, opt=True) for od in (O.ORDER_WAVE64, O.ORDER_SSE2, O.ORDER_RASTER)}
            g = (g_xy, g_st, None, res[O.ORDER_WAVE64][3], res[O.ORDER_WAVE64][4])
            orders = {"kernel_vs_sse2": O.order_agreement(g, res[O.ORDER_SSE2], po, 1e-3), "kernel_vs_raster": O.order_agreement(g, res[O.ORDER_RASTER], po, 1e-3),
                      "raster_vs_sse2": O.order_agreement(res[O.ORDER_RASTER], res[O.ORDER_SSE2], po, 1e-3)}
        k_ms = ms / max(cnt, 1)
        out[name] = {"kernel": kernel, "levels": b.levels, "pairs_per_launch": NP, "points_per_pair": n, "kernel_ms": k_ms,
                     "keypoints_per_sec_kernel_only": NP * n / (k_ms * 1e-3), "tracked_fraction": float(d_st.download((NP, n), np.uint8).mean()),
                     "verified_vs_oracle": ok, "summation_order_agreement": orders}
        if have_gpu:                                       # iteration-levels per point and per lock-step group (what tools/lk_valu_mix.py weights by)
            d_it = ctx.alloc(NP * n * b.levels * 4)
            d_next.upload(P); b.track(NP, 0, d_prev, d_next, n, d_st, d_err, params, d_iters=d_it); sync()
            itl = d_it.download((NP, n, b.levels), np.int32)
            it = itl.sum(2)
            ppw = int(kernel.split(",")[1])
            out[name]["mean_iteration_levels_per_point"] = float(it.mean())
            out[name]["mean_iteration_levels_of_slowest_in_group"] = float(it[:, :n // ppw * ppw].reshape(NP, -1, ppw).max(2).mean())
            out[name]["executed_iteration_levels_per_wave"] = float(itl[:, :n // ppw * ppw].reshape(NP, -1, ppw, b.levels).max(2).sum(2).mean())
            d_it.free()
        for d in (d_prev, d_next, d_st, d_err):
            d.free()
        b.close()
    return out


def stream_leg(dist, ctx, dr3, frames, P, B, n_pts, size, params, sync, steps, resident_xy, resident_st, have_gpu):
    """The headline step fed from the HOST, the way the reference receives every frame (src/handler.cpp:31-34,
    tests/test_pipeline.cpp:71-73): every step uploads B + 1 NEW frames from pinned host memory (dr3hip_batch_upload_async: copy stream,
    ring of device bounce buffers, pitch change + border on the compute stream), builds their pyramids and gradients, tracks B pairs and
    downloads positions and status (dr3hip_download_async).  The upload of step s + 1 is in flight while step s computes; two frame
    batches and two result sets alternate.  Three host frame sets rotate (set 0 = the resident run's frames: its results must come out
    bit for bit).  Reported beside `value`, never as `value`.  The frame carried over between batches is uploaded again (1 / (B + 1))."""
    if not have_gpu:
        return None
    import torch
    W, H = size
    n_sets = 3
    L = len(frames)
    host = [ctx.host_alloc((B + 1, H, W)) for _ in range(n_sets)]
    for k, hb in enumerate(host):
        for i in range(B + 1):
            hb[i] = frames[(i + 2 * k) % L] if k else frames[i]
    batches = [dr3.FrameBatch(ctx, W, H, win=(WIN, WIN), max_level=MAX_LEVEL, capacity=B + 1) for _ in range(2)]
    d_prev = ctx.to_device(P)
    res = [(ctx.to_device(P), ctx.alloc(B * n_pts), ctx.alloc(B * n_pts * 4)) for _ in range(2)]
    out = [(ctx.host_alloc((B, n_pts, 2), np.float32), ctx.host_alloc((B, n_pts), np.uint8)) for _ in range(2)]
    keep = {}

    def run(n_steps, check_step=None):
        batches[0].upload_async(0, host[0])
        for s in range(n_steps):
            b, r, o = batches[s % 2], res[s % 2], out[s % 2]
            b.build(0, B + 1, True)
            b.track(B, 0, d_prev, r[0], n_pts, r[1], r[2], params)
            batches[(s + 1) % 2].upload_async(0, host[(s + 1) % n_sets])      # crosses the link while step s computes
            ctx.download_wait()                                               # the results of step s - 1 are on the host
            if check_step is not None and s - 1 == check_step:
                keep["xy"], keep["st"] = out[(s - 1) % 2][0].copy(), out[(s - 1) % 2][1].copy()
            ctx.download_async(o[0], r[0]); ctx.download_async(o[1], r[1])
        ctx.download_wait()
        sync()

    run(4)
    dist.barrier(); sync(); t0 = time.perf_counter()
    run(steps, check_step=n_sets if steps > n_sets + 1 else None)             # step n_sets uses frame set 0 again
    sync(); dist.barrier()
    el = dist.max_over_ranks(time.perf_counter() - t0)
    same = None
    if "xy" in keep and resident_xy is not None:
        same = bool(np.array_equal(keep["xy"].view(np.uint32), resident_xy.view(np.uint32)) and np.array_equal(keep["st"], resident_st))
    # what the link delivers on this box: one pinned -> device copy of a frame set, HIP events on torch's stream
    nbytes = host[0].nbytes
    link = None
    try:
        th = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True); td = torch.empty(nbytes, dtype=torch.uint8, device="cuda:%d" % ctx.device)
        td.copy_(th, non_blocking=True); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            td.copy_(th, non_blocking=True)
        e1.record(); torch.cuda.synchronize()
        link = 5 * nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del th, td
    except Exception:                                                         # noqa: BLE001 -- informational only
        link = None
    up_bytes = nbytes * (steps + 1)
    down_bytes = (out[0][0].nbytes + out[0][1].nbytes) * steps
    h2d = up_bytes / el / 1e9
    kps = aggregate_rate(dist, B * n_pts * steps, el)
    for b in batches:
        b.upload_wait(); b.close()
    for hb in host:
        ctx.host_free(hb)
    for a, c in out:
        ctx.host_free(a); ctx.host_free(c)
    for r in res:
        for d in r:
            d.free()
    d_prev.free()
    return {"keypoints_per_sec": kps, "ms_per_step": el / steps * 1e3, "steps": steps, "frames_uploaded_per_step": B + 1,
            "h2d_GBps_achieved": h2d, "h2d_GBps_pinned_copy_measured": link, "h2d_frac_of_measured_link": None if not link else h2d / link,
            "d2h_bytes_per_step": down_bytes // max(steps, 1), "bound": "host link (PCIe)" if link and h2d >= 0.7 * link else "compute / pipeline",
            "results_identical_to_resident_run": same,
            "note": "inputs cross PCIe every step: never `value`; a step is B + 1 frames up (pinned host memory), pyramid + gradient pass, "
                    "tracker, positions + status down"}


def lk_real_pair_leg(dist, ctx, dr3, sync, have_gpu):
    """The tracker on REAL frames whose flows differ from point to point: the reference's own two-view pair (tests/golden/kitti_pair.npz =
    imgs/kitti0.png / kitti5.png of reference tests/test_init.cpp:24-25, five frames apart on a driving sequence), tiled to the headline
    launch shape -- 256 pairs in 257 slots A B A B ..., even pairs track A -> B from A's FAST corners, odd pairs B -> A from B's (one
    corner per 15-px cell, padded to a common count with DR3HIP_NO_POINT) -- at the headline's parameters (15x15, 4 levels, 30 iterations,
    eps 0.01) and at the reference's call-site literals (src/initialization.cpp:593-613: 30x30, 1000 -> 100 iterations, eps 1e-3,
    USE_INITIAL_FLOW from the identity).  Reports kernel ms, the mean number of Gauss-Newton iteration-levels per point and the mean
    over lock-step groups of the slowest point's (what a wave executes), pair 0 against the oracle.  Untimed extra."""
    fix = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "kitti_pair.npz")
    if not have_gpu or not os.path.exists(fix):
        return None
    z = np.load(fix)
    A, Bf = np.ascontiguousarray(z["kitti0"]), np.ascontiguousarray(z["kitti5"])
    H_, W_ = A.shape
    NP = 256

    def corners(img):
        x, y, sc, lv = ctx.fast_detect(img, 3, 15, 20, 20.0)
        m = lv >= 0
        return np.stack([x[m], y[m]], 1).astype(np.float32)
    pa, pb = corners(A), corners(Bf)
    n = max(len(pa), len(pb))
    pad = lambda q: np.concatenate([q, np.full((n - len(q), 2), dr3.NO_POINT, np.float32)])
    pa_, pb_ = pad(pa), pad(pb)
    P = np.ascontiguousarray(np.stack([pa_ if i % 2 == 0 else pb_ for i in range(NP)]), np.float32)
    valid = int(sum(len(pa) if i % 2 == 0 else len(pb) for i in range(NP)))
    out = {"frames": "reference imgs/kitti0.png, kitti5.png (1240x376), %d / %d FAST corners" % (len(pa), len(pb)),
           "pairs_per_launch": NP, "point_slots_per_pair": n, "valid_points_per_launch": valid}
    cases = (("headline_15x15", dict(win=(15, 15), max_level=3, max_iters=30, eps=0.01, flags=0)),
             ("reference_literals_30x30", dict(win=(30, 30), max_level=4, max_iters=1000, eps=1e-3, flags=dr3.LK_USE_INITIAL_FLOW)))
    for name, kw in cases:
        b = dr3.FrameBatch(ctx, W_, H_, win=kw["win"], max_level=kw["max_level"], capacity=NP + 1)
        for s0 in range(0, NP + 1, 64):
            b.upload(s0, np.stack([A if i % 2 == 0 else Bf for i in range(s0, min(s0 + 64, NP + 1))]))
        b.build(0, NP + 1, True)
        lv = b.levels
        d_prev, d_next = ctx.to_device(P), ctx.to_device(P)
        d_st, d_err, d_it = ctx.alloc(NP * n), ctx.alloc(NP * n * 4), ctx.alloc(NP * n * lv * 4)
        params = dr3.lk_params(**kw)
        ctx.profile_enable(True)
        for _ in range(2):
            d_next.upload(P); b.track(NP, 0, d_prev, d_next, n, d_st, d_err, params)
        sync(); ctx.profile_reset()
        for _ in range(5):
            d_next.upload(P)                                  # USE_INITIAL_FLOW reads next_xy: every launch starts from the identity guess
            b.track(NP, 0, d_prev, d_next, n, d_st, d_err, params)
        sync()
        ms, cnt = ctx.profile_read(dr3.PROF_LK)
        ctx.profile_enable(False)
        d_next.upload(P); b.track(NP, 0, d_prev, d_next, n, d_st, d_err, params, d_iters=d_it); sync()      # untimed: with the iteration counts
        itl_ = d_it.download((NP, n, lv), np.int32)
        it = itl_.sum(2)                                                                                    # iteration-levels per point slot
        ppw = int(lk_kernel_of(*kw["win"]).split(",")[1])
        vm = (P[:, :, 0] > -1.0e5)
        grp = it[:, :n // ppw * ppw].reshape(NP, -1, ppw)
        grp_live = vm[:, :n // ppw * ppw].reshape(NP, -1, ppw).any(2)                                         # groups that hold a point at all
        g_xy = d_next.download((NP, n, 2), np.float32); g_st = d_st.download((NP, n), np.uint8)
        ok = None
        if dist.rank == 0:
            from oracle import oracle as O
            po = O.lk_params(**{**kw, "flags": (O.LK_USE_INITIAL_FLOW if kw["flags"] else 0)})
            o = O.lk_track(A, Bf, pa, pa.copy() if kw["flags"] else None, po, order=O.ORDER_WAVE64)
            ok = bool(np.array_equal(g_xy[0, :len(pa)].view(np.uint32), o[0].view(np.uint32)) and np.array_equal(g_st[0, :len(pa)], o[1]))
        k_ms = ms / max(cnt, 1)
        out[name] = {"kernel": lk_kernel_of(*kw["win"]), "levels": lv, "kernel_ms": k_ms,
                     "keypoints_per_sec_kernel_only": valid / (k_ms * 1e-3), "tracked_fraction_of_valid": float(g_st[vm].mean()),
                     "mean_iteration_levels_per_point": float(it[vm].mean()), "points_per_wave": ppw,
                     "mean_iteration_levels_of_slowest_in_group": float(grp.max(2)[grp_live].mean()),
                     "executed_iteration_levels_per_wave": float(itl_[:, :n // ppw * ppw].reshape(NP, -1, ppw, lv).max(2).sum(2)[grp_live].mean()),
                     "points_at_the_iteration_clamp_frac": float((d_it.download((NP, n, lv), np.int32).max(2)[vm] >= min(kw["max_iters"], 100)).mean()),
                     "verified_vs_oracle": ok}
        for d in (d_prev, d_next, d_st, d_err, d_it):
            d.free()
        b.close()
    return out


def ref_bcast_leg(dist, ctx, dr3, batch, frames, levels, sync):
    """rank 0 builds slot 0, dr3hip_batch_bcast_slot sends it to every rank, every rank checks it holds the same
    planes.  Every torch.distributed call runs on every rank whatever happened locally, so a rank that failed
    cannot leave the others waiting in a collective."""
    err, moved, ms, got = None, 0, 0.0, 0.0
    try:
        uid_root = ctx.comm_unique_id() if dist.rank == 0 else b"\0" * dr3.COMM_ID_BYTES
    except Exception as e:                                       # noqa: BLE001
        uid_root, err = b"\0" * dr3.COMM_ID_BYTES, str(e)
    uid = dist.bcast_bytes(uid_root, dr3.COMM_ID_BYTES)
    if dist.max_over_ranks(1.0 if err else 0.0) == 0.0:          # the root has an id: all ranks enter the init
        try:
            ctx.comm_init(uid, dist.world, dist.rank)
            batch.upload(0, frames[0] if dist.rank == 0 else np.zeros_like(frames[0]))   # receivers start from zeros
            batch.build(0, 1, True)
            sync()
            ctx.timer_start()
            moved = batch.bcast_slot(0, root=0, with_deriv=True)
            ms = ctx.timer_stop()
            got = float(batch.download_image(0, levels - 1).astype(np.int64).sum()
                        + np.abs(batch.download_deriv(0, 0).astype(np.int64)).sum())
            ctx.comm_destroy()
        except Exception as e:                                   # noqa: BLE001 -- reported, not fatal
            err = str(e)
    n_err = dist.sum_over_ranks(1.0 if err else 0.0)
    hi, lo, ms_max = dist.max_over_ranks(got), -dist.max_over_ranks(-got), dist.max_over_ranks(ms)
    return {"bytes_per_rank": int(moved), "ms": ms_max, "identical_on_all_ranks": bool(n_err == 0 and hi == lo and hi > 0),
            "transport": "RCCL ncclBroadcast x %d in one group (dr3hip_batch_bcast_slot)" % (2 * levels),
            "ranks_with_error": int(n_err), "error": (err or "")[:300] or None}


def warp_leg(dist, ctx, dr3, synth, sync, steps):
    """configs[2]: 8 tiles 4000x3000, one homography per tile evaluated per pixel, bilinear, border 0.
    Run as 8UC3 (6 B/px) and as the reference's real 32FC4 form (32 B/px, src/panorama.cpp:178-192)."""
    TW_, TH_, NT_ = 4000, 3000, 8
    tile = cheap_tile(synth, TW_, TH_)
    res = {}
    for name, typ, esz in (("8uc3", dr3.T_8UC3, 3), ("32fc4", dr3.T_32FC4, 16)):
        if typ == dr3.T_32FC4:
            host = np.ones((TH_, TW_, 4), np.float32)
            host[..., :3] = tile
        else:
            host = tile
        stride = TW_ * esz
        srcs = [ctx.to_device(host) for _ in range(NT_)]
        dsts = [ctx.alloc(TH_ * stride) for _ in range(NT_)]
        jobs = []
        for i in range(NT_):
            M = synth.tile_homography(i, TW_, TH_)
            M[0, 2] -= 60.0; M[1, 2] -= 60.0            # keep most of each tile inside its canvas
            jobs.append(dict(src=srcs[i].ptr, sw=TW_, sh=TH_, sstride=stride,
                             dst=dsts[i].ptr, dw=TW_, dh=TH_, dstride=stride, M=M))
        ctx.profile_enable(True)
        dr3.warp_batch(ctx, jobs, typ)                   # warm-up
        sync()
        ctx.profile_reset()
        el = timed_region(dist, lambda: dr3.warp_batch(ctx, jobs, typ), sync, steps, 0)
        ms, n = ctx.profile_read(dr3.PROF_WARP)
        ctx.profile_enable(False)
        mpix = aggregate_rate(dist, NT_ * TW_ * TH_ * steps / 1e6, el)
        k_ms = ms / max(n, 1)
        ok = None
        if dist.rank == 0:                                # tile 0 of the timed launch against the oracle, outside the timed region
            from oracle import oracle as O
            got = dsts[0].download(host.shape, host.dtype)
            ok = bool(np.array_equal(got.view(np.uint8), O.warp_perspective(host, jobs[0]["M"], (TW_, TH_)).view(np.uint8)))
        alg = NT_ * TW_ * TH_ * 2 * esz                  # esz read + esz written per dst pixel
        plan = dr3.warp_plan(jobs[0], typ)
        kname = "warp_tile_8uc3_kernel" if plan["kernel"] == 2 else "warp_perspective_kernel<%d," % typ
        if plan["kernel"] == 2:                              # the strip kernel's grid (launch_warp): 64 x 64 tiles, padded to whole XCD rounds
            gthreads = strip_grid_threads(TW_, TH_, NT_)
        else:                                                # the direct kernel (launch_warp): a 256-thread block of four wave tiles, stacked in y
            gthreads = direct_grid_threads(TW_, TH_, NT_, typ in (dr3.T_8UC1, dr3.T_8UC3))
        traffic, tsrc = pmc_traffic(kname, 256, "pmc_traffic_warp.json", gthreads)
        res[name] = {"mpix_per_s": mpix, "kernel_ms": k_ms, "verified_vs_oracle": ok, "kernel_plan": plan,
                     "roofline": {"kernel": ("warp_tile_8uc3_kernel (LDS-staged strips)" if plan["kernel"] == 2 else "warp_perspective_kernel<%s>" % name.upper()), "bound": "hbm",
                                  "achieved": alg / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                  "frac": alg / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "traffic": traffic,
                                  "traffic_source": tsrc, "algorithmic_bytes_per_launch": alg}}
        for d in srcs + dsts:
            d.free()
    res["workload"] = "configs[2]: 8 tiles 4000x3000, per-pixel homography, bilinear, border 0"
    res["kitti_frames_8uc3"] = kitti_warp_leg(dist, ctx, dr3, synth, sync, tile, steps)
    res["paste"] = paste_leg(dist, ctx, dr3, synth, sync, tile, steps)
    res["stitch"] = stitch_leg(dist, ctx, dr3, synth, sync, tile, steps)
    res["halfsample"] = halfsample_leg(dist, ctx, dr3, sync, tile, steps)
    res["remap"] = remap_leg(dist, ctx, dr3, sync, tile, steps)
    return res


def rotating_sets(set_bytes):
    """how many distinct buffer sets a leg whose launch touches set_bytes must rotate over so that nothing it reads or writes is still in
    the 256 MiB Infinity Cache from its previous use: at least 1 GiB of other sets' traffic lies between two uses of a set"""
    return int(-(-(1 << 30) // max(int(set_bytes), 1))) + 1


def remap_leg(dist, ctx, dr3, sync, tile, steps):
    """cv::remap as utils::warp_local calls it (reference src/utils.cpp:191) on one device-resident 4000x3000 8UC3 tile, twice:
    driven by the spherical maps read from two f32 planes (row a6: 3 + 8 B read, 3 B written per pixel) and with the maps generated
    inside the kernel (row f3, utils::warp_spherical, src/utils.cpp:196-201: 3 + 3 B per pixel).  Both against the oracle, whole
    tile; the two outputs must also be equal to each other."""
    h, w = tile.shape[:2]
    f = 0.9 * w
    lib = dr3.lib()
    out = {}
    if dist.rank == 0:
        from oracle import oracle as O
        u, v = O.spherical_map(w, h, f)
    else:
        u = v = np.zeros((h, w), np.float32)
    # launches rotate over distinct buffer sets (rotating_sets): every launch streams from and to HBM, not the Infinity Cache
    n_a, n_b = rotating_sets(h * w * (3 + 8 + 3)), rotating_sets(h * w * 6)
    sets_a = [(ctx.to_device(tile), ctx.to_device(u), ctx.to_device(v), ctx.alloc(h * w * 3)) for _ in range(n_a)]
    sets_b = [(sets_a[i][0] if i < n_a else ctx.to_device(tile), ctx.alloc(h * w * 3)) for i in range(n_b)]
    it = {"a": 0, "b": 0}

    def run_a():
        d_src, d_u, d_v, d_a = sets_a[it["a"] % n_a]; it["a"] += 1
        ctx._chk(lib.dr3hip_remap_bilinear_dev(ctx.h, d_src.ptr, w, h, 3 * w, dr3.T_8UC3, d_a.ptr, w, h, 3 * w, d_u.ptr, d_v.ptr, 4 * w))

    def run_b():
        d_src, d_b = sets_b[it["b"] % n_b]; it["b"] += 1
        ctx._chk(lib.dr3hip_warp_local_dev(ctx.h, d_src.ptr, w, h, 3 * w, dr3.T_8UC3, d_b.ptr, 3 * w, dr3.WARP_SPHERICAL, C_float(f)))

    for name, bpp, run, n_sets in (("maps_from_memory", 14, run_a, n_a), ("spherical_maps_in_kernel", 6, run_b, n_b)):
        ctx.profile_enable(True)
        for _ in range(n_sets):
            run()
        sync(); ctx.profile_reset()
        timed_region(dist, run, sync, max(steps, n_sets), 0)
        ms, n = ctx.profile_read(dr3.PROF_REMAP)
        ctx.profile_enable(False)
        k_ms = ms / max(n, 1)
        ach = bpp * w * h / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        out[name] = {"kernel_ms": k_ms, "buffer_sets_rotated": n_sets,
                     "roofline": {"kernel": "remap_kernel", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS,
                                  "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS, "traffic": None,
                                  "algorithmic_bytes_per_launch": bpp * w * h}}
    if dist.rank == 0:
        a, b = sets_a[0][3].download((h, w, 3), np.uint8), sets_b[0][1].download((h, w, 3), np.uint8)
        ref = O.remap(tile, u, v, n_threads=0)
        out["verified_vs_oracle"] = bool(np.array_equal(a, ref) and np.array_equal(b, ref))
    else:
        out["verified_vs_oracle"] = None
    for t in sets_a:
        for d in t:
            d.free()
    for i, t in enumerate(sets_b):
        if i >= n_a:
            t[0].free()
        t[1].free()
    out["image"] = [w, h]
    return out


def halfsample_leg(dist, ctx, dr3, sync, tile, steps):
    """utils::reduce_to_half (reference src/utils.cpp:382-419, from Frame's create_img_pyramid, src/frame.cpp:18) on a device-resident
    16000x12000 u8 image (16 copies of the 4000x3000 tile's grey channel), the mode the reference executes on x86-64; 1.25 B per
    source pixel.  One quadrant against the oracle."""
    grey = np.ascontiguousarray(tile[..., 1])
    big = np.tile(grey, (4, 4))
    h, w = big.shape
    nbytes = w * h + (w // 2) * (h // 2)
    n_sets = rotating_sets(nbytes)                              # 240 MB per launch would sit in the 256 MiB Infinity Cache: rotate
    sets = [(ctx.to_device(big), ctx.alloc((h // 2) * (w // 2))) for _ in range(n_sets)]
    lib = dr3.lib()
    it = [0]

    def run():
        d_src, d_dst = sets[it[0] % n_sets]; it[0] += 1
        ctx._chk(lib.dr3hip_halfsample_dev(ctx.h, d_src.ptr, w, h, w, d_dst.ptr, w // 2, dr3.HALF_REF_X86))

    ctx.profile_enable(True)
    for _ in range(n_sets):
        run()
    sync(); ctx.profile_reset()
    timed_region(dist, run, sync, max(steps, n_sets), 0)
    ms, n = ctx.profile_read(dr3.PROF_HALF)
    ctx.profile_enable(False)
    k_ms = ms / max(n, 1)
    ok = None
    if dist.rank == 0:
        from oracle import oracle as O
        got = sets[0][1].download((h // 2, w // 2), np.uint8)
        ok = bool(np.array_equal(got[:1500, :2000], O.halfsample(grey, O.HALF_REF_X86)) and np.array_equal(got[:1500, :2000], got[1500:3000, 2000:4000]))
    for a_, b_ in sets:
        a_.free(); b_.free()
    ach = nbytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    return {"image": [w, h], "kernel_ms": k_ms, "verified_vs_oracle": ok, "buffer_sets_rotated": n_sets,
            # 16-byte aligned rows, the SSE2 arithmetic (REF_X86 at a width that is a multiple of 16) and >= 512 output pixels per row:
            # the streaming form (launch_halfsample in warp_kernels.hip)
            "roofline": {"kernel": "halfsample_wide_kernel" if (w % 16 == 0 and w // 2 >= 512) else "halfsample_kernel", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBPS, "traffic": None, "algorithmic_bytes_per_launch": nbytes}}


def kitti_warp_leg(dist, ctx, dr3, synth, sync, tile, steps):
    """the second half of BASELINE.json's metric: warped Mpix/s on 1241x376 frames -- 256 KITTI-sized 8UC3 frames, one
    homography each (the reference stitches KITTI frame pairs, imgs/results/stitch/KITTI07.jpg), all in ONE launch (the job
    index is a grid axis), sources and destinations resident in HBM"""
    W_, H_, NJ = 1241, 376, 256
    frame = np.ascontiguousarray(tile[:H_, :W_])
    pitch = (3 * W_ + 255) // 256 * 256
    host = np.zeros((H_, pitch), np.uint8); host[:, :3 * W_] = frame.reshape(H_, 3 * W_)
    d_src = ctx.to_device(np.broadcast_to(host[None], (NJ, H_, pitch)).copy())
    d_dst = ctx.alloc(NJ * H_ * pitch)
    jobs = []
    for i in range(NJ):
        M = synth.tile_homography(i, W_, H_)
        M[0, 2] -= 60.0; M[1, 2] -= 60.0
        jobs.append(dict(src=d_src.ptr + i * H_ * pitch, sw=W_, sh=H_, sstride=pitch,
                         dst=d_dst.ptr + i * H_ * pitch, dw=W_, dh=H_, dstride=pitch, M=M))
    ctx.profile_enable(True)
    dr3.warp_batch(ctx, jobs, dr3.T_8UC3); sync()
    ctx.profile_reset()
    el = timed_region(dist, lambda: dr3.warp_batch(ctx, jobs, dr3.T_8UC3), sync, steps, 0)
    ms, n = ctx.profile_read(dr3.PROF_WARP)
    ctx.profile_enable(False)
    ok = None
    if dist.rank == 0:
        from oracle import oracle as O
        got = d_dst.download((H_, pitch), np.uint8, offset=3 * H_ * pitch)[:, :3 * W_].reshape(H_, W_, 3)     # frame 3 of the launch
        ok = bool(np.array_equal(got, O.warp_perspective(frame, jobs[3]["M"], (W_, H_))))
    d_src.free(); d_dst.free()
    k_ms = ms / max(n, 1)
    alg = NJ * W_ * H_ * 6
    return {"frames_per_launch": NJ, "kernel_ms": k_ms, "verified_vs_oracle": ok, "mpix_per_s": aggregate_rate(dist, NJ * W_ * H_ * steps / 1e6, el),
            "kernel_only_mpix_per_s": NJ * W_ * H_ / 1e6 / (k_ms * 1e-3),
            "kernel_plan": dr3.warp_plan(jobs[3], dr3.T_8UC3),
            "roofline": {"kernel": "warp_tile_8uc3_kernel (LDS-staged strips)" if dr3.warp_plan(jobs[3], dr3.T_8UC3)["kernel"] == 2 else "warp_perspective_kernel<8UC3>",
                         "bound": "hbm", "achieved": alg / (k_ms * 1e-3) / 1e9,
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": alg / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                         "traffic": pmc_traffic("warp_tile_8uc3_kernel", 256, "pmc_traffic_warp.json", strip_grid_threads(W_, H_, NJ))[0]
                                    if dr3.warp_plan(jobs[3], dr3.T_8UC3)["kernel"] == 2 else None,
                         "algorithmic_bytes_per_launch": alg}}


def stitch_leg(dist, ctx, dr3, synth, sync, tile, steps):
    """reconstruct::Stitch::process after align_pair (src/stitch.cpp:33-80) on one 4000x3000 pair: geometry on the host, both
    8UC3 warps and the addWeighted blend as ONE kernel (the reference: two full-canvas warps + one blend pass)."""
    TH_, TW_ = tile.shape[:2]
    a = np.deg2rad(1.5)
    H = np.array([[np.cos(a), -np.sin(a), -0.45 * TW_], [np.sin(a), np.cos(a), 35.0], [1e-5, -6e-6, 1.0]], np.float32)
    Hinv, T, M, (nw, nh) = dr3.stitch_geometry(H, TW_, TH_)
    op = (3 * nw + 255) // 256 * 256
    right = np.roll(tile, -int(0.45 * TW_), axis=1)
    n_sets = rotating_sets(2 * TW_ * TH_ * 3 + op * nh)         # 130 MB per launch would sit in the 256 MiB Infinity Cache: rotate
    sets = [(ctx.to_device(tile), ctx.to_device(right), ctx.alloc(op * nh)) for _ in range(n_sets)]
    d_l, d_r, d_o = sets[0]
    Tc, Mc = np.ascontiguousarray(T.reshape(9)), np.ascontiguousarray(M.reshape(9))
    L = dr3.lib()
    it = [0]

    def step():
        a_l, a_r, a_o = sets[it[0] % n_sets]; it[0] += 1
        ctx._chk(L.dr3hip_stitch_compose_dev(ctx.h, a_l.ptr, TW_, TH_, 3 * TW_, a_r.ptr, TW_, TH_, 3 * TW_, Tc.ctypes.data, Mc.ctypes.data,
                                             nw, nh, 0.5, 0.5, 0.0, a_o.ptr, op))

    ctx.profile_enable(True)
    for _ in range(n_sets):
        step()
    sync()
    ctx.profile_reset()
    steps = max(steps, n_sets)
    el = timed_region(dist, step, sync, steps, 0)
    ms, n = ctx.profile_read(dr3.PROF_STITCH)
    ctx.profile_enable(False)
    ok = None
    if dist.rank == 0:                                     # the timed pair against the oracle's two warps + addWeighted
        from oracle import oracle as O
        got = d_o.download((nh, op), np.uint8)[:, :3 * nw].reshape(nh, nw, 3)
        ok = bool(np.array_equal(got, O.stitch_compose(tile, right, H)))
    for t in sets:
        for d in t:
            d.free()
    k_ms = ms / max(n, 1)
    alg = 2 * TW_ * TH_ * 3 + nw * nh * 3                 # both sources read once + the canvas written once
    plan = dr3.stitch_plan(TW_, TH_, TW_, TH_, T, M, (nw, nh), lstride=3 * TW_, rstride=3 * TW_)
    return {"canvas": [nw, nh], "kernel_ms": k_ms, "verified_vs_oracle": ok, "buffer_sets_rotated": n_sets, "canvas_mpix_per_s": aggregate_rate(dist, nw * nh * steps / 1e6, el),
            "kernel_plan": plan,
            "roofline": {"kernel": "stitch_strip_kernel (right image on LDS-staged strips, left image as its translation)" if plan == 2 else "stitch_compose_kernel", "bound": "hbm", "achieved": alg / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": alg / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "traffic": None, "algorithmic_bytes_per_launch": alg},
            "replaces": "reference passes: warp right (3 r + 3 w per canvas px), warp left (3 r + 3 w), addWeighted (6 r + 3 w)"}


def paste_geometry(dr3, synth, TW_, TH_):
    """the 8 forward homographies (already multiplied by top_left, src/panorama.cpp:148) of the 4 x 2 tile layout and the canvas size"""
    Ms = []
    for i in range(8):
        M = synth.tile_homography(i, TW_, TH_).astype(np.float64)
        T = np.eye(3); T[0, 2] = (i % 4) * TW_ * 0.9 + 100; T[1, 2] = (i // 4) * TH_ * 0.9 + 150
        Ms.append((T @ M).astype(np.float32))
    boxes = [dr3.panorama_bbox(M, TW_, TH_) for M in Ms]
    cw = (max(b[2] for b in boxes) + 64) // 64 * 64
    ch = max(b[3] for b in boxes) + 16
    return Ms, boxes, cw, ch


def paste_sharded_leg(dist, ctx, dr3, synth, sync, tile, steps, feather=300):
    """SURVEY 8(e), the warp side, as DESIGN section 6 states it: ONE panorama per step across the whole job (reference
    Panorama::paste_images, src/panorama.cpp:144-154).  The canvas rows are cut into `world` slabs; every rank pastes all 8 tiles
    into ITS slab (dr3hip_panorama_add_slab: the geometry stays that of the whole canvas) and normalises it -- no exchange between
    ranks, no reduction.  Time = MAX over ranks of the barrier-bracketed region; the rate is the JOB's (tile pixels of one panorama
    per step): strong scaling.  Check: every rank hashes its slab of the finished 8UC3 panorama, rank 0 pastes the whole canvas alone
    and compares its rows with the gathered hashes."""
    import hashlib
    TH_, TW_ = tile.shape[:2]
    Ms, boxes, cw, ch = paste_geometry(dr3, synth, TW_, TH_)
    y0, y1 = ch * dist.rank // dist.world, ch * (dist.rank + 1) // dist.world
    rows = y1 - y0
    d_img = ctx.to_device(tile)
    d_slab, d_out = ctx.alloc(cw * rows * 16), ctx.alloc(cw * rows * 3)

    def step():
        d_slab.zero()
        for M in Ms:
            ctx.panorama_add_slab(d_img, TW_, TH_, TW_ * 3, d_slab, cw, ch, cw * 16, M, feather, y0, y1)
        ctx.panorama_normalize(d_slab, cw, rows, cw * 16, d_out, cw * 3)

    step(); sync()
    el = timed_region(dist, step, sync, steps, 0)
    mine = hashlib.sha256(np.ascontiguousarray(d_out.download((rows, cw, 3), np.uint8)).tobytes()).digest()
    gathered = dist.all_gather_bytes(mine)
    ok = None
    if dist.rank == 0:
        d_c, d_o = ctx.alloc(cw * ch * 16), ctx.alloc(cw * ch * 3)
        d_c.zero()
        for M in Ms:
            ctx.panorama_add_slab(d_img, TW_, TH_, TW_ * 3, d_c, cw, ch, cw * 16, M, feather, 0, ch)
        ctx.panorama_normalize(d_c, cw, ch, cw * 16, d_o, cw * 3)
        sync()
        whole = d_o.download((ch, cw, 3), np.uint8)
        ok = all(hashlib.sha256(np.ascontiguousarray(whole[ch * r // dist.world:ch * (r + 1) // dist.world]).tobytes()).digest() == gathered[r]
                 for r in range(dist.world))
        d_c.free(); d_o.free()
    for d in (d_img, d_slab, d_out):
        d.free()
    return {"canvas": [cw, ch], "tiles": 8, "slabs": dist.world, "rows_per_slab": [ch * (r + 1) // dist.world - ch * r // dist.world for r in range(dist.world)],
            "scaling": "strong (one panorama per step across the job; canvas row slabs, no reduction)",
            "ms_per_panorama": el / steps * 1e3, "tile_mpix_per_s_job": 8 * TW_ * TH_ * steps / 1e6 / el,
            "slabs_equal_the_single_gpu_canvas": ok}


def paste_leg(dist, ctx, dr3, synth, sync, tile, steps):
    """configs[2] as the reference really runs it (Panorama::paste_images, src/panorama.cpp:143-154): the 8 tiles
    are pasted into ONE 32FC4 weighted canvas (4 x 2 layout, ~10 % overlap, feathering width 300) by the fused
    warp + feather + accumulate kernel, then normalised to 8UC3.  One step = 8 pastes + 1 normalise."""
    TH_, TW_ = tile.shape[:2]
    Ms, boxes, cw, ch = paste_geometry(dr3, synth, TW_, TH_)
    d_img = ctx.to_device(tile)
    d_canvas = ctx.alloc(cw * ch * 16)
    d_out = ctx.alloc(cw * ch * 3)
    plans = [dr3.panorama_plan(TW_, TH_, (cw, ch), M, istride=TW_ * 3) for M in Ms]      # (kernel, coordinate form, box) per paste
    box_px = sum((p[2][2] - p[2][0]) * (p[2][3] - p[2][1]) for p in plans)                # canvas pixels the eight launches evaluate
    add_kernel = "paste_strip_kernel" if all(p[0] == 2 for p in plans) else "panorama_add_kernel"

    def step():
        d_canvas.zero()
        for M in Ms:
            ctx.panorama_add(d_img, TW_, TH_, TW_ * 3, d_canvas, cw, ch, cw * 16, M, 300)
        ctx.panorama_normalize(d_canvas, cw, ch, cw * 16, d_out, cw * 3)

    ctx.profile_enable(True)
    step(); sync()
    ctx.profile_reset()
    el = timed_region(dist, step, sync, steps, 0)
    ms_add, n_add = ctx.profile_read(dr3.PROF_PANO_ADD)
    ms_norm, n_norm = ctx.profile_read(dr3.PROF_PANO_NORM)
    ctx.profile_enable(False)
    ok = None
    if dist.rank == 0:
        # tile 0 pasted (same kernels, same homography, same feathering) into a canvas just large enough for it, against the
        # oracle's three passes (32FC4 copy, warp onto the canvas, feather + accumulate) and its normalise; untimed
        from oracle import oracle as O
        vw, vh = (boxes[0][2] + 64) // 64 * 64, boxes[0][3] + 16
        d_vc, d_vo = ctx.alloc(vw * vh * 16), ctx.alloc(vw * vh * 3)
        d_vc.zero()
        ctx.panorama_add(d_img, TW_, TH_, TW_ * 3, d_vc, vw, vh, vw * 16, Ms[0], 300)
        ctx.panorama_normalize(d_vc, vw, vh, vw * 16, d_vo, vw * 3)
        want = np.zeros((vh, vw, 4), np.float32)
        O.panorama_add(tile, want, Ms[0], 300)
        ok = bool(np.array_equal(d_vc.download((vh, vw, 4), np.float32).view(np.uint32), want.view(np.uint32)) and
                  np.array_equal(d_vo.download((vh, vw, 3), np.uint8), O.normalize_canvas(want)))
        d_vc.free(); d_vo.free()
    for d in (d_img, d_canvas, d_out):
        d.free()
    add_ms = ms_add / max(n_add, 1)                      # per paste
    # algorithmic bytes of one paste: every canvas pixel of the box is warped (3 B of source per covered pixel)
    # and, where covered by the tile, read + written (16 + 16 B); covered ~ tile area
    alg_add = TW_ * TH_ * (3 + 32)
    alg_norm = cw * ch * (16 + 3)
    norm_ms = ms_norm / max(n_norm, 1)
    return {"canvas": [cw, ch], "tiles": 8, "feathering_width": 300, "verified_vs_oracle": ok,
            "tile_mpix_per_s": aggregate_rate(dist, 8 * TW_ * TH_ * steps / 1e6, el),
            "step_ms": el / steps * 1e3,
            "add_kernel_ms": add_ms, "box_pixels_per_step": box_px,
            "kernel_plan": {"strips": sum(1 for p in plans if p[0] == 2), "direct": sum(1 for p in plans if p[0] != 2),
                            "coordinate_forms": sorted(set(p[1] for p in plans)),
                            "note": "dr3hip_panorama_plan per paste: strips = paste_strip_kernel (the image's box through the LDS-staged strip "
                                    "machinery), direct = panorama_add_kernel; the plan is made over the box, not the whole canvas"},
            "roofline_add": {"kernel": add_kernel, "bound": "hbm", "achieved": alg_add / (add_ms * 1e-3) / 1e9,
                             "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": alg_add / (add_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                             "traffic": pmc_traffic(add_kernel, 256, "pmc_traffic_warp.json")[0],
                             "algorithmic_bytes_per_launch": alg_add},
            "roofline_normalize": {"kernel": "panorama_normalize_kernel", "bound": "hbm", "achieved": alg_norm / (norm_ms * 1e-3) / 1e9,
                                   "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": alg_norm / (norm_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                   "traffic": pmc_traffic("panorama_normalize_kernel", 256, "pmc_traffic_warp.json")[0],
                                   "algorithmic_bytes_per_launch": alg_norm, "kernel_ms": norm_ms},
            "replaces": "reference passes per tile: 32FC4 copy (3 r + 16 w), warp onto the full canvas (16 r + 16 w per canvas px), "
                        "feather + accumulate (32 r + 32 w per box px)"}


def host_cores():
    """usable host cores: the affinity mask capped by the cgroup CPU quota (a container may see 256
    CPUs in its mask while being allowed 16 cores' worth of time; oversubscribing OpenMP on that
    makes the baseline meaninglessly slow)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(frames, pts, seconds, with_warp, synth):
    """the oracle (a port of the reference's OpenCV calls) on all host cores, bounded sample.
    Mirrors the reference's call shape: one calcOpticalFlowPyrLK(prev, next) per pair, which
    rebuilds both Gaussian pyramids and the derivative pyramid each call (src/initialization.cpp:608)."""
    from oracle import oracle as O
    cores = host_cores()
    po = O.lk_params(win=(WIN, WIN), max_level=MAX_LEVEL, max_iters=MAX_ITERS, eps=EPS, flags=0)
    # -O3 -march=native.  Built here, on the host being timed; used only if it reproduces the portable
    # build's results bit for bit (same strict float flags, so it does)
    ref = O.lk_track(frames[0], frames[1], pts, None, po, n_threads=cores)
    build = "-O2 (portable build)"
    native = O.build_native()
    if native:
        O.use_library(native)
        try:
            chk = O.lk_track(frames[0], frames[1], pts, None, po, n_threads=cores)
            if np.array_equal(chk[0].view(np.uint32), ref[0].view(np.uint32)) and np.array_equal(chk[1], ref[1]):
                build = "-O3 -march=native, OpenMP, built on this host"
            else:
                O.use_library(None)
        except OSError:
            O.use_library(None)
    # two organisations of the same computation, identical results (tests/test_oracle_opt.py): the readable port (per-tap index
    # reflection, 25-tap pyrDown, OpenCV's scalar summation order) and "port-opt" (separable pyrDown, levels padded once, the window loops
    # on the SSE2 intrinsic sequences of OpenCV's x86-64 build, its summation order).  `value` is the FASTER one: the >= 50x target is
    # quoted against the less flattering denominator.
    def rate(opt, order, threads, budget, cap):
        O.lk_track(frames[0], frames[1], pts, None, po, order=order, n_threads=threads, opt=opt)          # warm-up
        t0 = time.perf_counter()
        k = 0
        while True:
            i = k % (len(frames) - 1)
            O.lk_track(frames[i], frames[i + 1], pts, None, po, order=order, n_threads=threads, opt=opt)
            k += 1
            if time.perf_counter() - t0 > budget or k >= cap:
                break
        el_ = time.perf_counter() - t0
        return k * len(pts) / el_, k, el_

    same = O.lk_track(frames[0], frames[1], pts, None, po, order=O.ORDER_SSE2, n_threads=cores, opt=True)
    plain = O.lk_track(frames[0], frames[1], pts, None, po, order=O.ORDER_SSE2, n_threads=cores)
    opt_ok = bool(np.array_equal(same[0].view(np.uint32), plain[0].view(np.uint32)) and np.array_equal(same[1], plain[1]) and
                  np.array_equal(same[2].view(np.uint32), plain[2].view(np.uint32)))
    v_opt, n_pairs, el = rate(True, O.ORDER_SSE2, cores, seconds * 0.45, 8192)
    v_opt1, n1, el1 = rate(True, O.ORDER_SSE2, 1, max(2.0, seconds * 0.15), 2048)
    v_port, n_p, el_p = rate(False, O.ORDER_RASTER, cores, seconds * 0.25, 4096)
    v_port1, n_p1, el_p1 = rate(False, O.ORDER_RASTER, 1, max(2.0, seconds * 0.15), 1024)
    out = {"value": v_opt, "unit": "keypoints/s", "cores": cores, "kind": "port",
           "variant": "port-opt: dr3o_lk_track_opt (separable pyrDown, padded levels, SSE2 window loops in OpenCV's x86-64 summation order)",
           "port_opt_equals_port_bit_for_bit": opt_ok,
           "single_thread_value": v_opt1, "single_thread_sample": "%d frame pairs, %.1f s" % (n1, el1),
           "readable_port": {"value": v_port, "cores": cores, "single_thread_value": v_port1,
                             "sample": "%d frame pairs, %.1f s (all cores); %d, %.1f s (one thread)" % (n_p, el_p, n_p1, el_p1),
                             "note": "dr3o_lk_track_ex: the checker's readable form (index reflection per tap, 25-tap pyrDown, scalar order)"},
           "cpu_model": cpu_model(), "build": build,
           "sample": "%d frame pairs x %d points of the same stream, one call per pair (both Gaussian pyramids + Scharr + tracker, as "
                     "src/initialization.cpp:608 does), OpenMP over rows / points, %.1f s" % (n_pairs, len(pts), el)}
    try:                                                 # the real thing, where a cv2 module happens to exist (it does not in this image)
        import cv2
        crit = (cv2.TERM_CRITERIA_COUNT + cv2.TERM_CRITERIA_EPS, MAX_ITERS, EPS)
        p0 = pts.reshape(-1, 1, 2).astype(np.float32)
        t2 = time.perf_counter(); n2 = 0
        while time.perf_counter() - t2 < max(2.0, seconds / 4) and n2 < 1024:
            i = n2 % (len(frames) - 1)
            cv2.calcOpticalFlowPyrLK(frames[i], frames[i + 1], p0, None, winSize=(WIN, WIN), maxLevel=MAX_LEVEL, criteria=crit)
            n2 += 1
        out["opencv"] = {"version": cv2.__version__, "keypoints_per_sec": n2 * len(pts) / (time.perf_counter() - t2),
                         "threads": cv2.getNumThreads(), "note": "cv2.calcOpticalFlowPyrLK on the same pairs"}
    except ImportError:
        out["opencv"] = None
    if with_warp:
        tile = cheap_tile(synth, 4000, 3000)
        M = synth.tile_homography(0, 4000, 3000)
        O.warp_perspective(tile, M, (4000, 3000), n_threads=cores)      # warm-up (page faults, threads)
        t0 = time.perf_counter()
        n_t = 0
        while n_t < 64:
            O.warp_perspective(tile, M, (4000, 3000), n_threads=cores)
            n_t += 1
            if time.perf_counter() - t0 > max(3.0, seconds / 4):
                break
        elw = time.perf_counter() - t0
        out["warp_8uc3_mpix_per_s"] = 12.0 * n_t / elw
        out["warp_sample"] = "%d x one 4000x3000 8UC3 tile, oracle dr3o_warp_perspective, %.2f s" % (n_t, elw)
    O.use_library(None)
    return out


if __name__ == "__main__":
    sys.exit(main())
