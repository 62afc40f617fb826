"""Three-way agreement of the tracker's f32 summation orders on pair 0 of every BASELINE configuration's stream (CPU only: the
kernel's order is the oracle's DR3O_ORDER_WAVE64, which the GPU tests and every bench leg prove equal to the kernel bit for bit).
  kernel : lane-exact integer partial sums + the lane tree of 3dr_amd/csrc/lk_kernels.hip
  raster : OpenCV's scalar path (a build without SIMD)
  sse2   : what an x86-64 OpenCV 3.x build executes behind src/initialization.cpp:608-613 (LKTrackerInvoker's CV_SSE2 blocks)
usage: python tools/order_agreement.py [out.json]   (writes profiles/r04_v1/order_agreement.json by default)"""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                  # noqa: E402  (stream_frames: the bench's own inputs)
from oracle import oracle as O                # noqa: E402

synth = importlib.import_module("3dr_amd.synth")


def configs():
    """(name, frame A, frame B, points, initial guess or None, lk params)"""
    k_frames, _ = bench.stream_frames(synth, 2, 1241, 376, period=9)
    hd_frames, _ = bench.stream_frames(synth, 2, 1920, 1080, period=3)
    p30 = synth.grid_points(1241, 376, 30)
    yield ("configs[1]: 1241x376, 2000 points, 15x15, 4 levels, eps 0.01", k_frames[0], k_frames[1], synth.grid_points(1241, 376, 15, n=2000), None,
           dict(win=(15, 15), max_level=3, max_iters=30, eps=0.01, flags=0))
    yield ("configs[0]: 1241x376, 546 points, 21x21, 3 levels, eps 0.01", k_frames[0], k_frames[1], p30, None,
           dict(win=(21, 21), max_level=2, max_iters=30, eps=0.01, flags=0))
    yield ("reference literals: 30x30, maxLevel 4, 1000 -> 100 iterations, eps 1e-3, USE_INITIAL_FLOW", k_frames[0], k_frames[1], p30, p30.copy(),
           dict(win=(30, 30), max_level=4, max_iters=1000, eps=1e-3, flags=O.LK_USE_INITIAL_FLOW))
    yield ("configs[4]: 1920x1080, 5000 points, 21x21, 5 levels, eps 0.01", hd_frames[0], hd_frames[1], synth.grid_points(1920, 1080, 20, n=5000), None,
           dict(win=(21, 21), max_level=4, max_iters=30, eps=0.01, flags=0))
    try:
        z = np.load(os.path.join(ROOT, "tests", "golden", "kitti_pair.npz"))
        imgs = [z[k] for k in z.files if z[k].ndim == 2][:2]
        ref = O.halfsample_pyramid(imgs[0], 3, O.HALF_REF_X86)
        x, y, s, l = O.fast_detect_grid(ref, 30, 20, 20.0)
        pts = np.stack([x[l >= 0], y[l >= 0]], 1).astype(np.float32)
        yield ("the reference's own pair imgs/kitti0.png -> kitti5.png, FAST corners, reference literals (30x30)", imgs[0], imgs[1], pts, pts.copy(),
               dict(win=(30, 30), max_level=4, max_iters=1000, eps=1e-3, flags=O.LK_USE_INITIAL_FLOW))
    except (OSError, KeyError, IndexError):
        pass


def three_way(A, B, pts, guess, kw, tol=1e-3):
    p = O.lk_params(**kw)
    res = {o: O.lk_track(A, B, pts, None if guess is None else guess.copy(), p, order=o, want_iters=True)
           for o in (O.ORDER_WAVE64, O.ORDER_RASTER, O.ORDER_SSE2)}
    out = {}
    for a, b in ((O.ORDER_WAVE64, O.ORDER_RASTER), (O.ORDER_WAVE64, O.ORDER_SSE2), (O.ORDER_RASTER, O.ORDER_SSE2)):
        out["%s_vs_%s" % (O.ORDER_NAMES[a], O.ORDER_NAMES[b])] = O.order_agreement(res[a], res[b], p, tol)
    return out


def main():
    dst = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r04_v1", "order_agreement.json")
    table = {}
    for name, A, B, pts, guess, kw in configs():
        table[name] = three_way(A, B, pts, guess, kw)
        for k, v in table[name].items():
            print("%-34s %-22s tracked %5d  same-path %5d (max %.2e)  flips %3d (max %.2e)  clamp %3d (max %.2e)  beyond 1e-3: %d  max %.2e"
                  % (name[:34], k, v["tracked_both"], v["same_path"], v["same_path_max_dev_px"], v["exit_flipped"], v["exit_flipped_max_dev_px"],
                     v["clamped"], v["clamped_max_dev_px"], v["beyond_tol"], v["max_dev_px"]), flush=True)
    json.dump({"tol_px": 1e-3, "note": "pair 0 of each configuration's synthetic stream (bench.stream_frames) and the reference's own KITTI pair; "
               "kernel = DR3O_ORDER_WAVE64 (equal to the HIP kernel bit for bit), raster = OpenCV scalar, sse2 = OpenCV 3.x x86-64 CV_SSE2",
               "configs": table}, open(dst, "w"), indent=1)
    print("wrote", dst)


if __name__ == "__main__":
    main()
