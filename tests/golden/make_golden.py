#!/usr/bin/env python3
"""Generates tests/golden/hotpath_small.npz.

SELF-GOLDEN vectors: the reference repository holds no fixtures for this path and cannot be run
here (OpenCV absent), so these are inputs + outputs of the CPU oracle (oracle/dr3_oracle.c), frozen
so that (a) later edits of the oracle cannot silently change its answers and (b) the HIP path can be
checked on the GPU box, where neither /root/reference nor a rebuild of history exists.
Run from the repo root:  python tests/golden/make_golden.py
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

synth = importlib.import_module("3dr_amd.synth")


def main():
    out = {}
    A, B, T = synth.frame_pair(160, 96)
    pts = synth.grid_points(160, 96, 16, border=4)
    pts = np.vstack([pts, [[1.5, 2.25], [158.5, 94.0], [-20, 10], [80, 300]]]).astype(np.float32)
    out["lk_prev"], out["lk_next"], out["lk_pts"] = A, B, pts
    out["lk_T"] = T
    for tag, kw in {"a": dict(win=(15, 15), max_level=3, max_iters=30, eps=0.01, flags=0),
                    "b": dict(win=(9, 7), max_level=2, max_iters=10, eps=0.03, flags=O.LK_USE_INITIAL_FLOW),
                    "c": dict(win=(21, 21), max_level=4, max_iters=30, eps=0.01, flags=O.LK_GET_MIN_EIGENVALS)}.items():
        p = O.lk_params(**kw)
        guess = (pts + np.float32(0.5)).astype(np.float32)
        out["lk_%s_params" % tag] = np.array([kw["win"][0], kw["win"][1], kw["max_level"], kw["max_iters"], kw["flags"]], np.int32)
        out["lk_%s_eps" % tag] = np.float64(kw["eps"])
        out["lk_%s_guess" % tag] = guess
        for oname, order in (("raster", O.ORDER_RASTER), ("wave64", O.ORDER_WAVE64)):
            xy, st, err, L = O.lk_track(A, B, pts, guess, p, order=order)
            out["lk_%s_%s_xy" % (tag, oname)], out["lk_%s_%s_st" % (tag, oname)] = xy, st
            out["lk_%s_%s_err" % (tag, oname)], out["lk_%s_%s_L" % (tag, oname)] = err, np.int32(L)
    out["pyr1"] = O.pyr_down(A)
    out["pyr2"] = O.pyr_down(out["pyr1"])
    out["scharr0"] = O.scharr(A)
    out["scharr1"] = O.scharr(out["pyr1"])

    rng = np.random.default_rng(synth.SEED)
    src3 = synth.panorama_tile(96, 64)
    src4 = np.ones((64, 96, 4), np.float32); src4[..., :3] = src3
    M = np.array([[0.98, -0.06, 7.25], [0.05, 1.03, -3.5], [2e-4, -1e-4, 1]], np.float32)
    out["warp_src3"], out["warp_M"] = src3, M
    out["warp_8uc3"] = O.warp_perspective(src3, M, (120, 80))
    out["warp_32fc4"] = O.warp_perspective(src4, M, (120, 80))
    out["warp_8uc3_inv"] = O.warp_perspective(src3, M, (120, 80), flags=O.WARP_INVERSE_MAP)
    mx = rng.uniform(-2, 98, (50, 70)).astype(np.float32); my = rng.uniform(-2, 66, (50, 70)).astype(np.float32)
    out["remap_mx"], out["remap_my"] = mx, my
    out["remap_8uc3"] = O.remap(src3, mx, my)
    img = rng.integers(0, 256, (48, 61), dtype=np.uint8)
    img16 = rng.integers(0, 256, (48, 64), dtype=np.uint8)
    out["half_src_odd"], out["half_src_16"] = img, img16
    for name, mode in (("trunc", O.HALF_TRUNC), ("sse2", O.HALF_SSE2), ("refscalar", O.HALF_REF_SCALAR), ("refx86", O.HALF_REF_X86)):
        out["half_odd_" + name] = O.halfsample(img, mode)
        out["half_16_" + name] = O.halfsample(img16, mode)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hotpath_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
