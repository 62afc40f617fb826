"""dr3o_lk_track_opt ("port-opt": separable pyrDown, padded levels, SSE2 window loops -- the organisation of an optimised CPU
library, timed by bench.py's cpu_baseline) returns what the readable port returns, bit for bit, in every summation order."""
import numpy as np
import pytest

from oracle import oracle as O

ORDERS = (O.ORDER_RASTER, O.ORDER_WAVE64, O.ORDER_SSE2)


def _same(a, b):
    assert a[3] == b[3]
    assert np.array_equal(a[1], b[1])
    assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)), np.nanmax(np.abs(a[0] - b[0]))
    assert np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32))
    assert np.array_equal(a[4], b[4])


@pytest.mark.parametrize("order", ORDERS)
@pytest.mark.parametrize("case", ["kitti15", "cfg0_21", "literals30", "small_odd"])
def test_opt_equals_plain(synth, order, case):
    if case == "small_odd":
        A, B, _ = synth.frame_pair(161, 97)
        pts = synth.grid_points(161, 97, 11, border=0)
        kw = dict(win=(9, 13), max_level=3, max_iters=20, eps=0.03)
    else:
        A, B, _ = synth.frame_pair()
        if case == "kitti15":
            pts, kw = synth.grid_points(synth.KITTI_W, synth.KITTI_H, 15, n=2000)[::4], dict(win=(15, 15), max_level=3, max_iters=30, eps=0.01)
        elif case == "cfg0_21":
            pts, kw = synth.grid_points(synth.KITTI_W, synth.KITTI_H, 30), dict(win=(21, 21), max_level=2, max_iters=30, eps=0.01)
        else:
            pts, kw = synth.grid_points(synth.KITTI_W, synth.KITTI_H, 30), dict(win=(30, 30), max_level=4, max_iters=1000, eps=1e-3, flags=O.LK_USE_INITIAL_FLOW)
    p = O.lk_params(**kw)
    guess = pts.copy() + 0.37 if kw.get("flags", 0) & 4 else None
    _same(O.lk_track(A, B, pts, guess, p, order=order, want_iters=True), O.lk_track(A, B, pts, guess, p, order=order, want_iters=True, opt=True))


@pytest.mark.parametrize("order", ORDERS)
def test_opt_equals_plain_on_borders_and_hostile_points(synth, order):
    """windows hanging over every edge (the padded planes stand in for the index reflection), points outside, non-finite points,
    both flags"""
    A, B, _ = synth.frame_pair(200, 120)
    rng = np.random.default_rng(5)
    pts = np.concatenate([rng.uniform(-20, 220, (300, 1)), rng.uniform(-20, 140, (300, 1))], 1).astype(np.float32)
    pts[:8] = [[0, 0], [199, 119], [-7, 3], [3, -7], [206.5, 60], [100, 126.5], [np.nan, 5], [5, np.inf]]
    for win, flags in (((7, 7), 0), ((15, 15), O.LK_GET_MIN_EIGENVALS), ((21, 9), O.LK_USE_INITIAL_FLOW), ((32, 32), 0)):
        p = O.lk_params(win=win, max_level=2, max_iters=12, eps=0.02, flags=flags)
        guess = pts + rng.uniform(-1.5, 1.5, pts.shape).astype(np.float32) if flags & O.LK_USE_INITIAL_FLOW else None
        _same(O.lk_track(A, B, pts, guess, p, order=order, want_iters=True), O.lk_track(A, B, pts, guess, p, order=order, want_iters=True, opt=True))


def test_opt_is_faster():
    import importlib
    import time
    synth = importlib.import_module("3dr_amd.synth")
    A, B, _ = synth.frame_pair()
    pts = synth.grid_points(synth.KITTI_W, synth.KITTI_H, 15, n=2000)
    p = O.lk_params(win=(15, 15), max_level=3, max_iters=30, eps=0.01)
    t = []
    for opt in (False, True):
        O.lk_track(A, B, pts, None, p, order=O.ORDER_SSE2, n_threads=1, opt=opt)
        t0 = time.perf_counter()
        O.lk_track(A, B, pts, None, p, order=O.ORDER_SSE2, n_threads=1, opt=opt)
        t.append(time.perf_counter() - t0)
    assert t[1] < t[0], t
