"""DR3O_ORDER_SSE2: the summation order an x86-64 build of OpenCV 3.x executes behind cv::calcOpticalFlowPyrLK (reference
src/initialization.cpp:608-613 -> LKTrackerInvoker's CV_SSE2 blocks).  The oracle keeps that order by plain-C bookkeeping (lane
indices worked out by hand, oracle/dr3_oracle.c wsum_add); oracle/sse2_replay.c runs the same accumulation sequences with the real
<emmintrin.h> intrinsics.  The two must agree bit for bit on every window shape."""
import numpy as np
import pytest

from oracle import oracle as O


def _random_window(rng, w, h, big):
    # gradients after CV_DESCALE: |Ix|, |Iy| <= 4080 (16 * 255); mismatches |d| <= 8160 (32 * 255)
    g = 4080 if big else 300
    dI = rng.integers(-g, g + 1, (h, w, 2)).astype(np.int16)
    diff = rng.integers(-8160 if big else -400, (8160 if big else 400) + 1, (h, w)).astype(np.int16)
    return dI, diff


@pytest.mark.parametrize("win", [(15, 15), (21, 21), (30, 30), (3, 3), (4, 4), (7, 5), (8, 8), (9, 31), (12, 16), (16, 9), (23, 17), (32, 32), (31, 3)])
def test_sse2_bookkeeping_equals_the_intrinsics(win):
    R = O.sse2_replay_lib()
    if R is None:
        pytest.skip("not an x86-64 host")
    rng = np.random.default_rng(hash(win) & 0xffff)
    w, h = win
    for rep in range(40):
        dI, diff = _random_window(rng, w, h, big=rep % 2 == 0)
        A, b = O.window_sums(dI, diff, O.ORDER_SSE2)
        rA, rb = np.zeros(3, np.float32), np.zeros(2, np.float32)
        R.sse2_replay_A(O._p(dI), w, h, O._p(rA))
        R.sse2_replay_B(O._p(diff), O._p(dI), w, h, O._p(rb))
        assert np.array_equal(A.view(np.uint32), rA.view(np.uint32)), (win, rep, A, rA)
        assert np.array_equal(b.view(np.uint32), rb.view(np.uint32)), (win, rep, b, rb)


def test_orders_differ_only_in_the_last_bits():
    """the three orders are three roundings of the same exact sums"""
    rng = np.random.default_rng(7)
    dI, diff = _random_window(rng, 21, 21, big=True)
    exactA = [float((dI[..., 0].astype(np.int64) ** 2).sum()), float((dI[..., 0].astype(np.int64) * dI[..., 1]).sum()), float((dI[..., 1].astype(np.int64) ** 2).sum())]
    scale = float((dI.astype(np.int64) ** 2).sum())
    seen = set()
    for order in (O.ORDER_RASTER, O.ORDER_WAVE64, O.ORDER_SSE2):
        A, b = O.window_sums(dI, diff, order)
        assert np.allclose(A, exactA, rtol=0, atol=scale * 21 * 21 * 2.0 ** -24)
        seen.add(A.tobytes() + b.tobytes())
    assert len(seen) >= 2          # random full-range terms: at least one order rounds differently


def test_sse2_order_tracks_like_the_other_orders():
    """whole tracker in the three orders on a synthetic pair: same status, positions within a few 1e-3 px of each other"""
    import importlib
    synth = importlib.import_module("3dr_amd.synth")
    A, B, _ = synth.frame_pair(320, 200)
    pts = synth.grid_points(320, 200, 24)
    p = O.lk_params(win=(15, 15), max_level=3, max_iters=30, eps=0.01)
    res = {o: O.lk_track(A, B, pts, None, p, order=o, want_iters=True) for o in (O.ORDER_RASTER, O.ORDER_WAVE64, O.ORDER_SSE2)}
    for o in (O.ORDER_WAVE64, O.ORDER_SSE2):
        agr = O.order_agreement(res[o], res[O.ORDER_RASTER], p, 1e-3)
        assert agr["status_equal"] and agr["max_dev_px"] < 0.05
    # the residual is a scalar raster sum of exact integers in every build
    assert np.array_equal(res[O.ORDER_SSE2][2], res[O.ORDER_RASTER][2]) or np.allclose(res[O.ORDER_SSE2][2], res[O.ORDER_RASTER][2], atol=1e-2)
