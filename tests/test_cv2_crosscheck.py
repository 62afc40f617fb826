"""Optional pin of the oracle against a real OpenCV ("skip-if-missing cv2 cross-check").  Neither this image nor
the GPU box ships cv2, so these tests are skipped there; wherever an OpenCV 3.x/4.x Python module exists they compare the
oracle's restatement of the OpenCV calls the reference makes with OpenCV itself, on the reference's own KITTI pair."""
import os

import numpy as np
import pytest

cv2 = pytest.importorskip("cv2")

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kitti_pair.npz")


@pytest.fixture(scope="module")
def kitti():
    z = np.load(FIX)
    return np.ascontiguousarray(z["kitti0"]), np.ascontiguousarray(z["kitti5"])


def test_pyrdown_and_scharr_match_opencv(O, kitti):
    A, _ = kitti
    assert np.array_equal(O.pyr_down(A), cv2.pyrDown(A))
    d = O.scharr(A).astype(np.int32)
    gx = cv2.Scharr(A, cv2.CV_16S, 1, 0).astype(np.int32); gy = cv2.Scharr(A, cv2.CV_16S, 0, 1).astype(np.int32)
    assert np.array_equal(d[..., 0], gx) and np.array_equal(d[..., 1], gy)


def test_lk_matches_opencv_within_tolerance(O, kitti):
    A, B = kitti
    pyr = O.halfsample_pyramid(A, 3, O.HALF_REF_X86)
    x, y, sc, lv = O.fast_detect_grid(pyr, 30, 20, 20.0)
    pts = np.stack([x[lv >= 0], y[lv >= 0]], 1).astype(np.float32)
    crit = (cv2.TERM_CRITERIA_COUNT + cv2.TERM_CRITERIA_EPS, 30, 0.01)
    c_xy, c_st, c_err = cv2.calcOpticalFlowPyrLK(A, B, pts.reshape(-1, 1, 2), None, winSize=(21, 21), maxLevel=2, criteria=crit)
    o_xy, o_st, o_err, _ = O.lk_track(A, B, pts, None, O.lk_params(win=(21, 21), max_level=2, max_iters=30, eps=0.01), order=O.ORDER_RASTER)
    c_xy = c_xy.reshape(-1, 2); c_st = c_st.reshape(-1)
    assert (c_st == o_st).mean() > 0.99
    both = (c_st == 1) & (o_st == 1)
    d = np.abs(c_xy - o_xy).max(1)[both]
    assert (d <= 1e-3).mean() > 0.97          # OpenCV's SIMD summation order differs from the scalar order restated here
    # the oracle's x86-64 order (LKTrackerInvoker's CV_SSE2 blocks of OpenCV 3.0 - 3.4): an OpenCV of that line on x86-64 should
    # agree bit for bit (later lines use universal intrinsics of other widths): the tolerance is asserted, the bit-identical share printed
    s_xy, s_st, _, _ = O.lk_track(A, B, pts, None, O.lk_params(win=(21, 21), max_level=2, max_iters=30, eps=0.01), order=O.ORDER_SSE2)
    both_s = (c_st == 1) & (s_st == 1)
    ds = np.abs(c_xy - s_xy).max(1)[both_s]
    print("cv2 %s vs oracle SSE2 order: %d / %d tracked points bit-identical, max deviation %.3g px" % (cv2.__version__, int((ds == 0).sum()), int(both_s.sum()), float(ds.max())))
    assert (ds <= 1e-3).mean() >= (d <= 1e-3).mean() - 0.01


def test_warp_perspective_matches_opencv(O, kitti, synth):
    A, B = kitti
    bgr = np.stack([A, B, A], -1)
    M = synth.tile_homography(3, 1240, 376)
    want = cv2.warpPerspective(bgr, M, (1400, 480), flags=cv2.INTER_LINEAR)
    assert np.array_equal(O.warp_perspective(bgr, M, (1400, 480)), want)
    f = bgr.astype(np.float32)
    f4 = np.dstack([f, np.ones(f.shape[:2], np.float32)])
    got = O.warp_perspective(f4, M, (1400, 480)); ref = cv2.warpPerspective(f4, M, (1400, 480), flags=cv2.INTER_LINEAR)
    assert np.abs(got - ref).max() <= np.spacing(np.float32(255.0))          # 1 ULP (north_star)


def test_remap_matches_opencv(O, kitti):
    """cv::remap as utils::warp_local calls it (reference src/utils.cpp:189-194): f32 maps, INTER_LINEAR, BORDER_CONSTANT 0 -- the
    spherical maps of utils::compute_spherical_warping and a map with values outside the image, on ties of the 1/32-px grid and
    beyond the int16 range of the fixed-point coordinates"""
    A, B = kitti
    bgr = np.stack([A, B, A], -1)
    h, w = A.shape
    u, v = O.spherical_map(w, h, 0.9 * w)
    want = cv2.remap(bgr, u, v, cv2.INTER_LINEAR, borderMode=cv2.BORDER_CONSTANT, borderValue=0)
    assert np.array_equal(O.remap(bgr, u, v), want)
    rng = np.random.default_rng(11)
    u2 = (rng.uniform(-40, w + 40, (h, w)).astype(np.float32) * 64).round() / 64          # every value on a 1/64-px tie or grid point
    v2 = (rng.uniform(-40, h + 40, (h, w)).astype(np.float32) * 64).round() / 64
    u2[::7, ::5] = 40000.5; v2[::9, ::11] = -40000.25
    want = cv2.remap(bgr, u2, v2, cv2.INTER_LINEAR, borderMode=cv2.BORDER_CONSTANT, borderValue=0)
    assert np.array_equal(O.remap(bgr, u2, v2), want)
    f = bgr.astype(np.float32)
    got = O.remap(f, u, v); ref = cv2.remap(f, u, v, cv2.INTER_LINEAR, borderMode=cv2.BORDER_CONSTANT, borderValue=0)
    assert np.abs(got - ref).max() <= np.spacing(np.float32(255.0))
