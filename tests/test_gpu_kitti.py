"""BASELINE configs[0] on the reference's own input data: the two KITTI frames its two-view test loads
(reference tests/test_init.cpp:24-25; packed by tests/golden/make_kitti_fixture.py).  The reference's flow
(Init::process_first_frame -> process_second_frame, src/initialization.cpp:552-613): FAST corners on the first frame's
half-sample pyramid, then calcOpticalFlowPyrLK into the second frame with the literals of :593-613.  GPU vs oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kitti_pair.npz")


@pytest.fixture(scope="module")
def kitti():
    z = np.load(FIX)
    return np.ascontiguousarray(z["kitti0"]), np.ascontiguousarray(z["kitti5"])


def test_detect_then_track_with_reference_literals(dr3, O, ctx, kitti):
    A, B = kitti
    assert A.shape == B.shape == (376, 1240) and A.dtype == np.uint8
    # Frame::Frame: create_img_pyramid(img, 3) ; FastDetector(w, h, cell 30, 3 levels).detect(.., 20.0)
    pyr = O.halfsample_pyramid(A, 3, O.HALF_REF_X86)
    assert all(np.array_equal(a, b) for a, b in zip(ctx.create_img_pyramid(A, 3), pyr))
    ox, oy, osc, ol = O.fast_detect_grid(pyr, 30, 20, 20.0)
    gx, gy, gsc, gl = ctx.fast_detect(A, 3, 30, 20, 20.0)
    m = ol >= 0
    assert np.array_equal(gl, ol) and np.array_equal(gx[m], ox[m]) and np.array_equal(gy[m], oy[m])
    assert np.array_equal(gsc[m].view(np.uint32), osc[m].view(np.uint32))
    assert 300 < m.sum() <= 42 * 13                                     # "500 FAST corners": one per 30-px cell at most
    pts = np.stack([ox[m], oy[m]], 1).astype(np.float32)                # Feature::px -> cv::Point2f (:570-575)
    # process_second_frame: 30x30, maxLevel 4, COUNT+EPS(1000, 1e-3), USE_INITIAL_FLOW with identity guess (:578-613)
    pg = dr3.lk_params_reference(); po = O.lk_params(win=(30, 30), max_level=4, max_iters=1000, eps=1e-3, flags=O.LK_USE_INITIAL_FLOW)
    g_xy, g_st, g_err, g_L = ctx.calcOpticalFlowPyrLK(A, B, pts, pts.copy(), pg)
    w_xy, w_st, w_err, w_L, w_it = O.lk_track(A, B, pts, pts.copy(), po, order=O.ORDER_WAVE64, want_iters=True)
    r_xy, r_st, _, _, r_it = O.lk_track(A, B, pts, pts.copy(), po, order=O.ORDER_RASTER, want_iters=True)
    assert g_L == w_L == 3                                              # 4 requested, 3 effective at KITTI size with 30x30
    assert np.array_equal(g_st, w_st) and np.array_equal(g_xy.view(np.uint32), w_xy.view(np.uint32))
    assert np.array_equal(g_err.view(np.uint32), w_err.view(np.uint32))
    # against OpenCV's scalar summation order: every point whose Gauss-Newton iteration converges (stays below the
    # 100-iteration clamp on every level) agrees within the north_star's 1e-3 px.  Five frames apart on a driving
    # sequence about a fifth of the corners never converge on the coarsest level (they run into the clamp); for those the
    # last bits of the f32 sums decide where the oscillation is cut off, in OpenCV's own scalar vs SIMD builds as well.
    both = (g_st == 1) & (r_st == 1)
    conv = both & (w_it.max(1) < 100) & (r_it.max(1) < 100)
    d = np.abs(g_xy - r_xy).max(1)
    assert conv.sum() > 150 and d[conv].max() <= 1e-3
    assert (d[both] <= 1e-3).mean() >= 0.95 and np.array_equal(g_st, r_st)
    assert g_st.mean() > 0.5                                            # most corners survive
    flow = (g_xy - pts)[g_st == 1]
    assert 1.0 < np.median(np.hypot(flow[:, 0], flow[:, 1])) < 80.0


def test_config1_window_21x21_three_levels(dr3, O, ctx, kitti):
    """BASELINE configs[0] as worded: 3-level pyramid (maxLevel 2), 21x21 window, ~500 corners"""
    A, B = kitti
    x, y, sc, lv = ctx.fast_detect(A, 3, 30, 20, 20.0)
    pts = np.stack([x[lv >= 0], y[lv >= 0]], 1).astype(np.float32)
    pg = dr3.lk_params(win=(21, 21), max_level=2, max_iters=30, eps=0.01); po = O.lk_params(win=(21, 21), max_level=2, max_iters=30, eps=0.01)
    g_xy, g_st, g_err, _ = ctx.calcOpticalFlowPyrLK(A, B, pts, None, pg)
    w_xy, w_st, w_err, _ = O.lk_track(A, B, pts, None, po, order=O.ORDER_WAVE64)
    assert np.array_equal(g_st, w_st) and np.array_equal(g_xy.view(np.uint32), w_xy.view(np.uint32)) and np.array_equal(g_err.view(np.uint32), w_err.view(np.uint32))


def test_real_frame_pyramid_gradients_and_warp(dr3, O, ctx, kitti, synth):
    """the dense passes on a real frame: LK pyramid + Scharr, and a stitch-style 8UC3 warp (src/stitch.cpp:73)"""
    A, B = kitti
    fb = dr3.FrameBatch(ctx, 1240, 376, win=(21, 21), max_level=3, capacity=1)
    try:
        fb.upload(0, A); fb.build(0, 1, True); ctx.sync()
        img = A
        for l in range(fb.levels):
            assert np.array_equal(fb.download_image(0, l), img) and np.array_equal(fb.download_deriv(0, l), O.scharr(img)), l
            img = O.pyr_down(img)
    finally:
        fb.close()
    bgr = np.stack([A, B, ((A.astype(np.uint16) + B) // 2).astype(np.uint8)], -1)      # a 3-channel image built from the two frames
    M = synth.tile_homography(3, 1240, 376)
    assert np.array_equal(ctx.warpPerspective(bgr, M, (1400, 480)), O.warp_perspective(bgr, M, (1400, 480)))
