"""CPU-side pins of the oracle (no GPU).  The reference holds no golden vectors for this path
("parity unpinned"), so the oracle is pinned by
  (i)  analytic known-answer tests derived from the algorithm definitions, and
  (ii) an independent numpy restatement that lays data out the way OpenCV does (physical padding)
       while the C oracle reflects indices -- the two must agree bit for bit."""
import numpy as np
import pytest

from oracle import np_restatement as R


# ---- (ii) C oracle == independent numpy restatement ---------------------------------------------
@pytest.mark.parametrize("h,w", [(37, 53), (64, 64), (5, 7), (6, 4), (101, 33), (376, 1241)])
def test_pyrdown_scharr_halfsample_match_numpy(O, h, w):
    rng = np.random.default_rng(h * 1000 + w)
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    assert np.array_equal(O.pyr_down(img), R.pyr_down(img))
    assert np.array_equal(O.scharr(img), R.scharr(img))
    assert np.array_equal(O.halfsample(img, O.HALF_TRUNC), R.halfsample_trunc(img))
    assert np.array_equal(O.halfsample(img, O.HALF_SSE2), R.halfsample_sse2(img))
    if h * w <= 40000:
        assert np.array_equal(O.halfsample(img, O.HALF_REF_SCALAR), R.halfsample_ref_scalar(img))


@pytest.mark.parametrize("win,ml,flags", [((21, 21), 3, 0), ((15, 15), 2, 4), ((8, 10), 2, 8), ((5, 5), 4, 12)])
def test_lk_matches_numpy_restatement_bit_exact(O, synth, win, ml, flags):
    A, B, T = synth.frame_pair(200, 140)
    pts = synth.grid_points(200, 140, 28, border=2)
    pts = np.vstack([pts, [[-3, 10], [199, 139], [0.5, 0.5], [230, 50]]]).astype(np.float32)
    guess = (synth.apply_T(T, pts) + 0.4).astype(np.float32)
    p = O.lk_params(win=win, max_level=ml, max_iters=20, eps=0.01, flags=flags)
    c_xy, c_st, c_err, c_L = O.lk_track(A, B, pts, guess, p, order=O.ORDER_RASTER)
    n_xy, n_st, n_err, n_L = R.lk_track(A, B, pts, guess, win=win, max_level=ml, max_iters=20, eps=0.01, flags=flags)
    assert c_L == n_L
    assert np.array_equal(c_st, n_st)
    assert np.array_equal(c_xy.view(np.uint32), n_xy.view(np.uint32))
    assert np.array_equal(c_err.view(np.uint32), n_err.view(np.uint32))


def test_warp_and_remap_match_numpy(O):
    rng = np.random.default_rng(1)
    src = rng.integers(0, 256, (60, 80, 3), dtype=np.uint8)
    srcf = rng.random((60, 80, 4)).astype(np.float32)
    for k in range(12):
        ang = rng.uniform(-0.3, 0.3); c, s = np.cos(ang), np.sin(ang)
        M = np.array([[c * rng.uniform(.8, 1.2), -s, rng.uniform(-10, 30)], [s, c * rng.uniform(.8, 1.2), rng.uniform(-10, 30)],
                      [rng.uniform(-1e-3, 1e-3), rng.uniform(-1e-3, 1e-3), 1]], np.float32)
        for S in (src, srcf, src[..., 0].copy()):
            assert np.array_equal(O.warp_perspective(S, M, (117, 93)), R.warp_perspective(S, M, (117, 93)))
        assert np.array_equal(O.warp_perspective(src, M, (117, 93), flags=O.WARP_INVERSE_MAP),
                              R.warp_perspective(src, M, (117, 93), inverse_map=True))
    mx = rng.uniform(-3, 83, (50, 70)).astype(np.float32); my = rng.uniform(-3, 63, (50, 70)).astype(np.float32)
    assert np.array_equal(O.remap(src, mx, my), R.remap(src, mx, my))
    assert np.array_equal(O.remap(srcf, mx, my), R.remap(srcf, mx, my))
    Minv, ok = O.invert3x3(M)
    assert ok and np.array_equal(Minv, R.invert3x3(M.astype(np.float64)))


# ---- (i) analytic known answers -----------------------------------------------------------------------
def test_pyrdown_constant_and_linear_ramp(O):
    assert (O.pyr_down(np.full((33, 47), 173, np.uint8)) == 173).all()
    # [1 4 6 4 1]/16 reproduces affine signals: out(X,Y) = in(2X,2Y) away from the border
    yy, xx = np.mgrid[0:40, 0:64]
    img = (2 * xx + yy).astype(np.uint8)
    out = O.pyr_down(img)
    assert out.shape == (20, 32)
    assert np.array_equal(out[1:-1, 1:-1], img[2:-2:2, 2:-2:2])


def test_pyrdown_impulse_response_is_binomial_outer_product(O):
    img = np.zeros((21, 21), np.uint8); img[10, 10] = 255
    out = O.pyr_down(img).astype(int)
    k = np.array([1, 4, 6, 4, 1])
    K = np.outer(k, k) * 255
    # output (X,Y) sees the impulse with taps (10-2X+2, 10-2Y+2): X,Y in {4,5,6}
    for X in (4, 5, 6):
        for Y in (4, 5, 6):
            assert out[Y, X] == (K[10 - 2 * Y + 2, 10 - 2 * X + 2] + 128) >> 8


def test_scharr_on_ramps(O):
    yy, xx = np.mgrid[0:20, 0:30]
    d = O.scharr((3 * xx).astype(np.uint8))
    assert (d[:, 1:-1, 0] == 32 * 3).all() and (d[..., 1] == 0).all()        # (3+10+3) * 2a = 32 a
    assert (d[:, 0, 0] == 0).all() and (d[:, -1, 0] == 0).all()              # REFLECT_101: zero slope at the edge
    d = O.scharr((5 * yy).astype(np.uint8))
    assert (d[1:-1, :, 1] == 32 * 5).all() and (d[..., 0] == 0).all()
    step = np.zeros((9, 9), np.uint8); step[:, 5:] = 255
    assert np.abs(O.scharr(step)).max() == 16 * 255                          # largest value fits int16


def test_effective_max_level_matches_survey(O):
    assert O.lk_effective_max_level(1241, 376, 30, 30, 4) == 3    # SURVEY A-1: level 4 = 78x24 <= 30
    assert O.lk_effective_max_level(1241, 376, 21, 21, 4) == 4
    assert O.lk_effective_max_level(1241, 376, 15, 15, 3) == 3
    assert O.lk_effective_max_level(1920, 1080, 21, 21, 4) == 4
    assert O.lk_effective_max_level(64, 64, 31, 31, 5) == 1
    assert O.lk_effective_max_level(64, 64, 32, 32, 5) == 0


def test_lk_recovers_known_flow(O, synth):
    """analytic frame pair with exact ground truth: sub-pixel translation + rotation + zoom"""
    A, B, T = synth.frame_pair()
    pts = synth.grid_points(synth.KITTI_W, synth.KITTI_H, 30)
    assert len(pts) == 42 * 13                                              # mirrors src/features.cpp:15-16
    xy, st, err, L = O.lk_track(A, B, pts, None, O.lk_params(win=(21, 21), max_level=2))
    gt = synth.apply_T(T, pts)
    ok = st == 1
    assert ok.mean() > 0.95
    e = np.abs(xy - gt)[ok]
    assert np.median(e) < 0.05 and np.percentile(e, 90) < 0.2


def test_lk_identity_and_status_semantics(O, synth):
    A, _, _ = synth.frame_pair(240, 160)
    pts = np.array([[60, 60], [120.5, 80.25], [-200, 5], [5, 900]], np.float32)
    xy, st, err, _ = O.lk_track(A, A, pts, None, O.lk_params(win=(15, 15), max_level=2))
    assert list(st) == [1, 1, 0, 0]
    assert np.abs(xy[:2] - pts[:2]).max() < 1e-4 and (err[:2] == 0).all()
    flat = np.full((160, 240), 9, np.uint8)
    _, st, _, _ = O.lk_track(flat, flat, pts[:2], None, O.lk_params(win=(15, 15), max_level=2))
    assert not st.any()                                                      # minEig below threshold


def test_lk_summation_orders_agree_within_tolerance(O, synth):
    """raster (OpenCV scalar) vs wave64 (HIP kernel) float summation order: the documented bar"""
    A, B, T = synth.frame_pair()
    pts = synth.grid_points(synth.KITTI_W, synth.KITTI_H, 15, n=2000)
    p = O.lk_params(win=(15, 15), max_level=3, max_iters=30, eps=0.01)
    r = O.lk_track(A, B, pts, None, p, order=O.ORDER_RASTER)
    w = O.lk_track(A, B, pts, None, p, order=O.ORDER_WAVE64)
    assert np.array_equal(r[1], w[1])
    d = np.abs(r[0] - w[0])[r[1] == 1]
    assert (d <= 1e-3).mean() >= 0.998 and d.max() < 0.05


def test_warp_identity_translation_border(O):
    rng = np.random.default_rng(2)
    src = rng.integers(0, 256, (60, 80, 3), dtype=np.uint8)
    srcf = rng.random((60, 80, 4)).astype(np.float32)
    I = np.eye(3, dtype=np.float32)
    assert np.array_equal(O.warp_perspective(src, I, (80, 60)), src)
    assert np.array_equal(O.warp_perspective(srcf, I, (80, 60)), srcf)
    M = np.array([[1, 0, 5], [0, 1, 3], [0, 0, 1]], np.float32)
    d = O.warp_perspective(src, M, (90, 70))
    assert np.array_equal(d[3:63, 5:85], src) and not d[:3].any() and not d[:, :5].any() and not d[63:].any()
    # half-pixel shift of a constant image: interior stays constant, the border row fades to half
    c = np.full((10, 10), 200, np.uint8)
    d = O.warp_perspective(c, np.array([[1, 0, 0.5], [0, 1, 0], [0, 0, 1]], np.float32), (10, 10))
    assert (d[:, 1:] == 200).all() and (d[:, 0] == 100).all()


def test_warp_quantises_to_one_32nd_pixel(O):
    """shifts that round to the same 1/32 px give identical images; the next 1/32 differs"""
    rng = np.random.default_rng(3)
    src = rng.integers(0, 256, (40, 50), dtype=np.uint8)
    T = lambda tx: np.array([[1, 0, tx], [0, 1, 0], [0, 0, 1]], np.float32)
    a = O.warp_perspective(src, T(0.25), (50, 40), flags=O.WARP_INVERSE_MAP)
    b = O.warp_perspective(src, T(0.25 + 1 / 128), (50, 40), flags=O.WARP_INVERSE_MAP)
    c = O.warp_perspective(src, T(0.25 + 1 / 32), (50, 40), flags=O.WARP_INVERSE_MAP)
    assert np.array_equal(a, b) and not np.array_equal(a, c)


def test_halfsample_reference_quirks(O):
    """SURVEY Appendix B-1/B-2"""
    q = np.array([[1, 0], [0, 0]], np.uint8)
    assert O.halfsample(q, O.HALF_TRUNC)[0, 0] == 0 and O.halfsample(q, O.HALF_SSE2)[0, 0] == 1
    # odd width: output row k of the scalar walk starts at flat offset 2k*stride - k
    w, h = 13, 8
    img = np.arange(w * h, dtype=np.int64).reshape(h, w).astype(np.uint8)
    out = O.halfsample(img, O.HALF_REF_SCALAR)
    flat = img.ravel().astype(int)
    for k in range(h // 2):
        t = 2 * k * w - k
        assert out[k, 0] == (flat[t] + flat[t + 1] + flat[t + w] + flat[t + w + 1]) // 4
    # even width: the walk is row aligned
    img = np.random.default_rng(4).integers(0, 256, (10, 12), dtype=np.uint8)
    assert np.array_equal(O.halfsample(img, O.HALF_REF_SCALAR), O.halfsample(img, O.HALF_TRUNC))
    # x86 dispatch: cols % 16 == 0 -> SSE2 arithmetic, otherwise the scalar walk (src/utils.cpp:387-392)
    a = np.random.default_rng(5).integers(0, 256, (8, 32), dtype=np.uint8)
    assert np.array_equal(O.halfsample(a, O.HALF_REF_X86), O.halfsample(a, O.HALF_SSE2))
    b = np.random.default_rng(6).integers(0, 256, (8, 31), dtype=np.uint8)
    assert np.array_equal(O.halfsample(b, O.HALF_REF_X86), O.halfsample(b, O.HALF_REF_SCALAR))
    pyr = O.halfsample_pyramid(np.zeros((376, 1241), np.uint8), 3)
    assert [p.shape for p in pyr] == [(376, 1241), (188, 620), (94, 310)]     # floor halving


def test_shi_tomasi_and_feather_helpers(O):
    flat = np.full((40, 40), 50, np.uint8)
    assert O.shi_tomasi_score(flat, 20, 20) == 0.0
    assert O.shi_tomasi_score(flat, 2, 20) == 0.0                              # too close to the border
    yy, xx = np.mgrid[0:40, 0:40]
    corner = (((xx > 20) & (yy > 20)) * 200).astype(np.uint8)
    edge = ((xx > 20) * 200).astype(np.uint8)
    assert O.shi_tomasi_score(corner, 20, 20) > 100 * max(O.shi_tomasi_score(edge, 20, 20), 1e-3)
    ls = O.linspace_f32(0.0, 1.0, 50)
    assert ls[0] == 0 and abs(ls[-1] - 1) < 1e-5 and len(ls) == 50
    canvas = np.zeros((4, 6, 4), np.float32)
    warped = np.ones((4, 6, 4), np.float32) * 10
    warped[0, 2, :3] = 0                                                      # black pixel: dropped (:205-207)
    ramp = np.array([0.5, 1.0, 1.0], np.float32)
    O.feather_accumulate(canvas, warped, 1, 4, ramp)
    assert canvas[1, 1, 0] == 5 and canvas[1, 1, 3] == 0.5 and canvas[0, 2, 3] == 0 and canvas[2, 0, 3] == 0
    out = O.normalize_canvas(canvas)
    assert out[1, 1, 0] == 10 and out[0, 2, 0] == 0


def test_oracle_matches_numpy_restatement_on_the_reference_kitti_pair(O):
    """the reference's own two-view input (tests/test_init.cpp:24-25, packed in tests/golden/kitti_pair.npz): the C oracle
    and the independently laid-out numpy restatement agree bit for bit on real image statistics too"""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kitti_pair.npz"))
    A, B = np.ascontiguousarray(z["kitti0"]), np.ascontiguousarray(z["kitti5"])
    assert A.shape == (376, 1240)
    assert np.array_equal(O.pyr_down(A), R.pyr_down(A)) and np.array_equal(O.scharr(A), R.scharr(A))
    assert np.array_equal(O.halfsample(A, O.HALF_REF_SCALAR), R.halfsample_ref_scalar(A))
    pyr = O.halfsample_pyramid(A, 3, O.HALF_REF_X86)
    x, y, sc, lv = O.fast_detect_grid(pyr, 30, 20, 20.0)
    pts = np.stack([x[lv >= 0], y[lv >= 0]], 1).astype(np.float32)[::12]            # ~30 corners keep the numpy loop short
    po = O.lk_params(win=(21, 21), max_level=2, max_iters=30, eps=0.01)
    o_xy, o_st, o_err, _ = O.lk_track(A, B, pts, None, po, order=O.ORDER_RASTER)
    n_xy, n_st, n_err, _ = R.lk_track(A, B, pts, None, win=(21, 21), max_level=2, max_iters=30, eps=0.01)
    assert np.array_equal(o_st, n_st) and np.array_equal(o_xy.view(np.uint32), np.asarray(n_xy, np.float32).view(np.uint32))


def test_fast10_known_answers(O):
    """FAST-10 by construction: the corner pixel of a bright square on a dark background sees 11 contiguous darker ring
    pixels (a straight edge sees at most 9), and the bisection score is the contrast minus one"""
    img = np.full((40, 40), 50, np.uint8)
    img[10:30, 10:30] = 150
    sm = O.fast10_score_map(img, 20)
    corners = {(10, 10), (29, 10), (10, 29), (29, 29)}                      # (x, y) of the square's corner pixels
    for x, y in corners:
        assert sm[y, x] == 99, (x, y, sm[y, x])                            # largest b with 10 contiguous < c - b: 150 - 50 - 1
    assert sm[10, 20] == 0 and sm[20, 10] == 0 and sm[20, 20] == 0          # edges and interior are no corners
    assert O.fast10_score_map(img, 100)[10, 10] == 0                        # threshold above the contrast: nothing
    assert (sm[:3] == 0).all() and (sm[:, :3] == 0).all()                   # the detector scans [3, size - 3)
    # 3x3 non-maximum suppression drops a corner when an adjacent corner scores >= it (ties fall together)
    m = np.zeros((6, 8), np.uint8); m[2, 2] = 30; m[2, 3] = 30; m[4, 6] = 40; m[3, 5] = 39
    keep = O.fast_nonmax_3x3(m)
    assert keep[2, 2] == 0 and keep[2, 3] == 0 and keep[4, 6] == 1 and keep[3, 5] == 0 and keep.sum() == 1


def test_check_fundamental_known_answers(O):
    """CheckFundamental by construction: for a pure x-translation F = [t]_x the epipolar lines are the rows, so matches on the
    same row have zero residual and each contributes 2 * 5.991; a match 3 rows off (sigma 1: chi2 = 9 > 3.841) contributes nothing"""
    F = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32)
    p1 = np.array([[10, 20], [300, 45], [77, 199], [5, 5]], np.float32)
    p2 = p1 + np.array([[7, 0], [-3, 0], [12, 0], [4, 3]], np.float32)
    score, inl = O.check_fundamental(F, p1, p2, 1.0)
    assert inl.tolist() == [1, 1, 1, 0]
    assert score == np.float32(5.991) + np.float32(5.991) + np.float32(5.991) + np.float32(5.991) + np.float32(5.991) + np.float32(5.991)
    score3, inl3 = O.check_fundamental(F, p1, p2, 3.0)                      # sigma 3: chi2 = 1 on both sides -> inlier, 2 * (5.991 - 1)
    assert inl3.tolist() == [1, 1, 1, 1] and abs(float(score3) - (6 * 5.991 + 2 * 4.991)) < 1e-4


def test_lk_non_finite_points_are_lost(O, synth):
    """cvFloor on the reference's x86 build is cvttss2si: NaN, infinities and values outside the int range all become
    INT_MIN, fail LKTrackerInvoker's bounds test on every level and leave status 0, err 0 and the (scaled and re-doubled,
    hence unchanged) input coordinate"""
    A, B, _ = synth.frame_pair(160, 96)
    pts = np.array([[np.nan, 20], [30, np.nan], [np.inf, 20], [30, -np.inf], [3e9, 20], [30, -1e20], [60, 40]], np.float32)
    for order in (O.ORDER_RASTER, O.ORDER_WAVE64):
        xy, st, err, _ = O.lk_track(A, B, pts, None, O.lk_params(win=(9, 9), max_level=2, max_iters=10, eps=0.01, flags=0), order=order)
        assert st.tolist() == [0, 0, 0, 0, 0, 0, 1] and not err[:6].any()
        assert np.array_equal(xy[:6], pts[:6], equal_nan=True)


def test_normalize_points_known_answers(O):
    """InitHelper::Normalize (src/initialization.cpp:365-410): a symmetric cross has mean 0 and mean |dev| = its arm lengths"""
    xy = np.array([[3, 0], [-3, 0], [0, 5], [0, -5]], np.float32) + np.float32([100, 40])
    out, T = O.normalize_points(xy)
    # mean (100, 40); mean |dx| = 6/4 = 1.5, mean |dy| = 10/4 = 2.5
    assert np.allclose(T, [[1 / 1.5, 0, -100 / 1.5], [0, 1 / 2.5, -40 / 2.5], [0, 0, 1]], rtol=1e-6)
    assert np.allclose(out, [[2, 0], [-2, 0], [0, 2], [0, -2]], atol=1e-5)
    assert np.allclose(np.abs(out).mean(0), 1.0, atol=1e-6)                     # the normalised points have unit mean deviation
    # T maps the original points onto the normalised ones
    h = (T @ np.c_[xy, np.ones(4)].T).T
    assert np.allclose(h[:, :2], out, atol=1e-4)
    # a masked point is skipped as if erased
    xy5 = np.vstack([xy, [[1e6, -1e6]]]).astype(np.float32)
    out5, T5 = O.normalize_points(xy5, np.array([1, 1, 1, 1, 0], np.uint8))
    assert np.array_equal(T5, T) and np.array_equal(out5[:4], out) and not out5[4].any()


def test_sparse_alignment_recovers_a_known_camera_motion(O, synth):
    """the extension of SURVEY 8(f) row 5 on an analytic scene: a textured plane seen from two poses (the second view is the
    plane-induced homography of the first).  From the identity the alignment recovers rotation to 1e-4 and translation to 1 %"""
    w, h = 640, 400
    K = np.array([[718.856, 0, 319.5], [0, 718.856, 199.5], [0, 0, 1.0]])
    R = synth.rodrigues([0.004, -0.006, 0.003]); t = np.array([0.05, -0.02, 0.30])
    A, B, H = synth.plane_scene(w, h, K, R, t)
    uv = synth.grid_points(w, h, 30, border=24)
    xyz = synth.plane_points(uv, K)
    # the scene is consistent: the 3-D points project through (R, t) where the homography sends their pixels
    p2 = (K @ (R @ xyz.T.astype(np.float64) + t[:, None])); p2 = (p2[:2] / p2[2]).T
    ph = (H @ np.c_[uv, np.ones(len(uv))].T); ph = (ph[:2] / ph[2]).T
    assert np.abs(p2 - ph).max() < 1e-3
    pa = O.align_params(K[0, 0], K[1, 1], K[0, 2], K[1, 2], max_level=4, min_level=1, n_iter=12, eps=1e-9)
    T, n_meas, chi2, iters = O.sparse_align(O.lk_pyramid(A, 5), O.lk_pyramid(B, 5), xyz, pa)
    assert np.abs(T[:, :3] - R).max() < 1e-4 and np.abs(T[:, 3] - t).max() < 3e-3, T
    assert n_meas > 0.9 * 16 * len(uv) and chi2 < 2.0 and iters[0] == 0 and all(iters[1:5] > 0)
    # identical frames: zero residual, the pose stays the identity
    T0, n0, c0, it0 = O.sparse_align(O.lk_pyramid(A, 5), O.lk_pyramid(A, 5), xyz, pa)
    assert c0 == 0 and np.abs(T0 - np.eye(3, 4)).max() < 1e-12
