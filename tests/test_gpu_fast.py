"""FAST-10 score map, 3x3 non-max and grid-best Shi-Tomasi selection on the GPU vs the oracle's restatement of
feature_detection::FastDetector::detect (reference src/features.cpp:43-98, src/utils.cpp:282-321).  Integer
scores are bit-exact; the f32 Shi-Tomasi scores are bit-exact as well (exact integer sums, same scalar tail)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _corner_rich(synth, w, h, seed):
    """analytic texture + random rectangles: plenty of FAST corners on every pyramid level"""
    rng = np.random.default_rng(seed)
    img = synth.frame_pair(w, h)[0].astype(np.int16)
    for _ in range(w * h // 900):
        x, y = int(rng.integers(0, w - 4)), int(rng.integers(0, h - 4))
        ww, hh = int(rng.integers(3, 40)), int(rng.integers(3, 40))
        img[y:y + hh, x:x + ww] += int(rng.integers(-90, 90))
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("w,h", [(1241, 376), (640, 480), (101, 67), (64, 16)])
def test_fast10_score_map_bit_exact(dr3, O, ctx, synth, w, h):
    img = _corner_rich(synth, w, h, w + h)
    want = O.fast10_score_map(img, 20)
    got = ctx.fast_score_map(img, 20)
    assert np.array_equal(got, want)
    assert (want > 0).sum() > (20 if w > 200 else 0)
    noise = np.random.default_rng(1).integers(0, 256, (h, w), dtype=np.uint8)       # extreme contrast: scores up to 254
    assert np.array_equal(ctx.fast_score_map(noise, 35), O.fast10_score_map(noise, 35))


@pytest.mark.parametrize("w,h", [(1, 1), (5, 9), (7, 7), (8, 40), (63, 31), (64, 32), (65, 33), (66, 40), (67, 41), (129, 80), (131, 81),
                                 (192, 94), (200, 120), (61, 200)])
def test_tile_kernel_at_the_edges_of_its_tiles(dr3, O, ctx, w, h):
    """the fused detector kernel decides 64 x 40 or 64 x 32 tiles (whichever gives fewer tile rows) out of LDS, with a border of 5 rows / 8
    columns: sizes around every one of those numbers, images smaller than the ring, thresholds at both ends -- score map, and corners
    through both entry points, on an image where most pixels are candidates (every list of the kernel fills up)"""
    rng = np.random.default_rng(1000 * w + h)
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    img[rng.random((h, w)) < 0.3] = 0
    for thr in (1, 20, 254):
        assert np.array_equal(ctx.fast_score_map(img, thr), O.fast10_score_map(img, thr)), thr
    if w >= 8 and h >= 8:
        for cell, thr in ((5, 10), (16, 30)):
            o = O.fast_detect_grid([img], cell, thr, 0.0)
            g = ctx.fast_detect(img, 1, cell, thr, 0.0)
            m = o[3] >= 0
            assert np.array_equal(g[3], o[3]), (cell, thr)
            assert np.array_equal(g[0][m], o[0][m]) and np.array_equal(g[1][m], o[1][m]) and np.array_equal(g[2][m].view(np.uint32), o[2][m].view(np.uint32))


@pytest.mark.parametrize("w,h,levels,cell", [(1241, 376, 3, 30), (640, 480, 3, 30), (1920, 1080, 5, 20), (203, 131, 2, 25)])
def test_fast_detect_grid_matches_oracle(dr3, O, ctx, synth, w, h, levels, cell):
    img = _corner_rich(synth, w, h, 7 * w + h)
    pyr = O.halfsample_pyramid(img, levels, O.HALF_REF_X86)
    ox, oy, osc, ol = O.fast_detect_grid(pyr, cell, 20, 20.0)
    gx, gy, gsc, gl = ctx.fast_detect(img, levels, cell, 20, 20.0)
    assert np.array_equal(gl, ol)
    m = ol >= 0
    assert m.sum() > 0.5 * len(ol)
    assert np.array_equal(gx[m], ox[m]) and np.array_equal(gy[m], oy[m])
    assert np.array_equal(gsc[m].view(np.uint32), osc[m].view(np.uint32))
    assert len(np.unique(ol[m])) >= 2                   # corners from several pyramid levels


def test_fast_detect_respects_occupancy_and_threshold(dr3, O, ctx, synth):
    img = _corner_rich(synth, 640, 480, 3)
    pyr = O.halfsample_pyramid(img, 3, O.HALF_REF_X86)
    gc, gr = -(-640 // 30), -(-480 // 30)
    occ = (np.arange(gc * gr) % 3 == 0).astype(np.uint8)              # Detector::flag_features_grid
    for thr in (20.0, 150.0):
        o = O.fast_detect_grid(pyr, 30, 20, thr, occ)
        g = ctx.fast_detect(img, 3, 30, 20, thr, occ)
        assert np.array_equal(g[3], o[3]) and not (g[3][occ == 1] >= 0).any()
        m = o[3] >= 0
        assert np.array_equal(g[0][m], o[0][m]) and np.array_equal(g[1][m], o[1][m]) and np.array_equal(g[2][m].view(np.uint32), o[2][m].view(np.uint32))
        assert (g[2][m] > thr).all()


def test_detected_corners_feed_the_tracker(dr3, O, ctx, synth):
    """Init::process_first_frame -> process_second_frame (src/initialization.cpp:552-613): detect on A, track into B"""
    A, B, T = synth.frame_pair(640, 400)
    x, y, sc, lv = ctx.fast_detect(A, 3, 30, 20, 20.0)
    pts = np.stack([x[lv >= 0], y[lv >= 0]], 1).astype(np.float32)
    assert len(pts) > 50
    p = dr3.lk_params(win=(21, 21), max_level=3, max_iters=30, eps=0.01)
    xy, st, err, _ = ctx.calcOpticalFlowPyrLK(A, B, pts, None, p)
    gt = synth.apply_T(T, pts)
    assert st.mean() > 0.9 and np.median(np.abs(xy - gt)[st == 1]) < 0.1


def test_check_fundamental_all_hypotheses_bit_exact(dr3, O, ctx, synth):
    """InitHelper::CheckFundamental (src/initialization.cpp:171-249) over the 200 RANSAC hypotheses of FindFundamental:
    scores (sequential f32 sums) and inlier masks equal the oracle bit for bit"""
    rng = np.random.default_rng(17)
    n = 1500
    p1 = np.stack([rng.uniform(8, 1233, n), rng.uniform(8, 368, n)], 1).astype(np.float32)
    # a plausible epipolar geometry: pure forward motion + small rotation, plus noise and 20 % outliers
    K = np.array([[718.856, 0, 607.1928], [0, 718.856, 185.2157], [0, 0, 1]])
    t = np.array([0.05, -0.02, 1.0]); tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    a = 0.01; R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    Ftrue = np.linalg.inv(K).T @ tx @ R @ np.linalg.inv(K)
    depth = rng.uniform(5, 60, n)
    X = (np.linalg.inv(K) @ np.c_[p1, np.ones(n)].T) * depth
    x2 = K @ (R @ X + t[:, None]); p2 = (x2[:2] / x2[2]).T + rng.normal(0, 0.4, (n, 2))
    out = rng.random(n) < 0.2
    p2[out] += rng.uniform(-40, 40, (int(out.sum()), 2))
    p2 = p2.astype(np.float32)
    Fs = np.stack([(Ftrue / np.linalg.norm(Ftrue) + rng.normal(0, s, (3, 3)) * 1e-6) for s in np.linspace(0, 3, 200)]).astype(np.float32)
    for sigma in (1.0, 2.5):
        scores, inl = ctx.check_fundamental(Fs, p1, p2, sigma)
        for h in range(len(Fs)):
            o_sc, o_in = O.check_fundamental(Fs[h], p1, p2, sigma)
            assert scores[h].view(np.uint32) == o_sc.view(np.uint32), (h, scores[h], o_sc)
            assert np.array_equal(inl[h], o_in), h
        assert scores[0] > 0.99 * scores.max() and scores[-1] < 0.8 * scores.max() and inl[0].mean() > 0.6
    sc, none = ctx.check_fundamental(Fs[:3], p1[:0], p2[:0], 1.0)            # no matches: score 0
    assert np.array_equal(sc, np.zeros(3, np.float32))


def test_normalize_points_bit_exact(dr3, O, ctx):
    """InitHelper::Normalize (src/initialization.cpp:365-410): normalised points and T equal the oracle's sequential f32
    arithmetic bit for bit; on device-resident sets with a mask (the tracker's status bytes) a lost point is skipped exactly as
    if it had been erased from the vector first (:621-635)"""
    rng = np.random.default_rng(5)
    for n in (1, 2, 9, 546, 5000):
        xy = np.stack([rng.uniform(0, 1241, n), rng.uniform(0, 376, n)], 1).astype(np.float32)
        g_xy, g_T = ctx.normalize_points(xy)
        o_xy, o_T = O.normalize_points(xy)
        assert np.array_equal(g_xy.view(np.uint32), o_xy.view(np.uint32)) and np.array_equal(g_T.view(np.uint32), o_T.view(np.uint32)), n
    n, n_sets = 2000, 5
    sets = np.stack([np.stack([rng.uniform(-50, 1300, n), rng.uniform(-20, 400, n)], 1) for _ in range(n_sets)]).astype(np.float32)
    mask = (rng.random((n_sets, n)) < 0.85).astype(np.uint8)
    mask[3] = 1                                                              # one set without losses
    d_xy, d_m = ctx.to_device(sets), ctx.to_device(mask)
    d_out, d_T = ctx.alloc(sets.nbytes), ctx.alloc(n_sets * 36)
    try:
        ctx.normalize_points_dev(d_xy, d_m, n, n_sets, d_out, d_T); ctx.sync()
        out = d_out.download((n_sets, n, 2), np.float32); T = d_T.download((n_sets, 3, 3), np.float32)
        for k in range(n_sets):
            keep = mask[k] == 1
            o_xy, o_T = O.normalize_points(sets[k][keep])                    # the reference's way: erase, then normalise
            assert np.array_equal(out[k][keep].view(np.uint32), o_xy.view(np.uint32)) and np.array_equal(T[k].view(np.uint32), o_T.view(np.uint32)), k
            assert not out[k][~keep].any()
            m_xy, m_T = O.normalize_points(sets[k], mask[k])                 # and the oracle's own mask form agrees with that
            assert np.array_equal(m_xy.view(np.uint32), out[k].view(np.uint32)) and np.array_equal(m_T.view(np.uint32), o_T.view(np.uint32))
        ctx.normalize_points_dev(d_xy, None, n, n_sets, d_out, d_T); ctx.sync()
        out = d_out.download((n_sets, n, 2), np.float32)
        assert np.array_equal(out[0].view(np.uint32), O.normalize_points(sets[0])[0].view(np.uint32))
    finally:
        for d in (d_xy, d_m, d_out, d_T):
            d.free()


@pytest.mark.parametrize("n, n_hyp", [(1, 1), (63, 5), (64, 64), (65, 70), (1000, 200), (20000, 70)])
def test_vote_in_two_phases_keeps_the_order_of_the_sum(dr3, O, ctx, n, n_hyp):
    """the vote runs as a parallel chi-square kernel + an ordered accumulation, in rounds of 16384 matches: partial tiles, hypothesis
    counts off the wave size, a mask, more matches than one round, non-finite values -- scores and inlier bytes bit for bit"""
    rng = np.random.default_rng(n * 7 + n_hyp)
    p1 = np.stack([rng.uniform(0, 1240, n), rng.uniform(0, 375, n)], 1).astype(np.float32)
    p2 = (p1 + rng.normal(0, 3, (n, 2))).astype(np.float32)
    if n > 10:
        p2[3] = np.nan; p1[7, 0] = np.inf
    Fs = rng.normal(0, 1e-4, (n_hyp, 3, 3)).astype(np.float32); Fs[:, 2, 2] = 1e-2
    if n_hyp > 2:
        Fs[2] = 0                                         # 0 / 0 in both chi-squares
    mask = (rng.random(n) < 0.8).astype(np.uint8)
    d_F, d_p1, d_p2, d_m = ctx.to_device(Fs), ctx.to_device(p1), ctx.to_device(p2), ctx.to_device(mask)
    d_sc, d_in = ctx.alloc(4 * n_hyp), ctx.alloc(n_hyp * n)
    for use_mask in (True, False):
        ctx.check_fundamental_dev(d_F, n_hyp, d_p1, d_p2, d_m if use_mask else None, n, 1.0, d_sc, d_in)
        ctx.sync()
        sc, inl = d_sc.download((n_hyp,), np.float32), d_in.download((n_hyp, n), np.uint8)
        keep = mask != 0 if use_mask else np.ones(n, bool)
        for h in sorted({0, n_hyp // 2, n_hyp - 1, min(2, n_hyp - 1)}):
            with np.errstate(all="ignore"):
                o_sc, o_in = O.check_fundamental(Fs[h], p1[keep], p2[keep], 1.0)
            # (a NaN score is a NaN on both sides; x86 and the GPU differ in the sign bit of the default NaN)
            assert sc[h].view(np.uint32) == np.float32(o_sc).view(np.uint32) or (np.isnan(sc[h]) and np.isnan(o_sc)), (h, sc[h], o_sc)
            assert np.array_equal(inl[h][keep], o_in) and not inl[h][~keep].any()


def test_tracks_stay_on_the_device_up_to_the_inlier_vote(dr3, O, ctx, synth):
    """LK -> Normalize -> CheckFundamental without the points leaving the GPU (SURVEY 8f.4's reason to exist): the tracker's
    device outputs (prev_xy, next_xy, status) feed dr3hip_normalize_points_dev and dr3hip_check_fundamental_dev directly; both
    equal the oracle run the reference's way (lost points erased from the vectors first)"""
    A, B, Tm = synth.frame_pair(640, 400)
    pts = synth.grid_points(640, 400, 20)
    pts = np.vstack([pts, [[-30, 5], [700, 100], [3, 399]]]).astype(np.float32)          # some points the tracker will lose
    n = len(pts)
    b = dr3.FrameBatch(ctx, 640, 400, win=(15, 15), max_level=3, capacity=2)
    d_prev, d_next = ctx.to_device(pts), ctx.to_device(pts)
    d_st, d_err = ctx.alloc(n), ctx.alloc(4 * n)
    n_hyp = 64
    rng = np.random.default_rng(9)
    Fs = rng.normal(0, 1e-4, (n_hyp, 3, 3)).astype(np.float32); Fs[:, 2, 2] = 1e-2
    d_F, d_sc, d_in = ctx.to_device(Fs), ctx.alloc(4 * n_hyp), ctx.alloc(n_hyp * n)
    d_n1, d_n2, d_T = ctx.alloc(8 * n), ctx.alloc(8 * n), ctx.alloc(72)
    try:
        b.upload(0, np.stack([A, B])); b.build(0, 2, True)
        b.track(1, 0, d_prev, d_next, n, d_st, d_err, dr3.lk_params(win=(15, 15), max_level=3, max_iters=30, eps=0.01))
        ctx.normalize_points_dev(d_prev, d_st, n, 1, d_n1, d_T)
        ctx.normalize_points_dev(d_next, d_st, n, 1, d_n2, d_T.ptr + 36)
        ctx.check_fundamental_dev(d_F, n_hyp, d_prev, d_next, d_st, n, 1.0, d_sc, d_in)
        ctx.sync()
        st = d_st.download((n,), np.uint8); nxt = d_next.download((n, 2), np.float32)
        keep = st == 1
        assert 0 < (~keep).sum() < n // 4
        o1, T1 = O.normalize_points(pts[keep]); o2, T2 = O.normalize_points(nxt[keep])
        T = d_T.download((2, 3, 3), np.float32)
        assert np.array_equal(T[0].view(np.uint32), T1.view(np.uint32)) and np.array_equal(T[1].view(np.uint32), T2.view(np.uint32))
        assert np.array_equal(d_n1.download((n, 2), np.float32)[keep].view(np.uint32), o1.view(np.uint32))
        assert np.array_equal(d_n2.download((n, 2), np.float32)[keep].view(np.uint32), o2.view(np.uint32))
        sc = d_sc.download((n_hyp,), np.float32); inl = d_in.download((n_hyp, n), np.uint8)
        for h in range(n_hyp):
            o_sc, o_in = O.check_fundamental(Fs[h], pts[keep], nxt[keep], 1.0)
            assert sc[h].view(np.uint32) == o_sc.view(np.uint32) and np.array_equal(inl[h][keep], o_in) and not inl[h][~keep].any(), h
    finally:
        for d in (d_prev, d_next, d_st, d_err, d_F, d_sc, d_in, d_n1, d_n2, d_T):
            d.free()
        b.close()


def test_batch_detect_matches_host_detect_and_feeds_the_tracker(dr3, O, ctx, synth):
    """device-resident front end (dr3hip_batch_detect): per slot the same cells as the host-image entry point and the
    oracle -- including the odd-width KITTI shape, where the reference's half-sampler walks the level-0 image as a
    continuous cv::Mat while the slot plane is pitched -- and its point array drives dr3hip_batch_track directly"""
    w, h, n_fr = 1241, 376, 3
    frames = [_corner_rich(synth, w, h, 40 + i) for i in range(n_fr)]
    frames[1] = np.roll(frames[0], (1, 2), axis=(0, 1))                  # slot 1 = slot 0 moved by (+2, +1) px
    b = dr3.FrameBatch(ctx, w, h, win=(15, 15), max_level=3, capacity=n_fr)
    bufs = []
    try:
        b.upload(0, np.stack(frames))
        b.build(0, n_fr, True)
        gc, gr = -(-w // 30), -(-h // 30)
        n_cells = gc * gr
        d_xy = ctx.alloc(n_fr * n_cells * 8); d_c = ctx.alloc(n_fr * n_cells * 16)
        bufs = [d_xy, d_c]
        assert b.detect(0, n_fr, 3, 30, 20, 20.0, d_xy=d_xy, d_corners=d_c) == n_cells
        ctx.sync()
        xy = d_xy.download((n_fr, n_cells, 2), np.float32)
        cor = d_c.download((n_fr, n_cells), np.dtype([("x", np.int32), ("y", np.int32), ("level", np.int32), ("score", np.float32)]))
        for i in range(n_fr):
            gx, gy, gs, gl = ctx.fast_detect(frames[i], 3, 30, 20, 20.0)
            ox, oy, os_, ol = O.fast_detect_grid(O.halfsample_pyramid(frames[i], 3, O.HALF_REF_X86), 30, 20, 20.0)
            assert np.array_equal(cor["level"][i], ol) and np.array_equal(gl, ol)
            m = ol >= 0
            assert m.sum() > 300
            assert np.array_equal(cor["x"][i][m], ox[m]) and np.array_equal(cor["y"][i][m], oy[m])
            assert np.array_equal(cor["score"][i][m].view(np.uint32), os_[m].view(np.uint32))
            assert np.array_equal(xy[i][m], np.stack([ox[m], oy[m]], 1).astype(np.float32))
            assert (xy[i][~m] == dr3.NO_POINT).all()
        # track slot 0 -> slot 1 from the detected points, no host hop
        d_next = ctx.alloc(n_cells * 8); d_st = ctx.alloc(n_cells); d_err = ctx.alloc(n_cells * 4)
        bufs += [d_next, d_st, d_err]
        d_next.upload(xy[0])
        p = dr3.lk_params(win=(15, 15), max_level=3, max_iters=30, eps=0.01)
        b.track(1, 0, d_xy, d_next, n_cells, d_st, d_err, p)
        ctx.sync()
        st = d_st.download((n_cells,), np.uint8); nxt = d_next.download((n_cells, 2), np.float32)
        m = cor["level"][0] >= 0
        assert not st[~m].any()                                          # "no feature" cells come back lost
        ok = m & (st == 1)
        assert ok.sum() > 0.8 * m.sum()
        assert np.median(np.abs(nxt[ok] - (xy[0][ok] + np.array([2, 1], np.float32)))) < 0.05
        o_xy, o_st, _, _ = O.lk_track(frames[0], frames[1], xy[0][m], None, O.lk_params(win=(15, 15), max_level=3, max_iters=30, eps=0.01),
                                      order=O.ORDER_WAVE64)
        assert np.array_equal(st[m], o_st) and np.array_equal(nxt[m].view(np.uint32), o_xy.view(np.uint32))
    finally:
        for d in bufs:
            d.free()
        b.close()
