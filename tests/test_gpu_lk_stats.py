"""the tracker's loss reasons and counters (counters returned in a result struct): DR3HIP_LK_LOST_* per point from
dr3hip_batch_track_diag, reduced on the device by dr3hip_lk_stats"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _track(dr3, ctx, A, B, pts, win=(15, 15), max_level=2, min_eig=1e-4):
    h, w = A.shape
    n = len(pts)
    p = dr3.lk_params(win=win, max_level=max_level, max_iters=30, eps=0.01, min_eig_threshold=min_eig)
    b = dr3.FrameBatch(ctx, w, h, win=win, max_level=max_level, capacity=2)
    b.upload(0, np.stack([A, B]))
    b.build(0, 2, True)
    P = np.ascontiguousarray(pts, np.float32)
    d_prev, d_next = ctx.to_device(P), ctx.to_device(P)
    d_st, d_err, d_rs = ctx.alloc(n), ctx.alloc(4 * n), ctx.alloc(n)
    L = b.track(1, 0, d_prev, d_next, n, d_st, d_err, p, d_reason=d_rs)
    d_it = ctx.alloc(4 * n * (L + 1))
    d_next2 = ctx.to_device(P)
    b.track(1, 0, d_prev, d_next2, n, d_st, d_err, p, d_iters=d_it, d_reason=d_rs)
    ctx.sync()
    st, rs = d_st.download((n,), np.uint8), d_rs.download((n,), np.uint8)
    it = d_it.download((n, L + 1), np.int32)
    xy = d_next2.download((n, 2), np.float32)
    return st, rs, it, xy, ctx.lk_stats(n, d_st, d_rs, d_it, L + 1), L


def test_reasons_and_counters_on_a_textured_pair(dr3, O, synth, ctx):
    A, B, T = synth.frame_pair(400, 260, motion=dict(tx=6.0, ty=-4.0, rot_deg=0.3, zoom=1.0), seed=5)
    rng = np.random.default_rng(3)
    pts = np.concatenate([synth.grid_points(400, 260, 20, border=2),
                          rng.uniform([-30, -30], [430, 290], (200, 2)),            # many start outside
                          np.array([[np.nan, 10.0], [5.0, np.inf]])]).astype(np.float32)
    st, rs, it, xy, c, L = _track(dr3, ctx, A, B, pts)
    n = len(pts)
    # a reason exactly where the status is 0, and only the four codes
    assert np.array_equal(rs == 0, st == 1) and rs.max() <= 4
    # START_OUTSIDE is decidable from the previous position alone: the 15x15 window at floor(p - 7) must start in [-15, size)
    with np.errstate(invalid="ignore"):
        ip = np.floor(pts - np.float32(7.0))
        outside = ~((ip[:, 0] >= -15) & (ip[:, 0] < 400) & (ip[:, 1] >= -15) & (ip[:, 1] < 260))
    assert np.array_equal(rs == dr3.LK_LOST_START_OUTSIDE, outside) and outside.sum() > 20
    # the status / positions are those of the plain call and of the checker
    o_xy, o_st, o_err, _ = O.lk_track(A, B, pts, None, O.lk_params(win=(15, 15), max_level=2, max_iters=30, eps=0.01), order=O.ORDER_WAVE64)
    assert np.array_equal(st, o_st) and np.array_equal(xy.view(np.uint32), o_xy.view(np.uint32))
    # counters = the host-side counts of the same arrays
    assert c["n_points"] == n and c["n_tracked"] == int(st.sum())
    for code, key in ((1, "n_lost_start_outside"), (2, "n_lost_min_eig"), (3, "n_lost_iterate_outside"), (4, "n_lost_result_outside")):
        assert c[key] == int((rs == code).sum()), key
    assert c["n_tracked"] + sum(c[k] for k in c if k.startswith("n_lost")) == n
    assert c["iters_total"] == int(it.sum())
    hist = np.bincount(np.minimum(it.ravel(), 31), minlength=32)
    assert c["iters_hist"] == [int(v) for v in hist]
    assert c["n_lost_iterate_outside"] + c["n_lost_result_outside"] > 0        # the motion pushes border points out


def test_flat_image_is_all_min_eig(dr3, synth, ctx):
    A = np.full((120, 160), 90, np.uint8)
    pts = synth.grid_points(160, 120, 16, border=20)
    st, rs, it, xy, c, L = _track(dr3, ctx, A, A, pts, win=(11, 11), max_level=1)
    assert not st.any() and (rs == dr3.LK_LOST_MIN_EIG).all()
    assert c["n_lost_min_eig"] == len(pts) and c["n_tracked"] == 0 and c["iters_total"] == 0


def test_stats_without_optional_arrays(dr3, synth, ctx):
    A, B, T = synth.frame_pair(200, 150, seed=9)
    pts = synth.grid_points(200, 150, 25, border=10)
    st, rs, it, xy, c, L = _track(dr3, ctx, A, B, pts)
    d_st = ctx.to_device(st)
    c2 = ctx.lk_stats(len(pts), d_st)
    assert c2["n_tracked"] == int(st.sum()) and c2["iters_total"] == 0 and sum(c2["iters_hist"]) == 0
    assert c2["n_lost_min_eig"] == 0
    assert ctx.lk_stats(0, d_st)["n_points"] == 0
