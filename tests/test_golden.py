"""Committed golden vectors (tests/golden/hotpath_small.npz, made by tests/golden/make_golden.py).
They are SELF-golden (oracle outputs; the reference has no fixtures for this path): the CPU test
pins the oracle against silent drift, the GPU test checks the HIP path where nothing but the repo
snapshot exists."""
import os

import numpy as np
import pytest

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hotpath_small.npz"))


def _lk_cases():
    for tag in "abc":
        w, h, ml, it, fl = [int(v) for v in G["lk_%s_params" % tag]]
        yield tag, dict(win=(w, h), max_level=ml, max_iters=it, eps=float(G["lk_%s_eps" % tag]), flags=fl)


def test_oracle_reproduces_golden(O):
    A, B, pts = G["lk_prev"], G["lk_next"], G["lk_pts"]
    for tag, kw in _lk_cases():
        for oname, order in (("raster", O.ORDER_RASTER), ("wave64", O.ORDER_WAVE64)):
            xy, st, err, L = O.lk_track(A, B, pts, G["lk_%s_guess" % tag], O.lk_params(**kw), order=order)
            assert L == int(G["lk_%s_%s_L" % (tag, oname)])
            assert np.array_equal(st, G["lk_%s_%s_st" % (tag, oname)])
            assert np.array_equal(xy.view(np.uint32), G["lk_%s_%s_xy" % (tag, oname)].view(np.uint32))
            assert np.array_equal(err.view(np.uint32), G["lk_%s_%s_err" % (tag, oname)].view(np.uint32))
    assert np.array_equal(O.pyr_down(A), G["pyr1"]) and np.array_equal(O.pyr_down(G["pyr1"]), G["pyr2"])
    assert np.array_equal(O.scharr(A), G["scharr0"]) and np.array_equal(O.scharr(G["pyr1"]), G["scharr1"])
    src3, M = G["warp_src3"], G["warp_M"]
    src4 = np.ones((64, 96, 4), np.float32); src4[..., :3] = src3
    assert np.array_equal(O.warp_perspective(src3, M, (120, 80)), G["warp_8uc3"])
    assert np.array_equal(O.warp_perspective(src4, M, (120, 80)).view(np.uint32), G["warp_32fc4"].view(np.uint32))
    assert np.array_equal(O.warp_perspective(src3, M, (120, 80), flags=O.WARP_INVERSE_MAP), G["warp_8uc3_inv"])
    assert np.array_equal(O.remap(src3, G["remap_mx"], G["remap_my"]), G["remap_8uc3"])
    for name, mode in (("trunc", O.HALF_TRUNC), ("sse2", O.HALF_SSE2), ("refscalar", O.HALF_REF_SCALAR), ("refx86", O.HALF_REF_X86)):
        assert np.array_equal(O.halfsample(G["half_src_odd"], mode), G["half_odd_" + name])
        assert np.array_equal(O.halfsample(G["half_src_16"], mode), G["half_16_" + name])


@pytest.mark.gpu
def test_hip_reproduces_golden(dr3, ctx):
    A, B, pts = G["lk_prev"], G["lk_next"], G["lk_pts"]
    for tag, kw in _lk_cases():
        xy, st, err, L = ctx.calcOpticalFlowPyrLK(A, B, pts, G["lk_%s_guess" % tag], dr3.lk_params(**kw))
        assert L == int(G["lk_%s_wave64_L" % tag])
        assert np.array_equal(st, G["lk_%s_wave64_st" % tag])
        assert np.array_equal(xy.view(np.uint32), G["lk_%s_wave64_xy" % tag].view(np.uint32))      # bit-exact, kernel order
        assert np.array_equal(err.view(np.uint32), G["lk_%s_wave64_err" % tag].view(np.uint32))
        r_xy, r_st = G["lk_%s_raster_xy" % tag], G["lk_%s_raster_st" % tag]
        both = (st == 1) & (r_st == 1)
        assert np.abs(xy - r_xy)[both].max() <= 1e-3                                               # north_star tolerance
    b = dr3.FrameBatch(ctx, 160, 96, win=(15, 15), max_level=3, capacity=1)
    try:
        b.upload(0, A[None]); b.build(0, 1, True); ctx.sync()
        assert np.array_equal(b.download_image(0, 1), G["pyr1"]) and np.array_equal(b.download_image(0, 2), G["pyr2"])
        assert np.array_equal(b.download_deriv(0, 0), G["scharr0"]) and np.array_equal(b.download_deriv(0, 1), G["scharr1"])
    finally:
        b.close()
    src3, M = G["warp_src3"], G["warp_M"]
    src4 = np.ones((64, 96, 4), np.float32); src4[..., :3] = src3
    assert np.array_equal(ctx.warpPerspective(src3, M, (120, 80)), G["warp_8uc3"])
    assert np.array_equal(ctx.warpPerspective(src4, M, (120, 80)).view(np.uint32), G["warp_32fc4"].view(np.uint32))
    assert np.array_equal(ctx.warpPerspective(src3, M, (120, 80), dr3.WARP_INVERSE_MAP), G["warp_8uc3_inv"])
    assert np.array_equal(ctx.remap(src3, G["remap_mx"], G["remap_my"]), G["remap_8uc3"])
    for name, mode in (("trunc", dr3.HALF_TRUNC), ("sse2", dr3.HALF_SSE2), ("refscalar", dr3.HALF_REF_SCALAR), ("refx86", dr3.HALF_REF_X86)):
        assert np.array_equal(ctx.create_img_pyramid(G["half_src_odd"], 2, mode)[1], G["half_odd_" + name])
        assert np.array_equal(ctx.create_img_pyramid(G["half_src_16"], 2, mode)[1], G["half_16_" + name])


F = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hotpath_rows_f.npz"))


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_oracle_reproduces_golden_rows_f(O):
    """the rows around the path (tests/golden/hotpath_rows_f.npz, made by make_golden_rows_f.py)"""
    img = F["fast_img"]
    assert np.array_equal(O.fast10_score_map(img, 20), F["fast_score_map"])
    x, y, sc, lv = O.fast_detect_grid(O.halfsample_pyramid(img, 3, O.HALF_REF_X86), 25, 20, 20.0, F["fast_occ"])
    assert np.array_equal(lv, F["fast_level"]) and np.array_equal(x, F["fast_x"]) and np.array_equal(y, F["fast_y"])
    assert np.array_equal(_bits(sc), _bits(F["fast_score"]))
    canvas = np.zeros((120, 300, 4), np.float32)
    for t, M in zip(F["pano_tiles"], F["pano_Ms"]):
        O.panorama_add(t, canvas, M, 18)
    assert np.array_equal(_bits(canvas), _bits(F["pano_canvas"])) and np.array_equal(O.normalize_canvas(canvas), F["pano_final"])
    assert tuple(F["pano_bbox0"]) == O.panorama_bbox(F["pano_Ms"][0], 120, 80)
    assert np.array_equal(_bits(O.feather_ramp(3, 60, 18)), _bits(F["pano_ramp"]))
    Hinv, T, M, size = O.stitch_geometry(F["stitch_H"], 120, 80)
    assert size == tuple(F["stitch_size"]) and np.array_equal(_bits(Hinv), _bits(F["stitch_Hinv"])) and np.array_equal(_bits(M), _bits(F["stitch_M"]))
    assert np.array_equal(O.stitch_compose(F["pano_tiles"][0], F["pano_tiles"][1], F["stitch_H"]), F["stitch_out"])
    assert np.array_equal(O.remap(F["pano_tiles"][0], *O.spherical_map(120, 80, 95.5)), F["sph"])
    assert np.array_equal(O.remap(F["pano_tiles"][0], *O.cylindrical_map(120, 80, 95.5)), F["cyl"])
    for k, Fk in enumerate(F["cf_F"]):
        s, inl = O.check_fundamental(Fk, F["cf_p1"], F["cf_p2"], 1.5)
        assert s.view(np.uint32) == F["cf_scores"][k].view(np.uint32) and np.array_equal(inl, F["cf_inliers"][k])
    nxy, nT = O.normalize_points(F["cf_p1"])
    assert np.array_equal(_bits(nxy), _bits(F["norm_xy"])) and np.array_equal(_bits(nT), _bits(F["norm_T"]))
    pa = O.align_params(300.0, 300.0, 119.5, 79.5, max_level=3, min_level=1, n_iter=10, eps=1e-9)
    aT, an, ac, ai = O.sparse_align(O.lk_pyramid(F["al_ref"], 4), O.lk_pyramid(F["al_cur"], 4), F["al_xyz"], pa)
    assert np.array_equal(aT.view(np.uint64), F["al_T"].view(np.uint64)) and an == F["al_n"] and np.array_equal(ai, F["al_iters"])
    assert np.float32(ac).view(np.uint32) == F["al_chi2"].view(np.uint32)


def test_host_geometry_reproduces_golden_rows_f(dr3):
    """the host-only entry points of the C ABI need no GPU"""
    assert dr3.panorama_bbox(F["pano_Ms"][0], 120, 80) == tuple(F["pano_bbox0"])
    assert np.array_equal(_bits(dr3.panorama_ramp(3, 60, 18)), _bits(F["pano_ramp"]))
    Hinv, T, M, size = dr3.stitch_geometry(F["stitch_H"], 120, 80)
    assert size == tuple(F["stitch_size"]) and np.array_equal(_bits(Hinv), _bits(F["stitch_Hinv"]))
    assert np.array_equal(_bits(T), _bits(F["stitch_T"])) and np.array_equal(_bits(M), _bits(F["stitch_M"]))


@pytest.mark.gpu
def test_hip_reproduces_golden_rows_f(dr3, ctx):
    img = F["fast_img"]
    assert np.array_equal(ctx.fast_score_map(img, 20), F["fast_score_map"])
    x, y, sc, lv = ctx.fast_detect(img, 3, 25, 20, 20.0, F["fast_occ"])
    m = F["fast_level"] >= 0
    assert np.array_equal(lv, F["fast_level"]) and np.array_equal(x[m], F["fast_x"][m]) and np.array_equal(y[m], F["fast_y"][m])
    assert np.array_equal(_bits(sc[m]), _bits(F["fast_score"][m]))
    canvas, final = ctx.paste_images(list(F["pano_tiles"]), list(F["pano_Ms"]), (300, 120), 18)
    assert np.array_equal(_bits(canvas), _bits(F["pano_canvas"])) and np.array_equal(final, F["pano_final"])
    assert np.array_equal(ctx.stitch_compose(F["pano_tiles"][0], F["pano_tiles"][1], F["stitch_H"]), F["stitch_out"])
    assert np.array_equal(ctx.warp_spherical(F["pano_tiles"][0], 95.5), F["sph"])
    assert np.array_equal(ctx.warp_cylindrical(F["pano_tiles"][0], 95.5), F["cyl"])
    scores, inl = ctx.check_fundamental(F["cf_F"], F["cf_p1"], F["cf_p2"], 1.5)
    assert np.array_equal(_bits(scores), _bits(F["cf_scores"])) and np.array_equal(inl, F["cf_inliers"])
    nxy, nT = ctx.normalize_points(F["cf_p1"])
    assert np.array_equal(_bits(nxy), _bits(F["norm_xy"])) and np.array_equal(_bits(nT), _bits(F["norm_T"]))
    b = dr3.FrameBatch(ctx, 240, 160, win=(15, 15), max_level=3, capacity=2)
    d_xyz, d_T = ctx.to_device(F["al_xyz"]), ctx.to_device(np.eye(3, 4))
    d_n, d_c, d_i = ctx.alloc(4), ctx.alloc(4), ctx.alloc(32)
    try:
        b.upload(0, np.stack([F["al_ref"], F["al_cur"]])); b.build(0, 2, True)
        b.align(1, 0, d_xyz, len(F["al_xyz"]), dr3.align_params(300.0, 300.0, 119.5, 79.5, max_level=3, min_level=1, n_iter=10, eps=1e-9), d_T, d_n, d_c, d_i)
        ctx.sync()
        assert np.array_equal(d_T.download((3, 4), np.float64).view(np.uint64), F["al_T"].view(np.uint64))
        assert d_n.download((1,), np.int32)[0] == F["al_n"] and np.array_equal(d_i.download((8,), np.int32)[:4], F["al_iters"])
        assert d_c.download((1,), np.float32).view(np.uint32)[0] == F["al_chi2"].view(np.uint32)
    finally:
        for d in (d_xyz, d_T, d_n, d_c, d_i):
            d.free()
        b.close()
