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
