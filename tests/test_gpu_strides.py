"""The C ABI takes images with a row stride (cv::Mat::step): a region of interest of a larger buffer must give what the packed
copy gives.  The Python mirror always hands over contiguous arrays, so these tests call the entry points directly with views."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _roi(rng, shape, dtype=np.uint8, pad=(5, 23)):
    """a non-contiguous view of the wanted shape inside a larger random buffer"""
    h, w = shape[:2]
    big = rng.integers(0, 256, (h + 2 * pad[0], w + 2 * pad[1]) + tuple(shape[2:]), dtype=np.uint8)
    if dtype == np.float32:
        big = big.astype(np.float32)
    v = big[pad[0]:pad[0] + h, pad[1]:pad[1] + w]
    assert not v.flags.c_contiguous
    return big, v


def test_strided_images_through_the_c_abi(dr3, O, ctx, synth):
    L = dr3.lib()
    rng = np.random.default_rng(8)
    # ---- LK on two regions of interest of one wide buffer
    A, B, T = synth.frame_pair(300, 200)
    wide = np.zeros((200, 1024), np.uint8); wide[:, 17:317] = A; wide[:, 517:817] = B
    va, vb = wide[:, 17:317], wide[:, 517:817]
    pts = synth.grid_points(300, 200, 24)
    n = len(pts)
    p = dr3.lk_params(win=(15, 15), max_level=2, max_iters=30, eps=0.01)
    out = pts.copy(); st = np.zeros(n, np.uint8); err = np.zeros(n, np.float32)
    rc = L.dr3hip_lk_track(ctx.h, va.ctypes.data, vb.ctypes.data, 300, 200, va.strides[0], vb.strides[0], pts.ctypes.data, out.ctypes.data, n,
                           st.ctypes.data, err.ctypes.data, C.byref(p))
    assert rc == 2
    c_xy, c_st, c_err, _ = ctx.calcOpticalFlowPyrLK(A, B, pts, None, p)
    assert np.array_equal(out, c_xy) and np.array_equal(st, c_st) and np.array_equal(err, c_err)
    # ---- warpPerspective: strided source AND strided destination (the bytes between destination rows must stay untouched)
    for dtype, typ in ((np.uint8, dr3.T_8UC3), (np.float32, dr3.T_32FC4)):
        cn = 3 if dtype == np.uint8 else 4
        big, src = _roi(rng, (90, 120, cn), dtype)
        M = synth.tile_homography(2, 120, 90)
        dbig = np.full((110, 200, cn), 77, dtype); dst = dbig[4:104, 10:150]
        Mc = np.ascontiguousarray(M, np.float32).reshape(9)
        assert L.dr3hip_warp_perspective(ctx.h, src.ctypes.data, 120, 90, src.strides[0], typ, dst.ctypes.data, 140, 100, dst.strides[0], Mc.ctypes.data, 0) == 0
        want = O.warp_perspective(np.ascontiguousarray(src), M, (140, 100))
        assert np.array_equal(np.ascontiguousarray(dst).view(np.uint8), want.view(np.uint8))
        guard = dbig.copy(); guard[4:104, 10:150] = 77
        assert (guard == 77).all()
    # ---- FAST detector and half-sample pyramid on a region of interest: the reference's odd-width walk depends on the step
    big, img = _roi(rng, (121, 203))
    img[...] = np.clip(synth.frame_pair(203, 121)[0].astype(int) + rng.integers(-50, 50, (121, 203)) * (rng.random((121, 203)) < 0.2), 0, 255)
    gc, gr = -(-203 // 25), -(-121 // 25)
    cells = (dr3.Corner * (gc * gr))()
    assert L.dr3hip_fast_detect(ctx.h, img.ctypes.data, 203, 121, img.strides[0], 2, 25, 20, 20.0, None, dr3.HALF_TRUNC, cells) >= 0
    a = np.frombuffer(cells, dtype=np.dtype([("x", np.int32), ("y", np.int32), ("level", np.int32), ("score", np.float32)]))
    ox, oy, osc, ol = O.fast_detect_grid(O.halfsample_pyramid(np.ascontiguousarray(img), 2, O.HALF_TRUNC), 25, 20, 20.0)
    m = ol >= 0
    assert np.array_equal(a["level"], ol) and np.array_equal(a["x"][m], ox[m]) and np.array_equal(a["score"][m].view(np.uint32), osc[m].view(np.uint32))
    lv1 = np.zeros((60, 101), np.uint8)
    ptrs = (C.c_void_p * 2)(img.ctypes.data, lv1.ctypes.data); strides = (C.c_int * 2)(img.strides[0], lv1.strides[0])
    for mode, omode in ((dr3.HALF_TRUNC, O.HALF_TRUNC), (dr3.HALF_REF_SCALAR, O.HALF_REF_SCALAR)):
        assert L.dr3hip_halfsample_pyramid(ctx.h, img.ctypes.data, 203, 121, img.strides[0], 2, mode, ptrs, strides) == 0
        want = np.empty((60, 101), np.uint8)
        O.lib().dr3o_halfsample_u8(img.ctypes.data, 203, 121, img.strides[0], want.ctypes.data, 101, omode)   # the oracle walks the same strided view
        assert np.array_equal(lv1, want), mode
