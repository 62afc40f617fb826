"""HIP pyramid + Scharr pass vs the oracle: integer work, bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SIZES = [(1241, 376), (1240, 376), (1920, 1080), (640, 480), (129, 65), (128, 32), (127, 33), (257, 131), (37, 41)]


def _oracle_pyramid(O, img, levels):
    pyr = [img]
    for _ in range(levels - 1):
        pyr.append(O.pyr_down(pyr[-1]))
    return pyr


@pytest.mark.parametrize("w,h", SIZES)
def test_pyramid_and_deriv_bit_exact(dr3, O, ctx, w, h):
    rng = np.random.default_rng(w * 7 + h)
    frames = rng.integers(0, 256, (3, h, w), dtype=np.uint8)
    win = (15, 15) if min(w, h) > 64 else (5, 5)
    b = dr3.FrameBatch(ctx, w, h, win=win, max_level=4, capacity=3)
    try:
        assert b.levels == O.lk_effective_max_level(w, h, win[0], win[1], 4) + 1
        b.upload(0, frames)
        b.build(0, 3, with_deriv=True)
        ctx.sync()
        for s in range(3):
            pyr = _oracle_pyramid(O, frames[s], b.levels)
            for l in range(b.levels):
                assert np.array_equal(b.download_image(s, l), pyr[l]), ("img", s, l)
                assert np.array_equal(b.download_deriv(s, l), O.scharr(pyr[l])), ("der", s, l)
    finally:
        b.close()


def test_build_without_deriv_leaves_deriv_untouched(dr3, O, ctx):
    w, h = 300, 200
    rng = np.random.default_rng(5)
    f = rng.integers(0, 256, (1, h, w), dtype=np.uint8)
    b = dr3.FrameBatch(ctx, w, h, win=(9, 9), max_level=2, capacity=1)
    try:
        b.upload(0, f)
        b.build(0, 1, with_deriv=False)
        ctx.sync()
        assert np.array_equal(b.download_image(0, 2), O.pyr_down(O.pyr_down(f[0])))
        assert not b.download_deriv(0, 0).any()       # zero-initialised, never written
    finally:
        b.close()


def test_extreme_images(dr3, O, ctx):
    """constant / saturated / checkerboard frames: max |derivative| = 16*255 fits int16"""
    w, h = 200, 120
    yy, xx = np.mgrid[0:h, 0:w]
    frames = np.stack([np.full((h, w), 255, np.uint8), np.zeros((h, w), np.uint8),
                       (((xx + yy) & 1) * 255).astype(np.uint8), ((xx > w // 2) * 255).astype(np.uint8)])
    b = dr3.FrameBatch(ctx, w, h, win=(7, 7), max_level=3, capacity=4)
    try:
        b.upload(0, frames)
        b.build(0, 4, True)
        ctx.sync()
        for s in range(4):
            pyr = _oracle_pyramid(O, frames[s], b.levels)
            for l in range(b.levels):
                assert np.array_equal(b.download_image(s, l), pyr[l])
                assert np.array_equal(b.download_deriv(s, l), O.scharr(pyr[l]))
        assert np.abs(b.download_deriv(3, 0)).max() == 16 * 255
    finally:
        b.close()


def test_rccl_ref_pyramid_broadcast_world1(dr3, O, synth):
    """the reference-pyramid broadcast through the C ABI's RCCL communicator.  One GPU per box,
    so world = 1 here: this pins symbol resolution, the id hand-over, group semantics and that the root's
    planes are left byte-identical; the N > 1 protocol is rehearsed with gloo in tests/test_dist.py."""
    A, _, _ = synth.frame_pair(320, 200)
    with dr3.Context(0) as ctx:
        fb = dr3.FrameBatch(ctx, 320, 200, win=(15, 15), max_level=3, capacity=2)
        fb.upload(0, A)
        fb.build(0, 1, True)
        ctx.sync()
        before = [(fb.download_image(0, l), fb.download_deriv(0, l)) for l in range(fb.levels)]
        ctx.comm_init(ctx.comm_unique_id(), 1, 0)
        assert ctx.comm_world() == 1
        moved = fb.bcast_slot(0, root=0, with_deriv=True)
        ctx.sync()
        want = sum(fb.level_info(l).img_pitch * fb.level_info(l).h + fb.level_info(l).der_pitch * fb.level_info(l).h
                   for l in range(fb.levels))
        assert want <= moved < 1.6 * want        # whole planes: levels >= 1 travel with their physical REFLECT_101 border
        for l, (img, der) in enumerate(before):
            assert np.array_equal(fb.download_image(0, l), img) and np.array_equal(fb.download_deriv(0, l), der), l
        intr = ctx.to_device(np.array([718.856, 718.856, 607.1928, 185.2157, 1241, 376], np.float64))
        ctx.bcast(intr, 48, root=0)
        ctx.sync()
        assert intr.download((6,), np.float64)[4] == 1241
        with pytest.raises(dr3.Dr3HipError):
            fb.bcast_slot(5, root=0)            # slot out of range
        ctx.comm_destroy()
        with pytest.raises(dr3.Dr3HipError):
            fb.bcast_slot(0, root=0)            # no communicator
        intr.free(); fb.close()
