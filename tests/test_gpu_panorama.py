"""Fused panorama paste on the GPU vs the oracle's three-pass composition of reference src/panorama.cpp:156-229
(32FC4 copy -> warpPerspective -> feather + accumulate; normalise).  Bar: the f32 canvas and the u8 panorama
are bit-exact (same operation order, no FMA)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _scene(synth, n, w, h, rng):
    imgs, Ms = [], []
    for i in range(n):
        img = synth.panorama_tile(w, h, seed=0x3D52 + i)
        img = np.roll(img, 17 * i, axis=1).copy()
        a = rng.uniform(-0.06, 0.06)
        M = np.array([[np.cos(a), -np.sin(a), 20 + i * (w * 0.55)],
                      [np.sin(a), np.cos(a), 15 + rng.uniform(-8, 8)],
                      [rng.uniform(-4e-5, 4e-5), rng.uniform(-4e-5, 4e-5), 1]], np.float32)
        imgs.append(img); Ms.append(M)
    return imgs, Ms


@pytest.mark.parametrize("w,h,n,fw", [(160, 120, 3, 20), (200, 150, 2, 150), (97, 61, 4, 7)])
def test_paste_and_normalize_bit_exact(dr3, O, ctx, synth, w, h, n, fw):
    rng = np.random.default_rng(w + n)
    imgs, Ms = _scene(synth, n, w, h, rng)
    cw, ch = int(40 + w * (1 + 0.55 * (n - 1)) + 30), h + 50
    want = np.zeros((ch, cw, 4), np.float32)
    for img, M in zip(imgs, Ms):
        O.panorama_add(img, want, M, fw)
    got_canvas, got_pan = ctx.paste_images(imgs, Ms, (cw, ch), fw)
    assert np.array_equal(_bits(got_canvas), _bits(want))
    assert np.array_equal(got_pan, O.normalize_canvas(want))
    assert got_canvas[..., 3].max() >= 1.0
    assert got_pan.any()


def test_paste_clips_boxes_that_leave_the_canvas(dr3, O, ctx, synth):
    """bounding boxes partly left of / right of the canvas: the reference indexes unchecked; the oracle and the
    kernel both skip those columns"""
    img = synth.panorama_tile(120, 90)
    for tx in (-40.5, 130.25):
        M = np.array([[1, 0, tx], [0, 1, 6.5], [1e-5, 0, 1]], np.float32)
        want = np.zeros((110, 200, 4), np.float32)
        O.panorama_add(img, want, M, 16)
        got, _ = ctx.paste_images([img], [M], (200, 110), 16)
        assert np.array_equal(_bits(got), _bits(want)), tx


def test_zero_pixels_carry_no_weight(dr3, O, ctx):
    """:205-207: a source pixel that is exactly black contributes neither colour nor alpha"""
    img = np.full((40, 60, 3), 90, np.uint8)
    img[10:20, 20:40] = 0
    M = np.array([[1, 0, 8], [0, 1, 4], [0, 0, 1]], np.float32)
    want = np.zeros((50, 80, 4), np.float32)
    O.panorama_add(img, want, M, 5)
    got, pan = ctx.paste_images([img], [M], (80, 50), 5)
    assert np.array_equal(_bits(got), _bits(want))
    assert got[16, 35, 3] == 0.0 and got[6, 35, 3] == 1.0 and pan[16, 35].tolist() == [0, 0, 0]


def test_normalize_truncation_order(dr3, O, ctx):
    """(int)canvas / alpha: the int cast comes before the divide"""
    rng = np.random.default_rng(9)
    canvas = (rng.random((33, 70, 4), dtype=np.float32) * np.float32(600)).astype(np.float32)
    canvas[..., 3] = rng.random((33, 70), dtype=np.float32) * 3
    canvas[5, :, 3] = 0.0
    canvas[6, :, 3] = -1.0
    d_c = ctx.to_device(canvas); d_o = ctx.alloc(33 * 70 * 3)
    ctx.panorama_normalize(d_c, 70, 33, 70 * 16, d_o, 70 * 3)
    got = d_o.download((33, 70, 3), np.uint8)
    d_c.free(); d_o.free()
    assert np.array_equal(got, O.normalize_canvas(canvas))
    assert not got[5].any() and not got[6].any()


def test_bad_arguments_return_errors(dr3, ctx):
    """error behaviour of the new entry points: negative codes + text, nothing thrown across the ABI"""
    img = np.zeros((20, 30, 3), np.uint8)
    d_img, d_c, d_o = ctx.to_device(img), ctx.alloc(40 * 50 * 16), ctx.alloc(40 * 50 * 3)
    try:
        M = np.eye(3, dtype=np.float32)
        with pytest.raises(dr3.Dr3HipError):
            ctx.panorama_add(d_img, 30, 20, 10, d_c, 50, 40, 50 * 16, M, 5)              # image stride < 3 w
        with pytest.raises(dr3.Dr3HipError):
            ctx.panorama_add(d_img, 30, 20, 90, d_c, 50, 40, 50 * 16 + 4, M, 5)          # canvas rows not 16-byte aligned
        with pytest.raises(dr3.Dr3HipError):
            ctx.panorama_add(d_img, 1, 20, 90, d_c, 50, 40, 50 * 16, M, 5)               # 1-column box: the reference throws
        with pytest.raises(dr3.Dr3HipError):
            ctx.panorama_normalize(d_c, 50, 40, 50 * 16, d_o, 100)                       # output stride < 3 w
        with pytest.raises(dr3.Dr3HipError):
            ctx._warp_local(img, 7, 100.0)                                               # unknown map kind
        with pytest.raises(dr3.Dr3HipError):
            ctx.fast_detect(np.zeros((4, 4), np.uint8))                                  # smaller than the FAST ring
        with pytest.raises(dr3.Dr3HipError):
            ctx.fast_detect(np.zeros((64, 64), np.uint8), fast_threshold=0)
        assert ctx.fast_detect(np.zeros((64, 64), np.uint8))[3].max() == -1             # flat image: no feature at all
    finally:
        for d in (d_img, d_c, d_o):
            d.free()


def test_canvas_slabs_of_several_gpus_equal_the_whole_canvas(dr3, O, ctx, synth):
    """SURVEY 8(e), warp side: the canvas is sharded by rows, every GPU pastes all images into its own slab, nothing is
    reduced.  Three slabs (as three GPUs would hold them) put together equal the single canvas bit for bit."""
    rng = np.random.default_rng(12)
    imgs, Ms = _scene(synth, 3, 160, 120, rng)
    cw, ch = 460, 170
    want = np.zeros((ch, cw, 4), np.float32)
    for img, M in zip(imgs, Ms):
        O.panorama_add(img, want, M, 25)
    bounds = [0, 57, 58, ch]                                   # uneven slabs, one of a single row
    got = np.zeros_like(want)
    for y0, y1 in zip(bounds[:-1], bounds[1:]):
        d_slab = ctx.alloc((y1 - y0) * cw * 16); d_slab.zero()
        for img, M in zip(imgs, Ms):
            d_img = ctx.to_device(img)
            ctx.panorama_add_slab(d_img, 160, 120, 480, d_slab, cw, ch, cw * 16, M, 25, y0, y1)
            ctx.sync(); d_img.free()
        got[y0:y1] = d_slab.download((y1 - y0, cw, 4), np.float32)
        d_slab.free()
    assert np.array_equal(_bits(got), _bits(want)) and got[..., 3].max() >= 1.0


def _paste_dev(ctx, img, M, canvas_size, fw, pad=0):
    """one paste through the device entry point with an explicit row pitch (pad bytes beyond 3 w): -> canvas (H, W, 4) f32"""
    cw, ch = canvas_size
    h, w = img.shape[:2]
    pitch = 3 * w + pad
    host = np.zeros((h, pitch), np.uint8); host[:, :3 * w] = img.reshape(h, 3 * w)
    d_img, d_c = ctx.to_device(host), ctx.alloc(cw * ch * 16)
    try:
        d_c.zero()
        ctx.panorama_add(d_img, w, h, pitch, d_c, cw, ch, cw * 16, M, fw)
        ctx.sync()
        return d_c.download((ch, cw, 4), np.float32)
    finally:
        d_img.free(); d_c.free()


@pytest.mark.parametrize("w,h", [(400, 300), (336, 210)])
def test_strip_form_and_direct_form_paste_the_same_canvas(dr3, O, ctx, synth, w, h):
    """16-byte aligned image rows and a nearly affine map take the strip form (paste_strip_kernel: the box of the image through the
    LDS-staged strips of the 8UC3 warp), anything else the direct one (panorama_add_kernel); both equal the oracle bit for bit --
    several tiles, box edges off the 64 x 16 grid, a perspective row, boxes cut by every canvas edge"""
    rng = np.random.default_rng(w)
    img = synth.panorama_tile(w, h, seed=0x3D52 + w)
    img[h // 3: h // 3 + 9, w // 4: w // 4 + 40] = 0                          # exactly black pixels carry no weight (:205-207)
    cases = []
    for k in range(5):
        a = rng.uniform(-0.05, 0.05)
        cases.append(np.array([[np.cos(a), -np.sin(a), 37.3 + 11 * k], [np.sin(a), np.cos(a), 21.7 + 5 * k],
                               [rng.uniform(-3e-5, 3e-5), rng.uniform(-3e-5, 3e-5), 1]], np.float32))
    cases += [np.array([[1, 0, -150.25], [0, 1, -70.5], [1e-5, 0, 1]], np.float32),          # cut by the left and top edges
              np.array([[1, 0, w * 0.8], [0, 1, h * 0.7], [0, 1e-5, 1]], np.float32)]         # ... by the right and bottom ones
    size = (w + 180, h + 130)
    assert (3 * w) % 16 == 0
    took = set()
    for M in cases:
        want = np.zeros((size[1], size[0], 4), np.float32)
        O.panorama_add(img, want, M, 40)
        k_strip, coords, box = dr3.panorama_plan(w, h, size, M, istride=3 * w)
        k_direct, _, _ = dr3.panorama_plan(w, h, size, M, istride=3 * w + 4)
        assert k_direct == 0 and 0 <= box[0] < box[2] <= size[0] and 0 <= box[1] < box[3] <= size[1]
        took.add(k_strip)
        assert np.array_equal(_bits(_paste_dev(ctx, img, M, size, 40)), _bits(want)), ("strips" if k_strip == 2 else "direct", M)
        assert np.array_equal(_bits(_paste_dev(ctx, img, M, size, 40, pad=4)), _bits(want)), ("direct", M)
    assert 2 in took                                                           # the strip form did run


def test_strip_paste_on_rounding_ties(dr3, O, ctx, synth):
    """a translation by an odd multiple of 1/64 px puts EVERY pixel on a rounding tie of the 1/32-px grid: the strip form pastes such
    pixels once, by OpenCV's own sequence, from its tie list (60 entries per strip) and, beyond it, from each lane's own bit mask"""
    img = synth.panorama_tile(208, 96, seed=7)
    for tx, ty in ((10.015625, 5.015625), (33.484375, 2.25), (7.5, 9.046875)):
        M = np.array([[1, 0, tx], [0, 1, ty], [0, 0, 1]], np.float32)
        want = np.zeros((140, 280, 4), np.float32)
        O.panorama_add(img, want, M, 30)
        assert dr3.panorama_plan(208, 96, (280, 140), M, istride=624)[0] == 2
        assert np.array_equal(_bits(_paste_dev(ctx, img, M, (280, 140), 30)), _bits(want)), (tx, ty)


def test_paste_of_a_tile_onto_a_canvas_wider_than_the_fast_coordinate_range(dr3, O, ctx, synth):
    """the canvas of a panorama is wider than the +-8000 source pixels the fast coordinate forms cover; the plan is made over the
    image's box, so the tile far right on a 12 000-px canvas still takes the strips (and the rows outside its box are not walked)"""
    w, h = 320, 200
    img = synth.panorama_tile(w, h, seed=11)
    M = np.array([[0.9995, -0.03, 11500.4], [0.03, 0.9995, 40.2], [2e-6, 1e-6, 1]], np.float32)
    size = (12000, 330)
    k, coords, box = dr3.panorama_plan(w, h, size, M, istride=3 * w)
    assert k == 2 and coords == 3 and box[0] > 11000 and box[3] - box[1] < 330
    want = np.zeros((size[1], size[0], 4), np.float32)
    O.panorama_add(img, want, M, 25)
    assert np.array_equal(_bits(_paste_dev(ctx, img, M, size, 25)), _bits(want))
