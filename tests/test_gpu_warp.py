"""HIP bilinear backward warp / remap / half-sampler vs the oracle.
Bar: 8-bit results bit-exact; f32 results bit-exact too (same operation order, no FMA), which is
tighter than the 1 ULP north_star allows."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rand_H(rng, scale=1.0):
    ang = rng.uniform(-0.3, 0.3)
    c, s = np.cos(ang), np.sin(ang)
    return np.array([[c * rng.uniform(.8, 1.25), -s, rng.uniform(-10, 30) * scale],
                     [s, c * rng.uniform(.8, 1.25), rng.uniform(-10, 30) * scale],
                     [rng.uniform(-1e-3, 1e-3), rng.uniform(-1e-3, 1e-3), 1]], np.float32)


def _eq(a, b):
    if a.dtype == np.float32:
        return np.array_equal(a.view(np.uint32), b.view(np.uint32))
    return np.array_equal(a, b)


@pytest.mark.parametrize("shape,dtype", [((60, 80, 3), np.uint8), ((60, 80, 4), np.float32), ((60, 80), np.uint8),
                                         ((61, 83, 4), np.uint8), ((45, 67), np.float32), ((480, 752, 3), np.uint8)])
def test_warp_perspective_bit_exact(dr3, O, ctx, shape, dtype):
    rng = np.random.default_rng(shape[0] * 31 + shape[1])
    src = (rng.integers(0, 256, shape, dtype=np.uint8) if dtype == np.uint8
           else (rng.random(shape, dtype=np.float32) * 255).astype(np.float32))
    h, w = shape[:2]
    for k in range(6):
        M = _rand_H(rng, w / 80)
        dsize = (int(w * 1.3) + k, int(h * 1.2) + 3 * k)
        assert _eq(ctx.warpPerspective(src, M, dsize), O.warp_perspective(src, M, dsize)), k
        assert _eq(ctx.warpPerspective(src, M, dsize, dr3.WARP_INVERSE_MAP),
                   O.warp_perspective(src, M, dsize, O.WARP_INVERSE_MAP)), k


def test_warp_identity_and_integer_translation(dr3, O, ctx):
    rng = np.random.default_rng(3)
    src = rng.integers(0, 256, (70, 90, 3), dtype=np.uint8)
    assert np.array_equal(ctx.warpPerspective(src, np.eye(3, dtype=np.float32), (90, 70)), src)
    M = np.array([[1, 0, 5], [0, 1, 3], [0, 0, 1]], np.float32)      # src/stitch.cpp:74: pure translation
    d = ctx.warpPerspective(src, M, (100, 80))
    assert np.array_equal(d[3:73, 5:95], src) and not d[:3].any() and not d[:, :5].any()
    # sub-pixel translation (the reference's translate_top_left with fractional min_X/min_Y)
    M = np.array([[1, 0, 5.37], [0, 1, 3.81], [0, 0, 1]], np.float32)
    assert np.array_equal(ctx.warpPerspective(src, M, (100, 80)), O.warp_perspective(src, M, (100, 80)))


def test_warp_pure_translations(dr3, O, ctx):
    """pure translations (the left image of a stitch, src/stitch.cpp:74): offsets on the 1/64-px ties of the 1/32-px grid, negative,
    large, barely fractional and tiny ones; forward and inverse maps; every pixel type"""
    rng = np.random.default_rng(8)
    offs = [(0.0, 0.0), (5.0, 3.0), (1 / 64, 1 / 64), (-3 / 64, 7 / 64), (-17.515625, 40.984375), (1234.5, -0.25), (2.0 ** -20, 0.0),
            (1e-12, 3.0), (100.015625, -60.046875), (0.49999997, 0.5000001)]
    for dtype, shape in ((np.uint8, (61, 83, 3)), (np.uint8, (61, 83)), (np.float32, (33, 45, 4)), (np.uint8, (40, 52, 4))):
        src = rng.integers(0, 256, shape).astype(dtype)
        for tx, ty in offs:
            M = np.array([[1, 0, tx], [0, 1, ty], [0, 0, 1]], np.float32)
            for inverse in (False, True):
                g = ctx.warpPerspective(src, M, (97, 71), flags=dr3.WARP_INVERSE_MAP if inverse else 0)
                o = O.warp_perspective(src, M, (97, 71), flags=O.WARP_INVERSE_MAP if inverse else 0)
                assert np.array_equal(g.view(np.uint8), o.view(np.uint8)), (dtype, shape, tx, ty, inverse)


def test_warp_rounding_ties_and_fast_path_guard(dr3, O, ctx):
    """source coordinates that land exactly on (or a hair beside) the 1/64-px rounding ties of the 1/32-px
    grid: the kernel's fast coordinate path must hand every such pixel to the exact f64 sequence"""
    rng = np.random.default_rng(21)
    src = rng.integers(0, 256, (97, 131, 3), dtype=np.uint8)
    for k in range(1, 64, 6):                                          # odd k: every pixel is an exact tie
        for eps in (0.0, 2.0 ** -20, -2.0 ** -20):
            M = np.array([[1, 0, k / 64 + eps], [0, 1, (64 - k) / 64 - eps], [0, 0, 1]], np.float32)
            for flags in (0, dr3.WARP_INVERSE_MAP):
                assert np.array_equal(ctx.warpPerspective(src, M, (140, 100), flags),
                                      O.warp_perspective(src, M, (140, 100), flags)), (k, eps, flags)
    # ties under a perspective divide: W = 1 + x/1024 exactly representable
    M = np.array([[1, 0, 0.015625], [0, 1, 0.046875], [1 / 1024, 0, 1]], np.float32)
    assert np.array_equal(ctx.warpPerspective(src, M, (256, 100), dr3.WARP_INVERSE_MAP),
                          O.warp_perspective(src, M, (256, 100), O.WARP_INVERSE_MAP))
    gray = src[..., 0].copy()
    M = np.array([[0.5, 0, 0.25], [0, 2, 0.5], [0, 0, 1]], np.float32)
    assert np.array_equal(ctx.warpPerspective(gray, M, (90, 120)), O.warp_perspective(gray, M, (90, 120)))


def test_warp_large_tile_random_homographies(dr3, O, ctx, synth):
    """1500x1100 8UC3 tiles (tens of millions of coordinates, so thousands pass through the tie guard)"""
    rng = np.random.default_rng(22)
    src = rng.integers(0, 256, (1100, 1500, 3), dtype=np.uint8)
    for k in range(3):
        M = synth.tile_homography(k, 1500, 1100)
        M[0, 2] -= 30; M[1, 2] -= 30
        assert np.array_equal(ctx.warpPerspective(src, M, (1500, 1100)), O.warp_perspective(src, M, (1500, 1100))), k


def test_config3_tiles_at_their_own_size_4000x3000(dr3, O, ctx, synth):
    """BASELINE configs[2] at its own size: one 4000x3000 tile as 8UC3 and one as the reference's 32FC4 form
    (src/panorama.cpp:178-192), each through its configs[2] homography (synth.tile_homography, shifted as bench.py shifts it),
    bit-exact against the oracle.  The guarded fast coordinate path is enabled per job from the coordinate magnitudes
    (warp_fast_coords_ok), so the size itself is part of what is checked; 12 Mpix put ~6000 pixels through the tie guard."""
    rng = np.random.default_rng(3)
    TW, TH = 4000, 3000
    low = rng.integers(0, 256, (TH // 4, TW // 4, 3), dtype=np.uint8)
    src = np.repeat(np.repeat(low, 4, axis=0), 4, axis=1)
    src ^= rng.integers(0, 32, (TH, TW, 1), dtype=np.uint8)                       # fine detail on top of the blocks
    for i in (0, 5):
        M = synth.tile_homography(i, TW, TH)
        M[0, 2] -= 60.0; M[1, 2] -= 60.0
        got = ctx.warpPerspective(src, M, (TW, TH))
        want = O.warp_perspective(src, M, (TW, TH))
        assert np.array_equal(got, want), (i, int((got != want).sum()))
        assert (got != 0).mean() > 0.9                                            # the tile really lands inside its canvas
    src4 = np.ones((TH, TW, 4), np.float32)
    src4[..., :3] = src
    M = synth.tile_homography(2, TW, TH)
    M[0, 2] -= 60.0; M[1, 2] -= 60.0
    got = ctx.warpPerspective(src4, M, (TW, TH))
    want = O.warp_perspective(src4, M, (TW, TH))
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_warp_small_and_degenerate(dr3, O, ctx):
    """dst smaller than one 16x64 block, singular-ish / far-outside maps -> zeros like the oracle"""
    rng = np.random.default_rng(4)
    src = rng.integers(0, 256, (20, 24, 3), dtype=np.uint8)
    for dsize in [(5, 3), (63, 15), (65, 17), (1, 1)]:
        M = _rand_H(rng, 0.2)
        assert np.array_equal(ctx.warpPerspective(src, M, dsize), O.warp_perspective(src, M, dsize)), dsize
    M = np.array([[1, 0, 1e6], [0, 1, 0], [0, 0, 1]], np.float32)
    assert not ctx.warpPerspective(src, M, (40, 30)).any()
    M = np.array([[1, 0, 0], [0, 1, 0], [0.05, 0.01, 1]], np.float32)  # W crosses 0 inside the image
    assert np.array_equal(ctx.warpPerspective(src, M, (80, 60)), O.warp_perspective(src, M, (80, 60)))


def test_panorama_style_32fc4(dr3, O, synth, ctx):
    """the src/panorama.cpp:178-192 form: (B,G,R,1.0) f32 image warped onto a larger canvas"""
    tile = synth.panorama_tile(200, 150)
    img = np.ones((150, 200, 4), np.float32)
    img[..., :3] = tile.astype(np.float32)
    M = synth.tile_homography(1, 200, 150)
    got = ctx.warpPerspective(img, M, (360, 300))
    want = O.warp_perspective(img, M, (360, 300))
    assert _eq(got, want)
    assert got[..., 3].max() == 1.0


@pytest.mark.parametrize("shape,dtype", [((60, 80, 3), np.uint8), ((60, 80, 4), np.float32), ((50, 70), np.uint8)])
def test_remap_bit_exact(dr3, O, ctx, shape, dtype):
    rng = np.random.default_rng(11)
    src = (rng.integers(0, 256, shape, dtype=np.uint8) if dtype == np.uint8 else rng.random(shape, dtype=np.float32))
    h, w = shape[:2]
    mx = rng.uniform(-3, w + 3, (57, 93)).astype(np.float32)
    my = rng.uniform(-3, h + 3, (57, 93)).astype(np.float32)
    mx[0, :8] = [0, 0.5, w - 1, w - 0.5, -1, -0.5, 1e9, -1e9]          # ties and saturating coordinates
    # non-finite and out-of-int-range map entries: cvRound is cvtss2si on x86-64, whose result for them is 0x80000000
    # (far outside on the left -> border 0), not the saturated value of the GPU's own conversion
    mx[1, :6] = [np.inf, -np.inf, np.nan, 3e9, -3e9, 7e7]
    my[2, :6] = [np.inf, -np.inf, np.nan, 3e9, -3e9, 7e7]
    assert _eq(ctx.remap(src, mx, my), O.remap(src, mx, my))


def test_remap_spherical_maps(dr3, O, ctx, synth):
    """utils::warp_spherical (src/utils.cpp:196-201): maps from the oracle's map generator"""
    img = synth.panorama_tile(160, 120)
    u, v = O.spherical_map(160, 120, 140.0)
    assert np.array_equal(ctx.remap(img, u, v), O.remap(img, u, v))


@pytest.mark.parametrize("w,h", [(1241, 376), (1240, 376), (1920, 1080), (64, 48), (33, 17), (101, 376)])
@pytest.mark.parametrize("mode", ["HALF_TRUNC", "HALF_SSE2", "HALF_REF_SCALAR", "HALF_REF_X86"])
def test_halfsample_pyramid_modes(dr3, O, ctx, w, h, mode):
    rng = np.random.default_rng(w + h)
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    n_levels = 3 if min(w, h) < 100 else 5
    got = ctx.create_img_pyramid(img, n_levels, getattr(dr3, mode))
    want = O.halfsample_pyramid(img, n_levels, getattr(O, mode))
    assert len(got) == n_levels
    for l in range(n_levels):
        assert got[l].shape == want[l].shape == (h >> l, w >> l)
        assert np.array_equal(got[l], want[l]), (mode, l)


@pytest.mark.parametrize("kind", ["spherical", "cylindrical"])
@pytest.mark.parametrize("shape,f", [((120, 160, 3), 140.0), ((97, 131, 3), 75.5), ((200, 300), 410.25), ((64, 48, 4), 33.0)])
def test_fused_map_generation_matches_maps_then_remap(dr3, O, ctx, synth, kind, shape, f):
    """utils::warp_spherical / warp_cylindrical (src/utils.cpp:125-272): the kernel that generates (u, v) itself
    equals the oracle's map generator followed by the oracle's remap, bit for bit"""
    rng = np.random.default_rng(shape[0] + int(f))
    img = rng.integers(0, 256, shape, dtype=np.uint8) if len(shape) == 2 or shape[2] == 3 else rng.random(shape, dtype=np.float32)
    h, w = shape[:2]
    u, v = (O.spherical_map if kind == "spherical" else O.cylindrical_map)(w, h, f)
    want = O.remap(img, u, v)
    got = (ctx.warp_spherical if kind == "spherical" else ctx.warp_cylindrical)(img, f)
    assert _eq(got, want)
    assert np.array_equal(ctx.remap(img, u, v) if img.dtype == np.uint8 else got, got)      # and the map-driven kernel agrees
    assert got.any()


@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
def test_warp_batch_heterogeneous_jobs_one_launch(dr3, O, ctx, dtype):
    """dr3hip_warp_perspective_batch (what the bench's configs[2] leg calls): jobs of different source and destination sizes
    in ONE launch -- the grid is sized for the largest job, the others are bounded per job -- including a destination lower
    than 16 rows (generic block origin), a map whose W changes sign (exact coordinate path) and a pitched destination"""
    rng = np.random.default_rng(77)
    typ, cn = (dr3.T_8UC3, 3) if dtype == np.uint8 else (dr3.T_32FC4, 4)
    esz = cn * np.dtype(dtype).itemsize
    specs = [((60, 80), (104, 72)), ((200, 150), (90, 210)), ((33, 47), (300, 11)), ((120, 90), (64, 64)), ((90, 40), (257, 130))]
    jobs, keep, want = [], [], []
    for i, ((sw, sh), (dw, dh)) in enumerate(specs):
        src = rng.integers(0, 256, (sh, sw, cn), dtype=np.uint8) if dtype == np.uint8 else (rng.random((sh, sw, cn), dtype=np.float32) * 255).astype(np.float32)
        M = _rand_H(rng, sw / 80)
        if i == 3:
            M[2] = [0.02, 0.01, 1.0]                                    # W crosses 0 inside the destination
        dpitch = (dw * esz + 255) // 256 * 256
        d_src, d_dst = ctx.to_device(src), ctx.alloc(dpitch * dh)
        d_dst.zero()
        keep += [d_src, d_dst]
        jobs.append(dict(src=d_src.ptr, sw=sw, sh=sh, sstride=sw * esz, dst=d_dst.ptr, dw=dw, dh=dh, dstride=dpitch, M=M))
        want.append(O.warp_perspective(src, M, (dw, dh)))
    try:
        dr3.warp_batch(ctx, jobs, typ)
        ctx.sync()
        for i, ((sw, sh), (dw, dh)) in enumerate(specs):
            dpitch = jobs[i]["dstride"]
            raw = keep[2 * i + 1].download((dh, dpitch), np.uint8)
            got = np.ascontiguousarray(raw[:, :dw * esz]).view(dtype).reshape(dh, dw, cn)
            assert _eq(got, want[i]), i
            assert not raw[:, dw * esz:].any()                             # nothing written past a job's own width
    finally:
        for d in keep:
            d.free()


def _near_affine(rng, w, h, angle=0.06, persp=1.5e-5):
    ang = rng.uniform(-angle, angle)
    c, s = np.cos(ang), np.sin(ang)
    sx, sy = rng.uniform(0.93, 1.08, 2)
    return np.array([[c * sx, -s, rng.uniform(-0.2, 0.2) * w], [s, c * sy, rng.uniform(-0.2, 0.2) * h],
                     [rng.uniform(-persp, persp), rng.uniform(-persp, persp), 1]], np.float32)


def _warp_dev(dr3, ctx, src, M, dsize, spitch=None, dpitch=None, flags=0):
    """8UC3 warp through dr3hip_warp_perspective_batch with explicit device pitches; returns (result, plan)"""
    h, w = src.shape[:2]
    dw, dh = dsize
    spitch = spitch or (3 * w + 15) // 16 * 16
    dpitch = dpitch or (3 * dw + 63) // 64 * 64
    hs = np.zeros((h, spitch), np.uint8); hs[:, :3 * w] = src.reshape(h, 3 * w)
    d_src, d_dst = ctx.to_device(hs), ctx.to_device(np.full((dh, dpitch), 0xA5, np.uint8))
    job = dict(src=d_src.ptr, sw=w, sh=h, sstride=spitch, dst=d_dst.ptr, dw=dw, dh=dh, dstride=dpitch, M=M)
    plan = dr3.warp_plan(job, dr3.T_8UC3, flags)
    dr3.warp_batch(ctx, [job], dr3.T_8UC3, flags)
    raw = d_dst.download((dh, dpitch), np.uint8)
    d_src.free(); d_dst.free()
    assert (raw[:, 3 * dw:] == 0xA5).all()                      # nothing written behind a destination row
    return raw[:, :3 * dw].reshape(dh, dw, 3), plan


@pytest.mark.parametrize("shape", [(376, 1241), (480, 752), (61, 83), (200, 64), (17, 300), (130, 257)])
def test_strip_kernel_nearly_affine_maps(dr3, O, ctx, shape):
    """the LDS-staged strip kernel is the one that runs nearly affine 8UC3 jobs: boxes inside the image, boxes that stick out
    on every side (border constant 0), destinations that end inside a 4-pixel group / a 16-row strip, sources wholly outside"""
    rng = np.random.default_rng(shape[0] * 7 + shape[1])
    h, w = shape
    src = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    for k in range(8):
        M = _near_affine(rng, w, h)
        if k == 6:
            M[0, 2] += 3.0 * w                                   # the source is nowhere in the destination
        dsize = (w + (k * 7) % 13, h + (k * 5) % 11) if k % 2 else (max(8, w - 3 * k), max(8, h - 2 * k))
        got, plan = _warp_dev(dr3, ctx, src, M, dsize)
        if dsize[1] >= 16:                                       # shorter destinations: OpenCV's block split is not a power of two there
            assert plan["kernel"] == 2 and plan["tie_guard"] in (1, 2), (k, plan)
        assert np.array_equal(got, O.warp_perspective(src, M, dsize)), (k, plan)
        got, plan = _warp_dev(dr3, ctx, src, M, dsize, flags=dr3.WARP_INVERSE_MAP)
        assert np.array_equal(got, O.warp_perspective(src, M, dsize, O.WARP_INVERSE_MAP)), (k, plan)


def test_strip_kernel_store_paths_and_eligibility(dr3, O, ctx):
    """destination rows that are not dword-aligned take the strip kernel's byte stores; source rows that are not 16-byte aligned and maps
    that are not nearly affine go to the direct kernel -- same bytes either way"""
    rng = np.random.default_rng(77)
    src = rng.integers(0, 256, (150, 210, 3), dtype=np.uint8)
    M = _near_affine(rng, 210, 150)
    want = O.warp_perspective(src, M, (203, 141))
    got, plan = _warp_dev(dr3, ctx, src, M, (203, 141), dpitch=3 * 203 + 2)         # odd destination pitch
    assert plan["kernel"] == 2 and np.array_equal(got, want)
    got, plan = _warp_dev(dr3, ctx, src, M, (203, 141), spitch=3 * 210 + 6)          # 4-byte aligned source rows only
    assert plan["kernel"] == 0 and np.array_equal(got, want)
    Mp = M.copy(); Mp[2, 1] = 6e-4                                                   # |M7 / W| above the strip kernel's bound
    got, plan = _warp_dev(dr3, ctx, src, Mp, (203, 141))
    assert plan["kernel"] == 0 and np.array_equal(got, O.warp_perspective(src, Mp, (203, 141)))
    Mz = M.copy(); Mz[0, 0] *= 3.2                                                   # strong minification along x: still one strip per LDS slice?
    got, plan = _warp_dev(dr3, ctx, src, Mz, (203, 141))
    assert np.array_equal(got, O.warp_perspective(src, Mz, (203, 141))), plan


def test_strip_kernel_every_pixel_on_a_tie(dr3, O, ctx):
    """translations by k / 64 px and a shear whose columns land on 1/64-px ties: every pixel of every strip goes through the tie list,
    overflows it and is finished by its own lane"""
    rng = np.random.default_rng(5)
    src = rng.integers(0, 256, (100, 200, 3), dtype=np.uint8)
    for tx, ty in ((3 / 64, 0.0), (-5 / 64, 9 / 64), (17.515625, -4.984375)):
        M = np.array([[1, 0, tx], [0, 1, ty], [0, 0, 1]], np.float32)
        got, plan = _warp_dev(dr3, ctx, src, M, (200, 100))
        assert plan["kernel"] == 2 and np.array_equal(got, O.warp_perspective(src, M, (200, 100))), (tx, ty)
    M = np.array([[1, 1 / 32, 1 / 64], [0, 1, 0.25], [0, 0, 1]], np.float32)            # X moves by 1/32 px per row: a tie on every row
    got, plan = _warp_dev(dr3, ctx, src, M, (200, 100))
    assert plan["kernel"] == 2 and np.array_equal(got, O.warp_perspective(src, M, (200, 100)))


@pytest.mark.parametrize("w,h,spitch,dpitch", [(2064, 37, 2064, 1032), (4100, 9, 4112, 2056), (1026, 5, 1040, 520), (3000, 6, 3008, 1500)])
@pytest.mark.parametrize("mode", ["HALF_TRUNC", "HALF_SSE2"])
def test_halfsample_streaming_form_equals_the_plain_one(dr3, O, ctx, w, h, spitch, dpitch, mode):
    """rows at a 16-byte aligned pitch and 512 or more output pixels take halfsample_wide_kernel (1024 output pixels per wave, 16-byte
    loads); the same image at an odd pitch takes the plain kernel: both equal the oracle -- row tails inside a segment, between the two
    segments of a lane and between workgroups"""
    rng = np.random.default_rng(w)
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    img[:, ::7] = 255; img[1::2, 3::5] = 0                       # rounding ties of both averages
    want = O.halfsample(img, getattr(O, mode))
    for sp, dp in ((spitch, dpitch), (spitch + 3, dpitch + 1)):
        host = np.zeros((h, sp), np.uint8); host[:, :w] = img
        d_src, d_dst = ctx.to_device(host), ctx.alloc(dp * (h // 2))
        try:
            d_dst.zero()
            ctx._chk(dr3.lib().dr3hip_halfsample_dev(ctx.h, d_src.ptr, w, h, sp, d_dst.ptr, dp, int(getattr(dr3, mode))))
            got = d_dst.download((h // 2, dp), np.uint8)
        finally:
            d_src.free(); d_dst.free()
        assert np.array_equal(got[:, :w // 2], want), (sp, dp)
        assert not got[:, w // 2:].any()                          # nothing written behind a row's last pixel
