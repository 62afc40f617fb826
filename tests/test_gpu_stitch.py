"""The fused stitch (reference src/stitch.cpp:33-80: warpPerspective(right, T * Hinv), warpPerspective(left, T), addWeighted) against the
oracle's three passes.  Since round 4 the fused kernel has a STRIP form -- the right image through the LDS-staged strip machinery, the
left image as the exact translation translate_top_left_mat is (one tap offset and one pair of fractions for the whole canvas, horizontal
sums shared between rows), the reference's 0.5 / 0.5 / 0 blend as a byte average -- and the direct form for everything else.  Both must be
bit-identical to the oracle; dr3hip_stitch_plan says which one a call takes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _H(tx, ty, rot_deg=1.5, px=1e-5, py=-6e-6, s=1.0):
    a = np.deg2rad(rot_deg)
    return np.array([[s * np.cos(a), -s * np.sin(a), tx], [s * np.sin(a), s * np.cos(a), ty], [px, py, 1.0]], np.float32)


@pytest.mark.parametrize("w,h,tx,ty", [(640, 400, -260.0, 21.0), (517, 333, -200.37, -17.81), (1200, 210, -512.5, 8.015625), (130, 70, -40.25, 3.5),
                                        (700, 500, 31.3, -44.9)])
def test_strip_form_equals_the_oracle(dr3, O, ctx, synth, w, h, tx, ty):
    left = synth.panorama_tile(w, h)
    right = np.roll(left, -int(abs(tx)), axis=1)
    H = _H(tx, ty)
    Hinv, T, M, size = dr3.stitch_geometry(H, w, h)
    assert dr3.stitch_plan(w, h, w, h, T, M, size) == 2                  # the strip form is what runs
    got = ctx.stitch_compose(left, right, H)
    want = O.stitch_compose(left, right, H)
    assert got.shape == want.shape and np.array_equal(got, want)


def test_translation_fractions_cover_the_table(dr3, O, ctx, synth):
    """the left image's constant fractions: every (fx, fy) class incl. 0, ties of the 1/64 grid (round half to even) and negatives"""
    w, h = 200, 96
    left = synth.panorama_tile(w, h); right = left[::-1].copy()
    for k, (tx, ty) in enumerate([(-50.0, 10.0), (-50.015625, 10.015625), (-50.046875, 9.984375), (-49.5, 10.96875), (-50.99, 11.01), (-3.3333, 0.7777),
                                  (-64.515625, 0.484375)]):
        H = _H(tx, ty, rot_deg=0.7)
        Hinv, T, M, size = dr3.stitch_geometry(H, w, h)
        assert dr3.stitch_plan(w, h, w, h, T, M, size) == 2, k
        assert np.array_equal(ctx.stitch_compose(left, right, H), O.stitch_compose(left, right, H)), (k, tx, ty)


def test_general_weights_take_the_float_blend(dr3, O, ctx, synth):
    w, h = 320, 200
    left = synth.panorama_tile(w, h); right = np.roll(left, 37, axis=0)
    H = _H(-120.4, 9.3)
    for al, be, ga in ((0.5, 0.5, 0.0), (0.3, 0.7, 0.0), (0.25, 0.5, 10.0), (1.0, 1.0, -20.0), (0.5, 0.5, 0.5)):
        got = ctx.stitch_compose(left, right, H, al, be, ga)
        assert np.array_equal(got, O.stitch_compose(left, right, H, al, be, ga)), (al, be, ga)


def test_left_and_right_of_different_sizes(dr3, O, ctx, synth):
    left = synth.panorama_tile(300, 180); right = synth.panorama_tile(410, 260)[:, ::-1].copy()
    H = _H(-150.2, -30.6, rot_deg=-2.0)
    Hinv, T, M, size = dr3.stitch_geometry(H, 410, 260)
    assert dr3.stitch_plan(300, 180, 410, 260, T, M, size) == 2
    assert np.array_equal(ctx.stitch_compose(left, right, H), O.stitch_compose(left, right, H))


def test_maps_the_strip_form_cannot_take_fall_back_to_the_direct_form(dr3, O, ctx, synth):
    w, h = 256, 160
    left = synth.panorama_tile(w, h); right = left.copy()
    H = _H(-90.0, 5.0, rot_deg=3.0, px=6e-4, py=4e-4)                     # strongly projective: not the strip kernel's nearly affine class
    Hinv, T, M, size = dr3.stitch_geometry(H, w, h)
    assert dr3.stitch_plan(w, h, w, h, T, M, size) == 0
    assert np.array_equal(ctx.stitch_compose(left, right, H), O.stitch_compose(left, right, H))
    # unaligned rows (a device-pointer caller with a 3 * w pitch that is not a multiple of 16)
    assert dr3.stitch_plan(w + 1, h, w + 1, h, T, M, size, lstride=3 * (w + 1), rstride=3 * (w + 1)) == 0


def test_strip_form_on_a_large_canvas_with_partial_tiles(dr3, O, ctx, synth):
    w, h = 1500, 1100
    left = synth.panorama_tile(w, h); right = np.roll(left, -600, axis=1)
    H = _H(-0.42 * w, 28.0, rot_deg=1.2)
    Hinv, T, M, size = dr3.stitch_geometry(H, w, h)
    assert dr3.stitch_plan(w, h, w, h, T, M, size) == 2 and (size[0] % 64 or size[1] % 64)
    assert np.array_equal(ctx.stitch_compose(left, right, H), O.stitch_compose(left, right, H))
