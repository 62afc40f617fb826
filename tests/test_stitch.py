"""reconstruct::Stitch::process after align_pair (reference src/stitch.cpp:33-80): geometry on the host (CPU test against the
oracle), both warps + cv::addWeighted fused in one kernel (GPU test, bit-exact against the oracle's three passes)."""
import os

import numpy as np
import pytest


def _H(rng, w, shift):
    a = rng.uniform(-0.05, 0.05)
    return np.array([[np.cos(a) * rng.uniform(.95, 1.05), -np.sin(a), -shift * w + rng.uniform(-5, 5)],
                     [np.sin(a), np.cos(a) * rng.uniform(.95, 1.05), rng.uniform(-12, 12)],
                     [rng.uniform(-4e-5, 4e-5), rng.uniform(-4e-5, 4e-5), 1]], np.float32)


def test_stitch_geometry_matches_oracle(dr3, O):
    rng = np.random.default_rng(31)
    for _ in range(200):
        w, h = int(rng.integers(40, 2000)), int(rng.integers(30, 1200))
        H = _H(rng, w, rng.uniform(0.2, 0.8))
        g, o = dr3.stitch_geometry(H, w, h), O.stitch_geometry(H, w, h)
        assert g[3] == o[3]
        for a, b in zip(g[:3], o[:3]):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    I = np.eye(3, dtype=np.float32)                        # identity pair: canvas = the image, no translation
    Hinv, T, M, size = dr3.stitch_geometry(I, 752, 480)
    assert size == (752, 480) and np.array_equal(T, I) and np.array_equal(M, I)
    with pytest.raises(dr3.Dr3HipError):
        dr3.stitch_geometry(np.zeros((3, 3), np.float32), 100, 100)
    with pytest.raises(ValueError):
        O.stitch_geometry(np.zeros((3, 3), np.float32), 100, 100)


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,shift", [(320, 200, 0.55), (752, 480, 0.4), (97, 61, 0.3), (1240, 376, 0.1)])
def test_stitch_compose_bit_exact(dr3, O, ctx, synth, w, h, shift):
    rng = np.random.default_rng(w)
    if (w, h) == (1240, 376):                              # the reference's own KITTI frames as a 3-channel pair
        z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kitti_pair.npz"))
        left = np.stack([z["kitti0"]] * 3, -1); right = np.stack([z["kitti5"]] * 3, -1)
    else:
        left = synth.panorama_tile(w, h)
        right = np.roll(left, -int(shift * w), axis=1).copy()
    for k in range(3):
        H = _H(rng, w, shift)
        assert np.array_equal(ctx.stitch_compose(left, right, H), O.stitch_compose(left, right, H)), k
    got = ctx.stitch_compose(left, right, H, 0.25, 0.75, 3.5)
    assert np.array_equal(got, O.stitch_compose(left, right, H, 0.25, 0.75, 3.5))
    assert got.any()
