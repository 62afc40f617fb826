"""The host-only planning entry points of the C ABI (no GPU needed): which kernel a warp / stitch / paste would take, and over which
pixels its coordinate form is decided.  The parity of the forms themselves is the GPU suite's business (tests/test_gpu_warp.py,
test_gpu_stitch.py, test_gpu_panorama.py); these tests hold the decisions."""
import numpy as np
import pytest


def _rot(a, tx, ty, p=(0.0, 0.0)):
    return np.array([[np.cos(a), -np.sin(a), tx], [np.sin(a), np.cos(a), ty], [p[0], p[1], 1.0]], np.float32)


def test_paste_plan_is_made_over_the_box_not_over_the_canvas(dr3):
    """reference src/panorama.cpp:192 warps onto the whole canvas, the launch evaluates the image's box: a tile far right on a canvas
    wider than the +-8000 source pixels of the fast coordinate forms still takes them (and the strips), and only the rows the tile can
    reach are walked"""
    w, h = 4000, 3000
    for i, tx in enumerate((100.0, 3700.0, 7300.0, 10900.0)):
        M = _rot(np.deg2rad(1.0 - 0.6 * i), tx, 150.0 + 40 * i, (2e-6, -1e-6))
        k, coords, box = dr3.panorama_plan(w, h, (15168, 6099), M, istride=3 * w)
        assert (k, coords) == (2, 3), (tx, k, coords)
        assert 0 <= box[0] < box[2] <= 15168 and box[2] - box[0] < w + 400
        assert 0 <= box[1] < box[3] <= 6099 and box[3] - box[1] < h + 400          # not all 6099 canvas rows
        assert abs(box[0] - tx) < 200


def test_paste_plan_alignment_and_affinity_conditions(dr3):
    M = _rot(0.02, 40.0, 30.0, (1e-5, 0.0))
    assert dr3.panorama_plan(640, 480, (900, 700), M, istride=1920)[0] == 2
    assert dr3.panorama_plan(640, 480, (900, 700), M, istride=1924)[0] == 0        # rows not 16-byte aligned: the direct form
    assert dr3.panorama_plan(640, 480, (900, 700), M, istride=1924)[1] >= 1        # ... still with fast coordinates
    # a strongly projective map: W changes too fast for one anchor per strip column (|M7 / W| > 2e-4) -> direct form
    k, coords, _ = dr3.panorama_plan(640, 480, (900, 700), _rot(0.02, 40.0, 30.0, (0.0, 6e-4)), istride=1920)
    assert k == 0 and coords < 3
    # W of the forward map changes sign over the image: no bound on the rows a sample can land in -- every canvas row is walked
    k, coords, box = dr3.panorama_plan(640, 480, (900, 700), _rot(0.0, 40.0, 30.0, (-3e-3, 0.0)), istride=1920)
    assert k == 0 and (box[1], box[3]) == (0, 700)


def test_paste_plan_of_a_slab_keeps_the_canvas_geometry(dr3):
    """a GPU that owns canvas rows [y0, y1) plans the same box, cut to its rows"""
    M = _rot(-0.03, 60.0, 45.0, (1e-6, 2e-6))
    _, _, whole = dr3.panorama_plan(800, 600, (1000, 760), M, istride=2400)
    parts = [dr3.panorama_plan(800, 600, (1000, 760), M, istride=2400, slab=s)[2] for s in ((0, 300), (300, 301), (301, 760))]
    assert all(p[0] == whole[0] and p[2] == whole[2] for p in parts)
    assert parts[0][1] == whole[1] and parts[0][3] == 300 and (parts[1][1], parts[1][3]) == (300, 301) and parts[2][3] == whole[3]


def test_paste_plan_rejects_what_the_reference_throws_on(dr3):
    with pytest.raises(dr3.Dr3HipError):
        dr3.panorama_plan(1, 20, (50, 40), np.eye(3, dtype=np.float32))           # a 1-column box (src/panorama.cpp:163)
    with pytest.raises(dr3.Dr3HipError):
        dr3.panorama_plan(30, 20, (50, 40), np.eye(3, dtype=np.float32), istride=80)   # stride < 3 w


def test_warp_and_stitch_plans(dr3):
    job = dict(sw=4000, sh=3000, sstride=12000, dw=4000, dh=3000, dstride=12000, M=_rot(0.03, 10.0, -5.0, (1e-6, 1e-6)))
    p = dr3.warp_plan(job, dr3.T_8UC3)
    assert p["kernel"] == 2 and 1024 <= p["lds_bytes"] <= 16128 and p["tie_guard"] in (1, 2)
    assert dr3.warp_plan(dict(job, sstride=12004), dr3.T_8UC3)["kernel"] == 0      # source rows not 16-byte aligned
    assert dr3.warp_plan(job, dr3.T_32FC4)["kernel"] == 0                            # the strip kernel is 8UC3's
    H = _rot(np.deg2rad(1.5), -1800.0, 35.0, (1e-5, -6e-6))
    Hinv, T, Mr, size = dr3.stitch_geometry(H, 4000, 3000)
    assert dr3.stitch_plan(4000, 3000, 4000, 3000, T, Mr, size, lstride=12000, rstride=12000) == 2
    assert dr3.stitch_plan(4000, 3000, 4000, 3000, T, Mr, size, lstride=12004, rstride=12000) == 0


def test_paste_plan_properties_over_random_homographies(dr3):
    """the box a plan reports is the reference's bounding box (src/panorama.cpp:112-138) cut to the canvas, rows included in the canvas;
    the strip form implies the anchored coordinate form and aligned rows; an unaligned pitch never changes the box or the coordinate form"""
    rng = np.random.default_rng(2026)
    seen = set()
    for _ in range(300):
        w, h = int(rng.integers(16, 900)) * 2, int(rng.integers(16, 700))
        cw, ch = int(rng.integers(w // 2, 3 * w)), int(rng.integers(h // 2, 3 * h))
        a = rng.uniform(-0.3, 0.3)
        s_ = rng.uniform(0.6, 1.5)
        M = np.array([[s_ * np.cos(a), -s_ * np.sin(a), rng.uniform(-0.5 * w, cw)], [s_ * np.sin(a), s_ * np.cos(a), rng.uniform(-0.5 * h, ch)],
                      [rng.uniform(-2e-4, 2e-4), rng.uniform(-2e-4, 2e-4), 1.0]], np.float32)
        bb = dr3.panorama_bbox(M, w, h)
        pitch = (3 * w + 15) // 16 * 16
        try:
            k, coords, box = dr3.panorama_plan(w, h, (cw, ch), M, istride=pitch)
        except dr3.Dr3HipError:
            assert bb[2] - bb[0] < 2 or bb[2] - bb[0] > (1 << 24)          # only what the reference throws on
            continue
        assert box[0] == max(bb[0], 0) and box[2] == min(bb[2], cw) and 0 <= box[1] and box[3] <= ch
        assert k in (0, 2) and 0 <= coords <= 3
        if k == 2:
            assert coords == 3 and box[2] > box[0] and box[3] > box[1]
        k2, c2, b2 = dr3.panorama_plan(w, h, (cw, ch), M, istride=pitch + 4)
        assert k2 == 0 and c2 == coords and b2 == box
        seen.add((k, coords))
    assert (2, 3) in seen and any(k == 0 for k, _ in seen)
