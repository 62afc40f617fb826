"""C++14 drop-in layer (include/dr3hip/compat.hpp) exercised through 3dr_amd/harness, which reproduces
the reference's call order around the hot path (src/initialization.cpp:587-635, src/panorama.cpp:156-229,
src/frame.cpp:18).  CPU: it compiles as C++14 and links against the C ABI alone.  GPU: its outputs equal
the oracle's for the same call sequence."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDIR = os.path.join(ROOT, "3dr_amd", "harness")
EXE = os.path.join(HDIR, "hotpath_harness")


def _build():
    subprocess.check_call(["make", "-C", HDIR], stdout=subprocess.DEVNULL)
    assert os.path.exists(EXE)


def test_compat_header_is_cxx14_and_links_against_c_abi(dr3):
    dr3.lib()
    _build()
    out = subprocess.check_output(["ldd", EXE], text=True)
    assert "libdr3hip.so" in out and "oracle" not in out and "opencv" not in out
    # header alone, strict C++14, no HIP/OpenCV headers on the include path
    subprocess.check_call(["g++", "-std=c++14", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           "-x", "c++", os.path.join(ROOT, "include", "dr3hip", "compat.hpp")])
    # the C header is plain C
    subprocess.check_call(["gcc", "-std=c99", "-fsyntax-only", "-Wall", "-Werror", "-x", "c", os.path.join(ROOT, "include", "dr3hip.h")])


def _bbox_x(M, w, h):
    """Panorama::set_bbox (src/panorama.cpp:113-142): corners through M, truncated toward zero"""
    pts = np.array([[0, 0, 1], [w, 0, 1], [0, h, 1], [w, h, 1]], np.float32).T
    t = M.astype(np.float32) @ pts
    X = t[0] / t[2]
    return int(X.min()), int(X.max())


@pytest.mark.gpu
def test_harness_matches_oracle(dr3, O, synth, tmp_path):
    _build()
    # (1) frames + points
    A, B, T = synth.frame_pair()
    h, w = A.shape
    pts = synth.grid_points(w, h, 30)
    # (2) one panorama tile
    tw, th, W, H, feather = 200, 150, 360, 300, 50
    tile = synth.panorama_tile(tw, th)
    M = synth.tile_homography(1, tw, th)
    min_x, max_x = _bbox_x(M, tw, th)
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        np.array([w, h, len(pts)], np.int32).tofile(f); A.tofile(f); B.tofile(f); pts.astype(np.float32).tofile(f)
        np.array([tw, th, W, H, feather], np.int32).tofile(f); tile.tofile(f); M.astype(np.float32).tofile(f)
        np.array([min_x, max_x], np.int32).tofile(f)
        Hs = np.array([[0.999, -0.02, -70.5], [0.02, 0.999, 6.25], [2e-5, -1e-5, 1]], np.float32)
        Hs.tofile(f); np.array([175.0], np.float32).tofile(f)
    r = subprocess.run([EXE, fin, fout], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr

    with open(fout, "rb") as f:
        n_kept = int(np.fromfile(f, np.int32, 1)[0])
        kept_ref = np.fromfile(f, np.float32, 2 * n_kept).reshape(-1, 2)
        kept_cur = np.fromfile(f, np.float32, 2 * n_kept).reshape(-1, 2)
        status = np.fromfile(f, np.uint8, len(pts))
        err = np.fromfile(f, np.float32, len(pts))
        pano = np.fromfile(f, np.uint8, W * H * 3).reshape(H, W, 3)
        pano_fused = np.fromfile(f, np.uint8, W * H * 3).reshape(H, W, 3)
        levels = []
        for _ in range(2):
            cw, ch = np.fromfile(f, np.int32, 2)
            levels.append(np.fromfile(f, np.uint8, cw * ch).reshape(ch, cw))
        sw_, sh_ = np.fromfile(f, np.int32, 2)
        hinv = np.fromfile(f, np.float32, 9).reshape(3, 3)
        stitched = np.fromfile(f, np.uint8, sw_ * sh_ * 3).reshape(sh_, sw_, 3)
        sph = np.fromfile(f, np.uint8, tw * th * 3).reshape(th, tw, 3)
        nc = int(np.fromfile(f, np.int32, 1)[0])
        corners = np.fromfile(f, np.dtype([("x", np.int32), ("y", np.int32), ("level", np.int32), ("score", np.float32)]), nc)

    # (1) LK with the reference literals, identity initial guess, then the erase loop
    po = O.lk_params(win=(30, 30), max_level=4, max_iters=1000, eps=1e-3, flags=O.LK_USE_INITIAL_FLOW)
    o_xy, o_st, o_err, L = O.lk_track(A, B, pts, pts.copy(), po, order=O.ORDER_WAVE64)
    assert L == 3
    assert np.array_equal(status, o_st) and np.array_equal(err.view(np.uint32), o_err.view(np.uint32))
    assert n_kept == int(o_st.sum())
    assert np.array_equal(kept_ref, pts[o_st == 1]) and np.array_equal(kept_cur.view(np.uint32), o_xy[o_st == 1].view(np.uint32))
    gt = synth.apply_T(T, pts)[o_st == 1]
    assert np.median(np.abs(kept_cur - gt)) < 0.05

    # (2) panorama paste: same sequence through the oracle
    src4 = np.ones((th, tw, 4), np.float32); src4[..., :3] = tile
    warped = O.warp_perspective(src4, M, (W, H))
    bw = feather if (max_x - min_x - 2 * feather) >= 0 else (max_x - min_x) // 2 - 1
    ramp = np.concatenate([O.linspace_f32(0.0, 1.0, bw), np.ones(max_x - min_x - 2 * bw, np.float32), O.linspace_f32(1.0, 0.0, bw)])
    canvas = np.zeros((H, W, 4), np.float32)
    O.feather_accumulate(canvas, warped, min_x, max_x, ramp)
    want = O.normalize_canvas(canvas)
    assert np.array_equal(pano, want)
    assert pano.any()
    # (2b) the fused device path (dr3hip::WeightedCanvas) against the oracle's own add_img_to_canvas
    canvas2 = np.zeros((H, W, 4), np.float32)
    O.panorama_add(tile, canvas2, M, feather)
    assert np.array_equal(pano_fused, O.normalize_canvas(canvas2))
    assert O.panorama_bbox(M, tw, th)[0::2] == (min_x, max_x) and np.array_equal(pano_fused, pano)

    # (4) stitch compose, spherical pre-warp and the FAST detector through the C++14 layer
    o_hinv, _, _, o_size = O.stitch_geometry(Hs, tw, th)
    assert (sw_, sh_) == o_size and np.array_equal(hinv.view(np.uint32), o_hinv.view(np.uint32))
    assert np.array_equal(stitched, O.stitch_compose(tile, tile, Hs))
    assert np.array_equal(sph, O.remap(tile, *O.spherical_map(tw, th, 175.0)))
    gc_ = -(-w // 30)
    occ = np.zeros(gc_ * -(-h // 30), np.uint8); occ[int(50 / 30) * gc_ + int(100 / 30)] = 1
    ox, oy, osc, ol = O.fast_detect_grid(O.halfsample_pyramid(A, 3, O.HALF_REF_X86), 30, 20, 20.0, occ)
    m = ol >= 0
    assert nc == int(m.sum()) and np.array_equal(corners["x"], ox[m]) and np.array_equal(corners["y"], oy[m])
    assert np.array_equal(corners["level"], ol[m]) and np.array_equal(corners["score"].view(np.uint32), osc[m].view(np.uint32))

    # (3) Frame-constructor pyramid, byte-for-byte what the reference computes on x86-64
    ref_pyr = O.halfsample_pyramid(A, 3, O.HALF_REF_X86)
    assert np.array_equal(levels[0], ref_pyr[1]) and np.array_equal(levels[1], ref_pyr[2])
