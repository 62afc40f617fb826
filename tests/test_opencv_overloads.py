"""The DR3HIP_WITH_OPENCV branch of include/dr3hip/compat.hpp: overloads taking const cv::Mat&, cv::Mat& dst,
std::vector<cv::Point2f>&, cv::Size, cv::TermCriteria and a cv::Mat M, so that the reference's call expressions
(src/initialization.cpp:608-613, src/stitch.cpp:73-74, src/panorama.cpp:192, src/utils.cpp:191, :196-201, src/frame.cpp:18) change by
namespace only.  OpenCV is absent from the image: tests/opencv_shim/opencv2/core.hpp is a TEST-ONLY stand-in that declares the
cv:: names involved (it is not OpenCV and pins nothing); tests/opencv_shim/callsites.cpp holds the call expressions verbatim.
CPU: the translation unit compiles warning-free as C++14 and links against the C ABI alone.  GPU: it runs, and every output
equals the oracle's for the same call."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "tests", "opencv_shim")
EXE = os.path.join(SHIM, "callsites")


def _build():
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-Wall", "-Wextra", "-Werror", "-I", SHIM, "-I", os.path.join(ROOT, "include"),
                           "-o", EXE, os.path.join(SHIM, "callsites.cpp"), "-L", os.path.join(ROOT, "3dr_amd"), "-ldr3hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "3dr_amd"), "-Wl,-rpath,/opt/rocm/lib"])


def test_call_sites_change_by_namespace_only(dr3):
    dr3.lib()
    _build()
    src = open(os.path.join(SHIM, "callsites.cpp")).read()
    body = src[src.index("int main"):]
    # no cast, no view construction, no copy-out at any call site
    assert "reinterpret_cast" not in body and "MatView" not in body and "dr3hip::Image" not in body
    # the call expressions of the reference, namespace swapped (whitespace-insensitive)
    flat = re.sub(r"\s+", "", body)
    for expr in ("dr3hip::calcOpticalFlowPyrLK(frame_ref->_img_pyr[0],frame_cur->_img_pyr[0],_kps_ref,_kps_cur,status,error,"
                 "cv::Size2i(klt_win_size,klt_win_size),4,termcrit,cv::OPTFLOW_USE_INITIAL_FLOW);",
                 "dr3hip::warpPerspective(_right_img,warped_right_image,translate_top_left_mat*Hinv,{new_width,new_height});",
                 "dr3hip::warpPerspective(_left_img,warped_left_image,translate_top_left_mat,{new_width,new_height});",
                 "dr3hip::remap(img,warped,u,v,CV_INTER_LINEAR);"):
        assert expr in flat, expr
    out = subprocess.check_output(["ldd", EXE], text=True)
    assert "libdr3hip.so" in out and "opencv" not in out and "oracle" not in out


def _rd_mat(f, dtype_of):
    r, c, t = np.fromfile(f, np.int32, 3)
    dt, cn = dtype_of[int(t)]
    return np.fromfile(f, dt, int(r) * int(c) * cn).reshape((int(r), int(c), cn) if cn > 1 else (int(r), int(c)))


@pytest.mark.gpu
def test_cv_typed_overloads_equal_the_oracle(dr3, O, synth, tmp_path):
    _build()
    A, B, _ = synth.frame_pair()
    h, w = A.shape
    pts = synth.grid_points(w, h, 30)
    tw, th = 320, 200
    tile = synth.panorama_tile(tw, th)
    H = np.array([[0.999, -0.02, -70.5], [0.02, 0.999, 6.25], [2e-5, -1e-5, 1]], np.float32)
    Hinv, T, M_right, (nw, nh) = dr3.stitch_geometry(H, tw, th)
    rng = np.random.default_rng(11)
    mu = rng.uniform(-4, tw + 4, (th, tw)).astype(np.float32); mv = rng.uniform(-4, th + 4, (th, tw)).astype(np.float32)
    f = np.float32(260.0)
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as fh:
        np.array([w, h, len(pts)], np.int32).tofile(fh); A.tofile(fh); B.tofile(fh); pts.astype(np.float32).tofile(fh)
        np.array([tw, th, nw, nh], np.int32).tofile(fh); tile.tofile(fh)
        Hinv.astype(np.float32).tofile(fh); np.array([-T[0, 2], -T[1, 2]], np.float32).tofile(fh)
        mu.tofile(fh); mv.tofile(fh); np.array([f], np.float32).tofile(fh)
    r = subprocess.run([EXE, fin, fout], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    types = {0: (np.uint8, 1), 16: (np.uint8, 3), 5: (np.float32, 1), 29: (np.float32, 4)}     # CV_8UC1, CV_8UC3, CV_32FC1, CV_32FC4
    n = len(pts)
    with open(fout, "rb") as fh:
        xy = np.fromfile(fh, np.float32, 2 * n).reshape(n, 2); st = np.fromfile(fh, np.uint8, n); err = np.fromfile(fh, np.float32, n)
        pyr1, pyr2 = _rd_mat(fh, types), _rd_mat(fh, types)
        wr, wl, w32 = _rd_mat(fh, types), _rd_mat(fh, types), _rd_mat(fh, types)
        rm, sph, cyl = _rd_mat(fh, types), _rd_mat(fh, types), _rd_mat(fh, types)
    po = O.lk_params(win=(30, 30), max_level=4, max_iters=1000, eps=1e-3, flags=O.LK_USE_INITIAL_FLOW)
    o_xy, o_st, o_err, _ = O.lk_track(A, B, pts, pts.copy(), po, order=O.ORDER_WAVE64)
    assert np.array_equal(xy.view(np.uint32), o_xy.view(np.uint32)) and np.array_equal(st, o_st) and np.array_equal(err.view(np.uint32), o_err.view(np.uint32))
    ref = O.halfsample_pyramid(A, 3, O.HALF_REF_X86)
    assert np.array_equal(pyr1, ref[1]) and np.array_equal(pyr2, ref[2])
    Mr = (T.astype(np.float64) @ Hinv.astype(np.float64)).astype(np.float32)      # the shim's product: f32 inputs, double accumulation, as cv::gemm
    assert np.array_equal(wr, O.warp_perspective(tile, Mr, (nw, nh))) and np.array_equal(wl, O.warp_perspective(tile, T, (nw, nh)))
    t4 = np.ones((th, tw, 4), np.float32); t4[..., :3] = tile
    assert np.array_equal(w32.view(np.uint32), O.warp_perspective(t4, Mr, (nw, nh)).view(np.uint32))
    assert np.array_equal(rm, O.remap(tile, mu, mv))
    assert np.array_equal(sph, O.remap(tile, *O.spherical_map(tw, th, float(f)))) and np.array_equal(cyl, O.remap(tile, *O.cylindrical_map(tw, th, float(f))))
