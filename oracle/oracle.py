"""ctypes front-end of the CPU oracle (oracle/dr3_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
PARITY UNPINNED for every row whose arithmetic is OpenCV's (absent here); the half-sampler's SSE2 path and linspace are pinned by
the reference's own executed code -- see dr3_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# DR3O_LIB: the sanitizer build (oracle/libdr3oracle_asan.so) under tests/test_oracle_asan.py
_LIB_PATH = os.environ.get("DR3O_LIB") or os.path.join(_HERE, "libdr3oracle.so")

# pixel types / flags (mirrors dr3_oracle.h)
T_8UC1, T_8UC3, T_32FC4, T_8UC4, T_32FC1 = 0, 1, 2, 3, 4
LK_USE_INITIAL_FLOW, LK_GET_MIN_EIGENVALS = 4, 8
WARP_INVERSE_MAP = 16
ORDER_RASTER, ORDER_WAVE64, ORDER_SSE2 = 0, 1, 2
ORDER_NAMES = {ORDER_RASTER: "raster", ORDER_WAVE64: "kernel", ORDER_SSE2: "sse2"}
HALF_TRUNC, HALF_SSE2, HALF_REF_SCALAR, HALF_REF_X86 = 0, 1, 2, 3
CRIT_COUNT, CRIT_EPS = 1, 2


class AlignParams(C.Structure):
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("max_level", C.c_int), ("min_level", C.c_int), ("n_iter", C.c_int), ("eps", C.c_double)]


class LKParams(C.Structure):
    _fields_ = [("win_w", C.c_int), ("win_h", C.c_int), ("max_level", C.c_int),
                ("crit_type", C.c_int), ("max_iters", C.c_int), ("eps", C.c_double),
                ("flags", C.c_int), ("min_eig_threshold", C.c_double)]


def _under_profiler():
    """rocprofv3 preloads a library that initialises the GPU in every process of the tree: build tools (make, sh, gcc ...) must not be
    started from such a process (this pool forbids an exec from a GPU-initialised process)"""
    return any(k.startswith(("ROCPROFILER", "ROCPROF_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", "")


def build(force=False):
    """Compile the oracle with the committed Makefile (gcc only)."""
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "dr3_oracle.c")):
        if _under_profiler():
            raise RuntimeError("the oracle library is missing or older than its source and this process runs under a profiler: build it "
                               "first, outside the profiler (python -c 'import __graft_entry__ as g; g.build()')")
        subprocess.check_call(["make", "-C", _HERE, os.path.basename(_LIB_PATH)], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def build_native():
    """-O3 -march=native build for the CPU baseline, made on the host that runs it (the file name carries the CPU model, so a
    build made elsewhere is never picked up).  Returns its path, or None if it cannot be built here."""
    import hashlib
    try:
        model = [l for l in open("/proc/cpuinfo") if l.startswith(("model name", "flags"))][:2]
    except OSError:
        model = []
    name = "libdr3oracle_native_%s.so" % hashlib.sha1("".join(model).encode()).hexdigest()[:10]
    path = os.path.join(_HERE, name)
    try:
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(os.path.join(_HERE, "dr3_oracle.c")):
            if _under_profiler():
                return None
            subprocess.check_call(["make", "-C", _HERE, "NATIVE=" + name, name], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return path
    except (subprocess.CalledProcessError, OSError):
        return None


def use_library(path=None):
    """switch the module to another build of the oracle (None = the portable default); the next lib() call loads it"""
    global _lib, _LIB_PATH
    _lib = None
    _LIB_PATH = path or os.environ.get("DR3O_LIB") or os.path.join(_HERE, "libdr3oracle.so")


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        vp, ip = C.c_void_p, C.c_int
        L.dr3o_lk_effective_max_level.argtypes = [ip] * 5
        L.dr3o_lk_effective_max_level.restype = ip
        L.dr3o_pyr_down_u8.argtypes = [vp, ip, ip, ip, vp, ip]
        L.dr3o_pyr_down_u8.restype = None
        L.dr3o_scharr_s16.argtypes = [vp, ip, ip, ip, vp, ip]
        L.dr3o_scharr_s16.restype = None
        L.dr3o_lk_track_ex.argtypes = [vp, vp, ip, ip, ip, ip, vp, vp, ip, vp, vp,
                                       C.POINTER(LKParams), ip, ip, vp]
        L.dr3o_lk_track_ex.restype = ip
        L.dr3o_lk_track_opt.argtypes = L.dr3o_lk_track_ex.argtypes
        L.dr3o_lk_track_opt.restype = ip
        L.dr3o_window_sums.argtypes = [vp, vp, ip, ip, ip, vp, vp]
        L.dr3o_window_sums.restype = None
        L.dr3o_halfsample_u8.argtypes = [vp, ip, ip, ip, vp, ip, ip]
        L.dr3o_halfsample_u8.restype = None
        L.dr3o_invert3x3.argtypes = [vp, vp]
        L.dr3o_invert3x3.restype = ip
        L.dr3o_warp_perspective.argtypes = [vp, ip, ip, ip, ip, vp, ip, ip, ip, vp, ip, ip]
        L.dr3o_warp_perspective.restype = ip
        L.dr3o_remap_bilinear.argtypes = [vp, ip, ip, ip, ip, vp, ip, ip, ip, vp, vp, ip, ip]
        L.dr3o_remap_bilinear.restype = ip
        L.dr3o_shi_tomasi_score.argtypes = [vp, ip, ip, ip, ip, ip]
        L.dr3o_shi_tomasi_score.restype = C.c_float
        L.dr3o_spherical_map.argtypes = [ip, ip, C.c_float, vp, vp]
        L.dr3o_spherical_map.restype = None
        L.dr3o_cylindrical_map.argtypes = [ip, ip, C.c_float, vp, vp]
        L.dr3o_cylindrical_map.restype = None
        L.dr3o_feather_accumulate.argtypes = [vp, ip, ip, ip, vp, ip, ip, ip, vp]
        L.dr3o_feather_accumulate.restype = None
        L.dr3o_normalize_canvas.argtypes = [vp, ip, ip, ip, vp, ip]
        L.dr3o_normalize_canvas.restype = None
        L.dr3o_linspace_f32.argtypes = [C.c_float, C.c_float, ip, vp]
        L.dr3o_linspace_f32.restype = None
        L.dr3o_fast10_score_map.argtypes = [vp, ip, ip, ip, ip, vp]
        L.dr3o_fast10_score_map.restype = ip
        L.dr3o_fast_nonmax_3x3.argtypes = [vp, ip, ip, vp]
        L.dr3o_fast_nonmax_3x3.restype = ip
        L.dr3o_fast_detect_grid.argtypes = [vp, vp, ip, ip, ip, ip, ip, C.c_double, vp, vp, vp, vp, vp]
        L.dr3o_fast_detect_grid.restype = ip
        L.dr3o_check_fundamental.argtypes = [vp, vp, vp, ip, C.c_float, vp]
        L.dr3o_check_fundamental.restype = C.c_float
        L.dr3o_normalize_points.argtypes = [vp, vp, ip, vp, vp]
        L.dr3o_normalize_points.restype = None
        L.dr3o_sparse_align.argtypes = [vp, vp, vp, vp, vp, ip, vp, ip, C.POINTER(AlignParams), vp, vp, vp, vp]
        L.dr3o_sparse_align.restype = ip
        L.dr3o_stitch_geometry.argtypes = [vp, ip, ip, vp, vp, vp, vp, vp]
        L.dr3o_stitch_geometry.restype = ip
        L.dr3o_stitch_compose.argtypes = [vp, ip, ip, ip, vp, ip, ip, ip, vp, vp, ip, ip, C.c_float, C.c_float, C.c_float, vp, ip, vp, ip]
        L.dr3o_stitch_compose.restype = ip
        L.dr3o_panorama_bbox.argtypes = [vp, ip, ip, vp, vp, vp, vp]
        L.dr3o_panorama_bbox.restype = None
        L.dr3o_feather_ramp.argtypes = [ip, ip, ip, vp]
        L.dr3o_feather_ramp.restype = ip
        L.dr3o_panorama_add.argtypes = [vp, ip, ip, ip, vp, ip, ip, ip, vp, ip, vp, ip]
        L.dr3o_panorama_add.restype = ip
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _gray(img):
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 2
    if img.strides[1] != 1:
        img = np.ascontiguousarray(img)
    return img


def lk_effective_max_level(w, h, win_w, win_h, max_level):
    return lib().dr3o_lk_effective_max_level(w, h, win_w, win_h, max_level)


def pyr_down(img):
    img = _gray(img)
    h, w = img.shape
    out = np.empty(((h + 1) // 2, (w + 1) // 2), np.uint8)
    lib().dr3o_pyr_down_u8(_p(img), w, h, img.strides[0], _p(out), out.strides[0])
    return out


def scharr(img):
    """returns int16 (h, w, 2): [...,0]=Ix, [...,1]=Iy (x32 scale)"""
    img = _gray(img)
    h, w = img.shape
    out = np.empty((h, w, 2), np.int16)
    lib().dr3o_scharr_s16(_p(img), w, h, img.strides[0], _p(out), 2 * w)
    return out


def lk_params(win=(21, 21), max_level=3, max_iters=30, eps=0.01, crit_type=CRIT_COUNT | CRIT_EPS,
              flags=0, min_eig_threshold=1e-4):
    return LKParams(int(win[0]), int(win[1]), int(max_level), int(crit_type), int(max_iters),
                    float(eps), int(flags), float(min_eig_threshold))


def lk_track(prev, nxt, prev_xy, next_xy0=None, params=None, order=ORDER_RASTER, n_threads=0,
             want_iters=False, opt=False):
    """cv::calcOpticalFlowPyrLK restatement.  Returns (next_xy, status, err, eff_max_level[, iters]).
    opt: the "port-opt" organisation of the same computation (dr3o_lk_track_opt: identical results, faster)."""
    prev, nxt = _gray(prev), _gray(nxt)
    assert prev.shape == nxt.shape
    h, w = prev.shape
    p = params if params is not None else lk_params()
    prev_xy = np.ascontiguousarray(prev_xy, np.float32).reshape(-1, 2)
    n = prev_xy.shape[0]
    nxt_xy = (prev_xy.copy() if next_xy0 is None
              else np.ascontiguousarray(next_xy0, np.float32).reshape(-1, 2).copy())
    status = np.zeros(n, np.uint8)
    err = np.zeros(n, np.float32)
    eff = lk_effective_max_level(w, h, p.win_w, p.win_h, p.max_level)
    iters = np.zeros((n, eff + 1), np.int32) if want_iters else None
    fn = lib().dr3o_lk_track_opt if opt else lib().dr3o_lk_track_ex
    r = fn(_p(prev), _p(nxt), w, h, prev.strides[0], nxt.strides[0],
                               _p(prev_xy), _p(nxt_xy), n, _p(status), _p(err), C.byref(p),
                               order, n_threads, _p(iters) if want_iters else None)
    if r < 0:
        raise ValueError("dr3o_lk_track failed: %d" % r)
    if want_iters:
        return nxt_xy, status, err, r, iters
    return nxt_xy, status, err, r


def window_sums(dI, diff, order):
    """(A11, A12, A22), (b1, b2) of one window from given int16 gradients dI (h, w, 2) and mismatches diff (h, w) in a summation
    order: the bookkeeping of the tracker's window sums alone (tests/test_oracle_sse2.py)"""
    dI = np.ascontiguousarray(dI, np.int16); diff = np.ascontiguousarray(diff, np.int16)
    h, w = diff.shape
    A, b = np.zeros(3, np.float32), np.zeros(2, np.float32)
    lib().dr3o_window_sums(_p(dI), _p(diff), w, h, order, _p(A), _p(b))
    return A, b


def sse2_replay_lib():
    """oracle/sse2_replay.c: OpenCV's CV_SSE2 accumulation sequences with real intrinsics (x86-64 only); None elsewhere"""
    import platform
    if platform.machine() not in ("x86_64", "AMD64"):
        return None
    path = os.path.join(_HERE, "libsse2_replay.so")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(os.path.join(_HERE, "sse2_replay.c")):
        if _under_profiler():
            return None
        subprocess.check_call(["make", "-C", _HERE, "libsse2_replay.so"], stdout=subprocess.DEVNULL)
    L = C.CDLL(path)
    L.sse2_replay_A.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.sse2_replay_A.restype = None
    L.sse2_replay_B.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.sse2_replay_B.restype = None
    return L


def effective_criteria(p):
    """TermCriteria clamping of SparsePyrLKOpticalFlowImpl::calc (SURVEY A-3): (max_count, eps)"""
    mc = min(max(int(p.max_iters), 0), 100) if p.crit_type & 1 else 30
    eps = min(max(float(p.eps), 0.0), 10.0) if p.crit_type & 2 else 0.01
    return mc, eps


def order_agreement(a, b, params, tol_px=1e-3):
    """How two runs of the tracker that differ only in the f32 summation order of the window sums agree (a, b = results of
    lk_track(..., want_iters=True); one of them may come from the GPU with the kernel-order oracle's iteration counts).
    The bar of BASELINE.json's north_star is 1e-3 px.  It holds for every point that takes the SAME ITERATION PATH in both
    orders: tracked in both, the same number of Gauss-Newton iterations on every level, below the iteration clamp.  The
    remainder are points where the last bit of an f32 sum flips an exit test (|delta|^2 <= eps, the oscillation test) or that
    never converge and run into the clamp; OpenCV's own scalar and SIMD builds differ on those the same way.  They are counted,
    and those that converged are bounded by one more Gauss-Newton step (2 eps)."""
    import numpy as _np
    mc, eps = effective_criteria(params)
    xy_a, st_a, it_a = a[0], a[1], a[4]
    xy_b, st_b, it_b = b[0], b[1], b[4]
    both = (st_a == 1) & (st_b == 1)
    clamped = both & ((it_a.max(1) >= mc) | (it_b.max(1) >= mc)) if (it_a.shape[1] and mc > 0) else both & False
    same = both & (it_a == it_b).all(1) & ~clamped
    flipped = both & ~same & ~clamped
    with _np.errstate(invalid="ignore"):
        d = _np.abs(xy_a - xy_b).max(1)
    return {"tracked_both": int(both.sum()), "status_equal": bool(_np.array_equal(st_a, st_b)),
            "same_path": int(same.sum()), "same_path_max_dev_px": float(d[same].max()) if same.any() else 0.0,
            "same_path_within_tol": bool((d[same] <= tol_px).all()) if same.any() else True,
            "within_tol_frac": float((d[both] <= tol_px).mean()) if both.any() else 1.0,
            "beyond_tol": int((d[both] > tol_px).sum()),
            "max_dev_px": float(d[both].max()) if both.any() else 0.0,
            "exit_flipped": int(flipped.sum()), "exit_flipped_max_dev_px": float(d[flipped].max()) if flipped.any() else 0.0,
            "clamped": int(clamped.sum()), "clamped_max_dev_px": float(d[clamped].max()) if clamped.any() else 0.0,
            "converged_max_dev_px": float(d[both & ~clamped].max()) if (both & ~clamped).any() else 0.0,
            "converged_within_tol": bool((d[both & ~clamped] <= tol_px).all()) if (both & ~clamped).any() else True,
            "eps": eps, "max_count": mc, "tol_px": tol_px}


def halfsample(img, mode=HALF_TRUNC):
    img = _gray(img)
    h, w = img.shape
    out = np.empty((h // 2, w // 2), np.uint8)
    lib().dr3o_halfsample_u8(_p(img), w, h, img.strides[0], _p(out), out.strides[0], mode)
    return out


def halfsample_pyramid(img, n_levels, mode=HALF_TRUNC):
    """utils::create_img_pyramid (src/utils.cpp:421-430)"""
    pyr = [_gray(img)]
    for _ in range(1, n_levels):
        pyr.append(halfsample(pyr[-1], mode))
    return pyr


def invert3x3(M):
    M = np.ascontiguousarray(M, np.float64).reshape(9)
    out = np.zeros(9, np.float64)
    ok = lib().dr3o_invert3x3(_p(M), _p(out))
    return out.reshape(3, 3), bool(ok)


def _img_type(img):
    img = np.asarray(img)
    if img.dtype == np.uint8:
        if img.ndim == 2:
            return T_8UC1
        return {1: T_8UC1, 3: T_8UC3, 4: T_8UC4}[img.shape[2]]
    if img.dtype == np.float32:
        if img.ndim == 2:
            return T_32FC1
        return {1: T_32FC1, 4: T_32FC4}[img.shape[2]]
    raise TypeError("unsupported image dtype %s" % img.dtype)


def warp_perspective(src, M, dsize, flags=0, n_threads=0):
    """cv::warpPerspective(src, dst, M, dsize=(W,H)) INTER_LINEAR, BORDER_CONSTANT 0."""
    src = np.ascontiguousarray(src)
    t = _img_type(src)
    sh, sw = src.shape[:2]
    dw, dh = int(dsize[0]), int(dsize[1])
    dst = np.zeros((dh, dw) + src.shape[2:], src.dtype)
    M = np.ascontiguousarray(M, np.float32).reshape(9)
    r = lib().dr3o_warp_perspective(_p(src), sw, sh, src.strides[0], t, _p(dst), dw, dh,
                                    dst.strides[0], _p(M), flags, n_threads)
    if r != 0:
        raise ValueError("dr3o_warp_perspective failed: %d" % r)
    return dst


def remap(src, mapx, mapy, n_threads=0):
    src = np.ascontiguousarray(src)
    t = _img_type(src)
    sh, sw = src.shape[:2]
    mapx = np.ascontiguousarray(mapx, np.float32)
    mapy = np.ascontiguousarray(mapy, np.float32)
    dh, dw = mapx.shape
    dst = np.zeros((dh, dw) + src.shape[2:], src.dtype)
    r = lib().dr3o_remap_bilinear(_p(src), sw, sh, src.strides[0], t, _p(dst), dw, dh,
                                  dst.strides[0], _p(mapx), _p(mapy), dw, n_threads)
    if r != 0:
        raise ValueError("dr3o_remap_bilinear failed: %d" % r)
    return dst


def shi_tomasi_score(img, u, v):
    img = _gray(img)
    h, w = img.shape
    return float(lib().dr3o_shi_tomasi_score(_p(img), w, h, img.strides[0], int(u), int(v)))


def spherical_map(w, h, f):
    u = np.empty((h, w), np.float32)
    v = np.empty((h, w), np.float32)
    lib().dr3o_spherical_map(w, h, f, _p(u), _p(v))
    return u, v


def cylindrical_map(w, h, f):
    u = np.empty((h, w), np.float32)
    v = np.empty((h, w), np.float32)
    lib().dr3o_cylindrical_map(w, h, f, _p(u), _p(v))
    return u, v


def linspace_f32(a, b, n):
    out = np.empty(n, np.float32)
    lib().dr3o_linspace_f32(a, b, n, _p(out))
    return out


def feather_accumulate(canvas, warped, min_x, max_x, ramp):
    """in-place on canvas and warped (both float32 HxWx4)"""
    assert canvas.dtype == np.float32 and warped.dtype == np.float32
    ch, cw = canvas.shape[:2]
    ramp = np.ascontiguousarray(ramp, np.float32)
    lib().dr3o_feather_accumulate(_p(canvas), cw, ch, canvas.strides[0], _p(warped),
                                  warped.strides[0], min_x, max_x, _p(ramp))


def normalize_canvas(canvas):
    ch, cw = canvas.shape[:2]
    out = np.zeros((ch, cw, 3), np.uint8)
    lib().dr3o_normalize_canvas(_p(canvas), cw, ch, canvas.strides[0], _p(out), out.strides[0])
    return out


def panorama_bbox(M, w, h):
    """Panorama::set_bbox (src/panorama.cpp:112-138) -> (min_x, min_y, max_x, max_y)"""
    M = np.ascontiguousarray(M, np.float32).reshape(9)
    out = [C.c_int() for _ in range(4)]
    lib().dr3o_panorama_bbox(_p(M), int(w), int(h), *[C.byref(o) for o in out])
    return tuple(o.value for o in out)


def feather_ramp(min_x, max_x, feathering_width):
    """the left/mid/right blend vector of add_img_to_canvas (src/panorama.cpp:163-177)"""
    n = max_x - min_x
    ramp = np.zeros(max(n, 0), np.float32)
    r = lib().dr3o_feather_ramp(int(min_x), int(max_x), int(feathering_width), _p(ramp))
    if r < 0:
        raise ValueError("the reference throws for a %d-column bounding box" % n)
    return ramp


def panorama_add(img, canvas, M, feathering_width, n_threads=0):
    """Panorama::add_img_to_canvas (src/panorama.cpp:156-212), canvas (H, W, 4) f32 updated in place"""
    img = np.ascontiguousarray(img, np.uint8)
    assert img.ndim == 3 and img.shape[2] == 3 and canvas.dtype == np.float32 and canvas.flags.c_contiguous
    ch, cw = canvas.shape[:2]
    M = np.ascontiguousarray(M, np.float32).reshape(9)
    scratch = np.empty((ch, cw, 4), np.float32)
    r = lib().dr3o_panorama_add(_p(img), img.shape[1], img.shape[0], img.strides[0], _p(canvas), cw, ch, canvas.strides[0],
                                _p(M), int(feathering_width), _p(scratch), int(n_threads))
    if r != 0:
        raise ValueError("panorama_add failed (%d)" % r)
    return canvas


def fast10_score_map(img, threshold=20):
    """FAST-10 corner scores (0 = no corner) as fast_corner_detect_10 + fast_corner_score_10 give them"""
    img = _gray(img)
    h, w = img.shape
    sm = np.zeros((h, w), np.uint8)
    lib().dr3o_fast10_score_map(_p(img), w, h, img.strides[0], int(threshold), _p(sm))
    return sm


def fast_nonmax_3x3(score_map):
    sm = np.ascontiguousarray(score_map, np.uint8)
    keep = np.zeros_like(sm)
    lib().dr3o_fast_nonmax_3x3(_p(sm), sm.shape[1], sm.shape[0], _p(keep))
    return keep


def fast_detect_grid(pyr, cell_size=30, fast_threshold=20, detection_threshold=20.0, occupancy=None):
    """FastDetector::detect (src/features.cpp:43-98) -> (x, y, score, level) arrays per grid cell; level -1 = empty"""
    pyr = [_gray(p) for p in pyr]
    h, w = pyr[0].shape
    n = len(pyr)
    gc, gr = -(-w // cell_size), -(-h // cell_size)
    ptrs = (C.c_void_p * n)(*[p.ctypes.data for p in pyr])
    strides = (C.c_int * n)(*[p.strides[0] for p in pyr])
    cx, cy, cl = (np.zeros(gc * gr, np.int32) for _ in range(3))
    cs = np.zeros(gc * gr, np.float32)
    occ = None if occupancy is None else np.ascontiguousarray(occupancy, np.uint8)
    r = lib().dr3o_fast_detect_grid(ptrs, strides, w, h, n, int(cell_size), int(fast_threshold), float(detection_threshold),
                                    None if occ is None else _p(occ), _p(cx), _p(cy), _p(cs), _p(cl))
    if r < 0:
        raise MemoryError("dr3o_fast_detect_grid")
    return cx, cy, cs, cl


def check_fundamental(F21, pts1, pts2, sigma=1.0):
    """InitHelper::CheckFundamental (src/initialization.cpp:171-249) -> (score f32, inliers u8[n])"""
    F = np.ascontiguousarray(F21, np.float32).reshape(9)
    p1 = np.ascontiguousarray(pts1, np.float32); p2 = np.ascontiguousarray(pts2, np.float32)
    inl = np.zeros(len(p1), np.uint8)
    sc = lib().dr3o_check_fundamental(_p(F), _p(p1), _p(p2), len(p1), C.c_float(sigma), _p(inl))
    return np.float32(sc), inl


def normalize_points(xy, mask=None):
    """InitHelper::Normalize (src/initialization.cpp:365-410) -> (normalised points (n,2) f32, T (3,3) f32)"""
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
    out = np.zeros_like(xy); T = np.zeros(9, np.float32)
    with np.errstate(all="ignore"):
        lib().dr3o_normalize_points(_p(xy), None if m is None else _p(m), len(xy), _p(out), _p(T))
    return out, T.reshape(3, 3)


def align_params(fx, fy, cx, cy, max_level=4, min_level=2, n_iter=10, eps=1e-8):
    return AlignParams(float(fx), float(fy), float(cx), float(cy), int(max_level), int(min_level), int(n_iter), float(eps))


def lk_pyramid(img, n_levels):
    """the Gaussian (pyrDown) pyramid the frame batch holds: level 0 .. n_levels - 1"""
    pyr = [_gray(img)]
    for _ in range(1, n_levels):
        pyr.append(pyr_down(pyr[-1]))
    return pyr


def sparse_align(ref_pyr, cur_pyr, xyz, params, T0=None):
    """6-DoF inverse-compositional sparse image alignment (EXTENSION specified by this build, dr3_oracle.h).
    ref_pyr / cur_pyr: lists of u8 levels; xyz (n,3) f32 in the reference camera frame.
    Returns (T (3,4) f64 [R|t], n_meas, chi2, iters per level)."""
    n_levels = len(ref_pyr)
    ref = [np.ascontiguousarray(a, np.uint8) for a in ref_pyr]; cur = [np.ascontiguousarray(a, np.uint8) for a in cur_pyr]
    rp = (C.c_void_p * n_levels)(*[a.ctypes.data for a in ref]); cp = (C.c_void_p * n_levels)(*[a.ctypes.data for a in cur])
    w = (C.c_int * n_levels)(*[a.shape[1] for a in ref]); h = (C.c_int * n_levels)(*[a.shape[0] for a in ref])
    st = (C.c_int * n_levels)(*[a.strides[0] for a in ref])
    assert all(a.shape == b.shape and a.strides == b.strides for a, b in zip(ref, cur))
    xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
    T = np.ascontiguousarray(np.eye(3, 4) if T0 is None else T0, np.float64).copy()
    n_meas = C.c_int(); chi2 = C.c_float(); iters = np.zeros(n_levels, np.int32)
    r = lib().dr3o_sparse_align(rp, cp, w, h, st, n_levels, _p(xyz), len(xyz), C.byref(params), _p(T), C.byref(n_meas), C.byref(chi2), _p(iters))
    if r < 0:
        raise ValueError("dr3o_sparse_align failed: %d" % r)
    return T, n_meas.value, np.float32(chi2.value), iters


def stitch_geometry(H, w, h):
    """Stitch::process geometry (src/stitch.cpp:33-71) -> (Hinv, T, M_right, (new_w, new_h))"""
    H = np.ascontiguousarray(H, np.float32).reshape(9)
    Hinv, T, M = (np.zeros(9, np.float32) for _ in range(3))
    nw, nh = C.c_int(), C.c_int()
    if lib().dr3o_stitch_geometry(_p(H), int(w), int(h), _p(Hinv), _p(T), _p(M), C.byref(nw), C.byref(nh)) != 0:
        raise ValueError("singular homography")
    return Hinv.reshape(3, 3), T.reshape(3, 3), M.reshape(3, 3), (nw.value, nh.value)


def stitch_compose(left, right, H, alpha=0.5, beta=0.5, gamma=0.0, n_threads=0):
    """Stitch::process after align_pair (src/stitch.cpp:33-80): the stitched 8UC3 image"""
    left = np.ascontiguousarray(left, np.uint8); right = np.ascontiguousarray(right, np.uint8)
    _, T, M, (nw, nh) = stitch_geometry(H, right.shape[1], right.shape[0])
    out = np.zeros((nh, nw, 3), np.uint8)
    scratch = np.zeros(2 * nw * nh * 3, np.uint8)
    r = lib().dr3o_stitch_compose(_p(left), left.shape[1], left.shape[0], left.strides[0], _p(right), right.shape[1], right.shape[0],
                                  right.strides[0], _p(np.ascontiguousarray(T.reshape(9))), _p(np.ascontiguousarray(M.reshape(9))), nw, nh,
                                  C.c_float(alpha), C.c_float(beta), C.c_float(gamma), _p(out), out.strides[0], _p(scratch), int(n_threads))
    if r != 0:
        raise ValueError("stitch_compose failed")
    return out
