"""3dr_amd -- MI355X (gfx950) implementation of 3DR's dense hot path, behind a C ABI.

The product is ``libdr3hip.so`` (hand-written HIP kernels + the C entry points of
``include/dr3hip.h``).  This Python package is a thin ctypes mirror of that ABI used by the tests
and the bench; its function names follow the calls the reference makes
(``cv::calcOpticalFlowPyrLK`` at src/initialization.cpp:608, ``cv::warpPerspective`` at
src/stitch.cpp:73-74 / src/panorama.cpp:192, ``cv::remap`` at src/utils.cpp:191,
``utils::create_img_pyramid`` at src/utils.cpp:421, ``FastDetector::detect`` at src/features.cpp:43,
``Panorama::add_img_to_canvas`` / ``normalize_canvas`` at src/panorama.cpp:156-229, ``Stitch::process`` at
src/stitch.cpp:33-80, ``utils::warp_spherical`` / ``warp_cylindrical`` at src/utils.cpp:196-272,
``InitHelper::CheckFundamental`` at src/initialization.cpp:171-249).

There is NO CPU fallback: every compute entry point needs the built library and a HIP device and
raises ``Dr3HipError`` otherwise.  (The directory name starts with a digit, so import it with
``importlib.import_module("3dr_amd")``.)
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DR3HIP_LIB") or os.path.join(_HERE, "libdr3hip.so")   # DR3HIP_LIB: kernel experiments (tools/) only
CSRC = os.path.join(_HERE, "csrc")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "dr3hip.h")

# constants of include/dr3hip.h
T_8UC1, T_8UC3, T_32FC4, T_8UC4, T_32FC1 = 0, 1, 2, 3, 4
LK_USE_INITIAL_FLOW, LK_GET_MIN_EIGENVALS = 4, 8
LK_DIAG_RESTAGES = 1 << 20      # diagnostic: d_iters carries the re-staging count of every (point, level) in bits 16..31
LK_LOST_START_OUTSIDE, LK_LOST_MIN_EIG, LK_LOST_ITERATE_OUTSIDE, LK_LOST_RESULT_OUTSIDE = 1, 2, 3, 4
WARP_INVERSE_MAP = 16
CRIT_COUNT, CRIT_EPS = 1, 2
HALF_TRUNC, HALF_SSE2, HALF_REF_SCALAR, HALF_REF_X86 = 0, 1, 2, 3
MAX_LEVELS, MAX_WIN = 8, 64
PROF_PYR, PROF_LK, PROF_WARP, PROF_REMAP, PROF_HALF, PROF_PYR_L0, PROF_PANO_ADD, PROF_PANO_NORM, PROF_FAST, PROF_STITCH, PROF_ALIGN = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10
COMM_ID_BYTES = 128
WARP_SPHERICAL, WARP_CYLINDRICAL = 1, 2
NO_POINT = np.float32(-1.0e6)


class Dr3HipError(RuntimeError):
    pass


class LKParams(C.Structure):
    _fields_ = [("win_w", C.c_int), ("win_h", C.c_int), ("max_level", C.c_int),
                ("crit_type", C.c_int), ("max_iters", C.c_int), ("eps", C.c_double),
                ("flags", C.c_int), ("min_eig_threshold", C.c_double)]


class LKCounters(C.Structure):
    _fields_ = [("n_points", C.c_ulonglong), ("n_tracked", C.c_ulonglong),
                ("n_lost_start_outside", C.c_ulonglong), ("n_lost_min_eig", C.c_ulonglong),
                ("n_lost_iterate_outside", C.c_ulonglong), ("n_lost_result_outside", C.c_ulonglong),
                ("iters_total", C.c_ulonglong), ("iters_hist", C.c_ulonglong * 32)]


class AlignParams(C.Structure):
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("max_level", C.c_int), ("min_level", C.c_int), ("n_iter", C.c_int), ("eps", C.c_double)]


class LevelInfo(C.Structure):
    _fields_ = [("w", C.c_int), ("h", C.c_int),
                ("img_pitch", C.c_int), ("img_slot_step", C.c_size_t), ("img", C.c_void_p),
                ("der_pitch", C.c_int), ("der_slot_step", C.c_size_t), ("der", C.c_void_p)]


class Corner(C.Structure):
    _fields_ = [("x", C.c_int), ("y", C.c_int), ("level", C.c_int), ("score", C.c_float)]


class WarpJob(C.Structure):
    _fields_ = [("src", C.c_void_p), ("sw", C.c_int), ("sh", C.c_int), ("sstride", C.c_int),
                ("dst", C.c_void_p), ("dw", C.c_int), ("dh", C.c_int), ("dstride", C.c_int),
                ("M", C.c_float * 9)]


def source_hash():
    """sha256 over the sources libdr3hip.so is built from (3dr_amd/csrc/*.{hip,cpp,h}, its Makefile, include/dr3hip.h)"""
    import hashlib
    h = hashlib.sha256()
    files = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp", ".h")) and f != "build_stamp.h") + ["Makefile"]
    for f in files:
        h.update(f.encode()); h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(open(HEADER, "rb").read())
    return h.hexdigest()[:20]


def build(force=False, extra=""):
    """Compile every HIP source into 3dr_amd/libdr3hip.so (hipcc --offload-arch=gfx950) and PROVE it: the library carries the
    hash of the sources it was built from (build_stamp.h -> dr3hip_build_info()); a stale library cannot pass for a fresh one."""
    stamp, want = os.path.join(CSRC, "build_stamp.h"), '#define DR3HIP_SOURCE_HASH "%s"\n' % source_hash()
    if not os.path.exists(stamp) or open(stamp).read() != want:
        open(stamp, "w").write(want)                # newer than dr3hip_api.o -> make recompiles it and relinks
    cmd = ["make", "-C", CSRC, "-j4"]
    if extra:
        cmd.append("EXTRA=" + extra)
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean"], stdout=subprocess.DEVNULL)
        open(stamp, "w").write(want)
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    built = built_from()
    if built != source_hash():
        raise Dr3HipError("libdr3hip.so reports sources %s, the tree holds %s: the build did not produce the library" % (built, source_hash()))
    return LIB_PATH


def built_from():
    """the source hash the library on disk was built from (read through a fresh handle, no GPU needed)"""
    L = C.CDLL(LIB_PATH)
    L.dr3hip_build_info.restype = C.c_char_p
    return L.dr3hip_build_info().decode()


_lib = None

_SIGNATURES = None


def _signatures():
    vp, ip, fp_ = C.c_void_p, C.c_int, C.POINTER(C.c_float)
    P = C.POINTER
    return {
        "dr3hip_version": ([], ip),
        "dr3hip_build_info": ([], C.c_char_p),
        "dr3hip_lk_params_reference": ([P(LKParams)], None),
        "dr3hip_device_count": ([], ip),
        "dr3hip_init": ([ip, P(vp)], ip),
        "dr3hip_destroy": ([vp], None),
        "dr3hip_last_error": ([vp], C.c_char_p),
        "dr3hip_sync": ([vp], ip),
        "dr3hip_stream": ([vp], vp),
        "dr3hip_set_stream": ([vp, vp], ip),
        "dr3hip_malloc": ([vp, C.c_size_t, P(vp)], ip),
        "dr3hip_free": ([vp, vp], ip),
        "dr3hip_memcpy_h2d": ([vp, vp, vp, C.c_size_t], ip),
        "dr3hip_memcpy_d2h": ([vp, vp, vp, C.c_size_t], ip),
        "dr3hip_memset": ([vp, vp, ip, C.c_size_t], ip),
        "dr3hip_lk_track": ([vp, vp, vp, ip, ip, ip, ip, vp, vp, ip, vp, vp, P(LKParams)], ip),
        "dr3hip_lk_effective_max_level": ([ip] * 5, ip),
        "dr3hip_halfsample_pyramid": ([vp, vp, ip, ip, ip, ip, ip, P(vp), P(ip)], ip),
        "dr3hip_warp_perspective": ([vp, vp, ip, ip, ip, ip, vp, ip, ip, ip, vp, ip], ip),
        "dr3hip_remap_bilinear": ([vp, vp, ip, ip, ip, ip, vp, ip, ip, ip, vp, vp, ip], ip),
        "dr3hip_batch_create": ([vp, ip, ip, ip, ip, ip, ip, P(vp)], ip),
        "dr3hip_batch_destroy": ([vp], None),
        "dr3hip_batch_levels": ([vp], ip),
        "dr3hip_batch_bytes": ([vp], C.c_size_t),
        "dr3hip_batch_level_info": ([vp, ip, P(LevelInfo)], ip),
        "dr3hip_batch_upload": ([vp, ip, ip, vp, ip, C.c_size_t], ip),
        "dr3hip_batch_download": ([vp, ip, ip, ip, vp, ip], ip),
        "dr3hip_batch_pad_level0": ([vp, ip, ip], ip),
        "dr3hip_batch_build": ([vp, ip, ip, ip], ip),
        "dr3hip_batch_build_ahead": ([vp, ip, ip, ip], ip),
        "dr3hip_batch_join": ([vp], ip),
        "dr3hip_batch_track": ([vp, ip, ip, vp, vp, vp, vp, ip, vp, vp, vp, P(LKParams)], ip),
        "dr3hip_batch_track_diag": ([vp, ip, ip, vp, vp, vp, vp, ip, vp, vp, vp, vp, P(LKParams)], ip),
        "dr3hip_lk_stats": ([vp, C.c_longlong, vp, vp, vp, ip, P(LKCounters)], ip),
        "dr3hip_batch_align": ([vp, ip, ip, vp, vp, vp, ip, P(AlignParams), vp, vp, vp, vp], ip),
        "dr3hip_warp_perspective_batch": ([vp, P(WarpJob), ip, ip, ip], ip),
        "dr3hip_warp_plan": ([P(WarpJob), ip, ip, P(C.c_int), P(C.c_int), P(C.c_int)], ip),
        "dr3hip_batch_upload_async": ([vp, ip, ip, vp, ip, C.c_size_t], ip),
        "dr3hip_batch_upload_wait": ([vp], ip),
        "dr3hip_host_alloc": ([vp, C.c_size_t, P(vp)], ip),
        "dr3hip_host_free": ([vp, vp], ip),
        "dr3hip_download_async": ([vp, vp, vp, C.c_size_t], ip),
        "dr3hip_download_wait": ([vp], ip),
        "dr3hip_remap_bilinear_dev": ([vp, vp, ip, ip, ip, ip, vp, ip, ip, ip, vp, vp, ip], ip),
        "dr3hip_halfsample_dev": ([vp, vp, ip, ip, ip, vp, ip, ip], ip),
        "dr3hip_fast_grid_size": ([ip, ip, ip, P(ip), P(ip)], ip),
        "dr3hip_fast_detect": ([vp, vp, ip, ip, ip, ip, ip, ip, C.c_double, vp, ip, P(Corner)], ip),
        "dr3hip_batch_detect": ([vp, ip, ip, ip, ip, ip, C.c_double, ip, vp, vp], ip),
        "dr3hip_fast_score_map": ([vp, vp, ip, ip, ip, ip, vp], ip),
        "dr3hip_check_fundamental": ([vp, vp, ip, vp, vp, ip, C.c_float, vp, vp], ip),
        "dr3hip_check_fundamental_dev": ([vp, vp, ip, vp, vp, vp, ip, C.c_float, vp, vp], ip),
        "dr3hip_normalize_points": ([vp, vp, ip, vp, vp], ip),
        "dr3hip_normalize_points_dev": ([vp, vp, vp, ip, ip, vp, vp], ip),
        "dr3hip_warp_local": ([vp, vp, ip, ip, ip, ip, vp, ip, ip, C.c_float], ip),
        "dr3hip_warp_local_dev": ([vp, vp, ip, ip, ip, ip, vp, ip, ip, C.c_float], ip),
        "dr3hip_stitch_geometry": ([vp, ip, ip, vp, vp, vp, P(ip), P(ip)], ip),
        "dr3hip_stitch_compose": ([vp, vp, ip, ip, ip, vp, ip, ip, ip, vp, vp, ip, ip, C.c_float, C.c_float, C.c_float, vp, ip], ip),
        "dr3hip_stitch_compose_dev": ([vp, vp, ip, ip, ip, vp, ip, ip, ip, vp, vp, ip, ip, C.c_float, C.c_float, C.c_float, vp, ip], ip),
        "dr3hip_stitch_plan": ([vp, ip, ip, ip, vp, ip, ip, ip, vp, vp, ip, ip, C.c_float, C.c_float, C.c_float, P(ip)], ip),
        "dr3hip_panorama_plan": ([vp, ip, ip, ip, ip, ip, vp, ip, ip, P(ip), P(ip), P(ip)], ip),
        "dr3hip_panorama_bbox": ([vp, ip, ip, P(ip), P(ip), P(ip), P(ip)], ip),
        "dr3hip_panorama_ramp": ([ip, ip, ip, vp], ip),
        "dr3hip_panorama_add": ([vp, vp, ip, ip, ip, vp, ip, ip, ip, vp, ip], ip),
        "dr3hip_panorama_add_slab": ([vp, vp, ip, ip, ip, vp, ip, ip, ip, vp, ip, ip, ip], ip),
        "dr3hip_panorama_normalize": ([vp, vp, ip, ip, ip, vp, ip], ip),
        "dr3hip_comm_unique_id": ([vp, vp], ip),
        "dr3hip_comm_init": ([vp, vp, ip, ip], ip),
        "dr3hip_comm_destroy": ([vp], ip),
        "dr3hip_comm_world": ([vp], ip),
        "dr3hip_comm_rank": ([vp], ip),
        "dr3hip_bcast": ([vp, vp, C.c_size_t, ip], ip),
        "dr3hip_batch_bcast_slot": ([vp, ip, ip, ip], C.c_longlong),
        "dr3hip_timer_start": ([vp], ip),
        "dr3hip_timer_stop": ([vp, fp_], ip),
        "dr3hip_profile_enable": ([vp, ip], ip),
        "dr3hip_profile_read": ([vp, ip, P(C.c_double), P(ip)], ip),
        "dr3hip_profile_reset": ([vp], ip),
    }


def exported_symbols():
    """names of every entry point include/dr3hip.h declares (used by the CPU-side ABI test)"""
    return sorted(_signatures().keys())


def lib():
    """dlopen libdr3hip.so and bind every symbol; loud failure if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Dr3HipError("libdr3hip.so is not built (%s missing): run __graft_entry__.build() "
                              "or `make -C 3dr_amd/csrc`; there is no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (argt, rest) in _signatures().items():
            fn = getattr(L, name)          # AttributeError if the symbol is not exported
            fn.argtypes = argt
            fn.restype = rest
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def lk_params(win=(21, 21), max_level=3, max_iters=30, eps=0.01, crit_type=CRIT_COUNT | CRIT_EPS,
              flags=0, min_eig_threshold=1e-4):
    return LKParams(int(win[0]), int(win[1]), int(max_level), int(crit_type), int(max_iters),
                    float(eps), int(flags), float(min_eig_threshold))


def align_params(fx, fy, cx, cy, max_level=4, min_level=2, n_iter=10, eps=1e-8):
    """parameters of dr3hip_batch_align (sparse image alignment, an extension specified by this build)"""
    return AlignParams(float(fx), float(fy), float(cx), float(cy), int(max_level), int(min_level), int(n_iter), float(eps))


def lk_params_reference():
    """the literals of src/initialization.cpp:593-613"""
    p = LKParams()
    lib().dr3hip_lk_params_reference(C.byref(p))
    return p


def lk_effective_max_level(w, h, win_w, win_h, max_level):
    return lib().dr3hip_lk_effective_max_level(w, h, win_w, win_h, max_level)


def _img_type(img):
    if img.dtype == np.uint8:
        return T_8UC1 if img.ndim == 2 else {1: T_8UC1, 3: T_8UC3, 4: T_8UC4}[img.shape[2]]
    if img.dtype == np.float32:
        return T_32FC1 if img.ndim == 2 else {1: T_32FC1, 4: T_32FC4}[img.shape[2]]
    raise TypeError("unsupported image dtype %s" % img.dtype)


class DeviceBuffer:
    """a device allocation owned by a Context"""

    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, int(nbytes)
        p = C.c_void_p()
        ctx._chk(lib().dr3hip_malloc(ctx.h, self.nbytes, C.byref(p)))
        self.ptr = p.value

    def upload(self, arr, offset=0):
        arr = np.ascontiguousarray(arr)
        assert offset + arr.nbytes <= self.nbytes
        self.ctx._chk(lib().dr3hip_memcpy_h2d(self.ctx.h, self.ptr + offset, _p(arr), arr.nbytes))
        return self

    def download(self, shape, dtype, offset=0):
        out = np.empty(shape, dtype)
        assert offset + out.nbytes <= self.nbytes
        self.ctx._chk(lib().dr3hip_memcpy_d2h(self.ctx.h, _p(out), self.ptr + offset, out.nbytes))
        return out

    def zero(self):
        self.ctx._chk(lib().dr3hip_memset(self.ctx.h, self.ptr, 0, self.nbytes))

    def free(self):
        if self.ptr:
            lib().dr3hip_free(self.ctx.h, self.ptr)
            self.ptr = None


class Context:
    """one dr3hip_ctx (one GPU, one stream).  Not thread-safe."""

    def __init__(self, device=0):
        L = lib()
        h = C.c_void_p()
        rc = L.dr3hip_init(int(device), C.byref(h))
        if rc != 0:
            raise Dr3HipError("dr3hip_init(%d) failed (%d): %s" % (device, rc, L.dr3hip_last_error(None).decode()))
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            lib().dr3hip_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc):
        if rc < 0:
            raise Dr3HipError("dr3hip error %d: %s" % (rc, lib().dr3hip_last_error(self.h).decode()))
        return rc

    def sync(self):
        self._chk(lib().dr3hip_sync(self.h))

    def lk_stats(self, n_points, d_status, d_reason=None, d_iters=None, levels=0):
        """the tracker's counters over device-resident outputs (dr3hip_lk_stats) as a dict"""
        ptr = lambda b: None if b is None else (b.ptr if isinstance(b, DeviceBuffer) else b)
        c = LKCounters()
        self._chk(lib().dr3hip_lk_stats(self.h, int(n_points), ptr(d_status), ptr(d_reason), ptr(d_iters), int(levels), C.byref(c)))
        out = {k: int(getattr(c, k)) for k, _ in LKCounters._fields_ if k != "iters_hist"}
        out["iters_hist"] = [int(v) for v in c.iters_hist]
        return out

    def stream(self):
        """the hipStream_t (as an integer) the context enqueues on"""
        return lib().dr3hip_stream(self.h) or 0

    def set_stream(self, hip_stream):
        """enqueue everything on the caller's hipStream_t from now on (an integer handle);
        None / 0 returns to the context's own stream"""
        self._chk(lib().dr3hip_set_stream(self.h, C.c_void_p(int(hip_stream)) if hip_stream else None))

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def fast_score_map(self, img, threshold=20):
        """FAST-10 corner scores of one gray image (0 = no corner)"""
        img = np.ascontiguousarray(img, np.uint8)
        sm = np.zeros(img.shape, np.uint8)
        self._chk(lib().dr3hip_fast_score_map(self.h, _p(img), img.shape[1], img.shape[0], img.strides[0], int(threshold), _p(sm)))
        return sm

    def fast_detect(self, img, n_levels=3, cell_size=30, fast_threshold=20, detection_threshold=20.0, occupancy=None,
                    mode=HALF_REF_X86):
        """FastDetector::detect (src/features.cpp:43-98) -> (x, y, score, level) arrays per grid cell; level -1 = empty"""
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        gc, gr = C.c_int(), C.c_int()
        lib().dr3hip_fast_grid_size(w, h, int(cell_size), C.byref(gc), C.byref(gr))
        n = gc.value * gr.value
        cells = (Corner * n)()
        occ = None if occupancy is None else np.ascontiguousarray(occupancy, np.uint8)
        self._chk(lib().dr3hip_fast_detect(self.h, _p(img), w, h, img.strides[0], int(n_levels), int(cell_size), int(fast_threshold),
                                           float(detection_threshold), None if occ is None else _p(occ), int(mode), cells))
        a = np.frombuffer(cells, dtype=np.dtype([("x", np.int32), ("y", np.int32), ("level", np.int32), ("score", np.float32)]))
        return a["x"].copy(), a["y"].copy(), a["score"].copy(), a["level"].copy()

    def check_fundamental(self, F21, pts1, pts2, sigma=1.0, want_inliers=True):
        """InitHelper::CheckFundamental (src/initialization.cpp:171-249) for all hypotheses: F21 (n_hyp, 3, 3)"""
        F = np.ascontiguousarray(F21, np.float32).reshape(-1, 9)
        p1 = np.ascontiguousarray(pts1, np.float32).reshape(-1, 2); p2 = np.ascontiguousarray(pts2, np.float32).reshape(-1, 2)
        assert len(p1) == len(p2)
        scores = np.zeros(len(F), np.float32)
        inl = np.zeros((len(F), len(p1)), np.uint8) if want_inliers else None
        self._chk(lib().dr3hip_check_fundamental(self.h, _p(F), len(F), _p(p1), _p(p2), len(p1), C.c_float(sigma), _p(scores),
                                                 None if inl is None else _p(inl)))
        return scores, inl

    def check_fundamental_dev(self, d_F21, n_hyp, d_pts1, d_pts2, d_mask, n, sigma, d_scores, d_inliers=None):
        """device-resident CheckFundamental (d_* = DeviceBuffer or raw pointers); enqueued, call sync() to wait"""
        ptr = lambda b: None if b is None else (b.ptr if isinstance(b, DeviceBuffer) else b)
        self._chk(lib().dr3hip_check_fundamental_dev(self.h, ptr(d_F21), int(n_hyp), ptr(d_pts1), ptr(d_pts2), ptr(d_mask), int(n),
                                                     C.c_float(sigma), ptr(d_scores), ptr(d_inliers)))

    def normalize_points(self, xy):
        """InitHelper::Normalize (src/initialization.cpp:365-410) -> (normalised points (n,2) f32, T (3,3) f32)"""
        xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
        out = np.zeros_like(xy); T = np.zeros(9, np.float32)
        self._chk(lib().dr3hip_normalize_points(self.h, _p(xy), len(xy), _p(out), _p(T)))
        return out, T.reshape(3, 3)

    def normalize_points_dev(self, d_xy, d_mask, n, n_sets, d_out_xy, d_T):
        ptr = lambda b: None if b is None else (b.ptr if isinstance(b, DeviceBuffer) else b)
        self._chk(lib().dr3hip_normalize_points_dev(self.h, ptr(d_xy), ptr(d_mask), int(n), int(n_sets), ptr(d_out_xy), ptr(d_T)))

    def stitch_compose(self, left, right, H, alpha=0.5, beta=0.5, gamma=0.0):
        """Stitch::process after align_pair (src/stitch.cpp:33-80): geometry on the host, both warps and the addWeighted
        blend in one kernel; returns the stitched 8UC3 image"""
        left = np.ascontiguousarray(left, np.uint8); right = np.ascontiguousarray(right, np.uint8)
        Hinv, T, M, (nw, nh) = stitch_geometry(H, right.shape[1], right.shape[0])
        out = np.zeros((nh, nw, 3), np.uint8)
        self._chk(lib().dr3hip_stitch_compose(self.h, _p(left), left.shape[1], left.shape[0], left.strides[0],
                                              _p(right), right.shape[1], right.shape[0], right.strides[0],
                                              _p(np.ascontiguousarray(T.reshape(9))), _p(np.ascontiguousarray(M.reshape(9))), nw, nh,
                                              C.c_float(alpha), C.c_float(beta), C.c_float(gamma), _p(out), out.strides[0]))
        return out

    def warp_spherical(self, img, f):
        """utils::warp_spherical (src/utils.cpp:196-201): map generation fused with the remap"""
        return self._warp_local(img, WARP_SPHERICAL, f)

    def warp_cylindrical(self, img, f):
        """utils::warp_cylindrical (src/utils.cpp:267-272)"""
        return self._warp_local(img, WARP_CYLINDRICAL, f)

    def _warp_local(self, img, kind, f):
        img = np.ascontiguousarray(img)
        t = _img_type(img)
        dst = np.zeros_like(img)
        self._chk(lib().dr3hip_warp_local(self.h, _p(img), img.shape[1], img.shape[0], img.strides[0], t,
                                          _p(dst), dst.strides[0], int(kind), C.c_float(f)))
        return dst

    # ---- panorama paste (reference src/panorama.cpp:156-229) -----------------------------------
    def panorama_add(self, d_img, w, h, istride, d_canvas, cw, ch, cstride, M, feathering_width):
        """device-resident Panorama::add_img_to_canvas (fused warp + feather + accumulate)"""
        M = np.ascontiguousarray(M, np.float32).reshape(9)
        ptr = lambda b: b.ptr if isinstance(b, DeviceBuffer) else b
        self._chk(lib().dr3hip_panorama_add(self.h, ptr(d_img), w, h, istride, ptr(d_canvas), cw, ch, cstride,
                                            _p(M), int(feathering_width)))

    def panorama_add_slab(self, d_img, w, h, istride, d_slab, cw, ch, cstride, M, feathering_width, y0, y1):
        """the paste restricted to canvas rows [y0, y1) held in d_slab (one GPU's share of the canvas)"""
        M = np.ascontiguousarray(M, np.float32).reshape(9)
        ptr = lambda b: b.ptr if isinstance(b, DeviceBuffer) else b
        self._chk(lib().dr3hip_panorama_add_slab(self.h, ptr(d_img), w, h, istride, ptr(d_slab), cw, ch, cstride,
                                                 _p(M), int(feathering_width), int(y0), int(y1)))

    def panorama_normalize(self, d_canvas, cw, ch, cstride, d_out, ostride):
        ptr = lambda b: b.ptr if isinstance(b, DeviceBuffer) else b
        self._chk(lib().dr3hip_panorama_normalize(self.h, ptr(d_canvas), cw, ch, cstride, ptr(d_out), ostride))

    def paste_images(self, images, Ms, canvas_size, feathering_width):
        """Panorama::paste_images + normalize_canvas (src/panorama.cpp:143-154, :215-229) for host images:
        returns (weighted canvas (H, W, 4) f32, panorama (H, W, 3) u8).  Ms are the forward homographies
        already multiplied by top_left, as at :148."""
        cw, ch = canvas_size
        d_canvas = self.alloc(cw * ch * 16)
        d_canvas.zero()
        d_out = self.alloc(cw * ch * 3)
        try:
            for img, M in zip(images, Ms):
                img = np.ascontiguousarray(img, np.uint8)
                d_img = self.to_device(img)
                self.panorama_add(d_img, img.shape[1], img.shape[0], img.strides[0], d_canvas, cw, ch, cw * 16, M, feathering_width)
                self.sync()
                d_img.free()
            self.panorama_normalize(d_canvas, cw, ch, cw * 16, d_out, cw * 3)
            return d_canvas.download((ch, cw, 4), np.float32), d_out.download((ch, cw, 3), np.uint8)
        finally:
            d_canvas.free(); d_out.free()

    # ---- multi-GPU: RCCL broadcast (include/dr3hip.h "multi-GPU") -----------------------------
    def comm_unique_id(self):
        """root only: the 128 bytes every rank passes to comm_init"""
        buf = C.create_string_buffer(COMM_ID_BYTES)
        self._chk(lib().dr3hip_comm_unique_id(self.h, buf))
        return buf.raw

    def comm_init(self, unique_id, world, rank):
        assert len(unique_id) == COMM_ID_BYTES
        self._chk(lib().dr3hip_comm_init(self.h, C.create_string_buffer(bytes(unique_id), COMM_ID_BYTES), int(world), int(rank)))

    def comm_destroy(self):
        self._chk(lib().dr3hip_comm_destroy(self.h))

    def comm_world(self):
        return lib().dr3hip_comm_world(self.h)

    def bcast(self, buf, nbytes, root=0):
        self._chk(lib().dr3hip_bcast(self.h, buf.ptr if isinstance(buf, DeviceBuffer) else buf, int(nbytes), int(root)))

    def host_alloc(self, shape, dtype=np.uint8):
        """a numpy array in pinned host memory (dr3hip_host_alloc); release it with host_free(array)"""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        self._chk(lib().dr3hip_host_alloc(self.h, n, C.byref(p)))
        arr = np.frombuffer((C.c_uint8 * max(n, 1)).from_address(p.value), dtype=dtype, count=int(np.prod(shape))).reshape(shape)
        self._pinned = getattr(self, "_pinned", {}); self._pinned[arr.ctypes.data] = p.value
        return arr

    def host_free(self, arr):
        p = getattr(self, "_pinned", {}).pop(arr.ctypes.data, None)
        if p is not None:
            self._chk(lib().dr3hip_host_free(self.h, p))

    def download_async(self, host_arr, dbuf, nbytes=None, offset=0):
        """device -> host on the download stream, behind everything enqueued on the compute stream so far (dr3hip_download_async)"""
        self._chk(lib().dr3hip_download_async(self.h, host_arr.ctypes.data, dbuf.ptr + offset, int(nbytes if nbytes is not None else host_arr.nbytes)))

    def download_wait(self):
        self._chk(lib().dr3hip_download_wait(self.h))

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        return DeviceBuffer(self, max(arr.nbytes, 1)).upload(arr)

    # ---- timing --------------------------------------------------------------------------------
    def timer_start(self):
        self._chk(lib().dr3hip_timer_start(self.h))

    def timer_stop(self):
        ms = C.c_float()
        self._chk(lib().dr3hip_timer_stop(self.h, C.byref(ms)))
        return ms.value

    def profile_enable(self, on=True):
        self._chk(lib().dr3hip_profile_enable(self.h, int(on)))

    def profile_reset(self):
        self._chk(lib().dr3hip_profile_reset(self.h))

    def profile_read(self, which):
        ms, n = C.c_double(), C.c_int()
        self._chk(lib().dr3hip_profile_read(self.h, which, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    # ---- host-buffer drop-in calls -----------------------------------------------------------------
    def calcOpticalFlowPyrLK(self, prev, nxt, prev_pts, next_pts=None, params=None):
        """cv::calcOpticalFlowPyrLK (src/initialization.cpp:608-613).
        Returns (next_pts (n,2) f32, status (n,) u8, err (n,) f32, effective_max_level)."""
        prev = np.ascontiguousarray(prev); nxt = np.ascontiguousarray(nxt)
        if prev.dtype != np.uint8 or prev.ndim != 2 or prev.shape != nxt.shape or nxt.dtype != np.uint8:
            raise ValueError("prev/next must be equal-sized 8-bit gray images")
        p = params if params is not None else lk_params()
        prev_pts = np.ascontiguousarray(prev_pts, np.float32).reshape(-1, 2)
        n = prev_pts.shape[0]
        out = (prev_pts.copy() if next_pts is None
               else np.ascontiguousarray(next_pts, np.float32).reshape(-1, 2).copy())
        status = np.zeros(n, np.uint8)
        err = np.zeros(n, np.float32)
        h, w = prev.shape
        L = self._chk(lib().dr3hip_lk_track(self.h, _p(prev), _p(nxt), w, h, prev.strides[0], nxt.strides[0],
                                            _p(prev_pts), _p(out), n, _p(status), _p(err), C.byref(p)))
        return out, status, err, L

    def warpPerspective(self, src, M, dsize, flags=0):
        """cv::warpPerspective(src, dst, M, dsize=(W,H)) INTER_LINEAR / BORDER_CONSTANT 0
        (src/stitch.cpp:73-74, src/panorama.cpp:192)."""
        src = np.ascontiguousarray(src)
        t = _img_type(src)
        sh, sw = src.shape[:2]
        dw, dh = int(dsize[0]), int(dsize[1])
        dst = np.empty((dh, dw) + src.shape[2:], src.dtype)
        M = np.ascontiguousarray(M, np.float32).reshape(9)
        self._chk(lib().dr3hip_warp_perspective(self.h, _p(src), sw, sh, src.strides[0], t, _p(dst), dw, dh,
                                                dst.strides[0], _p(M), int(flags)))
        return dst

    def remap(self, src, mapx, mapy):
        """cv::remap(src, dst, mapx, mapy, INTER_LINEAR) (src/utils.cpp:191)."""
        src = np.ascontiguousarray(src)
        t = _img_type(src)
        sh, sw = src.shape[:2]
        mapx = np.ascontiguousarray(mapx, np.float32); mapy = np.ascontiguousarray(mapy, np.float32)
        dh, dw = mapx.shape
        dst = np.empty((dh, dw) + src.shape[2:], src.dtype)
        self._chk(lib().dr3hip_remap_bilinear(self.h, _p(src), sw, sh, src.strides[0], t, _p(dst), dw, dh,
                                              dst.strides[0], _p(mapx), _p(mapy), mapx.strides[0]))
        return dst

    def create_img_pyramid(self, img, n_levels, mode=HALF_REF_X86):
        """utils::create_img_pyramid (src/utils.cpp:421-430): list of n_levels u8 images."""
        img = np.ascontiguousarray(img)
        if img.dtype != np.uint8 or img.ndim != 2:
            raise ValueError("8-bit gray image expected")
        h, w = img.shape
        pyr = [img]
        for _ in range(1, n_levels):
            ph, pw = pyr[-1].shape
            pyr.append(np.zeros((ph // 2, pw // 2), np.uint8))
        ptrs = (C.c_void_p * n_levels)(*[a.ctypes.data for a in pyr])
        strides = (C.c_int * n_levels)(*[a.strides[0] if a.size else 1 for a in pyr])
        self._chk(lib().dr3hip_halfsample_pyramid(self.h, _p(img), w, h, img.strides[0], n_levels, int(mode),
                                                  ptrs, strides))
        return pyr


class FrameBatch:
    """device-resident frame slots with LK pyramids (dr3hip_batch)"""

    def __init__(self, ctx, w, h, win=(21, 21), max_level=3, capacity=2):
        self.ctx = ctx
        hb = C.c_void_p()
        ctx._chk(lib().dr3hip_batch_create(ctx.h, w, h, int(win[0]), int(win[1]), int(max_level), int(capacity),
                                           C.byref(hb)))
        self.h = hb
        self.w, self.h_px, self.capacity = w, h, capacity
        self.levels = lib().dr3hip_batch_levels(hb)
        self.nbytes = lib().dr3hip_batch_bytes(hb)

    def close(self):
        if getattr(self, "h", None):
            lib().dr3hip_batch_destroy(self.h)
            self.h = None

    def level_info(self, level):
        li = LevelInfo()
        self.ctx._chk(lib().dr3hip_batch_level_info(self.h, level, C.byref(li)))
        return li

    def upload(self, first, frames):
        """frames: (n, h, w) u8 (or a list of (h, w) arrays)"""
        frames = np.ascontiguousarray(np.asarray(frames, np.uint8))
        if frames.ndim == 2:
            frames = frames[None]
        n, h, w = frames.shape
        assert (h, w) == (self.h_px, self.w)
        self.ctx._chk(lib().dr3hip_batch_upload(self.h, first, n, _p(frames), frames.strides[1], frames.strides[0]))

    def pad_level0(self, first, count):
        """level 0 was written through the device pointers of level_info(): give it its border (dr3hip_batch_pad_level0)"""
        self.ctx._chk(lib().dr3hip_batch_pad_level0(self.h, int(first), int(count)))

    def upload_async(self, first, frames):
        """the streaming upload (dr3hip_batch_upload_async): returns at once; frames: (count, h, w) u8, C-contiguous, best in pinned
        memory (Context.host_alloc)"""
        frames = np.asarray(frames)
        if frames.ndim == 2:
            frames = frames[None]
        assert frames.dtype == np.uint8 and frames.flags.c_contiguous and frames.shape[1:] == (self.h_px, self.w)
        self.ctx._chk(lib().dr3hip_batch_upload_async(self.h, int(first), int(frames.shape[0]), frames.ctypes.data, self.w, self.w * self.h_px))

    def upload_wait(self):
        self.ctx._chk(lib().dr3hip_batch_upload_wait(self.h))

    def build(self, first, count, with_deriv=True):
        self.ctx._chk(lib().dr3hip_batch_build(self.h, first, count, int(with_deriv)))

    def build_ahead(self, first, count, with_deriv=True):
        """dr3hip_batch_build_ahead: levels >= 1 on the context's auxiliary stream, under what the context's stream does next"""
        self.ctx._chk(lib().dr3hip_batch_build_ahead(self.h, first, count, int(with_deriv)))

    def join(self):
        self.ctx._chk(lib().dr3hip_batch_join(self.h))

    def detect(self, first, count, n_levels=3, cell_size=30, fast_threshold=20, detection_threshold=20.0, mode=HALF_REF_X86,
               d_xy=None, d_corners=None):
        """batched FastDetector::detect on slots first..first+count-1; returns n_cells (device outputs in d_xy / d_corners)"""
        ptr = lambda x: None if x is None else (x.ptr if isinstance(x, DeviceBuffer) else x)
        return self.ctx._chk(lib().dr3hip_batch_detect(self.h, int(first), int(count), int(n_levels), int(cell_size), int(fast_threshold),
                                                       float(detection_threshold), int(mode), ptr(d_xy), ptr(d_corners)))

    def bcast_slot(self, slot, root=0, with_deriv=True):
        """RCCL broadcast of one slot's pyramid (+ derivative pyramid) from `root`; returns bytes moved"""
        return self.ctx._chk(lib().dr3hip_batch_bcast_slot(self.h, int(slot), int(root), int(with_deriv)))

    def download_image(self, slot, level):
        li = self.level_info(level)
        out = np.empty((li.h, li.w), np.uint8)
        self.ctx._chk(lib().dr3hip_batch_download(self.h, slot, level, 0, _p(out), out.strides[0]))
        return out

    def download_deriv(self, slot, level):
        li = self.level_info(level)
        out = np.empty((li.h, li.w, 2), np.int16)
        self.ctx._chk(lib().dr3hip_batch_download(self.h, slot, level, 1, _p(out), out.strides[0]))
        return out

    def align(self, n_pairs, first_slot, d_xyz, n_pts, params, d_T, d_n_meas=None, d_chi2=None, d_iters=None,
              d_ref_slot=None, d_cur_slot=None):
        """sparse image alignment of n_pairs frame pairs (dr3hip_batch_align); all d_* device buffers; enqueued"""
        ptr = lambda b: None if b is None else (b.ptr if isinstance(b, DeviceBuffer) else b)
        self.ctx._chk(lib().dr3hip_batch_align(self.h, int(n_pairs), int(first_slot), ptr(d_ref_slot), ptr(d_cur_slot), ptr(d_xyz),
                                               int(n_pts), C.byref(params), ptr(d_T), ptr(d_n_meas), ptr(d_chi2), ptr(d_iters)))

    def track(self, n_pairs, first_slot, d_prev_xy, d_next_xy, n_pts, d_status, d_err, params,
              d_prev_slot=None, d_next_slot=None, d_iters=None, d_reason=None):
        """all d_* are DeviceBuffer (or raw device pointers); returns the effective maxLevel.  d_reason (n_pairs x n_pts bytes)
        receives the LK_LOST_* code of every point (dr3hip_batch_track_diag)."""
        ptr = lambda b: None if b is None else (b.ptr if isinstance(b, DeviceBuffer) else b)
        if d_reason is not None:
            return self.ctx._chk(lib().dr3hip_batch_track_diag(self.h, n_pairs, first_slot, ptr(d_prev_slot), ptr(d_next_slot),
                                                               ptr(d_prev_xy), ptr(d_next_xy), n_pts, ptr(d_status),
                                                               ptr(d_err), ptr(d_iters), ptr(d_reason), C.byref(params)))
        return self.ctx._chk(lib().dr3hip_batch_track(self.h, n_pairs, first_slot, ptr(d_prev_slot), ptr(d_next_slot),
                                                      ptr(d_prev_xy), ptr(d_next_xy), n_pts, ptr(d_status),
                                                      ptr(d_err), ptr(d_iters), C.byref(params)))


def warp_batch(ctx, jobs, img_type, flags=0):
    """jobs: list of dicts(src, sw, sh, sstride, dst, dw, dh, dstride, M) with device pointers"""
    arr = (WarpJob * len(jobs))()
    for i, j in enumerate(jobs):
        arr[i].src, arr[i].sw, arr[i].sh, arr[i].sstride = j["src"], j["sw"], j["sh"], j["sstride"]
        arr[i].dst, arr[i].dw, arr[i].dh, arr[i].dstride = j["dst"], j["dw"], j["dh"], j["dstride"]
        M = np.asarray(j["M"], np.float32).reshape(9)
        for k in range(9):
            arr[i].M[k] = float(M[k])
    ctx._chk(lib().dr3hip_warp_perspective_batch(ctx.h, arr, len(jobs), int(img_type), int(flags)))


def warp_plan(job, img_type, flags=0):
    """which kernel dr3hip_warp_perspective_batch runs this job on: dict(kernel=0 direct gather | 2 LDS-staged strips, lds_bytes, tie_guard)"""
    q = WarpJob()
    q.src, q.sw, q.sh, q.sstride = job.get("src", 0), job["sw"], job["sh"], job["sstride"]
    q.dst, q.dw, q.dh, q.dstride = job.get("dst", 0), job["dw"], job["dh"], job["dstride"]
    M = np.asarray(job["M"], np.float32).reshape(9)
    for k in range(9):
        q.M[k] = float(M[k])
    k, l, g = C.c_int(), C.c_int(), C.c_int()
    rc = lib().dr3hip_warp_plan(C.byref(q), int(img_type), int(flags), C.byref(k), C.byref(l), C.byref(g))
    if rc:
        raise Dr3HipError("dr3hip_warp_plan: %d" % rc)
    return {"kernel": k.value, "lds_bytes": l.value, "tie_guard": g.value}


def panorama_bbox(M, w, h):
    """Panorama::set_bbox (src/panorama.cpp:112-138), host-only -> (min_x, min_y, max_x, max_y)"""
    M = np.ascontiguousarray(M, np.float32).reshape(9)
    o = [C.c_int() for _ in range(4)]
    rc = lib().dr3hip_panorama_bbox(_p(M), int(w), int(h), *[C.byref(v) for v in o])
    if rc != 0:
        raise Dr3HipError("dr3hip_panorama_bbox failed (%d)" % rc)
    return tuple(v.value for v in o)


def panorama_ramp(min_x, max_x, feathering_width):
    """feather vector of add_img_to_canvas (src/panorama.cpp:163-177), host-only"""
    ramp = np.zeros(max(max_x - min_x, 1), np.float32)
    n = lib().dr3hip_panorama_ramp(int(min_x), int(max_x), int(feathering_width), _p(ramp))
    if n < 0:
        raise Dr3HipError("dr3hip_panorama_ramp: the reference throws for this box (%d)" % n)
    return ramp[:n]


def stitch_plan(lw, lh, rw, rh, T, M, size, alpha=0.5, beta=0.5, gamma=0.0, lstride=None, rstride=None):
    """which form a fused stitch of these images takes (dr3hip_stitch_plan): 2 = strips, 0 = direct; strides default to the 64-byte
    aligned pitches the host-buffer entry point stages with"""
    k = C.c_int()
    ls = lstride if lstride is not None else (3 * lw + 63) // 64 * 64
    rs = rstride if rstride is not None else (3 * rw + 63) // 64 * 64
    rc = lib().dr3hip_stitch_plan(None, lw, lh, ls, None, rw, rh, rs, _p(np.ascontiguousarray(np.asarray(T, np.float32).reshape(9))),
                                  _p(np.ascontiguousarray(np.asarray(M, np.float32).reshape(9))), int(size[0]), int(size[1]),
                                  C.c_float(alpha), C.c_float(beta), C.c_float(gamma), C.byref(k))
    if rc != 0:
        raise Dr3HipError("dr3hip_stitch_plan failed: %d" % rc)
    return k.value


def panorama_plan(w, h, canvas_size, M, istride=None, slab=None):
    """which form the paste of a w x h image with the forward matrix M onto a canvas takes (dr3hip_panorama_plan): (kernel, coords, box)
    -- kernel 2 = strips / 0 = direct, coords 0 .. 3 = the coordinate form over the box, box = (x0, y0, x1, y1) the launch evaluates;
    the stride defaults to the 64-byte aligned pitch the host-buffer entry point stages with"""
    cw, ch = canvas_size
    k, c = C.c_int(), C.c_int()
    box = (C.c_int * 4)()
    st = istride if istride is not None else (3 * w + 63) // 64 * 64
    y0, y1 = slab if slab is not None else (0, ch)
    rc = lib().dr3hip_panorama_plan(None, int(w), int(h), int(st), int(cw), int(ch), _p(np.ascontiguousarray(np.asarray(M, np.float32).reshape(9))),
                                    int(y0), int(y1), C.byref(k), C.byref(c), box)
    if rc != 0:
        raise Dr3HipError("dr3hip_panorama_plan failed: %d" % rc)
    return k.value, c.value, tuple(box)


def stitch_geometry(H, w, h):
    """Stitch::process geometry (src/stitch.cpp:33-71), host-only -> (Hinv, T, M_right, (new_w, new_h))"""
    H = np.ascontiguousarray(H, np.float32).reshape(9)
    Hinv, T, M = (np.zeros(9, np.float32) for _ in range(3))
    nw, nh = C.c_int(), C.c_int()
    rc = lib().dr3hip_stitch_geometry(_p(H), int(w), int(h), _p(Hinv), _p(T), _p(M), C.byref(nw), C.byref(nh))
    if rc != 0:
        raise Dr3HipError("dr3hip_stitch_geometry: singular homography (%d)" % rc)
    return Hinv.reshape(3, 3), T.reshape(3, 3), M.reshape(3, 3), (nw.value, nh.value)
