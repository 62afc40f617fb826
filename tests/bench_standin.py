"""TEST-ONLY stand-in for the product package, used by tests/test_bench_launch.py through DR3_BENCH_STANDIN to rehearse
bench.py's launch path (`--gpus N` -> N child ranks -> barrier-bracketed timing -> rank 0's JSON line) on a box without
GPUs.  Nothing here computes anything: a "step" is a sleep whose length depends on the rank, so that the test can tell
MAX-over-ranks time from any single rank's.  bench.py labels a line produced this way "data": "standin ..."; the product
path never imports this module (tests/test_abi.py checks that nothing under 3dr_amd/ or include/ refers to tests/).
The one exception to "computes nothing" is the panorama paste: the canvas-sharded leg (bench.paste_sharded_leg) checks that the row
slabs of all ranks equal the canvas one rank pastes alone, so the stand-in pastes for real -- through the CPU checker (oracle/), on
the tiny tiles the test asks for -- cutting the rows [y0, y1) out of the whole-canvas result as dr3hip_panorama_add_slab defines."""
import os
import time

import numpy as np

PROF_PYR, PROF_LK, PROF_WARP, PROF_REMAP, PROF_HALF, PROF_PYR_L0, PROF_PANO_ADD, PROF_PANO_NORM, PROF_FAST, PROF_STITCH, PROF_ALIGN = range(11)
COMM_ID_BYTES = 128
STEP_SECONDS = 0.01


class _Lib:
    @staticmethod
    def dr3hip_device_count():
        return 0


def lib():
    return _Lib


def lk_params(**kw):
    return dict(kw)


class _Buf:
    def __init__(self, nbytes, arr=None):
        self.nbytes, self.ptr, self.arr = int(nbytes), 1, arr

    def download(self, shape, dtype, offset=0):
        if self.arr is not None:
            return np.frombuffer(np.ascontiguousarray(self.arr).tobytes(), dtype)[:int(np.prod(shape))].reshape(shape).copy()
        return np.ones(shape, dtype)

    def zero(self):
        self.arr = None

    def free(self):
        pass


def panorama_bbox(M, w, h):
    from oracle import oracle as O
    return O.panorama_bbox(M, w, h)


class Context:
    def __init__(self, device=0):
        self.device = device
        self.rank = int(os.environ.get("RANK", "0"))
        self.prof = {}

    def sync(self): pass
    def close(self): pass
    def alloc(self, nbytes): return _Buf(nbytes)
    def to_device(self, arr): return _Buf(np.asarray(arr).nbytes, np.asarray(arr))

    def panorama_add_slab(self, d_img, w, h, istride, d_slab, cw, ch, cstride, M, feathering_width, y0, y1):
        from oracle import oracle as O
        full = np.zeros((ch, cw, 4), np.float32)
        if d_slab.arr is not None:
            full[y0:y1] = d_slab.arr
        O.panorama_add(d_img.arr, full, M, feathering_width)          # the whole-canvas geometry; the slab keeps its own rows
        d_slab.arr = full[y0:y1].copy()

    def panorama_normalize(self, d_canvas, cw, ch, cstride, d_out, ostride):
        from oracle import oracle as O
        d_out.arr = O.normalize_canvas(d_canvas.arr if d_canvas.arr is not None else np.zeros((ch, cw, 4), np.float32))
    def profile_enable(self, on=True): pass
    def profile_reset(self): self.prof = {}

    def profile_read(self, which):
        ms, n = self.prof.get(which, (0.0, 0))
        return (ms, n) if n else (1.0, 1)


class FrameBatch:
    def __init__(self, ctx, w, h, win=(21, 21), max_level=3, capacity=2):
        self.ctx, self.w, self.h_px, self.capacity, self.levels = ctx, w, h, capacity, max_level + 1

    def upload(self, first, frames): pass
    def close(self): pass

    def build(self, first, count, with_deriv=True):
        ms, n = self.ctx.prof.get(PROF_PYR, (0.0, 0)); self.ctx.prof[PROF_PYR] = (ms + 0.1, n + 1)
        self.ctx.prof[PROF_PYR_L0] = self.ctx.prof[PROF_PYR]

    def build_ahead(self, first, count, with_deriv=True):
        self.build(first, count, with_deriv)

    def join(self): pass

    def track(self, n_pairs, first_slot, d_prev, d_next, n_pts, d_st, d_err, params, *a, **k):
        t = STEP_SECONDS * (1 + 2 * self.ctx.rank)              # rank r takes (1 + 2r) x 10 ms: the job time is the last rank's
        time.sleep(t)
        ms, n = self.ctx.prof.get(PROF_LK, (0.0, 0)); self.ctx.prof[PROF_LK] = (ms + t * 1e3, n + 1)
        return self.levels - 1
