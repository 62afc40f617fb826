"""caller-stream form of the device calls (SURVEY 8b's _async variants): dr3hip_set_stream makes the context enqueue on a stream the
caller owns; results are those of the context's own stream, and the work really is on the caller's stream"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_set_stream_runs_the_batch_on_the_callers_stream(dr3, O, synth, ctx):
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")            # the caller's own stream, made with the HIP runtime directly
    w, h, n_fr = 320, 200, 3
    frames = [synth.frame_pair(w, h, seed=40 + i)[0] for i in range(n_fr)]
    pts = synth.grid_points(w, h, 24, border=12)
    n = len(pts)
    P = np.ascontiguousarray(np.broadcast_to(pts[None], (n_fr - 1, n, 2)), np.float32)
    p = dr3.lk_params(win=(15, 15), max_level=2, max_iters=30, eps=0.01)

    def run():
        b = dr3.FrameBatch(ctx, w, h, win=(15, 15), max_level=2, capacity=n_fr)
        b.upload(0, np.stack(frames))
        b.build(0, n_fr, True)
        d_prev, d_next = ctx.to_device(P), ctx.to_device(P)
        d_st, d_err = ctx.alloc((n_fr - 1) * n), ctx.alloc((n_fr - 1) * n * 4)
        b.track(n_fr - 1, 0, d_prev, d_next, n, d_st, d_err, p)
        return b, d_next, d_st, d_err

    own = ctx.stream()
    b0, x0, s0, e0 = run()
    ctx.sync()
    ref = (x0.download((n_fr - 1, n, 2), np.float32), s0.download(((n_fr - 1) * n,), np.uint8), e0.download(((n_fr - 1) * n,), np.float32))

    assert hip.hipSetDevice(0) == 0
    ts = C.c_void_p()
    assert hip.hipStreamCreate(C.byref(ts)) == 0 and ts.value
    ctx.set_stream(ts.value)
    try:
        assert ctx.stream() == ts.value and ts.value != own
        b1, x1, s1, e1 = run()
        assert hip.hipStreamSynchronize(ts) == 0          # the CALLER's stream is what has to be waited for
        got = (x1.download((n_fr - 1, n, 2), np.float32), s1.download(((n_fr - 1) * n,), np.uint8), e1.download(((n_fr - 1) * n,), np.float32))
    finally:
        ctx.set_stream(None)
        hip.hipStreamDestroy(ts)
    assert ctx.stream() == own
    for a, c in zip(ref, got):
        assert np.array_equal(a.view(np.uint8), c.view(np.uint8))
    assert ref[1].sum() > n                   # the case tracks something
