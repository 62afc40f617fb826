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


def _track_all(dr3, ctx, b, n_fr, P, n, p):
    d_prev, d_next = ctx.to_device(P), ctx.to_device(P)
    d_st, d_err = ctx.alloc((n_fr - 1) * n), ctx.alloc((n_fr - 1) * n * 4)
    b.track(n_fr - 1, 0, d_prev, d_next, n, d_st, d_err, p)
    ctx.sync()
    out = (d_next.download((n_fr - 1, n, 2), np.float32), d_st.download(((n_fr - 1) * n,), np.uint8), d_err.download(((n_fr - 1) * n,), np.float32))
    for d in (d_prev, d_next, d_st, d_err):
        d.free()
    return out


@pytest.mark.parametrize("win", [(15, 15), (21, 21)])
def test_streaming_upload_and_pad_level0_equal_the_synchronous_upload(dr3, synth, ctx, win):
    """dr3hip_batch_upload_async (pageable and pinned sources, several ring chunks) and level 0 written through the level_info
    pointers + dr3hip_batch_pad_level0 leave the slots -- pixels AND borders, which only the tracker reads -- as dr3hip_batch_upload does"""
    import ctypes as C
    w, h, n_fr = 333, 207, 41
    base = [synth.frame_pair(w, h, seed=7 + i)[0] for i in range(5)]
    frames = np.stack([np.roll(base[i % 5], (i // 5, 2 * (i // 5)), (0, 1)) for i in range(n_fr)])
    pts = synth.grid_points(w, h, 24, border=4)                    # windows reach into the border
    n = len(pts)
    P = np.ascontiguousarray(np.broadcast_to(pts[None], (n_fr - 1, n, 2)), np.float32)
    p = dr3.lk_params(win=win, max_level=2, max_iters=30, eps=0.01)

    def fresh():
        return dr3.FrameBatch(ctx, w, h, win=win, max_level=2, capacity=n_fr)
    b = fresh(); b.upload(0, frames); b.build(0, n_fr, True)
    ref = _track_all(dr3, ctx, b, n_fr, P, n, p); b.close()
    assert ref[1].sum() > n

    pinned = ctx.host_alloc(frames.shape); pinned[...] = frames
    for src in (frames, pinned):
        b = fresh(); b.upload_async(0, src); b.build(0, n_fr, True)
        got = _track_all(dr3, ctx, b, n_fr, P, n, p); b.upload_wait(); b.close()
        assert all(np.array_equal(a.view(np.uint8), c.view(np.uint8)) for a, c in zip(ref, got))
    ctx.host_free(pinned)

    b = fresh()
    li = b.level_info(0)
    for s in range(n_fr):                                           # the caller's own way of filling level 0: rows at the plane's pitch
        for y in range(h):
            ctx._chk(dr3.lib().dr3hip_memcpy_h2d(ctx.h, li.img + s * li.img_slot_step + y * li.img_pitch, frames[s, y].ctypes.data, w))
    b.pad_level0(0, n_fr); b.build(0, n_fr, True)
    got = _track_all(dr3, ctx, b, n_fr, P, n, p); b.close()
    assert all(np.array_equal(a.view(np.uint8), c.view(np.uint8)) for a, c in zip(ref, got))


def test_upload_of_the_next_batch_never_overtakes_the_tracker_of_this_one(dr3, synth, ctx):
    """six batches go through ONE set of slots without a single host wait in between: upload_async -> build -> track -> download_async,
    each into result buffers of its own.  Every batch must come out as if it had run alone."""
    w, h, n_fr, rounds = 640, 360, 24, 6
    base = [synth.frame_pair(w, h, seed=90 + i)[0] for i in range(4)]
    sets = [np.stack([np.roll(base[(i + r) % 4], (r + i // 4, 3 * r + i), (0, 1)) for i in range(n_fr)]) for r in range(rounds)]
    pts = synth.grid_points(w, h, 20, border=10)
    n = len(pts)
    P = np.ascontiguousarray(np.broadcast_to(pts[None], (n_fr - 1, n, 2)), np.float32)
    p = dr3.lk_params(win=(15, 15), max_level=3, max_iters=30, eps=0.01)
    b = dr3.FrameBatch(ctx, w, h, win=(15, 15), max_level=3, capacity=n_fr)
    ref = []
    for r in range(rounds):
        b.upload(0, sets[r]); b.build(0, n_fr, True)
        ref.append(_track_all(dr3, ctx, b, n_fr, P, n, p))
    assert not np.array_equal(ref[0][0], ref[1][0])
    host = [ctx.host_alloc(s.shape) for s in sets]
    for hbuf, s in zip(host, sets):
        hbuf[...] = s
    d_prev = ctx.to_device(P)
    res = [(ctx.to_device(P), ctx.alloc((n_fr - 1) * n), ctx.alloc((n_fr - 1) * n * 4)) for _ in range(rounds)]
    out = [(ctx.host_alloc((n_fr - 1, n, 2), np.float32), ctx.host_alloc(((n_fr - 1) * n,), np.uint8)) for _ in range(rounds)]
    for r in range(rounds):                                         # no wait anywhere in this loop
        b.upload_async(0, host[r]); b.build(0, n_fr, True)
        b.track(n_fr - 1, 0, d_prev, res[r][0], n, res[r][1], res[r][2], p)
        ctx.download_async(out[r][0], res[r][0]); ctx.download_async(out[r][1], res[r][1])
    ctx.download_wait(); b.upload_wait()
    for r in range(rounds):
        assert np.array_equal(out[r][0].view(np.uint32), ref[r][0].view(np.uint32)) and np.array_equal(out[r][1], ref[r][1]), r
    ctx.sync()
    for hb in host:
        ctx.host_free(hb)
    for a, c in out:
        ctx.host_free(a); ctx.host_free(c)
    b.close()


def test_build_ahead_overlaps_the_tracker_without_changing_anything(dr3, O, synth, ctx):
    """dr3hip_batch_build_ahead: a ring of two slot groups -- the levels >= 1 of group k + 1 are built on the auxiliary stream under the
    tracker of group k.  Every consumer waits for the slots it reads, so positions, status and planes equal the serial build's; frames that
    change between rounds must show up in the results (no stale level is ever read)."""
    w, h, B = 320, 200, 6
    frames, _ = synth.frame_stream(2 * (B + 1) + 3, w, h)
    sets = [np.stack(frames[0:B + 1]), np.stack(frames[B + 1:2 * B + 2]), np.stack(frames[2:B + 3])]      # three different contents for a group
    pts = synth.grid_points(w, h, 24)
    n = len(pts)
    P = np.ascontiguousarray(np.broadcast_to(pts[None], (B, n, 2)), np.float32)
    pg = dr3.lk_params(win=(15, 15), max_level=3, max_iters=20, eps=0.01)
    fb = dr3.FrameBatch(ctx, w, h, win=(15, 15), max_level=3, capacity=2 * (B + 1))
    d_prev = ctx.to_device(P)
    res = [(ctx.to_device(P), ctx.alloc(B * n), ctx.alloc(B * n * 4)) for _ in range(2)]
    try:
        def serial(content):
            fb.upload(0, content); fb.build(0, B + 1, True)
            fb.track(B, 0, d_prev, res[0][0], n, res[0][1], res[0][2], pg); ctx.sync()
            return res[0][0].download((B, n, 2), np.float32).copy(), res[0][1].download((B, n), np.uint8).copy()
        want = [serial(c) for c in sets]
        # the ring: group g = slots g (B + 1) .. g (B + 1) + B
        order = [0, 1, 2, 0, 2, 1, 1, 0]                                   # contents given to the groups, round by round
        fb.upload(0, sets[order[0]]); fb.build(0, B + 1, True)
        for r in range(len(order) - 1):
            g, gn = r % 2, (r + 1) % 2
            fb.upload(gn * (B + 1), sets[order[r + 1]])                    # new frames for the other group ...
            fb.build_ahead(gn * (B + 1), B + 1, True)                      # ... level 0 now, levels 1.. under the tracker below
            fb.track(B, g * (B + 1), d_prev, res[g][0], n, res[g][1], res[g][2], pg)
            if r == 3:                                                     # a plane of the group being built ahead: the download waits for it
                got = fb.download_image(gn * (B + 1) + 2, 2)
                ref = O.pyr_down(O.pyr_down(sets[order[r + 1]][2]))
                assert np.array_equal(got, ref)
            ctx.sync()
            xy, st = res[g][0].download((B, n, 2), np.float32), res[g][1].download((B, n), np.uint8)
            assert np.array_equal(xy.view(np.uint32), want[order[r]][0].view(np.uint32)) and np.array_equal(st, want[order[r]][1]), r
        fb.join()
        for lvl in range(fb.levels):                                       # and the planes built ahead equal the oracle's pyramid
            img = sets[order[-1]][1]
            for _ in range(lvl):
                img = O.pyr_down(img)
            gl = ((len(order) - 1) % 2) * (B + 1) + 1
            assert np.array_equal(fb.download_image(gl, lvl), img) and np.array_equal(fb.download_deriv(gl, lvl), O.scharr(img)), lvl
    finally:
        d_prev.free()
        for t in res:
            for d in t:
                d.free()
        fb.close()
