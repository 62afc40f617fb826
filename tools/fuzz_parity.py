"""randomised parity sweep, GPU vs oracle (bit-exact bars only): LK over random sizes / windows / levels / flags with
adversarial sub-pixel fractions, warps over random homographies and types, FAST over random images.
usage: python tools/fuzz_parity.py [n_cases] [seed]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dr3 = importlib.import_module("3dr_amd"); synth = importlib.import_module("3dr_amd.synth")
from oracle import oracle as O

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = dr3.Context(0)
f32 = np.float32
bad = 0


def adversarial_points(w, h, n):
    p = np.stack([rng.uniform(-4, w + 4, n), rng.uniform(-4, h + 4, n)], 1)
    kind = rng.integers(0, 6, n)
    fr = np.array([0.0, 0.5, 2.0 ** -15, 1 - 2.0 ** -15, 3.8e-5, 0.25])
    for i in range(n):
        if kind[i] < 5:
            p[i] = np.floor(p[i]) + fr[rng.integers(0, len(fr), 2)]
    p = p.astype(f32)
    wild = rng.random(n) < 0.02                    # non-finite / out-of-int-range coordinates (x86 cvFloor -> INT_MIN)
    p[wild, rng.integers(0, 2, int(wild.sum()))] = rng.choice(np.array([np.nan, np.inf, -np.inf, 3e9, -3e9, 1e20], f32), int(wild.sum()))
    return p


for case in range(n_cases):
    # ---- LK
    w, h = int(rng.integers(40, int(os.environ.get("FUZZ_MAXW", "700")))), int(rng.integers(36, int(os.environ.get("FUZZ_MAXH", "420"))))
    win = (int(rng.integers(3, 33)), int(rng.integers(3, 33)))
    if rng.random() < 0.25:                        # the compile-time window instantiations
        win = [(15, 15), (21, 21), (30, 30)][int(rng.integers(0, 3))]
    elif rng.random() < 0.08:                      # sides beyond 32: the plain kernel (lk_track_big_kernel)
        win = (int(rng.integers(3, 65)), int(rng.integers(33, 65)))
        if rng.random() < 0.5:
            win = win[::-1]
    if w <= win[0] + 2 or h <= win[1] + 2:
        continue
    ml = int(rng.integers(0, 6)); flags = int(rng.choice([0, 4, 8, 12]))
    mot = dict(tx=rng.uniform(-4, 4), ty=rng.uniform(-3, 3), rot_deg=rng.uniform(-1, 1), zoom=rng.uniform(0.98, 1.03))
    A, B, T = synth.frame_pair(w, h, motion=mot, seed=int(rng.integers(1, 1 << 20)))
    if rng.random() < 0.3:                         # harsh content: noise + blocks
        A = np.clip(A.astype(int) + rng.integers(-60, 60, A.shape), 0, 255).astype(np.uint8)
        B = np.clip(B.astype(int) + rng.integers(-60, 60, B.shape), 0, 255).astype(np.uint8)
    pts = adversarial_points(w, h, int(rng.integers(1, int(os.environ.get("FUZZ_MAXPTS", "400")))))
    with np.errstate(invalid="ignore", over="ignore"):
        guess = (synth.apply_T(T, pts) + rng.normal(0, 0.7, pts.shape)).astype(f32) if flags & 4 else None
    kw = dict(win=win, max_level=ml, max_iters=int(rng.integers(0, 40)), eps=float(rng.choice([0.0, 0.01, 0.03, 1e-3])), flags=flags)
    g = ctx.calcOpticalFlowPyrLK(A, B, pts, None if guess is None else guess.copy(), dr3.lk_params(**kw))
    o = O.lk_track(A, B, pts, None if guess is None else guess.copy(), O.lk_params(**kw), order=O.ORDER_WAVE64)
    fin = np.isfinite(o[0]) & np.isfinite(g[0])     # NaN payloads are not compared, NaN-ness is
    ok = g[3] == o[3] and np.array_equal(g[1], o[1]) and np.array_equal(g[0], o[0], equal_nan=True) and \
        np.array_equal(g[0].view(np.uint32)[fin], o[0].view(np.uint32)[fin]) and np.array_equal(g[2].view(np.uint32), o[2].view(np.uint32))
    if not ok:
        bad += 1
        d = np.where((g[0] != o[0]).any(1) | (g[1] != o[1]))[0]
        print("LK MISMATCH case", case, (w, h), kw, "n_bad", len(d), "first", pts[d[:3]].tolist(), flush=True)
    # ---- warp
    t = rng.choice(["u8c3", "u8c1", "f32c4", "u8c4", "f32c1"])
    shape = {"u8c3": (h, w, 3), "u8c1": (h, w), "f32c4": (h, w, 4), "u8c4": (h, w, 4), "f32c1": (h, w)}[t]
    src = rng.integers(0, 256, shape, dtype=np.uint8) if t.startswith("u8") else (rng.random(shape, dtype=f32) * 255).astype(f32)
    ang = rng.uniform(-0.6, 0.6); sc = rng.uniform(0.5, 2.0)
    M = np.array([[np.cos(ang) * sc, -np.sin(ang) * sc, rng.uniform(-30, 60)], [np.sin(ang) * sc, np.cos(ang) * sc, rng.uniform(-30, 60)],
                  [rng.uniform(-2e-3, 2e-3), rng.uniform(-2e-3, 2e-3), 1]], f32)
    if rng.random() < 0.2:
        M[0, 2] = np.floor(M[0, 2]) + rng.choice([0, 0.5, 1 / 64, 3 / 64]); M[2, :2] = 0; M[:2, :2] = np.eye(2)   # ties everywhere
    ds = (int(rng.integers(1, 900)), int(rng.integers(1, 500)))
    fl = int(rng.choice([0, dr3.WARP_INVERSE_MAP]))
    gw = ctx.warpPerspective(src, M, ds, fl); ow = O.warp_perspective(src, M, ds, O.WARP_INVERSE_MAP if fl else 0)
    if not np.array_equal(gw.view(np.uint8), ow.view(np.uint8)):
        bad += 1
        print("WARP MISMATCH case", case, t, (w, h), ds, fl, M.tolist(), int((gw != ow).sum()), flush=True)
    # ---- FAST + grid detect
    img = np.clip(A.astype(int) + rng.integers(-40, 40, A.shape) * (rng.random(A.shape) < 0.1), 0, 255).astype(np.uint8)
    if w >= 32 and h >= 32:
        thr = int(rng.integers(5, 60)); nl = int(rng.integers(1, 4)); cell = int(rng.integers(8, 40))
        if (w >> (nl - 1)) >= 8 and (h >> (nl - 1)) >= 8:
            og = O.fast_detect_grid(O.halfsample_pyramid(img, nl, O.HALF_REF_X86), cell, thr, 20.0)
            gg = ctx.fast_detect(img, nl, cell, thr, 20.0)
            m = og[3] >= 0
            if not (np.array_equal(gg[3], og[3]) and np.array_equal(gg[0][m], og[0][m]) and np.array_equal(gg[1][m], og[1][m]) and
                    np.array_equal(gg[2][m].view(np.uint32), og[2][m].view(np.uint32))):
                bad += 1
                print("FAST MISMATCH case", case, (w, h), thr, nl, cell, flush=True)
    # ---- remap with hostile maps, fused spherical / cylindrical, panorama paste, stitch compose
    if w >= 16 and h >= 16:
        c3 = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        mh, mw = int(rng.integers(1, 200)), int(rng.integers(1, 300))
        mx = rng.uniform(-8, w + 8, (mh, mw)).astype(f32); my = rng.uniform(-8, h + 8, (mh, mw)).astype(f32)
        sp = rng.random((mh, mw))
        mx[sp < 0.02] = rng.choice(np.array([0, 0.5, -0.5, w - 1, w - 0.5, 1e9, -1e9, 40000.0, -40000.0, np.inf, -np.inf, np.nan], f32), int((sp < 0.02).sum()))
        my[(sp > 0.98)] = rng.choice(np.array([0, 0.5, h - 1, h - 0.5, 1e9, -1e9, np.inf], f32), int((sp > 0.98).sum()))
        src_r = c3 if rng.random() < 0.6 else c3[..., 0].copy()
        if not np.array_equal(ctx.remap(src_r, mx, my), O.remap(src_r, mx, my)):
            bad += 1; print("REMAP MISMATCH case", case, (w, h), (mw, mh), flush=True)
        fl_ = float(rng.uniform(0.3, 3.0) * w)
        for kind, fg, fo in (("sph", ctx.warp_spherical, O.spherical_map), ("cyl", ctx.warp_cylindrical, O.cylindrical_map)):
            u_, v_ = fo(w, h, fl_)
            if not np.array_equal(fg(c3, fl_), O.remap(c3, u_, v_)):
                bad += 1; print("MAPGEN MISMATCH case", case, kind, (w, h), fl_, flush=True)
        Mp = M.copy(); Mp[:2, :2] /= max(sc, 1e-3); Mp[0, 2] = rng.uniform(-0.5 * w, 1.5 * w); Mp[1, 2] = rng.uniform(-20, 60); Mp[2, :2] *= 0.05
        cw, ch = int(rng.integers(w, 3 * w)), int(rng.integers(h, 2 * h))
        box = O.panorama_bbox(Mp, w, h)
        if box[2] - box[0] >= 2:
            fw = int(rng.integers(0, 80))
            want = np.zeros((ch, cw, 4), f32)
            want[...] = rng.random((ch, cw, 4), dtype=f32) * (rng.random() < 0.5)
            start = want.copy()
            O.panorama_add(c3, want, Mp, fw)
            # image rows at a 16-byte aligned pitch in two cases of three (the strip form's condition), packed otherwise (the direct form)
            pitch = (3 * w + 15) // 16 * 16 if case % 3 else 3 * w
            host = np.zeros((h, pitch), np.uint8); host[:, :3 * w] = c3.reshape(h, 3 * w)
            d_c = ctx.to_device(start); d_i = ctx.to_device(host); d_o = ctx.alloc(cw * ch * 3)
            ctx.panorama_add(d_i, w, h, pitch, d_c, cw, ch, cw * 16, Mp, fw)
            ctx.panorama_normalize(d_c, cw, ch, cw * 16, d_o, cw * 3)
            gc_ = d_c.download((ch, cw, 4), f32); gp_ = d_o.download((ch, cw, 3), np.uint8)
            d_c.free(); d_i.free(); d_o.free()
            if not (np.array_equal(gc_.view(np.uint32), want.view(np.uint32)) and np.array_equal(gp_, O.normalize_canvas(want))):
                bad += 1; print("PASTE MISMATCH case", case, (w, h), (cw, ch), fw, Mp.tolist(), flush=True)
        Hs = Mp.copy(); Hs[0, 2] = -rng.uniform(0.1, 0.7) * w
        try:
            ow_ = O.stitch_compose(c3, np.roll(c3, 5, 1), Hs)
            if ow_.shape[0] * ow_.shape[1] < 4e6 and not np.array_equal(ctx.stitch_compose(c3, np.roll(c3, 5, 1), Hs), ow_):
                bad += 1; print("STITCH MISMATCH case", case, (w, h), Hs.tolist(), flush=True)
        except ValueError:
            pass
    # ---- half-sample pyramid in every mode on odd / even sizes, and the device-resident batched detector on pitched slots
    if w >= 48 and h >= 48:
        nl2 = int(rng.integers(2, 4))
        for mode in ("HALF_TRUNC", "HALF_SSE2", "HALF_REF_SCALAR", "HALF_REF_X86"):
            gp = ctx.create_img_pyramid(img, nl2, getattr(dr3, mode)); op = O.halfsample_pyramid(img, nl2, getattr(O, mode))
            if not all(np.array_equal(a_, b_) for a_, b_ in zip(gp, op)):
                bad += 1; print("HALFSAMPLE MISMATCH case", case, (w, h), mode, flush=True)
        nf = 2
        fbatch = dr3.FrameBatch(ctx, w, h, win=(5, 5), max_level=1, capacity=nf)
        fr2 = [img, np.ascontiguousarray(img[::-1])]
        fbatch.upload(0, np.stack(fr2)); fbatch.build(0, nf, True)
        cell2 = int(rng.integers(8, 40)); thr2 = int(rng.integers(8, 50))
        gc2, gr2 = -(-w // cell2), -(-h // cell2)
        d_c2 = ctx.alloc(nf * gc2 * gr2 * 16)
        fbatch.detect(0, nf, nl2, cell2, thr2, 20.0, d_corners=d_c2); ctx.sync()
        cor = d_c2.download((nf, gc2 * gr2), np.dtype([("x", np.int32), ("y", np.int32), ("level", np.int32), ("score", np.float32)]))
        for i_ in range(nf):
            og2 = O.fast_detect_grid(O.halfsample_pyramid(fr2[i_], nl2, O.HALF_REF_X86), cell2, thr2, 20.0)
            m2 = og2[3] >= 0
            if not (np.array_equal(cor["level"][i_], og2[3]) and np.array_equal(cor["x"][i_][m2], og2[0][m2]) and
                    np.array_equal(cor["y"][i_][m2], og2[1][m2]) and np.array_equal(cor["score"][i_][m2].view(np.uint32), og2[2][m2].view(np.uint32))):
                bad += 1; print("BATCH DETECT MISMATCH case", case, (w, h), nl2, cell2, thr2, i_, flush=True)
        d_c2.free(); fbatch.close()
    if case % 10 == 9:
        print("case", case + 1, "mismatches so far", bad, flush=True)
print("done:", n_cases, "cases,", bad, "mismatches")
sys.exit(1 if bad else 0)
