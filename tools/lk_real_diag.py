"""the tracker on the reference's real KITTI pair at the headline launch shape: what a wave executes.  Per (point, level) iteration counts
and re-staging counts (DR3HIP_LK_DIAG_RESTAGES) -> per wave: sum over levels of the slowest group's iterations (the lock-step trips),
re-stagings per wave, kernel ms.  usage: python tools/lk_real_diag.py"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dr3 = importlib.import_module("3dr_amd")
ctx = dr3.Context(0)
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "kitti_pair.npz"))
A, Bf = np.ascontiguousarray(z["kitti0"]), np.ascontiguousarray(z["kitti5"])
H_, W_ = A.shape
NP = 256
def corners(img):
    x, y, sc, lv = ctx.fast_detect(img, 3, 15, 20, 20.0)
    m = lv >= 0
    return np.stack([x[m], y[m]], 1).astype(np.float32)
pa, pb = corners(A), corners(Bf)
n = max(len(pa), len(pb))
pad = lambda q: np.concatenate([q, np.full((n - len(q), 2), dr3.NO_POINT, np.float32)])
P = np.ascontiguousarray(np.stack([pad(pa) if i % 2 == 0 else pad(pb) for i in range(NP)]), np.float32)
kw = dict(win=(15, 15), max_level=3, max_iters=30, eps=0.01)
b = dr3.FrameBatch(ctx, W_, H_, win=kw["win"], max_level=kw["max_level"], capacity=NP + 1)
for s0 in range(0, NP + 1, 64):
    b.upload(s0, np.stack([A if i % 2 == 0 else Bf for i in range(s0, min(s0 + 64, NP + 1))]))
b.build(0, NP + 1, True)
lv = b.levels
d_prev, d_next = ctx.to_device(P), ctx.to_device(P)
d_st, d_err, d_it = ctx.alloc(NP * n), ctx.alloc(NP * n * 4), ctx.alloc(NP * n * lv * 4)
for flags in (0, dr3.LK_DIAG_RESTAGES):
    params = dr3.lk_params(flags=flags, **kw)
    ctx.profile_enable(True)
    for _ in range(3):
        b.track(NP, 0, d_prev, d_next, n, d_st, d_err, params)
    ctx.sync(); ctx.profile_reset()
    for _ in range(10):
        b.track(NP, 0, d_prev, d_next, n, d_st, d_err, params)
    ctx.sync()
    ms, cnt = ctx.profile_read(dr3.PROF_LK)
    print("flags %x: %.3f ms per launch" % (flags, ms / cnt))
b.track(NP, 0, d_prev, d_next, n, d_st, d_err, dr3.lk_params(flags=dr3.LK_DIAG_RESTAGES, **kw), d_iters=d_it); ctx.sync()
raw = d_it.download((NP, n, lv), np.int32)
it, rs = raw & 0xffff, raw >> 16
vm = P[:, :, 0] > -1e5
ng = n // 8 * 8
g_it = it[:, :ng].reshape(NP, -1, 8, lv); g_rs = rs[:, :ng].reshape(NP, -1, 8, lv)
print("points", int(vm.sum()), "mean iteration-levels per point %.1f" % it.sum(2)[vm].mean())
print("per wave: max-of-sums %.1f, sum over levels of the group max (trips) %.1f" % (g_it.sum(3).max(2).mean(), g_it.max(2).sum(2).mean()))
print("re-stagings per point %.3f, per wave %.2f; waves with >= 1: %.2f; per level (fine -> coarse) %s"
      % (rs.sum(2)[vm].mean(), g_rs.sum((2, 3)).mean(), (g_rs.sum((2, 3)) > 0).mean(), np.round(g_rs.sum(2).mean((0, 1)), 3).tolist()))
print("iterations per level (fine -> coarse): mean per point %s, group max %s" % (np.round(it[vm].mean(0), 2).tolist(), np.round(g_it.max(2).mean((0, 1)), 2).tolist()))
ctx.close()
