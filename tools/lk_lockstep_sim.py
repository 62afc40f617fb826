"""What the tracker's lock step costs on the reference's real KITTI pair, and what each way around it would cost (CPU only: the
per-(point, level) iteration counts are the kernel-order oracle's, equal to the kernel's; instruction counts are the shipped
lk_trackg_kernel<15, 8, 2>'s: 250 VALU per trip of a wave of 8 points, 700 per level of setup).  The table of profiles/r04_lk/README.md.
usage: python tools/lk_lockstep_sim.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O        # noqa: E402

IT, SETUP = 250, 700


def iteration_counts():
    z = np.load(os.path.join(ROOT, "tests", "golden", "kitti_pair.npz"))
    A, B = np.ascontiguousarray(z["kitti0"]), np.ascontiguousarray(z["kitti5"])
    x, y, s, l = O.fast_detect_grid(O.halfsample_pyramid(A, 3, O.HALF_REF_X86), 15, 20, 20.0)
    pts = np.stack([x[l >= 0], y[l >= 0]], 1).astype(np.float32)
    p = O.lk_params(win=(15, 15), max_level=3, max_iters=30, eps=0.01)
    return O.lk_track(A, B, pts, None, p, order=O.ORDER_WAVE64, want_iters=True, opt=True)[4].astype(int)


def desync(G, thr):
    """per-group level state in one uniform loop: a trip runs the setup code if >= thr groups wait for it (or nobody iterates) and the
    iteration code if any group iterates"""
    tot = 0
    for g in G:
        lvl, rem, state = np.zeros(8, int), np.zeros(8, int), np.zeros(8, int)       # state 0 waits for setup, 1 iterates, 2 done
        while (state != 2).any():
            parked, itr = state == 0, state == 1
            if parked.sum() >= thr or (parked.any() and not itr.any()):
                tot += SETUP
                for k in np.where(parked)[0]:
                    rem[k] = g[k, lvl[k]]; state[k] = 1
                    while state[k] == 1 and rem[k] == 0:
                        lvl[k] += 1
                        if lvl[k] == 4: state[k] = 2
                        else: rem[k] = g[k, lvl[k]]; tot += 0
                continue
            tot += IT
            for k in np.where(itr)[0]:
                rem[k] -= 1
                if rem[k] == 0:
                    lvl[k] += 1
                    state[k] = 2 if lvl[k] == 4 else 0
    return tot / len(G)


def chunks(it, ch):
    """per level: passes of at most ch[i] iterations, unfinished points compacted in between, setup re-done on resume"""
    n, tot = len(it), 0.0
    for l in range(4):
        rem, alive = it[:, l].copy(), np.arange(n)
        for K in ch:
            if not len(alive): break
            r = rem[alive]; m = len(alive) // 8 * 8
            trips = np.minimum(r[:m].reshape(-1, 8), K).max(1).sum() + (min(r[m:].max(), K) if m < len(alive) else 0)
            tot += trips * IT + (len(alive) + 7) // 8 * SETUP
            rem[alive] = np.maximum(r - K, 0); alive = alive[rem[alive] > 0]
    return tot / (n / 8)


def wg_compaction(it, every, X=80):
    """4 waves of 8 points re-pack their active points through LDS every `every` trips (X instructions per wave and exchange)"""
    n = len(it) // 32 * 32
    tot = 0.0
    for i in range(0, n, 32):
        W = it[i:i + 32]
        for l in range(4):
            rem = W[:, l].copy(); assign = [list(range(8 * k, 8 * k + 8)) for k in range(4)]; t = 0
            while (rem > 0).any():
                tot += IT * sum(1 for a in assign if any(rem[j] > 0 for j in a))
                rem = np.maximum(rem - 1, 0); t += 1
                if every and t % every == 0:
                    act = [j for j in range(32) if rem[j] > 0]
                    now, need = sum(1 for a in assign if any(rem[j] > 0 for j in a)), (len(act) + 7) // 8
                    if need < now:
                        assign = [act[8 * k:8 * k + 8] for k in range(need)]; tot += X * now
    return tot / (n / 8)


def main():
    it = iteration_counts()[:, ::-1]                      # coarse -> fine, the order of execution
    n = len(it) // 8 * 8
    G = it[:n].reshape(-1, 8, 4)
    now = G.max(1).sum(1).mean()
    print("points %d: mean iteration-levels %.1f; a wave executes %.1f (max of sums %.1f)" % (len(it), it.sum(1).mean(), now, G.sum(2).max(1).mean()))
    base = now * IT + 4 * SETUP
    print("now                                   %6.0f  1.00" % base)
    print("no lock step                          %6.0f  %.2f" % (it.sum(1).mean() * IT + 4 * SETUP, (it.sum(1).mean() * IT + 4 * SETUP) / base))
    for thr in (1, 4, 6):
        c = desync(G, thr); print("per-group level state, setup at >= %d   %6.0f  %.2f" % (thr, c, c / base))
    t = it[:n, 0].reshape(-1, 8).max(1).mean()
    for l in (1, 2, 3):
        o = np.argsort(it[:n, l - 1], kind="stable"); t += it[:n, l][o].reshape(-1, 8).max(1).mean()
    print("bin each level by the previous count  %6.0f  %.2f" % (t * IT + 4 * SETUP, (t * IT + 4 * SETUP) / base))
    for ch in ([8, 22], [6, 10, 14], [4, 8, 18]):
        c = chunks(it, ch); print("chunks %-30s %6.0f  %.2f" % (ch, c, c / base))
    for ev in (1, 4, 8):
        c = wg_compaction(it, ev); print("4-wave workgroup, re-pack every %d      %6.0f  %.2f (iteration part; now %.0f)" % (ev, c, c / wg_compaction(it, 0), wg_compaction(it, 0)))
    hist = np.zeros(9)
    for g in G:
        for l in range(4):
            s = np.sort(g[:, l])[::-1]
            for tt in range(1, s[0] + 1): hist[(s >= tt).sum()] += 1
    print("trips by number of active groups 1..8:", (hist[1:] / hist.sum()).round(3).tolist())
    rng = np.random.default_rng(0)
    print("random grouping: %.1f trips (raster order of the detector's cells: %.1f)" % (np.mean([it[rng.permutation(len(it))[:n]].reshape(-1, 8, 4).max(1).sum(1).mean() for _ in range(10)]), now))
    for ppw in (1, 2, 4, 8, 16):
        m = len(it) // ppw * ppw
        print("points per wave %2d: %.1f trips" % (ppw, it[:m].reshape(-1, ppw, 4).max(1).sum(1).mean()))


if __name__ == "__main__":
    main()
