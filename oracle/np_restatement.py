"""Independent numpy restatement of the same algorithms, written a different way (physical padding,
vectorised arithmetic) so that it can cross-check oracle/dr3_oracle.c.  TEST INFRASTRUCTURE ONLY.

It follows SURVEY.md Appendix A (OpenCV 3.4 pyrDown / calcSharrDeriv / LKTrackerInvoker /
WarpPerspectiveInvoker / remapBilinear) the way OpenCV itself lays the data out: levels are
*physically* padded by winSize (REFLECT_101 for images, constant 0 for derivatives), whereas the C
oracle and the HIP kernels reflect indices.  Agreement of the two is the check.
"""
import numpy as np


def pyr_down(img):
    img = np.asarray(img, np.uint8)
    h, w = img.shape
    dh, dw = (h + 1) // 2, (w + 1) // 2
    # pad enough for 2*x+2 with x = dw-1, reflect (numpy 'reflect' == REFLECT_101)
    p = np.pad(img.astype(np.int32), ((2, 3), (2, 3)), mode="reflect")
    k = (1, 4, 6, 4, 1)
    hs = sum(k[i] * p[:, i:i + 2 * dw:2] for i in range(5))            # rows: all, cols: dw
    vs = sum(k[i] * hs[i:i + 2 * dh:2, :] for i in range(5))
    return ((vs + 128) >> 8).astype(np.uint8)


def scharr(img):
    img = np.asarray(img, np.uint8)
    p = np.pad(img.astype(np.int32), 1, mode="reflect")
    h, w = img.shape
    c = lambda dy, dx: p[1 + dy:1 + dy + h, 1 + dx:1 + dx + w]
    ix = 3 * (c(-1, 1) - c(-1, -1)) + 10 * (c(0, 1) - c(0, -1)) + 3 * (c(1, 1) - c(1, -1))
    iy = 3 * (c(1, -1) - c(-1, -1)) + 10 * (c(1, 0) - c(-1, 0)) + 3 * (c(1, 1) - c(-1, 1))
    return np.stack([ix, iy], -1).astype(np.int16)


def effective_max_level(w, h, win, max_level):
    for level in range(max_level + 1):
        w, h = (w + 1) // 2, (h + 1) // 2
        if w <= win[0] or h <= win[1]:
            return level
    return max_level


def _seq_sum_f32(v):
    """sequential float32 accumulation (np.add.accumulate is a plain left fold)"""
    v = np.asarray(v, np.float32).ravel()
    if v.size == 0:
        return np.float32(0)
    return np.add.accumulate(v, dtype=np.float32)[-1]


def _weights(a, b):
    a, b = np.float32(a), np.float32(b)
    one = np.float32(1)
    s = np.float32(1 << 14)
    iw00 = int(np.rint((one - a) * (one - b) * s))
    iw01 = int(np.rint(a * (one - b) * s))
    iw10 = int(np.rint((one - a) * b * s))
    return iw00, iw01, iw10, (1 << 14) - iw00 - iw01 - iw10


def _patch(P, x, y, ww, wh, iw, shift):
    """bilinear fixed-point window starting at integer (x,y) of padded int32 plane P"""
    iw00, iw01, iw10, iw11 = iw
    a = P[y:y + wh, x:x + ww]; b = P[y:y + wh, x + 1:x + ww + 1]
    c = P[y + 1:y + wh + 1, x:x + ww]; d = P[y + 1:y + wh + 1, x + 1:x + ww + 1]
    v = a * iw00 + b * iw01 + c * iw10 + d * iw11
    return (v + (1 << (shift - 1))) >> shift


def lk_track(prev, nxt, prev_xy, next_xy0=None, win=(21, 21), max_level=3, max_iters=30,
             eps=0.01, flags=0, min_eig_threshold=1e-4, crit_type=3):
    """Pure numpy calcOpticalFlowPyrLK (raster summation order).  Slow: small cases only."""
    f32 = np.float32
    ww, wh = win
    prev = np.asarray(prev, np.uint8); nxt = np.asarray(nxt, np.uint8)
    h, w = prev.shape
    L = effective_max_level(w, h, win, max_level)
    max_count = min(max(max_iters, 0), 100) if (crit_type & 1) else 30
    epsilon = min(max(eps, 0.0), 10.0) if (crit_type & 2) else 0.01
    epsilon *= epsilon
    Is, Js = [prev], [nxt]
    for _ in range(L):
        Is.append(pyr_down(Is[-1])); Js.append(pyr_down(Js[-1]))
    # physically padded planes, as OpenCV builds them
    Ip = [np.pad(i.astype(np.int32), ((wh, wh), (ww, ww)), mode="reflect") for i in Is]
    Jp = [np.pad(j.astype(np.int32), ((wh, wh), (ww, ww)), mode="reflect") for j in Js]
    Dp = [np.pad(scharr(i).astype(np.int32), ((wh, wh), (ww, ww), (0, 0)), mode="constant") for i in Is]

    prev_xy = np.asarray(prev_xy, f32).reshape(-1, 2)
    n = len(prev_xy)
    out = prev_xy.copy() if next_xy0 is None else np.asarray(next_xy0, f32).reshape(-1, 2).copy()
    status = np.ones(n, np.uint8); err = np.zeros(n, f32)
    half = np.array([(ww - 1) * f32(0.5), (wh - 1) * f32(0.5)], f32)
    FLT_SCALE = f32(1.0 / (1 << 20))

    for i in range(n):
        npt = out[i].copy()
        for level in range(L, -1, -1):
            lh, lw = Is[level].shape
            scale = f32(1.0 / (1 << level))
            ppt = prev_xy[i] * scale
            if level == L:
                cpt = npt * scale if (flags & 4) else ppt.copy()
            else:
                cpt = npt * f32(2)
            npt = cpt.copy()
            ppt = ppt - half
            ipx, ipy = int(np.floor(ppt[0])), int(np.floor(ppt[1]))
            if ipx < -ww or ipx >= lw or ipy < -wh or ipy >= lh:
                if level == 0:
                    status[i] = 0; err[i] = 0
                continue
            iw = _weights(ppt[0] - f32(ipx), ppt[1] - f32(ipy))
            Iw = _patch(Ip[level], ipx + ww, ipy + wh, ww, wh, iw, 14 - 5)
            Ix = _patch(Dp[level][..., 0], ipx + ww, ipy + wh, ww, wh, iw, 14)
            Iy = _patch(Dp[level][..., 1], ipx + ww, ipy + wh, ww, wh, iw, 14)
            A11 = _seq_sum_f32((Ix * Ix).astype(f32)) * FLT_SCALE
            A12 = _seq_sum_f32((Ix * Iy).astype(f32)) * FLT_SCALE
            A22 = _seq_sum_f32((Iy * Iy).astype(f32)) * FLT_SCALE
            D = A11 * A22 - A12 * A12
            minEig = (A22 + A11 - np.sqrt((A11 - A22) * (A11 - A22) + f32(4) * A12 * A12)) / f32(2 * ww * wh)
            if flags & 8:
                err[i] = minEig
            if float(minEig) < min_eig_threshold or D < np.finfo(f32).eps:
                if level == 0:
                    status[i] = 0
                continue
            D = f32(1) / D
            cpt = cpt - half
            pd = np.zeros(2, f32)
            for j in range(max_count):
                inx, iny = int(np.floor(cpt[0])), int(np.floor(cpt[1]))
                if inx < -ww or inx >= lw or iny < -wh or iny >= lh:
                    if level == 0:
                        status[i] = 0
                    break
                iwj = _weights(cpt[0] - f32(inx), cpt[1] - f32(iny))
                diff = _patch(Jp[level], inx + ww, iny + wh, ww, wh, iwj, 14 - 5) - Iw
                b1 = _seq_sum_f32((diff * Ix).astype(f32)) * FLT_SCALE
                b2 = _seq_sum_f32((diff * Iy).astype(f32)) * FLT_SCALE
                delta = np.array([(A12 * b2 - A22 * b1) * D, (A12 * b1 - A11 * b2) * D], f32)
                cpt = cpt + delta
                npt = cpt + half
                if float(delta[0]) * float(delta[0]) + float(delta[1]) * float(delta[1]) <= epsilon:
                    break
                if j > 0 and abs(float(delta[0] + pd[0])) < 0.01 and abs(float(delta[1] + pd[1])) < 0.01:
                    npt = npt - delta * f32(0.5)
                    break
                pd = delta
            if status[i] and level == 0 and not (flags & 8):
                fpt = npt - half
                inx, iny = int(np.floor(fpt[0])), int(np.floor(fpt[1]))
                if inx < -ww or inx >= lw or iny < -wh or iny >= lh:
                    status[i] = 0
                    continue
                iwj = _weights(fpt[0] - f32(inx), fpt[1] - f32(iny))
                diff = _patch(Jp[level], inx + ww, iny + wh, ww, wh, iwj, 14 - 5) - Iw
                e = _seq_sum_f32(np.abs(diff.astype(f32)))
                err[i] = e * f32(1) / f32(32 * ww * wh)
        out[i] = npt
    return out, status, err, L


def invert3x3(S):
    S = np.asarray(S, np.float64).reshape(3, 3)
    det = (S[0, 0] * (S[1, 1] * S[2, 2] - S[1, 2] * S[2, 1]) - S[0, 1] * (S[1, 0] * S[2, 2] - S[1, 2] * S[2, 0])
           + S[0, 2] * (S[1, 0] * S[2, 1] - S[1, 1] * S[2, 0]))
    d = 1.0 / det
    t = np.empty((3, 3))
    t[0, 0] = (S[1, 1] * S[2, 2] - S[1, 2] * S[2, 1]) * d
    t[0, 1] = (S[0, 2] * S[2, 1] - S[0, 1] * S[2, 2]) * d
    t[0, 2] = (S[0, 1] * S[1, 2] - S[0, 2] * S[1, 1]) * d
    t[1, 0] = (S[1, 2] * S[2, 0] - S[1, 0] * S[2, 2]) * d
    t[1, 1] = (S[0, 0] * S[2, 2] - S[0, 2] * S[2, 0]) * d
    t[1, 2] = (S[0, 2] * S[1, 0] - S[0, 0] * S[1, 2]) * d
    t[2, 0] = (S[1, 0] * S[2, 1] - S[1, 1] * S[2, 0]) * d
    t[2, 1] = (S[0, 1] * S[2, 0] - S[0, 0] * S[2, 1]) * d
    t[2, 2] = (S[0, 0] * S[1, 1] - S[0, 1] * S[1, 0]) * d
    return t


def _sample(src, sx, sy, fx, fy):
    """remapBilinear on integer/fraction planes; src HxW[xC], u8 or f32; zero border"""
    src = np.asarray(src)
    if src.ndim == 2:
        src = src[..., None]
    h, w, _ = src.shape
    isf = src.dtype == np.float32
    P = np.zeros((h + 2, w + 2, src.shape[2]), np.float32 if isf else np.int64)
    P[1:-1, 1:-1] = src
    sxc = np.clip(sx, -1, w) + 1
    syc = np.clip(sy, -1, h) + 1
    # taps beyond one pixel outside are all-zero; clip keeps them on the zero frame or, for sx==w /
    # sy==h, on the frame itself whose right/bottom neighbour must also read zero:
    sx1 = np.clip(sxc + 1, 0, w + 1); sy1 = np.clip(syc + 1, 0, h + 1)
    far = (sx < -1) | (sx > w) | (sy < -1) | (sy > h)
    v00 = P[syc, sxc]; v01 = P[syc, sx1]; v10 = P[sy1, sxc]; v11 = P[sy1, sx1]
    if isf:
        f32 = np.float32
        x1 = fx.astype(f32) * f32(1 / 32); x0 = f32(1) - x1
        y1 = fy.astype(f32) * f32(1 / 32); y0 = f32(1) - y1
        w0 = (y0 * x0)[..., None]; w1 = (y0 * x1)[..., None]; w2 = (y1 * x0)[..., None]; w3 = (y1 * x1)[..., None]
        out = ((v00 * w0 + v01 * w1) + v10 * w2) + v11 * w3
        out[far] = 0
        return out.astype(f32)
    w0 = ((32 - fx) * (32 - fy) * 32)[..., None]; w1 = (fx * (32 - fy) * 32)[..., None]
    w2 = ((32 - fx) * fy * 32)[..., None]; w3 = (fx * fy * 32)[..., None]
    out = (v00 * w0 + v01 * w1 + v10 * w2 + v11 * w3 + (1 << 14)) >> 15
    out[far] = 0
    return np.clip(out, 0, 255).astype(np.uint8)


def warp_perspective(src, M, dsize, inverse_map=False):
    src = np.asarray(src)
    dw, dh = dsize
    Md = np.asarray(M, np.float32).astype(np.float64).reshape(3, 3)
    if not inverse_map:
        Md = invert3x3(Md)
    m = Md.reshape(9)
    bh0 = min(16, dh); bw0 = min(1024 // bh0, dw)
    xs = np.arange(dw); x0 = (xs // bw0) * bw0; x1 = xs - x0
    ys = np.arange(dh)[:, None].astype(np.float64)
    x0 = x0[None, :].astype(np.float64); x1 = x1[None, :].astype(np.float64)
    X0 = m[0] * x0 + m[1] * ys + m[2]
    Y0 = m[3] * x0 + m[4] * ys + m[5]
    W0 = m[6] * x0 + m[7] * ys + m[8]
    W = W0 + m[6] * x1
    with np.errstate(divide="ignore", invalid="ignore"):
        W = np.where(W != 0, 32.0 / W, 0.0)
    fX = np.clip((X0 + m[0] * x1) * W, -2147483648.0, 2147483647.0)
    fY = np.clip((Y0 + m[3] * x1) * W, -2147483648.0, 2147483647.0)
    X = np.rint(fX).astype(np.int64); Y = np.rint(fY).astype(np.int64)
    sx = np.clip(X >> 5, -32768, 32767); sy = np.clip(Y >> 5, -32768, 32767)
    out = _sample(src, sx, sy, X & 31, Y & 31)
    return out if src.ndim == 3 else out[..., 0]


def remap(src, mapx, mapy):
    src = np.asarray(src)
    f32 = np.float32
    sx = np.rint(np.asarray(mapx, f32) * f32(32)).astype(np.int64)
    sy = np.rint(np.asarray(mapy, f32) * f32(32)).astype(np.int64)
    out = _sample(src, np.clip(sx >> 5, -32768, 32767), np.clip(sy >> 5, -32768, 32767), sx & 31, sy & 31)
    return out if src.ndim == 3 else out[..., 0]


def halfsample_trunc(img):
    img = np.asarray(img, np.uint8).astype(np.int32)
    h, w = img.shape
    oh, ow = h // 2, w // 2
    s = img[0:2 * oh:2, 0:2 * ow:2] + img[0:2 * oh:2, 1:2 * ow:2] + img[1:2 * oh:2, 0:2 * ow:2] + img[1:2 * oh:2, 1:2 * ow:2]
    return (s // 4).astype(np.uint8)


def halfsample_sse2(img):
    img = np.asarray(img, np.uint8).astype(np.int32)
    h, w = img.shape
    oh, ow = h // 2, w // 2
    v = (img[0:2 * oh:2, :] + img[1:2 * oh:2, :] + 1) >> 1
    return ((v[:, 0:2 * ow:2] + v[:, 1:2 * ow:2] + 1) >> 1).astype(np.uint8)


def halfsample_ref_scalar(img):
    """the pointer walk of src/utils.cpp:402-418 on a contiguous image, literally"""
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    flat = img.ravel().astype(np.int32)
    oh, ow = h // 2, w // 2
    out = np.zeros((oh, ow), np.uint8)
    top, bottom, end = 0, w, w * h
    p = 0
    while bottom < end and p < oh * ow:
        for _ in range(ow):
            out.flat[p] = (flat[top] + flat[top + 1] + flat[bottom] + flat[bottom + 1]) // 4
            p += 1; top += 2; bottom += 2
        top += w; bottom += w
    return out
