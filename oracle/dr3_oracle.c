/*
 * dr3_oracle.c -- CPU restatement of the 3DR dense hot path.  TEST INFRASTRUCTURE ONLY
 * (see dr3_oracle.h for the scope statement and the "PARITY UNPINNED" notice).
 *
 * Build: gcc -O2 -std=c99 -fopenmp -ffp-contract=off -fno-fast-math -shared -fPIC
 * -ffp-contract=off matters: OpenCV's baseline x86-64 build (SSE3) issues no FMA, and the
 * HIP kernels are compiled the same way so float results can be compared bit for bit.
 */
#include "dr3_oracle.h"

#include <limits.h>
#include <stddef.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------ */
/* helpers                                                                                    */
/* ------------------------------------------------------------------------------------------ */

/* cv::borderInterpolate(p, len, BORDER_REFLECT_101) */
static inline int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else       p = 2 * len - 2 - p;
    }
    return p;
}

/* cvRound on x86-64 is cvtss2si / cvtsd2si with a 32-bit destination: round half to even, and the "integer indefinite"
 * 0x80000000 for NaN, infinities and anything outside the int range (a 64-bit lrint would wrap instead). */
static inline int cv_round_f(float v)  { return (v >= -2147483648.f && v < 2147483648.f) ? (int)lrintf(v) : (int)0x80000000; }
static inline int cv_round_d(double v) { return (v >= -2147483648.0 && v < 2147483647.5) ? (int)lrint(v) : (int)0x80000000; }
/* cvFloor(float) = (int)v adjusted downwards; (int)v is cvttss2si: INT_MIN for NaN and for anything outside the int range.
 * (For v < -2^31 the adjustment wraps INT_MIN to INT_MAX in the reference's build; either value fails every bounds test
 * that consumes it, which is all LK does with it, so the indefinite value is returned as is.) */
static inline int cv_floor_f(float v)  { return (v >= -2147483648.f && v < 2147483648.f) ? (int)floorf(v) : (int)0x80000000; }

static inline short sat_s16(int v)
{
    return (short)(v < SHRT_MIN ? SHRT_MIN : v > SHRT_MAX ? SHRT_MAX : v);
}

#define DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))

static void set_threads(int n_threads)
{
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#else
    (void)n_threads;
#endif
}

/* ------------------------------------------------------------------------------------------ */
/* A-1  pyramid                                                                               */
/* ------------------------------------------------------------------------------------------ */

/* cv::buildOpticalFlowPyramid's early stop: after producing level l the
 * next size is ((w+1)/2,(h+1)/2); if that is <= winSize in either dimension, level l is the last. */
int dr3o_lk_effective_max_level(int w, int h, int win_w, int win_h, int max_level)
{
    int level;
    for (level = 0; level <= max_level; ++level) {
        w = (w + 1) / 2;
        h = (h + 1) / 2;
        if (w <= win_w || h <= win_h) return level;
    }
    return max_level;
}

/* cv::pyrDown for CV_8U (pyramids.cpp, PyrDownInvoker<FixPtCast<uchar,8>>): separable
 * [1 4 6 4 1], integer row sums, (v+128)>>8, REFLECT_101 on the source level itself. */
void dr3o_pyr_down_u8(const uint8_t* src, int sw, int sh, int sstride,
                      uint8_t* dst, int dstride)
{
    const int dw = (sw + 1) / 2, dh = (sh + 1) / 2;
    int y;
#pragma omp parallel for schedule(static)
    for (y = 0; y < dh; ++y) {
        int x, k;
        for (x = 0; x < dw; ++x) {
            int rows[5];
            for (k = 0; k < 5; ++k) {
                const uint8_t* s = src + (size_t)reflect101(2 * y + k - 2, sh) * sstride;
                const int x0 = reflect101(2 * x - 2, sw), x1 = reflect101(2 * x - 1, sw);
                const int x2 = reflect101(2 * x, sw),     x3 = reflect101(2 * x + 1, sw);
                const int x4 = reflect101(2 * x + 2, sw);
                rows[k] = s[x2] * 6 + (s[x1] + s[x3]) * 4 + s[x0] + s[x4];
            }
            {
                const int v = rows[2] * 6 + (rows[1] + rows[3]) * 4 + rows[0] + rows[4];
                dst[(size_t)y * dstride + x] = (uint8_t)((v + 128) >> 8);
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* A-2  Scharr derivatives                                                                    */
/* ------------------------------------------------------------------------------------------ */

/* calcSharrDeriv (lkpyramid.cpp): vertical [3 10 3] / [-1 0 1] into two temp rows, replicate the
 * REFLECT_101 column, then horizontal [-1 0 1] / [3 10 3].  Un-normalised (x32). */
void dr3o_scharr_s16(const uint8_t* src, int w, int h, int stride,
                     int16_t* dst, int dstride)
{
    int y;
#pragma omp parallel for schedule(static)
    for (y = 0; y < h; ++y) {
        const uint8_t* r0 = src + (size_t)(y > 0 ? y - 1 : (h > 1 ? 1 : 0)) * stride;
        const uint8_t* r1 = src + (size_t)y * stride;
        const uint8_t* r2 = src + (size_t)(y < h - 1 ? y + 1 : (h > 1 ? h - 2 : 0)) * stride;
        int16_t* d = dst + (size_t)y * dstride;
        int x;
        for (x = 0; x < w; ++x) {
            const int xm = x > 0 ? x - 1 : (w > 1 ? 1 : 0);
            const int xp = x < w - 1 ? x + 1 : (w > 1 ? w - 2 : 0);
            /* trow0 = smoothed-in-y, trow1 = differentiated-in-y */
            const int t0m = (r0[xm] + r2[xm]) * 3 + r1[xm] * 10;
            const int t0p = (r0[xp] + r2[xp]) * 3 + r1[xp] * 10;
            const int t1m = r2[xm] - r0[xm];
            const int t1c = r2[x] - r0[x];
            const int t1p = r2[xp] - r0[xp];
            d[2 * x]     = (int16_t)(t0p - t0m);
            d[2 * x + 1] = (int16_t)((t1p + t1m) * 3 + t1c * 10);
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* A-3  Lucas-Kanade tracker                                                                  */
/* ------------------------------------------------------------------------------------------ */

typedef struct {
    const uint8_t* img; int w, h, stride;         /* un-padded level */
} lvl_img;
typedef struct {
    const int16_t* d; int w, h, stride;           /* interleaved (Ix,Iy), stride in int16 */
} lvl_deriv;

/* value of the REFLECT_101-padded level at virtual (x,y); OpenCV pads by winSize, and the
 * bounds tests keep every read inside that pad, so index reflection is equivalent. */
static inline int img_at(const lvl_img* L, int x, int y)
{
    return L->img[(size_t)reflect101(y, L->h) * L->stride + reflect101(x, L->w)];
}
/* derivative image is padded with constant 0 */
static inline int der_at(const lvl_deriv* D, int x, int y, int c)
{
    if (x < 0 || y < 0 || x >= D->w || y >= D->h) return 0;
    return D->d[(size_t)y * D->stride + 2 * x + c];
}

/* Window-sum accumulator that can emulate the summation orders. */
typedef struct {
    int   order;
    int   kind;                   /* WSUM_A / WSUM_B / WSUM_E: which loop of the tracker the sums belong to (matters for DR3O_ORDER_SSE2 only) */
    int   win_w;
    int   blk_shift, lanes;       /* kernel's order: log2(rows per lane-item) and lanes per point (64, 32 = half or 8 = an eighth of a wave) */
    long long lane[3][64];        /* up to three simultaneous sums: the lane-local part is an exact integer sum */
    float acc[3];
    float q[3][4];                /* SSE2 order: the four f32 lanes of the vector accumulators (A: qA11/qA12/qA22; b: the x / y lanes of qb0, qb1) */
    int   pend[3][4];             /* SSE2 order, b sums: the first product of a _mm_madd_epi16 pair, waiting for its partner four columns on */
} wsum;
enum { WSUM_A = 0, WSUM_B = 1, WSUM_E = 2 };

/* The HIP tracker (3dr_amd/csrc/lk_kernels.hip) tracks a point with a GROUP of lanes, and a lane owns taps of ONE window column.
 * Windows up to 16 x 16 (e.g. the bench's 15x15): an EIGHTH of a wave -- 8 lanes, a lane owns the columns c and c + 8 (32 taps, one
 * exact integer sum), eight points per wave64.  Every
 * other window is cut into lane-items of 8 consecutive rows of a column, item q = (ty / 8) * win_w + tx on lane q % lanes: at most
 * 64 items (e.g. 8x32) go to HALF a wave, two points per wave -- 21x21 (63 items) to a QUARTER, four points per wave, four items per
 * lane --; more (e.g. the reference's 30x30) to a whole wave.
 * DR3O_ORDER_WAVE64 follows whichever applies. */
static int kernel_points_per_wave(int win_w, int win_h)
{
    if (win_w > 32 || win_h > 32) return 1;             /* the plain form for windows beyond 32 x 32 (lk_track_big_kernel): always a whole wave */
    if (win_w <= 16 && win_h <= 16) return 8;           /* an EIGHTH of a wave: 8 lanes, lane = columns c and c + 8 */
    if (win_w == 21 && win_h == 21) return 4;           /* configs[0] / configs[4]: 8-row items on a QUARTER wave (lk_trackg_kernel<21, 4, 4>) */
    if (win_w * ((win_h + 7) / 8) <= 64) return 2;
    return 1;
}

static void wsum_init(wsum* s, int order, int kind, int win_w, int win_h)
{
    const int ppw = kernel_points_per_wave(win_w, win_h);
    s->order = order; s->kind = kind; s->win_w = win_w;
    s->blk_shift = (win_w <= 16 && win_h <= 16) ? 4 : 3;      /* whole columns (<= 16 rows) for the small windows, 8-row items otherwise */
    s->lanes = 64 / ppw;
    memset(s->lane, 0, sizeof(s->lane));
    memset(s->q, 0, sizeof(s->q));
    memset(s->pend, 0, sizeof(s->pend));
    s->acc[0] = s->acc[1] = s->acc[2] = 0.f;
}
/* Every term of the three LK window sums is an integer (a product of two fixed-point values, or |diff|).
 * RASTER (OpenCV's scalar path): converted to f32 and added in raster order, each add rounding.
 * WAVE64 (the HIP kernel): lane-item q = (ty >> blk_shift) * win_w + tx; lane = q % lanes;
 * the taps of one lane are summed as integers (v_dot2_i32_i16 accumulates exactly; at most 16 taps of < 2^25.1 each, so
 * int32 cannot overflow in the kernel), converted to f32 once, and the lane values go through the f32 tree of wsum_get.
 * SSE2 (what an x86-64 build of OpenCV 3.0 - 3.4 executes behind src/initialization.cpp:608-613: LKTrackerInvoker's CV_SSE2 blocks,
 * lkpyramid.cpp; restated from the published source, OpenCV being absent here -- tests/test_oracle_sse2.py replays the intrinsic
 * sequences with real <emmintrin.h> intrinsics against this bookkeeping):
 *   A sums (template loop): columns x < (win_w & ~3) go four at a time through _mm_cvtepi32_ps + _mm_mul_ps + _mm_add_ps into a
 *     4-lane f32 accumulator (lane = x & 3; the f32 product of two converted int16 values equals the converted integer product, both
 *     being the correctly rounded exact product), running over all rows; the remaining columns of every row go to the scalar accumulator
 *     in raster order; at the end  iA += ((q[0] + q[1]) + q[2]) + q[3].
 *   b sums (iteration loop): columns x < (win_w & ~7) go eight at a time: _mm_madd_epi16 adds the products of columns j and j + 4 of
 *     the chunk exactly in int32 (j = 0..3), _mm_cvtepi32_ps converts the pair sum, and it is added to lane (x: 0, y: 1) of qb0 for
 *     j = 0, lanes (2, 3) of qb0 for j = 1, lanes (0, 1) of qb1 for j = 2, lanes (2, 3) of qb1 for j = 3; the remaining columns go to
 *     the scalar accumulators; at the end bbuf = qb0 + qb1 lane-wise and  ib1 += bbuf[0] + bbuf[2], ib2 += bbuf[1] + bbuf[3], i.e.
 *     ib += (q[0] + q[2]) + (q[1] + q[3]) with q[j] the accumulator pair j feeds.
 *   err (level-0 residual): scalar in every build -- raster order. */
static inline void wsum_add(wsum* s, int which, int tx, int ty, int v)
{
    if (s->order == DR3O_ORDER_RASTER) { s->acc[which] += (float)v; return; }
    if (s->order == DR3O_ORDER_SSE2) {
        if (s->kind == WSUM_A && tx < (s->win_w & ~3)) s->q[which][tx & 3] += (float)v;
        else if (s->kind == WSUM_B && tx < (s->win_w & ~7)) {
            if ((tx & 7) < 4) s->pend[which][tx & 3] = v;
            else s->q[which][tx & 3] += (float)(s->pend[which][tx & 3] + v);
        } else s->acc[which] += (float)v;
        return;
    }
    {
        const int q = (ty >> s->blk_shift) * s->win_w + tx;
        s->lane[which][q & (s->lanes - 1)] += v;
    }
}
static float wsum_get(wsum* s, int which)
{
    if (s->order == DR3O_ORDER_RASTER) return s->acc[which];
    if (s->order == DR3O_ORDER_SSE2) {
        const float* q = s->q[which];
        if (s->kind == WSUM_A) return s->acc[which] + (((q[0] + q[1]) + q[2]) + q[3]);
        if (s->kind == WSUM_B) return s->acc[which] + ((q[0] + q[2]) + (q[1] + q[3]));
        return s->acc[which];
    }
    {
        /* the HIP kernel's tree (3dr_amd/csrc/lk_kernels.hip, wave_sum2 / wave_sum3 and group_sum2 / group_sum1): whole-wave
         * shape: fold the upper half-wave onto the lower first; whole- and half-wave shapes: fold the upper 16 of the 32 values
         * onto the lower 16; all shapes: an xor butterfly 1,2,4,8 over the 16 values */
        float a[32], c[16], b[16];
        int off, l;
        if (s->lanes == 64) for (l = 0; l < 32; ++l) a[l] = (float)s->lane[which][l] + (float)s->lane[which][l + 32];
        else                for (l = 0; l < 32; ++l) a[l] = (float)s->lane[which][l];
        if (s->lanes <= 16) for (l = 0; l < 16; ++l) c[l] = a[l];
        else                for (l = 0; l < 16; ++l) c[l] = a[l] + a[l + 16];
        for (off = 1; off < (s->lanes == 8 ? 8 : 16); off <<= 1) {
            for (l = 0; l < 16; ++l) b[l] = c[l] + c[l ^ off];
            memcpy(c, b, sizeof(c));
        }
        return c[0];
    }
}


/* debug sink (tools only, single-threaded callers): the iterate (cx, cy) -- window origin, before the step -- of every Gauss-Newton
 * iteration of the point being tracked: traj[(level * 101 + j) * 2 + {0,1}], j = 0 is the level's starting estimate */
static float* g_traj = NULL;
void dr3o_debug_set_traj(float* buf) { g_traj = buf; }

static void lk_track_point(const lvl_img* I, const lvl_img* J, const lvl_deriv* dI,
                           int max_level, const float* prev_pt, float* next_pt,
                           uint8_t* status, float* err,
                           int win_w, int win_h, int max_count, double epsilon,
                           int flags, double min_eig_threshold, int order,
                           int16_t* Ibuf, int16_t* dIbuf, int* iters)
{
    const float half_x = (win_w - 1) * 0.5f, half_y = (win_h - 1) * 0.5f;
    const int W_BITS = 14;
    const float FLT_SCALE = 1.f / (1 << 20);
    float nx = next_pt[0], ny = next_pt[1];   /* nextPts[ptidx] */
    int level;

    *status = 1;
    *err = 0.f;

    for (level = max_level; level >= 0; --level) {
        const lvl_img* Il = &I[level];
        const lvl_img* Jl = &J[level];
        const lvl_deriv* Dl = &dI[level];
        const float scale = (float)(1. / (1 << level));
        float px = prev_pt[0] * scale, py = prev_pt[1] * scale;
        float cx, cy;                 /* nextPt */
        int ipx, ipy, x, y, j;
        float a, b, A11, A12, A22, D, minEig;
        int iw00, iw01, iw10, iw11;
        wsum S;
        float pdx = 0.f, pdy = 0.f;   /* prevDelta */

        if (iters) iters[level] = 0;

        if (level == max_level) {
            if (flags & DR3O_LK_USE_INITIAL_FLOW) { cx = nx * scale; cy = ny * scale; }
            else { cx = px; cy = py; }
        } else {
            cx = nx * 2.f; cy = ny * 2.f;
        }
        nx = cx; ny = cy;

        px -= half_x; py -= half_y;
        ipx = cv_floor_f(px); ipy = cv_floor_f(py);

        if (ipx < -win_w || ipx >= Dl->w || ipy < -win_h || ipy >= Dl->h) {
            if (level == 0) { *status = 0; *err = 0.f; }
            continue;
        }

        a = px - ipx; b = py - ipy;
        iw00 = cv_round_f((1.f - a) * (1.f - b) * (1 << W_BITS));
        iw01 = cv_round_f(a * (1.f - b) * (1 << W_BITS));
        iw10 = cv_round_f((1.f - a) * b * (1 << W_BITS));
        iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;

        wsum_init(&S, order, WSUM_A, win_w, win_h);
        for (y = 0; y < win_h; ++y) {
            for (x = 0; x < win_w; ++x) {
                const int X = ipx + x, Y = ipy + y;
                const int ival = DESCALE(img_at(Il, X, Y) * iw00 + img_at(Il, X + 1, Y) * iw01 +
                                         img_at(Il, X, Y + 1) * iw10 + img_at(Il, X + 1, Y + 1) * iw11,
                                         W_BITS - 5);
                const int ixval = DESCALE(der_at(Dl, X, Y, 0) * iw00 + der_at(Dl, X + 1, Y, 0) * iw01 +
                                          der_at(Dl, X, Y + 1, 0) * iw10 + der_at(Dl, X + 1, Y + 1, 0) * iw11,
                                          W_BITS);
                const int iyval = DESCALE(der_at(Dl, X, Y, 1) * iw00 + der_at(Dl, X + 1, Y, 1) * iw01 +
                                          der_at(Dl, X, Y + 1, 1) * iw10 + der_at(Dl, X + 1, Y + 1, 1) * iw11,
                                          W_BITS);
                Ibuf[y * win_w + x] = (int16_t)ival;
                dIbuf[2 * (y * win_w + x)]     = (int16_t)ixval;
                dIbuf[2 * (y * win_w + x) + 1] = (int16_t)iyval;
                wsum_add(&S, 0, x, y, ixval * ixval);
                wsum_add(&S, 1, x, y, ixval * iyval);
                wsum_add(&S, 2, x, y, iyval * iyval);
            }
        }
        A11 = wsum_get(&S, 0) * FLT_SCALE;
        A12 = wsum_get(&S, 1) * FLT_SCALE;
        A22 = wsum_get(&S, 2) * FLT_SCALE;

        D = A11 * A22 - A12 * A12;
        minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) /
                 (float)(2 * win_w * win_h);

        if (flags & DR3O_LK_GET_MIN_EIGENVALS) *err = minEig;

        if ((double)minEig < min_eig_threshold || D < FLT_EPSILON) {
            if (level == 0) *status = 0;
            continue;
        }

        D = 1.f / D;
        cx -= half_x; cy -= half_y;

        for (j = 0; j < max_count; ++j) {
            const int inx = cv_floor_f(cx), iny = cv_floor_f(cy);
            float b1, b2, dx, dy;

            if (inx < -win_w || inx >= Jl->w || iny < -win_h || iny >= Jl->h) {
                if (level == 0) *status = 0;
                break;
            }
            if (iters) iters[level] = j + 1;
            if (g_traj) { g_traj[(level * 101 + j) * 2] = cx; g_traj[(level * 101 + j) * 2 + 1] = cy; }

            a = cx - inx; b = cy - iny;
            iw00 = cv_round_f((1.f - a) * (1.f - b) * (1 << W_BITS));
            iw01 = cv_round_f(a * (1.f - b) * (1 << W_BITS));
            iw10 = cv_round_f((1.f - a) * b * (1 << W_BITS));
            iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;

            wsum_init(&S, order, WSUM_B, win_w, win_h);
            for (y = 0; y < win_h; ++y) {
                for (x = 0; x < win_w; ++x) {
                    const int X = inx + x, Y = iny + y;
                    const int diff = DESCALE(img_at(Jl, X, Y) * iw00 + img_at(Jl, X + 1, Y) * iw01 +
                                             img_at(Jl, X, Y + 1) * iw10 + img_at(Jl, X + 1, Y + 1) * iw11,
                                             W_BITS - 5) - Ibuf[y * win_w + x];
                    wsum_add(&S, 0, x, y, diff * dIbuf[2 * (y * win_w + x)]);
                    wsum_add(&S, 1, x, y, diff * dIbuf[2 * (y * win_w + x) + 1]);
                }
            }
            b1 = wsum_get(&S, 0) * FLT_SCALE;
            b2 = wsum_get(&S, 1) * FLT_SCALE;

            dx = (float)((A12 * b2 - A22 * b1) * D);
            dy = (float)((A12 * b1 - A11 * b2) * D);

            cx += dx; cy += dy;
            nx = cx + half_x; ny = cy + half_y;

            if ((double)dx * dx + (double)dy * dy <= epsilon) break;

            if (j > 0 && fabs((double)(dx + pdx)) < 0.01 && fabs((double)(dy + pdy)) < 0.01) {
                nx -= dx * 0.5f; ny -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
        }

        if (*status && level == 0 && !(flags & DR3O_LK_GET_MIN_EIGENVALS)) {
            const float fx = nx - half_x, fy = ny - half_y;
            const int inx = cv_floor_f(fx), iny = cv_floor_f(fy);
            float errval;

            if (inx < -win_w || inx >= Jl->w || iny < -win_h || iny >= Jl->h) {
                *status = 0;
                continue;
            }
            a = fx - inx; b = fy - iny;
            iw00 = cv_round_f((1.f - a) * (1.f - b) * (1 << W_BITS));
            iw01 = cv_round_f(a * (1.f - b) * (1 << W_BITS));
            iw10 = cv_round_f((1.f - a) * b * (1 << W_BITS));
            iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;

            wsum_init(&S, order, WSUM_E, win_w, win_h);
            for (y = 0; y < win_h; ++y) {
                for (x = 0; x < win_w; ++x) {
                    const int X = inx + x, Y = iny + y;
                    const int diff = DESCALE(img_at(Jl, X, Y) * iw00 + img_at(Jl, X + 1, Y) * iw01 +
                                             img_at(Jl, X, Y + 1) * iw10 + img_at(Jl, X + 1, Y + 1) * iw11,
                                             W_BITS - 5) - Ibuf[y * win_w + x];
                    wsum_add(&S, 0, x, y, diff < 0 ? -diff : diff);
                }
            }
            errval = wsum_get(&S, 0);
            *err = errval * 1.f / (float)(32 * win_w * win_h);
        }
    }
    next_pt[0] = nx; next_pt[1] = ny;
}

/* The window sums of one template window (A11, A12, A22 from the int16 gradient pairs dI[h][w][2]) and of one iteration (b1, b2 from
 * the int16 mismatches diff[h][w] and the same gradients) in a given summation order: the bookkeeping of wsum_add / wsum_get alone, so
 * that tests can hold it against a replay of OpenCV's intrinsic sequences (oracle/sse2_replay.c). */
void dr3o_window_sums(const int16_t* dI, const int16_t* diff, int win_w, int win_h, int order, float A[3], float b[2])
{
    wsum S;
    int x, y;
    wsum_init(&S, order, WSUM_A, win_w, win_h);
    for (y = 0; y < win_h; ++y)
        for (x = 0; x < win_w; ++x) {
            const int ix = dI[2 * (y * win_w + x)], iy = dI[2 * (y * win_w + x) + 1];
            wsum_add(&S, 0, x, y, ix * ix); wsum_add(&S, 1, x, y, ix * iy); wsum_add(&S, 2, x, y, iy * iy);
        }
    A[0] = wsum_get(&S, 0); A[1] = wsum_get(&S, 1); A[2] = wsum_get(&S, 2);
    wsum_init(&S, order, WSUM_B, win_w, win_h);
    for (y = 0; y < win_h; ++y)
        for (x = 0; x < win_w; ++x) {
            const int d = diff[y * win_w + x];
            wsum_add(&S, 0, x, y, d * dI[2 * (y * win_w + x)]); wsum_add(&S, 1, x, y, d * dI[2 * (y * win_w + x) + 1]);
        }
    b[0] = wsum_get(&S, 0); b[1] = wsum_get(&S, 1);
}

int dr3o_lk_track_ex(const uint8_t* prev, const uint8_t* next, int w, int h,
                     int prev_stride, int next_stride,
                     const float* prev_xy, float* next_xy, int n,
                     uint8_t* status, float* err, const dr3o_lk_params* p,
                     int order, int n_threads, int* iters)
{
    enum { MAXL = 16 };
    lvl_img I[MAXL], J[MAXL];
    lvl_deriv dI[MAXL];
    uint8_t* ibuf[MAXL]; uint8_t* jbuf[MAXL]; int16_t* dbuf[MAXL];
    int max_level, l, i, max_count;
    double epsilon;

    if (!prev || !next || !p || w <= 0 || h <= 0 || n < 0) return -1;
    if (p->max_level < 0 || p->max_level >= MAXL || p->win_w <= 2 || p->win_h <= 2) return -2;
    set_threads(n_threads);

    max_level = dr3o_lk_effective_max_level(w, h, p->win_w, p->win_h, p->max_level);

    /* TermCriteria clamping (SparsePyrLKOpticalFlowImpl::calc) */
    max_count = (p->crit_type & 1) ? (p->max_iters < 0 ? 0 : p->max_iters > 100 ? 100 : p->max_iters) : 30;
    epsilon   = (p->crit_type & 2) ? (p->eps < 0. ? 0. : p->eps > 10. ? 10. : p->eps) : 0.01;
    epsilon *= epsilon;

    memset(ibuf, 0, sizeof(ibuf)); memset(jbuf, 0, sizeof(jbuf)); memset(dbuf, 0, sizeof(dbuf));
    I[0].img = prev; I[0].w = w; I[0].h = h; I[0].stride = prev_stride;
    J[0].img = next; J[0].w = w; J[0].h = h; J[0].stride = next_stride;
    for (l = 1; l <= max_level; ++l) {
        const int lw = (I[l - 1].w + 1) / 2, lh = (I[l - 1].h + 1) / 2;
        ibuf[l] = (uint8_t*)malloc((size_t)lw * lh);
        jbuf[l] = (uint8_t*)malloc((size_t)lw * lh);
        dr3o_pyr_down_u8(I[l - 1].img, I[l - 1].w, I[l - 1].h, I[l - 1].stride, ibuf[l], lw);
        dr3o_pyr_down_u8(J[l - 1].img, J[l - 1].w, J[l - 1].h, J[l - 1].stride, jbuf[l], lw);
        I[l].img = ibuf[l]; I[l].w = lw; I[l].h = lh; I[l].stride = lw;
        J[l].img = jbuf[l]; J[l].w = lw; J[l].h = lh; J[l].stride = lw;
    }
    for (l = 0; l <= max_level; ++l) {
        dbuf[l] = (int16_t*)malloc((size_t)I[l].w * I[l].h * 2 * sizeof(int16_t));
        dr3o_scharr_s16(I[l].img, I[l].w, I[l].h, I[l].stride, dbuf[l], 2 * I[l].w);
        dI[l].d = dbuf[l]; dI[l].w = I[l].w; dI[l].h = I[l].h; dI[l].stride = 2 * I[l].w;
    }

#pragma omp parallel
    {
        int16_t* Ibuf  = (int16_t*)malloc((size_t)p->win_w * p->win_h * sizeof(int16_t));
        int16_t* dIbuf = (int16_t*)malloc((size_t)p->win_w * p->win_h * 2 * sizeof(int16_t));
#pragma omp for schedule(dynamic, 16)
        for (i = 0; i < n; ++i) {
            lk_track_point(I, J, dI, max_level, prev_xy + 2 * i, next_xy + 2 * i,
                           status + i, err + i, p->win_w, p->win_h, max_count, epsilon,
                           p->flags, p->min_eig_threshold, order, Ibuf, dIbuf,
                           iters ? iters + (size_t)i * (max_level + 1) : NULL);
        }
        free(Ibuf); free(dIbuf);
    }

    for (l = 0; l <= max_level; ++l) { free(ibuf[l]); free(jbuf[l]); free(dbuf[l]); }
    return max_level;
}

int dr3o_lk_track(const uint8_t* prev, const uint8_t* next, int w, int h,
                  int prev_stride, int next_stride,
                  const float* prev_xy, float* next_xy, int n,
                  uint8_t* status, float* err, const dr3o_lk_params* p,
                  int order, int n_threads)
{
    return dr3o_lk_track_ex(prev, next, w, h, prev_stride, next_stride, prev_xy, next_xy, n,
                            status, err, p, order, n_threads, NULL);
}

/* ------------------------------------------------------------------------------------------ */
/* "port-opt": the same results, computed the way an optimised CPU library computes them       */
/* ------------------------------------------------------------------------------------------ */
/* dr3o_lk_track_opt returns exactly what dr3o_lk_track_ex returns (tests/test_oracle_opt.py: bit for bit, every order), organised
 * the way OpenCV organises the work instead of the way that is easiest to read: pyrDown separable with a ring of horizontally
 * filtered rows, Scharr through two row buffers, every level copied once into a plane padded by winSize (REFLECT_101 pixels, zero
 * derivatives -- buildOpticalFlowPyramid's layout) so that the tracker's windows are plain pointers with no per-tap reflection, and,
 * on x86-64 in DR3O_ORDER_SSE2, the window loops on the SSE2 intrinsic sequences of LKTrackerInvoker.  It exists so that bench.py's
 * cpu_baseline has a less flattering denominator than the readable port (round 3's verdict, weak #7); the readable functions above
 * stay the checker. */
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

static void pyr_down_u8_fast(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dstride)
{
    const int dw = (sw + 1) / 2, dh = (sh + 1) / 2;
    if (sw < 5 || sh < 3) { dr3o_pyr_down_u8(src, sw, sh, sstride, dst, dstride); return; }
#pragma omp parallel
    {
        int* ring = (int*)malloc((size_t)5 * dw * sizeof(int));
        int tag[5] = {-1, -1, -1, -1, -1};
        int y;
#pragma omp for schedule(static)
        for (y = 0; y < dh; ++y) {
            const int* rk[5];
            int k, x;
            for (k = 0; k < 5; ++k) {
                const int sy = reflect101(2 * y + k - 2, sh), slot = sy % 5;
                int* r = ring + (size_t)slot * dw;
                if (tag[slot] != sy) {
                    const uint8_t* s = src + (size_t)sy * sstride;
                    const int x_hi = (sw - 3) / 2;          /* last x with 2x + 2 <= sw - 1 */
                    tag[slot] = sy;
                    r[0] = s[0] * 6 + (s[reflect101(-1, sw)] + s[1]) * 4 + s[reflect101(-2, sw)] + s[2];
                    for (x = 1; x <= x_hi; ++x)
                        r[x] = s[2 * x] * 6 + (s[2 * x - 1] + s[2 * x + 1]) * 4 + s[2 * x - 2] + s[2 * x + 2];
                    for (; x < dw; ++x)
                        r[x] = s[reflect101(2 * x, sw)] * 6 + (s[reflect101(2 * x - 1, sw)] + s[reflect101(2 * x + 1, sw)]) * 4 +
                               s[reflect101(2 * x - 2, sw)] + s[reflect101(2 * x + 2, sw)];
                }
                rk[k] = r;
            }
            {
                uint8_t* d = dst + (size_t)y * dstride;
                const int *r0 = rk[0], *r1 = rk[1], *r2 = rk[2], *r3 = rk[3], *r4 = rk[4];
                for (x = 0; x < dw; ++x)
                    d[x] = (uint8_t)((r2[x] * 6 + (r1[x] + r3[x]) * 4 + r0[x] + r4[x] + 128) >> 8);
            }
        }
        free(ring);
    }
}

/* Scharr into a plane padded by (pad_x, pad_y) zeros (calloc'ed by the caller); d0 = the interior origin, stride in int16 */
static void scharr_s16_fast(const uint8_t* src, int w, int h, int stride, int16_t* d0, int dstride)
{
    if (w < 2 || h < 2) { dr3o_scharr_s16(src, w, h, stride, d0, dstride); return; }
#pragma omp parallel
    {
        int* t0 = (int*)malloc((size_t)(w + 2) * sizeof(int));      /* smoothed in y, with the REFLECT_101 columns at -1 and w */
        int* t1 = (int*)malloc((size_t)(w + 2) * sizeof(int));      /* differentiated in y */
        int y;
#pragma omp for schedule(static)
        for (y = 0; y < h; ++y) {
            const uint8_t* r0 = src + (size_t)(y > 0 ? y - 1 : 1) * stride;
            const uint8_t* r1 = src + (size_t)y * stride;
            const uint8_t* r2 = src + (size_t)(y < h - 1 ? y + 1 : h - 2) * stride;
            int16_t* d = d0 + (size_t)y * dstride;
            int x;
            for (x = 0; x < w; ++x) { t0[x + 1] = (r0[x] + r2[x]) * 3 + r1[x] * 10; t1[x + 1] = r2[x] - r0[x]; }
            t0[0] = t0[2]; t1[0] = t1[2]; t0[w + 1] = t0[w - 1]; t1[w + 1] = t1[w - 1];
            for (x = 0; x < w; ++x) {
                d[2 * x]     = (int16_t)(t0[x + 2] - t0[x]);
                d[2 * x + 1] = (int16_t)((t1[x + 2] + t1[x]) * 3 + t1[x + 1] * 10);
            }
        }
        free(t0); free(t1);
    }
}

typedef struct { const uint8_t* img; const int16_t* der; int w, h, stride, dstride; } plvl;   /* origins of the interiors of padded planes */

static uint8_t* pad_reflect101(const uint8_t* src, int w, int h, int stride, int px, int py, int* pstride)
{
    const int pw = w + 2 * px, ph = h + 2 * py;
    uint8_t* out = (uint8_t*)malloc((size_t)pw * ph + 16);
    int y;
#pragma omp parallel for schedule(static)
    for (y = 0; y < ph; ++y) {
        const uint8_t* s = src + (size_t)reflect101(y - py, h) * stride;
        uint8_t* d = out + (size_t)y * pw;
        int x;
        for (x = 0; x < px; ++x) d[x] = s[reflect101(x - px, w)];
        memcpy(d + px, s, (size_t)w);
        for (x = 0; x < px; ++x) d[px + w + x] = s[reflect101(w + x, w)];
    }
    *pstride = pw;
    return out;
}

static void lk_track_point_fast(const plvl* I, const plvl* J, int max_level, const float* prev_pt, float* next_pt,
                                uint8_t* status, float* err, int win_w, int win_h, int max_count, double epsilon,
                                int flags, double min_eig_threshold, int order, int16_t* Ibuf, int16_t* dIbuf, int* iters)
{
    const float half_x = (win_w - 1) * 0.5f, half_y = (win_h - 1) * 0.5f;
    const int W_BITS = 14;
    const float FLT_SCALE = 1.f / (1 << 20);
    float nx = next_pt[0], ny = next_pt[1];
    int level;
#if defined(__SSE2__)
    const int simd = order == DR3O_ORDER_SSE2;
#else
    const int simd = 0;
#endif
    *status = 1; *err = 0.f;
    for (level = max_level; level >= 0; --level) {
        const plvl* Il = &I[level];
        const plvl* Jl = &J[level];
        const int sI = Il->stride, sJ = Jl->stride, sD = Il->dstride;
        const float scale = (float)(1. / (1 << level));
        float px = prev_pt[0] * scale, py = prev_pt[1] * scale, cx, cy, a, b, A11, A12, A22, D, minEig, pdx = 0.f, pdy = 0.f;
        int ipx, ipy, x, y, j, iw00, iw01, iw10, iw11;
        wsum S;

        if (iters) iters[level] = 0;
        if (level == max_level) {
            if (flags & DR3O_LK_USE_INITIAL_FLOW) { cx = nx * scale; cy = ny * scale; } else { cx = px; cy = py; }
        } else { cx = nx * 2.f; cy = ny * 2.f; }
        nx = cx; ny = cy;
        px -= half_x; py -= half_y;
        ipx = cv_floor_f(px); ipy = cv_floor_f(py);
        if (ipx < -win_w || ipx >= Il->w || ipy < -win_h || ipy >= Il->h) {
            if (level == 0) { *status = 0; *err = 0.f; }
            continue;
        }
        a = px - ipx; b = py - ipy;
        iw00 = cv_round_f((1.f - a) * (1.f - b) * (1 << W_BITS));
        iw01 = cv_round_f(a * (1.f - b) * (1 << W_BITS));
        iw10 = cv_round_f((1.f - a) * b * (1 << W_BITS));
        iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;

        if (simd) {
#if defined(__SSE2__)
            const __m128i qw0 = _mm_set1_epi32(iw00 + (int)((unsigned)iw01 << 16)), qw1 = _mm_set1_epi32(iw10 + (int)((unsigned)iw11 << 16));
            const __m128i z = _mm_setzero_si128(), qdelta_d = _mm_set1_epi32(1 << (W_BITS - 1)), qdelta = _mm_set1_epi32(1 << (W_BITS - 5 - 1));
            __m128 qA11 = _mm_setzero_ps(), qA12 = _mm_setzero_ps(), qA22 = _mm_setzero_ps();
            float iA11 = 0.f, iA12 = 0.f, iA22 = 0.f, b11[4], b12[4], b22[4];
            for (y = 0; y < win_h; ++y) {
                const uint8_t* src = Il->img + (ptrdiff_t)(ipy + y) * sI + ipx;
                const int16_t* dsrc = Il->der + (ptrdiff_t)(ipy + y) * sD + 2 * ipx;
                int16_t* Iptr = Ibuf + y * win_w;
                int16_t* dIptr = dIbuf + 2 * y * win_w;
                x = 0;
                for (; x <= win_w - 4; x += 4, dsrc += 4 * 2, dIptr += 4 * 2) {
                    int t32[4];
                    __m128i v00, v01, v10, v11, t0, t1;
                    memcpy(&t32[0], src + x, 4); memcpy(&t32[1], src + x + 1, 4); memcpy(&t32[2], src + x + sI, 4); memcpy(&t32[3], src + x + sI + 1, 4);
                    v00 = _mm_unpacklo_epi8(_mm_cvtsi32_si128(t32[0]), z);
                    v01 = _mm_unpacklo_epi8(_mm_cvtsi32_si128(t32[1]), z);
                    v10 = _mm_unpacklo_epi8(_mm_cvtsi32_si128(t32[2]), z);
                    v11 = _mm_unpacklo_epi8(_mm_cvtsi32_si128(t32[3]), z);
                    t0 = _mm_add_epi32(_mm_madd_epi16(_mm_unpacklo_epi16(v00, v01), qw0), _mm_madd_epi16(_mm_unpacklo_epi16(v10, v11), qw1));
                    t0 = _mm_srai_epi32(_mm_add_epi32(t0, qdelta), W_BITS - 5);
                    _mm_storel_epi64((__m128i*)(Iptr + x), _mm_packs_epi32(t0, t0));
                    v00 = _mm_loadu_si128((const __m128i*)(dsrc));
                    v01 = _mm_loadu_si128((const __m128i*)(dsrc + 2));
                    v10 = _mm_loadu_si128((const __m128i*)(dsrc + sD));
                    v11 = _mm_loadu_si128((const __m128i*)(dsrc + sD + 2));
                    t0 = _mm_add_epi32(_mm_madd_epi16(_mm_unpacklo_epi16(v00, v01), qw0), _mm_madd_epi16(_mm_unpacklo_epi16(v10, v11), qw1));
                    t1 = _mm_add_epi32(_mm_madd_epi16(_mm_unpackhi_epi16(v00, v01), qw0), _mm_madd_epi16(_mm_unpackhi_epi16(v10, v11), qw1));
                    t0 = _mm_srai_epi32(_mm_add_epi32(t0, qdelta_d), W_BITS);
                    t1 = _mm_srai_epi32(_mm_add_epi32(t1, qdelta_d), W_BITS);
                    v00 = _mm_packs_epi32(t0, t1);
                    _mm_storeu_si128((__m128i*)dIptr, v00);
                    t0 = _mm_srai_epi32(v00, 16);
                    t1 = _mm_srai_epi32(_mm_slli_epi32(v00, 16), 16);
                    {
                        const __m128 fy = _mm_cvtepi32_ps(t0), fx = _mm_cvtepi32_ps(t1);
                        qA22 = _mm_add_ps(qA22, _mm_mul_ps(fy, fy));
                        qA12 = _mm_add_ps(qA12, _mm_mul_ps(fx, fy));
                        qA11 = _mm_add_ps(qA11, _mm_mul_ps(fx, fx));
                    }
                }
                for (; x < win_w; ++x, dsrc += 2, dIptr += 2) {
                    const int ival = DESCALE(src[x] * iw00 + src[x + 1] * iw01 + src[x + sI] * iw10 + src[x + sI + 1] * iw11, W_BITS - 5);
                    const int ixval = DESCALE(dsrc[0] * iw00 + dsrc[2] * iw01 + dsrc[sD] * iw10 + dsrc[sD + 2] * iw11, W_BITS);
                    const int iyval = DESCALE(dsrc[1] * iw00 + dsrc[3] * iw01 + dsrc[sD + 1] * iw10 + dsrc[sD + 3] * iw11, W_BITS);
                    Iptr[x] = (int16_t)ival; dIptr[0] = (int16_t)ixval; dIptr[1] = (int16_t)iyval;
                    iA11 += (float)(ixval * ixval); iA12 += (float)(ixval * iyval); iA22 += (float)(iyval * iyval);
                }
            }
            _mm_storeu_ps(b11, qA11); _mm_storeu_ps(b12, qA12); _mm_storeu_ps(b22, qA22);
            iA11 += b11[0] + b11[1] + b11[2] + b11[3];
            iA12 += b12[0] + b12[1] + b12[2] + b12[3];
            iA22 += b22[0] + b22[1] + b22[2] + b22[3];
            A11 = iA11 * FLT_SCALE; A12 = iA12 * FLT_SCALE; A22 = iA22 * FLT_SCALE;
#endif
        } else {
            wsum_init(&S, order, WSUM_A, win_w, win_h);
            for (y = 0; y < win_h; ++y) {
                const uint8_t* src = Il->img + (ptrdiff_t)(ipy + y) * sI + ipx;
                const int16_t* dsrc = Il->der + (ptrdiff_t)(ipy + y) * sD + 2 * ipx;
                for (x = 0; x < win_w; ++x, dsrc += 2) {
                    const int ival = DESCALE(src[x] * iw00 + src[x + 1] * iw01 + src[x + sI] * iw10 + src[x + sI + 1] * iw11, W_BITS - 5);
                    const int ixval = DESCALE(dsrc[0] * iw00 + dsrc[2] * iw01 + dsrc[sD] * iw10 + dsrc[sD + 2] * iw11, W_BITS);
                    const int iyval = DESCALE(dsrc[1] * iw00 + dsrc[3] * iw01 + dsrc[sD + 1] * iw10 + dsrc[sD + 3] * iw11, W_BITS);
                    Ibuf[y * win_w + x] = (int16_t)ival;
                    dIbuf[2 * (y * win_w + x)] = (int16_t)ixval; dIbuf[2 * (y * win_w + x) + 1] = (int16_t)iyval;
                    if (order == DR3O_ORDER_RASTER) { S.acc[0] += (float)(ixval * ixval); S.acc[1] += (float)(ixval * iyval); S.acc[2] += (float)(iyval * iyval); }
                    else { wsum_add(&S, 0, x, y, ixval * ixval); wsum_add(&S, 1, x, y, ixval * iyval); wsum_add(&S, 2, x, y, iyval * iyval); }
                }
            }
            A11 = wsum_get(&S, 0) * FLT_SCALE; A12 = wsum_get(&S, 1) * FLT_SCALE; A22 = wsum_get(&S, 2) * FLT_SCALE;
        }

        D = A11 * A22 - A12 * A12;
        minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * win_w * win_h);
        if (flags & DR3O_LK_GET_MIN_EIGENVALS) *err = minEig;
        if ((double)minEig < min_eig_threshold || D < FLT_EPSILON) {
            if (level == 0) *status = 0;
            continue;
        }
        D = 1.f / D;
        cx -= half_x; cy -= half_y;

        for (j = 0; j < max_count; ++j) {
            const int inx = cv_floor_f(cx), iny = cv_floor_f(cy);
            float b1, b2, dx, dy;
            if (inx < -win_w || inx >= Jl->w || iny < -win_h || iny >= Jl->h) {
                if (level == 0) *status = 0;
                break;
            }
            if (iters) iters[level] = j + 1;
            a = cx - inx; b = cy - iny;
            iw00 = cv_round_f((1.f - a) * (1.f - b) * (1 << W_BITS));
            iw01 = cv_round_f(a * (1.f - b) * (1 << W_BITS));
            iw10 = cv_round_f((1.f - a) * b * (1 << W_BITS));
            iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            if (simd) {
#if defined(__SSE2__)
                const __m128i qw0 = _mm_set1_epi32(iw00 + (int)((unsigned)iw01 << 16)), qw1 = _mm_set1_epi32(iw10 + (int)((unsigned)iw11 << 16));
                const __m128i z = _mm_setzero_si128(), qdelta = _mm_set1_epi32(1 << (W_BITS - 5 - 1));
                __m128 qb0 = _mm_setzero_ps(), qb1 = _mm_setzero_ps();
                float ib1 = 0.f, ib2 = 0.f, bbuf[4];
                for (y = 0; y < win_h; ++y) {
                    const uint8_t* Jptr = Jl->img + (ptrdiff_t)(iny + y) * sJ + inx;
                    const int16_t* Iptr = Ibuf + y * win_w;
                    const int16_t* dIptr = dIbuf + 2 * y * win_w;
                    x = 0;
                    for (; x <= win_w - 8; x += 8, dIptr += 8 * 2) {
                        __m128i diff0 = _mm_loadu_si128((const __m128i*)(Iptr + x)), diff1;
                        __m128i v00 = _mm_unpacklo_epi8(_mm_loadl_epi64((const __m128i*)(Jptr + x)), z);
                        __m128i v01 = _mm_unpacklo_epi8(_mm_loadl_epi64((const __m128i*)(Jptr + x + 1)), z);
                        __m128i v10 = _mm_unpacklo_epi8(_mm_loadl_epi64((const __m128i*)(Jptr + x + sJ)), z);
                        __m128i v11 = _mm_unpacklo_epi8(_mm_loadl_epi64((const __m128i*)(Jptr + x + sJ + 1)), z);
                        __m128i t0 = _mm_add_epi32(_mm_madd_epi16(_mm_unpacklo_epi16(v00, v01), qw0), _mm_madd_epi16(_mm_unpacklo_epi16(v10, v11), qw1));
                        __m128i t1 = _mm_add_epi32(_mm_madd_epi16(_mm_unpackhi_epi16(v00, v01), qw0), _mm_madd_epi16(_mm_unpackhi_epi16(v10, v11), qw1));
                        t0 = _mm_srai_epi32(_mm_add_epi32(t0, qdelta), W_BITS - 5);
                        t1 = _mm_srai_epi32(_mm_add_epi32(t1, qdelta), W_BITS - 5);
                        diff0 = _mm_subs_epi16(_mm_packs_epi32(t0, t1), diff0);
                        diff1 = _mm_unpackhi_epi16(diff0, diff0);
                        diff0 = _mm_unpacklo_epi16(diff0, diff0);
                        v00 = _mm_loadu_si128((const __m128i*)(dIptr));
                        v01 = _mm_loadu_si128((const __m128i*)(dIptr + 8));
                        v10 = _mm_unpacklo_epi16(v00, v01);
                        v11 = _mm_unpackhi_epi16(v00, v01);
                        v00 = _mm_unpacklo_epi16(diff0, diff1);
                        v01 = _mm_unpackhi_epi16(diff0, diff1);
                        v00 = _mm_madd_epi16(v00, v10);
                        v11 = _mm_madd_epi16(v01, v11);
                        qb0 = _mm_add_ps(qb0, _mm_cvtepi32_ps(v00));
                        qb1 = _mm_add_ps(qb1, _mm_cvtepi32_ps(v11));
                    }
                    for (; x < win_w; ++x, dIptr += 2) {
                        const int diff = DESCALE(Jptr[x] * iw00 + Jptr[x + 1] * iw01 + Jptr[x + sJ] * iw10 + Jptr[x + sJ + 1] * iw11, W_BITS - 5) - Iptr[x];
                        ib1 += (float)(diff * dIptr[0]); ib2 += (float)(diff * dIptr[1]);
                    }
                }
                _mm_storeu_ps(bbuf, _mm_add_ps(qb0, qb1));
                ib1 += bbuf[0] + bbuf[2]; ib2 += bbuf[1] + bbuf[3];
                b1 = ib1 * FLT_SCALE; b2 = ib2 * FLT_SCALE;
#endif
            } else {
                wsum_init(&S, order, WSUM_B, win_w, win_h);
                for (y = 0; y < win_h; ++y) {
                    const uint8_t* Jptr = Jl->img + (ptrdiff_t)(iny + y) * sJ + inx;
                    const int16_t* Iptr = Ibuf + y * win_w;
                    const int16_t* dIptr = dIbuf + 2 * y * win_w;
                    for (x = 0; x < win_w; ++x, dIptr += 2) {
                        const int diff = DESCALE(Jptr[x] * iw00 + Jptr[x + 1] * iw01 + Jptr[x + sJ] * iw10 + Jptr[x + sJ + 1] * iw11, W_BITS - 5) - Iptr[x];
                        if (order == DR3O_ORDER_RASTER) { S.acc[0] += (float)(diff * dIptr[0]); S.acc[1] += (float)(diff * dIptr[1]); }
                        else { wsum_add(&S, 0, x, y, diff * dIptr[0]); wsum_add(&S, 1, x, y, diff * dIptr[1]); }
                    }
                }
                b1 = wsum_get(&S, 0) * FLT_SCALE; b2 = wsum_get(&S, 1) * FLT_SCALE;
            }
            dx = (float)((A12 * b2 - A22 * b1) * D);
            dy = (float)((A12 * b1 - A11 * b2) * D);
            cx += dx; cy += dy;
            nx = cx + half_x; ny = cy + half_y;
            if ((double)dx * dx + (double)dy * dy <= epsilon) break;
            if (j > 0 && fabs((double)(dx + pdx)) < 0.01 && fabs((double)(dy + pdy)) < 0.01) {
                nx -= dx * 0.5f; ny -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
        }

        if (*status && level == 0 && !(flags & DR3O_LK_GET_MIN_EIGENVALS)) {
            const float fx = nx - half_x, fy = ny - half_y;
            const int inx = cv_floor_f(fx), iny = cv_floor_f(fy);
            if (inx < -win_w || inx >= Jl->w || iny < -win_h || iny >= Jl->h) { *status = 0; continue; }
            a = fx - inx; b = fy - iny;
            iw00 = cv_round_f((1.f - a) * (1.f - b) * (1 << W_BITS));
            iw01 = cv_round_f(a * (1.f - b) * (1 << W_BITS));
            iw10 = cv_round_f((1.f - a) * b * (1 << W_BITS));
            iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            wsum_init(&S, order, WSUM_E, win_w, win_h);
            for (y = 0; y < win_h; ++y) {
                const uint8_t* Jptr = Jl->img + (ptrdiff_t)(iny + y) * sJ + inx;
                const int16_t* Iptr = Ibuf + y * win_w;
                for (x = 0; x < win_w; ++x) {
                    const int diff = DESCALE(Jptr[x] * iw00 + Jptr[x + 1] * iw01 + Jptr[x + sJ] * iw10 + Jptr[x + sJ + 1] * iw11, W_BITS - 5) - Iptr[x];
                    wsum_add(&S, 0, x, y, diff < 0 ? -diff : diff);
                }
            }
            *err = wsum_get(&S, 0) * 1.f / (float)(32 * win_w * win_h);
        }
    }
    next_pt[0] = nx; next_pt[1] = ny;
}

int dr3o_lk_track_opt(const uint8_t* prev, const uint8_t* next, int w, int h, int prev_stride, int next_stride,
                      const float* prev_xy, float* next_xy, int n, uint8_t* status, float* err, const dr3o_lk_params* p,
                      int order, int n_threads, int* iters)
{
    enum { MAXL = 16 };
    plvl I[MAXL], J[MAXL];
    uint8_t *ibuf[MAXL], *jbuf[MAXL], *ipad[MAXL], *jpad[MAXL];
    int16_t* dpad[MAXL];
    const uint8_t *isrc[MAXL], *jsrc[MAXL];
    int lw[MAXL], lh[MAXL], lstr[MAXL], jstr[MAXL];
    int max_level, l, i, max_count;
    double epsilon;

    if (!prev || !next || !p || w <= 0 || h <= 0 || n < 0) return -1;
    if (p->max_level < 0 || p->max_level >= MAXL || p->win_w <= 2 || p->win_h <= 2) return -2;
    set_threads(n_threads);
    max_level = dr3o_lk_effective_max_level(w, h, p->win_w, p->win_h, p->max_level);
    max_count = (p->crit_type & 1) ? (p->max_iters < 0 ? 0 : p->max_iters > 100 ? 100 : p->max_iters) : 30;
    epsilon   = (p->crit_type & 2) ? (p->eps < 0. ? 0. : p->eps > 10. ? 10. : p->eps) : 0.01;
    epsilon *= epsilon;

    memset(ibuf, 0, sizeof(ibuf)); memset(jbuf, 0, sizeof(jbuf));
    isrc[0] = prev; jsrc[0] = next; lw[0] = w; lh[0] = h; lstr[0] = prev_stride; jstr[0] = next_stride;
    for (l = 1; l <= max_level; ++l) {
        lw[l] = (lw[l - 1] + 1) / 2; lh[l] = (lh[l - 1] + 1) / 2; lstr[l] = jstr[l] = lw[l];
        ibuf[l] = (uint8_t*)malloc((size_t)lw[l] * lh[l]);
        jbuf[l] = (uint8_t*)malloc((size_t)lw[l] * lh[l]);
        pyr_down_u8_fast(isrc[l - 1], lw[l - 1], lh[l - 1], lstr[l - 1], ibuf[l], lw[l]);
        pyr_down_u8_fast(jsrc[l - 1], lw[l - 1], lh[l - 1], jstr[l - 1], jbuf[l], lw[l]);
        isrc[l] = ibuf[l]; jsrc[l] = jbuf[l];
    }
    for (l = 0; l <= max_level; ++l) {
        /* one more column / row of slack than winSize: the SSE2 chunks read 4 or 8 bytes from the last tap's column */
        const int px = p->win_w + 8, py = p->win_h + 1;
        int ps, js;
        const size_t dstride = (size_t)2 * (lw[l] + 2 * px);
        ipad[l] = pad_reflect101(isrc[l], lw[l], lh[l], lstr[l], px, py, &ps);
        jpad[l] = pad_reflect101(jsrc[l], lw[l], lh[l], jstr[l], px, py, &js);
        dpad[l] = (int16_t*)calloc(dstride * (lh[l] + 2 * py) + 16, sizeof(int16_t));
        I[l].img = ipad[l] + (size_t)py * ps + px; I[l].stride = ps; I[l].w = lw[l]; I[l].h = lh[l];
        J[l].img = jpad[l] + (size_t)py * js + px; J[l].stride = js; J[l].w = lw[l]; J[l].h = lh[l];
        I[l].der = dpad[l] + (size_t)py * dstride + 2 * px; I[l].dstride = (int)dstride;
        J[l].der = NULL; J[l].dstride = 0;
        scharr_s16_fast(isrc[l], lw[l], lh[l], lstr[l], (int16_t*)I[l].der, (int)dstride);
    }
#pragma omp parallel
    {
        int16_t* Ibuf  = (int16_t*)malloc(((size_t)p->win_w * p->win_h + 8) * sizeof(int16_t));
        int16_t* dIbuf = (int16_t*)malloc(((size_t)p->win_w * p->win_h + 8) * 2 * sizeof(int16_t));
#pragma omp for schedule(dynamic, 16)
        for (i = 0; i < n; ++i)
            lk_track_point_fast(I, J, max_level, prev_xy + 2 * i, next_xy + 2 * i, status + i, err + i, p->win_w, p->win_h, max_count, epsilon,
                                p->flags, p->min_eig_threshold, order, Ibuf, dIbuf, iters ? iters + (size_t)i * (max_level + 1) : NULL);
        free(Ibuf); free(dIbuf);
    }
    for (l = 0; l <= max_level; ++l) { free(ibuf[l]); free(jbuf[l]); free(ipad[l]); free(jpad[l]); free(dpad[l]); }
    return max_level;
}

/* ------------------------------------------------------------------------------------------ */
/* reference in-repo code: half-sample pyramid, Shi-Tomasi                                    */
/* ------------------------------------------------------------------------------------------ */

static inline uint8_t half_trunc(int a, int b, int c, int d) { return (uint8_t)((a + b + c + d) / 4); }
/* _mm_avg_epu8(top,bottom) then _mm_avg_epu16(even,odd): src/utils.cpp:337,340 */
static inline uint8_t half_sse2(int a, int b, int c, int d)
{
    const int v0 = (a + c + 1) >> 1, v1 = (b + d + 1) >> 1;
    return (uint8_t)((v0 + v1 + 1) >> 1);
}

void dr3o_halfsample_u8(const uint8_t* src, int w, int h, int stride,
                        uint8_t* dst, int dstride, int mode)
{
    const int ow = w / 2, oh = h / 2;
    int y, x;
    if (mode == DR3O_HALF_REF_X86)
        mode = (w % 16 == 0) ? DR3O_HALF_SSE2 : DR3O_HALF_REF_SCALAR;

    if (mode == DR3O_HALF_REF_SCALAR) {
        /* src/utils.cpp:402-418: `top` advances by 2 per output pixel and by `stride` per output
         * row, i.e. 2*ow + stride per row; for odd w that is 2*stride-1 (sheared sampling). */
        const size_t total = (size_t)stride * h;
        for (y = 0; y < oh; ++y) {
            const size_t row_off = (size_t)y * (2 * (size_t)ow + stride);
            for (x = 0; x < ow; ++x) {
                const size_t t = row_off + 2 * (size_t)x, bt = t + stride;
                const int a = t < total ? src[t] : 0,       b = t + 1 < total ? src[t + 1] : 0;
                const int c = bt < total ? src[bt] : 0,     d = bt + 1 < total ? src[bt + 1] : 0;
                dst[(size_t)y * dstride + x] = half_trunc(a, b, c, d);
            }
        }
        return;
    }
    for (y = 0; y < oh; ++y) {
        const uint8_t* t = src + (size_t)(2 * y) * stride;
        const uint8_t* bt = t + stride;
        for (x = 0; x < ow; ++x) {
            const int a = t[2 * x], b = t[2 * x + 1], c = bt[2 * x], d = bt[2 * x + 1];
            dst[(size_t)y * dstride + x] = (mode == DR3O_HALF_SSE2) ? half_sse2(a, b, c, d)
                                                                    : half_trunc(a, b, c, d);
        }
    }
}

/* src/utils.cpp:282-321.  Unqualified sqrt() on a float inside namespace utils: resolves to
 * ::sqrt(double) (no `using namespace std` in that TU) -- restated so; flagged unpinned. */
float dr3o_shi_tomasi_score(const uint8_t* img, int w, int h, int stride, int u, int v)
{
    float dXX = 0.f, dYY = 0.f, dXY = 0.f;
    const int halfbox = 4, box = 8, area = 64;
    const int x_min = u - halfbox, x_max = u + halfbox, y_min = v - halfbox, y_max = v + halfbox;
    int x, y;
    if (x_min < 1 || x_max >= w - 1 || y_min < 1 || y_max >= h - 1) return 0.f;
    for (y = y_min; y < y_max; ++y) {
        for (x = 0; x < box; ++x) {
            const float dx = (float)(img[(size_t)y * stride + x_min + 1 + x] - img[(size_t)y * stride + x_min - 1 + x]);
            const float dy = (float)(img[(size_t)(y + 1) * stride + x_min + x] - img[(size_t)(y - 1) * stride + x_min + x]);
            dXX += dx * dx; dYY += dy * dy; dXY += dx * dy;
        }
    }
    dXX = (float)(dXX / (2.0 * area));
    dYY = (float)(dYY / (2.0 * area));
    dXY = (float)(dXY / (2.0 * area));
    return (float)(0.5 * (dXX + dYY - sqrt((double)((dXX + dYY) * (dXX + dYY) - 4 * (dXX * dYY - dXY * dXY)))));
}

/* ------------------------------------------------------------------------------------------ */
/* FAST-10 + 3x3 non-max + grid-best Shi-Tomasi (src/features.cpp:43-98)                        */
/* ------------------------------------------------------------------------------------------ */

/* Rosten's ring order (make_offsets): (0,3) (1,3) (2,2) (3,1) (3,0) (3,-1) (2,-2) (1,-3) (0,-3) ... */
static const int k_ring_dx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
static const int k_ring_dy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

/* the corner test of fast_corner_detect_10 at threshold b: 10 contiguous ring pixels > c + b or < c - b */
static int fast10_is_corner(const uint8_t* p, int stride, int b)
{
    const int c = p[0], cb = c + b, c_b = c - b;
    int start, k;
    for (start = 0; start < 16; ++start) {
        int brighter = 1, darker = 1;
        for (k = 0; k < 10 && (brighter || darker); ++k) {
            const int i = (start + k) & 15;
            const int v = p[k_ring_dy[i] * stride + k_ring_dx[i]];
            if (!(v > cb)) brighter = 0;
            if (!(v < c_b)) darker = 0;
        }
        if (brighter || darker) return 1;
    }
    return 0;
}

/* fast_corner_score_10: the library's bisection */
static int fast10_score(const uint8_t* p, int stride, int bstart)
{
    int bmin = bstart, bmax = 255, b = (bmax + bmin) / 2;
    for (;;) {
        if (fast10_is_corner(p, stride, b)) bmin = b; else bmax = b;
        if (bmin == bmax - 1 || bmin == bmax) return bmin;
        b = (bmin + bmax) / 2;
    }
}

int dr3o_fast10_score_map(const uint8_t* img, int w, int h, int stride, int threshold, uint8_t* score_map)
{
    int x, y, n = 0;
    memset(score_map, 0, (size_t)w * h);
    for (y = 3; y < h - 3; ++y)                     /* fast_corner_detect_10 scans [3, size - 3) */
        for (x = 3; x < w - 3; ++x) {
            const uint8_t* p = img + (size_t)y * stride + x;
            if (fast10_is_corner(p, stride, threshold)) {
                score_map[(size_t)y * w + x] = (uint8_t)fast10_score(p, stride, threshold);
                ++n;
            }
        }
    return n;
}

int dr3o_fast_nonmax_3x3(const uint8_t* score_map, int w, int h, uint8_t* keep)
{
    int x, y, dx, dy, n = 0;
    memset(keep, 0, (size_t)w * h);
    for (y = 0; y < h; ++y)
        for (x = 0; x < w; ++x) {
            const int s = score_map[(size_t)y * w + x];
            int ok = s > 0;
            for (dy = -1; dy <= 1 && ok; ++dy)
                for (dx = -1; dx <= 1 && ok; ++dx) {
                    const int xx = x + dx, yy = y + dy;
                    if ((dx || dy) && xx >= 0 && xx < w && yy >= 0 && yy < h) {
                        const int t = score_map[(size_t)yy * w + xx];
                        if (t > 0 && t >= s) ok = 0;      /* Compare(X, Y) ((X) >= (Y)) on adjacent corners */
                    }
                }
            if (ok) { keep[(size_t)y * w + x] = 1; ++n; }
        }
    return n;
}

int dr3o_fast_detect_grid(const uint8_t* const* levels, const int* strides, int w, int h, int n_levels,
                          int cell_size, int fast_threshold, double detection_threshold, const uint8_t* occupancy,
                          int* cell_x, int* cell_y, float* cell_score, int* cell_level)
{
    const int grid_cols = (int)ceil((double)w / cell_size), grid_rows = (int)ceil((double)h / cell_size);
    const int n_cells = grid_cols * grid_rows;
    int lvl, k, x, y, n = 0;
    for (k = 0; k < n_cells; ++k) { cell_x[k] = 0; cell_y[k] = 0; cell_score[k] = (float)detection_threshold; cell_level[k] = 0; }
    for (lvl = 0; lvl < n_levels; ++lvl) {
        const int lw = w >> lvl, lh = h >> lvl, scale = 1 << lvl;
        uint8_t* sm = (uint8_t*)malloc((size_t)lw * lh);
        uint8_t* keep = (uint8_t*)malloc((size_t)lw * lh);
        if (!sm || !keep) { free(sm); free(keep); return -1; }
        dr3o_fast10_score_map(levels[lvl], lw, lh, strides[lvl], fast_threshold, sm);
        dr3o_fast_nonmax_3x3(sm, lw, lh, keep);
        for (y = 0; y < lh; ++y)                    /* corners come out of the detector in raster order */
            for (x = 0; x < lw; ++x) {
                float score;
                if (!keep[(size_t)y * lw + x]) continue;
                k = ((y * scale) / cell_size) * grid_cols + (x * scale) / cell_size;
                if (occupancy && occupancy[k]) continue;
                score = dr3o_shi_tomasi_score(levels[lvl], lw, lh, strides[lvl], x, y);
                if (score > cell_score[k]) { cell_x[k] = x * scale; cell_y[k] = y * scale; cell_score[k] = score; cell_level[k] = lvl; }
            }
        free(sm); free(keep);
    }
    for (k = 0; k < n_cells; ++k) {
        if ((double)cell_score[k] > detection_threshold) ++n; else cell_level[k] = -1;
    }
    return n;
}

/* src/initialization.cpp:171-249 */
float dr3o_check_fundamental(const float F21[9], const float* pts1, const float* pts2, int n, float sigma, uint8_t* inliers)
{
    const float f11 = F21[0], f12 = F21[1], f13 = F21[2], f21 = F21[3], f22 = F21[4], f23 = F21[5], f31 = F21[6], f32 = F21[7], f33 = F21[8];
    const float th = 3.841, thScore = 5.991;
    const float invSigmaSquare = 1.0 / (sigma * sigma);
    float score = 0;
    int i;
    for (i = 0; i < n; ++i) {
        int bIn = 1;
        const float u1 = pts1[2 * i], v1 = pts1[2 * i + 1], u2 = pts2[2 * i], v2 = pts2[2 * i + 1];
        const float a2 = f11 * u1 + f12 * v1 + f13;
        const float b2 = f21 * u1 + f22 * v1 + f23;
        const float c2 = f31 * u1 + f32 * v1 + f33;
        const float num2 = a2 * u2 + b2 * v2 + c2;
        const float squareDist1 = num2 * num2 / (a2 * a2 + b2 * b2);
        const float chiSquare1 = squareDist1 * invSigmaSquare;
        float a1, b1, c1, num1, squareDist2, chiSquare2;
        if (chiSquare1 > th) bIn = 0; else score += thScore - chiSquare1;
        a1 = f11 * u2 + f21 * v2 + f31;
        b1 = f12 * u2 + f22 * v2 + f32;
        c1 = f13 * u2 + f23 * v2 + f33;
        num1 = a1 * u1 + b1 * v1 + c1;
        squareDist2 = num1 * num1 / (a1 * a1 + b1 * b1);
        chiSquare2 = squareDist2 * invSigmaSquare;
        if (chiSquare2 > th) bIn = 0; else score += thScore - chiSquare2;
        if (inliers) inliers[i] = (uint8_t)bIn;
    }
    return score;
}

/* InitHelper::Normalize (src/initialization.cpp:365-410): means and mean absolute deviations as SEQUENTIAL f32 sums in point
 * order, sX = 1.0 / meanDevX (a double division rounded to float: the same bits as the f32 division), points shifted and scaled,
 * T = [sX 0 -meanX*sX; 0 sY -meanY*sY; 0 0 1].  mask (may be NULL): points with mask[i] == 0 are skipped as if erased from the
 * vector beforehand (the reference erases lost points at :621-635); their output is (0, 0). */
void dr3o_normalize_points(const float* xy, const uint8_t* mask, int n, float* out_xy, float T[9])
{
    float meanX = 0, meanY = 0, meanDevX = 0, meanDevY = 0, sX, sY;
    int i, N = 0;
    for (i = 0; i < n; ++i) {
        if (mask && !mask[i]) continue;
        meanX += xy[2 * i];
        meanY += xy[2 * i + 1];
        ++N;
    }
    meanX = meanX / N;
    meanY = meanY / N;
    for (i = 0; i < n; ++i) {
        if (mask && !mask[i]) { out_xy[2 * i] = 0.f; out_xy[2 * i + 1] = 0.f; continue; }
        out_xy[2 * i] = xy[2 * i] - meanX;
        out_xy[2 * i + 1] = xy[2 * i + 1] - meanY;
        meanDevX += fabsf(out_xy[2 * i]);
        meanDevY += fabsf(out_xy[2 * i + 1]);
    }
    meanDevX = meanDevX / N;
    meanDevY = meanDevY / N;
    sX = (float)(1.0 / meanDevX);
    sY = (float)(1.0 / meanDevY);
    for (i = 0; i < n; ++i) {
        if (mask && !mask[i]) continue;
        out_xy[2 * i] = out_xy[2 * i] * sX;
        out_xy[2 * i + 1] = out_xy[2 * i + 1] * sY;
    }
    T[0] = sX; T[1] = 0.f; T[2] = -meanX * sX;
    T[3] = 0.f; T[4] = sY; T[5] = -meanY * sY;
    T[6] = 0.f; T[7] = 0.f; T[8] = 1.f;
}

/* ------------------------------------------------------------------------------------------ */
/* A-4 / A-5  warpPerspective, remap                                                          */
/* ------------------------------------------------------------------------------------------ */

int dr3o_invert3x3(const double S[9], double T[9])
{
    /* cv::invert, n == 3, CV_64F (matrix_decomp / lapack.cpp closed form) */
    const double d0 = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) +
                      S[2] * (S[3] * S[7] - S[4] * S[6]);
    double d, t[9];
    if (d0 == 0.) { memset(T, 0, 9 * sizeof(double)); return 0; }
    d = 1. / d0;
    t[0] = (S[4] * S[8] - S[5] * S[7]) * d;
    t[1] = (S[2] * S[7] - S[1] * S[8]) * d;
    t[2] = (S[1] * S[5] - S[2] * S[4]) * d;
    t[3] = (S[5] * S[6] - S[3] * S[8]) * d;
    t[4] = (S[0] * S[8] - S[2] * S[6]) * d;
    t[5] = (S[2] * S[3] - S[0] * S[5]) * d;
    t[6] = (S[3] * S[7] - S[4] * S[6]) * d;
    t[7] = (S[1] * S[6] - S[0] * S[7]) * d;
    t[8] = (S[0] * S[4] - S[1] * S[3]) * d;
    memcpy(T, t, sizeof(t));
    return 1;
}

static int type_channels(int type)
{
    switch (type) {
    case DR3O_8UC1: return 1; case DR3O_8UC3: return 3; case DR3O_8UC4: return 4;
    case DR3O_32FC4: return 4; case DR3O_32FC1: return 1; default: return 0;
    }
}
static int type_is_float(int type) { return type == DR3O_32FC4 || type == DR3O_32FC1; }

/* fixed-point bilinear table entry for fraction (fx,fy) in 1/32 px -- initInterTab2D(INTER_LINEAR,
 * fixpt) in imgwarp.cpp.  Entry (0,0) saturates 32768 -> 32767 in `short`; OpenCV's sum-repair loop
 * then adds the missing 1 to w11.  Neither changes any 8-bit result; kept for fidelity. */
static inline void bilinear_w_8u(int fx, int fy, int w[4])
{
    w[0] = (32 - fx) * (32 - fy) * 32; w[1] = fx * (32 - fy) * 32;
    w[2] = (32 - fx) * fy * 32;        w[3] = fx * fy * 32;
    if (fx == 0 && fy == 0) { w[0] = 32767; w[3] = 1; }
}
static inline void bilinear_w_32f(int fx, int fy, float w[4])
{
    const float x1 = fx * (1.f / 32), x0 = 1.f - x1, y1 = fy * (1.f / 32), y0 = 1.f - y1;
    w[0] = y0 * x0; w[1] = y0 * x1; w[2] = y1 * x0; w[3] = y1 * x1;
}

/* remapBilinear core for one destination pixel: (sx,sy) integer source, (fx,fy) 5-bit fraction,
 * BORDER_CONSTANT 0. */
static inline void sample_px(const uint8_t* src, int sw, int sh, int sstride, int type, int cn,
                             int sx, int sy, int fx, int fy, uint8_t* dpx)
{
    const int in00 = (sx >= 0 && sy >= 0 && sx < sw && sy < sh);
    const int in01 = (sx + 1 >= 0 && sy >= 0 && sx + 1 < sw && sy < sh);
    const int in10 = (sx >= 0 && sy + 1 >= 0 && sx < sw && sy + 1 < sh);
    const int in11 = (sx + 1 >= 0 && sy + 1 >= 0 && sx + 1 < sw && sy + 1 < sh);
    int k;
    if (!type_is_float(type)) {
        int w[4];
        bilinear_w_8u(fx, fy, w);
        for (k = 0; k < cn; ++k) {
            const int v0 = in00 ? src[(size_t)sy * sstride + sx * cn + k] : 0;
            const int v1 = in01 ? src[(size_t)sy * sstride + (sx + 1) * cn + k] : 0;
            const int v2 = in10 ? src[(size_t)(sy + 1) * sstride + sx * cn + k] : 0;
            const int v3 = in11 ? src[(size_t)(sy + 1) * sstride + (sx + 1) * cn + k] : 0;
            const int v = (v0 * w[0] + v1 * w[1] + v2 * w[2] + v3 * w[3] + (1 << 14)) >> 15;
            dpx[k] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    } else {
        float w[4];
        float* d = (float*)dpx;
        bilinear_w_32f(fx, fy, w);
        for (k = 0; k < cn; ++k) {
            const float v0 = in00 ? *(const float*)(src + (size_t)sy * sstride + ((size_t)sx * cn + k) * 4) : 0.f;
            const float v1 = in01 ? *(const float*)(src + (size_t)sy * sstride + ((size_t)(sx + 1) * cn + k) * 4) : 0.f;
            const float v2 = in10 ? *(const float*)(src + (size_t)(sy + 1) * sstride + ((size_t)sx * cn + k) * 4) : 0.f;
            const float v3 = in11 ? *(const float*)(src + (size_t)(sy + 1) * sstride + ((size_t)(sx + 1) * cn + k) * 4) : 0.f;
            d[k] = v0 * w[0] + v1 * w[1] + v2 * w[2] + v3 * w[3];
        }
    }
}

int dr3o_warp_perspective(const void* src_, int sw, int sh, int sstride, int type,
                          void* dst_, int dw, int dh, int dstride,
                          const float Mf[9], int flags, int n_threads)
{
    const uint8_t* src = (const uint8_t*)src_;
    uint8_t* dst = (uint8_t*)dst_;
    const int cn = type_channels(type);
    const int esz = cn * (type_is_float(type) ? 4 : 1);
    double M[9];
    int i, y, bh0, bw0;
    if (!src || !dst || cn == 0 || sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0) return -1;
    set_threads(n_threads);
    for (i = 0; i < 9; ++i) M[i] = (double)Mf[i];                 /* M0.convertTo(matM, CV_64F) */
    if (!(flags & DR3O_WARP_INVERSE_MAP)) dr3o_invert3x3(M, M);   /* invert(matM, matM)          */

    /* WarpPerspectiveInvoker block geometry: only the x0 + x1 split influences f64 rounding */
    bh0 = dh < 16 ? dh : 16;
    bw0 = (1024 / bh0) < dw ? (1024 / bh0) : dw;

#pragma omp parallel for schedule(static)
    for (y = 0; y < dh; ++y) {
        int x;
        for (x = 0; x < dw; ++x) {
            const int x0 = (x / bw0) * bw0, x1 = x - x0;
            const double X0 = M[0] * x0 + M[1] * y + M[2];
            const double Y0 = M[3] * x0 + M[4] * y + M[5];
            const double W0 = M[6] * x0 + M[7] * y + M[8];
            double W = W0 + M[6] * x1;
            double fX, fY;
            int X, Y;
            W = W ? 32. / W : 0;
            fX = (X0 + M[0] * x1) * W; fY = (Y0 + M[3] * x1) * W;
            /* std::max((double)INT_MIN, std::min((double)INT_MAX, v)) with the std:: argument order: a NaN becomes INT_MAX */
            fX = (fX < (double)INT_MAX) ? fX : (double)INT_MAX; fX = ((double)INT_MIN < fX) ? fX : (double)INT_MIN;
            fY = (fY < (double)INT_MAX) ? fY : (double)INT_MAX; fY = ((double)INT_MIN < fY) ? fY : (double)INT_MIN;
            X = cv_round_d(fX); Y = cv_round_d(fY);
            sample_px(src, sw, sh, sstride, type, cn, sat_s16(X >> 5), sat_s16(Y >> 5), X & 31, Y & 31,
                      dst + (size_t)y * dstride + (size_t)x * esz);
        }
    }
    return 0;
}

int dr3o_remap_bilinear(const void* src_, int sw, int sh, int sstride, int type,
                        void* dst_, int dw, int dh, int dstride,
                        const float* mapx, const float* mapy, int map_stride,
                        int n_threads)
{
    const uint8_t* src = (const uint8_t*)src_;
    uint8_t* dst = (uint8_t*)dst_;
    const int cn = type_channels(type);
    const int esz = cn * (type_is_float(type) ? 4 : 1);
    int y;
    if (!src || !dst || !mapx || !mapy || cn == 0) return -1;
    set_threads(n_threads);
#pragma omp parallel for schedule(static)
    for (y = 0; y < dh; ++y) {
        int x;
        for (x = 0; x < dw; ++x) {
            const int sx = cv_round_f(mapx[(size_t)y * map_stride + x] * 32.f);
            const int sy = cv_round_f(mapy[(size_t)y * map_stride + x] * 32.f);
            sample_px(src, sw, sh, sstride, type, cn, sat_s16(sx >> 5), sat_s16(sy >> 5), sx & 31, sy & 31,
                      dst + (size_t)y * dstride + (size_t)x * esz);
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* "next" rows: map generation, feather, normalise                                            */
/* ------------------------------------------------------------------------------------------ */

/* src/utils.cpp:125-187.  xf = (float)((j - 0.5*w)/f) (double arithmetic, stored float, :153-156);
 * unqualified sin/cos on float -> ::sin(double) in that TU; xhat = (float)(sin*cos) (:170-172);
 * xhat /= zhat in float (:179-180); u = 0.5*w + xhat*f evaluated by cv::MatExpr as
 * xhat*f + 0.5*w in float (scaleAdd), restated as float multiply-add without contraction. */
void dr3o_spherical_map(int w, int h, float f, float* u, float* v)
{
    int i, j;
    for (i = 0; i < h; ++i) {
        for (j = 0; j < w; ++j) {
            const float xf = (float)(((double)(float)j - 0.5 * w) / (double)f);
            const float yf = (float)(((double)(float)i - 0.5 * h) / (double)f);
            float xhat = (float)(sin((double)xf) * cos((double)yf));
            float yhat = (float)sin((double)yf);
            const float zhat = (float)(cos((double)xf) * cos((double)yf));
            xhat /= zhat; yhat /= zhat;
            u[(size_t)i * w + j] = (float)(0.5 * w) + xhat * f;
            v[(size_t)i * w + j] = (float)(0.5 * h) + yhat * f;
        }
    }
}

/* src/utils.cpp:203-265 */
void dr3o_cylindrical_map(int w, int h, float f, float* u, float* v)
{
    int i, j;
    for (i = 0; i < h; ++i) {
        for (j = 0; j < w; ++j) {
            const float th = (float)(((double)(float)j - 0.5 * w) / (double)f);
            const float hh = (float)(((double)(float)i - 0.5 * h) / (double)f);
            float xhat = (float)sin((double)th);
            float yhat = hh;
            const float zhat = (float)cos((double)th);
            xhat /= zhat; yhat /= zhat;
            u[(size_t)i * w + j] = (float)(0.5 * w) + xhat * f;
            v[(size_t)i * w + j] = (float)(0.5 * h) + yhat * f;
        }
    }
}

/* include/utils.hpp:37-46: val += h accumulation in T */
void dr3o_linspace_f32(float a, float b, int n, float* out)
{
    const float hstep = (b - a) / (float)(n - 1);
    float val = a;
    int i;
    for (i = 0; i < n; ++i) { out[i] = val; val += hstep; }
}

/* src/panorama.cpp:112-138 */
void dr3o_panorama_bbox(const float M[9], int w, int h, int* min_x, int* min_y, int* max_x, int* max_y)
{
    const float cx[4] = {0.f, (float)w, 0.f, (float)w}, cy[4] = {0.f, 0.f, (float)h, (float)h};
    float X[4], Y[4], mnx, mny, mxx, mxy;
    int i;
    for (i = 0; i < 4; ++i) {
        /* (M * corners^T)^T: one gemm, float inputs, double accumulation, one rounding to float */
        float t[3];
        int r;
        for (r = 0; r < 3; ++r)
            t[r] = (float)((double)M[3 * r] * (double)cx[i] + (double)M[3 * r + 1] * (double)cy[i] + (double)M[3 * r + 2] * 1.0);
        X[i] = t[0] / t[2];
        Y[i] = t[1] / t[2];
    }
    mnx = mxx = X[0]; mny = mxy = Y[0];
    for (i = 1; i < 4; ++i) {                      /* std::min_element / max_element: first extreme, '<' */
        if (X[i] < mnx) mnx = X[i];
        if (mxx < X[i]) mxx = X[i];
        if (Y[i] < mny) mny = Y[i];
        if (mxy < Y[i]) mxy = Y[i];
    }
    *min_x = (int)mnx; *min_y = (int)mny; *max_x = (int)mxx; *max_y = (int)mxy;
}

/* src/panorama.cpp:163-177 */
int dr3o_feather_ramp(int min_x, int max_x, int feathering_width, float* ramp)
{
    const int n = max_x - min_x;
    int temp = feathering_width, mid, i;
    if (n - 2 * feathering_width < 0) temp = n / 2 - 1;
    if (temp < 0) return -1;                       /* linspace(size_t(-1)) / vector(-k): the reference throws */
    mid = n - 2 * temp;
    if (mid < 0) return -1;
    dr3o_linspace_f32(0.f, 1.f, temp, ramp);
    for (i = 0; i < mid; ++i) ramp[temp + i] = 1.0f;
    dr3o_linspace_f32(1.f, 0.f, temp, ramp + temp + mid);
    return n;
}

/* src/panorama.cpp:194-212.  Scales BGR by the ramp, sets alpha to the ramp, drops pixels whose
 * scaled BGR is exactly (0,0,0), accumulates the rest into the canvas. */
void dr3o_feather_accumulate(float* canvas, int cw, int ch, int cstride,
                             float* warped, int wstride,
                             int min_x, int max_x, const float* ramp)
{
    int i, j;
    for (i = min_x; i != max_x; ++i) {
        if (i < 0 || i >= cw) continue;   /* reference indexes unclamped (Appendix B-5); OOB skipped here */
        for (j = 0; j < ch; ++j) {
            float* wp = (float*)((uint8_t*)warped + (size_t)j * wstride) + 4 * i;
            float* cp = (float*)((uint8_t*)canvas + (size_t)j * cstride) + 4 * i;
            const float r = ramp[i - min_x];
            wp[0] *= r; wp[1] *= r; wp[2] *= r;
            wp[3] = r;
            if (wp[0] == 0 && wp[1] == 0 && wp[2] == 0) {
                wp[3] = 0.0f;
            } else {
                cp[0] += wp[0]; cp[1] += wp[1]; cp[2] += wp[2]; cp[3] += wp[3];
            }
        }
    }
}

/* src/panorama.cpp:215-229: vec[c] = (int)canvas[c] / alpha  -> int cast first, float divide,
 * then float -> uchar narrowing (cv::Vec3b element assignment = plain C++ conversion). */
void dr3o_normalize_canvas(const float* canvas, int cw, int ch, int cstride,
                           uint8_t* out, int ostride)
{
    int i, j, c;
    for (i = 0; i < ch; ++i) {
        const float* cp = (const float*)((const uint8_t*)canvas + (size_t)i * cstride);
        for (j = 0; j < cw; ++j) {
            const float alpha = cp[4 * j + 3];
            uint8_t* o = out + (size_t)i * ostride + 3 * j;
            o[0] = o[1] = o[2] = 0;
            if (alpha > 0.0f) {
                for (c = 0; c < 3; ++c) {
                    const float q = (float)(int)cp[4 * j + c] / alpha;
                    o[c] = (uint8_t)(int)q;   /* float->uchar: UB when out of range; restated via int */
                }
            }
        }
    }
}

/* src/panorama.cpp:156-212 */
int dr3o_panorama_add(const uint8_t* img, int w, int h, int istride, float* canvas, int cw, int ch, int cstride,
                      const float M[9], int feathering_width, float* scratch, int n_threads)
{
    int min_x, min_y, max_x, max_y, n, i, j;
    float* ramp;
    float* wimg;
    dr3o_panorama_bbox(M, w, h, &min_x, &min_y, &max_x, &max_y);
    n = max_x - min_x;
    if (n < 2) return -1;
    ramp = (float*)malloc(sizeof(float) * (size_t)n);
    wimg = (float*)malloc(sizeof(float) * 4 * (size_t)w * h);
    if (!ramp || !wimg) { free(ramp); free(wimg); return -2; }
    if (dr3o_feather_ramp(min_x, max_x, feathering_width, ramp) != n) { free(ramp); free(wimg); return -1; }
    for (i = 0; i < h; ++i)                        /* :178-189: ones(32FC4), BGR from img.convertTo(CV_32FC3) */
        for (j = 0; j < w; ++j) {
            float* o = wimg + 4 * ((size_t)i * w + j);
            const uint8_t* s = img + (size_t)i * istride + 3 * j;
            o[0] = (float)s[0]; o[1] = (float)s[1]; o[2] = (float)s[2]; o[3] = 1.0f;
        }
    dr3o_warp_perspective(wimg, w, h, w * 16, DR3O_32FC4, scratch, cw, ch, cw * 16, M, 0, n_threads);   /* :192 */
    dr3o_feather_accumulate(canvas, cw, ch, cstride, scratch, cw * 16, min_x, max_x, ramp);              /* :194-212 */
    free(ramp); free(wimg);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Stitch::process after align_pair (src/stitch.cpp:33-80)                                     */
/* ------------------------------------------------------------------------------------------ */
static void gemm33_f32(const float A[9], const float B[9], float C[9])   /* CV_32F gemm: double accumulation, one rounding */
{
    int i, j, k;
    float t[9];
    for (i = 0; i < 3; ++i)
        for (j = 0; j < 3; ++j) {
            double acc = 0.0;
            for (k = 0; k < 3; ++k) acc += (double)A[3 * i + k] * (double)B[3 * k + j];
            t[3 * i + j] = (float)acc;
        }
    memcpy(C, t, sizeof(t));
}

int dr3o_stitch_geometry(const float H[9], int w, int h, float Hinv[9], float T[9], float M_right[9], int* new_w, int* new_h)
{
    /* cv::invert, CV_32F, n == 3: determinant and cofactors in double, result stored as float */
    const double d0 = (double)H[0] * ((double)H[4] * H[8] - (double)H[5] * H[7]) - (double)H[1] * ((double)H[3] * H[8] - (double)H[5] * H[6]) +
                      (double)H[2] * ((double)H[3] * H[7] - (double)H[4] * H[6]);
    double d;
    float inv[9], scale, X[8], Y[8], mnx, mny, mxx, mxy;
    const float cx[4] = {0.f, (float)w, 0.f, (float)w}, cy[4] = {0.f, 0.f, (float)h, (float)h};
    int i;
    if (d0 == 0.0) return -1;
    d = 1.0 / d0;
    inv[0] = (float)(((double)H[4] * H[8] - (double)H[5] * H[7]) * d); inv[1] = (float)(((double)H[2] * H[7] - (double)H[1] * H[8]) * d);
    inv[2] = (float)(((double)H[1] * H[5] - (double)H[2] * H[4]) * d); inv[3] = (float)(((double)H[5] * H[6] - (double)H[3] * H[8]) * d);
    inv[4] = (float)(((double)H[0] * H[8] - (double)H[2] * H[6]) * d); inv[5] = (float)(((double)H[2] * H[3] - (double)H[0] * H[5]) * d);
    inv[6] = (float)(((double)H[3] * H[7] - (double)H[4] * H[6]) * d); inv[7] = (float)(((double)H[1] * H[6] - (double)H[0] * H[7]) * d);
    inv[8] = (float)(((double)H[0] * H[4] - (double)H[1] * H[3]) * d);
    scale = (float)(1.0 / (double)inv[8]);                     /* Hinv * (1.0 / Hinv.at<float>(2,2)): float MatExpr scale */
    for (i = 0; i < 9; ++i) Hinv[i] = inv[i] * scale;
    for (i = 0; i < 4; ++i) {                                   /* (Hinv * points^T)^T, divided by the weight (:43-50) */
        float t[3]; int r;
        for (r = 0; r < 3; ++r)
            t[r] = (float)((double)Hinv[3 * r] * (double)cx[i] + (double)Hinv[3 * r + 1] * (double)cy[i] + (double)Hinv[3 * r + 2] * 1.0);
        X[i] = t[0] / t[2]; Y[i] = t[1] / t[2];
        X[4 + i] = cx[i]; Y[4 + i] = cy[i];                    /* the untransformed corners (:57-60) */
    }
    mnx = mxx = X[0]; mny = mxy = Y[0];
    for (i = 1; i < 8; ++i) {
        if (X[i] < mnx) mnx = X[i];
        if (mxx < X[i]) mxx = X[i];
        if (Y[i] < mny) mny = Y[i];
        if (mxy < Y[i]) mxy = Y[i];
    }
    *new_w = (int)(ceilf(mxx) - floorf(mnx));                  /* int(std::ceil(max_X) - std::floor(min_X)) on floats */
    *new_h = (int)(ceilf(mxy) - floorf(mny));
    T[0] = 1.f; T[1] = 0.f; T[2] = -mnx; T[3] = 0.f; T[4] = 1.f; T[5] = -mny; T[6] = 0.f; T[7] = 0.f; T[8] = 1.f;
    gemm33_f32(T, Hinv, M_right);                              /* translate_top_left_mat * Hinv (:73) */
    return 0;
}

int dr3o_stitch_compose(const uint8_t* left, int lw, int lh, int lstride, const uint8_t* right, int rw, int rh, int rstride,
                        const float T[9], const float M_right[9], int new_w, int new_h, float alpha, float beta, float gamma,
                        uint8_t* out, int ostride, uint8_t* scratch, int n_threads)
{
    uint8_t* wr = scratch;
    uint8_t* wl = scratch + (size_t)new_w * new_h * 3;
    int x, y;
    if (new_w < 1 || new_h < 1) return -1;
    dr3o_warp_perspective(right, rw, rh, rstride, DR3O_8UC3, wr, new_w, new_h, new_w * 3, M_right, 0, n_threads);   /* :73 */
    dr3o_warp_perspective(left, lw, lh, lstride, DR3O_8UC3, wl, new_w, new_h, new_w * 3, T, 0, n_threads);          /* :74 */
    for (y = 0; y < new_h; ++y)                                /* cv::addWeighted(left, alpha, right, beta, gamma) (:80) */
        for (x = 0; x < new_w * 3; ++x) {
            const float t = (float)wl[(size_t)y * new_w * 3 + x] * alpha + (float)wr[(size_t)y * new_w * 3 + x] * beta + gamma;
            const int v = (int)nearbyintf(t);                  /* cvRound: round half to even */
            out[(size_t)y * ostride + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* inverse-compositional sparse image alignment -- EXTENSION, see dr3_oracle.h */
/* ------------------------------------------------------------------------------------------ */
#define AL_LANES 256
#define AL_NSUM 28            /* 21 entries of the upper triangle of H (row-major), 6 of g = sum J r, chi2 */
#define AL_BORDER 3           /* patch half size 2 + 1 */

typedef struct { float Iref[16]; float J[16][6]; int vis; } al_cache;

static inline float al_bilin(const uint8_t* p, int stride, float wTL, float wTR, float wBL, float wBR)
{
    return ((wTL * (float)p[0] + wTR * (float)p[1]) + wBL * (float)p[stride]) + wBR * (float)p[stride + 1];
}

/* pixel position of a camera-frame point on level `scale`; 0 if the 4x4 patch (+1 px for its gradients) is not inside */
static inline int al_project(float X, float Y, float Z, float fx, float fy, float cx, float cy, float scale, int w, int h,
                             int* ui, int* vi, float wt[4], float* xn, float* yn)
{
    float x, y, u, v, fu, fv, sx, sy;
    if (!(Z > 1e-6f)) return 0;
    x = X / Z; y = Y / Z;
    u = (fx * x + cx) * scale; v = (fy * y + cy) * scale;
    if (!(fabsf(u) < 1.0e6f) || !(fabsf(v) < 1.0e6f)) return 0;
    fu = floorf(u); fv = floorf(v);
    *ui = (int)fu; *vi = (int)fv;
    if (*ui - AL_BORDER < 0 || *vi - AL_BORDER < 0 || *ui + AL_BORDER >= w || *vi + AL_BORDER >= h) return 0;
    sx = u - fu; sy = v - fv;
    wt[0] = (1.f - sx) * (1.f - sy); wt[1] = sx * (1.f - sy); wt[2] = (1.f - sx) * sy; wt[3] = sx * sy;
    *xn = x; *yn = y;
    return 1;
}

/* the tracker's wave tree over 64 lane values (lk_kernels.hip wave_sum): upper half-wave onto the lower, upper 16 onto the lower
 * 16, xor butterfly 1, 2, 4, 8 */
static float al_wave_tree(const float* p)
{
    float a[32], c[16], b[16];
    int l, off;
    for (l = 0; l < 32; ++l) a[l] = p[l] + p[l + 32];
    for (l = 0; l < 16; ++l) c[l] = a[l] + a[l + 16];
    for (off = 1; off < 16; off <<= 1) {
        for (l = 0; l < 16; ++l) b[l] = c[l] + c[l ^ off];
        memcpy(c, b, sizeof(c));
    }
    return c[0];
}

/* T <- T * exp(-x), x = (translation part, rotation part); 6-term series in theta^2 (exact rational coefficients) */
static void al_update(double T[12], const double x[6])
{
    const double u0 = -x[0], u1 = -x[1], u2 = -x[2], w0 = -x[3], w1 = -x[4], w2 = -x[5];
    const double th2 = (w0 * w0 + w1 * w1) + w2 * w2;
    const double A = 1.0 - th2 * (1.0 / 6.0) * (1.0 - th2 * (1.0 / 20.0) * (1.0 - th2 * (1.0 / 42.0) * (1.0 - th2 * (1.0 / 72.0) * (1.0 - th2 * (1.0 / 110.0)))));
    const double B = 0.5 * (1.0 - th2 * (1.0 / 12.0) * (1.0 - th2 * (1.0 / 30.0) * (1.0 - th2 * (1.0 / 56.0) * (1.0 - th2 * (1.0 / 90.0) * (1.0 - th2 * (1.0 / 132.0))))));
    const double C = (1.0 / 6.0) * (1.0 - th2 * (1.0 / 20.0) * (1.0 - th2 * (1.0 / 42.0) * (1.0 - th2 * (1.0 / 72.0) * (1.0 - th2 * (1.0 / 110.0) * (1.0 - th2 * (1.0 / 156.0))))));
    /* W = [w]x, W2 = w w^T - th2 I */
    const double W[9] = {0.0, -w2, w1, w2, 0.0, -w0, -w1, w0, 0.0};
    const double W2[9] = {w0 * w0 - th2, w0 * w1, w0 * w2, w1 * w0, w1 * w1 - th2, w1 * w2, w2 * w0, w2 * w1, w2 * w2 - th2};
    double Re[9], V[9], te[3], Rn[9], tn[3];
    int i, j;
    for (i = 0; i < 9; ++i) {
        const double id = (i == 0 || i == 4 || i == 8) ? 1.0 : 0.0;
        Re[i] = (id + A * W[i]) + B * W2[i];
        V[i] = (id + B * W[i]) + C * W2[i];
    }
    for (i = 0; i < 3; ++i) te[i] = (V[3 * i] * u0 + V[3 * i + 1] * u1) + V[3 * i + 2] * u2;
    for (i = 0; i < 3; ++i) {
        for (j = 0; j < 3; ++j) Rn[3 * i + j] = (T[4 * i] * Re[j] + T[4 * i + 1] * Re[3 + j]) + T[4 * i + 2] * Re[6 + j];
        tn[i] = ((T[4 * i] * te[0] + T[4 * i + 1] * te[1]) + T[4 * i + 2] * te[2]) + T[4 * i + 3];
    }
    for (i = 0; i < 3; ++i) { T[4 * i] = Rn[3 * i]; T[4 * i + 1] = Rn[3 * i + 1]; T[4 * i + 2] = Rn[3 * i + 2]; T[4 * i + 3] = tn[i]; }
}

/* H x = b, H symmetric 6x6 given by its upper triangle (row-major, 21 values), LDL^T without pivoting; 0 = not positive definite */
static int al_solve(const float* Hf, const float* g, double x[6])
{
    double H[6][6], L[6][6], D[6], z[6];
    int i, j, k, idx = 0;
    for (i = 0; i < 6; ++i) for (j = i; j < 6; ++j) { H[i][j] = (double)Hf[idx]; H[j][i] = H[i][j]; ++idx; }
    for (j = 0; j < 6; ++j) {
        double d = H[j][j];
        for (k = 0; k < j; ++k) d -= (L[j][k] * L[j][k]) * D[k];
        if (!(d > 0.0)) return 0;
        D[j] = d;
        for (i = j + 1; i < 6; ++i) {
            double s = H[i][j];
            for (k = 0; k < j; ++k) s -= (L[i][k] * L[j][k]) * D[k];
            L[i][j] = s / d;
        }
    }
    for (i = 0; i < 6; ++i) { double s = -(double)g[i]; for (k = 0; k < i; ++k) s -= L[i][k] * z[k]; z[i] = s; }
    for (i = 0; i < 6; ++i) z[i] = z[i] / D[i];
    for (i = 5; i >= 0; --i) { double s = z[i]; for (k = i + 1; k < 6; ++k) s -= L[k][i] * x[k]; x[i] = s; }
    for (i = 0; i < 6; ++i) if (!(fabs(x[i]) < 1.0e300)) return 0;
    return 1;
}

int dr3o_sparse_align(const uint8_t* const* ref_levels, const uint8_t* const* cur_levels, const int* w, const int* h, const int* stride,
                      int n_levels, const float* xyz, int n, const dr3o_align_params* p, double T[12],
                      int* n_meas, float* chi2_out, int* iters)
{
    al_cache* cache;
    int level, done = 0;
    if (!ref_levels || !cur_levels || !w || !h || !stride || !xyz || !p || !T || n < 0) return -1;
    if (p->max_level < p->min_level || p->min_level < 0 || p->max_level >= n_levels || p->n_iter < 0) return -2;
    cache = (al_cache*)malloc(sizeof(al_cache) * (size_t)(n > 0 ? n : 1));
    if (!cache) return -3;
    if (n_meas) *n_meas = 0;
    if (chi2_out) *chi2_out = 0.f;
    for (level = p->max_level; level >= p->min_level; --level) {
        const uint8_t* ref = ref_levels[level];
        const uint8_t* cur = cur_levels[level];
        const int lw = w[level], lh = h[level], ls = stride[level];
        const float scale = 1.0f / (float)(1 << level);
        const float fxl = p->fx * scale, fyl = p->fy * scale;
        float chi2_prev = 0.f;
        double T_old[12];
        int f, it;
        if (iters) iters[level] = 0;
        /* ---- reference patches and their Jacobians, once per level ---------------------------------------- */
        for (f = 0; f < n; ++f) {
            al_cache* c = &cache[f];
            const float X = xyz[3 * f], Y = xyz[3 * f + 1], Z = xyz[3 * f + 2];
            int ui, vi, r, cc, py, px, k;
            float wt[4], x, y, S[6][6], zi, j0[6], j1[6];
            c->vis = al_project(X, Y, Z, p->fx, p->fy, p->cx, p->cy, scale, lw, lh, &ui, &vi, wt, &x, &y);
            if (!c->vis) continue;
            for (r = 0; r < 6; ++r)
                for (cc = 0; cc < 6; ++cc)
                    S[r][cc] = al_bilin(ref + (size_t)(vi + r - 3) * ls + (ui + cc - 3), ls, wt[0], wt[1], wt[2], wt[3]);
            zi = 1.0f / Z;
            j0[0] = -zi; j0[1] = 0.f; j0[2] = x * zi; j0[3] = x * y; j0[4] = -(1.0f + x * x); j0[5] = y;
            j1[0] = 0.f; j1[1] = -zi; j1[2] = y * zi; j1[3] = 1.0f + y * y; j1[4] = -(x * y); j1[5] = -x;
            for (py = 0; py < 4; ++py)
                for (px = 0; px < 4; ++px) {
                    const float dx = 0.5f * (S[py + 1][px + 2] - S[py + 1][px]);
                    const float dy = 0.5f * (S[py + 2][px + 1] - S[py][px + 1]);
                    const float gx = dx * fxl, gy = dy * fyl;
                    c->Iref[4 * py + px] = S[py + 1][px + 1];
                    for (k = 0; k < 6; ++k) c->J[4 * py + px][k] = gx * j0[k] + gy * j1[k];
                }
        }
        /* ---- Gauss-Newton ------------------------------------------------------------------------------------ */
        for (it = 0; it < p->n_iter; ++it) {
            float part[AL_LANES][AL_NSUM];
            int cnt[AL_LANES];
            float tot[AL_NSUM], R32[9], t32[3], chi2m;
            double x[6], mx;
            int lane, wv, s, ntot = 0, i;
            memset(part, 0, sizeof(part)); memset(cnt, 0, sizeof(cnt));
            for (i = 0; i < 3; ++i) { R32[3 * i] = (float)T[4 * i]; R32[3 * i + 1] = (float)T[4 * i + 1]; R32[3 * i + 2] = (float)T[4 * i + 2]; t32[i] = (float)T[4 * i + 3]; }
            for (lane = 0; lane < AL_LANES; ++lane) {
                float* acc = part[lane];
                for (f = lane; f < n; f += AL_LANES) {
                    const al_cache* c = &cache[f];
                    const float X = xyz[3 * f], Y = xyz[3 * f + 1], Z = xyz[3 * f + 2];
                    float Xc, Yc, Zc, wt[4], xn, yn;
                    int ui, vi, py, px, a, b;
                    if (!c->vis) continue;
                    Xc = ((R32[0] * X + R32[1] * Y) + R32[2] * Z) + t32[0];
                    Yc = ((R32[3] * X + R32[4] * Y) + R32[5] * Z) + t32[1];
                    Zc = ((R32[6] * X + R32[7] * Y) + R32[8] * Z) + t32[2];
                    if (!al_project(Xc, Yc, Zc, p->fx, p->fy, p->cx, p->cy, scale, lw, lh, &ui, &vi, wt, &xn, &yn)) continue;
                    for (py = 0; py < 4; ++py)
                        for (px = 0; px < 4; ++px) {
                            const float Ic = al_bilin(cur + (size_t)(vi + py - 2) * ls + (ui + px - 2), ls, wt[0], wt[1], wt[2], wt[3]);
                            const float r = Ic - c->Iref[4 * py + px];
                            const float* J = c->J[4 * py + px];
                            int idx = 0;
                            for (a = 0; a < 6; ++a) for (b = a; b < 6; ++b) { acc[idx] += J[a] * J[b]; ++idx; }
                            for (a = 0; a < 6; ++a) acc[21 + a] += J[a] * r;
                            acc[27] += r * r;
                            ++cnt[lane];
                        }
                }
            }
            for (s = 0; s < AL_NSUM; ++s) {
                float wsum[4], lanes[64];
                for (wv = 0; wv < 4; ++wv) { for (lane = 0; lane < 64; ++lane) lanes[lane] = part[64 * wv + lane][s]; wsum[wv] = al_wave_tree(lanes); }
                tot[s] = ((wsum[0] + wsum[1]) + wsum[2]) + wsum[3];
            }
            for (lane = 0; lane < AL_LANES; ++lane) ntot += cnt[lane];
            if (n_meas) *n_meas = ntot;
            if (ntot == 0) break;
            chi2m = tot[27] / (float)ntot;
            if (chi2_out) *chi2_out = chi2m;
            if (it > 0 && chi2m > chi2_prev) { memcpy(T, T_old, sizeof(T_old)); break; }
            if (!al_solve(tot, tot + 21, x)) break;
            memcpy(T_old, T, sizeof(T_old));
            al_update(T, x);
            chi2_prev = chi2m;
            if (iters) iters[level] = it + 1;
            mx = 0.0;
            for (i = 0; i < 6; ++i) if (fabs(x[i]) > mx) mx = fabs(x[i]);
            if (mx < p->eps) break;
        }
        ++done;
    }
    free(cache);
    return done;
}
