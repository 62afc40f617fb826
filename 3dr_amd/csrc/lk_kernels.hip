// lk_kernels.hip -- pyramidal Lucas-Kanade tracker (gfx950).  Replaces the tracker inside
// cv::calcOpticalFlowPyrLK as called at reference src/initialization.cpp:608-613
// (LKTrackerInvoker).
//
// Mapping: ONE wave64 per (frame pair, point), one wave per workgroup; all pyramid levels run
// coarse -> fine inside the wave, the running estimate never leaves registers.  The window is cut
// into lane-items: item q = ty * rpr + r covers the 4 taps (ty, 4r .. 4r+3); lane = q % 64,
// pass = q / 64 (NPASS passes, template parameter so the patch lives in statically indexed VGPRs).
// Per level the wave
//   1. stages the J search region (window + RM px margin, REFLECT_101 already applied) from
//      global memory into its private LDS slice with aligned dword loads -- ONE global round trip
//      per level instead of one per Gauss-Newton iteration;
//   2. extracts its taps of the template patch (I, Ix, Iy: 14-bit fixed-point bilinear, integer
//      exact) once and keeps them in registers;
//   3. iterates out of LDS: two ds_read2_b32 per lane-item, v_perm_b32 + v_dot2_i32_i16 for the
//      bilinear J taps (the template value rides in the dot's accumulator, so the dot yields the
//      mismatch directly), mismatches packed in int16 pairs and multiplied onto the packed
//      gradients with v_dot2_i32_i16 again: 6.5 VALU per tap;
//   4. reduces the two sums across the wave with v_permlane32_swap + v_permlane16_swap + DPP adds (no
//      LDS), solves the 2x2 system redundantly in every lane and tests the wave-uniform exit conditions.
// Scalar registers are the scarce resource (96 at the 8 waves per SIMD the one-pass kernel is pinned to):
// the 15x15 and 21x21 windows have compile-time instantiations (WIN), and the one-pass kernel drops
// out-of-window taps and idle lanes through zero masks on the packed gradients instead of branches whose
// exec masks would occupy SGPR pairs for the kernel's lifetime.
// If an iterate leaves the staged region (rare: margin RM px per level) the region is re-staged
// around it, so results never depend on the margin.
// Border handling: OpenCV pads every level by winSize (REFLECT_101 for images, zeros for
// derivatives).  Level 0 -- the bulk of the bytes and the graded pyramid kernel's output -- is not
// padded: indices are reflected while staging / extracting.  Levels >= 1 are small and carry the
// border physically (LevelDev::pad_x / pad_y, written by pad_reflect_kernel), because a third of the
// windows cross the image edge there and the reflecting paths cost 15 % of this kernel.
//
// Summation order (defines bit-exactness against oracle order DR3O_ORDER_WAVE64): every term is an
// integer; the taps of one lane are summed exactly in int32 (<= 16 taps of < 2^25.1) and converted
// to f32 once into p[lane] (OpenCV converts every term and rounds at every add: the kernel's sum is
// the more accurate one, tests bound the difference); a[l] = p[l] + p[l+32] (l < 32);
// c[l] = a[l] + a[l+16] (l < 16); an xor butterfly with strides 1,2,4,8 over the 16 values gives the
// total.  wave_sum2 / wave_sum3 produce exactly that tree for two / three sums at once.  No FMA
// contraction (-ffp-contract=off), IEEE divide/sqrt (hipcc default).
#include "dr3hip_internal.h"
#include "wave_tree.h"

namespace dr3 {
namespace {

constexpr int W_BITS = 14;
constexpr int RM = 4;              // search margin (px) of the staged J region

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

// cvFloor as the reference's x86 build computes it for the bounds tests: cvttss2si yields INT_MIN for NaN, so a NaN
// coordinate is "outside" (status 0), where a plain float->int conversion would give 0 and let it through.
// v_med3_f32 returns the smaller bound when an operand is NaN; the conversion then saturates to INT_MIN.
__device__ __forceinline__ int floor_to_int(float floored)
{
    int i = (int)__builtin_amdgcn_fmed3f(floored, -3.0e9f, 3.0e9f);
    // Every caller broadcasts the result with v_readfirstlane.  Without this (empty) barrier the compiler hoists the
    // broadcast above the conversion, converts from the SGPR on the VALU and broadcasts again: one more instruction.
    asm volatile("" : "+v"(i));
    return i;
}

__device__ __forceinline__ int reflect101(int p, int n)
{
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

// ---- fixed-point bilinear blends ------------------------------------------------------------------
// 14-bit weights packed for the dot2 instructions: (w00 | w01 << 16), (w10 | w11 << 16)
struct Weights { uint32_t w01, w23; };

// cvRound(x * 2^14) of the three products, the fourth as the remainder.  Scaling one factor by 2^14 first gives the
// same bits as scaling the product (power of two, no under/overflow here).  Rounding: x + 1.5 * 2^23 rounds to
// nearest-even exactly like cvRound for 0 <= x <= 2^14 and leaves the integer in the low bits of the float's
// pattern (0x4B400000 + n), so the int16 halves are picked straight out of the patterns.
__device__ __forceinline__ Weights make_weights(float a, float b)
{
    const float MAGIC = 12582912.f;             // 1.5 * 2^23, pattern 0x4B400000
    const float na = (1.f - a) * (float)(1 << W_BITS), sa = a * (float)(1 << W_BITS), nb = 1.f - b;
    const uint32_t t00 = __float_as_uint(na * nb + MAGIC);
    const uint32_t t01 = __float_as_uint(sa * nb + MAGIC);
    const uint32_t t10 = __float_as_uint(na * b + MAGIC);
    const uint32_t w11 = (uint32_t)(1 << W_BITS) + 3u * 0x4B400000u - (t00 + t01 + t10);     // may be -1 or -2
    Weights W;
    W.w01 = __builtin_amdgcn_perm(t01, t00, 0x05040100u);
    W.w23 = __builtin_amdgcn_perm(w11, t10, 0x05040100u);
    return W;
}

// Two image rows as aligned 8-byte windows (lo, hi dwords); tap k uses bytes o+k, o+k+1.
struct Rows { uint32_t lo0, hi0, lo1, hi1; };

// v_perm_b32 selector picking bytes (o+k, o+k+1) of {hi,lo} into the two u16 halves (0x0c = zero)
__device__ __forceinline__ uint32_t tap_sel(int o, int k)
{
    return 0x0c000c00u | (uint32_t)(o + k) | ((uint32_t)(o + k + 1) << 16);
}

// (I00*iw00 + I01*iw01 + I10*iw10 + I11*iw11 + acc) >> (W_BITS - 5): two perms + two dot2.
// acc = BILIN_ROUND gives CV_DESCALE(..., W_BITS - 5); acc = BILIN_ROUND - (t << (W_BITS - 5)) gives that minus t
// (the subtrahend is a multiple of the shift's divisor, so floor commutes with it).
constexpr int BILIN_ROUND = 1 << (W_BITS - 5 - 1);
// The saturating form of v_dot2_i32_i16 is the three-operand encoding: the accumulator is a free source instead of
// being tied to the destination (the non-saturating builtin selects v_dot2c and a v_mov per use).  No sum here can
// reach the int32 limits (bilinear: < 2^31 / 64; window sums: <= 16 taps of < 2^25.1), so saturation never acts.
constexpr bool DOT_SAT = true;
__device__ __forceinline__ int bilin_u8(const Rows& r, uint32_t sel, const Weights& W, int acc)
{
    const uint32_t p0 = __builtin_amdgcn_perm(r.hi0, r.lo0, sel);
    const uint32_t p1 = __builtin_amdgcn_perm(r.hi1, r.lo1, sel);
    // signed dot: iw11 = 2^14 - iw00 - iw01 - iw10 is -1 or -2 when both fractions are a few 1e-5 (the three rounded
    // weights then sum to more than 2^14); OpenCV carries it as a negative int, so must the packed 16-bit lane
    int v = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, p0), __builtin_bit_cast(s16x2, W.w01), acc, DOT_SAT);
    v = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, p1), __builtin_bit_cast(s16x2, W.w23), v, DOT_SAT);
    return v >> (W_BITS - 5);
}

// (lo | hi << 16) of two values that fit int16
__device__ __forceinline__ uint32_t pack_s16(int lo, int hi)
{
    return __builtin_amdgcn_perm((uint32_t)hi, (uint32_t)lo, 0x05040100u);
}
__device__ __forceinline__ int dot_s16(uint32_t a, uint32_t b, int acc)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b), acc, DOT_SAT);
}

// CV_DESCALE(d00*iw00 + d01*iw01 + d10*iw10 + d11*iw11, W_BITS) on one int16 channel of the
// interleaved derivative; a0,a1 = dwords (Ix | Iy << 16) of columns k, k+1 of row y, b0,b1 of row y+1
template <uint32_t SEL>
__device__ __forceinline__ int bilin_s16(uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1, const Weights& W)
{
    const uint32_t p0 = __builtin_amdgcn_perm(a1, a0, SEL);
    const uint32_t p1 = __builtin_amdgcn_perm(b1, b0, SEL);
    int v = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, p0), __builtin_bit_cast(s16x2, W.w01), 1 << (W_BITS - 1), DOT_SAT);
    v = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, p1), __builtin_bit_cast(s16x2, W.w23), v, DOT_SAT);
    return v >> W_BITS;
}

// ---- global loads of the template (I) rows ----------------------------------------------------------
// fast: whole window inside the image -> aligned dword pairs, byte offset o = x & 3 (wave-uniform)
__device__ __forceinline__ Rows load_rows_fast(const uint8_t* __restrict__ img, int pitch, int x, int y)
{
    const uint8_t* p = img + (long long)y * pitch + (x & ~3);
    const uint2 a = *reinterpret_cast<const uint2*>(p);
    const uint2 b = *reinterpret_cast<const uint2*>(p + pitch);
    Rows r; r.lo0 = a.x; r.hi0 = a.y; r.lo1 = b.x; r.hi1 = b.y;
    return r;
}
// slow: REFLECT_101 index per byte (window crosses the image border); bytes packed at offset 0
__device__ __forceinline__ Rows load_rows_slow(const uint8_t* __restrict__ img, int pitch, int w, int h, int x, int y)
{
    const uint8_t* p0 = img + (long long)reflect101(y, h) * pitch;
    const uint8_t* p1 = img + (long long)reflect101(y + 1, h) * pitch;
    Rows r; r.lo0 = 0; r.lo1 = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int xr = reflect101(x + k, w);
        r.lo0 |= (uint32_t)p0[xr] << (8 * k);
        r.lo1 |= (uint32_t)p1[xr] << (8 * k);
    }
    const int x4 = reflect101(x + 4, w);
    r.hi0 = p0[x4]; r.hi1 = p1[x4];
    return r;
}

// ---- J search region in LDS --------------------------------------------------------------------------
struct Region { int x0, y0; };       // image coordinates of LDS (row 0, byte 0); x0 is a multiple of 4

// Stage rows [cy-RM, cy-RM+rh) x aligned byte columns [x0, x0+4*rwd) of level L of frame J.
__device__ __forceinline__ Region stage_region(const uint8_t* __restrict__ J, const LevelDev& L, int inx, int iny,
                                               uint32_t* __restrict__ lds, int rwd, int rh, int lpd, int cw_shift, int lane)
{
    Region R;
    R.x0 = (inx - RM) & ~3;
    R.y0 = iny - RM;
    // lanes form a (64 >> cw_shift) x (1 << cw_shift) grid over (rows, dword columns): no division
    const int c = lane & ((1 << cw_shift) - 1);
    const bool padded = L.pad_x > 0;       // the plane carries its REFLECT_101 border physically: plain addressing
    for (int r = lane >> cw_shift; r < rh; r += 64 >> cw_shift) {
        if (c >= rwd) continue;
        const int yy = R.y0 + r;
        const uint8_t* row = J + (long long)(padded ? yy : reflect101(yy, L.h)) * L.pitch;
        const int x = R.x0 + 4 * c;
        uint32_t v;
        if (padded || (x >= 0 && x + 3 < L.w)) {
            v = *reinterpret_cast<const uint32_t*>(row + x);
        } else {
            v = (uint32_t)row[reflect101(x, L.w)] | ((uint32_t)row[reflect101(x + 1, L.w)] << 8) |
                ((uint32_t)row[reflect101(x + 2, L.w)] << 16) | ((uint32_t)row[reflect101(x + 3, L.w)] << 24);
        }
        lds[r * lpd + c] = v;
    }
    __builtin_amdgcn_wave_barrier();      // one wave per workgroup: LDS ops of a wave are ordered
    return R;
}

__device__ __forceinline__ bool region_covers(const Region& R, int inx, int iny, int rpr, int win_h, int rwd, int rh)
{
    const int xb = inx & ~3;
    return xb >= R.x0 && xb + 4 * rpr + 4 <= R.x0 + 4 * rwd && iny >= R.y0 && iny + win_h + 1 <= R.y0 + rh;
}

// base = wave-uniform dword offset of (inx & ~3, iny) in the region; loff = the lane-item's own offset
__device__ __forceinline__ Rows lds_rows(const uint32_t* __restrict__ lds, int lpd, int base, int loff)
{
    const uint32_t* p = lds + base + loff;
    Rows r; r.lo0 = p[0]; r.hi0 = p[1]; r.lo1 = p[lpd]; r.hi1 = p[lpd + 1];
    return r;
}

// WIN > 0: a square WIN x WIN window known at compile time (the sizes the reference and OpenCV's default use); the window
// geometry then folds into immediates instead of living in SGPRs, of which this kernel is short.  WIN = 0: any window.
template <int NPASS, int WIN>
// 15x15-class windows (one pass): 8 waves per SIMD need <= 96 SGPRs; the compiler parks the excess in VGPR lanes
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(NPASS == 1 ? 8 : 4))) void lk_track_kernel(const LKArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];

    const int lane = threadIdx.x;
    const int pt = blockIdx.x, pair = blockIdx.y;
    const long long pidx = (long long)pair * a.n_pts + pt;
    const int ps = a.prev_slot ? a.prev_slot[pair] : a.first_slot + pair;
    const int ns = a.next_slot ? a.next_slot[pair] : a.first_slot + pair + 1;

    const float ppx = a.prev_xy[2 * pidx], ppy = a.prev_xy[2 * pidx + 1];
    float nx = a.next_xy[2 * pidx], ny = a.next_xy[2 * pidx + 1];

    const int win_w = WIN ? WIN : a.win_w, win_h = WIN ? WIN : a.win_h, rpr = WIN ? (WIN + 3) / 4 : a.rpr;
    const int rwd = WIN ? (WIN + 3) / 4 + 3 : a.rwd, rh = WIN ? WIN + 9 : a.rh, lpd = WIN ? (((WIN + 3) / 4 + 3) | 1) : a.lpd;
    const int cw_shift = WIN ? ((WIN + 3) / 4 + 3 <= 8 ? 3 : 4) : a.cw_shift;
    const int n_items = WIN ? ((WIN + 3) / 4) * WIN : a.n_items;
    const float half_x = (win_w - 1) * 0.5f, half_y = (win_h - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (1 << 20);

    // lane-item geometry (fixed for the whole kernel)
    int itx[NPASS], ity[NPASS], loff[NPASS];   // loff: lane's dword offset inside the staged region
    bool ivalid[NPASS];
    uint32_t m01[NPASS], m23[NPASS];           // int16-pair masks of the taps inside the window's last column
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int q = p * 64 + lane;
        ivalid[p] = q < n_items;
        ity[p] = WIN ? q / ((WIN + 3) / 4) : (int)(__umul24((unsigned)q, (unsigned)a.rpr_magic) >> 16);     // q / rpr (host-made reciprocal)
        itx[p] = 4 * (q - ity[p] * rpr);
        loff[p] = ity[p] * lpd + (itx[p] >> 2);
        const int n = win_w - itx[p];          // taps of this item inside the window (>= 1)
        m01[p] = n >= 2 ? 0xffffffffu : 0x0000ffffu;
        m23[p] = n >= 4 ? 0xffffffffu : (n == 3 ? 0x0000ffffu : 0u);
        if (NPASS == 1 && !ivalid[p]) {
            // one-pass kernel: lanes beyond the last item shadow item 0 (same, safe addresses) with all-zero masks, so
            // neither the template nor the iterations need a branch around them
            itx[p] = 0; ity[p] = 0; loff[p] = 0; m01[p] = 0u; m23[p] = 0u;
        }
    }

    int   status = 1;
    float err = 0.f;

    // template patch of this lane: per tap the dot accumulator BILIN_ROUND - (I << 9); Ix, Iy as int16 pairs of taps
    // (0,1) and (2,3), zero on the taps beyond the window's last column
    int Ib[NPASS][4];
    uint32_t Ix[NPASS][2], Iy[NPASS][2];

    for (int level = a.max_level; level >= 0; --level) {
        const LevelDev L = a.lv[level];
        const float scale = __int_as_float((127 - level) << 23);        // (float)(1./(1<<level))
        float px = ppx * scale, py = ppy * scale;
        float cx, cy;
        if (level == a.max_level) {
            if (a.flags & DR3HIP_LK_USE_INITIAL_FLOW) { cx = nx * scale; cy = ny * scale; }
            else { cx = px; cy = py; }
        } else {
            cx = nx * 2.f; cy = ny * 2.f;
        }
        nx = cx; ny = cy;
        int it_count = 0;

        px -= half_x; py -= half_y;
        const int ipx = __builtin_amdgcn_readfirstlane(floor_to_int(floorf(px)));
        const int ipy = __builtin_amdgcn_readfirstlane(floor_to_int(floorf(py)));

        if (ipx < -win_w || ipx >= L.w || ipy < -win_h || ipy >= L.h) {
            if (level == 0) { status = 0; err = 0.f; }
            if (a.iters && lane == 0) a.iters[pidx * (a.max_level + 1) + level] = 0;
            continue;
        }

        // ---- stage the J search region around the starting estimate (overlaps the I loads) ------
        const uint8_t* __restrict__ J = L.img + (long long)ns * L.step;
        Region R;
        {
            const int sx = __builtin_amdgcn_readfirstlane(floor_to_int(floorf(cx - half_x)));
            const int sy = __builtin_amdgcn_readfirstlane(floor_to_int(floorf(cy - half_y)));
            // a start outside the level is rejected by the first iteration's bounds test; clamp the
            // staging centre so the (unused) region stays addressable
            const int cxs = sx < -win_w ? -win_w : (sx >= L.w ? L.w - 1 : sx);
            const int cys = sy < -win_h ? -win_h : (sy >= L.h ? L.h - 1 : sy);
            R = stage_region(J, L, cxs, cys, lds, rwd, rh, lpd, cw_shift, lane);
        }

        // ---- template patch + structure tensor ------------------------------------------------
        float A11, A12, A22;
        {
            const Weights W = make_weights(px - (float)ipx, py - (float)ipy);
            const uint8_t* __restrict__ I = L.img + (long long)ps * L.step;
            const uint8_t* __restrict__ dI = reinterpret_cast<const uint8_t*>(L.der) + (long long)ps * L.dstep;
            // on a padded level every window that passed the bounds test above reads real memory: reflected pixels,
            // zero derivatives -- exactly what the per-tap slow path would synthesise
            const bool inside = L.pad_x > 0 || (ipx >= 0 && ipy >= 0 && ipx + win_w < L.w && ipy + win_h < L.h);
            const int o = inside ? (ipx & 3) : 0;
            const uint32_t sel0 = tap_sel(o, 0), sel1 = tap_sel(o, 1), sel2 = tap_sel(o, 2), sel3 = tap_sel(o, 3);
            int s11 = 0, s12 = 0, s22 = 0;
#pragma unroll
            for (int p = 0; p < NPASS; ++p) {
#pragma unroll
                for (int k = 0; k < 4; ++k) Ib[p][k] = 0;
                Ix[p][0] = Ix[p][1] = Iy[p][0] = Iy[p][1] = 0u;
                if (NPASS == 1 || ivalid[p]) {
                    const int x = ipx + itx[p], y = ipy + ity[p];
                    Rows r;
                    uint32_t d0[5], d1[5];     // packed (Ix | Iy<<16) of rows y, y+1, columns x..x+4
                    if (inside) {
                        r = load_rows_fast(I, L.pitch, x, y);
                        const uint32_t* q0 = reinterpret_cast<const uint32_t*>(dI + (long long)y * L.dpitch) + x;
                        const uint32_t* q1 = reinterpret_cast<const uint32_t*>(dI + (long long)(y + 1) * L.dpitch) + x;
#pragma unroll
                        for (int k = 0; k < 5; ++k) { d0[k] = q0[k]; d1[k] = q1[k]; }
                    } else {
                        r = load_rows_slow(I, L.pitch, L.w, L.h, x, y);
                        const bool v0 = y >= 0 && y < L.h, v1 = y + 1 >= 0 && y + 1 < L.h;
                        const uint32_t* q0 = reinterpret_cast<const uint32_t*>(dI + (long long)(v0 ? y : 0) * L.dpitch);
                        const uint32_t* q1 = reinterpret_cast<const uint32_t*>(dI + (long long)(v1 ? y + 1 : 0) * L.dpitch);
#pragma unroll
                        for (int k = 0; k < 5; ++k) {
                            const int xx = x + k;
                            const bool okx = xx >= 0 && xx < L.w;
                            d0[k] = (okx && v0) ? q0[xx] : 0u;
                            d1[k] = (okx && v1) ? q1[xx] : 0u;
                        }
                    }
                    // One-pass kernel: every tap is computed (the loads cover all four) and the taps beyond the window's
                    // last column get zero gradients through the pair masks, which removes them from every sum; that
                    // frees the four exec masks (SGPR pairs) the branches keep alive.  The multi-pass kernels are short of
                    // VGPRs instead and branch.
                    int gx[4], gy[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        gx[k] = 0; gy[k] = 0;
                        if (NPASS == 1 || itx[p] + k < win_w) {
                            const uint32_t sel = k == 0 ? sel0 : k == 1 ? sel1 : k == 2 ? sel2 : sel3;
                            Ib[p][k] = BILIN_ROUND - (bilin_u8(r, sel, W, BILIN_ROUND) << (W_BITS - 5));
                            gx[k] = bilin_s16<0x05040100u>(d0[k], d0[k + 1], d1[k], d1[k + 1], W);
                            gy[k] = bilin_s16<0x07060302u>(d0[k], d0[k + 1], d1[k], d1[k + 1], W);
                        }
                    }
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const uint32_t mh = NPASS == 1 ? (h ? m23[p] : m01[p]) : 0xffffffffu;
                        const uint32_t x2 = pack_s16(gx[2 * h], gx[2 * h + 1]) & mh, y2 = pack_s16(gy[2 * h], gy[2 * h + 1]) & mh;
                        Ix[p][h] = x2; Iy[p][h] = y2;
                        s11 = dot_s16(x2, x2, s11);
                        s12 = dot_s16(x2, y2, s12);
                        s22 = dot_s16(y2, y2, s22);
                    }
                }
            }
            wave_sum3((float)s11, (float)s12, (float)s22, A11, A12, A22);
            A11 *= FLT_SCALE; A12 *= FLT_SCALE; A22 *= FLT_SCALE;
        }

        float D = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) /
                             (float)(2 * win_w * win_h);
        if (a.flags & DR3HIP_LK_GET_MIN_EIGENVALS) err = minEig;

        if ((double)minEig < a.min_eig || D < 1.1920928955078125e-7f /* FLT_EPSILON */) {
            if (level == 0) status = 0;
            if (a.iters && lane == 0) a.iters[pidx * (a.max_level + 1) + level] = 0;
            continue;
        }
        D = 1.f / D;
        cx -= half_x; cy -= half_y;

        // ---- Gauss-Newton iterations (out of LDS) -----------------------------------------------
        float pdx = 0.f, pdy = 0.f;
        for (int j = 0; j < a.max_count; ++j) {
            // the floors stay floats for the fractions: (float)inx == fcx whenever the bounds test lets the point through
            const float fcx = floorf(cx), fcy = floorf(cy);
            const int inx = __builtin_amdgcn_readfirstlane(floor_to_int(fcx));
            const int iny = __builtin_amdgcn_readfirstlane(floor_to_int(fcy));
            if (inx < -win_w || inx >= L.w || iny < -win_h || iny >= L.h) {
                if (level == 0) status = 0;
                break;
            }
            it_count = j + 1;
            if (!region_covers(R, inx, iny, rpr, win_h, rwd, rh))
                R = stage_region(J, L, inx, iny, lds, rwd, rh, lpd, cw_shift, lane);
            const Weights W = make_weights(cx - fcx, cy - fcy);
            const int o = inx & 3;
            const uint32_t sel0 = tap_sel(o, 0), sel1 = tap_sel(o, 1), sel2 = tap_sel(o, 2), sel3 = tap_sel(o, 3);
            const int base = (iny - R.y0) * lpd + (((inx & ~3) - R.x0) >> 2);
            int s1 = 0, s2 = 0;
#pragma unroll
            for (int p = 0; p < NPASS; ++p) {
                if (NPASS == 1 || ivalid[p]) {
                    const Rows r = lds_rows(lds, lpd, base, loff[p]);
                    // taps beyond the window's last column carry Ix = Iy = 0: whatever mismatch they see adds nothing
                    const uint32_t m01 = pack_s16(bilin_u8(r, sel0, W, Ib[p][0]), bilin_u8(r, sel1, W, Ib[p][1]));
                    const uint32_t m23 = pack_s16(bilin_u8(r, sel2, W, Ib[p][2]), bilin_u8(r, sel3, W, Ib[p][3]));
                    s1 = dot_s16(m01, Ix[p][0], s1); s1 = dot_s16(m23, Ix[p][1], s1);
                    s2 = dot_s16(m01, Iy[p][0], s2); s2 = dot_s16(m23, Iy[p][1], s2);
                }
            }
            float b1, b2;
            wave_sum2((float)s1, (float)s2, b1, b2);
            b1 *= FLT_SCALE; b2 *= FLT_SCALE;

            const float dx = (A12 * b2 - A22 * b1) * D;
            const float dy = (A12 * b1 - A11 * b2) * D;
            cx += dx; cy += dy;
            nx = cx + half_x; ny = cy + half_y;

            const int stop_eps = ((double)dx * (double)dx + (double)dy * (double)dy <= a.epsilon) ? 1 : 0;
            if (__builtin_amdgcn_readfirstlane(stop_eps)) break;
            // std::abs(float) < 0.01 (a double) holds exactly for the floats <= 0.01f, the largest float below 0.01
            const int stop_osc = (j > 0 && fabsf(dx + pdx) <= 0.01f && fabsf(dy + pdy) <= 0.01f) ? 1 : 0;
            if (__builtin_amdgcn_readfirstlane(stop_osc)) {
                nx -= dx * 0.5f; ny -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
        }
        if (a.iters && lane == 0) a.iters[pidx * (a.max_level + 1) + level] = it_count;

        // ---- residual (level 0 only) -------------------------------------------------------------
        if (status && level == 0 && !(a.flags & DR3HIP_LK_GET_MIN_EIGENVALS)) {
            const float fx = nx - half_x, fy = ny - half_y;
            const int inx = __builtin_amdgcn_readfirstlane(floor_to_int(floorf(fx)));
            const int iny = __builtin_amdgcn_readfirstlane(floor_to_int(floorf(fy)));
            if (inx < -win_w || inx >= L.w || iny < -win_h || iny >= L.h) {
                status = 0;
                continue;
            }
            if (!region_covers(R, inx, iny, rpr, win_h, rwd, rh))
                R = stage_region(J, L, inx, iny, lds, rwd, rh, lpd, cw_shift, lane);
            const Weights W = make_weights(fx - (float)inx, fy - (float)iny);
            const int o = inx & 3;
            const uint32_t sel0 = tap_sel(o, 0), sel1 = tap_sel(o, 1), sel2 = tap_sel(o, 2), sel3 = tap_sel(o, 3);
            const int base = (iny - R.y0) * lpd + (((inx & ~3) - R.x0) >> 2);
            float se = 0.f;
#pragma unroll
            for (int p = 0; p < NPASS; ++p) {
                if (p * 64 + lane < n_items) {           // recomputed here: nothing else keeps this mask alive in the one-pass kernel
                    const Rows r = lds_rows(lds, lpd, base, loff[p]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (itx[p] + k < win_w) {
                            const uint32_t sel = k == 0 ? sel0 : k == 1 ? sel1 : k == 2 ? sel2 : sel3;
                            se += fabsf((float)bilin_u8(r, sel, W, Ib[p][k]));      // exact: 16 taps of <= 2^13
                        }
                    }
                }
            }
            float errval, unused;
            wave_sum2(se, se, errval, unused);
            err = errval * 1.f / (float)(32 * win_w * win_h);
        }
    }

    if (lane == 0) {
        a.next_xy[2 * pidx] = nx;
        a.next_xy[2 * pidx + 1] = ny;
        a.status[pidx] = (uint8_t)status;
        a.err[pidx] = err;
    }
}


// =====================================================================================================================
// Several points per wave, for the small windows (every window up to 16x16, the bench's 15x15 among them).
//   PPW = 4: win_w <= 16 and win_h <= 16.  A quarter wave (16 lanes = one DPP row) owns one point; a lane owns one window row
//            of up to 16 taps.
//   PPW = 2: the other windows of at most 32 lane-items of 8 taps (win_w <= 16, ceil(win_w / 8) * win_h <= 32, e.g. 8x32): half
//            a wave (DPP rows 0-1 or 2-3) owns one point; lane-item q = ty * ceil(win_w / 8) + r covers the 8 taps (ty, 8r .. 8r+7).
// Everything that is wave-uniform in the whole-wave kernel above -- window origins, bilinear weights, the 2x2 solve, the exit
// tests, the reductions' tails -- is computed once per wave-instruction for PPW points; the per-tap work per point is unchanged.
// The groups run in lock step: a level / an iteration is executed while any point of the wave needs it (consecutive points of a
// frame take the same number of iterations within a few per cent on the bench's stream), a point that is done sits the rest out
// under an exec mask, and every cross-lane instruction runs in wave-uniform control flow with all 64 lanes enabled
// (v_permlane16_swap and the DPP row operations never leave a group).
// Addressing: every level of a frame batch carries OpenCV's winSize border physically (LevelDev::pad_x / pad_y,
// REFLECT_101 pixels written by pad_reflect_kernel, zero derivatives), so every window that passes the bounds test is
// plain memory: one wave-uniform plane base in SGPRs + an unsigned 32-bit lane offset (global_load ... v, s[base]), no
// reflection code at all.
// Summation order (the checker's kernel order for these windows): the taps of a lane are summed exactly in int32, converted to
// f32 once into p[l]; PPW = 2: c[l] = p[l] + p[l+16] (l < 16); both: xor butterfly 1,2,4,8 over the 16 values.
// =====================================================================================================================

// sums over each group of two / three lane values; every lane receives its own group's totals
template <int PPW>
__device__ __forceinline__ void group_sum2(float a, float b, float& ra, float& rb)
{
    if (PPW == 4) { ra = row_butterfly(a); rb = row_butterfly(b); return; }
    const float v = row_butterfly(fold16(a, b));                 // row totals: a.A  b.A | a.B  b.B
    const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    ra = __uint_as_float(sw[0]);                                 // a.A  a.A | a.B  a.B
    rb = __uint_as_float(sw[1]);                                 // b.A  b.A | b.B  b.B
}
template <int PPW>
__device__ __forceinline__ float group_sum1(float c)
{
    return PPW == 4 ? row_butterfly(c) : row_butterfly(fold16(c, c));
}

// wave-uniform global base + unsigned 32-bit lane offset: selects the saddr form of global_load
typedef const __attribute__((address_space(1))) uint8_t* gbase_t;
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <class T>
__device__ __forceinline__ T ldg(gbase_t base, uint32_t off)
{
    return *reinterpret_cast<const __attribute__((address_space(1))) T*>(base + off);
}
__device__ __forceinline__ gbase_t as_global(const void* p)
{
    return (gbase_t)(const uint8_t*)p;
}

// the two aligned dword pairs tap K of a lane-item works on: taps 4j .. 4j+3 use dwords (j, j+1) of both rows, all with the same
// four byte selectors
template <int ND>
__device__ __forceinline__ Rows rows_of(const uint32_t (&ra)[ND], const uint32_t (&rb)[ND], int j)
{
    Rows o; o.lo0 = ra[j]; o.hi0 = ra[j + 1]; o.lo1 = rb[j]; o.hi1 = rb[j + 1];
    return o;
}

__device__ __forceinline__ bool any_lane(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

#ifndef LKG_WAVES
#define LKG_WAVES 4
#endif
// WIN > 0: square compile-time window (<= 16); WIN = 0: run-time window fitting PPW (see above)
template <int WIN, int PPW>
// (the run-time 16-tap form needs more registers than four waves per SIMD leave it)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu((PPW == 4 && WIN == 0) ? 3 : LKG_WAVES))) void lk_trackg_kernel(const LKArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_all[];
    constexpr int LANES = 64 / PPW;               // lanes per point
    constexpr int NT = PPW == 4 ? 16 : 8;         // taps per lane-item
    constexpr int ND = NT / 4 + 1;                // aligned dwords of an image row a lane-item reads
    constexpr int SR = LANES / 8;                 // rows a group stages per round (8 dword columns)

    const int lane = threadIdx.x, hl = lane & (LANES - 1), grp = lane / LANES;
    const int pair = blockIdx.y;
    const int pt_raw = PPW * blockIdx.x + grp;
    const bool pt_ok = pt_raw < a.n_pts;
    const int pt = pt_ok ? pt_raw : a.n_pts - 1;          // a spare group shadows the last point and stores nothing
    const long long pidx = (long long)pair * a.n_pts + pt;
    const int ps = a.prev_slot ? a.prev_slot[pair] : a.first_slot + pair;
    const int ns = a.next_slot ? a.next_slot[pair] : a.first_slot + pair + 1;

    const float ppx = a.prev_xy[2 * pidx], ppy = a.prev_xy[2 * pidx + 1];
    float nx = a.next_xy[2 * pidx], ny = a.next_xy[2 * pidx + 1];

    const int win_w = WIN ? WIN : a.win_w, win_h = WIN ? WIN : a.win_h;
    const int rpr = PPW == 4 ? 1 : (win_w + 7) >> 3;       // lane-items per window row
    const int rwd = win_w > 8 ? 7 : 5, rh = win_h + 9, lpd = rwd;      // staged J region: dwords per row (odd), rows
    const int rhs = (rh + SR - 1) / SR * SR;               // LDS rows per slice: staging runs in rounds of SR rows
    const int n_items = rpr * win_h;
    const float half_x = (win_w - 1) * 0.5f, half_y = (win_h - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (1 << 20);
    uint32_t* __restrict__ lds = lds_all + grp * (rhs * lpd);

    // lane-item geometry; lanes beyond the last item shadow item 0 with all-zero masks
    const bool ivalid = hl < n_items;
    const int r8 = (PPW == 2 && rpr == 2 && ivalid) ? (hl & 1) : 0;
    const int ity = ivalid ? ((PPW == 2 && rpr == 2) ? hl >> 1 : hl) : 0;
    const int itx = 8 * r8;
    const uint32_t loffB = (uint32_t)(ity * lpd + 2 * r8) * 4u;      // byte offset of the lane-item inside a staged region
    const int ntap = ivalid ? win_w - itx : 0;             // taps of this item inside the window, 0 for idle lanes
    // int16-pair mask of taps (2h, 2h+1): all ones, the low half only, or nothing.  With a compile-time window and one item per
    // row the pattern is a constant per h, gated by one lane mask
    const uint32_t lane_mask = ivalid ? 0xffffffffu : 0u;
    auto pair_mask = [ntap, lane_mask](int h) -> uint32_t {
        if (WIN && PPW == 4) return lane_mask & (WIN >= 2 * h + 2 ? 0xffffffffu : (WIN == 2 * h + 1 ? 0x0000ffffu : 0u));
        return ntap >= 2 * h + 2 ? 0xffffffffu : (ntap == 2 * h + 1 ? 0x0000ffffu : 0u);
    };
    // staging grid of the group: SR rows x 8 dword columns per round
    const int sc = hl & 7, sr = hl >> 3;
    uint32_t* __restrict__ lds_st = lds + sr * lpd + sc;

    int   status = 1;
    float err = 0.f;
    int Ib[NT];
    uint32_t Ix[NT / 2], Iy[NT / 2];

    for (int level = a.max_level; level >= 0; --level) {
        const LevelDev& L = a.lv[level];
        const int Lw = L.w, Lh = L.h, pitch = L.pitch, dpitch = L.dpitch, pad_x = L.pad_x, pad_y = L.pad_y;
        // padded-plane origins of this pair's planes (wave-uniform)
        const long long border = (long long)pad_y * pitch + pad_x;
        const gbase_t Ibase = as_global(L.img + (long long)ps * L.step - border);
        const gbase_t Jbase = as_global(L.img + (long long)ns * L.step - border);
        const gbase_t Dbase = as_global(reinterpret_cast<const uint8_t*>(L.der) + (long long)ps * L.dstep -
                                        ((long long)pad_y * dpitch + 4ll * pad_x));

        const float scale = __int_as_float((127 - level) << 23);
        float px = ppx * scale, py = ppy * scale;
        float cx, cy;
        if (level == a.max_level) {
            if (a.flags & DR3HIP_LK_USE_INITIAL_FLOW) { cx = nx * scale; cy = ny * scale; }
            else { cx = px; cy = py; }
        } else {
            cx = nx * 2.f; cy = ny * 2.f;
        }
        nx = cx; ny = cy;
        int it_count = 0;

        px -= half_x; py -= half_y;
        const float fpx = floorf(px), fpy = floorf(py);
        const int ipx = floor_to_int(fpx), ipy = floor_to_int(fpy);
        // -win <= ip < size  <=>  (unsigned)(ip + win) < (unsigned)(size + win)
        bool act = (unsigned)(ipx + win_w) < (unsigned)(Lw + win_w) && (unsigned)(ipy + win_h) < (unsigned)(Lh + win_h);
        if (!act && level == 0) { status = 0; err = 0.f; }
        if (!any_lane(act)) {
            if (a.iters && hl == 0 && pt_ok) a.iters[pidx * (a.max_level + 1) + level] = 0;
            continue;
        }
        cx -= half_x; cy -= half_y;

        // ---- stage the J search region around the starting estimate: rows [sy-RM, sy-RM+rh) x aligned byte columns
        // [x0, x0 + 4*rwd).  A start outside the level is rejected by the first iteration's bounds test; the staging
        // centre is clamped so that the (unused) region stays inside the padded plane.
        int rx0, ry0;                      // interior coordinates of LDS (row 0, byte 0) of this group's region; rx0 is a multiple of 4
        {
            const int sx = floor_to_int(floorf(cx)), sy = floor_to_int(floorf(cy));
            const int cxs = min(max(sx, -win_w), Lw - 1), cys = min(max(sy, -win_h), Lh - 1);
            rx0 = (cxs - RM) & ~3;
            ry0 = cys - RM;
            uint32_t off = __umul24((unsigned)(ry0 + pad_y + sr), (unsigned)pitch) + (unsigned)(rx0 + pad_x + 4 * sc);
            const uint32_t step = (uint32_t)SR * (unsigned)pitch;
            if (sc < rwd) {
                if (WIN) {
#pragma unroll
                    for (int k = 0; k < (WIN + 9 + SR - 1) / SR; ++k) {
                        if (SR * k + SR - 1 < WIN + 9 || sr + SR * k < rh) lds_st[SR * k * lpd] = ldg<uint32_t>(Jbase, off);
                        off += step;
                    }
                } else {
                    for (int r = sr; r < rh; r += SR) { lds_st[(r - sr) * lpd] = ldg<uint32_t>(Jbase, off); off += step; }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }

        // ---- template patch + structure tensor ----------------------------------------------------
        int s11 = 0, s12 = 0, s22 = 0;
        if (act) {
            const Weights W = make_weights(px - fpx, py - fpy);
            const int xp = ipx + pad_x + itx, yp = ipy + pad_y + ity;
            const uint32_t o = (uint32_t)xp & 3u;
            const uint32_t sel0 = 0x0c010c00u + o * 0x00010001u, sel1 = sel0 + 0x00010001u, sel2 = sel0 + 0x00020002u, sel3 = sel0 + 0x00030003u;
            const uint32_t offI = __umul24((unsigned)yp, (unsigned)pitch) + ((unsigned)xp & ~3u);
            const uint32_t offD = __umul24((unsigned)yp, (unsigned)dpitch) + 4u * (unsigned)xp;
            uint32_t ra[ND], rb[ND];             // image rows y, y+1 as aligned dwords
            uint32_t d0[NT + 1], d1[NT + 1];     // packed (Ix | Iy<<16) of rows y, y+1, columns x .. x+NT
            if (PPW == 4) {
                const u32x4 a4 = ldg<u32x4>(Ibase, offI), b4 = ldg<u32x4>(Ibase, offI + (unsigned)pitch);
                ra[0] = a4.x; ra[1] = a4.y; ra[2] = a4.z; ra[3] = a4.w; ra[ND - 1] = ldg<uint32_t>(Ibase, offI + 16u);
                rb[0] = b4.x; rb[1] = b4.y; rb[2] = b4.z; rb[3] = b4.w; rb[ND - 1] = ldg<uint32_t>(Ibase, offI + (unsigned)pitch + 16u);
            } else {
                const u32x3 a3 = ldg<u32x3>(Ibase, offI), b3 = ldg<u32x3>(Ibase, offI + (unsigned)pitch);
                ra[0] = a3.x; ra[1] = a3.y; ra[2] = a3.z; rb[0] = b3.x; rb[1] = b3.y; rb[2] = b3.z;
            }
#pragma unroll
            for (int q = 0; q < NT / 4; ++q) {
                const u32x4 u = ldg<u32x4>(Dbase, offD + 16u * q), v = ldg<u32x4>(Dbase, offD + (unsigned)dpitch + 16u * q);
                d0[4 * q] = u.x; d0[4 * q + 1] = u.y; d0[4 * q + 2] = u.z; d0[4 * q + 3] = u.w;
                d1[4 * q] = v.x; d1[4 * q + 1] = v.y; d1[4 * q + 2] = v.z; d1[4 * q + 3] = v.w;
            }
            d0[NT] = ldg<uint32_t>(Dbase, offD + 4u * NT);
            d1[NT] = ldg<uint32_t>(Dbase, offD + (unsigned)dpitch + 4u * NT);
            // every tap is computed; taps beyond the window's last column (and idle lanes) get zero gradients through the
            // pair masks, which removes them from every sum
#pragma unroll
            for (int h = 0; h < NT / 2; ++h) {
                int gx[2], gy[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int k = 2 * h + t;
                    const uint32_t sel = (k & 3) == 0 ? sel0 : (k & 3) == 1 ? sel1 : (k & 3) == 2 ? sel2 : sel3;
                    Ib[k] = BILIN_ROUND - (bilin_u8(rows_of<ND>(ra, rb, k >> 2), sel, W, BILIN_ROUND) << (W_BITS - 5));
                    gx[t] = bilin_s16<0x05040100u>(d0[k], d0[k + 1], d1[k], d1[k + 1], W);
                    gy[t] = bilin_s16<0x07060302u>(d0[k], d0[k + 1], d1[k], d1[k + 1], W);
                }
                const uint32_t mkh = pair_mask(h);
                const uint32_t x2 = pack_s16(gx[0], gx[1]) & mkh, y2 = pack_s16(gy[0], gy[1]) & mkh;
                Ix[h] = x2; Iy[h] = y2;
                s11 = dot_s16(x2, x2, s11);
                s12 = dot_s16(x2, y2, s12);
                s22 = dot_s16(y2, y2, s22);
            }
        }
        float A11, A12, A22;
        group_sum2<PPW>((float)s11, (float)s12, A11, A12);
        A22 = group_sum1<PPW>((float)s22);
        A11 *= FLT_SCALE; A12 *= FLT_SCALE; A22 *= FLT_SCALE;

        float D = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) /
                             (float)(2 * win_w * win_h);
        if (act) {
            if (a.flags & DR3HIP_LK_GET_MIN_EIGENVALS) err = minEig;
            if ((double)minEig < a.min_eig || D < 1.1920928955078125e-7f /* FLT_EPSILON */) {
                if (level == 0) status = 0;
                act = false;
            }
        }
        D = 1.f / D;

        // ---- Gauss-Newton iterations (out of LDS), all points of the wave in lock step -------------------------
        float pdx = 0.f, pdy = 0.f;
        for (int j = 0; j < a.max_count; ++j) {
            float fcx = 0.f, fcy = 0.f;
            int inx = 0, iny = 0;
            if (act) {
                fcx = floorf(cx); fcy = floorf(cy);
                inx = floor_to_int(fcx); iny = floor_to_int(fcy);
                if (!((unsigned)(inx + win_w) < (unsigned)(Lw + win_w) && (unsigned)(iny + win_h) < (unsigned)(Lh + win_h))) {
                    if (level == 0) status = 0;
                    act = false;
                }
            }
            if (!any_lane(act)) break;
            int s1 = 0, s2 = 0;
            if (act) {
                it_count = j + 1;
                uint32_t dxb = (uint32_t)((inx & ~3) - rx0), dyr = (uint32_t)(iny - ry0);     // byte column / row inside the region
                if (dxb > 8u || dyr > 8u) {
                    // the iterate left the staged region (margin RM px): stage it again around the iterate
                    rx0 = (inx - RM) & ~3;
                    ry0 = iny - RM;
                    uint32_t off = __umul24((unsigned)(ry0 + pad_y + sr), (unsigned)pitch) + (unsigned)(rx0 + pad_x + 4 * sc);
                    if (sc < rwd)
                        for (int r = sr; r < rh; r += SR) { lds_st[(r - sr) * lpd] = ldg<uint32_t>(Jbase, off); off += (uint32_t)SR * (unsigned)pitch; }
                    __builtin_amdgcn_wave_barrier();
                    dxb = (uint32_t)((inx & ~3) - rx0); dyr = (uint32_t)(iny - ry0);
                }
                const Weights W = make_weights(cx - fcx, cy - fcy);
                const uint32_t o = (uint32_t)inx & 3u;
                const uint32_t sel0 = 0x0c010c00u + o * 0x00010001u, sel1 = sel0 + 0x00010001u, sel2 = sel0 + 0x00020002u, sel3 = sel0 + 0x00030003u;
                const uint32_t* p = reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(lds) + (__umul24(dyr, (unsigned)(4 * lpd)) + dxb + loffB));
                uint32_t ra[ND], rb[ND];
#pragma unroll
                for (int q = 0; q < ND; ++q) { ra[q] = p[q]; rb[q] = p[lpd + q]; }
#pragma unroll
                for (int h = 0; h < NT / 2; ++h) {
                    const Rows r = rows_of<ND>(ra, rb, h >> 1);
                    const uint32_t m = (h & 1) ? pack_s16(bilin_u8(r, sel2, W, Ib[2 * h]), bilin_u8(r, sel3, W, Ib[2 * h + 1]))
                                               : pack_s16(bilin_u8(r, sel0, W, Ib[2 * h]), bilin_u8(r, sel1, W, Ib[2 * h + 1]));
                    s1 = dot_s16(m, Ix[h], s1);
                    s2 = dot_s16(m, Iy[h], s2);
                }
            }
            float b1, b2;
            group_sum2<PPW>((float)s1, (float)s2, b1, b2);
            if (act) {
                b1 *= FLT_SCALE; b2 *= FLT_SCALE;
                const float dx = (A12 * b2 - A22 * b1) * D;
                const float dy = (A12 * b1 - A11 * b2) * D;
                cx += dx; cy += dy;
                nx = cx + half_x; ny = cy + half_y;
                if ((double)dx * (double)dx + (double)dy * (double)dy <= a.epsilon) {
                    act = false;
                } else if (j > 0 && fabsf(dx + pdx) <= 0.01f && fabsf(dy + pdy) <= 0.01f) {
                    // std::abs(float) < 0.01 (a double) holds exactly for the floats <= 0.01f, the largest float below 0.01
                    nx -= dx * 0.5f; ny -= dy * 0.5f;
                    act = false;
                }
                pdx = dx; pdy = dy;
            }
        }
        if (a.iters && hl == 0 && pt_ok) a.iters[pidx * (a.max_level + 1) + level] = it_count;

        // ---- residual (level 0 only) -----------------------------------------------------------------
        if (level == 0 && !(a.flags & DR3HIP_LK_GET_MIN_EIGENVALS)) {
            bool res = status != 0;
            const float fx = nx - half_x, fy = ny - half_y;
            const float ffx = floorf(fx), ffy = floorf(fy);
            const int inx = floor_to_int(ffx), iny = floor_to_int(ffy);
            if (res && !((unsigned)(inx + win_w) < (unsigned)(Lw + win_w) && (unsigned)(iny + win_h) < (unsigned)(Lh + win_h))) { status = 0; res = false; }
            int se = 0;
            if (res) {
                uint32_t dxb = (uint32_t)((inx & ~3) - rx0), dyr = (uint32_t)(iny - ry0);
                if (dxb > 8u || dyr > 8u) {
                    rx0 = (inx - RM) & ~3;
                    ry0 = iny - RM;
                    uint32_t off = __umul24((unsigned)(ry0 + pad_y + sr), (unsigned)pitch) + (unsigned)(rx0 + pad_x + 4 * sc);
                    if (sc < rwd)
                        for (int r = sr; r < rh; r += SR) { lds_st[(r - sr) * lpd] = ldg<uint32_t>(Jbase, off); off += (uint32_t)SR * (unsigned)pitch; }
                    __builtin_amdgcn_wave_barrier();
                    dxb = (uint32_t)((inx & ~3) - rx0); dyr = (uint32_t)(iny - ry0);
                }
                const Weights W = make_weights(fx - ffx, fy - ffy);
                const uint32_t o = (uint32_t)inx & 3u;
                const uint32_t sel0 = 0x0c010c00u + o * 0x00010001u, sel1 = sel0 + 0x00010001u, sel2 = sel0 + 0x00020002u, sel3 = sel0 + 0x00030003u;
                const uint32_t* p = reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(lds) + (__umul24(dyr, (unsigned)(4 * lpd)) + dxb + loffB));
                uint32_t ra[ND], rb[ND];
#pragma unroll
                for (int q = 0; q < ND; ++q) { ra[q] = p[q]; rb[q] = p[lpd + q]; }
                // sum of |mismatch| over the taps inside the window: |d| <= 2^13 packed in int16 pairs, dotted with the pair
                // masks reduced to ones (exact integer sum of at most NT taps)
#pragma unroll
                for (int h = 0; h < NT / 2; ++h) {
                    const Rows r = rows_of<ND>(ra, rb, h >> 1);
                    const int da = bilin_u8(r, (h & 1) ? sel2 : sel0, W, Ib[2 * h]), db = bilin_u8(r, (h & 1) ? sel3 : sel1, W, Ib[2 * h + 1]);
                    se = dot_s16(pack_s16(da < 0 ? -da : da, db < 0 ? -db : db), pair_mask(h) & 0x00010001u, se);
                }
            }
            const float tot = group_sum1<PPW>((float)se);
            if (res) err = tot * 1.f / (float)(32 * win_w * win_h);
        }
    }

    if (hl == 0 && pt_ok) {
        a.next_xy[2 * pidx] = nx;
        a.next_xy[2 * pidx + 1] = ny;
        a.status[pidx] = (uint8_t)status;
        a.err[pidx] = err;
    }
}

}  // namespace

// points per wave by window (the CPU checker's summation-order rule is the same): 4 for windows up to 16 x 16, 2 for the other
// windows of at most 32 lane-items of 8 taps, 1 (the whole-wave kernel) beyond
static inline int lk_points_per_wave(int win_w, int win_h)
{
    if (win_w <= 16 && win_h <= 16) return 4;
    if (win_w <= 16 && ((win_w + 7) / 8) * win_h <= 32) return 2;
    return 1;
}

hipError_t launch_lk_track(const LKArgs& a, int n_pairs, hipStream_t s)
{
    if (n_pairs <= 0 || a.n_pts <= 0) return hipSuccess;
    const int ppw = lk_points_per_wave(a.win_w, a.win_h);
    if (ppw > 1) {
        const int sr = 64 / ppw / 8, rwd = a.win_w > 8 ? 7 : 5, rhs = (a.win_h + 9 + sr - 1) / sr * sr;
        dim3 grid((a.n_pts + ppw - 1) / ppw, n_pairs), block(64);
        const size_t shmem = (size_t)ppw * rhs * rwd * sizeof(uint32_t);
        if (ppw == 4) {
            if (a.win_w == 15 && a.win_h == 15) hipLaunchKernelGGL((lk_trackg_kernel<15, 4>), grid, block, shmem, s, a);
            else hipLaunchKernelGGL((lk_trackg_kernel<0, 4>), grid, block, shmem, s, a);
        } else {
            hipLaunchKernelGGL((lk_trackg_kernel<0, 2>), grid, block, shmem, s, a);
        }
        return hipGetLastError();
    }
    const int npass = (a.n_items + 63) / 64;
    dim3 grid(a.n_pts, n_pairs), block(64);
    const size_t shmem = (size_t)a.rh * a.lpd * sizeof(uint32_t);
    if (a.win_w == 21 && a.win_h == 21) { hipLaunchKernelGGL((lk_track_kernel<2, 21>), grid, block, shmem, s, a); return hipGetLastError(); }
    switch (npass) {
    case 1: hipLaunchKernelGGL((lk_track_kernel<1, 0>), grid, block, shmem, s, a); break;
    case 2: hipLaunchKernelGGL((lk_track_kernel<2, 0>), grid, block, shmem, s, a); break;
    case 3: hipLaunchKernelGGL((lk_track_kernel<3, 0>), grid, block, shmem, s, a); break;
    case 4: hipLaunchKernelGGL((lk_track_kernel<4, 0>), grid, block, shmem, s, a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace dr3
