// warp_kernels.hip -- bilinear backward warp (perspective and map-driven) and the 2x2 box
// half-sampler (gfx950).
//   warp_perspective : cv::warpPerspective at reference src/stitch.cpp:73-74 (8UC3) and
//                      src/panorama.cpp:192 (32FC4)            -- SURVEY Appendix A-4
//   remap            : cv::remap at reference src/utils.cpp:191 -- Appendix A-5
//   halfsample       : utils::reduce_to_half, reference src/utils.cpp:382-419 (SSE2 :324-349)
// All three are gathers/streams: one lane owns 4 adjacent destination pixels (8-bit 1/3-channel
// types, dword stores; a wave is a 64 x 4 pixel tile, a block OpenCV's own 64 x 16 block) or one
// pixel (4-byte and 16-byte pixel types), source taps come through the vector L1/L2.  Coordinates
// follow OpenCV bit for bit: f64 homography evaluation with the WarpPerspectiveInvoker's x0+x1
// block split, round-half-even to 1/32 px (a guarded fast evaluation hands every near-tie to the
// exact sequence), 15-bit integer weights for 8-bit data (evaluated as perm + dot4), f32 table
// weights summed left to right without FMA contraction for f32 data.
#include "dr3hip_internal.h"
#include <algorithm>

namespace dr3 {
namespace {

// Pointers read out of job structs are generic to the compiler and would become flat_load/flat_store;
// every image access goes through explicit global (address space 1) pointers instead.
#define DR3_G __attribute__((address_space(1)))
using gcu8 = const DR3_G uint8_t*;
using gu8  = DR3_G uint8_t*;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
template <class T> __device__ __forceinline__ const DR3_G T* gcast(gcu8 p) { return reinterpret_cast<const DR3_G T*>(p); }
template <class T> __device__ __forceinline__ DR3_G T* gcast(gu8 p) { return reinterpret_cast<DR3_G T*>(p); }

// destination stores of the gather kernels: every output byte is written once and not read again by the launch, so it is
// stored non-temporally (no L2 allocation; 2-4 % on the warps, measured in profiles/r02_warp/README.md)
#ifndef DR3_WARP_PLAIN_STORE
template <class T> __device__ __forceinline__ void st_out(DR3_G T* p, T v) { __builtin_nontemporal_store(v, p); }
#else
template <class T> __device__ __forceinline__ void st_out(DR3_G T* p, T v) { *p = v; }
#endif
// a source that the launch reads exactly once (the canvas of the normalise pass): a non-temporal load, 0.330 -> 0.264 ms there
template <class T> __device__ __forceinline__ T ld_once(const DR3_G T* p) { return __builtin_nontemporal_load(p); }

struct Tap { int sx, sy, fx, fy; };

__device__ __forceinline__ int sat_s16(int v) { return v < -32768 ? -32768 : v > 32767 ? 32767 : v; }

__device__ __forceinline__ Tap tap_from_fixed(int X, int Y)
{
    Tap t; t.sx = sat_s16(X >> 5); t.sy = sat_s16(Y >> 5); t.fx = X & 31; t.fy = Y & 31;
    return t;
}

// std::max((double)INT_MIN, std::min((double)INT_MAX, v)) with the std:: argument order: a NaN v
// becomes INT_MAX, which is also what fmin (IEEE minNum, one v_min_f64) returns.
__device__ __forceinline__ double clamp_int_range(double v)
{
    return __builtin_fmax(-2147483648.0, __builtin_fmin(2147483647.0, v));
}

// WarpPerspectiveInvoker coordinates (SURVEY A-4): the block origin x0 contributes X0/Y0/W0, the
// offset x1 inside the block is added per pixel; every product and sum is a separate f64 rounding.
struct PerspBase { double X0, Y0, W0; };

__device__ __forceinline__ PerspBase persp_base(const double* __restrict__ M, int x0, int y)
{
    PerspBase b;
    b.X0 = M[0] * x0 + M[1] * y + M[2];
    b.Y0 = M[3] * x0 + M[4] * y + M[5];
    b.W0 = M[6] * x0 + M[7] * y + M[8];
    return b;
}
__device__ __forceinline__ Tap persp_tap(const double* __restrict__ M, const PerspBase& b, double x1)
{
    double W = b.W0 + M[6] * x1;
    W = (W != 0.0) ? 32.0 / W : 0.0;
    const double fX = clamp_int_range((b.X0 + M[0] * x1) * W);
    const double fY = clamp_int_range((b.Y0 + M[3] * x1) * W);
    return tap_from_fixed(__double2int_rn(fX), __double2int_rn(fY));
}
__device__ __forceinline__ int block_origin(int x, int bw0, int bw0_shift)
{
    return bw0_shift >= 0 ? ((x >> bw0_shift) << bw0_shift) : (x / bw0) * bw0;
}
// Fast coordinates.  The exact path costs ~55 VALU-cycles-equivalents per pixel (f64 divide expansion,
// clamps, conversions) and makes the 8-bit warp VALU-bound.  The fast path evaluates the same rational
// function with FMAs and v_rcp_f64 + one Newton step (relative error < 2^-45), converts to fixed point
// with 13 extra fraction bits and is only accepted when the value is at least 2 units (2^-13 of 1/32 px)
// away from a rounding tie; otherwise -- ~5e-4 of the pixels -- the exact path recomputes the pixel, so the
// result is bit-identical by construction.  Valid while |src coordinate| < 8192 px decides inside/outside
// the same way.  The job flag `fast` is set by the host (make_job, warp_fast_coords_ok) only if, over the
// whole destination, W keeps its sign and stays away from 0, the numerator operands are < 2^32 * |W| (which
// bounds the fast path's deviation from the exact value by 1/4 unit) and every source coordinate lies
// within +-8000 px, so the fixed-point value fits int32 and the conversion is always in range.
struct PerspFast { double X0s, Y0s, M0s, M3s; };          // numerators pre-scaled by 32 * 2^13

__device__ __forceinline__ bool persp_tap_fast(const double* __restrict__ M, const PerspBase& b, const PerspFast& f,
                                               double x1, Tap& t)
{
    const double W = __builtin_fma(M[6], x1, b.W0);
    double r = __builtin_amdgcn_rcp(W);                 // v_rcp_f64, ~2^-24 relative
    r = __builtin_fma(__builtin_fma(-W, r, 1.0), r, r);
    const int aX = (int)(__builtin_fma(f.M0s, x1, f.X0s) * r) + 4096;    // |.| < 2^31: host check
    const int aY = (int)(__builtin_fma(f.M3s, x1, f.Y0s) * r) + 4096;
    t.sx = aX >> 18; t.fx = (aX >> 13) & 31;
    t.sy = aY >> 18; t.fy = (aY >> 13) & 31;
    const bool near_tie = (((uint32_t)(aX + 2) & 8191u) < 4u) | (((uint32_t)(aY + 2) & 8191u) < 4u);
    return !near_tie;
}

// Four adjacent pixels from ONE reciprocal: X_k = (N_0 + k M0) r_0 / (1 + k q) with q = M6 r_0 exactly, so
// X_k - X_0 = k c / (1 + k q), c = M0 r_0 - q X_0.  The lane evaluates pixel 0 as above (f64, fixed point with 13 extra bits) and
// the three offsets k c (1 - k q) in f32.  Error budget in units of 2^-13 of 1/32 px, bounded by the host for jobs flagged
// fast == 2 (warp_delta_coords_ok: |q| <= 1.2e-4, |dX/dx|, |dY/dx| <= 2): truncation of pixel 0 < 1, rounding of the offset 0.5,
// its f32 evaluation < 0.3, the dropped (k q)^2 term < 0.25 -- under 2.05 in all, so a value at least 3 units away from a
// rounding tie decides exactly like the f64 sequence; the others (1.5e-3 of the pixels) are recomputed by that sequence.
template <bool GENERIC>
__device__ __forceinline__ void persp_taps4_delta(const double* __restrict__ M, int bw0, int bw0_shift, int x, int y, Tap t[4])
{
    const double xd = (double)x, yd = (double)y;
    const double W0 = __builtin_fma(M[6], xd, __builtin_fma(M[7], yd, M[8]));
    const double NX = __builtin_fma(M[0], xd, __builtin_fma(M[1], yd, M[2]));
    const double NY = __builtin_fma(M[3], xd, __builtin_fma(M[4], yd, M[5]));
    double r = __builtin_amdgcn_rcp(W0);
    r = __builtin_fma(__builtin_fma(-W0, r, 1.0), r, r);
    const double rs = r * 262144.0;                       // 32 * 2^13
    const double X0 = NX * rs, Y0 = NY * rs;
    const double q = M[6] * r;
    const float cx = (float)__builtin_fma(-q, X0, M[0] * rs), cy = (float)__builtin_fma(-q, Y0, M[3] * rs), qf = (float)q;
    const float cx2 = cx * qf, cy2 = cy * qf;
    const int aX0 = (int)X0 + 4096, aY0 = (int)Y0 + 4096;  // |.| < 2^31: host check
    bool exact[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int aX = aX0, aY = aY0;
        if (k) {
            const float fk = (float)k;
            aX += __float2int_rn((cx - fk * cx2) * fk);
            aY += __float2int_rn((cy - fk * cy2) * fk);
        }
        t[k].sx = aX >> 18; t[k].fx = (aX >> 13) & 31;
        t[k].sy = aY >> 18; t[k].fy = (aY >> 13) & 31;
        exact[k] = (((uint32_t)(aX + 3) & 8191u) < 6u) | (((uint32_t)(aY + 3) & 8191u) < 6u);
    }
    if (exact[0] | exact[1] | exact[2] | exact[3]) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (exact[k]) {
                const int x0 = GENERIC ? block_origin(x + k, bw0, bw0_shift) : (((x + k) >> bw0_shift) << bw0_shift);
                t[k] = persp_tap(M, persp_base(M, x0, y), (double)(x + k - x0));
            }
    }
}

// NPX adjacent destination pixels x..x+NPX-1 of row y.  They share the block origin whenever the
// block width is a multiple of NPX (always, except for destinations shorter than 16 rows).
template <int NPX, bool GENERIC>
__device__ __forceinline__ void persp_taps(const double* __restrict__ M, int bw0, int bw0_shift, int fast, int x, int y, Tap t[NPX])
{
#ifndef DR3_WARP_NO_DELTA
    if (NPX == 4 && fast >= 2) { persp_taps4_delta<GENERIC>(M, bw0, bw0_shift, x, y, t); return; }
#endif
    // !GENERIC: every job of the launch has a power-of-two block width >= 4 (or one block per row), so
    // the NPX pixels of a lane share the origin and no integer division is compiled in
    const int x0 = GENERIC ? block_origin(x, bw0, bw0_shift) : ((x >> bw0_shift) << bw0_shift);
    if (!GENERIC || NPX == 1 || block_origin(x + NPX - 1, bw0, bw0_shift) == x0) {
        const PerspBase b = persp_base(M, x0, y);
        const double x1 = (double)(x - x0);
#ifndef DR3_WARP_EXACT_ONLY
        if (fast) {
            const double S = 262144.0;                    // 32 * 2^13
            PerspFast f; f.X0s = b.X0 * S; f.Y0s = b.Y0 * S; f.M0s = M[0] * S; f.M3s = M[3] * S;
#pragma unroll
            for (int i = 0; i < NPX; ++i)
                if (!persp_tap_fast(M, b, f, x1 + (double)i, t[i])) t[i] = persp_tap(M, b, x1 + (double)i);
            return;
        }
#endif
#pragma unroll
        for (int i = 0; i < NPX; ++i) t[i] = persp_tap(M, b, x1 + (double)i);
    } else {
#pragma unroll
        for (int i = 0; i < NPX; ++i) {
            const int xi0 = block_origin(x + i, bw0, bw0_shift);
            t[i] = persp_tap(M, persp_base(M, xi0, y), (double)(x + i - xi0));
        }
    }
}

__device__ __forceinline__ void weights_32f(int fx, int fy, float w[4])
{
    const float x1 = (float)fx * (1.f / 32), x0 = 1.f - x1, y1 = (float)fy * (1.f / 32), y0 = 1.f - y1;
    w[0] = y0 * x0; w[1] = y0 * x1; w[2] = y1 * x0; w[3] = y1 * x1;
}

// Wave tile: LX lanes along x (each PXT pixels wide) by LY = 64 / LX rows; a block stacks 4 waves in y.
#ifndef DR3_WT4_LOG
#define DR3_WT4_LOG 4      // 4-pixel lanes: 64 x 4 pixel wave tile, 64 x 16 block (OpenCV's own block shape)
#endif
#ifndef DR3_WT1_LOG
#define DR3_WT1_LOG 6      // 1-pixel lanes
#endif
template <int PXT> struct WaveTile;
template <> struct WaveTile<4> { static constexpr int LOG_LX = DR3_WT4_LOG, LX = 1 << LOG_LX, LY = 64 >> LOG_LX; };
template <> struct WaveTile<1> { static constexpr int LOG_LX = DR3_WT1_LOG, LX = 1 << LOG_LX, LY = 64 >> LOG_LX; };

template <int TYPE> struct PixelTraits;
template <> struct PixelTraits<DR3HIP_8UC1>  { static constexpr int CN = 1, PXT = 4; };
template <> struct PixelTraits<DR3HIP_8UC3>  { static constexpr int CN = 3, PXT = 4; };
template <> struct PixelTraits<DR3HIP_8UC4>  { static constexpr int CN = 4, PXT = 1; };
template <> struct PixelTraits<DR3HIP_32FC1> { static constexpr int CN = 1, PXT = 1; };
template <> struct PixelTraits<DR3HIP_32FC4> { static constexpr int CN = 4, PXT = 1; };

// ---- 8-bit bilinear blend ------------------------------------------------------------------------
// OpenCV: (v0 w0 + v1 w1 + v2 w2 + v3 w3 + 2^14) >> 15 with the 15-bit table w = 32 * a,
// a = {(32-fx)(32-fy), fx(32-fy), (32-fx)fy, fx fy}; the (0,0) entry is (32767,0,0,1) after the table's
// saturation + sum repair and still yields v0.  32*S + 2^14 = 32*(S + 2^9), so the result is
// (S + 512) >> 10 with S = (32-fy) * h0 + fy * h1, h = (32-fx) * v_left + fx * v_right  (integers
// distribute exactly).  h is one v_dot4_u32_u8 on bytes gathered by v_perm_b32; the vertical
// weights are pre-multiplied by 64 so that the result is byte 2 of the accumulator (no shift; h can be
// 32 * 255 + 16 and the weights 2048, both inside the 24-bit multiplier).
struct Blend8 { uint32_t wlo, whi, wy0, wy1; };   // (32-fx | fx<<8), the same <<16, 64*(32-fy), 64*fy

__device__ __forceinline__ Blend8 blend8_weights(int fx, int fy)
{
    Blend8 b;
    b.wlo = (uint32_t)(32 - fx) | ((uint32_t)fx << 8);
    b.whi = b.wlo << 16;
    b.wy0 = (uint32_t)(32 - fy) << 6;
    b.wy1 = (uint32_t)fy << 6;
    return b;
}
// The rounding constant rides in the dot4 accumulators: wy0 + wy1 = 64 * 32 always, so adding 16 to both
// horizontal sums adds 16 * 2048 = 2^15 to the vertical sum and the vertical stage is v_mul_u32_u24 +
// v_mad_u32_u24.  (No inline asm here: gfx950 needs wait states between a DOT write and a dependent VALU
// read, which the compiler only inserts for instructions it can see.)
#define DR3_HROUND 16u
__device__ __forceinline__ uint32_t blend8_vert(uint32_t h0, uint32_t h1, const Blend8& b)
{
    return __umul24(h0, b.wy0) + __umul24(h1, b.wy1);
}
// p = [top-left, top-right, bottom-left, bottom-right] bytes; returns the accumulator, result in byte 2
__device__ __forceinline__ uint32_t blend8_quad(uint32_t p, const Blend8& b)
{
    return blend8_vert(__builtin_amdgcn_udot4(p, b.wlo, DR3_HROUND, false), __builtin_amdgcn_udot4(p, b.whi, DR3_HROUND, false), b);
}
// p = [left, right, left', right'] bytes of ONE row for two channels; returns both horizontal sums
__device__ __forceinline__ void blend8_row2(uint32_t p, const Blend8& b, uint32_t& h, uint32_t& hp)
{
    h  = __builtin_amdgcn_udot4(p, b.wlo, DR3_HROUND, false);
    hp = __builtin_amdgcn_udot4(p, b.whi, DR3_HROUND, false);
}
__device__ __forceinline__ uint32_t byte2(uint32_t acc) { return (acc >> 16) & 0xffu; }

// per-tap checked byte fetch (border 0)
__device__ __forceinline__ uint32_t quad_checked(gcu8 __restrict__ src, int sw, int sh, int sstride,
                                                 int sx, int sy, int cn, int c)
{
    const bool x0ok = (unsigned)sx < (unsigned)sw, x1ok = (unsigned)(sx + 1) < (unsigned)sw;
    const bool y0ok = (unsigned)sy < (unsigned)sh, y1ok = (unsigned)(sy + 1) < (unsigned)sh;
    gcu8 r0 = src + (long long)sy * sstride + (long long)sx * cn + c;
    gcu8 r1 = r0 + sstride;
    const uint32_t v0 = (x0ok && y0ok) ? r0[0] : 0u, v1 = (x1ok && y0ok) ? r0[cn] : 0u;
    const uint32_t v2 = (x0ok && y1ok) ? r1[0] : 0u, v3 = (x1ok && y1ok) ? r1[cn] : 0u;
    return v0 | (v1 << 8) | (v2 << 16) | (v3 << 24);
}

// 8UC3 interior fast path, split in two so that a lane can have the loads of all its pixels in flight
// before the first blend: pixels sx..sx+3 exist on rows sy, sy+1, so three aligned dwords per row
// (bytes 3sx&~3 .. +11) stay inside the row.
struct Rows3 { uint32_t a0, a1, a2, b0, b1, b2, o; };

__device__ __forceinline__ bool interior_8uc3(const Tap t, int sw, int sh)
{
    return (unsigned)t.sx < (unsigned)(sw - 3) && (unsigned)t.sy < (unsigned)(sh - 1);
}
__device__ __forceinline__ Rows3 load_8uc3(gcu8 __restrict__ src, int sstride, const Tap t)
{
    // 32-bit offsets from the uniform base (src_aligned guarantees stride < 2^24 and image bytes < 2^31)
    const uint32_t bo = 3u * (uint32_t)t.sx;
    const uint32_t off = __umul24((uint32_t)t.sy, (uint32_t)sstride) + (bo & ~3u);
    const DR3_G uint32_t* q0 = gcast<uint32_t>(src + off);
    const DR3_G uint32_t* q1 = gcast<uint32_t>(src + (off + (uint32_t)sstride));
    Rows3 r;
    r.a0 = q0[0]; r.a1 = q0[1]; r.a2 = q0[2];
    r.b0 = q1[0]; r.b1 = q1[1]; r.b2 = q1[2];
    r.o = bo & 3u;
    return r;
}
__device__ __forceinline__ void blend_8uc3(const Rows3& r, const Tap t, uint32_t& aB, uint32_t& aG, uint32_t& aR)
{
    const Blend8 b = blend8_weights(t.fx, t.fy);
    // t0 = [B0 G0 R0 B1], t1 = [G1 R1 . .] of the top row; u0, u1 of the bottom row
    const uint32_t t0 = __builtin_amdgcn_alignbyte(r.a1, r.a0, r.o), t1 = __builtin_amdgcn_alignbyte(r.a2, r.a1, r.o);
    const uint32_t u0 = __builtin_amdgcn_alignbyte(r.b1, r.b0, r.o), u1 = __builtin_amdgcn_alignbyte(r.b2, r.b1, r.o);
    aB = blend8_quad(__builtin_amdgcn_perm(u0, t0, 0x07040300u), b);           // [B0 B1 B0' B1']
    uint32_t hG0, hR0, hG1, hR1;
    blend8_row2(__builtin_amdgcn_perm(t1, t0, 0x05020401u), b, hG0, hR0);      // [G0 G1 R0 R1]
    blend8_row2(__builtin_amdgcn_perm(u1, u0, 0x05020401u), b, hG1, hR1);
    aG = blend8_vert(hG0, hG1, b);
    aR = blend8_vert(hR0, hR1, b);
}

// One 8UC3 destination pixel anywhere -> accumulators of B, G, R (result = byte 2 of each).
__device__ __forceinline__ void pixel_8uc3(gcu8 __restrict__ src, int sw, int sh, int sstride, bool src_aligned,
                                           const Tap t, uint32_t& aB, uint32_t& aG, uint32_t& aR)
{
    if (src_aligned && interior_8uc3(t, sw, sh)) { blend_8uc3(load_8uc3(src, sstride, t), t, aB, aG, aR); return; }
    if (t.sx >= sw || t.sx + 1 < 0 || t.sy >= sh || t.sy + 1 < 0) { aB = aG = aR = 0u; return; }
    const Blend8 b = blend8_weights(t.fx, t.fy);
    aB = blend8_quad(quad_checked(src, sw, sh, sstride, t.sx, t.sy, 3, 0), b);
    aG = blend8_quad(quad_checked(src, sw, sh, sstride, t.sx, t.sy, 3, 1), b);
    aR = blend8_quad(quad_checked(src, sw, sh, sstride, t.sx, t.sy, 3, 2), b);
}

__device__ __forceinline__ uint32_t pixel_8uc1(gcu8 __restrict__ src, int sw, int sh, int sstride, const Tap t)
{
    if (t.sx >= sw || t.sx + 1 < 0 || t.sy >= sh || t.sy + 1 < 0) return 0u;
    return blend8_quad(quad_checked(src, sw, sh, sstride, t.sx, t.sy, 1, 0), blend8_weights(t.fx, t.fy));
}

// 4 adjacent destination pixels of an 8UC1 / 8UC3 row, gathered and stored (dword stores when the
// row is 4-byte aligned and all 4 pixels exist).
template <int TYPE>
__device__ __forceinline__ void quad_8u(gcu8 __restrict__ src, int sw, int sh, int sstride,
                                        gu8 __restrict__ dst, int dw, int dstride, int x, int y, const Tap t[4])
{
    constexpr int CN = PixelTraits<TYPE>::CN;
    // dword-aligned rows and 24/31-bit offset arithmetic: the precondition of the interior fast path
    const bool src_aligned = ((reinterpret_cast<uintptr_t>(src) | (uintptr_t)sstride) & 3) == 0 &&
                             sstride < (1 << 24) && sh < (1 << 24) && (long long)sstride * sh < (1LL << 31);
    const bool dst_aligned = ((reinterpret_cast<uintptr_t>(dst) | (uintptr_t)dstride) & 3) == 0;
    gu8 row = dst + (long long)y * dstride + (long long)x * CN;
    const bool full = x + 3 < dw;
    if (CN == 3) {
        uint32_t aB[4], aG[4], aR[4];
        if (full && src_aligned && interior_8uc3(t[0], sw, sh) && interior_8uc3(t[1], sw, sh) &&
            interior_8uc3(t[2], sw, sh) && interior_8uc3(t[3], sw, sh)) {
            // the common case: all 24 dword loads of the lane are issued before the first blend
            Rows3 r[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) r[i] = load_8uc3(src, sstride, t[i]);
#pragma unroll
            for (int i = 0; i < 4; ++i) blend_8uc3(r[i], t[i], aB[i], aG[i], aR[i]);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                aB[i] = aG[i] = aR[i] = 0u;
                if (full || x + i < dw) pixel_8uc3(src, sw, sh, sstride, src_aligned, t[i], aB[i], aG[i], aR[i]);
            }
        }
        if (full && dst_aligned) {
            // byte 2 of each accumulator -> 12 interleaved bytes
            const uint32_t bg0 = __builtin_amdgcn_perm(aG[0], aB[0], 0x0c0c0602u), rb0 = __builtin_amdgcn_perm(aB[1], aR[0], 0x0c0c0602u);
            const uint32_t gr1 = __builtin_amdgcn_perm(aR[1], aG[1], 0x0c0c0602u), bg2 = __builtin_amdgcn_perm(aG[2], aB[2], 0x0c0c0602u);
            const uint32_t rb2 = __builtin_amdgcn_perm(aB[3], aR[2], 0x0c0c0602u), gr3 = __builtin_amdgcn_perm(aR[3], aG[3], 0x0c0c0602u);
            DR3_G uint32_t* o = gcast<uint32_t>(row);
            st_out(o, bg0 | (rb0 << 16)); st_out(o + 1, gr1 | (bg2 << 16)); st_out(o + 2, rb2 | (gr3 << 16));
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (x + i < dw) { row[3 * i] = (uint8_t)byte2(aB[i]); row[3 * i + 1] = (uint8_t)byte2(aG[i]); row[3 * i + 2] = (uint8_t)byte2(aR[i]); }
        }
    } else {
        uint32_t a[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = (full || x + i < dw) ? pixel_8uc1(src, sw, sh, sstride, t[i]) : 0u;
        if (full && dst_aligned) {
            const uint32_t lo = __builtin_amdgcn_perm(a[1], a[0], 0x0c0c0602u), hi = __builtin_amdgcn_perm(a[3], a[2], 0x0c0c0602u);
            st_out(gcast<uint32_t>(row), lo | (hi << 16));
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) if (x + i < dw) row[i] = (uint8_t)byte2(a[i]);
        }
    }
}

// ---- one pixel per lane: 8UC4, 32FC1, 32FC4 ------------------------------------------------------
template <int TYPE>
__device__ __forceinline__ void single_px(gcu8 __restrict__ src, int sw, int sh, int sstride,
                                          gu8 __restrict__ dst, int dstride, int x, int y, const Tap t)
{
    const int sx = t.sx, sy = t.sy;
    gu8 row = dst + (long long)y * dstride;
    const bool outside = sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0;      // all four taps: border 0
    const bool x0ok = sx >= 0 && !outside, x1ok = sx + 1 < sw && !outside, y0ok = sy >= 0, y1ok = sy + 1 < sh;
    gcu8 r0 = src + (long long)sy * sstride;
    gcu8 r1 = r0 + sstride;
    if (TYPE == DR3HIP_32FC4) {
        float w[4]; weights_32f(t.fx, t.fy, w);
        f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0, v2 = v0, v3 = v0;
        if (x0ok && y0ok) v0 = gcast<f32x4>(r0)[sx];
        if (x1ok && y0ok) v1 = gcast<f32x4>(r0)[sx + 1];
        if (x0ok && y1ok) v2 = gcast<f32x4>(r1)[sx];
        if (x1ok && y1ok) v3 = gcast<f32x4>(r1)[sx + 1];
        f32x4 o;
        o.x = v0.x * w[0] + v1.x * w[1] + v2.x * w[2] + v3.x * w[3];
        o.y = v0.y * w[0] + v1.y * w[1] + v2.y * w[2] + v3.y * w[3];
        o.z = v0.z * w[0] + v1.z * w[1] + v2.z * w[2] + v3.z * w[3];
        o.w = v0.w * w[0] + v1.w * w[1] + v2.w * w[2] + v3.w * w[3];
        st_out(gcast<f32x4>(row) + x, o);
    } else if (TYPE == DR3HIP_32FC1) {
        float w[4]; weights_32f(t.fx, t.fy, w);
        const float v0 = (x0ok && y0ok) ? gcast<float>(r0)[sx] : 0.f;
        const float v1 = (x1ok && y0ok) ? gcast<float>(r0)[sx + 1] : 0.f;
        const float v2 = (x0ok && y1ok) ? gcast<float>(r1)[sx] : 0.f;
        const float v3 = (x1ok && y1ok) ? gcast<float>(r1)[sx + 1] : 0.f;
        gcast<float>(row)[x] = v0 * w[0] + v1 * w[1] + v2 * w[2] + v3 * w[3];
    } else {   // 8UC4
        const Blend8 b = blend8_weights(t.fx, t.fy);
        const uint32_t v0 = (x0ok && y0ok) ? gcast<uint32_t>(r0)[sx] : 0u;
        const uint32_t v1 = (x1ok && y0ok) ? gcast<uint32_t>(r0)[sx + 1] : 0u;
        const uint32_t v2 = (x0ok && y1ok) ? gcast<uint32_t>(r1)[sx] : 0u;
        const uint32_t v3 = (x1ok && y1ok) ? gcast<uint32_t>(r1)[sx + 1] : 0u;
        // channel pairs (0,1) and (2,3) of the top and the bottom row
        uint32_t h00, h01, h02, h03, h10, h11, h12, h13;
        blend8_row2(__builtin_amdgcn_perm(v1, v0, 0x05010400u), b, h00, h01);
        blend8_row2(__builtin_amdgcn_perm(v1, v0, 0x07030602u), b, h02, h03);
        blend8_row2(__builtin_amdgcn_perm(v3, v2, 0x05010400u), b, h10, h11);
        blend8_row2(__builtin_amdgcn_perm(v3, v2, 0x07030602u), b, h12, h13);
        const uint32_t c01 = __builtin_amdgcn_perm(blend8_vert(h01, h11, b), blend8_vert(h00, h10, b), 0x0c0c0602u);
        const uint32_t c23 = __builtin_amdgcn_perm(blend8_vert(h03, h13, b), blend8_vert(h02, h12, b), 0x0c0c0602u);
        st_out(gcast<uint32_t>(row) + x, c01 | (c23 << 16));
    }
}

template <int TYPE, bool GENERIC>
__global__ __launch_bounds__(256) void warp_perspective_kernel(const WarpJobDev* __restrict__ jobs)
{
    using P = PixelTraits<TYPE>;
    const WarpJobDev& jb = jobs[blockIdx.z];
    // a wave is a 2-D tile of the destination (WaveTile<PXT>): the source lines its lanes gather from are
    // then shared inside the wave and with the 3 waves below it instead of being spread over 256 columns
    using WT = WaveTile<P::PXT>;
    const int x = (blockIdx.x * WT::LX + (threadIdx.x & (WT::LX - 1))) * P::PXT;
    const int dw = jb.dw, dh = jb.dh;
    if (x >= dw) return;
    const int bw0 = jb.bw0, bw0_shift = jb.bw0_shift;
    double M[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) M[i] = jb.M[i];

    {
        const int y = (blockIdx.y * 4 + threadIdx.y) * WT::LY + (threadIdx.x >> WT::LOG_LX);
        if (y >= dh) return;
        Tap t[P::PXT];
        persp_taps<P::PXT, GENERIC>(M, bw0, bw0_shift, jb.fast, x, y, t);
        if (P::PXT == 4) quad_8u<TYPE>((gcu8)jb.src, jb.sw, jb.sh, jb.sstride, (gu8)jb.dst, dw, jb.dstride, x, y, t);
        else             single_px<TYPE>((gcu8)jb.src, jb.sw, jb.sh, jb.sstride, (gu8)jb.dst, jb.dstride, x, y, t[0]);
    }
}

// ==== LDS-staged 8UC3 gather ======================================================================================
// The one-lane-four-pixels gather above spends two thirds of its instructions outside the blend (f64 coordinates per
// lane, per-tap interior tests, 24 B fetched through the texture addresser per 3 B used) and sits at a third of HBM
// peak.  Here ONE WAVE owns a 64 x 16 strip of the destination and a slice of LDS; the four waves of a workgroup (a
// 64 x 64 tile) never wait for each other, so a CU overlaps the loads of some waves with the arithmetic of others:
//   1. a lane owns ONE destination column of the strip: one f64 evaluation at the column's centre (the anchor) serves its
//      16 pixels,
//          X(j) - X_c = j dX / (1 + j p),   p = M7 / W_c,   1 / (1 + e) ~ 1 - e + e^2,
//      a cubic in the row offset j = -7.5 .. 7.5 with per-lane f32 coefficients, evaluated in fixed point with 13 extra
//      fraction bits; a value within 2 units of a rounding tie (1e-3 of the pixels) is recomputed by OpenCV's own f64
//      sequence, so the taps are bit-identical by construction.  Error budget in those units (|dX/dy|, |dY/dy| <= 2 px/px
//      and |p| <= 2e-4, bounded by the host, tile_delta_coords_ok): |j dX| < 15 px = 3.9e6 units, so the f32 roundings of the
//      three fmas and of the coefficients stay under 1.2, the dropped e^3 term under 0.02, the conversion to integer 0.5;
//   2. W keeps its sign over the destination (job.fast >= 1), so the strip maps onto a convex quadrilateral and the taps of
//      its four corner pixels (first / last lane, first / last row: v_readlane, scalar min / max), +- 1 px, bound every tap;
//   3. that source box goes into the wave's LDS slice as one dword per pixel ([B G R x]: 12-byte row loads, three
//      v_alignbyte and one 16-byte LDS write per four pixels, all loads in flight before the first write), out-of-image
//      pixels as the border constant 0 -- no per-tap border or interior test is left in the pixel loop, the four taps of a
//      pixel are two aligned ds_read2_b32, and the 64 lanes read neighbouring dwords (no bank conflicts);
//   4. the blend is the perm + dot4 sequence of blend8_row2 on the BGRx dwords; the finished pixel dwords [B G R 0] of four
//      neighbouring lanes are packed into three dwords with one DPP move and one v_perm_b32, so a strip row leaves as
//      192 contiguous bytes.
// Jobs whose strips may not fit the LDS slice, whose rows are not 16-byte aligned or whose map is not nearly affine take the
// direct kernel above (the host groups the jobs of a call by kernel).
constexpr int kTileW = kWarpTileW, kTileH = kWarpTileH, kWaveRows = kWarpStripRows;
constexpr int kTieList = 60;               // entries of a wave's list of pixels on a rounding tie (256 B with its padding)

struct TileRegion {                        // wave-uniform
    int ox, oy;                            // source pixel of region column 0 / row 0 (ox a multiple of 4)
    int rows, cpr;                         // staged rows, 4-pixel chunks per row
    int pitch;                             // LDS bytes per row (16 cpr)
};
using lds_u32 = const __attribute__((address_space(3))) uint32_t*;
typedef short s16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// BGRx taps: tl, tr of the top row, bl, br of the bottom row -> accumulators, result in byte 2
__device__ __forceinline__ void blend_taps_bgrx(uint32_t tl, uint32_t tr, uint32_t bl, uint32_t br, const Blend8& b,
                                                uint32_t& aB, uint32_t& aG, uint32_t& aR)
{
    uint32_t hB0, hG0, hB1, hG1;
    blend8_row2(__builtin_amdgcn_perm(tr, tl, 0x05010400u), b, hB0, hG0);       // [B0 B1 G0 G1]
    blend8_row2(__builtin_amdgcn_perm(br, bl, 0x05010400u), b, hB1, hG1);
    const uint32_t hR0 = __builtin_amdgcn_udot4(__builtin_amdgcn_perm(tr, tl, 0x0c0c0602u), b.wlo, DR3_HROUND, false);   // [R0 R1 0 0]
    const uint32_t hR1 = __builtin_amdgcn_udot4(__builtin_amdgcn_perm(br, bl, 0x0c0c0602u), b.wlo, DR3_HROUND, false);
    aB = blend8_vert(hB0, hB1, b);
    aG = blend8_vert(hG0, hG1, b);
    aR = blend8_vert(hR0, hR1, b);
}

// The source box of one wave, staged into its LDS slice as BGRx dwords.  CHECKED: the box sticks out of the image (pixels
// outside are the border constant 0).  Lanes are dealt over (row, chunk) pairs, 64 / cpr rows per pass.  Every load is
// unconditional -- rows and chunks outside the image or behind the box are clamped onto valid ones and masked afterwards -- so
// that all loads of a pass are in flight before the first LDS write (a load under a lane condition makes the compiler wait
// for it at the end of the condition).
template <bool CHECKED>
__device__ __forceinline__ void stage_strip_bgrx(const TileRegion& g, gcu8 __restrict__ src, int sw, int sh, int sstride, uint32_t lds0, int lane)
{
    constexpr int U = 10;                                                   // row loads in flight per lane and pass
    // lane -> (row of the pass, chunk): lane / cpr by a 16-bit reciprocal (exact for lane < 64, cpr <= 64)
    const uint32_t inv = (uint32_t)(65536.0f / (float)g.cpr) + 1u;
    const int rpp = (int)((64u * inv) >> 16);                               // rows per pass = 64 / cpr
    const int lr = (int)(((uint32_t)lane * inv) >> 16), c = lane - lr * g.cpr;
    if (lr >= rpp) return;
    const int px0 = g.ox + 4 * c;                                           // a multiple of 4
    uint32_t mk[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    uint32_t ob = 4u, oc = 8u, cb = (uint32_t)(3 * px0);                    // byte offsets of the chunk and of its second / third dword
    if (CHECKED) {
#pragma unroll
        for (int i = 0; i < 4; ++i) mk[i] = (unsigned)(px0 + i) < (unsigned)sw ? 0xffffffffu : 0u;
        const int rowbytes = 3 * sw;
        if (!(px0 >= 0 && px0 < sw)) cb = 0u;                               // nothing of the chunk is inside: any valid address, all masked
        // the chunk that straddles the right edge: a dword whose first byte is valid is read whole (it stays in that byte's page),
        // the others are replaced by the first one and masked
        if ((int)cb + 4 >= rowbytes) ob = 0u;
        if ((int)cb + 8 >= rowbytes) oc = 0u;
    }
    const uint32_t lstep = (uint32_t)(rpp * g.pitch), gstep = __umul24((uint32_t)rpp, (uint32_t)sstride);
    uint32_t loff = lds0 + (uint32_t)(lr * g.pitch + 16 * c);
    uint32_t goff = cb + (uint32_t)((g.oy + lr) * sstride);                 // !CHECKED: every row of the box is inside the image
    const int last = g.rows - 1;
    for (int r0 = lr; r0 <= last; r0 += rpp * U) {
        uint32_t va[U], vb[U], vc[U], rm[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            rm[u] = 0xffffffffu;
            if (!CHECKED) {                                                 // behind the box: the last row again (written twice, same data)
                const DR3_G uint32_t* q4 = gcast<uint32_t>(src + (r0 + u * rpp <= last ? goff + (uint32_t)u * gstep : cb + (uint32_t)((g.oy + last) * sstride)));
                va[u] = q4[0]; vb[u] = q4[1]; vc[u] = q4[2];
            } else {
                int sy = g.oy + min(r0 + u * rpp, last);
                rm[u] = (unsigned)sy < (unsigned)sh ? 0xffffffffu : 0u; sy = min(max(sy, 0), sh - 1);
                gcu8 q = src + (__umul24((uint32_t)sy, (uint32_t)sstride) + cb);
                va[u] = *gcast<uint32_t>(q); vb[u] = *gcast<uint32_t>(q + ob); vc[u] = *gcast<uint32_t>(q + oc);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            u32x4 o;
            o.x = va[u] & mk[0] & rm[u];
            o.y = __builtin_amdgcn_alignbyte(vb[u], va[u], 3) & mk[1] & rm[u];
            o.z = __builtin_amdgcn_alignbyte(vc[u], vb[u], 2) & mk[2] & rm[u];
            o.w = (vc[u] >> 8) & mk[3] & rm[u];
            const uint32_t la = r0 + u * rpp <= last ? loff + (uint32_t)u * lstep : lds0 + (uint32_t)(last * g.pitch + 16 * c);
            *(__attribute__((address_space(3))) u32x4*)(uintptr_t)la = o;
        }
        loff += U * lstep; goff += U * gstep;
    }
}

// One 8UC3 destination column of 16 rows by OpenCV's own sequence and per-tap checked loads: the strip kernel's way out for a
// strip whose source box does not fit (the host routes such jobs to the direct kernel, so this is a safety net, not a path
// that is timed).
__device__ __noinline__ void tile_fallback_8uc3(const WarpJobDev& jb, int x, int y0)
{
    double M[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) M[i] = jb.M[i];
    gcu8 src = (gcu8)jb.src; gu8 dst = (gu8)jb.dst;
    if (x >= jb.dw) return;
    const int x0 = block_origin(x, jb.bw0, jb.bw0_shift);
    for (int j = 0; j < kWaveRows; ++j) {
        const int y = y0 + j;
        if (y >= jb.dh) return;
        const Tap t = persp_tap(M, persp_base(M, x0, y), (double)(x - x0));
        uint32_t aB = 0u, aG = 0u, aR = 0u;
        pixel_8uc3(src, jb.sw, jb.sh, jb.sstride, false, t, aB, aG, aR);
        gu8 o = dst + (long long)y * jb.dstride + (long long)x * 3;
        o[0] = (uint8_t)byte2(aB); o[1] = (uint8_t)byte2(aG); o[2] = (uint8_t)byte2(aR);
    }
}

// a constant the compiler must keep in a VGPR: on gfx950 a VOP2 instruction with a literal or an SGPR operand issues in 4
// clocks, with VGPR / inline operands in 2 (profiles/r02_lk/ubench_valu_rate.txt)
__device__ __forceinline__ uint32_t vgpr_const(uint32_t c) { uint32_t r; asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "s"(c)); return r; }

// One pixel on (or next to) a rounding tie: the guarded f64 evaluation of the direct kernel (reciprocal + Newton step, 13 extra
// fraction bits of its own) and, where that one cannot decide either, OpenCV's own sequence.  Taps from the staged box when they
// are inside it (a pixel that sits on a tie may round to the column or row next to the box), else from the image: [B G R 0]
__device__ __forceinline__ Tap tile_redo_tap(const double* __restrict__ M, int bw0_shift, int x, int y)
{
    const int x0 = (x >> bw0_shift) << bw0_shift;
    const PerspBase pb0 = persp_base(M, x0, y);
    const double S = 262144.0;
    PerspFast f; f.X0s = pb0.X0 * S; f.Y0s = pb0.Y0 * S; f.M0s = M[0] * S; f.M3s = M[3] * S;
    Tap e;
    if (!persp_tap_fast(M, pb0, f, (double)(x - x0), e)) e = persp_tap(M, pb0, (double)(x - x0));
    return e;
}
__device__ __forceinline__ uint32_t tile_redo_pixel(const double* __restrict__ M, int bw0_shift, const TileRegion& g, uint32_t lds0,
                                                    gcu8 __restrict__ src, int sw, int sh, int sstride, int x, int y)
{
    const Tap e = tile_redo_tap(M, bw0_shift, x, y);
    const int rx = e.sx - g.ox, ry = e.sy - g.oy;
    uint32_t aB, aG, aR;
    if (rx >= 0 && rx + 1 < 4 * g.cpr && ry >= 0 && ry + 1 < g.rows) {
        const uint32_t addr = lds0 + __umul24((uint32_t)ry, (uint32_t)g.pitch) + 4u * (uint32_t)rx;
        const lds_u32 pt = (lds_u32)(uintptr_t)addr, pb = (lds_u32)(uintptr_t)(addr + (uint32_t)g.pitch);
        blend_taps_bgrx(pt[0], pt[1], pb[0], pb[1], blend8_weights(e.fx, e.fy), aB, aG, aR);
    } else {
        aB = aG = aR = 0u;
        pixel_8uc3(src, sw, sh, sstride, false, e, aB, aG, aR);
    }
    return __builtin_amdgcn_perm(aR, __builtin_amdgcn_perm(aG, aB, 0x0c0c0602u), 0x0c060100u);
}

// cv::addWeighted for 8-bit data: saturate(cvRound(l * alpha + r * beta + gamma)) in f32
__device__ __forceinline__ uint32_t add_weighted_u8(uint32_t l, uint32_t r, float alpha, float beta, float gamma)
{
    const float t = (float)l * alpha + (float)r * beta + gamma;
    const int v = __float2int_rn(t);
    return (uint32_t)(v < 0 ? 0 : v > 255 ? 255 : v);
}

// ---- the left image of a fused stitch (reference src/stitch.cpp:74: warpPerspective(left, translate_top_left)) -----------------------
// A pure translation: OpenCV's coordinates are X = rint(32 x + 32 tx) = 32 x + Kx for every pixel (the f64 products and sums are exact
// and 32 x is even, so round-half-even decides every pixel alike), i.e. ONE tap offset (ix, iy) = (Kx >> 5, Ky >> 5) and ONE pair of
// fractions (Kx & 31, Ky & 31) for the whole canvas.  In the strip form of the fused stitch a lane reads the two taps of a row of its
// column as ONE aligned 12-byte load (the 64 lanes of a wave read 200 contiguous bytes: no LDS), blends with constant weights, and the horizontal sums of a row serve the pixel above and the one below (the integers
// distribute exactly, see Blend8).  Strips that touch the image's border take per-tap checked loads instead (4 % of a 4000 x 3000 image).
struct StitchLeft {
    const uint8_t* src; int sw, sh, sstride;   // rows 16-byte aligned
    int ix, iy;                                // tap of canvas pixel (x, y): (x + ix, y + iy)
    uint32_t wlo, whi, wy0, wy1;               // Blend8 of the constant fractions
    int fx, fy;                                // the fractions themselves (strips on the image border take checked taps)
    int avg;                                   // alpha == beta == 0.5, gamma == 0 (the reference's literals): the v_lerp_u8 form
    float alpha, beta, gamma;
};

// cv::addWeighted on [B G R 0] dwords.  avg: l * 0.5f + r * 0.5f + 0.f is exact in f32, so cvRound gives (l + r) / 2 rounded half to even:
// floor((l + r) / 2) per byte (v_lerp_u8) plus its own low bit where l + r is odd.
__device__ __forceinline__ uint32_t stitch_blend_px(uint32_t l, uint32_t r, const StitchLeft& sl)
{
    if (sl.avg) {
        const uint32_t f = __builtin_amdgcn_lerp(l, r, 0u);
        return f + (((l ^ r) & 0x01010101u) & f);
    }
    const uint32_t b = add_weighted_u8(l & 255u, r & 255u, sl.alpha, sl.beta, sl.gamma);
    const uint32_t g = add_weighted_u8((l >> 8) & 255u, (r >> 8) & 255u, sl.alpha, sl.beta, sl.gamma);
    const uint32_t c = add_weighted_u8((l >> 16) & 255u, (r >> 16) & 255u, sl.alpha, sl.beta, sl.gamma);
    return b | (g << 8) | (c << 16);
}

// horizontal sums (B, G, R) of the two taps [B0 G0 R0 B1 G1 R1] of one row: 12 bytes from the dword-aligned address p below the taps
// (ONE aligned global_load_dwordx3: the 64 lanes of a wave read 200 contiguous bytes; an unaligned 8-byte load per lane measured 28 %
// slower than staging the box through LDS), shifted into place by the lane's byte offset sh = taps' address & 3
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ u32x3 left_row_load(gcu8 p) { return *reinterpret_cast<const DR3_G u32x3*>(p); }
__device__ __forceinline__ void left_row_sums(const u32x3 v, uint32_t sh, const Blend8& w, uint32_t& hb, uint32_t& hg, uint32_t& hr)
{
    const uint32_t lo = __builtin_amdgcn_alignbyte(v.y, v.x, sh), hi = __builtin_amdgcn_alignbyte(v.z, v.y, sh);
    blend8_row2(__builtin_amdgcn_perm(hi, lo, 0x04010300u), w, hb, hg);                                      // [B0 B1 G0 G1]
    hr = __builtin_amdgcn_udot4(__builtin_amdgcn_perm(hi, lo, 0x0c0c0502u), w.wlo, DR3_HROUND, false);       // [R0 R1 0 0]
}
__device__ __forceinline__ uint32_t left_px(uint32_t hb0, uint32_t hg0, uint32_t hr0, uint32_t hb1, uint32_t hg1, uint32_t hr1, const Blend8& w)
{
    const uint32_t aB = blend8_vert(hb0, hb1, w), aG = blend8_vert(hg0, hg1, w), aR = blend8_vert(hr0, hr1, w);
    return __builtin_amdgcn_perm(aR, __builtin_amdgcn_perm(aG, aB, 0x0c0c0602u), 0x0c060100u);             // [B G R 0]
}

// One canvas column of 16 rows of a fused stitch by OpenCV's own sequences and per-tap checked loads (the strip form's way out)
__device__ __forceinline__ void stitch_fallback_8uc3(const WarpJobDev& jr, const WarpJobDev& jl, const StitchLeft& sl, int x, int y0)
{
    if (x >= jr.dw) return;
    gu8 dst = (gu8)jr.dst;
    for (int j = 0; j < kWaveRows; ++j) {
        const int y = y0 + j;
        if (y >= jr.dh) return;
        uint32_t px[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const WarpJobDev& jb = s ? jr : jl;
            double M[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) M[i] = jb.M[i];
            const int x0 = block_origin(x, jb.bw0, jb.bw0_shift);
            const Tap t = persp_tap(M, persp_base(M, x0, y), (double)(x - x0));
            uint32_t aB = 0u, aG = 0u, aR = 0u;
            pixel_8uc3((gcu8)jb.src, jb.sw, jb.sh, jb.sstride, false, t, aB, aG, aR);
            px[s] = byte2(aB) | (byte2(aG) << 8) | (byte2(aR) << 16);
        }
        const uint32_t v = stitch_blend_px(px[0], px[1], sl);
        gu8 o = dst + (long long)y * jr.dstride + (long long)x * 3;
        o[0] = (uint8_t)v; o[1] = (uint8_t)(v >> 8); o[2] = (uint8_t)(v >> 16);
    }
}

// ---- the fused paste on strips (reference src/panorama.cpp:178-212; the direct form and its semantics: panorama_add_kernel below) ------
// The strip machinery evaluates only the box of the pasted image (columns [x0, dw), rows [y0, dh) of the canvas, dw / dh in the job
// record); a pixel is the 32FC4 warp's f32 4-term sum of its taps times the feather ramp of its column, added to the canvas unless its
// scaled B, G and R are all exactly 0.
struct PasteStrip {
    float* canvas; int cstride, slab_y0;       // `canvas` addresses canvas row slab_y0
    const float* ramp; int min_x;              // ramp[x - min_x]
    int x0, y0;                                // the box's first column / row: origin of the strips
};
typedef float f32x4_t __attribute__((ext_vector_type(4)));

// (B, G, R) of one pasted pixel from the BGRx dwords of its taps: v = tl w0 + tr w1 + bl w2 + br w3 in that order, no contraction, times r
__device__ __forceinline__ void paste_px_bgrx(uint32_t tl, uint32_t tr, uint32_t bl, uint32_t br, int fx, int fy, float r, float o[3])
{
    float w[4]; weights_32f(fx, fy, w);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v0 = (float)((tl >> (8 * c)) & 0xffu), v1 = (float)((tr >> (8 * c)) & 0xffu);
        const float v2 = (float)((bl >> (8 * c)) & 0xffu), v3 = (float)((br >> (8 * c)) & 0xffu);
        o[c] = (v0 * w[0] + v1 * w[1] + v2 * w[2] + v3 * w[3]) * r;
    }
}
// one pasted pixel by per-tap checked loads (border 0) and the read-modify-write: the strip form's way for pixels on a rounding tie
// and for strips it cannot take
__device__ __forceinline__ void paste_px_checked(gcu8 __restrict__ src, int sw, int sh, int sstride, const Tap e, const PasteStrip& ps, int x, int y)
{
    if (e.sx >= sw || e.sx + 1 < 0 || e.sy >= sh || e.sy + 1 < 0) return;
    float w[4]; weights_32f(e.fx, e.fy, w);
    const float r = ps.ramp[x - ps.min_x];
    float o[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const uint32_t q = quad_checked(src, sw, sh, sstride, e.sx, e.sy, 3, c);
        const float v0 = (float)(q & 0xffu), v1 = (float)((q >> 8) & 0xffu), v2 = (float)((q >> 16) & 0xffu), v3 = (float)(q >> 24);
        o[c] = (v0 * w[0] + v1 * w[1] + v2 * w[2] + v3 * w[3]) * r;
    }
    if (o[0] == 0.f && o[1] == 0.f && o[2] == 0.f) return;
    DR3_G f32x4_t* cp = reinterpret_cast<DR3_G f32x4_t*>((gu8)reinterpret_cast<uint8_t*>(ps.canvas) + (long long)(y - ps.slab_y0) * ps.cstride) + x;
    f32x4_t c4 = *cp;
    c4.x += o[0]; c4.y += o[1]; c4.z += o[2]; c4.w += r;
    *cp = c4;
}
__device__ __forceinline__ void paste_fallback(const WarpJobDev& jb, const PasteStrip& ps, int x, int y0)
{
    double M[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) M[i] = jb.M[i];
    if (x >= jb.dw) return;
    const int x0 = block_origin(x, jb.bw0, jb.bw0_shift);
    for (int j = 0; j < kWaveRows; ++j) {
        const int y = y0 + j;
        if (y >= jb.dh) return;
        paste_px_checked((gcu8)jb.src, jb.sw, jb.sh, jb.sstride, persp_tap(M, persp_base(M, x0, y), (double)(x - x0)), ps, x, y);
    }
}

// One 64 x 64 tile (four strips, one per wave) of one job.  MODE 0: the 8UC3 warp.  MODE 2 (PASTE): the fused paste above.  MODE 1 (STITCH): jb is the RIGHT image's job of a fused stitch, jl the left image's
// (exact fallback), sl its translation form: the strip's pixels are addWeighted(left, right) instead of the warped right image.
// Every job: fast == 3, rows 16-byte aligned, power-of-two block split, strips inside the launch's LDS slice
// (WarpJobDev::tile == 2, tile_kernel_lds on the host).  LDS of wave w: [slice: lds_bytes][tie list: 256 B] at w (lds_bytes + 256).
template <int MODE>
__device__ __forceinline__ void strip_tile_body(const WarpJobDev& jb, const WarpJobDev* jl, const StitchLeft* slp, const PasteStrip* pp, int t, int tiles_x, int lds_bytes, uint8_t* s_dyn,
                                                int wv, int lds_slot)     // wv: the wave's strip of the tile; lds_slot: its slice of the block's LDS
{
    constexpr bool STITCH = MODE == 1, PASTE = MODE == 2;
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const int tx0 = (PASTE ? pp->x0 : 0) + tx * kTileW, ty0 = (PASTE ? pp->y0 : 0) + ty * kTileH;
    const int dw = jb.dw, dh = jb.dh;
    const int lane = threadIdx.x;
    const int x = tx0 + lane, y0 = ty0 + kWaveRows * wv;
    const int n_rows = min(kWaveRows, dh - y0);                             // wave-uniform
    if (tx0 >= dw || n_rows <= 0) return;
    const int bw0_shift = jb.bw0_shift;
    double M[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) M[i] = jb.M[i];
    gcu8 src = (gcu8)jb.src; gu8 dst = (gu8)jb.dst;
    const int sw = jb.sw, sh = jb.sh, sstride = jb.sstride, dstride = jb.dstride;
    const bool dst_aligned = ((reinterpret_cast<uintptr_t>(jb.dst) | (uintptr_t)dstride) & 3) == 0;
    const bool src16 = ((reinterpret_cast<uintptr_t>(jb.src) | (uintptr_t)sstride) & 15) == 0 &&
                       sstride < (1 << 24) && sh < (1 << 24) && (long long)sstride * sh < (1LL << 31);
    if (!(src16 && bw0_shift >= 0 && jb.fast == 3)) { if (STITCH) stitch_fallback_8uc3(jb, *jl, *slp, x, y0); else if (PASTE) paste_fallback(jb, *pp, x, y0); else tile_fallback_8uc3(jb, x, y0); return; }
    const uint32_t lds0 = (uint32_t)(uintptr_t)s_dyn + (uint32_t)lds_slot * (uint32_t)(lds_bytes + 256);    // this wave's slice (LDS byte address)
    const uint32_t list0 = lds0 + (uint32_t)lds_bytes;

    // anchor: (x, y0 + 7.5), f64; units: 2^-13 of 1/32 px
    const double xd = (double)x, yd = (double)y0 + 7.5;
    const double W0 = __builtin_fma(M[6], xd, __builtin_fma(M[7], yd, M[8]));
    const double NX = __builtin_fma(M[0], xd, __builtin_fma(M[1], yd, M[2]));
    const double NY = __builtin_fma(M[3], xd, __builtin_fma(M[4], yd, M[5]));
    double rw = __builtin_amdgcn_rcp(W0);
    rw = __builtin_fma(__builtin_fma(-W0, rw, 1.0), rw, rw);
    const double rs = rw * 262144.0;                                          // 32 * 2^13
    const double X0 = NX * rs, Y0 = NY * rs, pd = M[7] * rw;
    // X(j) - X_c = j a1 (1 - j p + j^2 p^2): Horner coefficients of the cubic, X and Y side by side (v_pk_fma_f32)
    const float pf = (float)pd;
    const f32x2 a1 = {(float)__builtin_fma(-pd, X0, M[1] * rs), (float)__builtin_fma(-pd, Y0, M[4] * rs)};
    const f32x2 a2 = -a1 * pf, a3 = a1 * (pf * pf);
    const double X0r = __builtin_rint(X0), Y0r = __builtin_rint(Y0);
    const f32x2 rem = {(float)(X0 - X0r), (float)(Y0 - Y0r)};
    // + 4096: round to nearest 1/32 px becomes a floor; + guard: the tie guard's bias (a value whose low 13 bits are < 2 guard is
    // within `guard` units of a tie; the host sets guard to 1 or 2 from the job's error bound).  The f32 offset (|.| < 2^22) is
    // rounded to an integer by adding 1.5 * 2^23; the bits of that constant leave with the anchor.
    const f32x2 magic = {12582912.0f, 12582912.0f};
    const int guard = jb.tile_guard;
    const uint32_t g2 = 2u * (uint32_t)guard;
    const int bX0 = (int)X0r + (4096 + guard) - 0x4B400000, bY0 = (int)Y0r + (4096 + guard) - 0x4B400000;     // absolute

    // the strip's source box from its four corner pixels (convex image of the strip)
    TileRegion g;
    {
        const float ja = -7.5f, jb_ = (float)(n_rows - 1) - 7.5f;
        const f32x2 jav = {ja, ja}, jbv = {jb_, jb_};
        const f32x2 da = __builtin_elementwise_fma(jav, __builtin_elementwise_fma(jav, __builtin_elementwise_fma(jav, a3, a2), a1), rem) + magic;
        const f32x2 db = __builtin_elementwise_fma(jbv, __builtin_elementwise_fma(jbv, __builtin_elementwise_fma(jbv, a3, a2), a1), rem) + magic;
        const int sxa = (bX0 + (int)__float_as_uint(da.x)) >> 18, sya = (bY0 + (int)__float_as_uint(da.y)) >> 18;
        const int sxb = (bX0 + (int)__float_as_uint(db.x)) >> 18, syb = (bY0 + (int)__float_as_uint(db.y)) >> 18;
        const int ll = min(kTileW, dw - tx0) - 1;                           // last lane inside the destination
        const int x_lo = min(min(__builtin_amdgcn_readlane(sxa, 0), __builtin_amdgcn_readlane(sxb, 0)),
                             min(__builtin_amdgcn_readlane(sxa, ll), __builtin_amdgcn_readlane(sxb, ll))) - 1;
        const int x_hi = max(max(__builtin_amdgcn_readlane(sxa, 0), __builtin_amdgcn_readlane(sxb, 0)),
                             max(__builtin_amdgcn_readlane(sxa, ll), __builtin_amdgcn_readlane(sxb, ll))) + 2;      // right tap + 1
        const int y_lo = min(min(__builtin_amdgcn_readlane(sya, 0), __builtin_amdgcn_readlane(syb, 0)),
                             min(__builtin_amdgcn_readlane(sya, ll), __builtin_amdgcn_readlane(syb, ll))) - 1;
        const int y_hi = max(max(__builtin_amdgcn_readlane(sya, 0), __builtin_amdgcn_readlane(syb, 0)),
                             max(__builtin_amdgcn_readlane(sya, ll), __builtin_amdgcn_readlane(syb, ll))) + 2;
        g.ox = x_lo & ~3; g.oy = y_lo;
        g.rows = y_hi - y_lo + 1;
        g.cpr = (x_hi - g.ox + 4) >> 2;
        g.pitch = g.cpr * 16;
    }
    if (!(g.cpr >= 1 && g.cpr <= 64 && g.rows >= 1 && g.rows <= 4096 && g.rows * g.pitch <= lds_bytes && g.ox > -16384 && g.oy > -16384)) {
        if (STITCH) stitch_fallback_8uc3(jb, *jl, *slp, x, y0); else if (PASTE) paste_fallback(jb, *pp, x, y0); else tile_fallback_8uc3(jb, x, y0);
        return;
    }
    // a box that lies outside the image by more than the pixel a rounding tie can move a tap: every pixel of the strip is the border
    // constant -- nothing is staged, no coordinate is evaluated (the canvas of a stitch is mostly such strips for one image or the other)
    const bool r_any = !(g.ox - 2 >= sw || g.ox + 4 * g.cpr + 2 <= 0 || g.oy - 2 >= sh || g.oy + g.rows + 2 <= 0);
    if (PASTE && !r_any) return;                                             // every scaled sample is exactly 0: nothing is added
    const int xc = min(x, dw - 1);                                          // PASTE: a lane behind the box's last column shadows it (loads only)
    float pr = 0.f;
    // PASTE: the canvas pixels of a group of four rows are fetched a group ahead of their read-modify-write, by every lane (a load
    // under a lane condition is waited for where the condition ends)
    gu8 crow0 = nullptr;
    long long cstep = 0;
    f32x4_t cv[4] = {};
    if (PASTE) {
        crow0 = (gu8)reinterpret_cast<uint8_t*>(pp->canvas) + (long long)(y0 - pp->slab_y0) * pp->cstride + 16ll * xc;
        cstep = pp->cstride;
#pragma unroll
        for (int u = 0; u < 4; ++u) cv[u] = *reinterpret_cast<DR3_G f32x4_t*>(crow0 + (long long)min(u, n_rows - 1) * cstep);
        pr = pp->ramp[xc - pp->min_x];
        __builtin_amdgcn_sched_barrier(0);                       // ahead of the source box's loads
    }
    if (r_any) {
        if (g.ox >= 0 && g.ox + 4 * g.cpr <= sw && g.oy >= 0 && g.oy + g.rows <= sh) stage_strip_bgrx<false>(g, src, sw, sh, sstride, lds0, lane);
        else                                                                          stage_strip_bgrx<true>(g, src, sw, sh, sstride, lds0, lane);
    }
    // STITCH: the left image under this strip: columns tx0 + ix .. + 65, rows y0 + iy .. + n_rows.  lmode 0: none of it inside the image;
    // 1: all of it (and the two bytes an 8-byte tap load reads beyond the right tap) inside: direct loads; 2: on the border: checked taps
    int lmode = 0;
    gcu8 lrow = nullptr;                                                    // dword below the lane's taps of the strip's first row
    uint32_t lsh = 0u;
    Blend8 wl; wl.wlo = wl.whi = wl.wy0 = wl.wy1 = 0u;
    if (STITCH) {
        const StitchLeft& sl = *slp;
        wl.wlo = sl.wlo; wl.whi = sl.whi; wl.wy0 = sl.wy0; wl.wy1 = sl.wy1;
        const int lx0 = tx0 + sl.ix, ly0 = y0 + sl.iy;
        const bool any = !(lx0 >= sl.sw || lx0 + kTileW + 1 <= 0 || ly0 >= sl.sh || ly0 + n_rows + 1 <= 0);
        const bool inside = lx0 >= 0 && lx0 + kTileW + 4 <= sl.sw && ly0 >= 0 && ly0 + n_rows + 1 <= sl.sh;      // + 4: the 12-byte load's reach
        lmode = !any ? 0 : inside ? 1 : 2;
        const long long lb = 3ll * (lx0 + lane);
        lsh = (uint32_t)lb & 3u;                                            // rows are 16-byte aligned: the same byte offset on every row
        lrow = (gcu8)sl.src + ((long long)ly0 * sl.sstride + (lb & ~3ll));
    }
    __builtin_amdgcn_wave_barrier();                            // a wave's LDS operations complete in order: no s_barrier
    const uint32_t pitch = (uint32_t)g.pitch;
    // region-relative: bits 18.. of aX / aY are the LDS column / row
    const int aX0 = bX0 - g.ox * 262144, aY0 = bY0 - g.oy * 262144;
    const uint32_t k8191 = vgpr_const(8191u), k7c0 = vgpr_const(0x7c0u), k800 = vgpr_const(0x800u), k64k = vgpr_const(65536u);
    const uint32_t kfffc = vgpr_const(0xfffcfffcu);
    const uint32_t kdot = 1u | ((pitch >> 2) << 16);                        // 4 rx * 1 + 4 ry * (pitch / 4)
    // output: lanes 4m .. 4m+2 write the three dwords of pixels 4m .. 4m+3
    const int sub = lane & 3;
    const uint32_t osel = sub == 0 ? 0x04020100u : sub == 1 ? 0x05040201u : 0x06050402u;
    const bool quad_ok = dst_aligned && (x | 3) < dw;                       // the whole 4-pixel group exists and rows are dword-aligned
    const bool st_dword = quad_ok && sub != 3, st_bytes = !quad_ok && x < dw;
    const uint32_t ooff = quad_ok ? (uint32_t)(4 * (lane - (lane >> 2))) : (uint32_t)(3 * lane);
    gu8 obase = dst + (long long)y0 * dstride + (long long)tx0 * 3;         // wave-uniform
    uint32_t own = 0u;                                                      // ties of this lane's column that the list had no room for
    int n_list = 0;                                                         // wave-uniform
    uint32_t lhb = 0u, lhg = 0u, lhr = 0u;                                  // STITCH: horizontal sums of the left image's row above the next pixel row
    // the left image's rows are fetched one group of four pixel rows ahead of their use (a load issued where it is used leaves the
    // wave waiting for memory once per row: the direct form's speed)
    u32x3 lv[4] = {};
    if (STITCH && lmode == 1) {
        left_row_sums(left_row_load(lrow), lsh, wl, lhb, lhg, lhr);
#pragma unroll
        for (int u = 0; u < 4; ++u) lv[u] = left_row_load(lrow + (long long)min(u + 1, n_rows) * slp->sstride);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j0 = 0; j0 < kWaveRows; j0 += 4) {
        if (j0 >= n_rows) break;
        uint32_t nearest = 0xffffffffu;
        int aX[4], aY[4];
        uint32_t P[4] = {0u, 0u, 0u, 0u};
        float po[4][3] = {};
        u32x3 ln[4] = {};
        if (STITCH && lmode == 1 && j0 + 4 < n_rows) {
#pragma unroll
            for (int u = 0; u < 4; ++u) ln[u] = left_row_load(lrow + (long long)min(j0 + 5 + u, n_rows) * slp->sstride);
            __builtin_amdgcn_sched_barrier(0);           // the scheduler would sink the loads to their uses, a group later
        }
        f32x4_t cn[4] = {};
        if (PASTE && j0 + 4 < n_rows) {
#pragma unroll
            for (int u = 0; u < 4; ++u) cn[u] = *reinterpret_cast<DR3_G f32x4_t*>(crow0 + (long long)min(j0 + 4 + u, n_rows - 1) * cstep);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (r_any) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float oj = (float)(j0 + u) - 7.5f;
            const f32x2 ojv = {oj, oj};
            const f32x2 d = __builtin_elementwise_fma(ojv, __builtin_elementwise_fma(ojv, __builtin_elementwise_fma(ojv, a3, a2), a1), rem) + magic;
            aX[u] = aX0 + (int)__float_as_uint(d.x); aY[u] = aY0 + (int)__float_as_uint(d.y);
            const uint32_t c = __builtin_amdgcn_perm((uint32_t)aY[u], (uint32_t)aX[u], 0x07060302u) & kfffc;    // [4 rx | 4 ry], fraction bits cleared
            const uint32_t addr = (uint32_t)__builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2_t, c), __builtin_bit_cast(s16x2_t, kdot), (int)lds0, false);
            const lds_u32 pt = (lds_u32)(uintptr_t)addr, pb = (lds_u32)(uintptr_t)(addr + pitch);
            const uint32_t tl = pt[0], tr = pt[1], bl = pb[0], br = pb[1];
            nearest = min(nearest, min((uint32_t)aX[u] & k8191, (uint32_t)aY[u] & k8191));
            if (PASTE) {
                paste_px_bgrx(tl, tr, bl, br, (int)__builtin_amdgcn_ubfe((uint32_t)aX[u], 13, 5), (int)__builtin_amdgcn_ubfe((uint32_t)aY[u], 13, 5), pr, po[u]);
                continue;
            }
            Blend8 w;
            w.wlo = __umul24(__builtin_amdgcn_ubfe((uint32_t)aX[u], 13, 5), 255u) + 32u;
            w.whi = __umul24(w.wlo, k64k);
            w.wy1 = ((uint32_t)aY[u] >> 7) & k7c0;
            w.wy0 = k800 - w.wy1;
            uint32_t aB, aG, aR;
            blend_taps_bgrx(tl, tr, bl, br, w, aB, aG, aR);
            P[u] = __builtin_amdgcn_perm(aR, __builtin_amdgcn_perm(aG, aB, 0x0c0c0602u), 0x0c060100u);           // [B G R 0]
        }
        }
        if (STITCH) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                uint32_t Lp = 0u;
                if (lmode == 1 && j0 + u < n_rows) {             // wave-uniform
                    uint32_t nb_, ng_, nr_;
                    left_row_sums(lv[u], lsh, wl, nb_, ng_, nr_);
                    Lp = left_px(lhb, lhg, lhr, nb_, ng_, nr_, wl);
                    lhb = nb_; lhg = ng_; lhr = nr_;
                } else if (lmode == 2 && j0 + u < n_rows) {
                    Tap tl_; tl_.sx = x + slp->ix; tl_.sy = y0 + j0 + u + slp->iy; tl_.fx = slp->fx; tl_.fy = slp->fy;
                    uint32_t aB = 0u, aG = 0u, aR = 0u;
                    pixel_8uc3((gcu8)slp->src, slp->sw, slp->sh, slp->sstride, false, tl_, aB, aG, aR);
                    Lp = __builtin_amdgcn_perm(aR, __builtin_amdgcn_perm(aG, aB, 0x0c0c0602u), 0x0c060100u);
                }
                P[u] = stitch_blend_px(Lp, P[u], *slp);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) lv[u] = ln[u];
        }
        if (r_any && __builtin_amdgcn_ballot_w64(nearest < g2)) {                                                                   // rare: note the pixels on a tie in the wave's list
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool tie = (min((uint32_t)aX[u] & 8191u, (uint32_t)aY[u] & 8191u) < g2) && x < dw && j0 + u < n_rows;
                const unsigned long long m = __builtin_amdgcn_ballot_w64(tie);
                if (tie) {
                    const int pos = n_list + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    if (pos < kTieList) *(__attribute__((address_space(3))) uint32_t*)(uintptr_t)(list0 + 4u * (uint32_t)pos) = ((uint32_t)(j0 + u) << 8) | (uint32_t)lane;
                    else own |= 1u << (j0 + u);
                }
                n_list += __builtin_popcountll(m);
            }
        }
        if (PASTE) {
            // written back by the lanes whose pixel is inside the box, off a rounding tie (those are pasted once, below) and not
            // "empty" (:205-207)
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (j0 + u < n_rows) {
                    const bool tie = min((uint32_t)aX[u] & 8191u, (uint32_t)aY[u] & 8191u) < g2;
                    if (x < dw && !tie && !(po[u][0] == 0.f && po[u][1] == 0.f && po[u][2] == 0.f)) {
                        f32x4_t v = cv[u];
                        v.x += po[u][0]; v.y += po[u][1]; v.z += po[u][2]; v.w += pr;
                        *reinterpret_cast<DR3_G f32x4_t*>(crow0 + (long long)(j0 + u) * cstep) = v;
                    }
                }
#pragma unroll
            for (int u = 0; u < 4; ++u) cv[u] = cn[u];
            continue;
        }
        uint32_t out[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)                                          // row_shl:1: the pixel of lane + 1
            out[u] = __builtin_amdgcn_perm((uint32_t)__builtin_amdgcn_update_dpp(0, (int)P[u], 0x101, 0xf, 0xf, true), P[u], osel);
        gu8 orow = obase + (long long)j0 * dstride;
        if (st_dword) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (j0 + u < n_rows) st_out(gcast<uint32_t>(orow + (long long)u * dstride + ooff), out[u]);
        } else if (st_bytes) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (j0 + u < n_rows) { gu8 o = orow + (long long)u * dstride + ooff; o[0] = (uint8_t)P[u]; o[1] = (uint8_t)(P[u] >> 8); o[2] = (uint8_t)(P[u] >> 16); }
        }
    }
    // pixels on a rounding tie (1e-3 of them): OpenCV's own sequence, stored over the first result (same wave: its stores to one
    // address keep their order)
    n_list = min(n_list, kTieList);
    __builtin_amdgcn_wave_barrier();
    for (int i = lane;;) {
        int xe, ye;
        if (i < n_list) { const uint32_t e = *(lds_u32)(uintptr_t)(list0 + 4u * (uint32_t)i); xe = tx0 + (int)(e & 255u); ye = y0 + (int)(e >> 8); i += 64; }
        else if (own) { xe = x; ye = y0 + __builtin_ctz(own); own &= own - 1u; }
        else break;
        if (PASTE) { paste_px_checked(src, sw, sh, sstride, tile_redo_tap(M, bw0_shift, xe, ye), *pp, xe, ye); continue; }
        uint32_t v = tile_redo_pixel(M, bw0_shift, g, lds0, src, sw, sh, sstride, xe, ye);
        if (STITCH) {
            uint32_t Lp = 0u;
            if (lmode) {
                Tap tl_; tl_.sx = xe + slp->ix; tl_.sy = ye + slp->iy; tl_.fx = slp->fx; tl_.fy = slp->fy;
                uint32_t aB = 0u, aG = 0u, aR = 0u;
                pixel_8uc3((gcu8)slp->src, slp->sw, slp->sh, slp->sstride, false, tl_, aB, aG, aR);
                Lp = __builtin_amdgcn_perm(aR, __builtin_amdgcn_perm(aG, aB, 0x0c0c0602u), 0x0c060100u);
            }
            v = stitch_blend_px(Lp, v, *slp);
        }
        gu8 o = dst + (long long)ye * dstride + (long long)xe * 3;
        o[0] = (uint8_t)v; o[1] = (uint8_t)(v >> 8); o[2] = (uint8_t)(v >> 16);
    }
}


__global__ __launch_bounds__(256) void warp_tile_8uc3_kernel(const WarpJobDev* __restrict__ jobs, int tiles_x, int n_tiles, int lds_bytes, int run, int n_jobs_arg)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
    // block -> (job, tile): linear block ids b, b + 8 share an XCD (and its L2).
    // run > 0: an XCD takes `run` x-adjacent tiles in a row, so that the source boxes and the 192-byte row segments of neighbouring tiles
    // meet in one L2 (run = 4 on wide destinations: 1.15 instead of 1.38 times the algorithmic bytes on the 4000 x 3000 tiles at the same
    // speed; run = 1, raster order: profiles/r03_warp/README.md).  `run` is a power of two, the grid a multiple of 8 run.
    // run == 0 (many jobs with small destinations, e.g. 256 frames of 1241 x 376): a JOB per XCD -- XCD k walks the jobs k, k + 8, ...
    // tile by tile, so every source row a job's strips share is fetched into ONE L2 once (raster order dealt the tiles of a frame over all
    // eight L2s: 1.37 times the algorithmic bytes).  The grid is one-dimensional then: 8-aligned jobs x n_tiles blocks, the rest behind.
    int job, t;
    if (run == 0) {
        const unsigned b = blockIdx.x, n_jobs = (unsigned)n_jobs_arg, full = (n_jobs & ~7u) * (unsigned)n_tiles;
        if (b < full) { const unsigned k = b >> 3, g = k / (unsigned)n_tiles; t = (int)(k - g * (unsigned)n_tiles); job = (int)(g * 8u + (b & 7u)); }
        else          { const unsigned r = b - full, g = r / (unsigned)n_tiles; t = (int)(r - g * (unsigned)n_tiles); job = (int)((n_jobs & ~7u) + g); }
        if (job >= (int)n_jobs) return;
    } else {
        const int b = blockIdx.x, r = b & (8 * run - 1);
        t = (b - r) + (r & 7) * run + (r / 8);
        job = blockIdx.y;
    }
    const WarpJobDev& jb = jobs[job];
    if (t >= n_tiles) return;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.y);
    strip_tile_body<0>(jb, nullptr, nullptr, nullptr, t, tiles_x, lds_bytes, s_dyn, wv, wv);
}

// the fused stitch on strips: a.job[1] (right image) through the strip machinery, a.job[0] (left image) in its translation form sl
// (the records are kernel arguments: everything that takes them is inlined, so they stay in the kernarg segment and are read with scalar
// loads -- a noinline callee would make every wave copy them to scratch, and a copy to device memory per call costs more than the kernel)
struct StitchDev { StitchArgs a; StitchLeft sl; };
__global__ __launch_bounds__(64) void stitch_strip_kernel(const StitchDev sd, int tiles_x, int n_tiles, int lds_bytes, int run)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
    // ONE wave per workgroup: a stitch is a single short launch (a 6000 x 3200 canvas is 19 000 strips, under four rounds of the chip's
    // wave slots), and a four-wave workgroup only retires when its slowest strip does -- the strips of a tile differ by up to 10x (both
    // images, one, none).  Blocks bq, bq + 8 share an XCD; the four strips of a tile are the blocks 8 i + x, i = 4 k .. 4 k + 3, of XCD x
    const int bq = blockIdx.x, wv = (bq >> 3) & 3, b = (bq & 7) + 8 * (bq >> 5), r = b & (8 * run - 1);
    const int t = (b - r) + (r & 7) * run + (r / 8);
    if (t >= n_tiles) return;
    strip_tile_body<1>(sd.a.job[1], &sd.a.job[0], &sd.sl, nullptr, t, tiles_x, lds_bytes, s_dyn, wv, 0);
}

// the fused paste on strips: one wave per workgroup like the stitch (strips outside the image cost nothing, the others differ)
struct PasteDev { WarpJobDev jb; PasteStrip ps; };
__global__ __launch_bounds__(64) void paste_strip_kernel(const PasteDev pd, int tiles_x, int n_tiles, int lds_bytes, int run)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
    const int bq = blockIdx.x, wv = (bq >> 3) & 3, b = (bq & 7) + 8 * (bq >> 5), r = b & (8 * run - 1);
    const int t = (b - r) + (r & 7) * run + (r / 8);
    if (t >= n_tiles) return;
    strip_tile_body<2>(pd.jb, nullptr, nullptr, &pd.ps, t, tiles_x, lds_bytes, s_dyn, wv, 0);
}

// MAPS = 0: (u, v) read from two f32 planes (cv::remap, reference src/utils.cpp:191).
// MAPS = 1 / 2: (u, v) generated in the kernel -- utils::compute_spherical_warping / compute_cylindrical_warping
// (reference src/utils.cpp:125-187, :203-265) fused with warp_local.  The maps are separable up to
// one product: xf depends on the column, yf on the row, so the host evaluates the w + h sin/cos pairs with the
// same libm the reference's CPU path calls (tables tx0/tx1 per column, ty0/ty1 per row, f64) and the kernel does
// the per-pixel part in the reference's exact order:
//   spherical   xhat = (float)(sin xf * cos yf), yhat = (float)sin yf, zhat = (float)(cos xf * cos yf)
//   cylindrical xhat = (float)sin th,            yhat = height,        zhat = (float)cos th
//   xhat /= zhat; yhat /= zhat;  u = xhat * f + 0.5 w;  v = yhat * f + 0.5 h      (all f32, no FMA)
// which replaces six full-image f32 passes and the 8 B/px map read by 16 (w + h) bytes of tables.
// cvRound(map * INTER_TAB_SIZE): f32 multiply, round half to even; NaN, infinities and out-of-range products give
// x86's "integer indefinite" 0x80000000 there (cvtss2si), not the saturated value of v_cvt_i32_f32
__device__ __forceinline__ Tap map_tap_uv(float u, float v)
{
    const float us = u * 32.f, vs = v * 32.f;
    const int X = (us >= -2147483648.f && us < 2147483648.f) ? __float2int_rn(us) : (int)0x80000000;
    const int Y = (vs >= -2147483648.f && vs < 2147483648.f) ? __float2int_rn(vs) : (int)0x80000000;
    return tap_from_fixed(X, Y);
}

// (u, v) of one pixel from its column table entries (c0 = tx0[x], c1 = tx1[x]) and row table entries (r0 = ty0[y], r1 = ty1[y]), in
// the reference's exact order (see above)
template <int MAPS>
__device__ __forceinline__ Tap map_tap_tables(const RemapArgs& a, double c0, double c1, double r0, double r1)
{
    float xhat, yhat, zhat;
    if (MAPS == 1) { xhat = (float)(c0 * r1); yhat = (float)r0; zhat = (float)(c1 * r1); }
    else           { xhat = (float)c0;        yhat = (float)r0; zhat = (float)c1; }
    xhat /= zhat; yhat /= zhat;
    return map_tap_uv(xhat * a.f + a.half_w, yhat * a.f + a.half_h);
}

template <int MAPS>
__device__ __forceinline__ Tap map_tap(const RemapArgs& a, const float* __restrict__ mx, const float* __restrict__ my, int x, int y)
{
    if (MAPS == 0) return map_tap_uv(mx[x], my[x]);
    return map_tap_tables<MAPS>(a, a.tx0[x], a.tx1[x], a.ty0[y], MAPS == 1 ? a.ty1[y] : 0.0);
}

template <int TYPE, int MAPS>
__global__ __launch_bounds__(256) void remap_kernel(const RemapArgs a)
{
    using P = PixelTraits<TYPE>;
    using WT = WaveTile<P::PXT>;
    const int x = (blockIdx.x * WT::LX + (threadIdx.x & (WT::LX - 1))) * P::PXT;
    const int y = (blockIdx.y * 4 + threadIdx.y) * WT::LY + (threadIdx.x >> WT::LOG_LX);
    if (x >= a.dw || y >= a.dh) return;
    const float* mx = MAPS ? nullptr : reinterpret_cast<const float*>(reinterpret_cast<const uint8_t*>(a.mapx) + (long long)y * a.map_stride);
    const float* my = MAPS ? nullptr : reinterpret_cast<const float*>(reinterpret_cast<const uint8_t*>(a.mapy) + (long long)y * a.map_stride);
    Tap t[P::PXT];
    bool have_taps = false;
    if constexpr (MAPS == 0 && P::PXT == 4) {
        // the map values of the lane's four pixels as one 16-byte load per plane (a vector-memory instruction costs the texture path
        // the same ~16 clocks whatever its width, tools/ubench/vmem_rate.hip) where the rows allow it
        if (x + 3 < a.dw && ((reinterpret_cast<uintptr_t>(mx) | reinterpret_cast<uintptr_t>(my)) & 15) == 0) {
            const float4 ux = *reinterpret_cast<const float4*>(mx + x), vy = *reinterpret_cast<const float4*>(my + x);
            const float u4[4] = {ux.x, ux.y, ux.z, ux.w}, v4[4] = {vy.x, vy.y, vy.z, vy.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) t[i] = map_tap_uv(u4[i], v4[i]);
            have_taps = true;
        }
    }
    if constexpr (MAPS != 0 && P::PXT == 4) {
        // in-kernel maps: the column tables of the lane's four pixels as two 16-byte loads each instead of four 8-byte ones
        if (x + 3 < a.dw && ((reinterpret_cast<uintptr_t>(a.tx0) | reinterpret_cast<uintptr_t>(a.tx1)) & 15) == 0) {
            const double2 c0a = *reinterpret_cast<const double2*>(a.tx0 + x), c0b = *reinterpret_cast<const double2*>(a.tx0 + x + 2);
            const double2 c1a = *reinterpret_cast<const double2*>(a.tx1 + x), c1b = *reinterpret_cast<const double2*>(a.tx1 + x + 2);
            const double t0[4] = {c0a.x, c0a.y, c0b.x, c0b.y}, t1[4] = {c1a.x, c1a.y, c1b.x, c1b.y};
            const double r0 = a.ty0[y], r1 = MAPS == 1 ? a.ty1[y] : 0.0;
#pragma unroll
            for (int i = 0; i < 4; ++i) t[i] = map_tap_tables<MAPS>(a, t0[i], t1[i], r0, r1);
            have_taps = true;
        }
    }
    if (!have_taps) {
#pragma unroll
        for (int i = 0; i < P::PXT; ++i) {
            t[i] = tap_from_fixed(0, 0);
            if (x + i < a.dw) t[i] = map_tap<MAPS>(a, mx, my, x + i, y);
        }
    }
    if (P::PXT == 4) quad_8u<TYPE>((gcu8)a.src, a.sw, a.sh, a.sstride, (gu8)a.dst, a.dw, a.dstride, x, y, t);
    else             single_px<TYPE>((gcu8)a.src, a.sw, a.sh, a.sstride, (gu8)a.dst, a.dstride, x, y, t[0]);
}

// ---- fused stitch compose ------------------------------------------------------------------------------------------
// reconstruct::Stitch::process after align_pair, reference src/stitch.cpp:72-80: warpPerspective(right, T * Hinv),
// warpPerspective(left, T), addWeighted(left, alpha, right, beta, gamma) -- three full-canvas passes (3 r + 3 w, 3 r + 3 w,
// 6 r + 3 w bytes per pixel) as one: a lane owns 4 adjacent output pixels, samples both sources with the 8UC3 machinery
// above (each with its own job record: inverse map, block split, fast-coordinate flag) and blends
// saturate(cvRound(l * alpha + r * beta + gamma)) in f32 as cv::addWeighted does for 8-bit data.

template <bool GENERIC>
__global__ __launch_bounds__(256) void stitch_compose_kernel(const StitchArgs a)
{
    using WT = WaveTile<4>;
    const int x = (blockIdx.x * WT::LX + (threadIdx.x & (WT::LX - 1))) * 4;
    const int y = (blockIdx.y * 4 + threadIdx.y) * WT::LY + (threadIdx.x >> WT::LOG_LX);
    if (x >= a.dw || y >= a.dh) return;
    uint32_t acc[2][3][4];                               // [source][channel][pixel] accumulators, result in byte 2
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const WarpJobDev& jb = a.job[s];
        double M[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) M[i] = jb.M[i];
        Tap t[4];
        persp_taps<4, GENERIC>(M, jb.bw0, jb.bw0_shift, jb.fast, x, y, t);
        gcu8 src = (gcu8)jb.src;
        const bool src_aligned = ((reinterpret_cast<uintptr_t>(jb.src) | (uintptr_t)jb.sstride) & 3) == 0 &&
                                 jb.sstride < (1 << 24) && jb.sh < (1 << 24) && (long long)jb.sstride * jb.sh < (1LL << 31);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[s][0][i] = acc[s][1][i] = acc[s][2][i] = 0u;
            if (x + i < a.dw) pixel_8uc3(src, jb.sw, jb.sh, jb.sstride, src_aligned, t[i], acc[s][0][i], acc[s][1][i], acc[s][2][i]);
        }
    }
    uint32_t o[12];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c) o[3 * i + c] = add_weighted_u8(byte2(acc[0][c][i]), byte2(acc[1][c][i]), a.alpha, a.beta, a.gamma);
    gu8 row = (gu8)a.dst + (long long)y * a.dstride + (long long)x * 3;
    if (x + 3 < a.dw && ((reinterpret_cast<uintptr_t>(a.dst) | (uintptr_t)a.dstride) & 3) == 0) {
        DR3_G uint32_t* q = gcast<uint32_t>(row);
#pragma unroll
        for (int d = 0; d < 3; ++d) st_out(q + d, o[4 * d] | (o[4 * d + 1] << 8) | (o[4 * d + 2] << 16) | (o[4 * d + 3] << 24));
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (x + i < a.dw) { row[3 * i] = (uint8_t)o[3 * i]; row[3 * i + 1] = (uint8_t)o[3 * i + 1]; row[3 * i + 2] = (uint8_t)o[3 * i + 2]; }
    }
}

// ---- fused panorama paste -----------------------------------------------------------------
// reconstruct::Panorama::add_img_to_canvas, reference src/panorama.cpp:178-212, in one pass over the
// bounding-box columns of the canvas: the reference builds a (B,G,R,1) 32FC4 copy of the image, warps it
// onto a canvas-sized 32FC4 image (:192), then walks the columns [min_x, max_x) scaling BGR by the feather
// ramp, setting alpha to the ramp, dropping pixels whose scaled BGR is exactly 0 and adding the rest to the
// canvas.  Here one lane warps one canvas pixel straight from the 8UC3 image (u8 -> f32 is exact, so the
// taps equal the 32FC4 copy's), with the 32FC4 arithmetic of the warp (f32 table weights, 4-term sum, no
// FMA), multiplies by ramp[x - min_x] and does the 16-byte read-modify-write.  Bytes per canvas pixel in
// the box: 3 read (image) + 16 + 16 (canvas) instead of 16 + 16 (warp) + 32 + 32 (feather pass).
#ifndef DR3_PANO_LOG
#define DR3_PANO_LOG 6
#endif
template <bool GENERIC>
__global__ __launch_bounds__(256) void panorama_add_kernel(const PanoAddArgs a)
{
    // wave tile (1 << DR3_PANO_LOG) x (64 >> DR3_PANO_LOG) canvas pixels, 4 waves stacked in y
    const int x = a.cx0 + blockIdx.x * (1 << DR3_PANO_LOG) + (threadIdx.x & ((1 << DR3_PANO_LOG) - 1));
    const int y = a.cy0 + (blockIdx.y * 4 + threadIdx.y) * (64 >> DR3_PANO_LOG) + (threadIdx.x >> DR3_PANO_LOG);
    if (x >= a.cx1 || y >= a.cy1) return;
    double M[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) M[i] = a.M[i];
    Tap t[1];
    persp_taps<1, GENERIC>(M, a.bw0, a.bw0_shift, a.fast, x, y, t);
    const int sx = t[0].sx, sy = t[0].sy, sw = a.w, sh = a.h;
    if (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0) return;      // every tap is border 0: scaled BGR == 0
    gcu8 src = (gcu8)a.img;
    const bool src_aligned = ((reinterpret_cast<uintptr_t>(a.img) | (uintptr_t)a.istride) & 3) == 0 &&
                             a.istride < (1 << 24) && sh < (1 << 24) && (long long)a.istride * sh < (1LL << 31);
    uint32_t pB, pG, pR;          // [top-left, top-right, bottom-left, bottom-right] bytes per channel
    if (src_aligned && interior_8uc3(t[0], sw, sh)) {
        const Rows3 r = load_8uc3(src, a.istride, t[0]);
        const uint32_t t0 = __builtin_amdgcn_alignbyte(r.a1, r.a0, r.o), t1 = __builtin_amdgcn_alignbyte(r.a2, r.a1, r.o);
        const uint32_t u0 = __builtin_amdgcn_alignbyte(r.b1, r.b0, r.o), u1 = __builtin_amdgcn_alignbyte(r.b2, r.b1, r.o);
        pB = __builtin_amdgcn_perm(u0, t0, 0x07040300u);
        const uint32_t gr0 = __builtin_amdgcn_perm(t1, t0, 0x05020401u), gr1 = __builtin_amdgcn_perm(u1, u0, 0x05020401u);   // [G0 G1 R0 R1]
        pG = __builtin_amdgcn_perm(gr1, gr0, 0x05040100u);
        pR = __builtin_amdgcn_perm(gr1, gr0, 0x07060302u);
    } else {
        pB = quad_checked(src, sw, sh, a.istride, sx, sy, 3, 0);
        pG = quad_checked(src, sw, sh, a.istride, sx, sy, 3, 1);
        pR = quad_checked(src, sw, sh, a.istride, sx, sy, 3, 2);
    }
    float w[4]; weights_32f(t[0].fx, t[0].fy, w);
    const float r = a.ramp[x - a.min_x];
    float o[3];
    const uint32_t p[3] = {pB, pG, pR};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v0 = (float)(p[c] & 0xffu), v1 = (float)((p[c] >> 8) & 0xffu), v2 = (float)((p[c] >> 16) & 0xffu), v3 = (float)(p[c] >> 24);
        o[c] = (v0 * w[0] + v1 * w[1] + v2 * w[2] + v3 * w[3]) * r;
    }
    if (o[0] == 0.f && o[1] == 0.f && o[2] == 0.f) return;               // :205-207
    // a.canvas addresses row a.slab_y0 of the canvas (a GPU may own just a slab of rows)
    DR3_G f32x4* cp = reinterpret_cast<DR3_G f32x4*>((gu8)reinterpret_cast<uint8_t*>(a.canvas) + (long long)(y - a.slab_y0) * a.cstride) + x;
    f32x4 c4 = *cp;
    c4.x += o[0]; c4.y += o[1]; c4.z += o[2]; c4.w += r;
    *cp = c4;
}

// Panorama::normalize_canvas, reference src/panorama.cpp:215-229: vec[c] = (int)canvas[c] / alpha_sum
// (int cast first, f32 divide, then narrowing to uchar), pixels with alpha_sum <= 0 stay 0.
__global__ __launch_bounds__(256) void panorama_normalize_kernel(const PanoNormArgs a)
{
    // A wave owns 256 canvas pixels of one row.  Lane i reads pixels xw + 64 j + i (j = 0..3): four fully coalesced
    // 1 KiB float4 loads; the finished [B G R 0] dwords are transposed through 1 KiB of LDS per wave (a wave's LDS
    // operations complete in order, no barrier) so that lane i stores the 12 contiguous bytes of pixels 4 i .. 4 i + 3.
    __shared__ __attribute__((aligned(16))) uint32_t lds[4][256];
    const int lane = threadIdx.x, wv = threadIdx.y;
    const int xw = blockIdx.x * 256;
    const int y = blockIdx.y * 4 + wv;
    if (xw >= a.cw || y >= a.ch) return;                                  // wave-uniform
    const DR3_G f32x4* cp = reinterpret_cast<const DR3_G f32x4*>((gcu8)reinterpret_cast<const uint8_t*>(a.canvas) + (long long)y * a.cstride) + xw;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = 64 * j + lane;
        uint32_t px = 0u;
        if (xw + i < a.cw) {
            const f32x4 c4 = ld_once(cp + i);
            if (c4.w > 0.0f) {
                const uint32_t b = (uint32_t)(int)((float)(int)c4.x / c4.w) & 0xffu;
                const uint32_t g = (uint32_t)(int)((float)(int)c4.y / c4.w) & 0xffu;
                const uint32_t r = (uint32_t)(int)((float)(int)c4.z / c4.w) & 0xffu;
                px = b | (g << 8) | (r << 16);
            }
        }
        lds[wv][i] = px;
    }
    __builtin_amdgcn_wave_barrier();
    const uint4 d = reinterpret_cast<const uint4*>(lds[wv])[lane];
    const int x = xw + 4 * lane;
    if (x >= a.cw) return;
    gu8 row = (gu8)a.out + (long long)y * a.ostride + (long long)x * 3;
    if (x + 3 < a.cw && ((reinterpret_cast<uintptr_t>(a.out) | (uintptr_t)a.ostride) & 3) == 0) {
        DR3_G uint32_t* o = gcast<uint32_t>(row);
        st_out(o, __builtin_amdgcn_perm(d.y, d.x, 0x04020100u));
        st_out(o + 1, __builtin_amdgcn_perm(d.z, d.y, 0x05040201u));
        st_out(o + 2, __builtin_amdgcn_perm(d.w, d.z, 0x06050402u));
    } else {
        const uint32_t px[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (x + i < a.cw) { row[3 * i] = (uint8_t)px[i]; row[3 * i + 1] = (uint8_t)(px[i] >> 8); row[3 * i + 2] = (uint8_t)(px[i] >> 16); }
    }
}

// ---- 2x2 box half-sampler ---------------------------------------------------------------------
typedef uint32_t u32x2_unaligned __attribute__((ext_vector_type(2), aligned(1)));
__device__ __forceinline__ uint32_t half_px(int a, int b, int c, int d, int mode)
{
    if (mode == DR3HIP_HALF_SSE2) {                       // _mm_avg_epu8 then _mm_avg_epu16
        const int v0 = (a + c + 1) >> 1, v1 = (b + d + 1) >> 1;
        return (uint32_t)((v0 + v1 + 1) >> 1);
    }
    return (uint32_t)((a + b + c + d) >> 2);              // (a+b+c+d)/4, all non-negative
}

// The streaming form (rows 16-byte aligned, the truncating or the SSE2 rounding, 512 output pixels or more per row): a wave owns 1024
// output pixels of one row; lane i reads bytes 16 i .. 16 i + 15 of the first KiB of both source rows and the same of the second KiB
// (four fully coalesced 1-KiB loads, read once: non-temporal), averages in 16-bit halves of a dword -- even and odd bytes of a row are
// the left and right pixels of two output pixels -- and stores 8 output bytes per segment (two coalesced 512-byte stores).
__device__ __forceinline__ uint32_t half_pairs(uint32_t t, uint32_t b, int mode)      // bytes [p0 p1 p2 p3] of two rows -> 16-bit halves [o0, o1]
{
    const uint32_t m = 0x00ff00ffu, te = t & m, to = (t >> 8) & m, be = b & m, bo = (b >> 8) & m;
    if (mode == DR3HIP_HALF_SSE2) {                       // _mm_avg_epu8 of the rows, then _mm_avg_epu16 of the even and odd bytes
        const uint32_t ve = ((te + be + 0x00010001u) >> 1) & m, vo = ((to + bo + 0x00010001u) >> 1) & m;
        return ((ve + vo + 0x00010001u) >> 1) & m;
    }
    return ((te + to + be + bo) >> 2) & m;                // (a + b + c + d) / 4
}
#ifndef DR3_HALF_SEGS
#define DR3_HALF_SEGS 2
#endif
__global__ __launch_bounds__(256) void halfsample_wide_kernel(const HalfArgs a)
{
    const int lane = threadIdx.x;
    const int Y = blockIdx.y * 4 + threadIdx.y;
    if (Y >= a.oh) return;
    gcu8 src = (gcu8)(a.src + (long long)blockIdx.z * a.sstep + (long long)(2 * Y) * a.sstride);
    gu8 drow = (gu8)(a.dst + (long long)blockIdx.z * a.dstep + (long long)Y * a.dstride);
#pragma unroll
    for (int seg = 0; seg < DR3_HALF_SEGS; ++seg) {
        const int X = blockIdx.x * (512 * DR3_HALF_SEGS) + 512 * seg + 8 * lane;        // 8 output pixels
        if (X >= a.ow) continue;
        gcu8 r0 = src + 2 * X, r1 = r0 + a.sstride;
        if (X + 8 <= a.ow) {
            const u32x4 t = ld_once(gcast<u32x4>(r0)), b = ld_once(gcast<u32x4>(r1));
            const uint32_t h0 = half_pairs(t.x, b.x, a.mode), h1 = half_pairs(t.y, b.y, a.mode), h2 = half_pairs(t.z, b.z, a.mode), h3 = half_pairs(t.w, b.w, a.mode);
            u32x2 o;
            o.x = __builtin_amdgcn_perm(h1, h0, 0x06040200u); o.y = __builtin_amdgcn_perm(h3, h2, 0x06040200u);
            st_out(gcast<u32x2>(drow + X), o);
        } else {
            for (int k = 0; X + k < a.ow; ++k) drow[X + k] = (uint8_t)half_px(r0[2 * k], r0[2 * k + 1], r1[2 * k], r1[2 * k + 1], a.mode);
        }
    }
}

__global__ __launch_bounds__(256) void halfsample_kernel(const HalfArgs a)
{
    const int X = (blockIdx.x * 64 + threadIdx.x) * 4;    // 4 output pixels per lane
    const int Y = blockIdx.y * 4 + threadIdx.y;
    if (X >= a.ow || Y >= a.oh) return;
    const uint8_t* __restrict__ src = a.src + (long long)blockIdx.z * a.sstep;      // grid z = frame slot
    uint8_t* __restrict__ dst = a.dst + (long long)blockIdx.z * a.dstep;
    uint32_t o[4] = {0, 0, 0, 0};
    // 4 output pixels of a lane = 8 source bytes of two rows: one 8-byte load per row at whatever alignment the plane has (the
    // half-sampled levels are continuous, their pitch is the odd width they happen to have)
    auto quad = [&](const uint8_t* r0, const uint8_t* r1, int mode) {
        const u32x2_unaligned t = *reinterpret_cast<const u32x2_unaligned*>(r0), b = *reinterpret_cast<const u32x2_unaligned*>(r1);
        const uint32_t tw[2] = {t.x, t.y}, bw[2] = {b.x, b.y};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t tt = tw[k >> 1] >> (16 * (k & 1)), bb = bw[k >> 1] >> (16 * (k & 1));
            o[k] = half_px(tt & 0xff, (tt >> 8) & 0xff, bb & 0xff, (bb >> 8) & 0xff, mode);
        }
    };
    if (a.mode == DR3HIP_HALF_REF_SCALAR) {
        // reference pointer walk: flat offset of output (Y, X) is Y*(2*ow + stride) + 2*X, where stride is the
        // step of the cv::Mat the reference holds (a.lstride); the bytes may sit in a wider-pitched plane here
        const int ls = a.lstride;
        const long long total = (long long)ls * a.h;
        // element i of the continuous walk = (row i / ls, column i % ls) of the pitched plane; i + 1 and i + ls follow from
        // (row, column) by a carry
        auto split = [&](long long t, long long& r, int& c) {
            if (total < (1LL << 31)) { const uint32_t q = (uint32_t)t / (uint32_t)ls; r = q; c = (int)((uint32_t)t - q * (uint32_t)ls); }
            else { r = t / ls; c = (int)(t - r * ls); }
        };
        const long long t0 = (long long)Y * (2LL * a.ow + ls) + 2LL * X;
        long long r; int c;
        split(t0, r, c);
        if (X + 3 < a.ow && c + 8 <= ls && t0 + ls + 8 <= total) {
            // the lane's 8 bytes of the walk and the 8 bytes one stride further exist and stay inside their rows: one division per lane
            const uint8_t* r0 = src + r * a.sstride + c;
            quad(r0, r0 + a.sstride, DR3HIP_HALF_TRUNC);
        } else {
            auto at = [&](long long i, long long rr, int cc) -> int { return i < total ? src[rr * a.sstride + cc] : 0; };
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (X + k < a.ow) {
                    const long long t = t0 + 2 * k;
                    split(t, r, c);
                    const bool wrap = c + 1 == ls;                                  // t + 1 starts the next row
                    const long long r1 = wrap ? r + 1 : r; const int c1 = wrap ? 0 : c + 1;
                    o[k] = half_px(at(t, r, c), at(t + 1, r1, c1), at(t + ls, r + 1, c), at(t + ls + 1, r1 + 1, c1), DR3HIP_HALF_TRUNC);
                }
            }
        }
    } else {
        const uint8_t* r0 = src + (long long)(2 * Y) * a.sstride + 2 * X;
        const uint8_t* r1 = r0 + a.sstride;
        if (X + 3 < a.ow) {
            quad(r0, r1, a.mode);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (X + k < a.ow) o[k] = half_px(r0[2 * k], r0[2 * k + 1], r1[2 * k], r1[2 * k + 1], a.mode);
        }
    }
    uint8_t* drow = dst + (long long)Y * a.dstride;
    if (X + 3 < a.ow) {                                   // one dword store at whatever alignment the row has
        typedef uint32_t u32_unaligned __attribute__((aligned(1)));
        *reinterpret_cast<u32_unaligned*>(drow + X) = o[0] | (o[1] << 8) | (o[2] << 16) | (o[3] << 24);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) if (X + k < a.ow) drow[X + k] = (uint8_t)o[k];
    }
}

}  // namespace

hipError_t launch_warp(const WarpJobDev* d_jobs, int n_jobs, int max_dw, int max_dh, int type, bool generic, hipStream_t s, int tile_mode, int tile_lds)
{
    if (n_jobs <= 0 || max_dw <= 0 || max_dh <= 0) return hipSuccess;
    const int pxt = (type == DR3HIP_8UC1 || type == DR3HIP_8UC3) ? 4 : 1;
    dim3 block(64, 4);
#ifndef DR3_WARP_NO_TILE
    if (type == DR3HIP_8UC3 && !generic && tile_mode == 2) {     // LDS-staged 64 x 64 tiles; jobs are grid y
        const int tiles_x = (max_dw + kTileW - 1) / kTileW;
        const long long n_tiles = (long long)tiles_x * ((max_dh + kTileH - 1) / kTileH);
        if (n_tiles < (1LL << 30)) {
            // many jobs of small destinations: a job per XCD (run 0; the job count rides in the kernel's n_jobs argument, the grid is linear)
#ifdef DR3_WARP_NO_PER_XCD
            const bool per_xcd = false;                                        // experiment builds: the tile-per-XCD order of round 3
#else
            const bool per_xcd = n_jobs >= 16 && n_tiles <= 1024 && n_tiles * (long long)n_jobs < (1LL << 30);
#endif
            const int run = per_xcd ? 0 : tiles_x >= 32 ? 4 : 1, grp = 8 * (run ? run : 1);
            dim3 tgrid(per_xcd ? (unsigned)(n_tiles * n_jobs) : (unsigned)((n_tiles + grp - 1) / grp * grp), per_xcd ? 1 : n_jobs, 1);
            int lds = (std::max(tile_lds, 1024) + 255) & ~255;               // a wave's slice; behind it the wave's tie list (256 B)
#ifdef DR3_WARP_EXPERIMENT
            if (const char* e = getenv("DR3HIP_TILE_LDS")) lds = std::min((std::max(atoi(e), 1024) + 255) & ~255, kWarpTileLdsBytes);      // slice-size sweeps: experiment builds only
#endif
            hipLaunchKernelGGL(warp_tile_8uc3_kernel, tgrid, block, 4 * (lds + 256), s, d_jobs, tiles_x, (int)n_tiles, lds, run, n_jobs);
            return hipGetLastError();
        }
    }
#endif
    const int tw = pxt == 4 ? WaveTile<4>::LX * 4 : WaveTile<1>::LX, th = 4 * (pxt == 4 ? WaveTile<4>::LY : WaveTile<1>::LY);
    dim3 grid((max_dw + tw - 1) / tw, (max_dh + th - 1) / th, n_jobs);
#define DR3_WARP_CASE(T) \
    case T: if (generic) hipLaunchKernelGGL((warp_perspective_kernel<T, true>),  grid, block, 0, s, d_jobs); \
            else         hipLaunchKernelGGL((warp_perspective_kernel<T, false>), grid, block, 0, s, d_jobs); break;
    switch (type) {
    DR3_WARP_CASE(DR3HIP_8UC1) DR3_WARP_CASE(DR3HIP_8UC3) DR3_WARP_CASE(DR3HIP_8UC4)
    DR3_WARP_CASE(DR3HIP_32FC1) DR3_WARP_CASE(DR3HIP_32FC4)
    default: return hipErrorInvalidValue;
    }
#undef DR3_WARP_CASE
    return hipGetLastError();
}

hipError_t launch_remap(const RemapArgs& a, int type, hipStream_t s)
{
    if (a.dw <= 0 || a.dh <= 0) return hipSuccess;
    const int pxt = (type == DR3HIP_8UC1 || type == DR3HIP_8UC3) ? 4 : 1;
    dim3 block(64, 4);
    const int tw = pxt == 4 ? WaveTile<4>::LX * 4 : WaveTile<1>::LX, th = 4 * (pxt == 4 ? WaveTile<4>::LY : WaveTile<1>::LY);
    dim3 grid((a.dw + tw - 1) / tw, (a.dh + th - 1) / th, 1);
#define DR3_REMAP_CASE(T) \
    case T: if (a.mode == 0)      hipLaunchKernelGGL((remap_kernel<T, 0>), grid, block, 0, s, a); \
            else if (a.mode == 1) hipLaunchKernelGGL((remap_kernel<T, 1>), grid, block, 0, s, a); \
            else                  hipLaunchKernelGGL((remap_kernel<T, 2>), grid, block, 0, s, a); break;
    switch (type) {
    DR3_REMAP_CASE(DR3HIP_8UC1) DR3_REMAP_CASE(DR3HIP_8UC3) DR3_REMAP_CASE(DR3HIP_8UC4)
    DR3_REMAP_CASE(DR3HIP_32FC1) DR3_REMAP_CASE(DR3HIP_32FC4)
    default: return hipErrorInvalidValue;
    }
#undef DR3_REMAP_CASE
    return hipGetLastError();
}

// The strip form of the fused stitch (stitch_strip_kernel): the right image's job runs on the strip kernel (WarpJobDev::tile == 2), the
// left image's map is EXACTLY a translation (what translate_top_left_mat is, reference src/stitch.cpp:70-74: the inverse of [1 0 -mx;
// 0 1 -my; 0 0 1] in double is [1 0 mx; 0 1 my; 0 0 1] bit for bit), its rows are 16-byte aligned, and both boxes of a strip fit the
// LDS slice of a wave.  Everything else takes the direct form (stitch_compose_kernel).
static bool stitch_strip_plan(const StitchArgs& a, StitchLeft& sl, int& lds)
{
#ifdef DR3_STITCH_NO_STRIPS
    return false;
#endif
    const WarpJobDev& jl = a.job[0];
    const WarpJobDev& jr = a.job[1];
    if (jr.tile != 2 || jr.tile_lds <= 0) return false;
    const double* M = jl.M;
    if (!(M[0] == 1.0 && M[1] == 0.0 && M[3] == 0.0 && M[4] == 1.0 && M[6] == 0.0 && M[7] == 0.0 && M[8] == 1.0)) return false;
    if (!(std::fabs(M[2]) < 30000.0 && std::fabs(M[5]) < 30000.0) || a.dw > 30000 || a.dh > 30000) return false;      // taps stay inside saturate_cast<short>
    if ((((uintptr_t)jl.src | (uintptr_t)jl.sstride) & 15) || jl.sstride >= (1 << 24) || jl.sh >= (1 << 24) || (long long)jl.sstride * jl.sh >= (1LL << 31)) return false;
    // OpenCV: X = saturate_cast<int>((X0 + M0 * x1) * (32 / W)) = rint(32 x + 32 tx): 32 x is an even integer, the f64 operations are
    // exact, so every pixel rounds like x = 0 does (round half to even): X = 32 x + K
    const long long Kx = (long long)std::nearbyint(32.0 * M[2]), Ky = (long long)std::nearbyint(32.0 * M[5]);
    sl.src = jl.src; sl.sw = jl.sw; sl.sh = jl.sh; sl.sstride = jl.sstride;
    sl.ix = (int)(Kx >> 5); sl.iy = (int)(Ky >> 5);
    const uint32_t fx = (uint32_t)(Kx & 31), fy = (uint32_t)(Ky & 31);          // blend8_weights on the host
    sl.wlo = (32u - fx) | (fx << 8); sl.whi = sl.wlo << 16; sl.wy0 = (32u - fy) << 6; sl.wy1 = fy << 6; sl.fx = (int)fx; sl.fy = (int)fy;
    sl.alpha = a.alpha; sl.beta = a.beta; sl.gamma = a.gamma;
    sl.avg = (a.alpha == 0.5f && a.beta == 0.5f && a.gamma == 0.0f) ? 1 : 0;
    lds = (std::max(jr.tile_lds, 1024) + 255) & ~255;
    return lds <= kWarpTileLdsBytes;
}

bool stitch_takes_strips(const StitchArgs& a) { StitchLeft sl; int lds = 0; return stitch_strip_plan(a, sl, lds); }

hipError_t launch_stitch_compose(const StitchArgs& a, hipStream_t s)
{
    if (a.dw <= 0 || a.dh <= 0) return hipSuccess;
    dim3 block(64, 4);
    StitchDev sd;
    int lds = 0;
    if (stitch_strip_plan(a, sd.sl, lds)) {
        sd.a = a;
        const int tiles_x = (a.dw + kTileW - 1) / kTileW;
        const long long n_tiles = (long long)tiles_x * ((a.dh + kTileH - 1) / kTileH);
        if (n_tiles < (1LL << 30)) {
            const int run = tiles_x >= 32 ? 4 : 1, grp = 8 * run;
            dim3 tgrid((unsigned)((n_tiles + grp - 1) / grp * grp * 4), 1, 1);
            hipLaunchKernelGGL(stitch_strip_kernel, tgrid, dim3(64, 1), lds + 256, s, sd, tiles_x, (int)n_tiles, lds, run);
            return hipGetLastError();
        }
    }
    const int tw = WaveTile<4>::LX * 4, th = 4 * WaveTile<4>::LY;
    dim3 grid((a.dw + tw - 1) / tw, (a.dh + th - 1) / th, 1);
    if (a.job[0].bw0_shift < 0 || a.job[1].bw0_shift < 0) hipLaunchKernelGGL(stitch_compose_kernel<true>,  grid, block, 0, s, a);
    else                                                   hipLaunchKernelGGL(stitch_compose_kernel<false>, grid, block, 0, s, a);
    return hipGetLastError();
}

// May the paste run on strips?  The plan of the box (PanoAddArgs::tile / tile_lds / tile_guard, made by the host over the box's
// pixels) says so; the strips start at the box's first column and row.
bool paste_takes_strips(const PanoAddArgs& a)
{
#ifdef DR3_PASTE_NO_STRIPS
    return false;
#endif
    return a.tile == 2 && a.tile_lds > 0 && a.fast == 3 && a.bw0_shift >= 0 && ((std::max(a.tile_lds, 1024) + 255) & ~255) <= kWarpTileLdsBytes;
}

hipError_t launch_panorama_add(const PanoAddArgs& a, hipStream_t s)
{
    if (a.cx1 <= a.cx0 || a.cy1 <= a.cy0) return hipSuccess;
    if (paste_takes_strips(a)) {
        PasteDev pd;
        WarpJobDev& j = pd.jb;
        j.src = a.img; j.sw = a.w; j.sh = a.h; j.sstride = a.istride;
        j.dst = nullptr; j.dw = a.cx1; j.dh = a.cy1; j.dstride = 0;          // the strips end at the box's last column / row
        for (int i = 0; i < 9; ++i) j.M[i] = a.M[i];
        j.bw0 = a.bw0; j.bw0_shift = a.bw0_shift; j.fast = a.fast; j.tile = 2; j.tile_lds = a.tile_lds; j.tile_guard = a.tile_guard;
        pd.ps.canvas = a.canvas; pd.ps.cstride = a.cstride; pd.ps.slab_y0 = a.slab_y0; pd.ps.ramp = a.ramp; pd.ps.min_x = a.min_x;
        pd.ps.x0 = a.cx0; pd.ps.y0 = a.cy0;
        const int lds = (std::max(a.tile_lds, 1024) + 255) & ~255;
        const int tiles_x = (a.cx1 - a.cx0 + kTileW - 1) / kTileW;
        const long long n_tiles = (long long)tiles_x * ((a.cy1 - a.cy0 + kTileH - 1) / kTileH);
        if (n_tiles < (1LL << 28)) {
            const int run = tiles_x >= 32 ? 4 : 1, grp = 8 * run;
            dim3 tgrid((unsigned)((n_tiles + grp - 1) / grp * grp * 4), 1, 1);
            hipLaunchKernelGGL(paste_strip_kernel, tgrid, dim3(64, 1), lds + 256, s, pd, tiles_x, (int)n_tiles, lds, run);
            return hipGetLastError();
        }
    }
    dim3 block(64, 4);
    const int tw = 1 << DR3_PANO_LOG, th = 4 * (64 >> DR3_PANO_LOG);
    dim3 grid((a.cx1 - a.cx0 + tw - 1) / tw, (a.cy1 - a.cy0 + th - 1) / th, 1);
    if (a.bw0_shift < 0) hipLaunchKernelGGL(panorama_add_kernel<true>,  grid, block, 0, s, a);
    else                 hipLaunchKernelGGL(panorama_add_kernel<false>, grid, block, 0, s, a);
    return hipGetLastError();
}

hipError_t launch_panorama_normalize(const PanoNormArgs& a, hipStream_t s)
{
    if (a.cw <= 0 || a.ch <= 0) return hipSuccess;
    dim3 block(64, 4);
    dim3 grid((a.cw + 255) / 256, (a.ch + 3) / 4, 1);
    hipLaunchKernelGGL(panorama_normalize_kernel, grid, block, 0, s, a);
    return hipGetLastError();
}

hipError_t launch_halfsample(const HalfArgs& a, hipStream_t s)
{
    if (a.ow <= 0 || a.oh <= 0) return hipSuccess;
    dim3 block(64, 4);
#ifndef DR3_HALF_NO_WIDE
    if (a.mode != DR3HIP_HALF_REF_SCALAR && a.ow >= 512 && (((uintptr_t)a.src | (uintptr_t)a.sstride | (uintptr_t)a.sstep) & 15) == 0 &&
        (((uintptr_t)a.dst | (uintptr_t)a.dstride | (uintptr_t)a.dstep) & 7) == 0) {
        dim3 wgrid((a.ow + 512 * DR3_HALF_SEGS - 1) / (512 * DR3_HALF_SEGS), (a.oh + 3) / 4, a.n_slots > 0 ? a.n_slots : 1);
        hipLaunchKernelGGL(halfsample_wide_kernel, wgrid, block, 0, s, a);
        return hipGetLastError();
    }
#endif
    dim3 grid((a.ow + 255) / 256, (a.oh + 3) / 4, a.n_slots > 0 ? a.n_slots : 1);
    hipLaunchKernelGGL(halfsample_kernel, grid, block, 0, s, a);
    return hipGetLastError();
}

}  // namespace dr3
