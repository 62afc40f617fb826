// lk_kernels.hip -- pyramidal Lucas-Kanade tracker (gfx950).  Replaces the tracker inside
// cv::calcOpticalFlowPyrLK as called at reference src/initialization.cpp:608-613
// (LKTrackerInvoker).
//
// Mapping: ONE wave64 per (frame pair, point); all pyramid levels run coarse -> fine inside the
// wave, the running estimate never leaves registers.  The window is cut into lane-items: item
// q = ty * rpr + r covers the 4 taps (ty, 4r .. 4r+3); lane = q % 64, pass = q / 64 (NPASS passes,
// template parameter so the patch lives in statically indexed VGPRs).  Per level each lane
//   1. extracts its taps of the template patch (I, Ix, Iy: 14-bit fixed-point bilinear, integer
//      exact) once and keeps them in registers,
//   2. per Gauss-Newton iteration fetches the matching J taps (two aligned dword loads per row,
//      funnel-shifted), forms the mismatch and the two b sums,
//   3. reduces the f32 sums across the wave with DPP adds (no LDS), solves the 2x2 system
//      redundantly in every lane and tests the (wave-uniform) exit conditions.
// Border handling: OpenCV pads every level by winSize (REFLECT_101 for images, zeros for
// derivatives); here windows that leave the image take a wave-uniform slow path that reflects
// indices instead, so no padded copy of the pyramid is ever written to HBM.
//
// Float summation order (defines bit-exactness against oracle order DR3O_ORDER_WAVE64):
// lane-local sequential over (pass, tap k), then an xor butterfly over lanes with strides
// 1,2,4,8,16,32.  The DPP sequence below (quad_perm, quad_perm, row_half_mirror, row_mirror,
// row_bcast15, row_bcast31, readlane 63) produces exactly that tree.  No FMA contraction
// (-ffp-contract=off), IEEE divide/sqrt (hipcc default).
#include "dr3hip_internal.h"

namespace dr3 {
namespace {

constexpr int W_BITS = 14;

__device__ __forceinline__ int reflect101(int p, int n)
{
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

// ---- wave64 all-reduce of one f32 ----------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v)
{
#if defined(DR3_REDUCE_SHFL)
    v += __shfl_xor(v, 1);  v += __shfl_xor(v, 2);  v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);  v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
    return v;
#else
    // symmetric exchanges inside each row of 16 lanes: after these every lane of a row holds the
    // row sum, associated exactly like xor strides 1,2,4,8
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));  // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true));  // row_mirror
    // rows 1,3 += last lane of rows 0,2 ; rows 2,3 += last lane of row 1  (old = 0 for masked rows)
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xa, 0xf, false)); // row_bcast15
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xc, 0xf, false)); // row_bcast31
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
#endif
}

struct Weights { int w00, w01, w10, w11; };

__device__ __forceinline__ Weights make_weights(float a, float b)
{
    Weights W;
    W.w00 = __float2int_rn((1.f - a) * (1.f - b) * (float)(1 << W_BITS));
    W.w01 = __float2int_rn(a * (1.f - b) * (float)(1 << W_BITS));
    W.w10 = __float2int_rn((1.f - a) * b * (float)(1 << W_BITS));
    W.w11 = (1 << W_BITS) - W.w00 - W.w01 - W.w10;
    return W;
}

// 5 consecutive bytes of two image rows starting at (x, y) -> r0, r1 (byte k in bits 8k..8k+7).
// fast: whole window inside the image, aligned dword pairs + 64-bit funnel shift.
__device__ __forceinline__ void load_rows_fast(const uint8_t* __restrict__ img, int pitch, int x, int y,
                                               uint64_t& r0, uint64_t& r1)
{
    const uint8_t* p = img + (long long)y * pitch + (x & ~3);
    const int sh = (x & 3) * 8;
    const uint32_t a0 = *reinterpret_cast<const uint32_t*>(p);
    const uint32_t a1 = *reinterpret_cast<const uint32_t*>(p + 4);
    const uint32_t b0 = *reinterpret_cast<const uint32_t*>(p + pitch);
    const uint32_t b1 = *reinterpret_cast<const uint32_t*>(p + pitch + 4);
    r0 = (((uint64_t)a1 << 32) | a0) >> sh;
    r1 = (((uint64_t)b1 << 32) | b0) >> sh;
}
// slow: REFLECT_101 index per byte (window crosses the image border)
__device__ __forceinline__ void load_rows_slow(const uint8_t* __restrict__ img, int pitch, int w, int h,
                                               int x, int y, uint64_t& r0, uint64_t& r1)
{
    const uint8_t* p0 = img + (long long)reflect101(y, h) * pitch;
    const uint8_t* p1 = img + (long long)reflect101(y + 1, h) * pitch;
    r0 = 0; r1 = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int xr = reflect101(x + k, w);
        r0 |= (uint64_t)p0[xr] << (8 * k);
        r1 |= (uint64_t)p1[xr] << (8 * k);
    }
}

__device__ __forceinline__ int byte_of(uint64_t r, int k) { return (int)((r >> (8 * k)) & 0xff); }

__device__ __forceinline__ int bilin_u8(uint64_t r0, uint64_t r1, int k, const Weights& W)
{
    // CV_DESCALE(I00*iw00 + I01*iw01 + I10*iw10 + I11*iw11, W_BITS - 5)
    const int v = byte_of(r0, k) * W.w00 + byte_of(r0, k + 1) * W.w01 +
                  byte_of(r1, k) * W.w10 + byte_of(r1, k + 1) * W.w11;
    return (v + (1 << (W_BITS - 5 - 1))) >> (W_BITS - 5);
}

template <int NPASS>
__global__ __launch_bounds__(64) void lk_track_kernel(const LKArgs a)
{
    const int lane = threadIdx.x;
    const int pt = blockIdx.x, pair = blockIdx.y;
    const long long pidx = (long long)pair * a.n_pts + pt;
    const int ps = a.prev_slot ? a.prev_slot[pair] : a.first_slot + pair;
    const int ns = a.next_slot ? a.next_slot[pair] : a.first_slot + pair + 1;

    const float ppx = a.prev_xy[2 * pidx], ppy = a.prev_xy[2 * pidx + 1];
    float nx = a.next_xy[2 * pidx], ny = a.next_xy[2 * pidx + 1];

    const int win_w = a.win_w, win_h = a.win_h;
    const float half_x = (win_w - 1) * 0.5f, half_y = (win_h - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (1 << 20);

    // lane-item geometry (fixed for the whole kernel)
    int itx[NPASS], ity[NPASS];
    bool ivalid[NPASS];
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int q = p * 64 + lane;
        ivalid[p] = q < a.n_items;
        ity[p] = q / a.rpr;
        itx[p] = 4 * (q - ity[p] * a.rpr);
    }

    int   status = 1;
    float err = 0.f;

    // template patch of this lane: I (x32), Ix, Iy per tap
    int Iv[NPASS][4], Ix[NPASS][4], Iy[NPASS][4];

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
        const int ipx = __builtin_amdgcn_readfirstlane((int)floorf(px));
        const int ipy = __builtin_amdgcn_readfirstlane((int)floorf(py));

        if (ipx < -win_w || ipx >= L.w || ipy < -win_h || ipy >= L.h) {
            if (level == 0) { status = 0; err = 0.f; }
            if (a.iters && lane == 0) a.iters[pidx * (a.max_level + 1) + level] = 0;
            continue;
        }

        // ---- template patch + structure tensor ------------------------------------------------
        float A11, A12, A22;
        {
            const Weights W = make_weights(px - (float)ipx, py - (float)ipy);
            const uint8_t* __restrict__ I = L.img + (long long)ps * L.step;
            const uint8_t* __restrict__ dI = reinterpret_cast<const uint8_t*>(L.der) + (long long)ps * L.dstep;
            const bool inside = ipx >= 0 && ipy >= 0 && ipx + win_w < L.w && ipy + win_h < L.h;
            float s11 = 0.f, s12 = 0.f, s22 = 0.f;
#pragma unroll
            for (int p = 0; p < NPASS; ++p) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { Iv[p][k] = 0; Ix[p][k] = 0; Iy[p][k] = 0; }
                if (ivalid[p]) {
                    const int x = ipx + itx[p], y = ipy + ity[p];
                    uint64_t r0, r1;
                    uint32_t d0[5], d1[5];     // packed (Ix | Iy<<16) of rows y, y+1, columns x..x+4
                    if (inside) {
                        load_rows_fast(I, L.pitch, x, y, r0, r1);
                        const uint32_t* q0 = reinterpret_cast<const uint32_t*>(dI + (long long)y * L.dpitch) + x;
                        const uint32_t* q1 = reinterpret_cast<const uint32_t*>(dI + (long long)(y + 1) * L.dpitch) + x;
#pragma unroll
                        for (int k = 0; k < 5; ++k) { d0[k] = q0[k]; d1[k] = q1[k]; }
                    } else {
                        load_rows_slow(I, L.pitch, L.w, L.h, x, y, r0, r1);
#pragma unroll
                        for (int k = 0; k < 5; ++k) {
                            const int xx = x + k;
                            const bool okx = xx >= 0 && xx < L.w;
                            d0[k] = (okx && y >= 0 && y < L.h)
                                        ? reinterpret_cast<const uint32_t*>(dI + (long long)y * L.dpitch)[xx] : 0u;
                            d1[k] = (okx && y + 1 >= 0 && y + 1 < L.h)
                                        ? reinterpret_cast<const uint32_t*>(dI + (long long)(y + 1) * L.dpitch)[xx] : 0u;
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (itx[p] + k < win_w) {
                            const int ival = bilin_u8(r0, r1, k, W);
                            const int x00 = (int)(short)(d0[k] & 0xffff),     y00 = (int)d0[k] >> 16;
                            const int x01 = (int)(short)(d0[k + 1] & 0xffff), y01 = (int)d0[k + 1] >> 16;
                            const int x10 = (int)(short)(d1[k] & 0xffff),     y10 = (int)d1[k] >> 16;
                            const int x11 = (int)(short)(d1[k + 1] & 0xffff), y11 = (int)d1[k + 1] >> 16;
                            const int ixval = (x00 * W.w00 + x01 * W.w01 + x10 * W.w10 + x11 * W.w11 + (1 << (W_BITS - 1))) >> W_BITS;
                            const int iyval = (y00 * W.w00 + y01 * W.w01 + y10 * W.w10 + y11 * W.w11 + (1 << (W_BITS - 1))) >> W_BITS;
                            Iv[p][k] = ival; Ix[p][k] = ixval; Iy[p][k] = iyval;
                            s11 += (float)(ixval * ixval);
                            s12 += (float)(ixval * iyval);
                            s22 += (float)(iyval * iyval);
                        }
                    }
                }
            }
            A11 = wave_sum(s11) * FLT_SCALE;
            A12 = wave_sum(s12) * FLT_SCALE;
            A22 = wave_sum(s22) * FLT_SCALE;
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

        // ---- Gauss-Newton iterations ----------------------------------------------------------
        const uint8_t* __restrict__ J = L.img + (long long)ns * L.step;
        float pdx = 0.f, pdy = 0.f;
        for (int j = 0; j < a.max_count; ++j) {
            const int inx = __builtin_amdgcn_readfirstlane((int)floorf(cx));
            const int iny = __builtin_amdgcn_readfirstlane((int)floorf(cy));
            if (inx < -win_w || inx >= L.w || iny < -win_h || iny >= L.h) {
                if (level == 0) status = 0;
                break;
            }
            it_count = j + 1;
            const Weights W = make_weights(cx - (float)inx, cy - (float)iny);
            const bool inside = inx >= 0 && iny >= 0 && inx + win_w < L.w && iny + win_h < L.h;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int p = 0; p < NPASS; ++p) {
                if (ivalid[p]) {
                    uint64_t r0, r1;
                    if (inside) load_rows_fast(J, L.pitch, inx + itx[p], iny + ity[p], r0, r1);
                    else        load_rows_slow(J, L.pitch, L.w, L.h, inx + itx[p], iny + ity[p], r0, r1);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        // masked taps carry Ix = Iy = 0: their terms are +0.0f, an exact no-op
                        const int diff = bilin_u8(r0, r1, k, W) - Iv[p][k];
                        s1 += (float)(diff * Ix[p][k]);
                        s2 += (float)(diff * Iy[p][k]);
                    }
                }
            }
            const float b1 = wave_sum(s1) * FLT_SCALE;
            const float b2 = wave_sum(s2) * FLT_SCALE;

            const float dx = (A12 * b2 - A22 * b1) * D;
            const float dy = (A12 * b1 - A11 * b2) * D;
            cx += dx; cy += dy;
            nx = cx + half_x; ny = cy + half_y;

            const int stop_eps = ((double)dx * (double)dx + (double)dy * (double)dy <= a.epsilon) ? 1 : 0;
            if (__builtin_amdgcn_readfirstlane(stop_eps)) break;
            const int stop_osc = (j > 0 && fabs((double)(dx + pdx)) < 0.01 && fabs((double)(dy + pdy)) < 0.01) ? 1 : 0;
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
            const int inx = __builtin_amdgcn_readfirstlane((int)floorf(fx));
            const int iny = __builtin_amdgcn_readfirstlane((int)floorf(fy));
            if (inx < -win_w || inx >= L.w || iny < -win_h || iny >= L.h) {
                status = 0;
                continue;
            }
            const Weights W = make_weights(fx - (float)inx, fy - (float)iny);
            const bool inside = inx >= 0 && iny >= 0 && inx + win_w < L.w && iny + win_h < L.h;
            float se = 0.f;
#pragma unroll
            for (int p = 0; p < NPASS; ++p) {
                if (ivalid[p]) {
                    uint64_t r0, r1;
                    if (inside) load_rows_fast(J, L.pitch, inx + itx[p], iny + ity[p], r0, r1);
                    else        load_rows_slow(J, L.pitch, L.w, L.h, inx + itx[p], iny + ity[p], r0, r1);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (itx[p] + k < win_w) {
                            const int diff = bilin_u8(r0, r1, k, W) - Iv[p][k];
                            se += fabsf((float)diff);
                        }
                    }
                }
            }
            const float errval = wave_sum(se);
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

}  // namespace

hipError_t launch_lk_track(const LKArgs& a, int n_pairs, hipStream_t s)
{
    if (n_pairs <= 0 || a.n_pts <= 0) return hipSuccess;
    const int npass = (a.n_items + 63) / 64;
    dim3 grid(a.n_pts, n_pairs), block(64);
    switch (npass) {
    case 1: hipLaunchKernelGGL((lk_track_kernel<1>), grid, block, 0, s, a); break;
    case 2: hipLaunchKernelGGL((lk_track_kernel<2>), grid, block, 0, s, a); break;
    case 3: hipLaunchKernelGGL((lk_track_kernel<3>), grid, block, 0, s, a); break;
    case 4: hipLaunchKernelGGL((lk_track_kernel<4>), grid, block, 0, s, a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace dr3
