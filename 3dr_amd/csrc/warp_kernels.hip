// warp_kernels.hip -- bilinear backward warp (perspective and map-driven) and the 2x2 box
// half-sampler (gfx950).
//   warp_perspective : cv::warpPerspective at reference src/stitch.cpp:73-74 (8UC3) and
//                      src/panorama.cpp:192 (32FC4)            -- SURVEY Appendix A-4
//   remap            : cv::remap at reference src/utils.cpp:191 -- Appendix A-5
//   halfsample       : utils::reduce_to_half, reference src/utils.cpp:382-419 (SSE2 :324-349)
// All three are HBM-bound gathers/streams: one lane owns 4 adjacent destination pixels (8-bit
// 1/3-channel types, dword stores, 768 B contiguous per wave for 8UC3) or one pixel (4-byte and
// 16-byte pixel types), source taps come through the vector L1/L2.  Coordinates follow OpenCV
// bit for bit: f64 homography evaluation with the WarpPerspectiveInvoker's x0+x1 block split,
// round-half-even to 1/32 px, 15-bit integer weights for 8-bit data, f32 table weights summed left
// to right without FMA contraction for f32 data.
#include "dr3hip_internal.h"

namespace dr3 {
namespace {

struct Tap { int sx, sy, fx, fy; };

__device__ __forceinline__ int sat_s16(int v) { return v < -32768 ? -32768 : v > 32767 ? 32767 : v; }

// std::max((double)INT_MIN, std::min((double)INT_MAX, v)) with the std:: argument order
__device__ __forceinline__ double clamp_int_range(double v)
{
    const double hi = 2147483647.0, lo = -2147483648.0;
    const double m = (v < hi) ? v : hi;
    return (lo < m) ? m : lo;
}

__device__ __forceinline__ Tap persp_tap(const double* __restrict__ M, int bw0, int bw0_shift, int x, int y)
{
    const int x0 = bw0_shift >= 0 ? ((x >> bw0_shift) << bw0_shift) : (x / bw0) * bw0;
    const int x1 = x - x0;
    const double X0 = M[0] * x0 + M[1] * y + M[2];
    const double Y0 = M[3] * x0 + M[4] * y + M[5];
    const double W0 = M[6] * x0 + M[7] * y + M[8];
    double W = W0 + M[6] * x1;
    W = (W != 0.0) ? 32.0 / W : 0.0;
    const double fX = clamp_int_range((X0 + M[0] * x1) * W);
    const double fY = clamp_int_range((Y0 + M[3] * x1) * W);
    const int X = __double2int_rn(fX), Y = __double2int_rn(fY);
    Tap t;
    t.sx = sat_s16(X >> 5); t.sy = sat_s16(Y >> 5); t.fx = X & 31; t.fy = Y & 31;
    return t;
}

__device__ __forceinline__ void weights_8u(int fx, int fy, int w[4])
{
    w[0] = (32 - fx) * (32 - fy) * 32; w[1] = fx * (32 - fy) * 32;
    w[2] = (32 - fx) * fy * 32;        w[3] = fx * fy * 32;
    if ((fx | fy) == 0) { w[0] = 32767; w[3] = 1; }   // OpenCV's saturated table entry + sum repair
}
__device__ __forceinline__ void weights_32f(int fx, int fy, float w[4])
{
    const float x1 = (float)fx * (1.f / 32), x0 = 1.f - x1, y1 = (float)fy * (1.f / 32), y0 = 1.f - y1;
    w[0] = y0 * x0; w[1] = y0 * x1; w[2] = y1 * x0; w[3] = y1 * x1;
}

template <int TYPE> struct PixelTraits;
template <> struct PixelTraits<DR3HIP_8UC1>  { static constexpr int CN = 1, PXT = 4; };
template <> struct PixelTraits<DR3HIP_8UC3>  { static constexpr int CN = 3, PXT = 4; };
template <> struct PixelTraits<DR3HIP_8UC4>  { static constexpr int CN = 4, PXT = 1; };
template <> struct PixelTraits<DR3HIP_32FC1> { static constexpr int CN = 1, PXT = 1; };
template <> struct PixelTraits<DR3HIP_32FC4> { static constexpr int CN = 4, PXT = 1; };

__device__ __forceinline__ uint32_t blend_8u(int v0, int v1, int v2, int v3, const int w[4])
{
    return (uint32_t)((v0 * w[0] + v1 * w[1] + v2 * w[2] + v3 * w[3] + (1 << 14)) >> 15);
}

// One destination pixel.  out: CN bytes (8-bit types, low bytes of out[0..]) or CN floats.
template <int TYPE>
__device__ __forceinline__ void sample(const uint8_t* __restrict__ src, int sw, int sh, int sstride,
                                       const Tap t, uint32_t out[4])
{
    using P = PixelTraits<TYPE>;
    const int sx = t.sx, sy = t.sy;
    if (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0) {      // all four taps outside: border 0
#pragma unroll
        for (int k = 0; k < 4; ++k) out[k] = 0;
        return;
    }
    const bool interior = sx >= 0 && sy >= 0 && sx + 1 < sw && sy + 1 < sh;
    const uint8_t* r0 = src + (long long)sy * sstride;
    const uint8_t* r1 = r0 + sstride;

    if (TYPE == DR3HIP_32FC4) {
        float w[4]; weights_32f(t.fx, t.fy, w);
        float4 v0 = make_float4(0, 0, 0, 0), v1 = v0, v2 = v0, v3 = v0;
        if (interior) {
            v0 = reinterpret_cast<const float4*>(r0)[sx];     v1 = reinterpret_cast<const float4*>(r0)[sx + 1];
            v2 = reinterpret_cast<const float4*>(r1)[sx];     v3 = reinterpret_cast<const float4*>(r1)[sx + 1];
        } else {
            const bool x0ok = sx >= 0, x1ok = sx + 1 < sw, y0ok = sy >= 0, y1ok = sy + 1 < sh;
            if (x0ok && y0ok) v0 = reinterpret_cast<const float4*>(r0)[sx];
            if (x1ok && y0ok) v1 = reinterpret_cast<const float4*>(r0)[sx + 1];
            if (x0ok && y1ok) v2 = reinterpret_cast<const float4*>(r1)[sx];
            if (x1ok && y1ok) v3 = reinterpret_cast<const float4*>(r1)[sx + 1];
        }
        out[0] = __float_as_uint(v0.x * w[0] + v1.x * w[1] + v2.x * w[2] + v3.x * w[3]);
        out[1] = __float_as_uint(v0.y * w[0] + v1.y * w[1] + v2.y * w[2] + v3.y * w[3]);
        out[2] = __float_as_uint(v0.z * w[0] + v1.z * w[1] + v2.z * w[2] + v3.z * w[3]);
        out[3] = __float_as_uint(v0.w * w[0] + v1.w * w[1] + v2.w * w[2] + v3.w * w[3]);
        return;
    }
    if (TYPE == DR3HIP_32FC1) {
        float w[4]; weights_32f(t.fx, t.fy, w);
        const bool x0ok = sx >= 0, x1ok = sx + 1 < sw, y0ok = sy >= 0, y1ok = sy + 1 < sh;
        const float v0 = (x0ok && y0ok) ? reinterpret_cast<const float*>(r0)[sx] : 0.f;
        const float v1 = (x1ok && y0ok) ? reinterpret_cast<const float*>(r0)[sx + 1] : 0.f;
        const float v2 = (x0ok && y1ok) ? reinterpret_cast<const float*>(r1)[sx] : 0.f;
        const float v3 = (x1ok && y1ok) ? reinterpret_cast<const float*>(r1)[sx + 1] : 0.f;
        out[0] = __float_as_uint(v0 * w[0] + v1 * w[1] + v2 * w[2] + v3 * w[3]);
        return;
    }

    int w[4]; weights_8u(t.fx, t.fy, w);
    if (TYPE == DR3HIP_8UC4) {
        const bool x0ok = sx >= 0, x1ok = sx + 1 < sw, y0ok = sy >= 0, y1ok = sy + 1 < sh;
        const uint32_t v0 = (x0ok && y0ok) ? reinterpret_cast<const uint32_t*>(r0)[sx] : 0u;
        const uint32_t v1 = (x1ok && y0ok) ? reinterpret_cast<const uint32_t*>(r0)[sx + 1] : 0u;
        const uint32_t v2 = (x0ok && y1ok) ? reinterpret_cast<const uint32_t*>(r1)[sx] : 0u;
        const uint32_t v3 = (x1ok && y1ok) ? reinterpret_cast<const uint32_t*>(r1)[sx + 1] : 0u;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            out[c] = blend_8u((v0 >> (8 * c)) & 0xff, (v1 >> (8 * c)) & 0xff, (v2 >> (8 * c)) & 0xff, (v3 >> (8 * c)) & 0xff, w);
        return;
    }
    if (TYPE == DR3HIP_8UC3) {
        // fast path: pixels sx..sx+3 exist on both rows, so three aligned dwords per row stay in-row
        if (sx >= 0 && sy >= 0 && sx + 3 < sw && sy + 1 < sh && ((reinterpret_cast<uintptr_t>(src) | (uintptr_t)sstride) & 3) == 0) {
            const int bo = 3 * sx;
            const uint32_t* q0 = reinterpret_cast<const uint32_t*>(r0 + (bo & ~3));
            const uint32_t* q1 = reinterpret_cast<const uint32_t*>(r1 + (bo & ~3));
            const int o = bo & 3;
            const uint32_t a0 = q0[0], a1 = q0[1], a2 = q0[2];
            const uint32_t b0 = q1[0], b1 = q1[1], b2 = q1[2];
            // bytes o..o+3 and o+4..o+7 of each row
            const uint32_t t0 = __builtin_amdgcn_alignbyte(a1, a0, o), t1 = __builtin_amdgcn_alignbyte(a2, a1, o);
            const uint32_t u0 = __builtin_amdgcn_alignbyte(b1, b0, o), u1 = __builtin_amdgcn_alignbyte(b2, b1, o);
            // pixel sx = bytes 0,1,2 ; pixel sx+1 = bytes 3,4,5
            out[0] = blend_8u(t0 & 0xff, t0 >> 24, u0 & 0xff, u0 >> 24, w);
            out[1] = blend_8u((t0 >> 8) & 0xff, t1 & 0xff, (u0 >> 8) & 0xff, u1 & 0xff, w);
            out[2] = blend_8u((t0 >> 16) & 0xff, (t1 >> 8) & 0xff, (u0 >> 16) & 0xff, (u1 >> 8) & 0xff, w);
            return;
        }
    }
    // generic 8-bit path (8UC1, 8UC3 near borders): per-tap checks, byte loads
    {
        constexpr int CN = P::CN;
        const bool x0ok = sx >= 0, x1ok = sx + 1 < sw, y0ok = sy >= 0, y1ok = sy + 1 < sh;
#pragma unroll
        for (int c = 0; c < CN; ++c) {
            const int v0 = (x0ok && y0ok) ? r0[sx * CN + c] : 0;
            const int v1 = (x1ok && y0ok) ? r0[(sx + 1) * CN + c] : 0;
            const int v2 = (x0ok && y1ok) ? r1[sx * CN + c] : 0;
            const int v3 = (x1ok && y1ok) ? r1[(sx + 1) * CN + c] : 0;
            out[c] = blend_8u(v0, v1, v2, v3, w);
        }
    }
}

// write PXT pixels starting at (x, y); px[i][c] as produced by sample()
template <int TYPE>
__device__ __forceinline__ void store_pixels(uint8_t* __restrict__ dst, int dw, int dstride, int x, int y,
                                             const uint32_t px[][4])
{
    using P = PixelTraits<TYPE>;
    uint8_t* row = dst + (long long)y * dstride;
    if (TYPE == DR3HIP_32FC4) {
        uint4 v; v.x = px[0][0]; v.y = px[0][1]; v.z = px[0][2]; v.w = px[0][3];
        reinterpret_cast<uint4*>(row)[x] = v;
    } else if (TYPE == DR3HIP_32FC1) {
        reinterpret_cast<uint32_t*>(row)[x] = px[0][0];
    } else if (TYPE == DR3HIP_8UC4) {
        reinterpret_cast<uint32_t*>(row)[x] = px[0][0] | (px[0][1] << 8) | (px[0][2] << 16) | (px[0][3] << 24);
    } else {
        constexpr int CN = P::CN;
        const bool aligned = ((reinterpret_cast<uintptr_t>(dst) | (uintptr_t)dstride) & 3) == 0;
        if (x + 3 < dw && aligned) {
            uint8_t b[4 * CN];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int c = 0; c < CN; ++c) b[i * CN + c] = (uint8_t)px[i][c];
            uint32_t* o = reinterpret_cast<uint32_t*>(row + x * CN);
#pragma unroll
            for (int d = 0; d < CN; ++d)
                o[d] = (uint32_t)b[4 * d] | ((uint32_t)b[4 * d + 1] << 8) | ((uint32_t)b[4 * d + 2] << 16) | ((uint32_t)b[4 * d + 3] << 24);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (x + i < dw)
#pragma unroll
                    for (int c = 0; c < CN; ++c) row[(x + i) * CN + c] = (uint8_t)px[i][c];
        }
    }
}

template <int TYPE>
__global__ __launch_bounds__(256) void warp_perspective_kernel(const WarpJobDev* __restrict__ jobs)
{
    using P = PixelTraits<TYPE>;
    const WarpJobDev& jb = jobs[blockIdx.z];
    const int x = (blockIdx.x * 64 + threadIdx.x) * P::PXT;
    const int y = blockIdx.y * 4 + threadIdx.y;
    const int dw = jb.dw, dh = jb.dh;
    if (x >= dw || y >= dh) return;
    const int bw0 = jb.bw0;
    const int bw0_shift = (bw0 & (bw0 - 1)) == 0 ? __builtin_ctz(bw0) : -1;
    double M[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) M[i] = jb.M[i];

    uint32_t px[P::PXT][4];
#pragma unroll
    for (int i = 0; i < P::PXT; ++i) {
        if (x + i < dw) {
            const Tap t = persp_tap(M, bw0, bw0_shift, x + i, y);
            sample<TYPE>(jb.src, jb.sw, jb.sh, jb.sstride, t, px[i]);
        }
    }
    store_pixels<TYPE>(jb.dst, dw, jb.dstride, x, y, px);
}

template <int TYPE>
__global__ __launch_bounds__(256) void remap_kernel(const RemapArgs a)
{
    using P = PixelTraits<TYPE>;
    const int x = (blockIdx.x * 64 + threadIdx.x) * P::PXT;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x >= a.dw || y >= a.dh) return;
    const float* mx = reinterpret_cast<const float*>(reinterpret_cast<const uint8_t*>(a.mapx) + (long long)y * a.map_stride);
    const float* my = reinterpret_cast<const float*>(reinterpret_cast<const uint8_t*>(a.mapy) + (long long)y * a.map_stride);
    uint32_t px[P::PXT][4];
#pragma unroll
    for (int i = 0; i < P::PXT; ++i) {
        if (x + i < a.dw) {
            // cvRound(mapx * INTER_TAB_SIZE): f32 multiply, round half to even
            const int sx = __float2int_rn(mx[x + i] * 32.f), sy = __float2int_rn(my[x + i] * 32.f);
            Tap t; t.sx = sat_s16(sx >> 5); t.sy = sat_s16(sy >> 5); t.fx = sx & 31; t.fy = sy & 31;
            sample<TYPE>(a.src, a.sw, a.sh, a.sstride, t, px[i]);
        }
    }
    store_pixels<TYPE>(a.dst, a.dw, a.dstride, x, y, px);
}

// ---- 2x2 box half-sampler ---------------------------------------------------------------------
__device__ __forceinline__ uint32_t half_px(int a, int b, int c, int d, int mode)
{
    if (mode == DR3HIP_HALF_SSE2) {                       // _mm_avg_epu8 then _mm_avg_epu16
        const int v0 = (a + c + 1) >> 1, v1 = (b + d + 1) >> 1;
        return (uint32_t)((v0 + v1 + 1) >> 1);
    }
    return (uint32_t)((a + b + c + d) >> 2);              // (a+b+c+d)/4, all non-negative
}

__global__ __launch_bounds__(256) void halfsample_kernel(const HalfArgs a)
{
    const int X = (blockIdx.x * 64 + threadIdx.x) * 4;    // 4 output pixels per lane
    const int Y = blockIdx.y * 4 + threadIdx.y;
    if (X >= a.ow || Y >= a.oh) return;
    uint32_t o[4] = {0, 0, 0, 0};
    if (a.mode == DR3HIP_HALF_REF_SCALAR) {
        // reference pointer walk: flat offset of output (Y, X) is Y*(2*ow + stride) + 2*X
        const long long total = (long long)a.sstride * a.h;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (X + k < a.ow) {
                const long long t = (long long)Y * (2LL * a.ow + a.sstride) + 2LL * (X + k), b = t + a.sstride;
                const int p = t < total ? a.src[t] : 0, q = t + 1 < total ? a.src[t + 1] : 0;
                const int r = b < total ? a.src[b] : 0, s = b + 1 < total ? a.src[b + 1] : 0;
                o[k] = half_px(p, q, r, s, DR3HIP_HALF_TRUNC);
            }
        }
    } else {
        const uint8_t* r0 = a.src + (long long)(2 * Y) * a.sstride + 2 * X;
        const uint8_t* r1 = r0 + a.sstride;
        const bool fast = (X + 3 < a.ow) && (((reinterpret_cast<uintptr_t>(a.src) | (uintptr_t)a.sstride) & 7) == 0);
        if (fast) {
            const uint2 t = *reinterpret_cast<const uint2*>(r0), b = *reinterpret_cast<const uint2*>(r1);
            const uint32_t tw[2] = {t.x, t.y}, bw[2] = {b.x, b.y};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t tt = tw[k >> 1] >> (16 * (k & 1)), bb = bw[k >> 1] >> (16 * (k & 1));
                o[k] = half_px(tt & 0xff, (tt >> 8) & 0xff, bb & 0xff, (bb >> 8) & 0xff, a.mode);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (X + k < a.ow) o[k] = half_px(r0[2 * k], r0[2 * k + 1], r1[2 * k], r1[2 * k + 1], a.mode);
        }
    }
    uint8_t* drow = a.dst + (long long)Y * a.dstride;
    if (X + 3 < a.ow && ((reinterpret_cast<uintptr_t>(a.dst) | (uintptr_t)a.dstride) & 3) == 0) {
        *reinterpret_cast<uint32_t*>(drow + X) = o[0] | (o[1] << 8) | (o[2] << 16) | (o[3] << 24);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) if (X + k < a.ow) drow[X + k] = (uint8_t)o[k];
    }
}

}  // namespace

hipError_t launch_warp(const WarpJobDev* d_jobs, int n_jobs, int max_dw, int max_dh, int type, hipStream_t s)
{
    if (n_jobs <= 0 || max_dw <= 0 || max_dh <= 0) return hipSuccess;
    const int pxt = (type == DR3HIP_8UC1 || type == DR3HIP_8UC3) ? 4 : 1;
    dim3 block(64, 4);
    dim3 grid((max_dw + 64 * pxt - 1) / (64 * pxt), (max_dh + 3) / 4, n_jobs);
    switch (type) {
    case DR3HIP_8UC1:  hipLaunchKernelGGL((warp_perspective_kernel<DR3HIP_8UC1>),  grid, block, 0, s, d_jobs); break;
    case DR3HIP_8UC3:  hipLaunchKernelGGL((warp_perspective_kernel<DR3HIP_8UC3>),  grid, block, 0, s, d_jobs); break;
    case DR3HIP_8UC4:  hipLaunchKernelGGL((warp_perspective_kernel<DR3HIP_8UC4>),  grid, block, 0, s, d_jobs); break;
    case DR3HIP_32FC1: hipLaunchKernelGGL((warp_perspective_kernel<DR3HIP_32FC1>), grid, block, 0, s, d_jobs); break;
    case DR3HIP_32FC4: hipLaunchKernelGGL((warp_perspective_kernel<DR3HIP_32FC4>), grid, block, 0, s, d_jobs); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_remap(const RemapArgs& a, int type, hipStream_t s)
{
    if (a.dw <= 0 || a.dh <= 0) return hipSuccess;
    const int pxt = (type == DR3HIP_8UC1 || type == DR3HIP_8UC3) ? 4 : 1;
    dim3 block(64, 4);
    dim3 grid((a.dw + 64 * pxt - 1) / (64 * pxt), (a.dh + 3) / 4, 1);
    switch (type) {
    case DR3HIP_8UC1:  hipLaunchKernelGGL((remap_kernel<DR3HIP_8UC1>),  grid, block, 0, s, a); break;
    case DR3HIP_8UC3:  hipLaunchKernelGGL((remap_kernel<DR3HIP_8UC3>),  grid, block, 0, s, a); break;
    case DR3HIP_8UC4:  hipLaunchKernelGGL((remap_kernel<DR3HIP_8UC4>),  grid, block, 0, s, a); break;
    case DR3HIP_32FC1: hipLaunchKernelGGL((remap_kernel<DR3HIP_32FC1>), grid, block, 0, s, a); break;
    case DR3HIP_32FC4: hipLaunchKernelGGL((remap_kernel<DR3HIP_32FC4>), grid, block, 0, s, a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_halfsample(const HalfArgs& a, hipStream_t s)
{
    if (a.ow <= 0 || a.oh <= 0) return hipSuccess;
    dim3 block(64, 4);
    dim3 grid((a.ow + 255) / 256, (a.oh + 3) / 4, 1);
    hipLaunchKernelGGL(halfsample_kernel, grid, block, 0, s, a);
    return hipGetLastError();
}

}  // namespace dr3
