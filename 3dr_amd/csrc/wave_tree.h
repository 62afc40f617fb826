// wave_tree.h -- the f32 wave64 reduction tree shared by the tracker (lk_kernels.hip) and the sparse image alignment
// (align_kernels.hip).  Its order is part of the definition of the kernels' summation order that the CPU checker restates (DESIGN.md
// section 5): a[l] = p[l] + p[l+32] (l < 32); c[l] = a[l] + a[l+16] (l < 16); xor butterfly 1, 2, 4, 8.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dr3 {
namespace {

// ---- wave64 reductions of several f32 sums at once ----------------------------------------------
// One tree for every sum: fold the upper half-wave onto the lower (v_permlane32_swap), fold the odd rows of 16 lanes
// onto the even ones (v_permlane16_swap), xor butterfly 1,2,4,8 inside the row.  The swaps move two (four) different
// sums into disjoint rows, so one butterfly finishes all of them.
template <int CTRL>
__device__ __forceinline__ float dpp_xchg(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// lanes 0-31 <- a[l] + a[l+32], lanes 32-63 <- b[l-32] + b[l]
__device__ __forceinline__ float fold32(float a, float b)
{
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
}
// rows (R0 R1 R2 R3) of u and w  ->  (u.R0 + u.R1, w.R0 + w.R1, u.R2 + u.R3, w.R2 + w.R3)
__device__ __forceinline__ float fold16(float u, float w)
{
    const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(u), __float_as_uint(w), false, false);
    return __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
}
__device__ __forceinline__ float row_butterfly(float v)
{
    v += dpp_xchg<0xB1>(v);     // quad_perm [1,0,3,2]   (xor 1)
    v += dpp_xchg<0x4E>(v);     // quad_perm [2,3,0,1]   (xor 2)
    v += dpp_xchg<0x141>(v);    // row_half_mirror       (= xor 4 once quads are uniform)
    v += dpp_xchg<0x140>(v);    // row_mirror            (= xor 8 once half-rows are uniform)
    return v;
}
__device__ __forceinline__ float lane_value(float v, int lane)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ void wave_sum2(float a, float b, float& ra, float& rb)
{
    const float u = fold32(a, b);
    const float v = row_butterfly(fold16(u, u));           // rows: a a b b
    ra = lane_value(v, 0); rb = lane_value(v, 32);
}
__device__ __forceinline__ void wave_sum3(float a, float b, float c, float& ra, float& rb, float& rc)
{
    const float v = row_butterfly(fold16(fold32(a, b), fold32(c, c)));      // rows: a c b c
    ra = lane_value(v, 0); rb = lane_value(v, 32); rc = lane_value(v, 16);
}

// the tree's total of one value per lane, in every lane
__device__ __forceinline__ float wave_sum1(float a)
{
    const float u = fold32(a, a);
    return row_butterfly(fold16(u, u));
}

}  // namespace
}  // namespace dr3
