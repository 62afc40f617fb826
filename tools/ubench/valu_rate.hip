// micro-benchmark: issue rate of a few VALU instructions on gfx950 (one wave per SIMD .. 8 waves per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
template <int OP> __global__ void k(uint32_t* out, int iters, uint32_t seed)
{
    uint32_t a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed * (threadIdx.x + 1) + i;
    uint32_t w = seed | 0x00010001u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) a[i] = a[i] * 3u + w;                                                     // v_mad_u32_u24 / mul_lo+add
            if (OP == 1) a[i] = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, a[i]), __builtin_bit_cast(u16x2, w), a[i], false);
            if (OP == 2) a[i] = (uint32_t)__builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, a[i]), __builtin_bit_cast(s16x2, w), (int)a[i], false);
            if (OP == 3) a[i] = __builtin_amdgcn_perm(a[i], w, 0x05040100u) ^ a[i];
            if (OP == 4) a[i] = __builtin_amdgcn_udot4(a[i], w, a[i], false);
            if (OP == 5) a[i] = (uint32_t)__mul24((int)a[i], (int)w) + 1u;
            if (OP == 6) a[i] = __float_as_uint((float)(int)a[i]) + 1u;                            // v_cvt_f32_i32
            if (OP == 7) a[i] = a[i] + w;                                                          // v_add_u32
        }
    }
    uint32_t s = 0; for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char* name, uint32_t* d)
{
    const int iters = 20000, blocks = 256 * 8, threads = 256;      // 8 waves per SIMD on 256 CUs
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 100, 7u);
    hipEventRecord(e0); hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, iters, 7u); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = (double)blocks * (threads / 64) * iters * 8;
    printf("%-24s %.3f ms  %.1f G wave-instr/s  (%.2f clk per wave-instr per SIMD at 2.4 GHz)\n", name, ms, wave_instr / ms / 1e6,
           ms * 1e-3 * 2.4e9 * 1024 / wave_instr);
}
int main()
{
    uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<7>("v_add_u32", d); run<0>("mul+add (u32)", d); run<5>("v_mul_i32_i24 + add", d); run<6>("v_cvt_f32_i32 + add", d);
    run<3>("v_perm_b32 + xor", d); run<1>("v_dot2_u32_u16", d); run<2>("v_dot2_i32_i16", d); run<4>("v_dot4_u32_u8", d);
    return 0;
}
