// micro-benchmark: what a vector-memory instruction costs the CU's texture path (TA / L1) on gfx950 as a function of its per-lane
// access shape -- the shapes the LK tracker's per-level phase is made of and the alternatives tried for it (profiles/r03_lk/README.md).
// Every wave holds eight "points" (8 lanes each) with pseudo-random window origins in an L2-resident plane and issues, per trip,
// 16 loads of consecutive rows of its windows; the loaded dwords are xor-ed into one register (nothing else happens), 12 waves per CU
// (3 per SIMD) like the tracker.  Printed: nanoseconds and shader clocks per wave-instruction PER CU (wall time x CUs / instructions).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/vmem_rate.hip -o tools/ubench/vmem_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) uint8_t* gptr;

template <class T> __device__ __forceinline__ uint32_t fold(T v);
template <> __device__ __forceinline__ uint32_t fold<uint32_t>(uint32_t v) { return v; }
template <> __device__ __forceinline__ uint32_t fold<u32x2>(u32x2 v) { return v.x ^ v.y; }
template <> __device__ __forceinline__ uint32_t fold<u32x3>(u32x3 v) { return v.x ^ v.y ^ v.z; }
template <> __device__ __forceinline__ uint32_t fold<u32x4>(u32x4 v) { return v.x ^ v.y ^ v.z ^ v.w; }

// SHAPE: 0 dwordx2, lane stride 4 B inside a point (the tracker's derivative / image loads: 8 lanes read 36 overlapping bytes)
//        1 dwordx3, lane stride 8 B (adjacent columns per lane: 68 contiguous bytes per point)
//        2 dwordx4, lane stride 16 B (128 contiguous bytes per point)
//        3 dword,   lane stride 4 B (the J region's staging: 32 contiguous bytes per point)
//        4 dwordx2, lane stride 8 B (64 contiguous bytes per point, no overlap)
//        5 dwordx2, all 64 lanes contiguous (512 B of one row)
//        6 dwordx4, all 64 lanes contiguous (1 KiB of one row)
//        7 dword,   all 64 lanes contiguous (256 B of one row)
//        8 dwordx2, lane stride 4 B, but the 8 points of a wave side by side in ONE row (60 B apart)
//        9 dwordx4, 5 lanes per point-row (80 contiguous bytes), 12 point-rows per instruction (60 lanes)
//       10 shape 2 as LDS-DMA (global_load_lds_dwordx4: no VGPR destination)
//       11 shape 9 as LDS-DMA
template <int SHAPE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3))) void vmem_kernel(const uint8_t* plane, int pitch, int rows, int trips, uint32_t* out)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_land[3][256];
    const int lane = threadIdx.x, p = (SHAPE == 9 || SHAPE == 11) ? lane / 5 : lane >> 3, c = (SHAPE == 9 || SHAPE == 11) ? lane % 5 : lane & 7;
    const uint32_t wid = blockIdx.x;
    // window origins: a hash of (wave, point), 4-byte aligned columns, rows so that 16 * trips rows fit below... (wrap by modulo)
    uint32_t h = (wid * 8u + (uint32_t)p) * 2654435761u;
    const uint32_t x0 = ((h >> 8) % (uint32_t)(pitch - 256)) & ~3u;
    uint32_t y = (h >> 20) % (uint32_t)(rows - 16);
    uint32_t lane_off;
    if (SHAPE == 0 || SHAPE == 3) lane_off = x0 + 4u * c;
    else if (SHAPE == 1 || SHAPE == 4) lane_off = x0 + 8u * c;
    else if (SHAPE == 2 || SHAPE == 9 || SHAPE == 10 || SHAPE == 11) lane_off = x0 + 16u * c;
    else if (SHAPE == 5) lane_off = (x0 & ~511u) % (uint32_t)(pitch - 1024) + 8u * lane;
    else if (SHAPE == 6) lane_off = (x0 & ~1023u) % (uint32_t)(pitch - 1024) + 16u * lane;
    else if (SHAPE == 7) lane_off = (x0 & ~255u) % (uint32_t)(pitch - 1024) + 4u * lane;
    else { const uint32_t xs = __builtin_amdgcn_readfirstlane(x0) % (uint32_t)(pitch - 1024); lane_off = xs + 60u * p + 4u * c; y = __builtin_amdgcn_readfirstlane(y); }
    gptr base = (gptr)plane;
    uint32_t acc = 0;
    for (int t = 0; t < trips; ++t) {
        uint32_t off = y * (uint32_t)pitch + lane_off;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (SHAPE == 0 || SHAPE == 4 || SHAPE == 5 || SHAPE == 8) acc ^= fold(*reinterpret_cast<const __attribute__((address_space(1))) u32x2*>(base + off));
            else if (SHAPE == 1) acc ^= fold(*reinterpret_cast<const __attribute__((address_space(1))) u32x3*>(base + off));
            else if (SHAPE == 2 || SHAPE == 6 || SHAPE == 9) acc ^= fold(*reinterpret_cast<const __attribute__((address_space(1))) u32x4*>(base + off));
            else if (SHAPE == 10 || SHAPE == 11) __builtin_amdgcn_global_load_lds(base + off, (__attribute__((address_space(3))) void*)s_land[r % 3], 16, 0, 0);
            else acc ^= fold(*reinterpret_cast<const __attribute__((address_space(1))) uint32_t*>(base + off));
            off += (uint32_t)pitch;
        }
        if (SHAPE == 10 || SHAPE == 11) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); acc ^= s_land[t % 3][lane]; }
        y = (y * 5u + 7u) % (uint32_t)(rows - 16);                       // another window next trip
    }
    if (acc == 0x12345678u) out[wid * 64 + lane] = acc;                   // keeps the loads alive
}

template <int SHAPE>
static float run(const uint8_t* plane, int pitch, int rows, int trips, uint32_t* out, int blocks)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(vmem_kernel<SHAPE>, dim3(blocks), dim3(64), 0, 0, plane, pitch, rows, trips, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(vmem_kernel<SHAPE>, dim3(blocks), dim3(64), 0, 0, plane, pitch, rows, trips, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main(int argc, char** argv)
{
    const int pitch = argc > 1 ? atoi(argv[1]) : 5120, rows = argc > 2 ? atoi(argv[2]) : 408;     // the tracker's level-0 derivative plane
    const int trips = 64, n_cu = 256, blocks = n_cu * 12 * 8;                                           // 8 rounds of 12 waves per CU
    uint8_t* plane; uint32_t* out;
    hipMalloc(&plane, (size_t)pitch * rows + 4096); hipMemset(plane, 1, (size_t)pitch * rows + 4096);
    hipMalloc(&out, (size_t)blocks * 64 * 4);
    const char* names[] = {"dwordx2, 8 lanes of a point 4 B apart (tracker template)", "dwordx3, lanes 8 B apart (adjacent columns)",
                           "dwordx4, lanes 16 B apart (128 B per point)", "dword, lanes 4 B apart (J staging)",
                           "dwordx2, lanes 8 B apart (64 B per point)", "dwordx2, 64 lanes contiguous", "dwordx4, 64 lanes contiguous",
                           "dword, 64 lanes contiguous", "dwordx2, lanes 4 B apart, the 8 points in one row",
                           "dwordx4, 5 lanes per point-row, 12 point-rows per instruction", "LDS-DMA dwordx4, lanes 16 B apart (128 B per point)",
                           "LDS-DMA dwordx4, 5 lanes per point-row, 12 point-rows"};
    float ms[12];
    ms[0] = run<0>(plane, pitch, rows, trips, out, blocks); ms[1] = run<1>(plane, pitch, rows, trips, out, blocks);
    ms[2] = run<2>(plane, pitch, rows, trips, out, blocks); ms[3] = run<3>(plane, pitch, rows, trips, out, blocks);
    ms[4] = run<4>(plane, pitch, rows, trips, out, blocks); ms[5] = run<5>(plane, pitch, rows, trips, out, blocks);
    ms[6] = run<6>(plane, pitch, rows, trips, out, blocks); ms[7] = run<7>(plane, pitch, rows, trips, out, blocks);
    ms[8] = run<8>(plane, pitch, rows, trips, out, blocks);
    ms[9] = run<9>(plane, pitch, rows, trips, out, blocks); ms[10] = run<10>(plane, pitch, rows, trips, out, blocks); ms[11] = run<11>(plane, pitch, rows, trips, out, blocks);
    const double insts_per_cu = (double)blocks / n_cu * trips * 16;
    printf("plane %d x %d bytes (%.1f MB), %d waves, %d loads per wave\n", pitch, rows, pitch * (double)rows / 1e6, blocks, trips * 16);
    for (int i = 0; i < 12; ++i)
        printf("%-62s %8.3f ms  %7.2f ns = %6.1f clk (2.4 GHz) per wave-instruction and CU\n", names[i], ms[i], ms[i] * 1e6 / insts_per_cu, ms[i] * 1e6 / insts_per_cu * 2.4);
    return 0;
}
