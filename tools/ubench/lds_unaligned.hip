// micro-benchmark: does gfx950's LDS serve a 4-byte read at a 2-byte-aligned address (ds_read_b32, unaligned access mode), and at
// what rate against the aligned forms the tracker uses now?  One wave per workgroup, 8 "points" of 8 lanes; every lane reads 17 rows
// of its point's region per trip, like the tracker's iteration.  MODE 0: ds_read2_b32 (two aligned dwords, the shipped form: byte
// image + v_perm), 1: one ds_read_b32 at a 2-byte-aligned address of a u16 image (half the lanes misaligned), 2: the same with every
// lane aligned (even columns only), 3: ds_read_u16 x 2.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_unaligned.hip -o tools/ubench/lds_unaligned
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void k(uint32_t* out, int trips, int check)
{
    __shared__ __attribute__((aligned(16))) uint16_t img[8][24][34];          // u16 image: value = point << 12 | row << 6 | col
    const int lane = threadIdx.x, g = lane >> 3, hl = lane & 7;
    for (int i = lane; i < 8 * 24 * 34; i += 64) { const int p = i / (24 * 34), r = (i / 34) % 24, c = i % 34; img[p][r][c] = (uint16_t)(p << 12 | r << 6 | c); }
    __syncthreads();
    uint32_t acc = 0, bad = 0;
    for (int t = 0; t < trips; ++t) {
        const int col = (hl + t) & 15, row0 = t & 7;
        const int cc = MODE == 2 ? (col & ~1) : col;
        const uint8_t* base = reinterpret_cast<const uint8_t*>(&img[g][row0][0]) + 2 * cc;
        uint32_t v[17];
        if (MODE == 1 || MODE == 2) {
            const uint32_t a = (uint32_t)(uintptr_t)base;         // LDS byte address
#pragma unroll
            for (int r = 0; r < 17; ++r) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v[r]) : "v"(a), "i"(r * 68));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int r = 0; r < 17; ++r) asm volatile("" : "+v"(v[r]));
        } else {
#pragma unroll
            for (int r = 0; r < 17; ++r) {
                if (MODE == 0) {
                    const uint32_t* q = reinterpret_cast<const uint32_t*>(base - 2 * (cc & 1) + r * 68);
                    const uint32_t lo = q[0], hi = q[1];
                    v[r] = __builtin_amdgcn_perm(hi, lo, (cc & 1) ? 0x05040302u : 0x03020100u);
                } else {
                    const uint16_t* q = reinterpret_cast<const uint16_t*>(base + r * 68);
                    v[r] = q[0] | (uint32_t)q[1] << 16;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 17; ++r) {
            if (check) {
                const uint32_t want = (uint32_t)(g << 12 | (row0 + r) << 6 | cc) | (uint32_t)(g << 12 | (row0 + r) << 6 | (cc + 1)) << 16;
                bad += v[r] != want;
            }
            acc += v[r];
        }
    }
    out[blockIdx.x * 64 + lane] = check ? bad : acc;
}

template <int MODE> void run(const char* name)
{
    const int blocks = 256 * 8 * 4, trips = 512;
    uint32_t* d; hipMalloc(&d, blocks * 64 * 4);
    k<MODE><<<16, 64>>>(d, 64, 1);
    std::vector<uint32_t> h(16 * 64); hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    long bad = 0; for (auto v : h) bad += v;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 64>>>(d, trips, 0);
    hipEventRecord(e0); k<MODE><<<blocks, 64>>>(d, trips, 0); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double reads = (double)blocks * trips * 17;
    printf("%-52s wrong values %ld   %.3f ms   %.2f clk (2.4 GHz) per wave-row-read and CU\n", name, bad, ms, ms * 1e-3 * 2.4e9 * 256 / reads);
    hipFree(d);
}

int main()
{
    run<0>("ds_read2_b32 aligned + v_perm (byte image form)");
    run<1>("ds_read_b32 at 2-byte alignment (u16 image)");
    run<2>("ds_read_b32, every lane aligned");
    run<3>("ds_read_u16 x 2 + pack");
    return 0;
}
