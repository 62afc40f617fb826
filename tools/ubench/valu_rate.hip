// micro-benchmark: ISSUE cost of the instructions the LK tracker is made of, on gfx950, at 1 / 2 / 4 / 8 waves per SIMD.
// Every stream is 8 independent dependency chains of one instruction (inline asm, so the instruction measured is the
// instruction written), unrolled 8 x 8 per loop trip; the clock the chip holds during the stream is measured in the
// kernel (s_memtime ticks per s_memrealtime 100 MHz tick), so "clk per wave-instruction per SIMD" is in real shader
// clocks, not in nominal 2.4 GHz ones.  "mix" streams interleave VALU with SALU the way lk_track_kernel does
// (about 4 VALU : 3 SALU) to see whether the scalar pipe costs VALU issue slots.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o tools/ubench/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-result"
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <vector>
#include <string>

struct Clk { unsigned long long t0, r0, t1, r1; };

#define CHAIN8(STMT) { STMT(0) STMT(1) STMT(2) STMT(3) STMT(4) STMT(5) STMT(6) STMT(7) }
#define REP8(X) X X X X X X X X

// OPS: name, per-chain asm statement on a[i] (uint32) with uniform operands w (VGPR) and sc (SGPR)
#define OP_LIST(F) \
    F(0,  "v_add_u32",           "v_add_u32 %0, %0, %1") \
    F(1,  "v_lshlrev_b32",       "v_lshlrev_b32 %0, 1, %0") \
    F(2,  "v_and_b32",           "v_and_b32 %0, %0, %1") \
    F(3,  "v_perm_b32",          "v_perm_b32 %0, %0, %1, %2") \
    F(4,  "v_dot2_i32_i16",      "v_dot2_i32_i16 %0, %0, %1, %0") \
    F(5,  "v_dot4_u32_u8",       "v_dot4_u32_u8 %0, %0, %1, %0") \
    F(6,  "v_ashrrev_i32",       "v_ashrrev_i32 %0, 1, %0") \
    F(7,  "v_add_f32",           "v_add_f32 %0, %0, %1") \
    F(8,  "v_mul_f32",           "v_mul_f32 %0, %0, %1") \
    F(9,  "v_fma_f32",           "v_fma_f32 %0, %0, %1, %1") \
    F(10, "v_cvt_f32_i32",       "v_cvt_f32_i32 %0, %0") \
    F(11, "v_floor_f32",         "v_floor_f32 %0, %0") \
    F(12, "v_med3_f32",          "v_med3_f32 %0, %0, %1, %1") \
    F(13, "v_cndmask_b32",       "v_cndmask_b32 %0, %0, %1, vcc") \
    F(14, "v_add_f32_dpp(quad)", "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") \
    F(15, "v_mov_b32",           "v_mov_b32 %0, %1") \
    F(16, "v_sqrt_f32",          "v_sqrt_f32 %0, %0") \
    F(17, "v_rcp_f32",           "v_rcp_f32 %0, %0") \
    F(18, "v_mad_u32_u24",       "v_mad_u32_u24 %0, %0, %1, %1") \
    F(19, "v_lshl_add_u32",      "v_lshl_add_u32 %0, %0, 1, %1") \
    F(20, "v_alignbyte_b32",     "v_alignbyte_b32 %0, %0, %1, 1") \
    F(21, "v_cmp_lt_i32(vcc)",   "v_cmp_lt_i32 vcc, %0, %1") \
    F(22, "v_fmac_f32 (VOP2)",   "v_fmac_f32 %0, %1, %1") \
    F(23, "v_dot2c_i32_i16 (VOP2)", "v_dot2c_i32_i16 %0, %1, %1") \
    F(24, "v_dot4c_i32_i8 (VOP2)",  "v_dot4c_i32_i8 %0, %1, %1") \
    F(25, "v_cvt_f32_ubyte0",    "v_cvt_f32_ubyte0 %0, %0") \
    F(26, "v_cvt_f32_ubyte2",    "v_cvt_f32_ubyte2 %0, %0") \
    F(27, "v_cvt_i32_f32",       "v_cvt_i32_f32 %0, %0") \
    F(28, "v_sub_f32",           "v_sub_f32 %0, %0, %1") \
    F(29, "v_lshrrev_b32",       "v_lshrrev_b32 %0, 1, %0") \
    F(30, "v_or_b32",            "v_or_b32 %0, %0, %1") \
    F(31, "v_xor_b32",           "v_xor_b32 %0, %0, %1") \
    F(32, "v_sub_u32",           "v_sub_u32 %0, %0, %1") \
    F(33, "v_mul_u32_u24 (VOP2)", "v_mul_u32_u24 %0, %0, %1") \
    F(34, "v_max_f32",           "v_max_f32 %0, %0, %1") \
    F(35, "v_lshlrev_b32 by vgpr", "v_lshlrev_b32 %0, %1, %0") \
    F(36, "v_add_u32 sgpr",      "v_add_u32 %0, %2, %0") \
    F(37, "v_mul_f32 sgpr",      "v_mul_f32 %0, %2, %0") \
    F(38, "v_pk_fmac_f16 (VOP2)", "v_pk_fmac_f16 %0, %1, %1") \
    F(39, "v_max_i32",           "v_max_i32 %0, %0, %1") \
    F(40, "v_add_f32 e64 (abs)", "v_add_f32_e64 %0, |%0|, %1") \
    F(41, "v_add3_u32",          "v_add3_u32 %0, %0, %1, %1") \
    F(42, "v_and_or_b32",        "v_and_or_b32 %0, %0, %1, %1") \
    F(43, "v_cvt_f32_u32",       "v_cvt_f32_u32 %0, %0") \
    F(44, "v_mul_i32_i24 (VOP2)", "v_mul_i32_i24 %0, %0, %1") \
    F(45, "v_mac_f32 legacy? v_fmac dpp", "v_fmac_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") \
    F(46, "v_bfe_u32",           "v_bfe_u32 %0, %0, 8, 8") \
    F(47, "v_cndmask_b32 (s pair)", "v_cndmask_b32_e64 %0, %0, %1, s[10:11]")

template <int OP> __device__ __forceinline__ void chain(uint32_t& a, uint32_t w, uint32_t sc)
{
#define F(ID, NAME, TXT) if (OP == ID) asm volatile(TXT : "+v"(a) : "v"(w), "s"(sc) : "vcc");
    OP_LIST(F)
#undef F
}

// 64-bit chains (register pairs)
template <int OP> __device__ __forceinline__ void chain64(double& a, double w)
{
    if (OP == 100) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(w));
    if (OP == 101) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(w));
    if (OP == 102) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a) : "v"(w));
    if (OP == 103) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a) : "v"(w));
    if (OP == 104) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(w));
    if (OP == 105) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a) : "v"(w));
}

__device__ __forceinline__ void stamp(unsigned long long& t, unsigned long long& r)
{
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "=s"(r)::"memory");
}

template <int OP> __global__ __launch_bounds__(256) void k32(uint32_t* out, Clk* clk, int iters, uint32_t seed)
{
    uint32_t a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed * (threadIdx.x + 1) + i;
    const uint32_t w = (OP >= 7 && OP <= 12) || OP == 16 || OP == 17 || OP == 22 || OP == 28 || OP == 34 || OP == 37 || OP == 40 || OP == 45 ? 0x3f800001u : (seed | 0x00010001u);
    const uint32_t sc = 0x05040100u;
    Clk c;
    stamp(c.t0, c.r0);
    for (int it = 0; it < iters; ++it) {
#define S(I) chain<OP>(a[I], w, sc);
        REP8(CHAIN8(S))
#undef S
    }
    stamp(c.t1, c.r1);
    uint32_t s = 0; for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *clk = c;
}

template <int OP> __global__ __launch_bounds__(256) void k64(uint32_t* out, Clk* clk, int iters, uint32_t seed)
{
    double a[8];
    for (int i = 0; i < 8; ++i) a[i] = OP == 105 ? __longlong_as_double(seed + i) : 1.0 + 1e-9 * (threadIdx.x + i);
    const double w = OP == 105 ? __longlong_as_double(3) : 1.0000001;
    Clk c;
    stamp(c.t0, c.r0);
    for (int it = 0; it < iters; ++it) {
#define S(I) chain64<OP>(a[I], w);
        REP8(CHAIN8(S))
#undef S
    }
    stamp(c.t1, c.r1);
    double s = 0; for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)__double_as_longlong(s);
    if (threadIdx.x == 0 && blockIdx.x == 0) *clk = c;
}

// VALU : SALU mixes.  MIX = 0: 4 v_perm + 0 SALU (control); 1: 4 v_perm + 3 s_add; 2: 4 v_perm + 3 (s_and_b64 / s_cselect / s_add);
// 3: 4 v_add_f32 + 3 s_add; 4: as 1 plus one v_readfirstlane per 4 VALU (counts as VALU); 5: 4 v_perm + 1 s_nop 1
template <int MIX> __global__ __launch_bounds__(256) void kmix(uint32_t* out, Clk* clk, int iters, uint32_t seed)
{
    uint32_t a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed * (threadIdx.x + 1) + i;
    const uint32_t w = MIX == 3 ? 0x3f800001u : (seed | 0x00010001u);
    uint32_t s0 = seed, s1 = seed + 1, s2 = seed + 2;
    Clk c;
    stamp(c.t0, c.r0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            // operands: %0-%3 chains, %4-%6 scalars (read-write), %7 uniform VGPR, %8 perm selector (SGPR)
#define MIXASM(TXT) asm volatile(TXT : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+s"(s0), "+s"(s1), "+s"(s2) \
                                 : "v"(w), "s"(0x05040100u) : "vcc", "scc")
            if (MIX == 3)
                MIXASM("v_add_f32 %0, %0, %7\n\ts_add_u32 %4, %4, 1\n\tv_add_f32 %1, %1, %7\n\ts_add_u32 %5, %5, 1\n\t"
                       "v_add_f32 %2, %2, %7\n\ts_add_u32 %6, %6, 1\n\tv_add_f32 %3, %3, %7");
            else if (MIX == 0)
                MIXASM("v_perm_b32 %0, %0, %7, %8\n\tv_perm_b32 %1, %1, %7, %8\n\tv_perm_b32 %2, %2, %7, %8\n\tv_perm_b32 %3, %3, %7, %8");
            else if (MIX == 1)
                MIXASM("v_perm_b32 %0, %0, %7, %8\n\ts_add_u32 %4, %4, 1\n\tv_perm_b32 %1, %1, %7, %8\n\ts_add_u32 %5, %5, 1\n\t"
                       "v_perm_b32 %2, %2, %7, %8\n\ts_add_u32 %6, %6, 1\n\tv_perm_b32 %3, %3, %7, %8");
            else if (MIX == 2)
                MIXASM("v_perm_b32 %0, %0, %7, %8\n\ts_and_b64 vcc, exec, vcc\n\tv_perm_b32 %1, %1, %7, %8\n\ts_cselect_b32 %4, %5, %6\n\t"
                       "v_perm_b32 %2, %2, %7, %8\n\ts_add_u32 %6, %6, 1\n\tv_perm_b32 %3, %3, %7, %8");
            else if (MIX == 4)
                MIXASM("v_perm_b32 %0, %0, %7, %8\n\ts_add_u32 %4, %4, 1\n\tv_perm_b32 %1, %1, %7, %8\n\ts_add_u32 %5, %5, 1\n\t"
                       "v_perm_b32 %2, %2, %7, %8\n\tv_readfirstlane_b32 %6, %3\n\tv_perm_b32 %3, %3, %7, %8");
            else
                MIXASM("v_perm_b32 %0, %0, %7, %8\n\tv_perm_b32 %1, %1, %7, %8\n\ts_nop 1\n\tv_perm_b32 %2, %2, %7, %8\n\tv_perm_b32 %3, %3, %7, %8");
#undef MIXASM
        }
    }
    stamp(c.t1, c.r1);
    uint32_t s = 0; for (int i = 0; i < 4; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + s0 + s1 + s2;
    if (threadIdx.x == 0 && blockIdx.x == 0) *clk = c;
}

struct Row { std::string name; int wps; double ms, ginstr, ghz, clk_per; };
static std::vector<Row> rows;
static int g_only_wps = 0;                              // argv[3]: run only this many waves per SIMD (for counter calibration runs)

template <typename K> static void run(const char* name, K kern, uint32_t* d, Clk* dclk, int valu_per_trip, int iters)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps : {1, 2, 4, 8}) {                      // waves per SIMD: blocks of 256 threads = one wave on each SIMD of a CU
        if (g_only_wps && wps != g_only_wps) continue;
        const int blocks = 256 * wps, threads = 256;
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, dclk, 64, 7u);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, dclk, iters, 7u);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        Clk c; hipMemcpy(&c, dclk, sizeof(c), hipMemcpyDeviceToHost);
        const double ghz = (double)(c.t1 - c.t0) / (double)(c.r1 - c.r0) * 0.1;
        const double wi = (double)blocks * 4 * iters * valu_per_trip;
        Row r{name, wps, ms, wi / ms / 1e6, ghz, ms * 1e-3 * ghz * 1e9 * 1024 / wi};
        rows.push_back(r);
        printf("%-34s %d waves/SIMD  %8.3f ms  %7.1f G VALU wave-instr/s  clock %.2f GHz  %.2f clk per VALU wave-instr per SIMD\n",
               name, wps, ms, r.ginstr, ghz, r.clk_per);
        fflush(stdout);
    }
}

int main(int argc, char** argv)
{
    uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    Clk* dclk; hipMalloc(&dclk, sizeof(Clk));
    const int iters = 4000;
    const int first_op = argc > 2 ? atoi(argv[2]) : 0;
    g_only_wps = argc > 3 ? atoi(argv[3]) : 0;       // argv[2]: skip the single-opcode streams below this id
#define F(ID, NAME, TXT) if (ID >= first_op) run(NAME, k32<ID>, d, dclk, 64, iters);
    OP_LIST(F)
#undef F
    run("v_mul_f64", k64<100>, d, dclk, 64, iters);
    run("v_add_f64", k64<101>, d, dclk, 64, iters);
    run("v_pk_mul_f32", k64<102>, d, dclk, 64, iters);
    run("v_pk_add_f32", k64<103>, d, dclk, 64, iters);
    run("v_pk_fma_f32", k64<104>, d, dclk, 64, iters);
    run("v_lshl_add_u64", k64<105>, d, dclk, 64, iters);
    run("mix 4 v_perm : 0 SALU", kmix<0>, d, dclk, 64, iters);
    run("mix 4 v_perm : 3 s_add", kmix<1>, d, dclk, 64, iters);
    run("mix 4 v_perm : 3 s_and64/csel/add", kmix<2>, d, dclk, 64, iters);
    run("mix 4 v_add_f32 : 3 s_add", kmix<3>, d, dclk, 64, iters);
    run("mix 4 v_perm+1 readfirstlane : 2 s_add", kmix<4>, d, dclk, 80, iters);
    run("mix 4 v_perm : 1 s_nop 1", kmix<5>, d, dclk, 64, iters);
    if (argc > 1) {
        FILE* f = fopen(argv[1], "w");
        fprintf(f, "[\n");
        for (size_t i = 0; i < rows.size(); ++i)
            fprintf(f, " {\"stream\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"G_valu_wave_instr_per_s\": %.2f, \"clock_ghz\": %.3f, \"clk_per_valu_wave_instr_per_simd\": %.3f}%s\n",
                    rows[i].name.c_str(), rows[i].wps, rows[i].ms, rows[i].ginstr, rows[i].ghz, rows[i].clk_per, i + 1 < rows.size() ? "," : "");
        fprintf(f, "]\n");
        fclose(f);
    }
    return 0;
}
