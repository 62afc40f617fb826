// lk_kernels.hip -- pyramidal Lucas-Kanade tracker (gfx950).  Replaces the tracker inside
// cv::calcOpticalFlowPyrLK as called at reference src/initialization.cpp:608-613
// (LKTrackerInvoker).
//
// One kernel shape for every window up to 32 x 32 (sides 33 .. 64: lk_track_big_kernel at the end of this file).  A lane owns taps of ONE window column: the window is cut into lane-items of
// consecutive rows of a column -- windows up to 16 x 16: the whole column (up to 16 taps); other windows: 8 rows, item
// q = (ty / 8) * win_w + tx.  A point is tracked by a GROUP of 64 / PPW lanes, a lane owns the items hl, hl + LANES, ... (NI of
// them), and a wave tracks PPW points in lock step:
//     windows up to 16 x 16 (the bench's 15x15)        PPW = 4   16 lanes = one DPP row per point, lane = column
//     other windows of <= 32 items (e.g. 8 x 32)        PPW = 2   32 lanes, 1 item per lane
//     windows of <= 64 items (21 x 21: 63 items)        PPW = 2   32 lanes, 2 items per lane
//     windows of <= 128 items (the reference's 30x30)   PPW = 1   64 lanes, 2 items per lane
// (windows up to 16 x 16 run as PPW = 8 since round 3: 8 lanes per point, a lane owns the columns hl and hl + 8, two lane-items.)
// What bounds the kernel's per-level phase is the NUMBER of vector-memory instructions: the CU's texture path takes about 16 clocks per
// wave-instruction whatever its width, plus a clock or so per extra cache line (tools/ubench/vmem_rate.hip).  Every form therefore
// stages the J region AND the template's image rows with 8-byte loads, a group's lanes = (row of the round, 8-byte piece) -- the
// eight-point form: 12 + 8 instead of 24 + 32 instructions per level; the image bytes then come out of LDS --, and with a compile-time
// window it issues the template's derivative loads together with those staging loads, before the LDS writes: one memory round trip per
// level (15 x 15: 52 loads in flight).
// Column ownership makes the 16 lanes of a column group load ADJACENT addresses (a wave-instruction touches a handful of cache
// lines; with a lane per window row it touched 64, and the L1's line rate bound the kernel), and lets tap r reuse the permuted
// byte pair of row r + 1 as tap r + 1's row r: one v_perm_b32 per row instead of two per tap.
// All pyramid levels run coarse -> fine inside the wave, the running estimate never leaves registers.  Per level a group
//   1. stages the J search region (window + RM px margin) from global memory into its private LDS slice with aligned dword
//      loads -- ONE global round trip per level instead of one per Gauss-Newton iteration;
//   2. extracts its taps of the template patch (I, Ix, Iy: 14-bit fixed-point bilinear, integer exact) once and keeps them in
//      registers: per tap the dot accumulator BILIN_ROUND - (I << 9), Ix and Iy as int16 pairs of vertically adjacent taps;
//   3. iterates out of LDS: per row one ds_read2_b32 + v_perm_b32 (the byte pair of columns c, c + 1), per tap two
//      v_dot2_i32_i16 (the template value rides in the dot's accumulator, so the dots yield the mismatch directly) and a shift,
//      mismatches packed in int16 pairs and multiplied onto the packed gradients with v_dot2_i32_i16 again: 5.6 VALU per tap;
//   4. reduces its sums across its lanes with v_permlane32_swap / v_permlane16_swap / DPP adds (no LDS), solves the 2x2 system
//      and tests the exit conditions.
// Everything that is uniform across a group -- window origins, the bilinear weights, the 2x2 solve, the exit tests, the
// reductions' tails -- is computed once per wave-instruction for PPW points.  The groups run in lock step: a level / an iteration
// executes while any point of the wave needs it (consecutive points of a frame take the same number of iterations to within a few
// per cent on the bench's stream), a point that is done sits the rest out under an exec mask, and every cross-lane instruction
// runs in wave-uniform control flow with all 64 lanes enabled (the swaps and DPP row operations used for a group never leave it).
// If an iterate leaves the staged region (rare: margin RM px per level) the region is re-staged around it, so results never
// depend on the margin.
// Addressing: every level of a frame batch carries OpenCV's winSize border physically (LevelDev::pad_x / pad_y: REFLECT_101
// pixels written with the upload / by pad_reflect_kernel, zero derivatives), so every window that passes the bounds test is plain
// memory: one wave-uniform plane base in SGPRs + an unsigned 32-bit lane offset (global_load ... v, s[base]); there is no
// reflection code.  The run-time-window forms also load (and mask) the rows of the taps below a short window: up to 16 rows
// past the bottom border, which the batch's allocation carries as slack.
//
// Summation order (defines bit-exactness against the CPU checker's kernel order): every term is an integer; the taps of one
// lane are summed exactly in int32 (<= 16 taps of < 2^25.1) and converted to f32 once into p[lane] (OpenCV converts every term
// and rounds at every add: the kernel's sum is the more accurate one, tests bound the difference); 64-lane groups:
// a[l] = p[l] + p[l+32]; 64- and 32-lane groups: c[l] = a[l] + a[l+16]; all: an xor butterfly with strides 1,2,4,8 over the 16
// values.  No FMA contraction (-ffp-contract=off), IEEE divide/sqrt (hipcc default).
#include "dr3hip_internal.h"
#include "wave_tree.h"

// This file is compiled TWICE (Makefile): LKG_TU = 0, the tracker's translation unit proper, and LKG_TU = 1, the launches of the forms
// that hold eight points per wave (windows up to 16 x 16: the headline's 15 x 15), built with LLVM's max-ILP machine scheduler
// (-mllvm -amdgpu-sched-strategy=max-ilp: 0.865 -> 0.82 ms per launch of the headline shape on the same box; the quarter-wave 21 x 21 form
// loses 1.4 % under it and the warp kernels 3 - 13 %, so only these forms get it -- profiles/r04_lk/README.md section 9).
#ifndef LKG_TU
#define LKG_TU 0
#endif

namespace dr3 {
namespace {

constexpr int W_BITS = 14;
constexpr int RM = 4;              // search margin (px) of the staged J region

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

// A bilinear tap: (I00*iw00 + I01*iw01 + I10*iw10 + I11*iw11 + acc) >> (W_BITS - 5), two v_dot2_i32_i16 on the byte pairs of rows
// y and y + 1.  acc = BILIN_ROUND gives CV_DESCALE(..., W_BITS - 5); acc = BILIN_ROUND - (t << (W_BITS - 5)) gives that minus t
// (the subtrahend is a multiple of the shift's divisor, so floor commutes with it).  The dots are SIGNED: iw11 = 2^14 - iw00 -
// iw01 - iw10 is -1 or -2 when both fractions are a few 1e-5 (the three rounded weights then sum to more than 2^14); OpenCV carries
// it as a negative int, so must the packed 16-bit lane.
constexpr int BILIN_ROUND = 1 << (W_BITS - 5 - 1);
// The saturating form of v_dot2_i32_i16 is the three-operand encoding: the accumulator is a free source instead of
// being tied to the destination (the non-saturating builtin selects v_dot2c and a v_mov per use).  No sum here can
// reach the int32 limits (bilinear: < 2^31 / 64; window sums: <= 16 taps of < 2^25.1), so saturation never acts.
constexpr bool DOT_SAT = true;
// (lo | hi << 16) of two values that fit int16
__device__ __forceinline__ uint32_t pack_s16(int lo, int hi)
{
    return __builtin_amdgcn_perm((uint32_t)hi, (uint32_t)lo, 0x05040100u);
}
__device__ __forceinline__ int dot_s16(uint32_t a, uint32_t b, int acc)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b), acc, DOT_SAT);
}

// 8-lane groups: the xor butterfly 1, 2, 4 inside half a DPP row
__device__ __forceinline__ float half_row_butterfly(float v)
{
    v += dpp_xchg<0xB1>(v);     // quad_perm [1,0,3,2]   (xor 1)
    v += dpp_xchg<0x4E>(v);     // quad_perm [2,3,0,1]   (xor 2)
    v += dpp_xchg<0x141>(v);    // row_half_mirror       (= xor 4 once quads are uniform)
    return v;
}
// sums over each group of two / three lane values; every lane receives its own group's totals
template <int PPW>
__device__ __forceinline__ void group_sum2(float a, float b, float& ra, float& rb)
{
    if (PPW == 8) { ra = half_row_butterfly(a); rb = half_row_butterfly(b); return; }
    if (PPW == 4) { ra = row_butterfly(a); rb = row_butterfly(b); return; }
    if (PPW == 1) { wave_sum2(a, b, ra, rb); return; }
    const float v = row_butterfly(fold16(a, b));                 // row totals: a.A  b.A | a.B  b.B
    const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    ra = __uint_as_float(sw[0]);                                 // a.A  a.A | a.B  a.B
    rb = __uint_as_float(sw[1]);                                 // b.A  b.A | b.B  b.B
}
template <int PPW>
__device__ __forceinline__ float group_sum1(float c)
{
    if (PPW == 8) return half_row_butterfly(c);
    if (PPW == 4) return row_butterfly(c);
    if (PPW == 1) return wave_sum1(c);
    return row_butterfly(fold16(c, c));
}

// wave-uniform global base + unsigned 32-bit lane offset: selects the saddr form of global_load
typedef const __attribute__((address_space(1))) uint8_t* gbase_t;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
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

__device__ __forceinline__ bool any_lane(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

#ifndef LKG_WAVES
#define LKG_WAVES 4
#endif
#ifndef LKG_RB
#define LKG_RB 9            // window rows whose LDS reads are issued together in an iteration
#endif
#ifndef LKG_WAVES_NI2
#define LKG_WAVES_NI2 4
#endif
#ifndef LKG_WAVES_NI4
#define LKG_WAVES_NI4 3       // four points per wave, four 8-row items per lane (21 x 21)
#endif
#ifndef LKG_WAVES_P8
#define LKG_WAVES_P8 3       // eight points per wave: two 16-tap columns per lane
#endif
#ifndef LKG_WIDE
#define LKG_WIDE 1           // the J region and the template's image rows are staged with 8-byte loads, several rows per instruction
                             // (1 = on).  The CU's texture path takes ~16 clocks per wave-instruction whatever its width
                             // (tools/ubench/vmem_rate.hip), and the tracker's per-level time is its vector-memory instructions times that
#endif
#ifndef LKG_PRE
#define LKG_PRE 1
#endif
#ifndef LKG_PPW16
#define LKG_PPW16 8          // points per wave for windows up to 16 x 16: 8 (two columns per lane) or 4 (one)
#endif
// WIN > 0: square compile-time window; WIN = 0: run-time window of the group shape PPW x NI (lk_group_shape below)
template <int WIN, int PPW, int NI>
// (the run-time forms holding 16 taps per lane need more registers than four waves per SIMD leave them)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PPW == 8 ? (WIN ? LKG_WAVES_P8 : 2) : NI == 4 ? LKG_WAVES_NI4 : (WIN == 0 && NI * (PPW == 4 ? 16 : 8) == 16) ? 3 : (NI == 2 ? LKG_WAVES_NI2 : LKG_WAVES)))) void lk_trackg_kernel(const LKArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_all[];
    constexpr int LANES = 64 / PPW;               // lanes per point
    constexpr bool WIDE = LKG_WIDE != 0;
    constexpr bool PRE = WIDE && WIN != 0 && LKG_PRE;          // the template's derivative loads are issued with the staging loads
#ifndef LKG_PRE_ITEMS4
#define LKG_PRE_ITEMS4 4
#endif
    constexpr int PRE_ITEMS = NI == 4 ? LKG_PRE_ITEMS4 : NI;  // ... those of the first PRE_ITEMS lane-items (four items of 9 rows are 72 registers)
    // COLF: the column form -- a lane-item is a whole window column (<= 16 taps): an eighth of a wave with two columns per lane, or a
    // quarter with one.  Otherwise items are 8 rows of a column; <21, 4, 4> is that form on a QUARTER wave, four items per lane
    constexpr bool COLF = PPW == 8 || (PPW == 4 && NI == 1);
    constexpr int NT = COLF ? 16 : 8;             // taps (consecutive window rows of one column) per lane-item
    // dword columns of the staging grid (>= the region's width: window + 1 + 2 RM + 3 bytes of alignment); an even number (8-byte pieces)
    constexpr int COLS = COLF ? 8 : (WIN && PPW == 4) ? (((WIN + 2 * RM + 4 + 3) / 4 + 1) & ~1) : 16;
    constexpr int SR = LANES / COLS > 0 ? LANES / COLS : 1;      // rows a group stages per round (the narrow staging only)
    // WIDE staging: a row is fetched as PCS pieces of 8 bytes (4 COLS bytes: the J region's row and the template's image row both fit),
    // a round of the group's lanes covers RPR rows
    constexpr int PCS = COLS / 2, RPR = LANES / PCS, IPITCH = COLS + 1;
    constexpr bool SPARE = RPR * PCS != LANES;    // lanes behind the last whole row of a round (16 lanes, 5 pieces): they load, but do not store
    // staged template image rows per point: the rows its lane-items read (a whole column + 1, or 8-row items up to 8 nblk + 1), in whole rounds
    constexpr int IROWS = ((WIN ? (COLF ? WIN + 1 : 8 * ((WIN + 7) / 8) + 1) : (COLF ? 17 : 33)) + RPR - 1) / RPR * RPR;
    static_assert((PPW != 4 || NI == 1 || NI == 4) && (PPW != 8 || NI == 2), "a quarter wave holds one window column per lane (or four 8-row items), an eighth two columns");

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
    const int nblk = COLF ? 1 : (win_h + 7) >> 3;          // lane-items per window column
    const int rwd = (win_w + 15) >> 2, rh = win_h + 9, lpd = WIDE ? COLS + 1 : rwd | 1;    // staged J region: dwords per row, rows, LDS row pitch (odd; WIDE: PCS pieces of 2 dwords + 1)
    // LDS rows per group: staging runs in rounds of SR rows.  Every form but the compile-time column forms (WIN with PPW >= 4, whose
    // tap_live() stops at the window) also reads (and masks) the rows of the taps below the window: a lane-item of NT rows starting at
    // ity, up to 8 rows into the region, reaches row 8 + ity + NT <= win_h + 15 of the slice
    constexpr int SRW = WIDE ? RPR : SR;          // rows per staging round
    const int rhs = (((WIN && COLF) ? rh : COLF ? 25 : win_h + 16) + SRW - 1) / SRW * SRW;
    static_assert(!(WIN && !COLF) || 8 + 8 * ((WIN + 7) / 8 - 1) + 8 <= WIN + 15, "the slice of a group holds every row its lane-items read");
    const float half_x = (win_w - 1) * 0.5f, half_y = (win_h - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (1 << 20);
    const float eps_lo = (float)(a.epsilon * (1.0 - 1e-6)), eps_hi = (float)(a.epsilon * (1.0 + 1e-6));
    uint32_t* __restrict__ lds = lds_all + grp * (rhs * lpd);
    uint32_t* __restrict__ ldsI = lds_all + PPW * (rhs * lpd) + grp * (IROWS * IPITCH);     // WIDE: the point's template image rows

    // lane-item geometry: item q = blk * win_w + c covers the taps (8 blk .. 8 blk + 7, c) -- windows up to 16 x 16: the whole column c --
    // and the lane owns the items hl, hl + LANES, ...; slots beyond the last item shadow item 0 with all-zero masks
    int itx[NI], ity[NI], ntap[NI];
    uint32_t loffB[NI];                                    // byte offset of the lane-item's first row inside a staged region
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int q = hl + i * LANES;
        const bool ivalid = q < win_w * nblk;
        const int blk = (COLF || !ivalid) ? 0 : (q >= win_w) + (q >= 2 * win_w) + (q >= 3 * win_w);
        itx[i] = ivalid ? q - blk * win_w : 0;
        ity[i] = 8 * blk;
        loffB[i] = (uint32_t)(ity[i] * lpd) * 4u;
        ntap[i] = ivalid ? win_h - ity[i] : 0;             // taps of this item inside the window (>= NT: all), 0 for idle slots
    }
    // int16-pair mask of taps (2h, 2h+1) of item i: all ones, the low half only, or nothing.  With a compile-time window and one
    // item per column the pattern is a constant per h, and the idle lanes are removed from the lane sums instead (lane_mask)
    constexpr bool LANE_MASKED = WIN && PPW == 4 && NI == 1;
    const int lane_mask = ntap[0] ? -1 : 0;
    // an eighth of a wave with a compile-time window: the pattern is a constant per h as well, but the lane's second column may lie
    // beyond the window (column 15 of a 15-wide one): its gradients are zeroed through item_mask
    constexpr bool ITEM_MASKED = WIN && PPW == 8;
    uint32_t item_mask[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) item_mask[i] = ntap[i] ? 0xffffffffu : 0u;
    auto pair_mask = [&ntap, &item_mask](int i, int h) -> uint32_t {
        if (LANE_MASKED) return WIN >= 2 * h + 2 ? 0xffffffffu : (WIN == 2 * h + 1 ? 0x0000ffffu : 0u);
        if (ITEM_MASKED) return (WIN >= 2 * h + 2 ? 0xffffffffu : (WIN == 2 * h + 1 ? 0x0000ffffu : 0u)) & item_mask[i];
        return ntap[i] >= 2 * h + 2 ? 0xffffffffu : (ntap[i] == 2 * h + 1 ? 0x0000ffffu : 0u);
    };
    // a quarter wave with a compile-time window: the taps below the window are not computed at all
    auto tap_live = [](int k) -> bool { return !(WIN && COLF) || k < WIN; };
    // staging grid of the group: SR rows x COLS dword columns per round
    const int sc = hl & (COLS - 1), sr = hl / COLS;
    uint32_t* __restrict__ lds_st = lds + sr * lpd + sc;
    // WIDE: the J region rows [ry, ry + rh) x bytes [rx, rx + 32) of the current level into the group's slice: lane = (row of a pair, 8-byte piece)
    gbase_t Jb = nullptr; int Jpitch = 0, Jpad_x = 0, Jpad_y = 0;            // set per level
    auto stage_J_wide = [&](int rx, int ry) {
        const int rr = hl / PCS, pc = hl - rr * PCS;
        const bool stg = !SPARE || rr < RPR;
        uint32_t off = __umul24((unsigned)(ry + Jpad_y + rr), (unsigned)Jpitch) + (unsigned)(rx + Jpad_x + 8 * pc);
        uint32_t* st = lds + rr * lpd + 2 * pc;
        if (WIN) {
            constexpr int JR = (WIN + 9 + RPR - 1) / RPR;                    // rows up to JR RPR - 1 < rhs exist in memory: slack rows
            u32x2 v[JR];
#pragma unroll
            for (int k = 0; k < JR; ++k) { v[k] = ldg<u32x2>(Jb, off); off += (unsigned)RPR * (unsigned)Jpitch; }
            if (stg) {
#pragma unroll
                for (int k = 0; k < JR; ++k) { st[RPR * k * lpd] = v[k].x; st[RPR * k * lpd + 1] = v[k].y; }
            }
        } else {
            for (int r = rr; r < rh && stg; r += RPR) { const u32x2 v = ldg<u32x2>(Jb, off); st[(r - rr) * lpd] = v.x; st[(r - rr) * lpd + 1] = v.y; off += (unsigned)RPR * (unsigned)Jpitch; }
        }
    };

    int   lost = 0;                // 0 = tracked, else the DR3HIP_LK_LOST_* reason (status = !lost)
    float err = 0.f;

    for (int level = a.max_level; level >= 0; --level) {
        const LevelDev& L = a.lv[level];
        const int Lw = L.w, Lh = L.h, pitch = L.pitch, dpitch = L.dpitch, pad_x = L.pad_x, pad_y = L.pad_y;
        // padded-plane origins of this pair's planes (wave-uniform)
        const long long border = (long long)pad_y * pitch + pad_x;
        const gbase_t Ibase = as_global(L.img + (long long)ps * L.step - border);
        const gbase_t Jbase = as_global(L.img + (long long)ns * L.step - border);
        Jb = Jbase; Jpitch = pitch; Jpad_x = pad_x; Jpad_y = pad_y;
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
        int it_count = 0, restages = 0;          // restages: how often an iterate left the staged J region (diagnostic, DR3HIP_LK_DIAG_RESTAGES)

        px -= half_x; py -= half_y;
        const float fpx = floorf(px), fpy = floorf(py);
        const int ipx = floor_to_int(fpx), ipy = floor_to_int(fpy);
        // -win <= ip < size  <=>  (unsigned)(ip + win) < (unsigned)(size + win)
        bool act = (unsigned)(ipx + win_w) < (unsigned)(Lw + win_w) && (unsigned)(ipy + win_h) < (unsigned)(Lh + win_h);
        if (!act && level == 0) { lost = DR3HIP_LK_LOST_START_OUTSIDE; err = 0.f; }
        if (!any_lane(act)) {
            if (a.iters && hl == 0 && pt_ok) a.iters[pidx * (a.max_level + 1) + level] = 0;
            continue;
        }
        cx -= half_x; cy -= half_y;

        // ---- stage the J search region around the starting estimate: rows [sy-RM, sy-RM+rh) x aligned byte columns
        // [x0, x0 + 4*rwd).  A start outside the level is rejected by the first iteration's bounds test; the staging
        // centre is clamped so that the (unused) region stays inside the padded plane.
        int rx0, ry0;                      // interior coordinates of LDS (row 0, byte 0) of this group's region; rx0 is a multiple of 4
        u32x2 ddv[PRE ? PRE_ITEMS : 1][PRE ? NT + 1 : 1];       // PRE: the template's derivative rows, loaded ahead
        {
            const int sx = floor_to_int(floorf(cx)), sy = floor_to_int(floorf(cy));
            const int cxs = min(max(sx, -win_w), Lw - 1), cys = min(max(sy, -win_h), Lh - 1);
            rx0 = (cxs - RM) & ~3;
            ry0 = cys - RM;
            if constexpr (WIDE) {
                // the template's image rows first (they depend on the previous point only): bytes (xp0 & ~3) .. + 4 COLS of IROWS rows, a lane =
                // (row of the round, 8-byte piece).  Then the J region the same way.
                const int xp0 = (act ? ipx : 0) + pad_x, yp0 = (act ? ipy : 0) + pad_y;
                const int rr = hl / PCS, pc = hl - rr * PCS;
                const bool stg = !SPARE || rr < RPR;
                uint32_t offI = __umul24((unsigned)(yp0 + rr), (unsigned)pitch) + ((unsigned)xp0 & ~3u) + 8u * (unsigned)pc;
                uint32_t* stI = ldsI + rr * IPITCH + 2 * pc;
                u32x2 vi[IROWS / RPR];
                // run-time windows: IROWS is the count for the largest window of the shape; rows past those the lane-items of THIS window
                // read (the plane's slack below its bottom border covers exactly those) are replaced by the last such row
                const int irows_need = COLF ? NT + 1 : 8 * nblk + 1;
#pragma unroll
                for (int k = 0; k < IROWS / RPR; ++k) {
                    if (WIN) { vi[k] = ldg<u32x2>(Ibase, offI); offI += (unsigned)RPR * (unsigned)pitch; }
                    else vi[k] = ldg<u32x2>(Ibase, __umul24((unsigned)(yp0 + min(RPR * k + rr, irows_need - 1)), (unsigned)pitch) + ((unsigned)xp0 & ~3u) + 8u * (unsigned)pc);
                }
                if constexpr (PRE) {
                    // compile-time window: the J region's loads, then the template's derivative loads (they wait for nothing either), and only
                    // then the LDS writes of the staged rows -- loads return in order, so waiting for the staged rows leaves the derivative
                    // rows in flight: ONE memory round trip per level instead of two
                    constexpr int JR = (WIN + 9 + RPR - 1) / RPR;
                    uint32_t off = __umul24((unsigned)(ry0 + pad_y + rr), (unsigned)pitch) + (unsigned)(rx0 + pad_x + 8 * pc);
                    uint32_t* st = lds + rr * lpd + 2 * pc;
                    u32x2 vj[JR];
#pragma unroll
                    for (int k = 0; k < JR; ++k) { vj[k] = ldg<u32x2>(Jbase, off); off += (unsigned)RPR * (unsigned)pitch; }
#pragma unroll
                    for (int i = 0; i < PRE_ITEMS; ++i) {
                        const uint32_t offD = __umul24((unsigned)(yp0 + ity[i]), (unsigned)dpitch) + 4u * (unsigned)(xp0 + itx[i]);
                        gbase_t rowD = Dbase;
#pragma unroll
                        for (int r = 0; r <= NT; ++r)
                            if (tap_live(r - 1) || tap_live(r)) {
#ifdef LKG_EXP_NODERIV      // TIMING EXPERIMENT ONLY (wrong results): the per-level phase without its 32 derivative loads -- the ceiling of what
                            // computing Scharr inside the tracker could save (profiles/r04_lk/README.md)
                                ddv[i][r] = vj[(r + i) % JR];
#else
                                ddv[i][r] = ldg<u32x2>(rowD, offD); rowD += dpitch;
#endif
                            }
                    }
                    __builtin_amdgcn_sched_barrier(0);           // the scheduler would sink the loads to their uses behind the LDS writes
#ifdef LKG_EXP_FILLER       // TIMING EXPERIMENT ONLY: LKG_EXP_FILLER extra 4-clock VALU instructions per lane and level on the staged image data -- what
                            // computing the template's Scharr gradients inside the tracker would add (profiles/r04_lk/README.md)
                    {
                        uint32_t f0 = vi[0].x, f1 = vi[0].y, f2 = vi[1].x, f3 = vi[1].y;
#pragma unroll
                        for (int q = 0; q < LKG_EXP_FILLER / 4; ++q) {
                            f0 = __builtin_amdgcn_perm(f0, f1, 0x06050403u); f1 = __builtin_amdgcn_perm(f1, f2, 0x06050403u);
                            f2 = __builtin_amdgcn_perm(f2, f3, 0x06050403u); f3 = __builtin_amdgcn_perm(f3, f0, 0x06050403u);
                        }
                        asm volatile("" :: "v"(f0), "v"(f1), "v"(f2), "v"(f3));
                    }
#endif
                    if (stg) {
#pragma unroll
                        for (int k = 0; k < JR; ++k) { st[RPR * k * lpd] = vj[k].x; st[RPR * k * lpd + 1] = vj[k].y; }
                    }
                } else {
                    stage_J_wide(rx0, ry0);
                }
                if (stg) {
#pragma unroll
                    for (int k = 0; k < IROWS / RPR; ++k) { stI[RPR * k * IPITCH] = vi[k].x; stI[RPR * k * IPITCH + 1] = vi[k].y; }
                }
            } else {
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
            }
            __builtin_amdgcn_wave_barrier();
        }

        // ---- template patch + structure tensor ----------------------------------------------------
        // Row r of a lane-item: the image byte pair (c, c + 1) and the derivative dword pair of the same columns, one 8-byte load
        // each; the 16 lanes of a column group read adjacent addresses.  Tap r blends rows r and r + 1: every row is permuted once.
        int Ib[NI][NT];                              // per tap: the dot accumulator BILIN_ROUND - (I << 9)
        uint32_t Ix[NI][NT / 2], Iy[NI][NT / 2];     // gradients as int16 pairs of taps (2h, 2h + 1)
        // Every lane computes it -- a point outside the level (act false) reads the window at the origin instead and its sums are
        // dropped -- so that the patch registers are defined afresh on every level and nothing is carried around the level loop.
        int s11 = 0, s12 = 0, s22 = 0;
        {
            const Weights W = make_weights(px - fpx, py - fpy);
            const int ipxs = act ? ipx : 0, ipys = act ? ipy : 0;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int xp = ipxs + pad_x + itx[i], yp = ipys + pad_y + ity[i];
                const uint32_t selI = 0x0c010c00u + ((uint32_t)xp & 3u) * 0x00010001u;
                const uint32_t offI = __umul24((unsigned)yp, (unsigned)pitch) + ((unsigned)xp & ~3u);
                const uint32_t offD = __umul24((unsigned)yp, (unsigned)dpitch) + 4u * (unsigned)xp;
                uint32_t Pp = 0, Xp = 0, Yp = 0;
                int gx[2], gy[2];
                gbase_t rowI = Ibase, rowD = Dbase;          // wave-uniform row bases, advanced on the scalar unit
                // WIDE: the image bytes come out of the staged rows: byte (first column & 3) + column of the row
                const uint32_t* pI = ldsI + ity[i] * IPITCH + (((((uint32_t)xp - (uint32_t)itx[i]) & 3u) + (uint32_t)itx[i]) >> 2);
#pragma unroll
                for (int r = 0; r <= NT; ++r) {
                    uint32_t P = 0, X = 0, Y = 0;
                    if (tap_live(r - 1) || tap_live(r)) {
                        u32x2 im;
                        if constexpr (WIDE) { im.x = pI[r * IPITCH]; im.y = pI[r * IPITCH + 1]; }
                        else { im = ldg<u32x2>(rowI, offI); rowI += pitch; }
                        u32x2 dd;
                        if (PRE && i < PRE_ITEMS) dd = ddv[i < PRE_ITEMS ? i : 0][r];
                        else { dd = ldg<u32x2>(rowD, offD); rowD += dpitch; }
                        P = __builtin_amdgcn_perm(im.y, im.x, selI);
                        X = __builtin_amdgcn_perm(dd.y, dd.x, 0x05040100u);
                        Y = __builtin_amdgcn_perm(dd.y, dd.x, 0x07060302u);
                    }
                    if (r > 0) {
                        const int k = r - 1;
                        if (tap_live(k)) {
                            Ib[i][k] = BILIN_ROUND - ((dot_s16(P, W.w23, dot_s16(Pp, W.w01, BILIN_ROUND)) >> (W_BITS - 5)) << (W_BITS - 5));
                            gx[k & 1] = dot_s16(X, W.w23, dot_s16(Xp, W.w01, 1 << (W_BITS - 1))) >> W_BITS;
                            gy[k & 1] = dot_s16(Y, W.w23, dot_s16(Yp, W.w01, 1 << (W_BITS - 1))) >> W_BITS;
                        } else {
                            Ib[i][k] = 0; gx[k & 1] = 0; gy[k & 1] = 0;
                        }
                        if (k & 1) {
                            // taps below the window's last row (and idle slots) get zero gradients through the pair masks, which
                            // removes them from every sum
                            const uint32_t mkh = pair_mask(i, k >> 1);
                            const uint32_t x2 = pack_s16(gx[0], gx[1]) & mkh, y2 = pack_s16(gy[0], gy[1]) & mkh;
                            Ix[i][k >> 1] = x2; Iy[i][k >> 1] = y2;
                            s11 = dot_s16(x2, x2, s11);
                            s12 = dot_s16(x2, y2, s12);
                            s22 = dot_s16(y2, y2, s22);
                        }
                    }
                    Pp = P; Xp = X; Yp = Y;
                }
            }
        }
        if (LANE_MASKED) { s11 &= lane_mask; s12 &= lane_mask; s22 &= lane_mask; }
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
                if (level == 0) lost = DR3HIP_LK_LOST_MIN_EIG;
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
                    if (level == 0) lost = DR3HIP_LK_LOST_ITERATE_OUTSIDE;
                    act = false;
                }
            }
            if (!any_lane(act)) break;
            int s1 = 0, s2 = 0;
            if (act) {
                it_count = j + 1;
                uint32_t dxb = (uint32_t)((inx & ~3) - rx0), dyr = (uint32_t)(iny - ry0);     // aligned byte column / row inside the region
                if (dxb > 8u || dyr > 8u) {
                    // the iterate left the staged region (margin RM px): stage it again around the iterate
                    rx0 = (inx - RM) & ~3;
                    ry0 = iny - RM;
                    if constexpr (WIDE) stage_J_wide(rx0, ry0);
                    else {
                    uint32_t off = __umul24((unsigned)(ry0 + pad_y + sr), (unsigned)pitch) + (unsigned)(rx0 + pad_x + 4 * sc);
                    if (sc < rwd)
                        for (int r = sr; r < rh; r += SR) { lds_st[(r - sr) * lpd] = ldg<uint32_t>(Jbase, off); off += (uint32_t)SR * (unsigned)pitch; }
                    }
                    __builtin_amdgcn_wave_barrier();
                    dxb = (uint32_t)((inx & ~3) - rx0); dyr = (uint32_t)(iny - ry0);
                    ++restages;
                }
                const Weights W = make_weights(cx - fcx, cy - fcy);
                const uint32_t base = __umul24(dyr, (unsigned)(4 * lpd));
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const uint32_t t = (uint32_t)(inx - rx0) + (uint32_t)itx[i];          // byte column of the item inside the region
                    const uint32_t sel = 0x0c010c00u + (t & 3u) * 0x00010001u;
                    const uint32_t* p = reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(lds) + (base + loffB[i] + (t & ~3u)));
                    uint32_t Pp = 0;
                    int m[2];
                    uint32_t lo[NT + 1], hi[NT + 1];
#pragma unroll
                    for (int r = 0; r <= NT; ++r) {
                        if (r % LKG_RB == 0) {
                            // the LDS reads of the next LKG_RB rows go out together, ahead of the arithmetic on them
#pragma unroll
                            for (int q = r; q < r + LKG_RB && q <= NT; ++q)
                                if (tap_live(q - 1) || tap_live(q)) { lo[q] = p[q * lpd]; hi[q] = p[q * lpd + 1]; }
                            asm volatile("" : "+v"(lo[r]));
                        }
                        uint32_t P = 0;
                        if (tap_live(r - 1) || tap_live(r)) P = __builtin_amdgcn_perm(hi[r], lo[r], sel);
                        if (r > 0) {
                            const int k = r - 1;
                            m[k & 1] = tap_live(k) ? dot_s16(P, W.w23, dot_s16(Pp, W.w01, Ib[i][k])) >> (W_BITS - 5) : 0;
                            if (k & 1) {
                                const uint32_t mm = pack_s16(m[0], m[1]);
                                s1 = dot_s16(mm, Ix[i][k >> 1], s1);
                                s2 = dot_s16(mm, Iy[i][k >> 1], s2);
                            }
                        }
                        Pp = P;
                    }
                }
            }
            if (LANE_MASKED) { s1 &= lane_mask; s2 &= lane_mask; }
            float b1, b2;
            group_sum2<PPW>((float)s1, (float)s2, b1, b2);
            if (act) {
                b1 *= FLT_SCALE; b2 *= FLT_SCALE;
                const float dx = (A12 * b2 - A22 * b1) * D;
                const float dy = (A12 * b1 - A11 * b2) * D;
                cx += dx; cy += dy;
                nx = cx + half_x; ny = cy + half_y;
                // OpenCV tests delta.ddot(delta) <= epsilon in double.  The f32 value of dx^2 + dy^2 is within 3 ulp of that double sum, so
                // outside [eps_lo, eps_hi] = epsilon (1 -+ 1e-6) it decides the same way; inside, and where the f32 squares may have underflowed,
                // the double sequence runs
                const float d2 = dx * dx + dy * dy;
                bool conv = d2 <= eps_lo;
                if ((d2 > eps_lo && d2 < eps_hi) || d2 < 1e-30f) conv = (double)dx * (double)dx + (double)dy * (double)dy <= a.epsilon;
                if (conv) {
                    act = false;
                } else if (j > 0 && fabsf(dx + pdx) <= 0.01f && fabsf(dy + pdy) <= 0.01f) {
                    // std::abs(float) < 0.01 (a double) holds exactly for the floats <= 0.01f, the largest float below 0.01
                    nx -= dx * 0.5f; ny -= dy * 0.5f;
                    act = false;
                }
                pdx = dx; pdy = dy;
            }
        }
        if (a.iters && hl == 0 && pt_ok) a.iters[pidx * (a.max_level + 1) + level] = it_count | ((a.flags & DR3HIP_LK_DIAG_RESTAGES) ? restages << 16 : 0);

        // ---- residual (level 0 only) -----------------------------------------------------------------
        if (level == 0 && !(a.flags & DR3HIP_LK_GET_MIN_EIGENVALS)) {
            bool res = lost == 0;
            const float fx = nx - half_x, fy = ny - half_y;
            const float ffx = floorf(fx), ffy = floorf(fy);
            const int inx = floor_to_int(ffx), iny = floor_to_int(ffy);
            if (res && !((unsigned)(inx + win_w) < (unsigned)(Lw + win_w) && (unsigned)(iny + win_h) < (unsigned)(Lh + win_h))) { lost = DR3HIP_LK_LOST_RESULT_OUTSIDE; res = false; }
            int se = 0;
            if (res) {
                uint32_t dxb = (uint32_t)((inx & ~3) - rx0), dyr = (uint32_t)(iny - ry0);
                if (dxb > 8u || dyr > 8u) {
                    rx0 = (inx - RM) & ~3;
                    ry0 = iny - RM;
                    if constexpr (WIDE) stage_J_wide(rx0, ry0);
                    else {
                    uint32_t off = __umul24((unsigned)(ry0 + pad_y + sr), (unsigned)pitch) + (unsigned)(rx0 + pad_x + 4 * sc);
                    if (sc < rwd)
                        for (int r = sr; r < rh; r += SR) { lds_st[(r - sr) * lpd] = ldg<uint32_t>(Jbase, off); off += (uint32_t)SR * (unsigned)pitch; }
                    }
                    __builtin_amdgcn_wave_barrier();
                    dxb = (uint32_t)((inx & ~3) - rx0); dyr = (uint32_t)(iny - ry0);
                }
                const Weights W = make_weights(fx - ffx, fy - ffy);
                const uint32_t base = __umul24(dyr, (unsigned)(4 * lpd));
                // sum of |mismatch| over the taps inside the window: |d| <= 2^13 packed in int16 pairs, dotted with the pair
                // masks reduced to ones (exact integer sum of at most 16 taps)
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const uint32_t t = (uint32_t)(inx - rx0) + (uint32_t)itx[i];
                    const uint32_t sel = 0x0c010c00u + (t & 3u) * 0x00010001u;
                    const uint32_t* p = reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(lds) + (base + loffB[i] + (t & ~3u)));
                    uint32_t Pp = 0;
                    int m[2];
#pragma unroll
                    for (int r = 0; r <= NT; ++r) {
                        uint32_t P = 0;
                        if (tap_live(r - 1) || tap_live(r)) P = __builtin_amdgcn_perm(p[r * lpd + 1], p[r * lpd], sel);
                        if (r > 0) {
                            const int k = r - 1;
                            const int d = tap_live(k) ? dot_s16(P, W.w23, dot_s16(Pp, W.w01, Ib[i][k])) >> (W_BITS - 5) : 0;
                            m[k & 1] = d < 0 ? -d : d;
                            if (k & 1) se = dot_s16(pack_s16(m[0], m[1]), pair_mask(i, k >> 1) & 0x00010001u, se);
                        }
                        Pp = P;
                    }
                }
            }
            if (LANE_MASKED) se &= lane_mask;
            const float tot = group_sum1<PPW>((float)se);
            if (res) err = tot * 1.f / (float)(32 * win_w * win_h);
        }
    }

    if (hl == 0 && pt_ok) {
        a.next_xy[2 * pidx] = nx;
        a.next_xy[2 * pidx + 1] = ny;
        a.status[pidx] = (uint8_t)(lost == 0);
        if (a.reason) a.reason[pidx] = (uint8_t)lost;
        a.err[pidx] = err;
    }
}


#if LKG_TU == 0
// ---- windows beyond 32 x 32 (up to DR3HIP_MAX_WIN = 64): the plain form ----------------------------------------------------------------
// cv::calcOpticalFlowPyrLK has no window limit; the reference's own 30 x 30 and every BASELINE configuration run on the kernel above,
// and this one only removes the limit: ONE wave per point, the same cut of the window into lane-items of 8 rows of a column
// (item q = (ty / 8) * win_w + tx on lane q % 64 -- the summation order the CPU checker's kernel order states for any window), the
// interpolated template (I, Ix, Iy as int16) in LDS, the search image read from the padded planes tap by tap.  Lane sums in int64
// (a lane may hold 64 taps of 2^25), converted to f32 once, then the wave tree.  Not tuned.
__global__ __launch_bounds__(64) void lk_track_big_kernel(const LKArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_all[];
    int16_t* __restrict__ tI = reinterpret_cast<int16_t*>(lds_all);                  // [win_h * win_w]
    const int win_w = a.win_w, win_h = a.win_h, n_taps = win_w * win_h;
    int16_t* __restrict__ tX = tI + ((n_taps + 7) & ~7);
    int16_t* __restrict__ tY = tX + ((n_taps + 7) & ~7);
    const int lane = threadIdx.x, pair = blockIdx.y, pt = blockIdx.x;
    const long long pidx = (long long)pair * a.n_pts + pt;
    const int ps = a.prev_slot ? a.prev_slot[pair] : a.first_slot + pair;
    const int ns = a.next_slot ? a.next_slot[pair] : a.first_slot + pair + 1;
    const float ppx = a.prev_xy[2 * pidx], ppy = a.prev_xy[2 * pidx + 1];
    float nx = a.next_xy[2 * pidx], ny = a.next_xy[2 * pidx + 1];
    const int nblk = (win_h + 7) >> 3, n_items = win_w * nblk;
    const float half_x = (win_w - 1) * 0.5f, half_y = (win_h - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (1 << 20);
    int lost = 0;
    float err = 0.f;

    for (int level = a.max_level; level >= 0; --level) {
        const LevelDev& L = a.lv[level];
        const int Lw = L.w, Lh = L.h, pitch = L.pitch, dpitch = L.dpitch;
        const uint8_t* __restrict__ Ip = L.img + (long long)ps * L.step;               // interior origins: negative offsets reach the border
        const uint8_t* __restrict__ Jp = L.img + (long long)ns * L.step;
        const int16_t* __restrict__ Dp = reinterpret_cast<const int16_t*>(reinterpret_cast<const uint8_t*>(L.der) + (long long)ps * L.dstep);
        const float scale = __int_as_float((127 - level) << 23);
        float px = ppx * scale, py = ppy * scale, cx, cy;
        if (level == a.max_level) {
            if (a.flags & DR3HIP_LK_USE_INITIAL_FLOW) { cx = nx * scale; cy = ny * scale; } else { cx = px; cy = py; }
        } else { cx = nx * 2.f; cy = ny * 2.f; }
        nx = cx; ny = cy;
        int it_count = 0;
        px -= half_x; py -= half_y;
        const float fpx = floorf(px), fpy = floorf(py);
        const int ipx = floor_to_int(fpx), ipy = floor_to_int(fpy);
        const bool inside = (unsigned)(ipx + win_w) < (unsigned)(Lw + win_w) && (unsigned)(ipy + win_h) < (unsigned)(Lh + win_h);
        if (!inside) {                                                                  // wave-uniform
            if (level == 0) { lost = DR3HIP_LK_LOST_START_OUTSIDE; err = 0.f; }
            if (a.iters && lane == 0) a.iters[pidx * (a.max_level + 1) + level] = 0;
            continue;
        }
        // ---- template + structure tensor
        long long s11 = 0, s12 = 0, s22 = 0;
        {
            const Weights W = make_weights(px - fpx, py - fpy);
            const int w00 = (int16_t)(W.w01 & 0xffffu), w01 = (int16_t)(W.w01 >> 16), w10 = (int16_t)(W.w23 & 0xffffu), w11 = (int16_t)(W.w23 >> 16);
            for (int q = lane; q < n_items; q += 64) {
                const int blk = q / win_w, tx = q - blk * win_w;
                const int y1 = min(8 * blk + 8, win_h);
                for (int ty = 8 * blk; ty < y1; ++ty) {
                    const uint8_t* r0 = Ip + (long long)(ipy + ty) * pitch + (ipx + tx);
                    const uint8_t* r1 = r0 + pitch;
                    const int16_t* d0 = reinterpret_cast<const int16_t*>(reinterpret_cast<const uint8_t*>(Dp) + (long long)(ipy + ty) * dpitch) + 2 * (ipx + tx);
                    const int16_t* d1 = reinterpret_cast<const int16_t*>(reinterpret_cast<const uint8_t*>(d0) + dpitch);
                    const int iv = (r0[0] * w00 + r0[1] * w01 + r1[0] * w10 + r1[1] * w11 + (1 << (W_BITS - 5 - 1))) >> (W_BITS - 5);
                    const int gx = (d0[0] * w00 + d0[2] * w01 + d1[0] * w10 + d1[2] * w11 + (1 << (W_BITS - 1))) >> W_BITS;
                    const int gy = (d0[1] * w00 + d0[3] * w01 + d1[1] * w10 + d1[3] * w11 + (1 << (W_BITS - 1))) >> W_BITS;
                    const int k = ty * win_w + tx;
                    tI[k] = (int16_t)iv; tX[k] = (int16_t)gx; tY[k] = (int16_t)gy;
                    s11 += gx * gx; s12 += gx * gy; s22 += gy * gy;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        float A11, A12, A22;
        wave_sum3((float)s11, (float)s12, (float)s22, A11, A12, A22);
        A11 *= FLT_SCALE; A12 *= FLT_SCALE; A22 *= FLT_SCALE;
        float D = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * win_w * win_h);
        if (a.flags & DR3HIP_LK_GET_MIN_EIGENVALS) err = minEig;
        if ((double)minEig < a.min_eig || D < 1.1920928955078125e-7f) {
            if (level == 0) lost = DR3HIP_LK_LOST_MIN_EIG;
            if (a.iters && lane == 0) a.iters[pidx * (a.max_level + 1) + level] = 0;
            continue;
        }
        D = 1.f / D;
        cx -= half_x; cy -= half_y;
        float pdx = 0.f, pdy = 0.f;
        for (int j = 0; j < a.max_count; ++j) {
            const float fcx = floorf(cx), fcy = floorf(cy);
            const int inx = floor_to_int(fcx), iny = floor_to_int(fcy);
            if (!((unsigned)(inx + win_w) < (unsigned)(Lw + win_w) && (unsigned)(iny + win_h) < (unsigned)(Lh + win_h))) {
                if (level == 0) lost = DR3HIP_LK_LOST_ITERATE_OUTSIDE;
                break;
            }
            it_count = j + 1;
            const Weights W = make_weights(cx - fcx, cy - fcy);
            const int w00 = (int16_t)(W.w01 & 0xffffu), w01 = (int16_t)(W.w01 >> 16), w10 = (int16_t)(W.w23 & 0xffffu), w11 = (int16_t)(W.w23 >> 16);
            long long s1 = 0, s2 = 0;
            for (int q = lane; q < n_items; q += 64) {
                const int blk = q / win_w, tx = q - blk * win_w;
                const int y1 = min(8 * blk + 8, win_h);
                for (int ty = 8 * blk; ty < y1; ++ty) {
                    const uint8_t* r0 = Jp + (long long)(iny + ty) * pitch + (inx + tx);
                    const uint8_t* r1 = r0 + pitch;
                    const int k = ty * win_w + tx;
                    const int diff = ((r0[0] * w00 + r0[1] * w01 + r1[0] * w10 + r1[1] * w11 + (1 << (W_BITS - 5 - 1))) >> (W_BITS - 5)) - tI[k];
                    s1 += diff * tX[k]; s2 += diff * tY[k];
                }
            }
            float b1, b2;
            wave_sum2((float)s1, (float)s2, b1, b2);
            b1 *= FLT_SCALE; b2 *= FLT_SCALE;
            const float dx = (A12 * b2 - A22 * b1) * D, dy = (A12 * b1 - A11 * b2) * D;
            cx += dx; cy += dy;
            nx = cx + half_x; ny = cy + half_y;
            if ((double)dx * (double)dx + (double)dy * (double)dy <= a.epsilon) break;
            if (j > 0 && fabsf(dx + pdx) <= 0.01f && fabsf(dy + pdy) <= 0.01f) { nx -= dx * 0.5f; ny -= dy * 0.5f; break; }
            pdx = dx; pdy = dy;
        }
        if (a.iters && lane == 0) a.iters[pidx * (a.max_level + 1) + level] = it_count;

        if (level == 0 && !(a.flags & DR3HIP_LK_GET_MIN_EIGENVALS) && lost == 0) {
            const float fx = nx - half_x, fy = ny - half_y;
            const float ffx = floorf(fx), ffy = floorf(fy);
            const int inx = floor_to_int(ffx), iny = floor_to_int(ffy);
            if (!((unsigned)(inx + win_w) < (unsigned)(Lw + win_w) && (unsigned)(iny + win_h) < (unsigned)(Lh + win_h))) { lost = DR3HIP_LK_LOST_RESULT_OUTSIDE; continue; }
            const Weights W = make_weights(fx - ffx, fy - ffy);
            const int w00 = (int16_t)(W.w01 & 0xffffu), w01 = (int16_t)(W.w01 >> 16), w10 = (int16_t)(W.w23 & 0xffffu), w11 = (int16_t)(W.w23 >> 16);
            long long se = 0;
            for (int q = lane; q < n_items; q += 64) {
                const int blk = q / win_w, tx = q - blk * win_w;
                const int y1 = min(8 * blk + 8, win_h);
                for (int ty = 8 * blk; ty < y1; ++ty) {
                    const uint8_t* r0 = Jp + (long long)(iny + ty) * pitch + (inx + tx);
                    const uint8_t* r1 = r0 + pitch;
                    const int diff = ((r0[0] * w00 + r0[1] * w01 + r1[0] * w10 + r1[1] * w11 + (1 << (W_BITS - 5 - 1))) >> (W_BITS - 5)) - tI[ty * win_w + tx];
                    se += diff < 0 ? -diff : diff;
                }
            }
            err = wave_sum1((float)se) * 1.f / (float)(32 * win_w * win_h);
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) {
        a.next_xy[2 * pidx] = nx;
        a.next_xy[2 * pidx + 1] = ny;
        a.status[pidx] = (uint8_t)(lost == 0);
        if (a.reason) a.reason[pidx] = (uint8_t)lost;
        a.err[pidx] = err;
    }
}

#endif
}  // namespace

#if LKG_TU == 0
// ---- counters over the tracker's outputs (dr3hip_lk_stats): per-wave ballots / LDS histogram, one atomic per counter and block -----
namespace {
__global__ __launch_bounds__(256) void lk_stats_kernel(long long n, const uint8_t* __restrict__ status, const uint8_t* __restrict__ reason,
                                                       const int* __restrict__ iters, int levels, unsigned long long* __restrict__ cnt)
{
    constexpr int NC = 6 + 32;                  // c[j] = field j + 1 of dr3hip_lk_counters: tracked, 4 reasons, iters_total, 32 bins
    __shared__ unsigned int c[NC];
    for (int i = threadIdx.x; i < NC; i += blockDim.x) c[i] = 0;
    __syncthreads();
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (status[i]) atomicAdd(&c[0], 1u);
        const int r = reason ? reason[i] : 0;
        if (r >= 1 && r <= 4) atomicAdd(&c[r], 1u);
        for (int l = 0; l < levels; ++l) {
            const int k = iters[i * levels + l];
            atomicAdd(&c[5], (unsigned)(k < 0 ? 0 : k));
            atomicAdd(&c[6 + (k < 0 ? 0 : k < 31 ? k : 31)], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NC; i += blockDim.x)
        if (c[i]) atomicAdd(&cnt[i + 1], (unsigned long long)c[i]);
}
}  // namespace

hipError_t launch_lk_stats(long long n, const uint8_t* status, const uint8_t* reason, const int* iters, int levels,
                           unsigned long long* cnt, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    const long long blocks = (n + 255) / 256;
    hipLaunchKernelGGL(lk_stats_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, s, n, status, reason, iters, levels, cnt);
    return hipGetLastError();
}

// group shape by window (the CPU checker's summation-order rule is the same): points per wave ppw (64 / ppw lanes per point) and
// lane-items per lane ni
#endif

#if LKG_TU == 1
// the launches of the eight-points-per-wave forms (this translation unit is built with the max-ILP scheduler)
hipError_t launch_lk_track_p8(const LKArgs& a, bool win15, dim3 grid, dim3 block, size_t shmem, hipStream_t s, const LaunchEv* ev)
{
    if (win15) DR3_LAUNCH(ev, (lk_trackg_kernel<15, 8, 2>), grid, block, shmem, s, a);
    else DR3_LAUNCH(ev, (lk_trackg_kernel<0, 8, 2>), grid, block, shmem, s, a);
    return hipGetLastError();
}
#else
hipError_t launch_lk_track_p8(const LKArgs& a, bool win15, dim3 grid, dim3 block, size_t shmem, hipStream_t s, const LaunchEv* ev);

static inline void lk_group_shape(int win_w, int win_h, int& ppw, int& ni)
{
    const int n8 = win_w * ((win_h + 7) / 8);
    if (win_w <= 16 && win_h <= 16) { ppw = LKG_PPW16; ni = LKG_PPW16 == 8 ? 2 : 1; }
#ifndef LKG_NO_21_QUARTER
    else if (win_w == 21 && win_h == 21) { ppw = 4; ni = 4; }      // configs[0] / configs[4]: 8-row items on a quarter wave, four points per wave
#endif
    else if (n8 <= 32) { ppw = 2; ni = 1; }
    else if (n8 <= 64) { ppw = 2; ni = 2; }
    else { ppw = 1; ni = 2; }
}

hipError_t launch_lk_track(const LKArgs& a, int n_pairs, hipStream_t s, const LaunchEv* ev)
{
    if (n_pairs <= 0 || a.n_pts <= 0) return hipSuccess;
    if (a.win_w < 3 || a.win_h < 3 || a.win_w > DR3HIP_MAX_WIN || a.win_h > DR3HIP_MAX_WIN) return hipErrorInvalidValue;
    if (a.win_w > 32 || a.win_h > 32) {                 // the plain form: one wave per point, template in LDS
        const size_t taps = ((size_t)a.win_w * a.win_h + 7) & ~(size_t)7;
        DR3_LAUNCH(ev, lk_track_big_kernel, dim3(a.n_pts, n_pairs), dim3(64), 3 * taps * sizeof(int16_t), s, a);
        return hipGetLastError();
    }
    int ppw, ni;
    lk_group_shape(a.win_w, a.win_h, ppw, ni);
    const bool wide = LKG_WIDE != 0;
    const bool colf = ppw == 8 || (ppw == 4 && ni == 1);                     // as in the kernel
    const bool fixed = a.win_w == a.win_h && (a.win_w == 15 || a.win_w == 21 || a.win_w == 30);
    const int cols = colf ? 8 : (fixed && ppw == 4) ? (((a.win_w + 2 * RM + 4 + 3) / 4 + 1) & ~1) : 16, lanes = 64 / ppw, rpr = lanes / (cols / 2);
    const int sr = wide ? rpr : std::max(lanes / cols, 1), lpd = wide ? cols + 1 : ((a.win_w + 15) / 4) | 1;
    const int rhs = (((fixed && colf) ? a.win_h + 9 : colf ? 25 : a.win_h + 16) + sr - 1) / sr * sr;     // as in the kernel
    dim3 grid((a.n_pts + ppw - 1) / ppw, n_pairs), block(64);
    // the groups' J regions; wide staging: behind them the template's staged image rows (IROWS x IPITCH dwords per point, as in the kernel)
    const int irows_need = fixed ? (colf ? a.win_w + 1 : 8 * ((a.win_w + 7) / 8) + 1) : (colf ? 17 : 33);
    const int irows = (irows_need + rpr - 1) / rpr * rpr;
    const size_t shmem = ((size_t)ppw * rhs * lpd + (wide ? (size_t)ppw * irows * (cols + 1) : 0)) * sizeof(uint32_t);
    const int sq = a.win_w == a.win_h ? a.win_w : 0;
    if (ppw == 8) {
        return launch_lk_track_p8(a, sq == 15, grid, block, shmem, s, ev);           // the other translation unit (LKG_TU = 1)
    } else if (ppw == 4 && ni == 4) {
        DR3_LAUNCH(ev, (lk_trackg_kernel<21, 4, 4>), grid, block, shmem, s, a);
    } else if (ppw == 4) {
        if (sq == 15) DR3_LAUNCH(ev, (lk_trackg_kernel<15, 4, 1>), grid, block, shmem, s, a);
        else DR3_LAUNCH(ev, (lk_trackg_kernel<0, 4, 1>), grid, block, shmem, s, a);
    } else if (ni == 1) {
        DR3_LAUNCH(ev, (lk_trackg_kernel<0, 2, 1>), grid, block, shmem, s, a);
    } else if (ppw == 2) {
        if (sq == 21) DR3_LAUNCH(ev, (lk_trackg_kernel<21, 2, 2>), grid, block, shmem, s, a);
        else DR3_LAUNCH(ev, (lk_trackg_kernel<0, 2, 2>), grid, block, shmem, s, a);
    } else {
        if (sq == 30) DR3_LAUNCH(ev, (lk_trackg_kernel<30, 1, 2>), grid, block, shmem, s, a);
        else DR3_LAUNCH(ev, (lk_trackg_kernel<0, 1, 2>), grid, block, shmem, s, a);
    }
    return hipGetLastError();
}

#endif

}  // namespace dr3
