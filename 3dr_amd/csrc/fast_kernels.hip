// fast_kernels.hip -- FAST-10 corner scores, 3x3 non-maximum suppression and the grid-best Shi-Tomasi
// selection of feature_detection::FastDetector::detect (reference src/features.cpp:43-98,
// utils::shi_tomasi_score src/utils.cpp:282-321) for gfx950.  SURVEY 8f.2: this is what produces the
// points the LK tracker consumes.  The `fast` library the reference links (CMakeLists.txt:17) is not
// vendored; its published algorithm is restated here ("FAST"), parity unpinned.
//   fast_tile_kernel    : one workgroup per 64 x 40 tile of a level, all phases out of LDS (see the kernel).  The library finds
//                         the score by bisection over the corner test; the fixed point of that search has a closed form -- the
//                         largest b for which 10 contiguous ring pixels are all > p + b (or all < p - b) is
//                         max over the 16 arcs of min over the arc of (ring - p) (resp. p - ring), minus 1 --
//                         evaluated with a doubling sliding minimum (windows 2, 4, 8, 10).  A pixel is a
//                         corner at threshold t iff that score is >= t.  Then per corner: 3x3 non-max (a corner falls when
//                         an adjacent corner scores >= it), occupancy test, Shi-Tomasi score of the 8x8 box (integer
//                         sums are exact in the reference's f32 accumulators: 64 * 255^2 < 2^24), then a
//                         64-bit atomicMax per grid cell on (score bits, ~raster order): the reference keeps
//                         the first corner of the best score in level-major raster order.
#include "dr3hip_internal.h"
#include "vote.h"

namespace dr3 {
namespace {

// ring order of Rosten's make_offsets: (0,3) (1,3) (2,2) (3,1) (3,0) (3,-1) (2,-2) (1,-3) (0,-3) (-1,-3) ... (-1,3)
__device__ __forceinline__ int max_arc10(const int v[16])
{
    int m2[16], m4[16], m8[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) m2[i] = min(v[i], v[(i + 1) & 15]);
#pragma unroll
    for (int i = 0; i < 16; ++i) m4[i] = min(m2[i], m2[(i + 2) & 15]);
#pragma unroll
    for (int i = 0; i < 16; ++i) m8[i] = min(m4[i], m4[(i + 4) & 15]);
    int best = -1000;
#pragma unroll
    for (int i = 0; i < 16; ++i) best = max(best, min(m8[i], m2[(i + 8) & 15]));      // window of 10 starting at i
    return best;
}

// full ring score of one pixel (closed form of the library's bisection), c = centre value
__device__ __forceinline__ int ring_score(const uint8_t* __restrict__ p, int stride, int c)
{
    int d[16], e[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int dy = (i < 2 || i == 15) ? 3 : (i == 2 || i == 14) ? 2 : (i == 3 || i == 13) ? 1 : (i == 4 || i == 12) ? 0
                     : (i == 5 || i == 11) ? -1 : (i == 6 || i == 10) ? -2 : -3;
        const int dx = (i == 0 || i == 8) ? 0 : (i == 1 || i == 7) ? 1 : (i == 2 || i == 6) ? 2 : (i >= 3 && i <= 5) ? 3
                     : (i == 9 || i == 15) ? -1 : (i == 10 || i == 14) ? -2 : -3;
        d[i] = (int)p[dy * stride + dx] - c;
        e[i] = -d[i];
    }
    return max(max_arc10(d), max_arc10(e)) - 1;
}

// rank of this lane among the set bits of a wave ballot, and the ballot's population
__device__ __forceinline__ int ballot_rank(unsigned long long m)
{
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// utils::shi_tomasi_score, reference src/utils.cpp:282-321, from the integer sums of the 8 x 8 box (exact in the reference's f32
// accumulators: 64 * 255^2 < 2^24)
__device__ __forceinline__ float shi_tomasi_from_sums(int sxx, int syy, int sxy)
{
    const float dXX = (float)((double)(float)sxx / 128.0), dYY = (float)((double)(float)syy / 128.0), dXY = (float)((double)(float)sxy / 128.0);
    const float t = dXX + dYY;
    const float disc = t * t - 4.f * (dXX * dYY - dXY * dXY);
    return (float)(0.5 * ((double)t - sqrt((double)disc)));
}

// One workgroup decides a kFtW x kFtH tile of one pyramid level of one frame, everything out of LDS:
//  0. the image tile with a border of 5 rows / 8 columns (ring radius 3 + the neighbour ring of the non-maximum test + the
//     Shi-Tomasi box) is staged with one dword load per lane and round -- byte-aligned, the half-sampled levels have odd pitches;
//  1. Rosten's quick rejection on every pixel of the tile and of the one-pixel ring around it (whose scores the non-maximum test
//     of the tile's edge needs), a lane per 4 adjacent pixels: any 10 contiguous ring pixels contain two of the four compass
//     pixels (ring positions 0, 4, 8, 12), so a corner at threshold t needs two of them > c + t or two < c - t -- i.e. the
//     second largest compass value > c + t or the second smallest < c - t (a 4-input sorting network of v_min / v_max instead of
//     eight compares).  Survivors are appended to an LDS list (ballot + mbcnt, one LDS atomic per wave and round);
//  2. the 16-pixel ring score of the listed pixels on densely packed lanes, written to a byte score tile in LDS
//     (and, when the caller wants the score map, from there to memory);
//  3. per listed corner of the tile proper: 3 x 3 non-maximum test on the score tile (a corner falls when an adjacent corner scores
//     >= it), image-border test of the Shi-Tomasi box, occupancy; the few survivors go to a second list;
//  4. Shi-Tomasi score of the 8 x 8 box out of the image tile, then a 64-bit atomicMax per grid cell on (score bits, ~raster
//     order): the reference keeps the first corner of the best score in level-major raster order.
// No score map is written or read back on the detector's path, and the image is read once.
// Tile height kFtH: 40 (42 x 18 groups of 4 pixels = three full rounds of 256 lanes) or 32 (the same three rounds, a fifth of the lanes
// idle in the last one): the launcher takes 40 where that means fewer tile rows, 32 otherwise (a 94-row level is three tile rows either way,
// and three rows of 32 share the work more evenly than 40 + 40 + 14)
constexpr int kFtW = 64;                                   // pixels a workgroup decides: kFtW x kFtH
constexpr int kFiPitch = 80;                               // image tile: columns x0 - 8 .. x0 + 71, rows y0 - 5 .. y0 + kFtH + 4
constexpr int kFsPitch = 72;                               // score tile: columns x0 - 4 .. x0 + 67, rows y0 - 1 .. y0 + kFtH
constexpr int kFsGroups = kFsPitch / 4;                    // 4-pixel groups of a score row
typedef uint32_t u32x4_unaligned __attribute__((ext_vector_type(4), aligned(1)));
typedef uint32_t u32x4_lds __attribute__((ext_vector_type(4)));
typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 as_u16x2(uint32_t v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ uint32_t as_u32(u16x2 v) { return __builtin_bit_cast(uint32_t, v); }
// sum over the 8 lanes of an aligned group (DPP: xor 1, xor 2, then the half-row mirror, which is xor 4 once the quads are uniform)
__device__ __forceinline__ int sum8(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true);
    return v;
}
// LDS fetch-and-add with a per-lane addend, as the one instruction it is: the compiler's atomic optimizer would turn the divergent
// addend into a loop over the active lanes (some 50 instructions per lane), which is the wrong trade when most lanes take part
__device__ __forceinline__ uint32_t lds_add_rtn(uint32_t* counter, uint32_t v)
{
    uint32_t old;
    const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)counter;
    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(old) : "v"(addr), "v"(v) : "memory");
    return old;
}
// per 16-bit lane: non-zero iff the second largest of (a, b, c, d) > ctr + t or the second smallest < ctr - t (all values <= 255)
__device__ __forceinline__ u16x2 second_extremes_beyond(u16x2 a, u16x2 b, u16x2 c, u16x2 d, u16x2 ctr, u16x2 t)
{
    const u16x2 h1 = __builtin_elementwise_max(a, b), l1 = __builtin_elementwise_min(a, b);
    const u16x2 h2 = __builtin_elementwise_max(c, d), l2 = __builtin_elementwise_min(c, d);
    const u16x2 A = __builtin_elementwise_min(h1, h2), B = __builtin_elementwise_max(l1, l2);
    const u16x2 s2l = __builtin_elementwise_max(A, B), s2s = __builtin_elementwise_min(A, B);
    const u16x2 hi = ctr + t, lo = __builtin_elementwise_sub_sat(ctr, t);              // c - t < 0: nothing is darker
    return __builtin_elementwise_sub_sat(s2l, hi) | __builtin_elementwise_sub_sat(lo, s2s);
}

template <int kFtH>
__global__ __launch_bounds__(256) void fast_tile_kernel(const FastArgs a)
{
    constexpr int kFiRows = kFtH + 10, kFsRows = kFtH + 2;
    static_assert(kFiRows * (kFiPitch / 16) <= 256 && (kFsRows * kFsPitch) % 16 == 0, "one staging round; the score tile is cleared 16 bytes at a time");
    __shared__ uint32_t s_img[kFiRows * kFiPitch / 4];
    __shared__ uint32_t s_score[kFsRows * kFsPitch / 4];
    __shared__ uint16_t s_cand[kFsRows * kFsPitch];          // (score row << 7) | score column
    __shared__ uint16_t s_surv[kFtW * kFtH];
    __shared__ uint32_t s_n[2];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * kFtW, y0 = blockIdx.y * kFtH, z = blockIdx.z;
    const uint8_t* __restrict__ img = a.img + (long long)z * a.img_step;
    const int t = a.threshold;

    // ---- 0. clear the score tile, stage the image tile: 16 bytes per lane, one round (pixels outside the image are never used) ----
    for (int i = tid; i < kFsRows * kFsPitch / 16; i += 256) reinterpret_cast<uint4*>(s_score)[i] = make_uint4(0u, 0u, 0u, 0u);
    if (tid < 2) s_n[tid] = 0u;
    if (tid < kFiRows * (kFiPitch / 16)) {
        const int row = (tid * 13108) >> 16, j = tid - row * (kFiPitch / 16);         // tid / 5, tid % 5
        const int gy = y0 - 5 + row, gx = x0 - 8 + 16 * j;                            // gx < 0 only as -8 (first tile of a row, j = 0)
        const bool ok = (unsigned)gy < (unsigned)a.h && gx < a.w;                     // the 16 bytes may run past w: row padding / slack
        u32x4_unaligned v = *reinterpret_cast<const u32x4_unaligned*>(img + (ok ? gy * a.stride + max(gx, 0) : 0));
        if (gx < 0) v = u32x4_unaligned{0u, 0u, v.x, v.y};
        if (!ok) v = u32x4_unaligned{0u, 0u, 0u, 0u};
        reinterpret_cast<u32x4_lds*>(s_img)[tid] = v;
    }
    __syncthreads();

    // ---- 1. quick rejection, a lane per 4 pixels of the score tile, in packed 16-bit lanes: E holds pixels 0 and 2, O pixels 1, 3 ----
    constexpr int kRounds = (kFsRows * kFsGroups + 255) / 256;
    static_assert(kRounds <= 8, "the candidate mask of a lane has 2 x 16 bits");
    const u16x2 T2 = {(uint16_t)t, (uint16_t)t}, ONE2 = {1, 1};
    uint32_t mask = 0;                    // round r, pixel k of the lane's group: bit 2 r + (k & 1) + 16 (k >> 1)
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
        const int i = tid + 256 * r;
        const int row = (i * 3641) >> 16, j = i - row * kFsGroups;                     // i / 18, i % 18
        const int y = y0 - 1 + row;
        if (i < kFsRows * kFsGroups && y >= 3 && y < a.h - 3) {                        // fast_corner_detect_10 scans [3, size - 3)
            const uint32_t* p = s_img + (row + 4) * (kFiPitch / 4) + j + 1;
            const uint32_t up = p[-3 * (kFiPitch / 4)], dn = p[3 * (kFiPitch / 4)], lo = p[-1], mid = p[0], hi = p[1];
            const uint32_t M = 0x00ff00ffu;
            const u16x2 cE = as_u16x2(mid & M), cO = as_u16x2((mid >> 8) & M);
            const u16x2 n0E = as_u16x2(dn & M), n0O = as_u16x2((dn >> 8) & M);         // ring 0:  (0, +3)
            const u16x2 n8E = as_u16x2(up & M), n8O = as_u16x2((up >> 8) & M);         // ring 8:  (0, -3)
            // ring 4 (+3, 0): bytes 3 .. 6 of (mid, hi);  ring 12 (-3, 0): bytes 1 .. 4 of (lo, mid)
            const u16x2 n4E = as_u16x2(__builtin_amdgcn_perm(hi, mid, 0x0c050c03u)), n4O = as_u16x2(hi & M);
            const u16x2 n12E = as_u16x2((lo >> 8) & M), n12O = as_u16x2(__builtin_amdgcn_perm(mid, lo, 0x0c040c02u));
            // a corner needs two compass pixels > c + t or two < c - t: the second largest above c + t or the second smallest below c - t
            const u16x2 rE = __builtin_elementwise_min(second_extremes_beyond(n0E, n4E, n8E, n12E, cE, T2), ONE2);
            const u16x2 rO = __builtin_elementwise_min(second_extremes_beyond(n0O, n4O, n8O, n12O, cO, T2), ONE2);
            mask |= (as_u32(rE) | (as_u32(rO) << 1)) << (2 * r);
        }
    }
    // the lane's candidates of all rounds go to the list behind one LDS fetch-and-add; columns outside [3, w - 3) are dropped here
    if (mask) {
        uint32_t keep = mask;
        if (x0 < 8 || x0 + kFsPitch > a.w - 3) {                                       // only the first and the last tiles of a row hold such columns
            keep = 0;
            for (uint32_t m = mask; m; m &= m - 1) {
                const int b = __builtin_ctz(m), r = (b & 15) >> 1, k = (b & 1) | ((b >> 4) << 1);
                const int i = tid + 256 * r, row = (i * 3641) >> 16, j = i - row * kFsGroups;
                const int x = x0 - 4 + 4 * j + k;
                if (x >= 3 && x < a.w - 3) keep |= 1u << b;
            }
        }
        if (keep) {
            uint32_t pos = lds_add_rtn(&s_n[0], (uint32_t)__popc(keep));
            for (uint32_t m = keep; m; m &= m - 1) {
                const int b = __builtin_ctz(m), r = (b & 15) >> 1, k = (b & 1) | ((b >> 4) << 1);
                const int i = tid + 256 * r, row = (i * 3641) >> 16, j = i - row * kFsGroups;
                s_cand[pos++] = (uint16_t)((row << 7) | (4 * j + k));
            }
        }
    }
    __syncthreads();

    // ---- 2. ring scores of the listed pixels ----
    const int ncand = (int)s_n[0];
    uint8_t* sc8 = reinterpret_cast<uint8_t*>(s_score);
    const uint8_t* im8 = reinterpret_cast<const uint8_t*>(s_img);
    for (int i = tid; i < ncand; i += 256) {
        const int e = s_cand[i], sr = e >> 7, sx = e & 127;
        const uint8_t* p = im8 + (sr + 4) * kFiPitch + sx + 4;
        const int score = ring_score(p, kFiPitch, p[0]);
        if (score >= t) sc8[sr * kFsPitch + sx] = (uint8_t)score;                      // <= 254
    }
    __syncthreads();
    if (a.score) {                                                                     // the caller wants the map: rows of 64 bytes, 0 = no corner
        uint8_t* sm = a.score + (long long)z * a.score_step;
        for (int i = tid; i < kFtH * (kFtW / 4); i += 256) {
            const int r = i >> 4, jj = i & 15, y = y0 + r;
            if (y < a.h) *reinterpret_cast<uint32_t*>(sm + (long long)y * a.spitch + x0 + 4 * jj) = s_score[(r + 1) * kFsGroups + 1 + jj];
        }                                                                              // spitch is a multiple of 64 >= w: pad bytes become 0
    }
    if (!a.cells || ncand == 0) return;

    // ---- 3. non-maximum test, image border of the Shi-Tomasi box, occupancy ----
    for (int i = tid; i < ncand; i += 256) {
        const int e = s_cand[i], sr = e >> 7, sx = e & 127;
        if (sr < 1 || sr > kFtH || sx < 4 || sx >= 4 + kFtW) continue;                 // the ring around the tile belongs to the neighbours
        const uint8_t* q = sc8 + sr * kFsPitch + sx;
        const int s = q[0];
        if (s == 0) continue;
        // fast_nonmax_3x3: a neighbour that is a corner (score > 0) and scores >= s removes this one; pixels outside the image hold 0
        const int nmax = max(max(max((int)q[-kFsPitch - 1], (int)q[-kFsPitch]), max((int)q[-kFsPitch + 1], (int)q[-1])),
                             max(max((int)q[1], (int)q[kFsPitch - 1]), max((int)q[kFsPitch], (int)q[kFsPitch + 1])));
        if (nmax >= s) continue;
        const int x = x0 - 4 + sx, y = y0 - 1 + sr;
        // utils::shi_tomasi_score returns 0 when the box leaves the image: never above the cell's initial Corner (threshold >= 0)
        if (x - 4 < 1 || x + 4 >= a.w - 1 || y - 4 < 1 || y + 4 >= a.h - 1) continue;
        const int scale = 1 << a.level;
        const int k = ((y * scale) / a.cell_size) * a.grid_cols + (x * scale) / a.cell_size;
        if (a.occupancy && a.occupancy[(long long)z * a.n_cells + k]) continue;
        s_surv[atomicAdd(&s_n[1], 1u)] = (uint16_t)e;
    }
    __syncthreads();

    // ---- 4. Shi-Tomasi of the survivors, eight lanes per corner (a row of the 8 x 8 box each), per-cell winner ----
    const int nsurv = (int)s_n[1];
    const int sub = tid & 7;
    for (int i0 = 0; i0 < nsurv; i0 += 32) {                                           // wave-uniform trip count
        const int i = i0 + (tid >> 3);
        const int e = i < nsurv ? s_surv[i] : ((1 << 7) | 4), sr = e >> 7, sx = e & 127;   // idle lanes read a valid address
        const uint8_t* b = im8 + (sr + 4 - 5 + sub) * kFiPitch + (sx + 4 - 5);         // pixel (y - 5 + sub, x - 5); the lane's box row is the next one
        int sxx = 0, syy = 0, sxy = 0;                                                 // exact: each fits 23 bits, so the order of the sums is free
#pragma unroll
        for (int xx = 0; xx < 8; ++xx) {
            const int dx = (int)b[kFiPitch + xx + 2] - (int)b[kFiPitch + xx];
            const int dy = (int)b[2 * kFiPitch + xx + 1] - (int)b[xx + 1];
            sxx += dx * dx; syy += dy * dy; sxy += dx * dy;
        }
        sxx = sum8(sxx); syy = sum8(syy); sxy = sum8(sxy);
        if (sub != 0 || i >= nsurv) continue;
        const float score = shi_tomasi_from_sums(sxx, syy, sxy);
        if (!(score > a.det_threshold)) continue;                                      // can never beat the cell's initial Corner
        const int x = x0 - 4 + sx, y = y0 - 1 + sr;
        const int scale = 1 << a.level;
        const int k = ((y * scale) / a.cell_size) * a.grid_cols + (x * scale) / a.cell_size;
        const uint32_t order = ((uint32_t)a.level << 28) | ((uint32_t)y << 14) | (uint32_t)x;
        const unsigned long long key = ((unsigned long long)__float_as_uint(score) << 32) | (uint32_t)~order;
        atomicMax(a.cells + (long long)z * a.n_cells + k, key);
    }
}

// per-cell keys -> Corner(x * scale, y * scale, score, level) (src/features.cpp:78-93) and the tracker's input points
__global__ __launch_bounds__(256) void cells_decode_kernel(const CellsArgs a)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n_total) return;
    const unsigned long long key = a.cells[i];
    const uint32_t order = ~(uint32_t)key;
    const float score = __uint_as_float((uint32_t)(key >> 32));
    const bool valid = key != a.empty_key && (double)score > (double)a.det_threshold;
    const int level = (int)(order >> 28);
    const int x = (int)(order & 0x3fffu) << level, y = (int)((order >> 14) & 0x3fffu) << level;
    if (a.corners) {
        dr3hip_corner c;
        c.x = valid ? x : 0; c.y = valid ? y : 0; c.level = valid ? level : -1; c.score = valid ? score : a.det_threshold;
        a.corners[i] = c;
    }
    if (a.xy) {
        a.xy[2 * i]     = valid ? (float)x : DR3HIP_NO_POINT;
        a.xy[2 * i + 1] = valid ? (float)y : DR3HIP_NO_POINT;
    }
}

// InitHelper::CheckFundamental (reference src/initialization.cpp:171-249) for all RANSAC hypotheses of FindFundamental (:80-127) at
// once -- SURVEY 8f.4.  Kept on the device for residency: tracked points never leave it between LK and the inlier vote.
// The arithmetic of a (match, hypothesis) pair depends on nothing else; only the accumulation of the score is ordered.  Phase 1
// (vote_chi_kernel) computes both chi-square values of every pair in parallel -- a wave owns 64 hypotheses x 64 matches, lane =
// hypothesis, the matches of the tile sit one per lane and are broadcast with v_readlane -- and writes them to scratch as
// chi[match][hypothesis] (512 contiguous bytes per wave-store) plus the inlier bytes (64 contiguous bytes per lane).  Phase 2
// (vote_sum_kernel), a lane per hypothesis, adds thScore - chi in the reference's order, 64 matches per memory round trip.  An
// erased match is written as chi = +inf: it fails both tests, which is exactly "contributes nothing, inlier 0".
typedef uint32_t u32_any __attribute__((aligned(1)));
__global__ __launch_bounds__(64) void vote_chi_kernel(const CheckFArgs a, float2* __restrict__ chi, int hp, int c0, int cn)
{
    const int lane = threadIdx.x;
    const int hyp_raw = blockIdx.x * 64 + lane;
    const int hyp = hyp_raw < a.n_hyp ? hyp_raw : a.n_hyp - 1;           // spare lanes shadow the last hypothesis; their columns are padding
    const float* F = a.F + 9 * hyp;
    const float f11 = F[0], f12 = F[1], f13 = F[2], f21 = F[3], f22 = F[4], f23 = F[5], f31 = F[6], f32 = F[7], f33 = F[8];
    const float th = 3.841f, inv = a.inv_sigma_square;
    const int m0 = blockIdx.y * 64;                                       // first match of the tile, relative to the round
    const int mi = m0 + lane < cn ? m0 + lane : cn - 1;
    const float2 p1 = reinterpret_cast<const float2*>(a.pts1)[c0 + mi], p2 = reinterpret_cast<const float2*>(a.pts2)[c0 + mi];
    const int kept = (m0 + lane < cn) && (!a.mask || a.mask[c0 + mi]);
    unsigned long long bits = 0ull;
#pragma unroll
    for (int j = 0; j < 64; ++j) {
        // the tile's match j, wave-uniform
        const float u1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p1.x), j));
        const float v1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p1.y), j));
        const float u2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p2.x), j));
        const float v2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p2.y), j));
        const int keep = __builtin_amdgcn_readlane(kept, j);
        const float a2 = f11 * u1 + f12 * v1 + f13, b2 = f21 * u1 + f22 * v1 + f23, c2 = f31 * u1 + f32 * v1 + f33;
        const float num2 = a2 * u2 + b2 * v2 + c2;
        float chi1 = num2 * num2 / (a2 * a2 + b2 * b2) * inv;
        const float a1 = f11 * u2 + f21 * v2 + f31, b1 = f12 * u2 + f22 * v2 + f32, c1 = f13 * u2 + f23 * v2 + f33;
        const float num1 = a1 * u1 + b1 * v1 + c1;
        float chi2 = num1 * num1 / (a1 * a1 + b1 * b1) * inv;
        if (!keep) { chi1 = __builtin_inff(); chi2 = __builtin_inff(); }
        if (m0 + j < cn) chi[(long long)(m0 + j) * hp + hyp_raw] = make_float2(chi1, chi2);
        if (!(chi1 > th) && !(chi2 > th)) bits |= 1ull << j;
    }
    if (a.inliers && hyp_raw < a.n_hyp) {
        uint8_t* row = a.inliers + (long long)hyp * a.n + c0 + m0;
        if (m0 + 64 <= cn) {
#pragma unroll
            for (int d = 0; d < 16; ++d) {
                const uint32_t nib = (uint32_t)(bits >> (4 * d)) & 0xfu;                   // four matches -> four bytes of 0 / 1
                *reinterpret_cast<u32_any*>(row + 4 * d) = (nib * 0x00204081u) & 0x01010101u;
            }
        } else {
            for (int j = 0; m0 + j < cn; ++j) row[j] = (uint8_t)((bits >> j) & 1ull);
        }
    }
}

__global__ __launch_bounds__(64) void vote_sum_kernel(const CheckFArgs a, const float2* __restrict__ chi, int hp, int cn, int first_round)
{
    const int hyp = blockIdx.x * 64 + threadIdx.x;                        // padded columns hold defined values: every lane runs
    const float th = 3.841f, thScore = 5.991f;
    float score = (first_round || hyp >= a.n_hyp) ? 0.f : a.scores[hyp];
    for (int m0 = 0; m0 < cn; m0 += 64) {
        float2 c[64];
#pragma unroll
        for (int j = 0; j < 64; ++j)
            c[j] = m0 + j < cn ? chi[(long long)(m0 + j) * hp + hyp] : make_float2(__builtin_inff(), __builtin_inff());
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            if (!(c[j].x > th)) score += thScore - c[j].x;
            if (!(c[j].y > th)) score += thScore - c[j].y;
        }
    }
    if (hyp < a.n_hyp) a.scores[hyp] = score;
}

// InitHelper::Normalize (reference src/initialization.cpp:365-410) for a batch of point sets, one workgroup per set.  The means
// and the mean absolute deviations are sequential f32 sums in point order -- the one thing that cannot be parallelised without
// changing the bits -- so two lanes (x and y) walk the points twice; the shift and scale of the points, which is the bulk of
// the bytes, runs on all 256 lanes.  With a mask (the tracker's status bytes) lost points are skipped exactly as if they had
// been erased from the vectors first (:621-635), which is what keeps LK's output on the device up to the inlier vote.
constexpr int NP_CHUNK = 2048;                     // points staged through LDS per round (coalesced loads by all 256 lanes)
__global__ __launch_bounds__(256) void normalize_points_kernel(const NormArgs a)
{
    __shared__ float sh[4];                         // meanX, meanY, sX, sY
    __shared__ float2 pt[NP_CHUNK];
    __shared__ uint8_t keep[NP_CHUNK];
    const int t = threadIdx.x;
    const float* __restrict__ xy = a.xy + (long long)blockIdx.x * a.xy_step;
    const uint8_t* __restrict__ mask = a.mask ? a.mask + (long long)blockIdx.x * a.mask_step : nullptr;
    float* __restrict__ out = a.out_xy + (long long)blockIdx.x * a.xy_step;
    // two sequential passes (the sum, then the absolute deviations from the mean) by lane 0 (x) and lane 1 (y); the points of a
    // round come out of LDS, where all lanes have put them, instead of one global load per element behind the other
    float acc = 0.f, mean = 0.f;
    int N = 0;
    for (int pass = 0; pass < 2; ++pass) {
        acc = 0.f;
        for (int c0 = 0; c0 < a.n; c0 += NP_CHUNK) {
            const int cn = a.n - c0 < NP_CHUNK ? a.n - c0 : NP_CHUNK;
            __syncthreads();
            for (int i = t; i < cn; i += 256) {
                pt[i] = reinterpret_cast<const float2*>(xy)[c0 + i];
                keep[i] = mask ? mask[c0 + i] : (uint8_t)1;
            }
            __syncthreads();
            if (t < 2) {
                // masked points add +0.0f (acc is never -0.0f: it starts at +0.0f), so the walk needs no branch and the LDS reads of
                // eight elements go out ahead of the eight dependent adds
                const float* v = reinterpret_cast<const float*>(pt) + t;
                int i = 0;
                for (; i + 8 <= cn; i += 8) {
                    float x[8]; uint8_t k[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) { x[j] = v[2 * (i + j)]; k[j] = keep[i + j]; }
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float term = pass == 0 ? x[j] : fabsf(x[j] - mean);
                        acc += k[j] ? term : 0.f;
                        if (pass == 0) N += k[j] ? 1 : 0;
                    }
                }
                for (; i < cn; ++i) {
                    const float term = pass == 0 ? v[2 * i] : fabsf(v[2 * i] - mean);
                    acc += keep[i] ? term : 0.f;
                    if (pass == 0) N += keep[i] ? 1 : 0;
                }
            }
        }
        if (pass == 0) mean = acc / (float)N;
    }
    if (t < 2) {
        const float dev = acc / (float)N;
        sh[t] = mean;
        sh[2 + t] = (float)(1.0 / (double)dev);
    }
    __syncthreads();
    const float meanX = sh[0], meanY = sh[1], sX = sh[2], sY = sh[3];
    for (int i = t; i < a.n; i += 256) {
        const bool ok = !mask || mask[i];
        out[2 * i] = ok ? (xy[2 * i] - meanX) * sX : 0.f;
        out[2 * i + 1] = ok ? (xy[2 * i + 1] - meanY) * sY : 0.f;
    }
    if (t == 0) {
        float* T = a.T + 9 * (long long)blockIdx.x;
        T[0] = sX; T[1] = 0.f; T[2] = -meanX * sX;
        T[3] = 0.f; T[4] = sY; T[5] = -meanY * sY;
        T[6] = 0.f; T[7] = 0.f; T[8] = 1.f;
    }
}

}  // namespace

hipError_t launch_normalize_points(const NormArgs& a, hipStream_t s)
{
    if (a.n_sets <= 0) return hipSuccess;
    hipLaunchKernelGGL(normalize_points_kernel, dim3(a.n_sets), dim3(256), 0, s, a);
    return hipGetLastError();
}

size_t vote_scratch_floats(int n_hyp, int n)
{
    const size_t hp = ((size_t)n_hyp + 63) / 64 * 64;
    return 2 * hp * (size_t)(n < VOTE_CHUNK ? (n > 0 ? n : 1) : VOTE_CHUNK);
}

hipError_t launch_check_fundamental_2p(const CheckFArgs& a, float* scratch, hipStream_t s)
{
    if (a.n_hyp <= 0) return hipSuccess;
    const int hb = (a.n_hyp + 63) / 64, hp = hb * 64;
    float2* chi = reinterpret_cast<float2*>(scratch);
    if (a.n <= 0) { hipLaunchKernelGGL(vote_sum_kernel, dim3(hb), dim3(64), 0, s, a, (const float2*)chi, hp, 0, 1); return hipGetLastError(); }
    for (int c0 = 0; c0 < a.n; c0 += VOTE_CHUNK) {
        const int cn = a.n - c0 < VOTE_CHUNK ? a.n - c0 : VOTE_CHUNK;
        hipLaunchKernelGGL(vote_chi_kernel, dim3(hb, (cn + 63) / 64), dim3(64), 0, s, a, chi, hp, c0, cn);
        hipLaunchKernelGGL(vote_sum_kernel, dim3(hb), dim3(64), 0, s, a, (const float2*)chi, hp, cn, c0 == 0 ? 1 : 0);
    }
    return hipGetLastError();
}

hipError_t launch_cells_decode(const CellsArgs& a, hipStream_t s)
{
    if (a.n_total <= 0) return hipSuccess;
    hipLaunchKernelGGL(cells_decode_kernel, dim3((a.n_total + 255) / 256), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_fast_level(const FastArgs& a, hipStream_t s)
{
    if (a.w <= 0 || a.h <= 0) return hipSuccess;
    const int th = (a.h + 39) / 40 < (a.h + 31) / 32 ? 40 : 32;
    dim3 grid((a.w + kFtW - 1) / kFtW, (a.h + th - 1) / th, a.n_slots > 0 ? a.n_slots : 1);
    if (th == 40) hipLaunchKernelGGL(fast_tile_kernel<40>, grid, dim3(256), 0, s, a);
    else          hipLaunchKernelGGL(fast_tile_kernel<32>, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace dr3
