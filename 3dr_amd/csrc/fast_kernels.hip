// fast_kernels.hip -- FAST-10 corner scores, 3x3 non-maximum suppression and the grid-best Shi-Tomasi
// selection of feature_detection::FastDetector::detect (reference src/features.cpp:43-98,
// utils::shi_tomasi_score src/utils.cpp:282-321) for gfx950.  SURVEY 8f.2: this is what produces the
// points the LK tracker consumes.  The `fast` library the reference links (CMakeLists.txt:17) is not
// vendored; its published algorithm is restated here ("FAST"), parity unpinned.
//   fast10_score_kernel : one lane per 4 pixels (compass quick-rejection from dword loads).  The library finds the score by bisection over the corner
//                         test; the fixed point of that search has a closed form -- the largest b for which
//                         10 contiguous ring pixels are all > p + b (or all < p - b) is
//                         max over the 16 arcs of min over the arc of (ring - p) (resp. p - ring), minus 1 --
//                         evaluated with a doubling sliding minimum (windows 2, 4, 8, 10).  A pixel is a
//                         corner at threshold t iff that score is >= t.
//   fast_select_kernel  : one lane per 8 bytes of the score map; per corner: 3x3 non-max (a corner falls when an adjacent
//                         corner scores >= it), occupancy test, Shi-Tomasi score of the 8x8 box (integer
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

// A wave owns 256 pixels of one row, a lane 4 adjacent ones (x % 4 == 0).
//  1. Rosten's quick rejection: any 10 contiguous ring pixels contain at least two of the four compass pixels (ring
//     positions 0, 4, 8, 12), so a corner at threshold t needs two of them > c + t or two < c - t.  The compass pixels
//     of the lane's 4 pixels come from five aligned dword loads (rows y - 3, y + 3: one dword; row y: bytes x-4 .. x+7).
//  2. The few survivors are compacted through LDS (ballot + mbcnt) so that the 16-pixel ring score runs on densely
//     packed lanes -- once per 64 candidates instead of once per pixel slot with any candidate in the wave.
//  3. Each lane collects its 4 result bytes from LDS and stores one dword of the score map.
__global__ __launch_bounds__(256) void fast10_score_kernel(const FastArgs a)
{
    __shared__ uint16_t s_cand[4][256];
    __shared__ uint32_t s_res[4][64];
    const int lane = threadIdx.x, wv = threadIdx.y;
    const int xw = blockIdx.x * 256;
    const int x = xw + lane * 4;
    const int y = blockIdx.y * 4 + wv;
    if (xw >= a.w || y >= a.h) return;                                   // wave-uniform
    const uint8_t* __restrict__ img = a.img + (long long)blockIdx.z * a.img_step;
    const uint8_t* row = img + (long long)y * a.stride;
    const int t = a.threshold;
    uint32_t cand = 0;                                                    // bit k: pixel x + k passed the quick test
    if (y >= 3 && y < a.h - 3 && x < a.w) {                              // fast_corner_detect_10 scans [3, size - 3)
        const bool wide = x >= 4 && x + 8 <= a.stride && (((uintptr_t)img | (uintptr_t)a.stride) & 3) == 0;   // dword loads stay in the row
        uint32_t up = 0, dn = 0, lo = 0, mid = 0, hi = 0;
        if (wide) {
            up = *reinterpret_cast<const uint32_t*>(row - 3 * a.stride + x);
            dn = *reinterpret_cast<const uint32_t*>(row + 3 * a.stride + x);
            lo = *reinterpret_cast<const uint32_t*>(row + x - 4);
            mid = *reinterpret_cast<const uint32_t*>(row + x);
            hi = *reinterpret_cast<const uint32_t*>(row + x + 4);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int xx = x + k;
            if (xx < 3 || xx >= a.w - 3) continue;
            int c, n0, n4, n8, n12;
            if (wide) {
                const uint64_t rowbits = ((uint64_t)hi << 32) | mid;    // bytes x .. x + 7
                const uint64_t lobits = ((uint64_t)mid << 32) | lo;     // bytes x - 4 .. x + 3
                c = (mid >> (8 * k)) & 0xff;
                n0 = (int)((dn >> (8 * k)) & 0xff) - c;                 // ring 0:  (0, +3)
                n8 = (int)((up >> (8 * k)) & 0xff) - c;                 // ring 8:  (0, -3)
                n4 = (int)((rowbits >> (8 * (k + 3))) & 0xff) - c;      // ring 4:  (+3, 0)
                n12 = (int)((lobits >> (8 * (k + 1))) & 0xff) - c;      // ring 12: (-3, 0)
            } else {
                const uint8_t* p = row + xx;
                c = p[0]; n0 = (int)p[3 * a.stride] - c; n4 = (int)p[3] - c; n8 = (int)p[-3 * a.stride] - c; n12 = (int)p[-3] - c;
            }
            const int nb = (n0 > t) + (n4 > t) + (n8 > t) + (n12 > t), nd = (n0 < -t) + (n4 < -t) + (n8 < -t) + (n12 < -t);
            if (nb >= 2 || nd >= 2) cand |= 1u << k;
        }
    }
    // compaction: candidate (lane, k) -> dense slot
    s_res[wv][lane] = 0u;
    int total = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const bool c = (cand >> k) & 1u;
        const unsigned long long m = __ballot(c);
        if (c) s_cand[wv][total + ballot_rank(m)] = (uint16_t)(4 * lane + k);
        total += __popcll(m);
    }
    __builtin_amdgcn_wave_barrier();
    uint8_t* res8 = reinterpret_cast<uint8_t*>(s_res[wv]);
    for (int i = lane; i < total; i += 64) {
        const int idx = s_cand[wv][i];
        const uint8_t* p = row + xw + idx;
        const int score = ring_score(p, a.stride, p[0]);
        if (score >= t) res8[idx] = (uint8_t)score;                     // <= 254
    }
    __builtin_amdgcn_wave_barrier();
    if (x < a.w) {
        uint8_t* orow = a.score + (long long)blockIdx.z * a.score_step + (long long)y * a.spitch + x;
        *reinterpret_cast<uint32_t*>(orow) = s_res[wv][lane];           // spitch is a multiple of 64 >= w: pad bytes become 0
    }
}

// utils::shi_tomasi_score, reference src/utils.cpp:282-321
__device__ float shi_tomasi(const uint8_t* __restrict__ img, int w, int h, int stride, int u, int v)
{
    const int x_min = u - 4, x_max = u + 4, y_min = v - 4, y_max = v + 4;
    if (x_min < 1 || x_max >= w - 1 || y_min < 1 || y_max >= h - 1) return 0.f;
    int sxx = 0, syy = 0, sxy = 0;                                    // exact: each fits 23 bits
    for (int y = y_min; y < y_max; ++y) {
        const uint8_t* r = img + (long long)y * stride + x_min;
        for (int x = 0; x < 8; ++x) {
            const int dx = (int)r[x + 1] - (int)r[x - 1];
            const int dy = (int)r[x + stride] - (int)r[x - stride];
            sxx += dx * dx; syy += dy * dy; sxy += dx * dy;
        }
    }
    const float dXX = (float)((double)(float)sxx / 128.0), dYY = (float)((double)(float)syy / 128.0), dXY = (float)((double)(float)sxy / 128.0);
    const float t = dXX + dYY;
    const float disc = t * t - 4.f * (dXX * dYY - dXY * dXY);
    return (float)(0.5 * ((double)t - sqrt((double)disc)));
}

// one surviving corner of the score map: non-max test, occupancy, Shi-Tomasi, per-cell atomicMax
__device__ void select_pixel(const FastArgs& a, const uint8_t* __restrict__ sp, int x, int y, int s, int z)
{
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            if ((dx | dy) == 0) continue;
            const int xx = x + dx, yy = y + dy;
            if (xx < 0 || xx >= a.w || yy < 0 || yy >= a.h) continue;
            const int t = sp[dy * a.spitch + dx];
            if (t > 0 && t >= s) return;                              // fast_nonmax_3x3
        }
    const int scale = 1 << a.level;
    const int k = ((y * scale) / a.cell_size) * a.grid_cols + (x * scale) / a.cell_size;
    if (a.occupancy && a.occupancy[(long long)z * a.n_cells + k]) return;
    const float score = shi_tomasi(a.img + (long long)z * a.img_step, a.w, a.h, a.stride, x, y);
    if (!(score > a.det_threshold)) return;                           // can never beat the cell's initial Corner
    const uint32_t order = ((uint32_t)a.level << 28) | ((uint32_t)y << 14) | (uint32_t)x;
    const unsigned long long key = ((unsigned long long)__float_as_uint(score) << 32) | (uint32_t)~order;
    atomicMax(a.cells + (long long)z * a.n_cells + k, key);
}

// scan of the score map: a wave owns 512 map bytes of one row, a lane 8 of them (rows are 64-byte pitched); the corners
// found are compacted through LDS so that the non-max test and the 8x8 Shi-Tomasi box run on densely packed lanes
__global__ __launch_bounds__(256) void fast_select_kernel(const FastArgs a)
{
    __shared__ uint16_t s_list[4][512];
    const int lane = threadIdx.x, wv = threadIdx.y;
    const int xw = blockIdx.x * 512;
    const int x = xw + lane * 8;
    const int y = blockIdx.y * 4 + wv;
    if (xw >= a.w || y >= a.h) return;                                   // wave-uniform
    const uint8_t* srow = a.score + (long long)blockIdx.z * a.score_step + (long long)y * a.spitch;
    uint2 v = make_uint2(0u, 0u);
    if (x < a.w) v = *reinterpret_cast<const uint2*>(srow + x);
    if (__ballot((v.x | v.y) != 0) == 0ull) return;                     // nothing in these 512 bytes
    int total = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const bool c = (((k < 4 ? v.x : v.y) >> (8 * (k & 3))) & 0xff) != 0 && x + k < a.w;
        const unsigned long long m = __ballot(c);
        if (c) s_list[wv][total + ballot_rank(m)] = (uint16_t)(8 * lane + k);
        total += __popcll(m);
    }
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < total; i += 64) {
        const int xx = xw + s_list[wv][i];
        select_pixel(a, srow + xx, xx, y, srow[xx], blockIdx.z);
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

hipError_t launch_fast_score(const FastArgs& a, hipStream_t s)
{
    if (a.w <= 0 || a.h <= 0) return hipSuccess;
    dim3 block(64, 4), grid((a.w + 255) / 256, (a.h + 3) / 4, a.n_slots > 0 ? a.n_slots : 1);
    hipLaunchKernelGGL(fast10_score_kernel, grid, block, 0, s, a);
    return hipGetLastError();
}

hipError_t launch_fast_select(const FastArgs& a, hipStream_t s)
{
    if (a.w <= 0 || a.h <= 0) return hipSuccess;
    dim3 block(64, 4), grid((a.w + 511) / 512, (a.h + 3) / 4, a.n_slots > 0 ? a.n_slots : 1);
    hipLaunchKernelGGL(fast_select_kernel, grid, block, 0, s, a);
    return hipGetLastError();
}

}  // namespace dr3
