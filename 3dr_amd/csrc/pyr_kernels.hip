// pyr_kernels.hip -- fused Gaussian pyrDown + Scharr derivative pass for one pyramid level of a
// frame batch (gfx950).  Replaces what cv::calcOpticalFlowPyrLK builds internally from the two
// images the reference passes at src/initialization.cpp:608-609: buildOpticalFlowPyramid/pyrDown
//  and calcSharrDeriv (A-2).  Integer-exact.
//
// One 256-thread workgroup owns a TW x TH tile of level l of one frame.  The tile plus its halo
// (2 rows above/below, 4 aligned bytes left/right) is loaded ONCE from HBM with coalesced dword row
// loads into LDS; REFLECT_101 is applied as a row-index reflection at load time and as a column
// patch in LDS.  From that tile the workgroup emits
//   - the level-l derivative tile: interleaved (Ix,Iy) int16, one 16-byte store per lane
//     (4 pixels), i.e. 1 KiB contiguous per wave-instruction;
//   - the level-(l+1) image tile (TW/2 x TH/2), one dword store per lane.
// HBM traffic per level-l pixel: 1 B read (+ halo re-reads served by L2), 4 B + 0.25 B written.
#include "dr3hip_internal.h"

namespace dr3 {
namespace {

#ifndef PYR_TW
#define PYR_TW 128
#define PYR_TH 32
#endif
constexpr int TW = PYR_TW;         // tile width  (level-l pixels), multiple of 8
constexpr int TH = PYR_TH;         // tile height, even; TW * TH = 4096 (the thread maps below)
constexpr int LW = TW + 8;         // LDS bytes used per row: x0-4 .. x0+TW+3
constexpr int LP = TW + 16;        // LDS row pitch in bytes (multiple of 8)
static_assert(TW * TH == 4096 && TW % 8 == 0 && TH % 2 == 0, "tile shape");
constexpr int LR = TH + 4;         // LDS rows: y0-2 .. y0+TH+1
constexpr int NT = 256;

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 as_u16x2(uint32_t v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ uint32_t as_u32(u16x2 v) { return __builtin_bit_cast(uint32_t, v); }

__device__ __forceinline__ int reflect101(int p, int n)
{
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

__device__ __forceinline__ uint32_t border_dword(const uint8_t* __restrict__ srow, int xq, int w)
{
    if (xq >= 0 && xq + 3 < w) return *reinterpret_cast<const uint32_t*>(srow + xq);
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) v |= (uint32_t)srow[reflect101(xq + k, w)] << (8 * k);
    return v;
}

// one wave-task of the border of level lv of frame slot (see pad_reflect_kernel below for the task list)
__device__ __forceinline__ void pad_task(const LevelDev& lv, int fill, int task, int slot, int lane)
{
    const int w = lv.w, h = lv.h, n_band = 2 * lv.pad_y;
    uint8_t* img = lv.img + (long long)slot * lv.step;
    if (task < n_band) {
        const int y = task < lv.pad_y ? task - lv.pad_y : h + (task - lv.pad_y);
        const uint8_t* srow = img + (long long)reflect101(y, h) * lv.pitch;
        uint8_t* drow = img + (long long)y * lv.pitch;
        const int wq = (w + 2 * fill + 3) >> 2;                           // dwords of a band row (from x = -fill)
        for (int q = lane; q < wq; q += 64)
            *reinterpret_cast<uint32_t*>(drow + (q * 4 - fill)) = border_dword(srow, q * 4 - fill, w);
        return;
    }
    const int y = (task - n_band) * 8 + (lane >> 3);
    const int c = lane & 7;
    if (y >= h) return;
    uint8_t* row = img + (long long)y * lv.pitch;
    const int r0 = w >> 2;                                                // the dword that holds the last interior bytes (mixed if w % 4)
    const int rq = ((w + fill + 3) >> 2) - r0;                            // right border dwords incl. that one
    if ((fill & 15) == 0 && w >= fill + 24) {
        const int nl = fill >> 4;                                         // 16-byte chunks of the left border
        if (c < nl) {
            // dst bytes x0 .. x0+15 (x0 = -fill + 16 c) = src[-x0], src[-x0 - 1], ...: dword j of the chunk is bytes
            // (s + 4, s + 3, s + 2, s + 1) with s = -x0 - 4 j - 4, i.e. [hi.b0, lo.b3, lo.b2, lo.b1] of the aligned pair at s
            const int s0 = fill - 16 * c - 16;                            // s of j = 3; multiple of 4
            const uint32_t* q = reinterpret_cast<const uint32_t*>(row + s0);
            const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3], d4 = q[4];
            uint4 o;
            o.x = __builtin_amdgcn_perm(d4, d3, 0x01020304u); o.y = __builtin_amdgcn_perm(d3, d2, 0x01020304u);
            o.z = __builtin_amdgcn_perm(d2, d1, 0x01020304u); o.w = __builtin_amdgcn_perm(d1, d0, 0x01020304u);
            *reinterpret_cast<uint4*>(row - fill + 16 * c) = o;
        } else if (c == nl) {
            *reinterpret_cast<uint32_t*>(row + 4 * r0) = border_dword(row, 4 * r0, w);        // the mixed dword
        } else {
            // pure border dwords xq = 4 (r0 + 1 + 4 m + i): bytes src[a], src[a-1], src[a-2], src[a-3], a = 2w - 2 - xq;
            // a mod 4 = (2w - 2) mod 4 = e for every one of them, so one selector serves the row: byte k of the result is byte
            // e - k of the aligned dword A = a >> 2, or byte 4 + e - k of dword A - 1
            const int m = c - nl - 1, first = 1 + 4 * m;                  // index (from r0) of the chunk's first dword
            if (first < rq) {
                const int e = (2 * w - 2) & 3;
                uint32_t sel = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) sel |= (uint32_t)(e - k >= 0 ? 4 + e - k : 4 + e - k) << (8 * k);   // index into {hi = A, lo = A-1}: 4+e-k
                const int xq = 4 * (r0 + first);
                const int A = (2 * w - 2 - xq) >> 2;                      // dword of the first result; the chunk needs A-4 .. A
                const uint32_t* q = reinterpret_cast<const uint32_t*>(row) + (A - 4);
                const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3], d4 = q[4];
                const uint32_t o0 = __builtin_amdgcn_perm(d4, d3, sel), o1 = __builtin_amdgcn_perm(d3, d2, sel);
                const uint32_t o2 = __builtin_amdgcn_perm(d2, d1, sel), o3 = __builtin_amdgcn_perm(d1, d0, sel);
                uint32_t* dq = reinterpret_cast<uint32_t*>(row + xq);
                const int n = rq - first;                                 // dwords left to write (>= 1)
                if (n >= 4) { dq[0] = o0; dq[1] = o1; dq[2] = o2; dq[3] = o3; }
                else { dq[0] = o0; if (n > 1) dq[1] = o1; if (n > 2) dq[2] = o2; }
            }
        }
    } else {
        // narrow levels / odd fill widths: one dword per step with the general reflection
        const int lq = fill >> 2;
        for (int k = c; k < lq + rq; k += 8) {
            const int xq = k < lq ? k * 4 - fill : (r0 + (k - lq)) * 4;
            *reinterpret_cast<uint32_t*>(row + xq) = border_dword(row, xq, w);
        }
    }
}

// TAG only separates the symbol of the level-0 launch (the dominant, roofline-graded kernel) from the
// smaller levels so that rocprofv3 --stats reports it on its own row.
template <bool DOWN, bool DERIV, int TAG>
__global__ __launch_bounds__(NT) void pyr_level_kernel(const PyrArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t tile[LR * LP];

    const int tid = threadIdx.x;
    // The border of the launch's source level (levels >= 1) rides along: the workgroups behind the tiles each run four wave-tasks
    // of pad_task; they read the level's interior, which the previous launch completed, and write where no tile reads or writes.
    if (TAG != 0 && a.pad_blocks > 0 && blockIdx.x >= (unsigned)(a.tiles_x * a.tiles_y) * (unsigned)a.n_slots) {
        const unsigned r = blockIdx.x - (unsigned)(a.tiles_x * a.tiles_y) * (unsigned)a.n_slots;
        const unsigned fz = r / (unsigned)a.pad_blocks, t4 = r - fz * (unsigned)a.pad_blocks;
        const int task = __builtin_amdgcn_readfirstlane((int)t4 * 4 + (tid >> 6));
        if (task < 2 * a.src.pad_y + (a.src.h + 7) / 8) pad_task(a.src, a.pad_fill, task, a.first + (int)fz, tid & 63);
        return;
    }
    // XCD-aware block -> (frame, tile) map.  Workgroups are dealt round-robin over the 8 XCDs (blocks b
    // and b+8 share an XCD and its private L2).  XCD x walks frames x, x+8, ... tile by tile in raster
    // order, so the halo lines a tile shares with its x/y neighbours are L2 hits instead of being
    // fetched again by another XCD (measured: ~3x read over-fetch with the plain 3-D grid).
    // Placement only affects speed, never results.
    int fz, tix, tiy;
    {
        const unsigned b = blockIdx.x, tpf = (unsigned)(a.tiles_x * a.tiles_y);
        const unsigned full = (unsigned)a.n_swz * tpf;
        unsigned t;
        if (b < full) { const unsigned k = b >> 3; const unsigned g = k / tpf; t = k - g * tpf; fz = (int)(g * 8u + (b & 7u)); }
        else          { const unsigned r = b - full; const unsigned g = r / tpf; t = r - g * tpf; fz = a.n_swz + (int)g; }
        tiy = (int)(t / (unsigned)a.tiles_x); tix = (int)(t - (unsigned)tiy * (unsigned)a.tiles_x);
    }
    const int x0 = tix * TW, y0 = tiy * TH;
    const int slot = a.first + fz;
    const int w = a.src.w, h = a.src.h, pitch = a.src.pitch;
    const uint8_t* __restrict__ src = a.src.img + (long long)slot * a.src.step;

    // ---- stage the tile: coalesced dword loads of whole rows --------------------------------
    for (int i = tid; i < LR * (LW / 4); i += NT) {
        const int r = i / (LW / 4), c = i - r * (LW / 4);
        const int y = reflect101(y0 - 2 + r, h);
        const int x = x0 - 4 + 4 * c;
        uint32_t v = 0;
        if (x >= 0 && x < pitch) v = *reinterpret_cast<const uint32_t*>(src + (long long)y * pitch + x);
        *reinterpret_cast<uint32_t*>(tile + r * LP + 4 * c) = v;
    }
    __syncthreads();

    // ---- REFLECT_101 column patch (only tiles touching the left / right image edge) ----------
    // LDS column of image column x is (x - x0 + 4).
    const bool left_edge = (x0 == 0);
    const bool right_edge = (x0 + TW + 4 > w);
    if (left_edge || right_edge) {
        for (int i = tid; i < LR * 5; i += NT) {
            const int r = i / 5, k = i - r * 5;
            int xd, xs;                        // destination / source image columns
            if (k < 2) { if (!left_edge) continue; xd = -1 - k; xs = 1 + k; }
            else       { if (!right_edge) continue; xd = w + (k - 2); xs = w - 2 - (k - 2); }
            const int cd = xd - x0 + 4, cs = xs - x0 + 4;
            if (cd >= 0 && cd < LW && cs >= 0 && cs < LW) tile[r * LP + cd] = tile[r * LP + cs];
        }
        __syncthreads();
    }

    // ---- Scharr derivatives of level l --------------------------------------------------------
    if (DERIV) {
        uint8_t* __restrict__ der = reinterpret_cast<uint8_t*>(a.src.der) + (long long)slot * a.src.dstep;
#pragma unroll
        for (int j = 0; j < (TW / 4) * TH / NT; ++j) {
            const int i = tid + j * NT;
            const int r = i / (TW / 4), c4 = i - r * (TW / 4);
            const int x = x0 + 4 * c4, y = y0 + r;
            if (x < w && y < h) {
                // image columns x-1 .. x+4  ->  LDS bytes 4*c4+3 .. 4*c4+8 of rows r+1 .. r+3.
                // Packed 16-bit lanes: per row the six columns become three u16 pairs A = (-1,0),
                // B = (1,2), C = (3,4) (one v_perm_b32 each); the vertical [3 10 3] / [-1 0 1] and the
                // horizontal differences then run two columns per v_pk_* instruction.
                const uint32_t* p0 = reinterpret_cast<const uint32_t*>(tile + (r + 1) * LP) + c4;
                const uint32_t* p1 = reinterpret_cast<const uint32_t*>(tile + (r + 2) * LP) + c4;
                const uint32_t* p2 = reinterpret_cast<const uint32_t*>(tile + (r + 3) * LP) + c4;
                const uint32_t u0 = p0[0], u1 = p0[1], u2 = p0[2];
                const uint32_t m0 = p1[0], m1 = p1[1], m2 = p1[2];
                const uint32_t d0 = p2[0], d1 = p2[1], d2 = p2[2];
                constexpr uint32_t SEL_EDGE = 0x0c040c03u;   // (S1.byte3, S0.byte0) -> two u16
                constexpr uint32_t SEL_MID  = 0x0c020c01u;   // (S1.byte1, S1.byte2) -> two u16
                const u16x2 uA = as_u16x2(__builtin_amdgcn_perm(u1, u0, SEL_EDGE)), uB = as_u16x2(__builtin_amdgcn_perm(u1, u1, SEL_MID)),
                            uC = as_u16x2(__builtin_amdgcn_perm(u2, u1, SEL_EDGE));
                const u16x2 mA = as_u16x2(__builtin_amdgcn_perm(m1, m0, SEL_EDGE)), mB = as_u16x2(__builtin_amdgcn_perm(m1, m1, SEL_MID)),
                            mC = as_u16x2(__builtin_amdgcn_perm(m2, m1, SEL_EDGE));
                const u16x2 dA = as_u16x2(__builtin_amdgcn_perm(d1, d0, SEL_EDGE)), dB = as_u16x2(__builtin_amdgcn_perm(d1, d1, SEL_MID)),
                            dC = as_u16x2(__builtin_amdgcn_perm(d2, d1, SEL_EDGE));
                const u16x2 three = {3, 3}, ten = {10, 10};
                // t0 = 3 (up + dn) + 10 mid ; t1 = dn - up   (mod 2^16 == exact: |values| <= 4080)
                const u16x2 t0A = (uA + dA) * three + mA * ten, t0B = (uB + dB) * three + mB * ten, t0C = (uC + dC) * three + mC * ten;
                const u16x2 t1A = dA - uA, t1B = dB - uB, t1C = dC - uC;
                const u16x2 ix01 = t0B - t0A, ix23 = t0C - t0B;                 // t0[k+1] - t0[k-1]
                const u16x2 t1M = as_u16x2(__builtin_amdgcn_alignbit(as_u32(t1B), as_u32(t1A), 16));   // (t1[0], t1[1])
                const u16x2 t1N = as_u16x2(__builtin_amdgcn_alignbit(as_u32(t1C), as_u32(t1B), 16));   // (t1[2], t1[3])
                const u16x2 iy01 = (t1A + t1B) * three + t1M * ten, iy23 = (t1B + t1C) * three + t1N * ten;
                uint32_t o[4];
                o[0] = __builtin_amdgcn_perm(as_u32(iy01), as_u32(ix01), 0x05040100u);   // (Ix0, Iy0)
                o[1] = __builtin_amdgcn_perm(as_u32(iy01), as_u32(ix01), 0x07060302u);   // (Ix1, Iy1)
                o[2] = __builtin_amdgcn_perm(as_u32(iy23), as_u32(ix23), 0x05040100u);
                o[3] = __builtin_amdgcn_perm(as_u32(iy23), as_u32(ix23), 0x07060302u);
                // zero the lanes' pixels beyond w so the pitch slack stays deterministic
#pragma unroll
                for (int k = 1; k < 4; ++k) if (x + k >= w) o[k] = 0;
                // written once, read by the tracker launches later from HBM: a non-temporal store (no L2 allocation) moves this
                // 80 % of the kernel's bytes a quarter faster than a plain one (0.155 -> 0.118 ms for the level-0 launch)
                typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
                const u32x4_t v = {o[0], o[1], o[2], o[3]};
#ifndef PYR_PLAIN_STORE
                __builtin_nontemporal_store(v, reinterpret_cast<u32x4_t*>(der + (long long)y * a.src.dpitch + 4 * x));
#else
                *reinterpret_cast<u32x4_t*>(der + (long long)y * a.src.dpitch + 4 * x) = v;
#endif
            }
        }
    }

    // ---- Gaussian pyrDown into level l+1 --------------------------------------------------------
    if (DOWN) {
        const int dw = a.dst.w, dh = a.dst.h;
        uint8_t* __restrict__ dst = a.dst.img + (long long)slot * a.dst.step;
        const int yi = tid / (TW / 8), g = tid - yi * (TW / 8);
        const int X = x0 / 2 + 4 * g, Y = y0 / 2 + yi;
        if (X < dw && Y < dh) {
            // v_dot4_u32_u8: the horizontal taps (1,4,6,4) of one output are one dot4 on 4 consecutive bytes,
            // the fifth tap a second dot4 with weights (1,0,0,0) on the next 4; the vertical weight of row j
            // is folded into both byte-weight vectors, so an output is a chain of 10 dot4 accumulations.
            uint32_t acc[4] = {128u, 128u, 128u, 128u};                       // rounding term of (v + 128) >> 8
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                // image columns 2X-2 .. 2X+8  ->  LDS bytes 8g+2 .. 8g+12 of row 2*yi + j
                const uint32_t* q = reinterpret_cast<const uint32_t*>(tile + (2 * yi + j) * LP + 8 * g);
                const uint32_t w0 = q[0], w1 = q[1], w2 = q[2], w3 = q[3];
                const uint32_t wj = (j == 0 || j == 4) ? 1u : (j == 2 ? 6u : 4u);
                const uint32_t Wv = wj * 0x04060401u;                            // (1,4,6,4) * wj, each <= 36
                const uint32_t Ev = wj;                                          // (1,0,0,0) * wj
                const uint32_t a02 = __builtin_amdgcn_alignbyte(w1, w0, 2);     // bytes 2..5
                const uint32_t a06 = __builtin_amdgcn_alignbyte(w2, w1, 2);     // bytes 6..9
                const uint32_t a10 = __builtin_amdgcn_alignbyte(w3, w2, 2);     // bytes 10..13
                acc[0] = __builtin_amdgcn_udot4(a02, Wv, __builtin_amdgcn_udot4(a06, Ev, acc[0], false), false);
                acc[1] = __builtin_amdgcn_udot4(w1,  Wv, __builtin_amdgcn_udot4(w2,  Ev, acc[1], false), false);
                acc[2] = __builtin_amdgcn_udot4(a06, Wv, __builtin_amdgcn_udot4(a10, Ev, acc[2], false), false);
                acc[3] = __builtin_amdgcn_udot4(w2,  Wv, __builtin_amdgcn_udot4(w3,  Ev, acc[3], false), false);
            }
            uint32_t o = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t v = (X + k < dw) ? (acc[k] >> 8) : 0u;
                o |= v << (8 * k);
            }
            *reinterpret_cast<uint32_t*>(dst + (long long)Y * a.dst.pitch + X) = o;     // read again by the next level's launch: a plain store
        }
    }
}

// REFLECT_101 border of the padded levels of a batch: OpenCV pads every pyramid level by winSize (SURVEY A-1); doing the
// same physically keeps the tracker on plain addressing everywhere.  The work of one frame is a flat list of wave-tasks
// over all levels (no idle workgroups): a band task is one of the pad_y rows above / below the image (the reflected
// source row copied with aligned dword loads, byte gathers only at its two ends); a side task is the left + right
// border of eight interior rows.  Side borders are mirror images of the row's own ends, so a lane builds a 16-byte chunk of
// border from five dwords of the row with four v_perm_b32 (the reversal is a constant byte selector on the left; on the right
// it depends on w mod 4 only) and stores it with one instruction: 7 lanes per row instead of 17 narrow accesses with byte
// gathers, which is what this kernel's time was made of.  Only fill_x columns each side are written (what the tracker can
// read; level 0 reserves more for alignment).  Reads touch the interior only, writes the border (the mixed dword at the right
// edge rewrites its interior bytes with themselves).  The derivative border is zero from allocation on.
__global__ __launch_bounds__(256) void pad_reflect_kernel(const PadArgs a)
{
    const int lane = threadIdx.x & 63;
    const int task_all = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + (int)(threadIdx.x >> 6));
    int l = 0;
    while (l + 1 < a.n_lv && task_all >= a.task0[l + 1]) ++l;           // wave-uniform
    pad_task(a.lv[l], a.fill_x[l], task_all - a.task0[l], a.first + (int)blockIdx.y, lane);
}

// A frame enters its slot: level 0 is written WITH its REFLECT_101 border -- whole padded rows, full-line stores -- while it is
// being written anyway (dr3hip_batch_upload: the host frame has crossed PCIe as one linear copy into a bounce buffer; this kernel
// is the device-side change of pitch).  pad_reflect_kernel then only serves the levels the pyramid pass produces.
__global__ __launch_bounds__(256) void upload_level0_kernel(const uint8_t* __restrict__ src0, long long spitch, LevelDev lv, int slot0, int fill,
                                                            long long frame_step)
{
    const uint8_t* __restrict__ src = src0 + (long long)blockIdx.z * frame_step;       // grid z: frames of one chunk
    const int slot = slot0 + (int)blockIdx.z;
    const int y = (int)blockIdx.y - lv.pad_y;                              // padded rows -pad_y .. h + pad_y - 1
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int wq = (lv.w + 2 * fill + 3) >> 2;
    if (q >= wq) return;
    const int xq = q * 4 - fill;
    const uint8_t* srow = src + (long long)reflect101(y, lv.h) * spitch;
    uint32_t v;
    if (xq >= 0 && xq + 3 < lv.w && ((reinterpret_cast<uintptr_t>(srow) + (uintptr_t)xq) & 3) == 0) {
        v = *reinterpret_cast<const uint32_t*>(srow + xq);
    } else {
        v = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) v |= (uint32_t)srow[reflect101(xq + k, lv.w)] << (8 * k);
    }
    uint8_t* img = lv.img + (long long)slot * lv.step;
    *reinterpret_cast<uint32_t*>(img + (long long)y * lv.pitch + xq) = v;
}

}  // namespace

hipError_t launch_pad_reflect(const LevelDev* lv, int n_lv, int fill_x, int first, int n_slots, hipStream_t s, const LaunchEv* ev)
{
    if (n_slots <= 0 || n_lv <= 0 || lv[0].pad_x <= 0) return hipSuccess;
    PadArgs a; a.first = first; a.n_lv = n_lv;
    int tasks = 0;
    for (int i = 0; i < n_lv; ++i) {
        a.lv[i] = lv[i];
        a.fill_x[i] = fill_x < lv[i].pad_x ? fill_x : lv[i].pad_x;
        a.task0[i] = tasks;
        tasks += (2 * lv[i].pad_y + (lv[i].h + 7) / 8 + 3) & ~3;          // whole workgroups per level: a wave never straddles two
    }
    dim3 block(256), grid(tasks / 4, n_slots);                           // one launch for every padded level
    DR3_LAUNCH(ev, pad_reflect_kernel, grid, block, 0, s, a);
    return hipGetLastError();
}

hipError_t launch_upload_level0(const uint8_t* d_src, size_t spitch, const LevelDev& lv, int slot, int fill_x, hipStream_t s, int n_frames,
                                size_t frame_step)
{
    if (n_frames <= 0) return hipSuccess;
    const int fill = fill_x < lv.pad_x ? fill_x : lv.pad_x;
    const int wq = (lv.w + 2 * fill + 3) / 4;
    for (int f0 = 0; f0 < n_frames; f0 += 32768) {
        const int n = n_frames - f0 < 32768 ? n_frames - f0 : 32768;
        dim3 block(256), grid((wq + 255) / 256, lv.h + 2 * lv.pad_y, n);
        hipLaunchKernelGGL(upload_level0_kernel, grid, block, 0, s, d_src + (size_t)f0 * frame_step, (long long)spitch, lv, slot + f0, fill,
                           (long long)frame_step);
    }
    return hipGetLastError();
}

hipError_t launch_pyr_level(const PyrArgs& a_in, int n_slots, bool down, bool deriv, bool level0, hipStream_t s, const LaunchEv* ev, int pad_fill)
{
    if (n_slots <= 0 || (!down && !deriv)) return hipSuccess;
    PyrArgs a = a_in;
    a.tiles_x = (a.src.w + TW - 1) / TW;
    a.tiles_y = (a.src.h + TH - 1) / TH;
    a.n_swz = n_slots & ~7;
    a.n_slots = n_slots;
    a.pad_fill = pad_fill < a.src.pad_x ? pad_fill : a.src.pad_x;
    a.pad_blocks = (!level0 && pad_fill > 0 && a.src.pad_x > 0) ? (2 * a.src.pad_y + (a.src.h + 7) / 8 + 3) / 4 : 0;
    dim3 grid(((unsigned)(a.tiles_x * a.tiles_y) + (unsigned)a.pad_blocks) * (unsigned)n_slots);
    dim3 block(NT);
    if (level0) {
        if (down && deriv)  DR3_LAUNCH(ev, (pyr_level_kernel<true, true, 0>),  grid, block, 0, s, a);
        else if (down)      DR3_LAUNCH(ev, (pyr_level_kernel<true, false, 0>), grid, block, 0, s, a);
        else                DR3_LAUNCH(ev, (pyr_level_kernel<false, true, 0>), grid, block, 0, s, a);
    } else {
        if (down && deriv)  DR3_LAUNCH(ev, (pyr_level_kernel<true, true, 1>),  grid, block, 0, s, a);
        else if (down)      DR3_LAUNCH(ev, (pyr_level_kernel<true, false, 1>), grid, block, 0, s, a);
        else                DR3_LAUNCH(ev, (pyr_level_kernel<false, true, 1>), grid, block, 0, s, a);
    }
    return hipGetLastError();
}

}  // namespace dr3
