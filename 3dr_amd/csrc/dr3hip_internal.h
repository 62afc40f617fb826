// dr3hip_internal.h -- shared between the host API (dr3hip_api.cpp) and the kernel launchers.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <string>

#include "../../include/dr3hip.h"

namespace dr3 {

// ---- device-side argument blocks (passed by value as kernel arguments) -----------------------

// one pyramid level of a frame batch
struct LevelDev {
    uint8_t* img;          // u8 plane of slot 0
    int16_t* der;          // interleaved (Ix,Iy) plane of slot 0
    int w, h;
    int pitch;             // bytes per image row   (>= w + 4, multiple of 64)
    int dpitch;            // bytes per derivative row (>= 4*w, multiple of 128)
    long long step;        // bytes between slots (image)
    long long dstep;       // bytes between slots (derivative)
    int pad_x, pad_y;      // physical border around the plane: REFLECT_101 pixels / zero derivatives (OpenCV pads every level
                           // by winSize the same way), so the tracker addresses every window plainly; img / der point at
                           // the interior origin.  pad_x is a multiple of 4.  0 = none (planes outside a frame batch)
};

struct PyrArgs {
    LevelDev src;          // level l
    LevelDev dst;          // level l + 1 (unused when !DOWN)
    int first;             // first slot of the launch
    int tiles_x, tiles_y;  // filled by the launcher
    int n_swz;             // slots (multiple of 8) covered by the XCD-aware block map
    // levels >= 1: the launch also writes the REFLECT_101 border of its own SOURCE level (read by the tracker only) with
    // pad_blocks extra workgroups per slot behind the tile workgroups (0: none); fill = border columns each side
    int pad_blocks, pad_fill, n_slots;
};

struct PadArgs {           // pad_reflect_kernel: the padded levels of one build, one launch
    LevelDev lv[DR3HIP_MAX_LEVELS];
    int fill_x[DR3HIP_MAX_LEVELS];   // border columns written each side (<= pad_x, multiple of 4)
    int task0[DR3HIP_MAX_LEVELS];    // first wave-task of each level in the flat task list of a frame
    int n_lv, first;
};

struct LKArgs {
    LevelDev lv[DR3HIP_MAX_LEVELS];
    int max_level;
    int win_w, win_h;
    int max_count;         // clamped iteration limit
    int flags;
    int n_pts;
    int first_slot;
    double epsilon;        // clamped and squared
    double min_eig;
    const int* prev_slot;
    const int* next_slot;
    const float* prev_xy;
    float* next_xy;
    uint8_t* status;
    float* err;
    int* iters;
    uint8_t* reason;       // optional: DR3HIP_LK_LOST_* per point (0 = tracked)
};

struct WarpJobDev {
    const uint8_t* src; int sw, sh, sstride;
    uint8_t* dst;       int dw, dh, dstride;
    double M[9];        // inverse map (dst -> src), already inverted on the host
    int bw0;            // WarpPerspectiveInvoker block width (x0/x1 split)
    int bw0_shift;      // log2(bw0) if bw0 is a power of two >= 64; 31 if one block spans the row; else -1 (generic kernel)
    int fast;           // 1: the guarded fast coordinate path may be used (finite M, source <= 8191 x 8191); 2: also its
                        // one-reciprocal-per-four-pixels form (nearly affine map: warp_delta_coords_ok); 3: also the tile kernel's
                        // one-anchor-per-column-of-16-pixels form (tile_delta_coords_ok)
    int tile;           // 8UC3: 0 = direct gather kernel; 2 = LDS-staged tile kernel (fast == 3 and every tile fits the LDS budget)
    int tile_lds;       // tile != 0: LDS bytes of the largest source box of a wave's strip (tile_kernel_lds; an estimate: a strip
                        // that needs more takes the kernel's slow way out)
    int tile_guard;     // tile != 0: the tie guard of the strip kernel in units of 2^-13 of 1/32 px (1 or 2, from the job's error bound)
};

struct StitchArgs {        // fused Stitch compose (src/stitch.cpp:72-80): job[0] = left (M = T), job[1] = right (M = T * Hinv)
    WarpJobDev job[2];     // .dst / .dw / .dh / .dstride of the jobs are the output canvas (same for both)
    uint8_t* dst; int dw, dh, dstride;
    float alpha, beta, gamma;
};

struct PanoAddArgs {       // fused Panorama::add_img_to_canvas (src/panorama.cpp:178-212)
    const uint8_t* img; int w, h, istride;        // 8UC3 source image
    float* canvas;      int cw, ch, cstride;      // 32FC4 weighted canvas (read-modify-write)
    double M[9];        int bw0, bw0_shift, fast; // inverse map + WarpPerspectiveInvoker block split, as WarpJobDev (fast: over the box)
    int tile, tile_lds, tile_guard;               // the strip plan of the box, as WarpJobDev (paste_strip_kernel)
    int cx0, cx1;                                 // canvas columns [cx0, cx1) = [min_x, max_x) clipped to the canvas
    int cy0, cy1;                                 // canvas rows that can receive a non-zero sample (host bound; all rows if unsure)
    int slab_y0;                                  // canvas row that `canvas` points at (0 unless a GPU owns a slab of rows)
    int min_x;                                    // ramp[x - min_x]
    const float* ramp;                            // device, max_x - min_x entries
};

struct PanoNormArgs {      // Panorama::normalize_canvas (src/panorama.cpp:215-229)
    const float* canvas; int cw, ch, cstride;
    uint8_t* out;        int ostride;
};

struct FastArgs {          // FAST-10 score map + grid-best selection of one pyramid level (fast_kernels.hip)
    const uint8_t* img; int w, h, stride;         // level image
    uint8_t* score;     int spitch;               // u8 score map (0 = no corner), null: not wanted; spitch a multiple of 64 >= w
    int threshold;                                // FAST threshold (reference: 20)
    int level, cell_size, grid_cols;
    float det_threshold;                          // (float)detection_threshold: the initial Corner score of every cell
    const uint8_t* occupancy;                     // device, grid cells, may be null
    unsigned long long* cells;                    // device, per cell: (score bits << 32) | ~raster order; null: score map only
    int n_slots;                                  // grid z: frames (0 or 1: a single image)
    long long img_step, score_step;               // bytes between frames
    int n_cells;                                  // cells (and occupancy bytes) per frame
};

struct AlignArgs {         // sparse_align_kernel (align_kernels.hip): one workgroup per frame pair
    LevelDev lv[DR3HIP_MAX_LEVELS];
    float fx, fy, cx, cy;
    int max_level, min_level, n_iter;
    double eps;
    int n_pts, first_slot;
    const int* ref_slot; const int* cur_slot;     // optional slot indirection, as dr3hip_batch_track
    const float* xyz;                             // n_pairs x n_pts x 3, reference camera frame, Z <= 0 = no point
    float* cache;                                 // workspace: n_pairs x ceil(n_pts / 256) x 16 x 7 x 256 floats
    double* T;                                    // n_pairs x 12, in/out
    int* n_meas; float* chi2; int* iters;         // n_pairs (iters: n_pairs x DR3HIP_MAX_LEVELS), each may be null
};

struct NormArgs {          // InitHelper::Normalize for n_sets point sets (fast_kernels.hip)
    const float* xy; long long xy_step;           // set k: n points (x, y) at xy + k * xy_step floats
    const uint8_t* mask; long long mask_step;     // optional: 0 = the point was erased (lost by the tracker); bytes between sets
    int n, n_sets;
    float* out_xy;                                // same layout as xy
    float* T;                                     // n_sets x 9
};

struct CheckFArgs {        // InitHelper::CheckFundamental for n_hyp matrices (fast_kernels.hip)
    const float* F; int n_hyp;                    // n_hyp x 9, row-major
    const uint8_t* mask;                          // optional (device entry point): matches with mask 0 were erased, they are skipped
    const float* pts1; const float* pts2; int n;  // n x 2 each
    float inv_sigma_square;
    float* scores;                                // n_hyp
    uint8_t* inliers;                             // n_hyp x n or null
};

struct CellsArgs {         // decode the per-cell keys of fast_select_kernel into corners / tracker input points
    const unsigned long long* cells; int n_total; // n_slots * n_cells
    unsigned long long empty_key;                 // the initial key: the cell holds no feature
    float det_threshold;
    float* xy;                                    // n_total x 2, level-0 pixels, or (DR3HIP_NO_POINT, DR3HIP_NO_POINT); may be null
    dr3hip_corner* corners;                       // n_total, may be null
};

struct RemapArgs {
    const uint8_t* src; int sw, sh, sstride;
    uint8_t* dst;       int dw, dh, dstride;
    const float* mapx; const float* mapy; int map_stride;   // bytes (mode 0)
    int mode;                                               // 0 maps from memory, 1 spherical, 2 cylindrical (generated)
    const double* tx0; const double* tx1;                   // per-column tables (dw entries each), see remap_kernel
    const double* ty0; const double* ty1;                   // per-row tables (dh entries each)
    float f, half_w, half_h;
};

struct HalfArgs {
    const uint8_t* src; int w, h, sstride;
    uint8_t* dst;       int ow, oh, dstride;
    int mode;           // DR3HIP_HALF_TRUNC / _SSE2 / _REF_SCALAR (REF_X86 resolved on the host)
    int lstride;        // the step of the cv::Mat the reference would hold (REF_SCALAR's flat walk); == sstride unless pitched
    int n_slots;        // grid z: frames, sstep / dstep bytes apart (0 or 1: a single image)
    long long sstep, dstep;
};

// ---- kernel launchers (defined in the .hip files) ----------------------------------------------
// Timing events that ride on a kernel dispatch (its own start / end timestamps: no event packets between kernels).  DR3_LAUNCH
// launches with them when a launcher was handed any, plainly otherwise.
struct LaunchEv { hipEvent_t start = nullptr, stop = nullptr; };
#define DR3_LAUNCH(ev, kernel, grid, block, shmem, stream, ...)                                                            \
    do {                                                                                                                   \
        if ((ev) && ((ev)->start || (ev)->stop))                                                                           \
            hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, (ev)->start, (ev)->stop, 0, __VA_ARGS__);            \
        else                                                                                                               \
            hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                                           \
    } while (0)

hipError_t launch_pyr_level(const PyrArgs& a, int n_slots, bool down, bool deriv, bool level0, hipStream_t s, const LaunchEv* ev = nullptr,
                            int pad_fill = 0);      // pad_fill > 0 (levels >= 1): the source level's border rides along
hipError_t launch_pad_reflect(const LevelDev* lv, int n_lv, int fill_x, int first, int n_slots, hipStream_t s, const LaunchEv* ev = nullptr);
// n_frames frames, frame_step bytes apart in d_src, go to the slots slot, slot + 1, ... (grid z)
hipError_t launch_upload_level0(const uint8_t* d_src, size_t spitch, const LevelDev& lv, int slot, int fill_x, hipStream_t s, int n_frames = 1,
                                size_t frame_step = 0);
hipError_t launch_lk_track(const LKArgs& a, int n_pairs, hipStream_t s, const LaunchEv* ev = nullptr);
// counters over the tracker's outputs: cnt = the unsigned long long fields of dr3hip_lk_counters in declaration order
hipError_t launch_lk_stats(long long n, const uint8_t* status, const uint8_t* reason, const int* iters, int levels,
                           unsigned long long* cnt, hipStream_t s);
// tile_mode (8UC3 only): 0 = direct gather kernel; 2 = LDS-staged tile kernel (jobs with WarpJobDev::tile == 2)
// tile_lds: LDS bytes of the largest tile source box of the launch's jobs (WarpJobDev::tile_lds)
hipError_t launch_warp(const WarpJobDev* d_jobs, int n_jobs, int max_dw, int max_dh, int type, bool generic, hipStream_t s, int tile_mode = 0,
                       int tile_lds = 0);
constexpr int kWarpTileW = 64, kWarpTileH = 64, kWarpStripRows = 16;   // warp_tile_8uc3_kernel: a workgroup's tile, a wave's strip of it
constexpr int kWarpTileLdsBytes = 15 * 1024;                          // the largest LDS slice of a wave (4 slices + tie lists <= 64 KiB)
hipError_t launch_remap(const RemapArgs& a, int type, hipStream_t s);
hipError_t launch_halfsample(const HalfArgs& a, hipStream_t s);
hipError_t launch_fast_level(const FastArgs& a, hipStream_t s);     // score (map if a.score) + selection (if a.cells) of one level
hipError_t launch_cells_decode(const CellsArgs& a, hipStream_t s);
hipError_t launch_normalize_points(const NormArgs& a, hipStream_t s);
hipError_t launch_sparse_align(const AlignArgs& a, int n_pairs, hipStream_t s);
bool stitch_takes_strips(const StitchArgs& a);      // host only: would launch_stitch_compose run the strip form on these records?
hipError_t launch_stitch_compose(const StitchArgs& a, hipStream_t s);
hipError_t launch_panorama_add(const PanoAddArgs& a, hipStream_t s);
bool paste_takes_strips(const PanoAddArgs& a);       // host only: would launch_panorama_add run the strip form on this record?
hipError_t launch_panorama_normalize(const PanoNormArgs& a, hipStream_t s);

// geometry helpers shared by host and launchers
// image sides the entry points accept: row bytes (side x 16-byte pixels) and 2 x side then stay inside int
static inline bool side_ok(int v) { return v >= 1 && v <= (1 << 26); }
static inline int align_up(int v, int a) { return (v + a - 1) / a * a; }
static inline long long align_up(long long v, long long a) { return (v + a - 1) / a * a; }
static inline int type_channels(int type) {
    switch (type) { case DR3HIP_8UC1: return 1; case DR3HIP_8UC3: return 3; case DR3HIP_8UC4: return 4;
                    case DR3HIP_32FC4: return 4; case DR3HIP_32FC1: return 1; default: return 0; }
}
static inline int type_elem_bytes(int type) {
    return type_channels(type) * ((type == DR3HIP_32FC4 || type == DR3HIP_32FC1) ? 4 : 1);
}

}  // namespace dr3
