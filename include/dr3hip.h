/*
 * dr3hip.h -- C ABI of libdr3hip.so: the MI355X (gfx950) implementation of 3DR's dense hot path.
 *
 * The reference (kvmanohar22/3DR) has no plugin/FFI layer; its boundary for this path is a set of
 * C++ call expressions into OpenCV and into its own utils::.  Each entry point below replaces one
 * of them (reference file:line cited per function).  Conventions:
 *   - plain pointers and sizes only; the caller owns every buffer it passes; nothing is freed or
 *     retained by the callee beyond the call (device-resident objects are explicit handles);
 *   - no C++ exceptions cross the boundary: every call returns 0 / a non-negative result on
 *     success or a negative DR3HIP_E* code, with text in dr3hip_last_error(ctx);
 *   - one dr3hip_ctx per GPU; a ctx is NOT thread-safe, distinct ctxs are;
 *   - "host" calls are synchronous on return; "device" calls are enqueued on the ctx stream and
 *     complete at dr3hip_sync().
 * There is no CPU fallback: without a HIP device dr3hip_init fails with DR3HIP_ENODEV.
 */
#ifndef DR3HIP_H
#define DR3HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DR3HIP_VERSION 104
#define DR3HIP_BATCH_MAX_W 4000000   /* widest frame of a dr3hip_batch: the tracker addresses rows of 4 (w + 2 pad) + 16 bytes with 24-bit pitches */

/* error codes */
#define DR3HIP_OK        0
#define DR3HIP_EINVAL   -1   /* bad argument                      */
#define DR3HIP_ENODEV   -2   /* no usable HIP device              */
#define DR3HIP_ENOMEM   -3   /* device or host allocation failed  */
#define DR3HIP_EHIP     -4   /* HIP runtime error (see last_error) */
#define DR3HIP_ERANGE   -5   /* parameter outside the supported range (e.g. window > 64) */

/* pixel types (the three the reference passes to the warp calls, plus two for generality) */
#define DR3HIP_8UC1   0      /* gray,  cv CV_8UC1                                  */
#define DR3HIP_8UC3   1      /* BGR,   cv CV_8UC3  (src/stitch.cpp:73-74)          */
#define DR3HIP_32FC4  2      /* BGRA f32, CV_32FC4 (src/panorama.cpp:192)          */
#define DR3HIP_8UC4   3
#define DR3HIP_32FC1  4

/* flags */
#define DR3HIP_LK_USE_INITIAL_FLOW    4   /* == cv::OPTFLOW_USE_INITIAL_FLOW (src/initialization.cpp:613) */
#define DR3HIP_LK_GET_MIN_EIGENVALS   8   /* == cv::OPTFLOW_LK_GET_MIN_EIGENVALS */
#define DR3HIP_LK_DIAG_RESTAGES (1 << 20) /* diagnostic (no OpenCV counterpart, results unchanged): the iteration counts written to d_iters carry, in bits 16..31,
                                           * how often an iterate left the staged search region of its level and the region was fetched again */
#define DR3HIP_WARP_INVERSE_MAP      16   /* == cv::WARP_INVERSE_MAP */
#define DR3HIP_CRIT_COUNT             1   /* == cv::TermCriteria::COUNT */
#define DR3HIP_CRIT_EPS               2   /* == cv::TermCriteria::EPS   */

/* half-sample modes: src/utils.cpp:382-419 has three behaviours */
#define DR3HIP_HALF_TRUNC       0   /* row-aligned (a+b+c+d)/4, the scalar/NEON arithmetic       */
#define DR3HIP_HALF_SSE2        1   /* row-aligned avg(avg(a,c),avg(b,d)), the SSE2 arithmetic   */
#define DR3HIP_HALF_REF_SCALAR  2   /* the scalar pointer walk as written: sheared for odd width */
#define DR3HIP_HALF_REF_X86     3   /* what the reference executes on x86-64: SSE2 if cols%16==0 else REF_SCALAR */

#define DR3HIP_MAX_LEVELS       8   /* pyramid levels 0..7 */
#define DR3HIP_MAX_WIN         64   /* LK window side limit of this build (sides up to 32: the tuned kernels; 33 .. 64: a plain form) */

typedef struct dr3hip_ctx   dr3hip_ctx;
typedef struct dr3hip_batch dr3hip_batch;

/* Parameters of cv::calcOpticalFlowPyrLK as the reference passes them
 * (src/initialization.cpp:593-599, :612-613).  Values are the caller's UN-clamped ones; the
 * callee applies OpenCV's clamping (maxCount to [0,100], epsilon to [0,10] then squared). */
typedef struct dr3hip_lk_params {
    int    win_w, win_h;        /* cv::Size winSize          (reference: 30,30)          */
    int    max_level;           /* maxLevel                  (reference: 4)              */
    int    crit_type;           /* TermCriteria type         (reference: COUNT+EPS = 3)  */
    int    max_iters;           /* TermCriteria maxCount     (reference: 1000)           */
    double eps;                 /* TermCriteria epsilon      (reference: 1e-3)           */
    int    flags;               /* DR3HIP_LK_*               (reference: USE_INITIAL_FLOW) */
    double min_eig_threshold;   /* minEigThreshold           (OpenCV default 1e-4)       */
} dr3hip_lk_params;

/* fills *p with the literals of src/initialization.cpp:593-613 */
void dr3hip_lk_params_reference(dr3hip_lk_params* p);

/* ---- context ------------------------------------------------------------------------------ */
int         dr3hip_version(void);
/* hash of the sources (3dr_amd/csrc, include/) this library was built from; "unstamped" for a plain `make` */
const char* dr3hip_build_info(void);
int         dr3hip_device_count(void);                       /* 0 when no GPU is visible */
int         dr3hip_init(int device, dr3hip_ctx** out);
void        dr3hip_destroy(dr3hip_ctx* ctx);
const char* dr3hip_last_error(const dr3hip_ctx* ctx);        /* ctx may be NULL: last init error */
int         dr3hip_sync(dr3hip_ctx* ctx);                    /* wait for the ctx stream */
void*       dr3hip_stream(dr3hip_ctx* ctx);                  /* the hipStream_t kernels are launched on */
/* Caller-stream form of every "device" call (SURVEY 8b's _async variants): from now on the context enqueues its kernels, copies
 * and collectives on hip_stream (a hipStream_t of the context's device, owned by the caller); NULL returns to the context's own
 * stream.  Work already enqueued is waited for first, so the order of the caller's calls is kept across the switch.  The caller's
 * stream must outlive the next dr3hip_set_stream / dr3hip_destroy; if it was destroyed earlier the switch still takes place (the
 * context does not stay on a dead handle) and the call returns DR3HIP_EHIP. */
int         dr3hip_set_stream(dr3hip_ctx* ctx, void* hip_stream);

/* device memory helpers so that a C/Go/Java caller needs no HIP headers */
int dr3hip_malloc(dr3hip_ctx* ctx, size_t bytes, void** dptr);
int dr3hip_free(dr3hip_ctx* ctx, void* dptr);
int dr3hip_memcpy_h2d(dr3hip_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int dr3hip_memcpy_d2h(dr3hip_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
int dr3hip_memset(dr3hip_ctx* ctx, void* dst_dev, int value, size_t bytes);

/* ---- host-buffer drop-in calls ------------------------------------------------------------ */

/* Replaces cv::calcOpticalFlowPyrLK at src/initialization.cpp:608-613.
 * prev/next: 8-bit gray, w x h, row strides in bytes.  prev_xy: n x (x,y) f32 (cv::Point2f AoS).
 * next_xy: in/out, read as the initial guess when DR3HIP_LK_USE_INITIAL_FLOW.  status: n bytes
 * (1 tracked / 0 lost).  err: n floats.  Returns the effective maxLevel (>=0) or <0. */
int dr3hip_lk_track(dr3hip_ctx* ctx,
                    const uint8_t* prev, const uint8_t* next, int w, int h,
                    int prev_stride, int next_stride,
                    const float* prev_xy, float* next_xy, int n,
                    uint8_t* status, float* err, const dr3hip_lk_params* p);

/* Effective maxLevel cv::buildOpticalFlowPyramid would return (no GPU work). */
int dr3hip_lk_effective_max_level(int w, int h, int win_w, int win_h, int max_level);

/* Replaces utils::create_img_pyramid (src/utils.cpp:421-430, called from src/frame.cpp:18).
 * out_levels[0] is ignored (level 0 is the input, as in the reference); out_levels[1..n-1] are
 * caller-allocated (w>>l) x (h>>l) planes with byte strides out_strides[l]. */
int dr3hip_halfsample_pyramid(dr3hip_ctx* ctx, const uint8_t* img, int w, int h, int stride,
                              int n_levels, int mode,
                              uint8_t* const* out_levels, const int* out_strides);

/* Replaces cv::warpPerspective at src/stitch.cpp:73-74 (8UC3) and src/panorama.cpp:192 (32FC4):
 * INTER_LINEAR, BORDER_CONSTANT 0.  M is the row-major f32 3x3 the reference passes: the forward
 * map src->dst, inverted internally unless DR3HIP_WARP_INVERSE_MAP.  Strides in bytes. */
int dr3hip_warp_perspective(dr3hip_ctx* ctx,
                            const void* src, int sw, int sh, int sstride, int type,
                            void* dst, int dw, int dh, int dstride,
                            const float M[9], int flags);

/* Replaces cv::remap at src/utils.cpp:191 (CV_32FC1 maps, INTER_LINEAR, BORDER_CONSTANT 0).
 * mapx/mapy: dw x dh f32 planes, map_stride in bytes. */
int dr3hip_remap_bilinear(dr3hip_ctx* ctx,
                          const void* src, int sw, int sh, int sstride, int type,
                          void* dst, int dw, int dh, int dstride,
                          const float* mapx, const float* mapy, int map_stride);

/* ---- device-resident batch API (BASELINE configs 2, 4, 5: frame streams / batched pairs) --- */

/* A batch owns `capacity` frame slots of w x h gray frames on the device, each with its LK
 * Gaussian pyramid (levels 0..L, L = effective maxLevel for the given window) and Scharr
 * derivative pyramid.  All kernels take the frame index as a grid axis. */
int  dr3hip_batch_create(dr3hip_ctx* ctx, int w, int h, int win_w, int win_h, int max_level,
                         int capacity, dr3hip_batch** out);
void dr3hip_batch_destroy(dr3hip_batch* b);
int  dr3hip_batch_levels(const dr3hip_batch* b);            /* L + 1 */
size_t dr3hip_batch_bytes(const dr3hip_batch* b);           /* device bytes held */

/* geometry + device pointers of one level's storage (slot s lives at ptr + s*slot_step) */
typedef struct dr3hip_level_info {
    int w, h;
    int img_pitch;    size_t img_slot_step;   void* img;    /* u8  */
    int der_pitch;    size_t der_slot_step;   void* der;    /* interleaved (Ix,Iy) int16, 4 B/px */
} dr3hip_level_info;
int  dr3hip_batch_level_info(const dr3hip_batch* b, int level, dr3hip_level_info* out);

/* copy `count` host frames (each w x h, row stride `stride`, frame-to-frame step `frame_step`
 * bytes) into slots first..first+count-1 (level 0).  Synchronous. */
int  dr3hip_batch_upload(dr3hip_batch* b, int first, int count,
                         const uint8_t* frames, int stride, size_t frame_step);
/* The streaming form, for frames that arrive from the host one batch after the other (the reference receives every frame that way:
 * src/handler.cpp:31-34, tests/test_pipeline.cpp:71-73).  Returns without waiting: the frames cross the link in chunks on the
 * context's upload stream -- through a ring of device bounce buffers -- while the compute stream works on what it was given before;
 * the compute stream picks each chunk up behind an event and writes the slots' level 0 with its border.  The slots are written by
 * the compute stream only, in its order, so an upload enqueued behind a dr3hip_batch_track never overtakes it.  Frames in pinned
 * host memory (dr3hip_host_alloc) are read by the copy engine until dr3hip_batch_upload_wait returns; frames in pageable memory
 * are staged through pinned buffers by the calling thread and may be reused on return. */
int  dr3hip_batch_upload_async(dr3hip_batch* b, int first, int count,
                               const uint8_t* frames, int stride, size_t frame_step);
int  dr3hip_batch_upload_wait(dr3hip_batch* b);
/* Pinned host memory for dr3hip_batch_upload_async / dr3hip_download_async. */
int  dr3hip_host_alloc(dr3hip_ctx* ctx, size_t bytes, void** out);
int  dr3hip_host_free(dr3hip_ctx* ctx, void* p);
/* Results leave on the context's download stream, behind everything the compute stream has been given so far, while the compute
 * stream goes on (host: pinned memory for a copy that really is asynchronous).  dr3hip_download_wait: all of them have arrived.
 * The compute stream does NOT wait for the copy: `dev` must not be written again (by a later dr3hip_batch_track, a memset, an
 * upload ...) until dr3hip_download_wait has returned -- a caller that keeps the compute stream busy alternates two result
 * buffers and waits one step behind, as INTEGRATION.md's streaming loop and bench.py's stream leg do. */
int  dr3hip_download_async(dr3hip_ctx* ctx, void* host, const void* dev, size_t bytes);
int  dr3hip_download_wait(dr3hip_ctx* ctx);
/* Every level of a slot carries OpenCV's winSize border physically (REFLECT_101 pixels; dr3hip_batch_upload writes level 0's with
 * the frame, dr3hip_batch_build those of the levels it produces).  A caller that writes level-0 pixels into a slot by other means
 * (through the device pointers of dr3hip_batch_level_info) calls this before dr3hip_batch_build / _track.  Enqueued on the stream. */
int  dr3hip_batch_pad_level0(dr3hip_batch* b, int first, int count);
/* copy one level of one slot back (tests): what = 0 image (u8, w*h), 1 derivative (int16 w*h*2) */
int  dr3hip_batch_download(dr3hip_batch* b, int slot, int level, int what, void* host, int host_stride);

/* Enqueue pyramid (+ derivative when with_deriv) construction for slots first..first+count-1.
 * This is the "pyramid + gradient pass" graded against the HBM roofline. */
int  dr3hip_batch_build(dr3hip_batch* b, int first, int count, int with_deriv);

/* dr3hip_batch_build with the launches of levels >= 1 (25 % of the pass's bytes in a third of its time: three short, tail-bound
 * launches) on an auxiliary stream of the context, behind the level-0 launch: they run UNDER whatever the context's stream is given
 * next -- typically the tracker of other, already built slots (a ring of two slot groups: build group k + 1 ahead, track group k).
 * Every later batch call that touches these slots (dr3hip_batch_track on pairs that read them, _build, _download, ...) makes the
 * context's stream wait for them first, so results never depend on the overlap; dr3hip_batch_join does so explicitly for all of
 * the batch's pending slots.  dr3hip_sync also waits for the auxiliary stream. */
int  dr3hip_batch_build_ahead(dr3hip_batch* b, int first, int count, int with_deriv);
int  dr3hip_batch_join(dr3hip_batch* b);

/* Enqueue tracking of n_pairs frame pairs.  Pair i tracks slot prev_slot[i] -> next_slot[i]
 * (device int arrays; NULL means prev = first_slot + i, next = first_slot + i + 1).
 * prev_xy / next_xy: device, n_pairs x n_pts x 2 f32; status: n_pairs x n_pts u8; err: f32.
 * iters: optional device int32 n_pairs x n_pts x (L+1) iteration counts (may be NULL).
 * The prev slots must have been built with derivatives. */
int  dr3hip_batch_track(dr3hip_batch* b, int n_pairs, int first_slot,
                        const int* d_prev_slot, const int* d_next_slot,
                        const float* d_prev_xy, float* d_next_xy, int n_pts,
                        uint8_t* d_status, float* d_err, int* d_iters,
                        const dr3hip_lk_params* p);

/* Why a point was lost (status 0), one byte per point; 0 = tracked.  OpenCV reports only the status; these are the four places
 * LKTrackerInvoker clears it. */
#define DR3HIP_LK_LOST_START_OUTSIDE    1   /* the template window at the previous position lies outside level 0 */
#define DR3HIP_LK_LOST_MIN_EIG          2   /* level 0: minimum eigenvalue below the threshold, or determinant < FLT_EPSILON */
#define DR3HIP_LK_LOST_ITERATE_OUTSIDE  3   /* level 0: an iterate's window left the image */
#define DR3HIP_LK_LOST_RESULT_OUTSIDE   4   /* level 0: the final position's window (residual) lies outside the image */

/* dr3hip_batch_track plus d_reason (device, n_pairs x n_pts bytes, may be NULL): the DR3HIP_LK_LOST_* code of every point. */
int  dr3hip_batch_track_diag(dr3hip_batch* b, int n_pairs, int first_slot,
                             const int* d_prev_slot, const int* d_next_slot,
                             const float* d_prev_xy, float* d_next_xy, int n_pts,
                             uint8_t* d_status, float* d_err, int* d_iters, uint8_t* d_reason, const dr3hip_lk_params* p);

/* The tracker's counters as a result struct (what the reference logs around its LK call, src/initialization.cpp:652-654):
 * one reduction over device-resident outputs of dr3hip_batch_track(_diag).  d_reason and d_iters may be NULL (their counters stay
 * 0); d_iters is n_points x levels as dr3hip_batch_track writes it.  iters_hist[k] counts the (point, level) entries that took k
 * iterations, the last bin k >= 31.  Synchronous on return. */
typedef struct dr3hip_lk_counters {
    unsigned long long n_points, n_tracked;
    unsigned long long n_lost_start_outside, n_lost_min_eig, n_lost_iterate_outside, n_lost_result_outside;
    unsigned long long iters_total;
    unsigned long long iters_hist[32];
} dr3hip_lk_counters;
int  dr3hip_lk_stats(dr3hip_ctx* ctx, long long n_points, const uint8_t* d_status, const uint8_t* d_reason,
                     const int* d_iters, int levels, dr3hip_lk_counters* out);

/* ---- device-resident warp (BASELINE config 3: panorama tiles) ----------------------------- */

typedef struct dr3hip_warp_job {
    const void* src; int sw, sh, sstride;     /* device pointers, strides in bytes */
    void*       dst; int dw, dh, dstride;
    float       M[9];                         /* as dr3hip_warp_perspective */
} dr3hip_warp_job;

/* Enqueue n_jobs warps of one pixel type in a single launch (job index is a grid axis). */
int dr3hip_warp_perspective_batch(dr3hip_ctx* ctx, const dr3hip_warp_job* jobs, int n_jobs,
                                  int type, int flags);
/* Which kernel dr3hip_warp_perspective_batch runs a job on, without running it: *kernel = 0 the direct gather kernel (any type, any
 * map), 2 the LDS-staged strip kernel (8UC3; rows 16-byte aligned, nearly affine map: DESIGN.md 4.3); *lds_bytes = the LDS slice of a
 * wave there, *tie_guard = its tie guard (units of 2^-13 of 1/32 px).  Any of the three may be null.  Host only. */
int dr3hip_warp_plan(const dr3hip_warp_job* job, int type, int flags, int* kernel, int* lds_bytes, int* tie_guard);
/* Device-pointer remap (maps on device, strides in bytes). */
int dr3hip_remap_bilinear_dev(dr3hip_ctx* ctx,
                              const void* d_src, int sw, int sh, int sstride, int type,
                              void* d_dst, int dw, int dh, int dstride,
                              const float* d_mapx, const float* d_mapy, int map_stride);
/* Device-pointer half-sampler for one level (src w x h -> dst (w/2) x (h/2)). */
int dr3hip_halfsample_dev(dr3hip_ctx* ctx, const uint8_t* d_src, int w, int h, int sstride,
                          uint8_t* d_dst, int dstride, int mode);

/* ---- FAST-10 + grid-best Shi-Tomasi detector --------------------------------------------- */
/* feature_detection::FastDetector::detect (reference src/features.cpp:43-98): on every level of the
 * frame's half-sample pyramid (utils::create_img_pyramid, src/frame.cpp:13-20) FAST-10 corners at
 * `fast_threshold` (reference: 20) with their bisection scores, 3x3 non-maximum suppression, and per grid
 * cell (cell_size, reference Config::cell_size() = 30) the corner with the best utils::shi_tomasi_score
 * (src/utils.cpp:282-321) above detection_threshold (reference Config::min_harris_corner_score() = 20).
 * These are the points dr3hip_lk_track receives.  The `fast` library is an un-vendored dependency of the
 * reference: its published algorithm is restated, parity unpinned. */
typedef struct dr3hip_corner {
    int x, y;        /* level-0 pixel coordinates (corner position * 2^level), as Corner::x / ::y */
    int level;       /* pyramid level the corner was found on; -1: the cell holds no feature */
    float score;     /* Shi-Tomasi score */
} dr3hip_corner;
/* grid geometry of Detector::Detector (src/features.cpp:7-17) */
int dr3hip_fast_grid_size(int w, int h, int cell_size, int* grid_cols, int* grid_rows);
/* img: host gray frame; occupancy: host, grid_cols*grid_rows bytes or NULL (Detector::flag_features_grid);
 * halfsample_mode: DR3HIP_HALF_* used for the pyramid; cells: grid_cols*grid_rows entries out.
 * Returns the number of features, or < 0. */
int dr3hip_fast_detect(dr3hip_ctx* ctx, const uint8_t* img, int w, int h, int stride, int n_levels,
                       int cell_size, int fast_threshold, double detection_threshold,
                       const uint8_t* occupancy, int halfsample_mode, dr3hip_corner* cells);
/* Device-resident, batched form on the frame slots of a dr3hip_batch (the front end of a stream: pyramid ->
 * detect -> track without a host hop).  For slots first..first+count-1: half-sample pyramid of the slot's frame,
 * FAST + non-max + grid-best selection on every level, then per cell either the corner or "no feature".
 * d_xy (device, count x n_cells x 2 f32, may be NULL): level-0 pixel positions in dr3hip_batch_track's layout; cells
 * without a feature hold DR3HIP_NO_POINT, which the tracker reports with status 0.
 * d_corners (device, count x n_cells dr3hip_corner, may be NULL).  Enqueued on the ctx stream.
 * Returns n_cells (grid_cols * grid_rows) or < 0. */
#define DR3HIP_NO_POINT (-1.0e6f)
int dr3hip_batch_detect(dr3hip_batch* b, int first, int count, int n_levels, int cell_size, int fast_threshold,
                        double detection_threshold, int halfsample_mode, float* d_xy, dr3hip_corner* d_corners);
/* the FAST-10 score map of one image alone (tests, tooling): score_map w*h u8, 0 = no corner */
int dr3hip_fast_score_map(dr3hip_ctx* ctx, const uint8_t* img, int w, int h, int stride, int fast_threshold,
                          uint8_t* score_map);

/* ---- post-LK geometry vote ---------------------------------------------------------------- */
/* InitHelper::CheckFundamental (reference src/initialization.cpp:171-249) for every RANSAC hypothesis of
 * FindFundamental (:80-127) in one launch.  F21: n_hyp x 9 f32 row-major (T2t * Fn * T1 as at :115);
 * pts1 / pts2: the n matched keypoints (kp1.pt, kp2.pt) as x,y f32 pairs; sigma as mSigma.
 * scores: n_hyp f32 out; inliers: n_hyp x n u8 out or NULL.  Host buffers. */
int dr3hip_check_fundamental(dr3hip_ctx* ctx, const float* F21, int n_hyp, const float* pts1, const float* pts2,
                             int n, float sigma, float* scores, uint8_t* inliers);
/* the same on device-resident data, enqueued on the context's stream (dr3hip_sync to wait): d_pts1 / d_pts2 are e.g.
 * dr3hip_batch_track's prev_xy / next_xy of one pair and d_mask its status bytes (NULL = every match counts): a match with mask 0
 * was lost by the tracker; the reference erases it from both vectors before the vote (src/initialization.cpp:621-635), here it is
 * skipped in place, which leaves the sequential f32 score bit-identical; its inlier byte is 0. */
int dr3hip_check_fundamental_dev(dr3hip_ctx* ctx, const float* d_F21, int n_hyp, const float* d_pts1, const float* d_pts2,
                                 const uint8_t* d_mask, int n, float sigma, float* d_scores, uint8_t* d_inliers);
/* InitHelper::Normalize (reference src/initialization.cpp:365-410, called twice from FindFundamental :84-85): the points shifted by
 * their mean and scaled by the inverse mean absolute deviation, and T = [sX 0 -meanX*sX; 0 sY -meanY*sY; 0 0 1] (f32 row-major).
 * Means and deviations are the reference's sequential f32 sums.  xy / out_xy: n (x, y) f32 pairs.  Host buffers. */
int dr3hip_normalize_points(dr3hip_ctx* ctx, const float* xy, int n, float* out_xy, float T[9]);
/* n_sets point sets of n points each (set k at d_xy + 2*n*k floats, its mask at d_mask + n*k bytes, its T at d_T + 9*k) in one
 * launch, device-resident, enqueued on the context's stream.  d_mask as for dr3hip_check_fundamental_dev (NULL = all points);
 * masked-out points come back as (0, 0). */
int dr3hip_normalize_points_dev(dr3hip_ctx* ctx, const float* d_xy, const uint8_t* d_mask, int n, int n_sets,
                                float* d_out_xy, float* d_T);

/* ---- 6-DoF inverse-compositional sparse image alignment ----------------------------------------------------
 * EXTENSION SPECIFIED BY THIS BUILD: the reference only names this step (timer "sparse_img_align", src/handler.cpp:23; empty
 * HandlerMono::process_frame, :80-82), so no reference parity exists; the algorithm (after Forster, Pizzoli, Scaramuzza, "SVO",
 * ICRA 2014, IV-A) is specified in DESIGN.md section 4.8 and restated on the CPU by the test infrastructure, which the kernel equals
 * bit for bit.  For every frame pair: features with known 3-D position in the reference camera frame (xyz, Z <= 0 = no point) and 4x4
 * patches; Gauss-Newton over se(3) from level max_level down to min_level of the batch's pyramid, Jacobians taken once per level
 * at the reference patch; T (12 doubles per pair, rows of [R | t], X_cur = R X_ref + t) is the initial guess and the result. */
typedef struct dr3hip_align_params {
    float fx, fy, cx, cy;      /* level-0 intrinsics (tests/test_pipeline.cpp:62-64) */
    int   max_level, min_level;/* coarsest / finest level used; both must exist in the batch */
    int   n_iter;              /* Gauss-Newton iterations per level (SVO: 10) */
    double eps;                /* stop when the largest component of the update is below eps */
} dr3hip_align_params;
/* All d_* are device pointers; enqueued on the context's stream (dr3hip_sync to wait).  Pair i aligns slot
 * d_cur_slot[i] (or first_slot + i + 1) to slot d_ref_slot[i] (or first_slot + i); d_xyz: n_pairs x n_pts x 3 f32;
 * d_T: n_pairs x 12 f64 in/out; d_n_meas / d_chi2 (n_pairs: patch pixels used and mean squared residual of the last evaluation)
 * and d_iters (n_pairs x DR3HIP_MAX_LEVELS, indexed by level) may be NULL. */
int dr3hip_batch_align(dr3hip_batch* b, int n_pairs, int first_slot, const int* d_ref_slot, const int* d_cur_slot,
                       const float* d_xyz, int n_pts, const dr3hip_align_params* p, double* d_T,
                       int* d_n_meas, float* d_chi2, int* d_iters);

/* ---- spherical / cylindrical pre-warp ------------------------------------------------- */
/* utils::warp_spherical / warp_cylindrical (reference src/utils.cpp:196-201, :267-272): map generation
 * (compute_spherical_warping :125-187, compute_cylindrical_warping :203-265) fused with warp_local's
 * cv::remap (:189-194).  dst has the size of src.  The w + h sin/cos evaluations run on the host with
 * the C library the reference's CPU path uses; everything per-pixel runs in the kernel. */
#define DR3HIP_WARP_SPHERICAL   1
#define DR3HIP_WARP_CYLINDRICAL 2
int dr3hip_warp_local(dr3hip_ctx* ctx, const void* src, int w, int h, int sstride, int type,
                      void* dst, int dstride, int kind, float f);
int dr3hip_warp_local_dev(dr3hip_ctx* ctx, const void* d_src, int w, int h, int sstride, int type,
                          void* d_dst, int dstride, int kind, float f);

/* ---- stitch compose: the caller of the 8UC3 warps, reconstruct::Stitch::process (reference src/stitch.cpp:33-80) ---- */
/* Geometry (:33-71), host-only: Hinv = H.inv() scaled by 1 / Hinv(2,2); bounding box of the right image's corners through
 * Hinv together with the untransformed corners; canvas size = ceil(max) - floor(min); T = translate(-min_X, -min_Y);
 * M_right = T * Hinv.  All f32 matrices row-major.  Returns DR3HIP_EINVAL for a singular H (the reference has a TODO there). */
int dr3hip_stitch_geometry(const float H[9], int w_right, int h_right, float Hinv[9], float T[9], float M_right[9],
                           int* new_w, int* new_h);
/* Compose (:72-80) in one kernel: out = addWeighted(warpPerspective(left, T), alpha, warpPerspective(right, M_right), beta,
 * gamma) on a new_w x new_h 8UC3 canvas (the reference uses alpha = beta = 0.5, gamma = 0).  Host buffers. */
int dr3hip_stitch_compose(dr3hip_ctx* ctx, const uint8_t* left, int lw, int lh, int lstride,
                          const uint8_t* right, int rw, int rh, int rstride,
                          const float T[9], const float M_right[9], int new_w, int new_h,
                          float alpha, float beta, float gamma, uint8_t* out, int ostride);
/* Which form dr3hip_stitch_compose_dev runs, without running it: *kernel = 2 the strip form (the right image through the LDS-staged
 * strip machinery of DESIGN.md 4.3.1, the left image as the exact translation it is, src/stitch.cpp:74), 0 the direct form (any
 * matrices, any alignment).  Pointers and strides as they would be passed (only their alignment matters).  Host only. */
int dr3hip_stitch_plan(const uint8_t* d_left, int lw, int lh, int lstride, const uint8_t* d_right, int rw, int rh, int rstride,
                       const float T[9], const float M_right[9], int new_w, int new_h, float alpha, float beta, float gamma, int* kernel);
/* the same with device pointers (no copies; enqueued on the ctx stream) */
int dr3hip_stitch_compose_dev(dr3hip_ctx* ctx, const uint8_t* d_left, int lw, int lh, int lstride,
                              const uint8_t* d_right, int rw, int rh, int rstride,
                              const float T[9], const float M_right[9], int new_w, int new_h,
                              float alpha, float beta, float gamma, uint8_t* d_out, int ostride);

/* ---- panorama paste: reconstruct::Panorama, reference src/panorama.cpp ------------- */
/* Panorama::set_bbox (src/panorama.cpp:112-138): bounding box of the image corners under the forward
 * f32 homography M, truncated with int().  Host-only. */
int dr3hip_panorama_bbox(const float M[9], int w, int h, int* min_x, int* min_y, int* max_x, int* max_y);
/* The feather vector of add_img_to_canvas (src/panorama.cpp:163-177), max_x - min_x entries, built with
 * the reference's accumulating utils::linspace (include/utils.hpp:37-46).  Returns the entry count, or
 * DR3HIP_EINVAL where the reference itself throws (a box narrower than 2 columns).  Host-only. */
int dr3hip_panorama_ramp(int min_x, int max_x, int feathering_width, float* ramp);
/* Panorama::add_img_to_canvas (src/panorama.cpp:156-212) fused into one kernel: warps the 8UC3 device
 * image d_img (as the (B,G,R,1) 32FC4 image the reference builds) onto the 32FC4 device canvas through the
 * forward homography M (= top_left * H, as at :148), feathers with the ramp and accumulates.  Columns
 * outside [0, cw) are skipped (the reference indexes them unchecked). */
int dr3hip_panorama_add(dr3hip_ctx* ctx, const uint8_t* d_img, int w, int h, int istride,
                        float* d_canvas, int cw, int ch, int cstride,
                        const float M[9], int feathering_width);
/* The same for a GPU that owns only rows [slab_y0, slab_y1) of the cw x ch canvas (canvas tiles are sharded
 * over GPUs, every GPU pastes all contributing images into its own rows, so no reduction is needed): d_canvas points at
 * row slab_y0.  The geometry (bounding box, ramp, block split of the warp) is that of the whole canvas, so the slabs of
 * all GPUs put together equal the single-GPU canvas bit for bit. */
int dr3hip_panorama_add_slab(dr3hip_ctx* ctx, const uint8_t* d_img, int w, int h, int istride,
                             float* d_canvas, int cw, int ch, int cstride,
                             const float M[9], int feathering_width, int slab_y0, int slab_y1);
/* Which form dr3hip_panorama_add(_slab) runs for this image, canvas and matrix, without running it: *kernel = 2 the strip form (the
 * image's box through the LDS-staged strip machinery of DESIGN.md 4.3.1), 0 the direct form (any matrix, any alignment); *coords = the
 * coordinate form over the box (0 OpenCV's own f64 sequence per pixel ... 3 one anchor per strip column); box = the canvas columns and
 * rows the launch evaluates {x0, y0, x1, y1}.  The pointer and stride as they would be passed (only their alignment matters); any of
 * the three outputs may be NULL.  Host only. */
int dr3hip_panorama_plan(const uint8_t* d_img, int w, int h, int istride, int cw, int ch, const float M[9], int slab_y0, int slab_y1,
                         int* kernel, int* coords, int box[4]);
/* Panorama::normalize_canvas (src/panorama.cpp:215-229): 32FC4 weighted canvas -> 8UC3 panorama. */
int dr3hip_panorama_normalize(dr3hip_ctx* ctx, const float* d_canvas, int cw, int ch, int cstride,
                              uint8_t* d_out, int ostride);

/* ---- multi-GPU: one process per GPU, one ctx per process --------------------------- */
/* The path shards by independent frame pairs / canvas tiles, so there is no reduction and no
 * exchange; the only collective is a broadcast from the rank that owns the shared state: the camera
 * intrinsics (reference tests/test_pipeline.cpp:62-64) and, for the one-keyframe-vs-many-candidates
 * pattern, the reference frame's Gaussian pyramid + Scharr derivative pyramid (5 S bytes) built
 * once by the root.  Transport is RCCL over xGMI (librccl is dlopen'ed on first use, so single-GPU
 * users never load it).  Bootstrap: the root calls dr3hip_comm_unique_id, the host side hands the
 * 128 bytes to every rank by any means (the bench uses its torch.distributed store), then every rank
 * calls dr3hip_comm_init with the same id.  Collectives are enqueued on the ctx stream. */
#define DR3HIP_COMM_ID_BYTES 128
int dr3hip_comm_unique_id(dr3hip_ctx* ctx, void* id128);
int dr3hip_comm_init(dr3hip_ctx* ctx, const void* id128, int world, int rank);
int dr3hip_comm_destroy(dr3hip_ctx* ctx);
int dr3hip_comm_world(const dr3hip_ctx* ctx);                 /* 0 when no communicator */
int dr3hip_comm_rank(const dr3hip_ctx* ctx);
/* broadcast `bytes` device bytes from rank `root` (in place on every rank) */
int dr3hip_bcast(dr3hip_ctx* ctx, void* dptr, size_t bytes, int root);
/* broadcast every level of slot `slot` (image plane, and derivative plane when with_deriv) from
 * `root` into the same slot of every rank's batch (same geometry on all ranks): 2 (L+1) broadcasts
 * fused in one RCCL group.  Returns the bytes moved per rank, or < 0. */
long long dr3hip_batch_bcast_slot(dr3hip_batch* b, int slot, int root, int with_deriv);

/* ---- timing ------------------------------------------------------------------------------- */
/* hipEvent pair on the ctx stream (the stream the kernels run on). */
int dr3hip_timer_start(dr3hip_ctx* ctx);
int dr3hip_timer_stop(dr3hip_ctx* ctx, float* ms);           /* synchronises on the stop event */
/* Named accumulating kernel timers: when enabled (two recycled events per bracket), every launch
 * group is bracketed by hipEvents on the ctx stream and the elapsed times accumulate per slot. */
#define DR3HIP_PROF_PYR     0   /* whole pyramid + gradient pass of one dr3hip_batch_build (all levels) */
#define DR3HIP_PROF_LK      1   /* dr3hip_batch_track launch                                           */
#define DR3HIP_PROF_WARP    2   /* warp_perspective launch                                             */
#define DR3HIP_PROF_REMAP   3
#define DR3HIP_PROF_HALF    4
#define DR3HIP_PROF_PYR_L0  5   /* the level-0 fused pyrDown+Scharr kernel alone (dominant HBM kernel) */
#define DR3HIP_PROF_PANO_ADD  6 /* fused panorama paste kernel                                        */
#define DR3HIP_PROF_PANO_NORM 7
#define DR3HIP_PROF_FAST      8 /* FAST score + select kernels of one dr3hip_fast_detect (all levels)   */
#define DR3HIP_PROF_STITCH    9 /* fused stitch compose kernel                                        */
#define DR3HIP_PROF_ALIGN    10 /* sparse image alignment kernel                                     */
#define DR3HIP_PROF_SLOTS   11
int dr3hip_profile_enable(dr3hip_ctx* ctx, int on);
int dr3hip_profile_read(dr3hip_ctx* ctx, int which, double* total_ms, int* n_launch_groups);
int dr3hip_profile_reset(dr3hip_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* DR3HIP_H */
