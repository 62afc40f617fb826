// dr3hip_api.cpp -- host side of libdr3hip.so: context, device memory, staging of caller-owned
// host buffers, and the launch sequences behind each C-ABI entry point of include/dr3hip.h.
// Compiled by hipcc as host code (-ffp-contract=off: the homography inversion below must round
// like cv::invert does on the reference's x86-64 build).
#include "dr3hip_internal.h"
#include "vote.h"
#include "build_stamp.h"
#include <algorithm>
#include <climits>
#include <cmath>
#include <dlfcn.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

using namespace dr3;

// -------------------------------------------------------------------------------------------------
struct ProfEvent { int which; hipEvent_t e0, e1; bool own0, own1; };     // own = false: the event belongs to another pending entry

struct dr3hip_ctx {
    int device = -1;
    hipStream_t stream = nullptr;          // the stream every launch and copy goes to: own_stream, or the caller's (dr3hip_set_stream)
    hipStream_t own_stream = nullptr;
    std::string err;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    // grow-only staging buffers for the host-buffer calls
    struct Buf { void* p = nullptr; size_t cap = 0; };
    Buf stage[9];          // [7] is the bounce buffer of copy2d_h2d / copy2d_d2h, [8] the sparse alignment's patch cache
    dr3hip_batch* lk_batch = nullptr;     // cached 2-slot batch behind dr3hip_lk_track
    void* pin_io = nullptr; size_t pin_io_cap = 0;      // pinned host buffer of dr3hip_lk_track: points in, results out, one copy each way
    bool prof = false;
    double prof_ms[DR3HIP_PROF_SLOTS] = {0};
    int prof_n[DR3HIP_PROF_SLOTS] = {0};
    std::vector<ProfEvent> pending;
    std::vector<hipEvent_t> ev_pool;      // recycled events: no hipEventCreate in the steady state
    // dr3hip_batch_build_ahead: the launches of levels >= 1 of a pass run on an auxiliary stream, under whatever the context's stream
    // does next (the tracker of OTHER slots); a consumer of those slots on the context's stream waits for their event first
    struct Ahead { dr3hip_batch* b; int first, count; hipEvent_t done; };
    hipStream_t aux = nullptr;
    hipEvent_t aux_fork = nullptr;
    std::vector<Ahead> ahead;
    void* comm = nullptr;                 // ncclComm_t (RCCL), see "multi-GPU" below
    int comm_world = 0, comm_rank = -1;
    // streaming transfers (dr3hip_batch_upload_async / dr3hip_download_async): a copy stream per direction and a ring of chunk
    // buffers -- a device bounce buffer (the frames as the host holds them; the compute stream changes the pitch out of it) and a
    // pinned host staging buffer (only used for pageable sources) per ring slot
    struct Xfer {
        static constexpr int K = 24;       // x 8 MiB chunks: a whole batch of frames can be in flight ahead of the compute stream
        hipStream_t up = nullptr, down = nullptr;
        void* dev[K] = {nullptr}; void* pin[K] = {nullptr}; size_t cap = 0;
        hipEvent_t copied[K] = {nullptr}, consumed[K] = {nullptr}; bool used[K] = {false};
        hipEvent_t computed = nullptr;    // marks the point of the compute stream a download waits for
        int next = 0;
        bool ready = false;               // every stream and event above exists
    } xf;
};

struct dr3hip_batch {
    dr3hip_ctx* ctx = nullptr;
    int w = 0, h = 0, win_w = 0, win_h = 0, levels = 0, capacity = 0;
    LevelDev lv[DR3HIP_MAX_LEVELS];
    void* img_base[DR3HIP_MAX_LEVELS] = {nullptr};   // allocation starts (lv[].img / .der point at the interior origin)
    void* der_base[DR3HIP_MAX_LEVELS] = {nullptr};
    size_t bytes = 0;
    // detector storage (dr3hip_batch_detect), allocated on first use: half-sample pyramid levels 1.. of every
    // slot (continuous rows like the reference's cv::Mat levels) and one cell-key array per slot
    struct Det { int n_levels = 0, n_cells = 0; uint8_t* hs[8] = {nullptr}; long long hs_step[8] = {0};
                 unsigned long long* keys = nullptr; } det;
};

static std::string g_init_err;

static int fail(dr3hip_ctx* ctx, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    if (ctx) ctx->err = buf; else g_init_err = buf;
    return code;
}

#define HIPCHK(ctx, call)                                                                        \
    do { hipError_t e_ = (call);                                                                   \
         if (e_ != hipSuccess) return fail((ctx), DR3HIP_EHIP, "%s: %s (%s:%d)", #call,          \
                                           hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

struct DeviceGuard {          // every entry point runs with the ctx's device current
    int prev = -1;
    explicit DeviceGuard(int dev) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; if (prev != dev) (void)hipSetDevice(dev); else prev = -1; }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

static int stage_reserve(dr3hip_ctx* ctx, int idx, size_t bytes)
{
    dr3hip_ctx::Buf& b = ctx->stage[idx];
    if (b.cap >= bytes) return DR3HIP_OK;
    if (b.p) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }
    const size_t cap = bytes + bytes / 4 + 4096;
    if (hipMalloc(&b.p, cap) != hipSuccess) { b.p = nullptr; return fail(ctx, DR3HIP_ENOMEM, "hipMalloc(%zu) failed", cap); }
    b.cap = cap;
    return DR3HIP_OK;
}

// Host <-> pitched-device 2-D copies.  hipMemcpy2DAsync between PAGEABLE host rows and a pitched device plane runs row
// by row (measured: 2.76 ms for one 1241 x 376 frame against 0.04 ms for the same bytes as one linear copy), so host
// images cross PCIe as ONE linear copy into / out of a device bounce buffer and the pitch change is a device-side 2-D
// copy.  Both calls are ordered on the ctx stream; pageable linear copies return when the host side is done with.
static int copy2d_h2d(dr3hip_ctx* ctx, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height)
{
    if (width == 0 || height == 0) return DR3HIP_OK;
    if (dpitch == spitch) {                                            // same layout: one linear copy does it
        HIPCHK(ctx, hipMemcpyAsync(dst, src, spitch * (height - 1) + width, hipMemcpyHostToDevice, ctx->stream));
        return DR3HIP_OK;
    }
    const size_t bytes = spitch * (height - 1) + width;
    int rc;
    if ((rc = stage_reserve(ctx, 7, bytes))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->stage[7].p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpy2DAsync(dst, dpitch, ctx->stage[7].p, spitch, width, height, hipMemcpyDeviceToDevice, ctx->stream));
    return DR3HIP_OK;
}

static int copy2d_d2h(dr3hip_ctx* ctx, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height)
{
    if (width == 0 || height == 0) return DR3HIP_OK;
    if (dpitch != width) {                                             // gaps between the host rows must stay untouched
        HIPCHK(ctx, hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, hipMemcpyDeviceToHost, ctx->stream));
        return DR3HIP_OK;
    }
    int rc;
    if ((rc = stage_reserve(ctx, 7, width * height))) return rc;
    HIPCHK(ctx, hipMemcpy2DAsync(ctx->stage[7].p, width, src, spitch, width, height, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(dst, ctx->stage[7].p, width * height, hipMemcpyDeviceToHost, ctx->stream));
    return DR3HIP_OK;
}

// ---- profiling brackets ----------------------------------------------------------------------------
static void prof_begin(dr3hip_ctx* ctx, int which, ProfEvent& pe)
{
    pe.which = -1;
    if (!ctx->prof || which < 0) return;
    hipEvent_t* ev[2] = {&pe.e0, &pe.e1};
    for (int i = 0; i < 2; ++i) {
        if (!ctx->ev_pool.empty()) { *ev[i] = ctx->ev_pool.back(); ctx->ev_pool.pop_back(); }
        else if (hipEventCreate(ev[i]) != hipSuccess) { if (i) ctx->ev_pool.push_back(pe.e0); return; }
    }
    pe.which = which; pe.own0 = pe.own1 = true;
    (void)hipEventRecord(pe.e0, ctx->stream);
}
// The hot slots (pyramid pass, tracker) do not put event packets between their kernels: their events ride on the kernel
// dispatches themselves (hipExtLaunchKernelGGL start / stop events = the dispatch's own timestamps).  prof_open only takes the
// events; the launches record them; prof_close queues the pair for prof_drain.  share0: use this event as e0 (the begin of an
// enclosing slot whose first kernel is also this slot's).
static void prof_open(dr3hip_ctx* ctx, int which, ProfEvent& pe, hipEvent_t share0 = nullptr)
{
    pe.which = -1; pe.e0 = pe.e1 = nullptr; pe.own0 = share0 == nullptr; pe.own1 = true;
    if (!ctx->prof || which < 0) return;
    hipEvent_t* ev[2] = {&pe.e0, &pe.e1};
    for (int i = share0 ? 1 : 0; i < 2; ++i) {
        if (!ctx->ev_pool.empty()) { *ev[i] = ctx->ev_pool.back(); ctx->ev_pool.pop_back(); }
        else if (hipEventCreate(ev[i]) != hipSuccess) { if (i && !share0) ctx->ev_pool.push_back(pe.e0); pe.e0 = pe.e1 = nullptr; return; }
    }
    if (share0) pe.e0 = share0;
    pe.which = which;
}
static void prof_close(dr3hip_ctx* ctx, ProfEvent& pe)
{
    if (pe.which >= 0) ctx->pending.push_back(pe);
}
static void prof_end(dr3hip_ctx* ctx, ProfEvent& pe)
{
    if (pe.which < 0) return;
    (void)hipEventRecord(pe.e1, ctx->stream);
    ctx->pending.push_back(pe);
}
static void prof_drain(dr3hip_ctx* ctx)
{
    for (ProfEvent& pe : ctx->pending) {
        float ms = 0.f;
        if (hipEventSynchronize(pe.e1) == hipSuccess && hipEventElapsedTime(&ms, pe.e0, pe.e1) == hipSuccess) {
            ctx->prof_ms[pe.which] += ms; ctx->prof_n[pe.which] += 1;
        }
        if (pe.own0) ctx->ev_pool.push_back(pe.e0);
        if (pe.own1) ctx->ev_pool.push_back(pe.e1);
    }
    ctx->pending.clear();
}

// ---- dr3hip_batch_build_ahead's bookkeeping ----------------------------------------------------------------------------
// Makes the context's stream wait for the auxiliary work on slots [first, first + count) of batch b (count < 0: every pending entry
// of b; b null: everything).  An entry the stream has been made to wait for is retired (its event goes back to the pool).
static int ahead_join(dr3hip_ctx* ctx, const dr3hip_batch* b, int first, int count)
{
    for (size_t i = 0; i < ctx->ahead.size();) {
        dr3hip_ctx::Ahead& e = ctx->ahead[i];
        const bool hit = (!b || e.b == b) && (count < 0 || (first < e.first + e.count && e.first < first + count));
        if (!hit) { ++i; continue; }
        HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, e.done, 0));
        ctx->ev_pool.push_back(e.done);
        ctx->ahead.erase(ctx->ahead.begin() + (long)i);
    }
    return DR3HIP_OK;
}
static void ahead_release(dr3hip_ctx* ctx)
{
    if (ctx->aux) (void)hipStreamSynchronize(ctx->aux);
    for (auto& e : ctx->ahead) if (e.done) (void)hipEventDestroy(e.done);
    ctx->ahead.clear();
    if (ctx->aux_fork) (void)hipEventDestroy(ctx->aux_fork);
    if (ctx->aux) (void)hipStreamDestroy(ctx->aux);
    ctx->aux = nullptr; ctx->aux_fork = nullptr;
}

// =================================================================================================
extern "C" {

int dr3hip_version(void) { return DR3HIP_VERSION; }
const char* dr3hip_build_info(void) { return DR3HIP_SOURCE_HASH; }

void dr3hip_lk_params_reference(dr3hip_lk_params* p)
{
    if (!p) return;
    p->win_w = 30; p->win_h = 30;          // klt_win_size, src/initialization.cpp:593
    p->max_level = 4;                       // :613
    p->crit_type = DR3HIP_CRIT_COUNT | DR3HIP_CRIT_EPS;   // :598
    p->max_iters = 1000;                    // klt_max_iter :594
    p->eps = 1e-3;                          // klt_eps :595
    p->flags = DR3HIP_LK_USE_INITIAL_FLOW;  // :613
    p->min_eig_threshold = 1e-4;            // OpenCV default argument
}

int dr3hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int dr3hip_init(int device, dr3hip_ctx** out)
{
    if (!out) return fail(nullptr, DR3HIP_EINVAL, "dr3hip_init: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, DR3HIP_ENODEV, "dr3hip_init: no HIP device (%s); libdr3hip has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (device < 0 || device >= n) return fail(nullptr, DR3HIP_EINVAL, "dr3hip_init: device %d out of range [0,%d)", device, n);
    dr3hip_ctx* ctx = new (std::nothrow) dr3hip_ctx();
    if (!ctx) return fail(nullptr, DR3HIP_ENOMEM, "dr3hip_init: out of host memory");
    ctx->device = device;
    if ((e = hipSetDevice(device)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreate(&ctx->t0)) != hipSuccess || (e = hipEventCreate(&ctx->t1)) != hipSuccess) {
        fail(nullptr, DR3HIP_EHIP, "dr3hip_init: %s", hipGetErrorString(e));
        delete ctx;
        return DR3HIP_EHIP;
    }
    ctx->own_stream = ctx->stream;
    *out = ctx;
    return DR3HIP_OK;
}

static void xfer_release(dr3hip_ctx* ctx)
{
    dr3hip_ctx::Xfer& x = ctx->xf;
    if (x.up) (void)hipStreamSynchronize(x.up);
    if (x.down) (void)hipStreamSynchronize(x.down);
    for (int i = 0; i < dr3hip_ctx::Xfer::K; ++i) {
        if (x.dev[i]) (void)hipFree(x.dev[i]);
        if (x.pin[i]) (void)hipHostFree(x.pin[i]);
        if (x.copied[i]) (void)hipEventDestroy(x.copied[i]);
        if (x.consumed[i]) (void)hipEventDestroy(x.consumed[i]);
        x.dev[i] = x.pin[i] = nullptr; x.copied[i] = x.consumed[i] = nullptr; x.used[i] = false;
    }
    if (x.computed) (void)hipEventDestroy(x.computed);
    if (x.up) (void)hipStreamDestroy(x.up);
    if (x.down) (void)hipStreamDestroy(x.down);
    x.computed = nullptr; x.up = x.down = nullptr; x.cap = 0; x.ready = false;
}

void dr3hip_destroy(dr3hip_ctx* ctx)
{
    if (!ctx) return;
    DeviceGuard g(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ahead_release(ctx);
    prof_drain(ctx);
    for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
    if (ctx->lk_batch) dr3hip_batch_destroy(ctx->lk_batch);
    (void)dr3hip_comm_destroy(ctx);
    for (auto& b : ctx->stage) if (b.p) (void)hipFree(b.p);
    if (ctx->pin_io) (void)hipHostFree(ctx->pin_io);
    xfer_release(ctx);
    if (ctx->t0) (void)hipEventDestroy(ctx->t0);
    if (ctx->t1) (void)hipEventDestroy(ctx->t1);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

const char* dr3hip_last_error(const dr3hip_ctx* ctx) { return ctx ? ctx->err.c_str() : g_init_err.c_str(); }

int dr3hip_sync(dr3hip_ctx* ctx)
{
    if (!ctx) return DR3HIP_EINVAL;
    DeviceGuard g(ctx->device);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->aux) HIPCHK(ctx, hipStreamSynchronize(ctx->aux));          // work built ahead is part of what the caller waits for
    return DR3HIP_OK;
}

void* dr3hip_stream(dr3hip_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int dr3hip_set_stream(dr3hip_ctx* ctx, void* hip_stream)
{
    if (!ctx) return DR3HIP_EINVAL;
    DeviceGuard g(ctx->device);
    // work already enqueued keeps its order: the old stream is drained (and its timing events read) before the switch.  A caller's
    // stream must outlive the switch away from it; if it did not (the drain fails), the switch still happens -- a context must not
    // stay stuck on a dead handle -- and the call reports the failure.
    if (ctx->aux) (void)hipStreamSynchronize(ctx->aux);               // work built ahead belongs to the stream being left
    for (auto& a : ctx->ahead) ctx->ev_pool.push_back(a.done);
    ctx->ahead.clear();
    const hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) (void)hipGetLastError();
    else prof_drain(ctx);
    if (e != hipSuccess && ctx->stream == ctx->own_stream) return fail(ctx, DR3HIP_EHIP, "set_stream: draining the context's stream failed: %s", hipGetErrorString(e));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    if (e != hipSuccess) {
        // the events queued on the dead stream are never read: destroy the ones this context owns rather than dropping the handles
        for (ProfEvent& pe : ctx->pending) {
            if (pe.own0 && pe.e0) (void)hipEventDestroy(pe.e0);
            if (pe.own1 && pe.e1) (void)hipEventDestroy(pe.e1);
        }
        ctx->pending.clear();
        (void)hipGetLastError();
        return fail(ctx, DR3HIP_EHIP, "set_stream: the previous stream could not be drained (%s); switched anyway", hipGetErrorString(e));
    }
    return DR3HIP_OK;
}

int dr3hip_malloc(dr3hip_ctx* ctx, size_t bytes, void** dptr)
{
    if (!ctx || !dptr) return DR3HIP_EINVAL;
    DeviceGuard g(ctx->device);
    *dptr = nullptr;
    if (hipMalloc(dptr, bytes ? bytes : 1) != hipSuccess) { *dptr = nullptr; return fail(ctx, DR3HIP_ENOMEM, "hipMalloc(%zu) failed", bytes); }
    return DR3HIP_OK;
}
int dr3hip_free(dr3hip_ctx* ctx, void* dptr)
{
    if (!ctx) return DR3HIP_EINVAL;
    if (!dptr) return DR3HIP_OK;
    DeviceGuard g(ctx->device);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipFree(dptr));
    return DR3HIP_OK;
}
int dr3hip_memcpy_h2d(dr3hip_ctx* ctx, void* dst, const void* src, size_t bytes)
{
    if (!ctx || (bytes && (!dst || !src))) return DR3HIP_EINVAL;
    DeviceGuard g(ctx->device);
    HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return DR3HIP_OK;
}
int dr3hip_memcpy_d2h(dr3hip_ctx* ctx, void* dst, const void* src, size_t bytes)
{
    if (!ctx || (bytes && (!dst || !src))) return DR3HIP_EINVAL;
    DeviceGuard g(ctx->device);
    HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return DR3HIP_OK;
}
int dr3hip_memset(dr3hip_ctx* ctx, void* dst, int value, size_t bytes)
{
    if (!ctx || (bytes && !dst)) return DR3HIP_EINVAL;
    DeviceGuard g(ctx->device);
    HIPCHK(ctx, hipMemsetAsync(dst, value, bytes, ctx->stream));
    return DR3HIP_OK;
}

// ---- FAST-10 + grid-best Shi-Tomasi -----------------------------------------------------------
int dr3hip_fast_grid_size(int w, int h, int cell_size, int* grid_cols, int* grid_rows)
{
    if (w < 1 || h < 1 || cell_size < 1 || !grid_cols || !grid_rows) return DR3HIP_EINVAL;
    *grid_cols = (int)std::ceil((double)w / cell_size);               // src/features.cpp:14-15
    *grid_rows = (int)std::ceil((double)h / cell_size);
    return DR3HIP_OK;
}

int dr3hip_fast_score_map(dr3hip_ctx* ctx, const uint8_t* img, int w, int h, int stride, int fast_threshold, uint8_t* score_map)
{
    if (!ctx) return DR3HIP_EINVAL;
    if (!img || !score_map || w < 1 || h < 1 || stride < w || fast_threshold < 1 || fast_threshold > 254)
        return fail(ctx, DR3HIP_EINVAL, "fast_score_map: bad arguments");
    DeviceGuard g(ctx->device);
    const int ip = align_up(w, 64);
    int rc;
    if ((rc = stage_reserve(ctx, 4, (size_t)ip * h + 256)) || (rc = stage_reserve(ctx, 1, (size_t)ip * h + 256))) return rc;
    { const int rc2_ = copy2d_h2d(ctx, ctx->stage[4].p, ip, img, stride, w, h); if (rc2_) return rc2_; }
    FastArgs a;
    a.img = (const uint8_t*)ctx->stage[4].p; a.w = w; a.h = h; a.stride = ip;
    a.score = (uint8_t*)ctx->stage[1].p; a.spitch = ip; a.threshold = fast_threshold;
    a.level = 0; a.cell_size = 1; a.grid_cols = 1; a.det_threshold = 0.f; a.occupancy = nullptr; a.cells = nullptr;
    a.n_slots = 1; a.img_step = 0; a.score_step = 0; a.n_cells = 0;
    HIPCHK(ctx, launch_fast_level(a, ctx->stream));
    { const int rc2_ = copy2d_d2h(ctx, score_map, w, a.score, ip, w, h); if (rc2_) return rc2_; }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return DR3HIP_OK;
}

int dr3hip_fast_detect(dr3hip_ctx* ctx, const uint8_t* img, int w, int h, int stride, int n_levels,
                       int cell_size, int fast_threshold, double detection_threshold,
                       const uint8_t* occupancy, int halfsample_mode, dr3hip_corner* cells)
{
    if (!ctx) return DR3HIP_EINVAL;
    if (!img || !cells || w < 8 || h < 8 || stride < w || n_levels < 1 || n_levels > 8 || cell_size < 1 ||
        fast_threshold < 1 || fast_threshold > 254 || w >= 16384 || h >= 16384 || (w >> (n_levels - 1)) < 1 || (h >> (n_levels - 1)) < 1)
        return fail(ctx, DR3HIP_EINVAL, "fast_detect: bad arguments");
    int gc, gr;
    dr3hip_fast_grid_size(w, h, cell_size, &gc, &gr);
    const int n_cells = gc * gr;
    DeviceGuard g(ctx->device);
    // device layout in stage[4]: the half-sample pyramid exactly as dr3hip_halfsample_pyramid lays it out
    // (level 0 keeps the caller's stride, deeper levels are continuous like the reference's cv::Mat levels)
    int rc;
    std::vector<int> lw(n_levels), lh(n_levels), lp(n_levels);
    std::vector<size_t> off(n_levels);
    size_t total = 0;
    for (int l = 0; l < n_levels; ++l) {
        lw[l] = l ? lw[l - 1] / 2 : w; lh[l] = l ? lh[l - 1] / 2 : h;
        lp[l] = l ? lw[l] : stride;
        off[l] = total;
        total += (size_t)align_up((long long)lp[l] * lh[l] + 64, 256ll);
    }
    const size_t cells_off = (size_t)align_up(n_cells, 256);          // stage[6]: occupancy bytes, then the cell keys
    if ((rc = stage_reserve(ctx, 4, total)) ||
        (rc = stage_reserve(ctx, 6, cells_off + sizeof(unsigned long long) * (size_t)n_cells))) return rc;
    uint8_t* base = (uint8_t*)ctx->stage[4].p;
    uint8_t* d_occ = (uint8_t*)ctx->stage[6].p;
    unsigned long long* d_cells = (unsigned long long*)(d_occ + cells_off);
    // every cell starts as Corner(0, 0, detection_threshold, 0) (src/features.cpp:49) and only a strictly greater score
    // replaces it: the kernel pushes keys for scores > threshold only, so "no key yet" can be the all-zero pattern
    const float thr_f = (float)detection_threshold;
    if (!(thr_f >= 0.f)) return fail(ctx, DR3HIP_EINVAL, "fast_detect: detection_threshold must be >= 0");
    HIPCHK(ctx, hipMemsetAsync(d_cells, 0, sizeof(unsigned long long) * (size_t)n_cells, ctx->stream));
    if (occupancy) HIPCHK(ctx, hipMemcpyAsync(d_occ, occupancy, (size_t)n_cells, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(base + off[0], img, (size_t)stride * (h - 1) + w, hipMemcpyHostToDevice, ctx->stream));
    for (int l = 1; l < n_levels; ++l)
        if ((rc = dr3hip_halfsample_dev(ctx, base + off[l - 1], lw[l - 1], lh[l - 1], lp[l - 1], base + off[l], lp[l], halfsample_mode))) return rc;
    ProfEvent pe; prof_begin(ctx, DR3HIP_PROF_FAST, pe);
    for (int l = 0; l < n_levels; ++l) {
        FastArgs a;
        a.img = base + off[l]; a.w = lw[l]; a.h = lh[l]; a.stride = lp[l];
        a.score = nullptr; a.spitch = 0; a.threshold = fast_threshold;                  // no score map: the kernel keeps the scores in LDS
        a.level = l; a.cell_size = cell_size; a.grid_cols = gc; a.det_threshold = thr_f;
        a.occupancy = occupancy ? d_occ : nullptr; a.cells = d_cells;
        a.n_slots = 1; a.img_step = 0; a.score_step = 0; a.n_cells = n_cells;
        HIPCHK(ctx, launch_fast_level(a, ctx->stream));
    }
    prof_end(ctx, pe);
    std::vector<unsigned long long> keys((size_t)n_cells);
    HIPCHK(ctx, hipMemcpyAsync(keys.data(), d_cells, sizeof(unsigned long long) * (size_t)n_cells, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    int n_feat = 0;
    for (int k = 0; k < n_cells; ++k) {
        dr3hip_corner& c = cells[k];
        const uint32_t bits = (uint32_t)(keys[k] >> 32), order = ~(uint32_t)keys[k];
        memcpy(&c.score, &bits, 4);
        if (keys[k] == 0ull || !((double)c.score > detection_threshold)) {             // :88 corner.score > detection_threshold
            c.x = 0; c.y = 0; c.level = -1; c.score = thr_f;
            continue;
        }
        c.level = (int)(order >> 28);
        c.x = (int)(order & 0x3fffu) << c.level;                                       // Corner(xy.x * scale, xy.y * scale, ..)
        c.y = (int)((order >> 14) & 0x3fffu) << c.level;
        ++n_feat;
    }
    return n_feat;
}

static int resolve_half_mode(int mode, int w);        // defined with the half-sampler entry points below

// Device-resident, batched detector on frame slots: the front end of a stream (Frame::Frame -> create_img_pyramid ->
// FastDetector::detect, reference src/frame.cpp:13-20, :57) without a host hop; its points feed dr3hip_batch_track.
int dr3hip_batch_detect(dr3hip_batch* b, int first, int count, int n_levels, int cell_size, int fast_threshold,
                        double detection_threshold, int halfsample_mode, float* d_xy, dr3hip_corner* d_corners)
{
    if (!b) return DR3HIP_EINVAL;
    dr3hip_ctx* ctx = b->ctx;
    const int w = b->w, h = b->h;
    if (first < 0 || count < 1 || first + count > b->capacity || n_levels < 1 || n_levels > 8 || cell_size < 1 ||
        fast_threshold < 1 || fast_threshold > 254 || w >= 16384 || h >= 16384 || (w >> (n_levels - 1)) < 1 || (h >> (n_levels - 1)) < 1 ||
        (!d_xy && !d_corners) || halfsample_mode < 0 || halfsample_mode > 3)
        return fail(ctx, DR3HIP_EINVAL, "batch_detect: bad arguments");
    const float thr_f = (float)detection_threshold;
    if (!(thr_f >= 0.f)) return fail(ctx, DR3HIP_EINVAL, "batch_detect: detection_threshold must be >= 0");
    int gc, gr;
    dr3hip_fast_grid_size(w, h, cell_size, &gc, &gr);
    const int n_cells = gc * gr;
    DeviceGuard g(ctx->device);
    dr3hip_batch::Det& D = b->det;
    if (D.n_levels < n_levels || D.n_cells < n_cells) {               // (re)allocate the detector storage
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        for (int l = 0; l < 8; ++l) if (D.hs[l]) { (void)hipFree(D.hs[l]); D.hs[l] = nullptr; }
        if (D.keys) { (void)hipFree(D.keys); D.keys = nullptr; }
        D.n_levels = 0; D.n_cells = 0;
        int lw = w, lh = h;
        for (int l = 1; l < n_levels; ++l) {
            lw /= 2; lh /= 2;
            D.hs_step[l] = align_up((long long)lw * lh + 64, 256ll);
            void* p = nullptr;
            if (hipMalloc(&p, (size_t)D.hs_step[l] * b->capacity + 256) != hipSuccess) return fail(ctx, DR3HIP_ENOMEM, "batch_detect: hipMalloc (level %d)", l);
            D.hs[l] = (uint8_t*)p; b->bytes += (size_t)D.hs_step[l] * b->capacity + 256;
        }
        void* pk = nullptr;
        if (hipMalloc(&pk, sizeof(unsigned long long) * (size_t)n_cells * b->capacity) != hipSuccess)
            return fail(ctx, DR3HIP_ENOMEM, "batch_detect: hipMalloc (cells)");
        D.keys = (unsigned long long*)pk;
        b->bytes += sizeof(unsigned long long) * (size_t)n_cells * b->capacity;
        D.n_levels = n_levels; D.n_cells = n_cells;
    }
    // cell keys: all-zero = still the initial Corner(0, 0, detection_threshold, 0) (src/features.cpp:49); the select
    // kernel pushes a key only for a score strictly above the threshold
    unsigned long long* keys = D.keys + (size_t)first * n_cells;
    HIPCHK(ctx, hipMemsetAsync(keys, 0, sizeof(unsigned long long) * (size_t)n_cells * count, ctx->stream));
    ProfEvent pe; prof_begin(ctx, DR3HIP_PROF_FAST, pe);
    // half-sample pyramid of every slot: level 0 is the slot's frame inside its pitched plane; the reference holds it as
    // a continuous cv::Mat (step == cols), which is what REF_SCALAR's flat walk must see (lstride)
    int lw = w, lh = h;
    for (int l = 1; l < n_levels; ++l) {
        HalfArgs ha;
        ha.src = (l == 1) ? b->lv[0].img + (size_t)b->lv[0].step * first : D.hs[l - 1] + (size_t)D.hs_step[l - 1] * first;
        ha.w = lw; ha.h = lh; ha.sstride = (l == 1) ? b->lv[0].pitch : lw; ha.lstride = lw;
        ha.dst = D.hs[l] + (size_t)D.hs_step[l] * first; ha.ow = lw / 2; ha.oh = lh / 2; ha.dstride = lw / 2;
        ha.mode = resolve_half_mode(halfsample_mode, lw);
        ha.n_slots = count; ha.sstep = (l == 1) ? b->lv[0].step : D.hs_step[l - 1]; ha.dstep = D.hs_step[l];
        HIPCHK(ctx, launch_halfsample(ha, ctx->stream));
        lw /= 2; lh /= 2;
    }
    lw = w; lh = h;
    for (int l = 0; l < n_levels; ++l) {
        FastArgs a;
        a.img = (l == 0) ? b->lv[0].img + (size_t)b->lv[0].step * first : D.hs[l] + (size_t)D.hs_step[l] * first;
        a.w = lw; a.h = lh; a.stride = (l == 0) ? b->lv[0].pitch : lw;
        a.img_step = (l == 0) ? b->lv[0].step : D.hs_step[l];
        a.score = nullptr; a.spitch = 0; a.score_step = 0;                                 // no score map: the kernel keeps the scores in LDS
        a.threshold = fast_threshold; a.level = l; a.cell_size = cell_size; a.grid_cols = gc; a.det_threshold = thr_f;
        a.occupancy = nullptr; a.cells = keys; a.n_slots = count; a.n_cells = n_cells;
        HIPCHK(ctx, launch_fast_level(a, ctx->stream));
        lw /= 2; lh /= 2;
    }
    CellsArgs ca;
    ca.cells = keys; ca.n_total = n_cells * count; ca.empty_key = 0ull; ca.det_threshold = thr_f;
    ca.xy = d_xy; ca.corners = d_corners;
    HIPCHK(ctx, launch_cells_decode(ca, ctx->stream));
    prof_end(ctx, pe);
    return n_cells;
}

// ---- CheckFundamental over all hypotheses ----------------------------------------------------
int dr3hip_check_fundamental(dr3hip_ctx* ctx, const float* F21, int n_hyp, const float* pts1, const float* pts2,
                             int n, float sigma, float* scores, uint8_t* inliers)
{
    if (!ctx) return DR3HIP_EINVAL;
    if (!F21 || !pts1 || !pts2 || !scores || n_hyp < 1 || n < 0 || n_hyp > (1 << 20) || n > (1 << 24))
        return fail(ctx, DR3HIP_EINVAL, "check_fundamental: bad arguments");
    DeviceGuard g(ctx->device);
    const size_t fb = sizeof(float) * 9 * (size_t)n_hyp, pb = sizeof(float) * 2 * (size_t)(n ? n : 1);
    const size_t sb = sizeof(float) * (size_t)n_hyp, ib = inliers ? (size_t)n_hyp * (size_t)(n ? n : 1) : 0;
    int rc;
    if ((rc = stage_reserve(ctx, 0, fb + 256)) || (rc = stage_reserve(ctx, 2, 2 * pb + 256)) || (rc = stage_reserve(ctx, 3, sb + 256)) ||
        (ib && (rc = stage_reserve(ctx, 1, ib + 256)))) return rc;
    uint8_t* dp = (uint8_t*)ctx->stage[2].p;
    HIPCHK(ctx, hipMemcpyAsync(ctx->stage[0].p, F21, fb, hipMemcpyHostToDevice, ctx->stream));
    if (n) {
        HIPCHK(ctx, hipMemcpyAsync(dp, pts1, pb, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(dp + pb, pts2, pb, hipMemcpyHostToDevice, ctx->stream));
    }
    CheckFArgs a;
    a.mask = nullptr;
    a.F = (const float*)ctx->stage[0].p; a.n_hyp = n_hyp;
    a.pts1 = (const float*)dp; a.pts2 = (const float*)(dp + pb); a.n = n;
    a.inv_sigma_square = (float)(1.0 / (double)(sigma * sigma));          // const float invSigmaSquare = 1.0/(sigma*sigma) (:190)
    a.scores = (float*)ctx->stage[3].p; a.inliers = ib ? (uint8_t*)ctx->stage[1].p : nullptr;
    if ((rc = stage_reserve(ctx, 4, sizeof(float) * vote_scratch_floats(n_hyp, n)))) return rc;
    HIPCHK(ctx, launch_check_fundamental_2p(a, (float*)ctx->stage[4].p, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(scores, a.scores, sb, hipMemcpyDeviceToHost, ctx->stream));
    if (ib && n) HIPCHK(ctx, hipMemcpyAsync(inliers, a.inliers, ib, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return DR3HIP_OK;
}

int dr3hip_check_fundamental_dev(dr3hip_ctx* ctx, const float* d_F21, int n_hyp, const float* d_pts1, const float* d_pts2,
                                 const uint8_t* d_mask, int n, float sigma, float* d_scores, uint8_t* d_inliers)
{
    if (!ctx) return DR3HIP_EINVAL;
    if (!d_F21 || !d_pts1 || !d_pts2 || !d_scores || n_hyp < 1 || n < 0 || n_hyp > (1 << 20) || n > (1 << 24))
        return fail(ctx, DR3HIP_EINVAL, "check_fundamental_dev: bad arguments");
    DeviceGuard g(ctx->device);
    CheckFArgs a;
    a.F = d_F21; a.n_hyp = n_hyp; a.mask = d_mask;
    a.pts1 = d_pts1; a.pts2 = d_pts2; a.n = n;
    a.inv_sigma_square = (float)(1.0 / (double)(sigma * sigma));
    a.scores = d_scores; a.inliers = d_inliers;
    { const int rc = stage_reserve(ctx, 4, sizeof(float) * vote_scratch_floats(n_hyp, n)); if (rc) return rc; }
    HIPCHK(ctx, launch_check_fundamental_2p(a, (float*)ctx->stage[4].p, ctx->stream));
    return DR3HIP_OK;
}

// ---- InitHelper::Normalize --------------------------------------------------------------------------
int dr3hip_normalize_points_dev(dr3hip_ctx* ctx, const float* d_xy, const uint8_t* d_mask, int n, int n_sets,
                                float* d_out_xy, float* d_T)
{
    if (!ctx) return DR3HIP_EINVAL;
    if (!d_xy || !d_out_xy || !d_T || n < 0 || n_sets < 0 || n > (1 << 24))
        return fail(ctx, DR3HIP_EINVAL, "normalize_points_dev: bad arguments");
    if (n_sets == 0) return DR3HIP_OK;
    DeviceGuard g(ctx->device);
    NormArgs a;
    a.xy = d_xy; a.xy_step = 2ll * n; a.mask = d_mask; a.mask_step = n; a.n = n; a.n_sets = n_sets;
    a.out_xy = d_out_xy; a.T = d_T;
    HIPCHK(ctx, launch_normalize_points(a, ctx->stream));
    return DR3HIP_OK;
}

int dr3hip_normalize_points(dr3hip_ctx* ctx, const float* xy, int n, float* out_xy, float T[9])
{
    if (!ctx) return DR3HIP_EINVAL;
    if (!xy || !out_xy || !T || n < 0 || n > (1 << 24)) return fail(ctx, DR3HIP_EINVAL, "normalize_points: bad arguments");
    DeviceGuard g(ctx->device);
    const size_t pb = sizeof(float) * 2 * (size_t)(n ? n : 1);
    int rc;
    if ((rc = stage_reserve(ctx, 2, 2 * pb + 256)) || (rc = stage_reserve(ctx, 3, 256))) return rc;
    uint8_t* dp = (uint8_t*)ctx->stage[2].p;
    if (n) HIPCHK(ctx, hipMemcpyAsync(dp, xy, pb, hipMemcpyHostToDevice, ctx->stream));
    rc = dr3hip_normalize_points_dev(ctx, (const float*)dp, nullptr, n, 1, (float*)(dp + pb), (float*)ctx->stage[3].p);
    if (rc) return rc;
    if (n) HIPCHK(ctx, hipMemcpyAsync(out_xy, dp + pb, pb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(T, ctx->stage[3].p, 9 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return DR3HIP_OK;
}

// ---- panorama paste -------------------------------------------------------------------------
static void make_job(WarpJobDev& j, const void* src, int sw, int sh, int sstride, void* dst, int dw, int dh, int dstride,
                     const float M[9], int flags);      // defined with the warp entry points below
static bool box_w_one_sign(const double M[9], int x0, int y0, int x1, int y1);
static void plan_rect(WarpJobDev& j, int x0, int y0, int x1, int y1);

int dr3hip_panorama_bbox(const float M[9], int w, int h, int* min_x, int* min_y, int* max_x, int* max_y)
{
    if (!M || !min_x || !min_y || !max_x || !max_y) return DR3HIP_EINVAL;
    // src/panorama.cpp:117-124: corners (0,0,1), (w,0,1), (0,h,1), (w,h,1); (M * corners^T)^T is one CV_32F
    // gemm, which accumulates float products in double and rounds once
    const float cx[4] = {0.f, (float)w, 0.f, (float)w}, cy[4] = {0.f, 0.f, (float)h, (float)h};
    float X[4], Y[4];
    for (int i = 0; i < 4; ++i) {
        float t[3];
        for (int r = 0; r < 3; ++r)
            t[r] = (float)((double)M[3 * r] * (double)cx[i] + (double)M[3 * r + 1] * (double)cy[i] + (double)M[3 * r + 2] * 1.0);
        X[i] = t[0] / t[2];                                               // :126-130
        Y[i] = t[1] / t[2];
    }
    *min_x = (int)*std::min_element(X, X + 4); *min_y = (int)*std::min_element(Y, Y + 4);      // :138-141
    *max_x = (int)*std::max_element(X, X + 4); *max_y = (int)*std::max_element(Y, Y + 4);
    return DR3HIP_OK;
}

static void linspace_ref(float a, float b, int n, float* out)            // include/utils.hpp:37-46
{
    const float h = (b - a) / (float)(n - 1);
    float val = a;
    for (int i = 0; i < n; ++i) { out[i] = val; val += h; }
}

int dr3hip_panorama_ramp(int min_x, int max_x, int feathering_width, float* ramp)
{
    if (!ramp) return DR3HIP_EINVAL;
    const int n = max_x - min_x;
    int temp = feathering_width;                                          // :166-170
    if (n - 2 * feathering_width < 0) temp = n / 2 - 1;
    const int mid = n - 2 * temp;
    if (temp < 0 || mid < 0) return DR3HIP_EINVAL;                        // size_t(-1) elements: the reference throws
    linspace_ref(0.f, 1.f, temp, ramp);                                   // :172-177
    for (int i = 0; i < mid; ++i) ramp[temp + i] = 1.0f;
    linspace_ref(1.f, 0.f, temp, ramp + temp + mid);
    return n;
}

int dr3hip_panorama_add(dr3hip_ctx* ctx, const uint8_t* d_img, int w, int h, int istride,
                        float* d_canvas, int cw, int ch, int cstride, const float M[9], int feathering_width)
{
    return dr3hip_panorama_add_slab(ctx, d_img, w, h, istride, d_canvas, cw, ch, cstride, M, feathering_width, 0, ch);
}

// the launch record of one paste without its ramp: box columns and rows, inverse map, and the coordinate form / strip plan of the box.
// Returns the box's column count (max_x - min_x), or a negative code for a box the reference throws on.
static int pano_add_args(const uint8_t* d_img, int w, int h, int istride, float* d_canvas, int cw, int ch, int cstride, const float M[9],
                         int slab_y0, int slab_y1, PanoAddArgs& a, int& min_x, int& max_x)
{
    int min_y, max_y;
    dr3hip_panorama_bbox(M, w, h, &min_x, &min_y, &max_x, &max_y);
    const int n = max_x - min_x;
    if (n < 2 || n > (1 << 24)) return DR3HIP_EINVAL;
    WarpJobDev j;                                                         // the :192 warp: dst = the whole canvas
    make_job(j, d_img, w, h, istride, d_canvas, cw, ch, cstride, M, 0);
    a.img = d_img; a.w = w; a.h = h; a.istride = istride;
    a.canvas = d_canvas; a.cw = cw; a.ch = ch; a.cstride = cstride;
    for (int i = 0; i < 9; ++i) a.M[i] = j.M[i];
    a.bw0 = j.bw0; a.bw0_shift = j.bw0_shift;
    a.cx0 = min_x < 0 ? 0 : min_x; a.cx1 = max_x > cw ? cw : max_x; a.min_x = min_x;
    // The reference walks every canvas row of the box columns; rows whose samples are all border-0 change nothing
    // (:205-207).  A canvas pixel samples the image only if its source coordinate lies within 1 px of it, so the
    // forward image of the rectangle grown by 1.5 px (a convex quadrilateral while W stays positive), grown by
    // another 2 canvas rows, bounds the rows that can change; anything doubtful keeps all rows.
    a.cy0 = 0; a.cy1 = ch;
    {
        const double cxs[2] = {-1.5, w + 0.5}, cys[2] = {-1.5, h + 0.5};
        double ylo = INFINITY, yhi = -INFINITY; bool ok = true;
        for (int i = 0; i < 2 && ok; ++i) for (int k = 0; k < 2 && ok; ++k) {
            const double W = (double)M[6] * cxs[i] + (double)M[7] * cys[k] + (double)M[8];
            const double Y = ((double)M[3] * cxs[i] + (double)M[4] * cys[k] + (double)M[5]) / W;
            if (!(W > 1e-9) || !std::isfinite(Y)) ok = false;
            ylo = std::fmin(ylo, Y); yhi = std::fmax(yhi, Y);
        }
        // W of the inverse map of one sign and bounded away from 0 over the box columns (all rows): no canvas pixel of those columns
        // reaches the image "through infinity"
        if (ok && a.cx1 > a.cx0 && box_w_one_sign(j.M, a.cx0, 0, a.cx1, ch)) {
            const double lo = std::floor(ylo) - 2.0, hi = std::ceil(yhi) + 3.0;
            a.cy0 = lo < 0.0 ? 0 : (lo > ch ? ch : (int)lo);
            a.cy1 = hi > ch ? ch : (hi < 0.0 ? 0 : (int)hi);
        }
    }
    a.slab_y0 = slab_y0;                                                 // only rows [slab_y0, slab_y1) live in d_canvas
    if (a.cy0 < slab_y0) a.cy0 = slab_y0;
    if (a.cy1 > slab_y1) a.cy1 = slab_y1;
    // the coordinate form the kernel may take is decided over the pixels it evaluates -- the box -- not over the whole canvas (a canvas
    // of several images is wider than the +-8000 source pixels the fast forms cover, the box of one image is not)
    plan_rect(j, a.cx0, a.cy0, a.cx1, a.cy1);
    a.fast = j.fast; a.tile = j.tile; a.tile_lds = j.tile_lds; a.tile_guard = j.tile_guard;
    a.ramp = nullptr;
    return n;
}

int dr3hip_panorama_plan(const uint8_t* d_img, int w, int h, int istride, int cw, int ch, const float M[9], int slab_y0, int slab_y1,
                         int* kernel, int* coords, int box[4])
{
    if (!M || w < 1 || h < 1 || cw < 1 || ch < 1 || istride < 3 * w || slab_y0 < 0 || slab_y1 > ch || slab_y0 >= slab_y1) return DR3HIP_EINVAL;
    PanoAddArgs a; int min_x, max_x;
    const int n = pano_add_args(d_img, w, h, istride, nullptr, cw, ch, 16 * cw, M, slab_y0, slab_y1, a, min_x, max_x);
    if (n < 0) return n;
    const bool empty = a.cx1 <= a.cx0 || a.cy1 <= a.cy0;
    if (kernel) *kernel = (!empty && paste_takes_strips(a)) ? 2 : 0;
    if (coords) *coords = a.fast;
    if (box) { box[0] = a.cx0; box[1] = a.cy0; box[2] = a.cx1; box[3] = a.cy1; }
    return DR3HIP_OK;
}

int dr3hip_panorama_add_slab(dr3hip_ctx* ctx, const uint8_t* d_img, int w, int h, int istride,
                             float* d_canvas, int cw, int ch, int cstride, const float M[9], int feathering_width,
                             int slab_y0, int slab_y1)
{
    if (!ctx) return DR3HIP_EINVAL;
    if (!d_img || !d_canvas || !M || w < 1 || h < 1 || cw < 1 || ch < 1 || istride < 3 * w || cstride < 16 * cw ||
        slab_y0 < 0 || slab_y1 > ch || slab_y0 >= slab_y1)
        return fail(ctx, DR3HIP_EINVAL, "panorama_add: bad arguments");
    if ((((uintptr_t)d_canvas | (uintptr_t)cstride) & 15) != 0) return fail(ctx, DR3HIP_EINVAL, "panorama_add: canvas rows must be 16-byte aligned");
    PanoAddArgs a; int min_x, max_x;
    const int n = pano_add_args(d_img, w, h, istride, d_canvas, cw, ch, cstride, M, slab_y0, slab_y1, a, min_x, max_x);
    if (n < 0) return fail(ctx, DR3HIP_EINVAL, "panorama_add: bounding box of %d columns (the reference throws below 2)", max_x - min_x);
    std::vector<float> ramp((size_t)n);
    if (dr3hip_panorama_ramp(min_x, max_x, feathering_width, ramp.data()) != n)
        return fail(ctx, DR3HIP_EINVAL, "panorama_add: feathering width %d does not fit a %d-column box", feathering_width, n);
    DeviceGuard g(ctx->device);
    int rc;
    if ((rc = stage_reserve(ctx, 6, sizeof(float) * (size_t)n))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->stage[6].p, ramp.data(), sizeof(float) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));                       // ramp is a local
    a.ramp = (const float*)ctx->stage[6].p;
    ProfEvent pe; prof_begin(ctx, DR3HIP_PROF_PANO_ADD, pe);
    HIPCHK(ctx, launch_panorama_add(a, ctx->stream));
    prof_end(ctx, pe);
    return DR3HIP_OK;
}

int dr3hip_panorama_normalize(dr3hip_ctx* ctx, const float* d_canvas, int cw, int ch, int cstride, uint8_t* d_out, int ostride)
{
    if (!ctx) return DR3HIP_EINVAL;
    if (!d_canvas || !d_out || cw < 1 || ch < 1 || cstride < 16 * cw || ostride < 3 * cw || (((uintptr_t)d_canvas | (uintptr_t)cstride) & 15) != 0)
        return fail(ctx, DR3HIP_EINVAL, "panorama_normalize: bad arguments");
    DeviceGuard g(ctx->device);
    PanoNormArgs a; a.canvas = d_canvas; a.cw = cw; a.ch = ch; a.cstride = cstride; a.out = d_out; a.ostride = ostride;
    ProfEvent pe; prof_begin(ctx, DR3HIP_PROF_PANO_NORM, pe);
    HIPCHK(ctx, launch_panorama_normalize(a, ctx->stream));
    prof_end(ctx, pe);
    return DR3HIP_OK;
}

// ---- stitch compose (src/stitch.cpp:33-80) -------------------------------------------------------------------------
static void gemm33_f32(const float A[9], const float B[9], float C[9])   // CV_32F gemm: double accumulation, one rounding
{
    float t[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double acc = 0.0;
            for (int k = 0; k < 3; ++k) acc += (double)A[3 * i + k] * (double)B[3 * k + j];
            t[3 * i + j] = (float)acc;
        }
    memcpy(C, t, sizeof(t));
}

int dr3hip_stitch_geometry(const float H[9], int w, int h, float Hinv[9], float T[9], float M_right[9], int* new_w, int* new_h)
{
    if (!H || !Hinv || !T || !M_right || !new_w || !new_h || w < 1 || h < 1) return DR3HIP_EINVAL;
    // cv::Mat::inv() on a CV_32F 3x3 (:34): determinant and cofactors in double, stored as float
    const double d0 = (double)H[0] * ((double)H[4] * H[8] - (double)H[5] * H[7]) - (double)H[1] * ((double)H[3] * H[8] - (double)H[5] * H[6]) +
                      (double)H[2] * ((double)H[3] * H[7] - (double)H[4] * H[6]);
    if (d0 == 0.0) return DR3HIP_EINVAL;
    const double d = 1.0 / d0;
    float inv[9];
    inv[0] = (float)(((double)H[4] * H[8] - (double)H[5] * H[7]) * d); inv[1] = (float)(((double)H[2] * H[7] - (double)H[1] * H[8]) * d);
    inv[2] = (float)(((double)H[1] * H[5] - (double)H[2] * H[4]) * d); inv[3] = (float)(((double)H[5] * H[6] - (double)H[3] * H[8]) * d);
    inv[4] = (float)(((double)H[0] * H[8] - (double)H[2] * H[6]) * d); inv[5] = (float)(((double)H[2] * H[3] - (double)H[0] * H[5]) * d);
    inv[6] = (float)(((double)H[3] * H[7] - (double)H[4] * H[6]) * d); inv[7] = (float)(((double)H[1] * H[6] - (double)H[0] * H[7]) * d);
    inv[8] = (float)(((double)H[0] * H[4] - (double)H[1] * H[3]) * d);
    const float scale = (float)(1.0 / (double)inv[8]);                  // Hinv * (1.0 / Hinv.at<float>(2, 2))  (:35)
    for (int i = 0; i < 9; ++i) Hinv[i] = inv[i] * scale;
    const float cx[4] = {0.f, (float)w, 0.f, (float)w}, cy[4] = {0.f, 0.f, (float)h, (float)h};
    float X[8], Y[8];
    for (int i = 0; i < 4; ++i) {                                       // :39-60
        float t[3];
        for (int r = 0; r < 3; ++r)
            t[r] = (float)((double)Hinv[3 * r] * (double)cx[i] + (double)Hinv[3 * r + 1] * (double)cy[i] + (double)Hinv[3 * r + 2] * 1.0);
        X[i] = t[0] / t[2]; Y[i] = t[1] / t[2];
        X[4 + i] = cx[i]; Y[4 + i] = cy[i];
    }
    const float mnx = *std::min_element(X, X + 8), mny = *std::min_element(Y, Y + 8);
    const float mxx = *std::max_element(X, X + 8), mxy = *std::max_element(Y, Y + 8);
    *new_w = (int)(std::ceil(mxx) - std::floor(mnx));                   // :67-68
    *new_h = (int)(std::ceil(mxy) - std::floor(mny));
    T[0] = 1.f; T[1] = 0.f; T[2] = -mnx; T[3] = 0.f; T[4] = 1.f; T[5] = -mny; T[6] = 0.f; T[7] = 0.f; T[8] = 1.f;   // :70-71
    gemm33_f32(T, Hinv, M_right);                                       // translate_top_left_mat * Hinv (:73)
    return DR3HIP_OK;
}

int dr3hip_stitch_plan(const uint8_t* d_left, int lw, int lh, int lstride, const uint8_t* d_right, int rw, int rh, int rstride,
                       const float T[9], const float M_right[9], int new_w, int new_h, float alpha, float beta, float gamma, int* kernel)
{
    if (!T || !M_right || !kernel || lw < 1 || lh < 1 || rw < 1 || rh < 1 || new_w < 1 || new_h < 1) return DR3HIP_EINVAL;
    StitchArgs a;
    make_job(a.job[0], d_left, lw, lh, lstride, nullptr, new_w, new_h, 3 * new_w, T, 0);
    make_job(a.job[1], d_right, rw, rh, rstride, nullptr, new_w, new_h, 3 * new_w, M_right, 0);
    a.dst = nullptr; a.dw = new_w; a.dh = new_h; a.dstride = 3 * new_w;
    a.alpha = alpha; a.beta = beta; a.gamma = gamma;
    *kernel = stitch_takes_strips(a) ? 2 : 0;
    return DR3HIP_OK;
}

int dr3hip_stitch_compose_dev(dr3hip_ctx* ctx, const uint8_t* d_left, int lw, int lh, int lstride,
                              const uint8_t* d_right, int rw, int rh, int rstride,
                              const float T[9], const float M_right[9], int new_w, int new_h,
                              float alpha, float beta, float gamma, uint8_t* d_out, int ostride)
{
    if (!ctx) return DR3HIP_EINVAL;
    if (!d_left || !d_right || !T || !M_right || !d_out || lw < 1 || lh < 1 || rw < 1 || rh < 1 || new_w < 1 || new_h < 1 ||
        lstride < 3 * lw || rstride < 3 * rw || ostride < 3 * new_w)
        return fail(ctx, DR3HIP_EINVAL, "stitch_compose: bad arguments");
    DeviceGuard g(ctx->device);
    StitchArgs a;
    make_job(a.job[0], d_left, lw, lh, lstride, d_out, new_w, new_h, ostride, T, 0);           // :74
    make_job(a.job[1], d_right, rw, rh, rstride, d_out, new_w, new_h, ostride, M_right, 0);    // :73
    a.dst = d_out; a.dw = new_w; a.dh = new_h; a.dstride = ostride;
    a.alpha = alpha; a.beta = beta; a.gamma = gamma;
    ProfEvent pe; prof_begin(ctx, DR3HIP_PROF_STITCH, pe);
    HIPCHK(ctx, launch_stitch_compose(a, ctx->stream));
    prof_end(ctx, pe);
    return DR3HIP_OK;
}

int dr3hip_stitch_compose(dr3hip_ctx* ctx, const uint8_t* left, int lw, int lh, int lstride,
                          const uint8_t* right, int rw, int rh, int rstride,
                          const float T[9], const float M_right[9], int new_w, int new_h,
                          float alpha, float beta, float gamma, uint8_t* out, int ostride)
{
    if (!ctx) return DR3HIP_EINVAL;
    if (!left || !right || !out || lw < 1 || lh < 1 || rw < 1 || rh < 1 || new_w < 1 || new_h < 1 ||
        lstride < 3 * lw || rstride < 3 * rw || ostride < 3 * new_w)
        return fail(ctx, DR3HIP_EINVAL, "stitch_compose: bad arguments");
    DeviceGuard g(ctx->device);
    const int lp = align_up(3 * lw, 64), rp = align_up(3 * rw, 64), op = align_up(3 * new_w, 64);
    int rc;
    if ((rc = stage_reserve(ctx, 0, (size_t)lp * lh)) || (rc = stage_reserve(ctx, 2, (size_t)rp * rh)) || (rc = stage_reserve(ctx, 1, (size_t)op * new_h))) return rc;
    if ((rc = copy2d_h2d(ctx, ctx->stage[0].p, lp, left, lstride, (size_t)3 * lw, lh))) return rc;
    if ((rc = copy2d_h2d(ctx, ctx->stage[2].p, rp, right, rstride, (size_t)3 * rw, rh))) return rc;
    if ((rc = dr3hip_stitch_compose_dev(ctx, (const uint8_t*)ctx->stage[0].p, lw, lh, lp, (const uint8_t*)ctx->stage[2].p, rw, rh, rp,
                                        T, M_right, new_w, new_h, alpha, beta, gamma, (uint8_t*)ctx->stage[1].p, op))) return rc;
    if ((rc = copy2d_d2h(ctx, out, ostride, ctx->stage[1].p, op, (size_t)3 * new_w, new_h))) return rc;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return DR3HIP_OK;
}

// ---- multi-GPU: RCCL broadcast --------------------------------------------------------------
// librccl is resolved with dlopen on first use: libdr3hip.so has no link-time dependency on it, and a
// process that already carries an RCCL (PyTorch bundles one under the same SONAME) shares that copy.
namespace {
struct RcclId128 { char b[128]; };          // ncclUniqueId: NCCL_UNIQUE_ID_BYTES = 128, passed by value
struct Rccl {
    void* h = nullptr;
    int  (*GetUniqueId)(void*) = nullptr;
    int  (*CommInitRank)(void**, int, RcclId128, int) = nullptr;
    int  (*CommDestroy)(void*) = nullptr;
    int  (*Broadcast)(const void*, void*, size_t, int /*ncclDataType_t*/, int, void*, hipStream_t) = nullptr;
    int  (*GroupStart)() = nullptr;
    int  (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string err;
};
Rccl g_rccl;

bool rccl_load()
{
    if (g_rccl.h) return true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) { g_rccl.err = std::string("dlopen(librccl) failed: ") + (dlerror() ? dlerror() : "?"); return false; }
#define DR3_SYM(field, name) \
    *(void**)(&g_rccl.field) = dlsym(h, name); \
    if (!g_rccl.field) { g_rccl.err = std::string("librccl lacks ") + name; dlclose(h); return false; }
    DR3_SYM(GetUniqueId, "ncclGetUniqueId") DR3_SYM(CommInitRank, "ncclCommInitRank") DR3_SYM(CommDestroy, "ncclCommDestroy")
    DR3_SYM(Broadcast, "ncclBroadcast") DR3_SYM(GroupStart, "ncclGroupStart") DR3_SYM(GroupEnd, "ncclGroupEnd")
    DR3_SYM(GetErrorString, "ncclGetErrorString")
#undef DR3_SYM
    g_rccl.h = h;
    return true;
}
const int kNcclUint8 = 1;     // ncclUint8 / ncclChar family: ncclInt8 = 0, ncclUint8 = 1 (rccl.h)
}  // namespace

#define NCCLCHK(ctx, call)                                                                       \
    do { int r_ = (call);                                                                         \
         if (r_ != 0) return fail((ctx), DR3HIP_EHIP, "%s: %s", #call, g_rccl.GetErrorString(r_)); } while (0)

int dr3hip_comm_unique_id(dr3hip_ctx* ctx, void* id128)
{
    if (!ctx || !id128) return DR3HIP_EINVAL;
    if (!rccl_load()) return fail(ctx, DR3HIP_ENODEV, "%s", g_rccl.err.c_str());
    NCCLCHK(ctx, g_rccl.GetUniqueId(id128));
    return DR3HIP_OK;
}

int dr3hip_comm_init(dr3hip_ctx* ctx, const void* id128, int world, int rank)
{
    if (!ctx || !id128 || world < 1 || rank < 0 || rank >= world) return fail(ctx, DR3HIP_EINVAL, "comm_init: bad arguments");
    if (ctx->comm) return fail(ctx, DR3HIP_EINVAL, "comm_init: this context already has a communicator");
    if (!rccl_load()) return fail(ctx, DR3HIP_ENODEV, "%s", g_rccl.err.c_str());
    DeviceGuard g(ctx->device);
    RcclId128 id; memcpy(id.b, id128, sizeof(id.b));
    void* comm = nullptr;
    NCCLCHK(ctx, g_rccl.CommInitRank(&comm, world, id, rank));
    ctx->comm = comm; ctx->comm_world = world; ctx->comm_rank = rank;
    return DR3HIP_OK;
}

int dr3hip_comm_destroy(dr3hip_ctx* ctx)
{
    if (!ctx) return DR3HIP_EINVAL;
    if (!ctx->comm) return DR3HIP_OK;
    DeviceGuard g(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)g_rccl.CommDestroy(ctx->comm);
    ctx->comm = nullptr; ctx->comm_world = 0; ctx->comm_rank = -1;
    return DR3HIP_OK;
}

int dr3hip_comm_world(const dr3hip_ctx* ctx) { return ctx ? ctx->comm_world : 0; }
int dr3hip_comm_rank(const dr3hip_ctx* ctx) { return ctx ? ctx->comm_rank : -1; }

int dr3hip_bcast(dr3hip_ctx* ctx, void* dptr, size_t bytes, int root)
{
    if (!ctx) return DR3HIP_EINVAL;
    if (!ctx->comm) return fail(ctx, DR3HIP_EINVAL, "bcast: no communicator (dr3hip_comm_init)");
    if (!dptr || root < 0 || root >= ctx->comm_world) return fail(ctx, DR3HIP_EINVAL, "bcast: bad arguments");
    if (bytes == 0) return DR3HIP_OK;
    DeviceGuard g(ctx->device);
    NCCLCHK(ctx, g_rccl.Broadcast(dptr, dptr, bytes, kNcclUint8, root, ctx->comm, ctx->stream));
    return DR3HIP_OK;
}

long long dr3hip_batch_bcast_slot(dr3hip_batch* b, int slot, int root, int with_deriv)
{
    if (!b) return DR3HIP_EINVAL;
    dr3hip_ctx* ctx = b->ctx;
    if (!ctx->comm) return fail(ctx, DR3HIP_EINVAL, "batch_bcast_slot: no communicator (dr3hip_comm_init)");
    if (slot < 0 || slot >= b->capacity || root < 0 || root >= ctx->comm_world) return fail(ctx, DR3HIP_EINVAL, "batch_bcast_slot: bad slot/root");
    DeviceGuard g(ctx->device);
    { const int rc = ahead_join(ctx, b, slot, 1); if (rc) return rc; }
    long long moved = 0;
    NCCLCHK(ctx, g_rccl.GroupStart());
    for (int l = 0; l < b->levels; ++l) {
        const LevelDev& v = b->lv[l];
        // whole rows including the pitch slack and the physical border: the receiver's plane becomes byte-identical
        uint8_t* pi = v.img + (size_t)v.step * slot - ((size_t)v.pad_y * v.pitch + v.pad_x);
        const size_t ib = (size_t)v.pitch * (v.h + 2 * v.pad_y);
        int r = g_rccl.Broadcast(pi, pi, ib, kNcclUint8, root, ctx->comm, ctx->stream);
        moved += (long long)ib;
        if (r == 0 && with_deriv) {
            uint8_t* pd = (uint8_t*)v.der + (size_t)v.dstep * slot - ((size_t)v.pad_y * v.dpitch + (size_t)4 * v.pad_x);
            const size_t db = (size_t)v.dpitch * (v.h + 2 * v.pad_y);
            r = g_rccl.Broadcast(pd, pd, db, kNcclUint8, root, ctx->comm, ctx->stream);
            moved += (long long)db;
        }
        if (r != 0) { (void)g_rccl.GroupEnd(); return fail(ctx, DR3HIP_EHIP, "ncclBroadcast (level %d): %s", l, g_rccl.GetErrorString(r)); }
    }
    NCCLCHK(ctx, g_rccl.GroupEnd());
    return moved;
}

// ---- timing ---------------------------------------------------------------------------------------
int dr3hip_timer_start(dr3hip_ctx* ctx)
{
    if (!ctx) return DR3HIP_EINVAL;
    DeviceGuard g(ctx->device);
    HIPCHK(ctx, hipEventRecord(ctx->t0, ctx->stream));
    return DR3HIP_OK;
}
int dr3hip_timer_stop(dr3hip_ctx* ctx, float* ms)
{
    if (!ctx || !ms) return DR3HIP_EINVAL;
    DeviceGuard g(ctx->device);
    HIPCHK(ctx, hipEventRecord(ctx->t1, ctx->stream));
    HIPCHK(ctx, hipEventSynchronize(ctx->t1));
    HIPCHK(ctx, hipEventElapsedTime(ms, ctx->t0, ctx->t1));
    return DR3HIP_OK;
}
int dr3hip_profile_enable(dr3hip_ctx* ctx, int on)
{
    if (!ctx) return DR3HIP_EINVAL;
    ctx->prof = on != 0;
    return DR3HIP_OK;
}
int dr3hip_profile_read(dr3hip_ctx* ctx, int which, double* total_ms, int* n)
{
    if (!ctx || which < 0 || which >= DR3HIP_PROF_SLOTS) return DR3HIP_EINVAL;
    DeviceGuard g(ctx->device);
    prof_drain(ctx);
    if (total_ms) *total_ms = ctx->prof_ms[which];
    if (n) *n = ctx->prof_n[which];
    return DR3HIP_OK;
}
int dr3hip_profile_reset(dr3hip_ctx* ctx)
{
    if (!ctx) return DR3HIP_EINVAL;
    DeviceGuard g(ctx->device);
    prof_drain(ctx);
    for (int i = 0; i < DR3HIP_PROF_SLOTS; ++i) { ctx->prof_ms[i] = 0; ctx->prof_n[i] = 0; }
    return DR3HIP_OK;
}

// ---- LK geometry ----------------------------------------------------------------------------------
int dr3hip_lk_effective_max_level(int w, int h, int win_w, int win_h, int max_level)
{
    // cv::buildOpticalFlowPyramid's early stop
    for (int level = 0; level <= max_level; ++level) {
        w = (w + 1) / 2; h = (h + 1) / 2;
        if (w <= win_w || h <= win_h) return level;
    }
    return max_level;
}

// ---- batch ---------------------------------------------------------------------------------------
int dr3hip_batch_create(dr3hip_ctx* ctx, int w, int h, int win_w, int win_h, int max_level,
                        int capacity, dr3hip_batch** out)
{
    if (!ctx || !out) return DR3HIP_EINVAL;
    *out = nullptr;
    if (w < 4 || h < 4 || capacity < 1 || max_level < 0 || win_w <= 2 || win_h <= 2)
        return fail(ctx, DR3HIP_EINVAL, "batch_create: bad geometry w=%d h=%d win=%dx%d maxLevel=%d capacity=%d", w, h, win_w, win_h, max_level, capacity);
    if (max_level >= DR3HIP_MAX_LEVELS) return fail(ctx, DR3HIP_ERANGE, "batch_create: maxLevel %d >= %d", max_level, DR3HIP_MAX_LEVELS);
    if (win_w > DR3HIP_MAX_WIN || win_h > DR3HIP_MAX_WIN)
        return fail(ctx, DR3HIP_ERANGE, "batch_create: window %dx%d exceeds this build's limit (%d)", win_w, win_h, DR3HIP_MAX_WIN);
    // an image no larger than the window is fine (cv::buildOpticalFlowPyramid keeps level 0 alone then; the border is REFLECT_101 repeated,
    // which is what the border fill computes)
    // sides as at every entry point, and rows the tracker can address: it forms row offsets with 24-bit multiplies (__umul24), so the
    // derivative pitch 4 (w + 2 pad) + 16 must stay below 2^24 -- a frame up to DR3HIP_BATCH_MAX_W pixels wide
    if (!side_ok(w) || !side_ok(h) || w > DR3HIP_BATCH_MAX_W)
        return fail(ctx, DR3HIP_ERANGE, "batch_create: frame %d x %d exceeds the batch limits (width <= %d)", w, h, DR3HIP_BATCH_MAX_W);
    DeviceGuard g(ctx->device);
    dr3hip_batch* b = new (std::nothrow) dr3hip_batch();
    if (!b) return fail(ctx, DR3HIP_ENOMEM, "batch_create: out of host memory");
    b->ctx = ctx; b->w = w; b->h = h; b->win_w = win_w; b->win_h = win_h; b->capacity = capacity;
    const int L = dr3hip_lk_effective_max_level(w, h, win_w, win_h, max_level);
    b->levels = L + 1;
    memset(b->lv, 0, sizeof(b->lv));
    int lw = w, lh = h;
    for (int l = 0; l <= L; ++l) {
        LevelDev& v = b->lv[l];
        v.w = lw; v.h = lh;
        // every level carries OpenCV's winSize border physically (lk_kernels.hip): wide enough for the template window,
        // the staged search region (window + 4 px margin) and their dword alignment slop.  Level 0's is a multiple of 64 so
        // that the interior rows the fused pyramid kernel streams keep their 64-byte alignment.
        v.pad_x = l ? align_up(win_w + 16, 4) : align_up(win_w + 16, 64);
        v.pad_y = win_h + 6;
        v.pitch = align_up(lw + 2 * v.pad_x + 8, 64);
        v.dpitch = align_up(4 * (lw + 2 * v.pad_x) + 16, 128);
        const int rows = lh + 2 * v.pad_y;
        if (v.dpitch >= (1 << 24)) { dr3hip_batch_destroy(b); return fail(ctx, DR3HIP_ERANGE, "batch_create: level %d rows of %d bytes exceed the tracker's 24-bit row pitch", l, v.dpitch); }
        // the kernels address a plane with 31-bit byte offsets
        if ((long long)v.dpitch * (rows + 17) > 0x7fffff00ll) {
            dr3hip_batch_destroy(b);
            return fail(ctx, DR3HIP_ERANGE, "batch_create: level %d plane of %d x %d exceeds the 2 GiB a plane may span", l, lw, lh);
        }
        v.step = align_up((long long)v.pitch * rows, 256ll);
        v.dstep = align_up((long long)v.dpitch * rows, 256ll);
        // slack: the tracker's run-time-window forms load (and mask) up to 16 rows below a window that touches the bottom border
        const size_t ib = (size_t)v.step * capacity + 256 + (size_t)17 * v.pitch, db = (size_t)v.dstep * capacity + 256 + (size_t)17 * v.dpitch;
        void *pi = nullptr, *pd = nullptr;
        if (hipMalloc(&pi, ib) != hipSuccess || hipMalloc(&pd, db) != hipSuccess) {
            if (pi) (void)hipFree(pi);
            v.img = nullptr; v.der = nullptr;
            dr3hip_batch_destroy(b);
            return fail(ctx, DR3HIP_ENOMEM, "batch_create: hipMalloc failed at level %d (%zu + %zu bytes)", l, ib, db);
        }
        b->img_base[l] = pi; b->der_base[l] = pd;
        // img / der address the interior origin; pitch slack and borders are read by aligned over-reads: keep them defined
        v.img = (uint8_t*)pi + (size_t)v.pad_y * v.pitch + v.pad_x;
        v.der = (int16_t*)((uint8_t*)pd + (size_t)v.pad_y * v.dpitch + (size_t)4 * v.pad_x);
        (void)hipMemsetAsync(pi, 0, ib, ctx->stream);
        (void)hipMemsetAsync(pd, 0, db, ctx->stream);
        b->bytes += ib + db;
        lw = (lw + 1) / 2; lh = (lh + 1) / 2;
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *out = b;
    return DR3HIP_OK;
}

void dr3hip_batch_destroy(dr3hip_batch* b)
{
    if (!b) return;
    DeviceGuard g(b->ctx->device);
    (void)ahead_join(b->ctx, b, 0, -1);                          // nothing of this batch stays in flight on the auxiliary stream
    if (b->ctx->aux) (void)hipStreamSynchronize(b->ctx->aux);
    (void)hipStreamSynchronize(b->ctx->stream);
    for (int l = 0; l < DR3HIP_MAX_LEVELS; ++l) {
        if (b->img_base[l]) (void)hipFree(b->img_base[l]);
        if (b->der_base[l]) (void)hipFree(b->der_base[l]);
        if (b->det.hs[l]) (void)hipFree(b->det.hs[l]);
    }
    if (b->det.keys) (void)hipFree(b->det.keys);
    if (b->ctx->lk_batch == b) b->ctx->lk_batch = nullptr;
    delete b;
}

int dr3hip_batch_levels(const dr3hip_batch* b) { return b ? b->levels : DR3HIP_EINVAL; }
size_t dr3hip_batch_bytes(const dr3hip_batch* b) { return b ? b->bytes : 0; }

int dr3hip_batch_level_info(const dr3hip_batch* b, int level, dr3hip_level_info* out)
{
    if (!b || !out || level < 0 || level >= b->levels) return DR3HIP_EINVAL;
    const LevelDev& v = b->lv[level];
    out->w = v.w; out->h = v.h;
    out->img_pitch = v.pitch; out->img_slot_step = (size_t)v.step; out->img = v.img;
    out->der_pitch = v.dpitch; out->der_slot_step = (size_t)v.dstep; out->der = v.der;
    return DR3HIP_OK;
}

int dr3hip_batch_upload(dr3hip_batch* b, int first, int count, const uint8_t* frames, int stride, size_t frame_step)
{
    if (!b || !frames || first < 0 || count < 0 || first + count > b->capacity || stride < b->w)
        return b ? fail(b->ctx, DR3HIP_EINVAL, "batch_upload: bad arguments") : DR3HIP_EINVAL;
    dr3hip_ctx* ctx = b->ctx;
    DeviceGuard g(ctx->device);
    const LevelDev& v = b->lv[0];
    // one linear PCIe copy per frame into the bounce buffer (copy2d_h2d explains why), then the device-side change of pitch
    // writes the slot's level 0 together with its REFLECT_101 border: whole padded rows, no second pass over the plane
    const size_t bytes = (size_t)stride * (b->h - 1) + b->w;
    int rc;
    if ((rc = stage_reserve(ctx, 7, bytes + 64))) return rc;
    for (int i = 0; i < count; ++i) {
        HIPCHK(ctx, hipMemcpyAsync(ctx->stage[7].p, frames + (size_t)i * frame_step, bytes, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, launch_upload_level0((const uint8_t*)ctx->stage[7].p, (size_t)stride, v, first + i, align_up(b->win_w + 16, 4), ctx->stream, 1, 0));
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return DR3HIP_OK;
}

// ---- streaming transfers ----------------------------------------------------------------------------------------
static int xfer_init(dr3hip_ctx* ctx, size_t chunk_bytes)
{
    dr3hip_ctx::Xfer& x = ctx->xf;
    if (!x.ready) {
        // every stream and event must exist before the transfer state counts as initialised: a half-built state is torn down again,
        // so that the next call starts from nothing instead of using null handles
        hipError_t e = hipStreamCreateWithFlags(&x.up, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&x.down, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&x.computed, hipEventDisableTiming);
        for (int i = 0; i < dr3hip_ctx::Xfer::K && e == hipSuccess; ++i) {
            e = hipEventCreateWithFlags(&x.copied[i], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&x.consumed[i], hipEventDisableTiming);
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            xfer_release(ctx);
            return fail(ctx, DR3HIP_EHIP, "streaming transfers: %s", hipGetErrorString(e));
        }
        x.ready = true;
    }
    if (x.cap >= chunk_bytes) return DR3HIP_OK;
    HIPCHK(ctx, hipStreamSynchronize(x.up));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const size_t cap = chunk_bytes + chunk_bytes / 8 + 4096;
    for (int i = 0; i < dr3hip_ctx::Xfer::K; ++i) {
        if (x.dev[i]) (void)hipFree(x.dev[i]);
        if (x.pin[i]) (void)hipHostFree(x.pin[i]);
        x.dev[i] = x.pin[i] = nullptr; x.used[i] = false;
        if (hipMalloc(&x.dev[i], cap) != hipSuccess) { x.dev[i] = nullptr; x.cap = 0; return fail(ctx, DR3HIP_ENOMEM, "hipMalloc(%zu) failed", cap); }
    }
    x.cap = cap;
    return DR3HIP_OK;
}

static bool host_pinned(const void* p)
{
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeHost;
}

int dr3hip_host_alloc(dr3hip_ctx* ctx, size_t bytes, void** out)
{
    if (!ctx || !out) return DR3HIP_EINVAL;
    DeviceGuard g(ctx->device);
    *out = nullptr;
    if (hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { *out = nullptr; return fail(ctx, DR3HIP_ENOMEM, "hipHostMalloc(%zu) failed", bytes); }
    return DR3HIP_OK;
}
int dr3hip_host_free(dr3hip_ctx* ctx, void* p)
{
    if (!ctx) return DR3HIP_EINVAL;
    if (!p) return DR3HIP_OK;
    DeviceGuard g(ctx->device);
    if (ctx->xf.up) HIPCHK(ctx, hipStreamSynchronize(ctx->xf.up));
    if (ctx->xf.down) HIPCHK(ctx, hipStreamSynchronize(ctx->xf.down));
    HIPCHK(ctx, hipHostFree(p));
    return DR3HIP_OK;
}

// Frames cross the link in chunks on the context's upload stream while the compute stream works on what it was given before: a
// chunk lands in a device bounce buffer in the host's layout (one linear copy; frames stored back to back travel as ONE copy),
// the compute stream waits for that copy's event and writes the slots' level 0 with its border (upload_level0_kernel, one launch
// per chunk), and the bounce buffer goes back to the ring when that kernel has run.  Nothing here waits on the host except for a
// ring slot whose previous use has not finished (pageable sources only: they are staged through pinned memory by the calling
// thread, so they may be reused on return; pinned sources -- dr3hip_host_alloc -- are read by the copy engine until
// dr3hip_batch_upload_wait).  The slots themselves are only written by the compute stream, in its order: an upload enqueued
// behind a tracker launch never overtakes it.
int dr3hip_batch_upload_async(dr3hip_batch* b, int first, int count, const uint8_t* frames, int stride, size_t frame_step)
{
    if (!b || !frames || first < 0 || count < 0 || first + count > b->capacity || stride < b->w)
        return b ? fail(b->ctx, DR3HIP_EINVAL, "batch_upload_async: bad arguments") : DR3HIP_EINVAL;
    dr3hip_ctx* ctx = b->ctx;
    if (count == 0) return DR3HIP_OK;
    DeviceGuard g(ctx->device);
    const LevelDev& v = b->lv[0];
    const size_t fbytes = (size_t)stride * (b->h - 1) + b->w;              // what is read of one frame
    if (count > 1 && frame_step < fbytes) return fail(ctx, DR3HIP_EINVAL, "batch_upload_async: frames overlap");
    const bool packed = count == 1 || frame_step <= fbytes + 4096;          // back to back: a chunk is one linear copy
    const size_t dstep = packed ? (count == 1 ? fbytes : frame_step) : (size_t)align_up((long long)fbytes, 256ll);
    const size_t target = (size_t)8 << 20;
    const int per_chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)count, target / dstep));
    int rc;
    if ((rc = xfer_init(ctx, (size_t)per_chunk * dstep))) return rc;
    dr3hip_ctx::Xfer& x = ctx->xf;
    const bool pinned = host_pinned(frames);
    for (int done = 0; done < count; done += per_chunk) {
        const int n = std::min(per_chunk, count - done), r = x.next;
        x.next = (x.next + 1) % dr3hip_ctx::Xfer::K;
        const uint8_t* src = frames + (size_t)done * frame_step;
        const size_t cbytes = (size_t)(n - 1) * dstep + fbytes;
        if (x.used[r]) HIPCHK(ctx, hipStreamWaitEvent(x.up, x.consumed[r], 0));      // the bounce buffer's previous content has been used
        const uint8_t* from = src;
        if (!pinned) {
            if (!x.pin[r] && hipHostMalloc(&x.pin[r], x.cap, hipHostMallocDefault) != hipSuccess) { x.pin[r] = nullptr; return fail(ctx, DR3HIP_ENOMEM, "hipHostMalloc(%zu) failed", x.cap); }
            if (x.used[r]) HIPCHK(ctx, hipEventSynchronize(x.copied[r]));            // the staging buffer's previous content has left
            if (packed) memcpy(x.pin[r], src, cbytes);
            else for (int i = 0; i < n; ++i) memcpy((uint8_t*)x.pin[r] + (size_t)i * dstep, src + (size_t)i * frame_step, fbytes);
            from = (const uint8_t*)x.pin[r];
        }
        if (packed || !pinned) HIPCHK(ctx, hipMemcpyAsync(x.dev[r], from, cbytes, hipMemcpyHostToDevice, x.up));
        else for (int i = 0; i < n; ++i)
            HIPCHK(ctx, hipMemcpyAsync((uint8_t*)x.dev[r] + (size_t)i * dstep, src + (size_t)i * frame_step, fbytes, hipMemcpyHostToDevice, x.up));
        HIPCHK(ctx, hipEventRecord(x.copied[r], x.up));
        HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, x.copied[r], 0));
        HIPCHK(ctx, launch_upload_level0((const uint8_t*)x.dev[r], (size_t)stride, v, first + done, align_up(b->win_w + 16, 4), ctx->stream, n, dstep));
        HIPCHK(ctx, hipEventRecord(x.consumed[r], ctx->stream));
        x.used[r] = true;
    }
    return DR3HIP_OK;
}

int dr3hip_batch_upload_wait(dr3hip_batch* b)
{
    if (!b) return DR3HIP_EINVAL;
    dr3hip_ctx* ctx = b->ctx;
    DeviceGuard g(ctx->device);
    if (ctx->xf.up) HIPCHK(ctx, hipStreamSynchronize(ctx->xf.up));
    return DR3HIP_OK;
}

// results leave on the download stream, behind everything the compute stream has been given so far, while it goes on
int dr3hip_download_async(dr3hip_ctx* ctx, void* host, const void* dev, size_t bytes)
{
    if (!ctx || (bytes && (!host || !dev))) return DR3HIP_EINVAL;
    if (!bytes) return DR3HIP_OK;
    DeviceGuard g(ctx->device);
    int rc;
    if ((rc = xfer_init(ctx, 0))) return rc;
    dr3hip_ctx::Xfer& x = ctx->xf;
    HIPCHK(ctx, hipEventRecord(x.computed, ctx->stream));
    HIPCHK(ctx, hipStreamWaitEvent(x.down, x.computed, 0));
    HIPCHK(ctx, hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, x.down));
    return DR3HIP_OK;
}
int dr3hip_download_wait(dr3hip_ctx* ctx)
{
    if (!ctx) return DR3HIP_EINVAL;
    DeviceGuard g(ctx->device);
    if (ctx->xf.down) HIPCHK(ctx, hipStreamSynchronize(ctx->xf.down));
    return DR3HIP_OK;
}

int dr3hip_batch_pad_level0(dr3hip_batch* b, int first, int count)
{
    if (!b || first < 0 || count < 0 || first + count > b->capacity)
        return b ? fail(b->ctx, DR3HIP_EINVAL, "batch_pad_level0: slots outside capacity") : DR3HIP_EINVAL;
    dr3hip_ctx* ctx = b->ctx;
    DeviceGuard g(ctx->device);
    for (int s0 = 0; s0 < count; s0 += 32768)
        HIPCHK(ctx, launch_pad_reflect(&b->lv[0], 1, align_up(b->win_w + 16, 4), first + s0, count - s0 < 32768 ? count - s0 : 32768, ctx->stream));
    return DR3HIP_OK;
}

int dr3hip_batch_download(dr3hip_batch* b, int slot, int level, int what, void* host, int host_stride)
{
    if (!b || !host || slot < 0 || slot >= b->capacity || level < 0 || level >= b->levels || (what != 0 && what != 1))
        return b ? fail(b->ctx, DR3HIP_EINVAL, "batch_download: bad arguments") : DR3HIP_EINVAL;
    dr3hip_ctx* ctx = b->ctx;
    DeviceGuard g(ctx->device);
    { const int rc = ahead_join(ctx, b, slot, 1); if (rc) return rc; }
    const LevelDev& v = b->lv[level];
    if (what == 0)
        { const int rc2_ = copy2d_d2h(ctx, host, host_stride, v.img + (size_t)slot * v.step, v.pitch, v.w, v.h); if (rc2_) return rc2_; }
    else
        { const int rc2_ = copy2d_d2h(ctx, host, host_stride, (uint8_t*)v.der + (size_t)slot * v.dstep, v.dpitch, (size_t)4 * v.w, v.h); if (rc2_) return rc2_; }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return DR3HIP_OK;
}

// ahead: the launches of levels >= 1 go to the context's auxiliary stream behind the level-0 launch (dr3hip_batch_build_ahead)
static int batch_build_impl(dr3hip_batch* b, int first, int count, int with_deriv, bool ahead)
{
    if (!b || first < 0 || count < 0 || first + count > b->capacity)
        return b ? fail(b->ctx, DR3HIP_EINVAL, "batch_build: slots [%d,%d) outside capacity %d", first, first + count, b->capacity) : DR3HIP_EINVAL;
    dr3hip_ctx* ctx = b->ctx;
    if (count == 0) return DR3HIP_OK;
    DeviceGuard g(ctx->device);
    { const int rc = ahead_join(ctx, b, first, count); if (rc) return rc; }      // slots with auxiliary work pending are about to be rewritten
    const int L = b->levels - 1;
    if (ahead && !ctx->aux) {
        // the LOWEST priority: the auxiliary launches fill what the stream's own kernels leave free and do not compete with them
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
#ifdef DR3_AUX_DEFAULT_PRIORITY
        HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->aux, hipStreamNonBlocking));
#else
        HIPCHK(ctx, hipStreamCreateWithPriority(&ctx->aux, hipStreamNonBlocking, least));
#endif
        HIPCHK(ctx, hipEventCreateWithFlags(&ctx->aux_fork, hipEventDisableTiming));
    }
    bool forked = false;
    // the launches of the pass: the level kernels (chunks of 32768 slots keep tiles * slots far below 2^31 blocks).  The physical
    // REFLECT_101 border of the levels this pass produces (read by the tracker only, never by the pyramid kernels) rides along: the
    // launch of level l >= 1 also writes level l's border (level 0 got its when the frame was uploaded, dr3hip_batch_upload).  Only
    // when the last level has no launch of its own (no derivatives wanted) does its border need one.
    struct Step { int level, s0; };                         // level < 0: the border launch of the last level
    std::vector<Step> steps;
    for (int l = 0; l <= L; ++l) {
        if (l == L && !with_deriv) continue;
        for (int s0 = 0; s0 < count; s0 += 32768) steps.push_back({l, s0});
    }
    for (int s0 = 0; s0 < count && L >= 1 && !with_deriv; s0 += 32768) steps.push_back({-1, s0});
    ProfEvent pe; prof_open(ctx, DR3HIP_PROF_PYR, pe);
    for (size_t i = 0; i < steps.size(); ++i) {
        const int l = steps[i].level, s0 = steps[i].s0, n = count - s0 < 32768 ? count - s0 : 32768;
        LaunchEv ev;
        if (pe.which >= 0) { if (i == 0) ev.start = pe.e0; if (i + 1 == steps.size()) ev.stop = pe.e1; }
        hipStream_t st = ctx->stream;
        if (ahead && l != 0) {                                  // everything behind level 0: the auxiliary stream, once level 0 is through
            if (!forked) {
                HIPCHK(ctx, hipEventRecord(ctx->aux_fork, ctx->stream));
                HIPCHK(ctx, hipStreamWaitEvent(ctx->aux, ctx->aux_fork, 0));
                forked = true;
            }
            st = ctx->aux;
        }
        if (l < 0) {
            HIPCHK(ctx, launch_pad_reflect(&b->lv[L], 1, align_up(b->win_w + 16, 4), first + s0, n, st, &ev));
            continue;
        }
        const bool down = l < L, deriv = with_deriv != 0;
        PyrArgs a;
        a.src = b->lv[l];
        a.dst = down ? b->lv[l + 1] : b->lv[l];
        a.first = first + s0;
        ProfEvent p0; prof_open(ctx, l == 0 ? DR3HIP_PROF_PYR_L0 : -1, p0, ev.start);          // the level-0 launch on its own
        if (p0.which >= 0) {
            if (!ev.start) ev.start = p0.e0;
            if (ev.stop) { ctx->ev_pool.push_back(p0.e1); p0.e1 = ev.stop; p0.own1 = false; } else ev.stop = p0.e1;
        }
        HIPCHK(ctx, launch_pyr_level(a, n, down, deriv, l == 0, st, &ev, l ? align_up(b->win_w + 16, 4) : 0));
        prof_close(ctx, p0);
    }
    prof_close(ctx, pe);
    if (forked) {                                               // consumers of these slots on the context's stream wait for this event
        dr3hip_ctx::Ahead e; e.b = b; e.first = first; e.count = count; e.done = nullptr;
        if (!ctx->ev_pool.empty()) { e.done = ctx->ev_pool.back(); ctx->ev_pool.pop_back(); }
        else HIPCHK(ctx, hipEventCreate(&e.done));
        HIPCHK(ctx, hipEventRecord(e.done, ctx->aux));
        ctx->ahead.push_back(e);
        if (ctx->ahead.size() > 8) { const int rc = ahead_join(ctx, nullptr, 0, -1); if (rc) return rc; }
    }
    return DR3HIP_OK;
}

int dr3hip_batch_build(dr3hip_batch* b, int first, int count, int with_deriv) { return batch_build_impl(b, first, count, with_deriv, false); }
int dr3hip_batch_build_ahead(dr3hip_batch* b, int first, int count, int with_deriv) { return batch_build_impl(b, first, count, with_deriv, true); }
int dr3hip_batch_join(dr3hip_batch* b)
{
    if (!b) return DR3HIP_EINVAL;
    DeviceGuard g(b->ctx->device);
    return ahead_join(b->ctx, b, 0, -1);
}

static int fill_lk_args(dr3hip_batch* b, const dr3hip_lk_params* p, LKArgs& a)
{
    dr3hip_ctx* ctx = b->ctx;
    if (p->win_w != b->win_w || p->win_h != b->win_h)
        return fail(ctx, DR3HIP_EINVAL, "track: window %dx%d differs from the batch's %dx%d", p->win_w, p->win_h, b->win_w, b->win_h);
    const int L = dr3hip_lk_effective_max_level(b->w, b->h, p->win_w, p->win_h, p->max_level);
    if (L + 1 > b->levels)
        return fail(ctx, DR3HIP_EINVAL, "track: needs %d levels, batch holds %d", L + 1, b->levels);
    memset(&a, 0, sizeof(a));
    for (int l = 0; l <= L; ++l) a.lv[l] = b->lv[l];
    a.max_level = L;
    a.win_w = p->win_w; a.win_h = p->win_h;
    // TermCriteria clamping of SparsePyrLKOpticalFlowImpl::calc
    a.max_count = (p->crit_type & DR3HIP_CRIT_COUNT) ? (p->max_iters < 0 ? 0 : p->max_iters > 100 ? 100 : p->max_iters) : 30;
    double eps = (p->crit_type & DR3HIP_CRIT_EPS) ? (p->eps < 0. ? 0. : p->eps > 10. ? 10. : p->eps) : 0.01;
    a.epsilon = eps * eps;
    a.flags = p->flags;
    a.min_eig = p->min_eig_threshold;
    return L;
}

int dr3hip_batch_track(dr3hip_batch* b, int n_pairs, int first_slot,
                       const int* d_prev_slot, const int* d_next_slot,
                       const float* d_prev_xy, float* d_next_xy, int n_pts,
                       uint8_t* d_status, float* d_err, int* d_iters, const dr3hip_lk_params* p)
{
    return dr3hip_batch_track_diag(b, n_pairs, first_slot, d_prev_slot, d_next_slot, d_prev_xy, d_next_xy, n_pts, d_status, d_err,
                                   d_iters, nullptr, p);
}

int dr3hip_batch_track_diag(dr3hip_batch* b, int n_pairs, int first_slot,
                            const int* d_prev_slot, const int* d_next_slot,
                            const float* d_prev_xy, float* d_next_xy, int n_pts,
                            uint8_t* d_status, float* d_err, int* d_iters, uint8_t* d_reason, const dr3hip_lk_params* p)
{
    if (!b) return DR3HIP_EINVAL;
    dr3hip_ctx* ctx = b->ctx;
    if (!p || n_pairs < 0 || n_pts < 0 || !d_prev_xy || !d_next_xy || !d_status || !d_err)
        return fail(ctx, DR3HIP_EINVAL, "batch_track: bad arguments");
    if (!d_prev_slot != !d_next_slot) return fail(ctx, DR3HIP_EINVAL, "batch_track: give both slot arrays or neither");
    if (!d_prev_slot && (first_slot < 0 || first_slot + n_pairs + 1 > b->capacity))
        return fail(ctx, DR3HIP_EINVAL, "batch_track: pairs [%d,%d] outside capacity %d", first_slot, first_slot + n_pairs, b->capacity);
    LKArgs a;
    const int L = fill_lk_args(b, p, a);
    if (L < 0) return L;
    if (n_pairs == 0 || n_pts == 0) return L;
    DeviceGuard g(ctx->device);
    // slots built ahead: the tracker waits for those it reads (all of the batch's when the pairs come as device arrays)
    { const int rc = d_prev_slot ? ahead_join(ctx, b, 0, -1) : ahead_join(ctx, b, first_slot, n_pairs + 1); if (rc) return rc; }
    ProfEvent pe; prof_open(ctx, DR3HIP_PROF_LK, pe);
    for (int p0 = 0; p0 < n_pairs; p0 += 32768) {              // grid.y limit
        LaunchEv ev;
        if (pe.which >= 0) { if (p0 == 0) ev.start = pe.e0; if (p0 + 32768 >= n_pairs) ev.stop = pe.e1; }
        const int n = n_pairs - p0 < 32768 ? n_pairs - p0 : 32768;
        const size_t off = (size_t)p0 * n_pts;
        a.n_pts = n_pts;
        a.first_slot = first_slot + p0;
        a.prev_slot = d_prev_slot ? d_prev_slot + p0 : nullptr;
        a.next_slot = d_next_slot ? d_next_slot + p0 : nullptr;
        a.prev_xy = d_prev_xy + 2 * off; a.next_xy = d_next_xy + 2 * off;
        a.status = d_status + off; a.err = d_err + off;
        a.iters = d_iters ? d_iters + off * (L + 1) : nullptr;
        a.reason = d_reason ? d_reason + off : nullptr;
        HIPCHK(ctx, launch_lk_track(a, n, ctx->stream, &ev));
    }
    prof_close(ctx, pe);
    return L;
}

int dr3hip_lk_stats(dr3hip_ctx* ctx, long long n_points, const uint8_t* d_status, const uint8_t* d_reason,
                    const int* d_iters, int levels, dr3hip_lk_counters* out)
{
    if (!ctx) return DR3HIP_EINVAL;
    if (!out || n_points < 0 || !d_status || (d_iters && (levels < 1 || levels > DR3HIP_MAX_LEVELS)))
        return fail(ctx, DR3HIP_EINVAL, "lk_stats: bad arguments");
    memset(out, 0, sizeof(*out));
    out->n_points = n_points;
    if (n_points == 0) return DR3HIP_OK;
    DeviceGuard g(ctx->device);
    int rc;
    const size_t words = sizeof(dr3hip_lk_counters) / sizeof(unsigned long long);
    if ((rc = stage_reserve(ctx, 3, words * sizeof(unsigned long long)))) return rc;
    unsigned long long* d_cnt = (unsigned long long*)ctx->stage[3].p;
    HIPCHK(ctx, hipMemsetAsync(d_cnt, 0, words * sizeof(unsigned long long), ctx->stream));
    HIPCHK(ctx, launch_lk_stats(n_points, d_status, d_reason, d_iters, d_iters ? levels : 0, d_cnt, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(out, d_cnt, sizeof(*out), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    out->n_points = n_points;
    return DR3HIP_OK;
}

// ---- sparse image alignment (extension) -----------------------------------------------------------------
int dr3hip_batch_align(dr3hip_batch* b, int n_pairs, int first_slot, const int* d_ref_slot, const int* d_cur_slot,
                       const float* d_xyz, int n_pts, const dr3hip_align_params* p, double* d_T,
                       int* d_n_meas, float* d_chi2, int* d_iters)
{
    if (!b) return DR3HIP_EINVAL;
    dr3hip_ctx* ctx = b->ctx;
    if (!p || !d_xyz || !d_T || n_pairs < 0 || n_pts < 0 || n_pts > (1 << 22)) return fail(ctx, DR3HIP_EINVAL, "batch_align: bad arguments");
    if (!d_ref_slot != !d_cur_slot) return fail(ctx, DR3HIP_EINVAL, "batch_align: give both slot arrays or neither");
    if (!d_ref_slot && (first_slot < 0 || first_slot + n_pairs + 1 > b->capacity))
        return fail(ctx, DR3HIP_EINVAL, "batch_align: pairs [%d,%d] outside capacity %d", first_slot, first_slot + n_pairs, b->capacity);
    if (p->min_level < 0 || p->max_level < p->min_level || p->max_level >= b->levels || p->n_iter < 0)
        return fail(ctx, DR3HIP_EINVAL, "batch_align: levels %d..%d not in the batch's pyramid (%d levels) or n_iter < 0", p->min_level, p->max_level, b->levels);
    if (n_pairs == 0) return DR3HIP_OK;
    DeviceGuard g(ctx->device);
    { const int rc = ahead_join(ctx, b, 0, -1); if (rc) return rc; }
    const size_t per_pair = (size_t)((n_pts + 255) / 256) * 16 * 7 * 256 * sizeof(float);
    int rc;
    if ((rc = stage_reserve(ctx, 8, per_pair * (size_t)n_pairs + 256))) return rc;
    AlignArgs a;
    memset(&a, 0, sizeof(a));
    for (int l = 0; l < b->levels; ++l) a.lv[l] = b->lv[l];
    a.fx = p->fx; a.fy = p->fy; a.cx = p->cx; a.cy = p->cy;
    a.max_level = p->max_level; a.min_level = p->min_level; a.n_iter = p->n_iter; a.eps = p->eps;
    a.n_pts = n_pts; a.first_slot = first_slot; a.ref_slot = d_ref_slot; a.cur_slot = d_cur_slot;
    a.xyz = d_xyz; a.cache = (float*)ctx->stage[8].p; a.T = d_T; a.n_meas = d_n_meas; a.chi2 = d_chi2; a.iters = d_iters;
    ProfEvent pe; prof_begin(ctx, DR3HIP_PROF_ALIGN, pe);
    HIPCHK(ctx, launch_sparse_align(a, n_pairs, ctx->stream));
    prof_end(ctx, pe);
    return DR3HIP_OK;
}

// ---- host-buffer LK -------------------------------------------------------------------------------
int dr3hip_lk_track(dr3hip_ctx* ctx, const uint8_t* prev, const uint8_t* next, int w, int h,
                    int prev_stride, int next_stride, const float* prev_xy, float* next_xy, int n,
                    uint8_t* status, float* err, const dr3hip_lk_params* p)
{
    if (!ctx) return DR3HIP_EINVAL;
    if (!prev || !next || !p || n < 0 || (n && (!prev_xy || !next_xy || !status || !err)) || prev_stride < w || next_stride < w)
        return fail(ctx, DR3HIP_EINVAL, "lk_track: bad arguments");
    DeviceGuard g(ctx->device);
    dr3hip_batch* b = ctx->lk_batch;
    const int L = dr3hip_lk_effective_max_level(w, h, p->win_w, p->win_h, p->max_level);
    if (!b || b->w != w || b->h != h || b->win_w != p->win_w || b->win_h != p->win_h || b->levels != L + 1) {
        if (b) dr3hip_batch_destroy(b);
        ctx->lk_batch = nullptr;
        const int rc = dr3hip_batch_create(ctx, w, h, p->win_w, p->win_h, p->max_level, 2, &b);
        if (rc != DR3HIP_OK) return rc;
        ctx->lk_batch = b;
    }
    // One call of the reference (src/initialization.cpp:608) is one frame pair: what it costs here is launches, copies and host waits, not
    // arithmetic.  Both frames cross PCIe into one bounce buffer and get their borders in ONE launch, both pyramids are ONE pass of four
    // launches (the second frame's derivatives ride along: one frame of them is cheaper than a launch), the points go in and the
    // results come out as ONE copy each through a pinned buffer, and the host waits once, at the end.
    int rc;
    const LevelDev& v = b->lv[0];
    const size_t fb0 = (size_t)prev_stride * (h - 1) + w, fb1 = (size_t)next_stride * (h - 1) + w;
    const size_t fstep = (size_t)align_up((long long)(fb0 > fb1 ? fb0 : fb1) + 64, 256ll);
    if ((rc = stage_reserve(ctx, 7, 2 * fstep))) return rc;
    uint8_t* bounce = (uint8_t*)ctx->stage[7].p;
    HIPCHK(ctx, hipMemcpyAsync(bounce, prev, fb0, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(bounce + fstep, next, fb1, hipMemcpyHostToDevice, ctx->stream));
    const int fill = align_up(b->win_w + 16, 4);
    if (prev_stride == next_stride) {
        HIPCHK(ctx, launch_upload_level0(bounce, (size_t)prev_stride, v, 0, fill, ctx->stream, 2, fstep));
    } else {
        HIPCHK(ctx, launch_upload_level0(bounce, (size_t)prev_stride, v, 0, fill, ctx->stream, 1, 0));
        HIPCHK(ctx, launch_upload_level0(bounce + fstep, (size_t)next_stride, v, 1, fill, ctx->stream, 1, 0));
    }
    if ((rc = dr3hip_batch_build(b, 0, 2, 1)) != DR3HIP_OK) return rc;
    if (n == 0) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); return L; }

    // device: [prev_xy 8 n][next_xy 8 n][err 4 n][status n]; the pinned buffer mirrors [prev_xy][next_xy] on the way in and
    // [next_xy][err][status] on the way out
    const size_t xyb = (size_t)n * 2 * sizeof(float), outb = xyb + (size_t)n * sizeof(float) + (size_t)n;
    if ((rc = stage_reserve(ctx, 0, xyb + outb))) return rc;
    if (ctx->pin_io_cap < 2 * xyb + outb) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->pin_io) { (void)hipHostFree(ctx->pin_io); ctx->pin_io = nullptr; ctx->pin_io_cap = 0; }
        const size_t cap = 2 * (2 * xyb + outb) + 4096;
        if (hipHostMalloc(&ctx->pin_io, cap, hipHostMallocDefault) != hipSuccess) { ctx->pin_io = nullptr; return fail(ctx, DR3HIP_ENOMEM, "hipHostMalloc(%zu) failed", cap); }
        ctx->pin_io_cap = cap;
    }
    float* d_prev = (float*)ctx->stage[0].p; float* d_next = d_prev + 2 * (size_t)n;
    float* d_err = d_next + 2 * (size_t)n; uint8_t* d_status = (uint8_t*)(d_err + n);
    uint8_t* pin = (uint8_t*)ctx->pin_io;
    const bool seeded = (p->flags & DR3HIP_LK_USE_INITIAL_FLOW) != 0;
    std::memcpy(pin, prev_xy, xyb);
    // without USE_INITIAL_FLOW OpenCV never reads nextPts; seed with prevPts so the upload is defined
    std::memcpy(pin + xyb, seeded ? next_xy : prev_xy, xyb);
    HIPCHK(ctx, hipMemcpyAsync(d_prev, pin, 2 * xyb, hipMemcpyHostToDevice, ctx->stream));
    rc = dr3hip_batch_track(b, 1, 0, nullptr, nullptr, d_prev, d_next, n, d_status, d_err, nullptr, p);
    if (rc < 0) return rc;
    uint8_t* pout = pin + 2 * xyb;
    HIPCHK(ctx, hipMemcpyAsync(pout, d_next, outb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    std::memcpy(next_xy, pout, xyb);
    std::memcpy(err, pout + xyb, (size_t)n * sizeof(float));
    std::memcpy(status, pout + xyb + (size_t)n * sizeof(float), (size_t)n);
    return rc;
}

// ---- half-sample pyramid ------------------------------------------------------------------------------
static int resolve_half_mode(int mode, int w)
{
    if (mode == DR3HIP_HALF_REF_X86) return (w % 16 == 0) ? DR3HIP_HALF_SSE2 : DR3HIP_HALF_REF_SCALAR;
    return mode;
}

int dr3hip_halfsample_dev(dr3hip_ctx* ctx, const uint8_t* d_src, int w, int h, int sstride,
                          uint8_t* d_dst, int dstride, int mode)
{
    if (!ctx) return DR3HIP_EINVAL;
    if (!d_src || !d_dst || w < 2 || h < 2 || sstride < w || dstride < w / 2 || mode < 0 || mode > 3)
        return fail(ctx, DR3HIP_EINVAL, "halfsample: bad arguments");
    DeviceGuard g(ctx->device);
    HalfArgs a;
    a.src = d_src; a.w = w; a.h = h; a.sstride = sstride;
    a.dst = d_dst; a.ow = w / 2; a.oh = h / 2; a.dstride = dstride;
    a.mode = resolve_half_mode(mode, w);
    a.lstride = sstride; a.n_slots = 1; a.sstep = 0; a.dstep = 0;
    ProfEvent pe; prof_begin(ctx, DR3HIP_PROF_HALF, pe);
    HIPCHK(ctx, launch_halfsample(a, ctx->stream));
    prof_end(ctx, pe);
    return DR3HIP_OK;
}

int dr3hip_halfsample_pyramid(dr3hip_ctx* ctx, const uint8_t* img, int w, int h, int stride,
                              int n_levels, int mode, uint8_t* const* out_levels, const int* out_strides)
{
    if (!ctx) return DR3HIP_EINVAL;
    if (!img || w < 1 || h < 1 || stride < w || n_levels < 1 || n_levels > 16 || (n_levels > 1 && (!out_levels || !out_strides)))
        return fail(ctx, DR3HIP_EINVAL, "halfsample_pyramid: bad arguments");
    if (n_levels == 1) return DR3HIP_OK;
    DeviceGuard g(ctx->device);
    // device layout: level l tightly pitched at align(w_l, 64); REF_SCALAR needs stride == the
    // caller's stride semantics, so level 0 keeps the caller's stride when it is the walk's input
    int rc;
    std::vector<int> lw(n_levels), lh(n_levels), lp(n_levels);
    std::vector<size_t> off(n_levels);
    size_t total = 0;
    lw[0] = w; lh[0] = h;
    for (int l = 0; l < n_levels; ++l) {
        if (l) { lw[l] = lw[l - 1] / 2; lh[l] = lh[l - 1] / 2; }
        // the reference's scalar walk depends on the source stride: cv::Mat levels are continuous
        // (stride == cols) and level 0 has the caller's stride; mirror both.
        lp[l] = (l == 0) ? stride : lw[l];
        off[l] = total;
        total += (size_t)align_up((long long)lp[l] * (lh[l] > 0 ? lh[l] : 1) + 64, 256ll);
    }
    if ((rc = stage_reserve(ctx, 4, total))) return rc;
    uint8_t* base = (uint8_t*)ctx->stage[4].p;
    HIPCHK(ctx, hipMemcpyAsync(base + off[0], img, (size_t)stride * (h - 1) + w, hipMemcpyHostToDevice, ctx->stream));
    for (int l = 1; l < n_levels; ++l) {
        if (lw[l] < 1 || lh[l] < 1) break;
        if ((rc = dr3hip_halfsample_dev(ctx, base + off[l - 1], lw[l - 1], lh[l - 1], lp[l - 1], base + off[l], lp[l], mode))) return rc;
        if (!out_levels[l] || out_strides[l] < lw[l]) return fail(ctx, DR3HIP_EINVAL, "halfsample_pyramid: out level %d missing or stride too small", l);
        { const int rc2_ = copy2d_d2h(ctx, out_levels[l], out_strides[l], base + off[l], lp[l], lw[l], lh[l]); if (rc2_) return rc2_; }
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return DR3HIP_OK;
}

// ---- warp -----------------------------------------------------------------------------------------
// cv::invert for a 3x3 CV_64F matrix (closed-form adjugate); returns false when singular
static bool invert3x3(const double S[9], double T[9])
{
    const double d0 = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) + S[2] * (S[3] * S[7] - S[4] * S[6]);
    if (d0 == 0.) { for (int i = 0; i < 9; ++i) T[i] = 0.; return false; }
    const double d = 1. / d0;
    double t[9];
    t[0] = (S[4] * S[8] - S[5] * S[7]) * d; t[1] = (S[2] * S[7] - S[1] * S[8]) * d; t[2] = (S[1] * S[5] - S[2] * S[4]) * d;
    t[3] = (S[5] * S[6] - S[3] * S[8]) * d; t[4] = (S[0] * S[8] - S[2] * S[6]) * d; t[5] = (S[2] * S[3] - S[0] * S[5]) * d;
    t[6] = (S[3] * S[7] - S[4] * S[6]) * d; t[7] = (S[1] * S[6] - S[0] * S[7]) * d; t[8] = (S[0] * S[4] - S[1] * S[3]) * d;
    for (int i = 0; i < 9; ++i) T[i] = t[i];
    return true;
}

// the destination pixels a launch evaluates: columns [x0, x1), rows [y0, y1) -- the whole destination for a warp, the bounding box of
// the pasted image for Panorama::add_img_to_canvas (whose destination is the whole canvas, of which it walks the box columns only)
struct DstRect { int x0, y0, x1, y1; };
static inline double rect_x(const DstRect& r, int a) { return (double)(a ? r.x1 - 1 : r.x0); }
static inline double rect_y(const DstRect& r, int b) { return (double)(b ? r.y1 - 1 : r.y0); }
static inline double rect_wmin(const double M[9], const DstRect& r)
{
    double wmin = INFINITY;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) wmin = std::fmin(wmin, std::fabs(M[6] * rect_x(r, a) + M[7] * rect_y(r, b) + M[8]));
    return wmin;
}

// W = M6 x + M7 y + M8 is linear, so its extremes over the rectangle are at the corners: of one sign there, |W| >= 1e-6 max|W| and
// within [1e-100, 1e100] -- the rectangle then maps onto a convex quadrilateral whose corners bound every pixel
static bool w_of_one_sign(const double M[9], const DstRect& r)
{
    if (r.x1 <= r.x0 || r.y1 <= r.y0) return false;
    for (int i = 0; i < 9; ++i) if (!std::isfinite(M[i])) return false;
    double wmin = INFINITY, wmax = 0.0; int pos = 0, neg = 0;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) {
        const double w = M[6] * rect_x(r, a) + M[7] * rect_y(r, b) + M[8];
        (w > 0 ? pos : neg)++;
        wmin = std::fmin(wmin, std::fabs(w)); wmax = std::fmax(wmax, std::fabs(w));
    }
    if (pos != 4 && neg != 4) return false;
    return wmin >= 1e-100 && wmax <= 1e100 && !(wmin < 1e-6 * wmax);
}

// May the kernel use its guarded fast coordinate path (warp_kernels.hip, persp_tap_fast)?  Conditions: source <= 8191 x 8191 and
// all source coordinates of the rectangle within +-8000 px (13 extra fraction bits fit int32), every M finite, W of one sign
// (w_of_one_sign), and numerator operands bounded by 2^32 |W|min: then the FMA/Newton evaluation differs
// from the exact f64 sequence by < 2^-52 * 2^32 * 2^18 = 1/4 unit of the 2-unit tie guard.
static bool warp_fast_coords_ok(const double M[9], int sw, int sh, const DstRect& r)
{
    if (sw > 8191 || sh > 8191 || !w_of_one_sign(M, r)) return false;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) {
        const double x = rect_x(r, a), y = rect_y(r, b), w = M[6] * x + M[7] * y + M[8];
        const double sx = (M[0] * x + M[1] * y + M[2]) / w, sy = (M[3] * x + M[4] * y + M[5]) / w;
        if (!(std::fabs(sx) < 8000.0) || !(std::fabs(sy) < 8000.0)) return false;
    }
    const double wmin = rect_wmin(M, r);
    const double ax = std::fmax(std::fabs((double)r.x0), std::fabs((double)r.x1)), ay = std::fmax(std::fabs((double)r.y0), std::fabs((double)r.y1));
    const double bx = std::fabs(M[0]) * ax + std::fabs(M[1]) * ay + std::fabs(M[2]);
    const double by = std::fabs(M[3]) * ax + std::fabs(M[4]) * ay + std::fabs(M[5]);
    return bx < 4294967296.0 * wmin && by < 4294967296.0 * wmin;
}

// May the 4-pixel lanes derive pixels 1..3 from pixel 0 in f32 (warp_kernels.hip, persp_taps4_delta)?  On top of the conditions
// above: q = M6 / W at most 1.2e-4 in magnitude and the source moves by at most 2 px per destination pixel along x, over the
// whole rectangle (both are largest where |W| is smallest; source coordinates are bounded by 8000 px there already).
static bool warp_delta_coords_ok(const double M[9], const DstRect& r)
{
    const double wmin = rect_wmin(M, r);
    const double q = std::fabs(M[6]) / wmin;
    const double dxdx = (std::fabs(M[0]) / wmin) + q * 8000.0, dydx = (std::fabs(M[3]) / wmin) + q * 8000.0;
    return q <= 1.2e-4 && dxdx <= 2.0 && dydx <= 2.0;
}

// May the tile kernel derive the 16 pixels of a lane's destination column from ONE f64 anchor at its centre (warp_kernels.hip,
// warp_tile_8uc3_kernel)?  On top of warp_fast_coords_ok: |p| <= 2e-4 (p = M7 / W) and the source moves by at most 2 px per
// destination pixel along y, over the whole rectangle (both largest where |W| is smallest; source coordinates are bounded by
// 8000 px there already).  The kernel's error budget is written for exactly these bounds.
static bool tile_delta_coords_ok(const double M[9], const DstRect& r)
{
    const double wmin = rect_wmin(M, r);
    const double p = std::fabs(M[7]) / wmin;
    const double dxdy = std::fabs(M[1]) / wmin + p * 8000.0, dydy = std::fabs(M[4]) / wmin + p * 8000.0;
    return p <= 2e-4 && dxdy <= 2.0 && dydy <= 2.0;
}

// 0: OpenCV's own f64 sequence per pixel; 1: the guarded fast coordinates; 2: + pixels 1..3 of a 4-pixel lane from pixel 0; 3: + one
// anchor per strip column (WarpJobDev::fast)
static int coords_level(const double M[9], int sw, int sh, const DstRect& r)
{
    return warp_fast_coords_ok(M, sw, sh, r) ? (warp_delta_coords_ok(M, r) ? (tile_delta_coords_ok(M, r) ? 3 : 2) : 1) : 0;
}

// May an 8UC3 job run on the LDS-staged strip kernel (warp_tile_8uc3_kernel), and how much LDS does a wave's 64 x 16 strip need?
// Rows 16-byte aligned, 32-bit offsets, a power-of-two block split, W of one sign (a strip then maps onto a convex quadrilateral)
// and a nearly affine map (fast == 3).  The LDS figure is an ESTIMATE: the source boxes (as the kernel forms them: corner taps,
// one pixel of margin, rows cut to multiples of four pixels) of nine strips -- the rectangle's corners, edge centres and
// centre -- plus one row and one chunk of slack; the box of a projective map grows monotonically towards one side, so this is its
// maximum in practice, and a strip that does need more takes the kernel's slow way out (correct, not timed).  Returns 0 when the
// job cannot run there.  The strips are those the kernel cuts: 64 x 16 pixels from the rectangle's origin.
static int tile_kernel_lds(const WarpJobDev& j, const double M[9], const DstRect& r)
{
    if (j.fast != 3 || j.bw0_shift < 0) return 0;
    if ((((uintptr_t)j.src | (uintptr_t)j.sstride) & 15) || j.sstride >= (1 << 24) || j.sh >= (1 << 24) || (long long)j.sstride * j.sh >= (1LL << 31)) return 0;
    const int tiles_x = (r.x1 - r.x0 + kWarpTileW - 1) / kWarpTileW, strips_y = (r.y1 - r.y0 + kWarpStripRows - 1) / kWarpStripRows;
    long long worst = 0;
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) {
        const int tx = a == 0 ? 0 : a == 1 ? tiles_x / 2 : tiles_x - 1, sy = b == 0 ? 0 : b == 1 ? strips_y / 2 : strips_y - 1;
        const int xa = r.x0 + tx * kWarpTileW, xb = std::min(xa + kWarpTileW, r.x1) - 1, ya = r.y0 + sy * kWarpStripRows, yb = std::min(ya + kWarpStripRows, r.y1) - 1;
        long long x_lo = LLONG_MAX, x_hi = LLONG_MIN, y_lo = LLONG_MAX, y_hi = LLONG_MIN;
        for (int c = 0; c < 4; ++c) {
            const double x = (c & 1) ? xb : xa, y = (c & 2) ? yb : ya, w = M[6] * x + M[7] * y + M[8];
            const long long sx = (long long)std::floor(((M[0] * x + M[1] * y + M[2]) / w) + 1.0 / 64), sy2 = (long long)std::floor(((M[3] * x + M[4] * y + M[5]) / w) + 1.0 / 64);
            x_lo = std::min(x_lo, sx); x_hi = std::max(x_hi, sx); y_lo = std::min(y_lo, sy2); y_hi = std::max(y_hi, sy2);
        }
        x_lo -= 1; x_hi += 2; y_lo -= 1; y_hi += 2;
        const long long ox = x_lo & ~3LL, cpr = ((x_hi - ox + 4) >> 2) + 1, rows = y_hi - y_lo + 2;
        if (cpr > 64) return 0;
        worst = std::max(worst, rows * cpr * 16);
    }
    return worst <= (long long)kWarpTileLdsBytes ? (int)worst : 0;
}

// The tie guard of the strip kernel (units: 2^-13 of 1/32 px): its coordinates come out of f32 arithmetic on offsets of at most
// T = 7.5 rows x the largest source step per destination row, so they carry the rounding of the step (1 ulp of T), of three fused
// multiply-adds (half an ulp of T each), the dropped cubic term T (7.5 p)^3 and the conversion to an integer (0.5).  Guard 1 if that
// stays under 0.95, else 2 (tile_delta_coords_ok bounds it by 1.75).
static int tile_kernel_guard(const double M[9], const DstRect& r)
{
    const double wmin = rect_wmin(M, r);
    const double p = std::fabs(M[7]) / wmin;
    const double step = std::fmax(std::fabs(M[1]) / wmin + p * 8000.0, std::fabs(M[4]) / wmin + p * 8000.0);
    const double T = 7.5 * step * 262144.0, e = 7.5 * p;
    const double err = 0.5 + 2.5 * T / 16777216.0 + T * e * e * e;
    return err < 0.95 ? 1 : 2;
}

static bool box_w_one_sign(const double M[9], int x0, int y0, int x1, int y1) { const DstRect r = {x0, y0, x1, y1}; return w_of_one_sign(M, r); }

// the coordinate form / strip plan of a job whose launch evaluates the pixels [x0, x1) x [y0, y1) of its destination only
static void plan_rect(WarpJobDev& j, int x0, int y0, int x1, int y1)
{
    const DstRect r = {x0, y0, x1, y1};
    j.fast = (x1 > x0 && y1 > y0) ? coords_level(j.M, j.sw, j.sh, r) : 0;
    j.tile_lds = (x1 > x0 && y1 > y0) ? tile_kernel_lds(j, j.M, r) : 0;
    j.tile_guard = j.tile_lds ? tile_kernel_guard(j.M, r) : 2;
    j.tile = j.tile_lds ? 2 : 0;
}

static void make_job(WarpJobDev& j, const void* src, int sw, int sh, int sstride, void* dst, int dw, int dh, int dstride,
                     const float M[9], int flags)
{
    j.src = (const uint8_t*)src; j.sw = sw; j.sh = sh; j.sstride = sstride;
    j.dst = (uint8_t*)dst; j.dw = dw; j.dh = dh; j.dstride = dstride;
    double Md[9];
    for (int i = 0; i < 9; ++i) Md[i] = (double)M[i];                     // M0.convertTo(matM, CV_64F)
    if (!(flags & DR3HIP_WARP_INVERSE_MAP)) invert3x3(Md, Md);            // invert(matM, matM)
    for (int i = 0; i < 9; ++i) j.M[i] = Md[i];
    // WarpPerspectiveInvoker block geometry (only the x0 + x1 split is observable)
    const int bh0 = dh < 16 ? dh : 16;
    j.bw0 = (1024 / bh0) < dw ? (1024 / bh0) : dw;
    j.bw0_shift = -1;
    if (j.bw0 >= dw) j.bw0_shift = 31;                                    // x0 == 0 for every x
    else if (j.bw0 >= 64 && (j.bw0 & (j.bw0 - 1)) == 0) j.bw0_shift = __builtin_ctz((unsigned)j.bw0);
    const DstRect whole = {0, 0, dw, dh};
    j.fast = (dw > 0 && dh > 0) ? coords_level(Md, sw, sh, whole) : 0;
    j.tile_lds = tile_kernel_lds(j, Md, whole);
    j.tile_guard = j.tile_lds ? tile_kernel_guard(Md, whole) : 2;
    j.tile = j.tile_lds ? 2 : 0;
}

int dr3hip_warp_plan(const dr3hip_warp_job* q, int type, int flags, int* kernel, int* lds_bytes, int* tie_guard)
{
    if (!q || type_channels(type) == 0 || !side_ok(q->sw) || !side_ok(q->sh) || !side_ok(q->dw) || !side_ok(q->dh)) return DR3HIP_EINVAL;
    WarpJobDev j;
    make_job(j, q->src, q->sw, q->sh, q->sstride, q->dst, q->dw, q->dh, q->dstride, q->M, flags);
    const bool strip = type == DR3HIP_8UC3 && j.tile == 2;
    if (kernel) *kernel = strip ? 2 : 0;
    if (lds_bytes) *lds_bytes = strip ? j.tile_lds : 0;
    if (tie_guard) *tie_guard = strip ? j.tile_guard : 0;
    return DR3HIP_OK;
}

int dr3hip_warp_perspective_batch(dr3hip_ctx* ctx, const dr3hip_warp_job* jobs, int n_jobs, int type, int flags)
{
    if (!ctx) return DR3HIP_EINVAL;
    if (!jobs || n_jobs < 0 || type_channels(type) == 0) return fail(ctx, DR3HIP_EINVAL, "warp_batch: bad arguments");
    if (n_jobs == 0) return DR3HIP_OK;
    if (n_jobs > 65535) return fail(ctx, DR3HIP_ERANGE, "warp_batch: more than 65535 jobs in one call");
    DeviceGuard g(ctx->device);
    const int esz = type_elem_bytes(type);
    std::vector<WarpJobDev> hj(n_jobs);
    for (int i = 0; i < n_jobs; ++i) {
        const dr3hip_warp_job& q = jobs[i];
        if (!q.src || !q.dst || !side_ok(q.sw) || !side_ok(q.sh) || !side_ok(q.dw) || !side_ok(q.dh) || q.sstride < q.sw * esz || q.dstride < q.dw * esz)
            return fail(ctx, DR3HIP_EINVAL, "warp_batch: job %d has bad geometry", i);
        if (esz >= 4 && (((uintptr_t)q.src | (uintptr_t)q.dst | (uintptr_t)q.sstride | (uintptr_t)q.dstride) & (uintptr_t)(esz - 1)))
            return fail(ctx, DR3HIP_EINVAL, "warp_batch: job %d pointers/strides must be %d-byte aligned", i, esz);
        make_job(hj[i], q.src, q.sw, q.sh, q.sstride, q.dst, q.dw, q.dh, q.dstride, q.M, flags);
        if (type != DR3HIP_8UC3) hj[i].tile = 0;
    }
    // 8UC3: the jobs are grouped by the kernel that takes them (LDS-staged tile kernel in its one-anchor form, with per-row
    // coordinates, or the direct gather); every group is one launch over its jobs (a job record carries its own pointers)
    std::stable_sort(hj.begin(), hj.end(), [](const WarpJobDev& a, const WarpJobDev& b) { return a.tile > b.tile; });
    int rc;
    if ((rc = stage_reserve(ctx, 5, sizeof(WarpJobDev) * (size_t)n_jobs))) return rc;
    // pageable source is copied out at enqueue time by hipMemcpyAsync, so hj may die at return
    HIPCHK(ctx, hipMemcpyAsync(ctx->stage[5].p, hj.data(), sizeof(WarpJobDev) * (size_t)n_jobs, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ProfEvent pe; prof_begin(ctx, DR3HIP_PROF_WARP, pe);
    for (int first = 0; first < n_jobs;) {
        int last = first, max_dw = 0, max_dh = 0, lds = 0;
        bool generic = false;
        for (; last < n_jobs && hj[last].tile == hj[first].tile; ++last) {
            max_dw = std::max(max_dw, hj[last].dw); max_dh = std::max(max_dh, hj[last].dh); lds = std::max(lds, hj[last].tile_lds);
            if (hj[last].bw0_shift < 0) generic = true;
        }
        HIPCHK(ctx, launch_warp((const WarpJobDev*)ctx->stage[5].p + first, last - first, max_dw, max_dh, type, generic, ctx->stream, hj[first].tile, lds));
        first = last;
    }
    prof_end(ctx, pe);
    return DR3HIP_OK;
}

int dr3hip_warp_perspective(dr3hip_ctx* ctx, const void* src, int sw, int sh, int sstride, int type,
                            void* dst, int dw, int dh, int dstride, const float M[9], int flags)
{
    if (!ctx) return DR3HIP_EINVAL;
    const int esz = type_elem_bytes(type);
    if (!src || !dst || !M || esz == 0 || !side_ok(sw) || !side_ok(sh) || !side_ok(dw) || !side_ok(dh) || sstride < sw * esz || dstride < dw * esz)
        return fail(ctx, DR3HIP_EINVAL, "warp_perspective: bad arguments");
    DeviceGuard g(ctx->device);
    const int sp = align_up(sw * esz, 64), dp = align_up(dw * esz, 64);
    int rc;
    if ((rc = stage_reserve(ctx, 0, (size_t)sp * sh + 64)) || (rc = stage_reserve(ctx, 1, (size_t)dp * dh + 64))) return rc;
    { const int rc2_ = copy2d_h2d(ctx, ctx->stage[0].p, sp, src, sstride, (size_t)sw * esz, sh); if (rc2_) return rc2_; }
    dr3hip_warp_job j;
    j.src = ctx->stage[0].p; j.sw = sw; j.sh = sh; j.sstride = sp;
    j.dst = ctx->stage[1].p; j.dw = dw; j.dh = dh; j.dstride = dp;
    memcpy(j.M, M, sizeof(j.M));
    if ((rc = dr3hip_warp_perspective_batch(ctx, &j, 1, type, flags))) return rc;
    { const int rc2_ = copy2d_d2h(ctx, dst, dstride, ctx->stage[1].p, dp, (size_t)dw * esz, dh); if (rc2_) return rc2_; }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return DR3HIP_OK;
}

int dr3hip_remap_bilinear_dev(dr3hip_ctx* ctx, const void* d_src, int sw, int sh, int sstride, int type,
                              void* d_dst, int dw, int dh, int dstride,
                              const float* d_mapx, const float* d_mapy, int map_stride)
{
    if (!ctx) return DR3HIP_EINVAL;
    const int esz = type_elem_bytes(type);
    if (!d_src || !d_dst || !d_mapx || !d_mapy || esz == 0 || !side_ok(sw) || !side_ok(sh) || !side_ok(dw) || !side_ok(dh) ||
        sstride < sw * esz || dstride < dw * esz || map_stride < dw * 4 || (map_stride & 3))
        return fail(ctx, DR3HIP_EINVAL, "remap: bad arguments");
    if (esz >= 4 && (((uintptr_t)d_src | (uintptr_t)d_dst | (uintptr_t)sstride | (uintptr_t)dstride) & (uintptr_t)(esz - 1)))
        return fail(ctx, DR3HIP_EINVAL, "remap: pointers/strides must be %d-byte aligned", esz);
    DeviceGuard g(ctx->device);
    RemapArgs a;
    a.src = (const uint8_t*)d_src; a.sw = sw; a.sh = sh; a.sstride = sstride;
    a.dst = (uint8_t*)d_dst; a.dw = dw; a.dh = dh; a.dstride = dstride;
    a.mapx = d_mapx; a.mapy = d_mapy; a.map_stride = map_stride;
    a.mode = 0; a.tx0 = a.tx1 = a.ty0 = a.ty1 = nullptr; a.f = a.half_w = a.half_h = 0.f;
    ProfEvent pe; prof_begin(ctx, DR3HIP_PROF_REMAP, pe);
    HIPCHK(ctx, launch_remap(a, type, ctx->stream));
    prof_end(ctx, pe);
    return DR3HIP_OK;
}

// utils::warp_spherical / warp_cylindrical: tables on the host, everything per-pixel in remap_kernel<TYPE, kind>
int dr3hip_warp_local_dev(dr3hip_ctx* ctx, const void* d_src, int w, int h, int sstride, int type,
                          void* d_dst, int dstride, int kind, float f)
{
    if (!ctx) return DR3HIP_EINVAL;
    const int esz = type_elem_bytes(type);
    if (!d_src || !d_dst || esz == 0 || !side_ok(w) || !side_ok(h) || sstride < w * esz || dstride < w * esz ||
        (kind != DR3HIP_WARP_SPHERICAL && kind != DR3HIP_WARP_CYLINDRICAL))
        return fail(ctx, DR3HIP_EINVAL, "warp_local: bad arguments");
    if (esz >= 4 && (((uintptr_t)d_src | (uintptr_t)d_dst | (uintptr_t)sstride | (uintptr_t)dstride) & (uintptr_t)(esz - 1)))
        return fail(ctx, DR3HIP_EINVAL, "warp_local: pointers/strides must be %d-byte aligned", esz);
    // per-column / per-row tables, reference src/utils.cpp:151-176 (spherical), :229-254 (cylindrical):
    //   xf = (float)((j - 0.5 w) / f) with the subtraction and the division in double (0.5 * w is a double), then
    //   the unqualified sin / cos of that TU on the float, i.e. ::sin(double)
    std::vector<double> tab((size_t)2 * w + (size_t)2 * h);
    double* tx0 = tab.data(); double* tx1 = tx0 + w; double* ty0 = tx1 + w; double* ty1 = ty0 + h;
    for (int j = 0; j < w; ++j) {
        const float xf = (float)(((double)(float)j - 0.5 * w) / (double)f);
        if (kind == DR3HIP_WARP_SPHERICAL) { tx0[j] = std::sin((double)xf); tx1[j] = std::cos((double)xf); }
        else { tx0[j] = (double)(float)std::sin((double)xf); tx1[j] = (double)(float)std::cos((double)xf); }
    }
    for (int i = 0; i < h; ++i) {
        const float yf = (float)(((double)(float)i - 0.5 * h) / (double)f);
        if (kind == DR3HIP_WARP_SPHERICAL) { ty0[i] = std::sin((double)yf); ty1[i] = std::cos((double)yf); }
        else { ty0[i] = (double)yf; ty1[i] = 0.0; }
    }
    DeviceGuard g(ctx->device);
    int rc;
    if ((rc = stage_reserve(ctx, 6, tab.size() * sizeof(double)))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->stage[6].p, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));                       // tab is a local
    RemapArgs a;
    a.src = (const uint8_t*)d_src; a.sw = w; a.sh = h; a.sstride = sstride;
    a.dst = (uint8_t*)d_dst; a.dw = w; a.dh = h; a.dstride = dstride;
    a.mapx = a.mapy = nullptr; a.map_stride = 0;
    a.mode = kind;
    a.tx0 = (const double*)ctx->stage[6].p; a.tx1 = a.tx0 + w; a.ty0 = a.tx1 + w; a.ty1 = a.ty0 + h;
    a.f = f; a.half_w = (float)(0.5 * w); a.half_h = (float)(0.5 * h);    // u = 0.5 * w + xhat * f as a float MatExpr (:186-187)
    ProfEvent pe; prof_begin(ctx, DR3HIP_PROF_REMAP, pe);
    HIPCHK(ctx, launch_remap(a, type, ctx->stream));
    prof_end(ctx, pe);
    return DR3HIP_OK;
}

int dr3hip_warp_local(dr3hip_ctx* ctx, const void* src, int w, int h, int sstride, int type,
                      void* dst, int dstride, int kind, float f)
{
    if (!ctx) return DR3HIP_EINVAL;
    const int esz = type_elem_bytes(type);
    if (!src || !dst || esz == 0 || !side_ok(w) || !side_ok(h) || sstride < w * esz || dstride < w * esz)
        return fail(ctx, DR3HIP_EINVAL, "warp_local: bad arguments");
    DeviceGuard g(ctx->device);
    const int sp = align_up(w * esz, 64);
    int rc;
    if ((rc = stage_reserve(ctx, 0, (size_t)sp * h)) || (rc = stage_reserve(ctx, 1, (size_t)sp * h))) return rc;
    { const int rc2_ = copy2d_h2d(ctx, ctx->stage[0].p, sp, src, sstride, (size_t)w * esz, h); if (rc2_) return rc2_; }
    if ((rc = dr3hip_warp_local_dev(ctx, ctx->stage[0].p, w, h, sp, type, ctx->stage[1].p, sp, kind, f))) return rc;
    { const int rc2_ = copy2d_d2h(ctx, dst, dstride, ctx->stage[1].p, sp, (size_t)w * esz, h); if (rc2_) return rc2_; }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return DR3HIP_OK;
}

int dr3hip_remap_bilinear(dr3hip_ctx* ctx, const void* src, int sw, int sh, int sstride, int type,
                          void* dst, int dw, int dh, int dstride,
                          const float* mapx, const float* mapy, int map_stride)
{
    if (!ctx) return DR3HIP_EINVAL;
    const int esz = type_elem_bytes(type);
    if (!src || !dst || !mapx || !mapy || esz == 0 || !side_ok(sw) || !side_ok(sh) || !side_ok(dw) || !side_ok(dh) ||
        sstride < sw * esz || dstride < dw * esz || map_stride < dw * 4)
        return fail(ctx, DR3HIP_EINVAL, "remap: bad arguments");
    DeviceGuard g(ctx->device);
    const int sp = align_up(sw * esz, 64), dp = align_up(dw * esz, 64), mp = align_up(dw * 4, 64);
    int rc;
    if ((rc = stage_reserve(ctx, 0, (size_t)sp * sh + 64)) || (rc = stage_reserve(ctx, 1, (size_t)dp * dh + 64)) ||
        (rc = stage_reserve(ctx, 2, (size_t)mp * dh)) || (rc = stage_reserve(ctx, 3, (size_t)mp * dh))) return rc;
    { const int rc2_ = copy2d_h2d(ctx, ctx->stage[0].p, sp, src, sstride, (size_t)sw * esz, sh); if (rc2_) return rc2_; }
    { const int rc2_ = copy2d_h2d(ctx, ctx->stage[2].p, mp, mapx, map_stride, (size_t)dw * 4, dh); if (rc2_) return rc2_; }
    { const int rc2_ = copy2d_h2d(ctx, ctx->stage[3].p, mp, mapy, map_stride, (size_t)dw * 4, dh); if (rc2_) return rc2_; }
    if ((rc = dr3hip_remap_bilinear_dev(ctx, ctx->stage[0].p, sw, sh, sp, type, ctx->stage[1].p, dw, dh, dp,
                                        (const float*)ctx->stage[2].p, (const float*)ctx->stage[3].p, mp))) return rc;
    { const int rc2_ = copy2d_d2h(ctx, dst, dstride, ctx->stage[1].p, dp, (size_t)dw * esz, dh); if (rc2_) return rc2_; }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return DR3HIP_OK;
}

}  // extern "C"
