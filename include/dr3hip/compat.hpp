// dr3hip/compat.hpp -- C++14 drop-in layer over the C ABI (include/dr3hip.h).
//
// It offers the four calls 3DR makes on its dense hot path with the argument lists OpenCV gives them,
// so the reference's call sites change by namespace only:
//     cv::calcOpticalFlowPyrLK   src/initialization.cpp:608-613   -> dr3hip::calcOpticalFlowPyrLK
//     cv::warpPerspective        src/stitch.cpp:73-74, src/panorama.cpp:192 -> dr3hip::warpPerspective
//     cv::remap                  src/utils.cpp:191                -> dr3hip::remap
//     utils::create_img_pyramid  src/utils.cpp:421-430 (src/frame.cpp:18) -> dr3hip::create_img_pyramid
//     feature_detection::FastDetector::detect  src/features.cpp:43-98 (src/frame.cpp:57, src/initialization.cpp:552)
//                                                                 -> dr3hip::FastDetector
//     utils::warp_spherical / warp_cylindrical  src/utils.cpp:196-201, :267-272 -> dr3hip::warp_spherical / _cylindrical
//     Stitch::process after align_pair  src/stitch.cpp:33-80       -> dr3hip::stitch_compose
//     Panorama::paste_images / add_img_to_canvas / normalize_canvas  src/panorama.cpp:143-229
//                                                                 -> dr3hip::WeightedCanvas
// Without OpenCV images travel as MatView {data, rows, cols, step, type} and outputs as dr3hip::Image.  With
// DR3HIP_WITH_OPENCV defined before including (the reference's build) every call above also exists with OpenCV's own
// argument types -- const cv::Mat&, cv::Mat& dst, std::vector<cv::Point2f>&, cv::Size, cv::TermCriteria, a cv::Mat M --
// so the reference's call expressions change by namespace ONLY (cv:: / utils:: -> dr3hip::), with no cast and no copy at the
// call site; include/features.hpp, include/stitch.hpp and include/panorama.hpp of the reference stay untouched.
// Error behaviour mirrors OpenCV's: bad arguments / device errors throw (dr3hip::Error instead of
// cv::Exception); per-point tracking failure is reported through status[i] == 0 only.
#ifndef DR3HIP_COMPAT_HPP
#define DR3HIP_COMPAT_HPP

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../dr3hip.h"

#ifdef DR3HIP_WITH_OPENCV
#include <opencv2/core.hpp>
#endif

namespace dr3hip {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

// cv-shaped value types (layout-compatible with cv::Point2f / cv::Size / cv::TermCriteria)
struct Point2f { float x, y; Point2f() : x(0), y(0) {} Point2f(float x_, float y_) : x(x_), y(y_) {} };
struct Size { int width, height; Size() : width(0), height(0) {} Size(int w, int h) : width(w), height(h) {} };
struct TermCriteria {
    enum { COUNT = 1, MAX_ITER = 1, EPS = 2 };
    int type; int maxCount; double epsilon;
    TermCriteria() : type(COUNT + EPS), maxCount(30), epsilon(0.01) {}
    TermCriteria(int t, int n, double e) : type(t), maxCount(n), epsilon(e) {}
};
enum { OPTFLOW_USE_INITIAL_FLOW = 4, OPTFLOW_LK_GET_MIN_EIGENVALS = 8 };
enum { INTER_LINEAR = 1, WARP_INVERSE_MAP = 16 };

// non-owning image view; type is one of DR3HIP_8UC1 / 8UC3 / 8UC4 / 32FC1 / 32FC4
struct MatView {
    void* data; int rows, cols; std::size_t step; int type;
    MatView() : data(nullptr), rows(0), cols(0), step(0), type(DR3HIP_8UC1) {}
    MatView(void* d, int r, int c, std::size_t s, int t) : data(d), rows(r), cols(c), step(s), type(t) {}
    bool empty() const { return data == nullptr || rows <= 0 || cols <= 0; }
    static int elem_size(int t) {
        switch (t) { case DR3HIP_8UC1: return 1; case DR3HIP_8UC3: return 3; case DR3HIP_8UC4: return 4;
                     case DR3HIP_32FC1: return 4; case DR3HIP_32FC4: return 16; default: return 0; }
    }
#ifdef DR3HIP_WITH_OPENCV
    static MatView of(const cv::Mat& m) {
        int t;
        switch (m.type()) {
        case CV_8UC1: t = DR3HIP_8UC1; break;   case CV_8UC3: t = DR3HIP_8UC3; break;
        case CV_8UC4: t = DR3HIP_8UC4; break;   case CV_32FC1: t = DR3HIP_32FC1; break;
        case CV_32FC4: t = DR3HIP_32FC4; break;
        default: throw Error(DR3HIP_EINVAL, "dr3hip::MatView::of: unsupported cv::Mat type");
        }
        return MatView(m.data, m.rows, m.cols, m.step, t);
    }
#endif
};

// owning image for outputs the reference receives as freshly created cv::Mat
struct Image {
    std::vector<unsigned char> buf; int rows, cols, type; std::size_t step;
    Image() : rows(0), cols(0), type(DR3HIP_8UC1), step(0) {}
    void create(int r, int c, int t) {
        rows = r; cols = c; type = t; step = (std::size_t)c * MatView::elem_size(t);
        buf.assign(step * (std::size_t)r, 0);
    }
    MatView view() { return MatView(buf.data(), rows, cols, step, type); }
    unsigned char* ptr(int r) { return buf.data() + step * (std::size_t)r; }
};

// One context per thread per device, created on first use (the reference is single-threaded;
// a dr3hip_ctx is not thread-safe, distinct ones are).
class Context {
public:
    static Context& instance(int device = 0) {
        static thread_local Context c(device);
        if (c.device_ != device) throw Error(DR3HIP_EINVAL, "dr3hip::Context: one device per thread");
        return c;
    }
    dr3hip_ctx* get() const { return ctx_; }
    // a caller that runs its own HIP work: enqueue the device-pointer calls on its stream (hipStream_t as void*; nullptr = back to
    // the context's own stream)
    void set_stream(void* hip_stream) { check(dr3hip_set_stream(ctx_, hip_stream), "dr3hip_set_stream"); }
    void* stream() const { return dr3hip_stream(ctx_); }
    int check(int rc, const char* where) const {
        if (rc < 0) throw Error(rc, std::string(where) + ": " + dr3hip_last_error(ctx_));
        return rc;
    }
    ~Context() { dr3hip_destroy(ctx_); }
private:
    explicit Context(int device) : ctx_(nullptr), device_(device) {
        const int rc = dr3hip_init(device, &ctx_);
        if (rc != DR3HIP_OK) throw Error(rc, std::string("dr3hip_init: ") + dr3hip_last_error(nullptr));
    }
    Context(const Context&); Context& operator=(const Context&);
    dr3hip_ctx* ctx_; int device_;
};

namespace detail {
// the tracker call shared by the MatView and the cv-typed front ends; PointT is any { float x, y; } pair
template <class PointT>
inline int lk_call(const MatView& prevImg, const MatView& nextImg, const std::vector<PointT>& prevPts, std::vector<PointT>& nextPts,
                   std::vector<unsigned char>& status, std::vector<float>& err, int win_w, int win_h, int maxLevel,
                   int crit_type, int crit_count, double crit_eps, int flags, double minEigThreshold, int device)
{
    static_assert(sizeof(PointT) == 2 * sizeof(float), "a point must be two packed floats (x, y)");
    if (prevImg.empty() || nextImg.empty() || prevImg.type != DR3HIP_8UC1 || nextImg.type != DR3HIP_8UC1 ||
        prevImg.rows != nextImg.rows || prevImg.cols != nextImg.cols)
        throw Error(DR3HIP_EINVAL, "calcOpticalFlowPyrLK: 8-bit single-channel images of equal size expected");
    const std::size_t n = prevPts.size();
    if (flags & 4 /* OPTFLOW_USE_INITIAL_FLOW */) {
        if (nextPts.size() != n) throw Error(DR3HIP_EINVAL, "calcOpticalFlowPyrLK: nextPts.size() != prevPts.size() with USE_INITIAL_FLOW");
    } else {
        nextPts.resize(n);
    }
    status.assign(n, 1);
    err.assign(n, 0.f);
    dr3hip_lk_params p;
    p.win_w = win_w; p.win_h = win_h; p.max_level = maxLevel;
    p.crit_type = crit_type; p.max_iters = crit_count; p.eps = crit_eps;
    p.flags = flags; p.min_eig_threshold = minEigThreshold;
    Context& c = Context::instance(device);
    return c.check(dr3hip_lk_track(c.get(), static_cast<const uint8_t*>(prevImg.data), static_cast<const uint8_t*>(nextImg.data),
                                   prevImg.cols, prevImg.rows, (int)prevImg.step, (int)nextImg.step,
                                   n ? &prevPts[0].x : nullptr, n ? &nextPts[0].x : nullptr, (int)n,
                                   n ? status.data() : nullptr, n ? err.data() : nullptr, &p),
                   "calcOpticalFlowPyrLK");
}
}  // namespace detail

// cv::calcOpticalFlowPyrLK(prevImg, nextImg, prevPts, nextPts, status, err, winSize, maxLevel,
//                          criteria, flags, minEigThreshold)
inline int calcOpticalFlowPyrLK(const MatView& prevImg, const MatView& nextImg,
                                const std::vector<Point2f>& prevPts, std::vector<Point2f>& nextPts,
                                std::vector<unsigned char>& status, std::vector<float>& err,
                                Size winSize = Size(21, 21), int maxLevel = 3,
                                TermCriteria criteria = TermCriteria(), int flags = 0,
                                double minEigThreshold = 1e-4, int device = 0)
{
    return detail::lk_call(prevImg, nextImg, prevPts, nextPts, status, err, winSize.width, winSize.height, maxLevel,
                           criteria.type, criteria.maxCount, criteria.epsilon, flags, minEigThreshold, device);
}

// cv::warpPerspective(src, dst, M, dsize, flags = INTER_LINEAR, BORDER_CONSTANT, 0); M: 3x3 f32 row-major
inline void warpPerspective(const MatView& src, Image& dst, const float M[9], Size dsize, int flags = INTER_LINEAR, int device = 0)
{
    if (src.empty()) throw Error(DR3HIP_EINVAL, "warpPerspective: empty source");
    if ((flags & 7) != INTER_LINEAR) throw Error(DR3HIP_EINVAL, "warpPerspective: only INTER_LINEAR is implemented (the reference uses no other)");
    dst.create(dsize.height > 0 ? dsize.height : src.rows, dsize.width > 0 ? dsize.width : src.cols, src.type);
    Context& c = Context::instance(device);
    c.check(dr3hip_warp_perspective(c.get(), src.data, src.cols, src.rows, (int)src.step, src.type,
                                    dst.buf.data(), dst.cols, dst.rows, (int)dst.step, M,
                                    (flags & WARP_INVERSE_MAP) ? DR3HIP_WARP_INVERSE_MAP : 0), "warpPerspective");
}

// cv::remap(src, dst, map1 (CV_32FC1 x), map2 (CV_32FC1 y), INTER_LINEAR)
inline void remap(const MatView& src, Image& dst, const MatView& map1, const MatView& map2, int interpolation = INTER_LINEAR, int device = 0)
{
    if (src.empty() || map1.empty() || map2.empty() || map1.type != DR3HIP_32FC1 || map2.type != DR3HIP_32FC1 ||
        map1.rows != map2.rows || map1.cols != map2.cols || map1.step != map2.step)
        throw Error(DR3HIP_EINVAL, "remap: CV_32FC1 maps of equal size expected");
    if (interpolation != INTER_LINEAR) throw Error(DR3HIP_EINVAL, "remap: only INTER_LINEAR is implemented");
    dst.create(map1.rows, map1.cols, src.type);
    Context& c = Context::instance(device);
    c.check(dr3hip_remap_bilinear(c.get(), src.data, src.cols, src.rows, (int)src.step, src.type,
                                  dst.buf.data(), dst.cols, dst.rows, (int)dst.step,
                                  static_cast<const float*>(map1.data), static_cast<const float*>(map2.data), (int)map1.step), "remap");
}

// utils::warp_spherical(img, f) / utils::warp_cylindrical(img, f) (src/utils.cpp:196-201, :267-272): the
// reference computes two f32 maps in six full-image passes and calls cv::remap; here the maps are generated
// inside the remap kernel.  Returns the warped image (the reference returns warped.clone()).
inline void warp_spherical(const MatView& img, float f, Image& warped, int device = 0)
{
    if (img.empty()) throw Error(DR3HIP_EINVAL, "warp_spherical: empty image");
    warped.create(img.rows, img.cols, img.type);
    Context& c = Context::instance(device);
    c.check(dr3hip_warp_local(c.get(), img.data, img.cols, img.rows, (int)img.step, img.type,
                              warped.buf.data(), (int)warped.step, DR3HIP_WARP_SPHERICAL, f), "warp_spherical");
}
inline void warp_cylindrical(const MatView& img, float f, Image& warped, int device = 0)
{
    if (img.empty()) throw Error(DR3HIP_EINVAL, "warp_cylindrical: empty image");
    warped.create(img.rows, img.cols, img.type);
    Context& c = Context::instance(device);
    c.check(dr3hip_warp_local(c.get(), img.data, img.cols, img.rows, (int)img.step, img.type,
                              warped.buf.data(), (int)warped.step, DR3HIP_WARP_CYLINDRICAL, f), "warp_cylindrical");
}

// utils::create_img_pyramid(img_lvl_0, n_levels, pyr).  Level 0 is a copy of the input (the
// reference stores the cv::Mat header, i.e. shares the pixels); levels 1.. are floor-halved.
// mode DR3HIP_HALF_REF_X86 reproduces what the reference executes on x86-64 byte for byte.
inline void create_img_pyramid(const MatView& img_lvl_0, int n_levels, std::vector<Image>& pyr,
                               int mode = DR3HIP_HALF_REF_X86, int device = 0)
{
    if (img_lvl_0.empty() || img_lvl_0.type != DR3HIP_8UC1 || n_levels < 1)
        throw Error(DR3HIP_EINVAL, "create_img_pyramid: 8-bit gray image and n_levels >= 1 expected");
    pyr.resize((std::size_t)n_levels);
    pyr[0].create(img_lvl_0.rows, img_lvl_0.cols, DR3HIP_8UC1);
    for (int r = 0; r < img_lvl_0.rows; ++r)
        std::memcpy(pyr[0].ptr(r), static_cast<const unsigned char*>(img_lvl_0.data) + img_lvl_0.step * (std::size_t)r, (std::size_t)img_lvl_0.cols);
    std::vector<uint8_t*> ptrs((std::size_t)n_levels, nullptr);
    std::vector<int> strides((std::size_t)n_levels, 0);
    for (int l = 1; l < n_levels; ++l) {
        pyr[l].create(pyr[l - 1].rows / 2, pyr[l - 1].cols / 2, DR3HIP_8UC1);
        ptrs[l] = pyr[l].buf.data(); strides[l] = (int)pyr[l].step;
        if (pyr[l].rows == 0 || pyr[l].cols == 0) throw Error(DR3HIP_EINVAL, "create_img_pyramid: too many levels for this image");
    }
    Context& c = Context::instance(device);
    c.check(dr3hip_halfsample_pyramid(c.get(), static_cast<const uint8_t*>(img_lvl_0.data), img_lvl_0.cols, img_lvl_0.rows,
                                      (int)img_lvl_0.step, n_levels, mode, ptrs.data(), strides.data()), "create_img_pyramid");
}

// reconstruct::Stitch::process after align_pair (src/stitch.cpp:33-80): H is what Stitch::align_pair returned (CV_32F 3x3);
// fills Hinv (the reference stores it in this->H, :37) and returns the stitched image (this->final_stitched_img, :80).
inline void stitch_compose(const MatView& left_img, const MatView& right_img, const float H[9], float Hinv[9], Image& stitched, int device = 0)
{
    if (left_img.empty() || right_img.empty() || left_img.type != DR3HIP_8UC3 || right_img.type != DR3HIP_8UC3)
        throw Error(DR3HIP_EINVAL, "stitch_compose: 8UC3 images expected");
    float T[9], M_right[9];
    int new_width = 0, new_height = 0;
    if (dr3hip_stitch_geometry(H, right_img.cols, right_img.rows, Hinv, T, M_right, &new_width, &new_height) != DR3HIP_OK)
        throw Error(DR3HIP_EINVAL, "stitch_compose: singular homography");
    stitched.create(new_height, new_width, DR3HIP_8UC3);
    Context& c = Context::instance(device);
    c.check(dr3hip_stitch_compose(c.get(), static_cast<const uint8_t*>(left_img.data), left_img.cols, left_img.rows, (int)left_img.step,
                                  static_cast<const uint8_t*>(right_img.data), right_img.cols, right_img.rows, (int)right_img.step,
                                  T, M_right, new_width, new_height, 0.5f, 0.5f, 0.0f, stitched.buf.data(), (int)stitched.step),
            "stitch_compose");
}

// feature_detection::FastDetector (include/features.hpp:117-130, src/features.cpp:36-98): same constructor
// arguments and the same grid bookkeeping (flag_grid / reset_grid); detect() takes the level-0 image (the
// half-sample pyramid the reference passes as img_pyr is rebuilt on the device) and returns the corners the
// reference turns into Feature objects (:86-93: one per cell whose score beats detection_threshold).
struct Corner { int x, y, level; float score; };       // feature_detection::Corner (include/features.hpp:66-76)
class FastDetector {
public:
    FastDetector(int img_w, int img_h, int cell_size, int n_pyr_levels, int device = 0)
        : w_(img_w), h_(img_h), cell_size_(cell_size), n_pyr_levels_(n_pyr_levels), device_(device) {
        if (dr3hip_fast_grid_size(img_w, img_h, cell_size, &grid_n_cols_, &grid_n_rows_) != DR3HIP_OK)
            throw Error(DR3HIP_EINVAL, "FastDetector: bad geometry");
        grid_occupancy_.assign((std::size_t)grid_n_cols_ * grid_n_rows_, 0);
    }
    void reset_grid() { grid_occupancy_.assign(grid_occupancy_.size(), 0); }                   // src/features.cpp:19-21
    void flag_grid(double px_x, double px_y) {                                                   // :23-27
        grid_occupancy_.at((std::size_t)(static_cast<int>(px_y / cell_size_) * grid_n_cols_ + static_cast<int>(px_x / cell_size_))) = 1;
    }
    void detect(const MatView& img_lvl_0, double detection_threshold, std::vector<Corner>& corners) {
        if (img_lvl_0.empty() || img_lvl_0.type != DR3HIP_8UC1 || img_lvl_0.cols != w_ || img_lvl_0.rows != h_)
            throw Error(DR3HIP_EINVAL, "FastDetector::detect: 8-bit gray image of the constructor's size expected");
        std::vector<dr3hip_corner> cells(grid_occupancy_.size());
        Context& c = Context::instance(device_);
        c.check(dr3hip_fast_detect(c.get(), static_cast<const uint8_t*>(img_lvl_0.data), w_, h_, (int)img_lvl_0.step, n_pyr_levels_,
                                   cell_size_, 20, detection_threshold, grid_occupancy_.data(), DR3HIP_HALF_REF_X86, cells.data()),
                "FastDetector::detect");
        for (std::size_t k = 0; k < cells.size(); ++k)
            if (cells[k].level >= 0) { Corner q; q.x = cells[k].x; q.y = cells[k].y; q.level = cells[k].level; q.score = cells[k].score; corners.push_back(q); }
        reset_grid();                                                                            // :95
    }
    int grid_n_cols() const { return grid_n_cols_; }
    int grid_n_rows() const { return grid_n_rows_; }
private:
    int w_, h_, cell_size_, n_pyr_levels_, device_, grid_n_cols_, grid_n_rows_;
    std::vector<unsigned char> grid_occupancy_;
};

// reconstruct::Panorama::paste_images / add_img_to_canvas / normalize_canvas (src/panorama.cpp:143-229).
// The reference's `cv::Mat weighted_panorama` (CV_32FC4, zeros(_W x _H), :144) becomes a device-resident
// canvas: add_img_to_canvas(img, M, feathering_width) is the fused warp + feather + accumulate kernel,
// normalize() is normalize_canvas into an 8UC3 Image (the reference's _final_panorama).
class WeightedCanvas {
public:
    WeightedCanvas(Size size, int device = 0) : c_(Context::instance(device)), W_(size.width), H_(size.height), canvas_(nullptr) {
        if (W_ < 1 || H_ < 1) throw Error(DR3HIP_EINVAL, "WeightedCanvas: empty canvas");
        void* p = nullptr;
        c_.check(dr3hip_malloc(c_.get(), (std::size_t)W_ * H_ * 16, &p), "WeightedCanvas");
        canvas_ = static_cast<float*>(p);
        c_.check(dr3hip_memset(c_.get(), canvas_, 0, (std::size_t)W_ * H_ * 16), "WeightedCanvas");
    }
    ~WeightedCanvas() { if (canvas_) dr3hip_free(c_.get(), canvas_); }
    Size size() const { return Size(W_, H_); }
    // Panorama::add_img_to_canvas(img, M, feathering_width, weighted_panorama): img 8UC3, M = top_left * H (f32)
    void add_img_to_canvas(const MatView& img, const float M[9], int feathering_width) {
        if (img.empty() || img.type != DR3HIP_8UC3) throw Error(DR3HIP_EINVAL, "add_img_to_canvas: 8UC3 image expected");
        void* d = nullptr;
        const std::size_t pitch = ((std::size_t)img.cols * 3 + 63) / 64 * 64;
        c_.check(dr3hip_malloc(c_.get(), pitch * img.rows, &d), "add_img_to_canvas");
        int rc = DR3HIP_OK;
        for (int r = 0; r < img.rows && rc == DR3HIP_OK; ++r)      // row copies keep the ABI to plain memcpy
            rc = dr3hip_memcpy_h2d(c_.get(), static_cast<unsigned char*>(d) + pitch * r,
                                   static_cast<const unsigned char*>(img.data) + img.step * (std::size_t)r, (std::size_t)img.cols * 3);
        if (rc == DR3HIP_OK)
            rc = dr3hip_panorama_add(c_.get(), static_cast<const uint8_t*>(d), img.cols, img.rows, (int)pitch,
                                     canvas_, W_, H_, W_ * 16, M, feathering_width);
        if (rc == DR3HIP_OK) rc = dr3hip_sync(c_.get());
        dr3hip_free(c_.get(), d);
        c_.check(rc, "add_img_to_canvas");
    }
    // Panorama::normalize_canvas(weighted_panorama) -> _final_panorama (8UC3, _W x _H)
    void normalize(Image& final_panorama) {
        final_panorama.create(H_, W_, DR3HIP_8UC3);
        void* d = nullptr;
        c_.check(dr3hip_malloc(c_.get(), (std::size_t)W_ * H_ * 3, &d), "normalize_canvas");
        int rc = dr3hip_panorama_normalize(c_.get(), canvas_, W_, H_, W_ * 16, static_cast<uint8_t*>(d), W_ * 3);
        if (rc == DR3HIP_OK) rc = dr3hip_memcpy_d2h(c_.get(), final_panorama.buf.data(), d, (std::size_t)W_ * H_ * 3);
        dr3hip_free(c_.get(), d);
        c_.check(rc, "normalize_canvas");
    }
    // the weighted canvas itself (tests; the reference never reads it back)
    void download(Image& weighted) {
        weighted.create(H_, W_, DR3HIP_32FC4);
        c_.check(dr3hip_memcpy_d2h(c_.get(), weighted.buf.data(), canvas_, (std::size_t)W_ * H_ * 16), "WeightedCanvas::download");
    }
private:
    WeightedCanvas(const WeightedCanvas&); WeightedCanvas& operator=(const WeightedCanvas&);
    Context& c_; int W_, H_; float* canvas_;
};

#ifdef DR3HIP_WITH_OPENCV
// =====================================================================================================================
// OpenCV-typed front ends: the reference's call expressions with `cv::` (or `utils::`) replaced by `dr3hip::`, nothing else.
//   src/initialization.cpp:608-613   dr3hip::calcOpticalFlowPyrLK(frame_ref->_img_pyr[0], frame_cur->_img_pyr[0], _kps_ref, _kps_cur,
//                                        status, error, cv::Size2i(klt_win_size, klt_win_size), 4, termcrit, cv::OPTFLOW_USE_INITIAL_FLOW);
//   src/stitch.cpp:73-74             dr3hip::warpPerspective(_right_img, warped_right_image, translate_top_left_mat * Hinv, {new_width, new_height});
//   src/panorama.cpp:192             dr3hip::warpPerspective(weighted_img, warped_img, M, this->get_final_size(), CV_INTER_LINEAR);
//   src/utils.cpp:191                dr3hip::remap(img, warped, u, v, CV_INTER_LINEAR);
//   src/frame.cpp:18                 dr3hip::create_img_pyramid(img, Config::n_pyr_levels(), _img_pyr);
//   src/utils.cpp:196-201, :267-272  dr3hip::warp_spherical(img, f) / dr3hip::warp_cylindrical(img, f)
// Outputs are created like OpenCV creates them (dst.create(dsize, src.type())); the kernels write straight into them.
// =====================================================================================================================
inline void calcOpticalFlowPyrLK(const cv::Mat& prevImg, const cv::Mat& nextImg,
                                 const std::vector<cv::Point2f>& prevPts, std::vector<cv::Point2f>& nextPts,
                                 std::vector<unsigned char>& status, std::vector<float>& err,
                                 cv::Size winSize = cv::Size(21, 21), int maxLevel = 3,
                                 cv::TermCriteria criteria = cv::TermCriteria(cv::TermCriteria::COUNT + cv::TermCriteria::EPS, 30, 0.01),
                                 int flags = 0, double minEigThreshold = 1e-4)
{
    detail::lk_call(MatView::of(prevImg), MatView::of(nextImg), prevPts, nextPts, status, err, winSize.width, winSize.height, maxLevel,
                    criteria.type, criteria.maxCount, criteria.epsilon, flags, minEigThreshold, 0);
}

namespace detail {
// the 3x3 matrix as the f32 values the C ABI takes.  The reference passes CV_32F (src/stitch.cpp:71-74, src/panorama.cpp:148);
// CV_64F is accepted when every entry is exactly a float (OpenCV itself widens M to double, so nothing is lost either way).
inline void mat3x3_f32(const cv::Mat& M, float out[9])
{
    if (M.rows != 3 || M.cols != 3 || (M.type() != CV_32FC1 && M.type() != CV_64FC1))
        throw Error(DR3HIP_EINVAL, "warpPerspective: M must be a 3x3 CV_32F or CV_64F matrix");
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            if (M.type() == CV_32FC1) out[3 * r + c] = M.ptr<float>(r)[c];
            else {
                const double v = M.ptr<double>(r)[c];
                out[3 * r + c] = static_cast<float>(v);
                if (static_cast<double>(out[3 * r + c]) != v && v == v)
                    throw Error(DR3HIP_EINVAL, "warpPerspective: CV_64F matrix entry is not representable in f32 (the C ABI takes the f32 matrix the reference passes)");
            }
        }
}
}  // namespace detail

inline void warpPerspective(const cv::Mat& src, cv::Mat& dst, const cv::Mat& M, cv::Size dsize, int flags = 1 /* cv::INTER_LINEAR */)
{
    if (src.empty()) throw Error(DR3HIP_EINVAL, "warpPerspective: empty source");
    if ((flags & 7) != INTER_LINEAR) throw Error(DR3HIP_EINVAL, "warpPerspective: only INTER_LINEAR is implemented (the reference uses no other)");
    float m[9];
    detail::mat3x3_f32(M, m);
    const MatView s = MatView::of(src);
    cv::Mat out;                                                         // src and dst may be the same Mat: warp into a fresh one
    out.create(dsize.height > 0 ? dsize.height : src.rows, dsize.width > 0 ? dsize.width : src.cols, src.type());
    Context& c = Context::instance(0);
    c.check(dr3hip_warp_perspective(c.get(), s.data, s.cols, s.rows, (int)s.step, s.type, out.data, out.cols, out.rows, (int)out.step, m,
                                    (flags & WARP_INVERSE_MAP) ? DR3HIP_WARP_INVERSE_MAP : 0), "warpPerspective");
    dst = out;
}

inline void remap(const cv::Mat& src, cv::Mat& dst, const cv::Mat& map1, const cv::Mat& map2, int interpolation)
{
    if (interpolation != INTER_LINEAR) throw Error(DR3HIP_EINVAL, "remap: only INTER_LINEAR is implemented");
    const MatView s = MatView::of(src), m1 = MatView::of(map1), m2 = MatView::of(map2);
    if (s.empty() || m1.empty() || m2.empty() || m1.type != DR3HIP_32FC1 || m2.type != DR3HIP_32FC1 ||
        m1.rows != m2.rows || m1.cols != m2.cols || m1.step != m2.step)
        throw Error(DR3HIP_EINVAL, "remap: CV_32FC1 maps of equal size expected");
    cv::Mat out;
    out.create(map1.rows, map1.cols, src.type());
    Context& c = Context::instance(0);
    c.check(dr3hip_remap_bilinear(c.get(), s.data, s.cols, s.rows, (int)s.step, s.type, out.data, out.cols, out.rows, (int)out.step,
                                  static_cast<const float*>(m1.data), static_cast<const float*>(m2.data), (int)m1.step), "remap");
    dst = out;
}

// utils::create_img_pyramid(img_lvl_0, n_levels, pyr) with ImgPyramid = std::vector<cv::Mat> (include/frame.hpp:17): level 0
// shares the caller's pixels exactly as src/utils.cpp:423 does, levels 1.. are new CV_8U images of floor-halved size
inline void create_img_pyramid(const cv::Mat& img_lvl_0, int n_levels, std::vector<cv::Mat>& pyr, int mode = DR3HIP_HALF_REF_X86)
{
    const MatView v0 = MatView::of(img_lvl_0);
    if (v0.empty() || v0.type != DR3HIP_8UC1 || n_levels < 1)
        throw Error(DR3HIP_EINVAL, "create_img_pyramid: 8-bit gray image and n_levels >= 1 expected");
    pyr.resize((std::size_t)n_levels);
    pyr[0] = img_lvl_0;
    std::vector<uint8_t*> ptrs((std::size_t)n_levels, nullptr);
    std::vector<int> strides((std::size_t)n_levels, 0);
    for (int l = 1; l < n_levels; ++l) {
        if (pyr[l - 1].rows / 2 == 0 || pyr[l - 1].cols / 2 == 0) throw Error(DR3HIP_EINVAL, "create_img_pyramid: too many levels for this image");
        pyr[l] = cv::Mat(pyr[l - 1].rows / 2, pyr[l - 1].cols / 2, CV_8UC1);
        ptrs[l] = pyr[l].data; strides[l] = (int)pyr[l].step;
    }
    Context& c = Context::instance(0);
    c.check(dr3hip_halfsample_pyramid(c.get(), static_cast<const uint8_t*>(v0.data), v0.cols, v0.rows, (int)v0.step, n_levels, mode,
                                      ptrs.data(), strides.data()), "create_img_pyramid");
}

inline cv::Mat warp_spherical(cv::Mat img, float f)
{
    const MatView s = MatView::of(img);
    if (s.empty()) throw Error(DR3HIP_EINVAL, "warp_spherical: empty image");
    cv::Mat warped;
    warped.create(img.rows, img.cols, img.type());
    Context& c = Context::instance(0);
    c.check(dr3hip_warp_local(c.get(), s.data, s.cols, s.rows, (int)s.step, s.type, warped.data, (int)warped.step, DR3HIP_WARP_SPHERICAL, f), "warp_spherical");
    return warped;
}
inline cv::Mat warp_cylindrical(cv::Mat img, float f)
{
    const MatView s = MatView::of(img);
    if (s.empty()) throw Error(DR3HIP_EINVAL, "warp_cylindrical: empty image");
    cv::Mat warped;
    warped.create(img.rows, img.cols, img.type());
    Context& c = Context::instance(0);
    c.check(dr3hip_warp_local(c.get(), s.data, s.cols, s.rows, (int)s.step, s.type, warped.data, (int)warped.step, DR3HIP_WARP_CYLINDRICAL, f), "warp_cylindrical");
    return warped;
}
#endif  // DR3HIP_WITH_OPENCV

}  // namespace dr3hip
#endif  // DR3HIP_COMPAT_HPP
