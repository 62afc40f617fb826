// TEST-ONLY STAND-IN for <opencv2/core.hpp>.  OpenCV is absent from this image, so the DR3HIP_WITH_OPENCV branch of
// include/dr3hip/compat.hpp could not be compiled at all.  This file declares just enough of the cv:: names that branch and
// the reference's call sites use -- cv::Mat (ref-counted pixels, create / ptr / type / step), cv::Point2f, cv::Size,
// cv::TermCriteria, a 3x3 f32 matrix product -- for tests/opencv_shim/callsites.cpp to compile and run.  It is a syntax and
// plumbing check of the drop-in layer: it is NOT OpenCV, it restates none of OpenCV's algorithms and it pins nothing.
#ifndef DR3_TEST_OPENCV_SHIM_CORE_HPP
#define DR3_TEST_OPENCV_SHIM_CORE_HPP
#include <cstddef>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <vector>

#define CV_8U 0
#define CV_32F 5
#define CV_64F 6
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn) - 1) << 3))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_8UC4 CV_MAKETYPE(CV_8U, 4)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)
#define CV_32FC3 CV_MAKETYPE(CV_32F, 3)
#define CV_32FC4 CV_MAKETYPE(CV_32F, 4)
#define CV_64FC1 CV_MAKETYPE(CV_64F, 1)
#define CV_INTER_LINEAR 1

namespace cv {
typedef unsigned char uchar;

template <class T> struct Point_ { T x, y; Point_() : x(0), y(0) {} Point_(T x_, T y_) : x(x_), y(y_) {} };
typedef Point_<float> Point2f;
template <class T> struct Size_ { T width, height; Size_() : width(0), height(0) {} Size_(T w, T h) : width(w), height(h) {} };
typedef Size_<int> Size2i;
typedef Size2i Size;

struct TermCriteria {
    enum { COUNT = 1, MAX_ITER = COUNT, EPS = 2 };
    int type, maxCount; double epsilon;
    TermCriteria() : type(0), maxCount(0), epsilon(0) {}
    TermCriteria(int t, int n, double e) : type(t), maxCount(n), epsilon(e) {}
};
enum { OPTFLOW_USE_INITIAL_FLOW = 4, OPTFLOW_LK_GET_MIN_EIGENVALS = 8 };
enum { INTER_LINEAR = 1, WARP_INVERSE_MAP = 16 };

class Mat {
public:
    int rows, cols; uchar* data; std::size_t step;
    Mat() : rows(0), cols(0), data(nullptr), step(0), type_(0) {}
    Mat(int r, int c, int t) : rows(0), cols(0), data(nullptr), step(0), type_(0) { create(r, c, t); }
    Mat(Size s, int t) : rows(0), cols(0), data(nullptr), step(0), type_(0) { create(s.height, s.width, t); }
    Mat(int r, int c, int t, void* user, std::size_t st = 0) : rows(r), cols(c), data(static_cast<uchar*>(user)), step(st ? st : (std::size_t)c * esz(t)), type_(t) {}
    Mat(Size s, int t, void* user, std::size_t st = 0) : rows(s.height), cols(s.width), data(static_cast<uchar*>(user)), step(st ? st : (std::size_t)s.width * esz(t)), type_(t) {}
    void create(int r, int c, int t) {
        if (data && r == rows && c == cols && t == type_ && own_) return;
        rows = r; cols = c; type_ = t; step = (std::size_t)c * esz(t);
        own_ = std::shared_ptr<uchar>(new uchar[step * (std::size_t)r + 64](), std::default_delete<uchar[]>());
        data = own_.get();
    }
    void create(Size s, int t) { create(s.height, s.width, t); }
    int type() const { return type_; }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    Size size() const { return Size(cols, rows); }
    std::size_t elemSize() const { return esz(type_); }
    template <class T> T* ptr(int r = 0) { return reinterpret_cast<T*>(data + step * (std::size_t)r); }
    template <class T> const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(data + step * (std::size_t)r); }
    Mat clone() const { Mat m(rows, cols, type_); for (int r = 0; r < rows; ++r) std::memcpy(m.data + m.step * r, data + step * r, (std::size_t)cols * esz(type_)); return m; }
    static std::size_t esz(int t) { static const int d[7] = {1, 1, 2, 2, 4, 4, 8}; return (std::size_t)d[t & 7] * ((t >> 3) + 1); }
private:
    int type_;
    std::shared_ptr<uchar> own_;
};

// the only matrix expression the hot path's call sites contain: a 3x3 CV_32F product (src/stitch.cpp:73)
inline Mat operator*(const Mat& a, const Mat& b) {
    if (a.type() != CV_32FC1 || b.type() != CV_32FC1 || a.cols != b.rows) throw std::runtime_error("shim: f32 matrix product only");
    Mat c(a.rows, b.cols, CV_32FC1);
    for (int i = 0; i < a.rows; ++i)
        for (int j = 0; j < b.cols; ++j) {
            double s = 0;                                     // cv::gemm on CV_32F accumulates in double
            for (int k = 0; k < a.cols; ++k) s += (double)a.ptr<float>(i)[k] * b.ptr<float>(k)[j];
            c.ptr<float>(i)[j] = (float)s;
        }
    return c;
}
}  // namespace cv
#endif
