// The reference's hot-path call expressions, verbatim except for the namespace (cv:: / utils:: -> dr3hip::), compiled against
// include/dr3hip/compat.hpp with DR3HIP_WITH_OPENCV and the TEST-ONLY <opencv2/core.hpp> stand-in of this directory.
//   src/initialization.cpp:592-613   src/stitch.cpp:70-74   src/panorama.cpp:191-192   src/utils.cpp:189-201   src/frame.cpp:18
// usage: callsites <in.bin> <out.bin>   (tests/test_opencv_overloads.py writes the input and checks the output on the GPU)
#define DR3HIP_WITH_OPENCV
#include <dr3hip/compat.hpp>

#include <cstdio>
#include <cstdlib>
#include <vector>

using std::vector;
typedef unsigned char uchar;
typedef std::vector<cv::Mat> ImgPyramid;                       // include/frame.hpp:17

static void rd(FILE* f, void* p, size_t n) { if (fread(p, 1, n, f) != n) { std::fprintf(stderr, "short read\n"); std::exit(2); } }
static void wr(FILE* f, const void* p, size_t n) { if (fwrite(p, 1, n, f) != n) { std::fprintf(stderr, "short write\n"); std::exit(2); } }
static void wr_mat(FILE* f, const cv::Mat& m) {
    int hdr[3] = {m.rows, m.cols, m.type()};
    wr(f, hdr, sizeof(hdr));
    for (int r = 0; r < m.rows; ++r) wr(f, m.data + m.step * r, (size_t)m.cols * m.elemSize());
}

struct FrameLike { ImgPyramid _img_pyr; };                     // dr3::Frame::_img_pyr (include/frame.hpp:33)

int main(int argc, char** argv)
{
    if (argc != 3) return 1;
    FILE* fi = std::fopen(argv[1], "rb"); FILE* fo = std::fopen(argv[2], "wb");
    if (!fi || !fo) return 1;
    int hdr[3]; rd(fi, hdr, sizeof(hdr));
    const int w = hdr[0], h = hdr[1], n = hdr[2];
    cv::Mat img0(h, w, CV_8UC1), img1(h, w, CV_8UC1);
    rd(fi, img0.data, (size_t)w * h); rd(fi, img1.data, (size_t)w * h);
    vector<cv::Point2f> _kps_ref((size_t)n), _kps_cur;
    rd(fi, _kps_ref.data(), (size_t)n * 8);
    _kps_cur = _kps_ref;                                       // src/initialization.cpp:578: the initial guess is the identity

    // ---- src/frame.cpp:18 ----------------------------------------------------------------------------------------
    FrameLike ref, cur; FrameLike* frame_ref = &ref; FrameLike* frame_cur = &cur;
    dr3hip::create_img_pyramid(img0, 3, frame_ref->_img_pyr);
    dr3hip::create_img_pyramid(img1, 3, frame_cur->_img_pyr);

    // ---- src/initialization.cpp:592-613 ----------------------------------------------------------------------------
    const int klt_win_size = 30;
    const int klt_max_iter = 1000;
    const double klt_eps = 1e-3;
    vector<uchar> status;
    vector<float> error;
    cv::TermCriteria termcrit(cv::TermCriteria::COUNT+cv::TermCriteria::EPS,
                              klt_max_iter, klt_eps);
    dr3hip::calcOpticalFlowPyrLK(frame_ref->_img_pyr[0],
                             frame_cur->_img_pyr[0],
                             _kps_ref, _kps_cur,
                             status, error,
                             cv::Size2i(klt_win_size, klt_win_size),
                             4, termcrit, cv::OPTFLOW_USE_INITIAL_FLOW);
    wr(fo, _kps_cur.data(), (size_t)n * 8); wr(fo, status.data(), (size_t)n); wr(fo, error.data(), (size_t)n * 4);
    wr_mat(fo, frame_ref->_img_pyr[1]); wr_mat(fo, frame_ref->_img_pyr[2]);

    // ---- src/stitch.cpp:70-74 ----------------------------------------------------------------------------------------
    int g[4]; rd(fi, g, sizeof(g));                             // tile w, h, new_width, new_height
    cv::Mat _right_img(g[1], g[0], CV_8UC3), _left_img;
    rd(fi, _right_img.data, (size_t)g[0] * g[1] * 3);
    _left_img = _right_img;
    float hinv[9], min_xy[2]; rd(fi, hinv, sizeof(hinv)); rd(fi, min_xy, sizeof(min_xy));
    cv::Mat Hinv(3, 3, CV_32F, hinv);
    const float min_X = min_xy[0], min_Y = min_xy[1];
    const int new_width = g[2], new_height = g[3];
    float translate_top_left[][3] = {{1, 0, -min_X}, {0, 1, -min_Y}, {0, 0, 1}};
    cv::Mat translate_top_left_mat(cv::Size(3, 3), CV_32F, &translate_top_left);
    cv::Mat warped_right_image, warped_left_image;
    dr3hip::warpPerspective(_right_img, warped_right_image, translate_top_left_mat * Hinv, {new_width, new_height});
    dr3hip::warpPerspective(_left_img, warped_left_image, translate_top_left_mat, {new_width, new_height});
    wr_mat(fo, warped_right_image); wr_mat(fo, warped_left_image);

    // ---- src/panorama.cpp:178-192 (the 32FC4 form) -------------------------------------------------------------------
    cv::Mat weighted_img(cv::Size(g[0], g[1]), CV_32FC4);
    for (int ii = 0; ii < g[1]; ++ii)
        for (int jj = 0; jj < g[0]; ++jj) {
            float* d = weighted_img.ptr<float>(ii) + 4 * jj; const uchar* s = _right_img.ptr<uchar>(ii) + 3 * jj;
            d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = 1.0f;
        }
    cv::Mat M = translate_top_left_mat * Hinv;
    cv::Mat warped_img;
    dr3hip::warpPerspective(weighted_img, warped_img, M, cv::Size(new_width, new_height), CV_INTER_LINEAR);
    wr_mat(fo, warped_img);

    // ---- src/utils.cpp:189-201 -----------------------------------------------------------------------------------------
    cv::Mat u(g[1], g[0], CV_32FC1), v(g[1], g[0], CV_32FC1);
    rd(fi, u.data, (size_t)g[0] * g[1] * 4); rd(fi, v.data, (size_t)g[0] * g[1] * 4);
    cv::Mat img = _right_img, warped;
    dr3hip::remap(img, warped, u, v, CV_INTER_LINEAR);
    wr_mat(fo, warped.clone());
    float f; rd(fi, &f, 4);
    wr_mat(fo, dr3hip::warp_spherical(img, f));
    wr_mat(fo, dr3hip::warp_cylindrical(img, f));
    std::fclose(fi); std::fclose(fo);
    return 0;
}
