// hotpath_harness.cpp -- C++14 harness that reproduces the reference's call order around the dense
// hot path with dr3hip:: in place of cv:: (include/dr3hip/compat.hpp).  Not a port of the reference's
// control plane: only the three call sequences the hot path sits in.
//   (1) init::Init::process_second_frame, src/initialization.cpp:587-635 (and tests/test_init_generalized.cpp):
//       LK with the literals 30x30 / maxLevel 4 / COUNT+EPS(1000, 1e-3) / USE_INITIAL_FLOW and an identity
//       initial guess (:578), then the erase-if-!status loop (:621-635).
//   (2) reconstruct::Panorama::add_img_to_canvas + normalize_canvas, src/panorama.cpp:156-229:
//       32FC4 (B,G,R,1) image, warpPerspective onto the canvas, feather multiply + accumulate, normalise.
//   (3) Frame::Frame -> utils::create_img_pyramid(img, 3, pyr), src/frame.cpp:18.
// Usage: hotpath_harness <in.bin> <out.bin>   (both written/read by tests/test_harness.py)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "dr3hip/compat.hpp"

namespace {

template <class T> void rd(FILE* f, T* p, size_t n) { if (fread(p, sizeof(T), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); } }
template <class T> void wr(FILE* f, const T* p, size_t n) { if (fwrite(p, sizeof(T), n, f) != n) { fprintf(stderr, "short write\n"); exit(2); } }

// include/utils.hpp:37-46
std::vector<float> linspace(float a, float b, size_t N)
{
    float h = (b - a) / static_cast<float>(N - 1);
    std::vector<float> xs(N);
    float val = a;
    for (auto& x : xs) { x = val; val += h; }
    return xs;
}

}  // namespace

int main(int argc, char** argv)
{
    if (argc != 3) { fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]); return 2; }
    FILE* fi = fopen(argv[1], "rb");
    FILE* fo = fopen(argv[2], "wb");
    if (!fi || !fo) { perror("open"); return 2; }
    try {
        // ---- (1) two-frame initialisation ------------------------------------------------------
        int hdr[3];
        rd(fi, hdr, 3);
        const int w = hdr[0], h = hdr[1], n = hdr[2];
        std::vector<unsigned char> ref((size_t)w * h), cur((size_t)w * h);
        rd(fi, ref.data(), ref.size()); rd(fi, cur.data(), cur.size());
        std::vector<dr3hip::Point2f> kps_ref((size_t)n);
        rd(fi, reinterpret_cast<float*>(kps_ref.data()), (size_t)2 * n);
        std::vector<dr3hip::Point2f> kps_cur(kps_ref);                       // src/initialization.cpp:578

        const int klt_win_size = 30, klt_max_iter = 1000;                    // :593-595
        const double klt_eps = 1e-3;
        std::vector<unsigned char> status;
        std::vector<float> error;
        dr3hip::TermCriteria termcrit(dr3hip::TermCriteria::COUNT + dr3hip::TermCriteria::EPS, klt_max_iter, klt_eps);
        dr3hip::MatView mref(ref.data(), h, w, (size_t)w, DR3HIP_8UC1), mcur(cur.data(), h, w, (size_t)w, DR3HIP_8UC1);
        dr3hip::calcOpticalFlowPyrLK(mref, mcur, kps_ref, kps_cur, status, error,
                                     dr3hip::Size(klt_win_size, klt_win_size), 4, termcrit,
                                     dr3hip::OPTFLOW_USE_INITIAL_FLOW);      // :608-613
        // :621-635 -- drop the points the tracker lost
        std::vector<float> disparities;
        size_t outlier_count = 0;
        auto r_it = kps_ref.begin(); auto c_it = kps_cur.begin();
        for (size_t i = 0; r_it != kps_ref.end(); ++i) {
            if (!status[i]) { r_it = kps_ref.erase(r_it); c_it = kps_cur.erase(c_it); ++outlier_count; continue; }
            disparities.push_back(std::sqrt((r_it->x - c_it->x) * (r_it->x - c_it->x) + (r_it->y - c_it->y) * (r_it->y - c_it->y)));
            ++r_it; ++c_it;
        }
        const int n_kept = (int)kps_cur.size();
        wr(fo, &n_kept, 1);
        wr(fo, reinterpret_cast<const float*>(kps_ref.data()), (size_t)2 * n_kept);
        wr(fo, reinterpret_cast<const float*>(kps_cur.data()), (size_t)2 * n_kept);
        wr(fo, status.data(), (size_t)n);
        wr(fo, error.data(), (size_t)n);

        // ---- (2) panorama paste ------------------------------------------------------------------
        int ph[5];
        rd(fi, ph, 5);
        const int tw = ph[0], th = ph[1], W = ph[2], H = ph[3], feather = ph[4];
        std::vector<unsigned char> img((size_t)tw * th * 3);
        rd(fi, img.data(), img.size());
        float M[9]; rd(fi, M, 9);
        int bbox[2]; rd(fi, bbox, 2);                                          // min_x, max_x from set_bbox (:113-142)
        const int min_x = bbox[0], max_x = bbox[1];
        int temp_blendwidth = feather;                                         // :165-169
        if (max_x - min_x - 2 * feather < 0) temp_blendwidth = (max_x - min_x) / 2 - 1;
        std::vector<float> left_blend = linspace(0.f, 1.f, (size_t)temp_blendwidth);     // :171-176
        std::vector<float> right_blend = linspace(1.f, 0.f, (size_t)temp_blendwidth);
        std::vector<float> mid_blend((size_t)(max_x - min_x - 2 * temp_blendwidth), 1.0f);
        left_blend.insert(left_blend.end(), mid_blend.begin(), mid_blend.end());
        left_blend.insert(left_blend.end(), right_blend.begin(), right_blend.end());

        dr3hip::Image weighted_img;                                            // :178-190
        weighted_img.create(th, tw, DR3HIP_32FC4);
        for (int ii = 0; ii < th; ++ii) {
            float* row = reinterpret_cast<float*>(weighted_img.ptr(ii));
            for (int jj = 0; jj < tw; ++jj) {
                row[4 * jj + 0] = (float)img[((size_t)ii * tw + jj) * 3 + 0];
                row[4 * jj + 1] = (float)img[((size_t)ii * tw + jj) * 3 + 1];
                row[4 * jj + 2] = (float)img[((size_t)ii * tw + jj) * 3 + 2];
                row[4 * jj + 3] = 1.0f;
            }
        }
        dr3hip::Image warped_img;
        dr3hip::warpPerspective(weighted_img.view(), warped_img, M, dr3hip::Size(W, H), dr3hip::INTER_LINEAR);   // :192

        std::vector<float> weighted_panorama((size_t)W * H * 4, 0.f);          // :145
        for (int i = min_x; i != max_x; ++i) {                                 // :194-212 (column-major walk)
            if (i < 0 || i >= W) continue;
            for (int j = 0; j < H; ++j) {
                float* wp = reinterpret_cast<float*>(warped_img.ptr(j)) + 4 * i;
                float* cp = &weighted_panorama[((size_t)j * W + i) * 4];
                wp[0] *= left_blend[i - min_x]; wp[1] *= left_blend[i - min_x]; wp[2] *= left_blend[i - min_x];
                wp[3] = left_blend[i - min_x];
                if (wp[0] == 0 && wp[1] == 0 && wp[2] == 0) { wp[3] = 0.0f; }
                else { cp[0] += wp[0]; cp[1] += wp[1]; cp[2] += wp[2]; cp[3] += wp[3]; }
            }
        }
        std::vector<unsigned char> final_panorama((size_t)W * H * 3, 0);       // :215-229
        for (int i = 0; i < H; ++i)
            for (int j = 0; j < W; ++j) {
                const float alpha_sum = weighted_panorama[((size_t)i * W + j) * 4 + 3];
                if (alpha_sum > 0.0f)
                    for (int c = 0; c < 3; ++c)
                        final_panorama[((size_t)i * W + j) * 3 + c] =
                            (unsigned char)(int)((int)weighted_panorama[((size_t)i * W + j) * 4 + c] / alpha_sum);
            }
        wr(fo, final_panorama.data(), final_panorama.size());

        // ---- (2b) the same paste as the library's fused device path: paste_images (:143-154) keeps the
        // weighted canvas on the GPU, add_img_to_canvas is one kernel, normalize_canvas another
        {
            dr3hip::WeightedCanvas weighted_canvas(dr3hip::Size(W, H));       // :144
            weighted_canvas.add_img_to_canvas(dr3hip::MatView(img.data(), th, tw, (size_t)tw * 3, DR3HIP_8UC3), M, feather);   // :149
            dr3hip::Image fused_panorama;
            weighted_canvas.normalize(fused_panorama);                          // :67
            wr(fo, fused_panorama.buf.data(), fused_panorama.buf.size());
        }

        // ---- (3) Frame constructor pyramid ------------------------------------------------------
        std::vector<dr3hip::Image> pyr;
        dr3hip::create_img_pyramid(mref, 3, pyr);                              // src/frame.cpp:18, n_pyr_levels = 3 (src/config.cpp)
        for (int l = 1; l < 3; ++l) {
            int dims[2] = {pyr[l].cols, pyr[l].rows};
            wr(fo, dims, 2);
            wr(fo, pyr[l].buf.data(), pyr[l].buf.size());
        }
        // ---- (4) the rest of the C++14 layer, in the order the reference's demo mains use it ---------------------
        // tests/test_stitch.cpp -> Stitch::process (src/stitch.cpp:33-80) with the homography align_pair would return
        {
            float Hs[9]; rd(fi, Hs, 9);
            float Hinv[9];
            dr3hip::Image stitched;
            dr3hip::MatView left(img.data(), th, tw, (size_t)tw * 3, DR3HIP_8UC3);       // the panorama tile doubles as both images
            dr3hip::stitch_compose(left, left, Hs, Hinv, stitched);
            int dims[2] = {stitched.cols, stitched.rows};
            wr(fo, dims, 2); wr(fo, Hinv, 9);
            wr(fo, stitched.buf.data(), stitched.buf.size());
        }
        // tests/test_panorama.cpp spherical pre-warp: utils::warp_spherical(img, f) (src/utils.cpp:196-201)
        {
            float focal; rd(fi, &focal, 1);
            dr3hip::Image sph;
            dr3hip::warp_spherical(dr3hip::MatView(img.data(), th, tw, (size_t)tw * 3, DR3HIP_8UC3), focal, sph);
            wr(fo, sph.buf.data(), sph.buf.size());
        }
        // Frame::Frame -> FastDetector::detect (src/frame.cpp:57): cell 30, 3 levels, min_harris_corner_score 20 (src/config.cpp:10-12)
        {
            dr3hip::FastDetector detector(w, h, 30, 3);
            detector.flag_grid(100.0, 50.0);                                               // an existing feature occupies its cell (:29-33)
            std::vector<dr3hip::Corner> corners;
            detector.detect(mref, 20.0, corners);
            const int nc = (int)corners.size();
            wr(fo, &nc, 1);
            for (const auto& c : corners) { int v[3] = {c.x, c.y, c.level}; wr(fo, v, 3); wr(fo, &c.score, 1); }
        }
        fprintf(stderr, "harness: %d/%d points tracked, %zu outliers, canvas %dx%d\n", n_kept, n, outlier_count, W, H);
    } catch (const dr3hip::Error& e) {
        fprintf(stderr, "dr3hip error %d: %s\n", e.code, e.what());
        return 1;
    }
    fclose(fi); fclose(fo);
    return 0;
}
