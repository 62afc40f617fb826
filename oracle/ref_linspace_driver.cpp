// ref_linspace_driver.cpp -- TEST INFRASTRUCTURE ONLY, build container only.  C wrapper around the reference's own utils::linspace
// (reference include/utils.hpp:36-46), the template behind the feather ramps of Panorama::add_img_to_canvas (src/panorama.cpp:171-172):
// its body needs nothing but <vector>, so it can be compiled and EXECUTED here.  oracle/Makefile's `ref` target extracts the function's
// text from /root/reference into oracle/_ref/linspace.inc, compiles this file around it and deletes the extracted text again.
#include <cstddef>
#include <vector>

namespace utils {
template <typename T>
#include "_ref/linspace.inc"             // std::vector<T> linspace(T a, T b, size_t N) { ... }
}

extern "C" void dr3ref_linspace_f32(float a, float b, int n, float* out)
{
    const std::vector<float> v = utils::linspace<float>(a, b, (size_t)n);
    for (int i = 0; i < n; ++i) out[i] = v[i];
}
