// ref_halfsample_driver.cpp -- TEST INFRASTRUCTURE ONLY, build container only.  C wrapper around the reference's own
// halfSampleSSE2 (reference src/utils.cpp:324-349): the one function on the hot path that needs nothing but <emmintrin.h>, so it is the
// one piece of the reference that can be compiled and EXECUTED here.  oracle/Makefile's `ref` target extracts the function's text from
// /root/reference into oracle/_ref/halfsample_sse2.inc, compiles this file around it into oracle/_ref/libdr3ref_halfsample.so and
// deletes the extracted text again: no reference source enters the repository or travels to the GPU box.  The outputs it produces on
// committed inputs are frozen in tests/golden/halfsample_sse2_ref.npz (tests/golden/make_halfsample_ref.py).
#include <emmintrin.h>

#include "_ref/halfsample_sse2.inc"      // void halfSampleSSE2(const unsigned char* in, unsigned char* out, int w, int h)

extern "C" void dr3ref_halfsample_sse2(const unsigned char* in, unsigned char* out, int w, int h)
{
    halfSampleSSE2(in, out, w, h);
}
