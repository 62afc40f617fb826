/*
 * sse2_replay.c -- TEST INFRASTRUCTURE ONLY (x86-64).  The accumulation sequences of LKTrackerInvoker's CV_SSE2 blocks (OpenCV 3.0 -
 * 3.4 modules/video/src/lkpyramid.cpp, the code behind cv::calcOpticalFlowPyrLK at reference src/initialization.cpp:608-613), replayed
 * with the real <emmintrin.h> intrinsics on given int16 gradients / mismatches.  Restated from the published source (OpenCV is absent
 * from this image): the unpack / madd / convert / add sequence and the final horizontal adds, nothing else.  tests/test_oracle_sse2.py
 * holds dr3_oracle.c's DR3O_ORDER_SSE2 bookkeeping (plain C, lane indices worked out by hand) against what the instructions really do.
 *
 * Build: gcc -O2 -msse2 -ffp-contract=off -shared -fPIC -o libsse2_replay.so sse2_replay.c
 */
#include <emmintrin.h>
#include <stdint.h>

/* template loop: per row, x-chunks of 4 read the interleaved (Ix, Iy) int16 pairs OpenCV has just stored to dIptr */
void sse2_replay_A(const int16_t* dI, int win_w, int win_h, float A[3])
{
    __m128 qA11 = _mm_setzero_ps(), qA12 = _mm_setzero_ps(), qA22 = _mm_setzero_ps();
    float iA11 = 0.f, iA12 = 0.f, iA22 = 0.f;
    float buf11[4], buf12[4], buf22[4];
    int x, y;
    for (y = 0; y < win_h; ++y) {
        const int16_t* dIptr = dI + 2 * y * win_w;
        x = 0;
        for (; x <= win_w - 4; x += 4, dIptr += 4 * 2) {
            __m128i v00 = _mm_loadu_si128((const __m128i*)dIptr);          /* Ix0 Iy0 Ix1 Iy1 ... */
            __m128i t0 = _mm_srai_epi32(v00, 16);                          /* Iy0 Iy1 Iy2 Iy3 */
            __m128i t1 = _mm_srai_epi32(_mm_slli_epi32(v00, 16), 16);      /* Ix0 Ix1 Ix2 Ix3 */
            __m128 fy = _mm_cvtepi32_ps(t0);
            __m128 fx = _mm_cvtepi32_ps(t1);
            qA22 = _mm_add_ps(qA22, _mm_mul_ps(fy, fy));
            qA12 = _mm_add_ps(qA12, _mm_mul_ps(fx, fy));
            qA11 = _mm_add_ps(qA11, _mm_mul_ps(fx, fx));
        }
        for (; x < win_w; ++x, dIptr += 2) {
            const int ixval = dIptr[0], iyval = dIptr[1];
            iA11 += (float)(ixval * ixval);
            iA12 += (float)(ixval * iyval);
            iA22 += (float)(iyval * iyval);
        }
    }
    _mm_storeu_ps(buf11, qA11); _mm_storeu_ps(buf12, qA12); _mm_storeu_ps(buf22, qA22);
    iA11 += buf11[0] + buf11[1] + buf11[2] + buf11[3];
    iA12 += buf12[0] + buf12[1] + buf12[2] + buf12[3];
    iA22 += buf22[0] + buf22[1] + buf22[2] + buf22[3];
    A[0] = iA11; A[1] = iA12; A[2] = iA22;
}

/* iteration loop: per row, x-chunks of 8; diff = the packed int16 mismatches (OpenCV: _mm_subs_epi16 of the blended J row and Iptr) */
void sse2_replay_B(const int16_t* diff, const int16_t* dI, int win_w, int win_h, float b[2])
{
    __m128 qb0 = _mm_setzero_ps(), qb1 = _mm_setzero_ps();
    float ib1 = 0.f, ib2 = 0.f;
    float bbuf[4];
    int x, y;
    for (y = 0; y < win_h; ++y) {
        const int16_t* dIptr = dI + 2 * y * win_w;
        const int16_t* dptr = diff + y * win_w;
        x = 0;
        for (; x <= win_w - 8; x += 8, dIptr += 8 * 2) {
            __m128i diff0 = _mm_loadu_si128((const __m128i*)(dptr + x)), diff1;
            __m128i v00, v01, v10, v11;
            diff1 = _mm_unpackhi_epi16(diff0, diff0);
            diff0 = _mm_unpacklo_epi16(diff0, diff0);                      /* It0 It0 It1 It1 ... */
            v00 = _mm_loadu_si128((const __m128i*)(dIptr));                /* Ix0 Iy0 Ix1 Iy1 ... */
            v01 = _mm_loadu_si128((const __m128i*)(dIptr + 8));
            v10 = _mm_unpacklo_epi16(v00, v01);
            v11 = _mm_unpackhi_epi16(v00, v01);
            v00 = _mm_unpacklo_epi16(diff0, diff1);
            v01 = _mm_unpackhi_epi16(diff0, diff1);
            v00 = _mm_madd_epi16(v00, v10);
            v11 = _mm_madd_epi16(v01, v11);
            qb0 = _mm_add_ps(qb0, _mm_cvtepi32_ps(v00));
            qb1 = _mm_add_ps(qb1, _mm_cvtepi32_ps(v11));
        }
        for (; x < win_w; ++x, dIptr += 2) {
            const int d = dptr[x];
            ib1 += (float)(d * dIptr[0]);
            ib2 += (float)(d * dIptr[1]);
        }
    }
    _mm_storeu_ps(bbuf, _mm_add_ps(qb0, qb1));
    ib1 += bbuf[0] + bbuf[2];
    ib2 += bbuf[1] + bbuf[3];
    b[0] = ib1; b[1] = ib2;
}
