// align_kernels.hip -- 6-DoF inverse-compositional sparse image alignment (gfx950), SURVEY 8(f) row 5.
// EXTENSION SPECIFIED BY THIS BUILD: the reference only names the step (timer "sparse_img_align" at src/handler.cpp:23, empty
// HandlerMono::process_frame at :80-82, README.md TODO), so there is nothing to be in parity with; the specification is
// DESIGN.md section 4.8 (after Forster, Pizzoli, Scaramuzza, "SVO", ICRA 2014, section IV-A), restated on the CPU by the test
// infrastructure, which this kernel equals bit for bit.
//
// Mapping: one 256-lane workgroup per frame pair; a lane owns the features lane, lane + 256, ... for the whole kernel.
//   per level, once : the lane takes the 7x7 reference pixels around each of its features, interpolates the 6x6 grid the 4x4
//                     patch and its central differences need, multiplies the gradients with the 2x6 projection Jacobian and
//                     parks [I_ref, J0..J5] of the 16 patch pixels in a per-pair workspace in HBM (written and read by the same
//                     lane only; laid out so that the 256 lanes of a wave-instruction touch consecutive floats);
//   per iteration   : project the feature with the current pose, sample the 16 pixels of the current image, accumulate the 21
//                     entries of J J^T, the 6 of J r, r^2 and the count per lane (f32, fixed order); reduce with the tracker's
//                     wave tree (wave_tree.h) and across the four waves through LDS; EVERY lane then solves the 6x6 system
//                     (f64 LDL^T) and updates its own copy of the pose -- no broadcast, no divergence.
// There is no dense contraction here (the "matrix" is 6x6): no MFMA.  Arithmetic: f32 per pixel, no contraction
// (-ffp-contract=off), IEEE division; the se(3) exponential is a fixed 6-term series in theta^2 so that no libm is involved.
#include "dr3hip_internal.h"
#include "wave_tree.h"

namespace dr3 {
namespace {

constexpr int AL_LANES = 256, AL_NSUM = 28, AL_BORDER = 3;

__device__ __forceinline__ float al_bilin(const uint8_t* __restrict__ p, int pitch, const float wt[4])
{
    return ((wt[0] * (float)p[0] + wt[1] * (float)p[1]) + wt[2] * (float)p[pitch]) + wt[3] * (float)p[pitch + 1];
}

__device__ __forceinline__ bool al_project(float X, float Y, float Z, const AlignArgs& a, float scale, int w, int h,
                                           int& ui, int& vi, float wt[4], float& xn, float& yn)
{
    if (!(Z > 1e-6f)) return false;
    const float x = X / Z, y = Y / Z;
    const float u = (a.fx * x + a.cx) * scale, v = (a.fy * y + a.cy) * scale;
    if (!(fabsf(u) < 1.0e6f) || !(fabsf(v) < 1.0e6f)) return false;
    const float fu = floorf(u), fv = floorf(v);
    ui = (int)fu; vi = (int)fv;
    if (ui - AL_BORDER < 0 || vi - AL_BORDER < 0 || ui + AL_BORDER >= w || vi + AL_BORDER >= h) return false;
    const float sx = u - fu, sy = v - fv;
    wt[0] = (1.f - sx) * (1.f - sy); wt[1] = sx * (1.f - sy); wt[2] = (1.f - sx) * sy; wt[3] = sx * sy;
    xn = x; yn = y;
    return true;
}

// T <- T * exp(-x), x = (translation part, rotation part); every operation in the order DESIGN.md 4.8 fixes
__device__ __forceinline__ void al_update(double T[12], const double x[6])
{
    const double u0 = -x[0], u1 = -x[1], u2 = -x[2], w0 = -x[3], w1 = -x[4], w2 = -x[5];
    const double th2 = (w0 * w0 + w1 * w1) + w2 * w2;
    const double A = 1.0 - th2 * (1.0 / 6.0) * (1.0 - th2 * (1.0 / 20.0) * (1.0 - th2 * (1.0 / 42.0) * (1.0 - th2 * (1.0 / 72.0) * (1.0 - th2 * (1.0 / 110.0)))));
    const double B = 0.5 * (1.0 - th2 * (1.0 / 12.0) * (1.0 - th2 * (1.0 / 30.0) * (1.0 - th2 * (1.0 / 56.0) * (1.0 - th2 * (1.0 / 90.0) * (1.0 - th2 * (1.0 / 132.0))))));
    const double C = (1.0 / 6.0) * (1.0 - th2 * (1.0 / 20.0) * (1.0 - th2 * (1.0 / 42.0) * (1.0 - th2 * (1.0 / 72.0) * (1.0 - th2 * (1.0 / 110.0) * (1.0 - th2 * (1.0 / 156.0))))));
    const double W[9] = {0.0, -w2, w1, w2, 0.0, -w0, -w1, w0, 0.0};
    const double W2[9] = {w0 * w0 - th2, w0 * w1, w0 * w2, w1 * w0, w1 * w1 - th2, w1 * w2, w2 * w0, w2 * w1, w2 * w2 - th2};
    double Re[9], V[9], te[3], Rn[9], tn[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const double id = (i == 0 || i == 4 || i == 8) ? 1.0 : 0.0;
        Re[i] = (id + A * W[i]) + B * W2[i];
        V[i] = (id + B * W[i]) + C * W2[i];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) te[i] = (V[3 * i] * u0 + V[3 * i + 1] * u1) + V[3 * i + 2] * u2;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) Rn[3 * i + j] = (T[4 * i] * Re[j] + T[4 * i + 1] * Re[3 + j]) + T[4 * i + 2] * Re[6 + j];
        tn[i] = ((T[4 * i] * te[0] + T[4 * i + 1] * te[1]) + T[4 * i + 2] * te[2]) + T[4 * i + 3];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) { T[4 * i] = Rn[3 * i]; T[4 * i + 1] = Rn[3 * i + 1]; T[4 * i + 2] = Rn[3 * i + 2]; T[4 * i + 3] = tn[i]; }
}

// H x = -g by LDL^T without pivoting (operation order fixed by DESIGN.md 4.8); false = not positive definite / not finite
__device__ __forceinline__ bool al_solve(const float* __restrict__ tot, double x[6])
{
    double H[6][6], L[6][6], D[6], z[6];
    int idx = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = i; j < 6; ++j) { H[i][j] = (double)tot[idx]; H[j][i] = H[i][j]; ++idx; }
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double d = H[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= (L[j][k] * L[j][k]) * D[k];
        if (!(d > 0.0)) ok = false;
        D[j] = d;
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
            double s = H[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) s -= (L[i][k] * L[j][k]) * D[k];
            L[i][j] = s / d;
        }
    }
    if (!ok) return false;
#pragma unroll
    for (int i = 0; i < 6; ++i) { double s = -(double)tot[21 + i];
#pragma unroll
        for (int k = 0; k < i; ++k) s -= L[i][k] * z[k];
        z[i] = s; }
#pragma unroll
    for (int i = 0; i < 6; ++i) z[i] = z[i] / D[i];
#pragma unroll
    for (int i = 5; i >= 0; --i) { double s = z[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k) s -= L[k][i] * x[k];
        x[i] = s; }
#pragma unroll
    for (int i = 0; i < 6; ++i) if (!(fabs(x[i]) < 1.0e300)) ok = false;
    return ok;
}

__global__ __launch_bounds__(AL_LANES) void sparse_align_kernel(const AlignArgs a)
{
    __shared__ float red[4][AL_NSUM + 1];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, pair = blockIdx.x;
    const int rs = a.ref_slot ? a.ref_slot[pair] : a.first_slot + pair;
    const int cs = a.cur_slot ? a.cur_slot[pair] : a.first_slot + pair + 1;
    const float* __restrict__ xyz = a.xyz + 3ll * a.n_pts * pair;
    const int n = a.n_pts, K = (n + AL_LANES - 1) / AL_LANES;
    // this lane's slice of the pair's workspace: entry ((k * 16 + p) * 7 + c) * 256 + tid
    float* __restrict__ ws = a.cache + (long long)pair * K * (16 * 7 * AL_LANES) + tid;

    double T[12], T_old[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) { T[i] = a.T[12ll * pair + i]; T_old[i] = T[i]; }
    int n_meas = 0;
    float chi2_last = 0.f;
    if (a.iters && tid < DR3HIP_MAX_LEVELS) a.iters[(long long)pair * DR3HIP_MAX_LEVELS + tid] = 0;     // levels that do not run report 0

    for (int level = a.max_level; level >= a.min_level; --level) {
        const LevelDev& L = a.lv[level];
        const uint8_t* __restrict__ ref = L.img + (long long)rs * L.step;
        const uint8_t* __restrict__ cur = L.img + (long long)cs * L.step;
        const int lw = L.w, lh = L.h, pitch = L.pitch;
        const float scale = __int_as_float((127 - level) << 23);
        const float fxl = a.fx * scale, fyl = a.fy * scale;
        int it_done = 0;

        // ---- reference patches and their Jacobians, once per level -----------------------------------------
        for (int k = 0; k < K; ++k) {
            const int f = tid + AL_LANES * k;
            float* __restrict__ c = ws + (long long)k * (16 * 7 * AL_LANES);
            bool vis = false;
            if (f < n) {
                const float X = xyz[3 * f], Y = xyz[3 * f + 1], Z = xyz[3 * f + 2];
                int ui, vi; float wt[4], x, y;
                vis = al_project(X, Y, Z, a, scale, lw, lh, ui, vi, wt, x, y);
                if (vis) {
                    float S[6][6];
#pragma unroll
                    for (int r = 0; r < 6; ++r)
#pragma unroll
                        for (int cc = 0; cc < 6; ++cc)
                            S[r][cc] = al_bilin(ref + (long long)(vi + r - 3) * pitch + (ui + cc - 3), pitch, wt);
                    const float zi = 1.0f / Z;
                    const float j0[6] = {-zi, 0.f, x * zi, x * y, -(1.0f + x * x), y};
                    const float j1[6] = {0.f, -zi, y * zi, 1.0f + y * y, -(x * y), -x};
#pragma unroll
                    for (int py = 0; py < 4; ++py)
#pragma unroll
                        for (int px = 0; px < 4; ++px) {
                            const float dx = 0.5f * (S[py + 1][px + 2] - S[py + 1][px]);
                            const float dy = 0.5f * (S[py + 2][px + 1] - S[py][px + 1]);
                            const float gx = dx * fxl, gy = dy * fyl;
                            float* __restrict__ e = c + (long long)(4 * py + px) * (7 * AL_LANES);
                            e[0] = S[py + 1][px + 1];
#pragma unroll
                            for (int q = 0; q < 6; ++q) e[(1 + q) * AL_LANES] = gx * j0[q] + gy * j1[q];
                        }
                }
            }
            if (f < n && !vis) c[0] = __int_as_float(0x7fc00000);      // NaN in I_ref of pixel 0 = "not visible in the reference"
        }

        // ---- Gauss-Newton: all lanes carry the same pose and take the same decisions -------------------------
        float chi2_prev = 0.f;
        for (int it = 0; it < a.n_iter; ++it) {
            float acc[AL_NSUM + 1];
#pragma unroll
            for (int s = 0; s <= AL_NSUM; ++s) acc[s] = 0.f;
            float R32[9], t32[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) { R32[3 * i] = (float)T[4 * i]; R32[3 * i + 1] = (float)T[4 * i + 1]; R32[3 * i + 2] = (float)T[4 * i + 2]; t32[i] = (float)T[4 * i + 3]; }
            for (int k = 0; k < K; ++k) {
                const int f = tid + AL_LANES * k;
                if (f >= n) continue;
                const float* __restrict__ c = ws + (long long)k * (16 * 7 * AL_LANES);
                const float i0 = c[0];
                if (i0 != i0) continue;
                const float X = xyz[3 * f], Y = xyz[3 * f + 1], Z = xyz[3 * f + 2];
                const float Xc = ((R32[0] * X + R32[1] * Y) + R32[2] * Z) + t32[0];
                const float Yc = ((R32[3] * X + R32[4] * Y) + R32[5] * Z) + t32[1];
                const float Zc = ((R32[6] * X + R32[7] * Y) + R32[8] * Z) + t32[2];
                int ui, vi; float wt[4], xn, yn;
                if (!al_project(Xc, Yc, Zc, a, scale, lw, lh, ui, vi, wt, xn, yn)) continue;
#pragma unroll 4
                for (int p = 0; p < 16; ++p) {
                    const int py = p >> 2, px = p & 3;
                    const float* __restrict__ e = c + (long long)p * (7 * AL_LANES);
                    const float Ic = al_bilin(cur + (long long)(vi + py - 2) * pitch + (ui + px - 2), pitch, wt);
                    const float r = Ic - e[0];
                    float J[6];
#pragma unroll
                    for (int q = 0; q < 6; ++q) J[q] = e[(1 + q) * AL_LANES];
                    int idx = 0;
#pragma unroll
                    for (int q = 0; q < 6; ++q)
#pragma unroll
                        for (int b = q; b < 6; ++b) { acc[idx] += J[q] * J[b]; ++idx; }
#pragma unroll
                    for (int q = 0; q < 6; ++q) acc[21 + q] += J[q] * r;
                    acc[27] += r * r;
                    acc[28] += 1.0f;                                    // the count: exact in f32 (< 2^24), any summation order
                }
            }
            // wave tree, then waves 0..3 in order (wave_tree.h, then ((w0 + w1) + w2) + w3)
            __syncthreads();                                            // red[] of the previous iteration has been read by everyone
#pragma unroll
            for (int s = 0; s <= AL_NSUM; ++s) {
                const float v = wave_sum1(acc[s]);
                if (lane == 0) red[wv][s] = v;
            }
            __syncthreads();
            float tot[AL_NSUM + 1];
#pragma unroll
            for (int s = 0; s <= AL_NSUM; ++s) tot[s] = ((red[0][s] + red[1][s]) + red[2][s]) + red[3][s];
            n_meas = (int)tot[28];
            if (n_meas == 0) break;
            const float chi2m = tot[27] / (float)n_meas;
            chi2_last = chi2m;
            if (it > 0 && chi2m > chi2_prev) {
#pragma unroll
                for (int i = 0; i < 12; ++i) T[i] = T_old[i];
                break;
            }
            double x[6];
            if (!al_solve(tot, x)) break;
#pragma unroll
            for (int i = 0; i < 12; ++i) T_old[i] = T[i];
            al_update(T, x);
            chi2_prev = chi2m;
            it_done = it + 1;
            double mx = 0.0;
#pragma unroll
            for (int i = 0; i < 6; ++i) if (fabs(x[i]) > mx) mx = fabs(x[i]);
            if (mx < a.eps) break;
        }
        if (a.iters && tid == 0) a.iters[(long long)pair * DR3HIP_MAX_LEVELS + level] = it_done;
    }
    if (tid == 0) {
#pragma unroll
        for (int i = 0; i < 12; ++i) a.T[12ll * pair + i] = T[i];
        if (a.n_meas) a.n_meas[pair] = n_meas;
        if (a.chi2) a.chi2[pair] = chi2_last;
    }
}

}  // namespace

hipError_t launch_sparse_align(const AlignArgs& a, int n_pairs, hipStream_t s)
{
    if (n_pairs <= 0) return hipSuccess;
    hipLaunchKernelGGL(sparse_align_kernel, dim3(n_pairs), dim3(AL_LANES), 0, s, a);
    return hipGetLastError();
}

}  // namespace dr3
