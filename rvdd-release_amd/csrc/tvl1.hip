// Dual TV-L1 optical flow on the GPU: the reference's flow pre-computation
// (libBridge.cpp:44-163 -> 3rdparty/tvl1flow/tvl1flow_lib.c:91-278, 343-472; mask.c; zoom.c;
// bicubic_interpolation.c), SURVEY.md section 8f rank 1.  Same algorithm, same hard-wired parameters
// (tau .25, lambda .15, theta .3, zoom .5, 5 warps, <= 300 iterations, epsilon .01), same quirks
// (listed in LABBOOK.md 4.4), fp32 with the reference's double-precision islands (Gaussian,
// bicubic cell, normalisation, hypot).  Compiled with -ffp-contract=off.
//
// All maps are planar fp32 [ny][nx].  One scale of the pyramid = ONE persistent cooperative kernel: gradient of I1,
// then 5 x [warp of I1 / I1x / I1y, then <= 300 x (u update | exchange | convergence test + p update)].  Three forms,
// bit-identical (the convergence sum is formed in fixed point, see err_fix):
//   scale_kernel_patch  the default: a block owns a 64 x PH patch, pixel state in registers, neighbours through LDS, the
//                       patch's perimeter and the blocks' convergence sums through tagged records that the readers poll --
//                       no grid barrier in the iteration, the convergence test one iteration late (4.4 us per iteration of
//                       two 640x360 pairs, against 13 with scale_kernel);
//   scale_kernel        interleaved row segments, one grid barrier per iteration (round 3's kernel; RVDD_TVL1_PATCH=0,
//                       and images whose patches do not fit the CUs);
//   scale_kernel_mem    pixel state in memory, two grid barriers per iteration (images beyond the register slots).
// As separate launches (three per iteration, the first version of this file) the path was bound by launch gaps.
#include "rvdd_internal.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace {

constexpr float kTau = 0.25f, kLambda = 0.15f, kTheta = 0.3f, kZoom = 0.5f, kEps = 0.01f;
constexpr int kWarps = 5, kMaxIter = 300, kMaxScales = 100;
constexpr int kCtlWords = 4 + kMaxScales;      // per pair: control words, then a grid barrier counter per scale
constexpr double kPresmooth = 0.8, kZoomSigma0 = 0.6;
constexpr float kGradIsZero = 1e-10f;

// ---------------------------------------------------------------- bicubic_interpolation.c --
__device__ __forceinline__ int neumann(int x, int n, bool& out) {
    if (x < 0) { out = true; return 0; }
    if (x >= n) { out = true; return n - 1; }
    return x;
}
__device__ __forceinline__ double cubic_cell(double v0, double v1, double v2, double v3, double x) {
    return v1 + 0.5 * x * (v2 - v0 + x * (2.0 * v0 - 5.0 * v1 + 4.0 * v2 - v3 + x * (3.0 * (v1 - v2) + v3 - v0)));
}
struct Bicubic {       // the 4x4 stencil of bicubic_interpolation_at (:133-231) for one coordinate
    int cx[4], cy[4];
    double fx, fy;
    bool out;
    __device__ __forceinline__ void setup(float uu, float vv, int nx, int ny) {
        const int sx = uu < 0 ? -1 : 1, sy = vv < 0 ? -1 : 1;
        const int iu = (int)uu, iv = (int)vv;          // truncation toward zero
        out = false;
        const int x = neumann(iu, nx, out), y = neumann(iv, ny, out);
        cx[0] = neumann(iu - sx, nx, out);
        cy[0] = neumann(iv - sx, ny, out);             // sx, as the reference (:165)
        cx[1] = x;
        cy[1] = y;
        cx[2] = neumann(iu + sx, nx, out);
        cy[2] = neumann(iv + sy, ny, out);
        cx[3] = neumann(iu + 2 * sx, nx, out);
        cy[3] = neumann(iv + 2 * sy, ny, out);
        fx = (double)(uu - (float)x);
        fy = (double)(vv - (float)y);
    }
    __device__ __forceinline__ float eval(const float* __restrict__ img, int nx) const {
        double col[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
            col[c] = cubic_cell((double)img[cx[c] + nx * cy[0]], (double)img[cx[c] + nx * cy[1]],
                                (double)img[cx[c] + nx * cy[2]], (double)img[cx[c] + nx * cy[3]], fy);
        return (float)cubic_cell(col[0], col[1], col[2], col[3], fx);
    }
};

// zoom_out / zoom_in resampling (zoom.c:63-74, 98-107): out(i1,j1) = in at (j1/fx, i1/fy), times mul
// The images of a pair (I0/I1, or the two flow components) go through every pre- and post-processing kernel together:
// blockIdx.z picks the image; the pairs of a batch call share the launches too (2 x kMaxLanes images).
constexpr int kMaxLanes = 8;        // pairs per launch set (the patch kernel; as many of them per scale as are resident together)
constexpr int kBarrierLanes = 2;    // pairs per launch of the kernels with a grid barrier
struct Imgs {
    const float* in[2 * kMaxLanes];
    float* out[2 * kMaxLanes];
};
using Two = Imgs;
__global__ void resample_kernel(Two io, int nx, int ny, int nxx, int nyy, float fx, float fy, float mul) {
    const float* __restrict__ in = io.in[blockIdx.z];
    float* __restrict__ out = io.out[blockIdx.z];
    const int j1 = blockIdx.x * blockDim.x + threadIdx.x, i1 = blockIdx.y;
    if (j1 >= nxx) return;
    Bicubic b;
    b.setup((float)j1 / fx, (float)i1 / fy, nx, ny);
    out[i1 * nxx + j1] = b.eval(in, nx) * mul;
}

// ------------------------------------------------------------------------------ mask.c --
struct GaussK {
    double B[16];
    int size;
};
// one line direction of the in-place Gaussian (mask.c:262-325): reflecting boundary that repeats
// the edge sample on the high side only; double accumulation in the reference's order
__global__ void gauss_kernel(Two io, int nx, int ny, GaussK k, int vertical) {
    const float* __restrict__ in = io.in[blockIdx.z];
    float* __restrict__ out = io.out[blockIdx.z];
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= nx) return;
    const int n = vertical ? ny : nx, pos = vertical ? i : j;
    auto R = [&](int t) -> double {       // R[size + t], t in [-size, n + size)
        int q = t;
        if (t < 0) q = -t;                              // R[i] = I[size - i]
        else if (t >= n) q = 2 * n - 1 - t;             // R[bd + i] = I[n - i - 1]
        return (double)(vertical ? in[q * nx + j] : in[i * nx + q]);
    };
    double sum = k.B[0] * R(pos);
    for (int t = 1; t < k.size; ++t) sum += k.B[t] * (R(pos - t) + R(pos + t));
    out[i * nx + j] = (float)sum;
}

__device__ __forceinline__ void centered_gradient_px(const float* __restrict__ I, float* __restrict__ dx,
                                                     float* __restrict__ dy, int nx, int ny, int i, int j) {
    const int p = i * nx + j;
    const float l = I[j > 0 ? p - 1 : p], r = I[j < nx - 1 ? p + 1 : p];
    const float u = I[i > 0 ? p - nx : p], d = I[i < ny - 1 ? p + nx : p];
    dx[p] = 0.5f * (r - l);          // one-sided at the borders with the SAME 0.5 factor (mask.c:170-205)
    dy[p] = 0.5f * (d - u);
}

// per-pair words of a batch call, pair = blockIdx.y
struct PairWords {
    float* mm[kMaxLanes];          // min / max keys
    int* ctl[kMaxLanes];           // control words, then the grid barrier counters of the scales
    float* top[2 * kMaxLanes];     // the coarsest scale's flow
};
// start of a run: min / max keys at +-FLT_MAX, control words and barrier counters cleared, zero flow at the coarsest scale
// (tvl1flow_lib.c:388-391) -- one launch instead of five fills per pair
__global__ void init_kernel(PairWords pw, int nwords, int ntop) {
    const int q = blockIdx.y;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ntop; i += gridDim.x * blockDim.x) {
        pw.top[2 * q][i] = 0.f;
        pw.top[2 * q + 1][i] = 0.f;
    }
    if (blockIdx.x == 0) {
        if (threadIdx.x < nwords) pw.ctl[q][threadIdx.x] = 0;
        if (threadIdx.x == 0) {
            reinterpret_cast<int*>(pw.mm[q])[0] = 0x7f7fffff;       // order-preserving keys of +FLT_MAX / -FLT_MAX
            reinterpret_cast<int*>(pw.mm[q])[1] = (int)0x80000000;
        }
    }
}
// min / max of both images (image_normalization, tvl1flow_lib.c:300-333)
__global__ void minmax_kernel(Imgs io, int n, PairWords pw) {
    const float* __restrict__ a = io.in[2 * blockIdx.y];
    const float* __restrict__ b = io.in[2 * blockIdx.y + 1];
    float* __restrict__ mm = pw.mm[blockIdx.y];
    float lo = 3.4e38f, hi = -3.4e38f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        lo = fminf(lo, fminf(a[i], b[i]));
        hi = fmaxf(hi, fmaxf(a[i], b[i]));
    }
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_down(lo, o));
        hi = fmaxf(hi, __shfl_down(hi, o));
    }
    if ((threadIdx.x & 63) == 0) {
        // float min/max through the integer atomics on the order-preserving key
        auto key = [](float f) { int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; };
        atomicMin(reinterpret_cast<int*>(mm), key(lo));
        atomicMax(reinterpret_cast<int*>(mm) + 1, key(hi));
    }
}
__global__ void normalize_kernel(Imgs io, int n, PairWords pw) {
    const float* __restrict__ a = io.in[2 * blockIdx.y];
    const float* __restrict__ b = io.in[2 * blockIdx.y + 1];
    float* __restrict__ an = io.out[2 * blockIdx.y];
    float* __restrict__ bn = io.out[2 * blockIdx.y + 1];
    const float* __restrict__ mm = pw.mm[blockIdx.y];
    auto unkey = [](int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); };
    const float mn = unkey(reinterpret_cast<const int*>(mm)[0]), mx = unkey(reinterpret_cast<const int*>(mm)[1]);
    const float den = mx - mn;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (den > 0) {
        an[i] = (float)(255.0 * (double)(a[i] - mn) / (double)den);
        bn[i] = (float)(255.0 * (double)(b[i] - mn) / (double)den);
    } else {
        an[i] = a[i];
        bn[i] = b[i];
    }
}

// --------------------------------------------------------------------- tvl1flow_lib.c --
// Values that cross workgroups INSIDE the persistent kernel (u and p of neighbouring tiles, the per-tile error
// sums) move with device-scope relaxed atomics: `sc1` stores write through to the memory side, `sc1` loads do
// not trust the XCD-local L2.  That keeps the eight L2s coherent for exactly these words, so the per-iteration
// grid barrier needs no L2 write-back / invalidate (which cost ~25 us each when every barrier did them).
__device__ __forceinline__ float ldc(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stc(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct Scale {
    int nx, ny;
    float *I0, *I1, *u1, *u2;
};
struct IterBufs {
    float *I1x, *I1y, *I1w, *p11, *p12, *p21, *p22;      // I1w: scratch of the Gaussian passes
};

// ------------------------------------------------------------- one scale, one kernel --
constexpr unsigned kSpinLimit = 1u << 21;      // ~ seconds; a barrier that is never completed ends the kernel, not the GPU

// Grid-wide barrier on a monotonically increasing arrival counter.  Returns false when it timed out or
// another block reported that (abort word): the caller returns, so the grid always drains.
//   HEAVY: release/acquire at agent scope on every thread (L2 write-back + invalidate across the XCDs) -- for
//          phases whose plain stores are read by other blocks (the gradient maps); once per scale.
//   light: only waits for this wave's (write-through) stores; the data exchanged across it is ldc/stc.
template <bool HEAVY>
__device__ __forceinline__ bool grid_sync(unsigned* count, unsigned& target, unsigned nblk, int* abort_word) {
    __shared__ int ok;
    if (HEAVY) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);                          // vmcnt(0): the sc1 stores of this wave have been acknowledged
    }
    __syncthreads();
    target += nblk;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int good = 1;
        unsigned spins = 0;
        while (__hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > kSpinLimit || __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                good = 0;
                break;
            }
        }
        ok = good;
    }
    __syncthreads();
    if (HEAVY)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    else
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return ok != 0;
}

// ---- the per-pixel arithmetic, shared by the two kernels below ----
// warp of I1, I1x, I1y by (u1,u2) with zero outside + grad, rho_c (tvl1flow_lib.c:144-163)
__device__ __forceinline__ void warp_px(const Scale& s, const IterBufs& b, int i, int j, int p, float u1, float u2, float& wx,
                                        float& wy, float& grad, float& rho_c) {
    Bicubic bc;
    bc.setup((float)((float)j + u1), (float)((float)i + u2), s.nx, s.ny);
    float w = 0.f;
    wx = wy = 0.f;
    if (!bc.out) {
        w = bc.eval(s.I1, s.nx);
        wx = bc.eval(b.I1x, s.nx);
        wy = bc.eval(b.I1y, s.nx);
    }
    const float Ix2 = wx * wx, Iy2 = wy * wy;
    grad = Ix2 + Iy2;
    rho_c = w - wx * u1 - wy * u2 - s.I0[p];
}
// where a pixel sits: the stencils of mask.c switch form in the first / last column and row
struct Edge {
    bool c0, cN, r0, rN;      // first column, last column, first row, last row
};
// thresholding step, divergence of p, update of u (tvl1flow_lib.c:171-215); returns the squared update.
// l11/l21: p11/p21 of the left neighbour, t12/t22: p12/p22 of the upper neighbour (unused at the borders)
__device__ __forceinline__ float u_px(Edge e, float& u1, float& u2, float wx, float wy, float g, float rho_c, float p11, float p12,
                                      float p21, float p22, float l11, float t12, float l21, float t22) {
    const float l_t = kLambda * kTheta;
    const float u1k = u1, u2k = u2;
    const float rho = rho_c + (wx * u1k + wy * u2k);
    // The reference's four-way branch (tvl1flow_lib.c:179-204) as selects in its order of precedence -- the same values
    // from the same operations, without the divergent branches (the quotient of the last case is formed everywhere and
    // dropped where another case applies: g = 0 gives an infinity or a NaN that the `g < kGradIsZero` select replaces).
    const float th = l_t * g, a1 = l_t * wx, a2 = l_t * wy, fi = -rho / g;
    float d1 = fi * wx, d2 = fi * wy;
    const bool flat = g < kGradIsZero, above = rho > th, below = rho < -th;
    d1 = flat ? 0.f : d1;
    d2 = flat ? 0.f : d2;
    d1 = above ? -a1 : d1;
    d2 = above ? -a2 : d2;
    d1 = below ? a1 : d1;
    d2 = below ? a2 : d2;
    const float v1 = u1k + d1, v2 = u2k + d2;
    // divergence (mask.c:40-90)
    const float ax1 = e.c0 ? p11 : (e.cN ? -l11 : p11 - l11);
    const float cy1 = e.r0 ? p12 : (e.rN ? -t12 : p12 - t12);
    const float ax2 = e.c0 ? p21 : (e.cN ? -l21 : p21 - l21);
    const float cy2 = e.r0 ? p22 : (e.rN ? -t22 : p22 - t22);
    u1 = v1 + kTheta * (ax1 + cy1);
    u2 = v2 + kTheta * (ax2 + cy2);
    return (u1 - u1k) * (u1 - u1k) + (u2 - u2k) * (u2 - u2k);
}
// sqrt(S) in double for S = a sum of two squares of floats: the compiler's expansion of the double-precision root (reciprocal
// square root, two coupled Newton steps, two residual corrections) WITHOUT its range scaling (for S < 2^-767; the smallest
// non-zero sum of squares of floats is 2^-298) and without its infinity test -- the same instructions on the same values,
// hence the same bits (tools/sqrt64_check.hip: 16.7 M pairs over every exponent, denormals and zeros included, none differ),
// a quarter fewer of them.
__device__ __forceinline__ double sqrt_sumsq(double S) {
    const double y = __builtin_amdgcn_rsq(S);
    const double g0 = S * y, h0 = y * 0.5;
    const double r0 = __builtin_fma(-h0, g0, 0.5);
    const double g1 = __builtin_fma(g0, r0, g0), h1 = __builtin_fma(h0, r0, h0);
    const double d0 = __builtin_fma(-g1, g1, S);
    const double g2 = __builtin_fma(d0, h1, g1);
    const double d1 = __builtin_fma(-g2, g2, S);
    const double r = __builtin_fma(d1, h1, g2);
    return S == 0.0 ? 0.0 : r;
}
// forward gradient of u at one pixel and the two denominators of the dual update (tvl1flow_lib.c:217-228).
// a, c: u1, u2 at the pixel; r*: at its right neighbour; d*: at its lower neighbour
struct DualStep {
    float u1x, u1y, u2x, u2y, ng1, ng2;
};
__device__ __forceinline__ DualStep dual_step(bool last_col, bool last_row, float a, float c, float r1, float d1, float r2, float d2) {
    const float taut = kTau / kTheta;
    DualStep o;
    o.u1x = last_col ? 0.f : r1 - a;
    o.u1y = last_row ? 0.f : d1 - a;
    o.u2x = last_col ? 0.f : r2 - c;
    o.u2y = last_row ? 0.f : d2 - c;
    // The reference calls hypot() in double and adds 1.0 in double (tvl1flow_lib.c:224-227).  The squares of
    // floats are exact in double, so sqrt(x*x + y*y) carries two roundings at 2^-53: after the conversion to
    // float it is the same number as any sub-ulp double hypot except on ~2^-28 of the inputs -- at a quarter
    // of the instructions of the library hypot (no scaling, no special cases: |x|,|y| < 2^60 here).  1 + t in
    // double then float equals the float sum for t >= 0 (a sum of two floats rounds once either way).
    const double x1 = (double)o.u1x, y1 = (double)o.u1y, x2 = (double)o.u2x, y2 = (double)o.u2y;
    const float g1 = (float)sqrt_sumsq(x1 * x1 + y1 * y1);
    const float g2 = (float)sqrt_sumsq(x2 * x2 + y2 * y2);
    o.ng1 = 1.0f + taut * g1;
    o.ng2 = 1.0f + taut * g2;
    return o;
}
// one component of the dual update (tvl1flow_lib.c:230-233)
__device__ __forceinline__ float dual_upd(float p, float grad, float ng) { return (p + (kTau / kTheta) * grad) / ng; }

// The convergence test (tvl1flow_lib.c:236-241) sums the squared update of every pixel.  The sum is formed in FIXED
// POINT: a pixel's term min(e, 128) * 2^s is rounded to an integer once, and integers add in any order to the same
// bits -- so the three kernels of this file (and any tiling, wave or block that does the adding) stop in the same
// iteration.  s = 47 - ceil(log2 npix): the whole image's sum stays below 2^55, a term's step is 2^-s (2^-29 at
// 640x360, against the test's threshold of 1e-4 per pixel).  A term above 128 (an update of more than 11 pixels
// in one iteration) only has to keep the sum above the threshold, which 128 does.
__device__ __forceinline__ int err_shift(int npix) { return 47 - (32 - __builtin_clz((unsigned)(npix > 1 ? npix - 1 : 1))); }
__device__ __forceinline__ unsigned long long err_fix(float e, int shift) {
    const float c = e < 128.f ? e : 128.f;
    return (unsigned long long)(__builtin_ldexp((double)c, shift) + 0.5);
}
__device__ __forceinline__ float err_value(unsigned long long sum, int shift, int npix) {
    return (float)(__builtin_ldexp((double)sum, -shift) / (double)npix);
}
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    return v;
}
// the same sum without LDS traffic (every lane active): two 32-bit sums on the vector ALU's lane-shift path.  v < 2^55 and
// the wave's sum too, so bits 24.. of the 64 values sum below 2^31 and bits 0..23 below 2^30.  Wave-uniform result.
__device__ __forceinline__ unsigned wave_sum_u32_dpp(unsigned v) {
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);      // row_shr:1 (zero shifted in): a scan ...
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);      // row_shr:2
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);      // row_shr:4
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);      // row_shr:8: lane 15 of a row = the row's sum
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true);      // row_bcast:15 into rows 1, 3
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true);      // row_bcast:31 into rows 2, 3
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned long long wave_sum_fix(unsigned long long v) {
    const unsigned lo = wave_sum_u32_dpp((unsigned)v & 0xffffffu), hi = wave_sum_u32_dpp((unsigned)(v >> 24));
    return ((unsigned long long)hi << 24) + lo;
}
__device__ __forceinline__ unsigned long long ldq(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void stq(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float error_of(const unsigned long long* partial, int ntiles, int npix, unsigned long long* shq) {
    const int tid = threadIdx.x;
    unsigned long long acc = 0;
    for (int k = tid; k < ntiles; k += 256) acc += ldq(partial + k);
    acc = wave_sum_u64(acc);
    if ((tid & 63) == 0) shq[tid >> 6] = acc;
    __syncthreads();
    return err_value((shq[0] + shq[1]) + (shq[2] + shq[3]), err_shift(npix), npix);
}

// what crosses workgroups inside a scale: u, double-buffered by iteration parity, and the per-tile error sums
struct Xch {
    float* u;                      // scale_kernel: [2 parities][2 components][npix]; scale_kernel_patch: [2 parities][npix] records
    unsigned long long* partial;   // [2 parities][ntiles] fixed-point sums (scale_kernel, scale_kernel_mem)
    unsigned long long* acc;       // [2 parities][kSumRecs] 16-byte records of the blocks' convergence sums (scale_kernel_patch)
};
// Several image pairs (the dataset's offline flow computation) share one launch: lane = blockIdx.x / gp.  The lanes
// never talk to each other -- each has its own buffers, barrier counter and iteration count -- they only fill the
// CUs that a single pair leaves idle between its memory round trips (a second cooperative kernel on another stream
// does not: cooperative launches are serialised).
struct Lane {
    Scale s;
    IterBufs b;
    Xch x;
    unsigned* bar;      // grid barrier counter of the lane
    int* ctl;           // ctl[2] += iterations run
};
struct Lanes {
    Lane l[kMaxLanes];
    int gp;             // blocks per lane
    int cus;            // compute units of the device: blocks numbered from here on are the second block of their CU (speed heuristic only)
    int* abort_word;    // shared: a barrier that gave up in any lane ends the launch
};
__device__ __forceinline__ float bld(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 16 /* sc1 */));
}
__device__ __forceinline__ void bst(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 16 /* sc1 */);
}

// Dual_TVL1_optic_flow (tvl1flow_lib.c:91-278) for one scale.  Work items are "tiles" of 256 consecutive pixels
// (row-major pixel index); block g owns tiles g, g + G, ... -- at most T of them, one pixel of each per thread.
// Everything a thread needs about ITS pixels lives in registers for the whole scale (u, the four dual fields,
// the warped gradient, rho_c), and so do the two dual values of the left neighbour and the two of the upper
// neighbour that its divergence reads: the thread recomputes those neighbours' dual update itself (same inputs,
// same operations -> the same bits as the neighbour's own copy).  So p never travels; memory only carries u,
// written once per iteration (write-through, `sc1`) into the buffer of the iteration's parity and read by the
// neighbours after ONE grid barrier per iteration:
//     [u update k | store u_k, tile error sums]  barrier  [error_k from the sums; load the 6 neighbours' u_k;
//      dual update k of own/left/upper pixel | stop? | u update k+1 | store into the other buffer]  barrier ...
// ctl[2] accumulates the iterations run (statistics), ctl[3] is the abort word of grid_sync.
template <int T>
__global__ __launch_bounds__(256) void scale_kernel(Lanes lanes) {
    __shared__ unsigned long long shf[T][4];
    __shared__ unsigned long long shd[4];
    const int lane_id = blockIdx.x / lanes.gp, gb = blockIdx.x - lane_id * lanes.gp;
    const unsigned gp = (unsigned)lanes.gp;
    const Scale s = lanes.l[lane_id].s;
    const IterBufs b = lanes.l[lane_id].b;
    const Xch x = lanes.l[lane_id].x;
    unsigned* const bar = lanes.l[lane_id].bar;
    int* const ctl = lanes.l[lane_id].ctl;
    int* const abort_word = lanes.abort_word;
    const int nx = s.nx, ny = s.ny, npix = nx * ny;
    const int ntiles = (npix + 255) >> 8;
    const int tid = threadIdx.x;
    const int eshift = err_shift(npix);
    const unsigned S = (unsigned)npix * 4u, row = (unsigned)nx * 4u;
    const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc((void*)x.u, 0, (int)(4u * S), 0x00020000);
    unsigned target = 0;
    int total = 0;
    unsigned po[T];                       // byte offset of the pixel inside one component array
    int pi[T], pj[T];
    bool act[T];
#pragma unroll
    for (int k = 0; k < T; ++k) {
        const int q = (gb + k * (int)gp) * 256 + tid;
        act[k] = q < npix;
        const int p = act[k] ? q : 0;
        pi[k] = p / nx;
        pj[k] = p - pi[k] * nx;
        po[k] = (unsigned)p * 4u;
    }
    // gradient of I1 (tvl1flow_lib.c:127): plain stores, made visible to the other blocks by the heavy barrier
#pragma unroll
    for (int k = 0; k < T; ++k)
        if (act[k]) centered_gradient_px(s.I1, b.I1x, b.I1y, nx, ny, pi[k], pj[k]);
    if (!grid_sync<true>(bar, target, gp, abort_word)) return;
    float u1[T], u2[T], p11[T], p12[T], p21[T], p22[T], wx[T], wy[T], grad[T], rho_c[T], l11[T], l21[T], t12[T], t22[T];
#pragma unroll
    for (int k = 0; k < T; ++k) {
        u1[k] = act[k] ? s.u1[po[k] >> 2] : 0.f;
        u2[k] = act[k] ? s.u2[po[k] >> 2] : 0.f;
        p11[k] = p12[k] = p21[k] = p22[k] = l11[k] = l21[k] = t12[k] = t22[k] = 0.f;      // tvl1flow_lib.c:131-139
        wx[k] = wy[k] = grad[k] = rho_c[k] = 0.f;
    }
    unsigned par = 0;                     // parity of the buffers the next u update writes
    for (int wp = 0; wp < kWarps; ++wp) {
#pragma unroll
        for (int k = 0; k < T; ++k) {
            if (act[k]) warp_px(s, b, pi[k], pj[k], (int)(po[k] >> 2), u1[k], u2[k], wx[k], wy[k], grad[k], rho_c[k]);
            __builtin_amdgcn_sched_barrier(0);   // one 3 x 16-tap stencil at a time
        }
        for (int n = 0;;) {
            const unsigned so1 = par * 2u * S, so2 = so1 + S;
            unsigned long long* part = x.partial + par * ntiles;
#pragma unroll
            for (int k = 0; k < T; ++k) {
                unsigned long long e = 0;
                if (act[k]) {
                    const Edge ed{pj[k] == 0, pj[k] == nx - 1, pi[k] == 0, pi[k] == ny - 1};
                    e = err_fix(u_px(ed, u1[k], u2[k], wx[k], wy[k], grad[k], rho_c[k], p11[k], p12[k], p21[k], p22[k], l11[k], t12[k],
                                     l21[k], t22[k]),
                                eshift);
                    bst(ur, po[k], so1, u1[k]);
                    bst(ur, po[k], so2, u2[k]);
                }
                e = wave_sum_u64(e);
                if ((tid & 63) == 0) shf[k][tid >> 6] = e;
            }
            __syncthreads();
            if (tid < T) {
                const int t = gb + tid * (int)gp;
                if (t < ntiles) stq(part + t, (shf[tid][0] + shf[tid][1]) + (shf[tid][2] + shf[tid][3]));
            }
            ++n;
            ++total;
            if (!grid_sync<false>(bar, target, gp, abort_word)) return;
            // u_n of the neighbours: right, below (own dual update); left, below-left (the left pixel's);
            // above, above-right (the upper pixel's).  Out-of-image offsets fall outside the buffer and read 0.
            float r1[T], r2[T], d1[T], d2[T], c1[T], c2[T], e1[T], e2[T], a1[T], a2[T], f1[T], f2[T];
#pragma unroll
            for (int k = 0; k < T; ++k) {
                r1[k] = bld(ur, po[k] + 4u, so1);
                r2[k] = bld(ur, po[k] + 4u, so2);
                d1[k] = bld(ur, po[k] + row, so1);
                d2[k] = bld(ur, po[k] + row, so2);
                c1[k] = bld(ur, po[k] - 4u, so1);
                c2[k] = bld(ur, po[k] - 4u, so2);
                e1[k] = bld(ur, po[k] + row - 4u, so1);
                e2[k] = bld(ur, po[k] + row - 4u, so2);
                a1[k] = bld(ur, po[k] - row, so1);
                a2[k] = bld(ur, po[k] - row, so2);
                f1[k] = bld(ur, po[k] - row + 4u, so1);
                f2[k] = bld(ur, po[k] - row + 4u, so2);
            }
            const float error = error_of(part, ntiles, npix, shd);
            // dual update n (tvl1flow_lib.c:217-234); the reference runs it in every pass that updated u, the last included
#pragma unroll
            for (int k = 0; k < T; ++k)
                if (act[k]) {
                    const bool cN = pj[k] == nx - 1, rN = pi[k] == ny - 1;
                    const DualStep o = dual_step(cN, rN, u1[k], u2[k], r1[k], d1[k], r2[k], d2[k]);
                    p11[k] = dual_upd(p11[k], o.u1x, o.ng1);
                    p12[k] = dual_upd(p12[k], o.u1y, o.ng1);
                    p21[k] = dual_upd(p21[k], o.u2x, o.ng2);
                    p22[k] = dual_upd(p22[k], o.u2y, o.ng2);
                    if (pj[k] > 0) {      // the left pixel (never in the last column; same row)
                        const DualStep l = dual_step(false, rN, c1[k], c2[k], u1[k], e1[k], u2[k], e2[k]);
                        l11[k] = dual_upd(l11[k], l.u1x, l.ng1);
                        l21[k] = dual_upd(l21[k], l.u2x, l.ng2);
                    }
                    if (pi[k] > 0) {      // the upper pixel (never in the last row; same column)
                        const DualStep t = dual_step(cN, false, a1[k], a2[k], f1[k], u1[k], f2[k], u2[k]);
                        t12[k] = dual_upd(t12[k], t.u1y, t.ng1);
                        t22[k] = dual_upd(t22[k], t.u2y, t.ng2);
                    }
                    __builtin_amdgcn_sched_barrier(0);   // one pixel's double-precision hypots at a time (registers)
                }
            par ^= 1u;
            if (!(error > kEps * kEps) || n >= kMaxIter) break;
        }
    }
#pragma unroll
    for (int k = 0; k < T; ++k)
        if (act[k]) {
            s.u1[po[k] >> 2] = u1[k];
            s.u2[po[k] >> 2] = u2[k];
        }
    if (gb == 0 && tid == 0) ctl[2] += total;
}

// ---- the register-state kernel without a grid barrier in its iteration ----
// scale_kernel's iteration is a chain of four memory-side round trips (stores acknowledged | arrival atomic | the
// counter seen complete | neighbour loads).  Here a block owns a 64 x PH PATCH of the image instead of interleaved
// row segments, so that
//   * neighbours inside the patch are read from an LDS copy of the patch's u (two parities);
//   * only the patch's perimeter travels: a perimeter pixel's u goes out as ONE 16-byte record {tag, u1, u2, tag}
//     (write-through), tag = launch epoch and publication number.  The owner of a ring cell polls the record until both
//     tags are this publication's -- the tags bracket the data, so whatever order the pieces of a record land in (first
//     to last or last to first, in one piece or several), two matching tags mean everything between them has landed: a
//     record needs no other ordering, no "stores acknowledged" wait, no flag;
//   * the convergence sum travels the same way: one tagged record per block, {tag, sum lo, sum hi, tag}, polled by
//     one wave of every block (fixed point: any order of adding, same bits).  It is the one thing of an iteration that
//     needs EVERY block, so it is read one iteration late: update n + 1 runs before the test of iteration n is known
//     and is dropped (u restored from a copy) when that test says n was the last -- the state the reference's loop
//     leaves, reached one discarded update later.  The ring only needs a block's neighbours.
//   * buffer reuse: ring records have two parities -- a block that publishes P + 1 has read its neighbours' P, so they
//     have all read its P - 1.  Sum records have four slots -- a block that publishes P has seen every block's P - 2,
//     so every block is at P - 2 or later and none still polls P - 4.
// Who does what between the two workgroup barriers of an iteration: waves 0-3 poll the ring cells, wave 4 polls the
// sums, waves 5-7 publish the perimeter (from the LDS copy) and the block's sum.  Publishers never wait for their
// stores and pollers have none in flight (on this chip a wave's loads return behind its earlier stores), and the
// barriers are `s_barrier` behind an LDS-only wait -- __syncthreads() would wait for the write-through stores too.
// One iteration = one-way store + one poll round trip + the arithmetic, which is scale_kernel's, operand for operand.
constexpr int kSumRecs = 256;                                // sum records per slot: one per block of a lane
constexpr int kSumSlots = 4;                                 // publications whose sums are kept (see the kernel's comment)
using u32x4 = __attribute__((vector_size(16))) unsigned;
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#ifdef RVDD_STAMPS
__device__ unsigned long long g_tvl1_stamps[512][8];
#define TS(v) const unsigned long long v = wall_clock64()
#else
#define TS(v)
#endif
// NT = 512 (patch 64 x 8T, one block per CU) or 256 (64 x 4T, two blocks per CU: one computes while the other polls).
template <int T, int NT>
__global__ __launch_bounds__(NT) void scale_kernel_patch(Lanes lanes, unsigned epoch) {
    constexpr int NW = NT / 64, PW = 64, PH = T * NW, MW = PW + 2, MH = PH + 2;
    constexpr int NRING = 2 * MW + 2 * PH;            // cells around the patch
    constexpr int NPERIM = 2 * PW + 2 * (PH - 2);     // cells on its edge
    constexpr int RW = NW / 2;                        // waves 0 .. RW-1 poll the ring, wave RW the sums, the rest publish
    constexpr int SUM0 = 64 * RW, PUB0 = SUM0 + 64;   // first thread of the sum pollers, of the publishers
    constexpr int RC = (NRING + SUM0 - 1) / SUM0;                       // ring cells per polling thread
    constexpr int PC = (NPERIM + (NT - PUB0) - 1) / (NT - PUB0);        // edge cells per publishing thread
    constexpr int NC = RC > PC ? RC : PC;
    __shared__ float2 map[2][MH][MW];            // u of the patch and of the ring around it, by iteration parity
    __shared__ unsigned long long red[NW];
    __shared__ unsigned long long sh_sum;
    __shared__ int sh_ok;
    // (lanes own consecutive block numbers: dealt out in turn, every lane gets its share of the blocks that came second to
    // their CU and lose its issue slots to the older block, and every lane runs at their pace -- 6.5 against 4.3 / 5.6 us
    // per iteration of two 640x360 pairs)
    const int lane_id = blockIdx.x / lanes.gp, gb = blockIdx.x - lane_id * lanes.gp;
    const unsigned gp = (unsigned)lanes.gp;
    const Scale s = lanes.l[lane_id].s;
    const IterBufs b = lanes.l[lane_id].b;
    const Xch x = lanes.l[lane_id].x;
    unsigned* const bar = lanes.l[lane_id].bar;
    int* const ctl = lanes.l[lane_id].ctl;
    int* const abort_word = lanes.abort_word;
    const int nx = s.nx, ny = s.ny, npix = nx * ny;
    const int tid = threadIdx.x, lx = tid & 63, wv = tid >> 6;
    const int npx = (nx + PW - 1) / PW;
    const int py = gb / npx, px = gb - py * npx;
    const int x0 = px * PW, y0 = py * PH;
    const int shift = err_shift(npix);
    const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc((void*)x.u, 0, (int)(32u * (unsigned)npix), 0x00020000);
    const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc((void*)x.acc, 0, kSumSlots * kSumRecs * 16, 0x00020000);
    const unsigned S = 16u * (unsigned)npix;      // one parity of records
    // this thread's pixels: column lx of the patch, rows wv * T .. wv * T + T - 1
    const int gx = x0 + lx, gy0 = y0 + wv * T;
    bool act[T];
#pragma unroll
    for (int k = 0; k < T; ++k) act[k] = gx < nx && gy0 + k < ny;
    const int pbase = gy0 * nx + gx;              // pixel index of slot 0 (slot k: + k * nx)
    // this thread's cells between the barriers: ring cells to fetch (waves < RW) or edge cells to publish (waves > RW)
    int cmy[NC], cmx[NC];                         // map coordinates (the patch's pixel (ly, lx) is map cell (ly + 1, lx + 1))
    bool cvalid[NC], ccell[NC];
    unsigned coff[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        cmy[i] = cmx[i] = 0;
        ccell[i] = false;
        if (tid < SUM0 && i < RC) {
            const int q = tid + i * SUM0;
            if (q < MW) { cmy[i] = 0; cmx[i] = q; }
            else if (q < 2 * MW) { cmy[i] = MH - 1; cmx[i] = q - MW; }
            else { const int r = q - 2 * MW; cmx[i] = r < PH ? 0 : MW - 1; cmy[i] = 1 + (r < PH ? r : r - PH); }
            ccell[i] = q < NRING;
        } else if (tid >= PUB0 && i < PC) {
            const int q = tid - PUB0 + i * (NT - PUB0);
            if (q < PW) { cmy[i] = 1; cmx[i] = 1 + q; }
            else if (q < 2 * PW) { cmy[i] = PH; cmx[i] = 1 + q - PW; }
            else { const int r = q - 2 * PW; cmx[i] = r < PH - 2 ? 1 : PW; cmy[i] = 2 + (r < PH - 2 ? r : r - (PH - 2)); }
            ccell[i] = q < NPERIM;
        }
        const int cgy = y0 + cmy[i] - 1, cgx = x0 + cmx[i] - 1;
        cvalid[i] = ccell[i] && cgx >= 0 && cgx < nx && cgy >= 0 && cgy < ny;      // a pixel of the image
        coff[i] = cvalid[i] ? 16u * (unsigned)(cgy * nx + cgx) : 0u;
    }
    if (tid == 0) sh_ok = 1;
#ifdef RVDD_STAMPS
    __shared__ unsigned long long sh_acc_t;
    if (tid == 0) sh_acc_t = 0;
    unsigned long long st_a = 0, st_w = 0, st_d = 0, st_ring = 0;
    const unsigned long long st_begin = wall_clock64();
#endif
    unsigned target = 0;
    int total = 0;
    // gradient of I1 (tvl1flow_lib.c:127): plain stores, made visible to the other blocks by the heavy barrier
#pragma unroll
    for (int k = 0; k < T; ++k)
        if (act[k]) centered_gradient_px(s.I1, b.I1x, b.I1y, nx, ny, gy0 + k, gx);
    if (!grid_sync<true>(bar, target, gp, abort_word)) return;
    // The dual fields of the left and of the upper pixel, which the divergence reads.  scale_kernel has every thread recompute
    // both neighbours' dual update; a thread of a patch finds most of them next to it:
    //   upper pixel: the thread's own slot k - 1; only slot 0 recomputes (t12, t22: the row above the wave's rows);
    //   left pixel: lane - 1's own p11 / p21 (a lane shift); only lane 0 has to recompute, and does so for its T rows in
    //   ONE pass with row k in lane k (lg11, lg21: lane k < T holds the left neighbour of lane 0's slot k).
    // Same operands, same operations, hence the same bits as the recomputation.
    float u1[T], u2[T], p11[T], p12[T], p21[T], p22[T], wx[T], wy[T], grad[T], rho_c[T];
    float t12 = 0.f, t22 = 0.f, lg11 = 0.f, lg21 = 0.f;
#pragma unroll
    for (int k = 0; k < T; ++k) {
        u1[k] = act[k] ? s.u1[pbase + k * nx] : 0.f;
        u2[k] = act[k] ? s.u2[pbase + k * nx] : 0.f;
        p11[k] = p12[k] = p21[k] = p22[k] = 0.f;      // tvl1flow_lib.c:131-139
        wx[k] = wy[k] = grad[k] = rho_c[k] = 0.f;
    }
    unsigned pub = 0;                            // publications of this launch (iterations, the discarded ones included)
    for (int wp = 0; wp < kWarps; ++wp) {
#pragma unroll
        for (int k = 0; k < T; ++k) {
            if (act[k]) warp_px(s, b, gy0 + k, gx, pbase + k * nx, u1[k], u2[k], wx[k], wy[k], grad[k], rho_c[k]);
            __builtin_amdgcn_sched_barrier(0);   // one 3 x 16-tap stencil at a time
        }
        __builtin_amdgcn_s_waitcnt(0);           // the warp's loads are the last this wave waits for until the scale's end
        for (int n = 0;;) {
            ++n;
            ++total;
            ++pub;
            TS(ts0);
            const unsigned tagv = epoch * 2048u + pub, par = pub & 1u;
            // Two blocks on a CU: the older one wins the issue slots whenever both compute, and the lane that holds the younger
            // blocks sets the launch's pace.  Every other iteration the younger block is given the higher priority instead.
            // (launches of two lanes, i.e. 640x360: with more lanes the pace is set by the lane with the most iterations, wherever it sits)
            if (blockIdx.x >= (unsigned)lanes.cus && par && gridDim.x == 2u * gp) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
            unsigned long long esum = 0;
            float u1s[T], u2s[T];                // u of iteration n - 1: this update is undone if that one turns out to have converged
#pragma unroll
            for (int k = 0; k < T; ++k) {
                u1s[k] = u1[k];
                u2s[k] = u2[k];
            }
#pragma unroll
            for (int k = 0; k < T; ++k) {
                // left neighbour's p11, p21: lane - 1's (wave_shr:1), lane 0 keeps `old` = its recomputed copy
                const float l11 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_amdgcn_readlane(__builtin_bit_cast(int, lg11), k),
                                                                                       __builtin_bit_cast(int, p11[k]), 0x138, 0xf, 0xf, false));
                const float l21 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_amdgcn_readlane(__builtin_bit_cast(int, lg21), k),
                                                                                       __builtin_bit_cast(int, p21[k]), 0x138, 0xf, 0xf, false));
                const float up12 = k == 0 ? t12 : p12[k > 0 ? k - 1 : 0], up22 = k == 0 ? t22 : p22[k > 0 ? k - 1 : 0];
                if (act[k]) {
                    const Edge ed{gx == 0, gx == nx - 1, gy0 + k == 0, gy0 + k == ny - 1};
                    esum += err_fix(u_px(ed, u1[k], u2[k], wx[k], wy[k], grad[k], rho_c[k], p11[k], p12[k], p21[k], p22[k], l11, up12, l21,
                                         up22),
                                    shift);
                }
                map[par][wv * T + k + 1][lx + 1] = float2{u1[k], u2[k]};
            }
            esum = wave_sum_fix(esum);
            if (lx == 0) red[wv] = esum;
            lds_barrier();
            TS(ts1);
            if (tid < SUM0) {                    // the ring: this thread's cells requested together, again until all are there
                float2 got[RC];
                bool need[RC];
#pragma unroll
                for (int i = 0; i < RC; ++i) {
                    got[i] = float2{0.f, 0.f};
                    need[i] = cvalid[i];
                }
                for (unsigned spins = 0;;) {
                    u32x4 v[RC];
#pragma unroll
                    for (int i = 0; i < RC; ++i)
                        if (need[i]) v[i] = __builtin_amdgcn_raw_buffer_load_b128(ur, coff[i], par * S, 16 /* sc1 */);
                    bool more = false;
#pragma unroll
                    for (int i = 0; i < RC; ++i)
                        if (need[i]) {
                            if (v[i][0] == tagv && v[i][3] == tagv) {
                                got[i] = float2{__uint_as_float(v[i][1]), __uint_as_float(v[i][2])};
                                need[i] = false;
                            } else {
                                more = true;
                            }
                        }
                    if (!more) break;
                    if (++spins > kSpinLimit || ((spins & 255u) == 0 && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                        __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        sh_ok = 0;
                        break;
                    }
                }
#pragma unroll
                for (int i = 0; i < RC; ++i)
                    if (ccell[i]) map[par][cmy[i]][cmx[i]] = got[i];
#ifdef RVDD_STAMPS
                if (tid == 0) st_ring += wall_clock64() - ts1;
#endif
            } else if (tid < PUB0) {             // every block's sum; the own one from LDS
                // lane l: the records of blocks l, l + 64, l + 128, l + 192, requested together; the own block's sum from LDS
                unsigned long long t = 0;
                // (the sums of the PREVIOUS publication: they have had an iteration's time to arrive)
                const unsigned ptag = tagv - 1u, pslot = ((pub - 1u) & (kSumSlots - 1u)) * (kSumRecs * 16u);
                bool need[kSumRecs / 64];
#pragma unroll
                for (int i = 0; i < kSumRecs / 64; ++i) {
                    const unsigned j = (unsigned)lx + 64u * i;
                    need[i] = j < gp && pub > 1u;
                }
                for (unsigned spins = 0;;) {
                    u32x4 v[kSumRecs / 64];
#pragma unroll
                    for (int i = 0; i < kSumRecs / 64; ++i)
                        if (need[i]) v[i] = __builtin_amdgcn_raw_buffer_load_b128(sr, 16u * ((unsigned)lx + 64u * i), pslot, 16);
                    bool more = false;
#pragma unroll
                    for (int i = 0; i < kSumRecs / 64; ++i)
                        if (need[i]) {
                            if (v[i][0] == ptag && v[i][3] == ptag) {
                                t += ((unsigned long long)v[i][2] << 32) | v[i][1];
                                need[i] = false;
                            } else {
                                more = true;
                            }
                        }
                    if (!more) break;
                    if (++spins > kSpinLimit || ((spins & 255u) == 0 && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                        __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        sh_ok = 0;
                        break;
                    }
                }
                t = wave_sum_fix(t);
                if (lx == 0) sh_sum = t;
#ifdef RVDD_STAMPS
                if (lx == 0) sh_acc_t += wall_clock64() - ts1;
#endif
            } else {                             // the patch's edge, and the block's sum
#pragma unroll
                for (int i = 0; i < PC; ++i)
                    if (cvalid[i]) {
                        const float2 v = map[par][cmy[i]][cmx[i]];
                        const u32x4 rec = {tagv, __float_as_uint(v.x), __float_as_uint(v.y), tagv};
                        __builtin_amdgcn_raw_buffer_store_b128(rec, ur, coff[i], par * S, 16 /* sc1 */);
                    }
                if (tid == NT - 1) {
                    unsigned long long t = 0;
#pragma unroll
                    for (int i = 0; i < NW; ++i) t += red[i];
                    const u32x4 rec = {tagv, (unsigned)t, (unsigned)(t >> 32), tagv};
                    __builtin_amdgcn_raw_buffer_store_b128(rec, sr, 16u * (unsigned)gb, (pub & (kSumSlots - 1u)) * (kSumRecs * 16u), 16 /* sc1 */);
                }
            }
            lds_barrier();
            TS(ts2);
            if (!sh_ok) return;
            // the convergence test of iteration n - 1 (tvl1flow_lib.c:236-241), one iteration late: its sum needs every block and
            // is the slowest thing to arrive, so the update above ran ahead of it; if n - 1 was the last, that update is dropped
            if (n > 1 && !(err_value(sh_sum, shift, npix) > kEps * kEps)) {
#pragma unroll
                for (int k = 0; k < T; ++k) {
                    u1[k] = u1s[k];
                    u2[k] = u2s[k];
                }
                --total;
                break;
            }
            // dual update n (tvl1flow_lib.c:217-234) of the own, the left and the upper pixel, as in scale_kernel
#pragma unroll
            for (int k = 0; k < T; ++k)
                if (act[k]) {
                    const int my = wv * T + k + 1, mx = lx + 1;
                    const float2 r = map[par][my][mx + 1], d = map[par][my + 1][mx];
                    const float2 a = map[par][my - 1][mx], f = map[par][my - 1][mx + 1];      // used by slot 0 only
                    const bool cN = gx == nx - 1, rN = gy0 + k == ny - 1;
                    const DualStep o = dual_step(cN, rN, u1[k], u2[k], r.x, d.x, r.y, d.y);
                    p11[k] = dual_upd(p11[k], o.u1x, o.ng1);
                    p12[k] = dual_upd(p12[k], o.u1y, o.ng1);
                    p21[k] = dual_upd(p21[k], o.u2x, o.ng2);
                    p22[k] = dual_upd(p22[k], o.u2y, o.ng2);
                    if (k == 0 && gy0 > 0) {      // the pixel above the wave's first row
                        const DualStep t = dual_step(cN, false, a.x, a.y, f.x, u1[k], f.y, u2[k]);
                        t12 = dual_upd(t12, t.u1y, t.ng1);
                        t22 = dual_upd(t22, t.u2y, t.ng2);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            if (lx < T && x0 > 0 && gy0 + lx < ny) {      // lane k: the pixel left of the patch in row k of this wave
                const int my = wv * T + lx + 1;
                const float2 c = map[par][my][0], e = map[par][my + 1][0], own = map[par][my][1];
                const DualStep l = dual_step(false, gy0 + lx == ny - 1, c.x, c.y, own.x, e.x, own.y, e.y);
                lg11 = dual_upd(lg11, l.u1x, l.ng1);
                lg21 = dual_upd(lg21, l.u2x, l.ng2);
            }
#ifdef RVDD_STAMPS
            { const unsigned long long ts3 = wall_clock64(); st_a += ts1 - ts0; st_w += ts2 - ts1; st_d += ts3 - ts2; }
#endif
            if (n >= kMaxIter) break;      // the reference's loop ends here whatever the error
        }
    }
#pragma unroll
    for (int k = 0; k < T; ++k)
        if (act[k]) {
            s.u1[pbase + k * nx] = u1[k];
            s.u2[pbase + k * nx] = u2[k];
        }
    if (gb == 0 && tid == 0) ctl[2] += total;
#ifdef RVDD_STAMPS
    if (tid == 0 && blockIdx.x < 512) {
        unsigned long long* o = g_tvl1_stamps[blockIdx.x];
        o[0] = (unsigned long long)total; o[1] = st_a; o[2] = st_w; o[3] = st_d; o[4] = st_ring; o[5] = sh_acc_t;
        o[6] = wall_clock64() - st_begin; o[7] = (unsigned long long)T;
    }
#endif
}

// The same scale for images whose pixels do not fit the register slots of one resident grid
// (> kMaxSlots x 256 x #CUs pixels): the same arithmetic on the same values, but a block walks its tiles one
// after the other and a pixel's state lives in memory (u in place, p in b.p*, the warped gradient and rho_c in
// `st`), two grid barriers per iteration -- one memory round trip per TILE and phase: several times slower.
struct StateBufs {
    float *wx, *wy, *grad, *rho_c;
};
__global__ __launch_bounds__(256) void scale_kernel_mem(Lanes lanes, StateBufs st) {
    __shared__ unsigned long long shf[4];
    __shared__ unsigned long long shd[4];
    const int gb = blockIdx.x;                        // one lane only: this form is the fallback for very large images
    const unsigned gp = (unsigned)lanes.gp;
    const Scale s = lanes.l[0].s;
    const IterBufs b = lanes.l[0].b;
    const Xch x = lanes.l[0].x;
    unsigned* const bar = lanes.l[0].bar;
    int* const ctl = lanes.l[0].ctl;
    int* const abort_word = lanes.abort_word;
    const int nx = s.nx, ny = s.ny, npix = nx * ny;
    const int ntiles = (npix + 255) >> 8;
    const int tid = threadIdx.x;
    unsigned long long* partial = x.partial;
    const int eshift = err_shift(npix);
    unsigned target = 0;
    int total = 0;
#define FOR_TILES(...)                                           \
    for (int t = gb; t < ntiles; t += (int)gp) {               \
        const int p = t * 256 + tid;                             \
        const bool on = p < npix;                                \
        const int i = on ? p / nx : 0, j = on ? p - i * nx : 0;  \
        __VA_ARGS__                                              \
    }
    FOR_TILES(if (on) {
        centered_gradient_px(s.I1, b.I1x, b.I1y, nx, ny, i, j);
        b.p11[p] = b.p12[p] = b.p21[p] = b.p22[p] = 0.f;
    })
    if (!grid_sync<true>(bar, target, gp, abort_word)) return;
    for (int wp = 0; wp < kWarps; ++wp) {
        FOR_TILES(if (on) {
            float wx, wy, g, rc;
            warp_px(s, b, i, j, p, s.u1[p], s.u2[p], wx, wy, g, rc);
            st.wx[p] = wx; st.wy[p] = wy; st.grad[p] = g; st.rho_c[p] = rc;
        })
        for (int n = 0;;) {
            FOR_TILES(
                unsigned long long e = 0;
                if (on) {
                    float u1 = s.u1[p], u2 = s.u2[p];
                    const float l11 = j > 0 ? ldc(b.p11 + p - 1) : 0.f, l21 = j > 0 ? ldc(b.p21 + p - 1) : 0.f;
                    const float t12 = i > 0 ? ldc(b.p12 + p - nx) : 0.f, t22 = i > 0 ? ldc(b.p22 + p - nx) : 0.f;
                    const Edge ed{j == 0, j == nx - 1, i == 0, i == ny - 1};
                    e = err_fix(u_px(ed, u1, u2, st.wx[p], st.wy[p], st.grad[p], st.rho_c[p], b.p11[p], b.p12[p], b.p21[p], b.p22[p], l11, t12,
                                     l21, t22),
                                eshift);
                    stc(s.u1 + p, u1);
                    stc(s.u2 + p, u2);
                }
                e = wave_sum_u64(e);
                if ((tid & 63) == 0) shf[tid >> 6] = e;
                __syncthreads();
                if (tid == 0) stq(partial + t, (shf[0] + shf[1]) + (shf[2] + shf[3]));
                __syncthreads();
            )
            if (!grid_sync<false>(bar, target, gp, abort_word)) return;
            const float error = error_of(partial, ntiles, npix, shd);
            ++n;
            ++total;
            FOR_TILES(if (on) {
                const float r1 = j < nx - 1 ? ldc(s.u1 + p + 1) : 0.f, r2 = j < nx - 1 ? ldc(s.u2 + p + 1) : 0.f;
                const float d1 = i < ny - 1 ? ldc(s.u1 + p + nx) : 0.f, d2 = i < ny - 1 ? ldc(s.u2 + p + nx) : 0.f;
                const DualStep o = dual_step(j == nx - 1, i == ny - 1, s.u1[p], s.u2[p], r1, d1, r2, d2);
                stc(b.p11 + p, dual_upd(b.p11[p], o.u1x, o.ng1));
                stc(b.p12 + p, dual_upd(b.p12[p], o.u1y, o.ng1));
                stc(b.p21 + p, dual_upd(b.p21[p], o.u2x, o.ng2));
                stc(b.p22 + p, dual_upd(b.p22[p], o.u2y, o.ng2));
            })
            if (!grid_sync<false>(bar, target, gp, abort_word)) return;
            if (!(error > kEps * kEps) || n >= kMaxIter) break;
        }
    }
#undef FOR_TILES
    if (gb == 0 && tid == 0) ctl[2] += total;
}

constexpr int kTileSlots[] = {1, 2, 3, 4, 5, 6, 8};
constexpr int kMaxSlots = 8;
using ScaleKernel = void (*)(Lanes);
using PatchKernel = void (*)(Lanes, unsigned);
// (slots, threads): patches of 64 x 4 ... 64 x 32 pixels
PatchKernel patch_kernel_for(int slots, int threads) {
    if (threads == 256) {
        switch (slots) {
            case 1: return scale_kernel_patch<1, 256>;
            case 2: return scale_kernel_patch<2, 256>;
            case 4: return scale_kernel_patch<4, 256>;
        }
    } else {
        switch (slots) {
            case 1: return scale_kernel_patch<1, 512>;
            case 2: return scale_kernel_patch<2, 512>;
            case 4: return scale_kernel_patch<4, 512>;
        }
    }
    return nullptr;
}
ScaleKernel scale_kernel_for(int slots) {
    switch (slots) {
        case 1: return scale_kernel<1>;
        case 2: return scale_kernel<2>;
        case 3: return scale_kernel<3>;
        case 4: return scale_kernel<4>;
        case 5: return scale_kernel<5>;
        case 6: return scale_kernel<6>;
        case 8: return scale_kernel<8>;
    }
    return nullptr;
}

GaussK make_gauss(double sigma) {
    GaussK k{};
    const double den = 2 * sigma * sigma;
    k.size = (int)(5 * sigma) + 1;
    for (int i = 0; i < k.size; ++i) k.B[i] = 1 / (sigma * std::sqrt(2.0 * 3.1415926)) * std::exp(-i * i / den);
    double norm = 0;
    for (int i = 0; i < k.size; ++i) norm += k.B[i];
    norm *= 2;
    norm -= k.B[0];
    for (int i = 0; i < k.size; ++i) k.B[i] /= norm;
    return k;
}

dim3 grid2(int nx, int ny) { return dim3((nx + 255) / 256, ny); }

}  // namespace

// Workspace: every buffer the pyramid needs for one (nx, ny), per lane; owned by the caller (runtime.hip).
struct Tvl1LaneBufs {
    std::vector<Scale> sc;
    IterBufs it{};
    float *tmp = nullptr, *tmp2 = nullptr, *mm = nullptr;
    Xch xch{};
    int* ctl = nullptr;         // {unused, unused, iterations, unused}
    unsigned* bar = nullptr;
};
struct Tvl1Workspace {
    int nx = 0, ny = 0, nscales = 0;
    std::vector<Tvl1LaneBufs> lanes;     // lane 0 at allocation, lane 1 with the first batch call
    int* abort_word = nullptr;
    bool unchecked = false;              // launches of an asynchronous batch are in flight / done whose abort word nobody has read yet
    int cus = 0;
    unsigned epoch = 0;         // scale_kernel_patch: one per launch, the upper bits of its records' tags
    StateBufs state{};          // scale_kernel_mem only: allocated when the finest scale exceeds the register slots
    bool patch = true;          // RVDD_TVL1_PATCH=0: scale_kernel (grid barrier per iteration) instead of scale_kernel_patch
    bool force_mem = false;     // RVDD_TVL1_MEM=1: take the memory-state kernel at every scale (equivalence tests)
    std::vector<void*> allocs;
};

void tvl1_free(Tvl1Workspace* w) {
    if (!w) return;
    for (void* p : w->allocs) (void)hipFree(p);
    delete w;
}

int tvl1_ws_nx(const Tvl1Workspace* w) { return w->nx; }
int tvl1_ws_ny(const Tvl1Workspace* w) { return w->ny; }

int tvl1_num_scales(int nx, int ny) {
    const float N = 1 + std::log(std::hypot((double)nx, (double)ny) / 16.0) / std::log(1 / kZoom);
    int ns = kMaxScales;
    if (N < ns) ns = (int)N;
    return ns < 1 ? 1 : ns;
}

// The zoom-out Gaussian (sigma 0.6*sqrt(3): 6 taps a side) is applied to every scale but the coarsest.  The
// reference sizes its pyramid from the image DIAGONAL (tvl1flow_lib.c / libBridge.cpp:131-136), so a very skinny image
// gets scales whose short side is below the filter's reach and the reference then reads outside its buffers
// (mask.c:262-325 has no guard).  Those sizes are refused here instead of reproducing undefined behaviour.
bool tvl1_size_ok(int nx, int ny) {
    const int ns = tvl1_num_scales(nx, ny);
    int sx = nx, sy = ny;
    for (int s = 0; s + 1 < ns; ++s) {
        if ((sx < sy ? sx : sy) < 6) return false;
        sx = (int)((float)sx * kZoom + 0.5f);
        sy = (int)((float)sy * kZoom + 0.5f);
    }
    return nx >= 16 && ny >= 16;
}

static hipError_t tvl1_add_lane(Tvl1Workspace* w) {
    hipError_t err = hipSuccess;
    auto A = [&](float** p, size_t n) {
        if (err == hipSuccess) {
            err = hipMalloc(reinterpret_cast<void**>(p), n * sizeof(float));
            if (err == hipSuccess) w->allocs.push_back(*p);
        }
    };
    Tvl1LaneBufs L;
    int sx = w->nx, sy = w->ny;
    for (int s = 0; s < w->nscales; ++s) {
        Scale sc{};
        sc.nx = sx;
        sc.ny = sy;
        const size_t n = (size_t)sx * sy;
        A(&sc.I0, n);
        A(&sc.I1, n);
        A(&sc.u1, n);
        A(&sc.u2, n);
        L.sc.push_back(sc);
        sx = (int)((float)sx * kZoom + 0.5f);       // zoom_size (zoom.c:22-34)
        sy = (int)((float)sy * kZoom + 0.5f);
    }
    const size_t n0 = (size_t)w->nx * w->ny;
    float** its[] = {&L.it.I1x, &L.it.I1y, &L.it.I1w, &L.it.p11, &L.it.p12, &L.it.p21, &L.it.p22, &L.tmp, &L.tmp2};
    for (float** p : its) A(p, n0);
    float* words64 = nullptr;
    A(&words64, 4 * ((n0 + 255) / 256));                      // 2 parities of 64-bit tile sums
    L.xch.partial = reinterpret_cast<unsigned long long*>(words64);
    A(&L.xch.u, 8 * n0);                                      // 2 parities of 16-byte records (scale_kernel: 4 planes of floats)
    if (err == hipSuccess) err = hipMemset(L.xch.u, 0, 8 * n0 * sizeof(float));      // tag 0: no launch's
    words64 = nullptr;
    A(&words64, kSumSlots * kSumRecs * 4);                    // kSumSlots publications of 16-byte sum records
    L.xch.acc = reinterpret_cast<unsigned long long*>(words64);
    if (err == hipSuccess) err = hipMemset(words64, 0, kSumSlots * kSumRecs * 16);
    A(&L.mm, 4);
    float* words = nullptr;
    A(&words, kCtlWords);                            // 4 control ints + one grid barrier counter per scale
    L.ctl = reinterpret_cast<int*>(words);
    L.bar = reinterpret_cast<unsigned*>(words) + 4;
    // the fills above ran on the null stream; the caller's stream may not be ordered behind it, and a fill that lands after
    // the first records were published would erase them (readers would poll for ever): once per lane and image size
    if (err == hipSuccess) err = hipDeviceSynchronize();
    if (err == hipSuccess) w->lanes.push_back(L);
    return err;
}

hipError_t tvl1_alloc(Tvl1Workspace** out, int nx, int ny) {
    Tvl1Workspace* w = new Tvl1Workspace();
    w->nx = nx;
    w->ny = ny;
    w->nscales = tvl1_num_scales(nx, ny);
    hipError_t err = tvl1_add_lane(w);
    if (err == hipSuccess) {
        err = hipMalloc(reinterpret_cast<void**>(&w->abort_word), sizeof(int));
        if (err == hipSuccess) w->allocs.push_back(w->abort_word);
    }
    if (err == hipSuccess) {
        int dev = 0;
        err = hipGetDevice(&dev);
        if (err == hipSuccess) err = hipDeviceGetAttribute(&w->cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (err == hipSuccess && w->cus < 1) err = hipErrorLaunchFailure;
    }
    const size_t n0 = (size_t)nx * ny;
    const char* fm = std::getenv("RVDD_TVL1_MEM");
    w->force_mem = fm && fm[0] == '1';
    const char* pk = std::getenv("RVDD_TVL1_PATCH");
    w->patch = !pk || std::atoi(pk) != 0;
    if (err == hipSuccess && (w->force_mem || (n0 + 255) / 256 > (size_t)w->cus * kMaxSlots))
        for (float** p : {&w->state.wx, &w->state.wy, &w->state.grad, &w->state.rho_c})
            if (err == hipSuccess) {
                err = hipMalloc(reinterpret_cast<void**>(p), n0 * sizeof(float));
                if (err == hipSuccess) w->allocs.push_back(*p);
            }
    if (err != hipSuccess) {
        tvl1_free(w);
        return err;
    }
    *out = w;
    return hipSuccess;
}

// Dual_TVL1_optic_flow_multiscale (tvl1flow_lib.c:343-472) for `np` (1..kMaxLanes) pairs of the same size in one set
// of launches.  I0, I1: [np][ny][nx]; u: [np][2][ny][nx] = per pair u(ny*nx) then v(ny*nx) as libBridge.cpp:150.
// `first` / `finish`: a batch runs several sets back to back on the stream; only the first clears the abort word and only the
// last (or every one, when the iteration counts are wanted) reads the control words back and synchronises.
static hipError_t tvl1_run_lanes(Tvl1Workspace* w, const float* I0, const float* I1, float* u, int np, hipStream_t st, int* iters,
                                 bool first = true, bool finish = true) {
#define CK(e)                                 \
    do {                                      \
        hipError_t e__ = (e);                 \
        if (e__ != hipSuccess) return e__;    \
    } while (0)
    const int nx = w->nx, ny = w->ny, n0 = nx * ny;
    while ((int)w->lanes.size() < np) CK(tvl1_add_lane(w));
    if (first && !w->unchecked) CK(hipMemsetAsync(w->abort_word, 0, sizeof(int), st));      // (an unread abort stays set until somebody reads it)
    const double zsigma = (double)(float)(kZoomSigma0 * std::sqrt(1.0 / ((double)kZoom * (double)kZoom) - 1.0));
    // Pre-processing of all pairs together (blockIdx.z / .y = image / pair): normalisation to [0,255], pre-smoothing, pyramid.
    // The finest flow is the caller's buffer (fin): u(ny*nx) then v(ny*nx) per pair.
    auto fin = [&](int q, int c) { return u + ((size_t)q * 2 + c) * n0; };
    auto u_of = [&](int q, int s, int c) { return s == 0 ? fin(q, c) : (c ? w->lanes[q].sc[s].u2 : w->lanes[q].sc[s].u1); };
    {
        PairWords pw{};
        Imgs raw{};                // the caller's images -> tmp / tmp2 (normalised)
        const Scale& top = w->lanes[0].sc[w->nscales - 1];
        for (int q = 0; q < np; ++q) {
            Tvl1LaneBufs& L = w->lanes[q];
            pw.mm[q] = L.mm;
            pw.ctl[q] = L.ctl;
            pw.top[2 * q] = u_of(q, w->nscales - 1, 0);
            pw.top[2 * q + 1] = u_of(q, w->nscales - 1, 1);
            raw.in[2 * q] = I0 + (size_t)q * n0;
            raw.in[2 * q + 1] = I1 + (size_t)q * n0;
            raw.out[2 * q] = L.tmp;
            raw.out[2 * q + 1] = L.tmp2;
        }
        const int ntop = top.nx * top.ny;
        hipLaunchKernelGGL(init_kernel, dim3((ntop + 255) / 256 < 64 ? (ntop + 255) / 256 : 64, np), dim3(256), 0, st, pw, kCtlWords, ntop);
        hipLaunchKernelGGL(minmax_kernel, dim3(256, np), dim3(256), 0, st, raw, n0, pw);
        hipLaunchKernelGGL(normalize_kernel, dim3((n0 + 255) / 256, np), dim3(256), 0, st, raw, n0, pw);
        // separable Gaussian of every image of the batch: in -> scratch (I1w / I1x, free until the scale kernels) -> out
        auto grid3 = [&](int gx, int gy) { return dim3((gx + 255) / 256, gy, 2 * np); };
        auto gauss2 = [&](auto in_of, auto out_of, int gx, int gy, double sigma) {
            const GaussK k = make_gauss(sigma);
            Imgs h{}, v{};
            for (int q = 0; q < np; ++q)
                for (int c = 0; c < 2; ++c) {
                    Tvl1LaneBufs& L = w->lanes[q];
                    float* scratch = c ? L.it.I1x : L.it.I1w;
                    h.in[2 * q + c] = in_of(L, c);
                    h.out[2 * q + c] = scratch;
                    v.in[2 * q + c] = scratch;
                    v.out[2 * q + c] = out_of(L, c);
                }
            hipLaunchKernelGGL(gauss_kernel, grid3(gx, gy), dim3(256), 0, st, h, gx, gy, k, 0);
            hipLaunchKernelGGL(gauss_kernel, grid3(gx, gy), dim3(256), 0, st, v, gx, gy, k, 1);
        };
        auto tmp_of = [](Tvl1LaneBufs& L, int c) { return c ? L.tmp2 : L.tmp; };
        gauss2(tmp_of, [](Tvl1LaneBufs& L, int c) { return c ? L.sc[0].I1 : L.sc[0].I0; }, nx, ny, kPresmooth);
        // pyramid (zoom_out, zoom.c:41-78)
        for (int s = 1; s < w->nscales; ++s) {
            const Scale& a = w->lanes[0].sc[s - 1];
            const Scale& b = w->lanes[0].sc[s];
            gauss2([s](Tvl1LaneBufs& L, int c) { return c ? L.sc[s - 1].I1 : L.sc[s - 1].I0; }, tmp_of, a.nx, a.ny, zsigma);
            Imgs z{};
            for (int q = 0; q < np; ++q)
                for (int c = 0; c < 2; ++c) {
                    Tvl1LaneBufs& L = w->lanes[q];
                    z.in[2 * q + c] = tmp_of(L, c);
                    z.out[2 * q + c] = c ? L.sc[s].I1 : L.sc[s].I0;
                }
            hipLaunchKernelGGL(resample_kernel, grid3(b.nx, b.ny), dim3(256), 0, st, z, a.nx, a.ny, b.nx, b.ny, kZoom, kZoom, 1.f);
        }
    }
    // Patch kernel forms: patches of 64 x rows pixels, one 512-thread block per CU (8, 16, 32 rows) or two 256-thread blocks
    // per CU (16 rows: as many pixels per CU as 32 rows in one block, but one block computes while the other waits for its ring).
    struct Form { int t, n, rows, per_cu; };
    static const Form forms[] = {{1, 256, 4, 2}, {1, 512, 8, 1}, {2, 256, 8, 2}, {2, 512, 16, 1}, {4, 256, 16, 2}, {4, 512, 32, 1}};
    static const int two = [] { const char* e = std::getenv("RVDD_TVL1_TWO"); return e ? std::atoi(e) : 1; }();
    // the first form (smallest patches) whose blocks are all resident with `cnt` pairs in the launch; null if none
    auto patch_form = [&](const Scale& ref, int cnt, int* blocks) -> const Form* {
        if (!w->patch || w->force_mem) return nullptr;
        for (size_t i = 0; i < sizeof forms / sizeof forms[0]; ++i) {
            const Form& f = forms[i];
            if (f.per_cu == 2 && !two) continue;
            // what the runtime will accept as co-resident (asked once per form): a form planned for two blocks per CU is only
            // taken where two fit
            static int occ[sizeof forms / sizeof forms[0]] = {};
            if (!occ[i]) {
                int per_cu = 0;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, patch_kernel_for(f.t, f.n), f.n, 0) != hipSuccess) per_cu = 0;
                occ[i] = per_cu > 0 ? per_cu : -1;
            }
            if (occ[i] < f.per_cu) continue;
            const int g = ((ref.nx + 63) / 64) * ((ref.ny + f.rows - 1) / f.rows);
            if ((long)cnt * g <= (long)w->cus * f.per_cu && g <= kSumRecs) { *blocks = g; return &f; }
        }
        return nullptr;
    };
    // one scale (gradient, 5 warps x <= 300 iterations) of the pairs q0 .. q0 + cnt - 1 in one cooperative launch
    auto launch_scale = [&](int s, int q0, int cnt) -> hipError_t {
        Lanes lanes{};
        lanes.abort_word = w->abort_word;
        lanes.cus = w->cus;
        for (int q = 0; q < cnt; ++q) {
            Tvl1LaneBufs& L = w->lanes[q0 + q];
            Scale sc = L.sc[s];
            sc.u1 = u_of(q0 + q, s, 0);
            sc.u2 = u_of(q0 + q, s, 1);
            lanes.l[q] = Lane{sc, L.it, L.xch, L.bar + s, L.ctl};      // a barrier counter per scale, cleared by init_kernel
        }
        const Scale& ref = w->lanes[0].sc[s];
        const int ntiles = (ref.nx * ref.ny + 255) / 256;
        int pg = 0;
        if (const Form* f = patch_form(ref, cnt, &pg)) {
            if (++w->epoch >= (1u << 21)) {       // tags repeat after 2^21 launches: start over from clean records
                w->epoch = 1;
                for (Tvl1LaneBufs& L : w->lanes) {
                    CK(hipMemsetAsync(L.xch.u, 0, 8 * (size_t)n0 * sizeof(float), st));
                    CK(hipMemsetAsync(L.xch.acc, 0, kSumSlots * kSumRecs * 16, st));
                }
            }
            lanes.gp = pg;
            unsigned epoch = w->epoch;
            void* args[] = {&lanes, &epoch};
            return hipLaunchCooperativeKernel(reinterpret_cast<const void*>(patch_kernel_for(f->t, f->n)), dim3(pg * cnt), dim3(f->n), args, 0, st);
        }
        // register-state kernel with a grid barrier per iteration: the smallest slot count T such that a lane has at most one
        // block per CU and all lanes together are co-resident.  With several lanes prefer the smallest T that gives every block
        // of every lane a CU of its own (cnt * g <= CUs) over two blocks per CU: two 640x360 pairs then take T = 8 on 113 CUs
        // each instead of T = 4 on all CUs twice -- 2.44 instead of 2.70 ms per flow, same bits (round 3; RVDD_TVL1_SPREAD=0
        // restores the old choice).
        int slots = 0, gp = 0;
        static const bool spread = [] { const char* e = std::getenv("RVDD_TVL1_SPREAD"); return !e || std::atoi(e) != 0; }();
        if (!w->force_mem) {
            if (spread && cnt > 1)
                for (int c : kTileSlots) {
                    const int g = (ntiles + c - 1) / c;
                    if ((long)cnt * g <= (long)w->cus) { slots = c; gp = g; break; }
                }
            if (!slots)
                for (int c : kTileSlots) {
                    const int g = (ntiles + c - 1) / c;
                    int per_cu = 0;
                    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, scale_kernel_for(c), 256, 0));
                    if (g <= w->cus && (long)cnt * g <= (long)w->cus * per_cu) { slots = c; gp = g; break; }
                }
        }
        if (slots) {
            lanes.gp = gp;
            void* args[] = {&lanes};
            return hipLaunchCooperativeKernel(reinterpret_cast<const void*>(scale_kernel_for(slots)), dim3(gp * cnt), dim3(256), args, 0, st);
        }
        if (!w->state.wx) return hipErrorInvalidValue;       // lanes of this size were not provisioned at allocation
        lanes.gp = ntiles < w->cus ? ntiles : w->cus;
        for (int q = 0; q < cnt; ++q) {                        // one lane per launch
            Lanes one = lanes;
            one.l[0] = lanes.l[q];
            void* args[] = {&one, &w->state};
            CK(hipLaunchCooperativeKernel(reinterpret_cast<const void*>(scale_kernel_mem), dim3(lanes.gp), dim3(256), args, 0, st));
        }
        return hipSuccess;
    };
    for (int s = w->nscales - 1; s >= 0; --s) {
        // as many pairs per launch as are resident together: all of them at the coarse scales (an iteration there is the
        // latency of one exchange whatever the number of pairs), two at 640x360
        const Scale& ref = w->lanes[0].sc[s];
        for (int q0 = 0; q0 < np;) {
            int cnt = np - q0, pg = 0;
            while (cnt > 1 && !patch_form(ref, cnt, &pg)) --cnt;
            // no patch form even for one pair (beyond 960x540): the kernels with a grid barrier, two pairs per launch where the
            // register-state kernel holds both (launch_scale's search), else one
            if (!patch_form(ref, cnt, &pg)) {
                cnt = np - q0 < kBarrierLanes ? np - q0 : kBarrierLanes;
                const int ntiles = (ref.nx * ref.ny + 255) / 256;
                bool fits = false;
                for (int c : kTileSlots) {
                    const int g = (ntiles + c - 1) / c;
                    int per_cu = 0;
                    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, scale_kernel_for(c), 256, 0));
                    if (g <= w->cus && (long)cnt * g <= (long)w->cus * per_cu) { fits = true; break; }
                }
                if (!fits) cnt = 1;
            }
            CK(launch_scale(s, q0, cnt));
#ifdef RVDD_STAMPS
            if (const char* e = std::getenv("RVDD_TVL1_STAMPS"); e && e[0] == '2') {      // every launch: first block of each lane
                static unsigned long long hs[512][8];
                CK(hipStreamSynchronize(st));
                CK(hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_tvl1_stamps), sizeof hs));
                int pgx = 0;
                const Form* f = patch_form(ref, cnt, &pgx);
                std::fprintf(stderr, "tvl1 scale %d (%dx%d) lanes %d form T=%d/%d blocks %d:", s, ref.nx, ref.ny, cnt, f ? f->t : 0, f ? f->n : 0, pgx);
                for (int q = 0; q < cnt && f; ++q) {
                    const unsigned long long* o = hs[q * pgx < 512 ? q * pgx : 0];
                    std::fprintf(stderr, " [%llu it, %.1f us/it: u %.1f w %.1f d %.1f, kernel %.0f us]", o[0], o[0] ? (o[1] + o[2] + o[3]) / 100.0 / o[0] : 0.0,
                                 o[0] ? o[1] / 100.0 / o[0] : 0.0, o[0] ? o[2] / 100.0 / o[0] : 0.0, o[0] ? o[3] / 100.0 / o[0] : 0.0, o[6] / 100.0);
                }
                std::fprintf(stderr, "\n");
            }
#endif
            q0 += cnt;
        }
        if (s == 0) break;
        // zoom_in + rescale by 1/zfactor (zoom.c:85-108, tvl1flow_lib.c:424-433)
        {
            const Scale& sc = w->lanes[0].sc[s];
            const Scale& f = w->lanes[0].sc[s - 1];
            const float fx = (float)f.nx / sc.nx, fy = (float)f.ny / sc.ny;
            Imgs z{};
            for (int q = 0; q < np; ++q)
                for (int c = 0; c < 2; ++c) {
                    z.in[2 * q + c] = u_of(q, s, c);
                    z.out[2 * q + c] = u_of(q, s - 1, c);
                }
            hipLaunchKernelGGL(resample_kernel, dim3((f.nx + 255) / 256, f.ny, 2 * np), dim3(256), 0, st, z, sc.nx, sc.ny, f.nx, f.ny, fx, fy,
                               1.0f / kZoom);
        }
    }
    if (finish) {   // always read the control words back: a grid barrier that gave up must not pass as a flow
        int ab = 0, c[kMaxLanes][8] = {};
        CK(hipMemcpyAsync(&ab, w->abort_word, sizeof ab, hipMemcpyDeviceToHost, st));
        for (int q = 0; q < np; ++q) CK(hipMemcpyAsync(c[q], w->lanes[q].ctl, 4 * sizeof(int), hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        w->unchecked = false;
        if (ab) return hipErrorLaunchTimeOut;
#ifdef RVDD_STAMPS
        if (std::getenv("RVDD_TVL1_STAMPS")) {
            static unsigned long long hs[512][8];
            CK(hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_tvl1_stamps), sizeof hs));
            for (int b : {0, 1, 57, 119, 120, 200}) {
                const unsigned long long* o = hs[b];
                if (!o[0]) continue;
                std::fprintf(stderr, "tvl1 stamps block %3d T=%llu iters %llu | per iteration (10 ns ticks): update %.1f wait %.1f (ring %.1f acc %.1f) dual %.1f | kernel %.1f us\n",
                             b, o[7], o[0], (double)o[1] / o[0], (double)o[2] / o[0], (double)o[4] / o[0], (double)o[5] / o[0], (double)o[3] / o[0], o[6] / 100.0);
            }
        }
#endif
        if (iters)
            for (int q = 0; q < np; ++q) iters[q] = c[q][2];
    }
    return hipGetLastError();
#undef CK
}

hipError_t tvl1_run(Tvl1Workspace* w, const float* I0, const float* I1, float* u, hipStream_t st, int* total_iters) {
    return tvl1_run_lanes(w, I0, I1, u, 1, st, total_iters);
}

// n pairs of the same size, kMaxLanes at a time.  async (and no iteration counts wanted): nothing is read back and the stream
// is not synchronised -- the flows are the stream's business, and a grid barrier that gave up (tvl1.hip grid_sync: never seen
// outside fault injection) leaves the abort word set for tvl1_check, or for the next synchronous run, to report.
hipError_t tvl1_run_batch(Tvl1Workspace* w, const float* I0, const float* I1, float* u, int n, hipStream_t st, int* iters, bool async) {
    const size_t n0 = (size_t)w->nx * w->ny;
    async = async && !iters;
    for (int q = 0; q < n; q += kMaxLanes) {
        const int np = n - q < kMaxLanes ? n - q : kMaxLanes;
        hipError_t e = tvl1_run_lanes(w, I0 + q * n0, I1 + q * n0, u + q * 2 * n0, np, st, iters ? iters + q : nullptr, q == 0,
                                      !async && (iters != nullptr || q + kMaxLanes >= n));
        if (e != hipSuccess) return e;
        if (async) w->unchecked = true;
    }
    return hipSuccess;
}

// the deferred half of an asynchronous batch: synchronise the stream and read the abort word (a no-op when nothing is pending)
hipError_t tvl1_check(Tvl1Workspace* w, hipStream_t st) {
    if (!w || !w->unchecked) return hipSuccess;
    int ab = 0;
    hipError_t e = hipMemcpyAsync(&ab, w->abort_word, sizeof ab, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return e;
    w->unchecked = false;
    if (ab) {
        (void)hipMemsetAsync(w->abort_word, 0, sizeof(int), st);
        return hipErrorLaunchTimeOut;
    }
    return hipSuccess;
}
