// Dual TV-L1 optical flow on the GPU: the reference's flow pre-computation
// (libBridge.cpp:44-163 -> 3rdparty/tvl1flow/tvl1flow_lib.c:91-278, 343-472; mask.c; zoom.c;
// bicubic_interpolation.c), SURVEY.md section 8f rank 1.  Same algorithm, same hard-wired parameters
// (tau .25, lambda .15, theta .3, zoom .5, 5 warps, <= 300 iterations, epsilon .01), same quirks
// (listed in DESIGN.md), fp32 with the reference's double-precision islands (Gaussian,
// bicubic cell, normalisation, hypot).  Compiled with -ffp-contract=off.
//
// All maps are planar fp32 [ny][nx].  The per-iteration work is three tiny HBM/latency-bound
// kernels; convergence is decided on the device (a `done` word that later iterations test), the
// host only peeks at it every few iterations to stop launching.
#include "rvdd_internal.h"

#include <cmath>
#include <vector>

namespace {

constexpr float kTau = 0.25f, kLambda = 0.15f, kTheta = 0.3f, kZoom = 0.5f, kEps = 0.01f;
constexpr int kWarps = 5, kMaxIter = 300, kMaxScales = 100;
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
__global__ void resample_kernel(const float* __restrict__ in, float* __restrict__ out, int nx, int ny, int nxx,
                                int nyy, float fx, float fy, float mul) {
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
__global__ void gauss_kernel(const float* __restrict__ in, float* __restrict__ out, int nx, int ny, GaussK k,
                             int vertical) {
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

__global__ void centered_gradient_kernel(const float* __restrict__ I, float* __restrict__ dx,
                                         float* __restrict__ dy, int nx, int ny) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= nx) return;
    const int p = i * nx + j;
    const float l = I[j > 0 ? p - 1 : p], r = I[j < nx - 1 ? p + 1 : p];
    const float u = I[i > 0 ? p - nx : p], d = I[i < ny - 1 ? p + nx : p];
    dx[p] = 0.5f * (r - l);          // one-sided at the borders with the SAME 0.5 factor (mask.c:170-205)
    dy[p] = 0.5f * (d - u);
}

// min / max of both images (image_normalization, tvl1flow_lib.c:300-333)
__global__ void minmax_kernel(const float* __restrict__ a, const float* __restrict__ b, int n, float* __restrict__ mm) {
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
__global__ void normalize_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ an,
                                 float* __restrict__ bn, int n, const float* __restrict__ mm) {
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
struct Scale {
    int nx, ny;
    float *I0, *I1, *u1, *u2;
};
struct IterBufs {
    float *I1x, *I1y, *I1w, *I1wx, *I1wy, *rho_c, *grad, *p11, *p12, *p21, *p22;
};

// warp of I1, I1x, I1y by (u1,u2) with zero outside + grad, rho_c (tvl1flow_lib.c:144-163)
__global__ void warp_rho_kernel(Scale s, IterBufs b, int* __restrict__ ctl) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j == 0 && i == 0) { ctl[0] = 0; ctl[1] = 0; }     // done flag, iteration counter of this warping
    if (j >= s.nx) return;
    const int p = i * s.nx + j;
    const float u1 = s.u1[p], u2 = s.u2[p];
    Bicubic bc;
    bc.setup((float)((float)j + u1), (float)((float)i + u2), s.nx, s.ny);
    float w = 0.f, wx = 0.f, wy = 0.f;
    if (!bc.out) {
        w = bc.eval(s.I1, s.nx);
        wx = bc.eval(b.I1x, s.nx);
        wy = bc.eval(b.I1y, s.nx);
    }
    b.I1w[p] = w;
    b.I1wx[p] = wx;
    b.I1wy[p] = wy;
    const float Ix2 = wx * wx, Iy2 = wy * wy;
    b.grad[p] = Ix2 + Iy2;
    b.rho_c[p] = w - wx * u1 - wy * u2 - s.I0[p];
}

// thresholding step, divergence of p, update of u, error (tvl1flow_lib.c:171-215)
__global__ void iter_u_kernel(Scale s, IterBufs b, const int* __restrict__ ctl, float* __restrict__ partial) {
    if (ctl[0]) return;
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    float e = 0.f;
    if (j < s.nx) {
        const int nx = s.nx, ny = s.ny, p = i * nx + j;
        const float l_t = kLambda * kTheta;
        const float u1k = s.u1[p], u2k = s.u2[p];
        const float wx = b.I1wx[p], wy = b.I1wy[p], g = b.grad[p];
        const float rho = b.rho_c[p] + (wx * u1k + wy * u2k);
        float d1, d2;
        if (rho < -l_t * g) {
            d1 = l_t * wx;
            d2 = l_t * wy;
        } else if (rho > l_t * g) {
            d1 = -l_t * wx;
            d2 = -l_t * wy;
        } else if (g < kGradIsZero) {
            d1 = d2 = 0.f;
        } else {
            const float fi = -rho / g;
            d1 = fi * wx;
            d2 = fi * wy;
        }
        const float v1 = u1k + d1, v2 = u2k + d2;
        // divergence (mask.c:40-90)
        auto dive = [&](const float* a, const float* c) {
            const float ax = j == 0 ? a[p] : (j == nx - 1 ? -a[p - 1] : a[p] - a[p - 1]);
            const float cy = i == 0 ? c[p] : (i == ny - 1 ? -c[p - nx] : c[p] - c[p - nx]);
            return ax + cy;
        };
        const float n1 = v1 + kTheta * dive(b.p11, b.p12);
        const float n2 = v2 + kTheta * dive(b.p21, b.p22);
        s.u1[p] = n1;
        s.u2[p] = n2;
        e = (n1 - u1k) * (n1 - u1k) + (n2 - u2k) * (n2 - u2k);
    }
    for (int o = 32; o > 0; o >>= 1) e += __shfl_down(e, o);
    __shared__ float sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = e;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.y * gridDim.x + blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ void iter_check_kernel(int* __restrict__ ctl, const float* __restrict__ partial, int nblk, int size) {
    if (ctl[0]) {                 // converged in an earlier pass: this pass did not move u, p must not move either
        if (threadIdx.x == 0) ctl[3] = 0;
        return;
    }
    double acc = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) acc += (double)partial[i];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    __shared__ double sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float error = (float)(((sh[0] + sh[1]) + (sh[2] + sh[3])) / (double)size);
        const int n = ctl[1] + 1;
        ctl[1] = n;
        ctl[2] += 1;                                   // total iterations (statistics)
        ctl[3] = 1;                                    // u moved in this pass -> its p update runs
        if (!(error > kEps * kEps) || n >= kMaxIter) ctl[0] = 1;
    }
}

// forward gradient of u, update of the dual variables (tvl1flow_lib.c:217-234)
__global__ void iter_p_kernel(Scale s, IterBufs b, const int* __restrict__ ctl) {
    if (!ctl[3]) return;          // the reference updates p in every pass that updated u, the last one included
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= s.nx) return;
    const int nx = s.nx, ny = s.ny, p = i * nx + j;
    const float taut = kTau / kTheta;
    const float a = s.u1[p], c = s.u2[p];
    const float u1x = j < nx - 1 ? s.u1[p + 1] - a : 0.f, u1y = i < ny - 1 ? s.u1[p + nx] - a : 0.f;
    const float u2x = j < nx - 1 ? s.u2[p + 1] - c : 0.f, u2y = i < ny - 1 ? s.u2[p + nx] - c : 0.f;
    const float g1 = (float)hypot((double)u1x, (double)u1y);
    const float g2 = (float)hypot((double)u2x, (double)u2y);
    const float ng1 = (float)(1.0 + (double)(taut * g1));
    const float ng2 = (float)(1.0 + (double)(taut * g2));
    b.p11[p] = (b.p11[p] + taut * u1x) / ng1;
    b.p12[p] = (b.p12[p] + taut * u1y) / ng1;
    b.p21[p] = (b.p21[p] + taut * u2x) / ng2;
    b.p22[p] = (b.p22[p] + taut * u2y) / ng2;
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

// Workspace: every buffer the pyramid needs for one (nx, ny); owned by the caller (runtime.hip).
struct Tvl1Workspace {
    int nx = 0, ny = 0, nscales = 0;
    std::vector<Scale> sc;
    IterBufs it{};
    float *tmp = nullptr, *tmp2 = nullptr, *partial = nullptr, *mm = nullptr;
    int* ctl = nullptr;
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

hipError_t tvl1_alloc(Tvl1Workspace** out, int nx, int ny) {
    Tvl1Workspace* w = new Tvl1Workspace();
    w->nx = nx;
    w->ny = ny;
    w->nscales = tvl1_num_scales(nx, ny);
    hipError_t err = hipSuccess;
    auto A = [&](float** p, size_t n) {
        if (err == hipSuccess) {
            err = hipMalloc(reinterpret_cast<void**>(p), n * sizeof(float));
            if (err == hipSuccess) w->allocs.push_back(*p);
        }
    };
    int sx = nx, sy = ny;
    for (int s = 0; s < w->nscales; ++s) {
        Scale sc{};
        sc.nx = sx;
        sc.ny = sy;
        const size_t n = (size_t)sx * sy;
        A(&sc.I0, n);
        A(&sc.I1, n);
        A(&sc.u1, n);
        A(&sc.u2, n);
        w->sc.push_back(sc);
        sx = (int)((float)sx * kZoom + 0.5f);       // zoom_size (zoom.c:22-34)
        sy = (int)((float)sy * kZoom + 0.5f);
    }
    const size_t n0 = (size_t)nx * ny;
    float** its[] = {&w->it.I1x, &w->it.I1y, &w->it.I1w, &w->it.I1wx, &w->it.I1wy, &w->it.rho_c, &w->it.grad,
                     &w->it.p11, &w->it.p12, &w->it.p21, &w->it.p22, &w->tmp, &w->tmp2};
    for (float** p : its) A(p, n0);
    A(&w->partial, ((size_t)(nx + 255) / 256) * ny);
    A(&w->mm, 4);
    if (err == hipSuccess) {
        err = hipMalloc(reinterpret_cast<void**>(&w->ctl), 4 * sizeof(int));
        if (err == hipSuccess) w->allocs.push_back(w->ctl);
    }
    if (err != hipSuccess) {
        tvl1_free(w);
        return err;
    }
    *out = w;
    return hipSuccess;
}

// Dual_TVL1_optic_flow_multiscale (tvl1flow_lib.c:343-472).  u = [u(ny*nx), v(ny*nx)] as libBridge.cpp:150.
hipError_t tvl1_run(Tvl1Workspace* w, const float* I0, const float* I1, float* u, hipStream_t st, int* total_iters) {
#define CK(e)                                 \
    do {                                      \
        hipError_t e__ = (e);                 \
        if (e__ != hipSuccess) return e__;    \
    } while (0)
    const int nx = w->nx, ny = w->ny, n0 = nx * ny;
    // normalisation to [0,255] and pre-smoothing
    const int init[2] = {0x7f7fffff, (int)0x80000000};   // order-preserving keys of +FLT_MAX / -FLT_MAX
    CK(hipMemcpyAsync(w->mm, init, sizeof init, hipMemcpyHostToDevice, st));
    CK(hipMemsetAsync(w->ctl, 0, 4 * sizeof(int), st));
    hipLaunchKernelGGL(minmax_kernel, dim3(256), dim3(256), 0, st, I0, I1, n0, w->mm);
    hipLaunchKernelGGL(normalize_kernel, dim3((n0 + 255) / 256), dim3(256), 0, st, I0, I1, w->tmp, w->tmp2, n0, w->mm);
    auto gauss = [&](const float* in, float* outp, int gx, int gy, double sigma, float* scratch) {
        const GaussK k = make_gauss(sigma);
        hipLaunchKernelGGL(gauss_kernel, grid2(gx, gy), dim3(256), 0, st, in, scratch, gx, gy, k, 0);
        hipLaunchKernelGGL(gauss_kernel, grid2(gx, gy), dim3(256), 0, st, scratch, outp, gx, gy, k, 1);
    };
    gauss(w->tmp, w->sc[0].I0, nx, ny, kPresmooth, w->it.I1w);
    gauss(w->tmp2, w->sc[0].I1, nx, ny, kPresmooth, w->it.I1w);
    // pyramid (zoom_out, zoom.c:41-78)
    const double zsigma = (double)(float)(kZoomSigma0 * std::sqrt(1.0 / ((double)kZoom * (double)kZoom) - 1.0));
    for (int s = 1; s < w->nscales; ++s) {
        const Scale& a = w->sc[s - 1];
        const Scale& b = w->sc[s];
        gauss(a.I0, w->tmp, a.nx, a.ny, zsigma, w->it.I1w);
        hipLaunchKernelGGL(resample_kernel, grid2(b.nx, b.ny), dim3(256), 0, st, w->tmp, b.I0, a.nx, a.ny, b.nx, b.ny, kZoom, kZoom, 1.f);
        gauss(a.I1, w->tmp, a.nx, a.ny, zsigma, w->it.I1w);
        hipLaunchKernelGGL(resample_kernel, grid2(b.nx, b.ny), dim3(256), 0, st, w->tmp, b.I1, a.nx, a.ny, b.nx, b.ny, kZoom, kZoom, 1.f);
    }
    Scale& top = w->sc[w->nscales - 1];
    CK(hipMemsetAsync(top.u1, 0, (size_t)top.nx * top.ny * sizeof(float), st));
    CK(hipMemsetAsync(top.u2, 0, (size_t)top.nx * top.ny * sizeof(float), st));

    for (int s = w->nscales - 1; s >= 0; --s) {
        Scale sc = w->sc[s];
        if (s == 0) {        // the finest flow is the caller's buffer
            CK(hipMemcpyAsync(u, sc.u1, (size_t)n0 * sizeof(float), hipMemcpyDeviceToDevice, st));
            CK(hipMemcpyAsync(u + n0, sc.u2, (size_t)n0 * sizeof(float), hipMemcpyDeviceToDevice, st));
            sc.u1 = u;
            sc.u2 = u + n0;
        }
        const size_t n = (size_t)sc.nx * sc.ny;
        const dim3 g = grid2(sc.nx, sc.ny);
        const int nblk = g.x * g.y;
        hipLaunchKernelGGL(centered_gradient_kernel, g, dim3(256), 0, st, sc.I1, w->it.I1x, w->it.I1y, sc.nx, sc.ny);
        for (float* p : {w->it.p11, w->it.p12, w->it.p21, w->it.p22}) CK(hipMemsetAsync(p, 0, n * sizeof(float), st));
        for (int wp = 0; wp < kWarps; ++wp) {
            hipLaunchKernelGGL(warp_rho_kernel, g, dim3(256), 0, st, sc, w->it, w->ctl);
            for (int it = 0; it < kMaxIter; ++it) {
                hipLaunchKernelGGL(iter_u_kernel, g, dim3(256), 0, st, sc, w->it, w->ctl, w->partial);
                hipLaunchKernelGGL(iter_check_kernel, dim3(1), dim3(256), 0, st, w->ctl, w->partial, nblk, (int)n);
                // later passes are no-ops once the done word is set (ctl[3] gates the p update)
                hipLaunchKernelGGL(iter_p_kernel, g, dim3(256), 0, st, sc, w->it, w->ctl);
                if ((it & 7) == 7) {             // peek at the done word to stop launching
                    int done = 0;
                    CK(hipMemcpyAsync(&done, w->ctl, sizeof(int), hipMemcpyDeviceToHost, st));
                    CK(hipStreamSynchronize(st));
                    if (done) break;
                }
            }
        }
        if (s == 0) break;
        // zoom_in + rescale by 1/zfactor (zoom.c:85-108, tvl1flow_lib.c:424-433)
        const Scale& f = w->sc[s - 1];
        const float fx = (float)f.nx / sc.nx, fy = (float)f.ny / sc.ny;
        hipLaunchKernelGGL(resample_kernel, grid2(f.nx, f.ny), dim3(256), 0, st, sc.u1, f.u1, sc.nx, sc.ny, f.nx, f.ny, fx, fy, 1.0f / kZoom);
        hipLaunchKernelGGL(resample_kernel, grid2(f.nx, f.ny), dim3(256), 0, st, sc.u2, f.u2, sc.nx, sc.ny, f.nx, f.ny, fx, fy, 1.0f / kZoom);
    }
    if (total_iters) {
        int c[4];
        CK(hipMemcpyAsync(c, w->ctl, sizeof c, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        *total_iters = c[2];
    }
    return hipGetLastError();
#undef CK
}
