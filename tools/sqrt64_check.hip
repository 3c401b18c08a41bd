// GPU box: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/sqrt64_check.hip -o /tmp/sq && /tmp/sq
// tvl1.hip's sqrt_sumsq (the double-precision root of x*x + y*y without the range scaling and special-case tests of the
// compiler's expansion, which sums of squares of floats never need) against __builtin_sqrt, bit for bit, over random and
// edge-case float pairs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>
#include <cstring>

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

__global__ void check(const float* x, const float* y, int n, unsigned long long* bad, double* first) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double xd = (double)x[i], yd = (double)y[i];
    const double S = xd * xd + yd * yd;
    const double a = __builtin_sqrt(S), b = sqrt_sumsq(S);
    if (__double_as_longlong(a) != __double_as_longlong(b)) {
        if (atomicAdd(bad, 1ull) == 0) { first[0] = S; first[1] = a; first[2] = b; }
    }
}

int main() {
    const int n = 1 << 24;
    std::vector<float> x(n), y(n);
    std::mt19937_64 rng(7);
    for (int i = 0; i < n; ++i) {
        // random bit patterns of finite floats (all exponents, denormals included), some exact zeros and equal pairs
        uint32_t a = (uint32_t)rng(), b = (uint32_t)rng();
        if ((a >> 23 & 0xff) == 0xff) a &= 0x7f7fffff;
        if ((b >> 23 & 0xff) == 0xff) b &= 0x7f7fffff;
        std::memcpy(&x[i], &a, 4);
        std::memcpy(&y[i], &b, 4);
        if (i % 97 == 0) x[i] = 0.f;
        if (i % 193 == 0) y[i] = 0.f;
        if (i % 389 == 0) y[i] = x[i];
        if (i % 5 == 0) { x[i] = (float)((int)(rng() % 2001) - 1000) * 1e-3f; y[i] = (float)((int)(rng() % 2001) - 1000) * 1e-3f; }   // the kernel's own range
    }
    float *dx, *dy; unsigned long long* dbad; double* dfirst;
    hipMalloc(&dx, n * 4); hipMalloc(&dy, n * 4); hipMalloc(&dbad, 8); hipMalloc(&dfirst, 24);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dy, y.data(), n * 4, hipMemcpyHostToDevice);
    hipMemset(dbad, 0, 8);
    hipLaunchKernelGGL(check, dim3(n / 256), dim3(256), 0, 0, dx, dy, n, dbad, dfirst);
    unsigned long long bad = 0; double first[3] = {};
    hipMemcpy(&bad, dbad, 8, hipMemcpyDeviceToHost); hipMemcpy(first, dfirst, 24, hipMemcpyDeviceToHost);
    std::printf("sqrt_sumsq vs __builtin_sqrt over %d pairs: %llu differ", n, bad);
    if (bad) std::printf(" (first: S = %a, builtin %a, custom %a)", first[0], first[1], first[2]);
    std::printf("\n");
    return bad ? 1 : 0;
}
