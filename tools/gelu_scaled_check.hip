// gelu_phi4_scaled (convnext.hip: the GELU on fc1's accumulator in the filters' power-of-two scale, constants scaled on the
// host) against gelu_phi4 on the unscaled argument, on the device: bit-identical over [-2.1, 2.1] x scale.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Irvdd-release_amd/csrc tools/gelu_scaled_check.hip -o /tmp/gt && /tmp/gt
#include "../rvdd-release_amd/csrc/convnext.hip"
#include <cstdio>
#include <cmath>
__global__ void k(const float* in, float* o1, float* o2, NextBlockW wt) {
    f32x4 v = *reinterpret_cast<const f32x4*>(in + threadIdx.x * 4);
    f32x4 a = gelu_phi4(v * wt.fc1_inv) * wt.fc1_scale;
    f32x4 b = gelu_phi4_scaled(v, wt.gelu_c);
    *reinterpret_cast<f32x4*>(o1 + threadIdx.x * 4) = a;
    *reinterpret_cast<f32x4*>(o2 + threadIdx.x * 4) = b;
}
int main() {
    NextBlockW w{};
    const int s1 = -2;
    w.fc1_scale = ldexpf(1.f, s1); w.fc1_inv = ldexpf(1.f, -s1);
    static const float C[6] = {2.992418740177527e-05f, -0.0007398742018267512f, 0.007977462373673916f, -0.05323818698525429f, -0.45891568064689636f, -1.1511471271514893f};
    for (int i = 0; i < 6; ++i) w.gelu_c[i][0] = w.gelu_c[i][1] = ldexpf(C[i], -s1 * (6 - i));
    w.gelu_c[6][0] = w.gelu_c[6][1] = ldexpf(6.36f, s1);
    float h[256], a[256], b[256];
    for (int i = 0; i < 256; ++i) h[i] = (i - 128) / 60.0f;
    float *d, *d1, *d2;
    hipMalloc(&d, 1024); hipMalloc(&d1, 1024); hipMalloc(&d2, 1024);
    hipMemcpy(d, h, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, d1, d2, w);
    hipMemcpy(a, d1, 1024, hipMemcpyDeviceToHost); hipMemcpy(b, d2, 1024, hipMemcpyDeviceToHost);
    double mx = 0; int at = 0;
    for (int i = 0; i < 256; ++i) if (fabs(a[i] - b[i]) > mx) { mx = fabs(a[i] - b[i]); at = i; }
    printf("max diff %g at h=%g: %g vs %g\n", mx, h[at], a[at], b[at]);
    for (int i = 0; i < 256; i += 37) printf("h %g: %g %g\n", h[i], a[i], b[i]);
    return 0;
}
