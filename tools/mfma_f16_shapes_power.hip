// Which F16 MFMA shape does more work per joule?  The whole chip runs independent MFMAs of one shape on non-trivial data for a
// few seconds at its power limit; the sustained TFLOP/s IS the efficiency (the limit is the same).  Shapes: 16x16x32 (the
// split-f16 kernels' shape) and 32x32x16, one and two waves per SIMD, operands "dense" (random mantissas) and "relu-like"
// (45 % zeros in the B operand).
//   hipcc -O3 --offload-arch=gfx950 -Wno-unused-value tools/mfma_f16_shapes_power.hip -o /tmp/mfp && /tmp/mfp [seconds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int SHAPE, bool SPARSE>
__global__ __launch_bounds__(512) void burn(float* out, int iters, float seed) {
    const int lane = threadIdx.x;
    h8 a[4], b[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // a[2] = -a[0], a[3] = -a[1]: the sums stay bounded with no vector instruction in the loop
            const float av = __sinf(seed + 0.37f * (lane * 8 + i + 64 * (r & 1))) * 0.9f;
            a[r][i] = (_Float16)((r & 2) ? -av : av);
            float bv = __cosf(seed + 0.11f * (lane * 8 + i + 17 * r + blockIdx.x)) * 0.9f;
            if (SPARSE && bv < 0.1f) bv = 0.f;
            b[r][i] = (_Float16)bv;
        }
    if constexpr (SHAPE == 0) {
        f32x4 acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[r], b[k & 3], acc[k], 0, 0, 0);
        }
        f32x4 s = acc[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) s += acc[k];
        if (s[0] == 12345.678f) out[blockIdx.x * 512 + lane] = s[1] + s[2] + s[3];
    } else {
        f32x16 acc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[r], b[k], acc[k], 0, 0, 0);
        }
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < 16; ++i) s += acc[k][i];
        if (s == 12345.678f) out[blockIdx.x * 512 + lane] = s;
    }
}

template <int SHAPE, bool SPARSE>
void run(const char* name, int threads, double secs, int cus, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    // flop per iteration and wave: 32 x 16384 (16x16x32) or 16 x 32768 (32x32x16): the same
    const double flop_launch = (double)cus * (threads / 64) * iters * 32.0 * 16384.0;
    hipLaunchKernelGGL((burn<SHAPE, SPARSE>), dim3(cus), dim3(threads), 0, 0, out, 1000, 0.1f);
    hipDeviceSynchronize();
    double total_ms = 0, flops = 0;
    float last = 0;
    while (total_ms < secs * 1e3) {
        hipEventRecord(e0);
        for (int k = 0; k < 4; ++k) hipLaunchKernelGGL((burn<SHAPE, SPARSE>), dim3(cus), dim3(threads), 0, 0, out, iters, 0.1f * (k + 1));
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        total_ms += ms;
        flops += 4.0 * flop_launch;
        last = (float)(4.0 * flop_launch / (ms * 1e9));
    }
    printf("%-52s %d wave(s)/SIMD: sustained %7.1f TFLOP/s over %.1f s (last interval %7.1f)\n", name, threads / 256, flops / (total_ms * 1e9),
           total_ms / 1e3, last);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 3.0;
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    float* out;
    hipMalloc(&out, (size_t)cus * 512 * 4);
    for (int threads : {256, 512}) {
        run<0, false>("v_mfma_f32_16x16x32_f16, dense operands", threads, secs, cus, out);
        run<1, false>("v_mfma_f32_32x32x16_f16, dense operands", threads, secs, cus, out);
        run<0, true>("v_mfma_f32_16x16x32_f16, B 45 % zeros", threads, secs, cus, out);
        run<1, true>("v_mfma_f32_32x32x16_f16, B 45 % zeros", threads, secs, cus, out);
    }
    return 0;
}
