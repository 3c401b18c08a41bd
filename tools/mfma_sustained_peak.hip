// Sustained dense f32 MFMA rate of the whole chip under its power limit: every SIMD issues independent
// v_mfma_f32_16x16x4_f32 back to back on non-trivial data for a few seconds; HIP events give TFLOP/s, the caller samples
// rocm-smi beside it (tools/power_probe-style).  hipcc -O3 --offload-arch=gfx950 tools/mfma_sustained_peak.hip -o /tmp/msp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void burn(float* out, int iters, float seed) {
    const int lane = threadIdx.x;
    // a[2] = -a[0], a[3] = -a[1]: the four MFMAs of an accumulator's round cancel, so the sums stay bounded with NO
    // vector instruction in the loop (a rescaling multiply per accumulator costs the MFMA pipe a third of its slots)
    float a[4], b[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        a[i] = __sinf(seed + 0.37f * (lane * 4 + i)) * 0.9f;
        a[i + 2] = -a[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) b[i] = __cosf(seed + 0.11f * (lane * 4 + i + blockIdx.x)) * 0.9f;
    f32x4 acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], b[k & 3], acc[k], 0, 0, 0);
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) s += acc[k];
    if (s[0] == 12345.678f) out[blockIdx.x * 256 + lane] = s[1] + s[2] + s[3];
}

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 4.0;
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    float* out;
    hipMalloc(&out, (size_t)cus * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;                       // 32 MFMAs x 2048 flop per iteration and wave
    burn<<<cus, 256>>>(out, 1000, 0.1f);
    hipDeviceSynchronize();
    double total_ms = 0, flops = 0;
    int launches = 0;
    while (total_ms < secs * 1e3) {
        hipEventRecord(e0);
        for (int k = 0; k < 8; ++k) burn<<<cus, 256>>>(out, iters, 0.1f * (k + 1));
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        total_ms += ms;
        launches += 8;
        flops += 8.0 * cus * 4 * (double)iters * 32 * 2048;
        printf("%.2f s: %.1f TFLOP/s over the last %.0f ms\n", total_ms / 1e3, 8.0 * cus * 4 * (double)iters * 32 * 2048 / (ms * 1e9), ms);
        fflush(stdout);
    }
    printf("{\"kernel\": \"v_mfma_f32_16x16x4_f32, 8 independent accumulators, nothing else in the loop, 1 wave per SIMD, %d CUs\", \"seconds\": %.2f, \"tflops\": %.2f, \"of_157.3\": %.3f}\n",
           cus, total_ms / 1e3, flops / (total_ms * 1e9), flops / (total_ms * 1e9) / 157.3);
    return 0;
}
