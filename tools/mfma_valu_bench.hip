// Microbenchmark (gfx950): how f32 MFMA and f32 VALU work share a SIMD, with one and with two waves per SIMD.
//
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_valu_bench.hip -o /tmp/mvb && /tmp/mvb
//
// Every wave repeats ITER times: NM v_mfma_f32_16x16x4_f32 on 12 independent accumulators, then NV vector
// instructions (v_add_f32 or v_pk_add_f32, independent registers).  One workgroup per CU (96 KiB of LDS claimed),
// 256 threads = one wave per SIMD or 512 threads = two.  Reported: shader cycles (s_memtime) per iteration of one
// wave, median over workgroups, and the same divided by the waves per SIMD = cycles the SIMD spends per
// (NM MFMA + NV VALU) of work.  "split": waves 0-3 issue only the MFMAs, waves 4-7 only the vector instructions;
// "split + prio": the same with s_setprio 3 on the vector waves (round 3: would a producer wave's vector bursts get
// through beside a consumer wave's MFMAs at their own cost?).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int ITER = 2000;

template <int NM, int NV, bool PK, int MODE>   // MODE 0: every wave does both; 1: split roles (needs 512 threads); 2: split + s_setprio 3 on the vector waves
__global__ __launch_bounds__(512) void bench(float* sink, long long* cycles) {
    extern __shared__ float lds[];
    const int wave = threadIdx.x >> 6;
    f32x4 acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f + threadIdx.x * 1e-4f;
    f32x2 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = f32x2{(float)i, (float)threadIdx.x};
    const f32x2 inc = {1e-3f, 2e-3f};
    const bool do_m = MODE == 0 || wave < 4;
    const bool do_v = MODE == 0 || wave >= 4;
    if (MODE == 2 && wave >= 4) __builtin_amdgcn_s_setprio(3);
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < ITER; ++it) {
        if (do_m) {
#pragma unroll
            for (int i = 0; i < NM; ++i) acc[i % 12] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i % 12], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (do_v) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                if (PK) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[i % 16]) : "v"(inc));
                else asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i % 16][0]) : "v"(inc[0]));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i][0] + v[i][1];
    if (s == 12345.678f) sink[0] = s + lds[threadIdx.x];
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int NM, int NV, bool PK, int MODE>
void run(int threads, const char* label, float* sink, long long* dcyc) {
    auto k = bench<NM, NV, PK, MODE>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    const int blocks = 256;
    hipMemset(dcyc, 0, blocks * 8 * sizeof(long long));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 96 * 1024, 0, sink, dcyc);   // warm-up
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 96 * 1024, 0, sink, dcyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks * 8);
    hipMemcpy(h.data(), dcyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    std::vector<double> per;
    const int waves = threads / 64;
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < waves; ++w) per.push_back((double)h[b * 8 + w] / ITER);
    std::sort(per.begin(), per.end());
    const double med = per[per.size() / 2];
    const int wps = threads / 256;
    const double per_work = MODE >= 1 ? med : med / wps;     // split: one iteration of the pair = one unit of work
    printf("%-34s NM=%2d NV=%2d %s thr=%3d  cyc/iter/wave %7.1f  SIMD cyc per (NM mfma + NV valu) %7.1f  [mfma alone %d]  wall %.3f ms\n",
           label, NM, NV, PK ? "pk " : "f32", threads, med, per_work, NM * 32, ms);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
}

int main() {
    float* sink;
    long long* dcyc;
    hipMalloc(&sink, 64);
    hipMalloc(&dcyc, 256 * 8 * sizeof(long long));
#define ROW(NV)                                                      \
    run<12, NV, false, 0>(256, "both, 1 wave/SIMD", sink, dcyc);     \
    run<12, NV, false, 0>(512, "both, 2 waves/SIMD", sink, dcyc);    \
    run<12, NV, false, 1>(512, "split roles, 2 waves/SIMD", sink, dcyc);  \
    run<12, NV, false, 2>(512, "split roles + prio", sink, dcyc);
    ROW(0) ROW(4) ROW(8) ROW(16) ROW(32) ROW(64)
#define ROWP(NV)                                                    \
    run<12, NV, true, 0>(256, "both, 1 wave/SIMD", sink, dcyc);      \
    run<12, NV, true, 0>(512, "both, 2 waves/SIMD", sink, dcyc);     \
    run<12, NV, true, 1>(512, "split roles, 2 waves/SIMD", sink, dcyc);   \
    run<12, NV, true, 2>(512, "split roles + prio", sink, dcyc);
    ROWP(4) ROWP(8) ROWP(16) ROWP(32)
    // VALU alone
    run<0, 64, false, 0>(256, "valu only, 1 wave/SIMD", sink, dcyc);
    run<0, 64, false, 0>(512, "valu only, 2 waves/SIMD", sink, dcyc);
    run<0, 32, true, 0>(256, "pk valu only, 1 wave/SIMD", sink, dcyc);
    run<0, 32, true, 0>(512, "pk valu only, 2 waves/SIMD", sink, dcyc);
    return 0;
}
