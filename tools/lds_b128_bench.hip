// Microbenchmark (gfx950): which address patterns of ds_read_b128 are bank-conflict free.  A B fragment of
// v_mfma_f32_16x16x32_f16 is 16 bytes per lane at  (lane & 15) * SN + (lane >> 4) * SG  (16 pixels x 4 channel groups);
// the kernels choose SN (LDS bytes per pixel) and the placement of the groups.  One workgroup per CU, 4 or 8 waves, each
// wave issuing NREAD reads per iteration into independent registers; reported: LDS bytes per clock per CU.
//
//   hipcc -O3 --offload-arch=gfx950 tools/lds_b128_bench.hip -o /tmp/ldsb && /tmp/ldsb
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int ITER = 1000, NREAD = 16;

__global__ __launch_bounds__(512) void bench(float* sink, long long* cycles, int sn, int o0, int o1, int o2, int o3) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 160 * 1024 / 4; i += blockDim.x) ((float*)lds)[i] = (float)i;
    __syncthreads();
    const int g = lane >> 4;
    unsigned base = (unsigned)((lane & 15) * sn + (g == 0 ? o0 : g == 1 ? o1 : g == 2 ? o2 : o3));
    (void)wave;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < ITER; ++it) {
        f32x4 v[NREAD];
#pragma unroll
        for (int i = 0; i < NREAD; ++i)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[i]) : "v"(base), "n"(i * 4096));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < NREAD; ++i) acc += v[i];
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
    if (lane == 0) cycles[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
    float* sink;
    long long* dcyc;
    hipMalloc(&sink, 64);
    hipMalloc(&dcyc, 256 * 8 * sizeof(long long));
    hipFuncSetAttribute(reinterpret_cast<const void*>(bench), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    struct P { int sn, o[4]; const char* what; };
    std::vector<P> pats = {{16, {0, 256, 512, 768}, "lane-linear (A fragments)"}};
    for (int S : {96, 160, 192, 208, 224, 240, 256, 272, 288}) {
        pats.push_back({S, {0, 16, 32, 48}, "groups 0-3 of one tap"});
        pats.push_back({S, {32, 48, 64, 80}, "groups 2-5 of one tap"});
        pats.push_back({S, {64, 80, S, S + 16}, "groups 4,5 of a tap and 0,1 of the next pixel"});
        pats.push_back({S, {64, 80, 16 * S, 16 * S + 16}, "groups 4,5 of a tap and 0,1 of the pixel 16 further (next row)"});
        pats.push_back({S, {0, 16, S, S + 16}, "groups 0,1 of two neighbouring pixels"});
        pats.push_back({S, {0, S, 2 * S, 18 * S}, "group 0 of four taps"});
    }
    for (int threads : {512})
        for (const P& p : pats) {
            hipMemset(dcyc, 0, 256 * 8 * 8);
            hipLaunchKernelGGL(bench, dim3(256), dim3(threads), 160 * 1024, 0, sink, dcyc, p.sn, p.o[0], p.o[1], p.o[2], p.o[3]);
            hipLaunchKernelGGL(bench, dim3(256), dim3(threads), 160 * 1024, 0, sink, dcyc, p.sn, p.o[0], p.o[1], p.o[2], p.o[3]);
            hipDeviceSynchronize();
            std::vector<long long> h(256 * 8);
            hipMemcpy(h.data(), dcyc, h.size() * 8, hipMemcpyDeviceToHost);
            std::vector<double> per;
            for (int b = 0; b < 256; ++b)
                for (int w = 0; w < threads / 64; ++w) per.push_back((double)h[b * 8 + w] / (ITER * NREAD));
            std::sort(per.begin(), per.end());
            const double med = per[per.size() / 2];
            printf("waves %d  S %4d off {%5d %5d %5d %5d}  %-64s %6.2f cyc/read/wave = %6.1f B/clk/CU\n", threads / 64, p.sn, p.o[0], p.o[1],
                   p.o[2], p.o[3], p.what, med, 1024.0 * (threads / 64) / med);
        }
    return 0;
}
