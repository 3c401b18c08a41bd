// Microbenchmark (gfx950): the f16 matrix pipe as an f32 GEMM engine.  An f32 operand x is split x = hi + lo with hi, lo
// in f16 (hi rounded toward zero, lo = x - hi exactly, rounded to nearest) and a product is hi.hi + hi.lo + lo.hi on
// v_mfma_f32_16x16x32_f16 with f32 accumulation.  Questions answered here, each needed by the conv / MLP kernels:
//   1. cycles per MFMA of the f16 forms next to v_mfma_f32_16x16x4_f32 (one wave per SIMD, back to back);
//   2. how many plain vector instructions of the SAME wave fit between two f16 MFMAs without stretching them, and what a
//      second wave per SIMD issuing only vector instructions costs (round 5: nothing -- the 52-55 cycles per MFMA the first version
//      reported for split roles were its own two branches per MFMA slot; the role is now decided outside the loop) (the f32 MFMA runs on the SIMD's f32 lanes, so there
//      the two add; the f16 MFMA should not);
//   3. does the f16 MFMA keep subnormal operands (lo of a small x is subnormal), and how close is the 3-product sum to
//      the exact product of the f32 values.
//
//   hipcc -O3 --offload-arch=gfx950 tools/f16_mfma_bench.hip -o /tmp/f16b && /tmp/f16b
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int ITER = 2000;

// KIND 0: f32 16x16x4, 1: f16 16x16x32, 2: f16 16x16x16, 3: f16 32x32x16
// MODE 0: every wave issues NM MFMAs with NV vector instructions spread between them; 1: waves 0-3 MFMAs, waves 4-7 vector only;
// 2: as 1 with s_setprio 3 on the MFMA waves; 3: as 1 with s_setprio 3 on the vector waves (round 5: which way does the arbiter lean?)
template <int KIND, int NM, int NV, int MODE>
__global__ __launch_bounds__(512) void bench(float* sink, long long* cycles) {
    extern __shared__ float lds[];
    const int wave = threadIdx.x >> 6;
    f32x4 acc[8];
    f32x16 acc32[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc32[i][j] = 0.f;
    float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f + threadIdx.x * 1e-4f;
    f16x8 a8, b8;
    f16x4 a4, b4;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a8[j] = (_Float16)(0.01f * (threadIdx.x % 7 + j));
        b8[j] = (_Float16)(0.02f * (threadIdx.x % 5 + j));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        a4[j] = a8[j];
        b4[j] = b8[j];
    }
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = (float)i + threadIdx.x;
    const float inc = 1e-3f;
    const bool do_m = MODE == 0 || wave < 4;
    const bool do_v = MODE == 0 || wave >= 4;
    if (MODE == 2 && wave < 4) __builtin_amdgcn_s_setprio(3);
    if (MODE == 3 && wave >= 4) __builtin_amdgcn_s_setprio(3);
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    auto mfma_slot = [&](int i) {
        if (KIND == 0) acc[i % 8] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i % 8], 0, 0, 0);
        if (KIND == 1) acc[i % 8] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[i % 8], 0, 0, 0);
        if (KIND == 2) acc[i % 8] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[i % 8], 0, 0, 0);
        if (KIND == 3) acc32[i % 2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, acc32[i % 2], 0, 0, 0);
    };
    if (MODE == 0) {
#pragma unroll 1
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int i = 0; i < (NM > 0 ? NM : 1); ++i) {
                if (NM > 0) mfma_slot(i);
                __builtin_amdgcn_sched_barrier(0);
                constexpr int PER = NM > 0 ? NV / NM : NV;
#pragma unroll
                for (int j = 0; j < PER; ++j) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[(i * PER + j) % 16]) : "v"(inc));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else if (do_m) {      // the role is decided OUTSIDE the loop (round 3's version branched twice per MFMA slot: 52 cycles per slot were the branches)
#pragma unroll 1
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                mfma_slot(i);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
#pragma unroll 1
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int j = 0; j < NV; ++j) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[j % 16]) : "v"(inc));
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 2; ++i) s += acc32[i][0] + acc32[i][15];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    if (s == 12345.678f) sink[0] = s + lds[threadIdx.x];
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int KIND, int NM, int NV, int MODE>
void run(int threads, const char* label, float* sink, long long* dcyc) {
    auto k = bench<KIND, NM, NV, MODE>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    const int blocks = 256;
    hipMemset(dcyc, 0, blocks * 8 * sizeof(long long));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 96 * 1024, 0, sink, dcyc);   // warm-up
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 96 * 1024, 0, sink, dcyc);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks * 8);
    hipMemcpy(h.data(), dcyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    std::vector<double> mw, vw;
    const int waves = threads / 64;
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < waves; ++w) ((MODE >= 1 && w >= 4) ? vw : mw).push_back((double)h[b * 8 + w] / ITER);
    std::sort(mw.begin(), mw.end());
    std::sort(vw.begin(), vw.end());
    static const char* kn[] = {"f32 16x16x4", "f16 16x16x32", "f16 16x16x16", "f16 32x32x16"};
    printf("%-30s %-13s NM=%2d NV=%3d thr=%3d  cyc/iter %8.1f", label, kn[KIND], NM, NV, threads, mw[mw.size() / 2]);
    if (NM > 0) printf("  = %6.2f cyc per MFMA slot", mw[mw.size() / 2] / NM);
    if (!vw.empty()) printf("   vector waves: %8.1f cyc/iter", vw[vw.size() / 2]);
    printf("\n");
}

// ---- numerics: C[16][16] = A[16][32] . B[32][16] by split f16, one wave
__device__ inline void split2(float x0, float x1, f16x2& hi, f16x2& lo) {
    typedef __fp16 h2 __attribute__((ext_vector_type(2)));
    const h2 t = __builtin_amdgcn_cvt_pkrtz(x0, x1);
    hi = __builtin_bit_cast(f16x2, t);
    lo = f16x2{(_Float16)(x0 - (float)hi[0]), (_Float16)(x1 - (float)hi[1])};
}

__global__ void numerics(const float* A, const float* B, float* C3, float* C1) {
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    f16x8 ah, al, bh, bl;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        f16x2 h, lo;
        split2(A[r * 32 + 8 * g + j], A[r * 32 + 8 * g + j + 1], h, lo);
        ah[j] = h[0]; ah[j + 1] = h[1]; al[j] = lo[0]; al[j + 1] = lo[1];
        split2(B[(8 * g + j) * 16 + r], B[(8 * g + j + 1) * 16 + r], h, lo);
        bh[j] = h[0]; bh[j + 1] = h[1]; bl[j] = lo[0]; bl[j + 1] = lo[1];
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, c, 0, 0, 0);
    f32x4 c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    c += c1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        C3[(4 * g + e) * 16 + r] = c[e];     // D: row 4 * (lane / 16) + e, column lane % 16
        C1[(4 * g + e) * 16 + r] = c1[e];
    }
}

static float rtz_f16(float x) {   // f32 -> f16 toward zero -> f32, host model
    _Float16 h = (_Float16)x;
    float hf = (float)h;
    if (std::fabs(hf) > std::fabs(x)) {
        unsigned short u;
        memcpy(&u, &h, 2);
        u -= 1;
        memcpy(&h, &u, 2);
        hf = (float)h;
    }
    return hf;
}

static void numerics_test(double scaleA, double scaleB, bool flush_model_too) {
    std::vector<float> A(16 * 32), B(32 * 16), C3(256), C1(256);
    srand(7);
    for (auto& x : A) x = (float)(scaleA * (2.0 * rand() / RAND_MAX - 1.0));
    for (auto& x : B) x = (float)(scaleB * (2.0 * rand() / RAND_MAX - 1.0));
    float *dA, *dB, *dC3, *dC1;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC3, 1024); hipMalloc(&dC1, 1024);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(numerics, dim3(1), dim3(64), 0, 0, dA, dB, dC3, dC1);
    hipMemcpy(C3.data(), dC3, 1024, hipMemcpyDeviceToHost);
    hipMemcpy(C1.data(), dC1, 1024, hipMemcpyDeviceToHost);
    double e_exact = 0, e_keep = 0, e_flush = 0, e_f32 = 0, mag = 0;
    for (int i = 0; i < 16; ++i)
        for (int n = 0; n < 16; ++n) {
            double ex = 0, keep = 0, fl = 0;
            float f32 = 0.f;
            for (int k = 0; k < 32; ++k) {
                const float a = A[i * 32 + k], b = B[k * 16 + n];
                ex += (double)a * b;
                f32 = fmaf(a, b, f32);
                const float ah = rtz_f16(a), bh = rtz_f16(b);
                const float al = (float)(_Float16)(a - ah), bl = (float)(_Float16)(b - bh);
                keep += (double)ah * bh + (double)ah * bl + (double)al * bh;
                const float alf = std::fabs(al) < 6.103515625e-05f ? 0.f : al, blf = std::fabs(bl) < 6.103515625e-05f ? 0.f : bl;
                fl += (double)ah * bh + (double)ah * blf + (double)alf * bh;
            }
            e_exact = std::max(e_exact, std::fabs(C3[i * 16 + n] - ex));
            e_keep = std::max(e_keep, std::fabs(C3[i * 16 + n] - keep));
            e_flush = std::max(e_flush, std::fabs(C3[i * 16 + n] - fl));
            e_f32 = std::max(e_f32, std::fabs((double)f32 - ex));
            mag = std::max(mag, std::fabs(ex));
        }
    printf("numerics |A|<=%g |B|<=%g: max|C| %.3e  GPU 3-product vs exact %.3e (f32 fmaf chain vs exact %.3e)  vs model keeping subnormals %.3e  vs model flushing them %.3e\n",
           scaleA, scaleB, mag, e_exact, e_f32, e_keep, e_flush);
    (void)flush_model_too;
    hipFree(dA); hipFree(dB); hipFree(dC3); hipFree(dC1);
}

int main() {
    float* sink;
    long long* dcyc;
    hipMalloc(&sink, 64);
    hipMalloc(&dcyc, 256 * 8 * sizeof(long long));
    numerics_test(1.0, 1.0, true);
    numerics_test(0.05, 0.05, true);     // lo parts subnormal in f16
    numerics_test(8.0, 0.01, true);
    numerics_test(1e-3, 1.0, true);
    // 1. back-to-back rate, one wave per SIMD
    run<0, 8, 0, 0>(256, "mfma only, 1 wave/SIMD", sink, dcyc);
    run<1, 8, 0, 0>(256, "mfma only, 1 wave/SIMD", sink, dcyc);
    run<2, 8, 0, 0>(256, "mfma only, 1 wave/SIMD", sink, dcyc);
    run<3, 8, 0, 0>(256, "mfma only, 1 wave/SIMD", sink, dcyc);
    run<1, 8, 0, 0>(512, "mfma only, 2 waves/SIMD", sink, dcyc);
    // 2. vector instructions of the same wave between MFMAs
#define SAME(K)                                                   \
    run<K, 8, 8, 0>(256, "same wave, 1 valu per mfma", sink, dcyc);  \
    run<K, 8, 16, 0>(256, "same wave, 2 valu per mfma", sink, dcyc); \
    run<K, 8, 24, 0>(256, "same wave, 3 valu per mfma", sink, dcyc); \
    run<K, 8, 32, 0>(256, "same wave, 4 valu per mfma", sink, dcyc); \
    run<K, 8, 48, 0>(256, "same wave, 6 valu per mfma", sink, dcyc); \
    run<K, 8, 64, 0>(256, "same wave, 8 valu per mfma", sink, dcyc);
    SAME(1) SAME(2) SAME(3) SAME(0)
    // two waves per SIMD, each doing both
    run<1, 8, 16, 0>(512, "2 waves/SIMD both, 2 valu/mfma", sink, dcyc);
    run<1, 8, 32, 0>(512, "2 waves/SIMD both, 4 valu/mfma", sink, dcyc);
    run<1, 8, 64, 0>(512, "2 waves/SIMD both, 8 valu/mfma", sink, dcyc);
    // split roles: waves 0-3 MFMA only, waves 4-7 vector only (NV per iteration)
    run<1, 8, 16, 1>(512, "split roles", sink, dcyc);
    run<1, 8, 32, 1>(512, "split roles", sink, dcyc);
    run<1, 8, 64, 1>(512, "split roles", sink, dcyc);
    run<1, 8, 16, 2>(512, "split roles, prio on MFMA waves", sink, dcyc);
    run<1, 8, 32, 2>(512, "split roles, prio on MFMA waves", sink, dcyc);
    run<1, 8, 64, 2>(512, "split roles, prio on MFMA waves", sink, dcyc);
    run<1, 8, 32, 3>(512, "split roles, prio on vector waves", sink, dcyc);
    run<0, 8, 32, 1>(512, "split roles", sink, dcyc);
    run<0, 8, 64, 1>(512, "split roles", sink, dcyc);
    run<1, 0, 64, 0>(256, "valu only, 1 wave/SIMD", sink, dcyc);
    run<1, 0, 64, 0>(512, "valu only, 2 waves/SIMD", sink, dcyc);
    return 0;
}
