// Microbenchmark (gfx950) for the NEXT organisation of the fused ConvBlock (LABBOOK.md 4.3e, "what a wave costs"): the MLP of a tile
// (per 16-pixel group 60 + 54 split-f16 MFMAs with a GELU + split of 192 hidden values between them) run
//   MODE 0  as convblock_pipe_kernel's back waves run it today: one wave per SIMD does fc1, GELU + split, fc2 for two pixel groups at a time;
//   MODE 1  by PURE roles: per SIMD one wave that issues nothing but MFMAs (and the LDS traffic around them) and one wave that does nothing
//           but GELU + split, the hidden activations handed over in 4-KiB slices through a two-slot LDS ring with flag words
// -- each with and without a third ("front") wave per SIMD that issues the depth-wise taps' stream (packed FMAs fed by 16-byte LDS reads).
// Timing only: operands are synthetic, nothing is stored.  Same instruction sequences as the kernel (its gelu_phi4_scaled, split4h, fragment reads).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -w -Irvdd-release_amd/csrc tools/mlp_roles_bench.hip -o tools/scratch/mlprb && tools/scratch/mlprb
#include "../rvdd-release_amd/csrc/convnext.hip"

#include <algorithm>
#include <cstdio>
#include <vector>

namespace {

struct GC {
    float c[7][2];
};
constexpr int RB_W_BYTES = F_W1H_BYTES + F_W2H_BYTES;        // 72 KiB of fragments (synthetic)
constexpr int RB_SLOTS = 3;                                  // the MFMA wave runs two steps ahead of what it consumes
constexpr int RB_RING_BYTES = 4 * RB_SLOTS * 4096;           // four pairs x three slots of 4 KiB (the GELU wave writes its result over the slice it read)
constexpr int RB_LDS = RB_W_BYTES + 2048 + RB_RING_BYTES + 256;

// one pair of hidden blocks for two pixel groups: the 20 fc1 MFMAs
#define RB_FC1()                                                                                                                              \
    _Pragma("unroll") for (int q = 0; q < 2; ++q) _Pragma("unroll") for (int k = 0; k < 2; ++k)                                                \
        hq[q][k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[k], B2[q], hq[q][k], 0, 0, 0);                                                   \
    _Pragma("unroll") for (int q = 0; q < 2; ++q) _Pragma("unroll") for (int k = 0; k < 2; ++k)                                                \
        hq[q][k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[k], B1[q], hq[q][k], 0, 0, 0);                                                   \
    _Pragma("unroll") for (int q = 0; q < 2; ++q) _Pragma("unroll") for (int k = 0; k < 2; ++k)                                                \
        hq[q][k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fc[k], B4[q], hq[q][k], 0, 0, 0);                                                   \
    _Pragma("unroll") for (int q = 0; q < 2; ++q) _Pragma("unroll") for (int k = 0; k < 2; ++k)                                                \
        hq[q][k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fc[k], B3[q], hq[q][k], 0, 0, 0);                                                   \
    _Pragma("unroll") for (int q = 0; q < 2; ++q) _Pragma("unroll") for (int k = 0; k < 2; ++k)                                                \
        hq[q][k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[k], B1[q], hq[q][k], 0, 0, 0)
// ... and the 18 fc2 MFMAs
#define RB_FC2()                                                                                                                              \
    _Pragma("unroll") for (int q = 0; q < 2; ++q) _Pragma("unroll") for (int mo = 0; mo < 3; ++mo)                                             \
        a2[q][mo] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gh[mo], Bhl[q], a2[q][mo], 0, 0, 0);                                               \
    _Pragma("unroll") for (int q = 0; q < 2; ++q) _Pragma("unroll") for (int mo = 0; mo < 3; ++mo)                                             \
        a2[q][mo] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gl[mo], Bhh[q], a2[q][mo], 0, 0, 0);                                               \
    _Pragma("unroll") for (int q = 0; q < 2; ++q) _Pragma("unroll") for (int mo = 0; mo < 3; ++mo)                                             \
        a2[q][mo] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gh[mo], Bhh[q], a2[q][mo], 0, 0, 0)

template <int MODE, int FRONT>
__global__ __launch_bounds__(256 * (1 + MODE + FRONT)) void mlp_roles(float* sink, long long* cycles, int ntiles, GC gc) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NW = 4 * (1 + MODE + FRONT);
    // synthetic fragments and biases: small f16 values, a few per cent of them zero
    for (int i = tid; i < (RB_W_BYTES + 2048) / 4; i += 64 * NW) {
        const _Float16 a = (_Float16)(0.01f * ((i * 7) % 23 - 11)), b = (_Float16)(0.003f * ((i * 13) % 17 - 8));
        h2v v = {a, b};
        smem[i] = __builtin_bit_cast(float, v);
    }
    typedef __attribute__((address_space(3))) unsigned lds_u32;
    unsigned* flags = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(smem) + RB_W_BYTES + 2048 + RB_RING_BYTES);
    if (tid < 64) flags[tid] = 0;      // (RB_LDS leaves 256 bytes for them)
    __syncthreads();
    const char* W1 = reinterpret_cast<const char*>(smem);
    const char* W2 = W1 + F_W1H_BYTES;
    typedef __attribute__((address_space(3))) f32x4 lds_f4;
    lds_f4* bvp = (lds_f4*)(smem + RB_W_BYTES / 4) + (lane >> 4);
    // MODE 2: TWO GELU waves per SIMD, one per pixel group of the slice
    const int role = MODE == 0 ? (wave < 4 ? 0 : 3) : (wave < 4 ? 1 : (wave < 4 + 4 * MODE ? 2 : 3));      // 0 mixed, 1 MFMA, 2 GELU, 3 front
    const int gsub = MODE == 2 ? (wave - 4) >> 2 : 0;
    const int pair = wave & 3;
    char* ring = reinterpret_cast<char*>(smem) + RB_W_BYTES + 2048 + pair * RB_SLOTS * 4096;
    unsigned* fA = flags + pair * 8;        // [slot]
    unsigned* fB = flags + pair * 8 + 4;      // MODE 2: the second GELU wave's at + 32
    const char* w1b = W1 + lane * 16;
    const char* w2b = W2 + lane * 16;
    auto FA = [&](int m, int f) { return __builtin_bit_cast(h8v, *reinterpret_cast<const f32x4*>(w1b + m * 3072 + f * 1024)); };
    auto FG = [&](int p, int mo, int hl) { return __builtin_bit_cast(h8v, *reinterpret_cast<const f32x4*>(w2b + ((p * 3 + mo) * 2 + hl) * 1024)); };
    bool gave_up = false;
    auto wait_flag = [&](unsigned* f, unsigned want) {      // (with a spin limit: a protocol error ends the launch instead of hanging the GPU)
        int spins = 0;
        while (!gave_up && (int)(__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) - want) < 0) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 20)) gave_up = true;
        }
    };
    auto post_flag = [&](unsigned* f, unsigned v) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(f, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    float keep = 0.f;
    const long long t0 = __builtin_amdgcn_s_memtime();

    if (role == 0 || role == 1) {
        // B operands of the two pixel groups (the LayerNorm output, split): synthetic
        h8v B1[2], B2[2], B3[2], B4[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            f32x4 x0 = {0.3f + 0.01f * lane, -0.7f + 0.02f * q, 1.1f, -0.2f * lane}, x1 = {0.9f, 0.05f * lane, -1.3f, 0.4f}, x2 = {-0.6f, 0.8f, 0.02f * lane, 1.7f};
            u32x2v xh[3], xl[3];
            split4h(x0, xh[0], xl[0]);
            split4h(x1, xh[1], xl[1]);
            split4h(x2, xh[2], xl[2]);
            B1[q] = cat8(xh[0], xh[1]);
            B2[q] = cat8(xl[0], xl[1]);
            B3[q] = cat8(xh[2], xh[2]);
            B4[q] = cat8(xl[2], u32x2v{0u, 0u});
        }
        unsigned seq = 0;
#pragma unroll 1
        for (int t = 0; t < ntiles; ++t) {
#pragma unroll 1
            for (int n2 = 0; n2 < 2; ++n2) {
                f32x4 a2[2][3];
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int mo = 0; mo < 3; ++mo) a2[q][mo] = bvp[48 + 4 * mo];
                if (role == 0) {
#pragma unroll
                    for (int p = 0; p < 6; ++p) {
                        h8v fa[2], fb[2], fc[2], gh[3], gl[3];
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            fa[k] = FA(2 * p + k, 0);
                            fb[k] = FA(2 * p + k, 1);
                            fc[k] = FA(2 * p + k, 2);
                        }
#pragma unroll
                        for (int mo = 0; mo < 3; ++mo) {
                            gh[mo] = FG(p, mo, 0);
                            gl[mo] = FG(p, mo, 1);
                        }
                        f32x4 hq[2][2];
#pragma unroll
                        for (int q = 0; q < 2; ++q)
#pragma unroll
                            for (int k = 0; k < 2; ++k) hq[q][k] = bvp[4 * (2 * p + k)];
                        __builtin_amdgcn_sched_barrier(0);
                        RB_FC1();
                        h8v Bhh[2], Bhl[2];
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            u32x2v hh[2], hl[2];
#pragma unroll
                            for (int k = 0; k < 2; ++k) split4h(gelu_phi4_scaled(hq[q][k], gc.c), hh[k], hl[k]);
                            Bhh[q] = cat8(hh[0], hh[1]);
                            Bhl[q] = cat8(hl[0], hl[1]);
                        }
                        RB_FC2();
                    }
                } else {
                    // pure MFMA role, two steps ahead of what it consumes: fc1 of step s goes out, then fc2 of step s - 2 on what the GELU wave
                    // has returned (the pipeline runs on across pixel-group pairs and tiles; timing only, the accumulators just keep adding)
#pragma unroll
                    for (int p = 0; p < 6; ++p) {
                        {
                            h8v fa[2], fb[2], fc[2];
#pragma unroll
                            for (int k = 0; k < 2; ++k) {
                                fa[k] = FA(2 * p + k, 0);
                                fb[k] = FA(2 * p + k, 1);
                                fc[k] = FA(2 * p + k, 2);
                            }
                            f32x4 hq[2][2];
#pragma unroll
                            for (int q = 0; q < 2; ++q)
#pragma unroll
                                for (int k = 0; k < 2; ++k) hq[q][k] = bvp[4 * (2 * p + k)];
                            __builtin_amdgcn_sched_barrier(0);
                            RB_FC1();
                            const unsigned s = seq + p;
                            char* in = ring + (s % RB_SLOTS) * 4096 + lane * 16;
#pragma unroll
                            for (int q = 0; q < 2; ++q)
#pragma unroll
                                for (int k = 0; k < 2; ++k) *reinterpret_cast<f32x4*>(in + (q * 2 + k) * 1024) = hq[q][k];
                            post_flag(fA + s % RB_SLOTS, s + 1);
                        }
                        if (seq + p >= 2) {
                            const unsigned s = seq + p - 2;
                            const int pp = (p + 4) % 6;
                            h8v gh[3], gl[3];
#pragma unroll
                            for (int mo = 0; mo < 3; ++mo) {
                                gh[mo] = FG(pp, mo, 0);
                                gl[mo] = FG(pp, mo, 1);
                            }
                            wait_flag(fB + s % RB_SLOTS, s + 1);
                            if (MODE == 2) wait_flag(fB + 32 + s % RB_SLOTS, s + 1);
                            const char* out = ring + (s % RB_SLOTS) * 4096 + lane * 16;
                            h8v Bhh[2], Bhl[2];
#pragma unroll
                            for (int q = 0; q < 2; ++q) {
                                Bhh[q] = __builtin_bit_cast(h8v, *reinterpret_cast<const f32x4*>(out + (q * 2) * 1024));
                                Bhl[q] = __builtin_bit_cast(h8v, *reinterpret_cast<const f32x4*>(out + (q * 2 + 1) * 1024));
                            }
                            RB_FC2();
                        }
                    }
                    seq += 6;
                }
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int mo = 0; mo < 3; ++mo) keep += a2[q][mo][0] + a2[q][mo][3];
            }
        }
    } else if (role == 2) {
        unsigned seq = 0;
#pragma unroll 1
        for (int t = 0; t < ntiles * 12; ++t) {
            const unsigned s = seq++;
            wait_flag(fA + s % RB_SLOTS, s + 1);
            const char* in = ring + (s % RB_SLOTS) * 4096 + lane * 16;
            char* out = ring + (s % RB_SLOTS) * 4096 + lane * 16;
            if (MODE == 2) {
                f32x4 h2q[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) h2q[k] = *reinterpret_cast<const f32x4*>(in + (gsub * 2 + k) * 1024);
                u32x2v hh[2], hl[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) split4h(gelu_phi4_scaled(h2q[k], gc.c), hh[k], hl[k]);
                *reinterpret_cast<h8v*>(out + (gsub * 2) * 1024) = cat8(hh[0], hh[1]);
                *reinterpret_cast<h8v*>(out + (gsub * 2 + 1) * 1024) = cat8(hl[0], hl[1]);
                post_flag(fB + 32 * gsub + s % RB_SLOTS, s + 1);
                continue;
            }
            f32x4 hq[2][2];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int k = 0; k < 2; ++k) hq[q][k] = *reinterpret_cast<const f32x4*>(in + (q * 2 + k) * 1024);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                u32x2v hh[2], hl[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) split4h(gelu_phi4_scaled(hq[q][k], gc.c), hh[k], hl[k]);
                *reinterpret_cast<h8v*>(out + (q * 2) * 1024) = cat8(hh[0], hh[1]);
                *reinterpret_cast<h8v*>(out + (q * 2 + 1) * 1024) = cat8(hl[0], hl[1]);
            }
            post_flag(fB + s % RB_SLOTS, s + 1);
        }
    } else {
        // the front waves' stream of a tile: three 16-channel chunks x seven filter rows x (10 + 7 sixteen-byte LDS reads, 56 packed FMAs)
        const float* tb = smem + (lane * 4) % 4096;
        f32x4 acc[4][3];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int t = 0; t < ntiles; ++t) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                f32x4 win[2][10], wv[2][7];
                auto read_row = [&](int ky, f32x4 (&wn)[10], f32x4 (&ww)[7]) {
#pragma unroll
                    for (int dx = 0; dx < 10; ++dx) wn[dx] = *reinterpret_cast<const f32x4*>(tb + (j * 7 + ky) * 256 + dx * 16);
#pragma unroll
                    for (int kx = 0; kx < 7; ++kx) ww[kx] = *reinterpret_cast<const f32x4*>(tb + 8192 + (j * 49 + ky * 7 + kx) * 16);
                };
                read_row(0, win[0], wv[0]);
#pragma unroll
                for (int ky = 0; ky < 7; ++ky) {
                    if (ky + 1 < 7) read_row(ky + 1, win[(ky + 1) & 1], wv[(ky + 1) & 1]);
#pragma unroll
                    for (int kx = 0; kx < 7; ++kx)
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[i][j] = acc[i][j] + win[ky & 1][i + kx] * wv[ky & 1][kx];
                    asm volatile("" : "+v"(acc[0][j]), "+v"(acc[1][j]), "+v"(acc[2][j]), "+v"(acc[3][j])::"memory");
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) keep += acc[i][j][0] + acc[i][j][2];
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (keep == 12345.678f) sink[0] = keep;
    if (lane == 0) cycles[blockIdx.x * 16 + wave] = gave_up ? -1 : t1 - t0;
}

template <int MODE, int FRONT>
void run(const char* label, float* sink, long long* dcyc, GC gc) {
    auto k = mlp_roles<MODE, FRONT>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, RB_LDS);
    const int blocks = 256, ntiles = 200, threads = 256 * (1 + MODE + FRONT);
    (void)hipMemset(dcyc, 0, blocks * 16 * sizeof(long long));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), RB_LDS, 0, sink, dcyc, 20, gc);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), RB_LDS, 0, sink, dcyc, ntiles, gc);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks * 16);
    (void)hipMemcpy(h.data(), dcyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    const int nw = threads / 64;
    printf("%-44s %7.1f us per tile and CU (%d waves)   cycles (100 MHz ticks x clock) per tile, median over CUs:", label, 1e3 * ms / ntiles, nw);
    const char* names[4] = {"mixed", "MFMA", "GELU", "front"};
    for (int r = 0; r < 4; ++r) {
        std::vector<double> v;
        for (int b = 0; b < blocks; ++b)
            for (int w = 0; w < nw; ++w) {
                const int role = MODE == 0 ? (w < 4 ? 0 : 3) : (w < 4 ? 1 : (w < 4 + 4 * MODE ? 2 : 3));
                if (role == r) v.push_back((double)h[b * 16 + w] / ntiles);
            }
        if (v.empty()) continue;
        std::sort(v.begin(), v.end());
        printf("  %s %.0f", names[r], v[v.size() / 2]);
    }
    printf("\n");
}

}      // namespace

int main() {
    float* sink;
    long long* dcyc;
    (void)hipMalloc(&sink, 64);
    (void)hipMalloc(&dcyc, 256 * 16 * sizeof(long long));
    GC gc;
    const double C[6] = {2.992418740177527e-05, -0.0007398742018267512, 0.007977462373673916, -0.05323818698525429, -0.45891568064689636, -1.1511471271514893};
    for (int i = 0; i < 6; ++i) gc.c[i][0] = gc.c[i][1] = (float)C[i];
    gc.c[6][0] = gc.c[6][1] = 6.36f;
    run<0, 0>("mixed wave per SIMD (today's back wave)", sink, dcyc, gc);
    run<1, 0>("MFMA wave + GELU wave per SIMD", sink, dcyc, gc);
    run<0, 1>("mixed wave + front wave per SIMD (today)", sink, dcyc, gc);
    run<1, 1>("MFMA wave + GELU wave + front wave per SIMD", sink, dcyc, gc);
    run<2, 0>("MFMA wave + 2 GELU waves per SIMD", sink, dcyc, gc);
    run<2, 1>("MFMA wave + 2 GELU waves + front wave per SIMD", sink, dcyc, gc);
    return 0;
}
