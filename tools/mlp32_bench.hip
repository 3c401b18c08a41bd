// Microbenchmark (gfx950), round 6: the MLP of a ConvBlock tile on v_mfma_f32_32x32x16_f16 instead of 16x16x32.
// tools/f16_mfma_bench.hip measured, for ONE wave's in-order stream, 8 vector instructions + one 32x32x16 at 45 cycles against 8 vector
// instructions + two 16x16x32 (the same matrix work) at 2 x 39.8: the 32-cycle MFMA covers ~6 issue slots, the 16-cycle one ~2.  The back
// waves of convblock_pipe_kernel carry 4 vector instructions per 16x16x32 (GELU + split), so the shape of the MFMA decides how much of the GELU hides.
//   MODE 0  today's back wave (two 16-pixel groups at a time, 114 MFMAs 16x16x32 per group), as tools/mlp_roles_bench.hip MODE 0
//   MODE 1  32x32x16, D[hidden 32][pixel 32]: fc1 = 6 row blocks x 3 k-steps x 3 products = 54, fc2 = 12 k-steps x 2 row blocks (48 -> 64, the last 16 rows zero) x 3 = 72
//           per 32 pixels; straight order per row block (fc1, GELU + split, fc2), the compiler schedules
//   MODE 2  as 1, software-pipelined by hand: fc1 of row block m + 1 is issued in front of the GELU of row block m, the GELU's instructions dealt ~6 per MFMA
// each alone and beside a "front-like" wave (the depth-wise taps' stream).  Timing only: synthetic operands, nothing stored.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -w -Irvdd-release_amd/csrc tools/mlp32_bench.hip -o tools/scratch/mlp32b && tools/scratch/mlp32b
#include "../rvdd-release_amd/csrc/convnext.hip"

#include <algorithm>
#include <cstdio>
#include <vector>

namespace {

struct GC {
    float c[7][2];
};
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int B_W1 = 36 * 1024, B_W2 = 48 * 1024;            // fc1 36 KiB either way; fc2 48 KiB uncompacted in MODE 1/2 (36 in MODE 0)
constexpr int B_LDS = B_W1 + B_W2 + 2048 + 8192 + 16384;     // + biases + a window for the front-like wave

template <int MODE, int FRONT>
__global__ __launch_bounds__(256 * (1 + FRONT), 2) void mlp32(float* sink, long long* cycles, int ntiles, GC gc) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NW = 4 * (1 + FRONT);
    for (int i = tid; i < B_LDS / 4; i += 64 * NW) {
        const _Float16 a = (_Float16)(0.01f * ((i * 7) % 23 - 11)), b = (_Float16)(0.003f * ((i * 13) % 17 - 8));
        h2v v = {a, b};
        smem[i] = __builtin_bit_cast(float, v);
    }
    __syncthreads();
    const char* W1 = reinterpret_cast<const char*>(smem);
    const char* W2 = W1 + B_W1;
    typedef __attribute__((address_space(3))) f32x4 lds_f4;
    lds_f4* bvp = (lds_f4*)(smem + (B_W1 + B_W2) / 4) + (lane >> 4);
    const char* w1b = W1 + lane * 16;
    const char* w2b = W2 + lane * 16;
    float keep = 0.f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4 && MODE == 0) {
        auto FA = [&](int m, int f) { return __builtin_bit_cast(h8v, *reinterpret_cast<const f32x4*>(w1b + m * 3072 + f * 1024)); };
        auto FG = [&](int p, int mo, int hl) { return __builtin_bit_cast(h8v, *reinterpret_cast<const f32x4*>(w2b + ((p * 3 + mo) * 2 + hl) * 1024)); };
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
#pragma unroll 1
        for (int t = 0; t < ntiles; ++t) {
#pragma unroll 1
            for (int n2 = 0; n2 < 2; ++n2) {
                f32x4 a2[2][3];
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int mo = 0; mo < 3; ++mo) a2[q][mo] = bvp[48 + 4 * mo];
                h8v fa[2][2], fb[2][2], fc[2][2];
                auto load_fc1 = [&](int p, int buf) {
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        fa[buf][k] = FA(2 * p + k, 0);
                        fb[buf][k] = FA(2 * p + k, 1);
                        fc[buf][k] = FA(2 * p + k, 2);
                    }
                };
                load_fc1(0, 0);
#pragma unroll
                for (int p = 0; p < 6; ++p) {      // the shipping kernel's order (convnext.hip back waves)
                    const int cb = p & 1;
                    f32x4 hq[2][2];
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int k = 0; k < 2; ++k) hq[q][k] = bvp[4 * (2 * p + k)];
                    if (p + 1 < 6) load_fc1(p + 1, cb ^ 1);
                    h8v gh[3], gl[3];
#pragma unroll
                    for (int mo = 0; mo < 3; ++mo) {
                        gh[mo] = FG(p, mo, 0);
                        gl[mo] = FG(p, mo, 1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    auto fc1_mfma = [&](int q, int i) {
                        const int k = i & 1, tt = i >> 1;
                        const h8v A = (tt == 0 || tt == 4) ? fa[cb][k] : tt == 1 ? fb[cb][k] : fc[cb][k];
                        const h8v Bv = tt == 0 ? B2[q] : (tt == 1 || tt == 4) ? B1[q] : tt == 2 ? B4[q] : B3[q];
                        hq[q][k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, Bv, hq[q][k], 0, 0, 0);
                    };
                    h8v Bhh[2], Bhl[2];
                    auto fc2_mfma = [&](int q, int i) {
                        const int mo = i % 3, tt = i / 3;
                        a2[q][mo] = __builtin_amdgcn_mfma_f32_16x16x32_f16(tt == 1 ? gl[mo] : gh[mo], tt == 0 ? Bhl[q] : Bhh[q], a2[q][mo], 0, 0, 0);
                    };
#pragma unroll
                    for (int i = 0; i < 10; ++i) fc1_mfma(0, i);
                    __builtin_amdgcn_sched_barrier(0);
                    GeluStages gs[2];
                    u32x2v hh[2], hl[2];
#define GS_(q, k, S) gelu_stage<S>(gs[k], hq[q][k], gc.c, hh[k], hl[k])
#define GELU_ALL_(q) GS_(q, 0, 0); GS_(q, 1, 0); GS_(q, 0, 1); GS_(q, 1, 1); GS_(q, 0, 2); GS_(q, 1, 2); GS_(q, 0, 3); GS_(q, 1, 3); \
                     GS_(q, 0, 4); GS_(q, 1, 4); GS_(q, 0, 5); GS_(q, 1, 5)
#pragma unroll
                    for (int i = 0; i < 10; ++i) fc1_mfma(1, i);
                    GELU_ALL_(0);
#pragma unroll
                    for (int i = 0; i < 10; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x402, 64, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    Bhh[0] = cat8(hh[0], hh[1]);
                    Bhl[0] = cat8(hl[0], hl[1]);
#pragma unroll
                    for (int i = 0; i < 9; ++i) fc2_mfma(0, i);
                    GELU_ALL_(1);
#pragma unroll
                    for (int i = 0; i < 9; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
                        __builtin_amdgcn_sched_group_barrier(0x002, 2, 1);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x402, 64, 1);
                    __builtin_amdgcn_sched_barrier(0);
#undef GELU_ALL_
#undef GS_
                    Bhh[1] = cat8(hh[0], hh[1]);
                    Bhl[1] = cat8(hl[0], hl[1]);
#pragma unroll
                    for (int i = 0; i < 9; ++i) fc2_mfma(1, i);
                }
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int mo = 0; mo < 3; ++mo) keep += a2[q][mo][0] + a2[q][mo][3];
            }
        }
    } else if (wave < 4) {
        // ---- 32x32x16: a wave's 64 pixels are two column blocks of 32; lane = (pixel n = lane & 31, half h = lane >> 5)
        auto F1 = [&](int mb, int s, int hl) { return __builtin_bit_cast(h8v, *reinterpret_cast<const f32x4*>(w1b + ((mb * 3 + s) * 2 + hl) * 1024)); };
        auto F2 = [&](int ks, int ob, int hl) { return __builtin_bit_cast(h8v, *reinterpret_cast<const f32x4*>(w2b + ((ks * 2 + ob) * 2 + hl) * 1024)); };
        h8v Xh[2][3], Xl[2][3];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                f32x4 x0 = {0.3f + 0.01f * lane, -0.7f + 0.02f * q, 1.1f + s, -0.2f * lane}, x1 = {0.9f, 0.05f * lane, -1.3f, 0.4f * s};
                u32x2v h0, l0, h1, l1;
                split4h(x0, h0, l0);
                split4h(x1, h1, l1);
                Xh[q][s] = cat8(h0, h1);
                Xl[q][s] = cat8(l0, l1);
            }
        auto bias16 = [&](int mb) {
            f32x16 H;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 b = bvp[(mb * 4 + j) % 60];
#pragma unroll
                for (int i = 0; i < 4; ++i) H[4 * j + i] = b[i];
            }
            return H;
        };
        auto gelu16 = [&](const f32x16& H, h8v (&Bhh)[2], h8v (&Bhl)[2]) {
            u32x2v hh[4], hl[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                split4h(gelu_phi4_scaled(f32x4{H[4 * j], H[4 * j + 1], H[4 * j + 2], H[4 * j + 3]}, gc.c), hh[j], hl[j]);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                Bhh[t] = cat8(hh[2 * t], hh[2 * t + 1]);
                Bhl[t] = cat8(hl[2 * t], hl[2 * t + 1]);
            }
        };
#pragma unroll 1
        for (int t = 0; t < ntiles; ++t) {
#pragma unroll 1
            for (int q = 0; q < 2; ++q) {
                f32x16 O[2];
                O[0] = bias16(12);
                O[1] = bias16(13);
                auto fc1 = [&](int mb, f32x16 H) {
                    h8v ah[3], al[3];
#pragma unroll
                    for (int s = 0; s < 3; ++s) {
                        ah[s] = F1(mb, s, 0);
                        al[s] = F1(mb, s, 1);
                    }
#pragma unroll
                    for (int s = 0; s < 3; ++s) H = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], Xl[q][s], H, 0, 0, 0);
#pragma unroll
                    for (int s = 0; s < 3; ++s) H = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s], Xh[q][s], H, 0, 0, 0);
#pragma unroll
                    for (int s = 0; s < 3; ++s) H = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], Xh[q][s], H, 0, 0, 0);
                    return H;
                };
                auto fc2 = [&](int mb, const h8v (&Bhh)[2], const h8v (&Bhl)[2]) {
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {
                        h8v gh[2], gl[2];
#pragma unroll
                        for (int ob = 0; ob < 2; ++ob) {
                            gh[ob] = F2(2 * mb + tt, ob, 0);
                            gl[ob] = F2(2 * mb + tt, ob, 1);
                        }
#pragma unroll
                        for (int ob = 0; ob < 2; ++ob) O[ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh[ob], Bhl[tt], O[ob], 0, 0, 0);
#pragma unroll
                        for (int ob = 0; ob < 2; ++ob) O[ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gl[ob], Bhh[tt], O[ob], 0, 0, 0);
#pragma unroll
                        for (int ob = 0; ob < 2; ++ob) O[ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh[ob], Bhh[tt], O[ob], 0, 0, 0);
                    }
                };
                if (MODE == 1) {
#pragma unroll
                    for (int mb = 0; mb < 6; ++mb) {
                        const f32x16 H = fc1(mb, bias16(mb));
                        h8v Bhh[2], Bhl[2];
                        gelu16(H, Bhh, Bhl);
                        fc2(mb, Bhh, Bhl);
                    }
                } else {
                    // fc1(m + 1) goes out in front of GELU(m); the GELU's ~136 instructions behind the 9 + 12 MFMAs around it, 6 per MFMA
                    f32x16 H = fc1(0, bias16(0));
#pragma unroll
                    for (int mb = 0; mb < 6; ++mb) {
                        f32x16 Hn;
                        __builtin_amdgcn_sched_barrier(0);
                        if (mb + 1 < 6) Hn = fc1(mb + 1, bias16(mb + 1));
                        h8v Bhh[2], Bhl[2];
                        gelu16(H, Bhh, Bhl);
                        if (mb + 1 < 6) {
#pragma unroll
                            for (int i = 0; i < 9; ++i) {
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                                __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                            }
                        }
                        __builtin_amdgcn_sched_group_barrier(0x402, 200, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        fc2(mb, Bhh, Bhl);
                        H = Hn;
                    }
                }
#pragma unroll
                for (int ob = 0; ob < 2; ++ob) keep += O[ob][0] + O[ob][7] + O[ob][15];
            }
        }
    } else {
        // the front waves' stream of a tile (as tools/mlp_roles_bench.hip): 3 chunks x 7 filter rows x (10 + 7 sixteen-byte LDS reads, 56 packed FMAs)
        const float* tb = smem + (B_W1 + B_W2 + 2048) / 4 + (lane * 4) % 2048;
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
                    for (int dx = 0; dx < 10; ++dx) wn[dx] = *reinterpret_cast<const f32x4*>(tb + ((j * 7 + ky) * 64 + dx * 16) % 2048);
#pragma unroll
                    for (int kx = 0; kx < 7; ++kx) ww[kx] = *reinterpret_cast<const f32x4*>(smem + (B_W1 + B_W2) / 4 + ((j * 49 + ky * 7 + kx) * 16 + 4 * (lane & 3)) % 512);
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
    if (lane == 0) cycles[blockIdx.x * 16 + wave] = t1 - t0;
}

template <int MODE, int FRONT>
void run(const char* label, float* sink, long long* dcyc, GC gc) {
    auto k = mlp32<MODE, FRONT>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, B_LDS);
    const int blocks = 256, ntiles = 200, threads = 256 * (1 + FRONT);
    (void)hipMemset(dcyc, 0, blocks * 16 * sizeof(long long));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), B_LDS, 0, sink, dcyc, 20, gc);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), B_LDS, 0, sink, dcyc, ntiles, gc);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks * 16);
    (void)hipMemcpy(h.data(), dcyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    const int nw = threads / 64;
    printf("%-56s %7.2f us per tile and CU (%d waves)   100-MHz ticks per tile, median over CUs:", label, 1e3 * ms / ntiles, nw);
    for (int r = 0; r < 1 + FRONT; ++r) {
        std::vector<double> v;
        for (int b = 0; b < blocks; ++b)
            for (int w = 4 * r; w < 4 * r + 4; ++w) v.push_back((double)h[b * 16 + w] / ntiles);
        std::sort(v.begin(), v.end());
        printf("  %s %.0f", r ? "front" : "mlp", v[v.size() / 2]);
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
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 0>("16x16x32 (today's back wave), alone", sink, dcyc, gc);
        run<1, 0>("32x32x16 straight, alone", sink, dcyc, gc);
        run<2, 0>("32x32x16 pipelined (fc1 m+1 | GELU m | fc2 m), alone", sink, dcyc, gc);
        run<0, 1>("16x16x32 + front-like wave", sink, dcyc, gc);
        run<1, 1>("32x32x16 straight + front-like wave", sink, dcyc, gc);
        run<2, 1>("32x32x16 pipelined + front-like wave", sink, dcyc, gc);
    }
    return 0;
}
