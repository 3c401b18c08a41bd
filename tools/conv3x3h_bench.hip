// Stand-alone timing of conv3x3h_kernel (the split-f16 3x3 conv) on random maps, with the kernel's phase stamps:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DRVDD_STAMPS -DRVDD_CONV_GROUPS2 -Irvdd-release_amd/csrc tools/conv3x3h_bench.hip -o /tmp/c3hb && /tmp/c3hb [B H W]
// Prints microseconds per launch (HIP events), shader cycles per tile and phase of wave 0 (s_memtime), and the clock
// the two imply.
#ifdef CONV_SRC          // an instrumented copy of the kernel (tools/conv3x3h_xp_patch.py)
#include CONV_SRC
#else
#include "../rvdd-release_amd/csrc/conv3x3h.hip"
#endif

#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 4, H = argc > 2 ? atoi(argv[2]) : 720, W = argc > 3 ? atoi(argv[3]) : 1280;
    const int groups = argc > 4 ? atoi(argv[4]) : 2;
    conv3x3h_set_groups(groups);
    const size_t px = (size_t)B * H * W;
    std::vector<float> x(px * 48);
    srand(3);
    for (auto& v : x) v = fmaxf((float)rand() / RAND_MAX * 2.f - 0.9f, 0.f);      // ReLU-like: 45 % zeros (random dense maps draw more power than the nets')
    std::vector<uint16_t> w(conv3x3h_weight_bytes(48) / 2);
    for (auto& v : w) {
        _Float16 hv = (_Float16)((float)rand() / RAND_MAX - 0.5f);
        memcpy(&v, &hv, 2);
    }
#if defined(RVDD_XP) && (RVDD_XP & 128)
    {   // the input as a map stored split: per pixel six blocks of [hi of 8 channels | lo of 8 channels], f16, hi toward zero
        auto rtz = [](float v) {
            _Float16 h = (_Float16)v;
            if (fabsf((float)h) > fabsf(v)) {
                uint16_t b;
                memcpy(&b, &h, 2);
                b -= 1;                       // one step toward zero (same sign, magnitude bits down)
                memcpy(&h, &b, 2);
            }
            return h;
        };
        std::vector<float> y(x.size());
        for (size_t p = 0; p < px; ++p)
            for (int c8 = 0; c8 < 6; ++c8) {
                _Float16 hi[8], lo[8];
                for (int i = 0; i < 8; ++i) {
                    const float v = x[p * 48 + 8 * c8 + i];
                    hi[i] = rtz(v);
                    lo[i] = (_Float16)(v - (float)hi[i]);
                }
                memcpy(&y[p * 48 + 8 * c8], hi, 16);
                memcpy(&y[p * 48 + 8 * c8 + 4], lo, 16);
            }
        x.swap(y);
    }
#endif
    float *din, *dout, *dbias;
    void* dw;
    hipMalloc(&din, px * 192);
    hipMalloc(&dout, px * 192);
    hipMalloc(&dbias, 192);
    hipMalloc(&dw, w.size() * 2);
    hipMemcpy(din, x.data(), px * 192, hipMemcpyHostToDevice);
    hipMemcpy(dw, w.data(), w.size() * 2, hipMemcpyHostToDevice);
    hipMemset(dbias, 0, 192);
    ConvArgs a{};
    a.in = din; a.w = (const float*)dw; a.bias = dbias; a.out = dout;
    a.B = B; a.H = H; a.W = W; a.Hout = H; a.Wout = W; a.wscale = 1.f / 1024;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch_conv3x3h(a, 48, EPI_RELU, 0);
    unsigned long long zero[64] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), zero, sizeof(zero));
    const int iters = 10;
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) launch_conv3x3h(a, 48, EPI_RELU, 0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long st[64];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof(st));
    const char* names[7] = {"loop top", "barrier (tile staged)", "addresses", "MFMA chunks (+ loads, stores, split)", "epilogue math", "barrier (tile consumed)", "LDS writes"};
    const double tiles = (double)st[7];
    double tot0 = 0;
    for (int i = 0; i < 7; ++i) tot0 += (double)st[i];
    printf("groups %d XP %d  B %d H %d W %d: %.1f us per launch; %.0f tiles per workgroup; wave 0: %.0f cycles per tile; implied clock %.2f GHz\n", groups,
#ifdef RVDD_XP
           RVDD_XP,
#else
           0,
#endif
           B, H, W,
           1e3 * ms / iters, tiles / iters / 256, tot0 / tiles, tot0 / iters / 256 / (1e3 * ms / iters) / 1e3);
    printf("  %-38s", "cycles per tile, wave:");
    for (int w = 0; w < 8; ++w) printf(" %7d", w);
    printf("\n");
    for (int i = 0; i < 7; ++i) {
        printf("  %-38s", names[i]);
        for (int w = 0; w < 8; ++w) printf(" %7.0f", st[w * 8 + i] / (double)st[w * 8 + 7]);
        printf("\n");
    }
    return 0;
}
