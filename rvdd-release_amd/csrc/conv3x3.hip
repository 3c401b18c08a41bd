// 3x3 convolution (padding 1, 48 output channels) as an implicit GEMM on the
// exact-f32 matrix cores of gfx950 (v_mfma_f32_16x16x4_f32): the >=99 % of the
// FLOPs of convunet (networks/unet.py:26-76 NConvBlock, :194-208 ConvMaxPool2d,
// :88-147 UpConv, :635-669 bottleneck, :699-720 PostConvs).
//
// Orientation:  D[cout][pixel] += W[cout][k] * X[k][pixel]
//   A operand = weights  (lane l supplies A[row = l&15][k = l>>4])
//   B operand = pixels   (lane l supplies B[k = l>>4][col = l&15])
//   D: lane l holds col = l&15 (pixel), rows 4*(l>>4)+r (4 consecutive couts)
// so every lane ends with float4 runs of output channels for its own pixel and
// the NHWC store is one 16-B store per accumulator.
//
// Work decomposition: one workgroup (4 waves, one per SIMD) owns an 8x16 pixel
// tile of one image; wave w owns rows 2w and 2w+1 (two B fragments) and all 48
// output channels (three A fragments) -> 6 independent accumulators, which
// covers the 40-cycle dependent latency of the 32-cycle MFMA.  The filter bank
// (9*CIN*48 floats, 81 KiB for CIN=48) is staged into LDS ONCE per workgroup
// and the workgroup then walks a grid-stride list of tiles (persistent grid,
// <= one workgroup per CU), so per tile only the 10x18xCIN input halo tile
// moves.
//
// K ordering: for a tap (ky,kx) and a 16-channel chunk j, lane group g=l>>4
// reads channels 16j+4g..+3 of its pixel with ONE ds_read_b128; MFMA number i
// of that chunk consumes element i, i.e. k-slot g of MFMA i is channel
// 16j+4g+i.  The weights are pre-arranged on the host (arrange_conv3x3,
// runtime.hip) as [tap][j][cout][g][i] so that the matching A fragment is one
// ds_read_b128 too and a wave's read covers a contiguous 1 KiB (conflict-free).
#include "rvdd_internal.h"

namespace {

constexpr int TH = 8, TW = 16, IH = TH + 2, IW = TW + 2;

template <int CIN>
struct Geo {
    static constexpr int NJ = CIN / 16;
    static constexpr int W_FLOATS = 9 * NJ * 48 * 16;
    static constexpr int I_FLOATS = IH * IW * CIN;
    static constexpr size_t LDS_BYTES = (size_t)(W_FLOATS + I_FLOATS) * sizeof(float);
};

template <int CIN, int EPI, bool ACC_IN>
__global__ __launch_bounds__(256, 1) void conv3x3_kernel(ConvArgs a) {
    using G = Geo<CIN>;
    constexpr int NJ = G::NJ;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wl = smem;
    float* Il = smem + G::W_FLOATS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int lr = lane & 15;
    const int g = lane >> 4;

    // filter bank -> LDS (linear copy; the host already arranged it)
    for (int q = tid; q < G::W_FLOATS / 4; q += 256)
        reinterpret_cast<f32x4*>(Wl)[q] = reinterpret_cast<const f32x4*>(a.w)[q];

    const int tiles_per_img = a.tiles_x * a.tiles_y;
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int b = tile / tiles_per_img;
        const int rr = tile - b * tiles_per_img;
        const int ty = rr / a.tiles_x;
        const int tx = rr - ty * a.tiles_x;
        const int y0 = ty * TH, x0 = tx * TW;
        const float* inb = a.in + (size_t)b * a.H * a.W * CIN;

        __syncthreads();   // previous tile's fragment reads done (and Wl visible on the first pass)
        // halo tile -> LDS, zero outside the image (padding=1 and ragged edges)
        constexpr int C4 = CIN / 4;
        for (int q = tid; q < IH * IW * C4; q += 256) {
            const int px = q / C4;
            const int c4 = q - px * C4;
            const int iy = px / IW;
            const int ix = px - iy * IW;
            const int gy = y0 - 1 + iy, gx = x0 - 1 + ix;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if ((unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W)
                v = *reinterpret_cast<const f32x4*>(inb + ((size_t)gy * a.W + gx) * CIN + c4 * 4);
            *reinterpret_cast<f32x4*>(Il + px * CIN + c4 * 4) = v;
        }
        __syncthreads();

        // ---- accumulators: [cout block m][row n]
        f32x4 acc[3][2];
        const int yA = y0 + 2 * wave;       // rows yA, yA+1
        const int xA = x0 + lr;
#pragma unroll
        for (int m = 0; m < 3; ++m) {
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                if constexpr (ACC_IN) {
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (yA + n < a.H && xA < a.W)
                        v = *reinterpret_cast<const f32x4*>(
                            a.acc_in + (((size_t)b * a.H + yA + n) * a.W + xA) * kF + 16 * m + 4 * g);
                    acc[m][n] = v;
                } else {
                    acc[m][n] = *reinterpret_cast<const f32x4*>(a.bias + 16 * m + 4 * g);
                }
            }
        }

        const float* wbase = Wl + lr * 16 + g * 4;
        const float* ibase = Il + ((2 * wave) * IW + lr) * CIN + g * 4;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap % 3;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                f32x4 wa[3], xb[2];
#pragma unroll
                for (int m = 0; m < 3; ++m)
                    wa[m] = *reinterpret_cast<const f32x4*>(wbase + ((tap * NJ + j) * 48 + 16 * m) * 16);
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    xb[n] = *reinterpret_cast<const f32x4*>(ibase + ((n + dy) * IW + dx) * CIN + 16 * j);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int m = 0; m < 3; ++m)
#pragma unroll
                        for (int n = 0; n < 2; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[m][i], xb[n][i],
                                                                             acc[m][n], 0, 0, 0);
            }
        }

        // ---- epilogue
        if constexpr (EPI == EPI_POOL) {
            // MaxPool2d(2) of the (un-activated) conv output: rows 2w,2w+1 are the
            // two accumulators of this lane, columns pair up across lanes l, l^1.
            const int py = (y0 >> 1) + wave;
            const int px = (x0 >> 1) + (lr >> 1);
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float t = fmaxf(acc[m][0][r], acc[m][1][r]);
                    float o = __shfl_xor(t, 1);
                    v[r] = fmaxf(t, o);
                }
                if ((lr & 1) == 0 && py < a.Hout && px < a.Wout)
                    *reinterpret_cast<f32x4*>(a.out + (((size_t)b * a.Hout + py) * a.Wout + px) * kF +
                                              16 * m + 4 * g) = v;
            }
        } else {
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int y = yA + n;
                if (y < a.H && xA < a.W) {
#pragma unroll
                    for (int m = 0; m < 3; ++m) {
                        f32x4 v = acc[m][n];
                        if constexpr (EPI == EPI_RELU || EPI == EPI_RELU_ADD2) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                        }
                        if constexpr (EPI == EPI_RELU_ADD2) {
                            const size_t o = (((size_t)b * a.H + y) * a.W + xA) * kF + 16 * m + 4 * g;
                            const f32x4 r1 = *reinterpret_cast<const f32x4*>(a.res1 + o);
                            const f32x4 r2 = *reinterpret_cast<const f32x4*>(a.res2 + o);
                            // s = e3 + d1 + d2 in the reference's order (unet.py:563-566)
                            v = (r1 + r2) + v;
                        }
                        *reinterpret_cast<f32x4*>(
                            a.out + (((size_t)b * a.Hout + y + a.oy) * a.Wout + xA + a.ox) * kF + 16 * m +
                            4 * g) = v;
                    }
                }
            }
        }
    }
}

template <int CIN, int EPI, bool ACC_IN>
hipError_t launch_t(const ConvArgs& a, hipStream_t s) {
    static bool attr_done = false;
    auto kern = conv3x3_kernel<CIN, EPI, ACC_IN>;
    constexpr size_t lds = Geo<CIN>::LDS_BYTES;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int grid = a.ntiles < cus ? a.ntiles : cus;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a);
    return hipGetLastError();
}

template <int CIN>
hipError_t launch_c(const ConvArgs& a, int epi, hipStream_t s) {
    const bool acc = a.acc_in != nullptr;
    switch (epi) {
        case EPI_NONE:
            return acc ? launch_t<CIN, EPI_NONE, true>(a, s) : launch_t<CIN, EPI_NONE, false>(a, s);
        case EPI_RELU:
            return acc ? launch_t<CIN, EPI_RELU, true>(a, s) : launch_t<CIN, EPI_RELU, false>(a, s);
        case EPI_POOL:
            return acc ? hipErrorInvalidValue : launch_t<CIN, EPI_POOL, false>(a, s);
        case EPI_RELU_ADD2:
            return acc ? hipErrorInvalidValue : launch_t<CIN, EPI_RELU_ADD2, false>(a, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace

size_t conv3x3_weight_floats(int cin) { return (size_t)9 * (cin / 16) * 48 * 16; }

hipError_t launch_conv3x3(const ConvArgs& a, int cin, int epi, hipStream_t s) {
    if (a.ntiles <= 0) return hipSuccess;
    if (cin == 48) return launch_c<48>(a, epi, s);
    if (cin == 16) return launch_c<16>(a, epi, s);
    return hipErrorInvalidValue;
}
