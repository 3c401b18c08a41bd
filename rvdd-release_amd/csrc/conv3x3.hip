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
// <= one workgroup per CU).  Per tile only the 10x18xCIN input halo tile moves:
// it is fetched by LDS-DMA (buffer_load_dwordx4 ... lds) into the other half
// of a double buffer while the MFMAs run on the current one; out-of-image
// pixels (padding=1 and ragged edges) are given an out-of-range buffer offset,
// which the hardware range check turns into zeros.
//
// K ordering: for a tap (ky,kx) and a 16-channel chunk j, lane group g=l>>4
// reads channels 16j+4g..+3 of its pixel with ONE ds_read_b128; MFMA number i
// of that chunk consumes element i, i.e. k-slot g of MFMA i is channel
// 16j+4g+i.  The weights are pre-arranged on the host (arrange_conv3x3,
// runtime.hip) as [tap][j][m][lane = 16g + cout&15][i] so that the matching A
// fragment is one ds_read_b128 too, lane-linear: each of the four 16-lane groups
// the read is served in covers one whole 256-B bank row (conflict-free).
// Fragments for group n+1 are read while the 24 MFMAs of group n issue.
#include "rvdd_internal.h"

namespace {

constexpr int TH = 8, TW = 16, IH = TH + 2, IW = TW + 2;

template <int CIN>
struct Geo {
    static constexpr int NJ = CIN / 16;
    static constexpr int C4 = CIN / 4;
    static constexpr int W_FLOATS = 9 * NJ * 48 * 16;
    static constexpr int I_FLOATS = IH * IW * CIN;
    static constexpr int NCHUNK = IH * IW * C4;               // 16-B pieces of one halo tile
    static constexpr int NINST = (NCHUNK + 63) / 64;           // wave-wide DMA instructions per tile
    static constexpr int KPW = (NINST + 3) / 4;                // ... per wave
    static constexpr int W_NINST = W_FLOATS / 256;             // weights: whole KiB pieces
    static constexpr size_t LDS_BYTES = (size_t)(W_FLOATS + 2 * I_FLOATS) * sizeof(float);
};

typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, float* lds_wave_base, unsigned voff) {
    // one 16-B piece per lane: LDS[lds_wave_base + 16*lane] <- buffer[voff]  (0 when voff is out of range)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_wave_base, 16, voff, 0, 0, 0);
}

template <int CIN, int EPI, bool ACC_IN, int VARIANT>
__global__ __launch_bounds__(256, 1) void conv3x3_kernel(ConvArgs a) {
    using G = Geo<CIN>;
    constexpr int NJ = G::NJ;
    constexpr int NG = 9 * NJ;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wl = smem;
    float* Ibuf = smem + G::W_FLOATS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15;
    const int g = lane >> 4;
    const int tiles_per_img = a.tiles_x * a.tiles_y;

    // ---- filter bank -> LDS by DMA (linear copy; the host already arranged it)
    {
        __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, G::W_FLOATS * 4, 0x00020000);
        for (int k = wave; k < G::W_NINST; k += 4) dma16(wr, Wl + k * 256, (unsigned)(k * 1024 + lane * 16));
    }

    // ---- per-lane description of its DMA pieces (tile independent)
    int rel[G::KPW];   // byte offset of the piece relative to the tile's (-1,-1) corner
    int iyx[G::KPW];   // (iy << 8) | ix, or -1 when the piece does not exist
#pragma unroll
    for (int k = 0; k < G::KPW; ++k) {
        const int q = (k * 4 + wave) * 64 + lane;
        const int px = q / G::C4;
        const int c4 = q - px * G::C4;
        const int iy = px / IW;
        const int ix = px - iy * IW;
        rel[k] = ((iy * a.W + ix) * CIN + c4 * 4) * 4;
        iyx[k] = q < G::NCHUNK ? ((iy << 8) | ix) : -1;
    }

    auto issue_tile = [&](int tile, float* dst) {
        // integer division runs on the VALU: mark the results wave-uniform so that the buffer
        // descriptor stays in SGPRs (no waterfall loop around each DMA)
        const int b = __builtin_amdgcn_readfirstlane(tile / tiles_per_img);
        const int rr = tile - b * tiles_per_img;
        const int ty = __builtin_amdgcn_readfirstlane(rr / a.tiles_x);
        const int tx = rr - ty * a.tiles_x;
        const int y0 = ty * TH - 1, x0 = tx * TW - 1;
        __amdgpu_buffer_rsrc_t ir = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(a.in + (size_t)b * a.H * a.W * CIN), 0, a.H * a.W * CIN * 4, 0x00020000);
        const int base = (y0 * a.W + x0) * CIN * 4;
#pragma unroll
        for (int k = 0; k < G::KPW; ++k) {
            if (iyx[k] >= 0) {
                const int gy = y0 + (iyx[k] >> 8), gx = x0 + (iyx[k] & 255);
                const bool ok = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
                dma16(ir, dst + (k * 4 + wave) * 256, ok ? (unsigned)(base + rel[k]) : 0x80000000u);
            }
        }
    };

    // bias is tile invariant
    f32x4 bv[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        if constexpr (ACC_IN) bv[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        else bv[m] = *reinterpret_cast<const f32x4*>(a.bias + 16 * m + 4 * g);
    }

    // Every wave issues exactly NS output stores per tile (buffer stores; lanes outside the
    // image get an out-of-range offset and are dropped by the range check), so that the wait
    // for the NEXT tile's DMA can be a counted vmcnt(NS) that leaves the stores in flight.
    constexpr int NS = EPI == EPI_POOL ? 3 : 6;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

    int tile = blockIdx.x;
    issue_tile(tile, Ibuf);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;

    for (; tile < a.ntiles; tile += gridDim.x) {
        const int b = __builtin_amdgcn_readfirstlane(tile / tiles_per_img);
        const int rr = tile - b * tiles_per_img;
        const int ty = __builtin_amdgcn_readfirstlane(rr / a.tiles_x);
        const int tx = rr - ty * a.tiles_x;
        const int y0 = ty * TH, x0 = tx * TW;
        const float* Il = Ibuf + cur * G::I_FLOATS;

        // ---- accumulators: [cout block m][row n].  Partial-sum loads go out BEFORE the next
        // tile's DMA so that waiting for them does not wait for the DMA (vmcnt is in order).
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
                    acc[m][n] = bv[m];
                }
            }
        }
        if constexpr (ACC_IN) __builtin_amdgcn_sched_barrier(0);
        // every wave is past the barrier that ended the previous tile: the other buffer is free
        if (tile + (int)gridDim.x < a.ntiles) issue_tile(tile + gridDim.x, Ibuf + (cur ^ 1) * G::I_FLOATS);

        const float* wbase = Wl + lane * 4;      // lane-linear fragments: each ds_read_b128 lane group covers one bank row
        const float* ibase = Il + ((2 * wave) * IW + lr) * CIN + g * 4;
        f32x4 wa[2][3], xb[2][2];
        auto ld = [&](int grp, int slot) {
            const int tap = grp / NJ, j = grp - tap * NJ;
            const int dy = tap / 3, dx = tap - 3 * dy;
#pragma unroll
            for (int m = 0; m < 3; ++m)
                wa[slot][m] = *reinterpret_cast<const f32x4*>(wbase + ((tap * NJ + j) * 48 + 16 * m) * 16);
#pragma unroll
            for (int n = 0; n < 2; ++n)
                xb[slot][n] = *reinterpret_cast<const f32x4*>(ibase + ((n + dy) * IW + dx) * CIN + 16 * j);
        };
        ld(0, 0);
        if constexpr (VARIANT == 2) __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
#pragma unroll
        for (int grp = 0; grp < NG; ++grp) {
            const int s = grp & 1;
            if (grp + 1 < NG) ld(grp + 1, s ^ 1);
            // pin the order [reads of group n+1] [24 MFMAs of group n]: left alone, hipcc sinks the
            // reads next to their first use and exposes the LDS latency once per group (-5 % at 720p)
            if constexpr (VARIANT == 0) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int m = 0; m < 3; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[s][m][i], xb[s][n][i], acc[m][n], 0,
                                                                         0, 0);
            if constexpr (VARIANT == 0) __builtin_amdgcn_sched_barrier(0);
            if constexpr (VARIANT == 2) {
                // spread the next group's five fragment reads between this group's MFMAs
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
        }

        // ---- epilogue
        __amdgpu_buffer_rsrc_t orr = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(a.out + (size_t)b * a.Hout * a.Wout * kF), 0, a.Hout * a.Wout * kF * 4, 0x00020000);
        if constexpr (EPI == EPI_POOL) {
            // MaxPool2d(2) of the (un-activated) conv output: rows 2w,2w+1 are the
            // two accumulators of this lane, columns pair up across lanes l, l^1.
            const int py = (y0 >> 1) + wave;
            const int px = (x0 >> 1) + (lr >> 1);
            const bool ok = (lr & 1) == 0 && py < a.Hout && px < a.Wout;
            const unsigned obase = ok ? (unsigned)(((py * a.Wout + px) * kF + 4 * g) * 4) : 0x80000000u;
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float t = fmaxf(acc[m][0][r], acc[m][1][r]);
                    float o = __shfl_xor(t, 1);
                    v[r] = fmaxf(t, o);
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), orr, obase + 64 * m, 0, 0);
            }
        } else {
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int y = yA + n;
                const bool ok = y < a.H && xA < a.W;
                const unsigned obase =
                    ok ? (unsigned)((((y + a.oy) * a.Wout + xA + a.ox) * kF + 4 * g) * 4) : 0x80000000u;
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    f32x4 v = acc[m][n];
                    if constexpr (EPI == EPI_RELU || EPI == EPI_RELU_ADD2) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                    }
                    if constexpr (EPI == EPI_RELU_ADD2) {
                        const size_t o = ok ? (((size_t)b * a.H + y) * a.W + xA) * kF + 16 * m + 4 * g : 0;
                        const f32x4 r1 = *reinterpret_cast<const f32x4*>(a.res1 + o);
                        const f32x4 r2 = *reinterpret_cast<const f32x4*>(a.res2 + o);
                        // s = e3 + d1 + d2 in the reference's order (unet.py:563-566)
                        v = (r1 + r2) + v;
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), orr, obase + 64 * m, 0, 0);
                }
            }
        }
        // the next tile's DMA (older than the NS stores just issued) has landed for this wave;
        // after the barrier it has landed for every wave and every wave is done reading `cur`
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        cur ^= 1;
    }
}

int g_variant = 0;   // A/B switch for the measurement hook (rvdd_debug_conv_bench)

template <int CIN, int EPI, bool ACC_IN, int VARIANT>
hipError_t launch_v(const ConvArgs& a, hipStream_t s) {
    static std::atomic<uint64_t> attr_done{0};
    auto kern = conv3x3_kernel<CIN, EPI, ACC_IN, VARIANT>;
    constexpr size_t lds = Geo<CIN>::LDS_BYTES;
    if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(kern), lds, attr_done); e != hipSuccess) return e;
    const int cus = current_device_cus();
    const int grid = a.ntiles < cus ? a.ntiles : cus;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a);
    return hipGetLastError();
}

template <int CIN, int EPI, bool ACC_IN>
hipError_t launch_t(const ConvArgs& a, hipStream_t s) {
    if constexpr (CIN == 48 && EPI == EPI_RELU && !ACC_IN) {
        if (g_variant == 1) return launch_v<CIN, EPI, ACC_IN, 1>(a, s);
        if (g_variant == 2) return launch_v<CIN, EPI, ACC_IN, 2>(a, s);
    }
    return launch_v<CIN, EPI, ACC_IN, 0>(a, s);
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

void conv3x3_set_variant(int v) { g_variant = v; }

size_t conv3x3_weight_floats(int cin) { return (size_t)9 * (cin / 16) * 48 * 16; }

hipError_t launch_conv3x3(const ConvArgs& a, int cin, int epi, hipStream_t s) {
    if (a.ntiles <= 0) return hipSuccess;
    if ((size_t)a.H * a.W * cin * 4 >= 0x80000000ull) return hipErrorInvalidValue;   // 32-bit buffer offsets
    if (cin == 48) return launch_c<48>(a, epi, s);
    if (cin == 16) return launch_c<16>(a, epi, s);
    return hipErrorInvalidValue;
}
