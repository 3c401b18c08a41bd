// 3x3 convolution 48 -> 48 (padding 1) as Winograd F(4x4,3x3) on the exact-f32 matrix cores:
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A   with 6x6 = 36 transform positions per 4x4-pixel output tile,
// 36 / 16 = 2.25 multiplies per output pixel and channel pair where F(2x2,3x3) (wino3x3.hip) needs 4 and the direct
// form 9: 1.78x fewer MFMAs than wino3x3.hip for the same networks/unet.py layers.
//
// The transformed filter bank of a layer is 36 x 48 x 48 floats = 324 KiB, twice the LDS.  It is cut by OUTPUT channels:
// a workgroup holds the bank of 16 couts (36 x 48 x 16 floats = 108 KiB) and computes those 16 output channels of its
// units; the three workgroups of a unit (blockIdx slots 3k, 3k+1, 3k+2 of one XCD) read the same input through that
// XCD's L2 and each repeats the input transform.  What makes this pay on gfx950: the matrix work falls by 1.78x, the
// vector work (input transform 864 packed ops per wave and unit, output transform ~200) is tripled but starts small.
//
// STATUS (round 3): parity-green (max-abs 1-2e-6 against the F(2x2,3x3) path at every size tried, ragged ones included),
// but SLOWER than wino3x3.hip on MI355X -- C2 408 against 505 frames/s -- and therefore OFF (rvdd_set_option "wino4" /
// RVDD_WINO4).  The matrix work does fall by 1.78x; what the kernel then waits for is memory: a 36-element patch of four
// channels per lane is 144 registers, so the next stage's patch can only be requested as the current one is consumed
// (two loads per MFMA pair-step), the three cout thirds re-read the input (6.75 patch bytes per output pixel and third
// against 4 for F(2x2)), and every workgroup stores only 64 of a pixel's 192 bytes.  Timing variants of this build
// (profiles/r03_wino4_f4x4.json): no patch loads 529 frames/s, no epilogue 524, no input transform 410 (the transform is
// already hidden in the memory stalls); all 36 loads at the start of a stage: 43 registers spilled, 361; 8-channel
// chunks (half the patch registers, full prefetch, twice the load instructions): 310.
//
// Lane <-> data map (the MFMA B/D map, as wino3x3.hip): lane l of a wave owns output tile l & 15 of the wave's row of 16
// tiles (64 x 4 pixels) and, per 16-channel chunk j, input channels 16j + 4g .. +3 (g = l >> 4); after the GEMMs it owns
// output channels 16 t + 4g .. +3 (t = the workgroup's cout third) of that tile.  A workgroup = 4 waves = 4 tile rows = a
// unit of 64 x 16 pixels.
#include "rvdd_internal.h"

#include <type_traits>

namespace {

constexpr int W4_POS = 36;
constexpr int W4_BANK_FLOATS = W4_POS * 3 * 256;        // one cout third: [pos 36][chunk 3][lane 64][4]
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ f32x4 bload(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff = 0) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ void bstore(__amdgpu_buffer_rsrc_t r, unsigned off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, off, 0, 0);
}
__device__ __forceinline__ f32x4 splat(float s) { return f32x4{s, s, s, s}; }
__device__ __forceinline__ f32x4 fma4(f32x4 a, float s, f32x4 c) { return __builtin_elementwise_fma(a, splat(s), c); }
// a - b as two v_pk_add_f32 with a negated operand (hipcc selects four scalar v_sub_f32 for a vector subtraction, and
// beside f32 MFMAs every vector instruction costs its full time)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 sub4(f32x4 a, f32x4 b) {
    f32x2 lo, hi;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(lo) : "v"(f32x2{a[0], a[1]}), "v"(f32x2{b[0], b[1]}));
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(hi) : "v"(f32x2{a[2], a[3]}), "v"(f32x2{b[2], b[3]}));
    return f32x4{lo[0], lo[1], hi[0], hi[1]};
}

// B^T (6x6) of F(4,3) on six values, in place:
//   t0 = 4 d0 - 5 d2 + d4          t1 = (d4 - 4 d2) + (d3 - 4 d1)      t2 = (d4 - 4 d2) - (d3 - 4 d1)
//   t3 = (d4 - d2) + 2 (d3 - d1)   t4 = (d4 - d2) - 2 (d3 - d1)        t5 = 4 d1 - 5 d3 + d5
__device__ __forceinline__ void bt6(f32x4& d0, f32x4& d1, f32x4& d2, f32x4& d3, f32x4& d4, f32x4& d5) {
    const f32x4 a = fma4(d2, -4.f, d4);
    const f32x4 b = fma4(d1, -4.f, d3);
    const f32x4 c = sub4(d4, d2);
    const f32x4 e = sub4(d3, d1);
    const f32x4 t0 = fma4(d0, 4.f, fma4(d2, -5.f, d4));
    const f32x4 t5 = fma4(d1, 4.f, fma4(d3, -5.f, d5));
    d0 = t0;
    d1 = a + b;
    d2 = sub4(a, b);
    d3 = fma4(e, 2.f, c);
    d4 = fma4(e, -2.f, c);
    d5 = t5;
}
// A^T (4x6) of F(4,3) on six values -> four:
//   y0 = m0 + (m1 + m2) + (m3 + m4)    y1 = (m1 - m2) + 2 (m3 - m4)    y2 = (m1 + m2) + 4 (m3 + m4)
//   y3 = (m1 - m2) + 8 (m3 - m4) + m5
__device__ __forceinline__ void at6(f32x4 m0, f32x4 m1, f32x4 m2, f32x4 m3, f32x4 m4, f32x4 m5, f32x4& y0, f32x4& y1, f32x4& y2,
                                    f32x4& y3) {
    const f32x4 s = m1 + m2, d = sub4(m1, m2), p = m3 + m4, q = sub4(m3, m4);
    y0 = (m0 + s) + p;
    y1 = fma4(q, 2.f, d);
    y2 = fma4(p, 4.f, s);
    y3 = fma4(q, 8.f, d) + m5;
}

struct UnitPos {
    int b, ty, tx;
};

template <int EPI, bool ACC_IN>
__global__ __launch_bounds__(256, 1) void wino4_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float U[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15;
    const int g = lane >> 4;
    // blockIdx -> (XCD, slot): slots 3k, 3k+1, 3k+2 of one XCD are the three cout thirds of the same unit sequence
    const int xcd = (int)(blockIdx.x & 7), slot = (int)(blockIdx.x >> 3);
    const int third = slot % 3;
    const int nranks = (int)(gridDim.x / 24) * 8;             // unit sequences = workgroups / 3
    int unit = (slot / 3) * 8 + xcd;

    {   // this third's bank -> LDS
        __amdgpu_buffer_rsrc_t wr =
            __builtin_amdgcn_make_buffer_rsrc((void*)(a.w + (size_t)third * W4_BANK_FLOATS), 0, W4_BANK_FLOATS * 4, 0x00020000);
        constexpr int NP = W4_BANK_FLOATS / 256;
        const int rot = (int)((blockIdx.x * 37u) % (unsigned)NP);
        for (int i = wave; i < NP; i += 4) {
            int k = i + rot;
            if (k >= NP) k -= NP;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void*)(U + k * 256), 16, (unsigned)(k * 1024 + lane * 16), 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (unit >= a.ntiles) return;

    const int units_per_img = a.tiles_x * a.tiles_y;
    const unsigned map_bytes = (unsigned)(a.H * a.W * kF * 4);
    const unsigned out_bytes = (unsigned)(a.Hout * a.Wout * kF * 4);
    const int row_bytes = a.W * kF * 4;
    auto locate = [&](int u_, UnitPos& u) {
        u.b = __builtin_amdgcn_readfirstlane(u_ / units_per_img);
        const int rr = u_ - u.b * units_per_img;
        const int uy = __builtin_amdgcn_readfirstlane(rr / a.tiles_x);
        const int ux = rr - uy * a.tiles_x;
        u.ty = uy * 4 + wave;
        u.tx = __builtin_amdgcn_readfirstlane(ux * 16) + lr;
    };
    // the 6x6 input patch of a tile, chunk j: rows 4 ty - 1 .. 4 ty + 4, columns 4 tx - 1 .. 4 tx + 4.  One buffer
    // descriptor per patch row (zero records for rows outside the image), columns right of the image fall out of the
    // row's range, the column left of it (x = -1) is a lane select: zero padding without a branch.
    auto load_patch = [&](f32x4 (&p)[36], const UnitPos& u, int j) {
        const float* img = a.in + (size_t)u.b * a.H * a.W * kF;
        const int y0 = 4 * u.ty - 1, x0 = 4 * u.tx - 1;
        const unsigned v1 = (unsigned)((x0 + 1) * (kF * 4) + 16 * g);       // pixel x0 + 1 >= 0
        const unsigned v0 = x0 >= 0 ? v1 - kF * 4 : 0xffffffffu;
#pragma unroll
        for (int dy = 0; dy < 6; ++dy) {
            const int y = y0 + dy;
            const bool row_ok = (unsigned)y < (unsigned)a.H;
            __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(img + (ptrdiff_t)y * a.W * kF), 0,
                                                                         row_ok ? row_bytes : 0, 0x00020000);
            p[dy * 6 + 0] = bload(r, v0, 64 * j);
#pragma unroll
            for (int dx = 1; dx < 6; ++dx) p[dy * 6 + dx] = bload(r, v1 + (dx - 1) * (kF * 4), 64 * j);
        }
    };

    typedef __attribute__((address_space(3))) f32x4 lds_frag;
    lds_frag* ubp[2];                                         // positions 0-17, 18-35 (the bank exceeds a ds_read's 64 KiB offset)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        ubp[k] = (lds_frag*)U + lane + k * 18 * 3 * 64;
        asm volatile("" : "+v"(ubp[k]));
    }
    auto ldsA = [&](int j, int pos) { return ubp[pos / 18][((pos % 18) * 3 + j) * 64]; };

    // one patch element (dy, dx) of chunk j: the loads of the NEXT stage's patch are issued two per MFMA pair-step of the
    // current stage, into the registers the consumed positions leave behind (the two patch arrays swap roles every stage)
    auto load_elem = [&](f32x4 (&pp)[36], const UnitPos& u, int j, int e) {
        const int dy = e / 6, dx = e - 6 * dy;
        const float* img = a.in + (size_t)u.b * a.H * a.W * kF;
        const int y = 4 * u.ty - 1 + dy, x0 = 4 * u.tx - 1;
        const bool row_ok = (unsigned)y < (unsigned)a.H;
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(img + (ptrdiff_t)y * a.W * kF), 0,
                                                                     row_ok ? row_bytes : 0, 0x00020000);
        const unsigned v1 = (unsigned)((x0 + 1) * (kF * 4) + 16 * g);
        pp[e] = bload(r, dx == 0 ? (x0 >= 0 ? v1 - kF * 4 : 0xffffffffu) : v1 + (dx - 1) * (kF * 4), 64 * j);
    };

    f32x4 acc[36];
    f32x4 pa[36], pb[36];
    f32x4 wq[2][2];
    UnitPos cur, nxt;
    locate(unit, cur);
    load_patch(pa, cur, 0);

    // One stage = chunk J of the current unit on the patch array P (raw on entry), the other array Q receiving the raw
    // patch of the stage after it (chunk J + 1, or chunk 0 of the next unit).
    auto stage = [&](auto JC, f32x4 (&P)[36], f32x4 (&Q)[36], const UnitPos& ld_u, int ld_j) {
        constexpr int J = decltype(JC)::value;
        // B^T d B in place: rows first (the last row's loads are the youngest), then columns
#pragma unroll
        for (int y = 0; y < 6; ++y) {
            bt6(P[6 * y], P[6 * y + 1], P[6 * y + 2], P[6 * y + 3], P[6 * y + 4], P[6 * y + 5]);
            if (y == 1 || y == 3) __builtin_amdgcn_sched_barrier(0);     // rows in the order their loads were issued
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int x = 0; x < 6; ++x) bt6(P[x], P[6 + x], P[12 + x], P[18 + x], P[24 + x], P[30 + x]);
        __builtin_amdgcn_sched_barrier(0);
        // 36 positions in 18 pairs: the two accumulator chains of a pair alternate (40-cycle dependent latency against a
        // 32-cycle issue interval); the A fragments of the next pair (of the next stage's first pair, at the end) are read
        // one pair ahead
#pragma unroll
        for (int s = 0; s < 18; ++s) {
            if (s + 1 < 18) {
                wq[(s + 1) & 1][0] = ldsA(J, 2 * s + 2);
                wq[(s + 1) & 1][1] = ldsA(J, 2 * s + 3);
            } else {
                wq[0][0] = ldsA((J + 1) % 3, 0);
                wq[0][1] = ldsA((J + 1) % 3, 1);
            }
            load_elem(Q, ld_u, ld_j, 2 * s);
            load_elem(Q, ld_u, ld_j, 2 * s + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int pos = 2 * s + h;
                    const f32x4 c = (J == 0 && i == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[pos];
                    acc[pos] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[s & 1][h][i], P[pos][i], c, 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto epilogue = [&]() {
        // ---- output transform A^T M A (over the row index first), bias / partial sums, ReLU, stores
        const f32x4 bias = ACC_IN ? f32x4{0.f, 0.f, 0.f, 0.f}
                                  : bload(__builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, kF * 4, 0x00020000),
                                          (unsigned)((16 * third + 4 * g) * 4));
        f32x4 t[4][6];     // after the pass over the row index: t[r][column position]
#pragma unroll
        for (int x = 0; x < 6; ++x) at6(acc[x], acc[6 + x], acc[12 + x], acc[18 + x], acc[24 + x], acc[30 + x], t[0][x], t[1][x], t[2][x], t[3][x]);
        __amdgpu_buffer_rsrc_t orr =
            __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (size_t)cur.b * a.Hout * a.Wout * kF), 0, out_bytes, 0x00020000);
        __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc(
            (void*)((ACC_IN ? a.acc_in : a.out) + (size_t)cur.b * a.H * a.W * kF), 0, map_bytes, 0x00020000);
        const unsigned chan = (unsigned)((16 * third + 4 * g) * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            f32x4 y[4];
            at6(t[r][0], t[r][1], t[r][2], t[r][3], t[r][4], t[r][5], y[0], y[1], y[2], y[3]);
            const int yy = 4 * cur.ty + r;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int xx = 4 * cur.tx + c;
                const bool ok = yy < a.H && xx < a.W;
                f32x4 v = y[c] + bias;
                if constexpr (ACC_IN) v = v + bload(pr, ok ? (unsigned)((yy * a.W + xx) * (kF * 4)) + chan : 0x80000000u);
                if constexpr (EPI == EPI_RELU) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
                }
                bstore(orr, ok ? (unsigned)(((yy + a.oy) * a.Wout + xx + a.ox) * (kF * 4)) + chan : 0x80000000u, v);
            }
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    wq[0][0] = ldsA(0, 0);
    wq[0][1] = ldsA(0, 1);
    // a unit has three stages, so the patch arrays swap roles from one unit to the next: two units per loop iteration
#pragma unroll 1
    for (;;) {
        locate(unit + nranks, nxt);          // past the end: every row out of range -> zeros, nothing stored
        if (unit + nranks >= a.ntiles) nxt.ty = 1 << 20;
        stage(I0{}, pa, pb, cur, 1);
        stage(I1{}, pb, pa, cur, 2);
        stage(I2{}, pa, pb, nxt, 0);
        epilogue();
        cur = nxt;
        unit += nranks;
        if (unit >= a.ntiles) break;
        locate(unit + nranks, nxt);
        if (unit + nranks >= a.ntiles) nxt.ty = 1 << 20;
        stage(I0{}, pb, pa, cur, 1);
        stage(I1{}, pa, pb, cur, 2);
        stage(I2{}, pb, pa, nxt, 0);
        epilogue();
        cur = nxt;
        unit += nranks;
        if (unit >= a.ntiles) break;
    }
}

template <int EPI, bool ACC_IN>
hipError_t launch_w4(const ConvArgs& a0, hipStream_t s) {
    static std::atomic<uint64_t> attr_done{0};
    void (*kern)(ConvArgs) = wino4_kernel<EPI, ACC_IN>;
    constexpr size_t LDS = (size_t)W4_BANK_FLOATS * 4;
    if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(kern), LDS, attr_done); e != hipSuccess) return e;
    ConvArgs a = a0;
    a.tiles_x = (a.W + 63) / 64;      // unit = 4 tile rows x 16 tile columns = 16 x 64 output pixels
    a.tiles_y = (a.H + 15) / 16;
    a.ntiles = a.B * a.tiles_x * a.tiles_y;
    // three workgroups (cout thirds) per unit sequence, in groups of 24 = 3 slots x 8 XCDs
    const int cus = current_device_cus();
    int groups = cus / 24;                                             // 10 on 256 CUs: 240 workgroups
    const int need = (a.ntiles + 7) / 8;                               // unit sequences are dealt 8 (one per XCD) at a time
    if (groups > need) groups = need;
    if (groups < 1) groups = 1;
    hipLaunchKernelGGL(kern, dim3(groups * 24), dim3(256), LDS, s, a);
    return hipGetLastError();
}

}  // namespace

size_t wino4x4_weight_floats() { return 3 * (size_t)W4_BANK_FLOATS; }

hipError_t launch_wino4x4(const ConvArgs& a, int epi, hipStream_t s) {
    if (a.ntiles <= 0) return hipSuccess;
    if (a.ups) return hipErrorInvalidValue;
    if ((size_t)a.H * a.W * kF * 4 >= 0x80000000ull || (size_t)a.Hout * a.Wout * kF * 4 >= 0x80000000ull) return hipErrorInvalidValue;
    const bool acc = a.acc_in != nullptr;
    switch (epi) {
        case EPI_NONE:
            return acc ? launch_w4<EPI_NONE, true>(a, s) : launch_w4<EPI_NONE, false>(a, s);
        case EPI_RELU:
            return acc ? launch_w4<EPI_RELU, true>(a, s) : launch_w4<EPI_RELU, false>(a, s);
    }
    return hipErrorInvalidValue;
}
