// 3x3 convolution 48 -> 48 (padding 1) as Winograd F(2x2,3x3) on the exact-f32
// matrix cores: Y = A^T [ (G g G^T) .* (B^T d B) ] A, the element-wise product
// summed over input channels being 16 independent [48 x 48] x [48 x tiles] GEMMs
// on v_mfma_f32_16x16x4_f32.  2.25x fewer MFMAs than the direct implicit GEMM of
// conv3x3.hip (36 instead of 81 per 16 output pixels and 48 couts), same
// networks/unet.py layers, same epilogues.
//
// Lane <-> data map (the MFMA B/D map again): lane l of a wave owns output tile
// (2x2 pixels) number l&15 of the wave's row of 16 tiles and, per 16-channel chunk
// j, input channels 16j+4g..+3 (g = l>>4); after the GEMMs it owns output channels
// 16m+4g..+3 of that tile for the three cout blocks m.
//  * The 4x4 input patch of a tile is read straight from HBM/L2 into registers
//    (16 buffer loads of 16 B per chunk; out-of-image pixels get an out-of-range
//    offset = hardware zero fill = padding 1) and transformed IN REGISTERS: there is no
//    LDS tile and therefore no barrier inside the tile loop -- the four waves of a
//    workgroup run independently after the filter bank is resident.
//  * The transformed filter bank U (16 positions x 48 x 48 floats = 144 KiB) is
//    DMA'd to LDS once per (persistent) workgroup, pre-arranged on the host as
//    [pos][j][m][lane = 16g + (cout&15)][i]: an A fragment is one ds_read_b128 per
//    four MFMAs whose four lane groups each cover one whole 256-B bank row.
//  * Per wave and unit (16 tiles = 2x32 pixels): 3 chunks x 16 positions x 3 cout
//    blocks x 4 k-steps = 576 MFMAs on 48 accumulators (192 VGPRs); the patch of the
//    next chunk / next unit is in flight while the current one is multiplied.
#include "rvdd_internal.h"

#include <type_traits>

namespace {

// transformed filter bank: 16 positions x NJ 16-channel input chunks x 3 cout blocks x 256 floats
// (NJ = 3: 48 -> 48, 144 KiB; NJ = 1: the first layer, 16 (6 or 9 real) -> 48, 48 KiB)
constexpr int u_floats(int nj) { return 16 * nj * 3 * 256; }
constexpr int U_FLOATS = u_floats(3);
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

// voff per lane, soff wave-uniform (an SGPR or an inline constant of the instruction: costs no vector register)
__device__ __forceinline__ f32x4 bload(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff = 0) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ void bstore(__amdgpu_buffer_rsrc_t r, unsigned off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, off, 0, 0);
}

// a - b as two v_pk_add_f32 with negated second operand: hipcc selects v_pk_add_f32 for vector
// adds but four scalar v_sub_f32 for vector subtractions, and on gfx950 the f32 MFMA shares the
// SIMD's f32 lanes with the VALU (VALU work is NOT hidden behind f32 MFMAs: measured, a finely
// interleaved schedule ran slower than a clumped one), so every VALU instruction saved counts.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 sub4(f32x4 a, f32x4 b) {
    f32x2 lo, hi;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(lo) : "v"(f32x2{a[0], a[1]}), "v"(f32x2{b[0], b[1]}));
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(hi) : "v"(f32x2{a[2], a[3]}), "v"(f32x2{b[2], b[3]}));
    return f32x4{lo[0], lo[1], hi[0], hi[1]};
}

// a * s + c with ONE rounding (the interpolation of the fused upsample: the same expression, written the same way, in
// upsample2x_kernel of prestage.hip, so that both produce the same bits)
__device__ __forceinline__ f32x4 fma4(f32x4 a, float s, f32x4 c) {
    return __builtin_elementwise_fma(a, f32x4{s, s, s, s}, c);
}

// B^T d B in place on a 4x4 patch of float4 (p[y*4+x]), in 8 slices so that the caller can
// spread it between MFMA groups: slices 0..3 = column pass (x = slice), 4..7 = row pass.
__device__ __forceinline__ void transform_slice(f32x4 (&p)[16], int sl) {
    if (sl < 4) {
        const int x = sl;
        const f32x4 d0 = p[x], d1 = p[4 + x], d2 = p[8 + x], d3 = p[12 + x];
        p[x] = sub4(d0, d2);
        p[4 + x] = d1 + d2;
        p[8 + x] = sub4(d2, d1);
        p[12 + x] = sub4(d1, d3);
    } else {
        const int y = sl - 4;
        const f32x4 t0 = p[4 * y], t1 = p[4 * y + 1], t2 = p[4 * y + 2], t3 = p[4 * y + 3];
        p[4 * y] = sub4(t0, t2);
        p[4 * y + 1] = t1 + t2;
        p[4 * y + 2] = sub4(t2, t1);
        p[4 * y + 3] = sub4(t1, t3);
    }
}

struct UnitPos {
    int b, ty, tx;
};

// UPS: the conv input is nn.Upsample(scale_factor=2, mode="bilinear") [align_corners=False] of the map `a.in`
// points to ([B][H/2][W/2][48], networks/unet.py:113-118 UpConv): the 4x4 patch of a tile is interpolated in
// registers from the 3x3 low-resolution pixels it depends on, the upsampled map is never written.
template <int EPI, bool ACC_IN, int NJ, bool UPS = false>
__device__ __forceinline__ void wino_body(const ConvArgs& a) {
    extern __shared__ __attribute__((aligned(16))) float U[];
    constexpr int CIN = 16 * NJ;          // channels per input pixel (NHWC); outputs and side inputs are always kF wide
    constexpr int UF = u_floats(NJ);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15;
    const int g = lane >> 4;

    {   // transformed filter bank -> LDS (linear copy of the host arrangement)
        __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, UF * 4, 0x00020000);
        // every workgroup copies the same 1-KiB pieces: each starts at a different one, so that the 256 copies do not
        // queue on the same L2 lines at the same time (+0.35 % on the frame rate; starting the first stage on the
        // first third of the bank while the rest lands was measured and lost 1.5 %: two more barriers per launch)
        constexpr int NP = UF / 256;
        const int rot = (int)((blockIdx.x * 37u) % (unsigned)NP);
        for (int i = wave; i < NP; i += 4) {
            int k = i + rot;
            if (k >= NP) k -= NP;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void*)(U + k * 256), 16, (unsigned)(k * 1024 + lane * 16), 0, 0, 0);
        }
    }
    // EPI_RELU_OUT3: the 1x1 conv's weights [3][48] and bias [3] behind the bank.  Read from global memory in the
    // epilogue they sat behind the epilogue's own stores (loads and stores retire in order on one counter): every
    // one of the 40 little loads waited for the stores before it, +190 us on a 616 us launch.
    float* W3l = U + UF;
    if constexpr (EPI == EPI_RELU_OUT3) {
        if (tid < 3 * kF) W3l[tid] = a.w3[tid];
        else if (tid < 3 * kF + 3) W3l[tid] = a.b3[tid - 3 * kF];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int units_per_img = a.tiles_x * a.tiles_y;
    const unsigned map_bytes = (unsigned)(a.H * a.W * kF * 4);      // 48-channel maps of the input's size (partial sums, residuals)
    const unsigned out_bytes = (unsigned)(a.Hout * a.Wout * kF * 4);
    const int row_bytes = a.W * CIN * 4;

    auto locate = [&](int unit, UnitPos& u) {
        // integer division runs on the VALU: tell hipcc the results are wave-uniform, or every
        // buffer descriptor built from them is wrapped in a waterfall loop
        u.b = __builtin_amdgcn_readfirstlane(unit / units_per_img);
        const int rr = unit - u.b * units_per_img;
        const int uy = __builtin_amdgcn_readfirstlane(rr / a.tiles_x);
        const int ux = rr - uy * a.tiles_x;
        u.ty = uy * 4 + wave;
        // the column of the unit's first tile as an opaque scalar: left to itself the optimiser strength-reduces
        // 32 unit + 2 lr (+ constants) into a per-lane induction variable that lives in a vector register for the whole kernel
        u.tx = __builtin_amdgcn_readfirstlane(ux * 16) + lr;
    };
    // The 16 patch pixels of a tile: 16 buffer loads of 16 B.  One buffer descriptor PER PATCH ROW (the row index
    // is wave-uniform, so this is scalar work): base = the row's first pixel, num_records = the row's bytes, or 0
    // for a row outside the image.  The hardware range check then zero-fills whole rows above / below the image and
    // every pixel right of it (= padding 1) with no per-lane select, and no offset ever exceeds a row, whatever the
    // size of the map.  Only the tile column left of the image (x = -1) needs a lane select.
    auto load_patch = [&](f32x4 (&p)[16], const UnitPos& u, int j) {
        const float* img = a.in + (size_t)u.b * a.H * a.W * CIN;
        const int y0 = 2 * u.ty - 1, x0 = 2 * u.tx - 1;
        const unsigned v1 = (unsigned)((x0 + 1) * (CIN * 4) + (16 * j + 4 * g) * 4);      // pixel x0 + 1 >= 0
        const unsigned v0 = x0 >= 0 ? v1 - CIN * 4 : 0xffffffffu;
#pragma unroll
        for (int dy = 0; dy < 4; ++dy) {
            const int y = y0 + dy;
            const bool row_ok = (unsigned)y < (unsigned)a.H;
            __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(img + (ptrdiff_t)y * a.W * CIN), 0, row_ok ? row_bytes : 0, 0x00020000);
            p[dy * 4 + 0] = bload(r, v0);
            p[dy * 4 + 1] = bload(r, v1);
            p[dy * 4 + 2] = bload(r, v1 + CIN * 4);
            p[dy * 4 + 3] = bload(r, v1 + 2 * CIN * 4);
        }
    };

    // ---- UPS: the 3x3 low-resolution pixels (rows ty-1..ty+1, columns tx-1..tx+1, indices clamped to the map as
    // ATen's upsample_bilinear2d clamps them) behind a tile's 4x4 patch of the upsampled map
    f32x4 lo[UPS ? 9 : 1];
    auto load_lo = [&](const UnitPos& u, int j) {
        const int hl = a.H >> 1, wl = a.W >> 1;
        const float* img = a.in + (size_t)u.b * hl * wl * CIN;
        unsigned off[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int c = min(max(u.tx - 1 + d, 0), wl - 1);
            off[d] = (unsigned)(c * (CIN * 4) + 16 * g);
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            // a tile row past the end of the map (the unit after the last one: its batch index is out of range too)
            // reads nothing: zero records
            const int yi = min(max(u.ty - 1 + r, 0), hl - 1);
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(img + (ptrdiff_t)yi * wl * CIN), 0, u.ty < hl ? wl * CIN * 4 : 0, 0x00020000);
#pragma unroll
            for (int d = 0; d < 3; ++d) lo[r * 3 + d] = bload(rs, off[d], 64 * j);
        }
    };
    // Patch row pr / column pc of tile (ty, tx) is upsampled row 2 ty - 1 + pr / column 2 tx - 1 + pc:
    //   pr 0: 0.75 d[i-1] + 0.25 d[i]   (the conv's zero padding when ty = 0)        i = ty, rows of `lo`: 0 = i-1, 1 = i, 2 = i+1
    //   pr 1: 0.25 d[i-1] + 0.75 d[i]   (d[i] alone when ty = 0: the source index is clamped at 0 with weight 1)
    //   pr 2: 0.75 d[i]   + 0.25 d[i+1]
    //   pr 3: 0.25 d[i]   + 0.75 d[i+1] (zero padding when ty is the last row)
    // the same along x; vertical pass first, then horizontal, as upsample2x_kernel (prestage.hip) evaluates it, so
    // the fused conv and "upsample, then conv" see the same bits.
    auto interp = [&](f32x4 (&p)[16], const UnitPos& u) {
        const int hl = a.H >> 1, wl = a.W >> 1;
        const bool x_first = u.tx == 0, x_last = u.tx >= wl - 1;
        const float wx[4][2] = {{x_first ? 0.f : 0.75f, x_first ? 0.f : 0.25f},
                                {x_first ? 0.f : 0.25f, x_first ? 1.f : 0.75f},
                                {0.75f, 0.25f},
                                {x_last ? 0.f : 0.25f, x_last ? 0.f : 0.75f}};
        const bool y_first = u.ty == 0, y_last = u.ty >= hl - 1;
        const float wy[4][2] = {{y_first ? 0.f : 0.75f, y_first ? 0.f : 0.25f},
                                {y_first ? 0.f : 0.25f, y_first ? 1.f : 0.75f},
                                {0.75f, 0.25f},
                                {y_last ? 0.f : 0.25f, y_last ? 0.f : 0.75f}};
        // row by row, vertical pass first: three temporaries instead of twelve (the kernel has no registers to spare)
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            const int r0 = pr < 2 ? 0 : 1;
            f32x4 vt[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) vt[d] = fma4(lo[(r0 + 1) * 3 + d], wy[pr][1], lo[r0 * 3 + d] * wy[pr][0]);
#pragma unroll
            for (int pc = 0; pc < 4; ++pc) {
                const int c0 = pc < 2 ? 0 : 1;
                p[pr * 4 + pc] = fma4(vt[c0 + 1], wx[pc][1], vt[c0] * wx[pc][0]);
            }
        }
    };

    // lane-linear fragments: each ds_read_b128 lane group covers one bank row.  The bank spans 144 KiB, more than
    // the 64 KiB a ds_read offset field reaches: three opaque LDS pointers (positions 0-6, 7-13, 14-15) keep every
    // fragment address "register + immediate"; left to itself hipcc builds an address register per fragment beyond
    // 64 KiB (76 v_add_u32 per unit, and the registers to hold them).
    typedef __attribute__((address_space(3))) f32x4 lds_frag;
    lds_frag* ubp[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        ubp[k] = (lds_frag*)U + lane + k * 7 * NJ * 3 * 64;
        asm volatile("" : "+v"(ubp[k]));
    }
    f32x4 pb[2][16];      // ping-pong patch buffers (raw patch -> transformed in place)
    f32x4 acc[16][3];
    f32x4 wq[2][3];       // A fragments: the step in flight and the one after it
    auto ldsA = [&](int j, int pos, int m) {
        return ubp[pos / 7][(((pos % 7) * NJ + j) * 3 + m) * 64];
    };

    // One pipeline stage = chunk J of the current unit: 16 steps (positions) of 3 A fragments and
    // 12 MFMAs on the transformed patch pb[X], issued k-step-major / cout-block-minor so that
    // consecutive MFMAs hit three different accumulators (the 16x16x4 MFMA has a 40-cycle
    // dependent latency against a 32-cycle issue interval).  At the start of the stage the raw
    // patch of the NEXT chunk is requested into the other buffer; it is transformed in 8 slices
    // spread over steps 8..15 (VALU under the MFMAs); the fragments of step s+1 are read at step
    // s.  sched_barrier pins this order (hipcc otherwise sinks each fragment read next to its
    // first use and exposes one LDS latency per fragment).
    auto stage = [&](auto JC, auto XC, bool first, const UnitPos& ld_u, int ld_j) {
        constexpr int J = decltype(JC)::value;
        constexpr int X = decltype(XC)::value;
        constexpr int Y = 1 - X;
        if constexpr (UPS) load_lo(ld_u, ld_j);
        else load_patch(pb[Y], ld_u, ld_j);
        // Position order: the accumulators that are SEEDED instead of zeroed come last, so that the
        // loads that seed them (issued at the start of the unit) have eleven steps to land:
        //   position (1,1) <- bias:        A^T M A with only M11 = b is b on all four outputs;
        //   the four corners <- the partial sums of the first 48 input channels (two-pass 96->48
        //   convs): M00 = P00, M03 = -P01, M30 = -P10, M33 = P11 give exactly Y += P.
        // Neither costs an epilogue add.
        constexpr int ORD[16] = {1, 2, 4, 6, 7, 8, 9, 10, 11, 13, 14, 5, 0, 3, 12, 15};
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            const int pos = ORD[st];
            const bool seeded = ACC_IN ? (pos == 0 || pos == 3 || pos == 12 || pos == 15) : pos == 5;
            // the fragments of the step after this one; the last step reads the first fragments of the NEXT stage
            // (chunks follow each other 0, 1, 2, 0, ...), so that no stage opens on an exposed LDS read
#pragma unroll
            for (int m = 0; m < 3; ++m)
                wq[(st + 1) & 1][m] = st + 1 < 16 ? ldsA(J, ORD[st + 1], m) : ldsA((J + 1) % NJ, ORD[0], m);
            if constexpr (UPS) {
                if (st == 8) interp(pb[Y], ld_u);
            }
            if (st >= 8) transform_slice(pb[Y], st - 8);
            if (first && ACC_IN && (pos == 3 || pos == 12)) {
#pragma unroll
                for (int m = 0; m < 3; ++m) acc[pos][m] = -acc[pos][m];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    const f32x4 c = (first && i == 0 && !seeded) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[pos][m];
                    acc[pos][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[st & 1][m][i], pb[X][pos][i], c, 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // the bias through a buffer descriptor (scalar base + one 32-bit lane offset): as a 64-bit per-lane pointer it was
    // a loop invariant that cost a register pair for the whole kernel
    __amdgpu_buffer_rsrc_t bias_r = __builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, kF * 4, 0x00020000);
    int unit = blockIdx.x;
    UnitPos cur, nxt;
    locate(unit, cur);
    if constexpr (UPS) {
        load_lo(cur, 0);
        interp(pb[0], cur);
    } else {
        load_patch(pb[0], cur, 0);
    }
#pragma unroll
    for (int sl = 0; sl < 8; ++sl) transform_slice(pb[0], sl);
#pragma unroll
    for (int m = 0; m < 3; ++m) wq[0][m] = ldsA(0, 1, m);      // position ORD[0] of chunk 0: the first step of the first stage

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;

    // one unit; PC = buffer that holds its transformed chunk 0 (the parity flips every unit)
    auto run_unit = [&](auto PC) {
        constexpr int P = decltype(PC)::value;
        using XP = std::integral_constant<int, P>;
        using XQ = std::integral_constant<int, 1 - P>;
        locate(unit + gridDim.x, nxt);       // past the end: every pixel out of range -> zeros, stores dropped
        if (unit + (int)gridDim.x >= a.ntiles) nxt.ty = 1 << 20;
        // seeds of this unit's accumulators (see the position order in `stage`)
        if constexpr (ACC_IN) {
            __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(a.acc_in + (size_t)cur.b * a.H * a.W * kF), 0, map_bytes, 0x00020000);
            constexpr int corner[4] = {0, 3, 12, 15};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int yy = 2 * cur.ty + (q >> 1), xx = 2 * cur.tx + (q & 1);
                const unsigned o = (yy < a.H && xx < a.W) ? (unsigned)(((yy * a.W + xx) * kF + 4 * g) * 4) : 0x80000000u;
#pragma unroll
                for (int m = 0; m < 3; ++m) acc[corner[q]][m] = bload(pr, o + 64 * m);
            }
        } else {
#pragma unroll
            for (int m = 0; m < 3; ++m) acc[5][m] = bload(bias_r, (unsigned)(16 * g), 64 * m);
        }
        if constexpr (NJ == 3) {
            stage(I0{}, XP{}, true, cur, 1);
            stage(I1{}, XQ{}, false, cur, 2);
            stage(I2{}, XP{}, false, nxt, 0);
        } else {
            stage(I0{}, XP{}, true, nxt, 0);     // one chunk per unit: the next unit's patch is the one in flight
        }

        // ---- output transform A^T M A, epilogue, stores
        __amdgpu_buffer_rsrc_t orr = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(a.out + (size_t)cur.b * a.Hout * a.Wout * kF), 0, out_bytes, 0x00020000);
        __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(
            (void*)((EPI == EPI_RELU_ADD2 ? a.res1 : a.out) + (size_t)cur.b * a.H * a.W * kF), 0, map_bytes, 0x00020000);
        __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(
            (void*)((EPI == EPI_RELU_ADD2 ? a.res2 : a.out) + (size_t)cur.b * a.H * a.W * kF), 0, map_bytes, 0x00020000);
        const int oy = 2 * cur.ty, ox = 2 * cur.tx;
        unsigned po[4], so[4];     // per output pixel q = 2*row+col: offsets in the input-size map / in `out`
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int yy = oy + (q >> 1), xx = ox + (q & 1);
            const bool ok = yy < a.H && xx < a.W;
            po[q] = ok ? (unsigned)(((yy * a.W + xx) * kF + 4 * g) * 4) : 0x80000000u;
            so[q] = ok ? (unsigned)((((yy + a.oy) * a.Wout + xx + a.ox) * kF + 4 * g) * 4) : 0x80000000u;
        }
        float o3[4][3];      // EPI_RELU_OUT3: this lane's share of the 1x1 conv, per output pixel q and channel
        if constexpr (EPI == EPI_RELU_OUT3) {
#pragma unroll
            for (int q = 0; q < 4; ++q) o3[q][0] = o3[q][1] = o3[q][2] = 0.f;
        }
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            f32x4 ra[4], rb[4];
            if constexpr (EPI == EPI_RELU_ADD2) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    ra[q] = bload(r1, po[q] + 64 * m);
                    rb[q] = bload(r2, po[q] + 64 * m);
                }
            }
            f32x4 s0[4], s1[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                s0[x] = (acc[x][m] + acc[4 + x][m]) + acc[8 + x][m];
                s1[x] = sub4(sub4(acc[4 + x][m], acc[8 + x][m]), acc[12 + x][m]);
            }
            f32x4 y[4];   // q = 2*row + col
            y[0] = (s0[0] + s0[1]) + s0[2];
            y[1] = sub4(sub4(s0[1], s0[2]), s0[3]);
            y[2] = (s1[0] + s1[1]) + s1[2];
            y[3] = sub4(sub4(s1[1], s1[2]), s1[3]);
            if constexpr (EPI == EPI_POOL) {
                // MaxPool2d(2) of the un-activated conv output = max over the tile's 2x2 pixels
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(fmaxf(y[0][r], y[1][r]), fmaxf(y[2][r], y[3][r]));
                const bool ok = cur.ty < a.Hout && cur.tx < a.Wout;
                bstore(orr, ok ? (unsigned)(((cur.ty * a.Wout + cur.tx) * kF + 16 * m + 4 * g) * 4) : 0x80000000u, v);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v = y[q];
                    if constexpr (EPI == EPI_RELU || EPI == EPI_RELU_ADD2 || EPI == EPI_RELU_OUT3) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                    }
                    if constexpr (EPI == EPI_RELU_ADD2) v = (ra[q] + rb[q]) + v;   // e3 + d1 + d2 (unet.py:563-566)
                    bstore(orr, so[q] + 64 * m, v);
                    if constexpr (EPI == EPI_RELU_OUT3) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const f32x4 w = *reinterpret_cast<const f32x4*>(W3l + c * kF + 16 * m + 4 * g);
                            o3[q][c] += (v[0] * w[0] + v[1] * w[1]) + (v[2] * w[2] + v[3] * w[3]);
                        }
                    }
                }
            }
        }
        if constexpr (EPI == EPI_RELU_OUT3) {
            // sum the four channel groups (lanes l, l^16, l^32, l^48), add the bias, lane group 0 stores
            const size_t hw = (size_t)a.H * a.W;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float t[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float v = o3[q][c];
                    v += __shfl_xor(v, 16);
                    v += __shfl_xor(v, 32);
                    t[c] = v + W3l[3 * kF + c];
                }
                const int yy = oy + (q >> 1), xx = ox + (q & 1);
                if (g == 0 && yy < a.H && xx < a.W) {
                    const size_t pidx = (size_t)yy * a.W + xx;
#pragma unroll
                    for (int c = 0; c < 3; ++c) a.out3_nchw[((size_t)cur.b * 3 + c) * hw + pidx] = t[c];
                    if (a.out3_nhwc4)
                        reinterpret_cast<f32x4*>(a.out3_nhwc4)[(size_t)cur.b * hw + pidx] = f32x4{t[0], t[1], t[2], 0.f};
                }
            }
        }
        cur = nxt;
    };

#pragma unroll 1
    for (;;) {
        run_unit(I0{});
        unit += gridDim.x;
        if (unit >= a.ntiles) break;
        run_unit(I1{});
        unit += gridDim.x;
        if (unit >= a.ntiles) break;
    }
}

template <int EPI, bool ACC_IN>
__global__ __launch_bounds__(256, 1) void wino3x3_kernel(ConvArgs a) {
    wino_body<EPI, ACC_IN, 3>(a);
}
// the first layer: 16-channel (zero-padded 6 / 9) network input
template <int EPI>
__global__ __launch_bounds__(256, 1) void wino3x3_c16_kernel(ConvArgs a) {
    wino_body<EPI, false, 1>(a);
}
// UpConv (networks/unet.py:88-147): bilinear x2 upsample fused into the patch load, conv, ReLU
template <int EPI>
__global__ __launch_bounds__(256, 1) void wino3x3_ups_kernel(ConvArgs a) {
    wino_body<EPI, false, 3, true>(a);
}

template <int EPI, bool ACC_IN, int NJ, bool UPS = false>
hipError_t launch_w(const ConvArgs& a0, hipStream_t s) {
    static std::atomic<uint64_t> attr_done{0};
    void (*kern)(ConvArgs);
    if constexpr (UPS) kern = wino3x3_ups_kernel<EPI>;
    else if constexpr (NJ == 3) kern = wino3x3_kernel<EPI, ACC_IN>;
    else kern = wino3x3_c16_kernel<EPI>;
    constexpr size_t U_LDS_BYTES = (size_t)u_floats(NJ) * 4 + (EPI == EPI_RELU_OUT3 ? (3 * kF + 4) * 4 : 0);
    if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(kern), U_LDS_BYTES, attr_done); e != hipSuccess)
        return e;
    ConvArgs a = a0;
    a.tiles_x = (a.W + 31) / 32;     // unit = 4 tile rows x 16 tile columns = 8 x 32 output pixels
    a.tiles_y = (a.H + 7) / 8;
    a.ntiles = a.B * a.tiles_x * a.tiles_y;
    const int cus = current_device_cus();
    const int grid = a.ntiles < cus ? a.ntiles : cus;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), U_LDS_BYTES, s, a);
    return hipGetLastError();
}

}  // namespace

size_t wino3x3_weight_floats() { return U_FLOATS; }

hipError_t launch_wino3x3(const ConvArgs& a, int cin, int epi, hipStream_t s) {
    if (a.ntiles <= 0) return hipSuccess;
    if (cin == 16) {
        if (a.acc_in || (size_t)a.Hout * a.Wout * kF * 4 >= 0x80000000ull) return hipErrorInvalidValue;
        if (epi == EPI_NONE) return launch_w<EPI_NONE, false, 1>(a, s);
        if (epi == EPI_RELU) return launch_w<EPI_RELU, false, 1>(a, s);
        return hipErrorInvalidValue;
    }
    if (cin != 48) return hipErrorInvalidValue;
    if (a.ups) {       // a.in = the map to upsample, [B][H/2][W/2][48]
        if (a.acc_in || epi != EPI_RELU || (a.H & 1) || (a.W & 1)) return hipErrorInvalidValue;
        if ((size_t)a.Hout * a.Wout * kF * 4 >= 0x80000000ull) return hipErrorInvalidValue;
        return launch_w<EPI_RELU, false, 3, true>(a, s);
    }
    // stores and the side inputs (partial sums, residuals) address their map with one 32-bit byte offset whose
    // out-of-image sentinel is 2^31; the conv input is addressed per row and has no such limit
    if ((size_t)a.H * a.W * kF * 4 >= 0x80000000ull || (size_t)a.Hout * a.Wout * kF * 4 >= 0x80000000ull)
        return hipErrorInvalidValue;
    const bool acc = a.acc_in != nullptr;
    switch (epi) {
        case EPI_NONE:
            return acc ? launch_w<EPI_NONE, true, 3>(a, s) : launch_w<EPI_NONE, false, 3>(a, s);
        case EPI_RELU:
            return acc ? launch_w<EPI_RELU, true, 3>(a, s) : launch_w<EPI_RELU, false, 3>(a, s);
        case EPI_POOL:
            return acc ? hipErrorInvalidValue : launch_w<EPI_POOL, false, 3>(a, s);
        case EPI_RELU_ADD2:
            return acc ? hipErrorInvalidValue : launch_w<EPI_RELU_ADD2, false, 3>(a, s);
        case EPI_RELU_OUT3:
            return acc ? hipErrorInvalidValue : launch_w<EPI_RELU_OUT3, false, 3>(a, s);
    }
    return hipErrorInvalidValue;
}
