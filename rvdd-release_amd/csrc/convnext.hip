// ConvNeXt ConvBlock (networks/new_unet.py:74-103) on gfx950:
//   x -> [proj 1x1 when Cin != 48] -> r = dwconv7x7 -> LayerNorm_C -> 1x1 48->192
//     -> GELU(erf) -> 1x1 192->48 ; out = x + layerscale * r
//
// Kernels, all NHWC fp32:
//   proj1x1_kernel        : 1x1 projection (9|6 -> 48 from the padded 16-channel network input, or 96 -> 48 from two
//                           48-channel maps = virtual concat) on f32 MFMA with the whole weight matrix held in registers
//                           (levels where the projection halves cannot ride in the blocks' epilogues: odd sizes).
//   convblock_pipe_kernel : the whole block per 16x16-pixel tile as a two-stage pipeline inside the workgroup (the default):
//                           depth-wise 7x7 + LayerNorm (biased variance, eps 1e-6, :12-28) of tile t + 1 on four waves beside the
//                           MLP of tile t on the other four -- both 1x1 convs on the F16 matrix pipe with split f32 operands in the
//                           orientation D[channel][pixel]: the accumulators of the first GEMM (lane = pixel, 4 consecutive hidden
//                           channels per register quad) ARE the B fragments of the second one, so the 192-channel hidden map never
//                           leaves the register file (the reference writes and re-reads it: 708 MB per block at 720p).
//   convblock_kernel      : the same block with its phases one after the other in all eight waves: SPLIT = false is the
//                           exact-f32-product form (f32 MFMA; option next_split 0, and blocks whose weights break the split's
//                           bound), SPLIT = true the same-bits sibling of the pipelined kernel (option next_pipe 0).
// (The two-kernel form of rounds 1-2, dwln_kernel + mlp_kernel, was retired in round 6; LABBOOK.md 4.3 describes it.)
//
// Lane <-> data map of the MLP (the MFMA B-operand map): lane l owns pixel l&15 of its 16-pixel group and, in every
// 16-channel chunk j, channels 16j+4g..+3 with g = l>>4.
#include "rvdd_internal.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace {

typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, float* lds_wave_base, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_wave_base, 16, voff, 0, 0, 0);
}

// --------------------------------------------------------------- proj 1x1 --
// w arranged [j][m][lr][g][i] = W[16m+lr][c(j,g,i)], j over the chunks of in1 then in2.
template <int C1, int C2>
__global__ __launch_bounds__(256) void proj1x1_kernel(const float* __restrict__ in1,
                                                      const float* __restrict__ in2,
                                                      const float* __restrict__ w,
                                                      const float* __restrict__ bias,
                                                      float* __restrict__ out, long npix) {
    constexpr int NJ1 = C1 / 16, NJ2 = C2 / 16, NJ = NJ1 + NJ2;
    const int lane = threadIdx.x & 63;
    const int lr = lane & 15, g = lane >> 4;
    f32x4 wa[NJ][3];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int m = 0; m < 3; ++m)
            wa[j][m] = *reinterpret_cast<const f32x4*>(w + (((size_t)(j * 3 + m) * 16 + lr) * 4 + g) * 4);
    f32x4 bv[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) bv[m] = *reinterpret_cast<const f32x4*>(bias + 16 * m + 4 * g);

    const long ngroups = (npix + 15) / 16;
    const long wave_global = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    for (long grp = wave_global; grp < ngroups; grp += nwaves) {
        const long pix = grp * 16 + lr;
        const long pc = pix < npix ? pix : npix - 1;
        f32x4 xb[NJ];
#pragma unroll
        for (int j = 0; j < NJ1; ++j) xb[j] = *reinterpret_cast<const f32x4*>(in1 + pc * C1 + 16 * j + 4 * g);
#pragma unroll
        for (int j = 0; j < NJ2; ++j)
            xb[NJ1 + j] = *reinterpret_cast<const f32x4*>(in2 + pc * C2 + 16 * j + 4 * g);
        f32x4 acc[3] = {bv[0], bv[1], bv[2]};
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int m = 0; m < 3; ++m)
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[j][m][i], xb[j][i], acc[m], 0, 0, 0);
        if (pix < npix) {
#pragma unroll
            for (int m = 0; m < 3; ++m) *reinterpret_cast<f32x4*>(out + pix * kF + 16 * m + 4 * g) = acc[m];
        }
    }
}

// ------------------------------------------------------------ dw 7x7 + LN --
constexpr int D_W_FLOATS = 49 * kF;                       // 2352: the 7x7 taps of the 48 channels, tap-major

// ------------------------------------------- dw 7x7 + LN, sliding windows --
// A pixel-per-thread kernel (round 1) reads one LDS value and one LDS weight per four FMAs and sits on the LDS
// bandwidth (294 ds_read_b128 per thread).  Here a thread owns FOUR adjacent pixels of a row and four channels:
// per filter row it reads the 10 input values its four 7-tap windows cover and the 7 weights once (17 reads for
// 28 float4 FMAs, 3.3x fewer).  To keep two workgroups per CU with a 16x16-pixel tile (halo 22x22: 1.9x the
// tile instead of 2.4x) the halo tile goes through LDS ONE 16-CHANNEL CHUNK AT A TIME (33 KiB), double-buffered:
// the DMA of chunk j+1 runs under the FMAs of chunk j; the 12 channels x 4 pixels a thread accumulates stay in
// registers until the LayerNorm (two __shfl_xor over the four channel-group lanes of a pixel).
//
// LDS image of a chunk: pixel pitch 64 B, row pitch 24 pixels, the four pixels of every aligned group of four
// XOR-swizzled by the group index (slot = ((p & 3) ^ (R & 3)) * 4 + g, R = p >> 2) -- applied on the SOURCE
// address of the LDS-DMA, whose destination is lane-linear.  With the lane map below (the 16 lanes that one
// ds_read_b128 lane group serves = 4 quads x 4 channel groups of ONE row) every read is conflict-free.
constexpr int E_TH = 16, E_TW = 16, E_IH = E_TH + 6, E_PITCH = 24;
constexpr int E_BUF_FLOATS = E_IH * E_PITCH * 16;          // 8448 floats = 33 KiB: one 16-channel chunk of the halo tile
constexpr int E_PIECES = E_BUF_FLOATS / 256;                // 33 LDS-DMA pieces of 1 KiB
constexpr int E_PAR_FLOATS = 3 * kF;                         // dw bias | LayerNorm weight | LayerNorm bias

// (dwln_kernel, the depth-wise + LayerNorm kernel of the two-kernel ConvBlock these constants were first written for, was retired in
// round 6 with mlp_kernel: the fused kernels below share its tile geometry, LDS image and lane map.  LABBOOK.md 4.3.)

// LDS: fc1 arranged [j(3)][m(12)][lane = 16g+lr][i] = W1[16m+lr][16j+4g+i]     (9216 floats)
//      fc2 arranged [m(12)][mo(3)][lane = 16g+lr][r] = W2[16mo+lr][16m+4g+r]   (9216 floats)
//      fc1_b (192) | fc2_b (48) | layerscale (48)
constexpr int M_W_FLOATS = 192 * 48;
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Two waves per SIMD (NW = 8 per workgroup, one workgroup per CU), each covering its own latencies.  Round 1's
// kernel leant on three waves per SIMD to hide its loads, but on gfx950 the f32 MFMA and the VALU share the SIMD's
// fp32 lanes, and a VALU instruction issued by ANOTHER wave among f32 MFMAs costs 7-40 cycles against 4.25 from the
// wave's own stream (profiles/r02_mfma_valu_microbench.log): the GELUs of one wave stalled the MFMAs of the other
// two.  Here a wave software-pipelines itself: the LayerNorm rows of its NEXT pixel group are requested before
// fc1, the residual rows before fc2, weight fragments and biases are read from LDS one step ahead (across the
// phases too), and the GELU of a pair of hidden blocks runs after the first MFMAs of the next pair.
// SQ counters (profiles/r02_k_mlp_*): the wave is never idle - MFMA 76 %, GELU 14 %, waits 10 % of its cycles.
typedef __attribute__((address_space(3))) f32x4 lds_frag;
constexpr int M2_BV_FLOATS = 192 + 48 + 48;                       // fc1_b | fc2_b | layerscale

__device__ __forceinline__ f32x4 bload(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void bstore(__amdgpu_buffer_rsrc_t r, unsigned off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, off, 0, 0);
}

// GELU(v) = v Phi(v) = max(v, 0) - |v| Phi(-|v|), Phi(-t) = erfc(t / sqrt2) / 2 = 2^(t P(t) - 1): P of degree 5,
// a weighted minimax fit (host, Lawson iterations) of log2 erfc(t / sqrt2) / t on [0, 6.36] (beyond it the term is
// below 1e-9).  max |error| 3.1e-7 on [-9, 9] in this float32 evaluation against the double-precision erf form;
// 8 scalar (two of them v_exp_f32) + 6 packed instructions per two values; round 1's 0.5 v (1 + erf) form with a
// degree-7 polynomial took 6 + 11.
__device__ __forceinline__ f32x4 gelu_phi4(f32x4 v) {
    // two packed Horner chains side by side: a v_pk_fma_f32 that reads the result of the one before it costs a
    // wait state
    const float cap = 6.36f, zero = 0.0f;
    f32x2 t[2], p[2], q[2];
    f32x4 relu, o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        // |.| as a source modifier, and no canonicalising v_max in front (fminf / fmaxf emit one)
        asm("v_min_f32 %0, |%1|, %2" : "=v"(t[k >> 1][k & 1]) : "v"(v[k]), "v"(cap));
        asm("v_max_f32 %0, %1, %2" : "=v"(relu[k]) : "v"(v[k]), "v"(zero));
    }
    auto K = [](float c) { return f32x2{c, c}; };
    constexpr float C[6] = {2.992418740177527e-05f, -0.0007398742018267512f, 0.007977462373673916f,
                            -0.05323818698525429f, -0.45891568064689636f, -1.1511471271514893f};
    p[0] = p[1] = K(C[0]);
#pragma unroll
    for (int i = 1; i < 6; ++i) {
        p[0] = __builtin_elementwise_fma(p[0], t[0], K(C[i]));
        p[1] = __builtin_elementwise_fma(p[1], t[1], K(C[i]));
    }
    q[0] = __builtin_elementwise_fma(t[0], p[0], K(-1.0f));
    q[1] = __builtin_elementwise_fma(t[1], p[1], K(-1.0f));
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = fmaf(-fabsf(v[k]), __builtin_amdgcn_exp2f(q[k >> 1][k & 1]), relu[k]);
    return o;
}

// The same for an argument known only up to a power of two, v = s h (h: fc1's accumulator in the filters' scale, s = 2^-k):
// returns GELU(v) / s from h.  With t' = min(|h|, cap / s), q = t' P'(t') - 1, P' = P's Horner chain on the coefficients
// C_i s^(6-i): every intermediate is the unscaled chain's times a power of two, so the result is gelu_phi4(s h) / s bit for
// bit -- and the 48 multiplications per pixel group that formed s h are gone.  gc[0..5] = the scaled coefficients,
// gc[6] = cap / s (NextBlockW::gelu_c, computed on the host; every constant twice, see there).
__device__ __forceinline__ f32x4 gelu_phi4_scaled(f32x4 h, const float (&gc)[7][2]) {
    const float zero = 0.0f, cap = gc[6][0];
    f32x2 t[2], p[2], q[2], relu[2], e[2];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        // v_med3_f32 through the intrinsic, NOT gelu_phi4's inline-asm v_min / v_max: h comes straight out of an MFMA here,
        // and the wait states a vector read of an MFMA result needs are only inserted in front of instructions the compiler
        // can see (inline asm reading the accumulator two cycles behind the MFMA read garbage)
        t[k >> 1][k & 1] = __builtin_amdgcn_fmed3f(__builtin_fabsf(h[k]), zero, cap);
    }
    {   // max(h, 0) on the BITS as signed integers: a float below zero (or -0) is a negative integer.  One v_max_i32; the float
        // forms (fmaxf, fmed3f with an infinity) cost a second instruction each, a canonicalising v_max_f32 h, h in front
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        const f32x4 r4 = __builtin_bit_cast(f32x4, __builtin_elementwise_max(__builtin_bit_cast(i32x4, h), i32x4{0, 0, 0, 0}));
        relu[0] = f32x2{r4[0], r4[1]};
        relu[1] = f32x2{r4[2], r4[3]};
    }
    auto K = [&](int i) { return f32x2{gc[i][0], gc[i][1]}; };
    p[0] = p[1] = K(0);
#pragma unroll
    for (int i = 1; i < 6; ++i) {
        p[0] = __builtin_elementwise_fma(p[0], t[0], K(i));
        p[1] = __builtin_elementwise_fma(p[1], t[1], K(i));
    }
    q[0] = __builtin_elementwise_fma(t[0], p[0], f32x2{-1.0f, -1.0f});
    q[1] = __builtin_elementwise_fma(t[1], p[1], f32x2{-1.0f, -1.0f});
#pragma unroll
    for (int k = 0; k < 4; ++k) e[k >> 1][k & 1] = __builtin_amdgcn_exp2f(q[k >> 1][k & 1]);
    // max(h, 0) - t Phi(-t) with the capped t: beyond the cap the product is below 1e-9 either way; packed
    const f32x2 o0 = __builtin_elementwise_fma(-t[0], e[0], relu[0]), o1 = __builtin_elementwise_fma(-t[1], e[1], relu[1]);
    return f32x4{o0[0], o0[1], o1[0], o1[1]};
}

// OUT3: the network's last block also applies the 1x1 conv 48 -> 3 of the post-processing (new_unet.py:414-430) to the
// map it has just formed, instead of a kernel that reads the 48-channel map back: each lane multiplies its 12 channels,
// two shuffles sum the four channel groups of a pixel, lane group 0 stores the planar frame and the NHWC4 copy.
struct Out3 {
    const float* w;        // [3][48]
    const float* b;        // [3]
    float* nchw;           // [B][3][hw] or null
    float* nhwc4;          // [B][hw][4] or null
    int hw;                // pixels per image
};
// ------------------------------------- the whole ConvBlock in one kernel --
// convblock_kernel = dwln_kernel's depth-wise phase + LayerNorm, then mlp_kernel's two GEMMs, per 16x16-pixel tile,
// in ONE persistent workgroup per CU (networks/new_unet.py:74-103).  What the fusion removes:
//   * the LayerNorm map: written (192 B/px) by one kernel and read back by the other, per block;
//   * the depth-wise kernel's exposed LDS-DMA latency: the two chunks a tile starts with are requested BEFORE the
//     MLP phase of the tile in front of it and land under its ~45 k cycles of MFMAs; only the third chunk of a tile is
//     requested inside the depth-wise phase (under the FMAs of the second);
//   * one kernel boundary per block (25 per frame-step).
// What it cannot remove: on gfx950 the f32 MFMA executes on the SIMD's fp32 lanes, so the depth-wise FMAs (13 % of
// the block's flops) cost their full VALU time beside the MFMAs -- they are now the ONLY thing the phase waits for.
//
// LDS (149.5 KiB, one workgroup of four waves per CU): fc1 | fc2 (72 KiB) | biases, layerscale, 1x1 (1.7 KiB) |
// depth-wise taps (9.2 KiB) | dw bias, LN weight, LN bias | two 33-KiB halo chunk buffers.  The LayerNorm result
// changes hands through the chunk buffers, which are dead by then: the depth-wise phase leaves a lane with 4 adjacent
// pixels x 4 channels per chunk (convnext.hip dwln lane map), the GEMMs want pixel = lane & 15, channels 4 (lane >> 4)..
// (the MFMA B-operand map).  A wave's four tile rows are its own four 16-pixel groups, so the exchange is wave-private:
// [row 4][chunk 3][pixel 16][16 floats], the four 16-byte slots of a pixel XOR-swizzled by (pixel >> 2) ^ sigma(group)
// with sigma = {0, 3, 1, 2}: every ds_read_b128 lane group {0-3, 12-15, 20-27}, ... then covers 16 distinct slots.
constexpr int F_BV_FLOATS = M2_BV_FLOATS + 148;                           // fc1_b | fc2_b | layerscale | w3 [3][48] | b3 (+pad)
constexpr int F_OFF_W2 = M_W_FLOATS;
constexpr int F_OFF_BV = 2 * M_W_FLOATS;
constexpr int F_OFF_DW = F_OFF_BV + F_BV_FLOATS;
constexpr int F_OFF_PAR = F_OFF_DW + D_W_FLOATS;
constexpr int F_OFF_T = F_OFF_PAR + E_PAR_FLOATS;
constexpr size_t F_LDS_BYTES = (size_t)(F_OFF_T + 2 * E_BUF_FLOATS) * 4;  // 153,040 B
static_assert(F_LDS_BYTES <= 160 * 1024 && (F_OFF_T % 4) == 0 && 4 * 3072 <= 2 * E_BUF_FLOATS, "LDS plan");
// SPLIT (the default): the two 1x1 convs on the F16 matrix pipe with split f32 operands, as conv3x3h.hip (x = hi + lo in
// f16, a product = hi.hi + hi.lo + lo.hi, f32 accumulation; filters scaled by a power of two before the split).  The lane
// map of the f32 form stays: a lane's 4-channel blocks ARE halves of an F16 MFMA's 8-deep k group, so the operands are
// register pairs side by side and the k order is whatever the host arranged the filter fragments for:
//   fc1, per 16 hidden units m, lane blocks x0 x1 x2 (channels 16j + 4kk ..+3):
//        [wh0 wh1].[xh0 xh1] + [wh0 wh1].[xl0 xl1] + [wl0 wl1].[xh0 xh1] + [wh2 wl2].[xh2 xh2] + [wh2 wl2].[xl2 xl2]
//        = 5 MFMAs on fragments Fa Fb Fc (1 KiB each): 3 KiB per m, 36 KiB.  (The last one carries wl2.xl2 along, a
//        lo.lo term: it fills the half k group that wh2.xl2 leaves, with the fragment the fourth already holds.  A
//        half-deep v_mfma_f32_16x16x16_f16 in its place gave NaNs: hipcc 7.2 schedules the 16x16x32 that reads its
//        result as accumulator two instructions behind it, which the hardware does not interlock.)
//   fc2, per pair of hidden blocks (2p, 2p+1) and 16 outputs mo:
//        [wh wh'].[hh hh'] + [wh wh'].[hl hl'] + [wl wl'].[hh hh'] = 3 MFMAs, fragments Gh Gl: 36 KiB
// 114 F16 MFMAs of 16 cycles per 16 pixels in place of 288 f32 ones of 32, and they leave the vector port free half of
// their time: the phase is bound by its GELU, no longer by the matrix pipe.
constexpr int F_W1H_BYTES = 12 * 3072, F_W2H_BYTES = 6 * 3 * 2 * 1024;
constexpr int F_SPLIT_SHIFT = (F_W1H_BYTES + F_W2H_BYTES) / 4 - 2 * M_W_FLOATS;      // floats the regions behind the filters move by
constexpr size_t F_LDS_BYTES_SPLIT = F_LDS_BYTES + (size_t)F_SPLIT_SHIFT * 4;           // the same 153,040 B
static_assert(F_LDS_BYTES_SPLIT <= 160 * 1024 && (F_SPLIT_SHIFT % 4) == 0 && F_SPLIT_SHIFT >= 0, "LDS plan (split)");
constexpr int P_W_BYTES = 3 * 3072;           // convblock_pipe_kernel PROJ: a 48 x 48 projection's fragments, fc1's arrangement with three row blocks
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef _Float16 h4v __attribute__((ext_vector_type(4)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef __fp16 fp16x2v __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));
// x (four f32) -> hi, lo (four f16 each): hi toward zero (saturating, never an infinity), lo = x - hi to nearest
__device__ __forceinline__ void split4h(f32x4 x, u32x2v& hi, u32x2v& lo) {
    const fp16x2v h01 = __builtin_amdgcn_cvt_pkrtz(x[0], x[1]);
    const fp16x2v h23 = __builtin_amdgcn_cvt_pkrtz(x[2], x[3]);
    const unsigned u01 = __builtin_bit_cast(unsigned, h01), u23 = __builtin_bit_cast(unsigned, h23);
    // x - hi in ONE instruction per value: v_fma_mix_f32 reads the f16 half in place (hipcc's own choice for the same
    // expression is two conversions and a packed subtraction per pair: 16 cycles of the vector port against 8)
    float r0, r1, r2, r3;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(u01), "v"(x[0]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(u01), "v"(x[1]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r2) : "v"(u23), "v"(x[2]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r3) : "v"(u23), "v"(x[3]));
    const h2v l01 = {(_Float16)r0, (_Float16)r1};
    const h2v l23 = {(_Float16)r2, (_Float16)r3};
    hi = u32x2v{u01, u23};
    lo = u32x2v{__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23)};
}
__device__ __forceinline__ h8v cat8(u32x2v a, u32x2v b) { return __builtin_bit_cast(h8v, u32x4{a[0], a[1], b[0], b[1]}); }

// gelu_phi4_scaled + split4h in six stages (the same instructions on the same values, hence the same bits), so that a wave can deal
// them out between the MFMAs of ANOTHER pixel group (convblock_pipe_kernel's back waves): 8 | 4 | 4 | 4 | 4 | 10 instructions.
struct GeluStages {
    f32x2 t[2], r[2], p[2];
};
template <int S>
__device__ __forceinline__ void gelu_stage(GeluStages& g, const f32x4& h, const float (&gc)[7][2], u32x2v& hi, u32x2v& lo) {
    auto K = [&](int i) { return f32x2{gc[i][0], gc[i][1]}; };
    if constexpr (S == 0) {
        const float zero = 0.0f, cap = gc[6][0];
#pragma unroll
        for (int k = 0; k < 4; ++k) g.t[k >> 1][k & 1] = __builtin_amdgcn_fmed3f(__builtin_fabsf(h[k]), zero, cap);
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        const f32x4 r4 = __builtin_bit_cast(f32x4, __builtin_elementwise_max(__builtin_bit_cast(i32x4, h), i32x4{0, 0, 0, 0}));
        g.r[0] = f32x2{r4[0], r4[1]};
        g.r[1] = f32x2{r4[2], r4[3]};
    } else if constexpr (S == 1) {
#pragma unroll
        for (int c = 0; c < 2; ++c) g.p[c] = __builtin_elementwise_fma(K(0), g.t[c], K(1));
#pragma unroll
        for (int c = 0; c < 2; ++c) g.p[c] = __builtin_elementwise_fma(g.p[c], g.t[c], K(2));
    } else if constexpr (S == 2) {
#pragma unroll
        for (int c = 0; c < 2; ++c) g.p[c] = __builtin_elementwise_fma(g.p[c], g.t[c], K(3));
#pragma unroll
        for (int c = 0; c < 2; ++c) g.p[c] = __builtin_elementwise_fma(g.p[c], g.t[c], K(4));
    } else if constexpr (S == 3) {
#pragma unroll
        for (int c = 0; c < 2; ++c) g.p[c] = __builtin_elementwise_fma(g.p[c], g.t[c], K(5));
#pragma unroll
        for (int c = 0; c < 2; ++c) g.p[c] = __builtin_elementwise_fma(g.t[c], g.p[c], f32x2{-1.0f, -1.0f});
    } else if constexpr (S == 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) g.p[k >> 1][k & 1] = __builtin_amdgcn_exp2f(g.p[k >> 1][k & 1]);
    } else {
        const f32x2 o0 = __builtin_elementwise_fma(-g.t[0], g.p[0], g.r[0]), o1 = __builtin_elementwise_fma(-g.t[1], g.p[1], g.r[1]);
        split4h(f32x4{o0[0], o0[1], o1[0], o1[1]}, hi, lo);
    }
}

// Vector-memory instructions EVERY wave issues in the MLP phase of a tile, i.e. after the LDS-DMA of the next tile's
// first two chunks: per 16-pixel group 3 residual loads + 3 stores through buffer descriptors (issued whether the row
// exists or not: a missing row has zero records).  vmcnt retires in issue order, so "all but the newest 24" covers the
// DMA; the conditional stores of the OUT3 epilogue are not counted -- a lower bound only makes the wait conservative.
constexpr int F_MLP_VMEM = 2 * 6;

// POOL: the block also writes MaxPool2d(2) of its output (DownConv = MaxPool2d then ConvBlock, new_unet.py:200-204: the
// block in front of a DownConv feeds the pool): a wave's two tile rows are one pooling row pair, horizontal neighbours
// are lanes lr and lr ^ 1 -- one max between the two groups' outputs, one DPP max, even lanes store; no maxpool kernel.
template <bool OUT3, bool POOL = false, bool SPLIT = true>
__global__ __launch_bounds__(512, 2) void convblock_kernel(const float* __restrict__ x, NextBlockW wt, float* __restrict__ out,
                                                          int B, int H, int W, int tiles_x, int tiles_y, int ntiles, Out3 o3,
                                                          float* __restrict__ pool) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int SH = SPLIT ? F_SPLIT_SHIFT : 0;
    float* W1 = smem;
    float* W2 = smem + (SPLIT ? F_W1H_BYTES / 4 : F_OFF_W2);
    float* BV = smem + F_OFF_BV + SH;
    float* Wl = smem + F_OFF_DW + SH;       // [49][48]
    float* Pl = smem + F_OFF_PAR + SH;      // dw_b | ln_w | ln_b
    float* Tl = smem + F_OFF_T + SH;        // two chunk buffers; the LayerNorm exchange between the phases
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- tiles: as dwln_kernel (persistent, the workgroups of one XCD walk one contiguous band)
    const int per_xcd = (ntiles + 7) >> 3;
    const int band_end = min(((int)(blockIdx.x & 7) + 1) * per_xcd, ntiles);
    const int stride = (int)(gridDim.x >> 3);
    int tile = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (tile >= band_end) return;
    const int tiles_per_img = tiles_x * tiles_y;
    struct TilePos { int b, y0, x0; };
    auto locate = [&](int t) {
        TilePos p;
        p.b = t / tiles_per_img;
        const int rr = t - p.b * tiles_per_img;
        const int ty = rr / tiles_x;
        p.y0 = ty * E_TH;
        p.x0 = (rr - ty * tiles_x) * E_TW;
        return p;
    };

    // ---- depth-wise phase: lane -> (row of the wave's four rows, quad of four pixels, channel group), see dwln_kernel
    const int g = lane & 3;
    const int idx = lane >> 2;
    const int quad = idx & 3;
    const int rw = ((idx >> 3) << 1) + (__builtin_popcount(idx & 7) & 1);      // row inside the wave's four
    const int row = (wave & 3) * 4 + rw;      // waves 0-3 run the depth-wise phase; 4-7 wait at its barriers
    int piece_yx[5];           // the 33 LDS-DMA pieces of a chunk are issued by all eight waves
#pragma unroll
    for (int n = 0; n < 5; ++n) {
        const int k = wave + 8 * n;
        const int R = k * 4 + (lane >> 4), sl = lane & 15;
        const int p = 4 * R + ((sl >> 2) ^ (R & 3));
        const int iy = p / E_PITCH, ix = p - iy * E_PITCH;
        piece_yx[n] = (k < E_PIECES && ix < E_TW + 6) ? (iy << 8) | ix : -1;
    }
    auto dma_chunk = [&](const TilePos& tp, int j, int buf) {
        __amdgpu_buffer_rsrc_t ir =
            __builtin_amdgcn_make_buffer_rsrc((void*)(x + (size_t)tp.b * H * W * kF), 0, H * W * kF * 4, 0x00020000);
        float* dst = Tl + buf * E_BUF_FLOATS;
#pragma unroll
        for (int n = 0; n < 5; ++n) {
            const int k = wave + 8 * n;
            if (k < E_PIECES) {
                const int gy = tp.y0 - 3 + (piece_yx[n] >> 8), gx = tp.x0 - 3 + (piece_yx[n] & 255);
                const bool ok = piece_yx[n] >= 0 && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
                dma16(ir, dst + k * 256, ok ? (unsigned)(((gy * W + gx) * kF + 16 * j + 4 * (lane & 3)) * 4) : 0x80000000u);
            }
        }
    };
    TilePos cur = locate(tile);
    dma_chunk(cur, 0, 0);
    dma_chunk(cur, 1, 1);
    {   // weights of the block -> LDS, once per workgroup
        if constexpr (SPLIT) {
            __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)wt.fc1_h, 0, F_W1H_BYTES, 0x00020000);
            __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void*)wt.fc2_h, 0, F_W2H_BYTES, 0x00020000);
            for (int k = wave; k < F_W1H_BYTES / 1024; k += 8) dma16(r1, W1 + k * 256, (unsigned)(k * 1024 + lane * 16));
            for (int k = wave; k < F_W2H_BYTES / 1024; k += 8) dma16(r2, W2 + k * 256, (unsigned)(k * 1024 + lane * 16));
            // biases in their accumulators' scale (fc1's: the filters'; fc2's: its filters' times fc1's, whose hidden values the
            // GELU hands over unscaled back -- gelu_phi4_scaled), the layerscale in the inverse of that: all exact
            for (int i = tid; i < M2_BV_FLOATS; i += 512)
                BV[i] = i < 192 ? wt.fc1_b[i] * wt.fc1_scale
                                : (i < 240 ? wt.fc2_b[i - 192] * (wt.fc2_scale * wt.fc1_scale) : wt.ls[i - 240] * (wt.fc2_inv * wt.fc1_inv));
        } else {
            __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)wt.fc1_w, 0, M_W_FLOATS * 4, 0x00020000);
            __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void*)wt.fc2_w, 0, M_W_FLOATS * 4, 0x00020000);
            for (int k = wave; k < M_W_FLOATS / 256; k += 8) {
                dma16(r1, W1 + k * 256, (unsigned)(k * 1024 + lane * 16));
                dma16(r2, W2 + k * 256, (unsigned)(k * 1024 + lane * 16));
            }
            for (int i = tid; i < M2_BV_FLOATS; i += 512) BV[i] = i < 192 ? wt.fc1_b[i] : (i < 240 ? wt.fc2_b[i - 192] : wt.ls[i - 240]);
        }
        if constexpr (OUT3) {
            if (tid < 147) BV[M2_BV_FLOATS + tid] = tid < 144 ? o3.w[tid] : o3.b[tid - 144];
        }
        for (int q = tid; q < D_W_FLOATS / 4; q += 512)
            reinterpret_cast<f32x4*>(Wl)[q] = reinterpret_cast<const f32x4*>(wt.dw_w)[q];
        if (tid < E_PAR_FLOATS) Pl[tid] = tid < kF ? wt.dw_b[tid] : (tid < 2 * kF ? wt.ln_w[tid - kF] : wt.ln_b[tid - 2 * kF]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                         // the bias is read before the first barrier of the tile loop

    const int t0 = (2 * row + quad) & 3;
    int rd0[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int d = 0; d < 4; ++d) rd0[c][d] = (row * 6 + quad) * 64 + ((d ^ ((t0 + c) & 3)) * 4 + g) * 4;

    // ---- MLP phase: lane -> (pixel lr of a 16-pixel group = one tile row, channel group kk), see mlp_kernel.  Its lane
    // constants (fragment pointers, row offsets, exchange addresses) are derived INSIDE the tile loop from an opaque copy
    // of the lane index: as loop invariants they would stay alive through the depth-wise phase, which has no register to
    // spare (the kernel is capped at 256 registers so that the MFMA results stay in VGPRs, where the GELU reads them).
    float* Xw = Tl;                         // the exchange: [tile row 16][chunk 3][pixel 16][16 floats]
    auto sigma = [](int q) { return q == 0 ? 0 : q == 1 ? 3 : q == 2 ? 1 : 2; };

#ifdef RVDD_STAMPS
    unsigned long long st_dw = 0, st_ln = 0, st_mlp = 0, st_wait = 0, st_n = 0, st_t = 0;
#define STAMP(acc_) do { const unsigned long long now__ = __builtin_amdgcn_s_memtime(); acc_ += now__ - st_t; st_t = now__; } while (0)
    st_t = __builtin_amdgcn_s_memtime();
#else
#define STAMP(acc_) do { } while (0)
#endif
#pragma unroll 1
    for (;;) {
        // =========================================================== depth-wise 7x7 (+bias), three 16-channel chunks
        f32x4 acc[4][3];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = *reinterpret_cast<const f32x4*>(Pl + 16 * j + 4 * g);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            // chunk 0 and 1 were requested before the MLP phase of the tile in front (or in the prologue) and every
            // wave has waited for its own pieces since; chunk 2 is requested below, once buffer 0 is free
            if (j == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (j == 1) dma_chunk(cur, 2, 0);
            if (wave >= 4) continue;
            const float* tb = Tl + (j & 1) * E_BUF_FLOATS;
            const float* wb = Wl + 16 * j + 4 * g;
            f32x4 win[2][10], wv[2][7];
            auto read_row = [&](int ky, f32x4 (&wn)[10], f32x4 (&ww)[7]) {
#pragma unroll
                for (int dx = 0; dx < 10; ++dx)
                    wn[dx] = *reinterpret_cast<const f32x4*>(tb + rd0[(2 * ky + (dx >> 2)) & 3][dx & 3] + ky * 6 * 64 + (dx >> 2) * 64);
#pragma unroll
                for (int kx = 0; kx < 7; ++kx) ww[kx] = *reinterpret_cast<const f32x4*>(wb + (ky * 7 + kx) * kF);
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
        STAMP(st_dw);
        // =========================================================== LayerNorm over the 48 channels of each pixel
        f32x4 lw[3], lb[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            lw[j] = *reinterpret_cast<const f32x4*>(Pl + kF + 16 * j + 4 * g);
            lb[j] = *reinterpret_cast<const f32x4*>(Pl + 2 * kF + 16 * j + 4 * g);
        }
        __syncthreads();                      // every wave is done with both chunk buffers: they become the exchange
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const int x_wr = ((wave & 3) * 4 + ((lane_o >> 5) << 1) + (__builtin_popcount((lane_o >> 2) & 7) & 1)) * 768 +
                         (4 * ((lane_o >> 2) & 3)) * 16 + ((((lane_o >> 2) & 3) ^ sigma(lane_o & 3)) * 4);   // + j * 256 + i * 16
#pragma unroll
        for (int i = 0; i < 4 && wave < 4; ++i) {
            float sm = 0.f;
#pragma unroll
            for (int j = 0; j < 3; ++j) sm += (acc[i][j][0] + acc[i][j][1]) + (acc[i][j][2] + acc[i][j][3]);
            sm = lane_xor_add<1>(sm);
            sm = lane_xor_add<2>(sm);
            const float u = sm / 48.f;
            float v2 = 0.f;
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float d = acc[i][j][r] - u;
                    v2 += d * d;
                }
            v2 = lane_xor_add<1>(v2);
            v2 = lane_xor_add<2>(v2);
            const float rden = 1.0f / sqrtf(v2 / 48.f + 1e-6f);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                f32x4 r;
#pragma unroll
                for (int k = 0; k < 4; ++k) r[k] = lw[j][k] * ((acc[i][j][k] - u) * rden) + lb[j][k];
                *reinterpret_cast<f32x4*>(Xw + x_wr + j * 256 + i * 16) = r;
            }
        }
        __syncthreads();                      // the rows of waves 0-3 are in the exchange
        // every wave takes two tile rows (= two groups of 16 pixels) into the MLP phase: pixel-per-lane rows
        const int lr = lane_o & 15, kk = lane_o >> 4;
        const int x_rd = lr * 16 + (((lr >> 2) ^ sigma(kk)) * 4);                                   // + n * 768 + j * 256
        lds_frag* w1p = (lds_frag*)W1 + lane_o;
        lds_frag* w2p = (lds_frag*)W2 + lane_o;
        lds_frag* bvp = (lds_frag*)BV + kk;
        asm volatile("" : "+v"(w1p), "+v"(w2p), "+v"(bvp));
        auto F1 = [&](int j, int m) { return w1p[(j * 12 + m) * 64]; };
        auto F2 = [&](int m, int mo) { return w2p[(m * 3 + mo) * 64]; };
        const unsigned lane_off = (unsigned)(lr * (kF * 4) + kk * 16);
        f32x4 xc[2][3];
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int j = 0; j < 3; ++j) xc[n][j] = *reinterpret_cast<const f32x4*>(Xw + x_rd + (2 * wave + n) * 768 + j * 256);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();                      // the exchange is over: the chunk buffers take the next tile's halo
        // (the next tile is located here, not at the top of the loop: its three scalars would live through the whole
        // depth-wise phase, and the OUT3 instantiation has no scalar register to spare)
        const int next_tile = tile + stride;
        const bool more = next_tile < band_end;
        const TilePos nxt = locate(more ? next_tile : tile);
        if (more) {
            dma_chunk(nxt, 0, 0);
            dma_chunk(nxt, 1, 1);
        }
        STAMP(st_ln);
        // =========================================================== MLP on the wave's four rows (groups of 16 pixels)
        f32x4 wq[2][2], b1n[2];
        if constexpr (!SPLIT) {
            wq[0][0] = F1(0, 0);
            wq[0][1] = F1(0, 1);
            b1n[0] = bvp[0];
            b1n[1] = bvp[4];
        }
        f32x4 vkeep[3];
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int y = cur.y0 + wave * 2 + n;
            // rows of `x` / `out` under this group: a descriptor over exactly its valid pixels (loads of the others
            // return zeros, their stores are dropped); EVERY wave issues the same number of memory instructions per
            // group, valid or not, because the waits on the next tile's chunks count them (f_mlp_vmem)
            const int valid = y < H ? min(E_TW, W - cur.x0) : 0;
            const size_t first = ((size_t)cur.b * H + min(y, H - 1)) * W + cur.x0;
            __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(x + first * kF), 0, valid * (kF * 4), 0x00020000);
            __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)(out + first * kF), 0, valid * (kF * 4), 0x00020000);
            f32x4 a2[3], xr[3], lv[3];
            if constexpr (SPLIT) {
                const char* w1b = reinterpret_cast<const char*>(W1) + lane_o * 16;
                const char* w2b = reinterpret_cast<const char*>(W2) + lane_o * 16;
                auto FA = [&](int m, int f) { return __builtin_bit_cast(h8v, *reinterpret_cast<const f32x4*>(w1b + m * 3072 + f * 1024)); };
                auto FG = [&](int p, int mo, int hl) { return __builtin_bit_cast(h8v, *reinterpret_cast<const f32x4*>(w2b + ((p * 3 + mo) * 2 + hl) * 1024)); };
                u32x2v xh[3], xl[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) split4h(xc[n][j], xh[j], xl[j]);
                const h8v B1 = cat8(xh[0], xh[1]), B2 = cat8(xl[0], xl[1]), B3 = cat8(xh[2], xh[2]);
                const h8v B4 = cat8(xl[2], u32x2v{0u, 0u});       // (its other half would meet the filters' lo halves: lo x lo, below the split's own error)
#pragma unroll
                for (int mo = 0; mo < 3; ++mo) a2[mo] = bvp[48 + 4 * mo];
#pragma unroll
                for (int mo = 0; mo < 3; ++mo) xr[mo] = bload(rx, lane_off + 64 * mo);
#pragma unroll
                for (int p = 0; p < 6; ++p) {
                    // fc1 for hidden blocks 2p, 2p+1 (two accumulator chains side by side), small terms first
                    f32x4 hq[2];
                    h8v fa[2], fb[2], fc[2];
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        hq[k] = bvp[4 * (2 * p + k)];
                        fa[k] = FA(2 * p + k, 0);
                        fb[k] = FA(2 * p + k, 1);
                        fc[k] = FA(2 * p + k, 2);
                    }
                    h8v gh[3], gl[3];
#pragma unroll
                    for (int mo = 0; mo < 3; ++mo) {
                        gh[mo] = FG(p, mo, 0);
                        gl[mo] = FG(p, mo, 1);
                    }
#pragma unroll
                    for (int k = 0; k < 2; ++k) hq[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[k], B2, hq[k], 0, 0, 0);
#pragma unroll
                    for (int k = 0; k < 2; ++k) hq[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[k], B1, hq[k], 0, 0, 0);
#pragma unroll
                    for (int k = 0; k < 2; ++k) hq[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fc[k], B4, hq[k], 0, 0, 0);
#pragma unroll
                    for (int k = 0; k < 2; ++k) hq[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fc[k], B3, hq[k], 0, 0, 0);
#pragma unroll
                    for (int k = 0; k < 2; ++k) hq[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[k], B1, hq[k], 0, 0, 0);
                    u32x2v hh[2], hl[2];
#pragma unroll
                    for (int k = 0; k < 2; ++k) split4h(gelu_phi4_scaled(hq[k], wt.gelu_c), hh[k], hl[k]);
                    const h8v Bhh = cat8(hh[0], hh[1]), Bhl = cat8(hl[0], hl[1]);
#pragma unroll
                    for (int mo = 0; mo < 3; ++mo) a2[mo] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gh[mo], Bhl, a2[mo], 0, 0, 0);
#pragma unroll
                    for (int mo = 0; mo < 3; ++mo) a2[mo] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gl[mo], Bhh, a2[mo], 0, 0, 0);
#pragma unroll
                    for (int mo = 0; mo < 3; ++mo) a2[mo] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gh[mo], Bhh, a2[mo], 0, 0, 0);
                }
#pragma unroll
                for (int mo = 0; mo < 3; ++mo) lv[mo] = bvp[60 + 4 * mo];
            } else {
            f32x4 hid[12], vq[2][3];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int st = 0; st < 18; ++st) {
                const int m = 2 * (st / 3), j = st % 3;
                if (j == 0) {
                    hid[m] = b1n[0];
                    hid[m + 1] = b1n[1];
                }
                if (st + 1 < 18) {
                    wq[(st + 1) & 1][0] = F1((st + 1) % 3, 2 * ((st + 1) / 3));
                    wq[(st + 1) & 1][1] = F1((st + 1) % 3, 2 * ((st + 1) / 3) + 1);
                    if (j == 2) {
                        b1n[0] = bvp[4 * (m + 2)];
                        b1n[1] = bvp[4 * (m + 3)];
                    }
                } else {
#pragma unroll
                    for (int mo = 0; mo < 3; ++mo) {
                        vq[0][mo] = F2(0, mo);
                        a2[mo] = bvp[48 + 4 * mo];
                    }
                }
                if (j == 1 && m >= 2) {
                    hid[m - 2] = gelu_phi4(hid[m - 2]);
                    hid[m - 1] = gelu_phi4(hid[m - 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    hid[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[st & 1][0][i], xc[n][j][i], hid[m], 0, 0, 0);
                    hid[m + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[st & 1][1][i], xc[n][j][i], hid[m + 1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) xr[j] = bload(rx, lane_off + 64 * j);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 12; ++m) {
                if (m + 1 < 12) {
#pragma unroll
                    for (int mo = 0; mo < 3; ++mo) vq[(m + 1) & 1][mo] = F2(m + 1, mo);
                } else {
                    wq[0][0] = F1(0, 0);
                    wq[0][1] = F1(0, 1);
                    b1n[0] = bvp[0];
                    b1n[1] = bvp[4];
#pragma unroll
                    for (int mo = 0; mo < 3; ++mo) lv[mo] = bvp[60 + 4 * mo];
                }
                if (m == 1) {
                    hid[10] = gelu_phi4(hid[10]);
                    hid[11] = gelu_phi4(hid[11]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int mo = 0; mo < 3; ++mo)
                        a2[mo] = __builtin_amdgcn_mfma_f32_16x16x4f32(vq[m & 1][mo][r], hid[m][r], a2[mo], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            }
            // ---- out = x + layerscale * r
            float part[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int mo = 0; mo < 3; ++mo) {
                const f32x4 v = xr[mo] + lv[mo] * a2[mo];
                bstore(ro, lane_off + 64 * mo, v);
                if constexpr (POOL) {
                    if (n == 0) {
                        vkeep[mo] = v;
                    } else {
                        // rows 2 wave and 2 wave + 1 of the tile, pixels lr and lr ^ 1: floor semantics of MaxPool2d(2)
                        // fall out of the descriptor (pooled rows / columns that do not exist have no records)
                        const int Hp = H >> 1, Wp = W >> 1;
                        const int pr = (cur.y0 >> 1) + wave, pc0 = cur.x0 >> 1;
                        const int nvalid = pr < Hp ? min(E_TW / 2, Wp - pc0) : 0;
                        __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(
                            (void*)(pool + (((size_t)cur.b * Hp + min(pr, Hp - 1)) * Wp + pc0) * kF), 0, max(nvalid, 0) * (kF * 4), 0x00020000);
                        f32x4 m;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float a = fmaxf(vkeep[mo][e], v[e]);
                            m[e] = lane_xor_max<1>(a);
                        }
                        bstore(rp, (lr & 1) ? 0x80000000u : (unsigned)((lr >> 1) * (kF * 4) + kk * 16 + 64 * mo), m);
                    }
                }
                if constexpr (OUT3) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const f32x4 w3 = bvp[M2_BV_FLOATS / 4 + c * 12 + 4 * mo];
                        part[c] += (v[0] * w3[0] + v[1] * w3[1]) + (v[2] * w3[2] + v[3] * w3[3]);
                    }
                }
            }
            if constexpr (OUT3) {
                float t[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float v = part[c];
                    v = lane_xor_add<16>(v);
                    v = lane_xor_add<32>(v);
                    t[c] = v + BV[M2_BV_FLOATS + 144 + c];
                }
                if (kk == 0 && lr < valid) {
                    const size_t p = (size_t)y * W + cur.x0 + lr;
                    if (o3.nchw) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) o3.nchw[((size_t)cur.b * 3 + c) * o3.hw + p] = t[c];
                    }
                    if (o3.nhwc4) reinterpret_cast<f32x4*>(o3.nhwc4)[(size_t)cur.b * o3.hw + p] = f32x4{t[0], t[1], t[2], 0.f};
                }
            }
        }
        STAMP(st_mlp);
#ifdef RVDD_STAMPS
        st_n += 1;
#endif
        if (!more) break;
        // the next tile's first two chunks were requested before the vector-memory instructions of this phase: all but
        // those may still be in flight
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(F_MLP_VMEM) : "memory");
        STAMP(st_wait);
        tile = next_tile;
        cur = nxt;
    }
#ifdef RVDD_STAMPS
    // diagnostic build only: cycles of wave 0 of workgroup 0 per phase, summed over its tiles, over the first pixel of `out`
    if (blockIdx.x == 0 && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        float* o = out;
        o[0] = (float)st_n; o[1] = (float)st_dw; o[2] = (float)st_ln; o[3] = (float)st_mlp; o[4] = (float)st_wait;
    }
#endif
}

// convblock_pipe_kernel: convblock_kernel<.., SPLIT = true> as a two-stage pipeline over tiles (option next_pipe).
// Per-wave stamps of convblock_kernel (LABBOOK.md section 8): waves 4-7 idle through the depth-wise phase and the
// LayerNorm, then both waves of a SIMD share its vector port through the MLP phase.  Here waves 0-3 ("front", one per
// SIMD) run the depth-wise taps and the LayerNorm of tile t+1 while waves 4-7 ("back", one per SIMD) run the MLP of tile
// t -- all 16 rows, four 16-pixel groups per wave.  Same LDS plan, same arithmetic in the same order per pixel, same bits.
//   * hand-over: the front writes the LayerNorm result into the exchange (the halo buffers, dead by then), workgroup
//     barrier A; the back reads its 12 fragments into registers, workgroup barrier B; the buffers are free again and the
//     front requests the first two halo chunks of the next tile.  Two s_barrier per tile for every wave.
//   * the halo chunks are requested and awaited by the front waves alone; "all four front waves are past this point" is
//     an LDS counter each of them adds one to and polls (fsync), since s_barrier counts all eight.
//
// PROJ: the 1x1 projection 96 -> 48 of the ConvBlock BEHIND a concat (new_unet.py:85-88, 321-329) is linear in the two
// concatenated maps, proj(cat(a, b)) = Wa a + Wb b + bias, and each of the two has the concat as its only reader: the block
// that forms a map applies that map's half of the projection to its own output in its epilogue -- the 12 values a lane holds
// are B operands as they stand -- and stores Wa a + bias, or Wa a + the other block's map (Wb b + bias, read at the same
// pixels: `pj.add`), in place of a.  The 96 -> 48 projection kernel (a pass of 4 S over maps of S bytes) is gone.  The output of a block has no
// a-priori bound (the MLP's operands have the LayerNorm's), so the split works in block floating point per PIXEL: a column
// of the product is W times one pixel's channels, any power of two per pixel factors out exactly.  POOL + PROJ: the pooled
// map is taken from the unprojected values.
template <bool OUT3, bool POOL, bool PROJ = false>
__global__ __launch_bounds__(512, 2) void convblock_pipe_kernel(const float* __restrict__ x, NextBlockW wt, float* __restrict__ out,
                                                               int B, int H, int W, int tiles_x, int tiles_y, int ntiles, Out3 o3,
                                                               float* __restrict__ pool, NextProj pj) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int SH = F_SPLIT_SHIFT;
    float* W1 = smem;
    float* W2 = smem + F_W1H_BYTES / 4;
    float* BV = smem + F_OFF_BV + SH;
    float* Wl = smem + F_OFF_DW + SH;       // [49][48]
    float* Pl = smem + F_OFF_PAR + SH;      // dw_b | ln_w | ln_b
    float* Tl = smem + F_OFF_T + SH;        // two chunk buffers; the LayerNorm exchange between the stages
    unsigned* Fs = reinterpret_cast<unsigned*>(smem + F_OFF_T + SH + 2 * E_BUF_FLOATS);       // the front's counter
    float* PW = smem + F_OFF_T + SH + 2 * E_BUF_FLOATS + 16;      // PROJ: the projection's fragments (9 KiB), then its bias
    float* PB = PW + P_W_BYTES / 4;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool front = wave < 4;

    const int per_xcd = (ntiles + 7) >> 3;
    const int band_end = min(((int)(blockIdx.x & 7) + 1) * per_xcd, ntiles);
    const int stride = (int)(gridDim.x >> 3);
    int tile = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (tile >= band_end) return;
    const int tiles_per_img = tiles_x * tiles_y;
    struct TilePos { int b, y0, x0; };
    auto locate = [&](int t) {
        TilePos p;
        p.b = t / tiles_per_img;
        const int rr = t - p.b * tiles_per_img;
        const int ty = rr / tiles_x;
        p.y0 = ty * E_TH;
        p.x0 = (rr - ty * tiles_x) * E_TW;
        return p;
    };

    {   // weights of the block -> LDS, once per workgroup (all eight waves)
        __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)wt.fc1_h, 0, F_W1H_BYTES, 0x00020000);
        __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void*)wt.fc2_h, 0, F_W2H_BYTES, 0x00020000);
        for (int k = wave; k < F_W1H_BYTES / 1024; k += 8) dma16(r1, W1 + k * 256, (unsigned)(k * 1024 + lane * 16));
        for (int k = wave; k < F_W2H_BYTES / 1024; k += 8) dma16(r2, W2 + k * 256, (unsigned)(k * 1024 + lane * 16));
        for (int i = tid; i < M2_BV_FLOATS; i += 512)
            BV[i] = i < 192 ? wt.fc1_b[i] * wt.fc1_scale
                                : (i < 240 ? wt.fc2_b[i - 192] * (wt.fc2_scale * wt.fc1_scale) : wt.ls[i - 240] * (wt.fc2_inv * wt.fc1_inv));
        if constexpr (OUT3) {
            if (tid < 147) BV[M2_BV_FLOATS + tid] = tid < 144 ? o3.w[tid] : o3.b[tid - 144];
        }
        for (int q = tid; q < D_W_FLOATS / 4; q += 512)
            reinterpret_cast<f32x4*>(Wl)[q] = reinterpret_cast<const f32x4*>(wt.dw_w)[q];
        if (tid < E_PAR_FLOATS) Pl[tid] = tid < kF ? wt.dw_b[tid] : (tid < 2 * kF ? wt.ln_w[tid - kF] : wt.ln_b[tid - 2 * kF]);
        if (tid == 0) Fs[0] = 0;
        if constexpr (PROJ) {
            __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)pj.frag, 0, P_W_BYTES, 0x00020000);
            for (int k = wave; k < P_W_BYTES / 1024; k += 8) dma16(rp, PW + k * 256, (unsigned)(k * 1024 + lane * 16));
            if (tid < kF) PB[tid] = pj.bias ? pj.bias[tid] : 0.f;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    float* Xw = Tl;                         // the exchange: [tile row 16][chunk 3][pixel 16][16 floats]
    auto sigma = [](int q) { return q == 0 ? 0 : q == 1 ? 3 : q == 2 ? 1 : 2; };
#ifdef RVDD_STAMPS
    unsigned long long ps[6] = {0, 0, 0, 0, 0, 0}, ps_t = __builtin_amdgcn_s_memtime(), ps_n = 0;
#define PSTAMP(i) do { const unsigned long long now__ = __builtin_amdgcn_s_memtime(); ps[i] += now__ - ps_t; ps_t = now__; } while (0)
#else
#define PSTAMP(i) do { } while (0)
#endif

    if (front) {
        // ================================================================================================ front
        const int g = lane & 3;
        const int idx = lane >> 2;
        const int quad = idx & 3;
        const int rw = ((idx >> 3) << 1) + (__builtin_popcount(idx & 7) & 1);      // row inside the wave's four
        const int row = wave * 4 + rw;
        int piece_yx[9];           // the 33 LDS-DMA pieces of a chunk over the four front waves
#pragma unroll
        for (int n = 0; n < 9; ++n) {
            const int k = wave + 4 * n;
            const int R = k * 4 + (lane >> 4), sl = lane & 15;
            const int p = 4 * R + ((sl >> 2) ^ (R & 3));
            const int iy = p / E_PITCH, ix = p - iy * E_PITCH;
            piece_yx[n] = (k < E_PIECES && ix < E_TW + 6) ? (iy << 8) | ix : -1;
        }
        auto dma_chunk = [&](const TilePos& tp, int j, int buf) {
            __amdgpu_buffer_rsrc_t ir =
                __builtin_amdgcn_make_buffer_rsrc((void*)(x + (size_t)tp.b * H * W * kF), 0, H * W * kF * 4, 0x00020000);
            float* dst = Tl + buf * E_BUF_FLOATS;
#pragma unroll
            for (int n = 0; n < 9; ++n) {
                const int k = wave + 4 * n;
                if (k < E_PIECES) {
                    const int gy = tp.y0 - 3 + (piece_yx[n] >> 8), gx = tp.x0 - 3 + (piece_yx[n] & 255);
                    const bool ok = piece_yx[n] >= 0 && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
                    dma16(ir, dst + k * 256, ok ? (unsigned)(((gy * W + gx) * kF + 16 * j + 4 * (lane & 3)) * 4) : 0x80000000u);
                }
            }
        };
        // all four front waves are past this point (and what they wrote to LDS, or had DMA'd, before it is visible)
        unsigned fs_target = 0;
        auto fsync = [&]() {
            fs_target += 4;
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(Fs, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            while ((int)(__hip_atomic_load(Fs, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) - fs_target) < 0) __builtin_amdgcn_s_sleep(1);
        };
        const int t0 = (2 * row + quad) & 3;
        int rd0[4][4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int d = 0; d < 4; ++d) rd0[c][d] = (row * 6 + quad) * 64 + ((d ^ ((t0 + c) & 3)) * 4 + g) * 4;

        TilePos cur = locate(tile);
        dma_chunk(cur, 0, 0);
        dma_chunk(cur, 1, 1);
#pragma unroll 1
        for (;;) {
            f32x4 acc[4][3];
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j] = *reinterpret_cast<const f32x4*>(Pl + 16 * j + 4 * g);
            PSTAMP(0);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                // j = 0: chunks 0 and 1 have landed (requested together, a tile ago); j = 2: so has chunk 2, requested below
                // when every front wave was done with chunk 0 (the wait inside fsync covers every request of this wave)
                if (j != 1) fsync();
                PSTAMP(1);
                const float* tb = Tl + (j & 1) * E_BUF_FLOATS;
                const float* wb = Wl + 16 * j + 4 * g;
                f32x4 win[2][10], wv[2][7];
                auto read_row = [&](int ky, f32x4 (&wn)[10], f32x4 (&ww)[7]) {
#pragma unroll
                    for (int dx = 0; dx < 10; ++dx)
                        wn[dx] = *reinterpret_cast<const f32x4*>(tb + rd0[(2 * ky + (dx >> 2)) & 3][dx & 3] + ky * 6 * 64 + (dx >> 2) * 64);
#pragma unroll
                    for (int kx = 0; kx < 7; ++kx) ww[kx] = *reinterpret_cast<const f32x4*>(wb + (ky * 7 + kx) * kF);
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
                PSTAMP(2);
                if (j == 0) {
                    fsync();                         // every front wave is done with buffer 0
                    dma_chunk(cur, 2, 0);
                }
            }
            // ---- LayerNorm over the 48 channels of each pixel
            f32x4 lw[3], lb[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                lw[j] = *reinterpret_cast<const f32x4*>(Pl + kF + 16 * j + 4 * g);
                lb[j] = *reinterpret_cast<const f32x4*>(Pl + 2 * kF + 16 * j + 4 * g);
            }
            fsync();                                 // every front wave is done with both buffers: they become the exchange
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));
            const int x_wr = (wave * 4 + ((lane_o >> 5) << 1) + (__builtin_popcount((lane_o >> 2) & 7) & 1)) * 768 +
                             (4 * ((lane_o >> 2) & 3)) * 16 + ((((lane_o >> 2) & 3) ^ sigma(lane_o & 3)) * 4);   // + j * 256 + i * 16
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float sm = 0.f;
#pragma unroll
                for (int j = 0; j < 3; ++j) sm += (acc[i][j][0] + acc[i][j][1]) + (acc[i][j][2] + acc[i][j][3]);
                sm = lane_xor_add<1>(sm);
                sm = lane_xor_add<2>(sm);
                const float u = sm / 48.f;
                float v2 = 0.f;
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float d = acc[i][j][r] - u;
                        v2 += d * d;
                    }
                v2 = lane_xor_add<1>(v2);
                v2 = lane_xor_add<2>(v2);
                const float rden = 1.0f / sqrtf(v2 / 48.f + 1e-6f);
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    f32x4 r;
#pragma unroll
                    for (int k = 0; k < 4; ++k) r[k] = lw[j][k] * ((acc[i][j][k] - u) * rden) + lb[j][k];
                    *reinterpret_cast<f32x4*>(Xw + x_wr + j * 256 + i * 16) = r;
                }
            }
            PSTAMP(3);
            __syncthreads();                         // A: the exchange holds this tile's LayerNorm result
            __syncthreads();                         // B: the back has it in registers; the buffers are free
            PSTAMP(4);
#ifdef RVDD_STAMPS
            ++ps_n;
#endif
            const int next_tile = tile + stride;
            if (next_tile >= band_end) break;
            tile = next_tile;
            cur = locate(tile);
            dma_chunk(cur, 0, 0);
            dma_chunk(cur, 1, 1);
        }
    } else {
        // ================================================================================================= back
        const int bw = wave - 4;                // rows 4 bw .. 4 bw + 3 of every tile
#pragma unroll 1
        for (;;) {
            const TilePos cur = locate(tile);
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));
            const int lr = lane_o & 15, kk = lane_o >> 4;
            const int x_rd = lr * 16 + (((lr >> 2) ^ sigma(kk)) * 4);                                   // + row * 768 + j * 256
            typedef __attribute__((address_space(3))) f32x4 lds_frag_t;
            lds_frag_t* bvp = (lds_frag_t*)BV + kk;
            asm volatile("" : "+v"(bvp));
            const unsigned lane_off = (unsigned)(lr * (kF * 4) + kk * 16);
            PSTAMP(0);
            __syncthreads();                         // A
            PSTAMP(1);
            f32x4 xc[4][3];
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int j = 0; j < 3; ++j) xc[n][j] = *reinterpret_cast<const f32x4*>(Xw + x_rd + (4 * bw + n) * 768 + j * 256);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();                         // B
            PSTAMP(2);
            // two 16-pixel groups (= two tile rows) at a time: they share every filter fragment, and their MFMA chains and GELUs
            // are independent, which is what hides each other's latencies now that the SIMD partner is in another phase
#pragma unroll
            for (int n2 = 0; n2 < 4; n2 += 2) {
                __amdgpu_buffer_rsrc_t rx[2], ro[2], ra[2];
                int yrow[2], valid[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    yrow[q] = cur.y0 + bw * 4 + n2 + q;
                    valid[q] = yrow[q] < H ? min(E_TW, W - cur.x0) : 0;
                    const size_t first = ((size_t)cur.b * H + min(yrow[q], H - 1)) * W + cur.x0;
                    rx[q] = __builtin_amdgcn_make_buffer_rsrc((void*)(x + first * kF), 0, valid[q] * (kF * 4), 0x00020000);
                    ro[q] = __builtin_amdgcn_make_buffer_rsrc((void*)(out + first * kF), 0, valid[q] * (kF * 4), 0x00020000);
                    if constexpr (PROJ)      // a null `add` has no records: the loads return zeros
                        ra[q] = __builtin_amdgcn_make_buffer_rsrc((void*)(pj.add ? pj.add + first * kF : x), 0,
                                                                  pj.add ? valid[q] * (kF * 4) : 0, 0x00020000);
                }
                f32x4 a2[2][3], xr[2][3], lv[3], av[2][3];
                const char* w1b = reinterpret_cast<const char*>(W1) + lane_o * 16;
                const char* w2b = reinterpret_cast<const char*>(W2) + lane_o * 16;
                auto FA = [&](int m, int f) { return __builtin_bit_cast(h8v, *reinterpret_cast<const f32x4*>(w1b + m * 3072 + f * 1024)); };
                auto FG = [&](int p, int mo, int hl) { return __builtin_bit_cast(h8v, *reinterpret_cast<const f32x4*>(w2b + ((p * 3 + mo) * 2 + hl) * 1024)); };
                h8v B1[2], B2[2], B3[2], B4[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    u32x2v xh[3], xl[3];
#pragma unroll
                    for (int j = 0; j < 3; ++j) split4h(xc[n2 + q][j], xh[j], xl[j]);
                    B1[q] = cat8(xh[0], xh[1]);
                    B2[q] = cat8(xl[0], xl[1]);
                    B3[q] = cat8(xh[2], xh[2]);
                    B4[q] = cat8(xl[2], u32x2v{0u, 0u});       // (the other half would meet the filters' lo halves: lo x lo; zeros also cost the matrix pipe less)
#pragma unroll
                    for (int mo = 0; mo < 3; ++mo) a2[q][mo] = bvp[48 + 4 * mo];
#pragma unroll
                    for (int mo = 0; mo < 3; ++mo) xr[q][mo] = bload(rx[q], lane_off + 64 * mo);
                }
                // fc1's fragments one pair of hidden blocks ahead, fc2's at the top of their pair: an LDS read issued right in
                // front of its MFMA is a wait of a few hundred cycles, and this wave has no SIMD partner in the same phase
                h8v fa[2][2], fb[2][2], fc[2][2];
                auto load_fc1 = [&](int p, int buf) {
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        fa[buf][k] = FA(2 * p + k, 0);
                        fb[buf][k] = FA(2 * p + k, 1);
                        fc[buf][k] = FA(2 * p + k, 2);
                    }
                };
                PSTAMP(3);      // (diagnostic: 3 = set-up of a row pair, 4 = its six pairs of hidden blocks, 5 = its epilogue)
                load_fc1(0, 0);
#pragma unroll
                for (int p = 0; p < 6; ++p) {
                    const int cb = p & 1;
                    f32x4 hq[2][2];
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int k = 0; k < 2; ++k) hq[q][k] = bvp[4 * (2 * p + k)];
                    if (p + 1 < 6) load_fc1(p + 1, cb ^ 1);
                    if constexpr (PROJ) {
                        // the other half of the projection, one pair of hidden blocks ahead of its use (the registers of the
                        // fc1 prefetch, which has nothing left to fetch, are free)
                        if (p == 5) {
#pragma unroll
                            for (int q = 0; q < 2; ++q)
#pragma unroll
                                for (int mo = 0; mo < 3; ++mo) av[q][mo] = bload(ra[q], lane_off + 64 * mo);
                        }
                    }
                    h8v gh[3], gl[3];
#pragma unroll
                    for (int mo = 0; mo < 3; ++mo) {
                        gh[mo] = FG(p, mo, 0);
                        gl[mo] = FG(p, mo, 1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    // The wave issues in order, and only the first two vector instructions behind an MFMA run under it.  Order of a pair
                    // of hidden blocks: fc1 of group 0 | fc1 of group 1, two instructions of GELU 0 behind each MFMA, then the rest of
                    // GELU 0 | fc2 of group 0 with GELU 1 likewise | fc2 of group 1.  Every accumulator takes its MFMAs in the order it
                    // did, every value the same instructions: same bits as the phased kernel.
                    auto fc1_mfma = [&](int q, int i) {        // i-th of ten: term i >> 1 into hidden block i & 1
                        const int k = i & 1, t = i >> 1;
                        const h8v A = (t == 0 || t == 4) ? fa[cb][k] : t == 1 ? fb[cb][k] : fc[cb][k];
                        const h8v Bv = t == 0 ? B2[q] : (t == 1 || t == 4) ? B1[q] : t == 2 ? B4[q] : B3[q];
                        hq[q][k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, Bv, hq[q][k], 0, 0, 0);
                    };
                    h8v Bhh[2], Bhl[2];
                    auto fc2_mfma = [&](int q, int i) {        // i-th of nine: term i / 3 into output block i % 3
                        const int mo = i % 3, t = i / 3;
                        a2[q][mo] = __builtin_amdgcn_mfma_f32_16x16x32_f16(t == 1 ? gl[mo] : gh[mo], t == 0 ? Bhl[q] : Bhh[q], a2[q][mo], 0, 0, 0);
                    };
#pragma unroll
                    for (int i = 0; i < 10; ++i) fc1_mfma(0, i);
                    __builtin_amdgcn_sched_barrier(0);
                    GeluStages gs[2];
                    u32x2v hh[2], hl[2];
#define GS_(q, k, S) gelu_stage<S>(gs[k], hq[q][k], wt.gelu_c, hh[k], hl[k])
#define GELU_ALL_(q) GS_(q, 0, 0); GS_(q, 1, 0); GS_(q, 0, 1); GS_(q, 1, 1); GS_(q, 0, 2); GS_(q, 1, 2); GS_(q, 0, 3); GS_(q, 1, 3); \
                     GS_(q, 0, 4); GS_(q, 1, 4); GS_(q, 0, 5); GS_(q, 1, 5)
                    // two vector instructions behind an MFMA run under it (tools/f16_mfma_bench.hip: 20.0 cycles per slot against 19.5;
                    // four cost 38.8): the first 2 N of a GELU go behind the N MFMAs of the other group, the rest follows in a run
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
                PSTAMP(4);
#pragma unroll
                for (int mo = 0; mo < 3; ++mo) lv[mo] = bvp[60 + 4 * mo];
                // ---- out = x + layerscale * r
                f32x4 v[2][3];
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int mo = 0; mo < 3; ++mo) {
                        v[q][mo] = xr[q][mo] + lv[mo] * a2[q][mo];
                        if constexpr (!PROJ) bstore(ro[q], lane_off + 64 * mo, v[q][mo]);
                    }
                if constexpr (PROJ) {
                    const char* pwb = reinterpret_cast<const char*>(PW) + lane_o * 16;
                    auto PF = [&](int m, int f) { return __builtin_bit_cast(h8v, *reinterpret_cast<const f32x4*>(pwb + m * 3072 + f * 1024)); };
                    h8v pa[3], pb[3], pc[3];
#pragma unroll
                    for (int m = 0; m < 3; ++m) {
                        pa[m] = PF(m, 0);
                        pb[m] = PF(m, 1);
                        pc[m] = PF(m, 2);
                    }
                    lds_frag_t* pbv = (lds_frag_t*)PB + kk;
                    h8v P1[2], P2[2], P3[2], P4[2];
                    float back[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        // max |.| over the pixel's 48 channels: 12 in this lane, the rest in the lanes 16, 32, 48 away.  The bits of
                        // non-negative floats order like unsigned integers
                        float mx = 0.f;
#pragma unroll
                        for (int mo = 0; mo < 3; ++mo)
                            mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v[q][mo][0]), fabsf(v[q][mo][1]))), fmaxf(fabsf(v[q][mo][2]), fabsf(v[q][mo][3])));
                        unsigned mb = __float_as_uint(mx);
                        auto s32 = __builtin_amdgcn_permlane32_swap(mb, mb, false, false);
                        mb = max(s32[0], s32[1]);
                        auto s16 = __builtin_amdgcn_permlane16_swap(mb, mb, false, false);
                        mb = max(s16[0], s16[1]);
                        // 2^(126 - e) puts the pixel's maximum into [1/2, 1); its inverse times the filters' 2^-s scales the sums
                        // back (exponent arithmetic on the bits; an absurd e -- zero, denormal, beyond 2^110 -- stays finite)
                        const int e = min((int)(mb >> 23) & 0xff, 252);
                        const float fwd = __uint_as_float((unsigned)(253 - e) << 23);
                        back[q] = __uint_as_float((unsigned)min(max(e + 1 + pj.inv_e, 1), 254) << 23);
                        u32x2v xh[3], xl[3];
#pragma unroll
                        for (int j = 0; j < 3; ++j) split4h(v[q][j] * fwd, xh[j], xl[j]);
                        P1[q] = cat8(xh[0], xh[1]);
                        P2[q] = cat8(xl[0], xl[1]);
                        P3[q] = cat8(xh[2], xh[2]);
                        P4[q] = cat8(xl[2], u32x2v{0u, 0u});
                    }
                    f32x4 pr[2][3];
                    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int m = 0; m < 3; ++m) pr[q][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pa[m], P2[q], zero4, 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int m = 0; m < 3; ++m) pr[q][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pb[m], P1[q], pr[q][m], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int m = 0; m < 3; ++m) pr[q][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pc[m], P4[q], pr[q][m], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int m = 0; m < 3; ++m) pr[q][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pc[m], P3[q], pr[q][m], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int m = 0; m < 3; ++m) pr[q][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pa[m], P1[q], pr[q][m], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int m = 0; m < 3; ++m) bstore(ro[q], lane_off + 64 * m, pr[q][m] * back[q] + (pj.add ? av[q][m] : pbv[4 * m]));
                }
                if constexpr (POOL) {
                    // the two rows are one pooling row pair, pixels lr and lr ^ 1; floor semantics of MaxPool2d(2) fall out of
                    // the descriptor (pooled rows / columns that do not exist have no records)
                    const int Hp = H >> 1, Wp = W >> 1;
                    const int pr = (cur.y0 >> 1) + 2 * bw + (n2 >> 1), pc0 = cur.x0 >> 1;
                    const int nvalid = pr < Hp ? min(E_TW / 2, Wp - pc0) : 0;
                    __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(
                        (void*)(pool + (((size_t)cur.b * Hp + min(pr, Hp - 1)) * Wp + pc0) * kF), 0, max(nvalid, 0) * (kF * 4), 0x00020000);
#pragma unroll
                    for (int mo = 0; mo < 3; ++mo) {
                        f32x4 m;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float a = fmaxf(v[0][mo][e], v[1][mo][e]);
                            m[e] = lane_xor_max<1>(a);
                        }
                        bstore(rp, (lr & 1) ? 0x80000000u : (unsigned)((lr >> 1) * (kF * 4) + kk * 16 + 64 * mo), m);
                    }
                }
                if constexpr (OUT3) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        float t[3];
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            float pc = 0.f;
#pragma unroll
                            for (int mo = 0; mo < 3; ++mo) {
                                const f32x4 w3 = bvp[M2_BV_FLOATS / 4 + c * 12 + 4 * mo];
                                pc += (v[q][mo][0] * w3[0] + v[q][mo][1] * w3[1]) + (v[q][mo][2] * w3[2] + v[q][mo][3] * w3[3]);
                            }
                            pc = lane_xor_add<16>(pc);
                            pc = lane_xor_add<32>(pc);
                            t[c] = pc + BV[M2_BV_FLOATS + 144 + c];
                        }
                        if (kk == 0 && lr < valid[q]) {
                            const size_t pp = (size_t)yrow[q] * W + cur.x0 + lr;
                            if (o3.nchw) {
#pragma unroll
                                for (int c = 0; c < 3; ++c) o3.nchw[((size_t)cur.b * 3 + c) * o3.hw + pp] = t[c];
                            }
                            if (o3.nhwc4) reinterpret_cast<f32x4*>(o3.nhwc4)[(size_t)cur.b * o3.hw + pp] = f32x4{t[0], t[1], t[2], 0.f};
                        }
                    }
                }
                PSTAMP(5);
            }
            PSTAMP(3);
#ifdef RVDD_STAMPS
            ++ps_n;
#endif
            const int next_tile = tile + stride;
            if (next_tile >= band_end) break;
            tile = next_tile;
        }
    }
#ifdef RVDD_STAMPS
    // diagnostic build only: per-phase cycles of waves 0 (front) and 4 (back) of workgroup 0 over the first pixels of `out`
    if (blockIdx.x == 0 && (tid == 0 || tid == 256)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        float* o = out + (tid >> 8) * 8;
        o[0] = (float)ps_n;
        for (int i = 0; i < 6; ++i) o[1 + i] = (float)ps[i];
    }
#endif
}

// zero_pad_features (networks/new_unet.py:56-66): src [B][h][w] -> dst [B][H][W] at (oy,ox), zeros elsewhere
__global__ void pad_copy_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int h, int w,
                                int H, int W, int oy, int ox) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t pix = gid / 12;
    const int c4 = gid - pix * 12;
    if (pix >= (size_t)B * H * W) return;
    const int X = pix % W, Y = (pix / W) % H, b = pix / ((size_t)W * H);
    const int y = Y - oy, xx = X - ox;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)y < (unsigned)h && (unsigned)xx < (unsigned)w)
        v = reinterpret_cast<const f32x4*>(src)[(((size_t)b * h + y) * w + xx) * 12 + c4];
    reinterpret_cast<f32x4*>(dst)[gid] = v;
}

int num_cus() { return current_device_cus(); }

}  // namespace

hipError_t launch_proj1x1(const float* in1, int c1, const float* in2, int c2, const float* w, const float* b,
                          float* out, int64_t npix, hipStream_t s) {
    if (npix <= 0) return hipSuccess;
    const long ngroups = (npix + 15) / 16;
    long blocks = (ngroups + 3) / 4;
    const long cap = (long)num_cus() * 8;
    if (blocks > cap) blocks = cap;
    if (c1 == 16 && c2 == 0)
        hipLaunchKernelGGL((proj1x1_kernel<16, 0>), dim3((unsigned)blocks), dim3(256), 0, s, in1, in2, w, b, out, (long)npix);
    else if (c1 == 48 && c2 == 48)
        hipLaunchKernelGGL((proj1x1_kernel<48, 48>), dim3((unsigned)blocks), dim3(256), 0, s, in1, in2, w, b, out, (long)npix);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

template <bool OUT3, bool POOL, bool SPLIT>
static hipError_t launch_block_t(const float* x, float* out, const NextBlockW& w, int B, int H, int W, Out3 o3, hipStream_t s,
                                 float* pool) {
    if ((size_t)H * W * kF * 4 >= 0x80000000ull) return hipErrorInvalidValue;
    static std::atomic<uint64_t> attr{0};
    constexpr size_t LDS = SPLIT ? F_LDS_BYTES_SPLIT : F_LDS_BYTES;
    if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(convblock_kernel<OUT3, POOL, SPLIT>), LDS, attr); e != hipSuccess)
        return e;
    const int tx = (W + E_TW - 1) / E_TW, ty = (H + E_TH - 1) / E_TH;
    const int ntiles = B * tx * ty;
    if (ntiles <= 0) return hipSuccess;
    // persistent: one workgroup per CU (LDS), never more workgroups than tiles; the XCD band map needs a multiple of 8
    const int grid = ((std::min(ntiles, num_cus()) + 7) / 8) * 8;
    hipLaunchKernelGGL((convblock_kernel<OUT3, POOL, SPLIT>), dim3(grid), dim3(512), LDS, s, x, w, out, B, H, W, tx, ty, ntiles, o3, pool);
    return hipGetLastError();
}
template <bool OUT3, bool POOL, bool PROJ = false>
static hipError_t launch_block_pipe(const float* x, float* out, const NextBlockW& w, int B, int H, int W, Out3 o3, hipStream_t s,
                                    float* pool, NextProj pj = NextProj{}) {
    if ((size_t)H * W * kF * 4 >= 0x80000000ull) return hipErrorInvalidValue;
    static std::atomic<uint64_t> attr{0};
    constexpr size_t LDS = F_LDS_BYTES_SPLIT + 64 + (PROJ ? P_W_BYTES + kF * 4 : 0);
    static_assert(LDS <= 160 * 1024, "LDS plan (pipe)");
    if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(convblock_pipe_kernel<OUT3, POOL, PROJ>), LDS, attr); e != hipSuccess)
        return e;
    const int tx = (W + E_TW - 1) / E_TW, ty = (H + E_TH - 1) / E_TH;
    const int ntiles = B * tx * ty;
    if (ntiles <= 0) return hipSuccess;
    const int grid = ((std::min(ntiles, num_cus()) + 7) / 8) * 8;
    hipLaunchKernelGGL((convblock_pipe_kernel<OUT3, POOL, PROJ>), dim3(grid), dim3(512), LDS, s, x, w, out, B, H, W, tx, ty, ntiles, o3,
                       pool, pj);
    return hipGetLastError();
}
// w.fc1_h set = the split-f16 MLP (the default; w.pipe: as a two-stage pipeline over tiles); null = the f32-MFMA form
// (option next_split 0: the A/B reference)
template <bool OUT3, bool POOL = false>
static hipError_t launch_block(const float* x, float* out, const NextBlockW& w, int B, int H, int W, Out3 o3, hipStream_t s,
                               float* pool = nullptr) {
    if (w.fc1_h && w.pipe) return launch_block_pipe<OUT3, POOL>(x, out, w, B, H, W, o3, s, pool);
    return w.fc1_h ? launch_block_t<OUT3, POOL, true>(x, out, w, B, H, W, o3, s, pool)
                   : launch_block_t<OUT3, POOL, false>(x, out, w, B, H, W, o3, s, pool);
}

hipError_t launch_next_block(const float* x, float* out, const NextBlockW& w, int B, int H, int W, hipStream_t s) {
    return launch_block<false>(x, out, w, B, H, W, Out3{}, s);
}

hipError_t launch_next_block_pool(const float* x, float* out, float* pooled, const NextBlockW& w, int B, int H, int W, hipStream_t s) {
    return launch_block<false, true>(x, out, w, B, H, W, Out3{}, s, pooled);
}

hipError_t launch_next_block_proj(const float* x, float* out, float* pooled, const NextBlockW& w, const NextProj& pj, int B, int H,
                                  int W, hipStream_t s) {
    if (!w.fc1_h || !w.pipe || !pj.frag) return hipErrorInvalidValue;
    return pooled ? launch_block_pipe<false, true, true>(x, out, w, B, H, W, Out3{}, s, pooled, pj)
                  : launch_block_pipe<false, false, true>(x, out, w, B, H, W, Out3{}, s, nullptr, pj);
}

hipError_t launch_next_block_out3(const float* x, float* out, const NextBlockW& w, int B, int H, int W, const float* w3x48,
                                  const float* b3, float* out_nchw, float* out_nhwc4, hipStream_t s) {
    return launch_block<true>(x, out, w, B, H, W, Out3{w3x48, b3, out_nchw, out_nhwc4, H * W}, s);
}

hipError_t launch_pad_copy(const float* src, float* dst, int B, int h, int w, int H, int W, int oy, int ox,
                           hipStream_t s) {
    const size_t n = (size_t)B * H * W * 12;
    hipLaunchKernelGGL(pad_copy_kernel, dim3((unsigned)((n + 191) / 192)), dim3(192), 0, s, src, dst, B, h, w, H, W,
                       oy, ox);
    return hipGetLastError();
}
