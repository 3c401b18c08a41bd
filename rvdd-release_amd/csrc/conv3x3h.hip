// 3x3 convolution (padding 1, 48 output channels) as an implicit GEMM on the F16 matrix pipe with SPLIT f32 operands:
// the same networks/unet.py layers as conv3x3.hip / wino3x3.hip (:26-76 NConvBlock, :194-208 ConvMaxPool2d, :635-669
// bottleneck, :699-720 PostConvs), f32 in, f32 out, f32 accumulation.
//
// Why.  On gfx950 v_mfma_f32_16x16x4_f32 runs at 64 FLOP/clk/SIMD (the vector rate, and ON the vector lanes: VALU work
// beside it adds to its time), v_mfma_f32_16x16x32_f16 at 16x that, and leaves the vector issue port free half of its
// cycles (tools/f16_mfma_bench.hip; profiles/r03e_f16_mfma_bench.txt).  An f32 value x is split exactly into
//     x = hi + lo + r,   hi = f16(x) rounded toward zero,  lo = f16(x - hi),  |r| <= 2^-22 |x|  (2^-25 absolute below 2^-3:
//                                                                                               lo is then subnormal, kept)
// and a product of two split values is taken as hi.hi + hi.lo + lo.hi (each exact in the f32 accumulator; the dropped
// lo.lo is 2^-22 of the product): three F16 MFMAs in place of sixteen-times-slower f32 ones.  The filters are scaled by
// a power of two per layer before the split (so their lo halves are normal numbers) and the sums scaled back in the
// epilogue, which is exact.  tests/split_precision_study.py runs the scheme through the oracle over the 30 / 90-frame
// fixtures: 2.3e-6 from the reference's frames (the f32 kernels of this library: 3e-6); f16 overflows at 65504, the
// largest activation of those runs is 8.8, and the round-toward-zero split saturates instead of producing infinities.
//
// Block floating point (rvdd_internal.h, amax_shift): the f16 exponent range is NOT a limit of this kernel.  Every input
// map carries words per sequence with the bits of its max |x| (written by the kernel that produced the map -- this
// kernel, netin_kernel; the warped features share the words of the features they were gathered from); where that maximum
// lies outside [2^-6, 2^12) the halo values are multiplied by the power of two that puts it into [2^3, 2^4) before they are split, and the sums are scaled back by its inverse in the epilogue's one fma --
// both exact.  Frames of any finite magnitude (1e5 times brighter, 2^-12 times dimmer than the documented [-1, 1]) keep
// the 22 significand bits of the split (tests/test_gpu_parity.py::test_split_path_any_magnitude).
//
// Orientation as conv3x3.hip: D[cout][pixel] += W[cout][k] X[k][pixel]; lane l holds pixel l & 15 and, per MFMA, the
// eight consecutive k of group l >> 4.  k runs (tap, channel): 8-channel group G = 4 chunk + (l >> 4) is channels
// 8 (G % 6) .. +7 of tap G / 6 (CIN = 48: 54 groups, 14 chunks of 32, the last two groups zero filters).
//
// Work: a persistent workgroup of 8 waves per CU walks 16x16-pixel tiles.  The 18x18 halo tile is fetched as f32 into
// registers one tile ahead (buffer loads, out-of-image pixels zero-filled by out-of-range offsets = padding 1), split
// ONCE per element and written to LDS as two planes, [pixel][hi 48 f16] and [pixel][lo 48 f16] (96 B per pixel, an odd
// multiple of 32: every B fragment read of the kernel is bank-conflict free, tools/lds_b128_bench.hip).  The staging
// stores between a tile's two barriers run at the LDS write port's ~60 B per clock whatever their width or pattern
// (8-byte, 16-byte after a lane exchange, conflict-free or not: 0.75-1.3 k cycles per tile).  The split filter bank
// (14 chunks x 3 cout blocks x {hi, lo} x 1 KiB lane-linear A fragments = 84 KiB) is DMA'd to LDS once per workgroup,
// beside the first tile's halo fetch.  Wave w owns tile rows 2w, 2w+1 (two B
// fragments per chunk) and all 48 couts (three A fragments): per chunk 6 + 4 ds_read_b128 (hi and lo) feed 18 MFMAs on
// six accumulators, the fragments of the next chunk being read while the current one is multiplied.
#include "rvdd_internal.h"

#include <type_traits>

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int TW = 16;
constexpr int NTHREADS = 512;

// NGRP = 1 (the library): the eight waves of a workgroup share one 16x16-pixel tile.  NGRP = 2 (round 4, the harness only):
// two GROUPS of four waves, one wave of each per SIMD, each group with its own 8x16-pixel tile, halo planes and barrier (an
// LDS counter), walking the tile list independently -- while one group is between its barriers (epilogue, staging stores) the
// other group's wave keeps the SIMD's matrix pipe busy.  Same sums in the same order, same bits; 2.6 % faster on its own
// (tools/conv3x3h_bench.hip), nothing in the net (profiles/r04_conv_two_groups_ab.txt): the tile loop is bound by what it
// issues besides the MFMAs (profiles/r04_conv3x3h_tile_loop_parts.txt), and the 8-row tile's halo is 1.41x its pixels.
// KS = 3: the convunet's 3x3 convs.  KS = 5, CIN = 16: preprocessing_layer (3x3, no activation, unet.py:742) composed with the
// first source of EncoderConvs[0][0] (3x3) into ONE 5x5 conv of the network input -- two linear maps in a row are one linear
// map (runtime.hip: compose_pre_enc0); what the zero padding BETWEEN the two does at the image border is put right by
// pre_border_fix_kernel below.
// MT = 3 (every launch with enough tiles): a workgroup forms all 48 output channels of its tiles.  MT = 1 (launches with at most a
// third of a tile per CU -- the coarse levels of one small sequence): a workgroup forms ONE block of 16 output channels
// (blockIdx.y), three workgroups share a tile: a launch of one tile per workgroup is one workgroup's serial path (fetch, staging,
// 252 MFMAs per wave = 4 us, epilogue, tail: 10 us whatever the level) on as many CUs as it has tiles, and a third of the MFMAs per
// workgroup on three times the CUs takes 3 us off it.  Same sums per output channel in the same order: the same bits.
template <int CIN, int NGRP, int KS = 3, int MT = 3>
struct HGeo {
    static constexpr int PAD = KS / 2;
    static constexpr int TH = 16 / NGRP, IH = TH + KS - 1;   // tile rows of a group (two per wave), with halo
    static constexpr int IW = TW + KS - 1;
    static constexpr int NT = NTHREADS / NGRP;               // threads of a group
    static constexpr int WPG = NT / 64;                      // waves of a group
    static constexpr int GPT = CIN / 8;                      // 8-channel groups per tap
    static constexpr int NG = KS * KS * GPT;                 // groups of the whole filter
    static constexpr int NCH = (NG + 3) / 4;                 // K chunks of 32 (one MFMA deep)
    static constexpr int HI = CIN * 2;                       // bytes of one pixel's hi half
    static constexpr int S = HI;                             // LDS bytes per pixel in EACH of the two planes (hi, lo): 96 / 32, an odd multiple of
                                                             // 32 -- every fragment pattern of the kernel is conflict-free
                                                             // (tools/lds_b128_bench.hip; 192 or 208 are 2-way), and 10 KiB less than one
                                                             // interleaved [hi | lo | pad] image of 224 B per pixel
    static constexpr int PLANE = IH * IW * HI;
    static constexpr int W_BYTES = NCH * MT * 2 * 1024;
    static constexpr int I_BYTES = 2 * PLANE;                // one group's two planes
    static constexpr int P_FLOATS = 48 + 3 * 48 + 4 + 8 + 4; // bias, PostConvs[1] weights and bias, one word per wave (amax reduction), the groups' barrier counters
    static constexpr int LDS_BYTES = W_BYTES + NGRP * I_BYTES + P_FLOATS * 4;
    static constexpr int SEG = CIN / 4;                      // 16-B pieces of one f32 pixel
    static constexpr int ROWSEG = IW * SEG;                  // ... of one halo row
    static constexpr int RPR = NT / ROWSEG;                  // halo rows fetched per round of one load per thread of the group
    static constexpr int NR = (IH + RPR - 1) / RPR;          // rounds per tile
    static constexpr int A_SPLIT = 7;                        // chunks reached from the first A base pointer (< 64 KiB of offsets)
};

__device__ __forceinline__ f32x4 bload(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff = 0) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ void bstore(__amdgpu_buffer_rsrc_t r, unsigned off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, off, 0, 0);
}

// x (four f32; SCALED: sc x, sc = the map's power-of-two scale) -> hi, lo (four f16 each, as two dwords): hi toward zero
// (never an infinity), lo = x - hi to nearest.  x - hi as ONE v_fma_mix_f32 per value (the f16 half read in place); x comes
// from a buffer load or an interpolation, never straight out of an MFMA (inline asm behind an MFMA gets no wait states).
// Issue cost per four values (tools/valu_rate_bench.hip, profiles/r04_valu_rate_bench.txt): 37 cycles; the scaling adds two
// v_pk_mul_f32 (10 cycles), which is why a map that needs no scaling (amax_shift) takes the unscaled form.  The same split
// made of v_fma_mixlo/hi_f16 -- scale, subtraction and rounding in one instruction per half -- is 8 instructions of 7.4
// cycles each: slower (measured in the kernel too).
template <bool SCALED>
__device__ __forceinline__ void split4(f32x4 x, float sc, u32x2& hi, u32x2& lo) {
    if constexpr (SCALED) x = x * sc;
    const fp16x2 h01 = __builtin_amdgcn_cvt_pkrtz(x[0], x[1]);
    const fp16x2 h23 = __builtin_amdgcn_cvt_pkrtz(x[2], x[3]);
    const unsigned u01 = __builtin_bit_cast(unsigned, h01), u23 = __builtin_bit_cast(unsigned, h23);
    float r0, r1, r2, r3;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(u01), "v"(x[0]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(u01), "v"(x[1]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r2) : "v"(u23), "v"(x[2]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r3) : "v"(u23), "v"(x[3]));
    const h2 l01 = {(_Float16)r0, (_Float16)r1};
    const h2 l23 = {(_Float16)r2, (_Float16)r3};
    hi = u32x2{u01, u23};
    lo = u32x2{__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23)};
}
struct TilePos {
    int b, y0, x0;
};

// Diagnostic build (-DRVDD_STAMPS, tools/conv3x3h_bench.hip): every wave adds the shader cycles it spends
// in each phase of the tile loop to g_stamps (s_memtime; the reads cost a few per cent themselves).
#ifdef RVDD_STAMPS
__device__ unsigned long long g_stamps[8 * 8];      // [wave][phase 0..6, tiles]
#define STAMP_DECL unsigned long long st_acc[7] = {0, 0, 0, 0, 0, 0, 0}, st_t = __builtin_amdgcn_s_memtime(), st_n = 0
#define STAMP(i)                                                       \
    do {                                                               \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();  \
        st_acc[i] += now_ - st_t;                                      \
        st_t = now_;                                                   \
    } while (0)
#define STAMP_FLUSH                                                                        \
    do {                                                                                   \
        if ((threadIdx.x & 63) == 0) {                                                     \
            const int w_ = threadIdx.x >> 6;                                               \
            for (int i_ = 0; i_ < 7; ++i_) atomicAdd(&g_stamps[w_ * 8 + i_], st_acc[i_]);   \
            atomicAdd(&g_stamps[w_ * 8 + 7], st_n);                                        \
        }                                                                                  \
    } while (0)
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_FLUSH
#endif

// UPS (UpConv, networks/unet.py:88-147): `a.in` is the half-resolution map [B][H/2][W/2][48] and the conv input is its
// bilinear x2 upsample (align_corners=False), interpolated in the halo fetch (2x2 blocks of halo pixels from four source
// pixels each) with the expressions of upsample2x_kernel (prestage.hip) in the same order: the same bits as "upsample,
// then conv", the upsampled map is never written.
template <int CIN, int EPI, bool ACC_IN, bool UPS = false, int NGRP = 2, int KS = 3, int MT = 3>
__global__ __launch_bounds__(NTHREADS) void conv3x3h_kernel(ConvArgs a) {
    using G = HGeo<CIN, NGRP, KS, MT>;
    static_assert(MT == 3 || (MT == 1 && EPI != EPI_RELU_OUT3 && NGRP == 1), "the 1x1 output needs all 48 channels of a pixel");
#ifdef RVDD_UPS_PAD      // diagnostic builds (tools/ups_layout_matrix.sh): the fused-upsample kernel's code moved by 4 x RVDD_UPS_PAD bytes
    if constexpr (UPS) {
#pragma unroll
        for (int i = 0; i < RVDD_UPS_PAD; ++i) asm volatile("s_nop 0");
    }
#endif
    const int mt0 = MT == 3 ? 0 : (int)blockIdx.y;      // first 16-channel output block of this workgroup
    constexpr int TH = G::TH, IH = G::IH, IW = G::IW, PAD = G::PAD;
    static_assert(KS == 3 || (!UPS && !ACC_IN), "the 5x5 form exists for the composed first layer only");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    typedef __attribute__((address_space(3))) f32x4 lds_f4;
    typedef __attribute__((address_space(3))) u32x2 lds_u2;
    lds_u8* L = (lds_u8*)smem;
    float* Pl = reinterpret_cast<float*>(smem + G::W_BYTES + NGRP * G::I_BYTES);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave / G::WPG, gw = wave - grp * G::WPG;      // group, wave inside it (waves w and w + 4 share a SIMD)
    const int gtid = tid - grp * G::NT;                           // thread inside the group
    const unsigned plane0 = (unsigned)(G::W_BYTES + grp * G::I_BYTES);      // the group's halo planes
    const int n = lane & 15;
    const int g = lane >> 4;

    // ---- tiles of this workgroup: the workgroups of one XCD (blockIdx & 7) share a contiguous eighth of the tile list,
    // walked side by side, so that the halo columns two neighbours both read meet in that XCD's L2
    int t, t_end, t_step;
    if ((gridDim.x & 7) == 0) {
        const int xcd = blockIdx.x & 7;
        t = (int)(((long long)a.ntiles * xcd) >> 3) + (int)(blockIdx.x >> 3);
        t_end = (int)(((long long)a.ntiles * (xcd + 1)) >> 3);
        t_step = gridDim.x >> 3;
    } else {
        t = blockIdx.x;
        t_end = a.ntiles;
        t_step = gridDim.x;
    }
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    auto locate = [&](int tile, TilePos& p) {
        p.b = __builtin_amdgcn_readfirstlane(tile / tiles_per_img);
        const int rr = tile - p.b * tiles_per_img;
        const int ty = __builtin_amdgcn_readfirstlane(rr / a.tiles_x);
        p.y0 = ty * TH;
        p.x0 = (rr - ty * a.tiles_x) * TW;
    };

    // block floating point: the power of two for a sequence of the input map and its inverse (1, 1 without amax words).
    // The word is requested with the next tile's addresses, IN FRONT of its halo loads, and looked at where the first of those
    // loads is split: the wait the compiler counts for it there is the wait for the halo data anyway.  (Looked at where it is
    // requested it cost 4 % of the layer: as a vector load an s_waitcnt vmcnt(0) in the chunk loop -- a drain of the loads and
    // stores in flight --, as a scalar load an s_waitcnt lgkmcnt(0) that also waits for the fragment reads.)
    unsigned ab_nxt = 0;
    auto amax_fetch = [&](int b, bool live) {      // lane l < kAmaxLines: word l of the sequence; no words / no tile: 0
        // (a plain load from a clamped address and a select: a buffer descriptor of its own cost the kernel four scalar
        // registers it does not have -- the two-pass and fused-upsample instantiations spilled eight)
        const bool ok = a.amax_in && live && b < a.B;
        const unsigned* words = a.amax_in ? a.amax_in : reinterpret_cast<const unsigned*>(a.bias);      // (never read through when null)
        const unsigned v = words[ok ? (b * kAmaxSeqWords + (lane & (kAmaxLines - 1)) * kAmaxLineWords) : 0];
        ab_nxt = (ok && lane < kAmaxLines) ? v : 0u;
    };
    auto scale_from = [&](unsigned bits, float& sc, float& inv) {
        const int k = amax_shift(amax_lines_max(bits));
        sc = pow2f(k);
        inv = pow2f(-k);
    };
    // Does any sequence this workgroup works on need the scaling at all?  With frames in the reference's [-1, 1] none does
    // (amax_shift: 0 inside [2^-6, 2^12)), and the tile loop then runs in its second form below: no per-tile word, no
    // multiplication in the split (two v_pk_mul_f32 per four values: 1.5 % of the layer), the filters' scale alone in the
    // epilogue.  Decided once, per workgroup.  The words (of up to four sequences; a workgroup whose tiles span more takes the
    // scaled form) are requested BEFORE the filter bank's DMA -- vmcnt retires in order: behind it the answer would wait for all
    // 84 KiB -- and looked at behind it, when they have long arrived.
    unsigned ab_first[4] = {0, 0, 0, 0};
    bool many_seqs = false;
    if (a.amax_in && t < t_end) {
        const int b_first = t / tiles_per_img, b_last = (t + (t_end - 1 - t) / t_step * t_step) / tiles_per_img;
        many_seqs = b_last - b_first >= 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            amax_fetch(b_first + i, b_first + i <= b_last);
            ab_first[i] = ab_nxt;
        }
    }

    {   // split filter bank -> LDS (linear copy of the host arrangement), every workgroup starting at another piece
        __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, G::NCH * 3 * 2 * 1024, 0x00020000);
        constexpr int NP = G::W_BYTES / 1024;
        const int rot = (int)((blockIdx.x * 37u) % (unsigned)NP);
        for (int i = wave; i < NP; i += NTHREADS / 64) {
            int k = i + rot;
            if (k >= NP) k -= NP;
            // (MT = 1: piece k = (chunk, hi | lo) of this workgroup's output block, out of the host's [chunk][block 3][hi lo] order)
            const unsigned src = MT == 3 ? (unsigned)k : (unsigned)(((k >> 1) * 3 + mt0) * 2 + (k & 1));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void*)(smem + k * 1024), 16, src * 1024u + (unsigned)(lane * 16), 0, 0, 0);
        }
    }
    if (tid < kF) Pl[tid] = a.bias[tid];
    if (tid >= 480 && tid < 484) reinterpret_cast<unsigned*>(Pl)[G::P_FLOATS - 4 + tid - 480] = 0u;      // the groups' barrier counters
    if constexpr (EPI == EPI_RELU_OUT3) {
        if (tid >= 64 && tid < 64 + 3 * kF) Pl[kF + tid - 64] = a.w3[tid - 64];
        else if (tid >= 256 && tid < 259) Pl[4 * kF + tid - 256] = a.b3[tid - 256];
    }
    // (no wait here: the bank's DMA runs beside the first tile's halo fetch; both are awaited in front of the tile loop,
    // whose first barrier publishes bank, parameters and tile together)


    // ---- halo fetch: thread -> (row rp of the round, halo column hx, 16-B piece `part`), the same for every tile
    const int rp = gtid / G::ROWSEG;
    const int rem = gtid - rp * G::ROWSEG;
    const int hx = rem / G::SEG;
    const int part = rem - hx * G::SEG;
    const bool ld_thread = rp < G::RPR;
    const int g_lane = (rp * a.W + hx) * (CIN * 4) + part * 16;           // byte offset inside the image, relative to the halo origin
    const unsigned l_lane = plane0 + (unsigned)((rp * IW + hx) * G::S + part * 8);
    // One round = one 16-B load per thread (RPR halo rows).  The rounds of the NEXT tile are issued one per chunk inside
    // this tile's MFMA loop (a CU's texture path moves 64 B per clock: the 72 KiB of a halo tile are 1100 cycles of it, and
    // issued in one burst they stood in front of the MFMAs), split in registers under the last chunks, and written to LDS
    // between the two barriers at the end of the tile.
    f32x4 pre[G::NR];
    u32x2 shi[G::NR], slo[G::NR];
    struct Src {
        __amdgpu_buffer_rsrc_t r;      // the sequence's image: offsets past its last byte read as zero
        unsigned base;                 // this thread's byte offset of (halo row rp, column hx, piece) in round 0 -- modulo 2^32: a row
                                       // above the image is a huge offset (out of range = the zero padding), as is a row below it;
                                       // 2^31 for a column outside the image or a thread without a piece (it stays out of range
                                       // through all the rounds: they add a few MB at most)
        int y0, x0;
        int soff;                      // UPS, interior tile (see fetch_ups): byte offset of the tile's first source pixel; -1: the general form
    };
    auto source = [&](const TilePos& p, bool live) {
        Src q;
        const int ih = UPS ? a.H >> 1 : a.H, iw = UPS ? a.W >> 1 : a.W;
        q.r = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (size_t)(live ? p.b : 0) * ih * iw * CIN), 0,
                                                live ? (unsigned)(ih * iw * CIN * 4) : 0, 0x00020000);
        const bool xok = ld_thread && (unsigned)(p.x0 - PAD + hx) < (unsigned)a.W;
        q.base = xok ? (unsigned)(g_lane + ((p.y0 - PAD) * a.W + (p.x0 - PAD)) * (CIN * 4)) : 0x80000000u;
        q.y0 = p.y0;
        q.x0 = p.x0;
        q.soff = -1;
        if constexpr (UPS) {
            if (p.y0 >= 16 && p.y0 + IH <= a.H && p.x0 >= 16 && p.x0 + IW <= a.W) q.soff = (((p.y0 >> 1) - 1) * iw + (p.x0 >> 1) - 1) * (CIN * 4);
        }
        return q;
    };
    // (one v_add per round; the image's rows above and below come out of the buffer's range check, see Src::base.  Rows of the
    // last round beyond the halo, CIN = 16, are loaded and not stored.)
    auto fetch_round = [&](const Src& q, int r0) { pre[r0] = bload(q.r, q.base + (unsigned)(r0 * G::RPR * a.W * (CIN * 4))); };
    // ---- UPS: the halo tile in 2x2 blocks.  Upsampled rows 2i+1, 2i+2 interpolate between the SAME two source rows
    // (i, i+1, clamped as ATen clamps them), columns alike, and a tile's halo starts at an odd row and column: its 18x18
    // pixels are 9x9 such blocks, each from four source pixels.  One work item = one block x one 16-B channel piece: four
    // loads, two vertical interpolations shared by the block's two columns, four outputs -- against four loads and three
    // interpolations per OUTPUT piece when every halo piece is fetched on its own (TA traffic / 4, a third fewer FMAs).
    // Two items per thread (972 of 1024 slots); a group of four waves with its 10x18 halo: 5x9 blocks, three per thread.
    constexpr int UNR = UPS ? (NGRP == 1 ? 2 : 3) : 1;
    constexpr int UBLK = (IH / 2) * 9;
    // per item: (block row | block column << 8 | piece << 16 | valid << 24) in ONE register, and the LDS byte address of the block's first
    // pixel and piece (the other three pixels of the block at immediate offsets): this instantiation sits at the 256-register limit
    int u_pk[UNR];
    unsigned u_lds[UNR];
#pragma unroll
    for (int r0 = 0; r0 < UNR; ++r0) {
        const int item = gtid + G::NT * r0, blk = item / 12;
        const int part_ = item - blk * 12, by_ = blk / 9, bx_ = blk - by_ * 9;
        u_pk[r0] = by_ | (bx_ << 8) | (part_ << 16) | ((blk < UBLK ? 1 : 0) << 24);
        u_lds[r0] = plane0 + (unsigned)(((2 * by_) * IW + 2 * bx_) * G::S + part_ * 8);
    }
    auto U_BY = [&](int r0) { return u_pk[r0] & 0xff; };
    auto U_BX = [&](int r0) { return (u_pk[r0] >> 8) & 0xff; };
    auto U_PART = [&](int r0) { return (u_pk[r0] >> 16) & 0xff; };
    auto U_OK = [&](int r0) { return (u_pk[r0] >> 24) != 0; };
    f32x4 ulo[UNR][4];                 // source pixels (row 0 col 0, row 0 col 1, row 1 col 0, row 1 col 1)
    u32x2 ushi[UNR][4], uslo[UNR][4];
    // The weights of a border tile's item -- of the second source row / column per block row / column, < 0: outside the map -- are formed
    // where they are used, in the interpolation, NOT kept from the fetch four chunks earlier.  Kept, they were the one state of this loop
    // that an interior tile leaves unwritten (it needs none), i.e. loop-carried registers whose value nobody wants on most trips; builds
    // with that shape staged wrong pixels in the first trip of the loop, rarely, and only when the kernel's code was not in the
    // instruction cache yet (profiles/r06s_upsample_nondeterminism.md: found by a run-to-run soak, every build without such registers is
    // clean at every code alignment tried).  Every register of this loop is now written in full on every path before it is read.
    auto ups_weights = [&](const Src& q, int r0, float (&ly)[2], float (&lx)[2]) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int Y = q.y0 - 1 + 2 * U_BY(r0) + e, X = q.x0 - 1 + 2 * U_BX(r0) + e;
            const float py = fmaxf(0.5f * ((float)Y + 0.5f) - 0.5f, 0.f), px = fmaxf(0.5f * ((float)X + 0.5f) - 0.5f, 0.f);
            ly[e] = (unsigned)Y < (unsigned)a.H ? py - (float)(int)py : -1.f;
            lx[e] = (unsigned)X < (unsigned)a.W ? px - (float)(int)px : -1.f;
        }
    };
    // INTERIOR tiles (the halo and the source rows / columns it interpolates between all inside the map: 93 % of a 720p level's tiles):
    // no clamp acts and no weight depends on the tile -- an item's four source pieces sit at a thread constant plus a wave-uniform
    // offset (scalar operand of the load, the column step in the instruction's immediate), the weights are 1/4 and 3/4.  No address
    // or weight arithmetic at all on the vector ALU, and no zero select (the general form below spends ~85 vector instructions per
    // item on them, beside the 64 of interpolation and split; this kernel's vector instructions add to its tile time one for one).
    // The same expressions on the same values as the general form: the same bits.
    int u_off00[UNR];
#pragma unroll
    for (int r0 = 0; r0 < UNR; ++r0)
        u_off00[r0] = U_OK(r0) ? ((U_BY(r0) * (a.W >> 1) + U_BX(r0)) * (CIN * 4) + U_PART(r0) * 16) : (int)0x80000000;
    auto fetch_ups = [&](const Src& q, int r0) {
        const int ih = a.H >> 1, iw = a.W >> 1;
        if (q.soff >= 0) {
            const int soff0 = q.soff, soff1 = soff0 + iw * (CIN * 4);
            ulo[r0][0] = bload(q.r, (unsigned)u_off00[r0], soff0);
            ulo[r0][1] = bload(q.r, (unsigned)u_off00[r0] + (unsigned)(CIN * 4), soff0);
            ulo[r0][2] = bload(q.r, (unsigned)u_off00[r0], soff1);
            ulo[r0][3] = bload(q.r, (unsigned)u_off00[r0] + (unsigned)(CIN * 4), soff1);
            return;
        }
        const int i = (q.y0 >> 1) - 1 + U_BY(r0), jx = (q.x0 >> 1) - 1 + U_BX(r0);
        const int rr0 = min(max(i, 0), ih - 1), cc0 = min(max(jx, 0), iw - 1);
        const int rr1 = rr0 + (rr0 < ih - 1 ? 1 : 0), cc1 = cc0 + (cc0 < iw - 1 ? 1 : 0);
        // (a block with no pixel inside the map is not fetched)
        const int Y0 = q.y0 - 1 + 2 * U_BY(r0), X0 = q.x0 - 1 + 2 * U_BX(r0);
        const bool any = U_OK(r0) && ((unsigned)Y0 < (unsigned)a.H || (unsigned)(Y0 + 1) < (unsigned)a.H) &&
                         ((unsigned)X0 < (unsigned)a.W || (unsigned)(X0 + 1) < (unsigned)a.W);
        const unsigned p16 = (unsigned)(U_PART(r0) * 16);
        ulo[r0][0] = bload(q.r, any ? (unsigned)((rr0 * iw + cc0) * (CIN * 4)) + p16 : 0x80000000u);
        ulo[r0][1] = bload(q.r, any ? (unsigned)((rr0 * iw + cc1) * (CIN * 4)) + p16 : 0x80000000u);
        ulo[r0][2] = bload(q.r, any ? (unsigned)((rr1 * iw + cc0) * (CIN * 4)) + p16 : 0x80000000u);
        ulo[r0][3] = bload(q.r, any ? (unsigned)((rr1 * iw + cc1) * (CIN * 4)) + p16 : 0x80000000u);
    };
    // vertical pass, then horizontal, each "a * wa, then one fused multiply-add": upsample2x_kernel's expressions in its order
    auto interp_ups = [&](const Src& q, int r0, float sc_st, auto scaled_tag) {
        constexpr bool SC_ = decltype(scaled_tag)::value;
        auto fma4 = [](f32x4 x, float sc, f32x4 c) { return __builtin_elementwise_fma(x, f32x4{sc, sc, sc, sc}, c); };
        if (q.soff >= 0) {      // halo rows / columns of an interior tile: odd first (weight 1/4 on the second source), then even (3/4)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float ly1 = e ? 0.75f : 0.25f, ly0 = 1.f - ly1;
                const f32x4 c0 = fma4(ulo[r0][2], ly1, ulo[r0][0] * ly0), c1 = fma4(ulo[r0][3], ly1, ulo[r0][1] * ly0);
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    const float lx1 = f ? 0.75f : 0.25f, lx0 = 1.f - lx1;
                    split4<SC_>(fma4(c1, lx1, c0 * lx0), sc_st, ushi[r0][2 * e + f], uslo[r0][2 * e + f]);
                }
            }
            return;
        }
        float u_ly[2], u_lx[2];
        ups_weights(q, r0, u_ly, u_lx);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float ly1 = u_ly[e], ly0 = 1.f - ly1;
            const f32x4 c0 = fma4(ulo[r0][2], ly1, ulo[r0][0] * ly0), c1 = fma4(ulo[r0][3], ly1, ulo[r0][1] * ly0);
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const float lx1 = u_lx[f], lx0 = 1.f - lx1;
                const f32x4 v = fma4(c1, lx1, c0 * lx0);
                split4<SC_>((ly1 < 0.f || lx1 < 0.f) ? f32x4{0.f, 0.f, 0.f, 0.f} : v, sc_st, ushi[r0][2 * e + f], uslo[r0][2 * e + f]);
            }
        }
    };
    auto write_ups = [&]() {
#pragma unroll
        for (int r0 = 0; r0 < UNR; ++r0)
            if (U_OK(r0)) {
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int f = 0; f < 2; ++f) {
                        const unsigned ad = u_lds[r0] + (unsigned)((e * IW + f) * G::S);
                        *(lds_u2*)(L + ad) = ushi[r0][2 * e + f];
                        *(lds_u2*)(L + ad + G::PLANE) = uslo[r0][2 * e + f];
                    }
            }
    };
    auto write_tile = [&]() {       // split halves -> LDS
        if constexpr (UPS) {
            write_ups();
            return;
        }
#pragma unroll
        for (int r0 = 0; r0 < G::NR; ++r0)
            if (ld_thread && G::RPR * r0 + rp < IH) {
                *(lds_u2*)(L + l_lane + r0 * G::RPR * IW * G::S) = shi[r0];
                *(lds_u2*)(L + l_lane + r0 * G::RPR * IW * G::S + G::PLANE) = slo[r0];
            }
    };

    // ---- fragment addresses
    // (kept as LDS POINTERS, the dynamic-LDS base added here once: as byte offsets added to `L` at every use they cost one
    // v_add_u32 v, 0, v per chunk inside the tile loop -- the base is a link-time zero the compiler cannot fold)
    lds_u8* boff[G::NCH];            // B: this lane's pixel of tile row 2 wave, at the tap and channel group of chunk j
#pragma unroll
    for (int j = 0; j < G::NCH; ++j) {
        int Gi = 4 * j + g;
        if (Gi >= G::NG) Gi = G::NG - 1;          // the zero-filter groups of the last chunk: any valid address
        const int tap = Gi / G::GPT;
        const int c0 = (Gi - tap * G::GPT) * 8;
        const int ky = tap / KS, kx = tap - KS * ky;
        boff[j] = L + (plane0 + (unsigned)(((2 * gw + ky) * IW + n + kx) * G::S + c0 * 2));
    }
    lds_u8* abase[2] = {L + (unsigned)(lane * 16), L + (unsigned)(lane * 16 + G::A_SPLIT * MT * 2 * 1024)};
    asm volatile("" : "+v"(abase[0]), "+v"(abase[1]));
    h8 Af[2][MT][2], Bf[2][2][2];
    auto read_frags = [&](int buf, int j) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int hl = 0; hl < 2; ++hl)
                Bf[buf][nt][hl] = __builtin_bit_cast(h8, *(lds_f4*)(boff[j] + nt * IW * G::S + hl * G::PLANE));
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int hl = 0; hl < 2; ++hl) {
                const int jj = j < G::A_SPLIT ? j : j - G::A_SPLIT;
                Af[buf][mt][hl] = __builtin_bit_cast(h8, *(lds_f4*)(abase[j < G::A_SPLIT ? 0 : 1] + ((jj * MT + mt) * 2 + hl) * 1024));
            }
    };

    const unsigned map_bytes = (unsigned)(a.H * a.W * kF * 4);
    const unsigned out_bytes = (unsigned)(a.Hout * a.Wout * kF * 4);

    float sc_nxt = 1.f, inv_cur = 1.f, inv_nxt = 1.f;
    float amx = 0.f;        // max |x| of what this wave has stored for sequence amx_b
    int amx_b = -1, pend_b = -1;
    unsigned* Rl = reinterpret_cast<unsigned*>(Pl + G::P_FLOATS - 12) + grp * G::WPG;      // one word per wave of the group
    auto amax_send = [&](int b) {      // the group's first wave, behind a barrier: the group's maximum for sequence b, to one of the kAmaxLines lines
        unsigned t = lane < G::WPG ? Rl[lane] : 0u;
        t = amax_lines_max(t);
        if (lane == 0 && t)
            atomicMax(a.amax_out + (size_t)b * kAmaxSeqWords + (((MT == 3 ? blockIdx.x : blockIdx.x + blockIdx.y * 5u) * NGRP + grp) % kAmaxLines) * kAmaxLineWords, t);
    };
    // The group's barrier.  One group = the whole workgroup: s_barrier.  Two groups: a counter in LDS every wave of the group
    // adds one to and polls (s_barrier counts all eight waves, and the point of the groups is that they do NOT wait for each
    // other).  LDS operations of a wave complete in order, so "my staging stores / fragment reads are done" is lgkmcnt(0) in
    // front of the add -- and nothing else: no vmcnt wait, the halo loads and result stores in flight stay in flight.
    typedef __attribute__((address_space(3))) unsigned lds_u32;
    lds_u32* gbar = (lds_u32*)(lds_u8*)(smem + G::W_BYTES + NGRP * G::I_BYTES + (G::P_FLOATS - 4) * 4) + grp;
    unsigned gcount = 0;
    auto gsync = [&]() {
        if constexpr (NGRP == 1) {
            __syncthreads();
        } else {
            gcount += G::WPG;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(gbar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            while (__hip_atomic_load(gbar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < gcount) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
        }
    };

    bool any_scaled = many_seqs;
    if (a.amax_in) {
        unsigned need = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) need |= (unsigned)amax_shift(amax_lines_max(ab_first[i]));
        any_scaled = any_scaled || need != 0;
    }

    // the groups take the workgroup's tiles in turn
    t += grp * t_step;
    t_step *= NGRP;

    auto tile_loop = [&](auto scaled_tag) {
    constexpr bool SC = decltype(scaled_tag)::value;
    TilePos cur, nxt;
    locate(t, cur);
    {   // the first tile: fetched, split and staged before the loop
        if constexpr (SC) {
            amax_fetch(cur.b, t < t_end);
            scale_from(ab_nxt, sc_nxt, inv_cur);
        }
        const Src q = source(cur, t < t_end);
        if constexpr (UPS) {
#pragma unroll
            for (int r0 = 0; r0 < UNR; ++r0) fetch_ups(q, r0);
#pragma unroll
            for (int r0 = 0; r0 < UNR; ++r0) interp_ups(q, r0, sc_nxt, scaled_tag);
        } else {
#pragma unroll
            for (int r0 = 0; r0 < G::NR; ++r0) fetch_round(q, r0);
#pragma unroll
            for (int r0 = 0; r0 < G::NR; ++r0) split4<SC>(pre[r0], sc_nxt, shi[r0], slo[r0]);
        }
        write_tile();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's pieces of the filter bank have landed
        __syncthreads();       // bank, parameters, barrier counters and both groups' first tiles: the one barrier all eight waves share
    }
    // results of the tile before, stored one 16-B piece per chunk inside the current tile's MFMA loop (before the first
    // tile: out-of-range offsets, the stores are dropped)
    constexpr int NOUT = (EPI == EPI_POOL ? 1 : 2) * MT;
    f32x4 outv[NOUT];
    unsigned so_prev[2] = {0x80000000u, 0x80000000u};
    __amdgpu_buffer_rsrc_t orr_prev = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, 0, 0x00020000);
#pragma unroll
    for (int i = 0; i < NOUT; ++i) outv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // one 16-B piece of the previous tile's results: they ride between the chunks of the current tile's MFMA loop
    auto store_prev = [&](int i) {
        bstore(orr_prev, so_prev[EPI == EPI_POOL ? 0 : i / MT] + 64 * (mt0 + i % MT), outv[i]);
    };
    STAMP_DECL;
#pragma unroll 1
    while (t < t_end) {
        STAMP(0);
        gsync();               // tile t is staged
        STAMP(1);
        // The first chunk's MFMAs go out at once; the tile's address work (next tile's position, this tile's store and
        // side-load offsets) follows them and runs under them, and everything that needs it starts at chunk SH.
        constexpr int SH = G::NCH >= 12 ? 1 : 0;
        // Where the memory instructions sit in the 14 chunks (profiles/r04_chunk_loop_schedule_ab.txt: +2.5 % on C2 over "everything
        // from chunk 1 on"): the next tile's halo loads at chunks 1-9, the previous tile's stores at 7-12 (two-pass layers: at 1-6,
        // and the partial sums' loads at 7-12) -- at most two memory instructions in front of any chunk's MFMAs.  One split per
        // chunk instead of three in each of the last three, halo loads from chunk 0 or 2, and the chunk's other instructions
        // dealt out BETWEEN its MFMAs (sched_group_barrier) all measured equal or worse.
        constexpr int ST = G::NCH >= 14 ? (ACC_IN ? 1 : 7) : (G::NCH == 13 ? 6 : SH);      // (13 chunks: the composed 5x5 first layer, -6 %)
        constexpr int US = 3, UL = NGRP == 1 ? 4 : 3, UI = NGRP == 1 ? 5 : 3;      // UPS: load stride, load -> use, use stride (in chunks)
        Src qn;
        const int yy0 = cur.y0 + 2 * gw, xx = cur.x0 + n;
        unsigned po[2], so[2];
        f32x4 side[2][MT];
        __amdgpu_buffer_rsrc_t pr;
        auto addresses = [&]() {
            locate(t + t_step, nxt);
            qn = source(nxt, t + t_step < t_end);
            if constexpr (SC) amax_fetch(nxt.b, t + t_step < t_end);
        };
        auto addresses2 = [&]() {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const bool ok = yy0 + nt < a.H && xx < a.W;
                po[nt] = ok ? (unsigned)((((yy0 + nt) * a.W + xx) * kF + 4 * g) * 4) : 0x80000000u;
                so[nt] = ok ? (unsigned)((((yy0 + nt + a.oy) * a.Wout + xx + a.ox) * kF + 4 * g) * 4) : 0x80000000u;
            }
            if constexpr (EPI == EPI_POOL) {
                const int pr_ = (cur.y0 >> 1) + gw, pc = xx >> 1;
                const bool ok = !(n & 1) && pr_ < a.Hout && pc < a.Wout;
                so[0] = ok ? (unsigned)(((pr_ * a.Wout + pc) * kF + 4 * g) * 4) : 0x80000000u;
            }
            pr = __builtin_amdgcn_make_buffer_rsrc((void*)((ACC_IN ? a.acc_in : a.out) + (size_t)cur.b * a.H * a.W * kF), 0, map_bytes,
                                                   0x00020000);
        };
        STAMP(2);

        // ---- 14 chunks x 18 MFMAs, and between the chunks: a store of the last tile, a side load of this one, a halo
        // load of the next one (nine chunks from SH on), the split of what those loads brought (the last three chunks)
        f32x4 acc[2][MT];
        read_frags(0, 0);
#pragma unroll
        for (int j = 0; j < G::NCH; ++j) {
            const int cb = j & 1;
            if (j + 1 < G::NCH) read_frags(cb ^ 1, j + 1);
            if (j == SH) addresses();
            if (j == (G::NCH >= 14 ? 6 : SH)) addresses2();      // this tile's store / side-load offsets: first needed at chunk 7 (0.4 % over chunk 1)
            if (j >= ST && j - ST < NOUT) store_prev(j - ST);
            if constexpr (ACC_IN) {
                // (behind the previous tile's stores, which take the first six chunks: one memory instruction of each kind per chunk)
                constexpr int SL = G::NCH >= 14 ? 7 : SH;
                if (j >= SL && j - SL < 2 * MT) side[(j - SL) / MT][(j - SL) % MT] = bload(pr, po[(j - SL) / MT], 64 * (mt0 + (j - SL) % MT));
            }
            if constexpr (SC) {
                if (j == (UPS ? SH + UL : G::NCH - 1 - (G::NR - 1) / 3)) scale_from(ab_nxt, sc_nxt, inv_nxt);      // the chunk of the first split
            }
            if constexpr (UPS) {
                // one workgroup-wide tile: the two items' loads at chunks 1 and 4, their interpolation and split at 5 and 10 (2 % faster
                // than loads at 1, 2 and interpolation at 6, 8: profiles/r04_chunk_loop_schedule_ab.txt); a group's tile: three items,
                // each loaded when the one before has been interpolated, so that only one item's sixteen source registers are alive
                // at a time (all three at once did not fit: scratch)
#pragma unroll
                for (int r0 = 0; r0 < UNR; ++r0) {
                    if (j == SH + US * r0) fetch_ups(qn, r0);
                    if (j == SH + UL + UI * r0) {
                        interp_ups(qn, r0, sc_nxt, scaled_tag);
#pragma unroll
                        for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(ushi[r0][e]), "+v"(uslo[r0][e]));      // here, not behind the barrier
                    }
                }
            } else {
                if (j >= SH && j - SH < G::NR) fetch_round(qn, j - SH);
#pragma unroll
                for (int r0 = 0; r0 < G::NR; ++r0)
                    if (j == G::NCH - 1 - (G::NR - 1 - r0) / 3) {
                        split4<SC>(pre[r0], sc_nxt, shi[r0], slo[r0]);
                        asm volatile("" : "+v"(shi[r0]), "+v"(slo[r0]));      // here, not sunk behind the barrier next to its use
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const int ha = p == 1 ? 1 : 0, hb = p == 0 ? 1 : 0;      // hi.lo, lo.hi, then hi.hi
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)      // (filter fragment constant over two MFMAs in a row: 0.3 % over the other nest)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        const f32x4 c = (j == 0 && p == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[nt][mt];
                        acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Af[cb][mt][ha], Bf[cb][nt][hb], c, 0, 0, 0);
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = G::NCH - ST; j < NOUT; ++j) store_prev(j);      // CIN 16: five chunks
        STAMP(3);

        // ---- epilogue: scale back, bias / partial sums, activation; the 48-channel results wait in registers for the
        // next tile's chunks
        f32x4 v[2][MT];
        float m3 = 0.f;        // EPI_RELU_OUT3: max |output frame| over this lane's two pixels
        const float ws = SC ? a.wscale * inv_cur : a.wscale;      // the filters' and the map's powers of two, undone together
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const f32x4 bias = *reinterpret_cast<const f32x4*>(Pl + 16 * (mt0 + mt) + 4 * g);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const f32x4 base = ACC_IN ? side[nt][mt] : bias;
                v[nt][mt] = __builtin_elementwise_fma(acc[nt][mt], f32x4{ws, ws, ws, ws}, base);
            }
        }
        if constexpr (EPI == EPI_POOL) {
            // MaxPool2d(2) of the un-activated conv output: the wave's two rows are one pooling row pair, columns n, n ^ 1
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float q = fmaxf(v[0][mt][e], v[1][mt][e]);
                    outv[mt][e] = lane_xor_max<1>(q);
                }
        } else {
            float o3[2][3];
            m3 = 0.f;
            if constexpr (EPI == EPI_RELU_OUT3) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) o3[nt][0] = o3[nt][1] = o3[nt][2] = 0.f;
            }
            __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(
                (void*)((EPI == EPI_RELU_ADD2 ? a.res1 : a.out) + (size_t)cur.b * a.H * a.W * kF), 0, map_bytes, 0x00020000);
            __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(
                (void*)((EPI == EPI_RELU_ADD2 ? a.res2 : a.out) + (size_t)cur.b * a.H * a.W * kF), 0, map_bytes, 0x00020000);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    f32x4 x = v[nt][mt];
                    if constexpr (EPI == EPI_RELU || EPI == EPI_RELU_ADD2 || EPI == EPI_RELU_OUT3) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) x[e] = __builtin_amdgcn_fmed3f(x[e], 0.f, __builtin_inff());      // ONE instruction (fmaxf: a canonicalising v_max in front)
                    }
                    if constexpr (EPI == EPI_RELU_ADD2) x = (bload(r1, po[nt], 64 * (mt0 + mt)) + bload(r2, po[nt], 64 * (mt0 + mt))) + x;   // e3 + d1 + d2 (unet.py:563-566)
                    outv[nt * MT + mt] = x;
                    if constexpr (EPI == EPI_RELU_OUT3) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const f32x4 w = *reinterpret_cast<const f32x4*>(Pl + kF + c * kF + 16 * mt + 4 * g);
                            o3[nt][c] += (x[0] * w[0] + x[1] * w[1]) + (x[2] * w[2] + x[3] * w[3]);
                        }
                    }
                }
            if constexpr (EPI == EPI_RELU_OUT3) {
                const size_t hw = (size_t)a.H * a.W;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    float tq[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        float q = o3[nt][c];
                        q = lane_xor_add<16>(q);
                        q = lane_xor_add<32>(q);
                        tq[c] = q + Pl[4 * kF + c];
                    }
                    // (the output frame's maximum joins the features' in this launch's words: the next step's input bound reads them)
                    if (yy0 + nt < a.H && xx < a.W) m3 = fmaxf(m3, fmaxf(fmaxf(fabsf(tq[0]), fabsf(tq[1])), fabsf(tq[2])));
                    if (g == 0 && yy0 + nt < a.H && xx < a.W) {
                        const size_t pidx = (size_t)(yy0 + nt) * a.W + xx;
#pragma unroll
                        for (int c = 0; c < 3; ++c) a.out3_nchw[((size_t)cur.b * 3 + c) * hw + pidx] = tq[c];
                        if (a.out3_nhwc4)
                            reinterpret_cast<f32x4*>(a.out3_nhwc4)[(size_t)cur.b * hw + pidx] = f32x4{tq[0], tq[1], tq[2], 0.f};
                    }
                }
            }
        }
        if (a.amax_out) {      // max |x| of what this tile stores (pixels outside the map do not count), per sequence
            if (cur.b != amx_b) {          // (every wave of the group is on the same tile: they all come through here together)
                if (amx_b >= 0) {
                    const unsigned wm = wave_max_u32(__float_as_uint(amx));
                    if (lane == 0) Rl[gw] = wm;
                    pend_b = amx_b;        // the group's first wave sends it off behind the barrier below
                }
                amx = 0.f;
                amx_b = cur.b;
            }
            // (v_max3_f32 with |.| operands, two values per instruction; outv comes from vector instructions, not straight out of
            // an MFMA, so inline asm is safe here)
            float m0 = 0.f, m1 = 0.f;
#pragma unroll
            for (int i = 0; i < NOUT; ++i) {
                float& m = (EPI == EPI_POOL || i < MT) ? m0 : m1;
                asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(m) : "v"(outv[i][0]), "v"(outv[i][1]));
                asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(m) : "v"(outv[i][2]), "v"(outv[i][3]));
            }
            if (so[0] == 0x80000000u) m0 = 0.f;
            if (so[1] == 0x80000000u) m1 = 0.f;
            asm("v_max3_f32 %0, %0, %1, %2" : "+v"(amx) : "v"(m0), "v"(m1));
            amx = fmaxf(amx, m3);
        }
        so_prev[0] = so[0];
        so_prev[1] = so[1];
        orr_prev = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (size_t)cur.b * a.Hout * a.Wout * kF), 0, out_bytes, 0x00020000);
        STAMP(4);
        gsync();               // every wave of the group has read its last fragment of this tile: the next one may be staged
        STAMP(5);
        if (pend_b >= 0) {     // the finished sequence's maximum: one atomic for the group
            if (gw == 0) amax_send(pend_b);
            pend_b = -1;
        }
        write_tile();
        STAMP(6);
#ifdef RVDD_STAMPS
        ++st_n;
#endif
        t += t_step;
        cur = nxt;
        inv_cur = inv_nxt;
    }
    // the last tile's results, and behind them (the stores are on their way while the waves meet) the last sequence's maximum
#pragma unroll
    for (int j = 0; j < NOUT; ++j) store_prev(j);
    if (a.amax_out && amx_b >= 0) {
        // (behind a barrier of its own: the group's first wave may still be reading Rl for the sequence that ended with the last
        // tile -- amax_send(pend_b) above -- when the other waves arrive here)
        gsync();
        const unsigned wm = wave_max_u32(__float_as_uint(amx));
        if (lane == 0) Rl[gw] = wm;
        gsync();
        if (gw == 0) amax_send(amx_b);
    }
    STAMP_FLUSH;
    };      // tile_loop
    if (any_scaled) tile_loop(std::true_type{});
    else tile_loop(std::false_type{});
}

bool g_conv3x3h_cout_split = true;      // conv3x3h_set_cout_split: false = every launch forms all 48 output channels per workgroup (A/B, tests)
template <int CIN, int EPI, bool ACC_IN, bool UPS, int NGRP, int KS = 3, int MT = 3>
hipError_t launch_g(const ConvArgs& a0, hipStream_t s) {
    static std::atomic<uint64_t> attr_done{0};
    using G = HGeo<CIN, NGRP, KS, MT>;
    constexpr int TH = G::TH;
    void (*kern)(ConvArgs) = conv3x3h_kernel<CIN, EPI, ACC_IN, UPS, NGRP, KS, MT>;
    if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(kern), G::LDS_BYTES, attr_done); e != hipSuccess) return e;
    ConvArgs a = a0;
    a.tiles_x = (a.W + TW - 1) / TW;
    a.tiles_y = (a.H + TH - 1) / TH;
    a.ntiles = a.B * a.tiles_x * a.tiles_y;
    const int cus = current_device_cus();
    const int want = (a.ntiles + NGRP - 1) / NGRP;         // one tile per group at least
    const int grid = want < cus ? want : cus;
    hipLaunchKernelGGL(kern, dim3(grid, MT == 3 ? 1 : 3), dim3(NTHREADS), G::LDS_BYTES, s, a);
    return hipGetLastError();
}
// Does a launch of this many 16x16 tiles take the output-channel split (MT = 1)?  At most a third of a tile per CU, i.e. every
// workgroup of the split launch has ONE tile and a CU of its own.  (Measured, profiles/r05y_c1_cout_split.txt: up to one tile per
// CU -- 85 walkers of three tiles each at 256 tiles -- loses what the coarse levels gain: the halo fetch of a tile does not shrink
// with the output channels.)
bool cout_split_applies(const ConvArgs& a) {
    const int ntiles = a.B * ((a.W + TW - 1) / TW) * ((a.H + 15) / 16);
    return g_conv3x3h_cout_split && 3 * ntiles <= current_device_cus();
}
int g_conv3x3h_groups = 1;
template <int CIN, int EPI, bool ACC_IN, bool UPS = false>
hipError_t launch_h(const ConvArgs& a, hipStream_t s) {
#ifdef RVDD_CONV_GROUPS2
    if (g_conv3x3h_groups == 2) return launch_g<CIN, EPI, ACC_IN, UPS, 2>(a, s);
#endif
    if constexpr (CIN == 48 && EPI != EPI_RELU_OUT3) {
        if (cout_split_applies(a)) return launch_g<CIN, EPI, ACC_IN, UPS, 1, 3, 1>(a, s);
    }
    return launch_g<CIN, EPI, ACC_IN, UPS, 1>(a, s);
}

}  // namespace

size_t conv3x3h_weight_bytes(int cin) { return cin == 48 ? HGeo<48, 1>::W_BYTES : HGeo<16, 1>::W_BYTES; }
size_t conv5x5h_weight_bytes() { return HGeo<16, 1, 5>::W_BYTES; }

// the composed 5x5 conv of the 16-channel network input (see HGeo): a.w = the bank arranged by arrange_conv3x3h(.., ks = 5), no
// activation, a.bias = the composed bias; a.out = the partial sums the second source's pass starts from
hipError_t launch_conv5x5h_c16(const ConvArgs& a, hipStream_t s) {
    if (a.B <= 0 || a.H <= 0 || a.W <= 0) return hipSuccess;
    if ((size_t)a.H * a.W * kF * 4 >= 0x80000000ull || a.acc_in || a.ups) return hipErrorInvalidValue;
    return launch_g<16, EPI_NONE, false, false, 1, 5>(a, s);
}
void conv3x3h_set_groups(int g) { g_conv3x3h_groups = g == 1 ? 1 : 2; }
void conv3x3h_set_cout_split(bool on) { g_conv3x3h_cout_split = on; }

hipError_t launch_conv3x3h(const ConvArgs& a, int cin, int epi, hipStream_t s) {
    if (a.B <= 0 || a.H <= 0 || a.W <= 0) return hipSuccess;
    // every map is addressed with one 32-bit byte offset per image whose out-of-image sentinel is 2^31
    if ((size_t)a.H * a.W * kF * 4 >= 0x80000000ull || (size_t)a.Hout * a.Wout * kF * 4 >= 0x80000000ull) return hipErrorInvalidValue;
    const bool acc = a.acc_in != nullptr;
    if (cin == 16) {      // the zero-padded network input (6 or 9 real channels)
        if (acc || a.ups) return hipErrorInvalidValue;
        if (epi == EPI_NONE) return launch_h<16, EPI_NONE, false>(a, s);
        if (epi == EPI_RELU) return launch_h<16, EPI_RELU, false>(a, s);
        return hipErrorInvalidValue;
    }
    if (cin != 48) return hipErrorInvalidValue;
    if (a.ups) {          // a.in = the map to upsample, [B][H/2][W/2][48]
        if (acc || epi != EPI_RELU || (a.H & 1) || (a.W & 1)) return hipErrorInvalidValue;
        return launch_h<48, EPI_RELU, false, true>(a, s);
    }
    switch (epi) {
        case EPI_NONE:
            return acc ? launch_h<48, EPI_NONE, true>(a, s) : launch_h<48, EPI_NONE, false>(a, s);
        case EPI_RELU:
            return acc ? launch_h<48, EPI_RELU, true>(a, s) : launch_h<48, EPI_RELU, false>(a, s);
        case EPI_POOL:
            return acc ? hipErrorInvalidValue : launch_h<48, EPI_POOL, false>(a, s);
        case EPI_RELU_ADD2:
            return acc ? hipErrorInvalidValue : launch_h<48, EPI_RELU_ADD2, false>(a, s);
        case EPI_RELU_OUT3:
            return acc ? hipErrorInvalidValue : launch_h<48, EPI_RELU_OUT3, false>(a, s);
    }
    return hipErrorInvalidValue;
}
