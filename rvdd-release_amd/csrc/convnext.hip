// ConvNeXt ConvBlock (networks/new_unet.py:74-103) on gfx950:
//   x -> [proj 1x1 when Cin != 48] -> r = dwconv7x7 -> LayerNorm_C -> 1x1 48->192
//     -> GELU(erf) -> 1x1 192->48 ; out = x + layerscale * r
//
// Three kernels, all NHWC fp32:
//   proj1x1_kernel  : 1x1 projection (9|6 -> 48 from the padded 16-channel network
//                     input, or 96 -> 48 from two 48-channel maps = virtual concat)
//                     on f32 MFMA with the whole weight matrix held in registers.
//   dwln_kernel     : depth-wise 7x7 (+bias) and the per-pixel channel LayerNorm
//                     (biased variance, eps 1e-6, :12-28) from an LDS halo tile.
//   mlp_kernel      : 48->192 (+bias) -> exact GELU -> 192->48 (+bias), layerscale,
//                     residual.  Both GEMMs run on v_mfma_f32_16x16x4_f32 in the
//                     orientation D[channel][pixel]; the accumulators of the first
//                     GEMM (lane = pixel, 4 consecutive hidden channels per register
//                     quad) ARE the B fragments of the second one, so the 192-channel
//                     hidden map never leaves the register file (the reference writes
//                     and re-reads it: 708 MB per block at 720p).
//
// Lane <-> data map shared by all three (the MFMA B-operand map): lane l owns pixel
// l&15 of its 16-pixel group and, in every 16-channel chunk j, channels 16j+4g..+3
// with g = l>>4.
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
constexpr size_t E_LDS_BYTES = (size_t)(D_W_FLOATS + 2 * E_BUF_FLOATS) * 4;

__global__ __launch_bounds__(256, 2) void dwln_kernel(const float* __restrict__ x, const float* __restrict__ dw_w,
                                                      const float* __restrict__ dw_b, const float* __restrict__ ln_w,
                                                      const float* __restrict__ ln_b, float* __restrict__ out,
                                                      int B, int H, int W, int tiles_x, int tiles_y, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wl = smem;                        // [49][48]
    float* Tl = smem + D_W_FLOATS;           // two chunk buffers
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // Persistent workgroups (two per CU).  Blocks that share an XCD (blockIdx & 7 under round-robin placement:
    // speed only) walk one contiguous band of tiles side by side, so the halo a tile shares with its neighbours
    // is served by that XCD's L2.
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

    // lane -> (row of the wave's four rows, quad of four pixels, channel group); popcount parity puts the 16
    // lanes of each ds_read_b128 lane group {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... in one row
    const int g = lane & 3;
    const int idx = lane >> 2;
    const int quad = idx & 3;
    const int row = wave * 4 + ((idx >> 3) << 1) + (__builtin_popcount(idx & 7) & 1);

    // which halo pixel / channel group each lane fetches for its (at most 9) DMA pieces: fixed for the kernel
    int piece_yx[9];
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
    TilePos cur = locate(tile);
    dma_chunk(cur, 0, 0);
    for (int q = tid; q < D_W_FLOATS / 4; q += 256)
        reinterpret_cast<f32x4*>(Wl)[q] = reinterpret_cast<const f32x4*>(dw_w)[q];

    // read addresses: pixel (row + ky, 4 quad + dx) -> R = (row + ky) * 6 + quad + (dx >> 2); float offset =
    // R * 64 + (((dx & 3) ^ (R & 3)) * 4 + g) * 4.  (R & 3) = (2 row + quad + 2 ky + (dx >> 2)) & 3: sixteen
    // registers base + swizzle[c][d] for c = (2 ky + (dx >> 2)) & 3, d = dx & 3; the rest is an immediate.
    const int t0 = (2 * row + quad) & 3;
    int rd0[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int d = 0; d < 4; ++d) rd0[c][d] = (row * 6 + quad) * 64 + ((d ^ ((t0 + c) & 3)) * 4 + g) * 4;

    int par = 0;                 // buffer that holds (or receives) the chunk about to be used
#pragma unroll 1
    for (;;) {
        const int next_tile = tile + stride;
        const bool more = next_tile < band_end;
        const TilePos nxt = locate(more ? next_tile : tile);
        f32x4 acc[4][3];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = *reinterpret_cast<const f32x4*>(dw_b + 16 * j + 4 * g);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                              // this chunk landed; every wave is done with the other buffer
            if (j + 1 < 3) dma_chunk(cur, j + 1, par ^ 1);
            else if (more) dma_chunk(nxt, 0, par ^ 1);    // the next tile's first chunk rides under this tile's last
            const float* tb = Tl + par * E_BUF_FLOATS;
            const float* wb = Wl + 16 * j + 4 * g;
            // rows are software-pipelined: the 17 reads of row ky + 1 are in flight under the 28 float4 FMAs of row ky
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
                // one filter row at a time: left alone, instruction selection emits the (unchained) FMAs of a chunk
                // behind ALL of its 119 LDS reads and spills a thousand registers.  The empty asm makes this row's
                // sums a side effect that is ordered with the reads that follow.
                asm volatile("" : "+v"(acc[0][j]), "+v"(acc[1][j]), "+v"(acc[2][j]), "+v"(acc[3][j])::"memory");
            }
            par ^= 1;
        }
        // LayerNorm over the 48 channels of each pixel: 12 here, the rest in lanes l^1, l^2, l^3
        const int y = cur.y0 + row;
        f32x4 lw[3], lb[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            lw[j] = *reinterpret_cast<const f32x4*>(ln_w + 16 * j + 4 * g);
            lb[j] = *reinterpret_cast<const f32x4*>(ln_b + 16 * j + 4 * g);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float sm = 0.f;
#pragma unroll
            for (int j = 0; j < 3; ++j) sm += (acc[i][j][0] + acc[i][j][1]) + (acc[i][j][2] + acc[i][j][3]);
            sm += __shfl_xor(sm, 1);
            sm += __shfl_xor(sm, 2);
            const float u = sm / 48.f;
            float v2 = 0.f;
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float d = acc[i][j][r] - u;
                    v2 += d * d;
                }
            v2 += __shfl_xor(v2, 1);
            v2 += __shfl_xor(v2, 2);
            // (x - u) / sqrt(var + eps) as (x - u) * (1 / sqrt(...)): one correctly rounded division per pixel instead
            // of twelve per lane (each ~10 instructions); the quotient moves by at most one ulp
            const float rden = 1.0f / sqrtf(v2 / 48.f + 1e-6f);
            const int xx = cur.x0 + 4 * quad + i;
            if (y < H && xx < W) {
                float* o = out + (((size_t)cur.b * H + y) * W + xx) * kF + 4 * g;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    f32x4 r;
#pragma unroll
                    for (int k = 0; k < 4; ++k) r[k] = lw[j][k] * ((acc[i][j][k] - u) * rden) + lb[j][k];
                    *reinterpret_cast<f32x4*>(o + 16 * j) = r;
                }
            }
        }
        if (!more) break;
        tile = next_tile;
        cur = nxt;
    }
}

// ---------------------------------------------------------------------- MLP --
// LDS: fc1 arranged [j(3)][m(12)][lane = 16g+lr][i] = W1[16m+lr][16j+4g+i]     (9216 floats)
//      fc2 arranged [m(12)][mo(3)][lane = 16g+lr][r] = W2[16mo+lr][16m+4g+r]   (9216 floats)
constexpr int M_W_FLOATS = 192 * 48;
constexpr size_t M_LDS_BYTES = (size_t)2 * M_W_FLOATS * 4;
constexpr int M_NPB = 1;                       // 16-pixel groups per wave iteration

// nn.GELU() default = exact erf form (networks/new_unet.py:94): 0.5 v (1 + erf(v/sqrt2)).
// On gfx950 the f32 MFMA and the VALU share the SIMD's fp32 lanes, so the 192 GELUs per pixel are not
// hidden behind the MFMAs: this single-branch form costs 16 instructions instead of 27 for a
// two-branch <1-ulp erf (and ocml's erff, inlined 48x per lane, spilled 260 registers).  erf(v/sqrt2) = sign(v) (1 - 2^(t P(t))), t = min(|v|, 6.36), P a
// degree-7 polynomial fitted (weighted least squares, host, float32 Horner) to log2 erfc(t/sqrt2)/t:
// max |GELU error| 4.3e-7 on [-9, 9] against the double-precision erf form (fp32 rounding of the
// exact form is ~2e-7 there).
__device__ __forceinline__ float gelu_erf(float v) {
    const float t = fminf(fabsf(v), 6.36f);
    float p = -2.116853238476324e-06f;
    p = fmaf(p, t, 3.1051968107931316e-05f);
    p = fmaf(p, t, -0.0001479804632253945f);
    p = fmaf(p, t, -0.00022579463256988674f);
    p = fmaf(p, t, 0.007174866273999214f);
    p = fmaf(p, t, -0.05256997048854828f);
    p = fmaf(p, t, -0.45918503403663635f);
    p = fmaf(p, t, -1.1511077880859375f);
    const float e = copysignf(1.0f - __builtin_amdgcn_exp2f(t * p), v);
    const float hv = 0.5f * v;
    return fmaf(hv, e, hv);
}
// the same function on two values at once: the polynomial and the two products on v_pk_fma_f32 / v_pk_mul_f32
// (bit-identical results; a packed op occupies the fp32 lanes as long as two scalar ones, but takes one issue slot
// between MFMAs instead of two)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 v) {
    // min(|v|, 6.36) in ONE instruction each (the |.| source modifier): fminf(fabsf()) compiles to a canonicalising
    // v_max_f32 |v|, |v| in front of the v_min_f32, 48 extra vector instructions per 16 pixels beside the MFMAs
    f32x2 t;
    const float cap = 6.36f;
    asm("v_min_f32 %0, |%1|, %2" : "=v"(t[0]) : "v"(v[0]), "v"(cap));
    asm("v_min_f32 %0, |%1|, %2" : "=v"(t[1]) : "v"(v[1]), "v"(cap));
    auto K = [](float c) { return f32x2{c, c}; };
    f32x2 p = K(-2.116853238476324e-06f);
    p = __builtin_elementwise_fma(p, t, K(3.1051968107931316e-05f));
    p = __builtin_elementwise_fma(p, t, K(-0.0001479804632253945f));
    p = __builtin_elementwise_fma(p, t, K(-0.00022579463256988674f));
    p = __builtin_elementwise_fma(p, t, K(0.007174866273999214f));
    p = __builtin_elementwise_fma(p, t, K(-0.05256997048854828f));
    p = __builtin_elementwise_fma(p, t, K(-0.45918503403663635f));
    p = __builtin_elementwise_fma(p, t, K(-1.1511077880859375f));
    const f32x2 tp = t * p;
    const f32x2 e = {copysignf(1.0f - __builtin_amdgcn_exp2f(tp[0]), v[0]), copysignf(1.0f - __builtin_amdgcn_exp2f(tp[1]), v[1])};
    const f32x2 hv = v * K(0.5f);
    return __builtin_elementwise_fma(hv, e, hv);
}

__global__ __launch_bounds__(768) void mlp_kernel(const float* __restrict__ ln, const float* __restrict__ x,
                                                     const float* __restrict__ fc1_w,
                                                     const float* __restrict__ fc1_b,
                                                     const float* __restrict__ fc2_w,
                                                     const float* __restrict__ fc2_b,
                                                     const float* __restrict__ ls, float* __restrict__ out,
                                                     long npix) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* W1 = smem;
    float* W2 = smem + M_W_FLOATS;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, g = lane >> 4;
    {
        __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)fc1_w, 0, M_W_FLOATS * 4, 0x00020000);
        __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void*)fc2_w, 0, M_W_FLOATS * 4, 0x00020000);
        const int nw = blockDim.x >> 6;
        for (int k = wave; k < M_W_FLOATS / 256; k += nw) {
            dma16(r1, W1 + k * 256, (unsigned)(k * 1024 + lane * 16));
            dma16(r2, W2 + k * 256, (unsigned)(k * 1024 + lane * 16));
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const float* w1b = W1 + lane * 4;      // lane-linear fragments: each ds_read_b128 lane group covers one bank row
    const float* w2b = W2 + lane * 4;
    const long nblk = (npix + 16 * M_NPB - 1) / (16 * M_NPB);
    const long wave_global = (long)blockIdx.x * (blockDim.x >> 6) + wave;
    const long nwaves = (long)gridDim.x * (blockDim.x >> 6);
    for (long blk = wave_global; blk < nblk; blk += nwaves) {
        long pix[M_NPB];
        f32x4 xb[M_NPB][3];
#pragma unroll
        for (int n = 0; n < M_NPB; ++n) {
            pix[n] = (blk * M_NPB + n) * 16 + lr;
            const long pc = pix[n] < npix ? pix[n] : npix - 1;
#pragma unroll
            for (int j = 0; j < 3; ++j) xb[n][j] = *reinterpret_cast<const f32x4*>(ln + pc * kF + 16 * j + 4 * g);
        }
        // ---- fc1 + bias + GELU: hid[m][n][r] = hidden channel 16m+4g+r of pixel lr of group n.
        // Two hidden blocks at a time: two independent accumulator chains per pixel group cover
        // the 40-cycle dependent latency of the 32-cycle MFMA.
        f32x4 hid[12][M_NPB];
#pragma unroll
        for (int m = 0; m < 12; m += 2) {
            const f32x4 b1a = *reinterpret_cast<const f32x4*>(fc1_b + 16 * m + 4 * g);
            const f32x4 b1b = *reinterpret_cast<const f32x4*>(fc1_b + 16 * (m + 1) + 4 * g);
#pragma unroll
            for (int n = 0; n < M_NPB; ++n) {
                hid[m][n] = b1a;
                hid[m + 1][n] = b1b;
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const f32x4 wa0 = *reinterpret_cast<const f32x4*>(w1b + (j * 12 + m) * 256);
                const f32x4 wa1 = *reinterpret_cast<const f32x4*>(w1b + (j * 12 + m + 1) * 256);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int n = 0; n < M_NPB; ++n) {
                        hid[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa0[i], xb[n][j][i], hid[m][n], 0, 0, 0);
                        hid[m + 1][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa1[i], xb[n][j][i], hid[m + 1][n], 0, 0, 0);
                    }
            }
#pragma unroll
            for (int n = 0; n < M_NPB; ++n)
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    const f32x2 a = gelu_erf2(f32x2{hid[m][n][r], hid[m][n][r + 1]});
                    const f32x2 c = gelu_erf2(f32x2{hid[m + 1][n][r], hid[m + 1][n][r + 1]});
                    hid[m][n][r] = a[0];
                    hid[m][n][r + 1] = a[1];
                    hid[m + 1][n][r] = c[0];
                    hid[m + 1][n][r + 1] = c[1];
                }
            __builtin_amdgcn_sched_barrier(0);   // keep hipcc from hoisting every fragment read (it spills)
        }
        // ---- fc2: the hidden accumulators are the B fragments (k-slot g of step (m,r) = 16m+4g+r)
        f32x4 acc[3][M_NPB];
#pragma unroll
        for (int mo = 0; mo < 3; ++mo) {
            const f32x4 b2 = *reinterpret_cast<const f32x4*>(fc2_b + 16 * mo + 4 * g);
#pragma unroll
            for (int n = 0; n < M_NPB; ++n) acc[mo][n] = b2;
        }
#pragma unroll
        for (int m = 0; m < 12; ++m)
#pragma unroll
            for (int mo = 0; mo < 3; ++mo) {
                const f32x4 wa = *reinterpret_cast<const f32x4*>(w2b + (m * 3 + mo) * 256);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int n = 0; n < M_NPB; ++n)
                        acc[mo][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[r], hid[m][n][r], acc[mo][n], 0, 0, 0);
                if (mo == 2) __builtin_amdgcn_sched_barrier(0);
            }
        // ---- out = x + layerscale * r
#pragma unroll
        for (int n = 0; n < M_NPB; ++n) {
            if (pix[n] < npix) {
#pragma unroll
                for (int mo = 0; mo < 3; ++mo) {
                    const size_t o = (size_t)pix[n] * kF + 16 * mo + 4 * g;
                    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + o);
                    const f32x4 lv = *reinterpret_cast<const f32x4*>(ls + 16 * mo + 4 * g);
                    *reinterpret_cast<f32x4*>(out + o) = xv + lv * acc[mo][n];
                }
            }
        }
    }
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

hipError_t launch_next_dwln(const float* x, float* ln_out, const NextBlockW& w, int B, int H, int W, hipStream_t s) {
    if ((size_t)H * W * kF * 4 >= 0x80000000ull) return hipErrorInvalidValue;
    static std::atomic<uint64_t> attr{0};
    if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(dwln_kernel), E_LDS_BYTES, attr); e != hipSuccess)
        return e;
    const int tx = (W + E_TW - 1) / E_TW, ty = (H + E_TH - 1) / E_TH;
    const int ntiles = B * tx * ty;
    // persistent: two workgroups per CU (LDS), never more blocks than tiles; the XCD band map needs a multiple of 8
    const int grid = ((std::min(ntiles, 2 * num_cus()) + 7) / 8) * 8;
    hipLaunchKernelGGL(dwln_kernel, dim3(grid), dim3(256), E_LDS_BYTES, s, x, w.dw_w, w.dw_b, w.ln_w, w.ln_b, ln_out, B, H, W,
                       tx, ty, ntiles);
    return hipGetLastError();
}

hipError_t launch_next_mlp(const float* ln, const float* x, float* out, const NextBlockW& w, int64_t npix,
                           hipStream_t s) {
    static std::atomic<uint64_t> attr{0};
    if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(mlp_kernel), M_LDS_BYTES, attr); e != hipSuccess)
        return e;
    if (npix <= 0) return hipSuccess;
    const long nblk = (npix + 16 * M_NPB - 1) / (16 * M_NPB);
    // one workgroup of 12 waves per CU: three waves per SIMD share one LDS copy of the two weight matrices
    constexpr int kMlpWaves = 12;
    long blocks = (nblk + kMlpWaves - 1) / kMlpWaves;
    const long cap = (long)num_cus();
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(mlp_kernel, dim3((unsigned)blocks), dim3(64 * kMlpWaves), M_LDS_BYTES, s, ln, x, w.fc1_w, w.fc1_b, w.fc2_w,
                       w.fc2_b, w.ls, out, (long)npix);
    return hipGetLastError();
}

hipError_t launch_pad_copy(const float* src, float* dst, int B, int h, int w, int H, int W, int oy, int ox,
                           hipStream_t s) {
    const size_t n = (size_t)B * H * W * 12;
    hipLaunchKernelGGL(pad_copy_kernel, dim3((unsigned)((n + 191) / 192)), dim3(192), 0, s, src, dst, B, h, w, H, W,
                       oy, ox);
    return hipGetLastError();
}
