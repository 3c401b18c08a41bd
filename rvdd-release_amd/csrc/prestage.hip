// HBM-bound stages around the U-Net: Hamilton-Adams demosaic, bicubic backward
// warp with the x2 flow upsample fused, map reshaping (bilinear x2, max-pool,
// NCHW<->NHWC), the final 1x1 conv and the loss reduction.
// This translation unit is compiled with -ffp-contract=off: the demosaic has
// hard sign() selections (util/Hamilton_Adam_demo.py:138-139,168-169) and the
// warp round-trips its coordinates through the [-1,1] normalisation in fp32
// (util/flow_utils.py:93-94), so the arithmetic is kept operation for
// operation as the reference's ATen ops evaluate it.
#include "rvdd_internal.h"

namespace {

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ float signf(float v) { return (float)((v > 0.f) - (v < 0.f)); }

// ------------------------------------------------------------------ demosaic --
// CFA of the packed GBRG frame (util/Hamilton_Adam_demo.py:226-234), replicate
// padded: coordinates are clamped BEFORE the lookup.
struct Cfa {
    const float* raw;  // [4][h][w] of one frame
    int h, w, H, W;
    __device__ __forceinline__ float at(int y, int x) const {
        y = clampi(y, 0, H - 1);
        x = clampi(x, 0, W - 1);
        return raw[((size_t)(((y & 1) << 1) | (x & 1)) * h + (y >> 1)) * w + (x >> 1)];
    }
    // sparse colour plane `site` (0 = G(e,e), 1 = B(e,o), 2 = R(o,e), 3 = G(o,o)), replicate padded
    __device__ __forceinline__ float plane(int y, int x, int site) const {
        y = clampi(y, 0, H - 1);
        x = clampi(x, 0, W - 1);
        const int s = ((y & 1) << 1) | (x & 1);
        return s == site ? raw[((size_t)s * h + (y >> 1)) * w + (x >> 1)] : 0.f;
    }
};

__device__ __forceinline__ float ha_green_at(const Cfa& c, int y, int x);
// algo1 (util/Hamilton_Adam_demo.py:123-142)
// rbs = floats from one sequence's raw frame to the next one's (4hw when dense; more for a channel slice of a wider tensor)
__global__ void ha_green_kernel(const float* __restrict__ raw, float* __restrict__ green, int n, int h,
                                int w, int64_t rbs) {
    const int H = 2 * h, W = 2 * w;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)n * H * W) return;
    const int x = idx % W;
    const int y = (idx / W) % H;
    const int b = idx / ((size_t)W * H);
    Cfa c{raw + (size_t)b * rbs, h, w, H, W};
    green[idx] = ha_green_at(c, y, x);
}
__device__ __forceinline__ float ha_green_at(const Cfa& c, int y, int x) {
    const float cc = c.at(y, x);
    float gval;
    if (((y ^ x) & 1) == 0) {
        gval = cc;  // measured green
    } else {
        const float l1 = c.at(y, x - 1), r1 = c.at(y, x + 1), l2 = c.at(y, x - 2), r2 = c.at(y, x + 2);
        const float u1 = c.at(y - 1, x), d1 = c.at(y + 1, x), u2 = c.at(y - 2, x), d2 = c.at(y + 2, x);
        const float Kh = 0.5f * l1 + 0.5f * r1;
        const float Kv = 0.5f * u1 + 0.5f * d1;
        const float Dh = (l2 + (-2.f) * cc) + r2;
        const float Dv = (u2 + (-2.f) * cc) + d2;
        const float Fh = l1 + (-1.f) * r1;
        const float Fv = u1 + (-1.f) * d1;
        const float rawh = Kh - Dh / 4.f;
        const float rawv = Kv - Dv / 4.f;
        const float CLh = fabsf(Fh) + fabsf(Dh);
        const float CLv = fabsf(Fv) + fabsf(Dv);
        const float sg = signf(CLh - CLv);
        gval = (1.f + sg) * rawv / 2.f + (1.f - sg) * rawh / 2.f;
    }
    return gval;
}

// The 3x3 neighbourhood of a pixel in the green plane and in the CFA (replicate padded: coordinates clamped before the lookup),
// loaded UNCONDITIONALLY and together: algo2 below picks among them by the pixel's CFA site, and a load inside one of its four
// site branches was a memory round trip of its own (the branches diverge inside a wave: every path ran, each with its loads and
// its s_waitcnt vmcnt(0) -- 25 dependent round trips per pixel in netin_kernel, profiles/r06g_netin_load_batches.txt).
struct Ring {
    float g[3][3], r[3][3];
    int site[3][3];      // CFA site of the (clamped) neighbour
};
__device__ __forceinline__ void load_ring(const Cfa& c, const float* __restrict__ gp, int H, int W, int y, int x, Ring& q) {
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int yy = clampi(y + dy - 1, 0, H - 1), xx = clampi(x + dx - 1, 0, W - 1);
            const int st = ((yy & 1) << 1) | (xx & 1);
            q.site[dy][dx] = st;
            q.g[dy][dx] = gp[(size_t)yy * W + xx];
            q.r[dy][dx] = c.raw[((size_t)st * c.h + (yy >> 1)) * c.w + (xx >> 1)];
        }
}

// algo2 (util/Hamilton_Adam_demo.py:145-172) for red (k = 0) and blue (k = 1) at the ring's centre pixel (y, x)
__device__ __forceinline__ void ha_red_blue(const Ring& q, int y, int x, float (&rb)[2]) {
    auto G = [&](int dy, int dx) { return q.g[dy + 1][dx + 1]; };
    // sparse colour plane `own` at a neighbour (Cfa::plane): the sample where the neighbour's site is `own`, zero elsewhere
    auto P = [&](int dy, int dx, int own) { return q.site[dy + 1][dx + 1] == own ? q.r[dy + 1][dx + 1] : 0.f; };
    const float g0 = q.g[1][1];
    const int site = ((y & 1) << 1) | (x & 1);   // 0 Gb(e,e) 1 B 2 R 3 Gr(o,o)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int own = k == 0 ? 2 : 1;          // the channel's own CFA site (R / B)
        const int hsite = k == 0 ? 3 : 0;        // sites with horizontal neighbours of the channel
        const int vsite = k == 0 ? 0 : 3;        // sites with vertical neighbours
        float v;
        if (site == own) {
            v = q.r[1][1];
        } else if (site == hsite) {
            const float Kh = 0.5f * P(0, -1, own) + 0.5f * P(0, 1, own);
            const float gD = (0.25f * G(0, -1) + (-0.5f) * g0) + 0.25f * G(0, 1);
            v = Kh - gD;
        } else if (site == vsite) {
            const float Kv = 0.5f * P(-1, 0, own) + 0.5f * P(1, 0, own);
            const float gD = (0.25f * G(-1, 0) + (-0.5f) * g0) + 0.25f * G(1, 0);
            v = Kv - gD;
        } else {  // mask_ochan site (B sites for red, R sites for blue): diagonal, hard selection
            const float a = P(-1, -1, own), d = P(1, 1, own);
            const float bq = P(-1, 1, own), cq = P(1, -1, own);
            const float Kp = 0.5f * a + 0.5f * d;
            const float Kn = 0.5f * bq + 0.5f * cq;
            const float Fp = (-1.f) * a + d;
            const float Fn = (-1.f) * bq + cq;
            const float gDp = (G(-1, -1) + (-2.f) * g0) + G(1, 1);
            const float gDn = (G(-1, 1) + (-2.f) * g0) + G(1, -1);
            const float Cp = Kp - gDp / 4.f;
            const float Cn = Kn - gDn / 4.f;
            const float CLp = fabsf(Fp) + fabsf(gDp);
            const float CLn = fabsf(Fn) + fabsf(gDn);
            const float sl = signf(CLp - CLn);
            v = (1.f + sl) * Cn / 2.f + (1.f - sl) * Cp / 2.f;
        }
        rb[k] = v;
    }
}

__global__ void ha_rb_kernel(const float* __restrict__ raw, const float* __restrict__ green,
                             float* __restrict__ out, int n, int h, int w, int64_t bstride, int pstride,
                             int cstride, int64_t rbs) {
    const int H = 2 * h, W = 2 * w;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)n * H * W) return;
    const int x = idx % W;
    const int y = (idx / W) % H;
    const int b = idx / ((size_t)W * H);
    Cfa c{raw + (size_t)b * rbs, h, w, H, W};
    const float* gp = green + (size_t)b * H * W;
    Ring q;
    load_ring(c, gp, H, W, y, x, q);
    const float g0 = q.g[1][1];
    float rb[2];
    ha_red_blue(q, y, x, rb);
    float* o = out + (size_t)b * bstride + ((size_t)y * W + x) * pstride;
    o[0] = rb[0];
    o[cstride] = g0;
    o[2 * (size_t)cstride] = rb[1];
}

// ---------------------------------------------------------------------- warp --
// Cubic-convolution weights, A = -0.75 (ATen UpSample.h get_cubic_upsample_coefficients).
__device__ __forceinline__ void cubic_w(float t, float w[4]) {
    const float A = -0.75f;
    float x = t + 1.f;
    w[0] = ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A;
    x = t;
    w[1] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
    x = 1.f - t;
    w[2] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
    x = 2.f - t;
    w[3] = ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A;
}

// full-resolution flow at (y,x) from the raw-resolution one:
// F.interpolate(x2, bilinear, align_corners=True) * 2  (util/flow_utils.py:159-174,
// models/recurrent_model.py:128-129)
struct FlowQ {      // the four raw-resolution flow vectors around a pixel and their bilinear weights
    float f[2][4];
    float ly0, ly1, lx0, lx1;
};
__device__ __forceinline__ void flow_fetch(const float* __restrict__ fr, int h, int w, int H, int W, int y, int x, FlowQ& q) {
    const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
    const float sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const float py = sy * (float)y, px = sx * (float)x;
    const int y0 = (int)py, x0 = (int)px;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    q.ly1 = py - (float)y0;
    q.lx1 = px - (float)x0;
    q.ly0 = 1.f - q.ly1;
    q.lx0 = 1.f - q.lx1;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const float* f = fr + (size_t)c * h * w;
        q.f[c][0] = f[y0 * w + x0];
        q.f[c][1] = f[y0 * w + x1];
        q.f[c][2] = f[y1 * w + x0];
        q.f[c][3] = f[y1 * w + x1];
    }
}
__device__ __forceinline__ void flow_combine(const FlowQ& q, float& fx, float& fy) {
    fx = (q.ly0 * (q.lx0 * q.f[0][0] + q.lx1 * q.f[0][1]) + q.ly1 * (q.lx0 * q.f[0][2] + q.lx1 * q.f[0][3])) * 2.f;
    fy = (q.ly0 * (q.lx0 * q.f[1][0] + q.lx1 * q.f[1][1]) + q.ly1 * (q.lx0 * q.f[1][2] + q.lx1 * q.f[1][3])) * 2.f;
}
__device__ __forceinline__ void flow_at(const float* __restrict__ fr, int h, int w, int H, int W, int y,
                                        int x, float& fx, float& fy) {
    FlowQ q;
    flow_fetch(fr, h, w, H, W, y, x, q);
    flow_combine(q, fx, fy);
}

// util/flow_utils.py:90-99 + ATen grid_sampler (bicubic, border, align_corners=True)
struct Taps {
    int xi[4], yi[4];
    float wx[4], wy[4];
};
__device__ __forceinline__ void make_taps(float fx, float fy, int x, int y, int H, int W, Taps& t) {
    const float vx = (float)x + fx, vy = (float)y + fy;
    const float gx = 2.0f * vx / (float)(W - 1) - 1.0f;
    const float gy = 2.0f * vy / (float)(H - 1) - 1.0f;
    const float ix = ((gx + 1.f) / 2.f) * (float)(W - 1);
    const float iy = ((gy + 1.f) / 2.f) * (float)(H - 1);
    const float x0 = floorf(ix), y0 = floorf(iy);
    cubic_w(ix - x0, t.wx);
    cubic_w(iy - y0, t.wy);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        // clip_coordinates on the float tap coordinate, then the integer cast
        float cx = x0 - 1.f + (float)i, cy = y0 - 1.f + (float)i;
        cx = fminf((float)(W - 1), fmaxf(cx, 0.f));
        cy = fminf((float)(H - 1), fmaxf(cy, 0.f));
        t.xi[i] = (int)cx;
        t.yi[i] = (int)cy;
    }
}

__global__ void warp3_kernel(const float* __restrict__ src4, const float* __restrict__ flow_raw,
                             float* __restrict__ dst, int dpstride, int B, int H, int W) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * H * W) return;
    const int x = idx % W;
    const int y = (idx / W) % H;
    const int b = idx / ((size_t)W * H);
    const int h = H / 2, w = W / 2;
    const f32x4* s = reinterpret_cast<const f32x4*>(src4) + (size_t)b * H * W;
    if (!flow_raw) {        // --no_warp: warp_frame returns its input (models/recurrent_model.py:156-158)
        const f32x4 v = s[(size_t)y * W + x];
        float* o = dst + idx * dpstride;
        o[0] = v[0];
        o[1] = v[1];
        o[2] = v[2];
        return;
    }
    float fx, fy;
    flow_at(flow_raw + (size_t)b * 2 * h * w, h, w, H, W, y, x, fx, fy);
    Taps t;
    make_taps(fx, fy, x, y, H, W, t);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f32x4 row = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) row = row + s[(size_t)t.yi[j] * W + t.xi[i]] * t.wx[i];
        acc = acc + row * t.wy[j];
    }
    float* o = dst + idx * dpstride;
    o[0] = acc[0];
    o[1] = acc[1];
    o[2] = acc[2];
}

// The network input of a frame in ONE pass (models/recurrent_model.py:299-324): per pixel the bicubic warp of the
// previous output (3 channels), the red / blue planes of the current frame's Hamilton-Adams demosaic (its green
// plane comes from ha_green_kernel) and, with a future frame, the warp of the demosaicked next frame, written as the
// pixel's whole NHWC16 vector -- three 16-B stores of full sectors.  Written by three kernels (red/blue, two warps)
// every pixel's 64 bytes were touched three times with 12-B partial stores.  Same arithmetic as ha_rb_kernel and
// warp3_kernel (the demosaic stays bit-exact).
// (the 16 taps of a pixel are loaded TOGETHER, then summed in warp3_kernel's order)
__device__ __forceinline__ void warp3_gather(const f32x4* __restrict__ s, const Taps& t, int W, f32x4 (&v)[16]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) v[4 * j + i] = s[(size_t)t.yi[j] * W + t.xi[i]];
}
__device__ __forceinline__ f32x4 warp3_sum(const Taps& t, const f32x4 (&v)[16]) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f32x4 row = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) row = row + v[4 * j + i] * t.wx[i];
        acc = acc + row * t.wy[j];
    }
    return acc;
}

// The 16 channels of one network-input pixel: warp3 of prev4 | demosaic of raw_cur | warp3 of next4 (zeros without one) | 0..
struct NetinArgs {
    const float *raw_cur, *green, *prev4, *flow_prev, *next4, *flow_next;
    int B, h, w;
    int64_t rbs, fbs;
};
// Three batches of loads, each in flight together: (1) the flow vectors around the pixel (both directions) and the 3x3 rings of the
// green plane and the CFA, (2) the 16 bicubic taps of the previous output, (3) those of the next frame -- with the demosaic's
// arithmetic between them.  Written as one expression after another (round 1-5) the compiler kept the kernel at 8 waves per SIMD by
// issuing every load next to its use: ~60 dependent memory round trips per pixel, waves 75 % of their life in s_waitcnt
// (profiles/r06f_c4_prestage_counters.json).  Same operations on the same values: same bits.
__device__ __forceinline__ void netin_pixel(const NetinArgs& a, size_t idx, f32x4 (&o)[3]) {
    const int H = 2 * a.h, W = 2 * a.w;
    const unsigned i32 = (unsigned)idx;      // (B H W < 2^32: launch_netin checks; 64-bit divisions are ~100 instructions each)
    const int x = (int)(i32 % (unsigned)W);
    const unsigned yb = i32 / (unsigned)W;
    const int y = (int)(yb % (unsigned)H);
    const int b = (int)(yb / (unsigned)H);
    Cfa c{a.raw_cur + (size_t)b * a.rbs, a.h, a.w, H, W};
    const float* gp = a.green + (size_t)b * H * W;
    const f32x4* sp = reinterpret_cast<const f32x4*>(a.prev4) + (size_t)b * H * W;
    const f32x4* sn = reinterpret_cast<const f32x4*>(a.next4) + (size_t)b * H * W;
    const bool warp_p = a.flow_prev != nullptr, has_n = a.next4 != nullptr, warp_n = has_n && a.flow_next != nullptr;
    // ---- batch 1
    FlowQ fq_p, fq_n;
    if (warp_p) flow_fetch(a.flow_prev + (size_t)b * a.fbs, a.h, a.w, H, W, y, x, fq_p);
    if (warp_n) flow_fetch(a.flow_next + (size_t)b * a.fbs, a.h, a.w, H, W, y, x, fq_n);
    Ring q;
    load_ring(c, gp, H, W, y, x, q);
    // ---- batch 2: the taps of the previous output, gathered together (--no_warp: the pixel itself, models/recurrent_model.py:156-158)
    Taps tp;
    f32x4 vt[16];
    f32x4 p, n = {0.f, 0.f, 0.f, 0.f};
    if (warp_p) {
        float fx, fy;
        flow_combine(fq_p, fx, fy);
        make_taps(fx, fy, x, y, H, W, tp);
        warp3_gather(sp, tp, W, vt);
    } else {
        p = sp[(size_t)y * W + x];
    }
    // ---- the demosaic's red and blue under the gather's flight
    float rb[2];
    ha_red_blue(q, y, x, rb);
    const float g0 = q.g[1][1];
    if (warp_p) p = warp3_sum(tp, vt);
    // ---- batch 3: the next frame's taps (one gather's 64 registers at a time: both in flight spill)
    if (warp_n) {
        float fx, fy;
        flow_combine(fq_n, fx, fy);
        Taps tn;
        make_taps(fx, fy, x, y, H, W, tn);
        warp3_gather(sn, tn, W, vt);
        n = warp3_sum(tn, vt);
    } else if (has_n) {
        n = sn[(size_t)y * W + x];
    }
    o[0] = f32x4{p[0], p[1], p[2], rb[0]};
    o[1] = f32x4{g0, rb[1], n[0], n[1]};
    o[2] = f32x4{n[2], 0.f, 0.f, 0.f};
}
// Workgroups go to the eight XCDs in turn; a pixel's stencils and bicubic taps reach two rows up and down, and a row is
// several workgroups long: numbered as they come, vertically adjacent workgroups sit on different XCDs and every L2
// fetches the same rows again.  Each XCD takes a contiguous eighth of the pixels instead.
__device__ __forceinline__ unsigned xcd_contiguous_block() {
    const unsigned per = gridDim.x >> 3;
    return blockIdx.x < 8u * per ? (blockIdx.x & 7u) * per + (blockIdx.x >> 3) : blockIdx.x;
}

// (at most five waves per SIMD asked of the compiler: with the default target of eight it serialises the loads again to save registers)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4))) void netin_kernel(NetinArgs a, float* __restrict__ netin) {
    const size_t idx = (size_t)xcd_contiguous_block() * blockDim.x + threadIdx.x;
    if (idx >= (size_t)a.B * 4 * a.h * a.w) return;
    f32x4 v[3];
    netin_pixel(a, idx, v);
    f32x4* o = reinterpret_cast<f32x4*>(netin) + idx * 4;
    o[0] = v[0];
    o[1] = v[1];
    o[2] = v[2];
}

// netin_kernel for ConvNeXtUnet: the network input has exactly one reader there, the 1x1 projection 9 | 6 -> 48 of the first
// ConvBlock (networks/new_unet.py:85-88), so the 16-channel map is never written: the 256 pixels of a workgroup change lanes
// through LDS (pixel pitch 24 floats: the 16 lanes of a ds_read_b128 group on 16 distinct bank quads) and each wave projects
// four 16-pixel groups on the f32 matrix pipe (16x16x4, exact f32 products).  pw as proj1x1_kernel<16, 0>'s:
// [m 3][lr 16][g 4][i 4] = W[16m+lr][4g+i] (zero for channels the input does not have).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4))) void netin_proj_kernel(NetinArgs a, const float* __restrict__ pw, const float* __restrict__ bias,
                                                         float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float s_t[256][24];
    const size_t total = (size_t)a.B * 4 * a.h * a.w;
    const size_t base = (size_t)xcd_contiguous_block() * 256;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, lr = lane & 15, g = lane >> 4;
    f32x4 wa[3], bv[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        wa[m] = *reinterpret_cast<const f32x4*>(pw + (((size_t)m * 16 + lr) * 4 + g) * 4);
        bv[m] = *reinterpret_cast<const f32x4*>(bias + 16 * m + 4 * g);
    }
    f32x4 v[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (base + threadIdx.x < total) netin_pixel(a, base + threadIdx.x, v);
#pragma unroll
    for (int k = 0; k < 3; ++k) *reinterpret_cast<f32x4*>(&s_t[threadIdx.x][4 * k]) = v[k];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int px = 64 * wv + 16 * q + lr;
        f32x4 xb = *reinterpret_cast<const f32x4*>(&s_t[px][4 * (g < 3 ? g : 0)]);
        if (g == 3) xb = f32x4{0.f, 0.f, 0.f, 0.f};           // channels 12..15 do not exist
        f32x4 acc[3] = {bv[0], bv[1], bv[2]};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int m = 0; m < 3; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[m][i], xb[i], acc[m], 0, 0, 0);
        if (base + px < total) {
#pragma unroll
            for (int m = 0; m < 3; ++m) *reinterpret_cast<f32x4*>(out + (base + px) * kF + 16 * m + 4 * g) = acc[m];
        }
    }
}

// An upper bound of max |network input| per sequence, for the block floating point of the conv behind it (rvdd_internal.h),
// from what the input is made of instead of from its 16-channel map (a pass over that map, or a maximum inside netin_kernel's
// 28 800 blocks, costs 100 us and more per frame-step at 720p; this kernel reads the packed raw frames, 1/16 of it):
//   |bicubic gather of x| <= 1.375^2 max |x| (A = -0.75: the taps' absolute sum peaks at 1.375 per axis),
//   |Hamilton-Adams(raw)| <= 3 max |raw| (green: a two-sample mean plus a quarter of a Laplacian, <= 2 M; red / blue: a
//   two-sample mean plus a quarter of a Laplacian of those greens, <= 3 M),
// so max |netin| <= 5.7 max(max |raw|, max |previous output|) < 16 max(...) -- and >= max |raw_cur|, whose samples the demosaic
// keeps: the bound is never more than 16 x too large, four of the seventeen binades the split's full precision spans.
// Up to three raw frames (the current one, the next one, and on the first step of a video the previous one, whose demosaic is
// the "previous output"; each nullable); `prev_words` (nullable) = words whose maximum bounds the previous output (the slot
// PostConvs wrote in the last step: features and output frame together); grid (blocks, B).
__global__ __launch_bounds__(256) void netin_bound_kernel(const float* __restrict__ raw_a, const float* __restrict__ raw_b,
                                                          const float* __restrict__ raw_c, int64_t n, int64_t rbs,
                                                          const unsigned* __restrict__ prev_words, unsigned* __restrict__ words,
                                                          unsigned* __restrict__ zero_a, size_t zero_na, unsigned* __restrict__ zero_b,
                                                          size_t zero_nb) {
    __shared__ unsigned red[4];
    const int b = blockIdx.y;
    // housekeeping for the step AFTER this one: its set of amax words and its features slot, which nobody touches meanwhile
    {
        const size_t me = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x, all = (size_t)gridDim.x * gridDim.y * 256;
        if (zero_a) for (size_t i = me; i < zero_na; i += all) zero_a[i] = 0u;
        if (zero_b) for (size_t i = me; i < zero_nb; i += all) zero_b[i] = 0u;
    }
    float m = 0.f;
    for (int pass = 0; pass < 3; ++pass) {
        const float* src = pass == 0 ? raw_a : pass == 1 ? raw_b : raw_c;
        if (!src) continue;
        const f32x4* p = reinterpret_cast<const f32x4*>(src + (size_t)b * rbs);
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (n >> 2); i += (int64_t)gridDim.x * 256) {
            const f32x4 v = p[i];
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        }
    }
    unsigned bits = __float_as_uint(m);
    if (blockIdx.x == 0 && prev_words && threadIdx.x < kAmaxLines)
        bits = max(bits, prev_words[(size_t)b * kAmaxSeqWords + threadIdx.x * kAmaxLineWords]);
    // x 16: four up on the exponent (a maximum that is zero, denormal, or within 2^4 of overflow stays as it is)
    const unsigned e = (bits >> 23) & 0xffu;
    if (e >= 1 && e < 250) bits += 4u << 23;
    amax_commit_block(words, b, blockIdx.x, __uint_as_float(bits), red);
}

// The three pre-stage launches of a frame-step -- netin_bound_kernel, ha_green_kernel, netin_kernel -- in ONE for small frames without
// a future frame (round 6: BASELINE's C1 as stated, one 256x256 sequence, is 28 launches of 5-15 us back to back; these three were
// 14.7 us of ~270).  A block owns a 16x16 tile: the green plane of the 18x18 pixels around it goes to LDS (ha_green_at at the
// clamped coordinates, i.e. what the green kernel wrote and netin_kernel read back through a clamped index), then netin_pixel's
// sequence per pixel with the ring's greens from LDS, then netin_bound_kernel's tail: every raw sample of the current frame is some
// thread's own CFA sample.  Same operations on the same values: same bits as the three kernels.  Only where the launch is small: one
// atomic per block on the amax words costs 100 us at 28 800 blocks (720p, round 4) -- launch_netin_small refuses above 1024 blocks.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4))) void netin_small_kernel(
    NetinArgs a, float* __restrict__ netin, const float* __restrict__ raw_prev, const unsigned* __restrict__ prev_words,
    unsigned* __restrict__ words, unsigned* __restrict__ zero_a, size_t zero_na, unsigned* __restrict__ zero_b, size_t zero_nb, int tiles_x) {
    __shared__ float gs[18][20];
    __shared__ unsigned red[4];
    const int H = 2 * a.h, W = 2 * a.w;
    const int b = blockIdx.y, t = threadIdx.x;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int y0 = ty * 16, x0 = tx * 16;
    {   // housekeeping for the step AFTER this one (netin_bound_kernel's)
        const size_t me = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + t, all = (size_t)gridDim.x * gridDim.y * 256;
        if (zero_a) for (size_t i = me; i < zero_na; i += all) zero_a[i] = 0u;
        if (zero_b) for (size_t i = me; i < zero_nb; i += all) zero_b[i] = 0u;
    }
    Cfa c{a.raw_cur + (size_t)b * a.rbs, a.h, a.w, H, W};
    for (int i = t; i < 18 * 18; i += 256) {
        const int r = i / 18, cc = i - r * 18;
        gs[r][cc] = ha_green_at(c, clampi(y0 - 1 + r, 0, H - 1), clampi(x0 - 1 + cc, 0, W - 1));
    }
    const int ly = t >> 4, lx = t & 15, y = y0 + ly, x = x0 + lx;
    const bool inside = y < H && x < W;
    // flow vectors and the CFA ring while the greens are being formed
    FlowQ fq;
    Ring q;
    const bool warp_p = a.flow_prev != nullptr;
    const f32x4* sp = reinterpret_cast<const f32x4*>(a.prev4) + (size_t)b * H * W;
    if (inside) {
        if (warp_p) flow_fetch(a.flow_prev + (size_t)b * a.fbs, a.h, a.w, H, W, y, x, fq);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int yy = clampi(y + dy - 1, 0, H - 1), xx = clampi(x + dx - 1, 0, W - 1);
                const int st = ((yy & 1) << 1) | (xx & 1);
                q.site[dy][dx] = st;
                q.r[dy][dx] = c.raw[((size_t)st * c.h + (yy >> 1)) * c.w + (xx >> 1)];
            }
    }
    __syncthreads();
    float m = 0.f;
    if (inside) {
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) q.g[dy][dx] = gs[ly + dy][lx + dx];
        Taps tp;
        f32x4 vt[16];
        f32x4 p;
        if (warp_p) {
            float fx, fy;
            flow_combine(fq, fx, fy);
            make_taps(fx, fy, x, y, H, W, tp);
            warp3_gather(sp, tp, W, vt);
        } else {
            p = sp[(size_t)y * W + x];
        }
        float rb[2];
        ha_red_blue(q, y, x, rb);
        const float g0 = q.g[1][1];
        if (warp_p) p = warp3_sum(tp, vt);
        f32x4* o = reinterpret_cast<f32x4*>(netin) + ((size_t)b * H * W + (size_t)y * W + x) * 4;
        o[0] = f32x4{p[0], p[1], p[2], rb[0]};
        o[1] = f32x4{g0, rb[1], 0.f, 0.f};
        o[2] = f32x4{0.f, 0.f, 0.f, 0.f};
        m = fabsf(q.r[1][1]);
        // (the first step of a video: the "previous output" is the demosaic of the previous raw frame, whose samples bound it)
        if (raw_prev) m = fmaxf(m, fabsf(raw_prev[(size_t)b * a.rbs + ((size_t)q.site[1][1] * c.h + (y >> 1)) * c.w + (x >> 1)]));
    }
    if (words) {      // netin_bound_kernel's tail
        unsigned bits = __float_as_uint(m);
        if (blockIdx.x == 0 && prev_words && t < kAmaxLines) bits = max(bits, prev_words[(size_t)b * kAmaxSeqWords + t * kAmaxLineWords]);
        const unsigned e = (bits >> 23) & 0xffu;
        if (e >= 1 && e < 250) bits += 4u << 23;
        amax_commit_block(words, b, blockIdx.x, __uint_as_float(bits), red);
    }
}

// grid = (ceil(W/32), H, B), 192 threads = 16 PAIRS of horizontally adjacent pixels x 12 float4 chunks.
//  * The 16 taps of a pixel (flow upsample, coordinate round trip, cubic weights: ~150 VALU instructions) are
//    computed ONCE per pixel by the first 32 threads and shared through LDS; computed by each of the 12 lanes of
//    a pixel they cost as much SIMD time as the gather itself.
//  * The gather is bound by the texture-address path (16 x 16-B loads per lane and pixel, not by HBM: neighbouring
//    pixels re-read the same lines from L1).  Where the flow is smooth the 4x4 footprints of the two pixels of a
//    pair are the same rows and columns shifted by one: 20 loads serve both instead of 32.  Pairs whose footprints
//    do not line up (flow discontinuities, the clamped image border) take the plain 2 x 16 path; the arithmetic of
//    a pixel is the same in both.
__global__ __launch_bounds__(192) void warp48_kernel(const float* __restrict__ src,
                                                     const float* __restrict__ flow_raw,
                                                     float* __restrict__ dst, int B, int H, int W, int64_t fbs) {
    __shared__ int s_i[32][8];      // xi[4], yi[4] * W
    __shared__ float s_w[32][8];    // wx[4], wy[4]
    const int y = blockIdx.y, b = blockIdx.z, x0 = blockIdx.x * 32;
    if (threadIdx.x < 32) {
        const int x = x0 + threadIdx.x;
        if (x < W) {
            const int h = H / 2, w = W / 2;
            float fx, fy;
            flow_at(flow_raw + (size_t)b * fbs, h, w, H, W, y, x, fx, fy);
            Taps t;
            make_taps(fx, fy, x, y, H, W, t);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s_i[threadIdx.x][k] = t.xi[k];
                s_i[threadIdx.x][4 + k] = t.yi[k] * W;
                s_w[threadIdx.x][k] = t.wx[k];
                s_w[threadIdx.x][4 + k] = t.wy[k];
            }
        }
    }
    __syncthreads();
    const int p = threadIdx.x / 12;
    const int c4 = threadIdx.x - p * 12;
    const int pa = 2 * p, pb = 2 * p + 1, xa = x0 + pa;
    if (xa >= W) return;
    const bool has_b = xa + 1 < W;
    const f32x4* s = reinterpret_cast<const f32x4*>(src) + (size_t)b * H * W * 12 + c4;
    f32x4* o = reinterpret_cast<f32x4*>(dst) + (((size_t)b * H + y) * W + xa) * 12 + c4;
    int xia[4], yia[4], xib[4], yib[4];
    float wxa[4], wya[4], wxb[4], wyb[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        xia[k] = s_i[pa][k]; yia[k] = s_i[pa][4 + k]; wxa[k] = s_w[pa][k]; wya[k] = s_w[pa][4 + k];
        xib[k] = s_i[pb][k]; yib[k] = s_i[pb][4 + k]; wxb[k] = s_w[pb][k]; wyb[k] = s_w[pb][4 + k];
    }
    const bool lined_up = has_b && yib[0] == yia[0] && yib[1] == yia[1] && yib[2] == yia[2] && yib[3] == yia[3] &&
                          xib[0] == xia[1] && xib[1] == xia[2] && xib[2] == xia[3];
    f32x4 acca = {0.f, 0.f, 0.f, 0.f}, accb = {0.f, 0.f, 0.f, 0.f};
    if (lined_up) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 v0 = s[(yia[j] + xia[0]) * 12], v1 = s[(yia[j] + xia[1]) * 12], v2 = s[(yia[j] + xia[2]) * 12];
            const f32x4 v3 = s[(yia[j] + xia[3]) * 12], v4 = s[(yia[j] + xib[3]) * 12];
            f32x4 ra = {0.f, 0.f, 0.f, 0.f}, rb = {0.f, 0.f, 0.f, 0.f};
            ra = ra + v0 * wxa[0]; ra = ra + v1 * wxa[1]; ra = ra + v2 * wxa[2]; ra = ra + v3 * wxa[3];
            rb = rb + v1 * wxb[0]; rb = rb + v2 * wxb[1]; rb = rb + v3 * wxb[2]; rb = rb + v4 * wxb[3];
            acca = acca + ra * wya[j];
            accb = accb + rb * wyb[j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 ra = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i) ra = ra + s[(yia[j] + xia[i]) * 12] * wxa[i];
            acca = acca + ra * wya[j];
        }
        if (has_b) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 rb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 4; ++i) rb = rb + s[(yib[j] + xib[i]) * 12] * wxb[i];
                accb = accb + rb * wyb[j];
            }
        }
    }
    o[0] = acca;
    if (has_b) o[12] = accb;
}

// warp48_kernel with a 48 -> 48 projection of the warped pixel on its way out: dst = W warp(src) + bias.  ConvNeXtUnet's first
// encoder block projects cat[y, warped features] 96 -> 48 (networks/new_unet.py:381-382, 85-88); the projection is linear
// in the two maps and the warped features have no other reader, so their half of it rides here and the other half in the
// epilogue of the block that forms y (convnext.hip PROJ) -- the projection kernel and its 4 S of traffic are gone.
// The 32 warped pixels of the workgroup change lanes through LDS (pixel pitch 52 floats: the 16 lanes of a ds_read_b128 group
// on 16 distinct bank quads); waves 0 and 1 then project one 16-pixel group each on the F16 matrix pipe exactly as convnext.hip's
// PROJ epilogue does: per-pixel power of two (the features have no a-priori bound), split into f16 halves, fc1's five-MFMA
// pattern on the three 16-row blocks of 2^s W (frag: runtime_next.inc's half[1] fragments, 9 KiB, copied to LDS once per
// workgroup), one fma back.  (The first form multiplied f32 on v_mfma_f32_16x16x4_f32: 72 per workgroup at 32 cycles each ON the
// vector lanes the gather's FMAs need -- 1 247 us against warp48_kernel's 899.)
typedef _Float16 wp_h8 __attribute__((ext_vector_type(8)));
typedef __fp16 wp_fp16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 wp_h2 __attribute__((ext_vector_type(2)));
typedef unsigned int wp_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void wp_split4(f32x4 x, unsigned (&hi)[2], unsigned (&lo)[2]) {      // convnext.hip split4h
    const wp_fp16x2 h01 = __builtin_amdgcn_cvt_pkrtz(x[0], x[1]);
    const wp_fp16x2 h23 = __builtin_amdgcn_cvt_pkrtz(x[2], x[3]);
    const unsigned u01 = __builtin_bit_cast(unsigned, h01), u23 = __builtin_bit_cast(unsigned, h23);
    float r0, r1, r2, r3;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(u01), "v"(x[0]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(u01), "v"(x[1]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r2) : "v"(u23), "v"(x[2]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r3) : "v"(u23), "v"(x[3]));
    const wp_h2 l01 = {(_Float16)r0, (_Float16)r1};
    const wp_h2 l23 = {(_Float16)r2, (_Float16)r3};
    hi[0] = u01; hi[1] = u23;
    lo[0] = __builtin_bit_cast(unsigned, l01); lo[1] = __builtin_bit_cast(unsigned, l23);
}
__global__ __launch_bounds__(192) void warp48_proj_kernel(const float* __restrict__ src, const float* __restrict__ flow_raw,
                                                          float* __restrict__ dst, int B, int H, int W, int64_t fbs,
                                                          const float* __restrict__ frag, int inv_e, const float* __restrict__ bias) {
    __shared__ int s_i[32][8];      // xi[4], yi[4] * W
    __shared__ float s_w[32][8];    // wx[4], wy[4]
    __shared__ __attribute__((aligned(16))) float s_t[32][52];
    __shared__ __attribute__((aligned(16))) float s_f[2304];       // the projection's fragments: [m 3][Fa Fb Fc][64 lanes][16 B]
    const int y = blockIdx.y, b = blockIdx.z, x0 = blockIdx.x * 32;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, lr = lane & 15, g = lane >> 4;
#pragma unroll
    for (int k = 0; k < 3; ++k)
        reinterpret_cast<f32x4*>(s_f)[threadIdx.x + 192 * k] = reinterpret_cast<const f32x4*>(frag)[threadIdx.x + 192 * k];
    if (threadIdx.x < 32) {
        const int x = x0 + threadIdx.x;
        if (x < W) {
            const int h = H / 2, w = W / 2;
            float fx, fy;
            flow_at(flow_raw + (size_t)b * fbs, h, w, H, W, y, x, fx, fy);
            Taps t;
            make_taps(fx, fy, x, y, H, W, t);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s_i[threadIdx.x][k] = t.xi[k];
                s_i[threadIdx.x][4 + k] = t.yi[k] * W;
                s_w[threadIdx.x][k] = t.wx[k];
                s_w[threadIdx.x][4 + k] = t.wy[k];
            }
        }
    }
    __syncthreads();
    const int p = threadIdx.x / 12;
    const int c4 = threadIdx.x - p * 12;
    const int pa = 2 * p, pb = 2 * p + 1, xa = x0 + pa;
    f32x4 acca = {0.f, 0.f, 0.f, 0.f}, accb = {0.f, 0.f, 0.f, 0.f};
    if (xa < W) {
        const bool has_b = xa + 1 < W;
        const f32x4* s = reinterpret_cast<const f32x4*>(src) + (size_t)b * H * W * 12 + c4;
        int xia[4], yia[4], xib[4], yib[4];
        float wxa[4], wya[4], wxb[4], wyb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            xia[k] = s_i[pa][k]; yia[k] = s_i[pa][4 + k]; wxa[k] = s_w[pa][k]; wya[k] = s_w[pa][4 + k];
            xib[k] = s_i[pb][k]; yib[k] = s_i[pb][4 + k]; wxb[k] = s_w[pb][k]; wyb[k] = s_w[pb][4 + k];
        }
        const bool lined_up = has_b && yib[0] == yia[0] && yib[1] == yia[1] && yib[2] == yia[2] && yib[3] == yia[3] &&
                              xib[0] == xia[1] && xib[1] == xia[2] && xib[2] == xia[3];
        if (lined_up) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 v0 = s[(yia[j] + xia[0]) * 12], v1 = s[(yia[j] + xia[1]) * 12], v2 = s[(yia[j] + xia[2]) * 12];
                const f32x4 v3 = s[(yia[j] + xia[3]) * 12], v4 = s[(yia[j] + xib[3]) * 12];
                f32x4 ra = {0.f, 0.f, 0.f, 0.f}, rb = {0.f, 0.f, 0.f, 0.f};
                ra = ra + v0 * wxa[0]; ra = ra + v1 * wxa[1]; ra = ra + v2 * wxa[2]; ra = ra + v3 * wxa[3];
                rb = rb + v1 * wxb[0]; rb = rb + v2 * wxb[1]; rb = rb + v3 * wxb[2]; rb = rb + v4 * wxb[3];
                acca = acca + ra * wya[j];
                accb = accb + rb * wyb[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 ra = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 4; ++i) ra = ra + s[(yia[j] + xia[i]) * 12] * wxa[i];
                acca = acca + ra * wya[j];
            }
            if (has_b) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 rb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int i = 0; i < 4; ++i) rb = rb + s[(yib[j] + xib[i]) * 12] * wxb[i];
                    accb = accb + rb * wyb[j];
                }
            }
        }
    }
    *reinterpret_cast<f32x4*>(&s_t[pa][4 * c4]) = acca;
    *reinterpret_cast<f32x4*>(&s_t[pb][4 * c4]) = accb;
    __syncthreads();
    if (wv >= 2) return;
    // ---- waves 0, 1: the 16 pixels of group wv, all 48 output channels
    f32x4 v[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) v[j] = *reinterpret_cast<const f32x4*>(&s_t[16 * wv + lr][16 * j + 4 * g]);
    float mx = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v[j][0]), fabsf(v[j][1]))), fmaxf(fabsf(v[j][2]), fabsf(v[j][3])));
    unsigned mb = __float_as_uint(mx);
    auto s32 = __builtin_amdgcn_permlane32_swap(mb, mb, false, false);
    mb = max(s32[0], s32[1]);
    auto s16 = __builtin_amdgcn_permlane16_swap(mb, mb, false, false);
    mb = max(s16[0], s16[1]);
    const int e = min((int)(mb >> 23) & 0xff, 252);
    const float fwd = __uint_as_float((unsigned)(253 - e) << 23);
    const float back = __uint_as_float((unsigned)min(max(e + 1 + inv_e, 1), 254) << 23);
    unsigned xh[3][2], xl[3][2];
#pragma unroll
    for (int j = 0; j < 3; ++j) wp_split4(v[j] * fwd, xh[j], xl[j]);
    const wp_h8 P1 = __builtin_bit_cast(wp_h8, wp_u32x4{xh[0][0], xh[0][1], xh[1][0], xh[1][1]});
    const wp_h8 P2 = __builtin_bit_cast(wp_h8, wp_u32x4{xl[0][0], xl[0][1], xl[1][0], xl[1][1]});
    const wp_h8 P3 = __builtin_bit_cast(wp_h8, wp_u32x4{xh[2][0], xh[2][1], xh[2][0], xh[2][1]});
    const wp_h8 P4 = __builtin_bit_cast(wp_h8, wp_u32x4{xl[2][0], xl[2][1], 0u, 0u});
    const int x = x0 + 16 * wv + lr;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const wp_h8 fa = __builtin_bit_cast(wp_h8, *reinterpret_cast<const f32x4*>(s_f + (m * 3 + 0) * 256 + lane * 4));
        const wp_h8 fb = __builtin_bit_cast(wp_h8, *reinterpret_cast<const f32x4*>(s_f + (m * 3 + 1) * 256 + lane * 4));
        const wp_h8 fc = __builtin_bit_cast(wp_h8, *reinterpret_cast<const f32x4*>(s_f + (m * 3 + 2) * 256 + lane * 4));
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, P2, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb, P1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(fc, P4, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(fc, P3, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, P1, acc, 0, 0, 0);
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + 16 * m + 4 * g);
        if (x < W) *reinterpret_cast<f32x4*>(dst + (((size_t)b * H + y) * W + x) * kF + 16 * m + 4 * g) = acc * back + bv;
    }
}

__global__ void warp_nchw_kernel(const float* __restrict__ xin, const float* __restrict__ flow,
                                 float* __restrict__ yout, int n, int c, int H, int W) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)n * H * W) return;
    const int x = idx % W;
    const int y = (idx / W) % H;
    const int b = idx / ((size_t)W * H);
    const size_t hw = (size_t)H * W;
    const float fx = flow[((size_t)b * 2 + 0) * hw + (size_t)y * W + x];
    const float fy = flow[((size_t)b * 2 + 1) * hw + (size_t)y * W + x];
    Taps t;
    make_taps(fx, fy, x, y, H, W, t);
    for (int ch = 0; ch < c; ++ch) {
        const float* s = xin + ((size_t)b * c + ch) * hw;
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float row = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) row = row + s[(size_t)t.yi[j] * W + t.xi[i]] * t.wx[i];
            acc = acc + row * t.wy[j];
        }
        yout[((size_t)b * c + ch) * hw + (size_t)y * W + x] = acc;
    }
}

__global__ void upsample_flow_kernel(const float* __restrict__ t, float* __restrict__ out, int nc, int h,
                                     int w, float mul) {
    const int H = 2 * h, W = 2 * w;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)nc * H * W) return;
    const int x = idx % W;
    const int y = (idx / W) % H;
    const int p = idx / ((size_t)W * H);
    const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
    const float sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const float py = sy * (float)y, px = sx * (float)x;
    const int y0 = (int)py, x0 = (int)px;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    const float ly1 = py - (float)y0, lx1 = px - (float)x0;
    const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
    const float* f = t + (size_t)p * h * w;
    out[idx] = (ly0 * (lx0 * f[y0 * w + x0] + lx1 * f[y0 * w + x1]) +
                ly1 * (lx0 * f[y1 * w + x0] + lx1 * f[y1 * w + x1])) * mul;
}

// ------------------------------------------------------------ map reshaping --
// nn.Upsample(scale_factor=2, mode="bilinear"[, align_corners]) on NHWC48,
// 12 threads per output pixel.  ATen upsample_bilinear2d source index:
//   align_corners=False: src = max(0.5*(dst+0.5)-0.5, 0);  True: src = dst*(in-1)/(out-1)
// grid = (ceil(Wout/16), Hout, B), 192 threads = 16 output pixels of one row x 12 float4 chunks: the row terms
// (source rows, vertical weights) depend on blockIdx only and stay in scalar registers, no 64-bit divisions
__global__ __launch_bounds__(192) void upsample2x_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int h,
                                                         int w, int Hout, int Wout, int oy, int ox, int align) {
    const int p = threadIdx.x / 12, c4 = threadIdx.x - p * 12;
    const int X = blockIdx.x * 16 + p, Y = blockIdx.y, b = blockIdx.z;
    if (X >= Wout) return;
    const int y = Y - oy, x = X - ox;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)y < (unsigned)(2 * h) && (unsigned)x < (unsigned)(2 * w)) {
        float py, px;
        if (align) {
            const float sy = 2 * h > 1 ? (float)(h - 1) / (float)(2 * h - 1) : 0.f;
            const float sx = 2 * w > 1 ? (float)(w - 1) / (float)(2 * w - 1) : 0.f;
            py = sy * (float)y;
            px = sx * (float)x;
        } else {
            py = fmaxf(0.5f * ((float)y + 0.5f) - 0.5f, 0.f);
            px = fmaxf(0.5f * ((float)x + 0.5f) - 0.5f, 0.f);
        }
        const int y0 = (int)py, x0 = (int)px;
        const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
        const float ly1 = py - (float)y0, lx1 = px - (float)x0;
        const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
        const f32x4* s = reinterpret_cast<const f32x4*>(in) + (size_t)b * h * w * 12 + c4;
        const f32x4 v00 = s[(y0 * w + x0) * 12], v01 = s[(y0 * w + x1) * 12];
        const f32x4 v10 = s[(y1 * w + x0) * 12], v11 = s[(y1 * w + x1) * 12];
        // vertical pass, then horizontal, each "a * wa, then one fused multiply-add": the expression of the fused
        // upsample in wino3x3.hip (`interp`), so that the two paths produce the same bits
        auto fma4 = [](f32x4 a, float s, f32x4 c) { return __builtin_elementwise_fma(a, f32x4{s, s, s, s}, c); };
        const f32x4 c0 = fma4(v10, ly1, v00 * ly0), c1 = fma4(v11, ly1, v01 * ly0);
        v = fma4(c1, lx1, c0 * lx0);
    }
    reinterpret_cast<f32x4*>(out)[(((size_t)b * Hout + Y) * Wout + X) * 12 + c4] = v;
}

__global__ void maxpool2_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H,
                                int W) {
    const int Ho = H / 2, Wo = W / 2;
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t pix = gid / 12;
    const int c4 = gid - pix * 12;
    if (pix >= (size_t)B * Ho * Wo) return;
    const int x = pix % Wo;
    const int y = (pix / Wo) % Ho;
    const int b = pix / ((size_t)Wo * Ho);
    const f32x4* s = reinterpret_cast<const f32x4*>(in) + (size_t)b * H * W * 12 + c4;
    const f32x4 a = s[((size_t)(2 * y) * W + 2 * x) * 12], bb = s[((size_t)(2 * y) * W + 2 * x + 1) * 12];
    const f32x4 c = s[((size_t)(2 * y + 1) * W + 2 * x) * 12],
                d = s[((size_t)(2 * y + 1) * W + 2 * x + 1) * 12];
    f32x4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = fmaxf(fmaxf(a[r], bb[r]), fmaxf(c[r], d[r]));
    reinterpret_cast<f32x4*>(out)[gid] = v;
}

__global__ void nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int C,
                                    int H, int W, int Cpad) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // over B*H*W*Cpad
    const size_t hw = (size_t)H * W;
    if (idx >= (size_t)B * hw * Cpad) return;
    const int c = idx % Cpad;
    const size_t p = (idx / Cpad) % hw;
    const int b = idx / ((size_t)Cpad * hw);
    out[idx] = c < C ? in[((size_t)b * C + c) * hw + p] : 0.f;
}

__global__ void nhwc_to_nchw_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int C,
                                    int H, int W, int Cpad) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // over B*C*H*W
    const size_t hw = (size_t)H * W;
    if (idx >= (size_t)B * C * hw) return;
    const size_t p = idx % hw;
    const int c = (idx / hw) % C;
    const int b = idx / ((size_t)C * hw);
    out[idx] = in[((size_t)b * hw + p) * Cpad + c];
}

// PostConvs[1]: 1x1 conv 48 -> 3 (networks/unet.py:713-720)
__global__ void conv1x1_out_kernel(const float* __restrict__ feat, const float* __restrict__ w,
                                   const float* __restrict__ bias, float* __restrict__ out_nchw,
                                   float* __restrict__ out_nhwc4, int B, int H, int W) {
    __shared__ float ws[3 * 48 + 3];
    if (threadIdx.x < 147) ws[threadIdx.x] = threadIdx.x < 144 ? w[threadIdx.x] : bias[threadIdx.x - 144];
    __syncthreads();
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t hw = (size_t)H * W;
    if (idx >= (size_t)B * hw) return;
    const f32x4* f = reinterpret_cast<const f32x4*>(feat) + idx * 12;
    float o0 = 0.f, o1 = 0.f, o2 = 0.f;
#pragma unroll
    for (int q = 0; q < 12; ++q) {
        const f32x4 v = f[q];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            o0 += v[r] * ws[4 * q + r];
            o1 += v[r] * ws[48 + 4 * q + r];
            o2 += v[r] * ws[96 + 4 * q + r];
        }
    }
    o0 += ws[144];
    o1 += ws[145];
    o2 += ws[146];
    const size_t b = idx / hw, p = idx - b * hw;
    if (out_nchw) {
        out_nchw[(b * 3 + 0) * hw + p] = o0;
        out_nchw[(b * 3 + 1) * hw + p] = o1;
        out_nchw[(b * 3 + 2) * hw + p] = o2;
    }
    if (out_nhwc4) reinterpret_cast<f32x4*>(out_nhwc4)[idx] = f32x4{o0, o1, o2, 0.f};
}

// sum |a-b| and sum (a-b)^2 (models/recurrent_model.py:512-525, util/util.py:9-20)
__global__ void loss_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n,
                                    double* __restrict__ partial) {
    double s1 = 0.0, s2 = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float d = a[i] - b[i];
        s1 += (double)fabsf(d);
        s2 += (double)d * (double)d;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_down(s1, o);
        s2 += __shfl_down(s2, o);
    }
    __shared__ double sh[2][4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) {
        sh[0][wv] = s1;
        sh[1][wv] = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
        partial[2 * blockIdx.x + 1] = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
    }
}
__global__ void loss_final_kernel(const double* __restrict__ partial, int nblk, double* __restrict__ res) {
    double s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 64) {
        s1 += partial[2 * i];
        s2 += partial[2 * i + 1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_down(s1, o);
        s2 += __shfl_down(s2, o);
    }
    if (threadIdx.x == 0) {
        res[0] = s1;
        res[1] = s2;
    }
}

inline unsigned nblocks(size_t n, int bs) { return (unsigned)((n + bs - 1) / bs); }

}  // namespace

hipError_t launch_demosaic(const float* raw, float* green_scratch, float* out, int n, int h, int w,
                           int64_t bstride, int pstride, int cstride, hipStream_t s, int64_t raw_bstride) {
    const size_t npix = (size_t)n * 4 * h * w;
    if (!npix) return hipSuccess;
    const int64_t rbs = raw_bstride ? raw_bstride : (int64_t)4 * h * w;
    hipLaunchKernelGGL(ha_green_kernel, dim3(nblocks(npix, 256)), dim3(256), 0, s, raw, green_scratch, n, h, w, rbs);
    hipLaunchKernelGGL(ha_rb_kernel, dim3(nblocks(npix, 256)), dim3(256), 0, s, raw, green_scratch, out, n, h,
                       w, bstride, pstride, cstride, rbs);
    return hipGetLastError();
}

// HamiltonAdam.remosaick (util/Hamilton_Adam_demo.py:237-246) from the NHWC4 previous output: packed GBRG planes
// [B][4][H/2][W/2] = G(even row, even col), B(even, odd), R(odd, even), G(odd, odd).  Pure indexing.
__global__ void remosaick4_kernel(const float* __restrict__ rgb4, float* __restrict__ raw, int B, int H, int W) {
    const int h = H / 2, w = W / 2;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * 4 * h * w) return;
    const int x = idx % w;
    const int y = (idx / w) % h;
    const int c = (idx / ((size_t)w * h)) % 4;
    const int b = idx / ((size_t)4 * w * h);
    const int yy = 2 * y + (c >> 1), xx = 2 * x + (c & 1);
    const int ch = c == 0 || c == 3 ? 1 : (c == 1 ? 2 : 0);
    raw[idx] = rgb4[(((size_t)b * H + yy) * W + xx) * 4 + ch];
}

hipError_t launch_remosaick4(const float* rgb4, float* raw, int B, int H, int W, hipStream_t s) {
    const size_t n = (size_t)B * H * W;
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(remosaick4_kernel, dim3(nblocks(n, 256)), dim3(256), 0, s, rgb4, raw, B, H, W);
    return hipGetLastError();
}

hipError_t launch_warp3(const float* src4, const float* flow_raw, float* dst, int dpstride, int B, int H,
                        int W, hipStream_t s) {
    const size_t n = (size_t)B * H * W;
    hipLaunchKernelGGL(warp3_kernel, dim3(nblocks(n, 256)), dim3(256), 0, s, src4, flow_raw, dst, dpstride, B,
                       H, W);
    return hipGetLastError();
}

hipError_t launch_netin(const float* raw_cur, float* green_scratch, const float* prev4, const float* flow_prev,
                        const float* next4, const float* flow_next, float* netin, int B, int h, int w, hipStream_t s,
                        int64_t raw_bstride, int64_t flow_bstride, const float* proj_w16, const float* proj_b, float* proj_out) {
    const size_t n = (size_t)B * 4 * h * w;
    if (!n) return hipSuccess;
    if (n >= 0x100000000ull) return hipErrorInvalidValue;      // netin_pixel's 32-bit pixel index
    const int64_t rbs = raw_bstride ? raw_bstride : (int64_t)4 * h * w, fbs = flow_bstride ? flow_bstride : (int64_t)2 * h * w;
    hipLaunchKernelGGL(ha_green_kernel, dim3(nblocks(n, 256)), dim3(256), 0, s, raw_cur, green_scratch, B, h, w, rbs);
    const NetinArgs a{raw_cur, green_scratch, prev4, flow_prev, next4, flow_next, B, h, w, rbs, fbs};
    if (proj_w16)
        hipLaunchKernelGGL(netin_proj_kernel, dim3(nblocks(n, 256)), dim3(256), 0, s, a, proj_w16, proj_b, proj_out);
    else
        hipLaunchKernelGGL(netin_kernel, dim3(nblocks(n, 256)), dim3(256), 0, s, a, netin);
    return hipGetLastError();
}

hipError_t launch_netin_bound(const float* raw_a, const float* raw_b, const float* raw_c, int B, int h, int w, int64_t raw_bstride,
                              const unsigned* prev_words, unsigned* words, hipStream_t s, unsigned* zero_a, size_t zero_na,
                              unsigned* zero_b, size_t zero_nb) {
    const int64_t n = (int64_t)4 * h * w;
    if (B <= 0 || n <= 0) return hipSuccess;
    if ((n & 3) || (raw_bstride & 3)) return hipErrorInvalidValue;      // 16-B loads (n = 4hw: always)
    int nblk = (int)((n / 4 + 256 * 8 - 1) / (256 * 8));
    nblk = nblk < 1 ? 1 : (nblk > 64 ? 64 : nblk);
    hipLaunchKernelGGL(netin_bound_kernel, dim3(nblk, B), dim3(256), 0, s, raw_a, raw_b, raw_c, n, raw_bstride ? raw_bstride : n,
                       prev_words, words, zero_a, zero_na, zero_b, zero_nb);
    return hipGetLastError();
}

bool g_small_prestage = true;      // prestage_set_small: false = the three pre-stage kernels at every size (A/B reference, tests)
void prestage_set_small(bool on) { g_small_prestage = on; }
// Can a frame-step of this size take the one-kernel pre-stage (netin_small_kernel)?  No future frame, at most 1024 tiles of 16x16.
bool netin_small_applies(int B, int h, int w, bool future) {
    const long tiles = (long)B * ((2 * h + 15) / 16) * ((2 * w + 15) / 16);
    return g_small_prestage && !future && h >= 1 && w >= 1 && tiles <= 1024;
}
hipError_t launch_netin_small(const float* raw_cur, const float* raw_prev, const float* prev4, const float* flow_prev, float* netin, int B,
                              int h, int w, int64_t raw_bstride, int64_t flow_bstride, const unsigned* prev_words, unsigned* words,
                              hipStream_t s, unsigned* zero_a, size_t zero_na, unsigned* zero_b, size_t zero_nb) {
    if (B <= 0 || h <= 0 || w <= 0) return hipSuccess;
    const int tiles_x = (2 * w + 15) / 16, tiles_y = (2 * h + 15) / 16;
    const int64_t rbs = raw_bstride ? raw_bstride : (int64_t)4 * h * w, fbs = flow_bstride ? flow_bstride : (int64_t)2 * h * w;
    const NetinArgs a{raw_cur, nullptr, prev4, flow_prev, nullptr, nullptr, B, h, w, rbs, fbs};
    hipLaunchKernelGGL(netin_small_kernel, dim3(tiles_x * tiles_y, B), dim3(256), 0, s, a, netin, raw_prev, prev_words, words, zero_a,
                       zero_na, zero_b, zero_nb, tiles_x);
    return hipGetLastError();
}

hipError_t launch_warp48(const float* src, const float* flow_raw, float* dst, int B, int H, int W,
                         hipStream_t s, int64_t flow_bstride) {
    if (!B || !H || !W) return hipSuccess;
    hipLaunchKernelGGL(warp48_kernel, dim3((W + 31) / 32, H, B), dim3(192), 0, s, src, flow_raw, dst, B, H, W,
                       flow_bstride ? flow_bstride : (int64_t)2 * (H / 2) * (W / 2));
    return hipGetLastError();
}

hipError_t launch_warp48_proj(const float* src, const float* flow_raw, float* dst, int B, int H, int W, const float* frag,
                              int inv_e, const float* bias, hipStream_t s, int64_t flow_bstride) {
    if (!B || !H || !W) return hipSuccess;
    const int64_t fbs = flow_bstride ? flow_bstride : (int64_t)2 * (H / 2) * (W / 2);
    hipLaunchKernelGGL(warp48_proj_kernel, dim3((W + 31) / 32, H, B), dim3(192), 0, s, src, flow_raw, dst, B, H, W, fbs, frag, inv_e, bias);
    return hipGetLastError();
}

// max |x| per sequence of a dense map [B][n] (rvdd_internal.h: amax words of the maps that no kernel of a frame-step wrote --
// caller-supplied network inputs and features, a recurrent state handed in through rvdd_set_state)
// (`up`: binades added to the maximum, for a map that stands for something up to 2^up times larger)
__global__ __launch_bounds__(256) void amax_reduce_kernel(const float* __restrict__ map, int64_t n, unsigned* __restrict__ words, int up) {
    __shared__ unsigned red[4];
    const int b = blockIdx.y;
    const f32x4* p = reinterpret_cast<const f32x4*>(map + (size_t)b * n);
    const int64_t n4 = n >> 2;
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f32x4 v = p[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = fmaxf(m, fabsf(map[(size_t)b * n + 4 * n4 + threadIdx.x]));
    unsigned bits = __float_as_uint(m);
    const unsigned e = (bits >> 23) & 0xffu;
    if (up && e >= 1 && e + up < 254) bits += (unsigned)up << 23;
    amax_commit_block(words, b, blockIdx.x, __uint_as_float(bits), red);
}

hipError_t launch_amax_reduce(const float* map, int B, int64_t hw_c, unsigned* words, hipStream_t s, int up) {
    if (B <= 0 || hw_c <= 0) return hipSuccess;
    if ((hw_c & 3) && B > 1) return hipErrorInvalidValue;      // 16-B loads per sequence
    int nblk = (int)((hw_c / 4 + 256 * 8 - 1) / (256 * 8));
    nblk = nblk < 1 ? 1 : (nblk > 128 ? 128 : nblk);
    hipLaunchKernelGGL(amax_reduce_kernel, dim3(nblk, B), dim3(256), 0, s, map, hw_c, words, up);
    return hipGetLastError();
}

// The border ring of the composed first layer (runtime.hip compose_pre_enc0).  The composition treats the preprocessing
// layer's output y as if it existed outside the image (b1 + partial windows of the zero-extended input); the reference pads it
// with ZEROS (networks/unet.py:742-743, padding=1 on both convs).  For a border pixel p the composed sum therefore carries
// sum over the taps d with q = p + d - 1 outside the image of W2[d] y~(q), y~(q) = b1 + sum over taps e with q + e - 1 inside of
// W1[e] x(q + e - 1): subtracted here.  One 64-thread block per border pixel (2 W + 2 (H - 2) of them per sequence), thread = channel.
// netin NHWC16, w1 [9][16][48 m], w2 [9][48 m][48 o], part NHWC48.
__global__ __launch_bounds__(64) void pre_border_fix_kernel(const float* __restrict__ netin, const float* __restrict__ w1,
                                                            const float* __restrict__ b1, const float* __restrict__ w2,
                                                            float* __restrict__ part, int H, int W) {
    __shared__ float ys[48];
    const int b = blockIdx.y, t = threadIdx.x;
    int i = blockIdx.x, py, px;
    if (i < W) { py = 0; px = i; }
    else if (i < 2 * W) { py = H - 1; px = i - W; }
    else if (i < 2 * W + H - 2) { py = i - 2 * W + 1; px = 0; }
    else { py = i - 2 * W - (H - 2) + 1; px = W - 1; }
    const float* x = netin + (size_t)b * H * W * kNetInC;
    float corr = 0.f;
    for (int d = 0; d < 9; ++d) {
        const int qy = py + d / 3 - 1, qx = px + d % 3 - 1;
        if ((unsigned)qy < (unsigned)H && (unsigned)qx < (unsigned)W) continue;      // q inside: the composition is right
        float y = 0.f;
        if (t < 48) {
            y = b1[t];
            for (int e = 0; e < 9; ++e) {
                const int ry = qy + e / 3 - 1, rx = qx + e % 3 - 1;
                if ((unsigned)ry >= (unsigned)H || (unsigned)rx >= (unsigned)W) continue;
                const float* xp = x + ((size_t)ry * W + rx) * kNetInC;
                for (int c = 0; c < kNetInC; ++c) y = fmaf(w1[(e * kNetInC + c) * 48 + t], xp[c], y);
            }
            ys[t] = y;
        }
        __syncthreads();
        if (t < 48)
            for (int m = 0; m < 48; ++m) corr = fmaf(w2[((size_t)d * 48 + m) * 48 + t], ys[m], corr);
        __syncthreads();
    }
    if (t < 48) part[((size_t)b * H * W + (size_t)py * W + px) * kF + t] -= corr;
}

hipError_t launch_pre_border_fix(const float* netin, const float* w1, const float* b1, const float* w2, float* part, int B, int H, int W,
                                 hipStream_t s) {
    if (B <= 0 || H < 2 || W < 2) return hipSuccess;
    hipLaunchKernelGGL(pre_border_fix_kernel, dim3(2 * W + 2 * (H - 2), B), dim3(64), 0, s, netin, w1, b1, w2, part, H, W);
    return hipGetLastError();
}

hipError_t launch_warp_nchw(const float* x, const float* flow, float* y, int n, int c, int H, int W,
                            hipStream_t s) {
    const size_t np = (size_t)n * H * W;
    if (!np) return hipSuccess;
    hipLaunchKernelGGL(warp_nchw_kernel, dim3(nblocks(np, 256)), dim3(256), 0, s, x, flow, y, n, c, H, W);
    return hipGetLastError();
}

hipError_t launch_upsample_flow(const float* t, float* out, int nc, int h, int w, float mul, hipStream_t s) {
    const size_t n = (size_t)nc * 4 * h * w;
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(upsample_flow_kernel, dim3(nblocks(n, 256)), dim3(256), 0, s, t, out, nc, h, w, mul);
    return hipGetLastError();
}

hipError_t launch_upsample2x(const float* in, float* out, int B, int h, int w, int Hout, int Wout, int oy,
                             int ox, bool align_corners, hipStream_t s) {
    if (!B || !Hout || !Wout) return hipSuccess;
    hipLaunchKernelGGL(upsample2x_kernel, dim3((Wout + 15) / 16, Hout, B), dim3(192), 0, s, in, out, B, h, w, Hout, Wout,
                       oy, ox, align_corners ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_maxpool2(const float* in, float* out, int B, int H, int W, hipStream_t s) {
    const size_t n = (size_t)B * (H / 2) * (W / 2) * 12;
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(maxpool2_kernel, dim3(nblocks(n, 192)), dim3(192), 0, s, in, out, B, H, W);
    return hipGetLastError();
}

hipError_t launch_nchw_to_nhwc(const float* in, float* out, int B, int C, int H, int W, int Cpad,
                               hipStream_t s) {
    const size_t n = (size_t)B * H * W * Cpad;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(nblocks(n, 256)), dim3(256), 0, s, in, out, B, C, H, W, Cpad);
    return hipGetLastError();
}

hipError_t launch_nhwc_to_nchw(const float* in, float* out, int B, int C, int H, int W, int Cpad,
                               hipStream_t s) {
    const size_t n = (size_t)B * C * H * W;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(nblocks(n, 256)), dim3(256), 0, s, in, out, B, C, H, W, Cpad);
    return hipGetLastError();
}

hipError_t launch_conv1x1_out(const float* feat, const float* w3x48, const float* b3, float* out_nchw,
                              float* out_nhwc4, int B, int H, int W, hipStream_t s) {
    const size_t n = (size_t)B * H * W;
    hipLaunchKernelGGL(conv1x1_out_kernel, dim3(nblocks(n, 256)), dim3(256), 0, s, feat, w3x48, b3, out_nchw,
                       out_nhwc4, B, H, W);
    return hipGetLastError();
}

hipError_t launch_loss_reduce(const float* a, const float* b, int64_t n, double* partial, int nblk,
                              double* result2, hipStream_t s) {
    hipLaunchKernelGGL(loss_partial_kernel, dim3(nblk), dim3(256), 0, s, a, b, n, partial);
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, s, partial, nblk, result2);
    return hipGetLastError();
}
