// sRGB post-processing of the network output and the display-domain metrics
// (reference: dataset/fwd_ppipe.py:48-86, :131-141; util/util.py:40).
//
// Compiled with -ffp-contract=off: the reference evaluates the chain one fp32 operation at a time
// (numpy, then torch), and the uint8 rounding at the end makes single-ulp differences visible.
//
// ppipe_kernel   pointwise, HBM-bound: 12 B read + 3 B written per pixel (+12 B with the float copy).
// ssd_kernel     exact integer sum of squared uint8 differences (PSNR).
// ssim_kernel    7x7 uniform-window SSIM; the window sums of uint8 data are exact in 32-bit integers,
//                only the per-pixel rational expression is float64 (skimage works in float64 throughout).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rvdd_internal.h"

namespace {

struct PpipeArgs {
    const float* img;
    uint8_t* out_u8;
    float* out_f32;
    int64_t sn, sc, sy, sx;
    int n, H, W;
    int bit_depth;      // -1: network output in [-1,1]; 0, 8, 10: fwd_ppipe.py:131-137; anything else: already [0,4095]
    int iso;
    float g[3];         // fwd_ppipe.py:29
    float rg[3];        // RN(1/g)
};

// x / c for a divisor known at launch time, correctly rounded (bit-identical to the IEEE division the reference
// performs) in 3 instructions instead of the ~10 of v_div_*: Markstein's sequence with rc = RN(1/c) --
// q = RN(x*rc); r = x - c*q (exact in an FMA); RN(q + r*rc) = RN(x/c).  Checked against np.float32 division on
// 16M samples per divisor used here; inputs are finite and far from the subnormal range.
__device__ __forceinline__ float div_c(float x, float c, float rc) {
    const float q = x * rc;
    const float r = __builtin_fmaf(-c, q, x);
    return __builtin_fmaf(r, rc, q);
}
#define DIVC(x, c) div_c((x), (c), (float)(1.0 / (double)(c)))

__device__ __forceinline__ float to_4095(float v, int bit_depth) {
    if (bit_depth == -1) {                 // util/util.py:40 then fwd_ppipe.py:134
        v = (v + 1.0f) / 2.0f * 255.0f;
        v = DIVC(v, 255.0f) * 4095.0f;
    } else if (bit_depth == 0) {
        v = v * 4095.0f;
    } else if (bit_depth == 8) {
        v = DIVC(v, 255.0f) * 4095.0f;
    } else if (bit_depth == 10) {
        v = v / 1024.0f * 4095.0f;
    }
    return v;
}

__device__ __forceinline__ float linearise(float v, int iso) {
    // fwd_ppipe.py:50-57: undo the REDS<->CRVD percentile matching, subtract the black level
    if (iso == 3200) v = DIVC((v - 266.0f) * 2060.0f, 3344.0f) + 245.0f;
    if (iso == 12800) v = DIVC((v - 268.0f) * 2060.0f, 3807.0f) + 245.0f;
    return DIVC(v - 240.0f, 3855.0f);
}

__device__ __forceinline__ float tone(float v) {
    // fwd_ppipe.py:69-72: gamma 1/2.2 above 1e-8, then the smoothstep 3x^2 - 2x^3
    if (v > 1e-8f) v = powf(v, (float)(1.0 / 2.2));
    const float v2 = v * v;
    return 3.0f * v2 - 2.0f * (v2 * v);
}

// one pixel: three linear camera values -> three display values x 255 (before rounding)
__device__ __forceinline__ void pixel(const PpipeArgs& a, float r, float g, float bl, float o[3]) {
    r = div_c(linearise(to_4095(r, a.bit_depth), a.iso), a.g[0], a.rg[0]);
    g = div_c(linearise(to_4095(g, a.bit_depth), a.iso), a.g[1], a.rg[1]);
    bl = div_c(linearise(to_4095(bl, a.bit_depth), a.iso), a.g[2], a.rg[2]);
    // apply_mat_inv_ccm (fwd_ppipe.py:20-26): out[c] = sum_j in[j] * inv_ccm[c][j]
    o[0] = tone(r * 1.07955733f + g * -0.40125771f + bl * 0.32170038f) * 255.0f;
    o[1] = tone(r * -0.15390743f + g * 1.35677921f + bl * -0.20287178f) * 255.0f;
    o[2] = tone(r * -0.00235972f + g * -0.55155296f + bl * 1.55391268f) * 255.0f;
}

// fwd_ppipe.py:141: round (half to even) -> clip -> uint8
__device__ __forceinline__ unsigned to_u8(float s) { return (unsigned)fminf(fmaxf(rintf(s), 0.0f), 255.0f); }

// any strides, one pixel per thread
__global__ __launch_bounds__(256) void ppipe_kernel(PpipeArgs a) {
    const int64_t total = (int64_t)a.n * a.H * a.W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % a.W);
        const int y = (int)((i / a.W) % a.H);
        const int b = (int)(i / ((int64_t)a.W * a.H));
        const float* p = a.img + b * a.sn + y * a.sy + x * a.sx;
        float o[3];
        pixel(a, p[0], p[a.sc], p[2 * a.sc], o);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (a.out_f32) a.out_f32[i * 3 + c] = o[c];
            a.out_u8[i * 3 + c] = (uint8_t)to_u8(o[c]);
        }
    }
}

// planar input with unit x stride (the network's NCHW output), W % 4 == 0, 16-byte aligned rows:
// four pixels per thread -- three 16-byte loads, one 12-byte store (+ three 16-byte stores for the float copy)
__global__ __launch_bounds__(256) void ppipe_planar4_kernel(PpipeArgs a) {
    const int W4 = a.W >> 2;
    const int64_t total = (int64_t)a.n * a.H * W4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x4 = (int)(i % W4);
        const int y = (int)((i / W4) % a.H);
        const int b = (int)(i / ((int64_t)W4 * a.H));
        const float* p = a.img + b * a.sn + y * a.sy + x4 * 4;
        const float4 R = *reinterpret_cast<const float4*>(p);
        const float4 G = *reinterpret_cast<const float4*>(p + a.sc);
        const float4 B = *reinterpret_cast<const float4*>(p + 2 * a.sc);
        const float rr[4] = {R.x, R.y, R.z, R.w}, gg[4] = {G.x, G.y, G.z, G.w}, bb[4] = {B.x, B.y, B.z, B.w};
        float o[12];
#pragma unroll
        for (int k = 0; k < 4; ++k) pixel(a, rr[k], gg[k], bb[k], o + 3 * k);
        unsigned w[3];
#pragma unroll
        for (int k = 0; k < 3; ++k)
            w[k] = to_u8(o[4 * k]) | (to_u8(o[4 * k + 1]) << 8) | (to_u8(o[4 * k + 2]) << 16) | (to_u8(o[4 * k + 3]) << 24);
        unsigned* q = reinterpret_cast<unsigned*>(a.out_u8 + i * 12);
        q[0] = w[0]; q[1] = w[1]; q[2] = w[2];
        if (a.out_f32) {
            float4* f = reinterpret_cast<float4*>(a.out_f32 + i * 12);
            f[0] = make_float4(o[0], o[1], o[2], o[3]);
            f[1] = make_float4(o[4], o[5], o[6], o[7]);
            f[2] = make_float4(o[8], o[9], o[10], o[11]);
        }
    }
}

__global__ __launch_bounds__(256) void ssd_kernel(const uint8_t* a, const uint8_t* b, int64_t per_image, unsigned long long* ssd) {
    const int img = blockIdx.y;
    const uint8_t* pa = a + (int64_t)img * per_image;
    const uint8_t* pb = b + (int64_t)img * per_image;
    unsigned long long acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_image; i += (int64_t)gridDim.x * blockDim.x) {
        const int d = (int)pa[i] - (int)pb[i];
        acc += (unsigned)(d * d);
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&ssd[img], acc);     // integer: order-independent
}

constexpr int ST_W = 32, ST_H = 8, WIN = 7, PAD = 3;

__global__ __launch_bounds__(ST_W * ST_H) void ssim_kernel(const uint8_t* a, const uint8_t* b, int H, int W, double* partial) {
    __shared__ uint8_t ta[ST_H + WIN - 1][(ST_W + WIN - 1) * 3];
    __shared__ uint8_t tb[ST_H + WIN - 1][(ST_W + WIN - 1) * 3];
    __shared__ double red[ST_W * ST_H / 64];
    const int img = blockIdx.z;
    const int vw = W - 2 * PAD, vh = H - 2 * PAD;               // the region skimage keeps (crop of (win-1)/2)
    const int ox = blockIdx.x * ST_W, oy = blockIdx.y * ST_H;   // tile origin in valid-region coordinates = image coords of the window's corner
    const uint8_t* pa = a + (int64_t)img * H * W * 3;
    const uint8_t* pb = b + (int64_t)img * H * W * 3;
    const int row_bytes = (ST_W + WIN - 1) * 3;
    for (int i = threadIdx.x; i < (ST_H + WIN - 1) * row_bytes; i += ST_W * ST_H) {
        const int ry = i / row_bytes, rb = i % row_bytes;
        const int y = oy + ry, xb = ox * 3 + rb;
        const bool ok = y < H && xb < W * 3;
        ta[ry][rb] = ok ? pa[(int64_t)y * W * 3 + xb] : 0;
        tb[ry][rb] = ok ? pb[(int64_t)y * W * 3 + xb] : 0;
    }
    __syncthreads();
    const int tx = threadIdx.x % ST_W, ty = threadIdx.x / ST_W;
    double s = 0.0;
    if (ox + tx < vw && oy + ty < vh) {
        const double NP = WIN * WIN, cov_norm = NP / (NP - 1.0);
        const double C1 = (0.01 * 255.0) * (0.01 * 255.0), C2 = (0.03 * 255.0) * (0.03 * 255.0);
        for (int c = 0; c < 3; ++c) {
            unsigned sa = 0, sb = 0, saa = 0, sbb = 0, sab = 0;
#pragma unroll
            for (int dy = 0; dy < WIN; ++dy)
#pragma unroll
                for (int dx = 0; dx < WIN; ++dx) {
                    const unsigned va = ta[ty + dy][(tx + dx) * 3 + c], vb = tb[ty + dy][(tx + dx) * 3 + c];
                    sa += va; sb += vb; saa += va * va; sbb += vb * vb; sab += va * vb;
                }
            const double ux = sa / NP, uy = sb / NP, uxx = saa / NP, uyy = sbb / NP, uxy = sab / NP;
            const double vx = cov_norm * (uxx - ux * ux), vy = cov_norm * (uyy - uy * uy), vxy = cov_norm * (uxy - ux * uy);
            s += ((2.0 * ux * uy + C1) * (2.0 * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2));
        }
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < ST_W * ST_H / 64; ++i) t += red[i];
        partial[((int64_t)img * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = t;
    }
}

// fixed-order sum of the per-tile partials: one block per image, deterministic
__global__ __launch_bounds__(256) void ssim_final_kernel(const double* partial, int per_image, double* out) {
    __shared__ double red[256];
    const double* p = partial + (int64_t)blockIdx.x * per_image;
    double s = 0.0;
    for (int i = threadIdx.x; i < per_image; i += 256) s += p[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}

}  // namespace

hipError_t launch_ppipe(const float* img, int n, int H, int W, int64_t sn, int64_t sc, int64_t sy, int64_t sx, int bit_depth,
                        const float gains[3], int iso, uint8_t* out_u8, float* out_f32, hipStream_t s) {
    PpipeArgs a{img, out_u8, out_f32, sn, sc, sy, sx, n, H, W, bit_depth, iso, {gains[0], gains[1], gains[2]},
                {(float)(1.0 / (double)gains[0]), (float)(1.0 / (double)gains[1]), (float)(1.0 / (double)gains[2])}};
    int64_t total = (int64_t)n * H * W;
    if (total == 0) return hipSuccess;
    const bool planar4 = sx == 1 && (W & 3) == 0 && ((sn | sc | sy) & 3) == 0 && (reinterpret_cast<uintptr_t>(img) & 15) == 0 &&
                         (reinterpret_cast<uintptr_t>(out_u8) & 3) == 0 && (reinterpret_cast<uintptr_t>(out_f32) & 15) == 0;
    if (planar4) total >>= 2;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (planar4)
        hipLaunchKernelGGL(ppipe_planar4_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL(ppipe_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
    return hipGetLastError();
}

size_t srgb_metrics_workspace(int n, int H, int W) {
    const int gx = (W - 2 * PAD + ST_W - 1) / ST_W, gy = (H - 2 * PAD + ST_H - 1) / ST_H;
    return (size_t)n * (sizeof(unsigned long long) + sizeof(double)) + (size_t)n * gx * gy * sizeof(double);
}

// ws layout: [n] u64 SSD | [n] double SSIM sums | [n*gx*gy] double partials.  The first 2n words are the result.
hipError_t launch_srgb_metrics(const uint8_t* a, const uint8_t* b, int n, int H, int W, void* ws, hipStream_t s) {
    const int gx = (W - 2 * PAD + ST_W - 1) / ST_W, gy = (H - 2 * PAD + ST_H - 1) / ST_H;
    unsigned long long* ssd = static_cast<unsigned long long*>(ws);
    double* sums = reinterpret_cast<double*>(ssd + n);
    double* partial = sums + n;
    hipError_t e = hipMemsetAsync(ssd, 0, (size_t)n * sizeof(unsigned long long), s);
    if (e != hipSuccess) return e;
    const int64_t per_image = (int64_t)H * W * 3;
    int bx = (int)((per_image + 256 * 16 - 1) / (256 * 16));
    if (bx > 2048) bx = 2048;
    hipLaunchKernelGGL(ssd_kernel, dim3(bx, n), dim3(256), 0, s, a, b, per_image, ssd);
    hipLaunchKernelGGL(ssim_kernel, dim3(gx, gy, n), dim3(ST_W * ST_H), 0, s, a, b, H, W, partial);
    hipLaunchKernelGGL(ssim_final_kernel, dim3(n), dim3(256), 0, s, partial, gx * gy, sums);
    return hipGetLastError();
}
