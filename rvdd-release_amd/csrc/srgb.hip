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
    float g0, g1, g2;   // fwd_ppipe.py:29
};

__device__ __forceinline__ float to_4095(float v, int bit_depth) {
    if (bit_depth == -1) {                 // util/util.py:40 then fwd_ppipe.py:134
        v = (v + 1.0f) / 2.0f * 255.0f;
        v = v / 255.0f * 4095.0f;
    } else if (bit_depth == 0) {
        v = v * 4095.0f;
    } else if (bit_depth == 8) {
        v = v / 255.0f * 4095.0f;
    } else if (bit_depth == 10) {
        v = v / 1024.0f * 4095.0f;
    }
    return v;
}

__device__ __forceinline__ float linearise(float v, int iso) {
    // fwd_ppipe.py:50-57: undo the REDS<->CRVD percentile matching, subtract the black level
    if (iso == 3200) v = (v - 266.0f) * 2060.0f / 3344.0f + 245.0f;
    if (iso == 12800) v = (v - 268.0f) * 2060.0f / 3807.0f + 245.0f;
    return (v - 240.0f) / 3855.0f;
}

__device__ __forceinline__ float tone(float v) {
    // fwd_ppipe.py:69-72: gamma 1/2.2 above 1e-8, then the smoothstep 3x^2 - 2x^3
    if (v > 1e-8f) v = powf(v, (float)(1.0 / 2.2));
    const float v2 = v * v;
    return 3.0f * v2 - 2.0f * (v2 * v);
}

__global__ __launch_bounds__(256) void ppipe_kernel(PpipeArgs a) {
    const int64_t total = (int64_t)a.n * a.H * a.W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % a.W);
        const int y = (int)((i / a.W) % a.H);
        const int b = (int)(i / ((int64_t)a.W * a.H));
        const float* p = a.img + b * a.sn + y * a.sy + x * a.sx;
        float r = linearise(to_4095(p[0], a.bit_depth), a.iso) / a.g0;
        float g = linearise(to_4095(p[a.sc], a.bit_depth), a.iso) / a.g1;
        float bl = linearise(to_4095(p[2 * a.sc], a.bit_depth), a.iso) / a.g2;
        // apply_mat_inv_ccm (fwd_ppipe.py:20-26): out[c] = sum_j in[j] * inv_ccm[c][j]
        float o[3];
        o[0] = r * 1.07955733f + g * -0.40125771f + bl * 0.32170038f;
        o[1] = r * -0.15390743f + g * 1.35677921f + bl * -0.20287178f;
        o[2] = r * -0.00235972f + g * -0.55155296f + bl * 1.55391268f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float s = tone(o[c]) * 255.0f;
            if (a.out_f32) a.out_f32[i * 3 + c] = s;
            // fwd_ppipe.py:141: round (half to even) -> clip -> uint8
            a.out_u8[i * 3 + c] = (uint8_t)fminf(fmaxf(rintf(s), 0.0f), 255.0f);
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
    PpipeArgs a{img, out_u8, out_f32, sn, sc, sy, sx, n, H, W, bit_depth, iso, gains[0], gains[1], gains[2]};
    const int64_t total = (int64_t)n * H * W;
    if (total == 0) return hipSuccess;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
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
