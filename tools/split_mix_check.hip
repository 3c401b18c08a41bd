// v_fma_mixlo_f16 / v_fma_mixhi_f16 as the split of conv3x3h.hip (hi = f16(sc x), lo = f16(sc x - hi), one instruction per half)
// against the host's _Float16 arithmetic, bit for bit: hipcc -O3 --offload-arch=gfx950 tools/split_mix_check.hip -o /tmp/smc && /tmp/smc
#include <hip/hip_runtime.h>
__global__ void k(const float* x, float sc, unsigned* out) {
    float x0 = x[threadIdx.x * 2], x1 = x[threadIdx.x * 2 + 1];
    unsigned hi = 0, lo = 0;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(hi) : "v"(x0), "v"(sc));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hi) : "v"(x1), "v"(sc));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(x0), "v"(sc), "v"(hi));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(x1), "v"(sc), "v"(hi));
    out[threadIdx.x * 2] = hi; out[threadIdx.x * 2 + 1] = lo;
}
int main() {
    float hx[128]; for (int i = 0; i < 128; ++i) hx[i] = (i - 64) * 0.37123f + 1e-3f * i * i;
    float* dx; unsigned* dout; hipMalloc(&dx, 512); hipMalloc(&dout, 512);
    hipMemcpy(dx, hx, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, 0.25f, dout);
    unsigned ho[128]; hipMemcpy(ho, dout, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 64; ++t) for (int e = 0; e < 2; ++e) {
        float x = hx[2 * t + e] * 0.25f;
        _Float16 h = (_Float16)x; float r = x - (float)h; _Float16 l = (_Float16)r;
        unsigned short hb, lb; __builtin_memcpy(&hb, &h, 2); __builtin_memcpy(&lb, &l, 2);
        unsigned short gh = (ho[2 * t] >> (16 * e)) & 0xffff, gl = (ho[2 * t + 1] >> (16 * e)) & 0xffff;
        if (gh != hb || gl != lb) { if (bad < 5) printf("mismatch t=%d e=%d x=%g got %04x %04x want %04x %04x\n", t, e, x, gh, gl, hb, lb); ++bad; }
    }
    printf("bad = %d\n", bad);
    return bad != 0;
}
