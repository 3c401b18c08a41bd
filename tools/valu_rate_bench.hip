// Issue cost of the vector instructions the split-f16 kernels are made of, in cycles per instruction per wave:
// 16 independent copies of one instruction in a loop, one wave per SIMD (blockDim 256) and two (512), s_memtime around it.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_rate_bench.hip -o /tmp/vrb && /tmp/vrb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ void k(float* out, unsigned long long* cyc, int iters) {
    float a[16], b[16];
    unsigned u[16];
    for (int i = 0; i < 16; ++i) {
        a[i] = threadIdx.x * 0.001f + i;
        b[i] = 1.0f + i * 0.125f;
        u[i] = threadIdx.x + i;
    }
    float sc = 0.5f + (float)blockIdx.x * 1e-9f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define OP0(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
#define OP1(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double*)&a[i & ~1]) : "v"(*(double*)&b[i & ~1]));
#define OP2(i) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(a[i]), "v"(b[i]));
#define OP3(i) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(a[i]) : "v"(u[i]), "v"(b[i]));
#define OP4(i) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(u[i]) : "v"(a[i]), "v"(sc));
#define OP5(i) asm volatile("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(u[i]) : "v"(a[i]), "v"(sc), "v"(u[(i + 1) & 15]));
#define OP6(i) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(a[i]), "v"(b[i]));
#define OP7(i) asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 15]));
#define OP8(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double*)&a[i & ~1]) : "v"(*(double*)&b[i & ~1]));
#define OP9(i) asm volatile("v_exp_f32 %0, %1" : "=v"(a[i]) : "v"(b[i]));
#define OP10(i) asm volatile("v_ldexp_f32 %0, %1, %2" : "=v"(a[i]) : "v"(b[i]), "v"(u[i]));
#define OP11(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
#define OP12(i) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
#define OP13(i) asm volatile("v_pk_fma_f16 %0, %0, %1, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
#define OP14(i) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "=v"(u[i]) : "v"(u[(i + 1) & 15]));
#define OP15(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 15]));
        if constexpr (OP == 0) { REP16(OP0) }
        if constexpr (OP == 1) { REP16(OP1) }
        if constexpr (OP == 2) { REP16(OP2) }
        if constexpr (OP == 3) { REP16(OP3) }
        if constexpr (OP == 4) { REP16(OP4) }
        if constexpr (OP == 5) { REP16(OP5) }
        if constexpr (OP == 6) { REP16(OP6) }
        if constexpr (OP == 7) { REP16(OP7) }
        if constexpr (OP == 8) { REP16(OP8) }
        if constexpr (OP == 9) { REP16(OP9) }
        if constexpr (OP == 10) { REP16(OP10) }
        if constexpr (OP == 11) { REP16(OP11) }
        if constexpr (OP == 12) { REP16(OP12) }
        if constexpr (OP == 13) { REP16(OP13) }
        if constexpr (OP == 14) { REP16(OP14) }
        if constexpr (OP == 15) { REP16(OP15) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += a[i] + (float)u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int OP>
void run(const char* name, float* out, unsigned long long* cyc) {
    const int iters = 2000;
    double res[2];
    for (int w = 0; w < 2; ++w) {
        const int threads = w ? 512 : 256;
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
        unsigned long long c;
        hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        res[w] = (double)c / (iters * 16.0);
    }
    // s_memtime ticks at 100 MHz on this part: convert with the shader clock by comparing against v_add_f32 (4 cycles) outside
    printf("%-44s 1 wave/SIMD: %7.3f ticks/instr   2 waves/SIMD: %7.3f ticks/instr (per wave)\n", name, res[0], res[1]);
}

int main() {
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, 8);
    run<0>("v_add_f32", out, cyc);
    run<11>("v_fma_f32", out, cyc);
    run<1>("v_pk_mul_f32", out, cyc);
    run<8>("v_pk_fma_f32", out, cyc);
    run<2>("v_cvt_pkrtz_f16_f32", out, cyc);
    run<6>("v_cvt_pk_f16_f32", out, cyc);
    run<3>("v_fma_mix_f32 (f16 src)", out, cyc);
    run<4>("v_fma_mixlo_f16", out, cyc);
    run<5>("v_fma_mixhi_f16 (f16 src2)", out, cyc);
    run<7>("v_max3_f32 |.|", out, cyc);
    run<15>("v_med3_f32", out, cyc);
    run<9>("v_exp_f32", out, cyc);
    run<10>("v_ldexp_f32", out, cyc);
    run<12>("v_pk_add_f16", out, cyc);
    run<13>("v_pk_fma_f16", out, cyc);
    run<14>("v_mov_b32_dpp row_shr:1", out, cyc);
    return 0;
}
