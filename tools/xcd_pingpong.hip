// Microbenchmark (gfx950), round 6: what a tagged 16-byte record costs between two workgroups -- TV-L1's exchange (tvl1.hip scale_kernel_patch: the
// publisher stores {tag, a, b, tag}, the poller loads it with sc1 until both tags match) -- by WHERE the two workgroups sit and by the store's flavour:
//   pairs on the SAME XCD (blocks b and b + 8 share one: MI355X_MICROARCH.md, workgroup dispatch; verified here with HW_REG_XCC_ID) or on different ones,
//   stores with sc1 (write-through past the XCD's L2, which drops the line: what the kernel uses, valid at any placement) or plain (the line stays in the
//   XCD's L2: only a reader on the same XCD can see it there).
// Each pair plays ping-pong N times; prints shader cycles per round trip (two one-way hand-offs) and whether every hand-off completed.
//   hipcc -O3 --offload-arch=gfx950 tools/xcd_pingpong.hip -o tools/scratch/xcdpp && tools/scratch/xcdpp
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

// block pairs: (p, partner(p)); role 0 starts.  rec[pair][2 directions] 16-byte records, 256 B apart.
template <bool PLAIN>
__global__ __launch_bounds__(64) void pingpong(u32x4* rec, long long* cyc, unsigned* xcc, int* fail, int n, int stride) {
    const int b = blockIdx.x;
    const int pair = (b % stride) + (b / (2 * stride)) * stride, role = (b / stride) & 1;      // partner = b +- stride
    if (threadIdx.x == 0) xcc[b] = xcc_id();
    if (threadIdx.x != 0) return;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)rec, 0, 1 << 20, 0x00020000);
    const unsigned mine = (unsigned)(pair * 2 + role) * 256u, theirs = (unsigned)(pair * 2 + (role ^ 1)) * 256u;
    const long long t0 = __builtin_amdgcn_s_memtime();
    bool ok = true;
    for (int i = 1; i <= n && ok; ++i) {
        const unsigned tag = (unsigned)i;
        if (role == 0) {
            const u32x4 v = {tag, tag * 3u, tag * 5u, tag};
            if (PLAIN) __builtin_amdgcn_raw_buffer_store_b128(v, r, mine, 0, 0);
            else __builtin_amdgcn_raw_buffer_store_b128(v, r, mine, 0, 16);
        }
        unsigned spins = 0;
        for (;;) {
            asm volatile("" ::: "memory");      // (the builtin is an ordinary read to the compiler: without this the poll is hoisted out of its loop)
            const u32x4 g = __builtin_amdgcn_raw_buffer_load_b128(r, theirs, 0, 16 /* sc1 */);
            if (g[0] == tag && g[3] == tag) {
                if (g[1] != tag * 3u || g[2] != tag * 5u) ok = false;
                break;
            }
            if (++spins > (1u << 18)) { ok = false; break; }
        }
        if (role == 1) {
            const u32x4 v = {tag, tag * 3u, tag * 5u, tag};
            if (PLAIN) __builtin_amdgcn_raw_buffer_store_b128(v, r, mine, 0, 0);
            else __builtin_amdgcn_raw_buffer_store_b128(v, r, mine, 0, 16);
        }
    }
    cyc[b] = __builtin_amdgcn_s_memtime() - t0;
    if (!ok) atomicAdd(fail, 1);
}

template <bool PLAIN>
void run(const char* label, int stride, int nblk) {
    u32x4* rec;
    long long* cyc;
    unsigned* xcc;
    int* fail;
    (void)hipMalloc(&rec, 1 << 20);
    (void)hipMalloc(&cyc, nblk * 8);
    (void)hipMalloc(&xcc, nblk * 4);
    (void)hipMalloc(&fail, 4);
    (void)hipMemset(rec, 0, 1 << 20);
    (void)hipMemset(fail, 0, 4);
    const int n = 4000;
    void* args[] = {&rec, &cyc, &xcc, &fail, (void*)&n, (void*)&stride};
    // cooperative: every block resident (the partners wait for each other)
    hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(pingpong<PLAIN>), dim3(nblk), dim3(64), args, 0, 0);
    (void)hipDeviceSynchronize();
    std::vector<long long> hc(nblk);
    std::vector<unsigned> hx(nblk);
    int hf = 0;
    (void)hipMemcpy(hc.data(), cyc, nblk * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hx.data(), xcc, nblk * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost);
    int same = 0, pairs = 0;
    double sum = 0;
    for (int b = 0; b < nblk; ++b) {
        if ((b / stride) & 1) continue;
        ++pairs;
        if (hx[b] == hx[b + stride]) ++same;
        sum += (double)hc[b] / n;
    }
    printf("%-52s launch %s: %d pairs, %d on one XCD; %.0f cycles (100 MHz ticks x clock) per round trip; failed hand-offs: %d\n", label, hipGetErrorName(e), pairs, same,
           sum / pairs, hf);
    (void)hipFree(rec); (void)hipFree(cyc); (void)hipFree(xcc); (void)hipFree(fail);
}

int main() {
    for (int rep = 0; rep < 2; ++rep) {
        run<false>("sc1 store, partner = block + 8 (same XCD)", 8, 32);
        run<true>("plain store, partner = block + 8 (same XCD)", 8, 32);
        run<false>("sc1 store, partner = block + 1 (another XCD)", 1, 32);
        run<true>("plain store, partner = block + 1 (another XCD)", 1, 32);
        run<false>("sc1 store, partner = block + 4 (another XCD)", 4, 32);
    }
    return 0;
}
