// Internal declarations shared by the HIP translation units of librvdd_hip.so.
// Activation maps live in HBM as NHWC fp32 with C = 48 (192 B per pixel, three
// 64-B lines); the network input is NHWC with C padded to 16; 3-channel
// frames that are gathered by the bicubic warp are NHWC with C padded to 4.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kF = 48;        // feature channels everywhere (filters=48)
constexpr int kNetInC = 16;   // padded channel count of the network input map

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: `done` (one per kernel
// instantiation) remembers the devices of this process that already have it, one bit per device ordinal.
inline hipError_t allow_dynamic_lds(const void* kern, size_t bytes, std::atomic<uint64_t>& done) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
    return e;
}
// compute units of the current device (queried once per device ordinal)
inline int current_device_cus() {
    static std::atomic<int> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    int v = cache[dev & 63].load(std::memory_order_relaxed);
    if (v > 0) return v;
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    cache[dev & 63].store(cus, std::memory_order_relaxed);
    return cus;
}

// ---------------------------------------------------------------- conv3x3 --
// EPI_RELU_OUT3 (Winograd kernel only): ReLU, store the 48-channel map AND apply the final 1x1 conv 48->3
enum ConvEpi { EPI_NONE = 0, EPI_RELU = 1, EPI_POOL = 2, EPI_RELU_ADD2 = 3, EPI_RELU_OUT3 = 4 };

struct ConvArgs {
    const float* in;      // NHWC [B][H][W][CIN]
    const float* w;       // arranged [9][CIN/16][3][64 lanes][4] (see arrange_conv3x3 / arrange_wino3x3 in runtime.hip)
    const float* bias;    // [48]   (ignored when acc_in != nullptr)
    const float* acc_in;  // NHWC48 [B][H][W] partial sums to start from, or nullptr
    const float* res1;    // EPI_RELU_ADD2: out = relu(conv) + res1 + res2
    const float* res2;
    float* out;           // NHWC48 [B][Hout][Wout]
    int B, H, W;          // conv domain (input size = conv output size)
    int Hout, Wout;       // size of the map `out` points to
    int oy, ox;           // placement of the conv output inside it (zero_pad_features)
    int ups;              // Winograd kernel, 48 -> 48, ReLU only: `in` is [B][H/2][W/2][48] and the conv input is its
                          // bilinear x2 upsample (align_corners=False), interpolated in the patch load
    int tiles_x, tiles_y, ntiles;
    // EPI_RELU_OUT3: PostConvs[1] (networks/unet.py:713-720) fused into PostConvs[0]'s epilogue
    const float* w3;      // [3][48]
    const float* b3;      // [3]
    float* out3_nchw;     // [B][3][H][W]
    float* out3_nhwc4;    // [B][H][W][4] copy for the next frame's warp, or nullptr
    float wscale;         // conv3x3h.hip: 2^-s, the filters having been scaled by 2^s before the f16 split
    // conv3x3h.hip, block floating point per map and sequence (see amax_shift below): amax_in[b][kAmaxSeqWords] = words whose
    // maximum is the bits of max |x| over sequence b of the input map (null: no scaling), amax_out[b][kAmaxSeqWords] receives
    // the same for the map this launch writes (null: nobody reads that map as a split operand)
    const unsigned* amax_in;
    unsigned* amax_out;
};

// ---- block floating point for the split-f16 operands --------------------------------------------------------------------
// An f16 half saturates at 65504 and loses its low bits below 2^-14; an f32 activation has neither limit.  Every map that
// a split-f16 kernel reads therefore carries, per sequence, kAmaxLines words (one per 128-byte line) whose maximum is the
// bits of max |x| over the map, written by the kernel that produced the map (atomic max on the integer bits, which order
// like the non-negative floats they are).  The reader multiplies by the power of two that puts that maximum into
// [2^3, 2^4) before it splits -- the range the trained nets' activations have by themselves -- and scales the sums back in
// its epilogue; both exact.  A zero / non-finite / absurdly small maximum (< 2^-95) means "no scaling".
// Atomics are the cost to watch here (measured, profiles/r04_amax_atomics.txt): the waves of a persistent kernel finish
// together, and one atomic per wave on one word per sequence (2048 on 8 addresses) serialised for 13 us at the tail of every
// conv launch; one per wave of the streaming kernels (10^5 per launch) cost 4 % of a frame-step even spread over 64 words of
// two cache lines.  So: a workgroup reduces through LDS first (amax_commit_block), and the workgroups of a launch spread
// over kAmaxLines separate lines per sequence.
constexpr int kAmaxTargetExp = 3;
constexpr int kAmaxLines = 16;          // words per sequence, each at the start of its own 128-byte line
constexpr int kAmaxLineWords = 32;
constexpr int kAmaxSeqWords = kAmaxLines * kAmaxLineWords;
// The shift for a map whose max |x| has these bits.  A maximum inside [2^-6, 2^12) -- every map of the trained nets on frames
// in the reference's [-1, 1] -- needs none: the f16 halves are as far from overflow (x16 at least) and as precise relative to
// the map's maximum (2^-19 at worst) as they need to be, the reader then skips the multiplication (it is not free: the chunk
// loop of conv3x3h_kernel has no spare vector-issue slot, 18 v_pk_mul_f32 per tile cost 1.5 % of a layer) and in-domain
// frames keep the bits they had before the scaling existed.  Outside the window: the shift that puts the maximum into [8, 16).
__host__ __device__ inline int amax_shift(unsigned bits) {
    const int e = (int)((bits >> 23) & 0xffu);
    if (e < 32 || e == 255) return 0;                      // zero, absurdly small, inf / nan: leave alone
    if (e >= 127 - 6 && e < 127 + 12) return 0;
    return kAmaxTargetExp - (e - 127);
}
__host__ __device__ inline float pow2f(int k) {          // 2^k, |k| <= 126
    const unsigned u = (unsigned)(127 + k) << 23;
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}
#ifdef __HIPCC__
// v (+ | max) the same variable of the lane whose index differs in bit 0 / 1 (a quad neighbour: DPP quad_perm) or in bit 4 / 5
// (v_permlane16_swap / v_permlane32_swap): what `v + __shfl_xor(v, MASK)` / `fmaxf(v, __shfl_xor(v, MASK))` give, bit for bit
// (addition and maximum commute) -- without the LDS round trip of the ds_bpermute_b32 hipcc lowers __shfl_xor to.  Every lane active.
template <int MASK>
__device__ __forceinline__ void lane_xor_pair(float v, float& a, float& b) {
    static_assert(MASK == 1 || MASK == 2 || MASK == 16 || MASK == 32, "quad neighbours and row pairs only");
    const int u = __builtin_bit_cast(int, v);
    if constexpr (MASK == 1) {
        a = v;
        b = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, u, 0xB1, 0xf, 0xf, true));      // quad_perm:[1,0,3,2]
    } else if constexpr (MASK == 2) {
        a = v;
        b = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, u, 0x4E, 0xf, 0xf, true));      // quad_perm:[2,3,0,1]
    } else if constexpr (MASK == 16) {
        // {own, partner} or {partner, own}.  (Elements into scalars first: __builtin_bit_cast of r[1] itself reads r[0].)
        const auto r = __builtin_amdgcn_permlane16_swap((unsigned)u, (unsigned)u, false, false);
        const unsigned r0 = r[0], r1 = r[1];
        a = __builtin_bit_cast(float, r0);
        b = __builtin_bit_cast(float, r1);
    } else {
        const auto r = __builtin_amdgcn_permlane32_swap((unsigned)u, (unsigned)u, false, false);
        const unsigned r0 = r[0], r1 = r[1];
        a = __builtin_bit_cast(float, r0);
        b = __builtin_bit_cast(float, r1);
    }
}
template <int MASK>
__device__ __forceinline__ float lane_xor_add(float v) {
    float a, b;
    lane_xor_pair<MASK>(v, a, b);
    return a + b;
}
template <int MASK>
__device__ __forceinline__ float lane_xor_max(float v) {
    float a, b;
    lane_xor_pair<MASK>(v, a, b);
    return fmaxf(a, b);
}
// max over the 64 lanes of a wave of an unsigned value (DPP; every lane must be active): the result, wave-uniform
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true));      // row_shr:1
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true));      // row_shr:2
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true));      // row_shr:4
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true));      // row_shr:8
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true));      // row_bcast:15 into rows 1, 3
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true));      // row_bcast:31 into rows 2, 3
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// the reader's side: lane l < kAmaxLines holds word l of its sequence (other lanes 0) -> the maximum, wave-uniform
__device__ __forceinline__ unsigned amax_lines_max(unsigned v) {
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 15);
}
// One WORKGROUP's contribution to the amax words of sequence b (the same b in every thread): m >= 0 per lane; `red` = one
// LDS word per wave; `spread` = any number that differs between the workgroups that finish together.  Contains a
// __syncthreads(): every thread of the workgroup must arrive.  One atomic per workgroup, nothing returned, nobody waits.
__device__ __forceinline__ void amax_commit_block(unsigned* words, int b, unsigned spread, float m, unsigned* red) {
    const unsigned mx = wave_max_u32(__float_as_uint(m));
    const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) red[wave] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned t = red[0];
        for (int i = 1; i < nw; ++i) t = max(t, red[i]);
        if (t) atomicMax(words + (size_t)b * kAmaxSeqWords + (spread % kAmaxLines) * kAmaxLineWords, t);
    }
}
#endif
// amax words of a dense NHWC map [B][HW][C] (the maps no split kernel's producer wrote: caller-supplied inputs and states)
hipError_t launch_amax_reduce(const float* map, int B, int64_t hw_c, unsigned* words, hipStream_t s, int up = 0);


// cin = 16 or 48.  Returns hipGetLastError().
hipError_t launch_conv3x3(const ConvArgs& a, int cin, int epi, hipStream_t s);
size_t conv3x3_weight_floats(int cin);
// Winograd F(2x2,3x3) variant for 48 -> 48 layers; a.w = bank arranged by arrange_wino3x3 (runtime.hip)
hipError_t launch_wino3x3(const ConvArgs& a, int cin, int epi, hipStream_t s);   // cin: 48, or 16 = the zero-padded network input
size_t wino3x3_weight_floats();
// the same layers on the F16 matrix pipe with split f32 operands (conv3x3h.hip); a.w = the split bank arranged by
// arrange_conv3x3h (runtime.hip), a.wscale its scale; cin 48 (every epilogue, with or without a.acc_in, with a.ups for
// UpConv's fused upsample) or 16 (the zero-padded network input): every 3x3 conv of the convunet by default
hipError_t launch_conv3x3h(const ConvArgs& a, int cin, int epi, hipStream_t s);
size_t conv3x3h_weight_bytes(int cin);
// preprocessing_layer composed with the first source of EncoderConvs[0][0]: one 5x5 conv of the 16-channel network input
// (conv3x3h.hip HGeo KS = 5; runtime.hip compose_pre_enc0), and the fix of its border ring: part -= sum over the 3x3 taps d
// whose pixel q = p + d - 1 lies OUTSIDE the image of W2[d] (b1 + sum over taps e inside the image of W1[e] x(q + e - 1))
hipError_t launch_conv5x5h_c16(const ConvArgs& a, hipStream_t s);
size_t conv5x5h_weight_bytes();
hipError_t launch_pre_border_fix(const float* netin, const float* w1, const float* b1, const float* w2, float* part, int B, int H, int W,
                                 hipStream_t s);
void conv3x3h_set_cout_split(bool on);   // false: no launch takes the output-channel split of small launches (conv3x3h.hip MT = 1); process-wide
void conv3x3h_set_groups(int g);   // stand-alone harness (-DRVDD_CONV_GROUPS2): 2 = two groups of four waves with an 8x16 tile each
void conv3x3_set_variant(int v);   // A/B switch used by rvdd_debug_conv_bench only

// -------------------------------------------------------------- pre-stages --
// Hamilton-Adams: raw [n][4][h][w] -> green plane scratch [n][2h][2w] -> RGB
// written at out[b*bstride + (y*W+x)*pstride + c*cstride].  raw_bstride: floats between the raw frames of consecutive
// sequences (0 = dense, 4hw): the caller's raw frames may be channel slices of a wider [B][4k][h][w] tensor.
hipError_t launch_demosaic(const float* raw, float* green_scratch, float* out, int n, int h, int w,
                           int64_t bstride, int pstride, int cstride, hipStream_t s, int64_t raw_bstride = 0);

// bicubic backward warp with the raw-resolution flow (x2 bilinear upsample,
// align_corners=True, times 2, fused).  src NHWC4 -> dst[(b*H*W + p)*dpstride + c], c<3.
hipError_t launch_warp3(const float* src4, const float* flow_raw, float* dst, int dpstride, int B,
                        int H, int W, hipStream_t s);
// The NHWC16 network input of a step in one pass: warp3 of prev4 | demosaic of raw_cur | warp3 of next4 (or zeros when
// next4 == nullptr); flows nullptr = --no_warp.  [B][4][h][w] raw, [B][2][h][w] flows, dense inside a sequence, with
// raw_bstride / flow_bstride floats from one sequence to the next (0 = dense); green_scratch [B][2h][2w].
// proj_w16 set (ConvNeXtUnet): the 16-channel map is not written; `proj_out` NHWC48 receives its 1x1 projection
// proj_w16 / proj_b (launch_proj1x1's (16, 0) arrangement) instead (prestage.hip netin_proj_kernel)
hipError_t launch_netin(const float* raw_cur, float* green_scratch, const float* prev4, const float* flow_prev,
                        const float* next4, const float* flow_next, float* netin, int B, int h, int w, hipStream_t s,
                        int64_t raw_bstride = 0, int64_t flow_bstride = 0, const float* proj_w16 = nullptr,
                        const float* proj_b = nullptr, float* proj_out = nullptr);
// amax words of that network input (block floating point of the split-f16 convs, below): an upper bound from the packed raw
// frames it is made of (up to three, each nullable) and from `prev_words`, words that bound the previous output (nullable)
// (zero_a / zero_b, nullable: word ranges the kernel also clears -- the next step's amax words, runtime.hip)
hipError_t launch_netin_bound(const float* raw_a, const float* raw_b, const float* raw_c, int B, int h, int w, int64_t raw_bstride,
                              const unsigned* prev_words, unsigned* words, hipStream_t s, unsigned* zero_a = nullptr, size_t zero_na = 0,
                              unsigned* zero_b = nullptr, size_t zero_nb = 0);
// src NHWC48 -> dst NHWC48.
// the three pre-stage kernels of a small frame-step without a future frame in one launch (prestage.hip netin_small_kernel); same bits
bool netin_small_applies(int B, int h, int w, bool future);
// (raw_prev: the first step of a video, whose bound also covers the previous raw frame -- prev_words is null then)
hipError_t launch_netin_small(const float* raw_cur, const float* raw_prev, const float* prev4, const float* flow_prev, float* netin, int B,
                              int h, int w, int64_t raw_bstride, int64_t flow_bstride, const unsigned* prev_words, unsigned* words,
                              hipStream_t s, unsigned* zero_a, size_t zero_na, unsigned* zero_b, size_t zero_nb);
void prestage_set_small(bool on);      // false: never; process-wide
hipError_t launch_warp48(const float* src, const float* flow_raw, float* dst, int B, int H, int W,
                         hipStream_t s, int64_t flow_bstride = 0);
// the same, and a 48 -> 48 projection of every warped pixel: dst = W warp(src) + bias; frag / inv_e = the projection as a
// NextProj's split-f16 fragments and scale (runtime_next.inc), applied per pixel in block floating point (prestage.hip)
hipError_t launch_warp48_proj(const float* src, const float* flow_raw, float* dst, int B, int H, int W, const float* frag,
                              int inv_e, const float* bias, hipStream_t s, int64_t flow_bstride = 0);
// generic NCHW warp with a full-resolution flow (util.flow_utils.warp).
hipError_t launch_remosaick4(const float* rgb4, float* raw, int B, int H, int W, hipStream_t s);
hipError_t launch_warp_nchw(const float* x, const float* flow, float* y, int n, int c, int H, int W,
                            hipStream_t s);
hipError_t launch_upsample_flow(const float* t, float* out, int nc, int h, int w, float mul,
                                hipStream_t s);

// ---------------------------------------------------------- map reshaping --
// bilinear x2 of a NHWC48 map [B][h][w] into [B][Hout][Wout] at offset (oy,ox);
// pixels outside the 2h x 2w window are zero.  align_corners selects
// nn.Upsample(align_corners=...) semantics.
hipError_t launch_upsample2x(const float* in, float* out, int B, int h, int w, int Hout, int Wout,
                             int oy, int ox, bool align_corners, hipStream_t s);
hipError_t launch_maxpool2(const float* in, float* out, int B, int H, int W, hipStream_t s);
hipError_t launch_nchw_to_nhwc(const float* in, float* out, int B, int C, int H, int W, int Cpad,
                               hipStream_t s);
hipError_t launch_nhwc_to_nchw(const float* in, float* out, int B, int C, int H, int W, int Cpad,
                               hipStream_t s);
// final 1x1 conv 48->3: feat NHWC48 -> out NCHW [B][3][H][W] (+ NHWC4 copy for the next warp)
hipError_t launch_conv1x1_out(const float* feat, const float* w3x48, const float* b3, float* out_nchw,
                              float* out_nhwc4, int B, int H, int W, hipStream_t s);
// partial[2*nblk] doubles scratch; result2 device floats {sum|d|, sum d^2} as doubles -> host math
hipError_t launch_loss_reduce(const float* a, const float* b, int64_t n, double* partial, int nblk,
                              double* result2, hipStream_t s);

// -------------------------------------------------------------- ConvNeXt ---
struct NextBlockW {          // device pointers, one ConvBlock (networks/new_unet.py:74-103)
    const float* proj_w;     // [CinP][48] (k-major) or nullptr
    const float* proj_b;     // [48]
    const float* dw_w;       // [49][48]  (tap-major)
    const float* dw_b;       // [48]
    const float* ln_w;       // [48]
    const float* ln_b;       // [48]
    const float* fc1_w;      // arranged for MFMA, see runtime.hip
    const float* fc1_b;      // [192]
    const float* fc2_w;
    const float* fc2_b;      // [48]
    const float* ls;         // [48]
    // the fused kernel's split-f16 MLP (convnext.hip SPLIT): filter fragments of 2^s fc1 / 2^s' fc2 as f16 hi, lo halves
    // (arranged in runtime_next.inc), the scales and their inverses; fc1_h null = the f32-MFMA form
    const float* fc1_h;
    const float* fc2_h;
    float fc1_scale, fc1_inv, fc2_scale, fc2_inv;
    float gelu_c[7][2];      // gelu_phi4_scaled's constants for s = fc1_inv: C_i s^(6-i), cap / s -- each TWICE: as scalar pairs
                             // they are packed-f32 operands as they stand (hipcc 7.2 folds a splat of ONE kernel-argument scalar
                             // into the pair that starts at it, i.e. (c_i, c_i+1): wrong numbers, found the hard way)
    int pipe;                // 1 = convblock_pipe_kernel (front / back waves pipelined over tiles) instead of convblock_kernel
};
// A 48 -> 48 projection applied to a fused block's OUTPUT in its epilogue (convblock_pipe_kernel PROJ): one half of the
// 96 -> 48 projection behind a concat, proj(cat(a, b)) = Wa a + Wb b + bias (networks/new_unet.py:85-88, 321-329)
struct NextProj {
    const float* frag;       // split-f16 fragments of 2^s W [48][48] (runtime_next.inc arrange_proj_half), 9 KiB
    const float* bias;       // [48] or null; used when `add` is null
    const float* add;        // NHWC48 map of the block's output size added to the projection (the other half, with the bias), or null
    int inv_e;               // -s: added to a float's exponent field it multiplies by 2^-s
};
// the whole ConvBlock in ONE kernel (convblock_kernel): x -> x + ls * MLP(LayerNorm(dwconv7x7(x))); x, out NHWC48
// [B][H][W], out != x.  _out3: also the 1x1 conv 48 -> 3 on the block's output (out_nchw [B][3][H*W], out_nhwc4
// [B][H*W][4]; either may be null)
hipError_t launch_next_block(const float* x, float* out, const NextBlockW& w, int B, int H, int W, hipStream_t s);
// the same, and MaxPool2d(2) of the block's output into pooled [B][H/2][W/2] (NHWC48) from the epilogue
hipError_t launch_next_block_pool(const float* x, float* out, float* pooled, const NextBlockW& w, int B, int H, int W, hipStream_t s);
hipError_t launch_next_block_out3(const float* x, float* out, const NextBlockW& w, int B, int H, int W, const float* w3x48,
                                  const float* b3, float* out_nchw, float* out_nhwc4, hipStream_t s);
// the fused block (pipelined split-f16 form only: w.fc1_h set) storing pj's projection of its output (+ pj.bias + pj.add) in
// place of the output; pooled (nullable): MaxPool2d(2) of the UNPROJECTED output, as launch_next_block_pool
hipError_t launch_next_block_proj(const float* x, float* out, float* pooled, const NextBlockW& w, const NextProj& pj, int B, int H,
                                  int W, hipStream_t s);
// 1x1 projection on MFMA: in1 NHWC[c1] (+ in2 NHWC[c2]) -> out NHWC48; (c1,c2) = (16,0) or (48,48)
hipError_t launch_proj1x1(const float* in1, int c1, const float* in2, int c2, const float* w,
                          const float* b, float* out, int64_t npix, hipStream_t s);
// zero_pad_features: src [B][h][w] -> dst [B][H][W] at (oy,ox), zeros elsewhere (NHWC48)
hipError_t launch_pad_copy(const float* src, float* dst, int B, int h, int w, int H, int W, int oy, int ox,
                           hipStream_t s);

// srgb.hip -- sRGB post-processing + display-domain metrics (dataset/fwd_ppipe.py)
hipError_t launch_ppipe(const float* img, int n, int H, int W, int64_t sn, int64_t sc, int64_t sy, int64_t sx, int bit_depth,
                        const float gains[3], int iso, uint8_t* out_u8, float* out_f32, hipStream_t s);
size_t srgb_metrics_workspace(int n, int H, int W);
hipError_t launch_srgb_metrics(const uint8_t* a, const uint8_t* b, int n, int H, int W, void* ws, hipStream_t s);

// ------------------------------------------------------------------ TV-L1 --
struct Tvl1Workspace;
hipError_t tvl1_alloc(Tvl1Workspace** out, int nx, int ny);
void tvl1_free(Tvl1Workspace* w);
int tvl1_num_scales(int nx, int ny);
bool tvl1_size_ok(int nx, int ny);   // false where the reference's own pyramid reads out of bounds (very skinny images)
// I0, I1 [ny][nx] -> u [2][ny][nx]; synchronises the stream every few iterations (convergence peek)
hipError_t tvl1_run(Tvl1Workspace* w, const float* I0, const float* I1, float* u, hipStream_t st, int* total_iters);
hipError_t tvl1_run_batch(Tvl1Workspace* w, const float* I0, const float* I1, float* u, int n, hipStream_t st, int* iters, bool async = false);
hipError_t tvl1_check(Tvl1Workspace* w, hipStream_t st);      // what an asynchronous batch left unread (synchronises when something is pending)
int tvl1_ws_nx(const Tvl1Workspace* w);
int tvl1_ws_ny(const Tvl1Workspace* w);
