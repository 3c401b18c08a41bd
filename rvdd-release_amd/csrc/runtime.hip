// librvdd_hip.so -- C ABI (include/rvdd.h), handle, weight repacking, workspace
// and the per-frame schedule of the recurrent denoise+demosaic path
// (models/recurrent_model.py:105-135 set_input, :161-349 forward, test branch).
#include "../../include/rvdd.h"
#include "rvdd_internal.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace {

thread_local std::string g_create_error;

struct HostTensor {
    std::vector<int64_t> shape;
    std::vector<float> data;
};

struct Conv3 {            // one 3x3 conv layer, split per 48-channel source
    int nsrc = 0;
    int cin_real[2] = {0, 0};
    int cin_pad[2] = {0, 0};
    float* w[2] = {nullptr, nullptr};
    float* wu[2] = {nullptr, nullptr};   // Winograd F(2x2,3x3) transformed bank (48-channel sources only)
    float* wh[2] = {nullptr, nullptr};   // split-f16 banks of conv3x3h.hip (hi / lo halves of 2^s w; 48-channel sources only)
    float wh_inv[2] = {1.f, 1.f};        // 2^-s of each
    float* bias = nullptr;
};

struct ProfClass {
    std::string name;
    int64_t seen = 0;
    int64_t launches = 0;
    double ms = 0, flops = 0, bytes = 0;
};
struct ProfPending {
    int cls;
    hipEvent_t e0, e1;
};

struct NextBlk {          // one ConvBlock of the ConvNeXt net (networks/new_unet.py:74-103)
    NextBlockW w{};
    int c1 = 0, c2 = 0;   // projection sources (0,0 = identity)
    // a 96 -> 48 projection as two halves for the epilogues of the blocks that form its two sources (convnext.hip PROJ):
    // half[0] over the first 48 input channels (frag, inv_e set; bias = proj_b), half[1] over the last 48 (no bias)
    NextProj half[2] = {};
};

// Every entry point acts on the handle's device, whatever the caller's current device is, and leaves
// the caller's current device as it found it.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int dev) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != dev) {
            err = hipSetDevice(dev);
            switched = err == hipSuccess;
        }
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// The 3x3 conv layers of the convunet by position in the schedule (run_convunet): names are resolved ONCE, in
// rvdd_finalize_weights, into h->cu[]; a frame-step touches no string and no map.
enum CuLayer {
    CU_PRE, CU_ENC0_0, CU_ENC0_1, CU_DOWN0, CU_ENC1_0, CU_ENC1_1, CU_DOWN1, CU_ENC2_0, CU_ENC2_1, CU_DOWN2, CU_ENC3_0,
    CU_ENC3_1, CU_BOT0, CU_BOT1, CU_UP0, CU_DEC0_0, CU_DEC0_1, CU_UP1, CU_DEC1_0, CU_DEC1_1, CU_UP2, CU_DEC2_0, CU_DEC2_1,
    CU_POST, CU_COUNT
};
const char* const kCuNames[CU_COUNT] = {
    "preprocessing_layer", "EncoderConvs.0.blocks.0.0", "EncoderConvs.0.blocks.1.0", "EncoderDown.0.conv",
    "EncoderConvs.1.blocks.0.0", "EncoderConvs.1.blocks.1.0", "EncoderDown.1.conv", "EncoderConvs.2.blocks.0.0",
    "EncoderConvs.2.blocks.1.0", "EncoderDown.2.conv", "EncoderConvs.3.blocks.0.0", "EncoderConvs.3.blocks.1.0",
    "bottleneck.0.0", "bottleneck.1.0", "DecoderUp.0.up.1", "DecoderConvs.0.blocks.0.0", "DecoderConvs.0.blocks.1.0",
    "DecoderUp.1.up.1", "DecoderConvs.1.blocks.0.0", "DecoderConvs.1.blocks.1.0", "DecoderUp.2.up.1",
    "DecoderConvs.2.blocks.0.0", "DecoderConvs.2.blocks.1.0", "PostConvs.0.0"};
constexpr int cu_enc(int level, int j) { return level == 0 ? CU_ENC0_0 + j : CU_ENC1_0 + 3 * (level - 1) + j; }
constexpr int cu_down(int i) { return CU_DOWN0 + 3 * i; }
constexpr int cu_up(int i) { return CU_UP0 + 3 * i; }
constexpr int cu_dec(int i, int j) { return CU_DEC0_0 + 3 * i + j; }
// amax words (rvdd_internal.h: block floating point of the split-f16 kernels): one slot of B x kAmaxSeqWords words per map a
// split kernel reads.  A SET of regular slots -- the output of every conv layer (its CuLayer), the network input, the features
// a caller hands to rvdd_unet_forward -- is written during one forward and must be zero when it starts.  Three sets: frame-steps
// use sets 0 and 1 in turn, and the first kernel of a step (netin_bound_kernel) zeroes the OTHER set for the step after it
// (nobody touches that set during this step; a memset node per step cost 2 % of a 0.3 ms frame); rvdd_unet_forward uses set 2
// and zeroes it itself.  The RECURRENT features' words cross the step boundary: the map PostConvs[0] writes in step t is the
// map step t + 1 gathers its warped features from (a bicubic gather never exceeds 1.9 x the map's maximum, far inside the
// margin of the scaling, so the warped map shares the words) -- three slots in rotation: step t reads (t + 2) % 3, writes
// t % 3, and its first kernel zeroes (t + 1) % 3.
enum { AMAX_REL_NETIN = CU_COUNT, AMAX_REL_FWDFEAT, AMAX_NREG };
enum { AMAX_FEAT0 = 3 * AMAX_NREG, AMAX_SLOTS = AMAX_FEAT0 + 3 };

// The ConvBlocks of the ConvNeXt net by position in the schedule (run_convnext), resolved once like the above.
enum NxBlock {
    NX_PRE, NX_ENC0_0, NX_ENC0_1, NX_DOWN0, NX_ENC1_0, NX_ENC1_1, NX_DOWN1, NX_ENC2_0, NX_ENC2_1, NX_DOWN2, NX_ENC3_0,
    NX_ENC3_1, NX_BOT0, NX_BOT1, NX_UP0, NX_DEC0_0, NX_DEC0_1, NX_UP1, NX_DEC1_0, NX_DEC1_1, NX_UP2, NX_DEC2_0, NX_DEC2_1,
    NX_POST0, NX_POST1, NX_COUNT
};
const char* const kNxNames[NX_COUNT] = {
    "preprocessing_layer.blocks.0", "encoder_convs.0.blocks.0", "encoder_convs.0.blocks.1", "encoder_downs.0.postconv",
    "encoder_convs.1.blocks.0", "encoder_convs.1.blocks.1", "encoder_downs.1.postconv", "encoder_convs.2.blocks.0",
    "encoder_convs.2.blocks.1", "encoder_downs.2.postconv", "encoder_convs.3.blocks.0", "encoder_convs.3.blocks.1",
    "bottleneck.blocks.0", "bottleneck.blocks.1", "decoder_ups.0.postconv", "decoder_convs.0.blocks.0",
    "decoder_convs.0.blocks.1", "decoder_ups.1.postconv", "decoder_convs.1.blocks.0", "decoder_convs.1.blocks.1",
    "decoder_ups.2.postconv", "decoder_convs.2.blocks.0", "decoder_convs.2.blocks.1", "postprocessing.0.blocks.0",
    "postprocessing.0.blocks.1"};
constexpr int nx_enc(int level, int j) { return level == 0 ? NX_ENC0_0 + j : NX_ENC1_0 + 3 * (level - 1) + j; }
constexpr int nx_down(int i) { return NX_DOWN0 + 3 * i; }
constexpr int nx_up(int i) { return NX_UP0 + 3 * i; }
constexpr int nx_dec(int i, int j) { return NX_DEC0_0 + 3 * i + j; }

struct Level {
    int H = 0, W = 0;
    float* t[3] = {nullptr, nullptr, nullptr};
    float* skip = nullptr;
    float* part = nullptr;
};

}  // namespace

struct rvdd_handle {
    rvdd_cfg cfg{};
    std::string err;
    bool finalized = false;
    bool need_init = true;
    bool force_wino = false;      // Winograd at every size (RVDD_CONV=winograd / rvdd_set_option "conv_kernel" 2): tests + measurement
    bool warp_raw = false;        // --warp_raw (rvdd_set_option): warp the re-mosaicked frames at raw resolution, demosaic afterwards
    bool prev_noisy = false;      // --prev_noisy_frame (rvdd_set_option): the next step's "previous frame" is the demosaiced noisy one
    bool no_warp = false;         // --no_warp (rvdd_set_option): previous output / features / next frame enter the net unwarped
    bool use_wino = true;         // 48->48 3x3 convs: Winograd F(2x2,3x3) (RVDD_CONV=direct selects the direct kernel)
    bool split16 = true;          // 48->48 3x3 convs on the F16 matrix pipe with split f32 operands (conv3x3h.hip); RVDD_CONV=f32 | direct |
                                  // winograd / "conv_kernel" 1, 2, 4 select the f32-MFMA kernels (the A/B reference)
    bool bfp = true;              // block floating point of the split-f16 convs (amax words per map and sequence; RVDD_BFP=0 / option "block_fp" 0:
                                  // operands split as they are, the round-3 behaviour with its 2^-14 .. 65504 domain -- A/B reference only)
    int seq_major = 0;            // 1 = full-resolution stages one sequence at a time (see seq_major_on)
    bool fuse_upsample = true;    // UpConv's bilinear x2 inside the Winograd patch load (RVDD_FUSE_UPSAMPLE=0: separate kernel)
    bool next_split = true;       // ConvNeXt, fused blocks: the two 1x1 convs on the F16 matrix pipe with split f32 operands (RVDD_NEXT_SPLIT=0: f32 MFMA)
    bool next_pipe = true;        // ConvNeXt, fused split-f16 blocks as a front / back pipeline over tiles (convblock_pipe_kernel; RVDD_NEXT_PIPE=0: convblock_kernel's phases)
    bool next_pool = true;        // ConvNeXt, fused blocks: MaxPool2d(2) from the epilogue of the block in front of a DownConv
    bool next_projfuse = true;    // ConvNeXt, pipelined split-f16 blocks: the 96 -> 48 projection behind a concat as two halves in the epilogues of
                                  // the blocks that form the concatenated maps (RVDD_NEXT_PROJFUSE=0 / option "next_projfuse" 0: proj1x1_kernel)
    bool netin_proj = false;      // lv[0].t[0] of the running step already holds the first ConvBlock's projection of the network input (run_prologue)
    bool featw_proj = false;      // `featw` of the running step holds W_f warp(features) + bias (run_prologue, next_pf_pre), not the warped features
    bool serpentine = false;      // sequence order of the current frame-step (flips every step when seq_major is on)
    std::map<std::string, HostTensor> staged;
    std::vector<void*> allocs;

    // weights
    Conv3 cu[CU_COUNT];       // convunet layers in schedule order (CuLayer)
    // preprocessing_layer composed with the first source of EncoderConvs[0][0] (feat nets; conv3x3h.hip HGeo, KS = 5)
    float* pre5_w = nullptr;  // the composed 5x5 bank (split f16)
    float pre5_inv = 1.f;
    float* pre5_b = nullptr;  // [48] composed bias
    float* pre_w1 = nullptr;  // [9][16][48]: preprocessing_layer weight, tap-major, input channel, its output channel m (border fix)
    float* pre_b1 = nullptr;  // [48]
    float* pre_w2 = nullptr;  // [9][48 m][48 o]: EncoderConvs[0][0] weight over the preprocessing layer's channels (border fix)
    bool fuse_pre = true;     // RVDD_FUSE_PRE=0 / option "fuse_pre" 0: the two layers one after the other (A/B reference)
    float* w_out = nullptr;   // [3][48]
    float* b_out = nullptr;   // [3]
    NextBlk nx[NX_COUNT];     // ConvNeXt blocks in schedule order (NxBlock)

    // workspace
    Level lv[4];
    float* netin = nullptr;      // NHWC16
    float* lastden4 = nullptr;   // NHWC4
    float* next4 = nullptr;      // NHWC4
    float* green = nullptr;      // [B][H][W]
    float* featw = nullptr;      // NHWC48 warped features
    float* lastfeat = nullptr;   // NHWC48 recurrent features
    unsigned* amax = nullptr;    // [AMAX_SLOTS][B][kAmaxSeqWords] max |x| per map and sequence, zeroed at the start of every forward
    int step_ctr = 0;            // frame-steps enqueued: picks the regular set (& 1) and the recurrent features' slots (% 3)
    int amax_base = 0;           // first slot of the regular set the current forward uses
    int amax_feat_in = 0, amax_post_out = 0;      // absolute slots the current forward reads the old features' words from / writes the new ones to
    bool amax_zero_pending = false;               // the next netin_bound launch zeroes the step-after-next's set and features slot
    double* loss_partial = nullptr;
    double* loss_result = nullptr;
    float* scratch = nullptr;
    size_t scratch_bytes = 0;
    Tvl1Workspace* tvl1 = nullptr;   // cached for the last (nx, ny)
    bool tvl1_async = false;         // option "tvl1_async": rvdd_tvl1flow_batch without iteration counts enqueues and returns (see rvdd.h)

    // hipGraph replay of a frame-step (see rvdd_step)
    struct StepKey {
        const void* p[6];
        int64_t stride[2];
        int flags;
        bool operator<(const StepKey& o) const {
            if (const int c = std::memcmp(p, o.p, sizeof p)) return c < 0;
            if (const int c = std::memcmp(stride, o.stride, sizeof stride)) return c < 0;
            return flags < o.flags;
        }
    };
    struct StepGraph {
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        uint64_t last_use = 0;
    };
    std::map<StepKey, StepGraph> graphs;
    uint64_t graph_tick = 0;
    int use_graphs = 0;             // RVDD_GRAPH=1 / rvdd_set_option "graphs" 1: replay captured frame-steps (measured slower, off)
    bool ran_eagerly = false;       // the first step of a handle is never captured (it sets the kernels' attributes)
    hipStream_t gstream = nullptr;  // the stream the graphs are captured on and replayed in
    hipEvent_t g_in = nullptr, g_out = nullptr;

    // measurement
    bool prof_on = false;
    std::string prof_filter;      // empty = every kernel class
    int prof_stride = 1;          // bracket every prof_stride-th launch of a class
    std::vector<ProfClass> prof;
    std::vector<ProfPending> pending;
    std::vector<hipEvent_t> event_pool;
    hipEvent_t t0 = nullptr, t1 = nullptr;

    bool has_feat() const { return cfg.arch == RVDD_ARCH_CONVUNET_FEAT || cfg.arch == RVDD_ARCH_CONVNEXT_FEAT; }
    bool is_next() const { return cfg.arch == RVDD_ARCH_CONVNEXT || cfg.arch == RVDD_ARCH_CONVNEXT_FEAT; }
    int cin_real() const { return 3 * (2 + cfg.future); }
};

namespace {

int fail(rvdd_t* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf; else g_create_error = buf;
    return code;
}

void drop_graphs(rvdd_t* h) {
    for (auto& kv : h->graphs) {
        if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
        if (kv.second.graph) (void)hipGraphDestroy(kv.second.graph);
    }
    h->graphs.clear();
}

#define ENTER(h)                                                                            \
    DeviceGuard guard__((h)->cfg.device);                                                   \
    if (guard__.err != hipSuccess)                                                          \
        return fail((h), RVDD_ERR_HIP, "cannot select device %d: %s", (h)->cfg.device, hipGetErrorString(guard__.err))

#define HIPCHK(h, expr)                                                                     \
    do {                                                                                    \
        hipError_t e__ = (expr);                                                            \
        if (e__ != hipSuccess)                                                              \
            return fail((h), RVDD_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                        __FILE__, __LINE__);                                                \
    } while (0)

#define RC(expr)               \
    do {                       \
        int rc__ = (expr);     \
        if (rc__) return rc__; \
    } while (0)

int dmalloc(rvdd_t* h, void** p, size_t bytes, bool zero = true) {
    HIPCHK(h, hipMalloc(p, bytes ? bytes : 16));
    h->allocs.push_back(*p);
    if (zero) HIPCHK(h, hipMemset(*p, 0, bytes ? bytes : 16));
    return RVDD_OK;
}

int upload(rvdd_t* h, float** dst, const std::vector<float>& v) {
    int rc = dmalloc(h, reinterpret_cast<void**>(dst), v.size() * sizeof(float), false);
    if (rc) return rc;
    HIPCHK(h, hipMemcpy(*dst, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    return RVDD_OK;
}

// OIHW [48][cin_total][3][3], channels [c0, c0+cn) -> [tap][j][m][lane = 16g + cout&15][i] with
// channel = c0 + 16j + 4g + i (zero beyond cn): the A-fragment order of conv3x3.hip (lane-linear).
std::vector<float> arrange_conv3x3(const HostTensor& t, int c0, int cn, int cin_pad) {
    const int cin_total = (int)t.shape[1];
    const int NJ = cin_pad / 16;
    std::vector<float> out((size_t)9 * NJ * 48 * 16, 0.f);
    for (int tap = 0; tap < 9; ++tap)
        for (int j = 0; j < NJ; ++j)
            for (int co = 0; co < 48; ++co)
                for (int g = 0; g < 4; ++g)
                    for (int i = 0; i < 4; ++i) {
                        const int c = 16 * j + 4 * g + i;
                        if (c >= cn) continue;
                        out[(((size_t)(tap * NJ + j) * 3 + co / 16) * 64 + g * 16 + co % 16) * 4 + i] =
                            t.data[(((size_t)co * cin_total + c0 + c) * 3 + tap / 3) * 3 + tap % 3];
                    }
    return out;
}

// OIHW [48][cin_total][3][3], channels [c0, c0+48) -> U = G g G^T per (cout, cin), stored
// [pos 16][j 3][m 3][lane = 16g + (cout&15)][i 4] with channel = c0 + 16j+4g+i: the A-fragment order of wino3x3.hip,
// lane-linear so that each lane group of a ds_read_b128 covers one whole 256-B bank row (no bank conflict).
// nj = 3: input channels c0 .. c0+47 of the filter; nj = 1: the first layer, channels 0 .. cin_total-1 (6 or 9)
// zero-padded to 16
std::vector<float> arrange_wino3x3(const HostTensor& t, int c0, int nj = 3) {
    const int cin_total = (int)t.shape[1];
    const int nc = nj == 3 ? 48 : (cin_total < 16 ? cin_total : 16);
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    std::vector<float> out((size_t)16 * nj * 3 * 256, 0.f);
    for (int co = 0; co < 48; ++co)
        for (int c = 0; c < nc; ++c) {
            const float* gk = &t.data[((size_t)co * cin_total + c0 + c) * 9];
            double tmp[4][3], u[4][4];
            for (int i = 0; i < 4; ++i)
                for (int k = 0; k < 3; ++k) tmp[i][k] = G[i][0] * gk[k] + G[i][1] * gk[3 + k] + G[i][2] * gk[6 + k];
            for (int i = 0; i < 4; ++i)
                for (int k = 0; k < 4; ++k) u[i][k] = tmp[i][0] * G[k][0] + tmp[i][1] * G[k][1] + tmp[i][2] * G[k][2];
            const int j = c / 16, g = (c % 16) / 4, ii = c % 4, m = co / 16, lr = co % 16;
            for (int pos = 0; pos < 16; ++pos)
                out[(((size_t)(pos * nj + j) * 3 + m) * 64 + g * 16 + lr) * 4 + ii] = (float)u[pos / 4][pos % 4];
        }
    return out;
}

// f32 -> f16 bits, toward zero (saturating) or to nearest even; f16 bits -> f32.  Host-side twins of v_cvt_pkrtz_f16_f32 /
// v_cvt_f16_f32 for the split filter banks.
uint16_t f16_bits(float x, bool toward_zero) {
    _Float16 hv = (_Float16)x;                   // to nearest even
    uint16_t u;
    std::memcpy(&u, &hv, 2);
    if (toward_zero) {
        if ((u & 0x7fffu) == 0x7c00u) u = (uint16_t)((u & 0x8000u) | 0x7bffu);      // an overflow saturates
        else if (std::fabs((float)hv) > std::fabs(x)) u = (uint16_t)(u - 1);        // magnitude one step down
    }
    return u;
}
float f16_value(uint16_t u) {
    _Float16 hv;
    std::memcpy(&hv, &u, 2);
    return (float)hv;
}

// OIHW [48][cin_total][3][3], channels [c0, c0+48) -> the split bank of conv3x3h.hip: w' = 2^s w (s = the largest power
// keeping |w'| <= 1024, returned as 2^-s), hi = f16(w') toward zero, lo = f16(w' - hi); stored
// [chunk 14][cout block 3][hi, lo][lane = 16g + (cout & 15)][e 8] as f16 with 8-channel group G = 4 chunk + g = channels
// 8 (G % 6) + e of tap G / 6 (groups 54, 55 zero): lane-linear 16-B A fragments of v_mfma_f32_16x16x32_f16.
// (ks = 5: the composed first layer, [48][cin][5][5], 13 chunks)
std::vector<float> arrange_conv3x3h(const HostTensor& t, int c0, float* inv_scale, int cin_pad = 48, int ks = 3) {
    const int cin_total = (int)t.shape[1];
    const int nc = cin_pad == 48 ? 48 : std::min(cin_total, 16);      // 16: the first layer, 6 or 9 real channels, zero filters beyond
    const int ntap = ks * ks;
    const int gpt = cin_pad / 8, ng = ntap * gpt, nch = (ng + 3) / 4;
    float mx = 0.f;
    for (int co = 0; co < 48; ++co)
        for (int c = 0; c < nc; ++c)
            for (int k = 0; k < ntap; ++k) mx = std::max(mx, std::fabs(t.data[((size_t)co * cin_total + c0 + c) * ntap + k]));
    int sft = 0;
    if (mx > 0.f && std::isfinite(mx)) sft = std::min(40, std::max(-40, (int)std::floor(std::log2(1024.0 / mx))));
    const float sc = std::ldexp(1.0f, sft);
    *inv_scale = std::ldexp(1.0f, -sft);
    const size_t halves = (size_t)nch * 3 * 2 * 512;
    if (halves * 2 != (ks == 5 ? conv5x5h_weight_bytes() : conv3x3h_weight_bytes(cin_pad))) return {};
    std::vector<uint16_t> bank(halves, 0);
    for (int G = 0; G < ng; ++G) {
        const int j = G / 4, g = G % 4, tap = G / gpt, cg = (G % gpt) * 8;
        for (int co = 0; co < 48; ++co)
            for (int e = 0; e < 8; ++e) {
                if (cg + e >= nc) continue;
                const float w = t.data[((size_t)co * cin_total + c0 + cg + e) * ntap + tap] * sc;      // exact: a power of two
                const uint16_t hi = f16_bits(w, true);
                const uint16_t lo = f16_bits(w - f16_value(hi), false);
                const size_t frag = ((size_t)(j * 3 + co / 16) * 2) * 512 + (size_t)(g * 16 + co % 16) * 8 + e;
                bank[frag] = hi;
                bank[frag + 512] = lo;
            }
    }
    std::vector<float> out(halves / 2);
    std::memcpy(out.data(), bank.data(), halves * 2);
    return out;
}


// preprocessing_layer (3x3, cin -> 48, NO activation: networks/unet.py:742) followed by the first 48 input channels of
// EncoderConvs[0][0] (3x3, :743) is one linear map of the network input: out(p) = sum_d W2[d] y(p + d - 1), y(q) = b1 +
// sum_e W1[e] x(q + e - 1)  =>  out(p) = b + sum_u Wc[u] x(p + u - 2), Wc[u] = sum_{d + e = u} W2[d] W1[e] (5x5, 48 x cin),
// b = sum_d W2[d] b1 (+ the layer's own bias, which the pass already adds).  Composed in double.  One 16-channel 5x5 conv
// (K = 400) instead of a 16 -> 48 conv, a full-resolution 48-channel map written and read back, and a 48 -> 48 conv (K = 576).
// The one thing the composition gets wrong is the zero padding BETWEEN the layers: y is ZERO outside the image, the composed
// conv sees b1 + (partial windows) there.  Only pixels on the image border are affected; launch_pre_border_fix subtracts those
// terms (it needs W1, b1, W2 in plain layouts).
int compose_pre_enc0(rvdd_t* h) {
    const HostTensor& w1 = h->staged.at("preprocessing_layer.weight");            // [48 m][cin][3][3]
    const HostTensor& b1 = h->staged.at("preprocessing_layer.bias");
    const HostTensor& w2 = h->staged.at("EncoderConvs.0.blocks.0.0.weight");      // [48 o][96][3][3], channels 0..47 = m
    const int cin = (int)w1.shape[1];
    HostTensor wc;
    wc.shape = {48, cin, 5, 5};
    std::vector<double> acc((size_t)48 * cin * 25, 0.0);
    for (int o = 0; o < 48; ++o)
        for (int m = 0; m < 48; ++m)
            for (int d = 0; d < 9; ++d) {
                const double a = w2.data[((size_t)o * 96 + m) * 9 + d];
                if (a == 0.0) continue;
                const int dy = d / 3, dx = d % 3;
                for (int c = 0; c < cin; ++c)
                    for (int e = 0; e < 9; ++e)
                        acc[((size_t)o * cin + c) * 25 + (dy + e / 3) * 5 + dx + e % 3] += a * (double)w1.data[((size_t)m * cin + c) * 9 + e];
            }
    wc.data.resize(acc.size());
    for (size_t i = 0; i < acc.size(); ++i) wc.data[i] = (float)acc[i];
    std::vector<float> bc(48);
    for (int o = 0; o < 48; ++o) {
        double b = 0.0;
        for (int m = 0; m < 48; ++m)
            for (int d = 0; d < 9; ++d) b += (double)w2.data[((size_t)o * 96 + m) * 9 + d] * (double)b1.data[m];
        bc[o] = (float)(b + (double)h->staged.at("EncoderConvs.0.blocks.0.0.bias").data[o]);      // + the layer's own bias (pass 1 adds it)
    }
    RC(upload(h, &h->pre5_w, arrange_conv3x3h(wc, 0, &h->pre5_inv, 16, 5)));
    RC(upload(h, &h->pre5_b, bc));
    std::vector<float> a1((size_t)9 * 16 * 48, 0.f), a2((size_t)9 * 48 * 48);
    for (int e = 0; e < 9; ++e)
        for (int c = 0; c < cin; ++c)
            for (int m = 0; m < 48; ++m) a1[((size_t)e * 16 + c) * 48 + m] = w1.data[((size_t)m * cin + c) * 9 + e];
    for (int d = 0; d < 9; ++d)
        for (int m = 0; m < 48; ++m)
            for (int o = 0; o < 48; ++o) a2[((size_t)d * 48 + m) * 48 + o] = w2.data[((size_t)o * 96 + m) * 9 + d];
    RC(upload(h, &h->pre_w1, a1));
    RC(upload(h, &h->pre_b1, b1.data));
    RC(upload(h, &h->pre_w2, a2));
    return RVDD_OK;
}

// ---- expected state_dict -------------------
struct KeySpec {
    std::string key;
    std::vector<int64_t> shape;
};

std::vector<std::string> convunet_conv_names(bool feat) {
    std::vector<std::string> n;
    for (int i = feat ? 0 : 1; i < CU_COUNT; ++i) n.push_back(kCuNames[i]);
    return n;
}

int convunet_cin(const rvdd_t* h, const std::string& name) {
    const bool feat = h->has_feat();
    if (name == "preprocessing_layer") return h->cin_real();
    if (name == "EncoderConvs.0.blocks.0.0") return feat ? 96 : h->cin_real();
    if (name.rfind("DecoderConvs.", 0) == 0 && name.find(".blocks.0.0") != std::string::npos) return 96;
    return 48;
}

std::vector<std::string> next_block_names(bool feat) {
    std::vector<std::string> n;
    for (int i = feat ? 0 : 1; i < NX_COUNT; ++i) n.push_back(kNxNames[i]);
    return n;
}

int next_proj_cin(const rvdd_t* h, const std::string& blk) {
    const bool feat = h->has_feat();
    if (blk == "preprocessing_layer.blocks.0") return h->cin_real();
    if (blk == "encoder_convs.0.blocks.0") return feat ? 96 : h->cin_real();
    if (blk.rfind("decoder_convs.", 0) == 0 && blk.find(".blocks.0") != std::string::npos) return 96;
    return 0;
}

std::vector<KeySpec> expected_keys(const rvdd_t* h) {
    std::vector<KeySpec> k;
    if (!h->is_next()) {
        for (const auto& n : convunet_conv_names(h->has_feat())) {
            k.push_back({n + ".weight", {48, convunet_cin(h, n), 3, 3}});
            k.push_back({n + ".bias", {48}});
        }
        k.push_back({"PostConvs.1.weight", {3, 48, 1, 1}});
        k.push_back({"PostConvs.1.bias", {3}});
    } else {
        for (const auto& b : next_block_names(h->has_feat())) {
            const int pc = next_proj_cin(h, b);
            if (pc) {
                k.push_back({b + ".proj.weight", {48, pc, 1, 1}});
                k.push_back({b + ".proj.bias", {48}});
            }
            k.push_back({b + ".block.0.weight", {48, 1, 7, 7}});
            k.push_back({b + ".block.0.bias", {48}});
            k.push_back({b + ".block.1.weight", {48}});
            k.push_back({b + ".block.1.bias", {48}});
            k.push_back({b + ".block.2.weight", {192, 48, 1, 1}});
            k.push_back({b + ".block.2.bias", {192}});
            k.push_back({b + ".block.4.weight", {48, 192, 1, 1}});
            k.push_back({b + ".block.4.bias", {48}});
            k.push_back({b + ".layerscale.layerscale", {48}});
        }
        k.push_back({"postprocessing.1.weight", {3, 48, 1, 1}});
        k.push_back({"postprocessing.1.bias", {3}});
    }
    return k;
}

// ---- measurement -------------------------------------------------------------
int prof_class(rvdd_t* h, const char* name) {
    for (size_t i = 0; i < h->prof.size(); ++i)
        if (h->prof[i].name == name) return (int)i;
    h->prof.push_back(ProfClass{name});
    return (int)h->prof.size() - 1;
}

hipEvent_t get_event(rvdd_t* h) {
    if (!h->event_pool.empty()) {
        hipEvent_t e = h->event_pool.back();
        h->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

struct Scope {   // brackets one launch with events when profiling is on
    rvdd_t* h;
    hipStream_t s;
    int cls = -1;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    Scope(rvdd_t* h_, hipStream_t s_, const char* name, double flops, double bytes) : h(h_), s(s_) {
        if (!h->prof_on) return;
        if (!h->prof_filter.empty() && h->prof_filter != name) return;
        const int c = prof_class(h, name);
        if (h->prof[c].seen++ % h->prof_stride) return;
        cls = c;
        h->prof[cls].flops += flops;
        h->prof[cls].bytes += bytes;
        h->prof[cls].launches += 1;
        e0 = get_event(h);
        e1 = get_event(h);
        (void)hipEventRecord(e0, s);
    }
    ~Scope() {
        if (cls < 0) return;
        (void)hipEventRecord(e1, s);
        h->pending.push_back({cls, e0, e1});
    }
};

int prof_flush(rvdd_t* h) {
    for (auto& p : h->pending) {
        HIPCHK(h, hipEventSynchronize(p.e1));
        float ms = 0.f;
        HIPCHK(h, hipEventElapsedTime(&ms, p.e0, p.e1));
        h->prof[p.cls].ms += ms;
        h->event_pool.push_back(p.e0);
        h->event_pool.push_back(p.e1);
    }
    h->pending.clear();
    return RVDD_OK;
}

// ---- conv helper ---------------------------------------------------------------
const char* conv_name(int cin, int epi, bool acc) {
    static const char* names[2][4][2] = {
        {{"conv3x3_kernel<16, 0, false>", "conv3x3_kernel<16, 0, true>"},
         {"conv3x3_kernel<16, 1, false>", "conv3x3_kernel<16, 1, true>"},
         {"conv3x3_kernel<16, 2, false>", "conv3x3_kernel<16, 2, true>"},
         {"conv3x3_kernel<16, 3, false>", "conv3x3_kernel<16, 3, true>"}},
        {{"conv3x3_kernel<48, 0, false>", "conv3x3_kernel<48, 0, true>"},
         {"conv3x3_kernel<48, 1, false>", "conv3x3_kernel<48, 1, true>"},
         {"conv3x3_kernel<48, 2, false>", "conv3x3_kernel<48, 2, true>"},
         {"conv3x3_kernel<48, 3, false>", "conv3x3_kernel<48, 3, true>"}}};
    return names[cin == 48][epi][acc];
}

const char* wino_name(int epi, bool acc) {
    static const char* names[5][2] = {{"wino3x3_kernel<0, false>", "wino3x3_kernel<0, true>"},
                                      {"wino3x3_kernel<1, false>", "wino3x3_kernel<1, true>"},
                                      {"wino3x3_kernel<2, false>", "wino3x3_kernel<2, true>"},
                                      {"wino3x3_kernel<3, false>", "wino3x3_kernel<3, true>"},
                                      {"wino3x3_kernel<4, false>", "wino3x3_kernel<4, true>"}};
    return names[epi][acc];
}

const char* conv_name_h(int epi, bool acc) {
    static const char* names[5][2] = {{"conv3x3h_kernel<48, 0, false, false>", "conv3x3h_kernel<48, 0, true, false>"},
                                      {"conv3x3h_kernel<48, 1, false, false>", "conv3x3h_kernel<48, 1, true, false>"},
                                      {"conv3x3h_kernel<48, 2, false, false>", "conv3x3h_kernel<48, 2, true, false>"},
                                      {"conv3x3h_kernel<48, 3, false, false>", "conv3x3h_kernel<48, 3, true, false>"},
                                      {"conv3x3h_kernel<48, 4, false, false>", "conv3x3h_kernel<48, 4, true, false>"}};
    return names[epi][acc];
}


// the amax words (rvdd_internal.h) of map `slot`, from sequence b0 on
unsigned* amax_words(const rvdd_t* h, int slot, size_t b0 = 0) { return h->amax + ((size_t)slot * h->cfg.batch + b0) * kAmaxSeqWords; }
constexpr size_t amax_bytes(int B, int nslots) { return (size_t)nslots * B * kAmaxSeqWords * sizeof(unsigned); }
int amax_layer(const rvdd_t* h, int layer) { return h->amax_base + layer; }

struct ConvCall {
    const float* in = nullptr;
    int src = 0;             // which weight slice of the layer
    const float* acc_in = nullptr;
    bool with_bias = true;   // informational: bias is used iff acc_in == nullptr
    int epi = EPI_RELU;
    const float* res1 = nullptr;
    const float* res2 = nullptr;
    float* out = nullptr;
    int H = 0, W = 0;        // conv domain
    int Hout = 0, Wout = 0, oy = 0, ox = 0;
    float* out3_nchw = nullptr;    // EPI_RELU_OUT3 targets
    float* out3_nhwc4 = nullptr;
    bool ups = false;        // `in` is the half-resolution map whose bilinear x2 upsample the conv reads (UpConv)
    int amax_in = -1;        // amax slot of `in` (-1: no scaling) and of `out` (-1: no split kernel reads it)
    int amax_out = -1;
};

bool wino_applies(const rvdd_t* h, int H, int W) {
    return h->use_wino && (h->force_wino || h->cfg.batch * ((W + 31) / 32) * ((H + 7) / 8) >= 200);
}

// Sequences [b0, b0 + nb) of the batch: the maps of a ConvCall are those of the WHOLE batch, a launch may cover a part.
struct Sub {
    int b0, nb;
};

// Full-resolution stages one sequence at a time (depth first) instead of all B sequences per layer -- an option
// (rvdd_set_option "seq_major"), off by default.  The idea: a 48-channel map of ONE 720p sequence (177 MB) stays in
// the 256 MiB Infinity Cache between the layer that writes it and the layer that reads it, the maps of four (708 MB)
// do not.  Measured (profiles/r02_c_seq_major.json): 424 frames/s against 456 batched -- per-sequence launches lose
// more to their tails (3600 units on 256 CUs = 14.06 rounds) and to four filter-bank loads per layer than the cache
// gives back.  Kept because it is free and pins an invariant the tests use: a launch's batch size does not enter a
// tile's sums, so both schedules give bit-identical frames.
bool seq_major_on(const rvdd_t* h) { return h->cfg.batch > 1 && h->seq_major == 1; }

int run_conv(rvdd_t* h, const Conv3& L, const ConvCall& c, hipStream_t s, Sub sub = Sub{0, -1}) {
    if (sub.nb < 0) sub.nb = h->cfg.batch;
    const int cin_in = L.cin_pad[c.src];
    const int Ho = c.epi == EPI_POOL ? c.H / 2 : (c.Hout ? c.Hout : c.H), Wo = c.epi == EPI_POOL ? c.W / 2 : (c.Wout ? c.Wout : c.W);
    const size_t px_in = (size_t)sub.b0 * c.H * c.W, px_out = (size_t)sub.b0 * Ho * Wo;
    ConvArgs a{};
    a.in = c.in + (c.ups ? px_in / 4 : px_in) * cin_in;
    a.ups = c.ups ? 1 : 0;
    a.w = L.w[c.src];
    a.bias = L.bias;
    a.acc_in = c.acc_in ? c.acc_in + px_in * kF : nullptr;
    a.res1 = c.res1 ? c.res1 + px_in * kF : nullptr;
    a.res2 = c.res2 ? c.res2 + px_in * kF : nullptr;
    a.out = c.out + px_out * kF;
    a.B = sub.nb;
    a.H = c.H;
    a.W = c.W;
    if (c.epi == EPI_POOL) {
        a.Hout = c.H / 2;
        a.Wout = c.W / 2;
    } else {
        a.Hout = c.Hout ? c.Hout : c.H;
        a.Wout = c.Wout ? c.Wout : c.W;
    }
    a.oy = c.oy;
    a.ox = c.ox;
    a.tiles_x = (c.W + 15) / 16;
    a.tiles_y = (c.H + 7) / 8;
    a.ntiles = a.B * a.tiles_x * a.tiles_y;
    const int cin = L.cin_pad[c.src];
    const double px = (double)a.B * c.H * c.W;
    const double flops = 2.0 * 9.0 * L.cin_real[c.src] * 48.0 * px;
    double bytes = px * 4.0 * (L.cin_real[c.src] + (c.epi == EPI_POOL ? 12.0 : 48.0));
    if (c.acc_in) bytes += px * 192.0;
    if (c.epi == EPI_RELU_ADD2) bytes += px * 384.0;
    // Winograd needs enough 8x32-pixel units to fill the chip (its 144 KiB filter bank is loaded once
    // per workgroup); the 1/8-resolution level of a single 720p sequence (60 units) runs faster direct
    a.w3 = h->w_out;
    a.b3 = h->b_out;
    a.out3_nchw = c.out3_nchw ? c.out3_nchw + px_in * 3 : nullptr;
    a.out3_nhwc4 = c.out3_nhwc4 ? c.out3_nhwc4 + px_in * 4 : nullptr;
    a.amax_in = (c.amax_in >= 0 && h->bfp) ? amax_words(h, c.amax_in, sub.b0) : nullptr;
    a.amax_out = (c.amax_out >= 0 && h->bfp) ? amax_words(h, c.amax_out, sub.b0) : nullptr;
    const bool c16_ok = cin != 48 && !c.acc_in && (c.epi == EPI_NONE || c.epi == EPI_RELU);
    if (c.ups && !(cin == 48 && ((h->split16 && L.wh[c.src]) || (L.wu[c.src] && wino_applies(h, c.H, c.W)))))
        return fail(h, RVDD_ERR_STATE, "run_conv: the fused upsample exists in the split-f16 and the Winograd kernels only");
    if (c.ups) bytes -= px * 4.0 * 36.0;          // reads the quarter-size map
    // the F16 matrix pipe with split operands: every layer of the convunet
    if (h->split16 && L.wh[c.src] && (cin == 48 || c16_ok)) {
        a.w = L.wh[c.src];
        a.wscale = L.wh_inv[c.src];
        Scope sc(h, s, c.ups ? "conv3x3h_kernel<48, 1, false, true>" : cin == 48 ? conv_name_h(c.epi, c.acc_in != nullptr)
                           : (c.epi == EPI_NONE ? "conv3x3h_kernel<16, 0, false, false>" : "conv3x3h_kernel<16, 1, false, false>"), flops, bytes);
        HIPCHK(h, launch_conv3x3h(a, cin == 48 ? 48 : 16, c.epi, s));
        return RVDD_OK;
    }
    if ((cin == 48 || c16_ok) && L.wu[c.src] && wino_applies(h, c.H, c.W)) {
        a.w = L.wu[c.src];
        Scope sc(h, s, c.ups ? "wino3x3_ups_kernel<1>" : cin == 48 ? wino_name(c.epi, c.acc_in != nullptr)
                                 : (c.epi == EPI_NONE ? "wino3x3_c16_kernel<0>" : "wino3x3_c16_kernel<1>"), flops, bytes);
        HIPCHK(h, launch_wino3x3(a, cin == 48 ? 48 : 16, c.epi, s));
        return RVDD_OK;
    }
    Scope sc(h, s, conv_name(cin, c.epi, c.acc_in != nullptr), flops, bytes);
    HIPCHK(h, launch_conv3x3(a, cin, c.epi, s));
    return RVDD_OK;
}


// The composed 5x5 conv of the network input (compose_pre_enc0) into `part`, and its border fix; sequences of `sub`.
int run_pre5(rvdd_t* h, const float* netin, float* part, hipStream_t s, Sub sub) {
    if (sub.nb < 0) sub.nb = h->cfg.batch;
    const int H = h->cfg.height, W = h->cfg.width;
    const size_t px0 = (size_t)sub.b0 * H * W;
    ConvArgs a{};
    a.in = netin + px0 * kNetInC;
    a.w = h->pre5_w;
    a.wscale = h->pre5_inv;
    a.bias = h->pre5_b;
    a.out = part + px0 * kF;
    a.B = sub.nb;
    a.H = a.Hout = H;
    a.W = a.Wout = W;
    a.amax_in = h->bfp ? amax_words(h, h->amax_base + AMAX_REL_NETIN, sub.b0) : nullptr;
    const double px = (double)sub.nb * H * W;
    {
        Scope sc(h, s, "conv5x5h_kernel<16>", 2.0 * 25.0 * h->cin_real() * 48.0 * px, px * 4.0 * (h->cin_real() + 48.0));
        HIPCHK(h, launch_conv5x5h_c16(a, s));
    }
    HIPCHK(h, launch_pre_border_fix(a.in, h->pre_w1, h->pre_b1, h->pre_w2, a.out, sub.nb, H, W, s));
    return RVDD_OK;
}

// What rvdd_step does in front of the net (demosaic, warps): the caller's frame and flow pointers of one step.
// run_convunet calls it per sequence when the full-resolution stages run depth first; null for rvdd_unet_forward.
struct StepInputs {
    const float* raw_prev = nullptr;      // only on the first step of a video
    const float* raw_cur = nullptr;
    const float* raw_next = nullptr;
    const float* flow_prev = nullptr;
    const float* flow_next = nullptr;
    size_t rawf = 0, flowf = 0;       // floats from one sequence to the next in the caller's raw / flow tensors
};
int run_prologue(rvdd_t* h, const StepInputs& in, Sub sb, hipStream_t s);

// networks/unet.py:544-588 as specialised by UNet_FixedFeatures[_feat] (:595-825).
int run_convunet(rvdd_t* h, const float* netin, const float* featw, float* feat_dst, float* out_nchw,
                 float* out_nhwc4, hipStream_t s, const StepInputs* prologue) {
    const bool feat = h->has_feat();
    const int B = h->cfg.batch;
    Level* lv = h->lv;
    const Conv3* cu = h->cu;
    // `from` = the amax slot of the input map (L(the layer that wrote it), AMAX_NETIN, h->amax_feat_in); a layer's output slot is L(its id)
    auto conv = [&](int layer, const float* in, int from, float* out, int lvl, int epi, Sub sub) {
        ConvCall c;
        c.in = in; c.out = out; c.H = lv[lvl].H; c.W = lv[lvl].W; c.epi = epi;
        c.amax_in = from; c.amax_out = amax_layer(h, layer);
        return run_conv(h, cu[layer], c, s, sub);
    };
    const auto L = [&](int layer) { return amax_layer(h, layer); };
    const int AMAX_NETIN = h->amax_base + AMAX_REL_NETIN;
    // two-source (virtual concat) conv: pass 1 leaves bias + sum over source A in `part`
    auto conv2 = [&](int layer, const float* inA, int fromA, const float* inB, int fromB, float* out, int lvl, Sub sub) {
        ConvCall c;
        c.in = inA; c.src = 0; c.out = lv[lvl].part; c.H = lv[lvl].H; c.W = lv[lvl].W; c.epi = EPI_NONE;
        c.amax_in = fromA;
        RC(run_conv(h, cu[layer], c, s, sub));
        c.in = inB; c.src = 1; c.acc_in = lv[lvl].part; c.out = out; c.epi = EPI_RELU;
        c.amax_in = fromB; c.amax_out = amax_layer(h, layer);
        return run_conv(h, cu[layer], c, s, sub);
    };
    const Sub all{0, B};
    // the full-resolution stages run per sequence when that keeps their maps in the Infinity Cache (seq_major_on),
    // in an order that alternates from frame to frame so that a step begins with the sequence the last one ended on
    const bool per_seq = seq_major_on(h);
    const int nsub = per_seq ? B : 1;
    auto sub_at = [&](int k) { return per_seq ? Sub{h->serpentine ? B - 1 - k : k, 1} : all; };

    // ---- pre-stages + encoder level 0
    for (int k = 0; k < nsub; ++k) {
        const Sub sb = sub_at(k);
        if (prologue) RC(run_prologue(h, *prologue, sb, s));
        if (feat && h->fuse_pre && h->split16 && h->pre5_w) {
            // preprocessing_layer (:742, no activation) and the first source of EncoderConvs[0][0] (:743) as ONE 5x5 conv of the
            // network input (compose_pre_enc0), its border ring put right, then the second source (the old features) as before
            RC(run_pre5(h, netin, lv[0].part, s, sb));
            ConvCall c;
            c.in = featw; c.src = 1; c.acc_in = lv[0].part; c.out = lv[0].t[1]; c.H = lv[0].H; c.W = lv[0].W; c.epi = EPI_RELU;
            c.amax_in = h->amax_feat_in; c.amax_out = L(CU_ENC0_0);
            RC(run_conv(h, cu[CU_ENC0_0], c, s, sb));
        } else if (feat) {
            RC(conv(CU_PRE, netin, AMAX_NETIN, lv[0].t[0], 0, EPI_NONE, sb));               // :742 (no activation)
            RC(conv2(CU_ENC0_0, lv[0].t[0], L(CU_PRE), featw, h->amax_feat_in, lv[0].t[1], 0, sb)); // cat[y, old_features] :743
        } else {
            RC(conv(CU_ENC0_0, netin, AMAX_NETIN, lv[0].t[1], 0, EPI_RELU, sb));
        }
        RC(conv(CU_ENC0_1, lv[0].t[1], L(CU_ENC0_0), lv[0].skip, 0, EPI_RELU, sb));
        RC(conv(CU_DOWN0, lv[0].skip, L(CU_ENC0_1), lv[1].t[0], 0, EPI_POOL, sb));          // :207-208
    }
    // ---- encoder levels 1..3 (all sequences per launch: these levels need the batch to fill the chip)
    for (int i = 1; i <= 3; ++i) {
        if (i > 1) RC(conv(cu_down(i - 1), lv[i - 1].skip, L(cu_enc(i - 1, 1)), lv[i].t[0], i - 1, EPI_POOL, all));
        RC(conv(cu_enc(i, 0), lv[i].t[0], L(cu_down(i - 1)), lv[i].t[1], i, EPI_RELU, all));
        RC(conv(cu_enc(i, 1), lv[i].t[1], L(cu_enc(i, 0)), i < 3 ? lv[i].skip : lv[3].t[2], i, EPI_RELU, all));
    }
    // ---- bottleneck: d = e3 + d1 + d2 (:561-567)
    float* e3 = lv[3].t[2];
    RC(conv(CU_BOT0, e3, L(cu_enc(3, 1)), lv[3].t[0], 3, EPI_RELU, all));
    {
        ConvCall c;
        c.in = lv[3].t[0]; c.out = lv[3].t[1]; c.H = lv[3].H; c.W = lv[3].W;
        c.epi = EPI_RELU_ADD2; c.res1 = e3; c.res2 = lv[3].t[0];
        c.amax_in = L(CU_BOT0); c.amax_out = L(CU_BOT1);
        RC(run_conv(h, cu[CU_BOT1], c, s));
    }
    const float* d = lv[3].t[1];
    int d_from = L(CU_BOT1);
    // ---- decoder (:570-579); its last level again per sequence, together with the post convs
    float* fdst = feat_dst ? feat_dst : lv[0].t[2];
    for (int i = 0; i < 3; ++i) {
        const int lo = 3 - i, hi = 2 - i;
        const int uh = 2 * lv[lo].H, uw = 2 * lv[lo].W;      // size after nn.Upsample(x2)
        for (int k = 0; k < (hi == 0 ? nsub : 1); ++k) {
            const Sub sb = hi == 0 ? sub_at(k) : all;
            const size_t lo_px = (size_t)sb.b0 * lv[lo].H * lv[lo].W, hi_px = (size_t)sb.b0 * lv[hi].H * lv[hi].W;
            // UpConv: bilinear x2, conv, ReLU (:137-142).  Where the Winograd kernel runs the conv, the interpolation
            // happens in its patch load and the upsampled map is never written; elsewhere it is made first.
            const bool fused = h->fuse_upsample && (h->split16 || wino_applies(h, uh, uw));
            if (!fused) {
                Scope sc(h, s, "upsample2x_kernel", 0.0, (double)sb.nb * uh * uw * 192.0 * 1.25);
                HIPCHK(h, launch_upsample2x(d + lo_px * kF, lv[hi].t[0] + (size_t)sb.b0 * uh * uw * kF, sb.nb, lv[lo].H, lv[lo].W, uh,
                                            uw, 0, 0, false, s));
            }
            // conv + ReLU at the upsampled size, written into a map of the skip's size
            // (zero_pad_features, :151-170; identity when sizes agree)
            ConvCall c;
            c.in = fused ? d : lv[hi].t[0]; c.ups = fused;
            c.out = lv[hi].t[1]; c.H = uh; c.W = uw; c.epi = EPI_RELU;
            c.Hout = lv[hi].H; c.Wout = lv[hi].W;
            c.oy = (lv[hi].H - uh) / 2; c.ox = (lv[hi].W - uw) / 2;
            // (an interpolated value never exceeds the map's maximum: the upsampled map shares the words of its source)
            c.amax_in = d_from; c.amax_out = L(cu_up(i));
            if (uh != lv[hi].H || uw != lv[hi].W)
                HIPCHK(h, hipMemsetAsync(lv[hi].t[1] + hi_px * kF, 0, (size_t)sb.nb * lv[hi].H * lv[hi].W * kF * sizeof(float), s));
            RC(run_conv(h, cu[cu_up(i)], c, s, sb));
            RC(conv2(cu_dec(i, 0), lv[hi].skip, L(cu_enc(hi, 1)), lv[hi].t[1], L(cu_up(i)), lv[hi].t[0], hi, sb));        // cat(skip, dec) :541
            RC(conv(cu_dec(i, 1), lv[hi].t[0], L(cu_dec(i, 0)), lv[hi].t[1], hi, EPI_RELU, sb));
            if (hi > 0) continue;
            // ---- post: hooked 48-ch map = next frame's features (:808-812), then 1x1 -> 3
            if (h->split16 || wino_applies(h, lv[0].H, lv[0].W)) {
                // PostConvs[1] (1x1, 48 -> 3) rides in the epilogue of PostConvs[0]'s kernel (split-f16: at every size; Winograd
                // f32: where it runs) -- and with it the output frame's share of the words the next step's input bound reads
                ConvCall pc;
                pc.in = lv[0].t[1]; pc.out = fdst; pc.H = lv[0].H; pc.W = lv[0].W; pc.epi = EPI_RELU_OUT3;
                pc.out3_nchw = out_nchw; pc.out3_nhwc4 = out_nhwc4;
                pc.amax_in = L(cu_dec(2, 1)); pc.amax_out = h->amax_post_out;
                RC(run_conv(h, cu[CU_POST], pc, s, sb));
            } else {
                ConvCall pc;
                pc.in = lv[0].t[1]; pc.out = fdst; pc.H = lv[0].H; pc.W = lv[0].W; pc.epi = EPI_RELU;
                pc.amax_in = L(cu_dec(2, 1)); pc.amax_out = h->amax_post_out;
                RC(run_conv(h, cu[CU_POST], pc, s, sb));
                const size_t px0 = (size_t)sb.b0 * h->cfg.height * h->cfg.width;
                const double px = (double)sb.nb * h->cfg.height * h->cfg.width;
                Scope sc(h, s, "conv1x1_out_kernel", 2.0 * 48 * 3 * px, px * (192.0 + 12.0 + 16.0));
                HIPCHK(h, launch_conv1x1_out(fdst + px0 * kF, h->w_out, h->b_out, out_nchw + px0 * 3,
                                             out_nhwc4 ? out_nhwc4 + px0 * 4 : nullptr, sb.nb, h->cfg.height, h->cfg.width, s));
            }
        }
        d = lv[hi].t[1];
        d_from = L(cu_dec(i, 1));
    }
    if (per_seq) h->serpentine = !h->serpentine;
    return RVDD_OK;
}

int run_net(rvdd_t* h, const float* netin, const float* featw, float* feat_dst, float* out_nchw,
            float* out_nhwc4, hipStream_t s, const StepInputs* prologue);

int ensure_scratch(rvdd_t* h, size_t bytes) {
    if (h->scratch_bytes >= bytes) return RVDD_OK;
    if (h->scratch) {
        HIPCHK(h, hipDeviceSynchronize());
        HIPCHK(h, hipFree(h->scratch));
        h->scratch = nullptr;
        h->scratch_bytes = 0;
    }
    HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&h->scratch), bytes));
    h->scratch_bytes = bytes;
    return RVDD_OK;
}

}  // namespace

#include "runtime_next.inc"

namespace {
int run_net(rvdd_t* h, const float* netin, const float* featw, float* feat_dst, float* out_nchw,
            float* out_nhwc4, hipStream_t s, const StepInputs* prologue) {
    if (!h->is_next()) {
        return run_convunet(h, netin, featw, feat_dst, out_nchw, out_nhwc4, s, prologue);
    }
    const int B = h->cfg.batch;
    h->featw_proj = false;          // a caller's own features (rvdd_unet_forward) come as they are
    h->netin_proj = false;          // and so does a caller's own network input
    if (prologue) RC(run_prologue(h, *prologue, Sub{0, B}, s));
    return run_convnext(h, netin, featw, feat_dst, out_nchw, out_nhwc4, s, Sub{0, B});
}
}  // namespace

// =============================================================== C ABI =====
extern "C" {

const char* rvdd_version(void) { return "rvdd-hip 0.1 (gfx950)"; }

const char* rvdd_last_error(const rvdd_t* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

namespace {
bool graph_stream(rvdd_t* h) {
    if (h->gstream) return true;
    if (hipStreamCreateWithFlags(&h->gstream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->g_in, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->g_out, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        if (h->gstream) (void)hipStreamDestroy(h->gstream);
        h->gstream = nullptr;
        return false;
    }
    return true;
}
}  // namespace

int rvdd_create(const rvdd_cfg* cfg, rvdd_t** out) {
    if (!cfg || !out) return fail(nullptr, RVDD_ERR_ARG, "rvdd_create: null argument");
    *out = nullptr;
    if (cfg->arch < 0 || cfg->arch > 3) return fail(nullptr, RVDD_ERR_ARG, "rvdd_create: unknown arch %d", cfg->arch);
    if (cfg->future < 0 || cfg->future > 1) return fail(nullptr, RVDD_ERR_ARG, "rvdd_create: future must be 0 or 1");
    if (cfg->batch < 1) return fail(nullptr, RVDD_ERR_ARG, "rvdd_create: batch must be >= 1");
    if (cfg->height < 16 || cfg->width < 16 || (cfg->height & 1) || (cfg->width & 1))
        return fail(nullptr, RVDD_ERR_ARG, "rvdd_create: frame size %dx%d must be even and >= 16", cfg->height, cfg->width);
    // byte offsets inside one 48-channel map are 32-bit in every kernel (buffer addressing): the map must stay below 2 GiB
    if ((size_t)cfg->height * cfg->width * kF * sizeof(float) >= 0x80000000ull)
        return fail(nullptr, RVDD_ERR_ARG, "rvdd_create: frame %dx%d too large: one 48-channel map must stay below 2 GiB (11.1 Mpx)",
                    cfg->height, cfg->width);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, RVDD_ERR_HIP, "rvdd_create: no HIP device available (%s); this runtime has no CPU fallback",
                    hipGetErrorString(e));
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(nullptr, RVDD_ERR_ARG, "rvdd_create: device %d out of range (0..%d)", cfg->device, ndev - 1);
    DeviceGuard guard(cfg->device);
    if (guard.err != hipSuccess) return fail(nullptr, RVDD_ERR_HIP, "cannot select device %d: %s", cfg->device, hipGetErrorString(guard.err));

    rvdd_t* h = new rvdd_handle();
    h->cfg = *cfg;
    if (const char* sm = std::getenv("RVDD_SEQ_MAJOR")) h->seq_major = std::atoi(sm) != 0;     // measurement switches
    if (const char* fu = std::getenv("RVDD_FUSE_UPSAMPLE")) h->fuse_upsample = std::atoi(fu) != 0;
    if (const char* bf = std::getenv("RVDD_BFP")) h->bfp = std::atoi(bf) != 0;
    if (const char* fp = std::getenv("RVDD_FUSE_PRE")) h->fuse_pre = std::atoi(fp) != 0;
    if (const char* np = std::getenv("RVDD_NEXT_POOL")) h->next_pool = std::atoi(np) != 0;
    if (const char* nsp = std::getenv("RVDD_NEXT_SPLIT")) h->next_split = std::atoi(nsp) != 0;
    if (const char* npp = std::getenv("RVDD_NEXT_PIPE")) h->next_pipe = std::atoi(npp) != 0;
    if (const char* npf = std::getenv("RVDD_NEXT_PROJFUSE")) h->next_projfuse = std::atoi(npf) != 0;
    if (const char* cv = std::getenv("RVDD_CONV")) {
        // f32 (the f32-MFMA kernels, direct or Winograd by launch size) | direct | winograd (that f32 kernel at every size) |
        // anything else = the default: split-f16 kernel for the 48-channel layers, f32 kernels by size for the rest
        h->use_wino = std::strcmp(cv, "direct") != 0;
        h->force_wino = std::strcmp(cv, "winograd") == 0;
        h->split16 = std::strcmp(cv, "direct") != 0 && std::strcmp(cv, "winograd") != 0 && std::strcmp(cv, "f32") != 0;
    }
    const int B = cfg->batch, H = cfg->height, W = cfg->width;
    int rc = RVDD_OK;
    auto A = [&](float** p, size_t floats) {
        if (rc == RVDD_OK) rc = dmalloc(h, reinterpret_cast<void**>(p), floats * sizeof(float));
    };
    int lh = H, lw = W;
    for (int l = 0; l < 4; ++l) {
        h->lv[l].H = lh;
        h->lv[l].W = lw;
        const size_t n = (size_t)B * lh * lw * kF;
        for (int k = 0; k < 3; ++k) A(&h->lv[l].t[k], n);
        A(&h->lv[l].skip, n);
        A(&h->lv[l].part, n);
        lh /= 2;
        lw /= 2;
    }
    const size_t npix = (size_t)B * H * W;
    A(&h->netin, npix * kNetInC);
    A(&h->lastden4, npix * 4);
    A(&h->next4, npix * 4);
    A(&h->green, npix);
    if (h->has_feat()) {
        A(&h->featw, npix * kF);
        A(&h->lastfeat, npix * kF);
    }
    if (rc == RVDD_OK) rc = dmalloc(h, reinterpret_cast<void**>(&h->amax), amax_bytes(B, AMAX_SLOTS));
    if (rc == RVDD_OK) rc = dmalloc(h, reinterpret_cast<void**>(&h->loss_partial), 2 * 1024 * sizeof(double));
    if (rc == RVDD_OK) rc = dmalloc(h, reinterpret_cast<void**>(&h->loss_result), 2 * sizeof(double));
    if (rc != RVDD_OK) {
        g_create_error = h->err;
        rvdd_destroy(h);
        return rc;
    }
    (void)hipEventCreate(&h->t0);
    (void)hipEventCreate(&h->t1);
    // The capture stream (option graphs) is created when the option first asks for it, not here: every stream a process holds is a
    // hardware queue the device's scheduler keeps mapped, and a handle that merely EXISTED beside another one -- with the two idle
    // streams every handle used to create -- made that one's cooperative TV-L1 launches and the kernels behind them 20 % slower
    // (profiles/r05k_online_flow_two_handles.txt)
    if (const char* cs = std::getenv("RVDD_COUT_SPLIT")) conv3x3h_set_cout_split(std::atoi(cs) != 0);      // process-wide A/B switch
    if (const char* sp = std::getenv("RVDD_SMALL_PRESTAGE")) prestage_set_small(std::atoi(sp) != 0);       // likewise
    if (const char* gv = std::getenv("RVDD_GRAPH")) h->use_graphs = std::atoi(gv) != 0 && graph_stream(h);
    *out = h;
    return RVDD_OK;
}

void rvdd_destroy(rvdd_t* h) {
    if (!h) return;
    DeviceGuard guard(h->cfg.device);
    (void)hipDeviceSynchronize();
    drop_graphs(h);
    if (h->g_in) (void)hipEventDestroy(h->g_in);
    if (h->g_out) (void)hipEventDestroy(h->g_out);
    if (h->gstream) (void)hipStreamDestroy(h->gstream);
    for (void* p : h->allocs) (void)hipFree(p);
    if (h->scratch) (void)hipFree(h->scratch);
    tvl1_free(h->tvl1);
    for (auto& p : h->pending) {
        (void)hipEventDestroy(p.e0);
        (void)hipEventDestroy(p.e1);
    }
    for (auto e : h->event_pool) (void)hipEventDestroy(e);
    if (h->t0) (void)hipEventDestroy(h->t0);
    if (h->t1) (void)hipEventDestroy(h->t1);
    delete h;
}

int rvdd_set_weight(rvdd_t* h, const char* key, const float* host, const int64_t* shape, int32_t ndim) {
    if (!h || !key || !host || !shape || ndim < 1 || ndim > 4) return fail(h, RVDD_ERR_ARG, "rvdd_set_weight: bad argument");
    if (h->finalized) return fail(h, RVDD_ERR_STATE, "rvdd_set_weight: weights already finalized");
    const auto exp = expected_keys(h);
    const KeySpec* spec = nullptr;
    for (const auto& k : exp)
        if (k.key == key) spec = &k;
    if (!spec) return fail(h, RVDD_ERR_WEIGHT, "unexpected state_dict key '%s' for this architecture", key);
    std::vector<int64_t> shp(shape, shape + ndim);
    if (shp != spec->shape) {
        std::string got, want;
        for (auto v : shp) got += std::to_string(v) + ",";
        for (auto v : spec->shape) want += std::to_string(v) + ",";
        return fail(h, RVDD_ERR_WEIGHT, "state_dict key '%s' has shape [%s] but [%s] is expected", key, got.c_str(), want.c_str());
    }
    size_t n = 1;
    for (auto v : shp) n *= (size_t)v;
    HostTensor t;
    t.shape = shp;
    t.data.assign(host, host + n);
    h->staged[key] = std::move(t);
    return RVDD_OK;
}

int rvdd_finalize_weights(rvdd_t* h) {
    if (!h) return RVDD_ERR_ARG;
    if (h->finalized) return RVDD_OK;
    for (const auto& k : expected_keys(h))
        if (!h->staged.count(k.key)) return fail(h, RVDD_ERR_WEIGHT, "missing state_dict key '%s'", k.key.c_str());
    ENTER(h);
    if (!h->is_next()) {
        for (const auto& n : convunet_conv_names(h->has_feat())) {
            const HostTensor& wt = h->staged.at(n + ".weight");
            Conv3 L;
            const int cin = (int)wt.shape[1];
            if (cin == 96) {
                L.nsrc = 2;
                for (int sidx = 0; sidx < 2; ++sidx) {
                    L.cin_real[sidx] = 48;
                    L.cin_pad[sidx] = 48;
                    RC(upload(h, &L.w[sidx], arrange_conv3x3(wt, 48 * sidx, 48, 48)));
                    RC(upload(h, &L.wu[sidx], arrange_wino3x3(wt, 48 * sidx)));
                    RC(upload(h, &L.wh[sidx], arrange_conv3x3h(wt, 48 * sidx, &L.wh_inv[sidx])));
                }
            } else {
                L.nsrc = 1;
                L.cin_real[0] = cin;
                L.cin_pad[0] = cin == 48 ? 48 : kNetInC;
                RC(upload(h, &L.w[0], arrange_conv3x3(wt, 0, cin, L.cin_pad[0])));
                RC(upload(h, &L.wu[0], cin == 48 ? arrange_wino3x3(wt, 0) : arrange_wino3x3(wt, 0, 1)));
                RC(upload(h, &L.wh[0], arrange_conv3x3h(wt, 0, &L.wh_inv[0], cin == 48 ? 48 : 16)));
            }
            RC(upload(h, &L.bias, h->staged.at(n + ".bias").data));
            for (int li = 0; li < CU_COUNT; ++li)
                if (n == kCuNames[li]) h->cu[li] = L;
        }
        RC(upload(h, &h->w_out, h->staged.at("PostConvs.1.weight").data));
        RC(upload(h, &h->b_out, h->staged.at("PostConvs.1.bias").data));
        if (h->has_feat()) RC(compose_pre_enc0(h));
    } else {
        RC(finalize_convnext(h));
    }
    h->staged.clear();
    h->finalized = true;
    return RVDD_OK;
}

// Host-only helper of the TIFF reader (rvdd-release_amd/tiffio.py): TIFF 6.0 section 13 LZW (MSB-first codes of
// 9..12 bits, ClearCode 256, EndOfInformation 257, "early change").  The pure-Python decoder does ~1 MB/s, a
// 1280x720 four-channel float frame is 15 MB.  Returns the number of bytes produced, or -1 on a corrupt stream / a
// full output buffer.
int64_t rvdd_tiff_lzw_decode(const uint8_t* in, int64_t n, uint8_t* out, int64_t cap) {
    if (!in || !out || n < 0 || cap < 0) return -1;
    struct Entry { int32_t prev; uint16_t len; uint8_t first, last; };
    static thread_local Entry tab[4096];
    for (int i = 0; i < 256; ++i) tab[i] = Entry{-1, 1, (uint8_t)i, (uint8_t)i};
    int next = 258, nbits = 9, prev = -1;
    uint32_t buf = 0;
    int have = 0;
    int64_t pos = 0, o = 0;
    for (;;) {
        while (have < nbits) {
            if (pos >= n) return o;                    // streams without an EOI code end with the data
            buf = (buf << 8) | in[pos++];
            have += 8;
        }
        const int code = (int)((buf >> (have - nbits)) & ((1u << nbits) - 1u));
        have -= nbits;
        if (code == 257) return o;
        if (code == 256) {
            next = 258;
            nbits = 9;
            prev = -1;
            continue;
        }
        int cur;
        if (prev < 0) {
            if (code >= 256) return -1;
            cur = code;
        } else if (code < next) {
            cur = code;
            if (next < 4096) {
                tab[next] = Entry{prev, (uint16_t)(tab[prev].len + 1), tab[prev].first, tab[code].first};
                ++next;
            }
        } else if (code == next && next < 4096) {
            tab[next] = Entry{prev, (uint16_t)(tab[prev].len + 1), tab[prev].first, tab[prev].first};
            cur = next++;
        } else {
            return -1;
        }
        const int len = tab[cur].len;
        if (o + len > cap) return -1;
        for (int e = cur, k = len - 1; k >= 0; --k, e = tab[e].prev) out[o + k] = tab[e].last;
        o += len;
        prev = cur;
        if (next >= 2047) nbits = 12;
        else if (next >= 1023) nbits = 11;
        else if (next >= 511) nbits = 10;
    }
}

int rvdd_set_option(rvdd_t* h, const char* name, int32_t value) {
    if (!h || !name) return RVDD_ERR_ARG;
    {
        ENTER(h);
        (void)hipDeviceSynchronize();
        drop_graphs(h);            // a captured step has the options it was captured with
    }
    if (std::strcmp(name, "graphs") == 0) {
        h->use_graphs = value != 0 && graph_stream(h);
        return RVDD_OK;
    }
    if (std::strcmp(name, "no_warp") == 0) {
        h->no_warp = value != 0;
        return RVDD_OK;
    }
    if (std::strcmp(name, "warp_raw") == 0) {
        // the reference warps the full-resolution features with the raw-resolution flow in this mode and fails on the shapes
        if (value && h->has_feat()) return fail(h, RVDD_ERR_ARG, "rvdd_set_option: warp_raw is not defined with feature recurrence (the reference fails there too)");
        h->warp_raw = value != 0;
        return RVDD_OK;
    }
    if (std::strcmp(name, "prev_noisy_frame") == 0) {
        h->prev_noisy = value != 0;
        return RVDD_OK;
    }
    if (std::strcmp(name, "fuse_upsample") == 0) {
        // 0 = UpConv's bilinear x2 always as its own kernel (A/B reference of the fused Winograd patch load)
        h->fuse_upsample = value != 0;
        return RVDD_OK;
    }
    if (std::strcmp(name, "next_split") == 0) {
        // 0 = the fused ConvBlock's two 1x1 convs on the f32 matrix pipe (exact-f32 products: the A/B reference of the split-f16 form)
        h->next_split = value != 0;
        return RVDD_OK;
    }
    if (std::strcmp(name, "next_pipe") == 0) {
        // 0 = the fused ConvBlock's phases one after the other in all eight waves (convblock_kernel) instead of the pipeline over
        // tiles (depth-wise + LayerNorm of tile t + 1 on waves 0-3 beside the MLP of tile t on waves 4-7); same bits
        h->next_pipe = value != 0;
        return RVDD_OK;
    }
    if (std::strcmp(name, "next_pool") == 0) {
        // 0 = MaxPool2d(2) as its own kernel behind the fused block (A/B reference of the pooling epilogue; same bits)
        h->next_pool = value != 0;
        return RVDD_OK;
    }
    if (std::strcmp(name, "next_projfuse") == 0) {
        // 0 = the 96 -> 48 projection behind a concat as its own kernel (A/B reference of the projection halves in the epilogues of
        // the blocks that form the two concatenated maps)
        h->next_projfuse = value != 0;
        return RVDD_OK;
    }
    if (std::strcmp(name, "cout_split") == 0) {
        // 0 = every launch of the split-f16 conv kernel forms all 48 output channels per workgroup (A/B reference of the
        // output-channel split that launches of at most a third of a tile per CU take; same bits).  Process-wide.
        conv3x3h_set_cout_split(value != 0);
        return RVDD_OK;
    }
    if (std::strcmp(name, "small_prestage") == 0) {
        // 0 = the pre-stage of a frame-step as its three kernels (input bound, green plane, network input) at every size: the A/B
        // reference of the one-kernel form that small launches without a future frame take (same bits).  Process-wide.
        prestage_set_small(value != 0);
        return RVDD_OK;
    }
    if (std::strcmp(name, "tvl1_async") == 0) {
        // 1 = rvdd_tvl1flow_batch without iteration counts enqueues its launches and returns; its control word is read by the next
        // synchronising call.  0 switches back and reports what is pending now.
        if (!value && h->tvl1) {
            HIPCHK(h, hipDeviceSynchronize());
            if (hipError_t e = tvl1_check(h->tvl1, nullptr); e != hipSuccess)
                return fail(h, RVDD_ERR_HIP, "rvdd_set_option(tvl1_async, 0): a pending asynchronous flow batch failed: %s", hipGetErrorString(e));
        }
        h->tvl1_async = value != 0;
        return RVDD_OK;
    }
    if (std::strcmp(name, "fuse_pre") == 0) {
        // 0 = preprocessing_layer and EncoderConvs[0][0] as the two convs they are, instead of their composition (the A/B
        // reference: same map up to fp32 rounding of a different summation order)
        h->fuse_pre = value != 0;
        return RVDD_OK;
    }
    if (std::strcmp(name, "block_fp") == 0) {
        // 0 = the split-f16 convs split their operands as they are (no per-map power of two): the round-3 behaviour, right only
        // while every activation stays inside 2^-14 .. 65504 -- kept as the A/B reference of the block floating point
        h->bfp = value != 0;
        if (h->amax) {
            ENTER(h);
            HIPCHK(h, hipMemset(h->amax, 0, amax_bytes(h->cfg.batch, AMAX_SLOTS)));      // no stale words across the switch
        }
        return RVDD_OK;
    }
    if (std::strcmp(name, "seq_major") == 0) {
        // 1 = the full-resolution stages of the convunet run one sequence at a time (measured slower: see seq_major_on)
        if (value < 0 || value > 1) return fail(h, RVDD_ERR_ARG, "rvdd_set_option: seq_major must be 0 or 1");
        h->seq_major = value;
        return RVDD_OK;
    }
    if (std::strcmp(name, "conv_kernel") == 0) {
        // which kernel runs the 3x3 convs: 0 = the default (48-channel layers on the split-f16 kernel, the others on an f32
        // kernel chosen by launch size), 1 = the direct f32 kernel everywhere, 2 = the Winograd f32 kernel everywhere (also
        // where it is the slower choice: tests and A/B measurements), 4 = f32 kernels chosen by launch size
        if (value < 0 || value > 4 || value == 3) return fail(h, RVDD_ERR_ARG, "rvdd_set_option: conv_kernel must be 0 (default), 1 (direct f32), 2 (winograd f32) or 4 (f32 by size)");
        h->use_wino = value != 1;
        h->force_wino = value == 2;
        h->split16 = value == 0;
        return RVDD_OK;
    }
    return fail(h, RVDD_ERR_ARG, "rvdd_set_option: unknown option '%s' (known: no_warp, warp_raw, prev_noisy_frame, conv_kernel, seq_major, graphs, fuse_upsample, next_split, next_pipe, next_pool, next_projfuse, tvl1_async, block_fp, fuse_pre, cout_split, small_prestage)", name);
}

int rvdd_reset(rvdd_t* h) {
    if (!h) return RVDD_ERR_ARG;
    h->need_init = true;
    return RVDD_OK;
}

}  // extern "C"

namespace {

// The stages in front of the net for sequences [sb.b0, sb.b0 + sb.nb): one launch covers them all -- the kernels take
// the caller's batch strides (channel slices of the reference's wider `n` / `flow` tensors are strided over the batch).
int run_prologue(rvdd_t* h, const StepInputs& in, Sub sb, hipStream_t s) {
    const bool nw = h->no_warp;
    const int H = h->cfg.height, W = h->cfg.width;
    const size_t img = (size_t)H * W, npix = (size_t)h->cfg.batch * img;
    const size_t o = (size_t)sb.b0;
    const int n = sb.nb;
    const float* rc_ = in.raw_cur + o * in.rawf;
    const float* fp_ = in.flow_prev ? in.flow_prev + o * in.flowf : nullptr;
    const float* rn_ = in.raw_next ? in.raw_next + o * in.rawf : nullptr;
    const float* fn_ = in.flow_next ? in.flow_next + o * in.flowf : nullptr;
    float* green = h->green + o * img;
    float* netin = h->netin + o * img * kNetInC;
    // amax words of the maps the split-f16 convs read first (block floating point, rvdd_internal.h)
    const bool bfp = h->bfp && h->split16 && !h->is_next();
    unsigned* amax_netin = bfp ? amax_words(h, h->amax_base + AMAX_REL_NETIN, o) : nullptr;
    // the zeroing for the step after this one rides in the first netin_bound launch of the step; a step without one memsets
    const int t1 = h->step_ctr + 1;
    unsigned* zero_a = amax_words(h, (t1 & 1) * AMAX_NREG);
    unsigned* zero_b = amax_words(h, AMAX_FEAT0 + t1 % 3);
    const size_t zero_na = amax_bytes(h->cfg.batch, AMAX_NREG) / 4, zero_nb = amax_bytes(h->cfg.batch, 1) / 4;
    const bool zero_now = bfp && h->amax_zero_pending;
    h->amax_zero_pending = false;
    if (h->warp_raw && !nw) {
        // warp_frame with --warp_raw (models/recurrent_model.py:149-152): HA(warp(remosaick(frame), raw-resolution flow)).
        // remosaick(HA(raw)) is raw itself, so the next frame is warped as it came.  next4 is free in this mode: its
        // first quarter holds the re-mosaicked previous output, the second the warped planes.  The generic NCHW warp
        // takes dense tensors: one sequence at a time when the caller's are strided (a mode without checkpoints of its own).
        const bool dense = in.rawf == (size_t)4 * (H / 2) * (W / 2) && in.flowf == (size_t)2 * (H / 2) * (W / 2);
        {
            Scope sc(h, s, "demosaic(ha_green+ha_rb)", 0.0, (double)n * img * 16.0);
            HIPCHK(h, launch_demosaic(rc_, green, netin + 3, n, H / 2, W / 2, (int64_t)H * W * kNetInC, kNetInC, 1, s, (int64_t)in.rawf));
        }
        for (int b = 0; b < n; b += dense ? n : 1) {
            const int nb = dense ? n : 1;
            float* packed = h->next4 + (o + b) * img;
            float* warped = h->next4 + npix + (o + b) * img;
            HIPCHK(h, launch_remosaick4(h->lastden4 + (o + b) * img * 4, packed, nb, H, W, s));
            HIPCHK(h, launch_warp_nchw(packed, fp_ + b * in.flowf, warped, nb, 4, H / 2, W / 2, s));
            HIPCHK(h, launch_demosaic(warped, green + b * img, netin + b * img * kNetInC + 0, nb, H / 2, W / 2, (int64_t)H * W * kNetInC,
                                      kNetInC, 1, s));
            if (h->cfg.future) {
                HIPCHK(h, launch_warp_nchw(rn_ + b * in.rawf, fn_ + b * in.flowf, warped, nb, 4, H / 2, W / 2, s));
                HIPCHK(h, launch_demosaic(warped, green + b * img, netin + b * img * kNetInC + 6, nb, H / 2, W / 2,
                                          (int64_t)H * W * kNetInC, kNetInC, 1, s));
            }
        }
        if (amax_netin) HIPCHK(h, launch_amax_reduce(netin, n, (int64_t)img * kNetInC, amax_netin, s));
        if (zero_now) {
            HIPCHK(h, hipMemsetAsync(zero_a, 0, zero_na * 4, s));
            HIPCHK(h, hipMemsetAsync(zero_b, 0, zero_nb * 4, s));
        }
    } else {
        // the whole NHWC16 input pixel in one pass: warp of the previous output | demosaic of the current frame |
        // warp of the demosaicked next frame
        float* next4 = nullptr;
        if (h->cfg.future) {
            next4 = h->next4 + o * img * 4;
            HIPCHK(h, launch_demosaic(rn_, green, next4, n, H / 2, W / 2, (int64_t)H * W * 4, 4, 1, s, (int64_t)in.rawf));
        }
        Scope sc(h, s, "netin(ha_green+netin_kernel)", 0.0, (double)n * img * (16.0 + 16.0 + 48.0 + (next4 ? 16.0 : 0.0)));
        // small frames without a future frame: the bound, the green plane and the network input in ONE launch
        if (!h->is_next() && !h->prev_noisy && netin_small_applies(n, H / 2, W / 2, h->cfg.future != 0)) {
            h->netin_proj = false;
            const float* rp_ = in.raw_prev ? in.raw_prev + o * in.rawf : nullptr;
            HIPCHK(h, launch_netin_small(rc_, rp_, h->lastden4 + o * img * 4, fp_, netin, n, H / 2, W / 2, (int64_t)in.rawf, (int64_t)in.flowf,
                                         amax_netin && !in.raw_prev ? amax_words(h, h->amax_feat_in, o) : nullptr, amax_netin, s,
                                         zero_now ? zero_a : nullptr, zero_na, zero_now ? zero_b : nullptr, zero_nb));
            goto prologue_features;
        }
        if (amax_netin) {
            // (block floating point) a bound of max |netin| from the raw frames and from the words PostConvs wrote last step;
            // with --prev_noisy_frame the "previous output" is a demosaicked frame whose raw data is gone: its own maximum
            const float* rp_ = in.raw_prev ? in.raw_prev + o * in.rawf : nullptr;
            HIPCHK(h, launch_netin_bound(rc_, rn_, rp_, n, H / 2, W / 2, (int64_t)in.rawf,
                                         in.raw_prev ? nullptr : amax_words(h, h->amax_feat_in, o), amax_netin, s,
                                         zero_now ? zero_a : nullptr, zero_na, zero_now ? zero_b : nullptr, zero_nb));
            if (h->prev_noisy && !in.raw_prev) HIPCHK(h, launch_amax_reduce(h->lastden4 + o * img * 4, n, (int64_t)img * 4, amax_netin, s, 1));
        }
        // ConvNeXtUnet: the input's only reader is the 1x1 projection of the first ConvBlock, which rides in the same kernel
        const NextBlk* first = h->is_next() && h->next_projfuse ? &h->nx[h->has_feat() ? NX_PRE : NX_ENC0_0] : nullptr;
        h->netin_proj = first != nullptr;
        HIPCHK(h, launch_netin(rc_, green, h->lastden4 + o * img * 4, fp_, next4, fn_, netin, n, H / 2, W / 2, s, (int64_t)in.rawf,
                               (int64_t)in.flowf, first ? first->w.proj_w : nullptr, first ? first->w.proj_b : nullptr,
                               first ? h->lv[0].t[0] + o * img * kF : nullptr));
    }
prologue_features:
    if (h->has_feat() && !nw) {
        h->featw_proj = next_pf_pre(h);
        Scope sc(h, s, "warp48_kernel", h->featw_proj ? 2.0 * 48 * 48 * n * img : 0.0, (double)n * img * (384.0 + 2.0));
        if (h->featw_proj)
            HIPCHK(h, launch_warp48_proj(h->lastfeat + o * img * kF, fp_, h->featw + o * img * kF, n, H, W, h->nx[NX_ENC0_0].half[1].frag,
                                         h->nx[NX_ENC0_0].half[1].inv_e, h->nx[NX_ENC0_0].w.proj_b, s, (int64_t)in.flowf));
        else
            HIPCHK(h, launch_warp48(h->lastfeat + o * img * kF, fp_, h->featw + o * img * kF, n, H, W, s, (int64_t)in.flowf));
    }
    return RVDD_OK;
}

// Every launch of one frame-step, in order, on stream s.  `init` = first frame of a video.
int enqueue_step(rvdd_t* h, const float* raw_prev, const float* raw_cur, const float* raw_next, const float* flow_prev,
                 const float* flow_next, int64_t raw_stride, int64_t flow_stride, float* out_rgb, bool init, hipStream_t s) {
    const bool nw = h->no_warp;
    const int B = h->cfg.batch, H = h->cfg.height, W = h->cfg.width;
    const size_t npix = (size_t)B * H * W;
    StepInputs in;
    in.raw_prev = init ? raw_prev : nullptr;
    in.raw_cur = raw_cur; in.raw_next = raw_next; in.flow_prev = flow_prev; in.flow_next = flow_next;
    in.rawf = raw_stride ? (size_t)raw_stride : (size_t)4 * (H / 2) * (W / 2);
    in.flowf = flow_stride ? (size_t)flow_stride : (size_t)2 * (H / 2) * (W / 2);
    // amax words: everything but the recurrent features' words this step reads (zero features at the start of a video: zero words)
    h->amax_base = (h->step_ctr & 1) * AMAX_NREG;
    h->amax_feat_in = AMAX_FEAT0 + (h->step_ctr + 2) % 3;
    h->amax_post_out = AMAX_FEAT0 + h->step_ctr % 3;
    if (h->bfp && h->split16 && !h->is_next()) {
        // the first step of a video starts from zero features: zero words; later steps find their set zeroed by the step before
        if (init) HIPCHK(h, hipMemsetAsync(h->amax, 0, amax_bytes(B, AMAX_SLOTS), s));
        h->amax_zero_pending = true;
    }
    if (init) {
        // lastden = n[:, :3] (demosaiced previous noisy frame), features = 0
        // (models/recurrent_model.py:233-245)
        HIPCHK(h, launch_demosaic(raw_prev, h->green, h->lastden4, B, H / 2, W / 2, (int64_t)H * W * 4, 4, 1, s, (int64_t)in.rawf));
        if (h->has_feat()) HIPCHK(h, hipMemsetAsync(h->lastfeat, 0, npix * kF * sizeof(float), s));
    }
    // without warping the previous features are read in place: the net consumes them in its first layer and only
    // its last one writes the new ones
    const int rc = run_net(h, h->netin, nw ? h->lastfeat : h->featw, h->lastfeat, out_rgb, h->lastden4, s, &in);
    if (rc == RVDD_OK && h->prev_noisy)     // store_frame = the noisy current frame (models/recurrent_model.py:335-337)
        HIPCHK(h, launch_demosaic(raw_cur, h->green, h->lastden4, B, H / 2, W / 2, (int64_t)H * W * 4, 4, 1, s, (int64_t)in.rawf));
    return rc;
}

constexpr size_t kMaxStepGraphs = 128;      // one per distinct set of caller buffers; least recently used goes first

}  // namespace

extern "C" {

// A frame-step is ~30-45 launches.  The schedule is fixed by (configuration, options, first-frame flag) and the six
// caller pointers, so each distinct pointer set can be captured once into a hipGraph (on a stream of the handle) and
// replayed afterwards -- rvdd_set_option(h, "graphs", 1) or RVDD_GRAPH=1; the caller's stream is joined on both sides
// with events, so stream order is what it would be launch by launch.  OFF by default: on ROCm 7.2 the replay is
// SLOWER than the eager launches it replaces at every size measured (profiles/r02_e_hipgraph_step_ab.log: 256x256
// B = 1 2110 vs 2440 frames/s, B = 4 5035 vs 5320; 720p B = 1 398.6 vs 404.5, B = 4 456.9 vs 459.4).  The eager path
// is not host-bound -- launches are asynchronous and the queue stays full -- and kernel boundaries cost the same
// either way, so a graph has only its own launch cost to add.  Parity is identical (the GPU suite passes in both modes).
int rvdd_step(rvdd_t* h, const float* raw_prev, const float* raw_cur, const float* raw_next,
              const float* flow_prev, const float* flow_next, float* out_rgb, void* stream) {
    return rvdd_step_strided(h, raw_prev, raw_cur, raw_next, flow_prev, flow_next, 0, 0, out_rgb, stream);
}

int rvdd_step_strided(rvdd_t* h, const float* raw_prev, const float* raw_cur, const float* raw_next,
                      const float* flow_prev, const float* flow_next, int64_t raw_stride, int64_t flow_stride,
                      float* out_rgb, void* stream) {
    if (!h) return RVDD_ERR_ARG;
    ENTER(h);
    if (!h->finalized) return fail(h, RVDD_ERR_STATE, "rvdd_step: weights not finalized");
    {
        const int64_t rd = (int64_t)4 * (h->cfg.height / 2) * (h->cfg.width / 2), fd = rd / 2;
        if ((raw_stride && raw_stride < rd) || (flow_stride && flow_stride < fd))
            return fail(h, RVDD_ERR_ARG, "rvdd_step_strided: a batch stride must be 0 (dense) or at least one sequence (%lld / %lld floats)",
                        (long long)rd, (long long)fd);
    }
    const bool nw = h->no_warp;
    if (!raw_cur || (!flow_prev && !nw) || !out_rgb) return fail(h, RVDD_ERR_ARG, "rvdd_step: raw_cur, flow_prev and out_rgb are required");
    if (h->need_init && !raw_prev) return fail(h, RVDD_ERR_ARG, "rvdd_step: raw_prev is required on the first step of a video");
    if (h->cfg.future && (!raw_next || (!flow_next && !nw))) return fail(h, RVDD_ERR_ARG, "rvdd_step: raw_next and flow_next are required when future=1");
    if (nw) flow_prev = flow_next = nullptr;      // the flows are not looked at (the reference's dataset does not even load them)
    hipStream_t s = static_cast<hipStream_t>(stream);
    // need_init is cleared only once the step has been enqueued: a step that failed half way leaves the handle asking
    // for the first frame of a video again (raw_prev, zeroed features), never a later frame on stale state
    const bool init = h->need_init;
    auto eager = [&]() -> int {
        const int rc = enqueue_step(h, raw_prev, raw_cur, raw_next, flow_prev, flow_next, raw_stride, flow_stride, out_rgb, init, s);
        if (rc == RVDD_OK) {
            h->need_init = false;
            h->step_ctr = (h->step_ctr + 1) % 6;      // (the amax words' set and slots follow it: & 1, % 3)
        }
        return rc;
    };
    if (!h->use_graphs || h->prof_on || !h->ran_eagerly || !h->gstream) {
        h->ran_eagerly = true;
        return eager();
    }
    rvdd_handle::StepKey key{{init ? raw_prev : nullptr, raw_cur, raw_next, flow_prev, flow_next, out_rgb},
                             {raw_stride, flow_stride}, (init ? 1 : 0) | (h->serpentine ? 2 : 0) | (h->step_ctr << 2)};
    auto it = h->graphs.find(key);
    if (it == h->graphs.end()) {
        hipGraph_t g = nullptr;
        hipGraphExec_t ex = nullptr;
        hipError_t e = hipStreamBeginCapture(h->gstream, hipStreamCaptureModeThreadLocal);
        int rc = RVDD_OK;
        if (e == hipSuccess) {
            const bool serp = h->serpentine;
            rc = enqueue_step(h, raw_prev, raw_cur, raw_next, flow_prev, flow_next, raw_stride, flow_stride, out_rgb, init, h->gstream);
            h->serpentine = serp;                              // the replay below advances it
            e = hipStreamEndCapture(h->gstream, &g);
        }
        if (e == hipSuccess && rc == RVDD_OK) e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
        if (e != hipSuccess || rc != RVDD_OK) {
            // no graph for this process: run this step and every later one launch by launch, on the caller's stream
            // (whatever failed -- the capture, the instantiation or a launch inside the capture -- nothing has run yet)
            if (g) (void)hipGraphDestroy(g);
            (void)hipGetLastError();
            h->use_graphs = 0;
            return eager();
        }
        if (h->graphs.size() >= kMaxStepGraphs) {
            auto old = h->graphs.begin();
            for (auto jt = h->graphs.begin(); jt != h->graphs.end(); ++jt)
                if (jt->second.last_use < old->second.last_use) old = jt;
            (void)hipGraphExecDestroy(old->second.exec);
            (void)hipGraphDestroy(old->second.graph);
            h->graphs.erase(old);
        }
        rvdd_handle::StepGraph sg;
        sg.graph = g;
        sg.exec = ex;
        it = h->graphs.emplace(key, sg).first;
    }
    it->second.last_use = ++h->graph_tick;
    HIPCHK(h, hipEventRecord(h->g_in, s));
    HIPCHK(h, hipStreamWaitEvent(h->gstream, h->g_in, 0));
    HIPCHK(h, hipGraphLaunch(it->second.exec, h->gstream));
    HIPCHK(h, hipEventRecord(h->g_out, h->gstream));
    HIPCHK(h, hipStreamWaitEvent(s, h->g_out, 0));
    h->need_init = false;
    h->step_ctr = (h->step_ctr + 1) % 6;
    if (seq_major_on(h)) h->serpentine = !h->serpentine;
    return RVDD_OK;
}

int rvdd_get_state(rvdd_t* h, float* lastden, float* lastfeat, void* stream) {
    if (!h) return RVDD_ERR_ARG;
    ENTER(h);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int B = h->cfg.batch, H = h->cfg.height, W = h->cfg.width;
    if (lastden) HIPCHK(h, launch_nhwc_to_nchw(h->lastden4, lastden, B, 3, H, W, 4, s));
    if (lastfeat) {
        if (!h->has_feat()) return fail(h, RVDD_ERR_ARG, "rvdd_get_state: this architecture has no recurrent features");
        HIPCHK(h, launch_nhwc_to_nchw(h->lastfeat, lastfeat, B, kF, H, W, kF, s));
    }
    return RVDD_OK;
}

int rvdd_set_state(rvdd_t* h, const float* lastden, const float* lastfeat, void* stream) {
    if (!h) return RVDD_ERR_ARG;
    ENTER(h);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int B = h->cfg.batch, H = h->cfg.height, W = h->cfg.width;
    if (lastfeat && !h->has_feat()) return fail(h, RVDD_ERR_ARG, "rvdd_set_state: this architecture has no recurrent features");
    if (lastden) {
        HIPCHK(h, launch_nchw_to_nhwc(lastden, h->lastden4, B, 3, H, W, 4, s));
        h->need_init = false;
    }
    if (lastfeat) HIPCHK(h, launch_nchw_to_nhwc(lastfeat, h->lastfeat, B, kF, H, W, kF, s));
    if ((lastden || lastfeat) && h->bfp && h->split16 && !h->is_next()) {
        // The words the next step reads as the bound of "the previous output" (block floating point): features and output frame
        // together, as PostConvs leaves them -- rebuilt from the state as it now stands, whichever half the caller replaced
        unsigned* w = amax_words(h, AMAX_FEAT0 + (h->step_ctr + 2) % 3);
        HIPCHK(h, hipMemsetAsync(w, 0, amax_bytes(B, 1), s));
        if (h->has_feat()) HIPCHK(h, launch_amax_reduce(h->lastfeat, B, (int64_t)H * W * kF, w, s));
        HIPCHK(h, launch_amax_reduce(h->lastden4, B, (int64_t)H * W * 4, w, s));
    }
    return RVDD_OK;
}

int rvdd_psnr_l1(rvdd_t* h, const float* den, const float* gt, int64_t count, float* out2, void* stream) {
    if (!h || !den || !gt || !out2 || count <= 0) return fail(h, RVDD_ERR_ARG, "rvdd_psnr_l1: bad argument");
    ENTER(h);
    hipStream_t s = static_cast<hipStream_t>(stream);
    int nblk = (int)((count + 256 * 16 - 1) / (256 * 16));
    if (nblk > 1024) nblk = 1024;
    HIPCHK(h, launch_loss_reduce(den, gt, count, h->loss_partial, nblk, h->loss_result, s));
    double r[2];
    HIPCHK(h, hipMemcpyAsync(r, h->loss_result, sizeof r, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    if (hipError_t e = tvl1_check(h->tvl1, s); e != hipSuccess)       // an asynchronous flow batch in front of the frames just measured
        return fail(h, RVDD_ERR_HIP, "rvdd_psnr_l1: an asynchronous rvdd_tvl1flow_batch before this call failed: %s", hipGetErrorString(e));
    out2[0] = (float)(100.0 * r[0] / (double)count);
    out2[1] = (float)(10.0 * std::log10(4.0 / (r[1] / (double)count)));
    return RVDD_OK;
}

int rvdd_unet_forward(rvdd_t* h, const float* x, const float* feat_in, float* out, float* feat_out, void* stream) {
    if (!h) return RVDD_ERR_ARG;
    ENTER(h);
    if (!h->finalized) return fail(h, RVDD_ERR_STATE, "rvdd_unet_forward: weights not finalized");
    if (!x || !out) return fail(h, RVDD_ERR_ARG, "rvdd_unet_forward: x and out are required");
    if (h->has_feat() && !feat_in)
        return fail(h, RVDD_ERR_STATE, "Old features is None, please call get_rec_nil_features first.");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int B = h->cfg.batch, H = h->cfg.height, W = h->cfg.width;
    HIPCHK(h, launch_nchw_to_nhwc(x, h->netin, B, h->cin_real(), H, W, kNetInC, s));
    if (h->has_feat()) HIPCHK(h, launch_nchw_to_nhwc(feat_in, h->featw, B, kF, H, W, kF, s));
    // amax words of the caller's maps (block floating point of the split-f16 convs); the recurrent features' words stay as they are
    h->amax_base = 2 * AMAX_NREG;              // the set of its own: the frame-steps' sets and the recurrent slots stay untouched
    h->amax_feat_in = h->amax_base + AMAX_REL_FWDFEAT;
    h->amax_post_out = amax_layer(h, CU_POST);
    if (h->bfp && h->split16 && !h->is_next()) {
        HIPCHK(h, hipMemsetAsync(amax_words(h, h->amax_base), 0, amax_bytes(B, AMAX_NREG), s));
        HIPCHK(h, launch_amax_reduce(h->netin, B, (int64_t)H * W * kNetInC, amax_words(h, h->amax_base + AMAX_REL_NETIN), s));
        if (h->has_feat()) HIPCHK(h, launch_amax_reduce(h->featw, B, (int64_t)H * W * kF, amax_words(h, h->amax_feat_in), s));
    }
    RC(run_net(h, h->netin, h->featw, h->lv[0].t[2], out, nullptr, s, nullptr));
    if (h->has_feat() && feat_out) HIPCHK(h, launch_nhwc_to_nchw(h->lv[0].t[2], feat_out, B, kF, H, W, kF, s));
    return RVDD_OK;
}

int rvdd_demosaic_ha(rvdd_t* h, const float* raw, int32_t n, int32_t hh, int32_t ww, float* rgb, void* stream) {
    if (h && n == 0) return RVDD_OK;       // an empty batch is valid and launches nothing
    if (!h || !raw || !rgb || n < 0 || hh < 1 || ww < 1) return fail(h, RVDD_ERR_ARG, "rvdd_demosaic_ha: bad argument");
    ENTER(h);
    hipStream_t s = static_cast<hipStream_t>(stream);
    RC(ensure_scratch(h, (size_t)n * 4 * hh * ww * sizeof(float)));
    const int64_t hw = (int64_t)4 * hh * ww;
    HIPCHK(h, launch_demosaic(raw, h->scratch, rgb, n, hh, ww, 3 * hw, 1, (int)hw, s));
    return RVDD_OK;
}

int rvdd_warp_bicubic(rvdd_t* h, const float* x, const float* flow, int32_t n, int32_t c, int32_t H, int32_t W,
                      float* y, void* stream) {
    if (h && n == 0) return RVDD_OK;
    if (!h || !x || !flow || !y || n < 0 || c < 1 || H < 2 || W < 2) return fail(h, RVDD_ERR_ARG, "rvdd_warp_bicubic: bad argument");
    ENTER(h);
    HIPCHK(h, launch_warp_nchw(x, flow, y, n, c, H, W, static_cast<hipStream_t>(stream)));
    return RVDD_OK;
}

int rvdd_upsample_factor_2(rvdd_t* h, const float* t, int32_t n, int32_t c, int32_t hh, int32_t ww,
                           float multiply_by, float* out, void* stream) {
    if (h && n == 0) return RVDD_OK;
    if (!h || !t || !out || n < 0 || c < 1 || hh < 1 || ww < 1) return fail(h, RVDD_ERR_ARG, "rvdd_upsample_factor_2: bad argument");
    ENTER(h);
    HIPCHK(h, launch_upsample_flow(t, out, n * c, hh, ww, multiply_by, static_cast<hipStream_t>(stream)));
    return RVDD_OK;
}

int rvdd_tvl1flow(rvdd_t* h, const float* I0, const float* I1, float* u, int32_t nx, int32_t ny, int32_t* iterations,
                  void* stream) {
    if (!h || !I0 || !I1 || !u || nx < 16 || ny < 16) return fail(h, RVDD_ERR_ARG, "rvdd_tvl1flow: bad argument (images must be >= 16x16)");
    ENTER(h);
    if (!tvl1_size_ok(nx, ny))
        return fail(h, RVDD_ERR_ARG, "rvdd_tvl1flow: image too skinny for its pyramid (the reference reads out of bounds at this size)");
    if (!h->tvl1 || tvl1_ws_nx(h->tvl1) != nx || tvl1_ws_ny(h->tvl1) != ny) {
        HIPCHK(h, hipDeviceSynchronize());
        HIPCHK(h, tvl1_check(h->tvl1, static_cast<hipStream_t>(stream)));      // what an asynchronous batch on the old workspace left unread
        tvl1_free(h->tvl1);
        h->tvl1 = nullptr;
        HIPCHK(h, tvl1_alloc(&h->tvl1, nx, ny));
    }
    int it = 0;
    HIPCHK(h, tvl1_run(h->tvl1, I0, I1, u, static_cast<hipStream_t>(stream), iterations ? &it : nullptr));
    if (iterations) *iterations = it;
    return RVDD_OK;
}

int rvdd_ppipe(rvdd_t* h, const float* img, int32_t n, int32_t H, int32_t W, int64_t sn, int64_t sc, int64_t sy, int64_t sx,
               int32_t bit_depth, double rgb_gain, double red_gain, double blue_gain, int32_t iso, uint8_t* out_u8,
               float* out_f32, void* stream) {
    if (h && n == 0) return RVDD_OK;
    if (!h || !img || !out_u8 || n < 0 || H < 1 || W < 1) return fail(h, RVDD_ERR_ARG, "rvdd_ppipe: bad argument");
    ENTER(h);
    if (!(rgb_gain != 0.0) || !(red_gain != 0.0) || !(blue_gain != 0.0)) return fail(h, RVDD_ERR_ARG, "rvdd_ppipe: zero gain");
    // fwd_ppipe.py:29: a float32 tensor of Python-double quotients
    const float gains[3] = {(float)(1.0 / (red_gain * rgb_gain)), (float)(1.0 / rgb_gain), (float)(1.0 / (blue_gain * rgb_gain))};
    HIPCHK(h, launch_ppipe(img, n, H, W, sn, sc, sy, sx, bit_depth, gains, iso, out_u8, out_f32, static_cast<hipStream_t>(stream)));
    return RVDD_OK;
}

int rvdd_srgb_metrics(rvdd_t* h, const uint8_t* a, const uint8_t* b, int32_t n, int32_t H, int32_t W, double* psnr,
                      double* ssim, void* stream) {
    if (!h || !a || !b || n < 1) return fail(h, RVDD_ERR_ARG, "rvdd_srgb_metrics: bad argument");
    ENTER(h);
    if (H < 7 || W < 7) return fail(h, RVDD_ERR_ARG, "rvdd_srgb_metrics: win_size exceeds image extent (images must be >= 7x7)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    RC(ensure_scratch(h, srgb_metrics_workspace(n, H, W)));
    HIPCHK(h, launch_srgb_metrics(a, b, n, H, W, h->scratch, s));
    std::vector<unsigned long long> ssd(n);
    std::vector<double> sums(n);
    HIPCHK(h, hipMemcpyAsync(ssd.data(), h->scratch, (size_t)n * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipMemcpyAsync(sums.data(), reinterpret_cast<char*>(h->scratch) + (size_t)n * 8, (size_t)n * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    for (int i = 0; i < n; ++i) {
        // mean((a/255 - b/255)^2) = SSD / 255^2 / count; 10 log10(1 / 0) = inf as in numpy
        if (psnr) psnr[i] = 10.0 * std::log10(1.0 / ((double)ssd[i] / (255.0 * 255.0) / ((double)H * W * 3)));
        if (ssim) ssim[i] = sums[i] / (3.0 * (double)(H - 6) * (double)(W - 6));
    }
    return RVDD_OK;
}

int rvdd_tvl1flow_batch(rvdd_t* h, const float* I0, const float* I1, float* u, int32_t n, int32_t nx, int32_t ny,
                        int32_t* iterations, void* stream) {
    if (h && n == 0) return RVDD_OK;
    if (!h || !I0 || !I1 || !u || n < 0 || nx < 16 || ny < 16)
        return fail(h, RVDD_ERR_ARG, "rvdd_tvl1flow_batch: bad argument (images must be >= 16x16)");
    ENTER(h);
    if (!tvl1_size_ok(nx, ny))
        return fail(h, RVDD_ERR_ARG, "rvdd_tvl1flow_batch: image too skinny for its pyramid (the reference reads out of bounds at this size)");
    if (!h->tvl1 || tvl1_ws_nx(h->tvl1) != nx || tvl1_ws_ny(h->tvl1) != ny) {
        HIPCHK(h, hipDeviceSynchronize());
        HIPCHK(h, tvl1_check(h->tvl1, static_cast<hipStream_t>(stream)));      // what an asynchronous batch on the old workspace left unread
        tvl1_free(h->tvl1);
        h->tvl1 = nullptr;
        HIPCHK(h, tvl1_alloc(&h->tvl1, nx, ny));
    }
    std::vector<int> it((size_t)n, 0);
    HIPCHK(h, tvl1_run_batch(h->tvl1, I0, I1, u, n, static_cast<hipStream_t>(stream), iterations ? it.data() : nullptr, h->tvl1_async));
    if (iterations)
        for (int i = 0; i < n; ++i) iterations[i] = it[(size_t)i];
    return RVDD_OK;
}

int rvdd_profile_enable(rvdd_t* h, int32_t on) {
    if (!h) return RVDD_ERR_ARG;
    ENTER(h);
    RC(prof_flush(h));
    if (on) for (auto& p : h->prof) { p.seen = p.launches = 0; p.ms = p.flops = p.bytes = 0; }
    h->prof_on = on != 0;
    return RVDD_OK;
}

int rvdd_profile_select(rvdd_t* h, const char* kernel_class, int32_t stride) {
    if (!h || stride < 1) return fail(h, RVDD_ERR_ARG, "rvdd_profile_select: bad argument");
    h->prof_filter = kernel_class ? kernel_class : "";
    h->prof_stride = stride;
    return RVDD_OK;
}

int rvdd_profile_count(const rvdd_t* h) { return h ? (int)h->prof.size() : 0; }

int rvdd_profile_read(rvdd_t* h, int32_t idx, char* name, int32_t name_cap, int64_t* launches,
                      double* total_ms, double* flops, double* bytes) {
    if (!h || idx < 0 || idx >= (int)h->prof.size()) return fail(h, RVDD_ERR_ARG, "rvdd_profile_read: bad index");
    ENTER(h);
    RC(prof_flush(h));
    const ProfClass& p = h->prof[idx];
    if (name && name_cap > 0) {
        std::strncpy(name, p.name.c_str(), name_cap - 1);
        name[name_cap - 1] = 0;
    }
    if (launches) *launches = p.launches;
    if (total_ms) *total_ms = p.ms;
    if (flops) *flops = p.flops;
    if (bytes) *bytes = p.bytes;
    return RVDD_OK;
}

int rvdd_debug_conv_bench(rvdd_t* h, int32_t variant, int32_t level, int32_t iters, float* ms, void* stream) {
    if (!h || !ms || level < 0 || level > 3 || iters < 1) return fail(h, RVDD_ERR_ARG, "rvdd_debug_conv_bench: bad argument");
    ENTER(h);
    if (!h->finalized || h->is_next()) return fail(h, RVDD_ERR_STATE, "rvdd_debug_conv_bench: needs a finalized convunet handle");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool was = h->prof_on, was_wino = h->use_wino, was_split = h->split16;
    h->prof_on = false;
    h->use_wino = h->force_wino = variant == 3;   // variant 3 = Winograd kernel, 4 = split-f16 kernel, 0..2 = direct kernel variants
    h->split16 = variant == 4;
    conv3x3_set_variant(variant >= 3 ? 0 : variant);
    ConvCall c;
    c.in = h->lv[level].t[0]; c.out = h->lv[level].t[1]; c.H = h->lv[level].H; c.W = h->lv[level].W; c.epi = EPI_RELU;
    const Conv3& L = h->cu[CU_ENC1_1];
    int rc = run_conv(h, L, c, s);   // warm-up (also sets the function attribute)
    if (rc == RVDD_OK) {
        (void)hipEventRecord(h->t0, s);
        for (int i = 0; i < iters && rc == RVDD_OK; ++i) rc = run_conv(h, L, c, s);
        (void)hipEventRecord(h->t1, s);
        (void)hipEventSynchronize(h->t1);
        float t = 0.f;
        (void)hipEventElapsedTime(&t, h->t0, h->t1);
        *ms = t / iters;
    }
    conv3x3_set_variant(0);
    h->prof_on = was;
    h->use_wino = was_wino;
    h->force_wino = false;
    h->split16 = was_split;
    return rc;
}

int rvdd_timer_start(rvdd_t* h, void* stream) {
    if (!h) return RVDD_ERR_ARG;
    ENTER(h);
    HIPCHK(h, hipEventRecord(h->t0, static_cast<hipStream_t>(stream)));
    return RVDD_OK;
}

int rvdd_timer_stop_ms(rvdd_t* h, void* stream, float* ms) {
    if (!h || !ms) return RVDD_ERR_ARG;
    ENTER(h);
    HIPCHK(h, hipEventRecord(h->t1, static_cast<hipStream_t>(stream)));
    HIPCHK(h, hipEventSynchronize(h->t1));
    HIPCHK(h, hipEventElapsedTime(ms, h->t0, h->t1));
    return RVDD_OK;
}

}  // extern "C"
