// api.hip — the C ABI of libmoonsr_hip.so (include/moonsr.h): handle, weight re-layout, workspace planning
// and the per-call launch plan of the generator(call).  Host logic only; kernels live in the other files.
#include "../../include/moonsr.h"
#include "kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

using namespace msr;

namespace {

thread_local std::string g_create_error;

struct WeightSpec {
    std::string name;
    std::vector<int64_t> shape;
    bool loaded = false;
};

enum OpType { OP_CONV, OP_SMALLCIN, OP_MOMENTS, OP_MOMENTS_SLABS, OP_NORMACT, OP_DENSE, OP_LATENT, OP_HEAD, OP_DIRECT, OP_GBR };
enum Family { FAM_CONV = 0, FAM_SMALLCIN, FAM_MOMENTS, FAM_NORMACT, FAM_DENSE, FAM_LATENT, FAM_HEAD, FAM_DIRECT,
              FAM_COUNT };
const char* kFamilyName[FAM_COUNT] = {"conv_igemm", "conv_smallcin", "moments", "norm_act", "dense",
                                      "latent", "head_up_conv4x4", "conv_direct"};

struct Op {
    OpType type;
    double flops = 0, bytes = 0;
    // flags for per-call pointers
    bool src_is_input = false, out_is_output = false, eps_is_input = false;
    bool on_aux = false;              // depends on the call's input only: runs on the handle's auxiliary stream
    hipEvent_t done = nullptr;        // recorded on the auxiliary stream after an on_aux op
    hipEvent_t wait = nullptr;        // the main stream waits for this before launching the op
    int aux_group = -1;               // on_aux ops and their consumers: one event / one wait per group
    ConvParams conv{}; int epi = 0, tile = 0;
    int stat_slabs = 0;               // > 0: the conv's epilogue also writes partial output moments (fused)
    SmallCinParams sc{};
    GbrParams gbr{};                  // OP_GBR: mask embedding + gamma|beta conv + SPADE epilogue in one launch (conv_gbr.hip)
    struct { const float* x; int G, P, C; float eps; float* mean; float* stdv; } mom{};
    NormActParams na{};
    struct { const float* x; const float* W; const float* bias; float* y; int B, K, N; } dense{};
    struct { const float* mv; float* z; int B, L, sampler; } lat{};
    struct { const float* x; const float* weff; float bias; int B, r, C; float slope; int tanh_out; int x_py, x_pb; } head{};
    DirectConvParams dc{};
};

struct ProfRec { int fam; hipEvent_t a, b; double flops, bytes; int launches; };

}  // namespace

struct msr_handle {
    msr_config cfg{};
    int S = 0, B = 0, L = 0, variant = 0;
    int prec = 0;                                // PREC_F32 or PREC_BF16X3 (cfg.flags & MSR_FLAG_BF16X3)
    bool gb_f16x2 = false;                       // MSR_FLAG_GB_F16X2: 2-term fp16 products in the gamma|beta convs
    bool fp8 = false;                            // MSR_FLAG_FP8: declared non-parity mode (fp8 weights x bf8 activations)
    bool f16c = false;                           // MSR_FLAG_F16C: fp16 main term + fp8 cross terms in the chip-filling convs
    bool f16m = false;                           // MSR_FLAG_F16_MAIN: F16C without the cross terms in the stream / resident kernels
    std::string err;
    std::vector<WeightSpec> specs;
    std::map<std::string, int> spec_index;
    std::map<std::string, float*> dev;          // device tensors: weights (re-laid-out) and workspace
    std::map<std::string, size_t> dev_bytes;
    std::map<std::string, std::vector<float>> host_small;   // small host copies needed at plan time (BN, head)
    size_t total_bytes = 0;
    bool planned = false;
    std::vector<Op> ops;
    double fwd_flops = 0;
    double* mom_partial = nullptr;
    float* dense_partial = nullptr;
    float* conv_partial = nullptr;     // split-K workspace [ksplit][M][N]
    size_t conv_partial_floats = 0;
    float* stat_ws = nullptr;          // fused-moments slabs [P][3][N] of the conv that ran last
    size_t stat_ws_floats = 0;
    float* z = nullptr;
    // tiler
    double* window = nullptr;     // [S-2p, S-2p] float64
    int* stitch_grid = nullptr;
    int stitch_grid_cap = 0;
    // auxiliary stream: the SPADE mask embeddings depend only on the call's input, so they are launched on a
    // second stream and overlap the encoder and the low-resolution (latency-bound) layers
    hipStream_t aux = nullptr;
    hipEvent_t ev_fork = nullptr;
    // HIP graphs of the launch plan, one per (input, noise, output) pointer triple (msr_graph_enable)
    struct GraphEntry { const float* in; const float* eps; float* out; hipGraph_t graph; hipGraphExec_t exec; uint64_t last_use; };
    struct Triple { const float* in; const float* eps; float* out; };
    int graph_on = 0;
    std::vector<GraphEntry> graphs;              // at most 8, least recently used evicted
    std::vector<Triple> seen_once;               // triples run eagerly once: a triple is captured on its SECOND sighting
    uint64_t graph_clock = 0;
    int gate_op = -1;                            // index of the first op of the matrix-bound part (msr_forward_gated)
    // profiling
    int prof_on = 0;                               // 0 off, 1 every launch, 2 runs of conv launches only
    std::vector<ProfRec> prof;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
};

namespace {

int fail(msr_handle* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf; else g_create_error = buf;
    return code;
}

#define HIPCHK(h, call)                                                                                     \
    do {                                                                                                    \
        hipError_t e_ = (call);                                                                             \
        if (e_ != hipSuccess)                                                                               \
            return fail(h, MSR_ERR_DEVICE, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                        __LINE__);                                                                          \
    } while (0)

const int kGenFilters[6] = {1024, 1024, 1024, 512, 256, 128};
const int kEncChannels[5] = {64, 128, 256, 512, 512};
const int kP2PDown[8] = {64, 128, 256, 512, 512, 512, 512, 512};
const int kP2PUp[7] = {512, 512, 512, 512, 256, 128, 64};

void add_spec(msr_handle* h, const std::string& name, std::vector<int64_t> shape) {
    h->spec_index[name] = (int)h->specs.size();
    h->specs.push_back({name, std::move(shape), false});
}

void build_specs(msr_handle* h) {
    char n[128];
    if (h->variant == MSR_PIX2PIX) {
        int cin = 2;
        for (int i = 1; i <= 8; ++i) {
            const int c = kP2PDown[i - 1];
            snprintf(n, sizeof n, "p2p.down%d.kernel", i); add_spec(h, n, {4, 4, cin, c});
            if (i > 1)
                for (const char* q : {"gamma", "beta", "moving_mean", "moving_variance"}) {
                    snprintf(n, sizeof n, "p2p.down%d.bn.%s", i, q); add_spec(h, n, {c});
                }
            cin = c;
        }
        for (int i = 1; i <= 7; ++i) {
            const int c = kP2PUp[i - 1];
            snprintf(n, sizeof n, "p2p.up%d.kernel", i); add_spec(h, n, {4, 4, c, cin});
            for (const char* q : {"gamma", "beta", "moving_mean", "moving_variance"}) {
                snprintf(n, sizeof n, "p2p.up%d.bn.%s", i, q); add_spec(h, n, {c});
            }
            cin = c + kP2PDown[6 - (i - 1)];
        }
        add_spec(h, "p2p.last.kernel", {4, 4, 1, cin});
        add_spec(h, "p2p.last.bias", {1});
        return;
    }
    const int S = h->S, L = h->L;
    int cin = 2;
    for (int i = 1; i <= 5; ++i) {
        const int c = kEncChannels[i - 1];
        snprintf(n, sizeof n, "enc.ds%d.kernel", i); add_spec(h, n, {3, 3, cin, c});
        if (i > 1) {
            snprintf(n, sizeof n, "enc.ds%d.in.gamma", i); add_spec(h, n, {c});
            snprintf(n, sizeof n, "enc.ds%d.in.beta", i); add_spec(h, n, {c});
        }
        cin = c;
    }
    const int64_t flat = (int64_t)(S / 32) * (S / 32) * 512;
    for (const char* q : {"mean", "variance"}) {
        snprintf(n, sizeof n, "enc.%s.kernel", q); add_spec(h, n, {flat, L});
        snprintf(n, sizeof n, "enc.%s.bias", q); add_spec(h, n, {L});
    }
    const int sw = S / 64;
    add_spec(h, "gen.dense.kernel", {L, (int64_t)sw * sw * 1024});
    add_spec(h, "gen.dense.bias", {(int64_t)sw * sw * 1024});
    cin = 1024;
    for (int i = 1; i <= 6; ++i) {
        const int f = kGenFilters[i - 1];
        const bool learned = f != cin;
        for (int j = 1; j <= (learned ? 3 : 2); ++j) {
            const int c = j == 2 ? f : cin;
            snprintf(n, sizeof n, "gen.rb%d.spade_%d.conv.kernel", i, j); add_spec(h, n, {3, 3, 2, 128});
            snprintf(n, sizeof n, "gen.rb%d.spade_%d.conv.bias", i, j); add_spec(h, n, {128});
            snprintf(n, sizeof n, "gen.rb%d.spade_%d.conv_gamma.kernel", i, j); add_spec(h, n, {3, 3, 128, c});
            snprintf(n, sizeof n, "gen.rb%d.spade_%d.conv_gamma.bias", i, j); add_spec(h, n, {c});
            snprintf(n, sizeof n, "gen.rb%d.spade_%d.conv_beta.kernel", i, j); add_spec(h, n, {3, 3, 128, c});
            snprintf(n, sizeof n, "gen.rb%d.spade_%d.conv_beta.bias", i, j); add_spec(h, n, {c});
        }
        for (int j = 1; j <= (learned ? 3 : 2); ++j) {
            const int ci = j == 2 ? f : cin;
            snprintf(n, sizeof n, "gen.rb%d.conv_%d.kernel", i, j); add_spec(h, n, {3, 3, ci, f});
            snprintf(n, sizeof n, "gen.rb%d.conv_%d.bias", i, j); add_spec(h, n, {f});
        }
        cin = f;
    }
    add_spec(h, "gen.head.kernel", {4, 4, 128, 1});
    add_spec(h, "gen.head.bias", {1});
}

int dev_alloc(msr_handle* h, const std::string& key, size_t floats, bool zero, float** out) {
    auto it = h->dev.find(key);
    if (it != h->dev.end()) {
        if (h->dev_bytes[key] != floats * sizeof(float))
            return fail(h, MSR_ERR_STATE, "buffer %s re-allocated with a different size", key.c_str());
        *out = it->second;
        return MSR_OK;
    }
    float* p = nullptr;
    hipError_t e = hipMalloc(&p, std::max<size_t>(floats, 4) * sizeof(float));
    if (e != hipSuccess)
        return fail(h, MSR_ERR_NOMEM, "hipMalloc of %zu bytes for %s failed: %s", floats * sizeof(float), key.c_str(),
                    hipGetErrorString(e));
    if (zero) {
        e = hipMemset(p, 0, std::max<size_t>(floats, 4) * sizeof(float));
        if (e != hipSuccess) return fail(h, MSR_ERR_DEVICE, "hipMemset for %s failed", key.c_str());
    }
    h->dev[key] = p;
    h->dev_bytes[key] = floats * sizeof(float);
    h->total_bytes += floats * sizeof(float);
    *out = p;
    return MSR_OK;
}

int upload(msr_handle* h, const std::string& key, const float* host, size_t floats) {
    float* d = nullptr;
    int rc = dev_alloc(h, key, floats, false, &d);
    if (rc) return rc;
    HIPCHK(h, hipMemcpy(d, host, floats * sizeof(float), hipMemcpyHostToDevice));
    return MSR_OK;
}

float* D(msr_handle* h, const std::string& key) {
    auto it = h->dev.find(key);
    return it == h->dev.end() ? nullptr : it->second;
}

// HWIO [kh,kw,Cin,Cout] -> [tap][Cout][Cin]  (K contiguous per output channel, the igemm B-operand layout)
void hwio_to_tap_oc_ic(const float* src, float* dst, int taps, int cin, int cout, int dst_rows, const int* rowmap) {
    for (int t = 0; t < taps; ++t)
        for (int ci = 0; ci < cin; ++ci) {
            const float* s = src + ((size_t)t * cin + ci) * cout;
            for (int co = 0; co < cout; ++co) {
                const int row = rowmap ? rowmap[co] : co;
                dst[((size_t)t * dst_rows + row) * cin + ci] = s[co];
            }
        }
}

// Weights consumed by conv_igemm_bf16x3 are uploaded in MFMA-fragment order (conv_igemm.hip):
//   [tap][chunk of 32 k][n-tile of 32][kg][hi|lo][lane = 32*h + j][8 bf16],  value = W[tap][32*nt + j][32*cc + 16*kg + 8*h + e]
// `host` is the kernel layout [taps][N][Cin].
int upload_conv_weight(msr_handle* h, const std::string& key, const float* host, size_t floats, int taps, int N,
                       int Cin, bool frag, bool f16 = false) {
    if (h->prec != PREC_BF16X3) return upload(h, key, host, floats);
    if (N % 32 || Cin % 32 || (size_t)taps * N * Cin != floats)
        return fail(h, MSR_ERR_INVALID, "%s: bf16x3 needs Cin and Cout multiples of 32", key.c_str());
    std::vector<float> t(floats);
    if (f16) {
        // split-fp16 image of [tap][N][Cin] (PREC_F16X2 reads only the hi half of every chunk)
        for (size_t i = 0; i + 3 < floats; i += 4)
            msr_store_split4_f16(t.data() + (i & ~(size_t)31), (int)(i & 31), host[i], host[i + 1], host[i + 2], host[i + 3]);
        return upload(h, key, t.data(), floats);
    }
    if (!frag) {
        // split-bf16 image of [tap][N][Cin]: every 32 consecutive k become [32 hi | 32 lo]
        for (size_t i = 0; i + 3 < floats; i += 4)
            msr_store_split4(t.data() + (i & ~(size_t)31), (int)(i & 31), host[i], host[i + 1], host[i + 2], host[i + 3]);
        return upload(h, key, t.data(), floats);
    }
    uint16_t* o = reinterpret_cast<uint16_t*>(t.data());
    const int chunks = Cin / 32, nt32 = N / 32;
    for (int tap = 0; tap < taps; ++tap)
        for (int cc = 0; cc < chunks; ++cc)
            for (int nt = 0; nt < nt32; ++nt) {
                uint16_t* blk = o + (((size_t)tap * chunks + cc) * nt32 + nt) * 2048;   // 1024 floats
                for (int kg = 0; kg < 2; ++kg)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int j = lane & 31, hh = lane >> 5;
                        const float* src = host + ((size_t)tap * N + nt * 32 + j) * Cin + cc * 32 + kg * 16 + hh * 8;
                        uint16_t* hi = blk + ((kg * 2 + 0) * 64 + lane) * 8;
                        uint16_t* lo = blk + ((kg * 2 + 1) * 64 + lane) * 8;
                        for (int e = 0; e < 8; ++e) {
                            unsigned a, b2;
                            msr_split_bf16(src[e], a, b2);
                            hi[e] = (uint16_t)a; lo[e] = (uint16_t)b2;
                        }
                    }
            }
    return upload(h, key, t.data(), floats);
}

// Effective per-parity taps of the head kernel, weff[py][px][dy + 1][dx + 1][C] (zero where a tap does not exist):
//  * Conv2D(1, 4, 'same') applied to a nearest-2x up-sampled tensor (networks.py:54-56), kernel HWIO [4,4,C,1]: TF SAME for
//    k = 4 pads 1 before / 2 after; output parity p reads up-sampled rows 2y + p - 1 + kh, i.e. half-resolution offsets
//    {-1, 0, 0, +1} (p = 0) or {0, 0, +1, +1} (p = 1) for kh = 0..3 — taps that land on the same pixel are summed;
//  * Conv2DTranspose(1, 4, strides 2, 'same') (pix2pix.py:53-57), kernel [4,4,1,C]: four stride-1 2 x 2 convolutions, one per
//    output parity: out[2y + py][2x + px] = sum_{t,u} in[y - 1 + py + t][x - 1 + px + u] * W[kmap(py,t)][kmap(px,u)],
//    kmap(0, .) = {3, 1}, kmap(1, .) = {2, 0}.
std::vector<float> head_weff_upconv(const float* k44c, int C) {
    std::vector<float> weff((size_t)36 * C, 0.f);
    auto dmap = [](int parity, int k) { return parity == 0 ? (k == 0 ? 0 : k == 3 ? 2 : 1) : (k < 2 ? 1 : 2); };
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px)
            for (int kh = 0; kh < 4; ++kh)
                for (int kw = 0; kw < 4; ++kw) {
                    float* dst = &weff[((((size_t)py * 2 + px) * 3 + dmap(py, kh)) * 3 + dmap(px, kw)) * C];
                    const float* src = k44c + ((size_t)kh * 4 + kw) * C;
                    for (int c = 0; c < C; ++c) dst[c] += src[c];
                }
    return weff;
}
std::vector<float> head_weff_transpose(const float* k44c, int C) {
    static const int kmap[2][2] = {{3, 1}, {2, 0}};
    std::vector<float> weff((size_t)36 * C, 0.f);
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px)
            for (int t = 0; t < 2; ++t)
                for (int u = 0; u < 2; ++u) {
                    const float* src = k44c + ((size_t)kmap[py][t] * 4 + kmap[px][u]) * C;
                    std::copy(src, src + C, &weff[((((size_t)py * 2 + px) * 3 + (py + t)) * 3 + (px + u)) * C]);
                }
    return weff;
}

struct ConvVariant { int tile; int wt_frag; int ksplit; };   // ksplit 0: conv_pick_ksplit decides (small tiles)
ConvVariant pick_conv_variant(int B, int rout, int N, int stride, int epi, int prec, int cin);

// MSR_FLAG_FP8: a 3x3 stride-1 conv runs the fp8 form of the persistent ping-pong kernel when it fills the chip with
// whole tiles (B * (r/16)^2 * (N/128) >= 256, no K split); its input then holds one byte per channel: 128 channels (one
// 128-byte chunk) or a multiple of 256 (chunk pairs).
bool conv_fills_pp(int B, int rout, int N) {
    return rout >= 16 && N % 128 == 0 && (long)B * (rout / 16) * (rout / 16) * (N / 128) >= 256;
}
// channels of an fp8 tensor: 128 (one chunk: the two-tiles-per-body form of the kernel) or a multiple of 256 (chunk pairs)
int fp8_pad(int cin) { return cin <= 128 ? 128 : (cin + 255) / 256 * 256; }
// gamma|beta conv of a SPADE layer normalising C channels at resolution r
bool gb_uses_fp8(msr_handle* h, int rout, int C) { return h->fp8 && conv_fills_pp(h->B, rout, 2 * C); }
// main conv cin -> cout fed by that SPADE layer (its producer must be a whole-tile ping-pong launch too: its epilogue
// is the one that writes bf8)
bool main_uses_fp8(msr_handle* h, int rout, int cin, int cout) {
    return h->fp8 && conv_fills_pp(h->B, rout, cout) && conv_fills_pp(h->B, rout, 2 * cin);
}

int upload(msr_handle* h, const std::string& key, const float* host, size_t floats);
// MSR_FLAG_F16C: same coverage rule as the fp8 mode (whole-tile ping-pong launches; a main conv only if the gamma|beta
// conv that writes its input is one too)
// Since the second half of round 2 the K-range launches of the ping-pong kernel (fewer tiles than CUs: rb2 at the BASELINE
// sizes, most layers of a B = 1 call) run the f16c form too, their split-K epilogue writes the f16c image
// (MSR_F16C_KSPLIT=0: the earlier rule, whole-tile launches only).
int pp_ksplit(int B, int rout, int N, int stride, int cin, long min_items);
bool conv_on_pp_f16c(int B, int rout, int N, int cin) {
    static const bool ks_off = std::getenv("MSR_F16C_KSPLIT") && std::atoi(std::getenv("MSR_F16C_KSPLIT")) == 0;
    if (ks_off) return conv_fills_pp(B, rout, N);
    return pp_ksplit(B, rout, N, 1, cin, 128) >= 1;
}
bool gb_uses_f16c(msr_handle* h, int rout, int C) { return h->f16c && conv_on_pp_f16c(h->B, rout, 2 * C, 128); }
bool main_uses_f16c(msr_handle* h, int rout, int cin, int cout) {
    return h->f16c && cin % 64 == 0 && conv_on_pp_f16c(h->B, rout, cout, cin) && conv_on_pp_f16c(h->B, rout, 2 * cin, 128);
}

// conv_gb_resident (conv_gbr.hip) takes a SPADE layer whose gamma|beta conv AND consumer conv run f16c, when the layer has
// enough 16 x 16 pixel tiles x channel-block ranges to fill the chip (conv_gbr_ranges; MSR_GBR=0 switches it off).  Its
// weights are the f16c6 image with the input channels of every 32-chunk in the kernel's position order (GBR_PERM below).
bool main_uses_f16c(msr_handle* h, int rout, int cin, int cout);
bool main_uses_f16c6(msr_handle* h, int rout, int cin, int cout);
bool gb_uses_gbr(msr_handle* h, int rout, int C, int cout) {
    return h->f16c && gb_uses_f16c(h, rout, C) && main_uses_f16c(h, rout, C, cout) && !main_uses_f16c6(h, rout, C, cout) &&
           conv_gbr_ranges(h->B, rout, 2 * C) > 0;
}
// position e of a 32-channel chunk holds channel GBR_PERM(e): the order in which phase 1 of conv_gb_resident leaves a pixel's
// channels in a lane (32 x 32 MFMA rows 8q + 4h + r, halves interleaved by v_cvt_scalef32_2xpk16_fp6_f32)
inline int gbr_perm(int e) { return 8 * (e >> 3) + 4 * (e & 1) + ((e >> 1) & 3); }

// PREC_F16C6 (fp6 cross terms, kernels.h), OPT-IN with MSR_F16C_FP6=1: the main convs that run the stream kernel (conv_sw.hip:
// whole tiles, Cin % 128 == 0) behind a gamma|beta conv that is a whole-tile ping-pong launch (its LDS-assembled epilogue
// writes the fp6 image).  Measured: the consumer gains 6.5 % on those convs, the producer's block-scale and 6-bit
// packing cost the gamma|beta epilogues more, net -1 % per call — it pays only once the gamma|beta convs consume fp6 too.
// Any MSR_F16C_SW other than 1 (the A/B dispatches) switches it off.
bool main_uses_f16c6(msr_handle* h, int rout, int cin, int cout) {
    static const bool off = !(std::getenv("MSR_F16C_FP6") && std::atoi(std::getenv("MSR_F16C_FP6")) == 1) ||
                            (std::getenv("MSR_F16C_SW") && std::atoi(std::getenv("MSR_F16C_SW")) != 1);
    return !off && main_uses_f16c(h, rout, cin, cout) && cin % 128 == 0 && conv_fills_pp(h->B, rout, cout) &&
           conv_fills_pp(h->B, rout, 2 * cin);
}

// f16c6 image of [taps][N][Cin] weights (kernels.h PREC_F16C6): per 32-channel chunk [32 x hi f16 | 24 B l6 | e8m0 | 0.. |
// 24 B h6 | e8m0 | 0..], one power-of-two scale per output channel and piece (2^E >= max / 7.5)
std::vector<float> build_f16c6_image(const float* host, int taps, int N, int Cin) {
    std::vector<float> img((size_t)taps * N * Cin, 0.f);
    auto pow2exp = [](float amax) {
        if (!(amax > 0.f)) return 0;
        int fe;
        const float m = std::frexp(amax / 7.5f, &fe);       // amax / 7.5 = m * 2^fe, m in [0.5, 1)
        return std::max(-100, std::min(100, m == 0.5f ? fe - 1 : fe));
    };
    for (int n = 0; n < N; ++n) {
        float ah = 0.f, al = 0.f;
        for (int t = 0; t < taps; ++t)
            for (int k = 0; k < Cin; ++k) {
                const float w = host[((size_t)t * N + n) * Cin + k];
                const float hi = (float)(_Float16)w;
                ah = std::max(ah, std::fabs(w));
                al = std::max(al, std::fabs(w - hi));
            }
        const int eh = pow2exp(ah), el = pow2exp(al);
        const float ih = std::ldexp(1.f, -eh), il = std::ldexp(1.f, -el);
        for (int t = 0; t < taps; ++t)
            for (int c0 = 0; c0 < Cin; c0 += 32) {
                unsigned char* chunk = reinterpret_cast<unsigned char*>(img.data() + ((size_t)t * N + n) * Cin + c0);
                unsigned long long bl[3] = {0, 0, 0}, bh[3] = {0, 0, 0};      // 192-bit little-endian strings
                for (int c = 0; c < 32; ++c) {
                    const float w = host[((size_t)t * N + n) * Cin + c0 + c];
                    const _Float16 hi = (_Float16)w;
                    reinterpret_cast<_Float16*>(chunk)[c] = hi;
                    const unsigned long long cl = msr_f32_to_e2m3((w - (float)hi) * il), ch = msr_f32_to_e2m3(w * ih);
                    const int pos = 6 * c;
                    bl[pos / 64] |= cl << (pos % 64);
                    if (pos % 64 > 58) bl[pos / 64 + 1] |= cl >> (64 - pos % 64);
                    bh[pos / 64] |= ch << (pos % 64);
                    if (pos % 64 > 58) bh[pos / 64 + 1] |= ch >> (64 - pos % 64);
                }
                std::memcpy(chunk + 64, bl, 24);
                chunk[88] = (unsigned char)(127 + el);
                std::memcpy(chunk + 96, bh, 24);
                chunk[120] = (unsigned char)(127 + eh);
            }
    }
    return img;
}
int upload_conv_weight_f16c6(msr_handle* h, const std::string& key, const float* host, int taps, int N, int Cin) {
    if (Cin % 32) return fail(h, MSR_ERR_INVALID, "%s: f16c6 needs Cin %% 32 == 0", key.c_str());
    const std::vector<float> img = build_f16c6_image(host, taps, N, Cin);
    return upload(h, key, img.data(), img.size());
}

// The weight stream of conv_gb_resident (conv_gbr.hip): the f16c6 image of [9][N][128] (input channels of every 32-chunk in
// the kernel's position order, gbr_perm) re-ordered into the order the kernel's waves load it — for channel block nt, wave q,
// tap pair P (K-steps 2P, 2P + 1 of the 36-step chunk-major sequence: step T = chunk T / 9, tap T % 9), column block j, piece
// (0 / 1: fp16 fragment of the even / odd step, 2 / 3: first / second 16 bytes of the lane's fp6 piece), lane: 16 bytes.
// Lane (px, cg): row = 128 nt + 64 (q >> 1) + 16 (q & 1) + 32 j + px; fp16 fragment = bytes 16 cg .. of the record; fp6 piece =
// bytes 64 + 32 (cg & 1) .. of the even step's record (cg < 2) or the odd step's (cg >= 2).
std::vector<float> gbr_weight_stream(const float* w_tap_n_k, int N) {
    const int Cin = 128;
    std::vector<float> perm((size_t)9 * N * Cin);
    for (size_t row = 0; row < (size_t)9 * N; ++row)
        for (int k = 0; k < Cin; ++k) perm[row * Cin + k] = w_tap_n_k[row * Cin + (k & ~31) + gbr_perm(k & 31)];
    const std::vector<float> img = build_f16c6_image(perm.data(), 9, N, Cin);
    const unsigned char* src = reinterpret_cast<const unsigned char*>(img.data());
    std::vector<float> out(img.size());
    unsigned char* dst = reinterpret_cast<unsigned char*>(out.data());
    auto rec = [&](int T, int row) { return src + (((size_t)(T % 9) * N + row) * 4 + T / 9) * 128; };
    for (int nt = 0; nt < N / 128; ++nt)
        for (int q = 0; q < 4; ++q)
            for (int P = 0; P < 18; ++P)
                for (int j = 0; j < 2; ++j)
                    for (int piece = 0; piece < 4; ++piece)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int px = lane & 15, cg = lane >> 4;
                            const int row = 128 * nt + 64 * (q >> 1) + 16 * (q & 1) + 32 * j + px;
                            const unsigned char* s;
                            if (piece < 2) s = rec(2 * P + piece, row) + 16 * cg;
                            else s = rec(2 * P + (cg >> 1), row) + 64 + 32 * (cg & 1) + 16 * (piece - 2);
                            std::memcpy(dst + ((((((size_t)nt * 4 + q) * 18 + P) * 2 + j) * 4 + piece) * 64 + lane) * 16, s, 16);
                        }
    return out;
}

// f16c image of [taps][N][Cin] weights (kernels.h PREC_F16C): per 32-channel chunk [32 x hi f16 | 32 x l8 | 32 x h8]
// with hi = f16_rn(w), l8 = e4m3((w - hi) * 2^-el), h8 = e4m3(w * 2^-eh), el / eh powers of two
// per output channel; key + ".wexp"[n] = (127 + el) | (127 + eh) << 8
int upload_conv_weight_f16c(msr_handle* h, const std::string& key, const float* host, int taps, int N, int Cin) {
    if (Cin % 32) return fail(h, MSR_ERR_INVALID, "%s: f16c needs Cin %% 32 == 0", key.c_str());
    std::vector<float> img((size_t)taps * N * Cin);
    std::vector<int> wexp(N);
    auto pow2exp = [](float amax) {
        if (!(amax > 0.f)) return 0;
        int fe;
        (void)std::frexp(amax / 448.f, &fe);
        return std::max(-100, std::min(100, fe));
    };
    for (int n = 0; n < N; ++n) {
        float ah = 0.f, al = 0.f;
        for (int t = 0; t < taps; ++t)
            for (int k = 0; k < Cin; ++k) {
                const float w = host[((size_t)t * N + n) * Cin + k];
                const float hi = (float)(_Float16)w;
                ah = std::max(ah, std::fabs(w));
                al = std::max(al, std::fabs(w - hi));
            }
        const int eh = pow2exp(ah), el = pow2exp(al);
        const float ih = std::ldexp(1.f, -eh), il = std::ldexp(1.f, -el);
        wexp[n] = (127 + el) | ((127 + eh) << 8);
        for (int t = 0; t < taps; ++t)
            for (int c0 = 0; c0 < Cin; c0 += 32) {
                unsigned char* chunk = reinterpret_cast<unsigned char*>(img.data() + ((size_t)t * N + n) * Cin + c0);
                for (int c = 0; c < 32; ++c) {
                    const float w = host[((size_t)t * N + n) * Cin + c0 + c];
                    const _Float16 hi = (_Float16)w;
                    reinterpret_cast<_Float16*>(chunk)[c] = hi;
                    chunk[64 + c] = msr_f32_to_e4m3((w - (float)hi) * il);
                    chunk[96 + c] = msr_f32_to_e4m3(w * ih);
                }
            }
    }
    int rc = upload(h, key, img.data(), img.size());
    if (rc) return rc;
    return upload(h, key + ".wexp", reinterpret_cast<const float*>(wexp.data()), wexp.size());
}

// fp8 e4m3 image of [taps][N][Cin] weights: bytes [taps][N][fp8_pad(Cin)] (zero padded), a power-of-two scale per output
// channel chosen so that the largest |w| of the channel lands in e4m3's top binade, its e8m0 exponent replicated in
// the four bytes of key + ".wexp"[n]
int upload_conv_weight_fp8(msr_handle* h, const std::string& key, const float* host, int taps, int N, int Cin) {
    const int cp = fp8_pad(Cin);
    std::vector<unsigned char> q((size_t)taps * N * cp, 0);
    std::vector<int> wexp(N);
    for (int n = 0; n < N; ++n) {
        float amax = 0.f;
        for (int t = 0; t < taps; ++t)
            for (int k = 0; k < Cin; ++k) amax = std::max(amax, std::fabs(host[((size_t)t * N + n) * Cin + k]));
        int e = 0;
        if (amax > 0.f) {
            int fe;
            (void)std::frexp(amax / 448.f, &fe);      // amax / 448 = m * 2^fe, m in [0.5, 1): 2^fe >= amax / 448
            e = fe;
        }
        e = std::max(-100, std::min(100, e));
        const float inv = std::ldexp(1.f, -e);
        for (int t = 0; t < taps; ++t)
            for (int k = 0; k < Cin; ++k)
                q[((size_t)t * N + n) * cp + k] = msr_f32_to_e4m3(host[((size_t)t * N + n) * Cin + k] * inv);
        const unsigned b = (unsigned)(127 + e);
        wexp[n] = (int)(b | (b << 8) | (b << 16) | (b << 24));
    }
    int rc = upload(h, key, reinterpret_cast<const float*>(q.data()), q.size() / 4);
    if (rc) return rc;
    return upload(h, key + ".wexp", reinterpret_cast<const float*>(wexp.data()), wexp.size());
}

// Output resolution of the conv a weight belongs to ("enc.ds3.kernel" -> S>>3, "gen.rb4...." -> sw<<3).
bool weight_conv_shape(msr_handle* h, const std::string& name, int* rout, int* stride) {
    int i = 0;
    *stride = 1;
    if (std::sscanf(name.c_str(), "enc.ds%d.", &i) == 1) { *rout = h->S >> i; *stride = 2; return true; }
    if (std::sscanf(name.c_str(), "gen.rb%d.", &i) == 1) { *rout = (h->S / 64) << (i - 1); return true; }
    return false;
}
bool weight_uses_frag(msr_handle* h, const std::string& name, int N, int epi, int cin) {
    if (h->prec != PREC_BF16X3) return false;
    int rout = 0, stride = 1;
    if (!weight_conv_shape(h, name, &rout, &stride)) return false;
    return pick_conv_variant(h->B, rout, N, stride, epi, h->prec, cin).wt_frag != 0;
}
// MSR_FLAG_GB_F16X2: the gamma|beta convs that run the persistent ping-pong kernel take 2-term fp16 products; the
// planner (conv precision, format of the mask embedding that feeds them) and the weight upload both ask this.
bool gb_uses_f16x2(msr_handle* h, int rout, int N, int cin) {
    if (!h->gb_f16x2) return false;
    const ConvVariant v = pick_conv_variant(h->B, rout, N, 1, EPI_SPADE, PREC_BF16X3, cin);
    return v.tile == TILE_256x128_PP && v.ksplit == 1;      // the K-split launches run the 3-term form
}

}  // namespace

// ================================================================================================
extern "C" {

int msr_abi_version(void) { return MSR_ABI_VERSION; }

uint32_t msr_crc32c(const void* host_data, uint64_t n, uint32_t crc) {
    static uint32_t table[8][256];
    static bool ready = false;
    if (!ready) {   // slicing-by-8 tables, reflected polynomial 0x82F63B78
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
            table[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int t = 1; t < 8; ++t) table[t][i] = (table[t - 1][i] >> 8) ^ table[0][table[t - 1][i] & 0xFF];
        ready = true;
    }
    const uint8_t* p = static_cast<const uint8_t*>(host_data);
    uint32_t c = crc ^ 0xFFFFFFFFu;
    while (n >= 8) {
        uint32_t lo, hi;
        std::memcpy(&lo, p, 4);
        std::memcpy(&hi, p + 4, 4);
        lo ^= c;
        c = table[7][lo & 0xFF] ^ table[6][(lo >> 8) & 0xFF] ^ table[5][(lo >> 16) & 0xFF] ^ table[4][lo >> 24] ^
            table[3][hi & 0xFF] ^ table[2][(hi >> 8) & 0xFF] ^ table[1][(hi >> 16) & 0xFF] ^ table[0][hi >> 24];
        p += 8;
        n -= 8;
    }
    while (n--) c = table[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

const char* msr_last_error(const msr_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int msr_create(const msr_config* cfg, msr_handle** out) {
    if (!cfg || !out) return fail(nullptr, MSR_ERR_INVALID, "msr_create: null argument");
    *out = nullptr;
    const int S = cfg->image_size, B = cfg->batch_size;
    if (cfg->variant < MSR_GAUGAN || cfg->variant > MSR_PIX2PIX)
        return fail(nullptr, MSR_ERR_INVALID, "unknown variant %d", cfg->variant);
    if (cfg->variant == MSR_PIX2PIX) {
        if (S != 256) return fail(nullptr, MSR_ERR_INVALID, "pix2pix input is fixed to 256x256 (pix2pix.py:7), got %d", S);
    } else {
        if (S < 64 || (S & (S - 1)))
            return fail(nullptr, MSR_ERR_INVALID, "image_size must be a power of two >= 64, got %d", S);
        if (cfg->latent_dim <= 0 || cfg->latent_dim % 4)
            return fail(nullptr, MSR_ERR_INVALID, "latent_dim must be a positive multiple of 4, got %d", cfg->latent_dim);
    }
    if (B < 1 || B > 16) return fail(nullptr, MSR_ERR_INVALID, "batch_size must be in [1,16], got %d", B);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(nullptr, MSR_ERR_DEVICE, "no HIP device visible: libmoonsr_hip has no CPU fallback");
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(nullptr, MSR_ERR_DEVICE, "device %d out of range (%d visible)", cfg->device, ndev);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess)
        return fail(nullptr, MSR_ERR_DEVICE, "hipGetDeviceProperties failed");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, MSR_ERR_DEVICE, "device %d is %s; this library is built for gfx950 (MI355X) only",
                    cfg->device, prop.gcnArchName);
    if (hipSetDevice(cfg->device) != hipSuccess) return fail(nullptr, MSR_ERR_DEVICE, "hipSetDevice failed");
#ifdef MSR_DIAG_BUILD   // stamp / what-if object (kernels.h): never the product
    if (!(std::getenv("MSR_ALLOW_DIAG_BUILD") && std::atoi(std::getenv("MSR_ALLOW_DIAG_BUILD")) == 1))
        return fail(nullptr, MSR_ERR_STATE, "this libmoonsr_hip.so is a diagnostic build (-DMSR_DIAG_BUILD: in-kernel stamps "
                    "or what-if switches that change results); set MSR_ALLOW_DIAG_BUILD=1 to use it for measurements");
#endif
    if (conv_igemm_init() != hipSuccess || conv_gbr_init() != hipSuccess)
        return fail(nullptr, MSR_ERR_DEVICE, "could not set the dynamic-LDS attribute of the conv kernels");
    auto h = std::make_unique<msr_handle>();
    h->cfg = *cfg;
    h->S = S; h->B = B; h->L = cfg->latent_dim; h->variant = cfg->variant;
    h->prec = (cfg->flags & MSR_FLAG_BF16X3) ? PREC_BF16X3 : PREC_F32;
    h->gb_f16x2 = h->prec == PREC_BF16X3 && (cfg->flags & MSR_FLAG_GB_F16X2);
    if ((cfg->flags & MSR_FLAG_GB_F16X2) && !(cfg->flags & MSR_FLAG_BF16X3))
        return fail(nullptr, MSR_ERR_INVALID, "MSR_FLAG_GB_F16X2 needs MSR_FLAG_BF16X3");
    if ((cfg->flags & MSR_FLAG_FP8) && (!(cfg->flags & MSR_FLAG_BF16X3) || (cfg->flags & MSR_FLAG_GB_F16X2)))
        return fail(nullptr, MSR_ERR_INVALID, "MSR_FLAG_FP8 goes with MSR_FLAG_BF16X3 alone (the layers it does not cover run bf16x3)");
    h->fp8 = (cfg->flags & MSR_FLAG_FP8) && cfg->variant != MSR_PIX2PIX;
    if ((cfg->flags & MSR_FLAG_F16C) && (!(cfg->flags & MSR_FLAG_BF16X3) || (cfg->flags & (MSR_FLAG_GB_F16X2 | MSR_FLAG_FP8))))
        return fail(nullptr, MSR_ERR_INVALID, "MSR_FLAG_F16C goes with MSR_FLAG_BF16X3 alone (the layers it does not cover run bf16x3)");
    h->f16c = (cfg->flags & MSR_FLAG_F16C) && cfg->variant != MSR_PIX2PIX;
    if ((cfg->flags & MSR_FLAG_F16_MAIN) && !(cfg->flags & MSR_FLAG_F16C))
        return fail(nullptr, MSR_ERR_INVALID, "MSR_FLAG_F16_MAIN modifies MSR_FLAG_F16C");
    h->f16m = h->f16c && (cfg->flags & MSR_FLAG_F16_MAIN);
    if (cfg->variant == MSR_PIX2PIX) { h->prec = PREC_F32; h->gb_f16x2 = false; }   // the parity config runs on the fp32 MFMA
    build_specs(h.get());
    *out = h.release();
    return MSR_OK;
}

int msr_destroy(msr_handle* h) {
    if (!h) return MSR_OK;
    hipSetDevice(h->cfg.device);
    hipDeviceSynchronize();
    for (auto& g : h->graphs) {
        if (g.exec) hipGraphExecDestroy(g.exec);
        if (g.graph) hipGraphDestroy(g.graph);
    }
    for (auto& kv : h->dev) hipFree(kv.second);
    if (h->mom_partial) hipFree(h->mom_partial);
    if (h->dense_partial) hipFree(h->dense_partial);
    if (h->conv_partial) hipFree(h->conv_partial);
    if (h->stat_ws) hipFree(h->stat_ws);
    if (h->window) hipFree(h->window);
    if (h->stitch_grid) hipFree(h->stitch_grid);
    for (auto e : h->ev_pool) hipEventDestroy(e);
    for (auto& op : h->ops)
        if (op.done) hipEventDestroy(op.done);
    if (h->ev_fork) hipEventDestroy(h->ev_fork);
    if (h->aux) hipStreamDestroy(h->aux);
    delete h;
    return MSR_OK;
}

int msr_weight_count(const msr_handle* h, int32_t* expected, int32_t* loaded) {
    if (!h) return MSR_ERR_INVALID;
    int l = 0;
    for (auto& s : h->specs) l += s.loaded;
    if (expected) *expected = (int32_t)h->specs.size();
    if (loaded) *loaded = l;
    return MSR_OK;
}

const char* msr_weight_name(const msr_handle* h, int32_t i, int64_t* shape4, int32_t* rank) {
    if (!h || i < 0 || i >= (int)h->specs.size()) return nullptr;
    const auto& s = h->specs[i];
    if (rank) *rank = (int32_t)s.shape.size();
    if (shape4)
        for (size_t k = 0; k < s.shape.size() && k < 4; ++k) shape4[k] = s.shape[k];
    return s.name.c_str();
}

static bool ends_with(const std::string& s, const char* suf) {
    const size_t n = std::strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

int msr_load_weight(msr_handle* h, const char* name_c, const float* host, const int64_t* shape, int32_t rank) {
    if (!h) return MSR_ERR_INVALID;
    if (!name_c || !host || !shape) return fail(h, MSR_ERR_INVALID, "msr_load_weight: null argument");
    const std::string name = name_c;
    auto it = h->spec_index.find(name);
    if (it == h->spec_index.end()) return fail(h, MSR_ERR_INVALID, "unexpected weight name '%s'", name_c);
    WeightSpec& sp = h->specs[it->second];
    if (rank != (int)sp.shape.size()) return fail(h, MSR_ERR_INVALID, "%s: rank %d, expected %zu", name_c, rank, sp.shape.size());
    size_t count = 1;
    for (int k = 0; k < rank; ++k) {
        if (shape[k] != sp.shape[k])
            return fail(h, MSR_ERR_INVALID, "%s: dim %d is %lld, expected %lld", name_c, k, (long long)shape[k],
                        (long long)sp.shape[k]);
        count *= (size_t)shape[k];
    }
    HIPCHK(h, hipSetDevice(h->cfg.device));
    h->planned = false;
    int rc = MSR_OK;
    const auto& s = sp.shape;
    if (name.rfind("p2p.", 0) == 0) {
        // Conv2DTranspose(k=4, s=2, 'same') is four stride-1 2x2 convolutions, one per output parity (py, px):
        // out[2y+py][2x+px] = sum_{t,u} in[y-1+py+t][x-1+px+u] * W[kmap(py,t)][kmap(px,u)], kmap(0,.) = {3,1},
        // kmap(1,.) = {2,0}  (from kh = o + 1 - 2i, the transpose of the 'same' stride-2 forward conv).
        static const int kmap[2][2] = {{3, 1}, {2, 0}};
        if (name == "p2p.down1.kernel") {
            rc = upload(h, name, host, count);                       // conv_direct reads HWIO
        } else if (name == "p2p.last.kernel") {
            // [4,4,1,C] -> the head kernel's effective taps weff[py][px][dy][dx][C], offset dy-1 = py+t-1
            const std::vector<float> weff = head_weff_transpose(host, (int)s[3]);
            rc = upload(h, "p2p.last.weff", weff.data(), weff.size());
        } else if (ends_with(name, ".kernel") && name.find(".down") == std::string::npos) {
            // [kh,kw,Cout,Cin] is already K-contiguous per output channel: four parity images [2x2 taps][Cout][Cin]
            const size_t co = (size_t)s[2], ci = (size_t)s[3];
            std::vector<float> t4(count);
            for (int py = 0; py < 2; ++py)
                for (int px = 0; px < 2; ++px)
                    for (int t = 0; t < 2; ++t)
                        for (int u = 0; u < 2; ++u) {
                            const float* src = host + ((size_t)kmap[py][t] * 4 + kmap[px][u]) * co * ci;
                            std::copy(src, src + co * ci, t4.data() + ((size_t)(py * 2 + px) * 4 + t * 2 + u) * co * ci);
                        }
            rc = upload(h, name, t4.data(), count);
        } else if (ends_with(name, ".kernel")) {
            // down2..8: HWIO -> [tap][Cout][Cin]
            std::vector<float> t(count);
            hwio_to_tap_oc_ic(host, t.data(), 16, (int)s[2], (int)s[3], (int)s[3], nullptr);
            rc = upload(h, name, t.data(), count);
        } else {
            h->host_small[name].assign(host, host + count);   // BN statistics / bias: folded at plan time
        }
    } else if (name == "enc.ds1.kernel" || ends_with(name, ".conv.kernel") || ends_with(name, ".conv.bias") ||
               ends_with(name, ".in.gamma") || ends_with(name, ".in.beta") || name == "gen.dense.kernel" ||
               name == "gen.dense.bias" || (name.find(".conv_") != std::string::npos && ends_with(name, ".bias") &&
                                            name.find("spade") == std::string::npos)) {
        rc = upload(h, name, host, count);   // used in the reference layout
        if (!rc && ends_with(name, ".conv.kernel") && name.find(".spade_") != std::string::npos) {
            // conv_gb_resident multiplies the mask embedding on the fp16 MFMA: its A operands (three fp16 terms per product)
            std::vector<float> e16(4096);
            conv_gbr_embed_image(host, e16.data());
            rc = upload(h, name + ".e16", e16.data(), e16.size());
        }
    } else if (name == "enc.mean.kernel" || name == "enc.variance.kernel") {
        // concatenate the two heads into one [K, 2L] matrix so the flatten is streamed once
        float* d = nullptr;
        rc = dev_alloc(h, "enc.heads.kernel", (size_t)s[0] * 2 * h->L, false, &d);
        if (!rc) {
            const size_t coff = name == "enc.mean.kernel" ? 0 : (size_t)h->L;
            HIPCHK(h, hipMemcpy2D(d + coff, (size_t)2 * h->L * sizeof(float), host, (size_t)h->L * sizeof(float),
                                  (size_t)h->L * sizeof(float), (size_t)s[0], hipMemcpyHostToDevice));
        }
    } else if (name == "enc.mean.bias" || name == "enc.variance.bias") {
        float* d = nullptr;
        rc = dev_alloc(h, "enc.heads.bias", (size_t)2 * h->L, false, &d);
        if (!rc) HIPCHK(h, hipMemcpy(d + (name == "enc.mean.bias" ? 0 : h->L), host, h->L * sizeof(float), hipMemcpyHostToDevice));
    } else if (name == "gen.head.kernel") {
        // effective per-parity taps of Conv2D(1,4,'same') applied to a nearest-2x up-sampled tensor
        const std::vector<float> weff = head_weff_upconv(host, (int)s[2]);
        rc = upload(h, "gen.head.weff", weff.data(), weff.size());
    } else if (name == "gen.head.bias") {
        h->host_small[name].assign(host, host + count);
    } else if (ends_with(name, ".conv_gamma.kernel") || ends_with(name, ".conv_beta.kernel")) {
        // gamma and beta convs share their input: ONE GEMM with N = 2C whose columns interleave
        // (32 gamma channels | 32 beta channels) so a wave holds both for the same pixel and channel.
        const bool is_beta = ends_with(name, ".conv_beta.kernel");
        const int cin = (int)s[2], C = (int)s[3];
        const std::string base = name.substr(0, name.rfind(".conv_"));
        {
            std::vector<int> rowmap(C);
            for (int c = 0; c < C; ++c) rowmap[c] = (c / 32) * 64 + (is_beta ? 32 : 0) + (c % 32);
            // stage through a host image of the combined tensor; the other half is filled by the sibling call
            std::vector<float>& img = h->host_small[base + ".gb.kernel"];
            img.resize((size_t)9 * 2 * C * cin);
            hwio_to_tap_oc_ic(host, img.data(), 9, cin, C, 2 * C, rowmap.data());
            int rout = 0, stride = 1;
            weight_conv_shape(h, name, &rout, &stride);
            int blk = 0, sj = 0;
            std::sscanf(name.c_str(), "gen.rb%d.spade_%d.", &blk, &sj);
            const int cout_main = blk >= 1 && blk <= 6 ? kGenFilters[blk - 1] : 0;     // every conv of block i has kGenFilters[i-1] outputs
            if (gb_uses_fp8(h, rout, C)) rc = upload_conv_weight_fp8(h, base + ".gb.kernel", img.data(), 9, 2 * C, cin);
            else if (cin == 128 && cout_main && gb_uses_gbr(h, rout, C, cout_main)) {
                const std::vector<float> ws = gbr_weight_stream(img.data(), 2 * C);
                rc = upload(h, base + ".gb.kernel", ws.data(), ws.size());
            }
            else if (gb_uses_f16c(h, rout, C)) rc = upload_conv_weight_f16c(h, base + ".gb.kernel", img.data(), 9, 2 * C, cin);
            else
                rc = upload_conv_weight(h, base + ".gb.kernel", img.data(), img.size(), 9, 2 * C, cin,
                                        weight_uses_frag(h, name, 2 * C, EPI_SPADE, cin), gb_uses_f16x2(h, rout, 2 * C, cin));
        }
    } else if (ends_with(name, ".conv_gamma.bias") || ends_with(name, ".conv_beta.bias")) {
        const bool is_beta = ends_with(name, ".conv_beta.bias");
        const int C = (int)s[0];
        const std::string base = name.substr(0, name.rfind(".conv_"));
        std::vector<float>& img = h->host_small[base + ".gb.bias"];
        img.resize((size_t)2 * C);
        for (int c = 0; c < C; ++c) img[(c / 32) * 64 + (is_beta ? 32 : 0) + (c % 32)] = host[c];
        rc = upload(h, base + ".gb.bias", img.data(), img.size());
    } else if (ends_with(name, ".kernel")) {
        // encoder ds2..5 and ResidualBlock conv_1/2/3: HWIO -> [tap][Cout][Cin]
        const int taps = (int)(s[0] * s[1]), cin = (int)s[2], cout = (int)s[3];
        std::vector<float> t(count);
        hwio_to_tap_oc_ic(host, t.data(), taps, cin, cout, cout, nullptr);
        int rout = 0, stride = 1;
        const bool gen_conv = name.rfind("gen.rb", 0) == 0 && weight_conv_shape(h, name, &rout, &stride);
        if (gen_conv && main_uses_fp8(h, rout, cin, cout)) rc = upload_conv_weight_fp8(h, name, t.data(), taps, cout, cin);
        else if (gen_conv && main_uses_f16c6(h, rout, cin, cout)) rc = upload_conv_weight_f16c6(h, name, t.data(), taps, cout, cin);
        else if (gen_conv && main_uses_f16c(h, rout, cin, cout)) rc = upload_conv_weight_f16c(h, name, t.data(), taps, cout, cin);
        else
            rc = upload_conv_weight(h, name, t.data(), count, taps, cout, cin, weight_uses_frag(h, name, cout, EPI_BIAS, cin));
    } else {
        rc = upload(h, name, host, count);
    }
    if (rc) return rc;
    sp.loaded = true;
    return MSR_OK;
}

}  // extern "C"

// ================================================================================================
// plan
// ================================================================================================
namespace {

struct Padded {   // zero-bordered NHWC activation [B, r+2, r+2, C]
    float* base = nullptr;
    int r = 0, C = 0;
    int py() const { return (r + 2) * C; }
    int pb() const { return (r + 2) * (r + 2) * C; }
    int interior() const { return py() + C; }
};

int alloc_padded(msr_handle* h, const std::string& key, int r, int C, Padded* out) {
    out->r = r; out->C = C;
    return dev_alloc(h, key, (size_t)h->B * (r + 2) * (r + 2) * C, true, &out->base);
}

// Kernel variant of one conv layer.  Under bf16x3 it also fixes the weight layout, so msr_load_weight and the
// planner must agree: both call this.
// K split of the persistent ping-pong kernel for layers with fewer 16 x 16 x 128 tiles than CUs: whole chunk pairs
// per range, a power of two, as many ranges as it takes to give every CU a work item.  0 = the layer is not one for
// that kernel (it needs stride 1, r >= 16, Cin % 64 == 0, an input below the 2 GiB buffer-descriptor range and, split
// or not, at least `min_items` work items — below that the small-tile split-K kernels are faster).
int pp_ksplit(int B, int rout, int N, int stride, int cin, long min_items = 128);
int pp_ksplit(int B, int rout, int N, int stride, int cin, long min_items) {
    static const bool off = std::getenv("MSR_PP_KSPLIT") && std::atoi(std::getenv("MSR_PP_KSPLIT")) == 0;
    if (stride != 1 || rout < 16 || cin % 64 || N % 128) return 0;
    if ((size_t)B * (rout + 2) * (rout + 2) * cin * sizeof(float) >= ((size_t)1 << 31)) return 0;
    const long tiles = (long)B * (rout / 16) * (rout / 16) * (N / 128);
    if (tiles >= 256) return 1;
    if (off) return 0;
    const int pairs = cin / 64;
    int ks = 1;
    while (tiles * ks * 2 <= 256 && pairs % (ks * 2) == 0) ks *= 2;
    return tiles * ks >= min_items ? ks : 0;
}

ConvVariant pick_conv_variant(int B, int rout, int N, int stride, int epi, int prec, int cin) {
    ConvVariant v;
    const int M = B * rout * rout;
    v.tile = conv_pick_tile(M, N, epi, prec, 9 * (cin / 32));
    v.wt_frag = 0;
    v.ksplit = 0;
    if (prec == PREC_BF16X3) {
        const long big_blocks = (long)((M + 127) / 128) * (N / 128);
        const int pks = pp_ksplit(B, rout, N, stride, cin);
        if (pks >= 1) {
            // LDS-staged input halo, 512-thread ping-pong form (one persistent workgroup per CU, 16 x 16 pixels x 128
            // channels per tile): 10-25 % faster than two 256-thread workgroups per CU as soon as it fills the chip
            // once; with fewer tiles than CUs, K ranges supply the work items (pks > 1).
            v.tile = TILE_256x128_PP;
            v.ksplit = pks;
        } else if (v.tile == TILE_64x64 || big_blocks < 256) {
            v.wt_frag = 1;   // few workgroups (with split-K): B fragments straight to VGPRs, +18 % on the small tile
        } else if (stride == 1 && rout >= 16 && cin % 64 == 0 &&
                   (size_t)B * (rout + 2) * (rout + 2) * cin * sizeof(float) < ((size_t)1 << 31)) {   // raw buffer loads: 2 GiB
            v.tile = TILE_128x128_HALO16;     // only reached with MSR_PP_KSPLIT=0: two 256-thread workgroups per CU
            v.ksplit = 1;
        }
    }
    return v;
}

Op conv_op(const Padded& in, const float* wt, const float* bias, int B, int rout, int N, int stride, int epi,
           int prec = PREC_F32) {
    Op op; op.type = OP_CONV; op.epi = epi;
    ConvParams& c = op.conv;
    c.in = stride == 1 ? in.base : in.base + in.interior();
    c.wt = wt; c.bias = bias;
    c.B = B; c.Hout = rout; c.Wout = rout; c.Cin = in.C; c.N = N;
    c.KH = 3; c.KW = 3; c.stride = stride;
    c.in_px = in.C; c.in_py = in.py(); c.in_pb = in.pb();
    c.slope = 0.2f;
    c.prec = prec;
    c.out_split = (epi == EPI_SPADE && prec == PREC_BF16X3) ? 1 : 0;   // a SPADE output always feeds a conv
    const ConvVariant cv = pick_conv_variant(B, rout, N, stride, epi, prec, in.C);
    op.tile = cv.tile;
    c.wt_frag = cv.wt_frag;
    c.ksplit = cv.ksplit > 0 ? cv.ksplit : conv_pick_ksplit(B * rout * rout, N, 9 * (in.C / 32), op.tile, prec);
    c.partial = nullptr;   // bound to the handle's workspace at launch
    op.flops = 2.0 * B * rout * rout * (double)in.C * N * 9;
    return op;
}

void set_out_dense(ConvParams& c, float* out, int r, int C) {
    c.out = out; c.out_px = C; c.out_py = r * C; c.out_pb = r * r * C; c.out_off = 0;
}
void set_out_padded(ConvParams& c, const Padded& p) {
    c.out = p.base; c.out_px = p.C; c.out_py = p.py(); c.out_pb = p.pb(); c.out_off = p.interior();
}
void set_aux_dense(ConvParams& c, const float* x, int rx, int C, int shift) {
    c.aux = x; c.aux_px = C; c.aux_py = rx * C; c.aux_pb = rx * rx * C; c.aux_shift = shift;
}

Op moments_op(const float* x, int G, int P, int C, float eps, float* mean, float* stdv) {
    Op op; op.type = OP_MOMENTS;
    op.mom = {x, G, P, C, eps, mean, stdv};
    op.bytes = (double)G * P * C * 4;
    return op;
}

// A split-K conv whose output feeds a normalisation takes the moments in its own epilogue (splitk_epilogue_mom_kernel).
bool fuse_moments_into_splitk(Op& cv, int G, float eps, float* mean, float* stdv) {
    if (cv.type != OP_CONV || cv.conv.ksplit <= 1 || (cv.epi != EPI_BIAS && cv.epi != EPI_RES) || cv.conv.N % 32)
        return false;
    static const bool off = std::getenv("MSR_FUSE_MOMENTS") && std::atoi(std::getenv("MSR_FUSE_MOMENTS")) == 0;
    if (off) return false;
    cv.conv.mom_mean = mean; cv.conv.mom_std = stdv; cv.conv.mom_eps = eps; cv.conv.mom_G = G;
    return true;
}

int plan_spade(msr_handle* h) {
    const int S = h->S, B = h->B, L = h->L;
    char n[160];
    int rc;
    auto need = [&](const std::string& k) -> float* { return D(h, k); };
    size_t mom_doubles = 0;
    auto mom_need = [&](int G, int P, int C) { mom_doubles = std::max(mom_doubles, (size_t)G * moments_chunks(G, P) * C * 2); };

    // ---------------- encoder (networks.py:8-34) ----------------
    Padded e_in;   // input of the next strided conv
    rc = alloc_padded(h, "ws.enc.p1", S / 2, 64, &e_in); if (rc) return rc;
    {
        Op op; op.type = OP_SMALLCIN; op.src_is_input = true;
        SmallCinParams& p = op.sc;
        p.w = need("enc.ds1.kernel"); p.bias = nullptr; p.out = e_in.base;
        p.B = B; p.S = S; p.Hout = S / 2; p.Cout = 64;
        p.ay = 2; p.cy = 0; p.lim = S; p.f = 1; p.o = 0;
        p.out_px = 64; p.out_py = e_in.py(); p.out_pb = e_in.pb(); p.out_off = e_in.interior();
        p.act = 2; p.slope = 0.2f;
        p.out_split = h->prec == PREC_BF16X3;
        op.flops = 2.0 * B * (S / 2) * (S / 2) * 18.0 * 64;
        h->ops.push_back(op);
    }
    float* flat = nullptr;
    const int rlast = S / 32;
    for (int i = 2; i <= 5; ++i) {
        const int c = kEncChannels[i - 1], r = S >> i;
        float* raw; float *mean, *stdv;
        snprintf(n, sizeof n, "ws.enc.raw%d", i); rc = dev_alloc(h, n, (size_t)B * r * r * c, false, &raw); if (rc) return rc;
        snprintf(n, sizeof n, "ws.enc.mean%d", i); rc = dev_alloc(h, n, (size_t)B * c, false, &mean); if (rc) return rc;
        snprintf(n, sizeof n, "ws.enc.std%d", i); rc = dev_alloc(h, n, (size_t)B * c, false, &stdv); if (rc) return rc;
        float* zero_bias; rc = dev_alloc(h, "ws.zero_bias", 2048, true, &zero_bias); if (rc) return rc;
        snprintf(n, sizeof n, "enc.ds%d.kernel", i);
        Op cv = conv_op(e_in, need(n), zero_bias, B, r, c, 2, EPI_BIAS, h->prec);
        set_out_dense(cv.conv, raw, r, c);
        const bool fused = fuse_moments_into_splitk(cv, B, 1e-3f, mean, stdv);
        h->ops.push_back(cv);
        if (!fused) h->ops.push_back(moments_op(raw, B, r * r, c, 1e-3f, mean, stdv));
        mom_need(B, r * r, c);
        Op na; na.type = OP_NORMACT;
        snprintf(n, sizeof n, "enc.ds%d.in.gamma", i); na.na.gamma = need(n);
        snprintf(n, sizeof n, "enc.ds%d.in.beta", i); na.na.beta = need(n);
        na.na.x = raw; na.na.mean = mean; na.na.stdv = stdv;
        na.na.B = B; na.na.H = r; na.na.W = r; na.na.C = c; na.na.slope = 0.2f;
        if (i < 5) {
            Padded nx;
            snprintf(n, sizeof n, "ws.enc.p%d", i); rc = alloc_padded(h, n, r, c, &nx); if (rc) return rc;
            na.na.out = nx.base; na.na.out_px = c; na.na.out_py = nx.py(); na.na.out_pb = nx.pb(); na.na.out_off = nx.interior();
            na.na.out_split = h->prec == PREC_BF16X3;
            e_in = nx;
        } else {
            rc = dev_alloc(h, "ws.enc.flat", (size_t)B * r * r * c, false, &flat); if (rc) return rc;
            na.na.out = flat; na.na.out_px = c; na.na.out_py = r * c; na.na.out_pb = r * r * c; na.na.out_off = 0;
        }
        na.bytes = 2.0 * B * r * r * c * 4;
        h->ops.push_back(na);
    }
    // Dense mean | variance (networks.py:32-33), then the sampler (sampling.py:16) or mean+variance (model.py:267)
    const int K = rlast * rlast * 512;
    float* mv; rc = dev_alloc(h, "ws.enc.mv", (size_t)B * 2 * L, false, &mv); if (rc) return rc;
    rc = dev_alloc(h, "ws.z", (size_t)B * L, false, &h->z); if (rc) return rc;
    size_t dense_part = dense_partial_floats(B, K, 2 * L);
    {
        Op op; op.type = OP_DENSE;
        op.dense = {flat, need("enc.heads.kernel"), need("enc.heads.bias"), mv, B, K, 2 * L};
        op.flops = 2.0 * B * K * 2.0 * L; op.bytes = (double)K * 2 * L * 4;
        h->ops.push_back(op);
        Op lt; lt.type = OP_LATENT; lt.eps_is_input = true;
        lt.lat = {mv, h->z, B, L, h->variant == MSR_GAUGAN ? 1 : 0};
        h->ops.push_back(lt);
    }
    // ---------------- generator (networks.py:37-57) ----------------
    const int sw = S / 64;
    const int N0 = sw * sw * 1024;
    float* x_prev; rc = dev_alloc(h, "ws.gen.x0", (size_t)B * N0, false, &x_prev); if (rc) return rc;
    dense_part = std::max(dense_part, dense_partial_floats(B, L, N0));
    {
        Op op; op.type = OP_DENSE;
        op.dense = {h->z, need("gen.dense.kernel"), need("gen.dense.bias"), x_prev, B, L, N0};
        op.flops = 2.0 * B * L * (double)N0; op.bytes = (double)L * N0 * 4;
        h->ops.push_back(op);
    }
    float *st_mean, *st_std;   // batch moments of the block input
    rc = dev_alloc(h, "ws.gen.mean_in0", 1024, false, &st_mean); if (rc) return rc;
    rc = dev_alloc(h, "ws.gen.std_in0", 1024, false, &st_std); if (rc) return rc;
    h->ops.push_back(moments_op(x_prev, 1, B * sw * sw, 1024, 1e-5f, st_mean, st_std));
    mom_need(1, B * sw * sw, 1024);

    int cin = 1024, r_prev = sw;
    for (int i = 1; i <= 6; ++i) {
        const int f = kGenFilters[i - 1];
        const int r = sw << (i - 1);
        const int shift = i > 1 ? 1 : 0;   // block input = UpSampling2D(previous output), folded into the index
        const bool learned = f != cin;
        float *x1, *skip = nullptr, *outb, *m1, *s1, *mo, *so;
        snprintf(n, sizeof n, "ws.gen.rb%d.x1", i); rc = dev_alloc(h, n, (size_t)B * r * r * f, false, &x1); if (rc) return rc;
        snprintf(n, sizeof n, "ws.gen.rb%d.out", i); rc = dev_alloc(h, n, (size_t)B * r * r * f, false, &outb); if (rc) return rc;
        snprintf(n, sizeof n, "ws.gen.rb%d.mean1", i); rc = dev_alloc(h, n, f, false, &m1); if (rc) return rc;
        snprintf(n, sizeof n, "ws.gen.rb%d.std1", i); rc = dev_alloc(h, n, f, false, &s1); if (rc) return rc;
        snprintf(n, sizeof n, "ws.gen.rb%d.meano", i); rc = dev_alloc(h, n, f, false, &mo); if (rc) return rc;
        snprintf(n, sizeof n, "ws.gen.rb%d.stdo", i); rc = dev_alloc(h, n, f, false, &so); if (rc) return rc;
        if (learned) { snprintf(n, sizeof n, "ws.gen.rb%d.skip", i); rc = dev_alloc(h, n, (size_t)B * r * r * f, false, &skip); if (rc) return rc; }

        // one SPADE layer + its consumer conv:  a = lrelu(SPADE(x)) ; y = conv(a)
        auto spade_then_conv = [&](int j, const float* x, int rx, int xshift, int C, const float* mean,
                                   const float* stdv, int conv_idx, float* y, int epi, const float* res, int res_r,
                                   int res_shift, bool want_stats) -> int {
            char k[160];
            Padded hb, ab;
            // MSR_FLAG_FP8: tensors that feed an fp8 conv hold one byte per channel, padded to 256 channels; the kernels
            // address them in float slots of 4 channels
            const bool gb8 = gb_uses_fp8(h, r, C), cv8 = main_uses_fp8(h, r, C, f);
            const bool gbc = gb_uses_f16c(h, r, C), cvc = main_uses_f16c(h, r, C, f), cv6 = main_uses_f16c6(h, r, C, f);
            const int hslots = gb8 ? fp8_pad(128) / 4 : 128, aslots = cv8 ? fp8_pad(C) / 4 : C;
            const bool gbr = gb_uses_gbr(h, r, C, f);
            int rc2;
            snprintf(k, sizeof k, "ws.gen.rb%d.a%d", i, j); rc2 = alloc_padded(h, k, r, aslots, &ab); if (rc2) return rc2;
            if (gbr) {
                // conv_gb_resident: the embedding never exists in HBM (no mask-embedding launch, no h buffer); one launch
                // does resize + embedding + gamma|beta conv + SPADE epilogue and writes the consumer's f16c image
                Op g; g.type = OP_GBR; g.src_is_input = true;
                GbrParams& q = g.gbr;
                snprintf(k, sizeof k, "gen.rb%d.spade_%d.conv.kernel", i, j); q.we = need(k);
                snprintf(k, sizeof k, "gen.rb%d.spade_%d.conv.kernel.e16", i, j); q.we16 = need(k);
                snprintf(k, sizeof k, "gen.rb%d.spade_%d.conv.bias", i, j); q.be = need(k);
                q.S = S; q.f = S / r; q.o = (S / r) / 2;
                snprintf(k, sizeof k, "gen.rb%d.spade_%d.gb.kernel", i, j); q.wt = need(k);
                snprintf(k, sizeof k, "gen.rb%d.spade_%d.gb.bias", i, j); q.bias = need(k);
                q.aux = x; q.aux_px = C; q.aux_py = rx * C; q.aux_pb = rx * rx * C; q.aux_shift = xshift;
                q.mean = mean; q.stdv = stdv;
                q.out = ab.base; q.out_px = ab.C; q.out_py = ab.py(); q.out_pb = ab.pb(); q.out_off = ab.interior();
                q.out_split = 4; q.slope = 0.2f;
                q.B = B; q.r = r; q.N = 2 * C;
                q.no_cross = h->f16m ? 1 : 0;
                g.flops = 2.0 * B * r * r * 128.0 * (2 * C) * 9 + 2.0 * B * r * r * 18.0 * 128;
                h->ops.push_back(g);
            } else {
            snprintf(k, sizeof k, "ws.gen.rb%d.h%d", i, j); rc2 = alloc_padded(h, k, r, hslots, &hb); if (rc2) return rc2;
            Op em; em.type = OP_SMALLCIN; em.src_is_input = true;
            SmallCinParams& p = em.sc;
            snprintf(k, sizeof k, "gen.rb%d.spade_%d.conv.kernel", i, j); p.w = need(k);
            snprintf(k, sizeof k, "gen.rb%d.spade_%d.conv.bias", i, j); p.bias = need(k);
            p.out = hb.base; p.B = B; p.S = S; p.Hout = r; p.Cout = 128;
            p.ay = 1; p.cy = -1; p.lim = r; p.f = S / r; p.o = (S / r) / 2;
            p.out_px = hslots; p.out_py = hb.py(); p.out_pb = hb.pb(); p.out_off = hb.interior();
            p.act = 1; p.slope = 0.f;
            const bool f16x2 = gb_uses_f16x2(h, r, 2 * C, 128);
            p.out_split = gb8 ? 3 : gbc ? 4 : f16x2 ? 2 : (h->prec == PREC_BF16X3 ? 1 : 0);
            em.flops = 2.0 * B * r * r * 18.0 * 128;
            em.on_aux = true;
            em.aux_group = i <= 4 ? 0 : 1;        // rb1-4 embeds are small and done early; rb5-6 carry the bytes
            if (hipEventCreateWithFlags(&em.done, hipEventDisableTiming) != hipSuccess)
                return fail(h, MSR_ERR_DEVICE, "hipEventCreate failed");
            h->ops.push_back(em);
            snprintf(k, sizeof k, "gen.rb%d.spade_%d.gb.kernel", i, j); const float* gbw = need(k);
            snprintf(k, sizeof k, "gen.rb%d.spade_%d.gb.bias", i, j); const float* gbb = need(k);
            Op gb = conv_op(hb, gbw, gbb, B, r, 2 * C, 1, EPI_SPADE, h->prec);
            if (f16x2) gb.conv.prec = PREC_F16X2;     // same tile, same layouts; fp16 encodings, 2 MFMAs per product
            if (gb8) {
                snprintf(k, sizeof k, "gen.rb%d.spade_%d.gb.kernel.wexp", i, j);
                gb.conv.prec = PREC_FP8;
                gb.conv.wexp = reinterpret_cast<const int*>(need(k));
                gb.flops = 2.0 * B * r * r * 128.0 * (2 * C) * 9;
                gb.tile = TILE_256x128_PP;            // gb_uses_fp8 checked that it fills the chip with whole tiles
                gb.conv.ksplit = 1;
                gb.conv.wt_frag = 0;
            }
            if (gbc) {
                snprintf(k, sizeof k, "gen.rb%d.spade_%d.gb.kernel.wexp", i, j);
                gb.conv.prec = PREC_F16C;
                gb.conv.wexp = reinterpret_cast<const int*>(need(k));
                gb.tile = TILE_256x128_PP;
                gb.conv.ksplit = pp_ksplit(B, r, 2 * C, 1, 128);      // > 1: K ranges (fewer tiles than CUs)
                gb.conv.wt_frag = 0;
            }
            if (cv8) gb.conv.out_split = 3;           // its epilogue writes bf8 bytes for the fp8 consumer
            if (cvc) gb.conv.out_split = cv6 ? 5 : 4; // ... the f16c (fp8 pieces) / f16c6 (fp6 pieces) chunk image for the consumer
            set_out_padded(gb.conv, ab);
            set_aux_dense(gb.conv, x, rx, C, xshift);
            gb.conv.mean = mean; gb.conv.stdv = stdv;
            gb.wait = em.done;
            gb.aux_group = em.aux_group;
            h->ops.push_back(gb);
            }
            snprintf(k, sizeof k, "gen.rb%d.conv_%d.kernel", i, conv_idx); const float* cw = need(k);
            snprintf(k, sizeof k, "gen.rb%d.conv_%d.bias", i, conv_idx); const float* cb = need(k);
            Op cv = conv_op(ab, cw, cb, B, r, f, 1, epi, h->prec);
            if (cv8) {
                snprintf(k, sizeof k, "gen.rb%d.conv_%d.kernel.wexp", i, conv_idx);
                cv.conv.prec = PREC_FP8;
                cv.conv.wexp = reinterpret_cast<const int*>(need(k));
                cv.flops = 2.0 * B * r * r * (double)C * f * 9;
                cv.tile = TILE_256x128_PP;
                cv.conv.ksplit = 1;
                cv.conv.wt_frag = 0;
            }
            if (cv6) {
                cv.conv.prec = PREC_F16C6;                    // the weight image carries its scales
                cv.conv.wexp = nullptr;
                cv.tile = TILE_256x128_PP;
                cv.conv.ksplit = 1;
                cv.conv.wt_frag = 0;
            } else if (cvc) {
                snprintf(k, sizeof k, "gen.rb%d.conv_%d.kernel.wexp", i, conv_idx);
                cv.conv.prec = PREC_F16C;
                cv.conv.wexp = reinterpret_cast<const int*>(need(k));
                cv.tile = TILE_256x128_PP;
                cv.conv.ksplit = pp_ksplit(B, r, f, 1, C);
                cv.conv.wt_frag = 0;
                cv.conv.no_cross = h->f16m ? 1 : 0;           // honoured by conv_igemm_f16c_sw (the long-K main convs)
            }
            set_out_dense(cv.conv, y, r, f);
            if (epi == EPI_RES) set_aux_dense(cv.conv, res, res_r, f, res_shift);
            // fused output moments (the tensor feeds a SPADE layer) unless the layer runs split-K
            if (want_stats && cv.conv.ksplit == 1) cv.stat_slabs = conv_stat_slabs(cv.conv, cv.tile);
            h->ops.push_back(cv);
            return MSR_OK;
        };
        // moments of a conv output: finalize the conv's own slabs if it emitted them, else read the tensor
        auto push_moments = [&](const float* x, int P, int C, float* mean, float* stdv) {
            Op& last = h->ops.back();
            if (fuse_moments_into_splitk(last, 1, 1e-5f, mean, stdv)) {
                mom_need(1, P, C);
            } else if (last.type == OP_CONV && last.stat_slabs > 0) {
                Op op; op.type = OP_MOMENTS_SLABS;
                op.mom = {nullptr, 1, last.stat_slabs, C, 1e-5f, mean, stdv};
                h->ops.push_back(op);
            } else {
                h->ops.push_back(moments_op(x, 1, P, C, 1e-5f, mean, stdv));
                mom_need(1, P, C);
            }
        };
        // x1 = conv_1(lrelu(spade_1(x)))                                   blocks.py:29-30
        rc = spade_then_conv(1, x_prev, r_prev, shift, cin, st_mean, st_std, 1, x1, EPI_BIAS, nullptr, 0, 0, true); if (rc) return rc;
        push_moments(x1, B * r * r, f, m1, s1);
        if (learned) {
            // skip = conv_3(lrelu(spade_3(x)))                             blocks.py:33-34
            rc = spade_then_conv(3, x_prev, r_prev, shift, cin, st_mean, st_std, 3, skip, EPI_BIAS, nullptr, 0, 0, false); if (rc) return rc;
            // out = skip + conv_2(lrelu(spade_2(x1)))                      blocks.py:31-32,38
            rc = spade_then_conv(2, x1, r, 0, f, m1, s1, 2, outb, EPI_RES, skip, r, 0, true); if (rc) return rc;
        } else {
            // out = x + conv_2(lrelu(spade_2(x1))), x read through the folded up-sample
            rc = spade_then_conv(2, x1, r, 0, f, m1, s1, 2, outb, EPI_RES, x_prev, r_prev, shift, true); if (rc) return rc;
        }
        // moments of the block output == moments of its nearest-2x up-sample (every value is repeated 4x)
        push_moments(outb, B * r * r, f, mo, so);
        x_prev = outb; r_prev = r; cin = f; st_mean = mo; st_std = so;
    }
    {
        Op hd; hd.type = OP_HEAD; hd.out_is_output = true;
        hd.head = {x_prev, need("gen.head.weff"), h->host_small["gen.head.bias"][0], B, r_prev, 128, 0.2f, 0, 0, 0};
        hd.flops = 2.0 * B * S * S * 16.0 * 128;
        h->ops.push_back(hd);
    }
    // A cross-stream wait stalls the main stream for ~16 us whether or not the event has fired, so the aux stream
    // signals once per group (after the group's last mask-embedding conv; the stream is in order) and only the
    // group's first consumer waits: two groups, i.e. two waits per call.
    for (int grp = 0; grp < 2; ++grp) {
        int last_aux = -1, first_wait = -1;
        for (size_t k = 0; k < h->ops.size(); ++k) {
            if (h->ops[k].aux_group != grp) continue;
            if (h->ops[k].on_aux) last_aux = (int)k;
            else if (first_wait < 0) first_wait = (int)k;
        }
        if (last_aux < 0) continue;
        for (size_t k = 0; k < h->ops.size(); ++k) {
            Op& op = h->ops[k];
            if (op.aux_group != grp) continue;
            if (op.on_aux && (int)k != last_aux) { hipEventDestroy(op.done); op.done = nullptr; }
            if (!op.on_aux) op.wait = (int)k == first_wait ? h->ops[last_aux].done : nullptr;
        }
    }
    mom_doubles = std::max<size_t>(mom_doubles, (size_t)128 * 3 * 1024);  // also the slab-group scratch
    HIPCHK(h, hipMalloc(&h->mom_partial, std::max<size_t>(mom_doubles, 16) * sizeof(double)));
    HIPCHK(h, hipMalloc(&h->dense_partial, std::max<size_t>(dense_part, 16) * sizeof(float)));
    h->total_bytes += mom_doubles * sizeof(double) + dense_part * sizeof(float);
    return MSR_OK;
}

int plan_pix2pix(msr_handle* h) {
    const int B = h->B;
    char n[128];
    int rc;
    auto fold_bn = [&](const std::string& prefix, int C, float** scale, float** shift) -> int {
        const auto& g = h->host_small[prefix + ".gamma"];
        const auto& b = h->host_small[prefix + ".beta"];
        const auto& m = h->host_small[prefix + ".moving_mean"];
        const auto& v = h->host_small[prefix + ".moving_variance"];
        std::vector<float> sc(C), sh(C);
        for (int c = 0; c < C; ++c) {
            sc[c] = g[c] / std::sqrt(v[c] + 1e-3f);   // keras BatchNormalization epsilon
            sh[c] = b[c] - m[c] * sc[c];
        }
        int r2 = upload(h, prefix + ".scale", sc.data(), C); if (r2) return r2;
        r2 = upload(h, prefix + ".shift", sh.data(), C); if (r2) return r2;
        *scale = D(h, prefix + ".scale"); *shift = D(h, prefix + ".shift");
        return MSR_OK;
    };
    // Activations live in zero-bordered concat buffers cat_i = [up_i | down_(8-i)] (pix2pix.py:99-104 concatenates
    // [x, skip]): a down block writes its half once, the next down block reads it as a channel slice
    // (in_px = total channels) and the up path reads the whole pixel.  No concat copy exists.
    Padded cat[8];           // cat[i], i = 1..7, at resolution 2^i
    Padded d8;               // the 1x1 bottleneck
    for (int i = 1; i <= 7; ++i) {
        const int cu = kP2PUp[i - 1], cd = kP2PDown[6 - (i - 1)];
        snprintf(n, sizeof n, "ws.p2p.cat%d", i);
        rc = alloc_padded(h, n, 1 << i, cu + cd, &cat[i]); if (rc) return rc;
    }
    rc = alloc_padded(h, "ws.p2p.down8", 1, 512, &d8); if (rc) return rc;
    auto igemm = [&](const float* in, int in_px, int in_py, int in_pb, int cin, const float* wt, const float* scale,
                     const float* shift, int rout, int N, int K, int stride, int act, float slope) {
        Op op; op.type = OP_CONV; op.epi = EPI_AFFINE;
        ConvParams& c = op.conv;
        c.in = in; c.wt = wt; c.bias = shift; c.scale = scale; c.act = act; c.slope = slope;
        c.B = B; c.Hout = rout; c.Wout = rout; c.Cin = cin; c.N = N; c.KH = K; c.KW = K; c.stride = stride;
        c.in_px = in_px; c.in_py = in_py; c.in_pb = in_pb;
        c.prec = PREC_F32;
        op.tile = conv_pick_tile(B * rout * rout, N, EPI_AFFINE, PREC_F32);
        c.ksplit = conv_pick_ksplit(B * rout * rout, N, K * K * (cin / 32), op.tile);
        op.flops = 2.0 * B * rout * rout * (double)cin * N * K * K;
        return op;
    };
    // ---- down1: 2 -> 64 channels, no BatchNormalization (pix2pix.py:27), on the direct kernel ----
    {
        const Padded& o = cat[7];
        Op op; op.type = OP_DIRECT; op.src_is_input = true;
        DirectConvParams& p = op.dc;
        p.in0 = nullptr; p.c0 = 2; p.in1 = nullptr; p.c1 = 0;
        p.in_px = 2; p.in_py = 256 * 2; p.in_pb = 256 * 256 * 2;
        p.w = D(h, "p2p.down1.kernel"); p.scale = p.shift = nullptr;
        p.out = o.base + o.interior() + kP2PUp[6]; p.out_px = o.C; p.out_py = o.py(); p.out_pb = o.pb();
        p.B = B; p.Hin = 256; p.Win = 256; p.Hout = 128; p.Wout = 128; p.Cout = 64;
        p.KH = 4; p.KW = 4; p.stride = 2; p.pad = 1; p.transposed = 0;
        p.act = 2; p.slope = 0.3f;   // keras LeakyReLU() default alpha (pix2pix.py:72)
        op.flops = 2.0 * B * 128 * 128 * 16.0 * 2 * 64;
        h->ops.push_back(op);
    }
    // ---- down2..8: 4x4 stride-2 implicit GEMM; the padded border is the 'same' padding (1 before, 1 after) ----
    for (int i = 2; i <= 8; ++i) {
        const int cin = kP2PDown[i - 2], c = kP2PDown[i - 1];
        const Padded& src = cat[8 - (i - 1)];
        const int src_off = kP2PUp[8 - (i - 1) - 1];          // the skip half starts after the up half
        const int rout = 256 >> i;
        snprintf(n, sizeof n, "p2p.down%d.bn", i);
        float *sc, *sh; rc = fold_bn(n, c, &sc, &sh); if (rc) return rc;
        snprintf(n, sizeof n, "p2p.down%d.kernel", i);
        Op op = igemm(src.base + src_off, src.C, src.py(), src.pb(), cin, D(h, n), sc, sh, rout, c, 4, 2, 2, 0.3f);
        if (i < 8) {
            const Padded& o = cat[8 - i];
            set_out_padded(op.conv, o);
            op.conv.out_off += kP2PUp[8 - i - 1];
        } else {
            set_out_padded(op.conv, d8);
        }
        h->ops.push_back(op);
    }
    // ---- up1..7: Conv2DTranspose + BatchNormalization (+ Dropout, identity at inference) + ReLU
    //      (pix2pix.py:76-94) as four parity sub-convolutions writing interleaved pixels of cat_i's up half ----
    for (int i = 1; i <= 7; ++i) {
        const Padded& src = i == 1 ? d8 : cat[i - 1];
        const Padded& o = cat[i];
        const int c = kP2PUp[i - 1], r = src.r;
        snprintf(n, sizeof n, "p2p.up%d.bn", i);
        float *sc, *sh; rc = fold_bn(n, c, &sc, &sh); if (rc) return rc;
        snprintf(n, sizeof n, "p2p.up%d.kernel", i);
        const float* w = D(h, n);
        for (int py = 0; py < 2; ++py)
            for (int px = 0; px < 2; ++px) {
                Op op = igemm(src.base + py * src.py() + px * src.C, src.C, src.py(), src.pb(), src.C,
                              w + (size_t)(py * 2 + px) * 4 * c * src.C, sc, sh, r, c, 2, 1, 1, 0.f);
                ConvParams& cp = op.conv;
                cp.out = o.base; cp.out_px = 2 * o.C; cp.out_py = 2 * o.py(); cp.out_pb = o.pb();
                cp.out_off = o.interior() + py * o.py() + px * o.C;
                h->ops.push_back(op);
            }
    }
    // ---- last: Conv2DTranspose(1, 4, 2, 'same', tanh) (pix2pix.py:53-57) = per-parity 2x2 taps on the head kernel ----
    {
        const Padded& src = cat[7];
        Op op; op.type = OP_HEAD;
        op.head = {src.base + src.interior(), D(h, "p2p.last.weff"), h->host_small["p2p.last.bias"][0], B, 128, src.C,
                   1.0f, 1, src.py(), src.pb()};
        op.flops = 2.0 * B * 128 * 128 * 16.0 * src.C;
        h->ops.push_back(op);
    }
    return MSR_OK;
}

void drop_graphs(msr_handle* h) {
    for (auto& g : h->graphs) {
        if (g.exec) hipGraphExecDestroy(g.exec);
        if (g.graph) hipGraphDestroy(g.graph);
    }
    h->graphs.clear();
    h->seen_once.clear();
}

int ensure_conv_partial(msr_handle* h, size_t floats) {
    if (floats <= h->conv_partial_floats) return MSR_OK;
    if (!h->graphs.empty()) {      // instantiated graphs hold the old pointer: a replay would write split-K partials into freed memory
        HIPCHK(h, hipDeviceSynchronize());
        drop_graphs(h);
    }
    if (h->conv_partial) HIPCHK(h, hipFree(h->conv_partial));
    h->conv_partial = nullptr;
    HIPCHK(h, hipMalloc(&h->conv_partial, floats * sizeof(float)));
    h->total_bytes += (floats - h->conv_partial_floats) * sizeof(float);
    h->conv_partial_floats = floats;
    return MSR_OK;
}

int ensure_plan(msr_handle* h) {
    if (h->planned) return MSR_OK;
    for (auto& s : h->specs)
        if (!s.loaded) return fail(h, MSR_ERR_STATE, "weight '%s' has not been loaded", s.name.c_str());
    for (auto& op : h->ops)
        if (op.done) hipEventDestroy(op.done);
    h->ops.clear();
    drop_graphs(h);                       // they hold the old plan's pointers
    if (!h->aux) {
        HIPCHK(h, hipStreamCreateWithFlags(&h->aux, hipStreamNonBlocking));   // (stream priority, low or high, changes nothing: measured)
        HIPCHK(h, hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        // First use now: HIP binds a stream to a hardware queue when it is first used, in order, and queues whose ids
        // are equal modulo 4 share a dispatch pipe (profiles/r02_raster_queue_pairing.txt).  Callers that pipeline two
        // handles plan them back to back (Generator.prepare) so that their four busy streams land on four pipes.
        float* touch = nullptr;
        if (dev_alloc(h, "ws.aux_touch", 4, false, &touch) == MSR_OK) HIPCHK(h, hipMemsetAsync(touch, 0, 16, h->aux));
    }
    if (h->mom_partial) { hipFree(h->mom_partial); h->mom_partial = nullptr; }
    if (h->dense_partial) { hipFree(h->dense_partial); h->dense_partial = nullptr; }
    int rc = h->variant == MSR_PIX2PIX ? plan_pix2pix(h) : plan_spade(h);
    if (rc) return rc;
    h->fwd_flops = 0;
    h->gate_op = -1;
    for (size_t k = 0; k < h->ops.size(); ++k)
        if (h->ops[k].type == OP_GBR ||
            (h->ops[k].type == OP_CONV && h->ops[k].tile == TILE_256x128_PP && h->ops[k].conv.ksplit == 1)) {
            h->gate_op = (int)k;     // first layer that fills the chip with persistent ping-pong tiles
            break;
        }
    size_t need = 0, stat_need = 0;
    for (auto& op : h->ops) {
        h->fwd_flops += op.flops;
        if (op.type == OP_CONV && op.conv.ksplit > 1)
            need = std::max(need, (size_t)op.conv.ksplit * op.conv.B * op.conv.Hout * op.conv.Wout * op.conv.N);
        if (op.type == OP_CONV && op.stat_slabs > 0)
            stat_need = std::max(stat_need, (size_t)op.stat_slabs * 3 * op.conv.N);
    }
    { int rc2 = ensure_conv_partial(h, need); if (rc2) return rc2; }
    if (stat_need > h->stat_ws_floats) {
        if (h->stat_ws) HIPCHK(h, hipFree(h->stat_ws));
        h->stat_ws = nullptr;
        HIPCHK(h, hipMalloc(&h->stat_ws, stat_need * sizeof(float)));
        h->total_bytes += (stat_need - h->stat_ws_floats) * sizeof(float);
        h->stat_ws_floats = stat_need;
    }
    HIPCHK(h, hipDeviceSynchronize());
    h->planned = true;
    return MSR_OK;
}

hipEvent_t get_event(msr_handle* h) {
    if (h->ev_used == h->ev_pool.size()) {
        hipEvent_t e;
        hipEventCreate(&e);
        h->ev_pool.push_back(e);
    }
    return h->ev_pool[h->ev_used++];
}

// The launch plan of one generator(call): every kernel of msr_forward, on `s` and the handle's auxiliary stream.
int launch_all(msr_handle* h, const float* in_dev, const float* eps_dev, float* out_dev, hipStream_t s,
               hipEvent_t gate = nullptr) {
    // Fork: ops that need only the call's input go to the auxiliary stream.  With per-kernel profiling on they are
    // simply not timed (the brackets of the main-stream kernels stay valid: waits sit before the start event).
    const bool use_aux = h->aux != nullptr;
    if (use_aux) {
        bool any = false;
        for (auto& op : h->ops) any |= op.on_aux;
        if (any) {
            // Under stream capture the fork must hang off a real node of the call's stream: with the event record as the very
            // first captured operation the auxiliary branch becomes a second ROOT of the graph, and a replay was observed to
            // start that branch before earlier work of the launch stream had finished (a torch copy into the input buffer:
            // tests/test_gpu_generator.py::test_graph_replay_equals_eager, only after other processes had used the GPU).  A
            // 16-byte memset node in front of the fork makes the graph single-rooted.
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            if (s && hipStreamIsCapturing(s, &cap) == hipSuccess && cap == hipStreamCaptureStatusActive) {
                float* touch = D(h, "ws.aux_touch");
                if (touch) HIPCHK(h, hipMemsetAsync(touch, 0, 16, s));
            }
            HIPCHK(h, hipEventRecord(h->ev_fork, s));
            HIPCHK(h, hipStreamWaitEvent(h->aux, h->ev_fork, 0));
            for (auto& op : h->ops) {
                if (!op.on_aux) continue;
                SmallCinParams p = op.sc;
                p.src = in_dev;
                hipError_t e = launch_conv_smallcin(p, h->aux);
                if (e != hipSuccess) return fail(h, MSR_ERR_DEVICE, "launch of conv_smallcin (aux) failed: %s", hipGetErrorString(e));
                if (op.done) HIPCHK(h, hipEventRecord(op.done, h->aux));
            }
        }
    }
    // prof_on == 2: only the dominant family is timed, and a run of consecutive conv launches shares one pair of
    // events (an event costs the stream 2-3 us; bracketing all ~100 launches of a call costs 7 % of the throughput)
    ProfRec run{FAM_CONV, nullptr, nullptr, 0.0, 0.0, 0};
    auto close_run = [&]() {
        if (run.launches > 0) {
            run.b = get_event(h);
            hipEventRecord(run.b, s);
            h->prof.push_back(run);
        }
        run = ProfRec{FAM_CONV, nullptr, nullptr, 0.0, 0.0, 0};
    };
    int op_index = -1;
    for (auto& op : h->ops) {
        ++op_index;
        if (use_aux && op.on_aux) continue;
        if (gate && op_index == h->gate_op) {
            // msr_forward_gated: the matrix-bound part of this call starts only after the caller's event (the end of
            // the previous call on another handle / stream); everything before it overlaps that call's tail
            if (h->prof_on == 2) close_run();
            HIPCHK(h, hipStreamWaitEvent(s, gate, 0));
        }
        if (h->prof_on == 2 && ((op.type != OP_CONV && op.type != OP_GBR) || (use_aux && op.wait))) close_run();
        if (use_aux && op.wait) HIPCHK(h, hipStreamWaitEvent(s, op.wait, 0));
        hipEvent_t ea = nullptr, eb = nullptr;
        if (h->prof_on == 1) { ea = get_event(h); eb = get_event(h); hipEventRecord(ea, s); }
        if (h->prof_on == 2 && (op.type == OP_CONV || op.type == OP_GBR)) {
            if (run.launches == 0) { run.a = get_event(h); hipEventRecord(run.a, s); }
            run.launches += 1;
            run.flops += op.flops;
            run.bytes += op.bytes;
        }
        hipError_t e = hipSuccess;
        int fam = 0;
        switch (op.type) {
            case OP_CONV: {
                fam = FAM_CONV;
                ConvParams cp = op.conv;
                cp.partial = h->conv_partial;
                cp.mom_partial = h->mom_partial;
                cp.stat_partial = op.stat_slabs > 0 ? h->stat_ws : nullptr;
                e = launch_conv_igemm(cp, op.epi, op.tile, s);
                break;
            }
            case OP_SMALLCIN: {
                fam = FAM_SMALLCIN;
                SmallCinParams p = op.sc;
                if (op.src_is_input) p.src = in_dev;
                e = launch_conv_smallcin(p, s);
                break;
            }
            case OP_MOMENTS:
                fam = FAM_MOMENTS;
                e = launch_moments(op.mom.x, op.mom.G, op.mom.P, op.mom.C, op.mom.eps, h->mom_partial, op.mom.mean,
                                   op.mom.stdv, s);
                break;
            case OP_MOMENTS_SLABS:
                fam = FAM_MOMENTS;
                e = launch_moments_from_slabs(h->stat_ws, op.mom.P, op.mom.C, op.mom.eps, h->mom_partial, op.mom.mean,
                                              op.mom.stdv, s);
                break;
            case OP_NORMACT: fam = FAM_NORMACT; e = launch_norm_act(op.na, s); break;
            case OP_DENSE:
                fam = FAM_DENSE;
                e = launch_dense(op.dense.x, op.dense.W, op.dense.bias, h->dense_partial, op.dense.y, op.dense.B,
                                 op.dense.K, op.dense.N, s);
                break;
            case OP_LATENT:
                fam = FAM_LATENT;
                e = launch_latent(op.lat.mv, eps_dev, op.lat.z, op.lat.B, op.lat.L, op.lat.sampler, s);
                break;
            case OP_HEAD:
                fam = FAM_HEAD;
                e = launch_head(op.head.x, op.head.weff, op.head.bias, out_dev, op.head.B, op.head.r, op.head.C,
                                op.head.slope, op.head.tanh_out, op.head.x_py, op.head.x_pb, s);
                break;
            case OP_GBR: {
                fam = FAM_CONV;
                GbrParams q = op.gbr;
                q.src = in_dev;
                e = launch_conv_gbr(q, conv_gbr_ranges(q.B, q.r, q.N), s);
                break;
            }
            case OP_DIRECT: {
                fam = FAM_DIRECT;
                DirectConvParams p = op.dc;
                if (op.src_is_input) p.in0 = in_dev;
                if (op.out_is_output) p.out = out_dev;
                e = launch_conv_direct(p, s);
                break;
            }
        }
        if (e != hipSuccess)
            return fail(h, MSR_ERR_DEVICE, "launch of %s failed: %s", kFamilyName[fam], hipGetErrorString(e));
        if (h->prof_on == 1) { hipEventRecord(eb, s); h->prof.push_back({fam, ea, eb, op.flops, op.bytes, 1}); }
    }
    if (h->prof_on == 2) close_run();
    return MSR_OK;
}

}  // namespace

extern "C" {

int msr_forward(msr_handle* h, const float* in_dev, const float* eps_dev, float* out_dev, int32_t batch,
                void* stream_v) {
    if (!h) return MSR_ERR_INVALID;
    if (!in_dev || !out_dev) return fail(h, MSR_ERR_INVALID, "msr_forward: null tensor pointer");
    if (batch != h->B)
        return fail(h, MSR_ERR_INVALID, "batch %d != batch_size %d the handle was created with "
                    "(the reference's sampler enforces the same, sampling.py:13-15)", batch, h->B);
    if (h->variant == MSR_GAUGAN && !eps_dev)
        return fail(h, MSR_ERR_INVALID, "variant gaugan needs the sampler noise eps [B, latent_dim]");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    int rc = ensure_plan(h);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream_v;
    if (!h->graph_on || h->prof_on) return launch_all(h, in_dev, eps_dev, out_dev, s);
    // Graph mode: the ~100 launches, the fork to the auxiliary stream and its joins are captured once per pointer
    // triple and replayed with one hipGraphLaunch (the B = 1 latency case: the early kernels of a call are shorter
    // than a launch, and every cross-stream wait costs the stream ~16 us when issued eagerly).
    for (auto& g : h->graphs)
        if (g.in == in_dev && g.eps == eps_dev && g.out == out_dev) {
            g.last_use = ++h->graph_clock;
            HIPCHK(h, hipGraphLaunch(g.exec, s));
            return MSR_OK;
        }
    if (s == nullptr) return launch_all(h, in_dev, eps_dev, out_dev, s);     // the legacy default stream cannot be captured
    // A triple is captured on its second sighting: a caller that draws a fresh noise tensor (a fresh pointer) per call
    // never pays capture + instantiate, and never fills the cache with one-shot graphs.
    {
        bool seen = false;
        for (auto& t : h->seen_once) seen |= t.in == in_dev && t.eps == eps_dev && t.out == out_dev;
        if (!seen) {
            if (h->seen_once.size() >= 16) h->seen_once.erase(h->seen_once.begin());
            h->seen_once.push_back({in_dev, eps_dev, out_dev});
            return launch_all(h, in_dev, eps_dev, out_dev, s);
        }
    }
    if (h->graphs.size() >= 8) {                      // evict the least recently used graph (nothing of it may still run)
        size_t lru = 0;
        for (size_t k = 1; k < h->graphs.size(); ++k)
            if (h->graphs[k].last_use < h->graphs[lru].last_use) lru = k;
        HIPCHK(h, hipDeviceSynchronize());
        if (h->graphs[lru].exec) hipGraphExecDestroy(h->graphs[lru].exec);
        if (h->graphs[lru].graph) hipGraphDestroy(h->graphs[lru].graph);
        h->graphs.erase(h->graphs.begin() + lru);
    }
    msr_handle::GraphEntry e{in_dev, eps_dev, out_dev, nullptr, nullptr, ++h->graph_clock};
    HIPCHK(h, hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
    rc = launch_all(h, in_dev, eps_dev, out_dev, s);
    hipError_t ce = hipStreamEndCapture(s, &e.graph);
    if (rc) { if (e.graph) hipGraphDestroy(e.graph); return rc; }
    if (ce != hipSuccess) return fail(h, MSR_ERR_DEVICE, "hipStreamEndCapture failed: %s", hipGetErrorString(ce));
    ce = hipGraphInstantiate(&e.exec, e.graph, nullptr, nullptr, 0);
    if (ce != hipSuccess) {
        hipGraphDestroy(e.graph);
        return fail(h, MSR_ERR_DEVICE, "hipGraphInstantiate failed: %s", hipGetErrorString(ce));
    }
    h->graphs.push_back(e);
    HIPCHK(h, hipGraphLaunch(e.exec, s));
    return MSR_OK;
}

int msr_forward_gated(msr_handle* h, const float* in_dev, const float* eps_dev, float* out_dev, int32_t batch,
                      void* stream_v, void* gate_event) {
    if (!h) return MSR_ERR_INVALID;
    if (!gate_event) return msr_forward(h, in_dev, eps_dev, out_dev, batch, stream_v);
    if (!in_dev || !out_dev) return fail(h, MSR_ERR_INVALID, "msr_forward_gated: null tensor pointer");
    if (batch != h->B) return fail(h, MSR_ERR_INVALID, "batch %d != batch_size %d", batch, h->B);
    if (h->variant == MSR_GAUGAN && !eps_dev)
        return fail(h, MSR_ERR_INVALID, "variant gaugan needs the sampler noise eps [B, latent_dim]");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    int rc = ensure_plan(h);
    if (rc) return rc;
    return launch_all(h, in_dev, eps_dev, out_dev, (hipStream_t)stream_v, (hipEvent_t)gate_event);
}

int msr_graph_enable(msr_handle* h, int32_t on) {
    if (!h) return MSR_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    h->graph_on = on ? 1 : 0;
    if (!on) { HIPCHK(h, hipDeviceSynchronize()); drop_graphs(h); }
    return MSR_OK;
}

int msr_last_latent(msr_handle* h, float* z_dev, void* stream) {
    if (!h || !z_dev) return MSR_ERR_INVALID;
    if (!h->z) return fail(h, MSR_ERR_STATE, "no latent: run msr_forward on a SPADE variant first");
    HIPCHK(h, hipMemcpyAsync(z_dev, h->z, (size_t)h->B * h->L * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return MSR_OK;
}

int msr_forward_flops(const msr_handle* hc, double* flops) {
    msr_handle* h = const_cast<msr_handle*>(hc);
    if (!h || !flops) return MSR_ERR_INVALID;
    int rc = ensure_plan(h);
    if (rc) return rc;
    *flops = h->fwd_flops;
    return MSR_OK;
}

static int op_conv_impl(msr_handle* h, const float* in_dev, const float* wt_dev, const float* bias_dev, float* out_dev,
                        int32_t B, int32_t rout, int32_t Cin, int32_t N, int32_t stride, int32_t epilogue,
                        const float* aux_dev, int32_t aux_shift, const float* mean_dev, const float* std_dev,
                        int32_t out_padded, int32_t tile, int prec, int out_split, void* stream) {
    if (!h) return MSR_ERR_INVALID;
    if (!in_dev || !wt_dev || !bias_dev || !out_dev || B < 1 || rout < 1 || (stride != 1 && stride != 2))
        return fail(h, MSR_ERR_INVALID, "msr_op_conv3x3: bad argument");
    if (epilogue < EPI_BIAS || epilogue > EPI_SPADE || (epilogue != EPI_BIAS && !aux_dev) ||
        (epilogue == EPI_SPADE && (!mean_dev || !std_dev)))
        return fail(h, MSR_ERR_INVALID, "msr_op_conv3x3: epilogue %d needs aux / mean / std", epilogue);
    HIPCHK(h, hipSetDevice(h->cfg.device));
    Padded in; in.base = const_cast<float*>(in_dev); in.r = rout * stride; in.C = Cin;
    Op op = conv_op(in, wt_dev, bias_dev, B, rout, N, stride, epilogue, prec);
    op.conv.out_split = (epilogue == EPI_SPADE && out_split) ? 1 : 0;
    const int Cout = epilogue == EPI_SPADE ? N / 2 : N;
    if (out_padded) { Padded o; o.base = out_dev; o.r = rout; o.C = Cout; set_out_padded(op.conv, o); }
    else set_out_dense(op.conv, out_dev, rout, Cout);
    if (epilogue != EPI_BIAS) set_aux_dense(op.conv, aux_dev, rout >> aux_shift, Cout, aux_shift);
    op.conv.mean = mean_dev; op.conv.stdv = std_dev;
    if (tile >= 0) {
        op.tile = tile & 0x3F;
        if (tile & 0x80) op.conv.prec = PREC_F16X2;            // operands are split-fp16 words (ping-pong tile only)
        op.conv.wt_frag = (tile & 0x40) ? 1 : 0;
        op.conv.ksplit = (tile >> 8) > 0 ? (tile >> 8) : 1;    // explicit tile: explicit split (default none)
    }
    if (op.conv.ksplit > 1) {
        int rc = ensure_conv_partial(h, (size_t)op.conv.ksplit * B * rout * rout * N);
        if (rc) return rc;
        op.conv.partial = h->conv_partial;
    }
    hipError_t e = launch_conv_igemm(op.conv, epilogue, op.tile, (hipStream_t)stream);
    if (e != hipSuccess) return fail(h, MSR_ERR_INVALID, "conv launch rejected (shape not tileable?): %s", hipGetErrorString(e));
    return MSR_OK;
}

int msr_op_conv3x3(msr_handle* h, const float* in_dev, const float* wt_dev, const float* bias_dev, float* out_dev,
                   int32_t B, int32_t rout, int32_t Cin, int32_t N, int32_t stride, int32_t epilogue,
                   const float* aux_dev, int32_t aux_shift, const float* mean_dev, const float* std_dev,
                   int32_t out_padded, int32_t tile, void* stream) {
    return op_conv_impl(h, in_dev, wt_dev, bias_dev, out_dev, B, rout, Cin, N, stride, epilogue, aux_dev, aux_shift,
                        mean_dev, std_dev, out_padded, tile, PREC_F32, 0, stream);
}

int msr_op_conv3x3_bf16x3(msr_handle* h, const float* in_dev, const float* wt_dev, const float* bias_dev,
                          float* out_dev, int32_t B, int32_t rout, int32_t Cin, int32_t N, int32_t stride,
                          int32_t epilogue, const float* aux_dev, int32_t aux_shift, const float* mean_dev,
                          const float* std_dev, int32_t out_padded, int32_t out_split, int32_t tile, void* stream) {
    if (tile < 0) return fail(h, MSR_ERR_INVALID, "msr_op_conv3x3_bf16x3 needs an explicit tile (the weight layout depends on it)");
    if ((tile & 0x3F) == TILE_128x128_K16) return fail(h, MSR_ERR_INVALID, "the bf16x3 path has no 16-channel K-step tile");
    return op_conv_impl(h, in_dev, wt_dev, bias_dev, out_dev, B, rout, Cin, N, stride, epilogue, aux_dev, aux_shift,
                        mean_dev, std_dev, out_padded, tile, PREC_BF16X3, out_split, stream);
}

int msr_op_conv3x3_f16c(msr_handle* h, const float* in_dev, const float* wt_dev, const int32_t* wexp_dev,
                        const float* bias_dev, float* out_dev, int32_t B, int32_t rout, int32_t Cin, int32_t N,
                        int32_t epilogue, const float* aux_dev, int32_t aux_shift, const float* mean_dev,
                        const float* std_dev, int32_t out_padded, int32_t out_mode, void* stream) {
    if (!h) return MSR_ERR_INVALID;
    if (!in_dev || !wt_dev || !bias_dev || !out_dev || B < 1 || rout < 16 || Cin % 64 || N % 128)
        return fail(h, MSR_ERR_INVALID, "msr_op_conv3x3_f16c: bad argument (Cin %% 64, N %% 128, rout >= 16)");
    // wexp_dev == nullptr: the operands are f16c6 images (fp6 pieces, scales inside; stream kernel, bias / residual epilogues)
    if (!wexp_dev && (epilogue == EPI_SPADE || Cin % 128))
        return fail(h, MSR_ERR_INVALID, "msr_op_conv3x3_f16c: the f16c6 form takes the bias / residual epilogues and Cin %% 128 == 0");
    if (epilogue < EPI_BIAS || epilogue > EPI_SPADE || (epilogue != EPI_BIAS && !aux_dev) ||
        (epilogue == EPI_SPADE && (!mean_dev || !std_dev)) || (out_mode != 0 && out_mode != 1 && out_mode != 4 && out_mode != 5) ||
        (out_mode != 0 && epilogue != EPI_SPADE))
        return fail(h, MSR_ERR_INVALID, "msr_op_conv3x3_f16c: bad epilogue / output mode");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    Padded in; in.base = const_cast<float*>(in_dev); in.r = rout; in.C = Cin;
    Op op = conv_op(in, wt_dev, bias_dev, B, rout, N, 1, epilogue, PREC_BF16X3);
    op.tile = TILE_256x128_PP;
    op.conv.ksplit = 1;
    op.conv.wt_frag = 0;
    op.conv.prec = wexp_dev ? PREC_F16C : PREC_F16C6;
    op.conv.wexp = wexp_dev;
    op.conv.out_split = epilogue == EPI_SPADE ? out_mode : 0;
    const int Cout = epilogue == EPI_SPADE ? N / 2 : N;
    if (out_padded) { Padded o; o.base = out_dev; o.r = rout; o.C = Cout; set_out_padded(op.conv, o); }
    else set_out_dense(op.conv, out_dev, rout, Cout);
    if (epilogue != EPI_BIAS) set_aux_dense(op.conv, aux_dev, rout >> aux_shift, Cout, aux_shift);
    op.conv.mean = mean_dev; op.conv.stdv = std_dev;
    hipError_t e = launch_conv_igemm(op.conv, epilogue, op.tile, (hipStream_t)stream);
    if (e != hipSuccess) return fail(h, MSR_ERR_INVALID, "f16c conv launch rejected: %s", hipGetErrorString(e));
    return MSR_OK;
}

int msr_op_spade_gbr(msr_handle* h, const float* src_dev, int32_t S, const float* we_dev, const float* be_dev,
                     const float* wt_dev, const float* bias_dev, float* out_dev, int32_t B, int32_t r, int32_t N,
                     const float* aux_dev, int32_t aux_shift, const float* mean_dev, const float* std_dev, void* stream) {
    if (!h) return MSR_ERR_INVALID;
    if (!src_dev || !we_dev || !be_dev || !wt_dev || !bias_dev || !out_dev || !aux_dev || !mean_dev || !std_dev || B < 1 ||
        r < 16 || S < r || S % r || N % 128 || aux_shift < 0 || aux_shift > 1)
        return fail(h, MSR_ERR_INVALID, "msr_op_spade_gbr: bad argument (r >= 16, S a multiple of r, N %% 128 == 0)");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int C = N / 2, rx = r >> aux_shift;
    GbrParams q{};
    q.src = src_dev; q.we = we_dev; q.be = be_dev; q.S = S; q.f = S / r; q.o = (S / r) / 2;
    q.wt = wt_dev; q.bias = bias_dev;
    q.aux = aux_dev; q.aux_px = C; q.aux_py = rx * C; q.aux_pb = rx * rx * C; q.aux_shift = aux_shift;
    q.mean = mean_dev; q.stdv = std_dev;
    Padded o; o.base = out_dev; o.r = r; o.C = C;
    q.out = o.base; q.out_px = C; q.out_py = o.py(); q.out_pb = o.pb(); q.out_off = o.interior();
    q.out_split = 4; q.slope = 0.2f; q.B = B; q.r = r; q.N = N;
    int ranges = conv_gbr_ranges(B, r, N);       // the planner's split; a layer it would not take runs one item per pixel tile
    if (ranges < 1) ranges = 1;
    // the embedding kernel as fp16 MFMA operands (msr_load_weight builds this image once per layer; this test entry per call)
    std::vector<float> we_host(9 * 2 * 128), e16(4096);
    HIPCHK(h, hipStreamSynchronize((hipStream_t)stream));
    HIPCHK(h, hipMemcpy(we_host.data(), we_dev, we_host.size() * sizeof(float), hipMemcpyDeviceToHost));
    conv_gbr_embed_image(we_host.data(), e16.data());
    float* e16_dev = nullptr;
    HIPCHK(h, hipMalloc(&e16_dev, e16.size() * sizeof(float)));
    hipError_t e = hipMemcpy(e16_dev, e16.data(), e16.size() * sizeof(float), hipMemcpyHostToDevice);
    q.we16 = e16_dev;
    if (e == hipSuccess) e = launch_conv_gbr(q, ranges, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    hipFree(e16_dev);
    if (e != hipSuccess) return fail(h, MSR_ERR_INVALID, "conv_gb_resident launch rejected: %s", hipGetErrorString(e));
    return MSR_OK;
}

int msr_op_head(msr_handle* h, const float* x_dev, const float* kernel_host, float bias, float* out_dev, int32_t B,
                int32_t r, int32_t C, float slope, int32_t variant, void* stream) {
    if (!h) return MSR_ERR_INVALID;
    if (!x_dev || !kernel_host || !out_dev || B < 1 || r < 16 || r % 16 || C < 16 || C % 16 || variant < 0 || variant > 1)
        return fail(h, MSR_ERR_INVALID, "msr_op_head: bad argument (r and C multiples of 16, variant 0 | 1)");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const std::vector<float> weff = variant ? head_weff_transpose(kernel_host, C) : head_weff_upconv(kernel_host, C);
    float* wd = nullptr;
    HIPCHK(h, hipMalloc(&wd, weff.size() * sizeof(float)));
    hipError_t e = hipMemcpy(wd, weff.data(), weff.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = launch_head(x_dev, wd, bias, out_dev, B, r, C, slope, variant, 0, 0, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    hipFree(wd);
    if (e != hipSuccess) return fail(h, MSR_ERR_DEVICE, "msr_op_head failed: %s", hipGetErrorString(e));
    return MSR_OK;
}

int64_t msr_quantize_e4m3(const float* host, int64_t n, uint8_t* out) {
    if (!host || !out || n < 0) return -1;
    for (int64_t i = 0; i < n; ++i) out[i] = msr_f32_to_e4m3(host[i]);
    return n;
}

int msr_op_conv3x3_fp8(msr_handle* h, const void* in_dev, const void* wt_dev, const int32_t* wexp_dev, const float* bias_dev,
                       float* out_dev, int32_t B, int32_t rout, int32_t Cpad, int32_t N, int32_t epilogue,
                       const float* aux_dev, int32_t aux_shift, const float* mean_dev, const float* std_dev,
                       int32_t out_padded, int32_t out_mode, void* stream) {
    if (!h) return MSR_ERR_INVALID;
    if (!in_dev || !wt_dev || !wexp_dev || !bias_dev || !out_dev || B < 1 || rout < 16 || (Cpad != 128 && Cpad % 256) ||
        N % 128)
        return fail(h, MSR_ERR_INVALID, "msr_op_conv3x3_fp8: bad argument (Cpad 128 or a multiple of 256, N %% 128, rout >= 16)");
    if (epilogue < EPI_BIAS || epilogue > EPI_SPADE || (epilogue != EPI_BIAS && !aux_dev) ||
        (epilogue == EPI_SPADE && (!mean_dev || !std_dev)) || (out_mode != 0 && out_mode != 1 && out_mode != 3) ||
        (out_mode != 0 && epilogue != EPI_SPADE))
        return fail(h, MSR_ERR_INVALID, "msr_op_conv3x3_fp8: bad epilogue / output mode");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    Padded in; in.base = const_cast<float*>(static_cast<const float*>(in_dev)); in.r = rout; in.C = Cpad / 4;
    Op op = conv_op(in, static_cast<const float*>(wt_dev), bias_dev, B, rout, N, 1, epilogue, PREC_BF16X3);
    op.tile = TILE_256x128_PP;
    op.conv.ksplit = 1;
    op.conv.wt_frag = 0;
    op.conv.prec = PREC_FP8;
    op.conv.wexp = wexp_dev;
    op.conv.out_split = epilogue == EPI_SPADE ? out_mode : 0;
    const int Cout = epilogue == EPI_SPADE ? N / 2 : N;
    const int oslots = out_mode == 3 ? fp8_pad(Cout) / 4 : Cout;
    if (out_padded) { Padded o; o.base = out_dev; o.r = rout; o.C = oslots; set_out_padded(op.conv, o); }
    else set_out_dense(op.conv, out_dev, rout, oslots);
    if (epilogue != EPI_BIAS) set_aux_dense(op.conv, aux_dev, rout >> aux_shift, Cout, aux_shift);
    op.conv.mean = mean_dev; op.conv.stdv = std_dev;
    hipError_t e = launch_conv_igemm(op.conv, epilogue, op.tile, (hipStream_t)stream);
    if (e != hipSuccess) return fail(h, MSR_ERR_INVALID, "fp8 conv launch rejected: %s", hipGetErrorString(e));
    return MSR_OK;
}

int msr_op_split_bf16(msr_handle* h, const float* in_dev, float* out_dev, int64_t count, void* stream) {
    if (!h || !in_dev || !out_dev || count < 0) return MSR_ERR_INVALID;
    if (count % 32) return fail(h, MSR_ERR_INVALID, "msr_op_split_bf16: count must be a multiple of 32 (channel chunks)");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, launch_split_bf16(in_dev, out_dev, (long)count, (hipStream_t)stream));
    return MSR_OK;
}

int msr_debug_tensor(msr_handle* h, const char* name, float* host_out, int64_t count) {
    if (!h || !name || !host_out) return MSR_ERR_INVALID;
    auto it = h->dev.find(name);
    if (it == h->dev.end()) return fail(h, MSR_ERR_INVALID, "no tensor named '%s'", name);
    if (count < 0 || (size_t)count * sizeof(float) > h->dev_bytes[name])
        return fail(h, MSR_ERR_INVALID, "%s holds %zu floats, %lld requested", name, h->dev_bytes[name] / sizeof(float),
                    (long long)count);
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipDeviceSynchronize());
    HIPCHK(h, hipMemcpy(host_out, it->second, (size_t)count * sizeof(float), hipMemcpyDeviceToHost));
    return MSR_OK;
}

int msr_device_bytes(const msr_handle* h, int64_t* bytes) {
    if (!h || !bytes) return MSR_ERR_INVALID;
    *bytes = (int64_t)h->total_bytes;
    return MSR_OK;
}

int msr_profile_enable(msr_handle* h, int32_t on) {
    if (!h) return MSR_ERR_INVALID;
    if (on < 0 || on > 2) return fail(h, MSR_ERR_INVALID, "msr_profile_enable: mode must be 0, 1 or 2");
    h->prof_on = on;
    return MSR_OK;
}

int msr_profile_reset(msr_handle* h) {
    if (!h) return MSR_ERR_INVALID;
    hipDeviceSynchronize();
    h->prof.clear();
    h->ev_used = 0;
    return MSR_OK;
}

int msr_profile_read(msr_handle* h, msr_kernel_stat* out, int32_t cap, int32_t* n) {
    if (!h || !out || !n) return MSR_ERR_INVALID;
    HIPCHK(h, hipDeviceSynchronize());
    msr_kernel_stat st[FAM_COUNT];
    std::memset(st, 0, sizeof st);
    for (int f = 0; f < FAM_COUNT; ++f)
        std::snprintf(st[f].name, sizeof st[f].name, "%s%s", kFamilyName[f],
                      f == FAM_CONV ? (h->prec == PREC_BF16X3 ? "_bf16x3" : "_f32") : "");
    for (auto& r : h->prof) {
        float ms = 0.f;
        HIPCHK(h, hipEventElapsedTime(&ms, r.a, r.b));
        st[r.fam].launches += r.launches;
        st[r.fam].device_ms += ms;
        st[r.fam].flops += r.flops;
        st[r.fam].bytes += r.bytes;
    }
    int k = 0;
    for (int f = 0; f < FAM_COUNT && k < cap; ++f)
        if (st[f].launches) out[k++] = st[f];
    *n = k;
    return MSR_OK;
}

int msr_profile_runs(msr_handle* h, void* ref_event, int32_t family, double* start_ms, double* end_ms, double* flops,
                     int64_t* launches, int32_t cap, int32_t* n) {
    if (!h || !ref_event || !start_ms || !end_ms || !flops || !launches || !n || family < 0 || family >= FAM_COUNT)
        return MSR_ERR_INVALID;
    HIPCHK(h, hipDeviceSynchronize());
    int k = 0;
    for (auto& r : h->prof) {
        if (r.fam != family) continue;
        if (k >= cap) return fail(h, MSR_ERR_INVALID, "msr_profile_runs: %d records do not fit", (int)h->prof.size());
        float a = 0.f, b = 0.f;
        HIPCHK(h, hipEventElapsedTime(&a, (hipEvent_t)ref_event, r.a));
        HIPCHK(h, hipEventElapsedTime(&b, (hipEvent_t)ref_event, r.b));
        start_ms[k] = a; end_ms[k] = b; flops[k] = r.flops; launches[k] = r.launches;
        ++k;
    }
    *n = k;
    return MSR_OK;
}

// ------------------------------------------------------------------------------------------------
// tiler / stitcher
// ------------------------------------------------------------------------------------------------
int msr_patch_stats(msr_handle* h, const float* img, const float* dem, int32_t rows, int32_t cols, const int32_t* ox,
                    const int32_t* oy, int32_t n, float no_value, uint8_t* valid, float* minmax, void* stream) {
    if (!h) return MSR_ERR_INVALID;
    if (!img || !dem || !ox || !oy || !valid || !minmax || n < 0 || rows <= 0 || cols <= 0)
        return fail(h, MSR_ERR_INVALID, "msr_patch_stats: bad argument");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, launch_patch_stats(img, dem, rows, cols, ox, oy, n, h->S, no_value, valid, minmax, (hipStream_t)stream));
    return MSR_OK;
}

int msr_extract_patches(msr_handle* h, const float* img, const float* dem, int32_t rows, int32_t cols,
                        const int32_t* ox, const int32_t* oy, const float* minmax, int32_t n, float* out,
                        void* stream) {
    if (!h) return MSR_ERR_INVALID;
    if (!img || !dem || !ox || !oy || !minmax || !out || n < 0)
        return fail(h, MSR_ERR_INVALID, "msr_extract_patches: bad argument");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, launch_extract_patches(img, dem, rows, cols, ox, oy, minmax, n, h->S, out, (hipStream_t)stream));
    return MSR_OK;
}

int msr_compact_patches(msr_handle* h, const uint8_t* valid, const int32_t* ox, const int32_t* oy, const float* minmax,
                        int32_t n, int32_t tile_x, int32_t tile_y, int32_t batch, int32_t cap, int32_t* sel_x,
                        int32_t* sel_y, float* sel_mm, int32_t* key, float* dmm, int32_t* meta, void* stream) {
    if (!h) return MSR_ERR_INVALID;
    if (!valid || !ox || !oy || !minmax || !sel_x || !sel_y || !sel_mm || !key || !dmm || !meta || n < 0 || batch < 1)
        return fail(h, MSR_ERR_INVALID, "msr_compact_patches: bad argument");
    if (cap < (n + batch - 1) / batch * batch)
        return fail(h, MSR_ERR_INVALID, "msr_compact_patches: cap %d < ceil(%d / %d) * %d", cap, n, batch, batch);
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, launch_compact_patches(valid, ox, oy, minmax, n, tile_x, tile_y, batch, cap, sel_x, sel_y, sel_mm, key, dmm,
                                     meta, (hipStream_t)stream));
    return MSR_OK;
}

int msr_resize_area(msr_handle* h, const float* src, int32_t rows, int32_t cols, int32_t factor, float* dst,
                    int32_t dst_rows, int32_t dst_cols, void* stream) {
    if (!h) return MSR_ERR_INVALID;
    if (!src || !dst || rows <= 0 || cols <= 0 || factor < 1 || dst_rows <= 0 || dst_cols <= 0)
        return fail(h, MSR_ERR_INVALID, "msr_resize_area: bad argument");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, launch_resize_area(src, rows, cols, dst, dst_rows, dst_cols, factor, (hipStream_t)stream));
    return MSR_OK;
}

int msr_resize_cubic(msr_handle* h, const float* src, int32_t rows, int32_t cols, float* dst, int32_t dst_rows,
                     int32_t dst_cols, void* stream) {
    if (!h) return MSR_ERR_INVALID;
    if (!src || !dst || rows <= 0 || cols <= 0 || dst_rows <= 0 || dst_cols <= 0)
        return fail(h, MSR_ERR_INVALID, "msr_resize_cubic: bad argument");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, launch_resize_cubic(src, rows, cols, dst, dst_rows, dst_cols, (hipStream_t)stream));
    return MSR_OK;
}

int msr_set_blend_window(msr_handle* h, const double* host_window, int32_t side) {
    if (!h) return MSR_ERR_INVALID;
    const int ws = h->S - 2 * (h->S / 16);
    if (!host_window || side != ws) return fail(h, MSR_ERR_INVALID, "blend window must be [%d,%d] float64", ws, ws);
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (!h->window) HIPCHK(h, hipMalloc(&h->window, (size_t)ws * ws * sizeof(double)));
    HIPCHK(h, hipMemcpy(h->window, host_window, (size_t)ws * ws * sizeof(double), hipMemcpyHostToDevice));
    return MSR_OK;
}

static int default_window(msr_handle* h) {
    // makeGaussianKernel (process_full_tiles.py:347-361) + 1e-7, purged (:391-393), in float64 like NumPy.
    const int S = h->S, p = S / 16, ws = S - 2 * p;
    std::vector<double> ax(S), k((size_t)S * S);
    const double start = -S / 2.0, stop = S / 2.0, step = (stop - start) / (S - 1);
    for (int i = 0; i < S; ++i) ax[i] = i * step + start;
    ax[S - 1] = stop;
    const double sx = S / 5.0;
    double mn = INFINITY, mx = -INFINITY;
    for (int y = 0; y < S; ++y)
        for (int x = 0; x < S; ++x) {
            const double v = 1.0 / (2.0 * M_PI * sx * sx) *
                             std::exp(-(std::pow(ax[x] - 0, 2.0) / (2.0 * std::pow(sx, 2.0)) +
                                        std::pow(ax[y] - 0, 2.0) / (2.0 * std::pow(sx, 2.0))));
            k[(size_t)y * S + x] = v;
            mn = std::min(mn, v); mx = std::max(mx, v);
        }
    std::vector<double> w((size_t)ws * ws);
    for (int y = 0; y < ws; ++y)
        for (int x = 0; x < ws; ++x) w[(size_t)y * ws + x] = (k[(size_t)(y + p) * S + x + p] - mn) / (mx - mn) + 1e-7;
    return msr_set_blend_window(h, w.data(), ws);
}

static int stitch_impl(msr_handle* h, const float* pred, const int32_t* key, const float* dmm, int32_t n,
                       int32_t tile_size, int32_t stride, float no_value, int32_t as_implemented, float* mean,
                       float* stdv, uint8_t* good, float* wsum_partial, void* stream, int pitch = 0, int resume = 0) {
    if (tile_size <= 0 || stride <= 0 || stride > h->S)
        return fail(h, MSR_ERR_INVALID, "msr_stitch_tile: tile_size %d / stride %d invalid for image_size %d", tile_size,
                    stride, h->S);
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (!h->window) { int rc = default_window(h); if (rc) return rc; }
    const int NG = (tile_size + h->S - 1) / stride;
    if (NG * NG > h->stitch_grid_cap) {
        if (h->stitch_grid) HIPCHK(h, hipFree(h->stitch_grid));
        HIPCHK(h, hipMalloc(&h->stitch_grid, (size_t)NG * NG * sizeof(int)));
        h->stitch_grid_cap = NG * NG;
    }
    HIPCHK(h, launch_stitch_tile(pred, key, dmm, n, h->S, tile_size, stride, no_value, as_implemented, h->window,
                                 h->stitch_grid, mean, stdv, good, (hipStream_t)stream, wsum_partial, pitch, resume));
    return MSR_OK;
}

int msr_stitch_tile(msr_handle* h, const float* pred, const int32_t* key, const float* dmm, int32_t n,
                    int32_t tile_size, int32_t stride, float no_value, int32_t as_implemented, float* mean,
                    float* stdv, uint8_t* good, void* stream) {
    if (!h) return MSR_ERR_INVALID;
    if (!mean || !stdv || !good || n < 0 || (n > 0 && (!pred || !key || !dmm)))
        return fail(h, MSR_ERR_INVALID, "msr_stitch_tile: null pointer");
    return stitch_impl(h, pred, key, dmm, n, tile_size, stride, no_value, as_implemented, mean, stdv, good, nullptr, stream);
}

int msr_stitch_partial(msr_handle* h, const float* pred, const int32_t* key, const float* dmm, int32_t n,
                       int32_t tile_size, int32_t stride, float* wsum, float* mean, float* s_acc, void* stream) {
    if (!h) return MSR_ERR_INVALID;
    if (!wsum || !mean || !s_acc || n < 0 || (n > 0 && (!pred || !key || !dmm)))
        return fail(h, MSR_ERR_INVALID, "msr_stitch_partial: null pointer");
    return stitch_impl(h, pred, key, dmm, n, tile_size, stride, 0.f, /*as_implemented=*/0, mean, s_acc, nullptr, wsum, stream);
}

int msr_stitch_accumulate(msr_handle* h, const float* pred, const int32_t* key, const float* dmm, int32_t n,
                          int32_t tile_size, int32_t stride, float* wsum, float* mean, float* s_acc, int32_t pitch,
                          int32_t resume, void* stream) {
    if (!h) return MSR_ERR_INVALID;
    if (!wsum || !mean || !s_acc || n < 0 || pitch < tile_size || (n > 0 && (!pred || !key || !dmm)))
        return fail(h, MSR_ERR_INVALID, "msr_stitch_accumulate: bad argument (pitch >= tile_size)");
    return stitch_impl(h, pred, key, dmm, n, tile_size, stride, 0.f, /*as_implemented=*/0, mean, s_acc, nullptr, wsum, stream,
                       pitch, resume ? 1 : 0);
}

int msr_halo_merge(msr_handle* h, const float* wa, const float* ma, const float* sa, const float* wb, const float* mb,
                   const float* sb, int64_t count, float no_value, float* mean, float* stdv, uint8_t* good,
                   void* stream) {
    if (!h) return MSR_ERR_INVALID;
    if (!wa || !ma || !sa || !mean || !stdv || !good || count < 0 || (wb && (!mb || !sb)))
        return fail(h, MSR_ERR_INVALID, "msr_halo_merge: bad argument");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, launch_halo_merge(wa, ma, sa, wb, mb, sb, (long)count, no_value, mean, stdv, good, (hipStream_t)stream));
    return MSR_OK;
}

}  // extern "C"
