// conv_igemm_f32 — NHWC implicit-GEMM convolution for gfx950 on v_mfma_f32_32x32x2_f32.
//
// Replaces the cuDNN/Eigen Conv2D calls behind spade.py:10-11,19-20 (gamma/beta convs, 49.9 % of the
// generator's FLOPs) and blocks.py:19-20,26,30-34 (ResidualBlock convs, 48.5 %), plus the strided
// encoder convs of blocks.py:52-61.  fp32-in / fp32-accumulate MFMA is bit-for-bit an fmaf chain
// (MI355X_MICROARCH.md "Matrix cores"), so parity with the fp32 reference holds at fp32 rounding.
//
// Structure (per workgroup of WM x WN waves):
//   tile  BM = WM*MT*32 output pixels  x  BN = WN*NT*32 output channels, K-step = 32 channels of one tap
//   A (pixels x k) and B (channels x k) tiles are staged global -> VGPR -> LDS with 16-byte accesses,
//   double-buffered, one barrier per K-step, loads for step t+1 issued before the MFMAs of step t.
//   LDS rows are [row][32 k + 4 pad] floats: the 144-byte pitch makes the ds_read_b128 fragment reads
//   bank-conflict free (16 lanes of a read group hit 16 distinct 16-byte slots).
//   Fragment trick: lane (i, h) reads k = 8*kk + 4*h + {0..3} as ONE ds_read_b128 and feeds element s
//   to MFMA s; the B lane reads the same k, so the four MFMAs cover the 8 k's exactly once.
//   Each wave owns MT x NT accumulator tiles of 32x32 (64 VGPRs for 2x2).
//   The input tensor carries a physical zero border, so no bounds checks exist in the K loop.
#include "kernels.h"

namespace msr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

static constexpr int BKC = 32;  // channels per K-step
static constexpr int BKP = 36;  // LDS row pitch in floats (32 + 4 pad)

__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
    // Blocks are dealt round-robin over the 8 XCDs; give each XCD a contiguous range of logical tiles so
    // that neighbouring tiles (same pixels, next channel block) share that XCD's L2.  Bijective for any nwg.
    const int q = nwg >> 3, r = nwg & 7, x = orig & 7;
    const int base = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (orig >> 3);
}

struct TileGeom {
    int th_l, tw_l, tb;            // log2 tile height/width, samples per tile
    int tiles_x, tiles_y, tiles_b, tiles_n;
};

template <int WM, int WN, int MT, int NT, int EPI>
__global__ void __launch_bounds__(WM * WN * 64) __attribute__((amdgpu_waves_per_eu(2, 2)))
conv_igemm_f32(const ConvParams p, const TileGeom g) {
    constexpr int NTHR = WM * WN * 64;
    constexpr int BM = WM * MT * 32;
    constexpr int BN = WN * NT * 32;
    constexpr int A_ITEMS = BM * 8 / NTHR;
    constexpr int B_ITEMS = BN * 8 / NTHR;
    static_assert(BM * 8 % NTHR == 0 && BN * 8 % NTHR == 0, "staging split");
    static_assert(EPI != EPI_SPADE || NT % 2 == 0, "SPADE epilogue pairs gamma/beta sub-tiles");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const As = smem;                      // [2][BM][BKP]
    float* const Bs = smem + 2 * BM * BKP;       // [2][BN][BKP]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int half = lane >> 5, l31 = lane & 31;

    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = bid % g.tiles_n;
    int tmi = bid / g.tiles_n;
    const int tx0 = (tmi % g.tiles_x) << g.tw_l;
    tmi /= g.tiles_x;
    const int ty0 = (tmi % g.tiles_y) << g.th_l;
    const int b0 = (tmi / g.tiles_y) * g.tb;
    const int n0 = tn * BN;
    const int twm = (1 << g.tw_l) - 1, thm = (1 << g.th_l) - 1;

    // ---- staging assignments ------------------------------------------------------------------
    int a_goff[A_ITEMS];   // global float offset of this thread's A rows (tap (0,0), channel chunk 0)
    int a_loff[A_ITEMS];
#pragma unroll
    for (int q = 0; q < A_ITEMS; ++q) {
        const int idx = tid + q * NTHR;
        const int row = idx >> 3, seg = idx & 7;
        const int tx = row & twm, ty = (row >> g.tw_l) & thm, tbi = row >> (g.tw_l + g.th_l);
        int b = b0 + tbi;
        b = b < p.B ? b : p.B - 1;   // rows past the batch read valid memory and are dropped in the epilogue
        a_goff[q] = b * p.in_pb + (ty0 + ty) * p.stride * p.in_py + (tx0 + tx) * p.stride * p.Cin + seg * 4;
        a_loff[q] = row * BKP + seg * 4;
    }
    int b_goff[B_ITEMS];
    int b_loff[B_ITEMS];
#pragma unroll
    for (int q = 0; q < B_ITEMS; ++q) {
        const int idx = tid + q * NTHR;
        const int row = idx >> 3, seg = idx & 7;
        b_goff[q] = (n0 + row) * p.Cin + seg * 4;
        b_loff[q] = row * BKP + seg * 4;
    }

    // ---- fragment read offsets ------------------------------------------------------------------
    int a_frag[MT], b_frag[NT];
#pragma unroll
    for (int m = 0; m < MT; ++m) a_frag[m] = ((wm * MT + m) * 32 + l31) * BKP + 4 * half;
#pragma unroll
    for (int n = 0; n < NT; ++n) b_frag[n] = ((wn * NT + n) * 32 + l31) * BKP + 4 * half;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    const int taps = p.KH * p.KW;
    const int chunks = p.Cin / BKC;
    const int steps = taps * chunks;
    const size_t w_tap_stride = (size_t)p.N * p.Cin;

    // K-step iterator: channel chunk outer, tap inner (kh, kw), kept as running scalars (no divisions).
    int it_kh = 0, it_kw = 0;
    const float* a_src = p.in;               // + kh*in_py + kw*Cin + cc*BKC
    const float* b_src = p.wt;               // + tap*N*Cin + cc*BKC

    // Staging registers.  Everything below is written so that ra/rb stay in VGPRs: fixed trip counts,
    // compile-time indices, and no conditional around the load/write pair (the last K-step is peeled).
    // Named scalars, not arrays: hipcc leaves a float4 array that crosses a sched_barrier in scratch memory.
    static_assert(A_ITEMS == 4 && B_ITEMS == 4, "staging is written for 4 + 4 16-byte items per thread");
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
#define MSR_ISSUE_LOADS()                                                                        \
    {                                                                                            \
        ra0 = *reinterpret_cast<const float4*>(a_src + a_goff[0]);                               \
        ra1 = *reinterpret_cast<const float4*>(a_src + a_goff[1]);                               \
        ra2 = *reinterpret_cast<const float4*>(a_src + a_goff[2]);                               \
        ra3 = *reinterpret_cast<const float4*>(a_src + a_goff[3]);                               \
        rb0 = *reinterpret_cast<const float4*>(b_src + b_goff[0]);                               \
        rb1 = *reinterpret_cast<const float4*>(b_src + b_goff[1]);                               \
        rb2 = *reinterpret_cast<const float4*>(b_src + b_goff[2]);                               \
        rb3 = *reinterpret_cast<const float4*>(b_src + b_goff[3]);                               \
    }
#define MSR_ADVANCE()                                                                            \
    {                                                                                            \
        ++it_kw;                                                                                 \
        a_src += p.Cin;                                                                          \
        b_src += w_tap_stride;                                                                   \
        if (it_kw == p.KW) {                                                                     \
            it_kw = 0;                                                                           \
            ++it_kh;                                                                             \
            a_src += p.in_py - p.KW * p.Cin;                                                     \
            if (it_kh == p.KH) {                                                                 \
                it_kh = 0;                                                                       \
                a_src += BKC - p.KH * p.in_py;                                                   \
                b_src += BKC - (size_t)taps * w_tap_stride;                                      \
            }                                                                                    \
        }                                                                                        \
    }
#define MSR_WRITE_LDS(buf)                                                                       \
    {                                                                                            \
        float* a_ = As + (buf) * BM * BKP;                                                       \
        float* b_ = Bs + (buf) * BN * BKP;                                                       \
        *reinterpret_cast<float4*>(a_ + a_loff[0]) = ra0;                                        \
        *reinterpret_cast<float4*>(a_ + a_loff[1]) = ra1;                                        \
        *reinterpret_cast<float4*>(a_ + a_loff[2]) = ra2;                                        \
        *reinterpret_cast<float4*>(a_ + a_loff[3]) = ra3;                                        \
        *reinterpret_cast<float4*>(b_ + b_loff[0]) = rb0;                                        \
        *reinterpret_cast<float4*>(b_ + b_loff[1]) = rb1;                                        \
        *reinterpret_cast<float4*>(b_ + b_loff[2]) = rb2;                                        \
        *reinterpret_cast<float4*>(b_ + b_loff[3]) = rb3;                                        \
    }
#define MSR_COMPUTE(buf)                                                                         \
    {                                                                                            \
        const float* a_ = As + (buf) * BM * BKP;                                                 \
        const float* b_ = Bs + (buf) * BN * BKP;                                                 \
        _Pragma("unroll") for (int kk = 0; kk < BKC / 8; ++kk) {                                 \
            float4 fa[MT], fb[NT];                                                               \
            _Pragma("unroll") for (int m = 0; m < MT; ++m)                                       \
                fa[m] = *reinterpret_cast<const float4*>(a_ + a_frag[m] + kk * 8);               \
            _Pragma("unroll") for (int n = 0; n < NT; ++n)                                       \
                fb[n] = *reinterpret_cast<const float4*>(b_ + b_frag[n] + kk * 8);               \
            _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                      \
                _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                 \
                    const float av = s == 0 ? fa[m].x : s == 1 ? fa[m].y : s == 2 ? fa[m].z : fa[m].w; \
                    _Pragma("unroll") for (int n = 0; n < NT; ++n) {                             \
                        const float bv = s == 0 ? fb[n].x : s == 1 ? fb[n].y : s == 2 ? fb[n].z : fb[n].w; \
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[m][n], 0, 0, 0); \
                    }                                                                            \
                }                                                                                \
            }                                                                                    \
        }                                                                                        \
    }

    MSR_ISSUE_LOADS();
    MSR_WRITE_LDS(0);
    __syncthreads();
    int cur = 0;
    for (int t = 0; t < steps - 1; ++t) {
        MSR_ADVANCE();
        MSR_ISSUE_LOADS();       // global loads of step t+1 fly while the MFMAs of step t run
        __builtin_amdgcn_sched_barrier(0);   // keep hipcc from sinking the loads below the MFMAs
        MSR_COMPUTE(cur);
        __builtin_amdgcn_sched_barrier(0);
        MSR_WRITE_LDS(cur ^ 1);  // the other buffer was last read before the previous barrier
        __syncthreads();
        cur ^= 1;
    }
    MSR_COMPUTE(cur);
#undef MSR_ISSUE_LOADS
#undef MSR_ADVANCE
#undef MSR_WRITE_LDS
#undef MSR_COMPUTE

    // ---- epilogue -----------------------------------------------------------------------------------
    // C/D map of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5).
    // Per-column constants (bias, SPADE mean / std) are loaded once, before the row loops.
    constexpr int NCH = EPI == EPI_SPADE ? NT / 2 : NT;
    float cb0[NCH], cb1[NCH], cmean[NCH], cstd[NCH];
    int ccol[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        if constexpr (EPI == EPI_SPADE) {
            const int colg = n0 + (wn * NT + 2 * j) * 32 + l31;   // gamma column; its beta twin is +32
            ccol[j] = (n0 + wn * NT * 32) / 2 + j * 32 + l31;    // channel
            cb0[j] = p.bias[colg];
            cb1[j] = p.bias[colg + 32];
            cmean[j] = p.mean[ccol[j]];
            cstd[j] = p.stdv[ccol[j]];
        } else {
            ccol[j] = n0 + (wn * NT + j) * 32 + l31;
            cb0[j] = p.bias[ccol[j]];
            cb1[j] = cmean[j] = cstd[j] = 0.f;
        }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (wm * MT + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const int tx = row & twm, ty = (row >> g.tw_l) & thm, tbi = row >> (g.tw_l + g.th_l);
            const int bb = b0 + tbi;
            if (tbi >= g.tb || bb >= p.B) continue;
            const int y = ty0 + ty, x = tx0 + tx;
            float* orow = p.out + (size_t)p.out_off + (size_t)bb * p.out_pb + y * p.out_py + x * p.out_px;
            if constexpr (EPI == EPI_SPADE) {
                const float* xrow = p.aux + (size_t)bb * p.aux_pb + (y >> p.aux_shift) * p.aux_py +
                                    (x >> p.aux_shift) * p.aux_px;
#pragma unroll
                for (int j = 0; j < NCH; ++j) {
                    const float gam = acc[m][2 * j][r] + cb0[j];
                    const float bet = acc[m][2 * j + 1][r] + cb1[j];
                    const float normalized = (xrow[ccol[j]] - cmean[j]) / cstd[j];
                    float v = gam * normalized + bet;
                    v = v >= 0.f ? v : v * p.slope;
                    orow[ccol[j]] = v;
                }
            } else {
                const float* rrow = nullptr;
                if constexpr (EPI == EPI_RES)
                    rrow = p.aux + (size_t)bb * p.aux_pb + (y >> p.aux_shift) * p.aux_py +
                           (x >> p.aux_shift) * p.aux_px;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    float v = acc[m][n][r] + cb0[n];
                    if constexpr (EPI == EPI_RES) v += rrow[ccol[n]];
                    orow[ccol[n]] = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
template <int WM, int WN, int MT, int NT>
struct TileCfg {
    static constexpr int BM = WM * MT * 32, BN = WN * NT * 32, NTHR = WM * WN * 64;
    static constexpr size_t LDS = (size_t)(2 * BM + 2 * BN) * BKP * sizeof(float);
};
using CfgBig = TileCfg<2, 2, 2, 2>;     // 128 x 128, 4 waves, 72 KiB LDS -> 2 workgroups per CU
using CfgSmall = TileCfg<2, 1, 1, 2>;   //  64 x  64, 2 waves, 36 KiB LDS -> 4 workgroups per CU

template <int WM, int WN, int MT, int NT, int EPI>
static hipError_t set_attr() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_f32<WM, WN, MT, NT, EPI>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)TileCfg<WM, WN, MT, NT>::LDS);
}

hipError_t conv_igemm_init() {
    hipError_t e;
    if ((e = set_attr<2, 2, 2, 2, EPI_BIAS>()) != hipSuccess) return e;
    if ((e = set_attr<2, 2, 2, 2, EPI_RES>()) != hipSuccess) return e;
    if ((e = set_attr<2, 2, 2, 2, EPI_SPADE>()) != hipSuccess) return e;
    if ((e = set_attr<2, 1, 1, 2, EPI_BIAS>()) != hipSuccess) return e;
    if ((e = set_attr<2, 1, 1, 2, EPI_RES>()) != hipSuccess) return e;
    if ((e = set_attr<2, 1, 1, 2, EPI_SPADE>()) != hipSuccess) return e;
    return hipSuccess;
}

static int ilog2_floor(int v) {
    int l = 0;
    while ((2 << l) <= v) ++l;
    return l;
}

static bool make_geom(const ConvParams& p, int BM, int BN, TileGeom& g) {
    auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
    if (!pow2(p.Hout) || !pow2(p.Wout)) return false;
    if (p.N % BN || p.Cin % BKC) return false;
    int tw = p.Wout < 16 ? p.Wout : 16;
    if (tw > BM) tw = BM;
    int th = BM / tw;
    if (th > p.Hout) th = p.Hout;
    int tb = BM / (tw * th);
    g.tw_l = ilog2_floor(tw);
    g.th_l = ilog2_floor(th);
    g.tb = tb;
    g.tiles_x = p.Wout / tw;
    g.tiles_y = p.Hout / th;
    g.tiles_b = (p.B + tb - 1) / tb;
    g.tiles_n = p.N / BN;
    return true;
}

int conv_pick_tile(int M, int N) {
    // The big tile needs >= ~2 waves of workgroups per CU to hide its barrier; otherwise take the small one.
    const long big_blocks = (long)((M + 127) / 128) * (N / 128);
    return (N % 128 == 0 && big_blocks >= 512) ? TILE_128x128 : TILE_64x64;
}

template <int WM, int WN, int MT, int NT>
static hipError_t launch_cfg(const ConvParams& p, int epi, hipStream_t s) {
    using C = TileCfg<WM, WN, MT, NT>;
    TileGeom g;
    if (!make_geom(p, C::BM, C::BN, g)) return hipErrorInvalidValue;
    const int grid = g.tiles_x * g.tiles_y * g.tiles_b * g.tiles_n;
    switch (epi) {
        case EPI_BIAS:
            conv_igemm_f32<WM, WN, MT, NT, EPI_BIAS><<<grid, C::NTHR, C::LDS, s>>>(p, g);
            break;
        case EPI_RES:
            conv_igemm_f32<WM, WN, MT, NT, EPI_RES><<<grid, C::NTHR, C::LDS, s>>>(p, g);
            break;
        case EPI_SPADE:
            conv_igemm_f32<WM, WN, MT, NT, EPI_SPADE><<<grid, C::NTHR, C::LDS, s>>>(p, g);
            break;
        default:
            return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_conv_igemm(const ConvParams& p, int epilogue, int tile, hipStream_t s) {
    if (tile == TILE_128x128) return launch_cfg<2, 2, 2, 2>(p, epilogue, s);
    return launch_cfg<2, 1, 1, 2>(p, epilogue, s);
}

}  // namespace msr
