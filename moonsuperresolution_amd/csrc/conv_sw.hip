// conv_igemm_f16c_sw — the f16c convolution (kernels.h PREC_F16C) as ONE software-pipelined stream per SIMD.
//
// Replaces the same Conv2D calls as conv_igemm.hip (ResidualBlock convs blocks.py:19-20,26,30-34; also able to run the SPADE
// gamma|beta convs spade.py:10-11,19-20) for the launches that fill the chip with whole 16 x 16 pixel x 128 channel tiles.
//
// Why another kernel: the ping-pong kernel (conv_igemm_bf16x3_pp) holds two waves per SIMD half a K-step apart, so a
// phase lasts max(memory segment ~560 cycles, matrix segment) + a barrier turn-around, and the f16c arithmetic (256 /
// 768 MFMA cycles on even / odd steps) leaves the matrix pipe ~64 % busy.  Here a workgroup is FOUR waves, one per SIMD,
// each with the whole 512-entry register file, and every wave runs MFMAs back to back with its own loads issued in the
// gaps (measured: 78 % busy, tools/gpu_sw_stamps.py — at a clock the chip lowers to ~1.75 GHz from the ping-pong kernel's
// ~1.94, which is why the layer-level gain is 3-5 % and not 20 %):
//   * wave q owns ALL 256 pixels of the tile and 32 of its 128 output columns (two 16-column blocks): 16 x 2
//     accumulator tiles of 16 x 16 = 128 registers.  Its weights are nobody else's, so they go global -> registers
//     directly (two 16-row fragments, f16 part and cross part, requested one tap pair = ~2500 cycles ahead) and never
//     touch LDS: no weight staging, no per-K-step barrier.
//   * the input halo (18 x 18 pixels x one 32-channel chunk, 160-byte rows, conflict-free ds_read_b128) is shared by the
//     four waves through a ring of THREE LDS buffers (155,520 B): chunk n+1 is staged while chunk n is multiplied, and
//     the cross-term MFMA that pairs tap 8 of chunk n with tap 0 of chunk n+1 may still read chunk n's buffer — the
//     third buffer is what allows exactly ONE workgroup barrier per chunk (9 K-steps) instead of two per K-step.
//   * pixel fragments stream through a short register ring: the ds_read_b128 of fragment i + PD is issued between the
//     MFMAs of fragment i.
// A tap pair (U) runs three phases: E (tap 2U: 32 f16 MFMAs), O (tap 2U+1: 32 f16 MFMAs), C (32 block-scaled K = 128 fp8
// MFMAs = the cross terms of both taps) — the operand layout and scales are those of the ping-pong kernel's PP_F16C mode.
// LDS hazards (chunk n lives in buffer n % 3; B(n) = the barrier before the first read of chunk n's buffer):
//   write of chunk n+1 (taps 3..8 of chunk n) -> buffer (n+1) % 3, last read in phase C of the pair straddling n-2 | n-1,
//   which every wave has left before B(n); first read of it after B(n+1).
// What one wave per SIMD cannot do is hide an epilogue: VALU instructions between its MFMAs cost their full issue time
// (4 fillers per K = 128 MFMA pair: +12 %, tools/gpu_sw_bench.py history in DESIGN.md), and the SPADE epilogue (a quarter of
// a gamma|beta layer when exposed) is hidden only by a second wave on the SIMD.  So the planner keeps the gamma|beta convs
// on the ping-pong kernel and gives this one the long-K main convs, whose epilogue is 1-2 % of a tile.
#include "kernels.h"
#include <cstdlib>

namespace msr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct SwGeom {
    int tiles_x, tiles_y, tiles_n, tiles_mn;   // 16 x 16 pixel tiles per row / column, 128-column blocks, all tiles
    int walk_pb, walk_nb;                      // tile walk (kernels.h conv_walk_pick / MSR_WALK)
};

static constexpr int SW_HW = 18, SW_HP = 18 * 18, SW_BKP = 40, SW_HPB = SW_HP * SW_BKP;   // floats per halo buffer
#ifdef MSR_SW_STAMPS
static constexpr size_t SW_LDS = (size_t)3 * SW_HPB * sizeof(float) + 8192;                // + the stamp words (diagnostic build)
#else
static constexpr size_t SW_LDS = (size_t)3 * SW_HPB * sizeof(float);                       // 155,520 B
#endif

__device__ __forceinline__ float sw_row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    return v;
}

// Epilogue of one wave: acc[i][j] = D[4 channels of column block j][pixel (row i, x = lane & 15)].
// Column blocks of wave q: 16 * (4 * (q >> 1) + (q & 1) + 2 * j): for EPI_SPADE (columns interleaved 32 gamma | 32 beta per
// 64) j = 0 is gamma and j = 1 beta of the SAME 16 channels.  Moment slabs are the ping-pong kernel's: one per 4 rows x 16
// pixels (slab = 2 * (8-row tile) + half), so conv_stat_slabs / moments_from_slabs are unchanged.
template <int EPI, int OS>
__device__ __forceinline__ void sw_epilogue_body(const ConvParams& p, const SwGeom& g, f32x4 (&acc)[16][2], int wq, int lane,
                                                 int n0, int tx0, int ty0, int b0) {
    const int px = lane & 15, cg = lane >> 4;
    const int x = tx0 + px;
    const int cb0 = n0 + (wq >> 1) * 64 + (wq & 1) * 16 + 4 * cg;       // first of the lane's 4 columns in block j = 0
    float* const obase = p.out + (size_t)p.out_off + (size_t)b0 * p.out_pb + x * p.out_px;
    if constexpr (EPI == EPI_SPADE) {
        const int ch = ((n0 + (wq >> 1) * 64) >> 1) + (wq & 1) * 16 + 4 * cg;
        const float4 gq4 = *reinterpret_cast<const float4*>(p.bias + cb0);
        const float4 bq4 = *reinterpret_cast<const float4*>(p.bias + cb0 + 32);
        const float4 mq4 = *reinterpret_cast<const float4*>(p.mean + ch);
        const float4 sq4 = *reinterpret_cast<const float4*>(p.stdv + ch);
        const float gq[4] = {gq4.x, gq4.y, gq4.z, gq4.w}, bq[4] = {bq4.x, bq4.y, bq4.z, bq4.w};
        const float mq[4] = {mq4.x, mq4.y, mq4.z, mq4.w};
        float sq[4] = {sq4.x, sq4.y, sq4.z, sq4.w};
        constexpr bool split = OS != 0;
        if constexpr (split) {
#pragma unroll
            for (int k = 0; k < 4; ++k) sq[k] = 1.f / sq[k];          // as the ping-pong kernel: multiply by 1/sigma
        }
        const float* const abase = p.aux + (size_t)b0 * p.aux_pb + (x >> p.aux_shift) * p.aux_px + ch;
#pragma unroll
        for (int i0 = 0; i0 < 16; i0 += 8) {
            float4 xin[8];
#pragma unroll
            for (int ii = 0; ii < 8; ++ii)
                xin[ii] = *reinterpret_cast<const float4*>(abase + ((ty0 + i0 + ii) >> p.aux_shift) * p.aux_py);
#pragma unroll
            for (int ii = 0; ii < 8; ++ii) {
                const int i = i0 + ii;
                float* orow = obase + (ty0 + i) * p.out_py;
                const float xq[4] = {xin[ii].x, xin[ii].y, xin[ii].z, xin[ii].w};
                float v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float normalized = split ? (xq[k] - mq[k]) * sq[k] : (xq[k] - mq[k]) / sq[k];
                    const float t = (acc[i][0][k] + gq[k]) * normalized + (acc[i][1][k] + bq[k]);
                    v[k] = t >= 0.f ? t : t * p.slope;
                }
                if constexpr (OS == 4) msr_store_f16c4_dev(orow, ch, v[0], v[1], v[2], v[3]);
                else if constexpr (OS == 1) msr_store_split4_dev(orow, ch, v[0], v[1], v[2], v[3]);
                else *reinterpret_cast<float4*>(orow + ch) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = cb0 + 32 * j;
            const float4 b4 = *reinterpret_cast<const float4*>(p.bias + col);
            const float bq[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {                           // groups of 4 rows = one moment slab
                float4 res[4];
                if constexpr (EPI == EPI_RES) {
                    const float* abase = p.aux + (size_t)b0 * p.aux_pb + (x >> p.aux_shift) * p.aux_px + col;
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii)
                        res[ii] = *reinterpret_cast<const float4*>(abase + ((ty0 + q4 * 4 + ii) >> p.aux_shift) * p.aux_py);
                }
                float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) {
                    const int i = q4 * 4 + ii;
                    float d[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) d[k] = acc[i][j][k];
                    if constexpr (EPI == EPI_RES) { d[0] += res[ii].x; d[1] += res[ii].y; d[2] += res[ii].z; d[3] += res[ii].w; }
                    *reinterpret_cast<float4*>(obase + (ty0 + i) * p.out_py + col) =
                        make_float4(d[0] + bq[0], d[1] + bq[1], d[2] + bq[2], d[3] + bq[3]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) { s1[k] += d[k]; s2[k] += d[k] * d[k]; }
                }
                if (p.stat_partial) {
                    float mean[4], m2[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float t1 = sw_row16_sum(s1[k]), t2 = sw_row16_sum(s2[k]);
                        mean[k] = bq[k] + t1 * (1.f / 64.f);
                        const float t = t2 - t1 * t1 * (1.f / 64.f);
                        m2[k] = t > 0.f ? t : 0.f;
                    }
                    if (px == 0) {
                        const int st = (b0 * (g.tiles_y * 2) + ((ty0 + q4 * 4) >> 3)) * g.tiles_x + (tx0 >> 4);
                        float* o = p.stat_partial + (size_t)(st * 2 + (q4 & 1)) * 3 * p.N + col;
                        *reinterpret_cast<float4*>(o) = make_float4(64.f, 64.f, 64.f, 64.f);
                        *reinterpret_cast<float4*>(o + p.N) = make_float4(mean[0], mean[1], mean[2], mean[3]);
                        *reinterpret_cast<float4*>(o + 2 * p.N) = make_float4(m2[0], m2[1], m2[2], m2[3]);
                    }
                }
            }
        }
    }
}

template <int EPI>
__device__ __forceinline__ void sw_epilogue(const ConvParams& p, const SwGeom& g, f32x4 (&acc)[16][2], int wq, int lane,
                                            int n0, int tx0, int ty0, int b0) {
    if constexpr (EPI == EPI_SPADE) {
        if (p.out_split == 4) sw_epilogue_body<EPI, 4>(p, g, acc, wq, lane, n0, tx0, ty0, b0);
        else if (p.out_split == 1) sw_epilogue_body<EPI, 1>(p, g, acc, wq, lane, n0, tx0, ty0, b0);
        else sw_epilogue_body<EPI, 0>(p, g, acc, wq, lane, n0, tx0, ty0, b0);
    } else {
        sw_epilogue_body<EPI, 0>(p, g, acc, wq, lane, n0, tx0, ty0, b0);
    }
}

#ifdef SW_NOBAR   // what-if build (wrong results): the cost of the per-chunk barrier
#define SW_BARRIER() {}
#else
#define SW_BARRIER()                                           \
    {                                                          \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); \
        __builtin_amdgcn_s_barrier();                          \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); \
    }
#endif
#ifndef SW_UNROLL2
#define SW_UNROLL2 1
#endif
#ifndef SW_PD
#define SW_PD 6
#endif
#ifndef SW_PDC
#define SW_PDC 3
#endif

// F6 = true: PREC_F16C6 operands (kernels.h): the cross terms run on fp6 e2m3 pieces, 16 instead of 32 matrix-pipe cycles
// per K = 128 MFMA.  A lane then reads ONE 32-byte half of ONE tap's chunk row (the even tap's for lane groups 0 / 1, the odd
// tap's for 2 / 3; first / second half for even / odd groups): the same two 16-byte loads as before, the operand in the
// first six registers, the block scale in byte 0 of the seventh.
// NOX = true ("f16" mode, ConvParams.no_cross): the cross terms are left out — one fp16 product per element, per-product
// error 2^-11 instead of ~2^-15 (a declared-tolerance mode, see include/moonsr.h MSR_FLAG_F16_MAIN); the tensors keep the
// f16c image, the cross pieces are simply not read.
template <int EPI, bool F6 = false, bool NOX = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
conv_igemm_f16c_sw(const ConvParams p, const SwGeom g) {
    MSR_SATURATING_CONVERSIONS();
    constexpr int HW = SW_HW, HP = SW_HP, BKP = SW_BKP, HPB = SW_HPB;
    constexpr int NTHR = 256, BKC = 32;
    constexpr int H_ITEMS = (HP * 8 + NTHR - 1) / NTHR;                // 11 16-byte items per thread and chunk
    static_assert(H_ITEMS == 11, "halo staging is written for 11 items per thread");
    constexpr int PD = SW_PD;                                          // f16 fragments requested ahead of their MFMAs
    constexpr int PDC = SW_PDC;                                        // cross fragments (two reads each) ahead
    constexpr int HD = 3;                                              // taps between a halo item's request and its LDS store

    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wq = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 15, cg = lane >> 4;

    // persistent tile walk: exactly the ping-pong kernel's (XCD x owns a contiguous range of tiles; its workgroups take
    // consecutive tiles = same pixels, next column block, so halo and weights are shared in that XCD's L2)
    const int items = g.tiles_mn;
    const int slots = gridDim.x >> 3, xcd = blockIdx.x & 7;
    const int tq = items >> 3, tr = items & 7;
    const int cnt = tq + (xcd < tr ? 1 : 0);
    const int base = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
    int tile = blockIdx.x >> 3;
    if (tile >= cnt) return;

    int n0, tx0, ty0, b0;
    unsigned h_tile, w_tile;
#define SW_DECODE(T_, N0_, TX_, TY_, B_, HT_, WT_)                                               \
    {                                                                                            \
        const int t_ = (T_);                                                                     \
        int tn_, tmi_;                                                                           \
        MSR_WALK(g, t_, tn_, tmi_)                                                               \
        TX_ = (tmi_ % g.tiles_x) << 4;                                                           \
        tmi_ /= g.tiles_x;                                                                       \
        TY_ = (tmi_ % g.tiles_y) << 4;                                                           \
        B_ = tmi_ / g.tiles_y;                                                                   \
        N0_ = tn_ * 128;                                                                         \
        HT_ = (unsigned)((B_) * p.in_pb + (TY_) * p.in_py + (TX_) * p.Cin) * 4u;                 \
        WT_ = (unsigned)((N0_) * p.Cin) * 4u;                                                    \
    }
    SW_DECODE(base + tile, n0, tx0, ty0, b0, h_tile, w_tile)

    // halo staging: 16-byte item q of a thread is halo pixel hp0 + 32 q (18 x 18 pixels, row-major), segment tid & 7 of its
    // 128-byte chunk row; items past the end (q = 10, hp >= 324) duplicate pixel 323.  Offsets are recomputed per item (6
    // VALU instructions in an f16 phase) instead of held in 22 registers.
    const int hp0 = tid >> 3, hseg = tid & 7;
    const int h_l0 = hp0 * BKP + hseg * 4;
    const bool h_live = !NOX || hseg < 4;          // NOX: only the fp16 half of a chunk row (bytes 0..63) is staged
#define SW_HPIX(q) ((q) < 10 ? hp0 + 32 * (q) : min(hp0 + 32 * (q), HP - 1))
#define SW_HGOFF(q) ((((SW_HPIX(q) * 3641) >> 16) * p.in_py + (SW_HPIX(q) - ((SW_HPIX(q) * 3641) >> 16) * HW) * p.Cin + hseg * 4) * 4)
#define SW_HLOFF(q) ((q) < 10 ? h_l0 + 32 * (q) * BKP : SW_HPIX(q) * BKP + hseg * 4)
    // weight rows of the wave: column block j -> rows 64 * (wq >> 1) + 16 * (wq & 1) + 32 * j + (lane & 15)
    int b_voff[2], b_xoff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row_ = (((wq >> 1) * 64 + (wq & 1) * 16 + 32 * j + px) * p.Cin) * 4;
        b_voff[j] = row_ + 16 * cg;
        b_xoff[j] = F6 ? row_ + 64 + 32 * (cg & 1) : row_ + 16 * cg + 64;     // the cross piece(s) of the lane
    }
    const int a_lane = px * BKP + 4 * cg;                               // float offset of the lane inside a fragment row
    const int x_lane = px * BKP + 16 + 8 * (cg & 1);                    // F6: the lane's 32-byte half of a chunk row

    const unsigned w_tap_bytes = (unsigned)((size_t)p.N * p.Cin * sizeof(float));
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.in), 0, (int)((size_t)p.B * p.in_pb * sizeof(float)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wt = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.wt), 0, (int)(9u * w_tap_bytes), 0x00020000);
#define SW_BUFLD(rs, voff, soff) __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (int)(soff), 0)

    const int asc = (cg & 1) ? 0x74747474 : 0x7F7F7F7F;                 // e8m0 of the activation pieces: l8 = x_lo * 2^11, h8 = x_hi
    int wsc[2];
    f32x4 acc[16][2];
    // weights of the current tap pair and of the next one (in flight)
    i32x4 bE[2], bO[2], xE[2], xO[2], nbE[2], nbO[2], nxE[2], nxO[2];
    // pixel fragments: f16 of the even / odd tap, cross pieces of both
    i32x4 fa[16], fb[16], ce[16], co[16];
    i32x4 rh[HD][2];                                                    // halo items in flight

    const int ppi = p.Cin / (2 * BKC);
    int rb = 0;                                                         // ring buffer of the body's first chunk

    // byte offset (from the tile's weights) of K-step T of the body: T < 18 inside the pair, 18 / 19 = the first two
    // taps of whatever comes after it (the next pair, or the next tile)
#define SW_WSOFF(T) ((T) < 18 ? w_cur + (unsigned)((T) / 9) * (BKC * 4) + (unsigned)((T) % 9) * w_tap_bytes \
                              : w_after + (unsigned)((T) - 18) * w_tap_bytes)
#define SW_LOAD_B(dstE, dstO, dxE, dxO, T, j)                                                    \
    {                                                                                            \
        dstE[j] = SW_BUFLD(rs_wt, b_voff[j], SW_WSOFF(T));                                       \
        dstO[j] = SW_BUFLD(rs_wt, b_voff[j], SW_WSOFF((T) + 1));                                 \
        if constexpr (NOX) {                                                                     \
        } else if constexpr (F6) {                                                                      \
            /* both 16-byte halves of the lane's piece, from the even tap's row (lane groups 0, 1) or the odd tap's (2, 3); \
               the scalar offset is the smaller of the two taps' (the other one's excess rides in the lane offset) */ \
            const unsigned lo_ = min(SW_WSOFF(T), SW_WSOFF((T) + 1));                            \
            const int dv_ = (int)((cg < 2 ? SW_WSOFF(T) : SW_WSOFF((T) + 1)) - lo_);             \
            dxE[j] = SW_BUFLD(rs_wt, b_xoff[j] + dv_, lo_);                                      \
            dxO[j] = SW_BUFLD(rs_wt, b_xoff[j] + dv_ + 16, lo_);                                 \
        } else {                                                                                 \
            dxE[j] = SW_BUFLD(rs_wt, b_xoff[j], SW_WSOFF(T));                                    \
            dxO[j] = SW_BUFLD(rs_wt, b_xoff[j], SW_WSOFF((T) + 1));                              \
        }                                                                                        \
    }
    // pixel fragment i of K-step T (T = 18.. : the next body's first chunk)
#define SW_APTR(T) ((T) < 9 ? A0 : (T) < 18 ? A1 : A2)
#define SW_AOFF(T, i) ((((i) + (((T) % 9) / 3)) * HW + (((T) % 9) % 3)) * BKP)
#define SW_RD_F(dst, T, i) dst = *reinterpret_cast<const i32x4*>(SW_APTR(T) + SW_AOFF(T, i))
#define SW_RD_X(dst, T, i) dst = *reinterpret_cast<const i32x4*>(SW_APTR(T) + SW_AOFF(T, i) + 16)
    // F6: the lane's chunk-row half of pixel fragment i of tap pair (TE, TE + 1): XP = per-lane pointer of that pair (below)
#define SW_XP(TE) ((cg < 2 ? SW_APTR(TE) + SW_AOFF(TE, 0) : SW_APTR((TE) + 1) + SW_AOFF((TE) + 1, 0)) - a_lane + x_lane)
#define SW_RD_X6(dlo, dhi, XP, i)                                                                \
    {                                                                                            \
        dlo = *reinterpret_cast<const i32x4*>((XP) + (i) * HW * BKP);                            \
        dhi = *reinterpret_cast<const i32x4*>((XP) + (i) * HW * BKP + 4);                        \
    }
#define SW_F16(v) __builtin_bit_cast(f16x8, v)
#define SW_CAT8(lo, hi) __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7)
    // halo staging of K-step T: chunk (T / 9) + 1 of the body (chunk 2 = whatever comes after the pair) travels while
    // chunk T / 9 is multiplied: items 2t, 2t + 1 are requested on tap t = 0..5 and stored to LDS on tap t + HD.  HD = 3 taps
    // rather than 1: vmcnt counts loads and stores in ONE in-order queue, so after a tile's epilogue stores every wait for a
    // younger load also waits for those stores; with three taps of slack the first such wait comes ~2000 cycles later.
#define SW_HALO(T)                                                                               \
    {                                                                                            \
        constexpr int t_ = (T) % 9;                                                              \
        float* const w_ = (T) < 9 ? W1 : W2;                                                     \
        const unsigned hs_ = (T) < 9 ? h_cur + BKC * 4 : h_after;                                \
        if constexpr (t_ >= HD && t_ <= 5 + HD) {                                                \
            constexpr int q_ = 2 * (t_ - HD);                                                    \
            if (h_live) {                                                                        \
            *reinterpret_cast<i32x4*>(w_ + SW_HLOFF(q_)) = rh[(t_ - HD) % HD][0];                \
            if constexpr (q_ + 1 < H_ITEMS) *reinterpret_cast<i32x4*>(w_ + SW_HLOFF((q_ + 1) % H_ITEMS)) = rh[(t_ - HD) % HD][1]; \
            }                                                                                    \
        }                                                                                        \
        if constexpr (t_ <= 5) {                                                                 \
            if (h_live) {                                                                        \
            rh[t_ % HD][0] = SW_BUFLD(rs_in, SW_HGOFF(2 * t_), hs_);                             \
            if constexpr (2 * t_ + 1 < H_ITEMS) rh[t_ % HD][1] = SW_BUFLD(rs_in, SW_HGOFF((2 * t_ + 1) % H_ITEMS), hs_); \
            }                                                                                    \
        }                                                                                        \
    }
    // Diagnostic build only (-DMSR_SW_STAMPS=1: one s_memtime stamp per tap pair, =2: per phase; tools/gpu_sw_stamps.py):
    // lane 0 of every wave of workgroup 8 stamps one body of its first tile into the LDS words behind the halo ring.
    // (A stamp waits for lgkmcnt(0): level 2 slows the stream down by a third.)
#ifdef MSR_SW_STAMPS
    unsigned* const dbg = reinterpret_cast<unsigned*>(smem + 3 * HPB) + wq * 64;
    int dbg_n = 0;
    bool dbg_on = false;
    if (lane == 0) dbg[63] = 0;
#define SW_STAMP(LVL)                                                                            \
    if (MSR_SW_STAMPS >= (LVL) && dbg_on && lane == 0 && dbg_n < 64) dbg[dbg_n++] = (unsigned)__builtin_amdgcn_s_memtime();
#else
#define SW_STAMP(LVL)
#endif
    // Phase E of pair U: tap T = 2U, f16 fragments fa; requests fa ahead, then fb of tap T + 1; the next pair's weights
#define SW_PHASE_E(U)                                                                            \
    {                                                                                            \
        constexpr int T = 2 * (U);                                                               \
        SW_STAMP(1)                                                                              \
        SW_HALO(T)                                                                               \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                         \
            if (T + 1 == 9 && i == 16 - PD) SW_BARRIER()                                         \
            if (i + PD < 16) SW_RD_F(fa[(i + PD) & 15], T, (i + PD) & 15);                       \
            else SW_RD_F(fb[(i + PD) & 15], T + 1, (i + PD) & 15);                               \
            if (i == 0) SW_LOAD_B(nbE, nbO, nxE, nxO, T + 2, 0)                                  \
            if (i == 8) SW_LOAD_B(nbE, nbO, nxE, nxO, T + 2, 1)                                  \
            acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(SW_F16(bE[0]), SW_F16(fa[i]), acc[i][0], 0, 0, 0); \
            acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(SW_F16(bE[1]), SW_F16(fa[i]), acc[i][1], 0, 0, 0); \
            __builtin_amdgcn_sched_barrier(0);                                                   \
        }                                                                                        \
    }
    // Phase O: tap T = 2U + 1, fragments fb; requests fb ahead, then the first cross pieces of the pair
#define SW_PHASE_O(U)                                                                            \
    {                                                                                            \
        constexpr int T = 2 * (U) + 1;                                                           \
        SW_STAMP(2)                                                                              \
        SW_HALO(T)                                                                               \
        const float* const xp_ = SW_XP(T - 1);                                                   \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                         \
            if (i + PD < 16) SW_RD_F(fb[(i + PD) & 15], T, (i + PD) & 15);                       \
            if constexpr (NOX) {    /* no phase C: its barrier and its requests of the next pair's fa move here */ \
                if (T + 1 == 18 && i == 16 - PD) SW_BARRIER()                                    \
                if (i >= 16 - PD) SW_RD_F(fa[(i + PD) & 15], T + 1, (i + PD) & 15);              \
            } else if (i >= 16 - PDC) {                                                          \
                if constexpr (F6) {                                                              \
                    SW_RD_X6(ce[(i + PDC) & 15], co[(i + PDC) & 15], xp_, (i + PDC) & 15)        \
                } else {                                                                         \
                    SW_RD_X(ce[(i + PDC) & 15], T - 1, (i + PDC) & 15);                          \
                    SW_RD_X(co[(i + PDC) & 15], T, (i + PDC) & 15);                              \
                }                                                                                \
            }                                                                                    \
            acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(SW_F16(bO[0]), SW_F16(fb[i]), acc[i][0], 0, 0, 0); \
            acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(SW_F16(bO[1]), SW_F16(fb[i]), acc[i][1], 0, 0, 0); \
            __builtin_amdgcn_sched_barrier(0);                                                   \
        }                                                                                        \
    }
    // Phase C: the cross terms of taps 2U and 2U + 1 (K = 128 block-scaled fp8 MFMA); requests the cross pieces ahead,
    // then fa of the next pair's first tap
#define SW_PHASE_C(U)                                                                            \
    {                                                                                            \
        constexpr int T = 2 * (U);                                                               \
        SW_STAMP(2)                                                                              \
        const i32x8 wq0_ = SW_CAT8(xE[0], xO[0]), wq1_ = SW_CAT8(xE[1], xO[1]);                  \
        const float* const xp_ = SW_XP(T);                                                       \
        if constexpr (!NOX) {                                                                    \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                         \
            if (T + 2 == 18 && i == 16 - PD) SW_BARRIER()                                        \
            if (i + PDC < 16) {                                                                  \
                if constexpr (F6) {                                                              \
                    SW_RD_X6(ce[(i + PDC) & 15], co[(i + PDC) & 15], xp_, (i + PDC) & 15)        \
                } else {                                                                         \
                    SW_RD_X(ce[(i + PDC) & 15], T, (i + PDC) & 15);                              \
                    SW_RD_X(co[(i + PDC) & 15], T + 1, (i + PDC) & 15);                          \
                }                                                                                \
            }                                                                                    \
            if (i >= 16 - PD) SW_RD_F(fa[(i + PD) & 15], T + 2, (i + PD) & 15);                  \
            const i32x8 aq_ = SW_CAT8(ce[i], co[i]);                                             \
            if constexpr (F6) {     /* fp6 x fp6: operands in registers 0..5, the block scales in byte 0 of register 6 */ \
                acc[i][0] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wq0_, aq_, acc[i][0], 2, 2, 0, wq0_[6], 0, aq_[6]); \
                acc[i][1] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wq1_, aq_, acc[i][1], 2, 2, 0, wq1_[6], 0, aq_[6]); \
            } else {                                                                             \
                acc[i][0] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wq0_, aq_, acc[i][0], 0, 0, 0, wsc[0], 0, asc); \
                acc[i][1] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wq1_, aq_, acc[i][1], 0, 0, 0, wsc[1], 0, asc); \
            }                                                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                   \
        }                                                                                        \
        }                                                                                        \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) { bE[j] = nbE[j]; bO[j] = nbO[j]; xE[j] = nxE[j]; xO[j] = nxO[j]; } \
    }
#define SW_PAIR(U) SW_PHASE_E(U) SW_PHASE_O(U) SW_PHASE_C(U)
#define SW_BODY() SW_PAIR(0) SW_PAIR(1) SW_PAIR(2) SW_PAIR(3) SW_PAIR(4) SW_PAIR(5) SW_PAIR(6) SW_PAIR(7) SW_PAIR(8)
    // per-body addresses: chunk pair PR of the tile (LAST: what follows is the next tile), the three ring buffers
#define SW_SETUP(PR, LAST)                                                                       \
    const unsigned w_cur = w_tile + (unsigned)(PR) * (2 * BKC * 4);                              \
    const unsigned h_cur = h_tile + (unsigned)(PR) * (2 * BKC * 4);                              \
    const unsigned w_after = (LAST) ? w_next : w_cur + 2 * BKC * 4;                              \
    const unsigned h_after = (LAST) ? h_next : h_cur + 2 * BKC * 4;                              \
    const int r1 = rb == 2 ? 0 : rb + 1, r2 = r1 == 2 ? 0 : r1 + 1;                              \
    const float* const A0 = smem + rb * HPB + a_lane;                                            \
    const float* const A1 = smem + r1 * HPB + a_lane;                                            \
    const float* const A2 = smem + r2 * HPB + a_lane;                                            \
    float* const W1 = smem + r1 * HPB;                                                           \
    float* const W2 = smem + r2 * HPB;

    // ---- prologue of the workgroup's first tile: chunk 0 of its halo into ring buffer 0, the weights of taps 0 and 1,
    // the first PD pixel fragments
    {
        i32x4 t[H_ITEMS];
#pragma unroll
        for (int q = 0; q < H_ITEMS; ++q) if (h_live) t[q] = SW_BUFLD(rs_in, SW_HGOFF(q), h_tile);
#pragma unroll
        for (int q = 0; q < H_ITEMS; ++q) if (h_live) *reinterpret_cast<i32x4*>(smem + SW_HLOFF(q)) = t[q];
    }
    {
        const unsigned w_cur = w_tile, w_after = w_tile;
        SW_LOAD_B(bE, bO, xE, xO, 0, 0)
        SW_LOAD_B(bE, bO, xE, xO, 0, 1)
    }
    SW_BARRIER()
    {
        const float* const A0 = smem + a_lane;
        const float* const A1 = A0;
        const float* const A2 = A0;
#pragma unroll
        for (int i = 0; i < PD; ++i) SW_RD_F(fa[i], 0, i);
    }

    for (;;) {
        const int tnext = tile + slots;
        const bool has_next = tnext < cnt;
        int n0n, tx0n, ty0n, b0n;
        unsigned h_next, w_next;
        SW_DECODE(base + (has_next ? tnext : tile), n0n, tx0n, ty0n, b0n, h_next, w_next)
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {       // byte 0 = e8m0 of the channel's w_lo pieces (even lane groups), byte 1 = of its w_hi pieces
            const int w_ = F6 ? 0 : p.wexp[n0 + (wq >> 1) * 64 + (wq & 1) * 16 + 32 * j + px];      // F6: the scales ride in the image
            wsc[j] = (((cg & 1) ? (w_ >> 8) : w_) & 0xFF) * 0x01010101;
        }
#if SW_UNROLL2   // two bodies per iteration (Cin % 128 == 0): halves the cost of the allocator's accumulator rotation
        for (int pr = 0; pr < ppi; pr += 2) {
            { SW_SETUP(pr, false) SW_BODY() rb = r2; }
            { SW_SETUP(pr + 1, pr + 2 == ppi) SW_BODY() rb = r2; }
        }
#else
        for (int pr = 0; pr < ppi; ++pr) {
            SW_SETUP(pr, pr == ppi - 1)
#ifdef MSR_SW_STAMPS
            dbg_on = blockIdx.x == 8 && pr == (ppi > 2 ? 2 : 1) && dbg_n == 0;
#endif
            SW_BODY()
            SW_STAMP(1)
#ifdef MSR_SW_STAMPS
            dbg_on = false;
#endif
            rb = r2;
        }
#endif
#ifdef MSR_SW_STAMPS
        const unsigned te0_ = (unsigned)__builtin_amdgcn_s_memtime();
#endif
#ifdef SW_NOEPI   // what-if build (wrong results): the cost of the exposed epilogue
        if (p.B < 0)
#endif
        sw_epilogue<EPI>(p, g, acc, wq, lane, n0, tx0, ty0, b0);
#ifdef MSR_SW_STAMPS
        if (blockIdx.x == 8 && lane == 0 && dbg_n > 0 && dbg_n < 63 && dbg[63] == 0) dbg[63] = (unsigned)__builtin_amdgcn_s_memtime() - te0_;
#endif
        if (!has_next) break;
        tile = tnext;
        n0 = n0n; tx0 = tx0n; ty0 = ty0n; b0 = b0n;
        h_tile = h_next;
        w_tile = w_next;
    }
#ifdef MSR_SW_STAMPS
    if (blockIdx.x == 8 && lane == 0) {
        for (int k = 0; k + 1 < dbg_n; ++k) printf("wave %d stamp %2d: %5u cycles\n", wq, k, dbg[k + 1] - dbg[k]);
        printf("wave %d body total %u, epilogue (issue) %u\n", wq, dbg[dbg_n - 1] - dbg[0], dbg[63]);
    }
#endif
#undef SW_STAMP
#undef SW_DECODE
#undef SW_HPIX
#undef SW_HGOFF
#undef SW_HLOFF
#undef SW_BUFLD
#undef SW_WSOFF
#undef SW_LOAD_B
#undef SW_APTR
#undef SW_AOFF
#undef SW_RD_F
#undef SW_RD_X
#undef SW_XP
#undef SW_RD_X6
#undef SW_F16
#undef SW_CAT8
#undef SW_HALO
#undef SW_PHASE_E
#undef SW_PHASE_O
#undef SW_PHASE_C
#undef SW_PAIR
#undef SW_BODY
#undef SW_SETUP
}

hipError_t conv_sw_init() {
    hipError_t e;
#define SW_SET(EPI)                                                                                          \
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_f16c_sw<EPI>),                    \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)SW_LDS)) != hipSuccess)    \
        return e;
    SW_SET(EPI_BIAS) SW_SET(EPI_RES) SW_SET(EPI_SPADE)
#undef SW_SET
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_f16c_sw<EPI_BIAS, true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)SW_LDS)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_f16c_sw<EPI_RES, true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)SW_LDS)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_f16c_sw<EPI_BIAS, false, true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)SW_LDS)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_f16c_sw<EPI_RES, false, true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)SW_LDS)) != hipSuccess) return e;
    return hipSuccess;
}

// Whole-tile f16c launches only (the planner's rule for PREC_F16C): 3 x 3 stride 1, r >= 16 a power of two, N % 128 == 0,
// Cin % 64 == 0, no K split.
hipError_t launch_conv_f16c_sw(const ConvParams& p, int epi, hipStream_t s) {
    auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
    if (epi == EPI_SPADE && p.out_split != 0 && p.out_split != 1 && p.out_split != 4) return hipErrorInvalidValue;
    const bool f6 = p.prec == PREC_F16C6;
    if ((p.prec != PREC_F16C && !f6) || (!f6 && !p.wexp) || p.ksplit > 1 || p.stride != 1 || p.KH != 3 || p.KW != 3) return hipErrorInvalidValue;
    if (f6 && epi == EPI_SPADE) return hipErrorInvalidValue;          // the fp6 form exists for the main convs
    if (!pow2(p.Hout) || !pow2(p.Wout) || p.Hout < 16 || p.Wout < 16 || p.N % 128 || p.Cin % (SW_UNROLL2 ? 128 : 64)) return hipErrorInvalidValue;
    if ((size_t)p.B * p.in_pb * sizeof(float) >= ((size_t)1 << 31)) return hipErrorInvalidValue;   // buffer descriptor range
    SwGeom g;
    g.tiles_x = p.Wout / 16;
    g.tiles_y = p.Hout / 16;
    g.tiles_n = p.N / 128;
    g.tiles_mn = g.tiles_x * g.tiles_y * p.B * g.tiles_n;
    conv_walk_pick(g.tiles_x * g.tiles_y * p.B, g.tiles_n, &g.walk_pb, &g.walk_nb);
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorInvalidValue;
        n_cu = prop.multiProcessorCount & ~7;
        if (n_cu < 8) n_cu = 8;
    }
    const int grid = g.tiles_mn < n_cu ? ((g.tiles_mn + 7) & ~7) : n_cu;
    if (f6) {
        if (epi == EPI_BIAS) conv_igemm_f16c_sw<EPI_BIAS, true><<<grid, 256, SW_LDS, s>>>(p, g);
        else conv_igemm_f16c_sw<EPI_RES, true><<<grid, 256, SW_LDS, s>>>(p, g);
        return hipGetLastError();
    }
    if (p.no_cross) {
        if (epi == EPI_BIAS) conv_igemm_f16c_sw<EPI_BIAS, false, true><<<grid, 256, SW_LDS, s>>>(p, g);
        else if (epi == EPI_RES) conv_igemm_f16c_sw<EPI_RES, false, true><<<grid, 256, SW_LDS, s>>>(p, g);
        else return hipErrorInvalidValue;          // the gamma|beta convs of the mode run conv_gb_resident
        return hipGetLastError();
    }
    switch (epi) {
        case EPI_BIAS: conv_igemm_f16c_sw<EPI_BIAS><<<grid, 256, SW_LDS, s>>>(p, g); break;
        case EPI_RES: conv_igemm_f16c_sw<EPI_RES><<<grid, 256, SW_LDS, s>>>(p, g); break;
        case EPI_SPADE: conv_igemm_f16c_sw<EPI_SPADE><<<grid, 256, SW_LDS, s>>>(p, g); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace msr
