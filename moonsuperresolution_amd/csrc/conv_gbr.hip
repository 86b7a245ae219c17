// conv_gb_resident — one SPADE layer's whole modulation path in ONE kernel (round 3):
//     mask = nearest_resize(source)            spade.py:17   (tf.image.resize, method="nearest")
//     e    = relu(conv3x3(mask, 2 -> 128))     spade.py:18   (self.conv, activation="relu")
//     g|b  = conv3x3(e, 128 -> 2C)             spade.py:19-20 (conv_gamma, conv_beta as ONE GEMM, kernels.h EPI_SPADE)
//     out  = leaky_relu(g * (x - mean) / std + b)   spade.py:21-24 + blocks.py:30-34
// for the launches that fill the chip (r >= 32 at the BASELINE sizes: 98 % of the gamma|beta FLOPs).
//
// Why: the gamma|beta convs are half of a call's FLOPs and ran the ping-pong kernel at 53-56 % matrix-pipe busy with NO memory
// resource near its limit (profiles/r03_conv_colimiters_f16c_before.txt: TA 36 %, LDS 38 %): the loss was the schedule — two
// barriers per K-step, an even / odd step imbalance, tile ends every 36 K-steps.  Their K is only 9 x 128, so the input of a
// 16 x 16 pixel tile is small enough to stay in LDS for ALL output channels, and it is a function of the call's 2-channel
// input alone, so it does not have to exist in HBM at all:
//   * phase 1 (per work item): the 18 x 18 pixel halo of the 128-channel embedding is COMPUTED here from a 20 x 20 patch of the
//     resized mask (fp32 FMA chain: bias, then (ky, kx, c) order) and written to LDS as the f16c6 operand (kernels.h PREC_F16C6:
//     fp16 main term + fp6 e2m3 cross pieces, one power-of-two scale per pixel and 32-channel chunk).  The fp6 pieces are what
//     makes it fit: 324 px x 4 chunks x (64 + 24 + 24) B = 145 KB; with fp8 pieces it is 166 KB.  The 15 mask-embedding kernels
//     of a call (1.33 GB written, then re-read ~3x, 1.47 ms of kernel time on the auxiliary stream) are gone for these layers.
//   * phase 2: the workgroup sweeps its channel blocks (128 GEMM columns each) over the resident halo with the stream kernel's
//     body (conv_sw.hip): four waves, one per SIMD, wave q owns all 256 pixels x 32 columns, its weights go global -> registers
//     a tap pair ahead, pixel fragments stream from LDS through a register ring.  No LDS write, no barrier and no halo load
//     inside the sweep: the only vector-memory traffic is the weight stream (16 KB per tap pair and wave).
//     A tap pair = 32 + 32 f16 MFMAs (x_hi * w_hi of two taps) + 32 block-scaled K = 128 fp6 MFMAs (16 cycles each: both cross
//     terms of both taps) = 1.5 MFMA-equivalents per product instead of f16c's 2.
//   * epilogue per channel block: the SPADE modulation, written as the consumer's f16c chunk image; a wave holds 16 of a
//     chunk's 32 channels, so the 64 bytes a pixel gets from it (32 B fp16 | 16 B h8 | 16 B l8) are assembled in a private LDS
//     line and leave as ONE 16-byte store per lane and tile row.
// LDS (plane layout, every plane has a pixel pitch of 336 = 324 + 12):
//   F[chunk][piece 0..3][336] x 16 B   fp16 of channels 8 * piece .. + 7          86,016 B   (ds_read_b128: 16 consecutive
//                                       pixels of two planes whose distance is 0 mod 16 slots -> conflict-free for any base)
//   X[chunk][kind h6 | l6][j 0..2][336] x 8 B   the j-th 8 bytes of a 24-byte fp6 piece   64,512 B   (ds_read_b64: pitch
//                                       16 mod 32 -> the two kinds of a 32-lane group use disjoint bank halves)
//   SC[chunk][kind][336] x 1 B          e8m0 of the piece (h6: E, l6: E - 11)      2,688 B
//   stage[wave] 2,560 B                 epilogue line assembly (phase 1: the 20 x 20 x 2 mask patch)   -> 163,456 B
#include "kernels.h"
#include <cstdlib>

namespace msr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x6 __attribute__((ext_vector_type(6)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

static constexpr int GB_PH = 336;
static constexpr int GB_F_BYTES = 16 * GB_PH * 16;
static constexpr int GB_XA_BYTES = 8 * GB_PH * 16;                   // [chunk][kind][336] x 16 B: bytes 0..15 of a 24-byte fp6 piece
static constexpr int GB_XB_BYTES = 8 * GB_PH * 8;                    // [chunk][kind][336] x 8 B: bytes 16..23
static constexpr int GB_X_BYTES = GB_XA_BYTES + GB_XB_BYTES;
static constexpr int GB_SC_BYTES = 3 * 4 * 2 * 18 * 16;              // [row shift][chunk][kind][halo column] x 16 B (below)
static constexpr int GB_STAGE_OFF = GB_F_BYTES + GB_X_BYTES + GB_SC_BYTES;
static constexpr int GB_STAGE_WAVE = 1504;                           // one 16 x 80-byte line image per wave (>= 1280)
static constexpr int GB_LINE = 80;                                   // bytes per staged pixel line (64 used)
#ifdef MSR_GB_STAMPS
static constexpr size_t GB_LDS = 163840;                             // the stamp words take the last 384 bytes
#else
static constexpr size_t GB_LDS = GB_STAGE_OFF + 4 * GB_STAGE_WAVE;   // 163,456 B
#endif
static_assert(GB_STAGE_OFF + 4 * GB_STAGE_WAVE <= 163840 - 384, "LDS budget");

struct GbrGeom {
    int tiles_x, tiles_y;     // 16 x 16 pixel tiles per row / column
    int tiles_p;              // pixel tiles = tiles_x * tiles_y * B
    int nr;                   // channel blocks (128 columns) per work item
    int items;                // tiles_p * (N / 128 / nr); item = range * tiles_p + pixel tile (pixel tile fastest: the
                              // workgroups of an XCD sweep the same weights at the same time)
};

#ifndef GB_PD
#define GB_PD 6
#endif
#ifndef GB_PDC
#define GB_PDC 3
#endif

// NOX = true (GbrParams.no_cross, the "f16" mode): the fp6 cross terms are left out of the sweep — one fp16 product per
// element (per-product error 2^-11; declared tolerance, include/moonsr.h MSR_FLAG_F16_MAIN).  Phase 1 and the weight
// stream are unchanged; the cross pieces are not read.
template <bool NOX>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
conv_gb_resident(const GbrParams p, const GbrGeom g) {
    MSR_SATURATING_CONVERSIONS();
    constexpr int PH = GB_PH, HW = 18, HP = 324;
    constexpr int PD = GB_PD, PDC = GB_PDC;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wq = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 15, cg = lane >> 4;

    // persistent walk: XCD x owns a contiguous range of items, its workgroups take consecutive items of it
    const int slots = gridDim.x >> 3, xcd = blockIdx.x & 7;
    const int tq = g.items >> 3, tr = g.items & 7;
    const int cnt = tq + (xcd < tr ? 1 : 0);
    const int base = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;

    // ---- lane constants of the sweep ----
    // fp16 fragment of pixel row i, tap (dy, dx), chunk c: plane (c, piece cg), pixel (i + dy) * 18 + dx + px
    const char* const FL0 = smem + (cg * PH + px) * 16;                          // chunks 0, 1
    const char* const FL1 = FL0 + 2 * 4 * PH * 16;                               // chunks 2, 3 (ds offsets are 16 bits)
    // fp6 piece of the lane: kind = cg & 1 (h6 | l6) of the pair's even tap (cg < 2) or odd tap
    // A pair's pieces are the even tap's for lane groups 0, 1 and the odd tap's for 2, 3: the odd tap's halo pixel is 1 (next
    // column), 16 (next row: 18 - 2) or, from tap 8 to tap 0 of the next chunk, one chunk minus 38 pixels further; with that
    // difference in the lane pointer the pair's own position is an immediate offset of the ds_read
    const char* const XA = smem + GB_F_BYTES + ((cg & 1) * PH + px) * 16;
    const char* const XB = smem + GB_F_BYTES + GB_XA_BYTES + ((cg & 1) * PH + px) * 8;
    const char* const XAd[3] = {XA + (cg < 2 ? 0 : 1 * 16), XA + (cg < 2 ? 0 : 16 * 16), XA + (cg < 2 ? 0 : (2 * PH - 38) * 16)};
    const char* const XBd[3] = {XB + (cg < 2 ? 0 : 1 * 8), XB + (cg < 2 ? 0 : 16 * 8), XB + (cg < 2 ? 0 : (2 * PH - 38) * 8)};
    // block scales: SC[row shift dy][chunk][kind][halo column hx] x 16 bytes, byte i = e8m0 of halo pixel (i + dy, hx): the 16
    // scales a lane needs for one tap (rows i + dy, column px + dx) are ONE aligned 16-byte read per tap pair, and fragment i
    // takes byte i & 3 of register i >> 2 through the MFMA's scale byte select
    const char* const SL = smem + GB_F_BYTES + GB_X_BYTES + ((cg & 1) * 18 + px) * 16;
    const char* const SLd[3] = {SL + (cg < 2 ? 0 : 16), SL + (cg < 2 ? 0 : (4 * 2 * 18 - 2) * 16),
                                SL + (cg < 2 ? 0 : (2 * 18 - 2 * 4 * 2 * 18 - 2) * 16)};
    char* const stage = smem + GB_STAGE_OFF + wq * GB_STAGE_WAVE;
    // Weight stream (GbrParams.wt, built by the host: api.hip gbr_weight_stream / ops.gbr_weight_image): for channel block nt,
    // wave q and tap pair P (K-steps 2P, 2P + 1 of the 36-step chunk-major sequence) 8 KB = [column block j][piece][lane] x 16 B
    // with piece 0 / 1 = the fp16 fragments of the even / odd tap, 2 / 3 = the two halves of the lane's fp6 piece: every load
    // instruction reads 1 KB of consecutive bytes (8 whole cache lines; the [tap][row][chunk] image made the fp6 loads touch
    // 32 lines each: texture addresser 75 % busy, 40 % address-stalled, profiles/r03_conv_colimiters_f16c_gbr_v1.txt).
    // Column block j of wave q = rows 64 * (q >> 1) + 16 * (q & 1) + 32 * j + px of the channel block (conv_sw.hip's map).
    constexpr int CIN = 128;
    const int wv[2] = {lane * 16, lane * 16 + 4096};
    const unsigned w_stream_bytes = (unsigned)((size_t)9 * p.N * CIN * sizeof(float));
    const __amdgpu_buffer_rsrc_t rs_wt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wt), 0, (int)w_stream_bytes, 0x00020000);
#define GB_BUFLD(voff, soff) __builtin_amdgcn_raw_buffer_load_b128(rs_wt, voff, (int)(soff), 0)

    f32x4 acc[16][2];
    i32x4 bE[2], bO[2], xE[2], xO[2], nbE[2], nbO[2], nxE[2], nxO[2];
    i32x4 fa[16], fb[16];
    i32x4 xa[16];
    i32x2 xb[16];
    i32x4 xsc;                                    // the lane's 16 block scales of the current tap pair

#ifdef MSR_GB_STAMPS
    unsigned* const dbg = reinterpret_cast<unsigned*>(smem + 163840 - 384) + wq * 24;
    int dbg_n = 0;
#define GB_STAMP() { if (blockIdx.x == 8 && lane == 0 && dbg_n < 24) dbg[dbg_n++] = (unsigned)__builtin_amdgcn_s_memtime(); }
#define GB_STAMP2() { if (MSR_GB_STAMPS == 2) GB_STAMP() }
#define GB_STAMP3() { if (MSR_GB_STAMPS == 3) GB_STAMP() }
#else
#define GB_STAMP() {}
#define GB_STAMP2() {}
#define GB_STAMP3() {}
#endif

    for (int it = blockIdx.x >> 3; it < cnt; it += slots) {
        const int item = base + it;
        const int rng = item / g.tiles_p;
        int tmi = item - rng * g.tiles_p;
        const int tx0 = (tmi % g.tiles_x) << 4;
        tmi /= g.tiles_x;
        const int ty0 = (tmi % g.tiles_y) << 4;
        const int b0 = tmi / g.tiles_y;
        const int nt0 = rng * g.nr;

        GB_STAMP()
        // Phase 1 operands (wave = chunk): A operand of MFMA step s = We[k = 2 s + h][32 wq + (lane & 31)] (HWIO [3][3][2][128]:
        // k = 2 * tap + c) and the bias of the lane's 16 result rows.  Requested before the barrier, reloaded per item (L2 hits)
        // so that they do not occupy registers in the sweep.
        // phase 1's lane arithmetic (tile pixel -> patch address, halo pixel -> plane address, ...) is invariant across the
        // items of the persistent loop: hoisted out of it, those ~40 values stay live through the sweep and spill to scratch
        // (5 exposed scratch loads per pair of tiles: phase 1 ran 3x longer).  An opaque copy of the lane id keeps them local.
        int lane1 = lane;
        asm volatile("" : "+v"(lane1));
        const int h = lane1 >> 5;
        i32x4 ewt[4];                          // A operands of the four f16 MFMAs (GbrParams.we16: [chunk][instr][lane] x 16 B)
        f32x16 ebias;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            ewt[j] = *reinterpret_cast<const i32x4*>(reinterpret_cast<const char*>(p.we16) + (((wq * 4 + j) * 64 + lane1) << 4));
#pragma unroll
        for (int t = 0; t < 16; t += 4) {
            const float4 b4 = *reinterpret_cast<const float4*>(p.be + 32 * wq + 8 * (t >> 2) + 4 * h);
            ebias[t] = b4.x; ebias[t + 1] = b4.y; ebias[t + 2] = b4.z; ebias[t + 3] = b4.w;
        }
        // the 20 x 20 patch of the nearest-resized 2-channel mask (zero outside the image): requested before the barrier too
        float2 pv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int q = tid + 256 * u;
            const int py = q / 20, pxx = q - py * 20;
            const int y = ty0 - 2 + py, x = tx0 - 2 + pxx;
            pv[u] = make_float2(0.f, 0.f);
            if (q < 400 && y >= 0 && y < p.r && x >= 0 && x < p.r)
                pv[u] = *reinterpret_cast<const float2*>(p.src + (((size_t)b0 * p.S + (y * p.f + p.o)) * p.S + (x * p.f + p.o)) * 2);
        }
        __syncthreads();                       // the previous item's sweep has left the planes and the stage lines
        GB_STAMP2()
        // ================= phase 1a: the patch into LDS ========================================================================
        {   // patch[kind][pixel]: one dword = the two mask channels as fp16; kind 0 = hi = f16(v), 1 = lo = f16(v - hi)
            unsigned* const patch = reinterpret_cast<unsigned*>(smem + GB_STAGE_OFF);
            typedef _Float16 h2p __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int q = tid + 256 * u;
                const h2p hi2 = {(_Float16)pv[u].x, (_Float16)pv[u].y};
                const h2p lo2 = {(_Float16)(pv[u].x - (float)hi2[0]), (_Float16)(pv[u].y - (float)hi2[1])};
                if (q < 400) {
                    patch[q] = __builtin_bit_cast(unsigned, hi2);
                    patch[400 + q] = __builtin_bit_cast(unsigned, lo2);
                }
            }
        }
        __syncthreads();
        GB_STAMP2()
        // ================= phase 1b: embedding halo -> F / X / SC planes; wave = chunk ======================================
        // E[ch][pixel] = bias[ch] + sum_k We[k][ch] * mask[pixel][k], k = 2 * tap + c (18 terms), as THREE fp16 products per term
        // — w_hi x_hi + w_hi x_lo + w_lo x_hi with hi = f16(v), lo = f16(v - hi): 22 of fp32's 24 bits, products exact, fp32
        // accumulation, f16 denormals honoured (tools/gpu_diag_mfma_f16.hip) — on v_mfma_f32_32x32x16_f16: the 54 products are
        // packed into the 64 K slots of FOUR instructions per tile of 32 channels x 32 pixels (128 matrix cycles; until round
        // 3's last step the exact-fp32 v_mfma_f32_32x32x2_f32 took 9 x 64 and, sharing the vector ALU's issue, did not overlap the
        // conversions).  Mask values beyond fp16's range (|v| > 65504; the tiler hands over [-0.5, 0.5]) saturate.  A lane of the
        // 32 x 32 result holds HALF of a pixel's 32 channels (rows 8q + 4h + r, h = lane >> 5); two tiles (pixels A, B) and
        // one v_permlane32_swap per register leave ALL 32 channels of pixel 64 * pair + lane in the lane, which is what the
        // fp6 converter wants (v_cvt_scalef32_2xpk16_fp6_f32: 32 values -> one 24-byte k-block, element 2t = a[t],
        // 2t + 1 = b[t]; tools/gpu_diag_gbr.hip).  Position e of a chunk is therefore channel GBR_PERM(e) =
        // 8 * (e >> 3) + 4 * (e & 1) + ((e >> 1) & 3); the gamma|beta weights are uploaded in the same order.
        // Software pipeline: the 18 MFMAs of pair p + 1 are issued before the conversion of pair p (an MFMA holds the vector
        // issue for 8 of its 64 cycles: the ~200 VALU instructions of a conversion run beside them).
        {
            const unsigned* const patch = reinterpret_cast<const unsigned*>(smem + GB_STAGE_OFF);
            // K slots of the four MFMAs: pair P = 8 j + 4 h + u (instr j, k-half h = lane >> 5, dword u of the lane's 8 halves)
            // is term P / 9, tap P % 9 for P < 27 (zero weights beyond): its patch dword sits at kind * 400 + dy * 20 + dx
#define GB_EMB_POFF(P) ((P) < 27 ? (((P) / 9 == 1) ? 400 : 0) + (((P) % 9) / 3) * 20 + ((P) % 9) % 3 : 0)
            int poff[16];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int u = 0; u < 4; ++u) poff[4 * j + u] = h ? GB_EMB_POFF(8 * j + 4 + u) : GB_EMB_POFF(8 * j + u);
            // B operands (patch values) of tile `tile_` (pixels 32 tile + (lane & 31), clamped): 16 LDS reads
#define GB_EMB_LOADB(bv_, tile_)                                                                  \
    {                                                                                            \
        int hq_ = 32 * (tile_) + (lane1 & 31);                                                    \
        hq_ = hq_ < HP ? hq_ : HP - 1;                                                           \
        const int hy_ = (hq_ * 3641) >> 16;              /* / 18 for 0 <= hq < 324 */            \
        const int hx_ = hq_ - hy_ * HW;                                                          \
        const unsigned* const pp_ = patch + (hy_ * 20 + hx_);                                    \
        _Pragma("unroll") for (int q_ = 0; q_ < 16; ++q_) bv_[q_] = pp_[poff[q_]];               \
    }
#define GB_EMB_B(bv_, J) __builtin_bit_cast(f16x8, i32x4{(int)bv_[4 * (J)], (int)bv_[4 * (J) + 1], (int)bv_[4 * (J) + 2], (int)bv_[4 * (J) + 3]})
            // the K-th of the 18 MFMAs of the NEXT pair (tile A step K >> 1 for even K, tile B for odd K), issued between the
            // conversion steps of the current pair: an MFMA occupies the matrix pipe for 64 cycles and the vector issue for a
            // few, so ~12 VALU instructions per MFMA run beside it
#define GB_EMB_MFMA(K)                                                                            \
    {                                                                                            \
        if ((K) & 1) { if (nextB) nB = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ewt[(K) >> 1]), GB_EMB_B(bvB, (K) >> 1), (K) >> 1 ? nB : ebias, 0, 0, 0); } \
        else if (nextA) nA = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ewt[(K) >> 1]), GB_EMB_B(bvA, (K) >> 1), (K) >> 1 ? nA : ebias, 0, 0, 0); \
    }
            // One pair: conversion of (tA, tB) = pixels 64 PAIR + lane, interleaved with the MFMAs of pair PAIR + 1 into (nA, nB).
            // lane l < 32 gets tile A's two channel halves of its pixel, l >= 32 tile B's (v_permlane32_swap).
#define GB_EMB_PAIR(PAIR)                                                                         \
    {                                                                                            \
        constexpr bool nextA = (PAIR) + 1 <= 5, nextB = (PAIR) + 1 <= 4;                          \
        unsigned bvA[16], bvB[16];                                                               \
        if (nextA) GB_EMB_LOADB(bvA, 2 * ((PAIR) + 1))                                            \
        if (nextB) GB_EMB_LOADB(bvB, 2 * ((PAIR) + 1) + 1)                                        \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        if ((PAIR) == 1) GB_STAMP3()                                                             \
        const int hp = 64 * (PAIR) + lane1;                                                       \
        const bool live = hp < HP;                                                               \
        const int hq = live ? hp : HP - 1;                                                       \
        const int hy = (hq * 3641) >> 16, hx = hq - hy * HW;                                     \
        const int y = ty0 - 1 + hy, x = tx0 - 1 + hx;                                            \
        /* relu, the conv's zero padding outside the image (not relu(bias)) and fp16's range (as msr_store_f16c4_dev) in \
           one v_med3_f32: clamp to [0, top] with top = 0 for a pixel outside the image */     \
        const float top = (y >= 0 && y < p.r && x >= 0 && x < p.r) ? 65504.f : 0.f;              \
        float lo[16], hi[16];                                                                    \
        float amax = 0.f;                                                                        \
        /* the pair's 32 results leave the accumulator registers BEFORE the next pair's MFMAs are issued: a             \
           v_accvgpr_read beside an MFMA in flight waits for it (measured: MFMAs and conversion ran one after the other) */ \
        _Pragma("unroll") for (int t = 0; t < 16; ++t) {                                         \
            /* (scalar temporaries: __builtin_bit_cast applied to a vector ELEMENT reads element 0, hipcc 7.2) */ \
            const float fa_ = tA[t], fb_ = tB[t];                                                \
            const unsigned ya_ = (PAIR) == 5 ? 0u : __builtin_bit_cast(unsigned, fb_);           \
            const auto r_ = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, fa_), ya_, false, false); \
            const unsigned r0_ = r_[0], r1_ = r_[1];                                             \
            lo[t] = __builtin_bit_cast(float, r0_);                                              \
            hi[t] = __builtin_bit_cast(float, r1_);                                              \
        }                                                                                        \
        asm volatile("" : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(lo[4]), "+v"(lo[5]), "+v"(lo[6]), "+v"(lo[7]),   \
                          "+v"(lo[8]), "+v"(lo[9]), "+v"(lo[10]), "+v"(lo[11]), "+v"(lo[12]), "+v"(lo[13]), "+v"(lo[14]), "+v"(lo[15])); \
        asm volatile("" : "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]), "+v"(hi[4]), "+v"(hi[5]), "+v"(hi[6]), "+v"(hi[7]),   \
                          "+v"(hi[8]), "+v"(hi[9]), "+v"(hi[10]), "+v"(hi[11]), "+v"(hi[12]), "+v"(hi[13]), "+v"(hi[14]), "+v"(hi[15])); \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        if ((PAIR) == 1) GB_STAMP3()                                                             \
        _Pragma("unroll") for (int t = 0; t < 16; ++t) {                                         \
            lo[t] = __builtin_amdgcn_fmed3f(lo[t], 0.f, top);                                    \
            hi[t] = __builtin_amdgcn_fmed3f(hi[t], 0.f, top);                                    \
            amax = fmaxf(amax, fmaxf(lo[t], hi[t]));                                             \
            if (t < 8 && !(t & 1)) GB_EMB_MFMA(t >> 1)          /* the next pair's 8 MFMAs (32 cycles each) beside the conversion */ \
            __builtin_amdgcn_sched_barrier(0);                                                   \
        }                                                                                        \
        if ((PAIR) == 1) GB_STAMP3()                                                             \
        const int eb = msr_block_e8m0_dev(amax);                                                 \
        const float s_hi = __builtin_bit_cast(float, eb << 23);                                  \
        const float s_lo = __builtin_bit_cast(float, (eb - 11) << 23);                           \
        f32x16 va, vb, la, lb;                                                                   \
        f16x8 hv[4];                                                                             \
        _Pragma("unroll") for (int t = 0; t < 16; ++t) {                                         \
            typedef _Float16 h2_ __attribute__((ext_vector_type(2)));                            \
            const h2_ pk = {(_Float16)lo[t], (_Float16)hi[t]};                                   \
            hv[t >> 2][2 * (t & 3)] = pk[0];             /* position 2t = lo[t], 2t + 1 = hi[t] */ \
            hv[t >> 2][2 * (t & 3) + 1] = pk[1];                                                 \
            va[t] = lo[t];                                                                       \
            vb[t] = hi[t];                                                                       \
            la[t] = lo[t] - (float)pk[0];                                                        \
            lb[t] = hi[t] - (float)pk[1];                                                        \
            if (t < 8 && !(t & 1)) GB_EMB_MFMA(4 + (t >> 1))                                     \
            __builtin_amdgcn_sched_barrier(0);                                                   \
        }                                                                                        \
        /* v_cvt_scalef32_2xpk16_fp6_f32 through inline asm with an EARLY-CLOBBER result: the builtin lets hipcc 7.2 put \
           the 6 result registers on top of the first source (v[34:39] <- v[34:49], ...), and the hardware then reads    \
           clobbered inputs (measured: the h6 piece came out partly wrong, tools/gpu_debug_gbr.py) */ \
        if ((PAIR) == 1) GB_STAMP3()                                                             \
        i32x6 h6, l6;                                                                            \
        if constexpr (!NOX) {                                                                    \
        asm volatile("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, %3" : "=&v"(h6) : "v"(va), "v"(vb), "v"(s_hi)); \
        asm volatile("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, %3" : "=&v"(l6) : "v"(la), "v"(lb), "v"(s_lo)); \
        }                                                                                        \
        if (live) {                                                                              \
            char* const fp = smem + (wq * 4 * PH + hp) * 16;                                     \
            _Pragma("unroll") for (int pc = 0; pc < 4; ++pc) *reinterpret_cast<f16x8*>(fp + pc * PH * 16) = hv[pc]; \
        }                                                                                        \
        if (!NOX && live) {       /* NOX: the fp6 pieces and their scales are not read: not made */ \
            char* const xpa = smem + GB_F_BYTES + (wq * 2 * PH + hp) * 16;                       \
            char* const xpb = smem + GB_F_BYTES + GB_XA_BYTES + (wq * 2 * PH + hp) * 8;          \
            *reinterpret_cast<i32x4*>(xpa) = i32x4{h6[0], h6[1], h6[2], h6[3]};                  \
            *reinterpret_cast<i32x4*>(xpa + PH * 16) = i32x4{l6[0], l6[1], l6[2], l6[3]};        \
            *reinterpret_cast<i32x2*>(xpb) = i32x2{h6[4], h6[5]};                                \
            *reinterpret_cast<i32x2*>(xpb + PH * 8) = i32x2{l6[4], l6[5]};                       \
            /* the pixel's scale: byte hy - dy of [dy][chunk][kind][hx] for the row shifts dy with 0 <= hy - dy < 16 */ \
            unsigned char* const sp = reinterpret_cast<unsigned char*>(smem) + GB_F_BYTES + GB_X_BYTES + (wq * 2 * 18 + hx) * 16 + hy; \
            _Pragma("unroll") for (int dy = 0; dy < 3; ++dy)                                     \
                if (hy >= dy && hy - dy < 16) {                                                  \
                    sp[dy * (4 * 2 * 18 * 16) - dy] = (unsigned char)eb;                         \
                    sp[dy * (4 * 2 * 18 * 16) - dy + 18 * 16] = (unsigned char)(eb - 11);        \
                }                                                                                \
        }                                                                                        \
        if ((PAIR) == 1) GB_STAMP3()                                                             \
        tA = nA;                                                                                 \
        tB = nB;                                                                                 \
    }
            f32x16 tA, tB, nA, nB;
            {   // pair 0's own MFMAs (nothing to hide them behind)
                unsigned bvA[16], bvB[16];
                GB_EMB_LOADB(bvA, 0)
                GB_EMB_LOADB(bvB, 1)
                tA = ebias;
                tB = ebias;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    tA = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ewt[j]), GB_EMB_B(bvA, j), tA, 0, 0, 0);
                    tB = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ewt[j]), GB_EMB_B(bvB, j), tB, 0, 0, 0);
                }
                nA = tA;
                nB = tB;
            }
            GB_EMB_PAIR(0) GB_EMB_PAIR(1) GB_EMB_PAIR(2) GB_EMB_PAIR(3) GB_EMB_PAIR(4) GB_EMB_PAIR(5)
            GB_STAMP2()
#undef GB_EMB_LOADB
#undef GB_EMB_POFF
#undef GB_EMB_B
#undef GB_EMB_MFMA
#undef GB_EMB_PAIR
        }
        __syncthreads();
        GB_STAMP()

        // ================= phase 2: sweep the item's channel blocks over the resident halo ====================================
        // K-step T of a body (CB = its first chunk, 0 or 2): chunk (CB + T / 9) & 3, tap T % 9; T = 18, 19 are the first two
        // steps of whatever comes next (the other body of this channel block, or the first body of the next block)
#define GB_CH(CB, T) (((CB) + (T) / 9) & 3)
#define GB_TAP(T) ((((T) % 9) / 3) * HW + ((T) % 9) % 3)
#define GB_RD_F(dst, CB, T, i)                                                                    \
    dst = *reinterpret_cast<const i32x4*>((GB_CH(CB, T) >> 1 ? FL1 : FL0) +                      \
                                          ((GB_CH(CB, T) & 1) * 4 * PH + (i) * HW + GB_TAP(T)) * 16)
        // per-lane pointers of a pair's fp6 pieces and scales: lane groups 0, 1 read the even tap, 2, 3 the odd one
#define GB_DSEL(TE) ((TE) % 9 == 8 ? 2 : (((TE) % 9) % 3 == 2 ? 1 : 0))
#define GB_XPA(CB, TE) (XAd[GB_DSEL(TE)] + (GB_CH(CB, TE) * 2 * PH + GB_TAP(TE)) * 16)
#define GB_XPB(CB, TE) (XBd[GB_DSEL(TE)] + (GB_CH(CB, TE) * 2 * PH + GB_TAP(TE)) * 8)
    // the 16 scales of the lane's tap: [dy][chunk][kind][px + dx]
#define GB_SP(CB, TE) (SLd[GB_DSEL(TE)] + (((((TE) % 9) / 3) * 4 + GB_CH(CB, TE)) * 2 * 18 + ((TE) % 9) % 3) * 16)
#define GB_RD_X(k, XPA, XPB, i)                                                                   \
    {                                                                                            \
        xa[k] = *reinterpret_cast<const i32x4*>((XPA) + (i) * HW * 16);                          \
        xb[k] = *reinterpret_cast<const i32x2*>((XPB) + (i) * HW * 8);                           \
    }
        // weights of tap pair P of the channel block (P = 18: pair 0 of the next block, or of this one again on the last)
#define GB_WSOFF(P) ((P) < 18 ? w_cur + (unsigned)(P) * 8192u : w_nxt)
#define GB_LOAD_B(dstE, dstO, dxE, dxO, P, j)                                                     \
    {                                                                                            \
        dstE[j] = GB_BUFLD(wv[j], GB_WSOFF(P));                                                  \
        dstO[j] = GB_BUFLD(wv[j] + 1024, GB_WSOFF(P));                                           \
        if constexpr (!NOX) {                                                                    \
            dxE[j] = GB_BUFLD(wv[j] + 2048, GB_WSOFF(P));                                        \
            dxO[j] = GB_BUFLD(wv[j] + 3072, GB_WSOFF(P));                                        \
        }                                                                                        \
    }
#define GB_F16(v) __builtin_bit_cast(f16x8, v)
#define GB_CAT8(lo, hi) __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7)
        // Phase E of pair U: tap 2U, fragments fa; requests fa ahead, then fb of tap 2U + 1; the next pair's weights
#define GB_PHASE_E(CB, U)                                                                         \
    {                                                                                            \
        constexpr int T = 2 * (U);                                                               \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                         \
            if (i + PD < 16) GB_RD_F(fa[(i + PD) & 15], CB, T, (i + PD) & 15);                   \
            else GB_RD_F(fb[(i + PD) & 15], CB, T + 1, (i + PD) & 15);                           \
            if (i == 0) GB_LOAD_B(nbE, nbO, nxE, nxO, ((CB) / 2) * 9 + (U) + 1, 0)               \
            if (i == 8) GB_LOAD_B(nbE, nbO, nxE, nxO, ((CB) / 2) * 9 + (U) + 1, 1)               \
            acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(GB_F16(bE[0]), GB_F16(fa[i]), acc[i][0], 0, 0, 0); \
            acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(GB_F16(bE[1]), GB_F16(fa[i]), acc[i][1], 0, 0, 0); \
            __builtin_amdgcn_sched_barrier(0);                                                   \
        }                                                                                        \
    }
    // Phase O: tap 2U + 1, fragments fb; requests fb ahead, then the first fp6 pieces of the pair
#define GB_PHASE_O(CB, U)                                                                         \
    {                                                                                            \
        constexpr int T = 2 * (U) + 1;                                                           \
        const char* const xpa_ = GB_XPA(CB, T - 1);                                              \
        const char* const xpb_ = GB_XPB(CB, T - 1);                                              \
        if constexpr (!NOX) xsc = *reinterpret_cast<const i32x4*>(GB_SP(CB, T - 1));             \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                         \
            if (i + PD < 16) GB_RD_F(fb[(i + PD) & 15], CB, T, (i + PD) & 15);                   \
            if constexpr (NOX) {    /* no phase C: its requests of the next pair's fa move here */ \
                if (i >= 16 - PD) GB_RD_F(fa[(i + PD) & 15], CB, T + 1, (i + PD) & 15);          \
            } else if (i >= 16 - PDC) GB_RD_X((i + PDC) & 15, xpa_, xpb_, (i + PDC) & 15)        \
            acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(GB_F16(bO[0]), GB_F16(fb[i]), acc[i][0], 0, 0, 0); \
            acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(GB_F16(bO[1]), GB_F16(fb[i]), acc[i][1], 0, 0, 0); \
            __builtin_amdgcn_sched_barrier(0);                                                   \
        }                                                                                        \
    }
    // Phase C: the cross terms of taps 2U and 2U + 1 (block-scaled K = 128 fp6 MFMA, operands in registers 0..5, the block
    // scales in byte 0 of the scale operands); requests the fp6 pieces ahead, then fa of the next pair's first tap
#define GB_C_STEP(CB, T, i)                                                                       \
    {                                                                                            \
        if ((i) + PDC < 16) GB_RD_X(((i) + PDC) & 15, xpa_, xpb_, ((i) + PDC) & 15)              \
        if ((i) >= 16 - PD) GB_RD_F(fa[((i) + PD) & 15], CB, (T) + 2, ((i) + PD) & 15);          \
        const i32x8 aq_ = {xa[i][0], xa[i][1], xa[i][2], xa[i][3], xb[i][0], xb[i][1], 0, 0};    \
        /* the scale byte select must be a literal: fragment i uses byte i & 3 of scale register i >> 2 */ \
        acc[i][0] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wq0_, aq_, acc[i][0], 2, 2, 0, wq0_[6], (i) & 3, xsc[(i) >> 2]); \
        acc[i][1] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wq1_, aq_, acc[i][1], 2, 2, 0, wq1_[6], (i) & 3, xsc[(i) >> 2]); \
        __builtin_amdgcn_sched_barrier(0);                                                       \
    }
#define GB_PHASE_C(CB, U)                                                                         \
    {                                                                                            \
        constexpr int T = 2 * (U);                                                               \
        const i32x8 wq0_ = GB_CAT8(xE[0], xO[0]), wq1_ = GB_CAT8(xE[1], xO[1]);                  \
        const char* const xpa_ = GB_XPA(CB, T);                                                  \
        const char* const xpb_ = GB_XPB(CB, T);                                                  \
        if constexpr (!NOX) {                                                                    \
        GB_C_STEP(CB, T, 0) GB_C_STEP(CB, T, 1) GB_C_STEP(CB, T, 2) GB_C_STEP(CB, T, 3)          \
        GB_C_STEP(CB, T, 4) GB_C_STEP(CB, T, 5) GB_C_STEP(CB, T, 6) GB_C_STEP(CB, T, 7)          \
        GB_C_STEP(CB, T, 8) GB_C_STEP(CB, T, 9) GB_C_STEP(CB, T, 10) GB_C_STEP(CB, T, 11)        \
        GB_C_STEP(CB, T, 12) GB_C_STEP(CB, T, 13) GB_C_STEP(CB, T, 14) GB_C_STEP(CB, T, 15)      \
        }                                                                                        \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) { bE[j] = nbE[j]; bO[j] = nbO[j]; xE[j] = nxE[j]; xO[j] = nxO[j]; } \
    }
#define GB_PAIR(CB, U) GB_PHASE_E(CB, U) GB_PHASE_O(CB, U) GB_PHASE_C(CB, U)
#define GB_BODY(CB) GB_PAIR(CB, 0) GB_PAIR(CB, 1) GB_PAIR(CB, 2) GB_PAIR(CB, 3) GB_PAIR(CB, 4) GB_PAIR(CB, 5) GB_PAIR(CB, 6) GB_PAIR(CB, 7) GB_PAIR(CB, 8)

        // prologue of the sweep: the weights of taps 0 and 1 of the first channel block, the first PD pixel fragments
        {
            const unsigned w_cur = (unsigned)((nt0 * 4 + wq) * 18) * 8192u, w_nxt = w_cur;
            GB_LOAD_B(bE, bO, xE, xO, 0, 0)
            GB_LOAD_B(bE, bO, xE, xO, 0, 1)
        }
#pragma unroll
        for (int i = 0; i < PD; ++i) GB_RD_F(fa[i], 0, 0, i);

        for (int nt = nt0; nt < nt0 + g.nr; ++nt) {
            const int n0 = nt * 128;
            const unsigned w_cur = (unsigned)((nt * 4 + wq) * 18) * 8192u;
            const unsigned w_nxt = nt + 1 < nt0 + g.nr ? w_cur + 4u * 18u * 8192u : w_cur;    // last block: re-reads its own (unused)
            const int x = tx0 + px;
            const int ch = ((n0 + (wq >> 1) * 64) >> 1) + (wq & 1) * 16 + 4 * cg;     // first of the lane's 4 output channels
            const int cb0 = n0 + (wq >> 1) * 64 + (wq & 1) * 16 + 4 * cg;             // its gamma column (beta: + 32)
            {   // the accumulators start at the conv biases (the lane's 4 gamma and 4 beta columns): nothing to add later
                const float4 g4 = *reinterpret_cast<const float4*>(p.bias + cb0);
                const float4 b4 = *reinterpret_cast<const float4*>(p.bias + cb0 + 32);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    acc[i][0] = f32x4{g4.x, g4.y, g4.z, g4.w};
                    acc[i][1] = f32x4{b4.x, b4.y, b4.z, b4.w};
                }
            }
            GB_BODY(0)
            // the epilogue's memory operands are requested ~2 tap pairs ahead (below)
            float4 xin[16];
            float4 mq4, sq4;
            {
                GB_PAIR(2, 0) GB_PAIR(2, 1) GB_PAIR(2, 2) GB_PAIR(2, 3) GB_PAIR(2, 4) GB_PAIR(2, 5) GB_PAIR(2, 6)
                {   // ~2 tap pairs (4000 cycles) before their use
                    const float* const abase = p.aux + (size_t)b0 * p.aux_pb + (x >> p.aux_shift) * p.aux_px + ch;
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        xin[i] = *reinterpret_cast<const float4*>(abase + ((ty0 + i) >> p.aux_shift) * p.aux_py);
                    mq4 = *reinterpret_cast<const float4*>(p.mean + ch);
                    sq4 = *reinterpret_cast<const float4*>(p.stdv + ch);
                }
                GB_PAIR(2, 7) GB_PAIR(2, 8)
            }
            // ---- SPADE epilogue: acc[i][0] = gamma, acc[i][1] = beta of channels ch .. ch + 3 at pixel (ty0 + i, x) ----
            GB_STAMP2()
            {
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                // out = lrelu((acc_g + bias_g) * (x - mean) / std + (acc_b + bias_b)): per channel nk = x * A + Bc
                const float A[4] = {1.f / sq4.x, 1.f / sq4.y, 1.f / sq4.z, 1.f / sq4.w};
                const float Bc[4] = {-mq4.x * A[0], -mq4.y * A[1], -mq4.z * A[2], -mq4.w * A[3]};
                const int half = wq & 1;
                // the lane's 16 bytes of the pixel's chunk line: fp16 of 8 channels (lane groups 0, 1), h8 / l8 of the 16 (2 / 3)
                const int doff = cg < 2 ? half * 8 + 4 * cg : (cg == 2 ? 16 + half * 4 : 24 + half * 4);
                unsigned* const obase = reinterpret_cast<unsigned*>(p.out + (size_t)p.out_off + (size_t)b0 * p.out_pb +
                                                                    x * p.out_px + (ch & ~31)) + doff;
                // Software pipeline over the 16 tile rows: iteration i requests the assembled line of row i - 1 from LDS, does
                // the arithmetic of row i beside that read, writes row i's pieces to the line buffer (LDS operations of a wave execute in order: the
                // read of row i - 1 is ahead of them) and stores row i - 1
                // (LDS operations of one wave execute in order; the round trip per row used to be exposed: 7.6k cycles per block)
                uint4 q;
#pragma unroll
                for (int i = 0; i <= 16; ++i) {
                    if (i > 0) {
                        asm volatile("" ::: "memory");          // the pieces were written through other types
                        q = *reinterpret_cast<const uint4*>(stage + px * GB_LINE + 16 * cg);
                        asm volatile("" ::: "memory");
                    }
                    if (i < 16) {
                        char* const line = stage + px * GB_LINE;
                        const float xq[4] = {xin[i].x, xin[i].y, xin[i].z, xin[i].w};
                        float v[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float nk = __builtin_fmaf(xq[k], A[k], Bc[k]);
                            const float t = __builtin_fmaf(acc[i][0][k], nk, acc[i][1][k]);
                            const float u = fmaxf(t, t * p.slope);                // leaky relu, 0 <= slope <= 1 (checked on the host)
                            v[k] = __builtin_amdgcn_fmed3f(u, -65504.f, 65504.f);
                        }
                        const h2 a = {(_Float16)v[0], (_Float16)v[1]}, b = {(_Float16)v[2], (_Float16)v[3]};
                        unsigned h8 = 0;
                        h8 = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], h8, false);
                        h8 = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], h8, true);
                        // l8 = e4m3((v - hi) * 2^11): the scaled converter divides by its scale operand (2^-11)
                        typedef short s2_ __attribute__((ext_vector_type(2)));
                        s2_ l8v = {0, 0};
                        l8v = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(l8v, v[0] - (float)a[0], v[1] - (float)a[1], 0x1p-11f, false);
                        l8v = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(l8v, v[2] - (float)b[0], v[3] - (float)b[1], 0x1p-11f, true);
                        const unsigned l8 = __builtin_bit_cast(unsigned, l8v);
                        *reinterpret_cast<uint2*>(line + 8 * cg) = make_uint2(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b));
                        *reinterpret_cast<unsigned*>(line + 32 + 4 * cg) = h8;
                        *reinterpret_cast<unsigned*>(line + 48 + 4 * cg) = l8;
                    }
                    if (i > 0) *reinterpret_cast<uint4*>(obase + (size_t)(ty0 + i - 1) * p.out_py) = q;
                }
            }
            GB_STAMP2()
        }
        GB_STAMP()
    }
#ifdef MSR_GB_STAMPS
    if (blockIdx.x == 8 && lane == 0)
        for (int k = 0; k + 1 < dbg_n; ++k) printf("gbr wave %d stamp %2d: %7u cycles\n", wq, k, dbg[k + 1] - dbg[k]);
#endif
#undef GB_STAMP
#undef GB_STAMP2
#undef GB_STAMP3
#undef GB_BUFLD
#undef GB_CH
#undef GB_TAP
#undef GB_RD_F
#undef GB_XPA
#undef GB_XPB
#undef GB_DSEL
#undef GB_SP
#undef GB_RD_X
#undef GB_WSOFF
#undef GB_LOAD_B
#undef GB_F16
#undef GB_CAT8
#undef GB_PHASE_E
#undef GB_PHASE_O
#undef GB_PHASE_C
#undef GB_C_STEP
#undef GB_PAIR
#undef GB_BODY
}

void conv_gbr_embed_image(const float* we, float* out4096) {
    unsigned short* o = reinterpret_cast<unsigned short*>(out4096);
    for (int wq = 0; wq < 4; ++wq)
        for (int j = 0; j < 4; ++j)
            for (int lane = 0; lane < 64; ++lane) {
                const int kh = lane >> 5, ch = 32 * wq + (lane & 31);
                for (int u = 0; u < 4; ++u) {
                    const int P = 8 * j + 4 * kh + u;
                    for (int c = 0; c < 2; ++c) {
                        unsigned short bits = 0;
                        if (P < 27) {
                            const float w = we[((P % 9) * 2 + c) * 128 + ch];
                            const _Float16 hi = (_Float16)w;
                            const _Float16 lo = (_Float16)(w - (float)hi);
                            bits = __builtin_bit_cast(unsigned short, P / 9 == 2 ? lo : hi);
                        }
                        o[(((wq * 4 + j) * 64 + lane) * 8) + 2 * u + c] = bits;
                    }
                }
            }
}

hipError_t conv_gbr_init() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gb_resident<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)GB_LDS);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gb_resident<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)GB_LDS);
}

// Work decomposition: a work item = one 16 x 16 pixel tile x `nr` channel blocks; the N / 128 blocks of a pixel tile are cut
// into the fewest ranges (a power of two) that give every CU an item.  Returns 0 if the layer is not one for this kernel.
int conv_gbr_ranges(int B, int r, int N) {
    static const bool off = std::getenv("MSR_GBR") && std::atoi(std::getenv("MSR_GBR")) == 0;
    if (off || r < 16 || (r & (r - 1)) || N % 128) return 0;
    const int tiles_p = B * (r / 16) * (r / 16), tiles_n = N / 128;
    int ranges = 1;
    while (tiles_p * ranges < 256 && ranges * 2 <= tiles_n && tiles_n % (ranges * 2) == 0) ranges *= 2;
    if (tiles_p * ranges < 128) return 0;               // too few items even at one block per item
    // one block per item on half of the CUs: still ahead of the ping-pong K ranges + split-K epilogue + embedding launch at
    // r = 16 (26 against 38 us at B = 8), not at higher resolution where those launches fill the chip
    if (tiles_n / ranges < 2 && tiles_p * ranges < 256 && r > 16) return 0;
    return ranges;
}

hipError_t launch_conv_gbr(const GbrParams& p, int ranges, hipStream_t s) {
    if (ranges < 1 || p.r < 16 || (p.r & (p.r - 1)) || p.N % 128 || (p.N / 128) % ranges || p.out_split != 4 || !p.src || !p.we16 || !p.be || !p.wt || !p.aux || !p.mean || !p.stdv || !p.out)
        return hipErrorInvalidValue;
    if (p.f < 1 || p.S != p.r * p.f || p.out_px % 32 || !(p.slope >= 0.f && p.slope <= 1.f)) return hipErrorInvalidValue;
    GbrGeom g;
    g.tiles_x = p.r / 16;
    g.tiles_y = p.r / 16;
    g.tiles_p = g.tiles_x * g.tiles_y * p.B;
    g.nr = p.N / 128 / ranges;
    g.items = g.tiles_p * ranges;
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorInvalidValue;
        n_cu = prop.multiProcessorCount & ~7;
        if (n_cu < 8) n_cu = 8;
    }
    const int grid = g.items < n_cu ? ((g.items + 7) & ~7) : n_cu;
    if (p.no_cross) conv_gb_resident<true><<<grid, 256, GB_LDS, s>>>(p, g);
    else conv_gb_resident<false><<<grid, 256, GB_LDS, s>>>(p, g);
    return hipGetLastError();
}

}  // namespace msr
