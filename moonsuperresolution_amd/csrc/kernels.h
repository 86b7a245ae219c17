// Internal launch interface between the C-ABI layer (api.hip) and the gfx950 kernels.
// Everything here is MI355X-only HIP; there is no other backend.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

// Diagnostic switches (in-kernel s_memtime stamps; what-if builds that compute WRONG results on purpose to price one
// cost: SW_NOBAR, SW_NOEPI, MSR_WI_ONESTORE, MSR_WI_NOSTORE) compile only into a library that declares itself a
// diagnostic build: -DMSR_DIAG_BUILD.  msr_create of such a library fails unless MSR_ALLOW_DIAG_BUILD=1 is set, so a
// what-if object can never be picked up as the product by accident (tools/gpu_*_stamps.py, tools/gpu_ab.sh set it).
#if defined(SW_NOBAR) || defined(SW_NOEPI) || defined(MSR_WI_ONESTORE) || defined(MSR_WI_NOSTORE) || \
    defined(MSR_SW_STAMPS) || defined(MSR_PP_STAMPS) || defined(MSR_GB_STAMPS)
#ifndef MSR_DIAG_BUILD
#error "stamp / what-if switches need -DMSR_DIAG_BUILD (msr_create then refuses the library unless MSR_ALLOW_DIAG_BUILD=1)"
#endif
#endif

namespace msr {

// ---------------------------------------------------------------------------------------------
// conv_igemm_f32: NHWC implicit-GEMM convolution on v_mfma_f32_32x32x2_f32 (exact fp32).
//   M = B*Hout*Wout output pixels, N = output channels, K = KH*KW*Cin.
// The input is a physically zero-padded NHWC tensor, so the K loop has no bounds checks:
//   in[(b*in_pb) + (y*stride + kh)*in_py + (x*stride + kw)*Cin + c]  is tap (kh,kw) of output (y,x).
// Weights are pre-laid-out [KH*KW][N][Cin] (K contiguous per output channel).
// ---------------------------------------------------------------------------------------------
enum ConvEpilogue : int {
    EPI_BIAS = 0,   // out = acc + bias                       (encoder convs pass bias = zeros)
    EPI_RES = 1,    // out = acc + bias + aux[(y>>s),(x>>s)]  (ResidualBlock skip add, blocks.py:38,
                    //                                          with UpSampling2D folded into the index)
    EPI_SPADE = 2,  // N = 2C, columns interleaved (gamma block of 32 | beta block of 32):
                    // out = leaky_relu((g+bg) * ((x-mean)/std) + (b+bb))   spade.py:21-24 + blocks.py:30-34
    EPI_PARTIAL = 3,  // split-K: raw accumulators to partial[ks][B,Hout,Wout,N]; splitk_epilogue finishes
    EPI_AFFINE = 4,   // out = act(acc * scale[c] + shift[c]): folded BatchNormalization + LeakyReLU / ReLU of the
                      // pix2pix blocks (pix2pix.py:65-86); scale == nullptr means 1
};

struct ConvParams {
    const float* in;
    const float* wt;
    const float* bias;    // [N] in GEMM column order
    float* out;
    const float* aux;     // EPI_RES: residual; EPI_SPADE: x (the tensor being normalised)
    const float* mean;    // EPI_SPADE: [C]
    const float* stdv;    // EPI_SPADE: [C]  sqrt(var + eps)
    int B, Hout, Wout, Cin, N;
    int KH, KW, stride;
    int in_px, in_py, in_pb;      // input pitches in floats (in_px = Cin unless the input is a channel slice of a
                                  // wider tensor, e.g. the skip half of a pix2pix concat buffer; generic kernel only)
    int out_px, out_py, out_pb;   // output pitches in floats
    int out_off;                  // offset of output pixel (0,0) channel 0 of sample 0
    int aux_px, aux_py, aux_pb;   // aux pitches
    int aux_shift;                // 1 = aux is at half resolution (nearest 2x up-sample folded in)
    const float* scale;           // EPI_AFFINE: per-channel scale (nullable); the shift is `bias`
    int act;                      // EPI_AFFINE: 0 none, 1 relu, 2 leaky(slope)
    float slope;                  // leaky-relu slope of EPI_SPADE / EPI_AFFINE
    int ksplit;                   // > 1: the K loop is cut into ksplit ranges, one workgroup each (low-res layers)
    float* partial;               // [ksplit][B*Hout*Wout][N] workspace for split-K
    int prec;                     // PREC_F32: operands are fp32; PREC_BF16X3: operands are split-bf16 words
    int out_split;                // EPI_SPADE only: 1 = write the split-bf16 image (the consumer conv runs PREC_BF16X3 / its
                                  // fp16 twin decided by the producer of the mask embedding), 3 = write bf8 e5m2 bytes
                                  // (PREC_FP8 consumer: one byte per channel, dword index = channel / 4), 4 = the f16c
                                  // chunk image (PREC_F16C consumer)
    float* stat_partial;          // EPI_BIAS / EPI_RES, ksplit == 1: per-wave partial moments of the OUTPUT,
                                  // [P][3][N] = (count, mean, M2) per 32- or 64-row slab (P = conv_stat_slabs)
    const int* wexp;              // PREC_FP8: [N] e8m0 exponent of every output channel's weight scale, replicated in the
                                  // four bytes of the word (the MFMA's scale operand)
    int wt_frag;                  // PREC_BF16X3: weights are in MFMA-fragment order (conv_igemm_bf16x3, B in VGPRs)
                                  // instead of the split-bf16 image of [tap][N][Cin] (LDS-staged B)
    // EPI_BIAS / EPI_RES, ksplit > 1, mom_mean != nullptr: the split-K epilogue also produces the moments of the output
    // (splitk_epilogue_mom_kernel): mom_G = 1 over the whole batch (a SPADE input), mom_G > 1 per sample (instance norm)
    double* mom_partial = nullptr;
    float* mom_mean = nullptr;
    float* mom_std = nullptr;
    float mom_eps = 0.f;
    int mom_G = 0;
    int no_cross = 0;             // PREC_F16C on conv_igemm_f16c_sw only: 1 = leave the cross terms out ("f16" mode)
};
hipError_t launch_splitk_epilogue_mom(const ConvParams& p, int epi, hipStream_t s);
int moments_chunks(int G, int P);

// Split-bf16 operand format ("bf16x3"): tensors keep their fp32 size and addressing, but every aligned group of
// 32 channels (128 bytes) holds [32 x hi bf16 | 32 x lo bf16] with hi = bf16_rn(v), lo = bf16_rn(v - hi).
// That is exactly the LDS row image the kernel wants (8 consecutive k of one half per ds_read_b128), so staging
// is a plain 16-byte copy.  The conv computes a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on v_mfma_f32_32x32x16_bf16 with
// fp32 accumulation: per-product error <= ~3*2^-18 instead of fp32's 2^-24, at 16/3 of the fp32-MFMA rate.
// PREC_F16X2 ("2-term" products, opt-in for the SPADE gamma|beta convs only): the activation is split in two fp16
// halves (hi = f16_rn(v), lo = f16_rn(v - hi), ~22 bits) and the weight is ONE fp16 (11 bits, the lo half of its
// chunk image is stored but not read): a * w ~= a_hi*w_hi + a_lo*w_hi on v_mfma_f32_16x16x32_f16, two MFMAs instead
// of three, per-product error <= 2^-12 (the weight's rounding).  Same tensor layout as split-bf16, fp16 encodings.
// PREC_FP8 (declared NON-parity mode, BASELINE configs[4]): weights quantised to fp8 e4m3 with a power-of-two scale per
// output channel, activations to bf8 e5m2 (no scale: its range covers the activations), one product per element on the
// block-scaled v_mfma_scale_f32_16x16x128_f8f6f4 (128 channels of one tap per instruction, twice the bf16 rate); the
// per-channel weight scale rides in the instruction's e8m0 scale operand, fp32 accumulation.  Tensors hold one byte per
// channel; the kernel sees them as float slots of 4 channels (Cin / 4 "channels", same indexing as the other modes).
// PREC_F16C ("fp16 main term + fp8 cross terms"): a*b = a_hi*b_hi + (a_hi*b_lo + a_lo*b_hi) with hi = f16_rn(v), lo = v - hi.
// The main term runs on v_mfma_f32_16x16x32_f16 (exact products), the two cross terms — 2^-11 of the product, so 4 bits of
// them are enough — on the block-scaled fp8 MFMA (K = 128: both cross terms of two taps per instruction, twice the bf16
// rate): two MFMA-equivalents per product instead of bf16x3's three, per-product error ~2^-15.  A 32-channel chunk (128
// bytes, the split-bf16 geometry) holds [32 x hi f16 | 32 x h8 | 32 x l8] with h8 = e4m3(v), l8 = e4m3(lo * 2^11) for
// activations; weights store [32 x hi f16 | 32 x l8 | 32 x h8] (so that byte t of one pairs with byte t of the other:
// w_lo * x_hi, w_hi * x_lo) with a power-of-two scale per output channel and kind (ConvParams.wexp: byte 0 = e8m0 of the
// lo bytes, byte 1 = of the hi bytes).  Operand map of v_mfma_scale_f32_16x16x128_f8f6f4 as measured: lane (row, g)'s 32
// bytes are k = 16g..16g+15 and 64+16g..64+16g+15; the e8m0 scale of k-block b (32 consecutive k) comes from lane group b.
// PREC_F16C6: PREC_F16C with the cross terms in fp6 e2m3 — the block-scaled MFMA runs fp6 operands at twice its fp8 rate, so a
// product costs 1.5 MFMA-equivalents instead of 2, at the same end-to-end accuracy (e2m3 has e4m3's three mantissa bits;
// tools/emulate_cross_formats.py: 5.6e-5 against 5.9e-5).  e2m3 spans only six binades, so the pieces carry a block scale:
// one power of two per pixel and 32-channel chunk for activations (2^E >= max|x| / 7.5; the lo piece uses 2^(E-11)), one
// per output channel and piece for weights.  Chunk image (128 bytes per 32 channels):
//   bytes   0.. 63  32 x hi f16
//   bytes  64.. 87  32 x 6-bit codes, little-endian bit string (channel c at bits 6c..6c+5): activations h6 = e2m3(x / 2^E),
//                   weights l6 = e2m3((w - hi) / 2^El)
//   byte   88       e8m0 of that piece's scale (127 + E, resp. 127 + El); bytes 89..95 zero
//   bytes  96..119  the other piece: activations l6 = e2m3((x - hi) / 2^(E-11)), weights h6 = e2m3(w / 2^Eh)
//   byte  120       its e8m0 (127 + E - 11, resp. 127 + Eh); bytes 121..127 zero
// so that a lane reads ONE 32-byte half (two 16-byte loads) and holds the MFMA's 6-register operand in the first six
// registers and its scale in byte 0 of the seventh.  Operand map of the instruction for fp6 (tools/gpu_diag_fp6.hip,
// profiles/r02_mfma_scale_operand_map.txt): lane group g's 32 elements are one k-block whose scale comes from lane group g;
// here g = 0 / 1 hold the first / second half of the EVEN tap's chunk row, g = 2 / 3 of the odd tap's.
enum ConvPrecision : int { PREC_F32 = 0, PREC_BF16X3 = 1, PREC_F16X2 = 2, PREC_FP8 = 3, PREC_F16C = 4, PREC_F16C6 = 5 };

__host__ __device__ inline unsigned msr_bf16_rn(float v) {   // round-to-nearest-even, finite inputs
    union { float f; unsigned u; } c;
    c.f = v;
    return (c.u + 0x7FFFu + ((c.u >> 16) & 1u)) >> 16;
}
__host__ __device__ inline void msr_split_bf16(float v, unsigned& hi, unsigned& lo) {
    union { float f; unsigned u; } c;
    hi = msr_bf16_rn(v);
    c.u = hi << 16;
    lo = msr_bf16_rn(v - c.f);
}
// 4 consecutive channels c..c+3 (c % 4 == 0) of one pixel -> two 8-byte stores into the pixel's chunk image.
// `pixel` points at channel 0 of the pixel (float units).
__host__ __device__ inline void msr_store_split4(float* pixel, int c, float v0, float v1, float v2, float v3) {
    unsigned h0, l0, h1, l1, h2, l2, h3, l3;
    msr_split_bf16(v0, h0, l0); msr_split_bf16(v1, h1, l1); msr_split_bf16(v2, h2, l2); msr_split_bf16(v3, h3, l3);
    unsigned* chunk = reinterpret_cast<unsigned*>(pixel) + (c & ~31);
    const int w = (c & 31) >> 1;
    chunk[w] = h0 | (h1 << 16); chunk[w + 1] = h2 | (h3 << 16);
    chunk[16 + w] = l0 | (l1 << 16); chunk[16 + w + 1] = l2 | (l3 << 16);
}

// Device form on the hardware converter: v_cvt_pk_bf16_f32 rounds two fp32 to a packed bf16 pair (RNE, the same
// bits as msr_bf16_rn for finite values), so a pair costs 5 VALU instructions instead of ~28.
typedef __bf16 msr_bf16x2 __attribute__((ext_vector_type(2)));
typedef float msr_f32x2 __attribute__((ext_vector_type(2)));
__device__ inline void msr_split_bf16_pk(float v0, float v1, unsigned& hiw, unsigned& low) {
    const msr_f32x2 v = {v0, v1};
    hiw = __builtin_bit_cast(unsigned, __builtin_convertvector(v, msr_bf16x2));
    const msr_f32x2 hf = {__builtin_bit_cast(float, hiw << 16), __builtin_bit_cast(float, hiw & 0xFFFF0000u)};
    low = __builtin_bit_cast(unsigned, __builtin_convertvector(v - hf, msr_bf16x2));
}
__device__ inline void msr_store_split4_dev(float* pixel, int c, float v0, float v1, float v2, float v3) {
    unsigned h01, l01, h23, l23;
    msr_split_bf16_pk(v0, v1, h01, l01);
    msr_split_bf16_pk(v2, v3, h23, l23);
    unsigned* chunk = reinterpret_cast<unsigned*>(pixel) + (c & ~31) + ((c & 31) >> 1);
    *reinterpret_cast<uint2*>(chunk) = make_uint2(h01, h23);
    *reinterpret_cast<uint2*>(chunk + 16) = make_uint2(l01, l23);
}

// fp16 twins of the split helpers (PREC_F16X2 tensors): v_cvt_f16_f32 rounds to nearest even
__host__ __device__ inline void msr_split_f16(float v, unsigned& hi, unsigned& lo) {
    const _Float16 h = (_Float16)v;
    const _Float16 l = (_Float16)(v - (float)h);
    hi = (unsigned)__builtin_bit_cast(unsigned short, h);
    lo = (unsigned)__builtin_bit_cast(unsigned short, l);
}
__host__ __device__ inline void msr_store_split4_f16(float* pixel, int c, float v0, float v1, float v2, float v3) {
    unsigned h0, l0, h1, l1, h2, l2, h3, l3;
    msr_split_f16(v0, h0, l0); msr_split_f16(v1, h1, l1); msr_split_f16(v2, h2, l2); msr_split_f16(v3, h3, l3);
    unsigned* chunk = reinterpret_cast<unsigned*>(pixel) + (c & ~31) + ((c & 31) >> 1);
    chunk[0] = h0 | (h1 << 16); chunk[1] = h2 | (h3 << 16);
    chunk[16] = l0 | (l1 << 16); chunk[17] = l2 | (l3 << 16);
}

// fp32 -> fp8 e4m3 (OCP "fn": bias 7, max 448, no infinity), round to nearest even, saturating; host side of the
// PREC_FP8 weight upload (the device side converts with v_cvt_pk_bf8_f32 / v_cvt_pk_fp8_f32)
inline unsigned char msr_f32_to_e4m3(float v) {
    const unsigned char sign = v < 0.f ? 0x80 : 0;
    float a = v < 0.f ? -v : v;
    if (!(a == a)) return 0x7F;
    if (a >= 448.f) return sign | 0x7E;
    if (a < 0.0009765625f) return sign;                      // below half the smallest subnormal (2^-9 / 2): zero
    int e;
    (void)__builtin_frexpf(a, &e);                           // a = m * 2^e, m in [0.5, 1)
    e -= 1;                                                  // a = 1.m * 2^e
    if (e < -6) e = -6;                                      // subnormal range shares the exponent of 2^-6
    const float q = __builtin_ldexpf(1.f, e - 3);            // spacing of representable values in this binade
    float r = __builtin_nearbyintf(a / q);                   // RNE in the default rounding mode
    if (r >= 16.f) { r = 8.f; e += 1; }                      // carried into the next binade
    if (e > 8 || (e == 8 && r > 14.f)) return sign | 0x7E;
    if (r < 8.f) return sign | (unsigned char)r;             // subnormal: exponent field 0, mantissa r
    return sign | (unsigned char)(((e + 7) << 3) | ((int)r - 8));
}

// fp32 -> fp6 e2m3 code (sign | 2 exponent bits, bias 1 | 3 mantissa bits; max 7.5, subnormal step 1/8), round to nearest even,
// saturating; host side of the PREC_F16C6 weight upload
inline unsigned msr_f32_to_e2m3(float v) {
    const unsigned sign = v < 0.f ? 0x20u : 0u;
    float a = v < 0.f ? -v : v;
    if (!(a == a)) return sign | 0x1F;
    if (a >= 7.5f) return sign | 0x1F;
    if (a < 1.f) return sign | (unsigned)__builtin_nearbyintf(a * 8.f);       // subnormals (and 1.0 = code 8 by carry)
    int e;
    (void)__builtin_frexpf(a, &e);                                             // a = m * 2^e, m in [0.5, 1)
    e -= 1;                                                                    // a = 1.m * 2^e, e in 0..2
    float r = __builtin_nearbyintf(__builtin_ldexpf(a, 3 - e));                // 8..16
    if (r >= 16.f) { r = 8.f; e += 1; }
    if (e > 2) return sign | 0x1F;
    return sign | (unsigned)(((e + 1) << 3) | ((int)r - 8));
}

// Device: four values already divided by the block scale (|q| <= 7.5) -> their four e2m3 codes packed in 24 bits.  The
// hardware converter does the rounding: e4m3 of q * 2^-6 has the same three mantissa bits and subnormal grid, and its
// exponent field stays below 4, so the code is the e4m3 byte's low five bits plus the sign.
__device__ inline unsigned msr_pack_e2m3x4_dev(float q0, float q1, float q2, float q3) {
    unsigned c8 = 0;
    c8 = __builtin_amdgcn_cvt_pk_fp8_f32(q0 * 0.015625f, q1 * 0.015625f, c8, false);
    c8 = __builtin_amdgcn_cvt_pk_fp8_f32(q2 * 0.015625f, q3 * 0.015625f, c8, true);
    const unsigned six = (c8 & 0x1F1F1F1Fu) | ((c8 >> 2) & 0x20202020u);
    return (six & 0x3Fu) | ((six >> 2) & 0xFC0u) | ((six >> 4) & 0x3F000u) | ((six >> 6) & 0xFC0000u);
}
// e8m0 byte of the block scale 2^E >= amax / 7.5 (E = ceil(log2(amax / 7.5)), clamped; amax = 0 -> 2^0)
__device__ inline int msr_block_e8m0_dev(float amax) {
    const int bits = __builtin_bit_cast(int, amax * (1.f / 7.5f)) + 0x7FFFFF;   // a non-zero mantissa carries into the exponent
    int eb = (bits >> 23) & 0xFF;
    eb = amax > 0.f ? eb : 127;
    return eb < 27 ? 27 : (eb > 247 ? 247 : eb);
}

// First statement of every kernel that converts activations to fp16 / fp8 pieces: MODE.FP16_OVFL = 1.  By default the fp8
// converters return NaN beyond the format's range (v_cvt_pk_fp8_f32(480) = 0x7f) and v_cvt_f16_f32 returns infinity; with the
// bit set they saturate (448 / 57344 / 65504; tools/gpu_diag_fp8_ovfl.hip, profiles/r03_fp8_conversion_overflow.txt).  A
// saturated piece costs accuracy (tests/test_gpu_conv_kernel.py::test_f16c_saturation_regimes); a NaN piece poisons the tile.
// MODE is per-wave state, initialised at wave launch: nothing outlives the kernel.
#define MSR_SATURATING_CONVERSIONS() asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1")

// Device: 4 consecutive channels c..c+3 (c % 4 == 0) of one pixel into the pixel's f16c chunk image (PREC_F16C above).
__device__ inline void msr_store_f16c4_dev(float* pixel, int c, float v0, float v1, float v2, float v3) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    // fp16's range, not fp32's: a value beyond 65504 saturates (finite, wrong) instead of turning the conv into infinities;
    // NaN passes through (fminf / fmaxf would drop it, the ternaries keep it)
    v0 = v0 > 65504.f ? 65504.f : (v0 < -65504.f ? -65504.f : v0);
    v1 = v1 > 65504.f ? 65504.f : (v1 < -65504.f ? -65504.f : v1);
    v2 = v2 > 65504.f ? 65504.f : (v2 < -65504.f ? -65504.f : v2);
    v3 = v3 > 65504.f ? 65504.f : (v3 < -65504.f ? -65504.f : v3);
    const h2 a = {(_Float16)v0, (_Float16)v1}, b = {(_Float16)v2, (_Float16)v3};
    const float l0 = v0 - (float)a[0], l1 = v1 - (float)a[1], l2 = v2 - (float)b[0], l3 = v3 - (float)b[1];
    unsigned* chunk = reinterpret_cast<unsigned*>(pixel) + (c & ~31);
    const int cc = c & 31;
    unsigned h8 = 0, l8 = 0;
    h8 = __builtin_amdgcn_cvt_pk_fp8_f32(v0, v1, h8, false);
    h8 = __builtin_amdgcn_cvt_pk_fp8_f32(v2, v3, h8, true);
    l8 = __builtin_amdgcn_cvt_pk_fp8_f32(l0 * 2048.f, l1 * 2048.f, l8, false);
    l8 = __builtin_amdgcn_cvt_pk_fp8_f32(l2 * 2048.f, l3 * 2048.f, l8, true);
#ifdef MSR_WI_ONESTORE   // what-if build (wrong layout): the lane's 16 bytes as one store
    *reinterpret_cast<uint4*>(chunk + cc) = make_uint4(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), h8, l8);
    return;
#endif
#ifdef MSR_WI_NOSTORE    // what-if build (no output)
    asm volatile("" : : "v"(h8), "v"(l8), "v"(chunk));
    return;
#endif
    *reinterpret_cast<uint2*>(chunk + (cc >> 1)) = make_uint2(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b));
    chunk[16 + (cc >> 2)] = h8;                                    // bytes 64..95: h8 of the 32 channels
    chunk[24 + (cc >> 2)] = l8;                                    // bytes 96..127: l8
}

enum ConvTile : int { TILE_128x128 = 0, TILE_64x64 = 1, TILE_128x128_K16 = 2, TILE_128x128_HALO = 3, TILE_128x128_HALO16 = 4, TILE_256x128_PP = 5 };

// Which tiles an XCD's 32 CUs hold at the same time decides what its 4 MB L2 must serve: a pixel tile's input halo is
// shared by the CUs that work on its channel blocks, a channel block's weights by the CUs that work on different pixel
// tiles.  Channel block fastest (all of a pixel tile's N / 128 blocks side by side) streams the WHOLE weight tensor
// through the L2 once per round of 32 tiles: measured (profiles/r03_spade512_f16c_traffic_by_layer_before.txt) 604 MB of
// weight re-reads per launch on the N = 2048 / 1024 gamma|beta convs, 3-4x their algorithmic bytes.  With 8 pixel tiles x
// 4 channel blocks per round the weights are read once per 8 pixel tiles and a halo at most N / 512 times: per-XCD bytes
// ~ w / 8 + h * 8 / 32 per tile instead of w / 2 + h / 16 (N = 2048; w = 590 KB of weights per channel block at Cin = 128,
// h = 166 KB of halo), the minimum over the splits of 32.  MSR_TILE_WALK=0 restores channel block fastest (A/B runs).
void conv_walk_pick(int tiles_m, int tiles_n, int* walk_pb, int* walk_nb);
// tile number -> (channel block, pixel tile) under the walk
#define MSR_WALK(g, t_, tn_, tmi_)                                                               \
    {                                                                                            \
        const int grp_ = (g).walk_pb * (g).tiles_n, sub_ = (g).walk_pb * (g).walk_nb;            \
        const int blk_ = (t_) / grp_, rem_ = (t_) - blk_ * grp_;                                 \
        const int ng_ = rem_ / sub_, j_ = rem_ - ng_ * sub_;                                     \
        const int jp_ = j_ / (g).walk_nb;                                                        \
        tn_ = ng_ * (g).walk_nb + (j_ - jp_ * (g).walk_nb);                                      \
        tmi_ = blk_ * (g).walk_pb + jp_;                                                         \
    }

// conv_gbr.hip: one SPADE layer's modulation path in one kernel for the launches that fill the chip — nearest resize of the
// call's 2-channel input + mask-embedding conv + ReLU (spade.py:17-18) computed per 16 x 16 pixel tile into LDS (f16c6
// operand), the gamma|beta conv (spade.py:19-20) swept over it for all output channels, SPADE epilogue (spade.py:21-24).
struct GbrParams {
    const float* src;       // [B, S, S, 2] the call's input
    const float* we;        // mask-embedding conv kernel, HWIO [3][3][2][128] (kept for reference / debugging; the kernel reads we16)
    const void* we16;       // the same kernel as the A operands of phase 1's four v_mfma_f32_32x32x16_f16: [chunk 4][instr 4][lane 64]
                            // x 8 fp16 (16 KB, conv_gbr_embed_image): K slot pair P = 8 j + 4 (lane >> 5) + u carries term P / 9
                            // (0: w_hi for x_hi, 1: w_hi for x_lo, 2: w_lo for x_hi) of tap P % 9, both mask channels; P >= 27 zero
    const float* be;        // its bias [128]
    int S, f, o;            // source size; nearest resize to r x r: source index = t * f + o (f = S / r, o = f / 2)
    const float* wt;        // gamma|beta weights: PREC_F16C6 image of [9][N][128] (columns interleaved 32 gamma | 32 beta)
    const float* bias;      // [N] in GEMM column order
    const float* aux;       // x, the tensor being normalised: [B, r >> aux_shift, r >> aux_shift, N / 2] dense
    int aux_px, aux_py, aux_pb, aux_shift;
    const float* mean;      // [N / 2]
    const float* stdv;      // [N / 2] sqrt(var + eps)
    float* out;             // zero-bordered input of the consumer conv (f16c chunk image)
    int out_px, out_py, out_pb, out_off;
    int out_split;          // 4: the f16c chunk image (PREC_F16C consumer)
    float slope;
    int B, r, N;
    int no_cross = 0;             // 1 = leave the fp6 cross terms out ("f16" mode)
};
hipError_t conv_gbr_init();
// > 0: the layer runs conv_gb_resident with the channel blocks of a pixel tile cut into that many work items; 0: not a layer
// for it (MSR_GBR=0 switches the kernel off: A/B runs)
int conv_gbr_ranges(int B, int r, int N);
// ranges: work items per pixel tile (a divisor of N / 128); the planner passes conv_gbr_ranges()
hipError_t launch_conv_gbr(const GbrParams& p, int ranges, hipStream_t s);
// host: HWIO [3][3][2][128] fp32 -> GbrParams.we16 image (4096 floats of storage)
void conv_gbr_embed_image(const float* we_hwio, float* out4096);

hipError_t conv_igemm_init();   // sets dynamic-LDS attributes once
// conv_sw.hip: PREC_F16C whole-tile launches as one software-pipelined wave per SIMD.  launch_conv_igemm sends it the
// long-K main convs (bias / residual epilogues, Cin % 128 == 0); MSR_F16C_SW = 0 keeps everything on the ping-pong kernel,
// 2 sends the gamma|beta convs there too (A/B runs)
hipError_t conv_sw_init();
hipError_t launch_conv_f16c_sw(const ConvParams& p, int epilogue, hipStream_t s);
hipError_t launch_conv_igemm(const ConvParams& p, int epilogue, int tile, hipStream_t s);
// picks the tile and the K split for a problem size (fills the chip for the low-resolution layers)
int conv_pick_tile(int M, int N, int epilogue, int prec, int ksteps = 0);
int conv_pick_ksplit(int M, int N, int ksteps, int tile, int prec = PREC_F32);
// number of partial-moment slabs the fused epilogue of this launch writes (0 if the shape is not tileable)
int conv_stat_slabs(const ConvParams& p, int tile);
// Chan-combines the slabs in fp64: mean and sqrt(biased var + eps) per channel (deterministic order)
// group_ws: 32 * 3 * C doubles of scratch
hipError_t launch_moments_from_slabs(const float* partial, int P, int C, float eps, double* group_ws, float* mean,
                                     float* stdv, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// Small kernels (memory-bound or tiny)
// ---------------------------------------------------------------------------------------------
// Direct 3x3 conv from the 2-channel source: SPADE mask embedding fused with the half-pixel
// nearest resize (spade.py:17-18) and the first encoder block (networks.py:16-18).
//   t = y*ay + kh + cy (valid iff 0 <= t < lim); source row = t*f + o.  Same along x.
struct SmallCinParams {
    const float* src;   // [B,S,S,2]
    const float* w;     // HWIO [3,3,2,Cout]
    const float* bias;  // [Cout] or nullptr
    float* out;
    int B, S, Hout, Cout;
    int ay, cy, lim, f, o;
    int out_px, out_py, out_pb, out_off;
    int act;            // 0 none, 1 relu, 2 leaky(slope)
    float slope;
    int out_split;      // 1: write split-bf16 words for a PREC_BF16X3 consumer; 2: split-fp16 words (PREC_F16X2);
                        // 3: bf8 e5m2 bytes (PREC_FP8; one dword per 4 channels); 4: the f16c chunk image (PREC_F16C)
};
hipError_t launch_conv_smallcin(const SmallCinParams& p, hipStream_t s);
hipError_t launch_split_bf16(const float* in, float* out, long n, hipStream_t s);

// Per-group, per-channel moments of x [G, P, C]: mean and sqrt(var_biased + eps) (fp64 accumulation).
//   G=1 -> tf.nn.moments over (N,H,W) (spade.py:21);  G=B -> tfa InstanceNormalization (blocks.py:63).
// partial must hold G * chunks * C * 2 doubles with chunks = moments_chunks(G, P).
int moments_chunks(int G, int P);
hipError_t launch_moments(const float* x, int G, int P, int C, float eps, double* partial, float* mean,
                          float* stdv, hipStream_t s);

// out = act(((x - mean[g,c]) / std[g,c]) * gamma[c] + beta[c]) written with arbitrary output pitches.
struct NormActParams {
    const float* x;       // [B, H, W, C] dense
    const float* mean;    // [B, C]
    const float* stdv;    // [B, C]
    const float* gamma;   // [C]
    const float* beta;    // [C]
    float* out;
    int B, H, W, C;
    int out_px, out_py, out_pb, out_off;
    float slope;
    int out_split;        // write split-bf16 words for a PREC_BF16X3 consumer
};
hipError_t launch_norm_act(const NormActParams& p, hipStream_t s);

// y[b, n] = sum_k x[b, k] * W[k, n] (+ bias) for tiny M = B <= 16: split-K weight streaming.
size_t dense_partial_floats(int B, int K, int N);
hipError_t launch_dense(const float* x, const float* W, const float* bias, float* partial, float* y, int B, int K,
                        int N, hipStream_t s);

// z = mean + exp(0.5*var)*eps (sampling.py:16) or mean + var (model.py:267); mv = [B, 2*L] (mean | var).
hipError_t launch_latent(const float* mv, const float* eps, float* z, int B, int L, int use_sampler,
                         hipStream_t s);

// Head: leaky_relu -> UpSampling2D(2) -> Conv2D(1, 4, 'same') (+tanh) fused (networks.py:54-56).
//   x [B, r, r, C] at half resolution, weff = [2][2][3][3][C] effective per-parity weights, out [B, 2r, 2r].
hipError_t launch_head(const float* x, const float* weff, float bias, float* out, int B, int r, int C, float slope,
                       int tanh_out, int x_py, int x_pb, hipStream_t s);   // x_py = 0: dense [B,r,r,C]

// ---------------------------------------------------------------------------------------------
// Generic direct convolution (pix2pix plumbing config: Conv2D / Conv2DTranspose 4x4 s2, BN, act, concat)
// ---------------------------------------------------------------------------------------------
struct DirectConvParams {
    const float* in0; int c0;     // first input: channels [0, c0) of pixel (b, y, x) at in0 + b*in_pb + y*in_py + x*in_px
    const float* in1; int c1;     // optional second input (channel concat), same pitches, c1 = 0 if none
    int in_px, in_py, in_pb;      // input pitches in floats
    const float* w;               // conv: [KH,KW,Cin,Cout]; transposed: [KH,KW,Cout,Cin]
    const float* scale;           // per-Cout affine (folded BN) or nullptr
    const float* shift;           // per-Cout shift (folded BN / bias) or nullptr
    float* out;                   // output pixel (b, y, x) channel co at out + b*out_pb + y*out_py + x*out_px + co
    int out_px, out_py, out_pb;
    int B, Hin, Win, Hout, Wout, Cout;
    int KH, KW, stride, pad;      // pad = padding before (TF SAME)
    int transposed;
    int act;                      // 0 none, 1 relu, 2 leaky, 3 tanh
    float slope;
};
hipError_t launch_conv_direct(const DirectConvParams& p, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// Tiler / stitcher
// ---------------------------------------------------------------------------------------------
hipError_t launch_patch_stats(const float* img, const float* dem, int rows, int cols, const int* ox, const int* oy,
                              int n, int S, float no_value, uint8_t* valid, float* minmax, hipStream_t s);
hipError_t launch_extract_patches(const float* img, const float* dem, int rows, int cols, const int* ox,
                                  const int* oy, const float* minmax, int n, int S, float* out, hipStream_t s);
hipError_t launch_compact_patches(const uint8_t* valid, const int* ox, const int* oy, const float* minmax, int n,
                                  int tile_x, int tile_y, int B, int cap, int* sel_x, int* sel_y, float* sel_mm,
                                  int* key, float* dmm, int* meta, hipStream_t s);
hipError_t launch_stitch_tile(const float* pred, const int* key, const float* dmm, int n, int S, int T, int stride,
                              float no_value, int as_implemented, const double* window, int* grid_ws,
                              float* mean, float* stdv, uint8_t* good, hipStream_t s, float* wsum_partial = nullptr,
                              int pitch = 0, int resume = 0);   // partial output only: row pitch of the three accumulator
                                                                // images (0 = T), resume = start from their current content
hipError_t launch_halo_merge(const float* wa, const float* ma, const float* sa, const float* wb, const float* mb,
                             const float* sb, long n, float no_value, float* mean, float* stdv, uint8_t* good,
                             hipStream_t s);
hipError_t launch_resize_area(const float* src, int h, int w, float* dst, int dh, int dw, int factor, hipStream_t s);
hipError_t launch_resize_cubic(const float* src, int h, int w, float* dst, int dh, int dw, hipStream_t s);

}  // namespace msr
