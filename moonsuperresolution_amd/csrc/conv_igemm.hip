// conv_igemm — NHWC implicit-GEMM convolution for gfx950: exact fp32 on v_mfma_f32_32x32x2_f32, or 3-term
// split-bf16 ("bf16x3") on v_mfma_f32_32x32x16_bf16 (see kernels.h ConvPrecision).
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
#include <cstdlib>

namespace msr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// Workgroup barrier of the ping-pong kernel, spelled as what it is on gfx950: a workgroup-scope release (every LDS
// store of this wave has completed: s_waitcnt lgkmcnt(0)), the hardware s_barrier, a workgroup-scope acquire.  That is
// exactly what __syncthreads() lowers to, but the ping-pong schedule executes its barriers under WAVE-GROUP-dependent
// control flow (group Y runs one barrier more at the start and one fewer at the end), which __syncthreads() — defined
// for barriers every thread reaches at the same textual call — does not promise to support.  s_barrier itself only
// counts arrivals: it releases when every wave of the workgroup has executed one more s_barrier, wherever that
// instruction sits in its stream.  The counts are balanced by construction (table at the kernel).
#define MSR_WG_BARRIER()                                       \
    {                                                          \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); \
        __builtin_amdgcn_s_barrier();                          \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); \
    }

// K-step = BKC channels of one tap; LDS rows are BKC + 4 floats.  Both pitches (36 and 20 floats) put the 16
// lanes of a ds_read_b128 group on 16 distinct 16-byte slots, i.e. the fragment reads are conflict-free.

__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
    // Blocks are dealt round-robin over the 8 XCDs; give each XCD a contiguous range of logical tiles so
    // that neighbouring tiles (same pixels, next channel block) share that XCD's L2.  Bijective for any nwg.
    const int q = nwg >> 3, r = nwg & 7, x = orig & 7;
    const int base = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (orig >> 3);
}

// LDS staging store of one 16-byte global item: a plain copy for both precisions (split-bf16 tensors already hold
// the [32 hi | 32 lo] chunk image in HBM, see kernels.h).
template <int PREC>
__device__ __forceinline__ void stage_store(float* dst, const float4& v) {
    *reinterpret_cast<float4*>(dst) = v;
}

struct TileGeom {
    int th_l, tw_l, tb;            // log2 tile height/width, samples per tile
    int tiles_x, tiles_y, tiles_b, tiles_n;
    int tiles_mn;                  // tiles_x * tiles_y * tiles_b * tiles_n (the grid is ksplit times that)
    // Tile walk of the persistent kernels (conv_walk, below): consecutive tile numbers cover walk_nb channel blocks of
    // walk_pb pixel tiles before they move to the next channel blocks of the same pixel tiles.  (1, tiles_n) = channel
    // block fastest (the round-2 walk).
    int walk_pb, walk_nb;
};

void conv_walk_pick(int tiles_m, int tiles_n, int* walk_pb, int* walk_nb) {
    static const bool off = std::getenv("MSR_TILE_WALK") && std::atoi(std::getenv("MSR_TILE_WALK")) == 0;
    *walk_pb = 1;
    *walk_nb = tiles_n;
    if (!off && tiles_n > 4 && tiles_n % 4 == 0 && tiles_m % 8 == 0) { *walk_pb = 8; *walk_nb = 4; }
}
static void conv_walk(TileGeom& g) { conv_walk_pick(g.tiles_x * g.tiles_y * g.tiles_b, g.tiles_n, &g.walk_pb, &g.walk_nb); }

// ------------------------------------------------------------------------------------------------------
// Epilogue shared by the fp32 and the split-bf16 kernels.
// C/D map of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5).
// Per m-tile the 16 rows a lane owns are handled in two phases — (1) addresses and ALL global loads (the tensor
// being normalised / the residual), (2) arithmetic and stores — so the loads of a tile are in flight together
// instead of one load-wait-use chain per row.  Rows outside the batch read a clamped (valid) address and are
// only masked at the store.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned lane_xor1(unsigned v) {
    // neighbour exchange lane <-> lane ^ 1 in the VALU (DPP quad_perm [1,0,3,2]), no LDS crossbar
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);
}

template <int WM, int WN, int MT, int NT, int EPI, bool SPLIT, int RB>
__device__ __forceinline__ void conv_epilogue_body(const ConvParams& p, const TileGeom& g, f32x16 (&acc)[MT][NT],
                                                   int wm, int wn, int half, int l31, int n0, int tx0, int ty0,
                                                   int b0, int stat_tile) {
    const int twm = (1 << g.tw_l) - 1, thm = (1 << g.th_l) - 1;
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
            cstd[j] = SPLIT ? 1.f / p.stdv[ccol[j]] : p.stdv[ccol[j]];   // bf16x3: multiply by 1/sigma
        } else {
            ccol[j] = n0 + (wn * NT + j) * 32 + l31;
            cb0[j] = p.bias[ccol[j]];
            cb1[j] = cmean[j] = cstd[j] = 0.f;
            if constexpr (EPI == EPI_AFFINE) cb1[j] = p.scale ? p.scale[ccol[j]] : 1.f;
        }
    }
    // fused output moments (EPI_BIAS / EPI_RES): shifted sums per lane and column, shift = the lane's first value
    float st_v0[NT], st_s1[NT], st_s2[NT], st_n = 0.f;
#pragma unroll
    for (int n = 0; n < NT; ++n) st_v0[n] = st_s1[n] = st_s2[n] = 0.f;
#pragma unroll
    for (int mr = 0; mr < MT * (16 / RB); ++mr) {
        const int m = mr / (16 / RB), r0 = (mr % (16 / RB)) * RB;   // RB rows of m-tile m per batch
        int ooff[RB];
        bool ok[RB];
        float xin[RB][NCH];
        // phase 1: addresses and loads
#pragma unroll
        for (int q = 0; q < RB; ++q) {
            const int r = r0 + q;
            const int row = (wm * MT + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const int tx = row & twm, ty = (row >> g.tw_l) & thm, tbi = row >> (g.tw_l + g.th_l);
            int bb = b0 + tbi;
            ok[q] = tbi < g.tb && bb < p.B;
            bb = bb < p.B ? bb : p.B - 1;
            const int y = ty0 + ty, x = tx0 + tx;
            ooff[q] = p.out_off + bb * p.out_pb + y * p.out_py + x * p.out_px;
            if constexpr (EPI == EPI_SPADE || EPI == EPI_RES) {
                const float* arow = p.aux + (size_t)bb * p.aux_pb + (y >> p.aux_shift) * p.aux_py +
                                    (x >> p.aux_shift) * p.aux_px;
#pragma unroll
                for (int j = 0; j < NCH; ++j) xin[q][j] = arow[ccol[j]];
            }
        }
        // phase 2: arithmetic and stores
#pragma unroll
        for (int q = 0; q < RB; ++q) {
            const int r = r0 + q;
            float* orow = p.out + ooff[q];
            if constexpr (EPI == EPI_SPADE) {
#pragma unroll
                for (int j = 0; j < NCH; ++j) {
                    const float gam = acc[m][2 * j][r] + cb0[j];
                    const float bet = acc[m][2 * j + 1][r] + cb1[j];
                    const float normalized = SPLIT ? (xin[q][j] - cmean[j]) * cstd[j] : (xin[q][j] - cmean[j]) / cstd[j];
                    float v = gam * normalized + bet;
                    v = v >= 0.f ? v : v * p.slope;
                    if constexpr (SPLIT) {
                        // lanes 0..31 of a half-wave hold the 32 channels of ONE chunk of this pixel: pair up
                        // neighbouring lanes so that every lane still issues one 4-byte store
                        unsigned hi, lo;
                        msr_split_bf16(v, hi, lo);
                        const unsigned nhi = lane_xor1(hi), nlo = lane_xor1(lo);
                        unsigned* chunk = reinterpret_cast<unsigned*>(orow) + (ccol[j] & ~31);
                        const unsigned word = (l31 & 1) ? (nlo | (lo << 16)) : (hi | (nhi << 16));
                        if (ok[q]) chunk[((l31 & 1) ? 16 : 0) + (l31 >> 1)] = word;
                    } else {
                        if (ok[q]) orow[ccol[j]] = v;
                    }
                }
            } else {
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    float v = acc[m][n][r] + cb0[n];
                    if constexpr (EPI == EPI_AFFINE) {
                        v = acc[m][n][r] * cb1[n] + cb0[n];
                        v = p.act == 1 ? fmaxf(v, 0.f) : (p.act == 2 ? (v >= 0.f ? v : v * p.slope) : v);
                    }
                    if constexpr (EPI == EPI_RES) v += xin[q][n];
                    if (ok[q]) orow[ccol[n]] = v;
                    if (mr == 0 && q == 0) st_v0[n] = v;
                    const float d = ok[q] ? v - st_v0[n] : 0.f;
                    st_s1[n] += d;
                    st_s2[n] += d * d;
                }
                st_n += ok[q] ? 1.f : 0.f;
            }
        }
    }
    if constexpr (EPI == EPI_BIAS || EPI == EPI_RES) {
        if (p.stat_partial) {
            // lane -> (count, mean, M2); lanes l and l ^ 32 hold the same columns for different rows: Chan-combine
            const int slab = (stat_tile * WM + wm);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const float cnt = st_n;
                const float inv = cnt > 0.f ? 1.f / cnt : 0.f;
                float mean = st_v0[n] + st_s1[n] * inv;
                float m2 = st_s2[n] - st_s1[n] * st_s1[n] * inv;
                const float ocnt = __shfl_xor(cnt, 32), omean = __shfl_xor(mean, 32), om2 = __shfl_xor(m2, 32);
                const float tot = cnt + ocnt;
                if (tot > 0.f) {
                    const float delta = omean - mean;
                    m2 = m2 + om2 + delta * delta * (cnt * ocnt / tot);
                    mean = mean + delta * (ocnt / tot);
                }
                if (half == 0) {
                    float* o = p.stat_partial + (size_t)slab * 3 * p.N + ccol[n];
                    o[0] = tot;
                    o[p.N] = mean;
                    o[2 * p.N] = m2 > 0.f ? m2 : 0.f;
                }
            }
        }
    }
}

template <int WM, int WN, int MT, int NT, int EPI, int RB = 16>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, const TileGeom& g, f32x16 (&acc)[MT][NT], int ks,
                                              int wm, int wn, int half, int l31, int n0, int tx0, int ty0, int b0) {
    // m-tile index of this workgroup (slab row of the fused output moments)
    const int stat_tile = ((b0 / g.tb) * g.tiles_y + (ty0 >> g.th_l)) * g.tiles_x + (tx0 >> g.tw_l);
    if constexpr (EPI == EPI_PARTIAL) {
        const int twm = (1 << g.tw_l) - 1, thm = (1 << g.th_l) - 1;
        float* pbase = p.partial + (size_t)ks * ((size_t)p.B * p.Hout * p.Wout * p.N);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (wm * MT + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int tx = row & twm, ty = (row >> g.tw_l) & thm, tbi = row >> (g.tw_l + g.th_l);
                const int bb = b0 + tbi;
                if (tbi >= g.tb || bb >= p.B) continue;
                float* orow = pbase + (((size_t)bb * p.Hout + ty0 + ty) * p.Wout + tx0 + tx) * p.N;
#pragma unroll
                for (int n = 0; n < NT; ++n) orow[n0 + (wn * NT + n) * 32 + l31] = acc[m][n][r];
            }
        }
    } else if constexpr (EPI == EPI_SPADE) {
        if (p.out_split) conv_epilogue_body<WM, WN, MT, NT, EPI, true, RB>(p, g, acc, wm, wn, half, l31, n0, tx0, ty0, b0, stat_tile);
        else conv_epilogue_body<WM, WN, MT, NT, EPI, false, RB>(p, g, acc, wm, wn, half, l31, n0, tx0, ty0, b0, stat_tile);
    } else {
        conv_epilogue_body<WM, WN, MT, NT, EPI, false, RB>(p, g, acc, wm, wn, half, l31, n0, tx0, ty0, b0, stat_tile);
    }
}

// ------------------------------------------------------------------------------------------------------
// Epilogue of the 16x16x32 halo kernels.  They issue the MFMA with the WEIGHT fragment as the row operand, i.e.
// they accumulate the transposed tile D[channel][pixel]: column = lane & 15 = pixel, row = 4 * (lane >> 4) + reg =
// channel, so a lane holds FOUR CONSECUTIVE CHANNELS of one pixel in the four registers of a sub-tile and every
// global access below is 16 bytes (8 for the split-bf16 halves).  A dword access costs the memory pipeline the
// same 16 cycles per wave-instruction as a 16-byte one: with one workgroup per CU the epilogue is exposed, and the
// dword form of it was 10-30 % of the short-K layers.
// A wave owns tile rows 4*wm .. 4*wm+3 (sub-tile i = one row of 16 pixels) and 64 output columns (sub-tile j = 16
// columns): lane (px, cg) holds pixel x = tx0 + px of each of its four rows.  The tile is always interior
// (tb == 1, r >= 16): nothing is masked.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 f4(const f32x4& v) { return make_float4(v[0], v[1], v[2], v[3]); }

// sum over the 16 lanes of a DPP row (here: the 16 pixels of a tile row), result in every lane; 4 VALU pairs
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
    return v;
}

// Everything the epilogue reads from memory, so that the ping-pong kernel can request it two K-steps before the end
// of a tile's main loop (the staging registers are idle by then) instead of paying the latency after it:
//   EPI_SPADE: xin[i][jj] = channels ch0 + 16*jj + 4*cg + {0..3} of x at pixel (y0 + i, x);
//              cv = {gamma bias, beta bias, mean, sigma} x {jj = 0, 1}
//   others   : cv[j] = bias of columns n0 + 64*wn + 16*j + 4*cg + {0..3}
template <int EPI>
__device__ __forceinline__ void halo16_epilogue_load(const ConvParams& p, float4 (&xin)[4][2], float4 (&cv)[8], int wm,
                                                     int wn, int lane, int n0, int tx0, int ty0, int b0) {
    const int px = lane & 15, cg = lane >> 4;
    if constexpr (EPI == EPI_SPADE) {
        const int x = tx0 + px, y0 = ty0 + wm * 4;
        const int ch0 = (n0 + wn * 64) >> 1;
        const float* const abase = p.aux + (size_t)b0 * p.aux_pb + (x >> p.aux_shift) * p.aux_px + ch0 + 4 * cg;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float* arow = abase + ((y0 + i) >> p.aux_shift) * p.aux_py;
            xin[i][0] = *reinterpret_cast<const float4*>(arow);
            xin[i][1] = *reinterpret_cast<const float4*>(arow + 16);
        }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int colg = n0 + wn * 64 + jj * 16 + 4 * cg;
            const int ch = ch0 + jj * 16 + 4 * cg;
            cv[jj] = *reinterpret_cast<const float4*>(p.bias + colg);
            cv[2 + jj] = *reinterpret_cast<const float4*>(p.bias + colg + 32);
            cv[4 + jj] = *reinterpret_cast<const float4*>(p.mean + ch);
            cv[6 + jj] = *reinterpret_cast<const float4*>(p.stdv + ch);
        }
    } else if constexpr (EPI != EPI_PARTIAL) {
#pragma unroll
        for (int j = 0; j < 4; ++j) cv[j] = *reinterpret_cast<const float4*>(p.bias + n0 + wn * 64 + j * 16 + 4 * cg);
    }
}

// K-split ping-pong launches (few tiles: a workgroup owns one K range of a tile): the raw accumulators of range `ks`
// go to partial[ks][B, Hout, Wout, N] with 16-byte stores; splitk_epilogue_kernel sums the ranges in a fixed order and
// applies the layer's epilogue.
__device__ __forceinline__ void halo16_epilogue_partial(const ConvParams& p, f32x4 (&acc)[4][4], int wm, int wn, int lane,
                                                        int n0, int tx0, int ty0, int b0, int ks) {
    const int px = lane & 15, cg = lane >> 4;
    const int x = tx0 + px, y0 = ty0 + wm * 4;
    float* const pbase = p.partial + (size_t)ks * ((size_t)p.B * p.Hout * p.Wout * p.N) +
                         ((size_t)b0 * p.Hout * p.Wout + x) * p.N + n0 + wn * 64 + 4 * cg;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<float4*>(pbase + (size_t)(y0 + i) * p.Wout * p.N + j * 16) = f4(acc[i][j]);
}

template <int EPI, bool SPLIT, bool OUT8 = false, bool OUTC = false>
__device__ __forceinline__ void halo16_epilogue_body(const ConvParams& p, f32x4 (&acc)[4][4], int wm, int wn, int lane,
                                                     int n0, int tx0, int ty0, int b0, int stat_tile,
                                                     float4 (&xin)[4][2], float4 (&cv)[8]) {
    const int px = lane & 15, cg = lane >> 4;
    const int x = tx0 + px, y0 = ty0 + wm * 4;
    float* const obase = p.out + (size_t)p.out_off + (size_t)b0 * p.out_pb + x * p.out_px;
    if constexpr (EPI == EPI_SPADE) {
        // columns come as (32 gamma | 32 beta) per 64: sub-tiles 0, 1 are gamma of channels ch0 + {0..15, 16..31},
        // sub-tiles 2, 3 their beta twins
        const int ch0 = (n0 + wn * 64) >> 1;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const float gq[4] = {cv[jj].x, cv[jj].y, cv[jj].z, cv[jj].w};
            const float bq[4] = {cv[2 + jj].x, cv[2 + jj].y, cv[2 + jj].z, cv[2 + jj].w};
            const float mq[4] = {cv[4 + jj].x, cv[4 + jj].y, cv[4 + jj].z, cv[4 + jj].w};
            float sq[4] = {cv[6 + jj].x, cv[6 + jj].y, cv[6 + jj].z, cv[6 + jj].w};
            if constexpr (SPLIT) {
#pragma unroll
                for (int k = 0; k < 4; ++k) sq[k] = 1.f / sq[k];          // bf16x3: multiply by 1/sigma
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float* orow = obase + (y0 + i) * p.out_py;
                const float xq[4] = {xin[i][jj].x, xin[i][jj].y, xin[i][jj].z, xin[i][jj].w};
                float v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float normalized = SPLIT ? (xq[k] - mq[k]) * sq[k] : (xq[k] - mq[k]) / sq[k];
                    const float t = (acc[i][jj][k] + gq[k]) * normalized + (acc[i][jj + 2][k] + bq[k]);
                    v[k] = t >= 0.f ? t : t * p.slope;
                }
                if constexpr (OUTC) {
                    msr_store_f16c4_dev(orow, ch0 + jj * 16 + 4 * cg, v[0], v[1], v[2], v[3]);   // PREC_F16C consumer
                } else if constexpr (OUT8) {
                    // bf8 e5m2 bytes for a PREC_FP8 consumer: 4 consecutive channels = one dword
                    unsigned w8 = 0;
                    w8 = __builtin_amdgcn_cvt_pk_bf8_f32(v[0], v[1], w8, false);
                    w8 = __builtin_amdgcn_cvt_pk_bf8_f32(v[2], v[3], w8, true);
                    reinterpret_cast<unsigned*>(orow)[(ch0 + jj * 16 + 4 * cg) >> 2] = w8;
                } else if constexpr (SPLIT) {
                    // chunk image of the pixel: 16 words of hi pairs, 16 words of lo pairs
                    unsigned h01, l01, h23, l23;
                    msr_split_bf16_pk(v[0], v[1], h01, l01);
                    msr_split_bf16_pk(v[2], v[3], h23, l23);
                    unsigned* chunk = reinterpret_cast<unsigned*>(orow) + ch0 + jj * 8 + 2 * cg;
                    *reinterpret_cast<uint2*>(chunk) = make_uint2(h01, h23);
                    *reinterpret_cast<uint2*>(chunk + 16) = make_uint2(l01, l23);
                } else {
                    *reinterpret_cast<float4*>(orow + ch0 + jj * 16 + 4 * cg) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
    } else {
        // Fused output moments: sums of d = v - bias (a per-channel constant shift, the same in every lane, so the
        // 16 lanes of a column group add up directly) and d^2 over the 64 pixels of the wave.
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = n0 + wn * 64 + j * 16 + 4 * cg;
            const float bq[4] = {cv[j].x, cv[j].y, cv[j].z, cv[j].w};
            float4 res[4];
            if constexpr (EPI == EPI_RES) {
                const float* abase = p.aux + (size_t)b0 * p.aux_pb + (x >> p.aux_shift) * p.aux_px + col;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    res[i] = *reinterpret_cast<const float4*>(abase + ((y0 + i) >> p.aux_shift) * p.aux_py);
            }
            float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float d[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) d[k] = acc[i][j][k];
                if constexpr (EPI == EPI_RES) { d[0] += res[i].x; d[1] += res[i].y; d[2] += res[i].z; d[3] += res[i].w; }
                *reinterpret_cast<float4*>(obase + (y0 + i) * p.out_py + col) =
                    make_float4(d[0] + bq[0], d[1] + bq[1], d[2] + bq[2], d[3] + bq[3]);
#pragma unroll
                for (int k = 0; k < 4; ++k) { s1[k] += d[k]; s2[k] += d[k] * d[k]; }
            }
            if (p.stat_partial) {
                float mean[4], m2[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float t1 = row16_sum(s1[k]), t2 = row16_sum(s2[k]);
                    mean[k] = bq[k] + t1 * (1.f / 64.f);
                    const float t = t2 - t1 * t1 * (1.f / 64.f);
                    m2[k] = t > 0.f ? t : 0.f;
                }
                if (px == 0) {
                    const int slab = stat_tile * 2 + wm;       // one slab per (8-row m-tile, wm): 64 pixels
                    float* o = p.stat_partial + (size_t)slab * 3 * p.N + col;
                    *reinterpret_cast<float4*>(o) = make_float4(64.f, 64.f, 64.f, 64.f);
                    *reinterpret_cast<float4*>(o + p.N) = make_float4(mean[0], mean[1], mean[2], mean[3]);
                    *reinterpret_cast<float4*>(o + 2 * p.N) = make_float4(m2[0], m2[1], m2[2], m2[3]);
                }
            }
        }
    }
}

// EPI_SPADE writing the f16c chunk image, assembled per pixel in LDS.  A lane's 4 channels are three pieces of the pixel's
// 128-byte chunk (8 bytes of fp16, 4 of h8, 4 of l8): stored straight from the lane that is SIX store instructions per tile
// row, each touching 16 lines with 4- or 8-byte pieces, and the epilogue is store-ISSUE-bound (tools/gpu_pp_stamps_gb.py: a
// gamma|beta tile takes 87.4k cycles, 79.4k with one 16-byte store per lane, 77.4k with none; MI355X_MICROARCH.md
// "epilogue store tail").  Here the wave writes the pieces of one tile row (16 pixels x 32 channels = 16 chunk lines) into
// a private 2.3 KB LDS image, reads each line back as two 16-byte quarters per lane and issues TWO stores per row, each
// 64 contiguous bytes per pixel.  Private to the wave (LDS operations of one wave execute in order): no barrier.
// F6 = true writes the PREC_F16C6 image (kernels.h): the block scale of a pixel's 32 channels needs the maximum over the four
// lanes that share the pixel (16 lanes apart) — they exchange it through the four pad dwords of the pixel's staged line — and
// the 6-bit codes of a lane's four channels are three bytes of the line, written as bytes.
template <bool F6>
__device__ __forceinline__ void halo16_epilogue_spade_f16c_staged(const ConvParams& p, f32x4 (&acc)[4][4], int wm, int wn,
                                                                  int lane, int n0, int tx0, int ty0, int b0,
                                                                  float4 (&xin)[4][2], float4 (&cv)[8], unsigned* stage) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    constexpr int SP = 36;                                     // dwords per staged line (32 + pad: spreads the pixels over banks)
    const int px = lane & 15, cg = lane >> 4;
    const int x = tx0 + px, y0 = ty0 + wm * 4;
    const int ch0 = (n0 + wn * 64) >> 1;                       // first of the wave's 32 output channels: one whole chunk
    float* const obase = p.out + (size_t)p.out_off + (size_t)b0 * p.out_pb + x * p.out_px + ch0;
    unsigned* const line = stage + px * SP;
    float rs[2][4];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        rs[jj][0] = 1.f / cv[6 + jj].x; rs[jj][1] = 1.f / cv[6 + jj].y; rs[jj][2] = 1.f / cv[6 + jj].z; rs[jj][3] = 1.f / cv[6 + jj].w;
    }
    if constexpr (F6) {     // the zero bytes behind the two scale bytes of a line (dwords 23 and 31) never change
        if (cg == 0) { line[23] = 0u; line[31] = 0u; }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float v[2][4];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const float gq[4] = {cv[jj].x, cv[jj].y, cv[jj].z, cv[jj].w};
            const float bq[4] = {cv[2 + jj].x, cv[2 + jj].y, cv[2 + jj].z, cv[2 + jj].w};
            const float mq[4] = {cv[4 + jj].x, cv[4 + jj].y, cv[4 + jj].z, cv[4 + jj].w};
            const float xq[4] = {xin[i][jj].x, xin[i][jj].y, xin[i][jj].z, xin[i][jj].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float normalized = (xq[k] - mq[k]) * rs[jj][k];
                const float t = (acc[i][jj][k] + gq[k]) * normalized + (acc[i][jj + 2][k] + bq[k]);
                float u = t >= 0.f ? t : t * p.slope;
                v[jj][k] = u > 65504.f ? 65504.f : (u < -65504.f ? -65504.f : u);           // as msr_store_f16c4_dev
            }
        }
        float inv = 1.f;        // F6: 2^-E of the pixel's block scale
        if constexpr (F6) {
            float m = 0.f;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int k = 0; k < 4; ++k) m = fmaxf(m, fabsf(v[jj][k]));       // NaN: dropped here, kept in the fp16 piece
            line[32 + cg] = __builtin_bit_cast(unsigned, m);
            asm volatile("" ::: "memory");
            const uint4 mm = *reinterpret_cast<const uint4*>(line + 32);
            asm volatile("" ::: "memory");
            const float amax = fmaxf(fmaxf(__builtin_bit_cast(float, mm.x), __builtin_bit_cast(float, mm.y)),
                                     fmaxf(__builtin_bit_cast(float, mm.z), __builtin_bit_cast(float, mm.w)));
            const int eb = msr_block_e8m0_dev(amax);
            inv = __builtin_bit_cast(float, (254 - eb) << 23);                   // 2^-(eb - 127)
            if (cg == 0) line[22] = (unsigned)eb;                                // byte 88: the h6 piece's e8m0
            if (cg == 1) line[30] = (unsigned)(eb - 11);                         // byte 120: the l6 piece's
        }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const h2 a = {(_Float16)v[jj][0], (_Float16)v[jj][1]}, b = {(_Float16)v[jj][2], (_Float16)v[jj][3]};
            const float l0 = (v[jj][0] - (float)a[0]) * 2048.f, l1 = (v[jj][1] - (float)a[1]) * 2048.f;
            const float l2 = (v[jj][2] - (float)b[0]) * 2048.f, l3 = (v[jj][3] - (float)b[1]) * 2048.f;
            // the chunk image of the pixel: dwords 0..15 fp16 pairs (channel 16 jj + 4 cg + {0..3})
            *reinterpret_cast<uint2*>(line + jj * 8 + 2 * cg) = make_uint2(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b));
            if constexpr (F6) {
                // 4 codes = 24 bits = bytes 64 + 3 * (4 jj + cg) .. of the line (l6: 96 + ...)
                const unsigned h6 = msr_pack_e2m3x4_dev(v[jj][0] * inv, v[jj][1] * inv, v[jj][2] * inv, v[jj][3] * inv);
                const unsigned l6 = msr_pack_e2m3x4_dev(l0 * inv, l1 * inv, l2 * inv, l3 * inv);
                unsigned char* lb = reinterpret_cast<unsigned char*>(line) + 3 * (4 * jj + cg);
                lb[64] = (unsigned char)h6; lb[65] = (unsigned char)(h6 >> 8); lb[66] = (unsigned char)(h6 >> 16);
                lb[96] = (unsigned char)l6; lb[97] = (unsigned char)(l6 >> 8); lb[98] = (unsigned char)(l6 >> 16);
            } else {
                unsigned h8 = 0, l8 = 0;
                h8 = __builtin_amdgcn_cvt_pk_fp8_f32(v[jj][0], v[jj][1], h8, false);
                h8 = __builtin_amdgcn_cvt_pk_fp8_f32(v[jj][2], v[jj][3], h8, true);
                l8 = __builtin_amdgcn_cvt_pk_fp8_f32(l0, l1, l8, false);
                l8 = __builtin_amdgcn_cvt_pk_fp8_f32(l2, l3, l8, true);
                line[16 + jj * 4 + cg] = h8;                   // dwords 16..23 h8, 24..31 l8
                line[24 + jj * 4 + cg] = l8;
            }
        }
        // quarter cg of each half of the line (compiler barriers: the pieces were written through other types)
        asm volatile("" ::: "memory");
        const uint4 q0 = *reinterpret_cast<const uint4*>(line + 4 * cg);
        const uint4 q1 = *reinterpret_cast<const uint4*>(line + 16 + 4 * cg);
        asm volatile("" ::: "memory");
        unsigned* orow = reinterpret_cast<unsigned*>(obase + (y0 + i) * p.out_py);
        *reinterpret_cast<uint4*>(orow + 4 * cg) = q0;
        *reinterpret_cast<uint4*>(orow + 16 + 4 * cg) = q1;
    }
}

template <int EPI>
__device__ __forceinline__ void halo16_epilogue(const ConvParams& p, const TileGeom& g, f32x4 (&acc)[4][4], int wm, int wn,
                                                int lane, int n0, int tx0, int ty0, int b0, float4 (&xin)[4][2],
                                                float4 (&cv)[8], unsigned* stage = nullptr) {
    const int stat_tile = (b0 * g.tiles_y + (ty0 >> g.th_l)) * g.tiles_x + (tx0 >> g.tw_l);
    if constexpr (EPI == EPI_SPADE) {
        if (p.out_split == 5 && stage) halo16_epilogue_spade_f16c_staged<true>(p, acc, wm, wn, lane, n0, tx0, ty0, b0, xin, cv, stage);
        else if (p.out_split == 4 && stage) halo16_epilogue_spade_f16c_staged<false>(p, acc, wm, wn, lane, n0, tx0, ty0, b0, xin, cv, stage);
        else if (p.out_split == 4) halo16_epilogue_body<EPI, true, false, true>(p, acc, wm, wn, lane, n0, tx0, ty0, b0, stat_tile, xin, cv);
        else if (p.out_split == 3) halo16_epilogue_body<EPI, true, true>(p, acc, wm, wn, lane, n0, tx0, ty0, b0, stat_tile, xin, cv);
        else if (p.out_split) halo16_epilogue_body<EPI, true>(p, acc, wm, wn, lane, n0, tx0, ty0, b0, stat_tile, xin, cv);
        else halo16_epilogue_body<EPI, false>(p, acc, wm, wn, lane, n0, tx0, ty0, b0, stat_tile, xin, cv);
    } else {
        halo16_epilogue_body<EPI, false>(p, acc, wm, wn, lane, n0, tx0, ty0, b0, stat_tile, xin, cv);
    }
}

template <int WM, int WN, int MT, int NT, int BKC, int EPI, int PREC>
__global__ void __launch_bounds__(WM * WN * 64) __attribute__((amdgpu_waves_per_eu(2, BKC == 16 ? 3 : 2)))
conv_igemm(const ConvParams p, const TileGeom g) {
    MSR_SATURATING_CONVERSIONS();
    static_assert(PREC == PREC_F32 || BKC == 32, "the split-bf16 path uses the 32-channel K-step");
    constexpr int NTHR = WM * WN * 64;
    constexpr int BM = WM * MT * 32;
    constexpr int BN = WN * NT * 32;
    constexpr int BKP = BKC + 4;
    constexpr int SEGS = BKC / 4;            // 16-byte segments per staged row
    constexpr int A_ITEMS = BM * SEGS / NTHR;
    constexpr int B_ITEMS = BN * SEGS / NTHR;
    static_assert(BM * SEGS % NTHR == 0 && BN * SEGS % NTHR == 0, "staging split");
    static_assert(EPI != EPI_SPADE || NT % 2 == 0, "SPADE epilogue pairs gamma/beta sub-tiles");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const As = smem;                      // [2][BM][BKP]
    float* const Bs = smem + 2 * BM * BKP;       // [2][BN][BKP]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int half = lane >> 5, l31 = lane & 31;

    const int bid_all = xcd_remap(blockIdx.x, gridDim.x);
    const int ks = bid_all / g.tiles_mn;          // split-K range index (0 when ksplit == 1)
    const int bid = bid_all - ks * g.tiles_mn;
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
        const int row = idx / SEGS, seg = idx % SEGS;
        const int tx = row & twm, ty = (row >> g.tw_l) & thm, tbi = row >> (g.tw_l + g.th_l);
        int b = b0 + tbi;
        b = b < p.B ? b : p.B - 1;   // rows past the batch read valid memory and are dropped in the epilogue
        a_goff[q] = b * p.in_pb + (ty0 + ty) * p.stride * p.in_py + (tx0 + tx) * p.stride * p.in_px + seg * 4;
        a_loff[q] = row * BKP + seg * 4;
    }
    int b_goff[B_ITEMS];
    int b_loff[B_ITEMS];
#pragma unroll
    for (int q = 0; q < B_ITEMS; ++q) {
        const int idx = tid + q * NTHR;
        const int row = idx / SEGS, seg = idx % SEGS;
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

    // K-step iterator, kept as running scalars (no divisions in the loop).  With split-K this workgroup owns steps
    // [t_begin, t_end).  Order: BKC = 32 -> channel chunk outer, tap inner; BKC = 16 -> tap outer, chunk inner, so
    // that consecutive steps read the two 64-byte halves of the same 128-byte lines.
    const int t_begin = (int)((long)ks * steps / p.ksplit), t_end = (int)((long)(ks + 1) * steps / p.ksplit);
    int it_kh, it_kw, it_cc;
    const float* a_src;
    const float* b_src;
    if constexpr (BKC == 32) {
        const int cc0 = t_begin / taps, tap0 = t_begin - cc0 * taps;
        it_cc = cc0; it_kh = tap0 / p.KW; it_kw = tap0 - it_kh * p.KW;
        a_src = p.in + (it_kh * p.in_py + it_kw * p.in_px + cc0 * BKC);
        b_src = p.wt + ((size_t)tap0 * w_tap_stride + cc0 * BKC);
    } else {
        const int tap0 = t_begin / chunks, cc0 = t_begin - tap0 * chunks;
        it_cc = cc0; it_kh = tap0 / p.KW; it_kw = tap0 - it_kh * p.KW;
        a_src = p.in + (it_kh * p.in_py + it_kw * p.in_px + cc0 * BKC);
        b_src = p.wt + ((size_t)tap0 * w_tap_stride + cc0 * BKC);
    }

    // Named scalars, not arrays: hipcc leaves a float4 array that crosses a sched_barrier in scratch memory.
    static_assert((A_ITEMS == 4 || A_ITEMS == 2) && A_ITEMS == B_ITEMS, "staging is written for 2+2 or 4+4 items");
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
#define MSR_ISSUE_LOADS()                                                                        \
    {                                                                                            \
        ra0 = *reinterpret_cast<const float4*>(a_src + a_goff[0]);                               \
        ra1 = *reinterpret_cast<const float4*>(a_src + a_goff[1]);                               \
        if constexpr (A_ITEMS == 4) {                                                            \
            ra2 = *reinterpret_cast<const float4*>(a_src + a_goff[2]);                           \
            ra3 = *reinterpret_cast<const float4*>(a_src + a_goff[3]);                           \
        }                                                                                        \
        rb0 = *reinterpret_cast<const float4*>(b_src + b_goff[0]);                               \
        rb1 = *reinterpret_cast<const float4*>(b_src + b_goff[1]);                               \
        if constexpr (B_ITEMS == 4) {                                                            \
            rb2 = *reinterpret_cast<const float4*>(b_src + b_goff[2]);                           \
            rb3 = *reinterpret_cast<const float4*>(b_src + b_goff[3]);                           \
        }                                                                                        \
    }
#define MSR_ADVANCE()                                                                            \
    {                                                                                            \
        if constexpr (BKC == 32) {                                                               \
            ++it_kw;                                                                             \
            a_src += p.in_px;                                                                    \
            b_src += w_tap_stride;                                                               \
            if (it_kw == p.KW) {                                                                 \
                it_kw = 0;                                                                       \
                ++it_kh;                                                                         \
                a_src += p.in_py - p.KW * p.in_px;                                               \
                if (it_kh == p.KH) {                                                             \
                    it_kh = 0;                                                                   \
                    a_src += BKC - p.KH * p.in_py;                                               \
                    b_src += BKC - (size_t)taps * w_tap_stride;                                  \
                }                                                                                \
            }                                                                                    \
        } else {                                                                                 \
            ++it_cc;                                                                             \
            a_src += BKC;                                                                        \
            b_src += BKC;                                                                        \
            if (it_cc == chunks) {                                                               \
                it_cc = 0;                                                                       \
                ++it_kw;                                                                         \
                a_src += p.in_px - chunks * BKC;                                                 \
                b_src += w_tap_stride - chunks * BKC;                                            \
                if (it_kw == p.KW) {                                                             \
                    it_kw = 0;                                                                   \
                    ++it_kh;                                                                     \
                    a_src += p.in_py - p.KW * p.in_px;                                           \
                }                                                                                \
            }                                                                                    \
        }                                                                                        \
    }
#define MSR_WRITE_LDS(buf)                                                                       \
    {                                                                                            \
        float* a_ = As + (buf) * BM * BKP;                                                       \
        float* b_ = Bs + (buf) * BN * BKP;                                                       \
        stage_store<PREC>(a_ + a_loff[0], ra0);                                                  \
        stage_store<PREC>(a_ + a_loff[1], ra1);                                                  \
        if constexpr (A_ITEMS == 4) {                                                            \
            stage_store<PREC>(a_ + a_loff[2], ra2);                                              \
            stage_store<PREC>(a_ + a_loff[3], ra3);                                              \
        }                                                                                        \
        stage_store<PREC>(b_ + b_loff[0], rb0);                                                  \
        stage_store<PREC>(b_ + b_loff[1], rb1);                                                  \
        if constexpr (B_ITEMS == 4) {                                                            \
            stage_store<PREC>(b_ + b_loff[2], rb2);                                              \
            stage_store<PREC>(b_ + b_loff[3], rb3);                                              \
        }                                                                                        \
    }
#define MSR_COMPUTE(buf)                                                                         \
    {                                                                                            \
        const float* a_ = As + (buf) * BM * BKP;                                                 \
        const float* b_ = Bs + (buf) * BN * BKP;                                                 \
        if constexpr (PREC == PREC_F32) {                                                        \
            _Pragma("unroll") for (int kk = 0; kk < BKC / 8; ++kk) {                             \
                float4 fa[MT], fb[NT];                                                           \
                _Pragma("unroll") for (int m = 0; m < MT; ++m)                                   \
                    fa[m] = *reinterpret_cast<const float4*>(a_ + a_frag[m] + kk * 8);           \
                _Pragma("unroll") for (int n = 0; n < NT; ++n)                                   \
                    fb[n] = *reinterpret_cast<const float4*>(b_ + b_frag[n] + kk * 8);           \
                _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                  \
                    _Pragma("unroll") for (int m = 0; m < MT; ++m) {                             \
                        const float av = s == 0 ? fa[m].x : s == 1 ? fa[m].y : s == 2 ? fa[m].z : fa[m].w; \
                        _Pragma("unroll") for (int n = 0; n < NT; ++n) {                         \
                            const float bv = s == 0 ? fb[n].x : s == 1 ? fb[n].y : s == 2 ? fb[n].z : fb[n].w; \
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[m][n], 0, 0, 0); \
                        }                                                                        \
                    }                                                                            \
                }                                                                                \
            }                                                                                    \
        } else {                                                                                 \
            /* rows are [32 hi bf16 | 32 lo bf16 | pad]; lane (i, h) takes k = 16*kg + 8*h + {0..7} */ \
            _Pragma("unroll") for (int kg = 0; kg < 2; ++kg) {                                   \
                bf16x8 ah[MT], al[MT], bh[NT], bl[NT];                                           \
                _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                 \
                    ah[m] = *reinterpret_cast<const bf16x8*>(a_ + a_frag[m] + kg * 8);           \
                    al[m] = *reinterpret_cast<const bf16x8*>(a_ + a_frag[m] + kg * 8 + 16);      \
                }                                                                                \
                _Pragma("unroll") for (int n = 0; n < NT; ++n) {                                 \
                    bh[n] = *reinterpret_cast<const bf16x8*>(b_ + b_frag[n] + kg * 8);           \
                    bl[n] = *reinterpret_cast<const bf16x8*>(b_ + b_frag[n] + kg * 8 + 16);      \
                }                                                                                \
                _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                 \
                    _Pragma("unroll") for (int n = 0; n < NT; ++n) {                             \
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[m], bh[n], acc[m][n], 0, 0, 0); \
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bl[n], acc[m][n], 0, 0, 0); \
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bh[n], acc[m][n], 0, 0, 0); \
                    }                                                                            \
                }                                                                                \
            }                                                                                    \
        }                                                                                        \
    }

    MSR_ISSUE_LOADS();
    MSR_WRITE_LDS(0);
    __syncthreads();
    int cur = 0;
    for (int t = t_begin; t < t_end - 1; ++t) {
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

    // the 16-channel K-step variant must stay under 168 VGPRs (3 workgroups per CU): small load batches there
    conv_epilogue<WM, WN, MT, NT, EPI, (BKC == 16 ? 4 : 16)>(p, g, acc, ks, wm, wn, half, l31, n0, tx0, ty0, b0);
}

// ------------------------------------------------------------------------------------------------------
// conv_igemm_bf16x3: the split-bf16 kernel with the WEIGHT operand kept out of LDS.
//
// At bf16 MFMA rates the 128x128 tile is LDS-bound (staging writes + fragment reads of both operands use ~85 % of
// the LDS), so the weights are stored in HBM in MFMA-fragment order,
//     wt[tap][chunk][n-tile of 32][kg][hi|lo][lane 0..63][8 bf16]          (1 KiB per wave-instruction)
// and every wave loads its own B fragments straight into VGPRs with coalesced global_load_dwordx4, one K-step
// ahead (two named register sets, the loop is unrolled by two).  Only the activation tile goes through LDS
// (global -> VGPR -> LDS, double-buffered, one barrier per K-step, as in the fp32 kernel).
// ------------------------------------------------------------------------------------------------------
template <int WM, int WN, int MT, int NT, int EPI>
__global__ void __launch_bounds__(WM * WN * 64) __attribute__((amdgpu_waves_per_eu(2, 2)))
conv_igemm_bf16x3(const ConvParams p, const TileGeom g) {
    MSR_SATURATING_CONVERSIONS();
    static_assert(NT == 2, "B register sets are written for two n-tiles per wave");
    constexpr int NTHR = WM * WN * 64;
    constexpr int BM = WM * MT * 32;
    constexpr int BN = WN * NT * 32;
    constexpr int BKC = 32, BKP = 36, SEGS = 8;
    constexpr int A_ITEMS = BM * SEGS / NTHR;
    static_assert(A_ITEMS == 4, "staging is written for 4 16-byte items per thread");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const As = smem;                      // [2][BM][BKP]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int half = lane >> 5, l31 = lane & 31;

    const int bid_all = xcd_remap(blockIdx.x, gridDim.x);
    const int ks = bid_all / g.tiles_mn;
    const int bid = bid_all - ks * g.tiles_mn;
    const int tn = bid % g.tiles_n;
    int tmi = bid / g.tiles_n;
    const int tx0 = (tmi % g.tiles_x) << g.tw_l;
    tmi /= g.tiles_x;
    const int ty0 = (tmi % g.tiles_y) << g.th_l;
    const int b0 = (tmi / g.tiles_y) * g.tb;
    const int n0 = tn * BN;
    const int twm = (1 << g.tw_l) - 1, thm = (1 << g.th_l) - 1;

    int a_goff[A_ITEMS], a_loff[A_ITEMS];
#pragma unroll
    for (int q = 0; q < A_ITEMS; ++q) {
        const int idx = tid + q * NTHR;
        const int row = idx / SEGS, seg = idx % SEGS;
        const int tx = row & twm, ty = (row >> g.tw_l) & thm, tbi = row >> (g.tw_l + g.th_l);
        int b = b0 + tbi;
        b = b < p.B ? b : p.B - 1;
        a_goff[q] = b * p.in_pb + (ty0 + ty) * p.stride * p.in_py + (tx0 + tx) * p.stride * p.Cin + seg * 4;
        a_loff[q] = row * BKP + seg * 4;
    }
    int a_frag[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) a_frag[m] = ((wm * MT + m) * 32 + l31) * BKP + 4 * half;

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
    const int nt32 = p.N / 32;
    // fragment-order weights: 1024 floats per (tap, chunk, n-tile): [kg][hi|lo][lane][4 floats]
    const size_t w_chunk_stride = (size_t)nt32 * 1024;           // next channel chunk, same tap
    const size_t w_tap_stride = (size_t)chunks * w_chunk_stride;  // next tap, same chunk

    const int t_begin = (int)((long)ks * steps / p.ksplit), t_end = (int)((long)(ks + 1) * steps / p.ksplit);
    const int cc0 = t_begin / taps, tap0 = t_begin - cc0 * taps;
    int it_kh = tap0 / p.KW, it_kw = tap0 - it_kh * p.KW;
    const float* a_src = p.in + (it_kh * p.in_py + it_kw * p.Cin + cc0 * BKC);
    const float* b_src = p.wt + (size_t)tap0 * w_tap_stride + (size_t)cc0 * w_chunk_stride +
                         (size_t)(n0 / 32 + wn * NT) * 1024 + lane * 4;

    float4 ra0, ra1, ra2, ra3;
    // B fragments of one K-step: [n-tile 0/1][kg 0/1][hi/lo]; two sets P (even steps) and Q (odd steps)
    bf16x8 P00h, P00l, P01h, P01l, P10h, P10l, P11h, P11l;
    bf16x8 Q00h, Q00l, Q01h, Q01l, Q10h, Q10l, Q11h, Q11l;

#define MSR_LDB(ptr, off) (*reinterpret_cast<const bf16x8*>((ptr) + (off)))
#define MSR_LOAD_B(S)                                                                            \
    {                                                                                            \
        S##00h = MSR_LDB(b_src, 0);        S##00l = MSR_LDB(b_src, 256);                         \
        S##01h = MSR_LDB(b_src, 512);      S##01l = MSR_LDB(b_src, 768);                         \
        S##10h = MSR_LDB(b_src, 1024);     S##10l = MSR_LDB(b_src, 1280);                        \
        S##11h = MSR_LDB(b_src, 1536);     S##11l = MSR_LDB(b_src, 1792);                        \
    }
#define MSR_LOAD_A()                                                                             \
    {                                                                                            \
        ra0 = *reinterpret_cast<const float4*>(a_src + a_goff[0]);                               \
        ra1 = *reinterpret_cast<const float4*>(a_src + a_goff[1]);                               \
        ra2 = *reinterpret_cast<const float4*>(a_src + a_goff[2]);                               \
        ra3 = *reinterpret_cast<const float4*>(a_src + a_goff[3]);                               \
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
                b_src += w_chunk_stride - (size_t)taps * w_tap_stride;                           \
            }                                                                                    \
        }                                                                                        \
    }
#define MSR_WRITE_A(buf)                                                                         \
    {                                                                                            \
        float* a_ = As + (buf) * BM * BKP;                                                       \
        *reinterpret_cast<float4*>(a_ + a_loff[0]) = ra0;                                        \
        *reinterpret_cast<float4*>(a_ + a_loff[1]) = ra1;                                        \
        *reinterpret_cast<float4*>(a_ + a_loff[2]) = ra2;                                        \
        *reinterpret_cast<float4*>(a_ + a_loff[3]) = ra3;                                        \
    }
#define MSR_MMA3(m, n, AH, AL, BH, BL)                                                           \
    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AL, BH, acc[m][n], 0, 0, 0);             \
    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH, BL, acc[m][n], 0, 0, 0);             \
    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH, BH, acc[m][n], 0, 0, 0);
#define MSR_COMPUTE(buf, S)                                                                      \
    {                                                                                            \
        const float* a_ = As + (buf) * BM * BKP;                                                 \
        bf16x8 ah[MT], al[MT];                                                                   \
        _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                         \
            ah[m] = *reinterpret_cast<const bf16x8*>(a_ + a_frag[m]);                            \
            al[m] = *reinterpret_cast<const bf16x8*>(a_ + a_frag[m] + 16);                       \
        }                                                                                        \
        _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                         \
            MSR_MMA3(m, 0, ah[m], al[m], S##00h, S##00l)                                         \
            MSR_MMA3(m, 1, ah[m], al[m], S##10h, S##10l)                                         \
        }                                                                                        \
        _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                         \
            ah[m] = *reinterpret_cast<const bf16x8*>(a_ + a_frag[m] + 8);                        \
            al[m] = *reinterpret_cast<const bf16x8*>(a_ + a_frag[m] + 8 + 16);                   \
        }                                                                                        \
        _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                         \
            MSR_MMA3(m, 0, ah[m], al[m], S##01h, S##01l)                                         \
            MSR_MMA3(m, 1, ah[m], al[m], S##11h, S##11l)                                         \
        }                                                                                        \
    }
#define MSR_STEP(CUR, NXT)                                                                       \
    {                                                                                            \
        MSR_ADVANCE();                                                                           \
        MSR_LOAD_A();                                                                            \
        MSR_LOAD_B(NXT);                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        MSR_COMPUTE(cur, CUR);                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        MSR_WRITE_A(cur ^ 1);                                                                    \
        __syncthreads();                                                                         \
        cur ^= 1;                                                                                \
    }

    MSR_LOAD_A();
    MSR_LOAD_B(P);
    MSR_WRITE_A(0);
    __syncthreads();
    int cur = 0;
    const int nsteps = t_end - t_begin;
    int i = 0;
    for (; i + 2 <= nsteps - 1; i += 2) {
        MSR_STEP(P, Q);
        MSR_STEP(Q, P);
    }
    if ((nsteps - 1) & 1) {
        MSR_STEP(P, Q);
        MSR_COMPUTE(cur, Q);
    } else {
        MSR_COMPUTE(cur, P);
    }
#undef MSR_LDB
#undef MSR_LOAD_B
#undef MSR_LOAD_A
#undef MSR_ADVANCE
#undef MSR_WRITE_A
#undef MSR_MMA3
#undef MSR_COMPUTE
#undef MSR_STEP

    conv_epilogue<WM, WN, MT, NT, EPI>(p, g, acc, ks, wm, wn, half, l31, n0, tx0, ty0, b0);
}

// ------------------------------------------------------------------------------------------------------
// conv_igemm_bf16x3_halo: split-bf16, 3x3 stride 1, LDS-staged INPUT HALO tile.
//
// The generic kernel re-stages the 128-pixel activation tile for each of the 9 taps.  Here the workgroup's
// 8 x 16 pixel tile is staged once per 32-channel chunk together with its one-pixel halo ((8+2) x (16+2) = 180
// pixels x 128 bytes) and the nine taps read it at nine constant LDS offsets: activation traffic (global -> LDS
// and LDS writes) drops ~9x; the weight tile (128 channels x 128 bytes per K-step) is double-buffered as before.
// K order: chunk outer, tap inner (unrolled).  At the chunk seam: barrier, halo write, barrier.
// LDS: 180*144 + 2*128*144 = 62.8 KB -> 2 workgroups per CU.
// ------------------------------------------------------------------------------------------------------
template <int EPI, int SH>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
conv_igemm_bf16x3_halo(const ConvParams p, const TileGeom g) {
    MSR_SATURATING_CONVERSIONS();
    constexpr int WM = 2, WN = 2, MT = 2, NT = 2;
    constexpr int NTHR = 256, BM = 128, BN = 128, BKC = 32, BKP = SH ? 40 : 36;
    constexpr int TH = 8, TW = 16, HH = TH + 2, HW = TW + 2, HP = HH * HW;   // 180 halo pixels
    constexpr int H_ITEMS = (HP * 8 + NTHR - 1) / NTHR;                        // 6 16-byte items per thread
    static_assert(H_ITEMS == 6, "halo staging is written for 6 items per thread");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const Ah = smem;                       // [HP][BKP]
    float* const Bs = smem + HP * BKP;            // [2][BN][BKP]

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
    const int b0 = tmi / g.tiles_y;               // tb == 1
    const int n0 = tn * BN;

    // halo staging items (items past the end duplicate the last one: same bytes to the same LDS slot)
    int h_goff[H_ITEMS], h_loff[H_ITEMS];
#pragma unroll
    for (int q = 0; q < H_ITEMS; ++q) {
        int idx = tid + q * NTHR;
        idx = idx < HP * 8 ? idx : HP * 8 - 1;
        const int hp = idx >> 3, seg = idx & 7;
        const int hy = hp / HW, hx = hp - hy * HW;
        h_goff[q] = b0 * p.in_pb + (ty0 + hy) * p.in_py + (tx0 + hx) * p.Cin + seg * 4;
        h_loff[q] = hp * BKP + seg * 4;
    }
    int b_goff[4], b_loff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int idx = tid + q * NTHR;
        const int row = idx >> 3, seg = idx & 7;
        b_goff[q] = (n0 + row) * p.Cin + seg * 4;
        b_loff[q] = row * BKP + seg * 4;
    }
    // SH == 1: v_mfma_f32_16x16x32_bf16, lane (i = lane & 15, g = lane >> 4) holds row i, k = 8g + {0..7}
    int a_frag16[4], b_frag16[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a_frag16[i] = ((wm * 4 + i) * HW + (lane & 15)) * BKP + 4 * (lane >> 4);
        b_frag16[i] = ((wn * 4 + i) * 16 + (lane & 15)) * BKP + 4 * (lane >> 4);
    }
    f32x4 acc16[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc16[i][j][r] = 0.f;
    int a_frag[MT], b_frag[NT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int row = (wm * MT + m) * 32 + l31;          // pixel (row >> 4, row & 15) of the 8 x 16 tile
        a_frag[m] = ((row >> 4) * HW + (row & 15)) * BKP + 4 * half;
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) b_frag[n] = ((wn * NT + n) * 32 + l31) * BKP + 4 * half;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    const int chunks = p.Cin / BKC;               // even (Cin % 64 == 0): the loop body is a PAIR of chunks
    // Buffer loads: a wave-uniform descriptor + a constant per-lane byte offset (VGPR) + a scalar byte offset that
    // carries the K-step; no 64-bit per-lane address arithmetic in the unrolled loop.
    const unsigned w_tap_bytes = (unsigned)((size_t)p.N * p.Cin * sizeof(float));
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.in), 0, (int)((size_t)p.B * p.in_pb * sizeof(float)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wt = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.wt), 0, (int)(9u * w_tap_bytes), 0x00020000);
    unsigned h_pair = 0;                          // byte offset of the current chunk pair in a pixel
    unsigned w_pair = 0;                          // byte offset of the current chunk pair in a weight row
#pragma unroll
    for (int q = 0; q < H_ITEMS; ++q) h_goff[q] *= 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) b_goff[q] *= 4;

    float4 rh0, rh1, rh2, rh3, rh4, rh5;
    float4 re0, re1, re2, re3;                    // weights of even K-steps in flight
    float4 ro0, ro1, ro2, ro3;                    // weights of odd K-steps in flight
#define MSR_BUFLD(rs, voff, soff) \
    __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (int)(soff), 0))
#define MSR_LOAD_H(soff)                                                                         \
    {                                                                                            \
        rh0 = MSR_BUFLD(rs_in, h_goff[0], soff);                                                 \
        rh1 = MSR_BUFLD(rs_in, h_goff[1], soff);                                                 \
        rh2 = MSR_BUFLD(rs_in, h_goff[2], soff);                                                 \
        rh3 = MSR_BUFLD(rs_in, h_goff[3], soff);                                                 \
        rh4 = MSR_BUFLD(rs_in, h_goff[4], soff);                                                 \
        rh5 = MSR_BUFLD(rs_in, h_goff[5], soff);                                                 \
    }
#define MSR_WRITE_H()                                                                            \
    {                                                                                            \
        *reinterpret_cast<float4*>(Ah + h_loff[0]) = rh0;                                        \
        *reinterpret_cast<float4*>(Ah + h_loff[1]) = rh1;                                        \
        *reinterpret_cast<float4*>(Ah + h_loff[2]) = rh2;                                        \
        *reinterpret_cast<float4*>(Ah + h_loff[3]) = rh3;                                        \
        *reinterpret_cast<float4*>(Ah + h_loff[4]) = rh4;                                        \
        *reinterpret_cast<float4*>(Ah + h_loff[5]) = rh5;                                        \
    }
// weights of K-step U of the current pair (U = 18, 19 are the first two steps of the next pair)
#define MSR_WPTR(U) (w_pair + ((U) / 9) * (BKC * 4) + (unsigned)((U) % 9) * w_tap_bytes)
#define MSR_LOAD_B(R, soff)                                                                      \
    {                                                                                            \
        R##0 = MSR_BUFLD(rs_wt, b_goff[0], soff);                                                \
        R##1 = MSR_BUFLD(rs_wt, b_goff[1], soff);                                                \
        R##2 = MSR_BUFLD(rs_wt, b_goff[2], soff);                                                \
        R##3 = MSR_BUFLD(rs_wt, b_goff[3], soff);                                                \
    }
#define MSR_WRITE_B(buf, R)                                                                      \
    {                                                                                            \
        float* b_ = Bs + (buf) * BN * BKP;                                                       \
        *reinterpret_cast<float4*>(b_ + b_loff[0]) = R##0;                                       \
        *reinterpret_cast<float4*>(b_ + b_loff[1]) = R##1;                                       \
        *reinterpret_cast<float4*>(b_ + b_loff[2]) = R##2;                                       \
        *reinterpret_cast<float4*>(b_ + b_loff[3]) = R##3;                                       \
    }
#define MSR_COMPUTE(buf, TAP)                                                                    \
    {                                                                                            \
        const float* a_ = Ah + (((TAP) / 3) * HW + ((TAP) % 3)) * BKP;                           \
        const float* b_ = Bs + (buf) * BN * BKP;                                                 \
        if constexpr (SH == 1) {                                                                 \
            bf16x8 ah[4], al[4];                                                                 \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                      \
                ah[i] = *reinterpret_cast<const bf16x8*>(a_ + a_frag16[i]);                      \
                al[i] = *reinterpret_cast<const bf16x8*>(a_ + a_frag16[i] + 16);                 \
            }                                                                                    \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                      \
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(b_ + b_frag16[j]);            \
                const bf16x8 bl = *reinterpret_cast<const bf16x8*>(b_ + b_frag16[j] + 16);       \
                _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                  \
                    /* weights as the row operand: D[channel][pixel], see halo16_epilogue_body */ \
                    acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, al[i], acc16[i][j], 0, 0, 0); \
                    acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, ah[i], acc16[i][j], 0, 0, 0); \
                    acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ah[i], acc16[i][j], 0, 0, 0); \
                }                                                                                \
            }                                                                                    \
        } else                                                                                   \
        _Pragma("unroll") for (int kg = 0; kg < 2; ++kg) {                                       \
            bf16x8 ah[MT], al[MT], bh[NT], bl[NT];                                               \
            _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                     \
                ah[m] = *reinterpret_cast<const bf16x8*>(a_ + a_frag[m] + kg * 8);               \
                al[m] = *reinterpret_cast<const bf16x8*>(a_ + a_frag[m] + kg * 8 + 16);          \
            }                                                                                    \
            _Pragma("unroll") for (int n = 0; n < NT; ++n) {                                     \
                bh[n] = *reinterpret_cast<const bf16x8*>(b_ + b_frag[n] + kg * 8);               \
                bl[n] = *reinterpret_cast<const bf16x8*>(b_ + b_frag[n] + kg * 8 + 16);          \
            }                                                                                    \
            _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                     \
                _Pragma("unroll") for (int n = 0; n < NT; ++n) {                                 \
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[m], bh[n], acc[m][n], 0, 0, 0); \
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bl[n], acc[m][n], 0, 0, 0); \
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bh[n], acc[m][n], 0, 0, 0); \
                }                                                                                \
            }                                                                                    \
        }                                                                                        \
    }
// K-step T (0..17, compile time) of a chunk pair.  Weights of step T+2 are requested into LD (the set that step
// T's weights have just left), the MFMAs of step T run from LDS buffer T & 1, then the weights of step T+1 (set
// WR, requested one step ago) go to the other buffer.  The halo of the next chunk is requested on tap 7 and
// replaces the old one after tap 8.  LASTP (compile time) drops everything that would reach past the last pair,
// so no load or LDS write sits under a run-time condition (hipcc would wait vmcnt(0) around those).
#define MSR_STEP(T, LD, WR, LASTP)                                                               \
    {                                                                                            \
        if (!(LASTP) || (T) + 2 < 18) MSR_LOAD_B(LD, MSR_WPTR((T) + 2));                         \
        if ((T) == 7) MSR_LOAD_H(h_pair + BKC * 4);                                              \
        if ((T) == 16 && !(LASTP)) MSR_LOAD_H(h_pair + 2 * BKC * 4);                             \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        MSR_COMPUTE((T) & 1, (T) % 9);                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        if (!(LASTP) || (T) + 1 < 18) MSR_WRITE_B(((T) & 1) ^ 1, WR);                            \
        if ((T) == 8 || ((T) == 17 && !(LASTP))) {                                               \
            __syncthreads();            /* every wave is done with the old halo */               \
            MSR_WRITE_H();                                                                       \
        }                                                                                        \
        if (!(LASTP) || (T) + 1 < 18) __syncthreads();                                           \
    }
#define MSR_PAIR(LASTP)                                                                          \
    MSR_STEP(0, re, ro, LASTP) MSR_STEP(1, ro, re, LASTP) MSR_STEP(2, re, ro, LASTP)             \
    MSR_STEP(3, ro, re, LASTP) MSR_STEP(4, re, ro, LASTP) MSR_STEP(5, ro, re, LASTP)             \
    MSR_STEP(6, re, ro, LASTP) MSR_STEP(7, ro, re, LASTP) MSR_STEP(8, re, ro, LASTP)             \
    MSR_STEP(9, ro, re, LASTP) MSR_STEP(10, re, ro, LASTP) MSR_STEP(11, ro, re, LASTP)           \
    MSR_STEP(12, re, ro, LASTP) MSR_STEP(13, ro, re, LASTP) MSR_STEP(14, re, ro, LASTP)          \
    MSR_STEP(15, ro, re, LASTP) MSR_STEP(16, re, ro, LASTP) MSR_STEP(17, ro, re, LASTP)

    // prologue: halo of chunk 0 and the weights of step 0 into LDS, the weights of step 1 stay in flight
    MSR_LOAD_H(h_pair);
    MSR_LOAD_B(re, MSR_WPTR(0));
    MSR_LOAD_B(ro, MSR_WPTR(1));
    MSR_WRITE_H();
    MSR_WRITE_B(0, re);
    __syncthreads();
    for (int pr = 0; pr < chunks / 2 - 1; ++pr) {
        MSR_PAIR(false)
        h_pair += 2 * BKC * 4;
        w_pair += 2 * BKC * 4;
    }
    MSR_PAIR(true)
#undef MSR_BUFLD
#undef MSR_LOAD_H
#undef MSR_WRITE_H
#undef MSR_WPTR
#undef MSR_LOAD_B
#undef MSR_WRITE_B
#undef MSR_COMPUTE
#undef MSR_STEP
#undef MSR_PAIR

    if constexpr (SH == 1) {
        float4 xin[4][2], cv[8];
        halo16_epilogue_load<EPI>(p, xin, cv, wm, wn, lane, n0, tx0, ty0, b0);
        halo16_epilogue<EPI>(p, g, acc16, wm, wn, lane, n0, tx0, ty0, b0, xin, cv);
    }
    else conv_epilogue<WM, WN, MT, NT, EPI>(p, g, acc, 0, wm, wn, half, l31, n0, tx0, ty0, b0);
}

// ------------------------------------------------------------------------------------------------------
// conv_igemm_bf16x3_pp: the halo kernel as a PING-PONG pair of wave groups (512 threads, one workgroup per CU).
//
// With two independent 256-thread workgroups per CU the two waves of a SIMD drift into lockstep: both issue
// their MFMAs together (the matrix pipe is per SIMD, so that is no faster than one wave) and both then sit in
// their LDS phase together, which left the pipe ~60 % busy.  Here the two waves of a SIMD belong to ONE workgroup
// and are held half a K-step apart by the barrier schedule:
//     phase 2t   : group X (waves 0-3)  R(t)  LDS fragment reads, weight staging      | group Y (waves 4-7)  M(t-1)
//     phase 2t+1 : group X              M(t)  48 x v_mfma_f32_16x16x32_bf16, registers | group Y              R(t)
// so every SIMD always has one wave in its matrix segment and one in its memory segment (MI355X_MICROARCH.md,
// "Two waves per SIMD", items 5 and 9).  X owns the top 8 rows of a 16 x 16 pixel tile, Y the bottom 8; both use
// the same 128-channel weight tile.  LDS: input halo (18 x 18 pixels x 32 channels) double-buffered by chunk,
// weight tile double-buffered by K-step, 160-byte rows: 2 * 324 * 160 + 2 * 128 * 160 = 144,640 B.
// Barrier count (s_barrier only counts arrivals; every wave must execute the same number): per K-step each wave
// runs two (after R, after M); group Y runs one extra before its first R and skips the one after its last M of the
// workgroup's last tile: X = 1 + 2 * steps, Y = 1 + 1 + 2 * steps - 1 — equal.  (MSR_WG_BARRIER, top of the file.)  On a
// tile's last step Y executes its second barrier AFTER its epilogue (X before): same count, and the two epilogues overlap.
// Hazards (b = barrier at the end of a phase): the weights of step t+1 go to Bs[(t+1)&1] during R(t) of both
// groups (phases 2t, 2t+1); that buffer was last read in R(t-1) (phases 2t-2, 2t-1) and is next read in R(t+1)
// (phases 2t+2, 2t+3).  The halo of chunk c+1 goes to Ah[(c+1)&1] on tap 7 of chunk c.
// ------------------------------------------------------------------------------------------------------
// F16X2 = true is the opt-in 2-term form for the gamma|beta convs (kernels.h PREC_F16X2): operands are split-fp16
// words, the weight's lo half is neither read from LDS nor multiplied: 32 MFMAs and 12 ds_read_b128 per K-step
// instead of 48 and 16.
// MODE 2 (PP_FP8) is the declared non-parity fp8 form (kernels.h PREC_FP8): a chunk row holds 128 one-byte channels,
// a K-step is 128 channels of one tap: 16 block-scaled MFMAs (K = 128 each), same staging and fragment reads.
// MODE 3 (PP_F16C, kernels.h PREC_F16C): fp16 main term + fp8 cross terms.  A chunk row holds [32 x hi f16 | 32 x h8 |
// 32 x l8] (weights: l8 then h8, so that byte t of one pairs with byte t of the other: w_lo*x_hi, w_hi*x_lo).  Every
// K-step runs the 16 f16 MFMAs of its tap (x_hi * w_hi); the lane's 16 bytes at +64 + 16 * (lane >> 4) (the same
// conflict-free read as the bf16 lo half) of an EVEN step and of the following ODD step make one 32-byte operand, and the
// odd step adds 16 block-scaled K = 128 fp8 MFMAs that cover the cross terms of both taps.  In that instruction a lane's
// first 16 bytes are k = 16g.. of the first 64 and its second 16 bytes of the second 64, and k-block b takes its e8m0
// scale from lane group b: blocks 0 / 2 are the even / odd tap's h8 (w: l8) bytes, blocks 1 / 3 their l8 (w: h8) bytes,
// so lane groups 0, 2 carry the scale of the first kind and 1, 3 of the second.  Two MFMA-equivalents per product
// instead of three; per-product error ~2^-15 (the fp8 rounding of a term that is 2^-11 of the product).
enum PpMode : int { PP_BF16X3 = 0, PP_F16X2 = 1, PP_FP8 = 2, PP_F16C = 3 };
// ONE = true: the input has ONE 32-slot chunk (the Cin = 128 convs of the fp8 mode: 128 one-byte channels).  The
// unrolled body of 18 K-steps then covers TWO work items (tiles) of 9 taps each instead of a chunk pair of one tile:
// item B takes the place of "chunk 1" (its halo is staged during A's taps into the other halo buffer, its weights follow
// A's in the weight ring), item A' of the next body the place of "the next tile"; A's epilogue runs between steps 8 and 9.
// The LDS schedule is unchanged.  With an odd number of items the last body computes its item twice (same stores).
template <int EPI, int MODE, bool ONE = false>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
conv_igemm_bf16x3_pp(const ConvParams p, const TileGeom g) {
    MSR_SATURATING_CONVERSIONS();
    constexpr bool F16X2 = MODE == PP_F16X2;
    constexpr int NTHR = 512, BN = 128, BKC = 32, BKP = 40;
    constexpr int TH = 16, TW = 16, HW = TW + 2, HP = (TH + 2) * HW;          // 324 halo pixels
    constexpr int H_ITEMS = (HP * 8 + NTHR - 1) / NTHR;                        // 6 16-byte items per thread
    static_assert(H_ITEMS == 6, "halo staging is written for 6 items per thread");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const Ah = smem;                       // [2][HP][BKP]
    float* const Bs = smem + 2 * HP * BKP;        // [2][BN][BKP]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // PP_F16C + EPI_SPADE: a private 16 x 36-dword line image per wave behind the tile buffers (epilogue store assembly)
#ifdef MSR_PP_STAMPS
    unsigned* const stage = nullptr;              // the stamp words live there in the diagnostic build
#else
    unsigned* const stage = (MODE == PP_F16C && EPI == EPI_SPADE)
        ? reinterpret_cast<unsigned*>(smem + (2 * HP + 2 * BN) * BKP) + wave * (16 * 36) : nullptr;
#endif
    const int grp = wave >> 2;                    // 0 = X, 1 = Y (wave-uniform, scalar)
    const int wm = (wave >> 1) & 1, wn = wave & 1;

    // Persistent: gridDim.x (a multiple of 8, one workgroup per CU) workgroups walk all tiles.  Workgroups are
    // dealt round-robin over the 8 XCDs, so XCD x owns a contiguous range of logical tiles (as xcd_remap) and its
    // gridDim.x / 8 workgroups take consecutive tiles of that range in every round: the tiles in flight on an XCD
    // share their halo (same pixels, next channel block) and weights in that XCD's L2.
    // With p.ksplit > 1 (fewer tiles than CUs) a work item is (K range, tile): range ks covers chunk pairs
    // [ks * ppi, (ks + 1) * ppi), items are numbered range-major so that neighbours still share their halo.
    const int ksn = p.ksplit > 1 ? p.ksplit : 1;
    const int ppi = ONE ? 1 : (p.Cin / (2 * BKC)) / ksn;                       // chunk pairs per item (ONE: one body = 2 items)
    const unsigned kbytes = (unsigned)ppi * 2u * BKC * 4u;                      // byte offset of one range (input and weights)
    const int items = g.tiles_mn * ksn;
    const int slots = gridDim.x >> 3, xcd = blockIdx.x & 7;
    const int tq = items >> 3, tr = items & 7;
    const int cnt = tq + (xcd < tr ? 1 : 0);
    const int base = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
    int tile = blockIdx.x >> 3;                   // index inside the XCD's range
    if (tile >= cnt) return;

    // per-tile state is scalar: tile origin (pixels, channel block) and its byte offsets in the input / weights
    int n0, tx0, ty0, b0, ks0;
    unsigned h_tile, w_tile;
#define MSR_DECODE(T_, N0_, TX_, TY_, B_, HT_, WT_, KS_)                                          \
    {                                                                                            \
        KS_ = (T_) / g.tiles_mn;                                                                 \
        const int t_ = (T_) - KS_ * g.tiles_mn;                                                  \
        int tn_, tmi_;                                                                           \
        MSR_WALK(g, t_, tn_, tmi_)                                                               \
        TX_ = (tmi_ % g.tiles_x) << 4;                                                           \
        tmi_ /= g.tiles_x;                                                                       \
        TY_ = (tmi_ % g.tiles_y) << 4;                                                           \
        B_ = tmi_ / g.tiles_y;                                                                   \
        N0_ = tn_ * BN;                                                                          \
        HT_ = (unsigned)((B_) * p.in_pb + (TY_) * p.in_py + (TX_) * p.Cin) * 4u + (unsigned)KS_ * kbytes; \
        WT_ = (unsigned)((N0_) * p.Cin) * 4u + (unsigned)KS_ * kbytes;                           \
    }
    MSR_DECODE(base + tile, n0, tx0, ty0, b0, h_tile, w_tile, ks0)

    int h_goff[H_ITEMS], h_loff[H_ITEMS];         // tile-relative byte offsets / LDS float offsets
#pragma unroll
    for (int q = 0; q < H_ITEMS; ++q) {
        int idx = tid + q * NTHR;
        idx = idx < HP * 8 ? idx : HP * 8 - 1;    // items past the end duplicate the last one
        const int hp = idx >> 3, seg = idx & 7;
        const int hy = hp / HW, hx = hp - hy * HW;
        h_goff[q] = (hy * p.in_py + hx * p.Cin + seg * 4) * 4;
        h_loff[q] = hp * BKP + seg * 4;
    }
    int b_goff[2], b_loff[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int idx = tid + q * NTHR;
        const int row = idx >> 3, seg = idx & 7;
        b_goff[q] = (row * p.Cin + seg * 4) * 4;
        b_loff[q] = row * BKP + seg * 4;
    }
    int a_frag[4], b_frag[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a_frag[i] = ((grp * 8 + wm * 4 + i) * HW + (lane & 15)) * BKP + 4 * (lane >> 4);
        b_frag[i] = ((wn * 4 + i) * 16 + (lane & 15)) * BKP + 4 * (lane >> 4);
    }
    f32x4 acc[4][4];
    int wsc[4] = {0x7F7F7F7F, 0x7F7F7F7F, 0x7F7F7F7F, 0x7F7F7F7F};   // PP_FP8: e8m0 weight scales of the wave's 4 x 16 rows

    const unsigned w_tap_bytes = (unsigned)((size_t)p.N * p.Cin * sizeof(float));
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.in), 0, (int)((size_t)p.B * p.in_pb * sizeof(float)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wt = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.wt), 0, (int)(9u * w_tap_bytes), 0x00020000);
    unsigned h_pair = 0, w_pair = 0;              // byte offsets of the current chunk pair
    unsigned h_next = 0, w_next = 0;              // byte offsets of the NEXT tile (of this one again on the last)
    unsigned h_b = 0, w_b = 0;                    // ONE: the body's second item
    int n0b = 0, tx0b = 0, ty0b = 0, b0b = 0, ksb = 0;

    float4 rh0, rh1, rh2, rh3, rh4, rh5;          // halo of the next chunk in flight
    float4 rw0, rw1;                              // weights of the next K-step in flight
    bf16x8 ah[4], al[4], bh[4], bl[4];            // fragments of the current K-step
    i32x8 qa0, qa1, qa2, qa3, qb0, qb1, qb2, qb3; // ... PP_FP8: the same 32 bytes per lane as ONE 8-register operand
    i32x4 ca0[4], ca1[4], cb0[4], cb1[4];         // ... PP_F16C: the cross-term pieces of an even step and of the odd one after it
    const int asc = ((lane >> 4) & 1) ? 0x74747474 : 0x7F7F7F7F;   // PP_F16C: e8m0 of the activation piece: l8 = x_lo * 2^11 (116), h8 = x_hi (127)
    float4 xpre[4][2], cpre[8];                   // the epilogue's memory operands, requested on step 16 of the last pair
#define MSR_BUFLD(rs, voff, soff) \
    __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (int)(soff), 0))
#define MSR_LOAD_H_LO(soff)                                                                      \
    {                                                                                            \
        rh0 = MSR_BUFLD(rs_in, h_goff[0], soff); rh1 = MSR_BUFLD(rs_in, h_goff[1], soff);        \
        rh2 = MSR_BUFLD(rs_in, h_goff[2], soff);                                                 \
    }
#define MSR_LOAD_H_HI(soff)                                                                      \
    {                                                                                            \
        rh3 = MSR_BUFLD(rs_in, h_goff[3], soff); rh4 = MSR_BUFLD(rs_in, h_goff[4], soff);        \
        rh5 = MSR_BUFLD(rs_in, h_goff[5], soff);                                                 \
    }
#define MSR_LOAD_H(soff) { MSR_LOAD_H_LO(soff) MSR_LOAD_H_HI(soff) }
#define MSR_WRITE_H_LO(buf)                                                                      \
    {                                                                                            \
        float* h_ = Ah + (buf) * HP * BKP;                                                       \
        *reinterpret_cast<float4*>(h_ + h_loff[0]) = rh0; *reinterpret_cast<float4*>(h_ + h_loff[1]) = rh1; \
        *reinterpret_cast<float4*>(h_ + h_loff[2]) = rh2;                                        \
    }
#define MSR_WRITE_H_HI(buf)                                                                      \
    {                                                                                            \
        float* h_ = Ah + (buf) * HP * BKP;                                                       \
        *reinterpret_cast<float4*>(h_ + h_loff[3]) = rh3; *reinterpret_cast<float4*>(h_ + h_loff[4]) = rh4; \
        *reinterpret_cast<float4*>(h_ + h_loff[5]) = rh5;                                        \
    }
#define MSR_WRITE_H(buf) { MSR_WRITE_H_LO(buf) MSR_WRITE_H_HI(buf) }
// weights of K-step U of a chunk pair, relative to the pair's first chunk
#define MSR_WOFF(U) (((U) / 9) * (BKC * 4) + (unsigned)((U) % 9) * w_tap_bytes)
#define MSR_LOAD_B(soff)                                                                         \
    { rw0 = MSR_BUFLD(rs_wt, b_goff[0], soff); rw1 = MSR_BUFLD(rs_wt, b_goff[1], soff); }
#define MSR_WRITE_B(buf)                                                                         \
    {                                                                                            \
        float* b_ = Bs + (buf) * BN * BKP;                                                       \
        *reinterpret_cast<float4*>(b_ + b_loff[0]) = rw0; *reinterpret_cast<float4*>(b_ + b_loff[1]) = rw1; \
    }
// R(T): memory segment of K-step T (0..17 within the pair, compile time).  The staging never stops: on the last
// pair of a tile (LASTP) the steps past its end are the first steps of the NEXT tile (weights of its steps 0 and
// 1, halo of its chunk 0), so no load or LDS write sits under a run-time condition; on the last tile of the
// workgroup "next" is the tile itself and the staged data is simply never read.
#define MSR_R(T, LASTP)                                                                          \
    {                                                                                            \
        MSR_WRITE_B(((T) + 1) & 1);                                                              \
        if constexpr (ONE) {                                                                     \
            if ((T) + 2 >= 18) MSR_LOAD_B(w_next + (unsigned)((T) + 2 - 18) * w_tap_bytes)       \
            else if ((T) + 2 >= 9) MSR_LOAD_B(w_b + (unsigned)((T) + 2 - 9) * w_tap_bytes)       \
            else MSR_LOAD_B(w_tile + (unsigned)((T) + 2) * w_tap_bytes);                         \
            if ((T) == 1) MSR_LOAD_H(h_b)                                                        \
            if ((T) == 10) MSR_LOAD_H(h_next)                                                    \
        } else {                                                                                 \
            if ((LASTP) && (T) + 2 >= 18) MSR_LOAD_B(w_next + MSR_WOFF((T) + 2 - 18))            \
            else MSR_LOAD_B(w_tile + w_pair + MSR_WOFF((T) + 2));                                \
            if ((T) % 9 == 1) {                                                                  \
                if ((LASTP) && (T) >= 9) MSR_LOAD_H(h_next)                                      \
                else MSR_LOAD_H(h_tile + h_pair + ((T) / 9 + 1) * BKC * 4);                      \
            }                                                                                    \
        }                                                                                        \
        /* the halo of the next chunk goes to LDS in two halves (taps 6 and 7): all six stores in one R make that   \
           segment longer than the partner's matrix segment (1060 vs 840 cycles) */              \
        if ((T) % 9 == 6) MSR_WRITE_H_LO((((T) / 9) & 1) ^ 1);                                   \
        if ((T) % 9 == 7) MSR_WRITE_H_HI((((T) / 9) & 1) ^ 1);                                   \
        const float* a_ = Ah + (((T) / 9) & 1) * HP * BKP + ((((T) % 9) / 3) * HW + (((T) % 9) % 3)) * BKP; \
        const float* b_ = Bs + ((T) & 1) * BN * BKP;                                             \
        if constexpr (MODE == PP_FP8) {                                                          \
            /* the K = 128 MFMA takes 8 consecutive registers per operand: both 16-byte halves into one vector */ \
            MSR_RD8(qa0, a_ + a_frag[0]) MSR_RD8(qa1, a_ + a_frag[1]) MSR_RD8(qa2, a_ + a_frag[2]) MSR_RD8(qa3, a_ + a_frag[3]) \
            MSR_RD8(qb0, b_ + b_frag[0]) MSR_RD8(qb1, b_ + b_frag[1]) MSR_RD8(qb2, b_ + b_frag[2]) MSR_RD8(qb3, b_ + b_frag[3]) \
        } else if constexpr (MODE == PP_F16C) {                                                  \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                      \
                ah[i] = *reinterpret_cast<const bf16x8*>(a_ + a_frag[i]);                        \
                bh[i] = *reinterpret_cast<const bf16x8*>(b_ + b_frag[i]);                        \
                if (((T) & 1) == 0) {                                                            \
                    ca0[i] = *reinterpret_cast<const i32x4*>(a_ + a_frag[i] + 16);               \
                    cb0[i] = *reinterpret_cast<const i32x4*>(b_ + b_frag[i] + 16);               \
                } else {                                                                         \
                    ca1[i] = *reinterpret_cast<const i32x4*>(a_ + a_frag[i] + 16);               \
                    cb1[i] = *reinterpret_cast<const i32x4*>(b_ + b_frag[i] + 16);               \
                }                                                                                \
            }                                                                                    \
        } else {                                                                                 \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                      \
                ah[i] = *reinterpret_cast<const bf16x8*>(a_ + a_frag[i]);                        \
                al[i] = *reinterpret_cast<const bf16x8*>(a_ + a_frag[i] + 16);                   \
                bh[i] = *reinterpret_cast<const bf16x8*>(b_ + b_frag[i]);                        \
                if constexpr (!F16X2) bl[i] = *reinterpret_cast<const bf16x8*>(b_ + b_frag[i] + 16); \
            }                                                                                    \
        }                                                                                        \
    }
// M(T): matrix segment, registers only (weights as the row operand: D[channel][pixel])
#define MSR_RD8(dst, ptr)                                                                        \
    {                                                                                            \
        const i32x4 lo_ = *reinterpret_cast<const i32x4*>(ptr);                                  \
        const i32x4 hi_ = *reinterpret_cast<const i32x4*>((ptr) + 16);                           \
        dst = __builtin_shufflevector(lo_, hi_, 0, 1, 2, 3, 4, 5, 6, 7);                         \
    }
#define MSR_MF8(J, WQ)                                                                           \
    acc[0][J] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(WQ, qa0, acc[0][J], 0, 1, 0, wsc[J], 0, 0x7F7F7F7F); \
    acc[1][J] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(WQ, qa1, acc[1][J], 0, 1, 0, wsc[J], 0, 0x7F7F7F7F); \
    acc[2][J] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(WQ, qa2, acc[2][J], 0, 1, 0, wsc[J], 0, 0x7F7F7F7F); \
    acc[3][J] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(WQ, qa3, acc[3][J], 0, 1, 0, wsc[J], 0, 0x7F7F7F7F);
#define MSR_F16(v) __builtin_bit_cast(f16x8, v)
#define MSR_CAT8(lo, hi) __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7)
#define MSR_M(T)                                                                                 \
    if constexpr (MODE == PP_F16C) {                                                             \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                          \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                        \
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(MSR_F16(bh[j]), MSR_F16(ah[i]), acc[i][j], 0, 0, 0); \
        }                                                                                        \
        if (((T) & 1) == 1) {                                                                    \
            /* the cross MFMA of an accumulator 16 instructions after its main one: no dependent-issue stall */ \
            __builtin_amdgcn_sched_barrier(0);                                                   \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                      \
                const i32x8 wq_ = MSR_CAT8(cb0[j], cb1[j]);                                      \
                _Pragma("unroll") for (int i = 0; i < 4; ++i)                                    \
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wq_, MSR_CAT8(ca0[i], ca1[i]), acc[i][j], \
                                                                                 0, 0, 0, wsc[j], 0, asc); \
            }                                                                                    \
        }                                                                                        \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                            \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(acc[i][j]));     \
    } else if constexpr (MODE == PP_FP8) {                                                       \
        /* weights fp8 e4m3 (row operand, per-channel e8m0 scale in wsc[j]) x activations bf8 e5m2 (unit scale) */ \
        MSR_MF8(0, qb0) MSR_MF8(1, qb1) MSR_MF8(2, qb2) MSR_MF8(3, qb3)                          \
        /* pin the accumulators here: without a use in this segment LLVM sinks the whole MFMA chain of the last chunk  \
           pair into the epilogue (per output row) and keeps 18 steps of fragments alive in scratch */ \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                            \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(acc[i][j]));     \
    } else if constexpr (F16X2) {                                                                \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                          \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                        \
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(MSR_F16(bh[j]), MSR_F16(al[i]), acc[i][j], 0, 0, 0); \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                        \
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(MSR_F16(bh[j]), MSR_F16(ah[i]), acc[i][j], 0, 0, 0); \
        }                                                                                        \
    } else {                                                                                     \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                          \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                        \
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], al[i], acc[i][j], 0, 0, 0); \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                        \
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], ah[i], acc[i][j], 0, 0, 0); \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                        \
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], ah[i], acc[i][j], 0, 0, 0); \
        }                                                                                        \
    }
// Diagnostic build only (-DMSR_PP_STAMPS, tools/gpu_pp_stamps.py): s_memtime stamps of waves 0 (X) and 4 (Y) of one
// workgroup around the segments of one chunk pair, kept in the LDS words behind the product's 144,640 bytes.
#ifdef MSR_PP_STAMPS
#define MSR_STAMP()                                                                              \
    if (dbg_on && lane == 0 && (wave & 3) == 0) dbg[(wave >> 2) * 1024 + dbg_n++] = (unsigned)__builtin_amdgcn_s_memtime();
#else
#define MSR_STAMP()
#endif
#define MSR_STEP(T, LASTP)                                                                       \
    {                                                                                            \
        MSR_STAMP()                                                                              \
        MSR_R(T, LASTP)                                                                          \
        if constexpr (ONE) {                                                                     \
            if ((T) == 7 && MODE != PP_FP8) halo16_epilogue_load<EPI>(p, xpre, cpre, wm, wn, lane, n0, tx0, ty0 + grp * 8, b0); \
            if ((T) == 16 && MODE != PP_FP8) halo16_epilogue_load<EPI>(p, xpre, cpre, wm, wn, lane, n0b, tx0b, ty0b + grp * 8, b0b); \
        } else {                                                                                 \
            if ((LASTP) && (T) == 16 && EPI != EPI_PARTIAL && MODE != PP_FP8 && MODE != PP_F16C) halo16_epilogue_load<EPI>(p, xpre, cpre, wm, wn, lane, n0, tx0, ty0 + grp * 8, b0); \
        }                                                                                        \
        MSR_STAMP()                                                                              \
        MSR_WG_BARRIER()                                                                         \
        MSR_STAMP()                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        MSR_M(T)                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        MSR_STAMP()                                                                              \
        /* Y's M on the workgroup's very last step has no partner segment.  On a tile's last step (not ONE) group Y    \
           postpones this barrier until after its epilogue (below the pair), so that both groups' epilogues share ONE \
           barrier interval */                                                                   \
        if (ONE || !((LASTP) && (T) == 17)) MSR_WG_BARRIER()                                     \
        else if (grp == 0) MSR_WG_BARRIER()                                                      \
        if constexpr (ONE) {                                                                     \
            if ((T) == 8) {   /* item A is complete: its epilogue, fresh accumulators, item B's weight scales */ \
                if constexpr (MODE == PP_FP8) halo16_epilogue_load<EPI>(p, xpre, cpre, wm, wn, lane, n0, tx0, ty0 + grp * 8, b0); \
                halo16_epilogue<EPI>(p, ge, acc, wm, wn, lane, n0, tx0, ty0 + grp * 8, b0, xpre, cpre); \
                _Pragma("unroll") for (int i = 0; i < 4; ++i)                                    \
                    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                \
                        _Pragma("unroll") for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;        \
                if constexpr (MODE == PP_FP8) {                                                  \
                    _Pragma("unroll") for (int j = 0; j < 4; ++j) wsc[j] = p.wexp[n0b + wn * 64 + j * 16 + (lane & 15)]; \
                }                                                                                \
            }                                                                                    \
        }                                                                                        \
    }
#define MSR_PAIR(LASTP)                                                                          \
    MSR_STEP(0, LASTP) MSR_STEP(1, LASTP) MSR_STEP(2, LASTP) MSR_STEP(3, LASTP) MSR_STEP(4, LASTP) \
    MSR_STEP(5, LASTP) MSR_STEP(6, LASTP) MSR_STEP(7, LASTP) MSR_STEP(8, LASTP) MSR_STEP(9, LASTP) \
    MSR_STEP(10, LASTP) MSR_STEP(11, LASTP) MSR_STEP(12, LASTP) MSR_STEP(13, LASTP)              \
    MSR_STEP(14, LASTP) MSR_STEP(15, LASTP) MSR_STEP(16, LASTP) MSR_STEP(17, LASTP)

#ifdef MSR_PP_STAMPS
    unsigned* dbg = reinterpret_cast<unsigned*>(smem + (2 * HP + 2 * BN) * BKP);
    int dbg_n = 0;
    bool dbg_on = false;
#endif
    // prologue of the workgroup's first tile: halo of chunk 0 and the weights of step 0 into LDS, the weights of
    // step 1 stay in flight
    MSR_LOAD_H(h_tile);
    MSR_LOAD_B(w_tile + MSR_WOFF(0));
    MSR_WRITE_H(0);
    MSR_WRITE_B(0);
    MSR_LOAD_B(w_tile + MSR_WOFF(1));
    MSR_WG_BARRIER()
    if (grp == 1) MSR_WG_BARRIER()                // Y starts half a step late (phase 0 is X's R(0) alone)
    TileGeom ge = g;                              // the epilogue numbers its moment slabs by 8-row tiles
    ge.th_l = 3;
    ge.tiles_y = g.tiles_y * 2;
#ifdef MSR_PP_STAMPS
    unsigned tstamp[20];
    int tstamp_n = 0;
#endif
    for (;;) {
#ifdef MSR_PP_STAMPS
        if (tstamp_n < 20) tstamp[tstamp_n++] = (unsigned)__builtin_amdgcn_s_memtime();      // coarse: one stamp per tile
#endif
        const int tnext = ONE ? tile + 2 * slots : tile + slots;
        const bool has_next = tnext < cnt;
        int n0n, tx0n, ty0n, b0n, ks0n;
        MSR_DECODE(base + (has_next ? tnext : tile), n0n, tx0n, ty0n, b0n, h_next, w_next, ks0n)
        if constexpr (ONE) {
            const int tb = tile + slots < cnt ? tile + slots : tile;      // no second item left: item A again
            MSR_DECODE(base + tb, n0b, tx0b, ty0b, b0b, h_b, w_b, ksb)
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
        if constexpr (MODE == PP_FP8) {
#pragma unroll
            for (int j = 0; j < 4; ++j) wsc[j] = p.wexp[n0 + wn * 64 + j * 16 + (lane & 15)];
        }
        if constexpr (MODE == PP_F16C) {    // byte 0 = e8m0 of the channel's w_lo pieces (even lane groups), byte 1 = of its w_hi pieces
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int w_ = p.wexp[n0 + wn * 64 + j * 16 + (lane & 15)];
                wsc[j] = ((((lane >> 4) & 1) ? (w_ >> 8) : w_) & 0xFF) * 0x01010101;
            }
        }
        h_pair = 0;
        w_pair = 0;
        for (int pr = 0; pr < ppi - 1; ++pr) {
#ifdef MSR_PP_STAMPS
            dbg_on = blockIdx.x == 8 && pr == 2 && dbg_n == 0;
#endif
            MSR_PAIR(false)
#ifdef MSR_PP_STAMPS
            dbg_on = false;
#endif
            h_pair += 2 * BKC * 4;
            w_pair += 2 * BKC * 4;
        }
        MSR_PAIR(true)
        // Both groups run their epilogue in the SAME barrier interval: X after the barrier that follows its last M (beside
        // its R(0) of the next tile), Y right after its last M, BEFORE that barrier.  (Each group used to run it after the
        // barrier: X's epilogue then faced only Y's last M and Y's only X's first M of the next tile, i.e. the two
        // epilogues — ~10k cycles each with their loads and stores — ran one after the other: a gamma|beta tile took 87k
        // cycles for 58k of K loop, tools/gpu_pp_stamps_gb.py.)  Its stores are not waited for.
        // PP_FP8: the scaled MFMA does not accumulate in place under register pressure, so its epilogue operands are
        // not held across the last K-steps but requested here
        if constexpr (ONE) {       // the body's second item
            if constexpr (MODE == PP_FP8) halo16_epilogue_load<EPI>(p, xpre, cpre, wm, wn, lane, n0b, tx0b, ty0b + grp * 8, b0b);
            halo16_epilogue<EPI>(p, ge, acc, wm, wn, lane, n0b, tx0b, ty0b + grp * 8, b0b, xpre, cpre);
        } else {
            if constexpr (MODE == PP_FP8 || MODE == PP_F16C) halo16_epilogue_load<EPI>(p, xpre, cpre, wm, wn, lane, n0, tx0, ty0 + grp * 8, b0);
            if constexpr (EPI == EPI_PARTIAL) halo16_epilogue_partial(p, acc, wm, wn, lane, n0, tx0, ty0 + grp * 8, b0, ks0);
            else halo16_epilogue<EPI>(p, ge, acc, wm, wn, lane, n0, tx0, ty0 + grp * 8, b0, xpre, cpre, stage);
            if (grp == 1 && has_next) MSR_WG_BARRIER()        // Y's barrier of the tile's last step (see MSR_STEP)
        }
        if (!has_next) break;
        tile = tnext;
        n0 = n0n; tx0 = tx0n; ty0 = ty0n; b0 = b0n; ks0 = ks0n;
        h_tile = h_next;
        w_tile = w_next;
    }
#ifdef MSR_PP_STAMPS
    if (blockIdx.x == 8 && lane == 0 && (wave & 3) == 0) {
        for (int k = 0; k + 1 < tstamp_n; ++k) printf("wave %d tile %2d: %6u cycles\n", wave, k, tstamp[k + 1] - tstamp[k]);
        const unsigned* d = dbg + (wave >> 2) * 1024;
        for (int k = 0; k + 3 < dbg_n; k += 4)
            printf("wave %d step %2d: R %4u  barrier %4u  M %4u  barrier+next %4u cycles\n", wave, k / 4, d[k + 1] - d[k],
                   d[k + 2] - d[k + 1], d[k + 3] - d[k + 2], k + 4 < dbg_n ? d[k + 4] - d[k + 3] : 0u);
    }
#endif
#undef MSR_STAMP
#undef MSR_DECODE
#undef MSR_BUFLD
#undef MSR_LOAD_H
#undef MSR_WRITE_H
#undef MSR_WRITE_H_LO
#undef MSR_WRITE_H_HI
#undef MSR_LOAD_H_LO
#undef MSR_LOAD_H_HI
#undef MSR_WOFF
#undef MSR_LOAD_B
#undef MSR_WRITE_B
#undef MSR_R
#undef MSR_M
#undef MSR_F16
#undef MSR_CAT8
#undef MSR_RD8
#undef MSR_MF8
#undef MSR_STEP
#undef MSR_PAIR
}

// ------------------------------------------------------------------------------------------------------
// splitk_epilogue: sums the ksplit partial accumulators in a fixed order (deterministic) and applies the same
// epilogue the fused kernel would have applied.  One thread per (pixel, 4 channels).
// ------------------------------------------------------------------------------------------------------
template <int EPI>
__global__ void __launch_bounds__(256) splitk_epilogue_kernel(const ConvParams p) {
    MSR_SATURATING_CONVERSIONS();
    const int Cout = EPI == EPI_SPADE ? p.N / 2 : p.N;
    const int quads = Cout / 4;
    const long M = (long)p.B * p.Hout * p.Wout;
    const long total = M * quads;
    const size_t pstride = (size_t)M * p.N;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int q = (int)(i % quads);
        const long pix = i / quads;
        const int x = (int)(pix % p.Wout);
        const int y = (int)((pix / p.Wout) % p.Hout);
        const int b = (int)(pix / ((long)p.Wout * p.Hout));
        const int c = q * 4;
        const int col = EPI == EPI_SPADE ? (c / 32) * 64 + (c % 32) : c;
        const float* pp = p.partial + (size_t)pix * p.N + col;
        float4 a = *reinterpret_cast<const float4*>(pp);
        float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (EPI == EPI_SPADE) bsum = *reinterpret_cast<const float4*>(pp + 32);
#pragma unroll 8      // the K ranges' loads in flight together, the additions in range order
        for (int k = 1; k < p.ksplit; ++k) {
            const float4 t = *reinterpret_cast<const float4*>(pp + k * pstride);
            a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
            if constexpr (EPI == EPI_SPADE) {
                const float4 u = *reinterpret_cast<const float4*>(pp + k * pstride + 32);
                bsum.x += u.x; bsum.y += u.y; bsum.z += u.z; bsum.w += u.w;
            }
        }
        const float4 b0v = *reinterpret_cast<const float4*>(p.bias + col);
        float4 v = make_float4(a.x + b0v.x, a.y + b0v.y, a.z + b0v.z, a.w + b0v.w);
        if constexpr (EPI == EPI_AFFINE) {
            float4 sc = make_float4(1.f, 1.f, 1.f, 1.f);
            if (p.scale) sc = *reinterpret_cast<const float4*>(p.scale + col);
            v = make_float4(a.x * sc.x + b0v.x, a.y * sc.y + b0v.y, a.z * sc.z + b0v.z, a.w * sc.w + b0v.w);
            if (p.act == 1) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            } else if (p.act == 2) {
                v.x = v.x >= 0.f ? v.x : v.x * p.slope; v.y = v.y >= 0.f ? v.y : v.y * p.slope;
                v.z = v.z >= 0.f ? v.z : v.z * p.slope; v.w = v.w >= 0.f ? v.w : v.w * p.slope;
            }
        }
        if constexpr (EPI == EPI_RES || EPI == EPI_SPADE) {
            const float4 xv = *reinterpret_cast<const float4*>(p.aux + (size_t)b * p.aux_pb +
                                                               (size_t)(y >> p.aux_shift) * p.aux_py +
                                                               (size_t)(x >> p.aux_shift) * p.aux_px + c);
            if constexpr (EPI == EPI_RES) {
                v.x += xv.x; v.y += xv.y; v.z += xv.z; v.w += xv.w;
            } else {
                const float4 b1v = *reinterpret_cast<const float4*>(p.bias + col + 32);
                const float4 mu = *reinterpret_cast<const float4*>(p.mean + c);
                const float4 sd = *reinterpret_cast<const float4*>(p.stdv + c);
                v.x = v.x * ((xv.x - mu.x) / sd.x) + (bsum.x + b1v.x);
                v.y = v.y * ((xv.y - mu.y) / sd.y) + (bsum.y + b1v.y);
                v.z = v.z * ((xv.z - mu.z) / sd.z) + (bsum.z + b1v.z);
                v.w = v.w * ((xv.w - mu.w) / sd.w) + (bsum.w + b1v.w);
                v.x = v.x >= 0.f ? v.x : v.x * p.slope; v.y = v.y >= 0.f ? v.y : v.y * p.slope;
                v.z = v.z >= 0.f ? v.z : v.z * p.slope; v.w = v.w >= 0.f ? v.w : v.w * p.slope;
            }
        }
        float* opix = p.out + (size_t)p.out_off + (size_t)b * p.out_pb + (size_t)y * p.out_py + (size_t)x * p.out_px;
        if (EPI == EPI_SPADE && p.out_split == 4) msr_store_f16c4_dev(opix, c, v.x, v.y, v.z, v.w);
        else if (EPI == EPI_SPADE && p.out_split) msr_store_split4_dev(opix, c, v.x, v.y, v.z, v.w);
        else *reinterpret_cast<float4*>(opix + c) = v;
    }
}

// ------------------------------------------------------------------------------------------------------
template <int WM, int WN, int MT, int NT, int BKC>
struct TileCfg {
    static constexpr int BM = WM * MT * 32, BN = WN * NT * 32, NTHR = WM * WN * 64;
    static constexpr size_t LDS = (size_t)(2 * BM + 2 * BN) * (BKC + 4) * sizeof(float);
};
// TILE_128x128   : 4 waves, K-step 32, 72 KiB LDS -> 2 workgroups (2 waves / SIMD) per CU
// TILE_64x64     : 2 waves, K-step 32, 36 KiB LDS -> 4 workgroups per CU (low-resolution layers, with split-K)
// TILE_128x128_K16: 4 waves, K-step 16, 40 KiB LDS -> 3 workgroups (3 waves / SIMD) per CU

template <int WM, int WN, int MT, int NT, int BKC, int EPI, int PREC>
static hipError_t set_attr() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm<WM, WN, MT, NT, BKC, EPI, PREC>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)TileCfg<WM, WN, MT, NT, BKC>::LDS);
}

template <int WM, int WN, int MT, int NT, int BKC, int PREC>
static hipError_t set_attr_all() {
    hipError_t e;
    if ((e = set_attr<WM, WN, MT, NT, BKC, EPI_BIAS, PREC>()) != hipSuccess) return e;
    if ((e = set_attr<WM, WN, MT, NT, BKC, EPI_RES, PREC>()) != hipSuccess) return e;
    if ((e = set_attr<WM, WN, MT, NT, BKC, EPI_SPADE, PREC>()) != hipSuccess) return e;
    if ((e = set_attr<WM, WN, MT, NT, BKC, EPI_AFFINE, PREC>()) != hipSuccess) return e;
    return set_attr<WM, WN, MT, NT, BKC, EPI_PARTIAL, PREC>();
}

template <int WM, int WN, int MT, int NT>
struct TileCfgB {   // split-bf16 kernel: only the activation tile lives in LDS
    static constexpr int BM = WM * MT * 32, BN = WN * NT * 32, NTHR = WM * WN * 64;
    static constexpr size_t LDS = (size_t)(2 * BM) * 36 * sizeof(float);
};

template <int WM, int WN, int MT, int NT>
static hipError_t set_attr_bf16x3() {
    hipError_t e;
    const int lds = (int)TileCfgB<WM, WN, MT, NT>::LDS;
#define MSR_SET(EPI)                                                                                          \
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_bf16x3<WM, WN, MT, NT, EPI>),       \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, lds)) != hipSuccess)              \
        return e;
    MSR_SET(EPI_BIAS) MSR_SET(EPI_RES) MSR_SET(EPI_SPADE) MSR_SET(EPI_PARTIAL)
#undef MSR_SET
    return hipSuccess;
}

static constexpr size_t HALO_LDS = (size_t)(180 + 2 * 128) * 36 * sizeof(float);
static constexpr size_t HALO16_LDS = (size_t)(180 + 2 * 128) * 40 * sizeof(float);

#ifdef MSR_PP_STAMPS
static constexpr size_t PP_LDS = (size_t)(2 * 324 + 2 * 128) * 40 * sizeof(float) + 8192;   // + the stamp words
#else
static constexpr size_t PP_LDS = (size_t)(2 * 324 + 2 * 128) * 40 * sizeof(float);
#endif
// PP_F16C + EPI_SPADE launches: + 8 waves x 16 lines x 36 dwords of epilogue store assembly = 163,072 B of the CU's 163,840
#ifdef MSR_PP_STAMPS
static constexpr size_t PP_STAGE_LDS = 0;
#else
static constexpr size_t PP_STAGE_LDS = (size_t)8 * 16 * 36 * sizeof(unsigned);
#endif

static hipError_t set_attr_halo() {
    hipError_t e;
#define MSR_SETPP(EPI, ...)                                                                                   \
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_bf16x3_pp<EPI, __VA_ARGS__>),       \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)PP_LDS)) != hipSuccess)      \
        return e;
    MSR_SETPP(EPI_BIAS, PP_BF16X3) MSR_SETPP(EPI_RES, PP_BF16X3) MSR_SETPP(EPI_SPADE, PP_BF16X3)
    MSR_SETPP(EPI_SPADE, PP_F16X2) MSR_SETPP(EPI_PARTIAL, PP_BF16X3)
    MSR_SETPP(EPI_BIAS, PP_FP8) MSR_SETPP(EPI_RES, PP_FP8) MSR_SETPP(EPI_SPADE, PP_FP8)
    MSR_SETPP(EPI_BIAS, PP_FP8, true) MSR_SETPP(EPI_RES, PP_FP8, true) MSR_SETPP(EPI_SPADE, PP_FP8, true)
    MSR_SETPP(EPI_BIAS, PP_F16C) MSR_SETPP(EPI_RES, PP_F16C) MSR_SETPP(EPI_PARTIAL, PP_F16C)
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_bf16x3_pp<EPI_SPADE, PP_F16C>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)(PP_LDS + PP_STAGE_LDS))) != hipSuccess)
        return e;
#undef MSR_SETPP
#define MSR_SET(EPI)                                                                                          \
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_bf16x3_halo<EPI, 0>),               \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)HALO_LDS)) != hipSuccess)    \
        return e;                                                                                             \
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_bf16x3_halo<EPI, 1>),               \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)HALO16_LDS)) != hipSuccess)  \
        return e;
    MSR_SET(EPI_BIAS) MSR_SET(EPI_RES) MSR_SET(EPI_SPADE)
#undef MSR_SET
    return hipSuccess;
}

hipError_t conv_igemm_init() {
    hipError_t e;
    if ((e = set_attr_halo()) != hipSuccess) return e;
    if ((e = conv_sw_init()) != hipSuccess) return e;
    if ((e = set_attr_all<2, 2, 2, 2, 32, PREC_F32>()) != hipSuccess) return e;
    if ((e = set_attr_all<2, 1, 1, 2, 32, PREC_F32>()) != hipSuccess) return e;
    if ((e = set_attr_all<2, 2, 2, 2, 16, PREC_F32>()) != hipSuccess) return e;
    if ((e = set_attr_all<2, 2, 2, 2, 32, PREC_BF16X3>()) != hipSuccess) return e;
    if ((e = set_attr_all<2, 1, 1, 2, 32, PREC_BF16X3>()) != hipSuccess) return e;
    if ((e = set_attr_bf16x3<2, 2, 2, 2>()) != hipSuccess) return e;
    return set_attr_bf16x3<2, 1, 1, 2>();
}

static int ilog2_floor(int v) {
    int l = 0;
    while ((2 << l) <= v) ++l;
    return l;
}

static bool make_geom(const ConvParams& p, int BM, int BN, int BKC, TileGeom& g) {
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
    g.tiles_mn = g.tiles_x * g.tiles_y * g.tiles_b * g.tiles_n;
    g.walk_pb = 1;
    g.walk_nb = g.tiles_n;
    return true;
}

int conv_stat_slabs(const ConvParams& p, int tile) {
    const int bm = tile == TILE_64x64 ? 64 : 128, wm = 2;
    TileGeom g;
    if (!make_geom(p, bm, bm, tile == TILE_128x128_K16 ? 16 : 32, g)) return 0;
    return g.tiles_x * g.tiles_y * g.tiles_b * wm;
}

// Stage 1: grid (C/32, groups): 32 slab slots x 32 channels per workgroup, sequential Chan per slot over the
// group's slab range, fixed-order combine of the 32 slots -> one (count, mean, M2) triple per (group, channel).
// Stage 2: the same kernel over the stage-1 triples with one group and final = 1.  fp64, deterministic order.
template <typename T>
__global__ void __launch_bounds__(1024) moments_from_slabs_kernel(const T* __restrict__ partial, int P, int C,
                                                                  int final_stage, float eps,
                                                                  double* __restrict__ group_out,
                                                                  float* __restrict__ mean, float* __restrict__ stdv) {
    __shared__ double red[32][32][3];
    const int c = blockIdx.x * 32 + (threadIdx.x & 31);
    const int slot = threadIdx.x >> 5;
    const int groups = gridDim.y, grp = blockIdx.y;
    const int per = (P + groups - 1) / groups;
    const int k0 = grp * per, k1 = min(P, k0 + per);
    double n = 0, mu = 0, m2 = 0;
    for (int k = k0 + slot; k < k1; k += 32) {
        const T* o = partial + (size_t)k * 3 * C + c;
        const double bn = (double)o[0], bmu = (double)o[C], bm2 = (double)o[2 * C];
        if (bn > 0) {
            const double tot = n + bn, delta = bmu - mu;
            m2 += bm2 + delta * delta * (n * bn / tot);
            mu += delta * (bn / tot);
            n = tot;
        }
    }
    red[slot][threadIdx.x & 31][0] = n; red[slot][threadIdx.x & 31][1] = mu; red[slot][threadIdx.x & 31][2] = m2;
    __syncthreads();
    if (slot == 0) {
        for (int k = 1; k < 32; ++k) {
            const double bn = red[k][threadIdx.x][0], bmu = red[k][threadIdx.x][1], bm2 = red[k][threadIdx.x][2];
            if (bn > 0) {
                const double tot = n + bn, delta = bmu - mu;
                m2 += bm2 + delta * delta * (n * bn / tot);
                mu += delta * (bn / tot);
                n = tot;
            }
        }
        if (final_stage) {
            const double var = n > 0 ? m2 / n : 0.0;
            mean[c] = (float)mu;
            stdv[c] = sqrtf((float)var + eps);
        } else {
            double* o = group_out + (size_t)grp * 3 * C + c;
            o[0] = n; o[C] = mu; o[2 * C] = m2;
        }
    }
}

hipError_t launch_moments_from_slabs(const float* partial, int P, int C, float eps, double* group_ws, float* mean,
                                     float* stdv, hipStream_t s) {
    if (C % 32 || P <= 0) return hipErrorInvalidValue;
    int groups = P < 512 ? 1 : P / 64;       // up to a few hundred slabs one launch is faster than two (5-8 us each); 512 slabs in
                                             // one launch were 16 sequential fp64 Chan updates per thread: 18 us
    if (groups > 128) groups = 128;          // 512 workgroups at C = 128 (32 groups = 128 workgroups pulled 12.6 MB of slabs in 17 us)
    if (groups <= 1) {
        moments_from_slabs_kernel<float><<<dim3(C / 32, 1), 1024, 0, s>>>(partial, P, C, 1, eps, nullptr, mean, stdv);
    } else {
        moments_from_slabs_kernel<float><<<dim3(C / 32, groups), 1024, 0, s>>>(partial, P, C, 0, eps, group_ws, mean, stdv);
        moments_from_slabs_kernel<double><<<dim3(C / 32, 1), 1024, 0, s>>>(group_ws, groups, C, 1, eps, nullptr, mean, stdv);
    }
    return hipGetLastError();
}

int conv_pick_tile(int M, int N, int epilogue, int prec, int ksteps) {
    // The big tile needs >= ~2 waves of workgroups per CU to hide its barrier; otherwise take the small one.
    // Measured on MI355X (tools/gpu_conv_bench.py): the 16-channel K-step (3 workgroups per CU) is ~8 % faster
    // than the 32-channel one for the SPADE epilogue (its long epilogue is covered by a third resident
    // workgroup) and ~3 % slower for plain long-K convs.
    // At bf16 rates the 128 x 128 tile is 1.6x as efficient as the 64 x 64 one, so one workgroup per CU is enough.
    const long big_blocks = (long)((M + 127) / 128) * (N / 128);
    if (N % 128 == 0 && big_blocks >= (prec == PREC_BF16X3 ? 256 : 512))
        return (epilogue == EPI_SPADE && prec == PREC_F32) ? TILE_128x128_K16 : TILE_128x128;
    // Few pixels, long K (the r <= 8 main convs, r = 16 at small batch): these layers stream their weights, and the
    // 128-row tile reads each weight for twice as many pixels; split-K (>= 72 K-steps) supplies the workgroups.
    // tools/gpu_smalltile_sweep.py: 93 -> 70 us (B=16, r=8, 1024 -> 1024), 171 -> 120 us (B=8, r=16).
    if (prec == PREC_BF16X3 && N % 128 == 0 && M >= 128 && ksteps >= 72 && big_blocks * 16 >= 256) return TILE_128x128;
    return TILE_64x64;
}

int conv_pick_ksplit(int M, int N, int ksteps, int tile, int prec) {
    // Low-resolution layers (M = B*r*r of a few hundred pixels) do not produce enough tiles to fill 256 CUs:
    // cut K so that about 1024 small (512 big) workgroups exist.  Every range keeps >= 4 K-steps.
    const int bm = tile == TILE_64x64 ? 64 : 128;
    const long blocks = (long)((M + bm - 1) / bm) * (N / bm);
    long want = tile == TILE_64x64 ? 1024 : 512;
    // short K at bf16 rates (the gamma/beta convs, 36 K-steps): the split-K pass costs more than it buys beyond one
    // workgroup per CU (tools/gpu_smalltile_sweep.py: 28 vs 33 us at B=16, r=8)
    if (prec == PREC_BF16X3 && ksteps < 72) want = 256;
    int ks = 1;
    while (blocks * ks < want && ks < 16 && ksteps / (ks * 2) >= 4) ks *= 2;
    return ks;
}

template <int WM, int WN, int MT, int NT, int BKC, int PREC>
static hipError_t launch_cfg(const ConvParams& p, int epi, hipStream_t s) {
    using C = TileCfg<WM, WN, MT, NT, BKC>;
    TileGeom g;
    if (!make_geom(p, C::BM, C::BN, BKC, g)) return hipErrorInvalidValue;
    if (p.ksplit > 1) {
        if (!p.partial || epi == EPI_PARTIAL) return hipErrorInvalidValue;
        conv_igemm<WM, WN, MT, NT, BKC, EPI_PARTIAL, PREC><<<g.tiles_mn * p.ksplit, C::NTHR, C::LDS, s>>>(p, g);
        if (p.mom_mean) return launch_splitk_epilogue_mom(p, epi, s);      // epilogue + the output's moments, one launch
        const int Cout = epi == EPI_SPADE ? p.N / 2 : p.N;
        long eb = ((long)p.B * p.Hout * p.Wout * (Cout / 4) + 255) / 256;
        if (eb > 4096) eb = 4096;
        if (epi == EPI_BIAS) splitk_epilogue_kernel<EPI_BIAS><<<(int)eb, 256, 0, s>>>(p);
        else if (epi == EPI_RES) splitk_epilogue_kernel<EPI_RES><<<(int)eb, 256, 0, s>>>(p);
        else if (epi == EPI_AFFINE) splitk_epilogue_kernel<EPI_AFFINE><<<(int)eb, 256, 0, s>>>(p);
        else splitk_epilogue_kernel<EPI_SPADE><<<(int)eb, 256, 0, s>>>(p);
        return hipGetLastError();
    }
    const int grid = g.tiles_mn;
    switch (epi) {
        case EPI_AFFINE:
            conv_igemm<WM, WN, MT, NT, BKC, EPI_AFFINE, PREC><<<grid, C::NTHR, C::LDS, s>>>(p, g);
            break;
        case EPI_BIAS:
            conv_igemm<WM, WN, MT, NT, BKC, EPI_BIAS, PREC><<<grid, C::NTHR, C::LDS, s>>>(p, g);
            break;
        case EPI_RES:
            conv_igemm<WM, WN, MT, NT, BKC, EPI_RES, PREC><<<grid, C::NTHR, C::LDS, s>>>(p, g);
            break;
        case EPI_SPADE:
            conv_igemm<WM, WN, MT, NT, BKC, EPI_SPADE, PREC><<<grid, C::NTHR, C::LDS, s>>>(p, g);
            break;
        default:
            return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <int WM, int WN, int MT, int NT>
static hipError_t launch_bf16x3(const ConvParams& p, int epi, hipStream_t s) {
    using C = TileCfgB<WM, WN, MT, NT>;
    TileGeom g;
    if (!make_geom(p, C::BM, C::BN, 32, g)) return hipErrorInvalidValue;
    const int grid = g.tiles_mn * (p.ksplit > 1 ? p.ksplit : 1);
    if (p.ksplit > 1) {
        if (!p.partial || epi == EPI_PARTIAL) return hipErrorInvalidValue;
        conv_igemm_bf16x3<WM, WN, MT, NT, EPI_PARTIAL><<<grid, C::NTHR, C::LDS, s>>>(p, g);
        if (p.mom_mean) return launch_splitk_epilogue_mom(p, epi, s);      // epilogue + the output's moments, one launch
        const int Cout = epi == EPI_SPADE ? p.N / 2 : p.N;
        long eb = ((long)p.B * p.Hout * p.Wout * (Cout / 4) + 255) / 256;
        if (eb > 4096) eb = 4096;
        if (epi == EPI_BIAS) splitk_epilogue_kernel<EPI_BIAS><<<(int)eb, 256, 0, s>>>(p);
        else if (epi == EPI_RES) splitk_epilogue_kernel<EPI_RES><<<(int)eb, 256, 0, s>>>(p);
        else splitk_epilogue_kernel<EPI_SPADE><<<(int)eb, 256, 0, s>>>(p);
        return hipGetLastError();
    }
    switch (epi) {
        case EPI_BIAS: conv_igemm_bf16x3<WM, WN, MT, NT, EPI_BIAS><<<grid, C::NTHR, C::LDS, s>>>(p, g); break;
        case EPI_RES: conv_igemm_bf16x3<WM, WN, MT, NT, EPI_RES><<<grid, C::NTHR, C::LDS, s>>>(p, g); break;
        case EPI_SPADE: conv_igemm_bf16x3<WM, WN, MT, NT, EPI_SPADE><<<grid, C::NTHR, C::LDS, s>>>(p, g); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

static hipError_t launch_halo(const ConvParams& p, int epi, int sh, hipStream_t s) {
    TileGeom g;
    if (!make_geom(p, 128, 128, 32, g)) return hipErrorInvalidValue;
    if (g.tb != 1 || g.th_l != 3 || g.tw_l != 4 || p.stride != 1 || p.KH != 3 || p.KW != 3 || p.ksplit > 1 ||
        p.Cin % 64)   // the K loop is unrolled by two steps: 9 * (Cin / 32) must be even
        return hipErrorInvalidValue;
    if ((size_t)p.B * p.in_pb * sizeof(float) >= ((size_t)1 << 31)) return hipErrorInvalidValue;   // buffer descriptor range
    if (sh) {
        switch (epi) {
            case EPI_BIAS: conv_igemm_bf16x3_halo<EPI_BIAS, 1><<<g.tiles_mn, 256, HALO16_LDS, s>>>(p, g); break;
            case EPI_RES: conv_igemm_bf16x3_halo<EPI_RES, 1><<<g.tiles_mn, 256, HALO16_LDS, s>>>(p, g); break;
            case EPI_SPADE: conv_igemm_bf16x3_halo<EPI_SPADE, 1><<<g.tiles_mn, 256, HALO16_LDS, s>>>(p, g); break;
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    switch (epi) {
        case EPI_BIAS: conv_igemm_bf16x3_halo<EPI_BIAS, 0><<<g.tiles_mn, 256, HALO_LDS, s>>>(p, g); break;
        case EPI_RES: conv_igemm_bf16x3_halo<EPI_RES, 0><<<g.tiles_mn, 256, HALO_LDS, s>>>(p, g); break;
        case EPI_SPADE: conv_igemm_bf16x3_halo<EPI_SPADE, 0><<<g.tiles_mn, 256, HALO_LDS, s>>>(p, g); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

static hipError_t launch_pp(const ConvParams& p, int epi, hipStream_t s) {
    TileGeom g;
    if (!make_geom(p, 256, 128, 32, g)) return hipErrorInvalidValue;
    const bool one = p.prec == PREC_FP8 && p.Cin == 32;            // one 128-byte chunk: two tiles per unrolled body
    if (!one) conv_walk(g);
    if (g.tb != 1 || g.th_l != 4 || g.tw_l != 4 || p.stride != 1 || p.KH != 3 || p.KW != 3 || (!one && p.Cin % 64))
        return hipErrorInvalidValue;
    const int ksn = p.ksplit > 1 ? p.ksplit : 1;
    if (!one && (p.Cin / 64) % ksn) return hipErrorInvalidValue;  // every K range is a whole number of chunk pairs
    if (one && ksn > 1) return hipErrorInvalidValue;
    if ((size_t)p.B * p.in_pb * sizeof(float) >= ((size_t)1 << 31)) return hipErrorInvalidValue;   // buffer descriptor range
    // persistent: one workgroup per CU (144 KB of LDS each), a multiple of 8 so that every XCD gets the same count
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorInvalidValue;
        n_cu = prop.multiProcessorCount & ~7;
        if (n_cu < 8) n_cu = 8;
    }
    const int items = g.tiles_mn * ksn;
    const int grid = items < n_cu ? ((items + 7) & ~7) : n_cu;
    if (ksn > 1) {
        // few tiles: K ranges fill the chip, raw accumulators go to the split-K workspace, one more pass finishes
        if (!p.partial || (p.prec != PREC_BF16X3 && p.prec != PREC_F16C)) return hipErrorInvalidValue;     // K ranges: 3-term and f16c forms
        if (p.prec == PREC_F16C) {
            if (!p.wexp) return hipErrorInvalidValue;
            conv_igemm_bf16x3_pp<EPI_PARTIAL, PP_F16C><<<grid, 512, PP_LDS, s>>>(p, g);
        } else {
            conv_igemm_bf16x3_pp<EPI_PARTIAL, PP_BF16X3><<<grid, 512, PP_LDS, s>>>(p, g);
        }
        if (p.mom_mean) return launch_splitk_epilogue_mom(p, epi, s);      // epilogue + the output's moments, one launch
        const int Cout = epi == EPI_SPADE ? p.N / 2 : p.N;
        long eb = ((long)p.B * p.Hout * p.Wout * (Cout / 4) + 255) / 256;
        if (eb > 4096) eb = 4096;
        if (epi == EPI_BIAS) splitk_epilogue_kernel<EPI_BIAS><<<(int)eb, 256, 0, s>>>(p);
        else if (epi == EPI_RES) splitk_epilogue_kernel<EPI_RES><<<(int)eb, 256, 0, s>>>(p);
        else if (epi == EPI_SPADE) splitk_epilogue_kernel<EPI_SPADE><<<(int)eb, 256, 0, s>>>(p);
        else return hipErrorInvalidValue;
        return hipGetLastError();
    }
    if (p.prec == PREC_F16X2) {
        if (epi != EPI_SPADE) return hipErrorInvalidValue;     // the 2-term form exists for the gamma|beta convs only
        conv_igemm_bf16x3_pp<EPI_SPADE, PP_F16X2><<<grid, 512, PP_LDS, s>>>(p, g);
        return hipGetLastError();
    }
    if (p.prec == PREC_F16C) {
        if (!p.wexp) return hipErrorInvalidValue;
        // conv_sw.hip (one software-pipelined wave per SIMD) takes the long-K main convs; the gamma|beta convs stay here, where
        // a second wave on the SIMD hides their SPADE epilogue.  MSR_F16C_SW = 0: everything here, 2: everything there (A/B).
        static const int sw_mode = std::getenv("MSR_F16C_SW") ? std::atoi(std::getenv("MSR_F16C_SW")) : 1;
        if (p.Cin % 128 == 0 && !(epi == EPI_SPADE && p.out_split == 5) &&     // (the fp6 image is written by this kernel's epilogue only)
            (sw_mode == 2 || (sw_mode == 1 && epi != EPI_SPADE)))
            return launch_conv_f16c_sw(p, epi, s);
        switch (epi) {
            case EPI_BIAS: conv_igemm_bf16x3_pp<EPI_BIAS, PP_F16C><<<grid, 512, PP_LDS, s>>>(p, g); break;
            case EPI_RES: conv_igemm_bf16x3_pp<EPI_RES, PP_F16C><<<grid, 512, PP_LDS, s>>>(p, g); break;
            case EPI_SPADE: conv_igemm_bf16x3_pp<EPI_SPADE, PP_F16C><<<grid, 512, PP_LDS + PP_STAGE_LDS, s>>>(p, g); break;
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    if (p.prec == PREC_FP8) {
        if (!p.wexp) return hipErrorInvalidValue;
        if (one) {
            switch (epi) {
                case EPI_BIAS: conv_igemm_bf16x3_pp<EPI_BIAS, PP_FP8, true><<<grid, 512, PP_LDS, s>>>(p, g); break;
                case EPI_RES: conv_igemm_bf16x3_pp<EPI_RES, PP_FP8, true><<<grid, 512, PP_LDS, s>>>(p, g); break;
                case EPI_SPADE: conv_igemm_bf16x3_pp<EPI_SPADE, PP_FP8, true><<<grid, 512, PP_LDS, s>>>(p, g); break;
                default: return hipErrorInvalidValue;
            }
            return hipGetLastError();
        }
        switch (epi) {
            case EPI_BIAS: conv_igemm_bf16x3_pp<EPI_BIAS, PP_FP8><<<grid, 512, PP_LDS, s>>>(p, g); break;
            case EPI_RES: conv_igemm_bf16x3_pp<EPI_RES, PP_FP8><<<grid, 512, PP_LDS, s>>>(p, g); break;
            case EPI_SPADE: conv_igemm_bf16x3_pp<EPI_SPADE, PP_FP8><<<grid, 512, PP_LDS, s>>>(p, g); break;
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    switch (epi) {
        case EPI_BIAS: conv_igemm_bf16x3_pp<EPI_BIAS, PP_BF16X3><<<grid, 512, PP_LDS, s>>>(p, g); break;
        case EPI_RES: conv_igemm_bf16x3_pp<EPI_RES, PP_BF16X3><<<grid, 512, PP_LDS, s>>>(p, g); break;
        case EPI_SPADE: conv_igemm_bf16x3_pp<EPI_SPADE, PP_BF16X3><<<grid, 512, PP_LDS, s>>>(p, g); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_conv_igemm(const ConvParams& p, int epilogue, int tile, hipStream_t s) {
    if (p.prec == PREC_F16C6) return tile == TILE_256x128_PP ? launch_conv_f16c_sw(p, epilogue, s) : hipErrorInvalidValue;
    if (p.prec == PREC_F16X2 || p.prec == PREC_FP8 || p.prec == PREC_F16C)
        return tile == TILE_256x128_PP ? launch_pp(p, epilogue, s) : hipErrorInvalidValue;
    if (p.prec == PREC_BF16X3) {
        if (tile == TILE_256x128_PP) return launch_pp(p, epilogue, s);
        if (tile == TILE_128x128_HALO) return launch_halo(p, epilogue, 0, s);
        if (tile == TILE_128x128_HALO16) return launch_halo(p, epilogue, 1, s);
        if (p.wt_frag) {
            if (tile == TILE_64x64) return launch_bf16x3<2, 1, 1, 2>(p, epilogue, s);
            return launch_bf16x3<2, 2, 2, 2>(p, epilogue, s);
        }
        if (tile == TILE_64x64) return launch_cfg<2, 1, 1, 2, 32, PREC_BF16X3>(p, epilogue, s);
        return launch_cfg<2, 2, 2, 2, 32, PREC_BF16X3>(p, epilogue, s);
    }
    if (tile == TILE_128x128) return launch_cfg<2, 2, 2, 2, 32, PREC_F32>(p, epilogue, s);
    if (tile == TILE_128x128_K16) return launch_cfg<2, 2, 2, 2, 16, PREC_F32>(p, epilogue, s);
    return launch_cfg<2, 1, 1, 2, 32, PREC_F32>(p, epilogue, s);
}

}  // namespace msr
