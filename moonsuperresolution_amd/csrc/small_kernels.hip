// Memory-bound and tiny kernels of the SPADE generator: mask embedding / first encoder conv, moments,
// instance-norm apply, dense layers, latent sampler and the fused up-sample + leaky-relu + 4x4 conv head.
// All are HBM-bound (or launch-bound) on MI355X: 16-byte coalesced accesses, 64-wide wave reductions.
#include "kernels.h"
#include <cstdlib>

namespace msr {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ------------------------------------------------------------------------------------------------
// conv_smallcin: 3x3 conv from the 2-channel source with an affine index map.
//   SPADE mask embedding:  mask = tf.image.resize(source, (r,r), 'nearest') ; relu(conv3x3(mask))
//                          (spade.py:17-18): t = y + kh - 1 valid in [0,r), source row = t*f + f/2.
//   encoder block 1:       Conv2D(64, 3, strides=2, 'same', no bias) + LeakyReLU (blocks.py:52-65,
//                          networks.py:16-18): t = 2y + kh valid in [0,S), source row = t.
// One thread owns 4 output channels (weights for its 18 taps live in registers) and walks pixels.
// ------------------------------------------------------------------------------------------------
template <int COUT, int PX>   // PX adjacent output pixels per thread share their 3 x (PX+2) source window
__global__ void __launch_bounds__(256) conv_smallcin_kernel(const SmallCinParams p) {
    MSR_SATURATING_CONVERSIONS();
    constexpr int QUADS = COUT / 4;          // threads per pixel group
    constexpr int GRP = 256 / QUADS;         // pixel groups per block pass
    const int q = threadIdx.x % QUADS;
    const int pl = threadIdx.x / QUADS;
    float4 w[18];
#pragma unroll
    for (int t = 0; t < 18; ++t) w[t] = *reinterpret_cast<const float4*>(p.w + t * COUT + q * 4);
    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) bias = *reinterpret_cast<const float4*>(p.bias + q * 4);
    const int gpr = p.Hout / PX;             // groups per output row
    const long total = (long)p.B * p.Hout * gpr;
    for (long grp = (long)blockIdx.x * GRP + pl; grp < total; grp += (long)gridDim.x * GRP) {
        const int x0 = (int)(grp % gpr) * PX;
        const int y = (int)((grp / gpr) % p.Hout);
        const int b = (int)(grp / ((long)gpr * p.Hout));
        const float* sb = p.src + (size_t)b * p.S * p.S * 2;
        // source window: rows kh = 0..2, columns j = 0..(PX-1)*ay+2  (ay = 1: PX+2 columns; ay = 2: 2*PX+1)
        constexpr int WMAX = 2 * PX + 1;
        float2 win[3][WMAX];
        const int ncol = (PX - 1) * p.ay + 3;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int ty = y * p.ay + kh + p.cy;
            const bool oky = ty >= 0 && ty < p.lim;
            const size_t rowoff = (size_t)(ty * p.f + p.o) * p.S;
#pragma unroll
            for (int j = 0; j < WMAX; ++j) {
                const int tx = x0 * p.ay + j + p.cy;
                const bool ok = oky && j < ncol && tx >= 0 && tx < p.lim;
                win[kh][j] = ok ? *reinterpret_cast<const float2*>(sb + (rowoff + (tx * p.f + p.o)) * 2)
                                : make_float2(0.f, 0.f);
            }
        }
        float* o = p.out + (size_t)p.out_off + (size_t)b * p.out_pb + (size_t)y * p.out_py + (size_t)x0 * p.out_px;
#pragma unroll
        for (int i = 0; i < PX; ++i) {
            float4 acc = bias;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    // column of tap kw for pixel i: i*ay + kw (ay is 1 or 2; both instantiated branches are static)
                    const float2 v = p.ay == 1 ? win[kh][(i + kw) < WMAX ? (i + kw) : 0]
                                               : win[kh][(2 * i + kw) < WMAX ? (2 * i + kw) : 0];
                    const float4 w0 = w[(kh * 3 + kw) * 2], w1 = w[(kh * 3 + kw) * 2 + 1];
                    acc.x += v.x * w0.x; acc.y += v.x * w0.y; acc.z += v.x * w0.z; acc.w += v.x * w0.w;
                    acc.x += v.y * w1.x; acc.y += v.y * w1.y; acc.z += v.y * w1.z; acc.w += v.y * w1.w;
                }
            }
            if (p.act == 1) {
                acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
            } else if (p.act == 2) {
                acc.x = acc.x >= 0.f ? acc.x : acc.x * p.slope; acc.y = acc.y >= 0.f ? acc.y : acc.y * p.slope;
                acc.z = acc.z >= 0.f ? acc.z : acc.z * p.slope; acc.w = acc.w >= 0.f ? acc.w : acc.w * p.slope;
            }
            if (p.out_split == 4) {
                msr_store_f16c4_dev(o + (size_t)i * p.out_px, q * 4, acc.x, acc.y, acc.z, acc.w);
            } else if (p.out_split == 3) {
                unsigned w8 = 0;
                w8 = __builtin_amdgcn_cvt_pk_bf8_f32(acc.x, acc.y, w8, false);
                w8 = __builtin_amdgcn_cvt_pk_bf8_f32(acc.z, acc.w, w8, true);
                reinterpret_cast<unsigned*>(o + (size_t)i * p.out_px)[q] = w8;
            } else if (p.out_split == 2) msr_store_split4_f16(o + (size_t)i * p.out_px, q * 4, acc.x, acc.y, acc.z, acc.w);
            else if (p.out_split) msr_store_split4_dev(o + (size_t)i * p.out_px, q * 4, acc.x, acc.y, acc.z, acc.w);
            else *reinterpret_cast<float4*>(o + (size_t)i * p.out_px + q * 4) = acc;
        }
    }
}

// Tiled form for Hout % 16 == 0: a workgroup owns a 16 x 16 output tile.  Its source window ((16 * ay + 2)^2 samples, already
// resized / strided, zero outside) is staged through LDS once, so the inner loop has no bounds tests and no 64-bit address
// arithmetic, and a thread's 18 tap weights serve 16 (COUT = 64) or 32 (COUT = 128) pixels instead of 4.  Same
// multiplications and additions in the same order as conv_smallcin_kernel: bit-identical output.
template <int COUT>
__global__ void __launch_bounds__(256) conv_smallcin_tiled_kernel(const SmallCinParams p) {
    MSR_SATURATING_CONVERSIONS();
    constexpr int QUADS = COUT / 4, GRP = 256 / QUADS, PX = 4, WMAX = 2 * PX + 1, WW = 34;
    __shared__ float2 win[WW * WW];
    const int tiles = p.Hout >> 4;
    int t = blockIdx.x;
    const int tx0 = (t % tiles) << 4;
    t /= tiles;
    const int ty0 = (t % tiles) << 4;
    const int b = t / tiles;
    const int q = threadIdx.x % QUADS, pl = threadIdx.x / QUADS;
    const int span = 15 * p.ay + 3;                       // window rows / columns the tile needs
    const float* sb = p.src + (size_t)b * p.S * p.S * 2;
    for (int i = threadIdx.x; i < span * span; i += 256) {
        const int wy = i / span, wx = i - wy * span;
        const int ty = ty0 * p.ay + wy + p.cy, tx = tx0 * p.ay + wx + p.cy;
        const bool ok = ty >= 0 && ty < p.lim && tx >= 0 && tx < p.lim;
        win[wy * WW + wx] = ok ? *reinterpret_cast<const float2*>(sb + ((size_t)(ty * p.f + p.o) * p.S + (tx * p.f + p.o)) * 2)
                               : make_float2(0.f, 0.f);
    }
    float4 w[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) w[k] = *reinterpret_cast<const float4*>(p.w + k * COUT + q * 4);
    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) bias = *reinterpret_cast<const float4*>(p.bias + q * 4);
    __syncthreads();
    const int ncol = (PX - 1) * p.ay + 3;
    for (int grp = pl; grp < 64; grp += GRP) {            // 64 groups of 4 adjacent pixels in the tile
        const int gx = (grp & 3) * PX, gy = grp >> 2;
        float2 v[3][WMAX];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int j = 0; j < WMAX; ++j)
                v[kh][j] = j < ncol ? win[(gy * p.ay + kh) * WW + gx * p.ay + j] : make_float2(0.f, 0.f);
        float* o = p.out + (size_t)p.out_off + (size_t)b * p.out_pb + (size_t)(ty0 + gy) * p.out_py + (size_t)(tx0 + gx) * p.out_px;
#pragma unroll
        for (int i = 0; i < PX; ++i) {
            float4 acc = bias;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const float2 x = p.ay == 1 ? v[kh][(i + kw) < WMAX ? (i + kw) : 0] : v[kh][(2 * i + kw) < WMAX ? (2 * i + kw) : 0];
                    const float4 w0 = w[(kh * 3 + kw) * 2], w1 = w[(kh * 3 + kw) * 2 + 1];
                    acc.x += x.x * w0.x; acc.y += x.x * w0.y; acc.z += x.x * w0.z; acc.w += x.x * w0.w;
                    acc.x += x.y * w1.x; acc.y += x.y * w1.y; acc.z += x.y * w1.z; acc.w += x.y * w1.w;
                }
            }
            if (p.act == 1) {
                acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
            } else if (p.act == 2) {
                acc.x = acc.x >= 0.f ? acc.x : acc.x * p.slope; acc.y = acc.y >= 0.f ? acc.y : acc.y * p.slope;
                acc.z = acc.z >= 0.f ? acc.z : acc.z * p.slope; acc.w = acc.w >= 0.f ? acc.w : acc.w * p.slope;
            }
            if (p.out_split == 4) {
                msr_store_f16c4_dev(o + (size_t)i * p.out_px, q * 4, acc.x, acc.y, acc.z, acc.w);
            } else if (p.out_split == 3) {
                unsigned w8 = 0;
                w8 = __builtin_amdgcn_cvt_pk_bf8_f32(acc.x, acc.y, w8, false);
                w8 = __builtin_amdgcn_cvt_pk_bf8_f32(acc.z, acc.w, w8, true);
                reinterpret_cast<unsigned*>(o + (size_t)i * p.out_px)[q] = w8;
            } else if (p.out_split == 2) msr_store_split4_f16(o + (size_t)i * p.out_px, q * 4, acc.x, acc.y, acc.z, acc.w);
            else if (p.out_split) msr_store_split4_dev(o + (size_t)i * p.out_px, q * 4, acc.x, acc.y, acc.z, acc.w);
            else *reinterpret_cast<float4*>(o + (size_t)i * p.out_px + q * 4) = acc;
        }
    }
}

template <int COUT>
static hipError_t launch_smallcin_t(const SmallCinParams& p, hipStream_t s) {
    static const bool tiled_off = std::getenv("MSR_SMALLCIN_TILED") && std::atoi(std::getenv("MSR_SMALLCIN_TILED")) == 0;
    // (a handful of tiles — the low-resolution embeddings — finish sooner one pixel group per thread)
    if (p.Hout % 16 == 0 && p.B * (p.Hout / 16) * (p.Hout / 16) >= 64 && !tiled_off) {
        conv_smallcin_tiled_kernel<COUT><<<p.B * (p.Hout / 16) * (p.Hout / 16), 256, 0, s>>>(p);
        return hipGetLastError();
    }
    constexpr int GRP = 256 / (COUT / 4);
    const int px = (p.Hout % 4 == 0) ? 4 : 1;
    const long total = (long)p.B * p.Hout * (p.Hout / px);
    long blocks = (total + GRP - 1) / GRP;
    if (blocks > 8192) blocks = 8192;
    if (px == 4) conv_smallcin_kernel<COUT, 4><<<(int)blocks, 256, 0, s>>>(p);
    else conv_smallcin_kernel<COUT, 1><<<(int)blocks, 256, 0, s>>>(p);
    return hipGetLastError();
}

hipError_t launch_conv_smallcin(const SmallCinParams& p, hipStream_t s) {
    if (p.ay != 1 && p.ay != 2) return hipErrorInvalidValue;
    if (p.Cout == 128) return launch_smallcin_t<128>(p, s);
    if (p.Cout == 64) return launch_smallcin_t<64>(p, s);
    return hipErrorInvalidValue;
}

// fp32 -> split-bf16 chunk image; the innermost dimension must be a multiple of 32 (n % 32 == 0)
__global__ void __launch_bounds__(256) split_bf16_kernel(const float* __restrict__ in, float* __restrict__ out, long n) {
    const long quads = n / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < quads; i += (long)gridDim.x * 256) {
        const float4 v = *reinterpret_cast<const float4*>(in + i * 4);
        const long e = i * 4;
        msr_store_split4_dev(out + (e & ~31L), (int)(e & 31), v.x, v.y, v.z, v.w);
    }
}
hipError_t launch_split_bf16(const float* in, float* out, long n, hipStream_t s) {
    if (n % 32) return hipErrorInvalidValue;
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    if (n > 0) split_bf16_kernel<<<(int)blocks, 256, 0, s>>>(in, out, n);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// moments: per-group per-channel mean and sqrt(biased var + eps), fp64 accumulation of sum and sum of
// squares (exact to fp64 rounding, so the one-pass form equals TF's two-pass moments at fp32 precision).
// Stage 1: grid (chunks, G); thread = (channel quad, pixel slot); Stage 2: one thread per (g, c).
// ------------------------------------------------------------------------------------------------
// Pixels per stage-1 workgroup: 128 for the big tensors (thousands of workgroups keep enough bytes in flight); the
// low-resolution tensors (a few hundred pixels) take smaller chunks so that at least ~256 workgroups exist —
// with 2 workgroups of 128 pixels the kernel was one 20 us latency chain.
static int moments_chunk_pixels(int G, int P) {
    int px = 128;
    while (px > 8 && (long)G * ((P + px - 1) / px) < 256) px >>= 1;
    return px;
}
// ... and when 8-pixel chunks are still too few workgroups, the channels are cut as well (>= 64 quads per workgroup)
static int moments_channel_blocks(int G, int P, int C) {
    const long blocks = (long)G * ((P + moments_chunk_pixels(G, P) - 1) / moments_chunk_pixels(G, P));
    int cb = 1;
    while (blocks * cb < 256 && C / 4 / (cb * 2) >= 64 && (C / 4) % (cb * 2) == 0) cb *= 2;
    return cb;
}
int moments_chunks(int G, int P) {
    const int px = moments_chunk_pixels(G, P);
    return (P + px - 1) / px;
}

// sums of one pixel chunk: the values come from `load(pixel, quad)` (a tensor in memory, or the split-K epilogue's freshly
// combined output).  partial layout: [G][chunks][C][2]
// A workgroup covers the channel quads [q_lo, q_lo + quads) of its pixel chunk (blockIdx.z cuts the channels when the
// pixels alone do not give ~256 workgroups).
template <typename Load>
__device__ __forceinline__ void moments_chunk_sums(Load load, int p0, int p1, int C, int q_lo, int quads, int g, int chunk,
                                                   int chunks, double* __restrict__ partial, double* red) {
    for (int qb = 0; qb < quads; qb += 256) {
        // layout A (quads >= 256): every thread one quad, all pixels.  layout B: several pixel slots per quad.
        const int tq = quads >= 256 ? 256 : quads;     // threads along channel quads
        const int slots = 256 / tq;                    // pixel slots
        const int q = q_lo + qb + threadIdx.x % tq;
        const int slot = threadIdx.x / tq;
        double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
        if (q < q_lo + quads) {
            // 4 independent 16-byte loads in flight per thread; short fp32 runs flushed into fp64 accumulators
            int pix = p0 + slot;
            for (; pix + 3 * slots < p1; pix += 4 * slots) {
                const float4 v0 = load(pix, q), v1 = load(pix + slots, q), v2 = load(pix + 2 * slots, q), v3 = load(pix + 3 * slots, q);
                s[0] += (double)((v0.x + v1.x) + (v2.x + v3.x)); s[1] += (double)((v0.y + v1.y) + (v2.y + v3.y));
                s[2] += (double)((v0.z + v1.z) + (v2.z + v3.z)); s[3] += (double)((v0.w + v1.w) + (v2.w + v3.w));
                ss[0] += (double)((v0.x * v0.x + v1.x * v1.x) + (v2.x * v2.x + v3.x * v3.x));
                ss[1] += (double)((v0.y * v0.y + v1.y * v1.y) + (v2.y * v2.y + v3.y * v3.y));
                ss[2] += (double)((v0.z * v0.z + v1.z * v1.z) + (v2.z * v2.z + v3.z * v3.z));
                ss[3] += (double)((v0.w * v0.w + v1.w * v1.w) + (v2.w * v2.w + v3.w * v3.w));
            }
            for (; pix < p1; pix += slots) {
                const float4 v = load(pix, q);
                s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
                ss[0] += (double)v.x * v.x; ss[1] += (double)v.y * v.y;
                ss[2] += (double)v.z * v.z; ss[3] += (double)v.w * v.w;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { red[threadIdx.x * 8 + k] = s[k]; red[threadIdx.x * 8 + 4 + k] = ss[k]; }
        __syncthreads();
        if (slot == 0 && q < q_lo + quads) {
            for (int k = 0; k < 4; ++k) {
                double a = 0, b = 0;
                for (int sl = 0; sl < slots; ++sl) {
                    a += red[(sl * tq + threadIdx.x) * 8 + k];
                    b += red[(sl * tq + threadIdx.x) * 8 + 4 + k];
                }
                double* o = partial + (((size_t)g * chunks + chunk) * C + q * 4 + k) * 2;
                o[0] = a; o[1] = b;
            }
        }
        __syncthreads();
    }
}

// one workgroup per (group, 32 channels): 32 chunk slots x 32 channels, fixed-order tree -> deterministic
__global__ void __launch_bounds__(1024) moments_final_kernel(const double* __restrict__ partial, int G, int chunks,
                                                            int C, int P, float eps, float* __restrict__ mean,
                                                            float* __restrict__ stdv) {
    __shared__ double red[32][32][2];
    const int cblocks = C / 32;
    const int g = blockIdx.x / cblocks, c = (blockIdx.x % cblocks) * 32 + (threadIdx.x & 31);
    const int slot = threadIdx.x >> 5;
    double s = 0, ss = 0;
    for (int k = slot; k < chunks; k += 32) {
        const double2 v = *reinterpret_cast<const double2*>(partial + (((size_t)g * chunks + k) * C + c) * 2);
        s += v.x; ss += v.y;
    }
    red[slot][threadIdx.x & 31][0] = s;
    red[slot][threadIdx.x & 31][1] = ss;
    __syncthreads();
    if (slot == 0) {
        for (int k = 1; k < 32; ++k) { s += red[k][threadIdx.x][0]; ss += red[k][threadIdx.x][1]; }
        const double m = s / P;
        double var = ss / P - m * m;
        if (var < 0) var = 0;
        mean[(size_t)g * C + c] = (float)m;
        // the reference adds eps in float32 and takes a float32 sqrt (spade.py:22); mirror that rounding
        stdv[(size_t)g * C + c] = sqrtf((float)var + eps);
    }
}

__global__ void __launch_bounds__(256) moments_partial_kernel(const float* __restrict__ x, int P, int C, int chunk_px,
                                                              double* __restrict__ partial) {
    __shared__ double red[256 * 8];
    const int chunk = blockIdx.x, g = blockIdx.y, chunks = gridDim.x;
    const int p0 = chunk * chunk_px;
    const int p1 = min(P, p0 + chunk_px);
    const int quads = C / 4 / gridDim.z;
    const float* xg = x + (size_t)g * P * C;
    moments_chunk_sums([&](int pix, int q) { return *reinterpret_cast<const float4*>(xg + (size_t)pix * C + q * 4); },
                       p0, p1, C, blockIdx.z * quads, quads, g, chunk, chunks, partial, red);
}

hipError_t launch_moments(const float* x, int G, int P, int C, float eps, double* partial, float* mean, float* stdv,
                          hipStream_t s) {
    if (C % 32) return hipErrorInvalidValue;
    const int chunks = moments_chunks(G, P);
    moments_partial_kernel<<<dim3(chunks, G, moments_channel_blocks(G, P, C)), 256, 0, s>>>(x, P, C, moments_chunk_pixels(G, P),
                                                                                          partial);
    moments_final_kernel<<<G * (C / 32), 1024, 0, s>>>(partial, G, chunks, C, P, eps, mean, stdv);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// splitk_epilogue_mom: the split-K epilogue (EPI_BIAS / EPI_RES) of a conv whose output feeds a normalisation, with that
// normalisation's moments: a workgroup combines the K ranges of one pixel chunk, writes the output and adds it up on the
// way — the tensor is not read again and the moments' first launch disappears (moments_final_kernel follows).  Same sums,
// same order as moments_partial_kernel over the finished tensor.  (A single launch that lets the last workgroup finish the
// sums was tried: the device-scope release every workgroup needs before its ticket writes back the XCD's L2 — 8 separate
// L2s on this part — and cost 20-120 us per layer, profiles/r03_early_phase.txt.)
// ------------------------------------------------------------------------------------------------
template <int EPI>
__global__ void __launch_bounds__(256) splitk_epilogue_mom_kernel(const ConvParams p, int chunk_px) {
    __shared__ double red[256 * 8];
    const int chunk = blockIdx.x, g = blockIdx.y, chunks = gridDim.x;
    const int P = (p.mom_G > 1 ? 1 : p.B) * p.Hout * p.Wout;        // pixels per group: one sample, or the whole batch
    const int p0 = chunk * chunk_px;
    const int p1 = min(P, p0 + chunk_px);
    const size_t pstride = (size_t)p.B * p.Hout * p.Wout * p.N;
    auto load = [&](int pl, int q) {
        // 32-bit pixel arithmetic (B * Hout * Wout < 2^31, checked by the launcher): the 64-bit divisions cost more than the loads
        const int pix = g * P + pl;
        const int x = pix % p.Wout;
        const int yb = pix / p.Wout;
        const int y = yb % p.Hout, b = yb / p.Hout;
        const int c = q * 4;
        const float* pp = p.partial + (size_t)pix * p.N + c;
        float4 a = *reinterpret_cast<const float4*>(pp);
        // the K ranges are added in range order (the plain epilogue's order); 8 loads in flight instead of one
#pragma unroll 8
        for (int k = 1; k < p.ksplit; ++k) {
            const float4 t = *reinterpret_cast<const float4*>(pp + k * pstride);
            a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
        }
        const float4 b0v = *reinterpret_cast<const float4*>(p.bias + c);
        float4 v = make_float4(a.x + b0v.x, a.y + b0v.y, a.z + b0v.z, a.w + b0v.w);
        if constexpr (EPI == EPI_RES) {
            const float4 xv = *reinterpret_cast<const float4*>(p.aux + (size_t)b * p.aux_pb +
                                                               (size_t)(y >> p.aux_shift) * p.aux_py +
                                                               (size_t)(x >> p.aux_shift) * p.aux_px + c);
            v.x += xv.x; v.y += xv.y; v.z += xv.z; v.w += xv.w;
        }
        float* opix = p.out + (size_t)p.out_off + (size_t)b * p.out_pb + (size_t)y * p.out_py + (size_t)x * p.out_px;
        *reinterpret_cast<float4*>(opix + c) = v;
        return v;
    };
    const int quads = p.N / 4 / gridDim.z;
    moments_chunk_sums(load, p0, p1, p.N, blockIdx.z * quads, quads, g, chunk, chunks, p.mom_partial, red);
}

hipError_t launch_splitk_epilogue_mom(const ConvParams& p, int epi, hipStream_t s) {
    const int G = p.mom_G > 1 ? p.B : 1;
    if ((epi != EPI_BIAS && epi != EPI_RES) || p.N % 32 || !p.mom_partial || (long)p.B * p.Hout * p.Wout >= (1L << 31))
        return hipErrorInvalidValue;
    const int P = (G > 1 ? 1 : p.B) * p.Hout * p.Wout;
    const int chunks = moments_chunks(G, P);
    const dim3 grid(chunks, G, moments_channel_blocks(G, P, p.N));
    if (epi == EPI_BIAS) splitk_epilogue_mom_kernel<EPI_BIAS><<<grid, 256, 0, s>>>(p, moments_chunk_pixels(G, P));
    else splitk_epilogue_mom_kernel<EPI_RES><<<grid, 256, 0, s>>>(p, moments_chunk_pixels(G, P));
    moments_final_kernel<<<G * (p.N / 32), 1024, 0, s>>>(p.mom_partial, G, chunks, p.N, P, p.mom_eps, p.mom_mean, p.mom_std);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// norm_act: tfa InstanceNormalization apply + LeakyReLU (blocks.py:62-65), written into the next
// conv's zero-bordered input (or the dense layer's flat input).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) norm_act_kernel(const NormActParams p) {
    const int quads = p.C / 4;
    const long total = (long)p.B * p.H * p.W * quads;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int q = (int)(i % quads);
        long pix = i / quads;
        const int x = (int)(pix % p.W);
        pix /= p.W;
        const int y = (int)(pix % p.H);
        const int b = (int)(pix / p.H);
        const float4 v = *reinterpret_cast<const float4*>(p.x + i * 4);
        const float4 m = *reinterpret_cast<const float4*>(p.mean + (size_t)b * p.C + q * 4);
        const float4 sd = *reinterpret_cast<const float4*>(p.stdv + (size_t)b * p.C + q * 4);
        const float4 ga = *reinterpret_cast<const float4*>(p.gamma + q * 4);
        const float4 be = *reinterpret_cast<const float4*>(p.beta + q * 4);
        float4 r;
        r.x = (v.x - m.x) / sd.x * ga.x + be.x;
        r.y = (v.y - m.y) / sd.y * ga.y + be.y;
        r.z = (v.z - m.z) / sd.z * ga.z + be.z;
        r.w = (v.w - m.w) / sd.w * ga.w + be.w;
        r.x = r.x >= 0.f ? r.x : r.x * p.slope; r.y = r.y >= 0.f ? r.y : r.y * p.slope;
        r.z = r.z >= 0.f ? r.z : r.z * p.slope; r.w = r.w >= 0.f ? r.w : r.w * p.slope;
        float* o = p.out + (size_t)p.out_off + (size_t)b * p.out_pb + (size_t)y * p.out_py + (size_t)x * p.out_px;
        if (p.out_split) msr_store_split4_dev(o, q * 4, r.x, r.y, r.z, r.w);
        else *reinterpret_cast<float4*>(o + q * 4) = r;
    }
}

hipError_t launch_norm_act(const NormActParams& p, hipStream_t s) {
    if (p.C % 4) return hipErrorInvalidValue;
    long total = (long)p.B * p.H * p.W * (p.C / 4);
    long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    norm_act_kernel<<<(int)blocks, 256, 0, s>>>(p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// dense for M = B <= 16 rows: weight streaming, split over K.  W [K, N] row-major: a thread owns 4
// consecutive columns (16-byte loads, a wave covers 1 KiB of one weight row), x chunk staged in LDS.
//   encoder heads: Dense(256) x2 on the 131072-wide flatten (networks.py:31-33)  -> K-split
//   generator:     Dense(16*sw*sw*64) on the latent (networks.py:41)               -> N-parallel
// ------------------------------------------------------------------------------------------------
static constexpr int DENSE_KCH = 256;   // largest K chunk (LDS staging of x)
static constexpr int DENSE_MAXB = 16;

static int dense_kch(int K, int N) {
    // enough workgroups to pull the weight matrix at HBM rate: >= 1024 blocks of 128 threads, 8 loads in flight each
    const int gx = (N / 4 + 127) / 128;
    int kch = DENSE_KCH;
    while (kch > 16 && (long)gx * ((K + kch - 1) / kch) < 1024) kch >>= 1;
    return kch;
}
size_t dense_partial_floats(int B, int K, int N) {
    const int kch = dense_kch(K, N);
    return (size_t)((K + kch - 1) / kch) * B * N;
}

// NB = the batch rounded up to 1 / 2 / 4 / 8 / 16 rows: the first version always multiplied 16 rows, so a B = 1 call spent
// 16x the FMAs and LDS reads its one row needs and the 268 MB encoder-head matrix streamed at 2.6 TB/s (VALU-bound).
template <int NB>
__global__ void __launch_bounds__(128) dense_partial_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                            float* __restrict__ partial, int B, int K, int N,
                                                            int kch) {
    // x chunk as [k][NB rows]: the row values of one k are broadcast LDS reads
    __shared__ __attribute__((aligned(16))) float xs[DENSE_KCH * NB];
    const int col = (blockIdx.x * 128 + threadIdx.x) * 4;
    const int k0 = blockIdx.y * kch;
    const int kn = min(kch, K - k0);
    for (int i = threadIdx.x; i < NB * kch; i += 128) {
        const int b = i / kch, k = i % kch;
        xs[k * NB + b] = (b < B && k < kn) ? x[(size_t)b * K + k0 + k] : 0.f;
    }
    __syncthreads();
    if (col >= N) return;
    float4 acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* wp = W + (size_t)k0 * N + col;
    auto fma_row = [&](int k, const float4 w) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const float xr = xs[k * NB + b];
            float4& a = acc[b];
            a.x = fmaf(xr, w.x, a.x); a.y = fmaf(xr, w.y, a.y);
            a.z = fmaf(xr, w.z, a.z); a.w = fmaf(xr, w.w, a.w);
        }
    };
    // Two register buffers of 8 weight rows: the next 8 rows are requested before the current 8 are consumed, so a wave
    // always has 8-16 KiB on its way (a single unrolled group left the memory pipe idle while it multiplied: 268 MB of
    // encoder-head weights streamed at 3 TB/s at B = 8).  Same additions in the same order.
    float4 wa[8], wb[8];
    int k = 0;
    if (kn >= 16) {
#pragma unroll
        for (int u = 0; u < 8; ++u) wa[u] = *reinterpret_cast<const float4*>(wp + (size_t)u * N);
        for (; k + 16 <= kn; k += 16) {
#pragma unroll
            for (int u = 0; u < 8; ++u) wb[u] = *reinterpret_cast<const float4*>(wp + (size_t)(k + 8 + u) * N);
#pragma unroll
            for (int u = 0; u < 8; ++u) fma_row(k + u, wa[u]);
            if (k + 32 <= kn) {
#pragma unroll
                for (int u = 0; u < 8; ++u) wa[u] = *reinterpret_cast<const float4*>(wp + (size_t)(k + 16 + u) * N);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) fma_row(k + 8 + u, wb[u]);
        }
    }
    for (; k < kn; ++k) fma_row(k, *reinterpret_cast<const float4*>(wp + (size_t)k * N));
#pragma unroll
    for (int b = 0; b < NB; ++b)
        if (b < B) *reinterpret_cast<float4*>(partial + ((size_t)blockIdx.y * B + b) * N + col) = acc[b];
}

// 16 outputs x 16 split slots per workgroup; fixed-order combine -> deterministic
__global__ void __launch_bounds__(256) dense_final_kernel(const float* __restrict__ partial,
                                                          const float* __restrict__ bias, float* __restrict__ y,
                                                          int splits, int B, int N) {
    __shared__ float red[16][16];
    const int lane = threadIdx.x & 15, slot = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + lane;
    float s = 0.f;
    if (i < B * N) {
#pragma unroll 8      // the loads of 8 slots in flight, the adds in the same order as before
        for (int k = slot; k < splits; k += 16) s += partial[(size_t)k * B * N + i];
    }
    red[slot][lane] = s;
    __syncthreads();
    if (slot == 0 && i < B * N) {
        for (int k = 1; k < 16; ++k) s += red[k][lane];
        y[i] = s + (bias ? bias[i % N] : 0.f);
    }
}

// few K chunks (the generator's Dense: 8 at S = 512), many outputs: one thread per output, chunks added in order
// (dense_final_kernel's 16 outputs per workgroup meant 32768 workgroups for 0.5 M outputs: 14.6 us for 16 MB)
__global__ void __launch_bounds__(256) dense_final_flat_kernel(const float* __restrict__ partial,
                                                               const float* __restrict__ bias, float* __restrict__ y,
                                                               int splits, int B, int N) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * N) return;
    float s = 0.f;
#pragma unroll 8
    for (int k = 0; k < splits; ++k) s += partial[(size_t)k * B * N + i];
    y[i] = s + (bias ? bias[i % N] : 0.f);
}

hipError_t launch_dense(const float* x, const float* W, const float* bias, float* partial, float* y, int B, int K,
                        int N, hipStream_t s) {
    if (B > DENSE_MAXB || N % 4) return hipErrorInvalidValue;
    const int kch = dense_kch(K, N);
    const int splits = (K + kch - 1) / kch;
    const dim3 grid((N / 4 + 127) / 128, splits);
    if (B <= 1) dense_partial_kernel<1><<<grid, 128, 0, s>>>(x, W, partial, B, K, N, kch);
    else if (B <= 2) dense_partial_kernel<2><<<grid, 128, 0, s>>>(x, W, partial, B, K, N, kch);
    else if (B <= 4) dense_partial_kernel<4><<<grid, 128, 0, s>>>(x, W, partial, B, K, N, kch);
    else if (B <= 8) dense_partial_kernel<8><<<grid, 128, 0, s>>>(x, W, partial, B, K, N, kch);
    else dense_partial_kernel<16><<<grid, 128, 0, s>>>(x, W, partial, B, K, N, kch);
    if (splits <= 16 && (long)B * N >= 65536) dense_final_flat_kernel<<<(B * N + 255) / 256, 256, 0, s>>>(partial, bias, y, splits, B, N);
    else dense_final_kernel<<<(B * N + 15) / 16, 256, 0, s>>>(partial, bias, y, splits, B, N);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
__global__ void latent_kernel(const float* __restrict__ mv, const float* __restrict__ eps, float* __restrict__ z,
                              int B, int L, int use_sampler) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * L) return;
    const int b = i / L, k = i % L;
    const float m = mv[(size_t)b * 2 * L + k], v = mv[(size_t)b * 2 * L + L + k];
    z[i] = use_sampler ? m + expf(0.5f * v) * eps[i] : m + v;
}

hipError_t launch_latent(const float* mv, const float* eps, float* z, int B, int L, int use_sampler, hipStream_t s) {
    latent_kernel<<<(B * L + 255) / 256, 256, 0, s>>>(mv, eps, z, B, L, use_sampler);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// head: leaky_relu(x) -> UpSampling2D(2) -> Conv2D(1, 4, 'same') [-> tanh]  (networks.py:54-56).
// Conv over the nearest-upsampled tensor == per-output-parity 3x3 conv over the half-resolution tensor
// with summed ("effective") taps; TF SAME for k=4 pads 1 before / 2 after, and out-of-range in the
// up-sampled image is exactly out-of-range at half resolution, so zero padding carries over.
//   weff [py][px][dy+1][dx+1][C]  (zero where a tap does not exist)
// One thread = one half-resolution pixel = a 2x2 block of outputs.  A workgroup stages its 16 x 16 pixel tile plus
// the one-pixel halo through LDS 16 channels at a time (leaky-relu applied once, on the way in; out-of-range
// pixels are zero) and every thread accumulates its 25 live taps from there; the tap weights are wave-uniform
// (scalar loads).  HBM-bound: the half-resolution tensor is read once (x 324/256 for the halo).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) head_kernel(const float* __restrict__ x, const float* __restrict__ weff,
                                                   float bias, float* __restrict__ out, int B, int r, int C,
                                                   float slope, int tanh_out, int x_py, int x_pb) {
    constexpr int HW = 18, HP = HW * HW, CH = 16, PITCH = CH + 4;
    __shared__ __attribute__((aligned(16))) float tile[HP * PITCH];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int tiles = r >> 4;
    int t = blockIdx.x;
    const int tx0 = (t % tiles) << 4;
    t /= tiles;
    const int ty0 = (t % tiles) << 4;
    const int b = t / tiles;
    const float* xb = x + (size_t)b * x_pb;
    float o00 = 0.f, o01 = 0.f, o10 = 0.f, o11 = 0.f;
    // staging items of this thread (6 x 16 bytes per 16-channel chunk), fetched one chunk ahead into registers
    constexpr int ITEMS = (HP * (CH / 4) + 255) / 256;
    int goff[ITEMS], loff[ITEMS];
    float4 pre[ITEMS];
#pragma unroll
    for (int q = 0; q < ITEMS; ++q) {
        const int it = threadIdx.x + q * 256;
        const int hp = it >> 2, seg = it & 3;
        const int hy = hp / HW, hx = hp - hy * HW;
        const int yy = ty0 + hy - 1, xx = tx0 + hx - 1;
        const bool ok = it < HP * (CH / 4) && yy >= 0 && yy < r && xx >= 0 && xx < r;
        goff[q] = ok ? yy * x_py + xx * C + seg * 4 : -1;
        loff[q] = it < HP * (CH / 4) ? hp * PITCH + seg * 4 : -1;
    }
#pragma unroll
    for (int q = 0; q < ITEMS; ++q)
        pre[q] = goff[q] >= 0 ? *reinterpret_cast<const float4*>(xb + goff[q]) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int cc = 0; cc < C; cc += CH) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < ITEMS; ++q) {
            float4 v = pre[q];
            v.x = v.x >= 0.f ? v.x : v.x * slope; v.y = v.y >= 0.f ? v.y : v.y * slope;
            v.z = v.z >= 0.f ? v.z : v.z * slope; v.w = v.w >= 0.f ? v.w : v.w * slope;
            if (loff[q] >= 0) *reinterpret_cast<float4*>(tile + loff[q]) = v;
        }
        __syncthreads();
        if (cc + CH < C) {
#pragma unroll
            for (int q = 0; q < ITEMS; ++q)
                pre[q] = goff[q] >= 0 ? *reinterpret_cast<const float4*>(xb + goff[q] + cc + CH) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const float* wc = weff + cc;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const float* tp = tile + ((ty + dy) * HW + tx + dx) * PITCH;
#pragma unroll
                for (int sg = 0; sg < CH / 4; ++sg) {
                    const float4 v = *reinterpret_cast<const float4*>(tp + sg * 4);
                    // parity 0 uses offsets {-1,0,+1} (dy, dx = 0..2), parity 1 uses {0,+1} (1..2)
                    const float* w00 = wc + (size_t)((0 * 3 + dy) * 3 + dx) * C + sg * 4;
                    const float* w01 = wc + (size_t)((1 * 3 + dy) * 3 + dx) * C + sg * 4;
                    const float* w10 = wc + (size_t)((2 * 3 + dy) * 3 + dx) * C + sg * 4;
                    const float* w11 = wc + (size_t)((3 * 3 + dy) * 3 + dx) * C + sg * 4;
                    o00 = fmaf(v.w, w00[3], fmaf(v.z, w00[2], fmaf(v.y, w00[1], fmaf(v.x, w00[0], o00))));
                    if (dx > 0) o01 = fmaf(v.w, w01[3], fmaf(v.z, w01[2], fmaf(v.y, w01[1], fmaf(v.x, w01[0], o01))));
                    if (dy > 0) o10 = fmaf(v.w, w10[3], fmaf(v.z, w10[2], fmaf(v.y, w10[1], fmaf(v.x, w10[0], o10))));
                    if (dy > 0 && dx > 0)
                        o11 = fmaf(v.w, w11[3], fmaf(v.z, w11[2], fmaf(v.y, w11[1], fmaf(v.x, w11[0], o11))));
                }
            }
    }
    o00 += bias; o01 += bias; o10 += bias; o11 += bias;
    if (tanh_out) { o00 = tanhf(o00); o01 = tanhf(o01); o10 = tanhf(o10); o11 = tanhf(o11); }
    const int S2 = 2 * r;
    float* ob = out + ((size_t)b * S2 + 2 * (ty0 + ty)) * S2 + 2 * (tx0 + tx);
    *reinterpret_cast<float2*>(ob) = make_float2(o00, o01);
    *reinterpret_cast<float2*>(ob + S2) = make_float2(o10, o11);
}

hipError_t launch_head(const float* x, const float* weff, float bias, float* out, int B, int r, int C, float slope,
                       int tanh_out, int x_py, int x_pb, hipStream_t s) {
    if (x_py <= 0) { x_py = r * C; x_pb = r * r * C; }
    if (r % 16 || C % 16) return hipErrorInvalidValue;
    const int blocks = B * (r / 16) * (r / 16);
    head_kernel<<<blocks, 256, 0, s>>>(x, weff, bias, out, B, r, C, slope, tanh_out, x_py, x_pb);
    return hipGetLastError();
}

}  // namespace msr
