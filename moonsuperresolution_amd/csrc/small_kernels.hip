// Memory-bound and tiny kernels of the SPADE generator: mask embedding / first encoder conv, moments,
// instance-norm apply, dense layers, latent sampler and the fused up-sample + leaky-relu + 4x4 conv head.
// All are HBM-bound (or launch-bound) on MI355X: 16-byte coalesced accesses, 64-wide wave reductions.
#include "kernels.h"

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
            if (p.out_split) msr_store_split4_dev(o + (size_t)i * p.out_px, q * 4, acc.x, acc.y, acc.z, acc.w);
            else *reinterpret_cast<float4*>(o + (size_t)i * p.out_px + q * 4) = acc;
        }
    }
}

template <int COUT>
static hipError_t launch_smallcin_t(const SmallCinParams& p, hipStream_t s) {
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
static constexpr int MOM_PIX_PER_CHUNK = 128;   // small chunks: thousands of workgroups keep enough bytes in flight

int moments_chunks(int P) { return (P + MOM_PIX_PER_CHUNK - 1) / MOM_PIX_PER_CHUNK; }

__global__ void __launch_bounds__(256) moments_partial_kernel(const float* __restrict__ x, int P, int C,
                                                              double* __restrict__ partial) {
    // partial layout: [G][chunks][C][2]
    __shared__ double red[256 * 8];
    const int quads = C / 4;
    const int chunk = blockIdx.x, g = blockIdx.y, chunks = gridDim.x;
    const int p0 = chunk * MOM_PIX_PER_CHUNK;
    const int p1 = min(P, p0 + MOM_PIX_PER_CHUNK);
    const float* xg = x + (size_t)g * P * C;
    for (int qb = 0; qb < quads; qb += 256) {
        // layout A (quads >= 256): every thread one quad, all pixels.  layout B: several pixel slots per quad.
        const int tq = quads >= 256 ? 256 : quads;     // threads along channel quads
        const int slots = 256 / tq;                    // pixel slots
        const int q = qb + threadIdx.x % tq;
        const int slot = threadIdx.x / tq;
        double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
        if (q < quads) {
            // 4 independent 16-byte loads in flight per thread; short fp32 runs flushed into fp64 accumulators
            const float* base = xg + q * 4;
            int pix = p0 + slot;
            for (; pix + 3 * slots < p1; pix += 4 * slots) {
                const float4 v0 = *reinterpret_cast<const float4*>(base + (size_t)pix * C);
                const float4 v1 = *reinterpret_cast<const float4*>(base + (size_t)(pix + slots) * C);
                const float4 v2 = *reinterpret_cast<const float4*>(base + (size_t)(pix + 2 * slots) * C);
                const float4 v3 = *reinterpret_cast<const float4*>(base + (size_t)(pix + 3 * slots) * C);
                s[0] += (double)((v0.x + v1.x) + (v2.x + v3.x)); s[1] += (double)((v0.y + v1.y) + (v2.y + v3.y));
                s[2] += (double)((v0.z + v1.z) + (v2.z + v3.z)); s[3] += (double)((v0.w + v1.w) + (v2.w + v3.w));
                ss[0] += (double)((v0.x * v0.x + v1.x * v1.x) + (v2.x * v2.x + v3.x * v3.x));
                ss[1] += (double)((v0.y * v0.y + v1.y * v1.y) + (v2.y * v2.y + v3.y * v3.y));
                ss[2] += (double)((v0.z * v0.z + v1.z * v1.z) + (v2.z * v2.z + v3.z * v3.z));
                ss[3] += (double)((v0.w * v0.w + v1.w * v1.w) + (v2.w * v2.w + v3.w * v3.w));
            }
            for (; pix < p1; pix += slots) {
                const float4 v = *reinterpret_cast<const float4*>(base + (size_t)pix * C);
                s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
                ss[0] += (double)v.x * v.x; ss[1] += (double)v.y * v.y;
                ss[2] += (double)v.z * v.z; ss[3] += (double)v.w * v.w;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { red[threadIdx.x * 8 + k] = s[k]; red[threadIdx.x * 8 + 4 + k] = ss[k]; }
        __syncthreads();
        if (slot == 0 && q < quads) {
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

hipError_t launch_moments(const float* x, int G, int P, int C, float eps, double* partial, float* mean, float* stdv,
                          hipStream_t s) {
    if (C % 32) return hipErrorInvalidValue;
    const int chunks = moments_chunks(P);
    moments_partial_kernel<<<dim3(chunks, G), 256, 0, s>>>(x, P, C, partial);
    moments_final_kernel<<<G * (C / 32), 1024, 0, s>>>(partial, G, chunks, C, P, eps, mean, stdv);
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
    // enough workgroups to pull the weight matrix at HBM rate: >= 512 blocks of 128 threads
    const int gx = (N / 4 + 127) / 128;
    int kch = DENSE_KCH;
    while (kch > 16 && (long)gx * ((K + kch - 1) / kch) < 512) kch >>= 1;
    return kch;
}
size_t dense_partial_floats(int B, int K, int N) {
    const int kch = dense_kch(K, N);
    return (size_t)((K + kch - 1) / kch) * B * N;
}

__global__ void __launch_bounds__(128) dense_partial_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                            float* __restrict__ partial, int B, int K, int N,
                                                            int kch) {
    __shared__ float xs[DENSE_MAXB][DENSE_KCH];
    const int col = (blockIdx.x * 128 + threadIdx.x) * 4;
    const int k0 = blockIdx.y * kch;
    const int kn = min(kch, K - k0);
    for (int i = threadIdx.x; i < B * kch; i += 128) {
        const int b = i / kch, k = i % kch;
        xs[b][k] = k < kn ? x[(size_t)b * K + k0 + k] : 0.f;
    }
    __syncthreads();
    if (col >= N) return;
    float4 acc[DENSE_MAXB];
#pragma unroll
    for (int b = 0; b < DENSE_MAXB; ++b) acc[b] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* wp = W + (size_t)k0 * N + col;
#pragma unroll 4
    for (int k = 0; k < kn; ++k) {
        const float4 w = *reinterpret_cast<const float4*>(wp + (size_t)k * N);
#pragma unroll
        for (int b = 0; b < DENSE_MAXB; ++b) {
            if (b < B) {
                const float xv = xs[b][k];
                acc[b].x += xv * w.x; acc[b].y += xv * w.y; acc[b].z += xv * w.z; acc[b].w += xv * w.w;
            }
        }
    }
#pragma unroll
    for (int b = 0; b < DENSE_MAXB; ++b)
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
    if (i < B * N)
        for (int k = slot; k < splits; k += 16) s += partial[(size_t)k * B * N + i];
    red[slot][lane] = s;
    __syncthreads();
    if (slot == 0 && i < B * N) {
        for (int k = 1; k < 16; ++k) s += red[k][lane];
        y[i] = s + (bias ? bias[i % N] : 0.f);
    }
}

hipError_t launch_dense(const float* x, const float* W, const float* bias, float* partial, float* y, int B, int K,
                        int N, hipStream_t s) {
    if (B > DENSE_MAXB || N % 4) return hipErrorInvalidValue;
    const int kch = dense_kch(K, N);
    const int splits = (K + kch - 1) / kch;
    dense_partial_kernel<<<dim3((N / 4 + 127) / 128, splits), 128, 0, s>>>(x, W, partial, B, K, N, kch);
    dense_final_kernel<<<(B * N + 15) / 16, 256, 0, s>>>(partial, bias, y, splits, B, N);
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
// A wave owns one half-resolution row segment; lanes split the C channels (2 each for C=128), keep the
// 25 live effective taps in registers, slide a 3x3 window along x and reduce across the wave.
// ------------------------------------------------------------------------------------------------
template <int CPL>   // channels per lane, C = 64 * CPL
__global__ void __launch_bounds__(256) head_kernel(const float* __restrict__ x, const float* __restrict__ weff,
                                                   float bias, float* __restrict__ out, int B, int r, float slope,
                                                   int tanh_out, int seg, int x_py, int x_pb) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    constexpr int C = 64 * CPL;
    const int segs = (r + seg - 1) / seg;
    const long item = (long)blockIdx.x * 4 + wave;   // (b, i, segment)
    if (item >= (long)B * r * segs) return;
    const int sg = (int)(item % segs);
    const int i = (int)((item / segs) % r);
    const int b = (int)(item / ((long)segs * r));
    float w[2][2][3][3][CPL];
#pragma unroll
    for (int a = 0; a < 36; ++a)
#pragma unroll
        for (int c = 0; c < CPL; ++c) (&w[0][0][0][0][0])[a * CPL + c] = weff[(size_t)a * C + lane * CPL + c];

    const float* xb = x + (size_t)b * x_pb;   // pixel (0,0) of image b; rows x_py apart (dense or zero-bordered)
    auto load = [&](int yy, int xx, float (&v)[CPL]) {
        const bool ok = yy >= 0 && yy < r && xx >= 0 && xx < r;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            float t = 0.f;
            if (ok) t = xb[(size_t)yy * x_py + (size_t)xx * C + lane * CPL + c];
            v[c] = t >= 0.f ? t : t * slope;
        }
    };
    const int j0 = sg * seg, j1 = min(r, j0 + seg);
    float win[3][3][CPL];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        load(i + dy - 1, j0 - 1, win[dy][1]);
        load(i + dy - 1, j0, win[dy][2]);
    }
    const int S2 = 2 * r;
    for (int j = j0; j < j1; ++j) {
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
#pragma unroll
            for (int c = 0; c < CPL; ++c) { win[dy][0][c] = win[dy][1][c]; win[dy][1][c] = win[dy][2][c]; }
            load(i + dy - 1, j + 1, win[dy][2]);
        }
        float o[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int px = 0; px < 2; ++px)
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        // parity 0 uses offsets {-1,0,+1}, parity 1 uses {0,+1}
                        if ((py == 1 && dy == 0) || (px == 1 && dx == 0)) continue;
#pragma unroll
                        for (int c = 0; c < CPL; ++c) o[py][px] += win[dy][dx][c] * w[py][px][dy][dx][c];
                    }
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int px = 0; px < 2; ++px) o[py][px] = wave_sum(o[py][px]);
        if (lane < 4) {
            const int py = lane >> 1, px = lane & 1;
            float v = (py ? (px ? o[1][1] : o[1][0]) : (px ? o[0][1] : o[0][0])) + bias;
            if (tanh_out) v = tanhf(v);
            out[((size_t)b * S2 + 2 * i + py) * S2 + 2 * j + px] = v;
        }
    }
}

hipError_t launch_head(const float* x, const float* weff, float bias, float* out, int B, int r, int C, float slope,
                       int tanh_out, int x_py, int x_pb, hipStream_t s) {
    if (x_py <= 0) { x_py = r * C; x_pb = r * r * C; }
    const int seg = r >= 64 ? 32 : r;
    const int segs = (r + seg - 1) / seg;
    const long items = (long)B * r * segs;
    const int blocks = (int)((items + 3) / 4);
    if (C == 128) head_kernel<2><<<blocks, 256, 0, s>>>(x, weff, bias, out, B, r, slope, tanh_out, seg, x_py, x_pb);
    else if (C == 64) head_kernel<1><<<blocks, 256, 0, s>>>(x, weff, bias, out, B, r, slope, tanh_out, seg, x_py, x_pb);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace msr
