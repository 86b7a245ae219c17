// conv_direct: generic direct convolution / transposed convolution for the pix2pix generator
// (pix2pix.py:65-108) — BASELINE config 1, the "plumbing / parity" configuration (11.9 GFLOP per patch).
// Lanes run along output channels (coalesced HWIO weight rows, wave-uniform input reads); the folded
// BatchNormalization affine, the skip-connection concat (two input pointers) and the activation are fused.
// It serves down1 (2 input channels: no K to tile); every other pix2pix layer runs on conv_igemm (4x4 s2 convs,
// transposed convs as four parity 2x2 convs) and the 1-channel tanh output on the head kernel.
#include "kernels.h"

namespace msr {

// One thread = 4 consecutive output channels of one pixel (float4 weight rows, float4 store); Cout % 4 == 0.
__global__ void __launch_bounds__(256) conv_direct_kernel(const DirectConvParams p) {
    const int cin = p.c0 + p.c1;
    const int cq = p.Cout >> 2;
    const long total = (long)p.B * p.Hout * p.Wout * cq;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int co = (int)(i % cq) * 4;
        long pix = i / cq;
        const int x = (int)(pix % p.Wout);
        pix /= p.Wout;
        const int y = (int)(pix % p.Hout);
        const int b = (int)(pix / p.Hout);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int kh = 0; kh < p.KH; ++kh) {
            int iy;
            if (p.transposed) {
                const int t = y + p.pad - kh;
                if (t < 0 || (t % p.stride)) continue;
                iy = t / p.stride;
            } else {
                iy = y * p.stride - p.pad + kh;
            }
            if (iy < 0 || iy >= p.Hin) continue;
            for (int kw = 0; kw < p.KW; ++kw) {
                int ix;
                if (p.transposed) {
                    const int t = x + p.pad - kw;
                    if (t < 0 || (t % p.stride)) continue;
                    ix = t / p.stride;
                } else {
                    ix = x * p.stride - p.pad + kw;
                }
                if (ix < 0 || ix >= p.Win) continue;
                const float* w = p.w + ((size_t)(kh * p.KW + kw) * cin) * p.Cout + co;
                const float* a0 = p.in0 + (size_t)b * p.in_pb + (size_t)iy * p.in_py + (size_t)ix * p.in_px;
                for (int ci = 0; ci < p.c0; ++ci) {
                    const float a = a0[ci];
                    const float4 wv = *reinterpret_cast<const float4*>(w + (size_t)ci * p.Cout);
                    acc.x += a * wv.x; acc.y += a * wv.y; acc.z += a * wv.z; acc.w += a * wv.w;
                }
                if (p.c1) {
                    const float* a1 = p.in1 + (size_t)b * p.in_pb + (size_t)iy * p.in_py + (size_t)ix * p.in_px;
                    const float* w1 = w + (size_t)p.c0 * p.Cout;
                    for (int ci = 0; ci < p.c1; ++ci) {
                        const float a = a1[ci];
                        const float4 wv = *reinterpret_cast<const float4*>(w1 + (size_t)ci * p.Cout);
                        acc.x += a * wv.x; acc.y += a * wv.y; acc.z += a * wv.z; acc.w += a * wv.w;
                    }
                }
            }
        }
        float v[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (p.scale) v[k] *= p.scale[co + k];
            if (p.shift) v[k] += p.shift[co + k];
            if (p.act == 1) v[k] = fmaxf(v[k], 0.f);
            else if (p.act == 2) v[k] = v[k] >= 0.f ? v[k] : v[k] * p.slope;
            else if (p.act == 3) v[k] = tanhf(v[k]);
        }
        *reinterpret_cast<float4*>(p.out + (size_t)b * p.out_pb + (size_t)y * p.out_py + (size_t)x * p.out_px + co) =
            make_float4(v[0], v[1], v[2], v[3]);
    }
}

hipError_t launch_conv_direct(const DirectConvParams& p, hipStream_t s) {
    if (p.Cout % 4 || p.out_px % 4 || p.out_py % 4 || p.out_pb % 4) return hipErrorInvalidValue;
    const long total = (long)p.B * p.Hout * p.Wout * (p.Cout / 4);
    long blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    conv_direct_kernel<<<(int)blocks, 256, 0, s>>>(p);
    return hipGetLastError();
}

}  // namespace msr
