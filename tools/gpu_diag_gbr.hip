// Diagnostic (not a pytest): semantics of the gfx950 instructions conv_gbr.hip's phase 1 relies on.
//   hipcc --offload-arch=gfx950 -O2 tools/gpu_diag_gbr.hip -o /tmp/diag_gbr && /tmp/diag_gbr
// 1. v_cvt_scalef32_2xpk16_fp6_f32(a[16], b[16], scale): element order of the 192-bit result and what `scale` does
// 2. v_permlane32_swap: which halves are exchanged
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v6i __attribute__((ext_vector_type(6)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

__global__ void k_cvt(const float* in, int* out, float scale) {
    v16f a, b;
    for (int i = 0; i < 16; ++i) { a[i] = in[i]; b[i] = in[16 + i]; }
    v6i r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a, b, scale);
    if (threadIdx.x == 0)
        for (int i = 0; i < 6; ++i) out[i] = r[i];
}
__global__ void k_swap(unsigned* out) {
    const unsigned x = 100 + threadIdx.x, y = 200 + threadIdx.x;
    v2u r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    out[threadIdx.x] = r[0];
    out[64 + threadIdx.x] = r[1];
}

static double e2m3(unsigned c) {
    const double s = (c & 32) ? -1.0 : 1.0;
    const unsigned e = (c >> 3) & 3, m = c & 7;
    return s * (e == 0 ? m / 8.0 : std::ldexp(1.0 + m / 8.0, (int)e - 1));
}

static void time_instructions();
int main() {
    float h_in[32];
    for (int i = 0; i < 32; ++i) h_in[i] = (i % 2 ? -1.f : 1.f) * 0.25f * (float)(i % 29);   // 0, -.25, .5, ... up to 7
    float* d_in; int* d_out; unsigned* d_sw;
    hipMalloc(&d_in, sizeof h_in); hipMalloc(&d_out, 6 * 4); hipMalloc(&d_sw, 128 * 4);
    hipMemcpy(d_in, h_in, sizeof h_in, hipMemcpyHostToDevice);
    for (float scale : {1.0f, 2.0f, 0.5f, 3.0f}) {
        k_cvt<<<1, 64>>>(d_in, d_out, scale);
        unsigned w[6];
        hipMemcpy(w, d_out, 24, hipMemcpyDeviceToHost);
        printf("scale %g: words %08x %08x %08x %08x %08x %08x\n  decoded (element e at bits 6e..6e+5):", scale, w[0], w[1], w[2], w[3], w[4], w[5]);
        for (int e = 0; e < 32; ++e) {
            const int bit = 6 * e;
            unsigned long long two = w[bit / 32];
            if (bit / 32 + 1 < 6) two |= (unsigned long long)w[bit / 32 + 1] << 32;
            printf(" %g", e2m3((unsigned)((two >> (bit % 32)) & 63)));
        }
        printf("\n  inputs:");
        for (int e = 0; e < 32; ++e) printf(" %g", h_in[e]);
        printf("\n");
    }
    k_swap<<<1, 64>>>(d_sw);
    unsigned s[128];
    hipMemcpy(s, d_sw, sizeof s, hipMemcpyDeviceToHost);
    printf("permlane32_swap(x = 100 + lane, y = 200 + lane):\n  r[0] lanes 0, 31, 32, 63: %u %u %u %u\n  r[1] lanes 0, 31, 32, 63: %u %u %u %u\n",
           s[0], s[31], s[32], s[63], s[64], s[95], s[96], s[127]);
    time_instructions();
    return 0;
}

// 3. issue cost of the converters and of v_permlane32_swap: 32 independent instructions between two s_memtime stamps
__global__ void k_time(const float* in, int* out, unsigned* cycles) {
    v16f a, b;
    for (int i = 0; i < 16; ++i) { a[i] = in[i] + threadIdx.x; b[i] = in[16 + i] - threadIdx.x; }
    v6i acc = {0, 0, 0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        v6i r;
        asm volatile("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, %3" : "=&v"(r) : "v"(a), "v"(b), "v"(1.0f + k));
        acc ^= r;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)");
    unsigned x = threadIdx.x, y = threadIdx.x * 3;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        v2u r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
        x = r[0] + k; y = r[1] ^ k;
    }
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)");
    if (threadIdx.x == 0) { cycles[0] = (unsigned)(t1 - t0); cycles[1] = (unsigned)(t2 - t1); }
    for (int i = 0; i < 6; ++i) out[threadIdx.x * 6 + i] = acc[i] + x + y;
}
static void time_instructions() {
    float h_in[32];
    for (int i = 0; i < 32; ++i) h_in[i] = 0.1f * i;
    float* d_in; int* d_out; unsigned* d_c;
    hipMalloc(&d_in, sizeof h_in); hipMalloc(&d_out, 64 * 6 * 4); hipMalloc(&d_c, 8);
    hipMemcpy(d_in, h_in, sizeof h_in, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep) k_time<<<1, 64>>>(d_in, d_out, d_c);
    unsigned c[2];
    hipMemcpy(c, d_c, 8, hipMemcpyDeviceToHost);
    printf("32 x v_cvt_scalef32_2xpk16_fp6_f32: %u cycles (%.1f each); 32 x dependent v_permlane32_swap: %u cycles (%.1f each)\n",
           c[0], c[0] / 32.0, c[1], c[1] / 32.0);
}
