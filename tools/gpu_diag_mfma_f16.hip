// Diagnostic (not a pytest): v_mfma_f32_32x32x16_f16 on gfx950 — operand map (which k a lane's 8 halves are) and whether
// f16 DENORMAL inputs are honoured or flushed.
//   hipcc --offload-arch=gfx950 -O2 tools/gpu_diag_mfma_f16.hip -o /tmp/diag_mfma && /tmp/diag_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// A[m][k] = (k == m % 16) ? av : 0,  B[k][n] = 100 k + n  ->  D[m][n] = av * (100 (m % 16) + n)
__global__ void k(float* out, float av) {
    const int lane = threadIdx.x, mn = lane & 31, kh = lane >> 5;
    f16x8 a, b;
    for (int t = 0; t < 8; ++t) {
        const int kk = 8 * kh + t;                  // hypothesis: lane (mn, kh) holds k = 8 kh .. 8 kh + 7
        a[t] = (_Float16)((kk == mn % 16) ? av : 0.f);
        b[t] = (_Float16)(float)(100 * kk + mn);
    }
    f32x16 c = {};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) out[lane * 16 + r] = c[r];
}
int main() {
    float* d; hipMalloc(&d, 64 * 16 * 4);
    float h[64 * 16];
    for (float av : {1.0f, 6.0e-6f}) {              // 6e-6 is an f16 denormal (min normal 6.1e-5)
        k<<<1, 64>>>(d, av);
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        int bad = 0;
        const float aq = (float)(_Float16)av;
        for (int lane = 0; lane < 64; ++lane)
            for (int r = 0; r < 16; ++r) {
                const int n = lane & 31, m = 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);      // the 32x32 result map
                const float want = aq * (100 * (m % 16) + n);
                if (fabsf(h[lane * 16 + r] - want) > 1e-3f * fabsf(want) + 1e-12f) ++bad;
            }
        printf("a = %g (as f16 %g): %d of 1024 results off;  D[1][0] = %g (want %g), D[17][5] = %g (want %g)\n", av, aq, bad,
               h[0 * 16 + 1], aq * 100.f, h[5 * 16 + 8 + 1], aq * 105.f);
    }
    return 0;
}
