// Diagnostic (not a pytest): what v_cvt_pk_fp8_f32 / v_cvt_scalef32_pk_fp8_f32 / v_cvt_f16_f32 return beyond their formats'
// range, with MODE.FP16_OVFL clear (default) and set (s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1).
//   hipcc --offload-arch=gfx950 -O2 tools/gpu_diag_fp8_ovfl.hip -o /tmp/diag_ovfl && /tmp/diag_ovfl
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s2_ __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, unsigned* out, int ovfl) {
    if (ovfl) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
    const float a = in[2 * threadIdx.x], b = in[2 * threadIdx.x + 1];
    unsigned w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
    s2_ l = {0, 0};
    l = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(l, a, b, 0x1p-11f, false);
    const _Float16 h = (_Float16)a;
    unsigned bw = 0;
    bw = __builtin_amdgcn_cvt_pk_bf8_f32(a, b, bw, false);
    out[4 * threadIdx.x] = w & 0xFFFF;
    out[4 * threadIdx.x + 1] = (unsigned)(unsigned short)l[0];
    out[4 * threadIdx.x + 2] = (unsigned)__builtin_bit_cast(unsigned short, h);
    out[4 * threadIdx.x + 3] = bw & 0xFFFF;
    if (ovfl) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 0");
}
int main() {
    const float h_in[8] = {448.f, 449.f, 480.f, 1.0e4f, -7.0e4f, 0.25f, 1.0e30f, -0.3f};
    float* d_in; unsigned* d_out;
    hipMalloc(&d_in, sizeof h_in); hipMalloc(&d_out, 16 * 4);
    hipMemcpy(d_in, h_in, sizeof h_in, hipMemcpyHostToDevice);
    for (int ovfl = 0; ovfl < 2; ++ovfl) {
        k<<<1, 4>>>(d_in, d_out, ovfl);
        unsigned o[16];
        hipMemcpy(o, d_out, sizeof o, hipMemcpyDeviceToHost);
        printf("FP16_OVFL = %d\n", ovfl);
        for (int t = 0; t < 4; ++t)
            printf("  (%g, %g): pk_fp8 %04x   scalef32(2^-11) pk_fp8 %04x   f16(a) %04x   pk_bf8 %04x\n", h_in[2 * t], h_in[2 * t + 1], o[4 * t],
                   o[4 * t + 1], o[4 * t + 2], o[4 * t + 3]);
    }
    return 0;
}
