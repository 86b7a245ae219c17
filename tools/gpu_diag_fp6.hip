// Diagnostic (not a test): operand / scale map of v_mfma_scale_f32_16x16x128_f8f6f4 with fp6 e2m3 (or fp4 e2m1) operands on
// gfx950.  One wave, random operand bits; the host evaluates the product under candidate maps and reports which one the
// hardware follows.   hipcc --offload-arch=gfx950 -O2 tools/gpu_diag_fp6.hip -o build/diag_fp6 && build/diag_fp6
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int FMT>   // 2 = fp6 e2m3, 4 = fp4 e2m1, 0 = fp8 e4m3
__global__ void one_mfma(const int* a, const int* b, const int* sa, const int* sb, float* d) {
    const int l = threadIdx.x;
    i32x8 A, B;
    for (int k = 0; k < 8; ++k) { A[k] = a[l * 8 + k]; B[k] = b[l * 8 + k]; }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, acc, FMT, FMT, 0, sa[l], 0, sb[l]);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = acc[r];
}

static float dec_e2m3(int c) { int s = (c >> 5) & 1, e = (c >> 3) & 3, m = c & 7; float v = e ? ldexpf(1.f + m / 8.f, e - 1) : m / 8.f; return s ? -v : v; }
static float dec_e2m1(int c) { int s = (c >> 3) & 1, e = (c >> 1) & 3, m = c & 1; float v = e ? ldexpf(1.f + m / 2.f, e - 1) : m / 2.f; return s ? -v : v; }
static float dec_e4m3(int c) { int s = (c >> 7) & 1, e = (c >> 3) & 15, m = c & 7; float v = e ? ldexpf(1.f + m / 8.f, e - 7) : ldexpf(m / 8.f, -6); if (e == 15 && m == 7) v = 0.f; return s ? -v : v; }

// element e (0..31) of a lane's operand under the "little-endian bit string" packing
static float elem(const int* regs, int e, int fmt) {
    const int bits = fmt == 2 ? 6 : fmt == 4 ? 4 : 8;
    const int pos = e * bits;
    unsigned long long w = (unsigned)regs[pos / 32];
    if (pos / 32 + 1 < 8) w |= (unsigned long long)(unsigned)regs[pos / 32 + 1] << 32;
    const int c = (int)((w >> (pos % 32)) & ((1u << bits) - 1));
    return fmt == 2 ? dec_e2m3(c) : fmt == 4 ? dec_e2m1(c) : dec_e4m3(c);
}

template <int FMT>
static void run(const char* name) {
    std::vector<int> a(64 * 8), b(64 * 8), sa(64), sb(64);
    std::vector<float> d(256);
    srand(7 + FMT);
    for (auto& v : a) v = (rand() << 16) ^ rand();
    for (auto& v : b) v = (rand() << 16) ^ rand();
    if (FMT == 0) {   // avoid fp8 NaN codes
        for (auto* vec : {&a, &b}) for (auto& v : *vec) { unsigned u = (unsigned)v; for (int k = 0; k < 4; ++k) if (((u >> (8 * k)) & 0x7F) == 0x7F) u &= ~(1u << (8 * k)); v = (int)u; }
    }
    int *da, *db, *dsa, *dsb; float* dd;
    hipMalloc(&da, a.size() * 4); hipMalloc(&db, b.size() * 4); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dd, 1024);
    hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice);
    // partial sums P[i][j][g][h]: lane group g, half h (elements 16h..16h+15), pairing (g, e) <-> (g, e)
    std::vector<double> P(16 * 16 * 4 * 2, 0.0);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int g = 0; g < 4; ++g) for (int e = 0; e < 32; ++e)
        P[((i * 16 + j) * 4 + g) * 2 + e / 16] += (double)elem(&a[(g * 16 + i) * 8], e, FMT) * elem(&b[(g * 16 + j) * 8], e, FMT);
    printf("== %s\n", name);
    for (int trial = 0; trial < 9; ++trial) {
        // trial 0: all scales 1.  trials 1-4: A-scale of lane group t-1 doubled.  trials 5-8: B-scale of lane group t-5 doubled.
        for (int l = 0; l < 64; ++l) { sa[l] = 127; sb[l] = 127; }
        if (trial >= 1 && trial <= 4) for (int l = 0; l < 16; ++l) sa[(trial - 1) * 16 + l] = 128;
        if (trial >= 5) for (int l = 0; l < 16; ++l) sb[(trial - 5) * 16 + l] = 128;
        hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
        one_mfma<FMT><<<1, 64>>>(da, db, dsa, dsb, dd);
        hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
        // which (g, h) blocks were doubled?  least squares is overkill: test the 8 single-block hypotheses and the plain one
        double best = 1e30; int bg = -1, bh = -1;
        for (int hg = -1; hg < 4; ++hg) for (int hh = 0; hh < (hg < 0 ? 1 : 2); ++hh) {
            double err = 0, mag = 0;
            for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
                const int i = 4 * (l / 16) + r, j = l % 16;      // D[row i of A][column j of B]
                double s = 0;
                for (int g = 0; g < 4; ++g) for (int h = 0; h < 2; ++h) s += P[((i * 16 + j) * 4 + g) * 2 + h] * ((g == hg && h == hh) ? 2.0 : 1.0);
                err = fmax(err, fabs(s - d[l * 4 + r])); mag = fmax(mag, fabs(s));
            }
            if (err / mag < best) { best = err / mag; bg = hg; bh = hh; }
        }
        // two-block hypotheses: (g pair {2q, 2q+1}, half h) as the fp8 map has it
        double best2 = 1e30; int q2 = -1, h2 = -1;
        for (int q = 0; q < 2; ++q) for (int hh = 0; hh < 2; ++hh) {
            double err = 0, mag = 0;
            for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
                const int i = 4 * (l / 16) + r, j = l % 16;
                double s = 0;
                for (int g = 0; g < 4; ++g) for (int h = 0; h < 2; ++h) s += P[((i * 16 + j) * 4 + g) * 2 + h] * ((g / 2 == q && h == hh) ? 2.0 : 1.0);
                err = fmax(err, fabs(s - d[l * 4 + r])); mag = fmax(mag, fabs(s));
            }
            if (err / mag < best2) { best2 = err / mag; q2 = q; h2 = hh; }
        }
        // whole-lane-group hypothesis: block = all 32 elements of lane group g
        double best3 = 1e30; int g3 = -1;
        for (int hg = 0; hg < 4; ++hg) {
            double err = 0, mag = 0;
            for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
                const int i = 4 * (l / 16) + r, j = l % 16;
                double s = 0;
                for (int g = 0; g < 4; ++g) for (int h = 0; h < 2; ++h) s += P[((i * 16 + j) * 4 + g) * 2 + h] * (g == hg ? 2.0 : 1.0);
                err = fmax(err, fabs(s - d[l * 4 + r])); mag = fmax(mag, fabs(s));
            }
            if (err / mag < best3) { best3 = err / mag; g3 = hg; }
        }
        printf("trial %d (%s): single block (g=%d,h=%d) err %.2e | pair (groups %d..%d, half %d) err %.2e | whole group %d err %.2e\n", trial,
               trial == 0 ? "unit scales" : trial <= 4 ? "A scale x2 in one lane group" : "B scale x2 in one lane group", bg, bh, best, 2 * q2,
               2 * q2 + 1, h2, best2, g3, best3);
    }
    hipFree(da); hipFree(db); hipFree(dsa); hipFree(dsb); hipFree(dd);
}

int main() {
    run<0>("fp8 e4m3 (the map the f16c kernels use: expect 'pair' hits)");
    run<2>("fp6 e2m3");
    run<4>("fp4 e2m1");
    return 0;
}
