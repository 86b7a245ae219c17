// Tiler / stitcher kernels: the NumPy half of the reference's driver (process_full_tiles.py) on the GPU.
// Compiled with -ffp-contract=off: every float op below mirrors one NumPy op of the reference, evaluated in
// the same type (float32 or float64) and the same order, so results are bit-exact with the CPU path.
#include "kernels.h"

namespace msr {

// ------------------------------------------------------------------------------------------------
// patch_stats: getPatch's validity test + normalize's four reductions for every patch of a tile
// (process_full_tiles.py:286-292, 307-309).  One workgroup per patch; min/max are exact.  NumPy's .min() / .max()
// return NaN as soon as one pixel is NaN (fminf / fmaxf would drop it), so a NaN flag per raster is reduced beside
// the extrema and turns both of that raster's figures into NaN: the whole patch then normalises to NaN like the
// reference's (a NaN pixel is not <= no_value, so the patch stays valid in both).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) patch_stats_kernel(const float* __restrict__ img, const float* __restrict__ dem,
                                                          int rows, int cols, const int* __restrict__ ox,
                                                          const int* __restrict__ oy, int S, float no_value,
                                                          uint8_t* __restrict__ valid, float* __restrict__ minmax) {
    __shared__ float red[4][4];
    __shared__ int bad[4];
    const int i = blockIdx.x;
    const int x0 = ox[i], y0 = oy[i];
    float imn = INFINITY, imx = -INFINITY, dmn = INFINITY, dmx = -INFINITY;
    int anybad = 0, inan = 0, dnan = 0;
    if (x0 < 0 || y0 < 0 || x0 + S > cols || y0 + S > rows) {
        anybad = 1;   // outside the canvas: not a patch the reference could cut
    } else {
        const int qpr = S / 4;   // float4 per patch row (S is a multiple of 64)
        const bool aligned = ((x0 & 3) == 0) && ((cols & 3) == 0);
        for (int e = threadIdx.x; e < S * qpr; e += 256) {
            const int y = e / qpr, q = e % qpr;
            const size_t off = (size_t)(y0 + y) * cols + x0 + q * 4;
            float4 a, d;
            if (aligned) {
                a = *reinterpret_cast<const float4*>(img + off);
                d = *reinterpret_cast<const float4*>(dem + off);
            } else {
                a = make_float4(img[off], img[off + 1], img[off + 2], img[off + 3]);
                d = make_float4(dem[off], dem[off + 1], dem[off + 2], dem[off + 3]);
            }
            imn = fminf(fminf(imn, a.x), fminf(fminf(a.y, a.z), a.w));
            imx = fmaxf(fmaxf(imx, a.x), fmaxf(fmaxf(a.y, a.z), a.w));
            dmn = fminf(fminf(dmn, d.x), fminf(fminf(d.y, d.z), d.w));
            dmx = fmaxf(fmaxf(dmx, d.x), fmaxf(fmaxf(d.y, d.z), d.w));
            anybad |= (a.x <= no_value) | (a.y <= no_value) | (a.z <= no_value) | (a.w <= no_value) |
                      (d.x <= no_value) | (d.y <= no_value) | (d.z <= no_value) | (d.w <= no_value);
            inan |= (a.x != a.x) | (a.y != a.y) | (a.z != a.z) | (a.w != a.w);
            dnan |= (d.x != d.x) | (d.y != d.y) | (d.z != d.z) | (d.w != d.w);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        imn = fminf(imn, __shfl_xor(imn, o));
        imx = fmaxf(imx, __shfl_xor(imx, o));
        dmn = fminf(dmn, __shfl_xor(dmn, o));
        dmx = fmaxf(dmx, __shfl_xor(dmx, o));
        anybad |= __shfl_xor(anybad, o);
        inan |= __shfl_xor(inan, o);
        dnan |= __shfl_xor(dnan, o);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[wave][0] = imn; red[wave][1] = imx; red[wave][2] = dmn; red[wave][3] = dmx;
        bad[wave] = anybad | (inan << 1) | (dnan << 2);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) {
            red[0][0] = fminf(red[0][0], red[w][0]); red[0][1] = fmaxf(red[0][1], red[w][1]);
            red[0][2] = fminf(red[0][2], red[w][2]); red[0][3] = fmaxf(red[0][3], red[w][3]);
            bad[0] |= bad[w];
        }
        valid[i] = (bad[0] & 1) ? 0 : 1;
        const float qnan = __builtin_nanf("");
        minmax[4 * i + 0] = (bad[0] & 2) ? qnan : red[0][0]; minmax[4 * i + 1] = (bad[0] & 2) ? qnan : red[0][1];
        minmax[4 * i + 2] = (bad[0] & 4) ? qnan : red[0][2]; minmax[4 * i + 3] = (bad[0] & 4) ? qnan : red[0][3];
    }
}

hipError_t launch_patch_stats(const float* img, const float* dem, int rows, int cols, const int* ox, const int* oy,
                              int n, int S, float no_value, uint8_t* valid, float* minmax, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    patch_stats_kernel<<<n, 256, 0, s>>>(img, dem, rows, cols, ox, oy, S, no_value, valid, minmax);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// extract_patches: normalize (process_full_tiles.py:307-310): ((p - min) / (max - min)) - 0.5 in float32,
// channel 0 = ortho, 1 = DEM.  Origin (-1,-1) = the zero padding patch of :468-474.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) extract_patches_kernel(const float* __restrict__ img,
                                                              const float* __restrict__ dem, int cols,
                                                              const int* __restrict__ ox, const int* __restrict__ oy,
                                                              const float* __restrict__ minmax, int S,
                                                              float* __restrict__ out) {
    const int i = blockIdx.y;
    const int x0 = ox[i], y0 = oy[i];
    float2* o = reinterpret_cast<float2*>(out) + (size_t)i * S * S;
    const bool pad = x0 < 0 || y0 < 0;
    const float imn = minmax[4 * i + 0], imx = minmax[4 * i + 1], dmn = minmax[4 * i + 2], dmx = minmax[4 * i + 3];
    const float irange = imx - imn, drange = dmx - dmn;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < S * S; e += gridDim.x * 256) {
        float2 v = make_float2(0.f, 0.f);
        if (!pad) {
            const int y = e / S, x = e % S;
            const size_t off = (size_t)(y0 + y) * cols + x0 + x;
            v.x = (img[off] - imn) / irange - 0.5f;
            v.y = (dem[off] - dmn) / drange - 0.5f;
        }
        o[e] = v;
    }
}

hipError_t launch_extract_patches(const float* img, const float* dem, int rows, int cols, const int* ox,
                                  const int* oy, const float* minmax, int n, int S, float* out, hipStream_t s) {
    (void)rows;
    if (n <= 0) return hipSuccess;
    int bx = (S * S + 255) / 256;
    if (bx > 64) bx = 64;
    extract_patches_kernel<<<dim3(bx, n), 256, 0, s>>>(img, dem, cols, ox, oy, minmax, S, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// compact_patches: the batch assembly of processTile (process_full_tiles.py:455-474) on the device.  The
// reference's Python loop skips invalid patches, appends the valid ones in generation order (y outer, x inner),
// cuts them into calls of B and pads the last call with zero patches keyed (-1, -1).  Here one workgroup makes a
// stable compaction of the tile's n candidates by their validity flags (ballot + popcount per wave, running base
// per 1024 candidates: order preserved exactly) and fills the tail up to `cap` with the padding entries, so the
// host only needs the two counts in `meta` = {valid patches, calls} — 8 bytes instead of the flags and a
// host-side compaction — and can fetch them while the previous tile is still generating.
//   sel_x / sel_y [cap]   origins in padded-canvas coordinates, (-1,-1) = zero patch (what extract_patches takes)
//   sel_mm [cap][4]       their {img_min, img_max, dem_min, dem_max}, zeros for padding
//   key [cap][2]          origins relative to the tile (the reference's dict keys), (-1,-1) for padding
//   dmm [cap][2]          {dem_min, dem_max} (what stitch_tile takes)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) compact_patches_kernel(const uint8_t* __restrict__ valid,
                                                               const int* __restrict__ ox, const int* __restrict__ oy,
                                                               const float* __restrict__ minmax, int n, int tile_x,
                                                               int tile_y, int B, int cap, int* __restrict__ sel_x,
                                                               int* __restrict__ sel_y, float* __restrict__ sel_mm,
                                                               int* __restrict__ key, float* __restrict__ dmm,
                                                               int* __restrict__ meta) {
    __shared__ int wsum[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int base = 0;
    for (int start = 0; start < n; start += 1024) {
        const int i = start + threadIdx.x;
        const int v = i < n ? (valid[i] != 0) : 0;
        const unsigned long long m = __ballot(v);
        const int pre = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(m);
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const int c = wsum[w];
            woff += w < wave ? c : 0;
            tot += c;
        }
        if (v) {
            const int d = base + woff + pre;
            const int x = ox[i], y = oy[i];
            sel_x[d] = x; sel_y[d] = y;
            key[2 * d] = x - tile_x; key[2 * d + 1] = y - tile_y;
            const float4 mm = *reinterpret_cast<const float4*>(minmax + 4 * (size_t)i);
            *reinterpret_cast<float4*>(sel_mm + 4 * (size_t)d) = mm;
            dmm[2 * d] = mm.z; dmm[2 * d + 1] = mm.w;
        }
        base += tot;
        __syncthreads();
    }
    for (int d = base + threadIdx.x; d < cap; d += 1024) {
        sel_x[d] = -1; sel_y[d] = -1;
        key[2 * d] = -1; key[2 * d + 1] = -1;
        *reinterpret_cast<float4*>(sel_mm + 4 * (size_t)d) = make_float4(0.f, 0.f, 0.f, 0.f);
        dmm[2 * d] = 0.f; dmm[2 * d + 1] = 0.f;
    }
    if (threadIdx.x == 0) { meta[0] = base; meta[1] = (base + B - 1) / B; }
}

hipError_t launch_compact_patches(const uint8_t* valid, const int* ox, const int* oy, const float* minmax, int n,
                                  int tile_x, int tile_y, int B, int cap, int* sel_x, int* sel_y, float* sel_mm,
                                  int* key, float* dmm, int* meta, hipStream_t s) {
    compact_patches_kernel<<<1, 1024, 0, s>>>(valid, ox, oy, minmax, n, tile_x, tile_y, B, cap, sel_x, sel_y, sel_mm,
                                              key, dmm, meta);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// stitch_tile: rebuildTile (process_full_tiles.py:363-414) as a GATHER: one thread per output pixel of the
// cropped [T,T] tile walks the <= ((S-2p)/s)^2 patches covering it in generation order (y outer, x inner —
// the insertion order of the reference's dict) and applies the weighted incremental update in registers.
// No atomics, no accumulator images in HBM, deterministic, same summation order as the reference.
// Types follow NumPy: window float64, accumulators float32, each update evaluated in float64 and rounded
// to float32 on store.  as_implemented reproduces the aliasing of :400-402 (S += w*(x-mean_new)^2).
// ------------------------------------------------------------------------------------------------
__global__ void stitch_grid_kernel(const int* __restrict__ key, int n, int stride, int NG, int* __restrict__ grid) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int kx = key[2 * i], ky = key[2 * i + 1];
    if (kx < 0 || ky < 0 || kx % stride || ky % stride) return;
    const int gx = kx / stride, gy = ky / stride;
    if (gx < NG && gy < NG) grid[gy * NG + gx] = i;
}

__global__ void __launch_bounds__(256) stitch_tile_kernel(const float* __restrict__ pred, const float* __restrict__ dmm,
                                                          const int* __restrict__ grid, int NG, int S, int T,
                                                          int stride, float no_value, int as_implemented,
                                                          const double* __restrict__ window, float* __restrict__ mean_o,
                                                          float* __restrict__ std_o, uint8_t* __restrict__ good_o,
                                                          float* __restrict__ wsum_o, int pitch, int resume) {
    const int tx = blockIdx.x * 16 + (threadIdx.x & 15);
    const int ty = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (tx >= T || ty >= T) return;
    const int halo = S - stride, purge = S / 16, ws = S - 2 * purge;
    const int ax = tx + halo, ay = ty + halo;
    // patches with ky in (ay - S + purge, ay - purge]
    auto lo_idx = [&](int a) { int v = a - S + purge + 1; return v <= 0 ? 0 : (v + stride - 1) / stride; };
    auto hi_idx = [&](int a) { int v = a - purge; return v < 0 ? -1 : min(v / stride, NG - 1); };
    const int gy0 = lo_idx(ay), gy1 = hi_idx(ay), gx0 = lo_idx(ax), gx1 = hi_idx(ax);
    const size_t o = (size_t)ty * pitch + tx;
    float w_sum = 0.f, mean = 0.f, s_acc = 0.f;
    if (resume) {      // halo mode, banded: continue the running update where the previous band of patch rows left it (bands
                       // come in generation order, so the sequence of updates per pixel is the all-at-once sequence)
        w_sum = wsum_o[o]; mean = mean_o[o]; s_acc = std_o[o];
    }
    for (int gy = gy0; gy <= gy1; ++gy) {
        const int py = ay - gy * stride;
        for (int gx = gx0; gx <= gx1; ++gx) {
            const int slot = grid[gy * NG + gx];
            if (slot < 0) continue;
            const int px = ax - gx * stride;
            const float pr = pred[(size_t)slot * S * S + (size_t)py * S + px] + 0.5f;       // :340
            const float lo = dmm[2 * slot], hi = dmm[2 * slot + 1];
            const float xval = pr * (hi - lo) + lo;                                          // :395
            const double w = window[(size_t)(py - purge) * ws + (px - purge)];
            w_sum = (float)((double)w_sum + w);                                              // :397
            const float d_old = xval - mean;
            const float mean_new = (float)((double)mean + (w / (double)w_sum) * (double)d_old);   // :401
            const float d_new = xval - mean_new;
            const double first = as_implemented ? (double)d_new : (double)d_old;
            s_acc = (float)((double)s_acc + (w * first) * (double)d_new);                    // :402
            mean = mean_new;
        }
    }
    if (wsum_o) {      // halo mode: the raw accumulators (w_sum, mean, S) of this rank's patches, merged later
        wsum_o[o] = w_sum; mean_o[o] = mean; std_o[o] = s_acc;
        return;
    }
    const bool good = w_sum > 0.f;                                                           // :409
    const float sd = sqrtf(s_acc / w_sum);                                                   // :411
    mean_o[o] = good ? mean : no_value;
    std_o[o] = good ? sd : no_value;
    good_o[o] = good ? 1 : 0;
}

hipError_t launch_stitch_tile(const float* pred, const int* key, const float* dmm, int n, int S, int T, int stride,
                              float no_value, int as_implemented, const double* window, int* grid_ws, float* mean,
                              float* stdv, uint8_t* good, hipStream_t s, float* wsum_partial, int pitch, int resume) {
    const int NG = (T + S - 1) / stride;   // len(range(0, T + S - stride, stride))
    hipError_t e = hipMemsetAsync(grid_ws, 0xFF, sizeof(int) * NG * NG, s);
    if (e != hipSuccess) return e;
    if (n > 0) stitch_grid_kernel<<<(n + 255) / 256, 256, 0, s>>>(key, n, stride, NG, grid_ws);
    stitch_tile_kernel<<<dim3((T + 15) / 16, (T + 15) / 16), 256, 0, s>>>(pred, dmm, grid_ws, NG, S, T, stride,
                                                                         no_value, as_implemented, window, mean,
                                                                         stdv, good, wsum_partial, pitch > 0 ? pitch : T,
                                                                         wsum_partial ? resume : 0);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// halo_merge: the exchange step of the patch-row-sharded ("halo") mode.  Two ranks hold, for the same output
// pixels, the weighted-Welford accumulators (w_sum, mean, S) of their own patches (stitch_tile with partial output,
// textbook West S); the pairwise combine (Chan et al.) of a = the lower rank's (earlier patches) and b = the upper
// rank's gives the accumulators of the union:
//     w = wa + wb,  d = mb - ma,  mean = ma + d * wb / w,  S = Sa + Sb + d^2 * wa * wb / w
// evaluated in float64, then finalised like rebuildTile (process_full_tiles.py:409-413): good = w > 0,
// std = sqrt(max(S, 0) / w), no_value where not good.  b == nullptr finalises a alone.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) halo_merge_kernel(const float* __restrict__ wa, const float* __restrict__ ma,
                                                         const float* __restrict__ sa, const float* __restrict__ wb,
                                                         const float* __restrict__ mb, const float* __restrict__ sb,
                                                         long n, float no_value, float* __restrict__ mean_o,
                                                         float* __restrict__ std_o, uint8_t* __restrict__ good_o) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        double w = wa[i], m = ma[i], S = sa[i];
        if (wb) {
            const double w2 = wb[i];
            if (w2 > 0.0) {
                if (w > 0.0) {
                    const double d = (double)mb[i] - m, wt = w + w2;
                    S = S + (double)sb[i] + d * d * (w * w2 / wt);
                    m = m + d * (w2 / wt);
                    w = wt;
                } else {
                    w = w2; m = mb[i]; S = sb[i];
                }
            }
        }
        const bool good = w > 0.0;
        mean_o[i] = good ? (float)m : no_value;
        // float32 rounding of the West update / the Chan combine can leave S a hair below zero: clamp, so that the
        // standard deviation is finite wherever good == 1 (NaN in S still propagates: fmaxf is not used)
        const float Sf = (float)S < 0.f ? 0.f : (float)S;
        std_o[i] = good ? sqrtf(Sf / (float)w) : no_value;
        good_o[i] = good ? 1 : 0;
    }
}

hipError_t launch_halo_merge(const float* wa, const float* ma, const float* sa, const float* wb, const float* mb,
                             const float* sb, long n, float no_value, float* mean, float* stdv, uint8_t* good,
                             hipStream_t s) {
    if (n <= 0) return hipSuccess;
    long blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    halo_merge_kernel<<<(int)blocks, 256, 0, s>>>(wa, ma, sa, wb, mb, sb, n, no_value, mean, stdv, good);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Raster resamplers of the reference's pre-processing (process_full_tiles.py:226-244: cv2.resize INTER_AREA x1/4,
// twice, then INTER_CUBIC back to full size).  One thread per destination pixel, float32 operation order of
// oracle/preprocess_ref.py (which restates OpenCV's published algorithm; this file is compiled with
// -ffp-contract=off), so the two agree bit for bit.  HBM-bound: the full-resolution raster is read / written once.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) resize_area_kernel(const float* __restrict__ src, int h, int w,
                                                          float* __restrict__ dst, int dh, int dw, int f) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)dh * dw) return;
    const int dx = (int)(i % dw), dy = (int)(i / dw);
    const int x0 = dx * f, y0 = dy * f;
    if (y0 >= h || x0 >= w) { dst[i] = 0.f; return; }
    if (y0 + f <= h && x0 + f <= w) {
        // full block: rows outer, four columns at a time (sum += S[k] + S[k+1] + S[k+2] + S[k+3]), times 1/area
        float sum = 0.f;
        for (int r = 0; r < f; ++r) {
            const float* S = src + (size_t)(y0 + r) * w + x0;
            int k = 0;
            for (; k + 4 <= f; k += 4) sum = sum + (((S[k] + S[k + 1]) + S[k + 2]) + S[k + 3]);
            for (; k < f; ++k) sum = sum + S[k];
        }
        dst[i] = sum * (1.0f / (float)(f * f));
    } else {
        float sum = 0.f;
        int n = 0;
        for (int yy = y0; yy < min(y0 + f, h); ++yy)
            for (int xx = x0; xx < min(x0 + f, w); ++xx) { sum = sum + src[(size_t)yy * w + xx]; ++n; }
        dst[i] = sum / (float)n;
    }
}

__device__ __forceinline__ void cubic_axis(int d, double scale, int n_src, int (&idx)[4], float (&c)[4]) {
    const float f = (float)(((double)d + 0.5) * scale - 0.5);
    const float fl = floorf(f);
    const int s = (int)fl;
    const float t = f - fl;
    const float A = -0.75f;
    c[0] = ((A * (t + 1.f) - 5.f * A) * (t + 1.f) + 8.f * A) * (t + 1.f) - 4.f * A;
    c[1] = ((A + 2.f) * t - (A + 3.f)) * t * t + 1.f;
    const float u = 1.f - t;
    c[2] = ((A + 2.f) * u - (A + 3.f)) * u * u + 1.f;
    c[3] = 1.f - c[0] - c[1] - c[2];
#pragma unroll
    for (int k = 0; k < 4; ++k) idx[k] = min(max(s - 1 + k, 0), n_src - 1);
}

__global__ void __launch_bounds__(256) resize_cubic_kernel(const float* __restrict__ src, int h, int w,
                                                           float* __restrict__ dst, int dh, int dw) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)dh * dw) return;
    const int dx = (int)(i % dw), dy = (int)(i / dw);
    int xi[4], yi[4];
    float a[4], b[4];
    cubic_axis(dx, (double)w / (double)dw, w, xi, a);
    cubic_axis(dy, (double)h / (double)dh, h, yi, b);
    float rows[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float* S = src + (size_t)yi[k] * w;
        rows[k] = ((S[xi[0]] * a[0] + S[xi[1]] * a[1]) + S[xi[2]] * a[2]) + S[xi[3]] * a[3];   // horizontal pass
    }
    dst[i] = ((rows[0] * b[0] + rows[1] * b[1]) + rows[2] * b[2]) + rows[3] * b[3];            // vertical pass
}

hipError_t launch_resize_area(const float* src, int h, int w, float* dst, int dh, int dw, int factor, hipStream_t s) {
    const long n = (long)dh * dw;
    if (n <= 0 || factor < 1) return hipErrorInvalidValue;
    resize_area_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(src, h, w, dst, dh, dw, factor);
    return hipGetLastError();
}

hipError_t launch_resize_cubic(const float* src, int h, int w, float* dst, int dh, int dw, hipStream_t s) {
    const long n = (long)dh * dw;
    if (n <= 0 || h < 1 || w < 1) return hipErrorInvalidValue;
    resize_cubic_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(src, h, w, dst, dh, dw);
    return hipGetLastError();
}

}  // namespace msr
