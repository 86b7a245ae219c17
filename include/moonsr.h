/*
 * moonsr.h — C ABI of libmoonsr_hip.so, the MI355X (gfx950) implementation of the tiled DEM
 * super-resolution inference path of AntoineRichard/MoonSuperResolution.
 *
 * The reference has NO native interface: its device boundary is the duck-typed Python call
 *     pred = self.model(np.array(batch), training=False)        (process_full_tiles.py:338)
 * on a Keras model built by GauGAN(image_size, batch_size, latent_dim) (process_full_tiles.py:28,
 * spade/models/model.py:340-380) — and its stitcher is NumPy (process_full_tiles.py:363-414).
 * This header is what a binding for that path would bind; INTEGRATION.md shows the ctypes stub.
 *
 * Conventions
 *  - plain pointers and sizes only; no torch / numpy types.
 *  - every function returns 0 on success or a negative msr_status; msr_last_error() gives the text.
 *  - tensors are NHWC float32; "dev" pointers are HIP device pointers owned by the CALLER.  The
 *    library owns only the weights and the workspace inside the handle.
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls are asynchronous
 *    on that stream; a handle is not re-entrant.
 *  - there is NO CPU fallback: if no gfx950 device is usable msr_create fails with MSR_ERR_DEVICE.
 */
#ifndef MOONSR_H
#define MOONSR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSR_ABI_VERSION 1

typedef enum {
    MSR_OK = 0,
    MSR_ERR_INVALID = -1,   /* bad argument / shape (the reference raises ValueError / assert) */
    MSR_ERR_DEVICE = -2,    /* HIP error or no usable device */
    MSR_ERR_STATE = -3,     /* call out of order (e.g. forward before all weights are loaded) */
    MSR_ERR_NOMEM = -4
} msr_status;

/* Generator families of the reference. */
typedef enum {
    MSR_GAUGAN = 0,        /* GauGAN.call        spade/models/model.py:564-567  z = mean + exp(var/2)*eps */
    MSR_GAUGAN_NO_KL = 1,  /* GauGAN_no_KL.call  spade/models/model.py:265-267  z = mean + var          */
    MSR_CNN = 2,           /* CNNSpade.call      spade/models/model.py:789-791  z = mean + var          */
    MSR_PIX2PIX = 3        /* Pix2Pix().generator  pix2pix.py:88-108 (image_size fixed to 256)            */
} msr_variant;

typedef struct {
    int32_t image_size;  /* S: generator input/output side, power of two >= 64 (GauGAN(image_size, ...)) */
    int32_t batch_size;  /* B: patches per call; baked in like the reference's sampler (sampling.py:13-15) */
    int32_t latent_dim;  /* 256 in every reference script (process_full_tiles.py:28) */
    int32_t variant;     /* msr_variant */
    int32_t device;      /* HIP device ordinal */
    int32_t flags;       /* MSR_FLAG_* */
} msr_config;

/* Conv arithmetic (msr_config.flags).  Inputs, outputs and weights of the C ABI are fp32 in every mode, and so is all
 * other arithmetic (moments, normalisation, epilogues, dense layers, head); only the conv products differ:
 *   0                 exact fp32 on v_mfma_f32_32x32x2_f32 (157 TFLOP/s peak).  NOTE: 0 is the C-ABI default; the Python
 *                     host (moonsuperresolution_amd.Generator) defaults to MSR_FLAG_BF16X3 | MSR_FLAG_F16C, ~4x faster.
 *   MSR_FLAG_BF16X3   3-term split-bf16 products (a_hi*b_hi + a_hi*b_lo + a_lo*b_hi) with fp32 accumulation, on
 *                     v_mfma_f32_16x16x32_bf16 in the LDS-halo kernels (the r >= 16 layers, 91 % of the FLOPs) and on
 *                     v_mfma_f32_32x32x16_bf16 in the small-tile ones: per-product error <= ~3*2^-18.
 *   MSR_FLAG_BF16X3 | MSR_FLAG_GB_F16X2   opt-in: as BF16X3, but the SPADE gamma|beta convs (spade.py:19-20, half of
 *                     the FLOPs) of the layers that run the persistent ping-pong kernel use 2-term fp16 products
 *                     (activation split in two fp16 halves, weight rounded to ONE fp16; v_mfma_f32_16x16x32_f16):
 *                     per-product error <= 2^-12, measured 2-5e-4 relative L-inf end to end (inside north_star's 1e-3,
 *                     with a 2-4x margin instead of 50x; tests/test_gpu_baseline_configs.py states the bound).
 * MSR_PIX2PIX ignores the flags and always computes on the fp32 MFMA. */
#define MSR_FLAG_BF16X3 1
#define MSR_FLAG_GB_F16X2 2
/*   MSR_FLAG_BF16X3 | MSR_FLAG_FP8   declared NON-parity mode (BASELINE.json configs[4], "fp8 MFMA conv"): every 3x3
 *                     stride-1 conv that fills the chip with whole ping-pong tiles (the gamma|beta and ResidualBlock
 *                     convs of the r >= 32 blocks at the BASELINE sizes, ~97 % of the FLOPs) multiplies fp8 e4m3 weights
 *                     (a power-of-two scale per output channel, carried by the instruction's e8m0 scale operand) by bf8
 *                     e5m2 activations on v_mfma_scale_f32_16x16x128_f8f6f4 with fp32 accumulation: 3 and 2 mantissa
 *                     bits per operand.  Its error against the float64 oracle is measured and stated in
 *                     tests/test_gpu_baseline_configs.py; it does NOT meet the 1e-3 bar and is never the default. */
#define MSR_FLAG_FP8 4
/*   MSR_FLAG_BF16X3 | MSR_FLAG_F16C   the same layers as MSR_FLAG_FP8 covers, on "fp16 main term + fp8 cross terms":
 *                     a*b = a_hi*b_hi (v_mfma_f32_16x16x32_f16, exact products) + (a_hi*b_lo + a_lo*b_hi) on the block-scaled fp8
 *                     MFMA (the cross terms are 2^-11 of the product, so 4 bits of them suffice; K = 128 covers both cross
 *                     terms of two taps per instruction at twice the bf16 rate): two MFMA-equivalents per product instead
 *                     of three, per-product error ~2^-15, 3-5e-5 relative L-inf end to end — a parity mode
 *                     (tests/test_gpu_baseline_configs.py). */
#define MSR_FLAG_F16C 8
/*   MSR_FLAG_BF16X3 | MSR_FLAG_F16C | MSR_FLAG_F16_MAIN   declared-tolerance fast mode ("f16", round 3): the F16C data
 *                     path with the cross terms left out of the two kernels that carry ~95 % of the FLOPs (conv_gb_resident,
 *                     conv_igemm_f16c_sw): ONE fp16 product per element on v_mfma_f32_16x16x32_f16, fp32 accumulation,
 *                     per-product error 2^-11.  Tensors, weights and every other layer are exactly F16C's.  It is the usable
 *                     reading of BASELINE.json configs[4] ("fp16 ... conv"): the measured end-to-end error is stated in
 *                     tests/test_gpu_baseline_configs.py::test_f16_mode_declared_tolerance (bound 3e-2 relative L-inf; it is
 *                     NOT inside north_star's 1e-3 and never the default). */
#define MSR_FLAG_F16_MAIN 16

typedef struct msr_handle msr_handle;

/* ---- lifetime -------------------------------------------------------------------------------- */
int msr_abi_version(void);
/* Replaces: GauGAN(image_size, batch_size, latent_dim) + .compile()  (process_full_tiles.py:28-29). */
int msr_create(const msr_config* cfg, msr_handle** out);
int msr_destroy(msr_handle* h);
/* Text of the last error on this handle (h == NULL: last error of a failed msr_create). */
const char* msr_last_error(const msr_handle* h);

/* ---- weights --------------------------------------------------------------------------------- */
/* Replaces: gaugan.load(path+'generator', ..., path+'encoder')  (process_full_tiles.py:30,
 * model.py:607-610).  `host` is a HOST float32 array in the reference's own layout: conv kernels
 * HWIO [kh,kw,Cin,Cout], Conv2DTranspose kernels [kh,kw,Cout,Cin], dense [in,out]; names as listed
 * by moonsuperresolution_amd.weights.weight_shapes().  The library re-lays them out for its kernels. */
int msr_load_weight(msr_handle* h, const char* name, const float* host, const int64_t* shape, int32_t rank);
/* Number of weights the variant expects / has received so far. */
int msr_weight_count(const msr_handle* h, int32_t* expected, int32_t* loaded);
/* Name of the i-th expected weight (NULL if out of range) and its shape. */
const char* msr_weight_name(const msr_handle* h, int32_t i, int64_t* shape4, int32_t* rank);

/* ---- the generator(call) --------------------------------------------------------------------- */
/* Replaces: self.model(np.array(batch), training=False)  (process_full_tiles.py:338).
 *   in_dev   [batch, S, S, 2]  channel 0 = ortho, 1 = low-res DEM, values in [-0.5, 0.5]
 *   eps_dev  [batch, latent]   the sampler's N(0,1) draw (sampling.py:13-15); required for
 *                              MSR_GAUGAN, ignored otherwise
 *   out_dev  [batch, S, S, 1]
 * batch must equal cfg.batch_size (the reference's sampler enforces the same). */
int msr_forward(msr_handle* h, const float* in_dev, const float* eps_dev, float* out_dev, int32_t batch,
                void* stream);
/* msr_forward for callers that pipeline independent calls over two handles on two streams (the tile loop of
 * process_full_tiles.py:453-474 issues its batches one after the other; they do not depend on each other): the call's
 * latency-bound first part (encoder, dense layers, the low-resolution blocks: ~20 % of a call at low occupancy) is
 * launched at once, its matrix-bound part (from the first layer that fills the chip) waits for `gate_event`
 * (a hipEvent_t the caller recorded at the end of the previous call, on the other stream).  The heavy parts of
 * consecutive calls then run back to back and every call's head hides under its predecessor's tail — free-running
 * streams settle into either a favourable or an unfavourable phase (+6 % / -4 % against one stream), this is the
 * favourable one by construction.  gate_event == NULL: plain msr_forward.  Not captured into graphs. */
int msr_forward_gated(msr_handle* h, const float* in_dev, const float* eps_dev, float* out_dev, int32_t batch,
                      void* stream, void* gate_event);
/* on != 0: msr_forward captures its launch plan (~100 kernels on the call's stream and the handle's auxiliary stream)
 * into a HIP graph the first time it sees an (in_dev, eps_dev, out_dev) pointer triple and replays it with ONE
 * hipGraphLaunch afterwards (up to 8 triples are kept; further ones, the NULL stream and profiled calls launch
 * eagerly).  For callers that reuse their buffers — the B = 1 latency case of the serial call at
 * process_full_tiles.py:338.  Results are identical (same kernels, same order).  on == 0 drops the graphs. */
int msr_graph_enable(msr_handle* h, int32_t on);
/* Latent z [batch, latent] of the last msr_forward (debug / parity aid), copied to a device buffer. */
int msr_last_latent(msr_handle* h, float* z_dev, void* stream);

/* ---- tiler / stitcher ------------------------------------------------------------------------ */
/* Replaces getPatch + normalize for a whole tile (process_full_tiles.py:269-311, 453-457).
 * For every patch origin (ox[i], oy[i]) (padded-canvas coordinates, int32 on device):
 *   valid[i]   = no pixel <= no_value in either raster             (uint8)
 *   minmax[i]  = {img_min, img_max, dem_min, dem_max}               (float32 x4)
 * img_dev / dem_dev are the padded float32 rasters [rows, cols] (pitch = cols). */
int msr_patch_stats(msr_handle* h, const float* img_dev, const float* dem_dev, int32_t rows, int32_t cols,
                    const int32_t* ox_dev, const int32_t* oy_dev, int32_t n, float no_value,
                    uint8_t* valid_dev, float* minmax_dev, void* stream);
/* Writes normalised patches [n, S, S, 2] for the given origins; origin (-1,-1) = the all-zero padding
 * patch of process_full_tiles.py:468-474. */
int msr_extract_patches(msr_handle* h, const float* img_dev, const float* dem_dev, int32_t rows, int32_t cols,
                        const int32_t* ox_dev, const int32_t* oy_dev, const float* minmax_dev, int32_t n,
                        float* out_dev, void* stream);
/* Replaces the batch assembly of processTile (process_full_tiles.py:455-474: skip invalid patches, keep generation
 * order, cut into calls of `batch`, pad the last call with zero patches keyed (-1,-1)) for one tile, on the device:
 * a stable compaction of the n candidates of msr_patch_stats by their validity flags.
 *   cap           capacity of the outputs = ceil(n / batch) * batch (all-valid worst case)
 *   sel_x / sel_y [cap] int32   origins (padded-canvas coordinates) in generation order, then (-1,-1) padding:
 *                               what msr_extract_patches takes, `batch` entries per generator call
 *   sel_minmax    [cap][4]      their {img_min, img_max, dem_min, dem_max}; zeros for padding
 *   key           [cap][2]      origins relative to (tile_x, tile_y) = the reference's dict keys; (-1,-1) padding
 *   dmm           [cap][2]      {dem_min, dem_max}: what msr_stitch_tile takes
 *   meta          [2] int32     {number of valid patches, number of generator calls}
 * The host reads back only `meta` (8 bytes). */
int msr_compact_patches(msr_handle* h, const uint8_t* valid_dev, const int32_t* ox_dev, const int32_t* oy_dev,
                        const float* minmax_dev, int32_t n, int32_t tile_x, int32_t tile_y, int32_t batch, int32_t cap,
                        int32_t* sel_x_dev, int32_t* sel_y_dev, float* sel_minmax_dev, int32_t* key_dev,
                        float* dmm_dev, int32_t* meta_dev, void* stream);
/* Replaces processBatch's "+0.5" and rebuildTile (process_full_tiles.py:340, 363-414) for one tile.
 *   pred_dev    [n, S, S]  generator outputs (last channel), in generation order
 *   key_dev     [n, 2]     int32 (x, y) of each patch relative to the tile origin (padded coords)
 *   dmm_dev     [n, 2]     float32 (dem_min, dem_max) of each patch
 *   mean/std    [T, T] float32, good [T, T] uint8
 * as_implemented != 0 reproduces the reference's aliased variance update. */
int msr_stitch_tile(msr_handle* h, const float* pred_dev, const int32_t* key_dev, const float* dmm_dev, int32_t n,
                    int32_t tile_size, int32_t stride, float no_value, int32_t as_implemented,
                    float* mean_dev, float* std_dev, uint8_t* good_dev, void* stream);

/* ---- patch-row-sharded ("halo") mode: north_star's exchange of overlap halos (second mode) ------------
 * The reference has no counterpart: it re-generates the halo patches of every tile (process_full_tiles.py:449-454).
 * msr_stitch_partial is msr_stitch_tile stopped before the finalisation of :409-413, with the textbook (West) variance
 * update: it returns the raw accumulators (w_sum, mean, S) [T, T] of the patches it was given, so that two ranks that
 * each hold part of the patches covering a pixel can combine them. */
int msr_stitch_partial(msr_handle* h, const float* pred_dev, const int32_t* key_dev, const float* dmm_dev, int32_t n,
                       int32_t tile_size, int32_t stride, float* wsum_dev, float* mean_dev, float* s_dev, void* stream);
/* msr_stitch_partial writing IN PLACE into a T x T window of larger accumulator images (row pitch `pitch` >= tile_size
 * elements) and, with resume != 0, continuing the running update from what the window already holds instead of from zero:
 * the banded form of the halo mode (moonsuperresolution_amd/halo.py) — a rank generates its patch rows band by band (in
 * generation order), accumulates each band into the canvas rows it reaches and frees its predictions.  The sequence of
 * updates per pixel is the all-at-once sequence, so the result does not depend on the band size, bit for bit. */
int msr_stitch_accumulate(msr_handle* h, const float* pred_dev, const int32_t* key_dev, const float* dmm_dev, int32_t n,
                          int32_t tile_size, int32_t stride, float* wsum_dev, float* mean_dev, float* s_dev, int32_t pitch,
                          int32_t resume, void* stream);
/* Pairwise (Chan) combine of two sets of accumulators of the same `count` pixels — a = the rank with the earlier patch
 * rows, b = the later one, or b == NULL — followed by rebuildTile's finalisation (good = w_sum > 0,
 * std = sqrt(S / w_sum), no_value where not good; process_full_tiles.py:409-413). */
int msr_halo_merge(msr_handle* h, const float* wa_dev, const float* ma_dev, const float* sa_dev, const float* wb_dev,
                   const float* mb_dev, const float* sb_dev, int64_t count, float no_value, float* mean_dev,
                   float* std_dev, uint8_t* good_dev, void* stream);

/* Optional: replace the library's own blending window (makeGaussianKernel + 1e-7, purged S//16 per side,
 * process_full_tiles.py:347-361,391-393) by a caller-computed float64 [S-2p, S-2p] HOST array — the Python
 * host passes NumPy's own evaluation so the GPU stitcher is bit-identical to the NumPy reference. */
int msr_set_blend_window(msr_handle* h, const double* host_window, int32_t side);

/* CRC-32C (Castagnoli) of a HOST buffer, continuing from `crc` (0 to start).  Host-only helper of the TensorBundle
 * weight reader (moonsuperresolution_amd/tf_checkpoint.py): checkpoint data and index blocks carry masked CRC-32C. */
uint32_t msr_crc32c(const void* host_data, uint64_t n, uint32_t crc);

/* ---- pre-processing resamplers ------------------------------------------------ */
/* Replaces cv2.resize(dem, (0,0), fx=1/factor, fy=1/factor, interpolation=cv2.INTER_AREA) of preprocess
 * (process_full_tiles.py:232,238) on a float32 raster [rows, cols] in device memory: dst [dst_rows, dst_cols] with
 * dst_* = cvRound(src_* / factor); every destination pixel is the mean of its factor x factor block (NaN propagates;
 * partial edge blocks average the pixels that exist). */
int msr_resize_area(msr_handle* h, const float* src_dev, int32_t rows, int32_t cols, int32_t factor, float* dst_dev,
                    int32_t dst_rows, int32_t dst_cols, void* stream);
/* Replaces cv2.resize(dem_rs, dsize, interpolation=cv2.INTER_CUBIC) (process_full_tiles.py:241): Keys cubic,
 * A = -0.75, pixel-centre mapping, replicated border, float32; dst [dst_rows, dst_cols]. */
int msr_resize_cubic(msr_handle* h, const float* src_dev, int32_t rows, int32_t cols, float* dst_dev,
                     int32_t dst_rows, int32_t dst_cols, void* stream);

/* TIFF 6.0 LZW (compression 5) of HOST buffers, for the GeoTIFF reader / writer (moonsuperresolution_amd/geotiff.py;
 * the reference reads and writes LZW GeoTIFFs through GDAL: process_full_tiles.py:158-182, 481-531).
 * Both return the number of bytes produced, or -1 on a malformed stream / too small output buffer. */
int64_t msr_lzw_decode(const uint8_t* in, int64_t n_in, uint8_t* out, int64_t n_out);
int64_t msr_lzw_encode(const uint8_t* in, int64_t n_in, uint8_t* out, int64_t cap);

/* ---- measurement ----------------------------------------------------------------------------- */
typedef struct {
    char name[48];        /* kernel family, e.g. "conv_igemm_f32" */
    int64_t launches;
    double device_ms;     /* sum of hipEvent-bracketed launch durations on the call's stream */
    double flops;         /* algorithmic FLOPs of those launches (2*MACs), 0 for memory-bound families */
    double bytes;         /* algorithmic bytes of those launches */
} msr_kernel_stat;
/* on = 1: every kernel launch of msr_forward is bracketed by hipEvents on the call's stream (an event costs the
 * stream 2-3 us: ~7 % of a call's throughput).  on = 2: only the conv family is timed and a run of consecutive conv
 * launches shares one pair of events (device_ms then includes the 1.5-3 us gaps inside a run; < 2 % overhead).
 * on = 0: off. */
int msr_profile_enable(msr_handle* h, int32_t on);
int msr_profile_reset(msr_handle* h);
/* Synchronises the recorded events and fills up to cap entries; *n = number of families. */
int msr_profile_read(msr_handle* h, msr_kernel_stat* out, int32_t cap, int32_t* n);
/* The recorded intervals of one family (0 = the conv family), in ms relative to ref_event (a hipEvent_t the caller
 * recorded before the calls): lets a caller that drives several handles on several streams take the UNION of the
 * family's busy intervals instead of their sum. */
int msr_profile_runs(msr_handle* h, void* ref_event, int32_t family, double* start_ms, double* end_ms, double* flops,
                     int64_t* launches, int32_t cap, int32_t* n);
/* Algorithmic FLOPs of one msr_forward call (all patches), the figure BASELINE.md section 2 derives. */
int msr_forward_flops(const msr_handle* h, double* flops);
/* Kernel-level entry (parity tests and micro-benchmarks of the dominant kernel): one conv_igemm_f32 launch.
 *   in_dev   zero-bordered NHWC input [B, rin+2, rin+2, Cin], rin = rout*stride
 *   wt_dev   weights already in the kernel layout [9][N][Cin]; bias_dev [N] (GEMM column order)
 *   epilogue 0 bias, 1 bias+residual(aux [B, rout>>aux_shift, ., N]), 2 SPADE (N = 2C, interleaved columns;
 *            aux = x [B, rout>>aux_shift, ., C], mean/std [C]); out_padded != 0 writes a zero-bordered tensor
 *   tile     -1 auto (tile and K split chosen by the library), else (0 = 128x128 | 1 = 64x64) + 256 * ksplit */
int msr_op_conv3x3(msr_handle* h, const float* in_dev, const float* wt_dev, const float* bias_dev, float* out_dev,
                   int32_t B, int32_t rout, int32_t Cin, int32_t N, int32_t stride, int32_t epilogue,
                   const float* aux_dev, int32_t aux_shift, const float* mean_dev, const float* std_dev,
                   int32_t out_padded, int32_t tile, void* stream);
/* The same with in_dev / wt_dev holding split-bf16 words (msr_op_split_bf16) and the bf16x3 arithmetic;
 * out_split != 0 (SPADE epilogue only) writes split-bf16 words too.  tile: (0 = 128x128 | 1 = 64x64 | 3, 4 = LDS-halo
 * forms | 5 = persistent ping-pong) + 0x40 (weights in MFMA-fragment order, tiles 0 / 1) + 0x80 (tile 5, SPADE epilogue
 * only: in_dev / wt_dev hold split-FP16 words and the products are the 2-term form of MSR_FLAG_GB_F16X2) + 256 * ksplit. */
int msr_op_conv3x3_bf16x3(msr_handle* h, const float* in_dev, const float* wt_dev, const float* bias_dev,
                          float* out_dev, int32_t B, int32_t rout, int32_t Cin, int32_t N, int32_t stride,
                          int32_t epilogue, const float* aux_dev, int32_t aux_shift, const float* mean_dev,
                          const float* std_dev, int32_t out_padded, int32_t out_split, int32_t tile, void* stream);
/* Kernel-level entry of the f16c form (MSR_FLAG_F16C) of the persistent ping-pong conv: in_dev / wt_dev hold f16c chunk
 * images (moonsuperresolution_amd.ops.f16c_activation_image / f16c_weight_image restate the format), wexp_dev [N] =
 * (127 + e_lo) | (127 + e_hi) << 8; out_mode (SPADE epilogue) 0 = fp32, 1 = split-bf16 words, 4 = f16c image. */
int msr_op_conv3x3_f16c(msr_handle* h, const float* in_dev, const float* wt_dev, const int32_t* wexp_dev,
                        const float* bias_dev, float* out_dev, int32_t B, int32_t rout, int32_t Cin, int32_t N,
                        int32_t epilogue, const float* aux_dev, int32_t aux_shift, const float* mean_dev,
                        const float* std_dev, int32_t out_padded, int32_t out_mode, void* stream);
/* Kernel-level entry of conv_gb_resident (csrc/conv_gbr.hip): ONE SPADE layer's modulation path in one launch —
 *   nearest resize of src to r x r + Conv2D(128, 3, relu) (spade.py:17-18), conv_gamma | conv_beta (spade.py:19-20) as one
 *   GEMM with N = 2C columns interleaved (32 gamma | 32 beta), gamma * (x - mean) / std + beta (spade.py:21-24), leaky_relu(0.2)
 *   (blocks.py:30-34), written as the f16c chunk image of the consumer conv.
 *   src_dev  [B, S, S, 2] fp32 (S = r * 2^k); we_dev HWIO [3,3,2,128]; be_dev [128]
 *   wt_dev   the kernel's weight stream (ops.gbr_weight_image restates it): the f16c6 image of [9][N][128] with the input
 *            channels of every 32-chunk in position order (position e holds channel 8 * (e >> 3) + 4 * (e & 1) + ((e >> 1) & 3)),
 *            re-ordered into [channel block][wave][tap pair][column block][piece][lane] x 16 bytes (csrc/conv_gbr.hip)
 *   bias_dev [N] in column order; aux_dev = x [B, r >> aux_shift, r >> aux_shift, N / 2]; mean_dev / std_dev [N / 2]
 *   out_dev  zero-bordered [B, r + 2, r + 2, N / 2] float slots, the interior is written
 * Needs r >= 32 (a power of two) and N % 128 == 0; MSR_ERR_INVALID otherwise. */
int msr_op_spade_gbr(msr_handle* h, const float* src_dev, int32_t S, const float* we_dev, const float* be_dev,
                     const float* wt_dev, const float* bias_dev, float* out_dev, int32_t B, int32_t r, int32_t N,
                     const float* aux_dev, int32_t aux_shift, const float* mean_dev, const float* std_dev, void* stream);
/* Kernel-level entry of the head kernel (csrc/small_kernels.hip head_kernel), synchronous:
 *   variant 0: leaky_relu(slope) -> UpSampling2D(2) -> Conv2D(1, 4, 'same') (networks.py:54-56), kernel_host = HWIO [4,4,C,1];
 *   variant 1: Conv2DTranspose(1, 4, strides 2, 'same') -> tanh (pix2pix.py:53-57; pass slope = 1), kernel_host = [4,4,1,C]
 *   (both are [kh][kw][C] in memory).  x_dev dense [B, r, r, C] (r, C multiples of 16); out_dev [B, 2r, 2r]. */
int msr_op_head(msr_handle* h, const float* x_dev, const float* kernel_host, float bias, float* out_dev, int32_t B,
                int32_t r, int32_t C, float slope, int32_t variant, void* stream);
/* HOST helper: the fp32 -> fp8 e4m3 (OCP "fn", round to nearest even, saturating at 448) conversion msr_load_weight
 * applies to the weights of the fp8 mode, exposed so that it can be checked against an independent implementation. */
int64_t msr_quantize_e4m3(const float* host, int64_t n, uint8_t* out);
/* Kernel-level entry of the fp8 form (MSR_FLAG_FP8) of the persistent ping-pong conv.
 *   in_dev   zero-bordered NHWC bf8 (e5m2) bytes [B, rout+2, rout+2, Cpad], Cpad = 128 or a multiple of 256 (channels beyond the real
 *            ones are zero);  wt_dev  fp8 (e4m3) bytes [9][N][Cpad];  wexp_dev [N] int32: the e8m0 exponent (127 + e,
 *            weight = byte value * 2^e) of every output channel, replicated in the word's four bytes
 *   out_mode (SPADE epilogue) 0 = fp32 [.., N/2], 1 = split-bf16 words, 3 = bf8 bytes, 128 channels or padded to a multiple of 256
 *   epilogue / aux / mean / std as msr_op_conv3x3. */
int msr_op_conv3x3_fp8(msr_handle* h, const void* in_dev, const void* wt_dev, const int32_t* wexp_dev,
                       const float* bias_dev, float* out_dev, int32_t B, int32_t rout, int32_t Cpad, int32_t N,
                       int32_t epilogue, const float* aux_dev, int32_t aux_shift, const float* mean_dev,
                       const float* std_dev, int32_t out_padded, int32_t out_mode, void* stream);
/* fp32 -> split-bf16 image: every aligned group of 32 values (one channel chunk; count % 32 == 0) becomes
 * [32 x hi bf16 | 32 x lo bf16], hi = bf16_rn(v), lo = bf16_rn(v - hi); size and addressing stay those of fp32. */
int msr_op_split_bf16(msr_handle* h, const float* in_dev, float* out_dev, int64_t count, void* stream);

/* Debug / per-block parity aid: copy a named workspace tensor of the last msr_forward to a HOST buffer
 * (names: "ws.gen.x0", "ws.gen.rb3.x1", "ws.gen.rb3.out", "ws.enc.mv", ...).  Synchronises the device. */
int msr_debug_tensor(msr_handle* h, const char* name, float* host_out, int64_t count);
/* Bytes of device memory held by the handle (weights + workspace). */
int msr_device_bytes(const msr_handle* h, int64_t* bytes);

#ifdef __cplusplus
}
#endif
#endif /* MOONSR_H */
