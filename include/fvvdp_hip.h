/*
 * fvvdp_hip.h -- C ABI of libfvvdp_hip.so: the FovVideoVDP per-frame visible-difference path on MI355X (gfx950).
 *
 * The reference (gfxdisp/FovVideoVDP v1.2.3) has no native/FFI layer: its boundary is the Python class
 * `pyfvvdp.fvvdp` (pyfvvdp/fvvdp.py:58).  The natural operator seams inside it are
 *     fvvdp_video_source_array._get_frame      pyfvvdp/video_source.py:180-208      (unpack + photometry + luminance)
 *     sliding window + temporal FIR            pyfvvdp/fvvdp.py:258-300
 *     fvvdp_contrast_pyr.decompose             pyfvvdp/fvvdp_lpyr_dec.py:246-273    (Gaussian/contrast pyramid)
 *     process_block_of_frames                  pyfvvdp/fvvdp.py:359-478            (CSF, masking, spatial pooling)
 * and each entry point below replaces one of them (cited per function).  A maintainer of the reference binds
 * these with ctypes; `fovvideovdp_amd/_native.py` is that binding.
 *
 * Conventions
 *   - plain C, no C++/torch types; every function returns 0 on success or a negative FVVDP_E* code and never
 *     throws; fvvdp_last_error() returns a thread-local message for the last failure.
 *   - pointers named d_* are DEVICE pointers (e.g. torch tensor.data_ptr() on PyTorch-ROCm); pointers named
 *     h_* are HOST pointers (small tables, copied during the call).
 *   - `stream` is a hipStream_t passed as void* (e.g. torch.cuda.current_stream().cuda_stream); all kernels are
 *     enqueued on it and nothing synchronises except where stated.
 *   - the context owns only its scratch (pyramid levels, tables, partial sums), allocated once in
 *     fvvdp_ctx_create (the heat-map functions add theirs on first use); per-call functions do not allocate
 *     otherwise.  One context per (device, stream); not thread-safe.
 *   - pyramid planes: P = 4 for video (test-sustained, ref-sustained, test-transient, ref-transient;
 *     pyfvvdp/fvvdp.py:293) or P = 2 for a still image (test, ref; pyfvvdp/fvvdp.py:251-253).
 */
#ifndef FVVDP_HIP_H
#define FVVDP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FVVDP_OK 0
#define FVVDP_EINVAL (-1)   /* bad argument */
#define FVVDP_EHIP (-2)     /* a HIP runtime call failed */
#define FVVDP_ENOMEM (-3)   /* scratch allocation failed */
#define FVVDP_ESTATE (-4)   /* call order violated (e.g. CSF table not set) */
#define FVVDP_EUNSUPPORTED (-5)   /* valid request outside this entry point's domain: use the general one */

#define FVVDP_MAX_BANDS 16
#define FVVDP_MAX_TAPS 256
#define FVVDP_LUT_N 32      /* knots per axis of the cached CSF LUT (pyfvvdp/csf_cache) */

typedef struct fvvdp_ctx fvvdp_ctx;

/* Calibration constants of the masking/pooling stage (pyfvvdp/fvvdp_data/fvvdp_parameters.json, read by
 * fvvdp.load_config pyfvvdp/fvvdp.py:113-145). */
typedef struct fvvdp_params {
    float mask_p;        /* 2.4                                   fvvdp.py:585 */
    float mask_q[2];     /* sustained, transient                  fvvdp.py:586 */
    float mask_k;        /* 10^mask_c                             fvvdp.py:555 */
    float beta;          /* spatial pooling exponent              fvvdp.py:467 */
    float sens_gain;     /* 10^(sensitivity_correction/20)        fvvdp.py:447 */
    float lbkg_min;      /* 0.1                                   fvvdp_lpyr_dec.py:265 */
    float contrast_max;  /* 1000                                  fvvdp_lpyr_dec.py:266 */
    float d_max;         /* 1e4                                   fvvdp.py:595 */
} fvvdp_params;

/* Source sample types accepted by fvvdp_temporal_channels (pyfvvdp/video_source.py:184-200). */
enum { FVVDP_U8 = 0, FVVDP_U16 = 1, FVVDP_F32 = 2 };

/* Display photometry applied per colour channel (pyfvvdp/fvvdp_display_model.py:147-165, :203-212). */
enum {
    FVVDP_EOTF_LUT = 0,       /* integer sources only: L = d_lut[code], table built by the caller (required for
                               * uint8; uint16 may also use a closed-form kind, evaluated on code / 65535)    */
    FVVDP_EOTF_SRGB = 1,      /* (Y_peak-Y_black)*srgb2lin(V)+Y_black                 :155-156, :17-19        */
    FVVDP_EOTF_GAMMA = 2,     /* (Y_peak-Y_black)*V^gamma+Y_black                     :157-158                */
    FVVDP_EOTF_PQ = 3,        /* clip(pq2lin(V),0.005,Y_peak)+Y_black                 :159-160, :100-112      */
    FVVDP_EOTF_LINEAR = 4,    /* clip(V,0.005,Y_peak)+Y_black                         :161-162                */
    FVVDP_EOTF_ABSOLUTE = 5,  /* clamp(V,L_min,L_max)  (fvvdp_display_photo_absolute) :203-212                */
    FVVDP_EOTF_NONE = 6       /* source already holds luminance in cd/m^2 (custom video sources)             */
};

typedef struct fvvdp_eotf {
    int32_t kind;
    float Y_peak;
    float Y_black;
    float gamma;
    float L_min, L_max;      /* FVVDP_EOTF_ABSOLUTE: the clamp.  FVVDP_EOTF_LUT (optional): smallest / largest entry of the
                              * table; with it the library can prove that clamps of the pyramid pass never bind on this
                              * content and drop them (same results).  A stated range (L_max > L_min >= 0, finite) is
                              * ENFORCED: the kernels clamp every table entry they read to [L_min, L_max], the identity for
                              * a table that keeps its word, so a wrong statement yields the results of the clamped table,
                              * never an out-of-range CSF query.  Anything else (e.g. 0, 0) = not stated: entries pass as
                              * they are and the pyramid pass keeps its clamps */
    const float* d_lut;      /* FVVDP_EOTF_LUT: 256 (U8) or 65536 (U16) luminances per code value */
} fvvdp_eotf;

/* Display geometry for foveated mode (pyfvvdp/fvvdp_display_model.py:383-526, pyfvvdp/fvvdp.py:416-437). */
typedef struct fvvdp_geom {
    float display_size_m[2];  /* width, height in metres              :395,417,420,430 */
    float distance_m;         /* viewing distance                     :401-408        */
    float ppd_centre;         /* pixels per degree at the centre      :436            */
} fvvdp_geom;

/* Optional per-band map outputs of fvvdp_bands_forward (any pointer may be NULL).  Planar fp32, device memory:
 *   d_D        [n][2][h_b][w_b]  difference map after masking, one plane per temporal channel (fvvdp.py:454)
 *   d_contrast [n][P][h_b][w_b]  contrast band times the band multiplier (lpyr.get_band, fvvdp_lpyr_dec.py:57-63)
 *   d_lbkg     [n][h_b][w_b]     background luminance (fvvdp_lpyr_dec.py:265)
 *   d_S        [n][2][h_b][w_b]  sensitivity before the gain (cached_sensitivity, fvvdp.py:520-537)             */
typedef struct fvvdp_band_maps {
    float* d_D;
    float* d_contrast;
    float* d_lbkg;
    float* d_S;
} fvvdp_band_maps;

/* ---- lifecycle ------------------------------------------------------------------------------------------ */

/* Replaces the per-resolution setup in predict_video_source (pyfvvdp/fvvdp.py:209-213: the pyramid object) and
 * the per-frame tensor allocations of the hot loop.  `n_bands` = lpyr.height (number of band-pass levels; the
 * Gaussian pyramid has n_bands+1 levels, fvvdp_lpyr_dec.py:15-49).  `max_frames` = number of frames whose
 * pyramids are resident at once (frame batch).  h_rho_band: n_bands+1 band frequencies in cpd. */
int fvvdp_ctx_create(fvvdp_ctx** out, int width, int height, int n_bands, int planes, int max_frames,
                     const double* h_rho_band, const fvvdp_params* prm);
void fvvdp_ctx_destroy(fvvdp_ctx* ctx);
const char* fvvdp_last_error(void);
/* Level geometry: level 0 is the frame; sizes follow ceil(/2) (fvvdp_lpyr_dec.py:198). */
int fvvdp_ctx_level_size(const fvvdp_ctx* ctx, int level, int* w, int* h);
/* Bytes of device scratch held by the context. */
size_t fvvdp_ctx_scratch_bytes(const fvvdp_ctx* ctx);

/* ---- CSF tables ------------------------------------------------------------------------------------------ */

/* Non-foveated mode: rho and eccentricity are constant per band, so the 32^3 LUT collapses exactly to a 1-D
 * table over log2(L_bkg) per (band, temporal channel) -- the rho/ecc interpolation of interp3
 * (pyfvvdp/interp.py:43-59) is done once by the caller with the same fp32 operations.
 *   h_Y_log  [32]               knots of the luminance axis (log2 cd/m^2)
 *   h_S_log  [n_bands][2][32]   log2 sensitivity at the knots                                           */
int fvvdp_ctx_set_csf_1d(fvvdp_ctx* ctx, const float* h_Y_log, const float* h_S_log);

/* Foveated mode: full LUT of one temporal channel (cached_sensitivity + interp3, pyfvvdp/fvvdp.py:520-537).
 *   h_S_log [32(Y)][32(rho)][32(ecc)], axes h_Y_log, h_rho_log, h_ecc_sqrt (32 each).                  */
int fvvdp_ctx_set_csf_3d(fvvdp_ctx* ctx, int temporal_channel, const float* h_S_log, const float* h_Y_log,
                         const float* h_rho_log, const float* h_ecc_sqrt);

/* Foveated mode with a USER geometry model (a subclass of fvvdp_display_geometry with its own pix2view_direction /
 * get_ppd, e.g. pytorch_examples/ex_custom_ppd.py:38-57): the caller evaluates, once per band, the view direction
 * of every band pixel and the resolution magnification with the user's object (fvvdp.py:424-437) and hands the maps
 * over; fvvdp_bands_forward is then called with geom == NULL and h_fixation holding the gaze VIEW DIRECTION in
 * degrees for each slot.  d_view_x, d_view_y, d_res_mag: [h_b][w_b] fp32, must stay alive; res_mag_min/max = range of
 * the magnification map (bounds the rho slice of the LUT).  Passing NULL maps clears map mode.             */
int fvvdp_ctx_set_view_maps(fvvdp_ctx* ctx, int band, const float* d_view_x, const float* d_view_y,
                            const float* d_res_mag, float res_mag_min, float res_mag_max);

/* ---- stage 1: frames -> temporal channels --------------------------------------------------------------- */

/* Replaces fvvdp_video_source_array._get_frame (video_source.py:180-208), fvvdp_display_photo_eotf.forward
 * (fvvdp_display_model.py:147-165) and the sliding-window temporal filter (fvvdp.py:258-300) for `n_out`
 * consecutive output frames; the result is written into pyramid level 0 of slots [slot0, slot0+n_out).
 *   d_test, d_ref   source videos, element (c, f, y, x) at  c*chan_stride + f*frame_stride + y*width + x
 *   dtype, C        FVVDP_U8/U16/F32; C in {1,3}
 *   h_rgb2y[3]      luminance weights (color_spaces.json RGB2Y), ignored when C == 1
 *   h_frame_idx     [fl-1+n_out] source frame of every virtual time step: the first fl-1 entries are the
 *                   history before the first output (temporal padding, or real frames when continuing a
 *                   video / a frame shard), entry fl-1+t is the newest frame of output t
 *   h_taps          [2][fl] temporal filters (get_temporal_filters, fvvdp.py:609-630); tap k weights the frame
 *                   k steps in the past.  planes == 2 (still image): fl must be 1 and only h_taps[0] is used.
 *   d_oob_flag      optional int: set to 1 if a float sample was outside [0,1] (the caller re-emits the
 *                   reference's warning "Pixel outside the valid range 0-1")
 * Asynchronous.  Filters of up to 32 taps (64 taps, i.e. up to 256 fps, for uint8 sources, 16-bit / float RGB behind an sRGB or
 * PQ display model and float luminance) run on the register-ring kernels in one pass; the other sample types / display models
 * with 33..64 taps take two passes (every source frame of the window -> fp32 luminance once, into a context-owned buffer that is
 * allocated on first use, then the 64-slot ring on those frames); more than 64 taps (above 256 fps) take a generic kernel that
 * re-reads the window per output frame (an order of magnitude slower) and, like more than 320 window entries in a still-image
 * context, synchronises the stream for two small table uploads.                                                   */
int fvvdp_temporal_channels(fvvdp_ctx* ctx, const void* d_test, const void* d_ref, int dtype, int C,
                            size_t chan_stride, size_t frame_stride, const fvvdp_eotf* eotf,
                            const float* h_rgb2y, const int32_t* h_frame_idx, const float* h_taps, int fl,
                            int n_out, int slot0, int32_t* d_oob_flag, void* stream);

/* The same for sources whose frames are SEPARATE allocations (what fvvdp_video_source.get_test_frame / get_reference_frame
 * of a user class return, video_source.py:14-36): h_test_frames / h_ref_frames are host arrays of n_frames device
 * pointers, one frame [C][H*W] each (channel c at c*chan_stride); h_frame_idx indexes those arrays.  No staging copy:
 * the kernels read the frames where they are.  Video contexts with fl <= 32; returns FVVDP_EUNSUPPORTED when the frames
 * lie more than 2^31 elements (x4 when every distance is a multiple of 4 elements) apart, or for other contexts -- the
 * caller then copies the frames into one array and uses fvvdp_temporal_channels.                                   */
int fvvdp_temporal_channels_frames(fvvdp_ctx* ctx, const void* const* h_test_frames, const void* const* h_ref_frames,
                                   int n_frames, int dtype, int C, size_t chan_stride, const fvvdp_eotf* eotf,
                                   const float* h_rgb2y, const int32_t* h_frame_idx, const float* h_taps, int fl, int n_out,
                                   int slot0, int32_t* d_oob_flag, void* stream);

/* Raw planar YUV sources (what video_reader_yuv_pytorch receives from ffmpeg's rawvideo pipe,
 * video_source_file.py:166-217): every frame is the Y plane [H][W] followed by the U and V planes ([H/2][W/2] for
 * 4:2:0, [H][W] for 4:4:4), uint8 for bit_depth 8, little-endian uint16 above. */
typedef struct fvvdp_yuv_format {
    int32_t bit_depth;       /* 8..16                                        video_source_file.py:196-197       */
    int32_t chroma_420;      /* 1 = 4:2:0 (bilinear x2 chroma upsampling), 0 = 4:4:4   :199-201, :268-272       */
    float ycbcr2rgb[9];      /* row-major 3x3, BT.709 or BT.2020nc              :225-235                          */
} fvvdp_yuv_format;

/* Replaces video_reader_yuv_pytorch.unpack + _fixed2float_upscale (video_source_file.py:219-276),
 * fvvdp_video_source_video_file._prepare_frame (:355-363) and the temporal filter (fvvdp.py:258-300): like
 * fvvdp_temporal_channels, but the source frames are raw planar YUV.  frame_stride in elements; the display model
 * must be closed-form (kind != FVVDP_EOTF_LUT); full-screen resizing: fvvdp_yuv_frame_resized below.  Up to 64 taps (256 fps): 1..32 taps in
 * one pass, 33..64 in two (YUV frames -> luminance frames once, then the 64-slot ring), FVVDP_EINVAL above. */
int fvvdp_temporal_channels_yuv(fvvdp_ctx* ctx, const void* d_test, const void* d_ref, const fvvdp_yuv_format* fmt,
                                size_t frame_stride, const fvvdp_eotf* eotf, const float* h_rgb2y,
                                const int32_t* h_frame_idx, const float* h_taps, int fl, int n_out, int slot0,
                                int32_t* d_oob_flag, void* stream);

/* Alternative entry for callers that already hold the temporal channels in the reference layout
 * R[n][P][H][W] (planar fp32, fvvdp.py:294): copies them into pyramid level 0 of slots [slot0, slot0+n). */
int fvvdp_load_channels_planar(fvvdp_ctx* ctx, const float* d_R, int n, int slot0, void* stream);

/* ---- stage 2: pyramid + CSF + masking + spatial pooling --------------------------------------------------- */

/* Replaces process_block_of_frames (fvvdp.py:359-478) incl. lpyr.decompose (fvvdp_lpyr_dec.py:248-273),
 * cached_sensitivity (fvvdp.py:520-537), apply_masking_model (fvvdp.py:574-596) and lp_norm (fvvdp.py:598-607)
 * for the frames in slots [0, n).
 *   d_Q           output, Q_per_ch[band][cc][q_stride] fp32; frame slot s is written at column q_col0+s.
 *                 cc=1 is written as 0 for planes == 2 (fvvdp.py:465).
 *   h_fixation    NULL = non-foveated; else [n][2] gaze (x,y) in frame pixels for each slot (fvvdp.py:417-431)
 *   geom          required when h_fixation != NULL
 *   maps          NULL or array of n_bands structs with optional per-band map outputs                     */
int fvvdp_bands_forward(fvvdp_ctx* ctx, int n, float* d_Q, int q_stride, int q_col0, const float* h_fixation,
                        const fvvdp_geom* geom, const fvvdp_band_maps* maps, void* stream);

/* ---- difference-map (heat-map) reconstruction --------------------------------------------------------------- */

/* Replaces the heat-map branch of process_block_of_frames (fvvdp.py:374-375,458-471) with heatmap_pyr.set_band /
 * reconstruct (fvvdp_lpyr_dec.py:65-71,94-101): per band b the map (D_sustained + w_transient*D_transient)/band_mul
 * is accumulated by expand-and-add from the coarsest band up, then dmap = img^beta_jod * |jod_a|.
 *   h_dD      host array of n_bands DEVICE pointers: the d_D maps written by fvvdp_bands_forward for the same n
 *             frames ([n][2][h_b][w_b]; for planes == 2 only channel 0 is used)
 *   d_out     [n][H][W] fp32 (the caller converts to fp16 like the reference, fvvdp.py:473)                  */
int fvvdp_heatmap_reconstruct(fvvdp_ctx* ctx, int n, const float* const* h_dD, float w_transient, float beta_jod,
                              float jod_a_abs, float* d_out, void* stream);

/* ---- stage 3: pooling over bands / channels / frames and the JOD regression ------------------------------------ */

/* Pooling + JOD regression on the device (reference: do_pooling_and_jods, fvvdp.py:337-357, lp_norm :598-607):
 *   Q_sc[c,f] = (sum_b |Q[b,c,f]*w_c|^beta_sch)^(1/beta_sch), w = (1, w_transient) for video
 *   Q_tc[f]   = (sum_c Q_sc^beta_tch)^(1/beta_tch);  Q = (mean_f Q_tc^beta_t)^(1/beta_t)
 *   JOD       = sign(jod_a) * (|jod_a|^(1/beta_jod) * Q)^beta_jod + 10
 * d_Q is the [n_bands][2][q_stride] array written by fvvdp_bands_forward (n_channels = 1 for images: plane 1 unused).
 * Saves the dozen tiny framework kernels of the Python version at the end of every call; the Python method stays for
 * callers that pool their own Q_per_ch (frame / pair sharding).                                                   */
typedef struct {
    float beta_sch, beta_tch, beta_t;   /* fvvdp_parameters.json: 1, 0.666092, 1 */
    float w_transient;                  /* 0.25 */
    float jod_a, beta_jod;              /* -0.0161713, 10^log_jod_exp */
} fvvdp_pool_params;
int fvvdp_pool_jod(const float* d_Q, int n_bands, int n_channels, int n_frames, int q_stride,
                   const fvvdp_pool_params* prm, float* d_jod, void* stream);

/* fvvdp_bands_forward (process_block_of_frames, fvvdp.py:359-478) and, when this batch completes the clip (q_col0 + n ==
 * q_stride), fvvdp_pool_jod (do_pooling_and_jods, fvvdp.py:337-357) over all q_stride frames of d_Q in the same call:
 * the result in d_jod[0] is bit-identical to the two separate calls (one launch and no host round trip more than
 * fvvdp_bands_forward).  Earlier batches of the clip (q_col0 + n < q_stride) only run the bands.                    */
int fvvdp_bands_forward_pool(fvvdp_ctx* ctx, int n, float* d_Q, int q_stride, int q_col0, const float* h_fixation,
                             const fvvdp_geom* geom, const fvvdp_band_maps* maps, const fvvdp_pool_params* pool,
                             float* d_jod, void* stream);

/* Colouring of difference maps for heatmap = "threshold" / "supra-threshold" (reference: visualize_diff_map,
 * vis_tonemap, log_luminance in pyfvvdp/visualize_diff_map.py, called at fvvdp.py:474-476): the map d_dmap[n][H][W]
 * (clamped to [0,1]) indexes a colour map (n_knots <= 8 knots h_knots, luminance-normalised colours h_rgb[n_knots][3])
 * and modulates a tone-mapped copy of the context frame = plane 0 of pyramid level 0 of slots [0,n) (histogram tone
 * curve per frame, 1024 bins; h_lin01 = torch.linspace(0,1,1024), the knot positions of the curve).  Output fp16,
 * element (c,f,y,x) at c*chan_stride + (f*H + y)*W + x.  Asynchronous except for the first call on a context.  */
int fvvdp_heatmap_colorize(fvvdp_ctx* ctx, int n, const float* d_dmap, const float* h_knots, const float* h_rgb,
                           int n_knots, const float* h_lin01, void* d_out_f16, size_t chan_stride, void* stream);

/* ---- introspection (tests, profiling) --------------------------------------------------------------------- */

/* Copy Gaussian level `level` of slots [0,n) to planar fp32 d_out[n][P][h][w] (gaussian_pyramid_dec,
 * fvvdp_lpyr_dec.py:144-158).  Levels that fvvdp_bands_forward kept on the chip (the middle level of a two-levels-per-
 * launch pass: large levels in plain evaluation; never with difference maps requested or FVVDP_BAND_FUSE=0 in the
 * environment) are not materialised and hold stale data. */
int fvvdp_export_level(fvvdp_ctx* ctx, int level, int n, float* d_out, void* stream);

/* Per-kernel HIP-event timing of the most recent calls.  When enabled, every kernel launch is bracketed by
 * events on the caller's stream.  fvvdp_ctx_timing_read synchronises the events and returns, per kernel id,
 * accumulated milliseconds and launch counts since the last reset.
 *   ids: 0 = temporal, 1..n_bands = band kernel of level id-1, n_bands+1 = finalize                       */
int fvvdp_ctx_timing_enable(fvvdp_ctx* ctx, int on);
int fvvdp_ctx_timing_read(fvvdp_ctx* ctx, float* h_ms, int32_t* h_count, int capacity, int reset);

/* Where level 0 of the context's scratch lives and how that was decided.  The large scratch levels are mapped from physical
 * chunks (HIP virtual-memory API).  How fast a buffer of several GB can be WRITTEN depends on the physical memory behind it: a box's
 * memory comes in two classes, streaming writes into one class top out at ~5.5 TB/s, into both at once at 7.0 (the temporal kernel:
 * 35-36 or 30-31 us per 4K frame; profiles/r05_k1_mode.md), and which class an allocation gets is the driver's business.  Level 0 of
 * a video context that holds >= 1 GiB therefore lives in TWO ranges -- even frame slots in one, odd slots in the other -- chosen by
 * fvvdp_ctx_create among N half-size candidates (default 6: chunk-mapped and hipMalloc in turn;
 * environment FVVDP_PLACEMENT_PROBE=n, 0 / 1 = one range as allocated): every pair is written at once by a streaming-write probe and
 * the pair with the highest rate is kept; if no pair reaches the rate of two different classes (6.75 TB/s), up to 8 further candidates
 * are taken while the first ones are held (FVVDP_PLACEMENT_EXTRA=n), each written together with one range of the best pair, until one
 * does.  ~0.1 s at creation (0.3 s with the further candidates), the candidates' memory held for the moment, nothing in any per-frame
 * call.  Results never depend on it.
 *   *state: always 9 (settled);  *chunk_mapped: one range: its kind (0 hipMalloc, 1 mapped from physical chunks); two ranges: 100 + 10 * kind of the odd slots' range + kind of the even slots' range
 *   h_us[capacity]: [0] always 0 (until round 5: a timing of the layout on a synthetic clip at creation; dropped, it had to borrow the
 *   context's tables), [1] / [2] highest / lowest streaming-write rate [TB/s] over the candidate pairs, [3] number of further candidates
 *   tried, 0 beyond.  The original range is given back only once two candidates are in hand: a probe that cannot allocate leaves the
 *   context with the range it came with.
 *   *n_timed: number of half-size candidates (0: no choice);  *kept: index of the even slots' candidate + 8 * index of the odd slots' (-1: no choice) */
int fvvdp_ctx_alloc_info(const fvvdp_ctx* ctx, int* state, int* chunk_mapped, float* h_us, int capacity, int* n_timed, int* kept);

/* Host synchronisations, device allocations and frees made INSIDE per-frame entry points (fvvdp_temporal_channels*,
 * fvvdp_bands_forward*, heat-map functions) since the context was created: h_counts3 = {syncs, allocations, frees}.  The video
 * path of the BASELINE configs makes none, from the first call on ("allocated once in ctx_create; no hidden
 * allocation in per-frame calls").  What does count: first use of an optional path (foveated tables when the geometry
 * changes, the luminance frames of a 33..64-tap filter, heat-map images, the colouring workspace).                      */
int fvvdp_ctx_call_stats(const fvvdp_ctx* ctx, int64_t* h_counts3);


/* ---- PU21-PSNR side metric ---------------------------------------------------
 * Replaces the per-frame body of pu_psnr.predict_video_source (pyfvvdp/pupsnr.py:64-76): luminance of both streams
 * through the same ingest as fvvdp_temporal_channels (video_source.py:180-208), PU.encode (pyfvvdp/utils.py:183-193,
 * clip to [L_min, L_max], V = p6*(((p0 + p1*Y^p3)/(1 + p2*Y^p3))^p4 - p5)) and the sum of squared differences of
 * each frame.  Stateless (no context): d_sse[f] = sum over the n_pixels pixels of frame f of (V_test - V_ref)^2 in
 * fp64, summed in a fixed order; d_partial is workspace of n_frames * FVVDP_PSNR_SLICES doubles.  The caller turns
 * it into dB: mean over frames of 20*log10(peak / sqrt(sse / n_pixels))  (psnr_fn, pupsnr.py:78-79).          */
#define FVVDP_PSNR_SLICES 256
typedef struct {
    float p[7];              /* PU21 parameters (utils.py:171-178, 'banding_glare' by default) */
    float L_min, L_max;      /* 0.005, 10000 */
} fvvdp_pu21;
int fvvdp_pu21_sse(const void* d_test, const void* d_ref, int dtype, int channels, size_t chan_stride,
                   size_t frame_stride, size_t n_pixels, const fvvdp_eotf* eotf, const float* h_rgb2y,
                   const fvvdp_pu21* pu, int n_frames, double* d_partial, double* d_sse, int32_t* d_oob_flag,
                   void* stream);


/* ---- Full-screen resize of planar YUV frames ("optional interpolate resize") ---------
 * Replaces video_reader_yuv_pytorch.unpack WITH resize_fn (pyfvvdp/video_source_file.py:219-244: unpack to RGB without
 * clipping, torch.nn.functional.interpolate(RGB, size=(out_h, out_w), mode=..., align_corners=False), clip to [0,1]) and
 * fvvdp_video_source_video_file._prepare_frame (:355-363: display model, RGB -> luminance) for ONE frame of ONE stream --
 * what the CLI's --full-screen-resize does per frame (run_fvvdp.py:84, :209-210).  Stateless.  d_frame: Y, U, V planes as
 * in fvvdp_temporal_channels_yuv; d_rgb_scratch: workspace of 3*W*H floats (the unclipped RGB planes at the source
 * resolution); d_lum: out_h*out_w floats, the luminance frame a video source hands to the metric (get_*_frame); d_rgb_out:
 * optional 3*out_h*out_w floats, the clipped RGB planes [3][out_h][out_w] (= unpack's return value, for parity tests).
 * The display model must be closed-form (kind != FVVDP_EOTF_LUT).  'area' is adaptive average pooling, as in torch.     */
enum { FVVDP_RESIZE_NEAREST = 0, FVVDP_RESIZE_BILINEAR = 1, FVVDP_RESIZE_BICUBIC = 2, FVVDP_RESIZE_AREA = 3 };
int fvvdp_yuv_frame_resized(const void* d_frame, const fvvdp_yuv_format* fmt, int W, int H, float* d_rgb_scratch,
                            int out_w, int out_h, int mode, const fvvdp_eotf* eotf, const float* h_rgb2y,
                            float* d_lum, float* d_rgb_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FVVDP_HIP_H */
