/* mi355fx.h — C ABI of the MI355X (gfx950) kernel library behind gst-plugins-rs' per-buffer
 * DSP elements (hsvfilter, hsvdetector, colorlut, rsaudioecho).
 *
 * This header is the drop-in boundary: each entry point replaces exactly one inner loop of the
 * reference (file:line cited per function, relative to the gst-plugins-rs tree) and takes what
 * that loop takes — plane pointer, stride, width/height, format, the settings snapshot.
 * The reference element keeps its GObject/BaseTransform surface and calls these through
 * `extern "C"` FFI (binding shown in INTEGRATION.md). Plain pointers and sizes only.
 *
 * Conventions
 *   - Every function returns MI355_OK (0) or a negative mi355_status; mi355_ctx_last_error()
 *     gives a human-readable message for the last failure on that context. The element maps any
 *     non-zero status to GST_FLOW_ERROR (transform vfuncs) or an ErrorMessage (start/setup).
 *   - One context = one element instance. A context is used by one thread at a time (the
 *     streaming thread, serialised by the pad stream lock); different contexts are independent.
 *   - "host" entry points borrow the caller's buffer for the duration of the call only
 *     (GstVideoFrame / GstBuffer map semantics): H2D copy, kernel, D2H copy, then return.
 *   - "_device" entry points take device pointers, enqueue on the context's stream and return
 *     without synchronising (adjacent mi355 elements, benchmarks).
 *   - There is no CPU fallback: without a usable gfx950 device every call fails loudly.
 */
#ifndef MI355FX_H
#define MI355FX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355FX_ABI_VERSION 1

typedef enum mi355_status {
  MI355_OK = 0,
  MI355_ERR_INVALID_ARG = -1,   /* bad pointer / size / format            */
  MI355_ERR_NO_DEVICE = -2,     /* no gfx950 device, or HIP init failed   */
  MI355_ERR_HIP = -3,           /* a HIP runtime call or kernel failed    */
  MI355_ERR_NOT_CONFIGURED = -4,/* e.g. colorlut without a loaded LUT (colorlut/imp.rs:209-212),
                                   echo before setup (audioecho/imp.rs:210 NotNegotiated) */
  MI355_ERR_OUT_OF_MEMORY = -5,
  MI355_ERR_UNSUPPORTED = -6,
  MI355_ERR_TIMEOUT = -7        /* (reserved: mi355_agroup_wait of a lock-step group returned it; the members of every audio group are independent now) */
} mi355_status;

/* Packed-RGB formats of the hot path. Values are stable ABI.
 * hsvfilter caps: video/hsv/src/hsvfilter/imp.rs:274-312; colorlut caps: video/colorlut/src/colorlut/imp.rs:118-157. */
typedef enum mi355_video_format {
  MI355_FMT_RGBX = 0,
  MI355_FMT_XRGB = 1,
  MI355_FMT_BGRX = 2,
  MI355_FMT_XBGR = 3,
  MI355_FMT_RGBA = 4,
  MI355_FMT_ARGB = 5,
  MI355_FMT_BGRA = 6,
  MI355_FMT_ABGR = 7,
  MI355_FMT_RGB = 8,
  MI355_FMT_BGR = 9,
  MI355_FMT_RGBA64_LE = 10,
  MI355_FMT_RGBA64_BE = 11
} mi355_video_format;

typedef struct mi355_ctx mi355_ctx;

/* ---------------------------------------------------------------- context / plumbing */

int mi355_abi_version(void);
/* Number of visible HIP devices (0 if none or the runtime cannot initialise). */
int mi355_device_count(void);
/* Create a context on `device` (its own non-blocking stream). NULL on failure; *status set if non-NULL. */
mi355_ctx *mi355_ctx_create(int device, int *status);
void mi355_ctx_destroy(mi355_ctx *ctx);
const char *mi355_ctx_last_error(const mi355_ctx *ctx);
const char *mi355_status_string(int status);
/* hipStream_t of the context (as void*). */
void *mi355_ctx_stream(mi355_ctx *ctx);
/* Use an externally owned stream (e.g. the stream of the neighbouring element) instead. */
int mi355_ctx_set_stream(mi355_ctx *ctx, void *hip_stream);
int mi355_ctx_synchronize(mi355_ctx *ctx);

/* Diagnostic switches (tests, A/B measurements). MI355_FLAG_FORCE_GENERIC=1 makes every element use
 * its literal-arithmetic GENERIC kernel instead of the strength-reduced FAST one. */
typedef enum mi355_flag {
  MI355_FLAG_FORCE_GENERIC = 1,
  MI355_FLAG_HSV_BLOCKS_PER_CU = 2, /* grid cap (blocks per CU) of the streaming hsvfilter kernel; tuning knob */
  /* colorlut kernel choice for packed RGBA8 frames. 0 = auto (default): the interpolating kernel and the 2^24-entry
   * memoised-table kernel (built on the device from the interpolating kernel, so bit-identical) are both timed on the
   * first four launches after a LUT load; afterwards the kind in use is re-timed every 8th-32nd launch and the other
   * one every 64-1024 launches, and the faster one serves the launches in between. mi355_hsv_colorlut_* does the same
   * with a table of the composed function. 6 = interpolating kernels only: the brick-cache kernel (colorlut_brick.hip),
   * with noise-like streams handed to the three-pass whole-plane kernel by a miss-counter watch that never blocks;
   * 7 = brick-cache kernel only; 3 = three-pass kernel only; 1 / 2 = the three-pass kernel's late-prefetch / lean-state
   * forms (tuning experiments); 4 / 5 = table through the gather kernels only, linear / Morton table index; 8 = Morton table
   * read through a block-shared LDS cache of table bricks (colorlut_window.hip) where the launch is large enough; 9 =
   * Morton table, 5's or 8's kernel by measurement. Auto builds the Morton table and picks as 9 does. Auto records and queries events on
   * the context's stream: pin a variant (7, 3, 5 or 8) before capturing that stream into a hipGraph. */
  MI355_FLAG_LUT_VARIANT = 4,
  /* hsvfilter on packed colour-first 4-byte frames through a memoised table: 0 (default) = auto choice as for colorlut,
   * but only for settings that need the literal GENERIC arithmetic (|hue-shift| > 360 or non-finite); 1 = auto choice
   * for all settings; 2 = table only; 3 = arithmetic kernels only */
  MI355_FLAG_HSV_TABLE = 6,
  MI355_FLAG_LUT_STAGGER = 5,  /* colorlut 3D LDS kernel: spread of the per-block start delay in units of 256 clock ticks (0 = off) */
  MI355_FLAG_FUSED_VARIANT = 3, /* fused hsv+colorlut tiling: 0 = hsv inline after the load (default); 1 = software-pipelined kernel */
  MI355_FLAG_BRICK_TILES_PER_RUN = 7, /* brick-cache kernel: tiles (128 pixels x 4 or 8 rows) in a run, the part of a strip one wave owns (0 = default: one run per wave of the chip) */
  MI355_FLAG_DSSIM_TRANSLUCENT = 11, /* Dssim on RGBA pixels with alpha < 255: 0 (default) = composed over the crate's coloured, position-dependent pattern (mi355_dssim_create_image), 1 = over black (premultiplied values as they are) */
  MI355_FLAG_BRICK_FOLD_AXIS = 10, /* accepted and ignored: the 32-set geometry of the brick-cache kernel is hashed over all three axes now (it used to give one axis 2 set residues instead of 4) */
  MI355_FLAG_BRICK_PRIO = 9, /* brick-cache kernel, how the waves of a block share work: bit 0 = waves lower their issue priority as they advance through their run, bit 1 = a wave that is done takes tiles from the run with most left (default 3) */
  MI355_FLAG_HRTF_METHOD = 12, /* hrtfrender convolution, read at mi355_hrtf_setup: 0 (default) = overlap-save FFT in LDS from 384-tap HRIRs on (the measured crossover), time-domain FIR below; 1 = FFT, 2 = FIR pinned (each only where it fits the LDS) */
  MI355_FLAG_BLOCKHASH_ANY_SIZE = 15, /* videocompare Blockhash on frames whose width or height is not a multiple of 8: 0 (default) = MI355_ERR_UNSUPPORTED, 1 = the crate's floating-point path (blockhash_slow: every pixel whole to block (floor(x / (w/8)), floor(y / (h/8))) in f32, block sums accumulated in pixel order - one lane per block, sequential by definition, 1-2 ms per 4K frame; restated from memory like the rest of the hash: parity unpinned) */
  MI355_FLAG_HSV_NT = 14, /* hsvfilter on packed 4-byte frames: 1 = loads and stores carry the non-temporal hint. The kernel alone is ~5 % faster, but its output then bypasses the Infinity Cache and the element behind it reads from HBM (bench.py's `hsvfilter_nontemporal_ab` leg measures exactly that); default 0 */
  MI355_FLAG_WINDOW_ORDER = 17, /* LDS-cached table kernel: how its blocks share the picture: 1 (default) = aligned fronts (block b takes strip b % n_strips, layer b / n_strips; the blocks of a layer walk down side by side, so the chip streams a few bands of full rows), 0 = a contiguous share of the column-major step list each (round 4); A/B knob */
  MI355_FLAG_WINDOW_STATS = 18, /* 1 = the LDS-cached table kernels count pixels / pixels served past the cache / bricks installed for mi355_colorlut_window_stats (three atomics per wave; default 0) */
  MI355_FLAG_WINDOW_MIN_STEPS = 13, /* LDS-cached table kernel (LUT variants 0 / 8): smallest launch it serves, in 256 x 32 pixel steps per CU (default 3; 0 = any size - its first step per block runs on a cold cache, so small launches are faster through the gather kernels) */
  MI355_FLAG_BRICK_SETS = 8 /* brick-cache kernels: 0 (default) = chosen by the content watch; pinned: 32 (16 waves per CU) or 64 (8 waves per CU) sets per WAVE cache, 512 = the block-shared cache of colorlut3d_shared_kernel (512 sets per CU, RGBA8 plain colorlut, launches of at least 3 steps of 256 x 32 pixels per CU; smaller ones take the 32-set kernel); two ways each */
} mi355_flag;
int mi355_ctx_set_flag(mi355_ctx *ctx, int flag, int value);

void *mi355_device_alloc(mi355_ctx *ctx, size_t bytes);
int mi355_device_free(mi355_ctx *ctx, void *dptr);
int mi355_memcpy_h2d(mi355_ctx *ctx, void *dptr, const void *host, size_t bytes); /* synchronous */
int mi355_memcpy_d2h(mi355_ctx *ctx, void *host, const void *dptr, size_t bytes); /* synchronous */

/* ---------------------------------------------------------------- hsvfilter
 * Replaces HsvFilter::hsv_filter + the format match of transform_frame_ip
 * (video/hsv/src/hsvfilter/imp.rs:76-120, :323-376) and the hsvutils conversions it calls
 * (video/hsv/src/hsvutils.rs:44-198). Settings snapshot = `Settings` (hsvfilter/imp.rs:33-39). */
typedef struct mi355_hsv_settings {
  float hue_shift;      /* "hue-shift"      default 0.0 */
  float saturation_mul; /* "saturation-mul" default 1.0 */
  float saturation_off; /* "saturation-off" default 0.0 */
  float value_mul;      /* "value-mul"      default 1.0 */
  float value_off;      /* "value-off"      default 0.0 */
} mi355_hsv_settings;

/* In place on plane 0 of a mapped frame in host memory. `data_len` = plane_data_mut(0).len();
 * rows processed = data_len / stride (chunks_exact_mut drops a trailing partial row). */
int mi355_hsvfilter_frame_ip(mi355_ctx *ctx, uint8_t *data, size_t data_len, int width, int stride,
                             int format, const mi355_hsv_settings *settings);
/* Same on `n_frames` device-resident frames, frame f at d_data + f*frame_pitch. Asynchronous. */
int mi355_hsvfilter_frames_device(mi355_ctx *ctx, uint8_t *d_data, int n_frames, size_t frame_pitch,
                                  int width, int height, int stride, int format,
                                  const mi355_hsv_settings *settings);

/* ---------------------------------------------------------------- hsvdetector
 * Replaces HsvDetector::hsv_detect + transform_frame's 6x4 format match
 * (video/hsv/src/hsvdetector/imp.rs:100-160, :423-707). */
typedef struct mi355_hsvdetect_settings {
  float hue_ref, hue_var;               /* defaults 0.0, 10.0 */
  float saturation_ref, saturation_var; /* defaults 0.0, 0.15 */
  float value_ref, value_var;           /* defaults 0.0, 0.3  */
} mi355_hsvdetect_settings;

int mi355_hsvdetect_frame(mi355_ctx *ctx, const uint8_t *src, size_t src_len, int src_stride,
                          int src_format, uint8_t *dst, size_t dst_len, int dst_stride,
                          int dst_format, int width, const mi355_hsvdetect_settings *settings);
int mi355_hsvdetect_frames_device(mi355_ctx *ctx, const uint8_t *d_src, size_t src_pitch,
                                  int src_stride, int src_format, uint8_t *d_dst, size_t dst_pitch,
                                  int dst_stride, int dst_format, int n_frames, int width,
                                  int height, const mi355_hsvdetect_settings *settings);

/* ---------------------------------------------------------------- colorlut
 * mi355_colorlut_load replaces the `State { lut }` installed by ColorLut::start
 * (video/colorlut/src/colorlut/imp.rs:168-194): the element keeps parsing the .cube text
 * (video/colorlut/src/parser.rs:110-281) and hands over the parsed CubeLut:
 *   is3d=1: `table` = Lut3D::as_flat(), size^3 cells of [r,g,b,1.0], index x + y*size + z*size^2
 *           (parser.rs:19-53, :253-256);  is3d=0: r[size], g[size], b[size] back to back
 *           (CubeLutKind::Lut1D, parser.rs:57-66).
 *   domain_scale / domain_offset: CubeLut fields (parser.rs:69-75, :264-274). */
int mi355_colorlut_load(mi355_ctx *ctx, int is3d, size_t size, const float *table,
                        const float domain_scale[3], const float domain_offset[3]);
/* ColorLut::stop (colorlut/imp.rs:196-199). */
int mi355_colorlut_unload(mi355_ctx *ctx);
/* Diagnostics for MI355_FLAG_LUT_VARIANT 0 (auto): which kernel kind serves packed RGBA8 launches right now
 * (*table_in_use: 0 interpolating kernel, 1 memoised table) and the last measured time of each kind in ms per
 * megapixel (0 = not measured yet); fused = 0 for mi355_colorlut_*, 1 for mi355_hsv_colorlut_*, 2 for
 * mi355_hsvfilter_* (MI355_FLAG_HSV_TABLE; *table_in_use then tells what the last call ran); fused = 10 / 11: the choice
 * INSIDE the table path of mi355_colorlut_* / mi355_hsv_colorlut_* - *table_in_use: 1 = the LDS-cached kernel
 * (colorlut_window.hip), 0 = the gather kernel; "compute" = the gather kernel's time, "table" = the LDS-cached kernel's. No
 * reference counterpart. */
int mi355_colorlut_kernel_choice(mi355_ctx *ctx, int fused, int *table_in_use, double *ms_per_mpx_compute, double *ms_per_mpx_table);
/* Number of memoised 64 MiB tables alive in this process (tables are shared by all contexts that ask for the same
 * function on the same device: same LUT and layout, same hsv settings). Diagnostic; no reference counterpart. */
int mi355_shared_table_count(void);
/* Name of the kernel that served the last mi355_colorlut_* / mi355_hsv_colorlut_* launch of this context ("" before the
 * first one). Diagnostic; no reference counterpart. */
const char *mi355_colorlut_last_kernel(mi355_ctx *ctx);
/* Brick-cache kernel diagnostics (synchronous): counters[0] = 256-pixel steps that found a brick missing in the wave's
 * LDS cache since the last reset, counters[1] = steps that still missed after the fill rounds (slow path);
 * *last_miss_fraction = miss fraction of the content watch's last snapshot, *level = the level it has settled on (0 brick
 * kernel with 32 sets, 1 with 64 sets, 2 three-pass kernel). reset != 0 clears the device counters. No reference counterpart. */
int mi355_colorlut_brick_stats(mi355_ctx *ctx, uint64_t counters[2], double *last_miss_fraction, int *level, int reset);
/* Diagnostics of the LDS-cached memoised-table kernel (csrc/colorlut_window.hip; synchronous): counters[0] = pixels it
 * has looked up on this context since the last reset, counters[1] = pixels whose table brick was not in the block's
 * LDS cache (served from the table in global memory), counters[2] = bricks installed. No reference counterpart. */
int mi355_colorlut_window_stats(mi355_ctx *ctx, uint64_t counters[3], int reset);
/* Host-logic self test of the content-watch policy (csrc/brickwatch.hpp) against a scripted stream, no GPU needed: call i
 * would show miss / slow step fractions miss0[i], slow0[i] on the 32-set brick kernel and miss1[i], slow1[i] on the 64-set
 * one; snapshots become readable `lag` calls late. level_out[i] = 0 / 1 / 2 as above. */
int mi355_selftest_brickwatch(int n_calls, const double *miss0, const double *slow0, const double *miss1, const double *slow1, int lag,
                              int *level_out);
/* Host-logic self test of the auto-choice policy against a scripted device (no GPU needed): call i has n_vec[i] 16-byte
 * pixel groups and, if measured, takes ms_compute[i] or ms_table[i] depending on the kind it ran; a measurement becomes
 * readable `lag` calls later. kind_out[i] = 0 interpolating / 1 table, measured_out[i] (optional) = launch was bracketed. */
int mi355_selftest_autopick(int n_calls, const uint64_t *n_vec, const double *ms_compute, const double *ms_table, int lag,
                            int *kind_out, int *measured_out);
/* Replaces transform_frame's body: transform_rgba / transform_rgba64::<LE>
 * (colorlut/imp.rs:203-223 -> :226-397). format in {RGBA, RGBA64_LE, RGBA64_BE}; src and dst are
 * plane 0 of two different frames with independent strides; rows = chunks(stride).take(height). */
int mi355_colorlut_frame(mi355_ctx *ctx, const uint8_t *src, int src_stride, uint8_t *dst,
                         int dst_stride, int width, int height, int format);
int mi355_colorlut_frames_device(mi355_ctx *ctx, const uint8_t *d_src, size_t src_pitch,
                                 int src_stride, uint8_t *d_dst, size_t dst_pitch, int dst_stride,
                                 int n_frames, int width, int height, int format);

/* Measurement plumbing, no reference counterpart: one round of n independent streams issued from ONE native loop - for
 * stream i, hsvfilter in place on the frame at d_src[i] and colorlut from it into d_dst[i], each through its own context
 * ctxs[i] (own HIP stream, LUT loaded) exactly as mi355_hsvfilter_frames_device + mi355_colorlut_frames_device would be
 * called by that stream's thread. Lets a benchmark written in an interpreter measure the device instead of its own call
 * overhead. Returns the first error; asynchronous like the calls it makes. */
int mi355_issue_streams_round(mi355_ctx *const *ctxs, int n_streams, uint8_t *const *d_src, uint8_t *const *d_dst,
                              int width, int height, int stride, int format, const mi355_hsv_settings *settings);

/* ---------------------------------------------------------------- many streams, few launches (csrc/group.hip)
 * GstBaseTransform hands an element ONE buffer per call (hsvfilter/imp.rs:323-376, colorlut/imp.rs:203-223): N streams through
 * `hsvfilter ! colorlut` are 2 N short launches per frame period, each paying its own ramp and tail. A group collects what the
 * streams submit and issues the frames of up to max_batch streams that agree in size, format, hsv settings and LUT as ONE
 * hsvfilter launch and ONE colorlut launch (the frame's base pointer is looked up per block); everything else goes through
 * its context's own path, in order. Results are those of mi355_hsvfilter_frames_device + mi355_colorlut_frames_device on that
 * frame, bit for bit. No reference counterpart (the reference has no device to batch for).
 *   create : max_batch 0 = default (8 frames per launch), at most 16.
 *   submit_chain : hsvfilter in place on the packed RGBA frame at d_src (stride bytes per row; MI355_FMT_RGBA, the one format both
 *           elements accept - anything else is MI355_ERR_INVALID_ARG), then colorlut from it into d_dst,
 *           for stream `ctx` (its LUT, loaded with mi355_colorlut_load; the frame starts after what ctx's HIP stream held at
 *           this call). Never blocks; launches only when max_batch frames are pending. One frame per stream and launch:
 *           frames of one stream run in submission order. *ticket identifies the frame.
 *   flush  : launch what is pending now.
 *   order_after : makes ctx's HIP stream wait (on the device, no host wait) for the frame `ticket`: what ctx enqueues on its
 *           own stream afterwards - a download - sees the result. Not implied by submit.
 *   wait   : host wait until the frame `ticket` is in d_dst; flushes first if the frame has not been launched - an element
 *           that works one frame deep (submit n, wait n-1) thereby batches with whatever the other streams submitted in
 *           between. The group's lock is not held while waiting.
 *   stats  : {frames, batched launch pairs, frames that went through their context's own path}.
 * Threads: every entry point may be called from any thread (one lock per group, not held during host waits). A flush launches
 * on behalf of EVERY stream with a pending frame, from whichever thread caused it, and reads those streams' contexts (LUT
 * table, flags): between submit and the return of wait for that ticket a context is used only through the group - no LUT
 * reload / unload, no flag change, no other entry point, no destroy. A launch that fails is reported to the call that caused
 * it and, once, to wait / order_after for each frame it carried (never as a silent success). */
typedef struct mi355_group mi355_group;
mi355_group *mi355_group_create(int device, int max_batch, int *status);
void mi355_group_destroy(mi355_group *group);
const char *mi355_group_last_error(mi355_group *group);
int mi355_group_submit_chain(mi355_group *group, mi355_ctx *ctx, uint8_t *d_src, uint8_t *d_dst, int width, int height,
                             int stride, int format, const mi355_hsv_settings *settings, uint64_t *ticket);
/* The same for the FUSED pair (mi355_hsv_colorlut_frames_device's semantics: d_src is read and left untouched, d_src == d_dst
 * allowed): frames that agree in size, hsv settings and LUT share ONE launch through the composed table of those settings
 * (built - or found with another stream of the process - once a stream has submitted with them eight times in a row: a hue
 * shift animated frame by frame never builds one). 3D LUTs; everything else goes through its context's own fused path. Bit-identical to mi355_hsv_colorlut_frames_device on that frame. */
int mi355_group_submit_fused(mi355_group *group, mi355_ctx *ctx, uint8_t *d_src, uint8_t *d_dst, int width, int height,
                             int stride, int format, const mi355_hsv_settings *settings, uint64_t *ticket);
int mi355_group_flush(mi355_group *group);
int mi355_group_order_after(mi355_group *group, mi355_ctx *ctx, uint64_t ticket);
int mi355_group_wait(mi355_group *group, uint64_t ticket);
int mi355_group_wait_all(mi355_group *group);
int mi355_group_stats(mi355_group *group, uint64_t stats[3]);
/* Measurement plumbing like mi355_issue_streams_round: one frame of each of n streams submitted from one native loop, then
 * flushed - what n streaming threads that submit within one frame period amount to. */
int mi355_group_submit_round(mi355_group *group, mi355_ctx *const *ctxs, int n_streams, uint8_t *const *d_src,
                             uint8_t *const *d_dst, int width, int height, int stride, int format,
                             const mi355_hsv_settings *settings);
int mi355_group_submit_round_fused(mi355_group *group, mi355_ctx *const *ctxs, int n_streams, uint8_t *const *d_src,
                                   uint8_t *const *d_dst, int width, int height, int stride, int format,
                                   const mi355_hsv_settings *settings);

/* videocompare across independent element instances. The reference's aggregate() hashes the reference pad's frame, then hashes and
 * compares every other pad's frame (video/videofx/src/videocompare/imp.rs:316-350 -> hashed_image.rs:24-79) - per element, one
 * synchronous engine call per frame. N two-pad elements in one process are N such sequences on N streams, which on one device
 * run SLOWER side by side than one after the other (32 4K Dssim streams over 8 contexts: 1.8 k comparisons/s; one context in
 * sequence: 2.2 k). submit_compare queues one (reference frame, frame) pair of stream `ctx`; pairs of one geometry, format and
 * algorithm run as ONE launch sequence on the group's compare stream - every pair's reference image is created right before the
 * kernels that read it (one image pool for the whole batch), pairs with the same d_ref (videocompare with several pads) share
 * it, the reductions and the copy of the scores happen once per batch - and wait_compare returns what
 * mi355_dssim_create_image + mi355_dssim_compare_frames (algo MI355_HASH_DSSIM: *distance = the dssim value) or
 * mi355_videocompare_hash_frame x 2 + mi355_videocompare_distance (MI355_HASH_BLOCKHASH: hashes[0] the reference's, hashes[1] the
 * frame's) give for that pair, bit for bit. Other algorithms: MI355_ERR_UNSUPPORTED (they go through their context).
 *   set_rendezvous : an element needs its result before it can return from aggregate(), so every stream submits and waits at
 *           once. With expected_streams > 0 a launch goes out as soon as that many pairs are pending, and a waiter whose pair
 *           has not been launched lingers up to linger_us for the others before it launches what is there (0 / 0 = a wait
 *           launches at once, the behaviour of mi355_group_wait).
 *   Frames are read until wait_compare for their ticket has returned. A result is collected once. Threads as for the group.
 *   set_compare_lanes : the pairs of a launch sequence are dealt out to this many HIP streams (default 8, 1..16; before the first
 *           submit_compare): the Dssim kernels are VALU-bound and ~80 % busy when one runs alone - a neighbour stream's launches
 *           fill its tails (32 4K pairs: 1.48 k comparisons/s on one stream, 1.84 k+ over eight).
 */
int mi355_group_set_rendezvous(mi355_group *group, int expected_streams, unsigned linger_us);
int mi355_group_set_compare_lanes(mi355_group *group, int lanes);
int mi355_group_submit_compare(mi355_group *group, mi355_ctx *ctx, const uint8_t *d_ref, const uint8_t *d_frame, int stride, int width, int height,
                               int format, int algo, uint64_t *ticket);
int mi355_group_wait_compare(mi355_group *group, uint64_t ticket, double *distance, uint64_t hashes[2]);
/* {pairs launched, launch sequences, pairs in the largest one} */
int mi355_group_compare_stats(mi355_group *group, uint64_t stats[3]);

/* ---------------------------------------------------------------- many AUDIO element instances, few launches (csrc/agroup.hip)
 * rsaudioecho (audio/audiofx/src/audioecho/imp.rs:205-227), ebur128level (audio/audiofx/src/ebur128level/imp.rs:682-745) and
 * audioloudnorm (audio/audiofx/src/audioloudnorm/imp.rs:1545-1586) are one instance per stream and one buffer per call; the batch
 * entry points above need ONE caller that owns all streams. An agroup is that caller for a process full of independent
 * instances of one kind and configuration: `n_members` of them, each submitting its buffer of the interval from its own
 * streaming thread and waiting for its ticket; the batch runs - one launch set for all - when every attached member has
 * submitted (on the thread that completes the set). Per-member results are those of a single-instance context fed the same
 * buffers, bit for bit.
 *   create_echo     : members are fully independent (own ring of ring_len f64 and position; per submit its own buffer length,
 *                     sample type, delay, intensity, feedback). A waiter lingers linger_us for the missing members, then launches
 *                     whoever is there (linger 0 = at once).
 *   create_ebur128  : members are independent meters of one configuration: per submit its own buffer length, each with its own
 *                     100 ms phase, and ebur128_reset (the element's `reset` action, ebur128level/imp.rs:124-139) for one member
 *                     alone; one sample format per launch set. Linger as for create_echo: a member that is late, paused or
 *                     detached simply does not advance.
 *   create_loudnorm : members are independent too: each stands at its own frame type (its first 3 s frame, 100 ms frames, the final
 *                     rest) and hands over whole frames of mi355_agroup_loudnorm_frame_size(group, member), or the shorter rest with
 *                     final_frame = 1 - what drain_full_frames / drain hand to State::process. A launch set runs one launch
 *                     sequence per class of members that stand at the same frame type and size (streams that started together);
 *                     linger as for create_echo. (timeout_ms of set_linger is accepted and ignored; MI355_ERR_TIMEOUT is not
 *                     returned any more: nobody waits for a member that does not come.)
 *   submit_*        : device_data = 0: host buffers (copied through one pinned slab: one upload and one download per interval
 *                     for all members); 1: device pointers. Buffers are borrowed until wait(ticket) returns.
 *   wait            : *out_frames (optional) = frames produced (audioloudnorm), samples processed (echo), frames metered.
 *   ebur128_loudness / _peak : the member's meter readings (what: 0 momentary, 1 short-term, 2 global, 3 relative threshold,
 *                     4 loudness range), computed once per interval for all members.
 *   stats           : {buffers, launch sets, buffers in the largest set}.
 * Threads: every entry point from any thread; one lock per group, held while a launch set runs. */
typedef struct mi355_agroup mi355_agroup;
mi355_agroup *mi355_agroup_create_echo(int device, int n_members, size_t ring_len, int *status);
mi355_agroup *mi355_agroup_create_ebur128(int device, int n_members, unsigned channels, unsigned rate, unsigned mode, const int *channel_class,
                                          int *status);
mi355_agroup *mi355_agroup_create_loudnorm(int device, int n_members, unsigned channels, double loudness_target, double loudness_range_target,
                                           double max_true_peak, double offset, int *status);
void mi355_agroup_destroy(mi355_agroup *group);
const char *mi355_agroup_last_error(mi355_agroup *group);
int mi355_agroup_set_linger(mi355_agroup *group, unsigned linger_us, unsigned timeout_ms);
int mi355_agroup_detach(mi355_agroup *group, int member);
int mi355_agroup_submit_echo(mi355_agroup *group, int member, void *data, size_t n, int is_f64, size_t delay_samples, double intensity,
                             double feedback, int device_data, uint64_t *ticket);
int mi355_agroup_submit_ebur128(mi355_agroup *group, int member, const void *data, size_t frames, int sample_format, int device_data,
                                uint64_t *ticket);
int mi355_agroup_ebur128_reset(mi355_agroup *group, int member);
int mi355_agroup_submit_loudnorm(mi355_agroup *group, int member, const double *data, size_t frames, double *out, size_t out_capacity_frames,
                                 int final_frame, int device_data, uint64_t *ticket);
size_t mi355_agroup_loudnorm_frame_size(mi355_agroup *group, int member);
/* audioloudnorm's sink_chain / drain for a member with the adapter on this side (what mi355_loudnorm_push / _drain are for a single
 * context; audioloudnorm/imp.rs:1545-1586, :226-310): push appends the buffer and hands every whole frame over together with
 * the other members (it blocks like wait); drain hands over the rest as the final frame. Host buffers. */
int mi355_agroup_loudnorm_push(mi355_agroup *group, int member, const double *data, size_t frames, double *out, size_t out_capacity_frames,
                               size_t *out_frames);
int mi355_agroup_loudnorm_drain(mi355_agroup *group, int member, double *out, size_t out_capacity_frames, size_t *out_frames, int *eos);
int mi355_agroup_wait(mi355_agroup *group, uint64_t ticket, size_t *out_frames);
int mi355_agroup_ebur128_loudness(mi355_agroup *group, int member, int what, double *out);
int mi355_agroup_ebur128_peak(mi355_agroup *group, int member, int true_peak, unsigned channel, double *out);
int mi355_agroup_echo_get_state(mi355_agroup *group, int member, double *ring_out, size_t ring_len, size_t *pos_out);
int mi355_agroup_stats(mi355_agroup *group, uint64_t stats[3]);
/* Process-wide groups. Elements of independent pipelines cannot hand a group to each other; what they share is the process
 * (gst/gstrsaudioecho.c, gstebur128level.c, gstaudioloudnorm.c, gstvideocompare.c with MI355_GROUP_MEMBERS=n in the environment).
 *   mi355_agroup_shared_* : THE group of this configuration (kind, device, member count, parameters), created at first use, and
 *           the next free member index in *member; a group whose members have all been handed out is not offered again.
 *   mi355_agroup_release  : detach; the last member out destroys the group.
 *   mi355_group_shared / _release : THE dispatcher of `device` (mi355_group_create(device, 0)), reference-counted. */
mi355_agroup *mi355_agroup_shared_echo(int device, int n_members, size_t ring_len, int *member, int *status);
mi355_agroup *mi355_agroup_shared_ebur128(int device, int n_members, unsigned channels, unsigned rate, unsigned mode, const int *channel_class,
                                          int *member, int *status);
mi355_agroup *mi355_agroup_shared_loudnorm(int device, int n_members, unsigned channels, double loudness_target, double loudness_range_target,
                                           double max_true_peak, double offset, int *member, int *status);
void mi355_agroup_release(mi355_agroup *group, int member);
mi355_group *mi355_group_shared(int device, int *status);
void mi355_group_release(mi355_group *group);

/* ---------------------------------------------------------------- hsvfilter ! colorlut, fused
 * The chain `hsvfilter ! colorlut` on RGBA (the only format both elements accept, hsvfilter/imp.rs:252-266 and
 * colorlut/imp.rs:125-137) as ONE pass: every pixel goes through hsv_filter's body (hsvfilter/imp.rs:96-118) and
 * then transform_rgba's body (colorlut/imp.rs:226-294) in registers. Output is bit-identical to
 * mi355_hsvfilter_frames_device followed by mi355_colorlut_frames_device; src is left untouched (src == dst
 * allowed). Needs a loaded LUT; 1D LUTs and non-contiguous frames run as the two element kernels. */
int mi355_hsv_colorlut_frames_device(mi355_ctx *ctx, const uint8_t *d_src, size_t src_pitch,
                                     int src_stride, uint8_t *d_dst, size_t dst_pitch, int dst_stride,
                                     int n_frames, int width, int height,
                                     const mi355_hsv_settings *settings);

/* ---------------------------------------------------------------- rsaudioecho
 * mi355_echo_setup replaces AudioEcho::setup's RingBuffer::new(buffer_size)
 * (audio/audiofx/src/audioecho/imp.rs:248-259, ring_buffer.rs:15-24): ring of `ring_len` f64
 * zeros, position 0. mi355_echo_reset = stop() dropping the state (imp.rs:229-234). */
int mi355_echo_setup(mi355_ctx *ctx, size_t ring_len);
int mi355_echo_reset(mi355_ctx *ctx);
/* Replaces AudioEcho::process::<f32|f64> (imp.rs:69-85) + RingBufferIter (ring_buffer.rs:37-82).
 * `delay_samples` = delay_frames of imp.rs:74-77 (interleaved samples, after the max-delay
 * clamp of imp.rs:207). In place on `n` interleaved samples in host memory. */
int mi355_echo_process_f32(mi355_ctx *ctx, float *data, size_t n, size_t delay_samples,
                           double intensity, double feedback);
int mi355_echo_process_f64(mi355_ctx *ctx, double *data, size_t n, size_t delay_samples,
                           double intensity, double feedback);
int mi355_echo_process_device(mi355_ctx *ctx, void *d_data, size_t n, int is_f64,
                              size_t delay_samples, double intensity, double feedback);
/* Test/diagnostic access to the element state (ring contents + write position). */
int mi355_echo_get_state(mi355_ctx *ctx, double *ring_out, size_t ring_len, size_t *pos_out);
/* Batches: `n_streams` independent AudioEcho instances (own ring, own delay / intensity / feedback - the per-call arrays
 * have n_streams entries) advanced together by `n` interleaved samples per call, three launches for the whole batch
 * instead of three per stream (one rsaudioecho buffer is a few thousand samples: a single stream is launch-bound).
 * Stream s processes d_data + s * stream_stride elements (f32 or f64), in place, device memory. The single-stream
 * entry points above are the n_streams == 1 case of the same kernels. */
int mi355_echo_setup_batch(mi355_ctx *ctx, int n_streams, size_t ring_len);
int mi355_echo_process_batch_device(mi355_ctx *ctx, void *d_data, size_t stream_stride, size_t n, int is_f64,
                                    const size_t *delay_samples, const double *intensity, const double *feedback);
int mi355_echo_get_state_batch(mi355_ctx *ctx, int stream, double *ring_out, size_t ring_len, size_t *pos_out);

/* ---------------------------------------------------------------- ebur128 (loudness meter)
 * Replaces the `ebur128::EbuR128` object (third-party crate ebur128 0.1.10, Cargo.lock:3685-3686) that
 * `ebur128level` feeds and queries: EbuR128::new(channels, rate, mode) + set_channel_map
 * (audio/audiofx/src/ebur128level/imp.rs:518-595), reset() (:329), add_frames_{i16,i32,f32,f64}[_planar]
 * (:690-739), loudness_momentary / loudness_shortterm / loudness_global / relative_threshold /
 * loudness_range / sample_peak / true_peak (:378-452). The crate's source is not vendored: this is the
 * published BS.1770-4 / EBU R128 algorithm in libebur128's formulation (histogram mode, as the element
 * always requests, imp.rs:55-56); parity with the crate is unpinned.
 *   mode: bit 0 momentary, 1 short-term, 2 global (integrated + relative threshold), 3 loudness range,
 *         4 sample peak, 5 true peak  (== GstEbuR128LevelMode, imp.rs:34-51).
 *   channel_class[c]: 0 unused (LFE / unknown position), 1 weight 1.0 (L, R, C, ...), 2 weight 1.41
 *         (surround: +-110/+-90/+-60 degrees), 3 dual mono (weight 2.0); NULL = libebur128's default map.
 *   sample_format: 0 S16, 1 S32, 2 F32, 3 F64 (native endian), interleaved or planar.
 * Loudness values are LUFS / LU as f64; "no signal above the gate" is -inf (global) / -70 (threshold). */
typedef enum mi355_ebur128_mode {
  MI355_EBUR128_MOMENTARY = 1, MI355_EBUR128_SHORT_TERM = 2, MI355_EBUR128_GLOBAL = 4,
  MI355_EBUR128_LOUDNESS_RANGE = 8, MI355_EBUR128_SAMPLE_PEAK = 16, MI355_EBUR128_TRUE_PEAK = 32
} mi355_ebur128_mode;
int mi355_ebur128_setup(mi355_ctx *ctx, unsigned channels, unsigned rate, unsigned mode, const int *channel_class);
/* Batch form: `n_streams` independent meters of ONE configuration fed in lock step, every kernel launched once for all of
 * them (grid y = stream). One stream's K-weighting recurrence is serial; hundreds of streams are what fills the GPU
 * (ebur128level is one instance per stream in the reference: ebur128level/imp.rs:682-745 runs per element; a transcoding
 * farm runs hundreds of them). `data` of add_frames_batch: n_streams buffers of frames x channels interleaved samples,
 * back to back (host pointer; _device: device pointer, asynchronous up to the read-back of the gating energies).
 * loudness_batch: what = 0 momentary, 1 short-term, 2 global, 3 relative threshold, 4 loudness range -> out[n_streams];
 * peak_batch -> out[n_streams][channels]. mi355_ebur128_reset / _teardown apply to the batch as a whole. Per-stream results
 * are identical to n_streams separate single-stream meters (tests/test_gpu_ebur128.py). Meters that are fed independently - other
 * buffer sizes, late or paused members, a reset of one of them - are members of an audio group (mi355_agroup_create_ebur128 above):
 * the engine underneath keeps a 100 ms phase per stream. */
int mi355_ebur128_setup_batch(mi355_ctx *ctx, unsigned n_streams, unsigned channels, unsigned rate, unsigned mode, const int *channel_class);
int mi355_ebur128_add_frames_batch(mi355_ctx *ctx, const void *data, size_t frames, int sample_format);
int mi355_ebur128_add_frames_batch_device(mi355_ctx *ctx, const void *d_data, size_t frames, int sample_format);
int mi355_ebur128_loudness_batch(mi355_ctx *ctx, int what, double *out);
int mi355_ebur128_peak_batch(mi355_ctx *ctx, int true_peak, double *out);
int mi355_ebur128_reset(mi355_ctx *ctx);
int mi355_ebur128_teardown(mi355_ctx *ctx);
int mi355_ebur128_add_frames(mi355_ctx *ctx, const void *data, size_t frames, int sample_format);
int mi355_ebur128_add_frames_planar(mi355_ctx *ctx, const void *const *planes, size_t frames, int sample_format);
int mi355_ebur128_loudness_momentary(mi355_ctx *ctx, double *out);
int mi355_ebur128_loudness_shortterm(mi355_ctx *ctx, double *out);
int mi355_ebur128_loudness_global(mi355_ctx *ctx, double *out);
int mi355_ebur128_relative_threshold(mi355_ctx *ctx, double *out);
int mi355_ebur128_loudness_range(mi355_ctx *ctx, double *out);
int mi355_ebur128_sample_peak(mi355_ctx *ctx, unsigned channel, double *out);
int mi355_ebur128_true_peak(mi355_ctx *ctx, unsigned channel, double *out);

/* ---------------------------------------------------------------- audioloudnorm
 * Replaces audioloudnorm's State (audio/audiofx/src/audioloudnorm/imp.rs:83-127) and its methods: State::new (:130-205),
 * drain_full_frames / drain (:226-310), process and the frame handlers (:312-828), the true-peak limiter (:845-1430),
 * detect_peak (:1438-1524), gaussian_filter (:1526-1541). Audio is interleaved f64 at 192 kHz, the only caps the
 * element accepts (:1848-1851); the element keeps its pads, timestamps and events.
 *   setup : State::new(settings, info) with the four properties (loudness-target [LUFS], loudness-range-target [LU],
 *           max-true-peak [dbTP], offset [LU]; imp.rs:37-40).
 *   push  : sink_chain -> adapter.push + drain_full_frames: every complete frame (3 s first, then 100 ms) is processed;
 *           `out` receives *out_frames frames (0 while the first 3 s are still being collected).
 *   drain : drain() at EOS / flush: the final frame (up to 3 s of latency comes out) or, with less than 3 s in total,
 *           the linear-gain path; *eos = 1 when there was nothing at all to drain (FlowError::Eos, :289-293).
 *           out_capacity_frames >= 30*19200 + pending frames covers every case. */
int mi355_loudnorm_setup(mi355_ctx *ctx, unsigned channels, double loudness_target,
                         double loudness_range_target, double max_true_peak, double offset);
int mi355_loudnorm_push(mi355_ctx *ctx, const double *data, size_t frames, double *out,
                        size_t out_capacity_frames, size_t *out_frames);
int mi355_loudnorm_drain(mi355_ctx *ctx, double *out, size_t out_capacity_frames, size_t *out_frames,
                         int *eos);
int mi355_loudnorm_teardown(mi355_ctx *ctx);
/* Batches (round 3): n_streams elements of one configuration in LOCK STEP - State::process (imp.rs:800-828) for all of them
 * per call, with the true-peak limiter's state machine (:845-1430) running on the device, one block per stream. The element
 * keeps its adapter: `process_batch` is called the way drain_full_frames (:226-268) calls State::process, with exactly
 * mi355_loudnorm_batch_frame_size() frames per stream (576,000 for the first call, then 19,200) - or, with final_frame = 1,
 * with the shorter rest at drain() (:270-310; 0 frames allowed). Stream s reads data + s * stream_stride and writes
 * out + s * out_stream_stride (interleaved f64, elements); every stream produces *out_frames frames. device_data = 1: both
 * are device pointers (no sample crosses PCIe); the call still waits for the device wherever State::process reads a meter
 * (the short-term / global loudness of every frame decides the gain on the host, as in the reference). The output capacity
 * is checked before anything changes: after "output buffer too small" the same call can be repeated with a larger buffer.
 * Per-stream samples are identical to n separate mi355_loudnorm_* contexts. mi355_loudnorm_teardown releases a batch as well. */
int mi355_loudnorm_setup_batch(mi355_ctx *ctx, unsigned n_streams, unsigned channels, double loudness_target,
                               double loudness_range_target, double max_true_peak, double offset);
size_t mi355_loudnorm_batch_frame_size(mi355_ctx *ctx);
int mi355_loudnorm_process_batch(mi355_ctx *ctx, const double *data, size_t stream_stride, size_t frames,
                                 double *out, size_t out_stream_stride, size_t out_capacity_frames,
                                 size_t *out_frames, int final_frame, int device_data);

/* ---------------------------------------------------------------- videocompare
 * Replaces HasherEngine::hash_image / compare (video/videofx/src/videocompare/hashed_image.rs:24-79) for
 * HashAlgorithm::Blockhash, the element default (videocompare/imp.rs:31): image_hasher 3.1.1 blockhash, 8x8 bits, on
 * the packed RGB / RGBA frame (rows addressed by `stride`, which covers tightly_packed_framebuffer :110-130).
 * `algo` takes GstVideoCompareHashAlgorithm values (videocompare/mod.rs:60-100). Blockhash on frame sizes that are not
 * divisible by 8 is MI355_ERR_UNSUPPORTED unless MI355_FLAG_BLOCKHASH_ANY_SIZE is set (the crate's floating-point path,
 * hashed_image.rs:37-44 -> image_hasher blockhash_slow); its hash is the 64 block bits, bit i = block i row-major. Mean / Gradient /
 * VertGradient / DoubleGradient (grayscale + Lanczos3 resize of the `image` crate + bit rule) give 64/64/64/40 bits in the
 * crate's iteration order. Dssim has no hash: see mi355_dssim_* below. */
typedef enum mi355_hash_algo {
  MI355_HASH_MEAN = 0, MI355_HASH_GRADIENT = 1, MI355_HASH_VERTGRADIENT = 2, MI355_HASH_DOUBLEGRADIENT = 3,
  MI355_HASH_BLOCKHASH = 4, MI355_HASH_DSSIM = 5
} mi355_hash_algo;
int mi355_videocompare_hash_frame(mi355_ctx *ctx, const uint8_t *data, int stride, int width, int height,
                                  int format, int algo, uint64_t *hash);
/* `n_frames` device-resident frames (frame f at d_frames + f*frame_pitch); `hashes` is a host array. Synchronous. */
int mi355_videocompare_hash_frames_device(mi355_ctx *ctx, const uint8_t *d_frames, size_t frame_pitch,
                                          int stride, int n_frames, int width, int height, int format,
                                          int algo, uint64_t *hashes);
/* ImageHash::dist as f64 (hashed_image.rs:64): Hamming distance. Negative for an unsupported `algo`. */
double mi355_videocompare_distance(int algo, uint64_t reference_hash, uint64_t frame_hash);
/* HashAlgorithm::Dssim (cargo feature `dssim`; hashed_image.rs:41-53,66-70): crate dssim-core 3.4.0.
 *   mi355_dssim_create_image  = Dssim::create_image_rgb / create_image_rgba on the packed frame (host pointer; the
 *                               _device variant takes a device-resident frame). The DssimImage<f32> stays on the device.
 *   mi355_dssim_compare       = Dssim::compare(original, modified).0 as f64 (0.0 for identical images).
 * format: MI355_FMT_RGB or MI355_FMT_RGBA. Translucent RGBA pixels (alpha < 255) are composed over dssim's coloured,
 * position-dependent background pattern (hashed_image.rs:54-55 -> create_image_rgba; written from memory of the crate:
 * parity unpinned like the rest of the engine; round 2 refused such frames); MI355_FLAG_DSSIM_TRANSLUCENT = 1 composes
 * them over black instead. create / free do not wait for the kernels (the host variant of create returns once `data` has
 * been uploaded and may be reused); mi355_dssim_compare does (it returns the value). An
 * image is created, compared and freed through ONE context. */
typedef struct mi355_dssim_image mi355_dssim_image;
int mi355_dssim_create_image(mi355_ctx *ctx, const uint8_t *data, int stride, int width, int height,
                             int format, mi355_dssim_image **out);
int mi355_dssim_create_image_device(mi355_ctx *ctx, const uint8_t *d_frame, int stride, int width,
                                    int height, int format, mi355_dssim_image **out);
void mi355_dssim_free_image(mi355_ctx *ctx, mi355_dssim_image *image);
int mi355_dssim_compare(mi355_ctx *ctx, const mi355_dssim_image *original,
                        const mi355_dssim_image *modified, double *dssim);
/* videocompare's loop over the non-reference pads of one aggregate (videocompare/imp.rs:316-345: each pad's frame goes
 * through HashedImage::new and HashedImage::compare against the reference pad's image, and its hash is dropped):
 * `n_frames` (<= 64) frames of the original's size are hashed AND compared against `original` in one pass per scale - their
 * DssimImages never reach memory - and dssim[i] receives what mi355_dssim_create_image + mi355_dssim_compare(original, .)
 * return for frames[i], bit for bit. One synchronisation per call. `frames` is a host array of host pointers (of device
 * pointers for the _device variant). */
int mi355_dssim_compare_frames(mi355_ctx *ctx, const mi355_dssim_image *original, const uint8_t *const *frames,
                               int n_frames, int stride, int width, int height, int format, double *dssim);
int mi355_dssim_compare_frames_device(mi355_ctx *ctx, const mi355_dssim_image *original,
                                      const uint8_t *const *d_frames, int n_frames, int stride, int width,
                                      int height, int format, double *dssim);
/* Device self test of the LAB conversion's cube root: for every f32 whose bit pattern lies in [lo_bits, hi_bits], the
 * Halley iteration with the range-restricted division the kernels use against the same iteration with the compiler's IEEE
 * division; *mismatches = how many differ in any bit (0 over (216/24389, 2], the whole domain of the conversion). */
int mi355_selftest_dssim_cbrt(mi355_ctx *ctx, uint32_t lo_bits, uint32_t hi_bits, uint64_t *mismatches);
/* Diagnostics: one f32 plane of the image (kind 0 = LAB plane, 1 = mu, 2 = img_sq_blur) copied to `out` (may be NULL
 * to query the scale's size only). */
int mi355_dssim_image_plane(mi355_ctx *ctx, const mi355_dssim_image *image, int scale, int channel, int kind,
                            float *out, int *width, int *height);

/* ---------------------------------------------------------------- hrtfrender
 * Replaces the per-block body of HrtfRender::process (audio/hrtf/src/hrtf/imp.rs:164-278) including the calls
 * into the `hrtf` crate (HrirSphere::new, HrtfProcessor::new, process_samples).
 *   mi355_hrtf_load_sphere : Settings::sphere(rate) -> HrirSphere::new(bytes, rate) (imp.rs:84-94). `bytes` is the
 *       crate's file format ("HRIR", u32 rate, u32 len, u32 n_vertices, u32 n_indices, indices, vertices). A sphere
 *       whose rate differs from `device_rate` is converted at load time, every HRIR by band-limited (windowed-sinc)
 *       interpolation to round(len * device_rate / file_rate) taps - the crate does this with the `rubato` resampler, whose
 *       sources are not in the reference tree: same method, parity unpinned (csrc/hrtf_kernels.hip: resample_hrir).
 *   mi355_hrtf_setup       : the ChannelProcessor vector of set_caps (imp.rs:662-680): one HrtfProcessor per input
 *       channel, zeroed tails, no previous vector/gain.
 *   mi355_hrtf_reset       : State::reset_processors (imp.rs:124-129).
 *   mi355_hrtf_process_block : one block of block_length*interpolation_steps frames (imp.rs:196-271): `in` is
 *       interleaved f32 [frames][channels], `out` interleaved stereo f32 [frames][2], overwritten.
 *       positions[c] = Settings::position(c) (already `.to_right_handed()`, imp.rs:64-73), gains[c] =
 *       Settings::distance_gain(c). The adapter / drain logic around it (imp.rs:281-420) stays in the element. */
int mi355_hrtf_load_sphere(mi355_ctx *ctx, const void *bytes, size_t len, uint32_t device_rate);
int mi355_hrtf_setup(mi355_ctx *ctx, int channels, int block_length, int interpolation_steps);
int mi355_hrtf_reset(mi355_ctx *ctx);
int mi355_hrtf_teardown(mi355_ctx *ctx);
int mi355_hrtf_process_block(mi355_ctx *ctx, const float *in, float *out, const float *positions_xyz,
                             const float *distance_gains);
/* Same with device-resident input/output (asynchronous on the context stream). */
int mi355_hrtf_process_block_device(mi355_ctx *ctx, const float *d_in, float *d_out,
                                    const float *positions_xyz, const float *distance_gains);
/* Sphere geometry as loaded: HRIR length, vertices, faces. */
int mi355_hrtf_sphere_info(mi355_ctx *ctx, uint32_t *hrir_len, uint32_t *n_vertices, uint32_t *n_faces);
/* Diagnostics: mesh face (or -1) and barycentric weights chosen per [channel][step] in the last block. */
int mi355_hrtf_last_lookup(mi355_ctx *ctx, int *faces, float *uvw);

/* ---------------------------------------------------------------- sofalizer
 * Replaces the per-block loop of Sofalizer::process (audio/hrtf/src/sofa/imp.rs:234-300): de-interleave each channel,
 * sofar's Renderer::process_block (uniformly partitioned convolution with `partition-length`, built in set_caps :793-798)
 * and the channel-ordered mix `out[2i] += l * gain, out[2i+1] += r * gain` (:282-297). The element keeps what is not
 * per-sample work: opening the SOFA file and looking up / interpolating the HRIR pair of a position (Sofar::filter in
 * State::update_filters, :129-160) - it hands each pair over with mi355_sofa_set_filter when a source has moved by more
 * than `update-threshold`, exactly where it calls Renderer::set_filter today - plus the adapter and drain logic.
 *   setup       block_length must be a multiple of partition_length (else INVALID_ARG with the reference's message,
 *               :775-781); partition_length a power of two in 8..2048 (UNSUPPORTED otherwise)
 *   set_filter  filter_len taps per ear + whole-sample onset delays (>= 0); effective from the next block
 *   set_drop    ChannelProcessor::Drop for LFE1 / LFE2 (:808-821)
 *   reset       flush-stop (:840-848): input history cleared
 *   process_block  in: block_length x channels interleaved f32; out: block_length x 2 interleaved f32
 * The crate's arithmetic is not in the reference tree: parity is that of a streaming linear convolution, within 2e-6 of
 * full scale against the time-domain oracle. */
int mi355_sofa_setup(mi355_ctx *ctx, int channels, int filter_len, int partition_length, int block_length);
int mi355_sofa_set_filter(mi355_ctx *ctx, int channel, const float *left, const float *right, int delay_left, int delay_right);
int mi355_sofa_set_drop(mi355_ctx *ctx, int channel, int drop);
int mi355_sofa_reset(mi355_ctx *ctx);
int mi355_sofa_teardown(mi355_ctx *ctx);
int mi355_sofa_process_block(mi355_ctx *ctx, const float *in, float *out, const float *distance_gains);
int mi355_sofa_process_block_device(mi355_ctx *ctx, const float *d_in, float *d_out, const float *distance_gains);

/* ---------------------------------------------------------------- pinned memory + asynchronous host-buffer pipeline
 * For the GStreamer shim (precedent video/colorlut/src/d3d12colorlut/imp.rs:385-492 and
 * audio/audiofx/src/audiornnoise/imp.rs:323-348): mi355_host_alloc backs a GstAllocator / buffer pool offered in
 * propose_allocation so that mapped GstBuffer payloads are page-locked; mi355_pipe_* overlaps the upload of frame
 * n+1 and the download of frame n-1 with the kernels of frame n on side streams. The submit functions take the
 * arguments of the synchronous entry point they mirror (mi355_hsvfilter_frame_ip, mi355_colorlut_frame,
 * mi355_hsv_colorlut_frames_device on one frame), return at once with a ticket, and borrow the buffers until
 * mi355_pipe_wait(ticket) (or a later submit that reclaims the slot) returns. Results are identical. */
void *mi355_host_alloc(mi355_ctx *ctx, size_t bytes);
int mi355_host_free(mi355_ctx *ctx, void *ptr);
typedef struct mi355_pipe mi355_pipe;
mi355_pipe *mi355_pipe_create(mi355_ctx *ctx, int depth, size_t max_frame_bytes);
void mi355_pipe_destroy(mi355_pipe *pipe);
int mi355_pipe_submit_hsvfilter(mi355_pipe *pipe, uint8_t *data, size_t data_len, int width, int stride,
                                int format, const mi355_hsv_settings *settings, uint64_t *ticket);
int mi355_pipe_submit_colorlut(mi355_pipe *pipe, const uint8_t *src, int src_stride, uint8_t *dst,
                               int dst_stride, int width, int height, int format, uint64_t *ticket);
int mi355_pipe_submit_hsv_colorlut(mi355_pipe *pipe, const uint8_t *src, int src_stride, uint8_t *dst,
                                   int dst_stride, int width, int height,
                                   const mi355_hsv_settings *settings, uint64_t *ticket);
int mi355_pipe_wait(mi355_pipe *pipe, uint64_t ticket);
int mi355_pipe_wait_all(mi355_pipe *pipe);
/* Many pipelines in one process: with a group (mi355_group_*) set, mi355_pipe_submit_hsv_colorlut hands
 * its frame to the group's dispatcher instead of launching (the frame then shares a launch pair with the other streams'
 * frames) and the download is enqueued behind that batch when the frame is waited for. Same bytes. NULL = back to own launches.
 * The group is not owned. */
struct mi355_group;
int mi355_pipe_set_group(mi355_pipe *pipe, struct mi355_group *group);

/* ---------------------------------------------------------------- measurement helpers
 * Used by bench.py: run `iters` back-to-back launches of one kernel on the context's stream
 * bracketed by hipEvents on THAT stream and return the average milliseconds per launch. */
int mi355_time_hsvfilter_device(mi355_ctx *ctx, uint8_t *d_data, int n_frames, size_t frame_pitch,
                                int width, int height, int stride, int format,
                                const mi355_hsv_settings *settings, int iters, float *ms_per_launch);
int mi355_time_hsv_colorlut_device(mi355_ctx *ctx, const uint8_t *d_src, size_t src_pitch, int src_stride,
                                   uint8_t *d_dst, size_t dst_pitch, int dst_stride, int n_frames, int width,
                                   int height, const mi355_hsv_settings *settings, int iters, float *ms_per_launch);
int mi355_time_colorlut_device(mi355_ctx *ctx, const uint8_t *d_src, size_t src_pitch, int src_stride,
                               uint8_t *d_dst, size_t dst_pitch, int dst_stride, int n_frames,
                               int width, int height, int format, int iters, float *ms_per_launch);

/* hsvfilter (in place on d_src[k]) then colorlut (d_src[k] -> d_dst[k]) for n_batches independent batches of n_frames packed frames
 * each, issued from ONE native call - what n_batches pairs of mi355_hsvfilter_frames_device + mi355_colorlut_frames_device do
 * (video/hsv/src/hsvfilter/imp.rs:323-376 then video/colorlut/src/colorlut/imp.rs:203-223 per buffer), same kernels, same bytes.
 * lanes: 1 = all on the context's stream; 2 is accepted and runs as 1 (round 5's side stream for odd batches measured 12 % slower -
 * two memory-bound launches side by side evict each other's intermediate from the Infinity Cache - and did not order one-off
 * table builds across the lanes; removed in round 6). */
int mi355_hsv_colorlut_chain_batches_device(mi355_ctx *ctx, uint8_t *const *d_src, uint8_t *const *d_dst, int n_batches, int n_frames,
                                            size_t frame_pitch, int stride, int width, int height, int format,
                                            const mi355_hsv_settings *settings, int lanes);

/* ---------------------------------------------------------------- device buffers: what a device GstMemory wraps
 * Precedent: video/colorlut/src/d3d12colorlut/imp.rs:385-492 (propose_allocation / decide_allocation offer a pool of GPU
 * memory; an element whose input memory is "ours" works on the GPU resource, anything else maps it). A mi355_buf is `size`
 * bytes of HBM with a lazily created pinned host shadow and three-state dirty tracking (in sync / host newer / device newer):
 *   mi355_buf_device_ptr(buf, ctx, flags)  the pointer the *_device entry points take. MI355_MAP_READ uploads the shadow first
 *       if it is newer (on ctx's stream); MI355_MAP_WRITE marks the device side newer; ctx's stream is ordered behind the last
 *       commit of another context (hipStreamWaitEvent, no host wait). NULL on error (mi355_ctx_last_error(ctx)).
 *   mi355_buf_commit(buf, ctx)             call after enqueuing the kernels that use the pointer: "ctx's stream now holds the
 *       last use of this buffer".
 *   mi355_buf_map_host / unmap_host        GstMemory mem_map / mem_unmap: the pinned shadow; downloaded first (and the calling
 *       thread waits) only if the device side is newer; a WRITE map makes the host side newer at unmap.
 *   mi355_buf_ref / unref                  GstMemory copies share the buffer; the last unref waits for the device and frees.
 *   mi355_buf_state                        0 in sync, 1 host newer, 2 device newer (diagnostics).
 * mi355_ctx_transfer_counts: host->device / device->host copies enqueued so far by this context's buffers and by its
 * mi355_h2d / mi355_d2h / mask uploads (a chain of elements on device buffers costs ONE upload and ONE download:
 * tests/test_gpu_buf.py). */
enum { MI355_MAP_READ = 1, MI355_MAP_WRITE = 2 };
typedef struct mi355_buf mi355_buf;
mi355_buf *mi355_buf_alloc(mi355_ctx *ctx, size_t size);
mi355_buf *mi355_buf_ref(mi355_buf *buf);
void mi355_buf_unref(mi355_buf *buf);
size_t mi355_buf_size(const mi355_buf *buf);
void *mi355_buf_device_ptr(mi355_buf *buf, mi355_ctx *ctx, int flags);
int mi355_buf_commit(mi355_buf *buf, mi355_ctx *ctx);
void *mi355_buf_map_host(mi355_buf *buf, int flags);
int mi355_buf_unmap_host(mi355_buf *buf);
int mi355_buf_state(mi355_buf *buf);
int mi355_ctx_transfer_counts(mi355_ctx *ctx, uint64_t *h2d, uint64_t *d2h);

/* ---------------------------------------------------------------- roundedcorners on device-resident frames
 * video/videofx/src/border/imp.rs: generate_alpha_mask (:57-180) renders ONE A8 plane per caps / radius change and
 * prepare_output_buffer (:482-559) appends that one shared memory to every buffer as plane 3 of A420 (stride[3] x
 * round_up_2(height) bytes, :469-470). The bytes are cairo's and are rendered on the host (mi355host_rounded_corners_mask);
 *   mi355_roundedcorners_set_mask       keeps that plane in HBM (mask == NULL: passthrough, the plane is dropped);
 *   mi355_roundedcorners_mask_device    the shared device plane - what a device GstMemory appended to every buffer wraps
 *                                       (no bytes move per buffer, as in the reference);
 *   mi355_roundedcorners_append_device  writes the plane behind the I420 planes of n_frames frames (frame f's plane 3 at
 *                                       d_frames + f * frame_pitch + alpha_offset) for consumers that want A420 contiguous: one launch. */
int mi355_roundedcorners_set_mask(mi355_ctx *ctx, const uint8_t *mask, int width, int height, int stride);
int mi355_roundedcorners_mask_device(mi355_ctx *ctx, const uint8_t **d_mask, size_t *size, int *stride);
int mi355_roundedcorners_append_device(mi355_ctx *ctx, uint8_t *d_frames, size_t frame_pitch, size_t alpha_offset, int n_frames);

#ifdef __cplusplus
}
#endif
#endif /* MI355FX_H */
