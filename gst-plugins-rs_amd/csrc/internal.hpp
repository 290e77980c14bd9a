// internal.hpp — shared declarations between the C-ABI layer (ctx.hip) and the kernel files.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>
#include "../../include/mi355fx.h"

#include "autopick.hpp"
#include "colorlut_brick.hpp"
#include "colorlut_window.hpp"

namespace mi355 {

// Device-side copy of a loaded CubeLut (video/colorlut/src/parser.rs:69-75).
// Choice between an interpolating ("compute") kernel and the memoised-table kernel for one entry point: the policy
// state (autopick.hpp) plus the two events that bracket a sampled launch (colorlut_kernels.hip: auto_launch).
struct AutoPick : AutoPolicy {
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

struct LutDevice {
  int is3d = 0;
  int size = 0;
  float scale[3] = {1, 1, 1};
  float offset[3] = {0, 0, 0};
  float *d_cells = nullptr;   // 3D: [size^3][4] as the reference stores it; 1D: r|g|b planes
  float *d_planar = nullptr;  // 3D only: three channel planes in the LDS image layout (see colorlut_kernels.hip)
  uint32_t *d_axis = nullptr; // 3D only: per-axis {offset, t} tables, 3 x 256 x 2 dwords
  size_t planar_plane_floats = 0;  // floats per padded plane
  int lds_Sy = 0, lds_Sz = 0;      // LDS row / plane strides (floats)
  size_t lds_bytes = 0;            // dynamic LDS per block
  bool lds_all_resident = false;   // all three planes staged once (small LUTs)
  bool lds_ok = false;        // LDS fast path legal for this LUT (fits, finite, bounded)
  // 2^24-entry memoised tables (RGBA8): see colorlut_table_kernel. [0] = colorlut alone, [1] = hsvfilter -> colorlut
  // under the hsv settings in table_hs.
  uint32_t *d_table[2] = {nullptr, nullptr};   // = table_ref[i]'s device pointer (shared between contexts, see SharedTable)
  std::shared_ptr<void> table_ref[2];          // keeps the shared table alive while this context uses it
  std::string table_key[2];                    // what the table in use was built from
  std::string digest;                          // identifies the loaded LUT (contents, size, kind, domain) in table keys
  int table_morton[2] = {-1, -1};  // layout of the table: 0 linear, 1 Morton, -1 not built
  mi355_hsv_settings table_hs{};   // settings d_table[1] was built for
  mi355_hsv_settings seen_hs{};    // settings of the previous fused call and for how many calls they have not changed
  unsigned seen_stable = 0;
  AutoPick pick[2];                // compute-kernel / table-kernel choice for the two entry points
  int last_sub[2] = {0, 0};        // which kernel read the table last: 0 gather, 1 LDS-cached (colorlut_window.hip) - by provenance of the input
  BrickLut brick;                  // brick form of a 3D LUT for the brick-cache interpolating kernel (colorlut_brick.hip)
  bool building_table = false;     // launch_*_compute is being run over the all-colours frame by table_ensure
  const char *last_kernel = "";    // name of the kernel that served the last mi355_colorlut_* / mi355_hsv_colorlut_* launch
  bool loaded = false;
};

// hsvfilter through a memoised table (colorlut_kernels.hip: launch_hsvfilter): the table of the element's per-pixel
// function under `hs` for colour-first 4-byte formats (bgr selects the byte order the table was built for).
struct HsvTable {
  uint32_t *d_table = nullptr;         // = table_ref's device pointer
  std::shared_ptr<void> table_ref;
  bool valid = false;
  int bgr = 0;
  mi355_hsv_settings hs{};       // settings the table was built for
  mi355_hsv_settings seen_hs{};  // settings of the previous call and for how many calls they (and the byte order) have not changed
  int seen_bgr = 0;
  unsigned seen_stable = 0;
  bool last_table = false;       // the last launch_hsvfilter call ran the table kernel
  AutoPick pick;
};

struct EchoDevice {
  double *d_ring = nullptr;   // n_streams rings of ring_len f64
  size_t ring_len = 0;
  size_t pos = 0;             // shared: every stream of a batch advances by the same number of samples per call
  int n_streams = 1;
  void *d_par = nullptr, *h_par = nullptr;  // per-stream {D, intensity, feedback}: device copy, pinned host copy
  hipEvent_t par_ev = nullptr;
  bool configured = false;
};

}  // namespace mi355

struct mi355_dssim_image;

struct mi355_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int n_cu = 256;
  // staging for the host entry points
  void *d_stage[2] = {nullptr, nullptr};
  size_t d_stage_bytes[2] = {0, 0};
  mi355::LutDevice lut;
  mi355::EchoDevice echo;
  void *ebur128 = nullptr;     // mi355::Ebur128State (ebur128_kernels.hip)
  void *hrtf = nullptr;        // mi355::HrtfState (hrtf_kernels.hip)
  void *sofa = nullptr;        // mi355::SofaState (sofa_kernels.hip)
  void *loudnorm = nullptr;    // mi355::LoudNormState (loudnorm.hip)
  void *loudnorm_batch = nullptr;  // mi355::LoudNormBatch (loudnorm.hip): n streams in lock step
  void *dssim_cache = nullptr; // mi355::DssimCache (dssim_kernels.hip)
  void *rounded = nullptr;     // mi355::RoundedMask (roundedcorners.hip): the element's alpha plane, device-resident
  // host <-> device copies this context has enqueued through the library's own entry points and mi355_buf objects (tests assert
  // that a chain of elements on device buffers costs ONE upload and ONE download: mi355_ctx_transfer_counts)
  unsigned long long n_h2d = 0, n_d2h = 0;
  bool force_generic = false;
  int fused_variant = 0;  // MI355_FLAG_FUSED_VARIANT
  int lut_variant = 0;    // MI355_FLAG_LUT_VARIANT
  int hsv_table_mode = 0; // MI355_FLAG_HSV_TABLE: 0 auto for GENERIC settings only (default), 1 auto for all, 2 table only, 3 off
  mi355::HsvTable hsv_table;
  int lut_stagger = 0;    // MI355_FLAG_LUT_STAGGER (x256 clock ticks)
  int brick_tiles_per_run = 0;  // MI355_FLAG_BRICK_TILES_PER_RUN (tuning; 0 = default)
  int brick_fold_axis = 2;      // MI355_FLAG_BRICK_FOLD_AXIS (accepted, ignored)
  hipEvent_t host_copy_ev = nullptr;  // marks "the caller's host buffer has been read" for entry points that return before their kernels end
  int dssim_translucent = 0;    // MI355_FLAG_DSSIM_TRANSLUCENT: 1 = alpha < 255 composed over black instead of over the crate's pattern
  int brick_prio = 3;           // MI355_FLAG_BRICK_PRIO: bit 0 progress-based wave priorities, bit 1 tile stealing within a block
  int brick_sets = 0;           // MI355_FLAG_BRICK_SETS: 0 = content watch decides (default); 32 (4x4x2 sets, 16 waves per CU) or 64 (4x4x4 sets, 8 waves per CU) pinned
  int hrtf_method = 0;         // MI355_FLAG_HRTF_METHOD: 0 = by HRIR length, 1 = overlap-save FFT, 2 = time-domain FIR (takes effect at mi355_hrtf_setup)
  int window_min_steps = mi355::kWindowMinStepsPerBlock;  // MI355_FLAG_WINDOW_MIN_STEPS
  int window_order = 1;         // MI355_FLAG_WINDOW_ORDER: how the blocks of colorlut_window_kernel share the steps (0 contiguous shares, 1 aligned fronts: default)
  bool window_stats_on = false; // MI355_FLAG_WINDOW_STATS: the LDS-cached kernels count {pixels, pixels past the cache, bricks installed} (three atomics per wave: off by default)
  unsigned long long *d_window_counters = nullptr;  // colorlut_window.hip: {pixels, pixels past the LDS cache, bricks installed} x 1024 slots
  int blockhash_any_size = 0;  // MI355_FLAG_BLOCKHASH_ANY_SIZE
  int hsv_nt = 0;              // MI355_FLAG_HSV_NT: 1 = the flat hsvfilter kernel loads and stores with the non-temporal hint (measurement A/B)
  mi355_hsv_settings group_fused_hs{};  // mi355_group_submit_fused: the settings of this stream's last fused submit ...
  unsigned group_fused_stable = 0;      // ... and how many submits in a row have carried them (a composed table is built for settings that stay)
  int hsv_blocks_per_cu = 64;  // grid cap of the flat hsvfilter kernel (tunable: MI355_FLAG_HSV_BLOCKS_PER_CU)
  std::string last_error;
};

namespace mi355 {

// frames of several streams handed to ONE launch (group.hip): base pointers by value in the kernel arguments
constexpr int kMultiFrames = 16;
struct MultiFramePtrs { uint8_t *p[kMultiFrames]; };

// pixel layout of a packed-RGB format
struct PixFmt {
  int pixel_stride;  // bytes per pixel: 3, 4 (8 for RGBA64)
  int first;         // byte offset of the colour triple
  int bgr;           // triple stored B,G,R
  int has_alpha;     // 4th byte is alpha (vs padding) — informational
};
bool pixfmt_of(int format, PixFmt *out);

int set_error(mi355_ctx *ctx, int status, const std::string &msg);
// Where a launch's input comes from decides which of the two table kernels reads it (colorlut_kernels.hip: launch_table_raw):
// note_written: a launch of this library that WRITES [p, p + bytes) has just been enqueued on `device`; recently_written: is
// [p, p + bytes) inside what one of the last two such launches wrote, and small enough to still be on-die (Infinity Cache)?
void note_written(int device, const void *p, size_t bytes);
bool recently_written(int device, const void *p, size_t bytes);
void forget_written(int device, const void *p, size_t bytes);   // a host copy has replaced that range
int check_hip(mi355_ctx *ctx, hipError_t e, const char *what);

// kernel launchers (asynchronous on ctx->stream)
int launch_hsvfilter(mi355_ctx *ctx, uint8_t *d_data, int n_frames, size_t frame_pitch, int width,
                     int height, int stride, const PixFmt &fmt, const mi355_hsv_settings &s);
int launch_hsvfilter_compute(mi355_ctx *ctx, uint8_t *d_data, int n_frames, size_t frame_pitch, int width,
                     int height, int stride, const PixFmt &fmt, const mi355_hsv_settings &s);
void hsv_table_release(mi355_ctx *ctx);
bool hsvfilter_multi_applicable(const uint8_t *const *frames, int n_frames, int width, int height, int stride, const PixFmt &fmt);
int launch_hsvfilter_multi(mi355_ctx *ctx, hipStream_t stream, uint8_t *const *frames, int n_frames, int width, int height, const PixFmt &fmt,
                           const mi355_hsv_settings &s);
// colorlut of n separate packed RGBA frames of one size through ctx's memoised (Morton) table, ONE launch on `stream`; the table
// is built on ctx->stream first if need be (*table_out = what the launch reads). MI355_ERR_UNSUPPORTED: not this path's geometry.
int colorlut_multi_table(mi355_ctx *ctx, const uint32_t **table_out);
int colorlut_multi_fused_table(mi355_ctx *ctx, const mi355_hsv_settings *hs, const uint32_t **table_out);
int launch_colorlut_multi(mi355_ctx *ctx, hipStream_t stream, const uint32_t *table, uint8_t *const *srcs, uint8_t *const *dsts, int n_frames, int width,
                          int height, bool from_hbm = false);
int launch_hsvdetect(mi355_ctx *ctx, const uint8_t *d_src, size_t src_pitch, int src_stride,
                     const PixFmt &sfmt, uint8_t *d_dst, size_t dst_pitch, int dst_stride,
                     int dst_alpha_first, int dst_bgr, int n_frames, int width, int height,
                     const mi355_hsvdetect_settings &s);
int launch_colorlut(mi355_ctx *ctx, const uint8_t *d_src, size_t src_pitch, int src_stride,
                    uint8_t *d_dst, size_t dst_pitch, int dst_stride, int n_frames, int width,
                    int height, int format);
int launch_hsv_colorlut(mi355_ctx *ctx, const uint8_t *d_src, size_t src_pitch, int src_stride,
                        uint8_t *d_dst, size_t dst_pitch, int dst_stride, int n_frames, int width,
                        int height, const mi355_hsv_settings &hs);
int lut_upload(mi355_ctx *ctx, int is3d, size_t size, const float *table, const float scale[3],
               const float offset[3]);
void lut_release(mi355_ctx *ctx);
int echo_setup(mi355_ctx *ctx, int n_streams, size_t ring_len);
void echo_release(mi355_ctx *ctx);
int launch_echo_batch(mi355_ctx *ctx, void *d_data, size_t stream_stride, size_t n, int is_f64, const size_t *delay, const double *intensity,
                      const double *feedback);
int launch_echo(mi355_ctx *ctx, void *d_data, size_t n, int is_f64, size_t delay, double intensity,
                double feedback);
int ebur128_setup(mi355_ctx *ctx, unsigned channels, unsigned rate, unsigned mode, const int *channel_class);
int ebur128_setup_batch(mi355_ctx *ctx, unsigned n_streams, unsigned channels, unsigned rate, unsigned mode, const int *channel_class);
int ebur128_add_frames_batch(mi355_ctx *ctx, const void *data, size_t frames, int fmt, int device_data);
int ebur128_query_batch(mi355_ctx *ctx, int what, double *out);
int ebur128_peak_batch(mi355_ctx *ctx, int true_peak, double *out);
int ebur128_reset(mi355_ctx *ctx);
int ebur128_reset_stream(mi355_ctx *ctx, unsigned stream);
int ebur128_add_frames_streams(mi355_ctx *ctx, const void *data, size_t slot_elems, const size_t *frames_per, int fmt, int device_data);
void ebur128_release(mi355_ctx *ctx);
int launch_blockhash(mi355_ctx *ctx, const uint8_t *d_frames, size_t frame_pitch, int stride, int n_frames, int width, int height,
                     int channels, unsigned long long *hashes);
int launch_imghash(mi355_ctx *ctx, const uint8_t *d_frames, size_t frame_pitch, int stride, int n_frames, int width, int height, int channels,
                   int algo, unsigned long long *hashes);
int loudnorm_setup(mi355_ctx *ctx, unsigned channels, double loudness_target, double loudness_range_target, double max_true_peak, double offset_db);
int loudnorm_push(mi355_ctx *ctx, const double *data, size_t frames, double *out, size_t out_cap_frames, size_t *out_frames);
int loudnorm_drain(mi355_ctx *ctx, double *out, size_t out_cap_frames, size_t *out_frames, int *eos);
void loudnorm_release(mi355_ctx *ctx);
int loudnorm_setup_batch(mi355_ctx *ctx, unsigned n_streams, unsigned channels, double loudness_target, double loudness_range_target, double max_true_peak, double offset_db);
size_t loudnorm_batch_frame_size(mi355_ctx *ctx);
size_t loudnorm_member_frame_size(mi355_ctx *ctx, unsigned stream);
int loudnorm_member_frame_type(mi355_ctx *ctx, unsigned stream);
int loudnorm_process_members(mi355_ctx *ctx, const unsigned char *members, const double *data, size_t stream_stride, size_t frames, double *out, size_t out_stride,
                             size_t out_cap_frames, size_t *out_frames, int device_data, int final_frame);
int loudnorm_process_batch(mi355_ctx *ctx, const double *data, size_t stream_stride, size_t frames, double *out, size_t out_stride, size_t out_cap_frames,
                           size_t *out_frames, int device_data, int final_frame);
void loudnorm_batch_release(mi355_ctx *ctx);
int dssim_create_image(mi355_ctx *ctx, const uint8_t *d_frame, int stride, int width, int height, int channels, mi355_dssim_image **out);
void dssim_free_image(mi355_ctx *ctx, mi355_dssim_image *img);
int dssim_compare(mi355_ctx *ctx, const mi355_dssim_image *a, const mi355_dssim_image *b, double *out);
int dssim_compare_frames(mi355_ctx *ctx, const mi355_dssim_image *a, const uint8_t *const *d_frames, int n_frames, int stride, int width, int height,
                         int channels, double *out);
int dssim_compare_pairs_enqueue(mi355_ctx *ctx, const uint8_t *const *d_refs, const uint8_t *const *d_frames, int n_pairs, int stride, int width, int height,
                                int channels, double *h_slots);
void dssim_scores_from_slots(int width, int height, const double *h_slots, int n_pairs, double *out);
constexpr int kDssimSlotDoubles = 15;  // per comparison: 5 scales x [sum, avg, dev]
int blockhash_enqueue(mi355_ctx *ctx, const uint8_t *const *d_frames, int n, int stride, int width, int height, int channels, uint32_t *d_sums,
                      unsigned long long *d_hashes);
int dssim_cbrt_selftest(mi355_ctx *ctx, uint32_t lo_bits, uint32_t hi_bits, uint64_t *mismatches);
void dssim_release(mi355_ctx *ctx);
void roundedcorners_release(mi355_ctx *ctx);
int dssim_image_plane(mi355_ctx *ctx, const mi355_dssim_image *img, int scale, int channel, int kind, float *out, int *w, int *h);
int hrtf_load_sphere(mi355_ctx *ctx, const unsigned char *bytes, size_t n, uint32_t device_rate);
int hrtf_setup(mi355_ctx *ctx, int channels, int block_len, int steps);
int hrtf_reset(mi355_ctx *ctx);
void hrtf_release(mi355_ctx *ctx);
int hrtf_process_block_device(mi355_ctx *ctx, const float *d_in, float *d_out, const float *positions, const float *gains);
int hrtf_process_block_host(mi355_ctx *ctx, const float *in, float *out, const float *positions, const float *gains);
int hrtf_last_lookup(mi355_ctx *ctx, int *faces, float *uvw);
int hrtf_info(mi355_ctx *ctx, uint32_t *len, uint32_t *vertices, uint32_t *faces);
int sofa_setup(mi355_ctx *ctx, int channels, int filter_len, int partition_len, int block_len);
int sofa_set_filter(mi355_ctx *ctx, int channel, const float *left, const float *right, int delay_left, int delay_right);
int sofa_set_drop(mi355_ctx *ctx, int channel, int drop);
int sofa_reset(mi355_ctx *ctx);
int sofa_process_block_device(mi355_ctx *ctx, const float *d_in, float *d_out, const float *gains);
int sofa_process_block_host(mi355_ctx *ctx, const float *in, float *out, const float *gains);
void sofa_release(mi355_ctx *ctx);
int ebur128_add_frames(mi355_ctx *ctx, const void *data, const void *const *planes, size_t frames, int fmt);
int ebur128_query(mi355_ctx *ctx, int what, double *out);
int ebur128_peak(mi355_ctx *ctx, int true_peak, unsigned channel, double *out);

}  // namespace mi355
