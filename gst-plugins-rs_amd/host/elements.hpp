// elements.hpp — host-side mirror of the reference elements' operator interface for the hot path.
//
// The reference elements are Rust GObject subclasses on gstreamer-rs; neither Rust nor the
// GStreamer/GLib headers exist in this image, so the element layer above the C ABI is restated here in
// C++ with the same names, property surface (docs/plugins/gst_plugins_cache.json), vfunc names,
// argument meaning and error behaviour:
//   HsvFilter    video/hsv/src/hsvfilter/imp.rs      (VideoFilter, AlwaysInPlace: transform_frame_ip)
//   HsvDetector  video/hsv/src/hsvdetector/imp.rs    (VideoFilter, NeverInPlace: transform_frame)
//   ColorLut     video/colorlut/src/colorlut/imp.rs  (VideoFilter, NeverInPlace: start/stop/transform_frame)
//   AudioEcho    audio/audiofx/src/audioecho/imp.rs  (AudioFilter, AlwaysInPlace: setup/transform_ip/stop)
//   HrtfRender   audio/hrtf/src/hrtf/imp.rs          (BaseTransform, NeverInPlace: set_caps/transform/drain/stop)
//   VideoCompare video/videofx/src/videocompare/imp.rs (VideoAggregator: aggregate_frames)
//   AudioLoudNorm audio/audiofx/src/audioloudnorm/imp.rs (Element: sink chain / drain)
//   RoundedCorners video/videofx/src/border/imp.rs    (BaseTransform: set_caps / prepare_output_buffer; host only)
// Each object owns one mi355_ctx (include/mi355fx.h) and forwards its per-buffer vfunc to the C ABI,
// exactly where the Rust element would call its inner loop. The GStreamer shim (gst/) wraps these.
#pragma once
#include <cstddef>
#include <cstdint>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mi355fx.h"

namespace mi355host {

// GstFlowReturn values (gst/gstpad.h)
enum class FlowReturn : int { Ok = 0, Eos = -3, NotNegotiated = -4, Error = -5 };

enum class PropType { Float, Double, UInt64, String, Flags, Boolean };
enum class Mutability { Ready, Playing };  // mutable_ready / mutable_playing

struct ParamSpec {
  std::string name, nick, blurb;
  PropType type;
  double def_num = 0, min_num = 0, max_num = 0;  // numeric types
  Mutability mutability = Mutability::Ready;
};

struct ElementMetadata {
  std::string long_name, klass, description, author;
};

// GstVideoFrame restricted to what the hot path reads: plane 0 of a packed-RGB frame.
struct VideoFrame {
  int format = MI355_FMT_RGBA;  // mi355_video_format
  int width = 0, height = 0;
  int stride = 0;               // plane_stride()[0]
  uint8_t *data = nullptr;      // plane_data(0)
  size_t size = 0;              // plane_data(0).len()
};

struct AudioInfo {
  int rate = 0, channels = 0;
  bool f64 = false;  // AUDIO_FORMAT_F64 vs AUDIO_FORMAT_F32 (interleaved)
};

class Element {
 public:
  virtual ~Element();
  virtual const char *factory_name() const = 0;   // "hsvfilter", ...
  virtual const char *type_name() const = 0;      // GType name "GstHsvFilter", ...
  virtual const ElementMetadata &metadata() const = 0;
  virtual const std::vector<ParamSpec> &properties() const = 0;
  virtual std::vector<int> sink_formats() const = 0;  // pad template caps (video formats / audio: F32,F64)
  virtual std::vector<int> src_formats() const = 0;

  // g_object_set / g_object_get for the element's own properties. Out-of-range values are rejected
  // (GLib warns and leaves the property unchanged); unknown names are an error.
  bool set_property(const std::string &name, double v);
  bool set_property(const std::string &name, const std::string &v);
  bool set_property_u64(const std::string &name, uint64_t v);
  bool get_property_u64(const std::string &name, uint64_t *v) const;
  bool get_property(const std::string &name, double *v) const;
  bool get_property(const std::string &name, std::string *v) const;

  // BaseTransform::start / stop (state change READY<->PAUSED)
  virtual bool start() { started_ = true; return true; }
  virtual bool stop() { started_ = false; return true; }
  bool started() const { return started_; }

  const std::string &last_error() const { return last_error_; }

 protected:
  explicit Element(int device);
  const ParamSpec *find_spec(const std::string &name) const;
  virtual bool store_number(const std::string &name, double v) = 0;
  virtual bool load_number(const std::string &name, double *v) const = 0;
  virtual bool store_u64(const std::string &, uint64_t) { return false; }
  virtual bool load_u64(const std::string &, uint64_t *) const { return false; }
  virtual bool store_string(const std::string &, const std::string &) { return false; }
  virtual bool load_string(const std::string &, std::string *) const { return false; }
  FlowReturn flow_from_status(int status);
  mi355_ctx *ctx_ = nullptr;
  mutable std::mutex settings_mutex_;  // settings behind a Mutex, snapshotted once per buffer
  bool started_ = false;
  std::string last_error_;
};

class HsvFilter final : public Element {
 public:
  explicit HsvFilter(int device);
  const char *factory_name() const override { return "hsvfilter"; }
  const char *type_name() const override { return "GstHsvFilter"; }
  const ElementMetadata &metadata() const override;
  const std::vector<ParamSpec> &properties() const override;
  std::vector<int> sink_formats() const override;
  std::vector<int> src_formats() const override { return sink_formats(); }
  // VideoFilterImpl::transform_frame_ip (hsvfilter/imp.rs:323-376)
  FlowReturn transform_frame_ip(VideoFrame &frame);

 private:
  bool store_number(const std::string &name, double v) override;
  bool load_number(const std::string &name, double *v) const override;
  mi355_hsv_settings settings_{0.0f, 1.0f, 0.0f, 1.0f, 0.0f};  // hsvfilter/imp.rs:25-29
};

class HsvDetector final : public Element {
 public:
  explicit HsvDetector(int device);
  const char *factory_name() const override { return "hsvdetector"; }
  const char *type_name() const override { return "GstHsvDetector"; }
  const ElementMetadata &metadata() const override;
  const std::vector<ParamSpec> &properties() const override;
  std::vector<int> sink_formats() const override;
  std::vector<int> src_formats() const override;
  // VideoFilterImpl::transform_frame (hsvdetector/imp.rs:423-707)
  FlowReturn transform_frame(const VideoFrame &in, VideoFrame &out);

 private:
  bool store_number(const std::string &name, double v) override;
  bool load_number(const std::string &name, double *v) const override;
  mi355_hsvdetect_settings settings_{0.0f, 10.0f, 0.0f, 0.15f, 0.0f, 0.3f};  // hsvdetector/imp.rs:25-30
};

class ColorLut final : public Element {
 public:
  explicit ColorLut(int device);
  const char *factory_name() const override { return "colorlut"; }
  const char *type_name() const override { return "GstColorLut"; }
  const ElementMetadata &metadata() const override;
  const std::vector<ParamSpec> &properties() const override;
  std::vector<int> sink_formats() const override;
  std::vector<int> src_formats() const override { return sink_formats(); }
  bool start() override;  // parse `location`, install the LUT (colorlut/imp.rs:168-194)
  bool stop() override;   // drop the LUT (colorlut/imp.rs:196-199)
  // VideoFilterImpl::transform_frame (colorlut/imp.rs:203-223)
  FlowReturn transform_frame(const VideoFrame &in, VideoFrame &out);

 private:
  bool store_number(const std::string &, double) override { return false; }
  bool load_number(const std::string &, double *) const override { return false; }
  bool store_string(const std::string &name, const std::string &v) override;
  bool load_string(const std::string &name, std::string *v) const override;
  bool have_location_ = false;
  std::string location_;
};

class AudioEcho final : public Element {
 public:
  explicit AudioEcho(int device);
  const char *factory_name() const override { return "rsaudioecho"; }
  const char *type_name() const override { return "GstRsAudioEcho"; }
  const ElementMetadata &metadata() const override;
  const std::vector<ParamSpec> &properties() const override;
  std::vector<int> sink_formats() const override { return {0, 1}; }  // F32, F64 interleaved
  std::vector<int> src_formats() const override { return {0, 1}; }
  // AudioFilterImpl::setup (audioecho/imp.rs:248-259)
  bool setup(const AudioInfo &info);
  // BaseTransformImpl::transform_ip (audioecho/imp.rs:205-227): `data` = the mapped buffer bytes
  FlowReturn transform_ip(void *data, size_t nbytes);
  bool stop() override;  // drops the state (audioecho/imp.rs:229-234)

 private:
  bool store_number(const std::string &name, double v) override;
  bool load_number(const std::string &name, double *v) const override;
  bool store_u64(const std::string &name, uint64_t v) override;
  bool load_u64(const std::string &name, uint64_t *v) const override;
  // audioecho/imp.rs:31-34
  uint64_t max_delay_ns_ = 1000000000ull, delay_ns_ = 500ull * 1000000000ull;
  double intensity_ = 0.5, feedback_ = 0.0;
  bool have_state_ = false;
  AudioInfo info_;
};

// ebur128level (audio/audiofx/src/ebur128level/imp.rs): AudioFilter in passthrough mode; chops the
// buffers at `interval` boundaries, feeds the loudness meter and queues one "ebur128-level" message per
// full interval (imp.rs:296-486).
struct EbuR128LevelMessage {
  uint64_t timestamp = 0;  // ns: the time until which measurements are included (imp.rs:353-361)
  unsigned fields = 0;     // bit set of present fields == the mode bits
  double momentary_loudness = 0, shortterm_loudness = 0, global_loudness = 0, relative_threshold = 0, loudness_range = 0;
  std::vector<double> sample_peak, true_peak;
};

class EbuR128Level final : public Element {
 public:
  explicit EbuR128Level(int device);
  const char *factory_name() const override { return "ebur128level"; }
  const char *type_name() const override { return "GstEbuR128Level"; }
  const ElementMetadata &metadata() const override;
  const std::vector<ParamSpec> &properties() const override;
  std::vector<int> sink_formats() const override { return {0, 1, 2, 3}; }  // S16, S32, F32, F64 (x interleaved / planar)
  std::vector<int> src_formats() const override { return {0, 1, 2, 3}; }
  // AudioFilterImpl::setup (imp.rs:513-613). sample_format 0..3; channel_class as mi355_ebur128_setup
  // (nullptr = no channel positions: every channel weighted like Center, imp.rs:589-595)
  bool setup(int rate, int channels, int sample_format, bool planar, const int *channel_class);
  // BaseTransformImpl::transform_ip_passthrough (imp.rs:296-486)
  FlowReturn transform_ip_passthrough(const void *data, const void *const *planes, size_t frames, uint64_t pts_ns);
  void reset_signal() { reset_requested_ = true; }  // the `reset` action signal (imp.rs:124-139)
  bool pop_message(EbuR128LevelMessage *out);
  bool stop() override;

 private:
  bool store_number(const std::string &name, double v) override;
  bool load_number(const std::string &name, double *v) const override;
  bool store_u64(const std::string &name, uint64_t v) override;
  bool load_u64(const std::string &name, uint64_t *v) const override;
  unsigned mode_ = 63;             // DEFAULT_MODE = Mode::all() (imp.rs:80)
  bool post_messages_ = true;      // imp.rs:81
  uint64_t interval_ns_ = 1000000000ull;  // imp.rs:82
  bool have_state_ = false, reset_requested_ = false;
  int rate_ = 0, channels_ = 0, format_ = 2;
  bool planar_ = false;
  unsigned state_mode_ = 0;
  uint64_t num_frames_ = 0, interval_frames_ = 0, interval_frames_remaining_ = 0;
  std::vector<EbuR128LevelMessage> queue_;
};

// hrtfrender (audio/hrtf/src/hrtf/imp.rs): BaseTransform, NeverInPlace; input goes through an adapter and is
// rendered in blocks of block-length * interpolation-steps frames; EOS drains the rest zero-padded.
struct SpatialObject {      // audio/hrtf/src/spatial.rs:118-137
  int coordinate_system = 1;  // GstHrtfCoordinateSystem: 0 cartesian, 1 left-handed (default), 2 right-handed
  float x = 0, y = 0, z = 0;
  float distance_gain = 1.0f;
};

class HrtfRender final : public Element {
 public:
  explicit HrtfRender(int device);
  const char *factory_name() const override { return "hrtfrender"; }
  const char *type_name() const override { return "GstHrtfRender"; }
  const ElementMetadata &metadata() const override;
  const std::vector<ParamSpec> &properties() const override;
  std::vector<int> sink_formats() const override { return {0}; }  // F32 interleaved, 1..=64 channels
  std::vector<int> src_formats() const override { return {0}; }   // F32 interleaved, 2 channels
  // "hrir-raw" (glib::Bytes) and "spatial-objects" (GstValueArray of GstStructure) have no scalar spelling
  void set_hrir_raw(const void *bytes, size_t len);
  bool set_spatial_objects(const std::vector<SpatialObject> &objs);  // ignored with a warning on a channel-count mismatch once negotiated
  std::vector<SpatialObject> spatial_objects() const;
  // BaseTransformImpl::set_caps (imp.rs:628-693). positions: GstAudioChannelPosition per channel, or nullptr
  // (unpositioned caps: spatial-objects must have been set)
  bool set_caps(int rate, int channels, const int *positions);
  // transform_size (imp.rs:574-599): output bytes for `in_bytes` more input
  size_t transform_size(size_t in_bytes) const;
  // transform (imp.rs:547-572): push into the adapter, render every complete block; `out` receives
  // transform_size(in_bytes) bytes of interleaved stereo
  FlowReturn transform(const float *in, size_t in_bytes, std::vector<float> *out);
  // sink_event(Eos) -> drain (imp.rs:281-352): the remainder zero-padded to a block, output truncated to the
  // real frame count; tails are reset afterwards
  FlowReturn drain(std::vector<float> *out);
  void flush_stop();  // sink_event(FlushStop) (imp.rs:710-718)
  bool stop() override;

 private:
  bool store_number(const std::string &name, double v) override;
  bool load_number(const std::string &name, double *v) const override;
  bool store_u64(const std::string &name, uint64_t v) override;
  bool load_u64(const std::string &name, uint64_t *v) const override;
  bool store_string(const std::string &name, const std::string &v) override;
  bool load_string(const std::string &name, std::string *v) const override;
  FlowReturn process_available(std::vector<float> *out);
  uint64_t interpolation_steps_ = 8, block_length_ = 512;  // imp.rs:36-37
  bool use_rayon_ = false;
  std::vector<SpatialObject> objects_;
  bool have_objects_ = false;
  std::vector<unsigned char> hrir_raw_;
  bool have_hrir_raw_ = false, have_hrir_file_ = false;
  std::string hrir_file_;
  bool have_state_ = false;
  int rate_ = 0, channels_ = 0;
  size_t block_samples_ = 0;
  std::vector<float> adapter_;
};

// audioloudnorm (audio/audiofx/src/audioloudnorm/imp.rs): plain Element with its own chain function; F64 interleaved
// at 192 kHz only; 3 s of latency.
class AudioLoudNorm final : public Element {
 public:
  explicit AudioLoudNorm(int device);
  const char *factory_name() const override { return "audioloudnorm"; }
  const char *type_name() const override { return "GstAudioLoudNorm"; }
  const ElementMetadata &metadata() const override;
  const std::vector<ParamSpec> &properties() const override;
  std::vector<int> sink_formats() const override { return {1}; }  // F64 interleaved, rate 192000
  std::vector<int> src_formats() const override { return {1}; }
  // sink_event(Caps) (imp.rs:1613-1650): a new State from the current settings; rate must be 192000
  bool set_caps(int rate, int channels);
  // sink_chain (imp.rs:1543-1611): `out` receives the frames completed by this buffer (possibly none)
  FlowReturn chain(const double *data, size_t frames, std::vector<double> *out);
  // sink_event(Eos / FlushStop...) -> drain (imp.rs:270-310); Eos when there was nothing to drain
  FlowReturn drain(std::vector<double> *out);
  bool stop() override;

 private:
  bool store_number(const std::string &name, double v) override;
  bool load_number(const std::string &name, double *v) const override;
  double loudness_target_ = -24.0, loudness_range_target_ = 7.0, max_true_peak_ = -2.0, offset_ = 0.0;  // imp.rs:37-40
  bool have_state_ = false;
  int channels_ = 0;
};

// roundedcorners (video/videofx/src/border/imp.rs): BaseTransform, AlwaysInPlace, I420 in -> I420 (passthrough) or A420.
// There is NO per-pixel work per buffer in the reference: an A8 alpha plane is rendered once per caps / radius
// change with cairo (four arcs, antialiased fill + 1 px stroke, imp.rs:57-180) and the same memory is appended to every
// output buffer (imp.rs:444-480, :482-559). The mask bytes are defined by cairo's rasteriser, so this mirror renders
// them the same way — through the system libcairo, loaded at run time. The device holds a copy of the plane for
// device-resident streams (csrc/roundedcorners.hip): prepare_output_buffer_device.
class RoundedCorners final : public Element {
 public:
  explicit RoundedCorners(int device);
  ~RoundedCorners() override;
  const char *factory_name() const override { return "roundedcorners"; }
  const char *type_name() const override { return "GstRoundedCorners"; }
  const ElementMetadata &metadata() const override;
  const std::vector<ParamSpec> &properties() const override;
  std::vector<int> sink_formats() const override { return {100}; }        // I420
  std::vector<int> src_formats() const override { return {100, 101}; }    // I420, A420
  // transform_caps sink->src (imp.rs:398-425): formats offered downstream for the current border radius
  std::vector<int> transform_caps_to_src() const;
  // set_caps (imp.rs:444-480): out_format 100 = I420 (passthrough) or 101 = A420 (alpha plane allocated,
  // stride = round_up_4(width), rows = round_up_2(height))
  bool set_caps(int width, int height, int out_format);
  bool passthrough() const { return passthrough_; }
  // prepare_output_buffer (imp.rs:482-559): regenerates the mask when the radius changed, then hands out the shared
  // alpha plane that the element appends to the buffer (plane 3 of A420)
  FlowReturn prepare_output_buffer(const uint8_t **alpha, size_t *size, int *stride);
  // the same for device-resident frames: the plane is kept in HBM (uploaded when it was regenerated) and written behind the
  // I420 planes of n_frames frames (frame f's plane 3 at d_frames + f * frame_pitch + alpha_offset) by one launch
  // (mi355_roundedcorners_set_mask / _append_device). Passthrough: nothing is written.
  FlowReturn prepare_output_buffer_device(uint8_t *d_frames, size_t frame_pitch, size_t alpha_offset, int n_frames);
  bool stop() override;

 private:
  bool store_number(const std::string &, double) override { return false; }
  bool load_number(const std::string &, double *) const override { return false; }
  bool store_u64(const std::string &name, uint64_t v) override;
  bool load_u64(const std::string &name, uint64_t *v) const override;
  bool generate_alpha_mask(uint32_t radius);
  uint32_t border_radius_px_ = 0;  // DEFAULT_BORDER_RADIUS (imp.rs:27)
  bool changed_ = false, have_state_ = false, passthrough_ = true;
  bool device_mask_stale_ = true;  // the plane in HBM is not the one in alpha_mem_
  int width_ = 0, height_ = 0, alpha_stride_ = 0;
  std::vector<uint8_t> alpha_mem_;
  void *cairo_ = nullptr;  // dlopen handle
};

// videocompare (video/videofx/src/videocompare/imp.rs): VideoAggregator; the first sink pad is the reference, every
// other pad's current frame is hashed and compared with it; one "videocompare" message is posted when any pad is
// within max-dist-threshold.
struct VideoCompareMessage {   // videocompare/mod.rs:104-170
  struct PadDistance { std::string pad; double distance = 0; };
  std::vector<PadDistance> pad_distances;
  bool have_running_time = false;
  uint64_t running_time = 0;
};

class VideoCompare final : public Element {
 public:
  explicit VideoCompare(int device);
  const char *factory_name() const override { return "videocompare"; }
  const char *type_name() const override { return "GstVideoCompare"; }
  const ElementMetadata &metadata() const override;
  const std::vector<ParamSpec> &properties() const override;
  std::vector<int> sink_formats() const override { return {MI355_FMT_RGB, MI355_FMT_RGBA}; }
  std::vector<int> src_formats() const override { return {MI355_FMT_RGB, MI355_FMT_RGBA}; }
  // VideoAggregatorImpl::aggregate_frames (imp.rs:259-390). frames[0] is the reference pad's prepared frame,
  // frames[k] the one of pad "sink_k". Returns Ok with *posted == true and *msg filled when a message is posted.
  FlowReturn aggregate_frames(const std::vector<VideoFrame> &frames, bool have_running_time, uint64_t running_time,
                              VideoCompareMessage *msg, bool *posted);

 private:
  bool store_number(const std::string &name, double v) override;
  bool load_number(const std::string &name, double *v) const override;
  bool store_string(const std::string &name, const std::string &v) override;   // "hash-algo" by nick
  bool load_string(const std::string &name, std::string *v) const override;
  int hash_algo_ = MI355_HASH_BLOCKHASH;  // DEFAULT_HASH_ALGO (imp.rs:31)
  double max_dist_threshold_ = 0.0;       // imp.rs:32
};

// gst_element_factory_make(): nullptr for an unknown factory name or when no device context can be made.
std::unique_ptr<Element> element_factory_make(const std::string &factory, int device, std::string *error);
// Names registered by the plugins (video/hsv/src/lib.rs:23-27, video/colorlut/src/lib.rs, audio/audiofx/src/lib.rs)
std::vector<std::string> registered_factories();

}  // namespace mi355host
