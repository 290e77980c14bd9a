// elements.cpp — see elements.hpp. Property tables transcribe docs/plugins/gst_plugins_cache.json
// (hsvfilter :6532, hsvdetector :6418, colorlut :2979, rsaudioecho :11757).
#include "elements.hpp"

#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>

#include "mi355fx_host.h"

namespace mi355host {

// ------------------------------------------------------------------ Element

Element::Element(int device) {
  int st = 0;
  ctx_ = mi355_ctx_create(device, &st);
  if (!ctx_) last_error_ = std::string("cannot create device context: ") + mi355_status_string(st);
}

Element::~Element() {
  if (ctx_) mi355_ctx_destroy(ctx_);
}

const ParamSpec *Element::find_spec(const std::string &name) const {
  for (const auto &p : properties())
    if (p.name == name) return &p;
  return nullptr;
}

bool Element::set_property(const std::string &name, double v) {
  const ParamSpec *ps = find_spec(name);
  if (!ps || (ps->type != PropType::Float && ps->type != PropType::Double && ps->type != PropType::Flags && ps->type != PropType::Boolean)) {
    last_error_ = "no numeric property '" + name + "'";
    return false;
  }
  if (ps->mutability == Mutability::Ready && started_ && (ps->type == PropType::Flags)) {
    last_error_ = "property '" + name + "' can only be changed in NULL or READY state";
    return false;
  }
  // g_param_value_validate clamps to [min,max] and g_object_set rejects (with a warning) any value the
  // clamp changed — that covers out-of-range numbers, infinities and NaN (NaN != NaN after the clamp).
  if (ps->type == PropType::Float && std::isfinite(v) && std::fabs(v) > (double)FLT_MAX) v = std::copysign(INFINITY, v);
  if (!(v >= ps->min_num && v <= ps->max_num)) {
    last_error_ = "value out of range for property '" + name + "'";
    return false;
  }
  std::lock_guard<std::mutex> g(settings_mutex_);
  return store_number(name, v);
}

bool Element::set_property_u64(const std::string &name, uint64_t v) {
  const ParamSpec *ps = find_spec(name);
  if (!ps || ps->type != PropType::UInt64) {
    last_error_ = "no guint64 property '" + name + "'";
    return false;
  }
  if (v == UINT64_MAX) {  // maximum(u64::MAX - 1)
    last_error_ = "value out of range for property '" + name + "'";
    return false;
  }
  std::lock_guard<std::mutex> g(settings_mutex_);
  return store_u64(name, v);
}

bool Element::get_property_u64(const std::string &name, uint64_t *v) const {
  const ParamSpec *ps = find_spec(name);
  if (!ps || ps->type != PropType::UInt64) return false;
  std::lock_guard<std::mutex> g(settings_mutex_);
  return load_u64(name, v);
}

bool Element::set_property(const std::string &name, const std::string &v) {
  const ParamSpec *ps = find_spec(name);
  if (!ps || ps->type != PropType::String) {
    last_error_ = "no string property '" + name + "'";
    return false;
  }
  // mutable_ready properties cannot change once the element has left READY
  if (ps->mutability == Mutability::Ready && started_) {
    last_error_ = "property '" + name + "' can only be changed in NULL or READY state";
    return false;
  }
  std::lock_guard<std::mutex> g(settings_mutex_);
  return store_string(name, v);
}

bool Element::get_property(const std::string &name, double *v) const {
  const ParamSpec *ps = find_spec(name);
  if (!ps || (ps->type != PropType::Float && ps->type != PropType::Double && ps->type != PropType::Flags && ps->type != PropType::Boolean)) return false;
  std::lock_guard<std::mutex> g(settings_mutex_);
  return load_number(name, v);
}

bool Element::get_property(const std::string &name, std::string *v) const {
  const ParamSpec *ps = find_spec(name);
  if (!ps || ps->type != PropType::String) return false;
  std::lock_guard<std::mutex> g(settings_mutex_);
  return load_string(name, v);
}

FlowReturn Element::flow_from_status(int status) {
  if (status == MI355_OK) return FlowReturn::Ok;
  last_error_ = ctx_ ? mi355_ctx_last_error(ctx_) : "no device context";
  if (status == MI355_ERR_NOT_CONFIGURED && std::strcmp(factory_name(), "rsaudioecho") == 0) return FlowReturn::NotNegotiated;
  return FlowReturn::Error;
}

static ParamSpec float_spec(const char *name, const char *nick, const char *blurb, double def, double lo, double hi,
                            Mutability m) {
  ParamSpec p;
  p.name = name; p.nick = nick; p.blurb = blurb; p.type = PropType::Float;
  p.def_num = def; p.min_num = lo; p.max_num = hi; p.mutability = m;
  return p;
}

// ------------------------------------------------------------------ HsvFilter

HsvFilter::HsvFilter(int device) : Element(device) {}

const ElementMetadata &HsvFilter::metadata() const {
  static const ElementMetadata m{"HSV filter", "Filter/Effect/Converter/Video",
                                 "Works within the HSV colorspace to apply transformations to incoming frames",
                                 "Julien Bardagi <julien.bardagi@gmail.com>"};
  return m;
}

const std::vector<ParamSpec> &HsvFilter::properties() const {
  static const std::vector<ParamSpec> p = {
      float_spec("hue-shift", "Hue shift", "Hue shifting in degrees", 0.0, -FLT_MAX, FLT_MAX, Mutability::Playing),
      float_spec("saturation-mul", "Saturation multiplier", "Saturation multiplier to apply to the saturation value (before offset)", 1.0, -FLT_MAX, FLT_MAX, Mutability::Playing),
      float_spec("saturation-off", "Saturation offset", "Saturation offset to add to the saturation value (after multiplier)", 0.0, -FLT_MAX, FLT_MAX, Mutability::Playing),
      float_spec("value-mul", "Value multiplier", "Value multiplier to apply to the value (before offset)", 1.0, -FLT_MAX, FLT_MAX, Mutability::Playing),
      float_spec("value-off", "Value offset", "Value offset to add to the value (after multiplier)", 0.0, -FLT_MAX, FLT_MAX, Mutability::Playing),
  };
  return p;
}

std::vector<int> HsvFilter::sink_formats() const {
  return {MI355_FMT_RGBX, MI355_FMT_XRGB, MI355_FMT_BGRX, MI355_FMT_XBGR, MI355_FMT_RGBA,
          MI355_FMT_ARGB, MI355_FMT_BGRA, MI355_FMT_ABGR, MI355_FMT_RGB, MI355_FMT_BGR};
}

bool HsvFilter::store_number(const std::string &n, double v) {
  const float f = (float)v;
  if (n == "hue-shift") settings_.hue_shift = f;
  else if (n == "saturation-mul") settings_.saturation_mul = f;
  else if (n == "saturation-off") settings_.saturation_off = f;
  else if (n == "value-mul") settings_.value_mul = f;
  else if (n == "value-off") settings_.value_off = f;
  else return false;
  return true;
}

bool HsvFilter::load_number(const std::string &n, double *v) const {
  if (n == "hue-shift") *v = settings_.hue_shift;
  else if (n == "saturation-mul") *v = settings_.saturation_mul;
  else if (n == "saturation-off") *v = settings_.saturation_off;
  else if (n == "value-mul") *v = settings_.value_mul;
  else if (n == "value-off") *v = settings_.value_off;
  else return false;
  return true;
}

FlowReturn HsvFilter::transform_frame_ip(VideoFrame &frame) {
  if (!ctx_) return FlowReturn::Error;
  bool ok = false;
  for (int f : sink_formats()) ok |= (f == frame.format);
  if (!ok) {  // `_ => unreachable!()` in the reference: negotiation guarantees a template format
    last_error_ = "hsvfilter: format not in the pad template";
    return FlowReturn::NotNegotiated;
  }
  mi355_hsv_settings snap;
  {
    std::lock_guard<std::mutex> g(settings_mutex_);  // `let settings = *self.settings.lock().unwrap();`
    snap = settings_;
  }
  return flow_from_status(mi355_hsvfilter_frame_ip(ctx_, frame.data, frame.size, frame.width, frame.stride, frame.format, &snap));
}

// ------------------------------------------------------------------ HsvDetector

HsvDetector::HsvDetector(int device) : Element(device) {}

const ElementMetadata &HsvDetector::metadata() const {
  static const ElementMetadata m{"HSV detector", "Filter/Effect/Converter/Video",
                                 "Works within the HSV colorspace to mark positive pixels",
                                 "Julien Bardagi <julien.bardagi@gmail.com>"};
  return m;
}

const std::vector<ParamSpec> &HsvDetector::properties() const {
  static const std::vector<ParamSpec> p = {
      float_spec("hue-ref", "Hue reference", "Hue reference in degrees", 0.0, -FLT_MAX, FLT_MAX, Mutability::Playing),
      float_spec("hue-var", "Hue variation", "Allowed hue variation from the reference hue angle, in degrees", 10.0, 0.0, 180.0, Mutability::Playing),
      float_spec("saturation-ref", "Saturation reference", "Reference saturation value", 0.0, 0.0, 1.0, Mutability::Playing),
      float_spec("saturation-var", "Saturation variation", "Allowed saturation variation from the reference value", 0.15, 0.0, 1.0, Mutability::Playing),
      float_spec("value-ref", "Value reference", "Reference value value", 0.0, 0.0, 1.0, Mutability::Playing),
      float_spec("value-var", "Value variation", "Allowed value variation from the reference value", 0.3, 0.0, 1.0, Mutability::Playing),
  };
  return p;
}

std::vector<int> HsvDetector::sink_formats() const {
  return {MI355_FMT_RGBX, MI355_FMT_XRGB, MI355_FMT_BGRX, MI355_FMT_XBGR, MI355_FMT_RGB, MI355_FMT_BGR};
}
std::vector<int> HsvDetector::src_formats() const { return {MI355_FMT_RGBA, MI355_FMT_ARGB, MI355_FMT_BGRA, MI355_FMT_ABGR}; }

bool HsvDetector::store_number(const std::string &n, double v) {
  const float f = (float)v;
  if (n == "hue-ref") settings_.hue_ref = f;
  else if (n == "hue-var") settings_.hue_var = f;
  else if (n == "saturation-ref") settings_.saturation_ref = f;
  else if (n == "saturation-var") settings_.saturation_var = f;
  else if (n == "value-ref") settings_.value_ref = f;
  else if (n == "value-var") settings_.value_var = f;
  else return false;
  return true;
}

bool HsvDetector::load_number(const std::string &n, double *v) const {
  if (n == "hue-ref") *v = settings_.hue_ref;
  else if (n == "hue-var") *v = settings_.hue_var;
  else if (n == "saturation-ref") *v = settings_.saturation_ref;
  else if (n == "saturation-var") *v = settings_.saturation_var;
  else if (n == "value-ref") *v = settings_.value_ref;
  else if (n == "value-var") *v = settings_.value_var;
  else return false;
  return true;
}

FlowReturn HsvDetector::transform_frame(const VideoFrame &in, VideoFrame &out) {
  if (!ctx_) return FlowReturn::Error;
  mi355_hsvdetect_settings snap;
  {
    std::lock_guard<std::mutex> g(settings_mutex_);
    snap = settings_;
  }
  return flow_from_status(mi355_hsvdetect_frame(ctx_, in.data, in.size, in.stride, in.format, out.data, out.size, out.stride,
                                                out.format, in.width, &snap));
}

// ------------------------------------------------------------------ ColorLut

ColorLut::ColorLut(int device) : Element(device) {}

const ElementMetadata &ColorLut::metadata() const {
  static const ElementMetadata m{"Color LUT", "Filter/Effect/Video", "Apply color lookup table",
                                 "Seungha Yang <seungha@centricular.com>"};
  return m;
}

const std::vector<ParamSpec> &ColorLut::properties() const {
  static const std::vector<ParamSpec> p = [] {
    ParamSpec s;
    s.name = "location"; s.nick = "Location"; s.blurb = "Location of the LUT file to read from";
    s.type = PropType::String; s.mutability = Mutability::Ready;
    return std::vector<ParamSpec>{s};
  }();
  return p;
}

std::vector<int> ColorLut::sink_formats() const { return {MI355_FMT_RGBA64_LE, MI355_FMT_RGBA64_BE, MI355_FMT_RGBA}; }

bool ColorLut::store_string(const std::string &n, const std::string &v) {
  if (n != "location") return false;
  location_ = v;
  have_location_ = true;
  return true;
}
bool ColorLut::load_string(const std::string &n, std::string *v) const {
  if (n != "location" || !have_location_) return false;
  *v = location_;
  return true;
}

bool ColorLut::start() {
  if (!ctx_) return false;
  std::string loc;
  {
    std::lock_guard<std::mutex> g(settings_mutex_);
    if (!have_location_) {  // ResourceError::Settings (colorlut/imp.rs:175-180)
      last_error_ = "LUT file location is not configured";
      return false;
    }
    loc = location_;
  }
  char err[512] = {0};
  mi355h_cube *cube = mi355h_cube_parse_file(loc.c_str(), err, sizeof err);
  if (!cube) {  // ResourceError::Read (colorlut/imp.rs:182-187)
    last_error_ = "Failed to parse LUT file " + loc + ": " + err;
    return false;
  }
  float scale[3], offset[3];
  mi355h_cube_domain(cube, scale, offset);
  const int rc = mi355_colorlut_load(ctx_, mi355h_cube_is3d(cube), mi355h_cube_size(cube), mi355h_cube_table(cube), scale, offset);
  mi355h_cube_free(cube);
  if (rc != MI355_OK) {
    last_error_ = mi355_ctx_last_error(ctx_);
    return false;
  }
  started_ = true;
  return true;
}

bool ColorLut::stop() {
  if (ctx_) mi355_colorlut_unload(ctx_);
  started_ = false;
  return true;
}

FlowReturn ColorLut::transform_frame(const VideoFrame &in, VideoFrame &out) {
  if (!ctx_) return FlowReturn::Error;
  if (in.format != out.format || in.width != out.width || in.height != out.height) {
    last_error_ = "colorlut: input and output caps differ";
    return FlowReturn::NotNegotiated;
  }
  return flow_from_status(mi355_colorlut_frame(ctx_, in.data, in.stride, out.data, out.stride, in.width, in.height, in.format));
}

// ------------------------------------------------------------------ AudioEcho

AudioEcho::AudioEcho(int device) : Element(device) {}

const ElementMetadata &AudioEcho::metadata() const {
  static const ElementMetadata m{"Audio echo", "Filter/Effect/Audio", "Adds an echo or reverb effect to an audio stream",
                                 "Sebastian Dröge <sebastian@centricular.com>"};
  return m;
}

const std::vector<ParamSpec> &AudioEcho::properties() const {
  static const std::vector<ParamSpec> p = [] {
    auto u64 = [](const char *n, const char *nick, const char *blurb, double def) {
      ParamSpec s;
      s.name = n; s.nick = nick; s.blurb = blurb; s.type = PropType::UInt64;
      s.def_num = def; s.min_num = 0; s.max_num = 18446744073709551614.0; s.mutability = Mutability::Ready;
      return s;
    };
    auto dbl = [](const char *n, const char *nick, const char *blurb, double def) {
      ParamSpec s;
      s.name = n; s.nick = nick; s.blurb = blurb; s.type = PropType::Double;
      s.def_num = def; s.min_num = 0.0; s.max_num = 1.0; s.mutability = Mutability::Ready;
      return s;
    };
    return std::vector<ParamSpec>{
        u64("max-delay", "Maximum Delay", "Maximum delay of the echo in nanoseconds (can't be changed in PLAYING or PAUSED state)", 1e9),
        u64("delay", "Delay", "Delay of the echo in nanoseconds", 5e11),
        dbl("intensity", "Intensity", "Intensity of the echo", 0.5),
        dbl("feedback", "Feedback", "Amount of feedback", 0.0),
    };
  }();
  return p;
}

bool AudioEcho::store_number(const std::string &n, double v) {
  if (n == "intensity") intensity_ = v;
  else if (n == "feedback") feedback_ = v;
  else return false;
  return true;
}
bool AudioEcho::load_number(const std::string &n, double *v) const {
  if (n == "intensity") *v = intensity_;
  else if (n == "feedback") *v = feedback_;
  else return false;
  return true;
}
bool AudioEcho::store_u64(const std::string &n, uint64_t v) {
  if (n == "max-delay") {
    if (!have_state_) max_delay_ns_ = v;  // only while there is no state (audioecho/imp.rs:137-142)
  } else if (n == "delay") {
    delay_ns_ = v;
  } else {
    return false;
  }
  return true;
}
bool AudioEcho::load_u64(const std::string &n, uint64_t *v) const {
  if (n == "max-delay") *v = max_delay_ns_;
  else if (n == "delay") *v = delay_ns_;
  else return false;
  return true;
}

bool AudioEcho::setup(const AudioInfo &info) {
  if (!ctx_) return false;
  uint64_t max_delay;
  {
    std::lock_guard<std::mutex> g(settings_mutex_);
    max_delay = max_delay_ns_;
  }
  // size = (max_delay * rate).seconds(); buffer_size = size * channels (audioecho/imp.rs:250-251)
  const uint64_t size = (max_delay * (uint64_t)info.rate) / 1000000000ull;
  const size_t ring_len = (size_t)size * (size_t)info.channels;
  if (mi355_echo_setup(ctx_, ring_len) != MI355_OK) {
    last_error_ = mi355_ctx_last_error(ctx_);
    return false;
  }
  info_ = info;
  have_state_ = true;
  return true;
}

FlowReturn AudioEcho::transform_ip(void *data, size_t nbytes) {
  if (!ctx_) return FlowReturn::Error;
  uint64_t delay, max_delay;
  double intensity, feedback;
  {
    std::lock_guard<std::mutex> g(settings_mutex_);
    delay = delay_ns_; max_delay = max_delay_ns_; intensity = intensity_; feedback = feedback_;
  }
  if (delay > max_delay) delay = max_delay;  // cmp::min(settings.max_delay, settings.delay) (imp.rs:207)
  if (!have_state_) {                         // ok_or(FlowError::NotNegotiated) (imp.rs:210)
    last_error_ = "rsaudioecho: not negotiated";
    return FlowReturn::NotNegotiated;
  }
  // delay_frames = (delay * channels * rate).seconds() (imp.rs:74-77)
  const size_t delay_samples = (size_t)((delay * (uint64_t)info_.channels * (uint64_t)info_.rate) / 1000000000ull);
  int rc;
  if (info_.f64) rc = mi355_echo_process_f64(ctx_, (double *)data, nbytes / sizeof(double), delay_samples, intensity, feedback);
  else rc = mi355_echo_process_f32(ctx_, (float *)data, nbytes / sizeof(float), delay_samples, intensity, feedback);
  return flow_from_status(rc);
}

bool AudioEcho::stop() {
  if (ctx_) mi355_echo_reset(ctx_);
  have_state_ = false;
  started_ = false;
  return true;
}

// ------------------------------------------------------------------ EbuR128Level

EbuR128Level::EbuR128Level(int device) : Element(device) {}

const ElementMetadata &EbuR128Level::metadata() const {
  static const ElementMetadata m{"EBU R128 Loudness Level Measurement", "Filter/Analyzer/Audio",
                                 "Measures different loudness metrics according to EBU R128",
                                 "Sebastian Dröge <sebastian@centricular.com>"};
  return m;
}

const std::vector<ParamSpec> &EbuR128Level::properties() const {
  static const std::vector<ParamSpec> p = [] {
    ParamSpec mode;
    mode.name = "mode"; mode.nick = "Mode"; mode.blurb = "Selection of metrics to calculate";
    mode.type = PropType::Flags; mode.def_num = 63; mode.min_num = 0; mode.max_num = 63; mode.mutability = Mutability::Ready;
    ParamSpec post;
    post.name = "post-messages"; post.nick = "Post Messages"; post.blurb = "Whether to post messages on the bus for each interval";
    post.type = PropType::Boolean; post.def_num = 1; post.min_num = 0; post.max_num = 1; post.mutability = Mutability::Playing;
    ParamSpec iv;
    iv.name = "interval"; iv.nick = "Interval"; iv.blurb = "Interval in nanoseconds for posting messages";
    iv.type = PropType::UInt64; iv.def_num = 1e9; iv.min_num = 0; iv.max_num = 18446744073709551614.0; iv.mutability = Mutability::Ready;
    return std::vector<ParamSpec>{mode, post, iv};
  }();
  return p;
}

bool EbuR128Level::store_number(const std::string &n, double v) {
  if (n == "mode") mode_ = (unsigned)v;
  else if (n == "post-messages") post_messages_ = v != 0.0;
  else return false;
  return true;
}
bool EbuR128Level::load_number(const std::string &n, double *v) const {
  if (n == "mode") *v = mode_;
  else if (n == "post-messages") *v = post_messages_ ? 1.0 : 0.0;
  else return false;
  return true;
}
bool EbuR128Level::store_u64(const std::string &n, uint64_t v) {
  if (n != "interval") return false;
  interval_ns_ = v;
  return true;
}
bool EbuR128Level::load_u64(const std::string &n, uint64_t *v) const {
  if (n != "interval") return false;
  *v = interval_ns_;
  return true;
}

bool EbuR128Level::setup(int rate, int channels, int sample_format, bool planar, const int *channel_class) {
  if (!ctx_) return false;
  unsigned mode;
  uint64_t interval;
  {
    std::lock_guard<std::mutex> g(settings_mutex_);
    mode = mode_;
    interval = interval_ns_;
  }
  std::vector<int> center(channels > 0 ? (size_t)channels : 0, 1);  // no positions: all weighted like Center
  if (mi355_ebur128_setup(ctx_, (unsigned)channels, (unsigned)rate, mode, channel_class ? channel_class : center.data()) != MI355_OK) {
    last_error_ = std::string("Failed to create EBU R128: ") + mi355_ctx_last_error(ctx_);
    return false;
  }
  rate_ = rate; channels_ = channels; format_ = sample_format; planar_ = planar; state_mode_ = mode;
  // interval.mul_div_floor(rate, SECOND) (imp.rs:597-601); 128-bit intermediate like gst_util_uint64_scale
  interval_frames_ = (uint64_t)(((unsigned __int128)interval * (unsigned __int128)rate) / 1000000000ull);
  interval_frames_remaining_ = interval_frames_;
  num_frames_ = 0;
  have_state_ = true;
  started_ = true;
  return true;
}

bool EbuR128Level::stop() {
  if (ctx_) mi355_ebur128_teardown(ctx_);
  have_state_ = false;
  started_ = false;
  queue_.clear();
  return true;
}

bool EbuR128Level::pop_message(EbuR128LevelMessage *out) {
  if (queue_.empty()) return false;
  *out = queue_.front();
  queue_.erase(queue_.begin());
  return true;
}

FlowReturn EbuR128Level::transform_ip_passthrough(const void *data, const void *const *planes, size_t frames, uint64_t pts_ns) {
  if (!ctx_) return FlowReturn::Error;
  bool post;
  {
    std::lock_guard<std::mutex> g(settings_mutex_);
    post = post_messages_;
  }
  if (!have_state_) {  // "Have no state yet" (imp.rs:302-306)
    last_error_ = "Have no state yet";
    return FlowReturn::NotNegotiated;
  }
  static const size_t kSampleBytes[4] = {2, 4, 4, 8};
  const size_t sb = kSampleBytes[format_];
  uint64_t timestamp = pts_ns;
  size_t done = 0;
  while (frames - done > 0) {
    if (reset_requested_) {  // imp.rs:320-333
      reset_requested_ = false;
      if (mi355_ebur128_reset(ctx_) != MI355_OK) return flow_from_status(MI355_ERR_HIP);
      interval_frames_remaining_ = interval_frames_;
      num_frames_ = 0;
    }
    const uint64_t left = frames - done;
    const uint64_t to_process = interval_frames_remaining_ < left ? interval_frames_remaining_ : left;
    int rc;
    if (planes) {
      std::vector<const void *> off((size_t)channels_);
      for (int c = 0; c < channels_; c++) off[(size_t)c] = (const char *)planes[c] + done * sb;
      rc = mi355_ebur128_add_frames_planar(ctx_, off.data(), (size_t)to_process, format_);
    } else {
      rc = mi355_ebur128_add_frames(ctx_, (const char *)data + done * sb * (size_t)channels_, (size_t)to_process, format_);
    }
    if (rc != MI355_OK) {
      last_error_ = std::string("Failed to process buffer: ") + mi355_ctx_last_error(ctx_);
      return FlowReturn::Error;
    }
    done += (size_t)to_process;
    interval_frames_remaining_ -= to_process;
    num_frames_ += to_process;
    timestamp += (uint64_t)(((unsigned __int128)to_process * 1000000000ull) / (unsigned __int128)rate_);  // mul_div_floor
    if (interval_frames_remaining_ == 0) {
      interval_frames_remaining_ = interval_frames_;
      if (post) {
        EbuR128LevelMessage m;
        m.timestamp = timestamp;
        m.fields = state_mode_;
        if (state_mode_ & MI355_EBUR128_MOMENTARY) mi355_ebur128_loudness_momentary(ctx_, &m.momentary_loudness);
        if (state_mode_ & MI355_EBUR128_SHORT_TERM) mi355_ebur128_loudness_shortterm(ctx_, &m.shortterm_loudness);
        if (state_mode_ & MI355_EBUR128_GLOBAL) {
          mi355_ebur128_loudness_global(ctx_, &m.global_loudness);
          mi355_ebur128_relative_threshold(ctx_, &m.relative_threshold);
        }
        if (state_mode_ & MI355_EBUR128_LOUDNESS_RANGE) mi355_ebur128_loudness_range(ctx_, &m.loudness_range);
        if (state_mode_ & MI355_EBUR128_SAMPLE_PEAK) {
          m.sample_peak.resize((size_t)channels_);
          for (int c = 0; c < channels_; c++) mi355_ebur128_sample_peak(ctx_, (unsigned)c, &m.sample_peak[(size_t)c]);
        }
        if (state_mode_ & MI355_EBUR128_TRUE_PEAK) {
          m.true_peak.resize((size_t)channels_);
          for (int c = 0; c < channels_; c++) mi355_ebur128_true_peak(ctx_, (unsigned)c, &m.true_peak[(size_t)c]);
        }
        queue_.push_back(std::move(m));
      }
    }
    if (interval_frames_ == 0) break;  // degenerate interval: avoid spinning (the reference would loop on 0-frame slices)
  }
  return FlowReturn::Ok;
}

// ------------------------------------------------------------------ HrtfRender

HrtfRender::HrtfRender(int device) : Element(device) {}

const ElementMetadata &HrtfRender::metadata() const {
  static const ElementMetadata m{"Head-Related Transfer Function (HRTF) renderer", "Filter/Effect/Audio",
                                 "Renders spatial sounds to a given position", "Tomasz Andrzejak <andreiltd@gmail.com>"};
  return m;
}

const std::vector<ParamSpec> &HrtfRender::properties() const {
  static const std::vector<ParamSpec> p = [] {
    auto u64 = [](const char *n, const char *nick, const char *blurb, double def) {
      ParamSpec s;
      s.name = n; s.nick = nick; s.blurb = blurb; s.type = PropType::UInt64;
      s.def_num = def; s.min_num = 0; s.max_num = 18446744073709551614.0; s.mutability = Mutability::Ready;
      return s;
    };
    ParamSpec file;
    file.name = "hrir-file"; file.nick = "Head Transform Impulse Response";
    file.blurb = "Head Transform Impulse Response file location to read from"; file.type = PropType::String; file.mutability = Mutability::Ready;
    ParamSpec rayon;
    rayon.name = "use-rayon"; rayon.nick = "Use Rayon"; rayon.blurb = "Use Rayon to process input channels in parallel";
    rayon.type = PropType::Boolean; rayon.def_num = 0; rayon.min_num = 0; rayon.max_num = 1; rayon.mutability = Mutability::Ready;
    // "hrir-raw" (GBytes) and "spatial-objects" (GstValueArray) are listed for introspection; they are set through
    // set_hrir_raw / set_spatial_objects (no scalar spelling)
    ParamSpec raw;
    raw.name = "hrir-raw"; raw.nick = "Head Transform Impulse Response"; raw.blurb = "Head Transform Impulse Response raw bytes";
    raw.type = PropType::String; raw.mutability = Mutability::Ready;
    ParamSpec objs;
    objs.name = "spatial-objects"; objs.nick = "Spatial Objects"; objs.blurb = "Spatial object Metadata to apply on input channels";
    objs.type = PropType::String; objs.mutability = Mutability::Playing;
    return std::vector<ParamSpec>{
        raw,
        objs,
        file,
        u64("interpolation-steps", "Interpolation Steps", "Interpolation Steps is the amount of slices to cut source to", 8),
        u64("block-length", "Block Length", "Block Length is the length of each slice", 512),
        rayon,
    };
  }();
  return p;
}

bool HrtfRender::store_number(const std::string &n, double v) {
  if (n != "use-rayon") return false;
  use_rayon_ = v != 0.0;  // kept for the property surface; channels are always processed in parallel on the device
  return true;
}
bool HrtfRender::load_number(const std::string &n, double *v) const {
  if (n != "use-rayon") return false;
  *v = use_rayon_ ? 1.0 : 0.0;
  return true;
}
bool HrtfRender::store_u64(const std::string &n, uint64_t v) {
  if (n == "interpolation-steps") interpolation_steps_ = v;
  else if (n == "block-length") block_length_ = v;
  else return false;
  return true;
}
bool HrtfRender::load_u64(const std::string &n, uint64_t *v) const {
  if (n == "interpolation-steps") *v = interpolation_steps_;
  else if (n == "block-length") *v = block_length_;
  else return false;
  return true;
}
bool HrtfRender::store_string(const std::string &n, const std::string &v) {
  if (n != "hrir-file") return false;
  hrir_file_ = v;
  have_hrir_file_ = true;
  return true;
}
bool HrtfRender::load_string(const std::string &n, std::string *v) const {
  if (n != "hrir-file") return false;
  *v = hrir_file_;
  return true;
}

void HrtfRender::set_hrir_raw(const void *bytes, size_t len) {
  std::lock_guard<std::mutex> g(settings_mutex_);
  hrir_raw_.assign((const unsigned char *)bytes, (const unsigned char *)bytes + len);
  have_hrir_raw_ = true;
}

bool HrtfRender::set_spatial_objects(const std::vector<SpatialObject> &objs) {
  std::lock_guard<std::mutex> g(settings_mutex_);
  if (have_state_ && (int)objs.size() != channels_) {  // imp.rs:440-451: warning, property unchanged
    last_error_ = "Could not update spatial objects, expected " + std::to_string(channels_) + " channels, got " + std::to_string(objs.size());
    return false;
  }
  objects_ = objs;
  have_objects_ = !objs.empty();  // an empty array unsets the property (imp.rs:453)
  return true;
}

std::vector<SpatialObject> HrtfRender::spatial_objects() const {
  std::lock_guard<std::mutex> g(settings_mutex_);
  return have_objects_ ? objects_ : std::vector<SpatialObject>();
}

// TryFrom<AudioChannelPosition> for SpatialObject (spatial.rs:177-222); GstAudioChannelPosition numbering
static bool object_from_position(int pos, SpatialObject *o) {
  struct P { int pos; float x, y, z; };
  static const P table[] = {
      {-2 /*Mono*/, 0.0f, 0.0f, 2.5f}, {0 /*FrontLeft*/, -1.45f, 0.0f, 2.5f}, {1 /*FrontRight*/, 1.45f, 0.0f, 2.5f},
      {2 /*FrontCenter*/, 0.0f, 0.0f, 2.5f}, {3 /*Lfe1*/, 0.0f, 0.0f, 0.0f}, {4 /*RearLeft*/, -1.45f, 0.0f, -2.5f},
      {5 /*RearRight*/, 1.45f, 0.0f, -2.5f}, {6 /*FrontLeftOfCenter*/, -0.72f, 0.0f, 2.5f}, {7 /*FrontRightOfCenter*/, 0.72f, 0.0f, 2.5f},
      {8 /*RearCenter*/, 0.0f, 0.0f, -2.5f}, {9 /*Lfe2*/, 0.0f, 0.0f, 0.0f}, {10 /*SideLeft*/, -2.5f, 0.0f, -0.44f},
      {11 /*SideRight*/, 2.5f, 0.0f, -0.44f}, {12 /*TopFrontLeft*/, -0.72f, 2.5f, 1.25f}, {13 /*TopFrontRight*/, 0.72f, 2.5f, 1.25f},
      {14 /*TopFrontCenter*/, 0.0f, 2.5f, 1.25f}, {15 /*TopCenter*/, 0.0f, 2.5f, 0.0f}, {16 /*TopRearLeft*/, -0.72f, 2.5f, -1.25f},
      {17 /*TopRearRight*/, 0.72f, 2.5f, -1.25f}, {18 /*TopSideLeft*/, -1.25f, 2.5f, -0.22f}, {19 /*TopSideRight*/, 1.25f, 2.5f, -0.22f},
      {20 /*TopRearCenter*/, 0.0f, 2.5f, -1.25f}, {21 /*BottomFrontCenter*/, 0.0f, -2.5f, 1.25f}, {22 /*BottomFrontLeft*/, -0.72f, -2.5f, 1.25f},
      {23 /*BottomFrontRight*/, 0.72f, -2.5f, 1.25f}, {24 /*WideLeft*/, -2.5f, 0.0f, 1.45f}, {25 /*WideRight*/, 2.5f, 0.0f, 1.45f},
      {26 /*SurroundLeft*/, -2.5f, 0.0f, -1.45f}, {27 /*SurroundRight*/, 2.5f, 0.0f, -1.45f}};
  for (const P &p : table)
    if (p.pos == pos) {
      o->coordinate_system = 1;  // Position::LeftHanded
      o->x = p.x; o->y = p.y; o->z = p.z;
      o->distance_gain = 1.0f;
      return true;
    }
  return false;  // FlowError::NotSupported -> "Unsupported channel position"
}

// the table above for the GStreamer shim: 0 and the left-handed position of a GstAudioChannelPosition, -1 if the reference
// answers FlowError::NotSupported for it
extern "C" int mi355host_hrtf_object_from_channel_position(int gst_audio_channel_position, float xyz_left_handed[3]) {
  SpatialObject o;
  if (!xyz_left_handed || !object_from_position(gst_audio_channel_position, &o)) return -1;
  xyz_left_handed[0] = o.x; xyz_left_handed[1] = o.y; xyz_left_handed[2] = o.z;
  return 0;
}

// Position::{to_cartesian, to_left_handed, to_right_handed}().to_vec3() (audio/hrtf/src/spatial.rs:40-70); systems numbered as
// GstHrtfCoordinateSystem: 0 Cartesian, 1 LeftHanded, 2 RightHanded. The known answers of spatial.rs:235-287 are replayed
// against this function in tests/test_oracle_hrtf.py.
extern "C" int mi355host_position_convert(int from, int to, const float in[3], float out[3]) {
  if (from < 0 || from > 2 || to < 0 || to > 2 || !in || !out) return -1;
  const float x = in[0], y = in[1], z = in[2];
  if (from == to) { out[0] = x; out[1] = y; out[2] = z; return 0; }
  if (to == 0) {         // to_cartesian (spatial.rs:40-48)
    if (from == 1) { out[0] = z; out[1] = -x; out[2] = y; }
    else { out[0] = -z; out[1] = -x; out[2] = y; }
  } else if (to == 1) {  // to_left_handed (spatial.rs:51-59)
    if (from == 0) { out[0] = -y; out[1] = z; out[2] = x; }
    else { out[0] = x; out[1] = y; out[2] = -z; }
  } else {               // to_right_handed (spatial.rs:62-70)
    if (from == 0) { out[0] = -y; out[1] = z; out[2] = -x; }
    else { out[0] = x; out[1] = y; out[2] = -z; }
  }
  return 0;
}

// Position::to_right_handed().to_vec3(): what HrtfRender hands the crate (hrtf/imp.rs:64-73)
static void to_right_handed(const SpatialObject &o, float out[3]) {
  const float in[3] = {o.x, o.y, o.z};
  (void)mi355host_position_convert(o.coordinate_system, 2, in, out);
}

bool HrtfRender::set_caps(int rate, int channels, const int *positions) {
  if (!ctx_) return false;
  std::lock_guard<std::mutex> g(settings_mutex_);
  have_state_ = false;
  if (channels < 1 || channels > 64 || rate <= 0) { last_error_ = "Failed to parse input caps"; return false; }
  if (!have_objects_) {
    if (!positions) { last_error_ = "Cannot infer object positions"; return false; }
    std::vector<SpatialObject> objs((size_t)channels);
    for (int c = 0; c < channels; c++)
      if (!object_from_position(positions[c], &objs[(size_t)c])) { last_error_ = "Unsupported channel position"; return false; }
    objects_ = objs;
    have_objects_ = true;
  }
  if ((int)objects_.size() != channels) { last_error_ = "Wrong number of spatial objects"; return false; }
  // Settings::sphere (imp.rs:84-94): raw bytes win over the file location
  std::vector<unsigned char> bytes;
  if (have_hrir_raw_) {
    bytes = hrir_raw_;
  } else if (have_hrir_file_) {
    FILE *f = std::fopen(hrir_file_.c_str(), "rb");
    if (!f) { last_error_ = "Failed to load sphere: cannot open " + hrir_file_; return false; }
    unsigned char buf[65536];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) bytes.insert(bytes.end(), buf, buf + n);
    std::fclose(f);
  } else {
    last_error_ = "Failed to load sphere: Impulse response not set";
    return false;
  }
  if (mi355_hrtf_load_sphere(ctx_, bytes.data(), bytes.size(), (uint32_t)rate) != MI355_OK) {
    last_error_ = std::string("Failed to load sphere: ") + mi355_ctx_last_error(ctx_);
    return false;
  }
  // block_samples = blklen.checked_mul(steps) (imp.rs:655-657)
  unsigned long long bs = 0;
  if (__builtin_umulll_overflow(block_length_, interpolation_steps_, &bs) || bs == 0 || bs > (1ull << 24)) {
    last_error_ = "Not enough memory for frame allocation";
    return false;
  }
  if (mi355_hrtf_setup(ctx_, channels, (int)block_length_, (int)interpolation_steps_) != MI355_OK) {
    last_error_ = mi355_ctx_last_error(ctx_);
    return false;
  }
  rate_ = rate; channels_ = channels; block_samples_ = (size_t)bs;
  adapter_.clear();
  have_state_ = true;
  return true;
}

size_t HrtfRender::transform_size(size_t in_bytes) const {
  if (!have_state_) return 0;
  const size_t inblk = block_samples_ * (size_t)channels_ * 4, outblk = block_samples_ * 8;
  return ((in_bytes + adapter_.size() * 4) / inblk) * outblk;
}

FlowReturn HrtfRender::process_available(std::vector<float> *out) {
  const size_t blk = block_samples_ * (size_t)channels_;
  std::vector<float> pos((size_t)channels_ * 3), gains((size_t)channels_);
  {
    std::lock_guard<std::mutex> g(settings_mutex_);  // settings snapshot for this buffer
    if (!have_objects_ || (int)objects_.size() != channels_) return FlowReturn::NotNegotiated;
    for (int c = 0; c < channels_; c++) {
      to_right_handed(objects_[(size_t)c], &pos[(size_t)c * 3]);
      gains[(size_t)c] = objects_[(size_t)c].distance_gain;
    }
  }
  size_t done = 0;
  while (adapter_.size() - done >= blk) {
    const size_t o = out->size();
    out->resize(o + block_samples_ * 2);
    const int rc = mi355_hrtf_process_block(ctx_, adapter_.data() + done, out->data() + o, pos.data(), gains.data());
    if (rc != MI355_OK) return flow_from_status(rc);
    done += blk;
  }
  adapter_.erase(adapter_.begin(), adapter_.begin() + (std::ptrdiff_t)done);
  return FlowReturn::Ok;
}

FlowReturn HrtfRender::transform(const float *in, size_t in_bytes, std::vector<float> *out) {
  if (!ctx_) return FlowReturn::Error;
  if (!have_state_) { last_error_ = "hrtfrender: not negotiated"; return FlowReturn::NotNegotiated; }
  out->clear();
  adapter_.insert(adapter_.end(), in, in + in_bytes / 4);
  return process_available(out);
}

FlowReturn HrtfRender::drain(std::vector<float> *out) {
  if (!ctx_) return FlowReturn::Error;
  if (!have_state_) return FlowReturn::NotNegotiated;
  out->clear();
  const size_t avail = adapter_.size();  // samples
  if (avail == 0) return FlowReturn::Ok;
  const size_t blk = block_samples_ * (size_t)channels_;
  const size_t out_frames = avail / (size_t)channels_;  // outputsz = avail / inbpf * outbpf (imp.rs:303)
  adapter_.resize(blk, 0.0f);                            // zero input of inblksz - avail bytes (imp.rs:305-318)
  const FlowReturn r = process_available(out);
  if (r != FlowReturn::Ok) return r;
  out->resize(out_frames * 2);                           // outbuf.set_size(outputsz)
  mi355_hrtf_reset(ctx_);                                // state.reset_processors()
  return FlowReturn::Ok;
}

void HrtfRender::flush_stop() {
  adapter_.clear();
  if (ctx_ && have_state_) mi355_hrtf_reset(ctx_);
}

bool HrtfRender::stop() {
  if (ctx_) mi355_hrtf_teardown(ctx_);
  have_state_ = false;
  adapter_.clear();
  started_ = false;
  return true;
}

// ------------------------------------------------------------------ AudioLoudNorm

AudioLoudNorm::AudioLoudNorm(int device) : Element(device) {}

const ElementMetadata &AudioLoudNorm::metadata() const {
  static const ElementMetadata m{"Audio loudness normalizer", "Filter/Effect/Audio", "Normalizes perceived loudness of an audio stream",
                                 "Sebastian Dröge <sebastian@centricular.com>"};
  return m;
}

const std::vector<ParamSpec> &AudioLoudNorm::properties() const {
  static const std::vector<ParamSpec> p = [] {
    auto dbl = [](const char *n, const char *nick, const char *blurb, double def, double lo, double hi) {
      ParamSpec s;
      s.name = n; s.nick = nick; s.blurb = blurb; s.type = PropType::Double;
      s.def_num = def; s.min_num = lo; s.max_num = hi; s.mutability = Mutability::Ready;
      return s;
    };
    return std::vector<ParamSpec>{
        dbl("loudness-target", "Loudness Target", "Loudness target in LUFS", -24.0, -70.0, -5.0),
        dbl("loudness-range-target", "Loudness Range Target", "Loudness range target in LU", 7.0, 1.0, 20.0),
        dbl("max-true-peak", "Maximum True Peak", "Maximum True Peak in dbTP", -2.0, -9.0, 0.0),
        dbl("offset", "Offset Gain", "Offset Gain in LU", 0.0, -99.0, 99.0),
    };
  }();
  return p;
}

bool AudioLoudNorm::store_number(const std::string &n, double v) {
  if (n == "loudness-target") loudness_target_ = v;
  else if (n == "loudness-range-target") loudness_range_target_ = v;
  else if (n == "max-true-peak") max_true_peak_ = v;
  else if (n == "offset") offset_ = v;
  else return false;
  return true;
}
bool AudioLoudNorm::load_number(const std::string &n, double *v) const {
  if (n == "loudness-target") *v = loudness_target_;
  else if (n == "loudness-range-target") *v = loudness_range_target_;
  else if (n == "max-true-peak") *v = max_true_peak_;
  else if (n == "offset") *v = offset_;
  else return false;
  return true;
}

bool AudioLoudNorm::set_caps(int rate, int channels) {
  if (!ctx_) return false;
  have_state_ = false;
  if (rate != 192000) { last_error_ = "audioloudnorm: caps must be F64 interleaved at 192000 Hz"; return false; }  // pad template (imp.rs:1848-1851)
  double t, r, p, o;
  {
    std::lock_guard<std::mutex> g(settings_mutex_);
    t = loudness_target_; r = loudness_range_target_; p = max_true_peak_; o = offset_;
  }
  if (mi355_loudnorm_setup(ctx_, (unsigned)channels, t, r, p, o) != MI355_OK) { last_error_ = mi355_ctx_last_error(ctx_); return false; }
  channels_ = channels;
  have_state_ = true;
  return true;
}

FlowReturn AudioLoudNorm::chain(const double *data, size_t frames, std::vector<double> *out) {
  if (!ctx_) return FlowReturn::Error;
  if (!have_state_) { last_error_ = "audioloudnorm: not negotiated"; return FlowReturn::NotNegotiated; }  // imp.rs:1556-1563
  const size_t cap = (frames / 19200 + 32) * 19200;
  out->assign(cap * (size_t)channels_, 0.0);
  size_t n = 0;
  const int rc = mi355_loudnorm_push(ctx_, data, frames, out->data(), cap, &n);
  out->resize(n * (size_t)channels_);
  return flow_from_status(rc);
}

FlowReturn AudioLoudNorm::drain(std::vector<double> *out) {
  if (!ctx_) return FlowReturn::Error;
  out->clear();
  if (!have_state_) return FlowReturn::Ok;  // no state: nothing to drain (imp.rs:1661-1664)
  const size_t cap = 31 * 19200 + 3 * 192000;
  out->assign(cap * (size_t)channels_, 0.0);
  size_t n = 0;
  int eos = 0;
  const int rc = mi355_loudnorm_drain(ctx_, out->data(), cap, &n, &eos);
  out->resize(n * (size_t)channels_);
  if (rc != MI355_OK) return flow_from_status(rc);
  return eos ? FlowReturn::Eos : FlowReturn::Ok;
}

bool AudioLoudNorm::stop() {
  if (ctx_) mi355_loudnorm_teardown(ctx_);
  have_state_ = false;
  started_ = false;
  return true;
}

// ------------------------------------------------------------------ RoundedCorners

RoundedCorners::RoundedCorners(int device) : Element(device) {}
RoundedCorners::~RoundedCorners() {
  if (cairo_) dlclose(cairo_);
}

const ElementMetadata &RoundedCorners::metadata() const {
  static const ElementMetadata m{"Rounded Corners", "Filter/Effect/Converter/Video", "Adds rounded corners to video",
                                 "Sanchayan Maity <sanchayan@asymptotic.io>"};
  return m;
}

const std::vector<ParamSpec> &RoundedCorners::properties() const {
  static const std::vector<ParamSpec> p = [] {
    ParamSpec s;
    s.name = "border-radius-px"; s.nick = "Border radius in pixels"; s.blurb = "Draw rounded corners with given border radius";
    s.type = PropType::UInt64;  // guint in the reference
    s.def_num = 0; s.min_num = 0; s.max_num = 4294967295.0; s.mutability = Mutability::Playing;
    return std::vector<ParamSpec>{s};
  }();
  return p;
}

bool RoundedCorners::store_u64(const std::string &n, uint64_t v) {
  if (n != "border-radius-px") return false;
  if (border_radius_px_ != (uint32_t)v) {  // imp.rs:296-308: remember the change, renegotiate
    changed_ = true;
    border_radius_px_ = (uint32_t)v;
  }
  return true;
}
bool RoundedCorners::load_u64(const std::string &n, uint64_t *v) const {
  if (n != "border-radius-px") return false;
  *v = border_radius_px_;
  return true;
}

std::vector<int> RoundedCorners::transform_caps_to_src() const {
  std::lock_guard<std::mutex> g(settings_mutex_);
  if (border_radius_px_ == 0) return {100, 101};
  return {101};
}

bool RoundedCorners::set_caps(int width, int height, int out_format) {
  std::lock_guard<std::mutex> g(settings_mutex_);
  have_state_ = false;
  if (width <= 0 || height <= 0 || (out_format != 100 && out_format != 101)) { last_error_ = "Failed to parse output caps"; return false; }
  if (out_format == 100) { passthrough_ = true; return true; }
  passthrough_ = false;
  width_ = width; height_ = height;
  alpha_stride_ = (width + 3) & ~3;                         // GstVideoInfo: A420 plane 3 stride = GST_ROUND_UP_4(width)
  const size_t ru2_height = ((size_t)height + 1) & ~(size_t)1;
  alpha_mem_.assign((size_t)alpha_stride_ * ru2_height, 0);  // alpha_mem_size (imp.rs:469-470)
  have_state_ = true;
  changed_ = true;
  return true;
}

// generate_alpha_mask + draw_rounded_corners (imp.rs:57-180) with the C API of the same library the cairo crate binds.
// `alpha`: the A8 plane (stride x round_up_2(height) bytes). One implementation for the C++ element and for the C entry
// point the GStreamer shim calls (mi355host_rounded_corners_mask).
static bool draw_rounded_mask(void **cairo_handle, uint8_t *alpha, size_t alpha_bytes, int width_, int height_, int alpha_stride_, uint32_t radius,
                              std::string &last_error_) {
  if (radius == 0) { std::fill(alpha, alpha + alpha_bytes, 0xff); return true; }
  std::fill(alpha, alpha + alpha_bytes, 0);
  void *&cairo_ = *cairo_handle;
  if (!cairo_) cairo_ = dlopen("libcairo.so.2", RTLD_NOW | RTLD_LOCAL);
  if (!cairo_) { last_error_ = "Failed to create cairo image surface: libcairo.so.2 not found"; return false; }
  typedef void *(*surf_create_t)(unsigned char *, int, int, int, int);
  typedef void *(*create_t)(void *);
  typedef void (*v_t)(void *);
  typedef void (*arc_t)(void *, double, double, double, double, double);
  typedef void (*rgb_t)(void *, double, double, double);
  typedef void (*rgba_t)(void *, double, double, double, double);
  typedef void (*lw_t)(void *, double);
  typedef int (*status_t)(void *);
  auto sym = [&](const char *n) { return dlsym(cairo_, n); };
  auto surf_create = (surf_create_t)sym("cairo_image_surface_create_for_data");
  auto create = (create_t)sym("cairo_create");
  auto new_sub_path = (v_t)sym("cairo_new_sub_path"), close_path = (v_t)sym("cairo_close_path"), fill_preserve = (v_t)sym("cairo_fill_preserve");
  auto stroke = (v_t)sym("cairo_stroke"), destroy = (v_t)sym("cairo_destroy"), surf_flush = (v_t)sym("cairo_surface_flush"), surf_destroy = (v_t)sym("cairo_surface_destroy");
  auto arc = (arc_t)sym("cairo_arc");
  auto set_rgb = (rgb_t)sym("cairo_set_source_rgb");
  auto set_rgba = (rgba_t)sym("cairo_set_source_rgba");
  auto set_lw = (lw_t)sym("cairo_set_line_width");
  auto surf_status = (status_t)sym("cairo_surface_status"), cr_status = (status_t)sym("cairo_status");
  if (!surf_create || !create || !new_sub_path || !close_path || !fill_preserve || !stroke || !destroy || !surf_flush || !surf_destroy || !arc || !set_rgb ||
      !set_rgba || !set_lw || !surf_status || !cr_status) { last_error_ = "Failed to create cairo image surface: symbols missing"; return false; }
  void *surface = surf_create(alpha, 2 /* CAIRO_FORMAT_A8 */, width_, height_, alpha_stride_);
  if (!surface || surf_status(surface) != 0) { last_error_ = "Failed to create cairo image surface"; if (surface) surf_destroy(surface); return false; }
  void *cr = create(surface);
  const double r = (double)radius, w = (double)width_, h = (double)height_, deg = 3.14159265358979323846 / 180.0;
  new_sub_path(cr);
  arc(cr, w - r, r, r, -90.0 * deg, 0.0 * deg);
  arc(cr, w - r, h - r, r, 0.0 * deg, 90.0 * deg);
  arc(cr, r, h - r, r, 90.0 * deg, 180.0 * deg);
  arc(cr, r, r, r, 180.0 * deg, 270.0 * deg);
  close_path(cr);
  set_rgb(cr, 0.0, 0.0, 0.0);
  fill_preserve(cr);
  set_rgba(cr, 0.0, 0.0, 0.0, 1.0);
  set_lw(cr, 1.0);
  stroke(cr);
  const bool ok = cr_status(cr) == 0;
  destroy(cr);
  surf_flush(surface);
  surf_destroy(surface);
  if (!ok) last_error_ = "Failed to draw rounded corners";
  return ok;
}

bool RoundedCorners::generate_alpha_mask(uint32_t radius) {
  device_mask_stale_ = true;
  return draw_rounded_mask(&cairo_, alpha_mem_.data(), alpha_mem_.size(), width_, height_, alpha_stride_, radius, last_error_);
}

extern "C" int mi355host_rounded_corners_mask(uint8_t *alpha, int width, int height, int stride, unsigned radius, char *err, size_t errlen) {
  static void *cairo = nullptr;  // (dlopen is reference counted and thread safe; a duplicate handle is harmless)
  std::string e;
  if (!alpha || width <= 0 || height <= 0 || stride < width) e = "bad argument";
  else if (draw_rounded_mask(&cairo, alpha, (size_t)stride * (((size_t)height + 1) & ~(size_t)1), width, height, stride, radius, e)) return 0;
  if (err && errlen) { std::snprintf(err, errlen, "%s", e.c_str()); }
  return -1;
}

FlowReturn RoundedCorners::prepare_output_buffer(const uint8_t **alpha, size_t *size, int *stride) {
  std::lock_guard<std::mutex> g(settings_mutex_);
  if (passthrough_) { if (alpha) *alpha = nullptr; if (size) *size = 0; if (stride) *stride = 0; return FlowReturn::Ok; }
  if (!have_state_) { last_error_ = "roundedcorners: not negotiated"; return FlowReturn::NotNegotiated; }
  if (changed_) {  // imp.rs:490-497
    if (!generate_alpha_mask(border_radius_px_)) return FlowReturn::NotNegotiated;
    changed_ = false;
  }
  if (alpha) *alpha = alpha_mem_.data();
  if (size) *size = alpha_mem_.size();
  if (stride) *stride = alpha_stride_;
  return FlowReturn::Ok;
}

FlowReturn RoundedCorners::prepare_output_buffer_device(uint8_t *d_frames, size_t frame_pitch, size_t alpha_offset, int n_frames) {
  const uint8_t *alpha = nullptr;
  size_t size = 0;
  int stride = 0;
  const FlowReturn r = prepare_output_buffer(&alpha, &size, &stride);  // regenerates the plane if the radius changed (imp.rs:490-497)
  if (r != FlowReturn::Ok || passthrough_) return r;
  std::lock_guard<std::mutex> g(settings_mutex_);
  if (!ctx_) { last_error_ = "roundedcorners: no device context"; return FlowReturn::Error; }
  if (device_mask_stale_) {
    const int rc = mi355_roundedcorners_set_mask(ctx_, alpha_mem_.data(), width_, height_, alpha_stride_);
    if (rc) return flow_from_status(rc);
    device_mask_stale_ = false;
  }
  return flow_from_status(mi355_roundedcorners_append_device(ctx_, d_frames, frame_pitch, alpha_offset, n_frames));
}

bool RoundedCorners::stop() {
  std::lock_guard<std::mutex> g(settings_mutex_);
  have_state_ = false;
  device_mask_stale_ = true;
  if (ctx_) (void)mi355_roundedcorners_set_mask(ctx_, nullptr, 0, 0, 0);
  alpha_mem_.clear();
  started_ = false;
  return true;
}

// ------------------------------------------------------------------ VideoCompare

VideoCompare::VideoCompare(int device) : Element(device) {}

const ElementMetadata &VideoCompare::metadata() const {
  static const ElementMetadata m{"Image comparison", "Filter/Video", "Compare similarity of video frames",
                                 "Rafael Caricio <rafael@caricio.com>"};
  return m;
}

static const char *const kHashAlgoNicks[] = {"mean", "gradient", "vertgradient", "doublegradient", "blockhash", "dssim"};

const std::vector<ParamSpec> &VideoCompare::properties() const {
  static const std::vector<ParamSpec> p = [] {
    ParamSpec algo;
    algo.name = "hash-algo"; algo.nick = "Hashing Algorithm"; algo.blurb = "Which hashing algorithm to use for image comparisons";
    algo.type = PropType::String; algo.def_num = MI355_HASH_BLOCKHASH; algo.mutability = Mutability::Ready;  // GEnum, set by nick
    ParamSpec thr;
    thr.name = "max-dist-threshold"; thr.nick = "Maximum Distance Threshold";
    thr.blurb = "Maximum distance threshold to emit messages when an image is detected, by default emits only on exact match";
    thr.type = PropType::Double; thr.def_num = 0.0; thr.min_num = 0.0; thr.max_num = DBL_MAX; thr.mutability = Mutability::Ready;
    return std::vector<ParamSpec>{algo, thr};
  }();
  return p;
}

bool VideoCompare::store_number(const std::string &n, double v) {
  if (n != "max-dist-threshold") return false;
  max_dist_threshold_ = v;
  return true;
}
bool VideoCompare::load_number(const std::string &n, double *v) const {
  if (n != "max-dist-threshold") return false;
  *v = max_dist_threshold_;
  return true;
}
bool VideoCompare::store_string(const std::string &n, const std::string &v) {
  if (n != "hash-algo") return false;
  for (int i = 0; i < 6; i++)
    if (v == kHashAlgoNicks[i]) { hash_algo_ = i; return true; }  // state.hasher = hash_algo.into() (imp.rs:112)
  last_error_ = "hash-algo: unknown nick " + v;
  return false;
}
bool VideoCompare::load_string(const std::string &n, std::string *v) const {
  if (n != "hash-algo") return false;
  *v = kHashAlgoNicks[hash_algo_];
  return true;
}

FlowReturn VideoCompare::aggregate_frames(const std::vector<VideoFrame> &frames, bool have_running_time, uint64_t running_time,
                                          VideoCompareMessage *msg, bool *posted) {
  *posted = false;
  if (!ctx_) return FlowReturn::Error;
  if (frames.empty()) { last_error_ = "No reference sink pad exists"; return FlowReturn::Eos; }  // imp.rs:275-278
  int algo;
  double threshold;
  {
    std::lock_guard<std::mutex> g(settings_mutex_);
    algo = hash_algo_; threshold = max_dist_threshold_;
  }
  const VideoFrame &ref = frames[0];
  if (!ref.data) return FlowReturn::Ok;  // reference pad has not produced a buffer: nothing to compare (imp.rs:283-296)
  VideoCompareMessage m;
  m.have_running_time = have_running_time;
  m.running_time = running_time;
  if (algo == MI355_HASH_DSSIM) {  // HasherEngine::DssimHasher (hashed_image.rs:41-53,66-70)
    mi355_dssim_image *ref_img = nullptr;
    int rc = mi355_dssim_create_image(ctx_, ref.data, ref.stride, ref.width, ref.height, ref.format, &ref_img);
    if (rc != MI355_OK) return flow_from_status(rc);
    // every other pad's frame is hashed and compared in one pass (mi355_dssim_compare_frames): one call for the whole
    // aggregate when the pads share stride and format, one call per pad otherwise
    bool uniform = true;
    for (size_t k = 1; k < frames.size(); k++) {
      const VideoFrame &f = frames[k];
      if (!f.data) { mi355_dssim_free_image(ctx_, ref_img); return FlowReturn::Ok; }
      if (f.width != ref.width || f.height != ref.height) {
        mi355_dssim_free_image(ctx_, ref_img);
        last_error_ = "Video streams do not have the same sizes (add videoscale and force the sizes to be equal on all sink pads)";
        return FlowReturn::NotNegotiated;
      }
      uniform = uniform && f.stride == frames[1].stride && f.format == frames[1].format;
    }
    const size_t n_other = frames.size() - 1;
    std::vector<const uint8_t *> ptrs(n_other);
    std::vector<double> dist(n_other, 0.0);
    for (size_t k = 0; k < n_other; k++) ptrs[k] = frames[k + 1].data;
    if (uniform && n_other > 0 && n_other <= 64) {
      rc = mi355_dssim_compare_frames(ctx_, ref_img, ptrs.data(), (int)n_other, frames[1].stride, ref.width, ref.height, frames[1].format, dist.data());
    } else {
      for (size_t k = 0; k < n_other && rc == MI355_OK; k++)
        rc = mi355_dssim_compare_frames(ctx_, ref_img, &ptrs[k], 1, frames[k + 1].stride, ref.width, ref.height, frames[k + 1].format, &dist[k]);
    }
    if (rc != MI355_OK) { mi355_dssim_free_image(ctx_, ref_img); return flow_from_status(rc); }
    for (size_t k = 0; k < n_other; k++) {
      VideoCompareMessage::PadDistance pd;
      pd.pad = "sink_" + std::to_string(k + 1);
      pd.distance = dist[k];
      m.pad_distances.push_back(pd);
    }
    mi355_dssim_free_image(ctx_, ref_img);
    bool any_d = false;
    for (const auto &pd : m.pad_distances) any_d = any_d || pd.distance <= threshold;
    if (any_d) { *msg = m; *posted = true; }
    return FlowReturn::Ok;
  }
  uint64_t ref_hash = 0;
  int rc = mi355_videocompare_hash_frame(ctx_, ref.data, ref.stride, ref.width, ref.height, ref.format, algo, &ref_hash);
  if (rc != MI355_OK) return flow_from_status(rc);
  for (size_t k = 1; k < frames.size(); k++) {
    const VideoFrame &f = frames[k];
    if (!f.data) return FlowReturn::Ok;  // pad without a prepared frame: skip this round (imp.rs:331-334)
    if (f.width != ref.width || f.height != ref.height) {  // imp.rs:337-347
      last_error_ = "Video streams do not have the same sizes (add videoscale and force the sizes to be equal on all sink pads)";
      return FlowReturn::NotNegotiated;
    }
    uint64_t h = 0;
    rc = mi355_videocompare_hash_frame(ctx_, f.data, f.stride, f.width, f.height, f.format, algo, &h);
    if (rc != MI355_OK) return flow_from_status(rc);
    VideoCompareMessage::PadDistance pd;
    pd.pad = "sink_" + std::to_string(k);
    pd.distance = mi355_videocompare_distance(algo, ref_hash, h);
    m.pad_distances.push_back(pd);
  }
  bool any = false;
  for (const auto &pd : m.pad_distances) any = any || pd.distance <= threshold;  // imp.rs:361-364
  if (any) { *msg = m; *posted = true; }
  return FlowReturn::Ok;
}

// ------------------------------------------------------------------ registry

std::vector<std::string> registered_factories() { return {"hsvfilter", "hsvdetector", "colorlut", "rsaudioecho", "ebur128level", "hrtfrender", "videocompare", "audioloudnorm", "roundedcorners"}; }

std::unique_ptr<Element> element_factory_make(const std::string &factory, int device, std::string *error) {
  std::unique_ptr<Element> e;
  if (factory == "hsvfilter") e.reset(new HsvFilter(device));
  else if (factory == "hsvdetector") e.reset(new HsvDetector(device));
  else if (factory == "colorlut") e.reset(new ColorLut(device));
  else if (factory == "rsaudioecho") e.reset(new AudioEcho(device));
  else if (factory == "ebur128level") e.reset(new EbuR128Level(device));
  else if (factory == "hrtfrender") e.reset(new HrtfRender(device));
  else if (factory == "videocompare") e.reset(new VideoCompare(device));
  else if (factory == "audioloudnorm") e.reset(new AudioLoudNorm(device));
  else if (factory == "roundedcorners") e.reset(new RoundedCorners(device));
  else {
    if (error) *error = "no such element factory: " + factory;
    return nullptr;
  }
  if (!e->last_error().empty()) {  // context creation failed: no device, no element (no CPU fallback)
    if (error) *error = e->last_error();
    return nullptr;
  }
  return e;
}

}  // namespace mi355host

// ------------------------------------------------------------------ flat C API over the element objects
// (what the Python tests and the compile-gated GStreamer shim call)
using namespace mi355host;

struct mi355el { std::unique_ptr<Element> e; std::string err; };

extern "C" {

mi355el *mi355el_factory_make(const char *factory, int device, char *err, size_t errlen) {
  std::string msg;
  auto e = element_factory_make(factory ? factory : "", device, &msg);
  if (!e) {
    if (err && errlen) { std::strncpy(err, msg.c_str(), errlen - 1); err[errlen - 1] = 0; }
    return nullptr;
  }
  auto *h = new mi355el();
  h->e = std::move(e);
  return h;
}
void mi355el_free(mi355el *h) { delete h; }
const char *mi355el_last_error(const mi355el *h) { return h ? h->e->last_error().c_str() : "null element"; }
const char *mi355el_type_name(const mi355el *h) { return h->e->type_name(); }
const char *mi355el_factory_name(const mi355el *h) { return h->e->factory_name(); }
const char *mi355el_klass(const mi355el *h) { return h->e->metadata().klass.c_str(); }
const char *mi355el_long_name(const mi355el *h) { return h->e->metadata().long_name.c_str(); }
int mi355el_n_properties(const mi355el *h) { return (int)h->e->properties().size(); }
const char *mi355el_property_name(const mi355el *h, int i) { return h->e->properties().at((size_t)i).name.c_str(); }
int mi355el_property_info(const mi355el *h, int i, int *type, double *def, double *lo, double *hi, int *mutable_playing) {
  const ParamSpec &p = h->e->properties().at((size_t)i);
  *type = (int)p.type; *def = p.def_num; *lo = p.min_num; *hi = p.max_num; *mutable_playing = p.mutability == Mutability::Playing;
  return 0;
}
int mi355el_n_formats(const mi355el *h, int src) { return (int)(src ? h->e->src_formats() : h->e->sink_formats()).size(); }
int mi355el_format(const mi355el *h, int src, int i) { return (src ? h->e->src_formats() : h->e->sink_formats()).at((size_t)i); }
int mi355el_set_double(mi355el *h, const char *name, double v) { return h->e->set_property(name, v) ? 0 : -1; }
int mi355el_get_double(const mi355el *h, const char *name, double *v) { return h->e->get_property(name, v) ? 0 : -1; }
int mi355el_set_u64(mi355el *h, const char *name, uint64_t v) { return h->e->set_property_u64(name, v) ? 0 : -1; }
int mi355el_get_u64(const mi355el *h, const char *name, uint64_t *v) { return h->e->get_property_u64(name, v) ? 0 : -1; }
int mi355el_set_string(mi355el *h, const char *name, const char *v) { return h->e->set_property(name, std::string(v ? v : "")) ? 0 : -1; }
int mi355el_start(mi355el *h) { return h->e->start() ? 0 : -1; }
int mi355el_stop(mi355el *h) { return h->e->stop() ? 0 : -1; }

int mi355el_transform_frame_ip(mi355el *h, int format, int width, int height, int stride, uint8_t *data, size_t size) {
  auto *f = dynamic_cast<HsvFilter *>(h->e.get());
  if (!f) return (int)FlowReturn::Error;
  VideoFrame fr;
  fr.format = format; fr.width = width; fr.height = height; fr.stride = stride; fr.data = data; fr.size = size;
  return (int)f->transform_frame_ip(fr);
}

int mi355el_transform_frame(mi355el *h, int in_format, int width, int height, int in_stride, const uint8_t *in, size_t in_size,
                            int out_format, int out_stride, uint8_t *out, size_t out_size) {
  VideoFrame a, b;
  a.format = in_format; a.width = width; a.height = height; a.stride = in_stride; a.data = const_cast<uint8_t *>(in); a.size = in_size;
  b.format = out_format; b.width = width; b.height = height; b.stride = out_stride; b.data = out; b.size = out_size;
  if (auto *c = dynamic_cast<ColorLut *>(h->e.get())) return (int)c->transform_frame(a, b);
  if (auto *d = dynamic_cast<HsvDetector *>(h->e.get())) return (int)d->transform_frame(a, b);
  return (int)FlowReturn::Error;
}

int mi355el_audio_setup(mi355el *h, int rate, int channels, int f64) {
  auto *e = dynamic_cast<AudioEcho *>(h->e.get());
  if (!e) return -1;
  AudioInfo info;
  info.rate = rate; info.channels = channels; info.f64 = f64 != 0;
  return e->setup(info) ? 0 : -1;
}
int mi355el_audio_transform_ip(mi355el *h, void *data, size_t nbytes) {
  auto *e = dynamic_cast<AudioEcho *>(h->e.get());
  if (!e) return (int)FlowReturn::Error;
  return (int)e->transform_ip(data, nbytes);
}


int mi355el_ebur128_setup(mi355el *h, int rate, int channels, int sample_format, int planar, const int *channel_class) {
  auto *e = dynamic_cast<EbuR128Level *>(h->e.get());
  if (!e) return -1;
  return e->setup(rate, channels, sample_format, planar != 0, channel_class) ? 0 : -1;
}
int mi355el_ebur128_push(mi355el *h, const void *data, const void *const *planes, size_t frames, uint64_t pts_ns) {
  auto *e = dynamic_cast<EbuR128Level *>(h->e.get());
  if (!e) return (int)FlowReturn::Error;
  return (int)e->transform_ip_passthrough(data, planes, frames, pts_ns);
}
void mi355el_ebur128_reset_signal(mi355el *h) {
  if (auto *e = dynamic_cast<EbuR128Level *>(h->e.get())) e->reset_signal();
}
// returns 1 and fills the outputs when a message was queued, else 0. peaks: up to `max_ch` per array.
int mi355el_ebur128_pop_message(mi355el *h, uint64_t *timestamp, unsigned *fields, double scalars[5], double *sample_peak, double *true_peak,
                                int max_ch, int *n_ch) {
  auto *e = dynamic_cast<EbuR128Level *>(h->e.get());
  EbuR128LevelMessage m;
  if (!e || !e->pop_message(&m)) return 0;
  *timestamp = m.timestamp;
  *fields = m.fields;
  scalars[0] = m.momentary_loudness; scalars[1] = m.shortterm_loudness; scalars[2] = m.global_loudness;
  scalars[3] = m.relative_threshold; scalars[4] = m.loudness_range;
  *n_ch = (int)(m.sample_peak.size() > m.true_peak.size() ? m.sample_peak.size() : m.true_peak.size());
  for (int c = 0; c < max_ch && c < (int)m.sample_peak.size(); c++) sample_peak[c] = m.sample_peak[(size_t)c];
  for (int c = 0; c < max_ch && c < (int)m.true_peak.size(); c++) true_peak[c] = m.true_peak[(size_t)c];
  return 1;
}


// ---- hrtfrender
static HrtfRender *as_hrtf(mi355el *h) { return h ? dynamic_cast<HrtfRender *>(h->e.get()) : nullptr; }

int mi355el_hrtf_set_hrir_raw(mi355el *h, const void *bytes, size_t len) {
  auto *e = as_hrtf(h);
  if (!e) return -1;
  e->set_hrir_raw(bytes, len);
  return 0;
}
// objects: n x {x, y, z, distance_gain}; coordinate_system[n] (GstHrtfCoordinateSystem)
int mi355el_hrtf_set_objects(mi355el *h, int n, const float *xyzg, const int *coordinate_system) {
  auto *e = as_hrtf(h);
  if (!e || n < 0) return -1;
  std::vector<SpatialObject> objs((size_t)n);
  for (int i = 0; i < n; i++) {
    objs[(size_t)i].x = xyzg[4 * i]; objs[(size_t)i].y = xyzg[4 * i + 1]; objs[(size_t)i].z = xyzg[4 * i + 2];
    objs[(size_t)i].distance_gain = xyzg[4 * i + 3];
    objs[(size_t)i].coordinate_system = coordinate_system ? coordinate_system[i] : 1;
  }
  return e->set_spatial_objects(objs) ? 0 : -1;
}
int mi355el_hrtf_get_objects(mi355el *h, int max, float *xyzg, int *coordinate_system) {
  auto *e = as_hrtf(h);
  if (!e) return -1;
  const auto objs = e->spatial_objects();
  for (int i = 0; i < (int)objs.size() && i < max; i++) {
    xyzg[4 * i] = objs[(size_t)i].x; xyzg[4 * i + 1] = objs[(size_t)i].y; xyzg[4 * i + 2] = objs[(size_t)i].z;
    xyzg[4 * i + 3] = objs[(size_t)i].distance_gain;
    if (coordinate_system) coordinate_system[i] = objs[(size_t)i].coordinate_system;
  }
  return (int)objs.size();
}
int mi355el_hrtf_set_caps(mi355el *h, int rate, int channels, const int *positions) {
  auto *e = as_hrtf(h);
  if (!e) return -1;
  return e->set_caps(rate, channels, positions) ? 0 : -1;
}
size_t mi355el_hrtf_transform_size(mi355el *h, size_t in_bytes) {
  auto *e = as_hrtf(h);
  return e ? e->transform_size(in_bytes) : 0;
}
// returns the GstFlowReturn value; *out_bytes = bytes written to `out` (never more than out_capacity)
int mi355el_hrtf_transform(mi355el *h, const float *in, size_t in_bytes, float *out, size_t out_capacity, size_t *out_bytes) {
  auto *e = as_hrtf(h);
  if (!e) return (int)FlowReturn::Error;
  std::vector<float> o;
  const FlowReturn r = e->transform(in, in_bytes, &o);
  const size_t n = o.size() * 4 < out_capacity ? o.size() * 4 : out_capacity;
  if (n) std::memcpy(out, o.data(), n);
  if (out_bytes) *out_bytes = o.size() * 4;
  return (int)r;
}
int mi355el_hrtf_drain(mi355el *h, float *out, size_t out_capacity, size_t *out_bytes) {
  auto *e = as_hrtf(h);
  if (!e) return (int)FlowReturn::Error;
  std::vector<float> o;
  const FlowReturn r = e->drain(&o);
  const size_t n = o.size() * 4 < out_capacity ? o.size() * 4 : out_capacity;
  if (n) std::memcpy(out, o.data(), n);
  if (out_bytes) *out_bytes = o.size() * 4;
  return (int)r;
}
void mi355el_hrtf_flush_stop(mi355el *h) {
  auto *e = as_hrtf(h);
  if (e) e->flush_stop();
}


// ---- videocompare: n frames of identical layout (frame 0 = reference pad). Returns the GstFlowReturn value;
// *posted = 1 when a message would be posted, distances[k-1] = distance of pad sink_k.
int mi355el_videocompare_aggregate(mi355el *h, int n_frames, const uint8_t *const *frames, int format, int width, int height,
                                   int stride, uint64_t running_time, int *posted, double *distances, int max_distances) {
  auto *e = h ? dynamic_cast<VideoCompare *>(h->e.get()) : nullptr;
  if (!e) return (int)FlowReturn::Error;
  std::vector<VideoFrame> fs((size_t)(n_frames > 0 ? n_frames : 0));
  for (int k = 0; k < n_frames; k++) {
    fs[(size_t)k].format = format; fs[(size_t)k].width = width; fs[(size_t)k].height = height; fs[(size_t)k].stride = stride;
    fs[(size_t)k].data = const_cast<uint8_t *>(frames[k]);
    fs[(size_t)k].size = (size_t)stride * (size_t)height;
  }
  VideoCompareMessage msg;
  bool p = false;
  const FlowReturn r = e->aggregate_frames(fs, true, running_time, &msg, &p);
  if (posted) *posted = p ? 1 : 0;
  for (int k = 0; p && k < (int)msg.pad_distances.size() && k < max_distances; k++) distances[k] = msg.pad_distances[(size_t)k].distance;
  return (int)r;
}


// ---- audioloudnorm
int mi355el_loudnorm_set_caps(mi355el *h, int rate, int channels) {
  auto *e = h ? dynamic_cast<AudioLoudNorm *>(h->e.get()) : nullptr;
  if (!e) return -1;
  return e->set_caps(rate, channels) ? 0 : -1;
}
// returns the GstFlowReturn value; *out_frames frames written (never more than out_capacity_frames)
int mi355el_loudnorm_chain(mi355el *h, const double *data, size_t frames, int channels, double *out, size_t out_capacity_frames, size_t *out_frames) {
  auto *e = h ? dynamic_cast<AudioLoudNorm *>(h->e.get()) : nullptr;
  if (!e) return (int)FlowReturn::Error;
  std::vector<double> o;
  const FlowReturn r = e->chain(data, frames, &o);
  const size_t n = channels > 0 ? o.size() / (size_t)channels : 0;
  const size_t m = n < out_capacity_frames ? n : out_capacity_frames;
  if (m) std::memcpy(out, o.data(), m * (size_t)channels * sizeof(double));
  if (out_frames) *out_frames = n;
  return (int)r;
}
int mi355el_loudnorm_drain(mi355el *h, int channels, double *out, size_t out_capacity_frames, size_t *out_frames) {
  auto *e = h ? dynamic_cast<AudioLoudNorm *>(h->e.get()) : nullptr;
  if (!e) return (int)FlowReturn::Error;
  std::vector<double> o;
  const FlowReturn r = e->drain(&o);
  const size_t n = channels > 0 ? o.size() / (size_t)channels : 0;
  const size_t m = n < out_capacity_frames ? n : out_capacity_frames;
  if (m) std::memcpy(out, o.data(), m * (size_t)channels * sizeof(double));
  if (out_frames) *out_frames = n;
  return (int)r;
}


// ---- roundedcorners
int mi355el_roundedcorners_set_caps(mi355el *h, int width, int height, int a420) {
  auto *e = h ? dynamic_cast<RoundedCorners *>(h->e.get()) : nullptr;
  if (!e) return -1;
  return e->set_caps(width, height, a420 ? 101 : 100) ? 0 : -1;
}
// copies the alpha plane (if any) into `out` (capacity bytes); *size / *stride describe it; returns the flow value
int mi355el_roundedcorners_prepare(mi355el *h, uint8_t *out, size_t capacity, size_t *size, int *stride, int *passthrough) {
  auto *e = h ? dynamic_cast<RoundedCorners *>(h->e.get()) : nullptr;
  if (!e) return (int)FlowReturn::Error;
  const uint8_t *a = nullptr;
  size_t n = 0;
  int st = 0;
  const FlowReturn r = e->prepare_output_buffer(&a, &n, &st);
  if (passthrough) *passthrough = e->passthrough() ? 1 : 0;
  if (size) *size = n;
  if (stride) *stride = st;
  if (a && out && n) std::memcpy(out, a, n < capacity ? n : capacity);
  return (int)r;
}
// device-resident A420 batch: plane 3 of every frame written by one launch on the element's context
int mi355el_roundedcorners_prepare_device(mi355el *h, uint8_t *d_frames, size_t frame_pitch, size_t alpha_offset, int n_frames) {
  auto *e = h ? dynamic_cast<RoundedCorners *>(h->e.get()) : nullptr;
  if (!e) return (int)FlowReturn::Error;
  return (int)e->prepare_output_buffer_device(d_frames, frame_pitch, alpha_offset, n_frames);
}
int mi355el_roundedcorners_src_formats(mi355el *h) {  // bit 0: I420 offered, bit 1: A420 offered
  auto *e = h ? dynamic_cast<RoundedCorners *>(h->e.get()) : nullptr;
  if (!e) return 0;
  int m = 0;
  for (int f : e->transform_caps_to_src()) m |= (f == 100) ? 1 : 2;
  return m;
}

}  // extern "C"
