/* gst/gstebur128level.c — `ebur128level` (GType GstEbuR128Level), a GstAudioFilter in passthrough mode that measures, over the
 * mi355fx C ABI. Surface mirrored from the reference (audio/audiofx/src/ebur128level/imp.rs): GType name :116-121 and
 * ebur128level/mod.rs (rank NONE; flags type GstEbuR128LevelMode :30-62), the `reset` action signal :124-139, properties
 * mode / post-messages / interval :145-225 (defaults :80-82), metadata :227-238, caps S16/S32/F32/F64, rate 1..2822399,
 * channels 1..63, interleaved or planar :490-511, passthrough :280-285, setup :513-613 (channel positions -> BS.1770
 * weights :520-595, interval in frames :597-601), transform_ip_passthrough :296-486: the buffer is cut at interval
 * boundaries, each piece goes to mi355_ebur128_add_frames[_planar] (the K-weighting biquads, gating blocks, true-peak FIR
 * of the `ebur128` crate run on the GPU) and every full interval posts one "ebur128-level" element message :371-470. */
#include <stdlib.h>
#include <gst/gst.h>
#include <gst/audio/audio.h>
#include <gst/audio/gstaudiofilter.h>
#include "../include/mi355fx.h"

GST_DEBUG_CATEGORY_STATIC(gst_ebur128_level_debug);
#define GST_CAT_DEFAULT gst_ebur128_level_debug

#define GST_TYPE_EBUR128_LEVEL (gst_ebur128_level_get_type())
G_DECLARE_FINAL_TYPE(GstEbuR128Level, gst_ebur128_level, GST, EBUR128_LEVEL, GstAudioFilter)

struct _GstEbuR128Level {
  GstAudioFilter parent;
  GMutex lock; /* settings */
  guint mode;
  gboolean post_messages;
  guint64 interval; /* ns */
  gint reset;       /* atomic: the `reset` signal was emitted (imp.rs:113) */
  /* state (streaming thread) */
  gboolean have_state;
  GstAudioInfo info;
  guint state_mode;
  gint sample_format; /* 0 S16, 1 S32, 2 F32, 3 F64 */
  guint64 num_frames, interval_frames, interval_frames_remaining;
  mi355_ctx *ctx;
  /* MI355_GROUP_MEMBERS=n: this process hosts n pipelines of one shape; their ebur128level instances (interleaved caps) are members
   * of the process-wide mi355_agroup of their configuration (include/mi355fx.h: mi355_agroup_shared_ebur128) - independent meters,
   * each with its own 100 ms phase and `reset`, that share the launches of every interval */
  mi355_agroup *agroup;
  int member;
};

G_DEFINE_TYPE(GstEbuR128Level, gst_ebur128_level, GST_TYPE_AUDIO_FILTER)

enum { PROP_0, PROP_MODE, PROP_POST_MESSAGES, PROP_INTERVAL };
enum { SIGNAL_RESET, LAST_SIGNAL };
static guint gst_ebur128_level_signals[LAST_SIGNAL];

/* GstEbuR128LevelMode (ebur128level/mod.rs:30-62): the nicks and bit values of the reference's flags type */
static GType gst_ebur128_level_mode_get_type(void) {
  static gsize type = 0;
  if (g_once_init_enter(&type)) {
    static const GFlagsValue values[] = {
        {MI355_EBUR128_MOMENTARY, "Calculate momentary loudness", "momentary"},
        {MI355_EBUR128_SHORT_TERM, "Calculate short-term loudness", "short-term"},
        {MI355_EBUR128_GLOBAL, "Calculate relative threshold and global loudness", "global"},
        {MI355_EBUR128_LOUDNESS_RANGE, "Calculate loudness range", "loudness-range"},
        {MI355_EBUR128_SAMPLE_PEAK, "Calculate sample peak", "sample-peak"},
        {MI355_EBUR128_TRUE_PEAK, "Calculate true peak", "true-peak"},
        {0, NULL, NULL}};
    g_once_init_leave(&type, g_flags_register_static("GstEbuR128LevelMode", values));
  }
  return (GType)type;
}

#define EBUR128_ALL_MODES (MI355_EBUR128_MOMENTARY | MI355_EBUR128_SHORT_TERM | MI355_EBUR128_GLOBAL | MI355_EBUR128_LOUDNESS_RANGE | MI355_EBUR128_SAMPLE_PEAK | MI355_EBUR128_TRUE_PEAK)
#define EBUR128_CAPS                                                                                                             \
  "audio/x-raw, format = (string) { " GST_AUDIO_NE(S16) ", " GST_AUDIO_NE(S32) ", " GST_AUDIO_NE(F32) ", " GST_AUDIO_NE(F64) " }, " \
  "rate = (int) [ 1, 2822399 ], channels = (int) [ 1, 63 ], layout = (string) { interleaved, non-interleaved }"

static void gst_ebur128_level_reset_action(GstEbuR128Level *self) { g_atomic_int_set(&self->reset, TRUE); } /* imp.rs:131-135 */

static void gst_ebur128_level_set_property(GObject *object, guint id, const GValue *value, GParamSpec *pspec) {
  GstEbuR128Level *self = GST_EBUR128_LEVEL(object);
  g_mutex_lock(&self->lock);
  switch (id) {
    case PROP_MODE: self->mode = g_value_get_flags(value); break;
    case PROP_POST_MESSAGES: self->post_messages = g_value_get_boolean(value); break;
    case PROP_INTERVAL: self->interval = g_value_get_uint64(value); break;
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID(object, id, pspec); break;
  }
  g_mutex_unlock(&self->lock);
}

static void gst_ebur128_level_get_property(GObject *object, guint id, GValue *value, GParamSpec *pspec) {
  GstEbuR128Level *self = GST_EBUR128_LEVEL(object);
  g_mutex_lock(&self->lock);
  switch (id) {
    case PROP_MODE: g_value_set_flags(value, self->mode); break;
    case PROP_POST_MESSAGES: g_value_set_boolean(value, self->post_messages); break;
    case PROP_INTERVAL: g_value_set_uint64(value, self->interval); break;
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID(object, id, pspec); break;
  }
  g_mutex_unlock(&self->lock);
}

static gboolean gst_ebur128_level_start(GstBaseTransform *trans) {
  GstEbuR128Level *self = GST_EBUR128_LEVEL(trans);
  int status = 0;
  self->ctx = mi355_ctx_create(0, &status);
  if (!self->ctx) {
    GST_ELEMENT_ERROR(self, LIBRARY, INIT, ("No MI355X context"), ("%s", mi355_status_string(status)));
    return FALSE;
  }
  return TRUE;
}

/* BaseTransformImpl::stop (imp.rs:287-294) */
static gboolean gst_ebur128_level_stop(GstBaseTransform *trans) {
  GstEbuR128Level *self = GST_EBUR128_LEVEL(trans);
  self->have_state = FALSE;
  if (self->agroup) mi355_agroup_release(self->agroup, self->member);
  self->agroup = NULL;
  if (self->ctx) {
    (void)mi355_ebur128_teardown(self->ctx);
    mi355_ctx_destroy(self->ctx);
  }
  self->ctx = NULL;
  return TRUE;
}

/* GstAudioChannelPosition -> the weight class of mi355_ebur128_setup: what ebur128::Channel the reference maps the position to
 * (imp.rs:520-580) and which BS.1770 weight the crate gives that channel: 0 unused, 1 weight 1.0, 2 weight 1.41, 3 dual mono */
static int gst_ebur128_level_channel_class(GstAudioChannelPosition p) {
  switch (p) {
    case GST_AUDIO_CHANNEL_POSITION_MONO: return 3;                                    /* DualMono */
    case GST_AUDIO_CHANNEL_POSITION_LFE1: case GST_AUDIO_CHANNEL_POSITION_LFE2: return 0; /* Unused */
    case GST_AUDIO_CHANNEL_POSITION_SIDE_LEFT: case GST_AUDIO_CHANNEL_POSITION_SIDE_RIGHT: return 2; /* Mp090 / Mm090 */
    case GST_AUDIO_CHANNEL_POSITION_INVALID: case GST_AUDIO_CHANNEL_POSITION_NONE: return 0;
    case GST_AUDIO_CHANNEL_POSITION_FRONT_LEFT: case GST_AUDIO_CHANNEL_POSITION_FRONT_RIGHT: case GST_AUDIO_CHANNEL_POSITION_FRONT_CENTER:
    case GST_AUDIO_CHANNEL_POSITION_REAR_LEFT: case GST_AUDIO_CHANNEL_POSITION_REAR_RIGHT: case GST_AUDIO_CHANNEL_POSITION_REAR_CENTER:
    case GST_AUDIO_CHANNEL_POSITION_FRONT_LEFT_OF_CENTER: case GST_AUDIO_CHANNEL_POSITION_FRONT_RIGHT_OF_CENTER:
    case GST_AUDIO_CHANNEL_POSITION_TOP_FRONT_LEFT: case GST_AUDIO_CHANNEL_POSITION_TOP_FRONT_RIGHT: case GST_AUDIO_CHANNEL_POSITION_TOP_FRONT_CENTER:
    case GST_AUDIO_CHANNEL_POSITION_TOP_CENTER: case GST_AUDIO_CHANNEL_POSITION_TOP_REAR_LEFT: case GST_AUDIO_CHANNEL_POSITION_TOP_REAR_RIGHT:
    case GST_AUDIO_CHANNEL_POSITION_TOP_SIDE_LEFT: case GST_AUDIO_CHANNEL_POSITION_TOP_SIDE_RIGHT: case GST_AUDIO_CHANNEL_POSITION_TOP_REAR_CENTER:
    case GST_AUDIO_CHANNEL_POSITION_BOTTOM_FRONT_CENTER: case GST_AUDIO_CHANNEL_POSITION_BOTTOM_FRONT_LEFT: case GST_AUDIO_CHANNEL_POSITION_BOTTOM_FRONT_RIGHT:
    case GST_AUDIO_CHANNEL_POSITION_WIDE_LEFT: case GST_AUDIO_CHANNEL_POSITION_WIDE_RIGHT:
    case GST_AUDIO_CHANNEL_POSITION_SURROUND_LEFT: case GST_AUDIO_CHANNEL_POSITION_SURROUND_RIGHT:
      return 1; /* Left, Right, Center, Mp135, Mm135, Mp180, MpSC, MmSC, U*, Tp000, B*: weight 1.0 in the crate */
    default: return 0; /* "Unknown channel position, ignoring channel" (imp.rs:568-577) */
  }
}

/* AudioFilterImpl::setup (imp.rs:513-613) */
static gboolean gst_ebur128_level_setup(GstAudioFilter *filter, const GstAudioInfo *info) {
  GstEbuR128Level *self = GST_EBUR128_LEVEL(filter);
  g_mutex_lock(&self->lock);
  const guint mode = self->mode;
  const guint64 interval = self->interval;
  g_mutex_unlock(&self->lock);
  const gint channels = GST_AUDIO_INFO_CHANNELS(info);
  int klass[64];
  for (gint c = 0; c < channels && c < 64; c++) /* unpositioned: every channel weighted like Center (imp.rs:589-595) */
    klass[c] = GST_AUDIO_INFO_IS_UNPOSITIONED(info) ? 1 : gst_ebur128_level_channel_class(GST_AUDIO_INFO_POSITION(info, c));
  if (self->agroup) mi355_agroup_release(self->agroup, self->member); /* renegotiation: a new meter, as EbuR128::new does */
  self->agroup = NULL;
  const char *members = g_getenv("MI355_GROUP_MEMBERS");
  if (members && atoi(members) > 1 && GST_AUDIO_INFO_LAYOUT(info) == GST_AUDIO_LAYOUT_INTERLEAVED) {
    int status = 0;
    self->agroup = mi355_agroup_shared_ebur128(0, atoi(members), (unsigned)channels, (unsigned)GST_AUDIO_INFO_RATE(info), mode, klass, &self->member, &status);
    if (!self->agroup) GST_WARNING_OBJECT(self, "no shared ebur128level group (%s): own launches", mi355_status_string(status));
    else (void)mi355_agroup_set_linger(self->agroup, g_getenv("MI355_GROUP_LINGER_US") ? (unsigned)atoi(g_getenv("MI355_GROUP_LINGER_US")) : 2000u, 0); /* a paused neighbour costs the others 2 ms, never a hang */
  }
  if (!self->agroup && mi355_ebur128_setup(self->ctx, (unsigned)channels, (unsigned)GST_AUDIO_INFO_RATE(info), mode, klass) != MI355_OK) {
    GST_ERROR_OBJECT(self, "Failed to create EBU R128: %s", mi355_ctx_last_error(self->ctx));
    return FALSE;
  }
  switch (GST_AUDIO_INFO_FORMAT(info)) {
    case GST_AUDIO_FORMAT_S16: self->sample_format = 0; break;
    case GST_AUDIO_FORMAT_S32: self->sample_format = 1; break;
    case GST_AUDIO_FORMAT_F32: self->sample_format = 2; break;
    case GST_AUDIO_FORMAT_F64: self->sample_format = 3; break;
    default: return FALSE;
  }
  self->info = *info;
  self->state_mode = mode;
  self->interval_frames = gst_util_uint64_scale(interval, (guint64)GST_AUDIO_INFO_RATE(info), GST_SECOND); /* mul_div_floor (imp.rs:597-601) */
  self->interval_frames_remaining = self->interval_frames;
  self->num_frames = 0;
  self->have_state = TRUE;
  return TRUE;
}

/* one "ebur128-level" message (imp.rs:365-470); a metric that cannot be read is logged and left out, as in the reference */
static void gst_ebur128_level_post(GstEbuR128Level *self, GstBaseTransform *trans, GstClockTime timestamp) {
  const GstSegment *segment = &trans->segment;
  GstStructure *s = gst_structure_new("ebur128-level", "timestamp", G_TYPE_UINT64, timestamp, "running-time", G_TYPE_UINT64,
                                      gst_segment_to_running_time(segment, GST_FORMAT_TIME, timestamp), "stream-time", G_TYPE_UINT64,
                                      gst_segment_to_stream_time(segment, GST_FORMAT_TIME, timestamp), NULL);
  double v = 0.0;
  const unsigned mode = self->state_mode;
  mi355_agroup *ag = self->agroup;
  const int me = self->member;
#define EB_ERR() (ag ? mi355_agroup_last_error(ag) : mi355_ctx_last_error(self->ctx))
  if (mode & MI355_EBUR128_MOMENTARY) {
    if ((ag ? mi355_agroup_ebur128_loudness(ag, me, 0, &v) : mi355_ebur128_loudness_momentary(self->ctx, &v)) == MI355_OK) gst_structure_set(s, "momentary-loudness", G_TYPE_DOUBLE, v, NULL);
    else GST_ERROR_OBJECT(self, "Failed to get momentary loudness: %s", EB_ERR());
  }
  if (mode & MI355_EBUR128_SHORT_TERM) {
    if ((ag ? mi355_agroup_ebur128_loudness(ag, me, 1, &v) : mi355_ebur128_loudness_shortterm(self->ctx, &v)) == MI355_OK) gst_structure_set(s, "shortterm-loudness", G_TYPE_DOUBLE, v, NULL);
    else GST_ERROR_OBJECT(self, "Failed to get shortterm loudness: %s", EB_ERR());
  }
  if (mode & MI355_EBUR128_GLOBAL) {
    if ((ag ? mi355_agroup_ebur128_loudness(ag, me, 2, &v) : mi355_ebur128_loudness_global(self->ctx, &v)) == MI355_OK) gst_structure_set(s, "global-loudness", G_TYPE_DOUBLE, v, NULL);
    else GST_ERROR_OBJECT(self, "Failed to get global loudness: %s", EB_ERR());
    if ((ag ? mi355_agroup_ebur128_loudness(ag, me, 3, &v) : mi355_ebur128_relative_threshold(self->ctx, &v)) == MI355_OK) gst_structure_set(s, "relative-threshold", G_TYPE_DOUBLE, v, NULL);
    else GST_ERROR_OBJECT(self, "Failed to get relative threshold: %s", EB_ERR());
  }
  if (mode & MI355_EBUR128_LOUDNESS_RANGE) {
    if ((ag ? mi355_agroup_ebur128_loudness(ag, me, 4, &v) : mi355_ebur128_loudness_range(self->ctx, &v)) == MI355_OK) gst_structure_set(s, "loudness-range", G_TYPE_DOUBLE, v, NULL);
    else GST_ERROR_OBJECT(self, "Failed to get loudness range: %s", EB_ERR());
  }
  for (int peak = 0; peak < 2; peak++) {
    if (!(mode & (peak ? MI355_EBUR128_TRUE_PEAK : MI355_EBUR128_SAMPLE_PEAK))) continue;
    GValue arr = G_VALUE_INIT;
    g_value_init(&arr, GST_TYPE_ARRAY);
    gboolean ok = TRUE;
    for (gint c = 0; c < GST_AUDIO_INFO_CHANNELS(&self->info) && ok; c++) {
      ok = (ag ? mi355_agroup_ebur128_peak(ag, me, peak, (unsigned)c, &v)
                : (peak ? mi355_ebur128_true_peak(self->ctx, (unsigned)c, &v) : mi355_ebur128_sample_peak(self->ctx, (unsigned)c, &v))) == MI355_OK;
      if (ok) {
        GValue d = G_VALUE_INIT;
        g_value_init(&d, G_TYPE_DOUBLE);
        g_value_set_double(&d, v);
        gst_value_array_append_and_take_value(&arr, &d);
      }
    }
    if (ok) gst_structure_set_value(s, peak ? "true-peak" : "sample-peak", &arr);
    else GST_ERROR_OBJECT(self, "Failed to get %s peaks: %s", peak ? "true" : "sample", EB_ERR());
    g_value_unset(&arr);
  }
  (void)gst_element_post_message(GST_ELEMENT(self), gst_message_new_element(GST_OBJECT(self), s));
}
#undef EB_ERR

/* BaseTransformImpl::transform_ip_passthrough (imp.rs:296-486) */
static GstFlowReturn gst_ebur128_level_transform_ip(GstBaseTransform *trans, GstBuffer *buf) {
  GstEbuR128Level *self = GST_EBUR128_LEVEL(trans);
  g_mutex_lock(&self->lock);
  const gboolean post = self->post_messages;
  g_mutex_unlock(&self->lock);
  if (!self->have_state) {
    GST_ELEMENT_ERROR(self, CORE, NEGOTIATION, ("Have no state yet"), (NULL));
    return GST_FLOW_NOT_NEGOTIATED;
  }
  GstAudioBuffer abuf;
  if (!gst_audio_buffer_map(&abuf, &self->info, buf, GST_MAP_READ)) {
    GST_ELEMENT_ERROR(self, RESOURCE, READ, ("Failed to map buffer"), (NULL));
    return GST_FLOW_ERROR;
  }
  static const size_t sample_bytes[4] = {2, 4, 4, 8};
  const size_t sb = sample_bytes[self->sample_format];
  const gint channels = GST_AUDIO_INFO_CHANNELS(&self->info);
  const gboolean planar = GST_AUDIO_INFO_LAYOUT(&self->info) == GST_AUDIO_LAYOUT_NON_INTERLEAVED;
  GstClockTime timestamp = GST_BUFFER_PTS(buf);
  size_t frames = abuf.n_samples, done = 0;
  GstFlowReturn ret = GST_FLOW_OK;
  while (frames - done > 0) {
    if (g_atomic_int_compare_and_exchange(&self->reset, TRUE, FALSE)) { /* imp.rs:320-333 */
      if ((self->agroup ? mi355_agroup_ebur128_reset(self->agroup, self->member) : mi355_ebur128_reset(self->ctx)) != MI355_OK) { ret = GST_FLOW_ERROR; break; }
      self->interval_frames_remaining = self->interval_frames;
      self->num_frames = 0;
    }
    const guint64 left = frames - done;
    const guint64 to_process = MIN(self->interval_frames_remaining, left);
    int rc;
    if (planar) {
      const void *planes[64];
      for (gint c = 0; c < channels && c < 64; c++) planes[c] = (const guint8 *)abuf.planes[c] + done * sb;
      rc = mi355_ebur128_add_frames_planar(self->ctx, planes, (size_t)to_process, self->sample_format);
    } else if (self->agroup) { /* this piece joins the launch set of the interval; the call returns when it has run */
      uint64_t ticket = 0;
      rc = mi355_agroup_submit_ebur128(self->agroup, self->member, (const guint8 *)abuf.planes[0] + done * sb * (size_t)channels, (size_t)to_process,
                                       self->sample_format, 0, &ticket);
      if (rc == MI355_OK) rc = mi355_agroup_wait(self->agroup, ticket, NULL);
    } else {
      rc = mi355_ebur128_add_frames(self->ctx, (const guint8 *)abuf.planes[0] + done * sb * (size_t)channels, (size_t)to_process, self->sample_format);
    }
    if (rc != MI355_OK) {
      GST_ELEMENT_ERROR(self, RESOURCE, READ, ("Failed to process buffer: %s", self->agroup ? mi355_agroup_last_error(self->agroup) : mi355_ctx_last_error(self->ctx)), (NULL));
      ret = GST_FLOW_ERROR;
      break;
    }
    done += (size_t)to_process;
    self->interval_frames_remaining -= to_process;
    self->num_frames += to_process;
    /* the timestamp until which measurements are included, not the starting one (imp.rs:353-361) */
    if (GST_CLOCK_TIME_IS_VALID(timestamp)) timestamp += gst_util_uint64_scale(to_process, GST_SECOND, (guint64)GST_AUDIO_INFO_RATE(&self->info));
    if (self->interval_frames_remaining == 0) {
      self->interval_frames_remaining = self->interval_frames;
      if (post) gst_ebur128_level_post(self, trans, timestamp);
    }
    if (self->interval_frames == 0) break; /* interval 0: nothing ever completes; do not spin */
  }
  gst_audio_buffer_unmap(&abuf);
  return ret;
}

static void gst_ebur128_level_finalize(GObject *object) {
  GstEbuR128Level *self = GST_EBUR128_LEVEL(object);
  g_mutex_clear(&self->lock);
  G_OBJECT_CLASS(gst_ebur128_level_parent_class)->finalize(object);
}

static void gst_ebur128_level_class_init(GstEbuR128LevelClass *klass) {
  GObjectClass *gobject = G_OBJECT_CLASS(klass);
  GstElementClass *element = GST_ELEMENT_CLASS(klass);
  GstBaseTransformClass *trans = GST_BASE_TRANSFORM_CLASS(klass);
  GstAudioFilterClass *afilter = GST_AUDIO_FILTER_CLASS(klass);
  gobject->set_property = gst_ebur128_level_set_property;
  gobject->get_property = gst_ebur128_level_get_property;
  gobject->finalize = gst_ebur128_level_finalize;
  const GParamFlags ready = (GParamFlags)(G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS | GST_PARAM_MUTABLE_READY);
  const GParamFlags playing = (GParamFlags)(G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS | GST_PARAM_MUTABLE_PLAYING);
  g_object_class_install_property(gobject, PROP_MODE,
      g_param_spec_flags("mode", "Mode", "Selection of metrics to calculate", gst_ebur128_level_mode_get_type(), EBUR128_ALL_MODES, ready));
  g_object_class_install_property(gobject, PROP_POST_MESSAGES,
      g_param_spec_boolean("post-messages", "Post Messages", "Whether to post messages on the bus for each interval", TRUE, playing));
  g_object_class_install_property(gobject, PROP_INTERVAL,
      g_param_spec_uint64("interval", "Interval", "Interval in nanoseconds for posting messages", 0, G_MAXUINT64 - 1, GST_SECOND, ready));
  /* the `reset` action signal (imp.rs:124-139) */
  gst_ebur128_level_signals[SIGNAL_RESET] = g_signal_new_class_handler("reset", G_TYPE_FROM_CLASS(klass), (GSignalFlags)(G_SIGNAL_RUN_LAST | G_SIGNAL_ACTION),
                                                                       G_CALLBACK(gst_ebur128_level_reset_action), NULL, NULL, NULL, G_TYPE_NONE, 0);
  gst_element_class_set_static_metadata(element, "EBU R128 Loudness Level Measurement", "Filter/Analyzer/Audio",
                                        "Measures different loudness metrics according to EBU R128", "Sebastian Dröge <sebastian@centricular.com>");
  GstCaps *caps = gst_caps_from_string(EBUR128_CAPS);
  gst_audio_filter_class_add_pad_templates(afilter, caps);
  gst_caps_unref(caps);
  trans->start = gst_ebur128_level_start;
  trans->stop = gst_ebur128_level_stop;
  trans->transform_ip = gst_ebur128_level_transform_ip;
  trans->passthrough_on_same_caps = TRUE;      /* the reference: mode AlwaysInPlace, passthrough_on_same_caps true, */
  trans->transform_ip_on_passthrough = TRUE;   /* transform_ip_on_passthrough true (imp.rs:280-285): the buffer is only read */
  afilter->setup = gst_ebur128_level_setup;
  GST_DEBUG_CATEGORY_INIT(gst_ebur128_level_debug, "ebur128level", 0, "EBU R128 Loudness Level (MI355X)");
}

static void gst_ebur128_level_init(GstEbuR128Level *self) {
  g_mutex_init(&self->lock);
  self->mode = EBUR128_ALL_MODES; /* DEFAULT_MODE = Mode::all() (imp.rs:80) */
  self->post_messages = TRUE;     /* imp.rs:81 */
  self->interval = GST_SECOND;    /* imp.rs:82 */
}

gboolean gst_ebur128_level_register(GstPlugin *plugin) {
  return gst_element_register(plugin, "ebur128level", GST_RANK_NONE, GST_TYPE_EBUR128_LEVEL); /* ebur128level/mod.rs */
}
