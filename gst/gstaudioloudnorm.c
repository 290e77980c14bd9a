/* gst/gstaudioloudnorm.c — `audioloudnorm` (GType GstAudioLoudNorm), a plain GstElement with its own chain function, over the
 * mi355fx C ABI. Surface mirrored from the reference (audio/audiofx/src/audioloudnorm/imp.rs): GType name :1716-1718 and
 * audioloudnorm/mod.rs (rank NONE), sink / src pads with PROXY_CAPS :1720-1750, four gdouble properties mutable in READY
 * :1754-1830 (defaults :37-40), metadata :1834-1845, caps F64 interleaved at 192 kHz, any channel count :1847-1870, the
 * chain function :1544-1583 (a DISCONT buffer drains and starts a new State), sink events :1585-1672 (caps / EOS / segment
 * drain, flush-stop resets), the latency query +3 s :1675-1700, change_state -> state dropped on PausedToReady :1877-1890.
 * State::new / drain_full_frames / drain / process (:130-828) and the true-peak limiter (:845-1430) are mi355_loudnorm_setup /
 * _push / _drain: the element keeps the pads, the timestamps and the events, the library the adapter and the DSP. */
#include <gst/gst.h>
#include <gst/audio/audio.h>
#include <stdlib.h>
#include "../include/mi355fx.h"

GST_DEBUG_CATEGORY_STATIC(gst_audio_loud_norm_debug);
#define GST_CAT_DEFAULT gst_audio_loud_norm_debug

#define GST_TYPE_AUDIO_LOUD_NORM (gst_audio_loud_norm_get_type())
G_DECLARE_FINAL_TYPE(GstAudioLoudNorm, gst_audio_loud_norm, GST, AUDIO_LOUD_NORM, GstElement)

#define LOUDNORM_FRAME 19200u /* 100 ms at 192 kHz (imp.rs:42-44) */

struct _GstAudioLoudNorm {
  GstElement parent;
  GstPad *sinkpad, *srcpad;
  GMutex lock; /* settings */
  gdouble loudness_target, loudness_range_target, max_true_peak, offset;
  /* state (streaming thread) */
  gboolean have_state;
  GstAudioInfo info;
  GstClockTime base_pts; /* timestamp of the first sample since the state was created */
  guint64 out_frames;    /* frames pushed downstream since then: the output timeline is continuous from base_pts */
  guint64 in_frames;     /* frames received and not yet accounted for by output (bounds the drain) */
  mi355_ctx *ctx;
  /* MI355_GROUP_MEMBERS=n: the n audioloudnorm instances of this process (one configuration) advance in lock step through the
   * process-wide mi355_agroup (include/mi355fx.h: mi355_agroup_shared_loudnorm): one launch set per 100 ms frame for all of them */
  mi355_agroup *agroup;
  int member;
};

G_DEFINE_TYPE(GstAudioLoudNorm, gst_audio_loud_norm, GST_TYPE_ELEMENT)

enum { PROP_0, PROP_LOUDNESS_TARGET, PROP_LOUDNESS_RANGE_TARGET, PROP_MAX_TRUE_PEAK, PROP_OFFSET };

static GstStaticPadTemplate sink_template = GST_STATIC_PAD_TEMPLATE("sink", GST_PAD_SINK, GST_PAD_ALWAYS,
    GST_STATIC_CAPS("audio/x-raw, format = (string) " GST_AUDIO_NE(F64) ", rate = (int) 192000, channels = (int) [ 1, MAX ], layout = (string) interleaved"));
static GstStaticPadTemplate src_template = GST_STATIC_PAD_TEMPLATE("src", GST_PAD_SRC, GST_PAD_ALWAYS,
    GST_STATIC_CAPS("audio/x-raw, format = (string) " GST_AUDIO_NE(F64) ", rate = (int) 192000, channels = (int) [ 1, MAX ], layout = (string) interleaved"));

static gdouble *gst_audio_loud_norm_field(GstAudioLoudNorm *self, guint id) {
  switch (id) {
    case PROP_LOUDNESS_TARGET: return &self->loudness_target;
    case PROP_LOUDNESS_RANGE_TARGET: return &self->loudness_range_target;
    case PROP_MAX_TRUE_PEAK: return &self->max_true_peak;
    case PROP_OFFSET: return &self->offset;
    default: return NULL;
  }
}

static void gst_audio_loud_norm_set_property(GObject *object, guint id, const GValue *value, GParamSpec *pspec) {
  GstAudioLoudNorm *self = GST_AUDIO_LOUD_NORM(object);
  gdouble *f = gst_audio_loud_norm_field(self, id);
  if (!f) { G_OBJECT_WARN_INVALID_PROPERTY_ID(object, id, pspec); return; }
  g_mutex_lock(&self->lock);
  *f = g_value_get_double(value);
  g_mutex_unlock(&self->lock);
}

static void gst_audio_loud_norm_get_property(GObject *object, guint id, GValue *value, GParamSpec *pspec) {
  GstAudioLoudNorm *self = GST_AUDIO_LOUD_NORM(object);
  gdouble *f = gst_audio_loud_norm_field(self, id);
  if (!f) { G_OBJECT_WARN_INVALID_PROPERTY_ID(object, id, pspec); return; }
  g_mutex_lock(&self->lock);
  g_value_set_double(value, *f);
  g_mutex_unlock(&self->lock);
}

/* State::new(settings, info) (imp.rs:130-205) */
static gboolean gst_audio_loud_norm_new_state(GstAudioLoudNorm *self) {
  g_mutex_lock(&self->lock);
  const gdouble lt = self->loudness_target, lrt = self->loudness_range_target, tp = self->max_true_peak, off = self->offset;
  g_mutex_unlock(&self->lock);
  const char *members = g_getenv("MI355_GROUP_MEMBERS");
  if (self->agroup) mi355_agroup_release(self->agroup, self->member); /* a new State: a new membership */
  self->agroup = NULL;
  if (members && atoi(members) >= 2) {
    int status = 0;
    self->agroup = mi355_agroup_shared_loudnorm(0, atoi(members), (unsigned)GST_AUDIO_INFO_CHANNELS(&self->info), lt, lrt, tp, off, &self->member, &status);
    if (!self->agroup) GST_WARNING_OBJECT(self, "no shared loudnorm group (%s): own launches", mi355_status_string(status));
    /* independent members: an instance that starts later, pauses or ends costs the others 2 ms per interval, never a hang */
    else (void)mi355_agroup_set_linger(self->agroup, g_getenv("MI355_GROUP_LINGER_US") ? (unsigned)atoi(g_getenv("MI355_GROUP_LINGER_US")) : 2000u, 0);
  }
  if (!self->agroup && mi355_loudnorm_setup(self->ctx, (unsigned)GST_AUDIO_INFO_CHANNELS(&self->info), lt, lrt, tp, off) != MI355_OK) {
    GST_ERROR_OBJECT(self, "mi355_loudnorm_setup: %s", mi355_ctx_last_error(self->ctx));
    self->have_state = FALSE;
    return FALSE;
  }
  self->base_pts = GST_CLOCK_TIME_NONE;
  self->out_frames = 0;
  self->in_frames = 0;
  self->have_state = TRUE;
  return TRUE;
}

/* wraps `frames` output frames into a buffer with the next timestamp of the continuous output timeline and pushes it */
static GstFlowReturn gst_audio_loud_norm_push_frames(GstAudioLoudNorm *self, gdouble *samples, gsize frames) {
  const guint bpf = (guint)GST_AUDIO_INFO_BPF(&self->info), rate = (guint)GST_AUDIO_INFO_RATE(&self->info);
  GstBuffer *out = gst_buffer_new_wrapped(samples, frames * bpf);
  if (GST_CLOCK_TIME_IS_VALID(self->base_pts)) GST_BUFFER_PTS(out) = self->base_pts + gst_util_uint64_scale(self->out_frames, GST_SECOND, rate);
  GST_BUFFER_DURATION(out) = gst_util_uint64_scale(frames, GST_SECOND, rate); /* imp.rs:248-256 */
  self->out_frames += frames;
  return gst_pad_push(self->srcpad, out);
}

/* State::drain (imp.rs:265-330): GST_FLOW_EOS when there was nothing at all to drain */
static GstFlowReturn gst_audio_loud_norm_drain(GstAudioLoudNorm *self) {
  const gsize cap = 30 * (gsize)LOUDNORM_FRAME + (gsize)self->in_frames + LOUDNORM_FRAME;
  gdouble *out = g_new(gdouble, cap * (gsize)GST_AUDIO_INFO_CHANNELS(&self->info));
  size_t n = 0;
  int eos = 0;
  if ((self->agroup ? mi355_agroup_loudnorm_drain(self->agroup, self->member, out, cap, &n, &eos) : mi355_loudnorm_drain(self->ctx, out, cap, &n, &eos)) != MI355_OK) {
    GST_ERROR_OBJECT(self, "mi355_loudnorm_drain: %s", mi355_ctx_last_error(self->ctx));
    g_free(out);
    return GST_FLOW_ERROR;
  }
  if (eos || n == 0) { g_free(out); return GST_FLOW_EOS; }
  return gst_audio_loud_norm_push_frames(self, out, n);
}

/* sink_chain (imp.rs:1544-1583) */
static GstFlowReturn gst_audio_loud_norm_chain(GstPad *pad, GstObject *parent, GstBuffer *buffer) {
  GstAudioLoudNorm *self = GST_AUDIO_LOUD_NORM(parent);
  if (!self->have_state) {
    GST_ERROR_OBJECT(self, "Not negotiated yet");
    gst_buffer_unref(buffer);
    return GST_FLOW_NOT_NEGOTIATED;
  }
  if (GST_BUFFER_IS_DISCONT(buffer)) { /* "Draining on discontinuity" :1560-1571 */
    const GstFlowReturn r = gst_audio_loud_norm_drain(self);
    if (r != GST_FLOW_OK && r != GST_FLOW_EOS) { gst_buffer_unref(buffer); return r; }
    if (!gst_audio_loud_norm_new_state(self)) { gst_buffer_unref(buffer); return GST_FLOW_ERROR; }
  }
  GstMapInfo map;
  if (!gst_buffer_map(buffer, &map, GST_MAP_READ)) { gst_buffer_unref(buffer); return GST_FLOW_ERROR; }
  const guint bpf = (guint)GST_AUDIO_INFO_BPF(&self->info);
  const gsize frames = map.size / bpf;
  if (!GST_CLOCK_TIME_IS_VALID(self->base_pts)) self->base_pts = GST_BUFFER_PTS(buffer);
  /* drain_full_frames can complete the 3 s first frame and any number of 100 ms frames with this buffer */
  const gsize cap = (gsize)self->in_frames + frames + 30 * (gsize)LOUDNORM_FRAME;
  gdouble *out = g_new(gdouble, cap * (gsize)GST_AUDIO_INFO_CHANNELS(&self->info));
  size_t n = 0;
  const int rc = self->agroup ? mi355_agroup_loudnorm_push(self->agroup, self->member, (const double *)map.data, frames, out, cap, &n)
                              : mi355_loudnorm_push(self->ctx, (const double *)map.data, frames, out, cap, &n);
  gst_buffer_unmap(buffer, &map);
  gst_buffer_unref(buffer);
  if (rc != MI355_OK) {
    GST_ERROR_OBJECT(self, "mi355_loudnorm_push: %s", mi355_ctx_last_error(self->ctx));
    g_free(out);
    return GST_FLOW_ERROR;
  }
  self->in_frames += frames;
  if (n == 0) { g_free(out); return GST_FLOW_OK; }
  self->in_frames = self->in_frames > n ? self->in_frames - n : 0;
  return gst_audio_loud_norm_push_frames(self, out, n);
}

/* sink_event (imp.rs:1585-1672) */
static gboolean gst_audio_loud_norm_sink_event(GstPad *pad, GstObject *parent, GstEvent *event) {
  GstAudioLoudNorm *self = GST_AUDIO_LOUD_NORM(parent);
  switch (GST_EVENT_TYPE(event)) {
    case GST_EVENT_CAPS: {
      GstCaps *caps = NULL;
      GstAudioInfo info;
      gst_event_parse_caps(event, &caps);
      if (!gst_audio_info_from_caps(&info, caps)) {
        GST_ERROR_OBJECT(pad, "Failed to parse caps");
        gst_event_unref(event);
        return FALSE;
      }
      if (self->have_state) { /* what the old state still holds leaves first (imp.rs:1606-1613) */
        const GstFlowReturn r = gst_audio_loud_norm_drain(self);
        if (r != GST_FLOW_OK && r != GST_FLOW_EOS) { gst_event_unref(event); return FALSE; }
      }
      self->info = info;
      if (!gst_audio_loud_norm_new_state(self)) { gst_event_unref(event); return FALSE; }
      break;
    }
    case GST_EVENT_EOS:
    case GST_EVENT_SEGMENT:
      if (self->have_state) { /* imp.rs:1629-1655 */
        const GstFlowReturn r = gst_audio_loud_norm_drain(self);
        if (r != GST_FLOW_OK && r != GST_FLOW_EOS) { gst_event_unref(event); return FALSE; }
        if (!gst_audio_loud_norm_new_state(self)) { gst_event_unref(event); return FALSE; }
      }
      break;
    case GST_EVENT_FLUSH_STOP:
      if (self->have_state && !gst_audio_loud_norm_new_state(self)) { gst_event_unref(event); return FALSE; } /* imp.rs:1657-1667 */
      break;
    default: break;
  }
  return gst_pad_event_default(pad, parent, event);
}

/* src_query (imp.rs:1675-1700): three seconds of latency on top of upstream's */
static gboolean gst_audio_loud_norm_src_query(GstPad *pad, GstObject *parent, GstQuery *query) {
  GstAudioLoudNorm *self = GST_AUDIO_LOUD_NORM(parent);
  if (GST_QUERY_TYPE(query) == GST_QUERY_LATENCY) {
    GstQuery *peer = gst_query_new_latency();
    gboolean ok = gst_pad_peer_query(self->sinkpad, peer);
    if (ok) {
      gboolean live;
      GstClockTime min, max;
      gst_query_parse_latency(peer, &live, &min, &max);
      gst_query_set_latency(query, live, min + 3 * GST_SECOND, GST_CLOCK_TIME_IS_VALID(max) ? max + 3 * GST_SECOND : max);
    }
    gst_query_unref(peer);
    return ok;
  }
  return gst_pad_query_default(pad, parent, query);
}

/* ElementImpl::change_state (imp.rs:1877-1890) + the context that comes and goes with READY */
static GstStateChangeReturn gst_audio_loud_norm_change_state(GstElement *element, GstStateChange transition) {
  GstAudioLoudNorm *self = GST_AUDIO_LOUD_NORM(element);
  if (transition == GST_STATE_CHANGE_NULL_TO_READY) {
    int status = 0;
    self->ctx = mi355_ctx_create(0, &status);
    if (!self->ctx) {
      GST_ELEMENT_ERROR(self, LIBRARY, INIT, ("No MI355X context"), ("%s", mi355_status_string(status)));
      return GST_STATE_CHANGE_FAILURE;
    }
  }
  const GstStateChangeReturn ret = GST_ELEMENT_CLASS(gst_audio_loud_norm_parent_class)->change_state(element, transition);
  if (transition == GST_STATE_CHANGE_PAUSED_TO_READY) { /* "Drop state" */
    self->have_state = FALSE;
    if (self->agroup) mi355_agroup_release(self->agroup, self->member); /* the others go on without this member */
    self->agroup = NULL;
    if (self->ctx) (void)mi355_loudnorm_teardown(self->ctx);
  }
  if (transition == GST_STATE_CHANGE_READY_TO_NULL && self->ctx) {
    mi355_ctx_destroy(self->ctx);
    self->ctx = NULL;
  }
  return ret;
}

static void gst_audio_loud_norm_finalize(GObject *object) {
  GstAudioLoudNorm *self = GST_AUDIO_LOUD_NORM(object);
  g_mutex_clear(&self->lock);
  G_OBJECT_CLASS(gst_audio_loud_norm_parent_class)->finalize(object);
}

static void gst_audio_loud_norm_class_init(GstAudioLoudNormClass *klass) {
  GObjectClass *gobject = G_OBJECT_CLASS(klass);
  GstElementClass *element = GST_ELEMENT_CLASS(klass);
  gobject->set_property = gst_audio_loud_norm_set_property;
  gobject->get_property = gst_audio_loud_norm_get_property;
  gobject->finalize = gst_audio_loud_norm_finalize;
  const GParamFlags f = (GParamFlags)(G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS | GST_PARAM_MUTABLE_READY);
  g_object_class_install_property(gobject, PROP_LOUDNESS_TARGET,
      g_param_spec_double("loudness-target", "Loudness Target", "Loudness target in LUFS", -70.0, -5.0, -24.0, f));
  g_object_class_install_property(gobject, PROP_LOUDNESS_RANGE_TARGET,
      g_param_spec_double("loudness-range-target", "Loudness Range Target", "Loudness range target in LU", 1.0, 20.0, 7.0, f));
  g_object_class_install_property(gobject, PROP_MAX_TRUE_PEAK,
      g_param_spec_double("max-true-peak", "Maximum True Peak", "Maximum True Peak in dbTP", -9.0, 0.0, -2.0, f));
  g_object_class_install_property(gobject, PROP_OFFSET, g_param_spec_double("offset", "Offset Gain", "Offset Gain in LU", -99.0, 99.0, 0.0, f));
  gst_element_class_set_static_metadata(element, "Audio loudness normalizer", "Filter/Effect/Audio", "Normalizes perceived loudness of an audio stream",
                                        "Sebastian Dröge <sebastian@centricular.com>");
  gst_element_class_add_static_pad_template(element, &sink_template);
  gst_element_class_add_static_pad_template(element, &src_template);
  element->change_state = gst_audio_loud_norm_change_state;
  GST_DEBUG_CATEGORY_INIT(gst_audio_loud_norm_debug, "audioloudnorm", 0, "Audio Loudless Normalization (MI355X)");
}

static void gst_audio_loud_norm_init(GstAudioLoudNorm *self) {
  g_mutex_init(&self->lock);
  self->loudness_target = -24.0;      /* imp.rs:37-40 */
  self->loudness_range_target = 7.0;
  self->max_true_peak = -2.0;
  self->offset = 0.0;
  self->sinkpad = gst_pad_new_from_static_template(&sink_template, "sink");
  gst_pad_set_chain_function(self->sinkpad, gst_audio_loud_norm_chain);
  gst_pad_set_event_function(self->sinkpad, gst_audio_loud_norm_sink_event);
  GST_PAD_SET_PROXY_CAPS(self->sinkpad);
  gst_element_add_pad(GST_ELEMENT(self), self->sinkpad);
  self->srcpad = gst_pad_new_from_static_template(&src_template, "src");
  gst_pad_set_query_function(self->srcpad, gst_audio_loud_norm_src_query);
  GST_PAD_SET_PROXY_CAPS(self->srcpad);
  gst_element_add_pad(GST_ELEMENT(self), self->srcpad);
}

gboolean gst_audio_loud_norm_register(GstPlugin *plugin) {
  return gst_element_register(plugin, "audioloudnorm", GST_RANK_NONE, GST_TYPE_AUDIO_LOUD_NORM); /* audioloudnorm/mod.rs */
}
