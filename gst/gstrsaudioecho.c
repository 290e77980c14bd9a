/* gst/gstrsaudioecho.c — `rsaudioecho` (GType GstRsAudioEcho), a GstAudioFilter that works in place, over the mi355fx C ABI.
 * Surface mirrored from the reference (audio/audiofx/src/audioecho/imp.rs): GType name :87-92 and audioecho/mod.rs (rank
 * NONE), four properties mutable in READY :96-167 (defaults :31-34), metadata :171-183, caps F32/F64 interleaved, any rate
 * and channel count :187-199 (allowed_caps of AudioFilterImpl), AlwaysInPlace :202-203, setup :248-259 (ring of
 * max-delay x rate x channels samples), transform_ip :205-227 -> mi355_echo_process_f32 / _f64 (the per-sample loop :64-84
 * runs on the GPU, ring state included), stop :229-234. */
#include <gst/gst.h>
#include <gst/audio/audio.h>
#include <gst/audio/gstaudiofilter.h>
#include <stdlib.h>
#include "../include/mi355fx.h"

GST_DEBUG_CATEGORY_STATIC(gst_rs_audio_echo_debug);
#define GST_CAT_DEFAULT gst_rs_audio_echo_debug

#define GST_TYPE_RS_AUDIO_ECHO (gst_rs_audio_echo_get_type())
G_DECLARE_FINAL_TYPE(GstRsAudioEcho, gst_rs_audio_echo, GST, RS_AUDIO_ECHO, GstAudioFilter)

struct _GstRsAudioEcho {
  GstAudioFilter parent;
  GMutex lock; /* settings */
  guint64 max_delay, delay; /* ns */
  gdouble intensity, feedback;
  gboolean have_state;
  gint rate, channels;
  gboolean f64;
  mi355_ctx *ctx;
  /* MI355_GROUP_MEMBERS=n: this process hosts n pipelines of one shape; their rsaudioecho instances share launch sets through the
   * process-wide mi355_agroup of their ring size (include/mi355fx.h: mi355_agroup_shared_echo) instead of three launches each */
  mi355_agroup *agroup;
  int member;
};

G_DEFINE_TYPE(GstRsAudioEcho, gst_rs_audio_echo, GST_TYPE_AUDIO_FILTER)

enum { PROP_0, PROP_MAX_DELAY, PROP_DELAY, PROP_INTENSITY, PROP_FEEDBACK };

#define ECHO_CAPS "audio/x-raw, format = (string) { " GST_AUDIO_NE(F32) ", " GST_AUDIO_NE(F64) " }, rate = (int) [ 0, MAX ], channels = (int) [ 0, MAX ], layout = (string) interleaved"

static void gst_rs_audio_echo_set_property(GObject *object, guint id, const GValue *value, GParamSpec *pspec) {
  GstRsAudioEcho *self = GST_RS_AUDIO_ECHO(object);
  g_mutex_lock(&self->lock);
  switch (id) {
    case PROP_MAX_DELAY:
      /* "can't be changed in PLAYING or PAUSED state": ignored once there is a ring (imp.rs:132-137) */
      if (!self->have_state) self->max_delay = g_value_get_uint64(value);
      break;
    case PROP_DELAY: self->delay = g_value_get_uint64(value); break;
    case PROP_INTENSITY: self->intensity = g_value_get_double(value); break;
    case PROP_FEEDBACK: self->feedback = g_value_get_double(value); break;
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID(object, id, pspec); break;
  }
  g_mutex_unlock(&self->lock);
}

static void gst_rs_audio_echo_get_property(GObject *object, guint id, GValue *value, GParamSpec *pspec) {
  GstRsAudioEcho *self = GST_RS_AUDIO_ECHO(object);
  g_mutex_lock(&self->lock);
  switch (id) {
    case PROP_MAX_DELAY: g_value_set_uint64(value, self->max_delay); break;
    case PROP_DELAY: g_value_set_uint64(value, self->delay); break;
    case PROP_INTENSITY: g_value_set_double(value, self->intensity); break;
    case PROP_FEEDBACK: g_value_set_double(value, self->feedback); break;
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID(object, id, pspec); break;
  }
  g_mutex_unlock(&self->lock);
}

static gboolean gst_rs_audio_echo_start(GstBaseTransform *trans) {
  GstRsAudioEcho *self = GST_RS_AUDIO_ECHO(trans);
  int status = 0;
  self->ctx = mi355_ctx_create(0, &status);
  if (!self->ctx) {
    GST_ELEMENT_ERROR(self, LIBRARY, INIT, ("No MI355X context"), ("%s", mi355_status_string(status)));
    return FALSE;
  }
  return TRUE;
}

/* BaseTransformImpl::stop (imp.rs:229-234): the state goes */
static gboolean gst_rs_audio_echo_stop(GstBaseTransform *trans) {
  GstRsAudioEcho *self = GST_RS_AUDIO_ECHO(trans);
  g_mutex_lock(&self->lock);
  self->have_state = FALSE;
  g_mutex_unlock(&self->lock);
  if (self->agroup) mi355_agroup_release(self->agroup, self->member);
  self->agroup = NULL;
  if (self->ctx) mi355_ctx_destroy(self->ctx);
  self->ctx = NULL;
  return TRUE;
}

/* AudioFilterImpl::setup (imp.rs:248-259): size = (max_delay * rate).seconds(), buffer_size = size * channels */
static gboolean gst_rs_audio_echo_setup(GstAudioFilter *filter, const GstAudioInfo *info) {
  GstRsAudioEcho *self = GST_RS_AUDIO_ECHO(filter);
  g_mutex_lock(&self->lock);
  const guint64 max_delay = self->max_delay;
  g_mutex_unlock(&self->lock);
  const guint64 size = gst_util_uint64_scale(max_delay, (guint64)GST_AUDIO_INFO_RATE(info), GST_SECOND);
  const char *members = g_getenv("MI355_GROUP_MEMBERS");
  if (self->agroup) mi355_agroup_release(self->agroup, self->member); /* renegotiation: a new ring, as RingBuffer::new does */
  self->agroup = NULL;
  if (members && atoi(members) >= 2) {
    int status = 0;
    self->agroup = mi355_agroup_shared_echo(0, atoi(members), (size_t)size * (size_t)GST_AUDIO_INFO_CHANNELS(info), &self->member, &status);
    if (!self->agroup) GST_WARNING_OBJECT(self, "no shared echo group (%s): own launches", mi355_status_string(status));
    else (void)mi355_agroup_set_linger(self->agroup, g_getenv("MI355_GROUP_LINGER_US") ? (unsigned)atoi(g_getenv("MI355_GROUP_LINGER_US")) : 2000u, 0); /* a paused neighbour costs the others 2 ms, never a hang */
  }
  if (!self->agroup && mi355_echo_setup(self->ctx, (size_t)size * (size_t)GST_AUDIO_INFO_CHANNELS(info)) != MI355_OK) {
    GST_ERROR_OBJECT(self, "mi355_echo_setup: %s", mi355_ctx_last_error(self->ctx));
    return FALSE;
  }
  g_mutex_lock(&self->lock);
  self->rate = GST_AUDIO_INFO_RATE(info);
  self->channels = GST_AUDIO_INFO_CHANNELS(info);
  self->f64 = GST_AUDIO_INFO_FORMAT(info) == GST_AUDIO_FORMAT_F64;
  self->have_state = TRUE;
  g_mutex_unlock(&self->lock);
  return TRUE;
}

/* BaseTransformImpl::transform_ip (imp.rs:205-227) */
static GstFlowReturn gst_rs_audio_echo_transform_ip(GstBaseTransform *trans, GstBuffer *buf) {
  GstRsAudioEcho *self = GST_RS_AUDIO_ECHO(trans);
  g_mutex_lock(&self->lock);
  const guint64 delay = MIN(self->max_delay, self->delay); /* cmp::min(settings.max_delay, settings.delay) (imp.rs:207) */
  const gdouble intensity = self->intensity, feedback = self->feedback;
  const gboolean have_state = self->have_state, f64 = self->f64;
  const gint rate = self->rate, channels = self->channels;
  g_mutex_unlock(&self->lock);
  if (!have_state) return GST_FLOW_NOT_NEGOTIATED; /* ok_or(FlowError::NotNegotiated) (imp.rs:210) */
  GstMapInfo map;
  if (!gst_buffer_map(buf, &map, GST_MAP_READWRITE)) return GST_FLOW_ERROR; /* map_writable().map_err(Error) (imp.rs:212) */
  /* delay_frames = (delay * channels * rate).seconds() (imp.rs:74-77) */
  const size_t delay_samples = (size_t)gst_util_uint64_scale(delay, (guint64)channels * (guint64)rate, GST_SECOND);
  int rc;
  if (self->agroup) { /* this buffer joins the launch set of the interval; the call returns when it has run */
    uint64_t ticket = 0;
    rc = mi355_agroup_submit_echo(self->agroup, self->member, map.data, map.size / (f64 ? sizeof(double) : sizeof(float)), f64 ? 1 : 0, delay_samples, intensity,
                                  feedback, 0, &ticket);
    if (rc == MI355_OK) rc = mi355_agroup_wait(self->agroup, ticket, NULL);
  } else {
    rc = f64 ? mi355_echo_process_f64(self->ctx, (double *)map.data, map.size / sizeof(double), delay_samples, intensity, feedback)
             : mi355_echo_process_f32(self->ctx, (float *)map.data, map.size / sizeof(float), delay_samples, intensity, feedback);
  }
  gst_buffer_unmap(buf, &map);
  if (rc != MI355_OK) {
    GST_ERROR_OBJECT(self, "mi355_echo_process: %s", self->agroup ? mi355_agroup_last_error(self->agroup) : mi355_ctx_last_error(self->ctx));
    return GST_FLOW_ERROR;
  }
  return GST_FLOW_OK;
}

static void gst_rs_audio_echo_finalize(GObject *object) {
  GstRsAudioEcho *self = GST_RS_AUDIO_ECHO(object);
  g_mutex_clear(&self->lock);
  G_OBJECT_CLASS(gst_rs_audio_echo_parent_class)->finalize(object);
}

static void gst_rs_audio_echo_class_init(GstRsAudioEchoClass *klass) {
  GObjectClass *gobject = G_OBJECT_CLASS(klass);
  GstElementClass *element = GST_ELEMENT_CLASS(klass);
  GstBaseTransformClass *trans = GST_BASE_TRANSFORM_CLASS(klass);
  GstAudioFilterClass *afilter = GST_AUDIO_FILTER_CLASS(klass);
  gobject->set_property = gst_rs_audio_echo_set_property;
  gobject->get_property = gst_rs_audio_echo_get_property;
  gobject->finalize = gst_rs_audio_echo_finalize;
  const GParamFlags f = (GParamFlags)(G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS | GST_PARAM_MUTABLE_READY);
  g_object_class_install_property(gobject, PROP_MAX_DELAY,
      g_param_spec_uint64("max-delay", "Maximum Delay", "Maximum delay of the echo in nanoseconds (can't be changed in PLAYING or PAUSED state)",
                          0, G_MAXUINT64 - 1, GST_SECOND, f));
  g_object_class_install_property(gobject, PROP_DELAY,
      g_param_spec_uint64("delay", "Delay", "Delay of the echo in nanoseconds", 0, G_MAXUINT64 - 1, 500 * GST_SECOND, f));
  g_object_class_install_property(gobject, PROP_INTENSITY, g_param_spec_double("intensity", "Intensity", "Intensity of the echo", 0.0, 1.0, 0.5, f));
  g_object_class_install_property(gobject, PROP_FEEDBACK, g_param_spec_double("feedback", "Feedback", "Amount of feedback", 0.0, 1.0, 0.0, f));
  gst_element_class_set_static_metadata(element, "Audio echo", "Filter/Effect/Audio", "Adds an echo or reverb effect to an audio stream",
                                        "Sebastian Dröge <sebastian@centricular.com>");
  GstCaps *caps = gst_caps_from_string(ECHO_CAPS);
  gst_audio_filter_class_add_pad_templates(afilter, caps); /* AudioFilterImpl::allowed_caps (imp.rs:187-199) */
  gst_caps_unref(caps);
  trans->start = gst_rs_audio_echo_start;
  trans->stop = gst_rs_audio_echo_stop;
  trans->transform_ip = gst_rs_audio_echo_transform_ip; /* only _ip installed == BaseTransformMode::AlwaysInPlace (imp.rs:202) */
  trans->passthrough_on_same_caps = FALSE;
  trans->transform_ip_on_passthrough = FALSE;
  afilter->setup = gst_rs_audio_echo_setup;
  GST_DEBUG_CATEGORY_INIT(gst_rs_audio_echo_debug, "rsaudioecho", 0, "Rust Audio Echo Filter (MI355X)");
}

static void gst_rs_audio_echo_init(GstRsAudioEcho *self) {
  g_mutex_init(&self->lock);
  self->max_delay = GST_SECOND;        /* DEFAULT_MAX_DELAY (imp.rs:31) */
  self->delay = 500 * GST_SECOND;      /* DEFAULT_DELAY (imp.rs:32) */
  self->intensity = 0.5;               /* imp.rs:33 */
  self->feedback = 0.0;                /* imp.rs:34 */
}

gboolean gst_rs_audio_echo_register(GstPlugin *plugin) {
  return gst_element_register(plugin, "rsaudioecho", GST_RANK_NONE, GST_TYPE_RS_AUDIO_ECHO); /* audioecho/mod.rs */
}
