/* gst/gsthrtfrender.c — `hrtfrender` (GType GstHrtfRender), a GstBaseTransform that never works in place, over the mi355fx C
 * ABI. Surface mirrored from the reference (audio/hrtf/src/hrtf/imp.rs): GType name :324-328 and hrtf/mod.rs (rank NONE),
 * properties hrir-raw / hrir-file / interpolation-steps / block-length / use-rayon (READY) and spatial-objects (PLAYING)
 * :332-399 with set/get :403-494 (defaults :36-39; a spatial-objects update with the wrong channel count is ignored with a
 * warning once negotiated :438-450), metadata :500-510, caps F32 interleaved, 1..=64 channels in, 2 out :512-545,
 * NeverInPlace :549-552, transform :554-572 (adapter; a full block renders), transform_size :574-599, transform_caps :602-646,
 * set_caps :648-707 (positions -> objects when none are set :660-672, sphere from raw bytes or file, block_samples =
 * block-length x interpolation-steps), sink_event :710-736 (flush-stop empties, EOS drains :281-352), stop :748-753.
 * HrtfProcessor::process_samples over every channel (the `hrtf` crate: interpolated HRIR lookup + block convolution +
 * overlap tails) is mi355_hrtf_process_block; `use-rayon` is accepted and means nothing here: the channels are the grid. */
#include <gst/gst.h>
#include <gst/audio/audio.h>
#include <gst/base/gstbasetransform.h>
#include <gst/base/gstadapter.h>
#include "../include/mi355fx.h"
#include "../gst-plugins-rs_amd/host/mi355fx_host.h"

GST_DEBUG_CATEGORY_STATIC(gst_hrtf_render_debug);
#define GST_CAT_DEFAULT gst_hrtf_render_debug

#define GST_TYPE_HRTF_RENDER (gst_hrtf_render_get_type())
G_DECLARE_FINAL_TYPE(GstHrtfRender, gst_hrtf_render, GST, HRTF_RENDER, GstBaseTransform)

#define HRTF_MAX_CHANNELS 64

typedef struct {
  gint coordinate_system; /* GstHrtfCoordinateSystem: 0 cartesian, 1 left-handed (default), 2 right-handed */
  gfloat xyz[3];
  gfloat distance_gain;
} HrtfObject;

struct _GstHrtfRender {
  GstBaseTransform parent;
  GMutex lock; /* settings */
  guint64 interpolation_steps, block_length;
  gboolean use_rayon;
  GBytes *hrir_raw;
  gchar *hrir_file;
  HrtfObject objects[HRTF_MAX_CHANNELS];
  guint n_objects; /* 0 = none set */
  /* state (streaming thread; `lock` also guards have_state / channels for the property setter) */
  gboolean have_state;
  gint rate, channels;
  gsize block_samples;
  GstAdapter *adapter;
  mi355_ctx *ctx;
};

G_DEFINE_TYPE(GstHrtfRender, gst_hrtf_render, GST_TYPE_BASE_TRANSFORM)

enum { PROP_0, PROP_HRIR_RAW, PROP_HRIR_FILE, PROP_INTERPOLATION_STEPS, PROP_BLOCK_LENGTH, PROP_USE_RAYON, PROP_SPATIAL_OBJECTS };

static GstStaticPadTemplate sink_template = GST_STATIC_PAD_TEMPLATE("sink", GST_PAD_SINK, GST_PAD_ALWAYS,
    GST_STATIC_CAPS("audio/x-raw, format = (string) " GST_AUDIO_NE(F32) ", rate = (int) [ 1, MAX ], channels = (int) [ 1, 64 ], layout = (string) interleaved"));
static GstStaticPadTemplate src_template = GST_STATIC_PAD_TEMPLATE("src", GST_PAD_SRC, GST_PAD_ALWAYS,
    GST_STATIC_CAPS("audio/x-raw, format = (string) " GST_AUDIO_NE(F32) ", rate = (int) [ 1, MAX ], channels = (int) 2, layout = (string) interleaved"));

/* GstHrtfCoordinateSystem (audio/hrtf/src/hrtf/mod.rs): cartesian, left-handed, right-handed */
static GType gst_hrtf_coordinate_system_get_type(void) {
  static gsize type = 0;
  if (g_once_init_enter(&type)) {
    static const GEnumValue values[] = {{0, "Cartesian", "cartesian"}, {1, "LeftHanded", "left-handed"}, {2, "RightHanded", "right-handed"}, {0, NULL, NULL}};
    g_once_init_leave(&type, g_enum_register_static("GstHrtfCoordinateSystem", values));
  }
  return (GType)type;
}

/* From<gst::Structure> for SpatialObject (spatial.rs:132-160): x, y, z are required, the rest defaults */
static gboolean gst_hrtf_render_object_from_structure(const GstStructure *s, HrtfObject *o) {
  gdouble d;
  o->distance_gain = 1.0f; /* DEFAULT_OBJECT_DISTANCE_GAIN */
  o->coordinate_system = 1; /* DEFAULT_OBJECT_COORDINATE_SYSTEM = LeftHanded */
  if (!gst_structure_get(s, "x", G_TYPE_FLOAT, &o->xyz[0], "y", G_TYPE_FLOAT, &o->xyz[1], "z", G_TYPE_FLOAT, &o->xyz[2], NULL)) return FALSE;
  if (gst_structure_get(s, "distance-gain", G_TYPE_FLOAT, &o->distance_gain, NULL)) { /* given */ }
  else if (gst_structure_get(s, "distance-gain", G_TYPE_DOUBLE, &d, NULL)) o->distance_gain = (gfloat)d;
  (void)gst_structure_get_enum(s, "coordinate-system", gst_hrtf_coordinate_system_get_type(), &o->coordinate_system);
  return TRUE;
}

static void gst_hrtf_render_set_property(GObject *object, guint id, const GValue *value, GParamSpec *pspec) {
  GstHrtfRender *self = GST_HRTF_RENDER(object);
  g_mutex_lock(&self->lock);
  switch (id) {
    case PROP_HRIR_RAW:
      if (self->hrir_raw) g_bytes_unref(self->hrir_raw);
      self->hrir_raw = (GBytes *)g_value_dup_boxed(value);
      break;
    case PROP_HRIR_FILE:
      g_free(self->hrir_file);
      self->hrir_file = g_value_dup_string(value);
      break;
    case PROP_INTERPOLATION_STEPS: self->interpolation_steps = g_value_get_uint64(value); break;
    case PROP_BLOCK_LENGTH: self->block_length = g_value_get_uint64(value); break;
    case PROP_USE_RAYON: self->use_rayon = g_value_get_boolean(value); break;
    case PROP_SPATIAL_OBJECTS: {
      const guint n = gst_value_array_get_size(value);
      if (self->have_state && n != (guint)self->channels) { /* imp.rs:438-450 */
        GST_WARNING_OBJECT(self, "Could not update spatial objects, expected %d channels, got %u", self->channels, n);
        break;
      }
      if (n > HRTF_MAX_CHANNELS) { GST_WARNING_OBJECT(self, "more than %d spatial objects", HRTF_MAX_CHANNELS); break; }
      HrtfObject objs[HRTF_MAX_CHANNELS];
      gboolean ok = TRUE;
      for (guint i = 0; i < n && ok; i++) {
        const GValue *v = gst_value_array_get_value(value, i);
        ok = GST_VALUE_HOLDS_STRUCTURE(v) && gst_hrtf_render_object_from_structure(gst_value_get_structure(v), &objs[i]);
      }
      if (!ok) { GST_WARNING_OBJECT(self, "spatial-objects: every entry needs float x, y, z"); break; }
      for (guint i = 0; i < n; i++) self->objects[i] = objs[i];
      self->n_objects = n; /* an empty array = None (imp.rs:452) */
      break;
    }
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID(object, id, pspec); break;
  }
  g_mutex_unlock(&self->lock);
}

static void gst_hrtf_render_get_property(GObject *object, guint id, GValue *value, GParamSpec *pspec) {
  GstHrtfRender *self = GST_HRTF_RENDER(object);
  g_mutex_lock(&self->lock);
  switch (id) {
    case PROP_HRIR_RAW: g_value_set_boxed(value, self->hrir_raw); break;
    case PROP_HRIR_FILE: g_value_set_string(value, self->hrir_file); break;
    case PROP_INTERPOLATION_STEPS: g_value_set_uint64(value, self->interpolation_steps); break;
    case PROP_BLOCK_LENGTH: g_value_set_uint64(value, self->block_length); break;
    case PROP_USE_RAYON: g_value_set_boolean(value, self->use_rayon); break;
    case PROP_SPATIAL_OBJECTS:
      for (guint i = 0; i < self->n_objects; i++) { /* From<SpatialObject> for gst::Structure (spatial.rs:162-176) */
        GValue v = G_VALUE_INIT;
        g_value_init(&v, GST_TYPE_STRUCTURE);
        GstStructure *s = gst_structure_new("application/spatial-object", "x", G_TYPE_FLOAT, self->objects[i].xyz[0], "y", G_TYPE_FLOAT, self->objects[i].xyz[1],
                                            "z", G_TYPE_FLOAT, self->objects[i].xyz[2], "distance-gain", G_TYPE_FLOAT, self->objects[i].distance_gain,
                                            "coordinate-system", gst_hrtf_coordinate_system_get_type(), self->objects[i].coordinate_system, NULL);
        g_value_take_boxed(&v, s);
        gst_value_array_append_and_take_value(value, &v);
      }
      break;
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID(object, id, pspec); break;
  }
  g_mutex_unlock(&self->lock);
}

static gboolean gst_hrtf_render_start(GstBaseTransform *trans) {
  GstHrtfRender *self = GST_HRTF_RENDER(trans);
  int status = 0;
  self->ctx = mi355_ctx_create(0, &status);
  if (!self->ctx) {
    GST_ELEMENT_ERROR(self, LIBRARY, INIT, ("No MI355X context"), ("%s", mi355_status_string(status)));
    return FALSE;
  }
  return TRUE;
}

/* BaseTransformImpl::stop (imp.rs:748-753) */
static gboolean gst_hrtf_render_stop(GstBaseTransform *trans) {
  GstHrtfRender *self = GST_HRTF_RENDER(trans);
  g_mutex_lock(&self->lock);
  self->have_state = FALSE;
  g_mutex_unlock(&self->lock);
  gst_adapter_clear(self->adapter);
  if (self->ctx) {
    (void)mi355_hrtf_teardown(self->ctx);
    mi355_ctx_destroy(self->ctx);
  }
  self->ctx = NULL;
  return TRUE;
}

/* BaseTransformImpl::transform_caps (imp.rs:602-646) */
static GstCaps *gst_hrtf_render_transform_caps(GstBaseTransform *trans, GstPadDirection direction, GstCaps *caps, GstCaps *filter) {
  GstHrtfRender *self = GST_HRTF_RENDER(trans);
  GstCaps *other = gst_caps_copy(caps);
  for (guint i = 0; i < gst_caps_get_size(other); i++) {
    GstStructure *s = gst_caps_get_structure(other, i);
    gst_structure_set(s, "format", G_TYPE_STRING, GST_AUDIO_NE(F32), "layout", G_TYPE_STRING, "interleaved", NULL);
    if (direction == GST_PAD_SINK) {
      gst_structure_set(s, "channels", G_TYPE_INT, 2, "channel-mask", GST_TYPE_BITMASK, (guint64)0x3, NULL);
    } else {
      g_mutex_lock(&self->lock);
      const guint n = self->n_objects;
      g_mutex_unlock(&self->lock);
      if (n) gst_structure_set(s, "channels", G_TYPE_INT, (gint)n, NULL);
      else gst_structure_set(s, "channels", GST_TYPE_INT_RANGE, 1, G_MAXINT, NULL);
      gst_structure_remove_field(s, "channel-mask");
    }
  }
  if (filter) {
    GstCaps *res = gst_caps_intersect_full(filter, other, GST_CAPS_INTERSECT_FIRST);
    gst_caps_unref(other);
    return res;
  }
  return other;
}

/* BaseTransformImpl::set_caps (imp.rs:648-707) */
static gboolean gst_hrtf_render_set_caps(GstBaseTransform *trans, GstCaps *incaps, GstCaps *outcaps) {
  GstHrtfRender *self = GST_HRTF_RENDER(trans);
  GstAudioInfo in;
  if (!gst_audio_info_from_caps(&in, incaps)) { GST_ERROR_OBJECT(self, "Failed to parse input caps"); return FALSE; }
  const gint channels = GST_AUDIO_INFO_CHANNELS(&in), rate = GST_AUDIO_INFO_RATE(&in);
  gboolean ok = FALSE;
  g_mutex_lock(&self->lock);
  self->have_state = FALSE;
  do {
    if (channels < 1 || channels > HRTF_MAX_CHANNELS) break;
    if (self->n_objects == 0) { /* infer the objects from the channel positions (imp.rs:660-672) */
      if (GST_AUDIO_INFO_IS_UNPOSITIONED(&in)) { GST_ERROR_OBJECT(self, "Cannot infer object positions"); break; }
      gint c;
      for (c = 0; c < channels; c++) {
        self->objects[c].coordinate_system = 1;
        self->objects[c].distance_gain = 1.0f;
        if (mi355host_hrtf_object_from_channel_position((int)GST_AUDIO_INFO_POSITION(&in, c), self->objects[c].xyz) != 0) break;
      }
      if (c < channels) { GST_ERROR_OBJECT(self, "Unsupported channel position"); break; }
      self->n_objects = (guint)channels;
    }
    if (self->n_objects != (guint)channels) { GST_ERROR_OBJECT(self, "Wrong number of spatial objects"); break; }
    /* Settings::sphere (imp.rs:84-94): raw bytes win over the file location */
    gchar *file_bytes = NULL;
    const void *bytes = NULL;
    gsize len = 0;
    if (self->hrir_raw) bytes = g_bytes_get_data(self->hrir_raw, &len);
    else if (self->hrir_file && g_file_get_contents(self->hrir_file, &file_bytes, &len, NULL)) bytes = file_bytes;
    if (!bytes) { GST_ERROR_OBJECT(self, "Failed to load sphere: %s", self->hrir_file ? "cannot read hrir-file" : "Impulse response not set"); break; }
    const int rc = mi355_hrtf_load_sphere(self->ctx, bytes, len, (uint32_t)rate);
    g_free(file_bytes);
    if (rc != MI355_OK) { GST_ERROR_OBJECT(self, "Failed to load sphere: %s", mi355_ctx_last_error(self->ctx)); break; }
    guint64 bs = 0;
    if (!g_uint64_checked_mul(&bs, self->block_length, self->interpolation_steps) || bs == 0 || bs > (1u << 24)) { /* checked_mul (imp.rs:655-657) */
      GST_ERROR_OBJECT(self, "Not enough memory for frame allocation");
      break;
    }
    if (mi355_hrtf_setup(self->ctx, channels, (int)self->block_length, (int)self->interpolation_steps) != MI355_OK) {
      GST_ERROR_OBJECT(self, "mi355_hrtf_setup: %s", mi355_ctx_last_error(self->ctx));
      break;
    }
    self->rate = rate;
    self->channels = channels;
    self->block_samples = (gsize)bs;
    self->have_state = TRUE;
    ok = TRUE;
  } while (0);
  g_mutex_unlock(&self->lock);
  gst_adapter_clear(self->adapter);
  return ok;
}

/* BaseTransformImpl::transform_size (imp.rs:574-599): whole blocks only */
static gboolean gst_hrtf_render_transform_size(GstBaseTransform *trans, GstPadDirection direction, GstCaps *caps, gsize size, GstCaps *othercaps, gsize *othersize) {
  GstHrtfRender *self = GST_HRTF_RENDER(trans);
  if (direction == GST_PAD_SRC || !self->have_state) return FALSE;
  const gsize inblk = self->block_samples * (gsize)self->channels * sizeof(gfloat), outblk = self->block_samples * 2 * sizeof(gfloat);
  *othersize = ((size + gst_adapter_available(self->adapter)) / inblk) * outblk;
  return TRUE;
}

/* renders every complete block in the adapter into `out` (process, imp.rs:186-279); returns the bytes written or -1 */
static gssize gst_hrtf_render_process(GstHrtfRender *self, gfloat *out, gsize out_bytes) {
  gfloat pos[HRTF_MAX_CHANNELS * 3], gains[HRTF_MAX_CHANNELS];
  g_mutex_lock(&self->lock); /* the objects as they are for this buffer (mutable in PLAYING) */
  for (gint c = 0; c < self->channels; c++) {
    (void)mi355host_position_convert(self->objects[c].coordinate_system, 2 /* right-handed: what the hrtf crate wants */, self->objects[c].xyz, &pos[3 * c]);
    gains[c] = self->objects[c].distance_gain;
  }
  g_mutex_unlock(&self->lock);
  const gsize inblk = self->block_samples * (gsize)self->channels * sizeof(gfloat), outblk = self->block_samples * 2 * sizeof(gfloat);
  gsize written = 0;
  while (gst_adapter_available(self->adapter) >= inblk && written + outblk <= out_bytes) {
    const gfloat *in = (const gfloat *)gst_adapter_map(self->adapter, inblk);
    const int rc = mi355_hrtf_process_block(self->ctx, in, out + written / sizeof(gfloat), pos, gains);
    gst_adapter_unmap(self->adapter);
    if (rc != MI355_OK) {
      GST_ERROR_OBJECT(self, "mi355_hrtf_process_block: %s", mi355_ctx_last_error(self->ctx));
      return -1;
    }
    gst_adapter_flush(self->adapter, inblk);
    written += outblk;
  }
  return (gssize)written;
}

/* BaseTransformImpl::transform (imp.rs:554-572) */
static GstFlowReturn gst_hrtf_render_transform(GstBaseTransform *trans, GstBuffer *inbuf, GstBuffer *outbuf) {
  GstHrtfRender *self = GST_HRTF_RENDER(trans);
  if (!self->have_state) return GST_FLOW_NOT_NEGOTIATED;
  gst_adapter_push(self->adapter, gst_buffer_ref(inbuf));
  GstMapInfo map;
  if (!gst_buffer_map(outbuf, &map, GST_MAP_WRITE)) return GST_FLOW_ERROR;
  const gssize n = gst_hrtf_render_process(self, (gfloat *)map.data, map.size);
  gst_buffer_unmap(outbuf, &map);
  if (n < 0) return GST_FLOW_ERROR;
  gst_buffer_set_size(outbuf, n);
  return GST_FLOW_OK;
}

/* drain (imp.rs:281-352): the rest zero-padded to a block, the output cut to the real frame count, tails reset */
static GstFlowReturn gst_hrtf_render_drain(GstHrtfRender *self) {
  if (!self->have_state) return GST_FLOW_OK;
  const gsize avail = gst_adapter_available(self->adapter);
  if (avail == 0) return GST_FLOW_OK;
  const gsize inbpf = (gsize)self->channels * sizeof(gfloat), outbpf = 2 * sizeof(gfloat);
  const gsize inblk = self->block_samples * inbpf, outblk = self->block_samples * outbpf;
  const gsize outputsz = avail / inbpf * outbpf;
  GstBuffer *pad = gst_buffer_new_allocate(NULL, inblk - avail, NULL);
  gst_buffer_memset(pad, 0, 0, inblk - avail);
  gst_adapter_push(self->adapter, pad);
  GstBuffer *out = gst_buffer_new_allocate(NULL, outblk, NULL);
  GstMapInfo map;
  if (!gst_buffer_map(out, &map, GST_MAP_WRITE)) { gst_buffer_unref(out); return GST_FLOW_ERROR; }
  const gssize n = gst_hrtf_render_process(self, (gfloat *)map.data, map.size);
  gst_buffer_unmap(out, &map);
  if (n < 0) { gst_buffer_unref(out); return GST_FLOW_ERROR; }
  gst_buffer_set_size(out, outputsz);
  (void)mi355_hrtf_reset(self->ctx); /* state.reset_processors() */
  return gst_pad_push(GST_BASE_TRANSFORM_SRC_PAD(self), out);
}

/* BaseTransformImpl::sink_event (imp.rs:710-736) */
static gboolean gst_hrtf_render_sink_event(GstBaseTransform *trans, GstEvent *event) {
  GstHrtfRender *self = GST_HRTF_RENDER(trans);
  switch (GST_EVENT_TYPE(event)) {
    case GST_EVENT_FLUSH_STOP:
      gst_adapter_clear(self->adapter);
      if (self->have_state) (void)mi355_hrtf_reset(self->ctx);
      break;
    case GST_EVENT_EOS:
      if (gst_hrtf_render_drain(self) != GST_FLOW_OK) GST_ELEMENT_WARNING(self, CORE, EVENT, ("Failed to drain internal buffer"), (NULL));
      break;
    default: break;
  }
  return GST_BASE_TRANSFORM_CLASS(gst_hrtf_render_parent_class)->sink_event(trans, event);
}

static void gst_hrtf_render_finalize(GObject *object) {
  GstHrtfRender *self = GST_HRTF_RENDER(object);
  if (self->hrir_raw) g_bytes_unref(self->hrir_raw);
  g_free(self->hrir_file);
  g_object_unref(self->adapter);
  g_mutex_clear(&self->lock);
  G_OBJECT_CLASS(gst_hrtf_render_parent_class)->finalize(object);
}

static void gst_hrtf_render_class_init(GstHrtfRenderClass *klass) {
  GObjectClass *gobject = G_OBJECT_CLASS(klass);
  GstElementClass *element = GST_ELEMENT_CLASS(klass);
  GstBaseTransformClass *trans = GST_BASE_TRANSFORM_CLASS(klass);
  gobject->set_property = gst_hrtf_render_set_property;
  gobject->get_property = gst_hrtf_render_get_property;
  gobject->finalize = gst_hrtf_render_finalize;
  const GParamFlags ready = (GParamFlags)(G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS | GST_PARAM_MUTABLE_READY);
  const GParamFlags playing = (GParamFlags)(G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS | GST_PARAM_MUTABLE_PLAYING);
  g_object_class_install_property(gobject, PROP_HRIR_RAW,
      g_param_spec_boxed("hrir-raw", "Head Transform Impulse Response", "Head Transform Impulse Response raw bytes", G_TYPE_BYTES, ready));
  g_object_class_install_property(gobject, PROP_HRIR_FILE,
      g_param_spec_string("hrir-file", "Head Transform Impulse Response", "Head Transform Impulse Response file location to read from", NULL, ready));
  g_object_class_install_property(gobject, PROP_INTERPOLATION_STEPS,
      g_param_spec_uint64("interpolation-steps", "Interpolation Steps", "Interpolation Steps is the amount of slices to cut source to", 0, G_MAXUINT64 - 1, 8, ready));
  g_object_class_install_property(gobject, PROP_BLOCK_LENGTH,
      g_param_spec_uint64("block-length", "Block Length", "Block Length is the length of each slice", 0, G_MAXUINT64 - 1, 512, ready));
  g_object_class_install_property(gobject, PROP_USE_RAYON,
      g_param_spec_boolean("use-rayon", "Use Rayon", "Use Rayon to process input channels in parallel", FALSE, ready));
  g_object_class_install_property(gobject, PROP_SPATIAL_OBJECTS,
      gst_param_spec_array("spatial-objects", "Spatial Objects", "Spatial object Metadata to apply on input channels",
                           g_param_spec_boxed("spatial-object", "Spatial Object", "Spatial Object Metadata", GST_TYPE_STRUCTURE,
                                              (GParamFlags)(G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS)),
                           playing));
  gst_element_class_set_static_metadata(element, "Head-Related Transfer Function (HRTF) renderer", "Filter/Effect/Audio",
                                        "Renders spatial sounds to a given position", "Tomasz Andrzejak <andreiltd@gmail.com>");
  gst_element_class_add_static_pad_template(element, &src_template);
  gst_element_class_add_static_pad_template(element, &sink_template);
  trans->start = gst_hrtf_render_start;
  trans->stop = gst_hrtf_render_stop;
  trans->transform_caps = gst_hrtf_render_transform_caps;
  trans->set_caps = gst_hrtf_render_set_caps;
  trans->transform_size = gst_hrtf_render_transform_size;
  trans->transform = gst_hrtf_render_transform; /* only `transform` installed == BaseTransformMode::NeverInPlace (imp.rs:549-552) */
  trans->sink_event = gst_hrtf_render_sink_event;
  trans->passthrough_on_same_caps = FALSE;
  trans->transform_ip_on_passthrough = FALSE;
  GST_DEBUG_CATEGORY_INIT(gst_hrtf_render_debug, "hrtfrender", 0, "Head-Related Transfer Function Renderer (MI355X)");
}

static void gst_hrtf_render_init(GstHrtfRender *self) {
  g_mutex_init(&self->lock);
  self->interpolation_steps = 8; /* imp.rs:36 */
  self->block_length = 512;      /* imp.rs:37 */
  self->use_rayon = FALSE;
  self->adapter = gst_adapter_new();
}

gboolean gst_hrtf_render_register(GstPlugin *plugin) {
  return gst_element_register(plugin, "hrtfrender", GST_RANK_NONE, GST_TYPE_HRTF_RENDER); /* hrtf/mod.rs */
}
