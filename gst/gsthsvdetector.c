/* gst/gsthsvdetector.c — `hsvdetector` (GType GstHsvDetector), a GstVideoFilter that never works in place, over the mi355fx
 * C ABI. Surface mirrored from the reference (video/hsv/src/hsvdetector/imp.rs): GType name :71-76 and hsvdetector/mod.rs
 * (rank NONE), six gfloat properties mutable in PLAYING :163-218 (defaults :25-31), metadata :323-334, sink formats
 * {RGBx,xRGB,BGRx,xBGR,RGB,BGR} / source formats {RGBA,ARGB,BGRA,ABGR} :336-370 (video_input_formats / video_output_formats),
 * NeverInPlace :373-378, transform_caps swaps the format list and keeps everything else :380-420, transform_frame's 6 x 4
 * format match :423-707 -> mi355_hsvdetect_frame (the per-pixel loop :100-160 runs on the GPU). */
#include "gstmi355common.h"

GST_DEBUG_CATEGORY_STATIC(gst_hsv_detector_debug);
#define GST_CAT_DEFAULT gst_hsv_detector_debug

#define GST_TYPE_HSV_DETECTOR (gst_hsv_detector_get_type())
G_DECLARE_FINAL_TYPE(GstHsvDetector, gst_hsv_detector, GST, HSV_DETECTOR, GstVideoFilter)

struct _GstHsvDetector {
  GstVideoFilter parent;
  GMutex lock; /* settings: set from application threads, snapshotted once per frame (imp.rs:108) */
  mi355_hsvdetect_settings settings;
  mi355_ctx *ctx;
};

G_DEFINE_TYPE(GstHsvDetector, gst_hsv_detector, GST_TYPE_VIDEO_FILTER)

enum { PROP_0, PROP_HUE_REF, PROP_HUE_VAR, PROP_SATURATION_REF, PROP_SATURATION_VAR, PROP_VALUE_REF, PROP_VALUE_VAR };

#define DET_IN_FORMATS "{ RGBx, xRGB, BGRx, xBGR, RGB, BGR }"
#define DET_OUT_FORMATS "{ RGBA, ARGB, BGRA, ABGR }"
static GstStaticPadTemplate sink_template =
    GST_STATIC_PAD_TEMPLATE("sink", GST_PAD_SINK, GST_PAD_ALWAYS, GST_STATIC_CAPS(GST_VIDEO_CAPS_MAKE(DET_IN_FORMATS)));
static GstStaticPadTemplate src_template =
    GST_STATIC_PAD_TEMPLATE("src", GST_PAD_SRC, GST_PAD_ALWAYS, GST_STATIC_CAPS(GST_VIDEO_CAPS_MAKE(DET_OUT_FORMATS)));

static float *gst_hsv_detector_field(GstHsvDetector *self, guint id) {
  switch (id) {
    case PROP_HUE_REF: return &self->settings.hue_ref;
    case PROP_HUE_VAR: return &self->settings.hue_var;
    case PROP_SATURATION_REF: return &self->settings.saturation_ref;
    case PROP_SATURATION_VAR: return &self->settings.saturation_var;
    case PROP_VALUE_REF: return &self->settings.value_ref;
    case PROP_VALUE_VAR: return &self->settings.value_var;
    default: return NULL;
  }
}

static void gst_hsv_detector_set_property(GObject *object, guint id, const GValue *value, GParamSpec *pspec) {
  GstHsvDetector *self = GST_HSV_DETECTOR(object);
  float *f = gst_hsv_detector_field(self, id);
  if (!f) { G_OBJECT_WARN_INVALID_PROPERTY_ID(object, id, pspec); return; }
  g_mutex_lock(&self->lock);
  *f = g_value_get_float(value);
  g_mutex_unlock(&self->lock);
}

static void gst_hsv_detector_get_property(GObject *object, guint id, GValue *value, GParamSpec *pspec) {
  GstHsvDetector *self = GST_HSV_DETECTOR(object);
  float *f = gst_hsv_detector_field(self, id);
  if (!f) { G_OBJECT_WARN_INVALID_PROPERTY_ID(object, id, pspec); return; }
  g_mutex_lock(&self->lock);
  g_value_set_float(value, *f);
  g_mutex_unlock(&self->lock);
}

static gboolean gst_hsv_detector_start(GstBaseTransform *trans) {
  GstHsvDetector *self = GST_HSV_DETECTOR(trans);
  int status = 0;
  self->ctx = mi355_ctx_create(0, &status);
  if (!self->ctx) {
    GST_ELEMENT_ERROR(self, LIBRARY, INIT, ("No MI355X context"), ("%s", mi355_status_string(status)));
    return FALSE;
  }
  return TRUE;
}

static gboolean gst_hsv_detector_stop(GstBaseTransform *trans) {
  GstHsvDetector *self = GST_HSV_DETECTOR(trans);
  if (self->ctx) mi355_ctx_destroy(self->ctx);
  self->ctx = NULL;
  return TRUE;
}

/* BaseTransformImpl::transform_caps (imp.rs:380-420): every structure keeps its fields, `format` becomes the other side's list */
static GstCaps *gst_hsv_detector_transform_caps(GstBaseTransform *trans, GstPadDirection direction, GstCaps *caps, GstCaps *filter) {
  static const gchar *const in_formats[] = {"RGBx", "xRGB", "BGRx", "xBGR", "RGB", "BGR"};
  static const gchar *const out_formats[] = {"RGBA", "ARGB", "BGRA", "ABGR"};
  const gchar *const *names = direction == GST_PAD_SRC ? in_formats : out_formats;
  const guint n_names = direction == GST_PAD_SRC ? G_N_ELEMENTS(in_formats) : G_N_ELEMENTS(out_formats);
  GValue list = G_VALUE_INIT;
  g_value_init(&list, GST_TYPE_LIST);
  for (guint i = 0; i < n_names; i++) {
    GValue v = G_VALUE_INIT;
    g_value_init(&v, G_TYPE_STRING);
    g_value_set_string(&v, names[i]);
    gst_value_list_append_and_take_value(&list, &v);
  }
  GstCaps *other = gst_caps_copy(caps);
  for (guint i = 0; i < gst_caps_get_size(other); i++) gst_structure_set_value(gst_caps_get_structure(other, i), "format", &list);
  g_value_unset(&list);
  if (filter) { /* filter.intersect_with_mode(&other_caps, CapsIntersectMode::First) (imp.rs:413-417) */
    GstCaps *res = gst_caps_intersect_full(filter, other, GST_CAPS_INTERSECT_FIRST);
    gst_caps_unref(other);
    return res;
  }
  return other;
}

static gboolean gst_hsv_detector_propose_allocation(GstBaseTransform *trans, GstQuery *decide_query, GstQuery *query) {
  GstHsvDetector *self = GST_HSV_DETECTOR(trans);
  if (!GST_BASE_TRANSFORM_CLASS(gst_hsv_detector_parent_class)->propose_allocation(trans, decide_query, query)) return FALSE;
  if (self->ctx) {
    (void)gst_mi355_propose_device_pool(trans, query); /* first choice: frames stay in HBM between mi355 elements */
    (void)gst_mi355_propose_pinned_pool(trans, query);
  }
  return TRUE;
}

static gboolean gst_hsv_detector_decide_allocation(GstBaseTransform *trans, GstQuery *query) {
  /* BEFORE chaining up (as gst_color_lut_decide_allocation does for its pinned pool): the parent class adds a system-memory video pool
   * of its own when the query has none, after which "downstream offered no pool" can no longer be seen */
  if (!gst_mi355_decide_device_pool(trans, query)) return FALSE;
  return GST_BASE_TRANSFORM_CLASS(gst_hsv_detector_parent_class)->decide_allocation(trans, query);
}

/* GstBaseTransformClass::transform, in front of GstVideoFilter's (which maps both buffers): input and output in our device
 * memory run mi355_hsvdetect_frames_device on them - no host copy; anything else goes the mapped way (transform_frame below). */
static GstFlowReturn gst_hsv_detector_transform(GstBaseTransform *trans, GstBuffer *inbuf, GstBuffer *outbuf) {
  GstHsvDetector *self = GST_HSV_DETECTOR(trans);
  GstVideoFilter *vf = GST_VIDEO_FILTER(trans);
  mi355_buf *bin = gst_mi355_buffer_peek_device(inbuf), *bout = gst_mi355_buffer_peek_device(outbuf);
  if (!bin || !bout || !vf->negotiated) return GST_BASE_TRANSFORM_CLASS(gst_hsv_detector_parent_class)->transform(trans, inbuf, outbuf);
  mi355_hsvdetect_settings s;
  g_mutex_lock(&self->lock);
  s = self->settings;
  g_mutex_unlock(&self->lock);
  const GstVideoInfo *ii = &vf->in_info, *oi = &vf->out_info;
  const int in_fmt = gst_mi355_format(GST_VIDEO_INFO_FORMAT(ii)), out_fmt = gst_mi355_format(GST_VIDEO_INFO_FORMAT(oi));
  if (in_fmt < 0 || out_fmt < 0) return GST_FLOW_NOT_NEGOTIATED;
  const uint8_t *d_src = mi355_buf_device_ptr(bin, self->ctx, MI355_MAP_READ);
  uint8_t *d_dst = mi355_buf_device_ptr(bout, self->ctx, MI355_MAP_WRITE);
  int rc = d_src && d_dst ? MI355_OK : MI355_ERR_HIP;
  if (rc == MI355_OK)
    rc = mi355_hsvdetect_frames_device(self->ctx, d_src, GST_VIDEO_INFO_SIZE(ii), GST_VIDEO_INFO_PLANE_STRIDE(ii, 0), in_fmt, d_dst, GST_VIDEO_INFO_SIZE(oi),
                                       GST_VIDEO_INFO_PLANE_STRIDE(oi, 0), out_fmt, 1, GST_VIDEO_INFO_WIDTH(ii), GST_VIDEO_INFO_HEIGHT(ii), &s);
  if (rc == MI355_OK) rc = mi355_buf_commit(bin, self->ctx);
  if (rc == MI355_OK) rc = mi355_buf_commit(bout, self->ctx);
  if (rc != MI355_OK) {
    GST_ERROR_OBJECT(self, "mi355_hsvdetect_frames_device: %s", mi355_ctx_last_error(self->ctx));
    return GST_FLOW_ERROR;
  }
  return GST_FLOW_OK;
}

/* VideoFilterImpl::transform_frame (imp.rs:423-707) */
static GstFlowReturn gst_hsv_detector_transform_frame(GstVideoFilter *filter, GstVideoFrame *in, GstVideoFrame *out) {
  GstHsvDetector *self = GST_HSV_DETECTOR(filter);
  mi355_hsvdetect_settings s;
  g_mutex_lock(&self->lock);
  s = self->settings;
  g_mutex_unlock(&self->lock);
  const int in_fmt = gst_mi355_format(GST_VIDEO_FRAME_FORMAT(in)), out_fmt = gst_mi355_format(GST_VIDEO_FRAME_FORMAT(out));
  if (in_fmt < 0 || out_fmt < 0) return GST_FLOW_NOT_NEGOTIATED; /* the reference's match ends in unimplemented!() */
  const int sstride = GST_VIDEO_FRAME_PLANE_STRIDE(in, 0), dstride = GST_VIDEO_FRAME_PLANE_STRIDE(out, 0);
  const size_t h = (size_t)GST_VIDEO_FRAME_HEIGHT(in);
  const int rc = mi355_hsvdetect_frame(self->ctx, GST_VIDEO_FRAME_PLANE_DATA(in, 0), (size_t)sstride * h, sstride, in_fmt,
                                       GST_VIDEO_FRAME_PLANE_DATA(out, 0), (size_t)dstride * h, dstride, out_fmt, GST_VIDEO_FRAME_WIDTH(in), &s);
  if (rc != MI355_OK) {
    GST_ERROR_OBJECT(self, "mi355_hsvdetect_frame: %s", mi355_ctx_last_error(self->ctx));
    return GST_FLOW_ERROR;
  }
  return GST_FLOW_OK;
}

static void gst_hsv_detector_finalize(GObject *object) {
  GstHsvDetector *self = GST_HSV_DETECTOR(object);
  g_mutex_clear(&self->lock);
  G_OBJECT_CLASS(gst_hsv_detector_parent_class)->finalize(object);
}

static void gst_hsv_detector_class_init(GstHsvDetectorClass *klass) {
  GObjectClass *gobject = G_OBJECT_CLASS(klass);
  GstElementClass *element = GST_ELEMENT_CLASS(klass);
  GstBaseTransformClass *trans = GST_BASE_TRANSFORM_CLASS(klass);
  GstVideoFilterClass *vfilter = GST_VIDEO_FILTER_CLASS(klass);
  gobject->set_property = gst_hsv_detector_set_property;
  gobject->get_property = gst_hsv_detector_get_property;
  gobject->finalize = gst_hsv_detector_finalize;
  const GParamFlags f = (GParamFlags)(G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS | GST_PARAM_MUTABLE_PLAYING);
  g_object_class_install_property(gobject, PROP_HUE_REF,
      g_param_spec_float("hue-ref", "Hue reference", "Hue reference in degrees", -G_MAXFLOAT, G_MAXFLOAT, 0.0f, f));
  g_object_class_install_property(gobject, PROP_HUE_VAR,
      g_param_spec_float("hue-var", "Hue variation", "Allowed hue variation from the reference hue angle, in degrees", 0.0f, 180.0f, 10.0f, f));
  g_object_class_install_property(gobject, PROP_SATURATION_REF,
      g_param_spec_float("saturation-ref", "Saturation reference", "Reference saturation value", 0.0f, 1.0f, 0.0f, f));
  g_object_class_install_property(gobject, PROP_SATURATION_VAR,
      g_param_spec_float("saturation-var", "Saturation variation", "Allowed saturation variation from the reference value", 0.0f, 1.0f, 0.15f, f));
  g_object_class_install_property(gobject, PROP_VALUE_REF,
      g_param_spec_float("value-ref", "Value reference", "Reference value value", 0.0f, 1.0f, 0.0f, f));
  g_object_class_install_property(gobject, PROP_VALUE_VAR,
      g_param_spec_float("value-var", "Value variation", "Allowed value variation from the reference value", 0.0f, 1.0f, 0.3f, f));
  gst_element_class_set_static_metadata(element, "HSV detector", "Filter/Effect/Converter/Video",
                                        "Works within the HSV colorspace to mark positive pixels", "Julien Bardagi <julien.bardagi@gmail.com>");
  gst_element_class_add_static_pad_template(element, &sink_template);
  gst_element_class_add_static_pad_template(element, &src_template);
  trans->start = gst_hsv_detector_start;
  trans->stop = gst_hsv_detector_stop;
  trans->transform_caps = gst_hsv_detector_transform_caps;
  trans->propose_allocation = gst_hsv_detector_propose_allocation;
  trans->decide_allocation = gst_hsv_detector_decide_allocation;
  trans->transform = gst_hsv_detector_transform; /* GstVideoFilter's transform is reached by chaining up */
  trans->passthrough_on_same_caps = FALSE;     /* imp.rs:376 */
  trans->transform_ip_on_passthrough = FALSE;  /* imp.rs:377 */
  vfilter->transform_frame = gst_hsv_detector_transform_frame; /* only the non-ip slot == BaseTransformMode::NeverInPlace */
  GST_DEBUG_CATEGORY_INIT(gst_hsv_detector_debug, "hsvdetector", 0, "HSV-based detection filter (MI355X)");
}

static void gst_hsv_detector_init(GstHsvDetector *self) {
  g_mutex_init(&self->lock);
  self->settings.hue_ref = 0.0f;         /* imp.rs:25-31 */
  self->settings.hue_var = 10.0f;
  self->settings.saturation_ref = 0.0f;
  self->settings.saturation_var = 0.15f;
  self->settings.value_ref = 0.0f;
  self->settings.value_var = 0.3f;
}

gboolean gst_hsv_detector_register(GstPlugin *plugin) {
  return gst_element_register(plugin, "hsvdetector", GST_RANK_NONE, GST_TYPE_HSV_DETECTOR); /* hsvdetector/mod.rs */
}
