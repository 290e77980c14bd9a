/* gst/gsthsvfilter.c — `hsvfilter` (GType GstHsvFilter), a GstVideoFilter working in place, over the mi355fx C ABI.
 * Surface mirrored from the reference (video/hsv/src/hsvfilter/imp.rs): factory / GType names :67-72 and
 * hsvfilter/mod.rs:20-27 (rank NONE), five gfloat properties with the full float range, mutable in PLAYING :122-160,
 * metadata :263-268, caps {RGBx,xRGB,BGRx,xBGR,RGBA,ARGB,BGRA,ABGR,RGB,BGR} :274-311, AlwaysInPlace / no passthrough
 * :315-320, transform_frame_ip :323-376 -> mi355_hsvfilter_frame_ip (the per-pixel loop :76-120 runs on the GPU).
 * Added by the shim: propose_allocation offers pinned buffers (gst_mi355_propose_pinned_pool); and when the DIRECT
 * downstream peer is this shim's colorlut on RGBA frames (custom query "mi355-fuse-hsv"), the element leaves the pixels
 * alone and tags the buffer with its settings snapshot (GstMi355HsvMeta): colorlut then runs the fused
 * hsvfilter -> colorlut kernel - one upload, one launch, one download for the pair instead of two PCIe round trips. */
#include "gstmi355common.h"

GST_DEBUG_CATEGORY_STATIC(gst_hsv_filter_debug);
#define GST_CAT_DEFAULT gst_hsv_filter_debug

#define GST_TYPE_HSV_FILTER (gst_hsv_filter_get_type())
G_DECLARE_FINAL_TYPE(GstHsvFilter, gst_hsv_filter, GST, HSV_FILTER, GstVideoFilter)

struct _GstHsvFilter {
  GstVideoFilter parent;
  GMutex lock;                /* settings: set from application threads, snapshotted once per frame (imp.rs:85) */
  mi355_hsv_settings settings;
  mi355_ctx *ctx;             /* created in start(), destroyed in stop() */
  gboolean fuse_checked;      /* the peer has been asked since the last (re)negotiation */
  gboolean fuse;              /* downstream is our colorlut: defer the filter to its fused kernel */
};

G_DEFINE_TYPE(GstHsvFilter, gst_hsv_filter, GST_TYPE_VIDEO_FILTER)

enum { PROP_0, PROP_HUE_SHIFT, PROP_SATURATION_MUL, PROP_SATURATION_OFF, PROP_VALUE_MUL, PROP_VALUE_OFF };

#define HSV_FORMATS "{ RGBx, xRGB, BGRx, xBGR, RGBA, ARGB, BGRA, ABGR, RGB, BGR }"
static GstStaticPadTemplate sink_template =
    GST_STATIC_PAD_TEMPLATE("sink", GST_PAD_SINK, GST_PAD_ALWAYS, GST_STATIC_CAPS(GST_VIDEO_CAPS_MAKE(HSV_FORMATS)));
static GstStaticPadTemplate src_template =
    GST_STATIC_PAD_TEMPLATE("src", GST_PAD_SRC, GST_PAD_ALWAYS, GST_STATIC_CAPS(GST_VIDEO_CAPS_MAKE(HSV_FORMATS)));

static void gst_hsv_filter_set_property(GObject *object, guint id, const GValue *value, GParamSpec *pspec) {
  GstHsvFilter *self = GST_HSV_FILTER(object);
  g_mutex_lock(&self->lock);
  switch (id) {
    case PROP_HUE_SHIFT: self->settings.hue_shift = g_value_get_float(value); break;
    case PROP_SATURATION_MUL: self->settings.saturation_mul = g_value_get_float(value); break;
    case PROP_SATURATION_OFF: self->settings.saturation_off = g_value_get_float(value); break;
    case PROP_VALUE_MUL: self->settings.value_mul = g_value_get_float(value); break;
    case PROP_VALUE_OFF: self->settings.value_off = g_value_get_float(value); break;
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID(object, id, pspec); break;
  }
  g_mutex_unlock(&self->lock);
}

static void gst_hsv_filter_get_property(GObject *object, guint id, GValue *value, GParamSpec *pspec) {
  GstHsvFilter *self = GST_HSV_FILTER(object);
  g_mutex_lock(&self->lock);
  switch (id) {
    case PROP_HUE_SHIFT: g_value_set_float(value, self->settings.hue_shift); break;
    case PROP_SATURATION_MUL: g_value_set_float(value, self->settings.saturation_mul); break;
    case PROP_SATURATION_OFF: g_value_set_float(value, self->settings.saturation_off); break;
    case PROP_VALUE_MUL: g_value_set_float(value, self->settings.value_mul); break;
    case PROP_VALUE_OFF: g_value_set_float(value, self->settings.value_off); break;
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID(object, id, pspec); break;
  }
  g_mutex_unlock(&self->lock);
}

static gboolean gst_hsv_filter_start(GstBaseTransform *trans) {
  GstHsvFilter *self = GST_HSV_FILTER(trans);
  int status = 0;
  self->ctx = mi355_ctx_create(0, &status); /* device choice: HIP_VISIBLE_DEVICES, one process per GPU */
  if (!self->ctx) {
    GST_ELEMENT_ERROR(self, LIBRARY, INIT, ("No MI355X context"), ("%s", mi355_status_string(status)));
    return FALSE;
  }
  return TRUE;
}

static gboolean gst_hsv_filter_stop(GstBaseTransform *trans) {
  GstHsvFilter *self = GST_HSV_FILTER(trans);
  if (self->ctx) mi355_ctx_destroy(self->ctx);
  self->ctx = NULL;
  return TRUE;
}

static gboolean gst_hsv_filter_propose_allocation(GstBaseTransform *trans, GstQuery *decide_query, GstQuery *query) {
  GstHsvFilter *self = GST_HSV_FILTER(trans);
  if (!GST_BASE_TRANSFORM_CLASS(gst_hsv_filter_parent_class)->propose_allocation(trans, decide_query, query)) return FALSE;
  if (self->ctx) (void)gst_mi355_propose_pinned_pool(trans, query);
  return TRUE;
}

/* (re)negotiation: ask the peer again on the next frame */
static gboolean gst_hsv_filter_set_info(GstVideoFilter *filter, GstCaps *incaps, GstVideoInfo *in_info, GstCaps *outcaps, GstVideoInfo *out_info) {
  GstHsvFilter *self = GST_HSV_FILTER(filter);
  self->fuse_checked = FALSE;
  self->fuse = FALSE;
  return TRUE;
}

static gboolean gst_hsv_filter_peer_fuses(GstHsvFilter *self) {
  if (g_getenv("MI355_GST_NO_FUSE")) return FALSE;
  GstQuery *q = gst_query_new_custom(GST_QUERY_CUSTOM, gst_structure_new_empty(GST_MI355_FUSE_QUERY_NAME));
  /* gst_pad_peer_query goes to the pad linked to our source pad, nobody further: only a direct colorlut neighbour answers */
  const gboolean ok = gst_pad_peer_query(GST_BASE_TRANSFORM_SRC_PAD(self), q);
  gst_query_unref(q);
  return ok;
}

/* VideoFilterImpl::transform_frame_ip (imp.rs:323-376): the base class hands a writable frame (AlwaysInPlace). */
static GstFlowReturn gst_hsv_filter_transform_frame_ip(GstVideoFilter *filter, GstVideoFrame *frame) {
  GstHsvFilter *self = GST_HSV_FILTER(filter);
  mi355_hsv_settings s;
  g_mutex_lock(&self->lock);
  s = self->settings;
  g_mutex_unlock(&self->lock);
  const int fmt = gst_mi355_format(GST_VIDEO_FRAME_FORMAT(frame));
  if (fmt < 0) return GST_FLOW_NOT_NEGOTIATED; /* the reference's match ends in unreachable!() (imp.rs:374) */
  if (!self->fuse_checked) {
    self->fuse = fmt == MI355_FMT_RGBA && gst_hsv_filter_peer_fuses(self);
    self->fuse_checked = TRUE;
    GST_INFO_OBJECT(self, "hsvfilter %s", self->fuse ? "deferred to the downstream colorlut (fused kernel)" : "runs its own kernel");
  }
  if (self->fuse) {
    /* the settings of THIS frame travel with it; colorlut applies them in the fused launch */
    return gst_buffer_add_mi355_hsv_meta(frame->buffer, &s) ? GST_FLOW_OK : GST_FLOW_ERROR;
  }
  guint8 *data = GST_VIDEO_FRAME_PLANE_DATA(frame, 0);
  const int stride = GST_VIDEO_FRAME_PLANE_STRIDE(frame, 0);
  /* plane_data().len() of the reference = the mapped plane: stride x height rows (imp.rs:87-97 walks chunks of `stride`) */
  const size_t len = (size_t)stride * (size_t)GST_VIDEO_FRAME_HEIGHT(frame);
  const int rc = mi355_hsvfilter_frame_ip(self->ctx, data, len, GST_VIDEO_FRAME_WIDTH(frame), stride, fmt, &s);
  if (rc != MI355_OK) {
    GST_ERROR_OBJECT(self, "mi355_hsvfilter_frame_ip: %s", mi355_ctx_last_error(self->ctx));
    return GST_FLOW_ERROR;
  }
  return GST_FLOW_OK;
}

static void gst_hsv_filter_finalize(GObject *object) {
  GstHsvFilter *self = GST_HSV_FILTER(object);
  g_mutex_clear(&self->lock);
  G_OBJECT_CLASS(gst_hsv_filter_parent_class)->finalize(object);
}

static void gst_hsv_filter_class_init(GstHsvFilterClass *klass) {
  GObjectClass *gobject = G_OBJECT_CLASS(klass);
  GstElementClass *element = GST_ELEMENT_CLASS(klass);
  GstBaseTransformClass *trans = GST_BASE_TRANSFORM_CLASS(klass);
  GstVideoFilterClass *vfilter = GST_VIDEO_FILTER_CLASS(klass);
  gobject->set_property = gst_hsv_filter_set_property;
  gobject->get_property = gst_hsv_filter_get_property;
  gobject->finalize = gst_hsv_filter_finalize;
  const GParamFlags f = (GParamFlags)(G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS | GST_PARAM_MUTABLE_PLAYING);
  g_object_class_install_property(gobject, PROP_HUE_SHIFT,
      g_param_spec_float("hue-shift", "Hue shift", "Hue shifting in degrees", -G_MAXFLOAT, G_MAXFLOAT, 0.0f, f));
  g_object_class_install_property(gobject, PROP_SATURATION_MUL,
      g_param_spec_float("saturation-mul", "Saturation multiplier", "Saturation multiplier to apply to the saturation value (before offset)",
                         -G_MAXFLOAT, G_MAXFLOAT, 1.0f, f));
  g_object_class_install_property(gobject, PROP_SATURATION_OFF,
      g_param_spec_float("saturation-off", "Saturation offset", "Saturation offset to add to the saturation value (after multiplier)",
                         -G_MAXFLOAT, G_MAXFLOAT, 0.0f, f));
  g_object_class_install_property(gobject, PROP_VALUE_MUL,
      g_param_spec_float("value-mul", "Value multiplier", "Value multiplier to apply to the value (before offset)", -G_MAXFLOAT, G_MAXFLOAT, 1.0f, f));
  g_object_class_install_property(gobject, PROP_VALUE_OFF,
      g_param_spec_float("value-off", "Value offset", "Value offset to add to the value (after multiplier)", -G_MAXFLOAT, G_MAXFLOAT, 0.0f, f));
  gst_element_class_set_static_metadata(element, "HSV filter", "Filter/Effect/Converter/Video",
                                        "Works within the HSV colorspace to apply transformations to incoming frames",
                                        "Julien Bardagi <julien.bardagi@gmail.com>");
  gst_element_class_add_static_pad_template(element, &sink_template);
  gst_element_class_add_static_pad_template(element, &src_template);
  trans->start = gst_hsv_filter_start;
  trans->stop = gst_hsv_filter_stop;
  trans->propose_allocation = gst_hsv_filter_propose_allocation;
  trans->passthrough_on_same_caps = FALSE;     /* imp.rs:318 */
  trans->transform_ip_on_passthrough = FALSE;  /* imp.rs:319 */
  vfilter->set_info = gst_hsv_filter_set_info;
  vfilter->transform_frame_ip = gst_hsv_filter_transform_frame_ip; /* only _ip installed == BaseTransformMode::AlwaysInPlace */
  GST_DEBUG_CATEGORY_INIT(gst_hsv_filter_debug, "hsvfilter", 0, "HSV transformation filter (MI355X)");
}

static void gst_hsv_filter_init(GstHsvFilter *self) {
  g_mutex_init(&self->lock);
  self->settings.hue_shift = 0.0f;       /* imp.rs:25-29 */
  self->settings.saturation_mul = 1.0f;
  self->settings.saturation_off = 0.0f;
  self->settings.value_mul = 1.0f;
  self->settings.value_off = 0.0f;
}

gboolean gst_hsv_filter_register(GstPlugin *plugin) {
  return gst_element_register(plugin, "hsvfilter", GST_RANK_NONE, GST_TYPE_HSV_FILTER); /* hsvfilter/mod.rs:20-27 */
}
