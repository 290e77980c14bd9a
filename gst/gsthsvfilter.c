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
  gint fuse_checked;          /* the peer has been asked since the last (re)negotiation / relink / reconfigure (atomic) */
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
  g_atomic_int_set(&self->fuse_checked, FALSE);
  return TRUE;
}

/* the source pad was linked to something else / unlinked, or downstream asked for a reconfigure: whoever answered before
 * may not be the neighbour any more */
static void gst_hsv_filter_src_relinked(GstPad *pad, GstPad *peer, gpointer user_data) {
  g_atomic_int_set(&GST_HSV_FILTER(user_data)->fuse_checked, FALSE);
}
static gboolean gst_hsv_filter_src_event(GstBaseTransform *trans, GstEvent *event) {
  if (GST_EVENT_TYPE(event) == GST_EVENT_RECONFIGURE) g_atomic_int_set(&GST_HSV_FILTER(trans)->fuse_checked, FALSE);
  return GST_BASE_TRANSFORM_CLASS(gst_hsv_filter_parent_class)->src_event(trans, event);
}

/* Is the element that OWNS the pad linked to our source pad a colorlut of this shim that will fuse? A custom query travels:
 * gst_pad_query_default and GstBaseTransform forward queries they do not know, tee answers TRUE when any branch does - so
 * "somebody answered" proves nothing about the neighbour (hsvfilter ! tee, hsvfilter ! videoscale ! colorlut would then pass
 * unfiltered frames to elements that are not colorlut). The question therefore carries no trust in reachability: the
 * answering element writes its own address into the query, and it must be the parent of our peer pad. A ghost pad's
 * parent is its bin: no fusion across bin boundaries. */
static gboolean gst_hsv_filter_peer_fuses(GstHsvFilter *self) {
  if (g_getenv("MI355_GST_NO_FUSE")) return FALSE;
  GstPad *peer = gst_pad_get_peer(GST_BASE_TRANSFORM_SRC_PAD(self));
  if (!peer) return FALSE;
  GstObject *owner = gst_pad_get_parent(peer);
  gboolean ok = FALSE;
  if (owner && GST_IS_ELEMENT(owner)) {
    GstQuery *q = gst_query_new_custom(GST_QUERY_CUSTOM, gst_structure_new(GST_MI355_FUSE_QUERY_NAME, GST_MI355_FUSE_QUERY_WHO, G_TYPE_POINTER, NULL, NULL));
    if (gst_pad_query(peer, q)) {
      gpointer who = NULL;
      if (gst_structure_get(gst_query_get_structure(q), GST_MI355_FUSE_QUERY_WHO, G_TYPE_POINTER, &who, NULL)) ok = who != NULL && who == (gpointer)owner;
    }
    gst_query_unref(q);
  }
  if (owner) gst_object_unref(owner);
  gst_object_unref(peer);
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
  if (!g_atomic_int_get(&self->fuse_checked)) {
    g_atomic_int_set(&self->fuse_checked, TRUE);  /* (set first: a relink while we ask clears it again) */
    self->fuse = fmt == MI355_FMT_RGBA && gst_hsv_filter_peer_fuses(self);
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
  trans->src_event = gst_hsv_filter_src_event;
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
  g_signal_connect(GST_BASE_TRANSFORM_SRC_PAD(self), "linked", G_CALLBACK(gst_hsv_filter_src_relinked), self);
  g_signal_connect(GST_BASE_TRANSFORM_SRC_PAD(self), "unlinked", G_CALLBACK(gst_hsv_filter_src_relinked), self);
}

gboolean gst_hsv_filter_register(GstPlugin *plugin) {
  return gst_element_register(plugin, "hsvfilter", GST_RANK_NONE, GST_TYPE_HSV_FILTER); /* hsvfilter/mod.rs:20-27 */
}
