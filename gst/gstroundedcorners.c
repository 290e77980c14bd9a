/* gst/gstroundedcorners.c — `roundedcorners` (GType GstRoundedCorners), a GstBaseTransform that always works in place, over
 * the host half of mi355fx. Surface mirrored from the reference (video/videofx/src/border/imp.rs): GType name :268-272 and
 * border/mod.rs (rank NONE), property border-radius-px (guint, PLAYING; a change reconfigures the src pad) :276-325 (default
 * :27), metadata :330-341, sink I420 / src I420 + A420 :343-372, AlwaysInPlace :376-379, transform_caps :389-442 (radius 0
 * offers I420 and A420, otherwise A420 only), set_caps :444-480 (I420 out = passthrough; A420 = an alpha memory of
 * stride[3] x round_up_2(height)), prepare_output_buffer :482-559 + add_video_meta :182-262 (the SAME alpha memory is appended
 * to every buffer and the video meta rewritten for four planes), transform_ip does nothing :561-563, propose_allocation
 * :565-572.
 * The reference has no per-buffer pixel loop here: the mask is drawn once per caps / radius change by cairo (four arcs,
 * antialiased fill + 1 px stroke, :57-180). The bytes are defined by cairo's rasteriser, so the mask comes from the same
 * library through mi355host_rounded_corners_mask. The shared alpha memory is DEVICE memory of ours (GstMi355DeviceAllocator) when
 * that allocator can be had: the mask is written once through a WRITE map (host side newer), the first mi355 element downstream
 * that takes its device pointer uploads it once, and every buffer carries a reference to that one memory - as every buffer of the
 * reference carries a reference to `alpha_mem` (:482-559). Anything that maps it gets the same bytes. (SURVEY.md §8 a8;
 * contiguous A420 in HBM: mi355_roundedcorners_append_device, csrc/roundedcorners.hip.) */
#include <gst/gst.h>
#include <gst/base/gstbasetransform.h>
#include <gst/video/video.h>
#include "../gst-plugins-rs_amd/host/mi355fx_host.h"
#include "gstmi355common.h"

GST_DEBUG_CATEGORY_STATIC(gst_rounded_corners_debug);
#define GST_CAT_DEFAULT gst_rounded_corners_debug

#define GST_TYPE_ROUNDED_CORNERS (gst_rounded_corners_get_type())
G_DECLARE_FINAL_TYPE(GstRoundedCorners, gst_rounded_corners, GST, ROUNDED_CORNERS, GstBaseTransform)

struct _GstRoundedCorners {
  GstBaseTransform parent;
  GMutex lock; /* settings + state */
  guint border_radius_px;
  gboolean changed;
  gboolean have_state;
  GstVideoInfo out_info;
  GstMemory *alpha_mem;
};

G_DEFINE_TYPE(GstRoundedCorners, gst_rounded_corners, GST_TYPE_BASE_TRANSFORM)

enum { PROP_0, PROP_BORDER_RADIUS_PX };

static GstStaticPadTemplate sink_template = GST_STATIC_PAD_TEMPLATE("sink", GST_PAD_SINK, GST_PAD_ALWAYS, GST_STATIC_CAPS(GST_VIDEO_CAPS_MAKE("I420")));
static GstStaticPadTemplate src_template = GST_STATIC_PAD_TEMPLATE("src", GST_PAD_SRC, GST_PAD_ALWAYS, GST_STATIC_CAPS(GST_VIDEO_CAPS_MAKE("{ I420, A420 }")));

static void gst_rounded_corners_set_property(GObject *object, guint id, const GValue *value, GParamSpec *pspec) {
  GstRoundedCorners *self = GST_ROUNDED_CORNERS(object);
  if (id != PROP_BORDER_RADIUS_PX) { G_OBJECT_WARN_INVALID_PROPERTY_ID(object, id, pspec); return; }
  g_mutex_lock(&self->lock);
  const guint r = g_value_get_uint(value);
  const gboolean differs = self->border_radius_px != r;
  if (differs) { /* imp.rs:297-310 */
    self->changed = TRUE;
    GST_INFO_OBJECT(self, "Changing border radius from %u to %u", self->border_radius_px, r);
    self->border_radius_px = r;
  }
  g_mutex_unlock(&self->lock);
  if (differs) gst_base_transform_reconfigure_src(GST_BASE_TRANSFORM(self));
}

static void gst_rounded_corners_get_property(GObject *object, guint id, GValue *value, GParamSpec *pspec) {
  GstRoundedCorners *self = GST_ROUNDED_CORNERS(object);
  if (id != PROP_BORDER_RADIUS_PX) { G_OBJECT_WARN_INVALID_PROPERTY_ID(object, id, pspec); return; }
  g_mutex_lock(&self->lock);
  g_value_set_uint(value, self->border_radius_px);
  g_mutex_unlock(&self->lock);
}

/* BaseTransformImpl::stop (imp.rs:381-387) */
static gboolean gst_rounded_corners_stop(GstBaseTransform *trans) {
  GstRoundedCorners *self = GST_ROUNDED_CORNERS(trans);
  g_mutex_lock(&self->lock);
  self->have_state = FALSE;
  if (self->alpha_mem) gst_memory_unref(self->alpha_mem);
  self->alpha_mem = NULL;
  g_mutex_unlock(&self->lock);
  return TRUE;
}

/* BaseTransformImpl::transform_caps (imp.rs:389-442) */
static GstCaps *gst_rounded_corners_transform_caps(GstBaseTransform *trans, GstPadDirection direction, GstCaps *caps, GstCaps *filter) {
  GstRoundedCorners *self = GST_ROUNDED_CORNERS(trans);
  GstCaps *other = gst_caps_copy(caps);
  if (direction == GST_PAD_SRC) {
    for (guint i = 0; i < gst_caps_get_size(other); i++) gst_structure_set(gst_caps_get_structure(other, i), "format", G_TYPE_STRING, "I420", NULL);
  } else {
    g_mutex_lock(&self->lock);
    const guint radius = self->border_radius_px;
    g_mutex_unlock(&self->lock);
    for (guint i = 0; i < gst_caps_get_size(other); i++) {
      GstStructure *s = gst_caps_get_structure(other, i);
      if (radius == 0) {
        GValue list = G_VALUE_INIT;
        g_value_init(&list, GST_TYPE_LIST);
        static const gchar *const names[] = {"I420", "A420"};
        for (guint k = 0; k < G_N_ELEMENTS(names); k++) {
          GValue v = G_VALUE_INIT;
          g_value_init(&v, G_TYPE_STRING);
          g_value_set_string(&v, names[k]);
          gst_value_list_append_and_take_value(&list, &v);
        }
        gst_structure_set_value(s, "format", &list);
        g_value_unset(&list);
      } else {
        gst_structure_set(s, "format", G_TYPE_STRING, "A420", NULL);
      }
    }
  }
  if (filter) {
    GstCaps *res = gst_caps_intersect_full(filter, other, GST_CAPS_INTERSECT_FIRST);
    gst_caps_unref(other);
    return res;
  }
  return other;
}

/* BaseTransformImpl::set_caps (imp.rs:444-480) */
static gboolean gst_rounded_corners_set_caps(GstBaseTransform *trans, GstCaps *incaps, GstCaps *outcaps) {
  GstRoundedCorners *self = GST_ROUNDED_CORNERS(trans);
  GstVideoInfo out_info;
  if (!gst_video_info_from_caps(&out_info, outcaps)) { GST_ERROR_OBJECT(self, "Failed to parse output caps"); return FALSE; }
  if (GST_VIDEO_INFO_FORMAT(&out_info) == GST_VIDEO_FORMAT_I420) {
    gst_base_transform_set_passthrough(trans, TRUE);
    return TRUE;
  }
  gst_base_transform_set_passthrough(trans, FALSE);
  const guint ru2_height = ((guint)GST_VIDEO_INFO_HEIGHT(&out_info) + 1u) & ~1u;
  const gsize alpha_mem_size = (gsize)GST_VIDEO_INFO_PLANE_STRIDE(&out_info, 3) * ru2_height;
  g_mutex_lock(&self->lock);
  if (self->alpha_mem) gst_memory_unref(self->alpha_mem);
  {
    GstAllocator *dev = gst_mi355_device_allocator_new(); /* NULL without an MI355X: plain system memory then, as the reference */
    GstAllocationParams params;
    gst_allocation_params_init(&params);
    self->alpha_mem = dev ? gst_allocator_alloc(dev, alpha_mem_size, &params) : NULL;
    if (dev) gst_object_unref(dev); /* the memory keeps its allocator alive */
    if (!self->alpha_mem) self->alpha_mem = gst_allocator_alloc(NULL, alpha_mem_size, NULL);
  }
  self->out_info = out_info;
  self->have_state = self->alpha_mem != NULL;
  self->changed = TRUE;
  g_mutex_unlock(&self->lock);
  return self->alpha_mem != NULL;
}

/* generate_alpha_mask (imp.rs:57-180); `lock` held */
static gboolean gst_rounded_corners_generate_alpha_mask(GstRoundedCorners *self) {
  GstMapInfo map;
  if (!gst_memory_map(self->alpha_mem, &map, GST_MAP_WRITE)) return FALSE;
  char err[256] = "";
  const int rc = mi355host_rounded_corners_mask(map.data, GST_VIDEO_INFO_WIDTH(&self->out_info), GST_VIDEO_INFO_HEIGHT(&self->out_info),
                                                GST_VIDEO_INFO_PLANE_STRIDE(&self->out_info, 3), self->border_radius_px, err, sizeof err);
  gst_memory_unmap(self->alpha_mem, &map);
  if (rc != 0) GST_ERROR_OBJECT(self, "%s", err);
  return rc == 0;
}

/* add_video_meta (imp.rs:182-262): four planes, the alpha plane where the appended memory starts */
static void gst_rounded_corners_add_video_meta(GstBuffer *buf, const GstVideoInfo *out_info, gsize alpha_plane_offset) {
  gint strides[GST_VIDEO_MAX_PLANES] = {0, 0, 0, 0};
  gsize offsets[GST_VIDEO_MAX_PLANES] = {0, 0, 0, 0};
  GstVideoFrameFlags flags = GST_VIDEO_FRAME_FLAG_NONE;
  GstVideoMeta *meta = gst_buffer_get_video_meta(buf);
  if (meta) {
    flags = meta->flags;
    for (guint p = 0; p < meta->n_planes && p < GST_VIDEO_MAX_PLANES; p++) { offsets[p] = meta->offset[p]; strides[p] = meta->stride[p]; }
    offsets[3] = alpha_plane_offset;
    strides[3] = GST_VIDEO_INFO_PLANE_STRIDE(out_info, 3);
    (void)gst_buffer_remove_meta(buf, (GstMeta *)meta);
  } else {
    for (guint p = 0; p < GST_VIDEO_INFO_N_PLANES(out_info); p++) { offsets[p] = GST_VIDEO_INFO_PLANE_OFFSET(out_info, p); strides[p] = GST_VIDEO_INFO_PLANE_STRIDE(out_info, p); }
  }
  (void)gst_buffer_add_video_meta_full(buf, flags, GST_VIDEO_INFO_FORMAT(out_info), GST_VIDEO_INFO_WIDTH(out_info), GST_VIDEO_INFO_HEIGHT(out_info),
                                       GST_VIDEO_INFO_N_PLANES(out_info), offsets, strides);
}

/* BaseTransformImpl::prepare_output_buffer (imp.rs:482-559) */
static GstFlowReturn gst_rounded_corners_prepare_output_buffer(GstBaseTransform *trans, GstBuffer *inbuf, GstBuffer **outbuf) {
  GstRoundedCorners *self = GST_ROUNDED_CORNERS(trans);
  if (gst_base_transform_is_passthrough(trans)) { *outbuf = inbuf; return GST_FLOW_OK; }
  g_mutex_lock(&self->lock);
  if (!self->have_state) {
    g_mutex_unlock(&self->lock);
    GST_ELEMENT_ERROR(self, CORE, NEGOTIATION, ("Have no state yet"), (NULL));
    return GST_FLOW_NOT_NEGOTIATED;
  }
  if (self->changed) {
    self->changed = FALSE;
    GST_DEBUG_OBJECT(self, "Caps or border radius changed, generating alpha mask");
    if (!gst_rounded_corners_generate_alpha_mask(self)) {
      g_mutex_unlock(&self->lock);
      GST_ELEMENT_ERROR(self, CORE, NEGOTIATION, ("Failed to generate alpha mask"), (NULL));
      return GST_FLOW_NOT_NEGOTIATED;
    }
  }
  GstMemory *alpha = gst_memory_ref(self->alpha_mem);
  const GstVideoInfo out_info = self->out_info;
  g_mutex_unlock(&self->lock);
  GstBuffer *buf = gst_buffer_is_writable(inbuf) ? inbuf : gst_buffer_copy(inbuf); /* InputBuffer::Writable / ::Readable */
  const gsize alpha_plane_offset = gst_buffer_get_size(buf);
  gst_buffer_append_memory(buf, alpha);
  gst_rounded_corners_add_video_meta(buf, &out_info, alpha_plane_offset);
  *outbuf = buf;
  return GST_FLOW_OK;
}

static GstFlowReturn gst_rounded_corners_transform_ip(GstBaseTransform *trans, GstBuffer *buf) { return GST_FLOW_OK; } /* imp.rs:561-563 */

static gboolean gst_rounded_corners_propose_allocation(GstBaseTransform *trans, GstQuery *decide_query, GstQuery *query) {
  gst_query_add_allocation_meta(query, GST_VIDEO_META_API_TYPE, NULL); /* imp.rs:570 */
  return GST_BASE_TRANSFORM_CLASS(gst_rounded_corners_parent_class)->propose_allocation(trans, decide_query, query);
}

static void gst_rounded_corners_finalize(GObject *object) {
  GstRoundedCorners *self = GST_ROUNDED_CORNERS(object);
  if (self->alpha_mem) gst_memory_unref(self->alpha_mem);
  g_mutex_clear(&self->lock);
  G_OBJECT_CLASS(gst_rounded_corners_parent_class)->finalize(object);
}

static void gst_rounded_corners_class_init(GstRoundedCornersClass *klass) {
  GObjectClass *gobject = G_OBJECT_CLASS(klass);
  GstElementClass *element = GST_ELEMENT_CLASS(klass);
  GstBaseTransformClass *trans = GST_BASE_TRANSFORM_CLASS(klass);
  gobject->set_property = gst_rounded_corners_set_property;
  gobject->get_property = gst_rounded_corners_get_property;
  gobject->finalize = gst_rounded_corners_finalize;
  g_object_class_install_property(gobject, PROP_BORDER_RADIUS_PX,
      g_param_spec_uint("border-radius-px", "Border radius in pixels", "Draw rounded corners with given border radius", 0, G_MAXUINT, 0,
                        (GParamFlags)(G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS | GST_PARAM_MUTABLE_PLAYING)));
  gst_element_class_set_static_metadata(element, "Rounded Corners", "Filter/Effect/Converter/Video", "Adds rounded corners to video",
                                        "Sanchayan Maity <sanchayan@asymptotic.io>");
  gst_element_class_add_static_pad_template(element, &sink_template);
  gst_element_class_add_static_pad_template(element, &src_template);
  trans->stop = gst_rounded_corners_stop;
  trans->transform_caps = gst_rounded_corners_transform_caps;
  trans->set_caps = gst_rounded_corners_set_caps;
  trans->prepare_output_buffer = gst_rounded_corners_prepare_output_buffer;
  trans->transform_ip = gst_rounded_corners_transform_ip; /* AlwaysInPlace (imp.rs:376-379) */
  trans->propose_allocation = gst_rounded_corners_propose_allocation;
  trans->passthrough_on_same_caps = FALSE;
  trans->transform_ip_on_passthrough = FALSE;
  GST_DEBUG_CATEGORY_INIT(gst_rounded_corners_debug, "roundedcorners", 0, "Rounded corners (MI355X host half)");
}

static void gst_rounded_corners_init(GstRoundedCorners *self) {
  g_mutex_init(&self->lock);
  self->border_radius_px = 0; /* DEFAULT_BORDER_RADIUS (imp.rs:27) */
}

gboolean gst_rounded_corners_register(GstPlugin *plugin) {
  return gst_element_register(plugin, "roundedcorners", GST_RANK_NONE, GST_TYPE_ROUNDED_CORNERS); /* border/mod.rs */
}
