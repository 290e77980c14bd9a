/* gst/gstvideocompare.c — `videocompare` (GType GstVideoCompare), a GstVideoAggregator, over the mi355fx C ABI. Surface
 * mirrored from the reference (video/videofx/src/videocompare/imp.rs): GType name :63-69 and videocompare/mod.rs (rank NONE;
 * enum GstVideoCompareHashAlgorithm :60-95; the "videocompare" / "pad-distance" structures :104-175), properties hash-algo
 * and max-dist-threshold (READY) :75-143 (defaults :38-57), metadata :145-156, request sink pads sink_%u and the src pad, RGB
 * / RGBA :158-186, the first sink pad created is the reference :210-233, release_pad picks another one :188-207,
 * update_src_caps = the reference pad's caps :235-256, aggregate_frames :259-388: the reference frame is passed through,
 * hashed, every other pad's frame is hashed and compared, and one element message is posted when any distance is within the
 * threshold. hasher.hash_image / compare (hashed_image.rs:27-79: image_hasher's five algorithms on 8x8 bits, or Dssim) are
 * mi355_videocompare_hash_frame + mi355_videocompare_distance and mi355_dssim_create_image + mi355_dssim_compare_frames -
 * with Dssim all pads of one aggregate go to the device in ONE call. */
#include <gst/gst.h>
#include <gst/video/video.h>
#include <gst/video/gstvideoaggregator.h>
#include <stdlib.h>
#include "gstmi355common.h"

GST_DEBUG_CATEGORY_STATIC(gst_video_compare_debug);
#define GST_CAT_DEFAULT gst_video_compare_debug

#define GST_TYPE_VIDEO_COMPARE (gst_video_compare_get_type())
G_DECLARE_FINAL_TYPE(GstVideoCompare, gst_video_compare, GST, VIDEO_COMPARE, GstVideoAggregator)

#define VIDEO_COMPARE_MAX_PADS 64

struct _GstVideoCompare {
  GstVideoAggregator parent;
  GMutex lock; /* settings + reference pad */
  gint hash_algo;
  gdouble max_dist_threshold;
  GstPad *reference_pad; /* not owned: cleared in release_pad */
  mi355_ctx *ctx;
  /* MI355_GROUP_MEMBERS=n: the n videocompare instances of this process hand their (reference, pad) pairs to the device's
   * dispatcher (mi355_group_shared -> mi355_group_submit_compare): the pairs of an interval are ONE launch sequence */
  mi355_group *group;
};

G_DEFINE_TYPE(GstVideoCompare, gst_video_compare, GST_TYPE_VIDEO_AGGREGATOR)

enum { PROP_0, PROP_HASH_ALGO, PROP_MAX_DIST_THRESHOLD };

#define COMPARE_CAPS GST_VIDEO_CAPS_MAKE("{ RGB, RGBA }")
static GstStaticPadTemplate sink_template = GST_STATIC_PAD_TEMPLATE("sink_%u", GST_PAD_SINK, GST_PAD_REQUEST, GST_STATIC_CAPS(COMPARE_CAPS));
static GstStaticPadTemplate src_template = GST_STATIC_PAD_TEMPLATE("src", GST_PAD_SRC, GST_PAD_ALWAYS, GST_STATIC_CAPS(COMPARE_CAPS));

/* GstVideoCompareHashAlgorithm (videocompare/mod.rs:60-95) */
static GType gst_video_compare_hash_algorithm_get_type(void) {
  static gsize type = 0;
  if (g_once_init_enter(&type)) {
    static const GEnumValue values[] = {
        {MI355_HASH_MEAN, "Mean: The Mean hashing algorithm.", "mean"},
        {MI355_HASH_GRADIENT, "Gradient: The Gradient hashing algorithm.", "gradient"},
        {MI355_HASH_VERTGRADIENT, "VertGradient: The Vertical-Gradient hashing algorithm.", "vertgradient"},
        {MI355_HASH_DOUBLEGRADIENT, "DoubleGradient: The Double-Gradient hashing algorithm.", "doublegradient"},
        {MI355_HASH_BLOCKHASH, "Blockhash: The Blockhash (block median value perceptual hash) algorithm.", "blockhash"},
        {MI355_HASH_DSSIM, "Dssim: Image similarity comparison simulating human perception.", "dssim"},
        {0, NULL, NULL}};
    g_once_init_leave(&type, g_enum_register_static("GstVideoCompareHashAlgorithm", values));
  }
  return (GType)type;
}

static void gst_video_compare_set_property(GObject *object, guint id, const GValue *value, GParamSpec *pspec) {
  GstVideoCompare *self = GST_VIDEO_COMPARE(object);
  g_mutex_lock(&self->lock);
  switch (id) {
    case PROP_HASH_ALGO: self->hash_algo = g_value_get_enum(value); break;
    case PROP_MAX_DIST_THRESHOLD: self->max_dist_threshold = g_value_get_double(value); break;
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID(object, id, pspec); break;
  }
  g_mutex_unlock(&self->lock);
}

static void gst_video_compare_get_property(GObject *object, guint id, GValue *value, GParamSpec *pspec) {
  GstVideoCompare *self = GST_VIDEO_COMPARE(object);
  g_mutex_lock(&self->lock);
  switch (id) {
    case PROP_HASH_ALGO: g_value_set_enum(value, self->hash_algo); break;
    case PROP_MAX_DIST_THRESHOLD: g_value_set_double(value, self->max_dist_threshold); break;
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID(object, id, pspec); break;
  }
  g_mutex_unlock(&self->lock);
}

static gboolean gst_video_compare_start(GstAggregator *agg) {
  GstVideoCompare *self = GST_VIDEO_COMPARE(agg);
  int status = 0;
  self->ctx = mi355_ctx_create(0, &status);
  if (!self->ctx) {
    GST_ELEMENT_ERROR(self, LIBRARY, INIT, ("No MI355X context"), ("%s", mi355_status_string(status)));
    return FALSE;
  }
  const char *members = g_getenv("MI355_GROUP_MEMBERS");
  if (members && atoi(members) >= 2 && (self->group = mi355_group_shared(0, &status)))
    (void)mi355_group_set_rendezvous(self->group, atoi(members), 2000); /* every instance submits + waits at once; a straggler is waited for 2 ms */
  /* The reference hashes frames of ANY size (hashed_image.rs:24-46 -> image_hasher's blockhash_slow for sizes that are not a
   * multiple of 8, e.g. 854x480); the library refuses those unless asked, because that path restates the crate's
   * floating-point code from memory (parity unpinned, include/mi355fx.h MI355_FLAG_BLOCKHASH_ANY_SIZE). An element that
   * errors out on a size the reference accepts is worse than one whose hash of such frames is unpinned: ask. */
  (void)mi355_ctx_set_flag(self->ctx, MI355_FLAG_BLOCKHASH_ANY_SIZE, 1);
  return GST_AGGREGATOR_CLASS(gst_video_compare_parent_class)->start ? GST_AGGREGATOR_CLASS(gst_video_compare_parent_class)->start(agg) : TRUE;
}

static gboolean gst_video_compare_stop(GstAggregator *agg) {
  GstVideoCompare *self = GST_VIDEO_COMPARE(agg);
  const gboolean ret = GST_AGGREGATOR_CLASS(gst_video_compare_parent_class)->stop ? GST_AGGREGATOR_CLASS(gst_video_compare_parent_class)->stop(agg) : TRUE;
  if (self->group) mi355_group_release(self->group);
  self->group = NULL;
  if (self->ctx) mi355_ctx_destroy(self->ctx);
  self->ctx = NULL;
  return ret;
}

/* AggregatorImpl::create_new_pad (imp.rs:210-233): the first sink pad is the reference */
static GstAggregatorPad *gst_video_compare_create_new_pad(GstAggregator *agg, GstPadTemplate *templ, const gchar *req_name, const GstCaps *caps) {
  GstVideoCompare *self = GST_VIDEO_COMPARE(agg);
  GstAggregatorPad *pad = GST_AGGREGATOR_CLASS(gst_video_compare_parent_class)->create_new_pad(agg, templ, req_name, caps);
  if (pad) {
    g_mutex_lock(&self->lock);
    if (!self->reference_pad && GST_PAD_DIRECTION(pad) == GST_PAD_SINK) {
      GST_INFO_OBJECT(self, "Reference sink pad selected: %s", GST_PAD_NAME(pad));
      self->reference_pad = GST_PAD(pad);
    }
    g_mutex_unlock(&self->lock);
  }
  return pad;
}

/* ElementImpl::release_pad (imp.rs:188-207) */
static void gst_video_compare_release_pad(GstElement *element, GstPad *pad) {
  GstVideoCompare *self = GST_VIDEO_COMPARE(element);
  g_mutex_lock(&self->lock);
  if (self->reference_pad == pad) { /* "we choose the first that comes" - the reference keeps the LAST other pad it visits */
    self->reference_pad = NULL;
    GST_OBJECT_LOCK(element);
    for (GList *l = element->sinkpads; l; l = l->next)
      if (l->data != (gpointer)pad) self->reference_pad = GST_PAD(l->data);
    GST_OBJECT_UNLOCK(element);
  }
  g_mutex_unlock(&self->lock);
  GST_ELEMENT_CLASS(gst_video_compare_parent_class)->release_pad(element, pad);
}

/* AggregatorImpl::update_src_caps (imp.rs:235-256) */
static GstFlowReturn gst_video_compare_update_src_caps(GstAggregator *agg, GstCaps *caps, GstCaps **ret) {
  GstVideoCompare *self = GST_VIDEO_COMPARE(agg);
  g_mutex_lock(&self->lock);
  GstCaps *sink_caps = self->reference_pad ? gst_pad_get_current_caps(self->reference_pad) : NULL;
  g_mutex_unlock(&self->lock);
  if (!sink_caps) sink_caps = gst_caps_ref(caps); /* "Allow any caps for now" */
  if (!gst_caps_can_intersect(sink_caps, caps)) {
    GST_ERROR_OBJECT(self, "Proposed src caps not supported, needs to intersect with the reference sink caps");
    gst_caps_unref(sink_caps);
    return GST_FLOW_NOT_NEGOTIATED;
  }
  *ret = sink_caps;
  return GST_FLOW_OK;
}

static int gst_video_compare_format(GstVideoFormat f) { return f == GST_VIDEO_FORMAT_RGB ? MI355_FMT_RGB : (f == GST_VIDEO_FORMAT_RGBA ? MI355_FMT_RGBA : -1); }

/* HasherEngine::hash_image (hashed_image.rs:24-46) of one prepared frame. A frame whose buffer is device memory of ours (an
 * upstream mi355 element wrote it) is hashed where it lies - the aggregator has mapped it for reading, which a mi355_buf allows
 * next to a device READ - instead of being uploaded again from the mapped bytes. */
static int gst_video_compare_hash(GstVideoCompare *self, GstVideoFrame *f, int w, int h, int fmt, int algo, uint64_t *hash) {
  mi355_buf *buf = gst_mi355_buffer_peek_device(f->buffer);
  if (buf) {
    const uint8_t *d = mi355_buf_device_ptr(buf, self->ctx, MI355_MAP_READ);
    if (d) {
      int rc = mi355_videocompare_hash_frames_device(self->ctx, d, GST_VIDEO_INFO_SIZE(&f->info), GST_VIDEO_FRAME_PLANE_STRIDE(f, 0), 1, w, h, fmt, algo, hash);
      if (rc == MI355_OK) rc = mi355_buf_commit(buf, self->ctx);
      return rc;
    }
  }
  return mi355_videocompare_hash_frame(self->ctx, GST_VIDEO_FRAME_PLANE_DATA(f, 0), GST_VIDEO_FRAME_PLANE_STRIDE(f, 0), w, h, fmt, algo, hash);
}

/* VideoAggregatorImpl::aggregate_frames (imp.rs:259-388) */
static GstFlowReturn gst_video_compare_aggregate_frames(GstVideoAggregator *vagg, GstBuffer *outbuf) {
  GstVideoCompare *self = GST_VIDEO_COMPARE(vagg);
  GstElement *element = GST_ELEMENT(vagg);
  g_mutex_lock(&self->lock);
  GstPad *reference_pad = self->reference_pad ? (GstPad *)gst_object_ref(self->reference_pad) : NULL;
  const gint algo = self->hash_algo;
  const gdouble threshold = self->max_dist_threshold;
  g_mutex_unlock(&self->lock);
  if (!reference_pad) {
    GST_WARNING_OBJECT(self, "No reference sink pad exists");
    return GST_FLOW_EOS;
  }
  GstFlowReturn ret = GST_FLOW_OK;
  GstVideoFrame *ref = gst_video_aggregator_pad_get_prepared_frame(GST_VIDEO_AGGREGATOR_PAD(reference_pad));
  if (!ref) {
    if (gst_aggregator_pad_is_eos(GST_AGGREGATOR_PAD(reference_pad))) ret = GST_FLOW_EOS;
    else GST_WARNING_OBJECT(self, "The reference sink pad '%s' has not produced a buffer, image comparison not possible", GST_PAD_NAME(reference_pad));
    gst_object_unref(reference_pad);
    return ret;
  }
  /* running time of the reference buffer, when it can be had (imp.rs:303-309) */
  GstClockTime running_time = GST_CLOCK_TIME_NONE;
  const GstSegment *seg = &GST_AGGREGATOR_PAD(reference_pad)->segment;
  if (seg->format == GST_FORMAT_TIME && GST_BUFFER_PTS_IS_VALID(ref->buffer)) running_time = gst_segment_to_running_time(seg, GST_FORMAT_TIME, GST_BUFFER_PTS(ref->buffer));
  /* output the reference buffer (imp.rs:311-315) */
  gst_buffer_remove_all_memory(outbuf);
  if (!gst_buffer_copy_into(outbuf, ref->buffer, GST_BUFFER_COPY_ALL, 0, -1)) { gst_object_unref(reference_pad); return GST_FLOW_ERROR; }

  /* the other pads' frames, in sink-pad order; any pad without one ends this aggregate quietly (imp.rs:331-334) */
  GstPad *pads[VIDEO_COMPARE_MAX_PADS];
  GstVideoFrame *frames[VIDEO_COMPARE_MAX_PADS];
  guint n = 0;
  gboolean missing = FALSE;
  GST_OBJECT_LOCK(element);
  for (GList *l = element->sinkpads; l && !missing; l = l->next) {
    GstPad *pad = GST_PAD(l->data);
    if (pad == reference_pad) continue;
    GstVideoFrame *f = gst_video_aggregator_pad_get_prepared_frame(GST_VIDEO_AGGREGATOR_PAD(pad));
    if (!f) { missing = TRUE; break; }
    if (n == VIDEO_COMPARE_MAX_PADS) break;
    pads[n] = pad;
    frames[n++] = f;
  }
  GST_OBJECT_UNLOCK(element);
  if (missing) { gst_object_unref(reference_pad); return GST_FLOW_OK; }

  const int rfmt = gst_video_compare_format(GST_VIDEO_FRAME_FORMAT(ref));
  const int w = GST_VIDEO_FRAME_WIDTH(ref), h = GST_VIDEO_FRAME_HEIGHT(ref);
  gdouble distances[VIDEO_COMPARE_MAX_PADS];
  for (guint i = 0; i < n; i++) {
    if (GST_VIDEO_FRAME_WIDTH(frames[i]) != w || GST_VIDEO_FRAME_HEIGHT(frames[i]) != h) { /* imp.rs:337-348 */
      GST_ERROR_OBJECT(self, "Video streams do not have the same sizes (add videoscale and force the sizes to be equal on all sink pads)");
      gst_object_unref(reference_pad);
      return GST_FLOW_NOT_NEGOTIATED;
    }
  }
  int rc = MI355_OK;
  /* the dispatcher takes device-resident pairs (frames an upstream mi355 element wrote) of its two batched algorithms */
  gboolean grouped = self->group && (algo == MI355_HASH_DSSIM || (algo == MI355_HASH_BLOCKHASH && w % 8 == 0 && h % 8 == 0)) && gst_mi355_buffer_peek_device(ref->buffer);
  for (guint i = 0; i < n && grouped; i++)
    grouped = gst_mi355_buffer_peek_device(frames[i]->buffer) && gst_video_compare_format(GST_VIDEO_FRAME_FORMAT(frames[i])) == rfmt &&
              GST_VIDEO_FRAME_PLANE_STRIDE(frames[i], 0) == GST_VIDEO_FRAME_PLANE_STRIDE(ref, 0);
  if (grouped) {
    uint64_t tickets[VIDEO_COMPARE_MAX_PADS];
    mi355_buf *rbuf = gst_mi355_buffer_peek_device(ref->buffer);
    const uint8_t *d_ref = mi355_buf_device_ptr(rbuf, self->ctx, MI355_MAP_READ);
    guint submitted = 0;
    for (guint i = 0; i < n && rc == MI355_OK && d_ref; i++) {
      const uint8_t *d = mi355_buf_device_ptr(gst_mi355_buffer_peek_device(frames[i]->buffer), self->ctx, MI355_MAP_READ);
      rc = d ? mi355_group_submit_compare(self->group, self->ctx, d_ref, d, GST_VIDEO_FRAME_PLANE_STRIDE(ref, 0), w, h, rfmt, algo, &tickets[i]) : MI355_ERR_HIP;
      if (rc == MI355_OK) submitted++;
    }
    if (!d_ref) rc = MI355_ERR_HIP;
    for (guint i = 0; i < submitted; i++) { /* every submitted pair is waited for: its frames are read until then */
      const int wrc = mi355_group_wait_compare(self->group, tickets[i], &distances[i], NULL);
      if (rc == MI355_OK) rc = wrc;
    }
    if (rc != MI355_OK) GST_ERROR_OBJECT(self, "grouped comparison failed: %s", mi355_group_last_error(self->group));
  } else if (algo == MI355_HASH_DSSIM) {
    mi355_dssim_image *ref_img = NULL;
    rc = mi355_dssim_create_image(self->ctx, GST_VIDEO_FRAME_PLANE_DATA(ref, 0), GST_VIDEO_FRAME_PLANE_STRIDE(ref, 0), w, h, rfmt, &ref_img);
    /* frames that share format and stride go to the device together; anything else one by one */
    for (guint i = 0; i < n && rc == MI355_OK;) {
      const uint8_t *ptrs[VIDEO_COMPARE_MAX_PADS];
      guint m = 0;
      const int fmt = gst_video_compare_format(GST_VIDEO_FRAME_FORMAT(frames[i])), stride = GST_VIDEO_FRAME_PLANE_STRIDE(frames[i], 0);
      while (i + m < n && gst_video_compare_format(GST_VIDEO_FRAME_FORMAT(frames[i + m])) == fmt && GST_VIDEO_FRAME_PLANE_STRIDE(frames[i + m], 0) == stride) {
        ptrs[m] = (const uint8_t *)GST_VIDEO_FRAME_PLANE_DATA(frames[i + m], 0);
        m++;
      }
      rc = mi355_dssim_compare_frames(self->ctx, ref_img, ptrs, (int)m, stride, w, h, fmt, &distances[i]);
      i += m;
    }
    if (ref_img) mi355_dssim_free_image(self->ctx, ref_img);
  } else {
    uint64_t ref_hash = 0;
    rc = gst_video_compare_hash(self, ref, w, h, rfmt, algo, &ref_hash);
    for (guint i = 0; i < n && rc == MI355_OK; i++) {
      uint64_t hash = 0;
      rc = gst_video_compare_hash(self, frames[i], w, h, gst_video_compare_format(GST_VIDEO_FRAME_FORMAT(frames[i])), algo, &hash);
      distances[i] = mi355_videocompare_distance(algo, ref_hash, hash);
    }
  }
  if (rc != MI355_OK) {
    GST_ERROR_OBJECT(self, "image comparison failed: %s", mi355_ctx_last_error(self->ctx));
    gst_object_unref(reference_pad);
    return GST_FLOW_ERROR;
  }
  gboolean detected = FALSE;
  for (guint i = 0; i < n; i++) detected |= distances[i] <= threshold;
  if (detected) { /* From<VideoCompareMessage> for gst::Structure (mod.rs:120-133) */
    GValue arr = G_VALUE_INIT;
    g_value_init(&arr, GST_TYPE_ARRAY);
    for (guint i = 0; i < n; i++) {
      GValue v = G_VALUE_INIT;
      g_value_init(&v, GST_TYPE_STRUCTURE);
      g_value_take_boxed(&v, gst_structure_new("pad-distance", "pad", GST_TYPE_PAD, pads[i], "distance", G_TYPE_DOUBLE, distances[i], NULL));
      gst_value_array_append_and_take_value(&arr, &v);
    }
    GstStructure *s = gst_structure_new("videocompare", "running-time", GST_TYPE_CLOCK_TIME, running_time, NULL);
    gst_structure_take_value(s, "pad-distances", &arr);
    (void)gst_element_post_message(element, gst_message_new_element(GST_OBJECT(self), s));
  } else {
    GST_DEBUG_OBJECT(self, "Compared images and could not find any frame with distance lower than the threshold of %f", threshold);
  }
  gst_object_unref(reference_pad);
  return GST_FLOW_OK;
}

static void gst_video_compare_finalize(GObject *object) {
  GstVideoCompare *self = GST_VIDEO_COMPARE(object);
  g_mutex_clear(&self->lock);
  G_OBJECT_CLASS(gst_video_compare_parent_class)->finalize(object);
}

static void gst_video_compare_class_init(GstVideoCompareClass *klass) {
  GObjectClass *gobject = G_OBJECT_CLASS(klass);
  GstElementClass *element = GST_ELEMENT_CLASS(klass);
  GstAggregatorClass *agg = GST_AGGREGATOR_CLASS(klass);
  GstVideoAggregatorClass *vagg = GST_VIDEO_AGGREGATOR_CLASS(klass);
  gobject->set_property = gst_video_compare_set_property;
  gobject->get_property = gst_video_compare_get_property;
  gobject->finalize = gst_video_compare_finalize;
  const GParamFlags f = (GParamFlags)(G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS | GST_PARAM_MUTABLE_READY);
  g_object_class_install_property(gobject, PROP_HASH_ALGO,
      g_param_spec_enum("hash-algo", "Hashing Algorithm", "Which hashing algorithm to use for image comparisons",
                        gst_video_compare_hash_algorithm_get_type(), MI355_HASH_BLOCKHASH, f));
  g_object_class_install_property(gobject, PROP_MAX_DIST_THRESHOLD,
      g_param_spec_double("max-dist-threshold", "Maximum Distance Threshold",
                          "Maximum distance threshold to emit messages when an image is detected, by default emits only on exact match", 0.0, G_MAXDOUBLE, 0.0, f));
  gst_element_class_set_static_metadata(element, "Image comparison", "Filter/Video", "Compare similarity of video frames", "Rafael Caricio <rafael@caricio.com>");
  gst_element_class_add_static_pad_template_with_gtype(element, &sink_template, GST_TYPE_VIDEO_AGGREGATOR_PAD);
  gst_element_class_add_static_pad_template_with_gtype(element, &src_template, GST_TYPE_AGGREGATOR_PAD);
  element->release_pad = gst_video_compare_release_pad;
  agg->start = gst_video_compare_start;
  agg->stop = gst_video_compare_stop;
  agg->create_new_pad = gst_video_compare_create_new_pad;
  agg->update_src_caps = gst_video_compare_update_src_caps;
  vagg->aggregate_frames = gst_video_compare_aggregate_frames;
  GST_DEBUG_CATEGORY_INIT(gst_video_compare_debug, "videocompare", 0, "Video frames comparison (MI355X)");
}

static void gst_video_compare_init(GstVideoCompare *self) {
  g_mutex_init(&self->lock);
  self->hash_algo = MI355_HASH_BLOCKHASH; /* imp.rs:38-46 */
  self->max_dist_threshold = 0.0;
}

gboolean gst_video_compare_register(GstPlugin *plugin) {
  return gst_element_register(plugin, "videocompare", GST_RANK_NONE, GST_TYPE_VIDEO_COMPARE); /* videocompare/mod.rs:97-104 */
}
