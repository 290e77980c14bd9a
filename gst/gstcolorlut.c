/* gst/gstcolorlut.c — `colorlut` (GType GstColorLut), a GstVideoFilter that never works in place, over the mi355fx C ABI.
 * Surface mirrored from the reference (video/colorlut/src/colorlut/imp.rs): GType name :61-66, `location` string
 * property mutable in READY :73-104, metadata :109-114, caps {RGBA64_LE, RGBA64_BE, RGBA} :119-158, NeverInPlace :162-166,
 * start() parses the .cube file and fails with ResourceError::Settings / ::Read :168-194, stop() drops the LUT :196-199,
 * transform_frame :203-223 -> mi355_colorlut_frame (the per-pixel loops :226-543 run on the GPU).
 * The .cube text is parsed by the product's host parser (libmi355fx_host.so, host/cube_lut.cpp = CubeLut::parse_file,
 * video/colorlut/src/parser.rs:105-281).
 *
 * Data path (round 3). The element is ONE FRAME DEEP: submit_input_buffer / generate_output replace the synchronous
 * transform (the way audio/audiofx/src/audiornnoise/imp.rs:323-348 queues input and produces output when it has enough;
 * its latency query :362-385 is mirrored in gst_color_lut_query). generate_output takes the queued input, allocates the
 * output from the negotiated (pinned) pool, hands both to the library's asynchronous pipeline (mi355_pipe_submit_colorlut:
 * upload of frame n, kernel of frame n and download of frame n-1 overlap on side streams) and returns the PREVIOUS frame,
 * whose ticket it waits for. One frame of latency is reported; EOS, flushes, caps changes and discontinuities drain the
 * queue. When the input buffer carries a GstMi355HsvMeta (a directly upstream hsvfilter of this shim deferred its work,
 * gstmi355common.h), the fused hsvfilter -> colorlut kernel runs instead (mi355_pipe_submit_hsv_colorlut): the pair
 * costs one upload, one launch and one download - DESIGN.md §6: 1.4 k frames/s at 4K, PCIe-bound, against 410 for two
 * synchronous elements. transform_frame stays as the synchronous fallback (no pipeline: mi355_pipe_create failed).
 * propose_allocation / decide_allocation put input and output buffers in pinned memory (d3d12colorlut/imp.rs:385-492 is
 * the reference's precedent for an element that brings its own memory). */
#include "gstmi355common.h"
#include "../gst-plugins-rs_amd/host/mi355fx_host.h"

GST_DEBUG_CATEGORY_STATIC(gst_color_lut_debug);
#define GST_CAT_DEFAULT gst_color_lut_debug

#define GST_TYPE_COLOR_LUT (gst_color_lut_get_type())
G_DECLARE_FINAL_TYPE(GstColorLut, gst_color_lut, GST, COLOR_LUT, GstVideoFilter)

#define COLOR_LUT_DEPTH 1 /* frames in flight behind the one being returned */

typedef struct {
  GstBuffer *inbuf, *outbuf;
  GstVideoFrame in, out; /* mapped for the duration of the job: the library borrows the pointers until the ticket is waited for */
  uint64_t ticket;
  gboolean device; /* both buffers are device memory of ours: nothing mapped, nothing to wait for on the host */
} ColorLutJob;

struct _GstColorLut {
  GstVideoFilter parent;
  GMutex lock;
  gchar *location;
  mi355_ctx *ctx;
  mi355_pipe *pipe;  /* asynchronous upload / compute / download pipeline of the library (NULL: synchronous fallback) */
  GQueue jobs;       /* ColorLutJob*, oldest first; streaming thread only */
  gboolean have_lut; /* State { lut: Option<CubeLut> } (imp.rs:55-58) */
};

G_DEFINE_TYPE(GstColorLut, gst_color_lut, GST_TYPE_VIDEO_FILTER)

enum { PROP_0, PROP_LOCATION };

#if G_BYTE_ORDER == G_BIG_ENDIAN
#define LUT_FORMATS "{ RGBA64_BE, RGBA64_LE, RGBA }"
#else
#define LUT_FORMATS "{ RGBA64_LE, RGBA64_BE, RGBA }"
#endif
static GstStaticPadTemplate sink_template =
    GST_STATIC_PAD_TEMPLATE("sink", GST_PAD_SINK, GST_PAD_ALWAYS, GST_STATIC_CAPS(GST_VIDEO_CAPS_MAKE(LUT_FORMATS)));
static GstStaticPadTemplate src_template =
    GST_STATIC_PAD_TEMPLATE("src", GST_PAD_SRC, GST_PAD_ALWAYS, GST_STATIC_CAPS(GST_VIDEO_CAPS_MAKE(LUT_FORMATS)));

static void gst_color_lut_set_property(GObject *object, guint id, const GValue *value, GParamSpec *pspec) {
  GstColorLut *self = GST_COLOR_LUT(object);
  if (id != PROP_LOCATION) { G_OBJECT_WARN_INVALID_PROPERTY_ID(object, id, pspec); return; }
  g_mutex_lock(&self->lock);
  g_free(self->location);
  self->location = g_value_dup_string(value);
  g_mutex_unlock(&self->lock);
}

static void gst_color_lut_get_property(GObject *object, guint id, GValue *value, GParamSpec *pspec) {
  GstColorLut *self = GST_COLOR_LUT(object);
  if (id != PROP_LOCATION) { G_OBJECT_WARN_INVALID_PROPERTY_ID(object, id, pspec); return; }
  g_mutex_lock(&self->lock);
  g_value_set_string(value, self->location);
  g_mutex_unlock(&self->lock);
}

/* BaseTransformImpl::start (imp.rs:168-194) */
static gboolean gst_color_lut_start(GstBaseTransform *trans) {
  GstColorLut *self = GST_COLOR_LUT(trans);
  g_mutex_lock(&self->lock);
  gchar *location = g_strdup(self->location);
  g_mutex_unlock(&self->lock);
  if (!location) {
    GST_ELEMENT_ERROR(self, RESOURCE, SETTINGS, ("LUT file location is not configured"), (NULL));
    return FALSE;
  }
  char err[512] = {0};
  mi355h_cube *cube = mi355h_cube_parse_file(location, err, sizeof err);
  if (!cube) {
    GST_ELEMENT_ERROR(self, RESOURCE, READ, ("Failed to parse LUT file %s: %s", location, err), (NULL));
    g_free(location);
    return FALSE;
  }
  g_free(location);
  int status = 0;
  self->ctx = mi355_ctx_create(0, &status);
  if (!self->ctx) {
    mi355h_cube_free(cube);
    GST_ELEMENT_ERROR(self, LIBRARY, INIT, ("No MI355X context"), ("%s", mi355_status_string(status)));
    return FALSE;
  }
  float scale[3], offset[3];
  mi355h_cube_domain(cube, scale, offset);
  const int rc = mi355_colorlut_load(self->ctx, mi355h_cube_is3d(cube), mi355h_cube_size(cube), mi355h_cube_table(cube), scale, offset);
  mi355h_cube_free(cube);
  if (rc != MI355_OK) {
    GST_ELEMENT_ERROR(self, LIBRARY, INIT, ("LUT upload failed"), ("%s", mi355_ctx_last_error(self->ctx)));
    mi355_ctx_destroy(self->ctx);
    self->ctx = NULL;
    return FALSE;
  }
  self->have_lut = TRUE;
  return TRUE;
}

/* ---- the one-frame-deep queue */

static void color_lut_job_free(ColorLutJob *job, gboolean keep_out) {
  if (!job->device) {
    gst_video_frame_unmap(&job->in);
    gst_video_frame_unmap(&job->out);
  }
  gst_buffer_unref(job->inbuf);
  if (!keep_out) gst_buffer_unref(job->outbuf);
  g_free(job);
}

/* waits for the oldest job; *outbuf takes its output buffer */
static GstFlowReturn color_lut_finish_oldest(GstColorLut *self, GstBuffer **outbuf) {
  ColorLutJob *job = g_queue_pop_head(&self->jobs);
  *outbuf = NULL;
  if (!job) return GST_FLOW_OK;
  /* (a device job is ordered on the device: whoever maps or uses the output memory next is ordered behind the commit) */
  const int rc = job->device ? MI355_OK : mi355_pipe_wait(self->pipe, job->ticket);
  if (rc != MI355_OK) {
    GST_ERROR_OBJECT(self, "mi355_pipe_wait: %s", mi355_ctx_last_error(self->ctx));
    color_lut_job_free(job, FALSE);
    return GST_FLOW_ERROR;
  }
  *outbuf = job->outbuf;
  color_lut_job_free(job, TRUE);
  return GST_FLOW_OK;
}

/* pushes everything still in flight downstream (EOS, a new segment or caps, a discontinuity) or drops it (flush) */
static GstFlowReturn color_lut_drain(GstColorLut *self, gboolean push) {
  GstFlowReturn ret = GST_FLOW_OK;
  while (!g_queue_is_empty(&self->jobs)) {
    GstBuffer *out = NULL;
    const GstFlowReturn r = color_lut_finish_oldest(self, &out);
    if (r != GST_FLOW_OK) { ret = r; continue; }
    if (push && ret == GST_FLOW_OK) ret = gst_pad_push(GST_BASE_TRANSFORM_SRC_PAD(self), out);
    else gst_buffer_unref(out);
  }
  return ret;
}

/* BaseTransformImpl::generate_output. The base class calls it after every submit_input_buffer until it returns no buffer. */
static GstFlowReturn gst_color_lut_generate_output(GstBaseTransform *trans, GstBuffer **outbuf) {
  GstColorLut *self = GST_COLOR_LUT(trans);
  GstVideoFilter *filter = GST_VIDEO_FILTER(trans);
  *outbuf = NULL;
  if (!self->pipe) /* synchronous fallback: the default implementation maps the frames and calls transform_frame */
    return GST_BASE_TRANSFORM_CLASS(gst_color_lut_parent_class)->generate_output(trans, outbuf);
  GstBuffer *inbuf = trans->queued_buf; /* take_queued_buffer */
  trans->queued_buf = NULL;
  if (inbuf) {
    if (!self->have_lut) {
      GST_ERROR_OBJECT(self, "No LUT configured"); /* imp.rs:210-213 */
      gst_buffer_unref(inbuf);
      return GST_FLOW_ERROR;
    }
    if (!filter->negotiated) {
      gst_buffer_unref(inbuf);
      return GST_FLOW_NOT_NEGOTIATED;
    }
    if (GST_BUFFER_IS_DISCONT(inbuf)) { /* frames before a discontinuity do not wait for frames after it */
      const GstFlowReturn r = color_lut_drain(self, TRUE);
      if (r != GST_FLOW_OK) { gst_buffer_unref(inbuf); return r; }
    }
    ColorLutJob *job = g_new0(ColorLutJob, 1);
    job->inbuf = inbuf;
    GstFlowReturn ret = GST_BASE_TRANSFORM_GET_CLASS(trans)->prepare_output_buffer(trans, inbuf, &job->outbuf); /* pool buffer + metadata copy */
    if (ret != GST_FLOW_OK || !job->outbuf) {
      gst_buffer_unref(inbuf);
      g_free(job);
      return ret != GST_FLOW_OK ? ret : GST_FLOW_ERROR;
    }
    {
      /* Frames in device memory of ours (an upstream mi355 element wrote them, or upstream took the pool offered in
       * propose_allocation) and an output buffer from a device pool: the `_device` entry points on the buffers' device
       * pointers, no map, no PCIe (the d3d12colorlut way: imp.rs:494-560 works on the GPU resource when the memory is its own). */
      mi355_buf *bin = gst_mi355_buffer_peek_device(inbuf), *bout = gst_mi355_buffer_peek_device(job->outbuf);
      const int dfmt = gst_mi355_format(GST_VIDEO_INFO_FORMAT(&filter->in_info));
      if (bin && bout && dfmt >= 0) {
        const GstVideoInfo *ii = &filter->in_info, *oi = &filter->out_info;
        const GstMi355HsvMeta *dhsv = dfmt == MI355_FMT_RGBA ? gst_buffer_get_mi355_hsv_meta(inbuf) : NULL;
        const uint8_t *d_src = mi355_buf_device_ptr(bin, self->ctx, MI355_MAP_READ);
        uint8_t *d_dst = mi355_buf_device_ptr(bout, self->ctx, MI355_MAP_WRITE);
        int drc = d_src && d_dst ? MI355_OK : MI355_ERR_HIP;
        if (drc == MI355_OK && dhsv)
          drc = mi355_hsv_colorlut_frames_device(self->ctx, d_src, GST_VIDEO_INFO_SIZE(ii), GST_VIDEO_INFO_PLANE_STRIDE(ii, 0), d_dst, GST_VIDEO_INFO_SIZE(oi),
                                                 GST_VIDEO_INFO_PLANE_STRIDE(oi, 0), 1, GST_VIDEO_INFO_WIDTH(ii), GST_VIDEO_INFO_HEIGHT(ii), &dhsv->settings);
        else if (drc == MI355_OK)
          drc = mi355_colorlut_frames_device(self->ctx, d_src, GST_VIDEO_INFO_SIZE(ii), GST_VIDEO_INFO_PLANE_STRIDE(ii, 0), d_dst, GST_VIDEO_INFO_SIZE(oi),
                                             GST_VIDEO_INFO_PLANE_STRIDE(oi, 0), 1, GST_VIDEO_INFO_WIDTH(ii), GST_VIDEO_INFO_HEIGHT(ii), dfmt);
        if (drc == MI355_OK) drc = mi355_buf_commit(bin, self->ctx);
        if (drc == MI355_OK) drc = mi355_buf_commit(bout, self->ctx);
        if (drc != MI355_OK) {
          GST_ERROR_OBJECT(self, "colorlut on device memory: %s", mi355_ctx_last_error(self->ctx));
          gst_buffer_unref(inbuf); gst_buffer_unref(job->outbuf); g_free(job);
          return GST_FLOW_ERROR;
        }
        if (dhsv) {
          GstMeta *m = gst_buffer_get_meta(job->outbuf, GST_MI355_HSV_META_API_TYPE);
          if (m) gst_buffer_remove_meta(job->outbuf, m);
        }
        job->device = TRUE;
        g_queue_push_tail(&self->jobs, job);
        return color_lut_finish_oldest(self, outbuf); /* no host work to overlap: the frame goes downstream at once, in order */
      }
    }
    if (!gst_video_frame_map(&job->in, &filter->in_info, inbuf, GST_MAP_READ)) {
      gst_buffer_unref(inbuf); gst_buffer_unref(job->outbuf); g_free(job);
      return GST_FLOW_ERROR;
    }
    if (!gst_video_frame_map(&job->out, &filter->out_info, job->outbuf, GST_MAP_WRITE)) {
      gst_video_frame_unmap(&job->in);
      gst_buffer_unref(inbuf); gst_buffer_unref(job->outbuf); g_free(job);
      return GST_FLOW_ERROR;
    }
    const int fmt = gst_mi355_format(GST_VIDEO_FRAME_FORMAT(&job->in));
    const GstMi355HsvMeta *hsv = fmt == MI355_FMT_RGBA ? gst_buffer_get_mi355_hsv_meta(inbuf) : NULL;
    int rc;
    if (hsv) /* an upstream hsvfilter left its work to the fused kernel */
      rc = mi355_pipe_submit_hsv_colorlut(self->pipe, GST_VIDEO_FRAME_PLANE_DATA(&job->in, 0), GST_VIDEO_FRAME_PLANE_STRIDE(&job->in, 0),
                                          GST_VIDEO_FRAME_PLANE_DATA(&job->out, 0), GST_VIDEO_FRAME_PLANE_STRIDE(&job->out, 0),
                                          GST_VIDEO_FRAME_WIDTH(&job->in), GST_VIDEO_FRAME_HEIGHT(&job->in), &hsv->settings, &job->ticket);
    else
      rc = mi355_pipe_submit_colorlut(self->pipe, GST_VIDEO_FRAME_PLANE_DATA(&job->in, 0), GST_VIDEO_FRAME_PLANE_STRIDE(&job->in, 0),
                                      GST_VIDEO_FRAME_PLANE_DATA(&job->out, 0), GST_VIDEO_FRAME_PLANE_STRIDE(&job->out, 0),
                                      GST_VIDEO_FRAME_WIDTH(&job->in), GST_VIDEO_FRAME_HEIGHT(&job->in), fmt, &job->ticket);
    if (rc != MI355_OK) {
      GST_ERROR_OBJECT(self, "mi355_pipe_submit: %s", mi355_ctx_last_error(self->ctx));
      color_lut_job_free(job, FALSE);
      return GST_FLOW_ERROR;
    }
    if (hsv) { /* the output is filtered: the deferral note must not travel further */
      GstMeta *m = gst_buffer_get_meta(job->outbuf, GST_MI355_HSV_META_API_TYPE);
      if (m) gst_buffer_remove_meta(job->outbuf, m);
    }
    g_queue_push_tail(&self->jobs, job);
  }
  /* hand out the oldest frame once more than COLOR_LUT_DEPTH are in flight: frame n-1 while frame n uploads and computes */
  if (g_queue_get_length(&self->jobs) > COLOR_LUT_DEPTH) return color_lut_finish_oldest(self, outbuf);
  return GST_FLOW_OK; /* GenerateOutputSuccess::NoOutput */
}

static gboolean gst_color_lut_sink_event(GstBaseTransform *trans, GstEvent *event) {
  GstColorLut *self = GST_COLOR_LUT(trans);
  switch (GST_EVENT_TYPE(event)) {
    case GST_EVENT_EOS: case GST_EVENT_SEGMENT: case GST_EVENT_CAPS: case GST_EVENT_GAP:
      /* audiornnoise drains on EOS before forwarding it (imp.rs:349-360); frames of the old segment / caps leave first too */
      if (color_lut_drain(self, TRUE) != GST_FLOW_OK && GST_EVENT_TYPE(event) == GST_EVENT_EOS) GST_WARNING_OBJECT(self, "drain at EOS failed");
      break;
    case GST_EVENT_FLUSH_STOP:
      (void)color_lut_drain(self, FALSE);
      break;
    default: break;
  }
  return GST_BASE_TRANSFORM_CLASS(gst_color_lut_parent_class)->sink_event(trans, event);
}

/* BaseTransformImpl::query: one frame of latency on top of upstream's (audiornnoise/imp.rs:362-385), and the answer to the
 * fusion question of a directly upstream hsvfilter of this shim */
static gboolean gst_color_lut_query(GstBaseTransform *trans, GstPadDirection direction, GstQuery *query) {
  GstColorLut *self = GST_COLOR_LUT(trans);
  GstVideoFilter *filter = GST_VIDEO_FILTER(trans);
  if (direction == GST_PAD_SINK && GST_QUERY_TYPE(query) == GST_QUERY_CUSTOM) {
    const GstStructure *s = gst_query_get_structure(query);
    if (s && gst_structure_has_name(s, GST_MI355_FUSE_QUERY_NAME)) {
      if (!(self->pipe != NULL && filter->negotiated && GST_VIDEO_INFO_FORMAT(&filter->in_info) == GST_VIDEO_FORMAT_RGBA)) return FALSE;
      /* say WHO answers: the asking hsvfilter only believes the element that owns its peer pad (the query may have been
       * forwarded to us through elements in between) */
      gst_structure_set(gst_query_writable_structure(query), GST_MI355_FUSE_QUERY_WHO, G_TYPE_POINTER, (gpointer)self, NULL);
      return TRUE;
    }
  }
  if (direction == GST_PAD_SRC && GST_QUERY_TYPE(query) == GST_QUERY_LATENCY && self->pipe) {
    GstQuery *upstream = gst_query_new_latency();
    if (gst_pad_peer_query(GST_BASE_TRANSFORM_SINK_PAD(trans), upstream)) {
      gboolean live;
      GstClockTime min, max;
      gst_query_parse_latency(upstream, &live, &min, &max);
      gst_query_unref(upstream);
      GstClockTime frame = 0;
      if (filter->negotiated && GST_VIDEO_INFO_FPS_N(&filter->in_info) > 0)
        frame = gst_util_uint64_scale(GST_SECOND * COLOR_LUT_DEPTH, GST_VIDEO_INFO_FPS_D(&filter->in_info), GST_VIDEO_INFO_FPS_N(&filter->in_info));
      min += frame;
      if (GST_CLOCK_TIME_IS_VALID(max)) max += frame;
      gst_query_set_latency(query, live, min, max);
      return TRUE;
    }
    gst_query_unref(upstream);
  }
  return GST_BASE_TRANSFORM_CLASS(gst_color_lut_parent_class)->query(trans, direction, query);
}

/* caps are known: size the pipeline's device slots for this frame size */
static gboolean gst_color_lut_set_info(GstVideoFilter *filter, GstCaps *incaps, GstVideoInfo *in_info, GstCaps *outcaps, GstVideoInfo *out_info) {
  GstColorLut *self = GST_COLOR_LUT(filter);
  /* A caps event has drained already (sink_event); a downstream RECONFIGURE gets here from the streaming thread with a
   * frame still queued, and its ticket belongs to the pipe that is about to go: finish and push it first. */
  if (color_lut_drain(self, TRUE) != GST_FLOW_OK) GST_WARNING_OBJECT(self, "frames in flight could not be pushed before the renegotiation");
  if (self->pipe) { (void)mi355_pipe_wait_all(self->pipe); mi355_pipe_destroy(self->pipe); self->pipe = NULL; }
  if (self->ctx) {
    const size_t bytes = MAX(GST_VIDEO_INFO_SIZE(in_info), GST_VIDEO_INFO_SIZE(out_info));
    self->pipe = mi355_pipe_create(self->ctx, COLOR_LUT_DEPTH + 1, bytes);
    if (!self->pipe) GST_WARNING_OBJECT(self, "no asynchronous pipeline (%s): synchronous transform", mi355_ctx_last_error(self->ctx));
    else gst_element_post_message(GST_ELEMENT(self), gst_message_new_latency(GST_OBJECT(self))); /* the reported latency changed */
  }
  return TRUE;
}

/* BaseTransformImpl::stop (imp.rs:196-199) */
static gboolean gst_color_lut_stop(GstBaseTransform *trans) {
  GstColorLut *self = GST_COLOR_LUT(trans);
  (void)color_lut_drain(self, FALSE);
  if (self->pipe) mi355_pipe_destroy(self->pipe);
  self->pipe = NULL;
  self->have_lut = FALSE;
  if (self->ctx) mi355_ctx_destroy(self->ctx);
  self->ctx = NULL;
  return TRUE;
}

static gboolean gst_color_lut_propose_allocation(GstBaseTransform *trans, GstQuery *decide_query, GstQuery *query) {
  GstColorLut *self = GST_COLOR_LUT(trans);
  if (!GST_BASE_TRANSFORM_CLASS(gst_color_lut_parent_class)->propose_allocation(trans, decide_query, query)) return FALSE;
  if (self->ctx) {
    (void)gst_mi355_propose_device_pool(trans, query); /* first choice: frames stay in HBM between mi355 elements */
    (void)gst_mi355_propose_pinned_pool(trans, query);
  }
  return TRUE;
}

/* decide_allocation: our own output buffers come from a pinned pool too (NeverInPlace: the base class allocates them) */
static gboolean gst_color_lut_decide_allocation(GstBaseTransform *trans, GstQuery *query) {
  GstColorLut *self = GST_COLOR_LUT(trans);
  if (self->ctx && gst_query_get_n_allocation_pools(query) == 0) {
    GstCaps *caps = NULL;
    GstVideoInfo info;
    gst_query_parse_allocation(query, &caps, NULL);
    if (caps && gst_video_info_from_caps(&info, caps)) {
      GstBufferPool *pool = gst_mi355_pinned_pool_new(caps, &info);
      if (pool) {
        gst_query_add_allocation_pool(query, pool, GST_VIDEO_INFO_SIZE(&info), COLOR_LUT_DEPTH + 2, 0);
        gst_object_unref(pool);
      }
    }
  }
  return GST_BASE_TRANSFORM_CLASS(gst_color_lut_parent_class)->decide_allocation(trans, query);
}

/* VideoFilterImpl::transform_frame (imp.rs:203-223): the synchronous form (fallback when no pipeline could be created) */
static GstFlowReturn gst_color_lut_transform_frame(GstVideoFilter *filter, GstVideoFrame *in, GstVideoFrame *out) {
  GstColorLut *self = GST_COLOR_LUT(filter);
  if (!self->have_lut) {
    GST_ERROR_OBJECT(self, "No LUT configured"); /* imp.rs:210-213 */
    return GST_FLOW_ERROR;
  }
  const int fmt = gst_mi355_format(GST_VIDEO_FRAME_FORMAT(in));
  if (fmt != MI355_FMT_RGBA && fmt != MI355_FMT_RGBA64_LE && fmt != MI355_FMT_RGBA64_BE) return GST_FLOW_NOT_NEGOTIATED;
  if (gst_buffer_get_mi355_hsv_meta(in->buffer)) {
    /* cannot happen: the fusion query is only answered while the pipeline exists, and this path only runs without one */
    GST_ERROR_OBJECT(self, "buffer carries a deferred hsvfilter but the fused path is not available");
    return GST_FLOW_ERROR;
  }
  const int rc = mi355_colorlut_frame(self->ctx, GST_VIDEO_FRAME_PLANE_DATA(in, 0), GST_VIDEO_FRAME_PLANE_STRIDE(in, 0),
                                      GST_VIDEO_FRAME_PLANE_DATA(out, 0), GST_VIDEO_FRAME_PLANE_STRIDE(out, 0), GST_VIDEO_FRAME_WIDTH(in),
                                      GST_VIDEO_FRAME_HEIGHT(in), fmt);
  if (rc != MI355_OK) {
    GST_ERROR_OBJECT(self, "mi355_colorlut_frame: %s", mi355_ctx_last_error(self->ctx));
    return GST_FLOW_ERROR;
  }
  return GST_FLOW_OK;
}

static void gst_color_lut_finalize(GObject *object) {
  GstColorLut *self = GST_COLOR_LUT(object);
  g_free(self->location);
  g_mutex_clear(&self->lock);
  G_OBJECT_CLASS(gst_color_lut_parent_class)->finalize(object);
}

static void gst_color_lut_class_init(GstColorLutClass *klass) {
  GObjectClass *gobject = G_OBJECT_CLASS(klass);
  GstElementClass *element = GST_ELEMENT_CLASS(klass);
  GstBaseTransformClass *trans = GST_BASE_TRANSFORM_CLASS(klass);
  GstVideoFilterClass *vfilter = GST_VIDEO_FILTER_CLASS(klass);
  gobject->set_property = gst_color_lut_set_property;
  gobject->get_property = gst_color_lut_get_property;
  gobject->finalize = gst_color_lut_finalize;
  g_object_class_install_property(gobject, PROP_LOCATION,
      g_param_spec_string("location", "Location", "Path to the LUT file (.cube)", NULL,
                          (GParamFlags)(G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS | GST_PARAM_MUTABLE_READY)));
  gst_element_class_set_static_metadata(element, "Color LUT", "Filter/Effect/Video", "Apply color lookup table",
                                        "Seungha Yang <seungha@centricular.com>");
  gst_element_class_add_static_pad_template(element, &sink_template);
  gst_element_class_add_static_pad_template(element, &src_template);
  trans->start = gst_color_lut_start;
  trans->stop = gst_color_lut_stop;
  trans->propose_allocation = gst_color_lut_propose_allocation;
  trans->decide_allocation = gst_color_lut_decide_allocation;
  trans->generate_output = gst_color_lut_generate_output; /* submit_input_buffer stays the base class's: it queues the buffer */
  trans->sink_event = gst_color_lut_sink_event;
  trans->query = gst_color_lut_query;
  trans->passthrough_on_same_caps = FALSE;     /* imp.rs:164 */
  trans->transform_ip_on_passthrough = FALSE;  /* imp.rs:165 */
  vfilter->set_info = gst_color_lut_set_info;
  vfilter->transform_frame = gst_color_lut_transform_frame; /* only the non-ip slot == BaseTransformMode::NeverInPlace */
  GST_DEBUG_CATEGORY_INIT(gst_color_lut_debug, "colorlut", 0, "Color LUT (MI355X)");
}

static void gst_color_lut_init(GstColorLut *self) {
  g_mutex_init(&self->lock);
  g_queue_init(&self->jobs);
}

gboolean gst_color_lut_register(GstPlugin *plugin) {
  return gst_element_register(plugin, "colorlut", GST_RANK_NONE, GST_TYPE_COLOR_LUT); /* colorlut/mod.rs */
}
