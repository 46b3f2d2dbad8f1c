/* Shared implementation of `bayer2rgb` and `rgb2bayer` -- MI355X-native.
 *
 * Everything a neighbouring element can observe is kept identical to the
 * reference elements (gst-plugins-bad 1.19.2):
 *   transform_caps / get_unit_size / set_caps
 *        gst/bayer/gstbayer2rgb.c:289-322 / :324-352 / :237-276
 *        gst/bayer/gstrgb2bayer.c:128-159 / :161-188 / :190-228
 *   1-in/1-out synchronous transform (the default mode)
 *        gstbayer2rgb.c:456-487, gstrgb2bayer.c:230-278
 * What changes is below the transform vfunc: the reference calls its CPU frame
 * loops (gst_bayer2rgb_process, gstbayer2rgb.c:475-477; the double loop of
 * gstrgb2bayer.c:254-268); here the mapped pointers and strides go to the HIP
 * path through the C ABI of mibayer.h.  There is no CPU fallback: without a
 * usable MI355X the elements post a RESOURCE error instead of converting on
 * the host.
 *
 * Additive, optional behaviour:
 *   - hipHostMalloc-pinned buffer pools are proposed upstream and used
 *     downstream when nobody offers a pool (propose/decide_allocation), so the
 *     H2D/D2H copies are asynchronous DMA;
 *   - `inflight` > 1 and/or `devices` switch to a queued mode: input buffers
 *     are submitted to a round-robin pool of GPUs (frame g -> devices[g % N])
 *     and outputs are pushed in order as they complete; pending frames are
 *     drained before EOS / caps / segment events and dropped on flush
 *     (FLUSH_START already, which arrives on another thread: the pool and the
 *     queues sit behind `flow_lock`);
 *   - a device that fails is dropped from the rotation by the pool, its frames are
 *     converted again on the survivors, and the element posts ONE warning per
 *     dropped device; the stream errors out only when no device is left.  Frames
 *     that were in flight on a device that stopped ANSWERING (timeout-ms) are
 *     dropped instead, and their buffers stay quarantined -- mapped, referenced,
 *     out of every buffer pool -- until the device has caught up: its queued
 *     copies may still run (mibayer.h, mibayer_pool_reclaim).
 * Geometry the HIP path cannot reproduce bit-exactly (odd width, width < 4,
 * height < 3: the reference itself reads stale scratch / out of bounds there,
 * gstbayer2rgb.c:365-380, :430-447) is refused in set_caps -> not-negotiated,
 * the reference's own failure style (:263-265), not at the first buffer.
 *     In-tree precedent for queueing in submit_input_buffer/generate_output:
 *     sys/va/gstvadeinterlace.c:186-232, :467-531.
 */
#ifdef HAVE_CONFIG_H
#include "config.h"
#endif

#include <stdlib.h>
#include <string.h>

#include "gstmibayerelement.h"
#include "gstmihostpool.h"

/* each element logs into its own category, named like the reference's
 * ("bayer2rgb", "rgb2bayer"); the category lives in the class */
#define EL_DEBUG(obj, ...) \
  GST_CAT_DEBUG_OBJECT (ELEMENT_CLASS_OF (obj)->cat, obj, __VA_ARGS__)
#define EL_WARNING(obj, ...) \
  GST_CAT_WARNING_OBJECT (ELEMENT_CLASS_OF (obj)->cat, obj, __VA_ARGS__)
#define EL_INFO(obj, ...) \
  GST_CAT_INFO_OBJECT (ELEMENT_CLASS_OF (obj)->cat, obj, __VA_ARGS__)

#define ELEMENT(obj) ((GstMiBayerElement *) (obj))
#define ELEMENT_CLASS_OF(obj) \
  ((GstMiBayerElementClass *) G_OBJECT_GET_CLASS (obj))
#define LABEL(obj) (ELEMENT_CLASS_OF (obj)->label)
#define IS_INVERSE(obj) (ELEMENT_CLASS_OF (obj)->inverse)
#define BASE_CLASS(obj) (ELEMENT_CLASS_OF (obj)->base_class)

enum
{
  PROP_0,
  PROP_DEVICE_ID,
  PROP_DEVICES,
  PROP_INFLIGHT,
  PROP_HIPGRAPH,
  PROP_PINNED_POOL,
  PROP_TIMEOUT_MS
};

#define DEFAULT_DEVICE_ID 0
#define DEFAULT_INFLIGHT 1
#define DEFAULT_HIPGRAPH FALSE
#define DEFAULT_PINNED_POOL TRUE
#define DEFAULT_TIMEOUT_MS 10000

/* one frame between submit and wait: both buffers stay mapped until the GPU
 * has written the output */
typedef struct
{
  GstBuffer *inbuf;
  GstBuffer *outbuf;
  GstBuffer *mosaic_buf;        /* == inbuf (bayer2rgb) or outbuf (rgb2bayer) */
  GstMapInfo mosaic;            /* the 8-bit mosaic side */
  GstVideoFrame video;          /* the 4-byte-per-pixel side */
  gboolean owns_outbuf;         /* queued mode: we hold the only reference to outbuf;
                                   synchronous mode: the base class owns it */
} PendingFrame;

/* ---- GPU pool --------------------------------------------------------------- */

/* unmap and free the bookkeeping; the output buffer is unreffed only if this
 * entry owns it and the caller does not take it over */
static void
pending_release (PendingFrame * p, gboolean caller_takes_outbuf)
{
  gst_video_frame_unmap (&p->video);
  gst_buffer_unmap (p->mosaic_buf, &p->mosaic);
  gst_buffer_unref (p->inbuf);
  if (p->owns_outbuf && !caller_takes_outbuf)
    gst_buffer_unref (p->outbuf);
  g_free (p);
}

static void
post_gpu_failure (GstMiBayerElement * self, int rc)
{
  GST_ELEMENT_ERROR (self, RESOURCE, FAILED,
      ("%s: GPU conversion failed", LABEL (self)),
      ("%s %s", mibayer_strerror (rc), mibayer_last_hip_error ()));
}

/* an error found while flow_lock is held: remembered (the first one wins), posted by element_post_notes once
 * the lock is released -- posting runs the bus' sync handler, i.e. application code */
static void
element_defer_error (GstMiBayerElement * self, GQuark domain, gint code,
    gchar * text, gchar * debug)
{
  if (self->error_text != NULL) {
    g_free (text);
    g_free (debug);
    return;
  }
  self->error_domain = domain;
  self->error_code = code;
  self->error_text = text;
  self->error_debug = debug;
}

/* the pool dropped a device (flow_lock held): remember ONE note for the bus
 * and follow the pool's new capacity */
static void
element_note_failures (GstMiBayerElement * self)
{
  char msg[300];
  int device = -1, alive = 0;

  if (self->pool == NULL
      || mibayer_pool_take_failure (self->pool, &device, &alive, msg,
          sizeof msg) <= 0)
    return;
  self->capacity = mibayer_pool_capacity (self->pool);
  if (self->failure_note == NULL) {
    self->failure_note = g_strdup (msg);
  } else {
    gchar *both = g_strconcat (self->failure_note, "; ", msg, NULL);

    g_free (self->failure_note);
    self->failure_note = both;
  }
}

/* outside flow_lock: posting may run application code */
static void
element_post_notes (GstMiBayerElement * self)
{
  gchar *note, *text, *debug;
  GQuark domain;
  gint code;

  g_mutex_lock (&self->flow_lock);
  note = self->failure_note;
  self->failure_note = NULL;
  text = self->error_text;
  debug = self->error_debug;
  domain = self->error_domain;
  code = self->error_code;
  self->error_text = self->error_debug = NULL;
  g_mutex_unlock (&self->flow_lock);
  if (text != NULL)             /* takes ownership of both strings */
    gst_element_message_full (GST_ELEMENT (self), GST_MESSAGE_ERROR, domain,
        code, text, debug, __FILE__, GST_FUNCTION, __LINE__);
  if (note != NULL) {
    GST_ELEMENT_WARNING (self, RESOURCE, FAILED,
        ("%s: a GPU failed and was dropped from the rotation; its frames were "
            "converted again on the remaining device(s) (frames in flight on a "
            "GPU that stopped answering are dropped)", LABEL (self)),
        ("%s", note));
    g_free (note);
  }
}

/* lost frames whose device has caught up since (flow_lock held): their buffers are ours again */
static void
element_reclaim_locked (GstMiBayerElement * self)
{
  void *tag = NULL;

  while (self->pool != NULL && !g_queue_is_empty (&self->quarantine)
      && mibayer_pool_reclaim (self->pool, &tag) == MIBAYER_OK) {
    if (g_queue_remove (&self->quarantine, tag))
      pending_release ((PendingFrame *) tag, FALSE);
  }
}

/* the pool is about to go (flow_lock held): whatever is still quarantined can never be handed back -- the device may
 * still write into it -- and is leaked on purpose, mapped and referenced */
static void
element_abandon_quarantine_locked (GstMiBayerElement * self)
{
  element_reclaim_locked (self);
  if (!g_queue_is_empty (&self->quarantine)) {
    EL_WARNING (self, "%u frame(s) lost on a GPU that never answered again: their buffers are leaked",
        g_queue_get_length (&self->quarantine));
    g_queue_clear (&self->quarantine);
  }
}

/* oldest frame (flow_lock held): wait for the GPU, unmap, hand the output buffer
 * back.  *owned tells whether the caller now holds the only reference.  A frame that
 * was in flight on a device that ran into the deadline comes back lost: no output
 * (GST_FLOW_OK, *outbuf == NULL), its buffers go into quarantine. */
static GstFlowReturn
element_collect_locked (GstMiBayerElement * self, GstBuffer ** outbuf,
    gboolean * owned, int *gpu_rc)
{
  PendingFrame *p = g_queue_pop_head (&self->pending);
  int rc;

  *outbuf = NULL;
  *owned = FALSE;
  *gpu_rc = MIBAYER_OK;
  if (p == NULL)
    return GST_FLOW_OK;
  rc = self->pool ? mibayer_pool_wait (self->pool, NULL) : MIBAYER_OK;
  element_note_failures (self);
  element_reclaim_locked (self);
  if (rc == MIBAYER_ERR_TIMEOUT) {
    /* the device may still read the input and write the output: nothing of this frame is released.  In the
     * synchronous mode the base class owns the output buffer and lets go of it when the transform returns: keep
     * a reference of our own, so that it does not go back to its pool */
    if (!p->owns_outbuf) {
      gst_buffer_ref (p->outbuf);
      p->owns_outbuf = TRUE;
    }
    g_queue_push_tail (&self->quarantine, p);
    self->frames_lost++;
    if (mibayer_pool_alive (self->pool) > 0)
      return GST_FLOW_OK;       /* the stream carries on without this frame */
    *gpu_rc = rc;
    return GST_FLOW_ERROR;
  }
  if (rc != MIBAYER_OK) {
    pending_release (p, FALSE);
    *gpu_rc = rc;
    return GST_FLOW_ERROR;
  }
  *outbuf = p->outbuf;
  *owned = p->owns_outbuf;
  pending_release (p, TRUE);
  return GST_FLOW_OK;
}

/* wait for everything in flight; push it downstream (push == TRUE) or drop it.
 * The lock is never held across a push: FLUSH_START must get through while a
 * push blocks downstream. */
static GstFlowReturn
element_drain (GstMiBayerElement * self, gboolean push)
{
  GstFlowReturn ret = GST_FLOW_OK;

  for (;;) {
    GstBuffer *out;
    gboolean owned = TRUE, have;
    int rc = MIBAYER_OK;

    g_mutex_lock (&self->flow_lock);
    out = g_queue_pop_head (&self->ready);
    have = out != NULL;
    if (!have && !g_queue_is_empty (&self->pending)) {
      have = TRUE;
      (void) element_collect_locked (self, &out, &owned, &rc);
    }
    g_mutex_unlock (&self->flow_lock);
    if (!have)
      break;
    element_post_notes (self);
    if (rc != MIBAYER_OK) {
      post_gpu_failure (self, rc);
      ret = GST_FLOW_ERROR;
    } else if (out != NULL && owned) {
      if (push && ret == GST_FLOW_OK && !g_atomic_int_get (&self->flushing))
        ret = gst_pad_push (GST_BASE_TRANSFORM_SRC_PAD (self), out);
      else
        gst_buffer_unref (out);
    }
  }
  return ret;
}

/* what the GPU path cost the host since the pool was created (flow_lock held): GST_DEBUG=<element>:4 */
static void
element_log_host_stats (GstMiBayerElement * self)
{
  mibayer_host_stats st;

  if (self->pool == NULL || mibayer_pool_get_host_stats (self->pool, &st) != MIBAYER_OK || st.submits == 0)
    return;
  EL_INFO (self, "host stats: frames=%" G_GUINT64_FORMAT " submit_cpu_us_per_frame=%.1f wait_cpu_us_per_frame=%.1f "
      "wait_wall_us_per_frame=%.1f polls_per_frame=%.1f naps_per_frame=%.1f lost=%u", st.submits,
      st.submit_cpu_ms * 1e3 / (double) st.submits, st.wait_cpu_ms * 1e3 / (double) st.submits,
      st.wait_wall_ms * 1e3 / (double) st.submits, (double) st.polls / (double) st.submits,
      (double) st.naps / (double) st.submits, self->frames_lost);
}

static void
element_drop_pool (GstMiBayerElement * self)
{
  element_drain (self, FALSE);
  g_mutex_lock (&self->flow_lock);
  if (self->pool) {
    element_log_host_stats (self);
    element_abandon_quarantine_locked (self);
    mibayer_pool_destroy (self->pool);
    self->pool = NULL;
  }
  self->pool_stride = 0;
  self->capacity = 0;
  g_mutex_unlock (&self->flow_lock);
}

/* reference gst_bayer2rgb_reset, gstbayer2rgb.c:278-287 */
static void
element_clear_negotiation (GstMiBayerElement * self)
{
  self->width = 0;
  self->height = 0;
  self->r_off = 0;
  self->g_off = 0;
  self->b_off = 0;
  self->format = MIBAYER_BGGR;
  gst_video_info_init (&self->info);
}

static gboolean
element_parse_devices (GstMiBayerElement * self, mibayer_pool_cfg * pc)
{
  gchar **tok, **t;

  pc->ndevices = 0;
  if (self->act.devices == NULL || self->act.devices[0] == '\0') {
    pc->devices[pc->ndevices++] = self->act.device_id;
    return TRUE;
  }
  tok = g_strsplit_set (self->act.devices, ",;: ", -1);
  for (t = tok; *t != NULL; t++) {
    gchar *end = NULL;
    glong v;

    if (**t == '\0')
      continue;
    v = strtol (*t, &end, 10);
    if (end == *t || *end != '\0' || v < 0
        || pc->ndevices >= MIBAYER_MAX_SHARDS) {
      g_strfreev (tok);
      return FALSE;
    }
    pc->devices[pc->ndevices++] = (int32_t) v;
  }
  g_strfreev (tok);
  return pc->ndevices > 0;
}

static gint
element_ndevices (GstMiBayerElement * self)
{
  mibayer_pool_cfg pc;

  return element_parse_devices (self, &pc) ? pc.ndevices : 1;
}

/* `video_stride` is the mapped stride of the 4-byte-per-pixel frame: the
 * destination stride of bayer2rgb (reference gstbayer2rgb.c:476), the source
 * stride of rgb2bayer (gstrgb2bayer.c:256).  The mosaic rows are always
 * GST_ROUND_UP_4 (width) apart (gstbayer2rgb.c:477, gstrgb2bayer.c:255). */
static gboolean
element_ensure_pool (GstMiBayerElement * self, gint video_stride)
{
  const gboolean inverse = IS_INVERSE (self);
  mibayer_pool_cfg pc;
  int rc;

  if (self->pool && self->pool_stride == video_stride)
    return TRUE;
  /* (flow_lock held) The mapped stride comes with every buffer (GstVideoMeta): it can change without a CAPS
   * event -- a RECONFIGURE, a new downstream pool -- while frames of the old stride are still in flight in queued
   * mode.  They belong to the old pool: finish them, in order, before it goes (their outputs wait in `ready`
   * and leave ahead of the new frame). */
  while (self->pool != NULL && !g_queue_is_empty (&self->pending)) {
    GstBuffer *done = NULL;
    gboolean owned = FALSE;
    int gpu_rc = MIBAYER_OK;

    if (element_collect_locked (self, &done, &owned, &gpu_rc) != GST_FLOW_OK) {
      element_defer_error (self, GST_RESOURCE_ERROR, GST_RESOURCE_ERROR_FAILED,
          g_strdup_printf ("%s: GPU conversion failed", LABEL (self)),
          g_strdup_printf ("%s %s", mibayer_strerror (gpu_rc),
              mibayer_last_hip_error ()));
      continue;
    }
    if (done != NULL && owned)
      g_queue_push_tail (&self->ready, done);
  }
  if (self->pool) {
    element_log_host_stats (self);
    element_abandon_quarantine_locked (self);
    mibayer_pool_destroy (self->pool);
    self->pool = NULL;
  }
  self->pool_stride = 0;
  self->capacity = 0;

  memset (&pc, 0, sizeof pc);
  pc.struct_size = sizeof pc;
  pc.stream.struct_size = sizeof pc.stream;
  pc.stream.width = self->width;
  pc.stream.height = self->height;
  if (inverse) {
    pc.stream.src_stride = video_stride;
    pc.stream.dst_stride = GST_ROUND_UP_4 (self->width);
  } else {
    pc.stream.src_stride = GST_ROUND_UP_4 (self->width);
    pc.stream.dst_stride = video_stride;
  }
  pc.stream.pattern = self->format;
  pc.stream.r_off = self->r_off;
  pc.stream.g_off = self->g_off;
  pc.stream.b_off = self->b_off;
  pc.stream.inflight = self->act.inflight;
  pc.stream.flags = (self->act.use_hipgraph ? MIBAYER_FLAG_HIPGRAPH : 0)
      | (inverse ? MIBAYER_FLAG_RGB2BAYER : 0);
  if (!element_parse_devices (self, &pc)) {
    element_defer_error (self, GST_LIBRARY_ERROR, GST_LIBRARY_ERROR_SETTINGS,
        g_strdup_printf ("%s: cannot parse devices=\"%s\"", LABEL (self),
            self->act.devices), NULL);
    return FALSE;
  }

  rc = mibayer_pool_create (&pc, &self->pool);
  if (rc != MIBAYER_OK) {
    self->pool = NULL;
    if (rc == MIBAYER_ERR_NO_DEVICE) {
      element_defer_error (self, GST_RESOURCE_ERROR, GST_RESOURCE_ERROR_NOT_FOUND,
          g_strdup_printf
          ("%s: no usable MI355X / HIP device (device-id=%d devices=%s)",
              LABEL (self), self->act.device_id,
              self->act.devices ? self->act.devices : ""),
          g_strdup_printf ("%s; this element has no CPU path",
              mibayer_strerror (rc)));
    } else if (rc == MIBAYER_ERR_GEOMETRY) {
      element_defer_error (self, GST_STREAM_ERROR, GST_STREAM_ERROR_FORMAT,
          g_strdup_printf ("%s: unsupported frame geometry %dx%d", LABEL (self),
              self->width, self->height), g_strdup (mibayer_strerror (rc)));
    } else {
      element_defer_error (self, GST_RESOURCE_ERROR, GST_RESOURCE_ERROR_FAILED,
          g_strdup_printf ("%s: cannot create GPU context", LABEL (self)),
          g_strdup_printf ("%s %s", mibayer_strerror (rc),
              mibayer_last_hip_error ()));
    }
    return FALSE;
  }
  /* a GPU that stops answering is dropped like one that reports an error, after this long */
  (void) mibayer_pool_set_wait_timeout (self->pool, self->act.timeout_ms);
  self->pool_stride = video_stride;
  self->capacity = mibayer_pool_capacity (self->pool);
  EL_DEBUG (self, "GPU pool: %d device(s), %d frame(s) in flight, "
      "%dx%d pattern %d stride %d%s", pc.ndevices, self->capacity, self->width,
      self->height, self->format, video_stride,
      self->act.use_hipgraph ? ", hipGraph per frame" : "");
  return TRUE;
}

static inline gboolean
element_is_queued_mode (GstMiBayerElement * self)
{
  return self->act.inflight > 1
      || (self->act.devices != NULL && strchr (self->act.devices, ',') != NULL);
}

/* ---- GObject ----------------------------------------------------------------- */

static void
element_set_property (GObject * object, guint prop_id, const GValue * value,
    GParamSpec * pspec)
{
  GstMiBayerElement *self = ELEMENT (object);

  GST_OBJECT_LOCK (self);
  switch (prop_id) {
    case PROP_DEVICE_ID:
      self->device_id = g_value_get_int (value);
      break;
    case PROP_DEVICES:
      g_free (self->devices);
      self->devices = g_value_dup_string (value);
      break;
    case PROP_INFLIGHT:
      self->inflight = g_value_get_int (value);
      break;
    case PROP_HIPGRAPH:
      self->use_hipgraph = g_value_get_boolean (value);
      break;
    case PROP_PINNED_POOL:
      self->pinned_pool = g_value_get_boolean (value);
      break;
    case PROP_TIMEOUT_MS:
      self->timeout_ms = g_value_get_int (value);
      break;
    default:
      G_OBJECT_WARN_INVALID_PROPERTY_ID (object, prop_id, pspec);
      break;
  }
  GST_OBJECT_UNLOCK (self);
}

static void
element_get_property (GObject * object, guint prop_id, GValue * value,
    GParamSpec * pspec)
{
  GstMiBayerElement *self = ELEMENT (object);

  GST_OBJECT_LOCK (self);
  switch (prop_id) {
    case PROP_DEVICE_ID:
      g_value_set_int (value, self->device_id);
      break;
    case PROP_DEVICES:
      g_value_set_string (value, self->devices);
      break;
    case PROP_INFLIGHT:
      g_value_set_int (value, self->inflight);
      break;
    case PROP_HIPGRAPH:
      g_value_set_boolean (value, self->use_hipgraph);
      break;
    case PROP_PINNED_POOL:
      g_value_set_boolean (value, self->pinned_pool);
      break;
    case PROP_TIMEOUT_MS:
      g_value_set_int (value, self->timeout_ms);
      break;
    default:
      G_OBJECT_WARN_INVALID_PROPERTY_ID (object, prop_id, pspec);
      break;
  }
  GST_OBJECT_UNLOCK (self);
}

static void
element_finalize (GObject * object)
{
  GstMiBayerElement *self = ELEMENT (object);

  element_drop_pool (self);
  g_free (self->devices);
  self->devices = NULL;
  g_free (self->act.devices);
  self->act.devices = NULL;
  g_free (self->failure_note);
  self->failure_note = NULL;
  g_free (self->error_text);
  g_free (self->error_debug);
  self->error_text = self->error_debug = NULL;
  g_mutex_clear (&self->flow_lock);
  G_OBJECT_CLASS (BASE_CLASS (self))->finalize (object);
}

/* ---- caps (identical semantics to the reference) ------------------------------ */

/* reference gstbayer2rgb.c:289-322 and its mirror image gstrgb2bayer.c:128-159:
 * the bayer side and the raw side differ only in the media type name and in the
 * fields that describe the pixel encoding */
static GstCaps *
element_transform_caps (GstBaseTransform * base, GstPadDirection direction,
    GstCaps * caps, GstCaps * filter)
{
  /* caps given for the pad that carries the mosaic -> produce raw caps */
  const gboolean to_raw = (direction == GST_PAD_SINK) != IS_INVERSE (base);
  GstCaps *result = gst_caps_copy (caps);
  guint i, n = gst_caps_get_size (result);

  for (i = 0; i < n; i++) {
    GstStructure *s = gst_caps_get_structure (result, i);

    if (to_raw) {
      gst_structure_set_name (s, "video/x-raw");
      gst_structure_remove_field (s, "format");
    } else {
      gst_structure_set_name (s, "video/x-bayer");
      gst_structure_remove_fields (s, "format", "colorimetry", "chroma-site",
          NULL);
    }
  }
  if (filter) {
    GstCaps *unfiltered = result;

    result = gst_caps_intersect_full (filter, unfiltered,
        GST_CAPS_INTERSECT_FIRST);
    gst_caps_unref (unfiltered);
  }
  EL_DEBUG (base, "transformed %" GST_PTR_FORMAT " into %"
      GST_PTR_FORMAT, caps, result);
  return result;
}

/* reference gstbayer2rgb.c:324-352, gstrgb2bayer.c:161-188 */
static gboolean
element_get_unit_size (GstBaseTransform * base, GstCaps * caps, gsize * size)
{
  GstStructure *s = gst_caps_get_structure (caps, 0);
  gint w, h;

  if (!gst_structure_get_int (s, "width", &w)
      || !gst_structure_get_int (s, "height", &h)) {
    GST_ELEMENT_ERROR (base, CORE, NEGOTIATION, (NULL),
        ("Incomplete caps, some required field missing"));
    return FALSE;
  }
  if (gst_structure_has_name (s, "video/x-raw"))
    *size = (gsize) w * h * 4;            /* always 32 bits per pixel */
  else
    *size = (gsize) GST_ROUND_UP_4 (w) * h;     /* 8-bit mosaic, rows padded to 4 */
  return TRUE;
}

/* reference gstbayer2rgb.c:237-276, gstrgb2bayer.c:190-228 */
static gboolean
element_set_caps (GstBaseTransform * base, GstCaps * incaps, GstCaps * outcaps)
{
  static const struct
  {
    const gchar *name;
    gint pattern;
  } orders[] = {
    {"bggr", MIBAYER_BGGR}, {"gbrg", MIBAYER_GBRG},
    {"grbg", MIBAYER_GRBG}, {"rggb", MIBAYER_RGGB}
  };
  GstMiBayerElement *self = ELEMENT (base);
  const gboolean inverse = IS_INVERSE (self);
  GstCaps *bayer_caps = inverse ? outcaps : incaps;
  GstCaps *raw_caps = inverse ? incaps : outcaps;
  GstStructure *s = gst_caps_get_structure (bayer_caps, 0);
  const gchar *order;
  GstVideoInfo info;
  guint i;

  EL_DEBUG (self, "in caps %" GST_PTR_FORMAT " out caps %"
      GST_PTR_FORMAT, incaps, outcaps);

  gst_structure_get_int (s, "width", &self->width);
  gst_structure_get_int (s, "height", &self->height);
  /* The reference's templates say [1, MAX] for both and its set_caps takes
   * anything, but its frame loop is only well defined for even widths >= 4 and
   * heights >= 3 (it leaves the last column of an odd width unwritten and reads
   * stale scratch for it, gstbayer2rgb.c:365-380; below three rows it reads rows
   * that do not exist, :430-447).  There is nothing bit-exact to reproduce
   * there, so such caps are refused here -- not-negotiated, the reference's own
   * failure style (:263-265) -- instead of at the first buffer.  rgb2bayer has
   * no neighbourhood and takes any size. */
  if (!inverse && (self->width < 4 || (self->width & 1) || self->height < 3)) {
    EL_WARNING (self, "refusing %dx%d: bayer2rgb needs an even width >= 4 and "
        "a height >= 3", self->width, self->height);
    return FALSE;
  }

  order = gst_structure_get_string (s, "format");
  if (order == NULL)
    return FALSE;
  for (i = 0; i < G_N_ELEMENTS (orders); i++) {
    if (g_str_equal (order, orders[i].name))
      break;
  }
  if (i == G_N_ELEMENTS (orders))
    return FALSE;
  self->format = orders[i].pattern;

  /* where R, G and B live inside the 4-byte pixel (rgb2bayer's ARGB: 1, 2, 3 =
   * the hard-coded offsets of gstrgb2bayer.c:259-266) */
  if (!gst_video_info_from_caps (&info, raw_caps))
    return FALSE;
  self->r_off = GST_VIDEO_INFO_COMP_OFFSET (&info, 0);
  self->g_off = GST_VIDEO_INFO_COMP_OFFSET (&info, 1);
  self->b_off = GST_VIDEO_INFO_COMP_OFFSET (&info, 2);
  self->info = info;

  /* geometry changed: the GPU pool is rebuilt on the next buffer, once the
   * mapped video stride is known */
  element_drop_pool (self);
  return TRUE;
}

/* ---- allocation: pinned pools -------------------------------------------------- */

static GstBufferPool *
element_make_pinned_pool (GstMiBayerElement * self, GstCaps * caps, guint size,
    guint min)
{
  GstBufferPool *pool;
  GstStructure *config;

  mibayer_pool_cfg pc;

  if (mibayer_device_count () <= 0)
    return NULL;
  /* Pinned memory next to the GPU that will read / write it: buffer k of the pool sits on the NUMA node of
   * devices[k % N], the device frame k of a `devices=` list goes to ("own pinned host
   * staging" per GPU; on a two-socket node a pool placed next to devices[0] alone puts half of the GPUs behind the
   * socket link).  Reference pattern: sys/nvcodec/gstcudabasetransform.c:301-329. */
  if (element_parse_devices (self, &pc))
    pool = gst_mi_host_pool_new_for_devices (pc.devices, (guint) pc.ndevices);
  else
    pool = gst_mi_host_pool_new (self->act.device_id);
  config = gst_buffer_pool_get_config (pool);
  gst_buffer_pool_config_set_params (config, caps, size, min, 0);
  if (!gst_buffer_pool_set_config (pool, config)) {
    gst_object_unref (pool);
    return NULL;
  }
  EL_DEBUG (self, "pinned pool: %u bytes per buffer, min %u", size,
      min);
  return pool;
}

/* upstream asks how to allocate the buffers it will send us */
static gboolean
element_propose_allocation (GstBaseTransform * base, GstQuery * decide_query,
    GstQuery * query)
{
  GstMiBayerElement *self = ELEMENT (base);
  GstCaps *caps = NULL;
  gboolean need_pool = FALSE;
  gsize size = 0;
  guint min;
  GstBufferPool *pool;

  if (!BASE_CLASS (self)->propose_allocation (base, decide_query, query))
    return FALSE;
  if (!self->act.pinned_pool)
    return TRUE;
  gst_query_parse_allocation (query, &caps, &need_pool);
  if (caps == NULL || !element_get_unit_size (base, caps, &size))
    return TRUE;
  /* every frame in flight keeps its input buffer mapped; a whole number of rounds over the devices, so that a
   * pool that recycles its buffers in order keeps buffer and device aligned */
  min = (guint) ((MAX (self->act.inflight, 1) + 1) * element_ndevices (self)
      + (element_ndevices (self) == 1 ? 1 : 0));
  pool = element_make_pinned_pool (self, caps, (guint) size, min);
  if (pool) {
    mibayer_pool_cfg pc;
    GstAllocationParams params;
    GstAllocator *allocator;

    gst_query_add_allocation_pool (query, pool, (guint) size, min, 0);
    gst_object_unref (pool);
    /* and the allocator behind it, for an upstream that builds its own pool */
    if (element_parse_devices (self, &pc))
      allocator = gst_mi_host_allocator_new_for_devices (pc.devices,
          (guint) pc.ndevices);
    else
      allocator = gst_mi_host_allocator_new (self->act.device_id);
    gst_allocation_params_init (&params);
    gst_query_add_allocation_param (query, allocator, &params);
    gst_object_unref (allocator);
    EL_DEBUG (self, "proposed a pinned input pool upstream");
  }
  return TRUE;
}

/* downstream answered our allocation query: if it brought no pool of its own,
 * allocate the output buffers from pinned memory */
static gboolean
element_decide_allocation (GstBaseTransform * base, GstQuery * query)
{
  GstMiBayerElement *self = ELEMENT (base);

  if (self->act.pinned_pool && gst_query_get_n_allocation_pools (query) == 0) {
    GstCaps *caps = NULL;
    gsize size = 0;

    gst_query_parse_allocation (query, &caps, NULL);
    if (caps != NULL && element_get_unit_size (base, caps, &size)) {
      const guint nd = (guint) element_ndevices (self);
      guint min = (guint) (self->capacity > 0 ? self->capacity + 1 : 2);
      GstBufferPool *pool;

      min = (min + nd - 1) / nd * nd;   /* whole rounds over the devices */
      pool = element_make_pinned_pool (self, caps, (guint) size, min);
      if (pool) {
        gst_query_add_allocation_pool (query, pool, (guint) size, min, 0);
        gst_object_unref (pool);
        EL_DEBUG (self, "using a pinned output pool");
      }
    }
  }
  return BASE_CLASS (self)->decide_allocation (base, query);
}

/* ---- data flow -------------------------------------------------------------------- */

/* map both buffers and hand the frame to the GPU pool (flow_lock held); on
 * success the mapped frame is appended to self->pending */
static GstFlowReturn
element_submit (GstMiBayerElement * self, GstBuffer * inbuf, GstBuffer * outbuf,
    gboolean owns_outbuf, int *gpu_rc)
{
  const gboolean inverse = IS_INVERSE (self);
  GstBuffer *mosaic_buf = inverse ? outbuf : inbuf;
  GstBuffer *video_buf = inverse ? inbuf : outbuf;
  PendingFrame *p = g_new0 (PendingFrame, 1);
  const guint8 *src;
  guint8 *dst;
  int rc;

  *gpu_rc = MIBAYER_OK;
  if (!gst_buffer_map (mosaic_buf, &p->mosaic,
          inverse ? GST_MAP_WRITE : GST_MAP_READ)) {
    g_free (p);
    return GST_FLOW_CUSTOM_ERROR;       /* map failure: see callers */
  }
  if (!gst_video_frame_map (&p->video, &self->info, video_buf,
          inverse ? GST_MAP_READ : GST_MAP_WRITE)) {
    gst_buffer_unmap (mosaic_buf, &p->mosaic);
    g_free (p);
    return GST_FLOW_CUSTOM_ERROR;
  }
  if (p->mosaic.size < (gsize) GST_ROUND_UP_4 (self->width) * self->height) {
    element_defer_error (self, GST_STREAM_ERROR, GST_STREAM_ERROR_FORMAT,
        g_strdup_printf ("%s: short %s buffer", LABEL (self),
            inverse ? "output" : "input"),
        g_strdup_printf ("%" G_GSIZE_FORMAT " bytes for %dx%d", p->mosaic.size,
            self->width, self->height));
    goto fail;
  }
  if (!element_ensure_pool (self, GST_VIDEO_FRAME_PLANE_STRIDE (&p->video, 0)))
    goto fail;

  if (inverse) {
    src = GST_VIDEO_FRAME_PLANE_DATA (&p->video, 0);
    dst = p->mosaic.data;
  } else {
    src = p->mosaic.data;
    dst = GST_VIDEO_FRAME_PLANE_DATA (&p->video, 0);
  }
  for (;;) {
    GstBuffer *done = NULL;
    gboolean owned = FALSE;

    /* the call that replaces gst_bayer2rgb_process (gstbayer2rgb.c:475-477) /
     * the pixel loop of gst_rgb2bayer_transform (gstrgb2bayer.c:254-268) */
    rc = mibayer_pool_submit (self->pool, src, dst, p);
    element_note_failures (self);
    if (rc != MIBAYER_ERR_BUSY || g_queue_is_empty (&self->pending))
      break;
    /* a device was dropped and the pool holds fewer frames now: finish the
     * oldest one to make room; it leaves the element before this one */
    if (element_collect_locked (self, &done, &owned, &rc) != GST_FLOW_OK)
      break;
    if (done != NULL && owned)
      g_queue_push_tail (&self->ready, done);
  }
  if (rc != MIBAYER_OK) {
    *gpu_rc = rc;
    goto fail;
  }
  p->inbuf = gst_buffer_ref (inbuf);
  p->outbuf = outbuf;
  p->mosaic_buf = mosaic_buf;
  p->owns_outbuf = owns_outbuf;
  g_queue_push_tail (&self->pending, p);
  return GST_FLOW_OK;

fail:
  gst_video_frame_unmap (&p->video);
  gst_buffer_unmap (mosaic_buf, &p->mosaic);
  g_free (p);
  return GST_FLOW_ERROR;
}

/* reference gstbayer2rgb.c:456-487, gstrgb2bayer.c:230-278 -- synchronous mode,
 * the default */
static GstFlowReturn
element_transform (GstBaseTransform * base, GstBuffer * inbuf,
    GstBuffer * outbuf)
{
  GstMiBayerElement *self = ELEMENT (base);
  GstFlowReturn ret;
  GstBuffer *done = NULL;
  gboolean owned;
  int rc = MIBAYER_OK;

  EL_DEBUG (self, "transforming buffer");

  g_mutex_lock (&self->flow_lock);
  ret = element_submit (self, inbuf, outbuf, FALSE, &rc);
  if (ret == GST_FLOW_OK)       /* done == outbuf, still owned by the base class */
    ret = element_collect_locked (self, &done, &owned, &rc);
  g_mutex_unlock (&self->flow_lock);
  element_post_notes (self);
  if (ret == GST_FLOW_CUSTOM_ERROR) {
    /* same as the reference: warn and skip (gstbayer2rgb.c:484-486,
     * gstrgb2bayer.c:274-276) */
    EL_WARNING (self, "Could not map buffer, skipping");
    return GST_FLOW_OK;
  }
  if (rc != MIBAYER_OK)
    post_gpu_failure (self, rc);
  if (ret == GST_FLOW_OK && done == NULL)
    return GST_BASE_TRANSFORM_FLOW_DROPPED;     /* lost on a GPU that stopped answering: no output for this input */
  return ret;
}

/* queued mode: take the input the base class parked in queued_buf, submit it,
 * and release the oldest frame once the pool is full */
static GstFlowReturn
element_generate_output (GstBaseTransform * base, GstBuffer ** outbuf)
{
  GstMiBayerElement *self = ELEMENT (base);
  GstBaseTransformClass *klass = GST_BASE_TRANSFORM_GET_CLASS (base);
  GstFlowReturn ret = GST_FLOW_OK;
  GstBuffer *inbuf;
  gboolean owned;
  int rc = MIBAYER_OK;

  if (!element_is_queued_mode (self))
    return BASE_CLASS (self)->generate_output (base, outbuf);

  *outbuf = NULL;
  inbuf = base->queued_buf;
  base->queued_buf = NULL;
  if (g_atomic_int_get (&self->flushing)) {
    if (inbuf != NULL)
      gst_buffer_unref (inbuf);
    return GST_FLOW_FLUSHING;
  }
  if (inbuf != NULL) {
    GstBuffer *out = NULL;

    ret = klass->prepare_output_buffer (base, inbuf, &out);
    if (ret != GST_FLOW_OK || out == NULL) {
      gst_buffer_unref (inbuf);
      return ret == GST_FLOW_OK ? GST_FLOW_ERROR : ret;
    }
    g_mutex_lock (&self->flow_lock);
    ret = element_submit (self, inbuf, out, TRUE, &rc);
    g_mutex_unlock (&self->flow_lock);
    gst_buffer_unref (inbuf);           /* the pending entry holds its own ref */
    element_post_notes (self);
    if (ret == GST_FLOW_CUSTOM_ERROR) {
      EL_WARNING (self, "Could not map buffer, skipping");
      gst_buffer_unref (out);
      return GST_FLOW_OK;
    }
    if (ret != GST_FLOW_OK) {
      if (rc != MIBAYER_OK)
        post_gpu_failure (self, rc);
      gst_buffer_unref (out);
      return ret;
    }
  }
  /* the base class calls again for as long as a buffer comes out */
  g_mutex_lock (&self->flow_lock);
  *outbuf = g_queue_pop_head (&self->ready);
  /* The first frame after a start or a flush is not held back: sinks preroll on
   * it, and holding it until the pool is full can deadlock against upstream
   * queues that fill up while another branch's sink sits prerolled. */
  if (*outbuf == NULL && !g_queue_is_empty (&self->pending)
      && (!self->prerolled || (self->capacity > 0
              && (gint) g_queue_get_length (&self->pending) >= self->capacity)))
    ret = element_collect_locked (self, outbuf, &owned, &rc);
  if (*outbuf != NULL)
    self->prerolled = TRUE;
  g_mutex_unlock (&self->flow_lock);
  element_post_notes (self);
  if (rc != MIBAYER_OK)
    post_gpu_failure (self, rc);
  return ret;
}

static gboolean
element_sink_event (GstBaseTransform * base, GstEvent * event)
{
  GstMiBayerElement *self = ELEMENT (base);

  switch (GST_EVENT_TYPE (event)) {
    case GST_EVENT_CAPS:
    case GST_EVENT_EOS:
    case GST_EVENT_SEGMENT:
    case GST_EVENT_GAP:
      /* frames in flight precede the event */
      element_drain (self, TRUE);
      break;
    case GST_EVENT_FLUSH_START:
      /* not serialised with the streaming thread: from here on nothing is
       * submitted or pushed, and what is in flight is dropped as soon as the GPU
       * lets go of the buffers (a DMA in progress cannot be recalled, but no
       * frame waits for a push that the flush has made pointless) */
      g_atomic_int_set (&self->flushing, 1);
      element_drain (self, FALSE);
      break;
    case GST_EVENT_FLUSH_STOP:
      /* serialised: whatever the streaming thread still managed to queue */
      element_drain (self, FALSE);
      g_atomic_int_set (&self->flushing, 0);
      self->prerolled = FALSE;
      break;
    default:
      break;
  }
  return BASE_CLASS (self)->sink_event (base, event);
}

/* queued mode holds up to `capacity` frames back: report that to live pipelines */
static gboolean
element_query (GstBaseTransform * base, GstPadDirection direction,
    GstQuery * query)
{
  GstMiBayerElement *self = ELEMENT (base);

  if (direction == GST_PAD_SRC && GST_QUERY_TYPE (query) == GST_QUERY_LATENCY
      && element_is_queued_mode (self)) {
    gboolean live = FALSE;
    GstClockTime min = 0, max = GST_CLOCK_TIME_NONE;
    gint fps_n = GST_VIDEO_INFO_FPS_N (&self->info);
    gint fps_d = GST_VIDEO_INFO_FPS_D (&self->info);

    if (!gst_pad_peer_query (GST_BASE_TRANSFORM_SINK_PAD (base), query))
      return FALSE;
    gst_query_parse_latency (query, &live, &min, &max);
    if (fps_n > 0 && fps_d > 0) {
      GstClockTime held =
          gst_util_uint64_scale_int (GST_SECOND * (guint64) (self->act.inflight
              * element_ndevices (self)), fps_d, fps_n);

      min += held;
      if (GST_CLOCK_TIME_IS_VALID (max))
        max += held;
      EL_DEBUG (self, "queued mode adds %" GST_TIME_FORMAT " latency",
          GST_TIME_ARGS (held));
    }
    gst_query_set_latency (query, live, min, max);
    return TRUE;
  }
  return BASE_CLASS (self)->query (base, direction, query);
}

/* READY -> PAUSED: latch the properties for the streaming thread */
static gboolean
element_start (GstBaseTransform * base)
{
  GstMiBayerElement *self = ELEMENT (base);

  GST_OBJECT_LOCK (self);
  self->act.device_id = self->device_id;
  g_free (self->act.devices);
  self->act.devices = g_strdup (self->devices);
  self->act.inflight = self->inflight;
  self->act.use_hipgraph = self->use_hipgraph;
  self->act.pinned_pool = self->pinned_pool;
  self->act.timeout_ms = self->timeout_ms;
  GST_OBJECT_UNLOCK (self);
  g_atomic_int_set (&self->flushing, 0);
  self->prerolled = FALSE;
  self->frames_lost = 0;
  return TRUE;
}

static gboolean
element_stop (GstBaseTransform * base)
{
  element_drop_pool (ELEMENT (base));
  return TRUE;
}

/* ---- set-up called by the two concrete types -------------------------------------- */

void
gst_mi_bayer_element_class_setup (GstMiBayerElementClass * klass,
    gboolean inverse, const gchar * label)
{
  GObjectClass *object_class = G_OBJECT_CLASS (klass);
  GstBaseTransformClass *transform_class = GST_BASE_TRANSFORM_CLASS (klass);

  klass->cat = NULL;
  GST_DEBUG_CATEGORY_INIT (klass->cat, label, 0,
      inverse ? "rgb2bayer element" : "bayer2rgb element");
  klass->base_class = g_type_class_peek_parent (klass);
  klass->inverse = inverse;
  klass->label = label;

  object_class->set_property = element_set_property;
  object_class->get_property = element_get_property;
  object_class->finalize = element_finalize;

  g_object_class_install_property (object_class, PROP_DEVICE_ID,
      g_param_spec_int ("device-id", "Device ID",
          "HIP ordinal of the MI355X that converts this stream (latched when "
          "the element starts)", 0, G_MAXINT,
          DEFAULT_DEVICE_ID, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (object_class, PROP_DEVICES,
      g_param_spec_string ("devices", "Devices",
          "Comma-separated HIP ordinals; frames are sharded round-robin over "
          "them (frame g -> devices[g % N]); a device that fails is dropped and "
          "its frames are redone on the others; empty = device-id only "
          "(latched when the element starts)", NULL,
          G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (object_class, PROP_INFLIGHT,
      g_param_spec_int ("inflight", "Frames in flight",
          "Frames in flight per device; 1 = strictly synchronous 1-in/1-out "
          "like the stock element, more = queued mode (adds latency; latched "
          "when the element starts)", 1, 16,
          DEFAULT_INFLIGHT, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (object_class, PROP_HIPGRAPH,
      g_param_spec_boolean ("hipgraph", "hipGraph-captured launch",
          "Replay the compute-queue segment of every frame (wait for the upload, "
          "kernel, signal the download) as a hipGraph captured once per ring "
          "slot; the pinned H2D / D2H copies stay asynchronous copies on the two "
          "copy queues (bayer2rgb direction)", DEFAULT_HIPGRAPH,
          G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (object_class, PROP_PINNED_POOL,
      g_param_spec_boolean ("pinned-pool", "Pinned buffer pools",
          "Propose hipHostMalloc-pinned buffer pools upstream and use them "
          "downstream when no other pool is offered", DEFAULT_PINNED_POOL,
          G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));

  g_object_class_install_property (object_class, PROP_TIMEOUT_MS,
      g_param_spec_int ("timeout-ms", "GPU wait deadline",
          "Milliseconds a GPU may take to hand a frame back before it counts as "
          "failed: it is dropped from the rotation like a device that reported an "
          "error and is not waited for again (with one device the stream errors "
          "out instead of hanging); 0 = wait for ever (latched when the element "
          "starts)", 0, 3600000, DEFAULT_TIMEOUT_MS,
          G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));

  transform_class->transform_caps = GST_DEBUG_FUNCPTR (element_transform_caps);
  transform_class->get_unit_size = GST_DEBUG_FUNCPTR (element_get_unit_size);
  transform_class->set_caps = GST_DEBUG_FUNCPTR (element_set_caps);
  transform_class->transform = GST_DEBUG_FUNCPTR (element_transform);
  transform_class->generate_output =
      GST_DEBUG_FUNCPTR (element_generate_output);
  transform_class->sink_event = GST_DEBUG_FUNCPTR (element_sink_event);
  transform_class->query = GST_DEBUG_FUNCPTR (element_query);
  transform_class->propose_allocation =
      GST_DEBUG_FUNCPTR (element_propose_allocation);
  transform_class->decide_allocation =
      GST_DEBUG_FUNCPTR (element_decide_allocation);
  transform_class->start = GST_DEBUG_FUNCPTR (element_start);
  transform_class->stop = GST_DEBUG_FUNCPTR (element_stop);
}

void
gst_mi_bayer_element_instance_setup (GstMiBayerElement * self)
{
  element_clear_negotiation (self);
  self->device_id = DEFAULT_DEVICE_ID;
  self->devices = NULL;
  self->inflight = DEFAULT_INFLIGHT;
  self->use_hipgraph = DEFAULT_HIPGRAPH;
  self->pinned_pool = DEFAULT_PINNED_POOL;
  self->timeout_ms = DEFAULT_TIMEOUT_MS;
  self->act.timeout_ms = DEFAULT_TIMEOUT_MS;
  self->error_text = self->error_debug = NULL;
  self->act.device_id = DEFAULT_DEVICE_ID;
  self->act.devices = NULL;
  self->act.inflight = DEFAULT_INFLIGHT;
  self->act.use_hipgraph = DEFAULT_HIPGRAPH;
  self->act.pinned_pool = DEFAULT_PINNED_POOL;
  self->pool = NULL;
  self->pool_stride = 0;
  self->capacity = 0;
  self->flushing = 0;
  self->prerolled = FALSE;
  self->failure_note = NULL;
  g_mutex_init (&self->flow_lock);
  g_queue_init (&self->pending);
  g_queue_init (&self->ready);
  g_queue_init (&self->quarantine);
}
