/* GstHarness / pipeline driver for the MI355X bayer plugin, used by tests/test_gst_harness.py.
 *
 * Follows the reference's element-test conventions: GstHarness with hand-built buffers and byte-for-byte
 * comparison (tests/check/elements/vkcolorconvert.c:61-113), state cycling of a whole pipeline
 * (tests/check/generic/states.c:106-216), no fork after the GPU runtime is initialised
 * (tests/check/elements/cudaconvert.c:194-195).
 *
 *   element_harness convert <launchline> <sinkcaps> <in.raw> <framebytes> <out.raw>
 *        push every frame of in.raw through the element, then EOS; write every buffer that comes out
 *   element_harness flush <launchline> <sinkcaps> <in.raw> <framebytes> <out.raw> <nbefore>
 *        push <nbefore> frames, FLUSH_START/FLUSH_STOP, push the rest, EOS; write what comes out
 *   element_harness renegotiate <launchline> <caps1> <in1.raw> <framebytes1> <caps2> <in2.raw> <framebytes2> <out.raw>
 *        push every frame of in1.raw under caps1, then switch to caps2 mid-stream (a new CAPS event) and push every
 *        frame of in2.raw, then EOS; write every buffer that comes out (frames in flight precede the new caps)
 *   element_harness states <pipeline> <cycles>
 *        NULL -> PLAYING -> (EOS) -> NULL, <cycles> times, on ONE pipeline instance
 *   element_harness caps <launchline> <sinkcaps> <framebytes>
 *        CAPS event + one buffer; prints caps_accepted=0|1 flow=<name> (set_caps refusing a geometry = not-negotiated)
 *   element_harness allocation <launchline> <sinkcaps>
 *        ALLOCATION query to the element's sink pad; uses the proposed allocator and pool with a prefix / padding /
 *        64-byte alignment; prints pools= params= alloc_ok= pool_ok=
 *
 * convert / flush also print warnings=<n> errors=<n> (element messages seen on a private bus) and, for flush,
 * released_at_flush_start=0|1: whether every input buffer pushed before the flush had been let go of by the element
 * when FLUSH_START returned (i.e. before FLUSH_STOP).  HARNESS_SET_MIDSTREAM="prop=value[,prop=value]" sets
 * properties on the element after the first buffer (they must not disturb the running stream: latched at start).
 * HARNESS_RESTRIDE="nbefore:width:height:stride" (4-byte-per-pixel input, i.e. rgb2bayer): from frame <nbefore> on
 * the input rows are <stride> bytes apart and the buffer says so in a GstVideoMeta -- a stride that changes
 * mid-stream WITHOUT a caps event, as after a RECONFIGURE.
 */
#include <gst/gst.h>
#include <gst/check/gstharness.h>
#include <gst/video/video.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int
drain_to_file (GstHarness * h, FILE * out)
{
  GstBuffer *buf;
  int n = 0;

  while ((buf = gst_harness_try_pull (h)) != NULL) {
    GstMapInfo map;

    if (!gst_buffer_map (buf, &map, GST_MAP_READ))
      return -1;
    fwrite (map.data, 1, map.size, out);
    gst_buffer_unmap (buf, &map);
    gst_buffer_unref (buf);
    n++;
  }
  return n;
}

static GstBus *
attach_bus (GstHarness * h)
{
  GstBus *bus = gst_bus_new ();

  gst_element_set_bus (h->element, bus);
  return bus;
}

static void
report_bus (GstBus * bus)
{
  GstMessage *msg;
  int warnings = 0, errors = 0;

  while ((msg = gst_bus_pop (bus)) != NULL) {
    if (GST_MESSAGE_TYPE (msg) == GST_MESSAGE_WARNING) {
      GError *err = NULL;
      gchar *dbg = NULL;

      gst_message_parse_warning (msg, &err, &dbg);
      fprintf (stderr, "bus warning: %s (%s)\n", err->message, dbg ? dbg : "");
      g_clear_error (&err);
      g_free (dbg);
      warnings++;
    } else if (GST_MESSAGE_TYPE (msg) == GST_MESSAGE_ERROR) {
      GError *err = NULL;
      gchar *dbg = NULL;

      gst_message_parse_error (msg, &err, &dbg);
      fprintf (stderr, "bus error: %s (%s)\n", err->message, dbg ? dbg : "");
      g_clear_error (&err);
      g_free (dbg);
      errors++;
    }
    gst_message_unref (msg);
  }
  fprintf (stdout, "warnings=%d errors=%d\n", warnings, errors);
}

static void
set_midstream_properties (GstHarness * h)
{
  const gchar *spec = g_getenv ("HARNESS_SET_MIDSTREAM");
  gchar **kv, **p;

  if (!spec)
    return;
  kv = g_strsplit (spec, ";", -1);
  for (p = kv; *p; p++) {
    gchar **pair = g_strsplit (*p, "=", 2);

    if (pair[0] && pair[1])
      gst_util_set_object_arg (G_OBJECT (h->element), pair[0], pair[1]);
    g_strfreev (pair);
  }
  g_strfreev (kv);
}

static int
run_caps (char **argv)
{
  GstHarness *h = gst_harness_new_parse (argv[2]);
  GstBus *bus;
  GstFlowReturn flow;
  gsize frame_bytes = (gsize) atol (argv[4]);

  if (!h)
    return 2;
  bus = attach_bus (h);
  /* A CAPS event is sticky: pushing it reports success whatever set_caps says, and the refusal surfaces as
   * not-negotiated at the first buffer -- which is what a pipeline sees too. */
  gst_harness_set_src_caps_str (h, argv[3]);
  flow = gst_harness_push (h, gst_buffer_new_allocate (NULL, frame_bytes, NULL));
  fprintf (stdout, "caps_accepted=%d flow=%s\n", flow == GST_FLOW_OK ? 1 : 0, gst_flow_get_name (flow));
  report_bus (bus);
  gst_element_set_bus (h->element, NULL);
  gst_object_unref (bus);
  gst_harness_teardown (h);
  return 0;
}

static int
run_harness (int argc, char **argv, int flush_after)
{
  const char *launch = argv[2], *caps = argv[3], *in_path = argv[4];
  size_t frame_bytes = (size_t) atol (argv[5]);
  const char *out_path = argv[6];
  GstHarness *h = gst_harness_new_parse (launch);
  FILE *in = fopen (in_path, "rb"), *out = fopen (out_path, "wb");
  guint8 *frame = g_malloc (frame_bytes);
  int pushed = 0, pulled = 0, n;
  GstBus *bus;
  GstBuffer *held[64];

  if (!h || !in || !out) {
    fprintf (stderr, "setup failed\n");
    return 2;
  }
  int rs_before = -1, rs_w = 0, rs_h = 0, rs_stride = 0;

  if (g_getenv ("HARNESS_RESTRIDE"))
    sscanf (g_getenv ("HARNESS_RESTRIDE"), "%d:%d:%d:%d", &rs_before, &rs_w, &rs_h, &rs_stride);
  bus = attach_bus (h);
  gst_harness_set_src_caps_str (h, caps);
  while (fread (frame, 1, frame_bytes, in) == frame_bytes) {
    GstBuffer *buf;

    if (rs_before >= 0 && pushed >= rs_before) {
      gsize offset[GST_VIDEO_MAX_PLANES] = { 0, };
      gint stride[GST_VIDEO_MAX_PLANES] = { rs_stride, };
      int y;

      buf = gst_buffer_new_allocate (NULL, (gsize) rs_stride * rs_h, NULL);
      gst_buffer_memset (buf, 0, 0xEE, (gsize) rs_stride * rs_h);
      for (y = 0; y < rs_h; y++)
        gst_buffer_fill (buf, (gsize) y * rs_stride, frame + (size_t) y * 4 * rs_w, (gsize) 4 * rs_w);
      gst_buffer_add_video_meta_full (buf, GST_VIDEO_FRAME_FLAG_NONE, GST_VIDEO_FORMAT_ARGB, rs_w, rs_h, 1,
          offset, stride);
    } else {
      buf = gst_buffer_new_allocate (NULL, frame_bytes, NULL);
      gst_buffer_fill (buf, 0, frame, frame_bytes);
    }
    GST_BUFFER_PTS (buf) = (GstClockTime) pushed * GST_SECOND / 30;
    if (flush_after > 0 && pushed < flush_after && pushed < 64)
      held[pushed] = gst_buffer_ref (buf);      /* to see when the element lets go of it */
    if (gst_harness_push (h, buf) != GST_FLOW_OK) {
      fprintf (stderr, "push %d failed\n", pushed);
      report_bus (bus);
      return 3;
    }
    pushed++;
    if (pushed == 1)
      set_midstream_properties (h);
    if (flush_after > 0 && pushed == flush_after) {
      /* everything still inside the element must be dropped, nothing may leak out later */
      int before = drain_to_file (h, out), i, released = 1;
      gst_harness_push_event (h, gst_event_new_flush_start ());
      /* FLUSH_START alone must already have dropped the frames in flight: ours is the only reference left */
      for (i = 0; i < flush_after && i < 64; i++) {
        if (GST_MINI_OBJECT_REFCOUNT_VALUE (held[i]) != 1)
          released = 0;
        gst_buffer_unref (held[i]);
      }
      fprintf (stdout, "released_at_flush_start=%d\n", released);
      gst_harness_push_event (h, gst_event_new_flush_stop (TRUE));
      fprintf (stdout, "before_flush_pulled=%d\n", before);
      pulled += before;
      /* a segment must follow a flush before new buffers */
      {
        GstSegment seg;
        gst_segment_init (&seg, GST_FORMAT_TIME);
        gst_harness_push_event (h, gst_event_new_segment (&seg));
      }
    }
    n = drain_to_file (h, out);
    if (n < 0)
      return 4;
    pulled += n;
  }
  gst_harness_push_event (h, gst_event_new_eos ());
  n = drain_to_file (h, out);
  pulled += n;
  fprintf (stdout, "pushed=%d pulled=%d\n", pushed, pulled);
  report_bus (bus);
  fclose (in);
  fclose (out);
  g_free (frame);
  gst_element_set_bus (h->element, NULL);
  gst_object_unref (bus);
  gst_harness_teardown (h);
  return 0;
}

static int
run_renegotiate (char **argv)
{
  const char *launch = argv[2], *out_path = argv[9];
  GstHarness *h = gst_harness_new_parse (launch);
  FILE *out = fopen (out_path, "wb");
  int pushed = 0, pulled = 0, seg, n;

  if (!h || !out) {
    fprintf (stderr, "setup failed\n");
    return 2;
  }
  for (seg = 0; seg < 2; seg++) {
    const char *caps = argv[3 + 3 * seg], *in_path = argv[4 + 3 * seg];
    size_t frame_bytes = (size_t) atol (argv[5 + 3 * seg]);
    FILE *in = fopen (in_path, "rb");
    guint8 *frame = g_malloc (frame_bytes);

    if (!in)
      return 2;
    gst_harness_set_src_caps_str (h, caps);     /* CAPS event; the second one arrives mid-stream */
    while (fread (frame, 1, frame_bytes, in) == frame_bytes) {
      GstBuffer *buf = gst_buffer_new_allocate (NULL, frame_bytes, NULL);

      gst_buffer_fill (buf, 0, frame, frame_bytes);
      GST_BUFFER_PTS (buf) = (GstClockTime) pushed * GST_SECOND / 30;
      if (gst_harness_push (h, buf) != GST_FLOW_OK) {
        fprintf (stderr, "push %d failed\n", pushed);
        return 3;
      }
      pushed++;
      if ((n = drain_to_file (h, out)) < 0)
        return 4;
      pulled += n;
    }
    fclose (in);
    g_free (frame);
  }
  gst_harness_push_event (h, gst_event_new_eos ());
  pulled += drain_to_file (h, out);
  fprintf (stdout, "pushed=%d pulled=%d\n", pushed, pulled);
  fclose (out);
  gst_harness_teardown (h);
  return 0;
}

static int
run_states (const char *desc, int cycles)
{
  GError *err = NULL;
  GstElement *pipe = gst_parse_launch (desc, &err);
  GstBus *bus;
  int c;

  if (!pipe) {
    fprintf (stderr, "parse: %s\n", err ? err->message : "?");
    return 2;
  }
  bus = gst_element_get_bus (pipe);
  for (c = 0; c < cycles; c++) {
    GstMessage *msg;

    if (gst_element_set_state (pipe, GST_STATE_PLAYING) == GST_STATE_CHANGE_FAILURE) {
      fprintf (stderr, "cycle %d: cannot go to PLAYING\n", c);
      return 3;
    }
    msg = gst_bus_timed_pop_filtered (bus, 60 * GST_SECOND,
        GST_MESSAGE_EOS | GST_MESSAGE_ERROR);
    if (!msg || GST_MESSAGE_TYPE (msg) == GST_MESSAGE_ERROR) {
      if (msg) {
        gchar *dbg = NULL;
        gst_message_parse_error (msg, &err, &dbg);
        fprintf (stderr, "cycle %d: %s (%s)\n", c, err->message, dbg ? dbg : "");
      } else
        fprintf (stderr, "cycle %d: timeout\n", c);
      return 4;
    }
    gst_message_unref (msg);
    gst_element_set_state (pipe, GST_STATE_NULL);
    gst_bus_set_flushing (bus, TRUE);
    gst_bus_set_flushing (bus, FALSE);
  }
  fprintf (stdout, "cycles_ok=%d\n", cycles);
  gst_object_unref (bus);
  gst_object_unref (pipe);
  return 0;
}

/* allocation <launch> <caps>: what the element proposes upstream.  Sends an ALLOCATION query to its sink pad, then
 * uses what came back the way an upstream element would: a memory from the proposed allocator with a prefix, a
 * padding and a 64-byte alignment, and a buffer from the proposed pool configured with the same parameters. */
static int
check_memory (GstMemory * mem, gsize size, gsize prefix, gsize padding)
{
  GstMapInfo map;
  gsize offset = 0, maxsize = 0;
  int ok;

  if (mem == NULL || gst_memory_get_sizes (mem, &offset, &maxsize) != size)
    return 0;
  ok = offset >= prefix && maxsize >= offset + size + padding;
  if (!gst_memory_map (mem, &map, GST_MAP_WRITE))
    return 0;
  ok = ok && map.size == size && (((gsize) map.data) & 63) == 0;
  memset (map.data, 0x5a, map.size);
  ok = ok && map.data[-1] == 0 && map.data[size] == 0;  /* ZERO_PREFIXED / ZERO_PADDED */
  gst_memory_unmap (mem, &map);
  return ok;
}

static int
run_allocation (char **argv)
{
  GstHarness *h = gst_harness_new_parse (argv[2]);
  GstCaps *caps = gst_caps_from_string (argv[3]);
  GstQuery *query;
  GstAllocationParams params;
  guint npools, nparams;
  int alloc_ok = 0, pool_ok = 0;

  if (!h || !caps)
    return 2;
  gst_harness_set_src_caps_str (h, argv[3]);
  query = gst_query_new_allocation (caps, TRUE);
  if (!gst_pad_peer_query (h->srcpad, query)) {
    fprintf (stdout, "query=0\n");
    return 0;
  }
  npools = gst_query_get_n_allocation_pools (query);
  nparams = gst_query_get_n_allocation_params (query);
  gst_allocation_params_init (&params);
  params.align = 63;
  params.prefix = 16;
  params.padding = 8;
  params.flags = GST_MEMORY_FLAG_ZERO_PREFIXED | GST_MEMORY_FLAG_ZERO_PADDED;
  if (nparams > 0) {
    GstAllocator *allocator = NULL;
    GstMemory *mem;

    gst_query_parse_nth_allocation_param (query, 0, &allocator, NULL);
    if (allocator != NULL) {
      mem = gst_allocator_alloc (allocator, 1000, &params);
      alloc_ok = check_memory (mem, 1000, 16, 8);
      if (mem)
        gst_memory_unref (mem);
      gst_object_unref (allocator);
    }
  }
  if (npools > 0) {
    GstBufferPool *pool = NULL;
    guint size = 0, min = 0, max = 0;

    gst_query_parse_nth_allocation_pool (query, 0, &pool, &size, &min, &max);
    if (pool != NULL) {
      GstStructure *config = gst_buffer_pool_get_config (pool);
      GstBuffer *buf = NULL;

      gst_buffer_pool_config_set_params (config, caps, size, min, max);
      gst_buffer_pool_config_set_allocator (config, NULL, &params);
      if (gst_buffer_pool_set_config (pool, config) && gst_buffer_pool_set_active (pool, TRUE)
          && gst_buffer_pool_acquire_buffer (pool, &buf, NULL) == GST_FLOW_OK) {
        pool_ok = gst_buffer_n_memory (buf) == 1 && check_memory (gst_buffer_peek_memory (buf, 0), size, 16, 8);
        gst_buffer_unref (buf);
      }
      gst_buffer_pool_set_active (pool, FALSE);
      gst_object_unref (pool);
    }
  }
  fprintf (stdout, "query=1 pools=%u params=%u alloc_ok=%d pool_ok=%d\n", npools, nparams, alloc_ok, pool_ok);
  gst_query_unref (query);
  gst_caps_unref (caps);
  gst_harness_teardown (h);
  return 0;
}

int
main (int argc, char **argv)
{
  gst_init (&argc, &argv);
  if (argc >= 7 && strcmp (argv[1], "convert") == 0)
    return run_harness (argc, argv, 0);
  if (argc >= 8 && strcmp (argv[1], "flush") == 0)
    return run_harness (argc, argv, atoi (argv[7]));
  if (argc >= 10 && strcmp (argv[1], "renegotiate") == 0)
    return run_renegotiate (argv);
  if (argc >= 5 && strcmp (argv[1], "caps") == 0)
    return run_caps (argv);
  if (argc >= 4 && strcmp (argv[1], "allocation") == 0)
    return run_allocation (argv);
  if (argc >= 4 && strcmp (argv[1], "states") == 0)
    return run_states (argv[2], atoi (argv[3]));
  fprintf (stderr, "usage: see the header of element_harness.c\n");
  return 64;
}
