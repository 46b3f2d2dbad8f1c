/* TEST DOUBLE of the per-device part of libmibayer.so -- NOT a conversion path, NOT shipped, NOT a fallback.
 *
 * The elements of plugin `bayer` (gst-plugins-bad_amd/gst/gstmibayerelement.c) reach the GPU through the
 * frame-sharding pool of include/mibayer.h; those of plugin `mihip` through the device-memory entry points.  This
 * file implements the PER-DEVICE CONTEXT below the pool (mibayer_create / submit / wait / ..., the seam of
 * csrc/mibayer_hooks.h) and the device-memory entry points with NO demosaic in them; the pool on top is the real
 * csrc/mibayer_pool.cpp, compiled into the same test library.  So the elements' own logic -- buffer ownership in the
 * synchronous and the queued mode, ordering, draining on EOS / caps / segment events, dropping on flush, pool
 * re-creation on renegotiation -- AND the pool's ordering, failover and helper threads can be exercised on a machine
 * without a GPU, under AddressSanitizer / ThreadSanitizer (tests/test_gst_element_logic.py, tests/test_pool_logic.py).
 *
 * What a "conversion" does here: nothing at submit time; at wait time (the moment the real library would have
 * finished its asynchronous work) it READS every source byte and WRITES every destination byte the real kernel
 * would write, so a buffer the element unmapped or released too early is a sanitizer report.  The output is a
 * stamp, not an image: bytes 0..3 of the frame = submission sequence number, every other written byte = the
 * first source byte of that frame.  MOCK_MIBAYER_FAIL="index:frames" turns the index-th context into a failed
 * device after that many frames; MOCK_MIBAYER_HANG="index:frames" into one that stops answering (its waits run into
 * the deadline of mibayer_set_wait_timeout and return MIBAYER_ERR_TIMEOUT; anything that would block on it for
 * ever -- a wait with no deadline, a destroy of a context that was never abandoned -- aborts the test);
 * MOCK_MIBAYER_PAGEABLE=1 sends every frame down the pool's helper-thread path; MOCK_MIBAYER_NUMA_NODES=k spreads
 * the fake devices over k NUMA nodes (device d -> node d % k) and mibayer_host_alloc_near() remembers where it
 * "placed" a block.  A failed context that still holds frames and was NOT abandoned writes them LATE, when it is
 * destroyed -- the DMA that a dead device's queue may still carry out -- so a pool that hands a re-done frame back
 * before abandoning the context it came from is a sanitizer report (the driver frees every frame on delivery).
 * A context that stopped answering KEEPS the frames it holds (mibayer_internal_abandon returns
 * MIBAYER_ERR_TIMEOUT for it) and carries them out late as well: when mibayer_internal_settled has been asked
 * MOCK_MIBAYER_RESUME_POLLS times (default 3; -1 = never) the "device" resumes, reads their sources and writes their
 * destinations, and only then reports settled -- or at destroy time at the latest.  A pool that converts such a
 * frame again into the same buffers and hands it back, or a caller that releases the buffers of a lost frame before
 * mibayer_pool_reclaim returned its tag, is a sanitizer report or a wrong stamp.
 */
#include "mibayer.h"

#include <stdlib.h>
#include <string.h>

#include <pthread.h>
#include <stdio.h>
#include <time.h>

#include "mibayer_hooks.h"

#define MOCK_MAX_PENDING 64

/* The frame-sharding pool on top of these contexts is the REAL one: csrc/mibayer_pool.cpp (pure host logic) is
 * compiled into the test double's libmibayer.so, so its ordering, failover and helper-thread code runs here
 * under the sanitizers.  Only the per-device context below is fake. */

typedef struct
{
  const uint8_t *src;
  uint8_t *dst;
  void *tag;
  uint32_t seq;
} mock_frame;

static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;     /* elements and helper threads run concurrently */
static uint32_t g_seq;          /* host-path frames submitted since the first context of a pool was created */
static int g_live_host_ctx;     /* contexts alive: the counter restarts with every new pool */
static int g_created;           /* contexts created so far == index of the next one (fault injection) */
static int g_plan_cached;       /* a context of this process ran mibayer_autotune_list: later ones report "cached" */
static int g_autotunes;

struct mibayer_ctx
{
  mibayer_cfg cfg;
  size_t src_bytes, dst_bytes;
  /* host path: a ring like the real context's */
  mock_frame ring[MOCK_MAX_PENDING];
  int head, count;
  int index;                    /* n-th context created in this process */
  int completed;                /* frames finished on this "device" */
  int fail_after;               /* MOCK_MIBAYER_FAIL=index:frames -> device error after that many frames; -1 never */
  int dead;
  int hang_after;               /* MOCK_MIBAYER_HANG=index:frames -> stops answering after that many frames; -1 never */
  int hung;
  int timeout_ms;               /* mibayer_set_wait_timeout; 0 = none */
  int abandoned;
  int settle_polls;             /* mibayer_internal_settled calls since it stopped answering */
  int plan_source;              /* MIBAYER_PLAN_* */
};

/* fake NUMA placement: blocks handed out by mibayer_host_alloc_near */
#define MOCK_MAX_BLOCKS 4096
static struct
{
  const uint8_t *base;
  size_t bytes;
  int node;
} g_blocks[MOCK_MAX_BLOCKS];
static int g_nblocks;
static int g_numa_local, g_numa_remote;

static int
mock_nodes (void)
{
  const char *e = getenv ("MOCK_MIBAYER_NUMA_NODES");

  return e ? atoi (e) : 0;
}

int
mibayer_device_numa_node (int device)
{
  const int k = mock_nodes ();

  return (k > 0 && device >= 0) ? device % k : -1;
}

int
mibayer_host_numa_node (const void *p)
{
  int i, node = -1;

  pthread_mutex_lock (&g_lock);
  for (i = 0; i < g_nblocks; i++)
    if ((const uint8_t *) p >= g_blocks[i].base && (const uint8_t *) p < g_blocks[i].base + g_blocks[i].bytes)
      node = g_blocks[i].node;
  pthread_mutex_unlock (&g_lock);
  return node;
}

/* element-level tests read the placement counters from the process' last words */
static void __attribute__ ((destructor))
mock_numa_report (void)
{
  if (getenv ("MOCK_MIBAYER_NUMA_REPORT"))
    fprintf (stdout, "numa_local=%d numa_remote=%d\n", g_numa_local, g_numa_remote);
}

/* frames converted on the node of their destination buffer / on another one (the driver prints both) */
void
mock_numa_counts (int *local, int *remote)
{
  *local = g_numa_local;
  *remote = g_numa_remote;
}

int
mibayer_device_count (void)
{
  const char *e = getenv ("MOCK_MIBAYER_DEVICES");

  return e ? atoi (e) : 1;
}

const char *
mibayer_strerror (int status)
{
  return status == MIBAYER_OK ? "ok" : (status == MIBAYER_ERR_HIP ? "mock device error" : "mock error");
}

const char *
mibayer_last_hip_error (void)
{
  return "";
}

void *
mibayer_host_alloc (size_t bytes)
{
  return malloc (bytes ? bytes : 1);    /* plain heap: the sanitizer sees its bounds */
}

void *
mibayer_host_alloc_near (int device, size_t bytes)
{
  void *p = mibayer_host_alloc (bytes);
  const int node = mibayer_device_numa_node (device);

  if (p && node >= 0) {
    pthread_mutex_lock (&g_lock);
    if (g_nblocks < MOCK_MAX_BLOCKS) {
      g_blocks[g_nblocks].base = p;
      g_blocks[g_nblocks].bytes = bytes ? bytes : 1;
      g_blocks[g_nblocks].node = node;
      g_nblocks++;
    }
    pthread_mutex_unlock (&g_lock);
  }
  return p;
}

/* MOCK_MIBAYER_PAGEABLE=1 makes every host buffer count as pageable */
int
mibayer_host_is_pinned (const void *p)
{
  const char *e = getenv ("MOCK_MIBAYER_PAGEABLE");

  return p != NULL && !(e && atoi (e) != 0);
}

void
mibayer_host_free (void *p)
{
  int i;

  pthread_mutex_lock (&g_lock);
  for (i = 0; i < g_nblocks; i++)
    if (g_blocks[i].base == (const uint8_t *) p) {
      g_blocks[i] = g_blocks[--g_nblocks];
      break;
    }
  pthread_mutex_unlock (&g_lock);
  free (p);
}

/* what a finished conversion leaves behind: reads every source byte, writes every destination byte the real
 * kernel would write */
static void
mock_convert (const mibayer_ctx * c, const mock_frame * fr)
{
  const mibayer_cfg *f = &c->cfg;
  const int inverse = (f->flags & MIBAYER_FLAG_RGB2BAYER) != 0;
  /* bytes per row the real path reads / writes */
  const int src_row = inverse ? 4 * f->width : ((f->width + 3) & ~3);
  const int dst_row = inverse ? ((f->width + 3) & ~3) : 4 * f->width;
  unsigned sum = 0;
  int y;

  for (y = 0; y < f->height; y++) {
    const uint8_t *s = fr->src + (size_t) y * f->src_stride;
    int x;

    for (x = 0; x < src_row; x++)
      sum += s[x];              /* the source must still be mapped and alive */
  }
  for (y = 0; y < f->height; y++)
    memset (fr->dst + (size_t) y * f->dst_stride, fr->src[0], (size_t) dst_row);
  memcpy (fr->dst, &fr->seq, 4);
  if (sum == 0xffffffffu)       /* keep the reads */
    fr->dst[4] ^= 1;
  if (mock_nodes () > 0) {
    const int bn = mibayer_host_numa_node (inverse ? (const void *) fr->src : (const void *) fr->dst);

    pthread_mutex_lock (&g_lock);
    if (bn >= 0 && bn == mibayer_device_numa_node (f->device))
      g_numa_local++;
    else
      g_numa_remote++;
    pthread_mutex_unlock (&g_lock);
  }
}

/* 1 when this "device" has stopped answering (from now on): the caller's wait runs into its deadline */
static int
mock_device_hangs (mibayer_ctx * c)
{
  if (c->hung)
    return 1;
  if (c->hang_after >= 0 && c->completed >= c->hang_after) {
    c->hung = 1;
    if (c->timeout_ms <= 0) {
      fprintf (stderr, "mock_mibayer: wait without a deadline on a device that never answers\n");
      abort ();
    }
    {
      struct timespec nap = { 0, (c->timeout_ms > 20 ? 20 : c->timeout_ms) * 1000000L };

      nanosleep (&nap, NULL);   /* the deadline passes */
    }
    return 1;
  }
  return 0;
}

/* 1 when this "device" fails now (and from now on) */
static int
mock_device_fails (mibayer_ctx * c)
{
  if (c->dead)
    return 1;
  if (c->fail_after >= 0 && c->completed >= c->fail_after) {
    c->dead = 1;
    return 1;
  }
  return 0;
}

int
mibayer_submit (mibayer_ctx * c, const uint8_t * src, uint8_t * dst, void *tag)
{
  mock_frame *fr;

  if (!c || !src || !dst)
    return MIBAYER_ERR_ARG;
  if (c->hung)
    return MIBAYER_ERR_TIMEOUT;
  if (c->dead)
    return MIBAYER_ERR_HIP;
  if (c->count == c->cfg.inflight)
    return MIBAYER_ERR_BUSY;
  fr = &c->ring[(c->head + c->count) % MOCK_MAX_PENDING];
  fr->src = src;
  fr->dst = dst;
  fr->tag = tag;
  pthread_mutex_lock (&g_lock);
  fr->seq = g_seq++;
  pthread_mutex_unlock (&g_lock);
  c->count++;
  return MIBAYER_OK;
}

int
mibayer_wait (mibayer_ctx * c, void **tag)
{
  mock_frame fr;

  if (!c)
    return MIBAYER_ERR_ARG;
  if (c->count == 0)
    return MIBAYER_ERR_EMPTY;
  if (mock_device_hangs (c))
    return MIBAYER_ERR_TIMEOUT; /* nothing comes back: the frame stays where it is */
  if (mock_device_fails (c))
    return MIBAYER_ERR_HIP;     /* nothing was converted: the frame stays where it is */
  fr = c->ring[c->head];
  c->head = (c->head + 1) % MOCK_MAX_PENDING;
  c->count--;
  mock_convert (c, &fr);
  c->completed++;
  if (tag)
    *tag = fr.tag;
  return MIBAYER_OK;
}

int
mibayer_pending (const mibayer_ctx * c)
{
  return c ? c->count : MIBAYER_ERR_ARG;
}

int
mibayer_get_cfg (const mibayer_ctx * c, mibayer_cfg * out)
{
  if (!c || !out)
    return MIBAYER_ERR_ARG;
  *out = c->cfg;
  return MIBAYER_OK;
}

/* ---- the seam the real pool uses (csrc/mibayer_hooks.h) ---- */

int
mibayer_internal_run_spare (mibayer_ctx * c, const uint8_t * src, uint8_t * dst)
{
  mock_frame fr;

  if (!c || !src || !dst)
    return MIBAYER_ERR_ARG;
  if (mock_device_hangs (c))
    return MIBAYER_ERR_TIMEOUT;
  if (mock_device_fails (c))
    return MIBAYER_ERR_HIP;
  fr.src = src;
  fr.dst = dst;
  fr.tag = NULL;
  pthread_mutex_lock (&g_lock);
  fr.seq = g_seq++;
  pthread_mutex_unlock (&g_lock);
  mock_convert (c, &fr);
  c->completed++;
  return MIBAYER_OK;
}

/* MOCK_MIBAYER_PAGEABLE=1: every host buffer counts as pageable, i.e. every frame takes the helper-thread path */
int
mibayer_internal_is_pageable (const void *p)
{
  const char *e = getenv ("MOCK_MIBAYER_PAGEABLE");

  return e && atoi (e) != 0;
}

void
mibayer_internal_private_queues (mibayer_ctx * c)
{
  (void) c;                     /* the double has no queues */
}

/* what a device that only stalled does when it comes back: the copies it had queued run after all */
static void
mock_late_writes (mibayer_ctx * c)
{
  while (c->count > 0) {
    mock_frame fr = c->ring[c->head];

    c->head = (c->head + 1) % MOCK_MAX_PENDING;
    c->count--;
    mock_convert (c, &fr);
  }
}

int
mibayer_internal_abandon (mibayer_ctx * c)
{
  if (!c)
    return MIBAYER_OK;
  c->abandoned = 1;
  if (c->hung)
    return c->count > 0 ? MIBAYER_ERR_TIMEOUT : MIBAYER_OK;    /* it keeps what it holds */
  c->count = 0;                 /* whatever a dead device held is never written */
  return MIBAYER_OK;
}

int
mibayer_internal_settled (mibayer_ctx * c)
{
  const char *e = getenv ("MOCK_MIBAYER_RESUME_POLLS");
  const int after = e ? atoi (e) : 3;

  if (!c || !c->hung || c->count == 0)
    return 1;
  if (after < 0 || ++c->settle_polls < after)
    return 0;
  mock_late_writes (c);         /* the device resumes: into buffers the caller must still hold */
  return 1;
}

int
mibayer_set_wait_spin (mibayer_ctx * c, int spin_us)
{
  (void) spin_us;
  return c ? MIBAYER_OK : MIBAYER_ERR_ARG;
}

int
mibayer_get_host_stats (const mibayer_ctx * c, mibayer_host_stats * out)
{
  if (!c || !out)
    return MIBAYER_ERR_ARG;
  memset (out, 0, sizeof *out);
  out->submits = (uint64_t) c->completed;
  return MIBAYER_OK;
}

int
mibayer_set_wait_timeout (mibayer_ctx * c, int ms)
{
  if (!c)
    return MIBAYER_ERR_ARG;
  c->timeout_ms = ms < 0 ? 10000 : ms;
  return MIBAYER_OK;
}

int
mibayer_internal_stall (mibayer_ctx * c, int ms)
{
  if (!c)
    return MIBAYER_ERR_ARG;
  c->hang_after = c->completed; /* from the next frame on */
  return MIBAYER_OK;
}

/* ---- the entry points plugin `mihip` uses (device memory, events, device-resident launches) ------------------
 *
 * "Device memory" is plain heap.  A launch does nothing when it is queued: it completes only when something
 * that the real runtime would order after it is waited for on the host -- an event recorded after it
 * (mibayer_dev_event_wait), or mibayer_sync / mibayer_destroy of its context.  Copies (dev_upload / dev_download)
 * deliberately do NOT complete pending launches: an element that reads a buffer without honouring its "last
 * access" event gets the bytes from before the launch, and the stamp check of the test fails.  Freeing memory
 * that a pending launch still uses aborts. */

#define MOCK_MAX_OPS 4096

typedef struct
{
  mibayer_ctx *ctx;             /* NULL: an asynchronous upload of `bytes` bytes, not a launch */
  const uint8_t *src;
  uint8_t *dst;
  size_t bytes;
  uint32_t seq;
  int done;
} mock_op;

static mock_op g_ops[MOCK_MAX_OPS];
static const void *g_queues_seen[16];
static int g_nqueues_seen;
static uint32_t g_nops;         /* launches and asynchronous copies queued so far, in device order */
static uint32_t g_nlaunches;    /* launches among them == stamp of the next launch */
static uint32_t g_event_records;        /* mibayer_dev_event_record calls (MOCK_MIBAYER_LOG_RECORDS): what the elements'
                                           bookkeeping costs the runtime per frame */
static uint32_t g_stream_waits;         /* mibayer_dev_stream_wait_event calls */

typedef struct
{
  uint32_t marker;              /* launches / async copies queued before the record */
  int queried;                  /* mibayer_dev_event_query says "not yet" once per record */
} mock_event;

static void
mock_complete_upto (uint32_t marker, const mibayer_ctx * only)
{
  uint32_t i;

  for (i = 0; i < g_nops && i < marker; i++) {
    mock_op *op = &g_ops[i % MOCK_MAX_OPS];
    unsigned sum = 0;
    size_t k;

    (void) only;                /* one in-order device: earlier work of other contexts completes too */
    if (op->done)
      continue;
    if (op->ctx == NULL) {
      /* asynchronous upload: the DMA reads the host buffer NOW -- if the element has let go of it, the
       * sanitizer reports the use after free */
      memcpy (op->dst, op->src, op->bytes);
      op->done = 1;
      continue;
    }
    for (k = 0; k < op->ctx->src_bytes; k++)
      sum += op->src[k];        /* the source must still be alive */
    memset (op->dst, op->src[0], op->ctx->dst_bytes);
    memcpy (op->dst, &op->seq, 4);
    if (sum == 0xffffffffu)
      op->dst[4] ^= 1;
    op->done = 1;
  }
}

int
mibayer_create (const mibayer_cfg * cfg, mibayer_ctx ** out)
{
  mibayer_ctx *c;
  int inverse;

  if (!cfg || !out || cfg->struct_size != sizeof (*cfg))
    return MIBAYER_ERR_ARG;
  if (mibayer_device_count () <= 0)
    return MIBAYER_ERR_NO_DEVICE;
  if (cfg->device >= mibayer_device_count ())
    return MIBAYER_ERR_NO_DEVICE;
  if (getenv ("MOCK_MIBAYER_LOG_DEVICES"))
    fprintf (stderr, "mock_mibayer: context on device %d\n", cfg->device);
  inverse = (cfg->flags & MIBAYER_FLAG_RGB2BAYER) != 0;
  /* the real library's geometry domain for bayer2rgb */
  if (!inverse && (cfg->width < 4 || (cfg->width & 1) || cfg->height < 3))
    return MIBAYER_ERR_GEOMETRY;
  c = calloc (1, sizeof *c);
  c->cfg = *cfg;
  if (c->cfg.src_stride == 0)
    c->cfg.src_stride = inverse ? 4 * cfg->width : ((cfg->width + 3) & ~3);
  if (c->cfg.dst_stride == 0)
    c->cfg.dst_stride = inverse ? ((cfg->width + 3) & ~3) : 4 * cfg->width;
  if (c->cfg.inflight <= 0)
    c->cfg.inflight = 2;
  if (c->cfg.inflight > MOCK_MAX_PENDING)
    c->cfg.inflight = MOCK_MAX_PENDING;
  c->src_bytes = (size_t) c->cfg.src_stride * cfg->height;
  c->dst_bytes = (size_t) c->cfg.dst_stride * cfg->height;
  c->fail_after = -1;
  c->hang_after = -1;
  c->timeout_ms = 10000;
  c->plan_source = g_plan_cached ? MIBAYER_PLAN_CACHED : MIBAYER_PLAN_DEFAULT;
  pthread_mutex_lock (&g_lock);
  if (g_live_host_ctx++ == 0)
    g_seq = 0;                  /* a new pool stamps its frames from 0 */
  c->index = g_created++;
  pthread_mutex_unlock (&g_lock);
  {
    /* MOCK_MIBAYER_FAIL="index:frames[,index:frames]": the index-th context created in this process turns into
     * a failed device once it has completed that many frames */
    const char *e = getenv ("MOCK_MIBAYER_FAIL");

    while (e && *e) {
      char *end = NULL;
      long idx = strtol (e, &end, 10), n;

      if (end == e || *end != ':')
        break;
      e = end + 1;
      n = strtol (e, &end, 10);
      if (end == e)
        break;
      if (idx == c->index)
        c->fail_after = (int) n;
      e = (*end == ',') ? end + 1 : end;
    }
    e = getenv ("MOCK_MIBAYER_HANG");
    while (e && *e) {
      char *end = NULL;
      long idx = strtol (e, &end, 10), n;

      if (end == e || *end != ':')
        break;
      e = end + 1;
      n = strtol (e, &end, 10);
      if (end == e)
        break;
      if (idx == c->index)
        c->hang_after = (int) n;
      e = (*end == ',') ? end + 1 : end;
    }
  }
  *out = c;
  return MIBAYER_OK;
}

int
mibayer_sync (mibayer_ctx * c)
{
  if (!c)
    return MIBAYER_ERR_ARG;
  pthread_mutex_lock (&g_lock);
  mock_complete_upto (g_nops, c);
  pthread_mutex_unlock (&g_lock);
  return MIBAYER_OK;
}

void
mibayer_destroy (mibayer_ctx * c)
{
  if (!c)
    return;
  if (getenv ("MOCK_MIBAYER_LOG_QUEUES"))
    fprintf (stderr, "mock_mibayer: device-resident launches went to %d distinct queue(s)\n", g_nqueues_seen);
  if (getenv ("MOCK_MIBAYER_LOG_RECORDS"))
    fprintf (stderr, "mock_mibayer: %u launch(es), %u event record(s), %u stream wait(s) so far\n", g_nlaunches,
        g_event_records, g_stream_waits);
  if (c->hung && !c->abandoned && c->count > 0) {
    fprintf (stderr, "mock_mibayer: destroy would block for ever on a device that never answers\n");
    abort ();
  }
  if (c->hung)
    mock_late_writes (c);       /* a stalled device comes back when it likes: at the latest now */
  /* a failed device whose context was never abandoned still carries out what it had queued: late writes */
  while (c->dead && !c->abandoned && c->count > 0) {
    mock_frame fr = c->ring[c->head];

    c->head = (c->head + 1) % MOCK_MAX_PENDING;
    c->count--;
    mock_convert (c, &fr);
  }
  mibayer_sync (c);
  pthread_mutex_lock (&g_lock);
  g_live_host_ctx--;
  pthread_mutex_unlock (&g_lock);
  free (c);
}

/* ---- launch plans: the double has one, "measured" once somebody asks for it ---- */

int
mibayer_plan_source (const mibayer_ctx * c)
{
  return c ? c->plan_source : MIBAYER_ERR_ARG;
}

int
mibayer_plan_from_cache (mibayer_ctx * c)
{
  if (!c)
    return MIBAYER_ERR_ARG;
  if (!g_plan_cached)
    return 0;
  c->plan_source = MIBAYER_PLAN_CACHED;
  return 1;
}

int
mibayer_get_plan (const mibayer_ctx * c, int *variant, int *band, int *align_stores)
{
  if (!c)
    return MIBAYER_ERR_ARG;
  if (variant)
    *variant = 1;
  if (band)
    *band = c->plan_source == MIBAYER_PLAN_DEFAULT ? 1 : 0;
  if (align_stores)
    *align_stores = 0;
  return MIBAYER_OK;
}

const char *
mibayer_ctx_variant_name (const mibayer_ctx * c)
{
  return c ? "mock_plan" : NULL;
}

/* ABI v5: one plan per launch class -- the double keeps one plan for all of them */
int
mibayer_get_plan_for (const mibayer_ctx * c, int nframes, int *variant, int *band, int *align_stores, int *source)
{
  if (!c || nframes < 1)
    return MIBAYER_ERR_ARG;
  if (source)
    *source = c->plan_source;
  return mibayer_get_plan (c, variant, band, align_stores);
}

const char *
mibayer_variant_name (int variant)
{
  return variant == 1 ? "mock_plan" : NULL;
}

int
mibayer_autotune_list (mibayer_ctx * c, const void *const *d_srcs, void *const *d_dsts, int nframes, char *report,
    size_t report_len)
{
  if (!c || !d_srcs || !d_dsts || nframes < 1)
    return MIBAYER_ERR_ARG;
  pthread_mutex_lock (&g_lock);
  g_plan_cached = 1;
  g_autotunes++;
  pthread_mutex_unlock (&g_lock);
  c->plan_source = MIBAYER_PLAN_MEASURED;
  if (report && report_len)
    snprintf (report, report_len, "mock autotune over %d frame(s)", nframes);
  if (getenv ("MOCK_MIBAYER_LOG_AUTOTUNE"))
    fprintf (stderr, "mock_mibayer: autotune #%d over %d frame(s)\n", g_autotunes, nframes);
  return MIBAYER_OK;
}

/* Like the real library's default (DeviceQueues): ONE compute queue per device, shared by every context on it -- the
 * stages of a device-resident pipeline are ordered by that queue, which is what makes their hand-overs free */
static char g_queue_tokens[1 + MIBAYER_FRAME_QUEUES];

void *
mibayer_ctx_stream (mibayer_ctx * c)
{
  return c ? &g_queue_tokens[0] : NULL;
}

/* the frame queues: more tokens.  The double executes every queued operation in ONE global order when something
 * ordered after it completes -- a legal schedule for any number of queues as long as the element orders each launch
 * after the last access of its buffers, which is what the tests check */
void *
mibayer_ctx_frame_queue (mibayer_ctx * c, int k)
{
  return (c && k >= 0 && k < MIBAYER_FRAME_QUEUES) ? &g_queue_tokens[1 + k] : NULL;
}

int
mibayer_process_device (mibayer_ctx * c, const void *d_src,
    size_t src_frame_bytes, void *d_dst, size_t dst_frame_bytes, int nframes,
    void *hip_stream)
{
  mock_op *op;

  if (!c || !d_src || !d_dst || nframes != 1)
    return MIBAYER_ERR_ARG;
  pthread_mutex_lock (&g_lock);
  {
    int k;                      /* which queues the element launches on (MOCK_MIBAYER_LOG_QUEUES) */

    for (k = 0; k < g_nqueues_seen && g_queues_seen[k] != hip_stream; k++);
    if (k == g_nqueues_seen && g_nqueues_seen < 16)
      g_queues_seen[g_nqueues_seen++] = hip_stream;
  }
  op = &g_ops[g_nops % MOCK_MAX_OPS];
  op->ctx = c;
  op->src = d_src;
  op->dst = d_dst;
  op->seq = g_nlaunches++;
  op->done = 0;
  g_nops++;
  pthread_mutex_unlock (&g_lock);
  return MIBAYER_OK;
}

int
mibayer_process_device_list (mibayer_ctx * c, const void *const *d_srcs, void *const *d_dsts, int nframes,
    void *hip_stream)
{
  int f, rc = MIBAYER_OK;

  if (!c || !d_srcs || !d_dsts || nframes < 0)
    return MIBAYER_ERR_ARG;
  for (f = 0; f < nframes && rc == MIBAYER_OK; f++)
    rc = mibayer_process_device (c, d_srcs[f], 0, d_dsts[f], 0, 1, hip_stream);
  return rc;
}

/* the synthetic-frame generator (hipbayersrc): queued like a launch would be too much for what the double checks --
 * the frame is written at once, its first four bytes carry the frame number */
int
mibayer_fill_synthetic (mibayer_ctx * c, void *d_src, size_t src_frame_bytes, uint32_t first_frame, int nframes,
    uint32_t seed, void *hip_stream)
{
  int f;

  if (!c || !d_src || nframes < 0)
    return MIBAYER_ERR_ARG;
  for (f = 0; f < nframes; f++) {
    uint8_t *p = (uint8_t *) d_src + (size_t) f * (src_frame_bytes ? src_frame_bytes : c->src_bytes);
    const uint32_t stamp = first_frame + (uint32_t) f;

    memset (p, (int) ((seed + stamp) & 0xff), c->src_bytes);
    memcpy (p, &stamp, 4);
  }
  return MIBAYER_OK;
}

void *
mibayer_dev_alloc (int device, size_t bytes)
{
  return (device >= 0 && device < mibayer_device_count ()) ? malloc (bytes ? bytes : 1) : NULL;
}

void
mibayer_dev_free (int device, void *d_ptr)
{
  uint32_t i;

  pthread_mutex_lock (&g_lock);
  for (i = 0; i < g_nops && i < MOCK_MAX_OPS; i++) {
    const mock_op *op = &g_ops[i];

    if (!op->done && (op->src == d_ptr || op->dst == d_ptr)) {
      fprintf (stderr, "mock_mibayer: device memory freed while launch %u still uses it\n", op->seq);
      abort ();
    }
  }
  pthread_mutex_unlock (&g_lock);
  free (d_ptr);
}

int
mibayer_dev_upload (int device, void *d_dst, const void *src, size_t bytes)
{
  memcpy (d_dst, src, bytes);
  return MIBAYER_OK;
}

void *
mibayer_dev_stream_create (int device)
{
  return device == 0 ? malloc (1) : NULL;
}

void
mibayer_dev_stream_destroy (int device, void *hip_stream)
{
  pthread_mutex_lock (&g_lock);
  mock_complete_upto (g_nops, NULL);    /* destroying a queue synchronises it */
  pthread_mutex_unlock (&g_lock);
  free (hip_stream);
}

/* queued, not executed: the copy happens when something ordered after it completes */
int
mibayer_dev_upload_async (int device, void *d_dst, const void *src, size_t bytes, void *hip_stream)
{
  mock_op *op;

  if (!d_dst || !src || !hip_stream)
    return MIBAYER_ERR_ARG;
  pthread_mutex_lock (&g_lock);
  op = &g_ops[g_nops % MOCK_MAX_OPS];
  op->ctx = NULL;
  op->src = src;
  op->dst = d_dst;
  op->bytes = bytes;
  op->seq = g_nops;
  op->done = 0;
  g_nops++;
  pthread_mutex_unlock (&g_lock);
  return MIBAYER_OK;
}

/* the other direction: the host buffer is WRITTEN when something ordered after the copy completes -- an output
 * buffer pushed downstream (or unmapped) before that is a sanitizer report or a wrong stamp */
int
mibayer_dev_download_async (int device, void *dst, const void *d_src, size_t bytes, void *hip_stream)
{
  return mibayer_dev_upload_async (device, dst, d_src, bytes, hip_stream);
}

int
mibayer_dev_event_query (int device, void *event)
{
  mock_event *ev = event;

  if (!ev)
    return MIBAYER_ERR_ARG;
  if (!ev->queried) {
    ev->queried = 1;
    return 0;                   /* "still copying" */
  }
  pthread_mutex_lock (&g_lock);
  mock_complete_upto (ev->marker, NULL);
  pthread_mutex_unlock (&g_lock);
  return 1;
}

int
mibayer_dev_download (int device, void *dst, const void *d_src, size_t bytes)
{
  memcpy (dst, d_src, bytes);
  return MIBAYER_OK;
}

void *
mibayer_dev_event_create (int device)
{
  return calloc (1, sizeof (mock_event));
}

void
mibayer_dev_event_destroy (int device, void *event)
{
  free (event);
}

int
mibayer_dev_event_record (int device, void *event, void *hip_stream)
{
  if (!event)
    return MIBAYER_ERR_ARG;
  pthread_mutex_lock (&g_lock);
  g_event_records++;
  ((mock_event *) event)->marker = g_nops;
  ((mock_event *) event)->queried = 0;
  pthread_mutex_unlock (&g_lock);
  return MIBAYER_OK;
}

int
mibayer_dev_event_wait (int device, void *event)
{
  if (!event)
    return MIBAYER_ERR_ARG;
  pthread_mutex_lock (&g_lock);
  mock_complete_upto (((mock_event *) event)->marker, NULL);
  pthread_mutex_unlock (&g_lock);
  return MIBAYER_OK;
}

int
mibayer_dev_stream_wait_event (int device, void *hip_stream, void *event)
{
  pthread_mutex_lock (&g_lock);
  g_stream_waits++;
  pthread_mutex_unlock (&g_lock);
  return event ? MIBAYER_OK : MIBAYER_ERR_ARG;  /* one in-order list of launches: nothing to do */
}
