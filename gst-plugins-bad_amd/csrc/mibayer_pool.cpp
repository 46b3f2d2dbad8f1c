/*
 * mibayer_pool -- round-robin frame sharding over GPUs (include/mibayer.h,
 * "multi-GPU frame sharding").  Pure host logic: no HIP call in this file, the
 * devices are reached through the per-context ABI and csrc/mibayer_hooks.h
 * only, so the same code runs over a test double of the contexts under
 * AddressSanitizer / ThreadSanitizer (tests/test_pool_logic.py).
 *
 * Frames are independent (the reference keeps no state between frames,
 * gstbayer2rgb.c:387-451): frame g goes to the next device in rotation and the
 * results come back in submission order.  On top of that, three things a
 * production multi-GPU stream needs:
 *
 *  - FAILOVER.  The reference's only failure handling on this path is "warn and
 *    carry on" (gstbayer2rgb.c:484-486).  Here a device whose context reports
 *    MIBAYER_ERR_HIP / MIBAYER_ERR_NOMEM is dropped from the rotation, the
 *    frames it still held are converted again on the surviving devices (same
 *    bytes: the conversion is a pure function of the frame), order is kept, and
 *    the event is reported once (mibayer_pool_take_failure).  The stream fails
 *    only when no device is left.
 *
 *  - PAGEABLE BUFFERS.  hipMemcpyAsync from / to pageable memory blocks the
 *    calling thread while the runtime stages it, so one streaming thread would
 *    feed N GPUs one after the other.  In a pool of more than one shard, the
 *    first frame whose source or destination is pageable turns its shard over to
 *    a helper thread (one per shard, started on first use): from then on the
 *    helper drives that context's submit / wait ring -- so uploads, kernels and
 *    downloads of consecutive frames keep overlapping inside the shard -- and
 *    the streaming thread only queues and, at wait time, blocks on the oldest
 *    frame.  Shards that only ever see pinned frames keep the direct,
 *    enqueue-only path, and so does a single-shard pool (a helper would add a
 *    hand-off and no parallelism).  (A second CPU copy through a pinned bounce
 *    buffer was rejected: the runtime's own staging moves ~40 GB/s, a memcpy
 *    thread ~10.)
 *
 *  - DEADLINES.  A device that stops answering completes nothing: every wait of a context is bounded
 *    (mibayer_set_wait_timeout), MIBAYER_ERR_TIMEOUT counts as a device failure like MIBAYER_ERR_HIP, and a
 *    context that timed out is never waited for again (mibayer_internal_abandon returns at once for it).
 *
 *  - NUMA-LOCAL ROUTING.  On a two-socket node a frame whose pinned buffer sits next to GPU k should go to GPU k:
 *    when the shards' devices span more than one NUMA node, a frame is routed to the live shard on the node that
 *    holds its 4-byte-per-pixel buffer as long as that keeps the rotation balanced (no shard more than its in-flight
 *    share ahead); otherwise, and whenever the node is unknown, strictly g mod N.  Results keep submission order.
 *
 *  - A SUBMIT THREAD PER SHARD (MIBAYER_POOL_THREADS=1): every shard is driven by its helper thread from the start,
 *    pinned frames included, so N GPUs are fed by N host threads instead of the one streaming thread.
 *
 *  - FAULT INJECTION for drills and tests: mibayer_pool_inject_fault() /
 *    MIBAYER_INJECT_FAULT=shard:frames make a shard report a device error after
 *    it has completed that many frames; mibayer_pool_inject_stall() /
 *    MIBAYER_INJECT_STALL=shard:frames:ms occupy a shard's compute queue for `ms`
 *    milliseconds once `frames` frames have been routed to it.
 */
#include "mibayer_hooks.h"

#include <sched.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <new>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

enum FrameState {
  F_DIRECT,     /* in the ring of its shard's context, submitted by the streaming thread */
  F_QUEUED,     /* in the job queue of its shard's helper thread                  */
  F_RUNNING,    /* the helper thread has it (being submitted, or in the context's ring) */
  F_DONE,       /* converted by a helper thread (or by a synchronous re-do)       */
  F_FAILED,     /* the helper thread got a device error for it                    */
  F_REDO,       /* its device failed: convert again on a surviving device         */
  F_LOST        /* it was in flight on a device that ran into the wait deadline: the
                   device may still read / write its buffers, so it is neither converted
                   again into them nor handed back as done (mibayer_pool_reclaim)      */
};

struct Frame {
  const uint8_t *src;
  uint8_t *dst;
  void *tag;
  int shard;    /* where it is (being) converted  */
  int owner;    /* whose in-flight budget it uses */
  int state;    /* FrameState; helper-thread frames: under Shard::mu */
  int rc;
  bool submitted;       /* handed to the context of `shard` (its ring may hold it) */
};

struct Shard {
  mibayer_ctx *ctx = nullptr;
  int device = 0;
  bool alive = true;
  int inflight = 0;             /* frames owned and not yet handed back (streaming thread) */
  int node = -1;                /* NUMA node next to the device; -1 unknown */
  long long assigned = 0;       /* frames routed here so far (balance of the NUMA-local routing) */
  long long stall_after = -1;   /* MIBAYER_INJECT_STALL: stall the compute queue once this many frames were routed here */
  int stall_ms = 0;
  std::atomic<long long> completions { 0 };
  std::atomic<long long> fail_after { -1 };     /* fault injection; -1 = never */
  /* helper thread */
  std::thread th;
  bool th_started = false;
  std::mutex mu;
  std::condition_variable cv_job, cv_done;
  std::deque<Frame *> jobs;     /* waiting for room in the context's ring             */
  std::deque<Frame *> ring;     /* in the context's ring, oldest first (helper mode)  */
  bool helper_mode = false;     /* the helper thread owns the context's submit / wait */
  bool pageable_seen = false;   /* a pageable frame came by: the context has queues of its own */
  int ring_room = 2;            /* stream.inflight                                     */
  bool quit = false;
  bool broken = false;
  bool timed_out = false;       /* what broke it was a wait deadline */
  int helper_lost = 0;          /* frames its helper thread marked F_LOST (for the failure note) */
  char errmsg[200] = "";

  /* true exactly once, when the shard has completed `fail_after` frames */
  bool fault_due ()
  {
    const long long n = ++completions;
    const long long at = fail_after.load ();
    return at >= 0 && n > at;
  }
};

bool device_failure (int rc)
{
  return rc == MIBAYER_ERR_HIP || rc == MIBAYER_ERR_NOMEM || rc == MIBAYER_ERR_TIMEOUT;
}

/* everything the helper still holds is re-done elsewhere (mu held) -- except, after a wait deadline, the frames
 * the context already has: those are lost (F_LOST).  kill_shard() has the last word on which of the two it is. */
void helper_give_up (Shard *sh, const char *why, bool timed_out = false)
{
  sh->broken = true;
  sh->timed_out = sh->timed_out || timed_out;
  if (why != sh->errmsg)
    snprintf (sh->errmsg, sizeof sh->errmsg, "%s", why ? why : "");
  for (Frame *f : sh->ring) {
    f->state = (sh->timed_out && f->submitted) ? F_LOST : F_REDO;
    sh->helper_lost += f->state == F_LOST ? 1 : 0;
  }
  sh->ring.clear ();
  for (Frame *f : sh->jobs)
    f->state = F_REDO;
  sh->jobs.clear ();
  sh->cv_done.notify_all ();
}

/* Helper mode: this thread is the only caller of the context's submit / wait.
 * It keeps the ring as full as the queue allows (a submit from pageable memory
 * blocks while the runtime stages the upload -- that is the point of being on a
 * thread of its own) and retires the oldest frame whenever nothing can be
 * submitted. */
/* A shard's thread runs on the CPUs of the NUMA node next to its GPU (hard part 7: "per-GPU
 * threads, NUMA-local pinned buffers"): its staging copies and the runtime's submission path then touch memory of
 * that socket only.  Best effort -- the node's CPU list comes from sysfs; nothing happens when the node is unknown,
 * the list cannot be read, or none of its CPUs is in the thread's current mask.  MIBAYER_POOL_PIN_THREADS=0: off. */
void pin_to_node (int node)
{
  if (node < 0)
    return;
  if (const char *e = getenv ("MIBAYER_POOL_PIN_THREADS"))
    if (atoi (e) == 0)
      return;
  char path[96];
  snprintf (path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
  FILE *f = fopen (path, "r");
  if (!f)
    return;
  char list[4096];
  const size_t n = fread (list, 1, sizeof list - 1, f);
  fclose (f);
  list[n] = 0;
  cpu_set_t allowed, want;
  CPU_ZERO (&want);
  if (sched_getaffinity (0, sizeof allowed, &allowed) != 0)
    return;
  int count = 0;
  for (char *q = list; *q;) {           /* "0-63,128-191" */
    char *end = NULL;
    const long a = strtol (q, &end, 10);
    if (end == q)
      break;
    long b = a;
    if (*end == '-') {
      q = end + 1;
      b = strtol (q, &end, 10);
      if (end == q)
        break;
    }
    for (long c = a; c <= b && c < CPU_SETSIZE; c++)
      if (c >= 0 && CPU_ISSET ((int) c, &allowed)) {
        CPU_SET ((int) c, &want);
        count++;
      }
    q = (*end == ',') ? end + 1 : end;
    if (*end != ',' )
      break;
  }
  if (count > 0)
    (void) sched_setaffinity (0, sizeof want, &want);
}

void helper_main (Shard *sh)
{
  pin_to_node (sh->node);
  std::unique_lock<std::mutex> lk (sh->mu);
  for (;;) {
    sh->cv_job.wait (lk, [sh] {
      return sh->quit || !sh->jobs.empty () || !sh->ring.empty ();
    });
    if (sh->quit && sh->jobs.empty () && sh->ring.empty ())
      return;
    if (sh->broken) {
      helper_give_up (sh, sh->errmsg);
      continue;
    }
    if (!sh->jobs.empty () && (int) sh->ring.size () < sh->ring_room) {
      Frame *f = sh->jobs.front ();
      sh->jobs.pop_front ();
      f->state = F_RUNNING;
      f->submitted = true;      /* from here on the context may hold it */
      sh->ring.push_back (f);
      lk.unlock ();
      const int rc = mibayer_submit (sh->ctx, f->src, f->dst, f);
      lk.lock ();
      if (rc != MIBAYER_OK) {
        f->rc = rc;
        f->submitted = false;   /* a failed submit leaves nothing of the frame behind (mibayer_submit drains) */
        if (!sh->ring.empty () && sh->ring.back () == f)
          sh->ring.pop_back ();
        if (device_failure (rc)) {
          f->state = F_REDO;
          helper_give_up (sh, mibayer_last_hip_error (), rc == MIBAYER_ERR_TIMEOUT);
        } else {
          f->state = F_FAILED;
          sh->cv_done.notify_all ();
        }
      }
      continue;
    }
    if (!sh->ring.empty ()) {
      Frame *f = sh->ring.front ();
      lk.unlock ();
      int rc = mibayer_wait (sh->ctx, NULL);
      if (rc == MIBAYER_OK && sh->fault_due ())
        rc = MIBAYER_ERR_HIP;
      lk.lock ();
      if (sh->ring.empty () || sh->ring.front () != f)
        continue;               /* the shard was given up meanwhile */
      if (rc == MIBAYER_OK) {
        sh->ring.pop_front ();
        f->state = F_DONE;
        sh->cv_done.notify_all ();
      } else if (device_failure (rc)) {
        f->rc = rc;
        /* f and everything behind it: F_REDO -- after a deadline, what the context holds: F_LOST */
        helper_give_up (sh, mibayer_last_hip_error (), rc == MIBAYER_ERR_TIMEOUT);
      } else {
        sh->ring.pop_front ();
        f->rc = rc;
        f->state = F_FAILED;
        sh->cv_done.notify_all ();
      }
    }
  }
}

}  /* namespace */

struct mibayer_pool {
  std::vector<Shard *> shards;
  std::deque<Frame> fifo;       /* submission order; element addresses are stable */
  int per_shard = 2;            /* stream.inflight */
  size_t rr = 0;                /* shard whose turn it is */
  size_t redo_rr = 0;
  bool use_helpers = true;      /* MIBAYER_POOL_HELPERS=0: pageable frames on the calling thread */
  bool numa_route = false;      /* the shards' devices span more than one NUMA node */
  bool inverse = false;         /* MIBAYER_FLAG_RGB2BAYER: the source is the 4 B/px side */
  std::unordered_map<const void *, int> node_of;        /* NUMA node of buffers seen so far (pools recycle them) */
  /* frames lost on a device that ran into the wait deadline: their buffers are the device's until it settles */
  struct Lost { void *tag; int shard; };
  std::vector<Lost> lost;
  /* failure report */
  int unreported = 0;
  int failed_device = -1;
  char failure_msg[320] = "";
};

static int alive_count (const mibayer_pool *pool)
{
  int n = 0;
  for (const Shard *s : pool->shards)
    n += s->alive ? 1 : 0;
  return n;
}

/* Drop shard `idx` from the rotation.  What it had not been given yet is converted on the others; what its context
 * already held is converted again too IF the context could be quiesced (mibayer_internal_abandon: its copies have
 * completed or its queues are dead).  After a wait deadline it cannot: the device may only be slow, its queued
 * uploads may still read those frames' sources and its downloads write their destinations later -- they are LOST
 * (handed back with MIBAYER_ERR_TIMEOUT, buffers quarantined until mibayer_pool_reclaim). */
static void kill_shard (mibayer_pool *pool, int idx, int rc, const char *why)
{
  Shard *sh = pool->shards[(size_t) idx];
  if (!sh->alive)
    return;
  sh->alive = false;
  char reason[200];
  {
    std::lock_guard<std::mutex> lk (sh->mu);
    snprintf (reason, sizeof reason, "%s", (why && why[0]) ? why : sh->errmsg);
    why = reason;
    sh->broken = true;
    if (rc == MIBAYER_ERR_TIMEOUT)
      sh->timed_out = true;
    for (Frame *f : sh->ring)
      f->state = F_REDO;
    sh->ring.clear ();
    for (Frame *f : sh->jobs)
      f->state = F_REDO;
    sh->jobs.clear ();
    sh->cv_job.notify_all ();
  }
  for (Frame &f : pool->fifo)
    if (f.shard == idx && f.state == F_DIRECT)
      f.state = F_REDO;
  /* after this nothing of that context touches the callers' buffers any more -- or the frames it held are lost */
  const bool held = mibayer_internal_abandon (sh->ctx) != MIBAYER_OK;
  int nlost = 0;
  {
    std::lock_guard<std::mutex> lk (sh->mu);
    for (Frame &f : pool->fifo)
      if (f.shard == idx && f.submitted && (f.state == F_REDO || f.state == F_LOST)) {
        nlost += (held && f.state == F_REDO) ? 1 : 0;
        f.state = held ? F_LOST : F_REDO;
      }
    nlost += held ? sh->helper_lost : 0;
    sh->cv_done.notify_all ();
  }
  pool->unreported++;
  pool->failed_device = sh->device;
  char lost_note[64] = "";
  if (nlost > 0)
    snprintf (lost_note, sizeof lost_note, "; %d frame(s) in flight on it lost", nlost);
  snprintf (pool->failure_msg, sizeof pool->failure_msg,
      "shard %d (HIP device %d) dropped from the rotation: %.60s%s%.120s%s; %d device(s) left",
      idx, sh->device, mibayer_strerror (rc), why && why[0] ? " -- " : "",
      why ? why : "", lost_note, alive_count (pool));
}

/* the helper thread of a live shard has seen a device error that the streaming
 * thread has not come across yet (the failed frame sits further back in the
 * queue): take the shard out now, or re-done frames would bounce off it forever */
static bool reap_if_broken (mibayer_pool *pool, int idx)
{
  Shard *sh = pool->shards[(size_t) idx];
  bool broken, timed_out;
  {
    std::lock_guard<std::mutex> lk (sh->mu);
    broken = sh->broken;
    timed_out = sh->timed_out;
  }
  if (broken && sh->alive)
    kill_shard (pool, idx, timed_out ? MIBAYER_ERR_TIMEOUT : MIBAYER_ERR_HIP, NULL);
  return broken;
}

static void start_helper (Shard *sh)
{
  if (!sh->th_started) {
    sh->th = std::thread (helper_main, sh);
    sh->th_started = true;
  }
}

static void queue_to_helper (Shard *sh, Frame *f)
{
  start_helper (sh);
  std::lock_guard<std::mutex> lk (sh->mu);
  f->state = F_QUEUED;
  sh->jobs.push_back (f);
  sh->cv_job.notify_one ();
}

/* The first pageable frame of a shard: from now on its helper thread is the only
 * one that touches the context's submit / wait ring.  Frames the streaming
 * thread had submitted itself and that are still in that ring are handed over,
 * in order, so the helper retires them first. */
static void enter_helper_mode (mibayer_pool *pool, int idx, bool pageable)
{
  Shard *sh = pool->shards[(size_t) idx];
  /* blocking (pageable) copies must not queue behind the other shards' (mibayer_hooks.h); pinned frames keep the
   * shared queues, where their copies go back to back */
  if (sh->helper_mode)
    return;                     /* its thread already drives the context: the queues stay as they are */
  if (pageable)
    mibayer_internal_private_queues (sh->ctx);
  start_helper (sh);
  std::lock_guard<std::mutex> lk (sh->mu);
  sh->helper_mode = true;
  for (Frame &f : pool->fifo)
    if (f.shard == idx && f.state == F_DIRECT) {
      f.state = F_RUNNING;      /* (submitted stays true: the context holds it) */
      sh->ring.push_back (&f);
    }
  sh->cv_job.notify_one ();
}

extern "C" int mibayer_pool_create (const mibayer_pool_cfg *cfg,
    mibayer_pool **out)
{
  if (!out)
    return MIBAYER_ERR_ARG;
  *out = NULL;
  if (!cfg || cfg->struct_size != sizeof (mibayer_pool_cfg))
    return MIBAYER_ERR_ARG;
  if (cfg->ndevices < 1 || cfg->ndevices > MIBAYER_MAX_SHARDS)
    return MIBAYER_ERR_ARG;
  mibayer_pool *pool = new (std::nothrow) mibayer_pool ();
  if (!pool)
    return MIBAYER_ERR_NOMEM;
  for (int i = 0; i < cfg->ndevices; i++) {
    mibayer_cfg one = cfg->stream;
    one.device = cfg->devices[i];
    if (one.device < 0) {
      mibayer_pool_destroy (pool);
      return MIBAYER_ERR_NO_DEVICE;
    }
    mibayer_ctx *c = NULL;
    int rc = mibayer_create (&one, &c);
    if (rc != MIBAYER_OK) {
      mibayer_pool_destroy (pool);
      return rc;
    }
    Shard *sh = new (std::nothrow) Shard ();
    if (!sh) {
      mibayer_destroy (c);
      mibayer_pool_destroy (pool);
      return MIBAYER_ERR_NOMEM;
    }
    sh->ctx = c;
    sh->device = one.device;
    pool->shards.push_back (sh);
  }
  {
    mibayer_cfg resolved;
    if (mibayer_get_cfg (pool->shards[0]->ctx, &resolved) == MIBAYER_OK
        && resolved.inflight > 0)
      pool->per_shard = resolved.inflight;
    for (Shard *sh : pool->shards)
      sh->ring_room = pool->per_shard;
  }
  /* More than one shard: whenever a frame is waited for, the other shards hold queued work, so no wait ever needs to
   * spin for latency's sake (mibayer_set_wait_spin; a context on its own cannot know that its neighbours are busy).
   * MIBAYER_WAIT_SPIN_US in the environment still decides. */
  if (pool->shards.size () > 1 && getenv ("MIBAYER_WAIT_SPIN_US") == NULL)
    for (Shard *sh : pool->shards)
      (void) mibayer_set_wait_spin (sh->ctx, 0);
  /* one shard: a helper would only add a hand-off */
  pool->use_helpers = pool->shards.size () > 1;
  if (const char *e = getenv ("MIBAYER_POOL_HELPERS"))
    pool->use_helpers = atoi (e) != 0;
  pool->inverse = (cfg->stream.flags & MIBAYER_FLAG_RGB2BAYER) != 0;
  {
    int first = -2;
    for (Shard *sh : pool->shards) {
      sh->node = mibayer_device_numa_node (sh->device);
      if (sh->node >= 0 && first == -2)
        first = sh->node;
      else if (sh->node >= 0 && sh->node != first)
        pool->numa_route = true;
    }
    if (const char *e = getenv ("MIBAYER_POOL_NUMA"))
      pool->numa_route = pool->numa_route && atoi (e) != 0;
  }
  /* A submit thread per shard, pinned frames included (include/mibayer.h).  Measured (profiles/r04_host_cpu.log,
   * bench.py host_path.host_cpu): queueing a 4K frame from PINNED buffers costs the calling thread 14 us of CPU at the
   * ABI and ~80 us in the element (buffer maps, pool traffic), waiting for it ~8-20 us (napping waits) -- one
   * streaming thread saturates at roughly 11 k frames/s, which is what EIGHT PCIe links carry (8 x ~1.4 k frames/s
   * at 4K).  So pools over six or more DISTINCT GPUs start with a thread per shard (each then spends ~12 % of a core,
   * pinned next to its GPU); smaller pools keep the enqueue-only streaming thread (and shards that see PAGEABLE
   * buffers, 200-300 us of staging copy per frame, get their helper thread either way).  MIBAYER_POOL_THREADS=0 / 1
   * decides otherwise. */
  {
    size_t distinct = 0;
    for (size_t i = 0; i < pool->shards.size (); i++) {
      bool seen = false;
      for (size_t k = 0; k < i; k++)
        seen = seen || pool->shards[k]->device == pool->shards[i]->device;
      distinct += seen ? 0 : 1;
    }
    bool threads = distinct >= 6;
    if (const char *e = getenv ("MIBAYER_POOL_THREADS"))
      threads = atoi (e) != 0;
    if (threads && pool->shards.size () > 1)
      for (size_t i = 0; i < pool->shards.size (); i++)
        enter_helper_mode (pool, (int) i, false);
  }
  if (const char *e = getenv ("MIBAYER_INJECT_FAULT")) {
    /* "shard:frames[,shard:frames...]" */
    while (*e) {
      char *end = NULL;
      const long s = strtol (e, &end, 10);
      if (end == e || *end != ':')
        break;
      e = end + 1;
      const long long n = strtoll (e, &end, 10);
      if (end == e)
        break;
      (void) mibayer_pool_inject_fault (pool, (int) s, n);
      e = (*end == ',') ? end + 1 : end;
    }
  }
  if (const char *e = getenv ("MIBAYER_INJECT_STALL")) {
    /* "shard:frames:ms[,shard:frames:ms...]" */
    while (*e) {
      char *end = NULL;
      const long s = strtol (e, &end, 10);
      if (end == e || *end != ':')
        break;
      e = end + 1;
      const long long n = strtoll (e, &end, 10);
      if (end == e || *end != ':')
        break;
      e = end + 1;
      const long ms = strtol (e, &end, 10);
      if (end == e)
        break;
      if (s >= 0 && s < (long) pool->shards.size () && n >= 0 && ms > 0) {
        pool->shards[(size_t) s]->stall_after = n;
        pool->shards[(size_t) s]->stall_ms = (int) ms;
      }
      e = (*end == ',') ? end + 1 : end;
    }
  }
  *out = pool;
  return MIBAYER_OK;
}

extern "C" void mibayer_pool_destroy (mibayer_pool *pool)
{
  if (!pool)
    return;
  for (Shard *sh : pool->shards) {
    if (sh->th_started) {
      {
        std::lock_guard<std::mutex> lk (sh->mu);
        sh->quit = true;
        sh->broken = true;      /* queued jobs are not started any more */
        sh->cv_job.notify_all ();
      }
      sh->th.join ();
    }
    mibayer_destroy (sh->ctx);
    delete sh;
  }
  delete pool;
}

extern "C" int mibayer_pool_capacity (const mibayer_pool *pool)
{
  if (!pool)
    return MIBAYER_ERR_ARG;
  return alive_count (pool) * pool->per_shard;
}

extern "C" int mibayer_pool_alive (const mibayer_pool *pool)
{
  return pool ? alive_count (pool) : MIBAYER_ERR_ARG;
}

extern "C" int mibayer_pool_pending (const mibayer_pool *pool)
{
  return pool ? (int) pool->fifo.size () : MIBAYER_ERR_ARG;
}

extern "C" int mibayer_pool_inject_fault (mibayer_pool *pool, int shard,
    long long after_frames)
{
  if (!pool || shard < 0 || shard >= (int) pool->shards.size ())
    return MIBAYER_ERR_ARG;
  Shard *sh = pool->shards[(size_t) shard];
  sh->fail_after.store (after_frames < 0 ? -1
      : sh->completions.load () + after_frames);
  return MIBAYER_OK;
}

extern "C" int mibayer_pool_set_wait_timeout (mibayer_pool *pool, int ms)
{
  if (!pool)
    return MIBAYER_ERR_ARG;
  for (Shard *sh : pool->shards)
    (void) mibayer_set_wait_timeout (sh->ctx, ms);
  return MIBAYER_OK;
}

extern "C" int mibayer_pool_inject_stall (mibayer_pool *pool, int shard, int ms)
{
  if (!pool || shard < 0 || shard >= (int) pool->shards.size ())
    return MIBAYER_ERR_ARG;
  return mibayer_internal_stall (pool->shards[(size_t) shard]->ctx, ms);
}

extern "C" int mibayer_pool_take_failure (mibayer_pool *pool, int *device,
    int *alive, char *msg, size_t msg_len)
{
  if (!pool)
    return MIBAYER_ERR_ARG;
  if (pool->unreported == 0)
    return 0;
  const int n = pool->unreported;
  pool->unreported = 0;
  if (device)
    *device = pool->failed_device;
  if (alive)
    *alive = alive_count (pool);
  if (msg && msg_len)
    snprintf (msg, msg_len, "%s", pool->failure_msg);
  return n;
}

extern "C" int mibayer_pool_submit (mibayer_pool *pool, const uint8_t *src,
    uint8_t *dst, void *tag)
{
  if (!pool || !src || !dst)
    return MIBAYER_ERR_ARG;
  const size_t n = pool->shards.size ();
  for (;;) {
    /* the next live shard in rotation */
    size_t idx = n;
    for (size_t k = 0; k < n; k++) {
      const size_t i = (pool->rr + k) % n;
      if (pool->shards[i]->alive) {
        idx = i;
        break;
      }
    }
    if (idx == n)
      return MIBAYER_ERR_HIP;   /* no device left */
    if (reap_if_broken (pool, (int) idx))
      continue;
    const size_t turn = idx;
    if (pool->numa_route) {
      /* the live shard next to the frame's big buffer, if taking it keeps the rotation balanced */
      const void *big = pool->inverse ? (const void *) src : (const void *) dst;
      int node;
      auto it = pool->node_of.find (big);
      if (it != pool->node_of.end ()) {
        node = it->second;
      } else {
        node = mibayer_host_numa_node (big);
        if (pool->node_of.size () > 4096)
          pool->node_of.clear ();
        pool->node_of.emplace (big, node);
      }
      if (node >= 0 && pool->shards[turn]->node != node) {
        for (size_t k = 1; k < n; k++) {
          const size_t i = (turn + k) % n;
          Shard *cand = pool->shards[i];
          if (cand->alive && cand->node == node && cand->inflight < pool->per_shard
              && cand->assigned < pool->shards[turn]->assigned + pool->per_shard) {
            bool broken;
            {
              std::lock_guard<std::mutex> lk (cand->mu);
              broken = cand->broken;
            }
            if (!broken) {
              idx = i;
              break;
            }
          }
        }
      }
    }
    Shard *sh = pool->shards[idx];
    if (sh->inflight >= pool->per_shard)
      return MIBAYER_ERR_BUSY;
    if (sh->stall_ms > 0 && sh->assigned >= sh->stall_after) {  /* drill: MIBAYER_INJECT_STALL */
      const int ms = sh->stall_ms;
      sh->stall_ms = 0;
      (void) mibayer_internal_stall (sh->ctx, ms);
    }
    Frame f = { src, dst, tag, (int) idx, (int) idx, F_DIRECT, MIBAYER_OK, false };
    if (pool->use_helpers && (mibayer_internal_is_pageable (src) || mibayer_internal_is_pageable (dst))
        && !sh->pageable_seen) {
      sh->pageable_seen = true;
      enter_helper_mode (pool, (int) idx, true);
    }
    if (sh->helper_mode) {
      pool->fifo.push_back (f);
      queue_to_helper (sh, &pool->fifo.back ());
    } else {
      const int rc = mibayer_submit (sh->ctx, src, dst, tag);
      if (device_failure (rc)) {
        kill_shard (pool, (int) idx, rc, mibayer_last_hip_error ());
        continue;               /* this frame goes to the next live shard */
      }
      if (rc != MIBAYER_OK)
        return rc;
      f.submitted = true;
      pool->fifo.push_back (f);
    }
    sh->inflight++;
    sh->assigned++;
    if (idx == turn)
      pool->rr = (idx + 1) % n;
    return MIBAYER_OK;
  }
}

extern "C" int mibayer_pool_wait (mibayer_pool *pool, void **tag)
{
  if (!pool)
    return MIBAYER_ERR_ARG;
  if (pool->fifo.empty ())
    return MIBAYER_ERR_EMPTY;
  Frame &f = pool->fifo.front ();
  const size_t n = pool->shards.size ();
  for (;;) {
    Shard *sh = pool->shards[(size_t) f.shard];
    int state;
    {
      std::unique_lock<std::mutex> lk (sh->mu);
      sh->cv_done.wait (lk, [&f] {
        return f.state != F_QUEUED && f.state != F_RUNNING;
      });
      state = f.state;
    }
    if (state == F_DONE)
      break;
    if (state == F_DIRECT) {
      int rc = mibayer_wait (sh->ctx, NULL);
      if (rc == MIBAYER_OK && sh->fault_due ())
        rc = MIBAYER_ERR_HIP;
      if (rc == MIBAYER_OK)
        break;
      if (!device_failure (rc))
        return rc;
      kill_shard (pool, f.shard, rc, mibayer_last_hip_error ());    /* f becomes F_REDO */
      continue;
    }
    if (state == F_FAILED)
      return f.rc;              /* not a device failure (those become F_REDO): the caller's to handle */
    if (state == F_LOST) {
      /* The helper thread may have marked it while the shard still counts as alive to this thread: take the shard
       * out first, so that mibayer_pool_alive() / take_failure already tell the caller what happened
       * when the lost frame comes back, and kill_shard() has its last word on held-vs-quiesced (a context that
       * could be quiesced after all turns the frame into F_REDO: converted elsewhere instead of lost). */
      (void) reap_if_broken (pool, f.shard);
      {
        std::lock_guard<std::mutex> lk (sh->mu);
        state = f.state;
      }
      if (state != F_LOST)
        continue;
      /* in flight on a device that ran into the deadline: handed back as lost, in order; its buffers stay the
       * device's until mibayer_pool_reclaim() says otherwise */
      pool->lost.push_back (mibayer_pool::Lost { f.tag, f.shard });
      if (tag)
        *tag = f.tag;
      pool->shards[(size_t) f.owner]->inflight--;
      pool->fifo.pop_front ();
      return MIBAYER_ERR_TIMEOUT;
    }
    /* F_REDO: once more on a device that is still in the rotation.  First the shard it came from: if its helper
     * thread saw the failure, the shard is still "alive" to this thread and its context still has the frame's
     * copies queued -- take it out and abandon it BEFORE the frame is converted elsewhere and handed back, or a late
     * DMA of the dead context lands in a buffer the caller has already released. */
    (void) reap_if_broken (pool, f.shard);
    {
      std::lock_guard<std::mutex> lk (sh->mu);
      if (f.state == F_LOST)    /* the quiesce ran into the deadline: not converted again after all */
        continue;
    }
    size_t idx = n;
    for (size_t k = 0; k < n; k++) {
      const size_t i = (pool->redo_rr + k) % n;
      if (pool->shards[i]->alive) {
        idx = i;
        break;
      }
    }
    if (idx == n)
      return MIBAYER_ERR_HIP;   /* no device left: the stream has failed */
    if (reap_if_broken (pool, (int) idx))
      continue;
    pool->redo_rr = (idx + 1) % n;
    Shard *to = pool->shards[idx];
    f.shard = (int) idx;
    f.submitted = false;
    if (to->helper_mode) {
      /* its helper thread owns the context: an ordinary job for it */
      queue_to_helper (to, &f);
      continue;
    }
    f.submitted = true;         /* while the spare slot of `to` works on it */
    f.state = F_REDO;
    int rc = mibayer_internal_run_spare (to->ctx, f.src, f.dst);
    if (rc == MIBAYER_OK && to->fault_due ())
      rc = MIBAYER_ERR_HIP;
    if (rc == MIBAYER_OK)
      break;
    if (!device_failure (rc))
      return rc;
    kill_shard (pool, (int) idx, rc, mibayer_last_hip_error ());       /* F_REDO again, or F_LOST after a deadline */
  }
  if (tag)
    *tag = f.tag;
  pool->shards[(size_t) f.owner]->inflight--;
  pool->fifo.pop_front ();
  return MIBAYER_OK;
}

extern "C" int mibayer_pool_lost (const mibayer_pool *pool)
{
  return pool ? (int) pool->lost.size () : MIBAYER_ERR_ARG;
}

extern "C" int mibayer_pool_reclaim (mibayer_pool *pool, void **tag)
{
  if (!pool)
    return MIBAYER_ERR_ARG;
  for (size_t i = 0; i < pool->lost.size (); i++) {
    const mibayer_pool::Lost l = pool->lost[i];
    if (mibayer_internal_settled (pool->shards[(size_t) l.shard]->ctx)) {
      if (tag)
        *tag = l.tag;
      pool->lost.erase (pool->lost.begin () + (long) i);
      return MIBAYER_OK;
    }
  }
  return MIBAYER_ERR_EMPTY;
}

extern "C" int mibayer_pool_set_wait_spin (mibayer_pool *pool, int spin_us)
{
  if (!pool)
    return MIBAYER_ERR_ARG;
  for (Shard *sh : pool->shards)
    (void) mibayer_set_wait_spin (sh->ctx, spin_us);
  return MIBAYER_OK;
}

extern "C" int mibayer_pool_get_host_stats (const mibayer_pool *pool, mibayer_host_stats *out)
{
  if (!pool || !out)
    return MIBAYER_ERR_ARG;
  memset (out, 0, sizeof *out);
  for (const Shard *sh : pool->shards) {
    mibayer_host_stats one;
    if (mibayer_get_host_stats (sh->ctx, &one) != MIBAYER_OK)
      continue;
    out->submits += one.submits;
    out->waits += one.waits;
    out->polls += one.polls;
    out->naps += one.naps;
    out->submit_cpu_ms += one.submit_cpu_ms;
    out->wait_cpu_ms += one.wait_cpu_ms;
    out->wait_wall_ms += one.wait_wall_ms;
  }
  return MIBAYER_OK;
}
