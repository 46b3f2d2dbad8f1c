/*
 * C-ABI shim of the MI355X bayer2rgb path (include/mibayer.h).
 *
 * Host-side replacement for the body of gst_bayer2rgb_process
 * (reference gst/bayer/gstbayer2rgb.c:387-451): what the reference does per
 * frame on the streaming thread -- pick the merge pair from the byte layout and
 * the Bayer order (:400-427), allocate scratch (:429), loop over rows (:438-448)
 * -- becomes, here, a per-stream plan computed once in mibayer_create() (v_perm
 * selectors, row-type swap, tile grid) and one kernel launch per frame or batch.
 * No CPU compute path exists in this library.
 */
#include "../../include/mibayer.h"
#include "mibayer_hooks.h"
#include "mibayer_internal.h"

#include <dlfcn.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <atomic>
#include <limits.h>
#include <mutex>
#include <new>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <vector>

using namespace mibayer;

/* Tuning / experiment knobs exist in the lab build only (`make lab`, -DMIBAYER_LAB: tools/, the sweep tests): the
 * product library reads MIBAYER_ROCTX, MIBAYER_WAIT_TIMEOUT_MS, MIBAYER_WAIT_SPIN_US, MIBAYER_PLAN_CACHE and the
 * pool's operational variables, nothing else. */
#ifdef MIBAYER_LAB
#define LAB_GETENV(name) getenv (name)
#else
#define LAB_GETENV(name) ((const char *) nullptr)
#endif

namespace {

thread_local char t_hip_error[256] = "";

bool hip_failed (hipError_t e, const char *what)
{
  if (e == hipSuccess)
    return false;
  snprintf (t_hip_error, sizeof t_hip_error, "%s: %s (%d)", what,
      hipGetErrorString (e), (int) e);
  return true;
}

#define HIP_TRY(expr)                                                          \
  do {                                                                         \
    if (hip_failed ((expr), #expr))                                            \
      return MIBAYER_ERR_HIP;                                                  \
  } while (0)

/* hipSetDevice is per-thread state: select the context's device for the call
 * and put the caller's device back afterwards. */
struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard (int dev)
  {
    if (hipGetDevice (&prev) != hipSuccess)
      prev = -1;
    if (prev != dev)
      ok = !hip_failed (hipSetDevice (dev), "hipSetDevice");
    else
      prev = -1;
  }
  ~DeviceGuard ()
  {
    if (prev >= 0)
      (void) hipSetDevice (prev);
  }
};

/* ROCTx ranges around the host side of every stage (upload / kernel / download
 * enqueue, the wait for a frame, a device-resident launch): `rocprofv3
 * --marker-trace` shows them on the timeline next to the copies and kernels they
 * queue.  The reference's tracer hooks on this path are its debug category and
 * the GST_DEBUG lines of the transform (gstbayer2rgb.c:92-93, :201, :465).  The
 * ROCTx library is looked up at run time (rocprofiler-sdk's, then the legacy
 * one): libmibayer.so keeps depending on the HIP runtime only, and without the
 * library a range costs one predictable branch. */
struct Roctx {
  int (*push) (const char *) = nullptr;
  int (*pop) () = nullptr;
};

const Roctx &roctx ()
{
  static const Roctx r = [] {
    Roctx x;
    const char *off = getenv ("MIBAYER_ROCTX");
    if (off && off[0] == '0')
      return x;
    for (const char *name : { "librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so",
            "libroctx64.so.4", "libroctx64.so" }) {
      void *h = dlopen (name, RTLD_NOW | RTLD_LOCAL);
      if (!h)
        continue;
      x.push = (int (*) (const char *)) dlsym (h, "roctxRangePushA");
      x.pop = (int (*) ()) dlsym (h, "roctxRangePop");
      if (x.push && x.pop)
        return x;
      x.push = nullptr;
      x.pop = nullptr;
      dlclose (h);
    }
    return x;
  } ();
  return r;
}

struct Range {
  bool on;
  explicit Range (const char *name) : on (roctx ().push != nullptr)
  {
    if (on)
      (void) roctx ().push (name);
  }
  ~Range ()
  {
    if (on)
      (void) roctx ().pop ();
  }
};

constexpr int kMaxHostBands = 8;
constexpr int kInverseBandUnit = 16;    /* rows; a multiple of every rows-per-block of the rgb2bayer kernel */

struct Slot {
  uint8_t *d_src = nullptr;
  uint8_t *d_dst = nullptr;
  hipEvent_t ev_in = nullptr;     /* H2D done   */
  hipEvent_t ev_kernel = nullptr; /* kernel done */
  hipEvent_t ev_out = nullptr;    /* D2H done   */
  /* banded frames (mibayer_ctx::host_bands > 1): upload / kernel done, per band */
  hipEvent_t ev_band_in[kMaxHostBands] = {};
  hipEvent_t ev_band_kernel[kMaxHostBands] = {};
  void *tag = nullptr;
  /* MIBAYER_FLAG_HIPGRAPH: the frame's upload -> kernel -> download chain as one
   * instantiated graph, launched on the slot's own stream */
  hipStream_t s_graph = nullptr;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  hipGraphNode_t n_h2d = nullptr, n_kernel = nullptr, n_d2h = nullptr;
  const void *g_src = nullptr;    /* host pointers currently baked into exec */
  void *g_dst = nullptr;
  /* MIBAYER_FLAG_HIPGRAPH, default form: the compute-queue segment of the frame
   * (wait for the upload event -> kernel -> record the kernel event) captured
   * once per slot; the copies stay on the copy queues */
  hipGraph_t cgraph = nullptr;
  hipGraphExec_t cexec = nullptr;
  bool cgraph_events = false;     /* the graph holds the event nodes too */
  unsigned plan_epoch = 0;        /* mibayer_ctx::plan_epoch both graphs were built under */
};

int device_count_cached ()
{
  static int n = -1;
  static std::once_flag once;
  std::call_once (once, [] {
    int c = 0;
    if (hipGetDeviceCount (&c) != hipSuccess)
      c = 0;
    n = c;
  });
  return n;
}

}  /* namespace */

/* The three queues of the host path (uploads, kernels, downloads), one set per
 * DEVICE, shared by every context on it.  Contexts used to own their queues;
 * with N contexts on one GPU (devices=0,0,0,0, or several elements on one card)
 * that put N download queues on the one PCIe-facing DMA engine, and the engine
 * spent its time switching between them: 4 shards x 2 frames through pinned
 * buffers ran at 25.6 GB/s D2H against 53 GB/s for one context
 * (profiles/r02_pool_pageable.log).  In one queue per direction the copies go
 * back to back; frames of different contexts are independent chains of events,
 * so sharing a queue costs no concurrency that the hardware could have used --
 * there is one DMA engine per direction and one kernel fills the GPU.
 * MIBAYER_SHARED_QUEUES=0 restores private queues (A/B). */
struct DeviceQueues {
  hipStream_t h2d = nullptr, compute = nullptr, d2h = nullptr;
  int refs = 0;
};

static std::mutex g_queues_mu;
static DeviceQueues g_queues[64];

/* Frame queues: MIBAYER_FRAME_QUEUES compute streams per DEVICE, each on a HARDWARE QUEUE OF ITS OWN, for independent
 * one-frame launches.  A launch over one frame is a single round of workgroups -- ramp-up, one burst of loads, one
 * burst of stores, drain: 4K 9.4 us against 6.5 us per frame inside a batch -- and never reaches a steady state by
 * itself; dealt round-robin over four queues, the ramp-up of frame n+1..n+3 overlaps the drain of frame n (4K: 54 %
 * of HBM peak on one queue, 63 % on two, 66-67 % on four; rgb2bayer 55 / 70 / 77-78 %; profiles/r05_single_frame.md).
 * Ordinary HIP streams do not do: the runtime multiplexes them onto a small pool of hardware queues (four per process
 * by default, shared with the copy queues), so "four streams" are two or three queues with launches serialised behind
 * each other again (3 streams measured WORSE than 2).  A stream made by hipExtStreamCreateWithCUMask owns its hardware
 * queue; the mask given is "every CU" -- a lone frame still gets the whole device -- partitioning the CUs instead
 * (queue k owns the CUs with index % 4 == k) measured within a point of it.  Hardware queues are a finite resource
 * (59 of them in one process slowed every launch of that process by 2x), hence per device and shared by its contexts.
 * Created on first use and NEVER destroyed -- they live as long as the process, like the runtime's own queue pool:
 * hipStreamDestroy of a CU-masked stream hangs intermittently in ROCm 7.2 (a context that made the four queues, ran
 * one host-path frame and was destroyed hung in 8 of 24 fresh processes, in 3 of 24 with a hipStreamSynchronize before
 * each destroy, in 0 of 24 when the queues were kept; a bare HIP program that destroys such streams with work pending:
 * 2 of 25; profiles/r05_frame_queue_teardown.log).  Four idle hardware queues per device are what that costs. */
struct FrameQueues {
  hipStream_t q[MIBAYER_FRAME_QUEUES] = {};
  bool tried = false;
};
static FrameQueues g_frame_queues[64];  /* g_queues_mu */

/* Device frames of contexts that were destroyed, kept for the next context of the same geometry on that device
 * (renegotiation, one element after another): hipFree waits for EVERY queue of the device to drain, so a context
 * that frees its ring on the way out would wait for its neighbours' batches.  Bounded (kCacheMax blocks per device);
 * emptied when the last context of the device goes.  Guarded by g_queues_mu. */
struct BufCache {
  std::vector<std::pair<void *, size_t>> bufs;
  size_t bytes = 0;
  int contexts = 0;
};
constexpr size_t kCacheMax = 16;
constexpr size_t kCacheMaxBytes = (size_t) 1 << 30;     /* per device */
static BufCache g_cache[64];

static hipError_t cached_malloc (int dev, void **p, size_t bytes)
{
  if (dev >= 0 && dev < 64) {
    std::lock_guard<std::mutex> lk (g_queues_mu);
    std::vector<std::pair<void *, size_t>> &v = g_cache[dev].bufs;
    for (size_t i = 0; i < v.size (); i++)
      if (v[i].second == bytes) {
        *p = v[i].first;
        v.erase (v.begin () + (long) i);
        g_cache[dev].bytes -= bytes;
        return hipSuccess;
      }
  }
  return hipMalloc (p, bytes);
}

static void cached_free (int dev, void *p, size_t bytes)
{
  if (!p)
    return;
  if (dev >= 0 && dev < 64) {
    std::lock_guard<std::mutex> lk (g_queues_mu);
    if (g_cache[dev].bufs.size () < kCacheMax && g_cache[dev].bytes + bytes <= kCacheMaxBytes) {
      g_cache[dev].bufs.emplace_back (p, bytes);
      g_cache[dev].bytes += bytes;
      return;
    }
  }
  (void) hipFree (p);
}

static bool shared_queues_enabled ()
{
  static const bool on = [] {
    const char *e = LAB_GETENV ("MIBAYER_SHARED_QUEUES");
    return !(e && e[0] == '0');
  } ();
  return on;
}

struct Wedge;

enum { PLAN_BATCH = 0, PLAN_FRAME = 1, PLAN_CLASSES = 2 };
struct Plan {
  const Variant *var = nullptr;
  int band = INT32_MIN;
  int align = 0;
  int source = MIBAYER_PLAN_DEFAULT;
};

struct mibayer_ctx {
  mibayer_cfg cfg;
  int device = 0;
  size_t src_bytes = 0;         /* one frame */
  size_t dst_bytes = 0;
  /* plan */
  uint32_t sel[4];
  int swap_rows = 0;
  bool inverse = false;                 /* MIBAYER_FLAG_RGB2BAYER */
  uint32_t r2b_lo[2], r2b_hi[2];        /* rgb2bayer v_perm selectors per row parity */
  /* Launch plan: tile shape (kernel variant), XCD band (INT32_MIN = the variant's; MIBAYER_XCD_BAND or
   * mibayer_autotune() set it), store alignment of the generic arm, and where the three came from
   * (mibayer_plan_source).  One plan per LAUNCH CLASS (round 5): PLAN_BATCH for launches that keep every workgroup
   * slot of the device busy for many rounds (the 64-frame batch of the bench), PLAN_FRAME for launches of at most
   * kFrameClassRounds rounds -- one frame per launch, what the elements and the host path issue -- where the number
   * of rounds the grid needs decides (launch_class(), frame_class_variant()).  A plan measured on one class never
   * becomes the default of the other. */
  Plan plan[PLAN_CLASSES];
  unsigned plan_epoch = 0;              /* bumped by every plan change: graphs captured under an older one are rebuilt */
  int num_cus = 256;                    /* hipDeviceProp_t.multiProcessorCount */
  int persist_wgs_per_cu = 4;           /* MIBAYER_PERSIST_WGS (tuning), persistent arms */
  bool rows_off_sector = false;         /* dst_stride % 64 != 0: see plain_store_twin () */
  /* Plan::align -- generic geometries with output rows off the sector grid: boundary (bytes) every wave-store starts
   * on (the shifted arm, bayer2rgb_lds_aligned_kernel); 0 = the unshifted generic arm.  Chosen by mibayer_autotune()
   * where it wins; MIBAYER_ALIGN_STORES = 0 | 64 | 128 forces it (-1 in the environment keeps it out of the
   * autotuner's candidates) */
  bool align_tunable = true;
  bool band_forced = false;             /* the block order was pinned from outside (lab builds: MIBAYER_XCD_BAND) */
  int graph_mode = 0;                   /* MIBAYER_FLAG_HIPGRAPH: 0 = the compute-queue segment of a frame as
                                           a graph per slot (default), 1 = the whole upload -> kernel ->
                                           download chain as a graph per slot on the slot's own queue
                                           (MIBAYER_GRAPH_MODE=chain; A/B arm) */
  int host_bands = 1;                   /* host path: horizontal bands a frame is cut into so that
                                           the upload of band b+1, the kernel of band b and the
                                           download of band b-1 overlap inside ONE frame */
  /* rgb2bayer launch shape (R2BParams); MIBAYER_R2B_FLAT / _PX / _LDNT / _ROWS override */
  int r2b_flat_k = 2, r2b_flat_px = 4, r2b_flat_ld = 1, r2b_rows = 2;
  int start_sleep = -1;                 /* s_sleep(1) iterations before a workgroup's first load;
                                           -1 = automatic (kStartSleepChunk with a band map on
                                           large grids, else 0); MIBAYER_START_SLEEP overrides */
  /* streams: uploads, kernels and downloads each get their own queue so that
   * frame n+1's H2D overlaps frame n's kernel and frame n-1's D2H */
  hipStream_t s_h2d = nullptr;
  hipStream_t s_compute = nullptr;
  hipStream_t s_d2h = nullptr;
  bool shared_queues = false;           /* the three above belong to g_queues[device] */
  /* Frame queues (g_frame_queues[device], mibayer_ctx_frame_queue): which of them this context has queued
   * device-resident work on since the last mibayer_sync, and whether it has ever asked for them (wedge fences) */
  bool dirty_frame[MIBAYER_FRAME_QUEUES] = {};
  bool uses_frame_queues = false;
  hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;
  hipEvent_t ev_fence = nullptr;        /* fence of mibayer_sync / of a failed submit */
  /* Deadline of every host-side wait for the device (mibayer_wait, the synchronous frame call, sync, destroy):
   * a GPU that has stopped answering returns nothing, so an unbounded hipEventSynchronize would hang the streaming
   * thread for good.  0 = wait for ever.  MIBAYER_WAIT_TIMEOUT_MS / mibayer_set_wait_timeout(). */
  int wait_timeout_ms = 10000;
  /* How long a host-side wait may spin on hipEventQuery before it starts to nap, in microseconds; -1 = automatic:
   * up to kAutoSpinUs when the frame waited for is the only one in flight (the synchronous 1-in/1-out use: the caller
   * can do nothing until it completes, and a nap's wake-up latency comes straight off the frame rate), none when
   * other frames are queued behind it (their copies and kernels keep the device busy while this thread sleeps).
   * MIBAYER_WAIT_SPIN_US / mibayer_set_wait_spin(). */
  int wait_spin_us = -1;
  mibayer_host_stats stats = {};        /* host CPU spent in submits and waits (mibayer_get_host_stats) */
  bool wedged = false;                  /* a wait ran into the deadline: nothing of this context is waited for again */
  Wedge *wedge = nullptr;               /* ... and this tells when the device has caught up with it (registry above) */
  bool counted = false;                 /* in g_cache[device].contexts */
  bool dirty_compute = false;           /* device-resident work was queued on s_compute through this context since
                                           the last mibayer_sync */
  std::vector<Slot> ring;
  Slot spare;                   /* mibayer_internal_run_spare: outside the ring */
  bool spare_ready = false;
  int head = 0;                 /* next slot to submit into */
  int tail = 0;                 /* oldest in-flight slot    */
  int pending = 0;
};

/* ---- bounded waits ---------------------------------------------------------- */

/* A context whose wait ran into its deadline ("wedged").  The device may only be slow -- the stall drill ends by
 * itself, a neighbour's long kernel ends, a reset completes -- so the state is neither global nor permanent: when the
 * deadline hits, a fence is recorded behind everything the context had queued; once those fences have fired the device
 * has caught up with the context ("settled").  Until then
 *   - the buffers of its in-flight frames belong to the device (mibayer_internal_settled, mibayer_pool_reclaim),
 *   - nothing of it is waited for or released by a call that could block behind the device: a context destroyed
 *     meanwhile leaves its device frames, events and queues to the registry, which releases them when the fences fire,
 *   - pinned blocks handed to mibayer_host_free go on a deferred list instead of hipHostFree (which waits for the
 *     devices to drain) and are freed when no wedge is outstanding any more.
 * The registry is polled -- hipEventQuery, never a blocking call -- from mibayer_host_free, mibayer_create,
 * mibayer_destroy and mibayer_internal_settled. */
struct Wedge {
  int device = 0;
  std::vector<hipEvent_t> fences;       /* behind everything the context had queued when the deadline hit */
  bool settled = false;
  bool incomplete = false;              /* a queue could not be fenced: never settles (fail closed) */
  bool orphan = false;                  /* its context has been destroyed: the registry owns what follows */
  std::vector<void *> dev_mem;
  std::vector<hipEvent_t> events;
  std::vector<hipStream_t> streams;
  std::vector<hipGraphExec_t> execs;
  std::vector<hipGraph_t> graphs;
};

static std::mutex g_wedge_mu;
static std::vector<Wedge *> g_wedges;
static std::vector<void *> g_deferred_host;     /* pinned blocks whose hipHostFree is waiting for the wedges to settle */
static std::atomic<int> g_deferred_count { 0 }; /* its size, for the lock-free fast paths */
static std::atomic<int> g_wedges_open { 0 };    /* unsettled wedges: the fast path of mibayer_host_free */

static void release_wedge_resources (Wedge *w)
{
  for (hipGraphExec_t e : w->execs)
    (void) hipGraphExecDestroy (e);
  for (hipGraph_t g : w->graphs)
    (void) hipGraphDestroy (g);
  for (hipEvent_t e : w->events)
    (void) hipEventDestroy (e);
  for (hipStream_t st : w->streams)
    (void) hipStreamDestroy (st);
  for (void *m : w->dev_mem)
    (void) hipFree (m);
  for (hipEvent_t e : w->fences)
    (void) hipEventDestroy (e);
  (void) hipGetLastError ();
}

/* errors after which nothing queued on the device will run any more (sticky: the process has lost its GPU state) */
static bool device_is_gone (hipError_t e)
{
  switch (e) {
    case hipErrorIllegalAddress:
    case hipErrorLaunchFailure:
    case hipErrorECCNotCorrectable:
    case hipErrorNoDevice:
    case hipErrorDeinitialized:
    case hipErrorContextIsDestroyed:
      return true;
    default:
      return false;
  }
}

/* g_wedge_mu held.  Non-blocking: one hipEventQuery per outstanding fence. */
static void poll_wedges_locked ()
{
  for (size_t i = 0; i < g_wedges.size ();) {
    Wedge *w = g_wedges[i];
    if (!w->settled) {
      int prev = -1;
      (void) hipGetDevice (&prev);
      if (prev != w->device)
        (void) hipSetDevice (w->device);
      /* Fail closed: a wedge that could not fence every queue never settles (its buffers are leaked
       * rather than handed back under a DMA that may still run), and only hipSuccess or an error that says the
       * device context is gone -- nothing queued will ever execute -- ends the wait for a fence. */
      bool done = !w->incomplete;
      for (hipEvent_t e : w->fences) {
        if (!done)
          break;
        const hipError_t q = hipEventQuery (e);
        if (q != hipSuccess && !device_is_gone (q))
          done = false;
      }
      (void) hipGetLastError ();
      if (done) {
        w->settled = true;
        g_wedges_open.fetch_sub (1);
        if (w->orphan)
          release_wedge_resources (w);
      }
      if (prev >= 0 && prev != w->device)
        (void) hipSetDevice (prev);
    }
    if (w->settled && w->orphan) {
      delete w;
      g_wedges.erase (g_wedges.begin () + (long) i);
      continue;
    }
    i++;
  }
  if (g_wedges_open.load () == 0 && !g_deferred_host.empty ()) {
    for (void *p : g_deferred_host)
      (void) hipHostFree (p);
    g_deferred_host.clear ();
    g_deferred_count.store (0);
  }
}

static void poll_wedges ()
{
  if (g_wedges_open.load () == 0 && g_deferred_count.load () == 0)
    return;
  std::lock_guard<std::mutex> lk (g_wedge_mu);
  poll_wedges_locked ();
}

static double now_ms ()
{
  timespec t;
  clock_gettime (CLOCK_MONOTONIC, &t);
  return (double) t.tv_sec * 1e3 + (double) t.tv_nsec * 1e-6;
}

/* the wait deadline has hit: fences behind everything the context has queued (recording never blocks) */
static void on_deadline (mibayer_ctx *c)
{
  c->wedged = true;
  if (c->wedge)
    return;
  Wedge *w = new (std::nothrow) Wedge ();
  if (!w)
    return;
  w->device = c->device;
  std::vector<hipStream_t> queues = { c->s_h2d, c->s_compute, c->s_d2h };
  if (c->uses_frame_queues && c->device < 64) {
    std::lock_guard<std::mutex> lk (g_queues_mu);
    for (hipStream_t q : g_frame_queues[c->device].q)
      queues.push_back (q);
  }
  for (Slot &sl : c->ring)
    if (sl.s_graph)
      queues.push_back (sl.s_graph);
  for (hipStream_t q : queues) {
    hipEvent_t ev = nullptr;
    if (!q)
      continue;
    if (hipEventCreateWithFlags (&ev, hipEventDisableTiming) != hipSuccess) {
      w->incomplete = true;     /* a queue without a fence: this wedge can never be called settled */
      continue;
    }
    if (hipEventRecord (ev, q) != hipSuccess) {
      (void) hipEventDestroy (ev);
      w->incomplete = true;
      continue;
    }
    w->fences.push_back (ev);
  }
  (void) hipGetLastError ();
  c->wedge = w;
  std::lock_guard<std::mutex> lk (g_wedge_mu);
  g_wedges.push_back (w);
  g_wedges_open.fetch_add (1);
}

/* has the device caught up with a context that ran into a deadline?  (true for one that never did) */
static bool ctx_settled (mibayer_ctx *c)
{
  if (!c->wedged)
    return true;
  if (!c->wedge)
    return false;
  std::lock_guard<std::mutex> lk (g_wedge_mu);
  poll_wedges_locked ();
  return c->wedge->settled;
}

/* A context that ran into a deadline works again once the device has caught up with everything it had queued at
 * that moment (its frames' events have fired, so they can be waited for and handed back in order): a spurious timeout
 * -- a small deadline, a neighbour process hogging the GPU, the stall drill -- heals by itself.  Polls, never waits. */
static bool wedged_for_good (mibayer_ctx *c)
{
  if (!c->wedged)
    return false;
  if (!ctx_settled (c))
    return true;
  std::lock_guard<std::mutex> lk (g_wedge_mu);
  for (size_t i = 0; i < g_wedges.size (); i++)
    if (g_wedges[i] == c->wedge)
      g_wedges.erase (g_wedges.begin () + (long) i);
  for (hipEvent_t e : c->wedge->fences)
    (void) hipEventDestroy (e);
  delete c->wedge;
  c->wedge = nullptr;
  c->wedged = false;
  return false;
}

static double thread_cpu_ms ()
{
  timespec t;
  clock_gettime (CLOCK_THREAD_CPUTIME_ID, &t);
  return (double) t.tv_sec * 1e3 + (double) t.tv_nsec * 1e-6;
}

/* adds the calling thread's CPU time between construction and destruction to one of the context's counters */
struct CpuMeter {
  double *sink;
  double t0;
  explicit CpuMeter (double *s) : sink (s), t0 (thread_cpu_ms ()) {}
  ~CpuMeter () { *sink += thread_cpu_ms () - t0; }
};

constexpr int kAutoSpinUs = 2000;       /* a 4K frame through the host path takes 0.6-0.8 ms */

/* Host wait for an event, bounded by the context's deadline (0 = none).  The event is polled: a tight loop for the
 * spin window (see mibayer_ctx::wait_spin_us; `alone` = nothing else of this context is queued behind what is waited
 * for), then naps that double from 20 us to 250 us -- a thread that waits for a frame while the next ones are already
 * queued costs a few wake-ups per frame instead of a core (profiles/r04_host_cpu.log).  MIBAYER_ERR_TIMEOUT marks the
 * context wedged.  Reference analogue of a bounded wait on a stream that may stall:
 * gst/debugutils/gstwatchdog.c:21-123. */
static int wait_event (mibayer_ctx *c, hipEvent_t ev, bool alone)
{
  if (wedged_for_good (c))
    return MIBAYER_ERR_TIMEOUT;
  c->stats.waits++;
  CpuMeter cpu (&c->stats.wait_cpu_ms);
  const double spin_ms = (c->wait_spin_us >= 0 ? c->wait_spin_us : (alone ? kAutoSpinUs : 0)) * 1e-3;
  const double t0 = now_ms ();
  long nap_ns = 20000;
  for (;;) {
    const hipError_t e = hipEventQuery (ev);
    c->stats.polls++;
    if (e == hipSuccess) {
      c->stats.wait_wall_ms += now_ms () - t0;
      return MIBAYER_OK;
    }
    if (e != hipErrorNotReady) {
      (void) hip_failed (e, "hipEventQuery");
      return MIBAYER_ERR_HIP;
    }
    (void) hipGetLastError ();
    const double dt = now_ms () - t0;
    if (c->wait_timeout_ms > 0 && dt >= (double) c->wait_timeout_ms) {
      on_deadline (c);
      c->stats.wait_wall_ms += dt;
      snprintf (t_hip_error, sizeof t_hip_error,
          "HIP device %d did not complete a frame within %d ms", c->device, c->wait_timeout_ms);
      return MIBAYER_ERR_TIMEOUT;
    }
    if (dt >= spin_ms) {
      const timespec nap = { 0, nap_ns };
      nanosleep (&nap, NULL);
      c->stats.naps++;
      if (nap_ns < 250000)
        nap_ns *= 2;
    }
  }
}

/* everything this context has queued so far on its three queues (a failed submit may have left a copy behind that
 * no slot event covers): one fence per queue, each waited for with the deadline */
static void fence_queues (mibayer_ctx *c)
{
  if (c->wedged || !c->ev_fence)
    return;
  std::vector<hipStream_t> own = { c->s_h2d, c->s_compute, c->s_d2h };
  if (c->uses_frame_queues && c->device < 64) {
    std::lock_guard<std::mutex> lk (g_queues_mu);
    for (int k = 0; k < MIBAYER_FRAME_QUEUES; k++)
      if (c->dirty_frame[k])
        own.push_back (g_frame_queues[c->device].q[k]);
  }
  for (hipStream_t q : own) {
    if (!q)
      continue;
    if (hipEventRecord (c->ev_fence, q) != hipSuccess) {
      (void) hipGetLastError ();
      continue;
    }
    if (wait_event (c, c->ev_fence, true) == MIBAYER_ERR_TIMEOUT)
      return;
  }
}

/* the frames this context still has in flight: their own download events, nothing of the neighbours that share
 * the device's queues */
static int wait_own_frames (mibayer_ctx *c)
{
  int rc = MIBAYER_OK;
  if (!c->ring.empty ()) {
    const int n = (int) c->ring.size ();
    for (int i = 0; i < c->pending && rc == MIBAYER_OK; i++)
      rc = wait_event (c, c->ring[(size_t) ((c->tail + i) % n)].ev_out, false);
  }
  return rc;
}

/* ---- plan ----------------------------------------------------------------- */

/* Byte layout check == the reference's merge-pair selection, which only knows
 * these four (r,g,b) triples (gstbayer2rgb.c:409-421). */
static bool layout_known (int r, int g, int b)
{
  return (r == 2 && g == 1 && b == 0) || (r == 3 && g == 2 && b == 1)
      || (r == 1 && g == 2 && b == 3) || (r == 0 && g == 1 && b == 2);
}

/* rgb2bayer: which input byte each CFA site takes -- reference
 * gstrgb2bayer.c:259-266 with the hard-coded ARGB offsets (+3 blue, +1 red,
 * +2 green) generalised to (r_off, g_off, b_off). */
static void make_inverse_plan (mibayer_ctx *c)
{
  const mibayer_cfg &f = c->cfg;
  for (int par = 0; par < 2; par++) {
    int off[2];
    for (int col = 0; col < 2; col++) {
      const int site = (par << 1) | col;          /* "is_blue" in the reference */
      if (site == f.pattern)
        off[col] = f.b_off;
      else if ((site ^ 3) == f.pattern)
        off[col] = f.r_off;
      else
        off[col] = f.g_off;
    }
    /* perm (S0 = odd pixel, S1 = even pixel): byte idx 0-3 = S1, 4-7 = S0 */
    c->r2b_lo[par] = (uint32_t) off[0] | ((uint32_t) (4 + off[1]) << 8)
        | (0x0cu << 16) | (0x0cu << 24);
    c->r2b_hi[par] = 0x0cu | (0x0cu << 8) | ((uint32_t) off[0] << 16)
        | ((uint32_t) (4 + off[1]) << 24);
  }
}

static void plan_selectors (const mibayer_cfg &f, uint32_t sel[4],
    int &swap_rows)
{
  int rp = f.r_off, bp = f.b_off, gp = f.g_off;
  /* "For RGGB, we swap the red offset and blue offset in the output.  For
   * GRBG, we swap the order of the merge functions.  For GBRG, do both."
   * (gstbayer2rgb.c:396-407, :422-427) */
  if (f.pattern == MIBAYER_RGGB || f.pattern == MIBAYER_GBRG) {
    int t = rp;
    rp = bp;
    bp = t;
  }
  swap_rows = (f.pattern == MIBAYER_GRBG || f.pattern == MIBAYER_GBRG);
  const int ap = 6 - rp - gp - bp;
  /* output pixel k = v_perm_b32 (M, G, sel[k]) with
   *   M = [r' b' r' b'] of pixels (k&~1), (k|1)  -> bytes 4..7 of {M,G}
   *   G = green of pixels 0..3                   -> bytes 0..3 of {M,G}
   * selector 0x0d yields the constant 0xff (orc:65 `mergebw ra, r, 255`). */
  for (int k = 0; k < 4; k++) {
    uint32_t s = 0;
    s |= (uint32_t) (4 + 2 * (k & 1)) << (8 * rp);
    s |= (uint32_t) (5 + 2 * (k & 1)) << (8 * bp);
    s |= (uint32_t) k << (8 * gp);
    s |= (uint32_t) 0x0d << (8 * ap);
    sel[k] = s;
  }
}

static void make_plan (mibayer_ctx *c)
{
  plan_selectors (c->cfg, c->sel, c->swap_rows);
}

static bool aligned16 (const void *p)
{
  return (((uintptr_t) p) & 15u) == 0;
}

/* Launches of at most this many rounds of workgroups take the PLAN_FRAME plan (a production workgroup covers 8192
 * pixels, a device holds 4 of them per CU): on 256 CUs up to 33.5 Mpixel -- one 8K frame, four 4K frames. */
constexpr int kFrameClassRounds = 4;

static int launch_class (const mibayer_ctx *c, long long nframes)
{
  const long long px = nframes * c->cfg.width * (long long) c->cfg.height;
  return px <= (long long) kFrameClassRounds * c->num_cus * 4 * 8192 ? PLAN_FRAME : PLAN_BATCH;
}

static const Plan &plan_for (const mibayer_ctx *c, long long nframes)
{
  return c->plan[launch_class (c, nframes)];
}

static Plan &plan_for (mibayer_ctx *c, long long nframes)
{
  return c->plan[launch_class (c, nframes)];
}

/* tile grid of one launch (host side; the kernel gets the division-free TileMap) */
struct Geometry {
  int tiles_x, tiles_y, band;
  long long tile_rows;
  const Variant *var;           /* the shape the launch class of this launch runs in */
};

static void fill_params (const mibayer_ctx *c, const Plan &pl, KParams &p, Geometry &g,
    const void *d_src, size_t src_frame_bytes, void *d_dst,
    size_t dst_frame_bytes, int nframes)
{
  const mibayer_cfg &f = c->cfg;
  g.var = pl.var;
  p.src = (const uint8_t *) d_src;
  p.dst = (uint8_t *) d_dst;
  p.src_frame_bytes = src_frame_bytes;
  p.dst_frame_bytes = dst_frame_bytes;
  p.width = f.width;
  p.height = f.height;
  p.src_stride = f.src_stride;
  p.dst_stride = f.dst_stride;
  p.wlimit4 = (f.width + 3) & ~3;
  p.dn_last = f.height >= 4 ? f.height - 4 : 1;   /* ring slot reuse, :430-447 */
  g.tiles_x = (f.width + pl.var->tile_w - 1) / pl.var->tile_w;
  g.tiles_y = (f.height + pl.var->tile_h - 1) / pl.var->tile_h;
  g.tile_rows = (long long) nframes * g.tiles_y;
  int band = pl.band != INT32_MIN ? pl.band : pl.var->band;
  if (band < 0)                 /* one contiguous chunk of tile rows per XCD */
    band = (int) ((g.tile_rows + kNumXcd - 1) / kNumXcd);
  g.band = band;
  for (int k = 0; k < 4; k++)
    p.sel[k] = c->sel[k];
  p.swap_rows = c->swap_rows;
  p.start_sleep = c->start_sleep > 0 ? c->start_sleep : 0;      /* auto: plan_launch */
  p.nlist = 0;
}

typedef void (*KernelFn) (KParams);

constexpr int kStartSleepChunk = 24;    /* x s_sleep(1) = 64 clocks each */

/* everything a launch needs: arguments, kernel (16-byte fast path or generic),
 * grid size */
static int plan_launch (const mibayer_ctx *c, const void *d_src,
    size_t src_frame_bytes, void *d_dst, size_t dst_frame_bytes, int nframes,
    KParams &p, KernelFn &kern, unsigned &grid, Geometry *geom = nullptr,
    long long tile_row0 = 0, long long ntile_rows = -1, int pointers_aligned16 = -1)
{
  Geometry g;
  const Plan &pl = plan_for (c, nframes);
  fill_params (c, pl, p, g, d_src, src_frame_bytes, d_dst, dst_frame_bytes,
      nframes);
  if (ntile_rows >= 0) {        /* one horizontal band of the batch */
    if (tile_row0 < 0 || tile_row0 + ntile_rows > g.tile_rows)
      return MIBAYER_ERR_ARG;
    g.tile_rows = ntile_rows;
    if (pl.band == INT32_MIN ? pl.var->band < 0 : pl.band < 0)
      g.band = (int) ((g.tile_rows + kNumXcd - 1) / kNumXcd);
  } else {
    tile_row0 = 0;
  }
  if (g.tile_rows * g.tiles_x > 0x7fffffffLL)
    return MIBAYER_ERR_GEOMETRY;
  const mibayer_cfg &f = c->cfg;
  /* pointers_aligned16 >= 0: a list launch, whose caller has looked at every
   * frame pointer itself (frame strides do not apply): 1 = all 16-byte aligned,
   * 2 = every destination 8-byte aligned, 0 = neither */
  static const bool force_generic = LAB_GETENV ("MIBAYER_FORCE_GENERIC") != NULL;   /* A/B: tools/sweep2.py */
  const bool fast = !force_generic && (f.width % 16 == 0) && (f.src_stride % 16 == 0)
      && (f.dst_stride % 16 == 0)
      && (pointers_aligned16 >= 0 ? pointers_aligned16 == 1
          : (aligned16 (d_src) && aligned16 (d_dst)
              && (nframes == 1 || (src_frame_bytes % 16 == 0
                      && dst_frame_bytes % 16 == 0))));
  kern = fast ? pl.var->fast : pl.var->generic;
  if (!fast && pl.align && (pl.align == 128 ? pl.var->aligned128 : pl.var->aligned64)) {
    /* output rows off the sector grid, all of them 8-byte aligned (even per-row shifts): the sector-aligned arm */
    const bool rows8 = (f.dst_stride % 8 == 0)
        && (pointers_aligned16 >= 0 ? pointers_aligned16 >= 1
            : ((((uintptr_t) d_dst) & 7u) == 0 && (nframes == 1 || dst_frame_bytes % 8 == 0)));
    const unsigned a = (unsigned) pl.align;
    const bool on_grid = (f.dst_stride % a == 0)
        && (pointers_aligned16 >= 0 ? false
            : ((((uintptr_t) d_dst) & (a - 1)) == 0 && (nframes == 1 || dst_frame_bytes % a == 0)));
    static const bool force_arm = LAB_GETENV ("MIBAYER_FORCE_ALIGNED_ARM") != NULL;     /* A/B: tools/sweep2.py */
    if (rows8 && (!on_grid || force_arm))
      kern = pl.align == 128 ? pl.var->aligned128 : pl.var->aligned64;
  }
  /* The variant's default band map is dropped for the identity order in two cases
   * (profiles/r01_sweep_narrow_frames.log, r01_sweep_tile_multiple_widths.log):
   *  - one tile per row: every band map degenerates to the identity order, and the
   *    start delay that comes with a band map only costs (1024-px rows: 72 vs 84 %);
   *  - the width is a whole number of tiles: a tile row is then a multiple of 32 KiB
   *    of output, the eight XCDs of a band map write at power-of-two distances from
   *    each other and collide in the memory channels (2048 / 3072 / 4096 / 5120 px:
   *    70-77 % with band 1, 81-83 % in identity order; 1920 / 3840 / 7680 px, whose
   *    last tile is partial, are the other way round by 2-4 points). */
  if (g.band > 0 && pl.band == INT32_MIN
      && (g.tiles_x == 1 || f.width % pl.var->tile_w == 0))
    g.band = 0;
  if (fast && pl.var->persistent) {
    /* persistent arm: only "one chunk per XCD" or "identity" make sense, and
     * the grid is a fixed number of workgroups per CU */
    if (g.band > 0)
      g.band = (int) ((g.tile_rows + kNumXcd - 1) / kNumXcd);
    const long long ntiles = g.tile_rows * g.tiles_x;
    long long n = (long long) c->num_cus * c->persist_wgs_per_cu;
    n -= n % kNumXcd;
    if (n > ntiles)
      n = (ntiles + kNumXcd - 1) / kNumXcd * kNumXcd;
    grid = (unsigned) (n > 0 ? n : kNumXcd);
  } else {
    const long long n = grid_blocks_for (g.tiles_x, g.tile_rows, g.band);
    if (n > 0x7fffffffLL)
      return MIBAYER_ERR_GEOMETRY;
    grid = (unsigned) n;
  }
  p.map = make_tile_map (g.tiles_x, g.tiles_y, g.tile_rows, g.band, tile_row0);
  /* Start delay (DESIGN.md "start delay"): with a band map every workgroup
   * sleeps ~1.5k cycles before its first load.  Measured +3..5 points of HBM
   * peak on every box for the chunk-per-XCD order (the previous workgroup's
   * stores drain before the new loads reach the L2, profiles/r01_delay_counters.md;
   * halving the occupancy instead costs 10 points), nothing
   * for the identity order, and pure latency for launches that do not even fill
   * the machine once -- so only grids of more than 4 workgroups per CU slot get it. */
  if (c->start_sleep < 0)
    p.start_sleep = (g.band > 0 && !pl.var->persistent
        && (long long) grid > 16LL * c->num_cus) ? kStartSleepChunk : 0;
  if (geom)
    *geom = g;
  return MIBAYER_OK;
}

static int launch (const mibayer_ctx *c, const void *d_src,
    size_t src_frame_bytes, void *d_dst, size_t dst_frame_bytes, int nframes,
    hipStream_t stream, long long tile_row0 = 0, long long ntile_rows = -1)
{
  if (nframes == 0 || ntile_rows == 0)
    return MIBAYER_OK;
  if (c->inverse) {
    const mibayer_cfg &f = c->cfg;
    R2BParams q;
    q.src = (const uint8_t *) d_src;
    q.dst = (uint8_t *) d_dst;
    q.src_frame_bytes = src_frame_bytes;
    q.dst_frame_bytes = dst_frame_bytes;
    q.width = f.width;
    q.height = f.height;
    q.src_stride = f.src_stride;
    q.dst_stride = f.dst_stride;
    q.out_dwords = ((f.width + 3) & ~3) / 4;
    q.total_rows = (long long) nframes * f.height;
    /* block order: identity for the flat kernel (82.5 % of peak against 77.5 % with one chunk of
     * the batch per XCD), one chunk per XCD for the tile kernel (74.4 vs 73.8 %) --
     * profiles/r02_rgb2bayer_sweep.log */
    q.band = c->plan[PLAN_BATCH].band != INT32_MIN ? c->plan[PLAN_BATCH].band
        : (c->r2b_flat_k > 0 ? 0 : -1);
    q.start_sleep = c->start_sleep > 0 ? c->start_sleep : 0;
    q.flat_k = c->r2b_flat_k;
    q.flat_px = c->r2b_flat_px;
    q.flat_ld = c->r2b_flat_ld;
    q.rows = c->r2b_rows;
    q.nlist = 0;
    for (int k = 0; k < 2; k++) {
      q.sel_lo[k] = c->r2b_lo[k];
      q.sel_hi[k] = c->r2b_hi[k];
    }
    const bool vec16 = (f.width % 4 == 0) && (f.src_stride % 16 == 0)
        && aligned16 (d_src) && (nframes == 1 || src_frame_bytes % 16 == 0);
    if (ntile_rows >= 0) {      /* band of 16-row units (host path) */
      const long long y0 = tile_row0 * kInverseBandUnit;
      long long y1 = (tile_row0 + ntile_rows) * kInverseBandUnit;
      if (y1 > q.total_rows)
        y1 = q.total_rows;
      HIP_TRY (launch_rgb2bayer (q, vec16, stream, y0, y1 - y0));
    } else {
      HIP_TRY (launch_rgb2bayer (q, vec16, stream));
    }
    return MIBAYER_OK;
  }
  KParams p;
  KernelFn kern;
  unsigned grid;
  Geometry g;
  int rc = plan_launch (c, d_src, src_frame_bytes, d_dst, dst_frame_bytes,
      nframes, p, kern, grid, &g, tile_row0, ntile_rows);
  if (rc != MIBAYER_OK)
    return rc;
  /* (lab builds) dynamic LDS per workgroup, only to cap the workgroups per CU in occupancy experiments */
  static const unsigned dyn_lds = [] {
    const char *e = LAB_GETENV ("MIBAYER_DYN_LDS");
    return e ? (unsigned) atoi (e) : 0u;
  } ();
  hipLaunchKernelGGL (kern, dim3 (grid), dim3 (g.var->threads), dyn_lds, stream, p);
  HIP_TRY (hipGetLastError ());
  return MIBAYER_OK;
}

/* ---- global ------------------------------------------------------------------ */

extern "C" int mibayer_abi_version (void)
{
  return MIBAYER_ABI_VERSION;
}

extern "C" int mibayer_is_lab_build (void)
{
#ifdef MIBAYER_LAB
  return 1;
#else
  return 0;
#endif
}

extern "C" int mibayer_device_count (void)
{
  return device_count_cached ();
}

/* "0000:c1:00.0" of a HIP ordinal: which physical card a rank / shard / element runs on (bench.py's per_gpu identity,
 * the elements' start-up log) */
extern "C" int mibayer_device_pci_bus_id (int device, char *out, size_t len)
{
  if (!out || len < 13)
    return MIBAYER_ERR_ARG;
  out[0] = 0;
  if (device < 0 || device >= device_count_cached ())
    return MIBAYER_ERR_NO_DEVICE;
  if (hip_failed (hipDeviceGetPCIBusId (out, (int) len, device), "hipDeviceGetPCIBusId"))
    return MIBAYER_ERR_HIP;
  return MIBAYER_OK;
}

extern "C" const char *mibayer_strerror (int status)
{
  switch (status) {
    case MIBAYER_OK: return "ok";
    case MIBAYER_ERR_ARG: return "invalid argument";
    case MIBAYER_ERR_GEOMETRY:
      return "unsupported geometry (need even width >= 4, height >= 3, "
          "strides multiple of 4 and large enough)";
    case MIBAYER_ERR_LAYOUT:
      return "unsupported (r,g,b) byte offsets (need RGBx/BGRx/xRGB/xBGR order)";
    case MIBAYER_ERR_NO_DEVICE: return "no usable HIP device";
    case MIBAYER_ERR_HIP: return "HIP runtime error";
    case MIBAYER_ERR_NOMEM: return "out of memory";
    case MIBAYER_ERR_BUSY: return "frames in flight: ring full or busy";
    case MIBAYER_ERR_EMPTY: return "no frame in flight";
    case MIBAYER_ERR_TIMEOUT: return "the device did not answer within the wait deadline";
    default: return "unknown mibayer status";
  }
}

extern "C" const char *mibayer_last_hip_error (void)
{
  return t_hip_error;
}

extern "C" int mibayer_variant_count (void)
{
  return variant_count ();
}

extern "C" const char *mibayer_variant_name (int v)
{
  if (v < 0 || v >= variant_count ())
    return NULL;
  return variant (v).name;
}

extern "C" int64_t mibayer_block_to_tile (int64_t block, int tiles_x,
    int64_t tile_rows, int band)
{
  if (block < 0 || block > 0x7fffffffLL || tiles_x <= 0 || tile_rows <= 0
      || tile_rows * tiles_x > 0x7fffffffLL)
    return -1;
  /* the very function the kernels run */
  const TileMap m = make_tile_map (tiles_x, 1, tile_rows, band);
  const TileId t = block_to_tile ((uint32_t) block, m);
  if (band <= 0 && block >= tile_rows * tiles_x)
    return -1;
  return t.valid ? (int64_t) t.row * tiles_x + t.tx : -1;
}

/* ---- context -------------------------------------------------------------------- */

static int validate (const mibayer_cfg *in, mibayer_cfg *out)
{
  if (!in || in->struct_size != sizeof (mibayer_cfg))
    return MIBAYER_ERR_ARG;
  mibayer_cfg f = *in;
  if (f.pattern < MIBAYER_BGGR || f.pattern > MIBAYER_RGGB)
    return MIBAYER_ERR_ARG;
  if (f.flags & ~(uint32_t) (MIBAYER_FLAG_HIPGRAPH | MIBAYER_FLAG_RGB2BAYER | MIBAYER_FLAG_HIPGRAPH_CHAIN))
    return MIBAYER_ERR_ARG;
  if ((f.flags & MIBAYER_FLAG_HIPGRAPH_CHAIN) && !(f.flags & MIBAYER_FLAG_HIPGRAPH))
    return MIBAYER_ERR_ARG;
  if (f.flags & MIBAYER_FLAG_RGB2BAYER) {
    /* inverse direction: src = 4 B/pixel, dst = mosaic.  The reference loop
     * (gstrgb2bayer.c:254-268) has no neighbourhood, so any size >= 1 is valid */
    if (f.width < 1 || f.height < 1 || f.width > (1 << 28)
        || f.height > (1 << 28))
      return MIBAYER_ERR_GEOMETRY;
    if (f.variant != 0 || f.inflight < 0 || f.inflight > 64)
      return MIBAYER_ERR_ARG;
    if (f.src_stride == 0)
      f.src_stride = 4 * f.width;
    if (f.dst_stride == 0)
      f.dst_stride = (f.width + 3) & ~3;          /* gstrgb2bayer.c:179, :255 */
    if (f.src_stride < 4 * f.width || (f.src_stride & 3))
      return MIBAYER_ERR_GEOMETRY;
    if (f.dst_stride < ((f.width + 3) & ~3) || (f.dst_stride & 3))
      return MIBAYER_ERR_GEOMETRY;
    for (int v : { f.r_off, f.g_off, f.b_off })
      if (v < 0 || v > 3)
        return MIBAYER_ERR_LAYOUT;
    if (f.r_off == f.g_off || f.g_off == f.b_off || f.r_off == f.b_off)
      return MIBAYER_ERR_LAYOUT;
    if (f.inflight == 0)
      f.inflight = 2;
    *out = f;
    return MIBAYER_OK;
  }
  if (f.variant < 0 || f.variant >= variant_count ())
    return MIBAYER_ERR_ARG;
  if (f.inflight < 0 || f.inflight > 64)
    return MIBAYER_ERR_ARG;
  if (f.width < 4 || (f.width & 1) || f.height < 3)
    return MIBAYER_ERR_GEOMETRY;
  if (f.width > (1 << 28) || f.height > (1 << 28))
    return MIBAYER_ERR_GEOMETRY;
  if (f.src_stride == 0)
    f.src_stride = (f.width + 3) & ~3;  /* GST_ROUND_UP_4, gstbayer2rgb.c:477 */
  if (f.dst_stride == 0)
    f.dst_stride = 4 * f.width;         /* gstbayer2rgb.c:344 */
  if (f.src_stride < ((f.width + 3) & ~3) || (f.src_stride & 3))
    return MIBAYER_ERR_GEOMETRY;
  if (f.dst_stride < 4 * f.width || (f.dst_stride & 3))
    return MIBAYER_ERR_GEOMETRY;
  if (!layout_known (f.r_off, f.g_off, f.b_off))
    return MIBAYER_ERR_LAYOUT;
  if (f.inflight == 0)
    f.inflight = 2;
  *out = f;
  return MIBAYER_OK;
}

static int choose_host_bands (const mibayer_ctx *c);
static bool plan_cache_load (mibayer_ctx *c);
extern "C" void mibayer_internal_private_queues (mibayer_ctx *c);

extern "C" int mibayer_create (const mibayer_cfg *cfg, mibayer_ctx **out)
{
  if (!out)
    return MIBAYER_ERR_ARG;
  *out = NULL;
  mibayer_cfg f;
  int rc = validate (cfg, &f);
  if (rc != MIBAYER_OK)
    return rc;
  const int ndev = device_count_cached ();
  if (ndev <= 0)
    return MIBAYER_ERR_NO_DEVICE;
  int dev = f.device;
  if (dev == -1) {
    if (hip_failed (hipGetDevice (&dev), "hipGetDevice"))
      return MIBAYER_ERR_HIP;
  }
  if (dev < 0 || dev >= ndev)
    return MIBAYER_ERR_NO_DEVICE;
  f.device = dev;

  poll_wedges ();
  mibayer_ctx *c = new (std::nothrow) mibayer_ctx ();
  if (!c)
    return MIBAYER_ERR_NOMEM;
  c->cfg = f;
  c->device = dev;
  c->src_bytes = (size_t) f.src_stride * f.height;
  c->dst_bytes = (size_t) f.dst_stride * f.height;
  c->inverse = (f.flags & MIBAYER_FLAG_RGB2BAYER) != 0;
  {
    int cus = 0;
    if (hipDeviceGetAttribute (&cus, hipDeviceAttributeMultiprocessorCount,
            dev) == hipSuccess && cus > 0)
      c->num_cus = cus;
  }
  Plan &pb = c->plan[PLAN_BATCH];
  pb.var = &variant (resolve_variant (f.variant, f.width));
  /* Output rows off the 64-byte sector grid (generic geometries; profiles/r03_generic_path.log).  Rows that fit one
   * tile keep their identity-order plan.  Wider ones:
   *  - rows 16-byte aligned (width % 4 == 0, e.g. 4056 px): every lane's 16-byte store is aligned, only the two
   *    ends of a wave-store share a line with a neighbour -> hybrid store policy (nt inside, write-back at the
   *    ragged ends; store_pixels_hybrid) in the allocation-independent band-1 order;
   *  - rows at an 8-byte phase (width % 4 == 2): every lane's store straddles a 16-byte boundary -> write-back
   *    stores + one chunk of the batch per XCD, where the L2 puts the pieces together (plain_store_twin).
   * mibayer_autotune() times all store policies, and the shifted arm, against each other. */
  c->rows_off_sector = !c->inverse && (f.dst_stride % 64) != 0;
  if (c->rows_off_sector && f.variant == 0 && f.width > pb.var->tile_w) {
    if (f.dst_stride % 16 == 0) {
      pb.var = &variant (hybrid_store_twin (resolve_variant (0, f.width)));
      pb.band = 1;
    } else {
      pb.var = &variant (plain_store_twin (resolve_variant (0, f.width)));
      pb.band = -1;
    }
  }
  if (f.variant == 0 && !c->inverse && !c->rows_off_sector) {
    int kv = 0, kb = 0;
    if (known_width_plan (f.width, &kv, &kb)) {         /* common sensor widths with a measured winner */
      pb.var = &variant (kv);
      pb.band = kb;
    }
  }
  /* One frame per launch (PLAN_FRAME): the same plan, except that for sector-aligned geometries "auto" takes the
   * shape whose grid needs the fewest rounds of workgroups (frame_class_variant) */
  c->plan[PLAN_FRAME] = pb;
  if (f.variant == 0 && !c->inverse && !c->rows_off_sector)
    c->plan[PLAN_FRAME] = Plan { &variant (frame_class_variant (f.width, f.height, c->num_cus * 4)), INT32_MIN, 0,
      MIBAYER_PLAN_DEFAULT };
  /* ... and for wide rows at an 8-byte phase (width % 4 == 2: 3838, 2046, 1366 px) the batch class's answer -- write-back
   * stores, one chunk of tile rows per XCD, where the L2 puts the pieces of the straddling stores together -- is the
   * wrong one for a launch of one or two rounds of workgroups: nothing stays in an L2 long enough to be merged, and a
   * chunk per XCD of a 270-tile-row grid leaves XCDs unevenly loaded.  Round 6 sweep (tools/csrc/frame_plan_sweep.c,
   * profiles/r06_single_frame.md): the streaming-store shape with the fewest rounds in its default block order, and
   * for the 256- and 512-px-tile shapes the store arm whose wave-stores start on 128-byte boundaries -- 3838x2160
   * 39.2 -> 48.4 % of HBM peak per frame, 7678x4320 58.6 -> 62.1 % with the arm, 2046x1080 22.3 -> 25.1 %, 1366x768
   * 13.2 -> 14.5 % (the last two are launch-issue-bound; 1024-px tiles gain nothing from the arm).
   * Rows that are 16-byte aligned keep the hybrid-store plan (4056x3040: 57.2 % against 53.0 % for streaming stores). */
  if (f.variant == 0 && !c->inverse && c->rows_off_sector && f.dst_stride % 16 != 0 && f.dst_stride % 8 == 0
      && f.width > pb.var->tile_w) {
    const int shape = frame_class_variant (f.width, f.height, c->num_cus * 4);
    c->plan[PLAN_FRAME] = Plan { &variant (shape), INT32_MIN, shape >= 2 ? 128 : 0, MIBAYER_PLAN_DEFAULT };
  }
  /* a plan measured earlier in this process for this geometry and launch class on this device (mibayer_autotune)
   * replaces the default */
  (void) plan_cache_load (c);
  if (const char *e = LAB_GETENV ("MIBAYER_XCD_BAND")) {
    for (Plan &pl : c->plan)
      pl.band = atoi (e);
    c->band_forced = true;
  }
  if (const char *e = LAB_GETENV ("MIBAYER_START_SLEEP"))
    c->start_sleep = atoi (e) >= 0 ? atoi (e) : -1;
  if (const char *e = LAB_GETENV ("MIBAYER_ALIGN_STORES")) {
    const int a = atoi (e);
    for (Plan &pl : c->plan)
      pl.align = (a == 64 || a == 128) ? a : 0;
    c->align_tunable = false;
  }
  if (const char *e = LAB_GETENV ("MIBAYER_R2B_FLAT"))
    c->r2b_flat_k = atoi (e);
  if (const char *e = LAB_GETENV ("MIBAYER_R2B_PX"))
    c->r2b_flat_px = atoi (e);
  if (const char *e = LAB_GETENV ("MIBAYER_R2B_LDNT"))
    c->r2b_flat_ld = atoi (e);
  if (const char *e = LAB_GETENV ("MIBAYER_R2B_ROWS"))
    c->r2b_rows = atoi (e);
  if (const char *e = getenv ("MIBAYER_WAIT_TIMEOUT_MS"))
    c->wait_timeout_ms = atoi (e) > 0 ? atoi (e) : 0;
  if (const char *e = getenv ("MIBAYER_WAIT_SPIN_US"))
    c->wait_spin_us = atoi (e) >= 0 ? atoi (e) : -1;
  c->graph_mode = (f.flags & MIBAYER_FLAG_HIPGRAPH_CHAIN) ? 1 : 0;
  if (const char *e = LAB_GETENV ("MIBAYER_GRAPH_MODE"))
    c->graph_mode = (strcmp (e, "chain") == 0 || strcmp (e, "1") == 0) ? 1 : 0;
  if (const char *e = LAB_GETENV ("MIBAYER_PERSIST_WGS"))
    c->persist_wgs_per_cu = atoi (e) > 0 ? atoi (e) : 4;
  if (c->inverse)
    make_inverse_plan (c);
  else
    make_plan (c);
  c->host_bands = choose_host_bands (c);

  DeviceGuard guard (dev);
  if (!guard.ok) {
    delete c;
    return MIBAYER_ERR_HIP;
  }
  if (dev < 64) {
    std::lock_guard<std::mutex> lk (g_queues_mu);
    g_cache[dev].contexts++;
    c->counted = true;
  }
  bool bad = false;
  if (shared_queues_enabled () && dev < 64) {
    std::lock_guard<std::mutex> lk (g_queues_mu);
    DeviceQueues &q = g_queues[dev];
    if (q.refs == 0) {
      bad |= hip_failed (hipStreamCreateWithFlags (&q.h2d, hipStreamNonBlocking),
          "hipStreamCreate");
      bad |= hip_failed (hipStreamCreateWithFlags (&q.compute,
              hipStreamNonBlocking), "hipStreamCreate");
      bad |= hip_failed (hipStreamCreateWithFlags (&q.d2h, hipStreamNonBlocking),
          "hipStreamCreate");
      if (bad) {
        for (hipStream_t *st : { &q.h2d, &q.compute, &q.d2h }) {
          if (*st)
            (void) hipStreamDestroy (*st);
          *st = nullptr;
        }
      }
    }
    if (!bad) {
      q.refs++;
      c->s_h2d = q.h2d;
      c->s_compute = q.compute;
      c->s_d2h = q.d2h;
      c->shared_queues = true;
    }
  } else {
    bad |= hip_failed (hipStreamCreateWithFlags (&c->s_h2d,
            hipStreamNonBlocking), "hipStreamCreate");
    bad |= hip_failed (hipStreamCreateWithFlags (&c->s_compute,
            hipStreamNonBlocking), "hipStreamCreate");
    bad |= hip_failed (hipStreamCreateWithFlags (&c->s_d2h,
            hipStreamNonBlocking), "hipStreamCreate");
  }
  bad |= hip_failed (hipEventCreate (&c->ev_t0), "hipEventCreate");
  bad |= hip_failed (hipEventCreate (&c->ev_t1), "hipEventCreate");
  bad |= hip_failed (hipEventCreateWithFlags (&c->ev_fence, hipEventDisableTiming), "hipEventCreate");
  if (bad) {
    mibayer_destroy (c);
    return MIBAYER_ERR_HIP;
  }
  /* the device-side frame ring of the host path is allocated on first use, so
   * that device-resident callers do not pay for it */
  *out = c;
  return MIBAYER_OK;
}

/* how a slot's resources go: into the per-device frame cache (everything queued on them is known to have completed),
 * straight back to the runtime (hipFree waits for the device: used when a wait of the context's own frames ended in
 * a device error, so that nothing in flight can outlive the buffers), or to the wedge registry (the device has not
 * answered: nothing may block behind it) */
enum SlotRelease { RELEASE_CACHE, RELEASE_FREE, RELEASE_ORPHAN };

static void free_slot (mibayer_ctx *c, Slot &s, SlotRelease how = RELEASE_CACHE)
{
  if (how == RELEASE_ORPHAN) {
    Wedge *w = c->wedge;
    for (void *m : { (void *) s.d_src, (void *) s.d_dst })
      if (m)
        w->dev_mem.push_back (m);
    for (hipEvent_t ev : { s.ev_in, s.ev_kernel, s.ev_out })
      if (ev)
        w->events.push_back (ev);
    for (int b = 0; b < kMaxHostBands; b++) {
      if (s.ev_band_in[b])
        w->events.push_back (s.ev_band_in[b]);
      if (s.ev_band_kernel[b])
        w->events.push_back (s.ev_band_kernel[b]);
    }
    for (hipGraphExec_t e : { s.cexec, s.exec })
      if (e)
        w->execs.push_back (e);
    for (hipGraph_t g : { s.cgraph, s.graph })
      if (g)
        w->graphs.push_back (g);
    if (s.s_graph)
      w->streams.push_back (s.s_graph);
    s = Slot ();
    return;
  }
  if (how == RELEASE_FREE) {
    if (s.d_src)
      (void) hipFree (s.d_src);
    if (s.d_dst)
      (void) hipFree (s.d_dst);
  } else {
    cached_free (c->device, s.d_src, c->src_bytes);
    cached_free (c->device, s.d_dst, c->dst_bytes);
  }
  if (s.ev_in)
    (void) hipEventDestroy (s.ev_in);
  if (s.ev_kernel)
    (void) hipEventDestroy (s.ev_kernel);
  if (s.ev_out)
    (void) hipEventDestroy (s.ev_out);
  for (int b = 0; b < kMaxHostBands; b++) {
    if (s.ev_band_in[b])
      (void) hipEventDestroy (s.ev_band_in[b]);
    if (s.ev_band_kernel[b])
      (void) hipEventDestroy (s.ev_band_kernel[b]);
  }
  if (s.cexec)
    (void) hipGraphExecDestroy (s.cexec);
  if (s.cgraph)
    (void) hipGraphDestroy (s.cgraph);
  if (s.exec)
    (void) hipGraphExecDestroy (s.exec);
  if (s.graph)
    (void) hipGraphDestroy (s.graph);
  if (s.s_graph)
    (void) hipStreamDestroy (s.s_graph);
  s = Slot ();
}

static void free_ring (mibayer_ctx *c, SlotRelease how = RELEASE_CACHE)
{
  for (Slot &s : c->ring)
    free_slot (c, s, how);
  c->ring.clear ();
}

/* Waits for the context's OWN frames (their download events), never for the queues it may share with the other
 * contexts of the device, and hands its device frames to the per-device cache instead of hipFree (which drains the
 * whole device).  If that wait ends in a device error the frames go straight back to the runtime instead (nothing in
 * flight may outlive them in the cache).  A context that ran into a wait deadline and whose device has not caught up
 * since waits for nothing and releases nothing itself: its device-side resources go to the wedge registry. */
extern "C" void mibayer_destroy (mibayer_ctx *c)
{
  if (!c)
    return;
  DeviceGuard guard (c->device);
  const int wrc = wait_own_frames (c);          /* returns at once for a wedged context; may be what finds it wedged */
  SlotRelease how = RELEASE_CACHE;
  if (c->wedged) {
    how = ctx_settled (c) ? RELEASE_FREE : RELEASE_ORPHAN;
    if (how == RELEASE_ORPHAN && !c->wedge)     /* no registry entry could be made: leave everything alone */
      how = RELEASE_CACHE;
  } else if (wrc != MIBAYER_OK) {
    fence_queues (c);
    how = c->wedged ? (c->wedge ? RELEASE_ORPHAN : RELEASE_CACHE) : RELEASE_FREE;
  }
  const bool leak = c->wedged && how != RELEASE_FREE;          /* nothing of the runtime is called on its behalf */
  if (!(leak && how == RELEASE_CACHE)) {
    free_ring (c, how);
    free_slot (c, c->spare, how);
  }
  for (hipEvent_t ev : { c->ev_t0, c->ev_t1, c->ev_fence }) {
    if (!ev)
      continue;
    if (how == RELEASE_ORPHAN)
      c->wedge->events.push_back (ev);
    else if (!leak)
      (void) hipEventDestroy (ev);
  }
  std::vector<std::pair<void *, size_t>> trim;
  {
    std::lock_guard<std::mutex> lk (g_queues_mu);
    if (c->shared_queues) {
      DeviceQueues &q = g_queues[c->device];
      if (--q.refs == 0) {
        for (hipStream_t st : { q.h2d, q.compute, q.d2h }) {
          if (how == RELEASE_ORPHAN)
            c->wedge->streams.push_back (st);
          else if (!leak)
            (void) hipStreamDestroy (st);
        }
        q = DeviceQueues ();
      }
    }
    if (c->device < 64 && c->counted && --g_cache[c->device].contexts == 0) {
      /* the last context of the device.  (Its frame queues stay: see FrameQueues.) */
      if (!leak) {
        trim.swap (g_cache[c->device].bufs);
        g_cache[c->device].bytes = 0;
      }
    }
  }
  if (!c->shared_queues) {
    for (hipStream_t st : { c->s_h2d, c->s_compute, c->s_d2h }) {
      if (!st)
        continue;
      if (how == RELEASE_ORPHAN)
        c->wedge->streams.push_back (st);
      else if (!leak)
        (void) hipStreamDestroy (st);
    }
  }
  for (auto &b : trim)          /* the last context of the device: nothing of ours is queued there any more */
    (void) hipFree (b.first);
  if (c->wedge) {
    Wedge *done = nullptr;
    {
      std::lock_guard<std::mutex> lk (g_wedge_mu);
      poll_wedges_locked ();
      if (c->wedge->settled) {  /* (possibly since the resources above were handed over: they are released here) */
        for (size_t i = 0; i < g_wedges.size (); i++)
          if (g_wedges[i] == c->wedge)
            g_wedges.erase (g_wedges.begin () + (long) i);
        done = c->wedge;
      } else {
        c->wedge->orphan = true;        /* the registry releases what was handed over once the fences fire */
      }
    }
    if (done) {
      release_wedge_resources (done);
      delete done;
    }
  }
  delete c;
  poll_wedges ();
}

extern "C" int mibayer_plan_selectors (const mibayer_cfg *cfg, uint32_t sel[4],
    int *swap_rows)
{
  if (!sel || !swap_rows)
    return MIBAYER_ERR_ARG;
  mibayer_cfg f;
  int rc = validate (cfg, &f);
  if (rc != MIBAYER_OK)
    return rc;
  if (f.flags & MIBAYER_FLAG_RGB2BAYER)
    return MIBAYER_ERR_ARG;
  plan_selectors (f, sel, *swap_rows);
  return MIBAYER_OK;
}

extern "C" int mibayer_auto_variant (int width)
{
  return resolve_variant (0, width);
}

extern "C" int mibayer_frame_class_variant (int width, int height, int compute_units)
{
  if (width < 1 || height < 1 || compute_units < 1)
    return MIBAYER_ERR_ARG;
  return frame_class_variant (width, height, compute_units * 4);
}

extern "C" int mibayer_known_width_plan (int width, int *variant, int *band)
{
  int v = 0, b = 0;
  if (!known_width_plan (width, &v, &b))
    return 0;
  if (variant)
    *variant = v;
  if (band)
    *band = b;
  return 1;
}

extern "C" const char *mibayer_ctx_variant_name (const mibayer_ctx *c)
{
  return c ? c->plan[PLAN_BATCH].var->name : NULL;
}

extern "C" int mibayer_get_cfg (const mibayer_ctx *c, mibayer_cfg *out)
{
  if (!c || !out)
    return MIBAYER_ERR_ARG;
  *out = c->cfg;
  return MIBAYER_OK;
}

extern "C" int mibayer_launch_geometry (const mibayer_ctx *c, int nframes,
    int *tile_w, int *tile_h, int *tiles_x, int64_t *tile_rows, int *band,
    int64_t *grid_blocks)
{
  if (!c || nframes < 0)
    return MIBAYER_ERR_ARG;
  KParams p;
  KernelFn kern;
  unsigned grid = 0;
  Geometry g;
  /* aligned dummy pointers: report the plan of the 16-byte fast path when the
   * geometry allows it */
  int rc = plan_launch (c, (const void *) 256, c->src_bytes, (void *) 256,
      c->dst_bytes, nframes > 0 ? nframes : 1, p, kern, grid, &g);
  if (rc != MIBAYER_OK)
    return rc;
  if (nframes == 0)
    grid = 0;
  if (tile_w)
    *tile_w = g.var->tile_w;
  if (tile_h)
    *tile_h = g.var->tile_h;
  if (tiles_x)
    *tiles_x = g.tiles_x;
  if (tile_rows)
    *tile_rows = nframes > 0 ? g.tile_rows : 0;
  if (band)
    *band = g.band;
  if (grid_blocks)
    *grid_blocks = grid;
  return MIBAYER_OK;
}

/* ---- host-memory frame path -------------------------------------------------------- */

/* device-side frames and events of one slot; `bands`: per-band events too */
static int alloc_slot (mibayer_ctx *c, Slot &s, bool bands)
{
  bool bad = false;
  const hipError_t e_src = cached_malloc (c->device, (void **) &s.d_src, c->src_bytes);
  const hipError_t e_dst = e_src == hipSuccess
      ? cached_malloc (c->device, (void **) &s.d_dst, c->dst_bytes) : e_src;
  if (e_src == hipErrorOutOfMemory || e_dst == hipErrorOutOfMemory) {
    (void) hip_failed (hipErrorOutOfMemory, "hipMalloc (frame ring)");
    (void) hipGetLastError ();
    return MIBAYER_ERR_NOMEM;
  }
  bad |= hip_failed (e_src, "hipMalloc");
  bad |= hip_failed (e_dst, "hipMalloc");
  bad |= hip_failed (hipEventCreateWithFlags (&s.ev_in,
          hipEventDisableTiming), "hipEventCreate");
  bad |= hip_failed (hipEventCreateWithFlags (&s.ev_kernel,
          hipEventDisableTiming), "hipEventCreate");
  bad |= hip_failed (hipEventCreateWithFlags (&s.ev_out,
          hipEventDisableTiming), "hipEventCreate");
  for (int b = 0; bands && b < c->host_bands && c->host_bands > 1; b++) {
    bad |= hip_failed (hipEventCreateWithFlags (&s.ev_band_in[b],
            hipEventDisableTiming), "hipEventCreate");
    bad |= hip_failed (hipEventCreateWithFlags (&s.ev_band_kernel[b],
            hipEventDisableTiming), "hipEventCreate");
  }
  return bad ? MIBAYER_ERR_HIP : MIBAYER_OK;
}

static int ensure_ring (mibayer_ctx *c)
{
  if (!c->ring.empty ())
    return MIBAYER_OK;
  c->ring.resize ((size_t) c->cfg.inflight);
  for (Slot &s : c->ring) {
    const int rc = alloc_slot (c, s, true);
    if (rc != MIBAYER_OK) {
      free_ring (c);
      return rc;
    }
  }
  c->head = c->tail = c->pending = 0;
  return MIBAYER_OK;
}

/* Graph form of one frame: H2D copy -> kernel -> D2H copy as three explicit
 * nodes, instantiated once per ring slot.  Only the two host pointers change
 * from frame to frame; they are patched into the instantiated graph
 * (hipGraphExecMemcpyNodeSetParams1D), or the graph is re-instantiated if the
 * runtime refuses the update. */
static int graph_submit (mibayer_ctx *c, Slot &s, const uint8_t *src,
    uint8_t *dst)
{
  if (!s.s_graph)
    HIP_TRY (hipStreamCreateWithFlags (&s.s_graph, hipStreamNonBlocking));
  if (s.exec && s.plan_epoch != c->plan_epoch) {
    /* the plan has changed since the kernel node was built (the slot's previous frame has completed: it is free) */
    (void) hipGraphExecDestroy (s.exec);
    (void) hipGraphDestroy (s.graph);
    s.exec = nullptr;
    s.graph = nullptr;
  }
  if (s.exec && (s.g_src != src || s.g_dst != dst)) {
    hipError_t e1 = hipGraphExecMemcpyNodeSetParams1D (s.exec, s.n_h2d,
        s.d_src, src, c->src_bytes, hipMemcpyHostToDevice);
    hipError_t e2 = hipGraphExecMemcpyNodeSetParams1D (s.exec, s.n_d2h, dst,
        s.d_dst, c->dst_bytes, hipMemcpyDeviceToHost);
    if (e1 != hipSuccess || e2 != hipSuccess) {
      (void) hipGetLastError ();
      (void) hipGraphExecDestroy (s.exec);
      (void) hipGraphDestroy (s.graph);
      s.exec = nullptr;
      s.graph = nullptr;
    }
  }
  if (!s.exec) {
    KParams p;
    KernelFn kern;
    unsigned grid;
    Geometry g;
    int rc = plan_launch (c, s.d_src, c->src_bytes, s.d_dst, c->dst_bytes, 1,
        p, kern, grid, &g);
    if (rc != MIBAYER_OK)
      return rc;
    void *args[1] = { &p };
    hipKernelNodeParams kp;
    memset (&kp, 0, sizeof kp);
    kp.func = (void *) kern;
    kp.gridDim = dim3 (grid);
    kp.blockDim = dim3 ((unsigned) g.var->threads);
    kp.sharedMemBytes = 0;
    kp.kernelParams = args;
    kp.extra = NULL;
    bool bad = hip_failed (hipGraphCreate (&s.graph, 0), "hipGraphCreate");
    bad = bad || hip_failed (hipGraphAddMemcpyNode1D (&s.n_h2d, s.graph, NULL,
            0, s.d_src, src, c->src_bytes, hipMemcpyHostToDevice),
        "hipGraphAddMemcpyNode1D");
    bad = bad || hip_failed (hipGraphAddKernelNode (&s.n_kernel, s.graph,
            &s.n_h2d, 1, &kp), "hipGraphAddKernelNode");
    bad = bad || hip_failed (hipGraphAddMemcpyNode1D (&s.n_d2h, s.graph,
            &s.n_kernel, 1, dst, s.d_dst, c->dst_bytes, hipMemcpyDeviceToHost),
        "hipGraphAddMemcpyNode1D");
    bad = bad || hip_failed (hipGraphInstantiate (&s.exec, s.graph, NULL, NULL,
            0), "hipGraphInstantiate");
    if (bad) {                  /* no half-built graph survives a failure */
      if (s.graph)
        (void) hipGraphDestroy (s.graph);
      s.graph = nullptr;
      s.exec = nullptr;
      return MIBAYER_ERR_HIP;
    }
    s.plan_epoch = c->plan_epoch;
  }
  s.g_src = src;
  s.g_dst = dst;
  HIP_TRY (hipGraphLaunch (s.exec, s.s_graph));
  HIP_TRY (hipEventRecord (s.ev_out, s.s_graph));
  return MIBAYER_OK;
}

/* Host path, one frame cut into horizontal bands: band b's kernel needs source
 * rows y0-1 .. y1 (one halo row above and below), so upload chunk b carries the
 * band's rows plus the row below it, and the chunks run down the frame in one
 * queue.  With the three queues of the ring, the download of band b-1 (4 B/px,
 * the long pole: PCIe is full duplex) overlaps the upload of band b+1 and the
 * kernel of band b inside ONE frame -- the synchronous 1-in/1-out element mode
 * gets most of what the queued mode gets from overlapping whole frames.
 * Bands are whole tile rows; the bottom-edge rule dn(H-1) = H-4
 * (gstbayer2rgb.c:430-447) needs the last band to hold at least 4 rows. */
static int choose_host_bands (const mibayer_ctx *c)
{
  int want = 4;
  if (const char *e = LAB_GETENV ("MIBAYER_HOST_BANDS"))
    want = atoi (e);
  if (want > kMaxHostBands)
    want = kMaxHostBands;
  const size_t big_side = c->inverse ? c->src_bytes : c->dst_bytes;     /* the 4 B/px frame */
  if (want < 2 || plan_for (c, 1).var->persistent
      || big_side < ((size_t) 16 << 20))        /* below ~4K the extra enqueues cost more
                                                   than the overlap gains (1080p: -10 %) */
    return 1;
  const int th = c->inverse ? kInverseBandUnit : plan_for (c, 1).var->tile_h;
  const int tiles_y = (c->cfg.height + th - 1) / th;
  while (want > 1) {
    const int per = (tiles_y + want - 1) / want;        /* tile rows per band */
    const int nb = (tiles_y + per - 1) / per;           /* bands actually needed */
    const int last_y0 = (nb - 1) * per * th;
    if (nb == want && per * th >= 8 && c->cfg.height - last_y0 >= 4)
      return want;
    want--;
  }
  return 1;
}

static int enqueue_frame_banded (mibayer_ctx *c, Slot &s, const uint8_t *src,
    uint8_t *dst, size_t row_bytes)
{
  const mibayer_cfg &f = c->cfg;
  const int th = c->inverse ? kInverseBandUnit : plan_for (c, 1).var->tile_h;
  const int halo = c->inverse ? 0 : 1;  /* rgb2bayer has no neighbourhood */
  const int tiles_y = (f.height + th - 1) / th;
  const int nb = c->host_bands;
  const int per = (tiles_y + nb - 1) / nb;
  int uploaded = 0;             /* source rows already queued for upload */
  /* the band count follows the plan (choose_host_bands after every plan change) and may have grown since the slot was
   * made: the events a band needs are made when it first needs them (free_slot destroys all kMaxHostBands) */
  for (int b = 0; b < nb; b++) {
    if (!s.ev_band_in[b])
      HIP_TRY (hipEventCreateWithFlags (&s.ev_band_in[b], hipEventDisableTiming));
    if (!s.ev_band_kernel[b])
      HIP_TRY (hipEventCreateWithFlags (&s.ev_band_kernel[b], hipEventDisableTiming));
  }
  for (int b = 0; b < nb; b++) {
    const int t0 = b * per;
    const int t1 = t0 + per < tiles_y ? t0 + per : tiles_y;
    const int y0 = t0 * th;
    const int y1 = t1 * th < f.height ? t1 * th : f.height;
    const int up_to = y1 + halo < f.height ? y1 + halo : f.height;     /* + the halo row below */
    if (up_to > uploaded) {
      const size_t off = (size_t) uploaded * f.src_stride;
      HIP_TRY (hipMemcpyAsync (s.d_src + off, src + off,
              (size_t) (up_to - uploaded) * f.src_stride, hipMemcpyHostToDevice,
              c->s_h2d));
      uploaded = up_to;
    }
    HIP_TRY (hipEventRecord (s.ev_band_in[b], c->s_h2d));
    HIP_TRY (hipStreamWaitEvent (c->s_compute, s.ev_band_in[b], 0));
    int rc = launch (c, s.d_src, c->src_bytes, s.d_dst, c->dst_bytes, 1,
        c->s_compute, t0, t1 - t0);
    if (rc != MIBAYER_OK)
      return rc;
    HIP_TRY (hipEventRecord (s.ev_band_kernel[b], c->s_compute));
    HIP_TRY (hipStreamWaitEvent (c->s_d2h, s.ev_band_kernel[b], 0));
    const size_t doff = (size_t) y0 * f.dst_stride;
    if ((size_t) f.dst_stride == row_bytes) {
      HIP_TRY (hipMemcpyAsync (dst + doff, s.d_dst + doff,
              (size_t) (y1 - y0) * f.dst_stride, hipMemcpyDeviceToHost,
              c->s_d2h));
    } else {
      /* padded destination rows: only the written bytes of each row */
      HIP_TRY (hipMemcpy2DAsync (dst + doff, (size_t) f.dst_stride,
              s.d_dst + doff, (size_t) f.dst_stride, row_bytes,
              (size_t) (y1 - y0), hipMemcpyDeviceToHost, c->s_d2h));
    }
  }
  HIP_TRY (hipEventRecord (s.ev_out, c->s_d2h));
  return MIBAYER_OK;
}

/* BASELINE.json configs[4] "pinned double-buffered H2D/D2H + hipGraph-captured
 * launch": what the compute queue does for a frame -- wait for the slot's upload
 * event, run the kernel on the slot's device frames, record the slot's kernel
 * event -- never changes from frame to frame (the slot's device pointers and
 * events are fixed), so it is captured once per slot as a three-node graph and
 * replayed with one hipGraphLaunch per frame; nothing is patched per frame.  The
 * copies stay plain asynchronous copies on the two copy queues, where frame
 * n+1's upload and frame n's download keep overlapping exactly as without the
 * graph.  If the runtime refuses the event nodes the graph holds the kernel only
 * and the wait / record stay stream calls. */
static int compute_graph_launch (mibayer_ctx *c, Slot &s)
{
  if (s.cexec && s.plan_epoch != c->plan_epoch) {       /* built under an older plan; the slot is free */
    (void) hipGraphExecDestroy (s.cexec);
    (void) hipGraphDestroy (s.cgraph);
    s.cexec = nullptr;
    s.cgraph = nullptr;
  }
  if (!s.cexec) {
    s.plan_epoch = c->plan_epoch;
    KParams p;
    KernelFn kern;
    unsigned grid;
    Geometry g;
    const int rc = plan_launch (c, s.d_src, c->src_bytes, s.d_dst, c->dst_bytes,
        1, p, kern, grid, &g);
    if (rc != MIBAYER_OK)
      return rc;
    void *args[1] = { &p };
    hipKernelNodeParams kp;
    memset (&kp, 0, sizeof kp);
    kp.func = (void *) kern;
    kp.gridDim = dim3 (grid);
    kp.blockDim = dim3 ((unsigned) g.var->threads);
    kp.kernelParams = args;
    for (int with_events = 1; with_events >= 0 && !s.cexec; with_events--) {
      hipGraphNode_t n_wait = nullptr, n_kernel = nullptr, n_rec = nullptr;
      bool bad = hip_failed (hipGraphCreate (&s.cgraph, 0), "hipGraphCreate");
      if (with_events)
        bad = bad || hip_failed (hipGraphAddEventWaitNode (&n_wait, s.cgraph,
                NULL, 0, s.ev_in), "hipGraphAddEventWaitNode");
      bad = bad || hip_failed (hipGraphAddKernelNode (&n_kernel, s.cgraph,
              with_events ? &n_wait : NULL, with_events ? 1 : 0, &kp),
          "hipGraphAddKernelNode");
      if (with_events)
        bad = bad || hip_failed (hipGraphAddEventRecordNode (&n_rec, s.cgraph,
                &n_kernel, 1, s.ev_kernel), "hipGraphAddEventRecordNode");
      bad = bad || hip_failed (hipGraphInstantiate (&s.cexec, s.cgraph, NULL,
              NULL, 0), "hipGraphInstantiate");
      if (bad) {
        (void) hipGetLastError ();
        if (s.cgraph)
          (void) hipGraphDestroy (s.cgraph);
        s.cgraph = nullptr;
        s.cexec = nullptr;
        if (!with_events)
          return MIBAYER_ERR_HIP;
      } else {
        s.cgraph_events = with_events != 0;
      }
    }
  }
  if (!s.cgraph_events)
    HIP_TRY (hipStreamWaitEvent (c->s_compute, s.ev_in, 0));
  HIP_TRY (hipGraphLaunch (s.cexec, c->s_compute));
  if (!s.cgraph_events)
    HIP_TRY (hipEventRecord (s.ev_kernel, c->s_compute));
  return MIBAYER_OK;
}

/* upload -> kernel -> download of one frame through slot `s`, chained by events
 * across the three queues */
static int enqueue_plain (mibayer_ctx *c, Slot &s, const uint8_t *src,
    uint8_t *dst, size_t row_bytes, bool compute_graph = false)
{
  {
    Range r ("mibayer:h2d");
    HIP_TRY (hipMemcpyAsync (s.d_src, src, c->src_bytes, hipMemcpyHostToDevice,
            c->s_h2d));
    HIP_TRY (hipEventRecord (s.ev_in, c->s_h2d));
  }
  if (compute_graph) {
    Range r ("mibayer:kernel(graph)");
    const int rc = compute_graph_launch (c, s);
    if (rc != MIBAYER_OK)
      return rc;
  } else {
    Range r ("mibayer:kernel");
    HIP_TRY (hipStreamWaitEvent (c->s_compute, s.ev_in, 0));
    const int rc = launch (c, s.d_src, c->src_bytes, s.d_dst, c->dst_bytes, 1,
        c->s_compute);
    if (rc != MIBAYER_OK)
      return rc;
    HIP_TRY (hipEventRecord (s.ev_kernel, c->s_compute));
  }
  Range r ("mibayer:d2h");
  HIP_TRY (hipStreamWaitEvent (c->s_d2h, s.ev_kernel, 0));
  if ((size_t) c->cfg.dst_stride == row_bytes) {
    HIP_TRY (hipMemcpyAsync (dst, s.d_dst, c->dst_bytes, hipMemcpyDeviceToHost,
            c->s_d2h));
  } else {
    /* padded destination rows: only the written bytes of each row may be
     * touched (the reference never writes the padding either) */
    HIP_TRY (hipMemcpy2DAsync (dst, (size_t) c->cfg.dst_stride, s.d_dst,
            (size_t) c->cfg.dst_stride, row_bytes, (size_t) c->cfg.height,
            hipMemcpyDeviceToHost, c->s_d2h));
  }
  HIP_TRY (hipEventRecord (s.ev_out, c->s_d2h));
  return MIBAYER_OK;
}

static size_t written_row_bytes (const mibayer_ctx *c)
{
  return c->inverse ? (size_t) ((c->cfg.width + 3) & ~3)
      : (size_t) 4 * c->cfg.width;
}

static int enqueue_frame (mibayer_ctx *c, const uint8_t *src, uint8_t *dst,
    void *tag, bool alone)
{
  int rc = ensure_ring (c);
  if (rc != MIBAYER_OK)
    return rc;
  if (c->pending == (int) c->ring.size ())
    return MIBAYER_ERR_BUSY;
  Slot &s = c->ring[(size_t) c->head];
  const size_t row_bytes = written_row_bytes (c);       /* bytes of a destination row that are written */
  const bool want_graph = (c->cfg.flags & MIBAYER_FLAG_HIPGRAPH) && !c->inverse;
  if (want_graph && c->graph_mode == 1
      && (size_t) c->cfg.dst_stride == row_bytes) {
    rc = graph_submit (c, s, src, dst);
    if (rc != MIBAYER_OK)
      return rc;
    s.tag = tag;
    c->head = (c->head + 1) % (int) c->ring.size ();
    c->pending++;
    return MIBAYER_OK;
  }
  /* bands pay when this frame has the link to itself (measured at 4K: +7 %
   * synchronous, -6 % with three frames in flight, which overlap anyway) */
  if (c->host_bands > 1 && alone && c->pending == 0 && !want_graph) {
    rc = enqueue_frame_banded (c, s, src, dst, row_bytes);
    if (rc != MIBAYER_OK)
      return rc;
    s.tag = tag;
    c->head = (c->head + 1) % (int) c->ring.size ();
    c->pending++;
    return MIBAYER_OK;
  }
  rc = enqueue_plain (c, s, src, dst, row_bytes, want_graph);
  if (rc != MIBAYER_OK)
    return rc;
  s.tag = tag;
  c->head = (c->head + 1) % (int) c->ring.size ();
  c->pending++;
  return MIBAYER_OK;
}

/* A submit that fails half-way (say the upload was queued and the launch was
 * refused) must not leave work behind that still touches the caller's buffers
 * after the error has been returned: drain whatever was queued. */
static int submit_locked (mibayer_ctx *c, const uint8_t *src, uint8_t *dst,
    void *tag, bool alone)
{
  if (wedged_for_good (c))
    return MIBAYER_ERR_TIMEOUT;
  const int rc = enqueue_frame (c, src, dst, tag, alone);
  if (rc != MIBAYER_OK && rc != MIBAYER_ERR_BUSY) {
    char keep[sizeof t_hip_error];
    memcpy (keep, t_hip_error, sizeof keep);    /* report the first error */
    fence_queues (c);
    (void) wait_own_frames (c);
    /* a context whose device call failed leaves the queues it shares with the other contexts of the device: what it
     * does from now on (the pool re-does its frames elsewhere and abandons it) stays its own */
    if (rc == MIBAYER_ERR_HIP)
      mibayer_internal_private_queues (c);
    memcpy (t_hip_error, keep, sizeof keep);
  }
  return rc;
}

static int wait_locked (mibayer_ctx *c, void **tag)
{
  if (c->pending == 0)
    return MIBAYER_ERR_EMPTY;
  Slot &s = c->ring[(size_t) c->tail];
  {
    Range r ("mibayer:wait");
    const int rc = wait_event (c, s.ev_out, c->pending == 1);
    if (rc != MIBAYER_OK)
      return rc;                /* the frame stays where it is: its buffers are still the device's */
  }
  if (tag)
    *tag = s.tag;
  c->tail = (c->tail + 1) % (int) c->ring.size ();
  c->pending--;
  return MIBAYER_OK;
}

extern "C" int mibayer_submit (mibayer_ctx *c, const uint8_t *src,
    uint8_t *dst, void *tag)
{
  if (!c || !src || !dst)
    return MIBAYER_ERR_ARG;
  DeviceGuard guard (c->device);
  if (!guard.ok)
    return MIBAYER_ERR_HIP;
  c->stats.submits++;
  CpuMeter cpu (&c->stats.submit_cpu_ms);
  /* a ring of one frame is the synchronous 1-in/1-out use */
  return submit_locked (c, src, dst, tag, c->cfg.inflight == 1);
}

extern "C" int mibayer_wait (mibayer_ctx *c, void **tag)
{
  if (!c)
    return MIBAYER_ERR_ARG;
  DeviceGuard guard (c->device);
  if (!guard.ok)
    return MIBAYER_ERR_HIP;
  return wait_locked (c, tag);
}

extern "C" int mibayer_pending (const mibayer_ctx *c)
{
  return c ? c->pending : MIBAYER_ERR_ARG;
}

extern "C" int mibayer_process_host (mibayer_ctx *c, const uint8_t *src,
    uint8_t *dst)
{
  if (!c || !src || !dst)
    return MIBAYER_ERR_ARG;
  if (c->pending != 0)
    return MIBAYER_ERR_BUSY;
  DeviceGuard guard (c->device);
  if (!guard.ok)
    return MIBAYER_ERR_HIP;
  int rc;
  {
    c->stats.submits++;
    CpuMeter cpu (&c->stats.submit_cpu_ms);
    rc = submit_locked (c, src, dst, NULL, true);
  }
  if (rc != MIBAYER_OK)
    return rc;
  return wait_locked (c, NULL);
}

/* ---- seam for the frame-sharding pool (mibayer_hooks.h, mibayer_pool.cpp) ------------------ */

extern "C" int mibayer_internal_run_spare (mibayer_ctx *c, const uint8_t *src,
    uint8_t *dst)
{
  if (!c || !src || !dst)
    return MIBAYER_ERR_ARG;
  DeviceGuard guard (c->device);
  if (!guard.ok)
    return MIBAYER_ERR_HIP;
  if (!c->spare_ready) {
    const int rc = alloc_slot (c, c->spare, false);
    if (rc != MIBAYER_OK) {
      free_slot (c, c->spare);
      return rc;
    }
    c->spare_ready = true;
  }
  if (wedged_for_good (c))
    return MIBAYER_ERR_TIMEOUT;
  int rc = enqueue_plain (c, c->spare, src, dst, written_row_bytes (c));
  if (rc != MIBAYER_OK) {
    /* nothing of a half-queued frame may touch the buffers after the error */
    fence_queues (c);
    return rc;
  }
  Range r ("mibayer:wait");
  return wait_event (c, c->spare.ev_out, true);
}

extern "C" int mibayer_internal_is_pageable (const void *p)
{
  hipPointerAttribute_t attr;
  memset (&attr, 0, sizeof attr);
  if (hipPointerGetAttributes (&attr, p) != hipSuccess) {
    (void) hipGetLastError ();  /* "invalid value" is the answer for plain malloc memory */
    return 1;
  }
  return attr.type == hipMemoryTypeUnregistered ? 1 : 0;
}

extern "C" void mibayer_internal_private_queues (mibayer_ctx *c)
{
  if (!c || !c->shared_queues)
    return;
  DeviceGuard guard (c->device);
  if (!guard.ok)
    return;
  hipStream_t q[3] = { nullptr, nullptr, nullptr };
  bool bad = false;
  for (hipStream_t &st : q)
    bad |= hip_failed (hipStreamCreateWithFlags (&st, hipStreamNonBlocking),
        "hipStreamCreate");
  if (bad) {                    /* keep the shared ones */
    for (hipStream_t st : q)
      if (st)
        (void) hipStreamDestroy (st);
    (void) hipGetLastError ();
    return;
  }
  /* what this context has in flight on the shared queues completes there: a
   * slot is reused only after its own download event has been waited for, and
   * that event was recorded on the queue the download ran on */
  {
    std::lock_guard<std::mutex> lk (g_queues_mu);
    DeviceQueues &dq = g_queues[c->device];
    if (--dq.refs == 0) {
      /* the last user of the shared set: whatever is still queued there is this context's own */
      (void) hipStreamDestroy (dq.h2d);
      (void) hipStreamDestroy (dq.compute);
      (void) hipStreamDestroy (dq.d2h);
      dq = DeviceQueues ();
    }
  }
  c->s_h2d = q[0];
  c->s_compute = q[1];
  c->s_d2h = q[2];
  c->shared_queues = false;
  /* graphs captured for the old compute queue replay on any queue; nothing to rebuild */
}

extern "C" int mibayer_host_is_pinned (const void *p)
{
  if (!p || device_count_cached () <= 0)
    return 0;
  hipPointerAttribute_t attr;
  memset (&attr, 0, sizeof attr);
  if (hipPointerGetAttributes (&attr, p) != hipSuccess) {
    (void) hipGetLastError ();
    return 0;
  }
  return attr.type == hipMemoryTypeHost ? 1 : 0;
}

extern "C" int mibayer_internal_abandon (mibayer_ctx *c)
{
  if (!c)
    return MIBAYER_OK;
  if (c->wedged)                /* a device that does not answer is not waited for again */
    return ctx_settled (c) ? MIBAYER_OK : MIBAYER_ERR_TIMEOUT;
  DeviceGuard guard (c->device);
  (void) wait_own_frames (c);
  fence_queues (c);
  (void) hipGetLastError ();
  return c->wedged ? MIBAYER_ERR_TIMEOUT : MIBAYER_OK;
}

extern "C" int mibayer_internal_settled (mibayer_ctx *c)
{
  if (!c)
    return 1;
  DeviceGuard guard (c->device);
  return ctx_settled (c) ? 1 : 0;
}

extern "C" int mibayer_internal_stall (mibayer_ctx *c, int ms)
{
  if (!c || ms < 1 || ms > 5000)
    return MIBAYER_ERR_ARG;
  DeviceGuard guard (c->device);
  if (!guard.ok)
    return MIBAYER_ERR_HIP;
  HIP_TRY (launch_stall (ms, c->s_compute));
  return MIBAYER_OK;
}

extern "C" int mibayer_set_wait_spin (mibayer_ctx *c, int spin_us)
{
  if (!c)
    return MIBAYER_ERR_ARG;
  c->wait_spin_us = spin_us < 0 ? -1 : (spin_us > 1000000 ? 1000000 : spin_us);
  return MIBAYER_OK;
}

extern "C" int mibayer_get_host_stats (const mibayer_ctx *c, mibayer_host_stats *out)
{
  if (!c || !out)
    return MIBAYER_ERR_ARG;
  *out = c->stats;
  return MIBAYER_OK;
}

extern "C" int mibayer_set_wait_timeout (mibayer_ctx *c, int ms)
{
  if (!c)
    return MIBAYER_ERR_ARG;
  c->wait_timeout_ms = ms < 0 ? 10000 : ms;
  return MIBAYER_OK;
}

/* ---- device-resident batch path ------------------------------------------------------ */

/* device-resident work is about to be queued on `s` through this context: remember which of the context's queues
 * mibayer_sync has to fence */
static void mark_dirty (mibayer_ctx *c, hipStream_t s)
{
  if (s == c->s_compute) {
    c->dirty_compute = true;
  } else if (s && c->uses_frame_queues && c->device < 64) {
    const FrameQueues &fq = g_frame_queues[c->device];  /* the pointers are stable while this context lives */
    for (int k = 0; k < MIBAYER_FRAME_QUEUES; k++)
      if (fq.q[k] == s)
        c->dirty_frame[k] = true;
  }
}

extern "C" int mibayer_process_device (mibayer_ctx *c, const void *d_src,
    size_t src_frame_bytes, void *d_dst, size_t dst_frame_bytes, int nframes,
    void *hip_stream)
{
  if (!c || !d_src || !d_dst || nframes < 0)
    return MIBAYER_ERR_ARG;
  if (nframes > 1 && (src_frame_bytes < c->src_bytes
          || dst_frame_bytes < c->dst_bytes || (src_frame_bytes & 3)
          || (dst_frame_bytes & 3)))
    return MIBAYER_ERR_GEOMETRY;
  if ((((uintptr_t) d_src) & 3) || (((uintptr_t) d_dst) & 3))
    return MIBAYER_ERR_ARG;
  DeviceGuard guard (c->device);
  if (!guard.ok)
    return MIBAYER_ERR_HIP;
  Range r ("mibayer:process_device");
  mark_dirty (c, (hipStream_t) hip_stream);
  return launch (c, d_src, src_frame_bytes, d_dst, dst_frame_bytes, nframes,
      (hipStream_t) hip_stream);
}

/* One launch over frames that are separate device allocations (a GstBuffer
 * each): hipbayer2rgb's batch mode.  A 4K frame is ~7 us of kernel, about what
 * a launch costs to issue, so a device-resident pipeline that converts frame by
 * frame is bound by launch latency (5.8 k fps = 48 Gpix/s, against 1.3 Tpix/s
 * for one launch over 64 resident frames); a list launch amortises it over up
 * to kMaxList frames without asking the caller to make them contiguous. */
extern "C" int mibayer_process_device_list (mibayer_ctx *c,
    const void *const *d_srcs, void *const *d_dsts, int nframes,
    void *hip_stream)
{
  if (!c || !d_srcs || !d_dsts || nframes < 0)
    return MIBAYER_ERR_ARG;
  for (int f = 0; f < nframes; f++)
    if (!d_srcs[f] || !d_dsts[f] || (((uintptr_t) d_srcs[f]) & 3)
        || (((uintptr_t) d_dsts[f]) & 3))
      return MIBAYER_ERR_ARG;
  DeviceGuard guard (c->device);
  if (!guard.ok)
    return MIBAYER_ERR_HIP;
  Range r ("mibayer:process_device_list");
  mark_dirty (c, (hipStream_t) hip_stream);
  if (c->inverse) {
    /* the sibling direction (reference loop gst/bayer/gstrgb2bayer.c:254-268): up to kMaxList separately allocated
     * frames per launch of the flat kernel; the tile kernel (MIBAYER_R2B_FLAT=0, tuning) has no table and goes
     * frame by frame */
    const mibayer_cfg &f = c->cfg;
    for (int f0 = 0; f0 < nframes; f0 += kMaxList) {
      const int n = nframes - f0 < kMaxList ? nframes - f0 : kMaxList;
      if (c->r2b_flat_k <= 0 || (long long) f.height * (((f.width + 3) & ~3) / 4) > 0x7fffffffLL) {
        for (int k = 0; k < n; k++) {
          const int rc = launch (c, d_srcs[f0 + k], c->src_bytes, d_dsts[f0 + k],
              c->dst_bytes, 1, (hipStream_t) hip_stream);
          if (rc != MIBAYER_OK)
            return rc;
        }
        continue;
      }
      R2BParams q;
      q.src = nullptr;
      q.dst = nullptr;
      q.src_frame_bytes = 0;
      q.dst_frame_bytes = 0;
      q.width = f.width;
      q.height = f.height;
      q.src_stride = f.src_stride;
      q.dst_stride = f.dst_stride;
      q.out_dwords = ((f.width + 3) & ~3) / 4;
      q.total_rows = f.height;
      q.band = c->plan[PLAN_BATCH].band != INT32_MIN ? c->plan[PLAN_BATCH].band : 0;
      q.start_sleep = c->start_sleep > 0 ? c->start_sleep : 0;
      q.flat_k = c->r2b_flat_k;
      q.flat_px = c->r2b_flat_px;
      q.flat_ld = c->r2b_flat_ld;
      q.rows = c->r2b_rows;
      for (int k = 0; k < 2; k++) {
        q.sel_lo[k] = c->r2b_lo[k];
        q.sel_hi[k] = c->r2b_hi[k];
      }
      q.nlist = n;
      bool src16 = true, dst8 = true;
      for (int k = 0; k < n; k++) {
        q.src_list[k] = (const uint8_t *) d_srcs[f0 + k];
        q.dst_list[k] = (uint8_t *) d_dsts[f0 + k];
        src16 = src16 && aligned16 (d_srcs[f0 + k]);
        dst8 = dst8 && (((uintptr_t) d_dsts[f0 + k]) & 7u) == 0;
      }
      const bool vec16 = (f.width % 4 == 0) && (f.src_stride % 16 == 0) && src16;
      HIP_TRY (launch_rgb2bayer_list (q, vec16, dst8, (hipStream_t) hip_stream));
    }
    return MIBAYER_OK;
  }
  for (int f0 = 0; f0 < nframes; f0 += kMaxList) {
    const int n = nframes - f0 < kMaxList ? nframes - f0 : kMaxList;
    bool all16 = true, dst8 = true;
    for (int f = 0; f < n; f++) {
      all16 = all16 && aligned16 (d_srcs[f0 + f]) && aligned16 (d_dsts[f0 + f]);
      dst8 = dst8 && (((uintptr_t) d_dsts[f0 + f]) & 7u) == 0;
    }
    KParams p;
    KernelFn kern;
    unsigned grid;
    /* planned as a batch of n frames; the 16-byte path needs every pointer aligned */
    Geometry g;
    const int rc = plan_launch (c, d_srcs[f0], c->src_bytes, d_dsts[f0],
        c->dst_bytes, n, p, kern, grid, &g, 0, -1, all16 ? 1 : (dst8 ? 2 : 0));
    if (rc != MIBAYER_OK)
      return rc;
    p.src = nullptr;
    p.dst = nullptr;
    p.nlist = n;
    for (int f = 0; f < n; f++) {
      p.src_list[f] = (const uint8_t *) d_srcs[f0 + f];
      p.dst_list[f] = (uint8_t *) d_dsts[f0 + f];
    }
    hipLaunchKernelGGL (kern, dim3 (grid), dim3 (g.var->threads), 0,
        (hipStream_t) hip_stream, p);
    HIP_TRY (hipGetLastError ());
  }
  return MIBAYER_OK;
}

extern "C" void *mibayer_ctx_stream (mibayer_ctx *c)
{
  return c ? (void *) c->s_compute : NULL;
}

extern "C" void *mibayer_ctx_frame_queue (mibayer_ctx *c, int k)
{
  if (!c || k < 0 || k >= MIBAYER_FRAME_QUEUES || c->device >= 64)
    return NULL;
  DeviceGuard guard (c->device);
  if (!guard.ok)
    return NULL;
  std::lock_guard<std::mutex> lk (g_queues_mu);
  FrameQueues &fq = g_frame_queues[c->device];
  if (!fq.tried) {
    fq.tried = true;
    /* "every CU" as a mask: what makes the stream own a hardware queue is the call, not the partition */
    uint32_t mask[32];
    const int words = (c->num_cus + 31) / 32 < 32 ? (c->num_cus + 31) / 32 : 32;
    for (int w = 0; w < words; w++) {
      const int bits = c->num_cus - 32 * w;
      mask[w] = bits >= 32 ? 0xffffffffu : ((1u << bits) - 1u);
    }
    for (hipStream_t &st : fq.q)
      if (hipExtStreamCreateWithCUMask (&st, (uint32_t) words, mask) != hipSuccess) {
        (void) hipGetLastError ();
        st = nullptr;
        /* no dedicated hardware queue to be had: an ordinary stream still overlaps some */
        if (hip_failed (hipStreamCreateWithFlags (&st, hipStreamNonBlocking), "hipStreamCreate"))
          st = nullptr;
      }
  }
  if (fq.q[k])
    c->uses_frame_queues = true;
  return (void *) fq.q[k];
}

/* Waits for what THIS context has in flight: the download events of its own pending frames and, if it queued
 * device-resident work on its compute queue since the last call, a fence behind that work.  The queues may be shared
 * with the other contexts of the device (DeviceQueues): their copies and kernels are not waited for. */
extern "C" int mibayer_sync (mibayer_ctx *c)
{
  if (!c)
    return MIBAYER_ERR_ARG;
  DeviceGuard guard (c->device);
  if (!guard.ok)
    return MIBAYER_ERR_HIP;
  int rc = wait_own_frames (c);
  if (rc != MIBAYER_OK)
    return rc;
  struct { hipStream_t q; bool *dirty; } own[1 + MIBAYER_FRAME_QUEUES] = { { c->s_compute, &c->dirty_compute } };
  if (c->uses_frame_queues && c->device < 64) {
    std::lock_guard<std::mutex> lk (g_queues_mu);
    for (int k = 0; k < MIBAYER_FRAME_QUEUES; k++)
      own[1 + k] = { g_frame_queues[c->device].q[k], &c->dirty_frame[k] };
  } else {
    for (int k = 0; k < MIBAYER_FRAME_QUEUES; k++)
      own[1 + k] = { nullptr, &c->dirty_frame[k] };
  }
  for (auto &o : own) {
    if (!*o.dirty || !o.q)
      continue;
    /* not `c->wedged`: with nothing pending wait_own_frames() polled nothing, so this is where a context whose
     * device-resident launch ran into a deadline finds out that the device has caught up */
    if (wedged_for_good (c))
      return MIBAYER_ERR_TIMEOUT;
    HIP_TRY (hipEventRecord (c->ev_fence, o.q));
    rc = wait_event (c, c->ev_fence, true);
    if (rc != MIBAYER_OK)
      return rc;
    *o.dirty = false;
  }
  return MIBAYER_OK;
}

/* `reps` back-to-back calls of `launch_once` (which queues work on the context's compute queue) between two HIP events
 * on that queue, after `warmup` untimed calls; mean milliseconds per call */
template <typename F>
static int time_launches (mibayer_ctx *c, F launch_once, int warmup, int reps, float *ms_per_launch)
{
  int rc;
  for (int i = 0; i < warmup; i++) {
    rc = launch_once ();
    if (rc != MIBAYER_OK)
      return rc;
  }
  HIP_TRY (hipEventRecord (c->ev_t0, c->s_compute));
  for (int i = 0; i < reps; i++) {
    rc = launch_once ();
    if (rc != MIBAYER_OK)
      return rc;
  }
  HIP_TRY (hipEventRecord (c->ev_t1, c->s_compute));
  HIP_TRY (hipEventSynchronize (c->ev_t1));
  float ms = 0.f;
  HIP_TRY (hipEventElapsedTime (&ms, c->ev_t0, c->ev_t1));
  *ms_per_launch = ms / (float) reps;
  return MIBAYER_OK;
}

extern "C" int mibayer_time_device (mibayer_ctx *c, const void *d_src,
    size_t src_frame_bytes, void *d_dst, size_t dst_frame_bytes, int nframes,
    int warmup, int reps, float *ms_per_launch)
{
  if (!c || !ms_per_launch || reps < 1 || warmup < 0)
    return MIBAYER_ERR_ARG;
  DeviceGuard guard (c->device);
  if (!guard.ok)
    return MIBAYER_ERR_HIP;
  return time_launches (c, [&] {
    return mibayer_process_device (c, d_src, src_frame_bytes, d_dst, dst_frame_bytes, nframes, c->s_compute);
  }, warmup, reps, ms_per_launch);
}

/* ---- process-wide plan cache ------------------------------------------------------------ */

/* What mibayer_autotune measured, kept per (device, geometry, launch class): the next context of the same stream
 * geometry on that device -- the second element instance, the context after a renegotiation, the other Bayer orders of
 * one camera -- starts from the measured plan instead of the static default, without measuring again.  The launch
 * class is part of the key: what a 64-frame batch measured is not what a frame-by-frame context should
 * start from, and the other way round.  Reference analogue (compile once per process, reuse): the once-guarded ORC
 * program set-up, gst/bayer/gstbayerorc-dist.c:321-397. */
namespace {
struct PlanEntry {
  int device, width, height, src_stride, dst_stride, klass;
  int variant, band, align;
};
std::mutex g_plan_mu;
std::vector<PlanEntry> g_plans;

bool plan_cache_enabled ()
{
  static const bool on = [] {
    const char *e = getenv ("MIBAYER_PLAN_CACHE");
    return !(e && e[0] == '0');
  } ();
  return on;
}

bool plan_key_is (const PlanEntry &e, const mibayer_ctx *c, int klass)
{
  return e.device == c->device && e.width == c->cfg.width && e.height == c->cfg.height
      && e.src_stride == c->cfg.src_stride && e.dst_stride == c->cfg.dst_stride && e.klass == klass;
}

void plan_cache_store (const mibayer_ctx *c, int klass)
{
  if (!plan_cache_enabled () || c->inverse || c->cfg.variant != 0)
    return;
  std::lock_guard<std::mutex> lk (g_plan_mu);
  PlanEntry *slot = nullptr;
  for (PlanEntry &e : g_plans)
    if (plan_key_is (e, c, klass))
      slot = &e;
  if (!slot) {
    if (g_plans.size () >= 256)
      g_plans.erase (g_plans.begin ());
    g_plans.push_back (PlanEntry ());
    slot = &g_plans.back ();
  }
  const Plan &pl = c->plan[klass];
  *slot = PlanEntry { c->device, c->cfg.width, c->cfg.height, c->cfg.src_stride, c->cfg.dst_stride, klass,
    (int) (pl.var - &variant (0)), pl.band, pl.align };
}

}  /* namespace */

static bool plan_cache_load (mibayer_ctx *c)
{
  if (!plan_cache_enabled () || c->inverse || c->cfg.variant != 0)
    return false;
  std::lock_guard<std::mutex> lk (g_plan_mu);
  bool hit = false;
  for (int klass = 0; klass < PLAN_CLASSES; klass++)
    for (const PlanEntry &e : g_plans)
      if (plan_key_is (e, c, klass)) {
        c->plan[klass] = Plan { &variant (e.variant), e.band, e.align, MIBAYER_PLAN_CACHED };
        c->plan_epoch++;
        hit = true;
      }
  return hit;
}

extern "C" void mibayer_plan_cache_clear (void)
{
  std::lock_guard<std::mutex> lk (g_plan_mu);
  g_plans.clear ();
}

extern "C" int mibayer_plan_from_cache (mibayer_ctx *c)
{
  if (!c)
    return MIBAYER_ERR_ARG;
  if (c->band_forced || !c->align_tunable)      /* (lab builds) a plan pinned from the environment stays */
    return 0;
  if (!plan_cache_load (c))
    return 0;
  c->host_bands = choose_host_bands (c);        /* the frame-class tile height may have changed */
  return 1;
}

extern "C" int mibayer_plan_source (const mibayer_ctx *c)
{
  return c ? c->plan[PLAN_BATCH].source : MIBAYER_ERR_ARG;
}

/* Measured plan selection.  The kernel is idempotent and deterministic, so the
 * candidates are simply run on the caller's real buffers: d_dst ends up holding
 * the correct output whichever plan wins. */
namespace {
struct Candidate {
  const Variant *var;
  int band;                     /* band_override value */
  int align;                    /* align_stores value */
  float ms;
  bool alive;
};
constexpr int kMaxCands = 48;
}

/* `launch_once` queues one conversion of the caller's buffers under the context's CURRENT plan of launch class `klass`
 * (the class those launches fall into) on the compute queue; `generic_off_grid`: those launches take the generic
 * kernel and the output rows sit off the sector grid */
template <typename F>
static int autotune_core (mibayer_ctx *c, int klass, F launch_once, bool generic_off_grid, char *report,
    size_t report_len)
{
  Plan &pl = c->plan[klass];
  /* Candidate plans.  Sector-aligned geometries (the 16-byte kernel): {configured shape, the other production
   * shapes under "auto"} x {band 1, one chunk per XCD, identity}; the band orders carry the automatic start delay.
   * Generic geometries whose rows sit off the sector grid add the store policy as a dimension -- streaming,
   * write-back, hybrid (mibayer_kernels.hip) -- and the shifted arm (every wave-store on a 128-byte boundary) in
   * the two narrow shapes: which of them wins depends on the row phase and on the box
   * (profiles/r03_generic_path.log). */
  const bool band_forced = c->band_forced;
  static const int kBands[3] = { 1, -1, 0 };
  const int nbands = band_forced ? 1 : 3;
  Candidate cand[kMaxCands];
  int ncand = 0;
  auto add = [&](const Variant *v, int band, int align) {
    if (align && !(align == 128 ? v->aligned128 : v->aligned64))
      align = 0;                /* this shape has no such arm: it runs unshifted */
    for (int i = 0; i < ncand; i++)
      if (cand[i].var == v && cand[i].band == band && cand[i].align == align)
        return;
    if (ncand < kMaxCands)
      cand[ncand++] = Candidate { v, band, align, 0.f, true };
  };
  const Plan keep = pl;
  const int keep_band = keep.band;
  const int keep_align = keep.align;
  /* a forced store alignment (MIBAYER_ALIGN_STORES) is part of every candidate */
  const int fixed_align = c->align_tunable ? 0 : keep_align;
  /* the configured plan first: it also wins ties */
  add (keep.var, band_forced ? keep_band : (keep_band == INT32_MIN ? kBands[0] : keep_band), fixed_align);
  if (c->cfg.variant != 0) {
    for (int bi = 0; bi < nbands; bi++)
      add (keep.var, band_forced ? keep_band : kBands[bi], fixed_align);
  } else {
    const int own_id = (int) (keep.var - &variant (0));
    for (int v = 3; v >= 1; v--) {
      int ids[3], nids = 0;
      if (generic_off_grid) {
        ids[nids++] = hybrid_store_twin (v);
        ids[nids++] = plain_store_twin (v);
        ids[nids++] = v;
      } else {
        /* the store policy the context was created with */
        ids[nids++] = own_id == plain_store_twin (production_shape_of (own_id)) && own_id != production_shape_of (own_id)
            ? plain_store_twin (v) : v;
      }
      for (int k = 0; k < nids; k++)
        for (int bi = 0; bi < nbands; bi++)
          add (&variant (ids[k]), band_forced ? keep_band : kBands[bi], fixed_align);
    }
    if (generic_off_grid && c->align_tunable) {
      add (&variant (3), band_forced ? keep_band : 0, 128);
      add (&variant (3), band_forced ? keep_band : 1, 128);
      add (&variant (2), band_forced ? keep_band : 0, 128);
      add (&variant (2), band_forced ? keep_band : 1, 128);
    }
  }

  /* The candidates are timed in interleaved rounds after a common, TIME-based
   * warm-up, and each is judged by the MEDIAN of its rounds: an idle GPU needs
   * tens of milliseconds to clock up (the first candidate used to lose for that
   * reason alone), one slow round (another process, a DVFS step) must not
   * decide the plan, and neither may one lucky round -- the plan has to be the
   * one that is fastest in steady state, which is what the caller then runs.
   * With many candidates (generic geometries) only those within 6 % of the best
   * of the first round stay in the race. */
  constexpr int kRounds = 5, kWarm = 16, kReps = 6, kKeep = 9;
  constexpr double kWarmMs = 60.0;
  float (*rm)[kRounds] = new (std::nothrow) float[kMaxCands][kRounds];
  if (!rm)
    return MIBAYER_ERR_NOMEM;
  int rc = MIBAYER_OK;
  float ms = 0.f;
  pl.align = fixed_align;
  {
    const double t0 = now_ms ();
    do {
      rc = time_launches (c, launch_once, 0, kWarm, &ms);
    } while (rc == MIBAYER_OK && now_ms () - t0 < kWarmMs);
  }
  for (int round = 0; round < kRounds && rc == MIBAYER_OK; round++) {
    for (int i = 0; i < ncand && rc == MIBAYER_OK; i++) {
      if (!cand[i].alive)
        continue;
      pl.var = cand[i].var;
      pl.band = cand[i].band;
      pl.align = cand[i].align;
      rc = time_launches (c, launch_once, 1, kReps, &ms);
      rm[i][round] = ms;
    }
    if (round == 0 && rc == MIBAYER_OK && ncand > kKeep) {
      float best = 0.f;
      for (int i = 0; i < ncand; i++)
        if (best == 0.f || rm[i][0] < best)
          best = rm[i][0];
      for (int i = 1; i < ncand; i++)           /* candidate 0, the configured plan, always stays */
        cand[i].alive = rm[i][0] <= best * 1.06f;
    }
  }
  if (rc != MIBAYER_OK) {
    pl = keep;
    delete[] rm;
    return rc;
  }
  int best = 0;
  size_t used = 0;
  for (int i = 0; i < ncand; i++) {
    if (!cand[i].alive) {
      cand[i].ms = rm[i][0];                     /* its one round, for the report */
    } else {
      float *r = rm[i];
      for (int a = 1; a < kRounds; a++)          /* insertion sort of 5 */
        for (int b = a; b > 0 && r[b] < r[b - 1]; b--) {
          const float t = r[b];
          r[b] = r[b - 1];
          r[b - 1] = t;
        }
      cand[i].ms = r[kRounds / 2];
      if (!cand[best].alive || cand[i].ms < cand[best].ms)
        best = i;
    }
    if (report && used < report_len) {
      int n = snprintf (report + used, report_len - used, "%s%s%s/band%d=%.4fms%s",
          used ? " " : "", cand[i].var->name, cand[i].align ? "+shift" : "",
          cand[i].band, cand[i].ms, cand[i].alive ? "" : "(dropped)");
      if (n > 0)
        used += (size_t) n;
    }
  }
  delete[] rm;
  pl = Plan { cand[best].var, cand[best].band, cand[best].align, MIBAYER_PLAN_MEASURED };
  c->plan_epoch++;
  c->host_bands = choose_host_bands (c);
  plan_cache_store (c, klass);
  return MIBAYER_OK;
}

extern "C" int mibayer_autotune (mibayer_ctx *c, const void *d_src,
    size_t src_frame_bytes, void *d_dst, size_t dst_frame_bytes, int nframes,
    char *report, size_t report_len)
{
  if (!c || !d_src || !d_dst || nframes < 1 || c->inverse)
    return MIBAYER_ERR_ARG;
  if (report && report_len)
    report[0] = 0;
  DeviceGuard guard (c->device);
  if (!guard.ok)
    return MIBAYER_ERR_HIP;
  bool generic_off_grid = false;
  {
    KParams p;
    KernelFn kern;
    unsigned grid;
    const int prc = plan_launch (c, d_src, src_frame_bytes, d_dst, dst_frame_bytes, nframes, p, kern, grid);
    if (prc != MIBAYER_OK)
      return prc;
    generic_off_grid = kern != plan_for (c, nframes).var->fast && c->rows_off_sector;
  }
  return autotune_core (c, launch_class (c, nframes), [&] {
    return mibayer_process_device (c, d_src, src_frame_bytes, d_dst, dst_frame_bytes, nframes, c->s_compute);
  }, generic_off_grid, report, report_len);
}

/* The same over frames that are separate device allocations (mibayer_process_device_list): what a device-resident
 * element holds (one GstBuffer per frame). */
extern "C" int mibayer_autotune_list (mibayer_ctx *c, const void *const *d_srcs, void *const *d_dsts, int nframes,
    char *report, size_t report_len)
{
  if (!c || !d_srcs || !d_dsts || nframes < 1 || c->inverse)
    return MIBAYER_ERR_ARG;
  if (report && report_len)
    report[0] = 0;
  DeviceGuard guard (c->device);
  if (!guard.ok)
    return MIBAYER_ERR_HIP;
  bool all16 = true;
  for (int f = 0; f < nframes; f++) {
    if (!d_srcs[f] || !d_dsts[f])
      return MIBAYER_ERR_ARG;
    all16 = all16 && aligned16 (d_srcs[f]) && aligned16 (d_dsts[f]);
  }
  const mibayer_cfg &f = c->cfg;
  const bool fast = all16 && (f.width % 16 == 0) && (f.src_stride % 16 == 0) && (f.dst_stride % 16 == 0);
  return autotune_core (c, launch_class (c, nframes < kMaxList ? nframes : kMaxList), [&] {
    return mibayer_process_device_list (c, d_srcs, d_dsts, nframes, c->s_compute);
  }, !fast && c->rows_off_sector, report, report_len);
}

extern "C" int mibayer_get_plan (const mibayer_ctx *c, int *variant_id, int *band, int *align_stores)
{
  if (!c)
    return MIBAYER_ERR_ARG;
  const Plan &pl = c->plan[PLAN_BATCH];
  if (variant_id)
    *variant_id = (int) (pl.var - &variant (0));
  if (band)
    *band = pl.band;
  if (align_stores)
    *align_stores = pl.align;
  return MIBAYER_OK;
}

/* the plan a launch over `nframes` frames runs under (its launch class), and where that plan came from */
extern "C" int mibayer_get_plan_for (const mibayer_ctx *c, int nframes, int *variant_id, int *band,
    int *align_stores, int *source)
{
  if (!c || nframes < 1)
    return MIBAYER_ERR_ARG;
  const Plan &pl = plan_for (c, nframes);
  if (variant_id)
    *variant_id = (int) (pl.var - &variant (0));
  if (band)
    *band = pl.band;
  if (align_stores)
    *align_stores = pl.align;
  if (source)
    *source = pl.source;
  return MIBAYER_OK;
}

static int plan_args_ok (const mibayer_ctx *c, int variant_id, int align_stores)
{
  if (!c || c->inverse)
    return MIBAYER_ERR_ARG;
  if (variant_id < 1 || variant_id >= variant_count ())
    return MIBAYER_ERR_ARG;
  if (align_stores != 0 && align_stores != 64 && align_stores != 128)
    return MIBAYER_ERR_ARG;
  if (align_stores && !(align_stores == 128 ? variant (variant_id).aligned128 : variant (variant_id).aligned64))
    return MIBAYER_ERR_ARG;
  return MIBAYER_OK;
}

/* pins the plan of ONE launch class: the one launches over `nframes` frames fall into */
extern "C" int mibayer_set_plan_for (mibayer_ctx *c, int nframes, int variant_id, int band, int align_stores)
{
  const int rc = plan_args_ok (c, variant_id, align_stores);
  if (rc != MIBAYER_OK || nframes < 1)
    return rc != MIBAYER_OK ? rc : MIBAYER_ERR_ARG;
  plan_for (c, nframes) = Plan { &variant (variant_id), band, align_stores, MIBAYER_PLAN_SET };
  c->plan_epoch++;
  c->host_bands = choose_host_bands (c);
  return MIBAYER_OK;
}

extern "C" int mibayer_set_plan (mibayer_ctx *c, int variant_id, int band, int align_stores)
{
  const int rc = plan_args_ok (c, variant_id, align_stores);
  if (rc != MIBAYER_OK)
    return rc;
  for (Plan &pl : c->plan)      /* an explicit pin holds for every launch class */
    pl = Plan { &variant (variant_id), band, align_stores, MIBAYER_PLAN_SET };
  c->plan_epoch++;
  c->host_bands = choose_host_bands (c);
  return MIBAYER_OK;
}

extern "C" int mibayer_copy_plan (mibayer_ctx *dst, const mibayer_ctx *src)
{
  if (!dst || !src)
    return MIBAYER_ERR_ARG;
  if (dst->cfg.width != src->cfg.width || dst->cfg.height != src->cfg.height)
    return MIBAYER_ERR_GEOMETRY;
  for (int k = 0; k < PLAN_CLASSES; k++) {
    dst->plan[k] = src->plan[k];
    dst->plan[k].source = MIBAYER_PLAN_SET;
  }
  dst->plan_epoch++;
  dst->host_bands = choose_host_bands (dst);
  return MIBAYER_OK;
}

/* ---- memory helpers --------------------------------------------------------------------- */

extern "C" void *mibayer_host_alloc (size_t bytes)
{
  void *p = NULL;
  if (device_count_cached () <= 0)
    return NULL;
  /* portable: a pool buffer may be read / written by any GPU of a `devices=` list */
  if (hip_failed (hipHostMalloc (&p, bytes ? bytes : 1, hipHostMallocPortable),
          "hipHostMalloc"))
    return NULL;
  return p;
}

/* NUMA node of the CPU memory closest to a device: the runtime's answer, else
 * the PCI device's numa_node in sysfs; -1 = unknown / not a NUMA machine */
extern "C" int mibayer_device_numa_node (int device)
{
  if (device < 0 || device >= device_count_cached ())
    return -1;
  int node = -1;
  if (hipDeviceGetAttribute (&node, hipDeviceAttributeHostNumaId, device)
      == hipSuccess && node >= 0)
    return node;
  (void) hipGetLastError ();
  char bdf[64] = "";
  if (hipDeviceGetPCIBusId (bdf, (int) sizeof bdf, device) != hipSuccess) {
    (void) hipGetLastError ();
    return -1;
  }
  for (char *q = bdf; *q; q++)
    if (*q >= 'A' && *q <= 'F')
      *q = (char) (*q - 'A' + 'a');
  char path[160];
  snprintf (path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bdf);
  FILE *f = fopen (path, "r");
  if (!f)
    return -1;
  node = -1;
  if (fscanf (f, "%d", &node) != 1)
    node = -1;
  fclose (f);
  return node;
}

/* Pinned memory on the NUMA node next to `device`, for buffer pools that feed
 * that GPU: on a two-socket host a pool allocated by whatever thread happened to
 * configure it puts half of the frames behind the socket link.  The thread's
 * memory policy is set to "prefer that node" around a hipHostMallocNumaUser
 * allocation (which follows the caller's policy) and restored; freed with
 * mibayer_host_free like any other pinned block.  Falls back to
 * mibayer_host_alloc when the node is unknown. */
extern "C" void *mibayer_host_alloc_near (int device, size_t bytes)
{
  const int node = mibayer_device_numa_node (device);
  if (node < 0 || node >= 1024)
    return mibayer_host_alloc (bytes);
  constexpr int kMpolPreferred = 1;
  unsigned long mask[1024 / (8 * sizeof (unsigned long))];
  memset (mask, 0, sizeof mask);
  mask[(size_t) node / (8 * sizeof (unsigned long))]
      |= 1ul << ((size_t) node % (8 * sizeof (unsigned long)));
  /* the calling thread's own policy (numactl --interleave, an application's set_mempolicy ...) is put back
   * exactly as it was: this runs on arbitrary streaming threads, once per pool buffer */
  int old_mode = 0;
  unsigned long old_mask[1024 / (8 * sizeof (unsigned long))];
  memset (old_mask, 0, sizeof old_mask);
  const bool saved = syscall (SYS_get_mempolicy, &old_mode, old_mask,
      (unsigned long) (8 * sizeof old_mask), NULL, 0ul) == 0;
  const bool bound = saved && syscall (SYS_set_mempolicy, kMpolPreferred, mask,
      (unsigned long) (8 * sizeof mask)) == 0;
  void *p = NULL;
  const bool bad = hip_failed (hipHostMalloc (&p, bytes ? bytes : 1,
          (bound ? hipHostMallocNumaUser : 0u) | hipHostMallocPortable),
      "hipHostMalloc");
  if (bound)
    (void) syscall (SYS_set_mempolicy, old_mode, old_mask, (unsigned long) (8 * sizeof old_mask));
  if (bad) {
    (void) hipGetLastError ();
    return mibayer_host_alloc (bytes);
  }
  return p;
}

/* NUMA node that holds the first page of `p` (-1 if the kernel does not say) */
extern "C" int mibayer_host_numa_node (const void *p)
{
  if (!p)
    return -1;
  void *page = (void *) ((uintptr_t) p & ~(uintptr_t) 4095);
  int status = -1;
  if (syscall (SYS_move_pages, 0, 1ul, &page, NULL, &status, 0) != 0)
    return -1;
  return status;
}

/* hipHostFree waits for the devices to drain, i.e. it would sit behind a GPU that has stopped answering, typically
 * while a pipeline is shutting down because of it: while a wedge is outstanding the block goes on the deferred list
 * and is returned to the runtime once every wedged device has caught up (polled here and at create / destroy). */
extern "C" void mibayer_host_free (void *p)
{
  if (!p)
    return;
  if (g_wedges_open.load () == 0 && g_deferred_count.load () == 0) {
    (void) hipHostFree (p);
    return;
  }
  std::lock_guard<std::mutex> lk (g_wedge_mu);
  poll_wedges_locked ();
  if (g_wedges_open.load () == 0) {
    (void) hipHostFree (p);
  } else {
    g_deferred_host.push_back (p);
    g_deferred_count.fetch_add (1);
  }
}

/* pinned blocks waiting on the deferred list / contexts whose device has not caught up yet (diagnostics, tests) */
extern "C" int mibayer_deferred_frees (void)
{
  std::lock_guard<std::mutex> lk (g_wedge_mu);
  poll_wedges_locked ();
  return (int) g_deferred_host.size ();
}

extern "C" int mibayer_wedged_contexts (void)
{
  std::lock_guard<std::mutex> lk (g_wedge_mu);
  poll_wedges_locked ();
  return g_wedges_open.load ();
}

extern "C" void *mibayer_device_alloc (mibayer_ctx *c, size_t bytes)
{
  if (!c)
    return NULL;
  DeviceGuard guard (c->device);
  if (!guard.ok)
    return NULL;
  void *p = NULL;
  if (hip_failed (hipMalloc (&p, bytes ? bytes : 1), "hipMalloc"))
    return NULL;
  return p;
}

extern "C" void mibayer_device_free (mibayer_ctx *c, void *d_ptr)
{
  if (!c || !d_ptr)
    return;
  DeviceGuard guard (c->device);
  (void) hipFree (d_ptr);
}

extern "C" int mibayer_copy_to_device (mibayer_ctx *c, void *d_dst,
    const void *src, size_t bytes)
{
  if (!c || !d_dst || !src)
    return MIBAYER_ERR_ARG;
  DeviceGuard guard (c->device);
  if (!guard.ok)
    return MIBAYER_ERR_HIP;
  HIP_TRY (hipMemcpy (d_dst, src, bytes, hipMemcpyHostToDevice));
  return MIBAYER_OK;
}

extern "C" int mibayer_copy_from_device (mibayer_ctx *c, void *dst,
    const void *d_src, size_t bytes)
{
  if (!c || !dst || !d_src)
    return MIBAYER_ERR_ARG;
  DeviceGuard guard (c->device);
  if (!guard.ok)
    return MIBAYER_ERR_HIP;
  HIP_TRY (hipMemcpy (dst, d_src, bytes, hipMemcpyDeviceToHost));
  return MIBAYER_OK;
}

extern "C" void *mibayer_dev_alloc (int device, size_t bytes)
{
  if (device < 0 || device >= device_count_cached ())
    return NULL;
  DeviceGuard guard (device);
  if (!guard.ok)
    return NULL;
  void *p = NULL;
  if (hip_failed (hipMalloc (&p, bytes ? bytes : 1), "hipMalloc"))
    return NULL;
  return p;
}

extern "C" void mibayer_dev_free (int device, void *d_ptr)
{
  if (!d_ptr || device < 0 || device >= device_count_cached ())
    return;
  DeviceGuard guard (device);
  (void) hipFree (d_ptr);
}

extern "C" int mibayer_dev_upload (int device, void *d_dst, const void *src,
    size_t bytes)
{
  if (!d_dst || !src)
    return MIBAYER_ERR_ARG;
  if (device < 0 || device >= device_count_cached ())
    return MIBAYER_ERR_NO_DEVICE;
  DeviceGuard guard (device);
  if (!guard.ok)
    return MIBAYER_ERR_HIP;
  HIP_TRY (hipMemcpy (d_dst, src, bytes, hipMemcpyHostToDevice));
  return MIBAYER_OK;
}

extern "C" int mibayer_dev_download (int device, void *dst, const void *d_src,
    size_t bytes)
{
  if (!dst || !d_src)
    return MIBAYER_ERR_ARG;
  if (device < 0 || device >= device_count_cached ())
    return MIBAYER_ERR_NO_DEVICE;
  DeviceGuard guard (device);
  if (!guard.ok)
    return MIBAYER_ERR_HIP;
  HIP_TRY (hipMemcpy (dst, d_src, bytes, hipMemcpyDeviceToHost));
  return MIBAYER_OK;
}

extern "C" void *mibayer_dev_stream_create (int device)
{
  if (device < 0 || device >= device_count_cached ())
    return NULL;
  DeviceGuard guard (device);
  if (!guard.ok)
    return NULL;
  hipStream_t s = nullptr;
  if (hip_failed (hipStreamCreateWithFlags (&s, hipStreamNonBlocking),
          "hipStreamCreate"))
    return NULL;
  return (void *) s;
}

extern "C" void mibayer_dev_stream_destroy (int device, void *hip_stream)
{
  if (!hip_stream || device < 0 || device >= device_count_cached ())
    return;
  DeviceGuard guard (device);
  (void) hipStreamSynchronize ((hipStream_t) hip_stream);
  (void) hipStreamDestroy ((hipStream_t) hip_stream);
}

extern "C" int mibayer_dev_upload_async (int device, void *d_dst,
    const void *src, size_t bytes, void *hip_stream)
{
  if (!d_dst || !src || !hip_stream)
    return MIBAYER_ERR_ARG;
  if (device < 0 || device >= device_count_cached ())
    return MIBAYER_ERR_NO_DEVICE;
  DeviceGuard guard (device);
  if (!guard.ok)
    return MIBAYER_ERR_HIP;
  Range r ("mibayer:h2d(async upload)");
  HIP_TRY (hipMemcpyAsync (d_dst, src, bytes, hipMemcpyHostToDevice,
          (hipStream_t) hip_stream));
  return MIBAYER_OK;
}

extern "C" int mibayer_dev_download_async (int device, void *dst,
    const void *d_src, size_t bytes, void *hip_stream)
{
  if (!dst || !d_src || !hip_stream)
    return MIBAYER_ERR_ARG;
  if (device < 0 || device >= device_count_cached ())
    return MIBAYER_ERR_NO_DEVICE;
  DeviceGuard guard (device);
  if (!guard.ok)
    return MIBAYER_ERR_HIP;
  Range r ("mibayer:d2h(async download)");
  HIP_TRY (hipMemcpyAsync (dst, d_src, bytes, hipMemcpyDeviceToHost,
          (hipStream_t) hip_stream));
  return MIBAYER_OK;
}

extern "C" int mibayer_dev_event_query (int device, void *event)
{
  if (!event)
    return MIBAYER_ERR_ARG;
  if (device < 0 || device >= device_count_cached ())
    return MIBAYER_ERR_NO_DEVICE;
  DeviceGuard guard (device);
  if (!guard.ok)
    return MIBAYER_ERR_HIP;
  const hipError_t e = hipEventQuery ((hipEvent_t) event);
  if (e == hipSuccess)
    return 1;
  if (e == hipErrorNotReady) {
    (void) hipGetLastError ();
    return 0;
  }
  (void) hip_failed (e, "hipEventQuery");
  return MIBAYER_ERR_HIP;
}

extern "C" void *mibayer_dev_event_create (int device)
{
  if (device < 0 || device >= device_count_cached ())
    return NULL;
  DeviceGuard guard (device);
  if (!guard.ok)
    return NULL;
  hipEvent_t ev = nullptr;
  if (hip_failed (hipEventCreateWithFlags (&ev, hipEventDisableTiming),
          "hipEventCreate"))
    return NULL;
  return (void *) ev;
}

extern "C" void mibayer_dev_event_destroy (int device, void *event)
{
  if (!event || device < 0 || device >= device_count_cached ())
    return;
  DeviceGuard guard (device);
  (void) hipEventDestroy ((hipEvent_t) event);
}

extern "C" int mibayer_dev_event_record (int device, void *event,
    void *hip_stream)
{
  if (!event)
    return MIBAYER_ERR_ARG;
  if (device < 0 || device >= device_count_cached ())
    return MIBAYER_ERR_NO_DEVICE;
  DeviceGuard guard (device);
  if (!guard.ok)
    return MIBAYER_ERR_HIP;
  HIP_TRY (hipEventRecord ((hipEvent_t) event, (hipStream_t) hip_stream));
  return MIBAYER_OK;
}

extern "C" int mibayer_dev_event_wait (int device, void *event)
{
  if (!event)
    return MIBAYER_ERR_ARG;
  if (device < 0 || device >= device_count_cached ())
    return MIBAYER_ERR_NO_DEVICE;
  DeviceGuard guard (device);
  if (!guard.ok)
    return MIBAYER_ERR_HIP;
  HIP_TRY (hipEventSynchronize ((hipEvent_t) event));
  return MIBAYER_OK;
}

extern "C" int mibayer_dev_stream_wait_event (int device, void *hip_stream,
    void *event)
{
  if (!event)
    return MIBAYER_ERR_ARG;
  if (device < 0 || device >= device_count_cached ())
    return MIBAYER_ERR_NO_DEVICE;
  DeviceGuard guard (device);
  if (!guard.ok)
    return MIBAYER_ERR_HIP;
  HIP_TRY (hipStreamWaitEvent ((hipStream_t) hip_stream, (hipEvent_t) event,
          0));
  return MIBAYER_OK;
}

extern "C" int mibayer_fill_synthetic (mibayer_ctx *c, void *d_src,
    size_t src_frame_bytes, uint32_t first_frame, int nframes, uint32_t seed,
    void *hip_stream)
{
  if (!c || !d_src || nframes < 0 || c->inverse)
    return MIBAYER_ERR_ARG;
  if (nframes > 1 && src_frame_bytes < c->src_bytes)
    return MIBAYER_ERR_GEOMETRY;
  DeviceGuard guard (c->device);
  if (!guard.ok)
    return MIBAYER_ERR_HIP;
  hipStream_t s = (hipStream_t) hip_stream;
  if (s == c->s_compute)
    c->dirty_compute = true;
  HIP_TRY (launch_fill_synthetic ((uint8_t *) d_src, c->cfg.width,
          c->cfg.height, c->cfg.src_stride, src_frame_bytes, first_frame,
          nframes, seed, s));
  return MIBAYER_OK;
}
