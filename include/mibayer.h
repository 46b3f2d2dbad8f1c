/*
 * mibayer.h -- C ABI of the MI355X-native bayer2rgb hot path.
 *
 * This is the drop-in boundary for gst-plugins-bad's `bayer2rgb` element
 * (reference v1.19.2, paths relative to the gst-plugins-bad tree):
 *
 *   the element's per-frame call
 *       gst_bayer2rgb_process (filter, output, frame.info.stride[0],
 *                              map.data, GST_ROUND_UP_4 (filter->width));
 *       gst/bayer/gstbayer2rgb.c:475-477  (body :387-451, row helper :354-381,
 *       ORC inner loops gst/bayer/gstbayerorc.orc:3-248)
 *   is replaced by  mibayer_process_host()  /  mibayer_submit()+mibayer_wait(),
 *   and the per-stream state the reference keeps in struct _GstBayer2RGB
 *   (gstbayer2rgb.c:115-127: width, height, r_off, g_off, b_off, format) is
 *   what mibayer_cfg carries into mibayer_create() from the element's
 *   set_caps vfunc (gstbayer2rgb.c:237-276).
 *
 * Plain C, no GLib / GStreamer / torch types.  All entry points return
 * MIBAYER_OK (0) or a negative mibayer_status; nothing throws across the ABI.
 * There is NO CPU fallback behind this ABI: without a HIP device every
 * compute entry point fails with MIBAYER_ERR_NO_DEVICE.
 *
 * Output is bit-exact (uint8) with the reference CPU/ORC path for every
 * geometry in which the reference is well defined: even width >= 4 and
 * height >= 3 (odd widths leave the last column unwritten and read
 * uninitialised scratch in the reference; height < 3 reads uninitialised or
 * out-of-bounds rows -- gstbayer2rgb.c:365-380, :430-447).  Anything else is
 * rejected with MIBAYER_ERR_GEOMETRY at mibayer_create() time.
 */
#ifndef MIBAYER_H
#define MIBAYER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: additive over 1 -- pool failover / fault injection / helper threads,
 * mibayer_process_device_list, mibayer_host_alloc_near + NUMA queries,
 * mibayer_dev_stream_* / mibayer_dev_upload_async / mibayer_dev_event_query;
 * MIBAYER_FLAG_HIPGRAPH now captures the compute-queue segment of a frame.
 * Every v1 entry point keeps its signature and meaning. */
/* 3: additive over 2 -- deadlines on every host-side wait (MIBAYER_ERR_TIMEOUT,
 * mibayer_set_wait_timeout, mibayer_pool_set_wait_timeout); mibayer_sync /
 * mibayer_destroy wait for the context's own frames only. */
/* 4: additive over 3 -- the launch plan as a value (mibayer_get_plan / mibayer_set_plan /
 * mibayer_plan_source), the process-wide plan cache behind mibayer_autotune and
 * mibayer_create (mibayer_plan_cache_clear), mibayer_autotune_list, the host-wait
 * policy (mibayer_set_wait_spin), MIBAYER_FLAG_HIPGRAPH_CHAIN, mibayer_is_lab_build. */
/* NOTE on 4: the numeric kernel-variant ids were renumbered in that version (4 used to be lds_1x8_r4_dpp and became
 * lds_4x2_r4_dpp; the former 20-24 became 4-9; ids >= 10 exist in the lab build only) -- the one non-additive change
 * of the ABI's history.  Every id is bit-exact, so a caller that kept a number got a different speed, not different
 * bytes, or MIBAYER_ERR_ARG for an id the product build no longer has.  Look variants up by NAME
 * (mibayer_variant_name) rather than keeping numbers across versions. */
/* 5: additive over 4 -- one launch plan per LAUNCH CLASS (launches of a few rounds of workgroups: one frame per
 * launch, vs. batch launches): mibayer_get_plan_for / mibayer_set_plan_for, the plan cache keyed by class;
 * mibayer_ctx_frame_queue (four compute queues per device on hardware queues of their own, for independent one-frame
 * launches), covered by mibayer_sync;
 * mibayer_device_pci_bus_id.
 * mibayer_get_plan / mibayer_plan_source / mibayer_ctx_variant_name keep describing the batch-class plan,
 * mibayer_set_plan / mibayer_copy_plan pin every class. */
#define MIBAYER_ABI_VERSION 5

/* MIBAYER ABI GROUPS -- which of the entry points are the drop-in and which are not.  The ABI is
 * FROZEN at version 5: round 6 added nothing.  A maintainer who replaces gst_bayer2rgb_process needs `core` only
 * (INTEGRATION.md section 2 uses create / process_host / destroy / strerror); everything else serves a named
 * consumer and can be ignored by anybody who is not that consumer.  The dynamic symbol table of libmibayer.so is
 * exactly this list (linker version script generated from this header; tests/test_abi.py checks both).
 *
 * group core: the boundary SURVEY.md section 8(b) proposed -- context, one frame host->host (synchronous and queued),
 *   a device-resident batch, pinned host memory
 *     mibayer_abi_version mibayer_device_count mibayer_strerror mibayer_last_hip_error mibayer_create mibayer_destroy
 *     mibayer_get_cfg mibayer_process_host mibayer_submit mibayer_wait mibayer_pending mibayer_process_device
 *     mibayer_sync mibayer_host_alloc mibayer_host_free
 * group sharding: frames round-robin over the GPUs of a node, with failover (north_star; bayer2rgb devices=...)
 *     mibayer_pool_create mibayer_pool_destroy mibayer_pool_capacity mibayer_pool_pending mibayer_pool_submit
 *     mibayer_pool_wait mibayer_pool_alive mibayer_pool_take_failure mibayer_pool_reclaim mibayer_pool_lost
 * group device-memory: what plugin `mihip` (memory:HIPMemory elements) and bench.py are built on --
 *   device allocations, copies, streams and events without a context, list launches, the synthetic-frame generator
 *     mibayer_process_device_list mibayer_ctx_stream mibayer_ctx_frame_queue mibayer_device_alloc mibayer_device_free
 *     mibayer_copy_to_device mibayer_copy_from_device mibayer_dev_alloc mibayer_dev_free mibayer_dev_upload
 *     mibayer_dev_download mibayer_dev_stream_create mibayer_dev_stream_destroy mibayer_dev_upload_async
 *     mibayer_dev_download_async mibayer_dev_event_query mibayer_dev_event_create mibayer_dev_event_destroy
 *     mibayer_dev_event_record mibayer_dev_event_wait mibayer_dev_stream_wait_event mibayer_host_alloc_near
 *     mibayer_host_is_pinned mibayer_device_numa_node mibayer_host_numa_node mibayer_fill_synthetic
 * group tuning: launch plans (measured / cached / set), wait policy and deadlines -- every one of them has a default
 *     mibayer_autotune mibayer_autotune_list mibayer_copy_plan mibayer_get_plan mibayer_set_plan mibayer_get_plan_for
 *     mibayer_set_plan_for mibayer_plan_source mibayer_plan_from_cache mibayer_plan_cache_clear mibayer_time_device
 *     mibayer_set_wait_timeout mibayer_set_wait_spin mibayer_pool_set_wait_timeout mibayer_pool_set_wait_spin
 * group diagnostics: what tests, bench.py and the drills read -- never needed to convert a frame
 *     mibayer_is_lab_build mibayer_device_pci_bus_id mibayer_get_host_stats mibayer_pool_get_host_stats
 *     mibayer_pool_inject_fault mibayer_pool_inject_stall mibayer_deferred_frees mibayer_wedged_contexts
 *     mibayer_variant_count mibayer_variant_name mibayer_auto_variant mibayer_frame_class_variant
 *     mibayer_known_width_plan mibayer_ctx_variant_name mibayer_plan_selectors mibayer_launch_geometry
 *     mibayer_block_to_tile
 * END OF MIBAYER ABI GROUPS */

/* Bayer order; numbering identical to the reference's anonymous enum
 * GST_BAYER_2_RGB_FORMAT_*, gstbayer2rgb.c:95-101. */
typedef enum mibayer_pattern {
  MIBAYER_BGGR = 0,
  MIBAYER_GBRG = 1,
  MIBAYER_GRBG = 2,
  MIBAYER_RGGB = 3
} mibayer_pattern;

typedef enum mibayer_status {
  MIBAYER_OK = 0,
  MIBAYER_ERR_ARG = -1,         /* NULL pointer, bad struct_size, bad enum   */
  MIBAYER_ERR_GEOMETRY = -2,    /* width/height/stride outside the domain    */
  MIBAYER_ERR_LAYOUT = -3,      /* (r,g,b)_off is none of the 4 byte layouts
                                   the reference has merge functions for
                                   (gstbayer2rgb.c:409-421)                  */
  MIBAYER_ERR_NO_DEVICE = -4,   /* no HIP device / ordinal out of range      */
  MIBAYER_ERR_HIP = -5,         /* a HIP runtime call failed                 */
  MIBAYER_ERR_NOMEM = -6,
  MIBAYER_ERR_BUSY = -7,        /* async ring full: call mibayer_wait()      */
  MIBAYER_ERR_EMPTY = -8,       /* mibayer_wait() with nothing in flight     */
  MIBAYER_ERR_TIMEOUT = -9      /* the device did not complete a frame within
                                   the wait deadline (mibayer_set_wait_timeout);
                                   the context is wedged: every later call
                                   returns this at once -- until the device has
                                   caught up with what the context had queued
                                   (polled, never waited for), then the context
                                   works again and its frames can be collected */
} mibayer_status;

/* Per-stream configuration == the fields of struct _GstBayer2RGB
 * (gstbayer2rgb.c:115-127) plus the strides gst_bayer2rgb_transform passes
 * (:475-477) and where to run. */
typedef struct mibayer_cfg {
  uint32_t struct_size;   /* = sizeof (mibayer_cfg)                          */
  int32_t width;          /* pixels; even, >= 4                              */
  int32_t height;         /* rows; >= 3                                      */
  int32_t src_stride;     /* bytes per mosaic row; 0 = GST_ROUND_UP_4(width),
                             the value the reference always passes (:477);
                             must be a multiple of 4 and >= ROUND_UP_4(width) */
  int32_t dst_stride;     /* bytes per output row; 0 = 4*width; multiple of 4,
                             >= 4*width (GstVideoMeta may pad, :476)         */
  int32_t pattern;        /* mibayer_pattern                                 */
  int32_t r_off;          /* byte index of R, G, B inside the 4-byte output  */
  int32_t g_off;          /* pixel == GST_VIDEO_INFO_COMP_OFFSET (info, 0..2) */
  int32_t b_off;          /* (:268-271).  The 4th byte is written as 255.    */
  int32_t device;         /* HIP device ordinal; -1 = the current device     */
  int32_t inflight;       /* host path: frames in flight (ring depth) for
                             mibayer_submit(); 0 = default (2)               */
  int32_t variant;        /* kernel variant; 0 = default.  Tuning knob, every
                             variant is bit-exact            */
  uint32_t flags;         /* MIBAYER_FLAG_* or 0                             */
} mibayer_cfg;

/* Host path: the compute-queue segment of every frame (wait for the upload ->
 * kernel -> signal the download) is captured once per ring slot as a hipGraph
 * and replayed with one hipGraphLaunch per frame; the pinned H2D / D2H copies
 * stay asynchronous copies on the two copy queues (BASELINE.json configs[4]:
 * "pinned double-buffered H2D/D2H + hipGraph-captured launch"). */
#define MIBAYER_FLAG_HIPGRAPH 1u

/* Inverse direction, the plugin's sibling element rgb2bayer (reference
 * gst/bayer/gstrgb2bayer.c:230-278): the source is 4 B/pixel (src_stride
 * default 4*width), the destination the 8-bit mosaic (dst_stride default
 * ROUND_UP_4(width), :179/:255); output byte (j,i) = the R, G or B byte of input
 * pixel (j,i) selected by the CFA site, with (r_off,g_off,b_off) = byte offsets
 * inside the INPUT pixel (the reference accepts ARGB only: 1,2,3).  Any
 * width/height >= 1.  Every host/device/pool entry point works unchanged;
 * variant must be 0; mibayer_autotune / mibayer_fill_synthetic do not apply. */
#define MIBAYER_FLAG_RGB2BAYER 2u

/* With MIBAYER_FLAG_HIPGRAPH: the WHOLE upload -> kernel -> download chain of a
 * ring slot as one graph on the slot's own queue (the host pointers are patched
 * into the instantiated graph per frame) instead of the compute-queue segment
 * only.  Measured slower at 4K (the copies of consecutive frames serialise per
 * slot); kept as the A/B arm bench.py reports. */
#define MIBAYER_FLAG_HIPGRAPH_CHAIN 4u

typedef struct mibayer_ctx mibayer_ctx;

/* ---- global ------------------------------------------------------------- */

int mibayer_abi_version (void);
/* PCI bus id ("0000:c1:00.0", NUL-terminated; len >= 13) of a HIP ordinal: the physical identity of the card a
 * context, pool shard or bench rank runs on (v5). */
int mibayer_device_pci_bus_id (int device, char *out, size_t len);
/* 1 = this library was built with -DMIBAYER_LAB (`make lab`): the experiment
 * kernel arms and the tuning environment variables of DESIGN.md are compiled in.
 * 0 = the product build: the three production tile shapes with their store-policy
 * twins, and no tuning environment. */
int mibayer_is_lab_build (void);
/* number of HIP devices, 0 if none (never negative) */
int mibayer_device_count (void);
const char *mibayer_strerror (int status);
/* text of the last HIP error seen by the calling thread ("" if none) */
const char *mibayer_last_hip_error (void);

/* ---- per-stream context (created in set_caps, destroyed in stop/finalize) -- */

/* Validates cfg, selects the device, creates the compute and copy streams and
 * the device-side frame ring.  Replaces the per-frame g_malloc'ed scratch of
 * the reference (:429) with state that lives as long as the caps do. */
int mibayer_create (const mibayer_cfg *cfg, mibayer_ctx **out);
void mibayer_destroy (mibayer_ctx *ctx);
/* the configuration with every default resolved */
int mibayer_get_cfg (const mibayer_ctx *ctx, mibayer_cfg *out);

/* ---- host-memory frame path (what the GStreamer element calls) ------------ */

/* Synchronous drop-in for gst_bayer2rgb_process (:475-477): `src` holds
 * src_stride*height bytes, `dst` receives dst_stride*height bytes (only the
 * first 4*width bytes of each row are written); returns when dst is complete.
 * H2D, kernel and D2H run on the context's streams.  src/dst may be pageable
 * or pinned (mibayer_host_alloc) memory. */
int mibayer_process_host (mibayer_ctx *ctx, const uint8_t *src, uint8_t *dst);

/* Asynchronous form: up to cfg.inflight frames between submit and wait.
 * Frames complete in submission order; `tag` is handed back by wait.
 * src/dst must stay valid until the matching wait returns. */
int mibayer_submit (mibayer_ctx *ctx, const uint8_t *src, uint8_t *dst,
    void *tag);
int mibayer_wait (mibayer_ctx *ctx, void **tag);
int mibayer_pending (const mibayer_ctx *ctx);
/* Deadline, in milliseconds, of every host-side wait for the device on behalf of
 * this context (mibayer_wait, mibayer_process_host, mibayer_sync, mibayer_destroy):
 * a GPU that has stopped answering completes nothing, and the reference's
 * streaming thread must not hang on it (cf. gst/debugutils/gstwatchdog.c:21-123).
 * Default 10000 (also MIBAYER_WAIT_TIMEOUT_MS); 0 = wait for ever; < 0 = default.
 * A wait that runs into the deadline returns MIBAYER_ERR_TIMEOUT and leaves the
 * frame where it is -- its buffers still belong to the device.  The deadline also
 * bounds mibayer_sync() after device-resident launches: a device that is shared
 * with long-running foreign kernels needs a longer one. */
int mibayer_set_wait_timeout (mibayer_ctx *ctx, int ms);
/* How a host-side wait spends its time.  The completion event is polled: a tight
 * loop for `spin_us` microseconds, then naps that double from 20 us to 250 us.
 * spin_us < 0 (the default; also MIBAYER_WAIT_SPIN_US) = automatic: spin (up to
 * 2 ms) only while the frame waited for is the only one in flight -- the
 * synchronous 1-in/1-out use, where a nap's wake-up latency comes straight off
 * the frame rate -- and nap from the start when other frames are queued behind
 * it (they keep the device busy while the thread sleeps): a few wake-ups per
 * frame instead of a core per streaming thread. */
int mibayer_set_wait_spin (mibayer_ctx *ctx, int spin_us);
/* Host CPU the context has cost its callers so far: calls and CPU time
 * (CLOCK_THREAD_CPUTIME_ID of the calling threads) inside submit and inside the
 * waits, the waits' wall time, hipEventQuery polls and naps. */
typedef struct mibayer_host_stats {
  uint64_t submits;
  uint64_t waits;
  uint64_t polls;
  uint64_t naps;
  double submit_cpu_ms;
  double wait_cpu_ms;
  double wait_wall_ms;
} mibayer_host_stats;
int mibayer_get_host_stats (const mibayer_ctx *ctx, mibayer_host_stats *out);

/* ---- multi-GPU frame sharding (host path) ------------------------------------ */

/* Frames are independent (the reference keeps no state between frames,
 * gstbayer2rgb.c:387-451), so a stream is sharded round-robin over GPUs with no
 * collective: frame g goes to shard g % ndevices, every shard is a mibayer_ctx
 * with its own device ring and (optionally) graphs, and results are handed back
 * in submission order.  By default the calling thread drives all shards -- with
 * pinned buffers every call below only enqueues work -- and a shard that sees
 * pageable buffers gets a helper thread (queueing a 4K frame from pinned buffers
 * costs the caller 14-80 us of CPU, from pageable ones 200-300 us: the runtime's
 * staging copy).  One thread saturates at ~11 k frames/s, what eight PCIe links
 * carry at 4K: pools over six or more DISTINCT GPUs give every shard its own
 * submit thread from the start; MIBAYER_POOL_THREADS=1 / 0 forces that on / off.  When the devices span more than
 * one NUMA node, a frame goes to the live shard next to its 4-byte-per-pixel
 * buffer as long as that keeps the rotation balanced (MIBAYER_POOL_NUMA=0:
 * strictly g % ndevices).  Ordinals may repeat (N logical shards on one GPU). */
#define MIBAYER_MAX_SHARDS 16
typedef struct mibayer_pool_cfg {
  uint32_t struct_size;         /* = sizeof (mibayer_pool_cfg)                  */
  mibayer_cfg stream;           /* geometry, order, layout, inflight PER SHARD,
                                   flags; .device is ignored                    */
  int32_t ndevices;             /* 1 .. MIBAYER_MAX_SHARDS                      */
  int32_t devices[MIBAYER_MAX_SHARDS];  /* HIP ordinals                         */
} mibayer_pool_cfg;

typedef struct mibayer_pool mibayer_pool;

int mibayer_pool_create (const mibayer_pool_cfg *cfg, mibayer_pool **out);
void mibayer_pool_destroy (mibayer_pool *pool);
/* total frames that may be in flight = live devices * stream.inflight (shrinks
 * when a device is dropped, see below) */
int mibayer_pool_capacity (const mibayer_pool *pool);
int mibayer_pool_pending (const mibayer_pool *pool);
/* MIBAYER_ERR_BUSY when the shard whose turn it is has its share of frames in
 * flight: call mibayer_pool_wait() first.  Frames whose src or dst is pageable
 * memory are converted by a helper thread of their shard (the copies block the
 * thread that issues them), so this call only queues either way. */
int mibayer_pool_submit (mibayer_pool *pool, const uint8_t *src, uint8_t *dst,
    void *tag);
/* oldest frame first */
int mibayer_pool_wait (mibayer_pool *pool, void **tag);

/* Failure handling.  The reference's only reaction to a failure on this path is
 * "warn and carry on" (gstbayer2rgb.c:484-486).  A device that reports a HIP
 * error (or runs out of memory) is dropped from the rotation, the frames it
 * still held are converted again on the surviving devices -- same bytes, same
 * order -- and the stream carries on; submit / wait return MIBAYER_ERR_HIP only
 * when no device is left.  mibayer_pool_take_failure() returns the number of
 * devices dropped since the last call (0 = none) with the ordinal of the latest
 * one, the number of devices left and a one-line description. */
int mibayer_pool_alive (const mibayer_pool *pool);
int mibayer_pool_take_failure (mibayer_pool *pool, int *device, int *alive,
    char *msg, size_t msg_len);
/* Failure drill: shard `shard` (index into devices[]) reports a device error
 * once it has completed `after_frames` more frames; negative = cancel.  The
 * environment variable MIBAYER_INJECT_FAULT="shard:frames[,shard:frames]" does
 * the same at mibayer_pool_create(). */
int mibayer_pool_inject_fault (mibayer_pool *pool, int shard,
    long long after_frames);
/* The wait deadline of every shard (mibayer_set_wait_timeout): a device that
 * does not complete a frame within `ms` milliseconds is dropped from the rotation
 * like one that reported an error, and is never waited for again. */
int mibayer_pool_set_wait_timeout (mibayer_pool *pool, int ms);
int mibayer_pool_set_wait_spin (mibayer_pool *pool, int spin_us);
/* sum over the shards */
int mibayer_pool_get_host_stats (const mibayer_pool *pool, mibayer_host_stats *out);
/* Frames that were IN FLIGHT on a device when it ran into the wait deadline are
 * not converted again behind the device's back: the device may only be slow, its
 * queued copies may still read the frame's source and write its destination
 * later.  mibayer_pool_wait() hands such a frame back -- in order, with its tag --
 * with MIBAYER_ERR_TIMEOUT: the frame is LOST (drop it), the stream carries on on
 * the remaining devices (mibayer_pool_alive() == 0: it has failed), and both
 * buffers of the frame stay the device's: do not free, unmap or reuse them until
 * mibayer_pool_reclaim() hands the tag back -- it does once the device has caught
 * up with everything it had queued (non-blocking; MIBAYER_ERR_EMPTY = none ready).
 * mibayer_pool_lost() = lost frames not reclaimed yet.  Buffers still lost when
 * the pool is destroyed must be leaked.  Frames the dropped device had not been
 * given yet are converted on the others as after a device error. */
int mibayer_pool_reclaim (mibayer_pool *pool, void **tag);
int mibayer_pool_lost (const mibayer_pool *pool);
/* Stall drill: the compute queue of shard `shard` is occupied for `ms`
 * milliseconds (1 .. 5000) by a kernel that only waits -- a device that has
 * stopped answering, as far as the host can tell. */
int mibayer_pool_inject_stall (mibayer_pool *pool, int shard, int ms);

/* ---- device-resident batch path (roofline runs, GPU-side consumers) -------- */

/* Enqueues ONE kernel launch converting `nframes` frames that already live in
 * device memory: frame f is read at d_src + f*src_frame_bytes and written at
 * d_dst + f*dst_frame_bytes.  `hip_stream` is the hipStream_t to launch on,
 * used as given (NULL is HIP's null stream); pass mibayer_ctx_stream() for the
 * context's own compute stream.  Returns after the launch is enqueued. */
int mibayer_process_device (mibayer_ctx *ctx, const void *d_src,
    size_t src_frame_bytes, void *d_dst, size_t dst_frame_bytes, int nframes,
    void *hip_stream);
/* The same for frames that are SEPARATE device allocations (one GstBuffer each):
 * frame f is read at d_srcs[f] and written at d_dsts[f]; up to 16 frames go into
 * one kernel launch (more are split).  For device-resident pipelines, where a
 * launch per 4K frame costs as much as the kernel runs (gst/gstmihipelements.c,
 * hipbayer2rgb batch=N). */
int mibayer_process_device_list (mibayer_ctx *ctx, const void *const *d_srcs,
    void *const *d_dsts, int nframes, void *hip_stream);
/* the context's compute stream (a hipStream_t), created non-blocking */
void *mibayer_ctx_stream (mibayer_ctx *ctx);
/* Frame queues (v5): MIBAYER_FRAME_QUEUES compute streams per DEVICE, each on a hardware queue of its own, shared by
 * the contexts of the device, created on first use (NULL: k out of range, or no stream could be made) and kept for the
 * life of the process (destroying such a stream hangs intermittently in ROCm 7.2).  A launch over
 * ONE frame is a single round of workgroups -- ramp-up, one burst of loads, one burst of stores, drain: 4K 9.4 us
 * against 6.5 us per frame inside a batch -- so a caller that converts frame after frame (one GstBuffer at a time) and
 * whose frames do not depend on each other deals them round-robin over the frame queues: the ramp-up of the next
 * frames overlaps the drain of frame n (4K: 54 % of HBM peak on one queue, 66-67 % on four; rgb2bayer 55 -> 77 %).
 * That is for launches WITHOUT per-frame dependencies on other queues: a frame that must wait for an event of its
 * producer's queue and is waited for by its consumer's pays more for the cross-queue dependencies than the overlap
 * gains (hipbayer2rgb from a device-resident source: 14.5 k fps with the frame queues, 36 k without -- the elements'
 * `overlap` property is therefore off by default).  Ordinary HIP streams share a small pool of hardware queues and serialise
 * behind each other again; these do not.  The queues are not ordered against each other or against
 * mibayer_ctx_stream(): order consumers with events (mibayer_dev_event_record / mibayer_dev_stream_wait_event) or with
 * mibayer_sync(), which covers every frame queue this context launched on.
 * One exception to "not ordered" (ADVICE r05): hipExtStreamCreateWithCUMask takes no flags, so these are BLOCKING
 * streams in the runtime's sense -- they synchronise implicitly with HIP's legacy NULL stream, in both directions,
 * unlike every other stream of this library (hipStreamNonBlocking), and unlike the plain non-blocking streams that
 * stand in for them when the CU-mask call fails.  A process that launches on the NULL stream (or on a framework's
 * "default stream" that is the NULL stream) serialises frame-queue launches against it: drive the frame queues from
 * processes whose other work is on explicit streams, as the elements and tools/single_frame_bench.py do. */
#define MIBAYER_FRAME_QUEUES 4
void *mibayer_ctx_frame_queue (mibayer_ctx *ctx, int k);
/* Waits (with the context's deadline) for what THIS context has in flight: its
 * pending host-path frames and the device-resident work queued through
 * mibayer_process_device[_list] / mibayer_fill_synthetic on mibayer_ctx_stream() and
 * mibayer_process_device[_list] on the frame queues (mibayer_ctx_frame_queue).
 * Work a caller put on that stream by other means (its own kernels, copies) is
 * NOT covered -- the queue may be shared with the other contexts of the device --
 * and neither is work on a caller-supplied stream: synchronise those yourself. */
int mibayer_sync (mibayer_ctx *ctx);

/* Times `reps` back-to-back launches of mibayer_process_device on the context's
 * compute stream with HIP events recorded on that stream (after `warmup` untimed
 * launches); writes the mean milliseconds per launch. */
int mibayer_time_device (mibayer_ctx *ctx, const void *d_src,
    size_t src_frame_bytes, void *d_dst, size_t dst_frame_bytes, int nframes,
    int warmup, int reps, float *ms_per_launch);

/* Measured launch-plan selection for the device-resident path.  MI355X boxes --
 * and allocations -- differ in which block->tile order streams best (DESIGN.md
 * "XCD map"): this call times the candidate plans on the caller's own buffers with
 * HIP events on the context's compute stream and keeps the fastest for every later
 * launch of this context THAT FALLS INTO THE SAME LAUNCH CLASS as the `nframes`-frame launches it timed (v5: launch
 * classes, below; a plan measured on 64-frame batches does not become the plan of one-frame launches).  Candidates: the three production tile shapes x {band 1,
 * one chunk per XCD, identity order}; for generic geometries whose output rows sit
 * off the 64-byte sector grid additionally x {streaming, write-back, hybrid}
 * stores, plus the shifted arm (every wave-store on a 128-byte boundary) in the two
 * narrow shapes.  A common time-based warm-up (>= 60 ms of launches), then five
 * interleaved rounds of a few launches per candidate; with more than nine
 * candidates only those within 6 % of the best of the first round stay in; the
 * MEDIAN round counts.  The kernel is idempotent, so d_dst holds the correct output
 * afterwards.  Synchronous.  `report` (may be NULL) receives a one-line summary. */
int mibayer_autotune (mibayer_ctx *ctx, const void *d_src,
    size_t src_frame_bytes, void *d_dst, size_t dst_frame_bytes, int nframes,
    char *report, size_t report_len);

/* The same over frames that are SEPARATE device allocations, timed through
 * mibayer_process_device_list (what a device-resident element holds: one GstBuffer
 * per frame). */
int mibayer_autotune_list (mibayer_ctx *ctx, const void *const *d_srcs,
    void *const *d_dsts, int nframes, char *report, size_t report_len);

/* Copies the launch plan chosen by mibayer_autotune() to another context of the
 * same width/height (e.g. the same camera geometry in another Bayer order). */
int mibayer_copy_plan (mibayer_ctx *dst, const mibayer_ctx *src);

/* The launch plan as a value: kernel variant id (1 .. mibayer_variant_count()-1),
 * block order (tile rows per XCD band: 1, N; 0 = identity; -1 = one chunk of the
 * batch per XCD; INT32_MIN = the variant's default) and the store alignment of the
 * shifted arm for generic geometries (0 = off, 64, 128).  Any out pointer may be
 * NULL.  Every plan is bit-exact; a plan only changes speed. */
int mibayer_get_plan (const mibayer_ctx *ctx, int *variant, int *band,
    int *align_stores);
int mibayer_set_plan (mibayer_ctx *ctx, int variant, int band, int align_stores);
/* Launch classes (v5).  A context keeps one plan per class of launch:
 *   batch class -- launches that keep every workgroup slot of the device busy for many rounds (the 64-frame batch);
 *   frame class -- launches of at most 4 rounds of workgroups (on 256 CUs: up to 33.5 Mpixel per launch = one 8K
 *                  frame, four 4K frames): what an element or the host path issues, one frame per launch.  Its default
 *                  shape is the production shape whose grid needs the FEWEST ROUNDS of the device's workgroup slots
 *                  (a 4K frame in 1024x8 tiles is 1080 workgroups on 1024 slots -- a second round that is 5 % full --
 *                  in 256x32 tiles 1020: 54.8 vs 47.9 % of peak frame by frame), the widest among equals.
 * Which class a launch over `nframes` frames falls into follows from the geometry alone.  mibayer_autotune[_list]
 * measures -- and the plan cache records -- the class of the launches it was given; mibayer_get_plan,
 * mibayer_plan_source and mibayer_ctx_variant_name describe the batch class; mibayer_set_plan / mibayer_copy_plan pin
 * every class.  The two below address the class of an `nframes`-frame launch; `source` receives MIBAYER_PLAN_*. */
int mibayer_get_plan_for (const mibayer_ctx *ctx, int nframes, int *variant, int *band, int *align_stores,
    int *source);
int mibayer_set_plan_for (mibayer_ctx *ctx, int nframes, int variant, int band, int align_stores);

/* Process-wide plan cache.  mibayer_autotune / mibayer_autotune_list record what
 * they measured under (device, width, height, src_stride, dst_stride, launch class);
 * mibayer_create of a stream with that geometry on that device (cfg.variant == 0)
 * starts from the recorded plan instead of the static default -- the second element
 * instance, the context after a renegotiation, the other three Bayer orders of one
 * camera do not measure again.  MIBAYER_PLAN_CACHE=0 in the environment turns
 * look-ups and recording off.  Reference analogue (set up once per process, reuse):
 * the once-guarded ORC programs, gst/bayer/gstbayerorc-dist.c:321-397. */
#define MIBAYER_PLAN_DEFAULT 0  /* the static per-geometry default of mibayer_create */
#define MIBAYER_PLAN_MEASURED 1 /* mibayer_autotune[_list] ran on this context       */
#define MIBAYER_PLAN_CACHED 2   /* taken from the process plan cache at create       */
#define MIBAYER_PLAN_SET 3      /* mibayer_set_plan / mibayer_copy_plan               */
int mibayer_plan_source (const mibayer_ctx *ctx);
/* Looks the context's geometry up in the cache again (a plan may have been measured
 * since the context was created): 1 = a cached plan was applied, 0 = none. */
int mibayer_plan_from_cache (mibayer_ctx *ctx);
void mibayer_plan_cache_clear (void);

/* ---- memory helpers --------------------------------------------------------- */

/* Pinned (hipHostMalloc) memory for buffer pools feeding the host path. */
void *mibayer_host_alloc (size_t bytes);
/* (while a context of this process is wedged -- a wait ran into its deadline and
 * the device has not caught up since -- the block goes on a deferred list instead
 * of hipHostFree, which would wait for that device; the list is returned to the
 * runtime as soon as no wedge is outstanding) */
void mibayer_host_free (void *p);
/* diagnostics: blocks on the deferred list / contexts whose device has not caught
 * up after a wait deadline (both poll, neither blocks) */
int mibayer_deferred_frees (void);
int mibayer_wedged_contexts (void);
/* The same, placed on the NUMA node next to HIP device `device` (the thread's
 * memory policy is set around a hipHostMallocNumaUser allocation): a pool that
 * feeds GPU k should not sit behind the socket link.  Falls back to
 * mibayer_host_alloc() when the node is unknown.  Free with mibayer_host_free(). */
void *mibayer_host_alloc_near (int device, size_t bytes);
/* 1 when p points into pinned / registered host memory (an asynchronous copy
 * from / to it really is asynchronous), 0 for ordinary pageable memory, where
 * hipMemcpyAsync blocks the caller while the runtime stages the data */
int mibayer_host_is_pinned (const void *p);
/* NUMA node next to a device (-1 = unknown) / holding the first page of p */
int mibayer_device_numa_node (int device);
int mibayer_host_numa_node (const void *p);

/* Device memory on the context's device, for C callers without another
 * allocator. */
void *mibayer_device_alloc (mibayer_ctx *ctx, size_t bytes);
void mibayer_device_free (mibayer_ctx *ctx, void *d_ptr);
int mibayer_copy_to_device (mibayer_ctx *ctx, void *d_dst, const void *src,
    size_t bytes);
int mibayer_copy_from_device (mibayer_ctx *ctx, void *dst, const void *d_src,
    size_t bytes);

/* Context-free device memory helpers (for a GstAllocator of device memory, the
 * `memory:HIPMemory` caps feature of gst/gstmihipmemory.c): synchronous. */
void *mibayer_dev_alloc (int device, size_t bytes);
void mibayer_dev_free (int device, void *d_ptr);
int mibayer_dev_upload (int device, void *d_dst, const void *src, size_t bytes);
int mibayer_dev_download (int device, void *dst, const void *d_src,
    size_t bytes);

/* Context-free copy queue and asynchronous upload, for an uploader that does
 * not want to wait for its DMA (gst/gstmihipelements.c, hipupload: the host
 * buffer is kept alive until the event recorded after the copy has fired, the
 * pattern of the reference tree's sys/nvcodec/gstcudaupload.c made
 * asynchronous).  `src` must stay valid until an event recorded on the stream
 * after the call has completed; from pageable memory the call itself blocks. */
void *mibayer_dev_stream_create (int device);
void mibayer_dev_stream_destroy (int device, void *hip_stream);
int mibayer_dev_upload_async (int device, void *d_dst, const void *src,
    size_t bytes, void *hip_stream);
/* the other direction (hipdownload's queued mode): `dst` is complete once an
 * event recorded on the stream after the call has fired */
int mibayer_dev_download_async (int device, void *dst, const void *d_src,
    size_t bytes, void *hip_stream);
/* 1 = the work recorded in the event has completed, 0 = not yet, < 0 = error */
int mibayer_dev_event_query (int device, void *event);

/* Context-free events, for stream-ordered hand-over of a device buffer from one
 * user to the next without a host round trip: record after the work that touches
 * the buffer, make the next user's stream wait for it -- or block the host on it.
 * (Plugin `mihip` records them lazily: one per stream when somebody has to wait
 * across queues or on the host, not one per buffer -- gst/gstmihipmemory.h.) */
void *mibayer_dev_event_create (int device);
void mibayer_dev_event_destroy (int device, void *event);
int mibayer_dev_event_record (int device, void *event,
    void *hip_stream /* used as given */);
int mibayer_dev_event_wait (int device, void *event);          /* host blocks */
int mibayer_dev_stream_wait_event (int device, void *hip_stream, void *event);

/* Counter-based synthetic mosaic generated on the device (stateless per byte;
 * definition in DESIGN.md "Synthetic input"): frames first_frame ..
 * first_frame+nframes-1 with the context's width/height/src_stride. */
int mibayer_fill_synthetic (mibayer_ctx *ctx, void *d_src,
    size_t src_frame_bytes, uint32_t first_frame, int nframes, uint32_t seed,
    void *hip_stream /* used as given, like mibayer_process_device */);

/* ---- introspection (tests) --------------------------------------------------- */

/* Number of kernel variants; variant ids are 0 .. n-1.  0 = "auto": the
 * production tile shape chosen from the stream width at mibayer_create(). */
int mibayer_variant_count (void);
const char *mibayer_variant_name (int variant);
/* the variant id "auto" (0) resolves to for a frame width by the padding rule (pure host arithmetic) */
int mibayer_auto_variant (int width);
/* ... for a launch of ONE frame of a sector-aligned geometry on a device with `compute_units` CUs: the production shape
 * whose grid needs the fewest rounds of the device's workgroup slots (4 per CU), the widest tile among equals (v5) */
int mibayer_frame_class_variant (int width, int height, int compute_units);
/* ... and the batch-class default of the common sensor widths whose measured winner the padding rule does not find:
 * 1 and (variant, band) if `width` is one of them, else 0 (v5) */
int mibayer_known_width_plan (int width, int *variant, int *band);
/* name of the concrete variant the context resolved to */
const char *mibayer_ctx_variant_name (const mibayer_ctx *ctx);
/* Pure host arithmetic, no device needed: the four v_perm_b32 selectors (output
 * pixel k = perm ({R'B' pair word, G word}, sel[k])) and the row-type swap that
 * mibayer_create() derives from (pattern, r_off, g_off, b_off) -- the kernel's
 * replacement for the reference's merge-function table and pointer swaps
 * (gstbayer2rgb.c:400-427).  Lets the byte-placement logic be checked on a CPU. */
int mibayer_plan_selectors (const mibayer_cfg *cfg, uint32_t sel[4],
    int *swap_rows);

/* Launch geometry the context would use for nframes frames: tile size in
 * pixels, tiles per tile row, tile rows in the batch (nframes * tiles_y), the
 * XCD band (tile rows per XCD band, 0 = identity map) and the grid size.  Any
 * pointer may be NULL. */
int mibayer_launch_geometry (const mibayer_ctx *ctx, int nframes, int *tile_w,
    int *tile_h, int *tiles_x, int64_t *tile_rows, int *band,
    int64_t *grid_blocks);
/* The XCD-aware block -> tile permutation used by the kernel, evaluated on the
 * host: linear tile id (tile_row * tiles_x + tx) that block `block` processes,
 * or -1 if that block idles. */
int64_t mibayer_block_to_tile (int64_t block, int tiles_x, int64_t tile_rows,
    int band);

#ifdef __cplusplus
}
#endif
#endif /* MIBAYER_H */
