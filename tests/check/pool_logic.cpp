/* TEST INFRASTRUCTURE.  Drives the REAL frame-sharding pool (gst-plugins-bad_amd/csrc/mibayer_pool.cpp) over the
 * test double of the per-device contexts (tests/check/mock_mibayer.c), built by tests/test_pool_logic.py with
 * AddressSanitizer and, separately, ThreadSanitizer.  No GPU, no conversion: the double stamps every destination
 * frame with the first source byte, so "every frame exactly once, in order, converted from ITS source" is checkable.
 *
 *   pool_logic <shards> <inflight> <frames> <pageable 0|1> <fault list "shard:after,..." or -> <expect ok|dead>
 *
 * Environment: POOL_LOGIC_DISTINCT=1 gives shard i the device ordinal i (the double needs MOCK_MIBAYER_DEVICES >= shards);
 * POOL_LOGIC_NEAR=1 allocates frame f's buffers with mibayer_host_alloc_near (devices[f % shards]) -- what the
 * element's pinned pool does -- so the NUMA-local routing has something to look at; POOL_LOGIC_TIMEOUT_MS sets the
 * pool's wait deadline; POOL_LOGIC_STALL="shard" calls mibayer_pool_inject_stall on that shard after a few frames.
 * Every frame's buffers are freed the moment it has been delivered and checked: anything that touches them later
 * (a DMA of a context that should have been abandoned first) is a sanitizer report.  A frame handed back as LOST
 * (MIBAYER_ERR_TIMEOUT: it was in flight on a device that ran into the wait deadline) keeps its buffers until
 * mibayer_pool_reclaim returns its tag -- the double's stalled device writes them late -- and the ones still lost
 * at the end are released only after the pool is gone.
 *
 * Prints: delivered=<n> lost=<n> reclaimed=<n> dropped_devices=<n> alive=<n> capacity=<n> rc=<last status> local=<n> remote=<n>
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "mibayer.h"

static const int W = 64, H = 8;

extern "C" void mock_numa_counts (int *local, int *remote);

int
main (int argc, char **argv)
{
  if (argc < 7) {
    fprintf (stderr, "usage: see the header of pool_logic.cpp\n");
    return 64;
  }
  const int nshards = atoi (argv[1]), inflight = atoi (argv[2]), nframes = atoi (argv[3]);
  const bool pageable = atoi (argv[4]) != 0;
  const char *faults = argv[5];
  const bool expect_dead = strcmp (argv[6], "dead") == 0;

  if (pageable)
    setenv ("MOCK_MIBAYER_PAGEABLE", "1", 1);
  mibayer_pool_cfg pc;
  memset (&pc, 0, sizeof pc);
  pc.struct_size = sizeof pc;
  pc.stream.struct_size = sizeof pc.stream;
  pc.stream.width = W;
  pc.stream.height = H;
  pc.stream.pattern = MIBAYER_RGGB;
  pc.stream.r_off = 0;
  pc.stream.g_off = 1;
  pc.stream.b_off = 2;
  pc.stream.inflight = inflight;
  pc.ndevices = nshards;
  const bool distinct = getenv ("POOL_LOGIC_DISTINCT") != NULL;
  const bool near_alloc = getenv ("POOL_LOGIC_NEAR") != NULL;
  for (int i = 0; i < nshards; i++)
    pc.devices[i] = distinct ? i : 0;
  mibayer_pool *pool = NULL;
  int rc = mibayer_pool_create (&pc, &pool);
  if (rc != MIBAYER_OK) {
    fprintf (stderr, "create failed %d\n", rc);
    return 2;
  }
  /* "shard:after[,shard:after]": device error once the shard has completed `after` frames */
  for (const char *e = faults; *e && *e != '-';) {
    char *end = NULL;
    const long s = strtol (e, &end, 10);
    if (*end != ':')
      return 64;
    const long long n = strtoll (end + 1, &end, 10);
    if (mibayer_pool_inject_fault (pool, (int) s, n) != MIBAYER_OK)
      return 64;
    e = *end == ',' ? end + 1 : end;
  }

  if (const char *e = getenv ("POOL_LOGIC_TIMEOUT_MS"))
    if (mibayer_pool_set_wait_timeout (pool, atoi (e)) != MIBAYER_OK)
      return 64;
  const int stall_shard = getenv ("POOL_LOGIC_STALL") ? atoi (getenv ("POOL_LOGIC_STALL")) : -1;
  const size_t src_bytes = (size_t) W * H, dst_bytes = (size_t) 4 * W * H;
  /* exact-size heap buffers per frame: the sanitizer sees every out-of-bounds or use-after-free */
  std::vector<uint8_t *> srcs ((size_t) nframes), dsts ((size_t) nframes);
  for (int f = 0; f < nframes; f++) {
    if (near_alloc) {
      srcs[(size_t) f] = (uint8_t *) mibayer_host_alloc_near (pc.devices[f % nshards], src_bytes);
      dsts[(size_t) f] = (uint8_t *) mibayer_host_alloc_near (pc.devices[f % nshards], dst_bytes);
    } else {
      srcs[(size_t) f] = (uint8_t *) malloc (src_bytes);
      dsts[(size_t) f] = (uint8_t *) malloc (dst_bytes);
    }
    memset (srcs[(size_t) f], 1 + f % 250, src_bytes);
    memset (dsts[(size_t) f], 0, dst_bytes);
  }

  int submitted = 0, delivered = 0, dropped = 0, last = MIBAYER_OK;
  int handed = 0;               /* frames handed back so far, delivered or lost: the next tag is handed + 1 */
  int lost = 0, reclaimed = 0;
  bool dead = false;
  auto release = [&](int f) {
    if (near_alloc) {
      mibayer_host_free (srcs[(size_t) f]);
      mibayer_host_free (dsts[(size_t) f]);
    } else {
      free (srcs[(size_t) f]);
      free (dsts[(size_t) f]);
    }
    srcs[(size_t) f] = dsts[(size_t) f] = NULL;
  };
  auto reclaim = [&]() {
    void *t = NULL;
    while (mibayer_pool_reclaim (pool, &t) == MIBAYER_OK) {
      const int f = (int) (intptr_t) t - 1;
      if (f < 0 || f >= nframes || srcs[(size_t) f] == NULL)
        exit (15);              /* a tag that was never lost, or twice */
      /* the stalled device has written it after all: the stamp of ITS source */
      for (size_t k = 4; k < dst_bytes; k++)
        if (dsts[(size_t) f][k] != (uint8_t) (1 + f % 250))
          exit (16);
      release (f);
      reclaimed++;
    }
  };
  auto collect = [&]() -> bool {
    void *tag = NULL;
    reclaim ();
    const int r = mibayer_pool_wait (pool, &tag);
    char msg[300];
    int dev = -1, alive = -1;
    const int nf = mibayer_pool_take_failure (pool, &dev, &alive, msg, sizeof msg);
    if (nf > 0) {
      dropped += nf;
      fprintf (stderr, "note: %s\n", msg);
      if (alive != mibayer_pool_alive (pool))
        exit (10);
    }
    if (r == MIBAYER_ERR_TIMEOUT && tag != NULL && mibayer_pool_alive (pool) > 0) {
      /* lost on a device that stopped answering: in order like any other frame, buffers kept */
      if ((intptr_t) tag != handed + 1)
        exit (11);
      if (mibayer_pool_lost (pool) < 1)
        exit (17);
      handed++;
      lost++;
      return true;
    }
    if (r != MIBAYER_OK) {
      last = r;
      return false;
    }
    /* oldest first, and converted from its own source */
    if ((intptr_t) tag != handed + 1)
      exit (11);
    const uint8_t *d = dsts[(size_t) handed];
    for (size_t k = 4; k < dst_bytes; k++)
      if (d[k] != (uint8_t) (1 + handed % 250))
        exit (12);
    /* handed back: the caller may release the buffers now */
    release (handed);
    handed++;
    delivered++;
    return true;
  };

  while (submitted < nframes && !dead) {
    rc = mibayer_pool_submit (pool, srcs[(size_t) submitted], dsts[(size_t) submitted],
        (void *) (intptr_t) (submitted + 1));
    if (rc == MIBAYER_ERR_BUSY) {
      if (mibayer_pool_pending (pool) == 0)
        exit (13);              /* busy with nothing in flight */
      if (!collect ())
        dead = true;
      continue;
    }
    if (rc != MIBAYER_OK) {
      last = rc;
      dead = true;
      break;
    }
    submitted++;
    if (submitted == 3 && stall_shard >= 0 && mibayer_pool_inject_stall (pool, stall_shard, 1000) != MIBAYER_OK)
      return 64;
    if (mibayer_pool_pending (pool) > mibayer_pool_capacity (pool) + inflight * nshards)
      exit (14);
  }
  while (!dead && mibayer_pool_pending (pool) > 0)
    if (!collect ())
      dead = true;
  {
    char msg[300];
    int dev, alive;
    dropped += mibayer_pool_take_failure (pool, &dev, &alive, msg, sizeof msg);
  }
  for (int k = 0; k < 8; k++)
    reclaim ();                 /* the double's stalled devices resume after a few polls */
  int local = 0, remote = 0;
  mock_numa_counts (&local, &remote);
  printf ("delivered=%d lost=%d reclaimed=%d dropped_devices=%d alive=%d capacity=%d rc=%d local=%d remote=%d\n",
      delivered, lost, reclaimed, dropped, mibayer_pool_alive (pool), mibayer_pool_capacity (pool), last, local,
      remote);
  mibayer_pool_destroy (pool);  /* a device that never resumed does so now, at the latest: into buffers still held */
  for (int f = 0; f < nframes; f++)
    if (srcs[(size_t) f] != NULL)
      release (f);
  if (expect_dead)
    return (dead && (last == MIBAYER_ERR_HIP || last == MIBAYER_ERR_TIMEOUT)) ? 0 : 20;
  return (!dead && delivered + lost == nframes) ? 0 : 21;
}
