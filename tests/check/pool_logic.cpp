/* TEST INFRASTRUCTURE.  Drives the REAL frame-sharding pool (gst-plugins-bad_amd/csrc/mibayer_pool.cpp) over the
 * test double of the per-device contexts (tests/check/mock_mibayer.c), built by tests/test_pool_logic.py with
 * AddressSanitizer and, separately, ThreadSanitizer.  No GPU, no conversion: the double stamps every destination
 * frame with the first source byte, so "every frame exactly once, in order, converted from ITS source" is checkable.
 *
 *   pool_logic <shards> <inflight> <frames> <pageable 0|1> <fault list "shard:after,..." or -> <expect ok|dead>
 *
 * Prints: delivered=<n> dropped_devices=<n> alive=<n> capacity=<n> rc=<last status>
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "mibayer.h"

static const int W = 64, H = 8;

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
  for (int i = 0; i < nshards; i++)
    pc.devices[i] = 0;
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

  const size_t src_bytes = (size_t) W * H, dst_bytes = (size_t) 4 * W * H;
  /* exact-size heap buffers per frame: the sanitizer sees every out-of-bounds or use-after-free */
  std::vector<uint8_t *> srcs ((size_t) nframes), dsts ((size_t) nframes);
  for (int f = 0; f < nframes; f++) {
    srcs[(size_t) f] = (uint8_t *) malloc (src_bytes);
    dsts[(size_t) f] = (uint8_t *) malloc (dst_bytes);
    memset (srcs[(size_t) f], 1 + f % 250, src_bytes);
    memset (dsts[(size_t) f], 0, dst_bytes);
  }

  int submitted = 0, delivered = 0, dropped = 0, last = MIBAYER_OK;
  bool dead = false;
  auto collect = [&]() -> bool {
    void *tag = NULL;
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
    if (r != MIBAYER_OK) {
      last = r;
      return false;
    }
    /* oldest first, and converted from its own source */
    if ((intptr_t) tag != delivered + 1)
      exit (11);
    const uint8_t *d = dsts[(size_t) delivered];
    for (size_t k = 4; k < dst_bytes; k++)
      if (d[k] != (uint8_t) (1 + delivered % 250))
        exit (12);
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
  printf ("delivered=%d dropped_devices=%d alive=%d capacity=%d rc=%d\n", delivered, dropped,
      mibayer_pool_alive (pool), mibayer_pool_capacity (pool), last);
  mibayer_pool_destroy (pool);
  for (int f = 0; f < nframes; f++) {
    free (srcs[(size_t) f]);
    free (dsts[(size_t) f]);
  }
  if (expect_dead)
    return (dead && last == MIBAYER_ERR_HIP) ? 0 : 20;
  return (!dead && delivered == nframes) ? 0 : 21;
}
