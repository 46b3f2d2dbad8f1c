// Does the ALIGNMENT / physical contiguity of the frame buffers decide the per-process state of the block orders
// (DESIGN.md "per-process state")?  The amdgpu page tables mark a run of pages as one "fragment" -- one TLB entry -- as far
// as virtual alignment and physical contiguity allow; hipMalloc aligns to 2 MiB.  This probe allocates the same 4K x 64
// source / destination pair twice in one process -- hipMalloc, and the HIP virtual-memory API with the address range
// aligned to 1 GiB (hipMemAddressReserve + one hipMemCreate per buffer) -- and times band 1 / chunk / identity on both in
// shuffled rounds through the C ABI.   Build + run: tools/vmm_probe.sh
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../include/mibayer.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf (stderr, "%s: %s\n", #x, hipGetErrorString (e_)); exit (1); } } while (0)

static void *vmm_alloc (size_t bytes, size_t va_align, int dev)
{
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  size_t gran = 0;
  CK (hipMemGetAllocationGranularity (&gran, &prop, hipMemAllocationGranularityRecommended));
  const size_t size = (bytes + gran - 1) / gran * gran;
  hipMemGenericAllocationHandle_t h;
  CK (hipMemCreate (&h, size, &prop, 0));
  void *va = nullptr;
  CK (hipMemAddressReserve (&va, size, va_align, nullptr, 0));
  CK (hipMemMap (va, size, 0, h, 0));
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  CK (hipMemSetAccess (va, size, &acc, 1));
  fprintf (stderr, "# vmm: %zu bytes (granularity %zu) at %p (requested alignment %zu)\n", size, gran, va, va_align);
  return va;
}

/* `bytes` of device memory, contiguous in virtual addresses, backed by separate physical allocations of `piece` bytes each:
 * the page tables cannot merge pages of different allocations into one fragment, so a TLB entry covers at most `piece` */
static void *vmm_alloc_pieces (size_t bytes, size_t piece, int dev)
{
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  const size_t n = (bytes + piece - 1) / piece;
  char *va = nullptr;
  CK (hipMemAddressReserve ((void **) &va, n * piece, 0, nullptr, 0));
  for (size_t i = 0; i < n; i++) {
    hipMemGenericAllocationHandle_t h;
    CK (hipMemCreate (&h, piece, &prop, 0));
    CK (hipMemMap (va + i * piece, piece, 0, h, 0));
    CK (hipMemRelease (h));       /* the mapping keeps the memory */
  }
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  CK (hipMemSetAccess (va, n * piece, &acc, 1));
  return va;
}

int main (int argc, char **argv)
{
  const int W = 3840, H = 2160, N = 64;
  const size_t va_align = argc > 1 ? strtoull (argv[1], nullptr, 0) : ((size_t) 1 << 30);
  struct Plan { const char *name; const char *band; int variant; mibayer_ctx *ctx; };
  std::vector<Plan> plans = { { "4x2/band1", "1", 1, nullptr }, { "4x2/chunk", "-1", 1, nullptr },
    { "1x8/chunk", "-1", 3, nullptr }, { "1x8/identity", "0", 3, nullptr } };
  for (Plan &p : plans) {
    setenv ("MIBAYER_XCD_BAND", p.band, 1);
    mibayer_cfg cfg = {};
    cfg.struct_size = sizeof cfg;
    cfg.width = W; cfg.height = H; cfg.pattern = MIBAYER_RGGB;
    cfg.r_off = 2; cfg.g_off = 1; cfg.b_off = 0;
    cfg.device = 0; cfg.variant = p.variant;
    if (mibayer_create (&cfg, &p.ctx) != MIBAYER_OK) { fprintf (stderr, "create failed\n"); return 1; }
  }
  const size_t sb = (size_t) W * H, db = (size_t) 4 * W * H;
  struct Pair { const char *how; void *src, *dst; };
  std::vector<Pair> pairs;
  /* "c...": the physically contiguous pairs are allocated FIRST, alternating with plain ones (order bias) */
  const bool contiguous_first = argc > 2 && argv[2][0] == 'c';
  auto plain_pair = [&] (const char *name) {
    void *s, *d; CK (hipMalloc (&s, N * sb)); CK (hipMalloc (&d, N * db)); pairs.push_back ({ name, s, d });
  };
  auto contiguous_pair = [&] (const char *name) {     /* KFD_IOC_ALLOC_MEM_FLAGS_CONTIGUOUS */
    void *s = nullptr, *d = nullptr;
    if (hipExtMallocWithFlags (&s, N * sb, hipDeviceMallocContiguous) != hipSuccess
        || hipExtMallocWithFlags (&d, N * db, hipDeviceMallocContiguous) != hipSuccess) {
      fprintf (stderr, "# hipDeviceMallocContiguous refused: %s\n", hipGetErrorString (hipGetLastError ()));
      return;
    }
    pairs.push_back ({ name, s, d });
  };
  if (contiguous_first) {
    contiguous_pair ("contiguous"); plain_pair ("hipMalloc"); contiguous_pair ("contiguous#2"); plain_pair ("hipMalloc#2");
    contiguous_pair ("contiguous#3"); plain_pair ("hipMalloc#3");
  } else {
    plain_pair ("hipMalloc");
    pairs.push_back ({ "vmm", vmm_alloc (N * sb, va_align, 0), vmm_alloc (N * db, va_align, 0) });
    plain_pair ("hipMalloc#2"); contiguous_pair ("contiguous"); contiguous_pair ("contiguous#2");
  }
  for (Pair &pr : pairs)
    if (mibayer_fill_synthetic (plans[0].ctx, pr.src, sb, 0, N, 2, mibayer_ctx_stream (plans[0].ctx)) != MIBAYER_OK) return 1;
  mibayer_sync (plans[0].ctx);
  float ms;
  for (int i = 0; i < 8; i++) mibayer_time_device (plans[0].ctx, pairs[0].src, sb, pairs[0].dst, db, N, 0, 40, &ms);
  struct Cell { int plan, pair; std::vector<float> t; };
  std::vector<Cell> cells;
  for (int a = 0; a < (int) plans.size (); a++) for (int b = 0; b < (int) pairs.size (); b++) cells.push_back ({ a, b, {} });
  std::mt19937 rng (11);
  for (int r = 0; r < 7; r++) {
    std::vector<int> order (cells.size ());
    for (size_t i = 0; i < order.size (); i++) order[i] = (int) i;
    std::shuffle (order.begin (), order.end (), rng);
    for (int i : order) {
      Cell &c = cells[i];
      if (mibayer_time_device (plans[c.plan].ctx, pairs[c.pair].src, sb, pairs[c.pair].dst, db, N, 1, 8, &ms) != MIBAYER_OK) return 1;
      if (r) c.t.push_back (ms);
    }
  }
  if (argc > 2 && argv[2][0] == 'a') {
    /* arena mode: ONE allocation (physically contiguous if the driver grants it), the source at its start, the destination
     * slid through it in 128 MiB steps -- is the state a function of the distance between the two streams? */
    const size_t step = (size_t) (argc > 4 ? atoi (argv[4]) : 128) << 20, first = (size_t) (argc > 5 ? atoi (argv[5]) : 1024) << 20;
    const int npos = argc > 3 ? atoi (argv[3]) : 36;
    const size_t arena_bytes = first + (size_t) npos * step + N * db;
    char *arena = nullptr;
    const char *how = "hipDeviceMallocContiguous";
    if (argv[2][1] == 'p' || hipExtMallocWithFlags ((void **) &arena, arena_bytes, hipDeviceMallocContiguous) != hipSuccess) {
      (void) hipGetLastError ();
      how = "hipMalloc";
      CK (hipMalloc ((void **) &arena, arena_bytes));
    }
    printf ("# arena %zu MiB by %s at %p; source at +0, destination at +offset; ms per launch band1 / chunk\n",
        arena_bytes >> 20, how, (void *) arena);
    if (mibayer_fill_synthetic (plans[0].ctx, arena, sb, 0, N, 2, mibayer_ctx_stream (plans[0].ctx)) != MIBAYER_OK) return 1;
    mibayer_sync (plans[0].ctx);
    std::vector<std::vector<float>> t (2 * npos);
    for (int r = 0; r < 6; r++) {
      std::vector<int> order (2 * npos);
      for (int i = 0; i < 2 * npos; i++) order[i] = i;
      std::shuffle (order.begin (), order.end (), rng);
      for (int i : order) {
        if (mibayer_time_device (plans[i & 1].ctx, arena, sb, arena + first + (size_t) (i >> 1) * step, db, N, 1, 8, &ms) != MIBAYER_OK) return 1;
        if (r) t[i].push_back (ms);
      }
    }
    for (int k = 0; k < npos; k++) {
      std::sort (t[2 * k].begin (), t[2 * k].end ());
      std::sort (t[2 * k + 1].begin (), t[2 * k + 1].end ());
      printf ("+%5zu MiB  %.4f  %.4f\n", (first + (size_t) k * step) >> 20, t[2 * k][2], t[2 * k + 1][2]);
    }
    return 0;
  }
  if (argc > 2 && argv[2][0] == 's') {
    /* stride mode: one arena (physically contiguous unless "sp"), frames `pad` bytes apart -- the eight streams of the
     * chunk order are then 8 x (frame + pad) bytes apart: which distances collide? */
    static const size_t pads[] = { 0, 256, 1024, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288, 1048576,
      2097152, 4096 + 256, 65536 + 4096, 1048576 + 65536 + 4096 + 256, 33177600 / 8, 3 * 1048576 };
    const int np = (int) (sizeof pads / sizeof pads[0]);
    const size_t maxpad = (size_t) 33177600 / 8;
    const size_t src_off_end = N * sb;
    const size_t dst_off = ((src_off_end + ((size_t) 1 << 30) - 1) >> 30) << 30;
    const size_t arena_bytes = dst_off + N * (db + maxpad);
    char *arena = nullptr;
    const char *how = "hipDeviceMallocContiguous";
    if (argv[2][1] == 'p' || hipExtMallocWithFlags ((void **) &arena, arena_bytes, hipDeviceMallocContiguous) != hipSuccess) {
      (void) hipGetLastError ();
      how = "hipMalloc";
      CK (hipMalloc ((void **) &arena, arena_bytes));
    }
    printf ("# arena %zu MiB by %s; destination frames (33177600 + pad) bytes apart; ms per launch band1 / chunk / identity(1x8)\n",
        arena_bytes >> 20, how);
    if (mibayer_fill_synthetic (plans[0].ctx, arena, sb, 0, N, 2, mibayer_ctx_stream (plans[0].ctx)) != MIBAYER_OK) return 1;
    mibayer_sync (plans[0].ctx);
    const int pl[3] = { 0, 1, 3 };
    std::vector<std::vector<float>> t (3 * np);
    for (int r = 0; r < 6; r++) {
      std::vector<int> order (3 * np);
      for (int i = 0; i < 3 * np; i++) order[i] = i;
      std::shuffle (order.begin (), order.end (), rng);
      for (int i : order) {
        if (mibayer_time_device (plans[pl[i % 3]].ctx, arena, sb, arena + dst_off, db + pads[i / 3], N, 1, 8, &ms) != MIBAYER_OK) return 1;
        if (r) t[i].push_back (ms);
      }
    }
    for (int k = 0; k < np; k++) {
      printf ("pad %9zu  stream distance %#11zx ", pads[k], 8 * (db + pads[k]));
      for (int a = 0; a < 3; a++) { std::sort (t[3 * k + a].begin (), t[3 * k + a].end ()); printf ("  %.4f", t[3 * k + a][2]); }
      printf ("\n");
    }
    return 0;
  }
  if (argc > 2 && argv[2][0] == 'v') {
    /* piece mode: the destination (and with "vs" the source too) backed by physical pieces of a given size */
    static const size_t piece_mb[] = { 2, 8, 32, 128, 512, 0 };
    const bool src_too = argv[2][1] == 's';
    struct Buf { size_t mb; void *src, *dst; };
    std::vector<Buf> bufs;
    void *src0 = nullptr;
    CK (hipMalloc (&src0, N * sb));
    for (size_t mb : piece_mb) {
      Buf b = { mb, src0, nullptr };
      if (mb) {
        b.dst = vmm_alloc_pieces (N * db, mb << 20, 0);
        if (src_too) b.src = vmm_alloc_pieces (N * sb, mb << 20, 0);
      } else {
        CK (hipExtMallocWithFlags (&b.dst, N * db, hipDeviceMallocContiguous));
        if (src_too) CK (hipExtMallocWithFlags (&b.src, N * sb, hipDeviceMallocContiguous));
      }
      if (mibayer_fill_synthetic (plans[0].ctx, b.src, sb, 0, N, 2, mibayer_ctx_stream (plans[0].ctx)) != MIBAYER_OK) return 1;
      bufs.push_back (b);
    }
    { Buf b = { 1, src0, nullptr }; CK (hipMalloc (&b.dst, N * db)); bufs.push_back (b); }   /* plain hipMalloc */
    mibayer_sync (plans[0].ctx);
    const int nb = (int) bufs.size (), npl = (int) plans.size ();
    std::vector<std::vector<float>> t (nb * npl);
    for (int r = 0; r < 6; r++) {
      std::vector<int> order (nb * npl);
      for (int i = 0; i < nb * npl; i++) order[i] = i;
      std::shuffle (order.begin (), order.end (), rng);
      for (int i : order) {
        if (mibayer_time_device (plans[i % npl].ctx, bufs[i / npl].src, sb, bufs[i / npl].dst, db, N, 1, 8, &ms) != MIBAYER_OK) return 1;
        if (r) t[i].push_back (ms);
      }
    }
    printf ("# destination%s in physical pieces of ... (0 = one physically contiguous allocation, 1 = plain hipMalloc); ms per launch\n%-10s", src_too ? " and source" : "", "piece MiB");
    for (Plan &p : plans) printf ("  %-12s", p.name);
    printf ("\n");
    for (int b = 0; b < nb; b++) {
      printf ("%-10zu", bufs[b].mb);
      for (int a = 0; a < npl; a++) { std::vector<float> &v = t[b * npl + a]; std::sort (v.begin (), v.end ()); printf ("  %.4f      ", v[2]); }
      printf ("\n");
    }
    return 0;
  }
  if (argc > 2 && argv[2][0] == 'P') {
    /* placement mode: ONE destination (physically contiguous, and a plain one), six small source candidates: how much
     * does choosing the source allocation by measurement gain, and does the choice hold when re-timed? */
    struct B { const char *how; void *p; };
    std::vector<B> srcs, dsts;
    for (int k = 0; k < 6; k++) {
      void *q = nullptr;
      if ((k & 1) == 0 && hipExtMallocWithFlags (&q, N * sb, hipDeviceMallocContiguous) == hipSuccess) srcs.push_back ({ "contig", q });
      else { (void) hipGetLastError (); CK (hipMalloc (&q, N * sb)); srcs.push_back ({ "plain", q }); }
    }
    { void *q = nullptr; if (hipExtMallocWithFlags (&q, N * db, hipDeviceMallocContiguous) == hipSuccess) dsts.push_back ({ "contig", q }); (void) hipGetLastError ();
      CK (hipMalloc (&q, N * db)); dsts.push_back ({ "plain", q }); }
    for (B &b : srcs) if (mibayer_fill_synthetic (plans[0].ctx, b.p, sb, 0, N, 2, mibayer_ctx_stream (plans[0].ctx)) != MIBAYER_OK) return 1;
    mibayer_sync (plans[0].ctx);
    for (int i = 0; i < 8; i++) mibayer_time_device (plans[0].ctx, srcs[0].p, sb, dsts[0].p, db, N, 0, 40, &ms);
    for (B &d : dsts) {
      for (int a = 0; a < 2; a++) {
        const int ns = (int) srcs.size ();
        std::vector<std::vector<float>> t (ns);
        for (int r = 0; r < 4; r++) {
          std::vector<int> order (ns);
          for (int i = 0; i < ns; i++) order[i] = i;
          std::shuffle (order.begin (), order.end (), rng);
          for (int i : order) {
            if (mibayer_time_device (plans[a].ctx, srcs[i].p, sb, d.p, db, N, 1, 6, &ms) != MIBAYER_OK) return 1;
            if (r) t[i].push_back (ms);
          }
        }
        int best = 0;
        printf ("dst %-6s %-10s src:", d.how, plans[a].name);
        for (int i = 0; i < ns; i++) {
          std::sort (t[i].begin (), t[i].end ());
          printf (" %s %.4f", srcs[i].how, t[i][1]);
          if (t[i][1] < t[best][1]) best = i;
        }
        float again[3];
        for (int k = 0; k < 3; k++) mibayer_time_device (plans[a].ctx, srcs[best].p, sb, d.p, db, N, 2, 20, &again[k]);
        printf (" | best #%d, re-timed %.4f %.4f %.4f (%.1f %%)\n", best, again[0], again[1], again[2],
            100.0 * 5.0 * W * H * N / (again[1] * 1e-3) / 8e12);
      }
    }
    return 0;
  }
  if (argc > 2 && argv[2][0] == 'x') {
    /* cross mode: source of pair i with destination of pair j -- does the state follow the source, the destination or the
     * combination? */
    const int np = (int) pairs.size ();
    for (int a = 0; a < 2; a++) {
      std::vector<std::vector<float>> t (np * np);
      for (int r = 0; r < 6; r++) {
        std::vector<int> order (np * np);
        for (int i = 0; i < np * np; i++) order[i] = i;
        std::shuffle (order.begin (), order.end (), rng);
        for (int i : order) {
          if (mibayer_time_device (plans[a].ctx, pairs[i / np].src, sb, pairs[i % np].dst, db, N, 1, 8, &ms) != MIBAYER_OK) return 1;
          if (r) t[i].push_back (ms);
        }
      }
      printf ("# %s: rows = source of, columns = destination of; ms per launch\n%-14s", plans[a].name, "");
      for (Pair &pr : pairs) printf ("  %-12s", pr.how);
      printf ("\n");
      for (int i = 0; i < np; i++) {
        printf ("%-14s", pairs[i].how);
        for (int j = 0; j < np; j++) {
          std::vector<float> &v = t[i * np + j];
          std::sort (v.begin (), v.end ());
          printf ("  %.4f      ", v[v.size () / 2]);
        }
        printf ("\n");
      }
    }
    return 0;
  }
  printf ("# 4K x 64, ms per launch (median of 6 shuffled rounds x 8 launches), %% of 8 TB/s\n%-14s", "plan");
  for (Pair &pr : pairs) printf ("  %-18s", pr.how);
  printf ("\n");
  for (int a = 0; a < (int) plans.size (); a++) {
    printf ("%-14s", plans[a].name);
    for (int b = 0; b < (int) pairs.size (); b++) {
      std::vector<float> &t = cells[a * pairs.size () + b].t;
      std::sort (t.begin (), t.end ());
      const float m = t[t.size () / 2];
      printf ("  %.4f (%.1f %%)  ", m, 100.0 * 5.0 * W * H * N / (m * 1e-3) / 8e12);
    }
    printf ("\n");
  }
  return 0;
}
