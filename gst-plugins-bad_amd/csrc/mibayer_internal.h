/* Internal declarations shared by the kernel TU and the C-ABI TU. */
#ifndef MIBAYER_INTERNAL_H
#define MIBAYER_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace mibayer {

constexpr int kNumXcd = 8;      /* MI355X: 8 XCDs, block b is dispatched to XCD b % 8 */

/* Unsigned 32-bit division by a launch-time constant without a divide
 * (Granlund & Montgomery): 3 scalar instructions instead of the ~100-instruction
 * 64-bit division sequence the compiler emits for `a / b` with runtime b. */
struct FastDiv {
  uint32_t d;                   /* the divisor          */
  uint32_t mul;                 /* magic multiplier     */
  uint32_t sh1, sh2;            /* post-shifts          */
};

inline FastDiv make_fastdiv (uint32_t d)
{
  FastDiv f;
  f.d = d ? d : 1;
  uint32_t l = 0;               /* ceil (log2 d) */
  while (l < 32 && (1ull << l) < f.d)
    l++;
  f.mul = (uint32_t) ((((1ull << l) - f.d) << 32) / f.d + 1);
  f.sh1 = l < 1 ? l : 1;
  f.sh2 = l > 0 ? l - 1 : 0;
  return f;
}

__host__ __device__ inline uint32_t fastdiv (uint32_t n, const FastDiv &f)
{
  const uint32_t t = (uint32_t) (((unsigned long long) n * f.mul) >> 32);
  return (t + ((n - t) >> f.sh1)) >> f.sh2;
}

/* XCD-aware block -> tile map, identical on host and device.
 *
 * Blocks are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8) and
 * each XCD has a private 4 MiB L2.  With the identity map horizontally adjacent
 * tiles land on different XCDs, so every tile's left/right halo dword drags a
 * full 128-byte line of its neighbour through the fabric a second time
 * (measured: L2->fabric reads = 2.0x the algorithmic bytes for 256-px tiles).
 *
 * The band map hands XCD k whole tile rows: tile row R (counted through the
 * batch, R = frame * tiles_y + ty) belongs to XCD (R / band) % 8, and an XCD
 * walks its rows left to right.  band = ceil (tile_rows / 8) gives every XCD one
 * contiguous chunk of the batch; band = 0 selects the identity map.
 * Everything is 32-bit (grids are < 2^31 blocks) and division-free. */
struct TileMap {
  FastDiv tiles_x;              /* tiles per tile row                           */
  FastDiv tiles_y;              /* tile rows per frame                          */
  FastDiv per_group;            /* band * tiles_x                               */
  uint32_t tile_rows;           /* tile rows of this launch (nframes * tiles_y, or
                                   the rows of one horizontal band of one frame) */
  uint32_t row0;                /* first tile row of this launch (0 unless a band) */
  int band;                     /* tile rows per XCD band; 0 = identity map     */
};

struct TileId {
  uint32_t row;                 /* tile row in the batch (TileMap.row0 included) */
  uint32_t tx;
  bool valid;
};

__host__ __device__ inline TileId block_to_tile (uint32_t block, const TileMap &m)
{
  TileId t;
  if (m.band <= 0) {
    t.row = fastdiv (block, m.tiles_x);
    t.tx = block - t.row * m.tiles_x.d;
  } else {
    const uint32_t xcd = block & (kNumXcd - 1);
    const uint32_t i = block / kNumXcd;         /* i-th block of this XCD */
    const uint32_t group = fastdiv (i, m.per_group);
    const uint32_t in_group = i - group * m.per_group.d;
    const uint32_t r_local = fastdiv (in_group, m.tiles_x);
    t.tx = in_group - r_local * m.tiles_x.d;
    t.row = (group * kNumXcd + xcd) * (uint32_t) m.band + r_local;
  }
  t.valid = t.row < m.tile_rows;
  t.row += m.row0;
  return t;
}

/* linear tile index (row * tiles_x + tx) -> TileId, for the persistent arm */
__host__ __device__ inline TileId linear_to_tile (uint32_t tile, const TileMap &m)
{
  TileId t;
  t.row = fastdiv (tile, m.tiles_x);
  t.tx = tile - t.row * m.tiles_x.d;
  t.valid = t.row < m.tile_rows;
  t.row += m.row0;
  return t;
}

inline long long grid_blocks_for (int tiles_x, long long tile_rows, int band)
{
  if (band <= 0)
    return tile_rows * tiles_x;
  const long long group_rows = (long long) kNumXcd * band;
  const long long groups = (tile_rows + group_rows - 1) / group_rows;
  return groups * group_rows * tiles_x;
}

inline TileMap make_tile_map (int tiles_x, int tiles_y, long long tile_rows,
    int band, long long row0 = 0)
{
  TileMap m;
  m.tiles_x = make_fastdiv ((uint32_t) tiles_x);
  m.tiles_y = make_fastdiv ((uint32_t) tiles_y);
  m.per_group = make_fastdiv ((uint32_t) (band > 0 ? band : 1)
      * (uint32_t) tiles_x);
  m.tile_rows = (uint32_t) tile_rows;
  m.row0 = (uint32_t) row0;
  m.band = band;
  return m;
}

constexpr int kMaxList = 16;    /* frames of one list launch (mibayer_process_device_list) */

/* Kernel arguments (passed by value -> SGPRs). */
struct KParams {
  const uint8_t *src;
  uint8_t *dst;
  unsigned long long src_frame_bytes;
  unsigned long long dst_frame_bytes;
  int width;
  int height;
  int src_stride;
  int dst_stride;
  int wlimit4;                  /* ROUND_UP_4(width): last readable column + 1 */
  int dn_last;                  /* source row standing in for row `height`     */
  TileMap map;
  int start_sleep;              /* s_sleep(1) iterations before a workgroup's first load */
  uint32_t sel[4];              /* v_perm_b32 selectors of output pixel 0..3   */
  int swap_rows;                /* 1 for grbg / gbrg                           */
  /* list launch: frame f is read at src_list[f] and written at dst_list[f] (frames that are
   * separate allocations, e.g. one GstBuffer each) instead of src / dst + f * frame bytes */
  int nlist;
  const uint8_t *src_list[kMaxList];
  uint8_t *dst_list[kMaxList];
};

/* frame index is wave-uniform: the table look-up is a scalar load from the kernel arguments */
__device__ __forceinline__ const uint8_t *frame_src (const KParams &p, uint32_t frame)
{
  return p.nlist ? p.src_list[frame] : p.src + frame * p.src_frame_bytes;
}

__device__ __forceinline__ uint8_t *frame_dst (const KParams &p, uint32_t frame)
{
  return p.nlist ? p.dst_list[frame] : p.dst + frame * p.dst_frame_bytes;
}

struct Variant {
  const char *name;
  int tile_w;                   /* pixels */
  int tile_h;                   /* rows   */
  int threads;
  int band;                     /* XCD band map: tile rows per band; 0 = identity, -1 = one
                                   contiguous chunk per XCD */
  int persistent;               /* 1: `fast` is a persistent kernel launched with a
                                   CU-count-sized grid that loops over the tiles */
  void (*fast) (KParams);       /* W%16==0, 16-byte aligned rows both sides */
  void (*generic) (KParams);    /* any even W >= 4, 4-byte aligned rows     */
  /* generic geometry whose output rows start off the 64-byte sector grid: every wave-store starts on a
   * 64- / 128-byte boundary (per-row lane shift, mibayer_kernels.hip); needs 8-byte aligned output rows.
   * nullptr: the variant has no such arm */
  void (*aligned64) (KParams);
  void (*aligned128) (KParams);
};

int variant_count ();
const Variant &variant (int id);
int resolve_variant (int id, int width);        /* 0 ("auto") -> a concrete id */
/* batch-class default (variant id 1-3, band) of a frame width with a measured winner the rules miss; false: none */
bool known_width_plan (int width, int *variant, int *band);
/* "auto" for a launch of one frame: the production shape whose grid needs the fewest rounds of `slots` workgroups */
int frame_class_variant (int width, int height, int slots);
/* the plain-store (write-back) arm of a production shape (ids 1-3), for output rows that start off a
 * 64-byte sector; any other id is returned unchanged */
int plain_store_twin (int id);
int hybrid_store_twin (int id);                 /* 1 -> 7, 2 -> 8, 3 -> 9: the hybrid store policy */
int production_shape_of (int id);               /* the inverse of both: 4, 7 -> 1; 5, 8 -> 2; 6, 9 -> 3 */

/* rgb2bayer (reference gst/bayer/gstrgb2bayer.c:230-278) */
struct R2BParams {
  const uint8_t *src;           /* 4 B/pixel */
  uint8_t *dst;                 /* 1 B/pixel mosaic */
  unsigned long long src_frame_bytes;
  unsigned long long dst_frame_bytes;
  int width;
  int height;
  int src_stride;
  int dst_stride;
  int out_dwords;               /* ROUND_UP_4(width) / 4 */
  long long total_rows;         /* nframes * height */
  int band;                     /* XCD band map (see block_to_tile); -1 = one chunk per XCD */
  int start_sleep;              /* s_sleep(1) iterations before the first load (tuning) */
  TileMap map;                  /* filled by launch_rgb2bayer: tiles = R rows x 1024 px */
  FastDiv div_height;
  uint32_t sel_lo[2];           /* v_perm selectors per row parity: pixels 0,1 */
  uint32_t sel_hi[2];           /*                                  pixels 2,3 */
  /* launch shape (tuning; every combination is bit-exact):
   *   flat_k  > 0: the flat kernel, the batch as one linear sequence of 4-pixel items (16 B in, one dword out),
   *                flat_k groups per thread kept in flight; 0: the tile kernel (rows x 1024 px per block)
   *   flat_px   4: one item per group (dword store), 8: two adjacent items per group (two 16-byte loads, one
   *                8-byte store; needs an even number of dwords per row and 8-byte aligned destination rows)
   *   flat_ld   1: nt hint on the loads (the input is read exactly once)
   *   rows      rows per block of the tile kernel (2, 4, 8, 16)                                             */
  int flat_k, flat_px, flat_ld, rows;
  FastDiv div_out_dwords;       /* filled by launch_rgb2bayer (flat kernel) */
  uint32_t item0, item_end;     /* first / one-past-last 4-pixel item of this launch (flat kernel) */
  /* list launch (flat kernel): frame f is read at src_list[f] and written at dst_list[f] -- frames that are
   * separate allocations, one GstBuffer each.  Every frame takes a whole number of blocks (blocks_per_frame), so
   * the frame index is block-uniform and the table look-up a scalar load from the kernel arguments */
  int nlist;
  FastDiv div_blocks_per_frame;
  const uint8_t *src_list[kMaxList];
  uint8_t *dst_list[kMaxList];
};
/* rows [row0, row0 + nrows) of the batch (nrows < 0: all); row0 must be a multiple of 16 */
hipError_t launch_rgb2bayer (const R2BParams &p, bool vec16, hipStream_t stream,
    long long row0 = 0, long long nrows = -1);
/* one launch over p.nlist (<= kMaxList) separately allocated frames (p.src_list / p.dst_list); needs the flat
 * kernel (p.flat_k > 0); dst8: every destination 8-byte aligned (two items per store allowed) */
hipError_t launch_rgb2bayer_list (const R2BParams &p, bool vec16, bool dst8, hipStream_t stream);

/* a kernel that only waits, `ms` milliseconds (drills: mibayer_internal_stall) */
hipError_t launch_stall (int ms, hipStream_t stream);

/* synthetic mosaic generator kernel launcher */
hipError_t launch_fill_synthetic (uint8_t *d_buf, int width, int height,
    int stride, unsigned long long frame_bytes, uint32_t first_frame,
    int nframes, uint32_t seed, hipStream_t stream);

}  /* namespace mibayer */
#endif
