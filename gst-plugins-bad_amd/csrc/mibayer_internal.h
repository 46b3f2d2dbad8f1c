/* Internal declarations shared by the kernel TU and the C-ABI TU. */
#ifndef MIBAYER_INTERNAL_H
#define MIBAYER_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace mibayer {

constexpr int kNumXcd = 8;      /* MI355X: 8 XCDs, block b is dispatched to XCD b % 8 */

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
  int tiles_x;
  int tiles_y;
  long long tile_rows;          /* nframes * tiles_y                            */
  int band;                     /* tile rows per XCD band; 0 = identity map     */
  int xcd_rot;                  /* tuning: XCD k takes the bands of XCD (k+rot)%8 */
  uint32_t sel[4];              /* v_perm_b32 selectors of output pixel 0..3   */
  int swap_rows;                /* 1 for grbg / gbrg                           */
};

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
 * walks its rows left to right.  Horizontal neighbours (and, inside a band,
 * vertical neighbours) then share an L2, while all 8 XCDs still work within
 * 8*band consecutive tile rows of the same frame, which keeps the DRAM write
 * stream compact.  band = 0 selects the identity map. */
__host__ __device__ inline long long
block_to_tile (long long block, int tiles_x, long long tile_rows, int band,
    int rot = 0)
{
  if (band <= 0) {
    return block < tile_rows * tiles_x ? block : -1;
  }
  const long long xcd = (block + rot) % kNumXcd;
  const long long i = block / kNumXcd;          /* i-th block of this XCD */
  const long long per_group = (long long) band * tiles_x;
  const long long group = i / per_group;
  const long long in_group = i - group * per_group;
  const long long r_local = in_group / tiles_x;
  const long long tx = in_group - r_local * tiles_x;
  const long long row = (group * kNumXcd + xcd) * band + r_local;
  return row < tile_rows ? row * tiles_x + tx : -1;
}

__host__ __device__ inline long long
grid_blocks_for (int tiles_x, long long tile_rows, int band)
{
  if (band <= 0)
    return tile_rows * tiles_x;
  const long long group_rows = (long long) kNumXcd * band;
  const long long groups = (tile_rows + group_rows - 1) / group_rows;
  return groups * group_rows * tiles_x;
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
};

int variant_count ();
const Variant &variant (int id);
int resolve_variant (int id, int width);        /* 0 ("auto") -> a concrete id */

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
  int tiles_x;                  /* filled by launch_rgb2bayer: 1024-px column strips */
  long long tile_rows;          /*                            groups of R2B_ROWS rows */
  int band;                     /* XCD band map (see block_to_tile); -1 = one chunk per XCD */
  uint32_t sel_lo[2];           /* v_perm selectors per row parity: pixels 0,1 */
  uint32_t sel_hi[2];           /*                                  pixels 2,3 */
};
hipError_t launch_rgb2bayer (const R2BParams &p, bool vec16, hipStream_t stream);

/* synthetic mosaic generator kernel launcher */
hipError_t launch_fill_synthetic (uint8_t *d_buf, int width, int height,
    int stride, unsigned long long frame_bytes, uint32_t first_frame,
    int nframes, uint32_t seed, hipStream_t stream);

}  /* namespace mibayer */
#endif
