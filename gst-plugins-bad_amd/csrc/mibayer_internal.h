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
  long long ntiles;             /* nframes * tiles_y * tiles_x                 */
  long long chunk;              /* tiles per XCD = ceil(ntiles / 8); 0 = identity */
  uint32_t sel[4];              /* v_perm_b32 selectors of output pixel 0..3   */
  int swap_rows;                /* 1 for grbg / gbrg                           */
};

/* XCD-aware block -> tile map, identical on host and device.
 * Blocks are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8), each
 * XCD has a private 4 MiB L2.  Giving XCD k the contiguous tile range
 * [k*chunk, (k+1)*chunk) makes tiles that share halo rows / columns run on the
 * same XCD at about the same time, so halo re-reads hit that XCD's L2. */
__host__ __device__ inline long long
block_to_tile (long long block, long long ntiles, long long chunk)
{
  if (chunk == 0)               /* identity map (A/B arm) */
    return block < ntiles ? block : -1;
  long long xcd = block % kNumXcd;
  long long i = block / kNumXcd;
  long long id = xcd * chunk + i;
  return (i < chunk && id < ntiles) ? id : -1;
}

struct Variant {
  const char *name;
  int tile_w;                   /* pixels */
  int tile_h;                   /* rows   */
  int threads;
  int xcd_remap;                /* 1: XCD-chunked block->tile map, 0: identity (A/B arm) */
  void (*fast) (KParams);       /* W%16==0, 16-byte aligned rows both sides */
  void (*generic) (KParams);    /* any even W >= 4, 4-byte aligned rows     */
};

int variant_count ();
const Variant &variant (int id);

/* synthetic mosaic generator kernel launcher */
hipError_t launch_fill_synthetic (uint8_t *d_buf, int width, int height,
    int stride, unsigned long long frame_bytes, uint32_t first_frame,
    int nframes, uint32_t seed, hipStream_t stream);

}  /* namespace mibayer */
#endif
