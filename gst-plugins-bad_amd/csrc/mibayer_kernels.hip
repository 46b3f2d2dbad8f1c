/*
 * bayer2rgb demosaic for MI355X (gfx950, CDNA4) -- device code.
 *
 * Replaces, in ONE fused kernel, the reference's per-row CPU loops
 *   gst_bayer2rgb_split_and_upsample_horiz   gst/bayer/gstbayer2rgb.c:354-381
 *   bayer_orc_horiz_upsample_unaligned        gst/bayer/gstbayerorc.orc:3-19
 *   bayer_orc_merge_{bg,gr}_{bgra,abgr,rgba,argb}   gstbayerorc.orc:43-248
 * and the frame loop around them, gst_bayer2rgb_process (gstbayer2rgb.c:387-451).
 *
 * Algorithm (all uint8, avg(a,b) = (a+b+1)>>1, the ORC `avgub`):
 *   for every source row y two lines are defined (reference :354-381)
 *     E_y[x] = S(y,x)                 x even      O_y[x] = S(y,x)                 x odd
 *            = avg(S(y,x-1),S(y,x+1)) x odd              = avg(S(y,x-1),S(y,x+1)) x even
 *     with the edge columns  O_y[0]=S(y,1), E_y[W-1]=S(y,W-2), O_y[W-2]=S(y,W-3)
 *   output row j uses rows u=up(j), j, d=dn(j) where up(0)=1 and dn(H-1)=H-4
 *   (the reference's 4-slot ring, :430-447), and with T = (j&1)^swap_rows
 *     T=0 (merge_bg): B'=E_j  R'=avg(O_u,O_d)  G = x even ? avg(avg(E_u,E_d),O_j) : O_j
 *     T=1 (merge_gr): B'=avg(E_u,E_d)  R'=O_j  G = x even ? E_j : avg(avg(O_u,O_d),E_j)
 *   R'/B' land on the r/b byte offsets (swapped for rggb/gbrg, :403-407), the
 *   remaining byte is 255.
 *
 * Mapping to the machine:
 *   - a lane owns 4 horizontally adjacent pixels = one source dword, so a wave64
 *     reads 256 contiguous bytes and writes 1 KiB contiguous bytes per row
 *     (global_store_dwordx4, every 128-byte line fully written by one instruction);
 *   - all arithmetic is 4-pixels-per-instruction packed bytes: v_lerp_u8 is
 *     exactly avgub on four bytes, v_alignbit_b32 builds the x-1 / x+1 neighbour
 *     dwords, v_bfi_b32 does the even/odd column select, v_perm_b32 interleaves
 *     R',G,B',255 into the output pixels (selectors are kernel arguments, so one
 *     kernel serves all 4 byte layouts and all 4 Bayer orders);
 *   - a workgroup stages its tile plus one halo row above and below (and one halo
 *     dword left and right) in LDS with 16-byte coalesced loads; each wave then
 *     marches down its rows with a 3-row sliding window of (E,O) in registers;
 *   - the x-1 / x+4 neighbour bytes come from the adjacent lanes through DPP
 *     wave shifts (no memory traffic); only lanes 0 and 63 take theirs from LDS;
 *   - blockIdx -> tile is XCD-aware (mibayer_internal.h: block_to_tile).
 * The op is a pure HBM stream (1 B read + 4 B written per pixel, ~25 integer
 * ops per 4 pixels): no MFMA.
 */
#include "mibayer_internal.h"

#include <stdlib.h>
#include <type_traits>

namespace mibayer {

typedef uint32_t u32x4 __attribute__ ((ext_vector_type (4)));
typedef uint32_t u32x2 __attribute__ ((ext_vector_type (2)));
/* the same at dword alignment: all a gfx950 global load / store needs.  The generic paths (rows
 * that are not 16-byte aligned: width % 16 != 0, padded strides, odd pointers) move 16 bytes per
 * instruction through these instead of dword by dword */
typedef uint32_t u32x4_a4 __attribute__ ((ext_vector_type (4), aligned (4)));
typedef uint32_t u32x2_a4 __attribute__ ((ext_vector_type (2), aligned (4)));

/* ------------------------------------------------------------------------- */
/* packed-byte primitives                                                     */
/* ------------------------------------------------------------------------- */

/* four avgub at once */
template <bool INTRIN>
__device__ __forceinline__ uint32_t avg4 (uint32_t a, uint32_t b)
{
  if constexpr (INTRIN) {
    /* v_lerp_u8: per byte (a + b + (c & 1)) >> 1 */
    return __builtin_amdgcn_lerp (a, b, 0x01010101u);
  } else {
    return (a | b) - (((a ^ b) >> 1) & 0x7f7f7f7fu);
  }
}

/* ({hi,lo} >> (8*nbytes)) & 0xffffffff */
template <bool INTRIN>
__device__ __forceinline__ uint32_t funnel_bytes (uint32_t hi, uint32_t lo,
    int nbytes)
{
  if constexpr (INTRIN) {
    return __builtin_amdgcn_alignbit (hi, lo, 8 * nbytes);
  } else {
    return (uint32_t) ((((unsigned long long) hi << 32) | lo) >> (8 * nbytes));
  }
}

/* v_perm_b32: byte i of the result is byte sel[i] of {s0,s1} (0-3 = s1, 4-7 = s0),
 * 12 = 0x00, >= 13 = 0xff */
template <bool INTRIN>
__device__ __forceinline__ uint32_t perm4 (uint32_t s0, uint32_t s1,
    uint32_t sel)
{
  if constexpr (INTRIN) {
    return __builtin_amdgcn_perm (s0, s1, sel);
  } else {
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      uint32_t idx = (sel >> (8 * i)) & 0xffu, b;
      if (idx < 4)
        b = (s1 >> (8 * idx)) & 0xffu;
      else if (idx < 8)
        b = (s0 >> (8 * (idx - 4))) & 0xffu;
      else if (idx == 12)
        b = 0;
      else
        b = 0xffu;
      r |= b << (8 * i);
    }
    return r;
  }
}

/* (m & a) | (~m & b)  ->  v_bfi_b32 */
__device__ __forceinline__ uint32_t bsel (uint32_t m, uint32_t a, uint32_t b)
{
  return (m & a) | (~m & b);
}

constexpr uint32_t kEvenBytes = 0x00ff00ffu;

/* DPP wave shifts (gfx9 encodings).  Lanes whose source lane does not exist
 * keep `edge`. */
__device__ __forceinline__ uint32_t from_lane_below (uint32_t edge, uint32_t v)
{
  /* wave_shr:1 -- lane i receives lane i-1 */
  return (uint32_t) __builtin_amdgcn_update_dpp ((int) edge, (int) v, 0x138,
      0xf, 0xf, false);
}

__device__ __forceinline__ uint32_t from_lane_above (uint32_t edge, uint32_t v)
{
  /* wave_shl:1 -- lane i receives lane i+1 */
  return (uint32_t) __builtin_amdgcn_update_dpp ((int) edge, (int) v, 0x130,
      0xf, 0xf, false);
}

/* ------------------------------------------------------------------------- */
/* per-row horizontal lines, reference gstbayer2rgb.c:354-381                  */
/* ------------------------------------------------------------------------- */

struct Lines { uint32_t e, o; };

/* c  = S[x0..x0+3], cl = dword left of it, cr = dword right of it.
 * first: x0 == 0.   lastmode: 1 = this lane holds columns W-4..W-1,
 *                             2 = this lane holds columns W-2..W-1 only. */
template <bool INTRIN, bool GENERIC>
__device__ __forceinline__ Lines row_lines (uint32_t c, uint32_t cl,
    uint32_t cr, bool first, int lastmode)
{
  /* x-1 neighbours: [cl.3, c0, c1, c2];  x+1 neighbours: [c1, c2, c3, cr.0] */
  uint32_t lsh = funnel_bytes<INTRIN> (c, cl, 3);
  uint32_t rsh = funnel_bytes<INTRIN> (cr, c, 1);
  /* O[0] = S[1] (:361): make the left neighbour of column 0 equal S[1] */
  uint32_t lsh_first = (lsh & 0xffffff00u) | ((c >> 8) & 0xffu);
  lsh = first ? lsh_first : lsh;
  /* E[W-1] = S[W-2], O[W-2] = S[W-3] (:372-380): right neighbours of the last
   * two columns are replaced by their left neighbours */
  uint32_t t = lsh >> 16;                    /* [c1, c2, 0, 0] */
  uint32_t rsh_last = t | (t << 16);          /* [c1, c2, c1, c2] */
  rsh = (lastmode == 1) ? rsh_last : rsh;
  if constexpr (GENERIC)
    rsh = (lastmode == 2) ? lsh : rsh;        /* [L, c0, -, -] */
  uint32_t a = avg4<INTRIN> (lsh, rsh);
  Lines r;
  r.e = bsel (kEvenBytes, c, a);
  r.o = bsel (kEvenBytes, a, c);
  return r;
}

/* ------------------------------------------------------------------------- */
/* vertical merge + interleave, reference gstbayerorc.orc:43-92                */
/* ------------------------------------------------------------------------- */

template <bool INTRIN>
__device__ __forceinline__ u32x4 merge_rows (const Lines &u, const Lines &c,
    const Lines &d, int type, const uint32_t (&sel)[4])
{
  uint32_t ve = avg4<INTRIN> (u.e, d.e);
  uint32_t vo = avg4<INTRIN> (u.o, d.o);
  uint32_t rq, bq, g;
  if (type == 0) {              /* merge_bg, orc:57-66 */
    bq = c.e;
    rq = vo;
    g = bsel (kEvenBytes, avg4<INTRIN> (ve, c.o), c.o);
  } else {                      /* merge_gr, orc:83-92 */
    bq = ve;
    rq = c.o;
    g = bsel (kEvenBytes, c.e, avg4<INTRIN> (vo, c.e));
  }
  /* [r0 b0 r1 b1], [r2 b2 r3 b3] */
  uint32_t m_lo = perm4<INTRIN> (rq, bq, 0x01050004u);
  uint32_t m_hi = perm4<INTRIN> (rq, bq, 0x03070206u);
  u32x4 px;
  px.x = perm4<INTRIN> (m_lo, g, sel[0]);
  px.y = perm4<INTRIN> (m_lo, g, sel[1]);
  px.z = perm4<INTRIN> (m_hi, g, sel[2]);
  px.w = perm4<INTRIN> (m_hi, g, sel[3]);
  return px;
}

/* ST = cache policy of the 16-byte output stores (the output is never re-read):
 * 0 plain, 1 nt (streaming hint), 2 sc1, 3 sc0 sc1 (write-through, line dropped
 * from the XCD L2), 4 nt sc1; + 8 = nt hint on the LDS kernel's row loads too,
 * + 16 = direct-to-LDS row loads */
template <int STLD, bool GENERIC>
__device__ __forceinline__ void store_pixels (uint8_t *p, u32x4 px,
    int lastmode)
{
  constexpr int ST = STLD & 7;  /* bit 3 = nt hint on the tile's row loads, bit 4 = direct-to-LDS loads */
  if constexpr (!GENERIC) {
    if constexpr (ST == 1 || ST == 5)         /* 5: the hybrid policy of generic geometries; aligned rows stream */
      __builtin_nontemporal_store (px, (u32x4 *) p);
    else if constexpr (ST == 2)
      asm volatile ("global_store_dwordx4 %0, %1, off sc1" :: "v" (p), "v" (px) : "memory");
    else if constexpr (ST == 3)
      asm volatile ("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v" (p), "v" (px) : "memory");
    else if constexpr (ST == 4)
      asm volatile ("global_store_dwordx4 %0, %1, off sc1 nt" :: "v" (p), "v" (px) : "memory");
    else
      *(u32x4 *) p = px;
  } else {
    /* rows start at dword alignment only: same 16-byte store, at that alignment; the lane that
     * holds the last two columns of a width % 4 == 2 frame writes two pixels */
    if (lastmode != 2) {
      if constexpr (ST == 1)
        __builtin_nontemporal_store (px, (u32x4_a4 *) p);
      else
        *(u32x4_a4 *) p = px;
    } else {
      u32x2 two;
      two.x = px.x;
      two.y = px.y;
      if constexpr (ST == 1)
        __builtin_nontemporal_store (two, (u32x2_a4 *) p);
      else
        *(u32x2_a4 *) p = two;
    }
  }
}

/* Store policy 5 ("hybrid"), generic geometries: output rows that start off the 128-byte line grid make every 1 KiB
 * wave-store begin and end inside a line that a neighbouring wave (or tile) completes.  Streaming (nt) stores push
 * such half-written lines out one half at a time and lose 4-11 points of HBM peak; write-back stores let the L2 put
 * the halves together but give up what nt gains on a pure output stream (3 points; profiles/r03_generic_path.log).
 * Here each lane picks by itself: nt when every line its 16 bytes touch lies completely inside this wave-store,
 * write-back for the one or two ragged lines at either end.  a7 = position of the lane's first byte in its line;
 * the wave-store covers bytes [-lane16, len - lane16) relative to it. */
__device__ __forceinline__ void store_pixels_hybrid (uint8_t *p, u32x4 px,
    int lastmode, int lane16, int len)
{
  const int a7 = (int) ((uint32_t) (uintptr_t) p & 127u);
  const int need = a7 > 112 ? 256 : 128;        /* the 16 bytes straddle a line boundary: both lines count */
  const bool inside = a7 <= lane16 && lane16 + need - a7 <= len;
  if (lastmode == 2) {
    u32x2 two;
    two.x = px.x;
    two.y = px.y;
    *(u32x2_a4 *) p = two;
  } else if (inside) {
    /* spelled out: the compiler folds an nt and a plain store of the same value under if / else into ONE plain store */
    asm volatile ("global_store_dwordx4 %0, %1, off nt" :: "v" (p), "v" (px) : "memory");
  } else {
    asm volatile ("global_store_dwordx4 %0, %1, off" :: "v" (p), "v" (px) : "memory");
  }
}

/* source row standing in for row y of the (virtually extended) frame:
 * up(0) = 1, dn(H-1) = dn_last (H-4, or 1 when H == 3) -- gstbayer2rgb.c:430-447 */
__device__ __forceinline__ int map_row (int y, int height, int dn_last)
{
  return y < 0 ? 1 : (y < height ? y : dn_last);
}

/* ------------------------------------------------------------------------- */
/* LDS-staged tile kernel                                                      */
/* ------------------------------------------------------------------------- */
/* WX x WY waves per workgroup; a wave covers 256 px (64 lanes x 4 px) and marches
 * RPW rows.  Tile = (256*WX) x (WY*RPW) px.
 * NEIGH: 0 = DPP wave shift, 1 = __shfl_up/down (ds_bpermute), 2 = LDS reads.   */
template <int WX, int WY, int RPW, int NEIGH, int ST, bool INTRIN,
    bool GENERIC>
__global__ void __launch_bounds__ (64 * WX * WY)
bayer2rgb_lds_kernel (KParams p)
{
  constexpr int NTHREADS = 64 * WX * WY;
  constexpr int TW = 256 * WX;
  constexpr int TR = WY * RPW;
  constexpr int NROWS = TR + 2;
  /* LDS row: 12 B pad | left halo dword | TW bytes | right halo dword | 12 B pad */
  constexpr int PITCH = TW + 32;
  constexpr int MAIN = 16;
  static_assert (RPW % 2 == 0, "row parity is derived from the in-tile row");

  __shared__ __attribute__ ((aligned (16))) uint8_t lds[NROWS * PITCH];

  const TileId tile = block_to_tile (blockIdx.x, p.map);
  if (!tile.valid)
    return;
  /* start delay: lets the store burst of the workgroup that just
   * retired on this CU drain before this one's loads reach the L2 (input-FIFO-full
   * cycles drop 5-9x, profiles/r01_delay_counters.md); +3..5 points of HBM
   * peak for the band / chunk block orders.  Measured alternatives that lost:
   * the same delay after the barrier or after the stores, a per-workgroup
   * stagger, lower occupancy (profiles/r01_sweep_start_delay.log). */
  for (int z = 0; z < p.start_sleep; z++)
    __builtin_amdgcn_s_sleep (1);
  const uint32_t frame = fastdiv (tile.row, p.map.tiles_y);
  const int ty = (int) (tile.row - frame * p.map.tiles_y.d);
  const uint8_t *src = frame_src (p, frame);
  uint8_t *dst = frame_dst (p, frame);
  const int tile_x = (int) tile.tx * TW;
  const int tile_y = ty * TR;
  const int tid = threadIdx.x;

  /* ---- stage rows tile_y-1 .. tile_y+TR (through map_row) into LDS ---------- */
  if constexpr (!GENERIC && (ST & 16) != 0) {
    /* experiment arm: direct global -> LDS loads (global_load_lds_dwordx4), no
     * staging registers and no ds_write pass.  The LDS destination of such a load
     * is wave-uniform base + lane * 16, so it needs one wave per 1 KiB tile row:
     * WX == 4.  Lanes right of the frame skip their load; what their LDS bytes
     * hold is never used (the last valid lane replaces its right neighbour). */
    static_assert (WX == 4, "one wave stages one 1024-px row");
    constexpr int RPP = NTHREADS / 64;    /* rows per pass = waves */
    constexpr int NPASS = (NROWS + RPP - 1) / RPP;
    const int c = (tid & 63) * 16;
    const int rr = tid >> 6;
#pragma unroll
    for (int i = 0; i < NPASS; i++) {
      const int r = i * RPP + rr;
      const int y = tile_y - 1 + r;
      if (r < NROWS && y <= p.height && tile_x + c < p.width) {
        const uint8_t *g = src
            + (size_t) map_row (y, p.height, p.dn_last) * p.src_stride
            + tile_x + c;
        __builtin_amdgcn_global_load_lds (
            (const __attribute__ ((address_space (1))) void *) g,
            (__attribute__ ((address_space (3))) void *) &lds[r * PITCH + MAIN],
            16, 0, 0);
      }
    }
  } else if constexpr (!GENERIC) {
    constexpr int TPR = TW / 16;          /* threads per row, 16 B each */
    constexpr int RPP = NTHREADS / TPR;   /* rows per pass */
    constexpr int NPASS = (NROWS + RPP - 1) / RPP;
    const int c = (tid % TPR) * 16;
    const int rr = tid / TPR;
    u32x4 v[NPASS];
#pragma unroll
    for (int i = 0; i < NPASS; i++) {
      const int r = i * RPP + rr;
      const int y = tile_y - 1 + r;
      v[i] = (u32x4) (0u);
      if (r < NROWS && y <= p.height && tile_x + c < p.width) {
        const uint8_t *g = src
            + (size_t) map_row (y, p.height, p.dn_last) * p.src_stride
            + tile_x + c;
        if constexpr ((ST & 8) != 0)
          v[i] = __builtin_nontemporal_load ((const u32x4 *) g);
        else
          v[i] = *(const u32x4 *) g;
      }
    }
#pragma unroll
    for (int i = 0; i < NPASS; i++) {
      const int r = i * RPP + rr;
      if (r < NROWS)
        *(u32x4 *) &lds[r * PITCH + MAIN + c] = v[i];
    }
  } else {
    /* generic geometry: the same 16-byte chunks, loaded at dword alignment; the chunk that
     * straddles the end of a row (width % 16 != 0) is read dword by dword up to ROUND_UP_4(width),
     * which the source stride always covers */
    constexpr int TPR = TW / 16;
    constexpr int RPP = NTHREADS / TPR;
    constexpr int NPASS = (NROWS + RPP - 1) / RPP;
    const int c = (tid % TPR) * 16;
    const int rr = tid / TPR;
    const int avail = p.wlimit4 - (tile_x + c);         /* readable bytes from this chunk on */
    u32x4 v[NPASS];
#pragma unroll
    for (int i = 0; i < NPASS; i++) {
      const int r = i * RPP + rr;
      const int y = tile_y - 1 + r;
      v[i] = (u32x4) (0u);
      if (r < NROWS && y <= p.height && avail > 0) {
        const uint8_t *g = src
            + (size_t) map_row (y, p.height, p.dn_last) * p.src_stride
            + tile_x + c;
        if (avail >= 16) {
          v[i] = *(const u32x4_a4 *) g;
        } else {
          const uint32_t *q = (const uint32_t *) g;
          v[i].x = q[0];
          if (avail > 4) v[i].y = q[1];
          if (avail > 8) v[i].z = q[2];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NPASS; i++) {
      const int r = i * RPP + rr;
      if (r < NROWS)
        *(u32x4 *) &lds[r * PITCH + MAIN + c] = v[i];
    }
  }
  /* halo dwords: column tile_x-4 and tile_x+TW of every staged row */
  for (int h = tid; h < 2 * NROWS; h += NTHREADS) {
    const int r = h >> 1;
    const int side = h & 1;
    const int y = tile_y - 1 + r;
    const int col = side ? tile_x + TW : tile_x - 4;
    uint32_t v = 0u;
    if (y <= p.height && col >= 0 && col < p.wlimit4)
      v = *(const uint32_t *) (src
          + (size_t) map_row (y, p.height, p.dn_last) * p.src_stride + col);
    *(uint32_t *) &lds[r * PITCH + (side ? MAIN + TW : MAIN - 4)] = v;
  }
  __syncthreads ();

  /* ---- per-wave march -------------------------------------------------------- */
  const int wave = tid >> 6;
  const int lane = tid & 63;
  const int wx = wave % WX;
  const int wy = wave / WX;
  const int xl = 256 * wx + 4 * lane;
  const int x0 = tile_x + xl;
  const bool first = (x0 == 0);
  const int lastmode = (x0 + 4 == p.width) ? 1
      : ((GENERIC && x0 + 2 == p.width) ? 2 : 0);
  const bool active = x0 < p.width;
  const uint8_t *lrow = &lds[MAIN + xl];
  /* lanes 0 / 63 take the neighbour that lives in another wave (or the halo) */
  const int edge_off = (lane == 0) ? -4 : 4;

  auto lines_of = [&](int r) -> Lines {
    const uint8_t *q = lrow + r * PITCH;
    const uint32_t c = *(const uint32_t *) q;
    uint32_t cl, cr;
    if constexpr (NEIGH == 2) {
      cl = *(const uint32_t *) (q - 4);
      cr = *(const uint32_t *) (q + 4);
    } else {
      const uint32_t edge = *(const uint32_t *) (q + edge_off);
      if constexpr (NEIGH == 0) {
        cl = from_lane_below (edge, c);
        cr = from_lane_above (edge, c);
      } else {
        cl = __shfl_up (c, 1);
        cr = __shfl_down (c, 1);
        cl = (lane == 0) ? edge : cl;
        cr = (lane == 63) ? edge : cr;
      }
    }
    return row_lines<INTRIN, GENERIC> (c, cl, cr, first, lastmode);
  };

  const int r0 = wy * RPW;      /* LDS row of up(first output row of this wave) */
  Lines up = lines_of (r0);
  Lines cur = lines_of (r0 + 1);
  uint8_t *out = dst + (size_t) (tile_y + r0) * p.dst_stride + (size_t) x0 * 4;
  const int nrows = p.height - (tile_y + r0);   /* rows of this wave inside the frame */
  /* bytes of a row this wave writes (store policy 5): 1 KiB, less for the wave that holds the end of the row */
  const int wave_left = 4 * (p.width - (tile_x + 256 * wx));
  const int wave_len = wave_left < 1024 ? wave_left : 1024;
#pragma unroll
  for (int k = 0; k < RPW; k++) {
    const Lines dn = lines_of (r0 + k + 2);
    const int type = (k & 1) ^ p.swap_rows;
    const u32x4 px = merge_rows<INTRIN> (up, cur, dn, type, p.sel);
    if (active && k < nrows) {
      if constexpr (GENERIC && (ST & 7) == 5)
        store_pixels_hybrid (out, px, lastmode, 16 * lane, wave_len);
      else
        store_pixels<ST, GENERIC> (out, px, lastmode);
    }
    out += p.dst_stride;
    up = cur;
    cur = dn;
  }
}

/* (E,O) of four columns whose source bytes start R bytes (0 or 2) behind the dword-aligned LDS address qa: the
 * column dword and its x-1 / x+1 neighbour windows all come out of the aligned dwords around it (for R = 2 out of
 * two of them), so no lane exchange and no wave-edge special case is needed.  EDGE: this wave holds the first column
 * of the frame, or its last ones (reference gstbayer2rgb.c:360-363, :372-380) */
template <int R, bool EDGE>
__device__ __forceinline__ Lines shifted_lines (const uint8_t *qa, int x, int width)
{
  const uint32_t d0 = *(const uint32_t *) qa;
  const uint32_t d1 = *(const uint32_t *) (qa + 4);
  uint32_t lsh, c, rsh;         /* columns x-1..x+2, x..x+3, x+1..x+4 */
  if constexpr (R == 0) {
    const uint32_t dm = *(const uint32_t *) (qa - 4);
    lsh = __builtin_amdgcn_alignbit (d0, dm, 24);
    c = d0;
    rsh = __builtin_amdgcn_alignbit (d1, d0, 8);
  } else {
    lsh = __builtin_amdgcn_alignbit (d1, d0, 8);
    c = __builtin_amdgcn_alignbit (d1, d0, 16);
    rsh = __builtin_amdgcn_alignbit (d1, d0, 24);
  }
  if constexpr (EDGE) {
    /* O[0] = S[1]: the left neighbour of column 0 is S[1] */
    const uint32_t lsh_first = (lsh & 0xffffff00u) | ((c >> 8) & 0xffu);
    lsh = (x == 0) ? lsh_first : lsh;
    /* E[W-1] = S[W-2], O[W-2] = S[W-3]: the right neighbours of the last two columns are their left ones */
    const uint32_t t = lsh >> 16;
    const uint32_t rsh_last = t | (t << 16);
    rsh = (x + 4 == width) ? rsh_last : rsh;
    rsh = (x + 2 == width) ? lsh : rsh;
  }
  const uint32_t a = avg4<true> (lsh, rsh);
  Lines r;
  r.e = bsel (kEvenBytes, c, a);
  r.o = bsel (kEvenBytes, a, c);
  return r;
}

template <int ST>
__device__ __forceinline__ void store_aligned16 (uint8_t *p, u32x4 px)
{
  if constexpr ((ST & 7) == 1)
    __builtin_nontemporal_store (px, (u32x4 *) p);
  else
    *(u32x4 *) p = px;
}

/* ------------------------------------------------------------------------- */
/* LDS-staged tile kernel, sector-aligned stores (generic geometries)          */
/* ------------------------------------------------------------------------- */
/* Output rows that do not start on a 64-byte sector (dst_stride % 64 != 0: every width % 16 != 0, e.g. 4056-,
 * 3838-, 1366-px frames).  In the kernel above a lane owns the SAME four columns in every row, so the 1 KiB a wave
 * stores per row starts wherever the row starts: each wave-store ends in a partial sector that a neighbouring wave
 * (or workgroup, or XCD) completes later.  Here the lane -> column map is shifted PER ROW instead: output row j
 * starts at address a_j = dst + j * dst_stride, s_j = ((-a_j) mod ALIGN) / 4 pixels, and lane l of wave wx of tile tx
 * converts columns  tile_x + 256 wx + 4 l + s_j ...+3  of that row.  Its 16-byte store then sits at
 * a_j + 4 s_j + 1024 (..) + 16 l: every wave-store is one 1 KiB run that starts on an ALIGN-byte boundary.  What is
 * left over is written once per row, by the first wave of the row: the s_j < ALIGN/4 columns in front of the first
 * boundary (the "head", at most ALIGN - 8 bytes); the row's ragged end is the last active lanes of the last tile.
 *
 * The price is arithmetic, not memory: the three source rows of an output row are looked up at that row's own
 * shift, so the (E,O) lines of a source row are built three times (once per output row that uses it) instead of
 * once -- ~45 instead of ~25 VALU instructions per 4 pixels, still a fraction of what the CUs have to spare on this
 * stream; waves that hold no frame edge in a row (all but the first and last of a row) run without the edge-column
 * selects.  A lane's column dword and both neighbour windows come out of the two or three ALIGNED LDS dwords around
 * them through v_alignbit_b32 (unaligned ds_reads were measured 2x slower), so the shifted arm needs no lane
 * exchange at all.  The tile carries ALIGN/4 + 4 more source columns on its right for it.
 *
 * Needs even shifts (a_j % 8 == 0 for every row: dst, dst_stride and the frame pitch multiples of 8); anything
 * else keeps the GENERIC arm of the kernel above.  Same arithmetic, bit-exact (tests/test_gpu_parity.py). */
template <int WX, int WY, int RPW, int ST, int ALIGN>
__global__ void __launch_bounds__ (64 * WX * WY)
bayer2rgb_lds_aligned_kernel (KParams p)
{
  constexpr int NTHREADS = 64 * WX * WY;
  constexpr int TW = 256 * WX;
  constexpr int TR = WY * RPW;
  constexpr int NROWS = TR + 2;
  constexpr int SMAX = ALIGN / 4;               /* shifts are 0, 2, .. SMAX - 2 pixels */
  /* LDS row: 12 B pad | left halo dword | TW bytes | RIGHT bytes.  The last lane of the tile at the largest shift
   * reads its right neighbour dword at bytes TW - 4 + (SMAX - 2) + 4 .. + 7 */
  constexpr int RIGHT = (SMAX + 4 + 15) & ~15;
  constexpr int NRIGHT = (SMAX + 4) / 4;        /* dwords staged right of the tile */
  constexpr int PITCH = 16 + TW + RIGHT;
  constexpr int MAIN = 16;
  static_assert (RPW % 2 == 0, "row parity is derived from the in-tile row");
  static_assert (ALIGN == 64 || ALIGN == 128, "a sector or an L2 line");

  __shared__ __attribute__ ((aligned (16))) uint8_t lds[NROWS * PITCH];

  const TileId tile = block_to_tile (blockIdx.x, p.map);
  if (!tile.valid)
    return;
  for (int z = 0; z < p.start_sleep; z++)
    __builtin_amdgcn_s_sleep (1);
  const uint32_t frame = fastdiv (tile.row, p.map.tiles_y);
  const int ty = (int) (tile.row - frame * p.map.tiles_y.d);
  const uint8_t *src = frame_src (p, frame);
  uint8_t *dst = frame_dst (p, frame);
  const int tile_x = (int) tile.tx * TW;
  const int tile_y = ty * TR;
  const int tid = threadIdx.x;

  /* ---- stage rows tile_y-1 .. tile_y+TR, columns tile_x-4 .. tile_x+TW+4*NRIGHT-1 ---------- */
  {
    constexpr int TPR = TW / 16;
    constexpr int RPP = NTHREADS / TPR;
    constexpr int NPASS = (NROWS + RPP - 1) / RPP;
    const int c = (tid % TPR) * 16;
    const int rr = tid / TPR;
    const int avail = p.wlimit4 - (tile_x + c);         /* readable bytes from this chunk on */
    u32x4 v[NPASS];
#pragma unroll
    for (int i = 0; i < NPASS; i++) {
      const int r = i * RPP + rr;
      const int y = tile_y - 1 + r;
      v[i] = (u32x4) (0u);
      if (r < NROWS && y <= p.height && avail > 0) {
        const uint8_t *g = src
            + (size_t) map_row (y, p.height, p.dn_last) * p.src_stride
            + tile_x + c;
        if (avail >= 16) {
          v[i] = *(const u32x4_a4 *) g;
        } else {
          const uint32_t *q = (const uint32_t *) g;
          v[i].x = q[0];
          if (avail > 4) v[i].y = q[1];
          if (avail > 8) v[i].z = q[2];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NPASS; i++) {
      const int r = i * RPP + rr;
      if (r < NROWS)
        *(u32x4 *) &lds[r * PITCH + MAIN + c] = v[i];
    }
    /* halo dwords: one left of the tile, NRIGHT right of it */
    for (int h = tid; h < (1 + NRIGHT) * NROWS; h += NTHREADS) {
      const int r = h / (1 + NRIGHT);
      const int k = h - r * (1 + NRIGHT);
      const int y = tile_y - 1 + r;
      const int off = k == 0 ? -4 : TW + 4 * (k - 1);
      const int col = tile_x + off;
      uint32_t hv = 0u;
      if (y <= p.height && col >= 0 && col < p.wlimit4)
        hv = *(const uint32_t *) (src
            + (size_t) map_row (y, p.height, p.dn_last) * p.src_stride + col);
      *(uint32_t *) &lds[r * PITCH + MAIN + off] = hv;
    }
  }
  __syncthreads ();

  /* ---- per-wave march -------------------------------------------------------- */
  const int wave = __builtin_amdgcn_readfirstlane (tid >> 6);
  const int lane = tid & 63;
  const int wx = wave % WX;
  const int wy = wave / WX;
  const int xl = 256 * wx + 4 * lane;   /* column inside the tile, before the row's shift */
  const int wave_x0 = tile_x + 256 * wx;
  const int r0 = wy * RPW;
  const int nrows = p.height - (tile_y + r0);   /* rows of this wave inside the frame */
  const bool head_wave = (tile.tx == 0 && wx == 0);
  const uint8_t *lrow = &lds[r0 * PITCH + MAIN + xl];

  /* one output row of this wave at shift s: convert, store */
  auto shifted_row = [&](auto rtag, auto etag, const uint8_t *qa, int x, int type, uint8_t *row, int lim) {
    constexpr int R = decltype (rtag)::value;
    constexpr bool EDGE = decltype (etag)::value;
    const Lines up = shifted_lines<R, EDGE> (qa, x, p.width);
    const Lines cur = shifted_lines<R, EDGE> (qa + PITCH, x, p.width);
    const Lines dn = shifted_lines<R, EDGE> (qa + 2 * PITCH, x, p.width);
    const u32x4 px = merge_rows<true> (up, cur, dn, type, p.sel);
    uint8_t *out = row + (size_t) x * 4;
    if constexpr (!EDGE) {
      store_aligned16<ST> (out, px);
    } else {
      /* lim: one past the last column this pass may write (the frame width, or the first boundary for the head) */
      if (x + 4 <= lim) {
        *(u32x4_a4 *) out = px;
      } else if (x + 2 == lim) {
        u32x2 two;
        two.x = px.x;
        two.y = px.y;
        *(u32x2_a4 *) out = two;
      }
    }
  };
  using R0 = std::integral_constant<int, 0>;
  using R2 = std::integral_constant<int, 2>;

#pragma unroll
  for (int k = 0; k < RPW; k++) {
    if (k >= nrows)
      break;
    uint8_t *row = dst + (size_t) (tile_y + r0 + k) * p.dst_stride;
    const int s = (int) (((0u - (uint32_t) (uintptr_t) row) & (uint32_t) (ALIGN - 1)) >> 2);
    const int type = (k & 1) ^ p.swap_rows;
    const uint8_t *qa = lrow + k * PITCH + (s & ~3);
    const int wave_x = wave_x0 + s;     /* first column of this wave in this row */
    const int x = wave_x + 4 * lane;
    if (wave_x > 0 && wave_x + 256 < p.width) {
      /* no frame edge inside this wave: every lane converts and stores four pixels */
      if (s & 2)
        shifted_row (R2 (), std::false_type (), qa, x, type, row, 0);
      else
        shifted_row (R0 (), std::false_type (), qa, x, type, row, 0);
    } else {
      if (s & 2)
        shifted_row (R2 (), std::true_type (), qa, x, type, row, p.width);
      else
        shifted_row (R0 (), std::true_type (), qa, x, type, row, p.width);
    }
    /* the head of the row: columns 0 .. s-1 in front of the first boundary, at the natural (unshifted) lane map */
    if (head_wave && s > 0)
      shifted_row (R0 (), std::true_type (), lrow + k * PITCH, 4 * lane, type, row, s < p.width ? s : p.width);
  }
}

#ifdef MIBAYER_LAB
/* ------------------------------------------------------------------------- */
/* direct kernel: no LDS, every wave streams its own strip                     */
/* ------------------------------------------------------------------------- */
/* Experiment arm (same arithmetic): rows come straight from global memory, the
 * wave-edge lanes fetch their neighbour dword with a second, 2-lane load.      */
template <int WX, int WY, int RPW, int ST, bool INTRIN, bool GENERIC>
__global__ void __launch_bounds__ (64 * WX * WY)
bayer2rgb_direct_kernel (KParams p)
{
  constexpr int TW = 256 * WX;
  constexpr int TR = WY * RPW;
  static_assert (RPW % 2 == 0, "row parity is derived from the in-tile row");

  const TileId tile = block_to_tile (blockIdx.x, p.map);
  if (!tile.valid)
    return;
  const uint32_t frame = fastdiv (tile.row, p.map.tiles_y);
  const int ty = (int) (tile.row - frame * p.map.tiles_y.d);
  const int tx = (int) tile.tx;
  const uint8_t *src = frame_src (p, frame);
  uint8_t *dst = frame_dst (p, frame);
  const int tid = threadIdx.x;
  const int wave = tid >> 6;
  const int lane = tid & 63;
  const int wx = wave % WX;
  const int wy = wave / WX;
  const int x0 = tx * TW + 256 * wx + 4 * lane;
  const int yb = ty * TR + wy * RPW;
  const bool first = (x0 == 0);
  const int lastmode = (x0 + 4 == p.width) ? 1
      : ((GENERIC && x0 + 2 == p.width) ? 2 : 0);
  const bool active = x0 < p.width;
  const bool readable = x0 < p.wlimit4;
  const int xe = (lane == 0) ? x0 - 4 : x0 + 4;
  const bool edge_lane = (lane == 0 || lane == 63) && xe >= 0
      && xe < p.wlimit4;

  uint32_t c[RPW + 2], e[RPW + 2];
#pragma unroll
  for (int r = 0; r < RPW + 2; r++) {
    const int y = yb - 1 + r;
    c[r] = 0u;
    e[r] = 0u;
    if (y <= p.height) {
      const uint8_t *row = src
          + (size_t) map_row (y, p.height, p.dn_last) * p.src_stride;
      if (readable)
        c[r] = *(const uint32_t *) (row + x0);
      if (edge_lane)
        e[r] = *(const uint32_t *) (row + xe);
    }
  }

  auto lines_of = [&](int r) -> Lines {
    const uint32_t cl = from_lane_below (e[r], c[r]);
    const uint32_t cr = from_lane_above (e[r], c[r]);
    return row_lines<INTRIN, GENERIC> (c[r], cl, cr, first, lastmode);
  };

  Lines up = lines_of (0);
  Lines cur = lines_of (1);
  uint8_t *out = dst + (size_t) yb * p.dst_stride + (size_t) x0 * 4;
  const int nrows = p.height - yb;              /* rows of this wave inside the frame */
#pragma unroll
  for (int k = 0; k < RPW; k++) {
    const Lines dn = lines_of (k + 2);
    const int type = (k & 1) ^ p.swap_rows;
    const u32x4 px = merge_rows<INTRIN> (up, cur, dn, type, p.sel);
    if (active && k < nrows)
      store_pixels<ST, GENERIC> (out, px, lastmode);
    out += p.dst_stride;
    up = cur;
    cur = dn;
  }
}

/* ------------------------------------------------------------------------- */
/* persistent, double-buffered form of the LDS kernel (experiment arm)          */
/* ------------------------------------------------------------------------- */
/* One workgroup per CU slot loops over its share of the tiles; the global loads
 * of tile n+1 are issued before tile n is computed and stored, so the HBM read
 * latency hides under the store phase instead of under other workgroups
 * (Little's-law check: the per-tile kernel keeps ~25 of 32 waves resident and
 * they spend 57 % of their life parked on memory, profiles/r01_sq_counters.md).
 * Fast path only (W % 16 == 0, aligned rows); same arithmetic, same stores.
 * MEASURED SLOWER (-12 points): one barrier per tile makes all 8 waves of a
 * workgroup load, then store, in lock step, while independent small workgroups
 * keep the memory pipes evenly fed.  Kept as an A/B arm, not used by "auto".  */
template <int WX, int WY, int RPW, int ST>
__global__ void __launch_bounds__ (64 * WX * WY)
bayer2rgb_persist_kernel (KParams p)
{
  constexpr int NTHREADS = 64 * WX * WY;
  constexpr int TW = 256 * WX;
  constexpr int TR = WY * RPW;
  constexpr int NROWS = TR + 2;
  constexpr int PITCH = TW + 32;
  constexpr int MAIN = 16;
  constexpr int TPR = TW / 16;
  constexpr int RPP = NTHREADS / TPR;
  constexpr int NPASS = (NROWS + RPP - 1) / RPP;
  static_assert (RPW % 2 == 0, "row parity is derived from the in-tile row");
  static_assert (2 * NROWS <= NTHREADS, "one halo dword per thread");

  __shared__ __attribute__ ((aligned (16))) uint8_t lds[2][NROWS * PITCH];

  /* this workgroup's tile sequence: first, first+step, ... < end (linear tile ids) */
  const uint32_t ntiles = p.map.tile_rows * p.map.tiles_x.d;
  uint32_t first, step, end;
  if (p.map.band <= 0) {
    first = blockIdx.x;
    step = gridDim.x;
    end = ntiles;
  } else {                      /* one contiguous chunk of tile rows per XCD */
    const uint32_t xcd = blockIdx.x % kNumXcd;
    const uint32_t chunk = (uint32_t) p.map.band * p.map.tiles_x.d;
    first = xcd * chunk + blockIdx.x / kNumXcd;
    step = gridDim.x / kNumXcd;
    end = (xcd + 1) * chunk < ntiles ? (xcd + 1) * chunk : ntiles;
  }
  if (first >= end)
    return;

  const int tid = threadIdx.x;
  const int c16 = (tid % TPR) * 16;
  const int rr = tid / TPR;
  const int wave = tid >> 6;
  const int lane = tid & 63;
  const int wx = wave % WX;
  const int wy = wave / WX;
  const int xl = 256 * wx + 4 * lane;
  const int edge_off = (lane == 0) ? -4 : 4;
  const int r0 = wy * RPW;

  struct Tile { const uint8_t *src; uint8_t *dst; int tile_x, tile_y; };
  auto decode = [&](uint32_t tile) -> Tile {
    const TileId id = linear_to_tile (tile, p.map);
    const uint32_t frame = fastdiv (id.row, p.map.tiles_y);
    const int ty = (int) (id.row - frame * p.map.tiles_y.d);
    const int tx = (int) id.tx;
    Tile t;
    t.src = frame_src (p, frame);
    t.dst = frame_dst (p, frame);
    t.tile_x = tx * TW;
    t.tile_y = ty * TR;
    return t;
  };

  u32x4 v[NPASS];
  uint32_t hv = 0u;
  auto issue = [&](const Tile &t) {
#pragma unroll
    for (int i = 0; i < NPASS; i++) {
      const int r = i * RPP + rr;
      const int y = t.tile_y - 1 + r;
      v[i] = (u32x4) (0u);
      if (r < NROWS && y <= p.height && t.tile_x + c16 < p.width)
        v[i] = *(const u32x4 *) (t.src
            + (size_t) map_row (y, p.height, p.dn_last) * p.src_stride
            + t.tile_x + c16);
    }
    hv = 0u;
    if (tid < 2 * NROWS) {
      const int r = tid >> 1;
      const int y = t.tile_y - 1 + r;
      const int col = (tid & 1) ? t.tile_x + TW : t.tile_x - 4;
      if (y <= p.height && col >= 0 && col < p.wlimit4)
        hv = *(const uint32_t *) (t.src
            + (size_t) map_row (y, p.height, p.dn_last) * p.src_stride + col);
    }
  };
  auto commit = [&](uint8_t *buf) {
#pragma unroll
    for (int i = 0; i < NPASS; i++) {
      const int r = i * RPP + rr;
      if (r < NROWS)
        *(u32x4 *) &buf[r * PITCH + MAIN + c16] = v[i];
    }
    if (tid < 2 * NROWS)
      *(uint32_t *) &buf[(tid >> 1) * PITCH + ((tid & 1) ? MAIN + TW : MAIN - 4)]
          = hv;
  };
  auto compute = [&](const Tile &t, const uint8_t *buf) {
    const int x0 = t.tile_x + xl;
    const bool first_lane = (x0 == 0);
    const int lastmode = (x0 + 4 == p.width) ? 1 : 0;
    const bool active = x0 < p.width;
    const uint8_t *lrow = &buf[MAIN + xl];
    auto lines_of = [&](int r) -> Lines {
      const uint8_t *q = lrow + r * PITCH;
      const uint32_t c = *(const uint32_t *) q;
      const uint32_t edge = *(const uint32_t *) (q + edge_off);
      return row_lines<true, false> (c, from_lane_below (edge, c),
          from_lane_above (edge, c), first_lane, lastmode);
    };
    Lines up = lines_of (r0);
    Lines cur = lines_of (r0 + 1);
    uint8_t *out = t.dst + (size_t) (t.tile_y + r0) * p.dst_stride
        + (size_t) x0 * 4;
    const int nrows = p.height - (t.tile_y + r0);
#pragma unroll
    for (int k = 0; k < RPW; k++) {
      const Lines dn = lines_of (r0 + k + 2);
      const int type = (k & 1) ^ p.swap_rows;
      const u32x4 px = merge_rows<true> (up, cur, dn, type, p.sel);
      if (active && k < nrows)
        store_pixels<ST, false> (out, px, lastmode);
      out += p.dst_stride;
      up = cur;
      cur = dn;
    }
  };

  Tile cur_tile = decode (first);
  issue (cur_tile);
  commit (lds[0]);
  __syncthreads ();
  int b = 0;
  for (uint32_t t = first; t < end; t += step) {
    const uint32_t nxt = t + step;
    const bool has_next = nxt < end;
    Tile next_tile = cur_tile;
    if (has_next) {
      next_tile = decode (nxt);
      issue (next_tile);                /* loads stay in flight across compute */
    }
    compute (cur_tile, lds[b]);
    if (has_next)
      commit (lds[b ^ 1]);
    __syncthreads ();
    b ^= 1;
    cur_tile = next_tile;
  }
}

#endif  /* MIBAYER_LAB */

/* ------------------------------------------------------------------------- */
/* variant table                                                               */
/* ------------------------------------------------------------------------- */

/* default block order per shape (measured on a dozen boxes, DESIGN.md "XCD map"):
 * 1024-px tiles -> band 1 (an XCD takes one full-width tile row at a time), the
 * only plan at 80-81.5 % of peak on EVERY box; narrower tiles -> identity */
#define LDS_VARIANT(name, WX, WY, RPW, NEIGH, ST, INTRIN)                      \
  { name, 256 * (WX), (WY) * (RPW), 64 * (WX) * (WY), (WX) == 4 ? 1 : 0, 0,    \
    bayer2rgb_lds_kernel<WX, WY, RPW, NEIGH, ST, INTRIN, false>,               \
    bayer2rgb_lds_kernel<WX, WY, RPW, NEIGH, ST, INTRIN, true>, nullptr, nullptr }
/* production shapes and their plain-store twins: + the sector-aligned arms for generic geometries (the 64-byte
 * flavour only in the lab build: the autotuner's candidates use the 128-byte one) */
#ifdef MIBAYER_LAB
#define ALIGNED64_ARM(WX, WY, RPW, ST) bayer2rgb_lds_aligned_kernel<WX, WY, RPW, ST, 64>
#else
#define ALIGNED64_ARM(WX, WY, RPW, ST) nullptr
#endif
#define LDS_VARIANT_AL(name, WX, WY, RPW, ST)                                  \
  { name, 256 * (WX), (WY) * (RPW), 64 * (WX) * (WY), (WX) == 4 ? 1 : 0, 0,    \
    bayer2rgb_lds_kernel<WX, WY, RPW, 0, ST, true, false>,                     \
    bayer2rgb_lds_kernel<WX, WY, RPW, 0, ST, true, true>,                      \
    ALIGNED64_ARM (WX, WY, RPW, ST),                                           \
    bayer2rgb_lds_aligned_kernel<WX, WY, RPW, ST, 128> }
#ifdef MIBAYER_LAB
#define PERSIST_VARIANT(name, WX, WY, RPW, ST)                                 \
  { name, 256 * (WX), (WY) * (RPW), 64 * (WX) * (WY), -1, 1,                   \
    bayer2rgb_persist_kernel<WX, WY, RPW, ST>,                                 \
    bayer2rgb_lds_kernel<WX, WY, RPW, 0, ST, true, true>, nullptr, nullptr }
#define DIRECT_VARIANT(name, WX, WY, RPW, ST, INTRIN)                          \
  { name, 256 * (WX), (WY) * (RPW), 64 * (WX) * (WY), -1, 0,                   \
    bayer2rgb_direct_kernel<WX, WY, RPW, ST, INTRIN, false>,                   \
    bayer2rgb_direct_kernel<WX, WY, RPW, ST, INTRIN, true>, nullptr, nullptr }
#endif

/* Measured on MI355X (profiles/sweep_r01_*.log, interleaved A/B, 4K x 64 frames):
 * 4 rows per wave and 8 waves per workgroup is the sweet spot (83-84 % of the
 * 8 TB/s HBM peak); 8 rows per wave costs 3 points, 2 rows per wave 15; nt
 * stores gain 1.3 points over plain stores, sc1 / sc0 sc1 stores lose 2; DPP,
 * ds_bpermute and LDS neighbour reads tie; the no-LDS arm loses 15 points. */
static const Variant kVariants[] = {
  /* 0: "auto" -- resolved per stream width by resolve_variant() below */
  { "auto", 0, 0, 0, -1, 0, nullptr, nullptr, nullptr, nullptr },
  /* 1-3: the production shapes (tile 1024x8, 512x16, 256x32; 512 threads), streaming (nt) stores */
  LDS_VARIANT_AL ("lds_4x2_r4_dpp_nt", 4, 2, 4, 1),
  LDS_VARIANT_AL ("lds_2x4_r4_dpp_nt", 2, 4, 4, 1),
  LDS_VARIANT_AL ("lds_1x8_r4_dpp_nt", 1, 8, 4, 1),
  /* 4-6: the same shapes with plain (write-back) stores: for output rows that start off a 64-byte sector
   * (width % 16 != 0) the L2 then completes the partial sectors two waves share before they go out */
  LDS_VARIANT_AL ("lds_4x2_r4_dpp", 4, 2, 4, 0),
  LDS_VARIANT_AL ("lds_2x4_r4_dpp", 2, 4, 4, 0),
  LDS_VARIANT_AL ("lds_1x8_r4_dpp", 1, 8, 4, 0),
  /* 7-9: the hybrid store policy (generic geometries: nt for the lines a wave-store covers completely, write-back for
   * its ragged ends; sector-aligned geometries: nt) */
  LDS_VARIANT ("lds_4x2_r4_dpp_hy", 4, 2, 4, 0, 5, true),
  LDS_VARIANT ("lds_2x4_r4_dpp_hy", 2, 4, 4, 0, 5, true),
  LDS_VARIANT ("lds_1x8_r4_dpp_hy", 1, 8, 4, 0, 5, true),
#ifdef MIBAYER_LAB
  /* 10.. : tuning / verification arms of the lab build (`make lab`), all bit-exact (tests/test_gpu_parity.py);
   * what each of them measured is in profiles/r01_sweep_*.log */
  LDS_VARIANT ("lds_1x8_r4_dpp_sc1", 1, 8, 4, 0, 2, true),
  LDS_VARIANT ("lds_1x8_r4_shfl_nt", 1, 8, 4, 1, 1, true),
  LDS_VARIANT ("lds_1x8_r4_ldsnb_nt", 1, 8, 4, 2, 1, true),
  LDS_VARIANT ("lds_1x8_r4_ldsnb_swar", 1, 8, 4, 2, 0, false),
  LDS_VARIANT ("lds_1x4_r8_dpp_nt", 1, 4, 8, 0, 1, true),
  LDS_VARIANT ("lds_1x16_r4_dpp_nt", 1, 16, 4, 0, 1, true),
  LDS_VARIANT ("lds_1x8_r2_dpp_nt", 1, 8, 2, 0, 1, true),
  LDS_VARIANT ("lds_4x1_r16_dpp_nt", 4, 1, 16, 0, 1, true),
  DIRECT_VARIANT ("direct_1x4_r8_nt", 1, 4, 8, 1, true),
  LDS_VARIANT ("lds_1x1_r4_dpp_nt", 1, 1, 4, 0, 1, true),
  LDS_VARIANT ("lds_1x4_r4_dpp_nt", 1, 4, 4, 0, 1, true),
  /* negative result kept as a verified arm: 64-68 % of peak vs 77-80 % for the
   * per-tile kernels in the same interleaved run (profiles/r01_sweep_persistent.log) */
  PERSIST_VARIANT ("persist_4x2_r4_nt", 4, 2, 4, 1),
  PERSIST_VARIANT ("persist_1x8_r4_nt", 1, 8, 4, 1),
  /* nt hint on the row loads as well (the mosaic is read once per XCD) */
  LDS_VARIANT ("lds_4x2_r4_dpp_nt_ldnt", 4, 2, 4, 0, 9, true),
  /* direct global -> LDS row loads (no staging registers, no ds_write pass) */
  LDS_VARIANT ("lds_4x2_r4_dpp_nt_glds", 4, 2, 4, 0, 17, true),
#endif
};

int variant_count ()
{
  return (int) (sizeof (kVariants) / sizeof (kVariants[0]));
}

const Variant &variant (int id)
{
  return kVariants[id];
}

/* Output rows that do not start on a 64-byte sector (4 * width, or a padded stride, not a multiple of
 * 64: 4056-px sensors, any width % 16 != 0): every wave-store of a row then ends in a partial sector that the
 * neighbouring wave -- or the neighbouring workgroup -- completes.  Streaming (nt) stores push those halves out
 * one by one; plain write-back stores let the L2 put them together first, provided both halves reach the SAME
 * L2, i.e. under the chunk-per-XCD order: 4056x3040 76.0 -> 78.9 %, 3838x2160 72.5 -> 78.5 %, 1366x768
 * 68.1 -> 73.3 % of peak (profiles/r02_generic_path.log); for sector-aligned rows nt + band 1 stays ahead
 * (80.8 vs 79.3 %).  Ids: production shape s in 1..3 -> plain-store twin s + 3, hybrid-store twin s + 6. */
int plain_store_twin (int id)
{
  return (id >= 1 && id <= 3) ? id + 3 : id;
}

/* the hybrid-store arm of a production shape (store_pixels_hybrid: nt inside, write-back at the ragged ends of a
 * wave-store), for generic geometries whose output rows are 16-byte aligned but off the line grid */
int hybrid_store_twin (int id)
{
  return (id >= 1 && id <= 3) ? id + 6 : id;
}

/* the production shape (ids 1-3) a plain-store or hybrid-store twin stands for; any other id is returned unchanged */
int production_shape_of (int id)
{
  return (id >= 4 && id <= 9) ? (id - 1) % 3 + 1 : id;
}

/* variant 0.  Rows that fit into ONE tile take the narrowest production tile that covers them (256 / 512 / 1024 px),
 * run in identity order without the start delay: 81-84 % of HBM peak for 320 ... 1024-px rows, against 50-79 % for
 * the other shapes (a second, mostly empty tile per row is what hurts: 640 px in 512-px tiles 59 %;
 * profiles/r01_sweep_narrow_frames.log).  Wider rows: 1024x8 tiles whenever they pad the frame width by no more
 * than 7 % beyond the best shape (3840 -> 4096 is fine) -- with the band-1 order they are the robust plan --
 * otherwise the production shape that wastes the fewest lanes, the widest among equals. */
int resolve_variant (int id, int width)
{
  if (id != 0)
    return id;
  if (width <= 256)
    return 3;
  if (width <= 512)
    return 2;
  if (width <= 1024)
    return 1;
  static const int tile_w[3] = { 1024, 512, 256 };      /* variants 1, 2, 3 */
  long long padded[3];
  int best = 0;
  for (int i = 0; i < 3; i++) {
    padded[i] = (long long) ((width + tile_w[i] - 1) / tile_w[i]) * tile_w[i];
    if (padded[i] < padded[best])
      best = i;
  }
  if (padded[0] * 100 <= padded[best] * 107)
    return 1;
  return best + 1;
}

/* Batch-class default for widths whose measured winner the rules above do not find.  Every entry beat the rule's pick
 * by the margin noted, on two boxes in two rounds (profiles/r04_plan_sweep_530Mpix.log, r05_plan_sweep_530Mpix.log:
 * ~530-Mpixel batches, 3 shapes x {band 1, identity}; the chunk order is left out on purpose -- whether it is fast
 * follows the allocation, DESIGN.md section 5 -- and stays the measured plan's business).  The rules cannot see these:
 * which block order collides in the memory channels depends on the row pitch in ways the sweeps show (3264 px in
 * 1024x8 tiles, identity order: 69 %; 2592 px in 512x16: 71 %) but nothing here models.  Sector-aligned geometries
 * with cfg.variant == 0 only; anything measured (mibayer_autotune, the plan cache) still overrides it. */
bool known_width_plan (int width, int *variant, int *band)
{
  static const struct { int width, variant, band; } kKnown[] = {
    { 2048, 3, 1 },     /* 256x32 band 1: 83.3 / 84.1 % against 82.3 / 82.6 (1024x8 identity) */
    { 4096, 3, 1 },     /*                83.3 / 83.7          81.0 / 81.2                     */
    { 8192, 3, 1 },     /*                81.9 / 82.7          81.1 / 82.0                     */
    { 2304, 2, 0 },     /* 512x16 identity: 80.7 / 80.6 against 78.8 / 78.8 (256x32 identity)  */
    { 3264, 2, 0 },     /*                  80.1 / 79.2          78.4 / 77.9                   */
    { 2448, 1, 0 },     /* 1024x8 identity: 79.5 / 79.8 against 77.7 / 78.3 (512x16 identity); 2448x2048: 79.1 / 76.8 */
    { 2560, 1, 0 },     /*                  81.1 / 81.2          79.2 / 78.9                   */
    { 2592, 1, 0 },     /*                  81.3 / 81.7          79.3 / 79.6 (256x32 identity) */
    { 2688, 1, 0 },     /*                  82.4 / 82.4          77.4 / 77.5 (256x32 identity) */
    { 4608, 1, 0 },     /*                  82.4 / 81.2          79.5 / 79.4 (512x16 identity) */
    { 4112, 2, 1 },     /* 512x16 band 1:   79.6 / 79.9 against 77.8 / 77.0 (256x32 identity)  */
    { 4208, 2, 1 },     /*                  78.9 / 79.8          77.1 / 77.3                   */
    { 6000, 2, 1 },     /*                  78.0 / 77.9          76.8 / 75.4 (1024x8 band 1)   */
  };
  for (const auto &k : kKnown)
    if (k.width == width) {
      *variant = k.variant;
      *band = k.band;
      return true;
    }
  return false;
}

/* Variant 0 for a launch of ONE frame (PLAN_FRAME, mibayer_abi.hip): such a launch is a handful of rounds of
 * workgroups at most -- a 4K frame in 1024x8 tiles is 1080 workgroups on the 1024 slots of 256 CUs x 4, i.e. a second
 * round that is 5 % full -- so the shape whose grid needs the fewest rounds wins (4K: 256x32 tiles = 1020 workgroups,
 * one round: 54.8 vs 47.9 % of peak launched frame by frame; 3264x2448: 53.9 vs 49.9 %; 8K: 512x16 = 3.96 rounds
 * against 4.25: 68.0 vs 64.2 %), the widest tile among equals (2592x1944, all one round: 47.1 / 45.7 / 44.6 %;
 * 4056x3040, all two: 53.6 / 51.8 / 47.7 %) -- profiles/r05_single_frame.md.  Rows that fit one tile keep the rule of
 * resolve_variant().  `slots`: workgroups resident at once (CUs x 4 for the 512-thread production shapes). */
int frame_class_variant (int width, int height, int slots)
{
  if (width <= 1024 || slots < 1)
    return resolve_variant (0, width);
  static const int tile_w[3] = { 1024, 512, 256 }, tile_h[3] = { 8, 16, 32 };      /* variants 1, 2, 3 */
  int best = 0;
  long long best_rounds = 0;
  for (int i = 0; i < 3; i++) {
    const long long tiles = (long long) ((width + tile_w[i] - 1) / tile_w[i]) * ((height + tile_h[i] - 1) / tile_h[i]);
    const long long rounds = (tiles + slots - 1) / slots;
    if (i == 0 || rounds < best_rounds) {
      best = i;
      best_rounds = rounds;
    }
  }
  return best + 1;
}

/* ------------------------------------------------------------------------- */
/* rgb2bayer: the sibling element's per-pixel gather                           */
/* ------------------------------------------------------------------------- */
/* Reference gst/bayer/gstrgb2bayer.c:254-268: output byte (j,i) is one channel
 * of input pixel (j,i), chosen by the CFA site ((j&1)<<1)|(i&1).  4 B read +
 * 1 B written per pixel: a read-dominated HBM stream.  A lane converts 4 pixels
 * (16 B in, one dword out); a 256-thread block walks R2B_ROWS rows of a
 * 1024-pixel column strip so every lane keeps several 16-byte loads in flight.
 * The two v_perm_b32 selectors per row parity come from the host. */
template <bool VEC16, int R2B_ROWS>
__global__ void __launch_bounds__ (256)
rgb2bayer_kernel (R2BParams p)
{
  /* same XCD-aware block -> tile map as the demosaic kernel: a "tile" is
   * R2B_ROWS rows x 1024 pixels, tile rows run through the whole batch */
  const TileId tile = block_to_tile (blockIdx.x, p.map);
  if (!tile.valid)
    return;
  const int xd = (int) tile.tx * 256 + threadIdx.x;     /* output dword in the row */
  if (xd >= p.out_dwords)
    return;
  for (int z = 0; z < p.start_sleep; z++)
    __builtin_amdgcn_s_sleep (1);
  const long long row0 = (long long) tile.row * R2B_ROWS;
  const int x0 = xd * 4;
  /* (frame, y) of row0; later rows only increment */
  const uint32_t f0 = fastdiv ((uint32_t) row0, p.div_height);
  const int y0 = (int) ((uint32_t) row0 - f0 * p.div_height.d);
  u32x4 px[R2B_ROWS];
#pragma unroll
  for (int k = 0; k < R2B_ROWS; k++) {
    const long long row = row0 + k;
    px[k] = (u32x4) (0u);
    if (row < p.total_rows) {
      int y = y0 + k;
      uint32_t f = f0;
      while (y >= p.height) {
        y -= p.height;
        f++;
      }
      const uint8_t *s = p.src + f * p.src_frame_bytes
          + (size_t) y * p.src_stride + (size_t) x0 * 4;
      if constexpr (VEC16) {
        px[k] = *(const u32x4 *) s;
      } else {
        const uint32_t *q = (const uint32_t *) s;
        if (x0 + 0 < p.width) px[k].x = q[0];
        if (x0 + 1 < p.width) px[k].y = q[1];
        if (x0 + 2 < p.width) px[k].z = q[2];
        if (x0 + 3 < p.width) px[k].w = q[3];
      }
    }
  }
  /* columns >= width inside the last dword are written as 0 */
  const int valid = p.width - x0;
  const uint32_t keep = valid >= 4 ? 0xffffffffu : ((1u << (8 * valid)) - 1u);
#pragma unroll
  for (int k = 0; k < R2B_ROWS; k++) {
    const long long row = row0 + k;
    if (row < p.total_rows) {
      int y = y0 + k;
      uint32_t f = f0;
      while (y >= p.height) {
        y -= p.height;
        f++;
      }
      const int par = y & 1;
      const uint32_t lo = __builtin_amdgcn_perm (px[k].y, px[k].x, p.sel_lo[par]);
      const uint32_t hi = __builtin_amdgcn_perm (px[k].w, px[k].z, p.sel_hi[par]);
      uint32_t *d = (uint32_t *) (p.dst + f * p.dst_frame_bytes
          + (size_t) y * p.dst_stride + (size_t) x0);
      __builtin_nontemporal_store ((lo | hi) & keep, d);
    }
  }
}

/* Flat form: with no neighbourhood the batch is one linear sequence of 4-pixel
 * items (16 B in -> one dword out); only the row parity (which pair of v_perm
 * selectors) and, for padded strides, the addresses depend on where an item sits.
 * No lane is wasted on a partial last tile (3840 px = 3.75 tiles of 1024 in the
 * tile kernel: 6 % idle lanes), every thread keeps K groups of loads in flight,
 * and with PX == 8 a lane owns 32 contiguous input bytes and stores 8 bytes
 * (512 B per wave-store instead of 256).  Items are decoded with two
 * multiply-shift divisions (row = item / dwords-per-row, frame = row / height). */
/* VEC = 16: width % 4 == 0 and 16-byte aligned frames -- every item is full and loaded
 * unconditionally.  VEC = 4: any other geometry -- full items still take one 16-byte load (at
 * dword alignment, which is all a gfx950 global load needs), the partial last item of a row is
 * read dword by dword */
/* The frame index is block-uniform and the table sits in the kernel arguments: read it straight out of the kernarg
 * segment (a scalar load at a run-time offset).  Indexing the by-value copy `p` instead made the compiler keep the
 * whole 480-byte argument block in scratch memory in this kernel (488 bytes per lane, 20x slower): R2BParams is the
 * kernel's only argument, so it starts at offset 0 of the segment. */
template <typename T>
__device__ __forceinline__ T kernarg_table_entry (size_t table_offset, uint32_t index)
{
  typedef const __attribute__ ((address_space (4))) char *kptr;
  kptr base = (kptr) __builtin_amdgcn_kernarg_segment_ptr ();
  return ((const __attribute__ ((address_space (4))) T *) (base + table_offset))[index];
}

template <int K, int PX, int LD, int VEC>
__global__ void __launch_bounds__ (256)
rgb2bayer_flat_kernel (R2BParams p)
{
  constexpr bool VEC16 = VEC == 16;
  constexpr int IPG = PX / 4;           /* items per group */
  const TileId tile = block_to_tile (blockIdx.x, p.map);       /* tiles_x == 1: .row = logical block */
  if (!tile.valid)
    return;
  for (int z = 0; z < p.start_sleep; z++)
    __builtin_amdgcn_s_sleep (1);
  /* list launch: this block's frame (block-uniform) and its first item inside that frame */
  const uint32_t lframe = p.nlist ? fastdiv (tile.row, p.div_blocks_per_frame) : 0u;
  const uint32_t lblock = tile.row - lframe * p.div_blocks_per_frame.d;
  const uint8_t *src_base = p.nlist
      ? kernarg_table_entry<const uint8_t *> (offsetof (R2BParams, src_list), lframe) : p.src;
  uint8_t *dst_base = p.nlist
      ? kernarg_table_entry<uint8_t *> (offsetof (R2BParams, dst_list), lframe) : p.dst;
  const uint32_t first = p.item0 + (lblock * (uint32_t) (256 * K) + threadIdx.x) * IPG;
  u32x4 px[K][IPG];
  uint32_t par[K];
  size_t doff[K];               /* byte offset into dst_base (kept as an offset: the stores stay global_store) */
  bool valid[K];
#pragma unroll
  for (int k = 0; k < K; k++) {
    const uint32_t item = first + (uint32_t) k * 256u * IPG;
#pragma unroll
    for (int h = 0; h < IPG; h++)
      px[k][h] = (u32x4) (0u);
    par[k] = 0;
    doff[k] = 0;
    valid[k] = item < p.item_end;
    if (valid[k]) {
      const uint32_t row = fastdiv (item, p.div_out_dwords);
      const uint32_t xd = item - row * p.div_out_dwords.d;
      /* list launch: items count inside the block's own frame, so row == y */
      const uint32_t f = p.nlist ? 0u : fastdiv (row, p.div_height);
      const uint32_t y = row - f * p.div_height.d;
      par[k] = y & 1u;
      const uint8_t *s = src_base + f * p.src_frame_bytes + (size_t) y * p.src_stride
          + (size_t) xd * 16;
      doff[k] = f * p.dst_frame_bytes + (size_t) y * p.dst_stride + (size_t) xd * 4;
#pragma unroll
      for (int h = 0; h < IPG; h++) {
        if constexpr (VEC16) {
          if constexpr (LD == 1)
            px[k][h] = __builtin_nontemporal_load ((const u32x4 *) s + h);
          else
            px[k][h] = ((const u32x4 *) s)[h];
        } else {
          const uint32_t *q = (const uint32_t *) s + 4 * h;
          const int x0 = (int) (xd + h) * 4;
          if (x0 + 3 < p.width) {
            /* a full item: one 16-byte load at the alignment the rows have (gfx950 global
             * loads need no more than dword alignment) */
            typedef uint32_t u32x4_a4 __attribute__ ((ext_vector_type (4), aligned (4)));
            if constexpr (LD == 1)
              px[k][h] = __builtin_nontemporal_load ((const u32x4_a4 *) q);
            else
              px[k][h] = *(const u32x4_a4 *) q;
          } else {
            if (x0 + 0 < p.width) px[k][h].x = q[0];
            if (x0 + 1 < p.width) px[k][h].y = q[1];
            if (x0 + 2 < p.width) px[k][h].z = q[2];
            if (x0 + 3 < p.width) px[k][h].w = q[3];
          }
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < K; k++) {
    if (valid[k]) {
      /* both selector pairs live in SGPRs: a select, not an indexed kernarg load */
      const uint32_t sel_lo = par[k] ? p.sel_lo[1] : p.sel_lo[0];
      const uint32_t sel_hi = par[k] ? p.sel_hi[1] : p.sel_hi[0];
      uint32_t out[IPG];
#pragma unroll
      for (int h = 0; h < IPG; h++) {
        /* pixels that were not loaded (x >= width) are zero, and so are their output bytes */
        const uint32_t lo = __builtin_amdgcn_perm (px[k][h].y, px[k][h].x, sel_lo);
        const uint32_t hi = __builtin_amdgcn_perm (px[k][h].w, px[k][h].z, sel_hi);
        out[h] = lo | hi;
      }
      uint8_t *d = dst_base + doff[k];
      if constexpr (IPG == 2) {
        u32x2 v;
        v.x = out[0];
        v.y = out[1];
        __builtin_nontemporal_store (v, (u32x2 *) d);
      } else {
        __builtin_nontemporal_store (out[0], (uint32_t *) d);
      }
    }
  }
}

typedef void (*R2BFn) (R2BParams);

template <int VEC>
static R2BFn flat_kernel_for (int k, int px, int ld)
{
#define R2B_FLAT(K, PX, LD) if (k == K && px == PX && ld == LD) return rgb2bayer_flat_kernel<K, PX, LD, VEC>
#ifndef MIBAYER_LAB
  R2B_FLAT (2, 4, 1);           /* the production launch shape (profiles/r02_rgb2bayer_sweep.log) */
#else
  R2B_FLAT (1, 4, 0); R2B_FLAT (1, 4, 1); R2B_FLAT (1, 8, 0); R2B_FLAT (1, 8, 1);
  R2B_FLAT (2, 4, 0); R2B_FLAT (2, 4, 1); R2B_FLAT (2, 8, 0); R2B_FLAT (2, 8, 1);
  R2B_FLAT (3, 4, 0); R2B_FLAT (3, 4, 1); R2B_FLAT (3, 8, 0); R2B_FLAT (3, 8, 1);
  R2B_FLAT (4, 4, 0); R2B_FLAT (4, 4, 1); R2B_FLAT (4, 8, 0); R2B_FLAT (4, 8, 1);
  R2B_FLAT (8, 4, 0); R2B_FLAT (8, 4, 1); R2B_FLAT (8, 8, 0); R2B_FLAT (8, 8, 1);
#endif
#undef R2B_FLAT
  return nullptr;
}

static bool aligned_to (const void *ptr, unsigned a)
{
  return (((uintptr_t) ptr) & (a - 1)) == 0;
}

hipError_t launch_rgb2bayer (const R2BParams &p, bool vec16, hipStream_t stream,
    long long row0, long long nrows)
{
  if (p.total_rows <= 0 || p.out_dwords <= 0 || nrows == 0)
    return hipSuccess;
  R2BParams q = p;
  if (nrows < 0) {
    row0 = 0;
    nrows = p.total_rows;
  }
  if (row0 < 0 || (row0 & 15) || row0 + nrows > p.total_rows)
    return hipErrorInvalidValue;
  if (p.total_rows > 0x7fffffffLL)
    return hipErrorInvalidValue;
  q.div_height = make_fastdiv ((uint32_t) p.height);
  q.div_out_dwords = make_fastdiv ((uint32_t) p.out_dwords);

  /* ---- flat kernel ---------------------------------------------------------- */
  const long long item_end = (row0 + nrows) * p.out_dwords;
  if (q.flat_k > 0 && item_end <= 0x7fffffffLL) {
    int px = q.flat_px == 8 ? 8 : 4;
    /* two items per group: they must be in one row and the 8-byte store aligned */
    if (px == 8 && ((p.out_dwords & 1) || (p.dst_stride & 7) || (p.dst_frame_bytes & 7)
            || !aligned_to (p.dst, 8) || !vec16))
      px = 4;
    R2BFn fn = vec16 ? flat_kernel_for<16> (q.flat_k, px, q.flat_ld ? 1 : 0)
        : flat_kernel_for<4> (q.flat_k, px, q.flat_ld ? 1 : 0);
    if (fn) {
      q.item0 = (uint32_t) (row0 * p.out_dwords);
      q.item_end = (uint32_t) item_end;
      const long long items = item_end - q.item0;
      const long long per_block = 256LL * q.flat_k * (px / 4);
      const long long nblocks = (items + per_block - 1) / per_block;
      if (q.band < 0)           /* one contiguous chunk of the launch per XCD */
        q.band = (int) ((nblocks + kNumXcd - 1) / kNumXcd);
      const long long grid = grid_blocks_for (1, nblocks, q.band);
      if (grid > 0x7fffffffLL)
        return hipErrorInvalidValue;
      q.map = make_tile_map (1, 1, nblocks, q.band, 0);
      hipLaunchKernelGGL (fn, dim3 ((unsigned) grid), dim3 (256), 0, stream, q);
      return hipGetLastError ();
    }
  }

  /* ---- tile kernel ------------------------------------------------------------ */
  q.total_rows = row0 + nrows;  /* the kernel's "row < total_rows" guard ends the band */
#ifdef MIBAYER_LAB
  const int R2B_ROWS = (q.rows == 4 || q.rows == 8 || q.rows == 16) ? q.rows : 2;
#else
  const int R2B_ROWS = 2;       /* the product build carries the tile kernel as the fallback for batches too long for
                                   the flat kernel's 32-bit item index, in one shape */
#endif
  const long long tile_rows = (nrows + R2B_ROWS - 1) / R2B_ROWS;
  const int tiles_x = (p.out_dwords + 255) / 256;
  if (q.band < 0)               /* one contiguous chunk of tile rows per XCD */
    q.band = (int) ((tile_rows + kNumXcd - 1) / kNumXcd);
  const long long grid = grid_blocks_for (tiles_x, tile_rows, q.band);
  if (grid > 0x7fffffffLL)
    return hipErrorInvalidValue;
  q.map = make_tile_map (tiles_x, 1, tile_rows, q.band, row0 / R2B_ROWS);
#define R2B_LAUNCH(V, R) hipLaunchKernelGGL ((rgb2bayer_kernel<V, R>), \
      dim3 ((unsigned) grid), dim3 (256), 0, stream, q)
  switch (R2B_ROWS * 2 + (vec16 ? 1 : 0)) {
    case 2 * 2 + 1: R2B_LAUNCH (true, 2); break;
#ifndef MIBAYER_LAB
    default: R2B_LAUNCH (false, 2); break;
#else
    case 2 * 2 + 0: R2B_LAUNCH (false, 2); break;
    case 4 * 2 + 1: R2B_LAUNCH (true, 4); break;
    case 4 * 2 + 0: R2B_LAUNCH (false, 4); break;
    case 8 * 2 + 1: R2B_LAUNCH (true, 8); break;
    case 8 * 2 + 0: R2B_LAUNCH (false, 8); break;
    case 16 * 2 + 1: R2B_LAUNCH (true, 16); break;
    default: R2B_LAUNCH (false, 16); break;
#endif
  }
#undef R2B_LAUNCH
  return hipGetLastError ();
}

/* reference loop per frame: gst/bayer/gstrgb2bayer.c:254-268; here up to kMaxList frames, each its own allocation,
 * in one launch of the flat kernel */
hipError_t launch_rgb2bayer_list (const R2BParams &p, bool vec16, bool dst8, hipStream_t stream)
{
  if (p.nlist <= 0 || p.height <= 0 || p.out_dwords <= 0)
    return hipSuccess;
  if (p.nlist > kMaxList || p.flat_k <= 0)
    return hipErrorInvalidValue;
  R2BParams q = p;
  q.total_rows = p.height;
  q.div_height = make_fastdiv ((uint32_t) p.height);
  q.div_out_dwords = make_fastdiv ((uint32_t) p.out_dwords);
  const long long items = (long long) p.height * p.out_dwords;
  if (items > 0x7fffffffLL)
    return hipErrorInvalidValue;
  int px = q.flat_px == 8 ? 8 : 4;
  if (px == 8 && ((p.out_dwords & 1) || (p.dst_stride & 7) || !dst8 || !vec16))
    px = 4;
  R2BFn fn = vec16 ? flat_kernel_for<16> (q.flat_k, px, q.flat_ld ? 1 : 0)
      : flat_kernel_for<4> (q.flat_k, px, q.flat_ld ? 1 : 0);
  if (!fn)
    return hipErrorInvalidValue;
  q.item0 = 0;
  q.item_end = (uint32_t) items;
  const long long per_block = 256LL * q.flat_k * (px / 4);
  const long long bpf = (items + per_block - 1) / per_block;
  q.div_blocks_per_frame = make_fastdiv ((uint32_t) bpf);
  const long long nblocks = bpf * p.nlist;
  if (q.band < 0)
    q.band = (int) ((nblocks + kNumXcd - 1) / kNumXcd);
  const long long grid = grid_blocks_for (1, nblocks, q.band);
  if (grid > 0x7fffffffLL)
    return hipErrorInvalidValue;
  q.map = make_tile_map (1, 1, nblocks, q.band, 0);
  hipLaunchKernelGGL (fn, dim3 ((unsigned) grid), dim3 (256), 0, stream, q);
  return hipGetLastError ();
}

/* ------------------------------------------------------------------------- */
/* stall drill                                                                 */
/* ------------------------------------------------------------------------- */
/* One wave that does nothing for `ticks` ticks of the 100 MHz wall clock: occupies a queue the way a device that
 * has stopped answering does, and ends by itself (mibayer_internal_stall). */
__global__ void __launch_bounds__ (64)
stall_kernel (unsigned long long ticks)
{
  const unsigned long long t0 = wall_clock64 ();
  while (wall_clock64 () - t0 < ticks)
    __builtin_amdgcn_s_sleep (127);
}

hipError_t launch_stall (int ms, hipStream_t stream)
{
  hipLaunchKernelGGL (stall_kernel, dim3 (1), dim3 (64), 0, stream,
      (unsigned long long) ms * 100000ull);
  return hipGetLastError ();
}

/* ------------------------------------------------------------------------- */
/* synthetic mosaic (counter-based, stateless per byte)                        */
/* ------------------------------------------------------------------------- */

__device__ __forceinline__ uint32_t fmix32 (uint32_t z)
{
  z ^= z >> 16;
  z *= 0x85EBCA6Bu;
  z ^= z >> 13;
  z *= 0xC2B2AE35u;
  z ^= z >> 16;
  return z;
}

/* byte(f,y,x) = fmix32((f*H*W + y*W + x) * 2654435761 + seed*0x9E3779B9) & 0xff; padding columns are 0.
 * Grid (x: 16-column groups, y: rows, z: frames): no division anywhere -- the round-1 form decoded a flat index with two
 * 64-bit divisions per dword and took 9.2-10.8 us per 4K frame (0.9 TB/s of stores), as long as the demosaic of that
 * frame (profiles/r06_element_host.md); a thread now writes four dwords of one row, 64 dwords apart. */
__global__ void __launch_bounds__ (256)
fill_synthetic_kernel (uint8_t *buf, int width, int height, int stride,
    unsigned long long frame_bytes, uint32_t first_frame, uint32_t frame0, int y_base, uint32_t seed)
{
  const int y = y_base + (int) (blockIdx.y * blockDim.y + threadIdx.y);
  const uint32_t f = frame0 + blockIdx.z;
  if (y >= height)
    return;
  const uint32_t row_base = (first_frame + f) * (uint32_t) height * (uint32_t) width + (uint32_t) y * (uint32_t) width;
  const uint32_t salt = seed * 0x9E3779B9u;
  uint32_t *row = (uint32_t *) (buf + (unsigned long long) f * frame_bytes + (size_t) y * stride);
  /* four dwords per thread, 64 dwords apart: every wave-store is 256 contiguous bytes */
#pragma unroll
  for (int d = 0; d < 4; d++) {
    const int xd = (int) blockIdx.x * 256 + d * 64 + (int) threadIdx.x;        /* dword in the row */
    const int x = xd * 4;
    if (x >= stride)
      break;
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (x + k < width)
        v |= (fmix32 ((row_base + (uint32_t) (x + k)) * 2654435761u + salt) & 0xffu) << (8 * k);
    row[xd] = v;
  }
}

hipError_t launch_fill_synthetic (uint8_t *d_buf, int width, int height,
    int stride, unsigned long long frame_bytes, uint32_t first_frame,
    int nframes, uint32_t seed, hipStream_t stream)
{
  if (nframes <= 0 || height <= 0 || stride <= 0)
    return hipSuccess;
  const dim3 block (64, 4);
  const unsigned gx = (unsigned) (((stride + 15) / 16 + 63) / 64);
  /* rows and frames in chunks the grid's y / z limits (65535) allow */
  for (int y0 = 0; y0 < height; y0 += 65535 * 4) {
    const int rows = height - y0 < 65535 * 4 ? height - y0 : 65535 * 4;
    for (int f0 = 0; f0 < nframes; f0 += 65535) {
      const int nf = nframes - f0 < 65535 ? nframes - f0 : 65535;
      hipLaunchKernelGGL (fill_synthetic_kernel, dim3 (gx, (unsigned) ((rows + 3) / 4), (unsigned) nf), block, 0, stream,
          d_buf, width, height, stride, frame_bytes, first_frame, (uint32_t) f0, y0, seed);
      const hipError_t e = hipGetLastError ();
      if (e != hipSuccess)
        return e;
    }
  }
  return hipSuccess;
}

}  /* namespace mibayer */
