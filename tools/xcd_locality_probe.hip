// Does an XCD reach all of the HBM equally fast?  (NPS1 / SPX: the address map is said to interleave every stack finely,
// in which case the answer is yes and the table below is flat.)  One physically contiguous arena; a streaming kernel in
// which only the workgroups of ONE XCD (block b runs on XCD b % 8; checked through HW_REG_XCC_ID) touch memory reads, then
// writes, one 256 MiB region at a time.  Output: GB/s per (XCD, region).
// Build: hipcc -O2 --offload-arch=gfx950 tools/xcd_locality_probe.hip -o tools/xcd_locality_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf (stderr, "%s: %s\n", #x, hipGetErrorString (e_)); exit (1); } } while (0)

typedef unsigned int u32x4 __attribute__ ((ext_vector_type (4)));

__global__ void __launch_bounds__ (256) stream_one_xcd (u32x4 *base, size_t n16, int xcd, int write, unsigned *wrong, u32x4 *sink)
{
  if ((int) (blockIdx.x & 7) != xcd)
    return;
  unsigned xcc;
  asm volatile ("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s" (xcc));
  if ((int) (xcc & 7) != xcd && threadIdx.x == 0)
    atomicAdd (wrong, 1u);
  const size_t nb = gridDim.x >> 3, b = blockIdx.x >> 3;
  u32x4 acc = { 0, 0, 0, 0 };
  for (size_t i = b * 256 + threadIdx.x; i < n16; i += nb * 256) {
    if (write)
      __builtin_nontemporal_store (u32x4 { (unsigned) i, 1, 2, 3 }, base + i);
    else
      acc += base[i];
  }
  if (!write && acc.x == 0x12345678u)
    *sink = acc;
}

int main (int argc, char **argv)
{
  const size_t region = (size_t) 256 << 20;
  const int nreg = argc > 1 ? atoi (argv[1]) : 96;
  const bool contiguous = !(argc > 2 && argv[2][0] == 'p');
  char *arena = nullptr;
  if (!contiguous || hipExtMallocWithFlags ((void **) &arena, region * nreg, hipDeviceMallocContiguous) != hipSuccess) {
    (void) hipGetLastError ();
    CK (hipMalloc ((void **) &arena, region * nreg));
    printf ("# arena: %d x 256 MiB by hipMalloc\n", nreg);
  } else {
    printf ("# arena: %d x 256 MiB, physically contiguous\n", nreg);
  }
  CK (hipMemset (arena, 1, region * nreg));
  unsigned *wrong; u32x4 *sink;
  CK (hipMalloc (&wrong, 4)); CK (hipMemset (wrong, 0, 4)); CK (hipMalloc (&sink, 16));
  hipEvent_t e0, e1;
  CK (hipEventCreate (&e0)); CK (hipEventCreate (&e1));
  const unsigned grid = 8 * 32 * 8;       /* 256 workgroups on the chosen XCD: 8 per CU */
  if (argc > 3 && argv[3][0] == 't') {
    /* time series: ONE region written over and over by XCD 0 -- is the ~3 % ripple of the table a function of the
     * position (then this series is flat) or of time (then it shows here too)? */
    printf ("# XCD 0 writing region 0, then region 5, 160 launches each, GB/s\n");
    for (int r : { 0, 5 }) {
      for (int i = 0; i < 160; i++) {
        CK (hipEventRecord (e0));
        stream_one_xcd<<<grid, 256>>> ((u32x4 *) (arena + region * r), region / 16, 0, 1, wrong, sink);
        CK (hipEventRecord (e1));
        CK (hipEventSynchronize (e1));
        float ms; CK (hipEventElapsedTime (&ms, e0, e1));
        printf (" %4.0f", region / (ms * 1e-3) / 1e9);
      }
      printf ("\n");
    }
    return 0;
  }
  for (int write = 0; write < 2; write++) {
    printf ("# %s GB/s: rows = XCD, columns = 256 MiB region of the arena\n", write ? "write (nt)" : "read");
    for (int xcd = 0; xcd < 8; xcd++) {
      printf ("xcd%d", xcd);
      for (int r = 0; r < nreg; r++) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
          CK (hipEventRecord (e0));
          stream_one_xcd<<<grid, 256>>> ((u32x4 *) (arena + region * r), region / 16, xcd, write, wrong, sink);
          CK (hipEventRecord (e1));
          CK (hipEventSynchronize (e1));
          float ms; CK (hipEventElapsedTime (&ms, e0, e1));
          if (ms < best) best = ms;
        }
        printf (" %4.0f", region / (best * 1e-3) / 1e9);
      }
      printf ("\n");
    }
  }
  unsigned w; CK (hipMemcpy (&w, wrong, 4, hipMemcpyDeviceToHost));
  printf ("# workgroups that ran on another XCD than blockIdx %% 8: %u\n", w);
  return 0;
}
