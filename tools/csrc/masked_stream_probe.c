/* Development probe: does creating / destroying CU-masked streams (hipExtStreamCreateWithCUMask, the frame queues of
 * mibayer_ctx_frame_queue) ever hang?  modes: 0 = create 4 + destroy unused; 1 = create, launch nothing, leave them to
 * process exit; 2 = create, use each (memset), sync, destroy; 3 = create, use two of four, destroy without sync */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

int main (int argc, char **argv)
{
  const int mode = argc > 1 ? atoi (argv[1]) : 0;
  hipStream_t q[4] = { 0 };
  uint32_t mask[8];
  void *buf = NULL;
  int i;
  for (i = 0; i < 8; i++)
    mask[i] = 0xffffffffu;
  if (hipSetDevice (0) != hipSuccess || hipMalloc (&buf, 1 << 24) != hipSuccess)
    return 2;
  for (i = 0; i < 4; i++)
    if (hipExtStreamCreateWithCUMask (&q[i], 8, mask) != hipSuccess)
      return 3;
  if (mode == 2)
    for (i = 0; i < 4; i++) {
      hipMemsetAsync (buf, i, 1 << 24, q[i]);
      hipStreamSynchronize (q[i]);
    }
  if (mode == 3)
    for (i = 0; i < 2; i++)
      hipMemsetAsync (buf, i, 1 << 24, q[i]);
  if (mode != 1)
    for (i = 0; i < 4; i++)
      if (hipStreamDestroy (q[i]) != hipSuccess)
        return 4;
  hipFree (buf);
  printf ("mode %d ok\n", mode);
  return 0;
}
