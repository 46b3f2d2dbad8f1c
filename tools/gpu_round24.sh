#!/bin/bash
set +e
export TMPDIR=/tmp
rocm-smi --showserial 2>/dev/null | grep Serial
for n in 1 2 4 8 16; do
  timeout 300 python tools/sweep2.py 3840 2160 $n 10 lds_4x2_r4_dpp_nt lds_4x2_r4_dpp_nt:0 lds_2x4_r4_dpp_nt lds_1x8_r4_dpp_nt lds_1x8_r4_dpp_nt:0 lds_1x8_r2_dpp_nt lds_1x4_r8_dpp_nt 2>&1 | grep -v d_src
done | tee gpurun_out/small_batch.log
