#!/bin/bash
set +e
export TMPDIR=/tmp
rocm-smi --showserial 2>/dev/null | grep Serial
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "variant or fuzz or randomised or golden or batch_launch" 2>&1 | tail -4
A="lds_4x2_r4_dpp_nt lds_4x2_r4_dpp_nt:0 lds_1x8_r4_dpp_nt lds_1x8_r4_dpp_nt:0 persist_4x2_r4_nt persist_4x2_r4_nt:0 persist_1x8_r4_nt persist_1x8_r4_nt:0 persist_2x4_r4_nt persist_4x2_r8_nt"
for wgs in 4 2 3; do
  echo "== persistent WGs per CU: $wgs"
  MIBAYER_PERSIST_WGS=$wgs timeout 300 python tools/sweep2.py 3840 2160 64 10 $A 2>&1 | grep -v d_src
done | tee gpurun_out/persist.log
