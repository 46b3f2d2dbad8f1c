#!/bin/bash
set +e
export TMPDIR=/tmp
for i in 1 2 3 4 5 6 7 8; do
  timeout 300 python tools/sweep2.py 3840 2160 64 8 lds_4x2_r4_dpp_nt lds_4x2_r4_dpp_nt:0 lds_1x8_r4_dpp_nt lds_1x8_r4_dpp_nt:0 2>&1 | grep -v "^3840"
done
