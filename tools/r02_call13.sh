#!/bin/bash
set +e
echo "== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -x -q --timeout 240 2>&1 | tail -5
for g in "4056 3040 32" "3838 2160 64" "1366 768 512"; do
  for lib in tools/libmibayer_before_generic.so gst-plugins-bad_amd/libmibayer.so; do
    echo "== $g  $lib"
    MIBAYER_LIB_PATH=$PWD/$lib timeout 300 python tools/sweep2.py $g 5 lds_4x2_r4_dpp_nt:1 lds_4x2_r4_dpp_nt:0 lds_2x4_r4_dpp_nt:0 lds_1x8_r4_dpp_nt:0 lds_1x8_r4_dpp_nt:-1 2>&1 | grep -v "^d_src"
  done
done
