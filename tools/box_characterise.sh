#!/bin/bash
# Development tool: characterise the box we landed on -- serial, XCD-affine write-stream probe, the 4 autotune arms.
set +e
export TMPDIR=/tmp
rocm-smi --showserial 2>/dev/null | grep Serial
./tools/xcd_stream_probe | sed -n '1,3p;9p;13p;21p'
./tools/hbm_probe 10 32768 | head -7
python tools/sweep2.py 3840 2160 64 8 lds_4x2_r4_dpp_nt lds_4x2_r4_dpp_nt:0 lds_1x8_r4_dpp_nt lds_1x8_r4_dpp_nt:0 2>&1 | grep -v "d_src\|interleaved"
