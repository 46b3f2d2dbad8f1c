#!/bin/bash
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
A="lds_1x8_r4_dpp_nt:0 lds_1x8_r4_dpp_nt:1 lds_1x8_r4_dpp_nt:2 lds_1x8_r4_dpp_nt:4 lds_1x8_r4_dpp_nt:16 lds_1x8_r4_dpp_nt:-1 lds_4x2_r4_dpp_nt:0 lds_4x2_r4_dpp_nt:1 lds_4x2_r4_dpp_nt:4 lds_4x2_r4_dpp_nt:-1 lds_2x4_r4_dpp_nt:0 lds_2x4_r4_dpp_nt:1 lds_2x4_r4_dpp_nt:-1 lds_1x4_r8_dpp_nt:0 lds_1x4_r8_dpp_nt:-1 lds_1x16_r4_dpp_nt:4 lds_1x16_r4_dpp_nt:-1 lds_1x8_r4_dpp:-1 lds_1x8_r4_ldsnb_nt:-1"
echo "== 4K"; timeout 600 python tools/sweep2.py 3840 2160 64 12 $A 2>&1 | tee gpurun_out/sweep5_4k.log
echo "== 1080p"; timeout 600 python tools/sweep2.py 1920 1080 256 8 $A 2>&1 | tee gpurun_out/sweep5_1080p.log
echo "== 8K"; timeout 600 python tools/sweep2.py 7680 4320 16 8 $A 2>&1 | tee gpurun_out/sweep5_8k.log
echo "== 4K again"; timeout 600 python tools/sweep2.py 3840 2160 64 12 $A 2>&1 | tee gpurun_out/sweep5_4k_b.log
