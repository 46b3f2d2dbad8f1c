#!/bin/bash
set +e
export TMPDIR=/tmp
rocm-smi --showserial 2>/dev/null | grep Serial
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "variant" 2>&1 | tail -2
A="lds_4x2_r4_dpp_nt lds_4x2_r4_dpp_nt:0 lds_1x8_r4_dpp_nt lds_1x8_r4_dpp_nt:0 lds_1x1_r4_dpp_nt lds_1x1_r4_dpp_nt:0 lds_1x2_r4_dpp_nt lds_1x2_r4_dpp_nt:0 lds_2x1_r4_dpp_nt lds_1x4_r4_dpp_nt lds_1x4_r4_dpp_nt:0 lds_4x1_r4_dpp_nt lds_4x1_r4_dpp_nt:0 lds_1x1_r8_dpp_nt"
timeout 300 python tools/sweep2.py 3840 2160 64 10 $A 2>&1 | grep -v d_src | tee gpurun_out/small_wg.log
