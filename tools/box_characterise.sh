#!/bin/bash
# Development tool: which box did we land on, and how do the three block orders the autotuner chooses between run on it?
set +e
export TMPDIR=/tmp
rocm-smi --showserial 2>/dev/null | grep Serial
python tools/sweep2.py 3840 2160 64 8 lds_4x2_r4_dpp_nt lds_4x2_r4_dpp_nt:-1 lds_1x8_r4_dpp_nt:-1 lds_1x8_r4_dpp_nt:0 2>&1 | grep -v "d_src\|interleaved"
