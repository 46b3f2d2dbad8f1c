#!/bin/bash
set +e
export TMPDIR=/tmp
rocm-smi --showserial 2>/dev/null | grep Serial
A=""
for b in -1 1080 540 270 135 64 32 8 0; do A="$A lds_4x2_r4_dpp_nt:$b"; done
for b in -1 272 136 68 34 0; do A="$A lds_1x8_r4_dpp_nt:$b"; done
timeout 600 python tools/sweep2.py 3840 2160 64 10 $A 2>&1
