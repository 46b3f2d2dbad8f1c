#!/bin/bash
# Evidence pass for the generic path (output rows off the 64-byte sector grid): where the loss against the
# sector-aligned path comes from, and what each store arm does about it.  Interleaved A/B (tools/sweep2.py), one box.
# Usage (GPU box): bash tools/generic_path_evidence.sh > gpurun_out/r03_generic_path.log
cd ${GRAFT_REPO_ROOT:-.}
rocm-smi --showserial | grep Serial
sw() { python tools/sweep2.py "$@" 2>&1 | grep -v "^d_src"; }
A="lds_4x2_r4_dpp_nt:1::0 lds_4x2_r4_dpp_nt:-1::0 lds_4x2_r4_dpp:-1::0 lds_4x2_r4_dpp_hy:1::0 lds_1x8_r4_dpp_nt:-1::0 lds_1x8_r4_dpp:-1::0 lds_1x8_r4_dpp_nt:0::0 lds_1x8_r4_dpp_nt:0::128"
echo "################ 1. decomposition: 3838x2160 x 64 (arm = variant:band::shift)"
echo "## 3840 px, the 16-byte path (reference)"; sw 3840 2160 64 5 $A
echo "## 3838 px, natural pitches (src 3840, dst 15352)"; sw 3838 2160 64 5 $A
echo "## 3838 px, rows on the grid (src 3840, dst 15360): ragged width only"; SWEEP_SRC_STRIDE=3840 SWEEP_DST_STRIDE=15360 sw 3838 2160 64 5 $A
echo "## 3838 px, source rows misaligned only (dst 15360)"; SWEEP_DST_STRIDE=15360 sw 3838 2160 64 5 $A
echo "## 3838 px, destination rows misaligned only (src 3840)"; SWEEP_SRC_STRIDE=3840 sw 3838 2160 64 5 $A
echo "################ 2. what the shifted arm costs when there is nothing to shift: 3840 px through the generic kernels, s = 0 in every row"
MIBAYER_FORCE_GENERIC=1 MIBAYER_FORCE_ALIGNED_ARM=1 sw 3840 2160 64 5 lds_4x2_r4_dpp_nt:1::0 lds_4x2_r4_dpp_nt:1::64 lds_2x4_r4_dpp_nt:1::0 lds_2x4_r4_dpp_nt:1::64 lds_1x8_r4_dpp_nt:0::0 lds_1x8_r4_dpp_nt:0::64
echo "################ 3. store arms x block orders per geometry (top 10 of 29 arms)"
for geo in "4056 3040 32" "2028 1520 128" "1928 1084 256" "1000 600 512" "3838 2160 64" "1366 768 512" "2590 1942 64" "4098 2170 32"; do
  arms=""
  for sh in 4x2 2x4 1x8; do
    for b in 1 -1 0; do arms="$arms lds_${sh}_r4_dpp_hy:$b::0"; done
    for b in 1 -1; do arms="$arms lds_${sh}_r4_dpp:$b::0 lds_${sh}_r4_dpp_nt:$b::0"; done
  done
  arms="$arms lds_1x8_r4_dpp_nt:0::0 lds_1x8_r4_dpp_nt:0::128 lds_1x8_r4_dpp_nt:0::64 lds_1x8_r4_dpp_nt:1::128 lds_2x4_r4_dpp_nt:1::128 lds_2x4_r4_dpp_nt:0::128 lds_4x2_r4_dpp_nt:1::128"
  echo "== $geo"; sw $geo 5 $arms | head -11
done
echo "################ 4. default plan (no autotune) and autotuned plan"
bash tools/gpu_call.sh r03 defaults 2>&1 | grep -v "^####"
