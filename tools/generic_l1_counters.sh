#!/bin/bash
# Counters BETWEEN the wave and the L2 for the generic path (output rows off the 64-byte sector grid) against its
# aligned neighbour geometry, SAME plan on both (shape 1024x8, nt stores, band 1), plus the generic geometry's own
# default plan: TCP->TCC requests, TCP / TA stall cycles, SQ instruction and wait counters, L2 request counts and
# EA write stalls.  Separate --pmc passes (kernel-trace + pmc only) over tools/run_geometry.py.
# Usage (GPU box): bash tools/generic_l1_counters.sh rNN ; then python tools/summarize_l1_counters.py rNN
set +e
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${TAG}_l1
mkdir -p $O
export TMPDIR=/tmp
(rocm-smi --showserial 2>&1 | grep "Serial Number:" | head -1) > $O/box.txt
cd /tmp
PASSES=(
  "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum"
  "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_TCR_TCP_STALL_CYCLES_sum"
  # (a TA_* pass -- TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
  #  TA_FLAT_WRITE_WAVEFRONTS_sum -- aborts rocprofv3 7.2 on gfx950 with signal 6 at finalisation, every time, and
  #  sits out its timeout: not collected)
  "SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM"
  "TCC_REQ_sum TCC_WRITE_sum"
  "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum"
)
run() {   # name W H N env...
  name=$1; W=$2; H=$3; N=$4; shift 4
  RUN_VARIANT=""
  for kv in "$@"; do case $kv in RUN_VARIANT=*) RUN_VARIANT=${kv#RUN_VARIANT=};; esac; done
  i=0
  for c in "${PASSES[@]}"; do
    d=$O/${name}_p$i; i=$((i+1))
    env "$@" timeout 90 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -o x -- \
      python $R/tools/run_geometry.py $W $H $N --reps 4 ${RUN_VARIANT:+--variant $RUN_VARIANT} 2>&1 | grep -E "GB/s|rror" | tail -1
  done
}
for pair in "3838 3840 2160 64" "4056 4064 3040 32" "2590 2592 1942 64"; do
  set -- $pair
  run g${1}x${3}x${4}_same $1 $3 $4 MIBAYER_XCD_BAND=1 RUN_VARIANT=lds_4x2_r4_dpp_nt
  run g${2}x${3}x${4}_same $2 $3 $4 MIBAYER_XCD_BAND=1 RUN_VARIANT=lds_4x2_r4_dpp_nt
  run g${1}x${3}x${4}_default $1 $3 $4 MIBAYER_UNUSED=1
done
cd $R; find $O -name "*counter_collection.csv" | wc -l
