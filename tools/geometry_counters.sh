#!/bin/bash
# HBM-side traffic per geometry and plan: separate --pmc FETCH_SIZE / WRITE_SIZE passes (kernel-trace + pmc only)
# over tools/run_geometry.py, for the store-alignment arms of the generic path and for the aligned geometries the
# autotuner serves.  Usage (GPU box): bash tools/geometry_counters.sh rNN ; then tools/summarize_geometry_counters.py rNN
set +e
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${TAG}_geom
mkdir -p $O
export TMPDIR=/tmp
(rocm-smi --showserial 2>&1 | grep "Serial Number:" | head -1) > $O/box.txt
python -c "import sys; sys.path.insert(0, '$R'); import bench; print(bench.build_hash())" > $O/build_hash.txt
cd /tmp
run() {   # name W H N env...
  name=$1; W=$2; H=$3; N=$4; shift 4
  RUN_VARIANT=""
  for kv in "$@"; do case $kv in RUN_VARIANT=*) RUN_VARIANT=${kv#RUN_VARIANT=};; esac; done
  for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
    d=$O/${name}_$(echo $c | tr ' ' '+')
    env "$@" timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -o x -- python $R/tools/run_geometry.py $W $H $N --reps 8 ${RUN_VARIANT:+--variant $RUN_VARIANT} 2>&1 | grep -E "GB/s|rror" | tail -1
  done
}
for geo in "4056 3040 32" "3838 2160 64" "1366 768 512"; do
  set -- $geo
  run g${1}x${2}x${3}_default $1 $2 $3 MIBAYER_UNUSED=1
  run g${1}x${2}x${3}_wb-chunk $1 $2 $3 MIBAYER_XCD_BAND=-1 RUN_VARIANT=lds_4x2_r4_dpp
  run g${1}x${2}x${3}_nt-band1 $1 $2 $3 MIBAYER_XCD_BAND=1 RUN_VARIANT=lds_4x2_r4_dpp_nt
  run g${1}x${2}x${3}_hybrid-band1 $1 $2 $3 MIBAYER_XCD_BAND=1 RUN_VARIANT=lds_4x2_r4_dpp_hy
  run g${1}x${2}x${3}_shift128-1x8 $1 $2 $3 MIBAYER_XCD_BAND=0 MIBAYER_ALIGN_STORES=128 RUN_VARIANT=lds_1x8_r4_dpp_nt
done
# aligned geometries: the plans autotune picks (chunk at 8K, band 1 at 1080p / 4K)
run g7680x4320x64_chunk 7680 4320 64 MIBAYER_XCD_BAND=-1
run g7680x4320x64_band1 7680 4320 64 MIBAYER_XCD_BAND=1
run g1920x1080x256_band1 1920 1080 256 MIBAYER_XCD_BAND=1
run g3840x2160x64_band1 3840 2160 64 MIBAYER_XCD_BAND=1
run g3840x2160x64_chunk 3840 2160 64 MIBAYER_XCD_BAND=-1
run g3840x2160x64_identity 3840 2160 64 MIBAYER_XCD_BAND=0
cd $R; find $O -name "*counter_collection.csv" | wc -l
