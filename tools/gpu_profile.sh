#!/bin/bash
# Round-end evidence pass: parity suite, bench line, rocprofv3 kernel stats of the same bench command, and
# HBM traffic counters (separate --pmc passes) for the bench kernel plus the calibration probe.
# Usage (on the GPU box, via gpurun): bash tools/gpu_profile.sh rNN
set +e
TAG=${1:-r01}
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
[ -x $R/tools/hbm_probe ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o $R/tools/hbm_probe $R/tools/hbm_probe.hip
python -c "import sys; sys.path.insert(0, '$R'); import bench; print(bench.build_hash())" > $O/build_hash.txt
echo "== box"; (rocm-smi --showmemorypartition --showcomputepartition --showclocks --showpower --showmaxpower --showserial 2>&1 | grep -v "^$" | head -60) | tee $O/box.txt
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.log
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/pytest_gpu.log
echo "== bench (the driver's arguments)"; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | tee $O/bench.json | cut -c1-600
cd /tmp
echo "== rocprof stats (same command as the bench line, fewer steps)"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o $TAG -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-host-path --no-traffic 2>&1 | grep -v "^W20" | tail -1 | cut -c1-300
# HBM-side traffic, separate --pmc passes, for the THREE block orders the autotuner chooses between
# (forced with MIBAYER_XCD_BAND, autotune off), plus the calibration probe
for c in FETCH_SIZE WRITE_SIZE; do
  for plan in band1:1 chunk:-1 identity:0; do
    name=${plan%%:*}; band=${plan##*:}
    MIBAYER_LIB_PATH=$R/gst-plugins-bad_amd/libmibayer_lab.so MIBAYER_XCD_BAND=$band timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_bench_${name}_$c -o $TAG -- python $R/bench.py --steps 8 --warmup 2 --prewarm-ms 0 --no-cpu --no-host-path --no-autotune --no-traffic 2>&1 | grep -v "^W20" | tail -1 | cut -c1-120
  done
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_probe_$c -o $TAG -- $R/tools/hbm_probe 2 32768 2>&1 | grep -v "^W20" | tail -1
done
cd $R; find gpurun_out/$TAG -name "*.csv" | wc -l
