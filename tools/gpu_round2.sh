#!/bin/bash
# second GPU pass: full parity incl. GStreamer element, wider sweep, bench, rocprof stats + PMC traffic
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 | tee gpurun_out/pytest_gpu.log
echo "== sweep"; timeout 400 python tools/sweep.py 2>&1 | tee gpurun_out/sweep.log
echo "== sweep 1080p x 256"; timeout 200 env SWEEP_VARIANTS=0,1,8,12 python tools/sweep.py 1920 1080 256 2>&1 | tee gpurun_out/sweep_1080p.log
echo "== sweep 8K x 16"; timeout 200 env SWEEP_VARIANTS=0,1,8,12 python tools/sweep.py 7680 4320 16 2>&1 | tee gpurun_out/sweep_8k.log
echo "== bench"; timeout 600 python bench.py 2>&1 | tail -3 | tee gpurun_out/bench.log
cd /tmp
echo "== rocprof stats"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -o r01 -- python $R/bench.py --steps 40 --warmup 8 --no-cpu --no-host-path 2>&1 | tail -2
echo "== pmc bench"
for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | tr ' ' '_')
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_bench_$n -o r01 -- python $R/bench.py --steps 6 --warmup 2 --no-cpu --no-host-path 2>&1 | tail -1
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_probe_$n -o r01 -- $R/tools/hbm_probe 2 32768 2>&1 | tail -1
done
cd $R
find gpurun_out -name "*.csv" | head -50
