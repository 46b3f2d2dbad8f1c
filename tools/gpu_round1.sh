#!/bin/bash
# first GPU pass: parity, variant sweep, HBM ceilings, bench, rocprof stats
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "== pytest gpu"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 | tee gpurun_out/pytest_gpu.log
echo "== sweep"; timeout 300 python tools/sweep.py 2>&1 | tee gpurun_out/sweep.log
echo "== probe"; timeout 200 ./tools/hbm_probe 2>&1 | tee gpurun_out/probe.log
echo "== bench"; timeout 600 python bench.py 2>&1 | tail -3 | tee gpurun_out/bench.log
echo "== rocprof"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 8 --no-cpu --no-host-path 2>&1 | tail -5)
ls -R gpurun_out/prof | head -30
