#!/bin/bash
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== pytest variants"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "variant or golden" 2>&1 | tail -5 | tee gpurun_out/pytest_variants.log
echo "== sweep A"; timeout 600 python tools/sweep.py 2>&1 | tee gpurun_out/sweep3a.log
echo "== sweep B"; timeout 600 python tools/sweep.py 2>&1 | tee gpurun_out/sweep3b.log
cd /tmp
for v in 0 2 9 33; do
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc3_v${v}_$c -o r01 -- python $R/bench.py --variant $v --steps 6 --warmup 2 --no-cpu --no-host-path 2>&1 | tail -1 | cut -c1-200
done; done
