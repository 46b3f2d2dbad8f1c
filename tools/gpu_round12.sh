#!/bin/bash
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/pytest_gpu.log
echo "== bench"; timeout 600 python bench.py --no-cpu 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'], d['host_path'])"
echo "== stream graph"; timeout 600 python bench.py --mode stream 2>&1 | tail -1
echo "== stream nograph"; timeout 600 python bench.py --mode stream --no-graph 2>&1 | tail -1
