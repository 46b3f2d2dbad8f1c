#!/bin/bash
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== sweep single arm sustained"; timeout 300 python tools/sweep2.py 3840 2160 64 30 lds_4x2_r4_dpp_nt 2>&1 | tee gpurun_out/s7_a.log
echo "== bench torch alloc"; timeout 300 python bench.py --no-cpu --no-host-path 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline'], d['ms_per_step'])"
echo "== bench hip alloc"; timeout 300 python bench.py --no-cpu --no-host-path --alloc hip 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline'], d['ms_per_step'])"
echo "== sweep 3 arms"; timeout 300 python tools/sweep2.py 3840 2160 64 12 lds_4x2_r4_dpp_nt lds_1x8_r4_dpp_nt lds_2x4_r4_dpp_nt lds_4x2_r4_dpp_nt:0 2>&1 | tee gpurun_out/s7_b.log
echo "== bench torch alloc again"; timeout 300 python bench.py --no-cpu --no-host-path 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline'], d['ms_per_step'])"
echo "== bench 1000 steps"; timeout 300 python bench.py --no-cpu --no-host-path --steps 1000 --warmup 50 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline'], d['ms_per_step'])"
