#!/bin/bash
# Development tool: does mibayer_autotune() pick a plan at least as fast as the default?  Alternates the autotuned and
# the default bench run on one box (run on the GPU box).
for i in 1 2; do
python bench.py --no-cpu --no-host-path --steps 200 --warmup 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('autotuned', d['roofline']['frac'], d['config']['kernel_variant'], d['config']['launch_plan']['band'], d['config']['autotune'])"
python bench.py --no-cpu --no-host-path --no-autotune --steps 200 --warmup 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default  ', d['roofline']['frac'], d['config']['kernel_variant'], d['config']['launch_plan']['band'])"
done
rocm-smi --showserial | grep Serial
