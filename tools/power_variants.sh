#!/bin/bash
# Development tool: package power / sclk of several kernel variants running the bench workload back to back (GPU box).
set +e
R=${GRAFT_REPO_ROOT:-$PWD}
rocm-smi --showserial 2>/dev/null | grep Serial
for v in ${VARIANTS:-1 3 13 8 4 14 10 16}; do
  python $R/bench.py --steps 14000 --warmup 20 --no-cpu --no-host-path --no-autotune --variant $v > /tmp/bench_load.json 2>/dev/null &
  P=$!
  sleep 5
  for i in 1 2 3; do
    echo -n "$v: "; rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power (W)" | sed 's/GPU\[0\]\t\t: //; s/Current Socket Graphics Package //' | tr '\n' ';'; echo
    sleep 0.5
  done
  wait $P
  python -c "import json; d=json.loads(open('/tmp/bench_load.json').read().strip().splitlines()[-1]); print('$v', d['config']['kernel_variant'], ': %.1f GB/s' % d['roofline']['achieved'], 'frac', d['roofline']['frac'], 'band', d['config']['launch_plan']['band'])"
done
