#!/bin/bash
# Development tool: clocks / power / temperatures of the box while the bench kernel runs back to back, for the default
# band-1 block order and for the chunk-per-XCD order (run on the GPU box).
set +e
R=${GRAFT_REPO_ROOT:-$PWD}
rocm-smi --showserial 2>/dev/null | grep Serial
echo "== idle"; rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|fclk\|power (W)" | tr '\n' ' '; echo
for band in 1 -1; do
  MIBAYER_XCD_BAND=$band python $R/bench.py --steps 20000 --warmup 20 --no-cpu --no-host-path --no-autotune > /tmp/bench_load.json 2>/dev/null &
  BP=$!
  sleep 5    # import + first steps
  for i in 1 2 3; do
    echo -n "band $band sample $i: "; rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -i "sclk\|fclk\|power (W)\|junction\|memory)" | sed 's/GPU\[0\]\t\t: //' | tr '\n' ';'; echo
    sleep 1
  done
  wait $BP
  python -c "import json; d=json.loads(open('/tmp/bench_load.json').read().strip().splitlines()[-1]); print('band $band: frac', d['roofline']['frac'], d['config']['kernel_variant'], 'ms', d['ms_per_step'])"
done
