#!/bin/bash
# Development tool: package power / sclk while plain streaming kernels (tools/hbm_probe) and the bench kernel run back to back.
# How much of the 1400 W budget does moving N TB/s cost?  (run on the GPU box)
set +e
R=${GRAFT_REPO_ROOT:-$PWD}
rocm-smi --showserial 2>/dev/null | grep Serial
sample () {   # label
  sleep $2
  for i in 1 2 3; do
    echo -n "$1: "; rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power (W)" | sed 's/GPU\[0\]\t\t: //; s/Current Socket Graphics Package //' | tr '\n' ';'; echo
    sleep 0.7
  done
}
echo -n "idle: "; rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power (W)" | sed 's/GPU\[0\]\t\t: //' | tr '\n' ';'; echo
for k in read fill_nt copy_nt mix1r4w_nt; do
  $R/tools/hbm_probe 12000 32768 $k > /tmp/probe_$k.log 2>&1 &
  P=$!
  sample $k 2.5
  wait $P; grep "grid" /tmp/probe_$k.log
done
python $R/bench.py --steps 20000 --warmup 20 --no-cpu --no-host-path --no-autotune > /tmp/bench_load.json 2>/dev/null &
P=$!
sample bayer2rgb 5
wait $P
python -c "import json; d=json.loads(open('/tmp/bench_load.json').read().strip().splitlines()[-1]); print('bayer2rgb: %.1f GB/s' % d['roofline']['achieved'], 'frac', d['roofline']['frac'])"
