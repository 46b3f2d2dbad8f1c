#!/bin/bash
# Round-2 GPU pass 2: parity suite (failover, pageable helpers, NUMA, configs[4] ring, flat rgb2bayer), the driver's
# bench command on this (second) box, the stream mode, sweeps and probes.   Usage: bash tools/r02_call2.sh TAG
set +e
TAG=${1:-r02b}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
(rocm-smi --showserial --showclocks --showpower 2>&1 | grep -v "^$" | head -40) > $O/box.txt
SERIAL=$(grep -m1 "Serial Number:" $O/box.txt | awk '{print $NF}')
echo "== box $SERIAL"; (lscpu | grep -E "Model name|Socket|NUMA node\(s\)|^CPU\(s\)"; numactl -H 2>/dev/null | head -4) | tee $O/host.txt
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 | tee $O/pytest_gpu.log
echo "== driver args"
for i in 1 2 3; do
  extra=""; [ $i -gt 1 ] && extra="--no-cpu --no-host-path"
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 $extra 2>/dev/null | tail -1 > $O/driver_args_$i.json
  python - $O/driver_args_$i.json $SERIAL <<'PY'
import json, sys
b = json.load(open(sys.argv[1])); r = b["roofline"]
print("box %s driver-args run: frac %.4f kernel_ms %.4f per-step %s value %.0f plan %s band %s\n    autotune %s" % (
    sys.argv[2], r["frac"], r["kernel_ms"], r.get("kernel_ms_per_step"), b["value"],
    b["config"]["kernel_variant"], b["config"]["launch_plan"]["band"], b["config"]["autotune"]))
PY
done
python -c "
import json; b=json.load(open('$O/driver_args_1.json')); print('host_path', b.get('host_path')); c=b.get('cpu_baseline',{}); print('cpu', c.get('value'), c.get('all_cores'))"
echo "== stream mode (configs[4])"; timeout 600 python bench.py --mode stream 2>/dev/null | tail -1 | tee $O/stream.json | cut -c1-1200
echo "== alloc placement probe"; timeout 600 python tools/alloc_placement_probe.py 5 2>&1 | tee $O/alloc_placement.log
echo "== pool pageable bench, shared per-device queues (default)"; timeout 600 python tools/pool_pageable_bench.py 4 200 2>&1 | tee $O/pool_pageable.log
echo "== pool pageable bench, private queues per context"; MIBAYER_SHARED_QUEUES=0 timeout 600 python tools/pool_pageable_bench.py 4 200 2>&1 | tee $O/pool_pageable_private.log
echo "== rgb2bayer sweep"; timeout 600 python tools/r2b_sweep.py 2>&1 | tee $O/r2b_sweep.log | head -16
