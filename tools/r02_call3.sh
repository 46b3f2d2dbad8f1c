#!/bin/bash
# Round-2 GPU pass 3 (third box): the full evidence pass (tools/gpu_profile.sh: parity suite, driver-args bench, rocprofv3
# stats + PMC traffic), rgb2bayer counters, the pool A/B of shared vs private queues, pipeline-level numbers.
set +e
TAG=${1:-r02}
bash tools/gpu_profile.sh $TAG
O=gpurun_out/$TAG
export TMPDIR=/tmp
SERIAL=$(grep -m1 "Serial Number:" $O/box.txt | awk '{print $NF}')
echo "== driver args, two more processes on box $SERIAL"
for i in 2 3; do
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-host-path 2>/dev/null | tail -1 > $O/driver_args_$i.json
  python - $O/driver_args_$i.json $SERIAL <<'PY'
import json, sys
b = json.load(open(sys.argv[1])); r = b["roofline"]
print("box %s driver-args run: frac %.4f kernel_ms %.4f per-step %s value %.0f plan %s band %s\n    autotune %s" % (
    sys.argv[2], r["frac"], r["kernel_ms"], r.get("kernel_ms_per_step"), b["value"],
    b["config"]["kernel_variant"], b["config"]["launch_plan"]["band"], b["config"]["autotune"]))
PY
done
echo "== rgb2bayer counters"; bash tools/r2b_counters.sh 2>&1 | tee $O/r2b_counters.log
echo "== pool bench, shared per-device queues (default)"; timeout 600 python tools/pool_pageable_bench.py 4 200 2>&1 | tee $O/pool_shared.log
echo "== pool bench, private queues per context"; MIBAYER_SHARED_QUEUES=0 timeout 600 python tools/pool_pageable_bench.py 4 200 2>&1 | tee $O/pool_private.log
echo "== stream mode"; timeout 600 python bench.py --mode stream 2>/dev/null | tail -1 | tee $O/stream.json | cut -c1-900
echo "== gst pipeline bench"; timeout 900 bash tools/gst_pipeline_bench.sh 2000 2>&1 | tee $O/gst_pipeline_bench.log
