#!/bin/bash
# Round-2 evidence pass: the parity suite, then the DRIVER'S exact bench command several times in fresh
# processes on this box (plus A/B arms that show what the pre-warm and the per-step events do), and the
# rgb2bayer launch-shape sweep.   Usage (GPU box): bash tools/r02_driver_args.sh TAG
set +e
TAG=${1:-r02a}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
(rocm-smi --showserial --showclocks --showpower 2>&1 | grep -v "^$" | head -40) > $O/box.txt
SERIAL=$(grep -m1 "Serial Number:" $O/box.txt | awk '{print $NF}')
echo "== box $SERIAL"
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $O/pytest_gpu.log
echo "== driver args: python bench.py --gpus 1 --steps 20 --warmup 5"
for i in 1 2 3; do
  extra=""; [ $i -gt 1 ] && extra="--no-cpu --no-host-path"
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 $extra 2>/dev/null | tail -1 > $O/driver_args_$i.json
  python - $O/driver_args_$i.json $SERIAL <<'PY'
import json, sys
b = json.load(open(sys.argv[1])); r = b["roofline"]
print("box %s driver-args run: frac %.4f kernel_ms %.4f per-step %s value %.0f prewarm_ms %s plan %s band %s" % (
    sys.argv[2], r["frac"], r["kernel_ms"], r.get("kernel_ms_per_step"), b["value"], b.get("prewarm_ms"),
    b["config"]["kernel_variant"], b["config"]["launch_plan"]["band"]))
PY
done
echo "== A/B: no per-step events"; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-host-path --no-step-events 2>/dev/null | tail -1 | tee $O/ab_no_step_events.json | cut -c1-400
echo "== A/B: no pre-warm (round-1 behaviour minus the parity gap)"; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-host-path --prewarm-ms 0 2>/dev/null | tail -1 | tee $O/ab_no_prewarm.json | cut -c1-400
echo "== default args (200 steps)"; timeout 600 python bench.py --no-cpu --no-host-path 2>/dev/null | tail -1 | tee $O/default_args.json | cut -c1-400
echo "== rgb2bayer sweep"; timeout 600 python tools/r2b_sweep.py 2>&1 | tee $O/r2b_sweep.log | head -60
