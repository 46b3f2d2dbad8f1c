#!/bin/bash
# Round-2 GPU pass 5: parity suite (list launches, batch / async elements, element failover), list-launch bench,
# placement probe fresh vs fragmented, the two-rank flow on one GPU, pipeline arms for the device-resident elements.
set +e
TAG=${1:-r02e}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
(rocm-smi --showserial 2>&1 | grep "Serial Number") | tee $O/box.txt
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest_gpu.log
echo "== list launch bench"; timeout 600 python tools/list_launch_bench.py 2>&1 | tee $O/list_launch.log
echo "== placement probe, fresh process"; timeout 600 python tools/alloc_placement_probe.py 3 2>&1 | tee $O/placement_fresh.log | head -12
echo "== placement probe, fragmented first"; timeout 600 python tools/alloc_placement_probe.py 3 frag 2>&1 | tee $O/placement_frag.log | head -12
echo "== placement probe, fresh again"; timeout 600 python tools/alloc_placement_probe.py 3 2>&1 | tee $O/placement_fresh2.log | head -12
echo "== two ranks on one GPU (gloo control plane)"; timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --backend gloo --share-gpu 2>&1 | tail -1 | tee $O/two_ranks.json | cut -c1-700
echo "== driver args"; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | tee $O/driver_args.json | cut -c1-300
