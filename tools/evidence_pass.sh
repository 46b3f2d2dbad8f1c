#!/bin/bash
# Round-end evidence pass (round 4 on): everything the round's profiles/rNN_* files come from, each step bounded by its
# own timeout so that one hung tool cannot eat the GPU budget.
# Usage (GPU box, via gpurun): bash tools/evidence_pass.sh rNN [quick]
set +e
TAG=${1:-r04}
QUICK=$2
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd $R
step() { echo "== $1 ($(date +%T))"; }
step box;    (rocm-smi --showmemorypartition --showcomputepartition --showclocks --showpower --showmaxpower --showserial 2>&1 | grep -v "^$" | head -60) > $O/box.txt
python -c "import sys; sys.path.insert(0, '$R'); import bench; print(bench.build_hash())" > $O/build_hash.txt
step smoke;  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.log
step "pytest -m gpu (product build)"
timeout 900 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -25 | tee $O/pytest_gpu.log | tail -3
step "pytest -m gpu (lab build as the library under test)"
MIBAYER_LIB_PATH=$R/gst-plugins-bad_amd/libmibayer_lab.so timeout 900 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -25 | tee $O/pytest_gpu_lab.log | tail -3
step "bench, the driver's arguments"
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/bench.err | grep '^{' | tail -1 | tee $O/bench.json | cut -c1-400
step "rocprofv3 --kernel-trace --stats of the same command"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o $TAG -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-host-path --no-traffic 2>&1 | grep -v "^W20" | tail -1 | cut -c1-300)
step "force-dist (RCCL group of one rank)"
timeout 300 python bench.py --gpus 1 --force-dist --steps 20 --warmup 5 --no-cpu --no-host-path --no-traffic 2>> $O/bench.err | grep '^{' | tail -1 > $O/bench_force_dist.json
step "stream mode (configs[4])"
timeout 300 python bench.py --mode stream 2>> $O/bench.err | grep '^{' | tail -1 > $O/stream_mode.json
step "eight ranks sharing the one GPU (functional: per_gpu, control-plane fields)"
timeout 400 python bench.py --gpus 8 --share-gpu --backend gloo --steps 5 --warmup 2 --no-cpu --no-host-path 2>> $O/bench.err | grep '^{' | tail -1 > $O/eight_ranks_one_gpu.json
step "host CPU per frame"
timeout 300 bash tools/host_cpu_bench.sh 1500 > $O/host_cpu.log 2>&1; tail -16 $O/host_cpu.log | cut -c1-150
step "default vs measured vs cached plan"
timeout 300 python tools/common_geometries.py > $O/common_geometries.log 2>&1; tail -17 $O/common_geometries.log | cut -c1-200
step "leaks (contexts, pools, wedge registry; GStreamer leak tracer)"
(timeout 300 python tools/leak_check.py 2>&1 | tail -4; timeout 300 bash tools/gst_leaks.sh 2>&1 | tail -14) > $O/leaks.log; tail -4 $O/leaks.log | cut -c1-160
if [ -z "$QUICK" ]; then
  step "element-level fps"
  timeout 400 bash tools/gst_pipeline_bench.sh 2000 > $O/gst_pipeline_bench.log 2>&1; tail -20 $O/gst_pipeline_bench.log
  step "parity fuzz soak"
  MIBAYER_FUZZ_SEED=404 MIBAYER_FUZZ_CASES=1200 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k randomised 2>&1 | tail -3 | tee $O/fuzz_soak.log
  MIBAYER_LIB_PATH=$R/gst-plugins-bad_amd/libmibayer_lab.so MIBAYER_FUZZ_SEED=405 MIBAYER_FUZZ_CASES=1200 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k randomised 2>&1 | tail -3 | tee -a $O/fuzz_soak.log
fi
step done; find $O -type f | wc -l
