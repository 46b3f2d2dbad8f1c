#!/bin/bash
# Round-end evidence pass (round 4 on): everything the round's profiles/rNN_* files come from, each step bounded by its
# own timeout so that one hung tool cannot eat the GPU budget.
# Usage (GPU box, via gpurun): bash tools/evidence_pass.sh rNN [quick | notests]   (notests: quick, and the two pytest runs are skipped)
set +e
TAG=${1:-r05}
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
if [ "$QUICK" != "notests" ]; then
step "pytest -m gpu (product build)"
timeout 900 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -25 | tee $O/pytest_gpu.log | tail -3
step "pytest -m gpu (lab build as the library under test)"
MIBAYER_LIB_PATH=$R/gst-plugins-bad_amd/libmibayer_lab.so timeout 900 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -25 | tee $O/pytest_gpu_lab.log | tail -3
fi
step "bench, the driver's arguments"
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/bench.err | grep '^{' | tail -1 | tee $O/bench.json | cut -c1-400
PLAN=$(python -c "import json; print(json.load(open('$O/bench.json'))['config']['plan'])" 2>/dev/null)
step "rocprofv3 --kernel-trace --stats of the same command, pinned to the plan that line ran ($PLAN)"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o $TAG -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-host-path --no-traffic ${PLAN:+--plan $PLAN} 2>&1 | grep -v "^W20" | tail -1 | cut -c1-300)
# ... and of EVERY production plan the autotune can return at 4K x 64 (the driver's box picks its own, and
# whichever it picks, profiles/ must hold THAT kernel's average): the two that have won on some box so far, each pinned
step "rocprofv3 --kernel-trace --stats, one pass per production plan"
for plan in lds_4x2_r4_dpp_nt:1:0 lds_1x8_r4_dpp_nt:-1:0 lds_2x4_r4_dpp_nt:1:0; do
  name=$(echo $plan | tr ':-' '_m'); mkdir -p $O/stats_$name
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$name -o $TAG -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-host-path --no-traffic --plan $plan 2>&1 | grep '^{' | tail -1 > $O/stats_$name/bench_line.json)
  echo "$plan: $(cut -c1-160 $O/stats_$name/bench_line.json)"
done
step "one launch per frame: C caller, list launches, kernel traces"
gcc -O2 -Wall -I include tools/csrc/frame_launch_bench.c -o /tmp/frame_launch_bench -Lgst-plugins-bad_amd -lmibayer -Wl,-rpath,$R/gst-plugins-bad_amd
(for g in "3840 2160" "1920 1080" "7680 4320" "2592 1944" "4056 3040" "3838 2160"; do timeout 60 /tmp/frame_launch_bench $g; done; timeout 60 /tmp/frame_launch_bench 3840 2160 inverse) > $O/frame_launch_c.log 2>&1; grep -c "queue" $O/frame_launch_c.log
timeout 300 python tools/list_launch_bench.py > $O/list_launch.log 2>&1; timeout 300 python tools/list_launch_bench.py inverse > $O/list_launch_inverse.log 2>&1; head -8 $O/list_launch.log | cut -c1-150
for arm in 1:d:1 auto:d:1 auto:d:4; do
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$arm -o t -- python $R/tools/single_frame_bench.py trace $arm 2>&1 | grep -v "^W20" | tail -1)
  (echo "## arm $arm (variant:band:queues)"; python tools/single_frame_bench.py gaps $O/trace_$arm) >> $O/single_frame_gaps.txt 2>&1
done
find $O -name "t_kernel_trace.csv" -size +3M -delete
step "force-dist (RCCL group of one rank)"
timeout 300 python bench.py --gpus 1 --force-dist --steps 20 --warmup 5 --no-cpu --no-host-path --no-traffic 2>> $O/bench.err | grep '^{' | tail -1 > $O/bench_force_dist.json
step "stream mode (configs[4])"
timeout 300 python bench.py --mode stream 2>> $O/bench.err | grep '^{' | tail -1 > $O/stream_mode.json
step "eight ranks sharing the one GPU (functional: per_gpu, control-plane fields)"
timeout 400 python bench.py --gpus 8 --share-gpu --backend gloo --steps 5 --warmup 2 --no-cpu --no-host-path 2>> $O/bench.err | grep '^{' | tail -1 > $O/eight_ranks_one_gpu.json
step "eight ranks on the one GPU over RCCL (expected to be refused or to fall back: the error path of the N > 1 flow, executed once before a real node does)"
(timeout 300 python bench.py --gpus 8 --share-gpu --backend nccl --steps 5 --warmup 2 --no-cpu --no-host-path --no-traffic 2>&1 | grep -v "^W20\|^\*\*\*" | tail -25) > $O/eight_ranks_one_gpu_rccl.log 2>&1; grep -c . $O/eight_ranks_one_gpu_rccl.log
step "host CPU per frame"
timeout 300 bash tools/host_cpu_bench.sh 1500 > $O/host_cpu.log 2>&1; tail -16 $O/host_cpu.log | cut -c1-150
step "every BASELINE geometry, device-resident"
timeout 400 python tools/config_table.py > $O/configs.md 2>&1; tail -8 $O/configs.md | cut -c1-200
step "default vs measured vs cached plan"
timeout 300 python tools/common_geometries.py > $O/common_geometries.log 2>&1; tail -17 $O/common_geometries.log | cut -c1-200
step "leaks (contexts, pools, wedge registry; GStreamer leak tracer)"
(timeout 300 python tools/leak_check.py 2>&1 | tail -4; timeout 300 bash tools/gst_leaks.sh 2>&1 | tail -14) > $O/leaks.log; tail -4 $O/leaks.log | cut -c1-160
step "device-resident pipeline: where the time goes (HIP calls per frame, kernel busy time, CPU samples)"
timeout 600 bash tools/element_host_profile.sh $TAG > $O/element_host.log 2>&1; grep -c "^==" $O/element_host.log
step "host path: default mode vs graph mode, fresh processes"
timeout 400 python tools/host_path_bimodal.py --reps 4 --arms events:auto,graph:auto,events:spin > $O/host_path_bimodal.log 2>&1; tail -4 $O/host_path_bimodal.log
if [ -z "$QUICK" ]; then
  step "element-level fps"
  timeout 900 bash tools/gst_pipeline_bench.sh 2000 > $O/gst_pipeline_bench.log 2>&1; tail -20 $O/gst_pipeline_bench.log
  step "parity fuzz soak"
  MIBAYER_FUZZ_SEED=404 MIBAYER_FUZZ_CASES=1200 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k randomised 2>&1 | tail -3 | tee $O/fuzz_soak.log
  MIBAYER_LIB_PATH=$R/gst-plugins-bad_amd/libmibayer_lab.so MIBAYER_FUZZ_SEED=405 MIBAYER_FUZZ_CASES=1200 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k randomised 2>&1 | tail -3 | tee -a $O/fuzz_soak.log
fi
step "HBM-side traffic per block order (product build, --plan), calibration probe"
[ -x $R/tools/hbm_probe ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o $R/tools/hbm_probe $R/tools/hbm_probe.hip
for c in FETCH_SIZE WRITE_SIZE; do
  for plan in band1:1 chunk:-1 identity:0; do
    name=${plan%%:*}; band=${plan##*:}
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_bench_${name}_$c -o $TAG -- python $R/bench.py --steps 8 --warmup 2 --prewarm-ms 0 --no-cpu --no-host-path --no-traffic --plan lds_4x2_r4_dpp_nt:$band:0 2>&1 | grep -v "^W20" | tail -1 | cut -c1-120)
  done
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_probe_$c -o $TAG -- $R/tools/hbm_probe 2 32768 2>&1 | grep -v "^W20" | tail -1)
done
if [ -z "$QUICK" ]; then
  step "HBM-side traffic per geometry and plan (lab build)"
  MIBAYER_LIB_PATH=$R/gst-plugins-bad_amd/libmibayer_lab.so timeout 1500 bash tools/geometry_counters.sh $TAG 2>&1 | tail -3
fi
step done; find $O -type f | wc -l
