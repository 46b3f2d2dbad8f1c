#!/bin/bash
# Round-2 GPU pass 4: parity suite after the pool-helper rework, the upload probe, element-level numbers with enough
# frames to drown process start-up noise, the pool A/B again, an 8K plan sweep, a ROCTx marker trace.
set +e
TAG=${1:-r02d}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(rocm-smi --showserial 2>&1 | grep "Serial Number") | tee $O/box.txt
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest_gpu.log
echo "== upload probe"; timeout 600 python tools/upload_probe.py 2>&1 | tee $O/upload_probe.log
echo "== pool bench (helpers drive the ring)"; timeout 600 python tools/pool_pageable_bench.py 4 300 2>&1 | tee $O/pool_shared.log
echo "== pool bench, 1 shard"; timeout 600 python tools/pool_pageable_bench.py 1 300 2>&1 | tee $O/pool_1shard.log
echo "== gst pipeline bench (2000 frames per arm)"; timeout 1200 bash tools/gst_pipeline_bench.sh 2000 2>&1 | tee $O/gst_pipeline_bench.log
echo "== 8K sweep"; timeout 600 python tools/sweep2.py 7680 4320 16 6 lds_4x2_r4_dpp_nt:1 lds_4x2_r4_dpp_nt:-1 lds_4x2_r4_dpp_nt:0 lds_2x4_r4_dpp_nt:0 lds_2x4_r4_dpp_nt:1 lds_2x4_r4_dpp_nt:-1 lds_1x8_r4_dpp_nt:0 lds_1x8_r4_dpp_nt:-1 2>&1 | tee $O/sweep_8k.log
echo "== 1080p sweep"; timeout 600 python tools/sweep2.py 1920 1080 256 6 lds_4x2_r4_dpp_nt:1 lds_4x2_r4_dpp_nt:-1 lds_4x2_r4_dpp_nt:0 lds_2x4_r4_dpp_nt:0 lds_2x4_r4_dpp_nt:1 lds_2x4_r4_dpp_nt:-1 lds_1x8_r4_dpp_nt:0 lds_1x8_r4_dpp_nt:-1 2>&1 | tee $O/sweep_1080p.log
echo "== ROCTx marker trace of the host path"
cd /tmp
timeout 300 rocprofv3 --marker-trace --kernel-trace --memory-copy-trace --output-format csv -d $R/$O/marker -o mk -- python $R/tools/host_bands.py >/dev/null 2>&1
cd $R; ls $O/marker 2>/dev/null | head; f=$(ls $O/marker/*marker_api_trace.csv 2>/dev/null | head -1); [ -n "$f" ] && (head -1 $f; awk -F, 'NR>1{c[$0 ~ /mibayer:h2d/ ? "h2d" : $0 ~ /mibayer:kernel/ ? "kernel" : $0 ~ /mibayer:d2h/ ? "d2h" : $0 ~ /mibayer:wait/ ? "wait" : "other"]++} END{for(k in c) print k, c[k]}' $f; sed -n 2,9p $f)
