set +e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O
cd $R
timeout 500 bash tools/gst_pipeline_bench.sh 2000 2>&1 | tee $O/gst_pipeline_bench.log
timeout 600 python -m pytest tests/test_gst_hipmemory.py tests/test_gst_element.py tests/test_gst_harness.py -m gpu -q --timeout 200 2>&1 | tail -2
