#!/bin/bash
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== pytest gst"; timeout 1500 python -m pytest tests/test_gst_element.py -m gpu -x -q 2>&1 | tail -30 | tee gpurun_out/pytest_gst.log
