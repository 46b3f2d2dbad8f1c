#!/bin/bash
set +e
export TMPDIR=/tmp
echo "== pytest harness"; timeout 900 python -m pytest tests/test_gst_harness.py -m gpu -x -q 2>&1 | tail -30
