#!/bin/bash
set +e
export TMPDIR=/tmp
echo "== pytest hipmemory"; timeout 900 python -m pytest tests/test_gst_hipmemory.py -x -q 2>&1 | tail -30
