#!/bin/bash
# rgb2bayer launch shapes at other geometries (does the 4K default hold?), incl. a width that needs the generic path
set +e
for g in 1920x1080x256 7680x4320x16 640x480x2048 3838x2160x64 1282x720x512; do
  echo "== $g"
  timeout 300 python tools/r2b_sweep.py $g 2:4:1:0:0 2:4:1:-1:0 2:4:0:0:0 4:4:1:0:0 1:4:1:0:0 2:8:1:0:0 0:2:0:-1:0 0:2:0:0:0 2>&1 | grep -v "^#"
done
