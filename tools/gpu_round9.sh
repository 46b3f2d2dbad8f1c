#!/bin/bash
set +e
export TMPDIR=/tmp
A="lds_4x2_r4_dpp_nt:-1:0 lds_4x2_r4_dpp_nt:-1:1 lds_4x2_r4_dpp_nt:-1:2 lds_4x2_r4_dpp_nt:-1:3 lds_4x2_r4_dpp_nt:-1:4 lds_4x2_r4_dpp_nt:-1:5 lds_4x2_r4_dpp_nt:-1:6 lds_4x2_r4_dpp_nt:-1:7 lds_4x2_r4_dpp_nt:0"
for i in 1 2 3; do
  timeout 300 python tools/sweep2.py 3840 2160 64 8 $A 2>&1 | grep -v "^3840"
done
echo "== after a dummy allocation history"
python - <<'PY'
import os, subprocess, sys
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
pkg = g.load_package()
c = pkg.Context(64, 64)
junk = [c.device_alloc(s) for s in (300 << 20, 1 << 30, 700 << 20)]
c.device_free(junk[1])
os.environ["KEEP"] = "1"
subprocess.run([sys.executable, "tools/sweep2.py", "3840", "2160", "64", "8"] + "lds_4x2_r4_dpp_nt:-1:0 lds_4x2_r4_dpp_nt:-1:3 lds_4x2_r4_dpp_nt:0".split())
PY
