#!/bin/bash
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== pytest variants"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "variant or golden or batch" 2>&1 | tail -5 | tee gpurun_out/pytest_variants.log
NAMES=$(python -c "
import __graft_entry__ as g; n=g.load_package().variant_names()
print(','.join(str(n.index(x)) for x in ['lds_1x8_r4_dpp_nt','lds_4x2_r4_dpp_nt','lds_2x4_r4_dpp_nt','lds_1x4_r8_dpp_nt','lds_1x16_r4_dpp_nt','lds_1x4_r4_dpp_nt']))")
for band in 0 1 2 4 8 16 32 -1; do
  echo "== band $band"; MIBAYER_XCD_BAND=$band SWEEP_VARIANTS=$NAMES timeout 300 python tools/sweep.py 2>&1 | tee -a gpurun_out/sweep4_band.log
done
echo "== 1080p"; for band in 0 1 4; do MIBAYER_XCD_BAND=$band SWEEP_VARIANTS=$NAMES timeout 300 python tools/sweep.py 1920 1080 256 2>&1 | tee -a gpurun_out/sweep4_1080p.log; done
echo "== 8K"; for band in 0 1 4; do MIBAYER_XCD_BAND=$band SWEEP_VARIANTS=$NAMES timeout 300 python tools/sweep.py 7680 4320 16 2>&1 | tee -a gpurun_out/sweep4_8k.log; done
cd /tmp
V=$(python -c "
import sys; sys.path.insert(0,'$R')
import __graft_entry__ as g; n=g.load_package().variant_names(); print(n.index('lds_1x8_r4_dpp_nt'))")
for band in 1 4; do
for c in FETCH_SIZE WRITE_SIZE; do
  MIBAYER_XCD_BAND=$band timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc4_band${band}_$c -o r01 -- python $R/bench.py --variant $V --steps 6 --warmup 2 --no-cpu --no-host-path 2>&1 | tail -1 | cut -c1-100
done; done
