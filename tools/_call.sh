set +e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05c; mkdir -p $O
cd $R
echo "== common geometries"; timeout 600 python tools/common_geometries.py 2>&1 | tee $O/common_geometries.log
echo "== plan sweep (defaults column now incl. the known-width table)"; timeout 900 python tools/plan_sweep.py $O/plan_sweep_530_after.json 530 common 2>&1 | cut -c1-75 | tee $O/plan_sweep_530Mpix_after.log
echo "== pytest parity"; timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -x 2>&1 | tail -8
