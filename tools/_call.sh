set +e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05d; mkdir -p $O
cd $R
gcc -std=c99 -I include tests/check/abi_c_consumer.c -o /tmp/consumer -Lgst-plugins-bad_amd -lmibayer -Wl,-rpath,$R/gst-plugins-bad_amd
hang=0; for i in $(seq 1 40); do timeout 12 /tmp/consumer >/dev/null 2>&1; rc=$?; [ $rc -eq 124 ] && hang=$((hang+1)); [ $rc -ne 0 ] && [ $rc -ne 124 ] && echo "rc $rc"; done
echo "final build: abi_c_consumer, frame queues kept for the process: $hang hangs of 40" | tee -a $O/fq_mode.log
for i in 1 2 3 4 5 6; do timeout 600 python -m pytest tests/test_gst_hipmemory.py tests/test_abi_c.py -m gpu -q --timeout 200 2>&1 | tail -1; done | tee -a $O/fq_mode.log
echo "== pytest product"; timeout 900 python -m pytest tests -m gpu -q --timeout 300 2>&1 | tail -4 | tee $O/pytest_gpu.log
echo "== pytest lab"; MIBAYER_LIB_PATH=$R/gst-plugins-bad_amd/libmibayer_lab.so timeout 900 python -m pytest tests -m gpu -q --timeout 300 2>&1 | tail -4 | tee $O/pytest_gpu_lab.log
