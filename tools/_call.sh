set +e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05b; mkdir -p $O
cd $R
/opt/rocm/bin/hipcc -x c -O2 -DWITH_CU_MASK -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include tools/csrc/frame_launch_bench.c -o /tmp/flb -Lgst-plugins-bad_amd -lmibayer -Wl,-rpath,$R/gst-plugins-bad_amd -L/opt/rocm/lib -lamdhip64
rm -f $O/frame_launch_cumask3.log
for v in 3 0; do
for s in mod:2 mod:3 mod:4 mod:6 mod:8 blk:4 blk:8 all:4 all:2; do echo "## variant $v scheme $s"; FLB_SCHEMES=$s FLB_VARIANT=$v timeout 120 /tmp/flb 3840 2160 | grep -v "^#"; done
done 2>&1 | tee -a $O/frame_launch_cumask3.log
for s in mod:2 mod:4 mod:8 all:4; do echo "## inverse scheme $s"; FLB_SCHEMES=$s timeout 120 /tmp/flb 3840 2160 inverse | grep -v "^#"; done 2>&1 | tee -a $O/frame_launch_cumask3.log
for g in "1920 1080" "7680 4320" "2592 1944" "4056 3040"; do for s in mod:2 mod:4 mod:8; do echo "## $g scheme $s"; FLB_SCHEMES=$s timeout 120 /tmp/flb $g | grep -v "^#"; done; done 2>&1 | tee -a $O/frame_launch_cumask3.log
