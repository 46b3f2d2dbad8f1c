set +e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05b; mkdir -p $O
cd $R
gcc -O2 -Wall -I include tools/csrc/frame_launch_bench.c -o /tmp/frame_launch_bench -Lgst-plugins-bad_amd -lmibayer -Wl,-rpath,$R/gst-plugins-bad_amd
for g in "3840 2160" "1920 1080" "7680 4320" "2592 1944"; do /tmp/frame_launch_bench $g; done 2>&1 | tee $O/frame_launch_c.log
/tmp/frame_launch_bench 3840 2160 inverse 2>&1 | tee -a $O/frame_launch_c.log
