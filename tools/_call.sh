set +e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05a; mkdir -p $O
cd $R
for g in 640x480 1280x720 1920x1080 2560x1440 2592x1944 3264x2448 3840x2160 4056x3040 4096x2160 4096x3072 5120x2880 7680x4320 1366x768 3838x2160; do
echo "== $g"; SFB_GEOMETRY=$g SFB_QUEUES=1,2 SFB_BANDS=d timeout 120 python tools/single_frame_bench.py 2>&1 | grep -v "round 0" | tee -a $O/single_frame_geometries.log
done
