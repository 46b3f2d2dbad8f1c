#!/bin/bash
# GPU pass 6: the parity suite after the preroll fix, pipeline arms for batch / async.
set +e
TAG=${1:-r02f}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
(rocm-smi --showserial 2>&1 | grep "Serial Number") | tee $O/box.txt
echo "== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -x -q --timeout 240 2>&1 | tail -6 | tee $O/pytest_gpu.log
R=${GRAFT_REPO_ROOT:-$PWD}
export GST_PLUGIN_SYSTEM_PATH_1_0=/opt/conda/lib/gstreamer-1.0 GST_PLUGIN_PATH_1_0=$R/gst-plugins-bad_amd \
       GST_PLUGIN_SCANNER=/opt/conda/libexec/gstreamer-1.0/gst-plugin-scanner GST_REGISTRY=/tmp/gst_bench.reg
/opt/conda/bin/gst-inspect-1.0 hipbayer2rgb >/dev/null 2>&1
W=3840; H=2160; DEV='video/x-raw(memory:HIPMemory),format=BGRx'
run () {
  local t0=$(date +%s.%N)
  timeout 300 /opt/conda/bin/gst-launch-1.0 -q fakesrc num-buffers=$1 sizetype=fixed sizemax=$((W*H)) filltype=nothing \
     ! video/x-bayer,format=rggb,width=$W,height=$H,framerate=0/1 ! $2 ! "$DEV" ! fakesink sync=false >/dev/null 2>&1
  local t1=$(date +%s.%N)
  echo "$t0 $t1" | awk '{print $2-$1}'
}
echo "== device-resident pipeline arms (20000 frames)"
for mode in "hipupload ! hipbayer2rgb" "hipupload ! hipbayer2rgb batch=8" "hipupload async=true ! hipbayer2rgb" "hipupload async=true ! hipbayer2rgb batch=8"; do
  a=$(run 20 "$mode"); b=$(run 20020 "$mode")
  echo "$mode | $a $b 20000" | awk -F'|' '{split($2,v," "); dt=v[2]-v[1]; printf "%-52s %7.1f fps  %8.1f Mpix/s  (%.3f s for %d frames)\n", $1, v[3]/dt, v[3]*3840*2160/dt/1e6, dt, v[3]}'
done | tee $O/device_resident_pipeline.log
