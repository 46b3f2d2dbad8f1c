#!/bin/bash
# Where does the streaming thread spend its time with hipupload async=true?  HIP API statistics (rocprofv3 --hip-trace
# --stats, no counters) of the device-resident pipeline in both uploader modes, 3000 4K frames each.
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/async_trace
mkdir -p $O
export GST_PLUGIN_SYSTEM_PATH_1_0=/opt/conda/lib/gstreamer-1.0 GST_PLUGIN_PATH_1_0=$R/gst-plugins-bad_amd \
       GST_PLUGIN_SCANNER=/opt/conda/libexec/gstreamer-1.0/gst-plugin-scanner GST_REGISTRY=/tmp/gst_bench.reg
/opt/conda/bin/gst-inspect-1.0 hipbayer2rgb >/dev/null 2>&1
W=3840; H=2160
cd /tmp
for mode in sync:false async:true; do
  name=${mode%%:*}; flag=${mode##*:}
  timeout 300 rocprofv3 --hip-trace --stats --output-format csv -d $O/$name -o t -- /opt/conda/bin/gst-launch-1.0 -q fakesrc num-buffers=3000 sizetype=fixed sizemax=$((W*H)) filltype=nothing \
     ! video/x-bayer,format=rggb,width=$W,height=$H,framerate=0/1 ! hipupload async=$flag ! hipbayer2rgb ! "video/x-raw(memory:HIPMemory),format=BGRx" ! fakesink sync=false >/dev/null 2>&1
  echo "== hipupload async=$flag: HIP API statistics (3000 frames)"
  f=$(ls $O/$name/*hip_api_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && (head -1 $f; sort -t, -k3 -n -r $f | head -14)
done
