#!/bin/bash
# hipupload async=true/false behind a producer that (a) uses the proposed pinned pool and (b) needs CPU time per frame:
# videotestsrc painting a 4K mosaic.  The asynchronous uploader lets the producer paint frame n+1 while frame n crosses PCIe.
set +e
R=${GRAFT_REPO_ROOT:-$PWD}
export GST_PLUGIN_SYSTEM_PATH_1_0=/opt/conda/lib/gstreamer-1.0 GST_PLUGIN_PATH_1_0=$R/gst-plugins-bad_amd \
       GST_PLUGIN_SCANNER=/opt/conda/libexec/gstreamer-1.0/gst-plugin-scanner GST_REGISTRY=/tmp/gst_bench.reg
/opt/conda/bin/gst-inspect-1.0 hipbayer2rgb >/dev/null 2>&1
W=3840; H=2160; DEV='video/x-raw(memory:HIPMemory),format=BGRx'
run () {
  local t0=$(date +%s.%N)
  timeout 600 /opt/conda/bin/gst-launch-1.0 -q videotestsrc pattern=$3 num-buffers=$1 ! video/x-bayer,format=rggb,width=$W,height=$H,framerate=120/1 \
     ! $2 ! "$DEV" ! fakesink sync=false >/dev/null 2>&1
  local t1=$(date +%s.%N)
  echo "$t0 $t1" | awk '{print $2-$1}'
}
for pat in black smpte; do
  for mode in "hipupload async=false ! hipbayer2rgb" "hipupload async=true ! hipbayer2rgb" "hipupload async=true ! hipbayer2rgb batch=4"; do
    a=$(run 20 "$mode" $pat); b=$(run 620 "$mode" $pat)
    echo "videotestsrc pattern=$pat ! $mode | $a $b 600" | awk -F'|' '{split($2,v," "); dt=v[2]-v[1]; printf "%-78s %7.1f fps  %8.1f Mpix/s  (%.3f s for %d frames)\n", $1, v[3]/dt, v[3]*3840*2160/dt/1e6, dt, v[3]}'
  done
done
# the producer alone, for scale
t0=$(date +%s.%N); /opt/conda/bin/gst-launch-1.0 -q videotestsrc pattern=black num-buffers=600 ! video/x-bayer,format=rggb,width=$W,height=$H,framerate=120/1 ! fakesink sync=false >/dev/null 2>&1; t1=$(date +%s.%N)
echo "$t0 $t1" | awk '{printf "videotestsrc pattern=black alone (malloc buffers)                                      %7.1f fps\n", 600/($2-$1)}'
