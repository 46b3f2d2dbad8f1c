#!/bin/bash
# Development tool: element-level throughput of bayer2rgb inside a GStreamer pipeline (4K, fakesrc -> fakesink),
# for the synchronous default and the queued modes.  Run on the GPU box.
R=${GRAFT_REPO_ROOT:-$PWD}
export GST_PLUGIN_SYSTEM_PATH_1_0=/opt/conda/lib/gstreamer-1.0 GST_PLUGIN_PATH_1_0=$R/gst-plugins-bad_amd \
       GST_PLUGIN_SCANNER=/opt/conda/libexec/gstreamer-1.0/gst-plugin-scanner GST_REGISTRY=/tmp/gst_bench.reg
N=${1:-600}
/opt/conda/bin/gst-inspect-1.0 bayer2rgb >/dev/null 2>&1   # build the registry before timing anything
W=3840; H=2160
run () {
  local t0=$(date +%s.%N)
  /opt/conda/bin/gst-launch-1.0 -q fakesrc num-buffers=$1 sizetype=fixed sizemax=$((W*H)) filltype=nothing \
     ! video/x-bayer,format=rggb,width=$W,height=$H,framerate=0/1 ! $2 ! "${3:-video/x-raw,format=BGRx}" ! fakesink sync=false >/dev/null 2>&1
  local t1=$(date +%s.%N)
  echo "$t0 $t1" | awk '{print $2-$1}'
}
line () {   # label, elapsed(20 frames), elapsed(n+20 frames), n (default N)
  echo "$1 | $2 $3 ${4:-$N}" | awk -F'|' '{split($2,v," "); dt=v[2]-v[1]; printf "%-44s %7.1f fps  %8.1f Mpix/s  (%.3f s for %d frames)\n", $1, v[3]/dt, v[3]*3840*2160/dt/1e6, dt, v[3]}'
}
# a device-resident PRODUCER (hipbayersrc: frames in HBM): nothing crosses PCIe, the converter is what is timed.
# prefill=8: the source generated its 8 frames when the caps were set and hands them out with NO GPU work per buffer -- the
# figure is the converter's; without it every frame also costs a generator kernel as long as the converter's own.
run_dev () {   # nframes width height converter [source properties]
  local t0=$(date +%s.%N)
  /opt/conda/bin/gst-launch-1.0 -q hipbayersrc $5 num-buffers=$1 ! "video/x-bayer(memory:HIPMemory),format=rggb,width=$2,height=$3,framerate=0/1" \
     ! $4 ! fakesink sync=false >/dev/null 2>&1
  local t1=$(date +%s.%N)
  echo "$t0 $t1" | awk '{print $2-$1}'
}
line_dev () {  # label, elapsed(200 frames), elapsed(n+200 frames), n, width, height
  echo "$1 | $2 $3 $4 $5 $6" | awk -F'|' '{split($2,v," "); dt=v[2]-v[1]; printf "%-66s %9.1f fps  %10.1f Mpix/s  %5.1f %% of 8 TB/s  (%.3f s for %d frames)\n", $1, v[3]/dt, v[3]*v[4]*v[5]/dt/1e6, 5.0*v[3]*v[4]*v[5]/dt/1e9/80, dt, v[3]}'
}
# frame counts: >= 1.5 s per run, so that the +-50 ms of process start-up in the difference of two runs stay below 3 %
for geo in "3840 2160 200000 400000" "7680 4320 80000 100000" "1920 1080 500000 1500000" "3838 2160 200000 300000"; do
  set -- $geo
  for conv in "hipbayer2rgb" "hipbayer2rgb batch=4" "hipbayer2rgb batch=16"; do
    n=$3; [ "$conv" = "hipbayer2rgb batch=16" ] && n=$4
    a=$(run_dev 200 $1 $2 "$conv" prefill=8); b=$(run_dev $(($n+200)) $1 $2 "$conv" prefill=8)
    line_dev "hipbayersrc prefill=8 $1x$2 ! $conv" $a $b $n $1 $2
  done
done
# prefill=64: 64 distinct sources (531 MB at 4K), out of the reach of the 256 MB Infinity Cache that serves the
# re-reads of 8 prefilled frames -- the figure HBM alone carries
for geo in "3840 2160 200000 400000" "1920 1080 500000 1500000"; do
  set -- $geo
  for conv in "hipbayer2rgb" "hipbayer2rgb batch=16"; do
    n=$3; [ "$conv" = "hipbayer2rgb batch=16" ] && n=$4
    a=$(run_dev 200 $1 $2 "$conv" prefill=64); b=$(run_dev $(($n+200)) $1 $2 "$conv" prefill=64)
    line_dev "hipbayersrc prefill=64 $1x$2 ! $conv" $a $b $n $1 $2
  done
done
for geo in "3840 2160 40000" "1920 1080 40000"; do
  set -- $geo
  for conv in "hipbayer2rgb" "hipbayer2rgb batch=16" "hipbayer2rgb overlap=true"; do
    a=$(run_dev 200 $1 $2 "$conv"); b=$(run_dev $(($3+200)) $1 $2 "$conv")
    line_dev "hipbayersrc (a generator kernel per frame) $1x$2 ! $conv" $a $b $3 $1 $2
  done
done
a=$(run_dev 200 3840 2160 "identity" prefill=8); b=$(run_dev 4000200 3840 2160 "identity" prefill=8)
line_dev "hipbayersrc prefill=8 3840x2160 ! identity (no converter)" $a $b 4000000 3840 2160
a=$(run_dev 200 3840 2160 "identity"); b=$(run_dev 40200 3840 2160 "identity")
line_dev "hipbayersrc 3840x2160 ! identity (generator kernel alone)" $a $b 40000 3840 2160
a=$(run_dev 200 3840 2160 "hipbayer2rgb ! hiprgb2bayer" prefill=8); b=$(run_dev 120200 3840 2160 "hipbayer2rgb ! hiprgb2bayer" prefill=8)
line_dev "hipbayersrc prefill=8 3840x2160 ! hipbayer2rgb ! hiprgb2bayer" $a $b 120000 3840 2160
a=$(run_dev 200 3840 2160 "hipbayer2rgb batch=16 ! hiprgb2bayer batch=16" prefill=8); b=$(run_dev 160200 3840 2160 "hipbayer2rgb batch=16 ! hiprgb2bayer batch=16" prefill=8)
line_dev "hipbayersrc prefill=8 3840x2160 ! hipbayer2rgb batch=16 ! hiprgb2bayer batch=16" $a $b 160000 3840 2160
[ -n "$ONLY_DEV" ] && exit 0
# device-resident output (rank 4 of SURVEY 8(f)): only the 1 B/px mosaic crosses PCIe
DEV='video/x-raw(memory:HIPMemory),format=BGRx'
# (ten times the frames: this pipeline is fast enough for process start-up noise to matter)
a=$(run 20 "hipupload ! hipbayer2rgb" "$DEV"); b=$(run $((10*N+20)) "hipupload ! hipbayer2rgb" "$DEV")
line "hipupload ! hipbayer2rgb (stays on GPU)" $a $b $((10*N))
a=$(run 20 "hipupload ! hipbayer2rgb overlap=true" "$DEV"); b=$(run $((10*N+20)) "hipupload ! hipbayer2rgb overlap=true" "$DEV")
line "hipupload ! hipbayer2rgb overlap=true" $a $b $((10*N))
a=$(run 20 "hipupload ! hipbayer2rgb batch=8" "$DEV"); b=$(run $((10*N+20)) "hipupload ! hipbayer2rgb batch=8" "$DEV")
line "hipupload ! hipbayer2rgb batch=8" $a $b $((10*N))
a=$(run 20 "hipupload async=false ! hipbayer2rgb" "$DEV"); b=$(run $((10*N+20)) "hipupload async=false ! hipbayer2rgb" "$DEV")
line "hipupload async=false ! hipbayer2rgb" $a $b $((10*N))
a=$(run 20 "hipupload ! hipbayer2rgb ! hipdownload"); b=$(run $((N+20)) "hipupload ! hipbayer2rgb ! hipdownload")
line "hipupload ! hipbayer2rgb ! hipdownload" $a $b
a=$(run 20 "hipupload ! hipbayer2rgb ! hipdownload async=false"); b=$(run $((N+20)) "hipupload ! hipbayer2rgb ! hipdownload async=false")
line "hipupload ! hipbayer2rgb ! hipdownload async=false" $a $b
# the inverse element: 4 B/px in, 1 B/px out
run_inv () {
  local t0=$(date +%s.%N)
  /opt/conda/bin/gst-launch-1.0 -q fakesrc num-buffers=$1 sizetype=fixed sizemax=$((W*H*4)) filltype=nothing \
     ! video/x-raw,format=ARGB,width=$W,height=$H,framerate=0/1 ! $2 ! video/x-bayer,format=rggb ! fakesink sync=false >/dev/null 2>&1
  local t1=$(date +%s.%N)
  echo "$t0 $t1" | awk '{print $2-$1}'
}
for mode in "rgb2bayer" "rgb2bayer inflight=4"; do
  a=$(run_inv 20 "$mode"); b=$(run_inv $((N+20)) "$mode")
  line "$mode" $a $b
done
for mode in "bayer2rgb" "bayer2rgb inflight=2" "bayer2rgb inflight=4" "bayer2rgb inflight=3 devices=0,0" "bayer2rgb inflight=4 hipgraph=true" "bayer2rgb pinned-pool=false" "bayer2rgb inflight=2 devices=0,0,0,0" "bayer2rgb inflight=2 devices=0,0,0,0 pinned-pool=false"; do
  a=$(run 20 "$mode"); b=$(run $((N+20)) "$mode")
  echo "$mode | $a $b $N" | awk -F'|' '{split($2,v," "); dt=v[2]-v[1]; printf "%-44s %7.1f fps  %8.1f Mpix/s  (%.3f s for %d frames)\n", $1, v[3]/dt, v[3]*3840*2160/dt/1e6, dt, v[3]}'
done
