#!/bin/bash
# Development tool: GStreamer's own leak tracer over the elements (GstBuffer / GstMemory / GstObject refcounts).
# Any "object-alive" line at exit is a leaked object.  Run on the GPU box.
R=${GRAFT_REPO_ROOT:-$PWD}
export GST_PLUGIN_SYSTEM_PATH_1_0=/opt/conda/lib/gstreamer-1.0 GST_PLUGIN_PATH_1_0=$R/gst-plugins-bad_amd \
       GST_PLUGIN_SCANNER=/opt/conda/libexec/gstreamer-1.0/gst-plugin-scanner GST_REGISTRY=/tmp/gst_leaks.reg
/opt/conda/bin/gst-inspect-1.0 bayer2rgb >/dev/null 2>&1
B='video/x-bayer,format=rggb,width=640,height=480,framerate=30/1'
for p in \
  "videotestsrc num-buffers=40 ! $B ! bayer2rgb ! fakesink" \
  "videotestsrc num-buffers=40 ! $B ! bayer2rgb inflight=3 ! fakesink" \
  "videotestsrc num-buffers=40 ! $B ! bayer2rgb inflight=2 devices=0,0 hipgraph=true ! fakesink" \
  "videotestsrc num-buffers=40 ! video/x-raw,format=ARGB,width=640,height=480 ! rgb2bayer inflight=2 ! bayer2rgb ! fakesink" \
  "videotestsrc num-buffers=40 ! $B ! hipupload ! hipbayer2rgb ! hipdownload ! fakesink" \
  "videotestsrc num-buffers=40 ! $B ! hipupload ! hipbayer2rgb ! fakesink" \
  "videotestsrc num-buffers=40 ! $B ! hipupload async=true ! hipbayer2rgb batch=4 ! hipdownload ! fakesink" \
  "videotestsrc num-buffers=40 ! $B ! hipupload async=false ! hipbayer2rgb batch=16 ! fakesink" \
  "videotestsrc num-buffers=40 ! $B ! hipupload ! hipbayer2rgb ! hipdownload async=false ! fakesink" \
  "hipbayersrc num-buffers=40 ! hipbayer2rgb ! hipdownload ! fakesink" \
  "hipbayersrc prefill=4 num-buffers=40 ! hipbayer2rgb batch=4 ! fakesink" \
  "hipbayersrc prefill=3 num-buffers=40 ! tee name=t t. ! queue ! hipbayer2rgb ! hipdownload ! fakesink t. ! queue ! hipbayer2rgb batch=4 ! hiprgb2bayer ! fakesink" \
  "FAULT videotestsrc num-buffers=40 ! $B ! bayer2rgb inflight=2 devices=0,0,0 ! fakesink" \
  "videotestsrc num-buffers=40 ! $B ! bayer2rgb inflight=2 devices=0,0,0 pinned-pool=false ! fakesink"; do
  fault=""
  case "$p" in FAULT*) p=${p#FAULT }; fault="1:3";; esac
  export MIBAYER_INJECT_FAULT=$fault
  out=$(GST_TRACERS=leaks GST_DEBUG=GST_TRACER:7 GST_DEBUG_NO_COLOR=1 /opt/conda/bin/gst-launch-1.0 -q $p 2>&1)
  rc=$?
  alive=$(echo "$out" | grep -c "object-alive, type-name=(string)[A-Za-z]")
  echo "rc=$rc alive=$alive ${fault:+(shard 1 fails after 3 frames) }:: $p"
  echo "$out" | grep "object-alive, type-name=(string)[A-Za-z]" | head -5
done
