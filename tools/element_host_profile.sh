#!/bin/bash
# Where does the time of a device-resident pipeline go (VERDICT r05 #1b)?  For each arm of
#     hipbayersrc [prefill=8] ! hipbayer2rgb [batch=N] ! fakesink
# (1) frames per second, untraced; (2) rocprofv3 --hip-trace --kernel-trace --stats: HIP calls per frame with their
# mean duration, kernel busy time per frame; (3) a CPU sample profile of the process (tools/csrc/cpu_sampler.c: this
# image has no `perf`).  Writes gpurun_out/<TAG>_element_host/{report.txt,*.csv,samples_*.txt}; run on the GPU box.
set +e
export TMPDIR=/tmp
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/${TAG}_element_host
mkdir -p $O
# EHP_PLUGDIR: another build of the plugins (the before / after table of profiles/r06_element_host.md was taken with the
# round-5 binaries, commit 64c9ff4 built into a scratch directory); EHP_ARMS=generated: only the arms a build without `prefill` can run
export GST_PLUGIN_SYSTEM_PATH_1_0=/opt/conda/lib/gstreamer-1.0 GST_PLUGIN_PATH_1_0=${EHP_PLUGDIR:-$R/gst-plugins-bad_amd} \
       GST_PLUGIN_SCANNER=/opt/conda/libexec/gstreamer-1.0/gst-plugin-scanner GST_REGISTRY=/tmp/gst_bench_${TAG}.reg
/opt/conda/bin/gst-inspect-1.0 hipbayer2rgb >/dev/null 2>&1
gcc -O2 -fPIC -shared -o /tmp/libcpu_sampler.so $R/tools/csrc/cpu_sampler.c -ldl
REP=$O/report.txt
: > $REP
cd /tmp

pipeline () {   # src-props width height nframes converter
  echo "hipbayersrc $1 num-buffers=$4 ! video/x-bayer(memory:HIPMemory),format=rggb,width=$2,height=$3,framerate=0/1 ! $5 ! fakesink sync=false"
}
timed () {      # seconds of one run
  local t0=$(date +%s.%N)
  /opt/conda/bin/gst-launch-1.0 -q $1 >/dev/null 2>&1
  local t1=$(date +%s.%N)
  echo "$t0 $t1" | awk '{print $2-$1}'
}

arm () {        # label src-props width height nframes converter
  local label=$1 sp=$2 w=$3 h=$4 n=$5 conv=$6
  local name=$(echo "$label" | tr ' =' '__')
  echo "=================================================================" >> $REP
  echo "== $label: $(pipeline "$sp" $w $h $n "$conv")" >> $REP
  # untraced: the difference of two runs, the long one >= 1.5 s so that +-50 ms of process start-up stay below 3 %
  local m=$((5*n))
  local a=$(timed "$(pipeline "$sp" $w $h 200 "$conv")") b=$(timed "$(pipeline "$sp" $w $h $((m+200)) "$conv")")
  echo "$a $b $m $w $h" | awk '{dt=$2-$1; printf "untraced: %.1f fps = %.2f us per frame = %.1f %% of 8 TB/s (5 B/px)\n", $3/dt, dt/$3*1e6, 5.0*$3*$4*$5/dt/1e9/80}' >> $REP
  # (2) HIP API + kernel statistics
  rm -rf $O/trace_$name
  timeout 600 rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d $O/trace_$name -o t -- \
     /opt/conda/bin/gst-launch-1.0 -q $(pipeline "$sp" $w $h $n "$conv") >/dev/null 2>&1
  local api=$(ls $O/trace_$name/*hip_api_stats.csv 2>/dev/null | head -1) ker=$(ls $O/trace_$name/*kernel_stats.csv 2>/dev/null | head -1)
  if [ -n "$api" ]; then
    echo "-- HIP API calls (traced run, $n frames): name, calls, calls per frame, mean ns, total us per frame" >> $REP
    python3 - "$api" $n >> $REP <<'EOF'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2])
tot = 0.0
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    calls, total = int(r["Calls"]), float(r["TotalDurationNs"])
    tot += total
    if calls >= n / 64 or total / n > 50:
        print("   %-34s %9d  %7.3f /frame  %9.0f ns  %8.3f us/frame" % (r["Name"], calls, calls / n, total / calls, total / n / 1e3))
print("   all HIP calls together: %.3f us per frame (under the tracer)" % (tot / n / 1e3))
EOF
    cp $api $O/${name}_hip_api_stats.csv
  fi
  if [ -n "$ker" ]; then
    echo "-- kernels: name, calls, mean ns, busy us per frame" >> $REP
    python3 - "$ker" $n >> $REP <<'EOF'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2])
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    print("   %-70s %8d  %9.0f ns  %8.3f us/frame" % (r["Name"][:70], int(r["Calls"]), float(r["TotalDurationNs"]) / int(r["Calls"]), float(r["TotalDurationNs"]) / n / 1e3))
EOF
    cp $ker $O/${name}_kernel_stats.csv
  fi
  rm -rf $O/trace_$name
  # (3) CPU samples
  CPU_SAMPLER_MATCH=gst-launch CPU_SAMPLER_OUT=$O/samples_$name.txt LD_PRELOAD=/tmp/libcpu_sampler.so \
     /opt/conda/bin/gst-launch-1.0 -q $(pipeline "$sp" $w $h $((4*n)) "$conv") >/dev/null 2>&1
  echo "-- CPU samples (self, top 16; full tables in samples_$name.txt)" >> $REP
  sed -n '/^## self/,/^## inclusive/p' $O/samples_$name.txt | head -17 | tail -16 >> $REP
  sed -n '/^## threads/,/^## self/p' $O/samples_$name.txt | grep tid >> $REP
}

if [ "$EHP_ARMS" = "generated" ]; then
  arm "4K generated batch=1" ""          3840 2160 40000 "hipbayer2rgb"
  arm "4K generated batch=16" ""         3840 2160 40000 "hipbayer2rgb batch=16"
  arm "1080p generated batch=1" ""       1920 1080 60000 "hipbayer2rgb"
  arm "1080p generated batch=16" ""      1920 1080 60000 "hipbayer2rgb batch=16"
  cat $REP
  exit 0
fi
arm "4K prefill batch=1"   "prefill=8" 3840 2160 60000 "hipbayer2rgb"
arm "4K prefill batch=16"  "prefill=8" 3840 2160 60000 "hipbayer2rgb batch=16"
arm "4K generated batch=1" ""          3840 2160 40000 "hipbayer2rgb"
arm "1080p prefill batch=16" "prefill=8" 1920 1080 120000 "hipbayer2rgb batch=16"
arm "1080p prefill batch=1" "prefill=8" 1920 1080 120000 "hipbayer2rgb"
arm "4K prefill identity (source + GStreamer alone)" "prefill=8" 3840 2160 1000000 "identity"
cat $REP
