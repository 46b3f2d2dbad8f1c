#!/bin/bash
# Host CPU of the element's GPU path: fps and CPU per frame of `fakesrc ! bayer2rgb <arm> ! fakesink`
# at 4K for the wait policies
#   auto       spin only while the frame waited for is alone in flight, otherwise naps from the start (the default)
#   spin2000   MIBAYER_WAIT_SPIN_US=2000: every wait spins its first 2 ms (what round 3 did)
#   nap        MIBAYER_WAIT_SPIN_US=0: never spin
# Arms marked PINNED take their input from videotestsrc (which uses the pinned pool the element proposes; the painting of
# the frames is then part of cpu_ms/frame, the element's own figures are not affected); the others from fakesrc, whose
# malloc'ed buffers make every upload a blocking staging copy -- that copy is the bulk of submit_cpu_us_per_frame there.
# Per arm: fps, CPU-ms per frame of the whole gst-launch process (user + sys, frames N+20 minus 20), and the element's
# own per-frame figures from its host-stats log line (submit / wait CPU of the streaming and helper threads, polls, naps).
# Usage (GPU box): bash tools/host_cpu_bench.sh [frames]
R=${GRAFT_REPO_ROOT:-$PWD}
export GST_PLUGIN_SYSTEM_PATH_1_0=/opt/conda/lib/gstreamer-1.0 GST_PLUGIN_PATH_1_0=$R/gst-plugins-bad_amd \
       GST_PLUGIN_SCANNER=/opt/conda/libexec/gstreamer-1.0/gst-plugin-scanner GST_REGISTRY=/tmp/gst_hostcpu.reg
N=${1:-600}
W=3840; H=2160
/opt/conda/bin/gst-inspect-1.0 bayer2rgb >/dev/null 2>&1
run () {   # frames, element-with-props -> "wall cpu" on stdout, the element's stats line in /tmp/hostcpu.err
  python3 - "$1" "$2" <<'PY'
import os, resource, subprocess, sys, time
n, el = sys.argv[1], sys.argv[2]
W, H = 3840, 2160
# fakesrc ignores the pinned pool the element proposes and hands over malloc memory (pageable input: the upload is the
# runtime's blocking staging copy); videotestsrc takes the pool (pinned input), at the price of painting each frame
if os.environ.get("HOSTCPU_PINNED_SRC"):
    src = ["videotestsrc", "num-buffers=" + n]
else:
    src = ["fakesrc", "num-buffers=" + n, "sizetype=fixed", "sizemax=%d" % (W * H), "filltype=nothing"]
cmd = ["/opt/conda/bin/gst-launch-1.0", "-q"] + src + \
      ["!", "video/x-bayer,format=rggb,width=%d,height=%d,framerate=%s" % (
          W, H, "30/1" if os.environ.get("HOSTCPU_PINNED_SRC") else "0/1"), "!"] + el.split() + \
      ["!", "video/x-raw,format=BGRx", "!", "fakesink", "sync=false"]
env = dict(os.environ, GST_DEBUG="bayer2rgb:4", GST_DEBUG_NO_COLOR="1")
t0 = time.perf_counter()
r0 = resource.getrusage(resource.RUSAGE_CHILDREN)
p = subprocess.run(cmd, env=env, capture_output=True, text=True)
r1 = resource.getrusage(resource.RUSAGE_CHILDREN)
t1 = time.perf_counter()
stats = [ln for ln in p.stderr.splitlines() if "host stats:" in ln]
open("/tmp/hostcpu.err", "w").write(stats[-1].split("host stats:")[1].strip() if stats else "no stats line (rc %d)" % p.returncode)
print("%.4f %.4f" % (t1 - t0, (r1.ru_utime + r1.ru_stime) - (r0.ru_utime + r0.ru_stime)))
PY
}
printf "%-46s %-9s %8s %14s   %s\n" "arm" "policy" "fps" "cpu_ms/frame" "element: per-frame host stats"
for arm in "bayer2rgb" "bayer2rgb inflight=2" "bayer2rgb inflight=4" "PINNED bayer2rgb inflight=2" "bayer2rgb inflight=2 devices=0,0,0,0" "THREADS bayer2rgb inflight=2 devices=0,0,0,0"; do
  for policy in auto spin2000 nap; do
    case $policy in auto) unset MIBAYER_WAIT_SPIN_US;; spin2000) export MIBAYER_WAIT_SPIN_US=2000;; nap) export MIBAYER_WAIT_SPIN_US=0;; esac
    el=$arm; unset MIBAYER_POOL_THREADS HOSTCPU_PINNED_SRC
    case $arm in THREADS*) export MIBAYER_POOL_THREADS=1; el=${arm#THREADS };; esac
    case $arm in PINNED*) export HOSTCPU_PINNED_SRC=1; el=${arm#PINNED };; esac   # videotestsrc: pinned input (its painting is in cpu_ms/frame)
    NN=$N; case $arm in PINNED*) NN=$((N/5));; esac       # (videotestsrc paints ~10 ms per 4K frame)
    a=($(run 20 "$el")); b=($(run $((NN+20)) "$el"))
    echo "${a[0]} ${a[1]} ${b[0]} ${b[1]} $NN" | awk -v arm="$arm" -v pol="$policy" -v st="$(cat /tmp/hostcpu.err)" \
      '{dt=$3-$1; cpu=$4-$2; printf "%-46s %-9s %8.1f %14.3f   %s\n", arm, pol, $5/dt, cpu*1e3/$5, st}'
  done
done
unset MIBAYER_WAIT_SPIN_US MIBAYER_POOL_THREADS HOSTCPU_PINNED_SRC
