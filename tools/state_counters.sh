#!/bin/bash
# The chunk-per-XCD block order runs at ~0.403 ms in some processes and ~0.433 ms in others on the same box
# (DESIGN.md section 5).  Which memory-system counters differ between the two states?  Every rocprofv3 pass is a process
# of its own, so it lands in one state or the other; each pass records its own kernel time next to its counters, and the
# summary groups the passes by that time.  REPS repetitions x 4 counter sets, chunk order, default start delay.
set +e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/state
REPS=${1:-6}
mkdir -p $OUT
cd /tmp
B="python $R/bench.py --steps 6 --warmup 2 --prewarm-ms 60 --no-cpu --no-host-path --no-autotune"
declare -A P
P[a]="TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum"
P[b]="TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum"
P[c]="TCC_TAG_STALL_sum TCC_SRC_FIFO_FULL_sum TCC_LATENCY_FIFO_FULL_sum TCC_IB_STALL_sum"
P[d]="TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"
for rep in $(seq 1 $REPS); do
  for k in a b c d; do
    d=$OUT/rep${rep}_$k
    MIBAYER_XCD_BAND=-1 timeout 300 rocprofv3 --kernel-trace --pmc ${P[$k]} --output-format csv -d $d -o p -- $B > $d.log 2>&1
  done
  # an unprofiled process in between: which state is the box handing out right now?
  MIBAYER_XCD_BAND=-1 timeout 300 $B 2>/dev/null | tail -1 | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('rep $rep unprofiled chunk kernel_ms', b['roofline']['kernel_ms'])"
done
cd $R; python - <<'PY'
import csv, glob, collections, os, re
rows_out = []
for d in sorted(glob.glob('gpurun_out/state/rep*_?')):
    f = glob.glob(d + '/*counter_collection.csv'); kt = glob.glob(d + '/*kernel_trace.csv')
    if not f or not kt:
        continue
    rows = [r for r in csv.DictReader(open(f[0])) if 'bayer2rgb' in r['Kernel_Name']]
    last = sorted({int(r['Dispatch_Id']) for r in rows})[-6:]
    agg = collections.defaultdict(list)
    for r in rows:
        if int(r['Dispatch_Id']) in last:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    kr = [r for r in csv.DictReader(open(kt[0])) if 'bayer2rgb' in r['Kernel_Name']][-6:]
    dur = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in kr) / len(kr) / 1e3
    rows_out.append((os.path.basename(d), dur, {k: sum(v) / len(v) for k, v in agg.items()}))
print('# every pass: kernel time of its own timed launches, then its counters (per launch, summed over 128 L2 channels)')
for name, dur, c in rows_out:
    print('%-10s %7.1f us  %s' % (name, dur, '  '.join('%s=%.0f' % (k.replace('_sum', ''), v) for k, v in sorted(c.items()))))
durs = sorted(d for _, d, _ in rows_out)
if durs:
    cut = (durs[0] + durs[-1]) / 2
    print('# fastest pass %.1f us, slowest %.1f us; split at %.1f us' % (durs[0], durs[-1], cut))
    for state, sel in (('FAST', lambda d: d < cut), ('SLOW', lambda d: d >= cut)):
        agg = collections.defaultdict(list)
        n = 0
        for name, dur, c in rows_out:
            if sel(dur):
                n += 1
                for k, v in c.items():
                    agg[k].append(v)
        print('== %s state: %d passes' % (state, n))
        for k in sorted(agg):
            print('   %-44s %16.0f  (n=%d)' % (k, sum(agg[k]) / len(agg[k]), len(agg[k])))
PY
