#!/bin/bash
# Development tool: L2 <-> fabric (TCC_EA) counters of the bench kernel with and without the start delay, for the band-1
# and the chunk-per-XCD block orders -- what does the ~1.5k-cycle delay change in the memory system?
# Separate --pmc passes (never combined with other trace domains).  Run on the GPU box; prints per-launch means.
set +e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/delay
mkdir -p $OUT
cd /tmp
B="python $R/bench.py --steps 6 --warmup 2 --no-cpu --no-host-path --no-autotune"
declare -A P
P[a]="TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum"
P[b]="TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum"
P[c]="TCC_TAG_STALL_sum TCC_SRC_FIFO_FULL_sum TCC_LATENCY_FIFO_FULL_sum TCC_IB_STALL_sum"
P[d]="TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"
P[e]="TCC_BUSY_sum TCC_CYCLE_sum GRBM_GUI_ACTIVE"
for cfg in "1 0" "1 24" "-1 0" "-1 24"; do
  set -- $cfg
  for k in a b c d e; do
    d=$OUT/band$1_sleep$2/$k
    mkdir -p $OUT/band$1_sleep$2
    MIBAYER_XCD_BAND=$1 MIBAYER_START_SLEEP=$2 timeout 300 rocprofv3 --kernel-trace --pmc ${P[$k]} \
        --output-format csv -d $d -o p -- $B > $d.log 2>&1
  done
done
cd $R; python - <<'PY'
import csv, glob, collections, os
for cfg in sorted(glob.glob('gpurun_out/delay/band*')):
    print('==', os.path.basename(cfg))
    for d in sorted(glob.glob(cfg + '/?')):
        f = glob.glob(d + '/*counter_collection.csv')
        if not f:
            print('  ', d[-1], 'no csv'); continue
        rows = [r for r in csv.DictReader(open(f[0])) if 'bayer2rgb' in r['Kernel_Name']]
        last = sorted({int(r['Dispatch_Id']) for r in rows})[-6:]
        agg = collections.defaultdict(list)
        for r in rows:
            if int(r['Dispatch_Id']) in last:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
        kt = glob.glob(d + '/*kernel_trace.csv')
        dur = ''
        if kt:
            kr = [r for r in csv.DictReader(open(kt[0])) if 'bayer2rgb' in r['Kernel_Name']][-6:]
            if kr:
                dur = 'kernel %.1f us' % (sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in kr) / len(kr) / 1e3)
        for k, v in agg.items():
            print('   %-40s %18.0f  %s' % (k, sum(v) / len(v), dur))
PY
