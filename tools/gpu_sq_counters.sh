#!/bin/bash
# SQ / TCC / GRBM counter passes for the bench kernel (separate --pmc runs); prints per-launch means of the timed steps.
set +e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/sq
cd /tmp
B="python $R/bench.py --steps 6 --warmup 2 --no-cpu --no-host-path --no-autotune"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/sq/a -o sq -- $B >/dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/sq/b -o sq -- $B >/dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/sq/c -o sq -- $B >/dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $R/gpurun_out/sq/d -o sq -- $B >/dev/null 2>&1
cd $R; python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/sq/*')):
    f = glob.glob(d + '/*counter_collection.csv')
    if not f: print(d, 'no csv'); continue
    rows = [r for r in csv.DictReader(open(f[0])) if 'bayer2rgb' in r['Kernel_Name']]
    last = sorted({int(r['Dispatch_Id']) for r in rows})[-6:]
    agg = collections.defaultdict(list)
    for r in rows:
        if int(r['Dispatch_Id']) in last: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items(): print('%-28s %16.0f  (n=%d)' % (k, sum(v) / len(v), len(v)))
    if rows: print('   kernel', rows[-1]['Kernel_Name'][:70], 'LDS', rows[-1]['LDS_Block_Size'], 'grid', rows[-1]['Grid_Size'], 'wg', rows[-1]['Workgroup_Size'])
PY
