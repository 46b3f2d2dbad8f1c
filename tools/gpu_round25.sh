#!/bin/bash
set +e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/sq
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/sq/a -o sq -- python $R/bench.py --steps 6 --warmup 2 --no-cpu --no-host-path --no-autotune 2>&1 | grep -v "^W20" | tail -1 | cut -c1-100
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD --output-format csv -d $R/gpurun_out/sq/b -o sq -- python $R/bench.py --steps 6 --warmup 2 --no-cpu --no-host-path --no-autotune 2>&1 | grep -v "^W20" | tail -1 | cut -c1-100
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $R/gpurun_out/sq/c -o sq -- python $R/bench.py --steps 6 --warmup 2 --no-cpu --no-host-path --no-autotune 2>&1 | grep -v "^W20" | tail -1 | cut -c1-100
timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $R/gpurun_out/sq/d -o sq -- python $R/bench.py --steps 6 --warmup 2 --no-cpu --no-host-path --no-autotune 2>&1 | grep -v "^W20" | tail -1 | cut -c1-100
cd $R; python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/sq/*')):
    f = glob.glob(d + '/*counter_collection.csv')
    if not f: print(d, 'no csv'); continue
    rows = [r for r in csv.DictReader(open(f[0])) if 'bayer2rgb' in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    agg = collections.defaultdict(list)
    last = sorted({int(r['Dispatch_Id']) for r in rows})[-6:]
    for r in rows:
        if int(r['Dispatch_Id']) in last: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items(): print('%-28s %16.0f  (n=%d)' % (k, sum(v) / len(v), len(v)))
    if rows: print('   kernel', rows[-1]['Kernel_Name'][:70], 'VGPR', rows[-1]['VGPR_Count'], 'SGPR', rows[-1]['SGPR_Count'], 'LDS', rows[-1]['LDS_Block_Size'], 'grid', rows[-1]['Grid_Size'], 'wg', rows[-1]['Workgroup_Size'])
PY
