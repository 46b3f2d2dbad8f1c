#!/bin/bash
# SQ / TCC / traffic counter passes for the rgb2bayer kernel (separate --pmc runs, kernel-trace only), for the default
# flat kernel and the round-1 tile kernel beside it.  Prints per-launch means; the raw CSVs stay under gpurun_out/r2bc.
set +e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2bc
mkdir -p $O
cd /tmp
for arm in flat:2 tile:0; do
  name=${arm%%:*}; flat=${arm##*:}
  i=0
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" \
             "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM" \
             "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
             "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
             "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    MIBAYER_R2B_FLAT=$flat timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/${name}_$i -o r2b -- python $R/tools/r2b_run.py 8 >/dev/null 2>&1
  done
done
cd $R; python - <<'PY'
import csv, glob, collections
for arm in ("flat", "tile"):
    print("== rgb2bayer %s kernel, per-launch means over the last 8 dispatches" % arm)
    for d in sorted(glob.glob('gpurun_out/r2bc/%s_*' % arm), key=lambda p: int(p.rsplit('_', 1)[1])):
        f = glob.glob(d + '/*counter_collection.csv')
        if not f:
            print(d, 'no csv'); continue
        rows = [r for r in csv.DictReader(open(f[0])) if 'rgb2bayer' in r['Kernel_Name']]
        last = sorted({int(r['Dispatch_Id']) for r in rows})[-8:]
        agg = collections.defaultdict(list)
        for r in rows:
            if int(r['Dispatch_Id']) in last:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in agg.items():
            print('%-30s %16.0f  (n=%d)' % (k, sum(v) / len(v), len(v)))
        if rows:
            r = rows[-1]
            print('   kernel', r['Kernel_Name'][:80], 'grid', r['Grid_Size'], 'wg', r['Workgroup_Size'], 'vgpr', r.get('VGPR_Count'), 'sgpr', r.get('SGPR_Count'))
PY
