cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for w in flux chord; do
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_q_$w -- python3 $R/tools/bench_configs.py --only $w --reps 2 > $R/gpurun_out/pmc_q_$w.log 2>&1
done
python3 - <<'PY'
import csv,glob,collections,os
R=os.environ['GRAFT_REPO_ROOT']
for w in ('flux','chord'):
    f=sorted(glob.glob(f'{R}/gpurun_out/pmc_q_{w}/*/*_counter_collection.csv'))[-1]
    by=collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if 'isx_trace_bin' in r['Kernel_Name']: by[r['Dispatch_Id']][r['Counter_Name']]=float(r['Counter_Value'])
    mx=max(v['SQ_INSTS_VALU'] for v in by.values())
    sel=[v for v in by.values() if v['SQ_INSTS_VALU']>0.5*mx]
    avg={k:sum(v[k] for v in sel)/len(sel) for k in sel[0]}
    print(w,{k:round(v/5e7,2) for k,v in avg.items()}, 'lane util',avg['SQ_THREAD_CYCLES_VALU']/64/avg['SQ_ACTIVE_INST_VALU'],'valu busy',4*avg['SQ_ACTIVE_INST_VALU']/(1024*avg['GRBM_GUI_ACTIVE']/8))
PY
