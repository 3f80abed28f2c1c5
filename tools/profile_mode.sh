#!/bin/bash
# PMC counters of one kernel variant: BM = bin_mode (2 = trace only), TM = trace_mode.
ROOTDIR="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$ROOTDIR/gpurun_out"; mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
export BM="${BM:-2}" TM="${TM:-0}"
rm -rf "$OUT/pm_a" "$OUT/pm_b"
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d "$OUT/pm_a" -- python3 $ROOTDIR/tools/run_mode.py > "$OUT/pm_a.log" 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pm_b" -- python3 $ROOTDIR/tools/run_mode.py > "$OUT/pm_b.log" 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("pm_a","pm_b"):
    f=sorted(glob.glob("$OUT/"+d+"/*/*_counter_collection.csv"))
    if not f: print(d,"no csv"); continue
    by=collections.defaultdict(dict)
    for r in csv.DictReader(open(f[-1])):
        if "isx_trace" in r["Kernel_Name"]: by[r["Dispatch_Id"]][r["Counter_Name"]]=float(r["Counter_Value"])
    last=list(by.values())[-1]
    print(d, {k: f"{v:.4g}" for k,v in last.items()})
PY
tail -1 "$OUT/pm_a.log"
