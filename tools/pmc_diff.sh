#!/bin/bash
# Differential PMC attribution (GPU box): the same 2e7-ray launch as (a) full kernel, (b) trace only (bin_mode 2),
# (c) trace + binning prep without the column loop (variants/libisx_preponly.so, built ad hoc).  Separate --pmc passes.
set -o pipefail
ROOTDIR="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOTDIR/gpurun_out/pmcdiff"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
SETS=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
      "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU")
run() { # name, BM, libpath
  local name=$1 bm=$2 lib=$3 k=0
  for S in "${SETS[@]}"; do
    BM=$bm ISX_LIB_PATH=$lib timeout -k 10 200 rocprofv3 --pmc $S --output-format csv -d "$OUT/${name}_$k" -- python3 "$ROOTDIR/tools/run_mode.py" > "$OUT/${name}_$k.log" 2>&1 || { echo "$name set $k failed"; tail -3 "$OUT/${name}_$k.log"; }
    k=$((k+1))
  done
}
run full 1 "$ROOTDIR/altair-raytracing_amd/csrc/libisx.so"
run trace 2 "$ROOTDIR/altair-raytracing_amd/csrc/libisx.so"
[ -f "$ROOTDIR/variants/libisx_preponly.so" ] && run prep 1 "$ROOTDIR/variants/libisx_preponly.so"
python3 - "$OUT" <<'PY'
import collections, csv, glob, os, sys
out = sys.argv[1]
res = collections.defaultdict(dict)
for f in glob.glob(os.path.join(out, "*", "*", "*_counter_collection.csv")):
    name = f[len(out) + 1:].split("/")[0].rsplit("_", 1)[0]
    by = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if "isx_trace_bin_kernel" in r["Kernel_Name"]:
            by[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    if by:
        last = by[sorted(by, key=int)[-1]]
        res[name].update(last)
keys = sorted({k for v in res.values() for k in v})
print("counter," + ",".join(res))
for k in keys:
    print(k + "," + ",".join(f"{res[n].get(k, float('nan')):.4g}" for n in res))
PY
