#!/bin/bash
# Run on the GPU box (via gpurun).  Kernel-trace stats and PMC passes are separate runs
# (never --pmc together with sys/hip/hsa traces).  Outputs under gpurun_out/prof_*.
set -o pipefail
ROOTDIR="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOTDIR/gpurun_out"
RAYS="${RAYS:-50000000}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOTDIR/bench.py --steps 3 --warmup 1 --cpu-rays 0 --rays $RAYS"
echo "== kernel trace + stats"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_kt" -- $BENCH > "$OUT/prof_kt.log" 2>&1 || { echo "kt failed"; tail -5 "$OUT/prof_kt.log"; exit 1; }
for C in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $C"
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d "$OUT/prof_pmc_$C" -- $BENCH > "$OUT/prof_pmc_$C.log" 2>&1 || { echo "pmc $C failed"; tail -5 "$OUT/prof_pmc_$C.log"; exit 1; }
done
echo "== pmc SQ set A"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d "$OUT/prof_pmc_sqA" -- $BENCH > "$OUT/prof_pmc_sqA.log" 2>&1 || { echo "sqA failed"; tail -5 "$OUT/prof_pmc_sqA.log"; exit 1; }
echo "== pmc SQ set B"
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d "$OUT/prof_pmc_sqB" -- $BENCH > "$OUT/prof_pmc_sqB.log" 2>&1 || { echo "sqB failed"; tail -5 "$OUT/prof_pmc_sqB.log"; }
find "$OUT" -name "*.csv" | head -40
