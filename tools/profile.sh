#!/bin/bash
# Run on the GPU box (via gpurun).  Kernel-trace stats and PMC passes are separate runs
# (never --pmc together with sys/hip/hsa traces).  Outputs under gpurun_out/prof_<NAME>_*.
#   NAME   label of the profile (default "flux")              WORKLOAD  bench | brdf | discs | perpos
#   RAYS   rays of the bench workload (default 5e7)
# The program after `--` is python3 itself (no env/bash hop: the profiler's library has initialised the GPU by then).
set -o pipefail
ROOTDIR="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOTDIR/gpurun_out"
RAYS="${RAYS:-50000000}"
NAME="${NAME:-flux}"
WORKLOAD="${WORKLOAD:-bench}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
if [ "$WORKLOAD" = "bench" ]; then
  BENCH="python3 $ROOTDIR/bench.py --steps 3 --warmup 1 --cpu-rays 0 --rays $RAYS"
else
  BENCH="python3 $ROOTDIR/tools/bench_configs.py --only $WORKLOAD --reps 3"
fi
P="$OUT/prof_${NAME}"
rm -rf "${P}_kt" "${P}_pmc_"*
echo "== kernel trace + stats ($BENCH)"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "${P}_kt" -- $BENCH > "${P}_kt.log" 2>&1 || { echo "kt failed"; tail -5 "${P}_kt.log"; exit 1; }
pmc() { # set name, counters...
  local set=$1; shift
  echo "== pmc $set"
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d "${P}_pmc_$set" -- $BENCH > "${P}_pmc_$set.log" 2>&1 || { echo "pmc $set failed"; tail -5 "${P}_pmc_$set.log"; return 1; }
}
pmc FETCH_SIZE FETCH_SIZE || exit 1
pmc WRITE_SIZE WRITE_SIZE || exit 1
pmc sqA SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY || exit 1
pmc sqB SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
# executed instruction mix ("also report executed FP64 ops from rocprof")
pmc f64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64
pmc f32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32
pmc int SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_LDS_ATOMIC
find "$OUT" -path "*prof_${NAME}_*" -name "*.csv" | head -40
