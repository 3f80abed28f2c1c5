#!/bin/bash
# The two main macros at the reference's own sizes through the C++ host driver (GPU box); output under gpurun_out/demo.
ROOTDIR="${GRAFT_REPO_ROOT:-$(pwd)}"
CLI="$ROOTDIR/altair-raytracing_amd/host/isx_macro"
OUT="$ROOTDIR/gpurun_out/demo"; rm -rf "$OUT"; mkdir -p "$OUT"; cd "$OUT"
( time timeout -k 10 300 "$CLI" fluxAtObserverOptimize::sweepDetector folder=out srcZ=-75 dirY=0 thetaMax=170 ) > perpos.log 2>&1
( time timeout -k 10 300 "$CLI" fluxAtObserverFast::sweepDetectorTraceOnce folder=out srcZ=-75 dirY=0 thetaMax=170 ) > traceonce.log 2>&1
tail -n 6 perpos.log; tail -n 3 out/fluxmap_50000rays_180x90_src-60_0_-75.csv
tail -n 8 traceonce.log; tail -n 5 out/fluxmap_traceonce_100000rays_180x90_src-60_0_-75.csv
"$CLI" --analyze out/fluxmap_50000rays_180x90_src-60_0_-75.csv | tail -n 3
