#!/bin/bash
# Tuning builds of libisx into variants/ (git-ignored; they travel to the GPU box with gpurun).  usage:
#   tools/build_variants.sh name:-DFLAG[,-DFLAG2] ...      e.g.  tools/build_variants.sh diag:-DISX_DIAG s4:-DISX_STEPS=4
# Built in parallel (8 at a time); time them with tools/ab.py variants/libisx_<name>.so ...
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "$ROOT/variants"
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-function -shared"
n=0
for v in "$@"; do
  name="${v%%:*}"; flags="${v#*:}"; [ "$flags" = "$v" ] && flags=""
  /opt/rocm/bin/hipcc $F ${flags//,/ } -o "$ROOT/variants/libisx_$name.so" "$ROOT/altair-raytracing_amd/csrc/isx_api.hip" || echo "FAILED $name" &
  n=$((n+1)); [ $((n % 8)) -eq 0 ] && wait
done
wait
ls -la "$ROOT/variants"
