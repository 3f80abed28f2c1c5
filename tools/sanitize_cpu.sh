#!/bin/bash
# AddressSanitizer + UBSan over the CPU-side code (GPU ASan is not available on the pool): the oracle through its ctypes binding,
# the host driver's CLI paths that need no GPU, and the collective's transport-double test.  Run from the repo root; needs gcc/g++.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="${TMPDIR:-/tmp}/isx_san"; mkdir -p "$OUT"
SAN="-O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer"
gcc $SAN -std=c11 -fPIC -ffp-contract=off -mfma -msse4.1 -fopenmp -D_GNU_SOURCE -shared -o "$OUT/libisx_oracle.so" "$ROOT/oracle/isx_oracle.c" -lm
HOSTSRC="$ROOT/altair-raytracing_amd/host"; CSRC="$ROOT/altair-raytracing_amd/csrc"
LINK="-L$CSRC -lisx -L/opt/rocm/lib -lrccl -lamdhip64 -Wl,-rpath,$CSRC -Wl,-rpath,/opt/rocm/lib -lpthread"
g++ $SAN -std=c++17 -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -o "$OUT/isx_macro_san" "$HOSTSRC/isx_macro_main.cpp" "$HOSTSRC/isx_macros.cpp" "$HOSTSRC/isx_comm.cpp" $LINK
g++ $SAN -std=c++17 -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -o "$OUT/comm_stub_san" "$ROOT/tests/native/comm_stub_test.cpp" "$HOSTSRC/isx_macros.cpp" "$HOSTSRC/isx_comm.cpp" $LINK
export ASAN_OPTIONS=detect_leaks=0
cd "$OUT"
./isx_macro_san --selftest-writer "$OUT/w.csv"; ./isx_macro_san --unique "$OUT/w.csv" > /dev/null; ./isx_macro_san --analyze "$OUT/w.csv" > /dev/null
ISX_RANK=1 ISX_WORLD=4 ./isx_macro_san --shard 10 > /dev/null
./comm_stub_san
LD_PRELOAD="$(gcc -print-file-name=libasan.so)" python3 - "$ROOT" "$OUT" <<'PY'
import sys, os
root, out = sys.argv[1], sys.argv[2]
sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, oracle as O
O.ORACLE_SO = os.path.join(out, "libisx_oracle.so")
c = O.default_config(); O.fluxmap(c, 20000, 1, 0)
c.source_model = 1; O.fluxmap(c, 5000, 1, 0)
c = O.default_config(); c.surface_model = 1; O.fluxmap(c, 3000, 2, 0)
c = O.default_config(); c.lambertian = 0; c.roughness_rad = 0.3; c.reflectance = 0.9; O.fluxmap(c, 3000, 2, 0)
c = O.default_config(); c.trace_mode = 1; O.fluxmap(c, 3000, 2, 0)
O.fluxmap_per_position(O.default_config(), 3, 5, 2)
ca = np.array([[0, 0, -200, 0, 0, 1.0], [10, 0, -199, 0, 0, 1.0]]); cc = O.default_config(); cc.r_out = 105; cc.reflectance = 1.0; cc.max_points = 10000
O.disc_sweep(cc, ca, 5.0, 0.1, 5000, 3); O.disc_sweep_per_position(cc, ca, 5.0, 0.1, 3000, 3); O.exit_dz_hist(cc, 5000, 3, 100)
print("sanitized oracle run complete")
PY
echo "sanitize_cpu: clean"
