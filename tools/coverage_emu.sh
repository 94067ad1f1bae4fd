#!/bin/bash
# Line coverage of the product's sources -- kernels included -- by the emulated GPU suite: the host build under the wave emulator with gcov
# instrumentation (tests/_build/emu_cov/libmm2amd_emu_cov.so, dropin_emu_cov, dropin_pipeline_emu_cov).  Which device-code lines does no test reach?
#   tools/coverage_emu.sh build
#   MM2AMD_EMU=1 MM2AMD_EMU_LIB=tests/_build/emu_cov/libmm2amd_emu_cov.so MM2AMD_DROPIN_EMU=tests/_build/emu_cov/dropin_emu_cov python -m pytest -m gpu tests/...
#   tools/coverage_emu.sh report > /tmp/cov.txt      (per file: lines executed / executable, then the unexecuted lines of the .hip files)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/tests/_build/emu_cov; EMU=$ROOT/tests/cpucheck/wave_emu; CSRC=$ROOT/minimap2_amd/csrc
CPP="align backend_hip capi_common capi_index capi_kernels capi_map chain_host device_ctx flat_index format hits ksw_host ksw_ll mapper options rmq_chain tables"
HIP="seed_chain index_build device_sort ksw_extd2 ksw_gapfill ksw_stream ksw_splice ksw_ext region_finish region_dev ksw_order"
if [ "$1" = build ]; then
  mkdir -p $OUT; rm -f $OUT/*.gcda
  FLAGS="-std=c++17 -O2 -g --coverage -fPIC -ffp-contract=off -Wno-unknown-pragmas -I$EMU -I$ROOT/include"
  pids=()
  for f in $CPP; do g++ $FLAGS -c $CSRC/$f.cpp -o $OUT/$f.o & pids+=($!); done
  for f in $HIP; do g++ $FLAGS -x c++ -c $CSRC/$f.hip -o $OUT/$f.hip.o & pids+=($!); done
  g++ $FLAGS -c $EMU/wave_emu.cpp -o $OUT/wave_emu.o & pids+=($!)
  gcc -O1 -g -DHAVE_KALLOC -I/root/reference -I$ROOT/include -c $ROOT/tests/dropin/dropin_main.c -o $OUT/dropin_main.o & pids+=($!)
  gcc -O1 -g -DHAVE_KALLOC -I/root/reference -I$ROOT/include -c $ROOT/tests/dropin/dropin_pipeline.c -o $OUT/dropin_pipeline.o & pids+=($!)
  for p in "${pids[@]}"; do wait $p; done
  LIBOBJS=$(ls $OUT/*.o | grep -v "dropin_")
  g++ -shared --coverage -o $OUT/libmm2amd_emu_cov.so $LIBOBJS -L$ROOT/oracle -loracle -Wl,-rpath,$ROOT/oracle -lpthread
  g++ --coverage -o $OUT/dropin_emu_cov $OUT/dropin_main.o $ROOT/oracle/_ref/libminimap2_ref.a -L$OUT -lmm2amd_emu_cov -Wl,-rpath,$OUT -lm -lz -lpthread
  g++ --coverage -o $OUT/dropin_pipeline_emu_cov $OUT/dropin_pipeline.o $ROOT/oracle/_ref/libminimap2_ref.a -L$OUT -lmm2amd_emu_cov -Wl,-rpath,$OUT -lm -lz -lpthread
  echo built $OUT
elif [ "$1" = report ]; then
  cd $OUT
  for f in $HIP; do gcov -o $OUT $f.hip.gcda >/dev/null 2>&1 || true; done
  for f in $CPP; do gcov -o $OUT $f.gcda >/dev/null 2>&1 || true; done
  for f in $CPP; do
    g=$OUT/$f.cpp.gcov
    [ -f $g ] || continue
    awk -F: -v name=$f.cpp '{c=$1; gsub(/ /,"",c); if (c=="#####") miss++; else if (c!="-" && c!="=====") hit++} END {printf "%-22s %5d of %5d executable lines reached (%.1f %%)\n", name, hit, hit+miss, 100*hit/(hit+miss)}' $g
  done
  python3 - $OUT $HIP <<'PY'
import re, sys
out, files = sys.argv[1], sys.argv[2:]
rows = []
for f in files:
    try: lines = open("%s/%s.hip.gcov" % (out, f), errors="replace").read().split("\n")
    except OSError: continue
    hit, text = {}, {}
    for l in lines:  # "count: lineno: source"; template instantiations repeat their lines in blocks of their own: a line counts as reached when any occurrence ran
        m = re.match(r"^\s*([^:]+):\s*(\d+):(.*)$", l)
        if not m: continue
        c, n, src = m.group(1).strip().rstrip("*"), int(m.group(2)), m.group(3)
        if n == 0 or c == "-": continue
        text[n] = src
        hit[n] = hit.get(n, False) or (c not in ("#####", "=====") and c != "0")
    rows.append((f, sum(hit.values()), len(hit), [(n, text[n]) for n in sorted(hit) if not hit[n]]))
for f, h, t, _ in rows: print("%-22s %5d of %5d executable lines reached in some instantiation (%.1f %%)" % (f + ".hip", h, t, 100.0 * h / max(t, 1)))
for f, h, t, miss in rows:
    print("---- %s.hip: lines no test reached" % f)
    for n, src in miss: print("%5d:%s" % (n, src[:170]))
PY
fi
