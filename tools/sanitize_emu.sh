#!/bin/bash
# The product's sources -- kernels included -- under AddressSanitizer: the host build under the wave emulator (tests/cpucheck/wave_emu), where
# "device memory" is heap memory, so an out-of-bounds access of a kernel is a heap-buffer-overflow report with a source line.
#   tools/sanitize_emu.sh [extra -D flags, e.g. -DMM2AMD_BT_LDS_CAP=64]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/tests/_build/emu_asan; EMU=$ROOT/tests/cpucheck/wave_emu; CSRC=$ROOT/minimap2_amd/csrc
mkdir -p $OUT
FLAGS="-std=c++17 -O1 -g -fPIC -ffp-contract=off -fsanitize=address -fno-omit-frame-pointer -Wno-unknown-pragmas -I$EMU -I$ROOT/include $*"
pids=()
for f in align backend_hip capi_common capi_index capi_kernels capi_map chain_host device_ctx flat_index format hits ksw_host ksw_ll mapper options rmq_chain tables; do
  g++ $FLAGS -c $CSRC/$f.cpp -o $OUT/$f.o & pids+=($!)
done
for f in seed_chain index_build device_sort ksw_extd2 ksw_gapfill ksw_stream ksw_band ksw_splice ksw_ext ksw_extq region_finish region_dev ksw_order; do
  g++ $FLAGS -x c++ -c $CSRC/$f.hip -o $OUT/$f.hip.o & pids+=($!)
done
g++ $FLAGS -c $EMU/wave_emu.cpp -o $OUT/wave_emu.o & pids+=($!)
gcc -O1 -g -fsanitize=address -DHAVE_KALLOC -I/root/reference -I$ROOT/include -c $ROOT/tests/dropin/dropin_main.c -o $OUT/dropin_main.o & pids+=($!)
for p in "${pids[@]}"; do wait $p; done
g++ -fsanitize=address -o $OUT/dropin_emu_asan $OUT/*.o $ROOT/oracle/_ref/libminimap2_ref.a -L$ROOT/oracle -loracle -Wl,-rpath,$ROOT/oracle -lm -lz -lpthread
# the same objects as a shared library, for the Python-driven GPU cases:
#   LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 MM2AMD_EMU=1 MM2AMD_EMU_LIB=tests/_build/emu_asan/libmm2amd_emu_asan.so python -m pytest -m gpu tests/test_gpu_ksw.py -k extension
g++ -shared -fsanitize=address -o $OUT/libmm2amd_emu_asan.so $(ls $OUT/*.o | grep -v dropin_main) -L$ROOT/oracle -loracle -Wl,-rpath,$ROOT/oracle -lpthread
echo built $OUT/dropin_emu_asan $OUT/libmm2amd_emu_asan.so
