#!/bin/bash
# tools/lane_utilisation_emu.sh [n_reads] : the streaming gap-fill kernel's lane utilisation -- cells / (128 x executed register-set rows) --
# counted WITHOUT a GPU: ksw_stream.hip built with -DMM2AMD_GF_COUNT=2 for the wave emulator (every wave prints its counts), map-ont reads of the
# benchmark's profile (10 kb, 12 % error) through the emulated library.  The count is a property of the kernel's schedule and of the job mix,
# not of the hardware.  Needs tests/_build/emu/*.o (make -C tests/cpucheck).
set -e
R=$(cd $(dirname $0)/.. && pwd); O=$R/tests/_build; V=$O/variants; mkdir -p $V
N=${1:-300}
g++ -std=c++17 -O2 -g -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-variable -Wno-unknown-pragmas -I$R/tests/cpucheck/wave_emu -I$R/include \
    -DMM2AMD_GF_COUNT=2 -x c++ -c $R/minimap2_amd/csrc/ksw_stream.hip -o $V/ksw_stream.count.o
OBJS=$(ls $O/emu/*.o | grep -v "/ksw_stream.hip.o")
g++ -shared -o $V/libmm2amd_emu_count.so $OBJS $V/ksw_stream.count.o -L$R/oracle -loracle -Wl,-rpath,$R/oracle -lpthread
MM2AMD_EMU_LIB=$V/libmm2amd_emu_count.so MM2AMD_HOST_PROF= python $R/tools/host_prof.py map-ont $N 5 2>/dev/null | python3 -c '
import sys, re
tot = {}
for line in sys.stdin:
    m = re.match(r"GFCOUNT stream<(\d)> slot (\d+): (\d+) register-set rows, (\d+) cells", line)
    if m:
        t = tot.setdefault(m.group(1), [0, 0, 0]); t[0] += int(m.group(3)); t[1] += int(m.group(4)); t[2] += 1
for k in sorted(tot):
    r, c, w = tot[k]
    print("ksw_stream_kernel<%s>: %d waves, %d register-set rows, %d cells, lane utilisation %.3f" % (k, w, r, c, c / (128.0 * r) if r else 0))
r = sum(t[0] for t in tot.values()); c = sum(t[1] for t in tot.values())
print("both classes: lane utilisation %.3f" % (c / (128.0 * r) if r else 0))
'
