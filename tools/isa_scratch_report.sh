#!/bin/bash
# tools/isa_scratch_report.sh file.hip : where a HIP file's kernels touch scratch memory, relative to their hot loops -- from the gfx950 assembly
# (hipcc -S), no GPU needed.  For every kernel: VGPRs / scratch bytes per lane, the line ranges that hold packed 16-bit arithmetic (v_pk_*: the DP
# cell, i.e. the row loop) and the scratch loads / stores inside and outside those ranges.  A spill that sits outside the row loop costs a few
# instructions per JOB, not per row.
set -e
R=$(cd $(dirname $0)/.. && pwd)
F=${1:-ksw_stream.hip}
S=/tmp/isa_$$.s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I$R/include -x hip --cuda-device-only -S $R/minimap2_amd/csrc/$F -o $S 2>/dev/null
c++filt < $S > $S.d 2>/dev/null || cp $S $S.d
python3 - $S <<'PY'
import re, sys
lines = open(sys.argv[1]).read().split("\n")
kern, start = None, 0
out = []
def report(name, a, b):
    body = lines[a:b]
    pk = [i for i, l in enumerate(body) if "\tv_pk_" in l]
    sc = [i for i, l in enumerate(body) if "\tscratch_" in l]
    if not pk and not sc:
        return
    # ranges of packed arithmetic: runs with gaps below 60 lines, merged
    rng = []
    for i in pk:
        if rng and i - rng[-1][1] < 60: rng[-1][1] = i
        else: rng.append([i, i])
    hot = [r for r in rng if sum(1 for i in pk if r[0] <= i <= r[1]) >= 30]
    inside = [i for i in sc if any(r[0] <= i <= r[1] for r in hot)]
    meta = {k: next((re.search(k + r"[ :]+(\d+)", l).group(1) for l in lines[b:b + 400] if re.search(k + r"[ :]+(\d+)", l)), "?") for k in ("amdhsa_next_free_vgpr", "amdhsa_private_segment_fixed_size")}
    print("%s\n  VGPRs %s, scratch %s B per lane; %d packed 16-bit instructions in %d hot range(s) %s; scratch instructions: %d, of which inside the hot ranges: %d"
          % (name, meta["amdhsa_next_free_vgpr"], meta["amdhsa_private_segment_fixed_size"], len(pk), len(hot), ["%d-%d" % (a + r[0] + 1, a + r[1] + 1) for r in hot], len(sc), len(inside)))
    for i in inside: print("    in loop: line %d %s" % (a + i + 1, body[i].strip()))
names = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
ends = [i for i, l in enumerate(lines) if "s_endpgm" in l]
for (i, n) in names:
    e = next((x for x in ends if x > i), len(lines))
    report(n, i, e)
PY
rm -f $S
