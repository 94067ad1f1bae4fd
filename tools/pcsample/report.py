"""Attribute tools/pcsample samples to functions: python tools/pcsample/report.py /tmp/pcsample.out [library substring] [top N]"""
import bisect
import subprocess
import sys
from collections import Counter

path = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "libmm2amd"
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
maps, samples = [], []
for l in open(path):
    if l.startswith("M "):
        f = l[2:].split()
        a, b = (int(x, 16) for x in f[0].split("-"))
        maps.append((a, b, int(f[2], 16), f[5] if len(f) > 5 else "?"))
    elif l.startswith("S "):
        samples.append(int(l[2:], 16))
by_lib = Counter()
in_lib = []
lib_path = None
for pc in samples:
    for a, b, off, name in maps:
        if a <= pc < b:
            by_lib[name.rsplit("/", 1)[-1]] += 1
            if want in name:
                in_lib.append(pc - a + off)
                lib_path = name
            break
    else:
        by_lib["?"] += 1
print("%d samples; by object:" % len(samples))
for k, v in by_lib.most_common(12):
    print("  %7d %5.1f %%  %s" % (v, 100.0 * v / len(samples), k))
if not lib_path:
    sys.exit(0)
syms = []
for l in subprocess.run(["nm", "-C", "--defined-only", "-n"] + (["-D"] if "libc.so" in lib_path else []) + [lib_path], stdout=subprocess.PIPE).stdout.decode().splitlines():
    f = l.split(" ", 2)
    if len(f) == 3 and f[1] in "tTwW":
        syms.append((int(f[0], 16), f[2]))
addrs = [s[0] for s in syms]
fn = Counter()
for off in in_lib:
    i = bisect.bisect_right(addrs, off) - 1
    fn[syms[i][1] if i >= 0 else "?"] += 1
print("%d samples in %s; by function:" % (len(in_lib), lib_path))
for k, v in fn.most_common(top):
    print("  %7d %5.1f %%  %s" % (v, 100.0 * v / len(in_lib), k[:150]))
