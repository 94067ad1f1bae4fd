"""python tools/pcsample/mreport.py /tmp/mcount.out [top N]: allocation calls by call site (operator new's callers are found one frame up only when inlined: sites in libstdc++ are lumped)"""
import subprocess
import sys
from collections import Counter
maps, calls = [], []
for l in open(sys.argv[1]):
    f = l.split()
    if f[0] == "M":
        a, b = (int(x, 16) for x in f[1].split("-"))
        maps.append((a, b, int(f[3], 16), f[6] if len(f) > 6 else "?"))
    elif f[0] == "C":
        calls.append((int(f[1], 16), int(f[2]), int(f[3])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
by_lib = Counter()
sites = []
for pc, n, b in calls:
    for a, e, off, name in maps:
        if a <= pc < e:
            by_lib[name.rsplit("/", 1)[-1]] += n
            sites.append((n, b, name, pc - a + off))
            break
print("calls by object:", dict(by_lib.most_common(8)))
sites.sort(reverse=True)
for n, b, name, off in sites[:top]:
    r = subprocess.run(["addr2line", "-C", "-f", "-i", "-e", name, hex(off - 1)], stdout=subprocess.PIPE).stdout.decode().split("\n")
    print("%9d calls %12d bytes  %s  %s" % (n, b, name.rsplit("/", 1)[-1], " <- ".join(x.strip()[:90] for x in r[:6] if x.strip())))
