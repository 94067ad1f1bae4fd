"""Time the register-resident splice gap-fill kernel (ksw_splice.hip) per launch class through the C ABI: intron-spanning gap fills with the
query lengths of each class (<=128: <2,pair>; 129..256: <4,pair>; 257..512: <4,strips> in one strip; 513..1024: two strips).  Prints the HIP-event
kernel times of the library's profiler and cells per second.   python tools/ksw_splice_microbench.py [jobs per class] [target length]
Measurement scaffolding."""
import sys
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from seqsim import spliced_pair
from reflib import ts_mat
import minimap2_amd as mm
import os
if os.environ.get("MM2AMD_LIB"):  # an experiment build (tools/build_variant.sh)
    mm.LIB_PATH = os.environ["MM2AMD_LIB"]

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
tl = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
rng = np.random.default_rng(3)
mat = ts_mat(1, 2, 1, 0)


def make(qlo, qhi, count):
    base = []
    for it in range(128):
        ql = int(rng.integers(qlo, qhi + 1))
        q, t = spliced_pair(rng, 2, 0.04, exon=(ql // 2 + 1, ql // 2 + 2), intron=(tl - ql - 8, tl - ql))
        base.append((q[:ql], t[:tl], -1, 200, -1, 0x08 | [0x100, 0x200][it & 1] | 0x400 | 0x800))
    return [base[i % 128] for i in range(count)]


def run(name, jobs):
    mm.ksw_exts2_batch(jobs[:256], mat, 2, 1, 32, 9)
    mm.profile_enable(True)
    mm.ksw_exts2_batch(jobs, mat, 2, 1, 32, 9)
    prof = mm.profile_get()
    mm.profile_enable(False)
    cells = sum(len(q) * len(t) for q, t, *_ in jobs)
    ms = sum(v["ms"] for v in prof.values())
    print("%s: %d jobs, %.3g cells, kernels %.2f ms %s -> %.1f Gcells/s" % (name, len(jobs), cells, ms, {k: round(v["ms"], 2) for k, v in prof.items()}, cells / ms / 1e6), flush=True)


run("q 65..128", make(65, 128, n))
run("q 129..256", make(129, 256, n))
run("q 200..256", make(200, 256, n))
run("q 257..384", make(257, 384, n // 2))
run("q 340..384", make(340, 384, n // 2))
run("q 257..512", make(257, 512, n // 2))
run("q 450..512", make(450, 512, n // 2))
run("q 513..1024", make(513, 1024, n // 4))
