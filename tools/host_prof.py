"""Host-side cycle attribution WITHOUT a GPU: the product's own host code around the wave emulator's kernels (tests/_build/libmm2amd_emu.so),
MM2AMD_HOST_PROF=1.  The emulated kernels' time means nothing; the host pieces' cycle counters (host_prof.hpp: rdtsc around the pieces on
whatever thread runs them) are the real code on a real CPU.  Usage: python tools/host_prof.py sr|map-ont [n_reads] [ref_mb]"""
import os
import sys
import time

os.environ["MM2AMD_HOST_PROF"] = "1"
os.environ["MM2AMD_EMU"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import conftest  # noqa: E402,F401  (binds the emulated library)
import minimap2_amd as mm  # noqa: E402
import synth  # noqa: E402

preset = sys.argv[1] if len(sys.argv) > 1 else "sr"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
ref_mb = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
rng = np.random.default_rng(5)
contigs = synth.gen_reference(rng, int(ref_mb * 1e6), 3)
refs = [synth.ACGT[c].tobytes() for c in contigs]
if preset == "sr":
    comp = np.array([3, 2, 1, 0], dtype=np.uint8)
    reads = []
    for i in range(n):
        c = contigs[int(rng.integers(0, len(contigs)))]
        frag = int(rng.integers(300, 600))
        s = int(rng.integers(0, len(c) - frag))
        f = c[s:s + frag].copy()
        if rng.random() < 0.5:
            f = comp[f[::-1]]
        r1, r2 = f[:150].copy(), comp[f[-150:][::-1]]
        for r in (r1, r2):
            m = rng.random(150) < 0.005
            r[m] = (r[m] + rng.integers(1, 4, int(m.sum()))) % 4
        reads.append(("p%d" % i, synth.ACGT[r1].tobytes(), synth.ACGT[r2].tobytes()))
else:
    rd = synth.gen_reads(rng, contigs, n, 10000, 1000, 0.12)
    reads = [("r%d" % i, synth.ACGT[r].tobytes()) for i, r in enumerate(rd)]
al = mm.Aligner(refs, preset=preset, n_threads=int(os.environ.get("THREADS", "4")))
t0 = time.time()
c0 = time.process_time()
for _ in range(int(os.environ.get("REPS", "1"))):
    al.stage(reads)
    raw = al.run(raw=True)
    txt = al.format_raw(*raw)
    al.free_raw(raw[0], raw[1])
print("wall %.2f s, process CPU %.2f s, %d bytes of text" % (time.time() - t0, time.process_time() - c0, len(txt)), file=sys.stderr)
al.close()
