"""Does a read's SAM record depend on where the read sits in the batch?  bench.py's splice workload (or --preset map-ont) mapped as rotations of the same batch;
every read's records compared across the rotations.  Run on the MI355X:  python tools/rotation_check.py [--preset splice] [--reads 50000] [--rot 0 20918 21915]
Measurement / debugging scaffolding."""
import argparse
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import minimap2_amd as mm  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--preset", default="splice")
ap.add_argument("--reads", type=int, default=50000)
ap.add_argument("--ref-mb", type=float, default=3000)
ap.add_argument("--rot", type=int, nargs="*", default=None)
ap.add_argument("--repeat", type=int, default=1, help="map every rotation this many times (run-to-run determinism)")
a = ap.parse_args()
dev = torch.device("cuda:0")
total = int(a.ref_mb * 1e6)
n_contig = max(1, min(24, total // 1000000))
codes, per = bench.gen_reference(torch, dev, 11, total, n_contig)
total = per * n_contig
genes = bench.plant_genes(torch, dev, 12, codes, per, n_contig, max(100, min(20000, total // 150000))) if a.preset == "splice" else None
refs = bench.reference_ascii(torch, dev, codes, per, n_contig)
names = ["chr%d" % (i + 1) for i in range(n_contig)]
if a.preset == "splice":
    reads = bench.gen_transcripts(torch, dev, 1000, codes, genes, a.reads, 0.05)
else:
    reads = bench.gen_reads(torch, dev, 1000, codes, per, n_contig, a.reads, 10000, 1000, 0.12)
del codes
torch.cuda.empty_cache()
al = mm.Aligner(refs, preset=a.preset, names=names, n_threads=min(16, mm.host_cpus()), sam=True)
named = [("read%d" % i, s) for i, s in enumerate(reads)]
n = len(named)
rots = a.rot if a.rot else [0, (100 * 997) % n, (101 * 997) % n]


def records(rot):
    batch = named[rot:] + named[:rot]
    al.stage(batch)
    raw = al.run(raw=True)
    txt = al.format_raw(*raw)
    al.free_raw(raw[0], raw[1])
    if isinstance(txt, str):
        txt = txt.encode()
    per_read = {}
    for line in txt.split(b"\n"):
        if not line or line.startswith(b"@"):
            continue
        per_read.setdefault(line.split(b"\t", 1)[0], []).append(line)
    return len(txt), per_read


ref_len, ref_rec = None, None
for rot in rots:
    for rep in range(a.repeat):
        ln, rec = records(rot)
        if ref_rec is None:
            ref_len, ref_rec = ln, rec
            print("rotation %d: %d bytes, %d reads with records" % (rot, ln, len(rec)))
            continue
        diff = [k for k in ref_rec if ref_rec[k] != rec.get(k)]
        print("rotation %d (pass %d): %d bytes, %d reads differ from the first" % (rot, rep, ln, len(diff)))
        for k in diff[:5]:
            for x, y in zip(ref_rec[k], rec.get(k, [])):
                if x != y:
                    fx, fy = x.split(b"\t"), y.split(b"\t")
                    print("  ", k.decode(), [(i, p.decode()[:60], q.decode()[:60]) for i, (p, q) in enumerate(zip(fx, fy)) if p != q][:6], len(fx), len(fy))
            if len(ref_rec[k]) != len(rec.get(k, [])):
                print("  ", k.decode(), "record counts", len(ref_rec[k]), len(rec.get(k, [])))
al.close()
