"""Does the mapping call slow down when the GPU is kept busy without a break?  Stages one BASELINE batch and maps it N times back to back,
then N times with a pause between the calls; prints the times.  (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MM2AMD_MALLOPT", "1")
import torch
import bench, minimap2_amd as mm
dev = torch.device("cuda", 0)
codes, per = bench.gen_reference(torch, dev, 11, 3000 * 1000 * 1000, 24)
refs = bench.reference_ascii(torch, dev, codes, per, 24)
reads = bench.gen_reads(torch, dev, 1000, codes, per, 24, 100000, 10000, 1000, 0.12)
del codes; torch.cuda.empty_cache()
al = mm.Aligner(refs, preset="map-ont", names=["chr%d" % (i + 1) for i in range(24)], n_threads=64, sam=True)
b = mm.Batch([("read%d" % i, s) for i, s in enumerate(reads)])
for pause in (0.0, 0.0, 0.3, 0.0):
    ts = []
    for i in range(8):
        al.stage(b)
        t = time.time(); n_reg, reg, _ = al.run(raw=True); ts.append(time.time() - t)
        al.free_raw(n_reg, reg)
        if pause: time.sleep(pause)
    print("pause %.1f s between calls: " % pause + " ".join("%.3f" % x for x in ts), flush=True)
al.close()
