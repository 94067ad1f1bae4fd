"""Why is a mapping call slower inside the pipeline than on its own?  Stages one BASELINE batch and maps it repeatedly (a) with the usual gaps,
(b) back to back without re-staging, (c) back to back with the output stage running beside it on another thread; samples the shader clock
(pp_dpm_sclk) meanwhile.  (GPU box)"""
import ctypes as C, glob, os, sys, threading, time
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
L = mm.lib()
clk, stop = [], [False]
def sample():
    files = glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")
    while not stop[0]:
        for f in files[:1]:
            try:
                cur = [l for l in open(f).read().split("\n") if l.endswith("*")]
                clk.append(cur[0].split()[1] if cur else "?")
            except Exception as e:
                clk.append("err")
        time.sleep(0.05)
th = threading.Thread(target=sample); th.start()
def runs(n, restage, beside=None):
    ts = []
    al.stage(b)
    keep = None
    for i in range(n):
        if restage and i: al.stage(b)
        t = time.time(); r = al.run(raw=True); ts.append(time.time() - t)
        if keep: al.free_raw(keep[0], keep[1])
        keep = r
    return ts, keep
ts, keep = runs(3, True); print("warm-up:", " ".join("%.3f" % x for x in ts), flush=True); al.free_raw(keep[0], keep[1])
del clk[:]
ts, keep = runs(8, True); print("re-staged between calls :", " ".join("%.3f" % x for x in ts), "| sclk", sorted(set(clk)), flush=True)
del clk[:]
ts2, keep2 = runs(8, False); print("back to back, no staging:", " ".join("%.3f" % x for x in ts2), "| sclk", sorted(set(clk)), flush=True); al.free_raw(keep2[0], keep2[1])
# (c) the output stage of `keep` running continuously on another thread
stop_fmt = [False]; n_fmt = [0]
def fmt():
    out, out_len = C.c_void_p(), C.c_size_t()
    while not stop_fmt[0]:
        L.mm_gpu_format_batch_view(b.n, b.seg_off, b.n_seg, b.arr, keep[0], keep[1], keep[2], C.byref(out), C.byref(out_len)); n_fmt[0] += 1
tf = threading.Thread(target=fmt); tf.start()
del clk[:]
ts3, keep3 = runs(8, False); stop_fmt[0] = True; tf.join()
print("back to back + formatting beside (%d format calls):" % n_fmt[0], " ".join("%.3f" % x for x in ts3), "| sclk", sorted(set(clk)), flush=True)
stop[0] = True; th.join()
al.close()
