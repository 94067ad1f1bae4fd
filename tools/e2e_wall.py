"""SURVEY.md 8(d) primary figure and BASELINE.json's condition "SAM diff == 0", at full size, as programs:

    minimap2 -ax map-ont -t$(nproc) ref.mmi reads.fa      (the unmodified reference, oracle/_ref/minimap2_ref)
    dropin_pipeline_gpu -x map-ont -a ref.mmi reads.fa     (the reference's reader + kt_pipeline around mm_gpu_map_batch and
                                                            mm_gpu_format_batch: tests/dropin/dropin_pipeline.c)

on BASELINE.json configs[1] (100 k synthetic ~10 kb reads, 12 % error, 3 Gb synthetic reference; the generators are bench.py's), both
from the same .mmi.  Reports, from each program's own stamps (index.c:132 / main.c:456 "loaded/built the index" .. the last
"[M::worker_pipeline::..] mapped", map.c:638): the mapping-phase wall with parsing and SAM output overlapped, the total wall, and
whether the two SAM streams are byte-identical apart from @PG.  Also compares, once, the full-size index the DEVICE builds
(mm2amd_idx_str) with the one the reference's mm_idx_gen built (the .mmi): order-independent digests over (minimizer, position list)
by oracle/refdrv.c.

    python tools/e2e_wall.py [--ref-mb 3000] [--reads 100000] [--out gpurun_out/r02_e2e_wall.json]

Test / measurement infrastructure: runs oracle/_ref binaries; nothing here is part of the product."""
import argparse
import ctypes as C
import hashlib
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def stamps(err):
    """loaded: the index is in memory; ready: the device mirror is (drop-in only); mapped: every "[M::worker_pipeline::t*..] mapped" stamp, one per mini-batch"""
    loaded = ready = None
    mapped = []
    for line in err.splitlines():
        m = re.match(r"\[M::(main|mm_idx_stat|worker_pipeline)::([0-9.]+)\*", line)
        if not m:
            continue
        t = float(m.group(2))
        if m.group(1) == "worker_pipeline":
            mapped.append(t)
        elif "device mirror" in line:
            ready = t
        elif loaded is None and ("loaded/built the index" in line or m.group(1) == "mm_idx_stat"):
            loaded = t
    return loaded, ready, mapped


def run(cmd, out_path=None):
    """runs cmd; its standard output (SAM) is hashed on the fly without the @PG line -- a 10-Gbase run writes 13 GB of text per program, which need not touch a disk --
    and kept in out_path only when one is given.  Returns wall seconds, stderr text, (md5, lines, bytes)."""
    import threading
    t = time.time()
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, bufsize=1 << 24)
    err_chunks = []
    th = threading.Thread(target=lambda: err_chunks.append(p.stderr.read()))
    th.start()
    try:  # a digest that keeps up with the programs' output: md5 at ~0.65 GB/s throttled the pipe (the drop-in wrote 13 GB at exactly that rate, call v5); XXH3 does > 5 GB/s
        import xxhash
        h = xxhash.xxh3_128()
    except Exception:
        h = hashlib.md5()
    n, nb = 0, 0
    fo = open(out_path, "wb") if out_path else None
    header, tail = True, b""
    while True:
        buf = p.stdout.read(1 << 24)
        if not buf:
            break
        nb += len(buf)
        if fo:
            fo.write(buf)
        if not header:
            h.update(buf)
            n += buf.count(b"\n")
            continue
        data = tail + buf  # still in the header: line by line, the @PG line left out of the digest; from the first record on, whole blocks
        pos = 0
        while header:
            if pos >= len(data):
                tail = b""
                break
            if data[pos:pos + 1] != b"@":
                header = False
                break
            e = data.find(b"\n", pos)
            if e < 0:
                tail = data[pos:]
                break
            if not data.startswith(b"@PG", pos):
                h.update(data[pos:e + 1])
                n += 1
            pos = e + 1
        if not header:
            h.update(data[pos:])
            n += data.count(b"\n", pos)
            tail = b""
    if header and tail:
        h.update(tail)
    p.wait()
    th.join()
    if fo:
        fo.close()
    wall = time.time() - t
    err = b"".join(err_chunks).decode(errors="replace")
    if p.returncode != 0:
        raise RuntimeError("%s failed:\n%s" % (cmd[0], err[-2000:]))
    return wall, err, (h.hexdigest(), n, nb)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref-mb", type=float, default=3000.0)
    ap.add_argument("--reads", type=int, default=100000)
    ap.add_argument("--dir", default="/tmp/e2e")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r06_e2e_wall.json"))
    ap.add_argument("--skip-index-check", action="store_true")
    ap.add_argument("--preset", default="map-ont", choices=["map-ont", "sr"], help="sr: --reads read PAIRS of 2 x 150 b in two files (round 6: pairs take the device region path)")
    a = ap.parse_args()
    os.makedirs(a.dir, exist_ok=True)
    import torch
    import bench
    import reflib
    dev = torch.device("cuda", 0)
    ncpu = os.cpu_count() or 1
    total = int(a.ref_mb * 1e6)
    n_contig = max(1, min(24, total // 1000000))
    codes, per = bench.gen_reference(torch, dev, 11, total, n_contig)
    refs = bench.reference_ascii(torch, dev, codes, per, n_contig)
    reads, mates, chunk = [], [], max(1, min(a.reads, 100000))  # in chunks of at most ~1 Gbase, as bench.py generates them (32-bit index arithmetic inside a call)
    for c0 in range(0, a.reads, chunk):
        if a.preset == "sr":
            r1, r2 = bench.gen_pairs(torch, dev, 1000 + 7919 * (c0 // chunk), codes, per, n_contig, min(chunk, a.reads - c0), 150, 0.005)
            reads += r1
            mates += r2
        else:
            reads += bench.gen_reads(torch, dev, 1000 + 7919 * (c0 // chunk), codes, per, n_contig, min(chunk, a.reads - c0), 10000, 1000, 0.12)
        torch.cuda.empty_cache()
    del codes
    torch.cuda.empty_cache()
    names = ["chr%d" % (i + 1) for i in range(n_contig)]
    ref_fa, reads_fa, mmi = (os.path.join(a.dir, x) for x in ("ref.fa", "reads.fa", "ref.mmi"))
    with open(ref_fa, "wb") as f:
        for nm, s in zip(names, refs):
            f.write(b">" + nm.encode() + b"\n" + s + b"\n")
    with open(reads_fa, "wb") as f:
        for i, s in enumerate(reads):
            f.write(b">read%d\n" % i + s + b"\n")
    read_files = [reads_fa]
    if mates:
        mates_fa = os.path.join(a.dir, "mates.fa")
        with open(mates_fa, "wb") as f:
            for i, s in enumerate(mates):
                f.write(b">read%d\n" % i + s + b"\n")
        read_files.append(mates_fa)
    bases = sum(len(s) for s in reads) + sum(len(s) for s in mates)
    res = {"workload": ("sr: %d synthetic read pairs of 2 x 150 b (0.5%% substitutions, %.3f Gbases) vs %d Mb synthetic ref (%d contigs), -a" if mates else
                        "map-ont: %d synthetic ~10 kb reads (12%% error, %.3f Gbases) vs %d Mb synthetic ref (%d contigs), -a") % (a.reads, bases / 1e9, total // 1000000, n_contig),
           "host_threads": ncpu}
    REF = os.path.join(ROOT, "oracle", "_ref", "minimap2_ref")
    OURS = os.path.join(ROOT, "tests", "_build", "dropin_pipeline_gpu")
    t = time.time()
    subprocess.run([REF, "-x", a.preset, "-t", str(ncpu), "-d", mmi, ref_fa], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    res["reference_index_build_s"] = round(time.time() - t, 1)
    keep = a.reads <= 200000  # small runs keep the two SAM files for inspection
    w_ref, e_ref, dg_r = run([REF, "-ax", a.preset, "-t", str(ncpu), mmi] + read_files, os.path.join(a.dir, "ref.sam") if keep else None)
    w_our, e_our, dg_o = run([OURS, "-x", a.preset, "-a", "-t", str(min(64, ncpu)), mmi] + read_files, os.path.join(a.dir, "ours.sam") if keep else None)
    for key, w, e in (("reference", w_ref, e_ref), ("gpu_dropin", w_our, e_our)):
        loaded, ready, mapped_all = stamps(e)
        mapped = mapped_all[-1] if mapped_all else None
        res[key] = {"total_wall_s": round(w, 2), "index_in_memory_at_s": loaded, "last_batch_done_at_s": mapped, "mini_batches": len(mapped_all),
                    "mapping_phase_s": round(mapped - loaded, 2) if loaded is not None and mapped is not None else None,
                    "gbases_per_s_mapping_phase": round(bases / (mapped - loaded) / 1e9, 4) if loaded is not None and mapped is not None else None}
        if ready is not None:
            res[key]["device_mirror_ready_at_s"] = ready
            res[key]["device_mirror_s"] = round(ready - loaded, 2)
            res[key]["mapping_phase_after_init_s"] = round(mapped - ready, 2)
            res[key]["gbases_per_s_after_init"] = round(bases / (mapped - ready) / 1e9, 4)
            if len(mapped_all) >= 3:  # the time split the round-5 verdict asked for: mirror | first mini-batch | the rest (the reader and the writer run beside the mapping)
                res[key]["first_mini_batch_done_after_init_s"] = round(mapped_all[0] - ready, 2)
                res[key]["steady_s_per_mini_batch"] = round((mapped_all[-1] - mapped_all[0]) / (len(mapped_all) - 1), 3)
                res[key]["steady_gbases_per_s"] = round(bases * (len(mapped_all) - 1) / len(mapped_all) / (mapped_all[-1] - mapped_all[0]) / 1e9, 4)
    res["sam_lines"] = [dg_r[1], dg_o[1]]
    res["sam_identical_without_pg"] = dg_r[0] == dg_o[0]
    res["sam_digest_without_pg"] = [dg_r[0], dg_o[0]]
    res["sam_bytes"] = dg_r[2]
    res["speedup_mapping_phase"] = round(res["reference"]["mapping_phase_s"] / res["gpu_dropin"]["mapping_phase_s"], 2) if res["reference"]["mapping_phase_s"] and res["gpu_dropin"]["mapping_phase_s"] else None
    res["commit"] = os.environ.get("MM2AMD_COMMIT")
    print(json.dumps(res), flush=True)
    if not a.skip_index_check and a.preset == "map-ont":
        import minimap2_amd as mm
        import numpy as np
        D = C.CDLL(reflib.REFDRV_SO)
        R = C.CDLL(reflib.REF_SO)
        R.mm_idx_reader_open.restype = C.c_void_p
        R.mm_idx_reader_open.argtypes = [C.c_char_p, C.c_void_p, C.c_char_p]
        R.mm_idx_reader_read.restype = C.c_void_p
        R.mm_idx_reader_read.argtypes = [C.c_void_p, C.c_int]
        io, mo = mm.IdxOpt(), mm.MapOpt()
        R.mm_set_opt(None, C.byref(io), C.byref(mo))
        R.mm_set_opt(b"map-ont", C.byref(io), C.byref(mo))
        rd = R.mm_idx_reader_open(mmi.encode(), C.byref(io), None)
        mi = R.mm_idx_reader_read(rd, ncpu)
        dg_ref = (C.c_uint64 * 3)()
        D.refdrv_idx_digest.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        D.refdrv_idx_digest(mi, ncpu, dg_ref)
        R.mm_idx_destroy.argtypes = [C.c_void_p]
        R.mm_idx_destroy(mi)
        t = time.time()
        al = mm.Aligner(refs, preset="map-ont", names=names, n_threads=min(64, ncpu))
        t_build = time.time() - t
        S, keys, val_off, pos = reflib.export_index(al)
        dg_dev = (C.c_uint64 * 3)()
        D.refdrv_flat_digest.argtypes = [C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        D.refdrv_flat_digest(len(keys), keys.ctypes.data, val_off.ctypes.data, pos.ctypes.data, ncpu, dg_dev)
        al.close()
        res["index_check"] = {"reference_mm_idx_gen": {"digest": "%016x" % dg_ref[0], "distinct_minimizers": int(dg_ref[1]), "positions": int(dg_ref[2])},
                              "device_built": {"digest": "%016x" % dg_dev[0], "distinct_minimizers": int(dg_dev[1]), "positions": int(dg_dev[2]), "build_s": round(t_build, 2)},
                              "identical": list(dg_ref) == list(dg_dev)}
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
