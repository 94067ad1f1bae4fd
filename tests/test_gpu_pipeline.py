"""The code path bench.py times, under hardware parity: the reference's three-step pipeline (worker_pipeline, map.c:541-643) as this
library replaces it -- mm_gpu_batch_stage_queued (hand-over of batch k+1 beside the mapping of batch k), mm_gpu_map_staged,
mm_gpu_format_batch_view (the batch's text in the reused buffer) -- driven by minimap2_amd.Aligner.pipeline(), exactly as bench.py drives it.

  * batches of thousands of ~10 kb ONT-like reads, a short batch and an empty one in between: every batch's SAM text must equal what the
    compiled reference (oracle/_ref/minimap2_ref -ax map-ont) prints for the same reads;
  * hundreds of tiny batches (1-50 reads): the shape in which a hand-over can finish between two steps of the mapping call (the
    Mapper::run race of round 3 was found in exactly this shape, on the emulator) -- pipeline == batch by batch, repeatedly;
  * the C binding of INTEGRATION.md section 1 (tests/dropin/dropin_pipeline.c inside the reference's own kt_pipeline, now on the staged
    trio) with mini-batches from a handful of reads to hundreds: SAM == the reference binary, == the un-pipelined pair (--one-call)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import synth  # noqa: E402

pytestmark = pytest.mark.gpu
EMU = os.environ.get("MM2AMD_EMU") == "1"
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "minimap2_ref")
PIPE_BIN = os.path.join(HERE, "_build", "dropin_pipeline_emu" if EMU else "dropin_pipeline_gpu")


def _records_by_read(sam):
    """SAM text without header -> {read name: its records, in order, as one bytes object}"""
    out = {}
    for line in sam.split(b"\n"):
        if line and not line.startswith(b"@"):
            out.setdefault(line[:line.index(b"\t")], []).append(line)
    return {k: b"\n".join(v) + b"\n" for k, v in out.items()}


def _workload(tmp, ref_mb, n_reads, mean, sd, err, seed, n_contig=4):
    rng = np.random.default_rng(seed)
    contigs = synth.gen_reference(rng, int(ref_mb * 1e6), n_contig)
    reads = synth.gen_reads(rng, contigs, n_reads, mean, sd, err)
    names = ["chr%d" % (i + 1) for i in range(n_contig)]
    ref_fa, rd_fa = os.path.join(tmp, "ref.fa"), os.path.join(tmp, "reads.fa")
    synth.write_fasta(ref_fa, names, contigs)
    rnames = ["read%d" % i for i in range(n_reads)]
    synth.write_fasta(rd_fa, rnames, reads)
    refs = [synth.ACGT[c].tobytes() for c in contigs]
    rds = [(nm, synth.ACGT[r].tobytes()) for nm, r in zip(rnames, reads)]
    return ref_fa, rd_fa, refs, names, rds


@pytest.mark.timeout(3600 if EMU else 900)
def test_pipeline_text_equals_reference_binary(tmp_path):
    """(a) >= 6 batches of >= 2000 ONT reads each through Aligner.pipeline -- with a short batch, an empty batch and a one-read batch between
    them -- and the text handed to on_text for every batch == the compiled reference's SAM records of the batch's reads, in order."""
    import minimap2_amd as mm
    sizes = [40, 30, 7, 0, 1, 25] if EMU else [2000, 2300, 37, 0, 2000, 1, 2600, 2000, 2100]
    n = sum(sizes)
    ref_fa, rd_fa, refs, names, rds = _workload(str(tmp_path), 2 if EMU else 48, n, 3000 if EMU else 10000, 300 if EMU else 1000, 0.12, 71)
    p = subprocess.run([REF_BIN, "-ax", "map-ont", "-t", "16", ref_fa, rd_fa], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()[-1000:]
    by_read = _records_by_read(p.stdout)
    assert len(by_read) == n
    cuts = np.concatenate([[0], np.cumsum(sizes)])
    groups = [rds[cuts[i]:cuts[i + 1]] for i in range(len(sizes))]
    want = [b"".join(by_read[nm.encode()] for nm, _ in g) for g in groups]
    al = mm.Aligner(refs, preset="map-ont", names=names, n_threads=16, sam=True)
    try:
        got, seen = [], []
        total = al.pipeline([mm.Batch(g) for g in groups], on_text=lambda b, addr, ln: (got.append(C.string_at(addr, ln)), seen.append(b.n)))
        # the same batches a second time through the same context: nothing may survive from one pass to the next
        again = []
        al.pipeline([mm.Batch(g) for g in reversed(groups)], on_text=lambda b, addr, ln: again.append(C.string_at(addr, ln)))
    finally:
        al.close()
    assert seen == sizes and total == sum(len(t) for t in want)
    for k, (g, w) in enumerate(zip(got, want)):
        if g != w:
            gl, wl = g.split(b"\n"), w.split(b"\n")
            bad = [i for i in range(min(len(gl), len(wl))) if gl[i] != wl[i]]
            raise AssertionError("batch %d (%d reads): %d of %d lines differ; first: %r" % (k, sizes[k], len(bad), len(wl), wl[bad[0]][:200] if bad else None))
    assert again == want[::-1]
    assert sum(1 for t in want if t.count(b"\n") > 1000) >= (0 if EMU else 6)


@pytest.mark.timeout(3600 if EMU else 900)
def test_pipeline_of_tiny_batches_equals_batch_by_batch():
    """(c) 200 batches of 1-50 short reads (and a few empty ones): with hand-overs that take microseconds, the stager is always a batch
    ahead of the mapper -- the text of every batch must be that of stage + run + format of the same batch, in several passes."""
    import minimap2_amd as mm
    rng = np.random.default_rng(5)
    contigs = synth.gen_reference(rng, 1500000, 2)
    pool = synth.gen_reads(rng, contigs, 120 if EMU else 600, 1500, 400, 0.1, min_len=200)
    refs = [synth.ACGT[c].tobytes() for c in contigs]
    rds = [("r%d" % i, synth.ACGT[r].tobytes()) for i, r in enumerate(pool)]
    n_batches = 40 if EMU else 200
    batches = []
    for k in range(n_batches):
        m = 0 if k % 41 == 17 else int(rng.integers(1, 51))
        pick = rng.integers(0, len(rds), m)
        batches.append(mm.Batch([rds[i] for i in pick]))
    al = mm.Aligner(refs, preset="map-ont", names=["chr1", "chr2"], n_threads=8, sam=True)
    try:
        want = []
        for b in batches:
            al.stage(b)
            n_reg, reg, rep = al.run(raw=True)
            want.append(al.format_raw(n_reg, reg, rep))
            al.free_raw(n_reg, reg)
        for rep_no in range(2 if EMU else 4):
            got = []
            total = al.pipeline(batches, on_text=lambda b, addr, ln: got.append(C.string_at(addr, ln)))
            bad = [k for k in range(n_batches) if got[k] != want[k]]
            assert not bad, "pass %d: batches %s differ (of %d)" % (rep_no, bad[:10], n_batches)
            assert total == sum(len(t) for t in want)
    finally:
        al.close()
    assert sum(len(t) for t in want) > 100000


@pytest.mark.timeout(3600 if EMU else 900)
def test_pipeline_early_start_across_batches(monkeypatch):
    """Round 4: the lanes start on the next handed-over batch while the current batch's last sub-batches finish (Mapper::stage(may_start_early)).
    Batches of a few hundred reads cut into sub-batches of 40 reads (a dozen per batch, five lanes): pipeline text == batch by batch, early starts
    counted; then the same batches with MM2AMD_NO_EARLY_START-equivalent un-pipelined calls."""
    import minimap2_amd as mm
    monkeypatch.setenv("MM2AMD_SUBBATCH_READS", "8" if EMU else "40")
    rng = np.random.default_rng(9)
    contigs = synth.gen_reference(rng, 2000000, 2)
    pool = synth.gen_reads(rng, contigs, 60 if EMU else 900, 2500, 600, 0.1, min_len=300)
    refs = [synth.ACGT[c].tobytes() for c in contigs]
    rds = [("r%d" % i, synth.ACGT[r].tobytes()) for i, r in enumerate(pool)]
    n_batches = 6 if EMU else 24
    batches = []
    for k in range(n_batches):
        m = 0 if k == 3 else int(rng.integers(20, 50) if EMU else rng.integers(200, 520))
        pick = rng.integers(0, len(rds), m)
        batches.append(mm.Batch([rds[i] for i in pick]))
    al = mm.Aligner(refs, preset="map-ont", names=["chr1", "chr2"], n_threads=8, sam=True)
    try:
        want = []
        for b in batches:
            al.stage(b)
            n_reg, reg, rep = al.run(raw=True)
            want.append(al.format_raw(n_reg, reg, rep))
            al.free_raw(n_reg, reg)
        n_early = [0]
        def on_mapped(b, n_reg, reg, rep_len):
            n_early[0] += int(al.last_stats()["n_early_sub"])
        for rep_no in range(2 if EMU else 3):
            got = []
            total = al.pipeline(batches, on_text=lambda b, addr, ln: got.append(C.string_at(addr, ln)), on_mapped=on_mapped)
            bad = [k for k in range(n_batches) if got[k] != want[k]]
            assert not bad, "pass %d: batches %s differ (of %d)" % (rep_no, bad[:10], n_batches)
            assert total == sum(len(t) for t in want)
        assert n_early[0] > 0, "no sub-batch was started before its batch's mapping call"
    finally:
        al.close()


def _sam(cmd, env=None):
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert p.returncode == 0, (cmd, p.stderr.decode()[-2000:])
    return b"\n".join(l for l in p.stdout.split(b"\n") if not l.startswith(b"@PG")), p.stderr.decode()


@pytest.mark.timeout(3600 if EMU else 600)
@pytest.mark.parametrize("batch", ["40k", "300k", "5M"])
def test_staged_c_pipeline_equals_reference(tmp_path, batch):
    """(b) dropin_pipeline.c = INTEGRATION.md section 1: mm_gpu_batch_stage_queued at the end of step 0, mm_gpu_map_staged as step 1,
    mm_gpu_format_batch_view as step 2, inside the reference's kt_pipeline; mini-batches of 40 kbases (a handful of reads: dozens of
    hand-overs racing the mapper), 300 kbases and 5 Mbases"""
    assert os.path.exists(PIPE_BIN) and os.path.exists(REF_BIN)
    ref, reads, _, _ = synth.make("ont", str(tmp_path), 2 if EMU else 6, 60 if EMU else 700, 59)
    want, _ = _sam([REF_BIN, "-x", "map-ont", "-a", "-t", "8", ref, reads])
    got, err = _sam([PIPE_BIN, "-x", "map-ont", "-a", "-t", "8", "-K", batch, ref, reads])
    assert got == want
    n_batches = err.count("[M::worker_pipeline::")
    assert n_batches >= {"40k": 12, "300k": 2, "5M": 1}[batch]
    if batch == "300k":
        one, _ = _sam([PIPE_BIN, "-x", "map-ont", "-a", "-t", "8", "-K", batch, "--one-call", ref, reads])
        assert one == want
