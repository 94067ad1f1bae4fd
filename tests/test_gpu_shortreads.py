"""Short reads (single- and paired-end, MM_F_SR / MM_F_HEAP_SORT / max_frag_len) and all-vs-all mapping on the GPU: SAM/PAF of
tests/_build/dropin_gpu (the reference's I/O around libmm2amd.so) against oracle/_ref/minimap2_ref on the same inputs, and the
device's sorted anchors (MM2AMD_SEED_DUMP) against the reference's --print-seeds."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import synth  # noqa: E402

pytestmark = pytest.mark.gpu
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "minimap2_ref")
DROPIN = os.path.join(HERE, "_build", "dropin_emu" if os.environ.get("MM2AMD_EMU") == "1" else "dropin_gpu")  # MM2AMD_EMU=1: tests/conftest.py


def _run(cmd):
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, (cmd, p.stderr.decode()[-2000:])
    return b"\n".join(l for l in p.stdout.split(b"\n") if not l.startswith(b"@PG"))


@pytest.mark.parametrize("args", [["-x", "ava-ont"], ["-x", "ava-pb"], ["-x", "ava-ont", "-c"], ["-x", "map-ont", "-D", "-c"],
                                  ["-x", "map-ont", "--dual=no"], ["-X", "-a"]])
def test_all_vs_all(args, tmp_path):  # skip_seed's read-name rules (map.c:81-91), MM_SEED_SELF (align.c:760-767)
    fa = synth.make_overlaps(str(tmp_path))
    assert _run([REF_BIN, "-t", "8"] + args + [fa, fa]) == _run([DROPIN, "-t", "8"] + args + [fa, fa])


SR_CASES = [["-x", "sr", "-a"], ["-x", "sr", "-c"], ["-x", "sr"], ["-x", "sr", "-a", "--heap-sort=no"], ["-x", "map-ont", "--heap-sort=yes", "-c"],
            ["-x", "map-ont", "-F", "2000", "-c"], ["-x", "sr", "-a", "-F", "300", "-g", "60"], ["-x", "sr", "-a", "-f", "2,20", "-N", "3"]]


@pytest.mark.parametrize("args", SR_CASES)
def test_short_reads_single_end(args, tmp_path):  # MM_F_SR / MM_F_HEAP_SORT / max_frag_len (anchor_heap_order_kernel, chain_gaps)
    ref, rd = synth.make_short(str(tmp_path))
    assert _run([REF_BIN, "-t", "8"] + args + [ref, rd]) == _run([DROPIN, "-t", "8"] + args + [ref, rd])


def test_short_reads_larger_set(tmp_path):
    ref, rd = synth.make_short(str(tmp_path), seed=93, n_reads=3000, genome=2000000)
    for args in (["-x", "sr", "-a"], ["-x", "sr", "-c", "--heap-sort=no"]):
        assert _run([REF_BIN, "-t", "8"] + args + [ref, rd]) == _run([DROPIN, "-t", "8"] + args + [ref, rd])


def _seed_blocks(lines):
    out, cur = {}, None
    for l in lines:
        f = l.split("\t")
        if f[0] == "QR":
            cur = f[1] if f[1] not in out else None  # a read seeded twice (max_occ pass): the reference prints the first pass only
            if cur is not None:
                out[cur] = []
        elif f[0] in ("SD", "RS") and cur is not None:
            out[cur].append(l)
    return out


@pytest.mark.parametrize("args", [["-x", "sr"], ["-x", "sr", "--heap-sort=no"], ["-x", "map-ont", "--heap-sort=yes"]])
def test_anchor_order_matches_print_seeds(args, tmp_path):
    """Every read's anchors, in order, from the device (MM2AMD_SEED_DUMP) against the reference's --print-seeds: the order of equal
    index hits (heap merge vs radix sort) is only visible here."""
    ref, rd = synth.make_short(str(tmp_path))
    p = subprocess.run([REF_BIN] + args + ["-t", "1", "--print-qname", "--print-seeds", ref, rd], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    assert p.returncode == 0
    want = _seed_blocks(p.stderr.decode().split("\n"))
    dump = str(tmp_path / "seeds.txt")
    subprocess.run([DROPIN] + args + ["-t", "4", ref, rd], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, MM2AMD_SEED_DUMP=dump), check=True)
    got = _seed_blocks(open(dump).read().split("\n"))
    assert len(want) > 300 and set(want) == set(got)
    bad = [k for k in want if want[k] != got[k]]
    assert not bad, bad[:5]


PE_CASES = [(["-x", "sr", "-a"], 2), (["-x", "sr", "-a"], 1), (["-x", "sr"], 2), (["-x", "sr", "-c"], 1), (["-x", "sr", "-a", "-F", "400"], 2),
            (["-x", "sr", "-a", "--heap-sort=no"], 2), (["-x", "sr", "-a", "-f", "2,20"], 1), (["-x", "sr", "-k", "15", "-w", "5", "-a"], 2)]


@pytest.mark.parametrize("args,n_files", PE_CASES)
def test_paired_end(args, n_files, tmp_path):
    """Read pairs: the mates' minimizer lists joined in seed_collect_kernel, segment-aware link_score, then the host's mm_seg_gen /
    mm_pair; two files or one interleaved file."""
    ref, f1, f2, inter = synth.make_pairs(str(tmp_path))
    files = [f1, f2] if n_files == 2 else [inter]
    assert _run([REF_BIN, "-t", "8"] + args + [ref] + files) == _run([DROPIN, "-t", "8"] + args + [ref] + files)


def test_paired_end_larger_set(tmp_path):
    ref, f1, f2, inter = synth.make_pairs(str(tmp_path), seed=97, n_pairs=4000, genome=3000000)
    assert _run([REF_BIN, "-t", "8", "-x", "sr", "-a", ref, f1, f2]) == _run([DROPIN, "-t", "8", "-x", "sr", "-a", ref, f1, f2])


def test_python_map_pairs_equals_the_reference_sam(tmp_path):
    """The mappy-shaped front end: Aligner(preset="sr").map_pairs(..., text=True) must print the reference's SAM records."""
    import minimap2_amd as mm
    ref, f1, f2, _ = synth.make_pairs(str(tmp_path), n_pairs=120)

    def fasta(path):
        names, seqs = [], []
        for line in open(path, "rb"):
            (names if line.startswith(b">") else seqs).append(line.strip().lstrip(b">"))
        return names, seqs

    rn, rs = fasta(ref)
    n1, s1 = fasta(f1)
    _, s2 = fasta(f2)
    al = mm.Aligner(rs, preset="sr", names=[x.decode() for x in rn], sam=True, n_threads=4)
    try:
        # the first mate's name goes in as the reader hands it over (with its /1): map.c:246 hashes it into the tie-break, the
        # formatter trims the suffix (format.c:529)
        got = al.map_pairs([(nm, a, b) for nm, a, b in zip(n1, s1, s2)], text=True)
        hits = al.map_pairs([(nm, a, b) for nm, a, b in zip(n1[:10], s1[:10], s2[:10])])
    finally:
        al.close()
    want = b"\n".join(l for l in _run([REF_BIN, "-t", "4", "-x", "sr", "-a", ref, f1, f2]).split(b"\n") if not l.startswith(b"@"))
    assert got.rstrip(b"\n") == want.rstrip(b"\n")
    assert len(hits) == 10 and all(len(h) == 2 for h in hits)


def _fasta(path):
    names, seqs = [], []
    for line in open(path, "rb"):
        (names if line.startswith(b">") else seqs).append(line.strip().lstrip(b">"))
    return names, seqs


def _pairs_text(rs, rn, triples):
    import minimap2_amd as mm
    al = mm.Aligner(rs, preset="sr", names=[x.decode() for x in rn], sam=True, n_threads=4)
    try:
        text = al.map_pairs(triples, text=True)
        st = al.last_stats()
    finally:
        al.close()
    return text, st


def test_read_pairs_take_the_device_region_path(tmp_path, monkeypatch):
    """Round 6: a pair's chains are cut per segment on the device (mm_seg_gen in chain_regs_kernel), each segment planned from its best run of seeds on one diagonal, the
    ungapped window settled without a DP job (region_plan_kernel), consumed and finished there: nearly every pair must be finished on the device, the SAM records must be
    the reference's, and MM2AMD_DEVICE_REGIONS=0 (the host's plan / consume rounds) must print the same text."""
    ref, f1, f2, _ = synth.make_pairs(str(tmp_path), seed=99, n_pairs=600, genome=600000)
    rn, rs = _fasta(ref)
    n1, s1 = _fasta(f1)
    _, s2 = _fasta(f2)
    triples = [(nm, a, b) for nm, a, b in zip(n1, s1, s2)]
    got, st = _pairs_text(rs, rn, triples)
    want = b"\n".join(l for l in _run([REF_BIN, "-t", "4", "-x", "sr", "-a", ref, f1, f2]).split(b"\n") if not l.startswith(b"@"))
    assert got.rstrip(b"\n") == want.rstrip(b"\n")
    assert st["n_region_reads_dev"] >= 0.9 * len(triples), st
    monkeypatch.setenv("MM2AMD_DEVICE_REGIONS", "0")
    host, st_host = _pairs_text(rs, rn, triples)
    assert host == got and st_host["n_region_reads_dev"] == 0


def test_single_end_short_reads_take_the_device_region_path(tmp_path, monkeypatch):
    import minimap2_amd as mm
    ref, rd = synth.make_short(str(tmp_path), seed=98, n_reads=800, genome=600000)
    rn, rs = _fasta(ref)
    n1, s1 = _fasta(rd)

    def run():
        al = mm.Aligner(rs, preset="sr", names=[x.decode() for x in rn], sam=True, n_threads=4)
        try:
            al.stage(list(zip([x.decode() for x in n1], s1)))
            n_reg, reg, rep_len = al.run(raw=True)
            try:
                return al.format_raw(n_reg, reg, rep_len), al.last_stats()
            finally:
                al.free_raw(n_reg, reg)
        finally:
            al.close()
    got, st = run()
    want = b"\n".join(l for l in _run([REF_BIN, "-t", "4", "-x", "sr", "-a", ref, rd]).split(b"\n") if not l.startswith(b"@"))
    assert got.rstrip(b"\n") == want.rstrip(b"\n")
    assert st["n_region_reads_dev"] >= 0.8 * len(s1), st  # (reads too short to seed have no chain: the device has nothing to do for them either)
    monkeypatch.setenv("MM2AMD_DEVICE_REGIONS", "0")
    host, st_host = run()
    assert host == got and st_host["n_region_reads_dev"] == 0


def test_read_pairs_through_the_pipeline_binding(tmp_path):
    """tests/dropin/dropin_pipeline.c (the hook inside the reference's own three-step kt_pipeline) with the reference's fragment reader in
    step 0 -- two files or one interleaved file, mini-batches that cut the input several times: the SAM stream must be the minimap2 binary's."""
    ref, f1, f2, inter = synth.make_pairs(str(tmp_path), seed=321, n_pairs=3000, genome=500000)
    pipe = os.path.join(HERE, "_build", "dropin_pipeline_emu" if os.environ.get("MM2AMD_EMU") == "1" else "dropin_pipeline_gpu")
    # an interleaved file in which every fifth read has lost its mate: single reads and pairs in one mini-batch (the device cuts a pair's chains per segment and
    # treats a single read as before; the per-read output records of such a batch come two per fragment)
    lines = open(inter, "rb").read().split(b"\n")
    recs = [(lines[i], lines[i + 1]) for i in range(0, len(lines) - 1, 2)]
    kept = []
    for k in range(0, len(recs) - 1, 2):
        kept.append(recs[k])
        if (k // 2) % 5 != 0:
            kept.append(recs[k + 1])
    mixed = os.path.join(str(tmp_path), "mixed.fa")
    with open(mixed, "wb") as f:
        f.write(b"\n".join(x for r in kept for x in r) + b"\n")
    for files in ([f1, f2], [inter], [mixed]):
        for k in ("500M", "100k"):
            assert _run([REF_BIN, "-t", "4", "-ax", "sr", "-K", k, ref] + files) == _run([pipe, "-x", "sr", "-a", "-t", "4", "-K", k, ref] + files), (files, k)
