"""CPU-only end-to-end check of the product's HOST pipeline (chains -> hits -> window planning -> CIGAR stitching -> MAPQ):
tests/_build/dropin_check links the same host sources as libmm2amd.so against the oracle-backed checker backend
(tests/cpucheck/backend_check.cpp) and must reproduce the committed golden output of the unmodified reference
(tests/golden/*.out, made by tests/golden/make_golden.py).  When the compiled reference is present it is re-run too, which
pins the goldens themselves."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as G  # noqa: E402

CHECK = os.path.join(HERE, "_build", "dropin_check")
REFTEST = "/root/reference/test"


def _build():
    if os.path.exists("/root/reference/minimap.h") or not os.path.exists(CHECK):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
        subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "cpucheck")], stdout=subprocess.DEVNULL)


@pytest.fixture(scope="module", autouse=True)
def built():
    _build()
    if not os.path.exists(CHECK):
        pytest.skip("tests/_build/dropin_check needs the reference headers to build (dev container only)")


@pytest.mark.parametrize("case", list(G.CASES))
def test_golden(case, tmp_path):
    meta = json.load(open(os.path.join(HERE, "golden", "inputs.json")))
    want = open(os.path.join(HERE, "golden", case + ".out"), "rb").read()
    got, m = G.run_case(CHECK, case, str(tmp_path))
    assert m == meta[case], "synthetic inputs drifted from the ones the golden was made with"
    assert got == want
    if os.path.exists(G.REF_BIN):
        ref, _ = G.run_case(G.REF_BIN, case, str(tmp_path))
        assert ref == want, "golden fixture no longer matches the compiled reference"


def _pair(args, a, b):
    outs = []
    for binary in (G.REF_BIN, CHECK):
        p = subprocess.run([binary] + args + [a, b], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode == 0, p.stderr.decode()[-1500:]
        outs.append(G.strip_pg(p.stdout))
    assert outs[0] == outs[1]
    return outs[0]


@pytest.mark.parametrize("case", list(G.FIXTURE_CASES))
def test_reference_fixtures(case):
    """the reference's OWN test inputs (test/MT-*.fa = BASELINE.json configs[0], test/t-inv.fa x q-inv.fa, test/x3s-*.fa; vendored under
    tests/golden/ref_fixtures/): host pipeline + oracle backend == committed golden == the compiled reference when it is present"""
    want = open(os.path.join(HERE, "golden", case + ".out"), "rb").read()
    got, _ = G.run_fixture(CHECK, case)
    assert got == want
    if os.path.exists(G.REF_BIN):
        ref, _ = G.run_fixture(G.REF_BIN, case)
        assert ref == want, "golden fixture no longer matches the compiled reference"
    if os.path.isdir(REFTEST):  # the vendored copies are the reference's files, byte for byte
        for f in G.FIXTURE_CASES[case][:2]:
            assert open(os.path.join(G.FIXDIR, f), "rb").read() == open(os.path.join(REFTEST, f), "rb").read()
    if case == "mt_sam":
        assert b"\t577\t60\t" in want  # one record, pos 577, MAPQ 60
    if case == "inv_paf":
        assert want.count(b"tp:A:I") == 2  # the inversion rescue path (align.c:916-971)
    if case == "x3s_paf":
        assert b"cg:Z:69M134N65M" in want and b"ts:A:+" in want  # test/x3s-aln.txt:1


def test_empty_and_tiny_reads(tmp_path):
    import numpy as np
    import synth
    rng = np.random.default_rng(5)
    contigs = synth.gen_reference(rng, 300000, 2)
    reads = synth.gen_reads(rng, contigs, 6, 3000, 300, 0.1)
    reads += [np.zeros(0, dtype=np.uint8), contigs[0][:10].copy(), contigs[1][100:140].copy()]
    ref, rd = str(tmp_path / "r.fa"), str(tmp_path / "q.fa")
    synth.write_fasta(ref, ["c1", "c2"], contigs)
    with open(rd, "wb") as f:
        for i, s in enumerate(reads):
            f.write(b">r%d\n" % i + synth.ACGT[s].tobytes() + b"\n")
        f.write(b">withN\n" + synth.ACGT[contigs[0][5000:6000]].tobytes()[:500] + b"NNNNNNNNNN" + synth.ACGT[contigs[0][5510:7000]].tobytes() + b"\n")
    if os.path.exists(G.REF_BIN):
        _pair(["-a"], ref, rd)


def test_single_affine_scoring(tmp_path):
    """equal first/second gap costs route the DP through ksw_extz2 (align.c:353-354); host logic + oracle backend vs the reference"""
    import synth
    if not os.path.exists(G.REF_BIN):
        pytest.skip("needs oracle/_ref")
    ref, reads, _, _ = synth.make("ont", str(tmp_path), 0.5, 20, 105)
    _pair(["-x", "map-ont", "-a", "-O", "4", "-E", "2"], ref, reads)


@pytest.mark.parametrize("args", [["-x", "splice", "-a"], ["-x", "splice:hq", "-c", "--cs"], ["-x", "splice", "-a", "-u", "f"],
                                  ["-x", "splice", "-c", "-u", "n", "-J", "0"], ["-x", "splice", "-a", "-G", "10000", "-C", "5", "--splice-flank=no"]])
def test_splice(tmp_path, args):
    """spliced alignment (-x splice): cDNA chaining, per-strand alignment with the splice-aware DP (oracle backend), strand pick,
    N operations and ts tags -- host logic vs the reference on synthetic multi-exon transcripts"""
    import synth
    if not os.path.exists(G.REF_BIN):
        pytest.skip("needs oracle/_ref")
    ref, reads, _, _ = synth.make("cdna", str(tmp_path), 2.0, 60 if args[1] == "splice" and "-u" not in args and "-G" not in args else 45, 106)  # the oracle-backed splice DP is slow: CPU suite time
    out = _pair(args, ref, reads)
    assert any(b"N" in l.split(b"\t")[5] for l in out.split(b"\n") if l and not l.startswith(b"@") and b"\t" in l) or "-c" in args


FORMAT_CASES = [("ont", ["-x", "map-ont", "-a"]), ("ont", ["-x", "map-ont", "-c", "--cs"]), ("ont", ["-x", "map-ont", "-a", "--MD", "-Y"]),
                ("ont", ["-x", "map-ont", "-c", "--ds"]), ("ont", ["-x", "map-ont", "-a", "--cs=long", "--eqx"]), ("cdna", ["-x", "splice", "-a", "--cs"]),
                ("cdna", ["-x", "splice", "-c", "--MD", "--secondary=no"]), ("hifi", ["-x", "map-hifi", "-a", "--secondary-seq", "-L"]),
                ("hifi", ["-x", "map-hifi", "--paf-no-hit", "-c"]), ("fastq", ["-x", "map-ont", "-a", "-y"]), ("fastq", ["-x", "map-ont", "-c", "-y", "--cs"])]


@pytest.mark.parametrize("kind,args", FORMAT_CASES)
def test_output_stage(tmp_path, kind, args):
    """mm_gpu_format_batch (the parallel host output stage) against the reference's own writers: SAM and PAF,
    cg/cs/ds/MD/ts/SA tags, clipping modes, comments and qualities, unmapped records"""
    import numpy as np
    import synth
    if not os.path.exists(G.REF_BIN):
        pytest.skip("needs oracle/_ref")
    if kind == "fastq":  # qualities, comments, chimeric reads (supplementary records with SA tags), a read that maps nowhere
        rng = np.random.default_rng(3)
        contigs = synth.gen_reference(rng, 1500000, 2)
        reads = synth.gen_reads(rng, contigs, 20, 4000, 500, 0.08)
        for i in range(8):
            a, b = contigs[0][10000 * i + 5000:10000 * i + 8000], synth.COMP[contigs[1][20000 * i + 100:20000 * i + 2600][::-1]]
            reads.append(synth.mutate_read(rng, np.concatenate([a, b]), 0.05))
        reads.append(rng.integers(0, 4, 3000, dtype=np.uint8))
        ref, rd = str(tmp_path / "ref.fa"), str(tmp_path / "reads.fq")
        synth.write_fasta(ref, ["c1", "c2"], contigs)
        with open(rd, "wb") as f:
            for i, s in enumerate(reads):
                f.write(b"@r%d comment %d\n" % (i, i) + synth.ACGT[s].tobytes() + b"\n+\n" + bytes(rng.integers(35, 74, len(s), dtype=np.uint8)) + b"\n")
    else:
        ref, rd, _, _ = synth.make(kind, str(tmp_path), 1.5, 40, 108)
    outs = []
    for cmd in ([G.REF_BIN] + args, [CHECK, "--format-lib"] + args):
        p = subprocess.run(cmd + ["-t", "4", ref, rd], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode == 0, p.stderr.decode()[-1500:]
        outs.append(G.strip_pg(p.stdout))
    assert outs[0] == outs[1]
    assert len(outs[0]) > 1000


def _contigs(tmp_path):
    """assembly-like queries: 100-300 kb pieces of the reference at 1-2 % divergence, one carrying a 5 kb deletion and a 3 kb inversion"""
    import numpy as np
    import synth
    rng = np.random.default_rng(41)
    contigs = synth.gen_reference(rng, 3000000, 2)
    reads = synth.gen_reads(rng, contigs, 5, 250000, 30000, 0.02, min_len=100000)
    s = contigs[0][200000:600000].copy()
    s = np.concatenate([s[:100000], s[105000:250000], synth.COMP[s[250000:253000][::-1]], s[253000:]])
    reads.append(synth.mutate_read(rng, s, 0.01))
    ref, rd = str(tmp_path / "ref.fa"), str(tmp_path / "contigs.fa")
    synth.write_fasta(ref, ["c1", "c2"], contigs)
    synth.write_fasta(rd, ["q%d" % i for i in range(len(reads))], reads)
    return ref, rd


@pytest.mark.parametrize("args", [["-x", "asm20", "-c"], ["-x", "asm5", "-a"], ["-x", "lr:hqae", "-c", "--cs"]])
def test_rmq_presets(tmp_path, args):
    """MM_F_RMQ presets: mg_lchain_rmq as the primary chainer (map.c:275-277), here on the host over the backend's sorted anchors"""
    if not os.path.exists(G.REF_BIN):
        pytest.skip("needs oracle/_ref")
    ref, rd = _contigs(tmp_path)
    out = _pair(args, ref, rd)
    assert len(out) > 500


def test_alt_contigs(tmp_path):
    """--alt: hits on ALT contigs are marked (mm_mark_alt, hit.c:90-98) and handicapped in the sorts and the parent assignment"""
    import synth
    if not os.path.exists(G.REF_BIN):
        pytest.skip("needs oracle/_ref")
    ref, rd, alt = synth.make_alt(str(tmp_path))
    with_alt = _pair(["-x", "map-ont", "-a", "--alt", alt], ref, rd)
    without = _pair(["-x", "map-ont", "-a"], ref, rd)
    assert with_alt != without  # the list matters on these inputs


def test_rechain_with_raised_occurrence_cap(tmp_path):
    """-f mid,max: reads that find no chain because all their minimizers are too frequent are seeded again with the cap raised to
    max_occ (map.c:293-316)"""
    import synth
    if not os.path.exists(G.REF_BIN):
        pytest.skip("needs oracle/_ref")
    ref, rd = synth.make_repeats(str(tmp_path))
    raised = _pair(["-x", "map-ont", "-c", "-f", "3,50", "-e", "0"], ref, rd)
    plain = _pair(["-x", "map-ont", "-c", "-f", "3", "-e", "0"], ref, rd)
    assert raised.count(b"\n") > 2 * plain.count(b"\n")  # the repeat-only reads map only with the raised cap


@pytest.mark.parametrize("args", [["-x", "map-ont", "-a"], ["-x", "map-hifi", "-c", "--cs"], ["-x", "asm20", "-c"], ["-x", "map-pb", "-a"]])
def test_edge_case_reads(tmp_path, args):
    import synth
    if not os.path.exists(G.REF_BIN):
        pytest.skip("needs oracle/_ref")
    ref, rd = synth.make_weird(str(tmp_path))
    _pair(args, ref, rd)


@pytest.mark.parametrize("kind,args", [("ont", ["-x", "map-ont"]), ("cdna", ["-x", "splice"]), ("hifi", ["-x", "asm20"]), ("ont", ["-x", "map-ont", "--paf-no-hit", "-P"])])
def test_chain_level_mapping(tmp_path, kind, args):
    """no -c / -a: the hits are the chains (align_regs returns early, map.c:217), dv instead of de, index built without sequence"""
    import synth
    if not os.path.exists(G.REF_BIN):
        pytest.skip("needs oracle/_ref")
    ref, rd, _, _ = synth.make(kind, str(tmp_path), 1.0, 40, 109)
    out = _pair(args, ref, rd)
    assert b"dv:f:" in out and b"cg:Z:" not in out


@pytest.mark.parametrize("flag", ["--for-only", "--rev-only"])
def test_one_strand_only(tmp_path, flag):
    import synth
    if not os.path.exists(G.REF_BIN):
        pytest.skip("needs oracle/_ref")
    ref, rd, _, _ = synth.make("ont", str(tmp_path), 1.0, 40, 110)
    out = _pair(["-x", "map-ont", "-c", flag], ref, rd)
    strands = {l.split(b"\t")[4] for l in out.split(b"\n") if l}
    assert strands == ({b"+"} if flag == "--for-only" else {b"-"})


@pytest.mark.skipif(not os.path.exists(G.REF_BIN), reason="needs the compiled reference")
@pytest.mark.parametrize("args", [["-x", "ava-ont"], ["-x", "ava-pb"], ["-x", "ava-ont", "-c"], ["-x", "map-ont", "-D", "-c"],
                                  ["-x", "map-ont", "--dual=no"], ["-X", "-a"]])
def test_all_vs_all(args, tmp_path):  # skip_seed's read-name rules (map.c:81-91), MM_SEED_SELF (align.c:760-767)
    import synth
    fa = synth.make_overlaps(str(tmp_path))
    out = _pair(args, fa, fa)
    assert out.count(b"\n") > 50


SR_CASES = [["-x", "sr", "-a"], ["-x", "sr", "-c"], ["-x", "sr"], ["-x", "sr", "-a", "--heap-sort=no"], ["-x", "map-ont", "--heap-sort=yes", "-c"],
            ["-x", "map-ont", "-F", "2000", "-c"], ["-x", "sr", "-a", "-F", "300", "-g", "60"], ["-x", "sr", "-a", "-f", "2,20", "-N", "3"]]


@pytest.mark.skipif(not os.path.exists(G.REF_BIN), reason="needs the compiled reference")
@pytest.mark.parametrize("args", SR_CASES)
def test_short_reads_single_end(args, tmp_path):  # MM_F_SR / MM_F_HEAP_SORT / max_frag_len paths (map.c:102-166,262-271; align.c:563-589,696-704,803-833)
    import synth
    ref, rd = synth.make_short(str(tmp_path))
    out = _pair(args, ref, rd)
    assert out.count(b"\n") > 300


@pytest.mark.skipif(not os.path.exists(G.REF_BIN), reason="needs the compiled reference")
@pytest.mark.parametrize("heap", ["yes", "no"])
def test_anchor_order_matches_print_seeds(heap, tmp_path):
    """The anchors of every read, in order, against the reference's --print-seeds dump: the only place where the order of equal
    index hits (heap merge vs radix sort) is directly visible."""
    import synth
    ref, rd = synth.make_short(str(tmp_path))

    def blocks(lines):
        out, cur = {}, None
        for l in lines:
            f = l.split("\t")
            if f[0] == "QR":
                cur = f[1] if f[1] not in out else None  # a read seeded twice (max_occ pass): the first dump is the one the reference prints
                if cur is not None:
                    out[cur] = []
            elif f[0] in ("SD", "RS") and cur is not None:
                out[cur].append(l)
        return out

    args = ["-x", "sr", "--heap-sort=" + heap, "-t", "1"]
    p = subprocess.run([G.REF_BIN] + args + ["--print-qname", "--print-seeds", ref, rd], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    assert p.returncode == 0
    want = blocks(p.stderr.decode().split("\n"))
    dump = str(tmp_path / "seeds.txt")
    env = dict(os.environ, MM2AMD_CHECK_SEED_DUMP=dump)
    subprocess.run([CHECK] + args + [ref, rd], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=env, check=True)
    got = blocks(open(dump).read().split("\n"))
    assert len(want) > 300 and set(want) == set(got)
    assert sum(1 for k in want if any(a.split("\t")[1:3] == b.split("\t")[1:3] for a, b in zip(want[k][1:], want[k][2:]))) > 10  # reads with equal hits exist
    bad = [k for k in want if want[k] != got[k]]
    assert not bad, bad[:5]


def test_device_heap_order_header_against_oracle():
    """minimap2_amd/csrc/heap_order.hpp (what anchor_heap_order_kernel runs per read) compiled for the host, vs the oracle."""
    exe = os.path.join(HERE, "_build", "heap_order_test")
    if not os.path.exists(exe):
        pytest.skip("tests/_build/heap_order_test not built")
    out = subprocess.run([exe, "3000"], stdout=subprocess.PIPE, check=True).stdout.split()
    assert out[0] == b"OK" and int(out[2]) > 1000


def test_device_sketch_header_against_oracle():
    """minimap2_amd/csrc/sketch_dev.hpp (the minimizer automaton the sketch kernels run per stretch of a read) compiled for the host:
    the stretches' outputs concatenate to mm_sketch's list, and a stretch reports owned positions only, each at most once (what the
    one-pass staging of sketch_wave_kernel relies on) -- N runs, homopolymers, repeats, strand-symmetric k-mers, HPC."""
    exe = os.path.join(HERE, "_build", "sketch_test")
    if not os.path.exists(exe):
        pytest.skip("tests/_build/sketch_test not built")
    out = subprocess.run([exe, "400"], stdout=subprocess.PIPE, check=True).stdout.split()
    assert out[0] == b"OK" and int(out[2]) > 100000


PE_CASES = [(["-x", "sr", "-a"], 2), (["-x", "sr", "-a"], 1), (["-x", "sr"], 2), (["-x", "sr", "-c"], 1), (["-x", "sr", "-a", "-F", "400"], 2),
            (["-x", "sr", "-a", "--heap-sort=no"], 2), (["-x", "sr", "-a", "-f", "2,20"], 1), (["-x", "sr", "-a", "-p", "0.3", "-N", "4"], 1),
            (["-x", "sr", "-a", "-g", "300", "-r", "50"], 2), (["-x", "sr", "-k", "15", "-w", "5", "-a"], 2), (["-x", "sr", "-a", "-A", "1", "-B", "3"], 2),
            (["-x", "map-ont", "-a"], 2), (["-x", "sr", "-a", "--no-pairing"], 2), (["-x", "sr", "-c", "--no-pairing"], 1)]


@pytest.mark.skipif(not os.path.exists(G.REF_BIN), reason="needs the compiled reference")
@pytest.mark.parametrize("args,n_files", PE_CASES)
def test_paired_end(args, n_files, tmp_path):
    """Two-segment fragments: joint seeding/chaining of the mates, mm_seg_gen, per-mate alignment, mm_pair (map.c:343-354, hit.c:342-396,
    pe.c), from two files or one interleaved file; SAM mate fields come from the reference's writer fed with our hit records."""
    import synth
    ref, f1, f2, inter = synth.make_pairs(str(tmp_path))
    outs = []
    for binary in (G.REF_BIN, CHECK):
        p = subprocess.run([binary] + args + [ref] + ([f1, f2] if n_files == 2 else [inter]), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode == 0, p.stderr.decode()[-1500:]
        outs.append(G.strip_pg(p.stdout))
    assert outs[0] == outs[1]
    assert outs[0].count(b"\n") > 500


@pytest.mark.skipif(not os.path.exists(G.REF_BIN), reason="needs the compiled reference")
@pytest.mark.parametrize("args,n_files", [(["-x", "sr", "-a"], 2), (["-x", "sr", "-a"], 1), (["-x", "sr"], 2), (["-x", "sr", "-c", "--cs"], 1),
                                          (["-x", "sr", "-a", "--MD", "-Y"], 2), (["-x", "sr", "-a", "-N", "3", "-p", "0.3", "--secondary-seq"], 2),
                                          (["-x", "sr", "-a", "--sam-hit-only"], 2), (["-x", "sr", "--paf-no-hit"], 1)])
def test_paired_end_records_from_the_library(args, n_files, tmp_path):  # mm_gpu_format_batch: mate fields of mm_write_sam3 (format.c:529-613), /1 /2 of mm_write_paf4
    import synth
    ref, f1, f2, inter = synth.make_pairs(str(tmp_path))
    files = [f1, f2] if n_files == 2 else [inter]
    want = subprocess.run([G.REF_BIN] + args + [ref] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True).stdout
    got = subprocess.run([CHECK] + args + ["--format-lib", ref] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True).stdout
    assert G.strip_pg(want) == G.strip_pg(got)


def test_staged_calls_equal_the_batch_call(tmp_path):  # mm_gpu_batch_stage + mm_gpu_map_staged (what bench.py and the Python Aligner use)
    import synth
    ref, rd, _, _ = synth.make("ont", str(tmp_path), 2, 60, 21)
    a = subprocess.run([CHECK, "-x", "map-ont", "-a", ref, rd], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True).stdout
    b = subprocess.run([CHECK, "-x", "map-ont", "-a", "--staged", "-K", "200000", ref, rd], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True).stdout
    assert G.strip_pg(a) == G.strip_pg(b) and a.count(b"\n") > 50
    ref, f1, f2, _ = synth.make_pairs(str(tmp_path / "pe"), n_pairs=80)  # read pairs through the staged calls
    a = subprocess.run([CHECK, "-x", "sr", "-a", ref, f1, f2], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True).stdout
    b = subprocess.run([CHECK, "-x", "sr", "-a", "--staged", "--format-lib", ref, f1, f2], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True).stdout
    assert G.strip_pg(a) == G.strip_pg(b) and a.count(b"\n") > 100


@pytest.mark.skipif(not os.path.exists(G.REF_BIN), reason="needs the compiled reference")
def test_three_step_pipeline_driver_equals_the_reference(tmp_path):
    """tests/dropin/dropin_pipeline.c: the reference's kt_pipeline with step 1 = mm_gpu_map_batch and step 2 = mm_gpu_format_batch (the
    output stage of batch k beside the mapping of batch k+1: the two calls hold the context shared, not exclusively)."""
    import synth
    ref, rd, _, _ = synth.make("ont", str(tmp_path), 2, 50, 27)
    exe = os.path.join(HERE, "_build", "dropin_pipeline_check")
    for args in (["-x", "map-ont", "-a", "-K", "60k"], ["-x", "map-ont", "-c", "-K", "2M"]):
        want = subprocess.run([G.REF_BIN] + args + ["-t", "4", ref, rd], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True).stdout
        p = subprocess.run([exe] + args + ["-t", "4", ref, rd], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
        assert G.strip_pg(want) == G.strip_pg(p.stdout)
        assert b"[M::worker_pipeline::" in p.stderr and b"loaded/built the index" in p.stderr


@pytest.mark.skipif(not os.path.exists(G.REF_BIN), reason="needs the compiled reference")
def test_batch_of_one_calls_with_the_reference_signatures(tmp_path):
    """mm_gpu_map / mm_gpu_map_frag (the reference's mm_map / mm_map_frag signatures, include/mm2amd.h): every fragment mapped by a call of
    its own, single reads and read pairs, rep_len / frag_gap through the mm_tbuf_t -- same records as the reference."""
    import synth
    ref, rd, _, _ = synth.make("ont", str(tmp_path), 1, 12, 29)
    _pair(["-x", "map-ont", "-a"], ref, rd)
    got = subprocess.run([CHECK, "-x", "map-ont", "-a", "--one-by-one", ref, rd], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True).stdout
    want = subprocess.run([G.REF_BIN, "-x", "map-ont", "-a", ref, rd], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True).stdout
    assert G.strip_pg(got) == G.strip_pg(want)
    ref, f1, f2, _ = synth.make_pairs(str(tmp_path / "pe"), n_pairs=25)
    got = subprocess.run([CHECK, "-x", "sr", "-a", "--one-by-one", ref, f1, f2], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True).stdout
    want = subprocess.run([G.REF_BIN, "-x", "sr", "-a", ref, f1, f2], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True).stdout
    assert G.strip_pg(got) == G.strip_pg(want)


def test_in_process_replicas_equal_one_mapper(tmp_path):
    """mm_gpu_init_multi's dispatcher (capi_map.cpp: shard_by_bases, one Mapper per replica on its own host thread, results into the
    caller's arrays): 1, 2 and 5 replicas print the same records -- long reads, staged calls, read pairs (joint, and mates mapped
    separately and paired afterwards, which must not be cut apart), more replicas than reads."""
    import synth
    def run(args, n):
        return G.strip_pg(subprocess.run([CHECK] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True, env=dict(os.environ, MM2AMD_GPUS=str(n))).stdout)
    ref, rd, _, _ = synth.make("ont", str(tmp_path), 2, 45, 23)
    one = run(["-x", "map-ont", "-a", ref, rd], 1)
    assert one.count(b"\n") > 40
    for n in (2, 5):
        assert run(["-x", "map-ont", "-a", ref, rd], n) == one
        assert run(["-x", "map-ont", "-a", "--staged", "-K", "150000", ref, rd], n) == one
    assert run(["-x", "map-ont", "-a", "-K", "12000", ref, rd], 16) == one  # mini-batches of one or two reads: most replicas get nothing
    ref, f1, f2, _ = synth.make_pairs(str(tmp_path / "pe"), n_pairs=70)
    for args in (["-x", "sr", "-a", ref, f1, f2], ["-x", "sr", "-a", "--no-pairing", ref, f1, f2]):
        assert run(args, 3) == run(args, 1)
    ref, r1, r2, bed = synth.make_rna_pairs(str(tmp_path / "rna"))
    assert run(["-x", "splice:sr", "-a", ref, r1, r2], 4) == run(["-x", "splice:sr", "-a", ref, r1, r2], 1)  # MM_F_WEAK_PAIRING


JUNC_CASES = [["-x", "splice", "-a"], ["-x", "splice", "-a", "--junc-bonus", "20"], ["-x", "splice:hq", "-a", "-u", "n"], ["-x", "splice", "-c", "--cs", "-u", "f"]]


@pytest.mark.skipif(not os.path.exists(G.REF_BIN), reason="needs the compiled reference")
@pytest.mark.parametrize("args", JUNC_CASES)
def test_junction_annotation(args, tmp_path):
    """--junc-bed: the annotated introns inside each DP window (mm_idx_bed_junc, index.c:803-826) reach the splice DP as bonus on
    donor / acceptor costs (ksw2_exts2_sse.c:201-217); most output lines change with the annotation, and must change the same way."""
    import synth
    ref, rd, bed = synth.make_junctions(str(tmp_path))
    out = _pair(args + ["--junc-bed", bed], ref, rd)
    plain = subprocess.run([G.REF_BIN] + args + [ref, rd], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True).stdout
    if "n" not in args:  # without an assumed transcript strand (-u n) there are no donor / acceptor costs to add the bonus to
        assert sum(1 for a, b in zip(out.split(b"\n"), G.strip_pg(plain).split(b"\n")) if a != b) > 10


@pytest.mark.skipif(not os.path.exists(G.REF_BIN), reason="needs the compiled reference")
@pytest.mark.parametrize("args", [["-x", "splice", "-a"], ["-x", "splice:hq", "-c", "--format-lib"], ["-x", "splice", "-a", "-u", "f", "--junc-bed", "BED"]])
def test_jump_annotation(args, tmp_path):
    """-j: alignment ends clipped next to an annotated junction hop over it when the clipped bases match the other side
    (mm_jump_split, jump.c; host-only post-processing, map.c:362-364).  The fixture has ~50 reads with 3-15 bases beyond an intron."""
    import synth
    ref, rd, bed = synth.make_junctions(str(tmp_path))
    args = [bed if a == "BED" else a for a in args] + ["-j", bed]
    want = subprocess.run([G.REF_BIN] + [a for a in args if a != "--format-lib"] + [ref, rd], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True).stdout
    got = subprocess.run([CHECK] + args + [ref, rd], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True).stdout
    assert G.strip_pg(want) == G.strip_pg(got)
    plain = subprocess.run([G.REF_BIN] + [a for a in args[:-2] if a != "--format-lib"] + [ref, rd], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True).stdout
    assert sum(1 for a, b in zip(G.strip_pg(want).split(b"\n"), G.strip_pg(plain).split(b"\n")) if a != b) > 20


@pytest.mark.skipif(not os.path.exists(G.REF_BIN), reason="needs the compiled reference")
def test_multi_part_index(tmp_path):  # -I: mm_gpu_init / mm_gpu_map_batch / mm_gpu_destroy once per index part (main.c:438-503)
    import synth
    ref, rd, _, _ = synth.make("ont", str(tmp_path), 3, 40, 23)
    for args in (["-x", "map-ont", "-a", "-I", "1200000"], ["-x", "map-ont", "-c", "-I", "1000000"]):
        out = _pair(args, ref, rd)
        assert out.count(b"\n") >= 40


@pytest.mark.skipif(not os.path.exists(G.REF_BIN), reason="needs the compiled reference")
@pytest.mark.parametrize("kind,args", [("weird", ["-x", "map-ont", "-a", "-T", "20"]), ("weird", ["-x", "map-ont", "-c", "-T", "5"]), ("weird", ["-x", "map-hifi", "-a", "-T", "30"]),
                                       ("pairs2", ["-x", "sr", "-a", "-T", "15"]), ("pairs1", ["-x", "sr", "-a", "-T", "10"]), ("weird", ["-x", "splice", "-a", "-T", "20"])])
def test_sdust_masking(kind, args, tmp_path):
    """-T: minimizers lying mostly in SDUST-masked regions are dropped before seeding (sdust.c, mm_dust_minier map.c:34-57) -- on reads
    with (AC)n / poly-A / tandem islands, and on read pairs, where the reference filters the second read with shifted positions."""
    import synth
    if kind == "weird":
        ref, rd = synth.make_weird(str(tmp_path))
        files = [rd]
    else:
        ref, f1, f2, inter = synth.make_pairs(str(tmp_path))
        files = [f1, f2] if kind == "pairs2" else [inter]
    outs = []
    for binary in (G.REF_BIN, CHECK):
        p = subprocess.run([binary] + args + [ref] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode == 0, p.stderr.decode()[-1500:]
        outs.append(G.strip_pg(p.stdout))
    assert outs[0] == outs[1]


def test_rmq_chainer_breaks_ties_like_the_reference():
    """minimap2_amd/csrc/rmq_chain.cpp against the reference's mg_lchain_rmq (lchain.c:250-368 over krmq.h) on anchor sets built to make
    range-minimum priorities tie, with constant eviction and size caps below the window (tests/cpucheck/rmq_test.cpp)."""
    exe = os.path.join(HERE, "_build", "rmq_test")
    if not os.path.exists(exe):
        pytest.skip("tests/_build/rmq_test not built (needs the compiled reference)")
    out = subprocess.run([exe, "3000"], stdout=subprocess.PIPE, check=True).stdout.split()
    assert out[0] == b"OK" and int(out[2]) > 500000


def test_host_hit_rules_against_the_reference_functions():
    """minimap2_amd/csrc/hits.cpp -- parents, secondaries, SAM primary, MAPQ, the filters, the rescoring, the hit sort, chains -> hits, the fragment
    split, the anchor squeeze, a fragment's secondaries, the pairing of two reads' hits -- against the reference's own functions (hit.c, pe.c, align.c's
    mm_update_dp_max) on random hit lists built to make the rules bite: equal scores and hashes, nested query intervals, dead hits, ALT hits, both reads'
    hits interleaved on the reference (tests/cpucheck/hits_test.cpp; it also checks that its cases reach the rules)."""
    exe = os.path.join(HERE, "_build", "hits_test")
    if not os.path.exists(exe):
        pytest.skip("tests/_build/hits_test not built (needs the compiled reference)")
    p = subprocess.run([exe, "3000"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()[-500:]
    assert p.stdout.split()[0] == b"OK"


def test_region_rules_against_the_reference_statics():
    """minimap2_amd/csrc/region_rules.hpp -- the two long-gap seed filters, the end trimming (what region_plan_kernel and align.cpp both call) -- plus
    chain_host.cpp's chain_cut and align.cpp's append_cigar, against the reference's own STATIC functions (mm_filter_bad_seeds, mm_filter_bad_seeds_alt,
    mm_fix_bad_ends, mm_append_cigar, mg_chain_bk_end: compiled where they lie by oracle/ref_align_shim.c and ref_lchain_shim.c) on 40 000 random chains
    built to reach the rules (tests/cpucheck/region_rules_test.cpp; it fails if a rule never took effect)."""
    exe = os.path.join(HERE, "_build", "region_rules_test")
    if not os.path.exists(exe):
        pytest.skip("tests/_build/region_rules_test not built (needs the compiled reference)")
    p = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()[-500:]
    assert b"== the reference's statics" in p.stdout


def test_local_score_of_the_inversion_test_against_the_reference():
    """minimap2_amd/csrc/ksw_ll.cpp (the anti-diagonal form of round 6, the lane-by-lane form behind it) against the reference's ksw_ll_i16: score and both end
    coordinates on 30 000 sequence pairs under random scorings (tests/cpucheck/ksw_ll_test.cpp)."""
    exe = os.path.join(HERE, "_build", "ksw_ll_test")
    if not os.path.exists(exe):
        pytest.skip("tests/_build/ksw_ll_test not built (needs the compiled reference)")
    p = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0 and p.stdout.startswith(b"OK"), p.stderr.decode()[-600:]


def test_device_sdust_header_against_the_reference():
    """minimap2_amd/csrc/sdust_core.hpp (what dust_filter_kernel runs per read) compiled for the host, vs the reference's sdust()."""
    exe = os.path.join(HERE, "_build", "sdust_test")
    if not os.path.exists(exe):
        pytest.skip("tests/_build/sdust_test not built (needs the compiled reference)")
    out = subprocess.run([exe, "300"], stdout=subprocess.PIPE, check=True).stdout.split()
    assert out[0] == b"OK" and int(out[2]) > 150


@pytest.mark.skipif(not os.path.exists(G.REF_BIN), reason="needs the compiled reference")
def test_pass1_junctions_mix_with_annotation(tmp_path):  # --pass1: MM_JUNC_MISC jumps with a score filter (main.c:478), annotated ones win (jump.c:90-95)
    import synth
    ref, rd, bed = synth.make_junctions(str(tmp_path), n_reads=40)
    lines = [l for l in open(bed).read().split("\n") if l]
    p1, anno = str(tmp_path / "pass1.bed"), str(tmp_path / "anno.bed")
    open(p1, "w").write("\n".join("\t".join(l.split("\t")[:4] + ["9" if i % 3 else "3"] + l.split("\t")[5:]) for i, l in enumerate(lines) if i % 2 == 0) + "\n")
    open(anno, "w").write("\n".join(l for i, l in enumerate(lines) if i % 2 == 1) + "\n")
    _pair(["-x", "splice", "-a", "--pass1", p1], ref, rd)
    _pair(["-x", "splice", "-a", "-j", anno, "--pass1", p1], ref, rd)


RNA_CASES = [(["-x", "splice:sr", "-a"], 2, False), (["-x", "splice:sr", "-a"], 1, False), (["-x", "splice:sr", "-a"], 2, True), (["-x", "splice:sr", "-c"], 2, True),
             (["-x", "splice:sr"], 2, False), (["-x", "splice:sr", "-a", "--format-lib"], 2, True), (["-x", "splice:sr", "-a", "-u", "f"], 2, False),
             (["-x", "splice:sr", "-a", "-b", "3"], 2, False)]  # -b: the flank-only job goes to ksw_exts2 without KSW_EZ_GENERIC_SC (align.c:393), the full window with it


@pytest.mark.skipif(not os.path.exists(G.REF_BIN), reason="needs the compiled reference")
@pytest.mark.parametrize("args,n_files,jump", RNA_CASES)
def test_short_rna_seq_pairs(args, n_files, jump, tmp_path):
    """-x splice:sr (MM_F_SR_RNA + MM_F_WEAK_PAIRING): mates mapped on their own and paired afterwards (map.c:382-387), the flank-only
    first attempt of mm_align_sr_rna (align.c:370-400), the ungapped shortcut, the one-strand rule (align.c:1072), optionally -j."""
    import synth
    ref, f1, f2, bed = synth.make_rna_pairs(str(tmp_path))
    args = args + (["-j", bed] if jump else [])
    files = [f1, f2] if n_files == 2 else [f1]
    want = subprocess.run([G.REF_BIN] + [a for a in args if a != "--format-lib"] + [ref] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True).stdout
    got = subprocess.run([CHECK] + args + [ref] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True).stdout
    assert G.strip_pg(want) == G.strip_pg(got)
    assert want.count(b"\n") > 150


@pytest.mark.skipif(not os.path.exists(G.REF_BIN), reason="needs the compiled reference")
@pytest.mark.parametrize("kind,args", [("ont", ["-x", "map-ont", "-c", "--qstrand"]), ("ont", ["-x", "map-ont", "--qstrand"]), ("ont", ["-x", "map-hifi", "-c", "--cs", "--qstrand", "--format-lib"]),
                                       ("weird", ["-x", "map-ont", "-c", "--qstrand"]), ("weird", ["-x", "asm20", "-c", "--qstrand", "--MD", "--format-lib"]),
                                       ("ont", ["-x", "map-ont", "-c", "--qstrand", "-P", "-z", "100,50"])])
def test_query_strand_mode(kind, args, tmp_path):
    """--qstrand: reverse-strand hits keep the query as given; anchors, DP targets and PAF coordinates are those of the
    reverse-complemented reference (map.c:192-196, align.c:780-786,894, format.c:343-346,440-443)."""
    import synth
    if kind == "ont":
        ref, rd, _, _ = synth.make("ont", str(tmp_path), 2, 60, 29)
    else:
        ref, rd = synth.make_weird(str(tmp_path))
    want = subprocess.run([G.REF_BIN] + [a for a in args if a != "--format-lib"] + [ref, rd], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True).stdout
    got = subprocess.run([CHECK] + args + [ref, rd], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True).stdout
    assert G.strip_pg(want) == G.strip_pg(got)
    assert sum(1 for l in want.split(b"\n") if b"\t-\t" in l) > 0  # reverse-strand hits are what the mode is about


@pytest.mark.skipif(not os.path.exists(G.REF_BIN), reason="needs the compiled reference")
@pytest.mark.parametrize("args", [["-x", "splice", "-a"], ["-x", "splice:hq", "-c", "--spsc-scale", "1.0", "--spsc0", "3", "-u", "f"], ["-x", "splice", "-a", "--junc-bed", "BED", "-j", "BED"]])
def test_splice_scores(args, tmp_path):
    """--spsc: every position of a splice DP window is priced by its score in the table or by junc_pen (mm_idx_spsc_get,
    index.c:1045-1066; ksw2_exts2_sse.c:196-200); the table takes precedence over --junc-bed (align.c:640-641)."""
    import synth
    ref, rd, bed = synth.make_junctions(str(tmp_path), n_reads=40)
    sp = synth.make_splice_scores(ref, str(tmp_path / "spsc.tsv"))
    args = [bed if a == "BED" else a for a in args] + ["--spsc", sp]
    out = _pair(args, ref, rd)
    plain = subprocess.run([G.REF_BIN] + args[:-2] + [ref, rd], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True).stdout
    assert sum(1 for a, b in zip(out.split(b"\n"), G.strip_pg(plain).split(b"\n")) if a != b) > 20
