"""End-to-end parity on the GPU: the reference's own I/O + SAM writer around OUR mapper (tests/_build/dropin_gpu, linked
against libmm2amd.so) must produce byte-identical SAM/PAF to oracle/_ref/minimap2_ref on the same inputs (only the @PG
header line, which embeds argv, is excluded)."""
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
EMU = os.environ.get("MM2AMD_EMU") == "1"
DROPIN = os.path.join(HERE, "_build", "dropin_emu" if os.environ.get("MM2AMD_EMU") == "1" else "dropin_gpu")  # MM2AMD_EMU=1: tests/conftest.py


def _run(cmd):
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, (cmd, p.stderr.decode()[-2000:])
    return b"\n".join(l for l in p.stdout.split(b"\n") if not l.startswith(b"@PG")), p.stderr.decode()


def _compare(tmp_path, kind, preset, ref_mb, n_reads, seed, extra=()):
    ref, reads, _, _ = synth.make(kind, str(tmp_path), ref_mb, n_reads, seed)
    want, _ = _run([REF_BIN, "-x", preset, "-t", "8"] + list(extra) + [ref, reads])
    got, err = _run([DROPIN, "-x", preset, "-t", "8", "--stats"] + list(extra) + [ref, reads])
    assert "backend=hip:gfx950" in err, err[-500:]
    if want != got:
        wl, gl = want.split(b"\n"), got.split(b"\n")
        bad = [i for i in range(min(len(wl), len(gl))) if wl[i] != gl[i]]
        raise AssertionError("%d/%d lines differ; first: %r" % (len(bad), len(wl), wl[bad[0]][:300] if bad else None))
    return len(want)


def test_prebuilt_binaries_present():
    assert os.path.exists(REF_BIN), "oracle/_ref/minimap2_ref must be built in the dev container (make -C oracle ref)"
    assert os.path.exists(DROPIN), "tests/_build/dropin_gpu must be built in the dev container (make -C tests/cpucheck)"


sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as G  # noqa: E402


@pytest.mark.parametrize("case", list(G.FIXTURE_CASES))
def test_reference_fixtures_through_hip(case):
    """the reference's own test inputs (vendored: tests/golden/ref_fixtures/) through the HIP path: MT-human x MT-orang -a = BASELINE.json
    configs[0] (one record, pos 577, MAPQ 60), t-inv x q-inv -c (Z-drop split + inversion rescue, align.c:916-971), x3s -x splice
    (cg:Z:69M134N65M, test/x3s-aln.txt:1); == committed golden == the compiled reference"""
    want = open(os.path.join(HERE, "golden", case + ".out"), "rb").read()
    got, err = G.run_fixture(DROPIN, case, ["--stats"])
    assert "backend=hip:gfx950" in err, err[-500:]
    assert got == want
    ref, _ = G.run_fixture(REF_BIN, case)
    assert ref == want
    if case == "mt_sam":
        assert b"\t577\t60\t" in got
    if case == "inv_paf":
        assert got.count(b"tp:A:I") == 2
    if case == "x3s_paf":
        assert b"cg:Z:69M134N65M" in got


@pytest.mark.parametrize("case", ["mt_sam", "inv_sam", "x3s_sam"])
def test_regions_finished_on_the_device(case, tmp_path):
    """MM2AMD_DEVICE_FINISH=1: region_finish_kernel instead of the host's mm_update_extra / mm_fix_cigar (align.c:105-181, :254-303): the
    reference's fixtures, and for the first one a batch of synthetic ONT reads against the compiled reference"""
    want = open(os.path.join(HERE, "golden", case + ".out"), "rb").read()
    env = dict(os.environ, MM2AMD_DEVICE_FINISH="1")
    got, err = G.run_fixture(DROPIN, case, ["--stats"], env=env)
    assert "backend=hip:gfx950" in err, err[-500:]
    assert got == want
    if case == "mt_sam":
        ref, reads, _, _ = synth.make("ont", str(tmp_path), 1.0, 60, 211)
        ours = subprocess.run([DROPIN, "-x", "map-ont", "-t", "4", "-a", ref, reads], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True, env=env).stdout
        theirs = subprocess.run([REF_BIN, "-x", "map-ont", "-t", "4", "-a", ref, reads], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True).stdout
        assert G.strip_pg(ours) == G.strip_pg(theirs)


@pytest.mark.parametrize("preset,err", [("map-ont", 0.1), ("map-hifi", 0.02), ("lr:hq", 0.05)])
def test_device_finish_on_repeat_rich_reads(preset, err, tmp_path):
    """region_finish_kernel (round 4: one wave per region, mm_fix_cigar's left alignment as a prefix scan, two compactions) on reads whose indels
    sit in tandem repeats and homopolymers: SAM == the compiled reference with the regions finished on the device and on the host"""
    ref, reads = synth.make_tandem_reads(str(tmp_path), seed=93, n_reads=12 if EMU else 400, mean=3000 if EMU else 6000, err=err, genome=200000 if EMU else 2000000)
    theirs = subprocess.run([REF_BIN, "-x", preset, "-t", "8", "-a", ref, reads], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True).stdout
    for fin in ("1", "0"):
        env = dict(os.environ, MM2AMD_DEVICE_FINISH=fin)
        ours = subprocess.run([DROPIN, "-x", preset, "-t", "8", "-a", ref, reads], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True, env=env).stdout
        assert G.strip_pg(ours) == G.strip_pg(theirs), "MM2AMD_DEVICE_FINISH=%s" % fin
    assert theirs.count(b"\n") > (12 if EMU else 400)


def test_one_by_one_calls_on_gpu(tmp_path):
    """mm_gpu_map / mm_gpu_map_frag (the reference's mm_map / mm_map_frag signatures, map.c:380-397): a batch of one per call through
    the HIP path == one mm_gpu_map_batch == the reference"""
    ref, reads, _, _ = synth.make("ont", str(tmp_path), 1, 12, 71)
    want, _ = _run([REF_BIN, "-x", "map-ont", "-t", "4", "-a", ref, reads])
    got, err = _run([DROPIN, "-x", "map-ont", "-t", "4", "-a", "--one-by-one", "--stats", ref, reads])
    assert "backend=hip:gfx950" in err
    assert got == want
    got, _ = G.run_fixture(DROPIN, "inv_paf", ["--one-by-one"])
    assert got == open(os.path.join(HERE, "golden", "inv_paf.out"), "rb").read()


def test_batch_call_that_names_its_index_and_options(tmp_path):
    """mm_gpu_map_batch_with(mi, opt, ...) (SURVEY.md 8(b)(2) as written; include/mm2amd.h): no mm_gpu_init, the context is built by the
    first batch and reused by the following ones == the reference, single reads and read pairs"""
    ref, reads, _, _ = synth.make("ont", str(tmp_path), 1, 40, 72)
    want, _ = _run([REF_BIN, "-x", "map-ont", "-t", "4", "-a", "-K", "100000", ref, reads])
    got, err = _run([DROPIN, "-x", "map-ont", "-t", "4", "-a", "-K", "100000", "--batch-with", "--stats", ref, reads])
    assert err.count("backend=hip:gfx950") >= 2  # several mini-batches through the one context
    assert got == want
    got, _ = G.run_fixture(DROPIN, "inv_paf", ["--batch-with"])
    assert got == open(os.path.join(HERE, "golden", "inv_paf.out"), "rb").read()


def test_partitions_walked_over_tapes(tmp_path):
    """anchor_sort_ties_kernel (reads with equal anchor keys replayed together; partitions into many buckets walked over tapes): the anchors in
    order == the reference's --print-seeds == the one-thread walk's == the in-launch replay's, reads of the 4 k, 7 k and 10 k classes"""
    import tie_cases
    for k, (seed, n_reads, mean) in enumerate(((93, 16, 8000), (94, 10, 16000))):
        d = tmp_path / str(k)
        d.mkdir()
        want, got = tie_cases.many_bucket_tie_case(DROPIN, REF_BIN, str(d), seed, n_reads, mean)
        assert len(want) > 40000
        if EMU:
            assert got["tapes"][1].count("tape walk") >= 16
        for name in got:
            assert got[name][0] == want, (seed, name)


def test_anchor_sort_classes_and_chain_fill_variants(tmp_path):
    """anchor_sort_kernel's launch classes (256 / 512 / 1024 threads in LDS, 1024 threads on global scratch: MM2AMD_SORT_MIN_CLASS pushes small
    reads through the large ones) with and without duplicated keys, and chain_fill_kernel with its LDS window against the all-global variant"""
    ref, reads, _, _ = synth.make("ont", str(tmp_path / "o"), 4, 150, 19)
    ref_w, reads_w = synth.make_weird(str(tmp_path / "w"))
    for r, q, extra in ((ref, reads, ["-x", "map-ont", "-a"]), (ref_w, reads_w, ["-x", "map-ont", "-c"])):
        want, _ = _run([REF_BIN, "-t", "8"] + extra + [r, q])
        for env in ({"MM2AMD_SORT_MIN_CLASS": "2"}, {"MM2AMD_SORT_MIN_CLASS": "4"}, {"MM2AMD_SORT_MIN_CLASS": "5"}, {"MM2AMD_CHAIN_FILL_GLOBAL": "1"}):
            p = subprocess.run([DROPIN, "-t", "8"] + extra + [r, q], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
            assert p.returncode == 0, p.stderr.decode()[-1000:]
            assert b"\n".join(l for l in p.stdout.split(b"\n") if not l.startswith(b"@PG")) == want, env


def test_ont_sam_identical(tmp_path):
    assert _compare(tmp_path, "ont", "map-ont", 8, 400, 11, ["-a"]) > 1000


def test_ont_paf_cigar_identical(tmp_path):
    _compare(tmp_path, "ont", "map-ont", 4, 150, 12, ["-c"])


def test_hifi_sam_identical(tmp_path):
    _compare(tmp_path, "hifi", "map-hifi", 8, 150, 13, ["-a"])


def test_ont_second_batch_and_tiny_reads(tmp_path):
    # -K forces several mini-batches through the same device context
    _compare(tmp_path, "ont", "map-ont", 2, 120, 14, ["-a", "-K", "300000"])


def test_single_affine_scoring_sam_identical(tmp_path):
    # -O4 -E2 with equal second gap cost: mm_align_pair takes ksw_extz2_sse (align.c:353-354)
    _compare(tmp_path, "ont", "map-ont", 3, 120, 15, ["-a", "-O", "4", "-E", "2"])
    _compare(tmp_path, "hifi", "map-hifi", 3, 60, 16, ["-c", "-O", "6,6", "-E", "2,2"])


def test_single_anchor_chains_identical(tmp_path):
    # -n 1 with a small -m: every anchor may be its own chain (lchain.c:66), so a read has up to as many chains as anchors
    _compare(tmp_path, "ont", "map-ont", 2, 60, 41, ["-a", "-n", "1", "-m", "10"])
    _compare(tmp_path, "ont", "map-ont", 2, 60, 42, ["-n", "1", "-m", "10", "-s", "20"])


def test_splice_sam_identical(tmp_path):
    # -x splice: chaining with is_cdna, both transcript strands aligned with the splice-aware DP (ksw_exts2), N in CIGARs, ts:A tags
    assert _compare(tmp_path, "cdna", "splice", 6, 500, 17, ["-a"]) > 500
    _compare(tmp_path, "cdna", "splice:hq", 4, 300, 18, ["-c", "--cs"])


def test_splice_variants_identical(tmp_path):
    # one strand only / no signal matching / old splice model / shorter maximum intron
    _compare(tmp_path, "cdna", "splice", 3, 200, 19, ["-a", "-u", "f"])
    _compare(tmp_path, "cdna", "splice", 3, 200, 19, ["-c", "-u", "n", "-J", "0"])
    _compare(tmp_path, "cdna", "splice", 3, 200, 19, ["-a", "-G", "10000", "-C", "5", "--splice-flank=no"])


def test_hpc_index_sam_identical(tmp_path):
    # -x map-pb: homopolymer-compressed minimizers (sketch.c:95-101), index built by the reference and adopted by mm_gpu_init
    _compare(tmp_path, "hifi", "map-pb", 3, 60, 20, ["-a"])
    _compare(tmp_path, "ont", "map-pb", 3, 60, 21, ["-c"])


def test_output_stage_on_gpu_results(tmp_path):
    # the records written by mm_gpu_format_batch from the GPU path's hits == the reference binary's output
    ref, reads, _, _ = synth.make("ont", str(tmp_path), 4, 200, 23)
    for extra in (["-a"], ["-c", "--cs"]):
        want, _ = _run([REF_BIN, "-x", "map-ont", "-t", "8"] + extra + [ref, reads])
        got, _ = _run([DROPIN, "--format-lib", "-x", "map-ont", "-t", "8"] + extra + [ref, reads])
        assert want == got


def test_rmq_presets_identical(tmp_path):
    # asm20 / lr:hqae: RMQ chaining on the host over the device-sorted anchors; 100-300 kb queries with a deletion and an inversion
    import numpy as np
    rng = np.random.default_rng(41)
    contigs = synth.gen_reference(rng, 3000000, 2)
    reads = synth.gen_reads(rng, contigs, 5, 250000, 30000, 0.02, min_len=100000)
    s = contigs[0][200000:600000].copy()
    s = np.concatenate([s[:100000], s[105000:250000], synth.COMP[s[250000:253000][::-1]], s[253000:]])
    reads.append(synth.mutate_read(rng, s, 0.01))
    ref, rd = str(tmp_path / "ref.fa"), str(tmp_path / "contigs.fa")
    synth.write_fasta(ref, ["c1", "c2"], contigs)
    synth.write_fasta(rd, ["q%d" % i for i in range(len(reads))], reads)
    for preset, extra in (("asm20", ["-c"]), ("lr:hqae", ["-a"]), ("asm5", ["-c", "--cs"])):
        want, _ = _run([REF_BIN, "-x", preset, "-t", "8"] + extra + [ref, rd])
        got, err = _run([DROPIN, "-x", preset, "-t", "8", "--stats"] + extra + [ref, rd])
        assert "backend=hip:gfx950" in err
        assert want == got, preset
    ref2, reads2, _, _ = synth.make("hifi", str(tmp_path), 3, 80, 24)
    want, _ = _run([REF_BIN, "-x", "lr:hqae", "-t", "8", "-a", ref2, reads2])
    got, _ = _run([DROPIN, "-x", "lr:hqae", "-t", "8", "-a", ref2, reads2])
    assert want == got


def test_rmq_chain_kernel_against_host_chainer_and_reference(tmp_path):
    # chain_rmq_kernel (seed_chain.hip) = rmq_chain.cpp on the host (MM2AMD_RMQ_ON_HOST=1) = the reference: accurate and noisy reads (the noisy
    # ones walk the close neighbourhood for most anchors), reads full of repeats (equal priorities: handed back to the host), a small size cap
    ref_o, reads_o, _, _ = synth.make("ont", str(tmp_path / "o"), 3, 150, 61)
    ref_h, reads_h, _, _ = synth.make("hifi", str(tmp_path / "h"), 3, 100, 62)
    ref_w, reads_w = synth.make_weird(str(tmp_path / "w"))
    for ref, reads, extra in ((ref_o, reads_o, ["-x", "lr:hqae", "-a"]), (ref_o, reads_o, ["-x", "asm20", "-c"]), (ref_h, reads_h, ["-x", "lr:hqae", "-c", "--cs"]),
                              (ref_h, reads_h, ["-x", "asm5", "-a", "--cap-kalloc=0"]), (ref_w, reads_w, ["-x", "lr:hqae", "-a"]), (ref_w, reads_w, ["-x", "asm20", "-c"])):
        extra = [e for e in extra if not e.startswith("--cap")]
        want, _ = _run([REF_BIN, "-t", "8"] + extra + [ref, reads])
        got, _ = _run([DROPIN, "-t", "8"] + extra + [ref, reads])
        env = dict(os.environ, MM2AMD_RMQ_ON_HOST="1")
        p = subprocess.run([DROPIN, "-t", "8"] + extra + [ref, reads], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        assert p.returncode == 0
        host = b"\n".join(l for l in p.stdout.split(b"\n") if not l.startswith(b"@PG"))
        assert host == want, ("host chainer", extra)
        assert got == want, ("device chainer", extra)


def test_alt_contigs_identical(tmp_path):
    ref, rd, alt = synth.make_alt(str(tmp_path))
    for extra in (["-a", "--alt", alt], ["-c", "--alt", alt]):
        want, _ = _run([REF_BIN, "-x", "map-ont", "-t", "8"] + extra + [ref, rd])
        got, _ = _run([DROPIN, "-x", "map-ont", "-t", "8"] + extra + [ref, rd])
        assert want == got


def test_rechain_with_raised_occurrence_cap_identical(tmp_path):
    ref, rd = synth.make_repeats(str(tmp_path))
    for extra in (["-c", "-f", "3,50", "-e", "0"], ["-a", "-f", "3,50"]):
        want, _ = _run([REF_BIN, "-x", "map-ont", "-t", "8"] + extra + [ref, rd])
        got, _ = _run([DROPIN, "-x", "map-ont", "-t", "8"] + extra + [ref, rd])
        assert want == got


def test_wide_minimizer_windows_identical(tmp_path):
    """-w beyond 32: the window automaton with the 256-slot ring (sketch_kernel<256> per read, idx_sketch_kernel<.., 256, ..> for the index) -- plain and
    homopolymer-compressed; no preset has such a window, and the emulator's line coverage showed that no case had one"""
    ref, rd = synth.make_weird(str(tmp_path))
    for extra in (["-c", "-k", "17", "-w", "40"], ["-c", "-H", "-k", "15", "-w", "50"], ["-a", "-k", "21", "-w", "33"]):
        want, _ = _run([REF_BIN, "-x", "map-ont", "-t", "8"] + extra + [ref, rd])
        got, _ = _run([DROPIN, "-x", "map-ont", "-t", "8"] + extra + [ref, rd])
        assert want == got, extra


def test_self_complementary_kmers_identical(tmp_path):
    """even k on (AT)n / (ACGT)n / (AATT)n islands: k-mers that equal their reverse complement take no window slot (sketch.c:108), the sketch
    kernels' lanes that start inside an island have to reach back beyond their fixed warm-up stretch (sketch_wave_kernel's base_at fallback; the
    emulator's line coverage showed that no case reached it)"""
    ref, rd = synth.make_palindromes(str(tmp_path))
    for extra in (["-c", "-k", "16", "-w", "10"], ["-a", "-k", "14", "-w", "5"], ["-c", "-k", "20", "-w", "19"], ["-c", "-H", "-k", "16", "-w", "10"], ["-c"]):
        want, _ = _run([REF_BIN, "-x", "map-ont", "-t", "8"] + extra + [ref, rd])
        got, _ = _run([DROPIN, "-x", "map-ont", "-t", "8"] + extra + [ref, rd])
        assert want == got, extra
    # tiles of 256 bases (sketch_wave_kernel cuts reads longer than 16 kb into tiles; MM2AMD_SKETCH_TILE forces small ones): tile borders inside the islands,
    # where the warm-up needs bases further back than the 128 a tile keeps resident before its start
    want, _ = _run([REF_BIN, "-x", "map-ont", "-t", "8", "-c", "-k", "16", "-w", "10", ref, rd])
    p = subprocess.run([DROPIN, "-x", "map-ont", "-t", "8", "-c", "-k", "16", "-w", "10", ref, rd], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, MM2AMD_SKETCH_TILE="256"))
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    assert p.stdout == want


@pytest.mark.parametrize("kind,preset,n", [("ont", "map-ont", 40), ("cdna", "splice", 60), ("hifi", "map-hifi", 20)])
def test_cigar_pool_overflow_retry_identical(kind, preset, n, tmp_path):
    """The DP kernels write their CIGARs into one pool sized at a quarter of the worst case; a kernel whose CIGAR does not fit sets a flag instead of
    writing (ksw_stream / ksw_gapfill / ksw_ext / ksw_extd2 / ksw_splice: `cigar_pool_cap`), and the batch's DP runs again with the worst-case pool
    (ksw_host.cpp).  No ordinary input overflows a quarter, so no case reached the flag or the retry: MM2AMD_CIGAR_POOL_DIV makes the first pool 1/300."""
    ref, rd, _, _ = synth.make(kind, str(tmp_path), 2, n, 9)
    want, _ = _run([REF_BIN, "-x", preset, "-t", "8", "-c", ref, rd])
    p = subprocess.run([DROPIN, "-x", preset, "-t", "8", "-c", ref, rd], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, MM2AMD_CIGAR_POOL_DIV="300"))
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    assert p.stdout == want


def test_edge_case_reads_identical(tmp_path):
    # tiny reads around k, all-N, IUPAC, lower case, low-complexity islands, a homopolymer, a whole-contig read, a chimera ...
    ref, rd = synth.make_weird(str(tmp_path))
    for preset, extra in (("map-ont", ["-a"]), ("map-hifi", ["-c", "--cs"]), ("asm20", ["-c"]), ("map-pb", ["-a"]), ("splice", ["-a"])):
        want, _ = _run([REF_BIN, "-x", preset, "-t", "8"] + extra + [ref, rd])
        got, _ = _run([DROPIN, "-x", preset, "-t", "8"] + extra + [ref, rd])
        assert want == got, preset


def test_chain_level_mapping_identical(tmp_path):
    # PAF without CIGAR: seeding and chaining only
    _compare(tmp_path, "ont", "map-ont", 4, 300, 25, [])
    _compare(tmp_path, "cdna", "splice", 3, 200, 26, [])
    _compare(tmp_path, "hifi", "asm20", 3, 40, 27, [])


def test_one_strand_only_identical(tmp_path):
    # --for-only / --rev-only: seed hits on the other strand are skipped before chaining (skip_seed, map.c:91-97)
    _compare(tmp_path, "ont", "map-ont", 3, 150, 28, ["-a", "--for-only"])
    _compare(tmp_path, "hifi", "map-hifi", 3, 60, 29, ["-c", "--rev-only"])


def test_jump_annotation_identical(tmp_path):
    # -j: clipped alignment ends hop over annotated junctions (mm_jump_split; host post-processing after the GPU stages)
    ref, rd, bed = synth.make_junctions(str(tmp_path))
    for extra in (["-a"], ["-c", "-u", "f"]):
        want, _ = _run([REF_BIN, "-x", "splice", "-t", "8", "-j", bed] + extra + [ref, rd])
        got, _ = _run([DROPIN, "-x", "splice", "-t", "8", "-j", bed] + extra + [ref, rd])
        assert want == got


def test_two_replicas_on_one_device_identical(tmp_path):
    # mm_gpu_init_multi with the same ordinal twice: two backends (own streams, buffers, index copy) on one GPU map the two halves of
    # every mini-batch concurrently -- the in-process multi-GPU path as far as a single-GPU box can run it
    ref, reads, _, _ = synth.make("ont", str(tmp_path), 4, 120, 51)
    want, _ = _run([REF_BIN, "-x", "map-ont", "-a", "-t", "8", ref, reads])
    env = dict(os.environ, MM2AMD_GPUS="2", MM2AMD_DEVICE_IDS="0,0")
    p = subprocess.run([DROPIN, "-x", "map-ont", "-a", "-t", "8", "--stats", ref, reads], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    got = b"\n".join(l for l in p.stdout.split(b"\n") if not l.startswith(b"@PG"))
    assert "replicas=2" in p.stderr.decode(), p.stderr.decode()[-500:]
    assert got == want


def test_three_step_pipeline_driver_identical(tmp_path):
    # tests/dropin/dropin_pipeline.c on the GPU library: parse / map / format of consecutive mini-batches overlap (kt_pipeline)
    ref, reads, _, _ = synth.make("ont", str(tmp_path), 4, 150, 53)
    want, _ = _run([REF_BIN, "-x", "map-ont", "-a", "-t", "8", ref, reads])
    got, err = _run([os.path.join(HERE, "_build", "dropin_pipeline_emu" if os.environ.get("MM2AMD_EMU") == "1" else "dropin_pipeline_gpu"), "-x", "map-ont", "-a", "-t", "8", "-K", "300k", ref, reads])
    assert err.count("[M::worker_pipeline::") >= 3
    assert got == want


def test_failed_batch_falls_back_to_the_reference_path(tmp_path):
    """SURVEY.md 8(b) "error conventions": a mini-batch the dispatcher fails on leaves its outputs untouched, and the hook maps it with the reference's
    own per-read path (tests/dropin/dropin_pipeline.c: kt_for over mm_map_frag).  MM2AMD_INJECT_BATCH_FAILURE=2 makes the
    second mapping call of the process report a failure after its batch was mapped: the stream must still equal the reference's, and the
    batches after it must map on the GPU again."""
    ref, reads, _, _ = synth.make("ont", str(tmp_path), 2, 60, 21)
    pipe = os.path.join(HERE, "_build", "dropin_pipeline_emu" if EMU else "dropin_pipeline_gpu")
    want, _ = _run([REF_BIN, "-x", "map-ont", "-t", "4", "-a", ref, reads])
    p = subprocess.run([pipe, "-x", "map-ont", "-t", "4", "-a", "-K", "100000", ref, reads], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=dict(os.environ, MM2AMD_INJECT_BATCH_FAILURE="2"))
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    assert p.stderr.decode().count("mapped by the reference's own path") == 1
    assert b"\n".join(l for l in p.stdout.split(b"\n") if not l.startswith(b"@PG")) == want
