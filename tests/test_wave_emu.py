"""The kernels' OWN source on the CPU: tests/_build/libmm2amd_emu.so is the product -- every .hip file of minimap2_amd/csrc included -- built
for the host under the wave emulator (tests/cpucheck/wave_emu: HIP threads as fibers that meet at cross-lane operations; only the gfx950
inline-assembly helpers are restated, wave_emu/ksw_pk_emu.hpp).  These cases run it end to end against the unmodified reference, so a kernel
that drifts from the reference is caught in a container without a GPU; the same cases and many more run on the hardware (`-m gpu`).
`MM2AMD_EMU=1 pytest -m gpu` runs the whole GPU suite on the emulator (tests/conftest.py)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as G  # noqa: E402
import reflib  # noqa: E402
import synth  # noqa: E402

EMU_SO = os.path.join(HERE, "_build", "libmm2amd_emu.so")
DROPIN_EMU = os.environ.get("MM2AMD_DROPIN_EMU") or os.path.join(HERE, "_build", "dropin_emu")  # (MM2AMD_DROPIN_EMU: e.g. tools/sanitize_emu.sh's AddressSanitizer build)


@pytest.fixture(scope="module")
def emu():
    if os.path.exists("/root/reference/minimap.h") or not os.path.exists(EMU_SO):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
        subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "cpucheck")], stdout=subprocess.DEVNULL)
    import minimap2_amd as mm
    saved = mm._lib
    mm._lib = mm._bind(C.CDLL(EMU_SO))
    yield mm
    mm._lib = saved


def _reads(seed, n, mean, err, ref_len=600000):
    rng = np.random.default_rng(seed)
    contigs = synth.gen_reference(rng, ref_len, 2)
    reads = synth.gen_reads(rng, contigs, n, mean, mean // 4, err)
    return [synth.ACGT[c].tobytes() for c in contigs], [("read%d" % i, synth.ACGT[r].tobytes()) for i, r in enumerate(reads)]


@pytest.mark.parametrize("preset,n,mean,err", [("map-ont", 10, 5000, 0.12), ("map-hifi", 6, 9000, 0.005), ("lr:hqae", 5, 6000, 0.02)])
def test_kernels_source_against_reference(emu, preset, n, mean, err):
    """sketch, seed collection, per-read LDS anchor sort, chaining with the LDS window (or the RMQ chainer), backtrack, the three DP kernels"""
    refs, rds = _reads(31, n, mean, err)
    al = emu.Aligner(refs, preset=preset, names=["chr1", "chr2"], n_threads=4)
    try:
        hits = al.map_batch(rds)
    finally:
        al.close()
    assert sum(1 for h in hits if h) >= n - 1
    if not os.path.exists(reflib.REF_SO):
        pytest.skip("oracle/_ref not built")
    assert [[a.key() for a in h] for h in hits] == reflib.ref_map_reads(refs, rds, preset)


def test_duplicated_anchor_keys_replayed(emu, tmp_path):
    """reads full of tandem repeats: equal anchor keys, the unstable radix sort's permutation replayed in LDS (anchor_sort_kernel)"""
    if not os.path.exists(G.REF_BIN) or not os.path.exists(DROPIN_EMU):
        pytest.skip("needs oracle/_ref and tests/_build/dropin_emu")
    ref, rd = synth.make_weird(str(tmp_path))
    want = subprocess.run([G.REF_BIN, "-x", "map-ont", "-t", "2", "-c", ref, rd], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True).stdout
    got = subprocess.run([DROPIN_EMU, "-x", "map-ont", "-t", "2", "-c", ref, rd], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True).stdout
    assert G.strip_pg(got) == G.strip_pg(want)


@pytest.mark.parametrize("case", ["mt_sam", "inv_paf", "x3s_paf"])
def test_reference_fixtures_through_the_kernels_source(case):
    if not os.path.exists(DROPIN_EMU):
        pytest.skip("tests/_build/dropin_emu needs the reference headers to build (dev container only)")
    got, err = G.run_fixture(DROPIN_EMU, case, ["--stats"])
    assert "backend=hip:gfx950" in err
    assert got == open(os.path.join(HERE, "golden", case + ".out"), "rb").read()


@pytest.mark.parametrize("case", ["mt_sam", "inv_paf", "x3s_paf"])
def test_regions_finished_on_the_device(case, emu, monkeypatch):
    """MM2AMD_DEVICE_FINISH=1: region_finish_kernel (windows' CIGARs stitched in LDS, mm_fix_cigar, the score walk as a 32-lane reduction) instead
    of the host's mm_update_extra: the reference's fixtures (a Z-drop split with an inversion, a spliced alignment) and the packed hit records,
    mm_extra_t::capacity included, of synthetic ONT reads"""
    if not os.path.exists(DROPIN_EMU):
        pytest.skip("needs tests/_build/dropin_emu")
    env = dict(os.environ, MM2AMD_DEVICE_FINISH="1")
    got, _ = G.run_fixture(DROPIN_EMU, case, env=env)
    assert got == open(os.path.join(G.HERE, case + ".out"), "rb").read()
    if case != "mt_sam" or not os.path.exists(reflib.REFDRV_SO):
        return
    from minimap2_amd import shard
    monkeypatch.setenv("MM2AMD_DEVICE_FINISH", "1")
    refs, rds = _reads(43, 24, 7000, 0.12, 600000)
    al = emu.Aligner(refs, preset="map-ont", names=["chr1", "chr2"], n_threads=4, sam=True)
    try:
        L = emu.lib()
        st = al.index_stat()
        S, keys, val_off, pos = reflib.export_index(al)
        drv = reflib.RefDriver(st["w"], st["k"], st["flag"], ["chr1", "chr2"], al.lens, S, keys, val_off, pos, 4)
        mo = drv.map_opt("map-ont", extra_flag=emu.F_OUT_SAM)
        _, nr, rg = drv.map(mo, rds, 4)
        want = shard.pack_hits(L, nr, rg).numpy().tobytes()
        L.mm2amd_free_regs(len(nr), nr, rg)
        drv.close()
        al.stage(rds)
        n_reg, reg, _ = al.run(raw=True)
        got = shard.pack_hits(L, n_reg, reg).numpy().tobytes()
        al.free_raw(n_reg, reg)
        emu.profile_enable(True)
        al.map_batch(rds[:4])
        prof = emu.profile_get()
        emu.profile_enable(False)
    finally:
        al.close()
    assert got == want
    assert "region_finish_kernel" in prof


def test_long_join_rechain_on_the_device(emu, monkeypatch):
    """map.c:283-292 on the device (the kernels' source under the emulator): reads with more than one chain have their chained anchors sorted
    by reference position again (the per-read LDS sort, replaying the reference's tie order), chained by chain_rmq_kernel with bw_long and
    backtracked again; a repeat-bearing reference makes every read take that path; the share is reported by mm2amd_last_stats; the same
    reads through the host's tie-exact tree (MM2AMD_LONG_JOIN_ON_HOST=1) give the same hits"""
    rng = np.random.default_rng(55)
    contig = synth.gen_duplicated_reference(rng)
    refs = [synth.ACGT[contig].tobytes(), synth.ACGT[rng.integers(0, 4, 200000, dtype=np.uint8)].tobytes()]
    reads = synth.gen_reads(rng, [contig], 30, 9000, 3000, 0.08)
    rds = [("rep%d" % i, synth.ACGT[r].tobytes()) for i, r in enumerate(reads)]
    al = emu.Aligner(refs, preset="map-ont", n_threads=4)
    try:
        got = [[a.key() for a in h] for h in al.map_batch(rds)]
        st = al.last_stats()
        monkeypatch.setenv("MM2AMD_LONG_JOIN_ON_HOST", "1")
        on_host = [[a.key() for a in h] for h in al.map_batch(rds)]
        st_host = al.last_stats()
    finally:
        al.close()
    assert st["n_long_join_dev"] >= 0.1 * len(rds) and st["n_long_join_host"] == 0
    assert st_host["n_long_join_dev"] == 0 and st_host["n_long_join_host"] == st["n_long_join_dev"]
    assert on_host == got
    if os.path.exists(reflib.REF_SO):
        assert got == reflib.ref_map_reads(refs, rds, "map-ont")


def test_two_bucket_partition_closed_form(emu, monkeypatch):
    """the replay of the reference's unstable anchor sort (ksort.h:101-151): where a partition has exactly two buckets (always so at the top:
    the strand bit) its result is computed by all threads from prefix counts instead of walked by one; reads with a tandem duplication have
    equal anchor keys on both strands' buckets -- hits equal to the walk's (MM2AMD_NO_TWO_BUCKET=1) and to the compiled reference's"""
    rng = np.random.default_rng(91)
    contigs = synth.gen_reference(rng, 400000, 2)
    reads = synth.gen_tandem_reads(rng, contigs, 24, 7000, 0.06)
    refs = [synth.ACGT[c].tobytes() for c in contigs]
    rds = [("tan%d" % i, synth.ACGT[r].tobytes()) for i, r in enumerate(reads)]
    al = emu.Aligner(refs, preset="map-ont", names=["chr1", "chr2"], n_threads=4)
    try:
        got = [[a.key() for a in h] for h in al.map_batch(rds)]
    finally:
        al.close()
    assert sum(1 for h in got if h) >= len(rds) - 1
    if os.path.exists(reflib.REF_SO):
        assert got == reflib.ref_map_reads(refs, rds, "map-ont")
    if os.path.exists(DROPIN_EMU) and os.path.exists(G.REF_BIN):  # the anchors themselves, in order (--print-seeds, map.c:255-260), walk against closed form
        import tempfile
        with tempfile.TemporaryDirectory() as tmp:
            ref_fa, rd_fa = os.path.join(tmp, "ref.fa"), os.path.join(tmp, "reads.fa")
            open(ref_fa, "w").write("".join(">chr%d\n%s\n" % (i + 1, r.decode()) for i, r in enumerate(refs)))
            open(rd_fa, "w").write("".join(">%s\n%s\n" % (n, r.decode()) for n, r in rds))
            outs = []
            for env in ({}, {"MM2AMD_NO_TWO_BUCKET": "1"}):
                dump = os.path.join(tmp, "seeds%d.txt" % len(outs))
                subprocess.run([DROPIN_EMU, "-x", "map-ont", "-t", "2", "-c", ref_fa, rd_fa], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True,
                               env=dict(os.environ, MM2AMD_SEED_DUMP=dump, **env))
                outs.append(open(dump).read())
            assert outs[0] == outs[1] and outs[0].count("SD\t") > 1000
            want = subprocess.run([G.REF_BIN, "-x", "map-ont", "-t", "1", "-c", "--print-seeds", ref_fa, rd_fa], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, check=True).stderr.decode()
            want_sd = [l for l in want.split("\n") if l.startswith("SD\t")]
            got_sd = [l for l in outs[0].split("\n") if l.startswith("SD\t")]
            assert got_sd == want_sd


def test_partitions_walked_over_tapes(tmp_path):
    """anchor_sort_ties_kernel: the reads with equal anchor keys replayed together after the sorting launch, a partition into more than two
    buckets as a walk over byte-sized tapes (one LDS round trip per element by one thread, everything else by the workgroup): the anchors in
    order equal the reference's --print-seeds, and the one-thread walk's and the in-launch replay's"""
    if not os.path.exists(DROPIN_EMU) or not os.path.exists(G.REF_BIN):
        pytest.skip("needs oracle/_ref and tests/_build/dropin_emu")
    import tie_cases
    want, got = tie_cases.many_bucket_tie_case(DROPIN_EMU, G.REF_BIN, str(tmp_path), 93, 8, 8000)
    assert len(want) > 20000
    assert got["tapes"][1].count("tape walk") >= 16 and "at shift 32" in got["tapes"][1] and "at shift 8" in got["tapes"][1]
    assert "tape walk" not in got["walk"][1] and "tape walk" not in got["inline"][1]
    for name in got:
        assert got[name][0] == want, name


def test_equal_keys_in_reads_beyond_the_lds_classes(tmp_path):
    """reads of ~30 kb: more than 10240 anchors, sorted on global scratch by the comparison network and, having equal keys, replayed there by their own
    workgroup (tie_exact_replay over the split key / index arrays: the one-thread walk, the wave-wide bucket scan, children side by side) -- the path
    tools/coverage_emu.sh showed no other case took; anchors in order == the reference's --print-seeds"""
    if not os.path.exists(DROPIN_EMU) or not os.path.exists(G.REF_BIN):
        pytest.skip("needs oracle/_ref and tests/_build/dropin_emu")
    import tie_cases
    want, got = tie_cases.many_bucket_tie_case(DROPIN_EMU, G.REF_BIN, str(tmp_path), 99, 3, 30000, modes=("tapes",))
    assert len(want) > 3 * 10240
    assert got["tapes"][0] == want


def _dropin_case(tmp_path, name, contigs, reads, runs):
    """SAM of tests/_build/dropin_emu == the compiled reference's, per (preset, environment)"""
    d = tmp_path / name
    d.mkdir()
    ref_fa, rd_fa = str(d / "ref.fa"), str(d / "reads.fa")
    open(ref_fa, "w").write("".join(">chr%d\n%s\n" % (i + 1, synth.ACGT[c].tobytes().decode()) for i, c in enumerate(contigs)))
    open(rd_fa, "w").write("".join(">%s%d\n%s\n" % (name, i, synth.ACGT[r].tobytes().decode()) for i, r in enumerate(reads)))
    wants = {}
    for preset, env in runs:
        if preset not in wants:
            wants[preset] = G.strip_pg(subprocess.run([G.REF_BIN, "-x", preset, "-t", "2", "-a", ref_fa, rd_fa], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True).stdout)
        p = subprocess.run([DROPIN_EMU, "-x", preset, "-t", "2", "-a", ref_fa, rd_fa], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
        assert p.returncode == 0, p.stderr.decode()[-800:]
        assert G.strip_pg(p.stdout) == wants[preset], (name, preset, env)
    return wants


def test_structural_variants_through_the_region_kernels(tmp_path):
    """what real long reads have and a single-base error model does not: 30-160 base insertions / deletions alone and in clusters (the device twins of
    mm_filter_bad_seeds / mm_filter_bad_seeds_alt in region_plan_kernel), gaps beyond the chaining gap and beyond the long-join bandwidth (two hits on one
    strand, each extension bounded by the other's seeds), reads at the ends of a sequence, error-free reads; and partial inverted copies (the strand rule's
    hand-back in chain_regs_kernel).  tools/coverage_emu.sh showed these lines of region_dev.hip unreached by everything else.  SAM == the reference's on the
    device path and on the host's"""
    if not os.path.exists(DROPIN_EMU) or not os.path.exists(G.REF_BIN):
        pytest.skip("needs oracle/_ref and tests/_build/dropin_emu")
    rng = np.random.default_rng(61)
    contigs = synth.gen_reference(rng, 800000, 2)
    reads = synth.gen_sv_reads(rng, contigs[0], 2)
    w = _dropin_case(tmp_path, "sv", contigs, reads, [("map-ont", {}), ("map-ont", {"MM2AMD_DEVICE_REGIONS": "0"}), ("map-pb", {})])
    assert sum(1 for l in w["map-ont"].split(b"\n") if l and not l.startswith(b"@")) > len(reads)  # supplementary alignments: the far-apart hits
    contigs, reads = synth.gen_inverted_copy_case(np.random.default_rng(62), 6)
    _dropin_case(tmp_path, "inv", contigs, reads, [("map-ont", {}), ("map-hifi", {})])


def test_short_reads_and_pairs_through_the_region_kernels(tmp_path):
    """round 6: `-x sr` on the device region path, the kernels' own source under the wave emulator -- a pair's chains cut per segment in chain_regs_kernel (mm_seg_gen,
    mm_select_sub_multi), the best diagonal run, the ungapped window and its Z-drop walk in region_plan_kernel, the literal piece in region_finish_kernel.  Pairs (two
    files), single-end reads, a small Z-drop (the ungapped window's walk trips it: the fragment goes to the host's rounds) and the host's stages instead of the device's:
    SAM == the compiled reference's every time"""
    if not os.path.exists(DROPIN_EMU) or not os.path.exists(G.REF_BIN):
        pytest.skip("needs oracle/_ref and tests/_build/dropin_emu")
    ref, f1, f2, _ = synth.make_pairs(str(tmp_path), seed=131, n_pairs=500, genome=300000)
    ref2, rd = synth.make_short(str(tmp_path / "se") if (tmp_path / "se").mkdir() is None else "", seed=132, n_reads=400, genome=300000)
    for args, files, env in ((["-x", "sr", "-a"], [ref, f1, f2], {}), (["-x", "sr", "-a"], [ref, f1, f2], {"MM2AMD_DEVICE_REGIONS": "0"}), (["-x", "sr", "-a", "-z", "5"], [ref, f1, f2], {}),
                             (["-x", "sr", "-c"], [ref2, rd], {}), (["-x", "sr", "-a", "-f", "2,10"], [ref, f1, f2], {})):
        want = G.strip_pg(subprocess.run([G.REF_BIN, "-t", "2"] + args + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True).stdout)
        p = subprocess.run([DROPIN_EMU, "-t", "2"] + args + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
        assert p.returncode == 0, p.stderr.decode()[-800:]
        assert G.strip_pg(p.stdout) == want, (args, env)


def test_pipeline_equals_batch_by_batch(emu):
    """hand-over of batch k+1 beside the mapping of batch k (mm_gpu_batch_stage_queued) and the output stage into the reused buffer
    (mm_gpu_format_batch_view) give the text of stage + run + format, batch by batch"""
    refs, rds = _reads(37, 12, 3000, 0.1, 300000)
    al = emu.Aligner(refs, preset="map-ont", names=["chr1", "chr2"], n_threads=4, sam=True)
    try:
        base = emu.Batch(rds)
        batches = [base.rotated(k) for k in (0, 5, 9, 2)] + [emu.Batch(rds[:3]), emu.Batch([])]
        want = []
        for b in batches:
            al.stage(b)
            n_reg, reg, rep = al.run(raw=True)
            want.append(al.format_raw(n_reg, reg, rep))
            al.free_raw(n_reg, reg)
        got = []
        total = al.pipeline(batches, on_text=lambda b, addr, ln: got.append(C.string_at(addr, ln)))
        assert got == want and total == sum(len(t) for t in want)
        assert len(want[0]) > 10000 and want[1] != want[0]
        # a batch handed over and then replaced is never mapped; mapping the staged batch twice gives the same records
        al.stage(batches[1])
        al.stage(batches[4])
        a = al.run(raw=True)
        ta = al.format_raw(*a)
        al.free_raw(a[0], a[1])
        assert ta == want[4]
    finally:
        al.close()


def test_pipeline_early_start_equals_batch_by_batch(emu, monkeypatch):
    """Round 4: lanes that run out of sub-batches of the batch being mapped start on the batch the pipeline has handed over next, before that
    batch's own mm_gpu_map_staged call (Mapper::stage(may_start_early)).  Batches cut into sub-batches of three reads, so that every batch has
    a tail: the text of every batch equals stage + run + format of the same batch, in several passes, early starts did happen, and the same
    with MM2AMD_NO_EARLY_START-style un-pipelined calls in between (a re-run of the current batch, a replaced hand-over)."""
    monkeypatch.setenv("MM2AMD_SUBBATCH_READS", "3")
    refs, rds = _reads(41, 14, 2500, 0.1, 300000)
    al = emu.Aligner(refs, preset="map-ont", names=["chr1", "chr2"], n_threads=4, sam=True)
    try:
        base = emu.Batch(rds)
        batches = [base.rotated(k) for k in (0, 3, 7)] + [emu.Batch(rds[:2]), emu.Batch([]), base.rotated(11), emu.Batch(rds[4:9])]
        want = []
        for b in batches:
            al.stage(b)
            n_reg, reg, rep = al.run(raw=True)
            want.append(al.format_raw(n_reg, reg, rep))
            al.free_raw(n_reg, reg)
            assert al.last_stats()["n_early_sub"] == 0  # (an un-queued hand-over never starts early)
        n_early = 0
        for _ in range(3):
            got = []
            def on_text(b, addr, ln):
                got.append(C.string_at(addr, ln))
            def on_mapped(b, n_reg, reg, rep_len):
                nonlocal n_early
                n_early += int(al.last_stats()["n_early_sub"])
            total = al.pipeline(batches, on_text=on_text, on_mapped=on_mapped)
            assert got == want and total == sum(len(t) for t in want)
            # between two pipelines: the un-pipelined calls still work on a context whose lanes have been running ahead
            al.stage(batches[1])
            al.stage(batches[3])
            a = al.run(raw=True)
            assert al.format_raw(*a) == want[3]
            al.free_raw(a[0], a[1])
        assert n_early > 0, "no sub-batch was started before its batch's mapping call"
    finally:
        al.close()


@pytest.mark.parametrize("preset,err", [("map-ont", 0.1), ("map-hifi", 0.02)])
def test_device_finish_on_repeat_rich_reads(preset, err, tmp_path):
    """region_finish_kernel (round 4: one wave per region, mm_fix_cigar's left alignment as a prefix scan over x -> min(m, L + x), empty operations
    and equal neighbours removed by two compactions) on reads whose indels sit in tandem repeats and homopolymers -- shifts through whole
    matches, chains of shifts, leading gaps: SAM == the compiled reference, == the host's mm_update_extra"""
    if not os.path.exists(G.REF_BIN) or not os.path.exists(DROPIN_EMU):
        pytest.skip("needs oracle/_ref and tests/_build/dropin_emu")
    ref, rd = synth.make_tandem_reads(str(tmp_path), seed=91 if preset == "map-ont" else 92, n_reads=16, mean=3000, err=err, genome=200000)
    want = subprocess.run([G.REF_BIN, "-x", preset, "-t", "2", "-a", ref, rd], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True).stdout
    outs = []
    for fin in ("1", "0"):
        outs.append(subprocess.run([DROPIN_EMU, "-x", preset, "-t", "2", "-a", ref, rd], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True,
                                   env=dict(os.environ, MM2AMD_DEVICE_FINISH=fin, MM2AMD_FIN_CHECK="1")).stdout)
    assert G.strip_pg(outs[0]) == G.strip_pg(want)
    assert G.strip_pg(outs[1]) == G.strip_pg(want)
    assert want.count(b"\n") > 16


def test_update_extra_kernel_equals_the_reference(emu):
    """region_finish_kernel's source through the kernel-level C ABI against the reference's own static mm_update_extra (oracle/ref_align_shim.c) on
    adversarial CIGARs: tests/test_gpu_update_extra.py's generator, the CPU suite's share"""
    import test_gpu_update_extra as T
    if not os.path.exists(reflib.REFALIGN_SO):
        pytest.skip("oracle/_ref/librefalign.so not built")
    rng = np.random.default_rng(5)
    mat = reflib.ts_mat(2, 4)
    jobs = [T.random_region(rng, int(rng.choice([1, 2, 3, 8, 40, 150])), int(rng.choice([1, 2, 4])), float(rng.choice([0.0, 0.03, 0.1])), float(rng.choice([0.0, 0.02, 0.1]))) for _ in range(150)]
    jobs += [(b"", b"", []), (b"\0\0\0\1", b"\0\1", [[2 << 4 | 1], [2 << 4]])]
    for log_gap in (1, 0):
        got = emu.update_extra_batch(jobs, mat, 4, 2, log_gap)
        for i, (qs, ts, pieces) in enumerate(jobs):
            assert got[i] == reflib.ref_update_extra(qs, ts, pieces, mat, 4, 2, log_gap), i


def test_device_sort_and_prefix_sum_kernels(emu):
    """the index build's hand-written device-wide radix sort and prefix sum (device_sort.hip) against numpy; tests/test_gpu_device_sort.py's
    checks at sizes around the tile and the chunk of tiles, the CPU suite's share"""
    import test_gpu_device_sort as T
    rng = np.random.default_rng(2)
    for n, bits in ((0, 30), (1, 30), (4097, 30), (4096 * 3 + 11, 64), (128 * 4096 + 1, 30), (129 * 4096 + 7, 13)):
        keys = rng.integers(0, 1 << 63, n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, n, dtype=np.uint64)
        vals = rng.integers(0, 1 << 62, n, dtype=np.uint64)
        k, v = emu.sort_pairs_u64(keys, vals, bits)
        wk, wv = T.want_sorted(keys, vals, bits)
        assert np.array_equal(k, wk) and np.array_equal(v, wv), (n, bits)
    dup = rng.integers(0, 5, 70000, dtype=np.uint64) << np.uint64(9)
    k, v = emu.sort_pairs_u64(dup, np.arange(dup.size, dtype=np.uint64), 30)
    assert np.array_equal(v, np.argsort(dup, kind="stable").astype(np.uint64))
    for n in (0, 1, 4096, 4097, 4096 * 4096 + 5):
        a = rng.integers(0, 1 << 32 if n < 10000 else 7, n, dtype=np.uint64).astype(np.uint32)
        want = np.concatenate([[0], np.cumsum(a.astype(np.uint64))]).astype(np.uint64) & np.uint64(0xFFFFFFFF)
        assert np.array_equal(emu.exclusive_sum_u32(a).astype(np.uint64), want), n


def test_cigar_pool_overflow_flag_and_retry(tmp_path):
    """the DP kernels' CIGAR-pool overflow flag and ksw_host.cpp's retry with the worst-case pool (tests/test_gpu_dropin.py's case, the CPU suite's share):
    a first pool of 1/300 of the worst case overflows on any input"""
    if not os.path.exists(G.REF_BIN) or not os.path.exists(DROPIN_EMU):
        pytest.skip("needs oracle/_ref and tests/_build/dropin_emu")
    ref, rd, _, _ = synth.make("ont", str(tmp_path), 1, 12, 9)
    want = subprocess.run([G.REF_BIN, "-x", "map-ont", "-t", "2", "-c", ref, rd], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True).stdout
    got = subprocess.run([DROPIN_EMU, "-x", "map-ont", "-t", "2", "-c", ref, rd], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True, env=dict(os.environ, MM2AMD_CIGAR_POOL_DIV="300")).stdout
    assert G.strip_pg(got) == G.strip_pg(want)


def test_chaining_in_pieces_equals_whole_reads(tmp_path):
    """round 6: a read with very many anchors is chained by several wavefronts, each from one cluster head to another (seed_chain.hip: chain_piece_bounds; the
    host lists the pieces: backend_hip.cpp make_pieces), and chain_rmq_kernel runs with a small neighbourhood buffer first and a second launch for what did not fit.
    The kernels' own source under the wave emulator on a reference of diverged copies (every read crosses several: many clusters, long-join re-chaining):
    mg_lchain_dp + the long-join RMQ (map-ont) and mg_lchain_rmq as the primary chainer (asm20), pieces of 7 / 5 and 64 / 32 anchors, never cut, the tiny first buffer, the
    RMQ chainer's long clusters on workgroups of 4 and 16 wavefronts (chain_rmq_wide_kernel) -- PAF with CIGARs identical every time, and identical to the compiled reference's"""
    if not os.path.exists(DROPIN_EMU) or not os.path.exists(G.REF_BIN):
        pytest.skip("needs oracle/_ref and tests/_build/dropin_emu")
    rng = np.random.default_rng(77)
    contig = synth.gen_duplicated_reference(rng)
    other = rng.integers(0, 4, 200000, dtype=np.uint8)
    reads = synth.gen_reads(rng, [contig], 12, 9000, 3000, 0.08) + synth.gen_reads(rng, [other], 3, 5000, 1000, 0.05)
    ref, rd = str(tmp_path / "ref.fa"), str(tmp_path / "reads.fa")
    synth.write_fasta(ref, ["dup", "plain"], [contig, other])
    synth.write_fasta(rd, ["r%d" % i for i in range(len(reads))], reads)
    for preset in ("map-ont", "asm20"):
        want = subprocess.run([G.REF_BIN, "-x", preset, "-t", "2", "-c", ref, rd], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True).stdout
        assert want.count(b"\n") >= len(reads)
        for env in ({"MM2AMD_CHAIN_PIECE": "0", "MM2AMD_RMQ_PIECE": "0"}, {"MM2AMD_CHAIN_PIECE": "7", "MM2AMD_RMQ_PIECE": "5", "MM2AMD_RMQ_DENSE": "0"},
                    {"MM2AMD_CHAIN_PIECE": "64", "MM2AMD_RMQ_PIECE": "32", "MM2AMD_RMQ_NEAR_TINY": "1", "MM2AMD_RMQ_DENSE": "0"}, {"MM2AMD_RMQ_NEAR_TINY": "1"},
                    # the RMQ kernel's long clusters by workgroups of 4 / 16 wavefronts (chain_rmq_wide_kernel): pieces of 40 anchors or more, of 8 or more
                    {"MM2AMD_CHAIN_PIECE": "64", "MM2AMD_RMQ_PIECE": "32", "MM2AMD_RMQ_DENSE": "40"}, {"MM2AMD_CHAIN_PIECE": "64", "MM2AMD_RMQ_PIECE": "16", "MM2AMD_RMQ_DENSE": "8", "MM2AMD_RMQ_NEAR_TINY": "1"}):
            p = subprocess.run([DROPIN_EMU, "-x", preset, "-t", "2", "-c", ref, rd], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, MM2AMD_PIECE_DEBUG="1", **env))
            assert p.returncode == 0, p.stderr.decode()[-800:]
            assert p.stdout == want, (preset, env)
            if env.get("MM2AMD_RMQ_DENSE", "0") != "0":
                assert b"chain_rmq_wide_kernel<1024>" in p.stderr and (env["MM2AMD_RMQ_DENSE"] != "40" or b"chain_rmq_wide_kernel<256>" in p.stderr), (preset, env)
            if "MM2AMD_CHAIN_PIECE" in env:  # (the default piece lengths cut some of these reads' re-chains too)
                assert (b"chaining work list" in p.stderr) == (env["MM2AMD_CHAIN_PIECE"] != "0"), (preset, env)
