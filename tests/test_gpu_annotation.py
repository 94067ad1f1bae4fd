"""Mapping options beyond the default presets on the device path: junction annotation and splice scores (the lane-exact kernel's
admit()), SDUST masking (host scan + dust_filter_kernel), short RNA-seq pairs (composed byte targets) and query-strand mode.  All
cases passed on an MI355X in round 2 (profiles/r02_gpu_pending_v2.log); the opt-in gate of round 1 is gone."""
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


@pytest.mark.parametrize("args", [["-x", "splice", "-a"], ["-x", "splice", "-c"], ["-x", "splice", "-a", "--junc-bonus", "20"], ["-x", "splice:hq", "-a"],
                                  ["-x", "splice", "-a", "-u", "n"], ["-x", "splice", "-c", "--cs", "-u", "f"]])
def test_junction_annotation(args, tmp_path):  # --junc-bed: annotation bonus in the lane-exact kernel's donor / acceptor costs (ksw_extd2.hip, admit())
    ref, rd, bed = synth.make_junctions(str(tmp_path))
    args = args + ["--junc-bed", bed]
    assert _run([REF_BIN, "-t", "8"] + args + [ref, rd]) == _run([DROPIN, "-t", "8"] + args + [ref, rd])


def test_junction_annotation_larger_set(tmp_path):
    ref, rd, bed = synth.make_junctions(str(tmp_path), seed=37, n_reads=600, ref_mb=4.0)
    args = ["-x", "splice", "-a", "--junc-bed", bed]
    assert _run([REF_BIN, "-t", "8"] + args + [ref, rd]) == _run([DROPIN, "-t", "8"] + args + [ref, rd])


@pytest.mark.parametrize("kind,args", [("weird", ["-x", "map-ont", "-a", "-T", "20"]), ("weird", ["-x", "map-ont", "-c", "-T", "5"]), ("weird", ["-x", "map-hifi", "-a", "-T", "30"]),
                                       ("pairs2", ["-x", "sr", "-a", "-T", "15"]), ("pairs1", ["-x", "sr", "-a", "-T", "10"]), ("weird", ["-x", "splice", "-a", "-T", "20"])])
def test_sdust_masking(kind, args, tmp_path):  # -T: dust_filter_kernel (sdust_core.hpp), one thread per read
    if kind == "weird":
        ref, rd = synth.make_weird(str(tmp_path))
        files = [rd]
    else:
        ref, f1, f2, inter = synth.make_pairs(str(tmp_path))
        files = [f1, f2] if kind == "pairs2" else [inter]
    assert _run([REF_BIN, "-t", "8"] + args + [ref] + files) == _run([DROPIN, "-t", "8"] + args + [ref] + files)


@pytest.mark.parametrize("args,n_files,jump", [(["-x", "splice:sr", "-a"], 2, False), (["-x", "splice:sr", "-a"], 1, False), (["-x", "splice:sr", "-a"], 2, True),
                                               (["-x", "splice:sr", "-c"], 2, True), (["-x", "splice:sr"], 2, False), (["-x", "splice:sr", "-a", "-b", "3"], 2, False)])
def test_short_rna_seq_pairs(args, n_files, jump, tmp_path):  # splice:sr: flank-only DP jobs with composed byte targets (KswScoring::tbytes), weak pairing
    ref, f1, f2, bed = synth.make_rna_pairs(str(tmp_path))
    args = args + (["-j", bed] if jump else [])
    files = [f1, f2] if n_files == 2 else [f1]
    assert _run([REF_BIN, "-t", "8"] + args + [ref] + files) == _run([DROPIN, "-t", "8"] + args + [ref] + files)


@pytest.mark.parametrize("kind,args", [("ont", ["-x", "map-ont", "-c", "--qstrand"]), ("ont", ["-x", "map-ont", "--qstrand"]), ("ont", ["-x", "map-hifi", "-c", "--cs", "--qstrand"]),
                                       ("weird", ["-x", "map-ont", "-c", "--qstrand"])])
def test_query_strand_mode(kind, args, tmp_path):  # --qstrand: flipped reference coordinates in seed_expand_kernel, reverse-complemented DP targets through the byte pool
    if kind == "ont":
        ref, rd, _, _ = synth.make("ont", str(tmp_path), 2, 60, 29)
    else:
        ref, rd = synth.make_weird(str(tmp_path))
    assert _run([REF_BIN, "-t", "8"] + args + [ref, rd]) == _run([DROPIN, "-t", "8"] + args + [ref, rd])


@pytest.mark.parametrize("args", [["-x", "splice", "-a"], ["-x", "splice", "-c", "--spsc-scale", "1.0"], ["-x", "splice:hq", "-a", "--spsc0", "3"], ["-x", "splice", "-a", "-u", "f"]])
def test_splice_scores(args, tmp_path):  # --spsc: per-position score lookups in the lane-exact kernel's admit()
    ref, rd, bed = synth.make_junctions(str(tmp_path), n_reads=40)
    sp = synth.make_splice_scores(ref, str(tmp_path / "spsc.tsv"))
    args = args + ["--spsc", sp]
    assert _run([REF_BIN, "-t", "8"] + args + [ref, rd]) == _run([DROPIN, "-t", "8"] + args + [ref, rd])
