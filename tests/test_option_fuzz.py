"""Option fuzzing of the drop-in against the reference binary (CPU, oracle-backed check backend): random subsets of minimap2's
mapping and output options with random values -- chaining thresholds, scoring, Z-drop, occurrence filters, hit selection,
output flavours -- on small ONT / HiFi / cDNA / ALT / repeat inputs.  The seeds are fixed, so this is a regression test; a
wider sweep of the same generator (hundreds of seeds) was run during development."""
import os
import random
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as G  # noqa: E402
import synth  # noqa: E402

CHECK = os.path.join(HERE, "_build", "dropin_check")
needs_dev = pytest.mark.skipif(not (os.path.exists(G.REF_BIN) and os.path.exists(CHECK)), reason="needs oracle/_ref and tests/_build (dev container)")

C = random.Random.choice
OPTS = [("-g", lambda r: C(r, ["500", "2000", "5000", "10000"])), ("-r", lambda r: C(r, ["100", "500,2000", "2000,20000", "50,50"])),
        ("-N", lambda r: C(r, ["0", "1", "5", "20"])), ("-p", lambda r: C(r, ["0.5", "0.8", "0.95", "0.2"])), ("-M", lambda r: C(r, ["0.2", "0.5", "0.9"])),
        ("-n", lambda r: C(r, ["1", "2", "3", "6"])), ("-m", lambda r: C(r, ["20", "40", "100"])), ("-s", lambda r: C(r, ["40", "80", "200", "1000"])),
        ("-z", lambda r: C(r, ["100", "400,200", "50,30", "1000,100"])), ("-A", lambda r: C(r, ["1", "2", "3"])), ("-B", lambda r: C(r, ["2", "4", "6"])),
        ("-O", lambda r: C(r, ["4,24", "6,26", "3", "5,12"])), ("-E", lambda r: C(r, ["2,1", "3,1", "2", "1,0"])), ("-b", lambda r: C(r, ["0", "2", "3"])),
        ("-P", None), ("-e", lambda r: C(r, ["0", "100", "500", "2000"])), ("--max-chain-skip", lambda r: C(r, ["5", "25", "100"])),
        ("--max-chain-iter", lambda r: C(r, ["50", "500", "5000"])), ("--min-dp-len", lambda r: C(r, ["50", "200", "500"])), ("--no-long-join", None),
        ("--end-bonus", lambda r: C(r, ["0", "5", "20", "100"])), ("--score-N", lambda r: C(r, ["0", "1", "3"])), ("--no-end-flt", None),
        ("--hard-mask-level", None), ("--max-qlen", lambda r: C(r, ["5000", "9000", "0"])), ("--chain-gap-scale", lambda r: C(r, ["0.5", "1.0", "2.0"])),
        ("--chain-skip-scale", lambda r: C(r, ["0.0", "0.5", "1.5"])), ("--mask-len", lambda r: C(r, ["100", "1000", "100000"])),
        ("--q-occ-frac", lambda r: C(r, ["0", "0.01", "0.1"])), ("--no-hash-name", None), ("-U", lambda r: C(r, ["10,100", "2,5", "50,500"])),
        ("-f", lambda r: C(r, ["0.0002", "0.01", "20", "3,50"])), ("--eqx", None), ("-Y", None), ("--secondary=no", None),
        ("-k", lambda r: C(r, ["13", "15", "17", "21"])), ("-w", lambda r: C(r, ["5", "10", "19", "40"]))]
FORMAT_OPTS = [("--MD", None), ("--cs", None), ("--ds", None), ("--cs=long", None), ("-L", None), ("-y", None), ("--secondary-seq", None), ("--paf-no-hit", None),
               ("--sam-hit-only", None)]


@pytest.fixture(scope="module")
def inputs(tmp_path_factory):
    d = tmp_path_factory.mktemp("fuzz")
    out = {}
    for kind, preset, mb, n, seed in (("ont", "map-ont", 1.0, 25, 201), ("hifi", "map-hifi", 1.0, 15, 202), ("cdna", "splice", 1.0, 40, 203)):
        ref, rd, _, _ = synth.make(kind, str(d / kind), mb, n, seed)
        out[kind] = (ref, rd, preset, [])
    ref, rd, alt = synth.make_alt(str(d / "alt"))
    out["alt"] = (ref, rd, "map-ont", ["--alt", alt])
    ref, rd = synth.make_repeats(str(d / "rep"))
    out["rep"] = (ref, rd, "map-ont", [])
    return out


def _case(inputs, seed, table, format_lib):
    r = random.Random(seed)
    ref, rd, preset, extra = inputs[["ont", "hifi", "cdna", "alt", "rep"][seed % 5]]
    args = ["-x", preset] + extra
    mode = r.choice(["-a", "-c", "-a", "-c", ""])  # "": chain-level mapping, PAF without CIGAR
    if mode:
        args.append(mode)
    for name, gen in r.sample(table, r.randint(1, 6)):
        args.append(name)
        if gen:
            args.append(gen(r))
    outs = []
    for binary, pre in ((G.REF_BIN, []), (CHECK, ["--format-lib"] if format_lib else [])):
        p = subprocess.run([binary] + pre + args + ["-t", "4", ref, rd], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        outs.append((p.returncode, G.strip_pg(p.stdout)))
    if outs[0][0] != 0:  # an option combination mm_check_opt rejects: we must reject it too
        assert outs[1][0] != 0, args
    else:
        assert outs[1][0] == 0 and outs[0][1] == outs[1][1], args


@needs_dev
@pytest.mark.parametrize("seed", range(1000, 1016))
def test_mapping_options(inputs, seed):
    _case(inputs, seed, OPTS, False)


@needs_dev
@pytest.mark.parametrize("seed", range(2000, 2010))
def test_output_options_through_the_library_formatter(inputs, seed):
    _case(inputs, seed, OPTS + FORMAT_OPTS * 3, True)


SR_OPTS = [("-F", lambda r: C(r, ["200", "400", "800", "2000"])), ("--heap-sort=no", None), ("--heap-sort=yes", None), ("-f", lambda r: C(r, ["2,20", "5,50", "0.001", "3"])),
           ("-g", lambda r: C(r, ["50", "100", "300"])), ("-r", lambda r: C(r, ["50", "100", "30,30"])), ("-s", lambda r: C(r, ["20", "40", "80"])),
           ("-n", lambda r: C(r, ["1", "2", "3"])), ("-m", lambda r: C(r, ["15", "25", "40"])), ("-k", lambda r: C(r, ["15", "19", "21"])), ("-w", lambda r: C(r, ["5", "8", "11"])),
           ("--no-pairing", None)]
SR_SKIP = ("-P", "-g", "-r", "-s", "-n", "-m", "-k", "-w", "-f", "--max-qlen")


@pytest.fixture(scope="module")
def short_inputs(tmp_path_factory):
    d = tmp_path_factory.mktemp("fuzz_sr")
    ref, rd = synth.make_short(str(d / "se"), n_reads=150)
    pref, f1, f2, inter = synth.make_pairs(str(d / "pe"), n_pairs=100)
    ovl = synth.make_overlaps(str(d / "ovl"), n_reads=40)
    return {"se": (ref, [rd]), "pe2": (pref, [f1, f2]), "pe1": (pref, [inter]), "ovl": (ovl, [ovl])}


@needs_dev
@pytest.mark.parametrize("seed", range(3000, 3016))
def test_short_read_pair_and_overlap_options(short_inputs, seed):
    """The same generator over the short-read (single-end, two files, interleaved) and all-vs-all inputs; every other case goes
    through the library's formatter (mate fields)."""
    r = random.Random(seed)
    kind = ["se", "pe2", "pe1", "ovl"][seed % 4]
    ref, files = short_inputs[kind]
    args = ["-x", r.choice(["ava-ont", "ava-pb"]) if kind == "ovl" else "sr"]
    mode = r.choice(["-a", "-c", "-a", ""])
    if mode:
        args.append(mode)
    table = [o for o in OPTS if o[0] != "--max-qlen"] if kind == "ovl" else SR_OPTS + [o for o in OPTS + FORMAT_OPTS if o[0] not in SR_SKIP]
    for name, gen in r.sample(table, r.randint(1, 5)):
        args.append(name)
        if gen:
            args.append(gen(r))
    outs = []
    for binary, pre in ((G.REF_BIN, []), (CHECK, ["--format-lib"] if seed % 2 else [])):
        p = subprocess.run([binary] + pre + args + ["-t", "4", ref] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        outs.append((p.returncode, G.strip_pg(p.stdout)))
    if outs[0][0] != 0:
        assert outs[1][0] != 0, args
    else:
        assert outs[1][0] == 0 and outs[0][1] == outs[1][1], args


@pytest.fixture(scope="module")
def annotated_inputs(tmp_path_factory):
    d = tmp_path_factory.mktemp("fuzz_anno")
    return {"rna": synth.make_rna_pairs(str(d / "rna"), n_tx=30), "junc": synth.make_junctions(str(d / "junc"), n_reads=30), "weird": synth.make_weird(str(d / "weird"))}


@needs_dev
@pytest.mark.parametrize("seed", range(4000, 4012))
def test_rna_seq_annotation_and_masking_options(annotated_inputs, seed):
    """splice:sr pairs (with / without -j), long cDNA reads with --junc-bed / -j / --junc-bonus, and SDUST masking (-T), each with
    random further options."""
    r = random.Random(seed)
    kind = seed % 3
    if kind == 0:
        ref, f1, f2, bed = annotated_inputs["rna"]
        files = r.choice([[f1, f2], [f1]])
        args = ["-x", "splice:sr"] + (["-j", bed] if r.random() < 0.5 else [])
        skip = ("--max-qlen", "-P", "-g", "-r", "-k", "-w", "--eqx")
    elif kind == 1:
        ref, rd, bed = annotated_inputs["junc"]
        files = [rd]
        args = ["-x", r.choice(["splice", "splice:hq"])] + r.choice([["--junc-bed", bed], ["-j", bed], ["--junc-bed", bed, "-j", bed, "--junc-bonus", "15"]])
        skip = ("--max-qlen", "-P", "--eqx")
    else:
        ref, rd = annotated_inputs["weird"]
        files = [rd]
        args = ["-x", r.choice(["map-ont", "map-hifi", "asm20"]), "-T", r.choice(["5", "10", "20", "40"])]
        skip = ("--max-qlen", "-P")
    mode = r.choice(["-a", "-c", "-a", ""])
    if mode:
        args.append(mode)
    for name, gen in r.sample([o for o in OPTS if o[0] not in skip], r.randint(0, 4)):
        args.append(name)
        if gen:
            args.append(gen(r))
    outs = []
    for binary, pre in ((G.REF_BIN, []), (CHECK, ["--format-lib"] if seed % 2 else [])):
        p = subprocess.run([binary] + pre + args + ["-t", "4", ref] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        outs.append((p.returncode, G.strip_pg(p.stdout)))
    if outs[0][0] != 0:
        assert outs[1][0] != 0, args
    else:
        assert outs[1][0] == 0 and outs[0][1] == outs[1][1], args
