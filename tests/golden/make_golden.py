"""Regenerates the golden SAM/PAF fixtures: output of the UNMODIFIED reference (oracle/_ref/minimap2_ref, built from
/root/reference by oracle/Makefile) on seeded synthetic inputs from tests/synth.py.  Run in the dev container:

    python tests/golden/make_golden.py

Each case stores the reference's stdout without the @PG line (it embeds argv) plus the md5 of the generated inputs,
so a test can tell "inputs drifted" from "results differ"."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth  # noqa: E402

REF_BIN = os.path.join(ROOT, "oracle", "_ref", "minimap2_ref")

# name -> (kind, preset, ref_mb, n_reads, seed, extra args)
CASES = {
    "ont_sam": ("ont", "map-ont", 1.0, 40, 101, ["-a"]),
    "ont_paf": ("ont", "map-ont", 0.5, 25, 102, ["-c"]),
    "hifi_sam": ("hifi", "map-hifi", 1.0, 20, 103, ["-a"]),
    "lrhq_paf_cs": ("hifi", "lr:hq", 0.5, 15, 104, ["-c", "--cs"]),
    "cdna_sam": ("cdna", "splice", 1.0, 60, 107, ["-a"]),
}


# The reference's OWN test inputs (/root/reference/test, vendored unchanged under tests/golden/ref_fixtures/ because the GPU box has no
# /root/reference): name -> (target, query, args).  mt_sam is BASELINE.json configs[0] (one record, pos 577, MAPQ 60);
# inv_paf exercises the Z-drop split and the inversion rescue (align.c:916-971: tp:A:I records); x3s_paf the spliced alignment the
# reference documents in test/x3s-aln.txt:1 (cg:Z:69M134N65M).
FIXDIR = os.path.join(HERE, "ref_fixtures")
FIXTURE_CASES = {
    "mt_sam": ("MT-human.fa", "MT-orang.fa", ["-a"]),
    "mt_paf_cs": ("MT-human.fa", "MT-orang.fa", ["-c", "--cs"]),
    "inv_paf": ("t-inv.fa", "q-inv.fa", ["-c"]),
    "inv_sam": ("t-inv.fa", "q-inv.fa", ["-a"]),
    "x3s_paf": ("x3s-ref.fa", "x3s-qry.fa", ["-x", "splice", "-c"]),
    "x3s_sam": ("x3s-ref.fa", "x3s-qry.fa", ["-x", "splice", "-a"]),
    "tiny_paf": ("t2.fa", "q2.fa", ["-c"]),
}


def run_fixture(binary, case, extra=(), env=None):
    t, q, args = FIXTURE_CASES[case]
    p = subprocess.run([binary] + list(args) + list(extra) + ["-t", "2", os.path.join(FIXDIR, t), os.path.join(FIXDIR, q)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    return strip_pg(p.stdout), p.stderr.decode()


def md5(path):
    return hashlib.md5(open(path, "rb").read()).hexdigest()


def strip_pg(b):
    return b"\n".join(l for l in b.split(b"\n") if not l.startswith(b"@PG"))


def run_case(binary, case, tmp):
    kind, preset, ref_mb, n_reads, seed, extra = CASES[case]
    ref, reads, _, _ = synth.make(kind, tmp, ref_mb, n_reads, seed)
    p = subprocess.run([binary, "-x", preset, "-t", "4"] + extra + [ref, reads], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    return strip_pg(p.stdout), {"ref_md5": md5(ref), "reads_md5": md5(reads)}


if __name__ == "__main__":
    meta = {}
    for case in CASES:
        with tempfile.TemporaryDirectory() as tmp:
            out, m = run_case(REF_BIN, case, tmp)
        open(os.path.join(HERE, case + ".out"), "wb").write(out)
        meta[case] = m
        print(case, len(out.split(b"\n")), "lines")
    for case in FIXTURE_CASES:
        out, _ = run_fixture(REF_BIN, case)
        open(os.path.join(HERE, case + ".out"), "wb").write(out)
        print(case, len(out.split(b"\n")), "lines")
    json.dump(meta, open(os.path.join(HERE, "inputs.json"), "w"), indent=1, sort_keys=True)
