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
    json.dump(meta, open(os.path.join(HERE, "inputs.json"), "w"), indent=1, sort_keys=True)
