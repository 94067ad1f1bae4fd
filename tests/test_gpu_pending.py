"""Device features written after the round's GPU budget was spent: validated against the reference through the host pipeline
(tests/test_host_pipeline.py, oracle-backed) but not yet on an MI355X.  They are opt-in in the library (MM2AMD_PENDING=1) and so
are these tests; once they have passed on hardware the gate goes away and the cases move into test_gpu_dropin.py."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import synth  # noqa: E402

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.environ.get("MM2AMD_PENDING"), reason="opt-in: MM2AMD_PENDING=1")]
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "minimap2_ref")
DROPIN = os.path.join(HERE, "_build", "dropin_gpu")


def _run(cmd):
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, (cmd, p.stderr.decode()[-2000:])
    return b"\n".join(l for l in p.stdout.split(b"\n") if not l.startswith(b"@PG"))


@pytest.mark.parametrize("args", [["-x", "ava-ont"], ["-x", "ava-pb"], ["-x", "ava-ont", "-c"], ["-x", "map-ont", "-D", "-c"],
                                  ["-x", "map-ont", "--dual=no"], ["-X", "-a"]])
def test_all_vs_all(args, tmp_path):  # skip_seed's read-name rules (map.c:81-91), MM_SEED_SELF (align.c:760-767)
    fa = synth.make_overlaps(str(tmp_path))
    assert _run([REF_BIN, "-t", "8"] + args + [fa, fa]) == _run([DROPIN, "-t", "8"] + args + [fa, fa])
