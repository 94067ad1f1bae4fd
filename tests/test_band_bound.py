"""The measurement behind DESIGN.md section 8b's "fewer gap-fill cells" (verdict item 5b), kept reproducible: for the gap-fill windows of ONT-like reads,
which cells of a window's q x t rectangle can no alignment as good as a narrow-band one pass through, and does the reference's own ksw_extd2 give the same
score and CIGAR inside the symmetric band that holds the rest?  The counting lives in the oracle-backed check backend (tests/cpucheck/backend_check.cpp,
MM2AMD_CHECK_BAND_STATS): test infrastructure, no product code involved."""
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import synth  # noqa: E402

DROPIN_CHECK = os.path.join(HERE, "_build", "dropin_check")


def test_gap_fill_cells_outside_the_provable_band(tmp_path):
    if not os.path.exists(DROPIN_CHECK):
        pytest.skip("tests/_build/dropin_check needs the reference headers to build (dev container only)")
    ref, reads, _, _ = synth.make("ont", str(tmp_path), 1, 24, 77)
    out = str(tmp_path / "band.tsv")
    p = subprocess.run([DROPIN_CHECK, "-x", "map-ont", "-a", "-t", "4", ref, reads], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
                       env=dict(os.environ, MM2AMD_CHECK_BAND_STATS=out))
    assert p.returncode == 0, p.stderr.decode()[-1000:]
    d = np.loadtxt(out, dtype=np.int64, ndmin=2)
    q, t, _, score, narrow, below_narrow, below_opt, w2, band_cells, same = d.T
    assert len(d) > 300
    assert (narrow <= score).all()                      # a band can only lose score
    assert (below_narrow <= below_opt).all()
    assert same.all()                                   # the reference's routine, run in the band that holds every cell not excluded: same score, same CIGAR
    cells = (q + 1) * (t + 1)
    assert below_narrow.sum() > 0.5 * cells.sum()       # most of the rectangle is provably off every optimal path
    assert band_cells.sum() < 0.5 * (q * t).sum()


def test_the_banded_kernels_acceptance_bound_is_sound_and_tight():
    """ksw_band.hpp's band_outside_bound against brute force over every alignment path that touches a cell outside the band (tests/cpucheck/band_bound_test.cpp):
    never exceeded -- the banded kernel may accept a result whose score is above it -- and reached exactly, so the kernel's comparison has to be strict."""
    exe = os.path.join(HERE, "_build", "band_bound_test")
    if not os.path.exists(exe):
        pytest.skip("tests/_build/band_bound_test not built")
    p = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()
    assert b"sound on" in p.stdout
