"""The option fuzzing of tests/test_option_fuzz.py on the GPU path: the same random option subsets, drop-in binary with the HIP
backend against the reference binary.  MM2AMD_FUZZ_SEEDS=n widens the sweep (default 24 cases)."""
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import test_option_fuzz as F  # noqa: E402

pytestmark = pytest.mark.gpu
DROPIN = os.path.join(HERE, "_build", "dropin_emu" if os.environ.get("MM2AMD_EMU") == "1" else "dropin_gpu")  # MM2AMD_EMU=1: tests/conftest.py
N = int(os.environ.get("MM2AMD_FUZZ_SEEDS", "24"))

inputs = F.inputs  # the module-scoped fixture


@pytest.mark.parametrize("seed", range(3000, 3000 + N))
def test_mapping_options_on_the_gpu(inputs, seed, monkeypatch):
    monkeypatch.setattr(F, "CHECK", DROPIN)
    F._case(inputs, seed, F.OPTS + F.FORMAT_OPTS, seed % 3 == 0)
