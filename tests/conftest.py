import os, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


def pytest_collection_modifyitems(config, items):
    # a GPU case that hangs must cost seconds of box time, not the whole call (pytest-timeout; the full-size cases set their own)
    for it in items:
        if it.get_closest_marker("gpu") and not it.get_closest_marker("timeout"):
            it.add_marker(pytest.mark.timeout(3600 if os.environ.get("MM2AMD_EMU") == "1" else 180))


# MM2AMD_EMU=1: run the GPU cases against tests/_build/libmm2amd_emu.so -- the product's own sources, kernels included, built for the host
# under the wave emulator (tests/cpucheck/wave_emu) -- in a container without a GPU.  Development aid: `MM2AMD_EMU=1 pytest -m gpu -k ...`;
# the CPU suite's own emulator cases are in tests/test_wave_emu.py.
EMU = os.environ.get("MM2AMD_EMU") == "1"


def emu_lib_path():
    return os.environ.get("MM2AMD_EMU_LIB") or os.path.join(ROOT, "tests", "_build", "libmm2amd_emu.so")  # (MM2AMD_EMU_LIB: e.g. tools/sanitize_emu.sh's AddressSanitizer build)


def use_emulated_library():
    import ctypes
    import minimap2_amd as mm
    mm._lib = mm._bind(ctypes.CDLL(emu_lib_path()))
    return mm


if EMU:
    use_emulated_library()
