"""The C-ABI library: it loads, exports every symbol include/mm2amd.h declares, mirrors the reference's struct layouts and
option presets, and fails loudly (never silently falls back) when no GPU is present.  No compute calls here."""
import ctypes as C
import os
import re
import subprocess

import pytest

import reflib

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
HDR = os.path.join(ROOT, "include", "mm2amd.h")
HAVE_REF = os.path.exists(reflib.REF_SO)


@pytest.fixture(scope="module")
def L():
    import minimap2_amd as mm
    if not os.path.exists(mm.LIB_PATH):
        from minimap2_amd import build
        build.build()
    return mm.lib()


def declared_functions():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mm2amd_[a-z0-9_]+|mm_gpu_[a-z0-9_]+)\s*\(", src)))


def test_exports_every_declared_symbol(L):
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), "libmm2amd.so does not export %s" % n


def test_product_library_does_not_link_the_oracle():
    import minimap2_amd as mm
    out = subprocess.run(["ldd", mm.LIB_PATH], stdout=subprocess.PIPE).stdout.decode()
    assert "oracle" not in out and "minimap2_ref" not in out
    syms = subprocess.run(["nm", "-D", mm.LIB_PATH], stdout=subprocess.PIPE).stdout.decode()
    assert " ora_" not in syms


def test_fails_loudly_without_gpu(L):
    import minimap2_amd as mm
    if L.mm2amd_device_count() > 0:
        pytest.skip("a GPU is visible")
    assert L.mm2amd_backend_name() == b"hip:gfx950"
    with pytest.raises(mm.Mm2AmdError):
        mm.ksw_extd2_batch([(b"\0\1\2", b"\0\1\2", -1, 400, -1, 0)], reflib.ts_mat(2, 4), 4, 2, 24, 1)
    with pytest.raises(mm.Mm2AmdError):
        mm.Aligner(b"ACGTACGTACGTACGTAGCTAGCTAGCTAGCATCGATCGATCGACTGACTAGC" * 10, preset="map-ont")
    n_reg, reg = (C.c_int * 1)(), (C.c_void_p * 1)()
    assert L.mm_gpu_map_staged(n_reg, reg, None, None) == -5  # MM2AMD_ESTATE: nothing initialised


PRESETS = [None, "lr", "map-ont", "ava-ont", "map10k", "map-pb", "ava-pb", "lr:hq", "map-hifi", "map-ccs", "lr:hqae", "map-iclr-prerender",
           "map-iclr", "asm5", "asm10", "asm20", "short", "sr", "splice", "cdna", "splice:hq", "splice:sr"]


@pytest.mark.skipif(not HAVE_REF, reason="needs oracle/_ref")
@pytest.mark.parametrize("preset", PRESETS)
def test_presets_match_reference(L, preset):
    import minimap2_amd as mm
    R = C.CDLL(reflib.REF_SO)
    io_r, mo_r, io_m, mo_m = mm.IdxOpt(), mm.MapOpt(), mm.IdxOpt(), mm.MapOpt()
    R.mm_set_opt(None, C.byref(io_r), C.byref(mo_r))
    L.mm2amd_set_opt(None, C.byref(io_m), C.byref(mo_m))
    if preset is not None:
        assert R.mm_set_opt(preset.encode(), C.byref(io_r), C.byref(mo_r)) == 0
        assert L.mm2amd_set_opt(preset.encode(), C.byref(io_m), C.byref(mo_m)) == 0
    assert bytes(io_r) == bytes(io_m)
    for name, _ in mm.MapOpt._fields_:
        assert getattr(mo_r, name) == getattr(mo_m, name), (preset, name)
    assert L.mm2amd_set_opt(b"no-such-preset", C.byref(io_m), C.byref(mo_m)) == -1
    assert R.mm_check_opt(C.byref(io_r), C.byref(mo_r)) == L.mm2amd_check_opt(C.byref(io_m), C.byref(mo_m)) == 0


@pytest.mark.skipif(not os.path.exists("/root/reference/minimap.h"), reason="needs the reference headers")
def test_struct_layouts_match_reference_headers(tmp_path):
    """sizeof/offsetof of every struct that crosses the drop-in boundary, taken from the reference's own headers."""
    import minimap2_amd as mm
    fields = {"mm_mapopt_t": [n for n, _ in mm.MapOpt._fields_], "mm_idxopt_t": [n for n, _ in mm.IdxOpt._fields_],
              "mm_reg1_t": ["id", "cnt", "rid", "score", "qs", "qe", "rs", "re", "parent", "subsc", "as", "mlen", "blen", "n_sub", "score0", "hash", "div", "p"],
              "mm_extra_t": ["capacity", "dp_score", "dp_max", "dp_max2", "dp_max0", "n_cigar", "cigar"],
              "mm_bseq1_t": ["l_seq", "rid", "name", "seq", "qual", "comment"],
              "mm_idx_t": ["b", "w", "k", "flag", "n_seq", "index", "n_alt", "seq", "S", "B", "I", "spsc", "J", "km", "h"],
              "mm_idx_seq_t": ["name", "offset", "len", "is_alt"]}
    prog = ['#include <stdio.h>', '#include <stddef.h>', '#include "minimap.h"', '#include "bseq.h"', 'int main(void){']
    for st, fl in fields.items():
        prog.append('printf("%s %%zu\\n", sizeof(%s));' % (st, st))
        for f in fl:
            prog.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (st, f, st, f))
    prog.append("return 0;}")
    src = tmp_path / "lay.c"
    src.write_text("\n".join(prog))
    exe = str(tmp_path / "lay")
    subprocess.check_call(["gcc", "-I/root/reference", str(src), "-o", exe])
    want = dict(l.split() for l in subprocess.check_output([exe]).decode().splitlines())
    mirror = {"mm_mapopt_t": mm.MapOpt, "mm_idxopt_t": mm.IdxOpt, "mm_reg1_t": mm.Reg1, "mm_extra_t": mm.Extra, "mm_bseq1_t": mm.Bseq1}
    for st, cls in mirror.items():
        assert C.sizeof(cls) == int(want[st]), st
        for f in fields[st]:
            pf = {"as": "as_"}.get(f, f)
            if hasattr(cls, pf):
                assert getattr(cls, pf).offset == int(want[st + "." + f]), (st, f)
    assert mm.Reg1.bits.offset == int(want["mm_reg1_t.hash"]) - 4
    # the C++ mirrors (abi_ref.hpp) are checked by static_asserts at build time plus this generated probe
    probe = tmp_path / "probe.cpp"
    lines = ['#include <cstdio>', '#include <cstddef>', '#include "%s/minimap2_amd/csrc/abi_ref.hpp"' % ROOT, 'using namespace mm2amd::ref;', 'int main(){']
    cxx = {"mm_mapopt_t": "MapOpt", "mm_idxopt_t": "IdxOpt", "mm_reg1_t": "Reg1", "mm_extra_t": "Extra", "mm_bseq1_t": "Bseq1", "mm_idx_t": "Idx", "mm_idx_seq_t": "IdxSeq"}
    for st, fl in fields.items():
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (st, cxx[st]))
        for f in fl:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (st, f, cxx[st], f))
    lines.append("return 0;}")
    probe.write_text("\n".join(lines))
    exe2 = str(tmp_path / "probe")
    subprocess.check_call(["g++", "-std=c++17", "-Wno-invalid-offsetof", str(probe), "-o", exe2])
    got = dict(l.split() for l in subprocess.check_output([exe2]).decode().splitlines())
    assert got == want


def test_python_stage_run_marshalling_with_a_fake_library(monkeypatch):
    """Aligner.stage()/run() without a GPU: a stand-in for the C library records what the ctypes layer hands over -- fragment
    table, read records -- for single reads and for read pairs, and hands back empty results."""
    import ctypes as C
    import minimap2_amd as mm

    calls = {}

    class Fake:
        def mm_gpu_batch_stage(self, n, seg_off, n_seg, arr):
            calls["stage"] = (n, list(seg_off)[:n], list(n_seg)[:n], [(arr[k].l_seq, arr[k].name, arr[k].seq) for k in range(sum(list(n_seg)[:n]))])
            return 0

        def mm_gpu_map_staged(self, n_reg, reg, rep_len, frag_gap):
            calls["run"] = len(n_reg)
            return 0

        def mm2amd_free_regs(self, n, n_reg, reg):
            calls["free"] = n

        def mm2amd_last_error(self):
            return b""

        def mm_gpu_context_generation(self):
            return calls.get("generation", 7)

    monkeypatch.setattr(mm, "lib", lambda path=None: Fake())
    al = object.__new__(mm.Aligner)
    al.names, al.lens, al._staged, al._idx, al._generation = ["c1"], [100], None, None, 7
    out = al.map_batch([("a", b"ACGT"), "GGCC", ("c", "TTTTT")])
    assert calls["stage"] == (3, [0, 1, 2], [1, 1, 1], [(4, b"a", b"ACGT"), (4, b"read1", b"GGCC"), (5, b"c", b"TTTTT")])
    assert calls["run"] == 3 and calls["free"] == 3 and out == [[], [], []]
    out = al.map_pairs([("p0", b"ACGT", b"TTGCA"), ("p1", "AAA", "CC")])
    assert calls["stage"] == (2, [0, 2], [2, 2], [(4, b"p0", b"ACGT"), (5, b"p0", b"TTGCA"), (3, b"p1", b"AAA"), (2, b"p1", b"CC")])
    assert calls["run"] == 4 and calls["free"] == 4 and out == [([], []), ([], [])]
    n_reg, reg, rep = (al.stage([("x", b"ACGT")]), al.run(raw=True))[1]
    assert len(n_reg) == 1 and len(reg) == 1 and len(rep) == 1
    calls["generation"] = 8  # another Aligner / mm_gpu_init replaced the process-wide context: this one must say so, not map on the wrong index
    with pytest.raises(mm.Mm2AmdError, match="no longer the process's active mapper"):
        al.map_batch([("a", b"ACGT")])


def test_output_stage_fraction_equals_printf():
    """the de:f / dv:f tags: "%.4f" written with exact integer arithmetic instead of printf -- every value must come out as printf rounds it
    (round half to even on the exact binary value), ties included"""
    import ctypes as C
    import numpy as np
    import minimap2_amd as mm
    L = mm.lib()  # (a host routine: no GPU needed)
    L.mm2amd_format_fraction.argtypes = [C.c_double, C.c_char_p]
    L.mm2amd_format_fraction.restype = C.c_int
    buf = C.create_string_buffer(32)
    rng = np.random.default_rng(5)
    vals = list(rng.random(200000)) + list(rng.random(20000) * 1e-3) + [1.0, 0.5, 0.25, 0.03125, 0.09375, 0.00005, 0.00015, 1e-300, 5e-324, 0.99995, 0.999949999, 0.12345, 0.12355]
    vals += [(2 * j + 1) / 2.0 ** k for k in range(1, 30) for j in range(0, min(2 ** (k - 1), 40))]  # dyadic fractions: the only exact ties
    vals += [float(np.nextafter(x, 0.0)) for x in (0.00005, 0.12345, 0.5)] + [float(np.nextafter(x, 1.0)) for x in (0.00005, 0.12345, 0.5)]
    for v in vals:
        n = L.mm2amd_format_fraction(float(v), buf)
        want = "0" if v == 0.0 else "%.4f" % v
        assert buf.value[:n].decode() == want, (v, buf.value, want)
