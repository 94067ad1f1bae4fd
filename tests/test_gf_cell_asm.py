"""The gap-fill kernels' cell bodies (gf_cell, gf_cell_k in minimap2_amd/csrc/ksw_gapfill_dev.hpp) are gfx950 inline assembly: the wave emulator runs
the kernels around them with C++ twins (tests/cpucheck/wave_emu/ksw_pk_emu.hpp), so until a GPU run nothing checked the instruction lists
themselves.  This case reads the assembly out of the header, interprets it instruction by instruction on both 16-bit halves (VOP3P semantics:
op_sel_hi picks the half of an operand that feeds the high lane, an inline constant has a zero high half) and compares every output with the
recurrences of ksw2_extd2_sse.c:165-272 (left-aligned gaps) written out in plain integers, on random cells of every preset's scoring; it also
checks the spacing rule the blocks were written to (no instruction reads the result of the packed instruction right before it).
Runs without a GPU; the hardware cases (tests/test_gpu_ksw.py) remain the proof that the assembler and the chip agree with this reading."""
import os
import re

import numpy as np
import pytest

HDR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "minimap2_amd", "csrc", "ksw_gapfill_dev.hpp")
PRESETS = {"ont": (2, 4, 4, 2, 24, 1), "hifi": (1, 4, 6, 2, 26, 1), "swap": (2, 4, 24, 1, 4, 2), "asm5": (1, 19, 39, 3, 81, 1), "asm20": (1, 4, 6, 2, 26, 1), "sr": (2, 8, 12, 2, 24, 1)}


def _function(name, path=HDR):
    src = open(path).read()
    m = re.search(r"__device__ __forceinline__ void %s\(" % name, src)
    assert m, name
    body = src[m.start():]
    end = body.index("\n}\n")
    return body[:end]


def _blocks(body):
    """[(instructions, {asm name: C variable})] per asm volatile block"""
    out = []
    for blk in re.findall(r"asm volatile\((.*?)\);", body, re.S):
        ins = []
        lines = blk.split("\n")
        k = 0
        while k < len(lines) and not lines[k].strip().startswith(":"):
            for s in re.findall(r'"([^"]*)"', lines[k]):
                s = s.replace("\\n\\t", "").strip()
                if s:
                    ins.append(s)
            k += 1
        binds = dict(re.findall(r'\[(\w+)\]\s*"[^"]*"\(([\w.]+)\)', "\n".join(lines[k:])))
        out.append((ins, binds))
    return out


def _halves(v):
    return v & 0xffff, (v >> 16) & 0xffff


def _s16(v):
    return v - 0x10000 if v & 0x8000 else v


OPS2 = {
    "v_pk_add_u16": lambda a, b: (a + b) & 0xffff,
    "v_pk_sub_u16": lambda a, b: (a - b) & 0xffff,
    "v_pk_min_u16": lambda a, b: min(a, b),
    "v_pk_max_i16": lambda a, b: a if _s16(a) > _s16(b) else b,
    "v_pk_min_i16": lambda a, b: a if _s16(a) < _s16(b) else b,
    "v_pk_mul_lo_u16": lambda a, b: (a * b) & 0xffff,
    "v_pk_lshrrev_b16": lambda a, b: b >> (a & 15),
}


OPS32 = {"v_and_b32": lambda a, b: a & b, "v_xor_b32": lambda a, b: a ^ b, "v_or_b32": lambda a, b: a | b}


def _run(blocks, env):
    """interpret; env maps C variable names to 32-bit values.  Returns the number of packed / 32-bit VALU instructions."""
    n_pk = n_32 = 0
    for ins, binds in blocks:
        prev_dst, prev_pk = None, False
        for line in ins:
            m = re.match(r"(\w+)\s+(.*)", line)
            op, rest = m.group(1), m.group(2)
            if op == "s_nop":
                prev_dst = None
                continue
            sel_hi = None
            ms = re.search(r"op_sel_hi:\[([01,]+)\]", rest)
            if ms:
                sel_hi = [int(x) for x in ms.group(1).split(",")]
                rest = rest[:ms.start()].strip()
            toks = [t.strip() for t in rest.split(",")]
            dst, srcs = toks[0], toks[1:]

            def val(tok):
                mm = re.match(r"%\[(\w+)\]", tok)
                if mm:
                    return env[binds[mm.group(1)]] & 0xffffffff
                return int(tok, 0) & 0xffffffff  # inline constant / literal: a 32-bit value (zero high half for the small ones)
            src_names = [binds[re.match(r"%\[(\w+)\]", t).group(1)] for t in srcs if t.startswith("%")]
            assert not (prev_pk and prev_dst in src_names), "%s reads the result of the packed instruction right before it" % line
            d = binds[re.match(r"%\[(\w+)\]", dst).group(1)]
            if op in OPS32:
                env[d] = OPS32[op](val(srcs[0]), val(srcs[1]))
                n_32 += 1
                prev_dst, prev_pk = d, False
                continue
            vs = [val(t) for t in srcs]
            if sel_hi is None:
                sel_hi = [1] * len(vs)
            assert len(sel_hi) == len(vs), line
            lo = [v & 0xffff for v in vs]
            hi = [(v >> 16) & 0xffff if s else v & 0xffff for v, s in zip(vs, sel_hi)]
            if op == "v_pk_mad_u16":
                r_lo, r_hi = (lo[0] * lo[1] + lo[2]) & 0xffff, (hi[0] * hi[1] + hi[2]) & 0xffff
            else:
                f = OPS2[op]
                r_lo, r_hi = f(lo[0], lo[1]) & 0xffff, f(hi[0], hi[1]) & 0xffff
            env[d] = r_lo | r_hi << 16
            n_pk += 1
            prev_dst, prev_pk = d, True
    return n_pk, n_32


def _plain_cell(tb, qb, xp, vp, x2p, u, y, y2, sc):
    """one valid cell, ksw2_extd2_sse.c:165-272 (left-aligned: the first candidate that reaches the maximum names the state)"""
    a_, b_, q, e, q2, e2 = sc
    if q2 + e2 < q + e:
        q, e, q2, e2 = q2, e2, q, e
    qe, qe2 = q + e, q2 + e2
    sc_n = -e2  # the presets' matrices have a zero N score: ksw2_extd2_sse.c:71
    z = sc_n if (tb | qb) & 4 else (a_ if tb == qb else -b_)
    a, b, a2, b2 = xp + vp, y + u, x2p + vp, y2 + u
    d = 0
    for k, c in enumerate((a, b, a2, b2)):
        if c > z:
            d, z = k + 1, c
    z = min(z, a_)
    un, vn = z - vp, z - u
    a, b, a2, b2 = a - (z - q), b - (z - q), a2 - (z - q2), b2 - (z - q2)
    for bit, c in ((0x08, a), (0x10, b), (0x20, a2), (0x40, b2)):
        if c > 0:
            d |= bit
    return un, vn, max(a, 0) - qe, max(b, 0) - qe, max(a2, 0) - qe2, max(b2, 0) - qe2, d


def _pk(lo, hi):
    return (lo & 0xffff) | (hi & 0xffff) << 16


@pytest.mark.parametrize("preset", list(PRESETS))
@pytest.mark.parametrize("keyed", [False, True])
def test_cell_assembly_against_the_recurrences(preset, keyed):
    sc = PRESETS[preset]
    a_, b_, q, e, q2, e2 = sc
    if q2 + e2 < q + e:
        q, e, q2, e2 = q2, e2, q, e
    qe, qe2 = q + e, q2 + e2
    blocks = _blocks(_function("gf_cell_k" if keyed else "gf_cell"))
    K = 8 if keyed else 1
    TS, TA, TB, TA2, TB2 = (7, 6, 5, 4, 3) if keyed else (0, 0, 0, 0, 0)
    BIAS = 0
    rng = np.random.default_rng(5 + len(preset))
    counts = None
    for it in range(4000):
        cells = []
        for h in range(2):
            tb, qb = int(rng.integers(0, 5)), int(rng.integers(0, 5))
            if it % 3 == 0:
                qb = tb  # more matches than chance gives
            # difference states in the ranges the recurrences keep them in: x, y in [-qe, -e], x2, y2 in [-qe2, -e2], u, v in [-qe2, a + qe2]
            xp, y = int(rng.integers(-qe, -e + 1)), int(rng.integers(-qe, -e + 1))
            x2p, y2 = int(rng.integers(-qe2, -e2 + 1)), int(rng.integers(-qe2, -e2 + 1))
            u, vp = int(rng.integers(-qe2, a_ + qe2 + 1)), int(rng.integers(-qe2, a_ + qe2 + 1))
            if it % 5 == 0:  # ties between candidates on purpose
                vp = u + y - xp
            cells.append((tb, qb, xp, vp, x2p, u, y, y2))
        A, B = cells
        env = dict(x1=_pk(A[0] ^ A[1], B[0] ^ B[1]), o1=_pk(A[0] | A[1], B[0] | B[1]),
                   xp=_pk(K * A[2] + TA, K * B[2] + TA), vp=_pk(K * A[3], K * B[3]), x2p=_pk(K * A[4] + TA2, K * B[4] + TA2),
                   u=_pk(K * A[5], K * B[5]), y=_pk(K * A[6] + TB, K * B[6] + TB), y2=_pk(K * A[7] + TB2, K * B[7] + TB2))
        pk2 = lambda v: _pk(v, v)
        if keyed:  # gf_k_consts (ksw_gapfill_dev.hpp)
            ka, kb, ka2, kb2 = TA - 8 * qe, TB - 8 * qe, TA2 - 8 * qe2, TB2 - 8 * qe2
            BIAS = (ka + 2 * kb + 4 * ka2 + 8 * kb2) & 0xff
            env.update({"P_MCHT": pk2(8 * a_ + TS), "K.misd8": pk2(8 * (-b_ - a_)), "K.scnt": pk2(8 * -e2 + TS), "K.mch8": pk2(8 * a_), "K.e8": pk2(8 * e), "K.e28": pk2(8 * e2),
                        "K.ka": pk2(ka), "K.kb": pk2(kb), "K.ka2": pk2(ka2), "K.kb2": pk2(kb2), "K.ka8": pk2(ka + 8), "K.kb8": pk2(kb + 8), "K.ka28": pk2(ka2 + 8), "K.kb28": pk2(kb2 + 8)})
        else:
            env.update(P_MCH=pk2(a_), S_MISD=pk2(-b_ - a_), S_SCN=pk2(-e2), S_Q=pk2(q), S_Q2=pk2(q2), S_QE=pk2(qe), S_QE2=pk2(qe2))
        counts = _run(blocks, env)
        for h, c in enumerate(cells):
            want = _plain_cell(*c, sc)
            half = lambda name: _s16(_halves(env[name])[h])
            got = (half("u" if keyed else "un"), half("vn"), half("xn") - TA, half("yn") - TB, half("x2n") - TA2, half("y2n") - TB2)
            assert got == tuple(K * w for w in want[:6]), (it, h, c, got, want)
            byte = _halves(env["e"])[h] & 0xff  # what the kernels store (v_perm picks the low byte of each half)
            d = ((byte - BIAS) ^ 7) & 0xff if keyed else byte
            assert d == want[6], (it, h, c, hex(byte), hex(d), hex(want[6]))
    assert counts == ((34, 2) if keyed else (50, 0))  # the operation counts DESIGN.md quotes
