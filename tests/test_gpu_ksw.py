"""HIP ksw_extd2 (through the C ABI) against the oracle restatement and, when present, the compiled reference."""
import os
import numpy as np
import pytest

import reflib
from reflib import ora_extd2, ts_mat
from seqsim import random_pair

pytestmark = pytest.mark.gpu

PRESETS = {"ont": (2, 4, 4, 2, 24, 1), "hifi": (1, 4, 6, 2, 26, 1), "swap": (2, 4, 24, 1, 4, 2), "asm5": (1, 19, 39, 3, 81, 1)}


def _run(jobs, preset, transition=0):
    import minimap2_amd as mm
    a, b, go, ge, go2, ge2 = PRESETS[preset]
    mat = ts_mat(a, b, 1, transition)
    got = mm.ksw_extd2_batch(jobs, mat, go, ge, go2, ge2)
    have_ref = os.path.exists(reflib.REF_SO)
    for k, (q, t, w, zdrop, eb, flag) in enumerate(jobs):
        want = ora_extd2(q, t, mat, go, ge, go2, ge2, w, zdrop, eb, flag)
        assert got[k] == want, ("oracle", k, len(q), len(t), w, zdrop, eb, hex(flag), preset)
        if have_ref and k % 7 == 0:
            assert got[k] == reflib.ref_extd2(q, t, mat, go, ge, go2, ge2, w, zdrop, eb, flag), ("reference", k)


@pytest.mark.parametrize("preset", list(PRESETS))
def test_unbanded_all_flags(preset):
    rng = np.random.default_rng(42)
    jobs = []
    for it in range(400):
        q, t = random_pair(rng, int(rng.integers(1, 500)), float(rng.choice([0.0, 0.05, 0.12, 0.3])), float(rng.choice([0, 0, 0.02])))
        jobs.append((q, t, 30001, int(rng.choice([-1, 200, 400])), int(rng.choice([-1, 10])), int(rng.choice([0x08, 0x00, 0x40, 0xC2, 0x41]))))
    _run(jobs, preset)


def test_generic_scoring_matrix():
    rng = np.random.default_rng(5)
    jobs = []
    for it in range(100):
        q, t = random_pair(rng, int(rng.integers(1, 300)), 0.1, 0.02)
        jobs.append((q, t, 30001, 400, 10, int(rng.choice([0x0C, 0x04, 0x44, 0xC6]))))
    _run(jobs, "ont", transition=3)


def test_band_binding_lane_exact():
    rng = np.random.default_rng(9)
    jobs = []
    for it in range(400):
        q, t = random_pair(rng, int(rng.integers(20, 900)), float(rng.choice([0.02, 0.12])), 0.0, int(rng.choice([0, 0, 30, -30, 150, -150])))
        jobs.append((q, t, int(rng.integers(1, 120)), int(rng.choice([-1, 100, 400])), int(rng.choice([-1, 10])), int(rng.choice([0x40, 0xC2, 0x00, 0x08]))))
    _run(jobs, "ont")


def test_multiple_of_16_targets():
    rng = np.random.default_rng(7)
    jobs = []
    for tl in (16, 32, 48, 64, 256, 512):
        for it in range(10):
            t = rng.integers(0, 4, tl, dtype=np.uint8)
            q = rng.integers(0, 4, int(rng.integers(1, min(2 * tl, 512))), dtype=np.uint8)
            for flag in (0x08, 0x40, 0xC2, 0):
                for w in (5, 751, 30001):
                    jobs.append((q, t, w, 400, 10, flag))
    _run(jobs, "ont")


def test_long_extensions_band_751():
    rng = np.random.default_rng(11)
    jobs = []
    for it in range(12):
        q, t = random_pair(rng, int(rng.integers(1500, 5000)), 0.12, 0.0, int(rng.choice([0, 700, -700, 900])))
        jobs.append((q, t, 751, 400, 10, [0x40, 0xC2][it & 1]))
    _run(jobs, "ont")


def test_degenerate_jobs():
    z = np.zeros(0, dtype=np.uint8)
    one = np.array([1], dtype=np.uint8)
    jobs = [(one, one, 751, 400, 10, 0x40), (one, np.array([2], dtype=np.uint8), 30001, 400, -1, 0x08), (one, one, 0, 400, -1, 0)]
    _run(jobs, "ont")
    import minimap2_amd as mm
    got = mm.ksw_extd2_batch([(z, one, 10, 400, -1, 0)], ts_mat(2, 4), 4, 2, 24, 1)
    assert got[0][0] == 0 and got[0][10] == ()


@pytest.mark.parametrize("preset", list(PRESETS))
def test_register_resident_gap_fill_kernel(preset):
    """jobs that take ksw_gapfill.hip (flag 0x08, non-binding band): every register-set boundary, N bases, long indels, unrelated
    sequences, extreme aspect ratios -- against the lane-exact oracle, and A/B against the exact HIP kernel"""
    import minimap2_amd as mm
    rng = np.random.default_rng(77)
    jobs = []
    for tl in (1, 2, 15, 16, 17, 63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 256, 257, 300, 383, 384, 385, 447, 448, 449, 511, 512, 513, 600, 639, 640,
               641, 700, 767, 768, 769):
        for rep in range(3):
            t = rng.integers(0, 4, tl, dtype=np.uint8)
            if rep == 0:
                q = t.copy()
            elif rep == 1:
                q, _ = random_pair(rng, tl, 0.15, 0.02)
                q = q[:1024]
            else:
                q = rng.integers(0, 4, int(rng.integers(1, 1025)), dtype=np.uint8)  # unrelated, any aspect ratio
            jobs.append((q, t, 30001, int(rng.choice([-1, 200, 400])), int(rng.choice([-1, 10])), 0x08))
    for it in range(300):
        q, t = random_pair(rng, int(rng.integers(1, 768)), float(rng.choice([0.0, 0.05, 0.12, 0.3, 0.6])), float(rng.choice([0, 0, 0.03])),
                           int(rng.choice([0, 0, 0, 40, -40, 200, -200])))
        if len(q) > 1024 or len(t) > 768:
            continue
        tight = max(len(q), len(t)) - 1  # the smallest band that cannot bind (ksw_host.cpp: band_cannot_bind); one less can
        w = int(rng.choice([30001, len(q) + len(t), len(q) + len(t) + 5, -1, tight, tight, max(tight - 1, 0)]))
        jobs.append((q, t, w, 400, -1, 0x08))
    # not eligible (band could bind / other flags): must still be exact
    for it in range(40):
        q, t = random_pair(rng, int(rng.integers(50, 400)), 0.12)
        jobs.append((q, t, len(q) + len(t) - 1, 400, -1, 0x08))
        jobs.append((q, t, 30001, 400, -1, 0x18))
    _run(jobs, preset)
    a, b, go, ge, go2, ge2 = PRESETS[preset]
    mat = ts_mat(a, b, 1, 0)
    fast = mm.ksw_extd2_batch(jobs, mat, go, ge, go2, ge2)
    os.environ["MM2AMD_KSW_EXACT_ONLY"] = "1"
    try:
        exact = mm.ksw_extd2_batch(jobs, mat, go, ge, go2, ge2)
    finally:
        del os.environ["MM2AMD_KSW_EXACT_ONLY"]
    assert fast == exact


@pytest.mark.parametrize("preset", list(PRESETS))
def test_streaming_gap_fill_kernel(preset, monkeypatch):
    """ksw_stream.hip runs the jobs of a wavefront back to back through the lanes (job k+1's first anti-diagonals fill the lanes job
    k's last ones have left): with four persistent waves, each half-wave streams ~200 jobs of every shape the two classes take
    (query <= 512, target <= 256 / <= 512) -- narrow after wide, long after short, single cells, N bases, unrelated sequences --
    against the lane-exact oracle; then the same jobs one pair per wave, and through the strip kernel"""
    import minimap2_amd as mm
    rng = np.random.default_rng(83)
    jobs = []
    for it in range(1600):
        kind = it % 8
        if kind == 0:
            q, t = rng.integers(0, 4, int(rng.integers(1, 513)), dtype=np.uint8), rng.integers(0, 4, int(rng.integers(1, 513)), dtype=np.uint8)
        elif kind == 1:
            q, t = rng.integers(0, 4, int(rng.integers(1, 6)), dtype=np.uint8), rng.integers(0, 4, int(rng.integers(1, 513)), dtype=np.uint8)
        elif kind == 2:
            q, t = rng.integers(0, 4, int(rng.integers(1, 513)), dtype=np.uint8), rng.integers(0, 4, int(rng.integers(1, 6)), dtype=np.uint8)
        else:
            q, t = random_pair(rng, int(rng.choice([1, 2, 30, 63, 64, 65, 100, 128, 200, 255, 256, 257, 300, 400, 511, 512])) if kind == 3 else int(rng.integers(1, 513)),
                               float(rng.choice([0.0, 0.05, 0.12, 0.3])), float(rng.choice([0, 0, 0.03])), int(rng.choice([0, 0, 0, 40, -40, 150, -150])))
            q, t = q[:512], t[:512]
        if len(q) == 0 or len(t) == 0:
            continue
        jobs.append((q, t, int(rng.choice([30001, -1])), 400, -1, 0x08))
    monkeypatch.setenv("MM2AMD_KSW_MAX_SLOTS", "4")
    _run(jobs, preset)
    a, b, go, ge, go2, ge2 = PRESETS[preset]
    mat = ts_mat(a, b, 1, 0)
    streamed = mm.ksw_extd2_batch(jobs, mat, go, ge, go2, ge2)
    monkeypatch.delenv("MM2AMD_KSW_MAX_SLOTS")
    assert mm.ksw_extd2_batch(jobs, mat, go, ge, go2, ge2) == streamed
    monkeypatch.setenv("MM2AMD_NO_STREAM", "1")
    assert mm.ksw_extd2_batch(jobs, mat, go, ge, go2, ge2) == streamed


@pytest.mark.parametrize("preset", ["ont", "asm5"])
def test_gap_fill_kernel_column_strips(preset):
    """ksw_gapfill.hip sweeps targets wider than 256 columns in strips of 256 (boundary column handed over through LDS), with two
    query capacities (512 / 1024) and targets up to three times that: strip boundaries, class boundaries, pairs whose jobs need
    different strip counts, queries shorter than / as long as / longer than the capacity -- against the lane-exact oracle"""
    rng = np.random.default_rng(79)
    jobs = []
    for tl in (257, 511, 512, 513, 767, 768, 769, 1023, 1024, 1025, 1535, 1536, 1537, 2047, 2049, 3071, 3072, 3073):
        for ql in (1, 40, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025):
            if rng.random() < 0.55:
                continue
            if rng.random() < 0.5:
                q, t = random_pair(rng, max(ql, tl), 0.12, 0.02)
                q, t = q[:ql], t[:tl]
            else:
                q, t = rng.integers(0, 4, ql, dtype=np.uint8), rng.integers(0, 4, tl, dtype=np.uint8)
            jobs.append((q, t, 30001, 400, -1, 0x08))
    for it in range(60):  # related sequences with a long indel crossing a strip boundary, some N bases
        n = int(rng.integers(300, 1000))
        q, t = random_pair(rng, n, float(rng.choice([0.05, 0.12])), 0.02, int(rng.choice([0, 60, -60, 300, -300])))
        if len(q) > 1024:
            q = q[:1024]
        if it % 5 == 0:
            t = t.copy()
            t[rng.integers(0, len(t), 5)] = 4
        jobs.append((q, t, -1 if it % 2 else len(q) + len(t), 400, -1, 0x08))
    _run(jobs, preset)


def test_jobs_longer_than_lds_use_the_hbm_state_kernel():
    """long jobs: a banded extension across a long gap and long thin global jobs (their state window is as wide as their widest anti-diagonal, so
    they still fit an LDS ring), and jobs whose anti-diagonals are wider than the largest LDS ring (8192 slots: query, target AND band beyond
    8128) -- the only ones that reach ksw_extd2_kernel<false, ...>, the instantiation with its per-position state in HBM ('ksw_extd2_kernel[hbm]';
    the emulator's line coverage showed that no case did)"""
    rng = np.random.default_rng(21)
    jobs = []
    t = rng.integers(0, 4, 15000, dtype=np.uint8)
    q = np.concatenate([t[:600], t[13000:13800]])            # 12.4 kb deletion in the query
    jobs.append((q, t, 30001, 400, -1, 0x08))                # global, approximate score
    jobs.append((q, t, 751, 400, 10, 0x40))                  # banded extension, exact maxima + Z-drop
    q2, t2 = random_pair(rng, 12500, 0.1)
    jobs.append((q2[:12000], t2[:12800], 751, 400, -1, 0xC2))
    _run(jobs, "ont")
    import minimap2_amd as mm
    q3, t3 = random_pair(np.random.default_rng(22), 8400, 0.1)
    wide = [(q3[:8300], t3[:8350], 30001, 400, -1, 0x08), (q3[:8200], t3[:8400], 9000, 400, 10, 0x40), (q3[:8250], t3[:8300], 8200, 400, -1, 0xC2)]
    mm.lib().mm2amd_profile_enable(1)
    try:
        _run(wide, "ont")
        stats = (mm.KernelStat * 64)()
        names = [stats[i].name.decode() for i in range(mm.lib().mm2amd_profile_get(stats, 64))]
    finally:
        mm.lib().mm2amd_profile_enable(0)
    assert "ksw_extd2_kernel[hbm]" in names, names


@pytest.mark.parametrize("sc", [(2, 4, 4, 2), (1, 4, 6, 2), (1, 9, 16, 2), (1, 2, 2, 1)])
def test_single_affine_extz2(sc):
    """the single-affine instantiation of the exact kernel (ksw_extz2_sse semantics) against its oracle and the reference"""
    import minimap2_amd as mm
    from reflib import ora_extz2, ref_extz2
    a, b, go, ge = sc
    rng = np.random.default_rng(sum(sc))
    mat = ts_mat(a, b, 1, 0)
    jobs = []
    for it in range(250):
        q, t = random_pair(rng, int(rng.integers(1, 500)), float(rng.choice([0.0, 0.05, 0.12, 0.3])), float(rng.choice([0, 0, 0.02])))
        jobs.append((q, t, 30001, int(rng.choice([-1, 100, 400])), int(rng.choice([-1, 10])), int(rng.choice([0x08, 0x00, 0x40, 0xC2, 0x41, 0x18]))))
    for it in range(150):  # binding bands
        q, t = random_pair(rng, int(rng.integers(20, 900)), float(rng.choice([0.02, 0.12])), 0.0, int(rng.choice([0, 0, 30, -30, 150, -150])))
        jobs.append((q, t, int(rng.integers(1, 120)), int(rng.choice([-1, 100, 400])), int(rng.choice([-1, 10])), int(rng.choice([0x40, 0xC2, 0x00, 0x08]))))
    for tl in (16, 64, 256):
        t = rng.integers(0, 4, tl, dtype=np.uint8)
        q = rng.integers(0, 4, int(rng.integers(1, 2 * tl)), dtype=np.uint8)
        for flag in (0x08, 0x40, 0xC2, 0):
            for w in (5, 751, 30001):
                jobs.append((q, t, w, 400, 10, flag))
    q, t = random_pair(rng, 2500, 0.12, 0.0, 700)
    jobs.append((q, t, 751, 400, 10, 0x40))
    got = mm.ksw_extz2_batch(jobs, mat, go, ge)
    have_ref = os.path.exists(reflib.REF_SO)
    for k, (q, t, w, zdrop, eb, flag) in enumerate(jobs):
        assert got[k] == ora_extz2(q, t, mat, go, ge, w, zdrop, eb, flag), (k, len(q), len(t), w, hex(flag))
        if have_ref and k % 9 == 0:
            assert got[k] == ref_extz2(q, t, mat, go, ge, w, zdrop, eb, flag)


@pytest.mark.parametrize("sc", [(1, 2, 2, 1, 32, 9), (1, 4, 6, 1, 24, 5)])
def test_splice_exts2(sc):
    """the splice instantiation of the exact kernel (ksw_exts2_sse semantics, junc == NULL) against its oracle and the reference:
    both transcript strands, both splice models, left/right extension and gap fill, multi-exon targets, an intron long enough
    for the HBM-state variant"""
    import minimap2_amd as mm
    from reflib import ora_exts2, ref_exts2
    from seqsim import spliced_pair
    a, b, go, ge, go2, noncan = sc
    rng = np.random.default_rng(sum(sc))
    mat = ts_mat(a, b, 1, 0)
    jobs = []
    for it in range(300):
        q, t = spliced_pair(rng, int(rng.integers(1, 5)), float(rng.choice([0.0, 0.03, 0.1])))
        base = int(rng.choice([0x08, 0x00, 0x40, 0xC2, 0x41, 0x18]))
        if base & 0x80:
            q, t = q[::-1].copy(), t[::-1].copy()
        flag = base | int(rng.choice([0x100, 0x200])) | 0x400 | (0x800 if it % 3 else 0)
        jobs.append((q, t, -1, int(rng.choice([-1, 100, 200])), int(rng.choice([-1, 5])), flag))
    for tl in (16, 64, 256):
        t = rng.integers(0, 4, tl, dtype=np.uint8)
        q = rng.integers(0, 4, int(rng.integers(1, 2 * tl)), dtype=np.uint8)
        for flag in (0x08, 0x40, 0xC2, 0):
            jobs.append((q, t, -1, 200, 5, flag | 0x100 | 0x400 | 0x800))
            jobs.append((q, t, 7, 200, 5, flag))  # no strand: no splice signal costs; w is ignored
    q, t = spliced_pair(rng, 3, 0.03, exon=(100, 200), intron=(9000, 14000))
    jobs.append((q, t, -1, 200, -1, 0x100 | 0x400 | 0x800))
    jobs.append((q, t, -1, 200, -1, 0x200 | 0x400 | 0x08))
    got = mm.ksw_exts2_batch(jobs, mat, go, ge, go2, noncan)
    have_ref = os.path.exists(reflib.REF_SO)
    for k, (q, t, w, zdrop, eb, flag) in enumerate(jobs):
        assert got[k] == ora_exts2(q, t, mat, go, ge, go2, noncan, zdrop, eb, 9, 5, flag), (k, len(q), len(t), hex(flag))
        if have_ref and k % 9 == 0:
            assert got[k] == ref_exts2(q, t, mat, go, ge, go2, noncan, zdrop, eb, 9, 5, flag)
    assert any(c & 0xf == 3 for r in got for c in r[10])  # introns were found


def test_state_window_wraps_with_overshoot_past_the_arrays():
    """jobs whose target is much longer than the exact kernel's state window (the rings wrap many times) and whose length is a
    multiple of 16 or just below one, with a query longer than the target: the reference's score fill then runs past its s[]
    array into the target copy (ksw2_extd2_sse.c:166-180); narrow, medium and absent bands"""
    import minimap2_amd as mm
    rng = np.random.default_rng(77)
    mat = ts_mat(2, 4, 1, 0)
    jobs = []
    for tl in (1024, 1023, 1017, 2048, 1536):
        for ql in (tl + 300, tl - 200, tl // 2):
            for w in (5, 100, 400, -1):
                q, t = random_pair(rng, tl, 0.1, 0.0, 0)
                t = t[:tl] if len(t) >= tl else np.concatenate([t, rng.integers(0, 4, tl - len(t), dtype=np.uint8)])
                q = q[:ql] if len(q) >= ql else np.concatenate([q, rng.integers(0, 4, ql - len(q), dtype=np.uint8)])
                jobs.append((q, t, w, 400, 10, int(rng.choice([0x40, 0xC2, 0x00, 0x08]))))
    got = mm.ksw_extd2_batch(jobs, mat, 4, 2, 24, 1)
    for k, (q, t, w, zdrop, eb, flag) in enumerate(jobs):
        assert got[k] == ora_extd2(q, t, mat, 4, 2, 24, 1, w, zdrop, eb, flag), (k, len(q), len(t), w, hex(flag))


@pytest.mark.parametrize("sc", [(1, 2, 2, 1, 32, 9), (1, 4, 6, 1, 24, 5)])
def test_splice_gap_fill_kernel(sc, monkeypatch):
    """the register-resident splice kernel (ksw_splice.hip: gap-fill calls, flag APPROX_MAX) against the oracle: paired jobs of
    different shapes, queries from 1 to 2049 bases (the paired variants, the one-job variant and its multi-strip sweeps), introns up to 30 kb, both strands / splice models /
    no strand, target shorter than the query; and the same jobs through the lane-exact kernel (MM2AMD_KSW_EXACT_ONLY)"""
    import minimap2_amd as mm
    from reflib import ora_exts2
    from seqsim import spliced_pair
    a, b, go, ge, go2, noncan = sc
    rng = np.random.default_rng(100 + sum(sc))
    mat = ts_mat(a, b, 1, 0)
    jobs = []
    for it in range(260):
        n_exon = int(rng.integers(1, 4))
        q, t = spliced_pair(rng, n_exon, float(rng.choice([0.0, 0.03, 0.1])), exon=(10, 170), intron=(30, 3000))
        strand = int(rng.choice([0x100, 0x200, 0x100, 0x200, 0]))
        flag = 0x08 | strand | (0x400 if it % 4 else 0) | (0x800 if it % 3 else 0)
        jobs.append((q[:512], t, -1, 200, -1, flag))
    for ql in (1, 2, 63, 64, 65, 128, 129, 256, 257, 511, 512):
        q = rng.integers(0, 4, ql, dtype=np.uint8)
        for tl in (1, 5, 64, 200, 1000):
            t = rng.integers(0, 4, tl, dtype=np.uint8)
            jobs.append((q, t, -1, 200, -1, 0x08 | 0x100 | 0x400 | 0x800))
    for ql in (257, 300, 511, 512, 513, 700, 1023, 1024, 1025, 1500, 2047, 2048, 2049):  # one job per wave, both register halves, 1..5 strips
        n_exon = 3
        q, t = spliced_pair(rng, n_exon, 0.04, exon=(ql // n_exon + 2, ql // n_exon + 3), intron=(100, 2500))
        q = q[:ql] if len(q) >= ql else np.concatenate([q, rng.integers(0, 4, ql - len(q), dtype=np.uint8)])
        jobs.append((q, t, -1, 200, -1, 0x08 | [0x100, 0x200][ql & 1] | 0x400 | 0x800))
        jobs.append((q, t[:max(1, ql // 3)], -1, 200, -1, 0x08 | 0x100 | 0x400 | 0x800))  # target shorter than the query
    for it in range(6):
        q, t = spliced_pair(rng, 3, 0.04, exon=(60, 120), intron=(8000, 30000))
        jobs.append((q, t, -1, 200, -1, 0x08 | [0x100, 0x200][it & 1] | 0x400 | 0x800))
    q, t = spliced_pair(rng, 2, 0.02, exon=(100, 101), intron=(300, 301))
    t[rng.random(len(t)) < 0.05] = 4  # ambiguous bases
    jobs.append((q, t, -1, 200, -1, 0x08 | 0x100 | 0x400 | 0x800))
    got = mm.ksw_exts2_batch(jobs, mat, go, ge, go2, noncan)
    for k, (q, t, w, zdrop, eb, flag) in enumerate(jobs):
        assert got[k] == ora_exts2(q, t, mat, go, ge, go2, noncan, zdrop, eb, 9, 5, flag), (k, len(q), len(t), hex(flag))
    monkeypatch.setenv("MM2AMD_KSW_EXACT_ONLY", "1")
    assert mm.ksw_exts2_batch(jobs, mat, go, ge, go2, noncan) == got


@pytest.mark.parametrize("preset", list(PRESETS))
def test_extension_calls(preset):
    """the two extension calls of mm_align1 (right extension 0x40, left extension 0xC2) at every size up to 768 x 1024: Z-drops
    (unrelated tails), end bonuses that decide between the best local end and the query end, N bases, extreme aspect ratios,
    bands that can and cannot clip a row -- against the lane-exact oracle"""
    import minimap2_amd as mm
    rng = np.random.default_rng(99)
    jobs = []

    def add(q, t, flag=None):
        q, t = q[:1024], t[:768]
        if len(q) == 0 or len(t) == 0:
            return
        tight = max(len(q), len(t)) - 1  # the smallest band that cannot bind; one less can (those jobs take the lane-exact kernel)
        w = int(rng.choice([751, -1, tight, tight, max(tight - 1, 0), 30001]))
        f = int(rng.choice([0x40, 0xC2])) if flag is None else flag
        jobs.append((q, t, w, int(rng.choice([-1, 30, 100, 400])), int(rng.choice([-1, 0, 10, 100])), f))

    for tl in (1, 2, 15, 16, 17, 63, 64, 65, 127, 128, 129, 255, 256, 257, 383, 511, 512, 513, 600, 700, 752, 753, 767, 768):
        for rep in range(4):
            t = rng.integers(0, 4, tl, dtype=np.uint8)
            if rep == 0:
                add(t.copy()[:512], t)
            elif rep == 1:
                q, _ = random_pair(rng, tl, 0.15, 0.02)
                add(q, t)
            elif rep == 2:
                add(rng.integers(0, 4, int(rng.integers(1, 1025)), dtype=np.uint8), t)  # unrelated: Z-drop or nothing to extend
            else:  # a good start, then an unrelated tail
                q, _ = random_pair(rng, tl, 0.08)
                k = max(1, len(q) // 2)
                add(np.concatenate([q[:k], rng.integers(0, 4, len(q) - k + 20, dtype=np.uint8)]), t)
    for it in range(400):
        q, t = random_pair(rng, int(rng.integers(1, 600)), float(rng.choice([0.0, 0.05, 0.12, 0.3])), float(rng.choice([0, 0, 0.03])),
                           int(rng.choice([0, 0, 0, 30, -30, 120, -120])))
        add(q, t)
    for it in range(30):  # not eligible (the band can clip): still exact
        q, t = random_pair(rng, int(rng.integers(100, 400)), 0.12)
        jobs.append((q, t, int(rng.integers(5, 80)), 400, 10, int(rng.choice([0x40, 0xC2]))))
    _run(jobs, preset)
    a, b, go, ge, go2, ge2 = PRESETS[preset]
    got = mm.ksw_extd2_batch(jobs, ts_mat(a, b, 1, 0), go, ge, go2, ge2)
    assert sum(r[1] for r in got) > 10 and sum(r[9] for r in got) > 10  # Z-drops and reach_end both occur


# ---------------------------------------------------------------------------------------------------------
# The banded gap-fill kernel (ksw_band.hip, round 6).  What the kernel ACCEPTS must be what the reference's unbanded ksw_extd2_sse gives (the mapper's call:
# align.c:810-844, a band that cannot bind); what it cannot prove goes to the wider band or to the full rectangle and must come back the same.
# ---------------------------------------------------------------------------------------------------------
def _band_delta(fn):
    import minimap2_amd as mm
    before = mm.band_counters()
    out = fn()
    after = mm.band_counters()
    return out, {k: after[k] - before[k] for k in after}


def _gap_cost(l, go, ge, go2, ge2):
    return 0 if l <= 0 else min(go + ge * l, go2 + ge2 * l)


def _edge_walkers(rng, a, go, ge, go2, ge2, W, cap=512, stretch=(120, 260)):
    """windows whose best alignment runs along ONE diagonal off the corners' own: a gap of d target bases first, a perfect (or nearly perfect) stretch,
    a gap of d query bases last.  With d on the band's last diagonal the band must still find it; one further out it must notice that it cannot -- and with
    a perfect stretch the alignment outside scores exactly the bound the acceptance test uses (ksw_band.hpp): the test is strict, so that tie is a reject."""
    jobs = []
    for upper in (True, False):
        for off in (-2, -1, 0, 1, 3):
            for mut in (0, 3):
                m = int(rng.integers(*stretch))
                # D = 0 here: c = W / 4, diagonals [-W/2, W/2 - 1]; the first diagonal outside: W/2 above, W/2 + 1 below
                d = (W // 2 if upper else W // 2 + 1) + off
                if d + m > cap:
                    continue
                core = rng.integers(0, 4, m, dtype=np.uint8)
                other = core.copy()
                for _ in range(mut):
                    p = int(rng.integers(0, m))
                    other[p] = (other[p] + 1) % 4
                g1, g2 = rng.integers(0, 4, d, dtype=np.uint8), rng.integers(0, 4, d, dtype=np.uint8)
                if upper:   # the target runs ahead: i - j = d on the stretch
                    t, q = np.concatenate([g1, core]), np.concatenate([other, g2])
                else:
                    q, t = np.concatenate([g1, core]), np.concatenate([other, g2])
                jobs.append((q, t, 30001, 400, -1, 0x08))
    return jobs


@pytest.mark.parametrize("preset", list(PRESETS))
def test_banded_gap_fill_equals_the_unbanded_reference(preset, monkeypatch):
    import minimap2_amd as mm
    a, b, go, ge, go2, ge2 = PRESETS[preset]
    mat = ts_mat(a, b, 1, 0)
    rng = np.random.default_rng(601)
    jobs = []
    for it in range(700):
        kind = it % 7
        L = int(rng.integers(1, 513))
        if kind == 0:   # unrelated sequences: low scores, nothing provable
            q, t = rng.integers(0, 4, L, dtype=np.uint8), rng.integers(0, 4, int(rng.integers(1, 513)), dtype=np.uint8)
        elif kind == 1:  # a long indel: the corners' diagonals far apart
            q, t = random_pair(rng, L, float(rng.choice([0.02, 0.12])), 0.0, int(rng.choice([40, -40, 90, -90, 130, -130, 250])))
        else:
            q, t = random_pair(rng, L, float(rng.choice([0.0, 0.03, 0.08, 0.12, 0.2, 0.35])), float(rng.choice([0, 0, 0.03])), int(rng.choice([0, 0, 0, 8, -8, 25, -25])))
        if len(q) > 512 or len(t) > 512:
            continue
        jobs.append((q, t, 30001, 400, -1, 0x08))
    for W in (128, 256):
        jobs += _edge_walkers(rng, a, go, ge, go2, ge2, W)
    have_ref = os.path.exists(reflib.REF_SO)
    want = [(reflib.ref_extd2 if have_ref else ora_extd2)(q, t, mat, go, ge, go2, ge2, w, zd, eb, fl) for (q, t, w, zd, eb, fl) in jobs]

    def run():
        return mm.ksw_extd2_batch(jobs, mat, go, ge, go2, ge2)

    got, n = _band_delta(run)
    assert got == want
    assert n["band128"] + n["band256"] > len(jobs) // 3, n  # the kernel was used ...
    first = n
    # ... and so were its ways out, when the launch classes are told to expect the impossible (every window tries the narrowest band first)
    monkeypatch.setenv("MM2AMD_BAND_RHO", "1.0")
    got, n = _band_delta(run)
    assert got == want
    assert n["band128"] > first["band128"] and n["widened"] > 0 and n["rectangle"] > 0, n
    # few persistent waves: each takes dozens of pairs, the lists are long
    monkeypatch.setenv("MM2AMD_KSW_MAX_SLOTS", "4")
    got, n = _band_delta(run)
    assert got == want
    monkeypatch.delenv("MM2AMD_BAND_RHO")
    # nothing accepted at the first attempt: every window through a list-fed launch (1: band-128 rejects try 256 diagonals; 2: straight to the rectangle)
    for mode in ("1", "2"):
        monkeypatch.setenv("MM2AMD_BAND_REJECT", mode)
        got, n = _band_delta(run)
        assert got == want, mode
        assert n["widened"] + n["rectangle"] >= n["band128"] + n["band256"], (mode, n)
    monkeypatch.delenv("MM2AMD_BAND_REJECT")
    monkeypatch.setenv("MM2AMD_NO_BAND", "1")  # the rectangles only: the A/B partner
    got, n = _band_delta(run)
    assert got == want and n["band128"] + n["band256"] == 0, n


@pytest.mark.parametrize("preset", ["ont", "swap"])
def test_banded_gap_fill_of_windows_beyond_512(preset, monkeypatch):
    """The four-set class of ksw_band.hip: 512 diagonals for windows with a side in 513..1024 (what the strip kernel computed as rectangles of two to sixteen
    256-column strips).  Accepted results against the reference's unbanded ksw_extd2_sse; walkers on and next to the band's last diagonal; the class's way out
    -- a list-fed launch of the strip kernel -- with every first attempt rejected and with few persistent waves; and the class switched off."""
    import minimap2_amd as mm
    a, b, go, ge, go2, ge2 = PRESETS[preset]
    mat = ts_mat(a, b, 1, 0)
    rng = np.random.default_rng(977)
    jobs = []
    for it in range(120):
        kind = it % 6
        L = int(rng.integers(400, 1025))
        if kind == 0:   # unrelated sequences: nothing provable
            q, t = rng.integers(0, 4, L, dtype=np.uint8), rng.integers(0, 4, int(rng.integers(513, 1025)), dtype=np.uint8)
        elif kind == 1:  # a long indel: the corners' diagonals far apart
            q, t = random_pair(rng, L, float(rng.choice([0.02, 0.12])), 0.0, int(rng.choice([90, -90, 250, -250, 400, -400, 500])))
        else:
            q, t = random_pair(rng, L, float(rng.choice([0.0, 0.03, 0.08, 0.12, 0.2, 0.35])), float(rng.choice([0, 0, 0.03])), int(rng.choice([0, 0, 0, 8, -8, 60, -60])))
        q, t = q[:1024], t[:1024]
        if len(q) <= 512 and len(t) <= 512:
            continue
        jobs.append((q, t, 30001, 400, -1, 0x08))
    for ql, tl in ((1024, 1024), (513, 1), (1, 513), (1024, 513), (513, 1024), (1024, 2), (700, 1024)):  # the corners of the class
        q, t = random_pair(rng, max(ql, tl), 0.05)
        q, t = np.resize(q, ql), np.resize(t, tl)
        jobs.append((q, t, 30001, 400, -1, 0x08))
    walkers = _edge_walkers(rng, a, go, ge, go2, ge2, 512, cap=1024, stretch=(300, 760))
    assert len(walkers) >= 12
    jobs += walkers
    have_ref = os.path.exists(reflib.REF_SO)
    want = [(reflib.ref_extd2 if have_ref else ora_extd2)(q, t, mat, go, ge, go2, ge2, w, zd, eb, fl) for (q, t, w, zd, eb, fl) in jobs]

    def run():
        return mm.ksw_extd2_batch(jobs, mat, go, ge, go2, ge2)

    monkeypatch.setenv("MM2AMD_BAND_RHO", "0.5")  # (pinned: the share the launch classes expect otherwise follows what earlier batches of the process reached)
    got, n = _band_delta(run)
    assert got == want
    assert n["band512"] > len(jobs) // 3 and n["band128"] + n["band256"] == 0, n
    monkeypatch.setenv("MM2AMD_BAND_RHO", "1.0")  # every window tries the band; the hopeless ones come back through the strip kernel's list
    got, n = _band_delta(run)
    assert got == want
    assert n["band512"] >= len(jobs) - 8 and n["rectangle_big"] > 0, n
    monkeypatch.setenv("MM2AMD_KSW_MAX_SLOTS", "4")
    got, n = _band_delta(run)
    assert got == want
    monkeypatch.setenv("MM2AMD_BAND_REJECT", "1")  # nothing accepted: every window through the list-fed strip kernel
    got, n = _band_delta(run)
    assert got == want
    assert n["rectangle_big"] == n["band512"] > 0, n
    monkeypatch.delenv("MM2AMD_BAND_REJECT")
    monkeypatch.delenv("MM2AMD_KSW_MAX_SLOTS")
    monkeypatch.delenv("MM2AMD_BAND_RHO")
    monkeypatch.setenv("MM2AMD_BAND_MAX", "512")  # the class off: the A/B partner
    got, n = _band_delta(run)
    assert got == want and n["band512"] == 0, n


@pytest.mark.parametrize("preset", ["ont", "hifi", "swap"])
def test_extension_kernel_with_the_query_across_the_lanes(preset, monkeypatch):
    """ksw_extq.hip (round 6): extension calls (KSW_EZ_EXTZ_ONLY, left extensions also RIGHT | REV_CIGAR) whose band cannot bind, classed by QUERY length (128 / 256 / 512
    positions in 2 / 4 / 8 register sets) with targets up to 2048 streaming through the lanes -- every register-set boundary on the query, targets shorter than, about twice and
    many times as long as the query (the shape of an end extension: align.c:716-718), Z-drops that fire early, late and never, end bonuses, N bases -- against the lane-exact
    oracle; then the same jobs through round 3's kernel (the target across the lanes) and through the lane-exact kernel."""
    import minimap2_amd as mm
    a, b, go, ge, go2, ge2 = PRESETS[preset]
    mat = ts_mat(a, b, 1, 0)
    rng = np.random.default_rng(905)
    jobs = []
    for ql in (1, 2, 63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 256, 257, 300, 383, 384, 385, 447, 448, 449, 500, 511, 512):
        for shape in range(4):
            q, t = random_pair(rng, ql, float(rng.choice([0.0, 0.05, 0.12, 0.3])), float(rng.choice([0, 0, 0.03])))
            q = q[:512]
            if shape == 1:    # the end-extension shape: the target about twice the query
                t = np.concatenate([t, rng.integers(0, 4, int(rng.integers(ql // 2, ql + 60)), dtype=np.uint8)])
            elif shape == 2:  # a long target: the query is used up early, the Z-drop decides
                t = np.concatenate([t, rng.integers(0, 4, int(rng.integers(500, 1500)), dtype=np.uint8)])
            elif shape == 3:  # a short one: the target ends first
                t = t[:max(1, len(t) // 3)]
            t = t[:2048]
            for flag in (0x40, 0xC2):
                w = max(len(q), len(t))  # cannot bind
                jobs.append((q, t, int(rng.choice([w, w + 7, 30001])), int(rng.choice([-1, 100, 400, 2000])), int(rng.choice([-1, 10, 40])), flag))
    monkeypatch.setenv("MM2AMD_EXT_MAX_Q", "512")  # (by default queries beyond 256 stay with the lane-exact kernel: ksw_host.cpp says why; here the eight-set class runs too)
    _run(jobs, preset)
    fast = mm.ksw_extd2_batch(jobs, mat, go, ge, go2, ge2)
    monkeypatch.delenv("MM2AMD_EXT_MAX_Q")
    assert mm.ksw_extd2_batch(jobs, mat, go, ge, go2, ge2) == fast
    monkeypatch.setenv("MM2AMD_KSW_MAX_SLOTS", "4")  # few persistent waves: each takes many pairs
    assert mm.ksw_extd2_batch(jobs, mat, go, ge, go2, ge2) == fast
    monkeypatch.delenv("MM2AMD_KSW_MAX_SLOTS")
    monkeypatch.setenv("MM2AMD_KSW_EXACT_ONLY", "1")
    assert mm.ksw_extd2_batch(jobs, mat, go, ge, go2, ge2) == fast
