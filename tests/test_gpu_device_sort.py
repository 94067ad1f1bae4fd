"""The index build's device-wide primitives (minimap2_amd/csrc/device_sort.hip) through the kernel-level C ABI: the stable LSD radix sort of
(key, value) pairs against numpy's stable argsort on the masked key -- the order radix_sort_128x (ksort.h:101-151; index.c:236) produces for
(hash, position) pairs whose positions arrive ascending -- and the exclusive prefix sum against numpy's cumsum.  Sizes straddle the tile
(4096 pairs), the chunk of tiles whose column sums are kept (128 tiles) and, on the GPU, the three levels of the prefix sum (4096^2 entries);
key distributions: uniform, a handful of distinct keys, all equal, already sorted, reversed."""
import numpy as np
import pytest

import minimap2_amd as mm

pytestmark = pytest.mark.gpu


def want_sorted(keys, vals, bits):
    mask = np.uint64((1 << bits) - 1) if bits < 64 else np.uint64(0xFFFFFFFFFFFFFFFF)
    order = np.argsort(keys & mask, kind="stable")
    return keys[order], vals[order]


def check(keys, vals, bits):
    k, v = mm.sort_pairs_u64(keys, vals, bits)
    wk, wv = want_sorted(keys, vals, bits)
    assert np.array_equal(k, wk) and np.array_equal(v, wv), "n=%d bits=%d" % (keys.size, bits)


SIZES = [0, 1, 2, 63, 64, 65, 1023, 4095, 4096, 4097, 128 * 4096 - 1, 128 * 4096, 128 * 4096 + 1, 1000003, 129 * 4096 * 3 + 5]


@pytest.mark.parametrize("bits", [1, 5, 8, 9, 30, 38, 56, 64])
def test_sort_uniform_keys(bits):
    rng = np.random.default_rng(bits)
    for n in SIZES:
        keys = rng.integers(0, 1 << 63, n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, n, dtype=np.uint64)  # all 64 bits in use
        check(keys, np.arange(n, dtype=np.uint64), bits)


def test_sort_is_stable_on_few_distinct_keys():
    rng = np.random.default_rng(7)
    for n in SIZES:
        for distinct in (1, 2, 3, 255, 257):
            keys = rng.integers(0, distinct, n, dtype=np.uint64) * np.uint64(0x0101010101010101 >> 8)  # the same digit pattern in every pass
            check(keys, rng.integers(0, 1 << 62, n, dtype=np.uint64), 56)


def test_sort_sorted_and_reversed_input():
    for n in SIZES:
        keys = np.arange(n, dtype=np.uint64) * np.uint64(2654435761)
        keys &= np.uint64((1 << 30) - 1)
        s = np.sort(keys)
        check(s, np.arange(n, dtype=np.uint64), 30)
        check(s[::-1].copy(), np.arange(n, dtype=np.uint64), 30)


def test_sort_minimizer_like_pairs():
    """what the index build sorts: 2k-bit hashes with duplicates (repeats), positions ascending -- after the sort positions ascend within a hash"""
    rng = np.random.default_rng(11)
    n = 3000000
    pool = rng.integers(0, 1 << 30, n // 3 + 1, dtype=np.uint64)
    keys = pool[rng.integers(0, pool.size, n)]
    vals = (np.arange(n, dtype=np.uint64) << np.uint64(1)) | rng.integers(0, 2, n, dtype=np.uint64)
    k, v = mm.sort_pairs_u64(keys, vals, 30)
    wk, wv = want_sorted(keys, vals, 30)
    assert np.array_equal(k, wk) and np.array_equal(v, wv)
    same = k[1:] == k[:-1]
    assert np.all(v[1:][same] > v[:-1][same])


def test_exclusive_sum():
    rng = np.random.default_rng(5)
    for n in [0, 1, 2, 255, 256, 257, 4095, 4096, 4097, 1000003, 4096 * 4096, 4096 * 4096 + 4097]:
        a = rng.integers(0, 1 << 32 if n < 100000 else 9, n, dtype=np.uint64).astype(np.uint32)  # (small sizes also wrap around 2^32)
        got = mm.exclusive_sum_u32(a)
        want = np.concatenate([[0], np.cumsum(a.astype(np.uint64))]).astype(np.uint64) & np.uint64(0xFFFFFFFF)
        assert np.array_equal(got.astype(np.uint64), want), n


@pytest.mark.timeout(900)
def test_sort_full_size_properties():
    """At the index build's own scale (hundreds of millions of pairs: tens of thousands of tiles, hundreds of chunks) numpy's argsort is too slow to be
    the checker; size-independent properties are not: the masked keys come out non-decreasing, every output pair is an input pair
    (keys_in[vals_out] == keys_out, the values being the input indices), and within equal keys the indices increase strictly -- stability, and with
    the gather identity it also makes the output a permutation of the input."""
    import os
    n = int(os.environ.get("MM2AMD_SORT_FULL_N", 3000000 if os.environ.get("MM2AMD_EMU") == "1" else 200000000))  # (MM2AMD_SORT_FULL_N=300000000 on the emulator: 73 k tiles, 573 chunks, passes in 4.5 minutes)
    bits = 30
    rng = np.random.default_rng(17)
    keys = rng.integers(0, 1 << 62, n, dtype=np.uint64)
    m = keys[3::7].size
    keys[0:7 * m:7] = keys[3::7]  # plenty of equal keys
    k, v = mm.sort_pairs_u64(keys, np.arange(n, dtype=np.uint64), bits)
    mask = np.uint64((1 << bits) - 1)
    km = k & mask
    assert np.all(km[1:] >= km[:-1])
    assert np.array_equal(keys[v], k)
    same = km[1:] == km[:-1]
    assert np.all(v[1:][same] > v[:-1][same])
    assert int(same.sum()) > n // 10
