"""Shared by the emulator's CPU case and the GPU case: the order of a read's anchors where keys are equal, against the reference's own
--print-seeds (map.c:255-260), for reads whose partitions by reference sequence have many buckets."""
import os
import subprocess

import numpy as np

import synth


def many_bucket_tie_case(dropin, ref_bin, tmp, seed, n_reads, mean_len, modes=None):
    """Reads with a tandem duplication (pairs of equal anchor keys) against 24 reference sequences with short minimizers (-k 11 -w 5): a few
    thousand anchors per read spread over all sequences and both strands, so the replay of the reference's unstable sort (ksort.h:101-151)
    partitions ranges of hundreds to thousands of elements into 24 (by sequence) and up to 256 (by position byte) buckets.  Returns the
    reference's SD lines and, per mode of the replay (default: reads with equal keys replayed together, partitions walked over tapes, a workgroup per strand;
    one workgroup for both strands; the one-thread walk; the replay inside the sorting launch), the SD lines of the device path and its stderr."""
    rng = np.random.default_rng(seed)
    contigs = synth.gen_reference(rng, 2400000, 24)
    reads = synth.gen_tandem_reads(rng, contigs, n_reads, mean_len, 0.06)
    ref_fa, rd_fa = os.path.join(tmp, "ref.fa"), os.path.join(tmp, "reads.fa")
    open(ref_fa, "w").write("".join(">chr%d\n%s\n" % (i + 1, synth.ACGT[c].tobytes().decode()) for i, c in enumerate(contigs)))
    open(rd_fa, "w").write("".join(">tan%d\n%s\n" % (i, synth.ACGT[r].tobytes().decode()) for i, r in enumerate(reads)))
    args = ["-x", "map-ont", "-k", "11", "-w", "5", "-c"]
    want = subprocess.run([ref_bin] + args + ["-t", "1", "--print-seeds", ref_fa, rd_fa], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, check=True).stderr.decode()
    want_sd = [l for l in want.split("\n") if l.startswith("SD\t")]
    got = {}
    for name, env in (("tapes", {}), ("one_workgroup", {"MM2AMD_TIE_NO_STRAND_SPLIT": "1"}), ("walk", {"MM2AMD_NO_TAPE_WALK": "1"}), ("inline", {"MM2AMD_TIE_REPLAY_INLINE": "1"})):
        if modes is not None and name not in modes:
            continue
        dump = os.path.join(tmp, "seeds_%s.txt" % name)
        p = subprocess.run([dropin] + args + ["-t", "2", ref_fa, rd_fa], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
                           env=dict(os.environ, MM2AMD_SEED_DUMP=dump, MM2AMD_TWO_BUCKET_TRACE="1", **env))
        assert p.returncode == 0, p.stderr.decode()[-1000:]
        got[name] = ([l for l in open(dump).read().split("\n") if l.startswith("SD\t")], p.stderr.decode())
    return want_sd, got
