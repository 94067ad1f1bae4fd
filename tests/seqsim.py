"""Seeded synthetic sequence generators shared by tests and bench (nt4 codes 0..3, 4 = N)."""
import numpy as np


def mutate(rng, seq, err, n_frac=0.0):
    """Return a copy of seq (uint8 nt4 array) with per-base error `err` split 1/3 sub, 1/3 ins, 1/3 del."""
    out = []
    r = rng.random(len(seq))
    kind = rng.integers(0, 3, len(seq))
    newb = rng.integers(0, 4, len(seq))
    for i in range(len(seq)):
        if r[i] < err:
            if kind[i] == 0:
                out.append((seq[i] + 1 + newb[i] % 3) % 4 if seq[i] < 4 else newb[i])
            elif kind[i] == 1:
                out.append(newb[i]); out.append(seq[i])
            # deletion: emit nothing
        else:
            out.append(seq[i])
    out = np.array(out, dtype=np.uint8)
    if n_frac > 0 and len(out):
        out[rng.random(len(out)) < n_frac] = 4
    return out


def random_pair(rng, qlen, err=0.12, n_frac=0.0, indel=0):
    """A target of about qlen bases and a noisy copy; optionally with one long indel in the middle."""
    t = rng.integers(0, 4, qlen, dtype=np.uint8)
    if n_frac > 0:
        t[rng.random(qlen) < n_frac] = 4
    src = t
    if indel > 0 and qlen > 4:
        cut = qlen // 2
        src = np.concatenate([t[:cut], t[min(qlen, cut + indel):]])
    elif indel < 0 and qlen > 4:
        cut = qlen // 2
        src = np.concatenate([t[:cut], rng.integers(0, 4, -indel, dtype=np.uint8), t[cut:]])
    q = mutate(rng, src, err, n_frac)
    if len(q) == 0:
        q = np.array([0], dtype=np.uint8)
    return q, t


def spliced_pair(rng, n_exon, err, with_signals=True, exon=(20, 160), intron=(30, 700)):
    """a target made of exons and introns (GT..AG or CT..AC at the intron ends), and the spliced, mutated query"""
    exons = [rng.integers(0, 4, int(rng.integers(exon[0], exon[1])), dtype=np.uint8) for _ in range(n_exon)]
    t = [exons[0]]
    strand = int(rng.integers(0, 2))
    for ex in exons[1:]:
        iv = rng.integers(0, 4, int(rng.integers(intron[0], intron[1])), dtype=np.uint8)
        if with_signals and rng.random() < 0.8:
            if strand == 0:
                iv[:2] = [2, 3]; iv[-2:] = [0, 2]
            else:
                iv[:2] = [1, 3]; iv[-2:] = [0, 1]
        t += [iv, ex]
    q = mutate(rng, np.concatenate(exons), err)
    if len(q) == 0:
        q = np.array([0], dtype=np.uint8)
    return q, np.concatenate(t)
