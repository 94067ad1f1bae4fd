"""Seeded synthetic workloads shaped like BASELINE.json's configs, at any scale.

    python tests/synth.py ont  OUTDIR --ref-mb 2 --reads 200 [--seed 11]
    python tests/synth.py hifi OUTDIR --ref-mb 2 --reads 100

Reference: uniform i.i.d. ACGT in contigs chr1..N.  Reads: uniformly placed substrings, half reverse-complemented,
per-base error split 1/3 substitution, 1/3 insertion, 1/3 deletion.  Vectorised so that the full-size workloads
(3 Gb, 100k reads) are generated in tens of seconds."""
import argparse
import os
import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = np.array([3, 2, 1, 0], dtype=np.uint8)


def gen_reference(rng, total, n_contig):
    per = total // n_contig
    return [rng.integers(0, 4, per, dtype=np.uint8) for _ in range(n_contig)]


def mutate_read(rng, s, err):
    """s: uint8 codes 0..3; returns mutated copy."""
    n = len(s)
    r = rng.random(n)
    kind = rng.integers(0, 3, n)
    sub = (r < err) & (kind == 0)
    ins = (r < err) & (kind == 1)
    dele = (r < err) & (kind == 2)
    out = s.copy()
    out[sub] = (s[sub] + rng.integers(1, 4, int(sub.sum()), dtype=np.uint8)) % 4
    keep = ~dele
    # insertion: emit a random base before the kept base
    reps = np.where(ins, 2, 1)[keep]
    base = np.repeat(out[keep], reps)
    idx = np.cumsum(reps) - reps  # first slot of each group
    ins_kept = ins[keep]
    base[idx[ins_kept]] = rng.integers(0, 4, int(ins_kept.sum()), dtype=np.uint8)
    return base


def gen_reads(rng, contigs, n_reads, mean_len, sd_len, err, min_len=1000):
    reads = []
    lens = np.clip(rng.normal(mean_len, sd_len, n_reads).astype(np.int64), min_len, None)
    for i in range(n_reads):
        c = int(rng.integers(0, len(contigs)))
        L = int(min(lens[i], len(contigs[c])))
        st = int(rng.integers(0, len(contigs[c]) - L + 1))
        s = contigs[c][st:st + L]
        if rng.random() < 0.5:
            s = COMP[s[::-1]]
        reads.append(mutate_read(rng, s, err))
    return reads


def gen_transcripts(rng, contigs, n_reads, err, max_intron=6000, introns=None):
    """Spliced reads: 2..8 exons of 60..500 bases separated by introns (mostly 80..max_intron, some ~20 kb), concatenated.  The
    reference is edited in place so that most introns carry the canonical signals (GT..AG for a gene on the + strand, CT..AC
    for one on the - strand); a read is the transcript or its reverse complement, with per-base error `err`."""
    reads = []
    for _ in range(n_reads):
        c = int(rng.integers(0, len(contigs)))
        ctg = contigs[c]
        n_exon = int(rng.integers(2, 9))
        ex = rng.integers(60, 500, n_exon)
        it = rng.integers(80, max_intron, n_exon - 1)
        it[rng.random(n_exon - 1) < 0.1] = int(rng.integers(15000, 25000))
        span = int(ex.sum() + it.sum())
        if span + 10 > len(ctg):
            it = np.minimum(it, 300)
            span = int(ex.sum() + it.sum())
        pos = int(rng.integers(0, len(ctg) - span))
        minus = rng.random() < 0.5
        parts = []
        for k in range(n_exon):
            parts.append((pos, pos + int(ex[k])))
            pos += int(ex[k])
            if k < n_exon - 1:
                if rng.random() < 0.9:
                    ctg[pos:pos + 2] = [1, 3] if minus else [2, 3]
                    ctg[pos + int(it[k]) - 2:pos + int(it[k])] = [0, 1] if minus else [0, 2]
                if introns is not None:
                    introns.append((c, pos, pos + int(it[k]), minus))
                pos += int(it[k])
        s = np.concatenate([ctg[a:b] for a, b in parts])
        if rng.random() < 0.5:
            s = COMP[s[::-1]]
        reads.append((s, err))
    return [mutate_read(rng, s, e) for s, e in reads]  # after all edits of the reference


def make_junctions(outdir, seed=31, n_reads=60, ref_mb=1.0):
    """cDNA reads plus a junction annotation in BED6 (one intron per line, as --junc-bed reads it): most of the true introns
    (including the ones without canonical signals, where the annotation bonus decides), some with the wrong strand, some shifted
    by a few bases, duplicates, and decoys elsewhere.  Returns (ref.fa, reads.fa, junc.bed)."""
    os.makedirs(outdir, exist_ok=True)
    rng = np.random.default_rng(seed)
    contigs = gen_reference(rng, int(ref_mb * 1e6), 2)
    introns = []
    reads = gen_transcripts(rng, contigs, n_reads, 0.04, introns=introns)
    # reads with a few bases on the far side of an intron: too few to be aligned across it, they end up clipped -- unless the
    # jump annotation (-j) lets the alignment end hop over the junction (mm_jump_split)
    r3 = np.random.default_rng(seed + 2)
    for k, (c, st, en, minus) in enumerate(introns[:50]):
        ctg = contigs[c]
        few = int(r3.integers(3, 16))
        if st < 400 or en + 400 > len(ctg):
            continue
        s = np.concatenate([ctg[st - few:st], ctg[en:en + 300]]) if k % 2 == 0 else np.concatenate([ctg[st - 300:st], ctg[en:en + few]])
        if k % 5 == 0:
            s = s.copy(); s[int(r3.integers(0, len(s)))] ^= 1   # one substitution somewhere
        reads.append(COMP[s[::-1]] if k % 3 == 0 else s)
    ref, rd, bed = os.path.join(outdir, "ref.fa"), os.path.join(outdir, "reads.fa"), os.path.join(outdir, "junc.bed")
    write_fasta(ref, ["chr1", "chr2"], contigs)
    write_fasta(rd, ["read%d" % i for i in range(len(reads))], reads)
    r2 = np.random.default_rng(seed + 1)
    with open(bed, "w") as f:
        for k, (c, st, en, minus) in enumerate(introns):
            u = r2.random()
            strand = "-" if minus else "+"
            if u < 0.15:
                continue                                   # not annotated
            if u < 0.22:
                strand = "+" if minus else "-"             # annotated on the wrong strand
            if u > 0.93:
                st, en = st + int(r2.integers(-6, 7)), en + int(r2.integers(-6, 7))  # slightly off
            f.write("chr%d\t%d\t%d\tj%d\t0\t%s\n" % (c + 1, st, en, k, strand))
            if k % 11 == 0:
                f.write("chr%d\t%d\t%d\tj%d_dup\t0\t%s\n" % (c + 1, st, en, k, strand))
        for k in range(40):                                # decoys
            c = int(r2.integers(0, 2)); st = int(r2.integers(0, len(contigs[c]) - 9000)); en = st + int(r2.integers(60, 8000))
            f.write("chr%d\t%d\t%d\tdecoy%d\t0\t%s\n" % (c + 1, st, en, k, "+-"[k % 2]))
        f.write("chr1\t100\t900\tnostrand\t0\t.\n")
    return ref, rd, bed


def make_rna_pairs(outdir, seed=41, n_tx=50, pairs_per_tx=4, ref_mb=1.0):
    """Short RNA-seq read pairs (2 x 60-100 bases, FR, 0.5 % substitutions) sampled from spliced transcripts: most reads cross
    exon junctions, some by a few bases only.  Returns (ref.fa, r1.fa, r2.fa, introns.bed)."""
    os.makedirs(outdir, exist_ok=True)
    rng = np.random.default_rng(seed)
    contigs = gen_reference(rng, int(ref_mb * 1e6), 2)
    introns = []
    txs = gen_transcripts(rng, contigs, n_tx, 0.0, max_intron=3000, introns=introns)
    r1, r2 = [], []
    for tx in txs:
        if len(tx) < 260:
            continue
        for _ in range(pairs_per_tx):
            L = int(rng.integers(60, 101))
            frag = int(rng.integers(2 * L - 30, min(len(tx), 400) + 1)) if min(len(tx), 400) >= 2 * L - 30 else len(tx)
            frag = max(frag, L)
            p = int(rng.integers(0, len(tx) - frag + 1))
            a, b = tx[p:p + L].copy(), COMP[tx[p + frag - L:p + frag][::-1]].copy()
            for r in (a, b):
                sub = rng.random(len(r)) < 0.005
                r[sub] = (r[sub] + rng.integers(1, 4, int(sub.sum()), dtype=np.uint8)) % 4
            if rng.random() < 0.5:
                a, b = b, a
            r1.append(a), r2.append(b)
    ref, f1, f2, bed = (os.path.join(outdir, n) for n in ("ref.fa", "r1.fa", "r2.fa", "introns.bed"))
    write_fasta(ref, ["chr1", "chr2"], contigs)
    names = ["rp%d" % k for k in range(len(r1))]
    write_fasta(f1, [n + "/1" for n in names], r1)
    write_fasta(f2, [n + "/2" for n in names], r2)
    with open(bed, "w") as f:
        for k, (c, st, en, minus) in enumerate(introns):
            if k % 7:
                f.write("chr%d\t%d\t%d\tj%d\t0\t%s\n" % (c + 1, st, en, k, "-" if minus else "+"))
    return ref, f1, f2, bed


def make_splice_scores(ref_fa, path, seed=77, every=5):
    """A splice-score table as --spsc reads it (contig, position, strand, D|A, score): a random site about every `every` bases,
    random strand / type, scores in [-12, 20], some positions listed twice with different scores."""
    rng = np.random.default_rng(seed)
    names, lens = [], []
    for line in open(ref_fa, "rb"):
        if line.startswith(b">"):
            names.append(line[1:].strip().decode())
        else:
            lens.append(len(line.strip()))
    with open(path, "w") as f:
        for nm, ln in zip(names, lens):
            pos = np.flatnonzero(rng.random(ln) < 1.0 / every)
            strand = rng.integers(0, 2, len(pos)); typ = rng.integers(0, 2, len(pos)); sc = rng.integers(-12, 21, len(pos))
            for k in range(len(pos)):
                f.write("%s\t%d\t%s\t%s\t%d\n" % (nm, pos[k], "+-"[strand[k]], "DA"[typ[k]], sc[k]))
                if k % 17 == 0:
                    f.write("%s\t%d\t%s\t%s\t%d\n" % (nm, pos[k], "+-"[strand[k]], "DA"[typ[k]], sc[k] - 5))
    return path


def make_alt(outdir, seed=51):
    """A primary assembly with two ALT contigs (diverged copies of primary regions, one with an insertion), reads from everywhere
    and from the duplicated regions in particular, and the ALT name list.  Returns (ref.fa, reads.fa, alt.txt)."""
    os.makedirs(outdir, exist_ok=True)
    rng = np.random.default_rng(seed)
    contigs = gen_reference(rng, 2000000, 2)
    alt1 = mutate_read(rng, contigs[0][300000:380000], 0.015)
    a2 = contigs[1][500000:560000]
    alt2 = mutate_read(rng, np.concatenate([a2[:30000], rng.integers(0, 4, 2000, dtype=np.uint8), a2[30000:]]), 0.01)
    allc = contigs + [alt1, alt2]
    reads = gen_reads(rng, allc, 80, 8000, 1500, 0.10, min_len=2000)
    for i in range(40):
        src = [contigs[0][300000:380000], alt1, a2, alt2][i % 4]
        st = int(rng.integers(0, len(src) - 9000))
        s = src[st:st + 8000]
        if rng.random() < 0.5:
            s = COMP[s[::-1]]
        reads.append(mutate_read(rng, s, 0.08))
    ref, rd, alt = os.path.join(outdir, "ref.fa"), os.path.join(outdir, "reads.fa"), os.path.join(outdir, "alt.txt")
    write_fasta(ref, ["chr1", "chr2", "chr1_alt1", "chr2_alt1"], allc)
    write_fasta(rd, ["r%d" % i for i in range(len(reads))], reads)
    open(alt, "w").write("chr1_alt1\nchr2_alt1\n")
    return ref, rd, alt


def make_repeats(outdir, seed=61):
    """One contig carrying 12 identical copies of a 3 kb unit, ordinary reads, and reads that lie entirely inside a copy: their
    minimizers all occur 12 times.  Returns (ref.fa, reads.fa)."""
    os.makedirs(outdir, exist_ok=True)
    rng = np.random.default_rng(seed)
    c = gen_reference(rng, 1500000, 1)[0]
    unit = rng.integers(0, 4, 3000, dtype=np.uint8)
    starts = [100000 + 110000 * i for i in range(12)]
    for st in starts:
        c[st:st + 3000] = unit
    reads = gen_reads(rng, [c], 30, 6000, 1000, 0.08, min_len=2000)
    for i, st in enumerate(starts):
        s = c[st + 200:st + 2800]
        reads.append(mutate_read(rng, s if i % 2 else COMP[s[::-1]], 0.05))
    ref, rd = os.path.join(outdir, "ref.fa"), os.path.join(outdir, "reads.fa")
    write_fasta(ref, ["c1"], [c])
    write_fasta(rd, ["r%d" % i for i in range(len(reads))], reads)
    return ref, rd


def make_weird(outdir, seed=71):
    """Edge-case reads against a reference with low-complexity islands ((AC)n, poly-A, a 7-mer tandem): lengths 1..100 around k,
    all-N, lower case, IUPAC codes, reads spanning / inside the islands, a homopolymer, every second base N, a read that is a
    whole contig, a 250 kb reverse-strand read, a three-piece chimera, an unrelated read, duplicate names.  Returns (ref.fa, reads.fa)."""
    rng = np.random.default_rng(seed)
    contigs = gen_reference(rng, 600000, 2)
    contigs[0][100000:103000] = np.tile(np.array([0, 1], dtype=np.uint8), 1500)
    contigs[0][200000:202000] = 0
    contigs[1][50000:56000] = np.tile(np.array([2, 0, 3, 3, 0, 1, 0], dtype=np.uint8), 858)[:6000]
    ref0, ref1 = ACGT[contigs[0]].tobytes(), ACGT[contigs[1]].tobytes()
    reads = [ref1[250000:250000 + L] for L in (1, 2, 5, 14, 15, 16, 17, 29, 30, 31, 40, 60, 100)]
    reads.append(b"N" * 500)
    reads.append(ref0[10000:12000].lower())
    iu = bytearray(ref0[20000:23000])
    for p in range(0, 3000, 97):
        iu[p] = b"RYKMSWBDHVN"[p % 11]
    reads.append(bytes(iu))
    reads += [ref0[99000:104000], ref0[100500:102500], ref0[199000:203000], b"A" * 3000, ref1[49000:57000], ref1[51000:55000]]
    half = bytearray(ref0[250000:254000])
    for p in range(0, 4000, 2):
        half[p] = ord("N")
    reads.append(bytes(half))
    reads.append(ref1)
    reads.append(ACGT[COMP[contigs[0][::-1]]].tobytes()[:250000])
    reads.append(ref0[150000:155000] + ref1[100000:105000] + ref0[160000:165000])
    reads.append(ACGT[rng.integers(0, 4, 8000, dtype=np.uint8)].tobytes())
    os.makedirs(outdir, exist_ok=True)
    ref, rd = os.path.join(outdir, "ref.fa"), os.path.join(outdir, "reads.fa")
    write_fasta(ref, ["c1", "c2"], contigs)
    with open(rd, "wb") as f:
        for i, s in enumerate(reads):
            f.write(b">w%d\n" % i + s + b"\n")
        f.write(b">dup\n" + ref0[5000:9000] + b"\n>dup\n" + ref0[5000:9000] + b"\n")
    return ref, rd


def make_palindromes(outdir, seed=83):
    """A reference with islands whose k-mers equal their own reverse complement for even k -- (AT)n, (ACGT)n, (AATT)n: such k-mers are
    skipped by mm_sketch without consuming a window slot (sketch.c:108), so a lane of the sketch kernels that starts inside an island
    cannot warm its window up from a fixed stretch -- and reads inside, across and next to the islands.  Returns (ref.fa, reads.fa)."""
    os.makedirs(outdir, exist_ok=True)
    rng = np.random.default_rng(seed)
    c = gen_reference(rng, 300000, 1)[0]
    units = [np.array(u, dtype=np.uint8) for u in ([0, 3], [0, 1, 2, 3], [0, 0, 3, 3], [1, 2])]
    islands = []
    for i, L in enumerate((600, 1500, 3000, 5000, 900, 2500, 150, 40)):
        st = 20000 + 33000 * i
        c[st:st + L] = np.tile(units[i % 4], L)[:L]
        islands.append((st, L))
    reads = gen_reads(rng, [c], 25, 5000, 1500, 0.06, min_len=1500)
    for st, L in islands:
        for a, b in ((st - 1500, st + L + 1500), (st + L // 3, st + L + 2500), (st - 2500, st + 2 * L // 3), (st + 5, st + L - 5)):
            seg = c[max(a, 0):b]
            reads.append(mutate_read(rng, seg if len(reads) % 2 else COMP[seg[::-1]], 0.04))
    ref, rd = os.path.join(outdir, "ref.fa"), os.path.join(outdir, "reads.fa")
    write_fasta(ref, ["c1"], [c])
    write_fasta(rd, ["r%d" % i for i in range(len(reads))], reads)
    return ref, rd


def make_tandem_reads(outdir, seed=91, n_reads=40, mean=4000, err=0.1, genome=400000):
    """Noisy reads over a reference peppered with short tandem repeats and homopolymer runs (every ~150 bases a run of 4..40 copies of a
    1..6-mer): the indels of their alignments fall into repeats, where mm_fix_cigar (align.c:105-181) left-aligns them as far as the
    repeat goes -- often through the whole match before them (empty operations, merged neighbours), one shift feeding the next.
    Returns (ref.fa, reads.fa)."""
    rng = np.random.default_rng(seed)
    contigs = gen_reference(rng, genome, 2)
    for c in contigs:
        p = int(rng.integers(50, 200))
        while p + 300 < len(c):
            unit = rng.integers(0, 4, int(rng.integers(1, 7)), dtype=np.uint8)
            L = int(len(unit) * rng.integers(4, 41))
            c[p:p + L] = np.tile(unit, L // len(unit) + 1)[:L]
            p += L + int(rng.integers(60, 240))
    reads = gen_reads(rng, contigs, n_reads, mean, mean // 4, err)
    os.makedirs(outdir, exist_ok=True)
    ref, rd = os.path.join(outdir, "ref.fa"), os.path.join(outdir, "reads.fa")
    write_fasta(ref, ["c1", "c2"], contigs)
    write_fasta(rd, ["rp%d" % i for i in range(n_reads)], reads)
    return ref, rd


def make_overlaps(outdir, seed=81, n_reads=60, genome=120000):
    """An all-vs-all read set: noisy reads drawn densely from a small genome (every read overlaps several others, on either
    strand), names in an order unrelated to position, two reads sharing one name, one read contained in another, a read
    that is an exact copy of another under a different name, and a read with an internal duplication.  The same file is target and query.  Returns reads.fa."""
    rng = np.random.default_rng(seed)
    g = gen_reference(rng, genome, 1)
    reads = gen_reads(rng, g, n_reads, 8000, 2000, 0.06, min_len=2500)
    names = ["rd%03d" % i for i in rng.permutation(len(reads))]
    names[7] = names[3]                      # duplicate name, different sequences
    reads.append(reads[5][1000:3500].copy()) # contained read
    names.append("inner")
    reads.append(reads[9].copy())            # identical sequence, another name
    names.append("twin")
    u = reads[11][500:2300]                  # a read with an internal duplication: off-diagonal hits on itself (MM_SEED_SELF)
    reads.append(np.concatenate([reads[11][:2300], rng.integers(0, 4, 900, dtype=np.uint8), mutate_read(rng, u, 0.03), reads[11][2300:4000]]))
    names.append("selfdup")
    os.makedirs(outdir, exist_ok=True)
    fa = os.path.join(outdir, "ovl.fa")
    write_fasta(fa, names, reads)
    return fa


def make_short(outdir, seed=91, n_reads=400, genome=400000):
    """Single-end short reads (100-250 bp, ~1 % substitutions, occasional small indels) on both strands of a two-contig
    reference that carries a few exact duplications (2 kb units copied 3-6 times) and short tandem arrays (a 40-mer repeated
    6 times: reads across them contain the same minimizer more than once, which is what makes the order of equal index hits
    observable).  Also: reads with a 15 bp deletion / insertion in the middle, reads lying entirely inside a duplicated unit, a
    read with Ns, reads too short to seed.  Returns (ref.fa, reads.fa)."""
    rng = np.random.default_rng(seed)
    contigs = gen_reference(rng, genome, 2)
    for c in contigs:
        for _ in range(3):
            unit = rng.integers(0, 4, 2000, dtype=np.uint8)
            for _ in range(int(rng.integers(3, 7))):
                st = int(rng.integers(0, len(c) - 2000))
                c[st:st + 2000] = unit
        tand = []
        for _ in range(12):
            st = int(rng.integers(1000, len(c) - 1000))
            c[st:st + 240] = np.tile(rng.integers(0, 4, 40, dtype=np.uint8), 6)
            tand.append(st)
        c_tand = tand
    reads = []
    def take(ci, st, ln):
        s = contigs[ci][st:st + ln].copy()
        return s
    for i in range(n_reads):
        ci = int(rng.integers(0, 2))
        ln = int(rng.integers(100, 251))
        st = int(rng.integers(0, len(contigs[ci]) - ln))
        if i % 10 == 0:  # across a tandem array of the second contig
            ci, st = 1, c_tand[(i // 10) % len(c_tand)] - int(rng.integers(20, 120))
        s = take(ci, st, ln)
        sub = rng.random(ln) < 0.01
        s[sub] = (s[sub] + rng.integers(1, 4, int(sub.sum()), dtype=np.uint8)) % 4
        if i % 7 == 3:
            m = ln // 2
            s = np.concatenate([s[:m], s[m + 15:]]) if i % 2 else np.concatenate([s[:m], rng.integers(0, 4, 15, dtype=np.uint8), s[m:]])
        elif i % 9 == 4:
            m = int(rng.integers(20, ln - 20))
            s = np.concatenate([s[:m], s[m + 1:]])
        if rng.random() < 0.5:
            s = COMP[s[::-1]]
        reads.append(ACGT[s].tobytes())
    nn = bytearray(reads[5]); nn[40:44] = b"NNNN"; reads.append(bytes(nn))
    reads += [reads[8][:20], reads[9][:30], b"ACGT" * 30]
    os.makedirs(outdir, exist_ok=True)
    ref, rd = os.path.join(outdir, "ref.fa"), os.path.join(outdir, "reads.fa")
    write_fasta(ref, ["s1", "s2"], contigs)
    with open(rd, "wb") as f:
        for i, s in enumerate(reads):
            f.write(b">sr%d\n" % i + s + b"\n")
    return ref, rd


def make_pairs(outdir, seed=95, n_pairs=300, genome=400000):
    """Paired-end reads (FR, 2 x 100-150 bp, inserts 250-700 bp, ~1 % substitutions, some small indels) from the reference of
    make_short (duplications, tandem arrays).  Also: pairs whose mates overlap or contain each other, pairs with one unmappable
    mate, mates on different contigs, a pair too far apart, a pair in the wrong orientation, mates of 30 bp.  Returns
    (ref.fa, r1.fa, r2.fa, interleaved.fa)."""
    ref, _ = make_short(outdir, seed=seed, n_reads=12, genome=genome)
    rng = np.random.default_rng(seed + 1000)
    contigs = []
    cur = []
    for line in open(ref, "rb"):
        if line.startswith(b">"):
            continue
        contigs.append(np.frombuffer(line.strip(), dtype=np.uint8))
    lut = np.zeros(256, dtype=np.uint8)
    lut[ord("C")], lut[ord("G")], lut[ord("T")] = 1, 2, 3
    contigs = [lut[c] for c in contigs]

    def noisy(s, k):
        s = s.copy()
        sub = rng.random(len(s)) < 0.01
        s[sub] = (s[sub] + rng.integers(1, 4, int(sub.sum()), dtype=np.uint8)) % 4
        if k % 8 == 5 and len(s) > 60:
            m = len(s) // 2
            s = np.concatenate([s[:m], s[m + 6:]]) if k % 16 == 5 else np.concatenate([s[:m], rng.integers(0, 4, 6, dtype=np.uint8), s[m:]])
        return s

    r1, r2 = [], []
    for k in range(n_pairs):
        ci = int(rng.integers(0, 2))
        c = contigs[ci]
        l1, l2 = int(rng.integers(100, 151)), int(rng.integers(100, 151))
        ins = int(rng.integers(250, 701))
        if k % 13 == 3: ins = int(rng.integers(80, 200))      # overlapping mates
        if k % 17 == 4: ins = 5000                            # too far apart
        if k % 29 == 7: l1 = l2 = 30
        ins = max(ins, l1, l2)
        st = int(rng.integers(0, len(c) - ins))
        a = c[st:st + l1]
        b = COMP[c[st + ins - l2:st + ins][::-1]]
        if k % 19 == 6:                                       # mate from the other contig
            o = contigs[1 - ci]; so = int(rng.integers(0, len(o) - l2)); b = COMP[o[so:so + l2][::-1]]
        if k % 23 == 8: b = rng.integers(0, 4, l2, dtype=np.uint8)  # unmappable mate
        if k % 31 == 9: b = COMP[b[::-1]]                     # wrong orientation (FF)
        a, b = noisy(a, k), noisy(b, k + 3)
        if rng.random() < 0.5:                                # the fragment comes from the other strand: mates swap roles
            a, b = b, a
        r1.append(a), r2.append(b)
    f1, f2, fi = (os.path.join(outdir, n) for n in ("r1.fa", "r2.fa", "inter.fa"))
    r1 = [ACGT[a].tobytes() for a in r1]
    r2 = [ACGT[b].tobytes() for b in r2]
    # odd mates: a 5-base read, an all-N read, a homopolymer, an 18-base read, lower case, N-rich, a doubled read
    k = min(40, n_pairs - 1)
    odd = [(b"ACGTA", r2[k]), (r1[k + 1 - 1], b"N" * 120), (b"A" * 100, r2[k - 2]), (r1[k - 3], r2[k - 3][:18]), (r1[k - 4].lower(), r2[k - 4]),
           (r1[k - 5], b"ACGTNNNNACGT" * 10), (r1[k - 6] + r1[k - 6], r2[k - 6])]
    for a, b in odd:
        r1.append(a), r2.append(b)
    names = ["pe%d" % k for k in range(len(r1))]
    for path, rr, suffix in ((f1, r1, b"/1"), (f2, r2, b"/2")):
        with open(path, "wb") as f:
            for n, s in zip(names, rr):
                f.write(b">" + n.encode() + suffix + b"\n" + s + b"\n")
    with open(fi, "wb") as f:
        for n, a, b in zip(names, r1, r2):
            f.write(b">" + n.encode() + b"/1\n" + a + b"\n>" + n.encode() + b"/2\n" + b + b"\n")
        f.write(b">lonely\n" + r1[0] + b"\n")   # a single read among the pairs
    return ref, f1, f2, fi


def write_fasta(path, names, seqs, width=0):
    with open(path, "wb") as f:
        for nm, s in zip(names, seqs):
            f.write(b">" + nm.encode() + b"\n")
            f.write(ACGT[s].tobytes())
            f.write(b"\n")


PROFILES = {"ont": (10000, 1000, 0.12), "hifi": (15000, 1500, 0.005), "cdna": (0, 0, 0.04)}


def make(kind, outdir, ref_mb, n_reads, seed=11, n_contig=None):
    os.makedirs(outdir, exist_ok=True)
    rng = np.random.default_rng(seed)
    total = int(ref_mb * 1e6)
    if n_contig is None:
        n_contig = max(1, min(24, total // 1000000))
    contigs = gen_reference(rng, total, n_contig)
    mean, sd, err = PROFILES[kind]
    reads = gen_transcripts(rng, contigs, n_reads, err) if kind == "cdna" else gen_reads(rng, contigs, n_reads, mean, sd, err)
    ref = os.path.join(outdir, "ref.fa")
    rd = os.path.join(outdir, "reads.fa")
    write_fasta(ref, ["chr%d" % (i + 1) for i in range(n_contig)], contigs)
    write_fasta(rd, ["read%d" % i for i in range(n_reads)], reads)
    return ref, rd, contigs, reads


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("kind", choices=list(PROFILES))
    ap.add_argument("outdir")
    ap.add_argument("--ref-mb", type=float, default=2.0)
    ap.add_argument("--reads", type=int, default=200)
    ap.add_argument("--seed", type=int, default=11)
    a = ap.parse_args()
    ref, rd, _, _ = make(a.kind, a.outdir, a.ref_mb, a.reads, a.seed)
    print(ref, rd)


def gen_duplicated_reference(rng, n_elem=40, elem_len=6000, div=0.03):
    """a contig in which every stretch of elem_len bases exists twice (the second copy diverged by `div` and separated by random spacers):
    a long read from it chains on both copies, so it has more than one chain and takes the long-join re-chaining of map.c:283-292"""
    parts = [rng.integers(0, 4, elem_len, dtype=np.uint8) for _ in range(n_elem)]
    for c in range(n_elem):
        e = parts[c].copy()
        mut = rng.random(len(e)) < div
        e[mut] = (e[mut] + rng.integers(1, 4, int(mut.sum()), dtype=np.uint8)) % 4
        parts.append(np.concatenate([rng.integers(0, 4, int(rng.integers(100, 2000)), dtype=np.uint8), e]))
    return np.concatenate(parts)


def gen_tandem_reads(rng, contigs, n, mean, err, dup_len=400):
    """reads with a stretch of dup_len bases repeated in tandem (a duplication the reference does not have): the two copies' minimizers hit
    the same reference positions, so the read's anchor list has pairs of equal keys and the sort's tie order becomes observable (ksort.h:101-151)"""
    out = []
    for r in gen_reads(rng, contigs, n, mean, mean // 5, 0.0):
        if len(r) > 3 * dup_len:
            k = int(rng.integers(dup_len, len(r) - 2 * dup_len))
            r = np.concatenate([r[:k + dup_len], r[k:]])
        sub = rng.random(len(r)) < err
        r = r.copy()
        r[sub] = (r[sub] + rng.integers(1, 4, int(sub.sum()), dtype=np.uint8)) % 4
        out.append(r)
    return out


def gen_sv_reads(rng, contig, n_each):
    """Reads with what real long reads have and single-base error models do not: (1) insertions and deletions of 30-160 bases, alone and in clusters a few
    dozen bases apart (mm_filter_bad_seeds / _alt, align.c:447-525: seeds between such gaps are dropped, clusters bridged by one long window); (2) a
    deletion or an insertion of 6-9 kb (beyond the chaining gap, within the long-join bandwidth); (3) one of 24-60 kb (beyond it: two hits on one strand,
    each one's extension bounded by the other's seeds, align.c:706-767); (4) reads at the very ends of the sequence; (5) short error-free reads.  codes."""
    c, L = contig, len(contig)
    rnd_seq = lambda k: rng.integers(0, 4, k, dtype=np.uint8)
    reads = []
    for i in range(n_each):
        a = int(rng.integers(20000, L - 60000))
        seg, parts, p = c[a:a + 14000], [], 0
        for _ in range(int(rng.integers(2, 6))):
            step = int(rng.integers(150, 2500)); parts.append(seg[p:p + step]); p += step
            k = int(rng.integers(30, 160))
            if rng.random() < 0.5: p += k
            else: parts.append(rnd_seq(k))
            if rng.random() < 0.6:
                step = int(rng.integers(20, 120)); parts.append(seg[p:p + step]); p += step
                k = int(rng.integers(30, 120))
                if rng.random() < 0.5: p += k
                else: parts.append(rnd_seq(k))
        parts.append(seg[p:])
        reads.append(np.concatenate(parts))
        b = int(rng.integers(20000, L - 60000))
        if i % 2 == 0: reads.append(np.concatenate([c[b:b + 5000], c[b + 5000 + int(rng.integers(6000, 9000)):][:5000]]))
        else: reads.append(np.concatenate([c[b:b + 5000], rnd_seq(int(rng.integers(6000, 8000))), c[b + 5000:b + 10000]]))
        d = int(rng.integers(20000, L - 120000))
        reads.append(np.concatenate([c[d:d + 6000], c[d + 6000 + int(rng.integers(30000, 60000)):][:6000]]))
        reads.append(np.concatenate([c[d:d + 5000], rnd_seq(int(rng.integers(24000, 30000))), c[d + 5000:d + 10000]]))
        reads.append(c[:int(rng.integers(6000, 9000))].copy())
        reads.append(c[L - int(rng.integers(6000, 9000)):].copy())
    out = []
    for k, r in enumerate(reads):
        r = mutate_read(rng, r, 0.05)
        out.append(COMP[r[::-1]] if k % 3 == 1 else r)
    for k in range(4 * n_each):
        e = int(rng.integers(1000, L - 5000))
        r = c[e:e + int(rng.integers(1500, 3000))].copy()
        out.append(COMP[r[::-1]] if k % 2 else r)
    return out


def gen_inverted_copy_case(rng, n_reads):
    """A 12 kb segment with partial INVERTED copies elsewhere (7 kb on the same sequence, 6 kb on the other): reads over the segment get a secondary chain on
    the other strand that the score ratio would drop and mm_select_sub's strand rule retains (hit.c:255-281), to be judged by its divergence later
    (mm_filter_strand_retained, hit.c:283-299).  Returns (contigs, reads) as codes."""
    contigs = gen_reference(rng, 600000, 2)
    c, c2 = contigs[0].copy(), contigs[1].copy()
    S = c[100000:112000].copy()
    c[200000:207000] = COMP[S[:7000][::-1]]
    c2[50000:56000] = COMP[S[3000:9000][::-1]]
    reads = []
    for k in range(n_reads):
        st = 100000 + int(rng.integers(-1500, 1500))
        r = mutate_read(rng, c[st:st + int(rng.integers(11000, 14000))], 0.04)
        reads.append(COMP[r[::-1]] if k % 2 else r)
    return [c, c2], reads
