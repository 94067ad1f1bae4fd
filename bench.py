#!/usr/bin/env python
"""bench.py -- aligned Gbases/s of the seed-chain-extend hot path on MI355X (BASELINE.json: map-ont, ~10 kb reads, -a).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--ref-mb 3000] [--reads 100000] [--scaling strong|weak]

One "step" = one mini-batch of --reads synthetic ONT-like reads through the reference's three pipeline steps (map.c:541-643) as this library
replaces them: hand-over of the reads' host buffers (mm_gpu_batch_stage_queued: pinned-memory packing + H2D, run beside the mapping
of the batch before), the hot path (mm_gpu_map_staged: encode -> sketch -> seed -> sort -> chain -> extend -> hits), the output stage
(mm_gpu_format_batch_view: SAM text of the batch, run beside the mapping of the batch after).  ALL of it is inside the clock; `value` is
whole-job throughput with the hand-over and the formatting in it.  config.resident_gbases_per_s is the mapping call alone on a batch that is
already resident in HBM (what rounds 1-2 reported).  The workload is BASELINE.json configs[1]: uniform-random reference of --ref-mb megabases in 24 contigs, reads ~N(10 kb, 1 kb) with 12 %
error (1/3 substitution, 1/3 insertion, 1/3 deletion), preset map-ont, CIGAR output.  N > 1 (one process per GPU under
torch.distributed.run): every rank builds the same index replica; with --scaling strong (the default: BASELINE.json configs[2],
"same workload sharded across 8 GPUs") all ranks generate the SAME batch and each maps its contiguous, base-balanced share
(minimap2_amd/shard.py: split_by_bases); with --scaling weak every rank maps a batch of --reads of its own.  The packed hit
records are gathered to rank 0 over RCCL inside the timed region (pinned buffers), and every rank formats the SAM text of its own shard.

Rank 0 prints ONE JSON line.  "roofline" is the dominant kernel's algorithmic bytes / its HIP-event time on the launch
stream; "cpu_baseline" is the UNMODIFIED reference's mm_map on all host cores over a bounded sample of the same batch
against the same index contents (oracle/_ref; N=1 only)."""
import argparse
import math
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
# The mapper drives eight lanes, each with a stream for the register-resident kernels and one for the lane-exact ones, plus the hand-over's stream.
# The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues -- 4 by default -- and launches that share a queue run
# one after the other: with 4, on average 3.4 kernels were in flight and the seeding kernels of one lane waited behind the DP kernels of another
# (profiles/r04: 1.85 -> 2.01 Gbases/s with 16).  Read when the runtime initialises, i.e. before the first HIP call of the process: set here, before
# torch touches the device; libmm2amd.so does the same for processes whose first HIP call is its own (capi_common.cpp); INTEGRATION.md section 5.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ.setdefault("MM2AMD_MALLOPT", "1")  # the bench owns its process: let the library keep freed host memory
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def gen_reference(torch, dev, seed, total, n_contig):
    """uniform i.i.d. ACGT; returns (flat uint8 code tensor on dev, contig length)"""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    per = total // n_contig
    return torch.randint(0, 4, (per * n_contig,), dtype=torch.uint8, device=dev, generator=g), per


def reference_ascii(torch, dev, codes, per, n_contig):
    """the contigs as ASCII byte strings (what the index builder takes)"""
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    asc = lut[codes.long()] if codes.numel() < (1 << 28) else torch.cat([lut[c.long()] for c in codes.split(1 << 28)])
    host = asc.cpu().numpy()
    del asc
    return [host[i * per:(i + 1) * per].tobytes() for i in range(n_contig)]


def plant_genes(torch, dev, seed, codes, per, n_contig, n_genes):
    """SURVEY.md 8(d) SPL: synthetic genes of 4-10 exons (80-400 bp) separated by introns of 200 bp - 50 kb (log-uniform) that
    carry GT..AG (gene on the + strand) or CT..AC (- strand) at their ends; the signals are written into the reference.
    Returns the gene table the read generator samples from."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    E = 10
    n_exon = torch.randint(4, E + 1, (n_genes,), device=dev, generator=g)
    col = torch.arange(E, device=dev)[None, :]
    ex = torch.randint(80, 401, (n_genes, E), device=dev, generator=g) * (col < n_exon[:, None])
    lo, hi = math.log(200.0), math.log(50000.0)
    it = torch.exp(torch.rand(n_genes, E, device=dev, generator=g) * (hi - lo) + lo).long() * (col < (n_exon - 1)[:, None])
    step = ex + it
    span = step.sum(1)
    cid = torch.randint(0, n_contig, (n_genes,), device=dev, generator=g)
    st = (torch.rand(n_genes, device=dev, generator=g, dtype=torch.float64) * (per - span - 1).double()).long()
    ex_st = cid[:, None] * per + st[:, None] + torch.cumsum(step, 1) - step  # flat start of every exon
    minus = torch.rand(n_genes, device=dev, generator=g) < 0.5
    has = it > 0
    i0 = (ex_st + ex)[has]
    i1 = i0 + it[has]
    first = torch.where(minus[:, None].expand(-1, E)[has], 1, 2).to(torch.uint8)  # C or G
    codes[i0] = first
    codes[i0 + 1] = 3
    codes[i1 - 2] = 0
    codes[i1 - 1] = first
    return {"ex_st": ex_st, "ex_len": ex, "minus": minus}


def mutate_reads(torch, dev, g, src, bounds, err):
    """per-base error split 1/3 substitution, 1/3 insertion (random base before the kept base), 1/3 deletion; returns ASCII strings"""
    n = int(src.numel())
    hit = torch.rand(n, device=dev, generator=g) < err
    kind = torch.randint(0, 3, (n,), device=dev, generator=g, dtype=torch.uint8)
    sub, ins, dele = hit & (kind == 0), hit & (kind == 1), hit & (kind == 2)
    del hit, kind
    src = torch.where(sub, (src + torch.randint(1, 4, (n,), device=dev, generator=g, dtype=torch.uint8)) & 3, src)
    reps = 1 + ins.long() - dele.long()
    out = torch.repeat_interleave(src, reps)
    first = torch.cumsum(reps, 0) - reps
    ins_at = first[ins]
    out[ins_at] = torch.randint(0, 4, (int(ins_at.numel()),), device=dev, generator=g, dtype=torch.uint8)
    cum = torch.cat([torch.zeros(1, dtype=torch.long, device=dev), torch.cumsum(reps, 0)])
    ob = cum[bounds].cpu().numpy()
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    asc = lut[out.long()].cpu().numpy()
    return [asc[ob[i]:ob[i + 1]].tobytes() for i in range(len(ob) - 1)]


def gen_transcripts(torch, dev, seed, codes, genes, n_reads, err):
    """cDNA reads: the concatenated exons of a random gene, in transcript orientation or its reverse complement"""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    pick = torch.randint(0, genes["ex_st"].shape[0], (n_reads,), device=dev, generator=g)
    ex_st, ex_len = genes["ex_st"][pick], genes["ex_len"][pick]
    E = ex_len.shape[1]
    lens = ex_len.sum(1)
    rev = torch.rand(n_reads, device=dev, generator=g) < 0.5
    bounds = torch.cat([torch.zeros(1, dtype=torch.long, device=dev), torch.cumsum(lens, 0)])
    # per exon segment: (read, exon) flattened; base k of the segment comes from ex_st + k
    seg_len = ex_len.reshape(-1)
    seg_st = ex_st.reshape(-1)
    seg_id = torch.repeat_interleave(torch.arange(n_reads * E, device=dev), seg_len)
    seg_b = torch.cumsum(seg_len, 0) - seg_len
    idx = seg_st[seg_id] + (torch.arange(int(seg_len.sum().item()), device=dev) - seg_b[seg_id])
    src = codes[idx]  # genomic + strand, exon order
    rid = seg_id // E
    j = torch.arange(src.numel(), device=dev) - bounds[:-1][rid]
    revb = rev[rid]
    src = torch.where(revb, 3 - src, src)
    dst = bounds[:-1][rid] + torch.where(revb, lens[rid] - 1 - j, j)
    out = torch.empty_like(src)
    out[dst] = src
    return mutate_reads(torch, dev, g, out, bounds, err)


def gen_reads(torch, dev, seed, codes, per, n_contig, n_reads, mean_len, sd_len, err):
    """ONT-like reads, all at once on the device: uniformly placed substrings, half reverse-complemented, per-base error
    split 1/3 substitution, 1/3 insertion (random base before the kept base), 1/3 deletion.  Returns ASCII byte strings."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    lens = torch.clamp((torch.randn(n_reads, device=dev, generator=g) * sd_len + mean_len).long(), 1000, per)
    cid = torch.randint(0, n_contig, (n_reads,), device=dev, generator=g)
    st = (torch.rand(n_reads, device=dev, generator=g, dtype=torch.float64) * (per - lens + 1).double()).long()
    rev = torch.rand(n_reads, device=dev, generator=g) < 0.5
    bounds = torch.cat([torch.zeros(1, dtype=torch.long, device=dev), torch.cumsum(lens, 0)])
    n = int(bounds[-1].item())
    rid = torch.repeat_interleave(torch.arange(n_reads, device=dev), lens)
    j = torch.arange(n, device=dev) - bounds[:-1][rid]
    revb = rev[rid]
    idx = (cid * per + st)[rid] + torch.where(revb, lens[rid] - 1 - j, j)
    src = codes[idx]
    src = torch.where(revb, 3 - src, src)
    del idx, j
    return mutate_reads(torch, dev, g, src, bounds, err)


def plant_repeats(torch, dev, seed, codes, per, n_contig, frac):
    """--workload repeats: copies of sequence families written over the uniform reference until about `frac` of it is repeat --
    short interspersed elements (300 b units, hundreds to thousands of copies, 8-15 % diverged), long ones (6 kb, tens to hundreds of copies, 2-6 %),
    segmental duplications (20-60 kb, 2-4 copies, 0.5-2 %) and tandem arrays (units of 20-200 b repeated head to tail over 1-5 kb).  Reads from such a
    reference get several chains, secondaries with equal scores, minimizers above mid_occ, anchors with equal keys (the tie replays) and the long-join
    re-chaining -- what the headline's i.i.d. reference never asks of the device path.  Returns the number of bases written."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    total = per * n_contig
    budget = int(total * frac)
    written = 0
    classes = [(300, 300, 200, 3000, 0.08, 0.15, 0.45), (6000, 6000, 20, 300, 0.02, 0.06, 0.30), (20000, 60000, 2, 4, 0.005, 0.02, 0.20)]  # unit min / max, copies min / max, divergence min / max, share of the budget
    for lo, hi, c_lo, c_hi, d_lo, d_hi, share in classes:
        left = int(budget * share)
        while left > 0:
            L = int(torch.randint(lo, hi + 1, (1,), device=dev, generator=g).item())
            C = int(torch.randint(c_lo, c_hi + 1, (1,), device=dev, generator=g).item())
            C = max(2, min(C, left // L + 1))
            unit = torch.randint(0, 4, (L,), dtype=torch.uint8, device=dev, generator=g)
            div = d_lo + (d_hi - d_lo) * float(torch.rand(1, device=dev, generator=g).item())
            cid = torch.randint(0, n_contig, (C,), device=dev, generator=g)
            st = (torch.rand(C, device=dev, generator=g, dtype=torch.float64) * float(per - L)).long()
            rev = torch.rand(C, device=dev, generator=g) < 0.5
            cop = unit[None, :].repeat(C, 1)
            sub = torch.rand(C, L, device=dev, generator=g) < div
            cop = torch.where(sub, (cop + torch.randint(1, 4, (C, L), device=dev, generator=g, dtype=torch.uint8)) & 3, cop)
            cop = torch.where(rev[:, None], 3 - cop.flip(1), cop)
            idx = (cid * per + st)[:, None] + torch.arange(L, device=dev)[None, :]
            codes[idx.reshape(-1)] = cop.reshape(-1)
            left -= L * C
            written += L * C
    left = int(budget * 0.05)  # tandem arrays
    while left > 0:
        u = int(torch.randint(20, 201, (1,), device=dev, generator=g).item())
        n = int(torch.randint(1000, 5001, (1,), device=dev, generator=g).item())
        unit = torch.randint(0, 4, (u,), dtype=torch.uint8, device=dev, generator=g)
        arr = unit.repeat(n // u + 1)[:n]
        sub = torch.rand(n, device=dev, generator=g) < 0.03
        arr = torch.where(sub, (arr + torch.randint(1, 4, (n,), device=dev, generator=g, dtype=torch.uint8)) & 3, arr)
        c = int(torch.randint(0, n_contig, (1,), device=dev, generator=g).item())
        p0 = int(torch.randint(0, per - n, (1,), device=dev, generator=g).item())
        codes[c * per + p0:c * per + p0 + n] = arr
        left -= n
        written += n
    return written


def gen_reads_sv(torch, dev, seed, codes, per, n_contig, n_reads, mean_len, sd_len, err, sv_frac):
    """gen_reads with a structural difference in a share sv_frac of the reads: a deletion (the read skips w reference bases), an insertion (w random bases) or an
    inversion (w bases reverse-complemented in place), w in 300..1500, somewhere in the read's middle -- the gap fill over it trips the Z-drop test, the region
    needs a second DP round, a split, maybe the inversion rescue (align.c:846-870, :916-971)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    lens = torch.clamp((torch.randn(n_reads, device=dev, generator=g) * sd_len + mean_len).long(), 4000, per // 2)
    cid = torch.randint(0, n_contig, (n_reads,), device=dev, generator=g)
    kind = torch.where(torch.rand(n_reads, device=dev, generator=g) < sv_frac, torch.randint(1, 4, (n_reads,), device=dev, generator=g), torch.zeros(n_reads, dtype=torch.long, device=dev))  # 0 none, 1 deletion, 2 insertion, 3 inversion
    w = torch.randint(300, 1501, (n_reads,), device=dev, generator=g) * (kind > 0)
    p = 1200 + (torch.rand(n_reads, device=dev, generator=g) * (lens - 2400 - w).clamp(min=1).float()).long()
    span = lens + torch.where(kind == 1, w, torch.zeros_like(w)) - torch.where(kind == 2, w, torch.zeros_like(w))  # reference bases under the read
    st = (torch.rand(n_reads, device=dev, generator=g, dtype=torch.float64) * (per - span - 1).clamp(min=1).double()).long()
    rev = torch.rand(n_reads, device=dev, generator=g) < 0.5
    bounds = torch.cat([torch.zeros(1, dtype=torch.long, device=dev), torch.cumsum(lens, 0)])
    n = int(bounds[-1].item())
    rid = torch.repeat_interleave(torch.arange(n_reads, device=dev), lens)
    j = torch.arange(n, device=dev) - bounds[:-1][rid]      # position in the read as sequenced from the + strand copy
    k_, w_, p_ = kind[rid], w[rid], p[rid]
    inside = (j >= p_) & (j < p_ + w_)
    off = torch.where((k_ == 1) & (j >= p_), j + w_, j)                                   # deletion: skip w reference bases at p
    off = torch.where((k_ == 2) & (j >= p_ + w_), j - w_, off)                            # insertion: the bases after it come from w earlier
    off = torch.where((k_ == 3) & inside, p_ + (p_ + w_ - 1 - j), off)                    # inversion: read backwards inside the segment
    src = codes[(cid * per + st)[rid] + off]
    src = torch.where((k_ == 3) & inside, 3 - src, src)
    src = torch.where((k_ == 2) & inside, torch.randint(0, 4, (n,), device=dev, generator=g, dtype=torch.uint8), src)
    del off, inside, k_, w_, p_
    # the read as a whole from either strand
    revb = rev[rid]
    dst = bounds[:-1][rid] + torch.where(revb, lens[rid] - 1 - j, j)
    out = torch.empty_like(src)
    out[dst] = torch.where(revb, 3 - src, src)
    del src, dst, j, rid
    return mutate_reads(torch, dev, g, out, bounds, err), int((kind > 0).sum().item())


def gen_pairs(torch, dev, seed, codes, per, n_contig, n_pairs, read_len, err):
    """Illumina-like read pairs (FR): fragments of ~N(450, 60) bases placed uniformly, a read of read_len bases from each end (the
    second one reverse-complemented), the fragment taken from either strand, substitutions only at rate err.  Returns
    (reads 1, reads 2) as ASCII byte strings."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    frag = torch.clamp((torch.randn(n_pairs, device=dev, generator=g) * 60 + 450).long(), read_len, per)
    cid = torch.randint(0, n_contig, (n_pairs,), device=dev, generator=g)
    st = (torch.rand(n_pairs, device=dev, generator=g, dtype=torch.float64) * (per - frag + 1).double()).long()
    swap = torch.rand(n_pairs, device=dev, generator=g) < 0.5   # fragment from the reverse strand: the mates swap roles
    j = torch.arange(read_len, device=dev)[None, :]
    base = (cid * per + st)[:, None]
    left = codes[base + j]                                        # forward read at the fragment's left end
    right = 3 - codes[base + (frag[:, None] - 1 - j)]             # reverse-complemented read at its right end
    r1 = torch.where(swap[:, None], right, left)
    r2 = torch.where(swap[:, None], left, right)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    out = []
    for r in (r1, r2):
        sub = torch.rand(r.shape, device=dev, generator=g) < err
        r = torch.where(sub, (r + torch.randint(1, 4, r.shape, device=dev, generator=g, dtype=torch.uint8)) & 3, r)
        asc = lut[r.long()].cpu().numpy()
        out.append([asc[i].tobytes() for i in range(n_pairs)])
    return out[0], out[1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--ref-mb", type=float, default=3000.0)
    ap.add_argument("--reads", type=int, default=100000, help="reads per step (strong scaling: of the whole job; weak: per GPU)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"], help="N > 1: shard one batch (strong) or give every rank its own (weak)")
    ap.add_argument("--threads", type=int, default=0, help="host threads per rank (0: min(64, cores / gpus))")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--preset", default="map-ont", choices=["map-ont", "map-hifi", "lr:hq", "splice", "sr"],
                    help="map-ont is the BASELINE.json metric; the others are BASELINE.json's further configs / experiments (sr: --reads read PAIRS of 2 x --read-len bases)")
    ap.add_argument("--read-len", type=int, default=0, help="mean read length (0: 10000 for map-ont, 15000 otherwise)")
    ap.add_argument("--err", type=float, default=-1.0, help="per-base error rate (<0: 0.12 for map-ont, 0.005 otherwise)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="reads in the CPU baseline sample (0: sized for ~10 s)")
    ap.add_argument("--as-rank-threads", type=int, default=0, help="--as-rank-of: host threads of the one rank (0: this box's CPUs / N, i.e. N ranks under ONE quota of this box's size; e.g. 16: a node that gives every rank what this box has)")
    ap.add_argument("--as-rank-of", type=int, default=0, help="N > 1 (one GPU): after the N=1 measurement, map what ONE rank of an N-GPU strong-scaling job maps -- a 1/N base-balanced "
                    "share of every batch, with host_cpus()/N threads, hit packing included -- and report config.as_rank_of: the share's rate, N x that rate, its ratio to the N=1 rate "
                    "(predicted strong scaling if the ranks do not contend) and the host core-seconds per Gbase, which is what bounds the N-GPU line under a shared CPU quota")
    ap.add_argument("--workload", default="plain", choices=["plain", "repeats"], help="plain: BASELINE.json's i.i.d. reference (the headline); repeats: a tenth of the reference is planted repeat families "
                    "(interspersed, segmental duplications, tandem arrays) and 5 %% of the reads carry a deletion / insertion / inversion -- a side figure: how much of the batch leaves the device path")
    ap.add_argument("--timed-only", action="store_true", help="profiling runs: nothing after the timed steps (no resident / one-lane / formatting / CPU passes), so that the end of a kernel trace IS the timed pipeline")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node %d" % a.gpus
    backend = os.environ.get("MM2AMD_BENCH_BACKEND", "nccl")  # "gloo": ranks share whatever GPUs exist (plumbing tests only)
    ncpu = os.cpu_count() or 1

    import torch
    import torch.distributed as dist
    emu = os.environ.get("MM2AMD_EMU") == "1"  # plumbing check of this script in a container without a GPU: tests/_build/libmm2amd_emu.so (tests/conftest.py), data generated on the CPU
    dev_id = local_rank % max(torch.cuda.device_count(), 1)
    os.environ.setdefault("MM2AMD_DEVICE", str(dev_id))
    if emu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import conftest
        conftest.use_emulated_library()
        dev = torch.device("cpu")
        torch.cuda.synchronize = lambda *a_, **k_: None
        torch.cuda.empty_cache = lambda: None
    else:
        torch.cuda.set_device(dev_id)
        dev = torch.device("cuda", dev_id)
    comm_dev = dev if backend == "nccl" else torch.device("cpu")
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import minimap2_amd as mm
    from minimap2_amd import shard
    # A container's CPU quota counts (cgroup cpu.max): the host stages cannot use more CPU seconds per second than it grants, and pool threads
    # beyond it spend the quota on waking up and spinning (32 threads on a 16-CPU quota: 10.6 core-seconds per step instead of 7-8, and the step
    # is then bounded by exactly that: profiles/README.md, r03_bench_full_v18).
    ncpu = min(ncpu, mm.host_cpus())
    n_threads = a.threads if a.threads > 0 else max(1, min(64, ncpu // max(world, 1)))

    t0 = time.time()
    total = int(a.ref_mb * 1e6)
    n_contig = max(1, min(24, total // 1000000))
    codes, per = gen_reference(torch, dev, 11, total, n_contig)
    total = per * n_contig
    genes = plant_genes(torch, dev, 12, codes, per, n_contig, max(100, min(20000, total // 150000))) if a.preset == "splice" else None
    n_repeat_bases = n_sv_reads = 0
    if a.workload == "repeats":
        n_repeat_bases = plant_repeats(torch, dev, 13, codes, per, n_contig, 0.10)
        log("rank %d: %.1f Mb of planted repeats" % (rank, n_repeat_bases / 1e6))
    refs = reference_ascii(torch, dev, codes, per, n_contig)
    names = ["chr%d" % (i + 1) for i in range(n_contig)]
    log("rank %d: reference %d Mb in %d contigs generated in %.1f s" % (rank, total // 1000000, n_contig, time.time() - t0))
    t0 = time.time()
    mean_len = a.read_len if a.read_len > 0 else {"map-ont": 10000, "splice": 2000, "sr": 150}.get(a.preset, 15000)
    err = a.err if a.err >= 0 else {"map-ont": 0.12, "splice": 0.05}.get(a.preset, 0.005)
    pairs = a.preset == "sr"
    strong = a.scaling == "strong" and world > 1
    rseed = 1000 if strong else 1000 + rank  # strong scaling: one batch, the same on every rank
    if pairs:
        reads, mates = gen_pairs(torch, dev, rseed, codes, per, n_contig, a.reads, mean_len, err)
    elif a.preset == "splice":
        reads = gen_transcripts(torch, dev, rseed, codes, genes, a.reads, err)
    else:  # in chunks of at most ~1 Gbase: the index arithmetic of one call is 32-bit in places
        reads, chunk = [], max(1, min(a.reads, int(1.0e9 // mean_len)))
        for c0 in range(0, a.reads, chunk):
            if a.workload == "repeats":
                rr, nsv = gen_reads_sv(torch, dev, rseed + 7919 * (c0 // chunk), codes, per, n_contig, min(chunk, a.reads - c0), mean_len, mean_len // 10, err, 0.05)
                reads += rr
                n_sv_reads += nsv
            else:
                reads += gen_reads(torch, dev, rseed + 7919 * (c0 // chunk), codes, per, n_contig, min(chunk, a.reads - c0), mean_len, mean_len // 10, err)
            torch.cuda.empty_cache()
    del codes
    torch.cuda.empty_cache()
    whole_named = None
    if strong:  # this rank's share: contiguous, balanced by bases (pairs stay together: both mates in one entry)
        cut = shard.split_by_bases([len(r) + (len(mates[i]) if pairs else 0) for i, r in enumerate(reads)], world)
        first_read = cut[rank]
        if rank == 0 and not a.timed_only:  # the whole batch, for the check that the N ranks' output IS the N = 1 output (after the clock)
            whole_named = [("read%d" % i, s, mates[i]) for i, s in enumerate(reads)] if pairs else [("read%d" % i, s) for i, s in enumerate(reads)]
        reads = reads[cut[rank]:cut[rank + 1]]
        if pairs:
            mates = mates[cut[rank]:cut[rank + 1]]
    else:
        first_read = 0
    batch_bases = sum(len(r) for r in reads) + (sum(len(r) for r in mates) if pairs else 0)
    log("rank %d: %d %s, %.3f Gbases generated in %.1f s" % (rank, len(reads), "read pairs" if pairs else "reads", batch_bases / 1e9, time.time() - t0))
    t0 = time.time()
    al = mm.Aligner(refs, preset=a.preset, names=names, n_threads=n_threads, sam=True)
    t_index = time.time() - t0
    st = al.index_stat()
    log("rank %d: device index built in %.1f s: %d distinct minimizers, %d positions, mid_occ=%d" % (rank, t_index, st["n_distinct"], st["n_minimizers"], al.map_opt.mid_occ))
    named = [("read%d" % (first_read + i), s, mates[i]) for i, s in enumerate(reads)] if pairs else [("read%d" % (first_read + i), s) for i, s in enumerate(reads)]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    L = mm.lib()
    n_mapped = n_hits = 0
    base = mm.Batch(named)  # the reader's product: mm_bseq1_t records over host buffers (built once; every step maps a rotation of it)
    gbuf = shard.GatherBuffers() if (world > 1 or os.environ.get("MM2AMD_BENCH_FORCE_GATHER")) else None
    step_done = []
    last_text = []  # (batch, address, length) of the last text the output stage produced: valid until the next mm_gpu_format_batch_view

    def on_mapped(b, n_reg, reg, rep_len):  # output thread: the final hit gather to rank 0, then this rank formats its own shard
        nonlocal n_mapped, n_hits
        if gbuf is not None:  # hit records + the reads' rep_len (the rl:i tag of mm_write_sam3): what the formatting rank needs to write the shard's records
            shard.gather_payloads(shard.pack_hits(L, n_reg, reg, gbuf, tail=rep_len), dst=0, device=comm_dev, bufs=gbuf)
        nr = np.frombuffer(n_reg, dtype=np.int32, count=len(b.items))
        n_mapped, n_hits = int((nr > 0).sum()), int(nr.sum())

    def run_steps(steps):
        """One pass of the three-step pipeline over len(steps) batches, ALL of it inside the clock: hand-over of the reads (pinned-memory
        packing + H2D, beside the mapping of the batch before), mapping, hit gather (N > 1), SAM formatting (beside the mapping of the
        batch after).  Same pool of reads every step, rotated; nothing is cached between steps."""
        batches = [base.rotated((st * 997) % max(len(named), 1)) for st in steps]
        del step_done[:]
        barrier()
        t = time.time()
        trace = []
        def on_text(b_, addr, ln):  # output thread, inside the clock: only remember where the text is
            step_done.append((time.time(), ln))
            last_text[:] = [(b_, addr, ln)]
        al.pipeline(batches, text=True, on_mapped=on_mapped, on_text=on_text, trace=trace)
        barrier()
        dt = time.time() - t
        if rank == 0 and os.environ.get("MM2AMD_BENCH_TRACE"):  # when each step of each batch ran, relative to the start of the clock
            for name, k, a0, a1 in sorted(trace, key=lambda x: x[2]):
                log("  %-6s batch %2d  %7.3f .. %7.3f  (%.3f s)" % (name, k, a0 - t, a1 - t, a1 - a0))
        return dt

    if a.warmup > 0:
        dt = run_steps(range(a.warmup))
        log("rank %d warmup (%d steps): %.3f s  stats=%s" % (rank, a.warmup, dt, {k: round(v, 3) for k, v in al.last_stats().items()}))
    mm.profile_enable(not os.environ.get("MM2AMD_BENCH_NO_PROFILE"))  # (experiments: the timed steps without the per-kernel HIP events; the line then has no roofline)
    import resource

    def thread_cpu():
        """CPU seconds of every thread this process has or had... of every LIVE thread, by OS thread name (/proc/self/task/*/stat: utime + stime);
        the library names its own threads (mm2pool: the host stages' workers, mm2side: hand-over packing / formatting workers, mm2lane: lane drivers,
        mm2-stager / -mapper / -output: the pipeline's three steps), the HIP runtime's threads keep the interpreter's name."""
        out, tck = {}, float(os.sysconf("SC_CLK_TCK"))
        by_tid.clear()
        try:
            for tid in os.listdir("/proc/self/task"):
                try:
                    st_ = open("/proc/self/task/%s/stat" % tid).read()
                except OSError:
                    continue
                nm, rest = st_[st_.index("(") + 1:st_.rindex(")")], st_[st_.rindex(")") + 2:].split()
                out[nm] = out.get(nm, 0.0) + (int(rest[11]) + int(rest[12])) / tck
                by_tid[tid] = (nm, (int(rest[11]) + int(rest[12])) / tck)
        except OSError:
            pass
        return out
    by_tid = {}
    tc0 = thread_cpu()
    tid0 = dict(by_tid)
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    total_t = run_steps(range(a.warmup, a.warmup + a.steps))
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    tc1 = thread_cpu()
    # the busiest single threads of the timed steps (thread name, CPU seconds per step): tells a runtime helper that spins from a pool that works
    top_threads = sorted(((nm, round((v - tid0.get(t_, (nm, 0.0))[1]) / max(a.steps, 1), 3)) for t_, (nm, v) in by_tid.items()), key=lambda x: -x[1])[:8]
    # (threads that ended inside the window -- the lane drivers of every batch, the pipeline's three -- are not in tc1: their share is the remainder)
    cpu_by_thread = {k: round((v - tc0.get(k, 0.0)) / max(a.steps, 1), 3) for k, v in sorted(tc1.items()) if v - tc0.get(k, 0.0) >= 0.005 * max(a.steps, 1)}
    sam_bytes_per_step = step_done[-1][1] if step_done else 0
    # Parity of the timed path itself: the SAM text the pipeline produced for the LAST TIMED step -- still in the library's
    # reused buffer; hashed here, after the clock has stopped -- against the text of the same batch mapped outside the pipeline (mm_gpu_batch_stage +
    # mm_gpu_map_staged + mm_gpu_format_batch, the path the -m gpu suite pins to the compiled reference).
    def text_hash(addr, ln):
        import hashlib
        h = hashlib.blake2b(digest_size=16)
        view = (C.c_char * ln).from_address(addr) if ln else b""
        h.update(memoryview(view))
        return h.hexdigest()
    pipeline_text_identical = None
    step_text_lengths = sorted({ln for _, ln in step_done})
    if last_text and world == 1 and not a.timed_only:
        b_last, addr, ln = last_text[0]
        h_pipe = text_hash(addr, ln)
        al.stage(b_last)
        n_reg, reg, rep = al.run(raw=True)
        out, out_len = C.c_void_p(), C.c_size_t()
        mm._check(L.mm_gpu_format_batch(b_last.n, b_last.seg_off, b_last.n_seg, b_last.arr, n_reg, reg, rep, C.byref(out), C.byref(out_len)))
        h_plain = text_hash(out.value, out_len.value)
        mm._libc_free(out)
        al.free_raw(n_reg, reg)
        pipeline_text_identical = bool(h_pipe == h_plain and ln == out_len.value and ln > 0)
        log("text of the last timed step: %d bytes, blake2b %s (pipeline) vs %s (stage + run + format): %s" % (ln, h_pipe, h_plain, "identical" if pipeline_text_identical else "DIFFERENT"))
    # N > 1 (strong scaling): is what the N ranks produce the N = 1 output?  One more batch through the SAME path (pipeline, hit gather to rank 0, every rank
    # formats its shard), after the clock and un-rotated, so that the ranks' shards in rank order are the whole batch in input order (map.c:585-623 prints one
    # ordered stream).  Rank 0 then maps the whole batch alone and compares (a) the ranks' texts, in rank order, with the corresponding slices of its own text
    # (digests and lengths travel, not the text) and (b) the text it formats from the GATHERED hit records with its own.
    text_identical_to_n1 = None
    n1_check = None
    if strong and not a.timed_only:
        ver = {}
        def on_mapped_v(b, n_reg, reg, rep_len):
            parts = shard.gather_payloads(shard.pack_hits(L, n_reg, reg, gbuf, tail=rep_len), dst=0, device=comm_dev, bufs=gbuf)
            if rank == 0:
                ver["parts"] = [np.array(p_.cpu().numpy(), copy=True) for p_ in parts]  # (the gather's buffers are reused)
        def on_text_v(b_, addr, ln):
            ver["text"] = (text_hash(addr, ln), int(ln))
        barrier()
        al.pipeline([base], text=True, on_mapped=on_mapped_v, on_text=on_text_v)
        shard_texts = [None] * world
        dist.all_gather_object(shard_texts, ver.get("text"))
        if rank == 0:
            try:
                wb = mm.Batch(whole_named)
                al.stage(wb)
                n_reg, reg, rep = al.run(raw=True)
                out, out_len = C.c_void_p(), C.c_size_t()
                mm._check(L.mm_gpu_format_batch(wb.n, wb.seg_off, wb.n_seg, wb.arr, n_reg, reg, rep, C.byref(out), C.byref(out_len)))
                h_n1, off, ok_shards = text_hash(out.value, out_len.value), 0, True
                for h_, ln_ in shard_texts:  # (a) rank r's text == bytes [off, off + ln) of the N = 1 text
                    ok_shards = ok_shards and off + ln_ <= out_len.value and text_hash(out.value + off, ln_) == h_
                    off += ln_
                ok_shards = bool(ok_shards and off == out_len.value and out_len.value > 0)
                mm._libc_free(out)
                al.free_raw(n_reg, reg)
                # (b) the gathered payloads: every rank's hit records then its rep_len array; unpacked in rank order they are the whole batch's hits
                n_all = len(whole_named)
                n_reg_g, reg_g, rep_g = (C.c_int * n_all)(), (C.c_void_p * n_all)(), (C.c_int * n_all)()
                for r_ in range(world):
                    m_ = cut[r_ + 1] - cut[r_]
                    part = ver["parts"][r_]
                    part = part[:part.size - 4 * (max(1, m_) - m_)]  # (an empty shard's arrays have one unused entry)
                    nr_, rg_ = shard.unpack_hits(L, part[:part.size - 4 * m_], m_)
                    n_reg_g[cut[r_]:cut[r_ + 1]] = nr_[:m_]
                    reg_g[cut[r_]:cut[r_ + 1]] = rg_[:m_]
                    rep_g[cut[r_]:cut[r_ + 1]] = np.frombuffer(part[part.size - 4 * m_:].tobytes(), dtype=np.int32).tolist()
                out2, out2_len = C.c_void_p(), C.c_size_t()
                mm._check(L.mm_gpu_format_batch(wb.n, wb.seg_off, wb.n_seg, wb.arr, n_reg_g, reg_g, rep_g, C.byref(out2), C.byref(out2_len)))
                h_g = text_hash(out2.value, out2_len.value)
                mm._libc_free(out2)
                L.mm2amd_free_regs(n_all, n_reg_g, reg_g)
                text_identical_to_n1 = bool(ok_shards and h_g == h_n1)
                n1_check = {"n1_text_bytes": int(out_len.value), "n1_text_blake2b": h_n1, "shard_text_bytes": [ln_ for _, ln_ in shard_texts],
                            "shard_texts_equal_n1_slices": ok_shards, "text_from_gathered_hits_blake2b": h_g, "text_from_gathered_hits_equals_n1": bool(h_g == h_n1)}
                log("N=%d output vs N=1 on the same batch: shard texts %s, text formatted from the gathered hits %s" % (world, "identical" if ok_shards else "DIFFERENT", "identical" if h_g == h_n1 else "DIFFERENT"))
            except Exception as e:  # reported, never hidden
                text_identical_to_n1, n1_check = False, {"error": str(e)}
        barrier()
    log("rank %d: %d steps in %.3f s (%.3f s per step; last batch's text %d bytes)  stats=%s" % (rank, a.steps, total_t, total_t / max(a.steps, 1), sam_bytes_per_step, {k: round(v, 3) for k, v in al.last_stats().items()}))
    drv_cpu_last_batch = {k: round(v, 3) for k, v in al.last_stats().items() if k.startswith("drv_cpu_")}  # the lane drivers' own CPU seconds per stage, last timed batch
    host_cpu_s = (ru1.ru_utime - ru0.ru_utime + ru1.ru_stime - ru0.ru_stime) / max(a.steps, 1)
    log("rank %d: host CPU time per step %.2f core-seconds (%d threads; hand-over, host stages of the mapping, hit gather, formatting)" % (rank, host_cpu_s, n_threads))
    prof = {k: v for k, v in mm.profile_get().items() if v["launches"] > 0}  # (entries without launches are counts that go with a kernel: see extra1 below)
    mm.profile_enable(False)

    def one_step(step, staged_outside=True):
        """one batch without the pipeline: hand-over, then the mapping alone inside the clock (staged_outside) or both"""
        b = base.rotated((step * 997) % max(len(named), 1))
        t = time.time()
        al.stage(b)
        if staged_outside:
            barrier()
            t = time.time()
        n_reg, reg, _ = al.run(raw=True)
        dt = time.time() - t
        al.free_raw(n_reg, reg)
        return dt

    # side figures (rank 0, N = 1): the mapping call alone with the batch already resident (rounds 1-2's headline), and hand-over + mapping
    # one after the other (no pipeline)
    resident = pcie = None
    if world == 1 and not a.timed_only:
        resident = min(one_step(a.warmup + a.steps + i) for i in range(2))
        pcie = one_step(a.warmup + a.steps + 2, staged_outside=False)
    # un-overlapped kernel times: one more pass over the same batch with ONE lane (sub-batches one after the other, so no two
    # kernels of the path run at the same time and HIP-event spans are costs); feeds roofline.valu and roofline.unoverlapped_ms
    prof1 = None
    extra1 = {}
    t_one = None
    stage_cpu = None
    if rank == 0 and world == 1 and not a.timed_only:
        os.environ["MM2AMD_ACTIVE_LANES"] = "1"
        os.environ["MM2AMD_NO_SIDE_STREAM"] = "1"  # the lane-exact DP launches after the gap-fill kernel instead of beside it
        mm.profile_enable(True)
        t_one = one_step(a.warmup + a.steps)
        stage_cpu = {k: round(v, 3) for k, v in al.last_stats().items() if k.startswith("cpu_") or k.startswith("drv_cpu_")}
        log("un-overlapped pass (one lane): %.3f s; process CPU seconds while each stage ran: %s" % (t_one, stage_cpu))
        prof1 = mm.profile_get()
        extra1 = {k: v for k, v in prof1.items() if v["launches"] == 0}  # counts that go with a kernel without being a launch (band_cells_computed<NB>)
        prof1 = {k: v for k, v in prof1.items() if v["launches"] > 0}
        mm.profile_enable(False)
        del os.environ["MM2AMD_ACTIVE_LANES"], os.environ["MM2AMD_NO_SIDE_STREAM"]
    fmt = None
    if world == 1 and not a.timed_only:  # the output stage on its own (it runs beside the mapping in the timed region)
        try:
            b = base
            al.stage(b)
            n_reg, reg, rep = al.run(raw=True)
            out, out_len = C.c_void_p(), C.c_size_t()
            best = 1e9
            for _ in range(2):
                t = time.time()
                mm._check(L.mm_gpu_format_batch_view(b.n, b.seg_off, b.n_seg, b.arr, n_reg, reg, rep, C.byref(out), C.byref(out_len)))
                best = min(best, time.time() - t)
            al.free_raw(n_reg, reg)
            fmt = {"sam_bytes": out_len.value, "seconds": round(best, 3), "GB_per_s": round(out_len.value / best / 1e9, 3), "threads": n_threads, "inside_the_clock": True}
        except Exception as e:
            fmt = {"error": str(e)}
    cpu_all = [host_cpu_s]
    if world > 1:
        tt = torch.tensor([total_t], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        total_t = float(tt.item())
        bb = torch.tensor([batch_bases], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(bb, op=dist.ReduceOp.SUM)
        all_bases = float(bb.item())
        cc = [torch.zeros(1, dtype=torch.float64, device=comm_dev) for _ in range(world)]
        dist.all_gather(cc, torch.tensor([host_cpu_s], dtype=torch.float64, device=comm_dev))
        cpu_all = [round(float(x.item()), 2) for x in cc]
    else:
        all_bases = float(batch_bases)

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    value = all_bases * a.steps / total_t / 1e9
    # roofline of the dominant kernel.  "hbm": algorithmic bytes / HIP-event time on the launch stream,
    # from the timed steps.  "valu": what actually bounds it -- DP cells per second of the un-overlapped pass against the VALU issue
    # peak of its own instruction stream (72 VALU instructions per 128 cells in the ISA, a wave64 VALU
    # instruction = 2 cycles on a SIMD-32, 1024 SIMDs at 2.4 GHz, every lane useful).
    roof = None
    if prof:
        family = lambda k: k.split("[")[0].split("<")[0]  # launch classes of one kernel (ksw_stream_kernel<4>[t256], ...) count together
        src = prof1 or prof  # dominance by un-overlapped cost when we have it
        fam_total = {}
        for k, v in src.items():
            fam_total[family(k)] = fam_total.get(family(k), 0.0) + v["ms"]
        fam = max(fam_total, key=fam_total.get)
        same = {k: v for k, v in prof.items() if family(k) == fam}
        fam_ms = sum(v["ms"] for v in same.values())
        fam_bytes = sum(v["alg_bytes"] for v in same.values())
        fam_launch = sum(v["launches"] for v in same.values())
        ach = fam_bytes / max(fam_ms * 1e-3, 1e-12) / 1e9
        roof = {"bound": "hbm", "kernel": fam, "achieved": round(ach, 3), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 6),
                "traffic": None, "avg_launch_ms": round(fam_ms / max(fam_launch, 1), 4), "launches": fam_launch,
                "alg_bytes_per_launch": round(fam_bytes / max(fam_launch, 1), 1),
                "note": "integer DP: bound by VALU issue, not by HBM (see 'valu'; DESIGN.md section 4); hbm figures = algorithmic bytes of the timed steps / HIP-event time of the family on its launch streams (lanes overlap); traffic = PMC FETCH_SIZE(x2 gfx950 correction)+WRITE_SIZE per launch, profiles/pmc_traffic.json",
                "kernels_ms": {k: round(v["ms"], 3) for k, v in sorted(prof.items())},
                # per launch class over the timed steps: algorithmic bytes as each launch accounts them (backend_hip.cpp, ksw_host.cpp; the per-unit
                # figures are DESIGN.md section 4's) and the launches -- traffic / algorithmic is recomputable from this record and profiles/pmc_traffic.json
                "kernels_alg_bytes": {k: round(v["alg_bytes"], 1) for k, v in sorted(prof.items())},
                "kernels_launches": {k: v["launches"] for k, v in sorted(prof.items())}}
        inst = {}  # the family's launches per compiled instantiation (what a rocprofv3 kernel trace lists as one kernel name)
        for k, v in same.items():
            a_ = inst.setdefault(k.split("[")[0], [0.0, 0])
            a_[0] += v["ms"]
            a_[1] += v["launches"]
        roof["avg_launch_ms_by_instantiation"] = {k: round(v[0] / max(v[1], 1), 4) for k, v in sorted(inst.items())}
        if prof1:
            # the register-resident gap-fill DP: the streaming kernel (targets <= 512: >95 % of the cells), else the strip kernel (MM2AMD_NO_STREAM)
            # (round 6: or the banded kernel, ksw_band.hip -- whichever of the three costs most)
            dp_fams = {f: sum(v["ms"] for k, v in prof1.items() if family(k) == f) for f in ("ksw_band_kernel", "ksw_stream_kernel", "ksw_gapfill_kernel")}
            vfam = max(dp_fams, key=dp_fams.get) if max(dp_fams.values()) > 0 else fam
            one = {k: v for k, v in prof1.items() if family(k) == vfam and v["units"] > 0}
            ms1, cells1 = sum(v["ms"] for v in one.values()), sum(v["units"] for v in one.values())
            rect_cells1 = cells1
            if vfam == "ksw_band_kernel":
                # the banded kernel's launches account the cells of the RECTANGLES they stand for (what the reference computes); the issue roofline counts the
                # cells it computes: rows x 64 lanes x register sets x 2 jobs (band_cells_computed<NB>, ksw_host.cpp)
                comp = {k.replace("ksw_band_kernel", "band_cells_computed").split("[")[0]: k for k in one}
                for ck, k in comp.items():
                    one[k] = dict(one[k], units=extra1.get(ck, {}).get("units", 0.0))
                cells1 = sum(v["units"] for v in one.values())
            # Issue cost of one register-set row (128 cells) of the kernel's hot loop: its VALU instructions as counted in the ISA (hipcc -S,
            # gfx950: streaming kernel 35 packed VOP3P (the keyed cell's 34 and the store's share) + 6 DPP moves + 1 v_perm (VOP3) + 7 VOP2 = 49 -- 51 + 6 + 1 + 6
            # = 64 with round 3's cell; strip kernel 34 + 6 + 11 others + 12 VOP2), priced
            # with the per-SIMD issue-rate table profiles/r03_valu_issue_bench_v1.txt -- waves that shared a SIMD found through HW_ID, columns
            # B=8 and B=16 agree: VOP3P / VOP3 / DPP 4.1 cycles per wave64 instruction, VOP2 2.2.  valu_busy is the hardware's own figure for
            # the same thing: 4 x SQ_ACTIVE_INST_VALU / (SIMDs x GRBM_GUI_ACTIVE per XCD) of profiles/r04_pmc_sq_v34.json (r03_pmc_sq_v1.json: round 3's cell).  Lane utilisation =
            # cells / (128 x executed register-set rows), counted by the MM2AMD_GF_COUNT build: at HEAD on the wave emulator (a property of the schedule and the job mix:
            # profiles/r04_stream_lane_utilisation_emu.txt, 0.860 / 0.882 for the two classes; round 2 on the MI355X: 0.856 / 0.87, profiles/r02_stream_lane_utilisation.txt).
            # (round 5) the counts come from profiles/isa_row_counts.json -- tools/isa_row_counts.py: the hot loops' VALU instructions per register-set row, by
            # encoding class, counted in the gfx950 assembly of the sources whose SHA-256 the file carries; nothing typed in here.  Launch classes of the
            # family (the 4- and the 8-set instantiation) are priced separately and combined by their cells.
            isa = {}
            try:
                isa = json.load(open(os.path.join(ROOT, "profiles", "isa_row_counts.json")))
            except Exception:
                pass
            def sources_now():
                import hashlib
                h = hashlib.sha256()
                for f_ in isa.get("sources", []):
                    h.update(open(os.path.join(ROOT, "minimap2_amd", "csrc", f_), "rb").read())
                return h.hexdigest()
            try:
                isa_current = bool(isa) and sources_now() == isa.get("sources_sha256")
            except Exception:
                isa_current = False
            lane_util = (isa.get("lane_utilisation") or {}).get(vfam, 0.872 if vfam == "ksw_stream_kernel" else 1.0 if vfam == "ksw_band_kernel" else 0.727)  # (banded kernel: every lane of every row counted as computed)
            t_issue = t_nominal = 0.0  # seconds the family's cells take at the issue peak / at the guide's 2 cycles per instruction
            rows = {}
            for k, v in one.items():
                inst = k.split("[")[0]  # ksw_stream_kernel<4>, ksw_gapfill_kernel<512>, ...
                key = next((kk for kk in (isa.get("kernels") or {}) if kk.startswith(inst.rstrip(">") + ",")), None)
                c = (isa["kernels"][key]["per_register_set_row"] if key else {"vop3p": 35.0, "dpp": 6.0, "vop3": 1.0, "sdwa": 0.0, "vop2": 7.0})
                n_slow, n_vop2 = c["vop3p"] + c["dpp"] + c["vop3"] + c["sdwa"], c["vop2"]
                rc = n_slow * 4.1 + n_vop2 * 2.2
                rows[k] = {"isa_key": key, "valu_per_register_set_row": round(n_slow + n_vop2, 2), "issue_cycles_per_row": round(rc, 1)}
                t_issue += v["units"] / (1024 * 2.4e9 * 128 / rc)
                t_nominal += v["units"] / (1024 * 2.4e9 * 128 / ((n_slow + n_vop2) * 2.0))
            rate = cells1 / max(ms1 * 1e-3, 1e-12)
            peak_cells = cells1 / max(t_issue, 1e-12) if cells1 else 1.0
            nominal = cells1 / max(t_nominal, 1e-12) if cells1 else 1.0
            busy, busy_src = None, None
            try:
                import glob as _glob
                cand = sorted(_glob.glob(os.path.join(ROOT, "profiles", "r06_pmc_sq_*.json"))) or sorted(_glob.glob(os.path.join(ROOT, "profiles", "r05_pmc_sq_*.json")))
                busy_src = os.path.basename(cand[-1])
                sq = json.load(open(cand[-1]))["kernels"]  # (one --pmc pass at 20 k-read launches: tools/pmc_sq.py)
                act = sum(v["SQ_ACTIVE_INST_VALU"] for k, v in sq.items() if k.startswith(vfam))
                gui = sum(v["GRBM_GUI_ACTIVE"] for k, v in sq.items() if k.startswith(vfam)) / 8.0
                busy = round(4.0 * act / (1024.0 * gui), 4)
            except Exception:
                pass
            roof["valu"] = {"bound": "valu", "kernel": vfam, "cells_per_s": round(rate, 1), "issue_peak_cells_per_s": round(peak_cells, 1),
                            "frac": round(rate / peak_cells, 4), "lane_utilisation": lane_util, "frac_at_measured_lane_utilisation": round(rate / peak_cells / lane_util, 4),
                            "valu_busy_sq_counters": busy, "valu_busy_source": busy_src,
                            "nominal_2cycle_peak_cells_per_s": round(nominal, 1), "frac_nominal": round(rate / nominal, 4),
                            "unoverlapped_ms_per_step": round(ms1, 2), "cells_per_step": cells1,
                            "rectangle_cells_per_step": rect_cells1, "rectangle_cells_per_s": round(rect_cells1 / max(ms1 * 1e-3, 1e-12), 1),  # the cells of the windows as the reference computes them (== cells_per_step except for the banded kernel)
                            "rows": rows, "isa_counts": {"file": "profiles/isa_row_counts.json", "commit": isa.get("commit"), "made_from_the_sources_this_run_uses": isa_current},
                            "gap_fill_family_unoverlapped_ms_per_step": round(sum(v["ms"] for k, v in prof1.items() if family(k) in ("ksw_band_kernel", "ksw_stream_kernel", "ksw_gapfill_kernel")), 2),
                            "basis": "one extra pass with a single lane and no side stream (no concurrent kernels); peak = 1024 SIMDs x 2.4 GHz x 128 cells / the issue cycles of one register-set row: the hot loop's VALU instructions per row by encoding class (profiles/isa_row_counts.json, counted in the assembly by tools/isa_row_counts.py) at the per-SIMD issue rates of profiles/r03_valu_issue_bench_v1.txt (VOP3P / VOP3 / DPP 4.1 cycles, VOP2 2.2), every lane useful; valu_busy_sq_counters = 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE per XCD) (the SIMDs' issue cycles that carried a VALU instruction, traceback and Z-drop walk included); nominal = the same instructions at the guide's 2 cycles per wave64 instruction"}
            # the same family without the other lanes beside it (the one-lane pass): per-launch durations in the timed steps depend on how many of the eight lanes'
            # launches of this kernel coincide -- the kernel is VALU-bound, eight coinciding launches each take eight times as long -- so `frac` moves between 0.04
            # and 0.06 at an unchanged step time; this one does not
            fam1 = {k: v for k, v in prof1.items() if family(k) == fam}
            if fam1:
                ms_f, by_f = sum(v["ms"] for v in fam1.values()), sum(v["alg_bytes"] for v in fam1.values())
                roof["achieved_unoverlapped"] = round(by_f / max(ms_f * 1e-3, 1e-12) / 1e9, 3)
                roof["frac_unoverlapped"] = round(by_f / max(ms_f * 1e-3, 1e-12) / 1e9 / HBM_PEAK_GBPS, 6)
            roof["unoverlapped_ms"] = {k: round(v["ms"], 3) for k, v in sorted(prof1.items())}
            roof["unoverlapped_alg_bytes"] = {k: round(v["alg_bytes"], 1) for k, v in sorted(prof1.items())}
            roof["unoverlapped_alg_gb_per_s"] = {k: round(v["alg_bytes"] / max(v["ms"], 1e-9) / 1e6, 1) for k, v in sorted(prof1.items()) if v["alg_bytes"] > 0}
            roof["unoverlapped_gcells_per_s"] = {k: round(v["units"] / max(v["ms"], 1e-9) / 1e6, 1) for k, v in sorted(prof1.items()) if v["units"] > 0}  # DP kernels: cells of the launch class / its time
            roof["unoverlapped_step_ms"] = round(t_one * 1e3, 1) if t_one else None
            try:  # index probes per second of seed_collect_kernel (one mm_idx_get per query minimizer; the launch accounts 36 B per minimizer at the expected density 2 / (w + 1))
                sc1 = prof1.get("seed_collect_kernel")
                if sc1 and sc1["ms"] > 0:
                    roof["index_probes"] = {"per_step": round(sc1["alg_bytes"] / 36.0), "per_s_unoverlapped": round(sc1["alg_bytes"] / 36.0 / (sc1["ms"] * 1e-3), 1),
                                            "basis": "minimizers probed (read bases x 2 / (w + 1)) / un-overlapped seed_collect_kernel time; a probe = bucket_start + keys + val_off + position list, 2-3 dependent sector reads"}
            except Exception:
                pass
        tj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tj):
            try:
                t = json.load(open(tj)).get(fam)
                if t:
                    roof["traffic"] = round(t.get("traffic_bytes_per_launch", t["fetch_bytes_x2"] + t["write_bytes_per_launch"]), 1)
                    # traffic and the algorithmic bytes it is compared with come from the SAME launches (the counter passes' own bench line: tools/pmc_traffic.py, 20 k-read
                    # launches, one lane); `alg_bytes_per_launch` above is this run's launch size -- the two are not to be divided by each other
                    roof["traffic_over_algorithmic"] = t.get("traffic_over_algorithmic")
                    roof["traffic_alg_bytes_per_launch_of_the_counter_pass"] = t.get("alg_bytes_per_launch")
                    roof["traffic_source"] = {"file": "profiles/pmc_traffic.json", "commit": t.get("commit"), "fetch_factor": t.get("fetch_factor", 2.0)}
            except Exception:
                pass

    cpu = None
    if world == 1 and not a.no_cpu_baseline and not a.timed_only:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import reflib
            if not os.path.exists(reflib.REFDRV_SO):
                raise RuntimeError("oracle/_ref/librefdrv.so not present")
            t0 = time.time()
            S, keys, val_off, pos = reflib.export_index(al)
            drv = reflib.RefDriver(st["w"], st["k"], st["flag"], names, al.lens, S, keys, val_off, pos, ncpu)
            del keys, val_off, pos
            mo = drv.map_opt(a.preset, extra_flag=mm.F_OUT_SAM)
            log("reference mm_idx_t adopted from the exported tables in %.1f s (mid_occ %d)" % (time.time() - t0, mo.mid_occ))
            # give the reference its best thread count on this host (it does not always scale to every hardware thread)
            probe = named[:min(2000, len(named))]
            best_thr, best_rate = ncpu, 0.0
            for thr in sorted({ncpu, max(1, ncpu // 2), max(1, ncpu // 4)}, reverse=True):
                t_probe, nr, rg = drv.map(mo, probe, thr)
                L.mm2amd_free_regs(len(nr), nr, rg)
                rate = len(probe) / max(t_probe, 1e-3)
                log("reference probe: %d threads -> %.0f reads/s" % (thr, rate))
                if rate > best_rate:
                    best_thr, best_rate = thr, rate
            n_s = a.cpu_sample
            if n_s <= 0:
                n_s = int(min(len(named), max(2000, best_rate * 10.0)))
            sample = named[:n_s]
            t_cpu, nr, rg = drv.map(mo, sample, best_thr)
            # parity spot check on the sample: the GPU path must reproduce the reference's hit records byte for byte
            want = shard.pack_hits(L, nr, rg).numpy().tobytes()
            L.mm2amd_free_regs(len(nr), nr, rg)
            al.stage(sample)
            n_reg, reg, _ = al.run(raw=True)
            got = shard.pack_hits(L, n_reg, reg).numpy().tobytes()
            al.free_raw(n_reg, reg)
            sb = sum(sum(len(x) for x in r[1:]) for r in sample)
            cpu = {"value": round(sb / t_cpu / 1e9, 5), "unit": "Gbases/s", "cores": best_thr, "kind": "reference",
                   "sample": "%d of the batch's reads or read pairs (%.3f Gbases), mm_map / mm_map_frag on %d threads (kt_for; best of %d/%d/%d threads on a 2000-read probe), mapping loop only, same index contents" % (n_s, sb / 1e9, best_thr, ncpu, ncpu // 2, ncpu // 4),
                   "hits_identical_to_gpu": got == want}
            drv.close()
        except Exception as e:  # the baseline is reported, never required
            cpu = {"value": None, "unit": "Gbases/s", "cores": ncpu, "kind": "reference", "sample": "unavailable: %s" % e}

    as_rank = None
    if a.as_rank_of > 1 and world == 1:
        N = a.as_rank_of
        try:
            al.close()
            thr_share = a.as_rank_threads if a.as_rank_threads > 0 else max(1, n_threads // N)
            cut = shard.split_by_bases([sum(len(x) for x in r[1:]) for r in named], N)
            share = named[cut[0]:cut[1]]
            share_bases = sum(sum(len(x) for x in r[1:]) for r in share)
            al = mm.Aligner(refs, preset=a.preset, names=names, n_threads=thr_share, sam=True)
            base_s = mm.Batch(share)
            gb_s = shard.GatherBuffers()
            def on_mapped_s(b, n_reg, reg, rep_len):  # a rank's part of the hit gather: the records packed into the pinned buffer (the transfer itself needs peers)
                shard.pack_hits(L, n_reg, reg, gb_s)
            def pipe_s(steps):
                bs = [base_s.rotated((st * 997) % max(len(share), 1)) for st in steps]
                barrier()
                t = time.time()
                al.pipeline(bs, text=True, on_mapped=on_mapped_s)
                barrier()
                return time.time() - t
            pipe_s(range(max(a.warmup, 2)))
            k_s = max(a.steps, 8)
            ru_a = resource.getrusage(resource.RUSAGE_SELF)
            t_s = pipe_s(range(100, 100 + k_s))
            ru_b = resource.getrusage(resource.RUSAGE_SELF)
            cpu_s_share = (ru_b.ru_utime - ru_a.ru_utime + ru_b.ru_stime - ru_a.ru_stime) / k_s
            rate_s = share_bases * k_s / t_s / 1e9
            as_rank = {"n_gpus": N, "reads_this_rank": len(share), "share_gbases": round(share_bases / 1e9, 4), "host_threads": thr_share, "steps": k_s,
                       "ms_per_step": round(t_s / k_s * 1e3, 2), "share_gbases_per_s": round(rate_s, 5), "n_x_share_gbases_per_s": round(N * rate_s, 4),
                       "predicted_strong_scaling": round(N * rate_s / value, 3), "host_cpu_s_per_gbase": round(cpu_s_share / max(share_bases / 1e9, 1e-12), 3),
                       "host_cpu_s_per_gbase_n1": round(host_cpu_s / max(batch_bases / 1e9, 1e-12), 3),
                       "cpu_quota_of_this_box": ncpu,
                       "cpu_bound_gbases_per_s_if_n_ranks_share_this_quota": round(ncpu / max(cpu_s_share / max(share_bases / 1e9, 1e-12), 1e-9), 3),
                       "note": "one rank's share of every batch on one GPU with 1/N of this box's CPU quota; N ranks that do not contend for CPU, PCIe or the formatting rank reach N x the share rate; under ONE shared quota of this size the job is bounded by quota / core-seconds per Gbase"}
            log("as rank of %d: %s" % (N, as_rank))
        except Exception as e:  # a side figure, never required
            as_rank = {"error": str(e)}
    rl = "2 x %d b reads" % mean_len if pairs else "%d kb reads" % (mean_len // 1000)
    out = {"metric": "aligned Gbases/sec (%s, %s, -a)" % (a.preset, rl), "value": round(value, 5), "unit": "Gbases/s", "n_gpus": world, "steps": a.steps,
           "warmup": a.warmup, "ms_per_step": round(total_t / a.steps * 1e3, 2), "higher_is_better": True, "scaling": a.scaling if world > 1 else "strong", "vs_baseline": None,
           "dtype": "int8 (ksw2 difference DP) / int32+f32 (chaining)", "data": "synthetic",
           "config": {"workload": "%s: %d synthetic %s %s (%g%% error) vs %d Mb synthetic ref (24 contigs), -a" % (a.preset, a.reads, ("read pairs, " + rl) if pairs else ("~" + rl), "per GPU" if (world > 1 and not strong) else "per step", err * 100, total // 1000000) +
                                  (" -- SIDE FIGURE, not BASELINE's workload: %.0f Mb of the reference are planted repeat families (interspersed elements, segmental duplications, tandem arrays), %d of the reads carry a 300-1500 b deletion / insertion / inversion" % (n_repeat_bases / 1e6, n_sv_reads) if a.workload == "repeats" else ""),
                      "reads_per_step": a.reads * (world if (world > 1 and not strong) else 1), "reads_this_rank": len(reads), "ref_mb": total // 1000000, "batch_gbases": round(batch_bases / 1e9, 4), "host_threads_per_rank": n_threads, "host_cpu_s_per_step": round(host_cpu_s, 2),
                      "parallelism": "replicated index, %s, RCCL hit gather to rank 0, every rank formats its shard" % ("one batch sharded %d-way by bases" % world if strong else "%d independent batches" % world) if world > 1 else "1 GPU",
                      "clock": "pipeline of hand-over | mapping | SAM formatting over the timed steps, all three inside the clock (map.c:541-643)",
                      "host_cpu_s_per_step_per_rank": cpu_all,
                      "host_cpu_s_per_step_by_thread_name": cpu_by_thread, "host_cpu_s_per_step_busiest_threads": top_threads,
                      "host_cpu_s_per_stage_one_lane_pass": stage_cpu,
                      "lane_driver_cpu_s_last_timed_batch": drv_cpu_last_batch,
                      "sam_bytes_per_step_this_rank": sam_bytes_per_step,
                      "pipeline_text_identical": pipeline_text_identical, "timed_steps_text_bytes": step_text_lengths,
                      "text_identical_to_n1": text_identical_to_n1, "n1_check": n1_check,
                      "resident_gbases_per_s": round(batch_bases / resident / 1e9, 5) if resident else None,
                      "handover_then_map_gbases_per_s": round(batch_bases / pcie / 1e9, 5) if pcie else None,
                      "index_build_s": round(t_index, 2), "reads_mapped": n_mapped, "hits": n_hits, "as_rank_of": as_rank,
                      "cpu_quota": ncpu,
                      # the banded gap fill (ksw_band.hip), counted over the whole process: windows tried in 128 / 256 diagonals, sent on to the wider band, recomputed as rectangles
                      "banded_gap_fill": {k: int(v) for k, v in al.last_stats().items() if k.startswith("n_band")},
                      # how much of the last batch left the device path (hand-backs go through the host's plan / consume rounds), long-join re-chains on the device / by the host's tie-exact tree
                      "device_path_last_batch": {k: int(v) for k, v in al.last_stats().items() if k.startswith("n_region_reads") or k.startswith("n_long_join") or k in ("n_jobs", "n_rounds")},
                      "arenas": {k: int(v) for k, v in al.last_stats().items() if k.startswith("arena_") or k in ("dev_allocs", "pin_allocs")}},
           "roofline": roof, "cpu_baseline": cpu, "output_stage": fmt}
    print(json.dumps(out), flush=True)
    al.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
