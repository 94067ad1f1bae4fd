"""The N>1 path on CPU: two gloo ranks each map their base-balanced shard (with the oracle-backed checker build of the host
pipeline, since there is no GPU here), the packed hit records are gathered to rank 0 and must equal a single-process run."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
CHECK_SO = os.path.join(HERE, "_build", "libmm2amd_check.so")
import reflib  # noqa: E402


def test_split_by_bases():
    from minimap2_amd.shard import split_by_bases
    assert split_by_bases([], 4) == [0, 0, 0, 0, 0]
    assert split_by_bases([10], 2) in ([0, 0, 1], [0, 1, 1])
    lens = np.random.default_rng(1).integers(1000, 20000, 1000)
    for w in (1, 2, 4, 8):
        b = split_by_bases(lens, w)
        assert b[0] == 0 and b[-1] == 1000 and all(b[i] <= b[i + 1] for i in range(w))
        tot = [int(lens[b[i]:b[i + 1]].sum()) for i in range(w)]
        assert max(tot) - min(tot) <= 2 * 20000


def _make_inputs():
    import synth
    rng = np.random.default_rng(77)
    contigs = synth.gen_reference(rng, 400000, 2)
    reads = synth.gen_reads(rng, contigs, 24, 4000, 1500, 0.1)
    return [synth.ACGT[c].tobytes() for c in contigs], [synth.ACGT[r].tobytes() for r in reads]


def _map_with_check_lib(refs, reads):
    """returns (L, n_reg, reg) for `reads` mapped against `refs` through the drop-in C ABI of the checker library"""
    import minimap2_amd as mm
    L = mm.lib(CHECK_SO)
    R = C.CDLL(reflib.REF_SO)
    io, mo = mm.IdxOpt(), mm.MapOpt()
    R.mm_set_opt(None, C.byref(io), C.byref(mo))
    mo.flag |= mm.F_CIGAR
    R.mm_idx_str.restype = C.c_void_p
    R.mm_idx_str.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p)]
    names = [b"c%d" % i for i in range(len(refs))]
    mi = R.mm_idx_str(io.w, io.k, 0, io.bucket_bits, len(refs), (C.c_char_p * len(refs))(*refs), (C.c_char_p * len(refs))(*names))
    R.mm_mapopt_update.argtypes = [C.c_void_p, C.c_void_p]
    R.mm_mapopt_update(C.byref(mo), mi)
    assert L.mm_gpu_init(mi, C.byref(mo), 2) == 0, L.mm2amd_last_error()
    n = len(reads)
    arr = (mm.Bseq1 * max(n, 1))()
    keep = [b"r%d" % i for i in range(n)]
    for i in range(n):
        arr[i].l_seq, arr[i].rid, arr[i].name, arr[i].seq = len(reads[i][1]), i, reads[i][0], reads[i][1]
    n_reg, reg = (C.c_int * max(n, 1))(), (C.c_void_p * max(n, 1))()
    seg_off, n_seg = (C.c_int * max(n, 1))(*range(n)), (C.c_int * max(n, 1))(*([1] * n))
    assert L.mm_gpu_map_batch(n, seg_off, n_seg, arr, n_reg, reg, None, None) == 0, L.mm2amd_last_error()
    L.mm_gpu_destroy()
    return L, n_reg, reg, keep


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from minimap2_amd import shard
    refs, reads = _make_inputs()
    named = [(b"read%d" % i, s) for i, s in enumerate(reads)]
    b = shard.split_by_bases([len(s) for s in reads], world)
    mine = named[b[rank]:b[rank + 1]]
    L, n_reg, reg, _ = _map_with_check_lib(refs, mine)
    payload = shard.pack_hits(L, (C.c_int * len(mine))(*n_reg[:len(mine)]), (C.c_void_p * len(mine))(*reg[:len(mine)]))
    parts = shard.gather_payloads(payload, dst=0)
    if rank == 0:
        got = b"".join(bytes(p.numpy().tobytes()) for p in parts)
        L1, n1, r1, _ = _map_with_check_lib(refs, named)
        want = shard.pack_hits(L1, n1, r1).numpy().tobytes()
        # round trip through unpack as well
        n2, r2 = shard.unpack_hits(L1, np.frombuffer(got, dtype=np.uint8), len(named))
        again = shard.pack_hits(L1, n2, r2).numpy().tobytes()
        q.put((got == want, again == want, sum(n1[:len(named)])))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(not (os.path.exists(CHECK_SO) and os.path.exists(reflib.REF_SO)), reason="needs tests/_build and oracle/_ref (dev container)")
def test_two_rank_gather_equals_single_process():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    same, roundtrip, n_hits = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert same and roundtrip and n_hits >= 20


EMU_SO = os.path.join(HERE, "_build", "libmm2amd_emu.so")


@pytest.mark.skipif(not os.path.exists(EMU_SO), reason="needs tests/_build/libmm2amd_emu.so (the product's sources under the wave emulator)")
@pytest.mark.parametrize("world", [2, 3])
def test_bench_n_ranks_produce_the_n1_text(world):
    """bench.py --gpus N, exactly as the driver launches it (torch.distributed.run, one process per rank), on the wave emulator with gloo:
    the ranks' SAM texts in rank order must be the text one rank writes for the whole batch, and so must the text rank 0 formats from the
    gathered hit records (map.c:585-623 prints ONE ordered stream; BASELINE.json: "SAM diff == 0" at 1/2/4/8)."""
    import json
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MM2AMD_BENCH_BACKEND="gloo", MM2AMD_EMU="1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "1", "--warmup", "0", "--ref-mb", "1", "--reads", "25", "--read-len", "3000", "--no-cpu-baseline"]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=900, cwd="/tmp")
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    line = json.loads(p.stdout.decode().strip().split("\n")[-1])
    c = line["config"]
    assert line["n_gpus"] == world and line["scaling"] == "strong"
    assert c["text_identical_to_n1"] is True, c["n1_check"]
    chk = c["n1_check"]
    assert chk["shard_texts_equal_n1_slices"] and chk["text_from_gathered_hits_equals_n1"]
    assert len(chk["shard_text_bytes"]) == world and sum(chk["shard_text_bytes"]) == chk["n1_text_bytes"] > 0
