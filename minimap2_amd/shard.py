"""Multi-GPU sharding of a read batch: reads never interact (worker_for touches only its own
fragment, map.c:425-474), so a batch is split into contiguous, base-balanced shards, one per rank; every rank holds a
full replica of the index and runs the whole hot path on its shard; the only exchange is the final gather of the packed
hit records to the formatting rank.  One process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on the GPU
box, "gloo" in the CPU tests)."""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist


def split_by_bases(lens, world_size):
    """Contiguous split of reads 0..n-1 into world_size shards with balanced total length.
    Returns boundaries b[0..world_size] (b[0] = 0, b[-1] = n); shard r is reads b[r]..b[r+1]-1."""
    lens = np.asarray(lens, dtype=np.int64)
    n = len(lens)
    cum = np.concatenate([[0], np.cumsum(lens)])
    total = int(cum[-1])
    b = [0]
    for r in range(1, world_size):
        target = total * r // world_size
        i = int(np.searchsorted(cum, target, side="left"))
        b.append(min(max(i, b[-1]), n))
    b.append(n)
    return b


class GatherBuffers(object):
    """Reused buffers of the per-batch hit gather: a pinned host buffer the payload is packed into and (on the formatting rank) received
    into, device send / receive buffers for the RCCL gather.  They only grow; after the first batches a gather allocates nothing."""

    def __init__(self):
        self.pack = self.send = self.recv = self.host = None

    @staticmethod
    def _grow(t, n, **kw):
        if t is None or t.numel() < n:
            t = torch.empty(int(n * 1.25) + 64, dtype=torch.uint8, **kw)
        return t

    def pack_buffer(self, n):
        pin = torch.cuda.is_available()
        self.pack = self._grow(self.pack, n, pin_memory=pin)
        return self.pack


def pack_hits(L, n_reg, reg, bufs=None, tail=None):
    """ctypes (n_reg, reg) arrays -> uint8 torch tensor holding the flat payload (mm2amd_pack_regs); with `bufs` into its pinned buffer.
    tail: a ctypes int array of one entry per fragment (mm_tbuf_t::rep_len, which the SAM writer's rl:i tag needs) appended after the records."""
    n = len(n_reg)
    need = L.mm2amd_pack_regs(n, n_reg, reg, None, 0)
    if need < 0:
        raise RuntimeError(L.mm2amd_last_error().decode())
    extra = 4 * n if tail is not None else 0
    if bufs is not None:
        t = bufs.pack_buffer(max(int(need) + extra, 1))
        got = L.mm2amd_pack_regs(n, n_reg, reg, C.c_void_p(t.data_ptr()), need)
        assert got == need
        if extra:
            C.memmove(t.data_ptr() + need, tail, extra)
        return t[:need + extra]
    buf = np.empty(max(int(need) + extra, 1), dtype=np.uint8)
    got = L.mm2amd_pack_regs(n, n_reg, reg, buf.ctypes.data_as(C.c_void_p), need)
    assert got == need
    if extra:
        C.memmove(buf.ctypes.data + need, tail, extra)
    return torch.from_numpy(buf[:need + extra])


def unpack_hits(L, payload, n_frag):
    """uint8 tensor/array -> freshly allocated (n_reg, reg) ctypes arrays (free with mm2amd_free_regs)."""
    arr = payload.cpu().numpy() if isinstance(payload, torch.Tensor) else np.asarray(payload, dtype=np.uint8)
    arr = np.ascontiguousarray(arr)
    n_reg = (C.c_int * n_frag)()
    reg = (C.c_void_p * n_frag)()
    rc = L.mm2amd_unpack_regs(arr.ctypes.data_as(C.c_void_p), arr.size, n_frag, n_reg, reg)
    if rc != 0:
        raise RuntimeError(L.mm2amd_last_error().decode())
    return n_reg, reg


def gather_payloads(payload, dst=0, device=None, bufs=None):
    """The final hit gather: every rank contributes one uint8 payload, rank `dst` receives them in rank order.
    Sizes travel in one all_gather; the data as UNPADDED point-to-point transfers, grouped into one RCCL launch (batch_isend_irecv: every peer
    sends exactly its bytes over its own xGMI link straight into its slice of one contiguous receive buffer -- round 3 gathered buffers padded
    to the largest payload).  With `bufs` (GatherBuffers) the device buffers are reused from batch to batch and the received payloads reach
    the host in ONE copy into a pinned buffer.
    Returns the list of payload tensors on dst (host tensors when `bufs` is given and the device is a GPU), None elsewhere."""
    if not dist.is_initialized():  # single process: the gather is the identity
        return [payload]
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = device if device is not None else payload.device
    size = torch.tensor([payload.numel()], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, size)
    sizes = [int(s.item()) for s in sizes]
    on_gpu = torch.device(dev).type == "cuda"
    n = payload.numel()
    if rank != dst:
        if n > 0:
            if bufs is not None:
                bufs.send = GatherBuffers._grow(bufs.send, n, device=dev)
                send = bufs.send[:n]
            else:
                send = torch.empty(n, dtype=torch.uint8, device=dev)
            send.copy_(payload, non_blocking=True)
            for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, send, dst)]):
                req.wait()
            if on_gpu:  # (the pinned pack buffer and the send buffer are reused by the next batch: the copy and the send must have left them)
                torch.cuda.current_stream().synchronize()
        return None
    total = sum(sizes)
    offs = [0]
    for r in range(world):
        offs.append(offs[-1] + sizes[r])
    if bufs is not None:
        bufs.recv = GatherBuffers._grow(bufs.recv, max(total, 1), device=dev)
        recv = bufs.recv[:max(total, 1)]
    else:
        recv = torch.empty(max(total, 1), dtype=torch.uint8, device=dev)
    ops = [dist.P2POp(dist.irecv, recv[offs[r]:offs[r + 1]], r) for r in range(world) if r != dst and sizes[r] > 0]
    if n > 0:
        recv[offs[dst]:offs[dst + 1]].copy_(payload, non_blocking=True)
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    if bufs is not None and on_gpu:  # device -> one pinned host buffer, the payloads back to back, one copy
        bufs.host = GatherBuffers._grow(bufs.host, max(total, 1), pin_memory=True)
        bufs.host[:total].copy_(recv[:total], non_blocking=True)
        torch.cuda.current_stream().synchronize()
        return [bufs.host[offs[r]:offs[r + 1]] for r in range(world)]
    return [recv[offs[r]:offs[r + 1]] for r in range(world)]
