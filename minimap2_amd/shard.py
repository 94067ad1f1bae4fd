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


def pack_hits(L, n_reg, reg):
    """ctypes (n_reg, reg) arrays -> uint8 torch tensor holding the flat payload (mm2amd_pack_regs)."""
    n = len(n_reg)
    need = L.mm2amd_pack_regs(n, n_reg, reg, None, 0)
    if need < 0:
        raise RuntimeError(L.mm2amd_last_error().decode())
    buf = np.empty(max(int(need), 1), dtype=np.uint8)
    got = L.mm2amd_pack_regs(n, n_reg, reg, buf.ctypes.data_as(C.c_void_p), need)
    assert got == need
    return torch.from_numpy(buf[:need])


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


def gather_payloads(payload, dst=0, device=None):
    """The final hit gather: every rank contributes one uint8 payload, rank `dst` receives them in rank order.
    Sizes travel in one all_gather; the data in one gather of equally padded buffers (one xGMI hop per peer).
    Returns the list of payload tensors on dst, None elsewhere."""
    if not dist.is_initialized():  # single process: the gather is the identity
        return [payload]
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = device if device is not None else payload.device
    size = torch.tensor([payload.numel()], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, size)
    sizes = [int(s.item()) for s in sizes]
    cap = max(max(sizes), 1)
    send = torch.zeros(cap, dtype=torch.uint8, device=dev)
    send[:payload.numel()] = payload.to(dev)
    recv = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(world)] if rank == dst else None
    dist.gather(send, recv, dst=dst)
    if rank != dst:
        return None
    return [recv[r][:sizes[r]] for r in range(world)]
