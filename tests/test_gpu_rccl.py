"""The RCCL transport of the final hit gather (minimap2_amd/shard.py) on the one GPU a test box has: a process group with backend "nccl" (= RCCL on
ROCm) of world size 1 -- RCCL refuses two ranks on one device -- through which the gather's own calls run: the all_gather of the payload sizes, the copy of the
local payload into the receive buffer, the pinned host copy; and the grouped point-to-point launch (batch_isend_irecv) the other ranks' payloads travel by, here
as a send to and a receive from the rank itself.  What a one-GPU box cannot show -- two devices, xGMI -- is covered at world size 2 and 3 by the gloo cases
(tests/test_sharding.py: same code, CPU tensors) and by the N-rank output check of bench.py (text_identical_to_n1)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import os, sys
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
from minimap2_amd import shard
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29591")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
g = torch.Generator().manual_seed(5)
bufs = shard.GatherBuffers()
for n in (0, 1, 4097, 3000000):
    payload = torch.randint(0, 256, (n,), dtype=torch.uint8, generator=g)
    got = shard.gather_payloads(payload, dst=0, device=dev, bufs=bufs)
    assert len(got) == 1 and got[0].numel() == n and bool((got[0].cpu() == payload).all()), n
# the grouped point-to-point launch the other ranks' payloads travel by, as a self send / receive
send = torch.arange(1 << 20, dtype=torch.int32, device=dev)
recv = torch.zeros_like(send)
for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, send, 0), dist.P2POp(dist.irecv, recv, 0)]):
    req.wait()
torch.cuda.synchronize()
assert bool((recv == send).all())
dist.barrier()
dist.destroy_process_group()
print("RCCL OK")
'''


def test_hit_gather_through_rccl_on_one_device():
    p = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=170,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert p.returncode == 0 and b"RCCL OK" in p.stdout, p.stderr.decode()[-1500:]
