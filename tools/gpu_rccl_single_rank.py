"""Diagnostic (not a pytest): RCCL comes up on this image / pool and the path's collectives run — ONE rank (a one-GPU box has
no second device): init_process_group("nccl"), barrier, all_gather_rows / all_gather_var_rows of the distributed module, and the
non-blocking isend / irecv pair of the halo exchange to itself is NOT possible with one rank, so only the collectives."""
import os
import sys
import torch
import torch.distributed as dist
sys.path.insert(0, ".")
from moonsuperresolution_amd.distributed import all_gather_rows, all_gather_var_rows
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dist.barrier()
t = torch.arange(2 * 1024 * 4096, dtype=torch.float32, device="cuda").reshape(2 * 1024, 4096)
g = all_gather_rows(t, 2, 1024, 1)
v = all_gather_var_rows(t, [2 * 1024])
torch.cuda.synchronize()
assert torch.equal(g, t) and torch.equal(v, t)
x = torch.ones(1 << 20, device="cuda")
dist.all_reduce(x)
torch.cuda.synchronize()
print("RCCL single-rank: init, barrier, all_gather_rows, all_gather_var_rows, all_reduce OK;", torch.cuda.nccl.version())
dist.destroy_process_group()
