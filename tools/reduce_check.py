#!/usr/bin/env python3
"""pisces_hip_comm_* / pisces_hip_reduce_summary with one process per GPU (what a C# host with one process per device calls; RCCL bound by the
library): launched as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/reduce_check.py`.  Every
rank hands in {rank + 1, 10 (rank + 1), 100, 1}; every rank must get the sums back.  Prints one line per rank and exits non-zero on a mismatch.
torch.distributed only carries the communicator's id from rank 0 to the others (gloo: no GPU collective of torch's is involved)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

from pisces_amd import _abi, engine

rank, world, local_rank = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
dist.init_process_group(backend="gloo", rank=rank, world_size=world)
torch.cuda.set_device(local_rank)
with engine.HipVariantCaller(_abi.default_config(), device=local_rank) as c:
    ids = [engine.HipVariantCaller.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(ids, src=0)
    c.comm_init(ids[0], rank, world)
    got = c.reduce_summary([rank + 1, 10 * (rank + 1), 100, 1])
want = [world * (world + 1) // 2, 10 * world * (world + 1) // 2, 100 * world, world]
print(f"reduce_check rank {rank}/{world} on cuda:{local_rank}: {got} (want {want})", flush=True)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if got == want else 1)
