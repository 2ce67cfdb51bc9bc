#!/usr/bin/env python3
"""Can RCCL run two ranks on ONE GPU?  (The pool hands out one-GPU boxes; config 5's gather is RCCL on the driver's node.)
usage: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 tools/rccl_probe.py
Every rank uses cuda:0.  Prints what init, an all-reduce and a batched uint8 send / recv (the config-5 gather's call) do."""
import os
import sys

import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    t = torch.full((4,), float(rank + 1), device="cuda")
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print("rank %d: all_reduce -> %s" % (rank, t.tolist()), flush=True)
    buf = torch.full((1 << 20,), rank + 7, dtype=torch.uint8, device="cuda")
    if rank == 0:
        got = torch.empty((world - 1, 1 << 20), dtype=torch.uint8, device="cuda")
        ops = [dist.P2POp(dist.irecv, got[r - 1], r) for r in range(1, world)]
    else:
        ops = [dist.P2POp(dist.isend, buf, 0)]
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    torch.cuda.synchronize()
    if rank == 0:
        print("rank 0: received %s" % [int(got[r - 1, 0]) for r in range(1, world)], flush=True)
    dist.destroy_process_group()
    print("rank %d: ok" % rank, flush=True)
except Exception as e:   # noqa: BLE001 -- the probe reports whatever RCCL says
    print("rank %d: FAILED: %s: %s" % (rank, type(e).__name__, str(e).splitlines()[0][:300]), flush=True)
    sys.exit(3)
