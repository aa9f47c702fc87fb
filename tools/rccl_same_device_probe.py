#!/usr/bin/env python3
"""Can RCCL run two ranks on ONE device (to rehearse gs_comm_init / gs_allreduce_grads with N = 2 on a one-GPU box)?
Two child processes, both on device 0, torch.distributed backend nccl (= RCCL), one tiny all-reduce.  Prints the outcome."""
import os, subprocess, sys

if len(sys.argv) > 1:
    import torch, torch.distributed as dist
    rank = int(sys.argv[1])
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29617", RANK=str(rank), WORLD_SIZE="2")
    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", rank=rank, world_size=2)
        t = torch.ones(4, device="cuda") * (rank + 1)
        dist.all_reduce(t)
        torch.cuda.synchronize()
        print(f"rank {rank}: all_reduce on one device worked: {t.tolist()}", flush=True)
    except Exception as e:                                   # noqa: BLE001 -- the message is the result
        print(f"rank {rank}: RCCL refused: {type(e).__name__}: {str(e).splitlines()[0][:300]}", flush=True)
    sys.exit(0)

procs = [subprocess.Popen([sys.executable, __file__, str(r)]) for r in range(2)]
for p in procs:
    try:
        p.wait(timeout=90)
    except subprocess.TimeoutExpired:
        p.kill()
        print("timeout: a rank hung in the rendezvous / collective", flush=True)
