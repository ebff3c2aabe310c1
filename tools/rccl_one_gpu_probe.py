#!/usr/bin/env python3
"""Can two processes of ONE GPU form an RCCL communicator (so that ncclSend/ncclRecv to another RANK could be executed on a one-GPU
box)?  python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/rccl_one_gpu_probe.py
Prints what RCCL answers; exits 0 either way."""
import os
import sys

import torch
import torch.distributed as dist


def main():
    rank = int(os.environ["RANK"])
    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", rank=rank, world_size=int(os.environ["WORLD_SIZE"]), device_id=torch.device("cuda", 0))
        x = torch.full((1024,), float(rank), device="cuda:0")
        if rank == 0:
            dist.send(x, 1)
        else:
            dist.recv(x, 0)
        torch.cuda.synchronize()
        print(f"rank {rank}: RCCL communicator of two ranks on one device works; received {x[0].item()}", flush=True)
        dist.destroy_process_group()
    except Exception as e:      # the answer, not a failure of the probe
        print(f"rank {rank}: RCCL refuses: {type(e).__name__}: {str(e)[:400]}", flush=True)
    sys.exit(0)


if __name__ == "__main__":
    main()
