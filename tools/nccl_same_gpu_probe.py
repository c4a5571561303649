"""Probe: can two ranks share ONE GPU under the nccl (= RCCL) backend?  (1-GPU boxes only have this option for exercising the
nccl branch of cald_amd.sweep.allgather_scores.)  Prints the outcome; never fails the caller."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port):
    try:
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world,
                                device_id=torch.device("cuda", 0))
        x = torch.full((4,), float(rank), device="cuda", dtype=torch.float64)
        out = torch.empty((world * 4,), device="cuda", dtype=torch.float64)
        dist.all_gather_into_tensor(out, x)
        torch.cuda.synchronize()
        print("rank", rank, "nccl same-GPU all_gather ok:", out.cpu().tolist(), flush=True)
        dist.destroy_process_group()
    except Exception as e:
        print("rank", rank, "nccl same-GPU FAILED:", repr(e)[:300], flush=True)


if __name__ == "__main__":
    port = int(sys.argv[1]) if len(sys.argv) > 1 else 29617
    mp.spawn(worker, args=(2, port), nprocs=2, join=True)
