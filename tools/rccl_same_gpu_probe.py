"""Probe: can two ranks that share ONE GPU run an RCCL (backend "nccl") all-reduce?  (The round's boxes have one GPU; the driver's
8-GPU node is the only place the real xGMI path runs.)  Prints the outcome per rank; never raises."""
import os, sys
import torch, torch.distributed as dist, torch.multiprocessing as mp


def worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    try:
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
        t = torch.full((1 << 20,), float(rank + 1), device="cuda:0")
        dist.all_reduce(t)
        torch.cuda.synchronize()
        print(f"rank {rank}: all_reduce over nccl on a shared GPU OK, value {t[0].item()}", flush=True)
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        print(f"rank {rank}: {type(e).__name__}: {str(e)[:300]}", flush=True)


if __name__ == "__main__":
    mp.start_processes(worker, args=(2, 29631), nprocs=2, join=True, start_method="spawn")
