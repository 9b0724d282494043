"""Probe (development tool): two processes on ONE GPU, gloo backend, all_reduce of a device tensor."""
import os, sys, torch, torch.distributed as dist, torch.multiprocessing as mp
def worker(rank, world):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = "29577"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    t = torch.full((1000,), float(rank + 1), device="cuda")
    w = dist.all_reduce(t, async_op=True); w.wait(); torch.cuda.synchronize()
    print(rank, "sum", float(t[0]), flush=True)
    dist.destroy_process_group()
if __name__ == "__main__":
    mp.spawn(worker, args=(2,), nprocs=2)
