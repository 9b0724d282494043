#!/usr/bin/env python
"""RCCL bus-bandwidth probe over the GPUs of one node (development tool).

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port 29555 tools/rccl_probe.py

Times all-reduce, reduce-scatter and all-gather of fp32 / bf16 buffers at the sizes of the data-parallel step (one 64 MB
gradient bucket, a 6-layer group's range 170 MB, the embedding tail 197 MB, the whole trainable range 890 MB) and prints
algorithm and bus bandwidth the way rccl-tests does (bus = alg x 2 (N-1)/N for all-reduce, x (N-1)/N for the other two).
xGMI is point-to-point: a ring is bound by ONE link (~153 GB/s per direction), direct reduce-scatter / all-gather can use all
seven - the numbers tell which one RCCL picked.  With one rank (this repository's test boxes) it only proves the transport
loads and runs.
"""
import os
import time

import torch
import torch.distributed as dist


def main():
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29555")
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    sizes_mb = [64, 170, 197, 890]
    rows = []
    for dtype, esz in ((torch.float32, 4), (torch.bfloat16, 2)):
        for mb in sizes_mb:
            n = (mb * 1024 * 1024 // 4) // (world * 256) * (world * 256)     # elements of the fp32 range; bf16 moves half the bytes
            full = torch.randn(n, device=dev).to(dtype)
            shard = torch.empty(n // world, device=dev, dtype=dtype)
            ops = {"all_reduce": lambda: dist.all_reduce(full),
                   "reduce_scatter": lambda: dist.reduce_scatter_tensor(shard, full),
                   "all_gather": lambda: dist.all_gather_into_tensor(full, shard)}
            for name, fn in ops.items():
                for _ in range(3):
                    fn()
                torch.cuda.synchronize(); dist.barrier()
                t0 = time.perf_counter()
                iters = 10
                for _ in range(iters):
                    fn()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / iters
                t = torch.tensor([dt], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t)
                nbytes = n * esz
                alg = nbytes / dt / 1e9
                bus = alg * (2.0 if name == "all_reduce" else 1.0) * (world - 1) / max(world, 1)
                rows.append((name, str(dtype).split(".")[-1], nbytes / 1e6, dt * 1e3, alg, bus))
    if rank == 0:
        print(f"world {world}")
        print(f"{'collective':15s} {'dtype':9s} {'MB':>8s} {'ms':>9s} {'alg GB/s':>10s} {'bus GB/s':>10s}")
        for r in rows:
            print(f"{r[0]:15s} {r[1]:9s} {r[2]:8.1f} {r[3]:9.3f} {r[4]:10.1f} {r[5]:10.1f}")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
