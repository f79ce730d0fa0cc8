#!/usr/bin/env python3
"""What one collective costs when it sits ON the critical path of a ~0.7 ms training step: a one-rank "nccl" (RCCL) process group on
one MI355X -- the floor of every multi-rank collective (kernel launch + RCCL's own protocol setup; wire time comes on top).

    python tools/rccl_latency.py            # prints a markdown table (gpurun_out/... > profiles/r06_rccl_latency.md)

Per message size: device time of one collective issued on an idle stream (event pair around it) and the time a DEPENDENT kernel
behind it starts later than it would without the collective (chain: fill -> collective -> fill, against fill -> fill).
Used by DESIGN.md 6 to size ZeRO-1 for this model: the sharded step trades a 58 us local optimizer pass for a scalar all-reduce (the
global norm) + an all-gather of the updated weights, both on the critical path."""
import os
import sys

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch
import torch.distributed as dist


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    dev = torch.device("cuda", 0)
    sizes = [4, 4096, 1 << 20, 4 << 20, 16 << 20, 33 << 20]
    ops = {
        "all_reduce": lambda t, o: dist.all_reduce(t),
        "all_gather_into_tensor": lambda t, o: dist.all_gather_into_tensor(o, t),
        "reduce_scatter_tensor": lambda t, o: dist.reduce_scatter_tensor(o, t),
    }
    print("| collective (1 rank, RCCL) | bytes | device us (median) | host enqueue us (median) |\n|---|---:|---:|---:|")
    import time
    for name, fn in ops.items():
        for nbytes in sizes:
            n = max(1, nbytes // 4)
            t = torch.ones(n, dtype=torch.float32, device=dev)
            o = torch.empty(n, dtype=torch.float32, device=dev)
            for _ in range(5):
                fn(t, o)
            torch.cuda.synchronize()
            dts, hts = [], []
            for _ in range(30):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                h0 = time.perf_counter()
                fn(t, o)
                hts.append((time.perf_counter() - h0) * 1e6)
                e1.record()
                torch.cuda.synchronize()
                dts.append(e0.elapsed_time(e1) * 1e3)
            dts.sort()
            hts.sort()
            print(f"| {name} | {nbytes} | {dts[len(dts) // 2]:.1f} | {hts[len(hts) // 2]:.1f} |", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
