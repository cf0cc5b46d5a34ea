"""Probe: can two ranks share ONE GPU under the nccl (RCCL) backend?  (tests/test_sharded.py wants to drive
HipStages + a real collective with world_size 2 on the single-GPU box.)  Falls back to reporting gloo."""
import os
import sys
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def worker(rank, world, port, backend, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    try:
        torch.cuda.set_device(0)
        dist.init_process_group(backend, rank=rank, world_size=world)
        t = torch.full((4,), float(rank + 1), device="cuda:0")
        out = torch.empty(world * 4, device="cuda:0")
        dist.all_gather_into_tensor(out, t)
        torch.cuda.synchronize()
        q.put((rank, backend, out.cpu().tolist()))
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        q.put((rank, backend, "ERR " + repr(e)[:300]))


if __name__ == "__main__":
    for backend in ("nccl", "gloo"):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        ps = [ctx.Process(target=worker, args=(r, 2, port, backend, q)) for r in range(2)]
        for p in ps:
            p.start()
        res = []
        for _ in range(2):
            try:
                res.append(q.get(timeout=120))
            except Exception as e:  # noqa: BLE001
                res.append(("timeout", backend, repr(e)))
        for p in ps:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
        print(backend, sorted(res, key=str), flush=True)
