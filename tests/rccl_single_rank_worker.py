"""The collectives bench.py and sdirt_amd/dist.py issue, on the RCCL backend with a single rank (the boxes of the
pool have one GPU, and RCCL does not let two ranks share one): process-group creation with a device id, a second
communicator, broadcast, MAX all-reduce of int32 lanes and of a float64 scalar, all_gather_into_tensor through
all_gather_shards, barrier.  Checks that this software stack accepts every call with the dtypes and layouts the
multi-GPU path uses; what it cannot show is anything about xGMI.  Exit code 0 = all held."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29611")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from sdirt_amd import dist as sd
    gather_group = dist.new_group()
    buf = torch.arange(12288, dtype=torch.float32, device=dev)          # the 48 KB pupil broadcast
    dist.broadcast(buf, src=0)
    lanes = (torch.arange(129 * 11, device=dev, dtype=torch.int32) % 2).view(129, 11).contiguous()
    want = lanes.clone()
    dist.all_reduce(lanes, op=dist.ReduceOp.MAX)
    assert torch.equal(lanes, want)
    local = torch.rand(37, 2, 21, 21, device=dev)
    for algo in ("allgather", "direct"):
        full = sd.all_gather_shards(local, 37, 1, group=gather_group, algo=algo)
        assert torch.equal(full, local)
    comm = torch.cuda.Stream(dev)
    with torch.cuda.stream(comm):                                        # the gather on its own stream, as in bench.py
        out = torch.empty_like(local)
        sd.all_gather_shards(local, 37, 1, group=gather_group, out=out)
    torch.cuda.synchronize(dev)
    assert torch.equal(out, local)
    t = torch.tensor([1.25], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t.item()) == 1.25
    dist.barrier()
    torch.cuda.synchronize(dev)
    dist.destroy_process_group()
    print("rccl single-rank collectives ok")


if __name__ == "__main__":
    main()
