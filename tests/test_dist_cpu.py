"""Multi-rank path on CPU: world_size-2 gloo processes exercise the partition,
the uniform broadcast, the OR-reduction of Newton convergence masks (so every
rank verifies the SAME batch-global trip table) and the all-gather that
reassembles the PSF volume.  The renderer is replaced by a deterministic CPU
stand-in: what is tested is sdirt_amd/dist.py and newton.py, not the kernels."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def fake_psf(points, u, ks):
    """Deterministic stand-in for lens.psf_lr: depends on the point AND on the
    shared uniforms, so a rank with different uniforms would be caught."""
    base = points.sum(-1).reshape(-1, 1, 1) + u[0][:4].sum() + 2 * u[2][:4].sum()
    grid = torch.arange(ks * ks, dtype=torch.float32).reshape(1, ks, ks)
    return base + grid, -(base + grid)


def _worker(rank, world, port, n_total, ks, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sdirt_amd import dist as sd
    from sdirt_amd.newton import TripPlanner
    torch.manual_seed(100 + rank)             # ranks deliberately start with different RNG states
    pts = torch.linspace(0, 1, n_total * 3).reshape(n_total, 3)

    sharded = sd.ShardedPSF(lambda p, u: fake_psf(p, u, ks), "cpu")
    a, b = sharded.local_slice(n_total)
    L, R = sharded.psf_volume(pts, spp=16, gather=True)
    Ll, Rl = sharded.psf_volume(pts, spp=16, gather=False)
    # the point-to-point ("direct") gather must reassemble the same tensor as the collective
    mine = torch.arange(b - a, dtype=torch.float32).reshape(-1, 1) + 100 * rank
    g1 = sd.all_gather_shards(mine, n_total, world, algo="allgather")
    g2 = sd.all_gather_shards(mine, n_total, world, algo="direct")
    assert torch.equal(g1, g2) and g1.shape[0] == n_total

    # batch-global trip table: rank 1 holds the slow ray on surface 1
    need = [10, 3 + rank, 0, 2]
    curved = [True, True, False, True]

    def launch(trips):
        m = []
        for T, t in zip(trips, need):
            m.append(sum(1 << j for j in range(1, int(T) + 1) if j < t))
        return sd.reduce_masks_or(torch.tensor(m, dtype=torch.int32)).numpy()
    trips = TripPlanner().run("k", curved, range(4), launch)
    torch.save(dict(L=L, R=R, Ll=Ll, a=a, b=b, trips=trips), os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [10, 7])
def test_sharded_volume_gloo_world2(tmp_path, n_total):
    world, ks = 2, 3
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_total, ks, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"r{i}.pt", weights_only=False) for i in range(world)]
    # every rank reassembled the same full volume
    assert r[0]["L"].shape == (n_total, ks, ks)
    assert torch.equal(r[0]["L"], r[1]["L"]) and torch.equal(r[0]["R"], r[1]["R"])
    assert torch.equal(r[0]["R"], -r[0]["L"])
    # shards are a contiguous partition and the gather put them in order
    assert (r[0]["a"], r[0]["b"], r[1]["a"], r[1]["b"]) == (0, n_total // 2, n_total // 2, n_total)
    # rank 0's uniforms were used everywhere: rebuild the expected volume from rank 0's seed
    torch.manual_seed(100)
    u = [torch.rand(16), torch.rand(16), torch.rand(2048), torch.rand(2048)]
    pts = torch.linspace(0, 1, n_total * 3).reshape(n_total, 3)
    exp, _ = fake_psf(pts, u, ks)
    assert torch.equal(r[0]["L"], exp)
    # un-gathered call returned only the local shard (a fresh draw -> compare shapes only)
    assert r[1]["Ll"].shape[0] == n_total - n_total // 2
    # both ranks converged to the table of the slowest ray ANYWHERE in the batch
    assert list(r[0]["trips"]) == [10, 4, 0, 2] and list(r[1]["trips"]) == [10, 4, 0, 2]


def test_shard_bounds_and_mask_reduce_single_process():
    from sdirt_amd import dist as sd
    assert sd.shard_bounds(16384, 8)[3] == (6144, 8192)
    b = sd.shard_bounds(10, 4)
    assert b[0][0] == 0 and b[-1][1] == 10 and all(x[1] == y[0] for x, y in zip(b, b[1:]))
    m = torch.tensor([0b1010, 0], dtype=torch.int32)
    assert torch.equal(sd.reduce_masks_or(m), m)          # no process group: identity
