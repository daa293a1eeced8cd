"""Multi-rank path on CPU: world_size-2 gloo processes exercise the partition,
the uniform broadcast, the OR-reduction of Newton convergence masks (so every
rank verifies the SAME batch-global trip table) and the all-gather that
reassembles the PSF volume.  The renderer is replaced by a deterministic CPU
stand-in: what is tested is sdirt_amd/dist.py and newton.py, not the kernels."""
import os
import socket
import sys

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
    # the form the Lensgroup-backed renderer takes: the shard is rendered IN PLACE into the [width, 2, ks, ks] block
    # that ONE collective sends (no torch.stack), L / R are views of the gathered volume
    torch.manual_seed(100 + rank)

    def render_into(p, u, out=None):
        l, r = fake_psf(p, u, ks)
        out[:, 0], out[:, 1] = l, r
    blocked = sd.ShardedPSF(render_into, "cpu", ks=ks)
    Lb, Rb = blocked.psf_volume(pts, spp=16, gather=True)
    assert torch.equal(Lb, L) and torch.equal(Rb, R)
    if n_total % world == 0:
        assert Lb._base is not None and Lb._base is Rb._base and Lb._base.shape == (n_total, 2, ks, ks)   # views
    with pytest.raises(ValueError, match="largest shard"):
        sd.all_gather_shards(torch.zeros(n_total + 1, 2), n_total, world, padded=True)
    # the point-to-point ("direct") gather must reassemble the same tensor as the collective
    mine = torch.arange(b - a, dtype=torch.float32).reshape(-1, 1) + 100 * rank
    g1 = sd.all_gather_shards(mine, n_total, world, algo="allgather")
    g2 = sd.all_gather_shards(mine, n_total, world, algo="direct")
    assert torch.equal(g1, g2) and g1.shape[0] == n_total

    # the checksum test of a gathered volume (bench.py's gather.volume_checksums_equal): true for what the gather delivered,
    # false ON EVERY RANK as soon as one rank's copy differs in one bit of one row
    rows = torch.arange((b - a) * 6, dtype=torch.float32).reshape(b - a, 2, 3) + 1000.0 * rank
    vol = sd.all_gather_shards(rows, n_total, world)
    assert sd.gathered_volume_holds_every_shard(vol, rows, n_total) is True
    broken = vol.clone()
    if rank == world - 1 and n_total:
        broken.view(torch.int32)[n_total // 2, 1, 2] ^= 1          # the last bit of one value, on one rank only
    assert sd.gathered_volume_holds_every_shard(broken, rows, n_total) is False

    # batch-global trip table: the LAST rank holds the slow ray on surface 1
    need = [10, 3 + (rank == world - 1), 0, 2]
    curved = [True, True, False, True]

    def launch(trips):
        m = []
        for T, t in zip(trips, need):
            m.append(sum(1 << j for j in range(1, int(T) + 1) if j < t))
        return sd.reduce_masks_or(torch.tensor(m, dtype=torch.int32)).numpy()
    trips = TripPlanner().run("k", curved, range(4), launch)
    keep = slice(None) if n_total <= 64 else slice(0, n_total, 4099)      # large grids: a sample + a checksum
    torch.save(dict(L=L[keep].clone(), R=R[keep].clone(), Ll_rows=Ll.shape[0], a=a, b=b, trips=trips,
                    sum_L=float(L.double().sum()), rows=L.shape[0],
                    direct_equal=bool(torch.equal(g1, g2))), os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_total", [(2, 10), (2, 7), (8, 65536 + 3), (8, 5), (8, 64)])
def test_sharded_volume_gloo(tmp_path, world, n_total):
    """world 2 and world 8 (the node size of BASELINE config 3): uneven shards (65539 = 8 x 8192 + 3)
    and EMPTY shards (5 points over 8 ranks: three ranks render nothing but must still enter every
    broadcast, mask reduction and gather), both gather algorithms."""
    ks = 3
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_total, ks, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"r{i}.pt", weights_only=False) for i in range(world)]
    from sdirt_amd.dist import shard_bounds
    bounds = shard_bounds(n_total, world)
    # every rank reassembled the same full volume
    assert all(x["rows"] == n_total for x in r)
    assert all(torch.equal(r[0]["L"], x["L"]) and torch.equal(r[0]["R"], x["R"]) for x in r)
    assert all(x["sum_L"] == r[0]["sum_L"] for x in r)
    assert torch.equal(r[0]["R"], -r[0]["L"])
    assert all(x["direct_equal"] for x in r)
    # shards are the contiguous partition and the gather put them in order
    assert [(x["a"], x["b"]) for x in r] == bounds
    if n_total == 5:
        assert sum(1 for a, b in bounds if a == b) == 3
    # rank 0's uniforms were used everywhere: rebuild the expected volume from rank 0's seed
    torch.manual_seed(100)
    u = [torch.rand(16), torch.rand(16), torch.rand(2048), torch.rand(2048)]
    pts = torch.linspace(0, 1, n_total * 3).reshape(n_total, 3)
    exp, _ = fake_psf(pts, u, ks)
    keep = slice(None) if n_total <= 64 else slice(0, n_total, 4099)
    assert torch.equal(r[0]["L"], exp[keep]) and r[0]["sum_L"] == float(exp.double().sum())
    # un-gathered call returned only the local shard
    assert [x["Ll_rows"] for x in r] == [b - a for a, b in bounds]
    # every rank converged to the table of the slowest ray ANYWHERE in the batch
    assert all(list(x["trips"]) == [10, 4, 0, 2] for x in r)


def test_shard_bounds_and_mask_reduce_single_process():
    from sdirt_amd import dist as sd
    assert sd.shard_bounds(16384, 8)[3] == (6144, 8192)
    b = sd.shard_bounds(10, 4)
    assert b[0][0] == 0 and b[-1][1] == 10 and all(x[1] == y[0] for x, y in zip(b, b[1:]))
    m = torch.tensor([0b1010, 0], dtype=torch.int32)
    assert torch.equal(sd.reduce_masks_or(m), m)          # no process group: identity
