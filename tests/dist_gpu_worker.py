"""One rank of tests/test_gpu_dist.py: the REAL renderer under a 2-rank process group.

Both ranks share cuda:0 (the GPU boxes of the pool have one GPU); collectives go through gloo,
exactly the control flow `bench.py` runs with SDIRT_BENCH_BACKEND=gloo and -- with RCCL in place
of gloo -- on an 8-GPU node.  Rank 0 then renders the whole grid alone from the same seed and
compares.  Prints one JSON line per rank; exit code 0 = all assertions held.
"""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from conftest import make_lens
    from sdirt_amd import dist as sd

    ks, spp, dp = 33, 1024, (0.78, 1.44, 0.3, 0.5)
    # 15 points: the first shard (7) is far and on axis, the second (8) near and off axis --
    # the ranks' OWN slowest rays differ, the batch-global trip table must still be common
    far = torch.tensor([[0.0, 0.0, -20000.0]]).repeat(7, 1) + torch.linspace(0, 0.05, 7)[:, None] * torch.tensor([[1.0, 1.0, 0.0]])
    g = torch.linspace(-0.95, 0.95, 8)
    near = torch.stack([g, -g, torch.linspace(-200.0, -400.0, 8)], dim=-1)
    points = torch.cat([far, near])

    if world > 2:
        # the node size of BASELINE config 3: 8 ranks, 64 points -> 8 per rank, depths and field
        # positions spread so that the ranks' own trip tables differ
        gq = torch.Generator().manual_seed(64)
        points = torch.stack([(torch.rand(64, generator=gq) - 0.5) * 1.9, (torch.rand(64, generator=gq) - 0.5) * 1.9,
                              -(200 + torch.linspace(0, 1, 64) ** 2 * 19800)], -1)
    n_pts = points.shape[0]

    lens = make_lens("rf50mm", "cuda:0")
    sharded = sd.ShardedPSF.from_lens(lens, ks, dp=dp)
    torch.manual_seed(1234 + rank)        # only rank 0's generator may matter
    if rank == 0:
        torch.manual_seed(7)
    L, R = sharded.psf_volume(points, spp, gather=True)
    tables = {str(k): v.tolist() for k, v in lens.trips.cache.items()}
    out = {"rank": rank, "shape": list(L.shape), "tables": tables,
           "relaunches": lens.trips.relaunches}

    # EMPTY shards: 1 point over the ranks -> all but the last rank render nothing but must still
    # enter the mask reductions (and take the same decisions) or the others would hang
    L1, R1 = sharded.psf_volume(points[8:9], spp, gather=True)
    assert L1.shape == (1, ks, ks) and float(L1.max()) > 0.99
    out["empty_shard_ok"] = True

    gathered = [None] * world
    dist.all_gather_object(gathered, tables)
    assert all(t == gathered[0] for t in gathered), f"ranks verified different trip tables: {gathered}"

    if rank == 0:
        solo = make_lens("rf50mm", "cuda:0")
        torch.manual_seed(7)
        Ls, Rs = solo.psf_lr(points, ks=ks, spp=spp, dp=dp)
        solo_tables = {str(k): v.tolist() for k, v in solo.trips.cache.items()}
        dl = float((L - Ls).abs().max())
        dr = float((R - Rs).abs().max())
        out.update(max_abs_diff_L=dl, max_abs_diff_R=dr, solo_tables=solo_tables)
        assert L.shape == (n_pts, ks, ks)
        assert solo_tables == tables, (solo_tables, tables)
        assert dl <= 3e-6 and dr <= 3e-6, (dl, dr)
        # the table a rank would have verified from ITS OWN rays only (what makes the OR-reduce
        # matter): rendered with fresh lenses, no reduction
        own = []
        for a, b in sd.shard_bounds(n_pts, world):
            l2 = make_lens("rf50mm", "cuda:0")
            torch.manual_seed(7)
            l2.psf_lr(points[a:b], ks=ks, spp=spp, dp=dp)
            own.append({str(k): v.tolist() for k, v in l2.trips.cache.items()})
        out["own_tables_differ"] = any(o != own[0] for o in own[1:])
    # ---- the render loop of a sharded volume (sdirt_amd.volume.VolumeStepper): ONE library call per step, every rank
    # draws the same uniforms, masks OR-ed over the ranks in front of the device-side trip rule, one all-gather per step
    from sdirt_amd.volume import VolumeStepper
    a, b = sd.shard_bounds(n_pts, world)[rank]
    torch.manual_seed(99)                                   # the same seed on every rank
    lens_s = make_lens("rf50mm", "cuda:0")
    st = VolumeStepper(lens_s, points[a:b], n_pts, ks, spp, dp, gather=True, depth=2)
    vols = []
    for _ in range(3):
        st.step()
        st.fence()
        vols.append(st.volume().clone())
    out["stepper_relaunches"] = st.relaunches
    if rank == 0:
        solo = make_lens("rf50mm", "cuda:0")
        torch.manual_seed(99)
        worst = 0.0
        for v in vols:
            Ls, Rs = solo.psf_lr(points, ks=ks, spp=spp, dp=dp)
            assert tuple(v.shape) == (n_pts, 2, ks, ks)
            worst = max(worst, float((v[:, 0] - Ls).abs().max()), float((v[:, 1] - Rs).abs().max()))
        out["stepper_max_abs_diff"] = worst
        assert worst == 0.0, worst            # float64 tiles at ks 33: the sharded loop equals the solo calls bit for bit
    # ranks whose generators have diverged are told so instead of rendering different batches
    torch.manual_seed(5 + rank)
    try:
        st.step()
        st.fence()
        out["divergence_detected"] = False
    except RuntimeError as e:
        out["divergence_detected"] = "different pupil uniforms" in str(e)
    assert out["divergence_detected"]
    print(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
