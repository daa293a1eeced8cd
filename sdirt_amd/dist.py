"""One process per GPU: shard a PSF volume over ranks, gather it back over RCCL.

The reference has no multi-GPU PSF path (1_fit_psfnet.py:7 pins one GPU).
Point sources are independent, so the (x, y, z) grid is partitioned
contiguously over the ranks; what the ranks must share to stay equal to ONE
reference call over the whole grid is

  * the pupil sample set (one set for all points, optics.py:482-490): rank 0
    draws the uniforms from torch's CPU generator, they are broadcast;
  * the batch-global Newton trip counts (surfaces.py:547): the per-surface
    convergence masks are OR-reduced over the ranks before the trip table is
    verified (newton.py), so every rank takes the same decision;
  * optionally the result: ONE all-gather of the [N/world, 2, ks, ks] shards
    (`torch.distributed` backend "nccl" is RCCL on ROCm; xGMI is point-to-point,
    a gather of a few hundred MB per rank keeps all 7 links of a GPU busy).  A
    rank renders its shard straight into that block (SDIRT_PSF_INTERLEAVED: the
    kernel writes point n's left grid at [n, 0] and its right grid at [n, 1]),
    and L = volume[:, 0], R = volume[:, 1] are views of the gathered array: no
    staging copy on either side of the collective.

Nothing here touches the data path on a single GPU.
"""
import torch
import torch.distributed as dist

from .basics import GEO_SPP


#: True: issue every collective even on a world-1 process group (bench.py --workload sweep: the per-step host work and
#: collective launches of a rank of an N-GPU run, measured on the one GPU a box has).  Default: a lone rank skips them.
FORCE_COLLECTIVES = False


_BITS = {}      # device -> arange(11): the bit lanes of reduce_masks_or


def _alone(group):
    return dist.get_world_size(group) == 1 and not FORCE_COLLECTIVES


def shard_bounds(n, world):
    """Contiguous, balanced partition of range(n): list of (start, stop)."""
    return [(n * r // world, n * (r + 1) // world) for r in range(world)]


def reduce_masks_or(mask, group=None):
    """Bitwise OR of int32 masks over ranks.  NCCL/RCCL has no BOR: expand the
    (at most 11 used) bits into 0/1 lanes and take MAX."""
    if not (dist.is_available() and dist.is_initialized()) or _alone(group):
        return mask
    bits = _BITS.get(mask.device)
    if bits is None:
        bits = _BITS[mask.device] = torch.arange(0, 11, device=mask.device, dtype=torch.int32)
    lanes = ((mask.to(torch.int32).unsqueeze(-1) >> bits) & 1).contiguous()
    dist.all_reduce(lanes, op=dist.ReduceOp.MAX, group=group)
    return (lanes << bits).sum(-1).to(mask.dtype)


def broadcast_uniforms(spp, device, n_center=GEO_SPP, group=None, src=0):
    """Rank `src` draws rand(spp) x2 then rand(n_center) x2 from the CPU default
    generator -- the reference's draw order inside one psf_diff call
    (optics.py:483-484 twice) -- every rank receives the same numbers."""
    total = 2 * spp + 2 * n_center
    if dist.get_rank(group) == src:
        u = torch.cat([torch.rand(spp), torch.rand(spp), torch.rand(n_center),
                       torch.rand(n_center)]).to(device)
    else:
        u = torch.empty(total, dtype=torch.float32, device=device)
    if not _alone(group):
        dist.broadcast(u, src=src, group=group)
    return u[:spp], u[spp:2 * spp], u[2 * spp:2 * spp + n_center], u[2 * spp + n_center:]


def broadcast_pupil_points(lens, spp, n_center=GEO_SPP, group=None, src=0, stream=None):
    """Rank `src` draws and maps the primary and the chief-ray pupil sample sets exactly as a
    single-GPU Lensgroup.psf_lr call would (same RNG order, same mapping); every rank
    receives the same points: (x2, y2, xc, yc).

    stream: a side stream for the whole of it (draw, mapping, broadcast) -- it depends on nothing the caller has
    queued, so it can run beside the previous step's kernel; the caller's stream waits for its event only.  Give the
    broadcast its own `group` then: collectives of one process group share one internal stream, and behind the
    previous step's mask all-reduce the broadcast would wait for that step's kernel after all."""
    main = torch.cuda.current_stream(lens.device) if stream is not None else None

    def run():
        total = 2 * spp + 2 * n_center
        if dist.get_rank(group) == src:
            # one draw, one upload, one device block [x2 | y2 | xc | yc] -- value for value the reference's four
            # consecutive draws (Lensgroup._pupil_samples_pair) -- which is the broadcast buffer as it stands
            _, pr = lens.entrance_pupil()
            parts = lens._pupil_samples_pair(spp, pr, n_center, pr * 0.25, side_stream=False)
            base = parts[0]._base
            if base is not None and base.numel() == total and base.is_contiguous() and parts[0].data_ptr() == base.data_ptr():
                buf = base
            else:                                              # pupil_mapping='host': four separate tensors
                buf = torch.cat(parts)
        else:
            buf = torch.empty(total, dtype=torch.float32, device=lens.device)
        if not _alone(group):
            dist.broadcast(buf, src=src, group=group)
        return buf
    if stream is None:
        buf = run()
    else:
        with torch.cuda.stream(stream):
            buf = run()
            done = torch.cuda.Event()
            done.record(stream)
        main.wait_event(done)
        buf.record_stream(main)
    return (buf[:spp], buf[spp:2 * spp], buf[2 * spp:2 * spp + n_center],
            buf[2 * spp + n_center:])


def all_gather_shards(local, n_total, world, group=None, out=None, algo=None, padded=False):
    """local: [n_local, ...] shard of a contiguous partition (shard_bounds) ->
    [n_total, ...] on every rank.  Shards are padded to the largest one so a
    single all_gather_into_tensor moves everything (padded=True: `local` already
    has the largest shard's width, its own rows first -- ShardedPSF.shard_buffer).

    algo (or env SDIRT_GATHER_ALGO): 'allgather' (default) = one RCCL all-gather;
    'direct' = every rank posts one send and one receive per peer in a single
    batch_isend_irecv group -- xGMI is a full mesh of point-to-point links, so the
    seven transfers of a GPU can use its seven links at once instead of being
    paced by a ring.  Same result; to be compared on an 8-GPU node."""
    import os
    algo = algo or os.environ.get("SDIRT_GATHER_ALGO", "allgather")
    bounds = shard_bounds(n_total, world)
    width = max(b - a for a, b in bounds)
    tail = tuple(local.shape[1:])
    if padded and local.shape[0] != width:
        raise ValueError(f"padded shard has {local.shape[0]} rows, the largest shard of the partition {width}")
    if local.shape[0] != width:
        pad = torch.zeros((width,) + tail, dtype=local.dtype, device=local.device)
        pad[:local.shape[0]] = local
        local = pad
    if out is None or out.shape != (world * width,) + tail:
        out = torch.empty((world * width,) + tail, dtype=local.dtype, device=local.device)
    if algo == "direct" and world > 1:
        rank = dist.get_rank(group)
        local = local.contiguous()
        out[rank * width:(rank + 1) * width].copy_(local)
        ops = []
        for peer in range(world):
            if peer == rank:
                continue
            ops.append(dist.P2POp(dist.isend, local, peer, group))
            ops.append(dist.P2POp(dist.irecv, out[peer * width:(peer + 1) * width], peer, group))
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    elif local.is_cuda and dist.get_backend(group) == "gloo":
        # gloo has no CUDA all-gather: stage through the host (dry runs only, see bench.py)
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, local.contiguous().cpu(), group=group)
        out.copy_(host)
    else:
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    if all(b - a == width for a, b in bounds):
        return out
    return torch.cat([out[r * width:r * width + (b - a)] for r, (a, b) in enumerate(bounds)])


def gathered_volume_holds_every_shard(volume, own, n_total, group=None):
    """Did the gather deliver every rank's rows to every rank?  `own` = the rows THIS rank contributed ([n_local, ...], may be
    empty), `volume` = its copy of the gathered [n_total, ...] volume.  Each rank sums the BIT PATTERNS of `own` (int64 sum of
    the int32 view: exact, order-free), the sums travel in one small all-reduce, every rank compares them with the sums of the
    corresponding rows of its copy, and the verdicts are reduced: True on every rank iff every copy holds every shard bit for
    bit.  Two collectives of a few bytes, whatever the volume's size."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = volume.device
    bits = lambda t: t.reshape(-1).view(torch.int32).sum(dtype=torch.int64)  # noqa: E731
    sums = torch.zeros(world, dtype=torch.int64, device=dev)
    if own.numel():
        sums[rank] = bits(own.contiguous())
    dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
    mine = torch.stack([bits(volume[a:b]) if b > a else torch.zeros((), dtype=torch.int64, device=dev)
                        for a, b in shard_bounds(n_total, world)])
    bad = torch.zeros(1, dtype=torch.int64, device=dev)
    bad[0] = int(not bool((mine == sums).all()))
    dist.all_reduce(bad, op=dist.ReduceOp.SUM, group=group)
    return int(bad.item()) == 0


class ShardedPSF:
    """psf_lr over a point grid partitioned across the ranks of `group`.

    render(points_local, u) -> (L, R) is normally Lensgroup-backed
    (`from_lens`); tests substitute a CPU stand-in to exercise the partition /
    broadcast / gather logic under gloo.  ks given: render(points_local, u, out=block)
    fills a [n_local, 2, ks, ks] block in place (what the Lensgroup-backed one does) and
    the gathered volume is assembled from those blocks by one collective.
    """

    def __init__(self, render, device, group=None, ks=None):
        self.render, self.device, self.group = render, torch.device(device), group
        self.lens, self.ks = None, ks
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    @classmethod
    def from_lens(cls, lens, ks, wvln=0.589, dp=(0.78, 1.44, 0.3, 0.5), group=None):
        lens.mask_reduce = lambda m: reduce_masks_or(m, group)

        def render(points_local, pupil, out=None, defer=False):
            # defer=True: every rank enqueues the same kernels and collectives in the same order
            # and takes the same re-launch decision later (the masks are reduced over ranks)
            # out: (L, R) [n, ks, ks] each, or ONE [n, 2, ks, ks] block (the form the gather moves)
            return lens.psf_lr(points_local, ks=ks, wvln=wvln, dp=dp, pupil_xy=(pupil[0], pupil[1]),
                               center_pupil_xy=(pupil[2], pupil[3]), out=out, defer=defer)
        self = cls(render, lens.device, group, ks=ks)
        self.lens = lens
        return self

    def shard_buffer(self, n_total):
        """The [width, 2, ks, ks] block this rank renders into and the gather sends: `width` = the largest shard of
        the partition (so that one all_gather_into_tensor moves everything), this rank's points in its first rows
        (rows beyond them -- uneven partitions only -- are zero)."""
        bounds = shard_bounds(n_total, self.world)
        width = max(b - a for a, b in bounds)
        n_local = bounds[self.rank][1] - bounds[self.rank][0]
        block = torch.empty((width, 2, self.ks, self.ks), dtype=torch.float32, device=self.device)
        if n_local < width:
            block[n_local:].zero_()
        return block

    def gather(self, block, n_total, out=None, group=None):
        """ONE collective: every rank's [width, 2, ks, ks] block (`shard_buffer`, filled by `render(..., out=block[:n])`)
        -> the [n_total, 2, ks, ks] volume on every rank; L = volume[:, 0], R = volume[:, 1] are views of it.
        group: the communicator the gather runs on (bench.py gives it its own, so that it does not hold back the next
        step's small broadcast on the default group's stream)."""
        return all_gather_shards(block, n_total, self.world, self.group if group is None else group, out=out,
                                 padded=True)

    def local_slice(self, n_total):
        return shard_bounds(n_total, self.world)[self.rank]

    def psf_volume(self, points, spp, gather=True):
        """points [N,3] (the same on every rank) -> (L, R) of all N points when
        gather, else of this rank's shard."""
        n_total = points.shape[0]
        a, b = self.local_slice(n_total)
        if self.lens is not None:
            u = broadcast_pupil_points(self.lens, spp, group=self.group)
        else:
            u = broadcast_uniforms(spp, self.device, group=self.group)
        from . import _lib
        # grids above SDIRT_MAX_KS go through the staged chain, which renders two tensors (Lensgroup.psf_lr refuses
        # out=block for them -- and a rank that raises before the collective would leave its peers hanging in it)
        two_tensors = self.ks is None or (self.lens is not None and self.ks > _lib.MAX_KS)
        if two_tensors or not gather or self.world == 1:
            L, R = self.render(points[a:b], u)
            if not gather or self.world == 1:
                return L, R
            block = torch.stack((L, R), dim=1)              # (a CPU stand-in, or the staged chain, renders two tensors)
        else:
            # the kernel writes the shard straight into the block the collective sends: no torch.stack
            block = self.shard_buffer(n_total)
            self.render(points[a:b], u, out=block[:b - a])
        full = all_gather_shards(block, n_total, self.world, self.group)
        return full[:, 0], full[:, 1]
