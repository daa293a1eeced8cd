"""The render loop of a PSF volume on one rank: the same psf call step after step over a FIXED set of points
(`1_fit_psfnet.py`'s evaluation grid, BASELINE configs 2-4; a rank's shard of it on N GPUs), with as little host work
per step as the C ABI allows.

`Lensgroup.psf_lr(defer=True)` is the general call: any batch, any option, ~0.2 ms of Python per call (0.4 ms with the
collectives of a sharded batch) -- nothing for a 9 ms step, a third of the 1.2 ms step a rank of 8 renders.  Here
everything a step needs is built ONCE per in-flight slot -- output block, page-locked uniforms, device scratch
(control block, uniforms, pupil points), page-locked copy of the control block, events, the argument list of the
library call -- and a step is

    draw the uniforms (torch's CPU generator, the reference's order: optics.py:483-484 twice)
    ONE library call: upload, both pupil mappings, control block cleared, fused kernel      (sdirt_psf_call)
    [sharded: masks -> 0/1 lanes, all-reduce(MAX) over ranks, lanes -> masks]              (read-back stream)
    the reference's batch-wide trip rule evaluated ON THE DEVICE, control block copied to the host
    ... `depth` steps later: wait for that copy, read ONE status word; sharded + gather: the shard's ONE all-gather

Same draws, same kernels, same rule as psf_lr: tests/test_gpu_volume.py compares the two bit for bit.  A status word
that is not 0 (the speculated trip tables were not the reference's for this batch: never in steady state on a fixed
grid) sends that step through the corrected tables again -- on every rank alike, they all read the same reduced status.

Sharded batches: every rank draws the SAME uniforms from its own generator (seed them alike -- `torch.manual_seed`
before the loop, as every rank of a job does anyway); the sum of each step's uniforms rides along in the all-reduced block, and ranks whose streams have diverged raise instead of accepting the step.  That replaces the pupil
broadcast of ShardedPSF.psf_volume (one collective per step less); SURVEY.md §8e allows either.
"""
import ctypes as C
import time

import numpy as np
import torch

from . import _hostrng, _lib
from .basics import DEFAULT_WAVE, GEO_SPP


_STREAMS = {}      # (device index, role, i) -> the package's stream of that role on that device
_GROUPS = {}       # id of the parent process group -> (mask communicator, gather communicator)


def _stream(dev, role, i=0):
    key = (torch.device(dev).index or 0, role, i)
    if key not in _STREAMS:
        _STREAMS[key] = torch.cuda.Stream(dev)
    return _STREAMS[key]


def _groups(dist, parent):
    """Two communicators beside `parent` (every rank of it calls this at the same point of its program)."""
    key = id(parent) if parent is not None else 0
    if key not in _GROUPS or _GROUPS[key][2] is not dist.group.WORLD:
        ranks = None if parent is None else dist.get_process_group_ranks(parent)
        _GROUPS[key] = (dist.new_group(ranks), dist.new_group(ranks), dist.group.WORLD)
    return _GROUPS[key][:2]


class _Slot:
    __slots__ = ("out", "u_host", "u_bits", "scratch", "ctl_host", "lanes", "ev_kernel", "ev_read", "ev_gather", "args", "lane_args",
                 "stream", "k0", "k1", "used")


class VolumeStepper:
    """step() -> the [width, 2, ks, ks] block the step renders into (valid once the step has settled: `fence()`, or
    `depth` steps later); L = block[:n_local, 0], R = block[:n_local, 1].

    lens          Lensgroup on a GPU (trip_policy 'reference', pupil_mapping 'device')
    points_local  [n_local, 3] normalised points of this rank (device or host tensor)
    n_total       points of the whole volume (== n_local on one rank)
    group         process group of the ranks that share the batch (None: the default group when one is initialised)
    gather        sharded only: ONE all-gather of every step's [width, 2, ks, ks] blocks into `volume()` on every rank
    depth         steps kept in flight (default: ~20 ms of queued GPU work, at least 8)
    streams       render streams that consecutive steps alternate between (1, or 2: the next launch's workgroups fill
                  what is left of the previous launch's end)
    time_steps    keep a HIP-event pair around every step's library call (`kernel_ms()`)
    """

    def __init__(self, lens, points_local, n_total=None, ks=31, spp=GEO_SPP, dp=(0.78, 1.44, 0.3, 0.5), wvln=DEFAULT_WAVE,
                 group=None, gather=True, depth=None, streams=1, time_steps=False, force_collectives=False):
        import torch.distributed as dist
        from . import dist as sd
        lens._require_gpu()
        if lens.trip_policy != "reference" or lens.pupil_mapping != "device":
            raise ValueError("VolumeStepper renders with the reference's batch-wide trip counts and the device pupil mapping")
        self.lens, self.ks, self.spp, self.dp, self.wvln = lens, int(ks), int(spp), tuple(dp), wvln
        self.device = lens.device
        self.dist, self.sd = dist, sd
        live = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if live else 1
        self.rank = dist.get_rank(group) if live else 0
        self.multi = live and (self.world > 1 or force_collectives)
        self.group = group
        self.points = points_local.to(self.device, torch.float32).contiguous()
        self.n_local = int(self.points.shape[0])
        self.n_total = int(n_total if n_total is not None else self.n_local)
        self.width = max(b - a for a, b in sd.shard_bounds(self.n_total, self.world)) if self.multi else self.n_local
        if self.ks > _lib.MAX_KS:
            raise ValueError(f"ks <= {_lib.MAX_KS} (larger grids: Lensgroup.psf_lr)")
        self.gather = bool(gather and self.multi)
        #: None: dist.all_gather_shards' default (env SDIRT_GATHER_ALGO, else ONE RCCL all-gather); 'direct': one send and one
        #: receive per peer in a single group (may be switched between two fences, like `gather` and `gather_group`)
        self.gather_algo = None
        self.time_steps = bool(time_steps)
        h = self.h = _lib.lib()
        dev = self.device
        self.K = len(lens.surfaces)
        self.Sc = GEO_SPP
        self.n_u = 2 * (self.spp + self.Sc)
        if self.n_local and lens._spp_slices(self.n_local, self.spp) != 1:
            raise ValueError("VolumeStepper is for batches with one workgroup per point (sdirt_psf_spp_slices == 1); "
                             "few points with many samples: Lensgroup.psf_lr")
        # The render stream(s) are the package's own, never the caller's current stream: the legacy default stream
        # synchronises implicitly with every blocking stream of the process (RCCL's internal ones among them: +0.12 ms per
        # 2048-point step with nothing but the mask all-reduce in the loop, profiles/r06/stream_reuse.txt).  And they --
        # like the read-back and comm streams and the two communicators below -- are made ONCE per device / parent group
        # and shared by every stepper of the process: HIP maps streams onto its hardware queues round-robin and a queue
        # runs in order, so a process that keeps creating streams ends up with its render stream behind somebody's small
        # kernels and event waits (the same sweep of ten steppers: a rank of 8 at 0.89 instead of 0.93 of its share).
        self.render_streams = [_stream(dev, "render", i) for i in range(max(1, streams))]
        self.rb_stream = _stream(dev, "readback")
        # (`gather` may be switched between two fences: the stream and the communicator exist whenever the batch is sharded)
        self.comm_stream = _stream(dev, "comm") if self.multi else None
        # communicators of their own: the collectives of one process group share one internal stream
        self.mask_group, self.gather_group = _groups(dist, group) if self.multi else (None, None)
        with torch.cuda.device(dev):
            self.po = lens._points_to_object_now(self.points) if self.n_local else torch.empty((0, 3), device=dev)
        self.centers = torch.empty((max(self.n_local, 1), 2), dtype=torch.float32, device=dev)
        self.scratch_bytes = int(h.sdirt_psf_call_scratch_bytes(self.n_local, self.spp, self.Sc))
        self.wkey = round(float(wvln if wvln < 10 else wvln * 1e-3), 6)
        self.keys = [("psf", self.wkey, lens.precision), ("center", lens.precision)]
        # the rule is evaluated (and the control block copied out) on the read-back stream, behind the mask reduction of
        # a sharded batch: the render stream carries nothing but uploads and kernels
        self.flags = _lib.PSF_NORMALIZE | lens._psf_flags() | _lib.PSF_INTERLEAVED | _lib.PSF_ZERO_CTL | _lib.PSF_NO_VERIFY
        self.dpp = _lib.DpParams(*[float(v) for v in self.dp])
        self.handle, self.handle_c = lens.dev_lens(wvln), lens.dev_lens(DEFAULT_WAVE)
        self.tables = None
        self.steps = self.relaunches = self.gathers = 0
        self.t_step = self.t_wait = 0.0
        self.in_flight = []
        self.gather_events = []
        self._volume = None
        self._slots = []
        self.depth = depth if depth is not None else (8 if self.n_local >= 8192 else 16)
        self._discover()
        self._build_slots(self.depth + 1)
        torch.cuda.synchronize(dev)

    # ------------------------------------------------------------------ set-up
    def _discover(self):
        """The trip tables of this batch, found once through the general call (a launch with 10 trips everywhere, then
        the verified table): they are lens state from then on (TripPlanner.cache) and what every step speculates."""
        lens = self.lens
        block = torch.empty((self.width, 2, self.ks, self.ks), dtype=torch.float32, device=self.device)
        old = lens.mask_reduce
        if self.multi:
            lens.mask_reduce = lambda m: self.sd.reduce_masks_or(m, self.mask_group)
        try:
            state = torch.get_rng_state()          # discovery must not consume the caller's random stream
            lens.psf_lr(self.points, ks=self.ks, wvln=self.wvln, spp=self.spp, dp=self.dp, out=block[:self.n_local])
            torch.set_rng_state(state)
        finally:
            lens.mask_reduce = old
        self._take_tables()

    def _take_tables(self):
        curved = self.lens._curved()
        self.tables = [np.asarray(self.lens.trips.initial(k, curved), np.int32) for k in self.keys]
        self._tp = (C.c_int32 * self.K)(*[int(t) for t in self.tables[0]])
        self._tc = (C.c_int32 * self.K)(*[int(t) for t in self.tables[1]])
        for s in self._slots:
            self._bind(s)

    def _build_slots(self, n):
        dev = self.device
        while len(self._slots) < n:
            s = _Slot()
            s.out = torch.zeros((self.width, 2, self.ks, self.ks), dtype=torch.float32, device=dev)
            s.u_host = torch.empty(self.n_u, dtype=torch.float32, pin_memory=True)
            s.u_bits = s.u_host.numpy().view(np.uint32)
            s.scratch = torch.zeros(self.scratch_bytes, dtype=torch.uint8, device=dev)
            s.ctl_host = torch.zeros(_lib.CTL_WORDS, dtype=torch.int32, pin_memory=True)
            s.lanes = torch.zeros(_lib.CTL_LANES, dtype=torch.int32, device=dev)
            s.ev_kernel, s.ev_read, s.ev_gather = torch.cuda.Event(), torch.cuda.Event(), None
            s.k0 = torch.cuda.Event(enable_timing=True) if self.time_steps else None
            s.k1 = torch.cuda.Event(enable_timing=True) if self.time_steps else None
            s.stream = self.render_streams[len(self._slots) % len(self.render_streams)]
            s.used = False
            self._bind(s)
            self._slots.append(s)

    def _bind(self, s):
        """The argument list of the slot's library call: everything but the stream handle is fixed for the slot."""
        lens, P = self.lens, C.c_void_p
        pz, pr = lens.entrance_pupil()
        prc = lens.entrance_pupil(shrink_pupil=True)[1]
        L = s.out[:self.n_local] if self.n_local else s.out
        s.args = (self.handle, self.handle_c, P(self.po.data_ptr()), self.n_local, P(s.u_host.data_ptr()), self.spp, self.Sc,
                  float(pr), float(prc), float(pz), float(lens.d_sensor), float(lens.pixel_size), self.ks, C.byref(self.dpp),
                  self._tp, self._tc, self.flags, P(self.centers.data_ptr()), P(L.data_ptr()),
                  P(L.data_ptr() + 4 * self.ks * self.ks), P(s.scratch.data_ptr()), None,
                  _lib.StreamArg(s.stream.cuda_stream, self.device.index))
        rb = _lib.StreamArg(self.rb_stream.cuda_stream, self.device.index)
        s.lane_args = ([P(s.scratch.data_ptr()), 0, P(s.lanes.data_ptr()), rb],
                       (P(s.lanes.data_ptr()), self.handle, self._tp, self._tc, P(s.scratch.data_ptr()), P(s.ctl_host.data_ptr()), rb))

    # ------------------------------------------------------------------ the loop
    def step(self):
        t0 = time.perf_counter()
        s = self._slots[self.steps % len(self._slots)]
        self.steps += 1
        r = s.stream
        if s.ev_gather is not None:                      # an earlier gather may still read this block
            r.wait_event(s.ev_gather)
            s.ev_gather = None
        _hostrng.rand_into(s.u_host)
        s.used = True
        if s.k0 is not None:
            s.k0.record(r)
        _lib.check(self.h.sdirt_psf_call(*s.args))
        if s.k1 is not None:
            s.k1.record(r)
        s.ev_kernel.record(r)
        # [sharded: masks OR-ed over the ranks;] the rule evaluated on the block, the block copied to the host -- on the
        # read-back stream: the render stream goes straight on to the next step's upload and kernel
        rb = self.rb_stream
        rb.wait_event(s.ev_kernel)
        # (sharded: the sum of the step's uniforms rides along -- ranks whose generators have diverged are told so)
        s.lane_args[0][1] = int(s.u_bits.sum(dtype=np.uint64)) & 0x3FFFFFFF if self.multi else 0
        _lib.check(self.h.sdirt_ctl_to_lanes(*s.lane_args[0]))
        if self.multi:
            with torch.cuda.stream(rb):
                self.dist.all_reduce(s.lanes, op=self.dist.ReduceOp.MAX, group=self.mask_group)
        _lib.check(self.h.sdirt_ctl_from_lanes(*s.lane_args[1]))
        s.ev_read.record(rb)
        self.in_flight.append(s)
        self.t_step += time.perf_counter() - t0
        self.settle(self.depth)
        return s.out

    def settle(self, keep=0):
        """Accept all but the `keep` newest steps: wait for the step's control block, read its status word; gather."""
        while len(self.in_flight) > keep:
            s = self.in_flight.pop(0)
            t0 = time.perf_counter()
            s.ev_read.synchronize()
            self.t_wait += time.perf_counter() - t0
            t0 = time.perf_counter()
            w = s.ctl_host.numpy()
            if int(w[_lib.CTL_TAG]) + int(w[_lib.CTL_TAG + 1]) != 0x3FFFFFFF:
                raise RuntimeError("the ranks of this batch drew different pupil uniforms: seed their CPU generators alike "
                                   "(torch.manual_seed) and draw nothing else from them between steps")
            if int(w[_lib.CTL_STATUS]) != 0:
                self._correct(s)
            assert int(s.ctl_host[_lib.CTL_ANY_VALID]) == 1 or self.n_total == 0, "No sampled rays is valid."   # optics.py:902
            if self.gather:
                buf = self._gather_buffer()
                cs = self.comm_stream
                cs.wait_event(s.ev_kernel)
                with torch.cuda.stream(cs):
                    g0 = torch.cuda.Event(enable_timing=True) if self.time_steps else None
                    if g0 is not None:
                        g0.record(cs)
                    self._volume = self.sd.all_gather_shards(s.out, self.n_total, self.world, self.gather_group, out=buf,
                                                             algo=self.gather_algo, padded=True)
                    done = torch.cuda.Event(enable_timing=self.time_steps)
                    done.record(cs)
                    if g0 is not None:
                        self.gather_events.append((g0, done))
                s.ev_gather = done
                self.gathers += 1
            self.t_step += time.perf_counter() - t0

    def _gather_buffer(self):
        bufs = self.__dict__.setdefault("_gather_bufs", [])
        if len(bufs) < 2:
            bufs.append(torch.empty((self.world * self.width, 2, self.ks, self.ks), dtype=torch.float32, device=self.device))
            return bufs[-1]
        return bufs[self.gathers % 2]

    def _correct(self, s):
        """The step ran with tables that are not the reference's for its batch (status word != 0; every rank reads the
        same reduced block and gets here together): run it again with the tables the device derived, same pupil points
        (they still stand in the slot's scratch), until the rule accepts; the new tables are what later steps speculate."""
        lens, h, P = self.lens, self.h, C.c_void_p
        K, S, Sc = self.K, self.spp, self.Sc
        pz = lens.entrance_pupil()[0]
        n = self.n_u
        xy = s.scratch.data_ptr() + self.scratch_bytes - 4 * n
        L = s.out[:self.n_local] if self.n_local else s.out
        with torch.cuda.stream(s.stream):
            st = _lib.StreamArg(s.stream.cuda_stream, self.device.index)
            for _ in range(3 * K + 3):
                w = s.ctl_host.numpy().view(np.uint32)
                unpack = lambda off: [int(np.int8((int(w[off + (k >> 2)]) >> ((k & 3) * 8)) & 0xFF)) for k in range(K)]
                tp, tc = unpack(_lib.CTL_TRIPS2), unpack(_lib.CTL_TRIPS2 + 16)
                self.relaunches += 1
                lens.trips.relaunches += 1
                ctp, ctc = (C.c_int32 * K)(*tp), (C.c_int32 * K)(*tc)
                ctl = s.scratch[:4 * _lib.CTL_WORDS].view(torch.int32)
                ctl.zero_()
                base = s.scratch.data_ptr()
                if self.n_local:
                    _lib.check(h.sdirt_psf_lr_centered(
                        self.handle, self.handle_c, P(self.po.data_ptr()), self.n_local, P(xy), P(xy + 4 * S), S,
                        P(xy + 8 * S), P(xy + 8 * S + 4 * Sc), Sc, float(pz), float(lens.d_sensor), float(lens.pixel_size),
                        self.ks, C.byref(self.dpp), ctp, ctc, self.flags & ~(_lib.PSF_ZERO_CTL | _lib.PSF_NO_VERIFY),
                        P(self.centers.data_ptr()), P(base + 4 * _lib.CTL_ANY_VALID), P(L.data_ptr()),
                        P(L.data_ptr() + 4 * self.ks * self.ks), P(base + 4 * _lib.CTL_MASKS), P(base + 4 * (_lib.CTL_MASKS + 64)), st))
                _lib.check(h.sdirt_ctl_to_lanes(P(base), 0, P(s.lanes.data_ptr()), st))
                if self.multi:
                    self.dist.all_reduce(s.lanes, op=self.dist.ReduceOp.MAX, group=self.mask_group)
                _lib.check(h.sdirt_ctl_from_lanes(P(s.lanes.data_ptr()), self.handle, ctp, ctc, P(base), P(s.ctl_host.data_ptr()), st))
                s.stream.synchronize()
                if int(s.ctl_host[_lib.CTL_STATUS]) == 0:
                    for key, t in zip(self.keys, (tp, tc)):
                        lens.trips.learn(key, np.asarray(t, np.int32))
                    self._take_tables()
                    s.ev_kernel.record(s.stream)
                    return
        raise RuntimeError("Newton trip-table speculation did not converge")  # pragma: no cover

    def fence(self):
        """Everything enqueued so far is rendered, accepted and (sharded + gather) gathered; ranks leave together."""
        self.settle(0)
        torch.cuda.synchronize(self.device)
        if self.multi:
            self.dist.barrier(group=self.group)
            torch.cuda.synchronize(self.device)

    def timed(self, k):
        """k steps between two fences -> wall seconds, MAX over ranks."""
        self.fence()
        t0 = time.perf_counter()
        for _ in range(k):
            self.step()
        self.fence()
        dt = time.perf_counter() - t0
        if self.multi:
            tmax = torch.tensor([dt], dtype=torch.float64, device=self.device)
            self.dist.all_reduce(tmax, op=self.dist.ReduceOp.MAX, group=self.group)
            dt = float(tmax.item())
        return dt

    def volume(self):
        """The most recently gathered [n_total, 2, ks, ks] volume (after fence())."""
        return self._volume

    def verify_gather(self):
        """After fence(), gathering on: every rank's copy of the most recently gathered volume holds every rank's block of the
        last step, bit for bit (dist.gathered_volume_holds_every_shard: checksums of the bit patterns, two tiny collectives).
        -> the same bool on every rank; None when nothing was gathered."""
        if not (self.gather and self.gathers and self._volume is not None):
            return None
        s = self._slots[(self.steps - 1) % len(self._slots)]
        return self.sd.gathered_volume_holds_every_shard(self._volume, s.out[:self.n_local], self.n_total, self.group)

    def kernel_ms(self):
        """Mean HIP-event time of the slots' most recent library call (upload of 48 KB + pupil mapping + fused kernel)
        on its render stream (time_steps=True; after fence()).  With two render streams consecutive launches overlap on
        the chip and a pair spans both."""
        ev = [(s.k0, s.k1) for s in self._slots if s.k0 is not None and s.used and s.k1.query()]
        return float(np.mean([a.elapsed_time(b) for a, b in ev])) if ev else None

    def gather_ms(self):
        """Mean HIP-event time of the all-gathers since the last call (time_steps=True, after fence()), or None."""
        ev, self.gather_events = self.gather_events, []
        return float(np.mean([a.elapsed_time(b) for a, b in ev])) if ev else None

    def reset_counters(self):
        self.t_step = self.t_wait = 0.0
        self.gather_events = []
