"""Host logic that reproduces the reference's batch-global Newton trip counts.

In the reference, Aspheric._newtons_method iterates
`while (abs(ft) > 50e-6).any() and it < 10` over the WHOLE ray tensor
(deeplens/surfaces.py:547): every ray of a call runs the same number of trips
on a surface, and that number depends on the slowest ray of the batch.  A
per-ray GPU kernel cannot ask "has any ray in 67 million not converged?" after
every iteration without a grid-wide synchronisation, so the kernels take the
trip table as an ARGUMENT and report, per surface, a bitmask OR-ed over all rays
(bit j: some ray still had |f(t)| > 50e-6 in trip j).  From the mask the host
can tell whether the table it speculated is exactly the one the reference's
loop would have produced for this batch:

    T is right for a surface  <=>  bits 1..T-1 are set and (bit T is clear or T == 10),
    provided every surface upstream was right (their output rays feed this one).

`TripPlanner.run` speculates (the table verified most often for this kind of
call so far, or 10 trips everywhere on first use), launches, verifies, and
re-launches only when the verification fails;
in steady state (same lens, similar batches) that is one launch per call plus
one K-word readback -- the reference synchronises on every Newton iteration.
"""
import numpy as np

NEWTON_MAXITER = 10


def first_clear_bit(mask, upto):
    """Smallest j in 1..upto whose bit is clear in mask, or None."""
    for j in range(1, upto + 1):
        if not (mask >> j) & 1:
            return j
    return None


def verify(trips, masks, order, curved, aggressive=False):
    """Check a speculated trip table against the convergence masks it produced.

    trips[k]  : trips run on surface k
    masks[k]  : OR over all rays of (1<<j) for trips j where |f| > 50e-6
    order     : surface indices in traversal order
    curved[k] : False for planes (no Newton solve, surfaces.py:409)

    Returns (ok, new_trips).  When not ok, new_trips holds the corrected count
    for the first wrong surface and best guesses (from the same masks) for the
    surfaces after it.  aggressive: a surface whose every trip was still open is
    re-run with the full 10 trips (which always reveals the exact count) instead
    of one more trip -- used from the second failed round on, so that a table far
    below the truth is reached in a bounded number of rounds.
    """
    trips = np.asarray(trips, np.int32).copy()
    masks = [int(m) for m in masks]
    new = trips.copy()
    failed = False
    for k in order:
        if not curved[k]:
            new[k] = 0
            continue
        T = int(trips[k])
        j = first_clear_bit(masks[k], T) if T >= 1 else None
        if not failed:
            # the loop always runs once (ft starts at 1e5), so T >= 1
            if T >= 1 and (j == T or (j is None and T == NEWTON_MAXITER)):
                continue                      # exactly what the reference would run
            failed = True
        # first wrong surface: exact correction when a clear bit was seen; when every trip
        # that ran still had an open ray, one more trip is by far the likeliest answer (a
        # batch flips between neighbouring counts when its slowest ray changes).  Downstream
        # surfaces: their masks came from slightly different rays, use them as the next guess.
        new[k] = j if j is not None else (NEWTON_MAXITER if aggressive else min(max(T, 0) + 1, NEWTON_MAXITER))
    return (not failed), new


class TripPlanner:
    """Caches one speculated trip table per key and runs launch/verify rounds.

    launch(trips) must enqueue the kernels with that table (after zeroing its
    mask buffer) and return the masks as a length-K integer array (this is the
    synchronisation point; a distributed caller reduces the masks over ranks
    with a bitwise OR before returning them).
    """

    def __init__(self):
        self.cache = {}          # key -> last verified table
        self.votes = {}          # key -> {table: times verified}
        self.launches = 0
        self.relaunches = 0          # rounds the HOST had to launch again
        self.device_relaunches = 0   # corrections the device made on its own (sdirt_psf_lr_verified)
        self._accepted = set()       # (trip table, masks) pairs verify() has already accepted

    def initial(self, key, curved):
        """Batches of one workload flip between a few neighbouring tables (the slowest ray of
        the batch decides); the table that was right most often is the better bet than the
        last one (ties: the most recent)."""
        t = self.cache.get(key)
        if t is None or len(t) != len(curved):
            return np.where(np.asarray(curved), NEWTON_MAXITER, 0).astype(np.int32)
        votes = self.votes.get(key, {})
        best = max(votes.values(), default=0)
        if votes.get(tuple(int(x) for x in t), 0) < best:
            t = next(tab for tab, n in votes.items() if n == best)
        return np.asarray(t, np.int32).copy()

    def learn(self, key, trips):
        self.cache[key] = np.asarray(trips, np.int32).copy()
        votes = self.votes.setdefault(key, {})
        tab = tuple(int(x) for x in trips)
        votes[tab] = votes.get(tab, 0) + 1
        if votes[tab] >= 64:     # keep the statistics adaptive
            for k in list(votes):
                votes[k] //= 2

    def run(self, key, curved, order, launch, max_rounds=None):
        K = len(curved)
        trips = self.initial(key, curved)
        rounds = max_rounds if max_rounds is not None else 3 * K + 3
        for i in range(rounds):
            masks = launch(trips)
            self.launches += 1
            ok, new = verify(trips, masks, order, curved, aggressive=i > 0)
            if ok:
                self.learn(key, trips)
                return trips
            self.relaunches += 1
            trips = new
        raise RuntimeError("Newton trip-table speculation did not converge")  # pragma: no cover

    def run_many(self, keys, curved, order, launch, max_rounds=None, first=None, done=None):
        """Several independent passes verified in ONE launch/readback round (the chief-ray pass
        and the primary pass of a psf call).  launch(list_of_trip_tables) -> list of mask
        arrays, one per key.  All passes are re-launched together when any table was wrong.
        first = (tables, masks): a round that was already launched with `tables` (speculated by
        `initial`) and has reported `masks` -- deferred verification, see Lensgroup.psf_lr.
        done = [(tables, masks), ...]: SEVERAL rounds the device has already run on its own
        (sdirt_psf_lr_verified: round 1 with the speculated tables, round 2 with the tables the
        device derived from round 1's masks).  Each is checked here with the same rule; the
        device's correction must be the one this planner would have made."""
        K = len(curved)
        if first is not None:
            done = [first]
        done = list(done or [])
        tables = [self.initial(k, curved) for k in keys] if not done else done[0][0]
        rounds = max_rounds if max_rounds is not None else 3 * K + 3
        identity = tuple(order) == tuple(range(K))
        curved_sig = tuple(bool(c) for c in curved)
        for i in range(rounds):
            masks = done[i][1] if i < len(done) else launch(tables)
            self.launches += 1
            # a caller rendering the same kind of batch over and over reports the same masks for the
            # same tables: what verify() accepted once it accepts again (for the same surface kinds: the
            # planner outlives edits of the prescription it serves)
            sig = None
            if identity:
                # (the arrays' bytes: ~1 us per pass; element by element this line was 0.12 ms of every psf call)
                sig = (curved_sig, tuple((np.asarray(t, dtype=np.int64).tobytes(), np.asarray(m, dtype=np.int64).tobytes())
                                         for t, m in zip(tables, masks)))
                if sig in self._accepted:
                    for k, t in zip(keys, tables):
                        self.learn(k, t)
                    return tables
            results = [verify(t, m, order, curved, aggressive=i > 0) for t, m in zip(tables, masks)]
            if all(ok for ok, _ in results):
                if sig is not None and len(self._accepted) < 4096:
                    self._accepted.add(sig)
                for k, t in zip(keys, tables):
                    self.learn(k, t)
                return tables
            tables = [t if ok else new for t, (ok, new) in zip(tables, results)]
            if i + 1 < len(done):
                # the device re-rendered on its own: its tables are the first-round correction
                self.device_relaunches += 1
                dev = done[i + 1][0]
                if not all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(dev, tables)):
                    raise RuntimeError(f"device-side trip correction {dev} differs from the host's {tables}")
            else:
                self.relaunches += 1
        raise RuntimeError("Newton trip-table speculation did not converge")  # pragma: no cover
