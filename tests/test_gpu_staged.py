"""The STAGED ray pipeline (rays in HBM as a point-major SoA bundle; deeplens/optics.py:460-494, :638-664,
deeplens/monte_carlo.py:9-68) through the C ABI: the vectorised sample / propagate kernels against their
one-ray-per-thread forms bit for bit, forward_integral with the grids in LDS against the CPU oracle's splat
and against the fused kernel, for every launch shape its planner produces (points per workgroup 1..8,
ragged point and sample counts, the spp axis cut or not, float tiles and the HBM fallback above SDIRT_MAX_KS)."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import load_state, make_lens

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
DP = (0.78, 1.44, 0.3, 0.5)


@pytest.fixture(scope="module")
def lens():
    return make_lens("rf50mm", DEV)


def _points(n, seed=3):
    g = torch.Generator().manual_seed(seed)
    p = torch.rand(n, 3, generator=g)
    p[:, :2] = p[:, :2] * 1.8 - 0.9
    p[:, 2] = -(200.0 + p[:, 2] ** 2 * 19800.0)
    return p


def _pupil(lens, spp, seed):
    torch.manual_seed(seed)
    return lens._pupil_samples(spp, lens.entrance_pupil()[1])


def _sample(lens, po, x2, y2, n=None):
    from sdirt_amd import _lib
    from sdirt_amd.basics import Ray, dptr, stream_ptr
    n = po.shape[0] if n is None else n
    ray = Ray.empty((x2.shape[0], n), 0.589, DEV)
    _lib.check(_lib.lib().sdirt_sample_rays(dptr(po), n, dptr(x2), dptr(y2), x2.shape[0],
                                            float(lens.entrance_pupil()[0]), ray.c_rays(), stream_ptr(torch.device(DEV))))
    return ray


@pytest.mark.parametrize("n", [256, 51])
def test_vectorised_sampler_and_propagate_equal_the_scalar_kernels_bit_for_bit(lens, n):
    """spp % 4 == 0 takes the dwordx4 kernels (four consecutive samples of a point per thread); the same points
    with the first spp - 1 samples take the one-ray-per-thread kernels: samples [:spp - 1] must be the same bits."""
    spp = 192
    pts = _points(n)
    po = lens._points_to_object(pts)
    x2, y2 = _pupil(lens, spp, 5)
    a = _sample(lens, po, x2, y2)
    b = _sample(lens, po, x2[:spp - 1], y2[:spp - 1])
    assert a.shape == (spp, n) and b.shape == (spp - 1, n)
    # point-major storage: ray (s, n) is element n * spp + s of every component array
    assert torch.equal(a.soa[0, :a.numel].view(n, spp).t(), a.o[..., 0])

    def bits(r, k):
        return torch.stack([r._field(c)[:k] for c in range(7)]).contiguous().view(torch.int32)
    assert torch.equal(bits(a, spp - 1), bits(b, spp - 1))
    assert torch.all(a.ra == 1) and torch.all(a.obliq == 1)
    a.propagate_to(-3.25)
    b.propagate_to(-3.25)
    assert torch.equal(bits(a, spp - 1), bits(b, spp - 1))
    # the sampler against its definition (optics.py:486-494): d = normalize(pupil point - o)
    o = po.double().cpu()
    tgt = torch.stack([x2.double().cpu()[:, None].expand(spp, n), y2.double().cpu()[:, None].expand(spp, n),
                       torch.full((spp, n), float(np.float32(lens.entrance_pupil()[0])), dtype=torch.float64)], -1)
    d = tgt - o[None]
    d = d / d.norm(dim=-1, keepdim=True)
    a2 = _sample(lens, po, x2, y2)
    assert float((a2.d.double().cpu() - d).abs().max()) < 1.5e-7


def _oracle_splat(oracle, st, ray, cen, ks, dp, n):
    """forward_integral of point n's rays by the CPU oracle's splat."""
    soa = np.stack([ray._field(c)[:, n].cpu().numpy() for c in range(7)])
    o = np.ascontiguousarray(soa[0:3].T[:, None, :])
    d = np.ascontiguousarray(soa[3:6].T[:, None, :])
    lg, rg = oracle.forward_integral(o, d, np.ascontiguousarray(soa[6][:, None]), st["pixel_size"], ks,
                                     cen[n:n + 1].cpu().numpy(), list(dp))
    return lg[0], rg[0]


@pytest.mark.parametrize("n,spp,ks,dp", [
    (300, 1024, 21, DP),          # one point per workgroup, one pass
    (1100, 200, 9, DP),           # two points per workgroup (fewer samples than threads), N % 2 != 0 ... ragged rows
    (520, 512, 65, DP),           # two points per workgroup on 135 KB of double accumulators
    (130, 50, 65, DP),            # eight points would not fit: LDS caps the points per workgroup
    (5, 1024, 120, DP),           # two double tiles no longer fit: float accumulators
    (3, 4096, 65, DP),            # a handful of points: the spp axis cut into slices, tiles added to the output
    (37, 700, 33, None),          # param_list=None: L only, R stays zero
    (130, 512, 21, (0.78, 1.44, 0.3, 0.6)),   # big-radius microlens branch
    (2, 2048, 150, DP),           # above SDIRT_MAX_KS: grids in HBM
])
def test_forward_integral_tiles_match_the_oracle_splat(lens, oracle, n, spp, ks, dp):
    from sdirt_amd import forward_integral_lr
    st = load_state("rf50mm")
    pts = _points(n, seed=n)
    po = lens._points_to_object(pts)
    x2, y2 = _pupil(lens, spp, 7)
    ray = _sample(lens, po, x2, y2)
    lens.trace2sensor(ray)
    g = torch.Generator().manual_seed(1)
    # centres a little off the spot (the window test gets rays on both sides) and a non-0/1 weight on some rays
    w, ox, oy = ray.ra, ray._field(0), ray._field(1)
    cen = torch.stack([-(ox * w).sum(0) / (w.sum(0) + 1e-9), -(oy * w).sum(0) / (w.sum(0) + 1e-9)], 1)
    cen = (cen + (torch.rand(n, 2, generator=g).to(DEV) - 0.5) * 0.1).contiguous()
    ray.ra = w * torch.where(torch.rand(spp, n, generator=g) < 0.2, 0.625, 1.0).to(DEV)
    lg, rg = forward_integral_lr(ray, lens.pixel_size, ks, cen, None if dp is None else list(dp) + ["l"])
    assert torch.isfinite(lg).all() and torch.isfinite(rg).all()
    if dp is None:
        assert float(rg.abs().max()) == 0.0
    worst = 0.0
    for k in sorted(set([0, 1, n // 2, n - 2, n - 1]) & set(range(n))):
        lo, ro = _oracle_splat(oracle, st, ray, cen, ks, dp if dp is not None else DP, k)
        scale = max(float(lo.max()), 1e-6)
        worst = max(worst, float(np.abs(lg[k].cpu().numpy() - lo).max()) / scale)
        if dp is not None:
            worst = max(worst, float(np.abs(rg[k].cpu().numpy() - ro).max()) / max(float(ro.max()), 1e-6))
    print(f"forward_integral n={n} spp={spp} ks={ks}: max |HIP - oracle| / peak = {worst:.2e}")
    # the segment areas of the default small-radius branch are a polynomial here and acos/sin in the oracle
    # (3e-7 absolute per weight); sums of up to 4096 fp32 terms in a different order
    assert worst < 2.4e-6                       # measured 2.0e-7 ... 7.6e-7
    # energy: what the window keeps of every point is what was splat (bilinear taps sum to the weight)
    tot = (lg.double().sum((1, 2)) + rg.double().sum((1, 2))).cpu()
    assert torch.isfinite(tot).all() and float(tot.min()) >= 0


def test_staged_chain_equals_fused_kernel_on_a_volume_slab(lens):
    """The whole staged chain (sample -> chief centre -> trace -> propagate -> forward_integral -> normalise)
    against the fused kernel on the same pupil samples: 512 points of the config-2 volume, 65x65."""
    import bench
    pts = bench.volume_points(1)[::32][:512].contiguous()
    torch.manual_seed(11)
    x2, y2, xc, yc = lens._pupil_samples_pair(1024, lens.entrance_pupil()[1], 2048,
                                              lens.entrance_pupil(shrink_pupil=True)[1], side_stream=False)
    kw = dict(ks=65, dp=DP, pupil_xy=(x2, y2), center_pupil_xy=(xc, yc))
    Lf, Rf = lens.psf_lr(pts, **kw)
    po = lens._points_to_object(pts)
    Ls, Rs = lens._psf_lr_staged(pts, po, pts.shape[0], 65, 0.589, 1024, True, DP, True, True, False, (x2, y2),
                                 (xc, yc), None, None, False)
    d = max(float((Lf - Ls).abs().max()), float((Rf - Rs).abs().max()))
    print(f"staged vs fused, 512 points x 1024 spp, ks 65 (normalised PSFs): max |diff| = {d:.2e}")
    assert d < 2e-6                             # measured 6.6e-7 ... 9.5e-7 (fp32 LDS-atomic order of the fused kernel)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("SDIRT_FUZZ_SEEDS", 6))))
def test_forward_integral_random_batch_shapes_against_the_oracle(oracle, seed):
    """Fuzz of the launch planner's whole range: random point counts (1 ... 700), samples per point (1 ... 3000), grid
    sizes (2 ... 150: double tiles, float tiles, the HBM path), L only / L + R, both microlens branches, both math
    policies; synthetic sensor-plane rays with dead, out-of-window and fractional-weight entries.  Every L and R grid
    of the HIP splat against the CPU oracle's on the same rays."""
    from sdirt_amd import forward_integral_lr
    from sdirt_amd.basics import Ray
    rng = np.random.default_rng(4000 + seed)
    N = int(rng.choice([1, 2, 3, 5, 17, 64, 130, 257, 700]))
    S = int(rng.choice([1, 3, 50, 63, 64, 65, 200, 1023, 1024, 1025, 3000]))
    if N * S > 400000:
        S = max(1, 400000 // N)
    ks = int(rng.choice([2, 5, 9, 21, 33, 65, 99, 100, 141, 150]))
    ps = 0.046875
    big = bool(rng.integers(0, 2))
    h = float(rng.uniform(0.4, 1.1)); f = h + float(rng.uniform(0.3, 1.2))
    dp = None if rng.random() < 0.2 else [h, f, float(rng.uniform(0.1, 0.45)),
                                            float(rng.uniform(0.51, 0.9) if big else rng.uniform(0.15, 0.5))]
    precision = "ieee" if rng.random() < 0.3 else "lean"
    half = ks / 2 * ps
    o = np.zeros((S, N, 3), np.float32)
    o[..., :2] = rng.uniform(-1.15 * half, 1.15 * half, (S, N, 2))          # some rays outside the window
    d = np.zeros((S, N, 3), np.float32)
    d[..., 0] = rng.normal(0, 0.15, (S, N)); d[..., 1] = rng.normal(0, 0.15, (S, N)); d[..., 2] = 1.0
    d = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)
    ra = rng.choice(np.array([0.0, 1.0, 1.0, 1.0, 0.625], np.float32), (S, N))
    cen = rng.uniform(-0.2 * half, 0.2 * half, (N, 2)).astype(np.float32)
    ray = Ray.from_normalized(torch.from_numpy(o), torch.from_numpy(d), ra=torch.from_numpy(ra), device=DEV)
    lg, rg = forward_integral_lr(ray, ps, ks, torch.from_numpy(cen), None if dp is None else dp + ["l"], precision=precision)
    lo, ro = oracle.forward_integral(o, d, ra, ps, ks, cen, dp)
    scale = max(float(lo.max()), float(ro.max()) if dp is not None else 0.0, 1e-6)
    dl = float(np.abs(lg.cpu().numpy() - lo).max()) / scale
    dr = float(np.abs(rg.cpu().numpy() - ro).max()) / scale if dp is not None else float(rg.abs().max())
    print(f"seed {seed}: N={N} S={S} ks={ks} dp={None if dp is None else [round(v, 3) for v in dp]} {precision}: "
          f"|L - oracle| {dl:.2e} |R - oracle| {dr:.2e} of the peak")
    assert dl <= 2e-6 and dr <= 2e-6


@pytest.mark.parametrize("precision", ["lean", "ieee"])
def test_fused_calls_equal_the_two_step_forms_bit_for_bit(lens, precision):
    """sdirt_trace2sensor = sdirt_trace_to + sdirt_propagate_to, and sdirt_forward_integral(SDIRT_PSF_NORMALIZE) =
    sdirt_forward_integral + sdirt_psf_normalize (the same values divided by the same maximum): the fused forms only
    skip a pass over memory."""
    from sdirt_amd import _lib
    from sdirt_amd.basics import Ray, dptr, stream_ptr
    h, st = _lib.lib(), stream_ptr(torch.device(DEV))
    lens.precision = precision
    try:
        pts = _points(96, seed=9)
        po = lens._points_to_object(pts)
        x2, y2 = _pupil(lens, 1500, 13)
        a = _sample(lens, po, x2, y2)
        b = a.clone()
        lens.trace2sensor(a)                                   # one pass
        lens.trace(b, forward=True)
        b.propagate_to(lens.d_sensor)                          # two passes
        assert torch.equal(a.soa.view(torch.int32), b.soa.view(torch.int32))
        flags = lens._math_flags()
        dp = _lib.DpParams(*DP)
        for n, ks in ((96, 33), (3, 21)):                      # a workgroup per point / few points: the spp axis is cut
            cen = torch.zeros((n, 2), device=DEV)
            sub = Ray.empty((1500, n), 0.589, DEV)
            sub.soa.copy_(a.soa.view(7, 96, 1500)[:, :n].reshape(7, -1))
            _lib.check(h.sdirt_center_from_rays(sub.c_rays(), 1500, n, dptr(cen), None, st))
            L1, R1 = torch.empty((n, ks, ks), device=DEV), torch.empty((n, ks, ks), device=DEV)
            L2, R2 = torch.empty_like(L1), torch.empty_like(R1)
            _lib.check(h.sdirt_forward_integral(sub.c_rays(), 1500, n, float(lens.pixel_size), ks, dptr(cen), C.byref(dp),
                                                flags | _lib.PSF_NORMALIZE, dptr(L1), dptr(R1), st))
            plan = (C.c_int64 * 6)()
            assert h.sdirt_forward_integral_plan(n, 1500, ks, 1, 256, plan) == 0
            _lib.check(h.sdirt_forward_integral(sub.c_rays(), 1500, n, float(lens.pixel_size), ks, dptr(cen), C.byref(dp),
                                                flags, dptr(L2), dptr(R2), st))
            _lib.check(h.sdirt_psf_normalize(dptr(L2), n, ks, st))
            _lib.check(h.sdirt_psf_normalize(dptr(R2), n, ks, st))
            # plan[3] == 1: the same float64 sums (their arrival order can move a rounding to fp32 by one ulp, ~1e-7 of the
            # pixels); > 1: partial tiles meet in HBM in arrival order
            tol = 1.2e-7 if plan[3] == 1 else 3e-7
            assert float((L1 - L2).abs().max()) <= tol and float((R1 - R2).abs().max()) <= tol
            assert float(L1.amax()) > 0.999 and float(R1.amax()) > 0.999
    finally:
        lens.precision = "lean"
