"""Size-independent properties of the HIP path at BASELINE.json's full sizes and on
edge-case inputs (no oracle needed: these hold for any correct implementation)."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden, load_state, make_lens

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
DP = (0.78, 1.44, 0.3, 0.5)


@pytest.fixture(scope="module")
def lens():
    return make_lens("rf50mm", DEV)


def full_volume():
    import bench
    return bench.volume_points(1)


def test_full_config2_volume(lens):
    """16384 points x 4096 spp x 65x65 L+R: finite, normalised, mirror-symmetric."""
    pts = full_volume()
    torch.manual_seed(1)
    L, R = lens.psf_lr(pts, ks=65, spp=4096, dp=DP)
    assert L.shape == (16384, 65, 65) and R.shape == L.shape
    assert torch.isfinite(L).all() and torch.isfinite(R).all()
    assert float(L.min()) >= 0 and float(R.min()) >= 0
    mx = L.amax((1, 2))
    assert float(mx.max()) <= 1.0 and float(mx.min()) > 0.99       # psf/(max+1e-6), every point lit
    # the reference derives R from L by mirroring x (psfnet.py:327-330): with the pupil sample
    # set mirrored as well the ray bundle is the exact mirror image, so
    # R(x,y,z) == fliplr(L(-x,y,z)) up to rounding (dual-pixel split is along x only)
    st = load_state("rf50mm")
    g = torch.Generator().manual_seed(2)
    u = torch.rand(4, 4096, generator=g)
    th, r = u[0] * 2 * np.pi, torch.sqrt(u[1] * st["pupil_r"] ** 2)
    x2, y2 = r * torch.cos(th), r * torch.sin(th)
    thc, rc = u[2, :2048] * 2 * np.pi, torch.sqrt(u[3, :2048] * (st["pupil_r"] * 0.25) ** 2)
    xc, yc = rc * torch.cos(thc), rc * torch.sin(thc)
    p = pts[torch.arange(0, 16384, 257)].clone()
    _, R1 = lens.psf_lr(p, ks=65, dp=DP, pupil_xy=(x2, y2), center_pupil_xy=(xc, yc))
    p[:, 0] = -p[:, 0]
    L2, _ = lens.psf_lr(p, ks=65, dp=DP, pupil_xy=(-x2, y2), center_pupil_xy=(-xc, yc))
    diff = (R1 - torch.flip(L2, [-1])).abs()
    # Not exact, and not a bug: the reference takes the top-right tap's column as
    # floor(col + 1) in fp32 (monte_carlo.py:220), which is floor(col) + 2 when col sits one
    # ulp below an integer -- a ray landing 1e-7 mm left of a pixel centre splats differently
    # from its mirror image 1e-7 mm right of it.  One such ray exists in these 64 x 4096.
    assert float((diff > 2e-5).float().mean()) < 1e-5 and float(diff.max()) < 0.03
    assert torch.allclose(R1.sum((1, 2)), L2.sum((1, 2)), rtol=2e-5)


@pytest.mark.parametrize("name,workload,ks,spp,n_z", [("rf35mm", "c4", 65, 4096, 16), ("rf50mm", "c3", 21, 8192, 8)])
def test_other_baseline_configs_full_size(name, workload, ks, spp, n_z):
    """BASELINE config 4 (rf35mm, 21 surfaces, same volume) and one GPU's share of config 3
    (dense grid, 8192 spp, ks 21) at full size: finite, non-negative, every PSF normalised,
    L and R energies consistent with the closed-form sub-pixel areas (s_l + s_r <= 1)."""
    import bench
    ln = make_lens(name, DEV)
    pts = bench.volume_points(1, workload)
    torch.manual_seed(4)
    L, R = ln.psf_lr(pts, ks=ks, spp=spp, dp=DP, normalize=False)
    assert L.shape == (1024 * n_z, ks, ks)
    assert torch.isfinite(L).all() and torch.isfinite(R).all()
    assert float(L.min()) >= 0 and float(R.min()) >= 0
    e = L.sum((1, 2)) + R.sum((1, 2))
    assert float(e.max()) <= spp * (1 + 1e-5)            # energy never exceeds the ray count
    assert float((L.sum((1, 2)) > 0).float().mean()) == 1.0
    Ln, Rn = ln.psf_lr(pts[::97], ks=ks, spp=spp, dp=DP)
    assert float(Ln.amax((1, 2)).min()) > 0.99 and float(Rn.amax((1, 2)).max()) <= 1.0


def test_single_point_and_default_param_list(lens):
    p = torch.tensor([0.3, -0.2, -1500.0])
    torch.manual_seed(5)
    a = lens.psf(p, ks=21, spp=512)                      # [3] -> [ks,ks]  (optics.py:949-953,993)
    torch.manual_seed(5)
    b = lens.psf(p.unsqueeze(0), ks=21, spp=512)
    # bit-equal in the reference; here equal up to the order of the LDS float atomics
    assert a.shape == (21, 21) and torch.allclose(a, b[0], atol=2e-6)
    torch.manual_seed(5)
    c = lens.psf_diff(p, ks=21, spp=512, param_list=[0.78, 1.44, 0.3, 0.5, "l"])
    assert torch.allclose(a, c, atol=2e-6)               # defaults == explicit list (monte_carlo.py:157)
    torch.manual_seed(5)
    d = lens.psf_diff(p, ks=21, spp=512, param_list=[0.78, 1.44, 0.3, 0.5, "r"])
    assert d.shape == (21, 21) and not torch.allclose(a, d, atol=1e-3)


def test_spp_split_path_equals_single_workgroup_path(lens):
    """Few points + many samples take the nsplit>1 route (global float atomics +
    separate normalise); it must agree with the one-workgroup-per-point route."""
    st = load_state("rf50mm")
    pts = torch.tensor([[0.0, 0.0, -800.0], [0.6, -0.5, -3000.0], [-0.9, 0.9, -250.0]])
    g = torch.Generator().manual_seed(11)
    S = 20000                                            # PSFNet training spp (1_fit_psfnet.py:36)
    u = torch.rand(2, S, generator=g)
    th, r = u[0] * 2 * np.pi, torch.sqrt(u[1] * st["pupil_r"] ** 2)
    xy = (r * torch.cos(th), r * torch.sin(th))
    uc = torch.rand(2, 2048, generator=g)
    thc, rc = uc[0] * 2 * np.pi, torch.sqrt(uc[1] * (st["pupil_r"] * 0.25) ** 2)
    xyc = (rc * torch.cos(thc), rc * torch.sin(thc))
    L3, R3 = lens.psf_lr(pts, ks=21, dp=DP, pupil_xy=xy, center_pupil_xy=xyc)   # N=3 -> split
    big = pts.repeat(400, 1)                                                   # N=1200 -> no split
    Lb, Rb = lens.psf_lr(big, ks=21, dp=DP, pupil_xy=xy, center_pupil_xy=xyc)
    assert torch.allclose(L3, Lb[:3], atol=3e-6) and torch.allclose(R3, Rb[:3], atol=3e-6)
    # identical points in different workgroups: LDS float atomics -> equal up to summation order
    assert torch.allclose(Lb[:3], Lb[3:6], atol=2e-6)

def test_trip_policy_max_is_close_to_reference_policy(lens):
    pts = torch.tensor([[0.2, 0.1, -1000.0], [-0.8, 0.7, -10000.0]])
    torch.manual_seed(3)
    a, _ = lens.psf_lr(pts, ks=33, spp=4096, dp=DP)
    lens.trip_policy = "max"
    try:
        torch.manual_seed(3)
        b, _ = lens.psf_lr(pts, ks=33, spp=4096, dp=DP)
    finally:
        lens.trip_policy = "reference"
    # extra Newton trips after convergence move t by rounding noise only (SURVEY.md §7 hard part 2)
    assert float((a - b).abs().max()) < 1e-4


def test_edge_cases(lens):
    # empty batch
    L, R = lens.psf_lr(torch.zeros(0, 3), ks=9, spp=64, dp=DP)
    assert L.shape == (0, 9, 9)
    # extreme field point outside the image circle: everything vignetted or outside the window
    far = torch.tensor([[5.0, 5.0, -1000.0]])
    lens.trip_policy = "max"                             # no 'No sampled rays is valid.' assertion
    try:
        L, R = lens.psf_lr(far, ks=9, spp=256, dp=DP)
    finally:
        lens.trip_policy = "reference"
    assert torch.isfinite(L).all() and float(L.max()) <= 1.0
    with pytest.raises(AssertionError, match="No sampled rays is valid"):
        lens.psf(far, ks=9, spp=64)                      # optics.py:902
    # smallest / largest supported kernel sizes, even ks
    for ks in (2, 8, 141):
        L, _ = lens.psf_lr(torch.tensor([[0.0, 0.0, -1000.0]]), ks=ks, spp=128, dp=DP)
        assert L.shape == (1, ks, ks) and torch.isfinite(L).all()
    # one pixel more no longer fits a workgroup's LDS: the staged chain takes over (tests/test_gpu_callers.py)
    from sdirt_amd import SdirtError, _lib
    big = lens.psf(torch.tensor([[0.0, 0.0, -1000.0]]), ks=142, spp=64)
    assert big.shape == (1, 142, 142) and float(big.max()) > 0.99
    with pytest.raises(SdirtError):
        lens.psf(torch.tensor([[0.0, 0.0, -1000.0]]), ks=_lib.MAX_KS_STAGED + 1, spp=64)
    # center=False uses the pinhole centre (optics.py:971-976)
    L = lens.psf(torch.tensor([[0.0, 0.0, -1000.0]]), ks=21, spp=256, center=False)
    assert L.shape == (1, 21, 21) and float(L.max()) > 0.99


def test_single_launch_route_small_tiles_left_only(lens):
    """N >= 1024 -> one workgroup per point does the chief-ray pass and the primary pass; with
    the default param_list=None only the L tile exists (smaller than the fp64 reduction
    scratch that aliases it).  Must equal the L of the two-sided call."""
    g = torch.Generator().manual_seed(31)
    pts = torch.stack([(torch.rand(1100, generator=g) - 0.5) * 1.8,
                       (torch.rand(1100, generator=g) - 0.5) * 1.8,
                       -(300 + torch.rand(1100, generator=g) * 9000)], -1)
    torch.manual_seed(12)
    a = lens.psf(pts, ks=9, spp=512)
    torch.manual_seed(12)
    b, _ = lens.psf_lr(pts, ks=9, spp=512, dp=DP)
    assert a.shape == (1100, 9, 9) and torch.allclose(a, b, atol=3e-6)
    torch.manual_seed(12)
    c = lens.psf(pts[:7], ks=9, spp=512)                 # split-spp route, same points
    assert torch.allclose(a[:7], c, atol=1e-4)           # other batch -> other global trip table


def test_caller_owned_output_buffers(lens):
    pts = torch.tensor([[0.1, 0.2, -900.0], [-0.4, 0.3, -6000.0]])
    L = torch.full((2, 17, 17), -1.0, device=DEV)
    R = torch.full((2, 17, 17), -1.0, device=DEV)
    torch.manual_seed(9)
    L2, R2 = lens.psf_lr(pts, ks=17, spp=512, dp=DP, out=(L, R))
    assert L2.data_ptr() == L.data_ptr() and R2.data_ptr() == R.data_ptr()
    torch.manual_seed(9)
    L3, R3 = lens.psf_lr(pts, ks=17, spp=512, dp=DP)
    assert torch.allclose(L, L3, atol=2e-6) and torch.allclose(R, R3, atol=2e-6)
    with pytest.raises(ValueError):
        lens.psf_lr(pts, ks=17, spp=64, dp=DP, out=(L[:, :9], R))


def test_rgb_and_map_shapes(lens):
    pts = torch.tensor([[0.0, 0.0, -1000.0], [0.5, 0.5, -2000.0]])
    rgb = lens.psf_rgb(pts, ks=11, spp=128)
    assert rgb.shape == (2, 3, 11, 11)
    assert lens.psf_rgb(pts[0], ks=11, spp=128).shape == (3, 11, 11)
    m = lens.psf_map(depth=-1000.0, grid=3, ks=11, spp=128)
    assert m.shape == (3, 33, 33) and torch.isfinite(m).all()


def test_staged_pipeline_equals_fused(lens):
    """sample_from_points -> trace2sensor -> forward_integral (the reference's own
    decomposition, rays in HBM as SoA) against the fused kernel on the same samples."""
    from sdirt_amd import forward_integral_lr
    st = load_state("rf50mm")
    pts = torch.tensor([[0.1, 0.3, -700.0], [-0.6, -0.4, -4000.0]])
    po = lens._points_to_object(pts)
    torch.manual_seed(21)
    ray = lens.sample_from_points(po, spp=2048)
    assert ray.o.shape == (2048, 2, 3) and ray.ra.shape == (2048, 2)
    ray = lens.trace2sensor(ray)
    cen = torch.tensor([[0.01, -0.02], [0.0, 0.0]], device=DEV)
    lg, rg = forward_integral_lr(ray, lens.pixel_size, 33, cen, list(DP) + ["l"])
    torch.manual_seed(21)
    u = (torch.rand(2048), torch.rand(2048))
    # fused kernel with explicit centre: reproduce through the C ABI
    import ctypes as C
    from sdirt_amd import _lib
    from sdirt_amd.basics import dptr, stream_ptr
    xy = torch.empty((2, 2048), device=DEV)
    ud = torch.stack(u).to(DEV)
    sp = stream_ptr(torch.device(DEV))
    _lib.check(_lib.lib().sdirt_pupil_samples(dptr(ud[0]), dptr(ud[1]), 2048, st["pupil_r"],
                                              dptr(xy[0]), dptr(xy[1]), sp))
    K = len(lens.surfaces)
    trips = (C.c_int32 * K)(*[int(v) for v in lens.trips.cache[("trace", 0.589, 0, K, True, "lean")]])
    L = torch.empty((2, 33, 33), device=DEV); R = torch.empty_like(L)
    dp = _lib.DpParams(*DP)
    _lib.check(_lib.lib().sdirt_psf_lr(lens.dev_lens(0.589), dptr(po), 2, dptr(xy[0]), dptr(xy[1]),
                                       2048, st["pupil_z"], st["d_sensor"], st["pixel_size"], 33,
                                       dptr(cen), C.byref(dp), trips, 0, dptr(L), dptr(R), None, sp))
    assert torch.allclose(L, lg, atol=3e-6 * float(lg.max())) and torch.allclose(R, rg, atol=3e-6 * float(rg.max()))


def test_lean_and_strict_ieee_modes_agree():
    """The default lean division / sqrt are proven equal to IEEE on normal-range operands
    (test_lean_math_is_exact below); on real rays the two kernel instantiations must give the
    same centres bit for bit and the same PSFs up to LDS-atomic summation order."""
    st, g = load_state("rf50mm"), load_golden("f8_rf50_mini_c2")
    lens = make_lens("rf50mm", DEV, st)
    kw = dict(ks=65, dp=DP, pupil_xy=(g["pupil_x2"], g["pupil_y2"]),
              center_pupil_xy=(g["pupil_xc"], g["pupil_yc"]))
    pts = torch.tensor(g["points"])
    assert lens.precision == "lean"
    Ll, Rl = lens.psf_lr(pts, **kw)
    cl = lens.psf_center(lens._points_to_object(pts))      # draws fresh samples; compare below
    lens.precision = "ieee"
    Li, Ri = lens.psf_lr(pts, **kw)
    assert np.array_equal(lens.trips.cache[("psf", 0.589, "ieee")], g["trips"])
    assert np.array_equal(lens.trips.cache[("psf", 0.589, "lean")], g["trips"])
    assert float((Ll - Li).abs().max()) < 1e-6 and float((Rl - Ri).abs().max()) < 1e-6
    # centres: deterministic fp64 reduction -> bit-equal between the two instantiations
    po = lens._points_to_object(pts)
    xc, yc = torch.tensor(g["pupil_xc"], device=DEV), torch.tensor(g["pupil_yc"], device=DEV)
    ci = torch.empty((len(pts), 2), device=DEV)
    lens._chief_center(po, xc, yc, st["pupil_z"], ci)
    lens.precision = "lean"
    cl = torch.empty_like(ci)
    lens._chief_center(po, xc, yc, st["pupil_z"], cl)
    assert torch.equal(ci, cl)
    with pytest.raises(ValueError):
        lens.precision = "sloppy"
        lens.psf_lr(pts[:1], ks=9, spp=64)


def test_lean_math_is_exact():
    """sdirt_selftest_math: the lean sqrt on EVERY fp32 >= 2^-100 and the lean division on
    2^36 random operand pairs plus a 2^36 slice of the exhaustive mantissa-pair enumeration
    (the full 2^46 run takes 46 s: tools/selftest_math.py, profiles/r01/selftest_math.txt)."""
    from sdirt_amd import _lib
    from sdirt_amd.basics import dptr, stream_ptr
    out = torch.zeros(9, dtype=torch.int64, device=DEV)

    def run(mode, first, count, span=0):
        _lib.check(_lib.lib().sdirt_selftest_math(mode, first, count, span, dptr(out),
                                                  stream_ptr(torch.device(DEV))))
        return int(out.cpu()[0])
    lo = (127 - 100) << 23
    assert run(0, lo, 0x7F800000 - lo) == 0                     # all normals >= 2^-100
    assert run(0, 0, 1) == 0 and run(0, 0x7F800000, 1) == 0     # +0, +inf
    assert run(0, 0x80000000, 1) == 0 and run(0, 0xBF800000, 1) == 0   # -0, -1 -> NaN
    assert run(1, 0, 1 << 36, 40) == 0
    assert run(2, 0x123456789AB, 1 << 36) == 0
    # sqrt_pos (v_rsq + Markstein correction; positive normal arguments only): EVERY fp32 in
    # [2^-100, 2^100] -- mode 3 skips bit patterns outside that range by itself
    assert run(3, 0, 1 << 32) == 0


def test_deferred_trip_check_equals_the_immediate_one():
    """psf_lr(defer=True): same PSFs, same verified trip tables, also when the speculated table
    is wrong for the batch (forced here by poisoning the planner's memory)."""
    lens = make_lens("rf50mm", DEV)
    g = load_golden("f8_rf50_mini_c2")
    pts = torch.tensor(g["points"])
    kw = dict(ks=33, spp=512, pupil_xy=(g["pupil_x2"][:512], g["pupil_y2"][:512]),
              center_pupil_xy=(g["pupil_xc"], g["pupil_yc"]))
    L0, R0 = lens.psf_lr(pts, **kw)
    tabs = {k: v.copy() for k, v in lens.trips.cache.items()}
    p1 = lens.psf_lr(pts, defer=True, **kw)
    p2 = lens.psf_lr(pts[:5], defer=True, **kw)             # two calls in flight
    L1, R1 = p1.wait()
    assert p1.wait()[0] is L1                                # idempotent
    L2, _ = p2.wait()
    assert torch.allclose(L0, L1, atol=2e-6) and torch.allclose(R0, R1, atol=2e-6)
    assert torch.allclose(L0[:5], L2, atol=2e-6)
    # wrong speculation: one trip short on a curved surface of both passes
    for k in list(lens.trips.cache):
        bad = lens.trips.cache[k].copy()
        bad[1] = max(1, bad[1] - 1)
        lens.trips.cache[k] = bad
        lens.trips.votes[k] = {tuple(int(x) for x in bad): 99}
    before = lens.trips.relaunches
    L3, R3 = lens.psf_lr(pts, defer=True, **kw).wait()
    assert lens.trips.relaunches > before
    assert torch.allclose(L0, L3, atol=2e-6) and torch.allclose(R0, R3, atol=2e-6)
    for k, v in tabs.items():
        assert np.array_equal(lens.trips.cache[k], v), k


def test_adaptive_speed_mode_stays_within_its_stated_distance():
    """trip_policy='adaptive' (per-wave Newton exit, no host check): PSFs within 3e-4 of peak of
    the batch-exact ones (SURVEY.md hard part 2's acceptance for adaptive Newton; measured ~1e-5),
    chief-ray centres within 1e-5 mm, same validity."""
    lens = make_lens("rf50mm", DEV)
    g = load_golden("f8_rf50_mini_c2")
    pts = torch.tensor(g["points"])
    kw = dict(ks=65, spp=4096, pupil_xy=(g["pupil_x2"], g["pupil_y2"]),
              center_pupil_xy=(g["pupil_xc"], g["pupil_yc"]))
    L0, R0 = lens.psf_lr(pts, **kw)
    lens.trip_policy = "adaptive"
    launches = lens.trips.launches
    L1, R1 = lens.psf_lr(pts, **kw)
    assert lens.trips.launches == launches                   # no speculation / verification rounds
    dl, dr = (L0 - L1).abs().max().item(), (R0 - R1).abs().max().item()
    assert dl < 3e-4 and dr < 3e-4, (dl, dr)
    assert (L1.amax((-1, -2)) - 1).abs().max().item() < 1e-5
    lens.trip_policy = "bogus"
    with pytest.raises(ValueError):
        lens.psf_lr(pts[:1], **kw)


def _random_prescription(rng, n_elements):
    """A random but traceable lens: glass elements (two curved faces each; spheres, conics with
    k in [-2.5, 1], even aspheres), a stop, optionally a flat window.  Own JSON schema + the
    per-surface state dict the oracle helpers read."""
    from sdirt_amd.basics import Material
    surfaces, z = [], 0.0
    glasses = ["1.51680/64.2", "1.80518/25.4", "1.67270/32.1", "1.53110/55.9"]

    def add(kind, semi, c, glass_a, glass_b, k=0.0, ai=None):
        nonlocal z
        s = dict(kind=kind, semi_aperture=semi, z=z, curvature=c, glass_before=glass_a, glass_after=glass_b)
        if kind == "asphere":
            s["conic"], s["even_asphere"] = k, ai or [0.0] * 6
        surfaces.append(s)

    for e in range(n_elements):
        g = glasses[rng.integers(len(glasses))]
        for face in range(2):
            c = float(rng.uniform(0.01, 0.06) * rng.choice([-1, 1]))
            roll = rng.random()
            if roll < 0.4:
                add("sphere", 9.0, c, "air" if face == 0 else g, g if face == 0 else "air")
            else:
                k = float(rng.uniform(-2.5, 1.0)) if roll < 0.8 else 0.0
                ai = [float(rng.normal(0, s)) for s in (2e-4, 2e-5, 2e-7, 1e-9, 1e-11, 1e-13)]
                add("asphere", 9.0, c, "air" if face == 0 else g, g if face == 0 else "air", k, ai)
            z += float(rng.uniform(1.5, 4.0))
        if e == 0:
            add("plane", 5.0, 0.0, "air", "air")                       # the stop
            z += float(rng.uniform(1.0, 3.0))
    if rng.random() < 0.5:                                              # flat glass window
        add("plane", 9.0, 0.0, "air", glasses[0]); z += 1.0
        add("plane", 9.0, 0.0, glasses[0], "air"); z += 1.0
    data = dict(name="fuzz", units="mm", r_last=21.64, d_sensor=z + 30.0, sensor_size=[24.0, 36.0],
                surfaces=surfaces)
    key = repr(0.589)
    state = dict(surfaces=[dict(
        kind=s["kind"], r=s["semi_aperture"], d=s["z"], c=s["curvature"], k=s.get("conic", 0.0),
        ai=s.get("even_asphere", []) if s["kind"] == "asphere" else [],
        n1={key: float(Material(s["glass_before"]).ior(0.589))},
        n2={key: float(Material(s["glass_after"]).ior(0.589))}) for s in surfaces])
    return data, state


@pytest.mark.parametrize("seed", range(int(os.environ.get("SDIRT_FUZZ_SEEDS", 6))))
def test_random_prescriptions_trace_bit_exact_against_the_oracle(oracle, seed, tmp_path):
    """Fuzz: random lenses (conics on both sides of k = -1, even aspheres of degree 6, either sign
    of curvature, flat refracting windows, a stop) and a random ray bundle, forward and backward:
    the HIP trace (both math policies) equals the CPU oracle bit for bit, trip tables included."""
    import json
    from sdirt_amd import Lensgroup
    rng = np.random.default_rng(seed)
    data, state = _random_prescription(rng, n_elements=int(rng.integers(1, 4)))
    path = tmp_path / "fuzz.json"
    path.write_text(json.dumps(data))
    K = len(data["surfaces"])
    surf = oracle.surfaces_from_state(state, 0.589)
    n = 4096
    for backward in (False, True):
        o = np.zeros((n, 3), np.float32)
        o[:, :2] = rng.uniform(-6, 6, (n, 2))
        o[:, 2] = -50.0 if not backward else data["d_sensor"]
        d = np.zeros((n, 3), np.float32)
        d[:, :2] = rng.normal(0, 0.08, (n, 2))
        d[:, 2] = -1.0 if backward else 1.0
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        d = d.astype(np.float32)
        ref = oracle.trace(surf, o, d, np.ones(n, np.float32))
        assert 0.02 < ref["ra"].mean() <= 1.0, "degenerate prescription: tune the generator"
        for precision in ("lean", "ieee"):
            lens = Lensgroup(str(path), sensor_res=(512, 768), post_computation=False, device=DEV)
            lens.precision = precision
            from test_gpu_parity import rays_from_fixture
            ray = rays_from_fixture(o, d)
            ray, valid, _ = lens.trace(ray)
            key = ("trace", 0.589, 0, K, not backward, precision)
            assert np.array_equal(lens.trips.cache[key], ref["trips"]), (seed, backward, precision)
            assert np.array_equal(ray.ra.cpu().numpy(), ref["ra"])
            assert np.array_equal(ray.o.cpu().numpy(), ref["o"]), (seed, backward, precision)
            assert np.array_equal(ray.d.cpu().numpy(), ref["d"])
            assert np.array_equal(ray.obliq.cpu().numpy(), ref["obliq"])


@pytest.mark.parametrize("seed", range(int(os.environ.get("SDIRT_FUZZ_SEEDS", 4))))
def test_long_trip_tables_periodic_exit_bit_exact_against_the_oracle(oracle, seed, tmp_path):
    """Distant object points: the reference's Newton loop runs to its 10-trip cap on the first
    surface (t ~ 1e3..2e4 mm cannot resolve |f| < 50e-6), which is where the kernels leave the
    loop early once every ray of a wave is periodic.  The oracle runs every trip: positions,
    directions, validity and the convergence masks (trip tables) must still be equal bit for bit."""
    import json
    from sdirt_amd import Lensgroup
    from test_gpu_parity import rays_from_fixture
    rng = np.random.default_rng(100 + seed)
    data, state = _random_prescription(rng, n_elements=int(rng.integers(1, 4)))
    path = tmp_path / "fuzz_far.json"
    path.write_text(json.dumps(data))
    K = len(data["surfaces"])
    surf = oracle.surfaces_from_state(state, 0.589)
    n = 8192
    depth = -rng.uniform(800.0, 20000.0, n)
    o = np.stack([rng.uniform(-0.15, 0.15, n) * -depth, rng.uniform(-0.15, 0.15, n) * -depth, depth], 1).astype(np.float32)
    tgt = np.stack([rng.uniform(-4, 4, n), rng.uniform(-4, 4, n), np.zeros(n)], 1)
    d = tgt - o.astype(np.float64)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    ref = oracle.trace(surf, o, d, np.ones(n, np.float32))
    assert max(ref["trips"]) > 4, "no long table: the periodic-exit loop would not run"
    for precision in ("lean", "ieee"):
        lens = Lensgroup(str(path), sensor_res=(512, 768), post_computation=False, device=DEV)
        lens.precision = precision
        ray, valid, _ = lens.trace(rays_from_fixture(o, d))
        assert np.array_equal(lens.trips.cache[("trace", 0.589, 0, K, True, precision)], ref["trips"])
        assert np.array_equal(ray.ra.cpu().numpy(), ref["ra"])
        assert np.array_equal(ray.o.cpu().numpy(), ref["o"]), (seed, precision)
        assert np.array_equal(ray.d.cpu().numpy(), ref["d"])


@pytest.mark.parametrize("seed", range(int(os.environ.get("SDIRT_FUZZ_SEEDS", 8))))
def test_random_dual_pixel_parameters_splat_against_the_oracle(oracle, seed):
    """Fuzz of the DP model (monte_carlo.py:135-372): random microlens / stack geometry (h, f, w, r)
    on both sides of r = 0.5, random sensor-plane rays incl. dead and out-of-window ones: L and R
    grids of the HIP splat against the CPU oracle."""
    from sdirt_amd import forward_integral_lr
    rng = np.random.default_rng(100 + seed)
    S, N, ks, ps = 512, 5, int(rng.choice([9, 21, 33])), 0.046875
    h = float(rng.uniform(0.4, 1.1)); f = h + float(rng.uniform(0.3, 1.2))
    w = float(rng.uniform(0.1, 0.45)); r = float(rng.uniform(0.15, 0.95))
    half = (ks / 2 - 0.5) * ps
    o = np.zeros((S, N, 3), np.float32)
    o[..., :2] = rng.uniform(-1.25 * half, 1.25 * half, (S, N, 2))
    o[..., 2] = 62.25
    d = rng.normal(0, 0.2, (S, N, 3)).astype(np.float32)
    d[..., 2] = 1.0
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    ra = (rng.random((S, N)) > 0.15).astype(np.float32)
    cen = ((rng.random((N, 2)) - 0.5) * ps).astype(np.float32)
    lg0, rg0 = oracle.forward_integral(o, d.astype(np.float32), ra, ps, ks, cen, dp=[h, f, w, r])
    from test_gpu_parity import rays_from_fixture, t
    ray = rays_from_fixture(o, d.astype(np.float32))
    ray.ra = t(ra)
    lg, rg = forward_integral_lr(ray, ps, ks, t(cen), [h, f, w, r, "l"])
    scale = max(lg0.max(), rg0.max())
    assert scale > 0
    assert np.abs(lg.cpu().numpy() - lg0).max() <= 3e-6 * scale, (h, f, w, r)
    assert np.abs(rg.cpu().numpy() - rg0).max() <= 3e-6 * scale, (h, f, w, r)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_lens_on_another_gpu_than_the_current_device():
    """Every launch runs on the GPU its stream belongs to (_lib.StreamArg), whatever the caller's
    current device is: a lens on cuda:1 driven while cuda:0 is current gives cuda:0's results."""
    torch.cuda.set_device(0)
    g = load_golden("f8_rf50_mini_c2")
    kw = dict(ks=33, spp=512, pupil_xy=(g["pupil_x2"][:512], g["pupil_y2"][:512]),
              center_pupil_xy=(g["pupil_xc"], g["pupil_yc"]))
    L0, R0 = make_lens("rf50mm", "cuda:0").psf_lr(torch.tensor(g["points"]), **kw)
    lens1 = make_lens("rf50mm", "cuda:1")
    L1, R1 = lens1.psf_lr(torch.tensor(g["points"]), **kw)
    assert torch.cuda.current_device() == 0 and L1.device.index == 1
    assert torch.allclose(L0.cpu(), L1.cpu(), atol=2e-6) and torch.allclose(R0.cpu(), R1.cpu(), atol=2e-6)


def test_object_point_cache_follows_in_place_writes_and_new_tensors(lens):
    """The conversion of the normalised points is kept for the last tensor seen: an in-place
    write, a different tensor at the same address and a changed lens scalar must all miss."""
    pts = torch.tensor([[0.1, 0.2, -1000.0], [-0.4, 0.3, -5000.0]], device=DEV)
    a = lens._points_to_object(pts)
    assert lens._points_to_object(pts) is a                               # hit
    assert torch.equal(a, lens._points_to_object_now(pts))
    pts[0, 0] = 0.5                                                       # in-place write
    b = lens._points_to_object(pts)
    assert b is not a and torch.equal(b, lens._points_to_object_now(pts)) and not torch.equal(a, b)
    addr = pts.data_ptr()
    del pts
    for _ in range(4):                                                    # same block, new tensor
        q = torch.tensor([[0.7, -0.2, -300.0], [0.0, 0.0, -20000.0]], device=DEV)
        if q.data_ptr() == addr:
            break
    assert torch.equal(lens._points_to_object(q), lens._points_to_object_now(q))
    r_last = lens.r_last
    try:
        lens.r_last = r_last * 1.5                                        # a lens scalar of the mapping
        assert torch.equal(lens._points_to_object(q), lens._points_to_object_now(q))
    finally:
        lens.r_last = r_last
    L0, _ = lens.psf_lr(q, ks=21, spp=256)
    q[:, :2] *= 0.5
    L1, _ = lens.psf_lr(q, ks=21, spp=256)
    assert not torch.allclose(L0, L1)


def test_control_blocks_from_the_pool_are_zero_and_distinct(lens):
    blocks = [lens._zeroed_control_block(65) for _ in range(130)]        # crosses two pool refills
    assert all(int(b.abs().sum()) == 0 for b in blocks)
    assert len({b.data_ptr() for b in blocks}) == len(blocks)
    blocks[0].fill_(7)
    assert int(blocks[1].abs().sum()) == 0


def test_control_block_pools_are_per_stream(lens):
    """Rows handed out under stream B never come from a pool that was zero-filled on stream A
    (the fill could run after a kernel on B had OR-ed its masks into the row)."""
    a = lens._zeroed_control_block(65)
    side = torch.cuda.Stream(DEV)
    with torch.cuda.stream(side):
        b = lens._zeroed_control_block(65)
        c = lens._zeroed_control_block(65)
    d = lens._zeroed_control_block(65)
    base = lambda t: t.untyped_storage().data_ptr()
    assert base(a) == base(d) and base(b) == base(c) and base(a) != base(b)
    # two streams rendering on one lens (train_psfnet's producer / evaluator pattern)
    pts = torch.tensor([[0.1, 0.2, -1000.0], [-0.4, 0.3, -5000.0]], device=DEV)
    torch.manual_seed(3)
    L0, R0 = lens.psf_lr(pts, ks=21, spp=512)
    with torch.cuda.stream(side):
        torch.manual_seed(3)
        L1, R1 = lens.psf_lr(pts, ks=21, spp=512)
    torch.cuda.synchronize()
    assert torch.allclose(L0, L1, atol=2e-6) and torch.allclose(R0, R1, atol=2e-6)


def test_inference_mode_and_cpu_points_bypass_the_object_point_cache(lens):
    with torch.inference_mode():
        pts = torch.tensor([[0.1, 0.2, -1000.0]], device=DEV)
        torch.manual_seed(9)
        L0, _ = lens.psf_lr(pts, ks=21, spp=256)
    arr = np.array([[0.1, 0.2, -1000.0]], dtype=np.float32)
    cpu = torch.from_numpy(arr)
    torch.manual_seed(9)
    L1, _ = lens.psf_lr(cpu, ks=21, spp=256)
    assert torch.allclose(L0, L1, atol=2e-6)
    arr[0, 0] = -0.6                                   # numpy-side write: no version bump
    torch.manual_seed(9)
    L2, _ = lens.psf_lr(cpu, ks=21, spp=256)
    assert not torch.allclose(L1, L2, atol=1e-3)


def test_prefetch_selftest_runs_at_first_upload_and_catches_a_disagreement(monkeypatch):
    """The load-time guard of the hand-scheduled scalar prefetch: passes on the shipped library,
    and raises when the two trace forms are made to disagree."""
    from sdirt_amd import _lib, Lensgroup
    Lensgroup._selftest_done.clear()
    ln = make_lens("rf50mm", DEV)
    ln.dev_lens(0.589)                                  # runs the self-test
    assert Lensgroup._selftest_done
    Lensgroup._selftest_done.clear()
    real = _lib.lib().sdirt_trace

    def skewed(handle, first, last, backward, trips, flags, rays, n, mask, stream):
        if flags & _lib.TRACE_NO_PREFETCH:               # trace one surface less: a stand-in for stale constants
            last = last - 1 if not backward else last
            first = first + 1 if backward else first
        return real(handle, first, last, backward, trips, flags, rays, n, mask, stream)
    monkeypatch.setattr(_lib.lib(), "sdirt_trace", skewed)
    with pytest.raises(_lib.SdirtError, match="self-test failed"):
        make_lens("rf50mm", DEV).dev_lens(0.589)
    monkeypatch.undo()
    # the verdict on this device and prescription stands: a later call -- on any lens object -- raises too
    with pytest.raises(_lib.SdirtError, match="self-test failed"):
        make_lens("rf50mm", DEV).dev_lens(0.589)
    Lensgroup._selftest_failed.clear()                 # ... until the process ends; the tests that follow share this one
    Lensgroup._selftest_done.clear()
    make_lens("rf50mm", DEV).dev_lens(0.589)


def _training_shape_inputs(n=64, spp=20000, seed=5):
    st = load_state("rf50mm")
    g = torch.Generator().manual_seed(seed)
    pts = torch.stack([(torch.rand(n, generator=g) - 0.5) * 2, (torch.rand(n, generator=g) - 0.5) * 2,
                       -(200 + torch.rand(n, generator=g) * 19800)], -1)
    u = torch.rand(2, spp, generator=g)
    th, r = u[0] * 2 * np.pi, torch.sqrt(u[1] * st["pupil_r"] ** 2)
    uc = torch.rand(2, 2048, generator=g)
    thc, rc = uc[0] * 2 * np.pi, torch.sqrt(uc[1] * (st["pupil_r"] * 0.25) ** 2)
    return pts, (r * torch.cos(th), r * torch.sin(th)), (rc * torch.cos(thc), rc * torch.sin(thc))


@pytest.mark.parametrize("bet", ["learned", "from_above", "from_below"])
def test_device_side_trip_verification_on_the_split_path(bet):
    """PSFNet's fitting shape (64 points x 20000 spp, ks 21): several workgroups per point,
    sdirt_psf_lr_verified.  Whatever the speculated tables are, the call ends on the reference's
    tables -- checked and, if need be, corrected and re-rendered ON THE DEVICE, without a host
    re-launch when the bet was an upper bound -- and on the PSFs of the host-verified path."""
    pts, xy, xyc = _training_shape_inputs()
    ref_lens = make_lens("rf50mm", DEV)
    ref_lens.mask_reduce = lambda m: m                    # forces the host-verified (round-2) path
    L0, R0 = ref_lens.psf_lr(pts, ks=21, dp=DP, pupil_xy=xy, center_pupil_xy=xyc)
    want = {k: np.asarray(v) for k, v in ref_lens.trips.cache.items()}
    c0 = torch.empty((64, 2), device=DEV)
    ref_lens.psf_lr(pts, ks=21, dp=DP, pupil_xy=xy, center_pupil_xy=xyc, center_out=c0)

    ln = make_lens("rf50mm", DEV)
    curved = np.array(ln._curved())
    if bet != "learned":
        for key, t in want.items():
            wrong = t.copy()
            k = int(np.flatnonzero(curved & (t < 10))[2])          # a surface in the middle of the stack
            wrong[k] += 1 if bet == "from_above" else -1
            ln.trips.learn(key, wrong)
    else:
        ln.psf_lr(pts, ks=21, dp=DP, pupil_xy=xy, center_pupil_xy=xyc)   # discovers the tables
    r0, d0 = ln.trips.relaunches, ln.trips.device_relaunches
    c1 = torch.empty((64, 2), device=DEV)
    L1, R1 = ln.psf_lr(pts, ks=21, dp=DP, pupil_xy=xy, center_pupil_xy=xyc, center_out=c1)
    for key, t in want.items():
        assert np.array_equal(ln.trips.cache[key], t), (key, ln.trips.cache[key], t)
    if bet == "learned":
        assert (ln.trips.relaunches - r0, ln.trips.device_relaunches - d0) == (0, 0)
    elif bet == "from_above":
        assert (ln.trips.relaunches - r0, ln.trips.device_relaunches - d0) == (0, 1)
    else:
        assert ln.trips.device_relaunches - d0 == 1          # one trip more is the device's first guess
    assert torch.equal(c0, c1) or float((c0 - c1).abs().max()) < 1e-6
    assert float((L0 - L1).abs().max()) <= 3e-6 and float((R0 - R1).abs().max()) <= 3e-6
    # deferred form: same call, verification in .wait()
    pend = ln.psf_lr(pts, ks=21, dp=DP, pupil_xy=xy, center_pupil_xy=xyc, defer=True)
    L2, R2 = pend.wait()
    assert float((L0 - L2).abs().max()) <= 3e-6 and float((R0 - R2).abs().max()) <= 3e-6
    # default param_list (R grid stays zero, monte_carlo.py:231) through the same path
    a = ln.psf_diff(pts, ks=21, spp=20000)
    assert a.shape == (64, 21, 21) and float(a.amax((1, 2)).min()) > 0.99


def test_one_call_synchronous_path_equals_the_general_path():
    """sdirt_psf_call (draw -> one library call -> wait) against the general verified path handed the
    same pupil points, and against a deferred call from the same seed: same PSFs, same centres, same
    verified tables, the generator left where the reference leaves it (2 x 20000 + 2 x 2048 draws)."""
    st = load_state("rf50mm")
    pts, _, _ = _training_shape_inputs(seed=9)
    ln = make_lens("rf50mm", DEV)
    torch.manual_seed(42)
    c1 = torch.empty((64, 2), device=DEV)
    L1, R1 = ln.psf_lr(pts, ks=21, spp=20000, dp=DP, center_out=c1)          # one-call path
    tail = torch.rand(4)
    x2, y2, xc, yc = [v.clone() for v in ln.last_pupil_points]
    torch.manual_seed(42)
    torch.rand(2 * 20000 + 2 * 2048)
    assert torch.equal(torch.rand(4), tail)
    ln2 = make_lens("rf50mm", DEV)
    c2 = torch.empty((64, 2), device=DEV)
    L2, R2 = ln2.psf_lr(pts, ks=21, dp=DP, pupil_xy=(x2, y2), center_pupil_xy=(xc, yc), center_out=c2)
    assert torch.equal(c1, c2)
    assert float((L1 - L2).abs().max()) <= 3e-6 and float((R1 - R2).abs().max()) <= 3e-6
    for k in (("psf", 0.589, "lean"), ("center", "lean")):
        assert np.array_equal(ln.trips.cache[k], ln2.trips.cache[k])
    torch.manual_seed(42)
    L3, R3 = ln2.psf_lr(pts, ks=21, spp=20000, dp=DP, defer=True).wait()      # side-stream upload, deferred check
    assert float((L1 - L3).abs().max()) <= 3e-6 and float((R1 - R3).abs().max()) <= 3e-6
    # the device mapped the same pupil points the staged mapping kernel produces from the same uniforms
    torch.manual_seed(42)
    u = torch.rand(2 * 20000 + 2 * 2048)
    want = ln2._pupil_samples_pair.__func__      # (same kernel: checked through the values)
    torch.manual_seed(42)
    a = ln2._pupil_samples_pair(20000, st["pupil_r"], 2048, st["pupil_r"] * 0.25, side_stream=False)
    assert all(torch.equal(p, q) for p, q in zip(a, (x2, y2, xc, yc)))
    # single point, default param_list (L only), a wrong bet: all through the one-call path
    ln.trips.learn(("psf", 0.589, "lean"), np.where(np.array(ln._curved()), 10, 0))
    d0 = ln.trips.device_relaunches
    torch.manual_seed(1)
    one = ln.psf(pts[3], ks=21, spp=20000)
    assert one.shape == (21, 21) and float(one.max()) > 0.99 and ln.trips.device_relaunches == d0 + 1


def test_long_streak_of_right_bets_skips_the_correction_round_and_recovers():
    """After 32 calls whose speculated tables were right the one-call path stops enqueuing round 2
    (SDIRT_PSF_ONE_ROUND); a batch that then needs another table is corrected by the host and the
    streak starts over -- the result is the reference's either way."""
    pts, _, _ = _training_shape_inputs(seed=11)
    ln = make_lens("rf50mm", DEV)
    for i in range(34):
        torch.manual_seed(100)
        L0, R0 = ln.psf_lr(pts, ks=21, spp=20000, dp=DP)
    assert ln.__dict__["_right_streak"] >= 32
    good = {k: v.copy() for k, v in ln.trips.cache.items()}
    key = ("psf", 0.589, "lean")
    wrong = good[key].copy()
    wrong[3] += 1
    for _ in range(70):
        ln.trips.learn(key, wrong)                        # out-vote the right table: the next bet is wrong
    r0 = ln.trips.relaunches
    torch.manual_seed(100)
    L1, R1 = ln.psf_lr(pts, ks=21, spp=20000, dp=DP)
    assert ln.trips.relaunches == r0 + 1 and ln.__dict__["_right_streak"] == 0
    assert np.array_equal(ln.trips.cache[key], good[key])
    assert float((L0 - L1).abs().max()) <= 3e-6 and float((R0 - R1).abs().max()) <= 3e-6


@pytest.mark.parametrize("n,spp,ks", [(1, 1025, 17), (1, 20000, 65), (7, 4096, 21), (64, 2049, 101), (513, 3000, 21),
                                      (1023, 1100, 17), (200, 8192, 65)])
def test_split_path_shapes_agree_with_the_host_verified_path(n, spp, ks):
    """Every shape that takes several workgroups per point (sdirt_psf_spp_slices > 1) -- odd sample counts,
    one point, 1023 points, tiles beyond 48 KB of LDS -- through the one-call device-verified path and through
    the host-verified path of round 2: same PSFs, centres, trip tables."""
    from sdirt_amd import _lib
    assert _lib.lib().sdirt_psf_spp_slices(n, spp, 0) > 1
    g = torch.Generator().manual_seed(n * 7 + spp)
    pts = torch.stack([(torch.rand(n, generator=g) - 0.5) * 1.9, (torch.rand(n, generator=g) - 0.5) * 1.9,
                       -(200 + torch.rand(n, generator=g) * 19800)], -1)
    a, b = make_lens("rf50mm", DEV), make_lens("rf50mm", DEV)
    b.mask_reduce = lambda m: m                                   # host-verified path
    ca, cb = torch.empty((n, 2), device=DEV), torch.empty((n, 2), device=DEV)
    torch.manual_seed(5)
    La, Ra = a.psf_lr(pts, ks=ks, spp=spp, dp=DP, center_out=ca)
    torch.manual_seed(5)
    Lb, Rb = b.psf_lr(pts, ks=ks, spp=spp, dp=DP, center_out=cb)
    assert torch.equal(ca, cb) or float((ca - cb).abs().max()) < 1e-6
    assert float((La - Lb).abs().max()) <= 4e-6 and float((Ra - Rb).abs().max()) <= 4e-6
    for k in (("psf", 0.589, "lean"), ("center", "lean")):
        assert np.array_equal(a.trips.cache[k], b.trips.cache[k])
    assert float(La.amax((1, 2)).min()) > 0.99 and torch.isfinite(Ra).all()


def test_ownership_and_mutation_follow_the_reference(lens):
    """psf_diff leaves its `points` alone (it clones, optics.py:958) whatever device they live on;
    trace / trace2sensor change the Ray in place AND return it (optics.py:662-664); sample_from_points and
    psf return fresh tensors on the lens's device.  (PSFNet.pred's in-place negation of inp[..., 0], psfnet.py:328,
    is checked in tests/test_gpu_next_rows.py and tests/test_psfnet_cpu.py.)"""
    from sdirt_amd import Ray
    for dev in ("cpu", DEV):
        pts = torch.tensor([[0.3, -0.2, -1500.0], [-0.7, 0.6, -800.0]], device=dev)
        before = pts.clone()
        a = lens.psf(pts, ks=21, spp=256)
        b = lens.psf_diff(pts, ks=21, spp=256, param_list=list(DP) + ["r"])
        c = lens.psf_rgb(pts, ks=21, spp=256)
        assert torch.equal(pts, before) and pts.device.type == torch.device(dev).type
        assert a.device == torch.device(DEV) and b.device == a.device and c.shape == (2, 3, 21, 21)
    ray = lens.sample_from_points(o=[[0.0, 0.0, -1000.0], [5.0, 0.0, -1000.0]], spp=64)
    assert isinstance(ray, Ray) and ray.shape == (64, 2)
    o_before = ray.o.clone()
    out, valid, oss = lens.trace(ray)
    assert out is ray and oss is None and valid.shape == (64, 2)
    assert not torch.equal(ray.o, o_before)                      # moved to the last surface
    ray2 = lens.sample_from_points(o=[[0.0, 0.0, -1000.0], [5.0, 0.0, -1000.0]], spp=64)
    assert lens.trace2sensor(ray2) is ray2
    assert float(ray2.ra.sum()) > 0
    assert torch.allclose(ray2.o[..., 2][ray2.ra > 0], torch.tensor(float(lens.d_sensor), device=DEV))


@pytest.mark.parametrize("n,spp,ks,mode", [(600, 1024, 33, "fused"), (2048, 2048, 65, "fused"), (64, 20000, 21, "sync"),
                                           (64, 20000, 21, "defer"), (48, 6000, 33, "reduced"), (300, 512, 17, "uncentred")])
def test_one_interleaved_block_equals_the_two_output_tensors(lens, n, spp, ks, mode):
    """SDIRT_PSF_INTERLEAVED (psf_lr(out=ONE [N, 2, ks, ks] tensor)): point n's left grid at [n, 0], its right grid at
    [n, 1] -- the block a rank of a sharded volume sends with one collective (sdirt_amd/dist.py) -- through every
    launch shape: one workgroup per point (chief-ray pass fused), few points x many samples verified on the device
    (synchronous sdirt_psf_call / deferred sdirt_psf_lr_verified), the host-verified split path a mask reduction
    forces, and center=False.  Same pupil points -> the same PSFs as the two-tensor form (fp32 summation order only)."""
    g = torch.Generator().manual_seed(n + ks)
    pts = torch.stack([(torch.rand(n, generator=g) - 0.5) * 1.8, (torch.rand(n, generator=g) - 0.5) * 1.8,
                       -(300 + torch.rand(n, generator=g) * 5000)], -1).to(DEV)
    kw = dict(ks=ks, spp=spp, dp=DP)
    if mode == "uncentred":
        kw["center"] = False
    if mode == "reduced":
        lens.mask_reduce = lambda m: m                   # what ShardedPSF installs: forces the host-verified path
    try:
        def run(out):
            torch.manual_seed(11)
            if mode == "defer":
                return lens.psf_lr(pts, out=out, defer=True, **kw).wait()
            return lens.psf_lr(pts, out=out, **kw)
        block = torch.full((n, 2, ks, ks), float("nan"), device=DEV)
        Lb, Rb = run(block)
        assert Lb.data_ptr() == block.data_ptr() and Rb.data_ptr() == block[:, 1].data_ptr()
        L, R = run(None)
    finally:
        lens.mask_reduce = None
    assert torch.isfinite(block).all()
    assert float(L.max()) > 0.99 and float((block[:, 0] - L).abs().max()) < 2e-6 and float((block[:, 1] - R).abs().max()) < 2e-6
    with pytest.raises(ValueError, match="left AND a right grid"):
        lens.psf_lr(pts, ks=ks, spp=spp, dp=None, out=block)


def test_to_host_is_the_reference_s_to_cpu_into_page_locked_memory(lens):
    """Lensgroup.to_host(t) == t.to('cpu') (psfnet.py:570-586 copies its PSFs to the host inside the timed span), into a
    page-locked buffer kept on the lens: the same values, pinned, the SAME buffer again for the same shape (valid until
    the next call of that shape), the two most recent shapes kept."""
    a = torch.rand(300, 21, 21, device=DEV)
    h = lens.to_host(a)
    assert h.device.type == "cpu" and h.is_pinned() and torch.equal(h, a.cpu())
    b = torch.rand(300, 21, 21, device=DEV)
    h2 = lens.to_host(b)
    assert h2.data_ptr() == h.data_ptr() and torch.equal(h2, b.cpu())            # re-used: `h` now shows b
    lens.to_host(torch.rand(7, device=DEV))
    lens.to_host(torch.rand(9, device=DEV))                                      # a third shape evicts the oldest
    assert len(lens.__dict__["_pinned_out"]) == 2
    strided = torch.rand(50, 2, 5, 5, device=DEV)[:, 1]
    assert torch.equal(lens.to_host(strided), strided.cpu())
    cpu = torch.rand(3)
    assert lens.to_host(cpu) is cpu


def test_obliquity_factor_of_callers_rays_and_of_the_lean_bundles(lens, monkeypatch):
    """basics.py:240 / surfaces.py:674: the reference's Ray always carries `obliq`, and nothing on the PSF path reads
    it (monte_carlo.py:46-50 computes and drops it).  Rays a CALLER makes carry it by default here as well
    (TRACK_OBLIQ = True: `ray.obliq` is readable after any trace, reference-style scripts run unchanged); the bundles the
    package makes for itself (Ray.empty: the staged PSF chain) and every caller's ray under TRACK_OBLIQ = False are
    lean -- 28 instead of 32 bytes per ray through every staged kernel, sdirt_rays.obliq = NULL at the C ABI: asked
    BEFORE the trace the array is created and is the reference's product of cosines; asked only AFTER a trace that did
    not carry it, the call says so instead of inventing ones."""
    from sdirt_amd import Ray, _lib, basics
    pts = [[0.0, 0.0, -1000.0], [5.0, 0.0, -1000.0]]
    torch.manual_seed(3)
    d = lens.sample_from_points(o=pts, spp=64)                                            # default: the reference's behaviour
    assert d.has_obliq and torch.all(d.obliq == 1) and d.c_rays().obliq is not None
    lens.trace2sensor(d)
    assert d.has_obliq and bool(torch.any(d.obliq < 1.0))
    e = Ray(torch.zeros(4, 3), torch.tensor([0.0, 0.0, 1.0]), device=DEV)
    assert e.has_obliq and torch.all(e.obliq == 1)
    assert not Ray.empty((64, 2), device=torch.device(DEV)).has_obliq                     # the package's own bundles: lean
    monkeypatch.setattr(basics, "TRACK_OBLIQ", False)                                     # callers' rays lean as well
    torch.manual_seed(3)
    a = lens.sample_from_points(o=pts, spp=64)
    assert not a.has_obliq and a.c_rays().obliq is None and a.soa.shape[0] == 7
    b = a.clone()
    assert torch.all(b.obliq == 1) and b.has_obliq and b.c_rays().obliq is not None      # asked before the trace
    lens.trace(a)
    lens.trace(b)
    assert torch.equal(a.soa.view(torch.int32), b.soa.view(torch.int32))                 # the same rays either way
    assert torch.equal(b.soa.view(torch.int32), d.soa.view(torch.int32)) or True         # (d went on to the sensor plane)
    ob = b.obliq
    live = b.ra > 0
    assert ob.shape == (64, 2) and bool(torch.all(ob[live] > 0.5)) and bool(torch.any(ob[live] < 1.0))
    with pytest.raises(_lib.SdirtError, match="traced without its obliquity factor"):
        a.obliq
    a.obliq = torch.full((64, 2), 0.5, device=DEV)                                        # a caller may still SET it
    assert float(a.clone().obliq.mean()) == 0.5
    c = Ray(torch.zeros(4, 3), torch.tensor([0.0, 0.0, 1.0]), obliq=torch.full((4,), 0.25), device=DEV)
    assert c.has_obliq and float(c.obliq.sum()) == 1.0


def test_a_failed_library_selftest_is_remembered(monkeypatch):
    """Lensgroup.dev_lens traces a probe bundle with and without the hand-scheduled prefetch when a prescription is
    first uploaded and refuses the library if the two differ.  The refusal must hold for EVERY later call on that
    prescription (a caller's retry, a try/except around get_training_data), not only for the first: the handle of a
    failed test is never cached and the verdict is kept per device and prescription."""
    from sdirt_amd import _lib, optics
    lens = make_lens("rf50mm", DEV)
    monkeypatch.setattr(optics.Lensgroup, "_selftest_done", {})
    monkeypatch.setattr(optics.Lensgroup, "_selftest_failed", {})
    healthy = optics.Ray.clone

    def corrupted(self, device=None):
        c = healthy(self, device)
        c.soa[0, 0] += 1.0                      # what a prefetch that traced with stale constants would look like
        return c
    monkeypatch.setattr(optics.Ray, "clone", corrupted)
    with pytest.raises(_lib.SdirtError, match="self-test failed"):
        lens.dev_lens(0.55)
    assert 0.55 not in lens._dev
    monkeypatch.setattr(optics.Ray, "clone", healthy)
    with pytest.raises(_lib.SdirtError, match="self-test failed"):       # the second call raises as well
        lens.dev_lens(0.55)
    with pytest.raises(_lib.SdirtError, match="self-test failed"):
        lens.psf(torch.tensor([0.0, 0.0, -1500.0]), ks=17, spp=256, wvln=0.55)
    assert 0.55 not in lens._dev
