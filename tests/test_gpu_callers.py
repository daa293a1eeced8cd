"""The reference's plotting callers of the PSF path (SURVEY.md §8b "Callers": draw_mtf, draw_psf_radial,
draw_psf_map, deeplens/optics.py:1884-1956, 2041-2067) on the HIP path, against what the reference itself
handed to its plots (fixtures F24 / F25, oracle/gen_golden_callers.py), and the staged chain behind grids
larger than a workgroup's LDS (draw_mtf asks for ks 256).  Needs an MI355X: `-m gpu`.
"""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden, load_state, make_lens

pytestmark = pytest.mark.gpu
DP = [0.78, 1.44, 0.3, 0.5]
DEV = "cuda:0"


def test_draw_mtf_against_the_reference_plot(oracle, tmp_path):
    """F24.  (1) the reference's pupil sets handed over: psf_diff(ks=256) within the reference's own
    noise of its PSFs, trip tables equal, within 5e-6 of the oracle; (2) the seeded call, as a user would
    make it: the curves of the reference's figure; the figure is written; the generator ends where the
    reference's run left it (12 vectors drawn)."""
    st, g = load_state("rf50mm"), load_golden("f24_rf50_draw_mtf")
    lens = make_lens("rf50mm", DEV, st)
    for i, fov in enumerate(g["relative_fov"]):
        pt = torch.tensor([float(fov), float(fov), float(g["depth"])])
        L, R = lens.psf_lr(pt, ks=256, pupil_xy=(g["pupil_x"][i], g["pupil_y"][i]),
                           center_pupil_xy=(g["pupil_xc"][i], g["pupil_yc"][i]))
        assert L.shape == (256, 256) and R.shape == (256, 256)
        assert np.array_equal(lens.trips.cache[("trace", 0.589, 0, len(lens.surfaces), True, "lean")], g["trips"][i])
        assert np.array_equal(lens.trips.cache[("center", "lean")], g["trips_center"][i])
        lo, ro, co, ok = oracle.psf(st, pt[None].numpy(), g["pupil_x"][i], g["pupil_y"][i], g["pupil_xc"][i],
                                    g["pupil_yc"][i], 256, dp=DP)
        d_ref = np.abs(L.cpu().numpy() - g["psf"][i]).max()
        d_orc = max(np.abs(L.cpu().numpy() - lo[0]).max(), np.abs(R.cpu().numpy() - ro[0]).max())
        print(f"ks 256, field {fov}: vs reference {d_ref:.2e}, vs oracle {d_orc:.2e}")
        assert d_ref <= 6e-5 and d_orc <= 5e-6
    torch.manual_seed(int(g["seed"]))
    out = str(tmp_path / "mtf")
    curves = lens.draw_mtf(save_name=out)
    end = torch.rand(2)
    assert os.path.getsize(out + ".png") > 10000
    assert [c["fov"] for c in curves] == [0.0, 0.7, 1.0]
    for i, c in enumerate(curves):
        assert c["fov_deg"] == round([0.0, 0.7, 1.0][i] * float(g["hfov"]) * 57.3, 1)
        assert np.array_equal(c["freq"], g["freq"][i])
        assert np.abs(c["psf"].cpu().numpy() - g["psf"][i]).max() <= 5e-4          # 2048 rays, own pupil mapping
        assert np.abs(c["tangential"] - g["tangential"][i]).max() <= 2e-3
        assert np.abs(c["sagittal"] - g["sagittal"][i]).max() <= 2e-3
    torch.manual_seed(int(g["seed"]))
    torch.rand(12 * 2048)
    assert torch.equal(torch.rand(2), end)


def test_draw_psf_radial_against_the_reference_plot(tmp_path):
    """F25: the PSFs draw_psf_radial hands to make_grid -- reference pupil sets handed over, then seeded,
    linear and log-scaled -- and the image file (make_grid / save_image restated in sdirt_amd/plots.py)."""
    from sdirt_amd import plots
    st, g = load_state("rf50mm"), load_golden("f25_rf50_draw_psf_radial")
    lens = make_lens("rf50mm", DEV, st)
    xs = torch.linspace(0, 1, 3)
    for i in range(3):
        pt = torch.stack((xs[i], xs[i], torch.tensor(float(g["depth"]))))
        psf = lens.psf_rgb(pt, ks=51, pupil_xy=np.transpose(g["pupil"][i], (1, 0, 2)),
                           center_pupil_xy=np.transpose(g["pupil_c"][i], (1, 0, 2)))
        psf = (psf / psf.max()).cpu().numpy()
        d = np.abs(psf - g["psfs"][i])
        print(f"radial field {i}: vs reference max {d.max():.2e}")
        assert d.max() <= 1e-4
    for log_scale, key in ((False, "psfs"), (True, "psfs_log")):
        torch.manual_seed(int(g["seed"]))
        out = str(tmp_path / f"radial{int(log_scale)}.png")
        psfs = lens.draw_psf_radial(M=3, ks=51, log_scale=log_scale, save_name=out)
        assert len(psfs) == 3 and os.path.getsize(out) > 500
        for i in range(3):
            got, want = psfs[i].cpu().numpy(), g[key][i]
            # log(psf + 1e-9) spans 20.7 before the stretch: 5e-4 on a pixel of 1e-2 is 2.4e-3 after it
            lit = g["psfs"][i] > (1e-2 if log_scale else 1e-3)
            assert got.shape == (3, 51, 51)
            assert np.abs(got - want)[lit].max() <= (5e-3 if log_scale else 5e-4)
    img = plots.tile_grid(psfs, nrow=3)
    assert img.shape == (3, 53, 157) and np.all(img[:, 0, :] == 0) and np.all(img[:, :, 52] == 0)
    assert np.array_equal(img[:, 1:52, 53:104], psfs[1].cpu().numpy())


def test_draw_psf_map_tiles(tmp_path):
    """draw_psf_map (optics.py:1884-1931): psf_map at GEO_SPP * 30 samples, every field divided by its own
    maximum over the three colours; equal to the psf_rgb of the same seeded draw, tile by tile."""
    from sdirt_amd import plots
    lens = make_lens("rf50mm", DEV)
    grid, ks = 3, 21
    torch.manual_seed(5)
    m = plots.psf_map_normalised(lens, grid=grid, depth=-1500.0, ks=ks)
    assert m.shape == (3, grid * ks, grid * ks)
    torch.manual_seed(5)
    field = lens.point_source_grid(depth=-1500.0, grid=grid).reshape(grid * grid, 3)
    tiles = lens.psf_rgb(points=field, ks=ks, center=True, spp=2048 * 30)
    for i in range(grid * grid):
        r, c = divmod(i, grid)
        want = tiles[i] / tiles[i].max()
        assert torch.allclose(m[:, r * ks:(r + 1) * ks, c * ks:(c + 1) * ks], want, atol=3e-6)
        assert float(m[:, r * ks:(r + 1) * ks, c * ks:(c + 1) * ks].max()) == 1.0
    torch.manual_seed(5)
    img = lens.draw_psf_map(grid=grid, depth=-1500.0, ks=ks, log_scale=True, save_name=str(tmp_path / "m"))
    assert img.shape == (grid * ks, grid * ks, 3)
    # a second run of the same draw: 3e-6 of summation-order noise on a floor of 1e-3
    assert np.allclose(img, np.log(m.permute(1, 2, 0).cpu().numpy() + 1e-3), atol=5e-3)
    assert os.path.getsize(str(tmp_path / "m_psf1500.0mm_left.png")) > 1000


@pytest.mark.parametrize("lens_name,ks,n,spp", [("rf50mm", 160, 6, 4096), ("rf35mm", 255, 3, 2048)])
def test_staged_chain_for_large_grids_vs_oracle(oracle, lens_name, ks, n, spp):
    """ks > SDIRT_MAX_KS: sample -> chief centre -> trace -> forward_integral (grids in HBM) -> normalise,
    L and R, against the oracle on the same pupil points: trip tables equal, centres bit-equal, PSFs
    within the atomic-order noise of the fused kernel's bar."""
    st = load_state(lens_name)
    lens = make_lens(lens_name, DEV, st)
    rng = np.random.default_rng(ks)
    pts = np.stack([rng.uniform(-1, 1, n), rng.uniform(-1, 1, n), -rng.uniform(300, 15000, n)], 1).astype(np.float32)
    x2, y2 = oracle.pupil_samples(rng.random(spp, np.float32), rng.random(spp, np.float32), st["pupil_r"])
    xc, yc = oracle.pupil_samples(rng.random(2048, np.float32), rng.random(2048, np.float32), st["pupil_r"] * 0.25)
    cen = torch.empty((n, 2), device=DEV)
    L, R = lens.psf_lr(torch.tensor(pts), ks=ks, pupil_xy=(x2, y2), center_pupil_xy=(xc, yc), dp=DP, center_out=cen)
    lo, ro, co, ok, tp, tc = oracle.psf(st, pts, x2, y2, xc, yc, ks, dp=DP, return_trips=True)
    assert ok and L.shape == (n, ks, ks)
    assert np.array_equal(lens.trips.cache[("trace", 0.589, 0, len(lens.surfaces), True, "lean")], tp)
    assert np.array_equal(lens.trips.cache[("center", "lean")], tc)
    assert np.array_equal(cen.cpu().numpy(), co)
    dl, dr = np.abs(L.cpu().numpy() - lo).max(), np.abs(R.cpu().numpy() - ro).max()
    print(f"{lens_name} ks {ks}: L {dl:.2e} R {dr:.2e}")
    assert dl <= 5e-6 and dr <= 5e-6
    # psf_diff's public form: L only by default, R on request; single point; uncentred
    only_l = lens.psf_diff(torch.tensor(pts[0]), ks=ks, spp=512)
    assert only_l.shape == (ks, ks) and float(only_l.max()) > 0.99
    unc = lens.psf_diff(torch.tensor(pts[:2]), ks=ks, spp=512, center=False, param_list=DP + ["r"])
    assert unc.shape == (2, ks, ks)


def test_staged_chain_equals_the_fused_kernel(monkeypatch):
    """The same call through both routes (the limit lowered for the test): identical rays, centres and trip
    tables, so the PSFs differ by summation order only; and the same number of random vectors drawn."""
    from sdirt_amd import _lib
    lens = make_lens("rf50mm", DEV)
    g = torch.Generator().manual_seed(3)
    pts = torch.stack([torch.rand(40, generator=g) * 2 - 1, torch.rand(40, generator=g) * 2 - 1,
                       -(300 + 9000 * torch.rand(40, generator=g))], 1)
    torch.manual_seed(9)
    Lf, Rf = lens.psf_lr(pts, ks=65, spp=4096, dp=DP)
    end = torch.rand(2)
    monkeypatch.setattr(_lib, "MAX_KS", 64)
    torch.manual_seed(9)
    Ls, Rs = lens.psf_lr(pts, ks=65, spp=4096, dp=DP)
    assert torch.equal(torch.rand(2), end)
    assert float((Ls - Lf).abs().max()) <= 3e-6 and float((Rs - Rf).abs().max()) <= 3e-6
    rgb_s = lens.psf_rgb(pts[:4], ks=65, spp=1024)
    monkeypatch.setattr(_lib, "MAX_KS", 141)
    torch.manual_seed(9)
    lens.psf_lr(pts, ks=65, spp=4096, dp=DP)
    torch.rand(2)
    rgb_f = lens.psf_rgb(pts[:4], ks=65, spp=1024)
    assert rgb_s.shape == rgb_f.shape == (4, 3, 65, 65)
    assert float((rgb_s - rgb_f).abs().max()) <= 3e-6


def test_grid_size_limits():
    from sdirt_amd import _lib
    lens = make_lens("rf50mm", DEV)
    with pytest.raises(_lib.SdirtError, match="outside"):
        lens.psf(torch.tensor([0.0, 0.0, -1000.0]), ks=_lib.MAX_KS_STAGED + 1)
    with pytest.raises(ValueError, match="defer"):
        lens.psf_lr(torch.tensor([[0.0, 0.0, -1000.0]]), ks=200, defer=True)
    big = lens.psf(torch.tensor([0.0, 0.0, -1000.0]), ks=_lib.MAX_KS_STAGED, spp=256)
    assert big.shape == (1024, 1024) and float(big.max()) > 0.99


def test_ray_reaction_surface_by_surface_equals_trace():
    """Aspheric.ray_reaction(ray) (surfaces.py:391-520), the per-surface public call of the reference: walking a
    bundle through the lens one surface object at a time -- forward, then a second bundle backward from the sensor --
    gives the rays of Lensgroup.trace bit for bit (each surface's batch-wide Newton trip count verified on its own)."""
    lens = make_lens("rf50mm", DEV)
    pts = torch.tensor([[0.0, 0.0, -1500.0], [0.5, -0.4, -3000.0], [-0.9, 0.9, -300.0]])
    po = lens._points_to_object(pts)
    torch.manual_seed(4)
    whole = lens.sample_from_points(po, spp=512)
    step = whole.clone()
    lens.trace(whole)
    for s_ in lens.surfaces:
        out = s_.ray_reaction(step)
        assert out is step                                   # mutated in place AND returned (surfaces.py:676-677)
    assert torch.equal(whole.soa.view(torch.int32), step.soa.view(torch.int32))
    assert 0.5 < float(step.ra.mean()) <= 1.0
    # backward: from sensor-side points through the surfaces in reverse order
    from sdirt_amd.basics import Ray
    o = torch.tensor([[0.0, 0.0, float(lens.d_sensor)], [2.0, -1.0, float(lens.d_sensor)]]).repeat(64, 1)
    aim = torch.stack([torch.linspace(-4, 4, 128), torch.linspace(3, -3, 128), torch.full((128,), float(lens.surfaces[-1].d))], 1)
    back = Ray(o, aim - o, device=DEV)
    back_step = back.clone()
    lens.trace(back, forward=False)
    for s_ in lens.surfaces[::-1]:
        s_.ray_reaction(back_step)
    assert torch.equal(back.soa.view(torch.int32), back_step.soa.view(torch.int32))
