"""Host-side logic of sdirt_amd (no GPU): lens loading and materials, the Newton
trip-table speculation/verification, the C-ABI surface of libsdirt_dp.so, error
behaviour without a device."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest
import torch

from conftest import DATA, ROOT, load_state

REF_JSON = "/root/reference/lenses/{}/lens_web.json"


@pytest.mark.parametrize("name", ["rf50mm", "rf35mm"])
def test_lens_loader_and_materials(name):
    from sdirt_amd import Lensgroup
    st = load_state(name)
    lens = Lensgroup(os.path.join(DATA, f"{name}.json"), sensor_res=(512, 768),
                     post_computation=False, device="cpu")
    assert lens.aper_idx == st["aper_idx"]                        # optics.py:193-201
    assert lens.pixel_size == st["pixel_size"] and float(lens.r_last) == st["r_last"]
    assert len(lens.surfaces) == len(st["surfaces"])
    kinds = {0: "plane", 1: "sphere", 2: "asphere"}
    for s, ref in zip(lens.surfaces, st["surfaces"]):
        assert kinds[s.kind] == ref["kind"]
        assert s.r == ref["r"] and float(s.d) == ref["d"] and float(s.c) == ref["c"]
        assert float(s.k) == ref["k"]
        assert [float(a) for a in (s.ai if s.ai is not None else [])] == ref["ai"]
        for w, n1 in ref["n1"].items():                           # Material.ior, float64 equal
            assert s.mat1.ior(float(w)) == n1
            assert s.mat2.ior(float(w)) == ref["n2"][w]
    # the reference's own JSON schema loads to the same prescription (container only)
    if os.path.exists(REF_JSON.format(name)):
        l2 = Lensgroup(REF_JSON.format(name), sensor_res=(512, 768), post_computation=False,
                       device="cpu")
        for a, b in zip(lens.surfaces, l2.surfaces):
            assert a.surf_dict() == b.surf_dict()
    d = lens.surfaces[0].desc(0.589)
    assert d.kind == 1 and d.n1 == 1.0 and d.n2 == st["surfaces"][0]["n2"]["0.589"]


def test_material_models():
    from sdirt_amd import Material
    from sdirt_amd.basics import register_material
    assert Material("air").ior(0.589) == 1.0 and Material(None).ior(0.45) == 1.0
    m = Material("1.5168/64.17")
    assert abs(m.ior(0.5893) - 1.5168) < 1e-9                     # Cauchy through n_d
    assert m.ior(0.486) > m.ior(0.656)                            # normal dispersion
    assert m.ior(589.3) == m.ior(0.5893)                          # nm input (basics.py:324)
    register_material("testglass", "sellmeier", [1.0396, 6.0006e-3, 2.3179e-1, 2.0017e-2, 1.0104, 103.56])
    assert abs(Material("testglass").ior(0.5876) - 1.5168) < 2e-4
    with pytest.raises(ValueError):
        Material("unobtainium")


def test_point_source_grid_and_scale():
    from sdirt_amd import Lensgroup
    st = load_state("rf50mm")
    lens = Lensgroup(os.path.join(DATA, "rf50mm.json"), sensor_res=(512, 768),
                     post_computation=False, device="cpu").set_state(hfov=st["hfov"])
    g = lens.point_source_grid(depth=-1000.0, grid=5)
    assert g.shape == (5, 5, 3) and float(g[0, 0, 0]) == pytest.approx(-0.98)
    assert float(g[0, 0, 1]) == pytest.approx(0.98) and (g[..., 2] == -1000).all()
    gc = lens.point_source_grid(depth=-1000.0, grid=5, center=True)
    assert float(gc[0, 0, 0]) == pytest.approx(-1 + 1 / 8)
    assert lens.point_source_grid(depth=-5.0, grid=1).shape == (1, 1, 3)
    # every option combination, bit for bit against the reference (fixture F19)
    from conftest import load_golden
    ref = load_golden("f19_point_source_grid")
    for key in ref.files:
        grid, center, quater, normalized = (int(v[1:]) for v in key.split("_"))
        mine = lens.point_source_grid(depth=-1234.5, grid=grid, normalized=bool(normalized),
                                      quater=bool(quater), center=bool(center))
        assert mine.shape == ref[key].shape and np.array_equal(mine.numpy(), ref[key]), key
    assert lens.calc_scale_pinhole(-1000.0) == pytest.approx(1000 * np.tan(st["hfov"]) / st["r_last"])
    assert lens.calc_efl() == pytest.approx(st["foclen"], rel=1e-12)


def test_intersect_lines_2d():
    from sdirt_amd.optics import intersect_lines_2d
    o = np.array([[0.0, 0.0], [2.0, 0.0], [0.0, 3.0]])
    d = np.array([[1.0, 1.0], [-1.0, 1.0], [1.0, 0.0]])
    p = intersect_lines_2d(o, d)
    assert p.shape == (3, 2)
    assert np.allclose(p[0], [1, 1]) and np.allclose(p[1], [3, 3]) and np.allclose(p[2], [-1, 3])
    assert len(intersect_lines_2d(o[:2], np.array([[1.0, 0.0], [2.0, 0.0]]))) == 0   # parallel


# ------------------------------------------------------------------ newton.py
def masks_for(trips, need):
    """Mask a kernel would report: bit j set while the slowest ray has not converged."""
    out = []
    for T, t in zip(trips, need):
        m = 0
        for j in range(1, int(T) + 1):
            if j < t or t > 10:
                m |= 1 << j
        out.append(m)
    return out


def test_verify_accepts_exact_table_and_corrects_wrong_ones():
    from sdirt_amd.newton import verify
    curved = [True, True, False, True]
    need = [10, 3, 0, 4]
    ok, new = verify([10, 3, 0, 4], masks_for([10, 3, 0, 4], need), range(4), curved)
    assert ok and list(new) == [10, 3, 0, 4]
    # too many trips on surface 1: the first clear bit is the reference's count
    ok, new = verify([10, 6, 0, 4], masks_for([10, 6, 0, 4], need), range(4), curved)
    assert not ok and new[1] == 3
    # too few: one more trip is the next guess
    ok, new = verify([10, 2, 0, 4], masks_for([10, 2, 0, 4], need), range(4), curved)
    assert not ok and new[1] == 3
    # never converging surface stays at the cap and is accepted (rule `it < 10`)
    ok, new = verify([10, 3, 0, 4], masks_for([10, 3, 0, 4], [11, 3, 0, 4]), range(4), curved)
    assert ok
    # reversed traversal order (backward tracing)
    ok, new = verify([4, 0, 3], masks_for([4, 0, 3], [4, 0, 3]), [2, 1, 0], [True, False, True])
    assert ok


def test_trip_planner_converges_and_caches():
    from sdirt_amd.newton import TripPlanner
    curved = [True] * 5 + [False] + [True] * 6
    need = [10, 3, 4, 3, 4, 0, 3, 3, 4, 4, 2, 3]                  # fixture f2's reference trips
    calls = []

    def launch(trips):
        calls.append(list(trips))
        return masks_for(trips, need)
    pl = TripPlanner()
    got = pl.run("k", curved, range(12), launch)
    assert list(got) == need and len(calls) == 2                  # cold: one discovery + one verify
    got = pl.run("k", curved, range(12), launch)
    assert list(got) == need and len(calls) == 3                  # warm: single launch
    # batch changes (one surface now needs one trip more): one corrective relaunch
    need[2] = 5
    got = pl.run("k", curved, range(12), launch)
    assert list(got) == need and len(calls) == 5                  # wrong + corrected
    # two trips short: one more, then the full 10 to read the exact count off the mask
    need[2] = 7
    n = len(calls)
    got = pl.run("k", curved, range(12), launch)
    assert list(got) == need and len(calls) - n == 4


def test_trip_planner_bets_on_the_most_frequent_table():
    """Batches that flip between two neighbouring tables: speculate the commoner one."""
    from sdirt_amd.newton import TripPlanner
    curved = [True, True]
    tables = {"a": [10, 3], "b": [10, 4]}
    pl, launched = TripPlanner(), []
    for which in "aaabaaabaaab":
        need = tables[which]

        def launch(trips, need=need):
            launched.append(list(trips))
            return masks_for(trips, need)
        assert list(pl.run("k", curved, range(2), launch)) == need
    # after the first 'b', a last-value predictor would also miss the 'a' that follows it
    first_guesses = []
    i = 0
    for which in "aaabaaabaaab":
        first_guesses.append(launched[i])
        i += 1 if launched[i] == tables[which] else 2
    assert first_guesses[4:] == [tables["a"]] * 8


# ------------------------------------------------------------------ C ABI
def header_symbols():
    txt = open(os.path.join(ROOT, "include", "sdirt_dp.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return set(re.findall(r"\b(sdirt_[a-z0-9_]+)\s*\(", txt))


def test_library_exports_every_declared_symbol():
    from sdirt_amd import _lib
    syms = header_symbols()
    assert len(syms) >= 19
    assert syms == set(_lib.SIGNATURES), syms ^ set(_lib.SIGNATURES)
    h = _lib.lib()                                                # loads, or raises loudly
    for s in syms:
        assert hasattr(h, s), f"libsdirt_dp.so does not export {s}"
    assert h.sdirt_abi_version() == 4
    # ... and nothing but them (no kernel stubs, no experiment hooks)
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = {l.split()[-1] for l in out.splitlines() if l.split()[-2] in "TDB"}
    assert exported == syms, exported ^ syms


def test_python_constants_match_the_c_header():
    """Every numeric SDIRT_* macro the binding mirrors has the header's value."""
    from sdirt_amd import _lib
    txt = open(os.path.join(ROOT, "include", "sdirt_dp.h")).read()
    macros = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+SDIRT_([A-Z0-9_]+)\s+(-?\d+)u?\b", txt)}
    mirrored = {k: v for k, v in macros.items() if hasattr(_lib, k)}
    assert {"MAX_SURFACES", "MAX_KS", "MAX_KS_STAGED", "PSF_NORMALIZE", "PSF_ONE_ROUND", "CTL_WORDS",
            "TRACE_NO_PREFETCH", "NEWTON_MAXITER", "MAX_WAVELENGTHS", "PSF_INTERLEAVED"} <= set(mirrored)
    for k, v in mirrored.items():
        assert getattr(_lib, k) == v, (k, v, getattr(_lib, k))


def test_ctypes_structs_match_the_c_header(tmp_path):
    """Compile include/sdirt_dp.h as plain C and compare struct layouts with the ctypes mirror."""
    import subprocess
    from sdirt_amd import _lib
    src = tmp_path / "abi.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "sdirt_dp.h"\n'
                   'int main(void){printf("%zu %zu %zu %zu %zu %zu %zu %d\\n",'
                   'sizeof(sdirt_surface_desc), offsetof(sdirt_surface_desc, r),'
                   'offsetof(sdirt_surface_desc, ai), offsetof(sdirt_surface_desc, n1),'
                   'sizeof(sdirt_rays), offsetof(sdirt_rays, obliq), sizeof(sdirt_dp_params),'
                   'SDIRT_MAX_SURFACES);return 0;}\n')
    exe = tmp_path / "abi"
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror",
                           "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    S = _lib.SurfaceDesc
    assert got == [ctypes.sizeof(S), S.r.offset, S.ai.offset, S.n1.offset, ctypes.sizeof(_lib.Rays),
                   _lib.Rays.obliq.offset, ctypes.sizeof(_lib.DpParams), _lib.MAX_SURFACES]


def test_no_product_import_of_the_oracle():
    """The product must never route through oracle/ (or any CPU path)."""
    for root, _, files in os.walk(os.path.join(ROOT, "sdirt_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(root, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
                assert "sdirt_oracle" not in src, f


def test_argument_validation_without_a_device():
    from sdirt_amd import _lib
    h = _lib.lib()
    assert h.sdirt_lens_create(None, 3, None) == -1
    assert b"null" in h.sdirt_last_error()
    arr = (_lib.SurfaceDesc * 1)()
    arr[0].kind = 7
    out = ctypes.c_void_p()
    assert h.sdirt_lens_create(arr, 1, ctypes.byref(out)) == -1 and b"kind" in h.sdirt_last_error()
    arr[0].kind, arr[0].c = 1, 0.0                                # sphere with c == 0
    arr[0].n1 = arr[0].n2 = 1.0
    assert h.sdirt_lens_create(arr, 1, ctypes.byref(out)) == -1
    assert h.sdirt_lens_create(arr, 0, ctypes.byref(out)) == -1
    assert h.sdirt_psf_normalize(None, 1, 5, None) == -1
    with pytest.raises(_lib.SdirtError):
        _lib.check(h.sdirt_points_to_object(None, 1, 0.1, 1.0, 36.0, 24.0, None, None))


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_tracing_without_gpu_fails_loudly():
    from sdirt_amd import Lensgroup, SdirtError
    from sdirt_amd import monte_carlo
    from sdirt_amd.render_psf import local_psf_render_fast
    lens = Lensgroup(os.path.join(DATA, "rf50mm.json"), sensor_res=(512, 768),
                     post_computation=False, device="cpu")
    lens.set_state(hfov=0.39, pupil=(22.5, 6.0))
    with pytest.raises(SdirtError):
        lens.psf(torch.tensor([[0.0, 0.0, -1000.0]]), ks=5, spp=8)
    with pytest.raises(SdirtError):
        lens.post_computation()                                   # geometric optics trace too
    with pytest.raises(SdirtError):
        monte_carlo.assign_points_to_pixels_small_r(torch.zeros(4, 2), 5, [-1, 1], [-1, 1],
                                                    torch.ones(4), x_tan=torch.zeros(4))
    with pytest.raises(SdirtError):
        local_psf_render_fast(torch.zeros(1, 3, 4, 4), torch.zeros(1, 4, 4, 2, 3, 3), 3)


def test_global_psf_convolution_matches_reference_fixture():
    """render_psf / render_psf_map (render_psf.py:12-73) are stock torch convolutions and run on
    any device; checked here on CPU against the reference's output."""
    from conftest import load_golden
    from sdirt_amd import render_psf, render_psf_map
    g = load_golden("f7_render")
    out = render_psf(torch.tensor(g["img"]), torch.tensor(g["psf_global"]))
    assert np.abs(out.numpy() - g["global"]).max() < 1e-6
    # a 1x1 "map" is the same thing
    out2 = render_psf_map(torch.tensor(g["img"]), torch.tensor(g["psf_global"]), 1)
    assert np.abs(out2.numpy() - g["global"]).max() < 1e-6
    # a 2x2 map of the same PSF: identical away from nothing -- tiles read across their borders
    tiled = torch.tensor(g["psf_global"]).repeat(1, 2, 2)
    out3 = render_psf_map(torch.tensor(g["img"]), tiled, 2)
    assert np.abs(out3.numpy() - g["global"]).max() < 1e-6


def test_host_pupil_mapping_expressions_match_the_reference():
    """pupil_mapping='host' evaluates optics.py:483-486 with torch CPU ops: on the machine that
    generated the fixtures the sample points are bit-identical, elsewhere (other MKL code
    path) within 1 ulp."""
    from conftest import load_golden, load_state, ulp_diff
    st, g = load_state("rf50mm"), load_golden("f8_rf50_mini_c2")
    torch.manual_seed(8)
    u_theta, u_r2 = torch.rand(4096), torch.rand(4096)
    assert np.array_equal(u_theta.numpy(), g["u_theta"]) and np.array_equal(u_r2.numpy(), g["u_r2"])
    theta = u_theta * 2 * np.pi
    r = torch.sqrt(u_r2 * st["pupil_r"] ** 2)
    x2, y2 = (r * torch.cos(theta)).numpy(), (r * torch.sin(theta)).numpy()
    assert ulp_diff(x2, g["pupil_x2"]).max() <= 2 and ulp_diff(y2, g["pupil_y2"]).max() <= 2


# ------------------------------------------------------------------ property tests (hypothesis)
def _needs():
    from hypothesis import strategies as st
    return st.lists(st.integers(min_value=0, max_value=11), min_size=1, max_size=24)


def test_trip_planner_always_lands_on_the_reference_table():
    """Whatever table is speculated first, launch/verify rounds end on exactly the per-surface
    counts the reference's batch-wide loop would run (need 11 = never converges -> cap 10)."""
    from hypothesis import given, settings, strategies as st
    from sdirt_amd.newton import TripPlanner

    @settings(max_examples=200, deadline=None)
    @given(_needs(), st.data())
    def check(need, data):
        curved = [n > 0 for n in need]
        want = [min(n, 10) if c else 0 for n, c in zip(need, curved)]
        order = list(range(len(need)))
        if data.draw(st.booleans()):
            order.reverse()                                   # backward tracing
        pl = TripPlanner()
        guess = data.draw(st.lists(st.integers(1, 10), min_size=len(need), max_size=len(need)))
        pl.cache["k"] = np.asarray([g if c else 0 for g, c in zip(guess, curved)], np.int32)
        rounds = []

        def launch(trips):
            rounds.append(list(trips))
            return masks_for(trips, need)
        got = pl.run("k", curved, order, launch)
        assert list(got) == want
        assert rounds[-1] == want and len(rounds) <= 3 * len(need) + 3
        # and a second call with the same batch is a single launch
        n = len(rounds)
        assert list(pl.run("k", curved, order, launch)) == want and len(rounds) == n + 1
    check()


def test_shard_bounds_partition_properties():
    from hypothesis import given, settings, strategies as st
    from sdirt_amd.dist import shard_bounds

    @settings(max_examples=200, deadline=None)
    @given(st.integers(0, 100000), st.integers(1, 64))
    def check(n, world):
        b = shard_bounds(n, world)
        assert len(b) == world and b[0][0] == 0 and b[-1][1] == n
        assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
        sizes = [hi - lo for lo, hi in b]
        assert max(sizes) - min(sizes) <= 1
    check()


def test_one_rand_call_equals_the_reference_s_four_consecutive_draws():
    """Lensgroup._pupil_samples_pair draws 2 spp + 2 x 2048 uniforms with ONE torch.rand call; the
    reference makes four (optics.py:483-484 in sample_from_points, then again inside psf_center).
    Same generator state, same numbers, whatever the sizes."""
    import torch
    for spp in (1, 17, 256, 4096, 20000):
        torch.manual_seed(spp)
        four = torch.cat([torch.rand(spp), torch.rand(spp), torch.rand(2048), torch.rand(2048)])
        after4 = torch.rand(3)
        torch.manual_seed(spp)
        one = torch.rand(2 * spp + 4096)
        after1 = torch.rand(3)
        assert torch.equal(four, one) and torch.equal(after4, after1)


def test_fast_host_random_stream_is_torch_rand():
    """sdirt_host_uniform_fill (block-wise MT19937) against torch.rand on the default CPU generator: same
    numbers and same generator state afterwards, for sizes around the 624-number block boundary, after
    other draws, interleaved with torch's own draws (randn in between)."""
    import torch
    from sdirt_amd import _hostrng
    assert _hostrng.enabled()
    for n in (1, 2, 619, 623, 624, 625, 1247, 4096, 44096):
        for pre in (0, 1, 620, 624, 1000):
            torch.manual_seed(1000 + n)
            if pre:
                torch.rand(pre)
            want = torch.rand(n)
            tail = (torch.randn(5), torch.rand(7))
            torch.manual_seed(1000 + n)
            if pre:
                torch.rand(pre)
            got = _hostrng.rand_into(torch.empty(n))
            tail2 = (torch.randn(5), torch.rand(7))
            assert torch.equal(got, want), (n, pre)
            assert torch.equal(tail[0], tail2[0]) and torch.equal(tail[1], tail2[1]), (n, pre)


def test_psf2mtf_matches_the_reference():
    """Lensgroup.psf2mtf (optics.py:1043-1080, the FFT consumer behind draw_mtf), fixture F23."""
    from conftest import load_golden, make_lens
    g = load_golden("f23_psf2mtf")
    lens = make_lens("rf50mm", "cpu")
    assert lens.pixel_size == float(g["pixel_size"])
    freq, tan, sag = lens.psf2mtf(g["psf"])
    assert np.array_equal(freq, g["freq"])
    np.testing.assert_allclose(tan, g["tangential"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(sag, g["sagittal"], rtol=1e-6, atol=1e-7)


def test_plot_helpers_restate_make_grid_and_save_image(tmp_path):
    """sdirt_amd/plots.py: tile_grid = torchvision make_grid for equal tiles (padding on the top / left of every
    tile and once more at the bottom / right), save_normalised = save_image(normalize=True)."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    from sdirt_amd import plots
    g = torch.Generator().manual_seed(0)
    tiles = [torch.rand(3, 5, 4, generator=g) for _ in range(5)]
    img = plots.tile_grid(tiles, nrow=3, padding=1, pad_value=0.25)
    assert img.shape == (3, 2 * 6 + 1, 3 * 5 + 1)
    assert np.array_equal(img[:, 1:6, 1:5], tiles[0].numpy()) and np.array_equal(img[:, 7:12, 6:10], tiles[4].numpy())
    assert np.all(img[:, 0] == 0.25) and np.all(img[:, 6] == 0.25) and np.all(img[:, :, 5] == 0.25)
    assert np.all(img[:, 7:12, 11:15] == 0.25)                     # the empty sixth cell
    one = plots.tile_grid(tiles[:2], nrow=8, padding=0)
    assert one.shape == (3, 5, 8) and np.array_equal(one[:, :, 4:], tiles[1].numpy())
    path = str(tmp_path / "t.png")
    rgb = plots.save_normalised(img, path)
    lo, hi = float(img.min()), float(img.max())
    want = np.clip((img - lo) / (hi - lo + 1e-5) * 255 + 0.5, 0, 255).astype(np.uint8).transpose(1, 2, 0)
    assert np.array_equal(rgb, want)
    back = (plt.imread(path)[..., :3] * 255 + 0.5).astype(np.uint8)
    assert np.array_equal(back, want)


def test_import_aliases_and_script_helpers(tmp_path):
    """sdirt_amd/compat: `deeplens.*` as the reference's scripts import it resolves to this package; set_seed /
    set_logger behave as the reference's (utils.py:136-164)."""
    import subprocess
    from sdirt_amd import compat
    code = ("import deeplens, sdirt_amd\n"
            "from deeplens.psfnet import *\n"
            "from deeplens.utils import set_logger, set_seed\n"
            "from deeplens.optics import Lensgroup\n"
            "from deeplens.monte_carlo import forward_integral, assign_points_to_pixels_small_r\n"
            "from deeplens.render_psf import local_psf_render_fast, local_dp_psf_render\n"
            "assert PSFNet is sdirt_amd.PSFNet and Lensgroup is sdirt_amd.Lensgroup and deeplens.Ray is sdirt_amd.Ray\n"
            "assert DMIN == 200 and DMAX == 20000 and GEO_SPP == 2048 and nn is torch.nn\n"
            "print('ok')\n")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([compat.path(), ROOT]))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr[-2000:]
    import logging
    import random
    from sdirt_amd.utils import set_logger, set_seed
    set_seed(5)
    a = (random.random(), np.random.rand(), torch.rand(3))
    set_seed(5)
    b = (random.random(), np.random.rand(), torch.rand(3))
    assert a[0] == b[0] and a[1] == b[1] and torch.equal(a[2], b[2])
    root = logging.getLogger()
    before = list(root.handlers)
    try:
        set_logger(str(tmp_path))
        logging.info("hello from the test")
        for h in root.handlers:
            h.flush()
        assert "INFO:hello from the test" in open(tmp_path / "output.log").read()
    finally:
        for h in [h for h in root.handlers if h not in before]:
            root.removeHandler(h)
            h.close()


def test_host_sag_for_drawing():
    """Aspheric.surface / surface_with_offset (surfaces.py:172-175, 766-771): sphere sag in closed form, apex
    value outside the conic's domain, polynomial terms."""
    from sdirt_amd.surfaces import Aspheric
    s = Aspheric(8.0, 2.5, c=1 / 20.0)
    r = np.linspace(-8, 8, 9, dtype=np.float32)
    want = 20.0 - np.sqrt(400.0 - r.astype(np.float64) ** 2)
    assert np.abs(s.surface(r, np.zeros_like(r)) - want).max() < 2e-6
    assert np.abs(s.surface_with_offset(r, np.zeros_like(r)) - (want + 2.5)).max() < 2e-6
    assert s.surface(np.float32(25.0), np.float32(0.0)) == 0.0          # outside the sphere: evaluated at the apex
    a = Aspheric(5.0, 0.0, c=0.05, k=-1.5, ai=[1e-3, 2e-5])
    r2 = 9.0
    want = r2 * 0.05 / (1 + np.sqrt(1 - (1 - 1.5) * r2 * 0.05 ** 2)) + 1e-3 * r2 + 2e-5 * r2 ** 2
    assert abs(float(a.surface(np.float32(3.0), np.float32(0.0))) - want) < 1e-6
    flat = Aspheric(5.0, 1.0, c=0.0)
    assert np.all(flat.surface_with_offset(r, r) == 1.0)


def test_forward_integral_launch_plan_for_every_grid_and_batch_shape():
    """sdirt_forward_integral_plan (pure host arithmetic, no GPU): whatever (points, spp, ks, L / L+R, CU count), the
    launch keeps a workgroup's tiles inside the 160 KiB of LDS, covers every point and every sample exactly once, prefers
    float64 accumulators while two tiles fit, falls back to float tiles and then to the HBM path, and cuts the spp axis
    only when the points alone leave CUs idle."""
    import ctypes as C
    import itertools
    from sdirt_amd import _lib
    h = _lib.lib()
    plan = (C.c_int64 * 6)()
    seen = set()
    for n, s, ks, both, cus in itertools.product([1, 2, 3, 64, 300, 2048, 16384, 65537], [1, 50, 64, 200, 1024, 4096, 20000],
                                                 [2, 9, 21, 65, 99, 100, 120, 141, 142, 256, 1024], [0, 1], [32, 256]):
        assert h.sdirt_forward_integral_plan(n, s, ks, both, cus, plan) == 0
        acc, P, groups, nsplit, chunk, lds = list(plan)
        tile = (2 if both else 1) * ks * ks
        seen.add(acc)
        if acc == 0:
            assert (tile | 1) * 4 > 160 * 1024 - 1024              # not even one float tile set fits
            continue
        assert acc == 8 or (tile | 1) * 8 > 160 * 1024 - 1024      # float tiles only when double ones do not fit
        assert lds == acc * P * (tile | 1) <= 160 * 1024 - 1024
        assert P in (1, 2, 4, 8) and groups == -(-n // P)
        rp = 512 // P                                               # 512-thread workgroups, whole waves per point
        assert chunk % rp == 0 and chunk >= rp
        assert nsplit == -(-s // chunk) and nsplit * chunk >= s > (nsplit - 1) * chunk
        if P > 1:
            assert rp >= s                                          # several points per workgroup only for short rows
        if nsplit > 1:
            assert groups < 2 * cus
    assert seen == {0, 4, 8}
    assert h.sdirt_forward_integral_plan(0, 10, 21, 1, 256, plan) != 0 and b"bad argument" in h.sdirt_last_error()


def test_bench_keeps_stdout_for_its_one_json_line(tmp_path):
    """The driver reads ONE JSON line from bench.py's stdout.  Native libraries print there too (RCCL's version banner when
    its first communicator comes up): bench.claim_stdout() points file descriptor 1 at stderr and emit_line() writes to
    the descriptor stdout had at start-up."""
    import subprocess
    import sys
    detail = str(tmp_path / "detail.json")
    code = ("import os, sys, json; sys.path.insert(0, %r); import bench; bench.DETAIL_PATH = %r; bench.claim_stdout(); "
            "os.write(1, b'RCCL version : banner\\n'); print('a stray print'); "
            "bench.emit_line({'metric': 'm', 'value': 1.5, 'kernels': {'a long table': list(range(3000))}}); "
            "os.write(1, b'more noise\\n')" % (ROOT, detail))
    p = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    import json
    assert len(lines) == 1 and json.loads(lines[0]) == {"metric": "m", "value": 1.5, "config": {}, "detail": detail}, p.stdout
    assert "RCCL version : banner" in p.stderr and "a stray print" in p.stderr and "more noise" in p.stderr
    # the full record went to the detail file
    assert json.load(open(detail))["kernels"]["a long table"][-1] == 2999


def test_the_one_json_line_stays_under_4_kb_and_keeps_the_evidence():
    """a 16 KB line lost its `also` values and the VALU fractions in the driver's `parsed`.  bench.compact()
    on the newest committed full record of the default run (profiles/rNN/bench_c2_detail.json): under 4000 bytes, the
    contract's keys whole, the roofline fractions that say something as flat scalars, one triple per side workload,
    one row per shard size."""
    import glob
    import json
    import bench
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "bench_c2_detail.json")))
    assert files, "no committed full record of the default bench run"
    full = json.load(open(files[-1]))
    assert len(json.dumps(full)) > 8000                   # (the record really is the long form)
    line = bench.compact(full)
    text = json.dumps(line)
    assert len(text) < bench.LINE_LIMIT <= 4000, len(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["config"]["workload"].startswith("BASELINE config 2") and "model" not in line["config"]
    rf = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "valu_flops_frac", "valu_issue_frac"):
        assert k in rf and not isinstance(rf[k], dict), k
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-6 and 0.1 < rf["valu_flops_frac"] < 0.5
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    assert {"f1", "c4", "c3", "tcp", "c5", "staged_ks65"} <= set(line["also_summary"])
    assert all(len(v) == 3 for v in line["also_summary"].values() if isinstance(v, list))
    assert {"single_gpu_loop_16384", "world8_shard_2048", "world8_shard_2048_two_streams"} <= set(line["sweep_summary"])
    eff = line["sweep_columns"].index("efficiency")
    assert all(0.0 < row[eff] <= 1.03 for row in line["sweep_summary"].values())      # physical figures only


def test_the_multi_gpu_line_keeps_the_gather_evidence():
    """The N > 1 line: `gather` with the checksum flag and the algorithm trial (both figures, which one `value` is),
    `value_allgather` / `value_direct` / `value_no_gather` as top-level scalars -- under 4000 bytes with everything on."""
    import json
    import bench
    full = {"metric": "rays/sec rf50mm 65x65 DP-PSF @4096spp", "value": 5.1e10, "unit": "rays/s", "n_gpus": 8, "steps": 20, "warmup": 5,
            "ms_per_step": 1.31, "higher_is_better": True, "scaling": "strong", "world_size": 8, "vs_baseline": None, "dtype": "f32",
            "data": "synthetic", "backend": "nccl", "value_no_gather": 5.6e10, "ms_per_step_no_gather": 1.19,
            "value_allgather": 4.2e10, "value_direct": 5.1e10, "render_streams": 2, "kernel_ms": 1.15,
            "config": {"workload": "BASELINE config 2: " + "x" * 300, "name": "c2", "points_per_gpu": 2048, "spp": 4096, "ks": 65,
                       "parallelism": "p" * 200, "gather": True, "newton_trip_policy": "reference", "relaunches_in_timed_region": 0},
            "roofline": {"bound": "valu", "achieved": 60.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.0075, "traffic": 5.6e8,
                         "valu_flops": {"frac": 0.24, "achieved": 37.9}, "valu_issue": {"frac": 0.71, "issue_bound": {"frac": 0.93}, "stale": False},
                         "kernel": "k" * 100, "note": "n" * 500},
            "gather": {"algo": "direct", "backend": "nccl", "world_size": 8, "gb_received_per_rank_per_step": 0.484557,
                       "collectives_per_step": 1, "ms": 1.28, "GBps_received_per_rank": 378.0, "compute_ms": 1.15, "gather_bound": True,
                       "volume_checksums_equal": True, "block": "b" * 80, "what": "w" * 300,
                       "trial": {"allgather_ms_per_step": 1.6, "direct_ms_per_step": 1.31, "direct_volume_checksums_equal": True,
                                 "direct_gather_ms": 1.28, "adopted": "direct", "what": "w" * 300}}}
    line = bench.compact(full)
    assert len(json.dumps(line)) < bench.LINE_LIMIT
    g = line["gather"]
    assert g["volume_checksums_equal"] is True and g["algo"] == "direct" and g["gather_bound"] is True
    assert g["trial"] == {"allgather_ms_per_step": 1.6, "direct_ms_per_step": 1.31, "direct_volume_checksums_equal": True,
                          "direct_gather_ms": 1.28, "adopted": "direct"}
    assert line["value_allgather"] == 4.2e10 and line["value_direct"] == 5.1e10 and line["value_no_gather"] == 5.6e10
    assert line["scaling"] == "strong" and line["n_gpus"] == 8 and line["config"]["points_per_gpu"] == 2048


def test_spp_slices_is_host_arithmetic_with_an_explicit_cu_count():
    """sdirt_psf_spp_slices(n_points, spp, n_cus) with n_cus > 0 touches no device (this suite runs without one): one
    workgroup per point once the points alone give four workgroups per CU, the spp axis cut otherwise -- and the cut
    depends on the CU count (so do, in their last bits, the PSFs of a split call: include/sdirt_dp.h)."""
    from sdirt_amd import _lib
    h = _lib.lib()
    assert h.sdirt_psf_spp_slices(16384, 4096, 256) == 1 and h.sdirt_psf_spp_slices(1024, 4096, 256) == 1
    assert h.sdirt_psf_spp_slices(64, 20000, 256) == 16          # the PSFNet fitting shape on an MI355X: 16 x 1280 samples
    assert h.sdirt_psf_spp_slices(64, 20000, 32) == 2            # one XCD of a partitioned chip
    assert h.sdirt_psf_spp_slices(1023, 4096, 256) == 2 and h.sdirt_psf_spp_slices(64, 1024, 256) == 1
    assert h.sdirt_psf_spp_slices(0, 4096, 256) == 1 and h.sdirt_psf_spp_slices(64, 0, 256) == 1


def test_bench_defaults_to_strong_scaling_and_names_the_baseline_config():
    """`bench.py --gpus N` measures what SURVEY.md §8e states unless told otherwise: the ONE volume of the workload cut into
    N shards (a weak-scaling default would have put the first 8-GPU record on a 131072-point volume BASELINE
    does not name)."""
    import subprocess
    import sys
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    flat = " ".join(p.stdout.split())
    assert "--scaling {weak,strong}" in flat and "strong (default; SURVEY.md §8e" in flat and "--detail-file" in flat
    import bench
    assert bench.volume_points(1, "c2").shape == (16384, 3) and bench.volume_points(8, "c3").shape == (65536, 3)
    from sdirt_amd import dist as sd
    assert [b - a for a, b in sd.shard_bounds(16384, 8)] == [2048] * 8
