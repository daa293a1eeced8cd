"""The reference's fitting script end to end (1_fit_psfnet.py: refocus, write_lens_json, two lens reports
`analysis(...)`, load_net, train_psfnet, compare_psf) against this package through the `deeplens` import aliases
(sdirt_amd/compat), and the pieces of the lens report against what the reference itself computed (fixture F26,
oracle/gen_golden_analysis.py): the ray fans of the layout figure, magnification by ray mapping, RMS spot radii.
Needs an MI355X: `-m gpu`.
"""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import DATA, ROOT, load_golden, load_state, make_lens

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_layout_fans_against_the_reference(tmp_path):
    """plot_setup2D_with_trace (optics.py:1686-1742) at the script's near depth: the three fans as sampled
    (sample_point_source_2D), their recorded paths from the reference's own rays, the title, the files."""
    from sdirt_amd import Ray, analysis
    st, g = load_state("rf50mm"), load_golden("f26_rf50_analysis")
    lens = make_lens("rf50mm", DEV, st)
    depth = float(g["depth"])
    assert lens.aper_idx == int(g["aper_idx"])
    assert analysis.layout_title(lens) == str(g["title"])
    assert abs(lens.calc_eqfl() - float(g["eqfl"])) < 1e-9 * float(g["eqfl"])
    half = np.rad2deg(lens.hfov)
    for i, view in enumerate([0, half * 0.707, half * 0.99]):
        ray = lens.sample_point_source_2D(depth=depth, view=view, M=9, entrance_pupil=True, wvln=float(g[f"fan{i}_wvln"]))
        assert np.abs(ray.o.cpu().numpy() - g[f"fan{i}_o"]).max() <= 2e-6
        assert np.abs(ray.d.cpu().numpy() - g[f"fan{i}_d"]).max() <= 2e-7
        hand = Ray.from_normalized(g[f"fan{i}_o"], g[f"fan{i}_d"], wvln=float(g[f"fan{i}_wvln"]), device=DEV)
        _, oss = lens.trace2sensor(ray=hand, record=True)
        assert np.array_equal(hand.ra.cpu().numpy(), g[f"fan{i}_ra"])
        worst = 0.0
        for path, n, want in zip(oss, g[f"fan{i}_len"], g[f"fan{i}_pts"]):
            assert len(path) == int(n)
            worst = max(worst, float(np.abs(np.stack(path) - want[:n]).max()))
        print(f"fan {i}: view {view:.2f} deg, recorded paths within {worst:.1e} mm of the reference's")
        assert worst <= 2e-5
    fans = lens.plot_setup2D_with_trace(filename=str(tmp_path / "layout"), entrance_pupil=True, depth=depth)
    assert len(fans) == 3 and os.path.getsize(tmp_path / "layout.png") > 10000
    lens.plot_setup2D_with_trace(filename=str(tmp_path / "multi"), depth=None, multi_plot=True)
    assert os.path.getsize(tmp_path / "multi.png") > 10000 and os.path.getsize(tmp_path / "multi.svg") > 10000


def test_magnification_and_rms_against_the_reference():
    """calc_magnification3 / calc_scale_ray (optics.py:1237-1321) and analysis_rms (optics.py:2103-2140) under
    the reference's seeds with the samples drawn on the CPU generator as its CPU run drew them: the first
    sample plane of each bundle, then the figures themselves (21 x 21 x 512 and 4 x 31 x 31 x 2048 rays through
    Lensgroup.trace as [spp, M, M] batches)."""
    from sdirt_amd.utils import set_seed
    st, g = load_state("rf50mm"), load_golden("f26_rf50_analysis")
    lens = make_lens("rf50mm", DEV, st)
    lens.sample_rng_device = "cpu"
    depth = float(g["depth"])
    set_seed(int(g["seed_mag"]))
    ray = lens.sample_point_source(M=21, spp=512, depth=depth, R=-depth * np.tan(lens.hfov) * 0.5, pupil=True)
    assert ray.shape == (512, 21, 21)
    assert np.abs(ray.o[0].cpu().numpy() - g["mag_o0"]).max() <= 1e-5
    assert np.abs(ray.d[0].cpu().numpy() - g["mag_d0"]).max() <= 5e-7        # MKL sin / cos differ by CPU model
    set_seed(int(g["seed_mag"]))
    mag = lens.calc_magnification3(depth)
    set_seed(int(g["seed_mag"]))
    scale = lens.calc_scale_ray(depth)
    print(f"magnification {mag:.8f} (reference {float(g['mag']):.8f}), scale {scale:.6f} ({float(g['scale']):.6f})")
    assert abs(mag - float(g["mag"])) <= 2e-6 * float(g["mag"])
    assert abs(scale - float(g["scale"])) <= 2e-6 * float(g["scale"])
    for ref, key in ((True, "rms"), (False, "rms_own")):
        set_seed(int(g["seed_rms"]))
        got = np.asarray([float(v) for v in lens.analysis_rms(depth=depth, ref=ref)])
        print(f"analysis_rms(ref={ref}): {got} mm (reference {g[key]})")
        assert np.abs(got - g[key]).max() <= 2e-5 * g[key].max()
    # the default: samples from the device generator, as the reference does on a GPU -- same optics, other draws
    lens.sample_rng_device = None
    torch.manual_seed(1)
    other = np.asarray([float(v) for v in lens.analysis_rms(depth=depth)])
    assert np.abs(other - g["rms"]).max() <= 1e-2 * g["rms"].max()    # Monte-Carlo noise of the one corner source (2048 rays)


SCRIPT = r'''
# What the reference's fitting script does with the package (1_fit_psfnet.py:9-40), as a user of the import
# aliases would write it: build the lens + network object, refocus to 1 m, dump the prescription, two lens reports
# (near and far object), load a checkpoint, fit, compare.  Shortened: 40 iterations instead of 90000; the checkpoint
# is written here first because the reference's ./ckpt file is not distributed.
import os
import sys

import torch
from deeplens.psfnet import PSFNet
from deeplens.utils import set_logger, set_seed

out_dir, prescription = sys.argv[1:3]
os.makedirs(out_dir, exist_ok=True)
set_logger(out_dir)
set_seed(0)

KS = 21
net = PSFNet(filename=prescription, sensor_res=(512, 768), kernel_size=KS, device="cuda")
sensor_z = net.d_sensor
net.refocus(-1000 + sensor_z)
net.write_lens_json(os.path.join(out_dir, "lens.json"))
print(net.d_sensor)

for object_z in (-500 + sensor_z, -20000 + sensor_z):
    net.analysis(save_name=os.path.join(out_dir, str(int(object_z))), depth=object_z, ks=KS)

checkpoint = os.path.join(out_dir, "start.pkl")
torch.save(net.psfnet.state_dict(), checkpoint)
net.load_net(checkpoint)
net.train_psfnet(iters=40, bs=64, lr=1e-4, spp=20000, evaluate_every=20, result_dir=out_dir)

os.chdir(out_dir)
net.compare_psf()
print("Finish PSF net fitting.")
'''


def test_fitting_script_runs_through_the_import_aliases(tmp_path):
    from sdirt_amd import compat
    script = tmp_path / "fit.py"
    script.write_text(SCRIPT)
    out = tmp_path / "results"
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([compat.path(), ROOT, os.environ.get("PYTHONPATH", "")]),
               MPLBACKEND="Agg")
    run = subprocess.run([sys.executable, str(script), str(out), os.path.join(DATA, "rf50mm.json")], env=env,
                         capture_output=True, text=True, timeout=900)
    log_dir = os.environ.get("SDIRT_TEST_LOG_DIR")
    if log_dir:
        with open(os.path.join(log_dir, "fit_script.log"), "w") as f:
            f.write(run.stdout + "\n--- stderr ---\n" + run.stderr[-4000:])
    assert run.returncode == 0, run.stderr[-3000:]
    assert run.stdout.count("On-axis RMS radius") == 2 and "Finish PSF net fitting." in run.stdout
    have = {os.path.basename(p) for p in glob.glob(str(out / "*"))}
    print(sorted(have))
    for name in ("lens.json", "output.log", "iter20.png", "iter40.png", "iter20_PSFNet_mlp.pkl", "PSFNet_mlp.pkl"):
        assert name in have, name
    for d_ori in (-500, -20000):
        for tag in ("v00", "v04", "v08"):
            assert f"rt_{d_ori}_{tag}.png" in have and f"pred_{d_ori}_{tag}.png" in have
    assert len([n for n in have if n.endswith("mm_left.png")]) == 2         # the two PSF maps
    assert len([n for n in have if n.endswith(".png") and n.lstrip("-").split(".")[0].isdigit()]) == 2   # layouts
    assert "1, " in open(out / "output.log").read() or "19, " in open(out / "output.log").read()
