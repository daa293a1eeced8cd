import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import make_lens, load_golden
lens = make_lens("rf50mm", "cuda:0")
g = load_golden("f8_rf50_mini_c2")
pts = torch.tensor(g["points"])
kw = dict(ks=65, spp=4096, pupil_xy=(g["pupil_x2"], g["pupil_y2"]), center_pupil_xy=(g["pupil_xc"], g["pupil_yc"]))
L0, R0 = lens.psf_lr(pts, **kw)
for pol in ("adaptive", "max"):
    lens.trip_policy = pol
    L1, R1 = lens.psf_lr(pts, **kw)
    d = (L0 - L1).abs().reshape(len(pts), -1).max(1).values
    print(pol, "max", d.max().item(), "median of per-PSF max", d.median().item(), "R max", (R0 - R1).abs().max().item())
