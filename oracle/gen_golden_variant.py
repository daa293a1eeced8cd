#!/usr/bin/env python3
"""Fixture F11: the reference on a VARIANT of the rf50mm prescription that reaches the branches
the two shipped lenses never take -- a conic constant <= -1 (the domain test of
surfaces.py:727-743 without the `r^2 < 1/(c^2 (1+k))` term), an ellipsoidal asphere with a
non-zero r^2 coefficient, and a flat refracting surface (surfaces.py:409-425 with eta != 1).

TEST INFRASTRUCTURE ONLY -- build container only.  Writes, next to the other fixtures,
  lens_state_rf50mm_variant.json   the scalars the hot path reads (as for the real lenses)
  lens_rf50mm_variant.json         the same prescription in this package's own schema
  f11_rf50_variant_pts4.npz        one psf_diff call with every per-surface checkpoint
"""
import argparse
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402  (imports the reference)

# surface index -> overrides, in the reference's lens_web.json vocabulary
EDITS = {8: {"k": -1.8}, 9: {"k": 0.6, "ai2": 1.0e-4}, 10: {"c": 0.0, "roc": 0.0}}


def variant_reference_json(path):
    with open("/root/reference/lenses/rf50mm/lens_web.json") as f:
        d = json.load(f)
    for i, e in EDITS.items():
        s = d["surfaces"][i]
        s.update(e)
        if "ai2" in e:
            s["ai"][0] = e["ai2"]
    with open(path, "w") as f:
        json.dump(d, f)


def variant_own_json(path):
    with open(os.path.join(HERE, "..", "sdirt_amd", "data", "rf50mm.json")) as f:
        d = json.load(f)
    d["name"] = "rf50mm_variant"
    for i, e in EDITS.items():
        s = d["surfaces"][i]
        if "k" in e:
            s["conic"] = e["k"]
        if "ai2" in e:
            s["even_asphere"][0] = e["ai2"]
        if "c" in e:
            s["curvature"] = e["c"]
            if e["c"] == 0.0:
                s["kind"] = "plane"
    with open(path, "w") as f:
        json.dump(d, f, indent=1)


def build(refreeze=False):
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "rf50mm_variant"))
    path = os.path.join(tmp, "rf50mm_variant", "lens_web.json")
    variant_reference_json(path)
    gg.set_seed(0)
    lens = gg.PSFNet(filename=path, sensor_res=(512, 768), kernel_size=21, device="cpu")
    lens.refocus(-1000 + lens.d_sensor)
    # the variant's pupils, hfov, foclen and fnum drift run to run like the shipped lenses' (hfov
    # 0.36507961 fresh against 0.36507955 committed): frozen like theirs
    return gg.freeze(lens, "rf50mm_variant", refreeze)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(HERE, "..", "tests", "golden"))
    ap.add_argument("--refreeze", action="store_true", help="record this run's pupils / hfov / foclen / fnum as the frozen ones")
    args = ap.parse_args()
    out_dir = os.path.abspath(args.out)
    lens = build(args.refreeze)
    st = gg.lens_state(lens, [0.589] + list(gg.WAVE_RGB))
    st["lens_name"] = "rf50mm_variant"
    with open(os.path.join(out_dir, "lens_state_rf50mm_variant.json"), "w") as f:
        json.dump(st, f, indent=1)
    variant_own_json(os.path.join(out_dir, "lens_rf50mm_variant.json"))
    pts4 = [[0.0, 0.0, -300.0], [0.0, 0.0, -20000.0], [0.95, -0.9, -300.0], [-0.98, 0.98, -20000.0]]
    gg.save(out_dir, "f11_rf50_variant_pts4", gg.twice(lambda: gg.run_psf_case(
        lens, pts4, ks=33, spp=64, wvln=0.589, seed=11)))
    print("kinds:", [s["kind"] for s in st["surfaces"]], "d_sensor", st["d_sensor"])


if __name__ == "__main__":
    main()
