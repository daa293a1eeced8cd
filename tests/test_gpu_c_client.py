"""A C host program on the C ABI, no Python in the process that renders ("callable without Python"):
tests/c_client/psf_client.c -- sdirt_lens_create, sdirt_points_to_object, sdirt_psf_call with the trip rule evaluated on
the device, corrected tables taken from the control block -- built here with the C compiler, run as a child process on
fixture F1 (the reference's own run of BASELINE config 1) and on a 12-point batch of fixture F2's lens, and compared with
the reference's PSF and with the oracle run on the same pupil points."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_golden, load_state

pytestmark = pytest.mark.gpu
KIND = {"plane": 0, "sphere": 1, "asphere": 2}


def build_client(tmp_path):
    exe = str(tmp_path / "psf_client")
    src = os.path.join(ROOT, "tests", "c_client", "psf_client.c")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cc = shutil.which("gcc") or shutil.which("cc")
    common = ["-std=c99", "-Wall", "-Wextra", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(rocm, "include"), "-I",
              os.path.join(ROOT, "include"), src, "-L", os.path.join(ROOT, "sdirt_amd"), "-lsdirt_dp", "-L",
              os.path.join(rocm, "lib"), "-lamdhip64", f"-Wl,-rpath,{os.path.join(ROOT, 'sdirt_amd')}",
              f"-Wl,-rpath,{os.path.join(rocm, 'lib')}", "-o", exe]
    cmd = [cc] + common if cc else [os.path.join(rocm, "bin", "hipcc"), "-x", "c"] + common
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0 and not out.stderr.strip(), out.stderr        # builds without a warning
    return exe


def write_input(path, st, points, uniforms, spp, ks, dp=None, wvln=0.589):
    surf = b""
    key = repr(float(wvln))
    for s in st["surfaces"]:
        ai = list(s["ai"]) + [0.0] * (8 - len(s["ai"]))
        # sdirt_surface_desc: int32 kind, ai_degree; double r; float d, c, k, ai[8]; (4 bytes padding); double n1, n2
        surf += struct.pack("<iid3f8f4xdd", KIND[s["kind"]], len(s["ai"]), s["r"], s["d"], s["c"], s["k"], *ai,
                            s["n1"][key], s["n2"][key])
    K, N = len(st["surfaces"]), len(points)
    assert len(surf) == 80 * K
    hdr = struct.pack("<8i9d4d", 0x53444952, K, N, spp, 2048, ks, 1 if dp else 0, 0,
                      st["pupil_r"], st["pupil_r"] * 0.25, st["pupil_z"], st["d_sensor"], st["pixel_size"],
                      float(np.tan(st["hfov"])), st["r_last"], st["sensor_size"][1], st["sensor_size"][0],
                      *(dp or [0.78, 1.44, 0.3, 0.5]))
    with open(path, "wb") as f:
        f.write(hdr + surf + np.ascontiguousarray(points, np.float32).tobytes()
                + np.ascontiguousarray(uniforms, np.float32).tobytes())


def read_output(path, K, N, ks):
    raw = open(path, "rb").read()
    rounds = struct.unpack_from("<i", raw, 0)[0]
    off = 4
    trips = np.frombuffer(raw, np.int32, K, off); off += 4 * K
    trips_c = np.frombuffer(raw, np.int32, K, off); off += 4 * K
    cen = np.frombuffer(raw, np.float32, 2 * N, off).reshape(N, 2); off += 8 * N
    L = np.frombuffer(raw, np.float32, N * ks * ks, off).reshape(N, ks, ks); off += 4 * N * ks * ks
    R = np.frombuffer(raw, np.float32, N * ks * ks, off).reshape(N, ks, ks)
    return rounds, trips, trips_c, cen, L, R


def test_header_struct_size_matches_what_the_test_packs():
    """sizeof(sdirt_surface_desc) == 80 and sizeof(struct client_header) == 136, as write_input packs them."""
    assert struct.calcsize("<iid3f8f4xdd") == 80 and struct.calcsize("<8i9d4d") == 136


def test_c_program_renders_config1_like_the_reference(tmp_path, oracle):
    exe = build_client(tmp_path)
    st, g = load_state("rf50mm"), load_golden("f1_rf50_c1")
    S, ks = int(g["spp"]), int(g["ks"])
    u = np.concatenate([g["u_theta"], g["u_r2"], g["uc_theta"], g["uc_r2"]])
    write_input(tmp_path / "in.bin", st, g["points"], u, S, ks)
    p = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    K = len(st["surfaces"])
    rounds, trips, trips_c, cen, L, R = read_output(tmp_path / "out.bin", K, 1, ks)
    # the device's rule lands on the reference's batch-wide trip counts: from 10 everywhere in one correction round
    assert np.array_equal(trips, g["trips"]) and np.array_equal(trips_c, g["trips_center"]), (trips, trips_c)
    assert rounds == 2
    assert np.abs(cen - g["center"]).max() < 4e-6                       # mm, vs the reference's chief-ray centre
    assert np.abs(L - g["psf"]).max() < 4e-6, np.abs(L - g["psf"]).max()   # vs the reference's PSF (smoke: 2.5e-6)
    assert not R.any()                                                  # param_list=None leaves R all-zero
    # vs the oracle on the pupil points the device mapped from the same uniforms
    x2, y2 = oracle.pupil_samples(g["u_theta"], g["u_r2"], st["pupil_r"])
    xc, yc = oracle.pupil_samples(g["uc_theta"], g["uc_r2"], st["pupil_r"] * 0.25)
    lo, _, co, ok = oracle.psf(st, g["points"], x2, y2, xc, yc, ks)
    assert ok and np.abs(L - lo).max() < 2e-6 and np.abs(cen - co).max() < 1e-6
    print(f"psf_client vs reference {np.abs(L - g['psf']).max():.2e}, vs oracle {np.abs(L - lo).max():.2e}; {p.stdout.strip()}")


def test_c_program_on_a_batch_with_both_subpixels(tmp_path, oracle):
    """Twelve points (the four of fixture F2 at three depths), 4096 spp, 33 x 33 L and R: the C program against the
    oracle on the same pupil points -- trip tables equal the oracle's batch-wide counts."""
    exe = build_client(tmp_path)
    st, g = load_state("rf50mm"), load_golden("f2_rf50_pts4")
    pts = np.concatenate([g["points"] * np.array([1, 1, s], np.float32) for s in (1.0, 0.5, 2.0)])
    S, ks, dp = 4096, 33, [0.78, 1.44, 0.3, 0.5]
    rng = np.random.default_rng(5)
    u = rng.random(2 * S + 2 * 2048, dtype=np.float32)
    write_input(tmp_path / "in.bin", st, pts, u, S, ks, dp=dp)
    p = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    K = len(st["surfaces"])
    rounds, trips, trips_c, cen, L, R = read_output(tmp_path / "out.bin", K, len(pts), ks)
    x2, y2 = oracle.pupil_samples(u[:S], u[S:2 * S], st["pupil_r"])
    xc, yc = oracle.pupil_samples(u[2 * S:2 * S + 2048], u[2 * S + 2048:], st["pupil_r"] * 0.25)
    lo, ro, co, ok, tp, tc = oracle.psf(st, pts, x2, y2, xc, yc, ks, dp=dp, return_trips=True)
    assert ok and np.array_equal(trips, tp) and np.array_equal(trips_c, tc), (trips, tp, trips_c, tc)
    assert ulp(cen, co) <= 1
    # 4096 samples per point: the whole-batch bar of tests/test_gpu_full_size.py (measured there: <= 3.9e-6)
    assert np.abs(L - lo).max() < 5e-6 and np.abs(R - ro).max() < 5e-6, (np.abs(L - lo).max(), np.abs(R - ro).max())
    print(f"psf_client, 12 points x 4096 spp vs oracle: {np.abs(L - lo).max():.2e} / {np.abs(R - ro).max():.2e}, rounds {rounds}")
    assert rounds <= 3


def ulp(a, b):
    from conftest import ulp_diff
    return int(ulp_diff(a, b).max())
