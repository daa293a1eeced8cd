#!/usr/bin/env python3
"""Kernel time of sdirt_psf_lr_centered (chief-ray pass + primary pass of a point in one workgroup) over batch shapes, by HIP
events, for A/Bs of how a launch ENDS (sdirt_psf.hip: prio_by_work_left; variants in tools/variants/prio_*.py) -- every
shape warmed for 0.3 s first: a chip that has just been idle holds lower clocks for its first ~100 ms of work, so the first
shape of a process and the first variant within a shape read 3-5 % high (what made round 6's first tail-split A/Bs look
better than they were, profiles/r06/tail_ab_map.txt).  Prints a digest of the results: variants must agree bit for bit
wherever the tiles are float64 (ks <= 49) and in centres and trip masks everywhere.

  [SDIRT_AMD_LIB=build/libsdirt_dp_<variant>.so] python tools/end_ab.py [--reps 20] [--shapes 2048:4096:65,16384:4096:65]

(The tail-split experiment itself -- tools/tail_ab.py with the tail_ws argument -- lives in commit acacd51.)
"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--shapes", default="2048:4096:65,4096:4096:65,16384:4096:65,8192:8192:21,1024:4096:65,1500:4096:65")
    ap.add_argument("--lens", default="rf50mm")
    ap.add_argument("--rounds", type=int, default=2, help="interleaved A/B rounds")
    args = ap.parse_args()
    from conftest import load_state, make_lens
    from sdirt_amd import _lib
    from sdirt_amd.basics import dptr, stream_ptr
    import bench
    dev = torch.device("cuda:0")
    st = load_state(args.lens)
    lens = make_lens(args.lens, "cuda:0", st)
    h = _lib.lib()
    sp = stream_ptr(dev)
    K = len(lens.surfaces)
    ncu = torch.cuda.get_device_properties(dev).multi_processor_count
    if args.lens == "rf50mm":
        tm, tc = [10, 3, 4, 3, 4, 0, 3, 3, 4, 4, 3, 3], [10, 3, 3, 3, 3, 0, 3, 3, 3, 4, 2, 3]
    else:
        tm = [10 if i == 0 else (0 if i == 7 else 3) for i in range(K)]; tc = tm
    trips, tripc = (C.c_int32 * K)(*tm), (C.c_int32 * K)(*tc)
    dp = _lib.DpParams(0.78, 1.44, 0.3, 0.5)
    hl = lens.dev_lens(0.589)
    all_pts = bench.volume_points(1)
    for shape in args.shapes.split(","):
        N, spp, ks = (int(v) for v in shape.split(":"))
        if N <= all_pts.shape[0]:
            pts = all_pts[:: max(1, all_pts.shape[0] // N)][:N]
        else:
            pts = all_pts.repeat((N + all_pts.shape[0] - 1) // all_pts.shape[0], 1)[:N]
        pts = pts.contiguous().to(dev)
        N = pts.shape[0]
        po = lens._points_to_object(pts)
        g = torch.Generator().manual_seed(123)
        u = torch.rand(4, max(spp, 2048), generator=g).to(dev)
        xy = torch.empty((2, spp), device=dev); xyc = torch.empty((2, 2048), device=dev)
        _lib.check(h.sdirt_pupil_samples(dptr(u[0]), dptr(u[1]), spp, st["pupil_r"], dptr(xy[0]), dptr(xy[1]), sp))
        _lib.check(h.sdirt_pupil_samples(dptr(u[2]), dptr(u[3]), 2048, st["pupil_r"] * 0.25, dptr(xyc[0]), dptr(xyc[1]), sp))
        out = {}
        for tail in (0,):
            out[tail] = dict(cen=torch.empty((N, 2), device=dev), L=torch.empty((N, ks, ks), device=dev),
                             R=torch.empty((N, ks, ks), device=dev), mask=torch.zeros((2, 64), dtype=torch.int32, device=dev),
                             anyv=torch.zeros(1, dtype=torch.int32, device=dev))

        def run(tail):
            o = out[tail]
            _lib.check(h.sdirt_psf_lr_centered(hl, hl, dptr(po), N, dptr(xy[0]), dptr(xy[1]), spp, dptr(xyc[0]), dptr(xyc[1]),
                                               2048, st["pupil_z"], st["d_sensor"], st["pixel_size"], ks, C.byref(dp), trips,
                                               tripc, 1, dptr(o["cen"]), dptr(o["anyv"]), dptr(o["L"]), dptr(o["R"]),
                                               dptr(o["mask"][0]), dptr(o["mask"][1]), sp))

        def timeit(tail):
            # ~0.3 s of launches first: a chip that has just been idle holds lower clocks for its first ~100 ms of work
            # (tools/clock_ramp.py) -- the first shape of a process, and within a shape the first variant, read 3-5 % high
            t_end = time.perf_counter() + 0.3
            while time.perf_counter() < t_end:
                for _ in range(10):
                    run(tail)
                torch.cuda.synchronize()
            ts = []
            for _ in range(args.reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); run(tail); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            return float(np.median(ts)), float(np.min(ts))
        t = [timeit(0) for _ in range(args.rounds)]
        out[0]["mask"].zero_(); out[0]["anyv"].zero_()
        run(0)
        torch.cuda.synchronize()
        o = out[0]
        import hashlib
        sha = lambda x: hashlib.sha1(x.cpu().numpy().tobytes()).hexdigest()[:12]
        print(f"lib={os.path.basename(_lib.LIB_PATH)} N={N} spp={spp} ks={ks} ms={float(np.median([x[0] for x in t])):.4f} "
              f"(min {min(x[1] for x in t):.4f}) | cen_sha={sha(o['cen'])} mask_sha={sha(o['mask'])} any_valid={int(o['anyv'])} "
              f"L_sha={sha(o['L'])} R_sha={sha(o['R'])} L_sum={o['L'].double().sum().item():.6f}", flush=True)


if __name__ == "__main__":
    main()
