#!/usr/bin/env python3
"""Kernel A/B harness: times k_chief_center and k_psf_lr on the config-2 volume
through the C ABI with a FIXED trip table and prints a checksum of the outputs,
so that kernel variants can be compared for speed and for bit-identical results.

  SDIRT_AMD_LIB=path/to/variant.so python tools/kbench.py [--reps 10] [--n 16384]
"""
import argparse
import ctypes as C
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--n", type=int, default=16384)
    ap.add_argument("--spp", type=int, default=4096)
    ap.add_argument("--ks", type=int, default=65)
    ap.add_argument("--lens", default="rf50mm")
    ap.add_argument("--flags", type=int, default=0, help="4 = SDIRT_PSF_STRICT_IEEE")
    ap.add_argument("--order", default="volume", choices=("volume", "shuffle", "sorted_depth", "same"),
                    help="order of the points in the launch: volume (z-major, as bench.py), shuffle (random permutation), "
                         "same (every point = the first one: equal work per workgroup)")
    ap.add_argument("--trips", type=int, default=None,
                    help="this many Newton trips on EVERY curved surface of both passes (for the per-trip "
                         "cost: time two values and divide the difference)")
    args = ap.parse_args()
    from conftest import load_state, make_lens
    from sdirt_amd import _lib
    from sdirt_amd.basics import dptr, stream_ptr
    import bench
    dev = torch.device("cuda:0")
    st = load_state(args.lens)
    lens = make_lens(args.lens, "cuda:0", st)
    pts = bench.volume_points(1)[:: max(1, 16384 // args.n)][: args.n]
    if args.order == "shuffle":
        pts = pts[torch.randperm(pts.shape[0], generator=torch.Generator().manual_seed(5))]
    elif args.order == "same":
        pts = pts[pts.shape[0] // 2 + 17].expand_as(pts)
    pts = pts.contiguous().to(dev)
    po = lens._points_to_object(pts)
    g = torch.Generator().manual_seed(123)
    u = torch.rand(4, max(args.spp, 2048), generator=g).to(dev)
    xy = torch.empty((2, args.spp), device=dev); xyc = torch.empty((2, 2048), device=dev)
    h = _lib.lib()
    sp = stream_ptr(dev)
    _lib.check(h.sdirt_pupil_samples(dptr(u[0]), dptr(u[1]), args.spp, st["pupil_r"], dptr(xy[0]), dptr(xy[1]), sp))
    _lib.check(h.sdirt_pupil_samples(dptr(u[2]), dptr(u[3]), 2048, st["pupil_r"] * 0.25, dptr(xyc[0]), dptr(xyc[1]), sp))
    K = len(lens.surfaces)
    if args.lens == "rf50mm":
        tm, tc = [10, 3, 4, 3, 4, 0, 3, 3, 4, 4, 2, 3], [10, 3, 3, 3, 3, 0, 3, 3, 3, 3, 2, 3]
    else:
        tm = [10 if i == 0 else (0 if i == 7 else 3) for i in range(K)]; tc = tm
    if args.trips is not None:
        tm = tc = [args.trips if s.kind != 0 else 0 for s in lens.surfaces]
    trips, tripc = (C.c_int32 * K)(*tm), (C.c_int32 * K)(*tc)
    N, ks = pts.shape[0], args.ks
    cen = torch.empty((N, 2), device=dev)
    L = torch.empty((N, ks, ks), device=dev); R = torch.empty_like(L)
    mask = torch.zeros(64, dtype=torch.int32, device=dev)
    dp = _lib.DpParams(0.78, 1.44, 0.3, 0.5)
    hl = lens.dev_lens(0.589)

    def center():
        _lib.check(h.sdirt_chief_center(hl, dptr(po), N, dptr(xyc[0]), dptr(xyc[1]), 2048, st["pupil_z"],
                                        st["d_sensor"], tripc, args.flags & 4, dptr(cen), None, dptr(mask), sp))

    def psf():
        _lib.check(h.sdirt_psf_lr(hl, dptr(po), N, dptr(xy[0]), dptr(xy[1]), args.spp, st["pupil_z"],
                                  st["d_sensor"], st["pixel_size"], ks, dptr(cen), C.byref(dp), trips, 1 | args.flags,
                                  dptr(L), dptr(R), dptr(mask), sp))

    def timeit(fn):
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(args.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return float(np.median(ts)), float(np.min(ts))
    tcm, tcmin = timeit(center)
    tpm, tpmin = timeit(psf)
    torch.cuda.synchronize()
    sha = lambda t: hashlib.sha1(t.cpu().numpy().tobytes()).hexdigest()[:12]
    # L/R are sums of LDS float atomics: order-dependent in the last bits -> also print a robust digest
    print(f"lib={os.path.basename(_lib.LIB_PATH)} order={args.order} N={N} spp={args.spp} ks={ks} "
          f"center_ms={tcm:.3f} (min {tcmin:.3f}) psf_ms={tpm:.3f} (min {tpmin:.3f}) "
          f"total={tcm + tpm:.3f} cen_sha={sha(cen)} L_sum={L.double().sum().item():.6f} "
          f"R_sum={R.double().sum().item():.6f} mask={mask[:K].tolist()}")


if __name__ == "__main__":
    main()
