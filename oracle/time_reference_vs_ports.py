#!/usr/bin/env python3
"""Is the CPU baseline bench.py reports at least as fast as the reference itself?  ("cross-checked ...
so the restatement is shown to be no slower than the reference".)

TEST INFRASTRUCTURE, build container only (imports the reference from /root/reference; nothing of it travels).
Times, on the SAME machine, with the SAME thread count, on the SAME inputs -- 256 points of the config-2 volume
(every 64th: all 16 depth planes), 4096 spp (+2048 chief-ray rays per point), 65 x 65:

  * reference      Lensgroup.psf_diff of the reference (deeplens/optics.py:934-996; L grid only, param_list=None)
  * torch port     oracle/torch_port.py (bench.py's `cpu_baseline.torch`): the same whole-tensor op sequence, L + R
  * C port         oracle/sdirt_oracle.c with OpenMP (bench.py's `cpu_baseline.value`), L + R

  python oracle/time_reference_vs_ports.py [--threads 8] [--repeat 3]   -> profiles/r04/time_reference_vs_ports.txt
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=len(os.sched_getaffinity(0)))
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--points", type=int, default=256)
    args = ap.parse_args()
    from _refimport import import_reference
    PSFNet, set_seed, _ = import_reference(num_threads=args.threads)
    import bench
    from conftest import load_state
    from oracle import oracle as orc
    from oracle import torch_port as tp
    N, S, KS, DP = args.points, 4096, 65, [0.78, 1.44, 0.3, 0.5]
    pts_all = bench.volume_points(1, "c2")
    pts = pts_all[:: len(pts_all) // N][:N].contiguous()

    # --- the reference itself
    set_seed(0)
    ref = PSFNet(filename="/root/reference/lenses/rf50mm/lens_web.json", sensor_res=(512, 768), kernel_size=21, device="cpu")
    ref.refocus(-1000 + ref.d_sensor)
    torch.set_num_threads(args.threads)

    def run_ref():
        set_seed(1)
        t0 = time.perf_counter()
        ref.psf_diff(points=pts, wvln=0.589, ks=KS, spp=S)
        return time.perf_counter() - t0

    # --- the two ports, on the lens state the fixtures pin (the reference's own scalars) and their own pupil points
    st = load_state("rf50mm")
    g = torch.Generator().manual_seed(1)
    u = torch.rand(2, S, generator=g).numpy()
    uc = torch.rand(2, 2048, generator=g).numpy()
    x2, y2 = orc.pupil_samples(u[0], u[1], st["pupil_r"])
    xc, yc = orc.pupil_samples(uc[0], uc[1], st["pupil_r"] * 0.25)
    orc.set_num_threads(args.threads)

    def run_c():
        t0 = time.perf_counter()
        orc.psf(st, pts.numpy(), x2, y2, xc, yc, KS, dp=DP)
        return time.perf_counter() - t0

    def run_t():
        t0 = time.perf_counter()
        tp.psf(st, pts.numpy(), x2, y2, xc, yc, KS, dp=DP, chunk=256)
        return time.perf_counter() - t0

    out = []
    for name, fn in (("reference Lensgroup.psf_diff (L only)", run_ref), ("oracle/torch_port.py (L + R)", run_t),
                     ("oracle/sdirt_oracle.c, OpenMP (L + R)", run_c)):
        fn()                                            # warm-up (thread pools, allocator)
        ts = [fn() for _ in range(args.repeat)]
        best = min(ts)
        out.append((name, best, N * S / best))
    cpu = next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "unknown")
    lines = [f"{N} points x {S} spp (+2048 chief-ray rays each), ks {KS}, rf50mm at 1 m; {args.threads} threads; best of "
             f"{args.repeat}; {cpu}; torch {torch.__version__}"]
    for name, best, rate in out:
        lines.append(f"  {name:42s} {best:7.2f} s   {rate / 1e6:6.2f} M primary rays/s   {rate / out[0][2]:5.2f} x the reference")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
